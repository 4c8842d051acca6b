// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
#include "stereo.hpp"

namespace oracle {

// cv::norm(IL, IR, NORM_L1) on two 11x11 8-bit windows: exact integer sum of absolute differences
static inline int sadL1(const Img& a, int ax, int ay, const Img& b, int bx, int by, int n) {
    int s = 0;
    for (int y = 0; y < n; ++y) {
        const uint8_t* pa = a.row(ay + y) + ax;
        const uint8_t* pb = b.row(by + y) + bx;
        for (int x = 0; x < n; ++x) s += std::abs((int)pa[x] - (int)pb[x]);
    }
    return s;
}

StereoResult ComputeStereoMatches(const ORBextractor& left, const ORBextractor& right, const std::vector<KeyPoint>& mvKeys,
                                  const std::vector<uint8_t>& mDescriptors, const std::vector<KeyPoint>& mvKeysRight,
                                  const std::vector<uint8_t>& mDescriptorsRight, float mbf, float mb) {
    const int N = (int)mvKeys.size();
    StereoResult res;
    res.uRight.assign(N, -1.0f);
    res.depth.assign(N, -1.0f);
    res.bestDist.assign(N, -1);
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = left.mvImagePyramid[0].h;
    const std::vector<float>& mvScaleFactors = left.mvScaleFactor;
    const std::vector<float>& mvInvScaleFactors = left.mvInvScaleFactor;

    std::vector<std::vector<size_t>> vRowIndices(nRows);
    const int Nr = (int)mvKeysRight.size();
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = mvKeysRight[iR];
        const float kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[kp.octave];
        const int maxr = (int)ceil(kpY + r);
        const int minr = (int)floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);  // always in range for extractor output
    }

    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    vDistIdx.reserve(N);

    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = mvKeys[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = &mDescriptors[(size_t)iL * 32];
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = DescriptorDistance(dL, &mDescriptorsRight[iR * 32]);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = mvKeysRight[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Img& pl = left.mvImagePyramid[kpL.octave];
            const Img& pr = right.mvImagePyramid[kpL.octave];
            int bestDistS = INT_MAX, bestincR = 0;
            const int L = 5;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= pr.w) continue;
            for (int incR = -L; incR <= +L; incR++) {
                float dist = (float)sadL1(pl, (int)(scaleduL - w), (int)(scaledvL - w), pr, (int)(scaleduR0 + incR - w),
                                          (int)(scaledvL - w), 2 * w + 1);
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                res.depth[iL] = mbf / disparity;
                res.uRight[iL] = bestuR;
                res.bestDist[iL] = bestDistS;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return res;  // the reference indexes an empty vector here (Frame.cc:998): undefined
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        res.uRight[vDistIdx[i].second] = -1;
        res.depth[vDistIdx[i].second] = -1;
    }
    return res;
}

void FeatureGrid::init(int cols, int rows) {
    mnMinX = 0.0f; mnMaxX = (float)cols; mnMinY = 0.0f; mnMaxY = (float)rows;
    mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (mnMaxX - mnMinX);
    mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (mnMaxY - mnMinY);
}

void FeatureGrid::assign(const std::vector<KeyPoint>& keys) {
    for (auto& col : cell) for (auto& c : col) c.clear();
    for (size_t i = 0; i < keys.size(); i++) {
        const int posX = (int)std::round((keys[i].x - mnMinX) * mfGridElementWidthInv);
        const int posY = (int)std::round((keys[i].y - mnMinY) * mfGridElementHeightInv);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
        cell[posX][posY].push_back(i);
    }
}

std::vector<size_t> FeatureGrid::GetFeaturesInArea(const std::vector<KeyPoint>& keys, float x, float y, float r, int minLevel,
                                                   int maxLevel) const {
    std::vector<size_t> vIndices;
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<size_t>& vCell = cell[ix][iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const KeyPoint& kpUn = keys[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x, disty = kpUn.y - y;
                if (fabs(distx) < factorX && fabs(disty) < factorY) vIndices.push_back(vCell[j]);
            }
        }
    return vIndices;
}

}  // namespace oracle
