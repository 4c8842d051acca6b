// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
// Restatement of SF/src/ORBextractor.cc; see orb.hpp for the parity note.
#include "orb.hpp"

namespace oracle {

const int8_t kOrbPattern[256 * 4] = {
#include "../tc2li-slam_amd/csrc/orb_pattern.inc"
};

// ORBextractor.cc:50-77
float IC_Angle(const Img& image, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const int step = image.stride;
    const uint8_t* center = image.row(cvRound(pty)) + cvRound(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
}

// ORBextractor.cc:80-120
static const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
void computeOrbDescriptor(const KeyPoint& kpt, const Img& img, const int8_t* pattern, uint8_t* desc) {
    float angle = (float)kpt.angle * factorPI;
    float a = (float)cosf(angle), b = (float)sinf(angle);
    const uint8_t* center = img.row(cvRound(kpt.y)) + cvRound(kpt.x);
    const int step = img.stride;
    auto value = [&](int idx) -> int {
        const float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
        return center[cvRound(px * b + py * a) * step + cvRound(px * a - py * b)];
    };
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int t0 = value(2 * k), t1 = value(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

// ORBextractor.cc:383-443
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST), scaleFactor(_scaleFactor) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mvBordered.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
        sumFeatures += mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// ORBextractor.cc:1143-1168
void ORBextractor::ComputePyramid(const Img& image) {
    for (int level = 0; level < nlevels; ++level) {
        float scale = mvInvScaleFactor[level];
        int sw = cvRound((float)image.w * scale), sh = cvRound((float)image.h * scale);
        if (level != 0) {
            Img resized(sw, sh);
            resizeLinearU8(mvImagePyramid[level - 1], resized);
            mvBordered[level] = makeBorder101(resized, EDGE_THRESHOLD);
        } else {
            mvBordered[level] = makeBorder101(image, EDGE_THRESHOLD);
        }
        Img view;
        view.w = sw; view.h = sh; view.stride = mvBordered[level].stride;
        view.p = mvBordered[level].store.data() + (size_t)EDGE_THRESHOLD * view.stride + EDGE_THRESHOLD;
        mvImagePyramid[level] = view;
    }
}

// ORBextractor.cc:454-510
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) {
    const int halfX = (int)ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = (int)ceil(static_cast<float>(BR.y - UL.y) / 2);
    n1.UL = UL;                        n1.UR = {UL.x + halfX, UL.y};
    n1.BL = {UL.x, UL.y + halfY};      n1.BR = {UL.x + halfX, UL.y + halfY};
    n2.UL = n1.UR;                     n2.UR = UR;
    n2.BL = n1.BR;                     n2.BR = {UR.x, UL.y + halfY};
    n3.UL = n1.BL;                     n3.UR = n1.BR;
    n3.BL = BL;                        n3.BR = {n1.BR.x, BL.y};
    n4.UL = n3.UR;                     n4.UR = n2.BR;
    n4.BL = n3.BR;                     n4.BR = BR;
    for (size_t i = 0; i < vKeys.size(); i++) {
        const KeyPoint& kp = vKeys[i];
        if (kp.x < n1.UR.x) {
            if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
            else n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
        else n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

// ORBextractor.cc:512-527
static bool compareNodes(std::pair<int, ExtractorNode*>& e1, std::pair<int, ExtractorNode*>& e2) {
    if (e1.first < e2.first) return true;
    if (e1.first > e2.first) return false;
    return e1.second->UL.x < e2.second->UL.x;
}

// ORBextractor.cc:529-753
std::vector<KeyPoint> ORBextractor::DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX,
                                                      int minY, int maxY, int N, int /*level*/) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
        ExtractorNode ni;
        ni.UL = {(int)(hX * static_cast<float>(i)), 0};
        ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
        ni.BL = {ni.UL.x, maxY - minY};
        ni.BR = {ni.UR.x, maxY - minY};
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint& kp = vToDistributeKeys[i];
        vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }
    bool bFinish = false;
    std::vector<std::pair<int, ExtractorNode*>> vSizeAndPointerToNode;
    vSizeAndPointerToNode.reserve(lNodes.size() * 4);

    auto addChild = [&](ExtractorNode& n, int* nToExpand) {
        if (n.vKeys.size() > 0) {
            lNodes.push_front(n);
            if (n.vKeys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) { lit++; continue; }
            ExtractorNode n1, n2, n3, n4;
            lit->DivideNode(n1, n2, n3, n4);
            addChild(n1, &nToExpand); addChild(n2, &nToExpand); addChild(n3, &nToExpand); addChild(n4, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<std::pair<int, ExtractorNode*>> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(), compareNodes);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    ExtractorNode n1, n2, n3, n4;
                    vPrev[j].second->DivideNode(n1, n2, n3, n4);
                    addChild(n1, nullptr); addChild(n2, nullptr); addChild(n3, nullptr); addChild(n4, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    std::vector<KeyPoint> vResultKeys;
    vResultKeys.reserve(nfeatures);
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
        std::vector<KeyPoint>& vNodeKeys = it->vKeys;
        KeyPoint* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++)
            if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

// ORBextractor.cc:755-870
void ORBextractor::ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.resize(nlevels);
    mvCandidates.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
        const Img& im = mvImagePyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KeyPoint> vToDistributeKeys;
        vToDistributeKeys.reserve(nfeatures * 10);
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                std::vector<KeyPoint> vKeysCell;
                const uint8_t* win = im.row((int)iniY) + (int)iniX;
                const int ww = (int)maxX - (int)iniX, wh = (int)maxY - (int)iniY;
                FAST9_16(win, im.stride, ww, wh, iniThFAST, true, vKeysCell);
                if (vKeysCell.empty()) FAST9_16(win, im.stride, ww, wh, minThFAST, true, vKeysCell);
                for (auto& kp : vKeysCell) {
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    vToDistributeKeys.push_back(kp);
                }
            }
        }
        mvCandidates[level] = vToDistributeKeys;
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                      mnFeaturesPerLevel[level], level);
        const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
        for (auto& kp : keypoints) {
            kp.x += minBorderX;
            kp.y += minBorderY;
            kp.octave = level;
            kp.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (auto& kp : allKeypoints[level]) kp.angle = IC_Angle(mvImagePyramid[level], kp.x, kp.y, umax);
}

// ORBextractor.cc:1060-1141
int ORBextractor::extract(const Img& image, std::vector<KeyPoint>& _keypoints, std::vector<uint8_t>& descriptors,
                          const int vLappingArea[2]) {
    if (image.w == 0 || image.h == 0) return -1;
    ComputePyramid(image);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    descriptors.assign((size_t)nkeypoints * 32, 0);
    _keypoints.assign(nkeypoints, KeyPoint());
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        if (keypoints.empty()) continue;
        Img workingMat = mvImagePyramid[level].clone();
        Img blurred(workingMat.w, workingMat.h);
        gaussianBlur7(workingMat, blurred);
        std::vector<uint8_t> desc(keypoints.size() * 32);
        for (size_t i = 0; i < keypoints.size(); i++)
            computeOrbDescriptor(keypoints[i], blurred, kOrbPattern, &desc[i * 32]);
        float scale = mvScaleFactor[level];
        int i = 0;
        for (auto& kp : keypoints) {
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            int dst;
            if (kp.x >= vLappingArea[0] && kp.x <= vLappingArea[1]) dst = stereoIndex--;
            else dst = monoIndex++;
            _keypoints[dst] = kp;
            std::memcpy(&descriptors[(size_t)dst * 32], &desc[(size_t)i * 32], 32);
            i++;
        }
    }
    return monoIndex;
}

}  // namespace oracle
