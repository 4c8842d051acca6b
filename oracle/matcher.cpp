// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
// See matcher.hpp for the reference locations.
#include "matcher.hpp"

namespace oracle {

int match_queries(const FrameView& F, const std::vector<ProjQuery>& queries, MatchMode mode, float mfNNratio,
                  std::vector<int>& match_of_query) {
    const int N = (int)F.keys.size();
    FeatureGrid grid;
    grid.init(F.cols, F.rows);
    grid.assign(F.keys);
    std::vector<uint8_t> taken(F.occupied);  // F.mvpMapPoints[idx] && Observations() > 0
    taken.resize(N, 0);
    match_of_query.assign(queries.size(), -1);
    int nmatches = 0;
    for (size_t q = 0; q < queries.size(); ++q) {
        const ProjQuery& Q = queries[q];
        if (!Q.valid) continue;
        const std::vector<size_t> vIndices = grid.GetFeaturesInArea(F.keys, Q.u, Q.v, Q.radius, Q.min_level, Q.max_level);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (taken[idx]) continue;
            if (F.uRight[idx] > 0) {
                const float er = std::fabs(Q.u_right - F.uRight[idx]);
                if (er > Q.radius) continue;
            }
            const int dist = DescriptorDistance(Q.desc, &F.desc[idx * 32]);
            if (mode == MATCH_BEST) {  // ORBmatcher.cc:1767-1771
                if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
            } else {  // ORBmatcher.cc:109-126
                if (dist < bestDist) {
                    bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.keys[idx].octave; bestIdx = (int)idx;
                } else if (dist < bestDist2) {
                    bestLevel2 = F.keys[idx].octave; bestDist2 = dist;
                }
            }
        }
        if (bestDist <= TH_HIGH) {
            if (mode == MATCH_RATIO) {
                if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
                if (!(bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2)) continue;
            }
            match_of_query[q] = bestIdx;
            if (Q.has_observations) taken[bestIdx] = 1;
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher.cc:2021-2062
static void ComputeThreeMaxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

int rotation_filter(const FrameView& F, const std::vector<ProjQuery>& queries, std::vector<int>& match_of_query) {
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (size_t q = 0; q < queries.size(); ++q) {
        if (match_of_query[q] < 0) continue;
        float rot = queries[q].angle - F.keys[match_of_query[q]].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back((int)q);
    }
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    int removed = 0;
    for (int i = 0; i < HISTO_LENGTH; i++)
        if (i != ind1 && i != ind2 && i != ind3)
            for (int q : rotHist[i]) { match_of_query[q] = -1; removed++; }
    return removed;
}

static inline void quat_rotate_f(const float q[4], const float v[3], float out[3]) {  // Eigen _transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static inline void matvec_f(const float R[9], const float v[3], float out[3]) {
    for (int r = 0; r < 3; ++r) out[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1]) + R[3 * r + 2] * v[2];
}
static inline void quat_to_matrix_f(const float q[4], float R[9]) {  // Eigen::Quaternionf::toRotationMatrix
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Sophus::SE3f::inverse().translation() = q^-1 * (t * -1)
static inline void inverse_translation_f(const SE3f& T, float out[3]) {
    const float qi[4] = {-T.q[0], -T.q[1], -T.q[2], T.q[3]};
    const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
    quat_rotate_f(qi, nt, out);
}

std::vector<ProjQuery> build_queries_last_frame(const SE3f& Tcw, const SE3f& Tlw, const CamF& cam, float mb, float mbf,
                                                const std::vector<float>& mvScaleFactors, int cols, int rows,
                                                const std::vector<uint8_t>& has_point, const std::vector<uint8_t>& outlier,
                                                const std::vector<float>& Xw, const std::vector<KeyPoint>& last_keys,
                                                const std::vector<uint8_t>& mp_desc, float th, bool bMono) {
    const int N = (int)last_keys.size();
    std::vector<ProjQuery> qs(N);
    // twc = Tcw.inverse().translation() = -(Rcw^T tcw); tlc = Tlw * twc
    float twc[3], tlc[3];
    inverse_translation_f(Tcw, twc);
    quat_rotate_f(Tlw.q, twc, tlc);  // Sophus: so3 * p + t
    for (int c = 0; c < 3; ++c) tlc[c] += Tlw.t[c];
    const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;
    const float mnMinX = 0.f, mnMaxX = (float)cols, mnMinY = 0.f, mnMaxY = (float)rows;
    for (int i = 0; i < N; i++) {
        if (!has_point[i] || outlier[i]) continue;
        float x3Dc[3];
        quat_rotate_f(Tcw.q, &Xw[3 * i], x3Dc);
        for (int c = 0; c < 3; ++c) x3Dc[c] += Tcw.t[c];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        const float u = cam.fx * x3Dc[0] / x3Dc[2] + cam.cx, v = cam.fy * x3Dc[1] / x3Dc[2] + cam.cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const int nLastOctave = last_keys[i].octave;
        ProjQuery& Q = qs[i];
        Q.radius = th * mvScaleFactors[nLastOctave];
        if (bForward) { Q.min_level = nLastOctave; Q.max_level = -1; }
        else if (bBackward) { Q.min_level = 0; Q.max_level = nLastOctave; }
        else { Q.min_level = nLastOctave - 1; Q.max_level = nLastOctave + 1; }
        Q.u = u; Q.v = v;
        Q.u_right = u - mbf * invzc;
        Q.angle = last_keys[i].angle;
        Q.valid = 1;
        std::memcpy(Q.desc, &mp_desc[(size_t)i * 32], 32);
    }
    return qs;
}

std::vector<ProjQuery> build_queries_local_map(const SE3f& Tcw, const CamF& cam, float mbf, const std::vector<float>& mvScaleFactors,
                                               float mfLogScaleFactor, int cols, int rows, const std::vector<MapPointView>& mps,
                                               float th, bool bFarPoints, float thFarPoints, float viewingCosLimit) {
    std::vector<ProjQuery> qs(mps.size());
    // Frame::UpdatePoseMatrices (Frame.cc:527-535): mRcw = Tcw.rotationMatrix(), mOw = Twc.translation()
    float mOw[3], mRcw[9];
    inverse_translation_f(Tcw, mOw);
    quat_to_matrix_f(Tcw.q, mRcw);
    const int nlevels = (int)mvScaleFactors.size();
    const bool bFactor = th != 1.0;
    for (size_t k = 0; k < mps.size(); ++k) {
        const MapPointView& M = mps[k];
        // Frame::isInFrustum, Nleft == -1 (Frame.cc:544-603)
        float Pc[3];
        matvec_f(mRcw, M.pos, Pc);
        for (int c = 0; c < 3; ++c) Pc[c] += Tcw.t[c];
        const float Pc_dist = std::sqrt((Pc[0] * Pc[0] + Pc[1] * Pc[1]) + Pc[2] * Pc[2]);
        const float PcZ = Pc[2];
        const float invz = 1.0f / PcZ;
        if (PcZ < 0.0f) continue;
        const float u = cam.fx * Pc[0] / Pc[2] + cam.cx, v = cam.fy * Pc[1] / Pc[2] + cam.cy;
        if (u < 0.f || u > (float)cols) continue;
        if (v < 0.f || v > (float)rows) continue;
        const float PO[3] = {M.pos[0] - mOw[0], M.pos[1] - mOw[1], M.pos[2] - mOw[2]};
        const float dist = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        if (dist < M.min_dist || dist > M.max_dist) continue;
        const float viewCos = ((PO[0] * M.normal[0] + PO[1] * M.normal[1]) + PO[2] * M.normal[2]) / dist;
        if (viewCos < viewingCosLimit) continue;
        // MapPoint::PredictScale (MapPoint.cc:540-555)
        const float ratio = M.mfMaxDistance / dist;
        int nScale = (int)std::ceil(std::log(ratio) / mfLogScaleFactor);
        if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
        // SearchByProjection(F, vpMapPoints, th, ...) window (ORBmatcher.cc:62-81)
        if (bFarPoints && Pc_dist > thFarPoints) continue;
        float r = viewCos > 0.998f ? 2.5f : 4.0f;  // RadiusByViewingCos
        if (bFactor) r *= th;
        ProjQuery& Q = qs[k];
        Q.u = u; Q.v = v;
        Q.u_right = u - mbf * invz;
        Q.radius = r * mvScaleFactors[nScale];
        Q.min_level = nScale - 1; Q.max_level = nScale;
        Q.valid = 1;
        std::memcpy(Q.desc, M.desc, 32);
    }
    return qs;
}

}  // namespace oracle
