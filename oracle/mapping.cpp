// TEST INFRASTRUCTURE ONLY -- see mapping.hpp.
#include "mapping.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {
namespace {

constexpr int kThLow = 50, kHistoLength = 30;  // ORBmatcher::kThLow, HISTO_LENGTH

inline void quat_mul_f(const float a[4], const float b[4], float o[4]) {  // Eigen::Quaternionf product (x, y, z, w)
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
inline void quat_rot_f(const float q[4], const float v[3], float out[3]) {  // Eigen _transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
inline void quat_to_mat_f(const float q[4], float R[9]) {
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline SE3f se3_inverse(const SE3f& T) {  // Sophus: (q^-1, q^-1 * (t * -1))
    SE3f o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
    quat_rot_f(o.q, nt, o.t);
    return o;
}
inline SE3f se3_mul(const SE3f& a, const SE3f& b) {  // (qa qb, qa * tb + ta)
    SE3f o;
    quat_mul_f(a.q, b.q, o.q);
    float r[3];
    quat_rot_f(a.q, b.t, r);
    for (int c = 0; c < 3; ++c) o.t[c] = r[c] + a.t[c];
    return o;
}
inline void mat3_mul_f(const float* a, const float* b, float* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
}
inline bool mat3_inv_f(const float* m, float* o) {  // cofactor form (Eigen's fixed 3x3 inverse)
    const float c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
    const float det = (m[0] * c00 + m[1] * c10) + m[2] * c20;
    const float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c10 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c20 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}
inline int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 8; ++i) { uint32_t x, y; std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4); d += __builtin_popcount(x ^ y); }
    return d;
}
// F12 = K1^-T [t12]x R12 K2^-1 (Pinhole.cpp:118-121), the same camera on both sides
void fundamental(const CamF& cam, const float R12[9], const float t12[3], float F[9]) {
    const float K[9] = {cam.fx, 0.f, cam.cx, 0.f, cam.fy, cam.cy, 0.f, 0.f, 1.f};
    const float Kt[9] = {K[0], K[3], K[6], K[1], K[4], K[7], K[2], K[5], K[8]};
    float KtInv[9], KInv[9], tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f}, a[9], b[9];
    mat3_inv_f(Kt, KtInv);
    mat3_inv_f(K, KInv);
    mat3_mul_f(KtInv, tx, a);
    mat3_mul_f(a, R12, b);
    mat3_mul_f(b, KInv, F);
}
bool epipolar_ok(const float F[9], const KeyPoint& kp1, const KeyPoint& kp2, float unc) {
    const float a = kp1.x * F[0] + kp1.y * F[3] + F[6];
    const float b = kp1.x * F[1] + kp1.y * F[4] + F[7];
    const float c = kp1.x * F[2] + kp1.y * F[5] + F[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)unc;
}
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cc:2021-2062
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

// smallest-eigenvalue eigenvector of the 4 x 4 symmetric M (cyclic Jacobi, double)
void smallest_eigenvector4(double M[16], double v[4]) {
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += M[4 * p + q] * M[4 * p + q];
        if (off < 1e-40) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                const double apq = M[4 * p + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (M[4 * q + q] - M[4 * p + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; ++k) { const double a = M[4 * k + p], b = M[4 * k + q]; M[4 * k + p] = c * a - s * b; M[4 * k + q] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = M[4 * p + k], b = M[4 * q + k]; M[4 * p + k] = c * a - s * b; M[4 * q + k] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - s * b; V[4 * k + q] = s * a + c * b; }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k) if (M[5 * k] < M[5 * best]) best = k;
    for (int k = 0; k < 4; ++k) v[k] = V[4 * k + best];
}

}  // namespace

int SearchForTriangulation(const KeyFrameView& kf1, const KeyFrameView& kf2, const CamF& cam, const std::vector<float>& sf,
                           const std::vector<float>& sigma2, bool only_stereo, bool coarse, bool check_orientation, std::vector<int>& match12,
                           const uint8_t* has_point1) {
    if (!has_point1) has_point1 = kf1.has_point;
    const SE3f Tw1 = se3_inverse(kf1.Tcw), Tw2 = se3_inverse(kf2.Tcw);
    const float* Cw = Tw1.t;  // GetCameraCenter
    float C2[3];
    quat_rot_f(kf2.Tcw.q, Cw, C2);
    for (int c = 0; c < 3; ++c) C2[c] += kf2.Tcw.t[c];
    const float ep[2] = {cam.fx * C2[0] / C2[2] + cam.cx, cam.fy * C2[1] / C2[2] + cam.cy};
    const SE3f T12 = se3_mul(kf1.Tcw, Tw2);
    float R12[9], F12[9];
    quat_to_mat_f(T12.q, R12);
    fundamental(cam, R12, T12.t, F12);
    match12.assign(kf1.n, -1);
    int nmatches = 0;
    std::vector<int> rotHist[kHistoLength];
    const float factor = 1.0f / kHistoLength;
    int a = 0, b = 0;
    while (a < kf1.n_nodes && b < kf2.n_nodes) {
        if (kf1.fv_node[a] == kf2.fv_node[b]) {
            for (int i1 = kf1.fv_off[a]; i1 < kf1.fv_off[a + 1]; ++i1) {
                const int idx1 = kf1.fv_idx[i1];
                if (has_point1[idx1]) continue;
                const bool bStereo1 = kf1.u_right[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const KeyPoint& kp1 = kf1.keys[idx1];
                const uint8_t* d1 = kf1.desc + 32 * (size_t)idx1;
                int bestDist = kThLow, bestIdx2 = -1;
                for (int i2 = kf2.fv_off[b]; i2 < kf2.fv_off[b + 1]; ++i2) {
                    const int idx2 = kf2.fv_idx[i2];
                    if (kf2.has_point[idx2]) continue;  // vbMatched2 is never set in the reference
                    const bool bStereo2 = kf2.u_right[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = descriptor_distance(d1, kf2.desc + 32 * (size_t)idx2);
                    if (dist > kThLow || dist > bestDist) continue;
                    const KeyPoint& kp2 = kf2.keys[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ep[0] - kp2.x, distey = ep[1] - kp2.y;
                        if (distex * distex + distey * distey < 100 * sf[kp2.octave]) continue;
                    }
                    if (coarse || epipolar_ok(F12, kp1, kp2, sigma2[kp2.octave])) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1.angle - kf2.keys[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == kHistoLength) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            ++a; ++b;
        } else if (kf1.fv_node[a] < kf2.fv_node[b]) {
            a = (int)(std::lower_bound(kf1.fv_node, kf1.fv_node + kf1.n_nodes, kf2.fv_node[b]) - kf1.fv_node);
        } else {
            b = (int)(std::lower_bound(kf2.fv_node, kf2.fv_node + kf2.n_nodes, kf1.fv_node[a]) - kf2.fv_node);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, kHistoLength, ind1, ind2, ind3);
        for (int i = 0; i < kHistoLength; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { match12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int FuseSearch(const FrameView& kf, const SE3f& Tcw, const CamF& cam, float bf, const std::vector<float>& sf, const std::vector<float>& inv_sigma2,
               float log_scale_factor, const std::vector<MapPointView>& points, const std::vector<uint8_t>& valid, float th,
               std::vector<int>& best_idx, std::vector<int>& best_dist) {
    FeatureGrid grid;
    grid.init(kf.cols, kf.rows);
    grid.assign(kf.keys);
    const SE3f Twc = se3_inverse(Tcw);
    const float* Ow = Twc.t;
    const int nlevels = (int)sf.size();
    const int n = (int)points.size();
    best_idx.assign(n, -1);
    best_dist.assign(n, 256);
    int nFused = 0;
    for (int i = 0; i < n; ++i) {
        if (!valid[i]) continue;
        const MapPointView& M = points[i];
        float p3Dc[3];
        quat_rot_f(Tcw.q, M.pos, p3Dc);
        for (int c = 0; c < 3; ++c) p3Dc[c] += Tcw.t[c];
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];
        const float u = cam.fx * p3Dc[0] / p3Dc[2] + cam.cx, v = cam.fy * p3Dc[1] / p3Dc[2] + cam.cy;
        if (!(u >= grid.mnMinX && u < grid.mnMaxX && v >= grid.mnMinY && v < grid.mnMaxY)) continue;  // IsInImage
        const float ur = u - bf * invz;
        const float PO[3] = {M.pos[0] - Ow[0], M.pos[1] - Ow[1], M.pos[2] - Ow[2]};
        const float dist3D = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
        if (dist3D < M.min_dist || dist3D > M.max_dist) continue;
        const float dotn = (PO[0] * M.normal[0] + PO[1] * M.normal[1]) + PO[2] * M.normal[2];
        if ((double)dotn < 0.5 * (double)dist3D) continue;
        const float ratio = M.mfMaxDistance / dist3D;  // MapPoint::PredictScale (MapPoint.cc:540-555)
        int level = (int)std::ceil(std::log(ratio) / log_scale_factor);
        if (level < 0) level = 0; else if (level >= nlevels) level = nlevels - 1;
        const float radius = th * sf[level];
        const std::vector<size_t> cand = grid.GetFeaturesInArea(kf.keys, u, v, radius, -1, -1);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (size_t idx : cand) {
            const KeyPoint& kp = kf.keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < level - 1 || kpLevel > level) continue;
            if (kf.uRight[idx] >= 0) {
                const float ex = u - kp.x, ey = v - kp.y, er = ur - kf.uRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * inv_sigma2[kpLevel]) > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * inv_sigma2[kpLevel]) > 5.99) continue;
            }
            const int dist = descriptor_distance(M.desc, kf.desc.data() + 32 * idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        best_dist[i] = bestDist;
        if (bestDist <= kThLow) { best_idx[i] = bestIdx; nFused++; }
    }
    return nFused;
}

std::vector<NewMapPoint> CreateNewMapPoints(const KeyFrameView& cur, const std::vector<KeyFrameView>& neigh, const CamF& cam,
                                            const std::vector<float>& sf, const std::vector<float>& sigma2, const MappingParams& prm,
                                            bool coarse) {
    std::vector<NewMapPoint> out;
    std::vector<uint8_t> has1(cur.has_point, cur.has_point + cur.n);
    float Rcw1[9];
    quat_to_mat_f(cur.Tcw.q, Rcw1);
    const float* tcw1 = cur.Tcw.t;
    const SE3f Twc1 = se3_inverse(cur.Tcw);
    const float* Ow1 = Twc1.t;
    const float invfx = 1.0f / cam.fx, invfy = 1.0f / cam.fy;
    const float ratioFactor = 1.5f * prm.scale_factor;
    for (size_t j = 0; j < neigh.size(); ++j) {
        const KeyFrameView& kf2 = neigh[j];
        const SE3f Twc2 = se3_inverse(kf2.Tcw);
        const float* Ow2 = Twc2.t;
        const float vb[3] = {Ow2[0] - Ow1[0], Ow2[1] - Ow1[1], Ow2[2] - Ow1[2]};
        const float baseline = std::sqrt(vb[0] * vb[0] + vb[1] * vb[1] + vb[2] * vb[2]);
        if (baseline < prm.mb) continue;
        std::vector<int> match12;
        SearchForTriangulation(cur, kf2, cam, sf, sigma2, false, coarse, false, match12, has1.data());
        float Rcw2[9];
        quat_to_mat_f(kf2.Tcw.q, Rcw2);
        const float* tcw2 = kf2.Tcw.t;
        for (int idx1 = 0; idx1 < cur.n; ++idx1) {
            const int idx2 = match12[idx1];
            if (idx2 < 0) continue;
            const KeyPoint& kp1 = cur.keys[idx1];
            const KeyPoint& kp2 = kf2.keys[idx2];
            const float kp1_ur = cur.u_right[idx1], kp2_ur = kf2.u_right[idx2];
            const bool bStereo1 = kp1_ur >= 0, bStereo2 = kp2_ur >= 0;
            const float xn1[3] = {(kp1.x - cam.cx) / cam.fx, (kp1.y - cam.cy) / cam.fy, 1.f};
            const float xn2[3] = {(kp2.x - cam.cx) / cam.fx, (kp2.y - cam.cy) / cam.fy, 1.f};
            float ray1[3], ray2[3];
            for (int r = 0; r < 3; ++r) {  // Rwc * xn
                ray1[r] = (Rcw1[r] * xn1[0] + Rcw1[3 + r] * xn1[1]) + Rcw1[6 + r] * xn1[2];
                ray2[r] = (Rcw2[r] * xn2[0] + Rcw2[3 + r] * xn2[1]) + Rcw2[6 + r] * xn2[2];
            }
            const float dot = (ray1[0] * ray2[0] + ray1[1] * ray2[1]) + ray1[2] * ray2[2];
            const float n1 = std::sqrt((ray1[0] * ray1[0] + ray1[1] * ray1[1]) + ray1[2] * ray1[2]);
            const float n2 = std::sqrt((ray2[0] * ray2[0] + ray2[1] * ray2[1]) + ray2[2] * ray2[2]);
            const float cosParallaxRays = dot / (n1 * n2);
            float cosParallaxStereo = cosParallaxRays + 1, cosParallaxStereo1 = cosParallaxStereo, cosParallaxStereo2 = cosParallaxStereo;
            if (bStereo1) cosParallaxStereo1 = (float)std::cos(2 * std::atan2((double)(prm.mb / 2), (double)cur.depth[idx1]));
            else if (bStereo2) cosParallaxStereo2 = (float)std::cos(2 * std::atan2((double)(prm.mb / 2), (double)kf2.depth[idx2]));
            cosParallaxStereo = std::min(cosParallaxStereo1, cosParallaxStereo2);
            float x3D[3];
            bool bPointStereo = false;
            if (cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 &&
                (bStereo1 || bStereo2 || (cosParallaxRays < 0.9996 && prm.inertial) || (cosParallaxRays < 0.9998 && !prm.inertial))) {
                // GeometricTools::Triangulate
                float A[16];
                const float T1[12] = {Rcw1[0], Rcw1[1], Rcw1[2], tcw1[0], Rcw1[3], Rcw1[4], Rcw1[5], tcw1[1], Rcw1[6], Rcw1[7], Rcw1[8], tcw1[2]};
                const float T2[12] = {Rcw2[0], Rcw2[1], Rcw2[2], tcw2[0], Rcw2[3], Rcw2[4], Rcw2[5], tcw2[1], Rcw2[6], Rcw2[7], Rcw2[8], tcw2[2]};
                for (int c = 0; c < 4; ++c) {
                    A[c] = xn1[0] * T1[8 + c] - T1[c];
                    A[4 + c] = xn1[1] * T1[8 + c] - T1[4 + c];
                    A[8 + c] = xn2[0] * T2[8 + c] - T2[c];
                    A[12 + c] = xn2[1] * T2[8 + c] - T2[4 + c];
                }
                double M[16], v[4];
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double s = 0; for (int k = 0; k < 4; ++k) s += (double)A[4 * k + r] * (double)A[4 * k + c]; M[4 * r + c] = s; }
                smallest_eigenvector4(M, v);
                const float h[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                if (h[3] == 0) continue;
                for (int c = 0; c < 3; ++c) x3D[c] = h[c] / h[3];
            } else if (bStereo1 && cosParallaxStereo1 < cosParallaxStereo2) {
                bPointStereo = true;
                const float z = cur.depth[idx1];
                if (!(z > 0)) continue;
                const float xc[3] = {(kp1.x - cam.cx) * z * invfx, (kp1.y - cam.cy) * z * invfy, z};
                for (int r = 0; r < 3; ++r) x3D[r] = ((Rcw1[r] * xc[0] + Rcw1[3 + r] * xc[1]) + Rcw1[6 + r] * xc[2]) + Ow1[r];
            } else if (bStereo2 && cosParallaxStereo2 < cosParallaxStereo1) {
                bPointStereo = true;
                const float z = kf2.depth[idx2];
                if (!(z > 0)) continue;
                const float xc[3] = {(kp2.x - cam.cx) * z * invfx, (kp2.y - cam.cy) * z * invfy, z};
                for (int r = 0; r < 3; ++r) x3D[r] = ((Rcw2[r] * xc[0] + Rcw2[3 + r] * xc[1]) + Rcw2[6 + r] * xc[2]) + Ow2[r];
            } else {
                continue;
            }
            auto gate = [&](const float* Rcw, const float* tcw, const KeyPoint& kp, bool stereo, float ur, float sig) -> bool {
                const float z = ((Rcw[6] * x3D[0] + Rcw[7] * x3D[1]) + Rcw[8] * x3D[2]) + tcw[2];
                if (z <= 0) return false;
                const float x = ((Rcw[0] * x3D[0] + Rcw[1] * x3D[1]) + Rcw[2] * x3D[2]) + tcw[0];
                const float y = ((Rcw[3] * x3D[0] + Rcw[4] * x3D[1]) + Rcw[5] * x3D[2]) + tcw[1];
                const float invz = (float)(1.0 / (double)z);
                if (!stereo) {
                    const float u = cam.fx * x / z + cam.cx, v = cam.fy * y / z + cam.cy;  // Pinhole::project
                    const float ex = u - kp.x, ey = v - kp.y;
                    return !((double)(ex * ex + ey * ey) > 5.991 * (double)sig);
                }
                const float u = cam.fx * x * invz + cam.cx, u_r = u - prm.mbf * invz, v = cam.fy * y * invz + cam.cy;
                const float ex = u - kp.x, ey = v - kp.y, er = u_r - ur;
                return !((double)(ex * ex + ey * ey + er * er) > 7.8 * (double)sig);
            };
            if (!gate(Rcw1, tcw1, kp1, bStereo1, kp1_ur, sigma2[kp1.octave])) continue;
            if (!gate(Rcw2, tcw2, kp2, bStereo2, kp2_ur, sigma2[kp2.octave])) continue;
            const float d1[3] = {x3D[0] - Ow1[0], x3D[1] - Ow1[1], x3D[2] - Ow1[2]}, d2[3] = {x3D[0] - Ow2[0], x3D[1] - Ow2[1], x3D[2] - Ow2[2]};
            const float dist1 = std::sqrt((d1[0] * d1[0] + d1[1] * d1[1]) + d1[2] * d1[2]), dist2 = std::sqrt((d2[0] * d2[0] + d2[1] * d2[1]) + d2[2] * d2[2]);
            if (dist1 == 0 || dist2 == 0) continue;
            if (prm.far_points && (dist1 >= prm.th_far_points || dist2 >= prm.th_far_points)) continue;
            const float ratioDist = dist2 / dist1;
            const float ratioOctave = sf[kp1.octave] / sf[kp2.octave];
            if (ratioDist * ratioFactor < ratioOctave || ratioDist > ratioOctave * ratioFactor) continue;
            NewMapPoint np{idx1, (int)j, idx2, bPointStereo ? 1 : 0, {x3D[0], x3D[1], x3D[2]}};
            out.push_back(np);
            has1[idx1] = 1;  // mpCurrentKeyFrame->AddMapPoint(pMP, idx1): later neighbours skip this keypoint
        }
    }
    return out;
}

}  // namespace oracle
