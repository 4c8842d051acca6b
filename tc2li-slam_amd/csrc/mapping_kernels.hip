// Kernels of the CreateNewMapPoints core:
//   k_tri_search   ORBmatcher::SearchForTriangulation (SF/src/ORBmatcher.cc:916-1150): one thread per (neighbour, feature-vector
//                  entry of the current keyframe); the features of the same vocabulary node in the neighbour are scanned in order
//   k_tri_points   the per-pair body of LocalMapping::CreateNewMapPoints (SF/src/LocalMapping.cc:497-723): parallax gates,
//                  GeometricTools::Triangulate or KeyFrame::UnprojectStereo, depth / reprojection / scale gates
// The reference lets a keypoint that received a point from an earlier neighbour drop out of the later neighbours; a pair's result
// does not depend on any other pair, so every (neighbour, keypoint) is evaluated here and the host keeps the first success.
#include <hip/hip_runtime.h>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "mapping_device.hpp"

namespace tc2li {

struct Kp { float x, y, angle; int octave; };
__device__ __forceinline__ Kp load_kp(const float* keys, int i) {
    const float* k = keys + 6 * (size_t)i;
    Kp o;
    o.x = k[0]; o.y = k[1]; o.angle = k[3]; o.octave = __float_as_int(k[5]);
    return o;
}

__global__ __launch_bounds__(256) void k_tri_search(MappingDev m) {
    const int j = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    const KfDev& k2 = m.neigh[j];
    const KfDev& k1 = m.cur;
    if (e >= k1.fv_off[k1.n_nodes] || k2.skip) return;
    // node of entry e: the last node whose offset is <= e
    int lo = 0, hi = k1.n_nodes - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (k1.fv_off[mid] <= e) lo = mid; else hi = mid - 1; }
    const int node = k1.fv_node[lo];
    const int idx1 = k1.fv_idx[e];
    if (k1.has_point[idx1]) return;
    const bool stereo1 = k1.u_right[idx1] >= 0;
    if (m.only_stereo && !stereo1) return;
    // the same node in the neighbour
    int a = 0, b = k2.n_nodes;
    while (a < b) { const int mid = (a + b) >> 1; if (k2.fv_node[mid] < node) a = mid + 1; else b = mid; }
    if (a >= k2.n_nodes || k2.fv_node[a] != node) return;
    const Kp kp1 = load_kp(k1.keys, idx1);
    const uint32_t* d1 = reinterpret_cast<const uint32_t*>(k1.desc) + 8 * (size_t)idx1;
    uint32_t q[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) q[w] = d1[w];
    const float* F = k2.F12;
    const float la = kp1.x * F[0] + kp1.y * F[3] + F[6];
    const float lb = kp1.x * F[1] + kp1.y * F[4] + F[7];
    const float lc = kp1.x * F[2] + kp1.y * F[5] + F[8];
    const float den = la * la + lb * lb;
    int bestDist = 50, bestIdx2 = -1;  // TH_LOW
    for (int i2 = k2.fv_off[a]; i2 < k2.fv_off[a + 1]; ++i2) {
        const int idx2 = k2.fv_idx[i2];
        if (k2.has_point[idx2]) continue;
        const bool stereo2 = k2.u_right[idx2] >= 0;
        if (m.only_stereo && !stereo2) continue;
        const uint32_t* d2 = reinterpret_cast<const uint32_t*>(k2.desc) + 8 * (size_t)idx2;
        int dist = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ d2[w]);
        if (dist > 50 || dist > bestDist) continue;
        const Kp kp2 = load_kp(k2.keys, idx2);
        if (!stereo1 && !stereo2) {
            const float dx = k2.ep[0] - kp2.x, dy = k2.ep[1] - kp2.y;
            if (dx * dx + dy * dy < 100 * m.scale_factors[kp2.octave]) continue;
        }
        bool ok = m.coarse != 0;
        if (!ok && den != 0) {  // Pinhole::epipolarConstrain
            const float num = la * kp2.x + lb * kp2.y + lc;
            const float dsqr = num * num / den;
            ok = (double)dsqr < 3.84 * (double)m.level_sigma2[kp2.octave];
        }
        if (ok) { bestIdx2 = idx2; bestDist = dist; }
    }
    if (bestIdx2 >= 0) m.match[(size_t)j * k1.n + idx1] = bestIdx2;
}

// eigenvector of the smallest eigenvalue of the symmetric 4 x 4 M (cyclic Jacobi, double) -- stands in for JacobiSVD<Matrix4f>
__device__ void smallest_eigenvector4(double (&M)[16], double (&v)[4]) {
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += M[4 * p + q] * M[4 * p + q];
        if (off < 1e-40) break;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = M[4 * p + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (M[4 * q + q] - M[4 * p + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double a = M[4 * k + p], b = M[4 * k + q]; M[4 * k + p] = c * a - s * b; M[4 * k + q] = s * a + c * b; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double a = M[4 * p + k], b = M[4 * q + k]; M[4 * p + k] = c * a - s * b; M[4 * q + k] = s * a + c * b; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - s * b; V[4 * k + q] = s * a + c * b; }
            }
    }
    // the diagonal entries through a select chain (no indexed register array)
    double best = M[0];
    int bi = 0;
    if (M[5] < best) { best = M[5]; bi = 1; }
    if (M[10] < best) { best = M[10]; bi = 2; }
    if (M[15] < best) { best = M[15]; bi = 3; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = bi == 0 ? V[4 * k] : (bi == 1 ? V[4 * k + 1] : (bi == 2 ? V[4 * k + 2] : V[4 * k + 3]));
}

__device__ __forceinline__ bool reproj_gate(const MappingDev& m, const float* Rcw, const float* tcw, const float x3D[3], const Kp& kp, bool stereo,
                                            float ur, float sig) {
    const float z = ((Rcw[6] * x3D[0] + Rcw[7] * x3D[1]) + Rcw[8] * x3D[2]) + tcw[2];
    if (z <= 0) return false;
    const float x = ((Rcw[0] * x3D[0] + Rcw[1] * x3D[1]) + Rcw[2] * x3D[2]) + tcw[0];
    const float y = ((Rcw[3] * x3D[0] + Rcw[4] * x3D[1]) + Rcw[5] * x3D[2]) + tcw[1];
    const float invz = (float)(1.0 / (double)z);
    if (!stereo) {
        const float u = m.fx * x / z + m.cx, v = m.fy * y / z + m.cy;
        const float ex = u - kp.x, ey = v - kp.y;
        return !((double)(ex * ex + ey * ey) > 5.991 * (double)sig);
    }
    const float u = m.fx * x * invz + m.cx, u_r = u - m.mbf * invz, v = m.fy * y * invz + m.cy;
    const float ex = u - kp.x, ey = v - kp.y, er = u_r - ur;
    return !((double)(ex * ex + ey * ey + er * er) > 7.8 * (double)sig);
}

__global__ __launch_bounds__(128) void k_tri_points(MappingDev m) {
    const int j = blockIdx.y, idx1 = blockIdx.x * 128 + threadIdx.x;
    const KfDev& k1 = m.cur;
    const KfDev& k2 = m.neigh[j];
    if (idx1 >= k1.n || k2.skip) return;
    const size_t slot = (size_t)j * k1.n + idx1;
    const int idx2 = m.match[slot];
    if (idx2 < 0) return;
    const Kp kp1 = load_kp(k1.keys, idx1), kp2 = load_kp(k2.keys, idx2);
    const float ur1 = k1.u_right[idx1], ur2 = k2.u_right[idx2];
    const bool stereo1 = ur1 >= 0, stereo2 = ur2 >= 0;
    const float* R1 = k1.Rcw; const float* R2 = k2.Rcw;
    const float xn1[3] = {(kp1.x - m.cx) / m.fx, (kp1.y - m.cy) / m.fy, 1.f};
    const float xn2[3] = {(kp2.x - m.cx) / m.fx, (kp2.y - m.cy) / m.fy, 1.f};
    float ray1[3], ray2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        ray1[r] = (R1[r] * xn1[0] + R1[3 + r] * xn1[1]) + R1[6 + r] * xn1[2];
        ray2[r] = (R2[r] * xn2[0] + R2[3 + r] * xn2[1]) + R2[6 + r] * xn2[2];
    }
    const float dot = (ray1[0] * ray2[0] + ray1[1] * ray2[1]) + ray1[2] * ray2[2];
    const float n1 = sqrtf((ray1[0] * ray1[0] + ray1[1] * ray1[1]) + ray1[2] * ray1[2]);
    const float n2 = sqrtf((ray2[0] * ray2[0] + ray2[1] * ray2[1]) + ray2[2] * ray2[2]);
    const float cosRays = dot / (n1 * n2);
    float cosS1 = cosRays + 1, cosS2 = cosRays + 1;
    if (stereo1) cosS1 = (float)cos(2 * atan2((double)(m.mb / 2), (double)k1.depth[idx1]));
    else if (stereo2) cosS2 = (float)cos(2 * atan2((double)(m.mb / 2), (double)k2.depth[idx2]));
    const float cosS = fminf(cosS1, cosS2);
    float x3D[3];
    bool point_stereo = false;
    const float invfx = 1.0f / m.fx, invfy = 1.0f / m.fy;
    if (cosRays < cosS && cosRays > 0 && (stereo1 || stereo2 || (cosRays < 0.9996 && m.inertial) || (cosRays < 0.9998 && !m.inertial))) {
        float A[16];
        const float T1[12] = {R1[0], R1[1], R1[2], k1.t[0], R1[3], R1[4], R1[5], k1.t[1], R1[6], R1[7], R1[8], k1.t[2]};
        const float T2[12] = {R2[0], R2[1], R2[2], k2.t[0], R2[3], R2[4], R2[5], k2.t[1], R2[6], R2[7], R2[8], k2.t[2]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[c] = xn1[0] * T1[8 + c] - T1[c];
            A[4 + c] = xn1[1] * T1[8 + c] - T1[4 + c];
            A[8 + c] = xn2[0] * T2[8 + c] - T2[c];
            A[12 + c] = xn2[1] * T2[8 + c] - T2[4 + c];
        }
        double M[16], v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) s += (double)A[4 * k + r] * (double)A[4 * k + c];
                M[4 * r + c] = s;
            }
        smallest_eigenvector4(M, v);
        const float h3 = (float)v[3];
        if (h3 == 0) return;
        x3D[0] = (float)v[0] / h3; x3D[1] = (float)v[1] / h3; x3D[2] = (float)v[2] / h3;
    } else if (stereo1 && cosS1 < cosS2) {
        point_stereo = true;
        const float z = k1.depth[idx1];
        if (!(z > 0)) return;
        const float xc[3] = {(kp1.x - m.cx) * z * invfx, (kp1.y - m.cy) * z * invfy, z};
#pragma unroll
        for (int r = 0; r < 3; ++r) x3D[r] = ((R1[r] * xc[0] + R1[3 + r] * xc[1]) + R1[6 + r] * xc[2]) + k1.Ow[r];
    } else if (stereo2 && cosS2 < cosS1) {
        point_stereo = true;
        const float z = k2.depth[idx2];
        if (!(z > 0)) return;
        const float xc[3] = {(kp2.x - m.cx) * z * invfx, (kp2.y - m.cy) * z * invfy, z};
#pragma unroll
        for (int r = 0; r < 3; ++r) x3D[r] = ((R2[r] * xc[0] + R2[3 + r] * xc[1]) + R2[6 + r] * xc[2]) + k2.Ow[r];
    } else {
        return;
    }
    if (!reproj_gate(m, R1, k1.t, x3D, kp1, stereo1, ur1, m.level_sigma2[kp1.octave])) return;
    if (!reproj_gate(m, R2, k2.t, x3D, kp2, stereo2, ur2, m.level_sigma2[kp2.octave])) return;
    const float d1[3] = {x3D[0] - k1.Ow[0], x3D[1] - k1.Ow[1], x3D[2] - k1.Ow[2]}, d2[3] = {x3D[0] - k2.Ow[0], x3D[1] - k2.Ow[1], x3D[2] - k2.Ow[2]};
    const float dist1 = sqrtf((d1[0] * d1[0] + d1[1] * d1[1]) + d1[2] * d1[2]), dist2 = sqrtf((d2[0] * d2[0] + d2[1] * d2[1]) + d2[2] * d2[2]);
    if (dist1 == 0 || dist2 == 0) return;
    if (m.far_points && (dist1 >= m.th_far || dist2 >= m.th_far)) return;
    const float ratioDist = dist2 / dist1;
    const float ratioOctave = m.scale_factors[kp1.octave] / m.scale_factors[kp2.octave];
    if (ratioDist * m.ratio_factor < ratioOctave || ratioDist > ratioOctave * m.ratio_factor) return;
    m.ok[slot] = point_stereo ? 3 : 1;
    m.x3D[3 * slot] = x3D[0]; m.x3D[3 * slot + 1] = x3D[1]; m.x3D[3 * slot + 2] = x3D[2];
}

// ORBmatcher::Fuse, the search of one map point (SF/src/ORBmatcher.cc:1185-1300): projection and gates, predicted level, the
// keypoints of the window on the keyframe's feature grid, level / reprojection gates, best descriptor distance
__device__ __forceinline__ void q_rot_dev(const float q[4], const float v[3], float out[3]) {  // Eigen _transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
constexpr int kFuseGridCols = 64, kFuseGridRows = 48;
__global__ __launch_bounds__(128) void k_fuse_search(FuseDev f) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= f.n_points) return;
    int best_idx = -1, best_dist = 256;
    const float* P = reinterpret_cast<const float*>(f.points + 68 * (size_t)i);  // pos 3, normal 3, min, max, max_raw, descriptor
    if (f.valid[i]) {
        const float pos[3] = {P[0], P[1], P[2]};
        float pc[3];
        q_rot_dev(f.q, pos, pc);
        pc[0] += f.t[0]; pc[1] += f.t[1]; pc[2] += f.t[2];
        bool go = !(pc[2] < 0.0f);
        float u = 0, v = 0, ur = 0, dist3D = 0;
        if (go) {
            const float invz = 1 / pc[2];
            u = f.fx * pc[0] / pc[2] + f.cx; v = f.fy * pc[1] / pc[2] + f.cy;
            go = u >= f.min_x && u < f.max_x && v >= f.min_y && v < f.max_y;
            ur = u - f.bf * invz;
        }
        if (go) {
            const float PO[3] = {pos[0] - f.Ow[0], pos[1] - f.Ow[1], pos[2] - f.Ow[2]};
            dist3D = sqrtf((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
            go = !(dist3D < P[6] || dist3D > P[7]);
            const float dotn = (PO[0] * P[3] + PO[1] * P[4]) + PO[2] * P[5];
            go = go && !((double)dotn < 0.5 * (double)dist3D);
        }
        if (go) {
            const float ratio = P[8] / dist3D;
            int level = (int)ceilf((float)log((double)ratio) / f.log_scale_factor);  // logf of the reference, evaluated through double
            level = level < 0 ? 0 : (level >= f.n_levels ? f.n_levels - 1 : level);
            const float r = f.th * f.scale_factors[level];
            const float invW = (float)kFuseGridCols / (f.max_x - f.min_x), invH = (float)kFuseGridRows / (f.max_y - f.min_y);
            const int minCX = max(0, (int)floorf((u - f.min_x - r) * invW)), maxCX = min(kFuseGridCols - 1, (int)ceilf((u - f.min_x + r) * invW));
            const int minCY = max(0, (int)floorf((v - f.min_y - r) * invH)), maxCY = min(kFuseGridRows - 1, (int)ceilf((v - f.min_y + r) * invH));
            if (!(minCX >= kFuseGridCols || maxCX < 0 || minCY >= kFuseGridRows || maxCY < 0)) {
                const uint32_t* qd = reinterpret_cast<const uint32_t*>(P + 9);
                uint32_t q[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) q[w] = qd[w];
                for (int ix = minCX; ix <= maxCX; ++ix)
                    for (int iy = minCY; iy <= maxCY; ++iy) {
                        const int c = ix * kFuseGridRows + iy;
                        for (int k = f.cell_start[c]; k < f.cell_start[c + 1]; ++k) {
                            const int idx = f.items[k];
                            const Kp kp = load_kp(f.keys, idx);
                            if (!(fabsf(kp.x - u) < r && fabsf(kp.y - v) < r)) continue;
                            if (kp.octave < level - 1 || kp.octave > level) continue;
                            const float kur = f.u_right[idx];
                            const float ex = u - kp.x, ey = v - kp.y;
                            if (kur >= 0) {
                                const float er = ur - kur;
                                const float e2 = ex * ex + ey * ey + er * er;
                                if ((double)(e2 * f.inv_level_sigma2[kp.octave]) > 7.8) continue;
                            } else {
                                const float e2 = ex * ex + ey * ey;
                                if ((double)(e2 * f.inv_level_sigma2[kp.octave]) > 5.99) continue;
                            }
                            const uint32_t* kd = reinterpret_cast<const uint32_t*>(f.desc) + 8 * (size_t)idx;
                            int dist = 0;
#pragma unroll
                            for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ kd[w]);
                            if (dist < best_dist) { best_dist = dist; best_idx = idx; }
                        }
                    }
            }
        }
    }
    f.best_dist[i] = best_dist;
    f.best_idx[i] = best_dist <= 50 ? best_idx : -1;  // TH_LOW
}

void launch_fuse_search(const FuseDev& f, hipStream_t st) {
    if (f.n_points > 0) TC2LI_LAUNCH(k_fuse_search, dim3((f.n_points + 127) / 128), dim3(128), 0, st, f);
}

void launch_tri_search(const MappingDev& m, int max_entries, hipStream_t st) {
    if (m.n_neigh > 0 && max_entries > 0) TC2LI_LAUNCH(k_tri_search, dim3((max_entries + 255) / 256, m.n_neigh), dim3(256), 0, st, m);
}
void launch_tri_points(const MappingDev& m, hipStream_t st) {
    if (m.n_neigh > 0 && m.cur.n > 0) TC2LI_LAUNCH(k_tri_points, dim3((m.cur.n + 127) / 128, m.n_neigh), dim3(128), 0, st, m);
}

}  // namespace tc2li
