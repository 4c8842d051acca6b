// Stereo matching of Frame::ComputeStereoMatches (SF/src/Frame.cc:841-1011) on gfx950: one wavefront per left
// keypoint.  The wave scans all right keypoints of the frame (row band / octave / disparity gates from compact key
// records staged in LDS), takes the best Hamming distance with a wave-wide min on (distance, index), then slides
// the 11x11 SAD window over +-5 px on the keypoint's pyramid level and fits the parabola.  Integer work is exact;
// the float tail uses explicitly rounded operations so that uRight/depth equal the CPU results bit for bit.
#include <hip/hip_runtime.h>

#include "launch.hpp"
// Bit-exactness with the CPU path needs every float operation rounded on its own: no FMA contraction (the HIP
// `__fmul_rn`-style intrinsics are plain operators unless OCML_BASIC_ROUNDED_OPERATIONS is defined, and `__fsqrt_rn` is
// the approximate native square root -- use sqrtf(), which hipcc rounds correctly by default).
#pragma clang fp contract(off)
#include <stdint.h>

#include <algorithm>
#include <cmath>

#include "det_math.hpp"
#include "orb_device.hpp"

namespace tc2li {

constexpr int kStereoMaxRight = 4096;  // right keypoints per frame that fit the LDS records
constexpr int kLeftPerBlock = 32;
constexpr int kStereoMaxRows = 2048;  // image rows the row lists of k_stereo_rows cover

struct RightRec {
    float x;
    int16_t minr, maxr;
    int32_t octave;
};

// vRowIndices of Frame::ComputeStereoMatches (SF/src/Frame.cc:858-866): for every image row the right keypoints whose band
// [floor(y - r), ceil(y + r)], r = 2 * scale[octave], covers it -- a counting sort per frame (one workgroup), lists in global memory.
// A left keypoint then compares descriptors with the candidates of its row only (some tens) instead of gating all right keypoints.
// The order inside a list is arbitrary: the match is the minimum over (distance, right index), which does not depend on it.
__global__ __launch_bounds__(256) void k_stereo_rows(ScaleTable sc, const StereoFrame* __restrict__ frames, const MatchKey* __restrict__ keys, int rows,
                                                     int entry_cap, int32_t* __restrict__ row_start, uint16_t* __restrict__ entries) {
    __shared__ int s_cnt[kStereoMaxRows];
    __shared__ int s_wave[4];
    const StereoFrame fr = frames[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const MatchKey* kr = keys + fr.right_off;
    int32_t* rs = row_start + (size_t)blockIdx.x * (rows + 1);
    uint16_t* out = entries + (size_t)blockIdx.x * entry_cap;
    for (int y = tid; y < rows; y += 256) s_cnt[y] = 0;
    __syncthreads();
    for (int i = tid; i < fr.n_right; i += 256) {
        const MatchKey k = kr[i];
        const float r = __fmul_rn(2.0f, sc.scale[k.octave]);
        const int maxr = min(rows - 1, (int)ceilf(__fadd_rn(k.y, r))), minr = max(0, (int)floorf(__fsub_rn(k.y, r)));
        for (int y = minr; y <= maxr; ++y) atomicAdd(&s_cnt[y], 1);
    }
    __syncthreads();
    // exclusive prefix over the rows: 8 consecutive rows per lane and pass
    int carry = 0;
    for (int y0 = 0; y0 < rows; y0 += 2048) {
        int c[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int y = y0 + tid * 8 + k; c[k] = y < rows ? s_cnt[y] : 0; sum += c[k]; }
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int base = carry, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { base += k < wave ? s_wave[k] : 0; tot += s_wave[k]; }
        int at = base + incl - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int y = y0 + tid * 8 + k; if (y < rows) { rs[y] = at; s_cnt[y] = at; } at += c[k]; }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) rs[rows] = carry;
    for (int i = tid; i < fr.n_right; i += 256) {
        const MatchKey k = kr[i];
        const float r = __fmul_rn(2.0f, sc.scale[k.octave]);
        const int maxr = min(rows - 1, (int)ceilf(__fadd_rn(k.y, r))), minr = max(0, (int)floorf(__fsub_rn(k.y, r)));
        for (int y = minr; y <= maxr; ++y) {
            const int pos = atomicAdd(&s_cnt[y], 1);
            if (pos < entry_cap) out[pos] = (uint16_t)i;
        }
    }
}

// kRows: candidates from the row lists of k_stereo_rows; otherwise (images taller than kStereoMaxRows) every right keypoint is gated.
template <bool kRows>
__global__ __launch_bounds__(256) void k_stereo_match(LevelTable left, LevelTable right, ScaleTable sc,
                                                      const StereoFrame* __restrict__ frames, const MatchKey* __restrict__ keys,
                                                      const uint8_t* __restrict__ desc, float mbf, float max_d,
                                                      float* __restrict__ u_right, float* __restrict__ depth,
                                                      int* __restrict__ best_sad, int gx, int nframes, int rows, int entry_cap,
                                                      const int32_t* __restrict__ row_start, const uint16_t* __restrict__ entries) {
    __shared__ RightRec recs[kRows ? 1 : kStereoMaxRight];
    // One-dimensional launch in XCD-contiguous order (workgroups reach the 8 XCDs round-robin by linear index, each XCD has its own
    // L2): XCD k takes the k-th eighth of the (frame, key block) list, so a frame's pyramid rows and right-image records are
    // fetched into one L2.
    const int total = gx * nframes, per_xcd = (total + 7) / 8;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= total) return;
    const int frame = logical / gx, bx = logical - frame * gx;
    const StereoFrame fr = frames[frame];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_block();
    const int first = bx * kLeftPerBlock;
    if (first >= fr.n_left) return;
    const MatchKey* kr = keys + fr.right_off;
    if (!kRows) {
        for (int i = tid; i < fr.n_right; i += 256) {
            const MatchKey k = kr[i];
            const float r = __fmul_rn(2.0f, sc.scale[k.octave]);
            RightRec rc;
            rc.x = k.x;
            rc.maxr = (int16_t)(int)ceilf(__fadd_rn(k.y, r));
            rc.minr = (int16_t)(int)floorf(__fsub_rn(k.y, r));
            rc.octave = k.octave;
            recs[i] = rc;
        }
        __syncthreads();
    }
    const int32_t* rs = kRows ? row_start + (size_t)frame * (rows + 1) : nullptr;
    const uint16_t* ent = kRows ? entries + (size_t)frame * entry_cap : nullptr;

    const uint32_t* dR = reinterpret_cast<const uint32_t*>(desc + (size_t)fr.right_off * 32);
    for (int li = first + wave; li < min(first + kLeftPerBlock, fr.n_left); li += 4) {
        const MatchKey kl = keys[fr.left_off + li];
        const uint32_t* dLp = reinterpret_cast<const uint32_t*>(desc + (size_t)(fr.left_off + li) * 32);
        uint32_t dl[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dl[k] = dLp[k];
        const int row = (int)kl.y;
        const float minU = __fsub_rn(kl.x, max_d), maxU = kl.x;
        uint32_t best = (100u << 16) | 0xffffu;  // TH_HIGH; ties resolve to the smallest right index
        if (kRows) {
            if (row >= 0 && row < rows) {
                const int c1 = rs[row + 1];
                for (int c = rs[row] + lane; c < c1; c += 64) {
                    const int i = ent[c];
                    const MatchKey k = kr[i];
                    if (k.octave >= kl.octave - 1 && k.octave <= kl.octave + 1 && k.x >= minU && k.x <= maxU) {
                        const uint32_t* p = dR + (size_t)i * 8;
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) dist += __popc(dl[w] ^ p[w]);
                        best = min(best, ((uint32_t)dist << 16) | (uint32_t)i);
                    }
                }
            }
        } else {
            for (int i = lane; i < fr.n_right; i += 64) {
                const RightRec rc = recs[i];
                const bool ok = row >= rc.minr && row <= rc.maxr && rc.octave >= kl.octave - 1 && rc.octave <= kl.octave + 1 &&
                                rc.x >= minU && rc.x <= maxU;
                if (ok) {
                    const uint32_t* p = dR + (size_t)i * 8;
                    int dist = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) dist += __popc(dl[k] ^ p[k]);
                    const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)i;
                    best = min(best, key);
                }
            }
        }
        best = wave_min_u32(best);
        const int bestDist = (int)(best >> 16), bestIdx = (int)(best & 0xffff);
        float out_u = -1.0f, out_d = -1.0f;
        int out_sad = -1;
        if (bestDist < 75 && !(maxU < 0)) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2
            const float uR0 = kRows ? kr[bestIdx].x : recs[bestIdx].x;
            const float sf = sc.inv_scale[kl.octave];
            const float scaleduL = roundf(__fmul_rn(kl.x, sf));
            const float scaledvL = roundf(__fmul_rn(kl.y, sf));
            const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
            const LevelDesc PL = left.lv[kl.octave], PR = right.lv[kl.octave];
            const float iniu = scaleduR0, endu = __fadd_rn(scaleduR0, 11.0f);  // scaleduR0+L-w, scaleduR0+L+w+1
            if (!(iniu < 0 || endu >= (float)PR.w)) {
                const uint8_t* il = PL.img + (size_t)fr.left_img * PL.img_stride + (size_t)((int)scaledvL - 5) * PL.pitch +
                                    ((int)scaleduL - 5);
                const uint8_t* ir = PR.img + (size_t)fr.right_img * PR.img_stride + (size_t)((int)scaledvL - 5) * PR.pitch +
                                    ((int)scaleduR0 - 5);
                // lane owns window pixels `lane` and `lane + 64` (121 in total)
                const int p0 = lane, p1 = lane + 64;
                const int y0 = p0 / 11, x0 = p0 - y0 * 11, y1 = p1 / 11, x1 = p1 - y1 * 11;
                const bool has1 = p1 < 121;
                // all 24 bytes of the lane are requested before the first is used (a lane without a second pixel reads its first one
                // again: an address select, so no load sits behind a branch -- with the load next to its use every one of the eleven
                // window positions cost the wavefront a trip to memory)
                const uint8_t* pl0 = il + y0 * PL.pitch + x0;
                const uint8_t* pl1 = has1 ? il + y1 * PL.pitch + x1 : pl0;
                const uint8_t* pr0 = ir + y0 * PR.pitch + x0;
                const uint8_t* pr1 = has1 ? ir + y1 * PR.pitch + x1 : pr0;
                const int a0 = *pl0, a1 = *pl1;
                int rv0[11], rv1[11];
#pragma unroll
                for (int k = 0; k < 11; ++k) { rv0[k] = pr0[k - 5]; rv1[k] = pr1[k - 5]; }
                int bestS = 0x7fffffff, bestInc = 0, sads[11];
#pragma unroll
                for (int inc = -5; inc <= 5; ++inc) {
                    int s = abs(a0 - rv0[inc + 5]);
                    if (has1) s += abs(a1 - rv1[inc + 5]);
                    s = wave_sum_i32(s);   // (a scalar: the comparisons below are scalar arithmetic)
                    sads[inc + 5] = s;
                    if (s < bestS) { bestS = s; bestInc = inc; }
                }
                if (bestInc != -5 && bestInc != 5) {
                    float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                    for (int k = 1; k < 10; ++k)
                        if (k == bestInc + 5) { d1 = (float)sads[k - 1]; d2 = (float)sads[k]; d3 = (float)sads[k + 1]; }
                    const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
                    const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), den);
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = __fmul_rn(sc.scale[kl.octave], __fadd_rn(__fadd_rn(scaleduR0, (float)bestInc), deltaR));
                        float disparity = __fsub_rn(kl.x, bestuR);
                        if (disparity >= 0 && disparity < max_d) {
                            if (disparity <= 0) {
                                disparity = (float)0.01;
                                bestuR = (float)__dsub_rn((double)kl.x, 0.01);
                            }
                            out_d = __fdiv_rn(mbf, disparity);
                            out_u = bestuR;
                            out_sad = bestS;
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            u_right[fr.out_off + li] = out_u;
            depth[fr.out_off + li] = out_d;
            best_sad[fr.out_off + li] = out_sad;
        }
    }
}

int stereo_row_entry_cap(const ScaleTable& sc, int n_levels, int max_right) {
    float smax = 1.f;
    for (int l = 0; l < n_levels; ++l) smax = std::max(smax, sc.scale[l]);
    return max_right * (2 * (int)ceilf(2.f * smax) + 3);
}

void launch_stereo_match(const LevelTable& left, const LevelTable& right, const ScaleTable& sc, const StereoFrame* frames,
                         int nframes, int max_left, const MatchKey* keys, const uint8_t* desc, float mbf, float max_d,
                         float* u_right, float* depth, int* best_sad, int rows, int entry_cap, int32_t* row_start, uint16_t* entries, hipStream_t st) {
    if (nframes <= 0 || max_left <= 0) return;
    const int gx = (max_left + kLeftPerBlock - 1) / kLeftPerBlock;
    const dim3 grid((gx * nframes + 7) / 8 * 8);
    if (row_start && entries && rows > 0 && rows <= kStereoMaxRows) {
        TC2LI_LAUNCH(k_stereo_rows, dim3(nframes), dim3(256), 0, st, sc, frames, keys, rows, entry_cap, row_start, entries);
        TC2LI_LAUNCH(k_stereo_match<true>, grid, dim3(256), 0, st, left, right, sc, frames, keys, desc, mbf, max_d, u_right, depth, best_sad, gx, nframes, rows,
                     entry_cap, row_start, entries);
    } else {
        TC2LI_LAUNCH(k_stereo_match<false>, grid, dim3(256), 0, st, left, right, sc, frames, keys, desc, mbf, max_d, u_right, depth, best_sad, gx, nframes, 0, 0,
                     nullptr, nullptr);
    }
}

}  // namespace tc2li
