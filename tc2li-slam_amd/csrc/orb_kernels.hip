// gfx950 kernels of the ORB extractor (replaces the internals of SF/src/ORBextractor.cc).
// All arithmetic is integer or explicitly rounded float, so the output is bit-identical to the CPU path.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <type_traits>

#include "launch.hpp"
// Bit-exactness with the CPU path needs every float operation rounded on its own: no FMA contraction (the HIP
// `__fmul_rn`-style intrinsics are plain operators unless OCML_BASIC_ROUNDED_OPERATIONS is defined, and `__fsqrt_rn` is
// the approximate native square root -- use sqrtf(), which hipcc rounds correctly by default).
#pragma clang fp contract(off)
#include <stdint.h>

#include "det_math.hpp"
#include "orb_device.hpp"

namespace tc2li {

// Pointers that arrive inside by-value tables (LevelTable) are generic to the compiler: loads through them become flat_load.
// They are device-global by construction; saying so gives global_load (no LDS aperture check, vmcnt only).
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
    return (const __attribute__((address_space(1))) T*)p;
}

// ------------------------------------------------------------------------------------------------------
// Pyramid level from the previous one: cv::resize(INTER_LINEAR) on 8-bit (SF/src/ORBextractor.cc:1156).
// Tables (xofs/ialpha/yofs/ibeta) are built on the host with the library's float/double recipe.  One thread
// produces 4 horizontally adjacent destination pixels and stores them as one dword.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t* __restrict__ src, int src_pitch, size_t src_img_stride,
                                                       int sw, int sh, uint8_t* __restrict__ dst, int dst_pitch,
                                                       size_t dst_img_stride, int dw, int dh,
                                                       const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                       const int* __restrict__ yofs, const short* __restrict__ ibeta, int gx, int gy, int nimg) {
    // One-dimensional launch, XCD-aware: workgroups go to the 8 XCDs round-robin by their linear index and each XCD has its own L2;
    // blocks that are neighbours in (x, y) read the same source rows, so XCD k takes the k-th contiguous eighth of the
    // (image, row block, column block) list instead of every 8th block.
    const int per_xcd = (gx * gy * nimg + 7) / 8;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (logical >= gx * gy * nimg) return;
    const int img = logical / (gx * gy), rem = logical - img * (gx * gy), by = rem / gx, bx = rem - by * gx;
    const int dy = by * 4 + wave_in_block();
    const int dx0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= dh || dx0 >= dw) return;
    const uint8_t* S = src + (size_t)img * src_img_stride;
    int sy0 = yofs[dy], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);
    sy1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
    const uint8_t* R0 = S + (size_t)sy0 * src_pitch;
    const uint8_t* R1 = S + (size_t)sy1 * src_pitch;
    const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint32_t packed = 0;
    uint8_t* D = dst + (size_t)img * dst_img_stride + (size_t)dy * dst_pitch + dx0;
    const int sx_first = xofs[dx0];
    if (dx0 + 3 < dw && sx_first >= 0 && sx_first + 15 <= sw && xofs[dx0 + 3] - sx_first <= 7) {
        // Interior: the four outputs read source columns sx_first .. sx_first + 8 at most (scale factor < 2): three aligned
        // dwords per row and one 16-byte load per table instead of sixteen byte loads and twelve table loads.
        const int4 xo = *reinterpret_cast<const int4*>(xofs + dx0);          // dx0 is a multiple of 4
        const uint4 al = *reinterpret_cast<const uint4*>(ialpha + 2 * dx0);  // eight shorts
        const uintptr_t p0 = (uintptr_t)(R0 + sx_first), p1 = (uintptr_t)(R1 + sx_first);
        const auto* q0 = as_global(reinterpret_cast<const uint32_t*>(p0 & ~(uintptr_t)3));
        const auto* q1 = as_global(reinterpret_cast<const uint32_t*>(p1 & ~(uintptr_t)3));
        const uint32_t u0 = q0[0], u1 = q0[1], u2 = q0[2], v0 = q1[0], v1 = q1[1], v2 = q1[2];
        const int sh0 = (int)(p0 & 3), sh1 = (int)(p1 & 3);
        auto byte_at = [](uint32_t w0, uint32_t w1, uint32_t w2, int j) -> int {
            const uint32_t w = j < 4 ? w0 : (j < 8 ? w1 : w2);
            return (int)((w >> (8 * (j & 3))) & 0xffu);
        };
        const int sxs[4] = {xo.x, xo.y, xo.z, xo.w};
        const uint32_t als[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = sxs[k] - sx_first;
            const int a0 = (short)(als[k] & 0xffffu), a1 = (short)(als[k] >> 16);
            const int r0 = byte_at(u0, u1, u2, sh0 + o) * a0 + byte_at(u0, u1, u2, sh0 + o + 1) * a1;
            const int r1 = byte_at(v0, v1, v2, sh1 + o) * a0 + byte_at(v0, v1, v2, sh1 + o + 1) * a1;
            const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xff) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(D) = packed;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = dx0 + k;
        if (dx < dw) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;  // weight of the clamped tap is 0
            const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            const int r0 = R0[sx] * a0 + R0[sx1] * a1;
            const int r1 = R1[sx] * a0 + R1[sx1] * a1;
            const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xff) << (8 * k);
        }
    }
    if (dx0 + 3 < dw) {
        *reinterpret_cast<uint32_t*>(D) = packed;  // dst_pitch and dx0 are multiples of 4
    } else {
        for (int k = 0; k < 4 && dx0 + k < dw; ++k) D[k] = (uint8_t)(packed >> (8 * k));
    }
}

// The same resize as column strips (round 3, default): the kernel above is bound by instruction issue (a wavefront instruction takes
// two cycles of its SIMD-32; 1.5e8 of them per launch are about half of the kernel's time when it runs alone, and its time follows
// its instruction count): ~26 lane-operations per destination pixel for the byte selection out of aligned dwords and the fixed-point
// blend, twice per pixel because every destination row repeats the horizontal pass of both its source rows.  Here a lane keeps four
// destination columns and walks kResizeRows destination rows down: the column tables (source offsets, alpha pairs) are loaded once, the
// two source bytes of a tap pair come from ONE unaligned 16-bit load (no selection arithmetic), and the horizontal pass of a source row
// that two consecutive destination rows share (scale < 2: every other row at 1.2) is computed once and kept in registers.  ~13
// lane-operations per pixel: 363 -> 196 us per launch of 1024 images.  Same integers as the kernel above: tests/test_orb_gpu.py
// compares every level with the oracle.
constexpr int kResizeRows = 16;
__device__ __forceinline__ uint32_t load_u16_unaligned(const uint8_t* p) {
    uint16_t w;
    __builtin_memcpy(&w, p, 2);
    return w;
}
// one wavefront's unit of the strip form: destination columns [dx0, dx0 + 4) of this lane, rows [dy_begin, dy_end) of one image
// (S: the source image, D: the destination image + dx0)
__device__ __forceinline__ void resize_strip_unit(const uint8_t* __restrict__ S, int src_pitch, int sw, int sh, uint8_t* __restrict__ D, int dst_pitch, int dw,
                                                  const int* __restrict__ xofs, const short* __restrict__ ialpha, const int* __restrict__ yofs,
                                                  const short* __restrict__ ibeta, int dx0, int dy_begin, int dy_end) {
    int sx[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = min(dx0 + k, dw - 1);
        sx[k] = xofs[dx]; a0[k] = ialpha[2 * dx]; a1[k] = ialpha[2 * dx + 1];
    }
    const bool interior = dx0 + 3 < dw && sx[0] >= 0 && sx[3] + 1 < sw;  // every tap pair inside the row: two bytes from one load
    // horizontal pass of source row `sy` for the lane's four columns, already shifted (r >> 4)
    auto hpass = [&](int sy, int h[4]) {
        const uint8_t* R = S + (size_t)sy * src_pitch;
        if (interior) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = load_u16_unaligned(R + sx[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = ((int)(w[k] & 0xffu) * a0[k] + (int)(w[k] >> 8) * a1[k]) >> 4;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x0 = sx[k], x1 = x0 + 1 < sw ? x0 + 1 : x0;  // weight of the clamped tap is 0
                h[k] = dx0 + k < dw ? ((int)R[x0] * a0[k] + (int)R[x1] * a1[k]) >> 4 : 0;
            }
        }
    };
    int have = -1, hc[4] = {0, 0, 0, 0};
    for (int dy = dy_begin; dy < dy_end; ++dy) {
        int sy0 = yofs[dy], sy1 = sy0 + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);
        sy1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        int h0[4], h1[4];
        if (sy0 == have) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h0[k] = hc[k];
        } else {
            hpass(sy0, h0);
        }
        if (sy1 == sy0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h1[k] = h0[k];
        } else {
            hpass(sy1, h1);
        }
        have = sy1;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hc[k] = h1[k];
            const int v = (((b0 * h0[k]) >> 16) + ((b1 * h1[k]) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xff) << (8 * k);
        }
        uint8_t* Drow = D + (size_t)dy * dst_pitch;
        if (dx0 + 3 < dw) {
            *reinterpret_cast<uint32_t*>(Drow) = packed;  // dst_pitch and dx0 are multiples of 4
        } else {
            for (int k = 0; k < 4 && dx0 + k < dw; ++k) Drow[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}


__global__ __launch_bounds__(256) void k_resize_strips(const uint8_t* __restrict__ src, int src_pitch, size_t src_img_stride, int sw, int sh,
                                                       uint8_t* __restrict__ dst, int dst_pitch, size_t dst_img_stride, int dw, int dh,
                                                       const int* __restrict__ xofs, const short* __restrict__ ialpha, const int* __restrict__ yofs,
                                                       const short* __restrict__ ibeta, int gx, int gy, int nimg, int rows_per_wave) {
    // one wavefront = 256 columns x rows_per_wave rows; XCD k takes the k-th contiguous eighth of the (image, row strip, column block) list
    const int n_units = gx * gy * nimg, per_xcd = (n_units + 7) / 8;
    const int u = ((int)blockIdx.x >> 3) * 4 + wave_in_block();  // the wavefront's unit inside its XCD's share
    const int logical = ((int)blockIdx.x & 7) * per_xcd + u;
    if (u >= per_xcd || logical >= n_units) return;
    const int img = logical / (gx * gy), rem = logical - img * (gx * gy), by = rem / gx, bx = rem - by * gx;
    const int dx0 = (bx * 64 + (int)(threadIdx.x & 63)) * 4;
    if (dx0 >= dw) return;
    const uint8_t* S = src + (size_t)img * src_img_stride;
    uint8_t* D = dst + (size_t)img * dst_img_stride + dx0;
    const int dy_begin = by * rows_per_wave, dy_end = min(dy_begin + rows_per_wave, dh);
    resize_strip_unit(S, src_pitch, sw, sh, D, dst_pitch, dw, xofs, ialpha, yofs, ibeta, dx0, dy_begin, dy_end);
}

// The small levels of the pyramid in ONE launch: level l is made from level l - 1 (SF/src/ORBextractor.cc:1159-1186), so the seven resizes of
// an extraction are a chain of dependent launches, and beside the other stages' kernels every link costs about 0.6 ms whatever its size.
// From level `l0` on a level is small enough for one workgroup per image to make it whole: the workgroup walks levels l0 .. n_levels - 1,
// its wavefronts take the strip units of a level in turn, a barrier between two levels (the level just written is read back by the same
// workgroup: through L2, no other workgroup touches this image).  Same arithmetic as k_resize_strips, unit by unit.
struct ResizeTables { const int* xofs[kMaxLevels]; const short* ialpha[kMaxLevels]; const int* yofs[kMaxLevels]; const short* ibeta[kMaxLevels]; };
constexpr int kResizeTailThreads = 1024;
__global__ __launch_bounds__(kResizeTailThreads) void k_resize_tail(LevelTable lv, ResizeTables tb, int l0, int n_levels) {
    const int img = blockIdx.x, wave = wave_in_block(), lane = threadIdx.x & 63;
    for (int l = l0; l < n_levels; ++l) {
        const LevelDesc S = lv.lv[l - 1], D = lv.lv[l];
        const uint8_t* src = S.img + (size_t)img * S.img_stride;
        uint8_t* dst = const_cast<uint8_t*>(D.img) + (size_t)img * D.img_stride;
        const int gx = (D.w + 255) / 256, gy = (D.h + kResizeRows - 1) / kResizeRows;
        for (int u = wave; u < gx * gy; u += kResizeTailThreads / 64) {
            const int by = u / gx, bx = u - by * gx;
            const int dx0 = (bx * 64 + lane) * 4;
            if (dx0 >= D.w) continue;
            resize_strip_unit(src, S.pitch, S.w, S.h, dst + dx0, D.pitch, D.w, tb.xofs[l], tb.ialpha[l], tb.yofs[l], tb.ibeta[l], dx0, by * kResizeRows,
                              min((by + 1) * kResizeRows, D.h));
        }
        __syncthreads();  // level l is complete (and visible to this workgroup) before level l + 1 reads it
    }
}

// ------------------------------------------------------------------------------------------------------
// FAST-9/16 per 35-px cell with the reference's threshold fallback (SF/src/ORBextractor.cc:776-846):
// one workgroup per (cell, image).  The cell window (cell + 6 px) is staged in LDS; S = the largest arc
// contrast is computed once (a pixel is a corner at threshold t iff S > t, its cv::FAST score is S-1), then
// 3x3 strict non-max suppression is evaluated for iniTh and, if that leaves the cell empty, for minTh.
// Survivors go into an LDS list through a counter atomic and are then ranked by counting the survivors that precede them in
// row-major order (the order cv::FAST returns them in), so the emitted list is ordered without a ballot pass (a ballot
// compaction was tried: 1.05 ms against 0.64 ms per 128 images).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool has_arc9(uint32_t m16) {
    uint32_t x = m16 | (m16 << 16);
    uint32_t r = x & (x >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= x >> 8;
    return (r & 0xffffu) != 0;
}

// i / n for 0 <= i < 2^16, 1 <= n < 256: the float quotient is off by less than one unit, one correction step makes it exact
__device__ __forceinline__ int row_of(int i, int n, float inv_n) {
    int q = (int)((float)i * inv_n);
    q -= q * n > i ? 1 : 0;
    q += (q + 1) * n <= i ? 1 : 0;
    return q;
}

template <bool DARK>
__device__ __forceinline__ int arc_contrast(const int (&p)[16], int v) {
    // max over the 16 arcs of 9 contiguous circle pixels of min(v - p) (DARK) or min(p - v)
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = DARK ? v - p[k] : p[k] - v;
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m8[k] = min(m4[k], m4[(k + 4) & 15]);
    int best = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) best = max(best, min(m8[k], d[(k + 8) & 15]));
    return best;
}

// TH x TW: the largest cell window the instantiation holds (LDS is sized by it: the small variant fits 8 workgroups per CU)
// Threads per cell.  A cell's six phases are barrier-separated steps of about a microsecond, and the kernel issues VALU instructions 18 % of
// the time (profiles/r02_pmc_instruction_mix.json): what counts is how many cells a CU has in flight.  128 threads per cell keep 15
// workgroups resident (10.3 KB of LDS each) where 256 allow 8: 0.53 -> 0.43 ms per 128 images; 64: 0.52.  (Round 1 measured the
// opposite, 1.13 against 0.89 ms, on the kernel that tested every pixel at the low threshold.)  Scoring inside the segment-test pass
// (one barrier-separated step less, worse lane balance) was tried: 0.45 ms.
constexpr int kFastThreads = 128;
template <int TH, int TW>
__global__ __launch_bounds__(kFastThreads) void k_fast_cells(LevelTable levels, const FastCell* __restrict__ cells, int ini_th,
                                                    int min_th, uint32_t* __restrict__ slab, size_t slab_img_stride,
                                                    int* __restrict__ cell_counts, int ncells, int nimg,
                                                    const int* __restrict__ cell_ids, int n_ids, int cells_per_wg) {
    // The window rows are staged as dwords read at the rows' own byte addresses (rows of the caller's image start anywhere): pixel
    // (x, y) of the window is byte y * kTileP + x of the tile.  (Round 1 / 2 staged the ALIGNED dwords and carried a per-row
    // misalignment term through every LDS address; the kernel's time follows its instruction count -- 1.9 G wavefront VALU
    // instructions per 1024 images at two cycles each are half of its 3.2 ms alone -- and that term was a sixth of them: 3.22 -> 2.66 ms.)
    constexpr int kTileP = TW + 4;
    __shared__ uint32_t tile32[TH * kTileP / 4];
    __shared__ uint8_t score[TH * TW];
    __shared__ uint16_t s_list[TH * TW];  // window offset | polarity << 14 of the pixels passing the segment test; after the NMS: offset | kept << 14
    constexpr int kMaxKept = ((TW - 6 + 1) / 2) * ((TH - 6 + 1) / 2);  // >= any cell's slab_cap
    __shared__ uint16_t s_kept[kMaxKept];
    __shared__ int s_cnt_ini, s_nlist, s_nkept;
    const uint8_t* tile = reinterpret_cast<const uint8_t*>(tile32);

    // Workgroups go to the 8 XCDs round-robin by their linear index, and each XCD has its own L2.  Neighbouring cell windows
    // share cache lines (a 42-byte window row is a third of a line, windows overlap by 6 px), so XCD k takes the k-th contiguous
    // eighth of the (image, cell) list instead of every 8th cell: without this every line was fetched by ~4 XCDs (389 MB of HBM
    // reads per launch for 96 MB of pixels in the first PMC profile).
    // cell_ids: the cells this launch covers (the windows that fit the instantiation's tile), n_ids of the ncells of an image
    // A workgroup takes cells_per_wg CONSECUTIVE cells of its XCD's share one after the other (batches: a cell is ~5 us of work, and beside the
    // other stages' kernels the dispatch of 2 x 10^5 workgroups per launch was part of what the launch cost)
    const int total = n_ids * nimg, per_xcd = (total + 7) / 8;
    const int tid = threadIdx.x;
    for (int rep = 0; rep < cells_per_wg; ++rep) {
    const int in_xcd = ((int)blockIdx.x >> 3) * cells_per_wg + rep;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + in_xcd;
    if (in_xcd >= per_xcd || logical >= total) break;  // (the same for every lane)
    if (rep) __syncthreads();  // the cell before has been emitted: its LDS arrays are free
    const int img = logical / n_ids, cell = cell_ids[logical - img * n_ids];
    const FastCell c = cells[cell];
    const LevelDesc L = levels.lv[c.level];
    const uint8_t* src = L.img + (size_t)img * L.img_stride + (size_t)c.y0 * L.pitch + c.x0;
    const int w = c.w, h = c.h;
    {
        constexpr int kDw = kTileP / 4;  // dwords per tile row
        constexpr int kRounds = (TH * kDw + kFastThreads - 1) / kFastThreads;
        // every load of the workgroup is requested before the first one is waited for (a dependent loop would pay the global
        // latency once per round: the staging was a quarter of the kernel)
        uint32_t v[kRounds];
        bool ok[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int i = tid + kFastThreads * r;
            const int y = i / kDw, j = i - y * kDw;
            ok[r] = i < h * kDw && 4 * j < w;
            v[r] = 0;
            if (ok[r]) { uint32_t t; __builtin_memcpy(&t, src + (size_t)y * L.pitch + 4 * j, 4); v[r] = t; }
        }
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            if (ok[r]) tile32[tid + kFastThreads * r] = v[r];
        uint32_t* score32 = reinterpret_cast<uint32_t*>(score);
        for (int i = tid; i < h * (TW / 4); i += kFastThreads) score32[i] = 0;
    }
    if (tid == 0) { s_cnt_ini = 0; s_nlist = 0; s_nkept = 0; }
    __syncthreads();

    const int ew = w - 6, eh = h - 6;
    // the 16 circle pixels and the centre of window position (cx, cy)
    auto circle = [&](int cx, int cy, int (&p)[16]) -> int {
        const uint8_t* r3 = tile + cy * kTileP + cx;
        const uint8_t *r0 = r3 - 3 * kTileP, *r1 = r3 - 2 * kTileP, *r2 = r3 - kTileP, *r4 = r3 + kTileP, *r5 = r3 + 2 * kTileP, *r6 = r3 + 3 * kTileP;
        p[0] = r6[0];  p[1] = r6[1];  p[2] = r5[2];  p[3] = r4[3];  p[4] = r3[3];  p[5] = r2[3];  p[6] = r1[2];  p[7] = r0[1];
        p[8] = r0[0];  p[9] = r0[-1]; p[10] = r1[-2]; p[11] = r2[-3]; p[12] = r3[-3]; p[13] = r4[-3]; p[14] = r5[-2]; p[15] = r6[-1];
        return r3[0];
    };
    // The reference runs cv::FAST(iniThFAST) on the cell and only if that finds nothing cv::FAST(minThFAST) (SF/src/ORBextractor.cc:799-812).
    // Same here: the passes below run at iniThFAST first -- far fewer pixels get past the compass pre-test at 20 grey levels than at 7 -- and
    // once more at minThFAST only for a cell without any corner (low-contrast regions, where few pixels survive either way).  The score of a
    // pixel does not depend on the threshold; non-maximum suppression only sees neighbours that are corners at the threshold of the attempt
    // (cv::FAST's score buffer): in the second attempt the survivors of the first are a subset, their scores are simply written again.
    int nlist = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
    const int th = attempt == 0 ? ini_th : min_th;
    if (attempt == 1) {
        __syncthreads();
        if (tid == 0) { s_nlist = 0; s_cnt_ini = 0; }
        __syncthreads();
    }
    // Pass 0, every pixel: a necessary condition of the segment test on the four compass pixels -- nine contiguous circle pixels
    // always contain pixel 0 or 8 and pixel 4 or 12, so a brighter (darker) arc needs (p0 | p8) & (p4 | p12) brighter (darker).
    // Most pixels of an image stop here; the others are appended to a list (any order).
    // FOUR pixels per lane (round 6; the pass was a byte load per compass pixel, an address and a loop step per pixel: 22 vector and 5 LDS
    // instructions each, and the kernel -- like the whole extraction -- is bound by instruction issue): the lane reads the ALIGNED dwords
    // around its four centres (three of the centre row, two of the rows three above and below: four LDS instructions), cuts the five
    // 4-pixel words out of them with v_alignbyte -- window pixel cx = 3 + 4 g + k sits in byte 3 of dword g or bytes 0..2 of dword
    // g + 1, the pixel three to its left in byte k of dword g, the one three to its right in dwords g + 1 / g + 2 -- and tests pixel k on
    // byte k of those words (the byte selects are operand modifiers, not instructions).
    {
        constexpr int kDw = kTileP / 4;
        const int ngx = (ew + 3) >> 2, ngroups = ew > 0 && eh > 0 ? ngx * eh : 0;
        const float inv_ngx = 1.0f / (float)(ngx > 0 ? ngx : 1);
        const int q_step = ngx > 0 ? kFastThreads / ngx : 0, r_step = kFastThreads - q_step * ngx;   // (row, group) of item tid + 128 k stepped, not divided
        int gy = row_of(tid, ngx > 0 ? ngx : 1, inv_ngx), g = tid - gy * ngx;
        for (int i = tid; i < ngroups; i += kFastThreads) {
            const int cy = gy + 3, g0 = g;
            g += r_step; gy += q_step;
            if (g >= ngx) { g -= ngx; ++gy; }
            const uint32_t* rc = tile32 + cy * kDw + g0;
            const uint32_t a0 = rc[0], a1 = rc[1], a2 = rc[2];
            const uint32_t u0 = rc[-3 * kDw], u1 = rc[-3 * kDw + 1], d0 = rc[3 * kDw], d1 = rc[3 * kDw + 1];
            const uint32_t V = __builtin_amdgcn_alignbyte(a1, a0, 3), P4 = __builtin_amdgcn_alignbyte(a2, a1, 2), P12 = a0;
            const uint32_t P8 = __builtin_amdgcn_alignbyte(u1, u0, 3), P0 = __builtin_amdgcn_alignbyte(d1, d0, 3);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = (int)((V >> (8 * k)) & 0xffu);
                const int p0 = (int)((P0 >> (8 * k)) & 0xffu), p8 = (int)((P8 >> (8 * k)) & 0xffu), p4 = (int)((P4 >> (8 * k)) & 0xffu), p12 = (int)((P12 >> (8 * k)) & 0xffu);
                const int most = min(max(p0, p8), max(p4, p12)), least = max(min(p0, p8), min(p4, p12));
                if ((most > v + th || least < v - th) && 4 * g0 + k < ew) s_list[atomicAdd(&s_nlist, 1)] = (uint16_t)(cy * TW + 3 + 4 * g0 + k);
            }
        }
    }
    __syncthreads();
    // Pass 1, the pixels that passed, packed densely over the lanes: the segment test proper (two 16-bit masks, "9 contiguous"
    // by shifts); the list is compacted in place -- a round reads its kFastThreads entries before any of them is overwritten, and what a
    // round appends lies below the entries of the later rounds.
    {
        const int npre = s_nlist;
        __syncthreads();
        if (tid == 0) s_nlist = 0;
        __syncthreads();
        for (int base = 0; base < npre; base += kFastThreads) {
            const int k = base + tid;
            uint32_t entry = 0;
            if (k < npre) {
                const int at = s_list[k], cy = at / TW, cx = at - cy * TW;
                int p[16];
                const int v = circle(cx, cy, p);
                uint32_t mb = 0, md = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    mb |= (uint32_t)(p[j] > v + th) << j;
                    md |= (uint32_t)(p[j] < v - th) << j;
                }
                const uint32_t pol = (has_arc9(md) ? 1u : 0u) | (has_arc9(mb) ? 2u : 0u);  // bit 0: darker arc, bit 1: brighter arc
                if (pol) entry = (uint32_t)at | (pol << 14);
            }
            __syncthreads();  // every entry of this round has been read
            if (entry) s_list[atomicAdd(&s_nlist, 1)] = (uint16_t)entry;
        }
    }
    __syncthreads();
    // Pass 2, survivors only, packed densely over the lanes: S = the largest arc contrast of the polarity that has an arc
    // (the other polarity cannot exceed the threshold, so it cannot be the maximum).
    nlist = s_nlist;
    for (int k = tid; k < nlist; k += kFastThreads) {
        const uint32_t e = s_list[k];
        const int at = e & 0x3fff, cy = at / TW, cx = at - cy * TW;
        int p[16];
        const int v = circle(cx, cy, p);
        int S = 0;
        if (e & 0x4000) S = arc_contrast<true>(p, v);
        if (e & 0x8000) S = max(S, arc_contrast<false>(p, v));
        score[at] = (uint8_t)S;
    }
    __syncthreads();
    // Pass 3, survivors only: 3x3 strict non-max suppression at the attempt's threshold (every survivor has S > th; a neighbour
    // counts with its score only where it is a corner at that threshold, cv::FAST's score buffer semantics)
    int my_kept = 0;
    for (int k = tid; k < nlist; k += kFastThreads) {
        const int at = s_list[k] & 0x3fff;
        const uint8_t* sc = score + at;
        const int S = sc[0];
        const int nb[8] = {sc[-TW - 1], sc[-TW], sc[-TW + 1], sc[-1],
                           sc[1], sc[TW - 1], sc[TW], sc[TW + 1]};
        bool keep = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n_th = nb[j] > th ? nb[j] - 1 : 0;
            keep = keep && (S - 1 > n_th);
        }
        s_list[k] = (uint16_t)(at | (keep ? 0x4000 : 0));  // the polarity bits have served (pass 2): bit 14 = kept
        my_kept += keep ? 1 : 0;
    }
    if (my_kept) atomicAdd(&s_cnt_ini, my_kept);
    __syncthreads();
    if (s_cnt_ini > 0) break;  // uniform: every lane reads the same count
    }
    // Emission in row-major order (the order cv::FAST returns the keypoints in): the kept survivors are gathered (a few tens
    // per cell; strict 3x3 maxima: at most one per 2x2 block), the rank of each among them is its output slot.
    for (int k = tid; k < nlist; k += kFastThreads)
        if (s_list[k] & 0x4000) {
            const int slot = atomicAdd(&s_nkept, 1);
            if (slot < kMaxKept) s_kept[slot] = (uint16_t)(s_list[k] & 0x3fff);
        }
    __syncthreads();
    const int nkept = s_nkept, nk = min(nkept, kMaxKept);
    uint32_t* out = slab + (size_t)img * slab_img_stride + c.slab_off;
    for (int k = tid; k < nk; k += kFastThreads) {
        const int at = s_kept[k];
        int rank = 0;
        for (int j = 0; j < nk; ++j) rank += (int)s_kept[j] < at ? 1 : 0;
        const int cy = at / TW, cx = at - cy * TW;
        const uint32_t scv = (uint32_t)score[at] - 1;
        // candidate coordinates in the border-free frame of the level (SF/src/ORBextractor.cc:833-838)
        const uint32_t px = (uint32_t)(c.x0 + cx - kMinBorder), py = (uint32_t)(c.y0 + cy - kMinBorder);
        if (rank < c.slab_cap) out[rank] = (py << 20) | (px << 8) | scv;
    }
    if (tid == 0) cell_counts[(size_t)img * ncells + cell] = s_nkept;
    }
}

// Ordered concatenation of the per-cell candidate lists of one (image, level) into a dense list: cell-major order,
// the order vToDistributeKeys is filled in (SF/src/ORBextractor.cc:829-842).
__global__ __launch_bounds__(256) void k_compact_cells(const FastCell* __restrict__ cells, const int* __restrict__ level_cell_begin,
                                                       const int* __restrict__ cell_counts, int ncells,
                                                       const uint32_t* __restrict__ slab, size_t slab_img_stride,
                                                       uint32_t* __restrict__ dense, const int* __restrict__ level_dense_off,
                                                       int* __restrict__ level_counts, int nlevels) {
    __shared__ int s_off[kMaxCellsPerLevel + 1];
    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int cb = level_cell_begin[level], ce = level_cell_begin[level + 1], n = ce - cb;
    const int* cnt = cell_counts + (size_t)img * ncells + cb;
    // exclusive scan of the n (<= kMaxCellsPerLevel) cell counts: 8 per thread, wave scan, cross-wave fix-up
    __shared__ int s_wsum[4];
    {
        int v[8], run = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid * 8 + k;
            v[k] = run;
            run += idx < n ? cnt[idx] : 0;
        }
        int incl = run;
        const int lane = tid & 63, wave = wave_in_block();
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < wave; ++k) wbase += s_wsum[k];
        const int excl = wbase + incl - run;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tid * 8 + k;
            if (idx <= n) s_off[idx] = excl + v[k];
        }
        if (tid == 0) level_counts[img * nlevels + level] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    }
    __syncthreads();
    const uint32_t* in = slab + (size_t)img * slab_img_stride;
    uint32_t* out = dense + (size_t)img * slab_img_stride + level_dense_off[level];
    const int lane = tid & 63, wave = wave_in_block();
    for (int k = wave; k < n; k += 4) {
        const int m = s_off[k + 1] - s_off[k];
        const uint32_t* ci = in + cells[cb + k].slab_off;
        for (int t = lane; t < m; t += 64) out[s_off[k] + t] = ci[t];
    }
}

// ------------------------------------------------------------------------------------------------------
// 7x7 sigma=2 Gaussian on 8-bit, REFLECT_101, OpenCV's fixed-point separable path
// (SF/src/ORBextractor.cc:1105-1106): taps {18,34,48,56,48,34,18}/256, 8.8 horizontal, 16.16 vertical,
// round to nearest.
// ------------------------------------------------------------------------------------------------------
// Every level in one launch, streaming: a wavefront owns a strip of 256 columns x 32 rows; a lane loads the dword of its four
// pixels per row at its byte address (rows of a caller image may start at any byte), takes its neighbours' pixels from the lanes
// beside it (one DPP move each way: wave_shr:1 / wave_shl:1 -- a vector instruction, no trip through the LDS as a shuffle is), filters
// horizontally into four 8.8 values and keeps the last six rows of them in registers for the vertical pass: every pixel is read
// once from HBM (plus a 6-row halo per 32 rows) and no LDS is used.
// Round 6: the kernel took the same time with every store suppressed (a probe, removed again): it is bound by instruction issue, like the
// whole extraction (3.2 G wavefront vector instructions per 1024 images at ~2 ns each per SIMD -- tools/probes/valu_rate.hip: v_dot4 /
// v_dot2 / v_perm / v_alignbyte 2.0 ns, v_add / v_fma 1.2 ns, a DPP move 2.7 ns, a shuffle through the LDS 10 ns -- are 6 of its 9.7 ms).
// It had 116 VGPRs (4 waves per SIMD), 89 branches (a byte-wise path per row for the two rows whose dword loads could leave the buffer,
// divergent border fix-ups), two LDS shuffles per row, three loads per row of which two served one lane each, and every row address as
// 64-bit vector arithmetic (the wavefront's number was not known to be uniform).  Now: 52 VGPRs, 47 vector instructions per row (58): the last
// lane of a row loads the row's LAST dword and shifts (no load leaves the buffer: no guarded rows), ONE edge load per row serves lane 0
// (left neighbour dword) and lane 63 (right neighbour dword), the border fix-ups are selects, and the vertical pass keeps ONE set of row
// pairs (rows in pairs: the even row of a pair takes three v_dot2 on the pairs (r-6, r-5), (r-4, r-3), (r-2, r-1) plus its own tap, the
// odd row four v_dot2 on the same pairs with the taps shifted by one row plus the pair it completes) instead of both parities.
constexpr int kStripW = 256, kStripRows = 32, kStripWaves = 4;
struct BlurWork { int32_t first_block[kMaxLevels + 1]; int32_t nsx[kMaxLevels], ncy[kMaxLevels]; int32_t nlevels, nimg; };

__device__ __forceinline__ int reflect101(int v, int n) {
    v = v < 0 ? -v : (v >= n ? 2 * n - 2 - v : v);
    return min(max(v, 0), n - 1);
}
// lane i receives lane i - 1's value (lane 0: `edge`) / lane i + 1's value (lane 63: `edge`): gfx9 DPP wave shifts (tools/probes/dpp_probe.hip)
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t edge) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v, uint32_t edge) { return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xf, 0xf, false); }

// ROUNDED selects the fixed-point taps of cv::GaussianBlur(7 x 7, sigma 2) on CV_8U (what getFixedpointGaussianKernel hands to the 8.8 path):
//   false: {18, 34, 48, 56, 48, 34, 18}, sum 256 -- the error-diffused quantiser (getGaussianKernelFixedPoint_ED of the later 3.4 / 4.x releases);
//   true:  {18, 34, 49, 55, 49, 34, 18}, sum 257 -- every tap rounded to nearest on its own (the earlier bit-exact path), result saturated.
// Which of the two OpenCV 4.2 contains cannot be checked here (no OpenCV): DESIGN.md section 2; the default is the first.
template <bool ROUNDED>
__global__ __launch_bounds__(64 * kStripWaves) void k_blur7_strips(LevelTable src, LevelTable dst, BlurWork wk) {
    // (an XCD-contiguous block order, which helps the cell and gather kernels, made this streaming kernel slower: 0.32 -> 0.38 ms)
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxLevels; ++l) if (l < wk.nlevels && (int)blockIdx.x >= wk.first_block[l]) level = l;
    const LevelDesc S = src.lv[level], D = dst.lv[level];
    int b = blockIdx.x - wk.first_block[level];
    const int per_img = wk.nsx[level] * wk.ncy[level];
    const int img = b / per_img;
    b -= img * per_img;
    const int cy = b / wk.nsx[level], sx = b - cy * wk.nsx[level];
    // (the wavefront's number through readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all 64 lanes, and without
    // that every row's address was 64-bit vector arithmetic in every lane -- with it the rows are scalar registers and the loads take a 32-bit lane offset)
    const int wave = wave_in_block(), lane = threadIdx.x & 63;
    const int y0 = (cy * kStripWaves + wave) * kStripRows;
    if (y0 >= S.h) return;
    const int w = S.w, h = S.h;
    const int x = sx * kStripW + 4 * lane;
    const uint8_t* base = S.img + (size_t)img * S.img_stride;
    uint8_t* out = const_cast<uint8_t*>(D.img) + (size_t)img * D.img_stride;
    const bool inside = x < w;                        // the lane owns at least one pixel
    const int y1 = min(y0 + kStripRows, h);
    constexpr uint32_t T2 = ROUNDED ? 49u : 48u, T3 = ROUNDED ? 55u : 56u;
    if (w < 16) {
        // an image narrower than four dwords (no pyramid level of a camera image is): pixel by pixel with REFLECT_101, the same arithmetic
        const uint32_t taps[7] = {18u, 34u, T2, T3, T2, 34u, 18u};
        for (int oy = y0; oy < y1; ++oy)
            for (int k = 0; k < 4 && x + k < w; ++k) {
                uint32_t acc = 32768u;
                for (int dy = -3; dy <= 3; ++dy) {
                    const uint8_t* row = base + (size_t)reflect101(oy + dy, h) * S.pitch;
                    uint32_t hz = 0;
                    for (int dx = -3; dx <= 3; ++dx) hz += taps[dx + 3] * row[reflect101(x + k + dx, w)];
                    acc += taps[dy + 3] * hz;
                }
                out[(size_t)oy * D.pitch + x + k] = (uint8_t)(ROUNDED ? min(acc >> 16, 255u) : (acc >> 16));
            }
        return;
    }
    // Border lanes: the lane at x = 0 mirrors its left neighbour dword out of its own pixels, the last lane of the row mirrors
    // the bytes past the row end out of its own and its left neighbour's pixels (REFLECT_101), each with one v_perm_b32 whose
    // selector is fixed for the whole strip.  Byte numbering of v_perm(S0, S1): S1 = bytes 0..3, S0 = bytes 4..7.
    const int nvalid = w - x;                          // pixels the lane owns when it is the last one: 1..4
    const bool is_first = x == 0, is_last = inside && nvalid <= 4;
    uint32_t sel_cur = 0x07060504u, sel_hi_last = 0;   // over (S0 = cur, S1 = lo)
    uint32_t shift_last = 0;                           // the last lane loads the row's last dword: its pixels are that dword's top bytes
    if (is_last) {
        sel_cur = 0;
        for (int i = 0; i < 4; ++i) {
            const int sc = i < nvalid ? 4 + i : 4 + 2 * nvalid - 2 - i;  // cur byte i
            const int sh_ = 4 + 2 * nvalid - 6 - i;                        // hi byte i (only needed while it maps into lo / cur)
            sel_cur |= (uint32_t)max(sc, 0) << (8 * i);
            sel_hi_last |= (uint32_t)max(sh_, 0) << (8 * i);
        }
        shift_last = 8u * (uint32_t)(4 - nvalid);
    }
    // lane 63 of a strip whose right neighbour dword [x + 4, x + 8) crosses the row end (the last lane sits in the next strip): it loads
    // the row's last dword instead, in which pixel x + 4 + i is byte i + (x + 8 - w); a pixel beyond the row is mirrored to
    // 2 (w - 1) - (x + 4 + i), which lies 2 (w - x) - 6 - i pixels right of x -- in that dword again or in the lane's own.  Over (S0 = the loaded dword, S1 = cur):
    uint32_t sel_hi63 = 0x07060504u;
    const bool hi_crosses = lane == 63 && x + 4 < w && x + 8 > w;
    if (hi_crosses) {
        sel_hi63 = 0;
        for (int i = 0; i < 4; ++i) {
            const int p = x + 4 + i < w ? 4 + i : 2 * (w - x) - 6 - i;           // the pixel's place relative to x (mirrored when beyond the row)
            sel_hi63 |= (uint32_t)(p >= 4 ? p + (x + 8 - w) : p) << (8 * i);       // in the loaded dword (shifted by x + 8 - w) or in cur
        }
    }
    // byte offsets of the lane's two loads inside a row: its own dword (the row's first dword for a lane beyond the row: loaded, not used)
    // and the edge dword -- lane 0: the dword left of the strip, lane 63: the one right of it (none of the two: the row's first dword)
    const uint32_t off_m = inside ? (uint32_t)(is_last ? w - 4 : x) : 0u;
    const uint32_t off_e = lane == 0 && x > 0 ? (uint32_t)(x - 4) : (lane == 63 && x + 4 < w ? (uint32_t)min(x + 4, w - 4) : 0u);
    const uint32_t keep = inside ? 0xffffffffu : 0u;
    // P[j][k]: a pair of rows' horizontal sums (8.8) of pixel k as two u16, older row low; E[k]: the even row of the pair being formed
    uint32_t P[3][4], E[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) P[j][k] = 0;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t {
        return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
    };
    constexpr uint32_t TA = 18u | (34u << 8) | (T2 << 16) | (T3 << 24), TB = T2 | (34u << 8) | (18u << 16);
    // even row r of a pair: rows (r-6, r-5) (r-4, r-3) (r-2, r-1) + 18 hz(r); odd row r + 1: the same pairs one tap on, + the pair (r, r + 1)
    constexpr uint32_t A0 = 18u | (34u << 16), A1 = T2 | (T3 << 16), A2 = T2 | (34u << 16);
    constexpr uint32_t B0 = 18u << 16, B1 = 34u | (T2 << 16), B2 = T3 | (T2 << 16), B3 = 34u | (18u << 16);
    const int r_end = y1 + 3;
    for (int r0 = y0 - 3; r0 < r_end; r0 += 6) {
        // The six rows of a round are REQUESTED first and used afterwards: with the load next to its use every row cost the wavefront a
        // full trip to memory.
        uint32_t ld_m[6], ld_e[6];
        auto load4 = [](const uint8_t* p) -> uint32_t { uint32_t t; __builtin_memcpy(&t, p, 4); return t; };
#pragma unroll
        for (int ph = 0; ph < 6; ++ph) {
            const int ry = reflect101(min(r0 + ph, r_end - 1), h);
            const uint8_t* row = base + (size_t)ry * S.pitch;
            ld_m[ph] = load4(row + off_m);
            ld_e[ph] = load4(row + off_e);
        }
#pragma unroll
        for (int ph = 0; ph < 6; ++ph) {
            const int r = r0 + ph;
            if (r < r_end) {
                uint32_t cur = (ld_m[ph] >> shift_last) & keep;
                uint32_t lo = from_lane_below(cur, ld_e[ph]);
                cur = __builtin_amdgcn_perm(cur, lo, sel_cur);                  // identity except in the last lane
                uint32_t hi = from_lane_above(cur, ld_e[ph]);
                hi = __builtin_amdgcn_perm(hi, cur, sel_hi63);                  // identity except where the row ends inside lane 63's right dword
                const uint32_t hi_last = __builtin_amdgcn_perm(cur, lo, sel_hi_last);
                hi = is_last ? hi_last : hi;
                const uint32_t lo_first = __builtin_amdgcn_perm(hi, cur, 0x01020304u);  // pixels -4..-1 = pixels 4, 3, 2, 1
                lo = is_first ? lo_first : lo;
                // horizontal: pixel k needs bytes k-3 .. k+3 around its own: two 4-byte windows, two v_dot4_u32_u8
                uint32_t hz[4];
                hz[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(cur, lo, 1), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(hi, cur, 1), TB, 0u, false), false);
                hz[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(cur, lo, 2), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(hi, cur, 2), TB, 0u, false), false);
                hz[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(cur, lo, 3), TA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(hi, cur, 3), TB, 0u, false), false);
                hz[3] = __builtin_amdgcn_udot4(cur, TA, __builtin_amdgcn_udot4(hi, TB, 0u, false), false);
                const int j0 = (ph / 2) % 3, j1 = (ph / 2 + 1) % 3, j2 = (ph / 2 + 2) % 3;   // the pairs (r-6, r-5), (r-4, r-3), (r-2, r-1) of an even row
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t acc;
                    if (ph % 2 == 0) {
                        acc = dot2(P[j0][k], A0, dot2(P[j1][k], A1, dot2(P[j2][k], A2, 18u * hz[k] + 32768u)));
                        E[k] = hz[k];
                    } else {
                        const uint32_t n = E[k] | (hz[k] << 16);
                        acc = dot2(P[j0][k], B0, dot2(P[j1][k], B1, dot2(P[j2][k], B2, dot2(n, B3, 32768u))));
                        P[j0][k] = n;                                            // the oldest pair's place: (r-6, r-5) is not needed again
                    }
                    packed |= (ROUNDED ? min(acc >> 16, 255u) : (acc >> 16)) << (8 * k);  // taps summing to 257 can reach 256: saturate_cast
                }
                const int oy = r - 3;
                // (the last lane's dword may end up to three bytes beyond the row: inside the level's pitch, which launch_blur_all checks)
                if (oy >= y0 && inside) *reinterpret_cast<uint32_t*>(out + (size_t)oy * D.pitch + x) = packed;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Orientation (IC_Angle, SF/src/ORBextractor.cc:50-77) on the un-blurred level and the rotated-BRIEF descriptor
// (computeOrbDescriptor, :81-120) on the blurred level: half a wavefront (32 lanes) per keypoint.  Lane u-16
// sums one column of the radius-15 disc; lane i then produces descriptor byte i (8 pattern pairs).
// ------------------------------------------------------------------------------------------------------
__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16];

__global__ __launch_bounds__(256) void k_orient_describe(LevelTable raw, LevelTable blurred, ScaleTable sc,
                                                         const DevKeypoint* __restrict__ kps, const int32_t* __restrict__ n_kp, int first_image,
                                                         int n_images, int kp_stride, float* __restrict__ angles, float* __restrict__ angles_dev, uint8_t* __restrict__ desc,
                                                         MatchKey* __restrict__ mkeys, uint8_t* __restrict__ desc_dev, int reps) {
    // XCD-contiguous block order (as in k_fast_cells): the keypoints come image by image and level by level, and XCD k works on the
    // k-th eighth of the list, so an image's levels are fetched into ONE L2 instead of all eight (307 MB of HBM reads per 64 k keypoints
    // before, against 90 MB of pyramid levels)
    // Keypoint slots: image i owns kp_stride of them and fills the first n_kp[i] (counts known on the device only: k_quadtree_gather)
    // A workgroup takes `reps` consecutive blocks of eight keypoints of its XCD's share, one after the other (batches: fewer, longer workgroups)
    const int nslot = n_images * kp_stride, nblk = (nslot + 7) / 8, per_xcd = (nblk + 7) / 8;
    const int lane = threadIdx.x & 31, kpi = threadIdx.x >> 5;
    // The two patches of a keypoint -- 31 x 31 of the level image (orientation), 37 x 37 of the blurred level (the rotated pattern
    // reaches 18 px: its corner points are (+-13, +-13)) -- are staged in LDS, 9 + 12 coalesced dword loads per lane, instead of 31 + 16
    // byte gathers per lane from global memory (the 16 descriptor samples of a lane fall on ~16 different cache lines).  The dwords are
    // read at the patch's own byte address (unaligned loads), so pixel (u, v) of a patch of half-size R is simply byte
    // (v + R) * 4 * DW + u + R, DW dwords per row: the kernel is bound by VALU issue, and the per-access alignment term of the
    // aligned-dword form (round 2) was a fifth of its instructions (1.92 -> ? ms per 1024 images).
    constexpr int kRawR = 15, kRawDw = 9, kBlurR = 18, kBlurDw = 10;
    __shared__ uint32_t s_raw[8][(2 * kRawR + 1) * kRawDw], s_blur[8][(2 * kBlurR + 1) * kBlurDw], s_icmask[kRawR + 1][kRawDw];
    if ((int)threadIdx.x < (kRawR + 1) * kRawDw) {  // the circular patch of the orientation as byte masks per row |v| and dword
        const int av = (int)threadIdx.x / kRawDw, j = (int)threadIdx.x - av * kRawDw, um = c_umax[av];
        uint32_t m = 0;
        for (int k = 0; k < 4; ++k) {
            const int u = 4 * j + k - kRawR;
            if (u >= -um && u <= um) m |= 0xffu << (8 * k);
        }
        s_icmask[av][j] = m;
    }
    for (int rep = 0; rep < reps; ++rep) {
    const int in_xcd = ((int)blockIdx.x >> 3) * reps + rep;
    const int logical = ((int)blockIdx.x & 7) * per_xcd + in_xcd;
    if (in_xcd >= per_xcd || logical >= nblk) break;  // whole workgroup
    if (rep) __syncthreads();                          // the block before has read its patches
    const int slot = (logical * 256 + (int)threadIdx.x) >> 5;
    bool active = slot < nslot;
    int slot_img = 0;
    if (active) {
        slot_img = first_image + slot / kp_stride;
        active = slot % kp_stride < n_kp[slot_img];
    }
    const int g = first_image * kp_stride + slot;
    DevKeypoint kp{};
    if (active) kp = kps[g];
    const int level = kp.img_level & 0xff, img = kp.img_level >> 8;
    const int x = (kp.packed >> 8) & 0xfff, y = kp.packed >> 20;
    auto stage = [&](const LevelDesc& D, auto r_tag, auto dw_tag, uint32_t* dst) {
        constexpr int R = decltype(r_tag)::value, DW = decltype(dw_tag)::value, N = (2 * R + 1) * DW, Q = (N + 31) / 32;
        const uint8_t* ibase = D.img + (size_t)img * D.img_stride;
        const uint32_t off0 = (uint32_t)(y - R) * (uint32_t)D.pitch + (uint32_t)(x - R);  // the patch's first byte inside the image's buffer
        // Would a dword of the patch cross the end of the image's buffer (only a patch in the last rows of an image whose pitch is its
        // width can)?  Decided once per keypoint: the staging loop is a third of the kernel's instructions, and a 64-bit compare and
        // select per load were half of the loop.
        const bool near_end = (size_t)off0 + (size_t)(2 * R) * D.pitch + 4 * DW > D.img_stride;
        uint32_t v[Q];
        if (!near_end) {
            // all loads first (dwords at byte addresses), 32-bit offsets, (row, j) of lane + 32 q stepped instead of divided
            const int row0 = lane / DW;
            int j = lane - DW * row0;
            uint32_t off = off0 + (uint32_t)row0 * (uint32_t)D.pitch + 4u * (uint32_t)j;
            const uint32_t step = (uint32_t)(32 / DW) * (uint32_t)D.pitch + 4u * (32 % DW), wrap = (uint32_t)D.pitch - 4u * DW;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool in = lane + 32 * q < N;
                uint32_t w;
                __builtin_memcpy(&w, ibase + (in ? off : 0u), 4);
                v[q] = w;
                j += 32 % DW; off += step;
                if (j >= DW) { j -= DW; off += wrap; }
            }
        } else {
            const uint8_t* corner = ibase + off0;
            const uint8_t* end = ibase + D.img_stride;  // nothing is read past the image's own buffer
#pragma unroll 1
            for (int q = 0; q < Q; ++q) {
                const int t = min(lane + 32 * q, N - 1), row = t / DW, j = t - DW * row;
                const uint8_t* a = corner + (size_t)row * D.pitch + 4 * j;
                uint32_t w = 0;
                if (a + 4 <= end) __builtin_memcpy(&w, a, 4);
                else for (int k = 0; k < 4; ++k) if (a + k < end) w |= (uint32_t)a[k] << (8 * k);
                v[q] = w;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) if (lane + 32 * q < N) dst[lane + 32 * q] = v[q];
    };
    if (active) {
        stage(raw.lv[level], std::integral_constant<int, kRawR>{}, std::integral_constant<int, kRawDw>{}, s_raw[kpi]);
        stage(blurred.lv[level], std::integral_constant<int, kBlurR>{}, std::integral_constant<int, kBlurDw>{}, s_blur[kpi]);
    }
    __syncthreads();
    if (!active) continue;
    const uint8_t* pblur = reinterpret_cast<const uint8_t*>(s_blur[kpi]);
    auto blur_at = [&](int u, int v) -> int { return pblur[(v + kBlurR) * 4 * kBlurDw + u + kBlurR]; };
    {
        // IC_Angle (:61-100): m10 = sum u I(u, v), m01 = sum v I(u, v) over the circular patch |u| <= umax[|v|].  Lane = row v: its 36
        // staged bytes as nine dwords, the bytes outside the circle masked off (s_icmask[|v|], built once per workgroup), then per dword
        // one v_dot4 with the byte weights u + 15 and one with ones -- integer sums, any order gives the reference's integers.  (A lane
        // per column walking the 31 rows byte by byte was a quarter of the kernel's instructions.)
        int m10 = 0, m01 = 0;
        if (lane < 2 * kRawR + 1) {
            const int v = lane - kRawR, av = v < 0 ? -v : v;
            const uint32_t* rowp = s_raw[kpi] + lane * kRawDw;
            const uint32_t* mk = s_icmask[av];
            uint32_t wsum = 0, sum = 0;
#pragma unroll
            for (int j = 0; j < kRawDw; ++j) {
                const uint32_t d = rowp[j] & mk[j];
                const uint32_t wts = (uint32_t)(4 * j) * 0x01010101u + 0x03020100u;  // u + 15 of the dword's four bytes
                wsum = __builtin_amdgcn_udot4(d, wts, wsum, false);
                sum = __builtin_amdgcn_udot4(d, 0x01010101u, sum, false);
            }
            m10 = (int)wsum - kRawR * (int)sum;
            m01 = v * (int)sum;
        }
        m10 = half_wave_sum_i32(m10);
        m01 = half_wave_sum_i32(m01);
        const float angle = fast_atan2_deg((float)m01, (float)m10);
        if (lane == 0) {
            angles[g] = angle;
            angles_dev[g] = angle;
            // level-0 coordinates as the reference scales them (SF/src/ORBextractor.cc:1122-1124)
            mkeys[g] = MatchKey{__fmul_rn((float)x, sc.scale[level]), __fmul_rn((float)y, sc.scale[level]), level};
        }
        constexpr float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
        float a, b;
        det_sincosf(__fmul_rn(angle, factorPI), &b, &a);
        const int8_t* pat = c_pattern + lane * 32;
        int val = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int t[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float px = (float)pat[4 * k + 2 * j], py = (float)pat[4 * k + 2 * j + 1];
                const int ry = cv_round(__fadd_rn(__fmul_rn(px, b), __fmul_rn(py, a)));
                const int rx = cv_round(__fsub_rn(__fmul_rn(px, a), __fmul_rn(py, b)));
                t[j] = blur_at(rx, ry);
            }
            val |= (t[0] < t[1]) << k;
        }
        desc[(size_t)g * 32 + lane] = (uint8_t)val;
        desc_dev[(size_t)g * 32 + lane] = (uint8_t)val;
    }
    }
}

// ---- launch wrappers (host side of this translation unit) ----------------------------------------------
void launch_resize(const LevelDesc& src, const LevelDesc& dst, const int* xofs, const short* ialpha, const int* yofs,
                   const short* ibeta, int nimg, hipStream_t st) {
    static const bool pixel_form = getenv("TC2LI_RESIZE_PIXELS") && atoi(getenv("TC2LI_RESIZE_PIXELS")) != 0;  // the round-1 kernel: four pixels per thread
    if (pixel_form) {
        const int gx = (dst.w + 255) / 256, gy = (dst.h + 3) / 4;
        TC2LI_LAUNCH(k_resize_linear, dim3(((gx * gy * nimg + 7) / 8) * 8), dim3(256), 0, st, src.img, src.pitch, src.img_stride, src.w, src.h,
                     const_cast<uint8_t*>(dst.img), dst.pitch, dst.img_stride, dst.w, dst.h, xofs, ialpha, yofs, ibeta, gx, gy, nimg);
        return;
    }
    // one wavefront per (256 columns, `rows` rows); four wavefronts per workgroup, the workgroup count a multiple of 8 (XCDs).  A batch
    // walks kResizeRows rows per wavefront (the shared source rows' horizontal pass is reused); a few images are a chain of seven
    // latency-bound launches, where four rows per wavefront put four times the wavefronts on the GPU (a stereo pair: 16 -> ? us per launch)
    const int rows = nimg >= 32 ? kResizeRows : 4;
    const int gx = (dst.w + 255) / 256, gy = (dst.h + rows - 1) / rows;
    const int per_xcd = (gx * gy * nimg + 7) / 8, wg_per_xcd = (per_xcd + 3) / 4;
    TC2LI_LAUNCH(k_resize_strips, dim3(wg_per_xcd * 8), dim3(256), 0, st, src.img, src.pitch, src.img_stride, src.w, src.h,
                 const_cast<uint8_t*>(dst.img), dst.pitch, dst.img_stride, dst.w, dst.h, xofs, ialpha, yofs, ibeta, gx, gy, nimg, rows);
}

void launch_resize_tail(const LevelTable& lv, const int* const* xofs, const short* const* ialpha, const int* const* yofs, const short* const* ibeta, int l0, int n_levels,
                        int nimg, hipStream_t st) {
    if (nimg <= 0 || l0 >= n_levels) return;
    ResizeTables tb{};
    for (int l = l0; l < n_levels; ++l) { tb.xofs[l] = xofs[l]; tb.ialpha[l] = ialpha[l]; tb.yofs[l] = yofs[l]; tb.ibeta[l] = ibeta[l]; }
    TC2LI_LAUNCH(k_resize_tail, dim3(nimg), dim3(kResizeTailThreads), 0, st, lv, tb, l0, n_levels);
}

void launch_fast(const LevelTable& levels, const FastCell* cells, int ncells, int ini_th, int min_th, uint32_t* slab,
                 size_t slab_img_stride, int* cell_counts, int nimg, const int* small_ids, int n_small, const int* large_ids, int n_large,
                 hipStream_t st) {
    // cell windows are 35-px cells + 6: 48 x 48 holds all but the levels whose height leaves one or two tall rows of cells; those go
    // through the full-size variant (a third of the LDS-limited occupancy), each window class in its own launch
    static const int kCellsEnv = getenv("TC2LI_FAST_CELLS_PER_WG") ? atoi(getenv("TC2LI_FAST_CELLS_PER_WG")) : 4;
    const int cpw = nimg >= 32 ? std::max(1, std::min(kCellsEnv, 16)) : 1;  // a few images: every cell its own workgroup (the launch is latency-bound)
    auto grid = [&](int n_ids) { const int per_xcd = (n_ids * nimg + 7) / 8; return ((per_xcd + cpw - 1) / cpw) * 8; };
    if (n_small > 0)
        TC2LI_LAUNCH((k_fast_cells<48, 48>), dim3(grid(n_small)), dim3(kFastThreads), 0, st, levels, cells, ini_th, min_th, slab,
                           slab_img_stride, cell_counts, ncells, nimg, small_ids, n_small, cpw);
    if (n_large > 0)
        TC2LI_LAUNCH((k_fast_cells<kFastTileH, kFastTilePitch>), dim3(grid(n_large)), dim3(kFastThreads), 0, st, levels, cells, ini_th,
                           min_th, slab, slab_img_stride, cell_counts, ncells, nimg, large_ids, n_large, cpw);
}

void launch_compact(const FastCell* cells, const int* level_cell_begin, const int* cell_counts, int ncells,
                    const uint32_t* slab, size_t slab_img_stride, uint32_t* dense, const int* level_dense_off,
                    int* level_counts, int nlevels, int nimg, hipStream_t st) {
    TC2LI_LAUNCH(k_compact_cells, dim3(nlevels, nimg), dim3(256), 0, st, cells, level_cell_begin, cell_counts,
                       ncells, slab, slab_img_stride, dense, level_dense_off, level_counts, nlevels);
}

void launch_blur_all(const LevelTable& src, const LevelTable& dst, int nlevels, int nimg, bool rounded_taps, hipStream_t st) {
    BlurWork wk{};
    wk.nlevels = nlevels; wk.nimg = nimg;
    int total = 0;
    for (int l = 0; l < nlevels; ++l) {
        const LevelDesc& L = src.lv[l];
        wk.first_block[l] = total;
        wk.nsx[l] = (L.w + kStripW - 1) / kStripW;
        wk.ncy[l] = (L.h + kStripRows * kStripWaves - 1) / (kStripRows * kStripWaves);
        total += wk.nsx[l] * wk.ncy[l] * nimg;
    }
    for (int l = nlevels; l <= kMaxLevels; ++l) wk.first_block[l] = total;
    for (int l = 0; l < nlevels; ++l)   // the kernel stores whole dwords: a destination row holds its width rounded up to four bytes (the extractor's levels: to 64)
        if (dst.lv[l].pitch < ((dst.lv[l].w + 3) & ~3)) { fprintf(stderr, "tc2li: blur destination pitch %d below the dwords of a row of %d pixels\n", dst.lv[l].pitch, dst.lv[l].w); abort(); }
    if (total && rounded_taps) TC2LI_LAUNCH(k_blur7_strips<true>, dim3(total), dim3(64 * kStripWaves), 0, st, src, dst, wk);
    else if (total) TC2LI_LAUNCH(k_blur7_strips<false>, dim3(total), dim3(64 * kStripWaves), 0, st, src, dst, wk);
}

void launch_orient_describe(const LevelTable& raw, const LevelTable& blurred, const ScaleTable& sc, const DevKeypoint* kps, const int32_t* n_kp,
                            int first_image, int n_images, int kp_stride, float* angles, float* angles_dev, uint8_t* desc, MatchKey* mkeys, uint8_t* desc_dev,
                            hipStream_t st) {
    const long nslot = (long)n_images * kp_stride;
    if (nslot <= 0) return;
    static const int kRepsEnv = getenv("TC2LI_DESCRIBE_BLOCKS_PER_WG") ? atoi(getenv("TC2LI_DESCRIBE_BLOCKS_PER_WG")) : 4;
    const int reps = n_images >= 32 ? std::max(1, std::min(kRepsEnv, 16)) : 1;
    const long per_xcd = ((nslot + 7) / 8 + 7) / 8;
    TC2LI_LAUNCH(k_orient_describe, dim3((unsigned)(((per_xcd + reps - 1) / reps) * 8)), dim3(256), 0, st, raw, blurred, sc, kps, n_kp, first_image, n_images,
                 kp_stride, angles, angles_dev, desc, mkeys, desc_dev, reps);
}

hipError_t upload_umax(const int* umax16) { return hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int)); }

}  // namespace tc2li
