// TEST INFRASTRUCTURE ONLY -- CPU oracle, never linked into or called from the product path.
//
// Restatement of the OpenCV 4.2 primitives the reference borrows on the ORB path
// (SURVEY.md Appendix B).  OpenCV is an un-vendored dependency of the reference
// (CMakeLists.txt:39 `find_package(OpenCV 4.2)`), absent from /root/reference and from this
// image, and the reference holds no test that pins these calls: PARITY UNPINNED against a
// real OpenCV build.  What is written here is the published algorithm of each call as used at
// the reference call sites quoted per function.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

// cvRound / cvFloor / cvCeil  (ORBextractor.cc:54,88,92-93,415,429-433,1148)
// cvRound is lrint(): round-half-to-even under the default rounding mode.
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRound(float v) { return (int)std::lrintf(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2(y, x) in degrees  (ORBextractor.cc:76)
static inline float fastAtan2(float y, float x) {
    const float k = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// A minimal stand-in for a single-channel 8-bit cv::Mat view.
struct Img {
    int w = 0, h = 0, stride = 0;
    std::vector<uint8_t> store;
    const uint8_t* p = nullptr;
    Img() {}
    Img(int w_, int h_) : w(w_), h(h_), stride(w_), store((size_t)w_ * h_) { p = store.data(); }
    Img(const Img& o) : w(o.w), h(o.h), stride(o.stride), store(o.store), p(o.store.empty() ? o.p : store.data()) {}
    Img(Img&& o) noexcept : w(o.w), h(o.h), stride(o.stride), store(std::move(o.store)), p(o.p) {}
    Img& operator=(Img o) noexcept {
        w = o.w; h = o.h; stride = o.stride; p = o.p;  // o.p already refers to o.store's heap block (or a foreign view)
        store = std::move(o.store);
        return *this;
    }
    static Img view(const uint8_t* ptr, int w_, int h_, int stride_) {
        Img v; v.w = w_; v.h = h_; v.stride = stride_; v.p = ptr; return v;
    }
    uint8_t* data() { return store.data(); }
    const uint8_t* row(int y) const { return p + (size_t)y * stride; }
    Img clone() const {
        Img o(w, h);
        for (int y = 0; y < h; ++y) std::memcpy(o.data() + (size_t)y * w, row(y), w);
        return o;
    }
};

// BORDER_REFLECT_101 index map: gfedcb|abcdefgh|gfedcba
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) on CV_8UC1  (ORBextractor.cc:1156)
// 11-bit fixed-point horizontal/vertical weights; see SURVEY.md Appendix B.
static inline void resizeLinearU8(const Img& src, Img& dst) {
    const int SC = 2048;
    const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    auto sat_short = [](float v) { int r = cvRound(v); return (short)std::min(32767, std::max(-32768, r)); };
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * SC);
        ialpha[2 * dx + 1] = sat_short(fx * SC);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short((1.f - fy) * SC);
        ibeta[2 * dy + 1] = sat_short(fy * SC);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hrow = [&](int sy, std::vector<int>& out) {
        sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        const uint8_t* S = src.row(sy);
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int s1 = sx + 1 < sw ? S[sx + 1] : 0;  // weight is 0 whenever sx+1 is outside
            out[dx] = S[sx] * ialpha[2 * dx] + s1 * ialpha[2 * dx + 1];
        }
    };
    uint8_t* D = dst.data();
    for (int dy = 0; dy < dh; ++dy) {
        hrow(yofs[dy], r0);
        hrow(yofs[dy] + 1, r1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        for (int x = 0; x < dw; ++x)
            D[(size_t)dy * dst.stride + x] =
                (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// cv::copyMakeBorder(src, dst, b, b, b, b, BORDER_REFLECT_101)  (ORBextractor.cc:1158-1164)
static inline Img makeBorder101(const Img& src, int b) {
    Img out(src.w + 2 * b, src.h + 2 * b);
    for (int y = 0; y < out.h; ++y) {
        const uint8_t* S = src.row(reflect101(y - b, src.h));
        uint8_t* D = out.data() + (size_t)y * out.stride;
        for (int x = 0; x < out.w; ++x) D[x] = S[reflect101(x - b, src.w)];
    }
    return out;
}

// cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1  (ORBextractor.cc:1106)
// 8-bit images take OpenCV's fixed-point separable path: 8.8 kernel whose taps sum to 256
// (error-diffused rounding of exp(-x^2/8)/sum -> {18,34,48,56,48,34,18}); the horizontal pass keeps
// 8.8 values, the vertical pass 16.16, the result is rounded to nearest.
// VARIANT (oracle_set_gauss_variant): which taps OpenCV 4.2 hands to that path cannot be checked here (no OpenCV in the container).
//   0 (default) "error-diffused": {18,34,48,56,48,34,18}, sum 256 -- getGaussianKernelFixedPoint_ED, present in the later 3.4.x / 4.x releases;
//   1 "rounded": {18,34,49,55,49,34,18}, sum 257 -- each tap of the bit-exact softdouble kernel rounded to 8.8 on its own, the earlier
//     form of getFixedpointGaussianKernel; the result is saturated (the sum can reach 256).
// Both have a golden file (tests/golden/orb_a.npz / orb_gauss_rounded.npz): the day a real OpenCV 4.2 is reachable, one
// cv::GaussianBlur of the stored image decides, and TC2LI_GAUSS_TAPS=rounded switches the product.
static const int kGaussTaps[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};
inline int& gauss_variant() { static int v = 0; return v; }
static inline void gaussianBlur7(const Img& src, Img& dst) {
    const int* kGauss7 = kGaussTaps[gauss_variant()];
    const int w = src.w, h = src.h;
    std::vector<uint16_t> tmp((size_t)w * h);
    std::vector<uint8_t> prow((size_t)w + 6);
    for (int y = 0; y < h; ++y) {
        const uint8_t* S = src.row(y);
        for (int x = -3; x < w + 3; ++x) prow[x + 3] = S[reflect101(x, w)];
        uint16_t* T = &tmp[(size_t)y * w];
        const uint8_t* P = prow.data();
        for (int x = 0; x < w; ++x)
            T[x] = (uint16_t)(kGauss7[0] * (P[x] + P[x + 6]) + kGauss7[1] * (P[x + 1] + P[x + 5]) +
                              kGauss7[2] * (P[x + 2] + P[x + 4]) + kGauss7[3] * P[x + 3]);
    }
    uint8_t* D = dst.data();
    for (int y = 0; y < h; ++y) {
        const uint16_t* R[7];
        for (int k = -3; k <= 3; ++k) R[k + 3] = &tmp[(size_t)reflect101(y + k, h) * w];
        uint8_t* O = D + (size_t)y * dst.stride;
        for (int x = 0; x < w; ++x) {
            const unsigned acc = kGauss7[0] * ((unsigned)R[0][x] + R[6][x]) + kGauss7[1] * ((unsigned)R[1][x] + R[5][x]) +
                                 kGauss7[2] * ((unsigned)R[2][x] + R[4][x]) + kGauss7[3] * (unsigned)R[3][x];
            const unsigned v = (acc + 32768u) >> 16;
            O[x] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

struct KeyPoint {
    float x = 0, y = 0, size = 0, angle = -1, response = 0;
    int octave = 0;
};

// cv::FAST(img, keypoints, threshold, true)  -- TYPE_9_16  (ORBextractor.cc:800,819)
// Streaming form with three score rows, as the library does it: rows 3..h-4 / cols 3..w-4 are tested,
// a pixel is a corner when 9 contiguous circle pixels are all > v+t or all < v-t, its score is the
// largest t for which that still holds, and a corner is kept when its score is strictly greater than
// the scores of its 8 neighbours (untested / non-corner neighbours score 0).  Output is row-major.
static const int kCircle16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                     {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static inline int fastCornerScore16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; ++k) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        for (int m = 3; m <= 8; ++m) a = std::min(a, (int)d[k + m]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        for (int m = 3; m <= 8; ++m) b = std::max(b, (int)d[k + m]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

static inline void FAST9_16(const uint8_t* img, int stride, int w, int h, int threshold, bool nms,
                            std::vector<KeyPoint>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < N; ++k) pixel[k] = kCircle16[k % 16][0] + kCircle16[k % 16][1] * stride;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> bufs((size_t)3 * w, 0);
    std::vector<int> cps((size_t)3 * (w + 1), 0);
    uint8_t* buf[3] = {bufs.data(), bufs.data() + w, bufs.data() + 2 * w};
    int* cpbuf[3] = {cps.data(), cps.data() + (w + 1), cps.data() + 2 * (w + 1)};
    for (int i = 3; i < h - 2; ++i) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3] + 1;
        std::memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; ++j, ++ptr) {
                const int v = ptr[0];
                // the library's high-speed pre-test over antipodal pairs: result-neutral, skips most pixels
                auto cls = [&](int k) { const int d = ptr[pixel[k]] - v; return d < -threshold ? 1 : (d > threshold ? 2 : 0); };
                int d = cls(0) | cls(8);
                if (d == 0) continue;
                d &= cls(2) | cls(10);
                d &= cls(4) | cls(12);
                d &= cls(6) | cls(14);
                if (d == 0) continue;
                d &= cls(1) | cls(9);
                d &= cls(3) | cls(11);
                d &= cls(5) | cls(13);
                d &= cls(7) | cls(15);
                bool corner = false;
                const int vt_lo = v - threshold, vt_hi = v + threshold;
                if (d & 1) {
                    int count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] < vt_lo) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    int count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] > vt_hi) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    if (nms) curr[j] = (uint8_t)fastCornerScore16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; ++k) {
            int j = cornerpos[k];
            int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                         score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                KeyPoint kp;
                kp.x = (float)j; kp.y = (float)(i - 1); kp.size = 7.f; kp.angle = -1.f; kp.response = (float)score;
                out.push_back(kp);
            }
        }
    }
}

}  // namespace oracle
