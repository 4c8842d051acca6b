// Deterministic scalar math shared by the gfx950 kernels: every operation is an explicitly rounded IEEE
// primitive (no FMA contraction, no fast-math), so that results match the x86 host arithmetic of the
// reference bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace tc2li {

// cvRound: round-half-to-even (SF/src/ORBextractor.cc:54,88,92-93)
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

// cv::fastAtan2 (degrees), the call at SF/src/ORBextractor.cc:76.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    constexpr float k = (float)(180.0 / 3.141592653589793238462643383279502884);
    constexpr float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    constexpr float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float)2.2204460492503131e-16;
    const bool xmaj = ax >= ay;
    const float num = xmaj ? ay : ax, den = __fadd_rn(xmaj ? ax : ay, eps);
    const float c = __fdiv_rn(num, den);
    const float c2 = __fmul_rn(c, c);
    float a = __fadd_rn(__fmul_rn(p7, c2), p5);
    a = __fadd_rn(__fmul_rn(a, c2), p3);
    a = __fadd_rn(__fmul_rn(a, c2), p1);
    a = __fmul_rn(a, c);
    if (!xmaj) a = __fsub_rn(90.f, a);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cosf/sinf for arguments in [0, 2*pi]: the sincosf algorithm of glibc >= 2.28 evaluated in double with
// explicitly rounded operations; tools/check_sincosf.c shows it equals glibc's cosf()/sinf() for every float
// in [0, 6.3] (what `(float)cos(angle)` / `(float)sin(angle)` at SF/src/ORBextractor.cc:85 produce on the host).
struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };

__device__ __forceinline__ double dmad(double a, double b, double c) { return __dadd_rn(a, __dmul_rn(b, c)); }

__device__ __forceinline__ float sincos_poly(double x, double x2, bool neg_tab, int n) {
    const double sg = neg_tab ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        const double x3 = __dmul_rn(x, x2);
        const double s1 = dmad(0x1.1107605230bc4p-7, x2, -0x1.994eb3774cf24p-13);
        const double x7 = __dmul_rn(x3, x2);
        const double s = dmad(x, x3, -0x1.555545995a603p-3);
        return (float)dmad(s, x7, s1);
    }
    const double x4 = __dmul_rn(x2, x2);
    const double c2 = dmad(sg * -0x1.6c087e89a359dp-10, x2, sg * 0x1.99343027bf8c3p-16);
    const double c1 = dmad(sg * 0x1p0, x2, sg * -0x1.ffffffd0c621cp-2);
    const double x6 = __dmul_rn(x4, x2);
    const double c = dmad(c1, x4, sg * 0x1.55553e1068f19p-5);
    return (float)dmad(c, x6, c2);
}

__device__ __forceinline__ void det_sincosf(float y, float* sin_out, float* cos_out) {
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < ((__float_as_uint(0x1.921FB6p-1f) >> 20) & 0x7ff)) {
        const double x2 = __dmul_rn(x, x);
        if (top < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ff)) { *sin_out = y; *cos_out = 1.0f; return; }
        *sin_out = sincos_poly(x, x2, false, 0);
        *cos_out = sincos_poly(x, x2, false, 1);
        return;
    }
    const double r = __dmul_rn(x, 0x1.45F306DC9C883p+23);
    const int n = ((int)r + 0x800000) >> 24;
    x = __dsub_rn(x, __dmul_rn((double)n, 0x1.921FB54442D18p0));
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const bool neg = (n & 2) != 0;
    const double xs = __dmul_rn(x, sign), x2 = __dmul_rn(x, x);
    *sin_out = sincos_poly(xs, x2, neg, n);
    *cos_out = sincos_poly(xs, x2, neg, n ^ 1);
}

}  // namespace tc2li
