// hm_device_math.h -- canonical fp32 transcendental functions for the gfx950 kernels.
//
// DESIGN.md "Canonical arithmetic": every distance / midpoint the engine produces is a fixed
// sequence of IEEE operations (add, mul, div, sqrt, explicit fma) so that results do not depend
// on a vendor math library and reproduce the fp32 bits of the reference's torch CPU path:
// torch_order_sum = ATen SumKernel.cpp reduction order, log1p = fdlibm s_log1pf.c (7-coefficient
// form), acosh structured as glibc e_acoshf.c (double-rounded-once above 2), expm1 = fdlibm
// s_expm1f.c, cosh / sinh as in e_coshf.c / e_sinhf.c with exp(x) = expm1(x) + 1.  The translation unit is compiled with -ffp-contract=off; fused
// multiply-adds appear only where written as __builtin_fmaf.  Division and sqrt are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Reference call sites these serve: torch.acosh in distance / log_map
// (embedding/lorentz_model.py:112,138,178), torch.cosh / torch.sinh in exp_map (:93).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace hm {

__device__ __forceinline__ uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float rsqrt_free_sqrt(float x) { return __builtin_sqrtf(x); }

// log(1 + x), x >= 0: fdlibm s_log1pf.c (7-coefficient form, as shipped by glibc 2.35).
__device__ __forceinline__ float log1p_c(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    int32_t hx = (int32_t)fbits(x), hu = 0, k = 1;
    float f = 0.0f, c = 0.0f, u;
    if (hx < 0) {
        if (fbits(x) == 0x80000000u) return x;
        return bitsf(0x7fc00000u);
    }
    if (hx >= 0x7f800000) return x + x;
    if (hx < 0x3ed413d0) {
        if (hx < 0x38000000) {
            if (hx < 0x33800000) return x;
            return x - (x * x) * 0.5f;
        }
        k = 0; f = x; hu = 1;
    }
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)fbits(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
            c = c / u;
        } else {
            u = x;
            hu = (int32_t)fbits(u);
            k = (hu >> 23) - 127;
            c = 0.0f;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f4) {
            u = bitsf((uint32_t)(hu | 0x3f800000));
        } else {
            k += 1;
            u = bitsf((uint32_t)(hu | 0x3f000000));
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    const float hfsq = (0.5f * f) * f;
    if (hu == 0) {
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c = c + (float)k * ln2_lo;
            return (float)k * ln2_hi + c;
        }
        const float R0 = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R0;
        return (float)k * ln2_hi - ((R0 - ((float)k * ln2_lo + c)) - f);
    }
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return (float)k * ln2_hi - ((hfsq - (s * (hfsq + R) + ((float)k * ln2_lo + c))) - f);
}

// log(1 + x) in double, x >= 0: fdlibm s_log1p.c.  Only for acosh(a), a > 2.
__device__ __forceinline__ double log1p_d(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ui = __builtin_bit_cast(uint64_t, x);
    uint32_t hx = (uint32_t)(ui >> 32), hu;
    int k = 1;
    double f = 0.0, c = 0.0, uf;
    if (hx >= 0x7ff00000u) return x;
    if (hx < 0x3fda827au) {
        if ((hx << 1) < (0x3ca00000u << 1)) return x;
        k = 0; c = 0.0; f = x;
    }
    if (k) {
        uf = 1.0 + x;
        ui = __builtin_bit_cast(uint64_t, uf);
        hu = (uint32_t)(ui >> 32);
        hu += 0x3ff00000u - 0x3fe6a09eu;
        k = (int)(hu >> 20) - 0x3ff;
        if (k < 54) { c = (k >= 2) ? 1.0 - (uf - x) : x - (uf - 1.0); c = c / uf; } else c = 0.0;
        hu = (hu & 0x000fffffu) + 0x3fe6a09eu;
        ui = ((uint64_t)hu << 32) | (ui & 0xffffffffull);
        uf = __builtin_bit_cast(double, ui);
        f = uf - 1.0;
    }
    const double hfsq = (0.5 * f) * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return ((s * (hfsq + R) + (dk * ln2_lo + c)) - hfsq + f) + dk * ln2_hi;
}

// acosh(a), a >= 1 (NaN propagates), structured as glibc's e_acoshf.c (what torch.acosh calls).
__device__ __forceinline__ float acosh_c(float a)
{
    if (a != a) return a;
    if (a <= 1.0f) return 0.0f;
    if (a <= 2.0f) {
        const float t = a - 1.0f;
        return log1p_c(t + __builtin_sqrtf(2.0f * t + t * t));
    }
    if (a > 3.0e38f) return a;
    const double x = (double)a;
    const double zz = 2.0 * x - 1.0 / (x + __builtin_sqrt(x * x - 1.0));
    return (float)log1p_d(zz - 1.0);
}

// Sum of n fp32 terms in the order of ATen's SumKernel.cpp inner reduction (float, 8-lane vectors,
// 4 interleaved accumulators; scalar 4-accumulator form below 8 terms).  term(k) -> k-th term.
template <class TERM>
__device__ __forceinline__ float torch_order_sum(TERM term, int n)
{
    if (n < 8) {
        float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int size_ilp = n / 4;
        for (int i = 0; i < size_ilp; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ps[k] = ps[k] + term(i * 4 + k);
        }
        for (int i = size_ilp * 4; i < n; ++i) ps[0] = ps[0] + term(i);
        ps[0] = ps[0] + ps[1];
        ps[0] = ps[0] + ps[2];
        ps[0] = ps[0] + ps[3];
        return ps[0];
    }
    float ps[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 8; ++l) ps[k][l] = 0.0f;
    const int vec_size = n / 8, size_ilp = vec_size / 4;
    for (int i = 0; i < size_ilp; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < 8; ++l) ps[k][l] = ps[k][l] + term((i * 4 + k) * 8 + l);
    }
    for (int i = size_ilp * 4; i < vec_size; ++i) {
#pragma unroll
        for (int l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + term(i * 8 + l);
    }
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + ps[k][l];
    float acc = 0.0f;
    for (int i = vec_size * 8; i < n; ++i) acc = acc + term(i);
#pragma unroll
    for (int l = 0; l < 8; ++l) acc = acc + ps[0][l];
    return acc;
}

// exp(x) - 1.
__device__ __forceinline__ float expm1_c(float x)
{
    const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                invln2 = 1.4426950216e+00f, Q1 = -3.3333212137e-2f, Q2 = 1.5807170421e-3f;
    uint32_t hx = fbits(x);
    const int sign = (int)(hx >> 31);
    int k;
    float hi, lo, c = 0.0f, t, e, y;
    hx &= 0x7fffffffu;
    if (hx >= 0x4195b844u) {
        if (hx > 0x7f800000u) return x;
        if (sign) return -1.0f;
        if (x > o_threshold) return x * 0x1p127f;
    }
    if (hx > 0x3eb17218u) {
        if (hx < 0x3F851592u) {
            if (!sign) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else       { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = (int)(invln2 * x + (sign ? -0.5f : 0.5f));
            t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {
        return x;
    } else {
        k = 0;
    }
    const float hfx = 0.5f * x;
    const float hxs = x * hfx;
    const float r1 = 1.0f + hxs * (Q1 + hxs * Q2);
    t = 3.0f - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0f - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = x * (e - c) - c;
    e = e - hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return 1.0f + 2.0f * (x - e);
    }
    const float twopk = bitsf((uint32_t)(0x7f + k) << 23);
    if (k < 0 || k > 56) {
        y = x - e + 1.0f;
        if (k == 128) y = y * 2.0f * 0x1p127f; else y = y * twopk;
        return y - 1.0f;
    }
    const float tk = bitsf((uint32_t)(0x7f - k) << 23);
    if (k < 23) y = (x - e + (1.0f - tk)) * twopk;
    else        y = (x - e - tk + 1.0f) * twopk;
    return y;
}

__device__ __forceinline__ float cosh_c(float x)
{
    if (x != x) return x;
    if (x < 0.0f) x = -x;
    if (x < 0.34657359f) {
        const float t = expm1_c(x);
        const float w = 1.0f + t;
        if (x < 0.000244140625f) return 1.0f;
        return 1.0f + (t * t) / (w + w);
    }
    const float t = expm1_c(x) + 1.0f;
    return 0.5f * t + 0.5f / t;
}

__device__ __forceinline__ float sinh_c(float x)
{
    float h = 0.5f;
    if (x != x) return x;
    if (x < 0.0f) { h = -0.5f; x = -x; }
    const float t = expm1_c(x);
    if (x < 1.0f) return h * (2.0f * t - (t * t) / (t + 1.0f));
    return h * (t + t / (t + 1.0f));
}

// cosh_c(x) and sinh_c(x) for x >= 0 (or NaN) with the one expm1 both are built on evaluated once: the same
// operations on the same values as the two functions above, so the same bits.
__device__ __forceinline__ void cosh_sinh_c(float x, float& ch, float& sh)
{
    if (x != x || x < 0.0f) { ch = cosh_c(x); sh = sinh_c(x); return; }
    const float t = expm1_c(x);
    if (x < 0.34657359f) {
        const float w = 1.0f + t;
        ch = (x < 0.000244140625f) ? 1.0f : 1.0f + (t * t) / (w + w);
    } else {
        const float e = t + 1.0f;
        ch = 0.5f * e + 0.5f / e;
    }
    if (x < 1.0f) sh = 0.5f * (2.0f * t - (t * t) / (t + 1.0f));
    else sh = 0.5f * (t + t / (t + 1.0f));
}

// torch.clamp(u, min = 1 + 1e-8) in fp32 (the bound is exactly 1.0f); NaN propagates.
__device__ __forceinline__ float clamp_min_one(float u)
{
    if (u != u) return u;
    return (u < 1.0f) ? 1.0f : u;
}

// distance from the clamped or unclamped argument of acosh.
__device__ __forceinline__ float dist_from_u(float u, float sqrt_c)
{
    return acosh_c(clamp_min_one(u)) / sqrt_c;
}

}  // namespace hm
