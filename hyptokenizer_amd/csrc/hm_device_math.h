// hm_device_math.h -- canonical fp32 transcendental functions for the gfx950 kernels.
//
// DESIGN.md "Canonical arithmetic": every distance / midpoint the engine produces is a fixed
// sequence of IEEE fp32 operations (add, mul, div, sqrt, explicit fma) so that results do not
// depend on a vendor math library.  log1p / expm1 follow the published fdlibm float algorithms
// (FreeBSD msun s_log1pf.c / s_expm1f.c; cosh / sinh as in e_coshf.c / e_sinhf.c with
// exp(x) = expm1(x) + 1).  The translation unit is compiled with -ffp-contract=off; fused
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

// log(1 + x), x >= 0.
__device__ __forceinline__ float log1p_c(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    const uint32_t ix = fbits(x);
    if (ix >= 0x7f800000u) {
        if (ix == 0x7f800000u || (ix & 0x7fffffffu) > 0x7f800000u) return x;   // +inf, NaN
        if (ix == 0x80000000u) return x;                                        // -0
        return bitsf(0x7fc00000u);                                              // negative: NaN
    }
    int k = 1;
    float f = 0.0f, c = 0.0f;
    if (ix < 0x3ed413d0u) {                    // 1 + x < sqrt(2)
        if (ix < 0x33800000u) return x;        // x < 2^-24
        k = 0;
        f = x;
    }
    if (k) {
        const float uf = 1.0f + x;
        uint32_t iu = fbits(uf);
        iu += 0x3f800000u - 0x3f3504f3u;
        k = (int)(iu >> 23) - 0x7f;
        if (k < 25) {
            c = (k >= 2) ? (1.0f - (uf - x)) : (x - (uf - 1.0f));
            c = c / uf;
        } else {
            c = 0.0f;
        }
        iu = (iu & 0x007fffffu) + 0x3f3504f3u;
        f = bitsf(iu) - 1.0f;
    }
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (Lg2 + w * Lg4);
    const float t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    return ((s * (hfsq + R) + (dk * ln2_lo + c)) - hfsq + f) + dk * ln2_hi;
}

// acosh(a), a >= 1 (NaN propagates).
__device__ __forceinline__ float acosh_c(float a)
{
    if (a != a) return a;
    if (a > 1.0e9f) return log1p_c(a + a);
    const float t = a - 1.0f;
    const float q = t * (t + 2.0f);
    const float y = t + __builtin_sqrtf(q);
    return log1p_c(y);
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
