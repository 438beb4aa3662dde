// hm_rows_device.h -- wave-cooperative midpoint / merge (device code shared by hm_rows.hip, hm_search.hip and
// hm_loops.hip).  Reference arithmetic: tokenizer/hyperbolic_merge.py:326-340 (log_map -> scale -> exp_map ->
// project), embedding/lorentz_model.py:41-56,73-119.
#pragma once
#include "hm_common.h"

// LDS scratch of one wave: the two operand rows, the scaled tangent and the result, in the reference's column
// order (column 0 = time).
struct MidScratch { float sx[HM_MAX_D1], sy[HM_MAX_D1], sv[HM_MAX_D1], so[HM_MAX_D1 + 36]; };

__device__ __forceinline__ void hm_wave_lds_sync()
{
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// stage image rows ri, rj into ms.sx / ms.sy (coalesced)
__device__ __forceinline__ void hm_wave_stage_rows(const float* img, int RS, int d, int64_t ri, int64_t rj, MidScratch& ms, int lane)
{
    for (int k = lane; k <= d; k += 64) {
        ms.sx[k] = k == 0 ? hm_img_time(img, RS, ri) : hm_img_spatial(img, RS, ri, k - 1);
        ms.sy[k] = k == 0 ? hm_img_time(img, RS, rj) : hm_img_spatial(img, RS, rj, k - 1);
    }
    hm_wave_lds_sync();
}

// One wave: ms.so[0..d] = exp_map(x, w * log_map(x, y)), projected onto the hyperboloid when `project`.
// Same operations in the same order as the one-lane form (oracle hmo_midpoint):
// element-wise steps spread over the lanes, the two torch-order reductions by hm_halfwave_sum (both half-waves
// compute them redundantly), the scalar transcendental steps on every lane (uniform).  Returns the squared
// spatial norm of the result (fmaf chain; only meaningful when `project`).
__device__ __forceinline__ float hm_wave_midpoint(int d, float w, float c, int sign_mode, MidScratch& ms, bool project, int lane)
{
    // log_map (embedding/lorentz_model.py:96-119)
    const float S = hm_halfwave_sum(d, lane, [&](int e) { return ms.sx[1 + e] * ms.sy[1 + e]; });
    const float t0 = ms.sx[0] * ms.sy[0];
    const float mref = t0 - S;
    const float u = sign_mode ? mref : -mref;
    const float m = -u;
    const float a = hm::clamp_min_one(u);
    float coef = hm::acosh_c(a) / __builtin_sqrtf(a * a - 1.0f);
    if (coef == coef && coef > 1.0e4f) coef = 1.0e4f;
    for (int k = lane; k <= d; k += 64) ms.sv[k] = (coef * (ms.sy[k] + m * ms.sx[k])) * w;     // w * log_map
    hm_wave_lds_sync();
    // exp_map (:73-93)
    float n2 = hm_halfwave_sum(d, lane, [&](int e) { return ms.sv[1 + e] * ms.sv[1 + e]; });
    if (n2 == n2 && n2 < 1.0e-8f) n2 = 1.0e-8f;
    const float nn = __builtin_sqrtf(n2);
    float ch, sh;
    hm::cosh_sinh_c(nn, ch, sh);                               // cosh_c / sinh_c with their common expm1 evaluated once
    for (int k = lane; k <= d; k += 64) ms.so[k] = ch * ms.sx[k] + sh * (ms.sv[k] / nn);
    if (lane < 32) ms.so[d + 1 + lane] = 0.0f;                 // zero tail: the chain below runs in steps of 32
    hm_wave_lds_sync();
    float r2 = 0.0f;
    if (project) {
        // project (:41-56): the canonical order is a sequential fmaf chain over k = 1..d; every lane runs it on LDS
        // broadcast reads, 32 operands fetched per step so that the chain waits for LDS d / 32 times, not d / 8 times
        // (fmaf(0, 0, r2) == r2: the zero tail does not change it)
#pragma unroll 1
        for (int k0 = 1; k0 <= d; k0 += 32) {
            float v[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = ms.so[k0 + q];
#pragma unroll
            for (int q = 0; q < 32; ++q) r2 = __builtin_fmaf(v[q], v[q], r2);
        }
        const float rr = __builtin_sqrtf(r2);
        const float x0 = __builtin_sqrtf(1.0f + (c * rr) * rr);
        hm_wave_lds_sync();
        if (lane == 0) ms.so[0] = x0;
        hm_wave_lds_sync();
    }
    return r2;
}

// write ms.so as row `row` of the caller's table (may be nullptr), of the fp32 image and of the bf16 image, and
// fold its norms into the bounds the pair scan's error margin uses
__device__ __forceinline__ void hm_wave_store_row(MidScratch& ms, float r2, int d, int RS, int KC, float* __restrict__ X, int64_t ld,
                                                  float* __restrict__ img, unsigned char* __restrict__ img16, int64_t row,
                                                  uint32_t* __restrict__ rmax2_bits, int lane)
{
    if (lane == 0) {
        const float x0 = ms.so[0];
        const float q2 = __builtin_fmaf(x0, x0, r2);
        if (q2 < 3.0e38f && q2 > 0.0f) {
            atomicMax(rmax2_bits, hm::fbits(q2 * 1.0001f));
            if (r2 > 0.0f) atomicMax(rmax2_bits + 1, hm::fbits(r2 * 1.0001f));
        }
    }
    float* ir = img + row * RS;
    for (int k = lane; k <= d; k += 64) {
        const float v = ms.so[k];
        if (X != nullptr) X[row * ld + k] = v;
        if (k == 0) ir[RS - 4] = v;
        else ir[hm_img_off(k - 1)] = v;
    }
    const int CH = hm_row16_chunks(KC);
    for (int cidx = lane; cidx < CH; cidx += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cidx >= KC) v.x = hm::fbits(ms.so[0]);
        else v = hm_bf16_chunk(ms.so + 1, ms.so[0], d, KC, cidx);
        *reinterpret_cast<uint4*>(img16 + ((int64_t)row * CH + cidx) * 16) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// row tiles: one fixed row against 64 CONSECUTIVE image rows per wave
// ------------------------------------------------------------------------------------------------
// The image rows of a tile are one contiguous block of 64 * RS floats: the wave copies it to LDS with fully coalesced
// 16-byte loads (RS / 4 of them per lane, all in flight together -- and issued BEFORE the fixed row is known, where
// the caller has something else to do first), then lane l evaluates the canonical u of row r0 + l on its own, in
// torch's reduction order (hm::torch_order_sum), out of LDS: RS / 4 is odd (hm_row_floats), so lane-private reads at
// stride RS hit 64 different banks.  Against a half-wave per row this is a sixth of the memory instructions, no
// shuffles, and the transcendental tail runs once per row on 64 rows at a time.
#define HM_TILE_ROWS 64
#define HM_TILE_MAXQ 33            // RS / 4 at d = 128

struct TileRegs { uint4 q[HM_TILE_MAXQ]; };

__device__ __forceinline__ void hm_tile_load(const float* __restrict__ img, int RS, int64_t r0, int64_t n, TileRegs& tr, int lane)
{
    const int64_t rows = n - r0 < HM_TILE_ROWS ? n - r0 : HM_TILE_ROWS;
    const int total = rows > 0 ? (int)rows * (RS >> 2) : 0;
    const uint4* src = reinterpret_cast<const uint4*>(img + r0 * RS);
#pragma unroll
    for (int i = 0; i < HM_TILE_MAXQ; ++i) {
        const int j = lane + 64 * i;
        tr.q[i] = j < total ? src[j] : make_uint4(0, 0, 0, 0);
    }
}

__device__ __forceinline__ void hm_tile_store(float* tile, int RS, const TileRegs& tr, int lane)
{
    uint4* dst = reinterpret_cast<uint4*>(tile);
    const int total = HM_TILE_ROWS * (RS >> 2);
#pragma unroll
    for (int i = 0; i < HM_TILE_MAXQ; ++i) {
        const int j = lane + 64 * i;
        if (j < total) dst[j] = tr.q[i];
    }
}

// The fixed row in IMAGE layout (RS floats: groups [s0, s2, s1, s3], time group last, zero padding) from the reference
// column order (xo[0] = time, xo[1 + e]); one wave, no synchronisation inside.
__device__ __forceinline__ void hm_tile_fixed_row(const float* xo, int RS, int d, float* ximg, int lane)
{
    for (int p = lane; p < RS; p += 64) ximg[p] = 0.0f;
    hm_wave_lds_sync();
    for (int e = lane; e < d; e += 64) ximg[hm_img_off(e)] = xo[1 + e];
    if (lane == 0) ximg[RS - 4] = xo[0];
    hm_wave_lds_sync();
}

// canonical u between tile row `lane` and the fixed row ximg (image layout, as the tile rows).  Both are read as
// 16-byte groups (the tile at stride RS: RS / 4 is odd, conflict-free for ds_read_b128; the fixed row is a broadcast);
// a group holds elements (4g, 4g + 2, 4g + 1, 4g + 3), and element e = 32 i + c feeds chain c of torch's 32 -- so the
// eight groups of one pass over the chains are fully unrolled with fixed accumulators.  Same sums in the same order as
// hm::torch_order_sum (rows shorter than 8 take that path).
__device__ __forceinline__ float hm_tile_u(const float* tile, int RS, int d, const float* ximg, int sign_mode, int lane)
{
    const float* r = tile + lane * RS;
    float S;
    if (d < 8) {
        S = hm::torch_order_sum([&](int e) { return r[hm_img_off(e)] * ximg[hm_img_off(e)]; }, d);
    } else {
        const float4* r4 = reinterpret_cast<const float4*>(r);
        const float4* x4 = reinterpret_cast<const float4*>(ximg);
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.0f;
        const int vec = d >> 3, ilp = vec >> 2;
        for (int i = 0; i < ilp; ++i) {
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
                const float4 R = r4[8 * i + gg], X = x4[8 * i + gg];
                acc[4 * gg + 0] = acc[4 * gg + 0] + R.x * X.x;
                acc[4 * gg + 2] = acc[4 * gg + 2] + R.y * X.y;
                acc[4 * gg + 1] = acc[4 * gg + 1] + R.z * X.z;
                acc[4 * gg + 3] = acc[4 * gg + 3] + R.w * X.w;
            }
        }
        // leftover vectors of 8 (two groups each): all into accumulator 0, i.e. chains 0..7
        for (int q = ilp * 4; q < vec; ++q) {
            const float4 R0 = r4[2 * q], X0 = x4[2 * q], R1 = r4[2 * q + 1], X1 = x4[2 * q + 1];
            acc[0] = acc[0] + R0.x * X0.x; acc[2] = acc[2] + R0.y * X0.y; acc[1] = acc[1] + R0.z * X0.z; acc[3] = acc[3] + R0.w * X0.w;
            acc[4] = acc[4] + R1.x * X1.x; acc[6] = acc[6] + R1.y * X1.y; acc[5] = acc[5] + R1.z * X1.z; acc[7] = acc[7] + R1.w * X1.w;
        }
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[l] = acc[l] + acc[8 * k + l];
        // scalar tail: elements 8 vec .. d - 1 in element order, summed from zero
        float tail = 0.0f;
        for (int e = vec * 8; e < d; ++e) tail = tail + r[hm_img_off(e)] * ximg[hm_img_off(e)];
        S = tail;
#pragma unroll
        for (int l = 0; l < 8; ++l) S = S + acc[l];
    }
    const float t = r[RS - 4] * ximg[RS - 4];
    const float m = t - S;
    return sign_mode ? m : -m;
}
