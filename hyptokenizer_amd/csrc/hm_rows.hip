// hm_rows.hip -- image construction, merge / midpoint, gathered and row-wise kernels, table re-projection and
// the coherence kernel of the enhanced tokenizer, with their C-ABI entry points (include/hypmerge.h).
//
// All of these are bandwidth / latency kernels: rows are read coalesced (a half-wave reads
// one 128-byte segment per instruction) and every canonical reduction is done wave-cooperatively
// (hm_halfwave_sum: ATen's 32 chains, one per lane).
#include "hm_common.h"
#include "hm_rows_device.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// image construction
// ------------------------------------------------------------------------------------------------
__global__ void hm_build_image_kernel(const float* __restrict__ X, int64_t ld, int d, int NG, float* __restrict__ img,
                                      int64_t row_begin, int64_t row_end)
{
    const int RS = hm_row_floats(NG);
    const int64_t total = (row_end - row_begin) * RS;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row_begin + t / RS;
        const int p = (int)(t % RS);
        const int g = p >> 2, q = p & 3;
        float v = 0.0f;
        if (g == RS / 4 - 1) {
            if (q == 0) v = X[row * ld];
        } else if (g < NG) {
            const int s = 4 * g + (((q & 1) << 1) | (q >> 1));   // inverse of hm_pos_in_group
            if (s < d) v = X[row * ld + 1 + s];
        }
        img[row * RS + p] = v;
    }
}

// largest squared row norm [0] and largest squared spatial norm [1] of the live rows (finite rows
// only), kept as float bits for atomicMax.  They scale the bound |u_f - u_c| used by the pair scan.
__global__ void hm_rownorm_kernel(const float* __restrict__ img, int RS, int64_t row_begin, int64_t row_end,
                                  uint32_t* __restrict__ rmax2_bits)
{
    const int64_t row = row_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float r2 = 0.0f, s2 = 0.0f;
    if (row < row_end) {
        const float* rr = img + row * RS;
        for (int k = 0; k < RS - 4; ++k) s2 = __builtin_fmaf(rr[k], rr[k], s2);
        r2 = __builtin_fmaf(rr[RS - 4], rr[RS - 4], s2);
    }
    if (!(r2 < 3.0e38f)) { r2 = 0.0f; s2 = 0.0f; }      // NaN / inf rows never form candidates
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r2 = __builtin_fmaxf(r2, __shfl_xor(r2, off, 64));
        s2 = __builtin_fmaxf(s2, __shfl_xor(s2, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (r2 > 0.0f) hm_raise_bits(rmax2_bits, hm::fbits(r2));
        if (s2 > 0.0f) hm_raise_bits(rmax2_bits + 1, hm::fbits(s2));
    }
}

// bf16 image row: KC chunks of 8 K-slots as bf16 (round to nearest even), followed -- when KC is even -- by one
// 16-byte chunk [x0 as fp32, 0, 0, 0] (chunks per row: always odd, so the ds_read_b128 fragment reads of
// 32 consecutive rows fall on distinct 16-byte bank slots).
__global__ void hm_build_image16_kernel(const float* __restrict__ X, int64_t ld, int d, int KC, unsigned char* __restrict__ img16,
                                        int64_t row_begin, int64_t row_end)
{
    const int CH = hm_row16_chunks(KC);               // 16-byte chunks per row
    const int64_t total = (row_end - row_begin) * CH;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row_begin + t / CH;
        const int c = (int)(t % CH);
        uint4 v = make_uint4(0, 0, 0, 0);
        const float* xr = X + row * ld;
        if (c >= KC) v.x = hm::fbits(xr[0]);
        else v = hm_bf16_chunk(xr + 1, xr[0], d, KC, c);
        *reinterpret_cast<uint4*>(img16 + (row * CH + c) * 16) = v;
    }
}

int hm_build_rows(hm_engine* e, const float* X, int64_t ld, int64_t r0, int64_t r1, hipStream_t s)
{
    if (r1 <= r0) return HM_OK;
    const int64_t total = (r1 - r0) * e->RS;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(hm_build_image_kernel, dim3(blocks), dim3(256), 0, s, X, ld, e->d, e->NG, e->img, r0, r1);
    HM_HIP(hipGetLastError());
    hipLaunchKernelGGL(hm_rownorm_kernel, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, s, e->img, e->RS, r0, r1, e->d_rmax2);
    HM_HIP(hipGetLastError());
    const int64_t total16 = (r1 - r0) * hm_row16_chunks(e->KC);
    hipLaunchKernelGGL(hm_build_image16_kernel, dim3((unsigned)std::min<int64_t>((total16 + 255) / 256, 4096)), dim3(256), 0, s, X, ld,
                       e->d, e->KC, e->img16, r0, r1);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// gathered / one-vs-all distances on the image (half-wave per output, coalesced row reads)
// ------------------------------------------------------------------------------------------------
// (each half-wave takes HM_GATHER consecutive outputs per round: hm_halfwave_gather)
__global__ __launch_bounds__(256) void hm_pairdist_kernel(const float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                                          const int32_t* __restrict__ J, int64_t b, float sqrt_c, int sign_mode,
                                                          float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, t = lane & 31;
    const int64_t nhw = ((int64_t)gridDim.x * blockDim.x) >> 5;
    const int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    for (int64_t base = (hw & ~(int64_t)1) * HM_GATHER; base < b; base += nhw * HM_GATHER) {      // wave-uniform trip count
        const int64_t mybase = base + (hw & 1) * HM_GATHER;
        const int64_t mine_idx = mybase + t < b ? mybase + t : b - 1;
        const int32_t my_i = I[mine_idx], my_j = J[mine_idx];
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const int32_t ri = __shfl(my_i, (lane & 32) + k, 64), rj = __shfl(my_j, (lane & 32) + k, 64);
            return hm_img_u_halfwave(img, RS, d, ri, rj, sign_mode, lane);
        });
        if (t < HM_GATHER && mybase + t < b) out[mybase + t] = hm::dist_from_u(u, sqrt_c);
    }
}

#ifndef HM_ROWVS_WAVES
#define HM_ROWVS_WAVES 2
#endif
#ifndef HM_ROWVS_GRID_CAP
#define HM_ROWVS_GRID_CAP 512      // = the blocks resident at once (two per CU): beyond one round a wave takes several tiles, the next one prefetched
#endif
__global__ __launch_bounds__(64 * HM_ROWVS_WAVES) void hm_rowvsall_kernel(const float* __restrict__ img, int RS, int d, int64_t row, int64_t n,
                                                                          float sqrt_c, int sign_mode, float* __restrict__ out)
{
    extern __shared__ __align__(16) float lds[];
    float* xs = lds;                                           // the fixed row: a copy of its image row (same layout as the tiles)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = lds + HM_MAX_D1 + 4 + wv * HM_TILE_ROWS * RS;   // (HM_MAX_D1 + 4) % 4 == 0 and RS <= HM_MAX_D1 + 4: tiles stay 16-byte aligned
    const int64_t nt = (n + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
    int64_t tl = (int64_t)blockIdx.x * HM_ROWVS_WAVES + wv;
    TileRegs tr;
    if (tl < nt) hm_tile_load(img, RS, tl * HM_TILE_ROWS, n, tr, lane);
    for (int k = threadIdx.x; k < RS; k += blockDim.x) xs[k] = img[row * RS + k];
    __syncthreads();
    for (; tl < nt; tl += (int64_t)gridDim.x * HM_ROWVS_WAVES) {
        hm_tile_store(tile, RS, tr, lane);
        hm_wave_lds_sync();
        const int64_t nxt = tl + (int64_t)gridDim.x * HM_ROWVS_WAVES;
        if (nxt < nt) hm_tile_load(img, RS, nxt * HM_TILE_ROWS, n, tr, lane);      // in flight while this tile is evaluated
        const float u = hm_tile_u(tile, RS, d, xs, sign_mode, lane);
        const int64_t r = tl * HM_TILE_ROWS + lane;
        if (r < n) out[r] = hm::dist_from_u(u, sqrt_c);
        hm_wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// midpoint / merge (one wave per pair)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hm_midpoint_kernel(const float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                                          const int32_t* __restrict__ J, const float* __restrict__ W, int64_t b, float c,
                                                          int sign_mode, float* __restrict__ out)
{
    __shared__ MidScratch ms[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wv;
    if (t >= b) return;                                       // wave-uniform
    hm_wave_stage_rows(img, RS, d, I[t], J[t], ms[wv], lane);
    hm_wave_midpoint(d, W[t], c, sign_mode, ms[wv], true, lane);
    for (int k = lane; k <= d; k += 64) out[t * (d + 1) + k] = ms[wv].so[k];
}

// fused merge: midpoint of image rows (i, j) -> table row and image rows `new_row` (hyperbolic_merge.py:326-351)
__global__ __launch_bounds__(64) void hm_merge_append_kernel(float* __restrict__ img, int RS, int d, int32_t i, int32_t j, float w, float c,
                                                             int sign_mode, float* __restrict__ X, int64_t ld, int64_t new_row,
                                                             uint32_t* __restrict__ rmax2_bits, unsigned char* __restrict__ img16, int KC)
{
    __shared__ MidScratch ms;
    const int lane = threadIdx.x;
    hm_wave_stage_rows(img, RS, d, i, j, ms, lane);
    const float r2 = hm_wave_midpoint(d, w, c, sign_mode, ms, true, lane);
    hm_wave_store_row(ms, r2, d, RS, KC, X, ld, img, img16, new_row, rmax2_bits, lane);
}

// several merges known in advance (the fast tokenizer knows every merge between two refreshes when the refresh
// returns): merge t reads rows (I[t], J[t]) -- which may be rows written by earlier merges of the same batch --
// and writes row first_row + t.  One wave, sequential: the chain is a true dependency.
__global__ __launch_bounds__(64) void hm_merge_batch_kernel(float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                                            const int32_t* __restrict__ J, const float* __restrict__ W, int count, float c,
                                                            int sign_mode, float* __restrict__ X, int64_t ld, int64_t first_row,
                                                            uint32_t* __restrict__ rmax2_bits, unsigned char* __restrict__ img16, int KC)
{
    __shared__ MidScratch ms;
    const int lane = threadIdx.x;
    for (int t = 0; t < count; ++t) {
        hm_wave_stage_rows(img, RS, d, I[t], J[t], ms, lane);
        const float r2 = hm_wave_midpoint(d, W[t], c, sign_mode, ms, true, lane);
        hm_wave_store_row(ms, r2, d, RS, KC, X, ld, img, img16, first_row + t, rmax2_bits, lane);
        __threadfence();                                      // the next merge of the batch may read this row
        hm_wave_lds_sync();
    }
}

// the same batch when no merge reads a row the batch writes (every operand row < first_row -- the fast tokenizer's case:
// cached candidates only name rows that existed at the refresh): one wave per merge, all at once
__global__ __launch_bounds__(256) void hm_merge_rows_kernel(float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                                            const int32_t* __restrict__ J, const float* __restrict__ W, int count, float c,
                                                            int sign_mode, float* __restrict__ X, int64_t ld, int64_t first_row,
                                                            uint32_t* __restrict__ rmax2_bits, unsigned char* __restrict__ img16, int KC)
{
    __shared__ MidScratch ms[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wv;
    if (t >= count) return;
    hm_wave_stage_rows(img, RS, d, I[t], J[t], ms[wv], lane);
    const float r2 = hm_wave_midpoint(d, W[t], c, sign_mode, ms[wv], true, lane);
    hm_wave_store_row(ms[wv], r2, d, RS, KC, X, ld, img, img16, first_row + t, rmax2_bits, lane);
}

// ------------------------------------------------------------------------------------------------
// enhanced tokenizer (BASELINE config 5)
// ------------------------------------------------------------------------------------------------
// tokenizer/enhanced_fast_hyperbolic_merge.py:308-333 (_compute_semantic_coherence), one wave per candidate:
// m = exp_map(x_i, w * log_map(x_i, x_j)) -- not projected -- kept in LDS, then distance(m, x_s) for the
// candidate's ns sampled rows, a half-wave per sample.  out[t * ns + s].
__global__ __launch_bounds__(256) void hm_coherence_kernel(const float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                                           const int32_t* __restrict__ J, const float* __restrict__ W,
                                                           const int32_t* __restrict__ S, int64_t b, int ns, float c, float sqrt_c,
                                                           int sign_mode, float* __restrict__ out)
{
    __shared__ MidScratch ms[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5;
    const int64_t t0 = (int64_t)blockIdx.x * 4 + wv;
    if (t0 >= b) return;
    MidScratch& m = ms[wv];
    hm_wave_stage_rows(img, RS, d, I[t0], J[t0], m, lane);
    hm_wave_midpoint(d, W[t0], c, sign_mode, m, false, lane);
    const float m0 = m.so[0];
    const int t = lane & 31;
    for (int s0 = 0; s0 < ns; s0 += 2 * HM_GATHER) {          // a round: half-wave h takes HM_GATHER samples from s0 + HM_GATHER * h
        const int sb = s0 + HM_GATHER * h;
        const int my_s = sb + t < ns ? sb + t : ns - 1;
        const int32_t my_row = S[t0 * ns + my_s];
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const float* row = img + (int64_t)__shfl(my_row, (lane & 32) + k, 64) * RS;
            const float Ssum = hm_halfwave_sum(d, lane, [&](int e) { return m.so[1 + e] * row[hm_img_off(e)]; });
            const float tp = m0 * row[RS - 4];
            const float mm = tp - Ssum;
            return sign_mode ? mm : -mm;
        });
        if (t < HM_GATHER && sb + t < ns) out[t0 * ns + sb + t] = hm::dist_from_u(u, sqrt_c);
    }
}

// project_to_hyperboloid over rows [0, n_rows) of the caller's table in place (enhanced...:784-792): only
// column 0 changes.  Rows below n_live also refresh the time slots of both images and the norm bounds.
// 64 rows per block are staged through LDS with coalesced loads; a thread then runs its row's fmaf chain
// (the canonical order of project is sequential) on LDS operands.
__global__ __launch_bounds__(64) void hm_project_table_kernel(float* __restrict__ X, int64_t ld, int d, int64_t n_rows, float c,
                                                              float* __restrict__ img, int RS, unsigned char* __restrict__ img16, int KC,
                                                              int64_t n_live, uint32_t* __restrict__ rmax2_bits,
                                                              uint32_t* __restrict__ retired_bits, ArgminSeed* __restrict__ seed)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {                // nothing on this stream reads them before the kernel ends
        retired_bits[0] = 0u; retired_bits[1] = 0u;
        ArgminSeed z; z.key = 0ull; z.i = 0u; z.valid = 0u;
        *seed = z;
    }
    extern __shared__ __align__(16) float tile[];             // 64 x stride
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)(n_rows - r0 < 64 ? n_rows - r0 : 64);
    const int lane = threadIdx.x;
    int stride, first;
    if (ld == d + 1 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0) {
        // the reference's layout: rows back to back -> one flat copy of 64 * ld floats (a multiple of 16 bytes from a 16-byte
        // aligned start: r0 is a multiple of 64), as 16-byte loads that are ALL in flight before the first one is stored
        stride = (int)ld; first = 1;
        const float* src = X + r0 * ld;
        const int total = rows * (int)ld, nvec = total >> 2;
        const uint4* src4 = reinterpret_cast<const uint4*>(src);
        TileRegs tr;
#pragma unroll
        for (int i = 0; i < HM_TILE_MAXQ; ++i) {
            const int q = lane + 64 * i;
            tr.q[i] = q < nvec ? src4[q] : make_uint4(0, 0, 0, 0);
        }
        float rem = 0.0f;
        if (lane < (total & 3)) rem = src[4 * nvec + lane];
        uint4* dst4 = reinterpret_cast<uint4*>(tile);
#pragma unroll
        for (int i = 0; i < HM_TILE_MAXQ; ++i) {
            const int q = lane + 64 * i;
            if (q < nvec) dst4[q] = tr.q[i];
        }
        if (lane < (total & 3)) tile[4 * nvec + lane] = rem;
    } else if (ld == d + 1) {
        stride = (int)ld; first = 1;
        const float* src = X + r0 * ld;
        const int total = rows * (int)ld;
#pragma unroll 8
        for (int q = lane; q < total; q += 64) tile[q] = src[q];
    } else {
        stride = d | 1; first = 0;                            // spatial part only, odd stride
        for (int q = lane; q < rows * d; q += 64) {
            const int r = q / d, k = q - r * d;
            tile[r * stride + k] = X[(r0 + r) * ld + 1 + k];
        }
    }
    __syncthreads();
    float r2 = 0.0f, x0 = 0.0f;
    if (lane < rows) {
        const float* rowp = tile + lane * stride + first;
        // the canonical order of project is ONE sequential fmaf chain; its operands are fetched sixteen at a time
        int k = 0;
        for (; k + 16 <= d; k += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = rowp[k + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) r2 = __builtin_fmaf(v[q], v[q], r2);
        }
        for (; k < d; ++k) r2 = __builtin_fmaf(rowp[k], rowp[k], r2);
        const float rr = __builtin_sqrtf(r2);
        x0 = __builtin_sqrtf(1.0f + (c * rr) * rr);
        const int64_t row = r0 + lane;
        X[row * ld] = x0;
        if (row < n_live) {
            img[row * RS + RS - 4] = x0;
            const int CH = hm_row16_chunks(KC);
            unsigned char* r16 = img16 + row * CH * 16;
            if (CH > KC) *reinterpret_cast<uint32_t*>(r16 + KC * 16) = hm::fbits(x0);
            if (KC > 0) {
                const __bf16 hb = (__bf16)x0;
                const float hi = (float)hb, lo = x0 - hi;
                uint32_t* slots = reinterpret_cast<uint32_t*>(r16 + (KC - 1) * 16 + 8);       // K-slots 8c+4 .. 8c+7 of the last chunk
                slots[0] = hm_pack_bf16(hi, lo);
                slots[1] = hm_pack_bf16(hi, 0.0f);
            }
        } else {
            r2 = 0.0f; x0 = 0.0f;
        }
    }
    float q2 = __builtin_fmaf(x0, x0, r2);
    if (!(q2 < 3.0e38f)) { q2 = 0.0f; r2 = 0.0f; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        q2 = __builtin_fmaxf(q2, __shfl_xor(q2, off, 64));
        r2 = __builtin_fmaxf(r2, __shfl_xor(r2, off, 64));
    }
    if (lane == 0) {
        if (q2 > 0.0f) hm_raise_bits(rmax2_bits, hm::fbits(q2 * 1.0001f));
        if (r2 > 0.0f) hm_raise_bits(rmax2_bits + 1, hm::fbits(r2 * 1.0001f));
    }
}

// ------------------------------------------------------------------------------------------------
// engine-independent kernels on row-major arrays (embedding/lorentz_model.py function surface)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hm_rm_u(const float* x, const float* y, int d1, int sign_mode)
{
    const float S = hm::torch_order_sum([&](int s) { return x[1 + s] * y[1 + s]; }, d1 - 1);
    const float t = x[0] * y[0];
    const float m = t - S;
    return sign_mode ? m : -m;
}

// half-wave per row pair (coalesced)
__device__ __forceinline__ float hm_rm_u_halfwave(const float* x, const float* y, int d1, int sign_mode, int lane)
{
    const float S = hm_halfwave_sum(d1 - 1, lane, [&](int e) { return x[1 + e] * y[1 + e]; });
    const float t = x[0] * y[0];
    const float m = t - S;
    return sign_mode ? m : -m;
}

// one half-wave per 32 consecutive outputs (row-major enumeration of out[i, j])
__global__ __launch_bounds__(256) void hm_dense_kernel(const float* __restrict__ X, int64_t n1, const float* __restrict__ Y, int64_t n2,
                                                       int64_t ldx, int64_t ldy, int d1, float sqrt_c, int sign_mode,
                                                       float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, t = lane & 31;
    const int64_t total = n1 * n2;
    const int64_t nhw = ((int64_t)gridDim.x * blockDim.x) >> 5;
    const int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    for (int64_t base = (hw & ~(int64_t)1) * HM_GATHER; base < total; base += nhw * HM_GATHER) {
        const int64_t mybase = base + (hw & 1) * HM_GATHER;
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const int64_t o = mybase + k < total ? mybase + k : total - 1;
            const int64_t i = o / n2, j = o - i * n2;
            return hm_rm_u_halfwave(X + i * ldx, Y + j * ldy, d1, sign_mode, lane);
        });
        if (t < HM_GATHER && mybase + t < total) out[mybase + t] = hm::dist_from_u(u, sqrt_c);
    }
}

__global__ void hm_rows_minkowski_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t b, int64_t ld, int d1,
                                         int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    // minkowski_dot under the active convention = -u
    out[t] = -hm_rm_u(x + t * ld, y + t * ld, d1, sign_mode);
}

__global__ __launch_bounds__(256) void hm_rows_distance_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t b, int64_t ld,
                                                               int d1, float sqrt_c, int sign_mode, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, t = lane & 31;
    const int64_t nhw = ((int64_t)gridDim.x * blockDim.x) >> 5;
    const int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    for (int64_t base = (hw & ~(int64_t)1) * HM_GATHER; base < b; base += nhw * HM_GATHER) {
        const int64_t mybase = base + (hw & 1) * HM_GATHER;
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const int64_t r = mybase + k < b ? mybase + k : b - 1;
            return hm_rm_u_halfwave(x + r * ld, y + r * ld, d1, sign_mode, lane);
        });
        if (t < HM_GATHER && mybase + t < b) out[mybase + t] = hm::dist_from_u(u, sqrt_c);
    }
}

__global__ void hm_rows_log_map_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t b, int64_t ld, int d1,
                                       int sign_mode, float* __restrict__ out, int64_t ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    const float* xr = x + t * ld;
    const float* yr = y + t * ld;
    const float u = hm_rm_u(xr, yr, d1, sign_mode);
    const float m = -u;
    const float a = hm::clamp_min_one(u);
    float coef = hm::acosh_c(a) / __builtin_sqrtf(a * a - 1.0f);
    if (coef == coef && coef > 1.0e4f) coef = 1.0e4f;
    for (int k = 0; k < d1; ++k) out[t * ldo + k] = coef * (yr[k] + m * xr[k]);
}

__global__ void hm_rows_exp_map_kernel(const float* __restrict__ x, const float* __restrict__ v, int64_t b, int64_t ld, int d1,
                                       float* __restrict__ out, int64_t ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    const float* xr = x + t * ld;
    const float* vr = v + t * ld;
    float n2 = hm::torch_order_sum([&](int s) { return vr[1 + s] * vr[1 + s]; }, d1 - 1);
    if (n2 == n2 && n2 < 1.0e-8f) n2 = 1.0e-8f;
    const float nn = __builtin_sqrtf(n2);
    const float ch = hm::cosh_c(nn), sh = hm::sinh_c(nn);
    for (int k = 0; k < d1; ++k) out[t * ldo + k] = ch * xr[k] + sh * (vr[k] / nn);
}

__global__ void hm_rows_project_kernel(const float* __restrict__ x, int64_t b, int64_t ld, int d1, float c, float* __restrict__ out,
                                       int64_t ldo)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    const float* xr = x + t * ld;
    float r2 = 0.0f;
    for (int k = 1; k < d1; ++k) r2 = __builtin_fmaf(xr[k], xr[k], r2);
    const float rr = __builtin_sqrtf(r2);
    const float x0 = __builtin_sqrtf(1.0f + (c * rr) * rr);
    for (int k = 1; k < d1; ++k) out[t * ldo + k] = xr[k];
    out[t * ldo] = x0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int hm_row_vs_all(hm_engine* e, int64_t row, int64_t n, float c, float* d_out_dev, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_row_vs_all: engine is NULL");
    if (!d_out_dev || row < 0 || row >= e->n || n < 0 || n > e->n || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_row_vs_all: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (n == 0) return HM_OK;
    {
        const size_t lds = sizeof(float) * ((size_t)HM_MAX_D1 + 4 + (size_t)HM_ROWVS_WAVES * HM_TILE_ROWS * e->RS);
        const void* fn = reinterpret_cast<const void*>(&hm_rowvsall_kernel);
        if (e->attr_done.find(fn) == e->attr_done.end()) {                      // per engine (= per device), not process-wide
            HM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(float) * ((size_t)HM_MAX_D1 + 4 + (size_t)HM_ROWVS_WAVES * HM_TILE_ROWS * 4 * HM_TILE_MAXQ))));
            e->attr_done.insert(fn);
        }
        const int64_t nt = (n + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
        hipLaunchKernelGGL(hm_rowvsall_kernel, dim3((unsigned)std::min<int64_t>((nt + HM_ROWVS_WAVES - 1) / HM_ROWVS_WAVES, HM_ROWVS_GRID_CAP)), dim3(64 * HM_ROWVS_WAVES), lds,
                           (hipStream_t)stream, e->img, e->RS, e->d, row, n, sqrtf(c), e->sign_mode, d_out_dev);
    }
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_pair_distance(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, int64_t b, float c, float* out_dev,
                                void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pair_distance: engine is NULL");
    if (b < 0 || (b > 0 && (!I_dev || !J_dev || !out_dev)) || !(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pair_distance: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_pairdist_kernel, dim3((unsigned)std::min<int64_t>((b + 8 * HM_GATHER - 1) / (8 * HM_GATHER), 8192)), dim3(256), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                       I_dev, J_dev, b, sqrtf(c), e->sign_mode, out_dev);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_midpoint_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev, int64_t b, float c,
                                 float* out_dev, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_midpoint_batch: engine is NULL");
    if (b < 0 || (b > 0 && (!I_dev || !J_dev || !W_dev || !out_dev))) return hm_fail(e, HM_E_ARG, "hm_midpoint_batch: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_midpoint_kernel, dim3((unsigned)((b + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                       I_dev, J_dev, W_dev, b, c, e->sign_mode, out_dev);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

static void hm_rows_changed(hm_engine* e, int64_t first_changed_row, hipStream_t s)
{
    if (first_changed_row < e->n) {          // an existing row changed: cut prediction, argmin seed and arming are void
        e->armed = false;
        e->have_cut = false;
        (void)hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), s);
    }
}

extern "C" int hm_merge_append(hm_engine* e, int32_t i, int32_t j, float w, float c, float* X_dev, int64_t ld, int64_t new_row,
                               void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_merge_append: engine is NULL");
    if (!X_dev || ld < e->d1 || i < 0 || j < 0 || i >= e->n || j >= e->n || new_row < 0 || new_row >= e->max_rows)
        return hm_fail(e, HM_E_ARG, "hm_merge_append: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    hipLaunchKernelGGL(hm_merge_append_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->img, e->RS, e->d, i, j, w, c,
                       e->sign_mode, X_dev, ld, new_row, e->d_rmax2, e->img16, e->KC);
    HM_HIP(hipGetLastError());
    hm_rows_changed(e, new_row, (hipStream_t)stream);
    if (new_row + 1 > e->n) e->n = new_row + 1;
    return HM_OK;
}

extern "C" int hm_merge_append_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev, int64_t count,
                                     float c, float* X_dev, int64_t ld, int64_t first_row, int independent, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_merge_append_batch: engine is NULL");
    if (count < 0 || (count > 0 && (!I_dev || !J_dev || !W_dev || !X_dev)) || ld < e->d1 || first_row < 0 ||
        first_row + count > e->max_rows || count > 1 << 20)
        return hm_fail(e, HM_E_ARG, "hm_merge_append_batch: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (count == 0) return HM_OK;
    if (independent)
        hipLaunchKernelGGL(hm_merge_rows_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                           I_dev, J_dev, W_dev, (int)count, c, e->sign_mode, X_dev, ld, first_row, e->d_rmax2, e->img16, e->KC);
    else
        hipLaunchKernelGGL(hm_merge_batch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->img, e->RS, e->d, I_dev, J_dev, W_dev,
                           (int)count, c, e->sign_mode, X_dev, ld, first_row, e->d_rmax2, e->img16, e->KC);
    HM_HIP(hipGetLastError());
    hm_rows_changed(e, first_row, (hipStream_t)stream);
    if (first_row + count > e->n) e->n = first_row + count;
    return HM_OK;
}

// the same with the batch in HOST memory (the fast tokenizer's plan): one pinned staging copy + one launch
extern "C" int hm_merge_append_batch_host(hm_engine* e, const int32_t* I_host, const int32_t* J_host, const float* W_host, int64_t count,
                                          float c, float* X_dev, int64_t ld, int64_t first_row, int independent, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_merge_append_batch_host: engine is NULL");
    if (count < 0 || count > HM_BATCH_MAX || (count > 0 && (!I_host || !J_host || !W_host)))
        return hm_fail(e, HM_E_ARG, "hm_merge_append_batch_host: bad arguments (at most 4096 merges per call)");
    if (count == 0) return HM_OK;
    for (int64_t t = 0; t < count; ++t) {
        const int64_t lim = independent ? first_row : first_row + t;      // what the merge may read
        if (I_host[t] < 0 || J_host[t] < 0 || I_host[t] >= lim || J_host[t] >= lim || lim > e->max_rows)
            return hm_fail(e, HM_E_ARG, "hm_merge_append_batch_host: a merge reads a row that does not exist yet");
    }
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    if (e->batch_in_flight) { HM_HIP(hipEventSynchronize(e->ev_batch)); e->batch_in_flight = false; }   // staging buffer free again
    memcpy(e->h_batch, I_host, sizeof(int32_t) * (size_t)count);
    memcpy(e->h_batch + HM_BATCH_MAX, J_host, sizeof(int32_t) * (size_t)count);
    memcpy(e->h_batch + 2 * HM_BATCH_MAX, W_host, sizeof(float) * (size_t)count);
    HM_HIP(hipMemcpyAsync(e->d_batch, e->h_batch, sizeof(int32_t) * 3 * HM_BATCH_MAX, hipMemcpyHostToDevice, s));
    HM_HIP(hipEventRecord(e->ev_batch, s));
    e->batch_in_flight = true;
    // (row-existence of the operands was checked above against first_row; hm_merge_append_batch checks the rest)
    const int64_t n_keep = e->n;
    if (first_row > e->n) return hm_fail(e, HM_E_ARG, "hm_merge_append_batch_host: first_row beyond the live rows");
    (void)n_keep;
    return hm_merge_append_batch(e, e->d_batch, e->d_batch + HM_BATCH_MAX, reinterpret_cast<const float*>(e->d_batch + 2 * HM_BATCH_MAX),
                                 count, c, X_dev, ld, first_row, independent, stream);
}

extern "C" int hm_coherence_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev, const int32_t* S_dev,
                                  int64_t b, int ns, float c, float* out_dev, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_coherence_batch: engine is NULL");
    if (b < 0 || ns < 0 || !(c > 0.0f) || (b > 0 && ns > 0 && (!I_dev || !J_dev || !W_dev || !S_dev || !out_dev)))
        return hm_fail(e, HM_E_ARG, "hm_coherence_batch: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (b == 0 || ns == 0) return HM_OK;
    hipLaunchKernelGGL(hm_coherence_kernel, dim3((unsigned)((b + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                       I_dev, J_dev, W_dev, S_dev, b, ns, c, sqrtf(c), e->sign_mode, out_dev);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_project_table(hm_engine* e, float* X_dev, int64_t ld, int64_t n_rows, float c, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_project_table: engine is NULL");
    if (!X_dev || ld < e->d1 || n_rows < 0 || n_rows < e->n || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_project_table: bad arguments (n_rows must cover the live rows)");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    if (n_rows == 0) return HM_OK;
    // ONE dispatch: the norm bounds are rebuilt in the engine's spare (zeroed) pair, which becomes the live one; the kernel
    // zeroes the old pair (the next spare) and the argmin seed (every live row changed) on the side
    uint32_t* fresh = e->d_rmax2 == e->d_rmax2_mem ? e->d_rmax2_mem + 2 : e->d_rmax2_mem;
    const size_t lds = sizeof(float) * 64 * (size_t)(e->d + 2);
    hipLaunchKernelGGL(hm_project_table_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), lds, s, X_dev, ld, e->d, n_rows, c,
                       e->img, e->RS, e->img16, e->KC, e->n, fresh, e->d_rmax2, e->d_seed);
    HM_HIP(hipGetLastError());
    e->d_rmax2 = fresh;
    e->armed = false;
    e->have_cut = false;
    e->topk_f32_thr = 0.0f; e->topk_exact_thr = 0.0f;
    return HM_OK;
}

// ---- engine-independent entry points ----
extern "C" int hm_batch_distance(const float* X_dev, int64_t n1, const float* Y_dev, int64_t n2, int64_t ld_x, int64_t ld_y, int d1,
                                 float c, int sign_mode, float* out_dev, void* stream)
{
    if (n1 < 0 || n2 < 0 || d1 < 2 || ld_x < d1 || ld_y < d1 || !(c > 0.0f)) return hm_fail(nullptr, HM_E_ARG, "hm_batch_distance: bad arguments");
    if (n1 == 0 || n2 == 0) return HM_OK;
    if (!X_dev || !Y_dev || !out_dev) return hm_fail(nullptr, HM_E_ARG, "hm_batch_distance: NULL pointer");
    const int64_t total = n1 * n2;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 8 * HM_GATHER - 1) / (8 * HM_GATHER), 8192);
    hipLaunchKernelGGL(hm_dense_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X_dev, n1, Y_dev, n2, ld_x, ld_y, d1,
                       sqrtf(c), sign_mode, out_dev);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_rows_minkowski(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, int sign_mode, float* out_dev,
                                 void* stream)
{
    if (b < 0 || d1 < 2 || ld < d1) return hm_fail(nullptr, HM_E_ARG, "hm_rows_minkowski: bad arguments");
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rows_minkowski_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x_dev, y_dev, b,
                       ld, d1, sign_mode, out_dev);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_rows_distance(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, float c, int sign_mode,
                                float* out_dev, void* stream)
{
    if (b < 0 || d1 < 2 || ld < d1 || !(c > 0.0f)) return hm_fail(nullptr, HM_E_ARG, "hm_rows_distance: bad arguments");
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rows_distance_kernel, dim3((unsigned)std::min<int64_t>((b + 8 * HM_GATHER - 1) / (8 * HM_GATHER), 8192)), dim3(256), 0, (hipStream_t)stream, x_dev, y_dev, b,
                       ld, d1, sqrtf(c), sign_mode, out_dev);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_rows_log_map(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, int sign_mode, float* out_dev,
                               int64_t ld_out, void* stream)
{
    if (b < 0 || d1 < 2 || ld < d1 || ld_out < d1) return hm_fail(nullptr, HM_E_ARG, "hm_rows_log_map: bad arguments");
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rows_log_map_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x_dev, y_dev, b,
                       ld, d1, sign_mode, out_dev, ld_out);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_rows_exp_map(const float* x_dev, const float* v_dev, int64_t b, int64_t ld, int d1, float* out_dev, int64_t ld_out,
                               void* stream)
{
    if (b < 0 || d1 < 2 || ld < d1 || ld_out < d1) return hm_fail(nullptr, HM_E_ARG, "hm_rows_exp_map: bad arguments");
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rows_exp_map_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x_dev, v_dev, b,
                       ld, d1, out_dev, ld_out);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_rows_project(const float* x_dev, int64_t b, int64_t ld, int d1, float c, float* out_dev, int64_t ld_out,
                               void* stream)
{
    if (b < 0 || d1 < 2 || ld < d1 || ld_out < d1) return hm_fail(nullptr, HM_E_ARG, "hm_rows_project: bad arguments");
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rows_project_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x_dev, b, ld, d1,
                       c, out_dev, ld_out);
    HM_HIP0(hipGetLastError());
    return HM_OK;
}
