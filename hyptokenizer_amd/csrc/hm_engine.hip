// hm_engine.hip -- gfx950 kernels and C ABI of the hyperbolic merge engine (include/hypmerge.h).
//
// Hot path: the all-pairs Lorentz-distance candidate search of
// HyperbolicTokenizer._find_merge_candidates (tokenizer/hyperbolic_merge.py:247-269) and
// FastHyperbolicTokenizer._find_merge_candidates_fast (tokenizer/fast_hyperbolic_merge.py:336-374),
// plus the log-map / exp-map "midpoint" of _merge_tokens (hyperbolic_merge.py:326-340).
//
// Kernel inventory
//   hm_scan_kernel      pair scan: X.G.X^T on the matrix cores as a PREFILTER -- bf16 form
//                       (v_mfma_f32_32x32x16_bf16 on a bf16 image, default for d >= 24) or fp32 form
//                       (v_mfma_f32_32x32x2_f32, exact fmaf chain) -- stationary rows in registers,
//                       partner rows streamed through LDS by LDS-DMA, epilogue = bound test on the
//                       acosh argument (widened by a rigorous error bound, hm_scan_delta) +
//                       wave-aggregated emission of survivors.  Never materialises the N x N matrix.
//   hm_post_*           exact canonical distance for the survivors, threshold test, exact
//                       (d, i, j) selection (min / radix narrowing / rank sort).
//   hm_seed_init_kernel counters + running-key seed of an argmin search (previous nearest pair).
//   hm_midpoint_kernel, hm_merge_append_kernel   log-map -> scale -> exp-map -> project.
//   hm_row_argmin_kernel, hm_pairdist_kernel, hm_rowvsall_kernel, hm_dense_kernel, hm_rows_* :
//                       one-vs-all / gathered / dense forms on the canonical arithmetic.
//
// Data layout in HBM: the fp32 "scan image" img[rows_alloc][RS], RS = 4*NG + 4 (+ 4 when needed to
// make the 16-byte chunks per row odd), NG = groups of 4 spatial coordinates.  Group g holds spatial
// coordinates s = 4g..4g+3 in the order [s0, s2, s1, s3] so that lane-half h of a wave reads ONE
// 8-byte word (position 2h) holding its operands for the two MFMA k-steps of the group; the time
// chunk [x0, 0, 0, 0] is last.  The bf16 image img16[rows_alloc][32*KS + 16 bytes]: KS k-steps of 16
// bf16 (spatial coordinates, then the time coordinate split hi + lo in the last four slots) and a
// trailing chunk [x0 fp32, 0, 0, 0] (odd chunk count: conflict-free ds_read_b128).  A 64-row tile of
// either image is one contiguous block -> LDS-DMA in 1 KiB pieces, no padding.
// The macros below are compile-time knobs; the ones marked "experiment" are measured dead ends kept
// for the record, HM_DIAG_* are timing diagnostics that break the results.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/hypmerge.h"
#include "hm_device_math.h"

#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------
#ifndef HM_ODD_STRIDE
#define HM_ODD_STRIDE 1            // pad image rows to an odd number of 16-byte chunks (LDS bank spread)
#endif
#ifndef HM_PREFETCH_B
#define HM_PREFETCH_B 1            // fetch B fragments one k-group ahead of their MFMAs
#endif
#ifndef HM_CHUNK_TILES
#define HM_CHUNK_TILES 32          // fp32 form: column tiles per block (upper bound; shrunk for small tables)
#endif
#ifndef HM_CHUNK_TILES_BF16
#define HM_CHUNK_TILES_BF16 96     // bf16 form: 64-column tiles, ~5x shorter per column than the fp32 form
#endif
#ifndef HM_TAIL_FRACTION
#define HM_TAIL_FRACTION 0.20      // share of the work issued last in quarter-size chunks
#endif
#ifndef HM_SKIP_EMPTY
#define HM_SKIP_EMPTY 0            // triangular item numbering (no block left of the diagonal): no gain measured
#endif
#ifndef HM_SWIZZLE_IDENTITY
#define HM_SWIZZLE_IDENTITY 0      // experiment: swizzle sizing without the mirrored mapping
#endif
#ifndef HM_SWIZZLE_EXTRA
#define HM_SWIZZLE_EXTRA 0         // experiment: extra chunks per row (breaks the XCD alignment)
#endif
#ifndef HM_TILE_ROTATE
#define HM_TILE_ROTATE 1           // per-block rotation of the tile order inside a chunk
#endif
#ifndef HM_XCD_SWIZZLE
#define HM_XCD_SWIZZLE 0           // XCD-aware slot -> chunk order (L2-local but slower: see DESIGN.md section 6)
#endif
#ifndef HM_DMA_INTERLEAVE
#define HM_DMA_INTERLEAVE 1        // issue the next tile's LDS-DMA between MFMAs instead of ahead of them
#endif
#ifndef HM_MIN_WAVES
#define HM_MIN_WAVES 2             // __launch_bounds__ second argument (waves per SIMD): 2 blocks per CU
#endif
#ifndef HM_TM_F32
#define HM_TM_F32 2                // fp32 form: 64-row MFMA tiles per wave (block = 256 rows, 2 blocks per CU, 226 VGPRs)
#endif
// timing diagnostics only (wrong results): drop one ingredient of the bf16 scan loop
#ifndef HM_DIAG_NO_DMA
#define HM_DIAG_NO_DMA 0
#endif
#ifndef HM_DIAG_NO_EPI
#define HM_DIAG_NO_EPI 0
#endif
#ifndef HM_DIAG_NO_BARRIER
#define HM_DIAG_NO_BARRIER 0
#endif
#ifndef HM_DIAG_NO_LDS
#define HM_DIAG_NO_LDS 0
#endif
#ifndef HM_DIAG_SAME_TILES
#define HM_DIAG_SAME_TILES 0
#endif
#ifndef HM_DIAG_NO_SLOW
#define HM_DIAG_NO_SLOW 0
#endif
#ifndef HM_DIAG_NEVER_SLOW
#define HM_DIAG_NEVER_SLOW 0
#endif
#ifndef HM_DIAG_TIMES
#define HM_DIAG_TIMES 0            // per-block {ticks, slow-path entries, passes, tiles} into the hist buffer (hm_debug_read_hist)
#endif
#ifndef HM_DIAG_NO_MFMA
#define HM_DIAG_NO_MFMA 0
#endif
#ifndef HM_EVENT_FLAGS
#define HM_EVENT_FLAGS hipEventDisableSystemFence   // scan timing events: no system-scope fence around the launch
#endif
#ifndef HM_ARM_NEXT
#define HM_ARM_NEXT 1              // an argmin search leaves counters + running key initialised for the next one of the same range
#endif
#ifndef HM_PERSIST
#define HM_PERSIST 0               // experiment (off): bf16 form, one resident block per slot walks an equal share of the
                                   // tile sequence -- no launch tail, but measured 50 % slower than the chunked grid
#endif
#ifndef HM_DYN_K1
#define HM_DYN_K1 32               // experiment (HM_PERSIST == 2): tiles per chunk taken from the atomic counter, early part
#endif
#ifndef HM_DYN_K2
#define HM_DYN_K2 8                // ... and late part of the tile sequence
#endif
#ifndef HM_DYN_SPLIT
#define HM_DYN_SPLIT 0.8
#endif
#ifndef HM_PERSIST_BLOCKS_PER_CU
#define HM_PERSIST_BLOCKS_PER_CU 2
#endif
#ifndef HM_SETTLE_PROLOGUE
#define HM_SETTLE_PROLOGUE 1       // compiler-visible vmcnt(0) before the tile loop (keeps hipcc's lazy waits out of it)
#endif
#ifndef HM_DMA_GROUPED_F32
#define HM_DMA_GROUPED_F32 1         // fp32 form: same grouped, unconditional LDS-DMA as the bf16 form (instead of one piece between MFMA groups)
#endif
#ifndef HM_DMA_GROUPED
#define HM_DMA_GROUPED 1           // bf16 form: LDS-DMA pieces issued four per statement, unconditionally (see hm_dma_group)
#endif
#ifndef HM_TM_BF16
#define HM_TM_BF16 2               // bf16 form: 64 stationary rows per wave (each LDS fragment read feeds two MFMAs)
#endif
#ifndef HM_DIST_BF16
#define HM_DIST_BF16 1             // bf16 form: tiles in flight ahead of the computed one (ring of DIST + 1 slots; deeper rings measured no gain)
#endif
#ifndef HM_TN_BF16
#define HM_TN_BF16 1               // bf16 form: 32-column MFMA tiles per accumulator group
#endif
#ifndef HM_SUB_BF16
#define HM_SUB_BF16 2              // bf16 form: a streamed tile (64 partner rows) is walked as SUB groups of 32 * TN columns,
                                   // accumulators reused: 64x32 outputs per wave and group, 149 VGPRs
#endif
#ifndef HM_MIN_WAVES_BF16
#define HM_MIN_WAVES_BF16 2        // bf16 form: blocks per CU the register budget is sized for
#endif
#ifndef HM_TM4_MIN_ROWS
#define HM_TM4_MIN_ROWS 80000      // bf16 form: from this many live rows on, 128 stationary rows per wave (512-row blocks)
#endif
#ifndef HM_WPB_BF16
#define HM_WPB_BF16 4              // bf16 form: waves per block (all share each streamed 64-row tile)
#endif
#define HM_MAX_BLOCK_ROWS 512
#define HM_MAX_D1 132              // largest table width (d + 1 <= 129) rounded up
#define HM_COLS_PER_TILE 64        // partner rows per LDS tile
#define HM_TIE_SLACK 1024u         // ulps of u' that are treated as "may still order before" (d is 2.5-ulp monotone)
#define HM_MODE_TOPK 0
#define HM_MODE_ARGMIN 1
#define HM_MODE_HIST 2
#define HM_HIST_BINS 256
#define HM_DIGIT_BINS 4096
#define HM_RANK_LIMIT 49152        // rank sort is O(M^2): narrow by radix digits above this

// items before local row R when row r owns (chunks - floor(r / m)) items
__host__ __device__ __forceinline__ int hm_tri_cum(int R, int chunks, int m)
{
    const int q = R / m;
    return R * chunks - (m * (q * (q - 1) / 2) + (R - m * q) * q);
}

struct ScanArgs {
    const float* img;
    const unsigned char* img16;   // bf16 image (BF = 1 kernels)
    int bf16;                     // host-side: which form this launch uses
    int tm4;                      // host-side: bf16 form with 128 stationary rows per wave (512-row blocks; large tables)
    int n;                  // live rows
    int row_begin, row_end; // i range
    int rb_first;           // first row block
    int nct;                // column tiles in total = ceil(n / 64)
    // work decomposition: 1-D grid.  Blocks [0, n_items_a) take `ch_a` column tiles each of row
    // blocks [rb_first, rb_split); the rest take `ch_b` (smaller) tiles of row blocks >= rb_split.
    // Big items first, small items last: the tail of the launch is made of short blocks.
    int n_items_a, chunks_a, ch_a, ctmin_a;
    int rb_split, chunks_b, ch_b, ctmin_b;
    // XCD-aware slot order (HM_XCD_SWIZZLE): chunks per row block are a multiple of 16 and the
    // chunk of slot s (= block id modulo chunks) is mirrored inside each group of 16 (slots 8..15 take
    // chunks 15..8).  Blocks b and b + 8 share an XCD (observed round-robin placement; a speed
    // assumption only), so every XCD keeps streaming the same <= n/8 partner rows -- they stay in
    // its 4 MiB L2 -- and the mirrored pairs give every XCD the same share of the triangle.
    int swizzle;
    int skip_empty, rows_a, m_a, rows_b, m_b;   // triangular item numbering (no empty blocks)
    // persistent decomposition (bf16 form, TOPK / ARGMIN): the tiles right of the diagonal of row blocks
    // rb_first .. rb_first + p_nrb - 1, in row-major order, form one sequence of p_total tiles; block b of
    // p_grid resident blocks walks the share [b * p_total / p_grid, (b + 1) * p_total / p_grid) and reloads
    // its stationary rows whenever the share crosses into the next row block.
    long long p_total;
    int p_nrb, p_grid;
    // HM_PERSIST == 2: the sequence is cut into chunks -- p_c1 chunks of p_k1 tiles, then chunks of p_k2 tiles --
    // which the resident blocks take from an atomic counter (ctr[5]; block b starts with chunk b): blocks
    // that run faster (the older resident block of a CU wins the arbitration) simply take more of them.
    int p_k1, p_k2, p_c1;
    float u_hi;             // candidate prefilter: u < u_hi
    float u_lo;             // surely-below-threshold bound: u' < u_lo
    uint32_t cut_bits;      // emit when bits(u') <= cut_bits (or not sure)
    int tie_imax;           // zero-distance ties are emitted only for rows i <= tie_imax
    int thr_pos;            // thr > 0: u' == 1 gives d == 0, surely a candidate
    uint4* ent;
    uint32_t ent_cap;
    uint32_t* ctr;              // [0] emitted entries
    unsigned long long* ctr64;  // [0] sure count  [1] running best key (argmin)
    uint32_t* hist;             // HIST mode: HM_HIST_BINS bins
    uint32_t hist_lo;
    uint32_t hist_shift;
    int sample_stride;
    const uint32_t* rmax2_bits; // [0] largest squared row norm, [1] largest squared spatial norm (float bits)
};

// ------------------------------------------------------------------------------------------------
// image construction
// ------------------------------------------------------------------------------------------------
// floats per image row: NG spatial chunks, an optional all-zero pad chunk that makes the number of
// 16-byte chunks odd (rows then start 16 chunk-slots apart modulo the 256-byte LDS bank row), and
// the time chunk LAST (hm_img_time reads RS - 4).
__host__ __device__ constexpr int hm_row_floats(int NG) { return 4 * NG + 4 + ((HM_ODD_STRIDE && ((NG + 1) % 2 == 0)) ? 4 : 0); }

__device__ __forceinline__ int hm_pos_in_group(int s) { return ((s & 1) << 1) | ((s >> 1) & 1); }  // 0,2,1,3

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

__device__ __forceinline__ float hm_img_spatial(const float* img, int RS, int64_t row, int s)
{
    return img[row * RS + 4 * (s >> 2) + hm_pos_in_group(s & 3)];
}
__device__ __forceinline__ float hm_img_time(const float* img, int RS, int64_t row) { return img[row * RS + RS - 4]; }

// canonical u (argument of acosh) between two image rows: products rounded separately, summed in
// torch's reduction order, then fl(fl(x0*y0) - S)  (DESIGN.md "Canonical arithmetic")
__device__ __forceinline__ float hm_img_u(const float* img, int RS, int d, int64_t a, int64_t b, int sign_mode)
{
    const float* ra = img + a * RS;
    const float* rb = img + b * RS;
    const float S = hm::torch_order_sum(
        [&](int s) {
            const int o = 4 * (s >> 2) + hm_pos_in_group(s & 3);
            return ra[o] * rb[o];
        },
        d);
    const float t = ra[RS - 4] * rb[RS - 4];
    const float m = t - S;
    return sign_mode ? m : -m;
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
        if (r2 > 0.0f) atomicMax(rmax2_bits, hm::fbits(r2));
        if (s2 > 0.0f) atomicMax(rmax2_bits + 1, hm::fbits(s2));
    }
}

// bf16 image row: KS x 16 K-slots as bf16 (round to nearest even) followed by one 16-byte chunk
// [x0 as fp32, 0, 0, 0] (2*KS + 1 chunks per row: always odd, so the ds_read_b128 fragment reads of
// 32 consecutive rows fall on distinct 16-byte bank slots).  Slots 0..d-1 hold the spatial
// coordinates; the LAST FOUR slots hold the time coordinate split as x0 ~ hi + lo:
//   streamed (B) encoding  [hi, lo, hi, 0];  the stationary (A) side rewrites its copy in registers to
//   [-hi, -hi, -lo, 0], so the MFMA adds -(hi*hi' + hi*lo' + lo*hi') = -x0*y0 (1 + O(2^-16)).
__device__ __forceinline__ uint32_t hm_pack_bf16(float lo, float hi)
{
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

__device__ __forceinline__ uint4 hm_bf16_chunk(const float* spatial /* x[1..d] */, float x0, int d, int KS, int c)
{
    // chunk c covers K-slots 8c .. 8c+7
    float f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int sidx = 8 * c + q; f[q] = sidx < d ? spatial[sidx] : 0.0f; }
    if (c == 2 * KS - 1) {
        const __bf16 hb = (__bf16)x0;
        const float hi = (float)hb;
        const float lo = x0 - hi;
        f[4] = hi; f[5] = lo; f[6] = hi; f[7] = 0.0f;
    }
    return make_uint4(hm_pack_bf16(f[0], f[1]), hm_pack_bf16(f[2], f[3]), hm_pack_bf16(f[4], f[5]), hm_pack_bf16(f[6], f[7]));
}

__global__ void hm_build_image16_kernel(const float* __restrict__ X, int64_t ld, int d, int KS, unsigned char* __restrict__ img16,
                                        int64_t row_begin, int64_t row_end)
{
    const int CH = 2 * KS + 1;                        // 16-byte chunks per row
    const int64_t total = (row_end - row_begin) * CH;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row_begin + t / CH;
        const int c = (int)(t % CH);
        uint4 v = make_uint4(0, 0, 0, 0);
        const float* xr = X + row * ld;
        if (c == CH - 1) v.x = hm::fbits(xr[0]);
        else v = hm_bf16_chunk(xr + 1, xr[0], d, KS, c);
        *reinterpret_cast<uint4*>(img16 + (row * CH + c) * 16) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// pair scan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hm_wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t hm_wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// Bound on |u_f - u_c| between the MFMA prefilter value and the canonical value of the same pair
// (gamma_n bounds on both roundings, |terms| <= rmax2).  `kterms` = fp32 form: floats per image row;
// bf16 form: 16 * k-steps.  bf16 operands: each spatial product carries <= 2 * 2^-9 (+ 2^-18) relative
// error, so the sum is off by <= 2^-8 (1 + 2^-9) ||x_s|| ||y_s|| <= 0.00392 * (largest squared spatial
// norm); the hi+lo split of the time coordinate leaves <= 2^-15 * x0*y0 (x0^2 <= rmax2).
__device__ __forceinline__ float hm_scan_delta(bool bf, int kterms, const uint32_t* rmax2_bits)
{
    const float rmax2 = hm::bitsf(rmax2_bits[0]);
    float delta = ((float)(kterms + 8) * 1.1920929e-07f) * rmax2 * 1.0001f;
    if (bf) delta += 0.00392f * hm::bitsf(rmax2_bits[1]) + 3.1e-5f * rmax2;
    return delta;
}

// Seed of the argmin search's running key, kept on the device between searches: the key of the last
// nearest pair found.  While rows are only appended that pair still exists, so its key bounds the next
// search from its first tile on (without it the bound is loose until some wave happens to reach a near
// pair).  `valid` is cleared whenever an existing row changes.
struct ArgminSeed { unsigned long long key; uint32_t i, valid; };

// also clears the scan's counters (one launch instead of a memset + a launch)
__global__ void hm_seed_init_kernel(const ArgminSeed* __restrict__ seed, unsigned long long* __restrict__ ctr64, uint32_t* __restrict__ ctr,
                                    int row_begin, int row_end)
{
    if (threadIdx.x < 8) ctr[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        const bool use = seed->valid != 0u && (int)seed->i >= row_begin && (int)seed->i < row_end;
        ctr64[0] = ~0ull;
        ctr64[1] = use ? seed->key : ~0ull;
    }
}

// CNT (1..4) consecutive 1 KiB LDS-DMA pieces in one statement: one M0 write; the immediate offset of
// global_load_lds applies to the global and to the LDS address alike, so the pieces share both bases.
template <int CNT>
__device__ __forceinline__ void hm_dma_group(const char* src, uint32_t dst)
{
    uint32_t keep;
    if constexpr (CNT == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else if constexpr (CNT == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else if constexpr (CNT == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// PIECES consecutive pieces starting at piece index FIRST, four per statement
template <int PIECES, int FIRST = 0>
__device__ __forceinline__ void hm_dma_run(const char* src, uint32_t dst)
{
    if constexpr (FIRST < PIECES) {
        hm_dma_group<(PIECES - FIRST < 4 ? PIECES - FIRST : 4)>(src + FIRST * 1024, dst + FIRST * 1024u);
        hm_dma_run<PIECES, FIRST + 4>(src, dst);
    }
}

// BF = 0: exact fp32 prefilter (v_mfma_f32_32x32x2_f32 on the fp32 image; NG = groups of 4 spatial
//         coordinates).  BF = 1: bf16 prefilter (v_mfma_f32_32x32x16_bf16 on the bf16 image, spatial
//         part only; NG = k-steps of 16; the time product x0*y0 is added in fp32 in the epilogue).
// Either way the result only selects survivors; every reported distance is re-evaluated with the
// canonical arithmetic, and the bound `delta` on |u_f - u_c| widens every comparison accordingly.
// TM = 32-row MFMA tiles per wave along the stationary rows (block = 4 waves = 128*TM rows).
template <int NG, int SIGN, int MODE, int BF, int TM, int WPB, int TN>
__global__ __launch_bounds__(64 * WPB, BF ? HM_MIN_WAVES_BF16 : HM_MIN_WAVES) void hm_scan_kernel(const ScanArgs p)
{
    constexpr int SUB = BF ? HM_SUB_BF16 : 1;      // column groups per streamed tile
    constexpr int SCOLS = 32 * TN;                 // columns per group (TN 32-column MFMA tiles)
    constexpr int COLS = SCOLS * SUB;              // partner rows per streamed tile
    constexpr int RS = hm_row_floats(NG);          // fp32 image: floats per row
    constexpr int RB16 = 32 * NG + 16;             // bf16 image: bytes per row (NG x 16 bf16 + [x0 fp32, pad])
    constexpr int TILE_BYTES = BF ? COLS * RB16 : COLS * RS * 4;
    constexpr int NP = BF ? NG : NG + 1;           // k-steps: fp32: NG spatial groups + time; bf16: NG steps of 16
    constexpr int NPIECE = TILE_BYTES / 1024;      // 1 KiB pieces per 64-row tile (one per 16-byte chunk column)
    constexpr int TCH = RS / 4 - 1;                // fp32 image: chunk index of the time group
    constexpr int PPW = (NPIECE + WPB - 1) / WPB;  // LDS-DMA pieces per wave and tile (tile padded to PPW * WPB KiB)
    constexpr int TILE_LDS = PPW * WPB * 1024;     // LDS bytes per ring slot
    constexpr int DIST = BF ? HM_DIST_BF16 : 1;    // tiles in flight ahead of the one being computed
    constexpr int NBUF = DIST + 1;                 // ring slots
    constexpr int WAVE_ROWS = 32 * TM;
    constexpr int BLOCK_ROWS = WPB * WAVE_ROWS;     // WPB waves per block share every streamed tile
    constexpr int NTHREADS = 64 * WPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF * TILE_LDS (+ hist)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    constexpr bool PERSIST = BF && HM_PERSIST && MODE != HM_MODE_HIST;
    static_assert(!PERSIST || (32 * TM * WPB) % (32 * TN) == 0, "row blocks must start on a tile boundary");
    constexpr int TPR = (32 * TM * WPB) / (32 * TN) > 0 ? (32 * TM * WPB) / (32 * TN) : 1;   // diagonal advance per row block, in tiles
    // tiles of the sequence in front of relative row block q
    auto seq_before = [&](long long q) { return q * p.nct - (long long)TPR * ((long long)p.rb_first * q + q * (q - 1) / 2); };
    constexpr bool DYN = PERSIST && HM_PERSIST == 2;
    long long pos = 0, pos_end = 0;
    uint32_t chunk = blockIdx.x;     // DYN: the chunk being processed
    uint32_t next_raw = 0;           // DYN: destination of the asynchronous counter fetch (wave 0, lane 0)
    bool next_pending = false;
    auto chunk_range = [&](uint32_t c) {         // DYN: [pos, pos_end) of chunk c; empty when the sequence is used up
        const long long c1 = p.p_c1;
        pos = (long long)c < c1 ? (long long)c * p.p_k1 : c1 * p.p_k1 + ((long long)c - c1) * p.p_k2;
        pos_end = pos + ((long long)c < c1 ? p.p_k1 : p.p_k2);
        if (pos_end > p.p_total) pos_end = p.p_total;
    };
    if (PERSIST && !DYN) {
        pos = p.p_total * (long long)blockIdx.x / (long long)gridDim.x;
        pos_end = p.p_total * ((long long)blockIdx.x + 1) / (long long)gridDim.x;
        if (pos >= pos_end) return;
    }
    uint32_t sure_total = 0;         // per-lane partial of the sure count
    unsigned long long gk = ~0ull;   // last value read of the running argmin key
    unsigned long long gk_raw = ~0ull;   // destination of the asynchronous key load
    bool gk_pending = false;
    const uint32_t lds_base = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);

    uint32_t* lhist = nullptr;
    if (MODE == HM_MODE_HIST) {
        lhist = reinterpret_cast<uint32_t*>(smem + NBUF * TILE_LDS);
        for (int t = threadIdx.x; t < HM_HIST_BINS; t += NTHREADS) lhist[t] = 0;
    }

#if HM_DIAG_TIMES
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t diag_slow = 0, diag_pass = 0, diag_tiles = 0;
    unsigned long long diag_slow_ticks = 0;
#endif
    uint32_t* s_next = reinterpret_cast<uint32_t*>(smem + NBUF * TILE_LDS);     // DYN: next chunk, published by wave 0
    for (;;) {   // DYN: one pass per chunk taken from the counter; otherwise a single pass
    if (DYN) {
        chunk_range(chunk);
        if (pos >= pos_end) break;               // every wave of the block sees the same chunk: uniform exit
        // fetch the next chunk now, use it when this one is done: the counter's latency hides behind the
        // chunk's tiles.  An asm atomic the compiler does not wait for; picked up behind the first tile's
        // counted wait and published through LDS (read by all waves after this chunk's last barrier).
        if (wave == 0) {
            if (lane == 0) {
                const uint32_t one = 1u;
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(next_raw) : "v"(&p.ctr[5]), "v"(one) : "memory");
            }
            next_pending = true;
        }
    }
    do {   // one pass per (row block, run of column tiles); a single pass unless PERSIST
#if HM_DIAG_TIMES
    ++diag_pass;
#endif
    int rb, ct0, ct1;
    if (PERSIST) {
        int lo = 0, hi = p.p_nrb - 1;                       // largest q with seq_before(q) <= pos
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seq_before(mid) <= pos) lo = mid; else hi = mid - 1;
        }
        const long long q0 = seq_before(lo), q1 = seq_before(lo + 1);
        const long long run_end = pos_end < q1 ? pos_end : q1;
        rb = p.rb_first + lo;
        ct0 = rb * TPR + (int)(pos - q0);
        ct1 = ct0 + (int)(run_end - pos);
        pos = run_end;
    }
    auto slot_to_chunk = [&](int slot) {
        if (!p.swizzle || HM_SWIZZLE_IDENTITY) return slot;
        const int q = slot & 15;
        return (slot & ~15) | (q < 8 ? q : 23 - q);
    };
    if (PERSIST) {
    } else if (p.skip_empty) {
        // triangular item numbering: local row r of a phase owns chunks floor(r / m) .. chunks - 1
        // (m = rows per chunk step of the diagonal), so no block is launched left of the diagonal
        const bool ph_a = (int)blockIdx.x < p.n_items_a;
        const int it = ph_a ? (int)blockIdx.x : (int)blockIdx.x - p.n_items_a;
        const int chunks = ph_a ? p.chunks_a : p.chunks_b;
        const int m = ph_a ? p.m_a : p.m_b;
        int lo = 0, hi = (ph_a ? p.rows_a : p.rows_b) - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (hm_tri_cum(mid, chunks, m) <= it) lo = mid; else hi = mid - 1;
        }
        const int c = lo / m + (it - hm_tri_cum(lo, chunks, m));
        rb = (ph_a ? p.rb_first : p.rb_split) + lo;
        ct0 = (ph_a ? p.ctmin_a : p.ctmin_b) + c * (ph_a ? p.ch_a : p.ch_b);
        ct1 = ct0 + (ph_a ? p.ch_a : p.ch_b);
    } else if ((int)blockIdx.x < p.n_items_a) {
        rb = p.rb_first + (int)blockIdx.x / p.chunks_a;
        ct0 = p.ctmin_a + slot_to_chunk((int)blockIdx.x % p.chunks_a) * p.ch_a;
        ct1 = ct0 + p.ch_a;
    } else {
        const int it = (int)blockIdx.x - p.n_items_a;
        rb = p.rb_split + it / p.chunks_b;
        ct0 = p.ctmin_b + slot_to_chunk(it % p.chunks_b) * p.ch_b;
        ct1 = ct0 + p.ch_b;
    }
    if (ct0 < (rb * BLOCK_ROWS) / COLS) ct0 = (rb * BLOCK_ROWS) / COLS;   // left of the diagonal: no i < j
    if (ct1 > p.nct) ct1 = p.nct;
    if (ct0 >= ct1) return;

    const int i0w = rb * BLOCK_ROWS + wave * WAVE_ROWS;   // first stationary row of this wave
    const bool wave_active = (i0w < p.row_end) && (i0w + WAVE_ROWS - 1 >= p.row_begin) && (i0w < p.n);
    const bool rows_full = (i0w >= p.row_begin) && (i0w + WAVE_ROWS - 1 < p.row_end);

    // The MFMA result u_f (plain fmaf chain) and the canonical u_c (torch reduction order) are two
    // roundings of the same exact form; |u_f - u_c| <= delta (gamma_n bound on both, |terms| <= rmax2).
    const float delta = hm_scan_delta(BF != 0, BF ? 16 * NG : RS, p.rmax2_bits);
    const float pre_f = p.u_hi + delta;              // candidate prefilter on u_f
    const float lo_f = p.u_lo - delta;               // u_f below this: canonical d < thr for sure
    const float zmax_f = 1.0f - delta;               // u_f at or below this: canonical u <= 1, d == 0
    const bool cut_all = (p.cut_bits == 0xffffffffu);
    const float cut_f = cut_all ? p.u_hi : hm::bitsf(p.cut_bits) + delta;


    // the running argmin key as it stands when the block starts (seeded, or found by earlier blocks): without it the
    // block's first tile would run on the threshold bound alone -- with a threshold inside the bulk of the distance
    // distribution that is thousands of emissions per block.  Waited for together with the fragment loads below.
    if (MODE == HM_MODE_ARGMIN) {
        const unsigned long long g0 = __hip_atomic_load(&p.ctr64[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g0 < gk) gk = g0;
    }

    // ---- stationary A fragments: lane (r, h) keeps its operands of every k-step in registers ----
    float2 a[BF ? 1 : TM][BF ? 1 : NP];            // fp32 form: 2 operands (two k-steps) per group
    uint4 a16[BF ? TM : 1][BF ? NP : 1];           // bf16 form: 8 bf16 per 16-wide k-step
    if constexpr (BF) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const unsigned char* src = p.img16 + (int64_t)(i0w + 32 * tm + r) * RB16 + 16 * h;
#pragma unroll
            for (int g = 0; g < NP; ++g) a16[tm][g] = *reinterpret_cast<const uint4*>(src + 32 * g);
            // stationary side of the time product: [hi, lo, hi, 0] -> [-hi, -hi, -lo, 0] (last 4 slots,
            // held by the h = 1 half of the last k-step)
            if (h == 1) {
                const uint32_t z = a16[tm][NP - 1].z;
                const uint32_t hi16 = z & 0xffffu, lo16 = z >> 16;
                a16[tm][NP - 1].z = (hi16 | (hi16 << 16)) ^ 0x80008000u;
                a16[tm][NP - 1].w = lo16 ^ 0x8000u;
            }
        }
    } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const float* src = p.img + (int64_t)(i0w + 32 * tm + r) * RS + 2 * h;
#pragma unroll
            for (int g = 0; g < NP; ++g) a[tm][g] = *reinterpret_cast<const float2*>(src + 4 * (g < NG ? g : TCH));
            a[tm][NG].x = -a[tm][NG].x;            // time step: acc = S - x0*y0 = -M
        }
    }

    // ---- LDS-DMA of one 64-row tile: NP pieces of 1 KiB, piece q handled by wave q % 4.
    // Issued through inline asm so that hipcc does not drain it (vmcnt(0)) before the LDS reads of
    // the tile being computed; the matching wait is the explicit vmcnt(0) in front of the barrier
    // that ends each iteration (cdna_hip_programming.md section 5.7, LDS-DMA recipe).
    auto dma_piece = [&](int ct, int buf, int q) {          // q: wave-uniform piece index in [0, PPW * WPB)
        if (HM_DIAG_NO_DMA && BF) return;
        // pieces past NPIECE (ring-slot padding) read the first KiB of the next tile: in bounds, unused
        const char* src = (BF ? reinterpret_cast<const char*>(p.img16) : reinterpret_cast<const char*>(p.img)) +
                          (int64_t)ct * TILE_BYTES + lane * 16 + q * 1024;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (uint32_t)TILE_LDS + (uint32_t)q * 1024u);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(dst)
                     : "memory");
    };
    auto dma_tile = [&](int ct, int buf) {
        if constexpr ((BF && HM_DMA_GROUPED) || (!BF && HM_DMA_GROUPED_F32)) {
            // bf16 form: wave w moves the PPW consecutive pieces [w * PPW, (w + 1) * PPW), four per statement
            if (HM_DIAG_NO_DMA) return;
            const char* src = (BF ? reinterpret_cast<const char*>(p.img16) : reinterpret_cast<const char*>(p.img)) +
                              (int64_t)(HM_DIAG_SAME_TILES ? (ct & 7) : ct) * TILE_BYTES + lane * 16 + wave * (PPW * 1024);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (uint32_t)TILE_LDS + (uint32_t)wave * (PPW * 1024u));
            hm_dma_run<PPW>(src, dst);
        } else {
#pragma unroll
            for (int t = 0; t < PPW; ++t) dma_piece(ct, buf, t * WPB + wave);
        }
    };
    // every wave issues exactly PPW pieces per tile, so "tile x has landed" is a counted wait:
    // all but the (tiles issued after x) * PPW youngest vector-memory operations are complete
    auto wait_tiles_in_flight = [&](int tiles) {
        if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW < 63 ? 3 * PPW : 63) : "memory");
    };

    // HIST mode visits every sample_stride-th tile only (a cheap estimate of the u' distribution)
    int ct_step = 1;
    if (MODE == HM_MODE_HIST) {
        ct_step = p.sample_stride;
        ct0 += (ct_step - (ct0 + rb * 7) % ct_step) % ct_step;
        if (ct0 >= ct1) return;
    }

    // Blocks that share a column chunk (and, with the XCD-aware slot order, an L2) would otherwise
    // walk the same tiles in lockstep and queue on the same cache lines: each block starts at its
    // own tile of the chunk and wraps around.
    const int ntile = (ct1 - ct0 + ct_step - 1) / ct_step;
    const int rot = (HM_TILE_ROTATE && p.swizzle) ? (int)((unsigned)(rb * 5) % (unsigned)ntile) : 0;
    auto tile_at = [&](int t) { int q = t + rot; if (q >= ntile) q -= ntile; return ct0 + q * ct_step; };

    // ring prologue: tiles 0 .. DIST-1 in flight, tile 0 landed
    constexpr bool ASYNC_KEY = (BF && HM_DMA_GROUPED) || (!BF && HM_DMA_GROUPED_F32) || DIST == 1;   // the running argmin key is re-read behind the ring's own wait
    constexpr bool ALWAYS = (BF && HM_DMA_GROUPED) || (!BF && HM_DMA_GROUPED_F32);    // the ring always holds DIST tiles in flight (a repeat of the last
                                                      // tile goes into the free slot when the chunk runs out): no branches
#pragma unroll
    for (int q = 0; q < DIST; ++q)
        if (ALWAYS || q < ntile) dma_tile(tile_at(q < ntile ? q : ntile - 1), q);
    wait_tiles_in_flight(ALWAYS ? DIST - 1 : (ntile < DIST ? ntile : DIST) - 1);
#if HM_SETTLE_PROLOGUE
    // The compiler waits for its own loads (the A fragments above) lazily, at their first use INSIDE the
    // loop -- with vmcnt(N) instructions that also wait for the ring's LDS-DMA (which it cannot see) on
    // every iteration.  A wait it can see, here, settles them before the loop is entered.
    if (DIST == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
#endif
    __syncthreads();

    int buf = 0;                     // ring slot of tile t


    for (int t = 0; t < ntile; ++t) {
        const int ct = tile_at(t);
        const bool has_next = (t + DIST < ntile);            // block-uniform: a tile to put in flight
        const int ct_next = has_next ? tile_at(t + DIST) : ct;
        int buf_next = buf + DIST;                           // slot of tile t + DIST = slot of tile t - 1:
        if (buf_next >= NBUF) buf_next -= NBUF;              // every wave left it at the previous barrier

        const int j0 = ct * COLS;
        const bool compute = wave_active && (j0 + COLS - 1 > i0w);

        // running best key, re-read every 8th tile by a load the compiler does not see (it would wait for it
        // with vmcnt(0) inside the MFMA loop and so drain the ring): the value is picked up behind this
        // iteration's own counted wait.  Issued ahead of the tile's DMA, so that wait covers it.
        static_assert(!ALWAYS || DIST <= 2, "the key load rides on the counted wait of a ring of <= 3 slots");
        if (ASYNC_KEY && MODE == HM_MODE_ARGMIN && (t & 7) == 0) {
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(gk_raw) : "v"(&p.ctr64[1]) : "memory");
            gk_pending = true;
        }
        if (ALWAYS) {
            dma_tile(ct_next, buf_next);
        } else if (!HM_DMA_INTERLEAVE || !compute || (!BF && NP - 1 < PPW)) {
            if (has_next) dma_tile(ct_next, buf_next);
        } else if (BF && NP < PPW) {
            // more pieces than k-steps: the first PPW - NP go out ahead of the MFMA loop
            if (has_next) {
#pragma unroll
                for (int q = NP; q < PPW; ++q) dma_piece(ct_next, buf_next, q * WPB + wave);
            }
        }
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
        const int j0s = j0 + sub * SCOLS;
        const bool compute_s = wave_active && (j0s + SCOLS - 1 > i0w);
        f32x16 acc[TM][TN];
        if (compute_s) {
            // running best key of the argmin search, refreshed every 8th tile only: hipcc waits for
            // this vector load with vmcnt(0), which also drains the LDS-DMA ring.  A stale key only
            // emits a few more entries.
            if (MODE == HM_MODE_ARGMIN && !ASYNC_KEY && (t & 7) == 0)
                gk = __hip_atomic_load(&p.ctr64[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.0f;

            if constexpr (BF) {
                // bf16 prefilter: S = sum over the spatial coordinates, 16 per MFMA, fp32 accumulate
                const char* bt = smem + buf * TILE_LDS + (sub * SCOLS + r) * RB16 + 16 * h;
                uint4 bc[TN], bn[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bc[tn] = *reinterpret_cast<const uint4*>(bt + 32 * tn * RB16);
#pragma unroll
                for (int g = 0; g < NP; ++g) {
                    if (g + 1 < NP) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            if (HM_DIAG_NO_LDS) { bn[tn] = bc[tn]; bn[tn].x += g; }
                            else bn[tn] = *reinterpret_cast<const uint4*>(bt + 32 * tn * RB16 + 32 * (g + 1));
                    }
                    if (!ALWAYS && HM_DMA_INTERLEAVE && g < PPW && g < NP && has_next) dma_piece(ct_next, buf_next, g * WPB + wave);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            if (HM_DIAG_NO_MFMA) acc[tm][tn][g] += __builtin_bit_cast(float, bc[tn].x ^ a16[tm][g].y);
                            else
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[tm][g]),
                                                                                 __builtin_bit_cast(bf16x8, bc[tn]), acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bc[tn] = bn[tn];
                }
                // acc = S - x0*y0 (= -M): the time product came out of the last k-step's split slots
            } else {
                static_assert(BF || TN == 2, "the fp32 form is written for two 32-column tiles per streamed tile");
                // B fragments (optionally fetched one k-group ahead of the MFMAs that consume them)
                const float* bt = reinterpret_cast<const float*>(smem + buf * TILE_LDS) + (sub * SCOLS + r) * RS + 2 * h;
#if HM_PREFETCH_B
                float2 b0 = *reinterpret_cast<const float2*>(bt);
                float2 b1 = *reinterpret_cast<const float2*>(bt + 32 * RS);
#endif
#pragma unroll
                for (int g = 0; g < NP; ++g) {
#if HM_PREFETCH_B
                    float2 n0 = b0, n1 = b1;
                    if (g + 1 < NP) {
                        n0 = *reinterpret_cast<const float2*>(bt + 4 * (g + 1 < NG ? g + 1 : TCH));
                        n1 = *reinterpret_cast<const float2*>(bt + 32 * RS + 4 * (g + 1 < NG ? g + 1 : TCH));
                    }
#else
                    const float2 b0 = *reinterpret_cast<const float2*>(bt + 4 * (g < NG ? g : TCH));
                    const float2 b1 = *reinterpret_cast<const float2*>(bt + 32 * RS + 4 * (g < NG ? g : TCH));
#endif
#if HM_DMA_INTERLEAVE
                    // next tile's LDS-DMA pieces are issued between the MFMAs of the first k-groups: their
                    // issue slots hide behind the 64-cycle matrix instructions
                    if (!ALWAYS && NP - 1 >= PPW && g >= 1 && g - 1 < PPW && has_next) dma_piece(ct_next, buf_next, (g - 1) * WPB + wave);
#endif
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].x, b0.x, acc[tm][0], 0, 0, 0);
                        acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].x, b1.x, acc[tm][1], 0, 0, 0);
                    }
                    if (g < NG) {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].y, b0.y, acc[tm][0], 0, 0, 0);
                            acc[tm][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].y, b1.y, acc[tm][1], 0, 0, 0);
                        }
                    }
#if HM_PREFETCH_B
                    b0 = n0;
                    b1 = n1;
#endif
                }
            }
            // acc = S - x0*y0 = -M.  u = -M (reference sign) = acc;  u = +M (lorentz) = -acc.

            // ---- fast check: the lane's most promising u against the current bound ----
            float ext = acc[0][0][0];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ext = SIGN ? __builtin_fmaxf(ext, acc[tm][tn][e]) : __builtin_fminf(ext, acc[tm][tn][e]);
            const float ext_u = SIGN ? -ext : ext;

            float bound_f = pre_f;
            uint32_t best_bits = 0xffffffffu, best_low = 0xffffffffu;
            if (MODE == HM_MODE_ARGMIN) {
                // running best key over everything published so far (requested before the MFMA loop so
                // that its latency is hidden); an entry can only order before it if its u_f is within
                // 2*delta (+ a few ulps of acosh wiggle) of the best u_f
                best_bits = (uint32_t)(gk >> 32);
                best_low = (uint32_t)gk;
                if (best_bits != 0xffffffffu) {
                    const float bb = hm::bitsf(best_bits + HM_TIE_SLACK) + 2.0f * delta;
                    if (bb < bound_f) bound_f = bb;
                }
            }

            float diag_sum = 0.0f;
            if (HM_DIAG_NO_EPI && BF) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) diag_sum += acc[tm][tn][0];
            }
            if (HM_DIAG_NO_EPI && BF ? (diag_sum == 12345.0f) : (__ballot(ext_u < bound_f) != 0ull && !(HM_DIAG_NEVER_SLOW && p.n > 0))) {
                const bool full = rows_full && (j0s > i0w + WAVE_ROWS - 1) && (j0s + SCOLS - 1 < p.n);
#if HM_DIAG_TIMES
                ++diag_slow;
                const unsigned long long diag_s0 = __builtin_amdgcn_s_memrealtime();
#endif
                // -------- slow path, one 32x32 MFMA tile at a time (a rolled loop: the hot loop must not
                // inherit its register pressure).  Per-element predicates are evaluated twice (count, then
                // write); the second evaluation runs on laundered copies of the bounds so that the compiler
                // does not keep the predicates alive across the wave scan.
                unsigned long long wkey = ~0ull;
                bool wrote = false;
#pragma unroll 1
                for (int st = 0; st < (HM_DIAG_NO_SLOW ? 0 : TM * TN); ++st) {
                    f32x16 w = acc[0][0];
#pragma unroll
                    for (int q = 1; q < TM * TN; ++q)
                        if (st == q) w = acc[q / TN][q % TN];
                    const int tm = st / TN, tn = st - tm * TN;
                    if (TM * TN > 1) {
                        float e1 = w[0];
#pragma unroll
                        for (int e = 1; e < 16; ++e) e1 = SIGN ? __builtin_fmaxf(e1, w[e]) : __builtin_fminf(e1, w[e]);
                        if (__ballot((SIGN ? -e1 : e1) < bound_f) == 0ull) continue;
                    }
                    uint32_t n_emit = 0, n_sure = 0;
                    float bnd = bound_f;
                    float cutv = cut_f;
                    uint32_t slot = 0;
                    const int ib = i0w + 32 * tm + 4 * h;
                    const int j = j0s + 32 * tn + r;
                    auto visit = [&](const float wv, const int e, const bool write) {
                        const float u = SIGN ? -wv : wv;
                        const int i = ib + (e & 3) + 8 * (e >> 2);
                        bool pass = u < bnd;
                        if (!full) pass = pass && (i < j) && (j < p.n) && (i >= p.row_begin) && (i < p.row_end);
                        if (!pass) return;
                        const float up = u < 1.0f ? 1.0f : u;
                        const uint32_t ub = hm::fbits(up);
                        if (MODE == HM_MODE_HIST) {
                            if (ub >= p.hist_lo) {
                                uint32_t bin = (ub - p.hist_lo) >> p.hist_shift;
                                if (bin > HM_HIST_BINS - 1) bin = HM_HIST_BINS - 1;
                                atomicAdd(&lhist[bin], 1u);
                            }
                            return;
                        }
                        bool emit;
                        uint32_t flag = 0;
                        const bool zero = (u <= zmax_f) && (p.thr_pos != 0);     // certainly d == 0 < thr
                        if (MODE == HM_MODE_TOPK) {
                            const bool sure = zero || (up < lo_f);
                            if (sure && !write) ++n_sure;
                            flag = sure ? 1u : 0u;
                            // zero-distance ties order by (i, j): a tie flood is cut by rows (tie_imax)
                            emit = zero ? (i <= p.tie_imax) : (!sure || cut_all || up <= cutv);
                        } else {
                            const uint32_t ubz = zero ? 0x3f7fffffu : ub;
                            const uint32_t low = ((uint32_t)i << 15) | ((uint32_t)j >> 2);
                            emit = !(zero && best_bits == 0x3f7fffffu) || (low <= best_low);
                            if (emit && !write) {
                                const unsigned long long k = ((unsigned long long)ubz << 32) | low;
                                wkey = k < wkey ? k : wkey;
                            }
                        }
                        if (!emit) return;
                        if (!write) { ++n_emit; return; }
                        if (slot < p.ent_cap) p.ent[slot] = make_uint4(ub, (uint32_t)i, (uint32_t)j, flag);
                        ++slot;
                    };
#pragma unroll
                    for (int e = 0; e < 16; ++e) visit(w[e], e, false);
                    if (MODE == HM_MODE_HIST) continue;
                    sure_total += n_sure;
                    const uint32_t incl = hm_wave_incl_scan(n_emit, lane);
                    const uint32_t total = __shfl(incl, 63, 64);
                    if (total == 0) continue;
                    uint32_t base = 0;
                    if (lane == 63) base = atomicAdd(&p.ctr[0], total);
                    base = __shfl(base, 63, 64);
                    slot = base + incl - n_emit;
                    asm volatile("" : "+v"(bnd), "+v"(cutv));      // opaque: no CSE with the count pass
#pragma unroll
                    for (int e = 0; e < 16; ++e) visit(w[e], e, true);
                    wrote = true;
                }
                if (MODE == HM_MODE_ARGMIN && wrote) {
                    const unsigned long long wk = hm_wave_min_u64(wkey);
                    if (lane == 0) atomicMin(&p.ctr64[1], wk);
                    // the slow path has drained the vector-memory queue anyway: refresh the running key
                    // now (cheap here) so that the next tiles see the tight bound
                    gk = __hip_atomic_load(&p.ctr64[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wk < gk) gk = wk;
                }
#if HM_DIAG_TIMES
                diag_slow_ticks += __builtin_amdgcn_s_memrealtime() - diag_s0;
#endif
            }
        }
        }
#if HM_DIAG_TIMES
        ++diag_tiles;
#endif
        if (ALWAYS) {
            // tile t+1 has landed (this wave's pieces) -- and so has the key load, if one was issued
            if (DIST - 1 <= 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gk_raw) : : "memory");
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gk_raw) : "n"((DIST - 1) * PPW) : "memory");
        } else {   // tile t+1 has landed (this wave's pieces): tiles t+2 .. min(t+DIST, ntile-1) may stay in flight
            const int last = (t + DIST < ntile - 1) ? t + DIST : ntile - 1;
            wait_tiles_in_flight(last - (t + 1));
            if (ASYNC_KEY) asm volatile("" : "+v"(gk_raw) : : "memory");   // DIST == 1: that wait was vmcnt(0)
        }
        if (ASYNC_KEY && MODE == HM_MODE_ARGMIN && gk_pending) {
            if (gk_raw < gk) gk = gk_raw;                   // the key only ever decreases
            gk_pending = false;
        }
        if (DYN && next_pending) {                           // wave 0, once per chunk: the fetch has returned (vmcnt(0) above)
            static_assert(!DYN || DIST == 1, "the counter fetch rides on the vmcnt(0) of a two-slot ring");
            asm volatile("" : "+v"(next_raw) : : "memory");
            if (lane == 0) *s_next = (uint32_t)gridDim.x + next_raw;
            next_pending = false;
        }
        if (!(HM_DIAG_NO_BARRIER && BF)) __syncthreads();   // ... and every wave's; all reads of slot `buf` done
        if (++buf == NBUF) buf = 0;
    }
    } while (PERSIST && pos < pos_end);
    if (!DYN) break;
    chunk = __builtin_amdgcn_readfirstlane(*s_next);        // written before a barrier every wave has passed
    }
#if HM_DIAG_TIMES
    if (MODE == HM_MODE_ARGMIN && lane == 0 && (blockIdx.x & 1) == 0 && blockIdx.x / 2 < HM_DIGIT_BINS / 16) {
        uint32_t* o = p.hist + (blockIdx.x / 2) * 16 + wave * 4;
        o[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - diag_t0);
        o[1] = wave == 1 ? (uint32_t)diag_t0 : diag_slow; o[2] = (uint32_t)diag_slow_ticks; o[3] = (diag_pass << 16) | diag_tiles;
    }
#endif

    if (MODE == HM_MODE_TOPK) {
        // one 64-bit atomic per wave for the sure count
        unsigned long long s = sure_total;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0 && s != 0) atomicAdd(&p.ctr64[0], s);
    }
    if (MODE == HM_MODE_HIST) {
        __syncthreads();
        for (int t = threadIdx.x; t < HM_HIST_BINS; t += NTHREADS)
            if (lhist[t]) atomicAdd(&p.hist[t], lhist[t]);
    }
}

// ------------------------------------------------------------------------------------------------
// post-processing of emitted entries
// ------------------------------------------------------------------------------------------------
struct ArgminRec { uint32_t found, dbits, i, j; };

__device__ __forceinline__ bool hm_key_less(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2)
{
    if (a0 != b0) return a0 < b0;
    if (a1 != b1) return a1 < b1;
    return a2 < b2;
}

// canonical distance of every entry, threshold, lexicographic min of (dbits, i, j).
// Stage 1: HM_ARGMIN_BLOCKS blocks, one partial record each; stage 2: one block over the partials.
#define HM_ARGMIN_BLOCKS 256
#define HM_POSTD_WAVE_ENTRIES 65536u  // ... and the top-k post kernel up to this many
#define HM_POST_WAVE_ENTRIES 4096u   // up to this many emitted entries the argmin post kernel works one wave per entry
struct ArgminPart { uint32_t dbits, i, j, pad; };

__device__ __forceinline__ void hm_block_min_key(uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t* s0, uint32_t* s1, uint32_t* s2)
{
    const int t = threadIdx.x;
    s0[t] = b0; s1[t] = b1; s2[t] = b2;
    __syncthreads();
    for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if (t < off) {
            const int o = t + off;
            if (hm_key_less(s0[o], s1[o], s2[o], s0[t], s1[t], s2[t])) { s0[t] = s0[o]; s1[t] = s1[o]; s2[t] = s2[o]; }
        }
        __syncthreads();
    }
    b0 = s0[0]; b1 = s1[0]; b2 = s2[0];
}

__global__ __launch_bounds__(256) void hm_post_argmin_kernel(const uint4* __restrict__ ent, const uint32_t* __restrict__ ctr,
                                                             uint32_t cap, const float* __restrict__ img, int RS, int d,
                                                             int sign_mode, float sqrt_c, float thr, ArgminPart* __restrict__ parts)
{
    __shared__ uint32_t s0[256], s1[256], s2[256];
    __shared__ float sp[4][HM_MAX_D1];
    uint32_t m = ctr[0];
    if (m > cap) m = cap;
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu;
    if (m <= HM_POST_WAVE_ENTRIES) {
        // the usual case (a few hundred survivors): one WAVE per entry -- the d products are formed by the lanes
        // (coalesced row reads) and summed by lane 0 in the canonical order; the step waits on this kernel's latency
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (uint32_t t = blockIdx.x * 4 + wv; t < m; t += gridDim.x * 4) {
            const uint4 en = ent[t];
            const float* ra = img + (int64_t)en.y * RS;
            const float* rb = img + (int64_t)en.z * RS;
            for (int k = lane; k < d; k += 64) {
                const int o = 4 * (k >> 2) + hm_pos_in_group(k & 3);
                sp[wv][k] = ra[o] * rb[o];
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (lane == 0) {
                const float S = hm::torch_order_sum([&](int q) { return sp[wv][q]; }, d);
                const float tt = ra[RS - 4] * rb[RS - 4];
                const float mm = tt - S;
                const float dd = hm::dist_from_u(sign_mode ? mm : -mm, sqrt_c);
                if (dd < thr) {
                    const uint32_t db = hm::fbits(dd);
                    if (hm_key_less(db, en.y, en.z, b0, b1, b2)) { b0 = db; b1 = en.y; b2 = en.z; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < m; t += gridDim.x * 256) {
            const uint4 en = ent[t];
            const float dd = hm::dist_from_u(hm_img_u(img, RS, d, en.y, en.z, sign_mode), sqrt_c);
            if (dd < thr) {
                const uint32_t db = hm::fbits(dd);
                if (hm_key_less(db, en.y, en.z, b0, b1, b2)) { b0 = db; b1 = en.y; b2 = en.z; }
            }
        }
    }
    hm_block_min_key(b0, b1, b2, s0, s1, s2);
    if (threadIdx.x == 0) { parts[blockIdx.x].dbits = b0; parts[blockIdx.x].i = b1; parts[blockIdx.x].j = b2; parts[blockIdx.x].pad = 0; }
}

// seed == nullptr: no seed update (hm_row_argmin).  Otherwise a final record (found == 1) becomes the next
// search's seed: key = (bits(u_c + delta), all ones) -- the pair's own prefilter value is <= u_c + delta, and
// an entry the scan skips on this key has u_f >= u_c + 3 delta, so it cannot order before the pair -- or,
// for a pair at distance 0 (u_c <= 1), the exact zero-class key with its (i, j) part.
__global__ __launch_bounds__(HM_ARGMIN_BLOCKS) void hm_post_argmin_final_kernel(const ArgminPart* __restrict__ parts, ArgminRec* out,
                                                                                const uint32_t* __restrict__ ctr, uint32_t cap,
                                                                                ArgminSeed* seed, const float* __restrict__ img, int RS, int d,
                                                                                int sign_mode, int bf, int kterms,
                                                                                const uint32_t* __restrict__ rmax2_bits, uint32_t* emitted_out,
                                                                                uint32_t* arm_ctr, unsigned long long* arm_ctr64, int arm_rb, int arm_re)
{
    __shared__ uint32_t s0[HM_ARGMIN_BLOCKS], s1[HM_ARGMIN_BLOCKS], s2[HM_ARGMIN_BLOCKS];
    uint32_t b0 = parts[threadIdx.x].dbits, b1 = parts[threadIdx.x].i, b2 = parts[threadIdx.x].j;
    hm_block_min_key(b0, b1, b2, s0, s1, s2);
    if (threadIdx.x == 0) {
        // found = 2: the emission buffer overflowed, the record is not final (caller reruns bounded)
        const uint32_t found = ctr[0] > cap ? 2u : ((b1 != 0xffffffffu) ? 1u : 0u);
        out->found = found;
        out->dbits = b0; out->i = b1; out->j = b2;
        if (emitted_out != nullptr) *emitted_out = ctr[0];      // rides back to the host with the record (one copy)
        if (seed != nullptr && found == 1u) {
            const float u = hm_img_u(img, RS, d, b1, b2, sign_mode);
            unsigned long long key;
            if (u <= 1.0f) key = (0x3f7fffffull << 32) | (unsigned long long)((b1 << 15) | (b2 >> 2));
            else key = ((unsigned long long)hm::fbits(u + hm_scan_delta(bf != 0, kterms, rmax2_bits)) << 32) | 0xffffffffull;
            seed->key = key; seed->i = b1; seed->valid = 1u;
        }
        // arm the next search over the same row range: counters cleared, running key = seed (what hm_seed_init_kernel
        // would do at its start) -- the merge loop then goes from the merge kernel straight into the scan
        if (arm_ctr != nullptr && found != 2u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) arm_ctr[q] = 0u;
            const bool use = seed != nullptr && seed->valid != 0u && (int)seed->i >= arm_rb && (int)seed->i < arm_re;
            arm_ctr64[0] = ~0ull;
            arm_ctr64[1] = use ? seed->key : ~0ull;
        }
    }
}

// K3 fused with a min-reduction: canonical distance from image row `row` to every row i < n_partners,
// smallest (d bits, i) with d < thr per block -> parts (pair = (i, row), i < row).
__global__ __launch_bounds__(256) void hm_row_argmin_kernel(const float* __restrict__ img, int RS, int d, int sign_mode, int64_t row,
                                                            int64_t n_partners, float sqrt_c, float thr, ArgminPart* __restrict__ parts)
{
    __shared__ uint32_t s0[256], s1[256], s2[256];
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_partners; i += (int64_t)gridDim.x * 256) {
        if (i == row) continue;
        const int64_t lo = i < row ? i : row, hi = i < row ? row : i;
        const float dd = hm::dist_from_u(hm_img_u(img, RS, d, lo, hi, sign_mode), sqrt_c);
        if (dd < thr) {
            const uint32_t db = hm::fbits(dd);
            if (hm_key_less(db, (uint32_t)lo, (uint32_t)hi, b0, b1, b2)) { b0 = db; b1 = (uint32_t)lo; b2 = (uint32_t)hi; }
        }
    }
    hm_block_min_key(b0, b1, b2, s0, s1, s2);
    if (threadIdx.x == 0) { parts[blockIdx.x].dbits = b0; parts[blockIdx.x].i = b1; parts[blockIdx.x].j = b2; parts[blockIdx.x].pad = 0; }
}

// entries {bits(u_f'), i, j, sure} -> {dbits | 0xffffffff, i, j, bits(u_c')} with the canonical
// distance; counts[0] valid, counts[1] valid & !sure, counts[3] sure & !valid (margin violated: must be 0)
__global__ void hm_post_distance_kernel(uint4* __restrict__ ent, const uint32_t* __restrict__ ctr, uint32_t cap,
                                        const float* __restrict__ img, int RS, int d, int sign_mode, float sqrt_c, float thr,
                                        uint32_t* __restrict__ counts)
{
    __shared__ float sp[4][HM_MAX_D1];
    uint32_t m = ctr[0];
    if (m > cap) m = cap;
    uint32_t nv = 0, nb = 0, bad = 0;
    if (m <= HM_POSTD_WAVE_ENTRIES && blockDim.x == 256) {
        // the usual refresh (~2 * cache_size survivors): one wave per entry -- coalesced row reads, products by the
        // lanes, canonical sum on lane 0 (a thread per entry reads two scattered rows with every lane)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (uint32_t t = blockIdx.x * 4 + wv; t < m; t += gridDim.x * 4) {
            const uint4 en = ent[t];
            const float* ra = img + (int64_t)en.y * RS;
            const float* rb = img + (int64_t)en.z * RS;
            for (int k = lane; k < d; k += 64) {
                const int o = 4 * (k >> 2) + hm_pos_in_group(k & 3);
                sp[wv][k] = ra[o] * rb[o];
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (lane == 0) {
                const float S = hm::torch_order_sum([&](int q) { return sp[wv][q]; }, d);
                const float tt = ra[RS - 4] * rb[RS - 4];
                const float mm = tt - S;
                const float uc = hm::clamp_min_one(sign_mode ? mm : -mm);
                const float dd = hm::acosh_c(uc) / sqrt_c;
                const bool valid = dd < thr;
                nv += valid ? 1u : 0u;
                nb += (valid && en.w == 0u) ? 1u : 0u;
                bad += (!valid && en.w != 0u) ? 1u : 0u;
                ent[t] = make_uint4(valid ? hm::fbits(dd) : 0xffffffffu, en.y, en.z, hm::fbits(uc));
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
            const uint4 en = ent[t];
            const float uc = hm::clamp_min_one(hm_img_u(img, RS, d, en.y, en.z, sign_mode));
            const float dd = hm::acosh_c(uc) / sqrt_c;
            const bool valid = dd < thr;
            nv += valid ? 1u : 0u;
            nb += (valid && en.w == 0u) ? 1u : 0u;
            bad += (!valid && en.w != 0u) ? 1u : 0u;
            ent[t] = make_uint4(valid ? hm::fbits(dd) : 0xffffffffu, en.y, en.z, hm::fbits(uc));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nv += __shfl_xor(nv, off, 64); nb += __shfl_xor(nb, off, 64); bad += __shfl_xor(bad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (nv) atomicAdd(&counts[0], nv);
        if (nb) atomicAdd(&counts[1], nb);
        if (bad) atomicAdd(&counts[3], bad);
    }
}

// radix narrowing: digit `level` (0..7) of the 96-bit key (x: 12|12|8, y: 12|12|8 ... see hm_digit)
struct Prefix { uint32_t val[3]; uint32_t mask[3]; };

__device__ __host__ __forceinline__ void hm_digit_pos(int level, int& word, int& shift, int& bits)
{
    word = level / 3;
    const int q = level % 3;
    shift = q == 0 ? 20 : (q == 1 ? 8 : 0);
    bits = q == 2 ? 8 : 12;
}

__global__ void hm_digit_hist_kernel(const uint4* __restrict__ ent, uint32_t m, Prefix pf, int level, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[HM_DIGIT_BINS];
    for (int t = threadIdx.x; t < HM_DIGIT_BINS; t += blockDim.x) lh[t] = 0;
    __syncthreads();
    int word, shift, bits;
    hm_digit_pos(level, word, shift, bits);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        const uint4 en = ent[t];
        const uint32_t k[3] = {en.x, en.y, en.z};
        if ((k[0] & pf.mask[0]) == pf.val[0] && (k[1] & pf.mask[1]) == pf.val[1] && (k[2] & pf.mask[2]) == pf.val[2])
            atomicAdd(&lh[(k[word] >> shift) & ((1u << bits) - 1u)], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < HM_DIGIT_BINS; t += blockDim.x)
        if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

// keep entries whose masked key <= prefix (lexicographic)
__global__ void hm_compact_prefix_kernel(const uint4* __restrict__ ent, uint32_t m, Prefix pf, uint4* __restrict__ out,
                                         uint32_t* __restrict__ out_count, uint32_t out_cap)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        const uint4 en = ent[t];
        const uint32_t a0 = en.x & pf.mask[0], a1 = en.y & pf.mask[1], a2 = en.z & pf.mask[2];
        const bool le = !hm_key_less(pf.val[0], pf.val[1], pf.val[2], a0, a1, a2);
        if (le) {
            const uint32_t s = atomicAdd(out_count, 1u);
            if (s < out_cap) out[s] = en;
        }
    }
}

// keep valid entries only (candidate listing)
__global__ void hm_compact_valid_kernel(const uint4* __restrict__ ent, uint32_t m, uint4* __restrict__ out,
                                        uint32_t* __restrict__ out_count, uint32_t out_cap)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        const uint4 en = ent[t];
        if (en.x != 0xffffffffu) {
            const uint32_t s = atomicAdd(out_count, 1u);
            if (s < out_cap) out[s] = en;
        }
    }
}

// exact rank of every key among m unique keys; rank < k is written to out[rank].  HM_RANK_SPLIT lanes share one
// key, each counting a residue class of the staged tile (the sort sits on the refresh's critical path and m is
// only ~2 * cache_size: one thread per key would leave most of the chip idle).
#define HM_RANK_SPLIT 8
__global__ __launch_bounds__(256) void hm_rank_sort_kernel(const uint4* __restrict__ ent, uint32_t m, uint4* __restrict__ out,
                                                           uint32_t k)
{
    __shared__ uint4 tile[1024];
    constexpr uint32_t KEYS = 256 / HM_RANK_SPLIT;              // keys per block
    const uint32_t part = threadIdx.x % HM_RANK_SPLIT;
    const uint32_t t = blockIdx.x * KEYS + threadIdx.x / HM_RANK_SPLIT;
    uint4 me = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
    if (t < m) me = ent[t];
    uint32_t rank = 0;
    for (uint32_t base = 0; base < m; base += 1024) {
        __syncthreads();
        for (int q = threadIdx.x; q < 1024; q += 256) {
            const uint32_t idx = base + q;
            tile[q] = idx < m ? ent[idx] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
        }
        __syncthreads();
        const uint32_t lim = (m - base) < 1024u ? (m - base) : 1024u;
        for (uint32_t q = part; q < lim; q += HM_RANK_SPLIT) {
            const uint4 o = tile[q];
            rank += hm_key_less(o.x, o.y, o.z, me.x, me.y, me.z) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 1; off < HM_RANK_SPLIT; off <<= 1) rank += __shfl_xor(rank, off, 64);
    if (part == 0 && t < m && rank < k) out[rank] = me;
}

// ------------------------------------------------------------------------------------------------
// gathered / dense kernels on the image
// ------------------------------------------------------------------------------------------------
__global__ void hm_pairdist_kernel(const float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                   const int32_t* __restrict__ J, int64_t b, float sqrt_c, int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    out[t] = hm::dist_from_u(hm_img_u(img, RS, d, I[t], J[t], sign_mode), sqrt_c);
}

__global__ void hm_rowvsall_kernel(const float* __restrict__ img, int RS, int d, int64_t row, int64_t n, float sqrt_c,
                                   int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[t] = hm::dist_from_u(hm_img_u(img, RS, d, row, t, sign_mode), sqrt_c);
}

// midpoint of (x, y) given as accessor lambdas; writes d1 values through `put`
template <class GX, class GY, class PUT>
__device__ __forceinline__ void hm_midpoint_core(int d, float w, float c, int sign_mode, GX gx, GY gy, PUT put, float* scratch)
{
    // log_map (embedding/lorentz_model.py:96-119)
    const float S = hm::torch_order_sum([&](int s) { return gx(1 + s) * gy(1 + s); }, d);
    const float t0 = gx(0) * gy(0);
    const float mref = t0 - S;
    const float u = sign_mode ? mref : -mref;
    const float m = -u;
    const float a = hm::clamp_min_one(u);
    float coef = hm::acosh_c(a) / __builtin_sqrtf(a * a - 1.0f);
    if (coef == coef && coef > 1.0e4f) coef = 1.0e4f;
    // v = w * log ; exp_map (:73-93)
    for (int k = 0; k <= d; ++k) scratch[k] = (coef * (gy(k) + m * gx(k))) * w;
    float n2 = hm::torch_order_sum([&](int s) { return scratch[1 + s] * scratch[1 + s]; }, d);
    if (n2 == n2 && n2 < 1.0e-8f) n2 = 1.0e-8f;
    const float nn = __builtin_sqrtf(n2);
    const float ch = hm::cosh_c(nn), sh = hm::sinh_c(nn);
    // project (:41-56)
    float r2 = 0.0f;
    for (int k = 1; k <= d; ++k) {
        const float e = ch * gx(k) + sh * (scratch[k] / nn);
        scratch[k] = e;
        r2 = __builtin_fmaf(e, e, r2);
    }
    const float rr = __builtin_sqrtf(r2);
    put(0, __builtin_sqrtf(1.0f + (c * rr) * rr));
    for (int k = 1; k <= d; ++k) put(k, scratch[k]);
}


__global__ void hm_midpoint_kernel(const float* __restrict__ img, int RS, int d, const int32_t* __restrict__ I,
                                   const int32_t* __restrict__ J, const float* __restrict__ W, int64_t b, float c,
                                   int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    float scratch[HM_MAX_D1];
    const int64_t ri = I[t], rj = J[t];
    auto gx = [&](int k) { return k == 0 ? hm_img_time(img, RS, ri) : hm_img_spatial(img, RS, ri, k - 1); };
    auto gy = [&](int k) { return k == 0 ? hm_img_time(img, RS, rj) : hm_img_spatial(img, RS, rj, k - 1); };
    float* o = out + t * (d + 1);
    hm_midpoint_core(d, W[t], c, sign_mode, gx, gy, [&](int k, float v) { o[k] = v; }, scratch);
}

// fused merge: midpoint of image rows (i, j) -> table row and image row `new_row`.
// One wave: the two rows are staged into LDS with coalesced loads, lane 0 runs the (inherently
// sequential) canonical arithmetic on LDS operands, all lanes write the result.
__global__ __launch_bounds__(64) void hm_merge_append_kernel(float* __restrict__ img, int RS, int d, int NG, int32_t i, int32_t j,
                                                             float w, float c, int sign_mode, float* __restrict__ X, int64_t ld,
                                                             int64_t new_row, uint32_t* __restrict__ rmax2_bits,
                                                             unsigned char* __restrict__ img16, int KS)
{
    __shared__ float sx[HM_MAX_D1], sy[HM_MAX_D1], sv[HM_MAX_D1], so[HM_MAX_D1];
    const int lane = threadIdx.x;
    const int64_t ri = i, rj = j;
    for (int k = lane; k <= d; k += 64) {
        sx[k] = k == 0 ? hm_img_time(img, RS, ri) : hm_img_spatial(img, RS, ri, k - 1);
        sy[k] = k == 0 ? hm_img_time(img, RS, rj) : hm_img_spatial(img, RS, rj, k - 1);
    }
    __syncthreads();
    // Same operations, same order as hm_midpoint_core -- the element-wise steps spread over the wave, the three
    // reductions (canonical order) and the scalar transcendental steps on lane 0.
    __shared__ float sc[4];                      // coef, m, nn | ch, sh via sc[2], sc[3] after the second reduction
    for (int k = lane; k < d; k += 64) so[k] = sx[1 + k] * sy[1 + k];              // products, rounded separately
    __syncthreads();
    if (lane == 0) {
        const float S = hm::torch_order_sum([&](int t) { return so[t]; }, d);
        const float t0 = sx[0] * sy[0];
        const float mref = t0 - S;
        const float u = sign_mode ? mref : -mref;
        const float a = hm::clamp_min_one(u);
        float coef = hm::acosh_c(a) / __builtin_sqrtf(a * a - 1.0f);
        if (coef == coef && coef > 1.0e4f) coef = 1.0e4f;
        sc[0] = coef;
        sc[1] = -u;
    }
    __syncthreads();
    {
        const float coef = sc[0], m = sc[1];
        for (int k = lane; k <= d; k += 64) {
            const float v = (coef * (sy[k] + m * sx[k])) * w;                      // w * log_map
            sv[k] = v;
            if (k >= 1) so[k - 1] = v * v;
        }
    }
    __syncthreads();
    if (lane == 0) {
        float n2 = hm::torch_order_sum([&](int t) { return so[t]; }, d);
        if (n2 == n2 && n2 < 1.0e-8f) n2 = 1.0e-8f;
        const float nn = __builtin_sqrtf(n2);
        sc[1] = nn;
        sc[2] = hm::cosh_c(nn);
        sc[3] = hm::sinh_c(nn);
    }
    __syncthreads();
    {
        const float nn = sc[1], ch = sc[2], sh = sc[3];
        for (int k = 1 + lane; k <= d; k += 64) so[k] = ch * sx[k] + sh * (sv[k] / nn);   // exp_map, spatial part
    }
    __syncthreads();
    if (lane == 0) {
        float r2 = 0.0f;                                                               // project: sequential fmaf chain
        for (int k = 1; k <= d; ++k) r2 = __builtin_fmaf(so[k], so[k], r2);
        const float rr = __builtin_sqrtf(r2);
        const float x0 = __builtin_sqrtf(1.0f + (c * rr) * rr);
        so[0] = x0;
        float q2 = __builtin_fmaf(x0, x0, 0.0f);                                       // norm bounds of the new row
        for (int k = 1; k <= d; ++k) q2 = __builtin_fmaf(so[k], so[k], q2);
        if (q2 < 3.0e38f && q2 > 0.0f) {
            atomicMax(rmax2_bits, hm::fbits(q2));
            const float s2 = q2 - x0 * x0;
            if (s2 > 0.0f) atomicMax(rmax2_bits + 1, hm::fbits(s2 * 1.0001f));
        }
    }
    __syncthreads();
    float* xr = X + new_row * ld;
    float* ir = img + new_row * RS;
    for (int k = lane; k <= d; k += 64) {
        const float v = so[k];
        xr[k] = v;
        if (k == 0) ir[RS - 4] = v;
        else ir[4 * ((k - 1) >> 2) + hm_pos_in_group((k - 1) & 3)] = v;
    }
    // bf16 image row
    const int CH = 2 * KS + 1;
    for (int cidx = lane; cidx < CH; cidx += 64) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cidx == CH - 1) v.x = hm::fbits(so[0]);
        else v = hm_bf16_chunk(so + 1, so[0], d, KS, cidx);
        *reinterpret_cast<uint4*>(img16 + ((int64_t)new_row * CH + cidx) * 16) = v;
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

__global__ void hm_dense_kernel(const float* __restrict__ X, int64_t n1, const float* __restrict__ Y, int64_t n2, int64_t ldx,
                                int64_t ldy, int d1, float sqrt_c, int sign_mode, float* __restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n2 || i >= n1) return;
    out[i * n2 + j] = hm::dist_from_u(hm_rm_u(X + i * ldx, Y + j * ldy, d1, sign_mode), sqrt_c);
}

__global__ void hm_rows_minkowski_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t b, int64_t ld, int d1,
                                         int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    // minkowski_dot under the active convention = -u
    out[t] = -hm_rm_u(x + t * ld, y + t * ld, d1, sign_mode);
}

__global__ void hm_rows_distance_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t b, int64_t ld, int d1,
                                        float sqrt_c, int sign_mode, float* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    out[t] = hm::dist_from_u(hm_rm_u(x + t * ld, y + t * ld, d1, sign_mode), sqrt_c);
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
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

struct HostCtl {                 // pinned host mirror of small device results
    uint32_t ctr[8];             // [0] emitted [1] valid [2] valid & !sure [3] compacted [4] margin violations
    unsigned long long ctr64[2]; // [0] sure count [1] argmin key
    ArgminRec rec;
    ArgminRec rec2[2];           // [0] record, [1].found = emitted count (argmin: one device-to-host copy)
    uint32_t hist[HM_DIGIT_BINS];
};

struct hm_engine {
    int device = 0;
    int n_cu = 256;                       // compute units of the device (resident-block count of the persistent scan)
    // work-decomposition knobs (defaults from the macros; HM_TUNE_* environment overrides are a tuning aid)
    int chunk_f32 = HM_CHUNK_TILES, chunk_bf16 = HM_CHUNK_TILES_BF16, tail_div = 4;
    double tail_fraction = HM_TAIL_FRACTION;
    int64_t tm4_min_rows = HM_TM4_MIN_ROWS;       // bf16 form: tables at least this large use 512-row blocks
    int dyn_k1 = HM_DYN_K1, dyn_k2 = HM_DYN_K2;   // HM_PERSIST == 2: tiles per chunk, early / late part of the sequence
    double dyn_split = HM_DYN_SPLIT;              // share of the sequence handed out in the larger chunks
    int64_t max_rows = 0, rows_alloc = 0, n = 0;
    int d1 = 0, d = 0, NG = 0, RS = 0, sign_mode = 0;
    float* img = nullptr;
    unsigned char* img16 = nullptr;       // bf16 image for the bf16 prefilter form
    int KS = 0, RB16 = 0;                 // k-steps of 16 and bytes per bf16 image row
    int precision = 0;                    // 0 = auto, 1 = fp32 prefilter, 2 = bf16 prefilter
    bool bf16_ok = true;                  // false: no bf16 image for this width (d > 124); the fp32 form is used whatever is asked
    uint4* ent = nullptr;
    uint4* ent2 = nullptr;
    uint4* sorted = nullptr;
    uint32_t ent_cap = 0;
    uint32_t* d_ctr = nullptr;            // 8 x u32
    uint32_t* d_rmax2 = nullptr;          // float bits: [0] largest squared row norm, [1] largest squared spatial norm
    unsigned long long* d_ctr64 = nullptr; // 2 x u64
    ArgminRec* d_rec = nullptr;
    float topk_f32_thr = 0.0f;            // > 0: top-k searches with a threshold at least this large needed the fp32 form on this table
    bool force_f32 = false;               // set for the duration of one call: use the fp32 form whatever the default is
    bool armed = false;                   // the last argmin search left counters + running key ready for a search of
    int64_t armed_rb = 0, armed_re = 0;   // rows [armed_rb, armed_re) as requested (cleared by every other entry point that uses them)
    ArgminSeed* d_seed = nullptr;         // running-key seed of the next argmin search (device-resident state)
    ArgminPart* d_parts = nullptr;
    uint32_t* d_hist = nullptr;           // HM_DIGIT_BINS
    HostCtl* h = nullptr;                 // pinned
    uint4* h_sorted = nullptr;            // pinned, sorted_cap entries
    uint32_t sorted_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // cut prediction for top-k: valid while rows are only appended
    bool have_cut = false;
    uint32_t last_cut_bits = 0;
    int64_t last_cut_k = 0;
    float last_cut_c = 0.f;
    // stats
    float last_scan_ms = 0.f;
    int64_t last_pairs = 0, last_emitted = 0;
    int last_passes = 0;
    bool pending_timing = false;   // ev0/ev1 recorded by an asynchronous call, not read yet
    int64_t pending_pairs = 0;
    double tot_scan_ms = 0.0;
    int64_t tot_pairs = 0, tot_launches = 0;
    std::string err;
};

static int hm_fail(hm_engine* e, int code, const std::string& msg)
{
    g_last_error = msg;
    if (e) e->err = msg;
    return code;
}

#define HM_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t _st = (call);                                                                      \
        if (_st != hipSuccess)                                                                        \
            return hm_fail(e, (int)_st, std::string(#call) + ": " + hipGetErrorString(_st));          \
    } while (0)

static const int kSupportedNG[] = {1, 2, 3, 4, 6, 8, 10, 13, 16, 20, 25, 28, 32};

static const int kSupportedKS[] = {1, 2, 4, 7, 8};      // bf16 form: 16 spatial coordinates per k-step
static int hm_pick_ks(int d)
{
    const int need = (d + 4 + 15) / 16;        // d spatial slots + 4 slots for the split time coordinate
    for (int v : kSupportedKS)
        if (v >= need) return v;
    return -1;
}

static int hm_pick_ng(int d)
{
    const int need = (d + 3) / 4;
    for (int v : kSupportedNG)
        if (v >= need) return v;
    return -1;
}

extern "C" int hm_abi_version(void) { return HM_ABI_VERSION; }

extern "C" const char* hm_last_error(const hm_engine* e)
{
    if (e && !e->err.empty()) return e->err.c_str();
    return g_last_error.c_str();
}

extern "C" int hm_engine_create(hm_engine** out, int device, int64_t max_rows, int d1, int sign_mode)
{
    hm_engine* e = nullptr;
    if (!out) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: out is NULL");
    *out = nullptr;
    if (d1 < 2 || d1 > 129) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: d1 must be in [2, 129]");
    if (max_rows < 2 || max_rows > 131072)
        return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: max_rows must be in [2, 131072]");
    if (sign_mode != 0 && sign_mode != 1) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: sign_mode must be 0 or 1");
    int ndev = 0;
    hipError_t st = hipGetDeviceCount(&ndev);
    if (st != hipSuccess || ndev <= 0)
        return hm_fail(nullptr, st != hipSuccess ? (int)st : (int)hipErrorNoDevice,
                       "hm_engine_create: no HIP device available (the merge engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: bad device index");
    e = new hm_engine();
    e->device = device;
    e->max_rows = max_rows;
    e->d1 = d1;
    e->d = d1 - 1;
    e->NG = hm_pick_ng(e->d);
    e->RS = hm_row_floats(e->NG);
    e->sign_mode = sign_mode;
    e->rows_alloc = (max_rows + HM_MAX_BLOCK_ROWS - 1) / HM_MAX_BLOCK_ROWS * HM_MAX_BLOCK_ROWS + HM_MAX_BLOCK_ROWS;
    e->KS = hm_pick_ks(e->d);
    if (e->KS < 0) {                     // d > 124: the k-steps instantiated do not hold d + 4 slots -- fp32 form only
        e->KS = 0;                       // (the bf16 image degenerates to its [x0] chunk)
        e->bf16_ok = false;
    }
    e->RB16 = 32 * e->KS + 16;
    {
        const char* pe = getenv("HM_SCAN_PRECISION");          // "f32" | "bf16" | unset = auto
        e->precision = (pe && !strcmp(pe, "f32")) ? 1 : (pe && !strcmp(pe, "bf16")) ? 2 : 0;
    }
    if (const char* t = getenv("HM_TUNE_CHUNK")) { const int v = atoi(t); if (v >= 4 && v <= 4096) e->chunk_f32 = e->chunk_bf16 = v; }
    if (const char* t = getenv("HM_TUNE_TAIL")) { const double v = atof(t); if (v >= 0.0 && v <= 0.9) e->tail_fraction = v; }
    if (const char* t = getenv("HM_TUNE_K1")) { const int v = atoi(t); if (v >= 1 && v <= 4096) e->dyn_k1 = v; }
    if (const char* t = getenv("HM_TUNE_K2")) { const int v = atoi(t); if (v >= 1 && v <= 4096) e->dyn_k2 = v; }
    if (const char* t = getenv("HM_TUNE_SPLIT")) { const double v = atof(t); if (v >= 0.0 && v <= 1.0) e->dyn_split = v; }
    if (const char* t = getenv("HM_TUNE_TM4_ROWS")) { const long v = atol(t); if (v >= 0) e->tm4_min_rows = v; }
    if (const char* t = getenv("HM_TUNE_TAIL_DIV")) { const int v = atoi(t); if (v >= 1 && v <= 16) e->tail_div = v; }
    e->ent_cap = 1u << 24;
    e->sorted_cap = 1u << 16;
    HM_HIP(hipSetDevice(device));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cu = cus;
    }
    HM_HIP(hipMalloc(&e->img, sizeof(float) * (size_t)e->rows_alloc * e->RS));
    HM_HIP(hipMemset(e->img, 0, sizeof(float) * (size_t)e->rows_alloc * e->RS));
    HM_HIP(hipMalloc(&e->img16, (size_t)e->rows_alloc * e->RB16));
    HM_HIP(hipMemset(e->img16, 0, (size_t)e->rows_alloc * e->RB16));
    HM_HIP(hipMalloc(&e->ent, sizeof(uint4) * (size_t)e->ent_cap));
    HM_HIP(hipMalloc(&e->ent2, sizeof(uint4) * (size_t)e->ent_cap));
    HM_HIP(hipMalloc(&e->sorted, sizeof(uint4) * (size_t)e->sorted_cap));
    HM_HIP(hipMalloc(&e->d_ctr, sizeof(uint32_t) * 8));
    HM_HIP(hipMalloc(&e->d_rmax2, sizeof(uint32_t) * 2));
    HM_HIP(hipMemset(e->d_rmax2, 0, sizeof(uint32_t) * 2));
    HM_HIP(hipMalloc(&e->d_ctr64, sizeof(unsigned long long) * 2));
    HM_HIP(hipMalloc(&e->d_seed, sizeof(ArgminSeed)));
    HM_HIP(hipMemset(e->d_seed, 0, sizeof(ArgminSeed)));
    HM_HIP(hipMalloc(&e->d_rec, 2 * sizeof(ArgminRec)));
    HM_HIP(hipMalloc(&e->d_parts, sizeof(ArgminPart) * HM_ARGMIN_BLOCKS));
    HM_HIP(hipMalloc(&e->d_hist, sizeof(uint32_t) * HM_DIGIT_BINS));
    HM_HIP(hipHostMalloc(&e->h, sizeof(HostCtl), hipHostMallocDefault));
    HM_HIP(hipHostMalloc(&e->h_sorted, sizeof(uint4) * (size_t)e->sorted_cap, hipHostMallocDefault));
    // timing events without the system-scope fence a default event adds around the scan
    HM_HIP(hipEventCreateWithFlags(&e->ev0, HM_EVENT_FLAGS));
    HM_HIP(hipEventCreateWithFlags(&e->ev1, HM_EVENT_FLAGS));
    *out = e;
    return HM_OK;
}

extern "C" int hm_engine_destroy(hm_engine* e)
{
    if (!e) return HM_OK;
    (void)hipSetDevice(e->device);
    void* dev_ptrs[] = {e->img, e->ent, e->ent2, e->sorted, e->d_ctr, e->d_ctr64, e->d_rec, e->d_hist, e->d_rmax2, e->d_parts, e->img16, e->d_seed};
    for (void* q : dev_ptrs) (void)hipFree(q);
    if (e->h) (void)hipHostFree(e->h);
    if (e->h_sorted) (void)hipHostFree(e->h_sorted);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return HM_OK;
}

extern "C" int64_t hm_rows(const hm_engine* e) { return e ? e->n : -1; }

static int hm_build_rows(hm_engine* e, const float* X, int64_t ld, int64_t r0, int64_t r1, hipStream_t s)
{
    if (r1 <= r0) return HM_OK;
    const int64_t total = (r1 - r0) * e->RS;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(hm_build_image_kernel, dim3(blocks), dim3(256), 0, s, X, ld, e->d, e->NG, e->img, r0, r1);
    HM_HIP(hipGetLastError());
    hipLaunchKernelGGL(hm_rownorm_kernel, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, s, e->img, e->RS, r0, r1, e->d_rmax2);
    HM_HIP(hipGetLastError());
    const int64_t total16 = (r1 - r0) * (2 * e->KS + 1);
    hipLaunchKernelGGL(hm_build_image16_kernel, dim3((unsigned)std::min<int64_t>((total16 + 255) / 256, 4096)), dim3(256), 0, s, X, ld,
                       e->d, e->KS, e->img16, r0, r1);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_set_table(hm_engine* e, const float* X_dev, int64_t ld, int64_t n_rows, void* stream)
{
    if (e) e->armed = false;
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_set_table: engine is NULL");
    if (!X_dev || ld < e->d1 || n_rows < 0 || n_rows > e->max_rows)
        return hm_fail(e, HM_E_ARG, "hm_set_table: bad table pointer / ld / n_rows");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    if (e->n > n_rows) { // rows that are no longer live must read as zeros (never candidates: masked)
        HM_HIP(hipMemsetAsync(e->img + n_rows * e->RS, 0, sizeof(float) * (size_t)(e->n - n_rows) * e->RS, s));
        HM_HIP(hipMemsetAsync(e->img16 + n_rows * e->RB16, 0, (size_t)(e->n - n_rows) * e->RB16, s));
    }
    HM_HIP(hipMemsetAsync(e->d_rmax2, 0, sizeof(uint32_t) * 2, s));
    int rc = hm_build_rows(e, X_dev, ld, 0, n_rows, s);
    if (rc) return rc;
    e->n = n_rows;
    e->have_cut = false;
    e->topk_f32_thr = 0.0f;
    HM_HIP(hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), s));      // new table: no seed
    return HM_OK;
}

extern "C" int hm_update_rows(hm_engine* e, const float* X_dev, int64_t ld, int64_t row_begin, int64_t row_end, void* stream)
{
    if (e) e->armed = false;
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_update_rows: engine is NULL");
    if (!X_dev || ld < e->d1 || row_begin < 0 || row_end < row_begin || row_end > e->max_rows)
        return hm_fail(e, HM_E_ARG, "hm_update_rows: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    int rc = hm_build_rows(e, X_dev, ld, row_begin, row_end, (hipStream_t)stream);
    if (rc) return rc;
    if (row_begin < e->n) {                          // an existing row changed: cut prediction and argmin seed void
        e->have_cut = false;
        HM_HIP(hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), (hipStream_t)stream));
    }
    if (row_end > e->n) e->n = row_end;
    return HM_OK;
}

// ---- threshold bounds in the u domain (double precision on the host) ----
struct Bounds { float u_hi, u_lo; bool none; int thr_pos; };

static Bounds hm_bounds(float thr, float c)
{
    Bounds b;
    b.none = !(thr > 0.0f);          // d >= 0 always: nothing is below a non-positive / NaN threshold
    b.thr_pos = thr > 0.0f ? 1 : 0;
    b.u_hi = 1.0f; b.u_lo = 1.0f;
    if (b.none) return b;
    const double sc = (double)sqrtf(c);
    const double a = (double)thr * sc;
    const double uh = cosh(a * (1.0 + 1e-5) + 1e-300);
    if (!(uh < 3.0e38)) {
        b.u_hi = INFINITY;
    } else {
        float f = (float)uh;
        if ((double)f < uh) f = nextafterf(f, INFINITY);
        f = nextafterf(nextafterf(f, INFINITY), INFINITY);
        b.u_hi = f;
    }
    const double ul = cosh(a * (1.0 - 1e-5));
    if (!(ul < 3.0e38)) {
        b.u_lo = 3.0e38f;
    } else {
        float f = (float)ul;
        if ((double)f > ul) f = nextafterf(f, 0.0f);
        f = nextafterf(nextafterf(f, 0.0f), 0.0f);
        b.u_lo = f < 1.0f ? 1.0f : f;
    }
    return b;
}

template <int NG, int SIGN, int MODE, int BF, int TM, int WPB, int TN>
static hipError_t hm_launch_scan_t(const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const size_t tile_bytes = BF ? (size_t)32 * TN * HM_SUB_BF16 * (32 * NG + 16) : sizeof(float) * 32 * TN * hm_row_floats(NG);
    const size_t ppw = (tile_bytes / 1024 + WPB - 1) / WPB;
    size_t lds = ((BF ? HM_DIST_BF16 : 1) + 1) * ppw * WPB * 1024;
    if (MODE == HM_MODE_HIST) lds += sizeof(uint32_t) * HM_HIST_BINS;
    else if (BF && HM_PERSIST == 2) lds += 64;               // the published next-chunk word
    static bool attr_set = false;    // per instantiation
    if (!attr_set && lds > 48 * 1024) {
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(&hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, TN>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
        attr_set = true;
    }
    // timed launches carry their events in the dispatch itself (start / stop timestamps of this kernel): a pair of
    // hipEventRecord calls around it costs two ~6 us bubbles on the stream
    if (ev0 != nullptr) hipExtLaunchKernelGGL((hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, TN>), grid, dim3(64 * WPB), lds, s, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, TN>), grid, dim3(64 * WPB), lds, s, a);
    return hipGetLastError();
}

template <int NG, int BF, int TM, int WPB, int TN>
static hipError_t hm_launch_scan_ng(int sign, int mode, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    if (sign) {
        if (mode == HM_MODE_TOPK) return hm_launch_scan_t<NG, 1, HM_MODE_TOPK, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
        if (mode == HM_MODE_ARGMIN) return hm_launch_scan_t<NG, 1, HM_MODE_ARGMIN, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
        return hm_launch_scan_t<NG, 1, HM_MODE_HIST, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
    }
    if (mode == HM_MODE_TOPK) return hm_launch_scan_t<NG, 0, HM_MODE_TOPK, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
    if (mode == HM_MODE_ARGMIN) return hm_launch_scan_t<NG, 0, HM_MODE_ARGMIN, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
    return hm_launch_scan_t<NG, 0, HM_MODE_HIST, BF, TM, WPB, TN>(a, grid, s, ev0, ev1);
}

// Which prefilter form a scan uses.  The bf16 form's error bound 0.00392 * max||x_s||^2 only costs
// extra emissions, never correctness; auto picks it from d >= 24 (below that the fp32 form is
// already short) unless a norm-based veto is set.
static bool hm_use_bf16(const hm_engine* e)
{
    if (!e->bf16_ok || e->force_f32) return false;
    if (e->precision == 1) return false;
    if (e->precision == 2) return true;
    return e->d >= 24 && e->bf16_ok;
}

static hipError_t hm_launch_scan(const hm_engine* e, int mode, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0 = nullptr,
                                 hipEvent_t ev1 = nullptr)
{
    if (a.bf16 && HM_PERSIST && mode != HM_MODE_HIST) grid = dim3((unsigned)a.p_grid, 1, 1);
    if (a.bf16 && a.tm4) {
        switch (e->KS) {
            case 1: return hm_launch_scan_ng<1, 1, 4, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 2: return hm_launch_scan_ng<2, 1, 4, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 4: return hm_launch_scan_ng<4, 1, 4, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 7: return hm_launch_scan_ng<7, 1, 4, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        }
        return hipErrorInvalidValue;
    }
    if (a.bf16) {
        switch (e->KS) {
            case 1: return hm_launch_scan_ng<1, 1, HM_TM_BF16, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 2: return hm_launch_scan_ng<2, 1, HM_TM_BF16, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 4: return hm_launch_scan_ng<4, 1, HM_TM_BF16, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 7: return hm_launch_scan_ng<7, 1, HM_TM_BF16, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
            case 8: return hm_launch_scan_ng<8, 1, HM_TM_BF16, HM_WPB_BF16, HM_TN_BF16>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        }
        return hipErrorInvalidValue;
    }
    switch (e->NG) {
        case 1: return hm_launch_scan_ng<1, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 2: return hm_launch_scan_ng<2, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 3: return hm_launch_scan_ng<3, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 4: return hm_launch_scan_ng<4, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 6: return hm_launch_scan_ng<6, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 8: return hm_launch_scan_ng<8, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 10: return hm_launch_scan_ng<10, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 13: return hm_launch_scan_ng<13, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 16: return hm_launch_scan_ng<16, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 20: return hm_launch_scan_ng<20, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 25: return hm_launch_scan_ng<25, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 28: return hm_launch_scan_ng<28, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 32: return hm_launch_scan_ng<32, 0, HM_TM_F32, 4, 2>(e->sign_mode, mode, a, grid, s, ev0, ev1);
    }
    return hipErrorInvalidValue;
}

// common argument preparation; returns false when the row range is empty
static int64_t hm_pairs_in_range(int64_t n, int64_t r0, int64_t r1);

static bool hm_prepare_scan(hm_engine* e, const Bounds& b, int64_t row_begin, int64_t row_end, ScanArgs& a, dim3& grid)
{
    if (row_end < 0 || row_end > e->n) row_end = e->n;
    if (row_begin < 0) row_begin = 0;
    if (row_end > e->n - 1) row_end = e->n - 1;        // the last row has no partner j > i
    if (row_begin >= row_end) return false;
    memset(&a, 0, sizeof(a));
    a.img = e->img;
    a.img16 = e->img16;
    a.bf16 = hm_use_bf16(e) ? 1 : 0;
    // large tables: 512-row blocks halve the L2 -> LDS fill traffic, which is what limits the bf16 form once the
    // launch tail no longer does (measured +7 % at 100 k rows, -4 % at 50 k); KS = 8 would not fit the registers
    // (decided by the pairs this launch covers: a row-range search of a sharded run is a small launch)
    a.tm4 = (a.bf16 && HM_TM_BF16 == 2 && e->KS <= 7 &&
             hm_pairs_in_range(e->n, row_begin, row_end) >= e->tm4_min_rows * (e->tm4_min_rows - 1) / 2) ? 1 : 0;
    const int block_rows = a.bf16 ? 32 * (a.tm4 ? 4 : HM_TM_BF16) * HM_WPB_BF16 : 128 * HM_TM_F32;
    const int cols = a.bf16 ? 32 * HM_TN_BF16 * HM_SUB_BF16 : 64;   // partner rows per streamed tile
    a.n = (int)e->n;
    a.row_begin = (int)row_begin;
    a.row_end = (int)row_end;
    a.rb_first = (int)(row_begin / block_rows);
    a.nct = (int)((e->n + cols - 1) / cols);
    a.u_hi = b.u_hi;
    a.u_lo = b.u_lo;
    a.thr_pos = b.thr_pos;
    a.cut_bits = 0xffffffffu;
    a.tie_imax = 0x7fffffff;
    a.ent = e->ent;
    a.ent_cap = e->ent_cap;
    a.ctr = e->d_ctr;
    a.ctr64 = e->d_ctr64;
    a.hist = e->d_hist;
    a.sample_stride = 1;
    a.rmax2_bits = e->d_rmax2;
    const int rb_last = (int)((row_end - 1) / block_rows);
    const int nrb = rb_last - a.rb_first + 1;
    const int tiles_per_rb = block_rows / cols;                              // diagonal advance per row block (>= 1)
    // column tiles per block: amortise the stationary-row load, but keep enough blocks in flight
    int ch = a.bf16 ? e->chunk_bf16 : e->chunk_f32;
    while (ch > 4 && (int64_t)nrb * ((a.nct + ch - 1) / ch) < 1024) ch >>= 1;
    // phase B = the last ~HM_TAIL_FRACTION of the work (row blocks near the bottom of the triangle),
    // cut into chunks a quarter the size
    a.ctmin_a = a.rb_first * tiles_per_rb;
    int rb_split = rb_last + 1;
    int ch_b = ch;
    if (ch >= 16 && nrb >= 16) {
        double total = 0.0, acc = 0.0;
        for (int rb = a.rb_first; rb <= rb_last; ++rb) total += (double)std::max(0, a.nct - rb * tiles_per_rb);
        for (int rb = rb_last; rb >= a.rb_first; --rb) {
            acc += (double)std::max(0, a.nct - rb * tiles_per_rb);
            if (acc >= e->tail_fraction * total) { rb_split = rb; break; }
        }
        ch_b = std::max(1, ch / e->tail_div);
    }
    a.ch_a = ch;
    a.chunks_a = std::max(1, (a.nct - a.ctmin_a + ch - 1) / ch);
    a.rb_split = rb_split;
    a.ch_b = ch_b;
    a.ctmin_b = rb_split * tiles_per_rb;
    a.chunks_b = std::max(1, (a.nct - a.ctmin_b + ch_b - 1) / ch_b);
    a.swizzle = 0;
#if HM_XCD_SWIZZLE
    if (a.chunks_a >= 12) {
        // round the chunk counts to multiples of 16 and re-derive the chunk widths
        a.chunks_a = (a.chunks_a + 15) / 16 * 16 + HM_SWIZZLE_EXTRA;
        a.ch_a = (a.nct - a.ctmin_a + a.chunks_a - 1) / a.chunks_a;
        const int wb = a.nct - a.ctmin_b;
        if (rb_split <= rb_last && wb > 0) {
            a.chunks_b = std::max(16, (a.chunks_b + 15) / 16 * 16);
            a.ch_b = std::max(1, (wb + a.chunks_b - 1) / a.chunks_b);
        }
        a.swizzle = 1;
    }
#endif
    a.n_items_a = (rb_split - a.rb_first) * a.chunks_a;
    int n_items_b = (rb_last + 1 - rb_split) * a.chunks_b;
    a.skip_empty = 0;
#if HM_SKIP_EMPTY
    if (!a.swizzle && a.ch_a % tiles_per_rb == 0 && a.ch_b % tiles_per_rb == 0) {
        a.skip_empty = 1;
        a.rows_a = rb_split - a.rb_first;
        a.m_a = a.ch_a / tiles_per_rb;
        a.rows_b = rb_last + 1 - rb_split;
        a.m_b = a.ch_b / tiles_per_rb;
        a.n_items_a = a.rows_a > 0 ? hm_tri_cum(a.rows_a, a.chunks_a, a.m_a) : 0;
        n_items_b = a.rows_b > 0 ? hm_tri_cum(a.rows_b, a.chunks_b, a.m_b) : 0;
    }
#endif
    grid = dim3((unsigned)std::max(1, a.n_items_a + n_items_b), 1, 1);
    // persistent decomposition (used by the bf16 TOPK / ARGMIN launches, see hm_scan_grid)
    a.p_nrb = nrb;
    a.p_total = (long long)nrb * a.nct - (long long)tiles_per_rb * ((long long)a.rb_first * nrb + (long long)nrb * (nrb - 1) / 2);
    a.p_grid = (int)std::max<long long>(1, std::min<long long>((long long)e->n_cu * HM_PERSIST_BLOCKS_PER_CU, a.p_total / 4));
    a.p_k1 = e->dyn_k1;
    a.p_k2 = e->dyn_k2;
    a.p_c1 = (int)((double)a.p_total * e->dyn_split / a.p_k1);
    return true;
}

static int64_t hm_pairs_in_range(int64_t n, int64_t r0, int64_t r1)
{
    // sum_{i=r0}^{r1-1} (n - 1 - i)
    const int64_t cnt = r1 - r0;
    return cnt * (n - 1) - (r0 + r1 - 1) * cnt / 2;
}

static void hm_flush_pending_timing(hm_engine* e)
{
    if (!e->pending_timing) return;
    float ms = 0.f;
    if (hipEventSynchronize(e->ev1) == hipSuccess && hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) {
        e->last_scan_ms = ms; e->last_pairs = e->pending_pairs; e->last_passes = 1;
        e->tot_scan_ms += ms; e->tot_pairs += e->pending_pairs; e->tot_launches += 1;
    }
    e->pending_timing = false;
}

// arguments of hm_post_argmin_final_kernel that describe the seed update
#define HM_SEED_ARGS(e, a) (e)->d_seed, (e)->img, (e)->RS, (e)->d, (e)->sign_mode, (a).bf16, ((a).bf16 ? 16 * (e)->KS : (e)->RS), (e)->d_rmax2

#if HM_DIAG_TIMES
extern "C" int hm_debug_read_hist(hm_engine* e, uint32_t* out, int n)
{
    HM_HIP(hipSetDevice(e->device));
    HM_HIP(hipDeviceSynchronize());
    HM_HIP(hipMemcpy(out, e->d_hist, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return HM_OK;
}
#endif

extern "C" int hm_pairwise_argmin_dev(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end, uint32_t* rec_dev,
                                      void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pairwise_argmin_dev: engine is NULL");
    if (!rec_dev) return hm_fail(e, HM_E_ARG, "hm_pairwise_argmin_dev: NULL record pointer");
    if (!(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pairwise_argmin_dev: curvature must be > 0");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    hm_flush_pending_timing(e);
    const Bounds b = hm_bounds(thr, c);
    ScanArgs a; dim3 grid;
    const int64_t req_rb = std::max<int64_t>(row_begin, 0), req_re = (row_end < 0 || row_end >= e->n) ? -1 : row_end;
    const bool skip_init = HM_ARM_NEXT && e->armed && e->armed_rb == req_rb && e->armed_re == req_re;
    e->armed = false;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid)) {
        HM_HIP(hipMemsetAsync(rec_dev, 0, sizeof(ArgminRec), s));        // found = 0
        return HM_OK;
    }
    if (!skip_init) {
        hipLaunchKernelGGL(hm_seed_init_kernel, dim3(1), dim3(64), 0, s, e->d_seed, e->d_ctr64, e->d_ctr, a.row_begin, a.row_end);
        HM_HIP(hipGetLastError());
    }
    HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, e->ev0, e->ev1));
    hipLaunchKernelGGL(hm_post_argmin_kernel, dim3(HM_ARGMIN_BLOCKS), dim3(256), 0, s, e->ent, e->d_ctr, e->ent_cap, e->img, e->RS,
                       e->d, e->sign_mode, sqrtf(c), thr, e->d_parts);
    HM_HIP(hipGetLastError());
    hipLaunchKernelGGL(hm_post_argmin_final_kernel, dim3(1), dim3(HM_ARGMIN_BLOCKS), 0, s, e->d_parts,
                       reinterpret_cast<ArgminRec*>(rec_dev), e->d_ctr, e->ent_cap, HM_SEED_ARGS(e, a), (uint32_t*)nullptr,
                       HM_ARM_NEXT ? e->d_ctr : (uint32_t*)nullptr, e->d_ctr64, (int)req_rb,
                       req_re < 0 ? 0x7fffffff : (int)std::min<int64_t>(req_re, 0x7fffffff));
    HM_HIP(hipGetLastError());
    // Armed optimistically: the host does not see this record.  Should the search have overflowed (found = 2, the
    // kernel then arms nothing), the next search of this range starts on the stale counters, reports found = 2 as
    // well, and its caller takes the bounded host path -- slower, never wrong.
    if (HM_ARM_NEXT) { e->armed = true; e->armed_rb = req_rb; e->armed_re = req_re; }
    e->pending_timing = true;
    e->pending_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
    return HM_OK;
}

extern "C" int hm_pairwise_argmin(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end, float* d, int32_t* i,
                                  int32_t* j, int32_t* found, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pairwise_argmin: engine is NULL");
    if (!d || !i || !j || !found) return hm_fail(e, HM_E_ARG, "hm_pairwise_argmin: NULL output pointer");
    if (!(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pairwise_argmin: curvature must be > 0");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *found = 0; *d = 0.f; *i = -1; *j = -1;
    hm_flush_pending_timing(e);
    e->last_scan_ms = 0.f; e->last_pairs = 0; e->last_emitted = 0; e->last_passes = 0;
    const Bounds b = hm_bounds(thr, c);
    ScanArgs a; dim3 grid;
    // the range as asked, "to the end" normalised (the table grows between searches): what "same range" means
    const int64_t req_rb = std::max<int64_t>(row_begin, 0), req_re = (row_end < 0 || row_end >= e->n) ? -1 : row_end;
    const bool skip_init = HM_ARM_NEXT && e->armed && e->armed_rb == req_rb && e->armed_re == req_re;
    e->armed = false;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid)) return HM_OK;
    const float sqrt_c = sqrtf(c);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 1 (after an overflow) keeps the final running key of pass 0: every wave then starts
        // with the tight bound and only the band around the minimum is emitted
        if (pass == 0) {
            if (!skip_init) {           // else: the previous search of this range left counters and key armed
                hipLaunchKernelGGL(hm_seed_init_kernel, dim3(1), dim3(64), 0, s, e->d_seed, e->d_ctr64, e->d_ctr, a.row_begin, a.row_end);
                HM_HIP(hipGetLastError());
            }
        } else {
            HM_HIP(hipMemsetAsync(e->d_ctr, 0, sizeof(uint32_t) * 8, s));
        }
        HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, e->ev0, e->ev1));
        hipLaunchKernelGGL(hm_post_argmin_kernel, dim3(HM_ARGMIN_BLOCKS), dim3(256), 0, s, e->ent, e->d_ctr, e->ent_cap, e->img,
                           e->RS, e->d, e->sign_mode, sqrt_c, thr, e->d_parts);
        HM_HIP(hipGetLastError());
        hipLaunchKernelGGL(hm_post_argmin_final_kernel, dim3(1), dim3(HM_ARGMIN_BLOCKS), 0, s, e->d_parts, e->d_rec, e->d_ctr, e->ent_cap,
                           HM_SEED_ARGS(e, a), reinterpret_cast<uint32_t*>(e->d_rec + 1), e->d_ctr, e->d_ctr64,
                           (int)std::max<int64_t>(req_rb, 0), req_re < 0 ? 0x7fffffff : (int)std::min<int64_t>(req_re, 0x7fffffff));
        HM_HIP(hipGetLastError());
        // record + emitted count (the slot behind the record) in one copy
        HM_HIP(hipMemcpyAsync(e->h->rec2, e->d_rec, 2 * sizeof(ArgminRec), hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        e->h->rec = e->h->rec2[0];
        e->h->ctr[0] = e->h->rec2[1].found;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
        e->last_scan_ms += ms;
        e->last_passes += 1;
        e->last_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
        e->tot_scan_ms += ms; e->tot_pairs += e->last_pairs; e->tot_launches += 1;
        e->last_emitted = e->h->ctr[0];
        if (e->h->ctr[0] <= e->ent_cap) break;
        // overflow: the running key is the exact minimum over all published waves; rerun bounded by it
        if (pass == 1) {
            if (a.bf16 && !e->force_f32) {      // the bf16 margin's shell around the bound is too populated: fp32 prefilter
                e->force_f32 = true;
                const int rc = hm_pairwise_argmin(e, c, thr, row_begin, row_end, d, i, j, found, stream);
                e->force_f32 = false;
                return rc;
            }
            // Still too many pairs inside the running key's slack band (a very dense table: the band is 1024 ulps of
            // u, which near u = 1 spans every distance below ~0.016): take the first entry of an exact top-1 search,
            // whose emission cut is found by histogram zooming instead.
            float d1 = 0.f; int32_t i1 = -1, j1 = -1; int64_t n1 = 0, cnt1 = 0;
            const int rc = hm_pairwise_topk(e, c, thr, 1, row_begin, row_end, &d1, &i1, &j1, &n1, &cnt1, stream);
            if (rc) return rc;
            if (n1 > 0) { *found = 1; *d = d1; *i = i1; *j = j1; }
            return HM_OK;
        }
    }
    if (e->h->rec.found == 1u) {
        union { uint32_t u; float f; } cv; cv.u = e->h->rec.dbits;
        *found = 1; *d = cv.f; *i = (int32_t)e->h->rec.i; *j = (int32_t)e->h->rec.j;
    }
    if (HM_ARM_NEXT && e->h->rec.found != 2u) { e->armed = true; e->armed_rb = req_rb; e->armed_re = req_re; }
    return HM_OK;
}

// exact selection of the k smallest keys among m entries of `src` (keys unique; invalid = 0xffffffff)
// result in e->sorted / e->h_sorted.  `other` is scratch of the same capacity.
static int hm_select_sorted(hm_engine* e, uint4* src, uint4* other, uint32_t m, uint32_t k, hipStream_t s)
{
    if (k == 0 || m == 0) return HM_OK;
    if (k > e->sorted_cap) return hm_fail(e, HM_E_CAPACITY, "top-k: k exceeds the engine's sorted capacity (65536)");
    uint4* cur = src;
    uint32_t mcur = m;
    const uint32_t rank_limit = std::min<uint32_t>(HM_RANK_LIMIT, std::max<uint32_t>(2u * k, 8192u));
    if (mcur > rank_limit) {
        // radix narrowing on the 96-bit key, 12/12/8-bit digits per word
        Prefix pf; memset(&pf, 0, sizeof(pf));
        uint32_t below = 0;      // keys strictly below the prefix: certainly selected
        int level = 0;
        for (; level < 9; ++level) {
            int word, shift, bits;
            hm_digit_pos(level, word, shift, bits);
            HM_HIP(hipMemsetAsync(e->d_hist, 0, sizeof(uint32_t) * HM_DIGIT_BINS, s));
            hipLaunchKernelGGL(hm_digit_hist_kernel, dim3(1024), dim3(256), 0, s, cur, mcur, pf, level, e->d_hist);
            HM_HIP(hipGetLastError());
            HM_HIP(hipMemcpyAsync(e->h->hist, e->d_hist, sizeof(uint32_t) * HM_DIGIT_BINS, hipMemcpyDeviceToHost, s));
            HM_HIP(hipStreamSynchronize(s));
            const uint32_t nb = 1u << bits;
            uint32_t cum = below, dsel = nb - 1, match = 0;
            for (uint32_t q = 0; q < nb; ++q) {
                if (cum + e->h->hist[q] >= k) { dsel = q; match = e->h->hist[q]; break; }
                cum += e->h->hist[q];
            }
            below = cum;
            pf.val[word] |= dsel << shift;
            pf.mask[word] |= ((1u << bits) - 1u) << shift;
            if (below + match <= rank_limit) break;
        }
        HM_HIP(hipMemsetAsync(e->d_ctr + 3, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(hm_compact_prefix_kernel, dim3(1024), dim3(256), 0, s, cur, mcur, pf, other, e->d_ctr + 3, e->ent_cap);
        HM_HIP(hipGetLastError());
        HM_HIP(hipMemcpyAsync(&e->h->ctr[3], e->d_ctr + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        cur = other;
        mcur = e->h->ctr[3];
        if (mcur > 4u * HM_RANK_LIMIT) return hm_fail(e, HM_E_CAPACITY, "top-k: radix narrowing did not converge");
    }
    hipLaunchKernelGGL(hm_rank_sort_kernel, dim3((mcur + 256 / HM_RANK_SPLIT - 1) / (256 / HM_RANK_SPLIT)), dim3(256), 0, s, cur, mcur, e->sorted, k);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// Choose an emission cut from sampled histograms of bits(u') so that roughly `target` entries
// (and certainly not more than the buffer holds) are emitted.  Pure performance heuristic: the
// caller verifies the outcome and widens the cut when fewer than k valid entries came back.
static int hm_estimate_cut(hm_engine* e, ScanArgs a, dim3 grid, int64_t target, uint32_t* cut_bits, int* tie_imax,
                           hipStream_t s)
{
    *cut_bits = 0xffffffffu;
    *tie_imax = 0x7fffffff;
    const int64_t pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
    if (pairs <= (int64_t)e->ent_cap / 2) return HM_OK;           // everything fits: emit all candidates
    union { uint32_t u; float f; } hi; hi.f = a.u_hi;
    uint32_t lo_bits = 0x3f800000u;
    uint32_t hi_bits = hi.f == INFINITY ? 0x7f800000u : hi.u;
    int stride = 1;
    while (stride < 64 && pairs / (stride * 2) > 40000000) stride *= 2;
    double base = 0.0;               // estimated entries below the current zoom window
    for (int zoom = 0; zoom < 6; ++zoom) {
        uint32_t span = hi_bits - lo_bits;
        uint32_t shift = 0;
        while ((span >> shift) > HM_HIST_BINS) ++shift;
        a.hist_lo = lo_bits;
        a.hist_shift = shift;
        a.sample_stride = stride;
        HM_HIP(hipMemsetAsync(e->d_hist, 0, sizeof(uint32_t) * HM_DIGIT_BINS, s));
        HM_HIP(hm_launch_scan(e, HM_MODE_HIST, a, grid, s));
        HM_HIP(hipMemcpyAsync(e->h->hist, e->d_hist, sizeof(uint32_t) * HM_HIST_BINS, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        e->last_passes += 1;
        double cum = base;
        int bsel = -1;
        for (int q = 0; q < HM_HIST_BINS; ++q) {
            cum += (double)e->h->hist[q] * stride;
            if (cum >= (double)target) { bsel = q; break; }
        }
        if (bsel < 0) return HM_OK;                                 // fewer than target below u_hi: emit all
        const double before = cum - (double)e->h->hist[bsel] * stride;
        const uint32_t edge_lo = lo_bits + ((uint32_t)bsel << shift);
        const uint32_t edge_hi = lo_bits + (((uint32_t)bsel + 1u) << shift);   // exclusive
        if (cum <= (double)e->ent_cap * 0.5 || shift == 0) {
            *cut_bits = edge_hi - 1u;
            if (cum > (double)e->ent_cap * 0.5) {
                // a single value of u' holds more entries than the buffer: tie flood.  Emit the tie
                // value only for the first rows; rows are visited in row-major order by the selection.
                // estimate rows needed from the average ties per row
                const double per_row = ((double)e->h->hist[bsel] * stride) / (double)(a.row_end - a.row_begin);
                double rows = ((double)target - before) / (per_row > 1e-9 ? per_row : 1e-9);
                int64_t imax = a.row_begin + (int64_t)(rows * 2.0) + 64;
                if (imax > a.row_end) imax = a.row_end;
                *tie_imax = (int)imax;
            }
            return HM_OK;
        }
        lo_bits = edge_lo;
        hi_bits = edge_hi;
        base = before;
    }
    return HM_OK;
}

static int hm_topk_core_form(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all,
                             int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s);

// The bf16 prefilter's margin (delta ~ 0.004 * max||x_s||^2 in u) makes a shell of undecided pairs around the
// threshold; every one of them has to be emitted to be decided exactly.  With a threshold inside the bulk of the
// distance distribution that shell alone can exceed the emission buffer: the search then runs again with the fp32
// prefilter, whose shell is ~100x thinner.
static int hm_topk_core(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all,
                        int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s)
{
    const bool had_bf16 = hm_use_bf16(e);
    int rc = HM_E_CAPACITY;
    if (!(had_bf16 && e->topk_f32_thr > 0.0f && thr >= e->topk_f32_thr))     // (a fallback is remembered per table)
        rc = hm_topk_core_form(e, c, thr, k, row_begin, row_end, list_all, n_valid_emitted, count, result_dev, s);
    if (rc == HM_E_CAPACITY && had_bf16) {
        e->force_f32 = true;
        rc = hm_topk_core_form(e, c, thr, k, row_begin, row_end, list_all, n_valid_emitted, count, result_dev, s);
        e->force_f32 = false;
        if (rc == HM_OK && !(e->topk_f32_thr > 0.0f && e->topk_f32_thr <= thr)) e->topk_f32_thr = thr;
    }
    return rc;
}

static int hm_topk_core_form(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all,
                             int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s)
{
    *n_valid_emitted = 0;
    *count = 0;
    *result_dev = nullptr;
    hm_flush_pending_timing(e);
    e->last_scan_ms = 0.f; e->last_pairs = 0; e->last_emitted = 0; e->last_passes = 0;
    const Bounds b = hm_bounds(thr, c);
    ScanArgs a; dim3 grid;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid)) return HM_OK;
    const float sqrt_c = sqrtf(c);
    const bool whole = (a.row_begin == 0 && a.row_end == e->n - 1);

    uint32_t cut_bits = 0xffffffffu;
    int tie_imax = 0x7fffffff;
    if (!list_all) {
        if (whole && e->have_cut && e->last_cut_k >= k && e->last_cut_c == c && e->last_cut_bits > 0x3f800000u) {
            cut_bits = e->last_cut_bits + HM_TIE_SLACK;
        } else {
            int rc = hm_estimate_cut(e, a, grid, 4 * k + 4096, &cut_bits, &tie_imax, s);
            if (rc) return rc;
        }
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        a.cut_bits = cut_bits;
        a.tie_imax = tie_imax;
        HM_HIP(hipMemsetAsync(e->d_ctr, 0, sizeof(uint32_t) * 8, s));
        HM_HIP(hipMemsetAsync(e->d_ctr64, 0, sizeof(unsigned long long) * 2, s));
        HM_HIP(hm_launch_scan(e, HM_MODE_TOPK, a, grid, s, e->ev0, e->ev1));
        hipLaunchKernelGGL(hm_post_distance_kernel, dim3(512), dim3(256), 0, s, e->ent, e->d_ctr, e->ent_cap, e->img, e->RS, e->d,
                           e->sign_mode, sqrt_c, thr, e->d_ctr + 1);
        HM_HIP(hipGetLastError());
        HM_HIP(hipMemcpyAsync(e->h->ctr, e->d_ctr, sizeof(uint32_t) * 8, hipMemcpyDeviceToHost, s));
        HM_HIP(hipMemcpyAsync(e->h->ctr64, e->d_ctr64, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
        e->last_scan_ms += ms;
        e->last_passes += 1;
        e->last_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
        e->tot_scan_ms += ms; e->tot_pairs += e->last_pairs; e->tot_launches += 1;
        e->last_emitted = e->h->ctr[0];
        if (e->h->ctr[4] != 0)
            return hm_fail(e, HM_E_STATE, "pair scan: prefilter margin violated (an entry classified as surely below the "
                                          "threshold is not); table holds non-finite rows other than all-NaN rows?");
        const uint64_t emitted = e->h->ctr[0];
        const bool overflow = emitted > e->ent_cap;
        const int64_t total = (int64_t)e->h->ctr64[0] + (int64_t)e->h->ctr[2];   // sure + valid borderline
        const int64_t valid = e->h->ctr[1];
        const bool emitted_all = (cut_bits == 0xffffffffu);
        if (overflow) {
            if (list_all) return hm_fail(e, HM_E_CAPACITY, "candidate listing: more candidates than the emission buffer holds");
            // estimate was too generous (or none was made): estimate with a smaller target
            int rc = hm_estimate_cut(e, a, grid, std::max<int64_t>((2 * k + 1024) >> attempt, k + 64), &cut_bits, &tie_imax, s);
            if (rc) return rc;
            if (cut_bits == 0xffffffffu) return hm_fail(e, HM_E_CAPACITY, "top-k: could not bound the emission");
            continue;
        }
        const int64_t want = std::min<int64_t>(k, total);
        if (!emitted_all && valid < want) {
            // the cut was too tight: widen geometrically in the ulp domain and retry
            if (tie_imax != 0x7fffffff) {
                tie_imax = tie_imax >= a.row_end ? 0x7fffffff : (int)std::min<int64_t>((int64_t)tie_imax * 4 + 256, a.row_end);
                if (tie_imax >= a.row_end) tie_imax = 0x7fffffff;
            } else {
                const uint32_t span = cut_bits - 0x3f800000u;
                const uint64_t nb = (uint64_t)cut_bits + std::max<uint32_t>(span, 1024u);
                cut_bits = nb >= 0x7f800000ull ? 0xffffffffu : (uint32_t)nb;
            }
            continue;
        }
        *count = total;
        *n_valid_emitted = valid;
        *result_dev = e->ent;
        return HM_OK;
    }
    return hm_fail(e, HM_E_CAPACITY, "top-k: emission cut did not converge");
}

extern "C" int hm_pairwise_topk(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, float* d_out,
                                int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count, void* stream)
{
    if (e) e->armed = false;
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pairwise_topk: engine is NULL");
    if (!n_out || !count || k < 0 || (k > 0 && (!d_out || !i_out || !j_out)))
        return hm_fail(e, HM_E_ARG, "hm_pairwise_topk: bad output pointers / k");
    if (!(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pairwise_topk: curvature must be > 0");
    if (k > (int64_t)e->sorted_cap) return hm_fail(e, HM_E_CAPACITY, "hm_pairwise_topk: k > 65536");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *n_out = 0; *count = 0;
    int64_t valid = 0, total = 0;
    uint4* res = nullptr;
    int rc = hm_topk_core(e, c, thr, k, row_begin, row_end, false, &valid, &total, &res, s);
    if (rc) return rc;
    *count = total;
    const uint32_t kk = (uint32_t)std::min<int64_t>(k, valid);
    if (kk == 0 || !res) return HM_OK;
    const uint32_t m = (uint32_t)std::min<uint64_t>(e->h->ctr[0], e->ent_cap);
    rc = hm_select_sorted(e, res, e->ent2, m, kk, s);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(e->h_sorted, e->sorted, sizeof(uint4) * kk, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    for (uint32_t t = 0; t < kk; ++t) {
        union { uint32_t u; float f; } cv; cv.u = e->h_sorted[t].x;
        d_out[t] = cv.f; i_out[t] = (int32_t)e->h_sorted[t].y; j_out[t] = (int32_t)e->h_sorted[t].z;
    }
    *n_out = kk;
    // remember the u' of the k-th entry: while rows are only appended, the k-th smallest key can
    // only move down, so this cut (+ tie slack) is a guaranteed superset for the next refresh
    if (kk == k && row_begin <= 0 && (row_end < 0 || row_end >= e->n - 1)) {
        e->have_cut = true;
        e->last_cut_bits = e->h_sorted[kk - 1].w;
        // entries are ordered by distance, not by u': take the max u' bits over the selection
        uint32_t mx = 0;
        for (uint32_t t = 0; t < kk; ++t) mx = std::max(mx, e->h_sorted[t].w);
        e->last_cut_bits = mx;
        e->last_cut_k = k;
        e->last_cut_c = c;
    } else {
        e->have_cut = false;
    }
    return HM_OK;
}

extern "C" int hm_pairwise_candidates(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end, int64_t cap,
                                      int32_t* i_out, int32_t* j_out, float* d_out, int64_t* total, void* stream)
{
    if (e) e->armed = false;
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pairwise_candidates: engine is NULL");
    if (!total || cap < 0 || (cap > 0 && (!i_out || !j_out || !d_out)))
        return hm_fail(e, HM_E_ARG, "hm_pairwise_candidates: bad output pointers");
    if (!(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pairwise_candidates: curvature must be > 0");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *total = 0;
    int64_t valid = 0, cnt = 0;
    uint4* res = nullptr;
    int rc = hm_topk_core(e, c, thr, 0, row_begin, row_end, true, &valid, &cnt, &res, s);
    if (rc) return rc;
    *total = cnt;
    if (!res || valid == 0 || cap == 0) return HM_OK;
    const uint32_t m = (uint32_t)std::min<uint64_t>(e->h->ctr[0], e->ent_cap);
    HM_HIP(hipMemsetAsync(e->d_ctr + 3, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(hm_compact_valid_kernel, dim3(1024), dim3(256), 0, s, res, m, e->ent2, e->d_ctr + 3, e->ent_cap);
    HM_HIP(hipGetLastError());
    const int64_t ncopy = std::min<int64_t>(valid, cap);
    std::vector<uint4> host((size_t)ncopy);
    HM_HIP(hipMemcpyAsync(host.data(), e->ent2, sizeof(uint4) * (size_t)ncopy, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    for (int64_t t = 0; t < ncopy; ++t) {
        union { uint32_t u; float f; } cv; cv.u = host[(size_t)t].x;
        d_out[t] = cv.f; i_out[t] = (int32_t)host[(size_t)t].y; j_out[t] = (int32_t)host[(size_t)t].z;
    }
    return HM_OK;
}

extern "C" int hm_row_vs_all(hm_engine* e, int64_t row, int64_t n, float c, float* d_out_dev, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_row_vs_all: engine is NULL");
    if (!d_out_dev || row < 0 || row >= e->n || n < 0 || n > e->n || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_row_vs_all: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (n == 0) return HM_OK;
    hipLaunchKernelGGL(hm_rowvsall_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                       row, n, sqrtf(c), e->sign_mode, d_out_dev);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_row_argmin(hm_engine* e, int64_t row, int64_t n_partners, float c, float thr, float* d, int32_t* i, int32_t* j,
                             int32_t* found, void* stream)
{
    if (e) e->armed = false;
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_row_argmin: engine is NULL");
    if (!d || !i || !j || !found) return hm_fail(e, HM_E_ARG, "hm_row_argmin: NULL output pointer");
    if (row < 0 || row >= e->n || n_partners < 0 || n_partners > e->n || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_row_argmin: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *found = 0; *d = 0.f; *i = -1; *j = -1;
    if (!(thr > 0.0f) || n_partners == 0) return HM_OK;
    HM_HIP(hipMemsetAsync(e->d_ctr, 0, sizeof(uint32_t) * 8, s));     // ctr[0] = 0: the final kernel's overflow test stays quiet
    hipLaunchKernelGGL(hm_row_argmin_kernel, dim3(HM_ARGMIN_BLOCKS), dim3(256), 0, s, e->img, e->RS, e->d, e->sign_mode, row,
                       n_partners, sqrtf(c), thr, e->d_parts);
    HM_HIP(hipGetLastError());
    hipLaunchKernelGGL(hm_post_argmin_final_kernel, dim3(1), dim3(HM_ARGMIN_BLOCKS), 0, s, e->d_parts, e->d_rec, e->d_ctr, e->ent_cap,
                       (ArgminSeed*)nullptr, e->img, e->RS, e->d, e->sign_mode, 0, e->RS, e->d_rmax2, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, (unsigned long long*)nullptr, 0, 0);
    HM_HIP(hipGetLastError());
    HM_HIP(hipMemcpyAsync(&e->h->rec, e->d_rec, sizeof(ArgminRec), hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    if (e->h->rec.found == 1u) {
        union { uint32_t u; float f; } cv; cv.u = e->h->rec.dbits;
        *found = 1; *d = cv.f; *i = (int32_t)e->h->rec.i; *j = (int32_t)e->h->rec.j;
    }
    return HM_OK;
}

extern "C" int hm_pair_distance(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, int64_t b, float c, float* out_dev,
                                void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pair_distance: engine is NULL");
    if (b < 0 || (b > 0 && (!I_dev || !J_dev || !out_dev)) || !(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pair_distance: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    if (b == 0) return HM_OK;
    hipLaunchKernelGGL(hm_pairdist_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, e->img, e->RS, e->d,
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
    hipLaunchKernelGGL(hm_midpoint_kernel, dim3((unsigned)((b + 63) / 64)), dim3(64), 0, (hipStream_t)stream, e->img, e->RS, e->d,
                       I_dev, J_dev, W_dev, b, c, e->sign_mode, out_dev);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

extern "C" int hm_merge_append(hm_engine* e, int32_t i, int32_t j, float w, float c, float* X_dev, int64_t ld, int64_t new_row,
                               void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_merge_append: engine is NULL");
    if (!X_dev || ld < e->d1 || i < 0 || j < 0 || i >= e->n || j >= e->n || new_row < 0 || new_row >= e->max_rows)
        return hm_fail(e, HM_E_ARG, "hm_merge_append: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    hipLaunchKernelGGL(hm_merge_append_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e->img, e->RS, e->d, e->NG, i, j, w, c,
                       e->sign_mode, X_dev, ld, new_row, e->d_rmax2, e->img16, e->KS);
    HM_HIP(hipGetLastError());
    if (new_row < e->n) {
        e->armed = false;
        e->have_cut = false;
        HM_HIP(hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), (hipStream_t)stream));
    }
    if (new_row + 1 > e->n) e->n = new_row + 1;
    return HM_OK;
}

// ---- engine-independent entry points ----
#define HM_HIP0(call)                                                                                 \
    do {                                                                                              \
        hipError_t _st = (call);                                                                      \
        if (_st != hipSuccess)                                                                        \
            return hm_fail(nullptr, (int)_st, std::string(#call) + ": " + hipGetErrorString(_st));    \
    } while (0)

extern "C" int hm_batch_distance(const float* X_dev, int64_t n1, const float* Y_dev, int64_t n2, int64_t ld_x, int64_t ld_y, int d1,
                                 float c, int sign_mode, float* out_dev, void* stream)
{
    if (n1 < 0 || n2 < 0 || d1 < 2 || ld_x < d1 || ld_y < d1 || !(c > 0.0f)) return hm_fail(nullptr, HM_E_ARG, "hm_batch_distance: bad arguments");
    if (n1 == 0 || n2 == 0) return HM_OK;
    if (!X_dev || !Y_dev || !out_dev) return hm_fail(nullptr, HM_E_ARG, "hm_batch_distance: NULL pointer");
    if (n1 > 2147483647LL / 1) return hm_fail(nullptr, HM_E_ARG, "hm_batch_distance: n1 too large");
    for (int64_t i0 = 0; i0 < n1; i0 += 32768) {
        const int64_t rows = std::min<int64_t>(32768, n1 - i0);
        hipLaunchKernelGGL(hm_dense_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                           X_dev + i0 * ld_x, rows, Y_dev, n2, ld_x, ld_y, d1, sqrtf(c), sign_mode, out_dev + i0 * n2);
        HM_HIP0(hipGetLastError());
    }
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
    hipLaunchKernelGGL(hm_rows_distance_kernel, dim3((unsigned)((b + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x_dev, y_dev, b,
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

extern "C" int hm_last_scan_stats(const hm_engine* e, float* scan_ms, int64_t* pairs, int64_t* emitted, int32_t* passes)
{
    if (!e) return HM_E_ARG;
    if (scan_ms) *scan_ms = e->last_scan_ms;
    if (pairs) *pairs = e->last_pairs;
    if (emitted) *emitted = e->last_emitted;
    if (passes) *passes = e->last_passes;
    return HM_OK;
}

extern "C" int hm_scan_totals(hm_engine* e, double* scan_ms, int64_t* pairs, int64_t* launches, int reset)
{
    if (!e) return HM_E_ARG;
    hm_flush_pending_timing(e);
    if (scan_ms) *scan_ms = e->tot_scan_ms;
    if (pairs) *pairs = e->tot_pairs;
    if (launches) *launches = e->tot_launches;
    if (reset) { e->tot_scan_ms = 0.0; e->tot_pairs = 0; e->tot_launches = 0; }
    return HM_OK;
}
