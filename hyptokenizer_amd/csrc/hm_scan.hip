// hm_scan.hip -- the pair scan: X.G.X^T on the matrix cores as a PREFILTER for the exact search.
//
// Replaces the N x N distance matrix of the reference (batch_distance + triu + "< thr" + nonzero,
// tokenizer/hyperbolic_merge.py:247-269, fast_hyperbolic_merge.py:336-355); never materialises it.
//
// Shape.  Block = WPB waves; each wave keeps 32*TM *stationary* rows as MFMA A-fragments in registers for the
// whole block; the *partner* rows stream through LDS as tiles of 32*SUB rows, filled by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of slots.  Every wave issues the same
// number of pieces per tile, so "tile landed" is a counted s_waitcnt vmcnt(N) followed by one s_barrier per tile.
// A tile is walked as SUB groups of 32 columns.  Two accumulator sets alternate between groups: while the MFMAs
// of group g fill one set, the bound test of group g-1 (v_max3 tree, compare, ballot) runs on the other, in the
// same basic block -- the matrix pipe never drains for an epilogue, and the first LDS fragment of group g+1 is
// requested before the branch that ends group g.
//
// Two prefilter forms, one result (every reported distance is re-evaluated canonically, hm_search.hip):
//   BF = 0  v_mfma_f32_32x32x2_f32 on the fp32 image (exact fmaf chain; NG = groups of 4 spatial coordinates)
//   BF = 1  v_mfma_f32_32x32x16_bf16 on the bf16 image (NG = chunks of 8 K-slots, two per k-step; an odd count ends on
//           one v_mfma_f32_32x32x8_bf16_1k half step: d = 100 -> 13 chunks; time coordinate in split slots)
// A bound delta >= |u_f - u_c| (hm_scan_delta) widens every comparison.
#include <type_traits>

#include "hm_common.h"

#pragma clang fp contract(off)

#ifndef HM_SUB_BF16
#define HM_SUB_BF16 4              // bf16 form: 32-column groups per streamed tile (128 partner rows per barrier)
#endif
#ifndef HM_SUB_F32
#define HM_SUB_F32 2
#endif
#ifndef HM_DIST_BF16
#define HM_DIST_BF16 1             // tiles in flight ahead of the computed one (ring of DIST + 1 slots)
#endif
#ifndef HM_WPB_BIG
#define HM_WPB_BIG 8               // bf16 form, large launches: 8 waves x 64 rows = 512-row blocks (half the L2 -> LDS fill traffic)
#endif
#ifndef HM_SCAN_INSTANTIATE_ALL
#define HM_SCAN_INSTANTIATE_ALL 1  // 0: only d = 100 (13 chunks / NG = 25) -- quick builds for tuning runs
#endif

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

#ifndef HM_SCAN_PIPE_MAX_TM
#define HM_SCAN_PIPE_MAX_TM 2
#endif
#ifndef HM_SCAN_DEEP_PREFETCH
#define HM_SCAN_DEEP_PREFETCH 1
#endif
template <int NG, int SIGN, int MODE, int BF, int TM, int WPB, int SUB>
__global__ __launch_bounds__(64 * WPB, 2) void hm_scan_kernel(const ScanArgs p)
{
    static_assert(SUB % 2 == 0, "the two accumulator sets alternate between column groups: an even number per tile");
    constexpr bool PIPE = (TM <= HM_SCAN_PIPE_MAX_TM);   // two accumulator sets fit the registers (128 stationary rows per wave: one set)
    constexpr int COLS = 32 * SUB;                 // partner rows per streamed tile
    constexpr int RS = hm_row_floats(NG);          // fp32 image: floats per row
    constexpr int RB16 = 16 * hm_row16_chunks(NG); // bf16 image: bytes per row (NG chunks of 8 bf16 [+ the [x0 fp32] chunk])
    constexpr bool HALF_LAST = (BF != 0) && (NG & 1);   // the last k-step covers one chunk only (8 K-slots)
    constexpr int ROW_BYTES = BF ? RB16 : RS * 4;
    constexpr int TILE_BYTES = COLS * ROW_BYTES;
    constexpr int NP = BF ? (NG + 1) / 2 : NG + 1; // k-steps: fp32: NG spatial groups + time; bf16: two chunks per step
    constexpr int NPIECE = TILE_BYTES / 1024;      // 1 KiB pieces per tile
    constexpr int TCH = RS / 4 - 1;                // fp32 image: chunk index of the time group
    constexpr int PPW = (NPIECE + WPB - 1) / WPB;  // LDS-DMA pieces per wave and tile (slot padded to PPW * WPB KiB)
    constexpr int TILE_LDS = PPW * WPB * 1024;     // LDS bytes per ring slot
    constexpr int DIST = BF ? HM_DIST_BF16 : 1;    // tiles in flight ahead of the one being computed
    constexpr int NBUF = DIST + 1;                 // ring slots
    constexpr int WAVE_ROWS = 32 * TM;
    constexpr int BLOCK_ROWS = WPB * WAVE_ROWS;
    constexpr int NTHREADS = 64 * WPB;
    static_assert(TILE_BYTES % 1024 == 0, "tiles are moved in 1 KiB pieces");
    static_assert(DIST * PPW <= 60, "the counted wait must fit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF * TILE_LDS (+ hist)

    if (p.stop != nullptr && *p.stop != 0u) return;     // a device-resident loop has ended
    if (p.order_seen != nullptr && __hip_atomic_load(p.order_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.order_need) {
        if (threadIdx.x == 0) __hip_atomic_store(p.order_fault, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;                                         // (pipelined loop: inputs not written yet -- see ScanArgs)
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    uint32_t sure_total = 0;             // per-lane partial of the sure count
    unsigned long long gk = ~0ull;       // last value read of the running argmin key
    unsigned long long gk_raw = ~0ull;   // destination of the asynchronous key load
    bool gk_pending = false;
    const uint32_t lds_base = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);

    uint32_t* lhist = nullptr;
    if (MODE == HM_MODE_HIST) {
        lhist = reinterpret_cast<uint32_t*>(smem + NBUF * TILE_LDS);
        for (int t = threadIdx.x; t < HM_HIST_BINS; t += NTHREADS) lhist[t] = 0;
    }

    // (row block, run of column tiles) being processed; the stationary rows in registers belong to row block rb_cur
    int rb = 0, ct0 = 0, ct1 = 0, rb_cur = -1;
    int i0w = 0;                        // first stationary row of this wave
    bool wave_active = false, rows_full = false;

    // The MFMA result u_f (plain fmaf chain) and the canonical u_c (torch reduction order) are two
    // roundings of the same exact form; |u_f - u_c| <= delta (gamma_n bound on both, |terms| <= rmax2).
    const float delta = hm_scan_delta(BF != 0, BF ? 8 * NG : RS, p.rmax2_bits);
    const float pre_f = p.u_hi + delta;              // candidate prefilter on u_f
    const float lo_f = p.u_lo - delta;               // u_f below this: canonical d < thr for sure
    const float zmax_f = 1.0f - delta;               // u_f at or below this: canonical u <= 1, d == 0
    const bool cut_all = (p.cut_bits == 0xffffffffu);
    const float cut_f = cut_all ? p.u_hi : hm::bitsf(p.cut_bits) + delta;

    // the running argmin key as it stands when the block starts (seeded, or found by earlier blocks)
    if (MODE == HM_MODE_ARGMIN) {
        const unsigned long long g0 = __hip_atomic_load(&p.ctr64[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g0 < gk) gk = g0;
    }

    // ---- stationary A fragments: lane (r, h) keeps its operands of every k-step in registers ----
    float2 a[BF ? 1 : TM][BF ? 1 : NP];            // fp32 form: 2 operands (two k-steps) per group
    uint4 a16[BF ? TM : 1][BF ? NP : 1];           // bf16 form: 8 bf16 per 16-wide k-step
    // bf16 form: the k-step g operand of a lane from its row pointer (row start + 16 * h): 8 K-slots = one 16-byte chunk;
    // the half step of an odd chunk count holds 4 slots per lane (8 bytes at chunk start + 8 * h)
    auto frag_at = [&](const auto* base, int g) -> uint4 {
        const char* q = reinterpret_cast<const char*>(base);
        if (HALF_LAST && g == NP - 1) {
            const uint2 v = *reinterpret_cast<const uint2*>(q + 32 * g - 8 * h);
            return make_uint4(v.x, v.y, 0u, 0u);
        }
        return *reinterpret_cast<const uint4*>(q + 32 * g);
    };
    auto mfma16 = [&](const uint4& av, const uint4& bv, const f32x16& cv, int g) -> f32x16 {
        if (HALF_LAST && g == NP - 1) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            const uint2 a2 = make_uint2(av.x, av.y), b2 = make_uint2(bv.x, bv.y);
            return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a2), __builtin_bit_cast(s16x4, b2), cv, 0, 0, 0);
        }
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), cv, 0, 0, 0);
    };
    auto load_a = [&]() {
    if constexpr (BF) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const unsigned char* src = p.img16 + (int64_t)(i0w + 32 * tm + r) * RB16 + 16 * h;
#pragma unroll
            for (int g = 0; g < NP; ++g) a16[tm][g] = frag_at(src, g);
            // stationary side of the time product: [hi, lo, hi, 0] -> [-hi, -hi, -lo, 0] (last 4 slots,
            // held by the h = 1 half of the last k-step)
            if (h == 1) {
                const uint32_t z = HALF_LAST ? a16[tm][NP - 1].x : a16[tm][NP - 1].z;
                const uint32_t hi16 = z & 0xffffu, lo16 = z >> 16;
                const uint32_t t0 = (hi16 | (hi16 << 16)) ^ 0x80008000u, t1 = lo16 ^ 0x8000u;
                if (HALF_LAST) { a16[tm][NP - 1].x = t0; a16[tm][NP - 1].y = t1; }
                else { a16[tm][NP - 1].z = t0; a16[tm][NP - 1].w = t1; }
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
    };

    // ---- LDS-DMA of one tile: wave w moves the PPW consecutive pieces [w * PPW, (w + 1) * PPW), four per
    // statement, through inline asm so that hipcc neither sees nor drains them; pieces past NPIECE (slot padding)
    // read the first KiB of the next tile: in bounds (the image is allocated with slack rows), unused.
    auto dma_tile = [&](int ct, int buf) {
        const char* src = (BF ? reinterpret_cast<const char*>(p.img16) : reinterpret_cast<const char*>(p.img)) +
                          (int64_t)ct * TILE_BYTES + lane * 16 + wave * (PPW * 1024);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)buf * (uint32_t)TILE_LDS + (uint32_t)wave * (PPW * 1024u));
        hm_dma_run<PPW>(src, dst);
    };

    // ---- per-group pieces ----
    f32x16 acc[PIPE ? 2 : 1][TM];                   // two accumulator sets (see the header) where they fit
    uint4 bpre16 = make_uint4(0, 0, 0, 0);          // first B fragment of the NEXT group, requested by the previous one
    // bf16 form with two accumulator sets: ALL k-step fragments of a column group are in registers before its MFMAs
    // start -- requested while the previous group of the tile computes (two register sets, alternating like the
    // accumulators).  With the fragments requested one k-step ahead, hipcc issued the LDS reads in pairs right
    // before their first use and the matrix pipe drained while they were in flight (ISA: ds_read x2, s_waitcnt,
    // 2 MFMA, s_waitcnt, 2 MFMA, ds_read x2, ...).
    constexpr bool DEEP = (BF != 0) && PIPE && (TM <= 2) && HM_SCAN_DEEP_PREFETCH;
    // (the one-set kernels -- 128-row waves -- have no registers left for a second fragment set: 60 spills when tried;
    // requesting all fragments of a group behind its first k-step measured 0.5 % slower than one-ahead requests there)
    uint4 bfr[2][DEEP ? NP : 1];
    float2 bpre = make_float2(0.f, 0.f);

    // all k-steps of column group `sub` of ring slot `buf` into acc[set].  `fresh`: the first fragment was not
    // requested by the previous group (first group of a tile, or the previous group was skipped).  The last k-step
    // always requests the first fragment of group sub + 1 (for the last group of a tile that lands in the slack
    // behind the slot and is never used).
    // `red_c` = true: the bound test's reduction over the OTHER accumulator set (the group finished before) is
    // folded into the k-steps, a few elements behind each step's MFMAs -- written into the same straight-line code
    // so that the vector ALU work rides in the matrix instructions' issue shadow; its result goes to `ext`.
    auto mma_group = [&](auto set_c, auto red_c, int buf, int sub, bool fresh, float& ext) {
        constexpr int set = decltype(set_c)::value;
        constexpr bool RED = decltype(red_c)::value;
        constexpr int QN = 16 * TM;
        if constexpr (RED) ext = acc[PIPE ? (set ^ 1) : 0][0][0];
        auto fold = [&](int g) {
            if constexpr (RED) {
#pragma unroll
                for (int q = (g * QN) / NP; q < ((g + 1) * QN) / NP; ++q) {
                    const float v = acc[PIPE ? (set ^ 1) : 0][q / 16][q % 16];
                    ext = SIGN ? __builtin_fmaxf(ext, v) : __builtin_fminf(ext, v);
                }
            }
        };
        if constexpr (BF) {
            const char* bt = smem + buf * TILE_LDS + (sub * 32 + r) * RB16 + 16 * h;
            if (fresh) bpre16 = frag_at(bt, 0);
            uint4 bc = bpre16, bn = bc;
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                bn = g + 1 < NP ? frag_at(bt, g + 1) : frag_at(bt + 32 * RB16, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    if (g == 0) {
                        f32x16 z;
#pragma unroll
                        for (int e = 0; e < 16; ++e) z[e] = 0.0f;
                        acc[set][tm] = mfma16(a16[tm][g], bc, z, g);
                    } else {
                        acc[set][tm] = mfma16(a16[tm][g], bc, acc[set][tm], g);
                    }
                }
                fold(g);
                bc = bn;
            }
            bpre16 = bn;
            // acc = S - x0*y0 (= -M): the time product came out of the last k-step's split slots
        } else {
            const float* bt = reinterpret_cast<const float*>(smem + buf * TILE_LDS) + (sub * 32 + r) * RS + 2 * h;
            if (fresh) bpre = *reinterpret_cast<const float2*>(bt);
            float2 bc = bpre, bn = bc;
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                bn = *reinterpret_cast<const float2*>(bt + (g + 1 < NP ? 4 * (g + 1 < NG ? g + 1 : TCH) : 32 * RS));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    if (g == 0) {
                        f32x16 z;
#pragma unroll
                        for (int e = 0; e < 16; ++e) z[e] = 0.0f;
                        acc[set][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].x, bc.x, z, 0, 0, 0);
                    } else {
                        acc[set][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].x, bc.x, acc[set][tm], 0, 0, 0);
                    }
                }
                if (g < NG) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        acc[set][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][g].y, bc.y, acc[set][tm], 0, 0, 0);
                }
                fold(g);
                bc = bn;
            }
            bpre = bn;
        }
        if constexpr (RED) ext = SIGN ? -ext : ext;
    };

    // The hot form of the bf16 kernel (both groups of a pass run and a finished group is pending): all k-step fragments
    // of the group are in bfr[set] before its MFMAs start -- requested by the previous group (`fresh` = false) -- and
    // the next group's go out to bfr[set ^ 1] first thing (`prefetch`), so the LDS latency is paid behind fourteen
    // MFMAs instead of in front of every second pair.  The other set's bound-test reduction rides along as in mma_group.
    auto mma_group_deep = [&](auto set_c, int buf, int sub, bool fresh, bool prefetch, float& ext) {
        constexpr int set = decltype(set_c)::value;
        constexpr int QN = 16 * TM;
        if constexpr (DEEP) {
            const char* bt = smem + buf * TILE_LDS + (sub * 32 + r) * RB16 + 16 * h;
            if (fresh) {
#pragma unroll
                for (int g = 0; g < NP; ++g) bfr[set][g] = frag_at(bt, g);
            }
            ext = acc[set ^ 1][0][0];
            auto kstep = [&](int g) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    if (g == 0) {
                        f32x16 z;
#pragma unroll
                        for (int e = 0; e < 16; ++e) z[e] = 0.0f;
                        acc[set][tm] = mfma16(a16[tm][g], bfr[set][g], z, g);
                    } else {
                        acc[set][tm] = mfma16(a16[tm][g], bfr[set][g], acc[set][tm], g);
                    }
                }
#pragma unroll
                for (int q = (g * QN) / NP; q < ((g + 1) * QN) / NP; ++q) {
                    const float v = acc[set ^ 1][q / 16][q % 16];
                    ext = SIGN ? __builtin_fmaxf(ext, v) : __builtin_fminf(ext, v);
                }
            };
            // the first k-step goes out before the next group's requests: LDS data returns in order, and a wait placed
            // behind fresh requests would hold the first MFMA until the first of THEM is back
            kstep(0);
            __builtin_amdgcn_sched_barrier(0);
            // unconditional (behind the last group of a tile the addresses fall into the next slot or the slack behind
            // the ring and the data is never used): a branch here makes hipcc count the waits of the MFMAs below for
            // the path WITHOUT the requests, which in the path with them means waiting for the requests themselves
            (void)prefetch;
#pragma unroll
            for (int g = 0; g < NP; ++g) bfr[set ^ 1][g] = frag_at(bt + 32 * RB16, g);
            __builtin_amdgcn_sched_barrier(0);      // the requests stay in front of the remaining MFMAs
#pragma unroll
            for (int g = 1; g < NP; ++g) kstep(g);
            ext = SIGN ? -ext : ext;
        }
    };

    // the lane's most promising u of accumulator set `set` (acc = -M: u = acc under the reference sign, -acc under lorentz)
    auto reduce_group = [&](auto set_c) -> float {
        constexpr int set = decltype(set_c)::value;
        float ext = acc[set][0][0];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) ext = SIGN ? __builtin_fmaxf(ext, acc[set][tm][e]) : __builtin_fminf(ext, acc[set][tm][e]);
        return SIGN ? -ext : ext;
    };

    // bound test of a finished group and, when some lane is below the bound, the slow path: one 32x32 MFMA tile at
    // a time (a rolled loop: the hot loop must not inherit its register pressure).  Per-element predicates are
    // evaluated twice (count, then write); the second evaluation runs on laundered copies of the bounds so that the
    // compiler does not keep the predicates alive across the wave scan.
    auto finish_group = [&](auto set_c, float ext_u, int j0s) {
        constexpr int set = decltype(set_c)::value;
        float bound_f = pre_f;
        uint32_t best_bits = 0xffffffffu, best_low = 0xffffffffu;
        if (MODE == HM_MODE_TOPK && !p.count_sure && !cut_all && cut_f < bound_f) bound_f = cut_f;   // nothing above the cut is visited
        if (MODE == HM_MODE_ARGMIN) {
            // an entry can only order before the running best if its u_f is within 2*delta (+ a few ulps of acosh
            // wiggle) of the best u_f
            best_bits = (uint32_t)(gk >> 32);
            best_low = (uint32_t)gk;
            if (best_bits != 0xffffffffu) {
                const float bb = hm::bitsf(best_bits + hm_tie_slack(best_bits)) + 2.0f * delta;
                if (bb < bound_f) bound_f = bb;
            }
        }
        if (__ballot(ext_u < bound_f) == 0ull) return;
        if (MODE == HM_MODE_ARGMIN && best_bits == 0x3f7fffffu && p.thr_pos != 0) {
            // The running best is a pair at distance exactly 0 (the literal sign mode: EVERY pair is one).  A pair that is
            // certainly at distance 0 too is only emitted when its (i, j) orders before the best's (`low <= best_low`
            // below); every pair of this group has low >= lowmin.  So when the whole group is certainly-zero and starts
            // behind the best pair, the slow path would emit nothing: skip it (1.6 ms -> 0.3 ms per tie-flood scan).
            const uint32_t lowmin = ((uint32_t)i0w << 15) | ((uint32_t)j0s >> 2);
            if (lowmin > best_low) {
                float worst = SIGN ? -acc[set][0][0] : acc[set][0][0];          // the LARGEST u of the lane's elements
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float u = SIGN ? -acc[set][tm][e] : acc[set][tm][e];
                        worst = __builtin_fmaxf(worst, u);
                    }
                if (__ballot(!(worst <= zmax_f)) == 0ull) return;                // (NaN counts as not certainly zero)
            }
        }
        const bool full = rows_full && (j0s > i0w + WAVE_ROWS - 1) && (j0s + 31 < p.n) && (j0s >= p.col_begin);
        unsigned long long wkey = ~0ull;
        bool wrote = false;
#pragma unroll 1
        for (int st = 0; st < TM; ++st) {
            // element-wise select chain: a whole-vector `if (st == q) w = acc[q]` makes hipcc keep the accumulators
            // in scratch memory in some instantiations (3x slower hot loop)
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[set][0][e];
#pragma unroll
                for (int q = 1; q < TM; ++q) v = (st == q) ? acc[set][q][e] : v;
                w[e] = v;
            }
            if (TM > 1) {
                float e1 = w[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) e1 = SIGN ? __builtin_fmaxf(e1, w[e]) : __builtin_fminf(e1, w[e]);
                if (__ballot((SIGN ? -e1 : e1) < bound_f) == 0ull) continue;
            }
            uint32_t n_emit = 0, n_sure = 0;
            float bnd = bound_f;
            float cutv = cut_f;
            unsigned long long slot = 0;
            const int ib = i0w + 32 * st + 4 * h;
            const int j = j0s + r;
            auto visit = [&](const float wv, const int e, const bool write) {
                const float u = SIGN ? -wv : wv;
                const int i = ib + (e & 3) + 8 * (e >> 2);
                bool pass = u < bnd;
                if (!full) pass = pass && (i < j) && (j < p.n) && (j >= p.col_begin) && (i >= p.row_begin) && (i < p.row_end);
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
                    emit = zero ? (i <= p.tie_imax) : ((!sure && p.count_sure) || cut_all || up <= cutv);
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
                if (slot < (unsigned long long)p.ent_cap) p.ent[slot] = make_uint4(ub, (uint32_t)i, (uint32_t)j, flag);
                ++slot;
            };
#pragma unroll
            for (int e = 0; e < 16; ++e) visit(w[e], e, false);
            if (MODE == HM_MODE_HIST) continue;
            sure_total += n_sure;
            const uint32_t incl = hm_wave_incl_scan(n_emit, lane);
            const uint32_t total = __shfl(incl, 63, 64);
            if (total == 0) continue;
            unsigned long long base = 0;
            if (lane == 63) base = atomicAdd(&p.ctr64[2], (unsigned long long)total);     // 64-bit: a tie flood cannot wrap it
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
            // the slow path has drained the vector-memory queue anyway: refresh the running key now
            gk = __hip_atomic_load(&p.ctr64[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wk < gk) gk = wk;
        }
    };

    // ---- work distribution: block b owns item b of the host's list (hm_prepare_scan); with the item queue (p.dyn) the grid
    // is the resident set and every block goes on drawing items of the same list, in order, from the queue word p.q_ctr ----
    constexpr int QOFF = NBUF * TILE_LDS + 32 * ROW_BYTES + (MODE == HM_MODE_HIST ? (int)sizeof(uint32_t) * HM_HIST_BINS : 0);
    volatile uint32_t* const qslot = reinterpret_cast<volatile uint32_t*>(smem + QOFF);      // the drawn counter value, for all waves
    const bool dyn = (MODE != HM_MODE_HIST) && (p.dyn != 0);
    unsigned long long q_raw = 0ull;     // destination of the asynchronous draw (thread 0)
    if (dyn && threadIdx.x == 0) (void)__hip_atomic_fetch_max(p.q_ctr, p.q_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int item = (int)blockIdx.x;
#pragma unroll 1
    for (;;) {
    bool have = true;                   // the item has tiles right of the diagonal
    {
        int it = item, ph = 0;
        while (ph + 1 < p.n_ph && it >= p.ph_items[ph]) { it -= p.ph_items[ph]; ++ph; }
        rb = p.ph_rb0[ph] + it / p.ph_chunks[ph];
        ct0 = p.ph_ctmin[ph] + (it % p.ph_chunks[ph]) * p.ph_ch[ph];
        ct1 = ct0 + p.ph_ch[ph];
    }
    if (ct0 < (rb * BLOCK_ROWS) / COLS) ct0 = (rb * BLOCK_ROWS) / COLS;   // left of the diagonal: no i < j
    if (ct0 < p.col_begin / COLS) ct0 = p.col_begin / COLS;               // partner rows in front of col_begin are not asked for
    if (ct1 > p.nct) ct1 = p.nct;
    if (ct0 >= ct1) have = false;

    // HIST mode visits every sample_stride-th tile only (a cheap estimate of the u' distribution)
    int ct_step = 1;
    if (MODE == HM_MODE_HIST && have) {
        ct_step = p.sample_stride;
        ct0 += (ct_step - (ct0 + rb * 7) % ct_step) % ct_step;
        if (ct0 >= ct1) have = false;
    }
    if (have) {
    if (rb != rb_cur) {
        rb_cur = rb;
        i0w = rb * BLOCK_ROWS + wave * WAVE_ROWS;
        wave_active = (i0w < p.row_end) && (i0w + WAVE_ROWS - 1 >= p.row_begin) && (i0w < p.n);
        rows_full = (i0w >= p.row_begin) && (i0w + WAVE_ROWS - 1 < p.row_end);
        load_a();
    }
    const int ntile = (ct1 - ct0 + ct_step - 1) / ct_step;
    auto tile_at = [&](int t) { return ct0 + t * ct_step; };

    // ring prologue: tiles 0 .. DIST-1 in flight (a repeat of the last tile when the run is shorter), tile 0 landed
#pragma unroll
    for (int q = 0; q < DIST; ++q) dma_tile(tile_at(q < ntile ? q : ntile - 1), q);
    if (DIST - 1 <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * PPW) : "memory");
    // hipcc waits for its own loads (the A fragments above) lazily, at their first use INSIDE the loop -- with
    // vmcnt(N) instructions that also wait for the ring's LDS-DMA (which it cannot see) on every iteration.
    // A wait it can see, here, settles them before the loop is entered.
    if (DIST == 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    __syncthreads();

    bool pend = false;               // a finished group waits in the other accumulator set for its bound test
    bool deep_valid = false;         // bfr[0] holds the fragments of the group about to run (requested by its predecessor)
    int pend_j0s = 0;
    int buf = 0;                     // ring slot of tile t

    for (int t = 0; t < ntile; ++t) {
        const int ct = tile_at(t);
        const int ct_next = (t + DIST < ntile) ? tile_at(t + DIST) : ct;      // (a repeat of this tile lands in the free slot)
        int buf_next = buf + DIST;                           // slot of tile t + DIST = slot of tile t - 1:
        if (buf_next >= NBUF) buf_next -= NBUF;              // every wave left it at the previous barrier
        const int j0 = ct * COLS;

        // running best key, re-read every 8th tile by a load the compiler does not see (it would wait for it with
        // vmcnt(0) inside the MFMA loop and so drain the ring): the value is picked up behind this iteration's own
        // counted wait.  Issued ahead of the tile's DMA, so that wait covers it.
        if (MODE == HM_MODE_ARGMIN && (t & 7) == 0) {
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(gk_raw) : "v"(&p.ctr64[1]) : "memory");
            gk_pending = true;
        }
        // item queue: the next item is drawn during this item's last tile (picked up behind the same counted wait)
        if (dyn && t == ntile - 1 && threadIdx.x == 0)
            asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0" : "=v"(q_raw) : "v"(p.q_ctr), "v"(1ull) : "memory");
        dma_tile(ct_next, buf_next);

        if constexpr (!PIPE) {
            // one accumulator set: MFMAs, then the bound test of the same group (the other resident block of the CU
            // fills the matrix pipe meanwhile); the first fragment of the next group is still requested early
            bool prev = false;
#pragma unroll 1
            for (int sub = 0; sub < SUB; ++sub) {
                const int j0s = j0 + sub * 32;
                const bool do_c = wave_active && (j0s + 31 > i0w);
                if (do_c) {
                    float ext_u = 0.0f;
                    mma_group(std::integral_constant<int, 0>{}, std::false_type{}, buf, sub, !prev, ext_u);
                    ext_u = reduce_group(std::integral_constant<int, 0>{});
                    finish_group(std::integral_constant<int, 0>{}, ext_u, j0s);
                }
                prev = do_c;
            }
        } else {
            // two groups per pass, the accumulator sets alternating (a rolled loop: two code sites for the slow path)
#pragma unroll 1
            for (int sp = 0; sp < SUB / 2; ++sp) {
                if constexpr (DEEP) {
                    const int ja = j0 + 2 * sp * 32, jb = ja + 32;
                    if (wave_active && (ja + 31 > i0w) && pend) {     // both groups of the pass run (jb > ja) and one is pending
                        float ea = 0.0f, eb = 0.0f;
                        mma_group_deep(std::integral_constant<int, 0>{}, buf, 2 * sp, !deep_valid, true, ea);
                        finish_group(std::integral_constant<int, 1>{}, ea, pend_j0s);
                        const bool more = 2 * sp + 2 < SUB;
                        mma_group_deep(std::integral_constant<int, 1>{}, buf, 2 * sp + 1, false, more, eb);
                        finish_group(std::integral_constant<int, 0>{}, eb, ja);
                        pend_j0s = jb;
                        deep_valid = more;                            // bfr[0] holds the fragments of the tile's next group
                        continue;
                    }
                    deep_valid = false;
                }
                {
                    const int sub = 2 * sp;
                    const int j0s = j0 + sub * 32;
                    const bool do_c = wave_active && (j0s + 31 > i0w);
                    float ext_u = 0.0f;
                    if (do_c && pend) {                           // the common case: one basic block, MFMAs + the other set's test
                        mma_group(std::integral_constant<int, 0>{}, std::true_type{}, buf, sub, sp == 0, ext_u);
                    } else {
                        if (do_c) mma_group(std::integral_constant<int, 0>{}, std::false_type{}, buf, sub, true, ext_u);
                        if (pend) ext_u = reduce_group(std::integral_constant<int, 1>{});
                    }
                    if (pend) finish_group(std::integral_constant<int, 1>{}, ext_u, pend_j0s);
                    pend = do_c;
                    pend_j0s = j0s;
                }
                {
                    const int sub = 2 * sp + 1;
                    const int j0s = j0 + sub * 32;
                    const bool do_c = wave_active && (j0s + 31 > i0w);
                    float ext_u = 0.0f;
                    if (do_c && pend) {                           // the previous group ran: its last k-step requested our first fragment
                        mma_group(std::integral_constant<int, 1>{}, std::true_type{}, buf, sub, false, ext_u);
                    } else {
                        if (do_c) mma_group(std::integral_constant<int, 1>{}, std::false_type{}, buf, sub, true, ext_u);
                        if (pend) ext_u = reduce_group(std::integral_constant<int, 0>{});
                    }
                    if (pend) finish_group(std::integral_constant<int, 0>{}, ext_u, pend_j0s);
                    pend = do_c;
                    pend_j0s = j0s;
                }
            }
        }

        // tile t+1 has landed (this wave's pieces) -- and so has the key load, if one was issued
        if (DIST - 1 <= 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gk_raw), "+v"(q_raw) : : "memory");
        else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gk_raw), "+v"(q_raw) : "n"((DIST - 1) * PPW) : "memory");
        if (MODE == HM_MODE_ARGMIN && gk_pending) {
            if (gk_raw < gk) gk = gk_raw;                   // the key only ever decreases
            gk_pending = false;
        }
        if (dyn && t == ntile - 1 && threadIdx.x == 0) { qslot[0] = (uint32_t)q_raw; qslot[1] = (uint32_t)(q_raw >> 32); }
        __syncthreads();                                     // ... and every wave's; all reads of slot `buf` done
        if (++buf == NBUF) buf = 0;
    }
    if constexpr (PIPE) {
        if (pend) {                                          // the last group of the run (set (SUB - 1) % 2 = 1)
            const float ext_u = reduce_group(std::integral_constant<int, 1>{});
            finish_group(std::integral_constant<int, 1>{}, ext_u, pend_j0s);
        }
    }

    if (DIST > 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // nothing of this run may land later
    }   // have

    if (!dyn) break;
    if (!have) {
        // an item left of the diagonal: draw at once (the barrier in front keeps the slot's previous value readable
        // until every wave has it)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long v = __hip_atomic_fetch_add(p.q_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qslot[0] = (uint32_t)v; qslot[1] = (uint32_t)(v >> 32);
        }
        __syncthreads();
    }
    {
        const uint32_t qlo = __builtin_amdgcn_readfirstlane(qslot[0]), qhi = __builtin_amdgcn_readfirstlane(qslot[1]);
        const unsigned long long q = ((unsigned long long)qhi << 32) | qlo;
        if ((q >> 24) != (p.q_tag >> 24)) break;            // (a foreign tag: never within one launch's lifetime)
        item = (int)gridDim.x + (int)(q & 0xffffffull);
        if (item >= p.n_items) break;
    }
    }   // items

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
// launch
// ------------------------------------------------------------------------------------------------
template <int NG, int SIGN, int MODE, int BF, int TM, int WPB, int SUB>
static hipError_t hm_launch_scan_t(hm_engine* e, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const size_t row_bytes = BF ? 16 * hm_row16_chunks(NG) : 4 * hm_row_floats(NG);
    const size_t tile_bytes = (size_t)32 * SUB * row_bytes;
    const size_t ppw = (tile_bytes / 1024 + WPB - 1) / WPB;
    size_t lds = ((BF ? HM_DIST_BF16 : 1) + 1) * ppw * WPB * 1024;
    lds += (size_t)32 * row_bytes;                                          // slack behind the ring: the early request of "the next group's" fragment
    if (MODE == HM_MODE_HIST) lds += sizeof(uint32_t) * HM_HIST_BINS;      // (HIST mode keeps its bins there; they are only read by that request)
    lds += 16;                                                              // the item queue's broadcast word
    const void* fn = reinterpret_cast<const void*>(&hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, SUB>);
    if (lds > 48 * 1024 && e->attr_done.find(fn) == e->attr_done.end()) {     // per engine (= per device), not process-wide
        hipError_t st = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
        e->attr_done.insert(fn);
    }
    // timed launches carry their events in the dispatch itself (start / stop timestamps of this kernel): a pair of
    // hipEventRecord calls around it costs two ~6 us bubbles on the stream
    ScanArgs b = a;
    b.n_items = (int)grid.x;
    b.dyn = 0;
    // item queue (ScanArgs::dyn): every launch with more items than resident slots, both prefilter forms (measured:
    // profiles/r03g_ab_item_queue_*: -3 % launch time for the bf16 form at V = 50 k / 100 k, -2.8 % for the fp32 form)
    if (e->dyn_queue && MODE != HM_MODE_HIST && grid.x < (1u << 24)) {
        auto it = e->scan_slots.find(fn);
        if (it == e->scan_slots.end()) {
            int per_cu = 0;
            hipError_t st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * WPB, lds);
            if (st != hipSuccess) return st;
            it = e->scan_slots.emplace(fn, std::max(1, per_cu) * e->n_cu).first;
        }
        const int slots = e->dyn_slots > 0 ? e->dyn_slots : it->second;
        if ((int)grid.x > slots) {
            grid.x = (unsigned)slots;
            b.dyn = 1;
            b.q_tag = (++e->scan_tag) << 24;
            b.q_ctr = e->d_queue + 16 * ((b.ctr64 == e->d_ctr64) ? 0 : 1);      // one word per counter set (the pipelined loop alternates them)
        }
    }
    if (ev0 != nullptr || ev1 != nullptr) hipExtLaunchKernelGGL((hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, SUB>), grid, dim3(64 * WPB), lds, s, ev0, ev1, 0, b);
    else hipLaunchKernelGGL((hm_scan_kernel<NG, SIGN, MODE, BF, TM, WPB, SUB>), grid, dim3(64 * WPB), lds, s, b);
    return hipGetLastError();
}

template <int NG, int BF, int TM, int WPB, int SUB>
static hipError_t hm_launch_scan_ng(hm_engine* e, int sign, int mode, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0,
                                    hipEvent_t ev1)
{
    if (sign) {
        if (mode == HM_MODE_TOPK) return hm_launch_scan_t<NG, 1, HM_MODE_TOPK, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
        if (mode == HM_MODE_ARGMIN) return hm_launch_scan_t<NG, 1, HM_MODE_ARGMIN, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
        return hm_launch_scan_t<NG, 1, HM_MODE_HIST, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
    }
    if (mode == HM_MODE_TOPK) return hm_launch_scan_t<NG, 0, HM_MODE_TOPK, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
    if (mode == HM_MODE_ARGMIN) return hm_launch_scan_t<NG, 0, HM_MODE_ARGMIN, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
    return hm_launch_scan_t<NG, 0, HM_MODE_HIST, BF, TM, WPB, SUB>(e, a, grid, s, ev0, ev1);
}

// Which prefilter form a scan uses.  The bf16 form's error bound 0.00782 * max||x_s||^2 only costs
// extra emissions, never correctness; auto picks it from d >= 24 (below that the fp32 form is already short).
bool hm_use_bf16(const hm_engine* e)
{
    if (!e->bf16_ok || e->force_f32) return false;
    if (e->precision == HM_PREFILTER_F32) return false;
    if (e->precision == HM_PREFILTER_BF16) return true;
    return e->d >= 24;
}

// bf16 block shapes: 0 = 4 waves x 64 rows (256-row blocks, two accumulator sets), 1 = 4 waves x 128 rows (512-row
// blocks: half the L2 -> LDS fill traffic and half the LDS fragment reads per flop, one accumulator set),
// 2 = 8 x 64, 3 = 8 x 128 (tuning builds only)
#ifndef HM_SCAN_ALL_SHAPES
#define HM_SCAN_ALL_SHAPES 0
#endif
#define HM_BF16_CASES(TMv, WPBv)                                                                                           \
    switch (e->KC) {                                                                                                       \
        HM_BF16_SMALL(TMv, WPBv)                                                                                           \
        case 13: return hm_launch_scan_ng<13, 1, TMv, WPBv, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);    \
    }                                                                                                                      \
    return hipErrorInvalidValue;
#if HM_SCAN_INSTANTIATE_ALL
#define HM_BF16_SMALL(TMv, WPBv)                                                                                           \
        case 2: return hm_launch_scan_ng<2, 1, TMv, WPBv, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);      \
        case 4: return hm_launch_scan_ng<4, 1, TMv, WPBv, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);      \
        case 8: return hm_launch_scan_ng<8, 1, TMv, WPBv, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);      \
        case 14: return hm_launch_scan_ng<14, 1, TMv, WPBv, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
#else
#define HM_BF16_SMALL(TMv, WPBv)
#endif

hipError_t hm_launch_scan(hm_engine* e, int mode, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    if (a.bf16 && a.shape == 1) { HM_BF16_CASES(4, 4) }
#if HM_SCAN_ALL_SHAPES
    if (a.bf16 && a.shape == 2) { HM_BF16_CASES(2, 8) }
    if (a.bf16 && a.shape == 3) { HM_BF16_CASES(4, 8) }
    if (a.bf16 && a.shape == 4) { HM_BF16_CASES(3, 4) }
#endif
    if (a.bf16) {
        if (e->KC == 16) {
#if HM_SCAN_INSTANTIATE_ALL
            return hm_launch_scan_ng<16, 1, 2, 4, HM_SUB_BF16>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
#else
            return hipErrorInvalidValue;
#endif
        }
        HM_BF16_CASES(2, 4)
    }
    switch (e->NG) {
#if HM_SCAN_INSTANTIATE_ALL
        case 1: return hm_launch_scan_ng<1, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 2: return hm_launch_scan_ng<2, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 3: return hm_launch_scan_ng<3, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 4: return hm_launch_scan_ng<4, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 6: return hm_launch_scan_ng<6, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 8: return hm_launch_scan_ng<8, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 10: return hm_launch_scan_ng<10, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 13: return hm_launch_scan_ng<13, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 16: return hm_launch_scan_ng<16, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 20: return hm_launch_scan_ng<20, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 28: return hm_launch_scan_ng<28, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
        case 32: return hm_launch_scan_ng<32, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
#endif
        case 25: return hm_launch_scan_ng<25, 0, 2, 4, HM_SUB_F32>(e, e->sign_mode, mode, a, grid, s, ev0, ev1);
    }
    return hipErrorInvalidValue;
}

// common argument preparation; returns false when the row range is empty
bool hm_prepare_scan(hm_engine* e, const Bounds& b, int64_t row_begin, int64_t row_end, ScanArgs& a, dim3& grid, int64_t n_limit,
                     int64_t col_begin)
{
    // n_limit: search among the first n_limit rows only (rows are only ever appended: the pairs of an earlier table)
    const int64_t n = (n_limit >= 0 && n_limit < e->n) ? n_limit : e->n;
    if (row_end < 0 || row_end > n) row_end = n;
    if (row_begin < 0) row_begin = 0;
    if (row_end > n - 1) row_end = n - 1;        // the last row has no partner j > i
    if (row_begin >= row_end) return false;
    memset(&a, 0, sizeof(a));
    a.img = e->img;
    a.img16 = e->img16;
    a.bf16 = hm_use_bf16(e) ? 1 : 0;
    // large launches: 512-row blocks halve the L2 -> LDS fill traffic, which is what limits the bf16 form once the
    // launch tail no longer does (decided by the pairs this launch covers: a row-range search of a sharded run is
    // a small launch)
    // (16 chunks would not fit the registers of the 128-row waves)
    a.shape = (a.bf16 && e->KC <= 14 && hm_pairs_in_range(n, row_begin, row_end) >= e->big_min_rows * (e->big_min_rows - 1) / 2) ? 1 : 0;
    if (a.bf16 && e->KC <= 14 && e->force_shape >= 0) a.shape = e->force_shape;
    static const int kShapeRows[5] = {256, 512, 512, 1024, 384};
    const int block_rows = a.bf16 ? kShapeRows[a.shape] : 256;
    const int cols = 32 * (a.bf16 ? HM_SUB_BF16 : HM_SUB_F32);     // partner rows per streamed tile
    a.n = (int)n;
    a.row_begin = (int)row_begin;
    a.row_end = (int)row_end;
    a.rb_first = (int)(row_begin / block_rows);
    a.nct = (int)((n + cols - 1) / cols);
    a.u_hi = b.u_hi;
    a.u_lo = b.u_lo;
    a.thr_pos = b.thr_pos;
    a.count_sure = 1;
    a.cut_bits = 0xffffffffu;
    a.tie_imax = 0x7fffffff;
    a.ent = e->ent;
    a.ent_cap = e->ent_cap;
    a.ctr64 = e->d_ctr64;
    a.hist = e->d_hist;
    a.sample_stride = 1;
    a.rmax2_bits = e->d_rmax2;
    a.stop = nullptr;
    const int rb_last = (int)((row_end - 1) / block_rows);
    const int nrb = rb_last - a.rb_first + 1;
    // diagonal advance per row block in tiles (rounded down: the kernel clamps to the exact diagonal)
    const int tiles_per_rb = std::max(1, block_rows / cols);
    // column tiles per block: amortise the stationary-row load, but keep enough blocks in flight.
    // (the knob is in 64-column units)
    int ch = (a.bf16 ? e->chunk_bf16 : e->chunk_f32) * 64 / cols;
    if (ch < 1) ch = 1;
    while (ch > 4 && (int64_t)nrb * ((a.nct + ch - 1) / ch) < 1024) ch >>= 1;
    // phases: cut the launch's row blocks (top down: longest rows first) at the cumulative work shares; chunk lengths
    // ch, ch / div[1], ch / div[2], ... (a small launch keeps one phase)
    a.col_begin = (int)std::max<int64_t>(col_begin, 0);
    const int ct_lo = a.col_begin / cols;
    int nph = 1;
    double share[HM_SCAN_PHASES] = {1.0};
    int div[HM_SCAN_PHASES];
    for (int q = 0; q < HM_SCAN_PHASES; ++q) div[q] = 1;
    if (ch >= 8 && nrb >= 16) {
        if (e->phases <= 2) {
            nph = 2;
            share[0] = 1.0 - e->tail_fraction; share[1] = e->tail_fraction;
            div[1] = e->tail_div;
        } else {
            nph = std::min(e->phases, HM_SCAN_PHASES);
            double rest = 1.0;
            for (int q = 0; q < nph - 1; ++q) { share[q] = e->ph_share[q]; rest -= share[q]; }
            share[nph - 1] = std::max(rest, 0.0);
            for (int q = 0; q < nph; ++q) div[q] = e->ph_div[q];
        }
    }
    double total = 0.0;
    for (int rb = a.rb_first; rb <= rb_last; ++rb) total += (double)std::max(0, a.nct - rb * tiles_per_rb);
    a.n_ph = nph;
    int rb_at = a.rb_first, n_items = 0;
    double acc = 0.0, want = 0.0;
    for (int q = 0; q < nph; ++q) {
        want += share[q] * total;
        int rb_end = rb_at;
        if (q == nph - 1) rb_end = rb_last + 1;
        else while (rb_end <= rb_last && acc < want) { acc += (double)std::max(0, a.nct - rb_end * tiles_per_rb); ++rb_end; }
        const int chq = std::max(1, ch / std::max(1, div[q]));
        a.ph_rb0[q] = rb_at;
        a.ph_ch[q] = chq;
        a.ph_ctmin[q] = std::max((int)((int64_t)rb_at * block_rows / cols), ct_lo);
        a.ph_chunks[q] = std::max(1, (a.nct - a.ph_ctmin[q] + chq - 1) / chq);
        a.ph_items[q] = (rb_end - rb_at) * a.ph_chunks[q];
        n_items += a.ph_items[q];
        rb_at = rb_end;
    }
    grid = dim3((unsigned)std::max(1, n_items), 1, 1);
    return true;
}
