// hm_search.hip -- exact re-evaluation and selection of what the pair scan emitted, and the search entry
// points of the C ABI (hm_pairwise_argmin / _argmin_dev / _topk / _candidates, hm_row_argmin).
//
// The scan (hm_scan.hip) only SELECTS: every reported distance is recomputed here in the canonical fp32
// arithmetic from the fp32 image -- a half-wave per entry (coalesced 128-byte row reads,
// ATen's 32 summation chains one per lane), thresholds compare in fp32, order = (distance bits, i, j).
#include "hm_common.h"
#include "hm_rows_device.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// argmin tail: re-evaluation + final record + seed + arming (+ the merge itself when asked for): ONE launch
// ------------------------------------------------------------------------------------------------
struct TailArgs {
    const uint4* ent;
    unsigned long long* ctr64;       // [1] running key (re-armed), [2] emitted (read, re-armed)
    uint32_t cap;
    float* img;
    unsigned char* img16;
    int RS, d, KC, sign_mode;
    float sqrt_c, thr;
    ArgminPart* parts;
    uint32_t* ticket;
    ArgminRec* out;                  // record
    ArgminRec* out2;                 // may be NULL: {found = emitted low, dbits = emitted high}
    ArgminSeed* seed;                // may be NULL: no seed update
    int bf, kterms;
    uint32_t* rmax2_bits;
    int arm, arm_rb, arm_re;         // leave counters + running key ready for the next search of rows [arm_rb, arm_re)
    MergeFuse mf;
};

__global__ __launch_bounds__(HM_TAIL_THREADS) void hm_argmin_tail_kernel(const TailArgs a)
{
    __shared__ uint32_t s0[HM_TAIL_THREADS / 64], s1[HM_TAIL_THREADS / 64], s2[HM_TAIL_THREADS / 64];
    __shared__ uint32_t s_last;
    __shared__ MidScratch ms;
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_s_setprio(3);          // in the pipelined loop this latency-bound chain shares its SIMDs with the next scan's MFMA waves
    LoopState* loop = a.mf.loop;
    if (loop != nullptr && loop->stop != 0u) {                 // the loop ended at an earlier step: this one is skipped
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ArgminRec r; r.found = 3u; r.dbits = 0; r.i = 0xffffffffu; r.j = 0xffffffffu;
            *a.out = r;
            if (a.mf.rec_ring) *a.mf.rec_ring = r;
        }
        return;
    }
    const unsigned long long emitted = a.ctr64[2];
    // (an overflowed search reports found = 2 whatever its entries say: none of them is read.  The pipelined loop's small
    // tail grid -- it runs beside a scan -- hands a step with more than HM_PIPE_TAIL_MAX survivors back the same way: the
    // host redoes that one step through hm_pairwise_argmin and its full-size tail)
    const bool over_small = a.mf.rowkey != nullptr && emitted <= (unsigned long long)a.cap && emitted > (unsigned long long)HM_PIPE_TAIL_MAX;
    const bool over = emitted > (unsigned long long)a.cap || over_small;
    const uint32_t m = over ? 0u : (uint32_t)emitted;
    uint32_t active = (m + HM_TAIL_SOLO - 1) / HM_TAIL_SOLO;       // HM_TAIL_SOLO = entries one block takes per two rounds
    if (active < 1) active = 1;
    if (active > gridDim.x) active = gridDim.x;
    if (blockIdx.x >= active) return;

    // ---- exact distance of this block's entries, lexicographic min of (d bits, i, j) with d < thr ----
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu;
    // a half-wave takes HM_GATHER consecutive entries per round (hm_halfwave_gather): lane t finishes entry t
    // (the block size is the launch's: 1024 threads, or 256 in the pipelined loop where the kernel has to fit beside a scan block)
    const uint32_t hw = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;               // half-wave id
    const uint32_t stride = active * (blockDim.x >> 5) * HM_GATHER;
    const int t32 = lane & 31;
    for (uint32_t base = (hw & ~1u) * HM_GATHER; base < m; base += stride) {             // wave-uniform trip count
        const uint32_t mybase = base + (hw & 1u) * HM_GATHER;
        const uint32_t mine = mybase + t32;
        const uint4 en = a.ent[mine < m ? mine : m - 1];
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const uint32_t ri = __shfl(en.y, (lane & 32) + k, 64), rj = __shfl(en.z, (lane & 32) + k, 64);
            return hm_img_u_halfwave(a.img, a.RS, a.d, ri, rj, a.sign_mode, lane);
        });
        const float dd = hm::dist_from_u(u, a.sqrt_c);
        if (t32 < HM_GATHER && mine < m && dd < a.thr) {
            const uint32_t db = hm::fbits(dd);
            if (hm_key_less(db, en.y, en.z, b0, b1, b2)) { b0 = db; b1 = en.y; b2 = en.z; }
        }
    }
    hm_block_min_key(b0, b1, b2, s0, s1, s2);

    if (active > 1) {
        // several blocks took part: partial records, then the last block to arrive reduces them (agent-scope
        // release / acquire around a ticket; rare path -- more than HM_TAIL_SOLO survivors)
        if (threadIdx.x == 0) {
            ArgminPart pt; pt.dbits = b0; pt.i = b1; pt.j = b2; pt.pad = 0;
            a.parts[blockIdx.x] = pt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == active - 1) ? 1u : 0u;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *a.ticket = 0u;
            }
        }
        __syncthreads();
        if (s_last == 0u) return;
        b0 = b1 = b2 = 0xffffffffu;
        if (threadIdx.x < active) {
            const ArgminPart pt = a.parts[threadIdx.x];
            b0 = pt.dbits; b1 = pt.i; b2 = pt.j;
        }
        hm_block_min_key(b0, b1, b2, s0, s1, s2);
    }

    // pipelined loop: the pairs of the newest row were searched by their own small kernel
    if (a.mf.rowkey != nullptr) {
        const unsigned long long rk = *a.mf.rowkey;
        if (rk != ~0ull) {
            const uint32_t db = (uint32_t)(rk >> 32), ri = (uint32_t)rk, rj = a.mf.rowkey_j;
            if (hm_key_less(db, ri, rj, b0, b1, b2)) { b0 = db; b1 = ri; b2 = rj; }
        }
    }
    // ---- final (one block): record, seed, arming, merge ----
    // found = 2: the emission buffer overflowed, the record is not final (the caller reruns bounded)
    const uint32_t found = over ? 2u : ((b1 != 0xffffffffu) ? 1u : 0u);
    if (threadIdx.x >= 64) return;                              // wave 0 finishes (b0, b1, b2 are block-uniform)
    const uint32_t bi = b1, bj = b2;
    if (lane == 0) {
        ArgminRec r; r.found = found; r.dbits = b0; r.i = bi; r.j = bj;
        *a.out = r;
        if (a.mf.rec_ring) *a.mf.rec_ring = r;
        if (a.out2) { a.out2->found = (uint32_t)emitted; a.out2->dbits = (uint32_t)(emitted >> 32); }
    }
    unsigned long long seed_key = ~0ull;
    uint32_t seed_row = 0xffffffffu;
    if (a.seed != nullptr) {
        if (found == 1u) {
            // a final record becomes the next search's seed: key = (bits(u_c + delta), all ones) -- the pair's own
            // prefilter value is <= u_c + delta, and an entry the scan skips on this key has u_f >= u_c + 3 delta, so it
            // cannot order before the pair -- or, for a pair at distance 0 (u_c <= 1), the exact zero-class key
            const float u = hm_img_u_halfwave(a.img, a.RS, a.d, bi, bj, a.sign_mode, lane);
            if (u <= 1.0f) seed_key = (0x3f7fffffull << 32) | (unsigned long long)((bi << 15) | (bj >> 2));
            else seed_key = ((unsigned long long)hm::fbits(u + hm_scan_delta(a.bf != 0, a.kterms, a.rmax2_bits)) << 32) | 0xffffffffull;
            seed_row = bi;
            if (lane == 0) { a.seed->key = seed_key; a.seed->i = bi; a.seed->valid = 1u; }
        } else if (a.seed->valid != 0u) {
            seed_key = a.seed->key;
            seed_row = a.seed->i;
        }
    }
    // arm the next search over the same row range: emitted = 0, running key = seed -- the merge loop then goes from
    // this kernel straight into the next scan
    if (a.arm && found != 2u && lane == 0) {
        const bool use = seed_row != 0xffffffffu && (int)seed_row >= a.arm_rb && (int)seed_row < a.arm_re;
        a.ctr64[0] = 0ull;
        a.ctr64[1] = use ? seed_key : ~0ull;
        a.ctr64[2] = 0ull;
        if (a.mf.rowkey != nullptr) *a.mf.rowkey = ~0ull;
    }
    // ---- fused merge of the pair just found (device-resident loop) ----
    if (a.mf.X != nullptr) {
        if (found == 1u) {
            const int32_t li = a.mf.len[bi], lj = a.mf.len[bj];
            const float w = (float)((double)lj / (double)(li + lj));     // len(tj) / (len(ti) + len(tj)), hyperbolic_merge.py:317-323
            hm_wave_stage_rows(a.img, a.RS, a.d, bi, bj, ms, lane);
            const float r2 = hm_wave_midpoint(a.d, w, a.mf.c, a.sign_mode, ms, true, lane);
            hm_wave_store_row(ms, r2, a.d, a.RS, a.KC, a.mf.X, a.mf.ld, a.img, a.img16, a.mf.new_row, a.rmax2_bits, lane);
            if (lane == 0) {
                a.mf.len_rw[a.mf.new_row] = li + lj;
                if (loop) loop->steps_done += 1u;
            }
        } else if (loop != nullptr && lane == 0) {
            loop->stop = found == 2u ? (over_small ? 6u : 2u) : 1u;       // (6: too many survivors for the pipelined loop's tail grid only)
        }
    }
    if (a.mf.rowkey != nullptr && loop != nullptr && lane == 0) loop->tails_done += 1u;      // (pipelined loop: the scans' order guard)
}

int hm_launch_argmin_tail(hm_engine* e, const ScanArgs& sa, float sqrt_c, float thr, ArgminRec* rec_out, bool with_seed,
                          int arm_rb, int arm_re, bool arm, const MergeFuse& mf, hipStream_t s)
{
    TailArgs t;
    memset(&t, 0, sizeof(t));
    t.ent = mf.pipe_ent ? mf.pipe_ent : e->ent; t.ctr64 = mf.pipe_ctr64 ? mf.pipe_ctr64 : e->d_ctr64; t.cap = e->ent_cap;
    t.img = e->img; t.img16 = e->img16; t.RS = e->RS; t.d = e->d; t.KC = e->KC; t.sign_mode = e->sign_mode;
    t.sqrt_c = sqrt_c; t.thr = thr;
    t.parts = e->d_parts; t.ticket = e->d_ctr + 6;
    t.out = rec_out; t.out2 = (rec_out == e->d_rec) ? e->d_rec + 1 : nullptr;
    t.seed = with_seed ? e->d_seed : nullptr;
    t.bf = sa.bf16; t.kterms = sa.bf16 ? 8 * e->KC : e->RS; t.rmax2_bits = e->d_rmax2;
    t.arm = arm ? 1 : 0; t.arm_rb = arm_rb; t.arm_re = arm_re;
    t.mf = mf;
    // Pipelined loop (mf.rowkey set): the kernel runs UNDER the next step's scan, whose two blocks per CU leave room for four
    // more waves at most -- 256-thread blocks; otherwise 1024
    // -- and four of them: every block of the grid has to find a slot before the kernel can end, whether it has entries or not
    hipLaunchKernelGGL(hm_argmin_tail_kernel, dim3(mf.rowkey != nullptr ? 4 : HM_TAIL_BLOCKS), dim3(mf.rowkey != nullptr ? 256 : HM_TAIL_THREADS), 0, s, t);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// counters + running-key seed of an argmin search that was not armed by its predecessor
__global__ void hm_seed_init_kernel(const ArgminSeed* __restrict__ seed, unsigned long long* __restrict__ ctr64, uint32_t* __restrict__ ctr,
                                    int row_begin, int row_end)
{
    if (threadIdx.x < 8) ctr[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        const bool use = seed->valid != 0u && (int)seed->i >= row_begin && (int)seed->i < row_end;
        ctr64[0] = 0ull;
        ctr64[1] = use ? seed->key : ~0ull;
        ctr64[2] = 0ull;
        ctr64[3] = 0ull;
    }
}

int hm_launch_seed_init(hm_engine* e, const ScanArgs& a, hipStream_t s)
{
    // (a.ctr64: the engine's first counter set, or the second one of the pipelined loop)
    hipLaunchKernelGGL(hm_seed_init_kernel, dim3(1), dim3(64), 0, s, e->d_seed, a.ctr64, e->d_ctr, a.row_begin, a.row_end);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// top-k: exact distances of the emitted entries
// ------------------------------------------------------------------------------------------------
// entries {bits(u_f'), i, j, sure} -> {dbits | 0xffffffff, i, j, bits(u_c')} with the canonical distance.
// counts[0] valid, counts[1] valid & !sure, counts[3] sure & !valid (margin violated: must be 0),
// counts[4] valid entries of the COMPLETE region -- the part of the pair space of which every pair was emitted:
//   without a row cut: u_c' <= 1 (distance 0) or bits(u_c') + hm_tie_slack(cut_bits) <= cut_bits (a pair with u_c' <= cut
//   has u_f' <= cut + delta and was emitted; the slack covers the few-ulp wiggle of acosh);
//   with a row cut (tie flood): the zero-distance pairs of rows i <= tie_imax.
__global__ __launch_bounds__(256) void hm_post_distance_kernel(uint4* __restrict__ ent, const unsigned long long* __restrict__ ctr64,
                                                               uint32_t cap, const float* __restrict__ img, int RS, int d, int sign_mode,
                                                               float sqrt_c, float thr, uint32_t cut_bits, int tie_imax,
                                                               uint32_t* __restrict__ counts)
{
    const unsigned long long emitted = ctr64[2];
    const uint32_t m = emitted > (unsigned long long)cap ? cap : (uint32_t)emitted;
    const int lane = threadIdx.x & 63;
    uint32_t nv = 0, nb = 0, bad = 0, nc = 0;
    const uint32_t hw = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const uint32_t stride = ((gridDim.x * blockDim.x) >> 5) * HM_GATHER;
    const int t32 = lane & 31;
    for (uint32_t base = (hw & ~1u) * HM_GATHER; base < m; base += stride) {
        const uint32_t mybase = base + (hw & 1u) * HM_GATHER;
        const uint32_t mine = mybase + t32;
        const uint4 en = ent[mine < m ? mine : m - 1];
        const float u = hm_halfwave_gather(lane, [&](int k) {
            const uint32_t ri = __shfl(en.y, (lane & 32) + k, 64), rj = __shfl(en.z, (lane & 32) + k, 64);
            return hm_img_u_halfwave(img, RS, d, ri, rj, sign_mode, lane);
        });
        const float uc = hm::clamp_min_one(u);
        const float dd = hm::acosh_c(uc) / sqrt_c;
        if (t32 < HM_GATHER && mine < m) {
            const bool valid = dd < thr;
            nv += valid ? 1u : 0u;
            nb += (valid && en.w == 0u) ? 1u : 0u;
            bad += (!valid && en.w != 0u) ? 1u : 0u;
            const uint32_t ub = hm::fbits(uc);
            const bool zero = ub <= 0x3f800000u;
            const bool complete = tie_imax != 0x7fffffff ? (zero && (int)en.y <= tie_imax)
                                                         : (zero || (cut_bits != 0xffffffffu ? (ub + hm_tie_slack(cut_bits) <= cut_bits) : true));
            nc += (valid && complete) ? 1u : 0u;
            ent[mine] = make_uint4(valid ? hm::fbits(dd) : 0xffffffffu, en.y, en.z, ub);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nv += __shfl_xor(nv, off, 64); nb += __shfl_xor(nb, off, 64); bad += __shfl_xor(bad, off, 64); nc += __shfl_xor(nc, off, 64);
    }
    if (lane == 0) {
        if (nv) atomicAdd(&counts[0], nv);
        if (nb) atomicAdd(&counts[1], nb);
        if (bad) atomicAdd(&counts[3], bad);
        if (nc) atomicAdd(&counts[4], nc);
    }
}

// radix narrowing: digit `level` (0..8) of the 96-bit key (12|12|8 bits per word)
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

// ---- exact order of m unique keys: sorted chunks in LDS, then rank = sum of lower bounds over the chunks ----
#define HM_SORT_CHUNK 2048
// bitonic sort of HM_SORT_CHUNK keys per block (padding = all ones), 1024 threads
__global__ __launch_bounds__(1024) void hm_chunk_sort_kernel(const uint4* __restrict__ ent, uint32_t m, uint4* __restrict__ out)
{
    __shared__ uint4 keys[HM_SORT_CHUNK];
    const uint32_t base = blockIdx.x * HM_SORT_CHUNK;
    for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 1024) {
        const uint32_t idx = base + q;
        keys[q] = idx < m ? ent[idx] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
    }
    __syncthreads();
    for (uint32_t k = 2; k <= HM_SORT_CHUNK; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t t = threadIdx.x;
            const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
            const uint32_t hi = lo | j;
            const bool up = (lo & k) == 0;
            const uint4 x = keys[lo], y = keys[hi];
            const bool swap = up ? hm_key_less(y.x, y.y, y.z, x.x, x.y, x.z) : hm_key_less(x.x, x.y, x.z, y.x, y.y, y.z);
            if (swap) { keys[lo] = y; keys[hi] = x; }
            __syncthreads();
        }
    }
    for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 1024) out[base + q] = keys[q];
}

// rank of every key = its index in its own chunk + the lower bounds in all other chunks; rank < k -> out[rank]
__global__ __launch_bounds__(256) void hm_rank_merge_kernel(const uint4* __restrict__ chunks, uint32_t m, uint32_t nchunks,
                                                            uint4* __restrict__ out, uint32_t k)
{
    __shared__ uint4 tile[HM_SORT_CHUNK];
    const uint32_t mine = blockIdx.x * 256 + threadIdx.x;              // position in the chunked array
    const uint32_t total = nchunks * HM_SORT_CHUNK;
    uint4 me = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
    if (mine < total) me = chunks[mine];
    const uint32_t my_chunk = mine / HM_SORT_CHUNK;
    uint32_t rank = mine % HM_SORT_CHUNK;                              // keys of the own chunk in front of this one
    for (uint32_t c = 0; c < nchunks; ++c) {
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 256) tile[q] = chunks[c * HM_SORT_CHUNK + q];
        __syncthreads();
        if (c == my_chunk) continue;
        uint32_t lo = 0, hi = HM_SORT_CHUNK;                           // first index whose key is not less than mine
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint4 o = tile[mid];
            if (hm_key_less(o.x, o.y, o.z, me.x, me.y, me.z)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (mine < total && me.y != 0xffffffffu && rank < k) out[rank] = me;
}

// ---- the same two kernels with the entry count taken from device memory (m = min(*m_dev, cap) + m_add): the
// incremental refresh enqueues its whole chain -- scan, distances, union with the previous list, sort, read-back -- and
// synchronises once; grids are sized for the most it accepts (HM_RANK_LIMIT entries), surplus blocks leave at once ----
__device__ __forceinline__ uint32_t hm_dev_count(const unsigned long long* m_dev, uint32_t cap, uint32_t m_add)
{
    const unsigned long long v = *m_dev;
    return (uint32_t)(v > (unsigned long long)cap ? cap : v) + m_add;
}

__global__ __launch_bounds__(256) void hm_append_prev_kernel(uint4* __restrict__ ent, const unsigned long long* __restrict__ m_dev, uint32_t cap,
                                                             const uint4* __restrict__ prev, uint32_t k)
{
    const unsigned long long m_new = *m_dev;
    if (m_new + (unsigned long long)k > (unsigned long long)cap) return;          // the host sees the count and falls back
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < k; t += gridDim.x * 256) ent[(uint32_t)m_new + t] = prev[t];
}

__global__ __launch_bounds__(1024) void hm_chunk_sort_dev_kernel(const uint4* __restrict__ ent, const unsigned long long* __restrict__ m_dev,
                                                                 uint32_t cap, uint32_t m_add, uint4* __restrict__ out)
{
    __shared__ uint4 keys[HM_SORT_CHUNK];
    const uint32_t m = hm_dev_count(m_dev, cap, m_add);
    const uint32_t base = blockIdx.x * HM_SORT_CHUNK;
    if (base >= m) return;                                                        // block-uniform
    for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 1024) {
        const uint32_t idx = base + q;
        keys[q] = idx < m ? ent[idx] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
    }
    __syncthreads();
    for (uint32_t k = 2; k <= HM_SORT_CHUNK; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t t = threadIdx.x;
            const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const uint32_t hi = lo | j;
            const bool up = (lo & k) == 0;
            const uint4 x = keys[lo], y = keys[hi];
            const bool swap = up ? hm_key_less(y.x, y.y, y.z, x.x, x.y, x.z) : hm_key_less(x.x, x.y, x.z, y.x, y.y, y.z);
            if (swap) { keys[lo] = y; keys[hi] = x; }
            __syncthreads();
        }
    }
    for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 1024) out[base + q] = keys[q];
}

__global__ __launch_bounds__(256) void hm_rank_merge_dev_kernel(const uint4* __restrict__ chunks, const unsigned long long* __restrict__ m_dev,
                                                                uint32_t cap, uint32_t m_add, uint4* __restrict__ out, uint32_t k)
{
    __shared__ uint4 tile[HM_SORT_CHUNK];
    const uint32_t m = hm_dev_count(m_dev, cap, m_add);
    const uint32_t nchunks = (m + HM_SORT_CHUNK - 1) / HM_SORT_CHUNK;
    const uint32_t total = nchunks * HM_SORT_CHUNK;
    if (blockIdx.x * 256 >= total) return;                                        // block-uniform
    const uint32_t mine = blockIdx.x * 256 + threadIdx.x;
    uint4 me = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0);
    if (mine < total) me = chunks[mine];
    const uint32_t my_chunk = mine / HM_SORT_CHUNK;
    uint32_t rank = mine % HM_SORT_CHUNK;
    for (uint32_t c = 0; c < nchunks; ++c) {
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < HM_SORT_CHUNK; q += 256) tile[q] = chunks[c * HM_SORT_CHUNK + q];
        __syncthreads();
        if (c == my_chunk) continue;
        uint32_t lo = 0, hi = HM_SORT_CHUNK;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint4 o = tile[mid];
            if (hm_key_less(o.x, o.y, o.z, me.x, me.y, me.z)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (mine < total && me.y != 0xffffffffu && rank < k) out[rank] = me;
}

// exact selection of the k smallest keys among m entries of `src` (keys unique; invalid = 0xffffffff)
// result in e->sorted.  `other` is scratch of the same capacity.
int hm_select_sorted(hm_engine* e, uint4* src, uint4* other, uint32_t m, uint32_t k, hipStream_t s)
{
    if (k == 0 || m == 0) return HM_OK;
    if (k > e->sorted_cap) return hm_fail(e, HM_E_CAPACITY, "top-k: k exceeds the engine's sorted capacity (65536)");
    uint4* cur = src;
    uint4* spare = other;
    uint32_t mcur = m;
    // up to HM_RANK_LIMIT entries are sorted outright: two more kernels cost less than the two host round trips of a
    // narrowing level (measured on the refresh of the fast tokenizer: ~100 us against ~30 us)
    const uint32_t rank_limit = HM_RANK_LIMIT;
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
        spare = src;
        mcur = e->h->ctr[3];
        if (mcur > 4u * HM_RANK_LIMIT) return hm_fail(e, HM_E_CAPACITY, "top-k: radix narrowing did not converge");
    }
    // (the scratch holds at least 65536 entries and mcur <= 4 * HM_RANK_LIMIT only when the buffers are 2^24 entries)
    const uint32_t nchunks = (mcur + HM_SORT_CHUNK - 1) / HM_SORT_CHUNK;
    if ((uint64_t)nchunks * HM_SORT_CHUNK > e->ent_cap) return hm_fail(e, HM_E_CAPACITY, "top-k: sort scratch too small");
    hipLaunchKernelGGL(hm_chunk_sort_kernel, dim3(nchunks), dim3(1024), 0, s, cur, mcur, spare);
    HM_HIP(hipGetLastError());
    hipLaunchKernelGGL(hm_rank_merge_kernel, dim3(nchunks * (HM_SORT_CHUNK / 256)), dim3(256), 0, s, spare, mcur, nchunks, e->sorted, k);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

// Choose an emission cut from sampled histograms of bits(u') so that roughly `target` entries
// (and certainly not more than the buffer holds) are emitted.  Pure performance heuristic: the
// caller verifies the outcome (complete-region count) and widens the cut when it was too tight.
static int hm_estimate_cut(hm_engine* e, ScanArgs a, dim3 grid, int64_t target, uint32_t* cut_bits, int* tie_imax,
                           hipStream_t s)
{
    *cut_bits = 0xffffffffu;
    *tie_imax = 0x7fffffff;
    const int64_t pairs = hm_pairs_in_range(a.n, a.row_begin, a.row_end);
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
                             bool want_count, int64_t n_limit, int64_t* n_valid_emitted, int64_t* count, uint4** result_dev,
                             hipStream_t s);

// The bf16 prefilter's margin (delta ~ 0.004 * max||x_s||^2 in u) makes a shell of undecided pairs around the
// threshold; every one of them has to be emitted to be decided exactly.  With a threshold inside the bulk of the
// distance distribution that shell alone can exceed the emission buffer: the search then runs again with the fp32
// prefilter, whose shell is ~100x thinner.
int hm_topk_core(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all, bool want_count,
                 int64_t n_limit, int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s)
{
    const bool had_bf16 = hm_use_bf16(e);
    int rc = HM_E_CAPACITY;
    const bool exact_known = !list_all && (e->force_exact || (e->topk_exact_thr > 0.0f && thr >= e->topk_exact_thr));      // (remembered per table, like the fp32 fallback)
    if (!exact_known && !(had_bf16 && e->topk_f32_thr > 0.0f && thr >= e->topk_f32_thr))
        rc = hm_topk_core_form(e, c, thr, k, row_begin, row_end, list_all, want_count, n_limit, n_valid_emitted, count, result_dev, s);
    if (!exact_known && rc == HM_E_CAPACITY && had_bf16 && !e->force_f32) {
        e->force_f32 = true;
        rc = hm_topk_core_form(e, c, thr, k, row_begin, row_end, list_all, want_count, n_limit, n_valid_emitted, count, result_dev, s);
        e->force_f32 = false;
        if (rc == HM_OK && !(e->topk_f32_thr > 0.0f && e->topk_f32_thr <= thr)) e->topk_f32_thr = thr;
    }
    if (rc == HM_E_CAPACITY && !list_all) {
        // No emission cut fits the buffer in either prefilter form (a table whose distances sit within a few hundred ulps of
        // u = 1): every pair evaluated exactly, selection by counting (hm_exact.hip).  Slow, and always an answer.
        rc = hm_topk_exact(e, c, thr, k, row_begin, row_end, n_limit, n_valid_emitted, count, result_dev, s);
        if (rc == HM_OK && !(e->topk_exact_thr > 0.0f && e->topk_exact_thr <= thr)) e->topk_exact_thr = thr;
    }
    return rc;
}

// *count: the exact number of candidates, or -1 when want_count is false and at least k exist (not counted)
static int hm_topk_core_form(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all,
                             bool want_count, int64_t n_limit, int64_t* n_valid_emitted, int64_t* count, uint4** result_dev,
                             hipStream_t s)
{
    *n_valid_emitted = 0;
    *count = 0;
    *result_dev = nullptr;
    hm_flush_pending_timing(e);
    e->last_scan_ms = 0.f; e->last_pairs = 0; e->last_emitted = 0; e->last_passes = 0;
    const Bounds b = hm_bounds(thr, c);
    ScanArgs a; dim3 grid;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid, n_limit)) return HM_OK;
    const float sqrt_c = sqrtf(c);
    const bool whole = (a.row_begin == 0 && a.row_end == e->n - 1);
    a.count_sure = (want_count || list_all) ? 1 : 0;

    uint32_t cut_bits = 0xffffffffu;
    int tie_imax = 0x7fffffff;
    if (!list_all && k == 0) {
        cut_bits = 0x3f800000u;          // a pure count: nothing has to be emitted but the undecided shell around the threshold
        tie_imax = -1;
    } else if (!list_all) {
        if (whole && e->have_cut && e->last_cut_k >= k && e->last_cut_c == c && e->last_cut_bits > 0x3f800000u) {
            cut_bits = e->debug_cut ? e->last_cut_bits : e->last_cut_bits + hm_tie_slack(e->last_cut_bits);
            e->debug_cut = false;
        } else {
            int rc = hm_estimate_cut(e, a, grid, 4 * k + 4096, &cut_bits, &tie_imax, s);
            if (rc) return rc;
        }
    }
    // bracket of the emission cut (bit patterns of u'): the largest cut found too tight (fewer than `want` entries in its
    // complete region) and the smallest found too generous (emission buffer overflow).  In high dimensions the distances
    // are so concentrated that P(u - 1 < x) grows like x^(d/2): widening a tight cut geometrically overshoots by orders of
    // magnitude and a fresh estimate undershoots again -- once both sides are known the cut is bisected between them.
    uint32_t tight_bits = 0u, loose_bits = 0xffffffffu;
    int stuck = 0;
    for (int attempt = 0; attempt < 40; ++attempt) {
        a.cut_bits = cut_bits;
        a.tie_imax = tie_imax;
        HM_HIP(hipMemsetAsync(e->d_ctr, 0, sizeof(uint32_t) * 8, s));
        HM_HIP(hipMemsetAsync(e->d_ctr64, 0, sizeof(unsigned long long) * 4, s));
        HM_HIP(hm_launch_scan(e, HM_MODE_TOPK, a, grid, s, e->ev0, e->ev1));
        hipLaunchKernelGGL(hm_post_distance_kernel, dim3(1024), dim3(256), 0, s, e->ent, e->d_ctr64, e->ent_cap, e->img, e->RS, e->d,
                           e->sign_mode, sqrt_c, thr, cut_bits, tie_imax, e->d_ctr + 1);
        HM_HIP(hipGetLastError());
        HM_HIP(hipMemcpyAsync(e->h->ctr, e->d_ctr, sizeof(uint32_t) * 8, hipMemcpyDeviceToHost, s));
        HM_HIP(hipMemcpyAsync(e->h->ctr64, e->d_ctr64, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
        e->last_scan_ms += ms;
        e->last_passes += 1;
        e->last_pairs = hm_pairs_in_range(a.n, a.row_begin, a.row_end);
        e->tot_scan_ms += ms; e->tot_pairs += e->last_pairs; e->tot_launches += 1;
        e->last_emitted = (int64_t)e->h->ctr64[2];
        if (e->h->ctr[4] != 0)
            return hm_fail(e, HM_E_STATE, "pair scan: prefilter margin violated (an entry classified as surely below the "
                                          "threshold is not); table holds non-finite rows other than all-NaN rows?");
        const uint64_t emitted = e->h->ctr64[2];
        const bool overflow = emitted > e->ent_cap;
        const int64_t valid = e->h->ctr[1];
        const int64_t complete = e->h->ctr[5];
        const bool emitted_all = (cut_bits == 0xffffffffu && tie_imax == 0x7fffffff);
        if (overflow) {
            if (list_all) return hm_fail(e, HM_E_CAPACITY, "candidate listing: more candidates than the emission buffer holds");
            if (tie_imax == 0x7fffffff && tight_bits != 0u) {
                if (cut_bits < loose_bits) loose_bits = cut_bits;
                // (a bracket that cannot be split: the shell of undecided pairs around the cut alone overflows the buffer --
                // the caller retries with the fp32 prefilter, whose shell is ~100x thinner)
                if (loose_bits <= tight_bits + 1u) return hm_fail(e, HM_E_CAPACITY, "top-k: could not bound the emission");
                cut_bits = tight_bits + (loose_bits - tight_bits) / 2u;
                continue;
            }
            if (tie_imax == 0x7fffffff && cut_bits < loose_bits) loose_bits = cut_bits;
            // estimate was too generous (or none was made): estimate with a smaller target
            const int64_t target = std::max<int64_t>((2 * k + 1024) >> std::min(attempt, 20), k + 64);
            int rc = hm_estimate_cut(e, a, grid, target, &cut_bits, &tie_imax, s);
            if (rc) return rc;
            if (cut_bits == 0xffffffffu) return hm_fail(e, HM_E_CAPACITY, "top-k: could not bound the emission");
            // the smallest target already, and its cut is one that is known to overflow: estimating again would give the same
            // cut (a table too dense for any cut of this prefilter form -- the caller moves on to the next form / the exact path)
            if (tie_imax == 0x7fffffff && target == k + 64 && cut_bits >= loose_bits && ++stuck >= 2)
                return hm_fail(e, HM_E_CAPACITY, "top-k: could not bound the emission");
            continue;
        }
        // exact total (when counted): sure + valid borderline; everything emitted: the valid ones
        const int64_t total = emitted_all ? valid : (a.count_sure ? (int64_t)e->h->ctr64[0] + (int64_t)e->h->ctr[2] : -1);
        const int64_t want = total >= 0 ? std::min<int64_t>(k, total) : k;
        if (!emitted_all && complete < want) {
            // fewer than `want` entries lie in the region that is known to be complete: the cut was too tight.
            // Widen geometrically in the ulp domain (rows for a tie flood) and retry.
            if (tie_imax != 0x7fffffff) {
                tie_imax = tie_imax >= a.row_end ? 0x7fffffff : (int)std::min<int64_t>((int64_t)tie_imax * 4 + 256, a.row_end);
                if (tie_imax >= a.row_end) tie_imax = 0x7fffffff;
            } else {
                if (cut_bits > tight_bits) tight_bits = cut_bits;
                const uint32_t span = cut_bits - 0x3f800000u;
                const uint64_t nb = (uint64_t)cut_bits + std::max<uint32_t>(span, 4u * hm_tie_slack(cut_bits));
                if (loose_bits != 0xffffffffu && nb >= (uint64_t)loose_bits) {
                    if (loose_bits <= tight_bits + 1u) return hm_fail(e, HM_E_CAPACITY, "top-k: could not bound the emission");
                    cut_bits = tight_bits + (loose_bits - tight_bits) / 2u;
                } else {
                    cut_bits = nb >= 0x7f800000ull ? 0xffffffffu : (uint32_t)nb;
                }
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

// ------------------------------------------------------------------------------------------------
// C ABI: searches
// ------------------------------------------------------------------------------------------------
static const MergeFuse kNoMerge = {nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};

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
    const bool skip_init = e->armed && e->armed_rb == req_rb && e->armed_re == req_re;
    e->armed = false;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid)) {
        HM_HIP(hipMemsetAsync(rec_dev, 0, sizeof(ArgminRec), s));        // found = 0
        return HM_OK;
    }
    if (!skip_init) {
        int rc0 = hm_launch_seed_init(e, a, s);
        if (rc0) return rc0;
    }
    MergeFuse mf = kNoMerge;
    if (e->shard_loop) {                          // inside hm_shard_loop_begin .. _end: a stopped loop skips its searches
        a.stop = &e->d_loop->stop;
        mf.loop = e->d_loop;
    }
    // (inside a sharded loop only the batch's first search carries the engine's own timing events: reading them back at the
    // next call is a host wait, and the point of the loop is that the host runs ahead of the device; the in-library loop in
    // its measurement mode hands every step its own pair -- e->step_ev0 / _ev1, read after the batch)
    const bool step_timed = e->step_ev0 != nullptr;
    const bool timed = !step_timed && (!e->shard_loop || e->n == e->shard_n0);
    HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, step_timed ? e->step_ev0 : (timed ? e->ev0 : nullptr),
                          step_timed ? e->step_ev1 : (timed ? e->ev1 : nullptr)));
    int rc = hm_launch_argmin_tail(e, a, sqrtf(c), thr, reinterpret_cast<ArgminRec*>(rec_dev), true, (int)req_rb,
                                   req_re < 0 ? 0x7fffffff : (int)std::min<int64_t>(req_re, 0x7fffffff), true, mf, s);
    if (rc) return rc;
    // Armed optimistically: the host does not see this record.  Should the search have overflowed (found = 2, the
    // kernel then arms nothing), the next search of this range starts on the stale counters, reports found = 2 as
    // well, and its caller takes the bounded host path -- slower, never wrong.
    e->armed = true; e->armed_rb = req_rb; e->armed_re = req_re;
    if (timed) {
        e->pending_timing = true;
        e->pending_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
    }
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
    const bool skip_init = e->armed && e->armed_rb == req_rb && e->armed_re == req_re;
    e->armed = false;
    if (b.none || e->n < 2 || !hm_prepare_scan(e, b, row_begin, row_end, a, grid)) return HM_OK;
    if (e->force_exact || (e->topk_exact_thr > 0.0f && thr >= e->topk_exact_thr)) {
        // this table's searches are known to end in the prefilter-free path (hm_exact.hip): no scan attempts first
        float d1 = 0.f; int32_t i1 = -1, j1 = -1; int64_t n1 = 0, cnt1 = 0;
        const int rc2 = hm_pairwise_topk_nocount(e, c, thr, 1, row_begin, row_end, &d1, &i1, &j1, &n1, &cnt1, stream);
        if (rc2) return rc2;
        if (n1 > 0) { *found = 1; *d = d1; *i = i1; *j = j1; }
        return HM_OK;
    }
    const float sqrt_c = sqrtf(c);
    const int arm_re = req_re < 0 ? 0x7fffffff : (int)std::min<int64_t>(req_re, 0x7fffffff);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 1 (after an overflow) keeps the final running key of pass 0: every wave then starts
        // with the tight bound and only the band around the minimum is emitted
        if (pass == 0) {
            if (!skip_init) {           // else: the previous search of this range left counters and key armed
                int rc0 = hm_launch_seed_init(e, a, s);
                if (rc0) return rc0;
            }
        } else {
            HM_HIP(hipMemsetAsync(e->d_ctr64 + 2, 0, sizeof(unsigned long long), s));     // emitted = 0, running key kept
        }
        HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, e->ev0, e->ev1));
        int rc = hm_launch_argmin_tail(e, a, sqrt_c, thr, e->d_rec, true, (int)req_rb, arm_re, true, kNoMerge, s);
        if (rc) return rc;
        // record + emitted count (the slot behind the record) in one copy
        HM_HIP(hipMemcpyAsync(e->h->rec2, e->d_rec, 2 * sizeof(ArgminRec), hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        e->h->rec = e->h->rec2[0];
        const uint64_t emitted = (uint64_t)e->h->rec2[1].found | ((uint64_t)e->h->rec2[1].dbits << 32);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
        e->last_scan_ms += ms;
        e->last_passes += 1;
        e->last_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
        e->tot_scan_ms += ms; e->tot_pairs += e->last_pairs; e->tot_launches += 1;
        e->last_emitted = (int64_t)emitted;
        if (emitted <= e->ent_cap) break;
        // overflow: the running key is the exact minimum over all published waves; rerun bounded by it
        if (pass == 1) {
            if (a.bf16 && !e->force_f32) {      // the bf16 margin's shell around the bound is too populated: fp32 prefilter
                e->force_f32 = true;
                const int rc2 = hm_pairwise_argmin(e, c, thr, row_begin, row_end, d, i, j, found, stream);
                e->force_f32 = false;
                return rc2;
            }
            // Still too many pairs inside the running key's slack band (a very dense table: the band is 1024 ulps of
            // u, which near u = 1 spans every distance below ~0.016): take the first entry of an exact top-1 search,
            // whose emission cut is found by histogram zooming instead.
            float d1 = 0.f; int32_t i1 = -1, j1 = -1; int64_t n1 = 0, cnt1 = 0;
            const int rc2 = hm_pairwise_topk(e, c, thr, 1, row_begin, row_end, &d1, &i1, &j1, &n1, &cnt1, stream);
            if (rc2) return rc2;
            if (n1 > 0) { *found = 1; *d = d1; *i = i1; *j = j1; }
            return HM_OK;
        }
    }
    if (e->h->rec.found == 1u) {
        union { uint32_t u; float f; } cv; cv.u = e->h->rec.dbits;
        *found = 1; *d = cv.f; *i = (int32_t)e->h->rec.i; *j = (int32_t)e->h->rec.j;
    }
    if (e->h->rec.found != 2u) { e->armed = true; e->armed_rb = req_rb; e->armed_re = req_re; }
    return HM_OK;
}

// Incremental refresh (rows are only ever appended).  The ordered list S[:k] of the previous
// whole-table search is still the k smallest of the pairs among the rows it saw; the k smallest of the grown table
// are therefore the k smallest of S[:k] + the pairs that involve a NEW row (j >= prev_n).  Only the last column
// tile(s) are scanned (cut = largest u' of S[:k], so every new pair that could enter is emitted) and the two lists
// are merged by the exact selection.  Preconditions checked by the caller; the exact total is not produced here.
// when the incremental refresh applies (checked by its callers)
static bool hm_topk_incremental_ok(const hm_engine* e, float c, float thr, int64_t k)
{
    return k > 0 && e->incremental_topk && e->have_cut && e->prev_valid && e->prev_k == k && e->last_cut_c == c && thr >= e->prev_thr &&
           e->n >= e->prev_n && e->n - e->prev_n <= 8192 && e->last_cut_bits > 0x3f800000u && !e->debug_cut;
}

// entries the device-side sort takes: the pairs of a few hundred new rows under the previous list's cut are often several
// ten thousand; 64 chunks keep the rank merge around 0.1 ms, beyond that the full search's radix narrowing is used
static uint32_t hm_topk_incremental_limit(const hm_engine* e) { return std::min<uint32_t>(e->ent_cap, 64u * HM_SORT_CHUNK); }

// everything of the incremental refresh up to the read-back, enqueued; no host wait
static int hm_topk_incremental_enqueue(hm_engine* e, float c, float thr, int64_t k, hipStream_t s)
{
    hm_flush_pending_timing(e);
    e->last_scan_ms = 0.f; e->last_pairs = 0; e->last_emitted = 0; e->last_passes = 0;
    const Bounds b = hm_bounds(thr, c);
    const uint32_t lim = hm_topk_incremental_limit(e);
    if ((uint64_t)k + 1 > lim) return HM_E_CAPACITY;
    HM_HIP(hipMemsetAsync(e->d_ctr, 0, sizeof(uint32_t) * 8, s));
    HM_HIP(hipMemsetAsync(e->d_ctr64, 0, sizeof(unsigned long long) * 4, s));
    if (e->n > e->prev_n) {
        ScanArgs a; dim3 grid;
        if (!hm_prepare_scan(e, b, 0, -1, a, grid, -1, e->prev_n)) return hm_fail(e, HM_E_STATE, "incremental refresh: empty scan");
        a.count_sure = 0;
        a.cut_bits = e->last_cut_bits + hm_tie_slack(e->last_cut_bits);
        a.tie_imax = 0x7fffffff;
        HM_HIP(hm_launch_scan(e, HM_MODE_TOPK, a, grid, s));
        hipLaunchKernelGGL(hm_post_distance_kernel, dim3(256), dim3(256), 0, s, e->ent, e->d_ctr64, e->ent_cap, e->img, e->RS, e->d,
                           e->sign_mode, sqrtf(c), thr, a.cut_bits, a.tie_imax, e->d_ctr + 1);
        HM_HIP(hipGetLastError());
        e->last_passes = 1;
    }
    // union with the previous list (appended behind the new entries; invalid new entries carry all-ones keys), sorted
    // outright: the count of new entries stays on the device
    const unsigned long long* m_dev = e->d_ctr64 + 2;
    const uint32_t room = lim - (uint32_t)k;                                       // new entries the sort can take
    hipLaunchKernelGGL(hm_append_prev_kernel, dim3(64), dim3(256), 0, s, e->ent, m_dev, lim, e->d_prev, (uint32_t)k);
    const uint32_t max_chunks = (lim + HM_SORT_CHUNK - 1) / HM_SORT_CHUNK;
    hipLaunchKernelGGL(hm_chunk_sort_dev_kernel, dim3(max_chunks), dim3(1024), 0, s, e->ent, m_dev, room, (uint32_t)k, e->ent2);
    hipLaunchKernelGGL(hm_rank_merge_dev_kernel, dim3(max_chunks * (HM_SORT_CHUNK / 256)), dim3(256), 0, s, e->ent2, m_dev, room, (uint32_t)k,
                       e->sorted, (uint32_t)k);
    HM_HIP(hipGetLastError());
    HM_HIP(hipMemcpyAsync(e->h->ctr64, e->d_ctr64, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, s));
    HM_HIP(hipMemcpyAsync(e->h_sorted, e->sorted, sizeof(uint4) * (size_t)k, hipMemcpyDeviceToHost, s));
    return HM_OK;
}

// the one synchronisation; HM_E_CAPACITY: too many new entries for the device-side sort (the caller runs the full search)
static int hm_topk_incremental_finish(hm_engine* e, int64_t k, hipStream_t s, uint32_t* kk_out)
{
    *kk_out = 0;
    HM_HIP(hipStreamSynchronize(s));
    e->last_emitted = (int64_t)e->h->ctr64[2];
    if (e->h->ctr64[2] > (unsigned long long)(hm_topk_incremental_limit(e) - (uint32_t)k)) return HM_E_CAPACITY;
    *kk_out = (uint32_t)k;
    return HM_OK;
}

// While rows are only appended, the k smallest keys of the grown table are the k smallest of (the previous
// list) + (the pairs that involve a NEW row, j >= prev_n): only the last column tiles are scanned (cut = largest u' of
// the previous list, so every new pair that could enter is emitted) and the union is sorted exactly.
static int hm_topk_incremental(hm_engine* e, float c, float thr, int64_t k, hipStream_t s, uint32_t* kk_out)
{
    *kk_out = 0;
    const int rc = hm_topk_incremental_enqueue(e, c, thr, k, s);
    if (rc) return rc;
    return hm_topk_incremental_finish(e, k, s, kk_out);
}

// host side of a finished top-k search whose ordered list sits in e->h_sorted: outputs and the state the next refresh uses
static void hm_topk_publish(hm_engine* e, uint32_t kk, int64_t k, float c, float thr, bool keep, float* d_out, int32_t* i_out, int32_t* j_out,
                            int64_t* n_out)
{
    uint32_t mx = 0;
    for (uint32_t t = 0; t < kk; ++t) {
        union { uint32_t u; float f; } cv; cv.u = e->h_sorted[t].x;
        d_out[t] = cv.f; i_out[t] = (int32_t)e->h_sorted[t].y; j_out[t] = (int32_t)e->h_sorted[t].z;
        mx = std::max(mx, e->h_sorted[t].w);     // entries are ordered by distance, not by u': the max u' bits of the selection
    }
    *n_out = kk;
    // remember the largest u' of the selection: while rows are only appended, the k-th smallest key can
    // only move down, so this cut (+ tie slack) is a guaranteed superset for the next refresh
    if (keep) {
        e->have_cut = true;
        e->last_cut_bits = mx;
        e->last_cut_k = k;
        e->last_cut_c = c;
        e->prev_valid = true;
        e->prev_k = k;
        e->prev_n = e->n;
        e->prev_thr = thr;
    } else {
        e->have_cut = false;
        e->prev_valid = false;
    }
}

// common body of hm_pairwise_topk / hm_pairwise_topk_nocount
static int hm_topk_impl(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool want_count,
                        float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count, void* stream)
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
    const bool whole = row_begin <= 0 && (row_end < 0 || row_end >= e->n - 1);
    uint32_t kk = 0;
    bool done = false;
    if (!want_count && whole && hm_topk_incremental_ok(e, c, thr, k)) {
        const int rc = hm_topk_incremental(e, c, thr, k, s, &kk);
        if (rc == HM_OK) { done = true; *count = -1; }
        else if (rc != HM_E_CAPACITY) return rc;
    }
    if (!done) {
        int64_t valid = 0, total = 0;
        uint4* res = nullptr;
        int rc = hm_topk_core(e, c, thr, k, row_begin, row_end, false, want_count || k == 0, -1, &valid, &total, &res, s);
        if (rc) return rc;
        *count = total;
        kk = (uint32_t)std::min<int64_t>(k, valid);
        if (kk == 0 || !res) { e->prev_valid = false; return HM_OK; }
        const uint32_t m = (uint32_t)std::min<uint64_t>(e->h->ctr64[2], e->ent_cap);
        rc = hm_select_sorted(e, res, e->ent2, m, kk, s);
        if (rc) return rc;
    }
    const bool keep = (kk == k && whole);
    if (keep) HM_HIP(hipMemcpyAsync(e->d_prev, e->sorted, sizeof(uint4) * kk, hipMemcpyDeviceToDevice, s));    // (stream-ordered: no wait needed for it)
    if (!done) {                                   // (the incremental refresh has read its list back already)
        HM_HIP(hipMemcpyAsync(e->h_sorted, e->sorted, sizeof(uint4) * kk, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
    }
    hm_topk_publish(e, kk, k, c, thr, keep, d_out, i_out, j_out, n_out);
    return HM_OK;
}

extern "C" int hm_pairwise_topk(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, float* d_out,
                                int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count, void* stream)
{
    return hm_topk_impl(e, c, thr, k, row_begin, row_end, true, d_out, i_out, j_out, n_out, count, stream);
}

extern "C" int hm_pairwise_topk_nocount(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end,
                                        float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count, void* stream)
{
    return hm_topk_impl(e, c, thr, k, row_begin, row_end, false, d_out, i_out, j_out, n_out, count, stream);
}

// The refresh of a table whose rows were only appended since the last whole-table top-k search, in two halves: begin
// enqueues the whole chain and returns; end waits and delivers.  Between the two the engine must not be used.
extern "C" int hm_topk_refresh_begin(hm_engine* e, float c, float thr, int64_t k, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_topk_refresh_begin: engine is NULL");
    if (e->refresh_pending) return hm_fail(e, HM_E_STATE, "hm_topk_refresh_begin: a refresh is already pending");
    if (!(c > 0.0f) || k <= 0 || k > (int64_t)e->sorted_cap) return hm_fail(e, HM_E_ARG, "hm_topk_refresh_begin: bad arguments");
    if (!hm_topk_incremental_ok(e, c, thr, k)) return HM_E_NA;                     // (no message: "not applicable" is an ordinary answer)
    HM_HIP(hipSetDevice(e->device));
    e->armed = false;
    const int rc = hm_topk_incremental_enqueue(e, c, thr, k, (hipStream_t)stream);
    if (rc) return rc;
    e->refresh_pending = true;
    e->refresh_k = k; e->refresh_c = c; e->refresh_thr = thr; e->refresh_stream = stream;
    return HM_OK;
}

extern "C" int hm_topk_refresh_end(hm_engine* e, float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_topk_refresh_end: engine is NULL");
    if (!e->refresh_pending) return hm_fail(e, HM_E_STATE, "hm_topk_refresh_end: no refresh pending");
    if (!d_out || !i_out || !j_out || !n_out) return hm_fail(e, HM_E_ARG, "hm_topk_refresh_end: NULL output pointer");
    HM_HIP(hipSetDevice(e->device));
    e->refresh_pending = false;
    hipStream_t s = (hipStream_t)e->refresh_stream;
    *n_out = 0;
    uint32_t kk = 0;
    const int rc = hm_topk_incremental_finish(e, e->refresh_k, s, &kk);
    if (rc) return rc;                                                             // HM_E_CAPACITY: run hm_pairwise_topk_nocount instead
    HM_HIP(hipMemcpyAsync(e->d_prev, e->sorted, sizeof(uint4) * kk, hipMemcpyDeviceToDevice, s));
    hm_topk_publish(e, kk, e->refresh_k, e->refresh_c, e->refresh_thr, true, d_out, i_out, j_out, n_out);
    return HM_OK;
}

extern "C" int hm_pairwise_count(hm_engine* e, float c, float thr, int64_t n_limit, int64_t* count, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_pairwise_count: engine is NULL");
    if (!count || !(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_pairwise_count: bad arguments");
    e->armed = false;
    HM_HIP(hipSetDevice(e->device));
    *count = 0;
    int64_t valid = 0, total = 0;
    uint4* res = nullptr;
    const bool had_cut = e->have_cut;             // a count does not disturb the cut prediction of the refreshes
    const uint32_t cb = e->last_cut_bits; const int64_t ck = e->last_cut_k; const float cc = e->last_cut_c;
    e->have_cut = false;
    int rc = hm_topk_core(e, c, thr, 0, 0, -1, false, true, n_limit, &valid, &total, &res, (hipStream_t)stream);
    e->have_cut = had_cut; e->last_cut_bits = cb; e->last_cut_k = ck; e->last_cut_c = cc;
    if (rc) return rc;
    *count = total;
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
    int rc = hm_topk_core(e, c, thr, 0, row_begin, row_end, true, true, -1, &valid, &cnt, &res, s);
    if (rc) return rc;
    *total = cnt;
    if (!res || valid == 0 || cap == 0) return HM_OK;
    const uint32_t m = (uint32_t)std::min<uint64_t>(e->h->ctr64[2], e->ent_cap);
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
    // row tiles (hm_newrow_key_kernel, the pipelined loop's row pass): one 64-bit key folded by atomicMin
    unsigned long long* key = e->d_rowkey + 2;
    HM_HIP(hipMemsetAsync(key, 0xff, sizeof(unsigned long long), s));
    int rc = hm_launch_row_key(e, row, n_partners, sqrtf(c), thr, key, s);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(&e->h->ctr64[3], key, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    const unsigned long long k64 = e->h->ctr64[3];
    if (k64 != ~0ull) {
        union { uint32_t u; float f; } cv; cv.u = (uint32_t)(k64 >> 32);
        const int64_t p = (int64_t)(uint32_t)k64;
        *found = 1; *d = cv.f; *i = (int32_t)(p < row ? p : row); *j = (int32_t)(p < row ? row : p);
    }
    return HM_OK;
}
