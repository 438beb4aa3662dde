// hm_exact.hip -- the prefilter-free search: last resort of hm_topk_core (hm_search.hip).
//
// The two-stage search (MFMA prefilter with a margin delta, exact re-evaluation of the survivors) needs the survivors of
// SOME emission cut to fit the emission buffer.  A table whose pairwise u = -<x, y>_L all lie within a few hundred ulps of 1
// (embeddings of scale 1e-3: millions of pairs per ulp of u, delta alone spans tens of ulps) has no such cut: every cut
// that completes k entries also emits millions.  The reference has no such limit -- it evaluates every pair
// (tokenizer/hyperbolic_merge.py:253-324, fast_hyperbolic_merge.py:153-232) -- so neither may the engine: this file
// evaluates EVERY pair in the canonical arithmetic (row tiles, hm_rows_device.h: no prefilter, no margin) and selects the
// k smallest (distance bits, i, j) by counting:
//   1. three histogram passes over the distance bits (11 + 11 + 10) find b* = the k-th smallest distance's bits, the number
//      of pairs below b* and at b*, and the exact candidate count;
//   2. if the pairs <= b* do not fit the emission buffer (a tie flood at b*), a per-row count of the pairs AT b* finds the row
//      i* up to which they are needed (order within equal bits is (i, j), row-major);
//   3. one emission pass writes the pairs below b* and the pairs at b* of rows <= i*; the caller's exact selection sorts them.
// Four or five passes of ~N^2/2 canonical distances each (a few ms at 25 k rows, ~0.1 s at 131 072): slow and always right.
#include <algorithm>
#include <vector>

#include "hm_common.h"
#include "hm_rows_device.h"

#pragma clang fp contract(off)

#define HM_EXACT_WAVES 3            // partner tiles per block; LDS = (1 + HM_EXACT_WAVES) row tiles + the histogram
#define HM_EXACT_BINS 2048

struct ExactArgs {
    const float* img;
    int RS, d, sign_mode;
    float sqrt_c, thr;
    int n;                          // partner rows j < n
    int row_begin, row_end;         // rows i in [row_begin, row_end)
    int ti0, ntj;                   // first row tile of the launch; partner tiles of the table
    int pass;                       // 0 histogram, 1 per-row count at bstar, 2 emission
    uint32_t prefix, prefix_mask;   // pass 0: only distances with (bits & prefix_mask) == prefix
    int shift, nbits;               //         digit = (bits >> shift) & ((1 << nbits) - 1)
    uint32_t bstar;
    int istar;                      // pass 2: bits < bstar, or bits == bstar and i <= istar
    uint32_t* hist;                 // [HM_EXACT_BINS]
    unsigned long long* total;      // pass 0, may be NULL: pairs with d < thr
    uint32_t* rowcnt;               // pass 1: [n]
    uint4* ent;
    unsigned long long* emitted;
    uint32_t cap;
};

__global__ __launch_bounds__(64 * HM_EXACT_WAVES) void hm_exact_scan_kernel(const ExactArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile_floats = HM_TILE_ROWS * a.RS;
    float* fixed = lds;                                          // 64 rows i, image layout
    float* tile = lds + (1 + wv) * tile_floats;                  // this wave's 64 partner rows j
    uint32_t* lhist = reinterpret_cast<uint32_t*>(lds + (1 + HM_EXACT_WAVES) * tile_floats);
    const int ti = a.ti0 + (int)blockIdx.y;
    const int tj = ti + (int)blockIdx.x * HM_EXACT_WAVES + wv;   // partner tiles start at the diagonal tile
    const bool active = tj < a.ntj;                              // wave-uniform
    TileRegs tr;
    if (active) hm_tile_load(a.img, a.RS, (int64_t)tj * HM_TILE_ROWS, a.n, tr, lane);
    {   // the fixed rows: one cooperative copy (rows past the table read as zeros and are never used)
        const int64_t r0 = (int64_t)ti * HM_TILE_ROWS;
        const int rows = (int)std::min<int64_t>(HM_TILE_ROWS, (int64_t)a.n - r0);
        const int nvec = rows > 0 ? rows * (a.RS >> 2) : 0;
        const uint4* src = reinterpret_cast<const uint4*>(a.img + r0 * a.RS);
        uint4* dst = reinterpret_cast<uint4*>(fixed);
        for (int q = threadIdx.x; q < HM_TILE_ROWS * (a.RS >> 2); q += blockDim.x) dst[q] = q < nvec ? src[q] : make_uint4(0, 0, 0, 0);
    }
    if (a.pass == 0)
        for (int q = threadIdx.x; q < HM_EXACT_BINS; q += blockDim.x) lhist[q] = 0u;
    if (active) hm_tile_store(tile, a.RS, tr, lane);
    __syncthreads();
    unsigned long long below_thr = 0ull;
    if (active) {
        const int j = tj * HM_TILE_ROWS + lane;
        const int i_lo = std::max(a.row_begin, ti * HM_TILE_ROWS), i_hi = std::min(a.row_end, (ti + 1) * HM_TILE_ROWS);
        for (int i = i_lo; i < i_hi; ++i) {
            const float u = hm_tile_u(tile, a.RS, a.d, fixed + (i - ti * HM_TILE_ROWS) * a.RS, a.sign_mode, lane);
            const float uc = hm::clamp_min_one(u);
            const float dd = hm::acosh_c(uc) / a.sqrt_c;
            const bool ok = j > i && j < a.n && dd < a.thr;
            const uint32_t db = hm::fbits(dd);
            if (a.pass == 0) {
                const bool in = ok && (db & a.prefix_mask) == a.prefix;
                const unsigned long long m_ok = __ballot(ok), m_in = __ballot(in);
                below_thr += (unsigned long long)__popcll(m_ok);
                if (m_in != 0ull) {
                    const uint32_t digit = (db >> a.shift) & ((1u << a.nbits) - 1u);
                    // a dense table puts a whole wave into one bin: one add for the wave then, lane-wise atomics otherwise
                    const uint32_t d0 = (uint32_t)__shfl((int)digit, __ffsll((long long)m_in) - 1, 64);
                    if (__ballot(in && digit != d0) == 0ull) {
                        if (lane == 0) atomicAdd(&lhist[d0], (uint32_t)__popcll(m_in));
                    } else if (in) {
                        atomicAdd(&lhist[digit], 1u);
                    }
                }
            } else if (a.pass == 1) {
                const unsigned long long m_at = __ballot(ok && db == a.bstar);
                if (m_at != 0ull && lane == 0) atomicAdd(&a.rowcnt[i], (uint32_t)__popcll(m_at));
            } else {
                const bool sel = ok && (db < a.bstar || (db == a.bstar && i <= a.istar));
                const unsigned long long m_sel = __ballot(sel);
                if (m_sel != 0ull) {
                    unsigned long long base = 0ull;
                    if (lane == 0) base = atomicAdd(a.emitted, (unsigned long long)__popcll(m_sel));
                    base = (unsigned long long)__shfl((long long)base, 0, 64);
                    const unsigned long long at = base + (unsigned long long)__popcll(m_sel & ((1ull << lane) - 1ull));
                    if (sel && at < (unsigned long long)a.cap) a.ent[at] = make_uint4(db, (uint32_t)i, (uint32_t)j, hm::fbits(uc));
                }
            }
        }
    }
    if (a.pass == 0) {
        if (a.total != nullptr && lane == 0 && below_thr != 0ull) atomicAdd(a.total, below_thr);
        __syncthreads();
        for (int q = threadIdx.x; q < HM_EXACT_BINS; q += blockDim.x)
            if (lhist[q]) atomicAdd(&a.hist[q], lhist[q]);
    }
}

static int hm_exact_launch(hm_engine* e, ExactArgs& a, hipStream_t s)
{
    const void* kfn = reinterpret_cast<const void*>(&hm_exact_scan_kernel);
    const size_t lds = sizeof(float) * (size_t)(1 + HM_EXACT_WAVES) * HM_TILE_ROWS * e->RS + sizeof(uint32_t) * HM_EXACT_BINS;
    if (e->attr_done.find(kfn) == e->attr_done.end()) {
        HM_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(float) * (size_t)(1 + HM_EXACT_WAVES) * HM_TILE_ROWS * 4 * HM_TILE_MAXQ + sizeof(uint32_t) * HM_EXACT_BINS)));
        e->attr_done.insert(kfn);
    }
    const int ti_first = a.row_begin / HM_TILE_ROWS, ti_last = (a.row_end - 1) / HM_TILE_ROWS;
    // row tiles in slabs of at most 32 768 (gridDim.y), partner-tile groups counted from the slab's first diagonal tile
    for (int t0 = ti_first; t0 <= ti_last; t0 += 32768) {
        const int nt = std::min(32768, ti_last - t0 + 1);
        a.ti0 = t0;
        const int groups = (a.ntj - t0 + HM_EXACT_WAVES - 1) / HM_EXACT_WAVES;
        if (groups <= 0) break;
        hipLaunchKernelGGL(hm_exact_scan_kernel, dim3((unsigned)groups, (unsigned)nt), dim3(64 * HM_EXACT_WAVES), lds, s, a);
        HM_HIP(hipGetLastError());
    }
    return HM_OK;
}

// Same contract as hm_topk_core_form (hm_search.hip): entries {distance bits, i, j, bits(u)} in e->ent, their number in
// *n_valid_emitted and e->h->ctr64[2]; every entry is a candidate (d < thr) and the k smallest keys are among them.
// *count = the exact number of candidates in the row range.
int hm_topk_exact(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, int64_t n_limit,
                  int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s)
{
    *n_valid_emitted = 0; *count = 0; *result_dev = nullptr;
    e->h->ctr64[2] = 0;
    const int64_t n = (n_limit >= 0 && n_limit < e->n) ? n_limit : e->n;
    if (row_end < 0 || row_end > n) row_end = n;
    if (row_begin < 0) row_begin = 0;
    if (row_end > n - 1) row_end = n - 1;
    if (n < 2 || row_begin >= row_end || !(thr > 0.0f)) return HM_OK;
    if (!e->d_rowcnt) HM_HIP(hipMalloc(&e->d_rowcnt, sizeof(uint32_t) * (size_t)e->max_rows));
    ExactArgs a;
    memset(&a, 0, sizeof(a));
    a.img = e->img; a.RS = e->RS; a.d = e->d; a.sign_mode = e->sign_mode;
    a.sqrt_c = sqrtf(c); a.thr = thr;
    a.n = (int)n; a.row_begin = (int)row_begin; a.row_end = (int)row_end;
    a.ntj = (int)((n + HM_TILE_ROWS - 1) / HM_TILE_ROWS);
    a.hist = e->d_hist; a.total = nullptr; a.rowcnt = e->d_rowcnt;
    a.ent = e->ent; a.emitted = e->d_ctr64 + 2; a.cap = e->ent_cap;
    // ---- 1. the k-th smallest distance's bits ----
    static const int kShift[3] = {21, 10, 0}, kBits[3] = {11, 11, 10};
    uint64_t below = 0, match = 0, total = 0;
    int64_t want = k;
    uint32_t prefix = 0, mask = 0;
    for (int level = 0; level < 3; ++level) {
        HM_HIP(hipMemsetAsync(e->d_hist, 0, sizeof(uint32_t) * HM_EXACT_BINS, s));
        a.pass = 0; a.prefix = prefix; a.prefix_mask = mask; a.shift = kShift[level]; a.nbits = kBits[level];
        a.total = nullptr;
        if (level == 0) {
            HM_HIP(hipMemsetAsync(e->d_ctr64, 0, sizeof(unsigned long long), s));
            a.total = e->d_ctr64;
        }
        int rc = hm_exact_launch(e, a, s);
        if (rc) return rc;
        HM_HIP(hipMemcpyAsync(e->h->hist, e->d_hist, sizeof(uint32_t) * HM_EXACT_BINS, hipMemcpyDeviceToHost, s));
        if (level == 0) HM_HIP(hipMemcpyAsync(e->h->ctr64, e->d_ctr64, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        if (level == 0) {
            total = e->h->ctr64[0];
            *count = (int64_t)total;
            want = std::min<int64_t>(k, (int64_t)total);
            if (want <= 0) return HM_OK;                      // a pure count, or no candidate
        }
        const uint32_t nb = 1u << kBits[level];
        uint64_t cum = below;
        uint32_t dsel = nb - 1;
        match = 0;
        for (uint32_t q = 0; q < nb; ++q) {
            if (cum + e->h->hist[q] >= (uint64_t)want) { dsel = q; match = e->h->hist[q]; break; }
            cum += e->h->hist[q];
        }
        below = cum;
        prefix |= dsel << kShift[level];
        mask |= (nb - 1u) << kShift[level];
    }
    a.bstar = prefix;
    a.istar = 0x7fffffff;
    // ---- 2. a tie flood at bstar: rows up to which its pairs are needed ----
    if (below + match > (uint64_t)e->ent_cap) {
        HM_HIP(hipMemsetAsync(e->d_rowcnt, 0, sizeof(uint32_t) * (size_t)n, s));
        a.pass = 1;
        int rc = hm_exact_launch(e, a, s);
        if (rc) return rc;
        std::vector<uint32_t> rows((size_t)n);
        HM_HIP(hipMemcpyAsync(rows.data(), e->d_rowcnt, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        const uint64_t need = (uint64_t)want - below;
        uint64_t cum = 0;
        for (int64_t i = row_begin; i < row_end; ++i) {
            cum += rows[(size_t)i];
            if (cum >= need) { a.istar = (int)i; break; }
        }
        if (below + cum > (uint64_t)e->ent_cap)
            return hm_fail(e, HM_E_CAPACITY, "exact search: more equal-distance pairs in one row than the emission buffer holds");
    }
    // ---- 3. emission ----
    HM_HIP(hipMemsetAsync(e->d_ctr64 + 2, 0, sizeof(unsigned long long), s));
    a.pass = 2;
    int rc = hm_exact_launch(e, a, s);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(&e->h->ctr64[2], e->d_ctr64 + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    if (e->h->ctr64[2] > (unsigned long long)e->ent_cap) return hm_fail(e, HM_E_CAPACITY, "exact search: emission count exceeds its own bound");
    *n_valid_emitted = (int64_t)e->h->ctr64[2];
    *result_dev = e->ent;
    e->last_emitted = (int64_t)e->h->ctr64[2];
    return HM_OK;
}
