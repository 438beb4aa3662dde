// hm_tokenize.hip -- batch form of HyperbolicTokenizer.tokenize (tokenizer/hyperbolic_merge.py:414-446).
//
// The reference starts from list(text) and repeats left-to-right passes over the token list: at position i the
// pair (tokens[i], tokens[i+1]) is looked up in {(old1, old2): new}; on a hit tokens[i] becomes `new`, position i+1
// is removed and i stays (so the merged token is tried against its new neighbour at once); on a miss i advances.
// Passes repeat until one changes nothing.  One pass is therefore a streaming fold: a held token `cur` is merged
// with the next token while a rule matches, else emitted.  That is what one lane does here for one line, over
// 32-bit symbol ids (the caller's mapping string -> symbol; negative symbols stand for characters no rule or
// vocabulary entry mentions and never match).  Lines are independent: one lane per line, rules in an
// open-addressing table of 8-byte entries that stays in L2.  Integer work bound by memory latency, not by arithmetic.
#include "hm_common.h"
#include "hm_rows_device.h"

namespace {

constexpr int HM_TOK_SYM_BITS = 21;            // symbols are < 2^21 - 1
constexpr int HM_TOK_VAL_BITS = 22;
constexpr uint64_t HM_TOK_VAL_MASK = (1ull << HM_TOK_VAL_BITS) - 1;

// Rule table: 8-byte entries  left:21 | right:21 | merged + 1 : 22  (0 = free), two entries per 16-byte bucket, at
// most one entry per four buckets on average.  A probe is ONE 16-byte load unless the bucket is full of other keys
// (about 0.2 % of the buckets): the fold below is a chain of dependent probes, and a wave waits for the slowest of
// its 64 lanes at every step, so the length of the longest probe sequence among 64 lanes -- not the mean -- sets the
// pace.  (Measured on the first form of this kernel, 16-byte entries with linear probing at load 1/2: 1.48 ms for
// the bench batch; load 1/4: 1.03 ms; extra look-ahead probes made it slower, an LDS cache of results did not help.)
__host__ __device__ __forceinline__ uint64_t hm_tok_tag(int32_t x, int32_t y) { return ((uint64_t)(uint32_t)x << HM_TOK_SYM_BITS) | (uint32_t)y; }
__host__ __device__ __forceinline__ uint64_t hm_tok_bucket(uint64_t tag, int shift) { return (tag * 0x9E3779B97F4A7C15ull) >> shift; }

struct TokArgs {
    const int32_t* in;        // symbols of all lines, concatenated
    const int64_t* offsets;   // [n_lines + 1]
    const int64_t* order;     // optional: line handled by thread t (longest first keeps a wave's lanes alike)
    int64_t n_lines;
    const uint4* table;       // buckets of two entries
    int shift;                // 64 - log2(number of buckets)
    uint32_t mask;            // number of buckets - 1
    int32_t* out;             // same layout as `in`; line l occupies out[offsets[l] .. offsets[l] + out_len[l])
    int32_t* out_len;         // [n_lines]
    int32_t* passes;          // [n_lines] or nullptr: passes the reference's while-loop runs for the line
};

__device__ __forceinline__ int32_t hm_tok_lookup(const TokArgs& a, int32_t x, int32_t y)
{
    if ((x | y) < 0) return -1;                    // characters outside every rule and the vocabulary
    const uint64_t tag = hm_tok_tag(x, y);
    uint32_t b = (uint32_t)hm_tok_bucket(tag, a.shift);
    for (;;) {
        const uint4 q = a.table[b];
        const uint64_t e0 = ((uint64_t)q.y << 32) | q.x, e1 = ((uint64_t)q.w << 32) | q.z;
        if (e0 != 0 && (e0 >> HM_TOK_VAL_BITS) == tag) return (int32_t)(e0 & HM_TOK_VAL_MASK) - 1;
        if (e1 != 0 && (e1 >> HM_TOK_VAL_BITS) == tag) return (int32_t)(e1 & HM_TOK_VAL_MASK) - 1;
        if (e0 == 0 || e1 == 0) return -1;         // a bucket with a free slot ends every probe sequence
        b = (b + 1) & a.mask;
    }
}

#ifndef HM_TOK_CH_SET
#define HM_TOK_CH_SET 32
#endif
constexpr int HM_TOK_CH = HM_TOK_CH_SET;       // symbols per line and staging round (32: one 128-byte segment per line)
constexpr int HM_TOK_LD = HM_TOK_CH + 1;       // odd row stride: lane-private reads of column k hit 64 different banks
constexpr int HM_TOK_G = 8;                    // symbols read from the lane's LDS row per group

// One wave = 64 lines, one lane per line.  A lane walking its own line in global memory would touch 64 different
// cache lines per instruction and keep 64 x 128 bytes live per wave -- far more than a wave's share of L1 / L2, so
// every 4-byte read would fetch a line again (measured: 7.5 ms for the bench batch against 3.8 ms staged).  But the
// read position r is the SAME for all lanes (each step consumes exactly one input symbol, whatever the rules do),
// so the wave stages [r0, r0 + 32) of all its lines into LDS with coalesced segment loads (two lines per
// instruction), every lane folds its 32 symbols out of LDS, and the tokens emitted in the round go back through
// the same LDS rows (emitted <= consumed) as coalesced segment stores.
__global__ __launch_bounds__(64) void hm_tokenize_kernel(TokArgs a)
{
    __shared__ int32_t buf[64 * HM_TOK_LD];
    __shared__ int64_t sbase[64];
    __shared__ int32_t slen[64], sw[64], scnt[64];
    const int lane = threadIdx.x;
    const int64_t t = (int64_t)blockIdx.x * 64 + lane;
    const bool mine = t < a.n_lines;
    const int64_t l = mine ? (a.order ? a.order[t] : t) : 0;
    const int64_t base = mine ? a.offsets[l] : 0;
    int32_t len = mine ? (int32_t)(a.offsets[l + 1] - base) : 0;      // the caller guarantees < 2^31 symbols per line
    sbase[lane] = base;
    bool active = mine;
    int32_t np = 0;
    const int32_t* src = a.in;
    constexpr int LPI = 64 / HM_TOK_CH;                                 // lines per staging instruction
    const int half = lane / HM_TOK_CH, col = lane % HM_TOK_CH;
    int32_t* row = buf + lane * HM_TOK_LD;
    while (__any(active)) {
        slen[lane] = active ? len : 0;
        sw[lane] = 0;
        int32_t mx = active ? len : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        hm_wave_lds_sync();
        bool changed = false;
        int32_t cur = -1, w = 0;
        for (int32_t r0 = 0; r0 < mx; r0 += HM_TOK_CH) {
            // stage: lines q + half, symbols [r0, r0 + CH); -1 beyond the end of a line
            int32_t v[HM_TOK_CH];
#pragma unroll
            for (int q = 0; q < 64; q += LPI) {
                const int line = q + half;
                const int32_t r = r0 + col;
                v[q / LPI] = r < slen[line] ? src[sbase[line] + r] : -1;
            }
#pragma unroll
            for (int q = 0; q < 64; q += LPI) buf[(q + half) * HM_TOK_LD + col] = v[q / LPI];
            hm_wave_lds_sync();
            // fold
            int32_t cnt = 0;
            const int32_t left = active ? len - r0 : 0;               // symbols of this line in the round (may be <= 0)
#pragma unroll 1
            for (int g = 0; g < HM_TOK_CH && g < left; g += HM_TOK_G) {
                int32_t nx[HM_TOK_G];
#pragma unroll
                for (int k = 0; k < HM_TOK_G; ++k) nx[k] = row[g + k];
#pragma unroll
                for (int k = 0; k < HM_TOK_G; ++k) {
                    if (g + k < left) {
                        if (r0 + g + k == 0) cur = nx[k];
                        else {
                            const int32_t m = hm_tok_lookup(a, cur, nx[k]);
                            if (m >= 0) { cur = m; changed = true; }
                            else { row[cnt++] = cur; cur = nx[k]; }   // cnt <= g + k: that slot has been consumed
                        }
                    }
                }
            }
            scnt[lane] = cnt;
            hm_wave_lds_sync();
            // flush: tokens emitted in this round, positions [sw, sw + scnt) < r0 + CH (already read: in place is safe)
#pragma unroll 8
            for (int q = 0; q < 64; q += LPI) {
                const int line = q + half;
                if (col < scnt[line]) a.out[sbase[line] + sw[line] + col] = buf[line * HM_TOK_LD + col];
            }
            w += cnt;
            hm_wave_lds_sync();
            sw[lane] = w;
        }
        if (active) {
            if (len > 0) a.out[base + w++] = cur;          // the held token ends the line
            len = w;
            ++np;
            active = changed;
        }
        src = a.out;
        __threadfence_block();                             // the next pass reads what other lanes of this wave stored
    }
    if (mine) {
        a.out_len[l] = len;
        if (a.passes) a.passes[l] = np;
    }
}

}  // namespace

extern "C" int64_t hm_tokenize_table_capacity(int64_t n_rules)
{
    int64_t cap = 16;
    while (cap < 8 * n_rules) cap <<= 1;
    return cap;
}

extern "C" int hm_tokenize_build_table(const int32_t* left, const int32_t* right, const int32_t* merged, int64_t n_rules,
                                       uint64_t* table_out, int64_t capacity)
{
    if (n_rules < 0 || capacity < 16 || (capacity & (capacity - 1)) || capacity < 8 * n_rules || capacity > ((int64_t)1 << 32))
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: bad sizes (capacity must be a power of two >= max(16, 8 * n_rules))");
    if ((n_rules && (!left || !right || !merged)) || !table_out)
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: NULL pointer");
    const int64_t n_buckets = capacity / 2;
    int shift = 64;
    for (int64_t c = n_buckets; c > 1; c >>= 1) --shift;
    for (int64_t s = 0; s < capacity; ++s) table_out[s] = 0;
    const int32_t lim = (1 << HM_TOK_SYM_BITS) - 1;
    for (int64_t r = 0; r < n_rules; ++r) {
        const int32_t x = left[r], y = right[r], z = merged[r];
        if (x < 0 || y < 0 || z < 0 || x >= lim || y >= lim || z >= lim)
            return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: rule symbols must lie in [0, 2^21 - 1)");
        const uint64_t tag = hm_tok_tag(x, y);
        const uint64_t entry = (tag << HM_TOK_VAL_BITS) | (uint64_t)(z + 1);
        uint64_t b = hm_tok_bucket(tag, shift);
        for (;;) {
            uint64_t* e = table_out + 2 * b;
            int slot = -1;
            for (int q = 0; q < 2 && slot < 0; ++q)
                if (e[q] == 0 || (e[q] >> HM_TOK_VAL_BITS) == tag) slot = q;
            if (slot >= 0) { e[slot] = entry; break; }   // a later rule for the same pair replaces the earlier one (dict assignment)
            b = (b + 1) & (uint64_t)(n_buckets - 1);
        }
    }
    return HM_OK;
}

extern "C" int hm_tokenize_batch(const int32_t* sym_dev, const int64_t* offsets_dev, const int64_t* order_dev, int64_t n_lines,
                                 const uint64_t* table_dev, int64_t capacity, int32_t* out_dev, int32_t* out_len_dev,
                                 int32_t* passes_dev, void* stream)
{
    if (n_lines < 0 || capacity < 16 || (capacity & (capacity - 1)) || capacity > ((int64_t)1 << 32))
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: bad sizes");
    if (n_lines == 0) return HM_OK;
    if (!offsets_dev || !table_dev || !out_len_dev) return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: NULL pointer");
    if (((uintptr_t)table_dev & 15) != 0) return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: table_dev must be 16-byte aligned");
    if (n_lines > (int64_t)0x7FFFFFFF * 64) return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: too many lines");
    TokArgs a;
    a.in = sym_dev; a.offsets = offsets_dev; a.order = order_dev; a.n_lines = n_lines;
    a.table = reinterpret_cast<const uint4*>(table_dev);
    const int64_t n_buckets = capacity / 2;
    a.mask = (uint32_t)(n_buckets - 1);
    a.shift = 64;
    for (int64_t c = n_buckets; c > 1; c >>= 1) --a.shift;
    a.out = out_dev; a.out_len = out_len_dev; a.passes = passes_dev;
    const unsigned grid = (unsigned)((n_lines + 63) / 64);
    hipLaunchKernelGGL(hm_tokenize_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return hm_fail(nullptr, (int)err, "hm_tokenize_batch: launch failed");
    return HM_OK;
}
