// hm_tokenize.hip -- batch form of HyperbolicTokenizer.tokenize (tokenizer/hyperbolic_merge.py:414-446).
//
// The reference starts from list(text) and repeats left-to-right passes over the token list: at position i the
// pair (tokens[i], tokens[i+1]) is looked up in {(old1, old2): new}; on a hit tokens[i] becomes `new`, position i+1
// is removed and i stays (so the merged token is tried against its new neighbour at once); on a miss i advances.
// Passes repeat until one changes nothing.  One pass is therefore a streaming fold: a held token `cur` is merged
// with the next token while a rule matches, else emitted.  That is what one lane does here for one line, over
// 32-bit symbol ids (the caller's mapping string -> symbol; negative symbols stand for characters no rule or
// vocabulary entry mentions and never match).  Lines are independent: one lane per line, rules in an
// open-addressing table that stays in L2, side flags to skip the lookup when a symbol is never a left / right
// operand.  Integer work bound by memory latency, not by arithmetic.
#include "hm_common.h"
#include <vector>

namespace {

constexpr uint64_t HM_TOK_EMPTY = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ __forceinline__ uint64_t hm_tok_key(int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }
__host__ __device__ __forceinline__ uint64_t hm_tok_slot(uint64_t key, int shift) { return (key * 0x9E3779B97F4A7C15ull) >> shift; }

struct TokArgs {
    const int32_t* in;        // symbols of all lines, concatenated
    const int64_t* offsets;   // [n_lines + 1]
    const int64_t* order;     // optional: line handled by thread t (longest first keeps a wave's lanes alike)
    int64_t n_lines;
    const uint64_t* keys;     // rule table: key (a << 32 | b), HM_TOK_EMPTY = free
    const int32_t* vals;
    int shift;                // 64 - log2(capacity)
    uint64_t mask;            // capacity - 1
    const uint8_t* flags;     // [n_sym]: bit 0 = occurs as a left operand, bit 1 = as a right operand
    int32_t* out;             // same layout as `in`; line l occupies out[offsets[l] .. offsets[l] + out_len[l])
    int32_t* out_len;         // [n_lines]
    int32_t* passes;          // [n_lines] or nullptr: passes the reference's while-loop runs for the line
};

__device__ __forceinline__ int32_t hm_tok_lookup(const TokArgs& a, int32_t x, int32_t y)
{
    if ((x | y) < 0) return -1;
    if (!(a.flags[x] & 1) || !(a.flags[y] & 2)) return -1;
    const uint64_t key = hm_tok_key(x, y);
    uint64_t s = hm_tok_slot(key, a.shift);
    for (;;) {
        const uint64_t k = a.keys[s];
        if (k == key) return a.vals[s];
        if (k == HM_TOK_EMPTY) return -1;          // the table is never full: the loop ends
        s = (s + 1) & a.mask;
    }
}

__global__ __launch_bounds__(64) void hm_tokenize_kernel(TokArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= a.n_lines) return;
    const int64_t l = a.order ? a.order[t] : t;
    const int64_t base = a.offsets[l];
    int64_t len = a.offsets[l + 1] - base;
    const int32_t* src = a.in + base;
    int32_t* dst = a.out + base;
    int32_t np = 0;
    bool changed = true;
    while (changed) {
        changed = false;
        int64_t w = 0;
        if (len > 0) {
            int32_t cur = src[0];
            for (int64_t r = 1; r < len; ++r) {
                const int32_t nx = src[r];
                const int32_t m = hm_tok_lookup(a, cur, nx);
                if (m >= 0) { cur = m; changed = true; }
                else { dst[w++] = cur; cur = nx; }          // w <= r - 1: in place is safe once src == dst
            }
            dst[w++] = cur;
        }
        len = w;
        src = dst;
        ++np;
    }
    a.out_len[l] = (int32_t)len;
    if (a.passes) a.passes[l] = np;
}

}  // namespace

extern "C" int64_t hm_tokenize_table_capacity(int64_t n_rules)
{
    int64_t cap = 16;
    while (cap < 2 * n_rules + 2) cap <<= 1;
    return cap;
}

extern "C" int hm_tokenize_build_table(const int32_t* left, const int32_t* right, const int32_t* merged, int64_t n_rules, int64_t n_sym,
                                       uint64_t* keys_out, int32_t* vals_out, int64_t capacity, uint8_t* flags_out)
{
    if (n_rules < 0 || n_sym < 0 || capacity < 16 || (capacity & (capacity - 1)) || capacity < 2 * n_rules + 2)
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: bad sizes (capacity must be a power of two >= 2 * n_rules + 2)");
    if ((n_rules && (!left || !right || !merged)) || !keys_out || !vals_out || (n_sym && !flags_out))
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: NULL pointer");
    int shift = 64;
    for (int64_t c = capacity; c > 1; c >>= 1) --shift;
    for (int64_t s = 0; s < capacity; ++s) { keys_out[s] = HM_TOK_EMPTY; vals_out[s] = -1; }
    for (int64_t s = 0; s < n_sym; ++s) flags_out[s] = 0;
    for (int64_t r = 0; r < n_rules; ++r) {
        const int32_t x = left[r], y = right[r], z = merged[r];
        if (x < 0 || y < 0 || z < 0 || x >= n_sym || y >= n_sym || z >= n_sym)
            return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_build_table: symbol out of range");
        const uint64_t key = hm_tok_key(x, y);
        uint64_t s = hm_tok_slot(key, shift);
        while (keys_out[s] != HM_TOK_EMPTY && keys_out[s] != key) s = (s + 1) & (uint64_t)(capacity - 1);
        keys_out[s] = key;
        vals_out[s] = z;                              // a later rule for the same pair replaces the earlier one (dict assignment)
        flags_out[x] |= 1;
        flags_out[y] |= 2;
    }
    return HM_OK;
}

extern "C" int hm_tokenize_batch(const int32_t* sym_dev, const int64_t* offsets_dev, const int64_t* order_dev, int64_t n_lines,
                                 const uint64_t* keys_dev, const int32_t* vals_dev, int64_t capacity, const uint8_t* flags_dev,
                                 int32_t* out_dev, int32_t* out_len_dev, int32_t* passes_dev, void* stream)
{
    if (n_lines < 0 || capacity < 16 || (capacity & (capacity - 1)))
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: bad sizes");
    if (n_lines == 0) return HM_OK;
    if (!offsets_dev || !keys_dev || !vals_dev || !flags_dev || !out_len_dev)
        return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: NULL pointer");
    if (n_lines > (int64_t)0x7FFFFFFF * 64) return hm_fail(nullptr, HM_E_ARG, "hm_tokenize_batch: too many lines");
    TokArgs a;
    a.in = sym_dev; a.offsets = offsets_dev; a.order = order_dev; a.n_lines = n_lines;
    a.keys = keys_dev; a.vals = vals_dev; a.mask = (uint64_t)(capacity - 1);
    a.shift = 64;
    for (int64_t c = capacity; c > 1; c >>= 1) --a.shift;
    a.flags = flags_dev; a.out = out_dev; a.out_len = out_len_dev; a.passes = passes_dev;
    const unsigned grid = (unsigned)((n_lines + 63) / 64);
    hipLaunchKernelGGL(hm_tokenize_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return hm_fail(nullptr, (int)err, "hm_tokenize_batch: launch failed");
    return HM_OK;
}
