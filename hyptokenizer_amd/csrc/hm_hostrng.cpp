// hm_hostrng.cpp -- host-side helper of the enhanced tokenizer: the first `ns` entries of torch.randperm(n),
// `count` times in a row, drawn from (and advancing) the state of torch's CPU generator.
//
// Reference call site: indices = torch.randperm(self.current_vocab_size)[:sample_size], once per scored
// candidate (tokenizer/enhanced_fast_hyperbolic_merge.py:324-325).  torch's randperm_cpu (ATen
// native/TensorFactories.cpp, n < 2^32 / 20) is a forward Fisher-Yates shuffle -- for i in [0, n-1):
// z = mt19937() % (n - i); swap(r[i], r[i + z]) -- so entry i is final after step i: the first ns entries need
// the first ns swaps on a sparse view of the array, and the remaining n - 1 - ns draws only have to be skipped.
// The generator is at::mt19937 (c10/.../MT19937RNGEngine.h): the standard MT19937 recurrence with torch's
// left / next bookkeeping, restated here.  At n = 100 000 torch.randperm costs ~16 ms (a random-access shuffle of
// an 800 kB array); this costs the ~0.1 ms it takes to run the recurrence over n draws.
// No GPU work: plain C++ (part of libhypmerge.so because the Python host calls it next to hm_coherence_batch).
#include <stdint.h>
#include <string.h>

#include "../../include/hypmerge.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;

inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((0u - (v & 1u)) & MATRIX_A); }

// One regeneration of the 624-word block.  Skipping the ~n draws a randperm(n) call makes beyond the prefix is 160 of
// these at n = 100 000, i.e. nearly all of the helper's time, and the recurrence vectorises: word k reads k, k + 1 and
// k + 397 (not yet rewritten) in the first 227 words, and k - 227 (rewritten 227 words earlier) afterwards -- no
// dependence closer than 227 words.  Compiled for AVX-512, AVX2 and the baseline ISA; the loader picks at run time.
__attribute__((target_clones("avx512f", "avx2", "default"), optimize("O3", "tree-vectorize")))
void mt_regenerate(uint32_t* __restrict__ st)
{
    constexpr int A = MT_N - MT_M;                 // 227
#pragma GCC ivdep
    for (int k = 0; k < A; ++k) st[k] = st[k + MT_M] ^ twist(st[k], st[k + 1]);
    // words [227, 623): blocks of 227 so that every read of st[k - 227] sees a finished word
    for (int b = A; b < MT_N - 1; b += A) {
        const int e = b + A < MT_N - 1 ? b + A : MT_N - 1;
#pragma GCC ivdep
        for (int k = b; k < e; ++k) st[k] = st[k - A] ^ twist(st[k], st[k + 1]);
    }
    st[MT_N - 1] = st[MT_M - 1] ^ twist(st[MT_N - 1], st[0]);
}

struct Mt {
    uint32_t* state;   // 624 words
    int left;
    uint32_t next;

    void next_state()
    {
        left = MT_N;
        next = 0;
        mt_regenerate(state);
    }

    inline uint32_t draw()
    {
        if (--left == 0) next_state();
        uint32_t y = state[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }

    void skip(int64_t k)
    {
        while (k > 0) {
            if (left == 1) {                   // the next draw regenerates the block and takes its first word
                left = 0;
                next_state();
                next = 1;
                --k;
            } else {
                const int64_t m = k < (int64_t)(left - 1) ? k : (int64_t)(left - 1);
                left -= (int)m;
                next += (uint32_t)m;
                k -= m;
            }
        }
    }
};
}  // namespace

extern "C" int hm_randperm_prefix(uint32_t* mt_state, int32_t* left, uint32_t* next, int64_t n, int32_t ns, int64_t count,
                                  int32_t* out)
{
    if (!mt_state || !left || !next || !out || n < 1 || n >= (int64_t)(0xffffffffu / 20) || ns < 0 || ns > n || count < 0 ||
        ns > 4096 || *left < 1 || *left > MT_N || *next > (uint32_t)MT_N)
        return HM_E_ARG;
    Mt g;
    g.state = mt_state; g.left = *left; g.next = *next;
    // sparse view of the shuffled array: (position, value) pairs for the positions that differ from the identity
    int64_t pos[4096 * 2];
    int64_t val[4096 * 2];
    for (int64_t t = 0; t < count; ++t) {
        int used = 0;
        auto get = [&](int64_t p) -> int64_t {
            for (int q = used - 1; q >= 0; --q)
                if (pos[q] == p) return val[q];
            return p;
        };
        auto put = [&](int64_t p, int64_t v) {
            for (int q = 0; q < used; ++q)
                if (pos[q] == p) { val[q] = v; return; }
            pos[used] = p; val[used] = v; ++used;
        };
        const int64_t steps = n - 1;               // draws of one randperm call
        const int64_t head = ns < steps ? ns : steps;
        for (int64_t i = 0; i < head; ++i) {
            const int64_t z = (int64_t)(g.draw() % (uint32_t)(n - i));
            const int64_t vi = get(i), vj = get(i + z);
            put(i + z, vi);
            put(i, vj);
        }
        for (int32_t i = 0; i < ns; ++i) out[t * ns + i] = (int32_t)get(i);
        g.skip(steps - head);
    }
    *left = g.left; *next = g.next;
    return HM_OK;
}
