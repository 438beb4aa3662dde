/*
 * hm_oracle.c -- CPU restatement of HypTokenizer's merge-candidate hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (hyptokenizer_amd/) never links, imports or calls anything in oracle/.
 *
 * What it restates (reference = /root/reference, read-only, never copied):
 *   embedding/lorentz_model.py:14-25    minkowski_dot      -> hmo_minkowski_u
 *   embedding/lorentz_model.py:41-56    project_to_hyperboloid -> hmo_project
 *   embedding/lorentz_model.py:73-93    exp_map            -> hmo_exp_map
 *   embedding/lorentz_model.py:96-119   log_map            -> hmo_log_map
 *   embedding/lorentz_model.py:122-138  distance           -> hmo_distance / hmo_pair_distance
 *   embedding/lorentz_model.py:141-178  batch_distance     -> hmo_batch_distance
 *   tokenizer/hyperbolic_merge.py:192-291  _find_merge_candidates (i<j, d<thr, row-major)
 *                                          -> hmo_pairwise_candidates / hmo_pairwise_count
 *   tokenizer/hyperbolic_merge.py:378      stable sort by distance, [0]  -> hmo_pairwise_topk (k=1)
 *   tokenizer/fast_hyperbolic_merge.py:336-374  sort + cache top-10000   -> hmo_pairwise_topk
 *   tokenizer/hyperbolic_merge.py:309-340  weighted "midpoint"           -> hmo_midpoint
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * golden vectors produced by importing the reference in the build container
 * (tests/golden/make_golden.py; literal mode and sign-corrected "lorentz" mode).
 *
 * Canonical arithmetic (DESIGN.md "Canonical arithmetic").  The reference runs on torch CPU
 * kernels whose fp32 summation order is an implementation detail of the torch build; it cannot be
 * reproduced bit-for-bit by any other program.  The restatement therefore fixes ONE fully
 * specified fp32 evaluation order, which the HIP kernels reproduce bit-for-bit:
 *   S   = fmaf chain over the spatial coordinates k = 1..d in ascending k, starting from +0
 *   M   = fmaf(x0, y0, -S)                         (= x0*y0 - sum, reference sign, line :25)
 *   u   = -M (sign_mode 0, reference as shipped)   or  +M (sign_mode 1, "lorentz")
 *   a   = clamp_min(u, 1.0f)  with NaN propagation (1.0 + 1e-8 == 1.0f in fp32, line :135)
 *   d   = acosh(a) / sqrtf((float)c),  acosh(a) = log1p(t + sqrt(t*(t+2))), t = a - 1
 * log1p / expm1 follow the published fdlibm (FreeBSD msun) float algorithms, restated below
 * with every operation in fp32 and no FMA contraction (build with -ffp-contract=off), so that
 * gcc on x86-64 and hipcc on gfx950 produce identical bits.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -mfma -mavx2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HMO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* bit helpers                                                                                */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------ */
/* canonical transcendental functions (fdlibm float algorithms, fp32 only)                    */
/* ------------------------------------------------------------------------------------------ */

/* log(1+x) for x >= 0 (NaN/inf pass through).  fdlibm s_log1pf algorithm. */
HMO_EXPORT float hmo_log1pf(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    int k = 1;
    float f = 0.0f, c = 0.0f;
    if (ix >= 0x7f800000u) {            /* +inf, NaN, or negative: not used by acosh, pass/NaN */
        if (ix == 0x7f800000u || (ix & 0x7fffffffu) > 0x7f800000u) return x;
        if (ix == 0x80000000u) return x; /* -0 */
        return u2f(0x7fc00000u);         /* negative argument: outside our domain */
    }
    if (ix < 0x3ed413d0u) {             /* 1+x < sqrt(2)+ */
        if (ix < 0x33800000u) return x; /* x < 2**-24 */
        k = 0; f = x; c = 0.0f;
    }
    if (k) {
        float uf = 1.0f + x;
        uint32_t iu = f2u(uf);
        iu += 0x3f800000u - 0x3f3504f3u;
        k = (int)(iu >> 23) - 0x7f;
        if (k < 25) {
            c = (k >= 2) ? (1.0f - (uf - x)) : (x - (uf - 1.0f));
            c = c / uf;
        } else {
            c = 0.0f;
        }
        iu = (iu & 0x007fffffu) + 0x3f3504f3u;
        f = u2f(iu) - 1.0f;
    }
    {
        float s = f / (2.0f + f);
        float z = s * s;
        float w = z * z;
        float t1 = w * (Lg2 + w * Lg4);
        float t2 = z * (Lg1 + w * Lg3);
        float R = t2 + t1;
        float hfsq = (0.5f * f) * f;
        float dk = (float)k;
        return ((s * (hfsq + R) + (dk * ln2_lo + c)) - hfsq + f) + dk * ln2_hi;
    }
}

/* acosh(a) for a >= 1 (NaN passes through). */
HMO_EXPORT float hmo_acoshf(float a)
{
    if (a != a) return a;
    if (a > 1.0e9f) return hmo_log1pf(a + a);          /* log(2a), rel. error < 1e-10 */
    {
        float t = a - 1.0f;
        float q = t * (t + 2.0f);
        float y = t + sqrtf(q);
        return hmo_log1pf(y);
    }
}

/* exp(x)-1.  fdlibm s_expm1f algorithm. */
HMO_EXPORT float hmo_expm1f(float x)
{
    const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                invln2 = 1.4426950216e+00f, Q1 = -3.3333212137e-2f, Q2 = 1.5807170421e-3f;
    uint32_t hx = f2u(x);
    int sign = (int)(hx >> 31);
    int k;
    float hi, lo, c = 0.0f, t, e, hfx, hxs, r1, twopk, y;
    hx &= 0x7fffffffu;
    if (hx >= 0x4195b844u) {            /* |x| >= 27 ln2 */
        if (hx > 0x7f800000u) return x; /* NaN */
        if (sign) return -1.0f;
        if (x > o_threshold) return x * 0x1p127f;
    }
    if (hx > 0x3eb17218u) {             /* |x| > 0.5 ln2 */
        if (hx < 0x3F851592u) {         /* |x| < 1.5 ln2 */
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
    } else if (hx < 0x33000000u) {      /* |x| < 2**-25 */
        return x;
    } else {
        k = 0;
    }
    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = 1.0f + hxs * (Q1 + hxs * Q2);
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
    twopk = u2f((uint32_t)(0x7f + k) << 23);
    if (k < 0 || k > 56) {
        y = x - e + 1.0f;
        if (k == 128) y = y * 2.0f * 0x1p127f; else y = y * twopk;
        return y - 1.0f;
    }
    {
        float tk = u2f((uint32_t)(0x7f - k) << 23); /* 2^-k */
        if (k < 23) y = (x - e + (1.0f - tk)) * twopk;
        else        y = (x - e - tk + 1.0f) * twopk;
    }
    return y;
}

/* cosh(x), sinh(x) for x >= 0 (exp_map feeds them a norm >= 1e-4).  fdlibm e_coshf / e_sinhf
 * structure with exp(x) taken as expm1(x)+1. */
HMO_EXPORT float hmo_coshf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) x = -x;
    if (x < 0.34657359f) {              /* 0.5 ln2 */
        float t = hmo_expm1f(x);
        float w = 1.0f + t;
        if (x < 0.000244140625f) return 1.0f; /* 2**-12: cosh = 1 to fp32 */
        return 1.0f + (t * t) / (w + w);
    }
    {
        float t = hmo_expm1f(x) + 1.0f;
        return 0.5f * t + 0.5f / t;
    }
}

HMO_EXPORT float hmo_sinhf(float x)
{
    float h = 0.5f, t;
    if (x != x) return x;
    if (x < 0.0f) { h = -0.5f; x = -x; }
    t = hmo_expm1f(x);
    if (x < 1.0f) return h * (2.0f * t - (t * t) / (t + 1.0f));
    return h * (t + t / (t + 1.0f));
}

/* ------------------------------------------------------------------------------------------ */
/* Lorentz primitives                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* lorentz_model.py:14-25 with the sign switch of SURVEY F5.
 * sign_mode 0: u = -(x0*y0 - sum)  (what distance/log_map feed to acosh as shipped)
 * sign_mode 1: u = +(x0*y0 - sum)  (standard Lorentz form) */
HMO_EXPORT float hmo_minkowski_u(const float* x, const float* y, int d1, int sign_mode)
{
    float acc = 0.0f;
    int k;
    for (k = 1; k < d1; ++k) acc = fmaf(x[k], y[k], acc);
    {
        float m = fmaf(x[0], y[0], -acc);       /* x0*y0 - S : reference minkowski_dot */
        return sign_mode ? m : -m;
    }
}

static inline float clamp_min_one(float u)      /* torch.clamp(min=1+1e-8) in fp32, NaN propagates */
{
    if (u != u) return u;
    return (u < 1.0f) ? 1.0f : u;
}

static inline float dist_from_u(float u, float sqrt_c)
{
    return hmo_acoshf(clamp_min_one(u)) / sqrt_c;
}

/* lorentz_model.py:122-138 */
HMO_EXPORT float hmo_distance(const float* x, const float* y, int d1, float c, int sign_mode)
{
    return dist_from_u(hmo_minkowski_u(x, y, d1, sign_mode), sqrtf(c));
}

/* lorentz_model.py:141-178 (and :181-210): all pairs, out[n1*n2] row-major. */
HMO_EXPORT void hmo_batch_distance(const float* X, int64_t n1, const float* Y, int64_t n2, int64_t ld,
                                   int d1, float c, int sign_mode, float* out)
{
    const float sc = sqrtf(c);
    int64_t i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < n1; ++i) {
        int64_t j;
        for (j = 0; j < n2; ++j)
            out[i * n2 + j] = dist_from_u(hmo_minkowski_u(X + i * ld, Y + j * ld, d1, sign_mode), sc);
    }
}

/* gathered pair distances (fast_hyperbolic_merge.py:448-455 calls distance() per sampled pair) */
HMO_EXPORT void hmo_pair_distance(const float* X, int64_t ld, int d1, const int32_t* I, const int32_t* J,
                                  int64_t b, float c, int sign_mode, float* out)
{
    const float sc = sqrtf(c);
    int64_t t;
    for (t = 0; t < b; ++t)
        out[t] = dist_from_u(hmo_minkowski_u(X + (int64_t)I[t] * ld, X + (int64_t)J[t] * ld, d1, sign_mode), sc);
}

/* one row against rows [0, n) */
HMO_EXPORT void hmo_row_vs_all(const float* X, int64_t n, int64_t ld, int d1, int64_t row, float c,
                               int sign_mode, float* out)
{
    const float sc = sqrtf(c);
    int64_t j;
#pragma omp parallel for schedule(static)
    for (j = 0; j < n; ++j)
        out[j] = dist_from_u(hmo_minkowski_u(X + row * ld, X + j * ld, d1, sign_mode), sc);
}

/* lorentz_model.py:96-119.  out_k = coef * (y_k + m * x_k), m = minkowski_dot(x,y) under the
 * active sign convention (m = -u). */
HMO_EXPORT void hmo_log_map(const float* x, const float* y, int d1, int sign_mode, float* out)
{
    float u = hmo_minkowski_u(x, y, d1, sign_mode);
    float m = -u;
    float a = clamp_min_one(u);
    float coef = hmo_acoshf(a) / sqrtf(a * a - 1.0f);
    int k;
    if (coef == coef && coef > 1.0e4f) coef = 1.0e4f;   /* clamp(max=1e4); NaN stays NaN (:113-117) */
    for (k = 0; k < d1; ++k) out[k] = coef * (y[k] + m * x[k]);
}

/* lorentz_model.py:73-93.  n = sqrt(max(sum_{k>=1} v_k^2, 1e-8)); the (n < 1e-6) mask can never
 * fire because n >= 1e-4, so direction = v / n. */
HMO_EXPORT void hmo_exp_map(const float* x, const float* v, int d1, float* out)
{
    float acc = 0.0f, n, ch, sh;
    int k;
    for (k = 1; k < d1; ++k) acc = fmaf(v[k], v[k], acc);
    if (acc == acc && acc < 1.0e-8f) acc = 1.0e-8f;
    n = sqrtf(acc);
    ch = hmo_coshf(n);
    sh = hmo_sinhf(n);
    for (k = 0; k < d1; ++k) out[k] = ch * x[k] + sh * (v[k] / n);
}

/* lorentz_model.py:41-56.  x0' = sqrt(1 + c*r*r), r = ||x_{1:}||_2; spatial part unchanged. */
HMO_EXPORT void hmo_project(const float* x, int d1, float c, float* out)
{
    float acc = 0.0f, r;
    int k;
    for (k = 1; k < d1; ++k) acc = fmaf(x[k], x[k], acc);
    r = sqrtf(acc);
    out[0] = sqrtf(1.0f + (c * r) * r);
    for (k = 1; k < d1; ++k) out[k] = x[k];
}

/* hyperbolic_merge.py:326-340: project(exp_map(x_i, w * log_map(x_i, x_j))). */
HMO_EXPORT void hmo_midpoint(const float* xi, const float* xj, float w, int d1, float c, int sign_mode,
                             float* out)
{
    float* v = (float*)malloc(sizeof(float) * (size_t)d1 * 2);
    float* e = v + d1;
    int k;
    hmo_log_map(xi, xj, d1, sign_mode, v);
    for (k = 0; k < d1; ++k) v[k] = v[k] * w;
    hmo_exp_map(xi, v, d1, e);
    hmo_project(e, d1, c, out);
    free(v);
}

HMO_EXPORT void hmo_midpoint_batch(const float* X, int64_t ld, int d1, const int32_t* I, const int32_t* J,
                                   const float* W, int64_t b, float c, int sign_mode, float* out)
{
    int64_t t;
    for (t = 0; t < b; ++t)
        hmo_midpoint(X + (int64_t)I[t] * ld, X + (int64_t)J[t] * ld, W[t], d1, c, sign_mode, out + t * d1);
}

/* ------------------------------------------------------------------------------------------ */
/* candidate search (plain form: every pair gets its distance, then the reference's predicate) */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint32_t dbits; int32_t i, j; } hmo_cand;

static int cand_cmp(const void* pa, const void* pb)
{
    const hmo_cand* a = (const hmo_cand*)pa;
    const hmo_cand* b = (const hmo_cand*)pb;
    if (a->dbits != b->dbits) return a->dbits < b->dbits ? -1 : 1;   /* d >= 0: bits are ordered */
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    return 0;
}

/* number of pairs i<j, row_begin <= i < row_end, with d < thr (fp32 compare; NaN never counts) */
HMO_EXPORT int64_t hmo_pairwise_count(const float* X, int64_t n, int64_t ld, int d1, float c, float thr,
                                      int sign_mode, int64_t row_begin, int64_t row_end)
{
    const float sc = sqrtf(c);
    int64_t total = 0, i;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total)
    for (i = row_begin; i < row_end; ++i) {
        int64_t j;
        for (j = i + 1; j < n; ++j) {
            float d = dist_from_u(hmo_minkowski_u(X + i * ld, X + j * ld, d1, sign_mode), sc);
            if (d < thr) ++total;
        }
    }
    return total;
}

/* hyperbolic_merge.py:247-269 / fast_hyperbolic_merge.py:336-355: candidates in row-major order.
 * Writes at most cap triples, returns the total number found. */
HMO_EXPORT int64_t hmo_pairwise_candidates(const float* X, int64_t n, int64_t ld, int d1, float c, float thr,
                                           int sign_mode, int64_t row_begin, int64_t row_end, int64_t cap,
                                           int32_t* out_i, int32_t* out_j, float* out_d)
{
    const float sc = sqrtf(c);
    int64_t total = 0, i, j;
    for (i = row_begin; i < row_end; ++i)
        for (j = i + 1; j < n; ++j) {
            float d = dist_from_u(hmo_minkowski_u(X + i * ld, X + j * ld, d1, sign_mode), sc);
            if (d < thr) {
                if (total < cap) { out_i[total] = (int32_t)i; out_j[total] = (int32_t)j; out_d[total] = d; }
                ++total;
            }
        }
    return total;
}

/* The k smallest candidates in the reference's order: ascending distance, ties by row-major
 * (i, j) (stable sort of the row-major list, hyperbolic_merge.py:378, fast...:371).
 * Returns the number written (min(k, count)); *count = total candidates. */
HMO_EXPORT int64_t hmo_pairwise_topk(const float* X, int64_t n, int64_t ld, int d1, float c, float thr,
                                     int sign_mode, int64_t row_begin, int64_t row_end, int64_t k,
                                     float* out_d, int32_t* out_i, int32_t* out_j, int64_t* count)
{
    const float sc = sqrtf(c);
    int64_t cap = 1 << 16, m = 0, total = 0, i, j, t;
    hmo_cand* buf = (hmo_cand*)malloc(sizeof(hmo_cand) * (size_t)cap);
    for (i = row_begin; i < row_end; ++i)
        for (j = i + 1; j < n; ++j) {
            float d = dist_from_u(hmo_minkowski_u(X + i * ld, X + j * ld, d1, sign_mode), sc);
            if (d < thr) {
                ++total;
                if (m == cap) {
                    /* keep memory bounded: sort, keep the best k, continue */
                    qsort(buf, (size_t)m, sizeof(hmo_cand), cand_cmp);
                    if (m > k) m = k;
                    if (m * 2 > cap) { cap *= 2; buf = (hmo_cand*)realloc(buf, sizeof(hmo_cand) * (size_t)cap); }
                }
                buf[m].dbits = f2u(d); buf[m].i = (int32_t)i; buf[m].j = (int32_t)j; ++m;
            }
        }
    qsort(buf, (size_t)m, sizeof(hmo_cand), cand_cmp);
    if (m > k) m = k;
    for (t = 0; t < m; ++t) { out_d[t] = u2f(buf[t].dbits); out_i[t] = buf[t].i; out_j[t] = buf[t].j; }
    free(buf);
    if (count) *count = total;
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* fast form for the timed CPU baseline (bench.py cpu_baseline leg)                            */
/* ------------------------------------------------------------------------------------------ */
/* Same arithmetic, same results, organised for the host cores: the table is transposed once
 * into K-major order so that the canonical fmaf chain runs in SIMD lanes ACROSS partner rows j
 * (each lane still performs the scalar chain in ascending k), OpenMP over row blocks, and the
 * acosh is evaluated only for pairs whose u is below a conservative bound u_hi (u >= u_hi implies
 * d >= thr with a 1e-5 relative margin, far above the 3e-7 error of the canonical acosh).
 * tests/test_oracle_fast.py checks it against hmo_pairwise_topk. */

#define HMO_JB 256   /* partner-row block processed per inner tile */

static float u_hi_for_threshold(float thr, float c)
{
    /* smallest safe bound: cosh(thr*sqrt(c)*(1+1e-5)) rounded up two ulps, computed in double */
    double a = (double)thr * sqrt((double)c) * (1.0 + 1e-5) + 1e-30;
    double uh = cosh(a);
    float f;
    if (!(uh < 3.0e38)) return INFINITY;
    f = (float)uh;
    if ((double)f < uh) f = nextafterf(f, INFINITY);
    f = nextafterf(f, INFINITY);
    f = nextafterf(f, INFINITY);
    return f;
}

HMO_EXPORT int64_t hmo_fast_pairwise_topk(const float* X, int64_t n, int64_t ld, int d1, float c, float thr,
                                          int sign_mode, int64_t row_begin, int64_t row_end, int64_t k,
                                          float* out_d, int32_t* out_i, int32_t* out_j, int64_t* count)
{
    const float sc = sqrtf(c);
    const float u_hi = u_hi_for_threshold(thr, c);
    const int64_t npad = (n + HMO_JB - 1) / HMO_JB * HMO_JB;
    float* XT = (float*)aligned_alloc(64, sizeof(float) * (size_t)npad * (size_t)d1);
    int nthreads = 1;
    hmo_cand** tbuf; int64_t* tm; int64_t* tcap; int64_t total = 0;
    int64_t i, t, m;
    int kk;
    memset(XT, 0, sizeof(float) * (size_t)npad * (size_t)d1);
    for (i = 0; i < n; ++i)
        for (kk = 0; kk < d1; ++kk) XT[(int64_t)kk * npad + i] = X[i * ld + kk];
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    tbuf = (hmo_cand**)calloc((size_t)nthreads, sizeof(hmo_cand*));
    tm = (int64_t*)calloc((size_t)nthreads, sizeof(int64_t));
    tcap = (int64_t*)calloc((size_t)nthreads, sizeof(int64_t));
#pragma omp parallel reduction(+ : total)
    {
        int tid = 0;
        float ubuf[HMO_JB] __attribute__((aligned(64)));
        int64_t ii;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        tcap[tid] = 1 << 14;
        tbuf[tid] = (hmo_cand*)malloc(sizeof(hmo_cand) * (size_t)tcap[tid]);
#pragma omp for schedule(dynamic, 8)
        for (ii = row_begin; ii < row_end; ++ii) {
            const float* xi = X + ii * ld;
            int64_t j0;
            for (j0 = (ii + 1) / HMO_JB * HMO_JB; j0 < n; j0 += HMO_JB) {
                int jj, k2;
                for (jj = 0; jj < HMO_JB; ++jj) ubuf[jj] = 0.0f;
                for (k2 = 1; k2 < d1; ++k2) {
                    const float xv = xi[k2];
                    const float* col = XT + (int64_t)k2 * npad + j0;
                    for (jj = 0; jj < HMO_JB; ++jj) ubuf[jj] = fmaf(xv, col[jj], ubuf[jj]);
                }
                {
                    const float x0 = xi[0];
                    const float* col = XT + j0;
                    for (jj = 0; jj < HMO_JB; ++jj) {
                        float mm = fmaf(x0, col[jj], -ubuf[jj]);
                        ubuf[jj] = sign_mode ? mm : -mm;
                    }
                }
                for (jj = 0; jj < HMO_JB; ++jj) {
                    int64_t j = j0 + jj;
                    float u = ubuf[jj];
                    if (j > ii && j < n && u < u_hi) {
                        float d = dist_from_u(u, sc);
                        if (d < thr) {
                            ++total;
                            if (tm[tid] == tcap[tid]) {
                                qsort(tbuf[tid], (size_t)tm[tid], sizeof(hmo_cand), cand_cmp);
                                if (tm[tid] > k) tm[tid] = k;
                                if (tm[tid] * 2 > tcap[tid]) {
                                    tcap[tid] *= 2;
                                    tbuf[tid] = (hmo_cand*)realloc(tbuf[tid], sizeof(hmo_cand) * (size_t)tcap[tid]);
                                }
                            }
                            tbuf[tid][tm[tid]].dbits = f2u(d);
                            tbuf[tid][tm[tid]].i = (int32_t)ii;
                            tbuf[tid][tm[tid]].j = (int32_t)j;
                            ++tm[tid];
                        }
                    }
                }
            }
        }
    }
    /* merge the per-thread lists */
    m = 0;
    for (t = 0; t < nthreads; ++t) m += tm[t];
    {
        hmo_cand* all = (hmo_cand*)malloc(sizeof(hmo_cand) * (size_t)(m > 0 ? m : 1));
        int64_t off = 0;
        for (t = 0; t < nthreads; ++t) {
            if (tm[t]) memcpy(all + off, tbuf[t], sizeof(hmo_cand) * (size_t)tm[t]);
            off += tm[t];
            free(tbuf[t]);
        }
        qsort(all, (size_t)m, sizeof(hmo_cand), cand_cmp);
        if (m > k) m = k;
        for (t = 0; t < m; ++t) { out_d[t] = u2f(all[t].dbits); out_i[t] = all[t].i; out_j[t] = all[t].j; }
        free(all);
    }
    free(tbuf); free(tm); free(tcap); free(XT);
    if (count) *count = total;
    return m;
}

HMO_EXPORT int hmo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

HMO_EXPORT void hmo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
