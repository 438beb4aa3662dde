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
 * Canonical arithmetic (DESIGN.md "Canonical arithmetic").  The restatement reproduces the fp32
 * bits the reference computes on torch's CPU kernels (verified bit-for-bit against torch in
 * tests/test_canonical_vs_torch.py), and the HIP kernels reproduce the restatement bit-for-bit:
 *   p_k = fl(x_k * y_k), k = 1..d                 (torch materialises the product tensor, :163)
 *   S   = sum of p_k in the order of ATen's SumKernel.cpp inner reduction for float:
 *         8 vector lanes x 4 interleaved accumulators, leftover vectors into accumulator 0,
 *         accumulators combined 0+1+2+3, then the scalar tail, then lanes 0..7 in order
 *         (rows shorter than 8 use the scalar 4-accumulator form); cascade levels only start
 *         at d >= 512 and are outside the supported range (d <= 128)
 *   M   = fl(fl(x0*y0) - S)                        (reference minkowski_dot, :25 / :160-169)
 *   u   = -M (sign_mode 0, reference as shipped)   or  +M (sign_mode 1, "lorentz")
 *   a   = clamp_min(u, 1.0f)  with NaN propagation (1.0 + 1e-8 == 1.0f in fp32, :135)
 *   d   = acosh(a) / sqrtf((float)c); acosh as glibc 2.35 acoshf (what torch.acosh calls):
 *         a <= 2: log1pf(t + sqrtf(2t + t*t)), t = a - 1, log1pf = fdlibm s_log1pf.c;
 *         a  > 2: log(2a - 1/(a + sqrt(a*a - 1))) evaluated in double and rounded once
 *         (glibc uses its table-driven logf there; agreement 87 %, otherwise 1 ulp).
 * expm1 (for cosh / sinh in exp_map) follows fdlibm s_expm1f.c.  Everything is fp32 with no FMA
 * contraction (build with -ffp-contract=off) except the double path above, so gcc on x86-64 and
 * hipcc on gfx950 produce identical bits.
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

/* log(1+x) for x >= 0 (NaN/inf pass through).  fdlibm s_log1pf.c (the form glibc 2.35 ships). */
HMO_EXPORT float hmo_log1pf(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    int32_t hx = (int32_t)f2u(x), hu = 0, k = 1;
    float f = 0.0f, c = 0.0f, hfsq, s, z, R, u;
    if (hx < 0) {                               /* negative / -0 / -NaN: outside the domain used here */
        if (f2u(x) == 0x80000000u) return x;
        return u2f(0x7fc00000u);
    }
    if (hx >= 0x7f800000) return x + x;         /* +inf, NaN */
    if (hx < 0x3ed413d0) {                      /* 1+x < sqrt(2)+ */
        if (hx < 0x38000000) {                  /* x < 2**-15 */
            if (hx < 0x33800000) return x;      /* x < 2**-24 */
            return x - (x * x) * 0.5f;
        }
        k = 0; f = x; hu = 1;
    }
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)f2u(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
            c = c / u;
        } else {
            u = x;
            hu = (int32_t)f2u(u);
            k = (hu >> 23) - 127;
            c = 0.0f;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f4) {                    /* u < sqrt(2) */
            u = u2f((uint32_t)(hu | 0x3f800000));
        } else {
            k += 1;
            u = u2f((uint32_t)(hu | 0x3f000000));
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = (0.5f * f) * f;
    if (hu == 0) {                              /* |f| < 2**-20 */
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c = c + (float)k * ln2_lo;
            return (float)k * ln2_hi + c;
        }
        R = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R;
        return (float)k * ln2_hi - ((R - ((float)k * ln2_lo + c)) - f);
    }
    s = f / (2.0f + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return (float)k * ln2_hi - ((hfsq - (s * (hfsq + R) + ((float)k * ln2_lo + c))) - f);
}

/* log(1+x) in double, x >= 0: fdlibm s_log1p.c.  Used only for acosh(a), a > 2. */
static double hmo_log1p_d(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ui; uint32_t hx, hu; int k = 1;
    double hfsq, f = 0.0, c = 0.0, s, z, R, w, t1, t2, dk, uf;
    memcpy(&ui, &x, 8);
    hx = (uint32_t)(ui >> 32);
    if (hx >= 0x7ff00000u) return x;            /* inf / NaN (negative inputs do not occur) */
    if (hx < 0x3fda827au) {                     /* 1+x < sqrt(2)+ */
        if ((hx << 1) < (0x3ca00000u << 1)) return x;   /* |x| < 2**-53 */
        k = 0; c = 0.0; f = x;
    }
    if (k) {
        uf = 1.0 + x;
        memcpy(&ui, &uf, 8);
        hu = (uint32_t)(ui >> 32);
        hu += 0x3ff00000u - 0x3fe6a09eu;
        k = (int)(hu >> 20) - 0x3ff;
        if (k < 54) { c = (k >= 2) ? 1.0 - (uf - x) : x - (uf - 1.0); c = c / uf; } else c = 0.0;
        hu = (hu & 0x000fffffu) + 0x3fe6a09eu;
        ui = ((uint64_t)hu << 32) | (ui & 0xffffffffull);
        memcpy(&uf, &ui, 8);
        f = uf - 1.0;
    }
    hfsq = (0.5 * f) * f;
    s = f / (2.0 + f);
    z = s * s;
    w = z * z;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    R = t2 + t1;
    dk = (double)k;
    return ((s * (hfsq + R) + (dk * ln2_lo + c)) - hfsq + f) + dk * ln2_hi;
}

/* acosh(a) for a >= 1 (NaN passes through), structured as glibc's e_acoshf.c. */
HMO_EXPORT float hmo_acoshf(float a)
{
    if (a != a) return a;
    if (a <= 1.0f) return 0.0f;
    if (a <= 2.0f) {
        float t = a - 1.0f;
        return hmo_log1pf(t + sqrtf(2.0f * t + t * t));
    }
    if (a > 3.0e38f) return a;                  /* +inf */
    {
        double x = (double)a;
        double zz = 2.0 * x - 1.0 / (x + sqrt(x * x - 1.0));
        return (float)hmo_log1p_d(zz - 1.0);
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

/* Sum of n fp32 terms in the order of ATen's SumKernel.cpp (cascade_sum, float, inner
 * reduction over a contiguous row; torch 2.x CPU, 8-float vectors):
 *   n >= 8: vectorized_inner_sum -> row_sum<Vec8> (4 interleaved vector accumulators over groups
 *           of 4 vectors, leftover vectors into accumulator 0, then 0 += 1, += 2, += 3),
 *           scalar tail summed from 0, then the 8 lanes added in order;
 *   n <  8: scalar row_sum (4 interleaved scalar accumulators, leftovers into 0, then combined).
 * The multi-level cascade only engages at >= 16 groups (n >= 512) and is not needed here.
 * `term(ctx, k)` returns the k-th term (already rounded to fp32). */
typedef float (*hmo_term_fn)(const void* ctx, int k);

static float torch_order_sum(hmo_term_fn term, const void* ctx, int n)
{
    int i, k, l;
    if (n < 8) {
        float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int size_ilp = n / 4;
        for (i = 0; i < size_ilp; ++i)
            for (k = 0; k < 4; ++k) ps[k] = ps[k] + term(ctx, i * 4 + k);
        for (i = size_ilp * 4; i < n; ++i) ps[0] = ps[0] + term(ctx, i);
        for (k = 1; k < 4; ++k) ps[0] = ps[0] + ps[k];
        return ps[0];
    }
    {
        float ps[4][8];
        const int vec_size = n / 8, size_ilp = vec_size / 4;
        float acc = 0.0f;
        for (k = 0; k < 4; ++k) for (l = 0; l < 8; ++l) ps[k][l] = 0.0f;
        for (i = 0; i < size_ilp; ++i)
            for (k = 0; k < 4; ++k)
                for (l = 0; l < 8; ++l) ps[k][l] = ps[k][l] + term(ctx, (i * 4 + k) * 8 + l);
        for (i = size_ilp * 4; i < vec_size; ++i)
            for (l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + term(ctx, i * 8 + l);
        for (k = 1; k < 4; ++k)
            for (l = 0; l < 8; ++l) ps[0][l] = ps[0][l] + ps[k][l];
        for (i = vec_size * 8; i < n; ++i) acc = acc + term(ctx, i);
        for (l = 0; l < 8; ++l) acc = acc + ps[0][l];
        return acc;
    }
}

typedef struct { const float* x; const float* y; } hmo_pair_ctx;
static float term_prod(const void* ctx, int k)
{
    const hmo_pair_ctx* c = (const hmo_pair_ctx*)ctx;
    return c->x[1 + k] * c->y[1 + k];           /* spatial coordinates start at column 1 */
}

/* lorentz_model.py:14-25 with the sign switch of SURVEY F5.
 * sign_mode 0: u = -(x0*y0 - sum)  (what distance/log_map feed to acosh as shipped)
 * sign_mode 1: u = +(x0*y0 - sum)  (standard Lorentz form) */
HMO_EXPORT float hmo_minkowski_u(const float* x, const float* y, int d1, int sign_mode)
{
    hmo_pair_ctx c;
    float S, t, m;
    c.x = x; c.y = y;
    S = torch_order_sum(term_prod, &c, d1 - 1);
    t = x[0] * y[0];
    m = t - S;                                  /* reference minkowski_dot */
    return sign_mode ? m : -m;
}

/* the fast (prefilter) form used by the timed baseline: plain fmaf chain, time coordinate last */
static inline float fast_u(const float* x, const float* y, int d1, int sign_mode)
{
    float acc = 0.0f, m;
    int k;
    for (k = 1; k < d1; ++k) acc = fmaf(x[k], y[k], acc);
    m = fmaf(x[0], y[0], -acc);
    return sign_mode ? m : -m;
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
    float acc, n, ch, sh;
    int k;
    hmo_pair_ctx c;
    c.x = v; c.y = v;
    acc = torch_order_sum(term_prod, &c, d1 - 1);      /* torch.sum(v[1:] * v[1:]) */
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
/* Same results, organised for the host cores in two stages (the GPU engine has the same shape):
 * (1) a fast prefilter: the table is transposed once into K-major order and a plain fmaf chain
 *     runs in SIMD lanes ACROSS partner rows j (OpenMP over row blocks); it differs from the
 *     canonical u by at most delta = (d+5) * 2^-23 * max_row ||x||^2 (both are roundings of the
 *     same exact form; standard gamma_n bound on each);
 * (2) pairs whose fast u is below u_hi + delta (u >= u_hi implies d >= thr with a 1e-5 relative
 *     margin, far above the few-ulp error of acosh) are re-evaluated with the canonical
 *     arithmetic and tested exactly.
 * tests/test_oracle_golden.py::test_fast_oracle_equals_plain checks it against hmo_pairwise_topk. */

#define HMO_JB 256   /* partner-row block processed per inner tile */
#define HMO_IB 4     /* stationary rows sharing each partner-column load */

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
    float u_pre;
    const int64_t npad = (n + HMO_JB - 1) / HMO_JB * HMO_JB;
    float* XT = (float*)aligned_alloc(64, sizeof(float) * (size_t)npad * (size_t)d1);
    int nthreads = 1;
    hmo_cand** tbuf; int64_t* tm; int64_t* tcap; int64_t total = 0;
    int64_t i, t, m;
    int kk;
    memset(XT, 0, sizeof(float) * (size_t)npad * (size_t)d1);
    {
        double r2max = 0.0;
        for (i = 0; i < n; ++i) {
            double r2 = 0.0;
            for (kk = 0; kk < d1; ++kk) {
                XT[(int64_t)kk * npad + i] = X[i * ld + kk];
                r2 += (double)X[i * ld + kk] * (double)X[i * ld + kk];
            }
            if (r2 == r2 && r2 < 1e300 && r2 > r2max) r2max = r2;   /* NaN / inf rows never qualify */
        }
        u_pre = u_hi + (float)((double)(d1 + 4) * 1.1920929e-07 * r2max * 1.0001);
    }
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    tbuf = (hmo_cand**)calloc((size_t)nthreads, sizeof(hmo_cand*));
    tm = (int64_t*)calloc((size_t)nthreads, sizeof(int64_t));
    tcap = (int64_t*)calloc((size_t)nthreads, sizeof(int64_t));
#pragma omp parallel reduction(+ : total)
    {
        int tid = 0;
        float ubuf[HMO_IB][HMO_JB] __attribute__((aligned(64)));
        int64_t ib;
        const int64_t nblk = (row_end - row_begin + HMO_IB - 1) / HMO_IB;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        tcap[tid] = 1 << 14;
        tbuf[tid] = (hmo_cand*)malloc(sizeof(hmo_cand) * (size_t)tcap[tid]);
#pragma omp for schedule(dynamic, 4)
        for (ib = 0; ib < nblk; ++ib) {
            const int64_t i0 = row_begin + ib * HMO_IB;
            const int nrow = (int)((row_end - i0) < HMO_IB ? (row_end - i0) : HMO_IB);
            const float* xr[HMO_IB];
            int64_t j0;
            int q;
            for (q = 0; q < HMO_IB; ++q) xr[q] = X + (i0 + (q < nrow ? q : 0)) * ld;
            for (j0 = (i0 + 1) / HMO_JB * HMO_JB; j0 < n; j0 += HMO_JB) {
                int jj, k2;
                for (q = 0; q < HMO_IB; ++q)
                    for (jj = 0; jj < HMO_JB; ++jj) ubuf[q][jj] = 0.0f;
                for (k2 = 1; k2 < d1; ++k2) {
                    const float* col = XT + (int64_t)k2 * npad + j0;
                    const float x0v = xr[0][k2], x1v = xr[1][k2], x2v = xr[2][k2], x3v = xr[3][k2];
                    for (jj = 0; jj < HMO_JB; ++jj) {
                        const float cv = col[jj];
                        ubuf[0][jj] = fmaf(x0v, cv, ubuf[0][jj]);
                        ubuf[1][jj] = fmaf(x1v, cv, ubuf[1][jj]);
                        ubuf[2][jj] = fmaf(x2v, cv, ubuf[2][jj]);
                        ubuf[3][jj] = fmaf(x3v, cv, ubuf[3][jj]);
                    }
                }
                for (q = 0; q < nrow; ++q) {
                    const int64_t ii = i0 + q;
                    const float* xi = xr[q];
                    const float x0 = xi[0];
                    const float* col = XT + j0;
                    for (jj = 0; jj < HMO_JB; ++jj) {
                        const int64_t j = j0 + jj;
                        const float mm = fmaf(x0, col[jj], -ubuf[q][jj]);
                        const float u = sign_mode ? mm : -mm;
                        if (j > ii && j < n && u < u_pre) {
                            float d = dist_from_u(hmo_minkowski_u(xi, X + j * ld, d1, sign_mode), sc);
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

/* ------------------------------------------------------------------------------------------ */
/* enhanced tokenizer (BASELINE config 5)                                                      */
/* ------------------------------------------------------------------------------------------ */

/* tokenizer/enhanced_fast_hyperbolic_merge.py:308-333 (_compute_semantic_coherence), the device part:
 * for candidate t the simulated merged embedding m = exp_map(x_i, w * log_map(x_i, x_j)) -- NOT
 * projected (:319-321) -- and the distances distance(m, x_s) to its ns sampled rows S[t*ns ..].
 * The sampling (torch.randperm), the skip of s in {i, j} and the mean / sigmoid stay with the caller. */
HMO_EXPORT void hmo_coherence_distances(const float* X, int64_t ld, int d1, const int32_t* I, const int32_t* J,
                                        const float* W, const int32_t* S, int64_t b, int ns, float c, int sign_mode,
                                        float* out)
{
    int64_t t;
#pragma omp parallel for schedule(static)
    for (t = 0; t < b; ++t) {
        float* v = (float*)malloc(sizeof(float) * (size_t)d1 * 2);
        float* m = v + d1;
        const float* xi = X + (int64_t)I[t] * ld;
        int k, s;
        hmo_log_map(xi, X + (int64_t)J[t] * ld, d1, sign_mode, v);
        for (k = 0; k < d1; ++k) v[k] = v[k] * W[t];
        hmo_exp_map(xi, v, d1, m);
        for (s = 0; s < ns; ++s)
            out[t * ns + s] = hmo_distance(m, X + (int64_t)S[t * ns + s] * ld, d1, c, sign_mode);
        free(v);
    }
}

/* enhanced...:784-792 (_project_embeddings) and :243-244: project_to_hyperboloid over rows [0, n) of
 * the table, in place (only column 0 changes). */
HMO_EXPORT void hmo_project_table(float* X, int64_t ld, int d1, int64_t n, float c)
{
    int64_t t;
#pragma omp parallel for schedule(static)
    for (t = 0; t < n; ++t) {
        float* x = X + t * ld;
        float acc = 0.0f, r;
        int k;
        for (k = 1; k < d1; ++k) acc = fmaf(x[k], x[k], acc);
        r = sqrtf(acc);
        x[0] = sqrtf(1.0f + (c * r) * r);
    }
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
