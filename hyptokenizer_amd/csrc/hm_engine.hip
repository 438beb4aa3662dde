// hm_engine.hip -- engine lifetime, table upload and statistics of libhypmerge.so (include/hypmerge.h).
//
// Hot path: the all-pairs Lorentz-distance candidate search of
// HyperbolicTokenizer._find_merge_candidates (tokenizer/hyperbolic_merge.py:247-269) and
// FastHyperbolicTokenizer._find_merge_candidates_fast (tokenizer/fast_hyperbolic_merge.py:336-374),
// plus the log-map / exp-map "midpoint" of _merge_tokens (hyperbolic_merge.py:326-340).
// Kernels live in hm_scan.hip (pair scan), hm_search.hip (exact selection), hm_rows.hip (row-wise work)
// and hm_loops.hip (device-resident merge loops); see hm_common.h for the data layout.
#include "hm_common.h"

static thread_local std::string g_last_error;

int hm_fail(hm_engine* e, int code, const std::string& msg)
{
    g_last_error = msg;
    if (e) e->err = msg;
    return code;
}

static const int kSupportedNG[] = {1, 2, 3, 4, 6, 8, 10, 13, 16, 20, 25, 28, 32};
static const int kSupportedKC[] = {2, 4, 8, 13, 14, 16};   // bf16 form: chunks of 8 K-slots (two per k-step; 13 ends on a half step)

static int hm_pick_kc(int d)
{
    const int need = (d + 4 + 7) / 8;          // d spatial slots + 4 slots for the split time coordinate
    for (int v : kSupportedKC)
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

// ---- work-decomposition knobs (test / tuning hook: hm_debug_set_knob, hm_debug_set_default_knob) ----
#if defined(HM_TUNING)
static const char* const kKnobNames[] = {"chunk", "tail", "tail_div", "big_rows", "shape", "incr_topk", "kc_even", "phases", "ph_share0", "ph_share1", "ph_share2", "ph_share3", "ph_share4",
                                        "ph_div1", "ph_div2", "ph_div3", "ph_div4", "ph_div5", "pipeline", "pipe_fault_at", "pipeline_pairs", "dyn", "dyn_slots"};
#endif
static std::map<std::string, double> g_default_knobs;          // applied to every engine created afterwards

static int hm_apply_knob(hm_engine* e, const char* name, double v)
{
    const std::string k = name ? name : "";
    if (k == "chunk") { if (!(v >= 4 && v <= 4096)) return HM_E_ARG; e->chunk_f32 = e->chunk_bf16 = (int)v; }
    else if (k == "tail") { if (!(v >= 0.0 && v <= 0.9)) return HM_E_ARG; e->tail_fraction = v; }
    else if (k == "tail_div") { if (!(v >= 1 && v <= 16)) return HM_E_ARG; e->tail_div = (int)v; }
    else if (k == "big_rows") { if (!(v >= 0)) return HM_E_ARG; e->big_min_rows = (int64_t)v; }          // 512-row blocks from this many rows' pairs on
    else if (k == "shape") { if (!(v >= -1 && v <= 4)) return HM_E_ARG; e->force_shape = (int)v; }
    else if (k == "incr_topk") e->incremental_topk = v != 0.0;
    else if (k == "phases") { if (!(v >= 1 && v <= HM_SCAN_PHASES)) return HM_E_ARG; e->phases = (int)v; }
    else if (k.size() == 9 && k.compare(0, 8, "ph_share") == 0 && k[8] >= '0' && k[8] < '0' + HM_SCAN_PHASES - 1) { if (!(v >= 0.0 && v <= 1.0)) return HM_E_ARG; e->ph_share[k[8] - '0'] = v; }
    else if (k.size() == 7 && k.compare(0, 6, "ph_div") == 0 && k[6] >= '1' && k[6] < '0' + HM_SCAN_PHASES) { if (!(v >= 1 && v <= 64)) return HM_E_ARG; e->ph_div[k[6] - '0'] = (int)v; }
    else if (k == "dyn_slots") { if (!(v >= 0 && v <= 65536)) return HM_E_ARG; e->dyn_slots = (int)v; }
    else if (k == "dyn") e->dyn_queue = v != 0.0;           // scan: resident grid + in-order item queue instead of one block per item
    else if (k == "pipeline") e->pipeline = v != 0.0;       // standard loop: the step's tail work under the next step's scan (0: strictly sequential)
    else if (k == "pipe_fault_at") e->pipe_fault_at = (int)v;
    else if (k == "exact_search") e->force_exact = v != 0.0;      // every top-k / count through the prefilter-free path (hm_exact.hip): tests
    else if (k == "pipeline_pairs") { if (!(v >= 0)) return HM_E_ARG; e->pipeline_min_pairs = (int64_t)v; }
    else if (k == "kc_even") {            // bf16 image with an even chunk count (whole k-steps only): default knob only, before the images exist
        if (e->img16 != nullptr) return HM_E_STATE;
        if (v != 0.0 && (e->KC & 1)) { e->KC += 1; e->RB16 = 16 * hm_row16_chunks(e->KC); }
    }
    else return HM_E_ARG;
    e->armed = false;
    return HM_OK;
}

extern "C" int hm_debug_set_knob(hm_engine* e, const char* name, double value)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_debug_set_knob: engine is NULL");
    const int rc = hm_apply_knob(e, name, value);
    return rc ? hm_fail(e, rc, "hm_debug_set_knob: unknown knob or value out of range") : HM_OK;
}

extern "C" int hm_debug_set_default_knob(const char* name, double value, int clear)
{
    const std::string k = name ? name : "";
    if (clear) { if (k.empty()) g_default_knobs.clear(); else g_default_knobs.erase(k); return HM_OK; }
    hm_engine probe;
    const int rc = hm_apply_knob(&probe, name, value);
    if (rc) return hm_fail(nullptr, rc, "hm_debug_set_default_knob: unknown knob or value out of range");
    g_default_knobs[k] = value;
    return HM_OK;
}

extern "C" const char* hm_last_error(const hm_engine* e)
{
    if (e && !e->err.empty()) return e->err.c_str();
    return g_last_error.c_str();
}

static int hm_engine_alloc(hm_engine* e)
{
    const int device = e->device;
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
    HM_HIP(hipMalloc(&e->d_prev, sizeof(uint4) * (size_t)e->sorted_cap));
    HM_HIP(hipMalloc(&e->d_batch, sizeof(int32_t) * 3 * HM_BATCH_MAX));
    HM_HIP(hipHostMalloc(&e->h_batch, sizeof(int32_t) * 3 * HM_BATCH_MAX, hipHostMallocDefault));
    HM_HIP(hipEventCreateWithFlags(&e->ev_batch, hipEventDisableTiming));
    HM_HIP(hipMalloc(&e->d_ctr, sizeof(uint32_t) * 8));
    HM_HIP(hipMemset(e->d_ctr, 0, sizeof(uint32_t) * 8));
    HM_HIP(hipMalloc(&e->d_rmax2_mem, sizeof(uint32_t) * 4));
    HM_HIP(hipMemset(e->d_rmax2_mem, 0, sizeof(uint32_t) * 4));
    e->d_rmax2 = e->d_rmax2_mem;
    HM_HIP(hipMalloc(&e->d_ctr64, sizeof(unsigned long long) * 8));            // two sets of 4 (the second: pipelined loop)
    HM_HIP(hipMemset(e->d_ctr64, 0, sizeof(unsigned long long) * 8));
    HM_HIP(hipMalloc(&e->d_queue, 256));
    HM_HIP(hipMemset(e->d_queue, 0, 256));
    HM_HIP(hipMalloc(&e->d_rowkey, sizeof(unsigned long long) * 3));
    HM_HIP(hipMemset(e->d_rowkey, 0xff, sizeof(unsigned long long) * 3));
    HM_HIP(hipMalloc(&e->d_seed, sizeof(ArgminSeed)));
    HM_HIP(hipMemset(e->d_seed, 0, sizeof(ArgminSeed)));
    HM_HIP(hipMalloc(&e->d_rec, 2 * sizeof(ArgminRec)));
    HM_HIP(hipMalloc(&e->d_loop_recs, HM_LOOP_MAX_STEPS * sizeof(ArgminRec)));
    HM_HIP(hipMalloc(&e->d_loop, sizeof(LoopState)));
    HM_HIP(hipMemset(e->d_loop, 0, sizeof(LoopState)));
    HM_HIP(hipMalloc(&e->d_len, sizeof(int32_t) * (size_t)e->max_rows));
    HM_HIP(hipMalloc(&e->d_parts, sizeof(ArgminPart) * HM_PART_SLOTS));
    HM_HIP(hipMalloc(&e->d_hist, sizeof(uint32_t) * HM_DIGIT_BINS));
    HM_HIP(hipHostMalloc(&e->h, sizeof(HostCtl), hipHostMallocDefault));
    HM_HIP(hipHostMalloc(&e->h_sorted, sizeof(uint4) * (size_t)e->sorted_cap, hipHostMallocDefault));
    // timing events without the system-scope fence a default event adds around the scan
    HM_HIP(hipEventCreateWithFlags(&e->ev0, hipEventDisableSystemFence));
    HM_HIP(hipEventCreateWithFlags(&e->ev1, hipEventDisableSystemFence));
    return HM_OK;
}

extern "C" int hm_engine_create(hm_engine** out, int device, int64_t max_rows, int d1, int sign_mode, int prefilter)
{
    if (!out) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: out is NULL");
    *out = nullptr;
    if (d1 < 2 || d1 > 129) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: d1 must be in [2, 129]");
    if (max_rows < 2 || max_rows > 131072)
        return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: max_rows must be in [2, 131072]");
    if (sign_mode != 0 && sign_mode != 1) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: sign_mode must be 0 or 1");
    if (prefilter < HM_PREFILTER_AUTO || prefilter > HM_PREFILTER_BF16)
        return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: prefilter must be HM_PREFILTER_AUTO / _F32 / _BF16");
    int ndev = 0;
    hipError_t st = hipGetDeviceCount(&ndev);
    if (st != hipSuccess || ndev <= 0)
        return hm_fail(nullptr, st != hipSuccess ? (int)st : (int)hipErrorNoDevice,
                       "hm_engine_create: no HIP device available (the merge engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return hm_fail(nullptr, HM_E_ARG, "hm_engine_create: bad device index");
    hm_engine* e = new hm_engine();
    e->device = device;
    e->max_rows = max_rows;
    e->d1 = d1;
    e->d = d1 - 1;
    e->NG = hm_pick_ng(e->d);
    e->RS = hm_row_floats(e->NG);
    e->sign_mode = sign_mode;
    e->rows_alloc = (max_rows + HM_MAX_BLOCK_ROWS - 1) / HM_MAX_BLOCK_ROWS * HM_MAX_BLOCK_ROWS + HM_MAX_BLOCK_ROWS;
    e->KC = hm_pick_kc(e->d);
    if (e->KC < 0) {                     // d > 124: the k-steps instantiated do not hold d + 4 slots -- fp32 form only
        e->KC = 0;                       // (the bf16 image degenerates to its [x0] chunk)
        e->bf16_ok = false;
    }
    e->RB16 = 16 * hm_row16_chunks(e->KC);
    e->precision = prefilter;
    if (const char* pe = getenv("HM_SCAN_PRECISION")) {       // override of the argument: "f32" | "bf16"
        if (!strcmp(pe, "f32")) e->precision = HM_PREFILTER_F32;
        else if (!strcmp(pe, "bf16")) e->precision = HM_PREFILTER_BF16;
    }
    for (const auto& kv : g_default_knobs) (void)hm_apply_knob(e, kv.first.c_str(), kv.second);
#if defined(HM_TUNING)
    // tuning builds (tools/build_variant.sh adds -DHM_TUNING) also take the knobs from the environment, HM_TUNE_<NAME>;
    // the shipped library reads no such variable
    for (const char* name : kKnobNames) {
        std::string var = "HM_TUNE_";
        for (const char* q = name; *q; ++q) var += (char)toupper(*q);
        if (const char* t = getenv(var.c_str())) (void)hm_apply_knob(e, name, atof(t));
    }
#endif
    // emission buffers: every pair of the largest table when that is small, 2^24 entries (256 MiB) at most
    {
        const uint64_t pairs = (uint64_t)max_rows * (uint64_t)(max_rows - 1) / 2;
        uint64_t cap = std::min<uint64_t>(pairs + 1024, 1ull << 24);
        cap = std::max<uint64_t>(cap, 1ull << 17);
        e->ent_cap = (uint32_t)cap;
    }
    e->sorted_cap = 1u << 16;
    const int rc = hm_engine_alloc(e);
    if (rc != HM_OK) {
        const std::string msg = e->err;
        hm_engine_destroy(e);            // frees whatever was allocated before the failure
        return hm_fail(nullptr, rc, msg);
    }
    *out = e;
    return HM_OK;
}

extern "C" int hm_engine_destroy(hm_engine* e)
{
    if (!e) return HM_OK;
    (void)hipSetDevice(e->device);
    (void)hm_comm_destroy(e);
    void* dev_ptrs[] = {e->img, e->ent, e->ent2, e->sorted, e->d_ctr, e->d_ctr64, e->d_rec, e->d_hist, e->d_rmax2_mem, e->d_parts,
                        e->img16, e->d_seed, e->d_loop_recs, e->d_loop, e->d_len, e->d_prev, e->d_batch, e->d_rowkey, e->d_rowcnt, e->d_queue};
    for (void* q : dev_ptrs)
        if (q) (void)hipFree(q);
    if (e->h) (void)hipHostFree(e->h);
    if (e->h_sorted) (void)hipHostFree(e->h_sorted);
    if (e->h_batch) (void)hipHostFree(e->h_batch);
    if (e->h_loop) (void)hipHostFree(e->h_loop);
    if (e->ev_batch) (void)hipEventDestroy(e->ev_batch);
    for (int q = 0; q < 2; ++q) {
        if (e->ev_scan[q]) (void)hipEventDestroy(e->ev_scan[q]);
        if (e->ev_tail[q]) (void)hipEventDestroy(e->ev_tail[q]);
    }
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    for (hipEvent_t ev : e->loop_evs) (void)hipEventDestroy(ev);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return HM_OK;
}

extern "C" int hm_set_prefilter(hm_engine* e, int prefilter)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_set_prefilter: engine is NULL");
    if (prefilter < HM_PREFILTER_AUTO || prefilter > HM_PREFILTER_BF16) return hm_fail(e, HM_E_ARG, "hm_set_prefilter: bad value");
    e->precision = prefilter;
    e->armed = false;                    // the seed's margin belongs to the form that wrote it: start the next search afresh
    e->have_cut = false;
    e->topk_f32_thr = 0.0f; e->topk_exact_thr = 0.0f;
    return HM_OK;
}

extern "C" int64_t hm_rows(const hm_engine* e) { return e ? e->n : -1; }

extern "C" int hm_set_table(hm_engine* e, const float* X_dev, int64_t ld, int64_t n_rows, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_set_table: engine is NULL");
    e->armed = false;
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
    e->topk_f32_thr = 0.0f; e->topk_exact_thr = 0.0f;
    HM_HIP(hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), s));      // new table: no seed
    return HM_OK;
}

extern "C" int hm_update_rows(hm_engine* e, const float* X_dev, int64_t ld, int64_t row_begin, int64_t row_end, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_update_rows: engine is NULL");
    e->armed = false;
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

Bounds hm_bounds(float thr, float c)
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

int64_t hm_pairs_in_range(int64_t n, int64_t r0, int64_t r1)
{
    // sum_{i=r0}^{r1-1} (n - 1 - i)
    const int64_t cnt = r1 - r0;
    return cnt * (n - 1) - (r0 + r1 - 1) * cnt / 2;
}

void hm_flush_pending_timing(hm_engine* e)
{
    if (!e->pending_timing) return;
    float ms = 0.f;
    if (hipEventSynchronize(e->ev1) == hipSuccess && hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) {
        e->last_scan_ms = ms; e->last_pairs = e->pending_pairs; e->last_passes = 1;
        e->tot_scan_ms += ms; e->tot_pairs += e->pending_pairs; e->tot_launches += 1;
    }
    e->pending_timing = false;
}

// the per-scan event pairs of the last complete device batch (hm_debug_time_loops), read on demand
void hm_read_loop_events(hm_engine* e)
{
    const int64_t steps = e->loop_unread_steps;
    if (steps <= 0 || e->loop_evs.empty()) return;
    e->loop_unread_steps = 0;
    e->last_batch_ms = e->last_batch_scan_ms = 0.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->loop_evs[2 * HM_LOOP_MAX_STEPS], e->loop_evs[2 * HM_LOOP_MAX_STEPS + 1]) == hipSuccess) e->last_batch_ms = ms;
    for (int64_t k = 0; k < steps; ++k) {
        if (hipEventElapsedTime(&ms, e->loop_evs[2 * k], e->loop_evs[2 * k + 1]) != hipSuccess) continue;
        e->last_batch_scan_ms += ms;
        e->tot_scan_ms += ms; e->tot_pairs += e->loop_unread_pairs[k]; e->tot_launches += 1;
    }
    e->last_batch_steps = steps;
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
    hm_read_loop_events(e);
    if (scan_ms) *scan_ms = e->tot_scan_ms;
    if (pairs) *pairs = e->tot_pairs;
    if (launches) *launches = e->tot_launches;
    if (reset) { e->tot_scan_ms = 0.0; e->tot_pairs = 0; e->tot_launches = 0; }
    return HM_OK;
}

// test hook: pretend the previous refresh ended on this emission cut (bits of u'); the next top-k search of the
// whole table starts from it and has to notice by itself when it is too tight (tests/test_gpu_engine.py)
extern "C" int hm_debug_force_cut(hm_engine* e, uint32_t cut_bits, int64_t k, float c)
{
    if (!e) return HM_E_ARG;
    e->have_cut = true;
    e->last_cut_bits = cut_bits;
    e->last_cut_k = k;
    e->last_cut_c = c;
    e->debug_cut = true;
    return HM_OK;
}

// Measurement aid (bench.py): with on != 0 the device-resident loops record an event pair around EVERY scan launch of a
// batch and around the batch as a whole; hm_last_loop_timing returns the sums of the last batch.  The extra event
// packets cost a microsecond or two per step, so the figure the bench reports as throughput is taken with this off.
extern "C" int hm_debug_time_loops(hm_engine* e, int on)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_debug_time_loops: engine is NULL");
    HM_HIP(hipSetDevice(e->device));
    if (on && e->loop_evs.empty()) {
        e->loop_evs.resize(2 * HM_LOOP_MAX_STEPS + 2, nullptr);
        for (auto& ev : e->loop_evs) HM_HIP(hipEventCreate(&ev));
    }
    e->time_loops = on != 0;
    return HM_OK;
}

extern "C" int hm_last_loop_timing(hm_engine* e, float* batch_ms, float* scan_ms, int64_t* steps)
{
    if (!e || !batch_ms || !scan_ms || !steps) return HM_E_ARG;
    hm_read_loop_events(e);
    *batch_ms = e->last_batch_ms;
    *scan_ms = e->last_batch_scan_ms;
    *steps = e->last_batch_steps;
    return HM_OK;
}
