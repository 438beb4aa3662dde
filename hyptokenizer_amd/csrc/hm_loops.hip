// hm_loops.hip -- device-resident merge loops: several steps of HyperbolicTokenizer.optimize_merges
// (tokenizer/hyperbolic_merge.py:357-412) per host call.
//
// What the host needs from a step is the pair (i, j) -- for the token strings -- and nothing else: the merge weight
// len(tj) / (len(ti) + len(tj)) (hyperbolic_merge.py:317-323) only needs token LENGTHS, which live in a device array
// here.  So K steps are enqueued back to back -- [pair scan -> tail kernel (exact re-evaluation, record, seed,
// arming of the next scan, the merge itself)] x K, two launches per step, no host round trip in between -- and the
// host reads the K records with ONE synchronisation.  A step that finds nothing (or overflows the emission buffer)
// raises a stop word in HBM that turns every later launch of the batch into a no-op.
//
// The incremental form (rows are only ever appended, so the nearest pair is a running minimum) is ONE
// launch per step: every block recomputes the merged row (cheaper than a grid-wide hand-off), block 0 stores it,
// all blocks scan their slice of the image against it, and the last block to finish folds the new row's nearest
// partner into the running minimum.
#include "hm_common.h"
#include "hm_rows_device.h"

#pragma clang fp contract(off)

extern "C" int hm_set_token_lengths(hm_engine* e, const int32_t* lens_host, int64_t n, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_set_token_lengths: engine is NULL");
    if (!lens_host || n < 0 || n > e->max_rows) return hm_fail(e, HM_E_ARG, "hm_set_token_lengths: bad arguments");
    HM_HIP(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (n) HM_HIP(hipMemcpyAsync(e->d_len, lens_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    HM_HIP(hipStreamSynchronize(s));          // the caller's buffer is pageable: done with it on return
    e->have_len = true;
    return HM_OK;
}

// forget rows >= n_rows (a caller that appended rows ahead of time and changed its mind)
extern "C" int hm_truncate(hm_engine* e, int64_t n_rows, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_truncate: engine is NULL");
    if (n_rows < 0 || n_rows > e->n) return hm_fail(e, HM_E_ARG, "hm_truncate: n_rows outside [0, live rows]");
    HM_HIP(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (e->n > n_rows) {
        HM_HIP(hipMemsetAsync(e->img + n_rows * e->RS, 0, sizeof(float) * (size_t)(e->n - n_rows) * e->RS, s));
        HM_HIP(hipMemsetAsync(e->img16 + n_rows * e->RB16, 0, (size_t)(e->n - n_rows) * e->RB16, s));
        e->armed = false;
        e->have_cut = false;
        HM_HIP(hipMemsetAsync(e->d_seed, 0, sizeof(ArgminSeed), s));
    }
    e->n = n_rows;
    return HM_OK;
}

static void hm_unpack_recs(const ArgminRec* recs, int64_t steps, uint32_t* rec_out, int64_t* done)
{
    int64_t ok = 0;
    bool counting = true;
    for (int64_t k = 0; k < steps; ++k) {
        rec_out[4 * k + 0] = recs[k].found; rec_out[4 * k + 1] = recs[k].dbits;
        rec_out[4 * k + 2] = recs[k].i; rec_out[4 * k + 3] = recs[k].j;
        if (counting && recs[k].found == 1u) ++ok; else counting = false;
    }
    *done = ok;
}

// ------------------------------------------------------------------------------------------------
// software-pipelined standard loop
// ------------------------------------------------------------------------------------------------
// A step's tail work -- exact re-evaluation, record, merge (one wave of latency-bound transcendental arithmetic: 12-15 us
// with its launch boundaries) -- used to sit between two scans.  Only the pairs of the NEWEST row depend on it.  So step k
// is split: scan_old(k) covers the pairs among the rows that existed two merges ago (all but the newest row) and follows
// scan_old(k - 1) back to back on the caller's stream; the newest row's nearest partner is found by a small row pass and
// folded in by tail(k), both on the engine's second stream -- under scan_old(k + 1).  Dependencies (events):
//   scan_old(k) after tail(k - 2);  rowpass(k) after tail(k - 1) (stream order);  tail(k) after scan_old(k) and rowpass(k).
// Two buffer sets (emission entries, counters, row key) alternate between the steps; tail(k) re-arms its set for step
// k + 2 with the seed of step k (the pair it merged from still exists: a valid bound).  Every pair is still evaluated in
// every step; results are bit-identical to the sequential chain (tests run both).

// nearest partner of image row `row` among rows [0, n_partners) other than itself: (bits(d) << 32) | i folded into *key by
// atomicMin (d < thr).  Row tiles: a wave copies 64 consecutive image rows to LDS, then one lane per row (hm_rows_device.h).
__global__ __launch_bounds__(64) void hm_newrow_key_kernel(const float* __restrict__ img, int RS, int d, int64_t row, int64_t n_partners,
                                                           float sqrt_c, float thr, int sign_mode, unsigned long long* __restrict__ key,
                                                           const uint32_t* __restrict__ stop)
{
    extern __shared__ __align__(16) float lds[];
    if (stop != nullptr && *stop != 0u) return;
    __builtin_amdgcn_s_setprio(3);          // (runs beside the scan's MFMA waves)
    float* xs = lds;
    const int lane = threadIdx.x;
    float* tile = lds + HM_MAX_D1 + 4;
    const int64_t nt = (n_partners + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
    int64_t tl = blockIdx.x;
    TileRegs tr;
    if (tl < nt) hm_tile_load(img, RS, tl * HM_TILE_ROWS, n_partners, tr, lane);
    for (int k = lane; k < RS; k += 64) xs[k] = img[row * RS + k];
    hm_wave_lds_sync();
    unsigned long long best = ~0ull;
    for (; tl < nt; tl += gridDim.x) {
        hm_tile_store(tile, RS, tr, lane);
        hm_wave_lds_sync();
        const int64_t nxt = tl + gridDim.x;
        if (nxt < nt) hm_tile_load(img, RS, nxt * HM_TILE_ROWS, n_partners, tr, lane);
        const float u = hm_tile_u(tile, RS, d, xs, sign_mode, lane);
        const int64_t i = tl * HM_TILE_ROWS + lane;
        const float dd = hm::dist_from_u(u, sqrt_c);
        if (i < n_partners && i != row && dd < thr) {
            const unsigned long long k64 = ((unsigned long long)hm::fbits(dd) << 32) | (unsigned long long)(uint32_t)i;
            best = k64 < best ? k64 : best;
        }
        hm_wave_lds_sync();
    }
    best = hm_wave_min_u64(best);
    if (lane == 0 && best != ~0ull) atomicMin(key, best);
}

// dynamic LDS of hm_newrow_key_kernel (the fixed row + one tile), with the kernel's limit raised once per engine
static int hm_row_key_lds(hm_engine* e, size_t* bytes)
{
    const void* kfn = reinterpret_cast<const void*>(&hm_newrow_key_kernel);
    *bytes = sizeof(float) * ((size_t)HM_MAX_D1 + 4 + (size_t)HM_TILE_ROWS * e->RS);
    if (e->attr_done.find(kfn) == e->attr_done.end()) {
        HM_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(float) * ((size_t)HM_MAX_D1 + 4 + (size_t)HM_TILE_ROWS * 4 * HM_TILE_MAXQ))));
        e->attr_done.insert(kfn);
    }
    return HM_OK;
}

// hm_row_argmin's device part: *key_dev (pre-set to all ones) <- min over rows i in [0, n_partners), i != row, d(row, i) < thr of
// (bits(d) << 32) | i.  Ordering by (bits, i) IS the engine's (bits, min(i,row), max(i,row)) order: partners below `row` sort
// before partners above it (their smaller index is i < row) and both groups sort by i.
int hm_launch_row_key(hm_engine* e, int64_t row, int64_t n_partners, float sqrt_c, float thr, unsigned long long* key_dev, hipStream_t s)
{
    size_t row_lds = 0;
    int rc = hm_row_key_lds(e, &row_lds);
    if (rc) return rc;
    const int64_t nt = (n_partners + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
    hipLaunchKernelGGL(hm_newrow_key_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(nt, 1024))), dim3(64), row_lds, s, e->img, e->RS,
                       e->d, row, n_partners, sqrt_c, thr, e->sign_mode, key_dev, (const uint32_t*)nullptr);
    HM_HIP(hipGetLastError());
    return HM_OK;
}

static int hm_pipeline_init(hm_engine* e)
{
    if (e->aux) return HM_OK;
    int lo = 0, hi = 0;
    HM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));              // (hi = the numerically lowest = highest priority)
    HM_HIP(hipStreamCreateWithPriority(&e->aux, hipStreamNonBlocking, hi));
    for (int q = 0; q < 2; ++q) {
        HM_HIP(hipEventCreateWithFlags(&e->ev_scan[q], hipEventDisableSystemFence));      // (recorded by the scan's dispatch, like the timing events)
        HM_HIP(hipEventCreateWithFlags(&e->ev_tail[q], hipEventDisableTiming));
    }
    HM_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    return HM_OK;
}

static int hm_std_merge_steps_pipelined(hm_engine* e, float c, float thr, const Bounds& b, float* X_dev, int64_t ld, int64_t steps, hipStream_t s,
                                        int64_t* all_pairs, bool time_all, int64_t* timed_pairs)
{
    int rc = hm_pipeline_init(e);
    if (rc) return rc;
    const int64_t n0 = e->n;
    const float sqrt_c = sqrtf(c);
    hipStream_t sb = e->aux;
    // the tail stream starts behind everything the caller's stream holds so far (table, token lengths, loop state)
    HM_HIP(hipMemsetAsync(e->d_rowkey, 0xff, sizeof(unsigned long long) * 2, s));
    HM_HIP(hipEventRecord(e->ev_join, s));
    HM_HIP(hipStreamWaitEvent(sb, e->ev_join, 0));
    size_t row_lds = 0;
    rc = hm_row_key_lds(e, &row_lds);
    if (rc) return rc;
    for (int64_t k = 0; k < steps; ++k) {
        const int set = (int)(k & 1);
        e->n = n0 + k;                                   // rows of step k's table (optimistic: corrected by the caller when the batch stops early)
        const int64_t n_old = k == 0 ? e->n : e->n - 1;  // scan_old(k): all rows but the newest (step 0: the whole table)
        ScanArgs a; dim3 grid;
        if (!hm_prepare_scan(e, b, 0, -1, a, grid, n_old)) return hm_fail(e, HM_E_STATE, "hm_std_merge_steps: empty scan");
        a.stop = &e->d_loop->stop;
        a.ent = set ? e->ent2 : e->ent;
        a.ctr64 = e->d_ctr64 + 4 * set;
        // ---- caller's stream: scan_old(k) ----
        if (k == 0) {
            // both sets armed from the engine's seed before any tail runs (later steps: their set is armed by tail(k - 2))
            ScanArgs a1 = a;
            a1.ctr64 = e->d_ctr64 + 4;
            rc = hm_launch_seed_init(e, a, s);
            if (rc == HM_OK) rc = hm_launch_seed_init(e, a1, s);
            if (rc) return rc;
        } else if (k >= 2) {
            // tail(k - 2) wrote this scan's newest row and armed its counters a whole scan ago, on the other stream: checked
            // by the scan's blocks themselves (ScanArgs::order_*), not waited for with an event on this stream
            a.order_seen = &e->d_loop->tails_done;
            a.order_need = (uint32_t)(k - 1) + (e->pipe_fault_at == (int)k ? 1000000u : 0u);
            a.order_fault = &e->d_loop->stop;
        }
        // The scan's completion event rides IN its dispatch (the kernel's own completion signal: no extra packet on this
        // stream, where an hipEventRecord costs a 6 us bubble): the tail stream waits for that event.
        const bool timed = (k == steps - 1);
        hipEvent_t ev_start = nullptr, ev_stop = e->ev_scan[set];
        if (time_all) {
            if (k == 0) HM_HIP(hipEventRecord(e->loop_evs[2 * HM_LOOP_MAX_STEPS], s));
            ev_start = e->loop_evs[2 * k]; ev_stop = e->loop_evs[2 * k + 1];
            all_pairs[k] = hm_pairs_in_range(e->n, 0, e->n - 1);
        } else if (timed) {
            ev_start = e->ev0; ev_stop = e->ev1;
        }
        HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, ev_start, ev_stop));
        if (timed) *timed_pairs = hm_pairs_in_range(e->n, 0, e->n - 1);
        // ---- tail stream: rowpass(k) (behind tail(k - 1)), then tail(k) behind scan_old(k) ----
        if (k >= 1) {
            const int64_t row = e->n - 1;
            const int64_t nt = (row + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
            hipLaunchKernelGGL(hm_newrow_key_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(nt, 128))), dim3(64), row_lds, sb, e->img, e->RS,
                               e->d, row, row, sqrt_c, thr, e->sign_mode, e->d_rowkey + set, &e->d_loop->stop);
            HM_HIP(hipGetLastError());
        }
        HM_HIP(hipStreamWaitEvent(sb, ev_stop, 0));
        MergeFuse mf;
        memset(&mf, 0, sizeof(mf));
        mf.X = X_dev; mf.ld = ld; mf.new_row = e->n; mf.c = c;
        mf.len = e->d_len; mf.len_rw = e->d_len; mf.loop = e->d_loop; mf.rec_ring = e->d_loop_recs + k;
        mf.pipe_ent = a.ent; mf.pipe_ctr64 = a.ctr64;
        mf.rowkey = e->d_rowkey + set; mf.rowkey_j = (uint32_t)(e->n - 1);
        rc = hm_launch_argmin_tail(e, a, sqrt_c, thr, e->d_rec, true, 0, 0x7fffffff, true, mf, sb);
        if (rc) return rc;
        HM_HIP(hipEventRecord(e->ev_tail[set], sb));
    }
    e->n = n0 + steps;
    // the caller's stream continues behind the last tails
    HM_HIP(hipStreamWaitEvent(s, e->ev_tail[(steps - 1) & 1], 0));
    if (steps >= 2) HM_HIP(hipStreamWaitEvent(s, e->ev_tail[(steps - 2) & 1], 0));
    return HM_OK;
}

// K steps of the standard loop.  rec_out: steps x {found, bits(d), i, j}; *done = leading steps that merged.
// A record with found = 0 ends the loop (no candidate), found = 2 asks the caller to run that step through
// hm_pairwise_argmin + hm_merge_append (emission overflow), found = 3 marks steps skipped after either.
extern "C" int hm_std_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* rec_out,
                                  int64_t* done, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_std_merge_steps: engine is NULL");
    if (!X_dev || ld < e->d1 || !rec_out || !done || steps < 0 || steps > HM_LOOP_MAX_STEPS || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_std_merge_steps: bad arguments");
    if (!e->have_len) return hm_fail(e, HM_E_STATE, "hm_std_merge_steps: token lengths not set (hm_set_token_lengths)");
    *done = 0;
    if (steps == 0) return HM_OK;
    if (e->n + steps > e->max_rows) return hm_fail(e, HM_E_CAPACITY, "hm_std_merge_steps: the table cannot take that many rows");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    hm_flush_pending_timing(e);
    const Bounds b = hm_bounds(thr, c);
    if (b.none || e->n < 2) {                 // nothing can be below the threshold
        for (int64_t k = 0; k < steps; ++k) { rec_out[4 * k] = k == 0 ? 0u : 3u; rec_out[4 * k + 1] = 0; rec_out[4 * k + 2] = rec_out[4 * k + 3] = 0xffffffffu; }
        return HM_OK;
    }
    const int64_t n0 = e->n;
    const float sqrt_c = sqrtf(c);
    HM_HIP(hipMemsetAsync(e->d_loop, 0, sizeof(LoopState), s));
    bool armed = e->armed && e->armed_rb == 0 && e->armed_re == -1;
    e->armed = false;
    int64_t timed_pairs = 0;
    const bool time_all = e->time_loops && !e->loop_evs.empty();
    if (time_all) hm_read_loop_events(e);         // the events are about to be reused
    int64_t all_pairs[HM_LOOP_MAX_STEPS];
    // pipelined when a scan is long against a tail kernel (the tail of step k - 2 has to be through while scan k - 1 runs; the
    // scans' order guard catches the rest): from ~40 000 rows on
    const bool piped = e->pipeline && steps >= 2 && hm_pairs_in_range(e->n, 0, e->n - 1) >= e->pipeline_min_pairs;
    if (piped) {
        const int rcp = hm_std_merge_steps_pipelined(e, c, thr, b, X_dev, ld, steps, s, all_pairs, time_all, &timed_pairs);
        if (rcp) { (void)hipStreamSynchronize(s); if (e->aux) (void)hipStreamSynchronize(e->aux); e->n = n0; return rcp; }
    }
    for (int64_t k = 0; k < (piped ? 0 : steps); ++k) {
        ScanArgs a; dim3 grid;
        if (!hm_prepare_scan(e, b, 0, -1, a, grid)) return hm_fail(e, HM_E_STATE, "hm_std_merge_steps: empty scan");
        a.stop = &e->d_loop->stop;
        if (!armed) {
            int rc0 = hm_launch_seed_init(e, a, s);
            if (rc0) { e->n = n0; return rc0; }
        }
        // the last scan of the batch carries the timing events (one event pair per engine); in the measurement mode of
        // hm_debug_time_loops every scan has its own pair
        const bool timed = (k == steps - 1);
        if (time_all) {
            if (k == 0) HM_HIP(hipEventRecord(e->loop_evs[2 * HM_LOOP_MAX_STEPS], s));
            HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, e->loop_evs[2 * k], e->loop_evs[2 * k + 1]));
            all_pairs[k] = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
        } else {
            HM_HIP(hm_launch_scan(e, HM_MODE_ARGMIN, a, grid, s, timed ? e->ev0 : nullptr, timed ? e->ev1 : nullptr));
        }
        if (timed) timed_pairs = hm_pairs_in_range(e->n, a.row_begin, a.row_end);
        MergeFuse mf;
        memset(&mf, 0, sizeof(mf));
        mf.X = X_dev; mf.ld = ld; mf.new_row = e->n; mf.c = c;
        mf.len = e->d_len; mf.len_rw = e->d_len; mf.loop = e->d_loop; mf.rec_ring = e->d_loop_recs + k;
        int rc = hm_launch_argmin_tail(e, a, sqrt_c, thr, e->d_rec, true, 0, 0x7fffffff, true, mf, s);
        if (rc) { e->n = n0; return rc; }
        armed = true;
        e->n += 1;                            // optimistic: corrected below when the batch stopped early
    }
    if (time_all) HM_HIP(hipEventRecord(e->loop_evs[2 * HM_LOOP_MAX_STEPS + 1], s));
    HM_HIP(hipMemcpyAsync(e->h->loop_recs, e->d_loop_recs, sizeof(ArgminRec) * (size_t)steps, hipMemcpyDeviceToHost, s));
    if (piped) HM_HIP(hipMemcpyAsync(&e->h->ctr[7], &e->d_loop->stop, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    hm_unpack_recs(e->h->loop_recs, steps, rec_out, done);
    e->n = n0 + *done;
    if (piped && *done < steps && e->h->ctr[7] == 5u) {
        // a scan's order guard tripped (never seen outside the test hook): the steps from there on, strictly sequentially
        e->armed = false;
        e->pipe_fault_at = -1;
        const bool keep = e->pipeline;
        e->pipeline = false;
        int64_t done2 = 0;
        const int rc2 = hm_std_merge_steps(e, c, thr, X_dev, ld, steps - *done, rec_out + 4 * *done, &done2, stream);
        e->pipeline = keep;
        if (rc2) return rc2;
        *done += done2;
        return HM_OK;
    }
    if (piped && *done < steps && e->h->ctr[7] == 6u) {
        // a step left more survivors than the pipelined tail's small grid takes (typically the first search on a new table,
        // before a seed bounds the emissions): that ONE step through the sequential chain and its full-size tail, the rest
        // pipelined again
        e->armed = false;
        const bool keep = e->pipeline;
        e->pipeline = false;
        int64_t done2 = 0;
        int rc2 = hm_std_merge_steps(e, c, thr, X_dev, ld, 1, rec_out + 4 * *done, &done2, stream);
        e->pipeline = keep;
        if (rc2) return rc2;
        *done += done2;
        if (done2 == 1 && *done < steps) {
            const int64_t before = *done;
            int64_t done3 = 0;
            rc2 = hm_std_merge_steps(e, c, thr, X_dev, ld, steps - before, rec_out + 4 * before, &done3, stream);
            if (rc2) return rc2;
            *done = before + done3;
        }
        return HM_OK;
    }
    e->pipe_fault_at = -1;
    e->armed = (*done == steps) && !piped;         // (the pipelined batch leaves its two sets armed for ITS next steps only)
    e->armed_rb = 0; e->armed_re = -1;
    if (time_all) {
        e->last_batch_ms = e->last_batch_scan_ms = 0.f;
        e->last_batch_steps = 0;
        if (*done == steps) {                     // read later (hm_read_loop_events): not in the call being timed
            e->loop_unread_steps = steps;
            e->loop_unread_pairs.assign(all_pairs, all_pairs + steps);
        }
        return HM_OK;
    }
    if (*done == steps) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) {
            // one launch of the batch is timed; the totals count it once (bench: mean launch duration)
            e->last_scan_ms = ms; e->last_pairs = timed_pairs; e->last_passes = 1;
            e->tot_scan_ms += ms; e->tot_pairs += timed_pairs; e->tot_launches += 1;
        }
    }
    return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// incremental loop
// ------------------------------------------------------------------------------------------------
struct IncrArgs {
    float* img;
    unsigned char* img16;
    int RS, d, KC, sign_mode;
    float c, sqrt_c, thr;
    float* X;
    int64_t ld;
    int64_t new_row;             // = partners [0, new_row)
    int32_t* len;
    LoopState* loop;
    ArgminRec* rec_ring;         // this step's record
    uint32_t* rmax2_bits;
    int step;                    // index inside the batch
};

// One launch per merge, no inter-block hand-off inside it: every block (1) folds the previous step's row pass into the
// running minimum -- the same few loads and compares in every block --, (2) recomputes the merged row (wave 0; block 0
// also stores it and writes the step's record), (3) scans its slice of the image against the new row and publishes its
// nearest partner with ONE 64-bit atomicMin.  The fold of THIS step's row pass happens at the start of the next launch
// (or on the host after the last one), so there is no ticket, no fence and no final block.
#define HM_INCR_WAVES 2
__global__ __launch_bounds__(64 * HM_INCR_WAVES) void hm_incr_step_kernel(const IncrArgs a)
{
    extern __shared__ __align__(16) float tiles[];
    __shared__ MidScratch ms;
    __shared__ __align__(16) float ximg[HM_MAX_D1 + 4];          // the merged row in image layout (what the tiles are compared with)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = tiles + wv * HM_TILE_ROWS * a.RS;
    // the first tile of partner rows does not depend on which pair is merged: its loads are in flight during the fold
    // of the previous step and the midpoint
    const int64_t nt = (a.new_row + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
    int64_t tl = (int64_t)blockIdx.x * HM_INCR_WAVES + wv;
    TileRegs tr;
    if (tl < nt) hm_tile_load(a.img, a.RS, tl * HM_TILE_ROWS, a.new_row, tr, lane);
    LoopState* loop = a.loop;
    const uint32_t stop = loop->stop;
    ArgminRec best = loop->best[a.step & 1];
    if (a.step > 0 && stop == 0u) {
        const unsigned long long rk = loop->rowkey[a.step - 1];        // complete: the previous launch has ended
        if (rk != ~0ull) {
            const uint32_t db = (uint32_t)(rk >> 32), ri = (uint32_t)rk, rj = (uint32_t)(a.new_row - 1);
            if (best.found != 1u || hm_key_less(db, ri, rj, best.dbits, best.i, best.j)) { best.found = 1u; best.dbits = db; best.i = ri; best.j = rj; }
        }
    }
    if (stop != 0u || best.found != 1u) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ArgminRec r; r.found = stop != 0u ? 3u : 0u; r.dbits = 0; r.i = 0xffffffffu; r.j = 0xffffffffu;
            *a.rec_ring = r;
            if (stop == 0u) loop->stop = 1u;              // no pair below the threshold: the loop ends here
        }
        return;
    }
    // ---- the merged row: every block computes it (wave 0), block 0 stores it ----
    if (wv == 0) {
        const int32_t li = a.len[best.i], lj = a.len[best.j];
        const float w = (float)((double)lj / (double)(li + lj));
        hm_wave_stage_rows(a.img, a.RS, a.d, best.i, best.j, ms, lane);
        const float r2 = hm_wave_midpoint(a.d, w, a.c, a.sign_mode, ms, true, lane);
        if (blockIdx.x == 0) {
            hm_wave_store_row(ms, r2, a.d, a.RS, a.KC, a.X, a.ld, a.img, a.img16, a.new_row, a.rmax2_bits, lane);
            if (lane == 0) {
                a.len[a.new_row] = li + lj;
                *a.rec_ring = best;                        // the pair this step merged
                loop->best[(a.step + 1) & 1] = best;       // the running minimum as of this step's start (read by the next launch)
                loop->steps_done += 1u;
            }
        }
        hm_tile_fixed_row(ms.so, a.RS, a.d, ximg, lane);
    }
    __syncthreads();
    // ---- nearest partner of the new row among rows [0, new_row): lane l of a wave takes row l of the wave's tile ----
    unsigned long long key = ~0ull;
    for (; tl < nt; tl += (int64_t)gridDim.x * HM_INCR_WAVES) {
        hm_tile_store(tile, a.RS, tr, lane);
        hm_wave_lds_sync();
        const int64_t nxt = tl + (int64_t)gridDim.x * HM_INCR_WAVES;
        if (nxt < nt) hm_tile_load(a.img, a.RS, nxt * HM_TILE_ROWS, a.new_row, tr, lane);
        const float u = hm_tile_u(tile, a.RS, a.d, ximg, a.sign_mode, lane);
        const int64_t i = tl * HM_TILE_ROWS + lane;
        const float dd = hm::dist_from_u(u, a.sqrt_c);
        if (i < a.new_row && dd < a.thr) {
            const unsigned long long k64 = ((unsigned long long)hm::fbits(dd) << 32) | (unsigned long long)(uint32_t)i;
            key = k64 < key ? k64 : key;
        }
        hm_wave_lds_sync();
    }
    key = hm_wave_min_u64(key);
    if (lane == 0 && key != ~0ull) atomicMin(&loop->rowkey[a.step], key);
}

// K steps of the incremental loop.  best_*: the current nearest pair (from a full search); on return the
// running minimum after the last executed step.  rec_out / done as hm_std_merge_steps (found = 2 never occurs).
extern "C" int hm_incr_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* best_io,
                                   uint32_t* rec_out, int64_t* done, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_incr_merge_steps: engine is NULL");
    if (!X_dev || ld < e->d1 || !rec_out || !done || !best_io || steps < 0 || steps > HM_LOOP_MAX_STEPS || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_incr_merge_steps: bad arguments");
    if (!e->have_len) return hm_fail(e, HM_E_STATE, "hm_incr_merge_steps: token lengths not set (hm_set_token_lengths)");
    *done = 0;
    if (steps == 0) return HM_OK;
    if (e->n + steps > e->max_rows) return hm_fail(e, HM_E_CAPACITY, "hm_incr_merge_steps: the table cannot take that many rows");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    e->armed = false;
    const int64_t n0 = e->n;
    // initial state in one copy: zeros, best[0] = the caller's nearest pair, row keys of this batch's steps = "none"
    if (!e->h_loop) HM_HIP(hipHostMalloc(&e->h_loop, sizeof(LoopState), hipHostMallocDefault));
    const size_t state_bytes = offsetof(LoopState, rowkey) + sizeof(unsigned long long) * (size_t)steps;
    memset(e->h_loop, 0, offsetof(LoopState, rowkey));
    memset(e->h_loop->rowkey, 0xff, sizeof(unsigned long long) * (size_t)steps);
    e->h_loop->best[0].found = best_io[0]; e->h_loop->best[0].dbits = best_io[1]; e->h_loop->best[0].i = best_io[2]; e->h_loop->best[0].j = best_io[3];
    HM_HIP(hipMemcpyAsync(e->d_loop, e->h_loop, state_bytes, hipMemcpyHostToDevice, s));
    IncrArgs a;
    a.img = e->img; a.img16 = e->img16; a.RS = e->RS; a.d = e->d; a.KC = e->KC; a.sign_mode = e->sign_mode;
    a.c = c; a.sqrt_c = sqrtf(c); a.thr = thr; a.X = X_dev; a.ld = ld; a.len = e->d_len; a.loop = e->d_loop;
    a.rmax2_bits = e->d_rmax2;
    const size_t lds = sizeof(float) * (size_t)HM_INCR_WAVES * HM_TILE_ROWS * e->RS;
    const void* incr_fn = reinterpret_cast<const void*>(&hm_incr_step_kernel);
    if (e->attr_done.find(incr_fn) == e->attr_done.end()) {                    // per engine (= per device), not process-wide
        HM_HIP(hipFuncSetAttribute(incr_fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(float) * (size_t)HM_INCR_WAVES * HM_TILE_ROWS * 4 * HM_TILE_MAXQ)));
        e->attr_done.insert(incr_fn);
    }
    for (int64_t k = 0; k < steps; ++k) {
        a.new_row = n0 + k;
        a.step = (int)k;
        a.rec_ring = e->d_loop_recs + k;
        const int64_t nt = (a.new_row + HM_TILE_ROWS - 1) / HM_TILE_ROWS;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nt + HM_INCR_WAVES - 1) / HM_INCR_WAVES, 512));
        hipLaunchKernelGGL(hm_incr_step_kernel, dim3(grid), dim3(64 * HM_INCR_WAVES), lds, s, a);
        HM_HIP(hipGetLastError());
    }
    HM_HIP(hipMemcpyAsync(e->h->loop_recs, e->d_loop_recs, sizeof(ArgminRec) * (size_t)steps, hipMemcpyDeviceToHost, s));
    HM_HIP(hipMemcpyAsync(e->h_loop, e->d_loop, state_bytes, hipMemcpyDeviceToHost, s));          // final state: best[] and the last row key
    HM_HIP(hipStreamSynchronize(s));
    // (the slot the last launch WROTE: step steps - 1 wrote best[steps & 1]; a launch that stopped wrote nothing -- unused then)
    e->h->rec = e->h_loop->best[steps & 1];
    e->h->ctr64[0] = e->h_loop->rowkey[steps - 1];
    hm_unpack_recs(e->h->loop_recs, steps, rec_out, done);
    // the running minimum after the last executed step: its start-of-step value folded with that step's row pass
    // (when the batch ran to its end; an earlier stop means the minimum did not exist: found = 0)
    ArgminRec fin = e->h->rec;
    if (*done == steps) {
        const unsigned long long rk = e->h->ctr64[0];
        if (rk != ~0ull) {
            const uint32_t db = (uint32_t)(rk >> 32), ri = (uint32_t)rk, rj = (uint32_t)(n0 + steps - 1);
            const bool less = fin.found != 1u || db < fin.dbits || (db == fin.dbits && (ri < fin.i || (ri == fin.i && rj < fin.j)));
            if (less) { fin.found = 1u; fin.dbits = db; fin.i = ri; fin.j = rj; }
        }
    } else if (*done < steps) {
        // the loop stopped at step *done: the fold of step *done - 1's row pass was made by that launch's blocks; what it
        // found (nothing below the threshold) is what stopped the loop
        fin.found = 0u;
    }
    best_io[0] = fin.found; best_io[1] = fin.dbits; best_io[2] = fin.i; best_io[3] = fin.j;
    e->n = n0 + *done;
    return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// row-sharded loop: every rank searches its row range, the ranks' records are all-gathered by the caller (RCCL, on the
// same stream), and every rank applies the same merge to its replica -- without the host looking at any of it
// ------------------------------------------------------------------------------------------------
struct ShardMergeArgs {
    const ArgminRec* recs;       // `world` records, rank order
    int world;
    float* img;
    unsigned char* img16;
    int RS, d, KC, sign_mode;
    float c;
    float* X;
    int64_t ld;
    int64_t new_row;
    int32_t* len;
    LoopState* loop;
    ArgminRec* rec_ring;
    uint32_t* rmax2_bits;
};

// one wave: lane r holds rank r's record; global nearest pair = lexicographic min of (d bits, i, j) over the ranks that
// found one (identical on every rank); an overflow anywhere (found = 2) or no pair at all ends the loop
__global__ __launch_bounds__(64) void hm_shard_merge_kernel(const ShardMergeArgs a)
{
    __shared__ MidScratch ms;
    const int lane = threadIdx.x;
    LoopState* loop = a.loop;
    if (loop->stop != 0u) {
        if (lane == 0) { ArgminRec r; r.found = 3u; r.dbits = 0; r.i = 0xffffffffu; r.j = 0xffffffffu; *a.rec_ring = r; }
        return;
    }
    ArgminRec mine; mine.found = 0u; mine.dbits = 0xffffffffu; mine.i = 0xffffffffu; mine.j = 0xffffffffu;
    if (lane < a.world) mine = a.recs[lane];
    const bool overflow = __ballot(lane < a.world && (mine.found == 2u || mine.found == 3u)) != 0ull;
    uint32_t b0 = mine.found == 1u ? mine.dbits : 0xffffffffu, b1 = mine.found == 1u ? mine.i : 0xffffffffu,
             b2 = mine.found == 1u ? mine.j : 0xffffffffu;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o0 = __shfl_xor(b0, off, 64), o1 = __shfl_xor(b1, off, 64), o2 = __shfl_xor(b2, off, 64);
        if (hm_key_less(o0, o1, o2, b0, b1, b2)) { b0 = o0; b1 = o1; b2 = o2; }
    }
    if (overflow || b1 == 0xffffffffu) {
        if (lane == 0) {
            ArgminRec r; r.found = overflow ? 2u : 0u; r.dbits = 0; r.i = 0xffffffffu; r.j = 0xffffffffu;
            *a.rec_ring = r;
            loop->stop = overflow ? 2u : 1u;
        }
        return;
    }
    const int32_t li = a.len[b1], lj = a.len[b2];
    const float w = (float)((double)lj / (double)(li + lj));
    hm_wave_stage_rows(a.img, a.RS, a.d, b1, b2, ms, lane);
    const float r2 = hm_wave_midpoint(a.d, w, a.c, a.sign_mode, ms, true, lane);
    hm_wave_store_row(ms, r2, a.d, a.RS, a.KC, a.X, a.ld, a.img, a.img16, a.new_row, a.rmax2_bits, lane);
    if (lane == 0) {
        a.len[a.new_row] = li + lj;
        ArgminRec r; r.found = 1u; r.dbits = b0; r.i = b1; r.j = b2;
        *a.rec_ring = r;
        loop->steps_done += 1u;
    }
}

extern "C" int hm_shard_loop_begin(hm_engine* e, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_shard_loop_begin: engine is NULL");
    if (!e->have_len) return hm_fail(e, HM_E_STATE, "hm_shard_loop_begin: token lengths not set (hm_set_token_lengths)");
    if (e->shard_loop) return hm_fail(e, HM_E_STATE, "hm_shard_loop_begin: a loop is already open");
    HM_HIP(hipSetDevice(e->device));
    HM_HIP(hipMemsetAsync(e->d_loop, 0, sizeof(LoopState), (hipStream_t)stream));
    e->shard_loop = true;
    e->shard_n0 = e->n;
    return HM_OK;
}

extern "C" int hm_shard_merge_step(hm_engine* e, const uint32_t* recs_dev, int world, float c, float* X_dev, int64_t ld, int64_t step,
                                   void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_shard_merge_step: engine is NULL");
    if (!e->shard_loop) return hm_fail(e, HM_E_STATE, "hm_shard_merge_step: no open loop (hm_shard_loop_begin)");
    if (!recs_dev || world < 1 || world > 64 || !X_dev || ld < e->d1 || step < 0 || step >= HM_LOOP_MAX_STEPS || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_shard_merge_step: bad arguments");
    if (e->n + 1 > e->max_rows) return hm_fail(e, HM_E_CAPACITY, "hm_shard_merge_step: the table is full");
    HM_HIP(hipSetDevice(e->device));
    ShardMergeArgs a;
    a.recs = reinterpret_cast<const ArgminRec*>(recs_dev); a.world = world;
    a.img = e->img; a.img16 = e->img16; a.RS = e->RS; a.d = e->d; a.KC = e->KC; a.sign_mode = e->sign_mode;
    a.c = c; a.X = X_dev; a.ld = ld; a.new_row = e->n; a.len = e->d_len; a.loop = e->d_loop;
    a.rec_ring = e->d_loop_recs + step; a.rmax2_bits = e->d_rmax2;
    hipLaunchKernelGGL(hm_shard_merge_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    HM_HIP(hipGetLastError());
    e->n += 1;                                    // optimistic: corrected by hm_shard_loop_end
    return HM_OK;
}

extern "C" int hm_shard_loop_end(hm_engine* e, int64_t steps, uint32_t* rec_out, int64_t* done, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_shard_loop_end: engine is NULL");
    if (!e->shard_loop) return hm_fail(e, HM_E_STATE, "hm_shard_loop_end: no open loop");
    if (!rec_out || !done || steps < 0 || steps > HM_LOOP_MAX_STEPS) return hm_fail(e, HM_E_ARG, "hm_shard_loop_end: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    e->shard_loop = false;
    *done = 0;
    if (steps > 0) {
        HM_HIP(hipMemcpyAsync(e->h->loop_recs, e->d_loop_recs, sizeof(ArgminRec) * (size_t)steps, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        hm_unpack_recs(e->h->loop_recs, steps, rec_out, done);
    }
    e->n = e->shard_n0 + *done;
    if (*done < steps) e->armed = false;          // a search that skipped itself armed nothing
    return HM_OK;
}
