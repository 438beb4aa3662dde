// hm_comm.hip -- the row-sharded search with its exchange step INSIDE the library:
// hm_comm_init binds an RCCL communicator to the engine; hm_shard_merge_steps enqueues, per step and without any host
// code in between, [pair scan of this rank's rows -> tail -> 16-byte record] -> ncclAllGather of the ranks' records ->
// [global minimum + merge into this rank's replica]; hm_global_argmin / hm_global_topk are the one-shot forms (C1 / C2).
//
// The reference has nothing here (single process); the loop accelerated is
// tokenizer/hyperbolic_merge.py:357-412 (search -> [0] -> merge), the refresh fast_hyperbolic_merge.py:336-374.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy torch has loaded when there is one, else the ROCm
// installation's): a process that never shards never touches it, and the library has no link-time dependency on it.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>

#include "hm_common.h"

struct HmRccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

static HmRccl* hm_rccl()
{
    static HmRccl r;
    if (r.lib || !r.err.empty()) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"); return &r; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        r.err = "librccl.so.1 lacks an expected symbol";
        dlclose(r.lib);
        r.lib = nullptr;
    }
    return &r;
}

#define HM_NCCL(call)                                                                                         \
    do {                                                                                                      \
        ncclResult_t _st = (call);                                                                            \
        if (_st != ncclSuccess) return hm_fail(e, HM_E_COMM, std::string(#call) + ": " + R->GetErrorString(_st)); \
    } while (0)

extern "C" int hm_comm_unique_id(void* id_out128)
{
    hm_engine* e = nullptr;
    if (!id_out128) return hm_fail(nullptr, HM_E_ARG, "hm_comm_unique_id: NULL output");
    HmRccl* R = hm_rccl();
    if (!R->lib) return hm_fail(nullptr, HM_E_COMM, R->err);
    ncclUniqueId id;
    HM_NCCL(R->GetUniqueId(&id));
    static_assert(sizeof(id) == HM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out128, &id, sizeof(id));
    return HM_OK;
}

extern "C" int hm_comm_init(hm_engine* e, const void* id128, int rank, int world)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_comm_init: engine is NULL");
    if (!id128 || world < 1 || world > 64 || rank < 0 || rank >= world) return hm_fail(e, HM_E_ARG, "hm_comm_init: bad arguments");
    if (e->comm) return hm_fail(e, HM_E_STATE, "hm_comm_init: the engine already has a communicator");
    HmRccl* R = hm_rccl();
    if (!R->lib) return hm_fail(e, HM_E_COMM, R->err);
    HM_HIP(hipSetDevice(e->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    HM_NCCL(R->CommInitRank(&comm, world, id, rank));
    e->comm = comm;
    e->rank = rank;
    e->world = world;
    // exchange buffers: this rank's record, the gathered records of a whole batch, the gathered lists of a refresh
    HM_HIP(hipMalloc(&e->d_shard_rec, sizeof(ArgminRec)));
    HM_HIP(hipMalloc(&e->d_shard_recs, sizeof(ArgminRec) * (size_t)HM_LOOP_MAX_STEPS * world));
    HM_HIP(hipMalloc(&e->d_gather, sizeof(uint4) * (size_t)(e->sorted_cap + 1) * (world + 1)));
    return HM_OK;
}

extern "C" int hm_comm_destroy(hm_engine* e)
{
    if (!e) return HM_OK;
    if (e->comm) {
        HmRccl* R = hm_rccl();
        (void)hipSetDevice(e->device);
        if (R->lib) (void)R->CommDestroy(reinterpret_cast<ncclComm_t>(e->comm));
        e->comm = nullptr;
    }
    if (e->d_shard_rec) (void)hipFree(e->d_shard_rec);
    if (e->d_shard_recs) (void)hipFree(e->d_shard_recs);
    if (e->d_gather) (void)hipFree(e->d_gather);
    e->d_shard_rec = nullptr; e->d_shard_recs = nullptr; e->d_gather = nullptr;
    e->world = 1; e->rank = 0;
    return HM_OK;
}

extern "C" int hm_comm_info(const hm_engine* e, int* rank, int* world)
{
    if (!e || !rank || !world) return HM_E_ARG;
    *rank = e->comm ? e->rank : -1;
    *world = e->comm ? e->world : 0;
    return HM_OK;
}

// Row cuts of equal pair count: b[p] = n (1 - sqrt(1 - p / world)), rounded to 256-row blocks on large tables (what
// hyptokenizer_amd/sharding.py's partition_rows does; only coverage matters: every rank runs this same code)
void hm_partition_rows(int64_t n, int world, int rank, int64_t* r0, int64_t* r1)
{
    auto cut = [&](int p) -> int64_t {
        if (p <= 0) return 0;
        if (p >= world) return n;
        double x = (double)n * (1.0 - sqrt(1.0 - (double)p / (double)world));
        if (n >= (int64_t)4 * 256 * world) x = nearbyint(x / 256.0) * 256.0;
        int64_t b = (int64_t)nearbyint(x);
        return std::min<int64_t>(std::max<int64_t>(b, 0), n);
    };
    int64_t lo = 0;
    for (int p = 1; p <= rank; ++p) lo = std::max(lo, cut(p));          // (monotone by construction; max() keeps it so after rounding)
    int64_t hi = lo;
    hi = rank + 1 >= world ? n : std::max(lo, cut(rank + 1));
    *r0 = lo; *r1 = hi;
}

extern "C" int hm_shard_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* rec_out,
                                    int64_t* done, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_shard_merge_steps: engine is NULL");
    if (!e->comm) return hm_fail(e, HM_E_STATE, "hm_shard_merge_steps: no communicator (hm_comm_init)");
    if (!X_dev || ld < e->d1 || !rec_out || !done || steps < 0 || steps > HM_LOOP_MAX_STEPS || !(c > 0.0f))
        return hm_fail(e, HM_E_ARG, "hm_shard_merge_steps: bad arguments");
    *done = 0;
    if (steps == 0) return HM_OK;
    if (e->n + steps > e->max_rows) return hm_fail(e, HM_E_CAPACITY, "hm_shard_merge_steps: the table cannot take that many rows");
    HmRccl* R = hm_rccl();
    hipStream_t s = (hipStream_t)stream;
    int rc = hm_shard_loop_begin(e, stream);
    if (rc) return rc;
    const bool time_all = e->time_loops && !e->loop_evs.empty();
    if (time_all) hm_read_loop_events(e);
    int64_t all_pairs[HM_LOOP_MAX_STEPS];
    const int64_t n0 = e->n;
    for (int64_t k = 0; k < steps && rc == HM_OK; ++k) {
        int64_t r0 = 0, r1 = 0;
        hm_partition_rows(e->n, e->world, e->rank, &r0, &r1);
        if (time_all) {
            if (k == 0) (void)hipEventRecord(e->loop_evs[2 * HM_LOOP_MAX_STEPS], s);
            e->step_ev0 = e->loop_evs[2 * k]; e->step_ev1 = e->loop_evs[2 * k + 1];
            all_pairs[k] = hm_pairs_in_range(e->n, r0, std::min<int64_t>(r1, e->n - 1));
        }
        rc = hm_pairwise_argmin_dev(e, c, thr, r0, r1, reinterpret_cast<uint32_t*>(e->d_shard_rec), stream);
        e->step_ev0 = e->step_ev1 = nullptr;
        if (rc) break;
        ArgminRec* recs = e->d_shard_recs + (size_t)k * e->world;
        const ncclResult_t st = R->AllGather(e->d_shard_rec, recs, 4, ncclInt32, reinterpret_cast<ncclComm_t>(e->comm), s);
        if (st != ncclSuccess) { rc = hm_fail(e, HM_E_COMM, std::string("ncclAllGather: ") + R->GetErrorString(st)); break; }
        rc = hm_shard_merge_step(e, reinterpret_cast<const uint32_t*>(recs), e->world, c, X_dev, ld, k, stream);
    }
    if (time_all && rc == HM_OK) (void)hipEventRecord(e->loop_evs[2 * HM_LOOP_MAX_STEPS + 1], s);
    const int rc_end = hm_shard_loop_end(e, rc == HM_OK ? steps : 0, rec_out, done, stream);
    if (rc) { e->n = n0; return rc; }
    if (rc_end) return rc_end;
    if (time_all) {
        e->last_batch_ms = e->last_batch_scan_ms = 0.f;
        e->last_batch_steps = 0;
        if (*done == steps) {
            e->loop_unread_steps = steps;
            e->loop_unread_pairs.assign(all_pairs, all_pairs + steps);
        }
    }
    return HM_OK;
}

extern "C" int hm_global_argmin(hm_engine* e, float c, float thr, float* d, int32_t* i, int32_t* j, int32_t* found, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_global_argmin: engine is NULL");
    if (!e->comm) return hm_fail(e, HM_E_STATE, "hm_global_argmin: no communicator (hm_comm_init)");
    if (!d || !i || !j || !found || !(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_global_argmin: bad arguments");
    HmRccl* R = hm_rccl();
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *found = 0; *d = 0.f; *i = -1; *j = -1;
    int64_t r0 = 0, r1 = 0;
    hm_partition_rows(e->n, e->world, e->rank, &r0, &r1);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) {
            int rc = hm_pairwise_argmin_dev(e, c, thr, r0, r1, reinterpret_cast<uint32_t*>(e->d_shard_rec), stream);
            if (rc) return rc;
        } else {
            // some rank's emission buffer overflowed: EVERY rank redoes its range through the bounded host form (a collective
            // decision: all ranks saw the same gathered records)
            float dd = 0.f; int32_t ii = -1, jj = -1, ff = 0;
            int rc = hm_pairwise_argmin(e, c, thr, r0, r1, &dd, &ii, &jj, &ff, stream);
            if (rc) return rc;
            union { float f; uint32_t u; } cv; cv.f = dd;
            e->h->rec.found = ff ? 1u : 0u; e->h->rec.dbits = cv.u; e->h->rec.i = (uint32_t)ii; e->h->rec.j = (uint32_t)jj;
            HM_HIP(hipMemcpyAsync(e->d_shard_rec, &e->h->rec, sizeof(ArgminRec), hipMemcpyHostToDevice, s));
        }
        HM_NCCL(R->AllGather(e->d_shard_rec, e->d_shard_recs, 4, ncclInt32, reinterpret_cast<ncclComm_t>(e->comm), s));
        HM_HIP(hipMemcpyAsync(e->h->loop_recs, e->d_shard_recs, sizeof(ArgminRec) * (size_t)e->world, hipMemcpyDeviceToHost, s));
        HM_HIP(hipStreamSynchronize(s));
        bool overflow = false;
        uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu;
        for (int r = 0; r < e->world; ++r) {
            const ArgminRec& q = e->h->loop_recs[r];
            if (q.found == 2u) overflow = true;
            if (q.found == 1u && (q.dbits < b0 || (q.dbits == b0 && (q.i < b1 || (q.i == b1 && q.j < b2))))) { b0 = q.dbits; b1 = q.i; b2 = q.j; }
        }
        if (overflow && pass == 0) { e->armed = false; continue; }
        if (b1 != 0xffffffffu) {
            union { uint32_t u; float f; } cv; cv.u = b0;
            *found = 1; *d = cv.f; *i = (int32_t)b1; *j = (int32_t)b2;
        }
        break;
    }
    return HM_OK;
}

// gathered lists -> one array: rank r's k entries behind each other (headers dropped; padding keys are all ones)
__global__ void hm_gather_flatten_kernel(const uint4* __restrict__ gathered, int world, uint32_t k, uint4* __restrict__ out)
{
    const uint32_t total = (uint32_t)world * k;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const uint32_t r = t / k, q = t - r * k;
        out[t] = gathered[(size_t)r * (k + 1) + 1 + q];
    }
}

__global__ void hm_gather_pack_kernel(const uint4* __restrict__ sorted, uint32_t kk, uint32_t k, unsigned long long count, uint4* __restrict__ out)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < k + 1; t += gridDim.x * blockDim.x) {
        if (t == 0) out[0] = make_uint4(kk, (uint32_t)(count & 0xffffffffull), (uint32_t)(count >> 32), 0u);
        else out[t] = (t - 1 < kk) ? sorted[t - 1] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);
    }
}

extern "C" int hm_global_topk(hm_engine* e, float c, float thr, int64_t k, float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out,
                              int64_t* count, void* stream)
{
    if (!e) return hm_fail(nullptr, HM_E_ARG, "hm_global_topk: engine is NULL");
    if (!e->comm) return hm_fail(e, HM_E_STATE, "hm_global_topk: no communicator (hm_comm_init)");
    if (!n_out || !count || k <= 0 || !d_out || !i_out || !j_out || !(c > 0.0f)) return hm_fail(e, HM_E_ARG, "hm_global_topk: bad arguments");
    if (k > (int64_t)e->sorted_cap) return hm_fail(e, HM_E_CAPACITY, "hm_global_topk: k > 65536");
    if ((uint64_t)k * (uint64_t)e->world > (uint64_t)e->ent_cap) return hm_fail(e, HM_E_CAPACITY, "hm_global_topk: world * k exceeds the selection buffer");
    HmRccl* R = hm_rccl();
    hipStream_t s = (hipStream_t)stream;
    HM_HIP(hipSetDevice(e->device));
    *n_out = 0; *count = 0;
    e->armed = false;
    int64_t r0 = 0, r1 = 0;
    hm_partition_rows(e->n, e->world, e->rank, &r0, &r1);
    // this rank's ordered list stays on the device (e->sorted); its exact candidate count comes back with it
    int64_t valid = 0, total = 0;
    uint4* res = nullptr;
    uint32_t kk = 0;
    if (r1 > r0) {
        int rc = hm_topk_core(e, c, thr, k, r0, r1, false, true, -1, &valid, &total, &res, s);
        if (rc) return rc;
        kk = (uint32_t)std::min<int64_t>(k, valid);
        if (kk > 0 && res) {
            const uint32_t m = (uint32_t)std::min<uint64_t>(e->h->ctr64[2], e->ent_cap);
            rc = hm_select_sorted(e, res, e->ent2, m, kk, s);
            if (rc) return rc;
        }
    }
    e->have_cut = false; e->prev_valid = false;              // (range searches do not feed the whole-table refresh state)
    uint4* mine = e->d_gather;                               // [k + 1] packed list of this rank, then [world][k + 1] gathered
    uint4* all = e->d_gather + (k + 1);
    hipLaunchKernelGGL(hm_gather_pack_kernel, dim3(64), dim3(256), 0, s, e->sorted, kk, (uint32_t)k, (unsigned long long)total, mine);
    HM_HIP(hipGetLastError());
    HM_NCCL(R->AllGather(mine, all, (size_t)(k + 1) * 4, ncclInt32, reinterpret_cast<ncclComm_t>(e->comm), s));
    // headers (entries, counts) of all ranks: one small read-back decides how many entries the merged list has
    for (int r = 0; r < e->world; ++r)
        HM_HIP(hipMemcpyAsync(&e->h_sorted[r], all + (size_t)r * (k + 1), sizeof(uint4), hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    uint64_t have = 0, cnt = 0;
    for (int r = 0; r < e->world; ++r) { have += e->h_sorted[r].x; cnt += (uint64_t)e->h_sorted[r].y | ((uint64_t)e->h_sorted[r].z << 32); }
    *count = (int64_t)cnt;
    const uint32_t want = (uint32_t)std::min<uint64_t>((uint64_t)k, have);
    if (want == 0) return HM_OK;
    const uint32_t m = (uint32_t)((uint64_t)e->world * (uint64_t)k);
    hipLaunchKernelGGL(hm_gather_flatten_kernel, dim3(256), dim3(256), 0, s, all, e->world, (uint32_t)k, e->ent);
    HM_HIP(hipGetLastError());
    int rc = hm_select_sorted(e, e->ent, e->ent2, m, want, s);
    if (rc) return rc;
    HM_HIP(hipMemcpyAsync(e->h_sorted, e->sorted, sizeof(uint4) * (size_t)want, hipMemcpyDeviceToHost, s));
    HM_HIP(hipStreamSynchronize(s));
    for (uint32_t t = 0; t < want; ++t) {
        union { uint32_t u; float f; } cv; cv.u = e->h_sorted[t].x;
        d_out[t] = cv.f; i_out[t] = (int32_t)e->h_sorted[t].y; j_out[t] = (int32_t)e->h_sorted[t].z;
    }
    *n_out = want;
    return HM_OK;
}
