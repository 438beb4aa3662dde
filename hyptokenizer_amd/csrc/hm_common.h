// hm_common.h -- internal declarations shared by the translation units of libhypmerge.so.
//
//   hm_engine.hip   engine lifetime, table upload, statistics            (host)
//   hm_scan.hip     the pair-scan kernel (MFMA prefilter) and its launch  (hot kernel)
//   hm_search.hip   exact re-evaluation / selection kernels and the search entry points of the ABI
//   hm_rows.hip     image construction, merge / midpoint, one-row-vs-all, gathered and row-wise kernels
//   hm_loops.hip    device-resident merge loops (several steps per host call)
//
// Data layout in HBM: the fp32 "scan image" img[rows_alloc][RS], RS = 4*NG + 4 (+ 4 when needed to
// make the 16-byte chunks per row odd), NG = groups of 4 spatial coordinates.  Group g holds spatial
// coordinates s = 4g..4g+3 in the order [s0, s2, s1, s3] so that lane-half h of a wave reads ONE
// 8-byte word (position 2h) holding its operands for the two MFMA k-steps of the group; the time
// chunk [x0, 0, 0, 0] is last.  The bf16 image img16[rows_alloc][16 * hm_row16_chunks(KC) bytes]: KC chunks of
// 8 K-slots as bf16 (spatial coordinates, then the time coordinate split hi + lo in the last four slots of the
// last chunk) and, when KC is even, a trailing chunk [x0 fp32, 0, 0, 0] (odd chunk count: conflict-free
// ds_read_b128).  A k-step of the MFMA is two chunks; an odd KC ends on a half step (v_mfma_f32_32x32x8_bf16_1k):
// d = 100 needs 104 slots = 13 chunks = 208 bytes per row.  A 64-row tile of either image is one contiguous
// block -> LDS-DMA in 1 KiB pieces, no padding.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/hypmerge.h"
#include "hm_device_math.h"

#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------
#define HM_MAX_BLOCK_ROWS 1024
#define HM_MAX_D1 132              // largest table width (d + 1 <= 129) rounded up
#define HM_TIE_SLACK 1024u         // ulps of u' that are treated as "may still order before" for u' >= 2 (see hm_tie_slack)
#define HM_MODE_TOPK 0
#define HM_MODE_ARGMIN 1
#define HM_MODE_HIST 2
#define HM_HIST_BINS 256
#define HM_DIGIT_BINS 4096
#define HM_RANK_LIMIT 49152        // rank sort is O(M^2): narrow by radix digits above this
#define HM_TAIL_BLOCKS 32          // blocks of the argmin tail kernel (one is enough for <= HM_TAIL_SOLO entries)
#define HM_TAIL_THREADS 1024
#define HM_TAIL_SOLO 1024u
#define HM_PIPE_TAIL_MAX 8192u     // survivors a pipelined step's small tail grid takes (more: found = 2, that step goes through the host path)
#define HM_ROWPASS_BLOCKS 256      // blocks of the one-row-vs-all reduction
#define HM_PART_SLOTS 256          // partial records (>= HM_TAIL_BLOCKS, HM_ROWPASS_BLOCKS)
#define HM_LOOP_MAX_STEPS 256
#define HM_BATCH_MAX 4096          // merges per hm_merge_append_batch_host call       // steps one device-resident loop call may enqueue

// prefilter forms (hm_engine_create / hm_set_prefilter)
#define HM_PREFILTER_AUTO 0
#define HM_PREFILTER_F32 1
#define HM_PREFILTER_BF16 2

// floats per fp32 image row
__host__ __device__ constexpr int hm_row_floats(int NG) { return 4 * NG + 4 + (((NG + 1) % 2 == 0) ? 4 : 0); }
// 16-byte chunks per bf16 image row: KC chunks of 8 K-slots (+ the [x0] chunk that makes an even count odd)
__host__ __device__ constexpr int hm_row16_chunks(int KC) { return KC + ((KC & 1) ? 0 : 1); }
__device__ __forceinline__ int hm_pos_in_group(int s) { return ((s & 1) << 1) | ((s >> 1) & 1); }  // 0,2,1,3
// offset of spatial coordinate s inside an fp32 image row
__device__ __forceinline__ int hm_img_off(int s) { return 4 * (s >> 2) + hm_pos_in_group(s & 3); }

struct ScanArgs {
    const float* img;
    const unsigned char* img16;   // bf16 image (BF = 1 kernels)
    int bf16;                     // host-side: which form this launch uses
    int shape;                    // host-side: block shape selector of the bf16 form (0: 256-row blocks, 1: 512-row blocks)
    int n;                        // live rows
    int row_begin, row_end;       // i range
    int col_begin;                // only pairs with j >= col_begin (incremental refresh: the rows appended since the last one)
    int rb_first;                 // first row block
    int nct;                      // column tiles in total = ceil(n / cols per tile)
    // work decomposition: 1-D grid of items in up to HM_SCAN_PHASES phases.  Phase q covers the row blocks from ph_rb0[q]
    // up to the next phase's first; each of its ph_items[q] blocks takes ph_ch[q] column tiles (item b of the phase: row block
    // ph_rb0[q] + b / ph_chunks[q], tiles from ph_ctmin[q] + (b % ph_chunks[q]) * ph_ch[q]).  Chunk lengths shrink from
    // phase to phase: long items first, ever shorter ones behind them, so that the slots that free up late are filled
    // with work that still ends with the rest (the launch's tail is made of the shortest blocks).
#define HM_SCAN_PHASES 6
    int n_ph;
    int ph_items[HM_SCAN_PHASES], ph_rb0[HM_SCAN_PHASES], ph_chunks[HM_SCAN_PHASES], ph_ch[HM_SCAN_PHASES], ph_ctmin[HM_SCAN_PHASES];
    float u_hi;                   // candidate prefilter: u < u_hi
    float u_lo;                   // surely-below-threshold bound: u' < u_lo
    uint32_t cut_bits;            // emit when bits(u') <= cut_bits (or not sure)
    int tie_imax;                 // zero-distance ties are emitted only for rows i <= tie_imax
    int thr_pos;                  // thr > 0: u' == 1 gives d == 0, surely a candidate
    int count_sure;               // TOPK mode: 1 = count every candidate (exact total); 0 = only what the cut asks for is visited
    uint4* ent;
    uint32_t ent_cap;
    unsigned long long* ctr64;    // [0] sure count  [1] running best key (argmin)  [2] emitted entries  [3] spare
    uint32_t* hist;               // HIST mode: HM_HIST_BINS bins
    uint32_t hist_lo;
    uint32_t hist_shift;
    int sample_stride;
    const uint32_t* rmax2_bits;   // [0] largest squared row norm, [1] largest squared spatial norm (float bits)
    const uint32_t* stop;         // device-resident loops: a non-zero word makes every block return at once (may be NULL)
    // pipelined loop: this scan reads what the tail kernel of two steps ago wrote (a row, the armed counters).  That kernel ran
    // on another stream a whole scan ago; instead of a stream-level event wait (6 us on this stream per step) every block
    // checks that `*order_seen >= order_need` and, should it ever not hold, raises the loop's stop word to 5 and leaves -- the
    // host then redoes the remaining steps strictly sequentially.  A guard, never a wait.
    const uint32_t* order_seen;
    uint32_t order_need;
    uint32_t* order_fault;
    // item queue (knob `dyn`, on by default): the grid is only as large as the device holds at once; block b
    // starts on item b and then draws further items of the SAME list, in order, from a device counter until n_items is
    // reached -- no dispatch gap between a slot's items, and the slots of a fast XCD take work that a static grid
    // (workgroup id mod 8 -> XCD) would have pinned to a slow one: 3 % off the launch at V = 50 k and 100 k.
    // The word is (launch tag << 24) | next item offset: every block raises it to its launch's tag first (atomicMax), so it
    // needs no reset between launches and a zeroed or stale word is harmless.  It lives in a cache line of its OWN
    // (hm_engine::d_queue): on the line of ctr64[] -- whose running key every wave polls -- each draw took ~20 us
    // (profiles/r03g_ab_item_queue_shared_line.txt: the launch 22 % slower than the static grid).
    int dyn;
    int n_items;
    unsigned long long q_tag;
    unsigned long long* q_ctr;
};

// Seed of the argmin search's running key, kept on the device between searches: the key of the last
// nearest pair found.  While rows are only appended that pair still exists, so its key bounds the next
// search from its first tile on.  `valid` is cleared whenever an existing row changes.
struct ArgminSeed { unsigned long long key; uint32_t i, valid; };
struct ArgminRec { uint32_t found, dbits, i, j; };      // found: 0 none, 1 pair, 2 emission overflow, 3 step skipped
struct ArgminPart { uint32_t dbits, i, j, pad; };
struct Prefix { uint32_t val[3]; uint32_t mask[3]; };

struct HostCtl {                 // pinned host mirror of small device results
    uint32_t ctr[8];             // [1] valid [2] valid & !sure [3] compacted [4] margin violations [5] complete-region count
    unsigned long long ctr64[4]; // [0] sure count [1] argmin key [2] emitted
    ArgminRec rec;
    ArgminRec rec2[2];           // [0] record, [1].found / .dbits = emitted count (low / high)
    ArgminRec loop_recs[HM_LOOP_MAX_STEPS];
    uint32_t hist[HM_DIGIT_BINS];
};

// state of the device-resident merge loops (one per engine, in HBM)
struct LoopState {
    uint32_t stop;               // 0 running, 1 no candidate, 2 emission overflow, 5 pipelined loop: a scan started before its inputs were
                                 // written, 6 pipelined loop: more survivors than its small tail grid takes
    uint32_t steps_done;
    // incremental loop: running nearest pair, double-buffered by step parity: the blocks of step k read best[k & 1]
    // (written by the previous launch, or by the host for k = 0) and block 0 writes the folded value to best[(k + 1) & 1]
    // -- no slot is read and written in one launch
    ArgminRec best[2];
    uint32_t tails_done;         // pipelined loop: tail kernels that have finished (the scans' order guard)
    uint32_t pad[1];
    // incremental loop: nearest partner of the row appended by step k, folded in by every block with one 64-bit
    // atomicMin: (bits(d) << 32) | i  (the partner is always paired with that new row); all ones = none
    unsigned long long rowkey[HM_LOOP_MAX_STEPS];
};

struct hm_engine {
    int device = 0;
    int n_cu = 256;
    // work-decomposition knobs (hm_debug_set_knob; tuning builds also read HM_TUNE_<NAME>)
    int chunk_f32 = 32, chunk_bf16 = 128, tail_div = 4;   // (bf16: 96 with the static grid; 112-144 measure alike with the item queue)
    double tail_fraction = 0.20;
    // phases of the item list: work share of each phase from the top of the triangle (the last one takes the rest), chunk
    // length divisor per phase.  phases = 2 is the round-2 list: [1 - tail_fraction, tail_fraction] at [1, tail_div]
    int phases = 4;                      // (measured, interleaved A/B at V = 50 k / 100 k: +1.3 % / +2.6 % over the two-phase list)
    double ph_share[HM_SCAN_PHASES - 1] = {0.70, 0.15, 0.10, 0.0, 0.0};
    int ph_div[HM_SCAN_PHASES] = {1, 2, 4, 8, 16, 32};
    unsigned long long* d_queue = nullptr; // item queue words: 2 x 128 B (one per counter set), never reset (ScanArgs::q_tag)
    int dyn_slots = 0;                    // knob `dyn_slots` (tests): resident-grid size of the item queue; 0 = what the device holds (occupancy x CUs)
    bool dyn_queue = true;                // knob `dyn`: resident grid + in-order item queue (ScanArgs::dyn) for ARGMIN / TOPK launches
    unsigned long long scan_tag = 0;      // launch tags of the item queue (monotonic per engine)
    std::map<const void*, int> scan_slots; // resident blocks per scan kernel instantiation on this engine's device
    int force_shape = -1;                 // HM_TUNE_SHAPE: bf16 block shape of every launch (tuning builds)
    int64_t big_min_rows = 80000;         // bf16 form: launches covering at least the pairs of this many rows use 512-row blocks
    int64_t max_rows = 0, rows_alloc = 0, n = 0;
    int d1 = 0, d = 0, NG = 0, RS = 0, sign_mode = 0;
    float* img = nullptr;
    unsigned char* img16 = nullptr;
    int KC = 0, RB16 = 0;                 // chunks of 8 K-slots and bytes per bf16 image row
    int precision = HM_PREFILTER_AUTO;
    bool bf16_ok = true;                  // false: no bf16 image for this width (d > 124)
    uint4* ent = nullptr;
    uint4* ent2 = nullptr;
    uint4* sorted = nullptr;
    uint32_t ent_cap = 0;
    uint32_t* d_ctr = nullptr;            // 8 x u32 (post-kernel counters)
    uint32_t* d_rmax2 = nullptr;
    uint32_t* d_rmax2_mem = nullptr;            // the allocation: two pairs; d_rmax2 = the live one, the other is kept zeroed (hm_project_table swaps them)
    unsigned long long* d_ctr64 = nullptr; // 4 x u64
    ArgminRec* d_rec = nullptr;            // 2 records
    ArgminRec* d_loop_recs = nullptr;      // HM_LOOP_MAX_STEPS records
    LoopState* d_loop = nullptr;
    int32_t* d_len = nullptr;              // token lengths (device-resident loops), max_rows entries
    bool have_len = false;
    float topk_f32_thr = 0.0f;
    bool force_exact = false;                   // knob exact_search
    float topk_exact_thr = 0.0f;                // > 0: whole searches at thresholds >= this one go straight to the exact path (hm_exact.hip)
    uint32_t* d_rowcnt = nullptr;               // hm_exact.hip: per-row counts (allocated on first use)
    LoopState* h_loop = nullptr;                // pinned host image of a LoopState: the incremental loop's initial state goes up, and its final state comes back, in ONE copy each
    bool force_f32 = false;
    bool armed = false;
    int64_t armed_rb = 0, armed_re = 0;
    ArgminSeed* d_seed = nullptr;
    ArgminPart* d_parts = nullptr;
    uint32_t* d_hist = nullptr;
    HostCtl* h = nullptr;                  // pinned
    uint4* h_sorted = nullptr;             // pinned
    uint32_t sorted_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // software-pipelined standard loop: a second (high-priority) stream for the step's tail work, which then runs under
    // the NEXT step's scan; per buffer set an event behind the scan and one behind the tail; the new row's key per set
    bool pipeline = true;
    int64_t pipeline_min_pairs = 800000000ll;   // (knob "pipeline_pairs"): ~40 000 rows -- a scan of 150 us against 50 us of tail work under it
    int pipe_fault_at = -1;              // test hook (knob "pipe_fault_at"): the scan of this step of the next batch is made to trip its order guard
    hipStream_t aux = nullptr;
    hipEvent_t ev_scan[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr}, ev_join = nullptr;
    unsigned long long* d_rowkey = nullptr;     // [0], [1]: the pipelined loop's two sets; [2]: hm_row_argmin
    // cut prediction for top-k: valid while rows are only appended
    bool have_cut = false;
    uint32_t last_cut_bits = 0;
    int64_t last_cut_k = 0;
    float last_cut_c = 0.f;
    // the ordered list of the last whole-table top-k search (device copy): while rows are only appended, the next one is
    // the k smallest of (that list) + (pairs with a new row)
    uint4* d_prev = nullptr;
    // staging of host-side merge batches (hm_merge_append_batch_host): pinned + device, HM_BATCH_MAX entries x {i, j, w}
    int32_t* h_batch = nullptr;
    int32_t* d_batch = nullptr;
    hipEvent_t ev_batch = nullptr;
    bool batch_in_flight = false;
    bool prev_valid = false;
    bool incremental_topk = true;         // HM_TUNE_INCR_TOPK=0: every refresh scans the whole triangle
    int64_t prev_k = 0, prev_n = 0;
    float prev_thr = 0.f;
    bool debug_cut = false;               // hm_debug_force_cut: the next top-k starts from last_cut_bits as given
    // hm_topk_refresh_begin .. _end
    bool refresh_pending = false;
    int64_t refresh_k = 0;
    float refresh_c = 0.f, refresh_thr = 0.f;
    void* refresh_stream = nullptr;
    // row-sharded device-resident loop (hm_shard_loop_begin .. _end): searches skip themselves once the loop has stopped
    bool shard_loop = false;
    int64_t shard_n0 = 0;
    // RCCL communicator bound by hm_comm_init (hm_comm.hip): opaque ncclComm_t, this rank, ranks, exchange buffers
    void* comm = nullptr;
    int rank = 0, world = 1;
    ArgminRec* d_shard_rec = nullptr;      // this rank's record of a step
    ArgminRec* d_shard_recs = nullptr;     // gathered records: HM_LOOP_MAX_STEPS x world
    uint4* d_gather = nullptr;             // refresh: this rank's packed list [k + 1], then the gathered lists [world][k + 1]
    hipEvent_t step_ev0 = nullptr, step_ev1 = nullptr;   // set around hm_pairwise_argmin_dev by the in-library sharded loop's timing mode
    // hm_debug_time_loops: an event pair around EVERY scan of a device-resident batch and around the batch itself
    bool time_loops = false;
    std::vector<hipEvent_t> loop_evs;     // 2 * HM_LOOP_MAX_STEPS + 2, created on first use
    float last_batch_ms = 0.f, last_batch_scan_ms = 0.f;
    int64_t last_batch_steps = 0;
    // the event pairs of the last complete batch are read when somebody asks (hm_scan_totals, hm_last_loop_timing) or
    // when the next batch needs the events -- not inside the call a benchmark is timing
    int64_t loop_unread_steps = 0;
    std::vector<int64_t> loop_unread_pairs;
    // stats
    float last_scan_ms = 0.f;
    int64_t last_pairs = 0, last_emitted = 0;
    int last_passes = 0;
    bool pending_timing = false;
    int64_t pending_pairs = 0;
    double tot_scan_ms = 0.0;
    int64_t tot_pairs = 0, tot_launches = 0;
    std::set<const void*> attr_done;      // kernels whose dynamic-LDS attribute is set on this engine's device
    std::string err;
};

int hm_fail(hm_engine* e, int code, const std::string& msg);

#define HM_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t _st = (call);                                                                      \
        if (_st != hipSuccess)                                                                        \
            return hm_fail(e, (int)_st, std::string(#call) + ": " + hipGetErrorString(_st));          \
    } while (0)
#define HM_HIP0(call)                                                                                 \
    do {                                                                                              \
        hipError_t _st = (call);                                                                      \
        if (_st != hipSuccess)                                                                        \
            return hm_fail(nullptr, (int)_st, std::string(#call) + ": " + hipGetErrorString(_st));    \
    } while (0)

// ---- threshold bounds in the u domain (double precision on the host) ----
struct Bounds { float u_hi, u_lo; bool none; int thr_pos; };
Bounds hm_bounds(float thr, float c);

// ---- hm_scan.hip ----
bool hm_use_bf16(const hm_engine* e);
bool hm_prepare_scan(hm_engine* e, const Bounds& b, int64_t row_begin, int64_t row_end, ScanArgs& a, dim3& grid, int64_t n_limit = -1,
                     int64_t col_begin = 0);
hipError_t hm_launch_scan(hm_engine* e, int mode, const ScanArgs& a, dim3 grid, hipStream_t s, hipEvent_t ev0 = nullptr,
                          hipEvent_t ev1 = nullptr);
int64_t hm_pairs_in_range(int64_t n, int64_t r0, int64_t r1);
void hm_flush_pending_timing(hm_engine* e);
void hm_read_loop_events(hm_engine* e);

// ---- hm_rows.hip ----
int hm_build_rows(hm_engine* e, const float* X, int64_t ld, int64_t r0, int64_t r1, hipStream_t s);
// (device) one merge by one wave: see hm_rows_device.h

// ---- hm_search.hip ----
// argmin tail: exact re-evaluation of the emitted entries, final record, seed, arming of the next search and --
// when merge.X != nullptr -- the merge of the found pair, all in one launch
struct MergeFuse {
    float* X;               // caller's table (nullptr: no fused merge)
    int64_t ld;
    int64_t new_row;
    float c;
    const int32_t* len;     // token lengths (device), updated in place
    int32_t* len_rw;
    LoopState* loop;        // stop flag / step counter (may be nullptr)
    ArgminRec* rec_ring;    // record of this step is also written here (may be nullptr)
    // software-pipelined loop (hm_loops.hip): this step's scan left its entries / counters in the buffer set given here
    // (nullptr: the engine's first set), and the newest row's nearest partner -- which that scan did not cover -- waits in
    // *rowkey as (bits(d) << 32) | i, all ones = none; the tail folds it in, re-arms the set and clears the key
    uint4* pipe_ent;
    unsigned long long* pipe_ctr64;
    unsigned long long* rowkey;
    uint32_t rowkey_j;
};
int hm_launch_seed_init(hm_engine* e, const ScanArgs& a, hipStream_t s);
int hm_launch_row_key(hm_engine* e, int64_t row, int64_t n_partners, float sqrt_c, float thr, unsigned long long* key_dev, hipStream_t s);   // hm_loops.hip
int hm_launch_argmin_tail(hm_engine* e, const ScanArgs& a, float sqrt_c, float thr, ArgminRec* rec_out, bool with_seed,
                          int arm_rb, int arm_re, bool arm, const MergeFuse& mf, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// device helpers shared by the kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hm_img_spatial(const float* img, int RS, int64_t row, int s)
{
    return img[row * RS + hm_img_off(s)];
}
__device__ __forceinline__ float hm_img_time(const float* img, int RS, int64_t row) { return img[row * RS + RS - 4]; }

// canonical u (argument of acosh) between two image rows, one lane: products rounded separately, summed in
// torch's reduction order, then fl(fl(x0*y0) - S)  (DESIGN.md "Canonical arithmetic")
__device__ __forceinline__ float hm_img_u(const float* img, int RS, int d, int64_t a, int64_t b, int sign_mode)
{
    const float* ra = img + a * RS;
    const float* rb = img + b * RS;
    const float S = hm::torch_order_sum(
        [&](int s) {
            const int o = hm_img_off(s);
            return ra[o] * rb[o];
        },
        d);
    const float t = ra[RS - 4] * rb[RS - 4];
    const float m = t - S;
    return sign_mode ? m : -m;
}

// The same sum computed by the 32 lanes of a half-wave (lanes sharing lane >> 5), bit for bit: ATen's order is
// 32 independent chains -- accumulator k (0..3) x vector lane l (0..7), chain (k, l) owned here by lane t = 8k + l
// and fed elements t, t + 32, t + 64, ... -- followed by a fixed combine: leftover vectors into accumulator 0,
// accumulators 0 += 1, += 2, += 3, the scalar tail summed from zero, then the 8 vector lanes in order.
// prod(e) returns the e-th product (already rounded to fp32); every lane may be asked for any e < d.
// The result is returned on every lane of the half-wave.  All 64 lanes of the wave must call this together.
template <class PROD>
__device__ __forceinline__ float hm_halfwave_sum(int d, int lane, PROD prod)
{
    const int t = lane & 31, base = lane & 32;
    if (d < 8) {            // scalar form of row_sum: four interleaved accumulators (every lane on its own)
        float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int si = d >> 2;
        for (int i = 0; i < si; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ps[k] = ps[k] + prod(i * 4 + k);
        }
        for (int i = si * 4; i < d; ++i) ps[0] = ps[0] + prod(i);
        ps[0] = ps[0] + ps[1];
        ps[0] = ps[0] + ps[2];
        ps[0] = ps[0] + ps[3];
        return ps[0];
    }
    const int vec = d >> 3, ilp = vec >> 2;
    float p = 0.0f;
    for (int i = 0; i < ilp; ++i) p = p + prod(i * 32 + t);
    const int nleft = vec - ilp * 4;                       // 0..3 leftover vectors, all into accumulator 0
    float lt = 0.0f;
    if ((t >> 3) < nleft) lt = prod((ilp * 4 + (t >> 3)) * 8 + (t & 7));
    for (int q = 0; q < nleft; ++q) {
        const float v = __shfl(lt, base + (t & 7) + 8 * q, 64);
        if (t < 8) p = p + v;
    }
    const float c1 = __shfl(p, base + (t & 7) + 8, 64);
    const float c2 = __shfl(p, base + (t & 7) + 16, 64);
    const float c3 = __shfl(p, base + (t & 7) + 24, 64);
    const float r = ((p + c1) + c2) + c3;                  // meaningful on lanes t < 8
    const int ntail = d - vec * 8;
    float tt = 0.0f;
    if (t < ntail) tt = prod(vec * 8 + t);
    float acc = 0.0f;
    for (int q = 0; q < ntail; ++q) acc = acc + __shfl(tt, base + q, 64);
#pragma unroll
    for (int l = 0; l < 8; ++l) acc = acc + __shfl(r, base + l, 64);
    return acc;
}

// canonical u between image rows a and b by a half-wave (coalesced 128-byte reads of both rows)
__device__ __forceinline__ float hm_img_u_halfwave(const float* img, int RS, int d, int64_t a, int64_t b, int sign_mode, int lane)
{
    const float* ra = img + a * RS;
    const float* rb = img + b * RS;
    const float S = hm_halfwave_sum(d, lane, [&](int e) { const int o = hm_img_off(e); return ra[o] * rb[o]; });
    const float t = ra[RS - 4] * rb[RS - 4];
    const float m = t - S;
    return sign_mode ? m : -m;
}

// A half-wave works through HM_GATHER items per round: item k's canonical u is computed by the 32 lanes together
// (u_of(k) returns it on every lane; both half-waves of the wave call in step, each for its own item k) and parked in
// lane k.  The caller then finishes ITS item once -- acosh, threshold, key -- so the transcendental part is evaluated once
// per HM_GATHER items instead of once per item (every lane executes it either way).  Few items per round on purpose:
// the rounds of different half-waves are what hides the L2 latency of the row reads (32 per round measured 2.4x slower).
#define HM_GATHER 8
template <class UF>
__device__ __forceinline__ float hm_halfwave_gather(int lane, UF u_of)
{
    const int t = lane & 31;
    float mine = 0.0f;
#pragma unroll
    for (int k = 0; k < HM_GATHER; ++k) {
        const float u = u_of(k);
        mine = (t == k) ? u : mine;
    }
    return mine;
}

// How many ulps of u' above a key may still order BEFORE it by computed distance: d = acosh(u') / sqrt(c) is monotone in u' only
// up to the few-ulp error of acosh (2.5 ulps of d), so a pair at u' + s certainly has the larger d once the true increase
// s * ulp(u') / sqrt(u'^2 - 1) exceeds ~5 ulps of d.  For large u' (d ~ ln 2u') that takes up to 16 d ~ hundreds of ulps of u'
// -- HM_TIE_SLACK; near u' = 1 (d ~ sqrt(2 (u' - 1)), steep) a handful do: with t = u' - 1 < 1 the condition is s > ~20 t.
// 32 + 128 t is used below u' = 2.  (A constant 1024 made the exact top-1 search of a very dense table impossible: at d = 5,
// scale 0.01 there are ~10^5 pairs per ulp of u' -- round-3 fuzz.)  Non-decreasing in the bits.
__host__ __device__ __forceinline__ uint32_t hm_tie_slack(uint32_t ubits)
{
    if (ubits >= 0x40000000u) return HM_TIE_SLACK;
    return 32u + ((ubits > 0x3f800000u ? ubits - 0x3f800000u : 0u) >> 16);
}

// atomicMax on a word that MANY blocks raise towards the same value (the norm bounds): same-address device-scope atomics
// serialise at the memory side (~10 ns each: 1 500 of them cost more than the kernel around them), so a wave first reads the
// word (device scope, fresh) and only sends the atomic when it would raise it -- ~ln(blocks) atomics instead of one per block
__device__ __forceinline__ void hm_raise_bits(uint32_t* word, uint32_t bits)
{
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(word, bits);
}

__device__ __forceinline__ bool hm_key_less(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2)
{
    if (a0 != b0) return a0 < b0;
    if (a1 != b1) return a1 < b1;
    return a2 < b2;
}

// lexicographic min of (b0, b1, b2) over the block; result on every thread.  s0/s1/s2: blockDim.x words each.
__device__ __forceinline__ void hm_block_min_key(uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t* s0, uint32_t* s1, uint32_t* s2)
{
    // wave level first (shuffles), then one LDS round over the waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o0 = __shfl_xor(b0, off, 64), o1 = __shfl_xor(b1, off, 64), o2 = __shfl_xor(b2, off, 64);
        if (hm_key_less(o0, o1, o2, b0, b1, b2)) { b0 = o0; b1 = o1; b2 = o2; }
    }
    if (lane == 0) { s0[wv] = b0; s1[wv] = b1; s2[wv] = b2; }
    __syncthreads();
    if (wv == 0) {
        uint32_t c0 = lane < nw ? s0[lane] : 0xffffffffu, c1 = lane < nw ? s1[lane] : 0xffffffffu,
                 c2 = lane < nw ? s2[lane] : 0xffffffffu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o0 = __shfl_xor(c0, off, 64), o1 = __shfl_xor(c1, off, 64), o2 = __shfl_xor(c2, off, 64);
            if (hm_key_less(o0, o1, o2, c0, c1, c2)) { c0 = o0; c1 = o1; c2 = o2; }
        }
        if (lane == 0) { s0[0] = c0; s1[0] = c1; s2[0] = c2; }
    }
    __syncthreads();
    b0 = s0[0]; b1 = s1[0]; b2 = s2[0];
    __syncthreads();
}

// Bound on |u_f - u_c| between the MFMA prefilter value and the canonical value of the same pair
// (gamma_n bounds on both roundings, |terms| <= rmax2).  `kterms` = fp32 form: floats per image row;
// bf16 form: 16 * k-steps.  bf16 operands: round-to-nearest to 8 significant bits is a relative error of at most
// 2^-8 PER OPERAND, so each spatial product carries <= 2 * 2^-8 + 2^-16 and the sum is off by at most
// 0.0078128 ||x_s|| ||y_s|| <= 0.00782 * (largest squared spatial norm).  (Rounds 1-2 used 0.00392 -- the error of ONE
// operand: never exceeded by the many-term sums of d >= 24, exceeded at d = 1 by a fuzz case in round 3.)  The hi + lo
// split of the time coordinate: x0 = hi + lo exactly, lo is stored rounded (<= 2^-8 |lo| <= 2^-16 x0) and the lo * lo'
// term is dropped (<= 2^-16 x0 y0): <= 3 * 2^-16 * x0 * y0 <= 4.7e-5 * rmax2 (x0^2 <= rmax2).
__device__ __forceinline__ float hm_scan_delta(bool bf, int kterms, const uint32_t* rmax2_bits)
{
    const float rmax2 = hm::bitsf(rmax2_bits[0]);
    float delta = ((float)(kterms + 8) * 1.1920929e-07f) * rmax2 * 1.0001f;
    if (bf) delta += 0.00782f * hm::bitsf(rmax2_bits[1]) + 4.7e-5f * rmax2;
    return delta;
}

__device__ __forceinline__ uint32_t hm_pack_bf16(float lo, float hi)
{
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

// chunk c (K-slots 8c .. 8c+7) of a bf16 image row; the LAST FOUR slots of the row hold the time coordinate
// split as x0 ~ hi + lo: streamed (B) encoding [hi, lo, hi, 0]; the stationary (A) side rewrites its copy in
// registers to [-hi, -hi, -lo, 0], so the MFMA adds -(hi*hi' + hi*lo' + lo*hi') = -x0*y0 (1 + O(2^-16)).
__device__ __forceinline__ uint4 hm_bf16_chunk(const float* spatial /* x[1..d] */, float x0, int d, int KC, int c)
{
    float f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int sidx = 8 * c + q; f[q] = sidx < d ? spatial[sidx] : 0.0f; }
    if (c == KC - 1) {
        const __bf16 hb = (__bf16)x0;
        const float hi = (float)hb;
        const float lo = x0 - hi;
        f[4] = hi; f[5] = lo; f[6] = hi; f[7] = 0.0f;
    }
    return make_uint4(hm_pack_bf16(f[0], f[1]), hm_pack_bf16(f[2], f[3]), hm_pack_bf16(f[4], f[5]), hm_pack_bf16(f[6], f[7]));
}

__device__ __forceinline__ unsigned long long hm_wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t hm_wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// ---- hm_search.hip (host) ----
int hm_select_sorted(hm_engine* e, uint4* src, uint4* other, uint32_t m, uint32_t k, hipStream_t s);
void hm_partition_rows(int64_t n, int world, int rank, int64_t* r0, int64_t* r1);      // hm_comm.hip
int hm_topk_exact(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, int64_t n_limit,
                  int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s);      // hm_exact.hip
int hm_topk_core(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end, bool list_all, bool want_count,
                 int64_t n_limit, int64_t* n_valid_emitted, int64_t* count, uint4** result_dev, hipStream_t s);
