/*
 * hypmerge.h -- C ABI of libhypmerge.so, the MI355X (gfx950) merge engine for HypTokenizer.
 *
 * The reference (sangaprabhav/HypTokenizer) has no FFI / plugin interface on this path: the hot
 * path is plain Python calling PyTorch.  Each entry point below therefore
 * cites the reference PYTHON call site it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add at that site.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types in any signature.
 *   - `*_dev` pointers are device addresses owned by the caller (e.g. tensor.data_ptr());
 *     all other pointers are host memory owned by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Functions that return
 *     results in HOST memory synchronise that stream before returning; functions that only
 *     write device memory are asynchronous on it.
 *   - return value: 0 = HM_OK; negative = argument / capacity error (HM_E_*); positive = hipError_t.
 *     hm_last_error() returns a human-readable message for the last non-zero status.
 *   - table layout handed in by the caller is the reference's: row-major fp32 [n_rows, ld],
 *     column 0 = time coordinate, columns 1..d = spatial (tokenizer/hyperbolic_merge.py:145-153).
 *   - sign_mode: 0 = arithmetic of the reference as shipped (u = -minkowski_dot),
 *                1 = standard Lorentz sign (u = +minkowski_dot).
 *   - candidate order everywhere: ascending fp32 distance, ties by row-major (i, j)
 *     (stable sort of the nonzero() list, hyperbolic_merge.py:263-269,378; fast...:349-355,371).
 *   - every distance is d = acosh(max(u,1)) / sqrt(c) evaluated with the canonical fp32
 *     arithmetic of DESIGN.md; thresholds compare in fp32 (d < thr, NaN never passes).
 */
#ifndef HYPMERGE_H
#define HYPMERGE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_ABI_VERSION 3

#define HM_OK            0
#define HM_E_ARG        (-1)   /* bad argument (null pointer, range, unsupported dimension) */
#define HM_E_CAPACITY   (-2)   /* an internal workspace was too small for this request      */
#define HM_E_STATE      (-3)   /* call order violated (e.g. scan before hm_set_table)        */
#define HM_E_NOMEM      (-4)
#define HM_E_COMM       (-5)   /* RCCL could not be loaded / a collective failed            */
#define HM_E_NA         (-6)   /* the request does not apply to the engine's current state: an ordinary answer, no message */

#define HM_SIGN_REFERENCE 0
#define HM_SIGN_LORENTZ   1

/* form of the pair scan's MFMA prefilter (results are identical: every reported distance is re-evaluated in the
 * canonical fp32 arithmetic): AUTO = bf16 MFMA from d >= 24, else fp32 MFMA */
#define HM_PREFILTER_AUTO 0
#define HM_PREFILTER_F32  1
#define HM_PREFILTER_BF16 2

typedef struct hm_engine hm_engine;

int         hm_abi_version(void);
const char* hm_last_error(const hm_engine* e);          /* e may be NULL: last global error */

/* One engine per device.  Allocates the scan image ([max_rows] rows) and all workspaces; nothing
 * is allocated in the per-call path afterwards.
 * LIMITS (narrower than the reference, which takes any max_vocab_size; both are rejected here with HM_E_ARG, by the Python
 * classes at construction with a ValueError):
 *   2 <= max_rows <= 131072   -- a row index has 17 bits in the 32-bit (i, j) word of the pair scan's 64-bit running key;
 *   2 <= d1 = d + 1 <= 129    -- kernels are instantiated for d <= 128; the bf16 prefilter exists for d <= 124 (d + 4 K-slots
 *                                in at most 16 chunks of 8), wider tables use the fp32 prefilter.
 * Replaces: the pre-allocated table of HyperbolicTokenizer.__init__ (hyperbolic_merge.py:144-153)
 * as far as the search kernels are concerned, and the FAISS index objects
 * (_init_faiss_index :593-605, _build_faiss_index fast...:195-240), which are not used at all. */
int hm_engine_create(hm_engine** out, int device, int64_t max_rows, int d1, int sign_mode, int prefilter);
int hm_engine_destroy(hm_engine* e);
/* Change the prefilter form of a live engine (HM_PREFILTER_*).  The environment variable HM_SCAN_PRECISION
 * ("f32" | "bf16") overrides the argument of hm_engine_create, not this call. */
int hm_set_prefilter(hm_engine* e, int prefilter);

/* (Re)build the scan image from rows [0, n_rows) of the caller's table.  Replaces nothing in the
 * reference (it re-reads self.embeddings[:n] every step, hyperbolic_merge.py:250). */
int hm_set_table(hm_engine* e, const float* X_dev, int64_t ld, int64_t n_rows, void* stream);
/* Refresh image rows [row_begin, row_end) after the caller changed those table rows. */
int hm_update_rows(hm_engine* e, const float* X_dev, int64_t ld, int64_t row_begin, int64_t row_end,
                   void* stream);
int64_t hm_rows(const hm_engine* e);                     /* live rows in the image */

/* K1: nearest pair.  Over all pairs i<j<n with row_begin <= i < row_end: the smallest (d, i, j)
 * with d < thr.  *found = 0 when no pair qualifies.  Results in host memory.
 * Replaces: _find_merge_candidates + sort + [0] of HyperbolicTokenizer.optimize_merges
 * (hyperbolic_merge.py:371-396, candidate search :247-269). */
int hm_pairwise_argmin(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end,
                       float* d, int32_t* i, int32_t* j, int32_t* found, void* stream);

/* K1, device-resident form for the multi-GPU exchange: same search, but the 16-byte record
 * {found, bits(d), i, j} (uint32 x 4; found = 2 means "emission buffer overflowed, call the host
 * form") is written to rec_dev and nothing is synchronised -- the record can feed an RCCL
 * all-gather on the same stream directly. */
int hm_pairwise_argmin_dev(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end,
                           uint32_t* rec_dev, void* stream);

/* K2: the k smallest candidates in order, and the exact number of candidates.
 * d_out/i_out/j_out have room for k entries (may be NULL when k == 0); *n_out = min(k, *count).
 * Replaces: the recompute branch of _find_merge_candidates_fast + candidates.sort() +
 * AdaptiveMergeCache.add_batch truncation to max_size (fast_hyperbolic_merge.py:336-355,371-374,78-95).
 * Always answers for k <= 65536: a table whose distances are so concentrated that no emission cut of the matrix-core
 * prefilter fits the engine's buffers (all u within a few hundred ulps of 1) is searched by evaluating every pair in the
 * canonical arithmetic and selecting by counting (hm_exact.hip) -- like the reference, only slower than the usual path
 * (56 ms at 25 000 rows).  The same holds for hm_pairwise_argmin, hm_pairwise_topk_nocount and hm_pairwise_count. */
int hm_pairwise_topk(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end,
                     float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count,
                     void* stream);

/* K2 without the exact total: the same ordered k smallest candidates; *count is the exact number of candidates
 * when fewer than k exist, else -1 ("at least k, not counted" -- the scan then visits only what lies below its
 * emission cut; hm_pairwise_count delivers the number when a caller asks for it).  The reference consumes the
 * total only in log lines (fast_hyperbolic_merge.py:521,526) and for emptiness (:529). */
int hm_pairwise_topk_nocount(hm_engine* e, float c, float thr, int64_t k, int64_t row_begin, int64_t row_end,
                             float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out, int64_t* count, void* stream);
/* The same refresh in two halves for callers that have host work to do meanwhile (the fast tokenizer's string
 * bookkeeping): hm_topk_refresh_begin enqueues the whole chain and returns at once -- HM_E_NA when
 * the refresh is not of the incremental kind (rows changed, other k / curvature, lower threshold: use
 * hm_pairwise_topk_nocount; HM_E_STATE stays a real error: a refresh already pending) --, hm_topk_refresh_end waits and delivers the ordered list (HM_E_CAPACITY: more new entries
 * than the device-side sort takes; the state is untouched, run hm_pairwise_topk_nocount).  No other call on the engine in
 * between. */
int hm_topk_refresh_begin(hm_engine* e, float c, float thr, int64_t k, void* stream);
int hm_topk_refresh_end(hm_engine* e, float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out);
/* Exact number of candidates among the first n_limit rows (n_limit < 0: all live rows).  Rows are only ever
 * appended, so this is len(candidates) of the search that ran when the table had n_limit rows. */
int hm_pairwise_count(hm_engine* e, float c, float thr, int64_t n_limit, int64_t* count, void* stream);

/* All candidates (unordered) -- the caller sorts them row-major.  At most cap triples are written;
 * *total is the exact number.  Replaces: the candidate list of _find_merge_candidates
 * (hyperbolic_merge.py:247-269) when a caller really wants every tuple. */
int hm_pairwise_candidates(hm_engine* e, float c, float thr, int64_t row_begin, int64_t row_end,
                           int64_t cap, int32_t* i_out, int32_t* j_out, float* d_out, int64_t* total,
                           void* stream);

/* K3: distances from image row `row` to image rows [0, n): d_out_dev[n] (device).
 * No reference equivalent (incremental maintenance). */
int hm_row_vs_all(hm_engine* e, int64_t row, int64_t n, float c, float* d_out_dev, void* stream);

/* K3 + reduction: the nearest partner of image row `row` among rows [0, n_partners) (itself
 * excluded): smallest (d, min(i,row), max(i,row)) with d < thr, in host memory.  Lets a caller
 * maintain the global nearest pair incrementally: rows are only ever appended, so after
 * a merge the global minimum is min(previous minimum, nearest partner of the new row).
 * No reference equivalent (the reference recomputes everything, hyperbolic_merge.py:247-269). */
int hm_row_argmin(hm_engine* e, int64_t row, int64_t n_partners, float c, float thr, float* d, int32_t* i,
                  int32_t* j, int32_t* found, void* stream);

/* K5: gathered pair distances on the image: out_dev[t] = d(row I[t], row J[t]).
 * Replaces: distance(...).item() loops (_compute_distance_statistics fast...:448-455,
 * n<=100 branch hyperbolic_merge.py:270-289). */
int hm_pair_distance(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, int64_t b, float c,
                     float* out_dev, void* stream);

/* K4: batched "midpoint" project(exp_map(x_i, w * log_map(x_i, x_j))) on the image rows,
 * out_dev[b, d1] in the reference's column order.
 * Replaces: _merge_tokens arithmetic (hyperbolic_merge.py:326-340) and the loop of
 * _evaluate_candidates_parallel (:568-587). */
int hm_midpoint_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev,
                      int64_t b, float c, float* out_dev, void* stream);

/* Fused merge step: midpoint of image rows (i, j) with weight w is written to row `new_row` of the
 * caller's table AND of the image; the live-row count becomes max(rows, new_row + 1).
 * Replaces: hyperbolic_merge.py:326-351 (log_map, scale, exp_map, project, embeddings.data[n] = x). */
int hm_merge_append(hm_engine* e, int32_t i, int32_t j, float w, float c, float* X_dev, int64_t ld,
                    int64_t new_row, void* stream);

/* Several merges known in advance, one launch: merge t = midpoint of image rows (I[t], J[t]) with weight W[t] ->
 * row first_row + t of the table and of the image.  independent = 0: a merge may read rows written by earlier merges
 * of the batch (sequential chain); independent = 1: the caller guarantees every I[t], J[t] < first_row (all at once).
 * Replaces: the hyperbolic_merge.py:326-351 arithmetic of the ~100 merges a FastHyperbolicTokenizer performs
 * between two refreshes (fast_hyperbolic_merge.py:546-549), all of which are known when the refresh returns. */
int hm_merge_append_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev, int64_t count,
                          float c, float* X_dev, int64_t ld, int64_t first_row, int independent, void* stream);
/* The same with I / J / W in HOST memory (at most 4096 merges): staged through the engine's pinned buffer. */
int hm_merge_append_batch_host(hm_engine* e, const int32_t* I_host, const int32_t* J_host, const float* W_host, int64_t count,
                               float c, float* X_dev, int64_t ld, int64_t first_row, int independent, void* stream);
/* Forget image rows >= n_rows (undo rows appended ahead of time). */
int hm_truncate(hm_engine* e, int64_t n_rows, void* stream);

/* ---- device-resident merge loops ------------------------------------------------------------------------
 * Token lengths len(vocab[r]) for rows [0, n): the only thing a merge needs from the token strings
 * (w = len(tj) / (len(ti) + len(tj)), hyperbolic_merge.py:317-323).  Host array; kept on the device and
 * extended by the loops below (len[new] = len[i] + len[j]). */
int hm_set_token_lengths(hm_engine* e, const int32_t* lens_host, int64_t n, void* stream);
/* `steps` (<= 256) iterations of HyperbolicTokenizer.optimize_merges (hyperbolic_merge.py:371-399: full search,
 * sort, [0], merge) enqueued back to back -- pair scan + one tail kernel per step, no host round trip -- and read
 * back with one synchronisation.  rec_out[4 * k] = {found, bits(d), i, j} of step k: found 1 = merged (i, j) into
 * row n + k; 0 = no candidate (the loop ends, hyperbolic_merge.py:373-375); 2 = emission overflow at this step (run it
 * through hm_pairwise_argmin + hm_merge_append); 3 = skipped after a 0 / 2.  *done = leading merged steps. */
int hm_std_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* rec_out,
                       int64_t* done, void* stream);
/* The same merges with the nearest pair maintained as a running minimum (rows are only appended):
 * one launch per step (merge + new row vs all rows + fold).  best_io = {found, bits(d), i, j}: in: the nearest
 * pair of the current table (from hm_pairwise_argmin); out: the running minimum after the last executed step. */
int hm_incr_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* best_io,
                        uint32_t* rec_out, int64_t* done, void* stream);

/* The same loop row-sharded over ranks, without a host round trip per step.  Between
 * hm_shard_loop_begin and hm_shard_loop_end the caller enqueues, per step: hm_pairwise_argmin_dev over the rank's row
 * range -> an all-gather of the ranks' 16-byte records on the same stream (RCCL) -> hm_shard_merge_step, which takes the
 * lexicographic minimum of the `world` (<= 64) gathered records (identical on every rank) and applies the merge to this
 * rank's replica (row = current row count, weights from the token lengths), or ends the loop on the device: no pair
 * anywhere (record found = 0), an emission overflow on some rank (found = 2: the caller runs that step through the
 * bounded host path); later steps of the batch then skip themselves (found = 3).  hm_shard_loop_end synchronises once
 * and returns the step records as hm_std_merge_steps does. */
int hm_shard_loop_begin(hm_engine* e, void* stream);
int hm_shard_merge_step(hm_engine* e, const uint32_t* recs_dev, int world, float c, float* X_dev, int64_t ld, int64_t step,
                        void* stream);
int hm_shard_loop_end(hm_engine* e, int64_t steps, uint32_t* rec_out, int64_t* done, void* stream);

/* ---- the exchange step inside the library (hm_comm_init, hm_global_argmin, hm_global_topk) ----
 * One process per GPU; every rank holds a replica of the table and an engine of its own.  hm_comm_unique_id (one rank)
 * produces the 128-byte RCCL id the host program hands to the other ranks by its own means (torch.distributed broadcast,
 * MPI, a file); hm_comm_init binds an RCCL communicator (over xGMI on one node) to the engine; collective calls below must
 * then be made by every rank with the same arguments.  librccl.so.1 is bound at run time (the copy already loaded by the
 * process when there is one): HM_E_COMM when it is missing.  Rows are cut into `world` ranges of equal pair count.
 * Replaces: nothing in the reference (single process); the loop sharded is hyperbolic_merge.py:357-412. */
#define HM_COMM_ID_BYTES 128
int hm_comm_unique_id(void* id_out128);
int hm_comm_init(hm_engine* e, const void* id128, int rank, int world);
int hm_comm_destroy(hm_engine* e);
int hm_comm_info(const hm_engine* e, int* rank, int* world);          /* rank = -1, world = 0 without a communicator */
/* `steps` (<= 256) iterations of the standard loop, row-sharded, enqueued from the library with no host code per step:
 * scan of this rank's rows -> record -> ncclAllGather (16 bytes per rank) -> global minimum + merge into this rank's
 * replica.  Records / *done as hm_std_merge_steps (identical on every rank). */
int hm_shard_merge_steps(hm_engine* e, float c, float thr, float* X_dev, int64_t ld, int64_t steps, uint32_t* rec_out,
                         int64_t* done, void* stream);
/* C1: the global nearest pair (search of this rank's rows, all-gather of the records, minimum).  Host results. */
int hm_global_argmin(hm_engine* e, float c, float thr, float* d, int32_t* i, int32_t* j, int32_t* found, void* stream);
/* C2: the k smallest candidates of the whole table in order and their exact number: every rank's ordered list of its rows
 * stays on the device, the lists are all-gathered and merged there by the exact selection (no host round trip of the
 * lists).  Arguments as hm_pairwise_topk without the row range. */
int hm_global_topk(hm_engine* e, float c, float thr, int64_t k, float* d_out, int32_t* i_out, int32_t* j_out, int64_t* n_out,
                   int64_t* count, void* stream);

/* Measurement aid: with on != 0, hm_std_merge_steps records a HIP event pair around every scan launch of a batch (all of
 * them enter hm_scan_totals) and around the whole batch; hm_last_loop_timing returns the last batch's wall time on the
 * device, the sum of its scan launches and its step count (0 when the batch stopped early). */
int hm_debug_time_loops(hm_engine* e, int on);
int hm_last_loop_timing(hm_engine* e, float* batch_ms, float* scan_ms, int64_t* steps);

/* ---- enhanced tokenizer (BASELINE config 5) ---------------------------------------------------------------
 * Semantic-coherence distances: for candidate t the simulated merged embedding
 * m = exp_map(x_I[t], W[t] * log_map(x_I[t], x_J[t])) -- NOT projected -- and out_dev[t * ns + s] =
 * distance(m, x_S[t * ns + s]).  Sampling, the skip of s in {i, j}, mean and sigmoid stay with the caller.
 * Replaces: the loop of tokenizer/enhanced_fast_hyperbolic_merge.py:308-333 (one midpoint + <= 50
 * distance().item() calls per candidate). */
int hm_coherence_batch(hm_engine* e, const int32_t* I_dev, const int32_t* J_dev, const float* W_dev, const int32_t* S_dev,
                       int64_t b, int ns, float c, float* out_dev, void* stream);
/* project_to_hyperboloid over rows [0, n_rows) of the caller's table IN PLACE (only column 0 changes) and the
 * matching refresh of the engine's images and norm bounds for the live rows.  n_rows >= live rows.
 * Replaces: _project_embeddings (enhanced_fast_hyperbolic_merge.py:784-792) and the constructor's :243-244. */
int hm_project_table(hm_engine* e, float* X_dev, int64_t ld, int64_t n_rows, float c, void* stream);

/* Host-only helper (no GPU work): out[t * ns + q] = torch.randperm(n)[q] for t = 0..count-1 and q < ns, drawn from
 * the MT19937 state of torch's CPU generator -- mt_state[624] words, *left, *next as in at::mt19937 -- which is
 * advanced exactly as `count` calls of torch.randperm(n) would advance it (n < 2^32 / 20, ns <= 4096).
 * Replaces: torch.randperm(current_vocab_size)[:sample_size] per scored candidate
 * (enhanced_fast_hyperbolic_merge.py:324-325): O(n) draws of the recurrence instead of an O(n) random-access shuffle. */
int hm_randperm_prefix(uint32_t* mt_state, int32_t* left, uint32_t* next, int64_t n, int32_t ns, int64_t count, int32_t* out);

/* ---- batch tokenizer ---------------------------------------------------------------
 * HyperbolicTokenizer.tokenize for many lines at once, over 32-bit symbols: the caller maps every distinct
 * string (characters, rule operands, rule results) to a symbol >= 0 and any other character to a NEGATIVE
 * symbol (never matches a rule, passes through).  Semantics are the reference's positional fixed point:
 * repeated left-to-right passes, a hit replaces tokens[i] by the rule's result, removes tokens[i+1] and stays
 * at i, until a pass changes nothing.
 * Replaces: tokenizer/hyperbolic_merge.py:414-446 (and with it encode :448-459) as driven per line by
 * scripts/benchmark_efficiency.py:58-94.
 *
 * Symbols of rules lie in [0, 2^21 - 1) (two million distinct strings).
 *
 * hm_tokenize_table_capacity / hm_tokenize_build_table are host-only: they build the open-addressing rule table
 * {(left, right) -> merged} as `capacity` 8-byte entries left:21 | right:21 | merged+1:22 in buckets of two (a later
 * rule for the same pair replaces the earlier one, as dict assignment does, :425-428); the caller copies the
 * `capacity` words to 16-byte-aligned device memory. */
int64_t hm_tokenize_table_capacity(int64_t n_rules);
int hm_tokenize_build_table(const int32_t* left, const int32_t* right, const int32_t* merged, int64_t n_rules,
                            uint64_t* table_out, int64_t capacity);
/* sym_dev: symbols of all lines concatenated (each line < 2^31 symbols); offsets_dev[n_lines + 1]; order_dev:
 * optional permutation of the lines (lane t handles line order_dev[t]; longest-first keeps the 64 lines of a wave
 * alike), may be NULL.  Line l's tokens are written to out_dev[offsets[l] .. offsets[l] + out_len_dev[l]);
 * passes_dev (optional) receives the number of passes the reference's while-loop runs.  Asynchronous on `stream`. */
int hm_tokenize_batch(const int32_t* sym_dev, const int64_t* offsets_dev, const int64_t* order_dev, int64_t n_lines,
                      const uint64_t* table_dev, int64_t capacity, int32_t* out_dev, int32_t* out_len_dev,
                      int32_t* passes_dev, void* stream);

/* Dense distance block between two arbitrary device arrays: out_dev[n1, n2].
 * Replaces: batch_distance / batch_distance_optimized (embedding/lorentz_model.py:141-210) and
 * _compute_pairwise_distances (hyperbolic_merge.py:166-190).  Engine-independent. */
int hm_batch_distance(const float* X_dev, int64_t n1, const float* Y_dev, int64_t n2, int64_t ld_x,
                      int64_t ld_y, int d1, float c, int sign_mode, float* out_dev, void* stream);

/* Row-wise Lorentz primitives on device arrays [b, d1] with leading dimension ld
 * (embedding/lorentz_model.py).  Engine-independent.
 *   hm_rows_minkowski : out[b]      = minkowski_dot(x, y)            (:14-25, sign_mode applies)
 *   hm_rows_distance  : out[b]      = distance(x, y, c)              (:122-138)
 *   hm_rows_log_map   : out[b, d1]  = log_map(x, y)                  (:96-119)
 *   hm_rows_exp_map   : out[b, d1]  = exp_map(x, v)                  (:73-93)
 *   hm_rows_project   : out[b, d1]  = project_to_hyperboloid(x, c)   (:41-56)           */
int hm_rows_minkowski(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, int sign_mode,
                      float* out_dev, void* stream);
int hm_rows_distance(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, float c,
                     int sign_mode, float* out_dev, void* stream);
int hm_rows_log_map(const float* x_dev, const float* y_dev, int64_t b, int64_t ld, int d1, int sign_mode,
                    float* out_dev, int64_t ld_out, void* stream);
int hm_rows_exp_map(const float* x_dev, const float* v_dev, int64_t b, int64_t ld, int d1, float* out_dev,
                    int64_t ld_out, void* stream);
int hm_rows_project(const float* x_dev, int64_t b, int64_t ld, int d1, float c, float* out_dev,
                    int64_t ld_out, void* stream);

/* Timing of the last scan launched by hm_pairwise_argmin / hm_pairwise_topk on this engine,
 * measured with HIP events on the stream the kernel ran on (bench.py roofline).
 * *scan_ms = duration of the dominant pair-scan kernel launch(es); *pairs = pairs it covered. */
int hm_last_scan_stats(const hm_engine* e, float* scan_ms, int64_t* pairs, int64_t* emitted, int32_t* passes);

/* Running totals over every pair-scan launch (argmin / top-k modes, not the sampled estimate
 * passes) since engine creation or the last reset: summed event-timed kernel duration, pairs
 * covered and number of launches.  reset != 0 clears the totals after reading. */
int hm_scan_totals(hm_engine* e, double* scan_ms, int64_t* pairs, int64_t* launches, int reset);

/* Test hook: pretend the previous refresh ended on this emission cut (bits of u'); the next whole-table top-k
 * search starts from it as given and has to notice by itself when it is too tight. */
int hm_debug_force_cut(hm_engine* e, uint32_t cut_bits, int64_t k, float c);

/* Test / tuning hook for the pair scan's work decomposition (results never depend on it): "big_rows" (512-row blocks for
 * launches covering at least the pairs of this many rows), "chunk", "tail", "tail_div", "shape", "incr_topk", "phases" / "ph_share0..4" / "ph_div1..5" (item list of the scan), "dyn" (0: one block per item instead of the resident grid
 * that draws its items from a device counter) and "dyn_slots" (size of that resident grid; 0 = what the device holds), "pipeline" (0: the standard loop
 * strictly sequential), "pipeline_pairs", "pipe_fault_at", "exact_search" (1: every top-k / count through the prefilter-free
 * exact path that is otherwise the last resort of a search whose survivors fit no emission cut), "kc_even" (default knob only:
 * bf16 image rows padded to whole 16-slot k-steps).  hm_debug_set_default_knob applies to every
 * engine created afterwards in this process (clear != 0 removes the default `name`, or all of them when name is NULL / "").
 * The shipped library reads no environment variable for these; tuning builds (-DHM_TUNING) also accept HM_TUNE_<NAME>. */
int hm_debug_set_knob(hm_engine* e, const char* name, double value);
int hm_debug_set_default_knob(const char* name, double value, int clear);

#ifdef __cplusplus
}
#endif
#endif /* HYPMERGE_H */
