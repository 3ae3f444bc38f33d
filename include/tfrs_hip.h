/*
 * tfrs_hip.h -- C ABI of libtfrs_hip.so, the MI355X (gfx950) implementation of the
 * tensorflow/recommenders retrieval hot path.
 *
 * The reference's boundary for this path is a Python class API (Keras layers), not
 * an FFI; each entry point below names the reference method whose arithmetic it
 * replaces (paths relative to tensorflow_recommenders/).  The Python host classes in
 * recommenders_amd/ bind these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _h;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only
 *     enqueue work, they never synchronise the device or allocate in launch paths:
 *     the caller supplies outputs and a workspace sized by the matching
 *     *_workspace_bytes() query (index handles own their packed corpus);
 *   - return value: 0 = OK, <0 = error (TFRS_E*); the message is available through
 *     tfrs_last_error() (thread-local);
 *   - float32 arithmetic throughout; candidate indices are int32 row numbers
 *     (the reference's default identifiers are an int32 range/counter,
 *     layers/factorized_top_k.py:380-382,544-545); identifier lookup stays on the host side;
 *   - functions are re-entrant on distinct streams/workspaces; an index handle is
 *     read-only while queries run.
 *
 * Numerics contract: a score is ONE float32 fma chain over d = 0..D-1 in increasing d
 * starting from +0 (what v_mfma_f32_32x32x2_f32 computes); top-K order is score
 * descending, ties to the lower row index (tf.math.top_k); results are therefore
 * reproducible bit for bit and comparable with == against oracle/.
 */
#ifndef TFRS_HIP_H_
#define TFRS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFRS_OK 0
#define TFRS_EINVAL (-1)   /* bad argument (shape, k, alignment, NULL) */
#define TFRS_ENOTIMPL (-2) /* valid request outside the implemented envelope */
#define TFRS_EHIP (-3)     /* a HIP runtime call failed */
#define TFRS_ENOMEM (-4)   /* workspace too small / allocation failed */
#define TFRS_ESTATE (-5)   /* handle not indexed yet */

#define TFRS_MAX_DIM 128 /* embedding dims above this are TFRS_ENOTIMPL for top-K */
#define TFRS_MAX_K 1024

int tfrs_version(void);
const char *tfrs_last_error(void);
/* Fills compute-unit count, LDS bytes per CU and the gcnArchName of device `dev`. */
int tfrs_device_info(int dev, int *cu_count_h, int *lds_bytes_h, char *arch_h,
                     int arch_len);
/* The library's configuration plane: every measurement / test switch (the TFRS_* names listed in
 * INTEGRATION.md "Runtime switches") is read through one accessor -- a value set here wins, else
 * the process environment is consulted at the time of the call.  value == NULL removes the
 * override.  tfrs_get_option returns 1 and the value in force, 0 when the option is unset.
 * Process-wide (like the environment); defaults are what bench.py and the tests run. */
int tfrs_set_option(const char *name, const char *value);
int tfrs_get_option(const char *name, char *value_h, int value_len);


/* Measurement hook (bench.py): while enabled, every launch of the fused score+filter scan
 * kernel is bracketed by HIP events on its launch stream.  tfrs_profile_read returns the
 * summed kernel time, the number of launches and their algorithmic flop (2*nq*rows*d) since
 * the last read, after waiting for the recorded events. */
int tfrs_profile_enable(int on);
int tfrs_profile_read(double *scan_ms_h, int *launches_h, double *flop_h);
/* Same sums restricted to one scan kernel (0 = exact f32 scan, 1 = fp16 filter pass over all
 * rows, 2 = fp16 threshold pass over the sampled stages); does not reset -- call before
 * tfrs_profile_read. */
int tfrs_profile_read_kind(int kind, double *scan_ms_h, int *launches_h, double *flop_h);

/* Per-box ceilings measured in the caller's process (bench.py `roofline.measured_ceiling`; no reference
 * counterpart -- measurement plumbing).  Boxes of one pool differ by several per cent in what any kernel
 * gets (the shader clock follows the power the operand data draws), so a fraction of the spec peak is only
 * comparable between runs next to the box's own ceiling.  Both calls time with HIP events on `stream` and WAIT.
 *   tfrs_calibrate_mfma_f16: saturating v_mfma_f32_32x32x16_f16 loop on uniform random fp16 operands in
 *     registers, two waves per SIMD on every CU, `iters` x 64 MFMAs per wave; returns TFLOP/s and the shader
 *     clock (MHz) the loop ran at.  `ws`: tfrs_calibrate_workspace_bytes() bytes.
 *   tfrs_calibrate_copy: float4 grid-stride copy of ws_bytes / 2 bytes from the first half of `ws` to the
 *     second (>= 64 MiB, 16-byte aligned), `iters` times; returns (bytes read + bytes written) / time in GB/s. */
size_t tfrs_calibrate_workspace_bytes(void);
int tfrs_calibrate_mfma_f16(void *ws, size_t ws_bytes, int iters, double *tflops_h, double *shader_mhz_h,
                            void *stream);
int tfrs_calibrate_copy(void *ws, size_t ws_bytes, int iters, double *gbs_h, void *stream);

/* ------------------------------------------------------------------------- *
 * Candidate index (BruteForce.index, layers/factorized_top_k.py:540-584).
 * The handle owns a device copy of the candidates in an MFMA/LDS-friendly packed
 * layout (even/odd feature planes per row, rows of 4 * padded_dim bytes with NO pad
 * slot in memory -- a dim-64 row is two whole 128-byte lines; the scan kernels
 * re-pitch rows to an odd number of 16-byte slots when they stage them in LDS, so
 * that ds_read_b128 of 16 consecutive rows is bank-conflict free).  Re-indexing
 * drops and recreates the copy, as the reference does (:163-164).
 * ------------------------------------------------------------------------- */
typedef struct tfrs_index tfrs_index_t;

int tfrs_index_create(tfrs_index_t **out_h);
int tfrs_index_destroy(tfrs_index_t *index);
/* Copies+packs candidates[n, d] (row-major f32).  May (re)allocate device memory. */
int tfrs_index_set(tfrs_index_t *index, const float *candidates, int64_t n, int d,
                   void *stream);
/* Incremental ingestion for TopK.index_from_dataset (:179-215): reserve once, then
 * append blocks in dataset order. */
int tfrs_index_reserve(tfrs_index_t *index, int64_t capacity, int d, void *stream);
int tfrs_index_append(tfrs_index_t *index, const float *block, int64_t nb, void *stream);
/* Non-finite inputs (the reference's tf.math.top_k, layers/factorized_top_k.py:605, tolerates NaN / Inf scores; this
 * library's fp16-prefiltered search does NOT: its error bound is built from row norms, and the filter kernels are
 * compiled without NaN semantics).  CONTRACT: candidates and queries must be finite, |x| < 1.8e19 (so that a squared
 * norm stays finite).  Violations are recorded, never silent:
 *   bit 0 (1)  an indexed candidate row holds NaN / Inf -- set by tfrs_index_set / _append; the host layer raises
 *              ValueError from BruteForce.index / index_from_dataset and drops the index;
 *   bit 1 (2)  a query row of a tfrs_bruteforce_topk[_below] call held NaN / Inf.  Only THAT row of the result is
 *              affected (its scores are non-finite, its indices valid but unspecified); every other row of the call is
 *              the exact top-K as always.  The host layer raises ValueError at the next call (or at once under
 *              BruteForce(check_finite=True), which synchronises).
 * The word lives in pinned host memory the kernels OR into: reading it never touches the stream, and reflects
 * every launch that has completed.  reset_mask: bits to clear after the read. */
int tfrs_index_nonfinite(const tfrs_index_t *index, int reset_mask, int32_t *flags_h);
/* ORs `bits` into the handle's flag word when any element of x[0, count) or y[0, county) (either may be NULL / empty) is
 * NaN / Inf: ONE launch, no synchronisation.  For the searches that have no indexed corpus -- Streaming over blocks read
 * in place (layers/factorized_top_k.py:404-509) records its queries and its carried state through an otherwise empty
 * handle; the flag word is allocated on first use. */
int tfrs_index_note_nonfinite(tfrs_index_t *index, const float *x, int64_t count, const float *y, int64_t county,
                              int bits, void *stream);
int64_t tfrs_index_size(const tfrs_index_t *index);
int tfrs_index_dim(const tfrs_index_t *index);
/* Writes the original row-major candidates[n, d] back (checkpoint/state_dict). */
int tfrs_index_unpack(const tfrs_index_t *index, float *candidates_out, void *stream);

/* ------------------------------------------------------------------------- *
 * BruteForce.call (layers/factorized_top_k.py:586-607; scores :320-333):
 *   scores = q @ candidates^T ; values, indices = tf.math.top_k(scores, k)
 * out_scores[nq, k] f32, out_idx[nq, k] i32 (row numbers; identifiers[idx] is a
 * host-side gather, :607).  Requires k <= index size (TopKV2 raises otherwise).
 * The [nq, n] score matrix is never materialised.
 * ------------------------------------------------------------------------- */
size_t tfrs_bruteforce_topk_workspace_bytes(int64_t nq, int64_t n, int d, int k);
int tfrs_bruteforce_topk(const tfrs_index_t *index, const float *queries, int64_t nq,
                         int k, float *out_scores, int32_t *out_idx, void *workspace,
                         size_t workspace_bytes, void *stream);

/* Measurement hook (bench.py, tests): after a tfrs_bruteforce_topk call with the same
 * (workspace, nq, n, k) has completed on `stream`, returns how many of its queries took the
 * exact-recompute ("redo") path of the fp16-prefiltered search (survivor list overflow or
 * retained set too large; 0 on well-behaved data, and always 0 on the all-f32 path).
 * Synchronises `stream`. */
/* De-duplicated index (BruteForce.call on corpora with many EXACT copies of a row; tf.math.top_k
 * breaks ties by the lower index, layers/factorized_top_k.py:605, so every copy of a top-K row is a
 * candidate and no threshold separates them).  The host indexes the distinct rows:
 *   tfrs_row_hash64: out[r] = 63-bit hash of row r's bit pattern (equal rows hash equal; the caller
 *     compares hash neighbours exactly);
 *   tfrs_topk_expand_duplicates: scores / distinct_rows[nq, k_in] = the best distinct rows of each
 *     query (descending; row < 0 = empty); distinct row u stands for the original rows
 *     dup_rows[dup_start[u] .. dup_start[u + 1]) (ascending).  Writes the exact top-k_out of the
 *     ORIGINAL corpus, order (score descending, original row ascending); empty slots carry row -1. */
int tfrs_row_hash64(const float *rows, int64_t n, int d, uint64_t *out, void *stream);
int tfrs_topk_expand_duplicates(const float *scores, const int32_t *distinct_rows, int64_t nq, int k_in,
                                const int64_t *dup_start, const int32_t *dup_rows, int k_out,
                                float *out_scores, int32_t *out_rows, void *stream);

/* Paged search for k beyond TFRS_MAX_K (tf.math.top_k has no limit, layers/factorized_top_k.py:605):
 * the best k <= TFRS_MAX_K rows among those strictly AFTER (last_scores[q * last_ld],
 * last_rows[q * last_ld]) in the result order (score descending, row ascending); NULL / NULL = the
 * first page.  Concatenated pages are exactly the sorted top-(sum of k); the caller keeps
 * sum of k <= rows.  All-f32 rounds.  Workspace: tfrs_bruteforce_topk_below_workspace_bytes. */
size_t tfrs_bruteforce_topk_below_workspace_bytes(int64_t nq, int64_t n, int d, int k);
int tfrs_bruteforce_topk_below(const tfrs_index_t *index, const float *queries, int64_t nq, int k,
                               const float *last_scores, const int32_t *last_rows, int64_t last_ld,
                               float *out_scores, int32_t *out_idx, void *workspace,
                               size_t workspace_bytes, void *stream);
int tfrs_bruteforce_topk_redo_count(const void *workspace, int64_t nq, int64_t n, int k,
                                    int32_t *redo_count_h, void *stream);
/* Same contract; reasons_h[4] = queries flagged because {0: a survivor segment or the list
 * overflowed, 1: the statistically chosen bound of a shuffled index did not hold,
 * 2: NOT redone -- queries whose retained set (rows within 2 eps of the K-th prefilter score: near-
 * duplicate clusters) exceeded K + band and was re-scored in full inside the list kernel};
 * reasons_h[3] = length of the longest survivor list of the call
 * when one exceeded 768 entries, else 0 (the list kernel holds 1024 entries per query). */
int tfrs_bruteforce_topk_redo_reasons(const void *workspace, int64_t nq, int64_t n, int k,
                                      int32_t *reasons_h, void *stream);

/* Test hook (host code only, no GPU needed): the threshold-pass plan the fp16-prefiltered search
 * would use for a corpus of n rows and top-k -- plan_h[5] = {sampling stride, sampled stages,
 * stages per bin, rank of the bin maximum the bound is taken from, 1 when that rank is a
 * statistical one (rank < k: the list kernel verifies the bound), else 0}.  shuffled != 0 asks
 * for the plan of a shuffled index (>= 65536 rows), 0 for the guaranteed plan.  sampled
 * stages == 0: the corpus is too small for the prefilter (all-f32 rounds are used). */
int tfrs_debug_topk_plan(int64_t n, int k, int shuffled, int64_t *plan_h);

/* Test hook: raw scores of the fp16 PREFILTER (never returned by the product path) for rows
 * [row_begin, row_end) of the index, row_begin a multiple of 128: out[nq, ld] with
 * ld = (row_end - row_begin) rounded up to 128; scratch: 2 * nq floats.  tests/ check
 * |s_16 - s_f32| against the bound ||q|| ||c|| * 0.0011 that the filter relies on. */
int tfrs_debug_fp16_scores(const tfrs_index_t *index, const float *queries, int64_t nq,
                           int64_t row_begin, int64_t row_end, float *out, float *scratch,
                           void *stream);

/* ------------------------------------------------------------------------- *
 * Streaming.call (layers/factorized_top_k.py:404-509): one candidate block.
 * Replaces top_scores (:424-438) + the reduce step top_k (:440-472) for the block
 * cand_block[nb, d], whose rows carry the global row numbers base_row..base_row+nb-1
 * (enumerate_rows :474-480).  The running state is state_scores/state_idx[nq, k]
 * with state_len valid, sorted entries per row; it is updated in place and the new
 * length min(k, state_len + nb) is returned through *new_len_h
 * (handle_incomplete_batches=True semantics, :431-434,:465-468).
 * ------------------------------------------------------------------------- */
size_t tfrs_streaming_topk_workspace_bytes(int64_t nq, int64_t nb, int d, int k);
int tfrs_streaming_topk_update(const float *queries, int64_t nq, int d,
                               const float *cand_block, int64_t nb, int64_t base_row,
                               int k, float *state_scores, int32_t *state_idx,
                               int32_t state_len, int32_t *new_len_h, void *workspace,
                               size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Streaming.call over a GROUP of consecutive candidate blocks read in place
 * (layers/factorized_top_k.py:404-509; the layer keeps only a reference to its dataset and
 * re-reads it on every call, :384-390,:496-507).  blocks_h[nblocks] are DEVICE pointers to
 * row-major float32 [block_rows_h[b], d] blocks (the arrays themselves live on the host), whose
 * rows carry the global row numbers base_row, base_row + 1, ... in block order (:474-480);
 * seen_rows = candidates already folded into the state by earlier calls.  One call replaces
 * nblocks calls of tfrs_streaming_topk_update with identical results: no packed copy of the
 * blocks is built -- query batches up to 256 take ONE pass over the f32 blocks per
 * range: an fp16 filter fed by the blocks themselves, each wave converting its 32 rows of a stage in
 * registers, with exact re-scoring of the survivors from the blocks; up to 32 queries below d = 128 an
 * HBM-bound exact f32-MFMA scan; large batches one fp16 prefilter image of the group with exact
 * re-scoring from the blocks (csrc/topk_raw.hip).  Requirements: d in {8, 16, 32, 64, 128}, 16-byte aligned
 * block pointers, nblocks <= 192 (TFRS_EINVAL otherwise: use the per-block entry point); the blocks
 * must stay valid until the work enqueued on `stream` has run.  Workspace from
 * tfrs_streaming_topk_blocks_workspace_bytes(nq, total rows of the group, d, k).
 * ------------------------------------------------------------------------- */
size_t tfrs_streaming_topk_blocks_workspace_bytes(int64_t nq, int64_t total_rows, int d, int k);
int tfrs_streaming_topk_update_blocks(const float *queries, int64_t nq, int d,
                                      const float *const *blocks_h, const int64_t *block_rows_h,
                                      int nblocks, int64_t base_row, int64_t seen_rows, int k,
                                      float *state_scores, int32_t *state_idx, int32_t state_len,
                                      int32_t *new_len_h, void *workspace, size_t workspace_bytes,
                                      void *stream);

/* ------------------------------------------------------------------------- *
 * Embedding dims above TFRS_MAX_DIM (the fused scan kernels keep a row in registers): the same
 * result through materialised score blocks.
 *   tfrs_compute_scores: scores[nq, nc] = q @ c^T (TopK._compute_score,
 *     layers/factorized_top_k.py:320-333; also Retrieval.call scores tasks/retrieval.py:172-180),
 *     the candidate matrix read in place as the transposed operand; f16 != 0 selects the
 *     split-fp16 MFMA GEMM (workspace from tfrs_gemm_f16_workspace_bytes(nq, nc, d)).
 *   tfrs_topk_update_from_scores: folds a score block (columns = rows base_row.. of the corpus)
 *     into the running top-K state exactly like tfrs_streaming_topk_update (:440-472).
 * Scores of this path are GEMM sums (f32 accuracy, not the bit-defined fma chain).
 * ------------------------------------------------------------------------- */
int tfrs_compute_scores(const float *queries, const float *candidates, int64_t nq, int nc, int d,
                        float *out, int f16, void *workspace, size_t workspace_bytes,
                        void *stream);
int tfrs_topk_update_from_scores(const float *scores, int64_t nq, int64_t nb, int64_t ld,
                                 int64_t base_row, int k, float *state_scores,
                                 int32_t *state_idx, int32_t state_len, int32_t *new_len_h,
                                 void *stream);

/* ------------------------------------------------------------------------- *
 * Merge of partial top-K lists (the multi-GPU exchange step and the general form of
 * the Streaming reduce, :459-472): parts are scores[nparts, nq, k_in] /
 * idx[nparts, nq, k_in] (each sorted or not), result out[nq, k_out] under
 * (score desc, idx asc).  k_out <= nparts * k_in.
 * ------------------------------------------------------------------------- */
size_t tfrs_topk_merge_workspace_bytes(int64_t nq, int nparts, int k_in, int k_out);
int tfrs_topk_merge(const float *scores_parts, const int32_t *idx_parts, int nparts,
                    int64_t nq, int k_in, int k_out, float *out_scores,
                    int32_t *out_idx, void *workspace, size_t workspace_bytes,
                    void *stream);

/* Same with parts that are part_stride elements apart (>= nq * k_in): lets the multi-GPU
 * exchange gather ONE buffer [world, 2, nq, k] (scores and rows of a rank back to back)
 * with a single all-gather and merge it in place. */
int tfrs_topk_merge_strided(const float *scores_parts, const int32_t *idx_parts, int nparts,
                            int64_t part_stride, int64_t nq, int k_in, int k_out,
                            float *out_scores, int32_t *out_idx, void *stream);

/* ------------------------------------------------------------------------- *
 * _exclude (layers/factorized_top_k.py:83-115) on int32 identifiers:
 *   isin = any(ids[:, :, None] == exclude[:, None, :]); adjusted = scores - 1e5*isin;
 *   top_k(adjusted, min(k, kin)); gather ORIGINAL scores and ids.
 * scores/ids[nq, kin], exclude[nq, ne] -> out_scores/out_ids[nq, kout=min(k,kin)].
 * ------------------------------------------------------------------------- */
int tfrs_topk_exclude(const float *scores, const int32_t *ids, int64_t nq, int kin,
                      const int32_t *exclude, int ne, int k, float *out_scores,
                      int32_t *out_ids, void *stream);

/* ------------------------------------------------------------------------- *
 * FactorizedTopK.update_state, score-based branch
 * (metrics/factorized_top_k.py:133-134,181-192):
 *   pos = sum_d q*c (same fma chain as the scores);
 *   hit[b, j] = in_top_k(target 0, concat([pos, topk]), ks[j])
 *             = (#{topk[b, :] > pos[b]} < ks[j]) && isfinite(pos[b]).
 * out_hits[nks, nq] f32 (0/1).  ks_h is a HOST array.
 * ------------------------------------------------------------------------- */
int tfrs_rank_of_positive(const float *queries, const float *true_candidates,
                          int64_t nq, int d, const float *topk_scores, int kmax,
                          const int32_t *ks_h, int nks, float *out_hits, void *stream);
/* id-based branch (:141-180): hit = any(ids[b, :ks[j]] == true_id[b]). */
int tfrs_id_match_topk(const int32_t *retrieved_ids, const int32_t *true_ids,
                       int64_t nq, int kmax, const int32_t *ks_h, int nks,
                       float *out_hits, void *stream);

/* The same score-based branch WITHOUT the top-K (metrics/factorized_top_k.py:133-137,181-192):
 * the retrieved list holds the max(ks) best scores of the corpus and in_top_k counts strictly
 * greater predictions, so for every k <= max(ks)
 *     hit_k[b] = (#{candidates j : score(q_b, cand_j) > pos_b} < k) && isfinite(pos_b),
 * i.e. the metric needs ONE count per query, not a sorted list.
 *   tfrs_rank_count_accumulate: counts[b] += #{j < nc : score(q_b, row_j) > pos_b} for one block of
 *     candidates, row_j = candidates[cand_ids[j]] (cand_ids int32 / int64; NULL: row_j =
 *     candidates[j]; ids outside [0, vocab) score as zero rows, like tfrs_embedding_gather_fwd) --
 *     with an id indirection the reference's `movies.batch(128).map(item_model)` candidate sweep of
 *     an Embedding tower (README.md:69-80) is one launch: gather, scores, rank.  f32 MFMA, the
 *     d-ordered fma chain of every scoring kernel (the positive ties with its own copy).  d <= 128.
 *     `counts` (uint32 [nq]) must be zero before the first block of a sweep (first_block = 1 also
 *     flags non-finite positives in bit 31); tfrs_topk_hits_update re-arms it.
 *   tfrs_topk_hits_update: state[i] += sum_b w_b hit_ks[i][b], state[nks + i] += sum_b w_b
 *     (tf.keras.metrics.Mean per k, :85-89,191-192; sample_weight NULL = ones),
 *     results[i] = state[i] / state[nks + i] (0 when empty), optional per-example hits[nks, nq];
 *     zeroes `counts`.  One workgroup, fixed reduction order.  ks_h is a HOST array (<= 16). */
int tfrs_rank_count_accumulate(const float *queries, const float *true_candidates, int64_t nq, int d,
                               const float *candidates, const void *cand_ids, int ids_i64, int64_t nc,
                               int64_t vocab, uint32_t *counts, int first_block, void *stream);
int tfrs_topk_hits_update(uint32_t *counts, int64_t nq, const int32_t *ks_h, int nks,
                          const float *sample_weight, float *state, float *results, float *hits,
                          void *stream);

/* ------------------------------------------------------------------------- *
 * Embedding lookup (tf.keras.layers.Embedding as called at README.md:62-66,77-78;
 * TPUEmbedding CPU branch layers/embedding/tpu_embedding_layer.py:913-919).
 * ids are int32 or int64 (ids_are_i64).  Out-of-range ids set *err_flag (device
 * int32, may be NULL) and produce a zero row.
 * ------------------------------------------------------------------------- */
int tfrs_embedding_gather_fwd(const float *table, int64_t vocab, int d, const void *ids,
                              int ids_are_i64, int64_t n, float *out, int32_t *err_flag,
                              void *stream);
/* combiner: 0 = sum, 1 = mean, 2 = sqrtn.  CSR segments row_splits[nrows + 1] (i32/i64
 * like ids); weights may be NULL.  out[nrows, d]. */
int tfrs_embedding_segment_reduce_fwd(const float *table, int64_t vocab, int d,
                                      const void *ids, const void *row_splits,
                                      int ids_are_i64, const float *weights,
                                      int64_t nrows, int combiner, float *out,
                                      int32_t *err_flag, void *stream);
/* Backward of the combiner lookup: grad_rows[p, :] = (grad_out[b, :] / den_b) * w_p for every
 * entry p of segment b (den_b = 1 | sum w | sqrt(sum w^2) as in the forward), i.e. the values
 * of the IndexedSlices gradient whose indices are the looked-up ids; feed (ids, grad_rows) to
 * tfrs_embedding_scatter_add_* (TPUEmbedding CPU branch, tpu_embedding_layer.py:913-919 under
 * tape.gradient, models/base.py:77).  grad_rows[nnz, d], nnz = row_splits[nrows]. */
int tfrs_embedding_segment_reduce_bwd(const float *grad_out, int d, const void *row_splits,
                                      int splits_are_i64, const float *weights, int64_t nrows,
                                      int combiner, float *grad_rows, void *stream);
/* Backward of gather: deterministic, atomics-free scatter-add.  `perm`/`sorted_ids`
 * are caller-provided sort results (ids sorted ascending, perm = source positions).
 * Produces the dense grad_table[vocab, d] rows for the touched ids only (other rows
 * untouched) -- or, when adagrad != 0, applies the fused row-wise Adagrad update
 * (models/base.py:77-78 with Adagrad, README.md:84):
 *   g = sum of duplicate grads; acc += g*g; row -= lr * g / sqrt(acc + eps)      (adagrad == 1:
 *   tf.keras.optimizers.Adagrad of TF >= 2.11 / tf-keras), or ... / (sqrt(acc) + eps)   (adagrad == 2: the optimizer_v2
 *   kernel ResourceApplyAdagradV2 of TF <= 2.10, also torch.optim.Adagrad's form).
 * Negative ids (the padding slots of a max_sequence_length feature) contribute nothing. */
int tfrs_embedding_scatter_add_bwd(const float *grad_out, const int64_t *sorted_ids,
                                   const int64_t *perm, int64_t n, int d,
                                   float *grad_table_or_table, float *accum, float lr,
                                   float eps, int adagrad, void *stream);
/* The same from UNSORTED ids (int32 / int64): the library sorts (id, position) pairs itself
 * (stable LSD radix sort, 8 bits per pass over the bits of vocab) and then runs the segmented
 * scatter-add / fused Adagrad.  ids outside [0, vocab) are ignored -- they can never write
 * outside the table.  workspace from tfrs_embedding_scatter_add_workspace_bytes(n). */
size_t tfrs_embedding_scatter_add_workspace_bytes(int64_t n);
int tfrs_embedding_scatter_add_unsorted(const float *grad_out, const void *ids, int ids_are_i64,
                                        int64_t n, int d, int64_t vocab,
                                        float *grad_table_or_table, float *accum, float lr,
                                        float eps, int adagrad, void *workspace,
                                        size_t workspace_bytes, void *stream);

/* Same result without the sort, for small vocabularies: one wave per table row scans the id
 * list and sums matching gradient rows in occurrence order (O(vocab * n / 64) wave steps;
 * d <= 256).  In dense mode (adagrad == 0) EVERY row of grad_table[vocab, d] is written
 * (zeros for untouched rows). */
int tfrs_embedding_scatter_add_rowscan(const float *grad_out, const void *ids, int ids_are_i64,
                                       int64_t n, int d, int64_t vocab,
                                       float *grad_table_or_table, float *accum, float lr,
                                       float eps, int adagrad, void *stream);
/* The same for up to 8 small tables in ONE launch (the user and item tables of a two-tower step):
 * the *_h arguments are HOST arrays of `ntables` entries holding the per-table arguments of
 * tfrs_embedding_scatter_add_rowscan. */
int tfrs_embedding_scatter_add_rowscan_multi(int ntables, const float *const *grad_out_h,
                                             const void *const *ids_h, const int *ids_are_i64_h,
                                             const int64_t *n_h, const int *d_h,
                                             const int64_t *vocab_h, float *const *tables_h,
                                             float *const *accum_h, float lr, float eps,
                                             int adagrad, void *stream);
/* Dense Adagrad of up to 32 parameters in ONE launch (models/base.py:77-78 with tf.keras.optimizers.Adagrad on the dense
 * variables -- Cross kernels, MLP kernels and biases): acc += g * g; p -= lr * g / denom, denom = sqrt(acc + eps) (mode 1)
 * or sqrt(acc) + eps (mode 2), the same arithmetic as the fused sparse update above.  Host arrays of device pointers. */
int tfrs_adagrad_dense_multi(int ntensors, float *const *params_h, float *const *accum_h, const float *const *grads_h,
                             const int64_t *n_h, float lr, float eps, int mode, void *stream);
/* Up to 16 device buffers copied in ONE launch: a batch's input tensors into the static buffers of a captured
 * train / test step (the `Model.fit` loop of models/base.py:64-85 replays HIP graphs; README.md:84-98).  Host arrays of
 * device pointers and byte counts; buffers must not overlap. */
int tfrs_copy_multi(int nbuffers, void *const *dst_h, const void *const *src_h, const int64_t *bytes_h, void *stream);

/* ------------------------------------------------------------------------- *
 * tf.keras.layers.Hashing(num_bins, salt=[s0, s1]) as UnifiedEmbedding applies it per
 * feature chunk (layers/feature_multiplexing/unified_embedding.py:116-119,155-159,198-205):
 * bucket = SipHash-2-4_{(s0, s1)}(bytes) mod num_bins (unsigned), where bytes is the decimal
 * string of an integer id (tf.as_string) or the string itself.  out[n] int64.
 *   _ids:   ids[n] int32/int64 on the device.
 *   _bytes: n strings packed back to back in `bytes`, string i = bytes[offsets[i]:offsets[i+1]].
 * ------------------------------------------------------------------------- */
int tfrs_hash_bucket_strong_ids(const void *ids, int ids_are_i64, int64_t n, int64_t num_bins,
                                uint64_t salt0, uint64_t salt1, int64_t *out, void *stream);
int tfrs_hash_bucket_strong_bytes(const unsigned char *bytes, const int64_t *offsets, int64_t n,
                                  int64_t num_bins, uint64_t salt0, uint64_t salt1,
                                  int64_t *out, void *stream);

/* Fused UnifiedEmbedding.call for one feature (unified_embedding.py:198-215): for value v and
 * chunk c, bucket = SipHash-2-4_{(salt0[c], salt1[c])}(value) mod num_bins and
 *   out[v, c*d:(c+1)*d] = tables[c][bucket, :]
 * i.e. Hashing -> lookup -> concat of the feature's n_chunks components in one pass.  Values are
 * ids[n_values] (int32/int64) or, when `bytes` != NULL, strings packed as in
 * tfrs_hash_bucket_strong_bytes.  `tables`, `salt0`, `salt1` are HOST arrays of n_chunks
 * entries (device table pointers, each [num_bins, d]); d must be 4 * 2^k <= 256, otherwise
 * TFRS_ENOTIMPL.  out[n_values, n_chunks * d]; buckets[n_values, n_chunks] (optional, may be
 * NULL) keeps the bucket of every lookup for the backward (the IndexedSlices indices). */
int tfrs_unified_embedding_fwd(const void *ids, int ids_are_i64, const unsigned char *bytes,
                               const int64_t *offsets, int64_t n_values, int n_chunks,
                               const float *const *tables, const uint64_t *salt0,
                               const uint64_t *salt1, int64_t num_bins, int d, float *out,
                               int64_t *buckets, void *stream);

/* The same for EVERY dense integer feature of a UnifiedEmbedding layer in one launch
 * (unified_embedding.py:186-215 loops over the features).  A unit u is one (feature, chunk):
 *   outs[u][v, chunk_index[u]*d : (chunk_index[u]+1)*d] = tables[u][bucket(ids[u][v], salt0[u], salt1[u]), :]
 * where outs[u] is the [n_values, feature_chunks[u] * d] output of the unit's feature (the units of one
 * feature share it).  All host arrays have n_units entries; every ids[u] holds n_values ids of the same
 * integer width.  buckets[n_values, n_units] (optional) keeps the buckets for the backward. */
int tfrs_unified_embedding_fwd_multi(int n_units, const void *const *ids, int ids_are_i64,
                                     int64_t n_values, const float *const *tables,
                                     const uint64_t *salt0, const uint64_t *salt1, int64_t num_bins,
                                     int d, float *const *outs, const int32_t *feature_chunks,
                                     const int32_t *chunk_index, int64_t *buckets, void *stream);

/* ------------------------------------------------------------------------- *
 * Retrieval.call loss (tasks/retrieval.py:172-210, layers/loss.py:114-158):
 * in-batch sampled softmax without materialising the [nq, nc] logits.
 *   S = q c^T [/ temperature] [- log clip(p_c, 1e-6, 1)]
 *       [+ MIN_FLOAT where cand_ids[c] == cand_ids[b], c != b] [MIN_FLOAT where !mask]
 *   loss = sum_b w_b * (logsumexp_c S_bc - S_bb)
 * out_loss: device float; out_lse[nq], out_pos[nq] saved for backward.  Optional
 * inputs may be NULL (sample_weight, log_q_correction = log clip(p), cand_ids, mask).
 * Requires nc >= nq (labels = eye(nq, nc)) and d <= 128.  Forward and backward share one
 * workspace size query.
 * ------------------------------------------------------------------------- */
size_t tfrs_inbatch_softmax_workspace_bytes(int64_t nq, int64_t nc, int d);
int tfrs_inbatch_softmax_ce_fwd(const float *q, const float *c, int64_t nq, int64_t nc,
                                int d, const float *sample_weight, float inv_temperature,
                                const float *log_q_correction, const int64_t *cand_ids,
                                const uint8_t *score_mask, float *out_loss,
                                float *out_lse, float *out_pos, void *workspace,
                                size_t workspace_bytes, void *stream);
/* dq[nq, d], dc[nc, d] for upstream scalar gradient `gloss` (device float, NULL = 1).
 * reuse_forward_workspace != 0 promises that `workspace` is the buffer the forward call was
 * given for the same q, c, sample_weight and has not been written since: the backward then
 * reuses the operand images the forward left there instead of rebuilding them. */
int tfrs_inbatch_softmax_ce_bwd(const float *q, const float *c, int64_t nq, int64_t nc,
                                int d, const float *sample_weight, float inv_temperature,
                                const float *log_q_correction, const int64_t *cand_ids,
                                const uint8_t *score_mask, const float *lse,
                                const float *gloss, float *dq, float *dc, void *workspace,
                                size_t workspace_bytes, int reuse_forward_workspace,
                                void *stream);

/* ------------------------------------------------------------------------- *
 * Keras CategoricalCrossentropy(from_logits=True, reduction=SUM) on an EXPLICIT logits matrix
 * (tasks/retrieval.py:86-87, :210) -- only for the Retrieval paths that must build [nq, nc] (multi-head queries
 * :172-176, dims above TFRS_MAX_DIM, batch metrics / hard negatives after a logit adjustment :205-208); the default
 * loss is the fused tfrs_inbatch_softmax_ce_*.  labels[nq, nc] as the reference builds them (:185, loss.py:108-109).
 *   fwd: row_loss[i] = w_i * (lse_i * sum_j y_ij - sum_j y_ij s_ij); lse[nq], ysum[nq] are kept for the backward
 *   bwd: dlogits[i, j] = grad_scale[0] * w_i * (softmax(S_i)_j * ysum_i - y_ij)   (grad_scale: device scalar)
 * ------------------------------------------------------------------------- */
int tfrs_logits_ce_fwd(const float *logits, const float *labels, int64_t nq, int64_t nc,
                       const float *sample_weight, float *row_loss, float *lse, float *ysum, void *stream);
int tfrs_logits_ce_bwd(const float *logits, const float *labels, int64_t nq, int64_t nc,
                       const float *sample_weight, const float *lse, const float *ysum,
                       const float *grad_scale, float *dlogits, void *stream);

/* ------------------------------------------------------------------------- *
 * Cross.call (layers/feature_interaction/dcn.py:151-186), full rank, linear
 * preactivation:  y = x0 * (x @ kernel + bias + diag_scale * x) + x
 * kernel[d, d] is Keras Dense layout [in, out]; bias may be NULL.
 * ------------------------------------------------------------------------- */
int tfrs_cross_fwd(const float *x0, const float *x, const float *kernel,
                   const float *bias, float diag_scale, int64_t batch, int d, float *y,
                   void *stream);
/* Gradients of the same layer (its backward under tfrs.Model.train_step, models/base.py:77):
 * with z = x @ kernel + bias + diag_scale * x and dz = dy * x0
 *   dx0 = dy * z ;  dx = dz @ kernel^T + dy + diag_scale * dz ;
 *   dkernel = x^T @ dz ;  dbias[d] = column sums of dz (NULL to skip).
 * Three fused GEMM launches: z and dz never exist in HBM and nothing is transposed.
 * tfrs_cross_bwd = f32 MFMA; tfrs_cross_bwd_f16 = split-fp16 MFMA (large products), same
 * results to f32 accuracy.  workspace from tfrs_cross_bwd_workspace_bytes(batch, d, f16). */
size_t tfrs_cross_bwd_workspace_bytes(int64_t batch, int d, int f16);
int tfrs_cross_bwd(const float *x0, const float *x, const float *kernel, const float *bias,
                   float diag_scale, const float *dy, int64_t batch, int d, float *dx0,
                   float *dx, float *dkernel, float *dbias, void *workspace,
                   size_t workspace_bytes, void *stream);
int tfrs_cross_bwd_f16(const float *x0, const float *x, const float *kernel, const float *bias,
                       float diag_scale, const float *dy, int64_t batch, int d, float *dx0,
                       float *dx, float *dkernel, float *dbias, void *workspace,
                       size_t workspace_bytes, void *stream);
/* Training pair that trades 4 * batch * d bytes for one of the three products: the forward also
 * stores u = x @ kernel + bias + diag_scale * x (u_out[batch, d]; workspace as tfrs_cross_fwd_f16),
 * the backward forms dx0 = dy * u elementwise and runs the other two fused GEMMs
 * (workspace from tfrs_cross_bwd_workspace_bytes(batch, d, 1)). */
int tfrs_cross_fwd_f16_train(const float *x0, const float *x, const float *kernel,
                             const float *bias, float diag_scale, int64_t batch, int d, float *y,
                             float *u_out, void *workspace, size_t workspace_bytes, void *stream);
int tfrs_cross_bwd_f16_saved(const float *x0, const float *x, const float *u, const float *kernel,
                             float diag_scale, const float *dy, int64_t batch, int d, float *dx0,
                             float *dx, float *dkernel, float *dbias, void *workspace,
                             size_t workspace_bytes, void *stream);
/* tfrs_cross_bwd_f16_saved inside a STACK of Cross layers on one x0 (layers/feature_interaction/dcn.py:47-56,
 * "x1 = Cross()(x0, x0); x2 = Cross()(x0, x1)"): dx0 = dy * u + dx0_add, where dx0_add (NULL or [batch, d]) may be dx0
 * itself -- x0's gradient accumulates in place from layer to layer instead of through batch x d additions of the host
 * framework's autograd; add_dx != 0 (the stack's first layer, whose x is x0) folds that layer's dx in as well:
 * dx0 = dy * u + dx0_add + dx. */
int tfrs_cross_bwd_f16_saved_acc(const float *x0, const float *x, const float *u, const float *kernel,
                                 float diag_scale, const float *dy, int64_t batch, int d, const float *dx0_add,
                                 int add_dx, float *dx0, float *dx, float *dkernel, float *dbias, void *workspace,
                                 size_t workspace_bytes, void *stream);
/* Low-rank form (dcn.py:131-148, multi_layer_dcn.py:147-153): a[batch, ka] = x @ U is
 * computed first (tfrs_dense_fwd); this call does  y = x0 * (a @ kernel[ka, d] + bias +
 * diag_scale * x) + x  with the same fused epilogue. */
int tfrs_cross_fwd_ex(const float *x0, const float *x, const float *a, int ka,
                      const float *kernel, const float *bias, float diag_scale,
                      int64_t batch, int d, float *y, void *stream);
/* z = x @ kernel (+bias) only (used for low-rank U/V and activations): out[batch, dout]. */
int tfrs_dense_fwd(const float *x, const float *kernel, const float *bias, int64_t batch,
                   int din, int dout, float *out, void *stream);
/* Gradients of the same Dense (layers/blocks.py:46-61 MLP layers, low-rank Cross projections
 * dcn.py:176-180): dx[batch, din] = dy @ kernel^T, dkernel[din, dout] = x^T @ dy,
 * dbias[dout] = column sums of dy; NULL outputs are skipped.  The transposed operands are read
 * in place.  f16 != 0 selects the split-fp16 MFMA path (large products). */
size_t tfrs_dense_bwd_workspace_bytes(int64_t batch, int din, int dout, int f16);
int tfrs_dense_bwd(const float *x, const float *kernel, const float *dy, int64_t batch, int din,
                   int dout, float *dx, float *dkernel, float *dbias, int f16, void *workspace,
                   size_t workspace_bytes, void *stream);

/* Activations fused into the product's epilogue ("tfrs_cross_fwd(x0, x, w_or_(u,v), b, diag, act,
 * y)"; Keras Dense(activation=...) of layers/blocks.py:46-52 and Cross(preactivation=...) of dcn.py:173-181).
 * Codes: */
#define TFRS_ACT_NONE 0
#define TFRS_ACT_RELU 1
#define TFRS_ACT_SIGMOID 2
#define TFRS_ACT_TANH 3
#define TFRS_ACT_SILU 4   /* swish */
#define TFRS_ACT_GELU 5   /* exact (erf) form, Keras gelu(approximate=False) */
/*   tfrs_dense_fwd_act: out = act(x @ kernel + bias); pre_out (optional, [batch, dout]) receives the
 *     pre-activation.  f16 != 0: split-fp16 MFMA, workspace from tfrs_gemm_f16_workspace_bytes(batch, dout, din).
 *   tfrs_cross_fwd_act: y = x0 * (act(a @ kernel + bias) + diag_scale * x) + x with a[batch, ka] = x (full rank,
 *     ka == d) or x @ U (low rank); pre_out (optional) receives p = a @ kernel + bias.
 *   tfrs_act_pointwise_bwd: the element-wise part of their backward in ONE pass over `count` elements:
 *       dp  = dy * x0 * act'(pre)       (x0 == NULL: dy * act'(pre))            [optional output]
 *       dx0 = dy * (act(pre) + diag_scale * x)                                  [optional output]
 *       dxd = dy * (1 + diag_scale * x0)   (the direct terms of dx)             [optional output]
 *     ref_is_output != 0: `pre` holds y = act(p) (relu / sigmoid / tanh only).
 *   tfrs_dense_bwd_add: tfrs_dense_bwd with dx = dy @ kernel^T + addend[batch, din] (addend may be NULL): with
 *     dy = dp and addend = dxd this is the input gradient of a full-rank Cross layer with a preactivation. */
int tfrs_dense_fwd_act(const float *x, const float *kernel, const float *bias, int64_t batch, int din,
                       int dout, int act, float *out, float *pre_out, int f16, void *workspace,
                       size_t workspace_bytes, void *stream);
int tfrs_cross_fwd_act(const float *x0, const float *x, const float *a, int ka, const float *kernel,
                       const float *bias, float diag_scale, int act, int64_t batch, int d, float *y,
                       float *pre_out, int f16, void *workspace, size_t workspace_bytes, void *stream);
int tfrs_act_pointwise_bwd(int act, int ref_is_output, const float *pre, const float *dy, const float *x0,
                           const float *x, float diag_scale, int64_t count, float *dp, float *dx0,
                           float *dxd, void *stream);
int tfrs_dense_bwd_add(const float *x, const float *kernel, const float *dy, const float *addend,
                       int64_t batch, int din, int dout, float *dx, float *dkernel, float *dbias, int f16,
                       void *workspace, size_t workspace_bytes, void *stream);

/* The same two products on the fp16 matrix cores with split operands (x = hi + lo per
 * power-of-two-scaled row / column; hi*hi + hi*lo + lo*hi with f32 accumulation): f32-grade
 * results at ~3x the rate for large shapes.  The caller provides the workspace that holds the
 * operand images (tfrs_gemm_f16_workspace_bytes(batch, dout, din)). */
size_t tfrs_gemm_f16_workspace_bytes(int64_t m, int n, int k);
int tfrs_dense_fwd_f16(const float *x, const float *kernel, const float *bias, int64_t batch,
                       int din, int dout, float *out, void *workspace, size_t workspace_bytes,
                       void *stream);
int tfrs_cross_fwd_f16(const float *x0, const float *x, const float *kernel, const float *bias,
                       float diag_scale, int64_t batch, int d, float *y, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Row-sharded embedding lookup, owner bucketing (the multi-device
 * embedding of the reference is TPUEmbedding, layers/embedding/tpu_embedding_layer.py:699-720):
 * rank r owns table rows [r * rows_per_rank, (r + 1) * rows_per_rank).  For n lookups:
 *   counts[o]    (int64, DEVICE) = lookups served by owner o  (the all-to-all split sizes)
 *   send_ids[p]  (int64) shard-local row of the lookup in send slot p, owner by owner, STABLE
 *                within an owner (-1: id outside [0, input_dim) -> zero row, routed to owner 0)
 *   perm[i] = send slot of lookup i;  order[p] = lookup in send slot p   (int32)
 * so the rows that come back are gathered to their final positions through `perm`
 * (tfrs_embedding_gather_fwd with ids = perm) and the gradient rows are laid out for the way
 * back through `order`.  No sort, no host synchronisation, three short launches.
 * ------------------------------------------------------------------------- */
size_t tfrs_shard_route_workspace_bytes(int64_t n, int world);
int tfrs_shard_route_ids(const void *ids, int ids_are_i64, int64_t n, int64_t input_dim,
                         int64_t rows_per_rank, int world, int64_t *send_ids, int32_t *perm,
                         int32_t *order, int64_t *counts, void *workspace, size_t workspace_bytes,
                         void *stream);

/* ------------------------------------------------------------------------- *
 * DotInteraction.call (layers/feature_interaction/dot_interaction.py:53-104):
 * x[batch, f, d] -> lower-triangle pairwise dots, row-major; self_interaction adds the
 * diagonal; skip_gather emits the full f*f matrix with the upper part zeroed.
 * ------------------------------------------------------------------------- */
int tfrs_dot_interaction_fwd(const float *x, int64_t batch, int f, int d,
                             int self_interaction, int skip_gather, float *out,
                             void *stream);
int tfrs_dot_interaction_bwd(const float *x, const float *dout, int64_t batch, int f,
                             int d, int self_interaction, int skip_gather, float *dx,
                             void *stream);
/* Row-strided variants (packed-triangle output only): sample b's pairs live at out + b * out_stride
 * (dout + b * dout_stride), i.e. inside a wider [batch, out_stride] matrix -- lets the ranking model
 * write / read the block next to the bottom-stack output it is concatenated with
 * (experimental/models/ranking.py:225-232) without concat / slice copies.  TFRS_ENOTIMPL for shapes
 * outside the default kernels (fall back to the contiguous calls). */
int tfrs_dot_interaction_fwd_strided(const float *x, int64_t batch, int f, int d,
                                     int self_interaction, float *out, int64_t out_stride,
                                     void *stream);
int tfrs_dot_interaction_bwd_strided(const float *x, const float *dout, int64_t dout_stride,
                                     int64_t batch, int f, int d, int self_interaction, float *dx,
                                     void *stream);
/* 1 when BOTH strided entry points cover (batch, f, d): callers ask before they take the fused
 * concat path, so a forward can never succeed where its backward would return TFRS_ENOTIMPL. */
int tfrs_dot_interaction_strided_supported(int64_t batch, int f, int d, int self_interaction);

#ifdef __cplusplus
}
#endif
#endif /* TFRS_HIP_H_ */
