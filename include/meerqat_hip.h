/*
 * meerqat_hip.h -- C ABI of libmeerqat_hip.so, the MI355X (gfx950) implementation of the
 * arithmetic that PaulLerner/ViQuAE's dense-retrieval path delegates to FAISS and Hugging Face.
 *
 * The reference has no FFI of its own (it is pure Python); every entry point below names the
 * reference call site whose arithmetic it replaces.  Conventions:
 *   - plain pointers and sizes only, no C++/torch types; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - every `*_dev` / device pointer is HBM memory owned by the CALLER (the Python host allocates
 *     through torch); nothing is allocated, freed or retained across calls;
 *   - return value: MQ_OK (0) or a negative MQ_E* code; no exceptions cross the boundary;
 *     kernels are enqueued on `stream` and NOT synchronised (results are valid after the caller
 *     synchronises the stream);
 *   - re-entrant per stream.  The library keeps no state about a caller's data; what it does keep
 *     is process-wide bookkeeping that never changes a result: the last HIP error code of the
 *     calling THREAD (thread-local, mq_last_hip_error()), the CU count of the device (read once),
 *     one "dynamic-LDS limit raised" bit per (kernel, device), and the MQ_KNN_QPX tuning knob read
 *     from the environment once.  No environment variable is ever turned into a pointer in this
 *     build (cycle-accounting builds compiled with -DMQ_TIMING are a developer tool).
 */
#ifndef MEERQAT_HIP_H
#define MEERQAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_OK 0
#define MQ_EINVAL (-1)     /* bad argument (NULL pointer, non-positive size, unsupported k, ...) */
#define MQ_EWORKSPACE (-2) /* workspace too small */
#define MQ_EHIP (-3)       /* a HIP runtime call failed; see mq_last_hip_error() */
#define MQ_EUNSUPPORTED (-4)

#define MQ_METRIC_IP 0 /* faiss.METRIC_INNER_PRODUCT, "metric_type": 0 in experiments/ir/..../search/config.json */
#define MQ_METRIC_L2 1 /* faiss.METRIC_L2 (FAISS default when metric_type is None) */
/* Accepted by the mq_knn_screen_* entries, mq_knn_search_screened_f32, mq_knn_workspace_bytes_metric and mq_knn_screen_scan_kind only:
 * the inner product served by the CENTRED-QUERY screen (round 5).  With c the centre of the bf16 copy, q.x = (q - c).(x - c) +
 * c.(x - c) + q.c: the rows of a query rank by (q - c).(x - c) plus the per-row term c.(x - c), which the copy carries in two extra
 * columns like the L2 metric's -||x||^2 / 2; the rounding error -- hence the margin, hence the candidates per query -- then follows
 * the spread of the scores even when every vector shares a large common component (image features).  Results are the plain inner
 * product's, bit for bit (the screen is a lossless filter either way).  The index is built with this code (screen_bytes /
 * prepare / add_rows: a centre is required) and searched with it; xstats_dev then holds 4 + d floats: the four statistics, then
 * the centre. */
#define MQ_METRIC_IP_CENTRED 2

/* `--k` is a user option of the reference (meerqat/ir/search.py:12,135; default 100) and faiss IndexFlat takes any k.  One
 * fused scan keeps up to MQ_KNN_FUSED_K neighbours; a larger k (up to MQ_KNN_MAX_K, FAISS-GPU's own limit) is served by
 * ceil(k / 128) scans, round r + 1 admitting only candidates strictly below the last (score, id) key of round r -- the
 * rounds' results concatenate into the exact sorted top-k.  The rounds are exact fp32 scans; mq_knn_search_screened_f32 serves
 * k <= 224 through its screen and 225 <= k <= 1792 through the screen over ceil(k / 112) contiguous ROW RANGES whose top-224 lists
 * are merged -- the merge proves the union's top-k (no range delivered its whole list into it) or hands the query tile to the
 * rounds -- same results either way; it takes the rounds beyond 1792 and on shards too small for ranges (< 16,384 rows each). */
#define MQ_KNN_FUSED_K 128
#define MQ_KNN_MAX_K 2048

/* `flags` of the search entry points (the argument was `l2norm_queries`, 0 / 1, before the tie order became a parameter):
 *   MQ_KNN_FLAG_L2NORM_QUERIES  apply the index's "L2norm," transform to the queries first;
 *   MQ_KNN_FLAG_TIE_ID_DESC     among EXACTLY equal scores the HIGHER id is the better one -- for membership at the k-th
 *                               boundary and for the order of the output alike.  Default (bit clear): the LOWER id is better.
 * Which of several exactly tied rows FAISS keeps and in which order it reports them depends on its version and on k
 * (value-only heaps before 1.7.3, cmp2(val, id) heaps since, a reservoir + partition at k >= 100; oracle/knn_oracle.c
 * states what is known): both orders here are THIS LIBRARY'S documented policies, neither is a claim about FAISS.
 * mq_topk_merge_*: OR MQ_MERGE_TIE_ID_DESC into `metric` to merge shard results produced with MQ_KNN_FLAG_TIE_ID_DESC. */
#define MQ_KNN_FLAG_L2NORM_QUERIES 1
#define MQ_KNN_FLAG_TIE_ID_DESC 2
#define MQ_KNN_FLAG_L2NORM_FAISS 4 /* the query transform in FAISS's arithmetic (below); implies MQ_KNN_FLAG_L2NORM_QUERIES */
/* mq_knn_search_screened_f32 only: one search issued as two calls with the SAME arguments and workspace, so that a caller
 * searching several query chunks can put the short second half of chunk i (candidates -> exact re-scoring -> exact top-k ->
 * recomputation of flagged tiles) on another stream, under the scan of chunk i+1 (which then needs its own workspace):
 *   MQ_KNN_FLAG_PHASE_FRONT  query preparation + the screening scan only (D, I are not written yet);
 *   MQ_KNN_FLAG_PHASE_TAIL   everything after the scan, from the workspace the FRONT call filled (the caller orders the two:
 *                            the TAIL stream waits for an event recorded after the FRONT call).
 * Neither bit or both: the whole search in one call.  Searches the screen does not serve (k beyond its range, FAISS's small
 * L2 batches) run entirely in the FRONT call; their TAIL call returns MQ_OK without work.  Results are those of the
 * one-call search, bit for bit. */
#define MQ_KNN_FLAG_PHASE_FRONT 8
#define MQ_KNN_FLAG_PHASE_TAIL 16
#define MQ_MERGE_TIE_ID_DESC 0x100

/* The two arithmetics of the "L2norm," prefix -- the `l2norm` argument of mq_pack_rows_f32 / mq_knn_screen_add_rows_f32 (0 = no
 * transform), the `form` of mq_l2norm_rows_form_f32, and MQ_KNN_FLAG_L2NORM_FAISS for the queries of a search:
 *   MQ_L2NORM_NUMPY  x / sqrtf(sum x^2): the reference's L2norm() (meerqat/ir/search.py:43-46), which it also applies to the KB
 *                    column itself when `device` is given (its GPU work-around, :238-244).  A zero row becomes NaN.
 *   MQ_L2NORM_FAISS  FAISS's NormalizationTransform -> fvec_renorm_L2 (faiss/utils/distances.cpp, as published): when
 *                    nr = sum x^2 > 0, x *= (float)(1.0 / sqrtf(nr)) -- one double reciprocal per row rounded to float, one
 *                    multiplication per element; a row with nr not > 0 (zero, underflowed, NaN) stays as it is (a zero row
 *                    remains retrievable with score 0).  What "L2norm,Flat" computes with `device: null`, i.e. in every shipped
 *                    config (:230-245): for the KB rows on add and, inside the index, for the (already host-normalised) queries.
 * In both forms sum x^2 is the k-ordered fp32 fma chain (FAISS's SIMD summation order is not knowable here: parity unpinned). */
#define MQ_L2NORM_NUMPY 1
#define MQ_L2NORM_FAISS 2

/* faiss::distance_compute_blas_threshold.  A METRIC_L2 search of FEWER queries than this takes FAISS's
 * sequential path: distances are the direct sums of (q[k] - x[k])^2 (faiss fvec_L2sqr), not the BLAS form
 * ||q||^2 + ||x||^2 - 2<q,x> clamped at 0 that larger batches get (faiss/utils/distances.cpp, knn_L2sqr).
 * Both search entries below switch form on the `nq` of the CALL, so a host that cuts a large batch into
 * several calls must not leave a piece of fewer than 20 queries (viquae_amd.index.query_chunks). */
#define MQ_KNN_L2_DIRECT_BELOW 20

const char *mq_version(void);
const char *mq_strerror(int code);
int mq_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * KB matrix in HBM: the "panel" layout.  Replaces faiss IndexFlat's row-major storage filled by
 * FaissIndex.add_vectors (datasets/search.py:255-313, reached from meerqat/ir/search.py:245).
 * Row i, component k lives at   packed[((i / 64) * dpad + k) * 64 + (i % 64)],
 * dpad = d rounded up to 16, rows padded (zero) to a multiple of 256.
 * ------------------------------------------------------------------------------------------- */
int64_t mq_padded_rows(int64_t n_rows);
int mq_padded_dim(int d);
size_t mq_packed_bytes(int64_t n_rows, int d);

/* Pack `n` row-major fp32 rows (device) into the panel buffer at rows [row_offset, row_offset+n).
 * l2norm != 0 applies FAISS's "L2norm," NormalizationTransform (string_factory "L2norm,Flat",
 * meerqat/ir/search.py:230-233) to each row first.  sqnorm_dev[row] receives ||x||^2 of the stored
 * row (used by the L2 metric).  capacity_rows = mq_padded_rows(total rows) of the destination;
 * the caller zero-fills the panel buffer once before the first call. */
int mq_pack_rows_f32(const float *rows_dev, int64_t n, int d, int64_t row_offset, int l2norm, float *packed_dev,
                     int64_t capacity_rows, float *sqnorm_dev, void *stream);

/* Inverse of mq_pack_rows_f32 (FaissIndex.save / reconstruct, datasets/search.py:387-397). */
int mq_unpack_rows_f32(const float *packed_dev, int64_t capacity_rows, int d, int64_t row_offset, int64_t n,
                       float *rows_dev, void *stream);

/* In-place row L2 normalisation of a row-major [n,d] device matrix: L2norm(), meerqat/ir/search.py:43-46. */
int mq_l2norm_rows_f32(float *rows_dev, int64_t n, int d, void *stream);                     /* MQ_L2NORM_NUMPY */
int mq_l2norm_rows_form_f32(float *rows_dev, int64_t n, int d, int form, void *stream);      /* form: MQ_L2NORM_NUMPY | MQ_L2NORM_FAISS */

/* ---------------------------------------------------------------------------------------------
 * Exact brute-force top-k: replaces faiss IndexFlat.search behind FaissIndex.search_batch
 * (datasets/search.py:369-385), called by KnowledgeBase.search_batch (meerqat/ir/search.py:146).
 *   packed_dev/sqnorm_dev : the KB shard (N rows) as written by mq_pack_rows_f32
 *   queries_dev           : [nq, d] row-major fp32
 *   metric                : MQ_METRIC_IP or MQ_METRIC_L2
 *   flags                 : MQ_KNN_FLAG_* (above); 0 or 1 mean what `l2norm_queries` meant
 *   id_offset             : added to every returned row id (global id of the shard's row 0)
 *   D_dev [nq,k] fp32, I_dev [nq,k] int64: best first; equal scores by ascending id (descending with
 *                           MQ_KNN_FLAG_TIE_ID_DESC); unfilled slots are (-FLT_MAX | +FLT_MAX, -1): FAISS's heap
 *                           neutral values (CMin::neutral() / CMax::neutral()), what it reports when k > ntotal
 *   ws_dev/ws_bytes       : scratch of at least mq_knn_workspace_bytes_metric(N, d, nq, k, metric)
 * Scores are the k-ordered fp32 fma chain (see oracle/knn_oracle.c); selection is exact.
 * mq_knn_workspace_bytes covers either metric; the _metric variant leaves out what an inner-product search never touches
 * (the [nq][N] distance matrix of FAISS's small-batch L2 form: 114 MB per 1.5M-row shard at 19 queries).
 * ------------------------------------------------------------------------------------------- */
size_t mq_knn_workspace_bytes(int64_t N, int d, int nq, int k);
size_t mq_knn_workspace_bytes_metric(int64_t N, int d, int nq, int k, int metric);
int mq_knn_search_f32(const float *packed_dev, const float *sqnorm_dev, int64_t N, int d, const float *queries_dev,
                      int nq, int k, int metric, int flags, int64_t id_offset, float *D_dev,
                      int64_t *I_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* Same call; additionally records the caller's hipEvent_t handles (may be NULL) on `stream`
 * immediately before and after the scan kernel (knn_scan_kernel), so that a benchmark can time the
 * dominant kernel on the stream it is launched on. */
int mq_knn_search_f32_ev(const float *packed_dev, const float *sqnorm_dev, int64_t N, int d, const float *queries_dev,
                         int nq, int k, int metric, int flags, int64_t id_offset, float *D_dev,
                         int64_t *I_dev, void *ws_dev, size_t ws_bytes, void *stream, void *ev_scan_begin,
                         void *ev_scan_end);

/* ---------------------------------------------------------------------------------------------
 * Screened search: SAME exact result as mq_knn_search_f32 (both metrics), computed as a bf16
 * screening scan over a bf16 copy of the shard + exact fp32 re-scoring of the few survivors
 * (csrc/knn_screen.inc states the error bound that makes the screen lossless).  For MQ_METRIC_L2 the bf16 copy carries
 * two extra columns, the bf16 pair of -||x||^2/2 (queries get 1, 1), so that the same scan ranks by q.x - ||x||^2/2.
 * The bf16 copy is metric-specific: pass the same `metric` to the three calls.  Extra shard buffers:
 *   rowmajor_dev [N, d] fp32  : the stored rows in row-major order (re-scoring operand)
 *   bf16_dev                  : mq_knn_screen_bytes(N, d, metric) bytes, bf16 copy (rows padded to 256, d (+2) to 64); an
 *                               opaque buffer owned by these three calls -- stored tile by tile ([row / 256][col / 64][256][64]:
 *                               one K step's operand is one contiguous 32-KiB block), a prefix of the rows is a prefix of it
 *   xstats_dev                : FOUR floats kept by mq_knn_screen_prepare (zero them before its first call):
 *                               max ||x||^2, max ||xc - bf16(xc)||^2, max ||xc||^2 and max |c . xc| over the shard (xc = x -
 *                               centre c), the inputs of the error bound; under MQ_METRIC_IP_CENTRED 4 + d floats: the
 *                               caller stores the centre behind the statistics, where the search reads it
 *   center_dev                : NULL, or d floats subtracted from every row before the bf16 rounding (both metrics;
 *                               ANY fixed vector is valid: q.(x - c) ranks the rows of a query like q.x, and for
 *                               embeddings with a large shared component the screen's margin then follows ||x - c||);
 *                               the same vector must be passed for every row range of a shard
 * mq_knn_screen_prepare fills rowmajor/bf16 for rows [row_offset, row_offset+n) from the panel buffer.
 * mq_knn_screen_add_rows_f32 does the same straight from incoming rows (row-major [n, d], any row_offset), applying the
 * optional "L2norm," transform and writing ||x||^2 with mq_pack_rows_f32's arithmetic -- for a shard that keeps NO panel copy
 * (1.5x the matrix in HBM instead of 2.5x): mq_knn_search_screened_f32 then takes packed_dev = NULL.
 * Query tiles whose bounded candidate buffers overflow are recomputed by the exact scan inside the
 * same call (from the panel copy, or from the row-major copy when packed_dev is NULL: same MFMA sequence, same bits, a
 * slower operand path); FAISS's small-batch L2 form (MQ_KNN_L2_DIRECT_BELOW) likewise reads whichever copy exists.
 * Workspace: mq_knn_workspace_bytes (covers all paths).  The bounded screening buffers are sized for the reference's k = 100 and hold
 * up to k = 224; 225 <= k <= 1792 runs the screen per row range and merges (above; 1.5M x 768, 4096 queries: k = 100 / 224 / 256 / 512 /
 * 1024 / 1792 = 8.6 / 10.4 / 14.3 / 17.7 / 25.6 / 32.8 ms, where ceil(k / 128) exact rounds took 147 ms at k = 256 and 294 ms at 512);
 * MQ_KNN_PARTITIONS=0 or k > 1792: the exact rounds.
 * ------------------------------------------------------------------------------------------- */
size_t mq_knn_screen_bytes(int64_t n_rows, int d, int metric);
int mq_knn_screen_prepare(const float *packed_dev, const float *sqnorm_dev, int64_t capacity_rows, int d, int metric,
                          int64_t row_offset, int64_t n, float *rowmajor_dev, uint16_t *bf16_dev, float *xstats_dev,
                          const float *center_dev, void *stream);
int mq_knn_screen_add_rows_f32(const float *rows_dev, int64_t n, int d, int64_t row_offset, int l2norm, int metric,
                               int64_t capacity_rows, float *sqnorm_dev, float *rowmajor_dev, uint16_t *bf16_dev,
                               float *xstats_dev, const float *center_dev, void *stream);
int mq_knn_search_screened_f32(const float *packed_dev, const float *sqnorm_dev, const float *rowmajor_dev,
                               const uint16_t *bf16_dev, const float *xstats_dev, int64_t N, int d,
                               const float *queries_dev, int nq, int k, int metric, int flags, int64_t id_offset,
                               float *D_dev, int64_t *I_dev, void *ws_dev, size_t ws_bytes, void *stream,
                               void *ev_scan_begin, void *ev_scan_end);
/* Telemetry of the last screened search held in ws_dev (synchronises the stream): out[0] = query
 * tiles recomputed by the exact scan, out[1] = candidates re-scored in total, out[2] = max per query,
 * out[3] / out[4] = max / total slab-pool entries, out[5] = max margin (1e-6 units), out[6] = slabs. */
int mq_knn_screen_stats(int64_t N, int d, int nq, int k, const void *ws_dev, int64_t out[8], void *stream);
/* Which screening scan mq_knn_search_screened_f32 runs for this problem on this device: the 256 x 256 tile kernel
 * (screen_scan_kernel) or, for ONE query tile (nq <= 256 -- the reference's Dataset.map batch,
 * experiments/ir/viquae/dpr/search/config.json:25 -> meerqat/ir/search.py:146) over a shard of at least 65,536 rows with at
 * most 768 bf16 columns and k <= 128, the streaming kernel that keeps the queries in registers (csrc/knn_small.inc).
 * For 225 <= k <= 1792 the answer describes the scan of one row range.
 * MQ_SCAN_KIND_NONE: the call is not served by a screening scan at all (k > 1792 or row ranges too small: exact rounds; fewer
 * than 20 L2 queries: FAISS's direct form).  Negative: MQ_EINVAL.  MQ_KNN_OPT_SMALL_SCAN = 0 (below) switches the streaming kernel
 * off, MQ_KNN_OPT_SMALL_MIN_TILES = <n> sets its floor of 32-row tiles per workgroup (default 8). */
#define MQ_SCAN_KIND_NONE 0
#define MQ_SCAN_KIND_TILE 1
#define MQ_SCAN_KIND_STREAM 2
int mq_knn_screen_scan_kind(int64_t N, int d, int nq, int k, int metric);
/* Process-wide A/B switches of the search paths.  Their first values come from the environment, read ONCE (MQ_KNN_SMALL,
 * MQ_KNN_SMALL_MIN_TILES, MQ_KNN_PARTITIONS); a running process flips them with mq_knn_set_option (atomic; a search already in
 * its planning step may still see the old value), never by editing the environment.  Values are >= 0.
 *   MQ_KNN_OPT_SMALL_SCAN        1 (default): one query tile may be served by the streaming kernel; 0: always the tile kernel
 *   MQ_KNN_OPT_SMALL_MIN_TILES   the streaming kernel's floor of 32-row tiles per workgroup; 0 (default) = the built-in 8
 *   MQ_KNN_OPT_PARTITIONS        1 (default): 225 <= k <= 1792 is served over row ranges; 0: exact rounds
 *   MQ_KNN_OPT_SMALL_WAVES       8 (default): the streaming kernel with two waves per SIMD (csrc/knn_small8.inc, round 5);
 *                                4: round 4's one wave per SIMD (csrc/knn_small.inc).  Environment: MQ_KNN_SMALL_WAVES
 * mq_knn_set_option returns the previous value, mq_knn_get_option the current one; MQ_EINVAL for an unknown key / negative value. */
#define MQ_KNN_OPT_SMALL_SCAN 0
#define MQ_KNN_OPT_SMALL_MIN_TILES 1
#define MQ_KNN_OPT_PARTITIONS 2
#define MQ_KNN_OPT_SMALL_WAVES 3
#define MQ_KNN_OPT_COUNT 4
int mq_knn_set_option(int key, int value);
int mq_knn_get_option(int key);

/* Name and launch geometry of the scan kernel for the given problem (for bench.py / profiles):
 * out[0]=workgroups, out[1]=threads per workgroup, out[2]=LDS bytes (exact scan), out[3]=query tiles,
 * out[4]=KB slabs, out[5]=KB chunks (256 rows each), out[6]/out[7]=threads / LDS bytes of the screening scan. */
int mq_knn_launch_info(int64_t N, int d, int nq, int k, int64_t out[8]);

/* Merge per-shard results after the all-gather (new step; no reference
 * counterpart): Ds/Is [nshards, nq, k] with GLOBAL ids -> the k best per query.  metric: MQ_METRIC_IP / MQ_METRIC_L2,
 * optionally OR-ed with MQ_MERGE_TIE_ID_DESC; any k <= MQ_KNN_MAX_K. */
int mq_topk_merge_f32(const float *Ds_dev, const int64_t *Is_dev, int nshards, int nq, int k, int metric,
                      float *D_dev, int64_t *I_dev, void *stream);

/* The same merge over the buffer ONE all-gather delivers ("one RCCL all-gather of
 * (score f32, id i64)[nq,k] per rank").  A rank's record is mq_shard_record_bytes(nq, k) bytes:
 *   [0, nq*k*4)                                   fp32 scores [nq,k]
 *   [mq_shard_record_ids_offset(nq,k), + nq*k*8)  int64 GLOBAL ids [nq,k]
 * (offsets rounded so that ids are 8-byte and records 16-byte aligned).  A shard's search writes D / I
 * straight into its record (D_dev = record, I_dev = record + ids offset), the collective concatenates the
 * records rank-major, and records_dev = that [nshards * record_bytes] buffer (8-byte aligned). */
size_t mq_shard_record_bytes(int nq, int k);
size_t mq_shard_record_ids_offset(int nq, int k);
int mq_topk_merge_records_f32(const void *records_dev, int nshards, int nq, int k, int metric, float *D_dev,
                              int64_t *I_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder building blocks (fp32): the arithmetic of Hugging Face DPRContextEncoder /
 * DPRQuestionEncoder (reached from meerqat/ir/embedding.py:226; op order stated in-tree by
 * meerqat/models/bert.py:12-380) and CLIPModel.get_image_features (meerqat/image/embedding.py:156-161).
 * Activations are row-major fp32 [tokens, features]; weights keep nn.Linear's [out, in] layout.
 * ------------------------------------------------------------------------------------------- */
#define MQ_EPI_NONE 0           /* C = A.W^T                       (CLIP patch embedding, visual projection) */
#define MQ_EPI_BIAS 1           /* C = A.W^T + b                   (Q/K/V projections)                        */
#define MQ_EPI_BIAS_GELU 2      /* C = gelu_erf(A.W^T + b)         (BertIntermediate, bert.py:217-229)         */
#define MQ_EPI_BIAS_QUICKGELU 3 /* C = x*sigmoid(1.702x), x=A.W^T+b (CLIP MLP fc1)                            */
#define MQ_EPI_BIAS_RESIDUAL 4  /* C = A.W^T + b + R               (BertSelfOutput/BertOutput dense + input)  */

/* A/B switch of the split-bf16 GEMM kernels (both give bit-identical outputs).  First value from the environment, read ONCE
 * (MQ_GEMM_WIDE); a running process flips it with mq_gemm_set_option (atomic), which returns the previous value (MQ_EINVAL for an
 * unknown key).
 *   MQ_GEMM_OPT_WIDE   1 (default): eight waves of 128 x 64 outputs, two per SIMD, fragments prefetched across the mid-step
 *                      barrier, a two-step LDS-DMA lookahead that runs across tile boundaries (csrc/gemm_x3w.inc, round 6);
 *                      0: sixteen waves of 64 x 64 (gemm_nt_x3s_kernel, rounds 2-5).
 *   MQ_GEMM_OPT_STAGGER  (eight-wave kernel) start offset between the four phases of workgroups, in units of 1024 shader
 *                      clocks; 0 = all workgroups start together.  Environment: MQ_GEMM_STAGGER */
#define MQ_GEMM_OPT_WIDE 0
#define MQ_GEMM_OPT_STAGGER 1
#define MQ_GEMM_OPT_COUNT 2
int mq_gemm_set_option(int key, int value);

/* nn.Linear with a fused epilogue: A [M,K], W [N,K], bias [N] or NULL, R [M,N] or NULL, C [M,N].
 * K must be a multiple of 16; A and W 16-byte aligned. */
int mq_gemm_nt_f32(const float *A_dev, const float *W_dev, const float *bias_dev, const float *residual_dev,
                   float *C_dev, int M, int N, int K, int epilogue, void *stream);

/* Split-bf16 variant of mq_gemm_nt_f32: W is given as two bf16 matrices Wh + Wl (mq_split_bf16_f32, once per
 * weight), A stays fp32 and is split in registers; C ~= Ah.Wh + Al.Wh + Ah.Wl with fp32 accumulation --
 * relative error ~1e-5 (fp32-class) at 3/16 of the fp32-MFMA cycles.  K must be a multiple of 32. */
int mq_split_bf16_f32(const float *src_dev, int64_t n, uint16_t *hi_dev, uint16_t *lo_dev, void *stream);
/* The same split of a weight matrix W [N, K] (K a multiple of 32), stored in TILE layout [ceil(N / 256)][K / 32][256][32]
 * (rows beyond N are zeros): the operand of one K step of a GEMM tile is then one contiguous 16-KiB block.  hi / lo hold
 * mq_split_bf16_tiled_elems(N, K) elements each.  Pass such a pair to mq_gemm_nt_bf16x3_f32 / mq_gemm_nt_bf16x3s_f32 with
 * MQ_GEMM_W_TILED or-ed into `epilogue`: same products in the same order, bit-identical results, 3.5-6 % faster GEMMs. */
#define MQ_GEMM_W_TILED 0x100
int64_t mq_split_bf16_tiled_elems(int N, int K);
int mq_split_bf16_tiled_f32(const float *W_dev, int N, int K, uint16_t *hi_dev, uint16_t *lo_dev, void *stream);
int mq_gemm_nt_bf16x3_f32(const float *A_dev, const uint16_t *Wh_dev, const uint16_t *Wl_dev, const float *bias_dev,
                          const float *residual_dev, float *C_dev, int M, int N, int K, int epilogue, void *stream);

/* nn.LayerNorm over the last dimension (C <= 1024); X and Y may alias. */
int mq_layernorm_f32(const float *X_dev, const float *gamma_dev, const float *beta_dev, float *Y_dev, int M, int C,
                     float eps, void *stream);

/* BertEmbeddings (meerqat/models/bert.py:153-214): LayerNorm(word[ids] + type[token_type_ids] + pos[0..L)).
 * ids / token_type_ids are int64 [B,L] (token_type_ids may be NULL = zeros); out [B*L, H]. */
int mq_bert_embed_ln_f32(const int64_t *input_ids_dev, const int64_t *token_type_ids_dev, const float *word_dev,
                         const float *pos_dev, const float *type_dev, const float *gamma_dev, const float *beta_dev,
                         float *out_dev, int B, int L, int H, float eps, void *stream);

/* Multi-head self-attention core (BertSelfAttention.forward, meerqat/models/bert.py:44-136):
 * qkv [B*L, 3*heads*head_dim] = [q | k | v] projections; attention_mask int64 [B,L] (1 = attend, as the
 * tokenizer emits it; NULL = attend everywhere); out [B*L, heads*head_dim] = softmax(q k^T * scale + mask) v.
 * head_dim must be 64; any L (sequences beyond 256 keys are processed in 256-key blocks with the online softmax). */
int mq_attention_f32(const float *qkv_dev, const int64_t *attention_mask_dev, float *out_dev, int B, int L, int heads,
                     int head_dim, float scale, void *stream);

/* CLIPVisionEmbeddings.patch_embedding as a GEMM operand: pixels [B,C,S,S] -> [B*(S/P)^2, C*P*P]. */
int mq_clip_patchify_f32(const float *pixels_dev, float *patches_dev, int B, int channels, int image_size,
                         int patch_size, void *stream);

/* CLIPVisionEmbeddings + pre_layrnorm: out [B*tokens, H] = LayerNorm([class | patch_emb] + position). */
int mq_clip_assemble_ln_f32(const float *patch_emb_dev, const float *class_emb_dev, const float *pos_emb_dev,
                            const float *gamma_dev, const float *beta_dev, float *out_dev, int B, int tokens, int H,
                            float eps, void *stream);

/* CLIPModel.get_text_features (the `call` of experiments/ir/viquae/clip/config.json:15):
 * mq_attention_causal_f32 = mq_attention_f32 with the text tower's causal mask (key <= query) on top of the
 * optional padding mask; mq_clip_text_embed_f32 = token + position embeddings -> out [B*L, H];
 * mq_clip_eos_pool_ln_f32 = final LayerNorm of the hidden state at the end-of-text token of each sequence
 * (eos_token_id == 2, the published checkpoints' legacy config: position of the largest id; otherwise the first
 * position equal to eos_token_id) -> out [B, H]. */
int mq_attention_causal_f32(const float *qkv_dev, const int64_t *attention_mask_dev, float *out_dev, int B, int L, int heads,
                            int head_dim, float scale, int causal, void *stream);
int mq_clip_text_embed_f32(const int64_t *input_ids_dev, const float *token_emb_dev, const float *pos_emb_dev, float *out_dev,
                           int B, int L, int H, void *stream);
int mq_clip_eos_pool_ln_f32(const float *hidden_dev, const int64_t *input_ids_dev, int64_t eos_token_id, const float *gamma_dev,
                            const float *beta_dev, float *out_dev, int B, int L, int H, float eps, void *stream);
/* mq_clip_text_embed_f32 over a PACKED token matrix (the real tokens of right-padded titles, see mq_attention_packed_f32):
 * out[t] = token_emb[input_ids[t]] + pos_emb[position_ids[t]], t < T. */
int mq_clip_text_embed_packed_f32(const int64_t *input_ids_dev, const int32_t *position_ids_dev, const float *token_emb_dev,
                                  const float *pos_emb_dev, float *out_dev, int T, int H, void *stream);

/* Split activations: a tensor that only feeds GEMMs is kept as its (hi, lo) bf16 pair (hi = bf16(x), lo = bf16(x - hi),
 * two uint16 arrays of the tensor's shape), written by the kernel that produces it, so that the consuming GEMM streams
 * bf16 operands and converts nothing in its MFMA loop.  Results are bit-identical to the fp32-activation entry points.
 *   mq_gemm_nt_bf16x3s_f32     mq_gemm_nt_bf16x3_f32 with A given as (Ah, Al); output EITHER fp32 C OR the pair (Ch, Cl)
 *   mq_layernorm_split_f32     mq_layernorm_f32 writing Y (fp32, may be NULL) and/or the pair (Yh, Yl)
 *   mq_bert_embed_ln_split_f32 mq_bert_embed_ln_f32 writing fp32 and, optionally, the pair
 *   mq_attention_split_f32     mq_attention_causal_f32 writing fp32 (may be NULL) and/or the pair; bf16x3 != 0 computes
 *                              q.k and p.v as three-term split-bf16 products on the bf16 matrix pipe (fp32-class accuracy)
 * PAIR LAYOUT: the two arrays of a pair [M, K] are stored tile by tile -- element (row, col) at
 * [row / 256][col / 32][row % 256][col % 32] -- so that the operand of one K step of a GEMM tile is one contiguous 16-KiB
 * block (the layout of mq_split_bf16_tiled_f32, which can mint such a pair from an fp32 matrix).  Each array holds
 * mq_split_bf16_tiled_elems(M, K) = ceil(M / 256) * 256 * K elements (rows beyond M are neither written nor read) and K must
 * be a multiple of 32 (MQ_EUNSUPPORTED otherwise).  Every producer below writes it, mq_gemm_nt_bf16x3s_f32 reads it; the
 * fp32 outputs stay row-major. */
int mq_gemm_nt_bf16x3s_f32(const uint16_t *Ah_dev, const uint16_t *Al_dev, const uint16_t *Wh_dev, const uint16_t *Wl_dev,
                           const float *bias_dev, const float *residual_dev, float *C_dev, uint16_t *Ch_dev, uint16_t *Cl_dev,
                           int M, int N, int K, int epilogue, void *stream);
/* mq_gemm_nt_bf16x3s_f32 with epilogue MQ_EPI_BIAS_RESIDUAL whose residual [M, N] is given as a split pair in PAIR LAYOUT (value
 * hi + lo, exact in fp32; N a multiple of 32): a LayerNorm output that only feeds GEMMs and shortcuts then needs no fp32 copy. */
int mq_gemm_nt_bf16x3s_respair_f32(const uint16_t *Ah_dev, const uint16_t *Al_dev, const uint16_t *Wh_dev,
                                   const uint16_t *Wl_dev, const float *bias_dev, const uint16_t *Rh_dev, const uint16_t *Rl_dev,
                                   float *C_dev, uint16_t *Ch_dev, uint16_t *Cl_dev, int M, int N, int K, int epilogue,
                                   void *stream);
/* Split-K form of mq_gemm_nt_bf16x3s_f32 for a long K over a small output (ArcFace's head: [faces, 512] over K = 25,088):
 * `nsplit` workgroup rows each accumulate a contiguous range of K steps into fp32 partials (partials_dev: nsplit * M * N floats),
 * then C = bias + the partials summed in split order (bias_dev may be NULL).  fp32 output only; w_tiled != 0: Wh / Wl in tile
 * layout.  The sum order differs from the one-pass GEMM's: equal within fp32 rounding, not bit for bit. */
int mq_gemm_nt_bf16x3s_splitk_f32(const uint16_t *Ah_dev, const uint16_t *Al_dev, const uint16_t *Wh_dev,
                                  const uint16_t *Wl_dev, const float *bias_dev, float *C_dev, int M, int N, int K,
                                  int w_tiled, int nsplit, float *partials_dev, void *stream);
int mq_layernorm_split_f32(const float *X_dev, const float *gamma_dev, const float *beta_dev, float *Y_dev, uint16_t *Yh_dev,
                           uint16_t *Yl_dev, int M, int C, float eps, void *stream);
int mq_bert_embed_ln_split_f32(const int64_t *input_ids_dev, const int64_t *token_type_ids_dev, const float *word_dev,
                               const float *pos_dev, const float *type_dev, const float *gamma_dev, const float *beta_dev,
                               float *out_dev, uint16_t *out_h_dev, uint16_t *out_l_dev, int B, int L, int H, float eps,
                               void *stream);
int mq_attention_split_f32(const float *qkv_dev, const int64_t *attention_mask_dev, float *out_dev, uint16_t *out_h_dev,
                           uint16_t *out_l_dev, int B, int L, int heads, int head_dim, float scale, int causal, int bf16x3,
                           void *stream);

/* Packed (variable-length) batch: the REAL tokens of all sequences, concatenated -- row r of every activation
 * matrix is one token, sequence b = rows [cu_seqlens[b], cu_seqlens[b+1]).  The reference pads every passage to
 * max_length 256 (experiments/ir/viquae/dpr/passages/config.json:11-14) although a passage has ~130 tokens; GEMMs and
 * LayerNorm are row-wise, so on the packed matrix they simply skip the padding, and these two entries supply the only
 * sequence-aware operations:
 *   mq_bert_embed_ln_packed_f32: BertEmbeddings + LayerNorm with an explicit position id per row;
 *   mq_attention_packed_f32: attention of the n_seqs sequences listed in seq_ids_dev (one launch per length class:
 *     max_len = the longest of them, it selects the key-tile count), each over exactly its own keys.  A padded key
 *     of the dense forward contributes exactly 0 to every sum, so the outputs equal the dense forward's bit for bit. */
int mq_bert_embed_ln_packed_f32(const int64_t *input_ids_dev, const int64_t *token_type_ids_dev, const int32_t *position_ids_dev,
                                const float *word_dev, const float *pos_dev, const float *type_dev, const float *gamma_dev,
                                const float *beta_dev, float *out_dev, uint16_t *out_h_dev, uint16_t *out_l_dev, int T, int H,
                                float eps, void *stream);
int mq_attention_packed_f32(const float *qkv_dev, const int32_t *cu_seqlens_dev, const int32_t *seq_ids_dev, int n_seqs,
                            int max_len, float *out_dev, uint16_t *out_h_dev, uint16_t *out_l_dev, int heads, int head_dim,
                            float scale, int causal, int bf16x3, void *stream);

/* Multimodal encoders of the reference (meerqat/models/mm.py: ECAEncoder :557-754, IntermediateLinearFusion :773-861) reuse
 * the entry points above; the only extra arithmetic is the sum of an example's face embeddings into its text vector
 * (mm.py:838-843): out[g, :] = init[g, :] + sum_j x[g, j, :], x [G, n, H], init (may be NULL) and out [G, H]. */
int mq_sum_groups_f32(const float *x_dev, const float *init_dev, float *out_dev, int G, int n, int H, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Late fusion of several runs on the device: replaces, for integer document
 * ids, `default_minimum` (meerqat/ir/fuse.py:129-146), `gzmuv_norm` (:86-126) and ranx's
 * `fuse(runs, norm, method="wsum", params={"weights": ...})` as called by `Fusion.test` (:215-236) from
 * `dataset_search` (meerqat/ir/search.py:514-524).
 *   ids_dev    [n_runs, nq, K] int64   document ids, -1 = empty slot; unique within one (run, query);
 *                                      0 <= id < 2^58; every run lists the same nq queries
 *   scores_dev [n_runs, nq, K] f64
 *   weights_host [n_runs] f64 (HOST memory), n_runs <= MQ_FUSE_MAX_RUNS, n_runs*K <= 4096
 *   norm       MQ_FUSE_NORM_NONE | _GZMUV (one mean/std per run over all its queries, population std,
 *              denominator max(std, 1e-9)) | _ZMUV (ranx "zmuv": the same per query)
 *   defmin     != 0: before normalising, every run that has results for a query receives the documents
 *              only other runs retrieved, at its own minimum score for that query
 *   out_ids_dev [nq, n_runs*K] int64 / out_scores_dev [nq, n_runs*K] f64: the fused run, best first,
 *              equal scores by ascending id, (-1, 0.0) after out_count_dev[q] entries
 * Fused score of a document = ((0.0 + w0*s0) + w1*s1) + ... over the runs that hold it, in run order,
 * f64 without contraction (what ranx's comb_sum computes).  ws_dev: mq_fuse_workspace_bytes().
 * ------------------------------------------------------------------------------------------- */
#define MQ_FUSE_MAX_RUNS 32
#define MQ_FUSE_NORM_NONE 0
#define MQ_FUSE_NORM_GZMUV 1
#define MQ_FUSE_NORM_ZMUV 2
size_t mq_fuse_workspace_bytes(int n_runs, int nq, int K);
int mq_fuse_wsum_f64(const int64_t *ids_dev, const double *scores_dev, int n_runs, int nq, int K,
                     const double *weights_host, int norm, int defmin, int64_t *out_ids_dev, double *out_scores_dev,
                     int32_t *out_count_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Rank metrics of a run against relevance judgements -- what `ranx.compare(qrels, runs, metrics=[...])` reports at
 * the end of a search job (meerqat/ir/search.py:397,500-512; ranx is un-vendored, `ranx>=0.3.2`: the published
 * definitions are restated, parity unpinned):
 *   ids_dev [nq, K] int64    one run, best first; a row ends at its first negative id
 *   rel_ptr_dev [nq + 1] / rel_ids_dev    CSR of the RELEVANT documents (judgement >= 1) of every query, ascending
 *                                         inside a query
 *   codes_host / ks_host [n_metrics]      MQ_RANK_METRIC_* and the cut k (0 = the whole run); n_metrics <= MQ_RANK_MAX_METRICS
 *     MRR        1 / (1 + rank of the first relevant document inside the cut), else 0
 *     PRECISION  (relevant inside the cut) / k     (k = 0: / the length of the query's run; an empty run scores 0)
 *     HIT_RATE   1 when a relevant document is inside the cut
 *     RECALL     (relevant inside the cut) / (relevant documents of the query), 0 when it has none
 *   per_query_dev [n_metrics, nq] f64, mean_dev [n_metrics] f64 = numpy's mean of each row (np.mean's pairwise
 *   summation restated: the number ranx.evaluate returns)
 * ------------------------------------------------------------------------------------------- */
#define MQ_RANK_METRIC_MRR 0
#define MQ_RANK_METRIC_PRECISION 1
#define MQ_RANK_METRIC_HIT_RATE 2
#define MQ_RANK_METRIC_RECALL 3
#define MQ_RANK_MAX_METRICS 16
int mq_run_metrics_f64(const int64_t *ids_dev, int nq, int K, const int64_t *rel_ptr_dev, const int64_t *rel_ids_dev,
                       int n_metrics, const int *codes_host, const int *ks_host, double *per_query_dev,
                       double *mean_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The weight search of `Fusion.fit` (meerqat/ir/fuse.py:193-217 -> `ranx.optimize_fusion(method="wsum")`): every
 * trial weight vector fuses the runs like mq_fuse_wsum_f64 and is scored by ONE rank metric, all trials in one
 * launch (a workgroup sorts and normalises a query's entries once, then walks the trials).
 *   ids_dev / scores_dev / n_runs / nq / K / norm / defmin: as mq_fuse_wsum_f64
 *   trials_dev [n_trials, n_runs] f64 (DEVICE memory)
 *   rel_ptr_dev / rel_ids_dev: as mq_run_metrics_f64;  metric / metric_k: one MQ_RANK_METRIC_* and its cut
 *   per_query_dev [n_trials, nq] f64: the metric of query q under trial t (rank = fused score descending, equal
 *   scores by ascending id -- the order mq_fuse_wsum_f64 writes); mean_dev [n_trials] f64: np.mean of each row.
 *   ws_dev: mq_fuse_workspace_bytes(n_runs, nq, K).
 * ------------------------------------------------------------------------------------------- */
int mq_fuse_fit_wsum_f64(const int64_t *ids_dev, const double *scores_dev, int n_runs, int nq, int K,
                         const double *trials_dev, int n_trials, int norm, int defmin, const int64_t *rel_ptr_dev,
                         const int64_t *rel_ids_dev, int metric, int metric_k, double *per_query_dev, double *mean_dev,
                         void *ws_dev, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Run files -- HOST function, no device work: replaces the per-hit Python objects behind
 * `run.save(metric_save_path / f"{index_name}.json")` (meerqat/ir/search.py:485-498) over the dicts of :413-440.  Formats the
 * rows of a result block kept as arrays into the text `json.dump({q_id: {str(doc): float(score)}})` writes, byte for byte
 * (", " and ": " separators, CPython's float repr of the score widened to double, NaN / Infinity spelled as json does),
 * WITHOUT the enclosing braces:   "<q0>": {"<id>": <score>, ...}, "<q1>": {...}
 *   qid_json / qid_off [nq + 1]: the already JSON-encoded question ids (quotes included), concatenated, and their offsets;
 *   ids [nq, stride] int64, scores [nq, stride] fp32 (scores_f64 = 0) or f64 (= 1), host memory; a row ends at counts[q]
 *   entries (counts NULL: stride) or at its first negative id (FAISS's -1 padding);
 *   out / out_cap: destination, at least nq * (8 + 52 * stride) + the length of qid_json bytes (the worst case);
 *   n_threads: 0 = all host cores (at most MQ_RUN_JSON_MAX_PARTS); parts [2 * MQ_RUN_JSON_MAX_PARTS] int64 (host).
 * The rows are cut into P contiguous parts formatted concurrently, each into its own region of `out`; returns P and sets
 * parts[2 * p] / parts[2 * p + 1] to the offset / length of part p: the run text is the parts in order, joined by ", " (written
 * one after the other by the caller: no second copy of a 50 MB run is made).  0 for nq = 0.  If out_cap is too small nothing
 * is written and the return value r < -16 encodes the size needed as -(r + 16); MQ_EINVAL for bad arguments. */
#define MQ_RUN_JSON_MAX_PARTS 32
int64_t mq_format_run_json(const char *qid_json, const int64_t *qid_off, int64_t nq, const int64_t *ids, const void *scores,
                           int scores_f64, int64_t stride, const int32_t *counts, char *out, int64_t out_cap, int n_threads,
                           int64_t *parts);

/* ---------------------------------------------------------------------------------------------
 * Image preprocessing in front of the CLIP tower: replaces
 * `transform(images, return_tensors="pt")` in `embed` (meerqat/image/embedding.py:141-152), the transform being the
 * `CLIPFeatureExtractor` of experiments/image_embedding/clip/vit_config.json:13-17 -- Pillow's 8-bit
 * `Image.resize(..., BICUBIC | BILINEAR)`, centre crop, `float32(float64(u8) * rescale_factor)`, `(x - mean) / std`.
 * Bit-identical to Pillow 12 + transformers 5 (`CLIPImageProcessorPil`).
 *
 * mq_image_plan (HOST arithmetic, needs no GPU): sizes_host [n_images][2] = (height, width) of the decoded RGB
 * images -> geom_host [n_images][MQ_IMAGE_GEOM] int64 (source byte offset of the image in the packed buffer, sizes,
 * resized sizes, crop origin, workspace offsets, tap counts) and totals_host[MQ_IMAGE_TOTALS] = {bytes of the packed uint8
 * source buffer (each image HWC at a 16-byte aligned offset, the buffer itself 16-byte aligned), workspace bytes, largest
 * image height, coefficient ints, largest image width}.
 *   resize_mode  MQ_IMAGE_RESIZE_NONE | _SHORTEST (shortest edge -> size_h, long edge int(size_h * long / short), the
 *                `get_resize_output_image_size(default_to_square=False)` rule) | _EXACT (size_h x size_w)
 *   MQ_EUNSUPPORTED when a resized image is smaller than the crop window (HF zero-pads; not provided) or for filters
 *   other than bilinear / bicubic.
 * mq_image_preprocess_u8: src_dev = the packed source buffer, geom_dev = geom_host copied to the device,
 *   totals_host = what the plan returned, flags = MQ_IMAGE_RESCALE | MQ_IMAGE_NORMALIZE, mean3_host / std3_host = 3
 *   floats in HOST memory; out_dev float32 [n_images][3][crop_h][crop_w].  Images wider than 21 k pixels: MQ_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
#define MQ_IMAGE_GEOM 12
#define MQ_IMAGE_TOTALS 5
#define MQ_IMAGE_BILINEAR 2 /* PIL.Image.Resampling values */
#define MQ_IMAGE_BICUBIC 3
#define MQ_IMAGE_RESIZE_NONE 0
#define MQ_IMAGE_RESIZE_SHORTEST 1
#define MQ_IMAGE_RESIZE_EXACT 2
#define MQ_IMAGE_RESCALE 1
#define MQ_IMAGE_NORMALIZE 2
int mq_image_plan(const int64_t *sizes_host, int n_images, int resize_mode, int size_h, int size_w, int crop_h, int crop_w,
                  int filter, int64_t *geom_host, int64_t *totals_host);
int mq_image_preprocess_u8(const uint8_t *src_dev, const int64_t *geom_dev, int n_images, const int64_t *totals_host, int crop_h,
                           int crop_w, int filter, int flags, double rescale_factor, const float *mean3_host,
                           const float *std3_host, float *out_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Baseline-JPEG decoding split between host and GPU (csrc/jpeg.hip): replaces, per image,
 * `Image.open(path).convert('RGB')` of `load_image` (meerqat/data/loading.py:108-124, called by meerqat/image/embedding.py:127)
 * for the files it covers -- 8-bit Huffman-coded JPEGs, sequential (SOF0 / SOF1) with ONE interleaved scan or progressive (SOF2)
 * with a complete, regular progression, grey or YCbCr with luma sampling 1x1 / 2x1 / 1x2 / 2x2 and chroma 1x1 -- with the same RGB bytes as Pillow 12 / libjpeg-turbo (ISLOW inverse DCT,
 * fancy upsampling, fixed-point colour tables; oracle/jpeg.py, pinned against Pillow).  Everything else, and any irregularity of
 * the entropy-coded data, is declined and stays Pillow's: the caller keeps the reference's errors and warnings.
 *
 * mq_jpeg_probe (HOST, needs no GPU): file bytes -> info_host[MQ_JPEG_INFO] = {height, width, components, 8 x 8 blocks,
 *   staging bytes (header + the larger of the coefficient blocks and the RGB image, a multiple of 16), luma sampling h * 16 + v
 *   (0 for grey)}.  MQ_OK | MQ_EUNSUPPORTED (a JPEG of another kind) | MQ_EINVAL (not a JPEG / damaged headers).
 * mq_jpeg_read_coefficients (HOST): Huffman-decodes the scan into staging_host (4-byte aligned, >= the probe's staging bytes):
 *   a MQ_JPEG_HEADER_BYTES header (int32 words: magic, height, width, components, hmax, vmax, MCUs across / down, then per
 *   component h, v, blocks across / down, first block, [total blocks], real samples across / down; uint16 quantisation tables
 *   [3][64] in natural order at byte 128) followed by the quantised coefficients, int16 [block][64] in natural order,
 *   component after component.  MQ_EINVAL when the scan is irregular in any way (the staging area is then undefined).
 *   A caller that decodes such a file by other means stores its H x W x 3 RGB bytes behind a header whose words 0-2 are
 *   MQ_JPEG_MAGIC_RGB, height, width.
 * mq_jpeg_decode_rgb_u8 (GPU): items_dev int64 [n_images][2] = (byte offset of an image's header, byte offset of its RGB
 *   output) inside buf_dev (16-byte aligned offsets and buffer); max_blocks / max_strips = the largest block count and the
 *   largest ceil(height / 2) * ceil(width / 16) of the batch (the colour kernel runs one thread per 16 x 2 pixels).  The inverse DCT runs in place (the coefficient area is consumed); the H x W x 3 bytes are what
 *   mq_image_preprocess_u8 / mq_warp_affine_faces_f32 read.
 * ------------------------------------------------------------------------------------------- */
#define MQ_JPEG_INFO 6
#define MQ_JPEG_HEADER_BYTES 512
#define MQ_JPEG_MAGIC_COEFFICIENTS 0x4745504A /* "JPEG" */
#define MQ_JPEG_MAGIC_RGB 0x20424752          /* "RGB " */
#define MQ_JPEG_MAX_PIXELS (1 << 26)
int mq_jpeg_probe(const uint8_t *file_host, size_t nbytes, int64_t *info_host);
int mq_jpeg_read_coefficients(const uint8_t *file_host, size_t nbytes, void *staging_host, size_t staging_cap);
int mq_jpeg_decode_rgb_u8(uint8_t *buf_dev, const int64_t *items_dev, int n_images, int max_blocks, int64_t max_strips,
                          void *stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (csrc/diag.hip; no reference counterpart, not on the product path): `workgroups` x 16 waves loop over
 * `iters` x 16 v_mfma_f32_32x32x16_bf16 on register operands (zero, or N(0,1)-like random when random_operands != 0), no
 * memory traffic: 2*32*32*16 * 16 * iters * 16 * workgroups FLOP per launch.  bench.py times it for the matrix-pipe rate
 * this chip SUSTAINS on such operands (its power management lowers the clock on random data).  out_dev: workgroups * 1024
 * floats (keeps the accumulators observable).
 * ------------------------------------------------------------------------------------------- */
int mq_diag_mfma_bf16_loop(int iters, int random_operands, int workgroups, float *out_dev, void *stream);

/* What the matrix instruction RETURNS: out_dev[32][32] fp32 = A . B^T for A_dev, B_dev = 32 rows x dp bf16 (row-major, dp a
 * multiple of 16), one v_mfma_f32_32x32x16_bf16 per 16 columns in column order, each chained on the previous result -- the
 * accumulation of csrc/knn_screen.inc's screening scan.  tests/test_screened_gpu.py bounds its distance to the float64 sum of
 * the same bf16 operands by the accumulation term of the screening margin (the one term of the bound that rests on the
 * instruction's undocumented internal summation). */
int mq_diag_mfma_bf16_dot(const uint16_t *A_dev, const uint16_t *B_dev, int dp, float *out_dev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * ArcFace r50 face encoder (meerqat/image/face_recognition.py:44-102; insightface arcface_torch IResNet-50 and OpenCV /
 * scikit-image are un-vendored and not installable here: parity unpinned, oracle/arcface.py restates the published algorithms).
 * A convolution = mq_im2col_split_f32 (patch gather + the layer's elementwise pre-operations -> the GEMM's split pair) followed
 * by mq_gemm_nt_bf16x3s_f32 (BatchNorms behind the convolution folded into W / bias, residual in the epilogue); NHWC.
 *   x_dev   fp32 activation, NHWC [B, H, W, C] (nchw != 0: NCHW [B, C, H, W], the network's input)
 *   A       rows = B * Ho * Wo output pixels (Ho = (H + 2 pad - KH) / stride + 1), column (kh * KW + kw) * C + c, zero beyond
 *           KH * KW * C up to Kpad (a multiple of 32); PAIR LAYOUT, mq_split_bf16_tiled_elems(rows, Kpad) elements per array
 *   pre-operations on every in-bounds element, in this order: PReLU (prelu_slope_dev [C] or NULL), then x * scale + shift
 *           (scale_dev / shift_dev [C], both or neither: an eval-mode BatchNorm in front of the convolution); padding is zero.
 * mq_warp_affine_faces_f32: cv2.warpAffine(image, M, (size, size), borderValue = 0) in OpenCV's fixed-point bilinear
 * arithmetic + ToTensor + Normalize(0.5, 0.5) for `nfaces` faces.  images_dev: the decoded RGB images of a batch packed in one
 * uint8 buffer (image i = H_i x W_i x 3 at byte offsets_dev[i], hw_dev[2 i] = H_i, hw_dev[2 i + 1] = W_i); face f is cut from image
 * face_image_dev[f] with the INVERTED 2 x 3 matrix minv_dev[6 f ..] (doubles); out_dev fp32 [nfaces, 3, size, size].
 * ------------------------------------------------------------------------------------------- */
int mq_im2col_split_f32(const float *x_dev, int B, int H, int W, int C, int nchw, int KH, int KW, int stride, int pad,
                        const float *prelu_slope_dev, const float *scale_dev, const float *shift_dev, uint16_t *Ah_dev,
                        uint16_t *Al_dev, int Kpad, void *stream);
/* A 3 x 3 convolution (padding 1, stride 1 or 2) as an IMPLICIT GEMM: the patch matrix is never written -- the GEMM's LDS-DMA
 * gathers its A stage from the input pair (zeros at the border) -- and the layer's elementwise operations ride on the epilogue of
 * the convolution that PRODUCES a tensor instead of on the im2col that consumes it.  Same products in the same order as
 * mq_im2col_split_f32 + mq_gemm_nt_bf16x3s_f32: the same bits.
 *   Xh / Xl       the convolution's input AFTER its pre-operations, as a split pair in PAIR LAYOUT over [B * H * W, C] (NHWC; C a
 *                 multiple of 32): what mq_im2col_split_f32 with KH = KW = 1 or a previous call of this function wrote
 *   Wh / Wl       mq_split_bf16_tiled_f32 of W [N, 9 C], column (kh * 3 + kw) * C + c (N a multiple of 64); bias_dev [N]
 *   prelu_slope_dev != NULL:  P = split(prelu(conv + bias))                       (Y, residual, scale, shift must be NULL)
 *   prelu_slope_dev == NULL:  Y = conv + bias + residual (fp32 [M, N], M = B * Ho * Wo, Ho = (H - 1) / stride + 1), and, when
 *                 scale_dev / shift_dev [N] are given, P = split(Y * scale + shift) (the BatchNorm in front of the NEXT convolution)
 *   Ph / Pl       pair output, PAIR LAYOUT over [M, N], mq_split_bf16_tiled_elems(M, N) elements each
 *   zeros_dev     at least 16 bytes of zeros on the device (the padded border's source)
 *   tile          MQ_CONV_TILE_AUTO, or one of the workgroup tiles (rows x columns; the column count must divide N) */
#define MQ_CONV_TILE_AUTO 0
#define MQ_CONV_TILE_256x256 1
#define MQ_CONV_TILE_512x128 2
#define MQ_CONV_TILE_256x128 3
#define MQ_CONV_TILE_512x64 4
#define MQ_CONV_TILE_PATCH_256x64 5 /* stride 1, W <= 127, with MQ_CONV_K_CHANNEL_MAJOR: the nine taps of a channel block read one
                                     * LDS-resident input patch (256 + 2 W + 2 pixels) instead of nine gathers; same bits */
/* OR into `tile`: walk K as (32-channel block, tap) instead of (tap, block) -- the nine taps of a block re-read the same input
 * lines back to back, so a workgroup's live footprint in L2 is one block's rows instead of all C channels'.  The sum order then
 * differs from the explicit path's (equal within fp32 rounding, not bit for bit). */
#define MQ_CONV_K_CHANNEL_MAJOR 0x100
int mq_conv3x3_pair_f32(const uint16_t *Xh_dev, const uint16_t *Xl_dev, int B, int H, int W, int C, int stride,
                        const uint16_t *Wh_dev, const uint16_t *Wl_dev, int N, const float *bias_dev,
                        const float *prelu_slope_dev, const float *residual_dev, const float *scale_dev,
                        const float *shift_dev, float *Y_dev, uint16_t *Ph_dev, uint16_t *Pl_dev, const void *zeros_dev,
                        int tile, void *stream);
/* The stem of IResNet: 3 -> 64 channels, 3 x 3, stride 1, padding 1, as a direct fp32 convolution (K = 27 has no matrix-pipe shape),
 * fused with what follows it: v = prelu(conv(x) + bias);
 *   P = split(v * scale + shift)  PAIR LAYOUT over [B * H * W, 64]: the first block's conv1 input (its bn1 applied)
 *   D = split(v) at even (h, w)   PAIR LAYOUT over [B * H/2 * W/2, 64]: the A operand of the first block's strided 1 x 1 downsample
 *                                 (Dh / Dl may both be NULL)
 * x_dev fp32 NCHW [B, 3, H, W] (W a multiple of 4, H even when D is asked for); wt_dev fp32 [27, 64]: row (kh * 3 + kw) * 3 + c,
 * the BatchNorm behind the convolution folded in; bias / prelu_slope / scale / shift fp32 [64]. */
int mq_stem_conv3x3_f32(const float *x_dev, int B, int H, int W, const float *wt_dev, const float *bias_dev,
                        const float *prelu_slope_dev, const float *scale_dev, const float *shift_dev, uint16_t *Ph_dev,
                        uint16_t *Pl_dev, uint16_t *Dh_dev, uint16_t *Dl_dev, void *stream);
int mq_warp_affine_faces_f32(const uint8_t *images_dev, const int64_t *offsets_dev, const int32_t *hw_dev,
                             const int32_t *face_image_dev, const double *minv_dev, int nfaces, int size, float *out_dev,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MEERQAT_HIP_H */
