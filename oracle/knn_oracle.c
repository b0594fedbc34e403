/*
 * oracle/knn_oracle.c -- CPU restatement of the brute-force kNN the reference
 * delegates to FAISS IndexFlat.  TEST INFRASTRUCTURE ONLY: imported by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
 * path (viquae_amd/ fails loudly when the HIP library is missing).
 *
 * PARITY STATUS: "parity unpinned" against FAISS itself (tie membership / order
 * inside equal-score runs and FAISS's BLAS summation order in particular).  The arithmetic lives
 * in the third-party module `faiss` (faiss-gpu>=1.7.1, /root/reference/
 * requirements.txt:14), which is neither vendored in /root/reference nor
 * installable here, and the reference's only test (tests/search.py:1-20) is
 * stale and pins nothing on this path.  What IS pinned: (i) the plumbing, by
 * running the reference's own KnowledgeBase.search_batch /
 * search_batch_if_not_None (meerqat/ir/search.py:135-171) over this restatement
 * (tools/make_golden.py); (ii) the result on integer-lattice data, where every
 * fp32 summation order gives the same scores, against an independent
 * torch.mm + stable argsort computation (tests/golden/knn_*.npz).
 *
 * Algorithm restated (call sites: meerqat/ir/search.py:146 ->
 * datasets/search.py:384 FaissIndex.search_batch -> faiss IndexFlat.search):
 *   - METRIC_INNER_PRODUCT (metric_type 0, experiments/ir/viquae/dpr/search/
 *     config.json:18): score = <q, x>, larger is better, rows sorted descending.
 *   - METRIC_L2 (metric_type 1 / FAISS default): squared distance, smaller is
 *     better, ascending.  FAISS switches form on the batch size
 *     (faiss/utils/distances.cpp, knn_L2sqr: `nx < distance_compute_blas_threshold`
 *     with the threshold's default 20):
 *       nq >= 20  (the reference searches 256 queries per batch,
 *                 dpr/search/config.json:25): the BLAS path evaluates
 *                 ||q||^2 + ||x||^2 - 2<q,x>, clamped at 0;
 *       nq <  20  (the last Dataset.map batch of a run, meerqat/ir/search.py:482,
 *                 or interact/system.py's single query): the sequential path sums
 *                 (q[k] - x[k])^2 directly -- no cancellation, no clamp.
 *     The two forms differ in the last bits of every distance and markedly near 0
 *     (near-duplicates), so both are restated; `l2_form` selects (0 = FAISS's rule).
 *   - a result list per query, initialised with (-FLT_MAX | +FLT_MAX, id -1)
 *     (faiss heap neutral values: CMin::neutral() = lowest(), CMax::neutral() =
 *     max()); a score must beat the current k-th best STRICTLY to enter
 *     (FAISS: `if (C::cmp(threshold, dis))`), so NaN, +-inf on the wrong side and
 *     the neutral value itself never enter; unfilled slots keep id -1 and
 *     +-FLT_MAX (3.4028235e+38, what FAISS prints when k > ntotal).
 *   - TIES ARE THIS LIBRARY'S DOCUMENTED POLICY, NOT FAISS'S.  Which of several
 *     rows with EXACTLY equal scores is kept at the k-th boundary, and in which
 *     order an equal-score run is reported, depends in FAISS on the version and on
 *     k (stated from FAISS's public sources as best known -- no copy of FAISS
 *     exists in this image, nothing here can check it):
 *       faiss 1.7.1 - 1.7.2 (the reference pins >= 1.7.1): value-only binary heaps
 *         (HeapResultHandler / heap_replace_top compare `val` alone); membership
 *         and order inside a tie follow the heap's sift pattern -- deterministic
 *         but not a function of the ids;
 *       faiss >= 1.7.3: the heaps sift with C::cmp2(val1, val2, id1, id2); for inner
 *         product (CMin) the heap top is the smallest (score, id), so among entries
 *         tied at the current minimum the LOWEST id is evicted first and
 *         heap_reorder emits equal scores by DESCENDING id; for L2 (CMax) the
 *         largest (distance, id) is evicted first and equal distances come out by
 *         ASCENDING id.  A newcomer still needs a strictly better VALUE than the
 *         threshold to enter, whatever its id;
 *       k >= distance_compute_min_k_reservoir (= 100, the reference's k) on the
 *         BLAS path (>= 20 queries): ReservoirBlockResultHandler -- candidates
 *         strictly better than the threshold are appended to a 2k-entry reservoir
 *         that is cut back to k with partition_fuzzy when full; membership among
 *         boundary ties is whatever that partition leaves, and the final order comes
 *         from the heap the reservoir is poured into.
 *     This file implements ONE total order, parameterised by `tie_order`:
 *       0 (id_asc, default): better = higher score, then LOWER id.  Equal scores are
 *         reported by ascending id and the lowest ids win membership at the k-th
 *         boundary.  (FAISS >= 1.7.3's L2 heap order; pre-1.7.3 folklore for IP.)
 *       1 (id_desc): better = higher score, then HIGHER id (FAISS >= 1.7.3's
 *         inner-product heap ORDER; note FAISS's membership rule is still "strictly
 *         better value", which no total order on (score, id) reproduces for k-th
 *         boundary ties -- a user who needs FAISS's exact tie sets must compare
 *         equal-score runs as sets).
 *     The HIP library implements the same two orders (MQ_KNN_FLAG_TIE_ID_DESC);
 *     bit-exact tests compare like with like.  Wherever a test or document says
 *     "FAISS" about ties, it means "unverified"; equal-score runs should be compared
 *     as sets against any real FAISS build.
 *
 * Summation order (the one thing FAISS leaves to BLAS): every inner product and
 * squared norm here is the k-ordered fp32 chain acc = fmaf(a[k], b[k], acc),
 * k = 0..d-1, acc0 = +0.  This is bit-for-bit what the gfx950
 * v_mfma_f32_32x32x2_f32 instruction computes when K is walked in order
 * (cdna_hip_programming.md section 3), so the HIP path is compared BIT-EXACTLY
 * (scores and ids) with this file on arbitrary fp32 data.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QB 32 /* queries per register block */

typedef struct {
    float g; /* goodness: ip, or -dist for L2 */
    int64_t id;
} ent_t;

/* k-ordered fmaf chain: sum_k a[k]*b[k] */
static float chain_dot(const float *a, const float *b, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

/* rows[i] /= sqrt(sum_k rows[i][k]^2): the "L2norm," prefix of the reference's
 * string_factory (meerqat/ir/search.py:230-233; FAISS NormalizationTransform) and
 * L2norm() (meerqat/ir/search.py:43-46).  Norm^2 is the fmaf chain, sqrtf and the
 * division are IEEE correctly rounded.  Like the reference there is no epsilon: a
 * zero row becomes NaN. */
void oracle_l2norm_rows_f32(float *rows, int64_t n, int d) {
    for (int64_t i = 0; i < n; ++i) {
        float *r = rows + i * (int64_t)d;
        float nr = sqrtf(chain_dot(r, r, d));
        for (int k = 0; k < d; ++k) r[k] = r[k] / nr;
    }
}

/* The same prefix in FAISS's own arithmetic: NormalizationTransform::apply_noalloc -> fvec_renorm_L2
 * (faiss/utils/distances.cpp, as published, un-vendored faiss-gpu>=1.7.1):
 *     float nr = fvec_norm_L2sqr(xi, d);
 *     if (nr > 0) { const float inv_nr = 1.0 / sqrtf(nr); for (j...) xi[j] *= inv_nr; }
 * i.e. ONE reciprocal per row -- a double division (the literal 1.0 is a double) rounded to float -- and a multiplication per
 * element, and a row whose squared norm is not > 0 (a zero row, an underflowed one, NaN) is left exactly as it is: it stays
 * retrievable with score 0 where the numpy form above turns it into NaN.  This is what the reference runs for KB rows and,
 * inside the index, for queries when "L2norm,Flat" meets `device: null` (every shipped config: meerqat/ir/search.py:230-245);
 * the numpy form is its GPU work-around (:238-244) and its host-side L2norm() (:43-46).  fvec_norm_L2sqr's SIMD summation
 * order is not restated (unknowable here): the squared norm is the k-ordered fmaf chain, like everywhere in this oracle. */
void oracle_l2norm_rows_faiss_f32(float *rows, int64_t n, int d) {
    for (int64_t i = 0; i < n; ++i) {
        float *r = rows + i * (int64_t)d;
        const float nr = chain_dot(r, r, d);
        if (nr > 0) {
            const float inv_nr = (float)(1.0 / (double)sqrtf(nr));
            for (int k = 0; k < d; ++k) r[k] *= inv_nr;
        }
    }
}

void oracle_sqnorm_rows_f32(const float *rows, int64_t n, int d, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = chain_dot(rows + i * (int64_t)d, rows + i * (int64_t)d, d);
}

/* insert (g,id) into a best-first sorted list of length k (caller checked that (g,id) beats
 * list[k-1] or that the slot is empty).  ids arrive ascending: under id_asc an equal-score
 * newcomer goes AFTER the existing ones, under id_desc BEFORE them. */
static inline void list_insert(ent_t *list, int k, float g, int64_t id, int tie_desc) {
    int p = k - 1;
    while (p > 0 && (list[p - 1].id < 0 || list[p - 1].g < g || (tie_desc && list[p - 1].g == g))) {
        list[p] = list[p - 1];
        --p;
    }
    list[p].g = g;
    list[p].id = id;
}

/*
 * metric: 0 = inner product, 1 = squared L2.
 * X [N,d] row-major, Q [nq,d] row-major, D [nq,k] fp32, I [nq,k] int64.
 * ids are reported as row + id_offset.  Returns 0, or -1 on bad arguments.
 */
#define FAISS_BLAS_THRESHOLD 20 /* faiss::distance_compute_blas_threshold */

/* l2_form (metric 1 only): 0 = FAISS's rule (direct below 20 queries, expanded otherwise),
 * 1 = expanded ||q||^2 + ||x||^2 - 2<q,x> clamped at 0, 2 = direct sum of (q-x)^2. */
int oracle_knn_f32_ex2(const float *X, int64_t N, int d, const float *Q, int nq, int k, int metric, int l2_form,
                       int tie_order, int64_t id_offset, float *D, int64_t *I) {
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || (metric != 0 && metric != 1)) return -1;
    if (l2_form < 0 || l2_form > 2 || tie_order < 0 || tie_order > 1) return -1;
    const int tie_desc = tie_order == 1;
    const int direct = metric == 1 && (l2_form == 2 || (l2_form == 0 && nq < FAISS_BLAS_THRESHOLD));
    int nblk = (nq + QB - 1) / QB;
    float *xn = NULL;
    if (metric == 1 && !direct) {
        xn = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
        oracle_sqnorm_rows_f32(X, N, d, xn);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < nblk; ++b) {
        int q0 = b * QB, nb = nq - q0 < QB ? nq - q0 : QB;
        /* Qt[k][j]: the block's queries, k-major, so the j loop vectorises while every
         * lane keeps its own k-ordered chain */
        float *Qt = (float *)calloc((size_t)d * QB, sizeof(float));
        ent_t *lists = (ent_t *)malloc(sizeof(ent_t) * (size_t)k * QB);
        float qn[QB], thr[QB];
        for (int j = 0; j < nb; ++j) {
            for (int kk = 0; kk < d; ++kk) Qt[(size_t)kk * QB + j] = Q[(size_t)(q0 + j) * d + kk];
            qn[j] = (metric == 1 && !direct) ? chain_dot(Q + (size_t)(q0 + j) * d, Q + (size_t)(q0 + j) * d, d) : 0.0f;
        }
        for (int j = 0; j < QB; ++j) {
            thr[j] = -FLT_MAX;
            for (int s = 0; s < k; ++s) {
                lists[(size_t)j * k + s].g = -FLT_MAX;
                lists[(size_t)j * k + s].id = -1;
            }
        }
        for (int64_t i = 0; i < N; ++i) {
            const float *x = X + i * (int64_t)d;
            float acc[QB];
            for (int j = 0; j < QB; ++j) acc[j] = 0.0f;
            if (direct) {
                /* faiss fvec_L2sqr: tmp = x[i] - y[i]; res += tmp * tmp (x = query, y = database row) */
                for (int kk = 0; kk < d; ++kk) {
                    const float xv = x[kk];
                    const float *qr = Qt + (size_t)kk * QB;
                    for (int j = 0; j < QB; ++j) {
                        const float t = qr[j] - xv;
                        acc[j] = fmaf(t, t, acc[j]);
                    }
                }
                for (int j = 0; j < QB; ++j) acc[j] = -acc[j];
            } else {
                for (int kk = 0; kk < d; ++kk) {
                    const float xv = x[kk];
                    const float *qr = Qt + (size_t)kk * QB;
                    for (int j = 0; j < QB; ++j) acc[j] = fmaf(xv, qr[j], acc[j]);
                }
            }
            if (metric == 1 && !direct) {
                for (int j = 0; j < QB; ++j) {
                    float dis = (qn[j] + xn[i]) - 2.0f * acc[j];
                    if (dis < 0.0f) dis = 0.0f;
                    acc[j] = -dis;
                }
            }
            for (int j = 0; j < nb; ++j) {
                /* strict against the k-th best (false for NaN; the neutral value never enters); under id_desc a later
                 * (higher) id with the k-th best's score beats it */
                ent_t *l = lists + (size_t)j * k;
                if (acc[j] > thr[j] || (tie_desc && acc[j] == thr[j] && l[k - 1].id >= 0)) {
                    list_insert(l, k, acc[j], i, tie_desc);
                    thr[j] = l[k - 1].id < 0 ? -FLT_MAX : l[k - 1].g;
                }
            }
        }
        for (int j = 0; j < nb; ++j) {
            for (int s = 0; s < k; ++s) {
                const ent_t e = lists[(size_t)j * k + s];
                float out;
                if (e.id < 0)
                    out = metric == 1 ? FLT_MAX : -FLT_MAX;
                else
                    out = metric == 1 ? -e.g : e.g;
                D[(size_t)(q0 + j) * k + s] = out + 0.0f; /* -0 -> +0 */
                I[(size_t)(q0 + j) * k + s] = e.id < 0 ? -1 : e.id + id_offset;
            }
        }
        free(Qt);
        free(lists);
    }
    free(xn);
    return 0;
}

int oracle_knn_f32_ex(const float *X, int64_t N, int d, const float *Q, int nq, int k, int metric, int l2_form,
                      int64_t id_offset, float *D, int64_t *I) {
    return oracle_knn_f32_ex2(X, N, d, Q, nq, k, metric, l2_form, 0, id_offset, D, I);
}

int oracle_knn_f32(const float *X, int64_t N, int d, const float *Q, int nq, int k, int metric,
                   int64_t id_offset, float *D, int64_t *I) {
    return oracle_knn_f32_ex(X, N, d, Q, nq, k, metric, 0, id_offset, D, I);
}

/*
 * Merge per-shard results (the new multi-GPU step): Ds/Is are
 * [nshards, nq, k] best-first lists with global ids; output the k best per query,
 * ordered by (score better first, id ascending); id -1 entries are empty.
 */
int oracle_topk_merge_ex(const float *Ds, const int64_t *Is, int nshards, int nq, int k, int metric, int tie_order,
                         float *D, int64_t *I) {
    if (nshards <= 0 || nq < 0 || k <= 0 || tie_order < 0 || tie_order > 1) return -1;
    for (int q = 0; q < nq; ++q) {
        int *pos = (int *)calloc((size_t)nshards, sizeof(int));
        for (int s = 0; s < k; ++s) {
            int best = -1;
            float bg = 0;
            int64_t bid = 0;
            for (int h = 0; h < nshards; ++h) {
                if (pos[h] >= k) continue;
                size_t o = ((size_t)h * nq + q) * k + pos[h];
                if (Is[o] < 0) continue;
                float g = metric == 1 ? -Ds[o] : Ds[o];
                if (best < 0 || g > bg || (g == bg && (tie_order ? Is[o] > bid : Is[o] < bid))) {
                    best = h;
                    bg = g;
                    bid = Is[o];
                }
            }
            if (best < 0) {
                D[(size_t)q * k + s] = metric == 1 ? FLT_MAX : -FLT_MAX;
                I[(size_t)q * k + s] = -1;
            } else {
                D[(size_t)q * k + s] = (metric == 1 ? -bg : bg) + 0.0f;
                I[(size_t)q * k + s] = bid;
                pos[best]++;
            }
        }
        free(pos);
    }
    return 0;
}

int oracle_topk_merge(const float *Ds, const int64_t *Is, int nshards, int nq, int k, int metric, float *D,
                      int64_t *I) {
    return oracle_topk_merge_ex(Ds, Is, nshards, nq, k, metric, 0, D, I);
}

/*
 * FAISS's ORGANISATION of the search for >= 20 queries (faiss/utils/distances.cpp,
 * exhaustive_inner_product_blas / exhaustive_L2sqr_blas): blocks of
 * distance_compute_blas_query_bs = 4096 queries x distance_compute_blas_database_bs = 1024 rows go
 * through sgemm, then a result handler walks each query's row of the block in ascending database
 * order and lets a score in iff it beats the query's current k-th best strictly
 * (HeapBlockResultHandler::add_results: `if (C::cmp(thresh, dis)) heap_replace_top(...)`).
 * The caller (oracle/knn.py::knn_blas) runs the sgemm on the host BLAS and hands the block here;
 * this is the CPU baseline's fast leg, not the bit-exact checker (its scores carry the BLAS's
 * summation order).  Binary heap per query, top = worst entry under THIS library's id_asc order
 * (smaller goodness, then larger id), so that the final membership equals oracle_knn_f32's on
 * exact ties; a newcomer needs a strictly better value, like FAISS.
 *   S   [nq][ld] goodness (inner product, or -distance) of database rows j0 .. j0 + nb - 1
 *   hv / hi [nq][k] heap storage, prepared by oracle_heap_init
 */
static inline int heap_worse(float ga, int64_t ia, float gb, int64_t ib) { return ga < gb || (ga == gb && ia > ib); }

void oracle_heap_init(int nq, int k, float *hv, int64_t *hi) {
    for (size_t e = 0; e < (size_t)nq * k; ++e) {
        hv[e] = -FLT_MAX;
        hi[e] = -1;
    }
}

void oracle_heap_add_block(const float *S, int nq, int64_t ld, int64_t j0, int nb, int k, float *hv, int64_t *hi) {
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        const float *s = S + (size_t)q * ld;
        float *v = hv + (size_t)q * k;
        int64_t *id = hi + (size_t)q * k;
        float thr = v[0];
        for (int j = 0; j < nb; ++j) {
            const float g = s[j];
            if (!(g > thr)) continue; /* strict; false for NaN */
            /* replace the top (worst) and sift down */
            int p = 0;
            const int64_t gid = j0 + j;
            for (;;) {
                int c = 2 * p + 1;
                if (c >= k) break;
                if (c + 1 < k && heap_worse(v[c + 1], id[c + 1] < 0 ? INT64_MAX : id[c + 1], v[c], id[c] < 0 ? INT64_MAX : id[c])) ++c;
                if (!heap_worse(v[c], id[c] < 0 ? INT64_MAX : id[c], g, gid)) break;
                v[p] = v[c];
                id[p] = id[c];
                p = c;
            }
            v[p] = g;
            id[p] = gid;
            thr = v[0];
        }
    }
}

/* oracle_heap_add_block over the panel-major score block of oracle_sgemm_nt (S[panel][nq][32]; columns >= nb are padding) */
void oracle_heap_add_block_panels(const float *S, int nq, int64_t j0, int nb, int k, float *hv, int64_t *hi) {
    const int npan = (nb + 31) / 32;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        float *v = hv + (size_t)q * k;
        int64_t *id = hi + (size_t)q * k;
        float thr = v[0];
        for (int pn = 0; pn < npan; ++pn) {
            const float *s = S + ((size_t)pn * (size_t)nq + (size_t)q) * 32;
            const int nc = nb - pn * 32 < 32 ? nb - pn * 32 : 32;
            for (int j = 0; j < nc; ++j) {
                const float g = s[j];
                if (!(g > thr)) continue;
                int p = 0;
                const int64_t gid = j0 + pn * 32 + j;
                for (;;) {
                    int c = 2 * p + 1;
                    if (c >= k) break;
                    if (c + 1 < k && heap_worse(v[c + 1], id[c + 1] < 0 ? INT64_MAX : id[c + 1], v[c], id[c] < 0 ? INT64_MAX : id[c])) ++c;
                    if (!heap_worse(v[c], id[c] < 0 ? INT64_MAX : id[c], g, gid)) break;
                    v[p] = v[c];
                    id[p] = id[c];
                    p = c;
                }
                v[p] = g;
                id[p] = gid;
                thr = v[0];
            }
        }
    }
}

static int ent_cmp_best_first(const void *a, const void *b) {
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (x->id < 0 || y->id < 0) return (x->id < 0) - (y->id < 0);
    if (x->g != y->g) return x->g > y->g ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/* heap_reorder: best first (equal scores by ascending id), neutral values in the unfilled slots */
void oracle_heap_finish(int nq, int k, int metric, const float *hv, const int64_t *hi, float *D, int64_t *I) {
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; ++q) {
        ent_t *e = (ent_t *)malloc(sizeof(ent_t) * (size_t)k);
        for (int s = 0; s < k; ++s) {
            e[s].g = hv[(size_t)q * k + s];
            e[s].id = hi[(size_t)q * k + s];
        }
        qsort(e, (size_t)k, sizeof(ent_t), ent_cmp_best_first);
        for (int s = 0; s < k; ++s) {
            if (e[s].id < 0) {
                D[(size_t)q * k + s] = metric == 1 ? FLT_MAX : -FLT_MAX;
                I[(size_t)q * k + s] = -1;
            } else {
                D[(size_t)q * k + s] = (metric == 1 ? -e[s].g : e[s].g) + 0.0f;
                I[(size_t)q * k + s] = e[s].id;
            }
        }
        free(e);
    }
}

/*
 * A host sgemm for the FAISS-organised leg: S[nq][nb] = Q[nq][d] . X[nb][d]^T (+ FAISS's L2 epilogue in the caller).
 * The two BLAS libraries this image ships reach 0.5-0.9 TFLOP/s on the GPU box's 128 EPYC cores (MKL takes its non-Intel
 * code path, numpy's OpenBLAS stops at 64 threads), which would understate what FAISS + a tuned BLAS does there; this is a
 * plain register-blocked kernel -- 14 x 32 accumulators in AVX-512 registers (6 x 16 in AVX2), operands packed per panel,
 * one OpenMP thread per 32-column panel of X -- good for a few TFLOP/s.  Same role as the BLAS call: scores carry ITS
 * summation order (k ascending per accumulator, fused multiply-add), not the fmaf chain's bits... they are in fact the same
 * chain per output, but nothing relies on it.
 *   Qp: queries packed by oracle_pack_queries ([ceil(nq / MR)][d][MR], zero padded); S row stride ld, or ld < 0 for the
 *   PANEL-MAJOR layout S[ceil(nb / 32)][nq][32] (what a thread computes is then contiguous: row-major S makes every
 *   14 x 32 micro-tile a set of 128-byte pieces one row stride apart, which thrashes the L1 sets -- measured 67 GFLOP/s at a
 *   64 KB stride against 669 at 4 KB); oracle_heap_add_block_panels reads that layout.
 */
#include <immintrin.h>

static int have_avx512(void) { return __builtin_cpu_supports("avx512f"); }
int oracle_sgemm_mr(void) { return have_avx512() ? 14 : 6; }

void oracle_pack_queries(const float *Q, int nq, int d, float *Qp) {
    const int MR = oracle_sgemm_mr();
    const int np = (nq + MR - 1) / MR;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < np; ++p) {
        float *dst = Qp + (size_t)p * d * MR;
        for (int k = 0; k < d; ++k)
            for (int i = 0; i < MR; ++i) {
                const int q = p * MR + i;
                dst[(size_t)k * MR + i] = q < nq ? Q[(size_t)q * d + k] : 0.0f;
            }
    }
}

__attribute__((target("avx512f"))) static void panel_avx512(const float *Qp, int nq, int d, const float *Bp, float *S, int64_t ld,
                                                            int j0, int ncols) {
    const int MR = 14;
    const int np = (nq + MR - 1) / MR;
    for (int p = 0; p < np; ++p) {
        const float *a = Qp + (size_t)p * d * MR;
        __m512 c0[14], c1[14];
        for (int i = 0; i < 14; ++i) { c0[i] = _mm512_setzero_ps(); c1[i] = _mm512_setzero_ps(); }
        for (int k = 0; k < d; ++k) {
            const __m512 b0 = _mm512_loadu_ps(Bp + (size_t)k * 32), b1 = _mm512_loadu_ps(Bp + (size_t)k * 32 + 16);
            const float *ak = a + (size_t)k * MR;
#pragma GCC unroll 14
            for (int i = 0; i < 14; ++i) {
                const __m512 av = _mm512_set1_ps(ak[i]);
                c0[i] = _mm512_fmadd_ps(av, b0, c0[i]);
                c1[i] = _mm512_fmadd_ps(av, b1, c1[i]);
            }
        }
        for (int i = 0; i < 14; ++i) {
            const int q = p * MR + i;
            if (q >= nq) break;
            if (ld < 0) { /* panel-major output: S[panel][q][32], contiguous per thread */
                float *dst = S + ((size_t)(j0 / 32) * (size_t)nq + (size_t)q) * 32;
                _mm512_storeu_ps(dst, c0[i]);
                _mm512_storeu_ps(dst + 16, c1[i]);
                continue;
            }
            float tmp[32];
            _mm512_storeu_ps(tmp, c0[i]);
            _mm512_storeu_ps(tmp + 16, c1[i]);
            memcpy(S + (size_t)q * ld + j0, tmp, sizeof(float) * (size_t)ncols);
        }
    }
}

__attribute__((target("avx2,fma"))) static void panel_avx2(const float *Qp, int nq, int d, const float *Bp, float *S, int64_t ld,
                                                           int j0, int ncols) {
    const int MR = 6;
    const int np = (nq + MR - 1) / MR;
    for (int p = 0; p < np; ++p) {
        const float *a = Qp + (size_t)p * d * MR;
        for (int h = 0; h < 2; ++h) { /* the 32-column panel as two 16-column halves */
            __m256 c0[6], c1[6];
            for (int i = 0; i < 6; ++i) { c0[i] = _mm256_setzero_ps(); c1[i] = _mm256_setzero_ps(); }
            for (int k = 0; k < d; ++k) {
                const __m256 b0 = _mm256_loadu_ps(Bp + (size_t)k * 32 + 16 * h), b1 = _mm256_loadu_ps(Bp + (size_t)k * 32 + 16 * h + 8);
                const float *ak = a + (size_t)k * MR;
#pragma GCC unroll 6
                for (int i = 0; i < 6; ++i) {
                    const __m256 av = _mm256_set1_ps(ak[i]);
                    c0[i] = _mm256_fmadd_ps(av, b0, c0[i]);
                    c1[i] = _mm256_fmadd_ps(av, b1, c1[i]);
                }
            }
            for (int i = 0; i < 6; ++i) {
                const int q = p * MR + i;
                if (q >= nq) break;
                if (ld < 0) {
                    float *dst = S + ((size_t)(j0 / 32) * (size_t)nq + (size_t)q) * 32 + 16 * h;
                    _mm256_storeu_ps(dst, c0[i]);
                    _mm256_storeu_ps(dst + 8, c1[i]);
                    continue;
                }
                float tmp[16];
                _mm256_storeu_ps(tmp, c0[i]);
                _mm256_storeu_ps(tmp + 8, c1[i]);
                const int n = ncols - 16 * h;
                if (n > 0) memcpy(S + (size_t)q * ld + j0 + 16 * h, tmp, sizeof(float) * (size_t)(n < 16 ? n : 16));
            }
        }
    }
}

void oracle_sgemm_nt(const float *Qp, int nq, int d, const float *X, int nb, float *S, int64_t ld) {
    const int npan = (nb + 31) / 32;
    const int wide = have_avx512();
#pragma omp parallel
    {
        float *Bp = (float *)aligned_alloc(64, sizeof(float) * (size_t)d * 32);
#pragma omp for schedule(dynamic, 1)
        for (int pn = 0; pn < npan; ++pn) {
            const int j0 = pn * 32, ncols = nb - j0 < 32 ? nb - j0 : 32;
            for (int k = 0; k < d; ++k)
                for (int j = 0; j < 32; ++j) Bp[(size_t)k * 32 + j] = j < ncols ? X[(size_t)(j0 + j) * d + k] : 0.0f;
            if (wide)
                panel_avx512(Qp, nq, d, Bp, S, ld, j0, ncols);
            else
                panel_avx2(Qp, nq, d, Bp, S, ld, j0, ncols);
        }
        free(Bp);
    }
}

/* how many OpenMP threads the following calls use (the FAISS-organised leg runs faster on ONE socket of a two-socket host:
 * 1.5-1.8 TFLOP/s on 64 cores against 0.7-1.2 on 128, whose second half reads the packed queries across the socket link) */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
