// fuse.hip -- late fusion of several retrieval runs on gfx950: the arithmetic of
// meerqat/ir/fuse.py `default_minimum` (:129-146) and `gzmuv_norm` (:86-126) followed by ranx's weighted
// sum (`fuse(..., method="wsum")`, meerqat/ir/fuse.py:225-230), kept on the device so that search results
// never become Python dicts between search and fusion.  C ABI: include/meerqat_hip.h.
//
// A run is a padded [nq, K] table of (document id int64, score f64), -1 = empty slot.  One workgroup owns one
// query: its R*K entries are sorted by (id, run) in LDS (bitonic), which makes every document a contiguous
// segment holding the runs that retrieved it, in run order:
//
//   fuse_stats_kernel    per (run, query): count, minimum, sum, size of the union over runs; and the per-query
//                        moments the "zmuv" norm needs (with the default-minimum fill counted in)
//   fuse_moments_kernel  per run: ONE mean / std over all its scores (the "gzmuv" norm), two-pass like np.std,
//                        fill entries counted in closed form ((union - count) copies of the minimum)
//   fuse_combine_kernel  per query: normalise, walk each segment in run order accumulating weight*score from
//                        0.0 (the order ranx's comb_sum adds them in), fill missing runs with their normalised
//                        minimum, sort by (fused score desc, id asc) and write the fused run
//
// All arithmetic is f64 without contraction (the library is built with -ffp-contract=off), so given the same
// moments the fused scores are bit-identical to the Python restatement; the moments themselves are sums in a
// different order than numpy's pairwise summation (agreement ~1e-15 relative, the tests state the tolerance).
// This path is byte/latency work on a few MB: no MFMA, no HBM pressure -- its point is residency, not FLOPs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/meerqat_hip.h"
#include "launch_attr.h"

extern "C" void mq_internal_set_hip_error(int e);

namespace {

#define FUSE_HIP(call)                                 \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) { mq_internal_set_hip_error((int)_e); return MQ_EHIP; } \
    } while (0)

constexpr int FT = 256;          // threads per query workgroup
constexpr int FCAP = 4096;       // most entries (R*K, rounded up to a power of two) one query may hold
constexpr int FPER = FCAP / FT;  // sorted positions per thread
constexpr int RUN_BITS = 5;      // run index packed under the id in the sort key
constexpr int STAT = 6;          // per (run, query): count, min, sum, union, zmuv mean, zmuv denominator
constexpr uint64_t EMPTY = ~0ull;

struct FuseArgs {
    const int64_t* ids;     // [R, nq, K]
    const double* scores;   // [R, nq, K]
    double* stats;          // [R, nq, STAT]
    double* moments;        // [R, 2] mean, denominator (gzmuv)
    int64_t* out_ids;       // [nq, R*K]
    double* out_scores;     // [nq, R*K]
    int32_t* out_count;     // [nq]
    int R, nq, K, nsort, norm, defmin;
    double w[MQ_FUSE_MAX_RUNS];
};

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

// ascending bitonic sort of nsort (a, b) pairs held in LDS, lexicographic on (a, b)
__device__ void bitonic_pairs(uint64_t* a, uint64_t* b, int nsort) {
    const int t = threadIdx.x;
    for (int k = 2; k <= nsort; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < nsort; i += FT) {
                int p = i ^ j;
                if (p > i) {
                    uint64_t ai = a[i], ap = a[p], bi = b[i], bp = b[p];
                    bool gt = ai > ap || (ai == ap && bi > bp);
                    bool up = (i & k) == 0;
                    if (gt == up) { a[i] = ap; a[p] = ai; b[i] = bp; b[p] = bi; }
                }
            }
            __syncthreads();
        }
    }
}

// loads one query's R*K entries as sort keys ((id << RUN_BITS) | run, payload); payload = f(run, slot, score)
template <typename F>
__device__ void load_entries(const FuseArgs& A, int q, uint64_t* a, uint64_t* b, F payload) {
    const int total = A.R * A.K;
    for (int i = threadIdx.x; i < A.nsort; i += FT) {
        uint64_t key = EMPTY, val = EMPTY;
        if (i < total) {
            int r = i / A.K, s = i - r * A.K;
            size_t at = ((size_t)r * A.nq + q) * A.K + s;
            int64_t id = A.ids[at];
            if (id >= 0) {
                key = ((uint64_t)id << RUN_BITS) | (uint64_t)r;
                val = payload(r, A.scores[at]);
            }
        }
        a[i] = key;
        b[i] = val;
    }
    __syncthreads();
}

__global__ __launch_bounds__(FT) void fuse_stats_kernel(FuseArgs A) {
    extern __shared__ uint64_t lds[];
    uint64_t* a = lds;
    uint64_t* b = lds + A.nsort;
    __shared__ int heads;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) heads = 0;
    load_entries(A, q, a, b, [](int, double s) { return (uint64_t)__double_as_longlong(s); });
    bitonic_pairs(a, b, A.nsort);
    int mine = 0;
    for (int i = t; i < A.nsort; i += FT)
        if (a[i] != EMPTY && (i == 0 || (a[i - 1] >> RUN_BITS) != (a[i] >> RUN_BITS))) ++mine;
    if (mine) atomicAdd(&heads, mine);
    __syncthreads();
    const double uni = (double)heads;
    // one wave per run: count / min / sum, then the per-query moments with the fill counted in
    for (int r = wave; r < A.R; r += FT / 64) {
        const size_t base = ((size_t)r * A.nq + q) * A.K;
        double n = 0.0, mn = INFINITY, sm = 0.0;
        for (int s = lane; s < A.K; s += 64)
            if (A.ids[base + s] >= 0) { double v = A.scores[base + s]; n += 1.0; mn = fmin(mn, v); sm += v; }
        n = wave_sum(n); mn = wave_min(mn); sm = wave_sum(sm);
        const double c = (A.defmin && n > 0.0) ? uni : n;   // entries after default_minimum
        double mean = 0.0, den = 1.0;
        if (c > 0.0) {
            mean = (sm + (c - n) * mn) / c;
            double dev = 0.0;
            for (int s = lane; s < A.K; s += 64)
                if (A.ids[base + s] >= 0) { double d = A.scores[base + s] - mean; dev += d * d; }
            dev = wave_sum(dev);
            if (c > n) { double d = mn - mean; dev += (c - n) * (d * d); }
            den = fmax(sqrt(dev / c), 1e-9);
        }
        if (lane == 0) {
            double* st = A.stats + ((size_t)r * A.nq + q) * STAT;
            st[0] = n; st[1] = mn; st[2] = sm; st[3] = uni; st[4] = mean; st[5] = den;
        }
    }
}

__device__ double block_sum_1024(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    return tot;
}

__global__ __launch_bounds__(1024) void fuse_moments_kernel(FuseArgs A) {
    __shared__ double red[16];
    const int r = blockIdx.x, t = threadIdx.x;
    const double* st = A.stats + (size_t)r * A.nq * STAT;
    double tot = 0.0, cnt = 0.0;
    for (int q = t; q < A.nq; q += 1024) {
        double n = st[q * STAT], mn = st[q * STAT + 1], sm = st[q * STAT + 2], uni = st[q * STAT + 3];
        if (n > 0.0) {
            double c = A.defmin ? uni : n;
            tot += sm + (c - n) * mn;
            cnt += c;
        }
    }
    tot = block_sum_1024(tot, red);
    cnt = block_sum_1024(cnt, red);
    double mean = 0.0, den = 1.0;
    if (cnt > 0.0) {
        mean = tot / cnt;
        double dev = 0.0;
        const size_t base = (size_t)r * A.nq * A.K, total = (size_t)A.nq * A.K;
        for (size_t i = t; i < total; i += 1024)
            if (A.ids[base + i] >= 0) { double d = A.scores[base + i] - mean; dev += d * d; }
        for (int q = t; q < A.nq; q += 1024) {
            double n = st[q * STAT], mn = st[q * STAT + 1], uni = st[q * STAT + 3];
            if (n > 0.0 && A.defmin && uni > n) { double d = mn - mean; dev += (uni - n) * (d * d); }
        }
        dev = block_sum_1024(dev, red);
        den = fmax(sqrt(dev / cnt), 1e-9);
    }
    if (t == 0) { A.moments[2 * r] = mean; A.moments[2 * r + 1] = den; }
}

__global__ __launch_bounds__(FT) void fuse_combine_kernel(FuseArgs A) {
    extern __shared__ uint64_t lds[];
    uint64_t* a = lds;
    uint64_t* b = lds + A.nsort;
    __shared__ double r_mean[MQ_FUSE_MAX_RUNS], r_den[MQ_FUSE_MAX_RUNS], r_fill[MQ_FUSE_MAX_RUNS];
    __shared__ int r_has[MQ_FUSE_MAX_RUNS];
    __shared__ int heads;
    const int q = blockIdx.x, t = threadIdx.x;
    if (t == 0) heads = 0;
    if (t < A.R) {
        const double* st = A.stats + ((size_t)t * A.nq + q) * STAT;
        double mean = 0.0, den = 1.0;
        if (A.norm == MQ_FUSE_NORM_GZMUV) { mean = A.moments[2 * t]; den = A.moments[2 * t + 1]; }
        else if (A.norm == MQ_FUSE_NORM_ZMUV) { mean = st[4]; den = st[5]; }
        r_mean[t] = mean; r_den[t] = den;
        r_has[t] = st[0] > 0.0;
        r_fill[t] = st[0] > 0.0 ? (A.norm == MQ_FUSE_NORM_NONE ? st[1] : (st[1] - mean) / den) : 0.0;
    }
    __syncthreads();
    const int norm = A.norm;
    load_entries(A, q, a, b, [&](int r, double s) {
        double x = norm == MQ_FUSE_NORM_NONE ? s : (s - r_mean[r]) / r_den[r];
        return (uint64_t)__double_as_longlong(x);
    });
    bitonic_pairs(a, b, A.nsort);
    // segment heads accumulate their document's weighted sum, in run order, from 0.0
    uint64_t k1[FPER], k2[FPER];
    int mine = 0;
#pragma unroll
    for (int m = 0; m < FPER; ++m) {
        const int i = t + m * FT;
        k1[m] = EMPTY; k2[m] = EMPTY;
        if (i < A.nsort && a[i] != EMPTY && (i == 0 || (a[i - 1] >> RUN_BITS) != (a[i] >> RUN_BITS))) {
            const uint64_t id = a[i] >> RUN_BITS;
            int j = i;
            double acc = 0.0;
            for (int r = 0; r < A.R; ++r) {
                if (j < A.nsort && a[j] == ((id << RUN_BITS) | (uint64_t)r)) {
                    acc = acc + A.w[r] * __longlong_as_double((long long)b[j]);
                    ++j;
                } else if (A.defmin && r_has[r]) {
                    acc = acc + A.w[r] * r_fill[r];
                }
            }
            if (acc == 0.0) acc = 0.0;  // -0.0 and +0.0 rank equal
            uint64_t u = (uint64_t)__double_as_longlong(acc);
            u = (u >> 63) ? ~u : (u | (1ull << 63));  // ascending-orderable image of the double
            k1[m] = ~u;                                // ascending sort on ~u = best score first
            k2[m] = id;
            ++mine;
        }
    }
    if (mine) atomicAdd(&heads, mine);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < FPER; ++m) {
        const int i = t + m * FT;
        if (i < A.nsort) { a[i] = k1[m]; b[i] = k2[m]; }
    }
    __syncthreads();
    bitonic_pairs(a, b, A.nsort);
    const int count = heads, width = A.R * A.K;
    for (int i = t; i < width; i += FT) {
        int64_t id = -1;
        double s = 0.0;
        if (i < count) {
            uint64_t u = ~a[i];
            u = (u >> 63) ? (u ^ (1ull << 63)) : ~u;
            s = __longlong_as_double((long long)u);
            id = (int64_t)b[i];
        }
        A.out_ids[(size_t)q * width + i] = id;
        A.out_scores[(size_t)q * width + i] = s;
    }
    if (t == 0) A.out_count[q] = count;
}

// ------------------------------------------------------------------------------------------------------------
// Rank metrics of a run against qrels, and the weight search of `Fusion.fit` (meerqat/ir/fuse.py:193-217 ->
// ranx.optimize_fusion; the metric report of meerqat/ir/search.py:397,500-512 -> ranx.compare).  qrels are a CSR
// table: rel_ptr [nq + 1], rel_ids ascending inside one query (only the documents whose judgement is >= 1).
//   mrr@k       1 / (1 + rank of the first relevant document among the first k), else 0
//   precision@k (relevant among the first k) / k             -- k = 0: the length of the query's run
//   hit_rate@k  1 when a relevant document is among the first k
//   recall@k    (relevant among the first k) / (relevant documents of the query); 0 when it has none
// ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool is_relevant(const int64_t* rel, int64_t n, int64_t id) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        int64_t v = rel[mid];
        if (v < id) lo = mid + 1; else hi = mid;
    }
    return lo < n && rel[lo] == id;
}

// first: rank (0-based) of the best relevant document inside the cut, or -1; hits: relevant inside the cut
__device__ __forceinline__ double metric_value(int code, int k, int len, int first, int hits, int64_t n_rel) {
    int cut = k == 0 ? len : k;
    if (cut == 0) return 0.0;
    switch (code) {
        case MQ_RANK_METRIC_MRR: return first >= 0 ? 1.0 / (double)(first + 1) : 0.0;
        case MQ_RANK_METRIC_PRECISION: return (double)hits / (double)cut;
        case MQ_RANK_METRIC_HIT_RATE: return hits > 0 ? 1.0 : 0.0;
        default: return n_rel > 0 ? (double)hits / (double)n_rel : 0.0;
    }
}

struct MetricArgs {
    const int64_t* ids;      // [nq, K] best first, a row ends at its first negative id
    const int64_t* rel_ptr;  // [nq + 1]
    const int64_t* rel_ids;
    double* out;             // [n_metrics, nq]
    int nq, K, n_metrics;
    int code[MQ_RANK_MAX_METRICS], k[MQ_RANK_MAX_METRICS];
};

// one wave per query: relevance flags of the row as 64-bit ballots, every metric from the same flags
__global__ __launch_bounds__(256) void run_metrics_kernel(MetricArgs A) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= A.nq) return;
    const int64_t* row = A.ids + (size_t)q * A.K;
    const int64_t r0 = A.rel_ptr[q], n_rel = A.rel_ptr[q + 1] - r0;
    const int64_t* rel = A.rel_ids + r0;
    int len = A.K;
    for (int base = 0; base < A.K; base += 64) {
        int s = base + lane;
        bool dead = s < A.K && row[s] < 0;
        unsigned long long m = __ballot(dead);
        if (m) { len = base + __ffsll((long long)m) - 1; break; }
    }
    for (int mi = 0; mi < A.n_metrics; ++mi) {
        const int k = A.k[mi];
        const int cut = k == 0 ? len : (k < len ? k : len);
        int first = -1, hits = 0;
        for (int base = 0; base < cut; base += 64) {
            int s = base + lane;
            bool r = s < cut && is_relevant(rel, n_rel, row[s]);
            unsigned long long m = __ballot(r);
            if (m) {
                if (first < 0) first = base + __ffsll((long long)m) - 1;
                hits += __popcll(m);
            }
        }
        if (lane == 0) A.out[(size_t)mi * A.nq + q] = metric_value(A.code[mi], k, len, first, hits, n_rel);
    }
}

// np.mean of each row of v [rows, n] (numpy's pairwise summation, restated: blocks of <= 128 summed with eight
// interleaved accumulators, halves split on a multiple of 8), one thread per row -- the mean ranx.evaluate takes
__device__ double np_block_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

__global__ void np_mean_rows_kernel(const double* v, double* out, int rows, int n) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const double* a = v + (size_t)row * n;
    // iterative form of the recursion sum(a, n) = n <= 128 ? block : sum(a, h) + sum(a + h, n - h), h = n/2 - (n/2) % 8
    struct Frame { int off, n, state; double left; };
    Frame st[40];
    int sp = 0;
    st[0].off = 0; st[0].n = n; st[0].state = 0; st[0].left = 0.0;
    double ret = 0.0;
    while (sp >= 0) {
        Frame& f = st[sp];
        if (f.n <= 128) { ret = np_block_sum(a + f.off, f.n); --sp; continue; }
        int h = f.n / 2; h -= h % 8;
        if (f.state == 0) { f.state = 1; ++sp; st[sp].off = f.off; st[sp].n = h; st[sp].state = 0; }
        else if (f.state == 1) { f.left = ret; f.state = 2; ++sp; st[sp].off = f.off + h; st[sp].n = f.n - h; st[sp].state = 0; }
        else { ret = f.left + ret; --sp; }
    }
    out[row] = n > 0 ? ret / (double)n : 0.0;
}

struct FitArgs {
    FuseArgs F;
    const double* trials;    // [T, R] device
    const int64_t* rel_ptr;
    const int64_t* rel_ids;
    double* out;             // [T, nq]
    int T, code, k;
};

// one workgroup per query, all trials: the (id, run) sort and the normalisation happen once; a trial costs one weighted
// sum per document and, for the relevant documents only, a count of the documents ranked in front of them
// (fused score descending, then id ascending: the order fuse_combine_kernel writes)
__global__ __launch_bounds__(FT) void fuse_fit_kernel(FitArgs G) {
    const FuseArgs& A = G.F;
    extern __shared__ uint64_t lds[];
    uint64_t* a = lds;                       // sort keys (id, run)
    uint64_t* b = lds + A.nsort;             // normalised scores
    double* fs = (double*)(lds + 2 * A.nsort);        // fused score per head
    int* hpos = (int*)(lds + 3 * A.nsort);            // sorted position of head h
    int* rlist = hpos + A.nsort;                      // heads that are relevant
    __shared__ double r_mean[MQ_FUSE_MAX_RUNS], r_den[MQ_FUSE_MAX_RUNS], r_fill[MQ_FUSE_MAX_RUNS], w[MQ_FUSE_MAX_RUNS];
    __shared__ int r_has[MQ_FUSE_MAX_RUNS];
    __shared__ int heads, nrel_heads, red_first[FT / 64], red_hits[FT / 64];
    const int q = blockIdx.x, t = threadIdx.x;
    if (t == 0) { heads = 0; nrel_heads = 0; }
    if (t < A.R) {
        const double* st = A.stats + ((size_t)t * A.nq + q) * STAT;
        double mean = 0.0, den = 1.0;
        if (A.norm == MQ_FUSE_NORM_GZMUV) { mean = A.moments[2 * t]; den = A.moments[2 * t + 1]; }
        else if (A.norm == MQ_FUSE_NORM_ZMUV) { mean = st[4]; den = st[5]; }
        r_mean[t] = mean; r_den[t] = den;
        r_has[t] = st[0] > 0.0;
        r_fill[t] = st[0] > 0.0 ? (A.norm == MQ_FUSE_NORM_NONE ? st[1] : (st[1] - mean) / den) : 0.0;
    }
    __syncthreads();
    const int norm = A.norm;
    load_entries(A, q, a, b, [&](int r, double s) {
        double x = norm == MQ_FUSE_NORM_NONE ? s : (s - r_mean[r]) / r_den[r];
        return (uint64_t)__double_as_longlong(x);
    });
    bitonic_pairs(a, b, A.nsort);
    const int64_t r0 = G.rel_ptr[q], n_rel = G.rel_ptr[q + 1] - r0;
    const int64_t* rel = G.rel_ids + r0;
    // heads in sorted (= ascending id) order: head h sits at hpos[h]; since ids ascend with h, "id ascending" = "h ascending"
    for (int base = 0; base < A.nsort; base += FT) {
        const int i = base + t;
        const bool head = a[i] != EMPTY && (i == 0 || (a[i - 1] >> RUN_BITS) != (a[i] >> RUN_BITS));
        // stable compaction: positions before `base` are already numbered
        unsigned long long m = __ballot(head);
        __shared__ int wave_cnt[FT / 64];
        if ((t & 63) == 0) wave_cnt[t >> 6] = __popcll(m);
        __syncthreads();
        int before = heads;
        for (int wv = 0; wv < (t >> 6); ++wv) before += wave_cnt[wv];
        if (head) hpos[before + __popcll(m & ((1ull << (t & 63)) - 1ull))] = i;
        __syncthreads();
        if (t == 0) { int tot = 0; for (int wv = 0; wv < FT / 64; ++wv) tot += wave_cnt[wv]; heads += tot; }
        __syncthreads();
    }
    const int nh = heads;
    for (int h = t; h < nh; h += FT)
        if (is_relevant(rel, n_rel, (int64_t)(a[hpos[h]] >> RUN_BITS))) rlist[atomicAdd(&nrel_heads, 1)] = h;
    __syncthreads();
    const int nr = nrel_heads;
    const int cut = G.k == 0 ? nh : (G.k < nh ? G.k : nh);
    for (int trial = 0; trial < G.T; ++trial) {
        if (t < A.R) w[t] = G.trials[(size_t)trial * A.R + t];
        __syncthreads();
        for (int h = t; h < nh; h += FT) {
            const int i = hpos[h];
            const uint64_t id = a[i] >> RUN_BITS;
            int j = i;
            double acc = 0.0;
            for (int r = 0; r < A.R; ++r) {
                if (j < A.nsort && a[j] == ((id << RUN_BITS) | (uint64_t)r)) {
                    acc = acc + w[r] * __longlong_as_double((long long)b[j]);
                    ++j;
                } else if (A.defmin && r_has[r]) {
                    acc = acc + w[r] * r_fill[r];
                }
            }
            if (acc == 0.0) acc = 0.0;
            fs[h] = acc;
        }
        __syncthreads();
        int first = 0x7fffffff, hits = 0;
        for (int x = t; x < nr; x += FT) {
            const int h = rlist[x];
            const double s = fs[h];
            int rank = 0;
            for (int e = 0; e < nh; ++e) {
                const double se = fs[e];
                rank += (se > s || (se == s && e < h)) ? 1 : 0;
            }
            if (rank < cut) { ++hits; first = rank < first ? rank : first; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            int of = __shfl_xor(first, o), oh = __shfl_xor(hits, o);
            first = of < first ? of : first; hits += oh;
        }
        if ((t & 63) == 0) { red_first[t >> 6] = first; red_hits[t >> 6] = hits; }
        __syncthreads();
        if (t == 0) {
            for (int wv = 1; wv < FT / 64; ++wv) { first = red_first[wv] < first ? red_first[wv] : first; hits += red_hits[wv]; }
            G.out[(size_t)trial * A.nq + q] = metric_value(G.code, G.k, nh, first == 0x7fffffff ? -1 : first, hits, n_rel);
        }
        __syncthreads();
    }
}

int next_pow2(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" size_t mq_fuse_workspace_bytes(int n_runs, int nq, int K) {
    if (n_runs <= 0 || nq <= 0 || K <= 0) return 0;
    return ((size_t)n_runs * nq * STAT + 2 * (size_t)n_runs) * sizeof(double);
}

extern "C" int mq_fuse_wsum_f64(const int64_t* ids_dev, const double* scores_dev, int n_runs, int nq, int K,
                                const double* weights_host, int norm, int defmin, int64_t* out_ids_dev,
                                double* out_scores_dev, int32_t* out_count_dev, void* ws_dev, size_t ws_bytes,
                                void* stream) {
    if (!ids_dev || !scores_dev || !weights_host || !out_ids_dev || !out_scores_dev || !out_count_dev || !ws_dev)
        return MQ_EINVAL;
    if (n_runs <= 0 || n_runs > MQ_FUSE_MAX_RUNS || nq <= 0 || K <= 0) return MQ_EINVAL;
    if ((int64_t)n_runs * K > FCAP) return MQ_EUNSUPPORTED;
    if (norm != MQ_FUSE_NORM_NONE && norm != MQ_FUSE_NORM_GZMUV && norm != MQ_FUSE_NORM_ZMUV) return MQ_EINVAL;
    if (ws_bytes < mq_fuse_workspace_bytes(n_runs, nq, K)) return MQ_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    FuseArgs A;
    A.ids = ids_dev; A.scores = scores_dev;
    A.stats = (double*)ws_dev;
    A.moments = A.stats + (size_t)n_runs * nq * STAT;
    A.out_ids = out_ids_dev; A.out_scores = out_scores_dev; A.out_count = out_count_dev;
    A.R = n_runs; A.nq = nq; A.K = K; A.nsort = next_pow2(n_runs * K); A.norm = norm; A.defmin = defmin ? 1 : 0;
    for (int r = 0; r < MQ_FUSE_MAX_RUNS; ++r) A.w[r] = r < n_runs ? weights_host[r] : 0.0;
    const size_t lds = (size_t)A.nsort * 2 * sizeof(uint64_t);
    MQ_DYNAMIC_LDS_WITH(FUSE_HIP, mq_detail::LDS_PER_CU, fuse_stats_kernel);
    MQ_DYNAMIC_LDS_WITH(FUSE_HIP, mq_detail::LDS_PER_CU, fuse_combine_kernel);
    hipLaunchKernelGGL(fuse_stats_kernel, dim3(nq), dim3(FT), lds, st, A);
    FUSE_HIP(hipGetLastError());
    if (norm == MQ_FUSE_NORM_GZMUV) {
        hipLaunchKernelGGL(fuse_moments_kernel, dim3(n_runs), dim3(1024), 0, st, A);
        FUSE_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fuse_combine_kernel, dim3(nq), dim3(FT), lds, st, A);
    FUSE_HIP(hipGetLastError());
    return MQ_OK;
}

extern "C" int mq_run_metrics_f64(const int64_t* ids_dev, int nq, int K, const int64_t* rel_ptr_dev,
                                  const int64_t* rel_ids_dev, int n_metrics, const int* codes_host, const int* ks_host,
                                  double* per_query_dev, double* mean_dev, void* stream) {
    if (!ids_dev || !rel_ptr_dev || !codes_host || !ks_host || !per_query_dev || !mean_dev) return MQ_EINVAL;
    if (nq <= 0 || K <= 0 || n_metrics <= 0 || n_metrics > MQ_RANK_MAX_METRICS) return MQ_EINVAL;
    MetricArgs A;
    A.ids = ids_dev; A.rel_ptr = rel_ptr_dev; A.rel_ids = rel_ids_dev; A.out = per_query_dev;
    A.nq = nq; A.K = K; A.n_metrics = n_metrics;
    for (int m = 0; m < n_metrics; ++m) {
        if (codes_host[m] < MQ_RANK_METRIC_MRR || codes_host[m] > MQ_RANK_METRIC_RECALL || ks_host[m] < 0) return MQ_EINVAL;
        A.code[m] = codes_host[m]; A.k[m] = ks_host[m];
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(run_metrics_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, A);
    FUSE_HIP(hipGetLastError());
    hipLaunchKernelGGL(np_mean_rows_kernel, dim3((n_metrics + 63) / 64), dim3(64), 0, st, per_query_dev, mean_dev, n_metrics, nq);
    FUSE_HIP(hipGetLastError());
    return MQ_OK;
}

extern "C" int mq_fuse_fit_wsum_f64(const int64_t* ids_dev, const double* scores_dev, int n_runs, int nq, int K,
                                    const double* trials_dev, int n_trials, int norm, int defmin,
                                    const int64_t* rel_ptr_dev, const int64_t* rel_ids_dev, int metric, int metric_k,
                                    double* per_query_dev, double* mean_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    if (!ids_dev || !scores_dev || !trials_dev || !rel_ptr_dev || !per_query_dev || !mean_dev || !ws_dev) return MQ_EINVAL;
    if (n_runs <= 0 || n_runs > MQ_FUSE_MAX_RUNS || nq <= 0 || K <= 0 || n_trials <= 0 || metric_k < 0) return MQ_EINVAL;
    if (metric < MQ_RANK_METRIC_MRR || metric > MQ_RANK_METRIC_RECALL) return MQ_EINVAL;
    if ((int64_t)n_runs * K > FCAP) return MQ_EUNSUPPORTED;
    if (norm != MQ_FUSE_NORM_NONE && norm != MQ_FUSE_NORM_GZMUV && norm != MQ_FUSE_NORM_ZMUV) return MQ_EINVAL;
    if (ws_bytes < mq_fuse_workspace_bytes(n_runs, nq, K)) return MQ_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    FitArgs G;
    FuseArgs& A = G.F;
    A.ids = ids_dev; A.scores = scores_dev;
    A.stats = (double*)ws_dev;
    A.moments = A.stats + (size_t)n_runs * nq * STAT;
    A.out_ids = nullptr; A.out_scores = nullptr; A.out_count = nullptr;
    A.R = n_runs; A.nq = nq; A.K = K; A.nsort = next_pow2(n_runs * K); A.norm = norm; A.defmin = defmin ? 1 : 0;
    if (A.nsort < FT) A.nsort = FT;
    for (int r = 0; r < MQ_FUSE_MAX_RUNS; ++r) A.w[r] = 0.0;
    G.trials = trials_dev; G.rel_ptr = rel_ptr_dev; G.rel_ids = rel_ids_dev; G.out = per_query_dev;
    G.T = n_trials; G.code = metric; G.k = metric_k;
    const size_t lds = (size_t)A.nsort * 2 * sizeof(uint64_t);
    const size_t lds_fit = (size_t)A.nsort * (3 * sizeof(uint64_t) + 2 * sizeof(int));
    MQ_DYNAMIC_LDS_WITH(FUSE_HIP, mq_detail::LDS_PER_CU, fuse_stats_kernel);
    MQ_DYNAMIC_LDS_WITH(FUSE_HIP, mq_detail::LDS_PER_CU, fuse_fit_kernel);
    hipLaunchKernelGGL(fuse_stats_kernel, dim3(nq), dim3(FT), lds, st, A);
    FUSE_HIP(hipGetLastError());
    if (norm == MQ_FUSE_NORM_GZMUV) {
        hipLaunchKernelGGL(fuse_moments_kernel, dim3(n_runs), dim3(1024), 0, st, A);
        FUSE_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fuse_fit_kernel, dim3(nq), dim3(FT), lds_fit, st, G);
    FUSE_HIP(hipGetLastError());
    hipLaunchKernelGGL(np_mean_rows_kernel, dim3((n_trials + 63) / 64), dim3(64), 0, st, per_query_dev, mean_dev, n_trials, nq);
    FUSE_HIP(hipGetLastError());
    return MQ_OK;
}
