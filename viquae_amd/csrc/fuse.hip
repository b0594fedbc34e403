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
