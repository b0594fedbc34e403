// knn.hip -- exact brute-force top-k over a KB matrix resident in HBM, for gfx950 (MI355X).
//
// Replaces the arithmetic of faiss IndexFlat behind datasets' FaissIndex (add_vectors /
// search_batch, datasets/search.py:255-313,369-385), reached from the reference at
// meerqat/ir/search.py:146,245.  C ABI: include/meerqat_hip.h.  Design notes: DESIGN.md.
//
// Kernels in this file
//   pack_rows_kernel     row-major fp32 rows -> 64-row "panel" layout (k-major inside a panel),
//                        optional FAISS "L2norm," transform, ||x||^2 per stored row
//   unpack_rows_kernel   inverse (index save)
//   l2norm_rows_kernel   in-place row normalisation (queries; meerqat/ir/search.py:43-46)
//   knn_scan_kernel      the hot kernel: S = X . Q^T on v_mfma_f32_32x32x2_f32 with the top-k
//                        selection fused behind the accumulators (no score matrix in HBM)
//   slab_merge_kernel    per-query merge of the per-slab sorted lists -> D, I
//   shard_merge_kernel   per-query merge of per-GPU results after the RCCL all-gather
//
// Numerics: every inner product is the k-ordered fp32 chain acc = fma(x[k], q[k], acc); this is
// what v_mfma_f32_32x32x2_f32 computes when K is walked in order, and what oracle/knn_oracle.c
// restates on the CPU, so results are compared bit-exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "../../include/meerqat_hip.h"
#include "launch_attr.h"

typedef unsigned long long u64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Code placement of the MFMA loop moves kernel time by +-4.5 % between 4-byte phases (measured,
// profiles/r01_notes.md); MQ_PAD_NOPS shifts it.  Re-tune with tools/tune_placement.sh after edits.
#ifndef MQ_PAD_NOPS
#define MQ_PAD_NOPS 3
#endif

namespace {

// A/B switches of the search paths (mq_knn_set_option, include/meerqat_hip.h): process-wide atomics whose FIRST values come
// from the environment (MQ_KNN_SMALL, MQ_KNN_SMALL_MIN_TILES, MQ_KNN_PARTITIONS), read once.  Tests and bench.py flip them
// through the setter: no getenv() on the search path (it races with a putenv from another thread), one rule for all three.
std::atomic<int> g_knn_opt[MQ_KNN_OPT_COUNT];
std::once_flag g_knn_opt_once;
void knn_opt_init() {
    std::call_once(g_knn_opt_once, [] {
        const char* e;
        g_knn_opt[MQ_KNN_OPT_SMALL_SCAN].store((e = getenv("MQ_KNN_SMALL")) ? (atoi(e) != 0) : 1);
        g_knn_opt[MQ_KNN_OPT_SMALL_MIN_TILES].store((e = getenv("MQ_KNN_SMALL_MIN_TILES")) ? atoi(e) : 0);  // 0: the built-in floor
        g_knn_opt[MQ_KNN_OPT_PARTITIONS].store((e = getenv("MQ_KNN_PARTITIONS")) ? (atoi(e) != 0) : 1);
        g_knn_opt[MQ_KNN_OPT_SMALL_WAVES].store((e = getenv("MQ_KNN_SMALL_WAVES")) ? (atoi(e) == 4 ? 4 : 8) : 8);
    });
}
inline int knn_opt(int key) {
    knn_opt_init();
    return g_knn_opt[key].load(std::memory_order_relaxed);
}

constexpr int PANEL = 64;   // KB rows per panel
constexpr int BK = 16;      // k-depth of one LDS stage
constexpr int TQ = 256;     // queries per workgroup tile
constexpr int TN = 256;     // KB rows per chunk
constexpr int NWAVES = 16;  // 4 (row panels) x 4 (query panels), one 64x64 sub-tile per wave
constexpr int NSTAGE = 3;   // LDS ring depth of the operand tiles
constexpr int POOL = 512;   // per-(query, slab) candidate pool in HBM: sorted top-k prefix + unsorted tail
constexpr int PR = POOL / 64;  // pool keys per lane when one wave sorts a pool

// LDS carve (bytes)
constexpr int LDS_XS = 0;                                   // [NSTAGE][4][16][64] f32
constexpr int LDS_QS = LDS_XS + NSTAGE * 4 * BK * 64 * 4;   // [NSTAGE][4][16][64] f32
constexpr int LDS_CNT = LDS_QS + NSTAGE * 4 * BK * 64 * 4;  // gcnt[256] int, tau[256] f32
constexpr int LDS_TOTAL = LDS_CNT + 2 * TQ * 4;
constexpr int KF = MQ_KNN_FUSED_K;  // neighbours one fused scan keeps; a larger k runs ceil(k / KF) rounds (see knn_search_impl)
// neighbours the screened search serves itself (final_select sorts up to 256 exact keys).  Measured at 1.5M x 768, 4096 queries:
// k = 129 / 200 / 224 / 240: 8.8 / 9.8 / 10.2 / 11.1 ms; from k = 248 the stripe bound (256 slots per query) no longer yields a
// threshold, the slab pools overflow and every tile falls back to the exact rounds (160-176 ms against 146 for the rounds alone).
constexpr int SCREEN_MAX_K = 224;
constexpr int FINAL_SORT_KEYS = 256;
static_assert(KF + TN <= POOL, "a compacted pool must take every row of one more chunk");
static_assert(KF == 128, "sort128_desc / the merge kernels are written for 128-entry lists");

static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

__device__ __forceinline__ u64 make_key(float g, unsigned row) {
    unsigned b = __float_as_uint(g + 0.0f);  // -0 -> +0
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((u64)b << 32) | (u64)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_score(u64 key) {
    unsigned b = (unsigned)(key >> 32);
    b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(b);
}
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long)(lds_char*)p; }
// one 1-KiB LDS-DMA piece: LDS destination = lds_dst (wave-uniform) + lane*16, source per lane
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// same piece with a wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the per-step
// address update is scalar arithmetic only
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// the same piece with the non-temporal cache policy (streamed-once operands; measured per kernel)
__device__ __forceinline__ void dma16s_nt(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// a 4-byte-per-lane LDS-DMA piece (256 bytes per full wave; inactive lanes move nothing): LDS destination = lds_dst + lane * 4
__device__ __forceinline__ void dma4s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned key_row(u64 key) { return 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull); }

// ------------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------------

// The "L2norm," prefix in its two arithmetics (include/meerqat_hip.h: MQ_L2NORM_NUMPY / MQ_L2NORM_FAISS; oracle/knn_oracle.c):
// the per-row factor from the squared norm nr (the k-ordered fmaf chain), and its application to one element.
//   numpy form (meerqat/ir/search.py:43-46, and :238-244 for the KB rows when `device` is given):  x / sqrtf(nr)
//   FAISS form (NormalizationTransform -> fvec_renorm_L2, what "L2norm,Flat" runs with `device: null`):
//       if (nr > 0) x *= (float)(1.0 / (double)sqrtf(nr));   rows whose nr is not > 0 stay as they are
__device__ __forceinline__ float l2norm_scale(float nr, int form) {
    if (form == MQ_L2NORM_FAISS) return nr > 0.f ? (float)(1.0 / (double)sqrtf(nr)) : 1.0f;
    return sqrtf(nr);
}
__device__ __forceinline__ float l2norm_apply(float x, float scale, int form) {
    return form == MQ_L2NORM_FAISS ? x * scale : x / scale;
}

// 256 threads; loads rows [row0, row0+64) x k [kc, kc+64) of a row-major [n,d] matrix into
// tile[64][65] (zero outside the matrix)
__device__ __forceinline__ void load_tile64(const float* __restrict__ src, int64_t n, int d, int64_t row0, int kc,
                                            float (*tile)[65]) {
    const int t = threadIdx.x;
    const int kq = (t & 15) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (t >> 4) + 16 * j;
        const int64_t row = row0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < n) {
            const float* p = src + row * (int64_t)d + kc + kq;
            if (kc + kq + 3 < d && ((((uintptr_t)p) & 15) == 0)) {
                const float4 f = *reinterpret_cast<const float4*>(p);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kc + kq + e < d) v[e] = p[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][kq + e] = v[e];
    }
}

// one workgroup (256 threads) per 64-row panel
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int64_t n, int d, int dpad,
                                                        int64_t row_offset, int l2norm, float* __restrict__ packed,
                                                        float* __restrict__ sqnorm, const int* __restrict__ only_tiles = nullptr,
                                                        int64_t n_pad = 0) {
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * PANEL;  // within this call's rows
    // screened search: the panel copy of the queries only serves the exact-scan fallback of flagged query tiles
    if (only_tiles && !only_tiles[(row0 + row_offset) / TQ]) return;
    const int64_t panel = (row_offset / PANEL) + blockIdx.x;
    if (l2norm) {
        float acc = 0.f;
        for (int kc = 0; kc < d; kc += 64) {
            load_tile64(src, n, d, row0, kc, tile);
            __syncthreads();
            if (t < 64) {
#pragma unroll 8
                for (int k = 0; k < 64; ++k) acc = fmaf(tile[t][k], tile[t][k], acc);
            }
            __syncthreads();
        }
        if (t < 64) nrm[t] = l2norm_scale(acc, l2norm);
        __syncthreads();
    }
    float acc2 = 0.f;
    for (int kc = 0; kc < dpad; kc += 64) {
        load_tile64(src, n, d, row0, kc, tile);
        __syncthreads();
        if (l2norm) {
            // numpy form: x / ||x||, the oracle's two roundings (sqrtf, then one division per element); FAISS form: x * inv
            for (int e = t; e < 64 * 64; e += 256) {
                const int r = e >> 6, k = e & 63;
                if (row0 + r < n && kc + k < d) tile[r][k] = l2norm_apply(tile[r][k], nrm[r], l2norm);
            }
            __syncthreads();
        }
        if (t < 64) {
#pragma unroll 8
            for (int k = 0; k < 64; ++k) acc2 = fmaf(tile[t][k], tile[t][k], acc2);
        }
        // write [k][r]: thread -> 4 consecutive r at one k
        const int r4 = (t & 15) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = (t >> 4) + 16 * j;
            if (kc + k < dpad) {
                float4 f;
                f.x = tile[r4 + 0][k]; f.y = tile[r4 + 1][k]; f.z = tile[r4 + 2][k]; f.w = tile[r4 + 3][k];
                *reinterpret_cast<float4*>(packed + ((panel * dpad + kc + k) * PANEL + r4)) = f;
            }
        }
        __syncthreads();
    }
    // n_pad > n: the panels and norms of padding rows [n, n_pad) are written too (zeros), for a destination nobody cleared
    if (t < 64 && row0 + t < (n_pad > n ? n_pad : n)) sqnorm[row_offset + row0 + t] = acc2;
}

__global__ void unpack_rows_kernel(const float* __restrict__ packed, int d, int dpad, int64_t row_offset, int64_t n,
                                   float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * (int64_t)d) return;
    const int64_t r = e / d;
    const int k = (int)(e - r * d);
    const int64_t row = row_offset + r;
    dst[e] = packed[((row / PANEL) * dpad + k) * PANEL + (row % PANEL)];
}

// pack_rows_kernel's arithmetic (the "L2norm," transform, the ||x||^2 chain over the padded dimension) with a ROW-MAJOR
// destination dst[row_offset + r][k] -- the stored rows of a screened index that keeps no panel copy; any row_offset.
__global__ __launch_bounds__(256) void store_rows_kernel(const float* __restrict__ src, int64_t n, int d, int dpad,
                                                         int64_t row_offset, int l2norm, float* __restrict__ dst,
                                                         float* __restrict__ sqnorm) {
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    if (l2norm) {
        float acc = 0.f;
        for (int kc = 0; kc < d; kc += 64) {
            load_tile64(src, n, d, row0, kc, tile);
            __syncthreads();
            if (t < 64) {
#pragma unroll 8
                for (int k = 0; k < 64; ++k) acc = fmaf(tile[t][k], tile[t][k], acc);
            }
            __syncthreads();
        }
        if (t < 64) nrm[t] = l2norm_scale(acc, l2norm);
        __syncthreads();
    }
    float acc2 = 0.f;
    for (int kc = 0; kc < dpad; kc += 64) {
        load_tile64(src, n, d, row0, kc, tile);
        __syncthreads();
        if (l2norm) {
            for (int e = t; e < 64 * 64; e += 256) {
                const int r = e >> 6, k = e & 63;
                if (row0 + r < n && kc + k < d) tile[r][k] = l2norm_apply(tile[r][k], nrm[r], l2norm);
            }
            __syncthreads();
        }
        if (t < 64) {
#pragma unroll 8
            for (int k = 0; k < 64; ++k) acc2 = fmaf(tile[t][k], tile[t][k], acc2);
        }
        for (int e = t; e < 64 * 64; e += 256) {
            const int r = e >> 6, k = e & 63;
            if (row0 + r < n && kc + k < d) dst[(row_offset + row0 + r) * (int64_t)d + kc + k] = tile[r][k];
        }
        __syncthreads();
    }
    if (t < 64 && row0 + t < n) sqnorm[row_offset + row0 + t] = acc2;
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ rows, int64_t n, int d, int form = MQ_L2NORM_NUMPY) {
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    float acc = 0.f;
    for (int kc = 0; kc < d; kc += 64) {
        load_tile64(rows, n, d, row0, kc, tile);
        __syncthreads();
        if (t < 64) {
#pragma unroll 8
            for (int k = 0; k < 64; ++k) acc = fmaf(tile[t][k], tile[t][k], acc);
        }
        __syncthreads();
    }
    if (t < 64) nrm[t] = l2norm_scale(acc, form);
    __syncthreads();
    for (int64_t e = t; e < 64 * (int64_t)d; e += 256) {
        const int r = (int)(e / d);
        const int k = (int)(e - (int64_t)r * d);
        if (row0 + r < n) {
            float* p = rows + (row0 + r) * (int64_t)d + k;
            *p = l2norm_apply(*p, nrm[r], form);
        }
    }
}

// The same normalisation for FEW rows (a search's queries: 256 per call in the reference's batches): l2norm_rows_kernel gives 64
// rows to a workgroup, so 256 queries of 2048 columns were four workgroups walking 32 dependent load -> barrier -> chain steps
// each -- 229 us of a 2.25-ms search over the 2048-d "L2norm,Flat" index (rocprofv3, round 5).  Here a workgroup takes 8 rows and
// 1024 columns per step, 32 unconditional loads in flight per thread (256 columns per step, each load under its bounds test:
// 50 us at d = 2048 -- one round trip per load); the sum of squares stays the k-ordered fp32 fma chain of one thread per
// row, the bits are the same.
__global__ __launch_bounds__(256) void l2norm_rows_small_kernel(float* __restrict__ rows, int64_t n, int d, int form) {
    constexpr int CH = 1024;
    __shared__ float tile[8][CH + 1];
    __shared__ float nrm[8];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 8;
    float acc = 0.f;
    for (int kc = 0; kc < d; kc += CH) {
        // every load at an address inside the matrix (clamped), all 32 in flight, then the stores: a load under a condition is
        // a branch and a wait of its own (32 round trips per step)
        float v[8][CH / 256];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t rr = row0 + r < n ? row0 + r : n - 1;
#pragma unroll
            for (int j = 0; j < CH / 256; ++j) {
                const int c = kc + t + 256 * j;
                v[r][j] = rows[rr * (int64_t)d + (c < d ? c : d - 1)];
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < CH / 256; ++j) tile[r][t + 256 * j] = v[r][j];  // (columns >= d, rows >= n: never read)
        }
        __syncthreads();
        if (t < 8) {
            const int kn = d - kc < CH ? d - kc : CH;
            int k = 0;
            for (; k + 16 <= kn; k += 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fmaf(tile[t][k + u], tile[t][k + u], acc);
            }
            for (; k < kn; ++k) acc = fmaf(tile[t][k], tile[t][k], acc);
        }
        __syncthreads();
    }
    if (t < 8) nrm[t] = l2norm_scale(acc, form);
    __syncthreads();
    for (int r = 0; r < 8; ++r) {
        if (row0 + r >= n) break;
        float* p = rows + (row0 + r) * (int64_t)d;
        const float sc = nrm[r];
        for (int k = t; k < d; k += 256) p[k] = l2norm_apply(p[k], sc, form);
    }
}
constexpr int64_t L2NORM_SMALL_ROWS = 4096;  // up to here the 8-row kernel (>= 512 workgroups beyond)
inline void launch_l2norm_rows(float* rows, int64_t n, int d, int form, hipStream_t st) {
    if (n <= L2NORM_SMALL_ROWS)
        hipLaunchKernelGGL(l2norm_rows_small_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st, rows, n, d, form);
    else
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, rows, n, d, form);
}

// ------------------------------------------------------------------------------------------------
// the scan kernel
// ------------------------------------------------------------------------------------------------
struct ScanArgs {
    const float* Xp;   // KB, panel layout (knn_scan_kernel<*, true>: the row-major [N][d] copy)
    const float* Qp;   // queries, panel layout (nqt * 4 panels)
    const float* xn;   // ||x||^2 per KB row (L2 only)
    const float* qn;   // ||q||^2 per query   (L2 only)
    u64* lists;        // [nqt][S][256][POOL] keys: after the scan, entries [0,k) are the sorted top-k
    long long N;
    int dpad, nqt, S, k, qpx;
    long long nchunks;
    unsigned long long* dbg;  // MQ_TIMING builds only
    int d;                    // row stride of the row-major operand (XROW instantiations)
    const int* only;          // optional [nqt] flags: scan only the flagged query tiles (fallback of the screened path)
    unsigned flip;            // tie order: 0 = lower id wins an exact tie (key low word = ~row), 0xFFFFFFFF = higher id wins
    const u64* ceil;          // CEIL instantiations: per-query key ceiling -- only keys strictly below it are candidates
};

// Tie order.  A key is (orderable score << 32) | ~(row ^ flip): with flip = 0 the lower row has the larger key (id_asc, the
// default), with flip = 0xFFFFFFFF the higher row has (id_desc).  The score compare against tau stays STRICT; under id_desc a
// later row that ties the k-th best must still enter (it beats it), so compact_pool publishes the next float BELOW the k-th
// best score as tau.
__device__ __forceinline__ float tau_of(u64 T, unsigned flip) {
    const float t = key_score(T);
    return flip ? nextafterf(t, -INFINITY) : t;
}

// ---- per-query candidate pool (HBM, owned by one workgroup) ------------------------------------
// A candidate that beats the query's threshold tau is appended, unsorted, to the query's pool: slot
// from an LDS atomic counter, one 8-byte global store.  A pool is COMPACTED only when another whole
// chunk might not fit (count > POOL - TN): one wave bitonic-sorts its POOL keys in registers, keeps
// the k best (written back sorted) and refreshes tau.  Between compactions tau is stale but valid
// (the k-th best of a subset).  Each refill takes ~(POOL-TN-k)/k times more rows than the previous
// one, so a pool is compacted a handful of times per slab, and the pool can never overflow: no
// staging buffers, no overflow path, no re-scan of the accumulators.

// wave-wide bitonic sort, descending, of 2 x 64 keys (element index = r*64 + lane): final ordering of a
// compacted pool (k <= 128), once per query per slab
__device__ __forceinline__ void sort128_desc(u64 (&v)[2], int lane) {
#pragma unroll
    for (int kk = 2; kk <= 128; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            if (j == 64) {
                const u64 a = v[0], b = v[1];  // kk == 128: descending
                v[0] = a > b ? a : b;
                v[1] = a > b ? b : a;
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bool up = ((r * 64 + lane) & kk) != 0;
                    const bool lower = (lane & j) == 0;
                    const unsigned lo = __shfl_xor((unsigned)v[r], j);
                    const unsigned hi = __shfl_xor((unsigned)(v[r] >> 32), j);
                    const u64 o = ((u64)hi << 32) | lo;
                    const bool take_max = (lower != up);
                    v[r] = take_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
                }
            }
        }
    }
}

// the same network for R x 64 keys (R a power of two): strides below 64 exchange lanes, strides of 64 and more exchange
// registers.  R = 2 is sort128_desc (kept as it is: it sits in the scans' compaction path).
template <int R>
__device__ __forceinline__ void sort_desc_n(u64 (&v)[R], int lane) {
#pragma unroll
    for (int kk = 2; kk <= 64 * R; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {
                const int jr = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & jr) continue;
                    const bool up = ((r * 64) & kk) != 0;  // (lane < 64 never reaches bit kk >= 128)
                    const u64 a = v[r], b = v[r | jr];
                    const u64 mx = a > b ? a : b, mn = a > b ? b : a;
                    v[r] = up ? mn : mx;
                    v[r | jr] = up ? mx : mn;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool up = ((r * 64 + lane) & kk) != 0;
                    const bool lower = (lane & j) == 0;
                    const unsigned lo = __shfl_xor((unsigned)v[r], j);
                    const unsigned hi = __shfl_xor((unsigned)(v[r] >> 32), j);
                    const u64 o = ((u64)hi << 32) | lo;
                    const bool take_max = (lower != up);
                    v[r] = take_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
                }
            }
        }
    }
}

// k-th largest of the wave's PR x 64 unique keys (k <= number of non-zero keys): bisection on the key
// bits, high (score) word first with 32-bit compares; the low (row) word only matters when several
// keys share the k-th score.
__device__ __forceinline__ u64 wave_kth_largest(const u64 (&v)[PR], int k) {
    unsigned hi[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) hi[r] = (unsigned)(v[r] >> 32);
    unsigned Th = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = Th | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < PR; ++r) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hi[r] >= cand));
        if (c >= k) Th = cand;
    }
    int cgt = 0, ceq = 0;
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        cgt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hi[r] > Th));
        ceq += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hi[r] == Th));
    }
    if (cgt + ceq == k) return (u64)Th << 32;  // every key of the k-th score is kept
    const int need = k - cgt;                   // of the ceq keys with the k-th score, keep the `need` largest low words
    unsigned Tl = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = Tl | (1u << bit);
        int c = 0;
#pragma unroll
        for (int r = 0; r < PR; ++r)
            c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hi[r] == Th && (unsigned)v[r] >= cand));
        if (c >= need) Tl = cand;
    }
    return ((u64)Th << 32) | Tl;
}

// Keep the k best of the pool's first g entries at [0,k) and refresh tau; returns the new count.
// No sort: the k-th largest key T is found by bisection on the key bits (keys are unique, so exactly
// k keys are >= T), survivors are packed with ballot prefix sums.  `sorted`: additionally order the
// survivors best-first and zero-fill up to k (end of the slab).
// The entries were stored by other waves of this workgroup: read them past the CU's L1 (agent-scope
// relaxed loads are served by L2), after the caller's __syncthreads().
__device__ __forceinline__ int compact_pool(u64* __restrict__ P, int g, int k, float* tau_q, int lane, bool sorted, unsigned flip) {
    u64 v[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r)
        v[r] = (r * 64 + lane < g) ? __hip_atomic_load(P + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    int ng = g;
    if (g > k) {
        const u64 T = wave_kth_largest(v, k);
        // pack the k survivors to the front (order irrelevant)
        int base = 0;
        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const bool keep = v[r] >= T;
            const u64 m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) P[base + __builtin_popcountll(m & lt)] = v[r];
            base += __builtin_popcountll(m);
        }
        if (lane == 0) *tau_q = tau_of(T, flip);
        ng = k;
        if (!sorted) return ng;
        // the packed survivors were written by this wave; re-read them (own stores, same wave: program order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < PR; ++r)
            v[r] = (r < 2 && r * 64 + lane < k) ? __hip_atomic_load(P + r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    if (sorted) {
        u64 s2[2] = {v[0], v[1]};  // g <= k <= 128 or the packed survivors: everything lives in idx < 128
        sort128_desc(s2, lane);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = r * 64 + lane;
            if (idx < k) P[idx] = s2[r];  // zeros (empty) sort last: slots >= ng are zero-filled
        }
    }
    return ng;
}

#ifdef MQ_TIMING
// cycle accounting per wave (tools/ only): [0] K loop, [1] scan+append, [2] barrier, [3] compaction
#define MQ_T(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - tlast_; tlast_ = now_; }
#define MQ_T_DUMP if (lane == 0) { for (int i_ = 0; i_ < 4; ++i_) a.dbg[((size_t)blockIdx.x * NWAVES + w) * 4 + i_] = tacc_[i_]; }
#else
#define MQ_T(i)
#define MQ_T_DUMP
#endif

// XROW: the KB operand is read from a ROW-MAJOR fp32 matrix (a.Xp = [N][a.d]) instead of the panel layout -- the exact-scan
// fallback of a screened index that keeps no panel copy (mq_knn_search_screened_f32 with packed_dev = NULL).  Each thread
// loads four consecutive k of one row to registers one K-step ahead and lays them out in the LDS stage exactly as the
// LDS-DMA of the panel layout would, so the MFMA sequence -- hence every score bit -- is the same; rows >= N and k >= d read
// as zero like the panel padding.  Slower than the DMA path (64-byte row pieces, a 4-way bank conflict on the transposing
// ds_write): it only runs for query tiles whose screening buffers overflowed.
// CEIL: one round of a search for more than KF neighbours -- a candidate must also lie strictly below the query's key
// ceiling (the last key the previous round reported), tested on the rare append path only.
template <int METRIC, bool XROW = false, bool CEIL = false>
__global__ __launch_bounds__(1024) void knn_scan_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xs = reinterpret_cast<float*>(smem + LDS_XS);
    float* Qs = reinterpret_cast<float*>(smem + LDS_QS);
    int* gcnt = reinterpret_cast<int*>(smem + LDS_CNT);
    float* tau = reinterpret_cast<float*>(gcnt + TQ);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // (query tile, KB slab) of this workgroup.  Workgroup b runs on XCD b % 8 (observed dispatch
    // rule, used for speed only): all query tiles of one slab share an XCD, so the slab is read
    // from HBM once and served to the other query tiles from that XCD's L2.
    int qt, slab;
    {
        const int b = blockIdx.x;
        if (a.qpx > 0) {
            // XCD groups: `qpx` query tiles per XCD; the XCDs of one group split the slabs
            const int ngroups = a.nqt / a.qpx, xpg = 8 / ngroups, spx = a.S / xpg;
            const int xcd = b & 7, j = b >> 3;
            qt = (xcd / xpg) * a.qpx + (j % a.qpx);
            slab = (xcd % xpg) * spx + j / a.qpx;
        } else {
            slab = b / a.nqt;
            qt = b % a.nqt;
        }
    }
    qt = __builtin_amdgcn_readfirstlane(qt);
    slab = __builtin_amdgcn_readfirstlane(slab);
    if (a.only && a.only[qt] == 0) return;
    const int c0 = __builtin_amdgcn_readfirstlane((int)((a.nchunks * slab) / a.S));
    const int c1 = __builtin_amdgcn_readfirstlane((int)((a.nchunks * (slab + 1)) / a.S));
    const int k = a.k;
    u64* mylists = a.lists + ((size_t)qt * a.S + slab) * (size_t)TQ * POOL;

    if (tid < TQ) {
        gcnt[tid] = 0;
        tau[tid] = -FLT_MAX;  // FAISS heap neutral value: a score must beat it strictly to enter (NaN, -inf, -FLT_MAX never do)
    }

    const int nkb = a.dpad / BK;
    const int pp = w >> 2, quarter = w & 3;
    // DMA addressing: per-lane byte offset (constant for the whole kernel, the same for the KB and the
    // query operand) + wave-uniform base that advances by scalar arithmetic; LDS destination is
    // wave-uniform base + lane*16 (hardware rule)
    const unsigned dma_voff = (unsigned)((((size_t)pp * a.dpad + quarter * 4) * PANEL + lane * 4) * 4);
    const int lds_piece = pp * (BK * 64) + quarter * 256;
    const char* const xbase0 = reinterpret_cast<const char*>(a.Xp);
    const char* const qbase0 = reinterpret_cast<const char*>(a.Qp) + (size_t)qt * 4 * a.dpad * (PANEL * 4);
    const unsigned chunk_bytes = (unsigned)(4 * a.dpad * (PANEL * 4));  // 4 panels (< 4 GiB: dpad < 2^20)

    // LDS-DMA (global_load_lds_dwordx4) issued from inline asm so that hipcc does not count it:
    // with the builtin it drains vmcnt(0) in front of the next ds_read and the prefetch of step t+1
    // would serialise with the MFMAs of step t.  Completion is awaited by hand (vmcnt(0) right
    // before the barrier that opens the step which reads the data).
    const unsigned lds_x = __builtin_amdgcn_readfirstlane(lds_addr(Xs + lds_piece));
    const unsigned lds_q = __builtin_amdgcn_readfirstlane(lds_addr(Qs + lds_piece));
    auto issue = [&](int c, int kb, int stage) __attribute__((always_inline)) {
        const char* xb = xbase0 + (size_t)c * chunk_bytes + (size_t)kb * (BK * PANEL * 4);
        const char* qb = qbase0 + (size_t)kb * (BK * PANEL * 4);
        if (!XROW) dma16s(xb, dma_voff, lds_x + stage * (4096 * 4));
        dma16s(qb, dma_voff, lds_q + stage * (4096 * 4));
    };
    // XROW operand path: thread -> (row = tid / 4 of the chunk, k = 4 (tid % 4) .. + 3 of the K-step)
    float xr0 = 0.f, xr1 = 0.f, xr2 = 0.f, xr3 = 0.f;
    int xci = c0, xkbi = 0;
    auto xrow_load = [&]() __attribute__((always_inline)) {
        xr0 = xr1 = xr2 = xr3 = 0.f;
        if (xci < c1) {
            const long long row = (long long)xci * TN + (tid >> 2);
            const int kk = xkbi * BK + (tid & 3) * 4;
            if (row < a.N) {
                const float* p = a.Xp + (size_t)row * a.d + kk;
                if (kk + 3 < a.d && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    const float4 f = *reinterpret_cast<const float4*>(p);
                    xr0 = f.x; xr1 = f.y; xr2 = f.z; xr3 = f.w;
                } else {
                    if (kk < a.d) xr0 = p[0];
                    if (kk + 1 < a.d) xr1 = p[1];
                    if (kk + 2 < a.d) xr2 = p[2];
                    if (kk + 3 < a.d) xr3 = p[3];
                }
            }
            if (++xkbi == nkb) { xkbi = 0; ++xci; }
        }
    };
    auto xrow_store = [&](int stage) __attribute__((always_inline)) {
        float* dst = Xs + stage * 4096 + (tid >> 8) * (BK * 64) + (tid & 3) * (4 * 64) + ((tid >> 2) & 63);
        dst[0] = xr0; dst[64] = xr1; dst[128] = xr2; dst[192] = xr3;
    };

    const int i2 = (lane & 31) * 2;
    const int kh = lane >> 5;
    // this lane's two queries (tile-local) and their norms
    const int q0 = wc * 64 + i2, q1 = q0 + 1;
    float qn0 = 0.f, qn1 = 0.f;
    if (METRIC == MQ_METRIC_L2) {
        qn0 = a.qn[qt * TQ + q0];
        qn1 = a.qn[qt * TQ + q1];
    }

#if MQ_PAD_NOPS > 0
#pragma unroll
    for (int i_ = 0; i_ < MQ_PAD_NOPS; ++i_) asm volatile("s_nop 0");
#endif
    // Three-stage LDS ring, ONE barrier per K-step placed in the MIDDLE of the step's MFMA block:
    //   step t:  MFMA kk=0..3 | vmcnt(0) + barrier + issue DMA(t+2) | MFMA kk=4..7 (+ first reads of step t+1)
    // The barrier proves (a) every wave's DMA pieces of step t+1 have landed, (b) every wave is done
    // with step t-1, whose stage DMA(t+2) overwrites.  Waves arrive with MFMAs queued on both sides
    // of it and the operands of the next MFMAs already in registers, so the matrix pipe does not
    // drain at the barrier (it did, ~19 % of the time, with the barrier at the step boundary).
    int ci = c0;
    int kbi = 0;
    auto issue_next = [&](int stg) __attribute__((always_inline)) {
        if (ci < c1) {
            issue(ci, kbi, stg);
            if (++kbi == nkb) { kbi = 0; ++ci; }
        }
    };
    issue_next(0);
    issue_next(1);
    if (XROW) {  // stages 0 and 1 by hand; the rows of step 2 wait in registers
        xrow_load();
        xrow_store(0);
        xrow_load();
        xrow_store(1);
        xrow_load();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    int xoff = wr * (BK * 64) + kh * 64 + i2;
    int qoff = wc * (BK * 64) + kh * 64 + i2;
#define MQ_LD2(p) (*reinterpret_cast<const float2*>(p))
#define MQ_M4(X, Q)                                                          \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.x, Q.x, acc00, 0, 0, 0);  \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.x, Q.y, acc01, 0, 0, 0);  \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.y, Q.x, acc10, 0, 0, 0);  \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.y, Q.y, acc11, 0, 0, 0);
    float2 fx0 = MQ_LD2(Xs + xoff), fq0 = MQ_LD2(Qs + qoff);
    float2 fx1 = MQ_LD2(Xs + xoff + 128), fq1 = MQ_LD2(Qs + qoff + 128);

#ifdef MQ_TIMING
    unsigned long long tacc_[4] = {0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    for (int c = c0; c < c1; ++c) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        // pin the fragment offsets in registers HERE: if they are reloaded from a spill slot inside the
        // K loop, hipcc's vmcnt(0) for that reload also drains the LDS-DMA prefetch half a step early
        asm volatile("" : "+v"(xoff), "+v"(qoff));
        for (int kb = 0; kb < nkb; ++kb) {
            const float* xs = Xs + cur * 4096 + xoff;
            const float* qs = Qs + cur * 4096 + qoff;
            const float2 fx2 = MQ_LD2(xs + 2 * 128), fq2 = MQ_LD2(qs + 2 * 128);
            MQ_M4(fx0, fq0)
            const float2 fx3 = MQ_LD2(xs + 3 * 128), fq3 = MQ_LD2(qs + 3 * 128);
            MQ_M4(fx1, fq1)
            const float2 fx4 = MQ_LD2(xs + 4 * 128), fq4 = MQ_LD2(qs + 4 * 128);
            MQ_M4(fx2, fq2)
            const float2 fx5 = MQ_LD2(xs + 5 * 128), fq5 = MQ_LD2(qs + 5 * 128);
            MQ_M4(fx3, fq3)
            // raw s_barrier, no fence: __syncthreads() would also wait lgkmcnt(0), i.e. park every wave
            // until its just-issued fragment reads return -- 16 waves' reads in lockstep are a ~500-cycle
            // LDS burst during which the matrix pipe starves.  The only cross-wave facts needed here are
            // the DMA landing (vmcnt) and program order; the empty asm statements keep hipcc from moving
            // LDS accesses across the barrier.
            if (XROW) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // + the previous step's ds_write of the X rows
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (XROW) {
                xrow_store(cur >= 1 ? cur - 1 : 2);  // the rows of step t + 2, loaded during step t - 1
                xrow_load();                         // step t + 3
            }
            issue_next(cur >= 1 ? cur - 1 : 2);  // stage (cur + 2) % 3
            const float2 fx6 = MQ_LD2(xs + 6 * 128), fq6 = MQ_LD2(qs + 6 * 128);
            MQ_M4(fx4, fq4)
            const float2 fx7 = MQ_LD2(xs + 7 * 128), fq7 = MQ_LD2(qs + 7 * 128);
            MQ_M4(fx5, fq5)
            cur = cur == 2 ? 0 : cur + 1;
            const float* xn_ = Xs + cur * 4096 + xoff;
            const float* qn_ = Qs + cur * 4096 + qoff;
            fx0 = MQ_LD2(xn_); fq0 = MQ_LD2(qn_);
            MQ_M4(fx6, fq6)
            fx1 = MQ_LD2(xn_ + 128); fq1 = MQ_LD2(qn_ + 128);
            MQ_M4(fx7, fq7)
        }
#undef MQ_M4
#undef MQ_LD2

        // ---------------- selection epilogue ----------------
        // C/D map of 32x32x2: column j = lane&31 (query 2j+b of the wave's panel),
        // row i' = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  (KB row 2i'+a of the wave's panel)
        const unsigned rowbase = (unsigned)c * TN + wr * 64 + 8 * kh;
        const unsigned nrows = (unsigned)a.N;
        const bool ragged = ((long long)c + 1) * TN > a.N;

        if (METRIC == MQ_METRIC_L2) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const unsigned r0 = rowbase + 2 * ((reg & 3) + 8 * (reg >> 2));
                const float2 xn2 = *reinterpret_cast<const float2*>(a.xn + r0);  // rows r0, r0+1 (padded alloc)
                float d;
                d = (qn0 + xn2.x) - 2.0f * acc00[reg]; acc00[reg] = -(d < 0.f ? 0.f : d);
                d = (qn1 + xn2.x) - 2.0f * acc01[reg]; acc01[reg] = -(d < 0.f ? 0.f : d);
                d = (qn0 + xn2.y) - 2.0f * acc10[reg]; acc10[reg] = -(d < 0.f ? 0.f : d);
                d = (qn1 + xn2.y) - 2.0f * acc11[reg]; acc11[reg] = -(d < 0.f ? 0.f : d);
            }
        }

        // append every score that beats the query's threshold (strict: an equal score from a later row
        // never displaces an earlier id, like FAISS's heap)
        MQ_T(0)
        {
            const float t0 = tau[q0], t1 = tau[q1];
            u64* P0 = mylists + (size_t)q0 * POOL;
            u64* P1 = P0 + POOL;
            const unsigned flip = a.flip;
            u64 ceil0 = ~0ull, ceil1 = ~0ull;
            if (CEIL) {
                ceil0 = a.ceil[qt * TQ + q0];
                ceil1 = a.ceil[qt * TQ + q1];
            }
            // `rag` (compile-time): the chunk reaches past the last row (the shard's last chunk only); the ordinary chunk gets
            // its own copy of the loop without the row-bound tests
            auto append = [&](auto rag) __attribute__((always_inline)) {
                constexpr bool RAG = decltype(rag)::value;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const unsigned r0 = rowbase + 2 * ((reg & 3) + 8 * (reg >> 2));
                    const bool v0 = !RAG || r0 < nrows;
                    const bool v1 = !RAG || r0 + 1 < nrows;
                    const float g00 = acc00[reg], g01 = acc01[reg], g10 = acc10[reg], g11 = acc11[reg];
                    const bool p00 = v0 && g00 > t0;
                    const bool p01 = v0 && g01 > t1;
                    const bool p10 = v1 && g10 > t0;
                    const bool p11 = v1 && g11 > t1;
                    if (__builtin_amdgcn_ballot_w64(p00 || p01 || p10 || p11) == 0ull) continue;
                    if (CEIL) {
                        const u64 k00 = make_key(g00, r0 ^ flip), k01 = make_key(g01, r0 ^ flip);
                        const u64 k10 = make_key(g10, (r0 + 1) ^ flip), k11 = make_key(g11, (r0 + 1) ^ flip);
                        if (p00 && k00 < ceil0) P0[atomicAdd(&gcnt[q0], 1)] = k00;
                        if (p01 && k01 < ceil1) P1[atomicAdd(&gcnt[q1], 1)] = k01;
                        if (p10 && k10 < ceil0) P0[atomicAdd(&gcnt[q0], 1)] = k10;
                        if (p11 && k11 < ceil1) P1[atomicAdd(&gcnt[q1], 1)] = k11;
                    } else {
                        if (p00) P0[atomicAdd(&gcnt[q0], 1)] = make_key(g00, r0 ^ flip);
                        if (p01) P1[atomicAdd(&gcnt[q1], 1)] = make_key(g01, r0 ^ flip);
                        if (p10) P0[atomicAdd(&gcnt[q0], 1)] = make_key(g10, (r0 + 1) ^ flip);
                        if (p11) P1[atomicAdd(&gcnt[q1], 1)] = make_key(g11, (r0 + 1) ^ flip);
                    }
                }
            };
            if (ragged) append(std::true_type{}); else append(std::false_type{});
        }
        MQ_T(1)
        __syncthreads();  // appended keys are in L2 (vmcnt(0) precedes the barrier), counters final
        MQ_T(2)
        const bool last = (c + 1 == c1);
        for (int j = 0; j < TQ / NWAVES; ++j) {
            const int q = w + NWAVES * j;
            const int g = __builtin_amdgcn_readfirstlane(gcnt[q]);
            if (g > POOL - TN || last) {
                u64* P = mylists + (size_t)q * POOL;
                const int ng = compact_pool(P, g, k, &tau[q], lane, last, a.flip);
                if (lane == 0) gcnt[q] = ng;
            }
        }
        // tau / gcnt updates are ordered before their next use by the barriers of the next chunk's K loop
        MQ_T(3)
    }
    MQ_T_DUMP
}

// ------------------------------------------------------------------------------------------------
// merges
// ------------------------------------------------------------------------------------------------
struct Ent {
    float g;
    long long id;  // < 0: empty
};
// total order of the results: higher goodness first, then the lower id (desc = false) or the higher id (desc = true)
__device__ __forceinline__ bool better(const Ent& x, const Ent& y, bool desc) {
    if (y.id < 0) return x.id >= 0;
    if (x.id < 0) return false;
    return x.g > y.g || (x.g == y.g && (desc ? x.id > y.id : x.id < y.id));
}

// count of entries in sorted (best first) list L[0..k) that are better than e
__device__ __forceinline__ int count_better(const Ent* L, int k, const Ent& e, bool desc) {
    int lo = 0, hi = k;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (better(L[mid], e, desc)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Sequentially rank-merges nlists sorted lists of k entries; entry fetch is a functor so the same
// body serves the slab lists (packed keys) and the shard lists (D, I arrays).  128 threads, k <= 128.
template <typename Fetch>
__device__ __forceinline__ void merge_lists(int nlists, int k, Fetch fetch, Ent* Ra, Ent* Rb, Ent* Ls, Ent& out, bool desc) {
    const int t = threadIdx.x;
    const Ent empty = {0.f, -1};
    Ra[t] = (t < k) ? fetch(0, t) : empty;
    __syncthreads();
    Ent* cur = Ra;
    Ent* nxt = Rb;
    for (int s = 1; s < nlists; ++s) {
        Ls[t] = (t < k) ? fetch(s, t) : empty;
        nxt[t] = empty;
        __syncthreads();
        const Ent r = cur[t], l = Ls[t];
        if (r.id >= 0) {
            const int p = t + count_better(Ls, k, r, desc);
            if (p < k) nxt[p] = r;
        }
        if (l.id >= 0) {
            const int p = t + count_better(cur, k, l, desc);
            if (p < k) nxt[p] = l;
        }
        __syncthreads();
        Ent* tmp = cur; cur = nxt; nxt = tmp;
    }
    out = cur[t];
}

// kout / col0: the [nq, kout] result arrays receive this launch's k entries at columns [col0, col0 + k) (kout = k, col0 = 0
// for an ordinary search; a search for more than KF neighbours writes one KF-wide column block per round).  next_ceil
// (optional): the key of the round's LAST entry -- the next round's candidates lie strictly below it -- or 0 ("nothing
// left") when the round found fewer than k rows.
template <int METRIC>
__global__ __launch_bounds__(128) void slab_merge_kernel(const u64* __restrict__ lists, int nq, int S, int k,
                                                         long long id_offset, float* __restrict__ D,
                                                         long long* __restrict__ I, const int* __restrict__ only,
                                                         unsigned flip, int kout, int col0, u64* __restrict__ next_ceil) {
    __shared__ Ent Ra[128], Rb[128], Ls[128];
    const int q = blockIdx.x;
    if (only && only[q / TQ] == 0) return;
    const int qt = q / TQ, ql = q % TQ;
    auto fetch = [&](int s, int t) {
        const u64 key = lists[(((size_t)qt * S + s) * TQ + ql) * (size_t)POOL + t];
        Ent e;
        e.g = key_score(key);
        e.id = key ? (long long)(key_row(key) ^ flip) : -1;
        return e;
    };
    Ent out;
    merge_lists(S, k, fetch, Ra, Rb, Ls, out, flip != 0u);
    const int t = threadIdx.x;
    if (t < k) {
        float d;
        if (out.id < 0) d = (METRIC == MQ_METRIC_L2) ? FLT_MAX : -FLT_MAX;  // unfilled slot: FAISS reports its heap neutral value
        else d = ((METRIC == MQ_METRIC_L2) ? -out.g : out.g) + 0.0f;
        D[(size_t)q * kout + col0 + t] = d;
        I[(size_t)q * kout + col0 + t] = out.id < 0 ? -1 : out.id + id_offset;
        if (next_ceil && t == k - 1) next_ceil[q] = out.id < 0 ? 0ull : make_key(out.g, (unsigned)out.id ^ flip);
    }
}

template <int METRIC>
__global__ __launch_bounds__(128) void shard_merge_kernel(const float* __restrict__ Ds, const long long* __restrict__ Is,
                                                          size_t d_stride, size_t i_stride, int nshards, int nq, int k,
                                                          float* __restrict__ D, long long* __restrict__ I, bool desc) {
    // d_stride / i_stride: elements between two shards' [nq,k] blocks (nq*k for two dense arrays; the
    // all-gathered per-rank records {scores | ids} give record_bytes/4 and record_bytes/8)
    __shared__ Ent Ra[128], Rb[128], Ls[128];
    const int q = blockIdx.x;
    auto fetch = [&](int s, int t) {
        const size_t o = (size_t)q * (size_t)k + t;
        Ent e;
        e.id = Is[(size_t)s * i_stride + o];
        const float d = Ds[(size_t)s * d_stride + o];
        e.g = (METRIC == MQ_METRIC_L2) ? -d : d;
        return e;
    };
    Ent out;
    merge_lists(nshards, k, fetch, Ra, Rb, Ls, out, desc);
    const int t = threadIdx.x;
    if (t < k) {
        float d;
        if (out.id < 0) d = (METRIC == MQ_METRIC_L2) ? FLT_MAX : -FLT_MAX;  // unfilled slot: FAISS reports its heap neutral value
        else d = ((METRIC == MQ_METRIC_L2) ? -out.g : out.g) + 0.0f;
        D[(size_t)q * k + t] = d;
        I[(size_t)q * k + t] = out.id < 0 ? -1 : out.id;
    }
}

// Shard merge for k > KF (any k): every entry of every list finds its rank in the merged order by counting, with binary
// searches in the other (sorted, best-first) lists, the entries that beat it; ids are unique across shards, so ranks are
// unique.  One 256-thread workgroup per query; the lists stay in HBM / L2 (W x k x 12 bytes per query).
template <int METRIC>
__global__ __launch_bounds__(256) void shard_merge_big_kernel(const float* __restrict__ Ds, const long long* __restrict__ Is,
                                                              size_t d_stride, size_t i_stride, int nshards, int nq, int k,
                                                              float* __restrict__ D, long long* __restrict__ I, bool desc) {
    const int q = blockIdx.x;
    const size_t qo = (size_t)q * (size_t)k;
    auto fetch = [&](int s, int t) {
        Ent e;
        e.id = Is[(size_t)s * i_stride + qo + t];
        const float d = Ds[(size_t)s * d_stride + qo + t];
        e.g = (METRIC == MQ_METRIC_L2) ? -d : d;
        return e;
    };
    for (int t = threadIdx.x; t < k; t += 256) {
        D[qo + t] = (METRIC == MQ_METRIC_L2) ? FLT_MAX : -FLT_MAX;
        I[qo + t] = -1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nshards * k; e += 256) {
        const int s = e / k, t = e % k;
        const Ent x = fetch(s, t);
        if (x.id < 0) continue;
        int rank = t;
        for (int o = 0; o < nshards && rank < k; ++o) {
            if (o == s) continue;
            int lo = 0, hi = k;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (better(fetch(o, mid), x, desc)) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            D[qo + rank] = ((METRIC == MQ_METRIC_L2) ? -x.g : x.g) + 0.0f;
            I[qo + rank] = x.id;
        }
    }
}

#include "knn_screen.inc"
#include "knn_small.inc"
#include "knn_small8.inc"
#include "knn_direct.inc"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local int g_last_hip_error = 0;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return MQ_EHIP;
}
#define MQ_HIP(call)                                   \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) return hip_fail(_e);     \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Cycle-accounting builds (-DMQ_TIMING, tools/scan_timing.py) hand the kernels a device buffer for their
// per-wave timestamps through the environment; the shipped library never turns an environment variable
// into a pointer.
inline unsigned long long* dbg_ptr() {
#ifdef MQ_TIMING
    const char* e = getenv("MQ_DBG_PTR");
    return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr;
#else
    return nullptr;
#endif
}

#define MQ_DYNAMIC_LDS(bytes, ...) MQ_DYNAMIC_LDS_WITH(MQ_HIP, bytes, __VA_ARGS__)

int num_cus() {
    static int cached = 0;  // idempotent, benign race
    if (cached) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cached = n;
    return n;
}

struct Geometry {
    int nqt, S, dpad, qpx, qpx_screen;
    int64_t nqpad, nchunks;
    size_t off_qp, off_qn, off_qtmp, off_lists, total;
    // screened path extras
    int dp;
    size_t off_qb, off_margin, off_pcount, off_ovf, off_gthr, off_cand, off_ckeys, off_ccount, off_smax;
    int ms, sps;  // stripe-maxima slots per query / per slab
    // FAISS's small-batch L2 path (knn_direct.inc): transposed queries + the [nq][npad] distance matrix
    size_t off_dqt, off_ddist;
    int64_t npad;
    size_t off_ceil;  // u64 [nqpad]: per-query key ceilings of a search for more than KF neighbours
};

// bf16 row length of the screening copy: d (inner product) or d + 2 (L2: the (h, l) pair of -||x||^2 / 2), padded to 64
// (two extra columns carry the per-row term of the L2 metric and of the centred-query inner product: knn_screen.inc)
int screen_dp(int d, int metric) { return (int)round_up(d + ((metric == MQ_METRIC_L2 || metric == MQ_METRIC_IP_CENTRED) ? 2 : 0), SBK); }
inline bool screen_metric_ok(int metric) { return metric == MQ_METRIC_IP || metric == MQ_METRIC_L2 || metric == MQ_METRIC_IP_CENTRED; }
inline int true_metric(int metric) { return metric == MQ_METRIC_IP_CENTRED ? MQ_METRIC_IP : metric; }

// metric: MQ_METRIC_IP / MQ_METRIC_L2, or -1 = unknown (reserve what either needs).  Only the L2 metric has the small-batch
// direct form whose [nq][npad] distance matrix lives at the END of the workspace, so every other offset is metric-independent.
Geometry geometry(int64_t N, int d, int nq, int k, int cus, int metric = -1) {
    Geometry g;
    g.dpad = (int)round_up(d, BK);
    g.nqpad = round_up(nq > 0 ? nq : 1, TQ);
    g.nqt = (int)(g.nqpad / TQ);
    g.nchunks = round_up(N > 0 ? N : 1, TN) / TN;
    int64_t S = cus / g.nqt;
    if (S < 1) S = 1;
    if (S >= 8) S = S / 8 * 8;
    if (S > g.nchunks) S = g.nchunks;
    g.S = (int)S;
    // Workgroup -> (query tile, slab) placement (speed only).  Workgroup b runs on XCD b % 8: give
    // each XCD `qpx` query tiles so that their Q panels (qpx * 256 * dpad * 4 B, re-read once per KB
    // chunk) stay in that XCD's 4 MiB L2, and let the XCDs that share those query tiles split the slabs.
    g.qpx = 0;
    g.qpx_screen = 0;
    {
        static const int want_env = [] { const char* e = getenv("MQ_KNN_QPX"); return e ? atoi(e) : 0; }();  // tuning knob, read once
        const int want = want_env;
        for (int q = 1; q <= g.nqt; ++q) {
            if (g.nqt % q) continue;
            const int ngroups = g.nqt / q;
            if (ngroups > 8 || 8 % ngroups) continue;
            const int xpg = 8 / ngroups;
            if (g.S % xpg) continue;
            if (want > 0) { if (q == want) g.qpx = g.qpx_screen = q; continue; }
            // default: the largest Q working set that still leaves half of L2 to the KB stream (the screening scan's
            // queries are bf16: twice as many tiles fit, and every XCD group then streams a smaller part of the shard:
            // 4 tiles per XCD instead of 2 took 2.7 % off the 4096-query scan; 8 tiles -- 3.1 MB of bf16 queries at d = 768, the
            // shard then crosses the fabric twice instead of four times -- another 0.8 %: 8.27 -> 8.20 ms; 16 tiles 8.52)
            if ((size_t)q * TQ * g.dpad * 4 <= (size_t)2 << 20 || g.qpx == 0) g.qpx = q;
            if ((size_t)q * TQ * screen_dp(d, MQ_METRIC_IP) * 2 <= (size_t)13 << 18 || g.qpx_screen == 0) g.qpx_screen = q;
        }
    }
    size_t o = 0;
    g.off_qp = o;    o += (size_t)g.nqpad * g.dpad * 4;
    g.off_qn = o;    o += (size_t)g.nqpad * 4;
    g.off_qtmp = o;  o += (size_t)round_up((int64_t)(nq > 0 ? nq : 1) * d * 4, 256);
    g.off_lists = o; o += (size_t)g.nqt * g.S * TQ * (size_t)SPOOL * 8;  // SPOOL >= POOL: shared by both paths
    g.dp = (int)round_up(d + 2, SBK);  // sized for the L2 screen (two extra columns); the search uses screen_dp()
    const size_t nq1 = (size_t)(nq > 0 ? nq : 1);
    g.off_qb = o;     o += (size_t)round_up((int64_t)g.nqpad * g.dp * 2, 256);
    g.off_margin = o; o += (size_t)g.nqpad * 4;
    g.off_pcount = o; o += (size_t)g.nqt * g.S * TQ * NSL * 4;
    g.sps = g.S <= SMAX_SLOTS / 16 ? 16 : (SMAX_SLOTS / g.S > 0 ? SMAX_SLOTS / g.S : 1);
    g.ms = g.S * g.sps < SMAX_SLOTS ? g.S * g.sps : SMAX_SLOTS;
    // ovf | gthr | smax are zeroed by ONE memset per search: keep them adjacent
    g.off_ovf = o;    o += (size_t)round_up((int64_t)g.nqt * 4, 256);
    g.off_gthr = o;   o += (size_t)g.nqpad * 4;
    g.off_smax = o;   o += (size_t)g.nqpad * g.ms * 4;
    g.off_cand = o;   o += nq1 * RMAX * 4;
    g.off_ckeys = o;  o += nq1 * RMAX * 8;
    g.off_ccount = o; o += (size_t)round_up((int64_t)nq1 * 4, 256);
    g.off_ceil = o;   o += (size_t)g.nqpad * 8;
    g.npad = round_up(N > 0 ? N : 1, TN);
    g.off_dqt = g.off_ddist = o;
    if (nq > 0 && nq < MQ_KNN_L2_DIRECT_BELOW && metric != MQ_METRIC_IP) {
        g.off_dqt = o;   o += (size_t)round_up((int64_t)g.dpad * DIRECT_NQ * 4, 256);
        g.off_ddist = o; o += (size_t)nq * (size_t)g.npad * 4;
    }
    g.total = round_up((int64_t)o, 256);
    return g;
}

// The streaming kernel for one query tile (knn_small.inc) serves a screened search when its geometry holds: one query tile,
// one slab per stripe slot (S = 256 = TQ: the workgroup of slab s owns query s), at most 12 K blocks of queries in registers,
// k within the stripe bound, and enough 32-row tiles per slab for the ring to pay.  MQ_KNN_OPT_SMALL_SCAN = 0 switches it off,
// MQ_KNN_OPT_SMALL_MIN_TILES = <n> lowers the tiles-per-slab floor (tests) -- mq_knn_set_option.
// `rowterm` (out): the L2 metric / the centred inner product whose two row-term columns sit alone in a 13th K block (d a multiple of 64, dp = d + 64 = 832):
// the two-waves-per-SIMD kernel reads 12 K blocks and takes the term as fp32 (knn_small8.inc).
bool small_scan_serves(const Geometry& g, int64_t N, int d, int metric, int k, int* nkb_out = nullptr, bool* rowterm_out = nullptr) {
    const int enabled = knn_opt(MQ_KNN_OPT_SMALL_SCAN), floor_opt = knn_opt(MQ_KNN_OPT_SMALL_MIN_TILES);
    const int min_tiles = floor_opt > 0 ? floor_opt : SM_MIN_TILES_PER_SLAB;
    if (!enabled || g.nqt != 1 || g.S != TQ || g.ms != SMAX_SLOTS || g.sps != 1) return false;
    int nkb = screen_dp(d, metric) / SBK;
    bool rowterm = false;
    if (nkb == SM_MAX_NKB + 1 && (metric == MQ_METRIC_L2 || metric == MQ_METRIC_IP_CENTRED) && d % SBK == 0 &&
        knn_opt(MQ_KNN_OPT_SMALL_WAVES) == 8) {
        nkb = SM_MAX_NKB;
        rowterm = true;
    }
    if (nkb > SM_MAX_NKB || k > KF) return false;
    if (nkb_out) *nkb_out = nkb;
    if (rowterm_out) *rowterm_out = rowterm;
    return (N + SM_ROWS - 1) / SM_ROWS >= (int64_t)g.S * (min_tiles > 1 ? min_tiles : 1);
}

template <int NKB>
int launch_small_scan_n(const SmallArgs& sa, int S, hipStream_t st) {
    constexpr int lds = sm_nst(NKB) * (NKB * SM_PIECE + 2 * SM_AUX) + 2 * TQ * 4 + SM_AUX;
    if (knn_opt(MQ_KNN_OPT_SMALL_WAVES) == 8) {  // two waves per SIMD (knn_small8.inc)
        if (NKB == SM_MAX_NKB && sa.rowterm) {  // the row term as fp32 beside 12 K blocks
            constexpr int lds_rt = lds + sm_nst(SM_MAX_NKB) * SM_AUX;  // + the ring slots' row terms
            MQ_DYNAMIC_LDS(lds_rt, screen_small8_kernel<SM_MAX_NKB, true>);
            hipLaunchKernelGGL((screen_small8_kernel<SM_MAX_NKB, true>), dim3((unsigned)S), dim3(512), lds_rt, st, sa);
            return MQ_OK;
        }
        MQ_DYNAMIC_LDS(lds, screen_small8_kernel<NKB>);
        hipLaunchKernelGGL(screen_small8_kernel<NKB>, dim3((unsigned)S), dim3(512), lds, st, sa);
        return MQ_OK;
    }
    MQ_DYNAMIC_LDS(lds, screen_small_kernel<NKB>);
    hipLaunchKernelGGL(screen_small_kernel<NKB>, dim3((unsigned)S), dim3(256), lds, st, sa);
    return MQ_OK;
}

int launch_small_scan(const SmallArgs& sa, int nkb, int S, hipStream_t st) {
    switch (nkb) {
        case 1: return launch_small_scan_n<1>(sa, S, st);
        case 2: return launch_small_scan_n<2>(sa, S, st);
        case 3: return launch_small_scan_n<3>(sa, S, st);
        case 4: return launch_small_scan_n<4>(sa, S, st);
        case 5: return launch_small_scan_n<5>(sa, S, st);
        case 6: return launch_small_scan_n<6>(sa, S, st);
        case 7: return launch_small_scan_n<7>(sa, S, st);
        case 8: return launch_small_scan_n<8>(sa, S, st);
        case 9: return launch_small_scan_n<9>(sa, S, st);
        case 10: return launch_small_scan_n<10>(sa, S, st);
        case 11: return launch_small_scan_n<11>(sa, S, st);
        case 12: return launch_small_scan_n<12>(sa, S, st);
        default: return MQ_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" {

const char* mq_version(void) { return "meerqat_hip 0.6 (gfx950)"; }

const char* mq_strerror(int code) {
    switch (code) {
        case MQ_OK: return "ok";
        case MQ_EINVAL: return "invalid argument";
        case MQ_EWORKSPACE: return "workspace too small";
        case MQ_EHIP: return "HIP runtime error (see mq_last_hip_error)";
        case MQ_EUNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

int mq_knn_set_option(int key, int value) {
    if (key < 0 || key >= MQ_KNN_OPT_COUNT || value < 0) return MQ_EINVAL;
    knn_opt_init();
    return g_knn_opt[key].exchange(value);
}
int mq_knn_get_option(int key) {
    if (key < 0 || key >= MQ_KNN_OPT_COUNT) return MQ_EINVAL;
    return knn_opt(key);
}

int mq_last_hip_error(void) { return g_last_hip_error; }
void mq_internal_set_hip_error(int e) { g_last_hip_error = e; }  /* used by encoder.hip */

int64_t mq_padded_rows(int64_t n_rows) { return round_up(n_rows > 0 ? n_rows : 1, TN); }
int mq_padded_dim(int d) { return (int)round_up(d, BK); }
size_t mq_packed_bytes(int64_t n_rows, int d) { return (size_t)mq_padded_rows(n_rows) * (size_t)mq_padded_dim(d) * 4; }

int mq_pack_rows_f32(const float* rows_dev, int64_t n, int d, int64_t row_offset, int l2norm, float* packed_dev,
                     int64_t capacity_rows, float* sqnorm_dev, void* stream) {
    if (n == 0) return MQ_OK;
    if (!rows_dev || !packed_dev || !sqnorm_dev || n < 0 || d <= 0 || row_offset < 0) return MQ_EINVAL;
    if (row_offset % PANEL != 0 || capacity_rows % TN != 0 || row_offset + n > capacity_rows) return MQ_EINVAL;
    const int dpad = mq_padded_dim(d);
    const unsigned grid = (unsigned)((n + PANEL - 1) / PANEL);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows_dev, n, d, dpad, row_offset,
                       l2norm, packed_dev, sqnorm_dev);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_unpack_rows_f32(const float* packed_dev, int64_t capacity_rows, int d, int64_t row_offset, int64_t n,
                       float* rows_dev, void* stream) {
    if (n == 0) return MQ_OK;
    if (!packed_dev || !rows_dev || n < 0 || d <= 0 || row_offset < 0 || row_offset + n > capacity_rows) return MQ_EINVAL;
    const int dpad = mq_padded_dim(d);
    const int64_t total = n * (int64_t)d;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, packed_dev, d, dpad, row_offset, n,
                       rows_dev);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_l2norm_rows_form_f32(float* rows_dev, int64_t n, int d, int form, void* stream) {
    if (n == 0) return MQ_OK;
    if (!rows_dev || n < 0 || d <= 0 || (form != MQ_L2NORM_NUMPY && form != MQ_L2NORM_FAISS)) return MQ_EINVAL;
    launch_l2norm_rows(rows_dev, n, d, form, (hipStream_t)stream);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_l2norm_rows_f32(float* rows_dev, int64_t n, int d, void* stream) {
    return mq_l2norm_rows_form_f32(rows_dev, n, d, MQ_L2NORM_NUMPY, stream);
}

// ---- k beyond the screen's own range (SCREEN_MAX_K) ------------------------------------------------------------------------
// The screen's bounded buffers serve k <= 224 (256 stripe maxima per query, 256 keys in the final sort).  A larger k is served by
// PARTITIONS: the rows are cut into P contiguous ranges, each range is searched for its exact top-224 by the screened pipeline, the
// P lists are merged, and the merge PROVES the result: the top-k of the union is the top-k of the lists unless some range delivered
// its whole list into it (it might have held a 225th) -- then that query tile is recomputed by the exact rounds, like a tile whose
// screening buffers overflowed.  P = ceil(k / 112): on exchangeable data a range holds ~k / P <= 112 of the top k, half of what it
// can deliver; a KB sorted by similarity to the query defeats it and pays the exact rounds for those tiles.  Cost: one pass of the
// scan over all rows + P tails, instead of ceil(k / 128) exact fp32 scans (k = 256 at 1.5M x 768, 4096 queries: 146 ms).
constexpr int PART_K = SCREEN_MAX_K;       // neighbours asked of each range
constexpr int PART_SHARE = 112;            // expected neighbours per range at most
constexpr int PART_MAX = 16;               // ranges at most: k <= 1792
constexpr int PART_MIN_ROWS = 16384;       // ranges smaller than this: the exact rounds are as fast

struct PartPlan { int P; int64_t per; };
static PartPlan partition_plan(int64_t N, int k) {
    PartPlan pl{0, 0};
    if (!knn_opt(MQ_KNN_OPT_PARTITIONS) || k <= SCREEN_MAX_K) return pl;
    const int P = (k + PART_SHARE - 1) / PART_SHARE;
    if (P > PART_MAX) return pl;
    const int64_t per = round_up((N + P - 1) / P, TQ);  // the bf16 copy is stored in 256-row tiles
    if (per < PART_MIN_ROWS || (int64_t)(P - 1) * per >= N) return pl;
    pl.P = P; pl.per = per;
    return pl;
}
static size_t partition_extra_bytes(int64_t N, int nq, int k) {
    const PartPlan pl = partition_plan(N, k);
    if (!pl.P) return 0;
    const size_t e = (size_t)pl.P * (size_t)(nq > 0 ? nq : 1) * PART_K;
    return round_up((int64_t)(e * 4), 256) + round_up((int64_t)(e * 8), 256) + round_up((int64_t)((nq + TQ - 1) / TQ + 1) * 4, 256);
}

// One workgroup per query: the P lists (PART_K entries each, best first) -> the k best of their union, best first, and the proof
// obligation: a range all of whose PART_K entries are inside the answer (and which holds more rows than that) flags the query's tile.
__global__ __launch_bounds__(256) void partition_merge_kernel(const float* __restrict__ Dp, const long long* __restrict__ Ip, int P, int nq,
                                                              int k, int l2, unsigned flip, long long id_offset, long long per,
                                                              long long N, float* __restrict__ D, long long* __restrict__ I,
                                                              int* __restrict__ flags) {
    __shared__ u64 keys[PART_MAX * 256];
    __shared__ int cnt[PART_MAX];
    const int q = blockIdx.x, t = threadIdx.x;
    const int total = P * PART_K;
    int n2 = 256;
    while (n2 < total) n2 <<= 1;
    if (t < PART_MAX) cnt[t] = 0;
    for (int e = t; e < n2; e += 256) {
        u64 key = 0ull;
        if (e < total) {
            const int p = e / PART_K, j = e - p * PART_K;
            const size_t at = ((size_t)p * nq + q) * PART_K + j;
            const long long id = Ip[at];
            if (id >= 0) {
                const float sc = Dp[at];
                key = make_key(l2 ? -sc : sc, (unsigned)id ^ flip);
            }
        }
        keys[e] = key;
    }
    __syncthreads();
    // bitonic sort, descending
    for (int sz = 2; sz <= n2; sz <<= 1)
        for (int j = sz >> 1; j > 0; j >>= 1) {
            for (int e = t; e < n2 / 2; e += 256) {
                const int lo = ((e & ~(j - 1)) << 1) | (e & (j - 1)), hi = lo | j;
                const bool desc = (lo & sz) == 0;
                const u64 x = keys[lo], y = keys[hi];
                if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    for (int e = t; e < k; e += 256) {
        const u64 key = keys[e];
        float sc = l2 ? FLT_MAX : -FLT_MAX;  // FAISS's heap neutral values in unfilled slots
        long long id = -1;
        if (key) {
            const unsigned row = key_row(key) ^ flip;
            const float g = key_score(key);
            sc = l2 ? -g : g;
            id = (long long)row + id_offset;
            atomicAdd(&cnt[(int)(row / (unsigned long long)per)], 1);
        }
        D[(size_t)q * k + e] = sc;
        I[(size_t)q * k + e] = id;
    }
    __syncthreads();
    if (t < P) {
        const long long rows = (t + 1) * per <= N ? per : N - t * per;
        if (cnt[t] >= PART_K && rows > PART_K) flags[q / TQ] = 1;
    }
}

size_t mq_knn_workspace_bytes(int64_t N, int d, int nq, int k) {
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || k > MQ_KNN_MAX_K) return 0;
    return geometry(N, d, nq, k, num_cus()).total + partition_extra_bytes(N, nq, k);
}

size_t mq_knn_workspace_bytes_metric(int64_t N, int d, int nq, int k, int metric) {
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || k > MQ_KNN_MAX_K || !screen_metric_ok(metric)) return 0;
    return geometry(N, d, nq, k, num_cus(), true_metric(metric)).total + partition_extra_bytes(N, nq, k);
}

int mq_knn_launch_info(int64_t N, int d, int nq, int k, int64_t out[8]) {
    if (!out || N < 0 || d <= 0 || nq < 0 || k <= 0 || k > MQ_KNN_MAX_K) return MQ_EINVAL;
    const Geometry g = geometry(N, d, nq, k, num_cus());
    out[0] = (int64_t)g.nqt * g.S;
    out[1] = 1024;
    out[2] = LDS_TOTAL;
    out[3] = g.nqt;
    out[4] = g.S;
    out[5] = g.nchunks;
    out[6] = 1024;         // screening scan: threads per workgroup
    out[7] = S_LDS_TOTAL;  // screening scan: LDS bytes
    return MQ_OK;
}

int mq_knn_screen_scan_kind(int64_t N, int d, int nq, int k, int metric) {
    if (N <= 0 || d <= 0 || nq <= 0 || k <= 0 || k > MQ_KNN_MAX_K) return MQ_EINVAL;
    if (!screen_metric_ok(metric)) return MQ_EINVAL;
    if (metric == MQ_METRIC_L2 && nq < MQ_KNN_L2_DIRECT_BELOW) return MQ_SCAN_KIND_NONE;
    if (const PartPlan pl = partition_plan(N, k); pl.P) { N = pl.per; k = PART_K; }  // the scan of each row range
    if (k > SCREEN_MAX_K) return MQ_SCAN_KIND_NONE;
    const Geometry g = geometry(N, d, nq, k, num_cus(), true_metric(metric));
    return small_scan_serves(g, N, d, metric, k) ? MQ_SCAN_KIND_STREAM : MQ_SCAN_KIND_TILE;
}

// "L2norm," arithmetic of a call's query transform: 0 = none, MQ_L2NORM_NUMPY, MQ_L2NORM_FAISS (the flag alone means FAISS form too)
static inline int query_l2norm_form(int flags) {
    if (flags & MQ_KNN_FLAG_L2NORM_FAISS) return MQ_L2NORM_FAISS;
    return (flags & MQ_KNN_FLAG_L2NORM_QUERIES) ? MQ_L2NORM_NUMPY : 0;
}
constexpr int MQ_KNN_ALL_FLAGS = MQ_KNN_FLAG_L2NORM_QUERIES | MQ_KNN_FLAG_TIE_ID_DESC | MQ_KNN_FLAG_L2NORM_FAISS;
constexpr int MQ_KNN_SCREENED_FLAGS = MQ_KNN_ALL_FLAGS | MQ_KNN_FLAG_PHASE_FRONT | MQ_KNN_FLAG_PHASE_TAIL;

// tie order of a call: key low word = ~(row ^ flip)
static inline unsigned tie_flip(int flags) { return (flags & MQ_KNN_FLAG_TIE_ID_DESC) ? 0xFFFFFFFFu : 0u; }

// rounds of a search: ceil(k / KF) fused selections of up to KF neighbours each; round r > 0 only admits keys strictly below
// the last key of round r - 1 (keys are unique, so the rounds' results concatenate into the exact sorted top-k)
static inline int n_rounds(int k) { return (k + KF - 1) / KF; }

// metric L2 with fewer than 20 queries: FAISS's sequential path, d = sum_k (q[k] - x[k])^2 (knn_direct.inc)
static int knn_search_l2_direct(const float* packed_dev, const float* rowmajor_dev, int64_t N, int d, const float* queries_dev, int nq, int k,
                                int flags, int64_t id_offset, float* D_dev, int64_t* I_dev, void* ws_dev,
                                const Geometry& g, hipStream_t st) {
    char* ws = (char*)ws_dev;
    float* qtmp = (float*)(ws + g.off_qtmp);
    float* Qt = (float*)(ws + g.off_dqt);
    float* dist = (float*)(ws + g.off_ddist);
    u64* lists = (u64*)(ws + g.off_lists);
    u64* ceil = (u64*)(ws + g.off_ceil);
    const unsigned flip = tie_flip(flags);
    const float* q_rm = queries_dev;
    if (query_l2norm_form(flags)) {
        MQ_HIP(hipMemcpyAsync(qtmp, queries_dev, (size_t)nq * d * 4, hipMemcpyDeviceToDevice, st));
        launch_l2norm_rows(qtmp, (int64_t)nq, d, query_l2norm_form(flags), st);
        MQ_HIP(hipGetLastError());
        q_rm = qtmp;
    }
    hipLaunchKernelGGL(direct_transpose_queries_kernel, dim3((unsigned)((g.dpad * DIRECT_NQ + 255) / 256)), dim3(256), 0, st, q_rm,
                       nq, d, g.dpad, Qt);
    MQ_HIP(hipGetLastError());
    const long long seg = (long long)((g.nchunks + g.S - 1) / g.S) * TN;  // rows per selection segment
    if (N > 0) {
        if (packed_dev)
            hipLaunchKernelGGL(l2_direct_dist_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, packed_dev, Qt, (long long)N,
                               d, g.dpad, nq, (long long)g.npad, dist);
        else
            hipLaunchKernelGGL(l2_direct_dist_rows_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rowmajor_dev, Qt,
                               (long long)N, d, nq, (long long)g.npad, dist);
        MQ_HIP(hipGetLastError());
    }
    const int R = n_rounds(k);
    for (int r = 0; r < R; ++r) {
        const int kr = k - r * KF < KF ? k - r * KF : KF;
        // segment winners live in the (much larger) pool region of the other paths: nq * S * 128 keys
        hipLaunchKernelGGL(l2_direct_select_kernel, dim3((unsigned)g.S, (unsigned)nq), dim3(256), 0, st, dist, (long long)N,
                           (long long)g.npad, seg, kr, lists, flip, r ? (const u64*)ceil : (const u64*)nullptr);
        MQ_HIP(hipGetLastError());
        hipLaunchKernelGGL(l2_direct_final_kernel, dim3((unsigned)nq), dim3(256), 0, st, lists, g.S, kr, (long long)id_offset, D_dev,
                           (long long*)I_dev, flip, k, r * KF, r + 1 < R ? ceil : (u64*)nullptr);
        MQ_HIP(hipGetLastError());
    }
    return MQ_OK;
}

}  // extern "C"
// launch of one exact scan (8 instantiations: metric x operand layout x key ceiling)
template <int METRIC>
static int launch_scan_m(bool xrow, bool ceil, const ScanArgs& a, hipStream_t st) {
    const dim3 grid((unsigned)(a.nqt * a.S)), block(1024);
    if (!xrow && !ceil) { MQ_DYNAMIC_LDS(LDS_TOTAL, knn_scan_kernel<METRIC, false, false>); hipLaunchKernelGGL((knn_scan_kernel<METRIC, false, false>), grid, block, LDS_TOTAL, st, a); }
    else if (xrow && !ceil) { MQ_DYNAMIC_LDS(LDS_TOTAL, knn_scan_kernel<METRIC, true, false>); hipLaunchKernelGGL((knn_scan_kernel<METRIC, true, false>), grid, block, LDS_TOTAL, st, a); }
    else if (!xrow) { MQ_DYNAMIC_LDS(LDS_TOTAL, knn_scan_kernel<METRIC, false, true>); hipLaunchKernelGGL((knn_scan_kernel<METRIC, false, true>), grid, block, LDS_TOTAL, st, a); }
    else { MQ_DYNAMIC_LDS(LDS_TOTAL, knn_scan_kernel<METRIC, true, true>); hipLaunchKernelGGL((knn_scan_kernel<METRIC, true, true>), grid, block, LDS_TOTAL, st, a); }
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}
static int launch_scan(int metric, bool xrow, bool ceil, const ScanArgs& a, hipStream_t st) {
    return metric == MQ_METRIC_L2 ? launch_scan_m<MQ_METRIC_L2>(xrow, ceil, a, st) : launch_scan_m<MQ_METRIC_IP>(xrow, ceil, a, st);
}
extern "C" {
static int launch_slab_merge(int metric, const u64* lists, int nq, int S, int k, int64_t id_offset, float* D_dev, int64_t* I_dev,
                             const int* only, unsigned flip, int kout, int col0, u64* next_ceil, hipStream_t st) {
    if (metric == MQ_METRIC_IP)
        hipLaunchKernelGGL(slab_merge_kernel<MQ_METRIC_IP>, dim3((unsigned)nq), dim3(128), 0, st, lists, nq, S, k, (long long)id_offset,
                           D_dev, (long long*)I_dev, only, flip, kout, col0, next_ceil);
    else
        hipLaunchKernelGGL(slab_merge_kernel<MQ_METRIC_L2>, dim3((unsigned)nq), dim3(128), 0, st, lists, nq, S, k, (long long)id_offset,
                           D_dev, (long long*)I_dev, only, flip, kout, col0, next_ceil);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

// The exact scan over a shard whose query panels (a.Qp, a.qn) are ready: one fused scan + slab merge per round of KF
// neighbours (ONE round for k <= 128, the reference's k = 100).  a.Xp = the panel copy, or the row-major copy (xrow).
static int exact_scan_rounds(int metric, bool xrow, ScanArgs a, const Geometry& g, int nq, int k, int64_t id_offset,
                             float* D_dev, int64_t* I_dev, char* ws, const int* only, hipStream_t st, void* ev0, void* ev1) {
    u64* ceil = (u64*)(ws + g.off_ceil);
    const int R = n_rounds(k);
    for (int r = 0; r < R; ++r) {
        const int kr = k - r * KF < KF ? k - r * KF : KF;
        a.k = kr;
        a.ceil = r ? ceil : nullptr;
        a.only = only;
        if (r == 0 && ev0) MQ_HIP(hipEventRecord((hipEvent_t)ev0, st));
        const int rc = launch_scan(metric, xrow, r > 0, a, st);
        if (rc != MQ_OK) return rc;
        if (r == 0 && ev1) MQ_HIP(hipEventRecord((hipEvent_t)ev1, st));
        const int rm = launch_slab_merge(metric, a.lists, nq, g.S, kr, id_offset, D_dev, I_dev, only, a.flip, k, r * KF,
                                         r + 1 < R ? ceil : (u64*)nullptr, st);
        if (rm != MQ_OK) return rm;
    }
    return MQ_OK;
}

static int knn_search_impl(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                           int nq, int k, int metric, int flags, int64_t id_offset, float* D_dev, int64_t* I_dev,
                           void* ws_dev, size_t ws_bytes, void* stream, void* ev_scan_begin, void* ev_scan_end) {
    if (nq == 0) return MQ_OK;
    if (!packed_dev || !sqnorm_dev || !queries_dev || !D_dev || !I_dev || !ws_dev) return MQ_EINVAL;
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || (metric != MQ_METRIC_IP && metric != MQ_METRIC_L2)) return MQ_EINVAL;
    if (flags & ~MQ_KNN_ALL_FLAGS) return MQ_EINVAL;
    if (k > MQ_KNN_MAX_K) return MQ_EUNSUPPORTED;
    if (N >= 0xFFFFFFFFll) return MQ_EUNSUPPORTED;  // 32-bit local row ids
    const Geometry g = geometry(N, d, nq, k, num_cus(), metric);
    if (ws_bytes < g.total) return MQ_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_dev;
    float* Qp = (float*)(ws + g.off_qp);
    float* qn = (float*)(ws + g.off_qn);
    u64* lists = (u64*)(ws + g.off_lists);
    if (metric == MQ_METRIC_L2 && nq < MQ_KNN_L2_DIRECT_BELOW)
        return knn_search_l2_direct(packed_dev, nullptr, N, d, queries_dev, nq, k, flags, id_offset, D_dev, I_dev, ws_dev, g, st);

    // queries -> panel layout (+ optional "L2norm," transform, + ||q||^2); padded queries are zero
    MQ_HIP(hipMemsetAsync(Qp, 0, (size_t)g.nqpad * g.dpad * 4 + (size_t)g.nqpad * 4, st));
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((nq + PANEL - 1) / PANEL)), dim3(256), 0, st, queries_dev, (int64_t)nq, d,
                       g.dpad, (int64_t)0, query_l2norm_form(flags), Qp, qn, (const int*)nullptr);
    MQ_HIP(hipGetLastError());

    ScanArgs a;
    a.Xp = packed_dev; a.Qp = Qp; a.xn = sqnorm_dev; a.qn = qn; a.lists = lists;
    a.N = N; a.dpad = g.dpad; a.nqt = g.nqt; a.S = g.S; a.k = k; a.nchunks = g.nchunks; a.qpx = g.qpx; a.dbg = dbg_ptr(); a.only = nullptr; a.d = d;
    a.flip = tie_flip(flags); a.ceil = nullptr;
    if (N > 0)
        return exact_scan_rounds(metric, false, a, g, nq, k, id_offset, D_dev, I_dev, ws, nullptr, st, ev_scan_begin, ev_scan_end);
    MQ_HIP(hipMemsetAsync(lists, 0, (size_t)g.nqt * g.S * TQ * (size_t)POOL * 8, st));
    for (int r = 0; r < n_rounds(k); ++r) {
        const int kr = k - r * KF < KF ? k - r * KF : KF;
        const int rm = launch_slab_merge(metric, lists, nq, g.S, kr, id_offset, D_dev, I_dev, nullptr, a.flip, k, r * KF, nullptr, st);
        if (rm != MQ_OK) return rm;
    }
    return MQ_OK;
}

int mq_knn_search_f32(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                      int nq, int k, int metric, int flags, int64_t id_offset, float* D_dev, int64_t* I_dev,
                      void* ws_dev, size_t ws_bytes, void* stream) {
    return knn_search_impl(packed_dev, sqnorm_dev, N, d, queries_dev, nq, k, metric, flags, id_offset, D_dev,
                           I_dev, ws_dev, ws_bytes, stream, nullptr, nullptr);
}

int mq_knn_search_f32_ev(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                         int nq, int k, int metric, int flags, int64_t id_offset, float* D_dev, int64_t* I_dev,
                         void* ws_dev, size_t ws_bytes, void* stream, void* ev_scan_begin, void* ev_scan_end) {
    return knn_search_impl(packed_dev, sqnorm_dev, N, d, queries_dev, nq, k, metric, flags, id_offset, D_dev,
                           I_dev, ws_dev, ws_bytes, stream, ev_scan_begin, ev_scan_end);
}

/* ---- screened path (bf16 screening + exact re-scoring): see knn_screen.inc ---- */
size_t mq_knn_screen_bytes(int64_t n_rows, int d, int metric) {
    return (size_t)mq_padded_rows(n_rows) * (size_t)screen_dp(d, metric) * 2;
}

// bf16 screening copy + error statistics of rows [row_offset, row_offset + n) of the row-major store
static int screen_copy_and_stats(const float* sqnorm_dev, int d, int metric, int64_t row_offset, int64_t n, float* rowmajor_dev,
                                 uint16_t* bf16_dev, float* xstats_dev, const float* center_dev, hipStream_t st) {
    const int dp = screen_dp(d, metric);
    const float* rm = rowmajor_dev + (size_t)row_offset * d;
    const int64_t quads = n * (int64_t)(dp / 4);
    if (metric == MQ_METRIC_IP_CENTRED)  // rows of x - c and their row term c . (x - c): 64 rows per workgroup
        hipLaunchKernelGGL(to_bf16_rows_centred_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, rm, n, d, dp,
                           (unsigned short*)bf16_dev, center_dev, row_offset);
    else
        hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, rm, n, d, dp,
                           (unsigned short*)bf16_dev, metric == MQ_METRIC_L2 ? 1 : 0, sqnorm_dev + row_offset, center_dev, row_offset);
    MQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(row_err_stats_kernel, dim3((unsigned)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048)), dim3(256), 0, st, rm,
                       (const unsigned short*)bf16_dev, n, d, dp, (unsigned*)xstats_dev, center_dev, row_offset);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_knn_screen_prepare(const float* packed_dev, const float* sqnorm_dev, int64_t capacity_rows, int d, int metric,
                          int64_t row_offset, int64_t n, float* rowmajor_dev, uint16_t* bf16_dev, float* xstats_dev,
                          const float* center_dev, void* stream) {
    if (n == 0) return MQ_OK;
    if (!packed_dev || !sqnorm_dev || !rowmajor_dev || !bf16_dev || !xstats_dev || n < 0 || d <= 0 || row_offset < 0 ||
        row_offset + n > capacity_rows)
        return MQ_EINVAL;
    if (!screen_metric_ok(metric) || (metric == MQ_METRIC_IP_CENTRED && !center_dev)) return MQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int dpad = mq_padded_dim(d);
    float* rm = rowmajor_dev + (size_t)row_offset * d;
    const int64_t total = n * (int64_t)d;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, packed_dev, d, dpad, row_offset, n, rm);
    MQ_HIP(hipGetLastError());
    return screen_copy_and_stats(sqnorm_dev, d, metric, row_offset, n, rowmajor_dev, bf16_dev, xstats_dev, center_dev, st);
}

int mq_knn_screen_add_rows_f32(const float* rows_dev, int64_t n, int d, int64_t row_offset, int l2norm, int metric,
                               int64_t capacity_rows, float* sqnorm_dev, float* rowmajor_dev, uint16_t* bf16_dev,
                               float* xstats_dev, const float* center_dev, void* stream) {
    if (n == 0) return MQ_OK;
    if (!rows_dev || !sqnorm_dev || !rowmajor_dev || !bf16_dev || !xstats_dev || n < 0 || d <= 0 || row_offset < 0 ||
        row_offset + n > capacity_rows)
        return MQ_EINVAL;
    if (!screen_metric_ok(metric) || (metric == MQ_METRIC_IP_CENTRED && !center_dev)) return MQ_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(store_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, rows_dev, n, d, mq_padded_dim(d), row_offset,
                       l2norm, rowmajor_dev, sqnorm_dev);
    MQ_HIP(hipGetLastError());
    return screen_copy_and_stats(sqnorm_dev, d, metric, row_offset, n, rowmajor_dev, bf16_dev, xstats_dev, center_dev, st);
}

int mq_knn_search_screened_f32(const float* packed_dev, const float* sqnorm_dev, const float* rowmajor_dev,
                               const uint16_t* bf16_dev, const float* xstats_dev, int64_t N, int d, const float* queries_dev,
                               int nq, int k, int metric, int flags, int64_t id_offset, float* D_dev, int64_t* I_dev,
                               void* ws_dev, size_t ws_bytes, void* stream, void* ev_scan_begin, void* ev_scan_end) {
    if (nq == 0) return MQ_OK;
    if (!screen_metric_ok(metric)) return MQ_EINVAL;
    if (flags & ~MQ_KNN_SCREENED_FLAGS) return MQ_EINVAL;
    // the two halves of one search (MQ_KNN_FLAG_PHASE_*): neither or both bits = the whole search
    const int phase = flags & (MQ_KNN_FLAG_PHASE_FRONT | MQ_KNN_FLAG_PHASE_TAIL);
    const bool do_front = phase != MQ_KNN_FLAG_PHASE_TAIL, do_tail = phase != MQ_KNN_FLAG_PHASE_FRONT;
    const int l2norm_queries = query_l2norm_form(flags);
    const unsigned flip = tie_flip(flags);
    // MQ_METRIC_IP_CENTRED: the index was built for the centred-query screen (two extra columns = the row term c . (x - c), the
    // centre itself behind the statistics: xstats_dev[4 .. 4 + d)); everything exact -- re-scoring, fallback scans, merges -- is
    // the plain inner product
    const int l2 = metric == MQ_METRIC_L2;
    const int ipc = metric == MQ_METRIC_IP_CENTRED;
    const int tm = true_metric(metric);
    const int dp = screen_dp(d, metric);
    // packed_dev may be NULL (an index that keeps no panel copy): the exact-scan fallback and FAISS's small-batch L2 form then
    // read the row-major copy
    if (!sqnorm_dev || !rowmajor_dev || !bf16_dev || !xstats_dev || !queries_dev || !D_dev || !I_dev || !ws_dev)
        return MQ_EINVAL;
    if (N <= 0 || d <= 0 || nq < 0 || k <= 0) return MQ_EINVAL;
    if (k > MQ_KNN_MAX_K) return MQ_EUNSUPPORTED;
    if (N >= 0xFFFFFFFFll) return MQ_EUNSUPPORTED;
    const Geometry g = geometry(N, d, nq, k, num_cus(), tm);
    if (ws_bytes < g.total) return MQ_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // searches served without the screen are one piece: the FRONT call does all of it, the TAIL call has nothing left to do
    if ((l2 && nq < MQ_KNN_L2_DIRECT_BELOW) || k > SCREEN_MAX_K) {
        if (!do_front) return MQ_OK;
    }
    if (l2 && nq < MQ_KNN_L2_DIRECT_BELOW)
        return knn_search_l2_direct(packed_dev, rowmajor_dev, N, d, queries_dev, nq, k, flags & MQ_KNN_ALL_FLAGS, id_offset, D_dev,
                                    I_dev, ws_dev, g, st);
    char* ws = (char*)ws_dev;
    float* Qp = (float*)(ws + g.off_qp);
    float* qn = (float*)(ws + g.off_qn);
    float* qtmp = (float*)(ws + g.off_qtmp);
    u64* pools = (u64*)(ws + g.off_lists);
    unsigned short* Qb = (unsigned short*)(ws + g.off_qb);
    float* margin = (float*)(ws + g.off_margin);
    int* pcount = (int*)(ws + g.off_pcount);
    int* ovf = (int*)(ws + g.off_ovf);
    unsigned* cand = (unsigned*)(ws + g.off_cand);
    u64* ckeys = (u64*)(ws + g.off_ckeys);
    int* ccount = (int*)(ws + g.off_ccount);

    // queries: optional "L2norm," transform applied ONCE in row-major form (so that the panel copy, the
    // bf16 copy and the re-scoring all see the same fp32 values), then panel pack (+ ||q||^2) and bf16 copy
    const float* q_rm = queries_dev;
    if (l2norm_queries) {
        if (do_front) {
            MQ_HIP(hipMemcpyAsync(qtmp, queries_dev, (size_t)nq * d * 4, hipMemcpyDeviceToDevice, st));
            launch_l2norm_rows(qtmp, (int64_t)nq, d, l2norm_queries, st);
            MQ_HIP(hipGetLastError());
        }
        q_rm = qtmp;
    }
    if (const PartPlan pl = partition_plan(N, k); pl.P) {
        // SCREEN_MAX_K < k <= 1792: P contiguous row ranges, each searched for its exact top-224 by this same function, then merged
        // and proved (partition_merge_kernel); flagged query tiles are recomputed by the exact rounds below
        char* extra = ws + g.total;
        const size_t e = (size_t)pl.P * (size_t)nq * PART_K;
        float* Dp = (float*)extra;
        long long* Ip = (long long*)(extra + round_up((int64_t)(e * 4), 256));
        int* pflags = (int*)((char*)Ip + round_up((int64_t)(e * 8), 256));
        if (ws_bytes < g.total + partition_extra_bytes(N, nq, k)) return MQ_EWORKSPACE;
        const int dpb = screen_dp(d, metric);
        for (int p_ = 0; p_ < pl.P; ++p_) {
            const int64_t r0 = (int64_t)p_ * pl.per, rows = r0 + pl.per <= N ? pl.per : N - r0;
            const int rc = mq_knn_search_screened_f32(nullptr, sqnorm_dev + r0, rowmajor_dev + (size_t)r0 * d, bf16_dev + (size_t)r0 * dpb,
                                                      xstats_dev, rows, d, queries_dev, nq, PART_K, metric, flags & MQ_KNN_ALL_FLAGS, r0,
                                                      Dp + (size_t)p_ * nq * PART_K, (int64_t*)Ip + (size_t)p_ * nq * PART_K, ws_dev,
                                                      g.total, stream, p_ == 0 ? ev_scan_begin : nullptr,
                                                      p_ == pl.P - 1 ? ev_scan_end : nullptr);
            if (rc != MQ_OK) return rc;
        }
        MQ_HIP(hipMemsetAsync(pflags, 0, (size_t)g.nqt * 4, st));
        hipLaunchKernelGGL(partition_merge_kernel, dim3((unsigned)nq), dim3(256), 0, st, Dp, Ip, pl.P, nq, k, l2, flip, (long long)id_offset,
                           (long long)pl.per, (long long)N, D_dev, (long long*)I_dev, pflags);
        MQ_HIP(hipGetLastError());
        // flagged tiles: the exact rounds (no-op otherwise: every workgroup of an unflagged tile returns at once)
        MQ_HIP(hipMemsetAsync(Qp, 0, (size_t)g.nqpad * g.dpad * 4 + (size_t)g.nqpad * 4, st));
        hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((nq + PANEL - 1) / PANEL)), dim3(256), 0, st, q_rm, (int64_t)nq, d,
                           g.dpad, (int64_t)0, 0, Qp, qn, (const int*)pflags);
        MQ_HIP(hipGetLastError());
        ScanArgs a;
        a.Xp = packed_dev ? packed_dev : rowmajor_dev; a.Qp = Qp; a.xn = sqnorm_dev; a.qn = qn; a.lists = pools; a.d = d;
        a.N = N; a.dpad = g.dpad; a.nqt = g.nqt; a.S = g.S; a.k = k; a.nchunks = g.nchunks; a.qpx = g.qpx; a.dbg = nullptr; a.only = pflags;
        a.flip = flip; a.ceil = nullptr;
        return exact_scan_rounds(tm, packed_dev == nullptr, a, g, nq, k, id_offset, D_dev, I_dev, ws, pflags, st, nullptr, nullptr);
    }
    if (k > SCREEN_MAX_K) {
        // More than SCREEN_MAX_K neighbours: the bounded screening buffers are sized for the reference's k = 100 (the pools hold
        // 512 survivors per slab, the stripe bound works on 256 slots: k <= 224 still goes through the screen, round 3); the exact
        // scan serves the call in ceil(k / 128) rounds (from the panel copy, or from the row-major copy of an index without one).
        MQ_HIP(hipMemsetAsync(Qp, 0, (size_t)g.nqpad * g.dpad * 4 + (size_t)g.nqpad * 4, st));
        hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((nq + PANEL - 1) / PANEL)), dim3(256), 0, st, q_rm, (int64_t)nq, d,
                           g.dpad, (int64_t)0, 0, Qp, qn, (const int*)nullptr);
        MQ_HIP(hipGetLastError());
        ScanArgs a;
        a.Xp = packed_dev ? packed_dev : rowmajor_dev; a.Qp = Qp; a.xn = sqnorm_dev; a.qn = qn; a.lists = pools; a.d = d;
        a.N = N; a.dpad = g.dpad; a.nqt = g.nqt; a.S = g.S; a.k = k; a.nchunks = g.nchunks; a.qpx = g.qpx; a.dbg = nullptr; a.only = nullptr;
        a.flip = flip; a.ceil = nullptr;
        return exact_scan_rounds(tm, packed_dev == nullptr, a, g, nq, k, id_offset, D_dev, I_dev, ws, nullptr, st, ev_scan_begin,
                                 ev_scan_end);
    }
    if (do_front) {
    // The fp32 panel copy of the queries (+ ||q||^2) serves the exact-scan fallback -- and, for the L2 metric, the
    // re-scoring (||q||^2).  With the inner product it is made after the screened pipeline, for flagged tiles only.
    if (l2) {
        MQ_HIP(hipMemsetAsync(Qp, 0, (size_t)g.nqpad * g.dpad * 4 + (size_t)g.nqpad * 4, st));
        hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((nq + PANEL - 1) / PANEL)), dim3(256), 0, st, q_rm, (int64_t)nq, d,
                           g.dpad, (int64_t)0, 0, Qp, qn, (const int*)nullptr);
        MQ_HIP(hipGetLastError());
    }
    if (g.nqpad > nq)  // bf16 rows of the padding queries: the last tile (tile layout: its real rows are written below)
        MQ_HIP(hipMemsetAsync(Qb + (size_t)(g.nqpad - TQ) * dp, 0, (size_t)TQ * dp * 2, st));
    {
        const int64_t quads = (int64_t)nq * (dp / 4);
        hipLaunchKernelGGL(to_bf16_rows_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, q_rm, (int64_t)nq, d, dp, Qb,
                           (l2 || ipc) ? 2 : 0, (const float*)nullptr, ipc ? xstats_dev + 4 : (const float*)nullptr, (int64_t)0);
        MQ_HIP(hipGetLastError());
        hipLaunchKernelGGL(screen_margin_kernel, dim3((unsigned)((g.nqpad + 3) / 4)), dim3(256), 0, st, q_rm, Qb, xstats_dev, nq,
                           (int)g.nqpad, d, dp, margin, l2 ? 1 : (ipc ? 2 : 0), ipc ? xstats_dev + 4 : (const float*)nullptr,
                           (unsigned*)ovf, (g.off_smax + (size_t)g.nqpad * g.ms * 4 - g.off_ovf) / 4);  // clears ovf + gthr + smax
        MQ_HIP(hipGetLastError());
    }
    // 1. bf16 screening scan
    {
        ScreenArgs a;
        a.Xb = bf16_dev; a.Qb = Qb; a.margin = margin; a.pools = pools; a.pcount = pcount; a.ovf = ovf; a.gthr = (unsigned*)(ws + g.off_gthr);
        a.smax = (unsigned*)(ws + g.off_smax); a.ms = g.ms; a.sps = g.sps;
        a.dbg = dbg_ptr();
        a.N = N; a.dp = dp; a.nqt = g.nqt; a.S = g.S; a.k = k; a.qpx = g.qpx_screen; a.nchunks = g.nchunks;
        if (ev_scan_begin) MQ_HIP(hipEventRecord((hipEvent_t)ev_scan_begin, st));
        int small_nkb = 0;
        bool small_rowterm = false;
        if (small_scan_serves(g, N, d, metric, k, &small_nkb, &small_rowterm)) {
            // one query tile: the streaming kernel with the queries in registers (knn_small8.inc / knn_small.inc)
            SmallArgs sa;
            sa.s = a;
            sa.ntiles = (N + SM_ROWS - 1) / SM_ROWS;
            sa.rowterm = small_rowterm ? 1 : 0;
            sa.sqn = (small_rowterm && metric == MQ_METRIC_L2) ? sqnorm_dev : nullptr;
            sa.nkb_copy = dp / SBK;
            const int rc = launch_small_scan(sa, small_nkb, g.S, st);
            if (rc != MQ_OK) return rc;
        } else {
            MQ_DYNAMIC_LDS(S_LDS_TOTAL, screen_scan_kernel);
            hipLaunchKernelGGL(screen_scan_kernel, dim3((unsigned)(g.nqt * g.S)), dim3(1024), S_LDS_TOTAL, st, a);
        }
        MQ_HIP(hipGetLastError());
        if (ev_scan_end) MQ_HIP(hipEventRecord((hipEvent_t)ev_scan_end, st));
    }
    }  // do_front
    if (!do_tail) return MQ_OK;
    // 2.-4. candidates -> exact scores -> exact top-k
#ifdef MQ_TIMING
    {
        unsigned long long* p = dbg_ptr();
        MQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cs_dbg), &p, sizeof(p)));
    }
#endif
    {
        const int* layout_word = g.nqt == 1 ? (const int*)(ovf + 1) : (const int*)nullptr;
        const unsigned* gthr = (const unsigned*)(ws + g.off_gthr);
        const unsigned* smax = (const unsigned*)(ws + g.off_smax);
        if (small_scan_serves(g, N, d, metric, k) && knn_opt(MQ_KNN_OPT_SMALL_WAVES) == 8)  // (checked against the word the scan left)
            hipLaunchKernelGGL(cand_select_kernel<CSEL_WALK_HALVES>, dim3((unsigned)nq), dim3(256), 0, st, pools, pcount, margin, gthr,
                               smax, g.ms, ovf, nq, g.S, k, cand, ccount, layout_word);
        else if (g.S >= 64)
            hipLaunchKernelGGL(cand_select_kernel<CSEL_WALK_SLICE_LINES>, dim3((unsigned)nq), dim3(256), 0, st, pools, pcount, margin, gthr,
                               smax, g.ms, ovf, nq, g.S, k, cand, ccount, layout_word);
        else
            hipLaunchKernelGGL(cand_select_kernel<CSEL_WALK_SLICES>, dim3((unsigned)nq), dim3(256), 0, st, pools, pcount, margin, gthr,
                               smax, g.ms, ovf, nq, g.S, k, cand, ccount, layout_word);
    }
    MQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)nq, RMAX / 64), dim3(256), 0, st, rowmajor_dev, q_rm, d, cand, ccount, ckeys,
                       l2 ? (const float*)qn : (const float*)nullptr, sqnorm_dev, flip);
    MQ_HIP(hipGetLastError());
    if (k <= KF)
        hipLaunchKernelGGL(final_select_kernel<128>, dim3((unsigned)nq), dim3(64), 0, st, ckeys, ccount, ovf, k, (long long)id_offset,
                           D_dev, (long long*)I_dev, l2, flip);
    else
        hipLaunchKernelGGL(final_select_kernel<FINAL_SORT_KEYS>, dim3((unsigned)nq), dim3(64), 0, st, ckeys, ccount, ovf, k,
                           (long long)id_offset, D_dev, (long long*)I_dev, l2, flip);
    MQ_HIP(hipGetLastError());
    // 5. query tiles whose bounded buffers overflowed are recomputed by the exact scan (no-op otherwise:
    //    every workgroup of an unflagged tile returns at once)
    {
        if (!l2) {  // (panels of flagged tiles only, their padding rows included: no clearing launch in front)
            hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)(g.nqpad / PANEL)), dim3(256), 0, st, q_rm, (int64_t)nq, d,
                               g.dpad, (int64_t)0, 0, Qp, qn, (const int*)ovf, (int64_t)g.nqpad);
            MQ_HIP(hipGetLastError());
        }
        ScanArgs a;
        a.Xp = packed_dev ? packed_dev : rowmajor_dev; a.Qp = Qp; a.xn = sqnorm_dev; a.qn = qn; a.lists = pools; a.d = d;
        a.N = N; a.dpad = g.dpad; a.nqt = g.nqt; a.S = g.S; a.k = k; a.nchunks = g.nchunks; a.qpx = g.qpx; a.dbg = nullptr; a.only = ovf;
        a.flip = flip; a.ceil = nullptr;
        const int rc = exact_scan_rounds(tm, packed_dev == nullptr, a, g, nq, k, id_offset, D_dev, I_dev, ws, ovf, st, nullptr, nullptr);
        if (rc != MQ_OK) return rc;
    }
    return MQ_OK;
}

/* Debug/telemetry of the last screened search held in `ws_dev`: out[0] = flagged query tiles,
 * out[1] = total candidates re-scored, out[2] = max candidates of one query.  Synchronises the stream. */
int mq_knn_screen_stats(int64_t N, int d, int nq, int k, const void* ws_dev, int64_t out[8], void* stream) {
    if (!ws_dev || !out || nq <= 0) return MQ_EINVAL;
    const Geometry g = geometry(N, d, nq, k, num_cus());  // the offsets read here do not depend on the metric
    MQ_HIP(hipStreamSynchronize((hipStream_t)stream));
    const size_t npc = (size_t)g.nqt * g.S * TQ * NSL;
    int* ovf = (int*)malloc((size_t)g.nqt * 4);
    int* cc = (int*)malloc((size_t)nq * 4);
    int* pc = (int*)malloc(npc * 4);
    float* mg = (float*)malloc((size_t)nq * 4);
    if (!ovf || !cc || !pc || !mg) { free(ovf); free(cc); free(pc); free(mg); return MQ_EINVAL; }
    hipError_t e1 = hipMemcpy(ovf, (const char*)ws_dev + g.off_ovf, (size_t)g.nqt * 4, hipMemcpyDeviceToHost);
    hipError_t e2 = hipMemcpy(cc, (const char*)ws_dev + g.off_ccount, (size_t)nq * 4, hipMemcpyDeviceToHost);
    hipError_t e3 = hipMemcpy(pc, (const char*)ws_dev + g.off_pcount, npc * 4, hipMemcpyDeviceToHost);
    hipError_t e4 = hipMemcpy(mg, (const char*)ws_dev + g.off_margin, (size_t)nq * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int i = 0; i < g.nqt; ++i) out[0] += ovf[i] != 0;
    for (int i = 0; i < nq; ++i) { out[1] += cc[i]; if (cc[i] > out[2]) out[2] = cc[i]; }
    for (size_t i = 0; i < npc; i += NSL) {  // one pool = NSL slices
        int64_t n = 0;
        for (int z = 0; z < NSL; ++z) n += pc[i + z];
        out[4] += n;
        if (n > out[3]) out[3] = n;
    }
    float mm = 0.f;
    for (int i = 0; i < nq; ++i) if (mg[i] > mm || mg[i] != mg[i]) mm = mg[i];
    out[5] = (int64_t)(mm * 1e6f);  /* max margin in 1e-6 units */
    out[6] = g.S;
    free(ovf); free(cc); free(pc); free(mg);
    if (e1 != hipSuccess) return hip_fail(e1);
    if (e2 != hipSuccess) return hip_fail(e2);
    if (e3 != hipSuccess) return hip_fail(e3);
    if (e4 != hipSuccess) return hip_fail(e4);
    return MQ_OK;
}

static int topk_merge_impl(const float* Ds_dev, const int64_t* Is_dev, size_t d_stride, size_t i_stride, int nshards, int nq,
                           int k, int metric_and_tie, float* D_dev, int64_t* I_dev, void* stream) {
    if (nq == 0) return MQ_OK;
    if (!Ds_dev || !Is_dev || !D_dev || !I_dev || nshards <= 0 || nq < 0 || k <= 0) return MQ_EINVAL;
    const bool desc = (metric_and_tie & MQ_MERGE_TIE_ID_DESC) != 0;
    const int metric = metric_and_tie & ~MQ_MERGE_TIE_ID_DESC;
    if (metric != MQ_METRIC_IP && metric != MQ_METRIC_L2) return MQ_EINVAL;
    if (k > MQ_KNN_MAX_K) return MQ_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (k > KF) {
        if (metric == MQ_METRIC_IP)
            hipLaunchKernelGGL(shard_merge_big_kernel<MQ_METRIC_IP>, dim3((unsigned)nq), dim3(256), 0, st, Ds_dev, (const long long*)Is_dev,
                               d_stride, i_stride, nshards, nq, k, D_dev, (long long*)I_dev, desc);
        else
            hipLaunchKernelGGL(shard_merge_big_kernel<MQ_METRIC_L2>, dim3((unsigned)nq), dim3(256), 0, st, Ds_dev, (const long long*)Is_dev,
                               d_stride, i_stride, nshards, nq, k, D_dev, (long long*)I_dev, desc);
    } else if (metric == MQ_METRIC_IP)
        hipLaunchKernelGGL(shard_merge_kernel<MQ_METRIC_IP>, dim3((unsigned)nq), dim3(128), 0, st, Ds_dev,
                           (const long long*)Is_dev, d_stride, i_stride, nshards, nq, k, D_dev, (long long*)I_dev, desc);
    else
        hipLaunchKernelGGL(shard_merge_kernel<MQ_METRIC_L2>, dim3((unsigned)nq), dim3(128), 0, st, Ds_dev,
                           (const long long*)Is_dev, d_stride, i_stride, nshards, nq, k, D_dev, (long long*)I_dev, desc);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_topk_merge_f32(const float* Ds_dev, const int64_t* Is_dev, int nshards, int nq, int k, int metric, float* D_dev,
                      int64_t* I_dev, void* stream) {
    const size_t n = (size_t)(nq > 0 ? nq : 0) * (size_t)(k > 0 ? k : 0);
    return topk_merge_impl(Ds_dev, Is_dev, n, n, nshards, nq, k, metric, D_dev, I_dev, stream);
}

size_t mq_shard_record_bytes(int nq, int k) {
    if (nq < 0 || k <= 0) return 0;
    return (size_t)round_up((int64_t)nq * k * 12, 16);
}

size_t mq_shard_record_ids_offset(int nq, int k) {
    if (nq < 0 || k <= 0) return 0;
    return (size_t)round_up((int64_t)nq * k * 4, 8);
}

int mq_topk_merge_records_f32(const void* records_dev, int nshards, int nq, int k, int metric, float* D_dev, int64_t* I_dev,
                              void* stream) {
    if (nq == 0) return MQ_OK;
    if (!records_dev || nq < 0 || k <= 0) return MQ_EINVAL;
    if (((uintptr_t)records_dev & 7) != 0) return MQ_EINVAL;
    const size_t rec = mq_shard_record_bytes(nq, k), ids = mq_shard_record_ids_offset(nq, k);
    return topk_merge_impl((const float*)records_dev, (const int64_t*)((const char*)records_dev + ids), rec / 4, rec / 8, nshards,
                           nq, k, metric, D_dev, I_dev, stream);
}

}  // extern "C"
