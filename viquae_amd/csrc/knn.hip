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
#include <math.h>
#include <stdlib.h>

#include "../../include/meerqat_hip.h"

typedef unsigned long long u64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Code placement of the MFMA loop moves kernel time by +-4.5 % between 4-byte phases (measured,
// profiles/r01_notes.md); MQ_PAD_NOPS shifts it.  Re-tune with tools/tune_placement.sh after edits.
#ifndef MQ_PAD_NOPS
#define MQ_PAD_NOPS 3
#endif

namespace {

constexpr int PANEL = 64;   // KB rows per panel
constexpr int BK = 16;      // k-depth of one LDS stage
constexpr int TQ = 256;     // queries per workgroup tile
constexpr int TN = 256;     // KB rows per chunk
constexpr int NWAVES = 16;  // 4 (row panels) x 4 (query panels), one 64x64 sub-tile per wave
constexpr int CAND = 32;    // candidate slots per query between two list merges
constexpr int FLUSH_AT = 24;
constexpr int KCAP = MQ_KNN_MAX_K;  // list scratch per wave

// LDS carve (bytes)
constexpr int LDS_XS = 0;                              // [2][4][16][64] f32
constexpr int LDS_QS = LDS_XS + 2 * 4 * BK * 64 * 4;   // [2][4][16][64] f32
constexpr int LDS_CAND = LDS_QS + 2 * 4 * BK * 64 * 4; // [256][CAND] u64
constexpr int LDS_SCRL = LDS_CAND + TQ * CAND * 8;     // [16][KCAP] u64
constexpr int LDS_CNT = LDS_SCRL + NWAVES * KCAP * 8;  // cnt[256], cnt0[256], tau[256], qflag[256], wgflag[4]
constexpr int LDS_TOTAL = LDS_CNT + 4 * TQ * 4 + 16;

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
__device__ __forceinline__ unsigned key_row(u64 key) { return 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull); }

// ------------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------------

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
                                                        float* __restrict__ sqnorm) {
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * PANEL;  // within this call's rows
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
        if (t < 64) nrm[t] = sqrtf(acc);
        __syncthreads();
    }
    float acc2 = 0.f;
    for (int kc = 0; kc < dpad; kc += 64) {
        load_tile64(src, n, d, row0, kc, tile);
        __syncthreads();
        if (l2norm) {
            // x / ||x||: same two roundings as the oracle (sqrtf, then one division per element)
            for (int e = t; e < 64 * 64; e += 256) {
                const int r = e >> 6, k = e & 63;
                if (row0 + r < n && kc + k < d) tile[r][k] = tile[r][k] / nrm[r];
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
    if (t < 64 && row0 + t < n) sqnorm[row_offset + row0 + t] = acc2;
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

__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ rows, int64_t n, int d) {
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
    if (t < 64) nrm[t] = sqrtf(acc);
    __syncthreads();
    for (int64_t e = t; e < 64 * (int64_t)d; e += 256) {
        const int r = (int)(e / d);
        const int k = (int)(e - (int64_t)r * d);
        if (row0 + r < n) {
            float* p = rows + (row0 + r) * (int64_t)d + k;
            *p = *p / nrm[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the scan kernel
// ------------------------------------------------------------------------------------------------
struct ScanArgs {
    const float* Xp;   // KB, panel layout
    const float* Qp;   // queries, panel layout (nqt * 4 panels)
    const float* xn;   // ||x||^2 per KB row (L2 only)
    const float* qn;   // ||q||^2 per query   (L2 only)
    u64* lists;        // [nqt][S][256][k] sorted keys
    long long N;
    int dpad, nqt, S, k, qpx;
    long long nchunks;
};

// One wave merges the first n candidate keys of query q into its sorted list (global memory,
// owned by this workgroup) and refreshes the query's threshold.
__device__ __forceinline__ void flush_query(u64* __restrict__ Lg, const u64* __restrict__ B, int n, int k,
                                            u64* __restrict__ sL, float* tau_q, int lane) {
    const u64 l0 = (lane < k) ? Lg[lane] : 0ull;
    const u64 l1 = (lane + 64 < k) ? Lg[lane + 64] : 0ull;
    sL[lane] = l0;
    sL[lane + 64] = l1;
    const u64 mb = (lane < n) ? B[lane] : 0ull;
    int cB = 0, c0 = 0, c1 = 0;
    for (int j = 0; j < n; ++j) {
        const u64 bj = B[j];
        cB += (bj > mb);
        c0 += (bj > l0);
        c1 += (bj > l1);
    }
    int lo = 0, hi = k;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sL[mid] > mb) lo = mid + 1; else hi = mid;
    }
    const int pB = cB + lo;
    const int p0 = lane + c0;
    const int p1 = lane + 64 + c1;
    if (mb != 0ull && pB < k) { Lg[pB] = mb; if (pB == k - 1) *tau_q = key_score(mb); }
    if (l0 != 0ull && p0 < k) { Lg[p0] = l0; if (p0 == k - 1) *tau_q = key_score(l0); }
    if (l1 != 0ull && p1 < k) { Lg[p1] = l1; if (p1 == k - 1) *tau_q = key_score(l1); }
}

template <int METRIC>
__global__ __launch_bounds__(1024) void knn_scan_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xs = reinterpret_cast<float*>(smem + LDS_XS);
    float* Qs = reinterpret_cast<float*>(smem + LDS_QS);
    u64* cand = reinterpret_cast<u64*>(smem + LDS_CAND);
    u64* scrL = reinterpret_cast<u64*>(smem + LDS_SCRL);
    int* cnt = reinterpret_cast<int*>(smem + LDS_CNT);
    int* cnt0 = cnt + TQ;
    float* tau = reinterpret_cast<float*>(cnt0 + TQ);
    int* qflag = reinterpret_cast<int*>(tau + TQ);
    int* wgflag = qflag + TQ;

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
    const long long c0 = (a.nchunks * slab) / a.S;
    const long long c1 = (a.nchunks * (slab + 1)) / a.S;
    const int k = a.k;
    u64* mylists = a.lists + ((size_t)qt * a.S + slab) * (size_t)TQ * k;

    for (int i = tid; i < TQ * k; i += 1024) mylists[i] = 0ull;
    if (tid < TQ) {
        cnt[tid] = 0;
        cnt0[tid] = 0;
        tau[tid] = -INFINITY;
        qflag[tid] = 0;
    }
    if (tid == 0) wgflag[0] = 0;

    const int nkb = a.dpad / BK;
    const int pp = w >> 2, quarter = w & 3;
    // per-lane DMA sources; LDS destination is wave-uniform base + lane*16 (hardware rule)
    const float* qsrc0 = a.Qp + (((size_t)(qt * 4 + pp) * a.dpad + quarter * 4) * PANEL) + lane * 4;
    const int lds_piece = pp * (BK * 64) + quarter * 256;

    // LDS-DMA (global_load_lds_dwordx4) issued from inline asm so that hipcc does not count it:
    // with the builtin it drains vmcnt(0) in front of the next ds_read and the prefetch of step t+1
    // would serialise with the MFMAs of step t.  Completion is awaited by hand (vmcnt(0) right
    // before the barrier that opens the step which reads the data).
    const unsigned lds_x = __builtin_amdgcn_readfirstlane(lds_addr(Xs + lds_piece));
    const unsigned lds_q = __builtin_amdgcn_readfirstlane(lds_addr(Qs + lds_piece));
    auto issue = [&](long long c, int kb, int stage) __attribute__((always_inline)) {
        const float* xsrc = a.Xp + (((size_t)(c * 4 + pp) * a.dpad + (size_t)kb * BK + quarter * 4) * PANEL) + lane * 4;
        const float* qsrc = qsrc0 + (size_t)kb * BK * PANEL;
        dma16(xsrc, lds_x + stage * (4096 * 4));
        dma16(qsrc, lds_q + stage * (4096 * 4));
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
    if (c0 < c1) issue(c0, 0, 0);
    int stage = 0;

    for (long long c = c0; c < c1; ++c) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        for (int kb = 0; kb < nkb; ++kb) {
            // the DMA of this step has landed (vmcnt(0) precedes the barrier) and every wave is
            // done reading the other stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kb + 1 < nkb) issue(c, kb + 1, stage ^ 1);
            else if (c + 1 < c1) issue(c + 1, 0, stage ^ 1);
            const float* xs = Xs + stage * 4096 + wr * (BK * 64) + kh * 64 + i2;
            const float* qs = Qs + stage * 4096 + wc * (BK * 64) + kh * 64 + i2;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float2 xa = *reinterpret_cast<const float2*>(xs + kk * 128);
                const float2 qb = *reinterpret_cast<const float2*>(qs + kk * 128);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, qb.x, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, qb.y, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, qb.x, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, qb.y, acc11, 0, 0, 0);
            }
            stage ^= 1;
        }

        // ---------------- selection epilogue ----------------
        // C/D map of 32x32x2: column j = lane&31 (query 2j+b of the wave's panel),
        // row i' = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  (KB row 2i'+a of the wave's panel)
        const long long rowbase = c * TN + wr * 64 + 8 * kh;
        const bool ragged = (c + 1) * (long long)TN > a.N;

        if (METRIC == MQ_METRIC_L2) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long r0 = rowbase + 2 * ((reg & 3) + 8 * (reg >> 2));
                const float2 xn2 = *reinterpret_cast<const float2*>(a.xn + r0);  // rows r0, r0+1 (padded alloc)
                float d;
                d = (qn0 + xn2.x) - 2.0f * acc00[reg]; acc00[reg] = -(d < 0.f ? 0.f : d);
                d = (qn1 + xn2.x) - 2.0f * acc01[reg]; acc01[reg] = -(d < 0.f ? 0.f : d);
                d = (qn0 + xn2.y) - 2.0f * acc10[reg]; acc10[reg] = -(d < 0.f ? 0.f : d);
                d = (qn1 + xn2.y) - 2.0f * acc11[reg]; acc11[reg] = -(d < 0.f ? 0.f : d);
            }
        }

        // sub = -1: all registers, all queries; sub >= 0: only registers with (reg>>1)==sub and only
        // queries being re-done after an overflow
        auto scan_acc = [&](int sub) __attribute__((always_inline)) {
            const float t0 = tau[q0], t1 = tau[q1];
            const bool do0 = sub < 0 || qflag[q0] == 2;
            const bool do1 = sub < 0 || qflag[q1] == 2;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (sub >= 0 && (reg >> 1) != sub) continue;
                const long long r0 = rowbase + 2 * ((reg & 3) + 8 * (reg >> 2));
                const bool v0 = !ragged || r0 < a.N;
                const bool v1 = !ragged || r0 + 1 < a.N;
                const float g00 = acc00[reg], g01 = acc01[reg], g10 = acc10[reg], g11 = acc11[reg];
                const bool p00 = do0 && v0 && g00 > t0;
                const bool p01 = do1 && v0 && g01 > t1;
                const bool p10 = do0 && v1 && g10 > t0;
                const bool p11 = do1 && v1 && g11 > t1;
                if (__builtin_amdgcn_ballot_w64(p00 || p01 || p10 || p11) == 0ull) continue;
                if (p00) { const int s = atomicAdd(&cnt[q0], 1); if (s < CAND) cand[q0 * CAND + s] = make_key(g00, (unsigned)r0); }
                if (p01) { const int s = atomicAdd(&cnt[q1], 1); if (s < CAND) cand[q1 * CAND + s] = make_key(g01, (unsigned)r0); }
                if (p10) { const int s = atomicAdd(&cnt[q0], 1); if (s < CAND) cand[q0 * CAND + s] = make_key(g10, (unsigned)(r0 + 1)); }
                if (p11) { const int s = atomicAdd(&cnt[q1], 1); if (s < CAND) cand[q1 * CAND + s] = make_key(g11, (unsigned)(r0 + 1)); }
            }
        };

        scan_acc(-1);
        __syncthreads();
        if (tid < TQ) {
            const int cq = cnt[tid];
            int f = 0;
            if (cq > CAND) f = 2;                                    // overflow: roll this chunk back, merge, re-do
            else if (cq >= FLUSH_AT || (c + 1 == c1 && cq > 0)) f = 1;  // nearly full (or end of slab): merge
            qflag[tid] = f;
            if (f) atomicOr(&wgflag[0], f);
            else cnt0[tid] = cq;
        }
        __syncthreads();
        const int wf = wgflag[0];
        if (wf) {
            for (int j = 0; j < TQ / NWAVES; ++j) {
                const int q = w + NWAVES * j;
                const int f = __builtin_amdgcn_readfirstlane(qflag[q]);
                if (f) {
                    const int n = __builtin_amdgcn_readfirstlane((f == 2) ? cnt0[q] : cnt[q]);
                    flush_query(mylists + (size_t)q * k, cand + q * CAND, n, k, scrL + w * KCAP, &tau[q], lane);
                    if (lane == 0) { cnt[q] = 0; cnt0[q] = 0; }
                }
            }
            __syncthreads();
            if (wf & 2) {
                for (int sub = 0; sub < 8; ++sub) {
                    scan_acc(sub);  // <= 2 regs x 2 rows x 2 halves x 4 row panels = 32 = CAND appends per query
                    __syncthreads();
                    for (int j = 0; j < TQ / NWAVES; ++j) {
                        const int q = w + NWAVES * j;
                        const int n = __builtin_amdgcn_readfirstlane(qflag[q] == 2 ? cnt[q] : 0);
                        if (n > 0) {
                            flush_query(mylists + (size_t)q * k, cand + q * CAND, n, k, scrL + w * KCAP, &tau[q], lane);
                            if (lane == 0) cnt[q] = 0;
                        }
                    }
                    __syncthreads();
                }
            }
            if (tid == 0) wgflag[0] = 0;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merges
// ------------------------------------------------------------------------------------------------
struct Ent {
    float g;
    long long id;  // < 0: empty
};
__device__ __forceinline__ bool better(const Ent& x, const Ent& y) {
    if (y.id < 0) return x.id >= 0;
    if (x.id < 0) return false;
    return x.g > y.g || (x.g == y.g && x.id < y.id);
}

// count of entries in sorted (best first) list L[0..k) that are better than e
__device__ __forceinline__ int count_better(const Ent* L, int k, const Ent& e) {
    int lo = 0, hi = k;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (better(L[mid], e)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Sequentially rank-merges nlists sorted lists of k entries; entry fetch is a functor so the same
// body serves the slab lists (packed keys) and the shard lists (D, I arrays).  128 threads, k <= 128.
template <typename Fetch>
__device__ __forceinline__ void merge_lists(int nlists, int k, Fetch fetch, Ent* Ra, Ent* Rb, Ent* Ls, Ent& out) {
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
            const int p = t + count_better(Ls, k, r);
            if (p < k) nxt[p] = r;
        }
        if (l.id >= 0) {
            const int p = t + count_better(cur, k, l);
            if (p < k) nxt[p] = l;
        }
        __syncthreads();
        Ent* tmp = cur; cur = nxt; nxt = tmp;
    }
    out = cur[t];
}

template <int METRIC>
__global__ __launch_bounds__(128) void slab_merge_kernel(const u64* __restrict__ lists, int nq, int S, int k,
                                                         long long id_offset, float* __restrict__ D,
                                                         long long* __restrict__ I) {
    __shared__ Ent Ra[128], Rb[128], Ls[128];
    const int q = blockIdx.x;
    const int qt = q / TQ, ql = q % TQ;
    auto fetch = [&](int s, int t) {
        const u64 key = lists[(((size_t)qt * S + s) * TQ + ql) * (size_t)k + t];
        Ent e;
        e.g = key_score(key);
        e.id = key ? (long long)key_row(key) : -1;
        return e;
    };
    Ent out;
    merge_lists(S, k, fetch, Ra, Rb, Ls, out);
    const int t = threadIdx.x;
    if (t < k) {
        float d;
        if (out.id < 0) d = (METRIC == MQ_METRIC_L2) ? INFINITY : -INFINITY;
        else d = ((METRIC == MQ_METRIC_L2) ? -out.g : out.g) + 0.0f;
        D[(size_t)q * k + t] = d;
        I[(size_t)q * k + t] = out.id < 0 ? -1 : out.id + id_offset;
    }
}

template <int METRIC>
__global__ __launch_bounds__(128) void shard_merge_kernel(const float* __restrict__ Ds, const long long* __restrict__ Is,
                                                          int nshards, int nq, int k, float* __restrict__ D,
                                                          long long* __restrict__ I) {
    __shared__ Ent Ra[128], Rb[128], Ls[128];
    const int q = blockIdx.x;
    auto fetch = [&](int s, int t) {
        const size_t o = ((size_t)s * nq + q) * (size_t)k + t;
        Ent e;
        e.id = Is[o];
        const float d = Ds[o];
        e.g = (METRIC == MQ_METRIC_L2) ? -d : d;
        return e;
    };
    Ent out;
    merge_lists(nshards, k, fetch, Ra, Rb, Ls, out);
    const int t = threadIdx.x;
    if (t < k) {
        float d;
        if (out.id < 0) d = (METRIC == MQ_METRIC_L2) ? INFINITY : -INFINITY;
        else d = ((METRIC == MQ_METRIC_L2) ? -out.g : out.g) + 0.0f;
        D[(size_t)q * k + t] = d;
        I[(size_t)q * k + t] = out.id < 0 ? -1 : out.id;
    }
}

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
    int nqt, S, dpad, qpx;
    int64_t nqpad, nchunks;
    size_t off_qp, off_qn, off_qtmp, off_lists, total;
};

Geometry geometry(int64_t N, int d, int nq, int k, int cus) {
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
    {
        int want = 0;
        const char* e = getenv("MQ_KNN_QPX");
        if (e) want = atoi(e);
        for (int q = 1; q <= g.nqt; ++q) {
            if (g.nqt % q) continue;
            const int ngroups = g.nqt / q;
            if (ngroups > 8 || 8 % ngroups) continue;
            const int xpg = 8 / ngroups;
            if (g.S % xpg) continue;
            if (want > 0) { if (q == want) g.qpx = q; continue; }
            // default: the largest Q working set that still leaves half of L2 to the KB stream
            if ((size_t)q * TQ * g.dpad * 4 <= (size_t)2 << 20 || g.qpx == 0) g.qpx = q;
        }
    }
    size_t o = 0;
    g.off_qp = o;    o += (size_t)g.nqpad * g.dpad * 4;
    g.off_qn = o;    o += (size_t)g.nqpad * 4;
    g.off_qtmp = o;  o += (size_t)round_up((int64_t)(nq > 0 ? nq : 1) * d * 4, 256);
    g.off_lists = o; o += (size_t)g.nqt * g.S * TQ * (size_t)k * 8;
    g.total = round_up((int64_t)o, 256);
    return g;
}

}  // namespace

extern "C" {

const char* mq_version(void) { return "meerqat_hip 0.1 (gfx950)"; }

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

int mq_l2norm_rows_f32(float* rows_dev, int64_t n, int d, void* stream) {
    if (n == 0) return MQ_OK;
    if (!rows_dev || n < 0 || d <= 0) return MQ_EINVAL;
    const unsigned grid = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows_dev, n, d);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

size_t mq_knn_workspace_bytes(int64_t N, int d, int nq, int k) {
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || k > MQ_KNN_MAX_K) return 0;
    return geometry(N, d, nq, k, num_cus()).total;
}

int mq_knn_launch_info(int64_t N, int d, int nq, int k, int64_t out[6]) {
    if (!out || N < 0 || d <= 0 || nq < 0 || k <= 0 || k > MQ_KNN_MAX_K) return MQ_EINVAL;
    const Geometry g = geometry(N, d, nq, k, num_cus());
    out[0] = (int64_t)g.nqt * g.S;
    out[1] = 1024;
    out[2] = LDS_TOTAL;
    out[3] = g.nqt;
    out[4] = g.S;
    out[5] = g.nchunks;
    return MQ_OK;
}

static int knn_search_impl(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                           int nq, int k, int metric, int l2norm_queries, int64_t id_offset, float* D_dev, int64_t* I_dev,
                           void* ws_dev, size_t ws_bytes, void* stream, void* ev_scan_begin, void* ev_scan_end) {
    if (nq == 0) return MQ_OK;
    if (!packed_dev || !sqnorm_dev || !queries_dev || !D_dev || !I_dev || !ws_dev) return MQ_EINVAL;
    if (N < 0 || d <= 0 || nq < 0 || k <= 0 || (metric != MQ_METRIC_IP && metric != MQ_METRIC_L2)) return MQ_EINVAL;
    if (k > MQ_KNN_MAX_K) return MQ_EUNSUPPORTED;
    if (N >= 0xFFFFFFFFll) return MQ_EUNSUPPORTED;  // 32-bit local row ids
    const Geometry g = geometry(N, d, nq, k, num_cus());
    if (ws_bytes < g.total) return MQ_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)ws_dev;
    float* Qp = (float*)(ws + g.off_qp);
    float* qn = (float*)(ws + g.off_qn);
    float* qtmp = (float*)(ws + g.off_qtmp);
    u64* lists = (u64*)(ws + g.off_lists);

    // queries -> panel layout (+ optional "L2norm," transform, + ||q||^2); padded queries are zero
    MQ_HIP(hipMemsetAsync(Qp, 0, (size_t)g.nqpad * g.dpad * 4 + (size_t)g.nqpad * 4, st));
    const float* qsrc = queries_dev;
    (void)qtmp;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((nq + PANEL - 1) / PANEL)), dim3(256), 0, st, qsrc, (int64_t)nq, d,
                       g.dpad, (int64_t)0, l2norm_queries, Qp, qn);
    MQ_HIP(hipGetLastError());

    if (N > 0) {
        ScanArgs a;
        a.Xp = packed_dev; a.Qp = Qp; a.xn = sqnorm_dev; a.qn = qn; a.lists = lists;
        a.N = N; a.dpad = g.dpad; a.nqt = g.nqt; a.S = g.S; a.k = k; a.nchunks = g.nchunks; a.qpx = g.qpx;
        const dim3 grid((unsigned)(g.nqt * g.S)), block(1024);
        if (ev_scan_begin) MQ_HIP(hipEventRecord((hipEvent_t)ev_scan_begin, st));
        if (metric == MQ_METRIC_IP) {
            MQ_HIP(hipFuncSetAttribute((const void*)knn_scan_kernel<MQ_METRIC_IP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
            hipLaunchKernelGGL(knn_scan_kernel<MQ_METRIC_IP>, grid, block, LDS_TOTAL, st, a);
        } else {
            MQ_HIP(hipFuncSetAttribute((const void*)knn_scan_kernel<MQ_METRIC_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
            hipLaunchKernelGGL(knn_scan_kernel<MQ_METRIC_L2>, grid, block, LDS_TOTAL, st, a);
        }
        MQ_HIP(hipGetLastError());
        if (ev_scan_end) MQ_HIP(hipEventRecord((hipEvent_t)ev_scan_end, st));
    } else {
        MQ_HIP(hipMemsetAsync(lists, 0, (size_t)g.nqt * g.S * TQ * (size_t)k * 8, st));
    }
    if (metric == MQ_METRIC_IP)
        hipLaunchKernelGGL(slab_merge_kernel<MQ_METRIC_IP>, dim3((unsigned)nq), dim3(128), 0, st, lists, nq, g.S, k,
                           (long long)id_offset, D_dev, (long long*)I_dev);
    else
        hipLaunchKernelGGL(slab_merge_kernel<MQ_METRIC_L2>, dim3((unsigned)nq), dim3(128), 0, st, lists, nq, g.S, k,
                           (long long)id_offset, D_dev, (long long*)I_dev);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_knn_search_f32(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                      int nq, int k, int metric, int l2norm_queries, int64_t id_offset, float* D_dev, int64_t* I_dev,
                      void* ws_dev, size_t ws_bytes, void* stream) {
    return knn_search_impl(packed_dev, sqnorm_dev, N, d, queries_dev, nq, k, metric, l2norm_queries, id_offset, D_dev,
                           I_dev, ws_dev, ws_bytes, stream, nullptr, nullptr);
}

int mq_knn_search_f32_ev(const float* packed_dev, const float* sqnorm_dev, int64_t N, int d, const float* queries_dev,
                         int nq, int k, int metric, int l2norm_queries, int64_t id_offset, float* D_dev, int64_t* I_dev,
                         void* ws_dev, size_t ws_bytes, void* stream, void* ev_scan_begin, void* ev_scan_end) {
    return knn_search_impl(packed_dev, sqnorm_dev, N, d, queries_dev, nq, k, metric, l2norm_queries, id_offset, D_dev,
                           I_dev, ws_dev, ws_bytes, stream, ev_scan_begin, ev_scan_end);
}

int mq_topk_merge_f32(const float* Ds_dev, const int64_t* Is_dev, int nshards, int nq, int k, int metric, float* D_dev,
                      int64_t* I_dev, void* stream) {
    if (nq == 0) return MQ_OK;
    if (!Ds_dev || !Is_dev || !D_dev || !I_dev || nshards <= 0 || nq < 0 || k <= 0) return MQ_EINVAL;
    if (metric != MQ_METRIC_IP && metric != MQ_METRIC_L2) return MQ_EINVAL;
    if (k > MQ_KNN_MAX_K) return MQ_EUNSUPPORTED;
    if (metric == MQ_METRIC_IP)
        hipLaunchKernelGGL(shard_merge_kernel<MQ_METRIC_IP>, dim3((unsigned)nq), dim3(128), 0, (hipStream_t)stream, Ds_dev,
                           (const long long*)Is_dev, nshards, nq, k, D_dev, (long long*)I_dev);
    else
        hipLaunchKernelGGL(shard_merge_kernel<MQ_METRIC_L2>, dim3((unsigned)nq), dim3(128), 0, (hipStream_t)stream, Ds_dev,
                           (const long long*)Is_dev, nshards, nq, k, D_dev, (long long*)I_dev);
    MQ_HIP(hipGetLastError());
    return MQ_OK;
}

}  // extern "C"
