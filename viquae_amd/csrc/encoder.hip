// encoder.hip -- fp32 transformer-encoder building blocks for gfx950 (MI355X): the arithmetic the
// reference delegates to Hugging Face DPRContextEncoder / CLIPModel.get_image_features
// (meerqat/ir/embedding.py:226, meerqat/image/embedding.py:156-161; op order stated in-tree by
// meerqat/models/bert.py:12-380).  C ABI: include/meerqat_hip.h.
//
//   gemm_nt_kernel        C[M,N] = A[M,K] . W[N,K]^T (+bias)(+GELU|quick_gelu)(+residual) on
//                         v_mfma_f32_32x32x2_f32, 256x256x16 tiles, LDS-DMA double buffer
//   layernorm_kernel      row LayerNorm (one wave per row)
//   bert_embed_ln_kernel  word + token-type + position gather, LayerNorm
//   attention_mfma_kernel softmax(q k^T * scale + padding/causal mask) v per (sequence, head, 128 queries) on fp32 MFMA
//   clip_patchify_kernel  NCHW pixels -> [B*patches, C*P*P] rows (the conv-as-GEMM operand)
//   clip_assemble_ln_kernel  [CLS | patch embeddings] + position embeddings, pre-LayerNorm
//
// All activations are row-major fp32 [tokens, features]; weights keep the PyTorch nn.Linear layout
// [out_features, in_features] (K contiguous for both GEMM operands, no repacking).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "../../include/meerqat_hip.h"
#include "launch_attr.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

extern "C" void mq_internal_set_hip_error(int e);

namespace {

#define ENC_HIP(call)                                  \
    do {                                               \
        hipError_t _e = (call);                        \
        if (_e != hipSuccess) { mq_internal_set_hip_error((int)_e); return MQ_EHIP; } \
    } while (0)

typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long)(lds_char*)p; }
// one 1-KiB LDS-DMA piece: LDS destination = lds_dst (wave-uniform) + lane*16, global source per lane.
// Issued from inline asm so hipcc does not drain it in front of the next ds_read (see knn.hip).
// wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the per-step update is scalar
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------------------
// GEMM  C = A . W^T
// ------------------------------------------------------------------------------------------------
constexpr int GT = 256;   // tile edge (rows of A and rows of W per workgroup)
constexpr int GBK = 16;   // k-depth per LDS stage: one 64-byte row segment per tile row
constexpr int G_STAGE_FLOATS = GT * GBK;            // one operand, one stage
constexpr int G_NSTAGE = 3;                         // LDS ring depth
constexpr int G_LDS_BYTES = G_NSTAGE * 2 * G_STAGE_FLOATS * 4;  // A and W, three stages = 96 KiB

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_QUICKGELU = 3, EPI_BIAS_RESIDUAL = 4 };

// GELU (erf form) of the GEMM epilogues: 7.5e9 activations per bert-base forward of 2048 x 100 tokens, so the instruction count
// matters (the library erff costs ~40 VALU instructions per element with both of its branches executed).
//   gelu(x) = x Phi(x) = max(x, 0) - |x| erfc(|x| / sqrt 2) / 2,    erfc(z) = t P(t) exp(-z^2),  t = 1 / (1 + 0.39 z)
// with a degree-6 P fitted on z in [0, 4] (tools/fit_gelu_erfc.py: |erfc error| < 9e-9 in exact arithmetic, ~1e-7 in fp32;
// beyond z = 4 the product underflows like erfc itself): 15 full-rate + 2 quarter-rate (v_rcp_f32, v_exp_f32) instructions,
// no branch.  |gelu error| <= 1.2e-7 max(1, |x|); NaN in, NaN out (+-inf give NaN).
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39f, z, 1.0f));
    float p = fmaf(-0.2277139058741977f, t, 0.8873881791534648f);
    p = fmaf(p, t, -0.6365790927274033f);
    p = fmaf(p, t, 0.6505137809967253f);
    p = fmaf(p, t, 0.09100438095962154f);
    p = fmaf(p, t, 0.2353866503768924f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float half_erfc = (p * t) * (0.5f * e);
    return fmaxf(x, 0.0f) - fabsf(x) * half_erfc;
}
// x sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp each) instead of expf and an IEEE division
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f)));
}

// LDS image of a tile stage: [256 rows][16 floats], the four 16-byte chunks of a row XOR-swizzled by
// (row>>2)&3 so that ds_read_b128 of one k-chunk across 32 consecutive rows is bank-conflict free.
// The DMA writes LDS linearly (lane*16), so the swizzle is applied to the per-lane SOURCE chunk.
template <int EPI>
__global__ __launch_bounds__(1024) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ R,
                                                       float* __restrict__ C, int M, int N, int K, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Ws = As + G_NSTAGE * G_STAGE_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // tile of this workgroup: consecutive workgroups on one XCD (b % 8) walk the N tiles of one M tile
    int mt, nt;
    {
        const int b = blockIdx.x;
        if ((ntm & 7) == 0) {
            const int xcd = b & 7, j = b >> 3;
            mt = (j / ntn) * 8 + xcd;
            nt = j % ntn;
        } else {
            mt = b / ntn;
            nt = b % ntn;
        }
    }
    mt = __builtin_amdgcn_readfirstlane(mt);
    nt = __builtin_amdgcn_readfirstlane(nt);
    const int m0 = mt * GT, n0 = nt * GT;

    // DMA: wave w moves rows [16w, 16w+16) of both operand tiles; lane -> (row 16w + lane/4, chunk lane%4).
    // Source = wave-uniform tile base (advances 64 B per K step, scalar) + per-lane byte offset (constant).
    const int drow = 16 * w + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);
    int am = m0 + drow; if (am > M - 1) am = M - 1;
    int wn = n0 + drow; if (wn > N - 1) wn = N - 1;
    const unsigned a_voff = (unsigned)(((size_t)(am - m0) * K + dchunk * 4) * 4);
    const unsigned w_voff = (unsigned)(((size_t)(wn - n0) * K + dchunk * 4) * 4);
    const char* const abase = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const char* const wbase = reinterpret_cast<const char*>(W + (size_t)n0 * K);
    const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds_addr_of(As + 16 * w * GBK));
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds_addr_of(Ws + 16 * w * GBK));
    auto issue = [&](int kb, int stg) __attribute__((always_inline)) {
        dma16s(abase + (size_t)kb * (GBK * 4), a_voff, lds_a + stg * (G_STAGE_FLOATS * 4));
        dma16s(wbase + (size_t)kb * (GBK * 4), w_voff, lds_w + stg * (G_STAGE_FLOATS * 4));
    };

    // fragment reads: lane (i = lane&31, kh = lane>>5) reads chunk (2j+kh) of row (64*wr + 32a + i)
    const int i = lane & 31, kh = lane >> 5;
    const int sw = (i >> 2) & 3;
    const int c0 = ((0 + kh) ^ sw) * 4, c1 = ((2 + kh) ^ sw) * 4;  // float offsets of chunks j=0,1 inside the row
    const int arow = (64 * wr + i) * GBK, brow = (64 * wc + i) * GBK;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nk = K / GBK;
    // Three-stage ring, one raw barrier per K step in the MIDDLE of the step's MFMA block (see knn.hip)
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#define MQ_LD4(p) (*reinterpret_cast<const float4*>(p))
#define MQ_MFMA4(A0, A1, B0, B1, E)                                            \
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.E, B0.E, acc00, 0, 0, 0); \
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.E, B1.E, acc01, 0, 0, 0); \
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.E, B0.E, acc10, 0, 0, 0); \
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.E, B1.E, acc11, 0, 0, 0);
    float4 a0j0 = MQ_LD4(As + arow + c0), a1j0 = MQ_LD4(As + arow + 32 * GBK + c0);
    float4 b0j0 = MQ_LD4(Ws + brow + c0), b1j0 = MQ_LD4(Ws + brow + 32 * GBK + c0);
    int cur = 0;
    for (int kb = 0; kb < nk; ++kb) {
        const float* as = As + cur * G_STAGE_FLOATS + arow;
        const float* ws = Ws + cur * G_STAGE_FLOATS + brow;
        const float4 a0j1 = MQ_LD4(as + c1), a1j1 = MQ_LD4(as + 32 * GBK + c1);
        const float4 b0j1 = MQ_LD4(ws + c1), b1j1 = MQ_LD4(ws + 32 * GBK + c1);
        MQ_MFMA4(a0j0, a1j0, b0j0, b1j0, x)
        MQ_MFMA4(a0j0, a1j0, b0j0, b1j0, y)
        MQ_MFMA4(a0j0, a1j0, b0j0, b1j0, z)
        MQ_MFMA4(a0j0, a1j0, b0j0, b1j0, w)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kb + 2 < nk) issue(kb + 2, cur >= 1 ? cur - 1 : 2);
        cur = cur == 2 ? 0 : cur + 1;
        const float* an = As + cur * G_STAGE_FLOATS + arow;
        const float* wn_ = Ws + cur * G_STAGE_FLOATS + brow;
        a0j0 = MQ_LD4(an + c0); a1j0 = MQ_LD4(an + 32 * GBK + c0);
        b0j0 = MQ_LD4(wn_ + c0); b1j0 = MQ_LD4(wn_ + 32 * GBK + c0);
        MQ_MFMA4(a0j1, a1j1, b0j1, b1j1, x)
        MQ_MFMA4(a0j1, a1j1, b0j1, b1j1, y)
        MQ_MFMA4(a0j1, a1j1, b0j1, b1j1, z)
        MQ_MFMA4(a0j1, a1j1, b0j1, b1j1, w)
    }
#undef MQ_MFMA4
#undef MQ_LD4

    // epilogue: C/D map of 32x32x2: col j = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int nb0 = n0 + 64 * wc + i, nb1 = nb0 + 32;
    float bias0 = 0.f, bias1 = 0.f;
    if (EPI != EPI_NONE) {
        if (nb0 < N) bias0 = bias[nb0];
        if (nb1 < N) bias1 = bias[nb1];
    }
    const int mbase = m0 + 64 * wr + 4 * kh;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int mr0 = mbase + (reg & 3) + 8 * (reg >> 2), mr1 = mr0 + 32;
        float v00 = acc00[reg] + bias0, v01 = acc01[reg] + bias1, v10 = acc10[reg] + bias0, v11 = acc11[reg] + bias1;
        if (EPI == EPI_BIAS_GELU) { v00 = gelu_erf(v00); v01 = gelu_erf(v01); v10 = gelu_erf(v10); v11 = gelu_erf(v11); }
        if (EPI == EPI_BIAS_QUICKGELU) { v00 = quick_gelu(v00); v01 = quick_gelu(v01); v10 = quick_gelu(v10); v11 = quick_gelu(v11); }
        if (mr0 < M) {
            if (nb0 < N) { if (EPI == EPI_BIAS_RESIDUAL) v00 += R[(size_t)mr0 * N + nb0]; C[(size_t)mr0 * N + nb0] = v00; }
            if (nb1 < N) { if (EPI == EPI_BIAS_RESIDUAL) v01 += R[(size_t)mr0 * N + nb1]; C[(size_t)mr0 * N + nb1] = v01; }
        }
        if (mr1 < M) {
            if (nb0 < N) { if (EPI == EPI_BIAS_RESIDUAL) v10 += R[(size_t)mr1 * N + nb0]; C[(size_t)mr1 * N + nb0] = v10; }
            if (nb1 < N) { if (EPI == EPI_BIAS_RESIDUAL) v11 += R[(size_t)mr1 * N + nb1]; C[(size_t)mr1 * N + nb1] = v11; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 GEMM: fp32-class accuracy on the bf16 matrix pipe.
//   A = Ah + Al (+ r, |r| <= 2^-18 |A|), W = Wh + Wl (bf16 each);  C ~= Ah.Wh + Al.Wh + Ah.Wl  (fp32 accumulate)
// drops only Al.Wl and the r terms: relative error ~ 3 * 2^-18 per product, i.e. ~1e-5 on a GEMM output,
// two orders below the 1e-3 parity bar of the encoders, for 3/16 of the fp32-MFMA cycles.
// W is split once when the model is loaded (mq_split_bf16_f32); A stays fp32 in HBM and LDS and is split
// in registers right before the MFMAs (v_cvt_pk_bf16_f32), so no producer kernel changes.
// Tile 256x256, BK 32, 2-stage LDS: per stage A fp32 [256][32] (128-B rows, chunks swizzled by (r>>1)&7)
// + Wh, Wl bf16 [256][32] (64-B rows, chunks swizzled by (r>>2)&3) = 64 KiB.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int XBK = 32;
constexpr int X_A_BYTES = GT * XBK * 4;   // 32 KiB
constexpr int X_W_BYTES = GT * XBK * 2;   // 16 KiB
constexpr int X_STAGE = X_A_BYTES + 2 * X_W_BYTES;
constexpr int X_LDS_BYTES = 2 * X_STAGE;  // 128 KiB

__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8_t& hi, bf16x8_t& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}

// four values -> their packed (hi, lo) bf16 quadruples (8 bytes each), the arithmetic of split8()
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(const float x0, const float x1, const float x2, const float x3, uint2& hi, uint2& lo) {
    const float x[4] = {x0, x1, x2, x3};
    bf16x4_t h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)x[e];
        h4[e] = h;
        l4[e] = (__bf16)(x[e] - (float)h);
    }
    hi = __builtin_bit_cast(uint2, h4);
    lo = __builtin_bit_cast(uint2, l4);
}

__global__ void split_bf16_kernel(const float* __restrict__ src, long long n, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float x = src[e];
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    hi[e] = __builtin_bit_cast(unsigned short, h);
    lo[e] = __builtin_bit_cast(unsigned short, l);
}

// Weights in TILE layout [N / 256][K / 32][256][32] (rows padded to 256 with zeros): the W operand of one K step of a GEMM
// tile is one contiguous 16-KiB block instead of 256 pieces of 64 bytes K x 2 bytes apart (same arithmetic, other addresses:
// -3 ... -12 % per GEMM, tools/bench_gemm_shapes.py).  One thread per 4 consecutive k of one (padded) row.
__global__ void split_bf16_tiled_kernel(const float* __restrict__ src, int N, int K, unsigned short* __restrict__ hi,
                                        unsigned short* __restrict__ lo) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = K / 4;
    const long long npad = (long long)((N + GT - 1) / GT) * GT;
    if (e >= npad * per_row) return;
    const long long n = e / per_row;
    const int k = (int)(e - n * per_row) * 4;
    unsigned short h[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
    if (n < N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = src[n * K + k + j];
            const __bf16 hb = (__bf16)x;
            const __bf16 lb = (__bf16)(x - (float)hb);
            h[j] = __builtin_bit_cast(unsigned short, hb);
            l[j] = __builtin_bit_cast(unsigned short, lb);
        }
    }
    const size_t at = (((size_t)(n / GT) * (size_t)(K / XBK) + (size_t)(k / XBK)) * GT + (size_t)(n % GT)) * XBK + (size_t)(k % XBK);
    *reinterpret_cast<uint2*>(hi + at) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(lo + at) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}

// Split activations.  An activation tensor that only feeds GEMMs is kept as its (hi, lo) bf16 pair, written by the
// kernel that PRODUCES it (LayerNorm, attention, the GELU epilogue): the consuming GEMM then streams bf16 operands
// only and carries no conversions in its MFMA loop (measured: the in-loop split cost 18-22 % of encoder time).
// hi = bf16(x), lo = bf16(x - hi): exactly what split8() computes, so both GEMM variants give identical results.
// store_split_pair: lanes l and l^1 of a wave hold ADJACENT columns (even column in the even lane); after one DPP
// exchange the even lane stores the packed hi pair, the odd lane the packed lo pair -- one 4-byte store per lane.
__device__ __forceinline__ unsigned split_bits(float x) {
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// PAIR LAYOUT.  The (hi, lo) arrays of a split activation [M, K] (K a multiple of 32) are stored tile by tile, exactly like the
// weights of split_bf16_tiled_kernel: element (row, col) lives at pair_index(row, col, K) = [row / 256][col / 32][row % 256][col % 32],
// and the arrays hold ceil(M / 256) * 256 * K elements (rows beyond M are never written, and never read: the GEMM's A stream
// re-reads row M - 1 instead).  The A operand of one K step of a GEMM tile is then one contiguous 16-KiB block; 8 consecutive
// columns of a row (what the producers store at once) stay 16 contiguous bytes.
__device__ __forceinline__ size_t pair_index(size_t row, int col, int K) {
    return ((row >> 8) * (size_t)(K >> 5) + (size_t)(col >> 5)) * (GT * XBK) + ((row & 255) << 5) + (size_t)(col & 31);
}
__device__ __forceinline__ void store_split_pair(float x, bool valid, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                 size_t even_col_index, int lane) {
    const unsigned mine = split_bits(x);
    const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);  // lane ^ 1
    const bool odd = lane & 1;
    const unsigned ev = odd ? other : mine, od = odd ? mine : other;  // the pair's even / odd column
    const unsigned word = odd ? ((ev >> 16) | (od & 0xFFFF0000u)) : ((ev & 0xFFFFu) | (od << 16));
    if (valid) *reinterpret_cast<unsigned*>((odd ? lo : hi) + even_col_index) = word;
}

#ifndef MQ_GEMM_F32_VIA_LDS
// fp32 outputs of whole tiles leave through the LDS transposition (16-byte stores) like the split ones, unless a residual is added:
// round 5, tools/probe_gemm_m.py, same box -- QKV (bias) 1101-1107 -> 1159-1172 executed TFLOP/s, bias-only K = 3072: 1118 -> 1265;
// with a residual (out-proj 1010-1018 -> 918, FFN2 1172-1233 -> 1198-1203) the direct epilogue stays (1 = the LDS path for those too)
#define MQ_GEMM_F32_VIA_LDS 0
#endif
constexpr int XS_STAGE = 4 * X_W_BYTES;      // Ah, Al, Wh, Wl: [256][32] bf16 each
constexpr int XS_LDS_BYTES = 2 * XS_STAGE;   // 128 KiB

// gemm_nt_x3s_kernel: gemm_nt_x3_kernel with A given as its (hi, lo) pair; SPLIT_OUT writes C as a pair as well.
template <int EPI, bool SPLIT_OUT>
__global__ __launch_bounds__(1024) void gemm_nt_x3s_kernel(const unsigned short* __restrict__ Ah, const unsigned short* __restrict__ Al,
                                                           const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl,
                                                           const float* __restrict__ bias, const float* __restrict__ R,
                                                           float* __restrict__ C, unsigned short* __restrict__ Ch,
                                                           unsigned short* __restrict__ Cl, int M, int N, int K, int ntm, int ntn,
                                                           int wt, int nsplit, const unsigned short* __restrict__ Rl) {
    // wt != 0: Wh, Wl are in tile layout (split_bf16_tiled_kernel).
    // Rl != null (EPI_BIAS_RESIDUAL): the residual is given as a split PAIR in pair layout, R = its hi array, Rl its lo array, value
    // hi + lo (exact in fp32) -- a LayerNorm output that only feeds GEMMs and shortcuts then never exists in fp32 (round 4).
    // nsplit > 1 (split-K, gridDim.y = nsplit): workgroup row y accumulates K steps [y per, (y + 1) per) only and writes its fp32
    // partial to C + y M N (EPI_NONE; mq_gemm_nt_bf16x3s_splitk_f32 sums them) -- for the one shape where K is long and the
    // output tiny (ArcFace's head: 256 x 512 outputs over K = 25,088: two workgroups walking 784 K steps otherwise).
    // Each workgroup walks the output tiles blockIdx.x, + gridDim.x, ... (one tile per workgroup by default; a persistent
    // launch of ~#CU workgroups under MQ_GEMM_WGS).  The first K stage of the NEXT tile is requested during the last K step
    // of the current one, so its DMA round trip overlaps the C-store epilogue.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    // Tile b runs on XCD b & 7 (workgroups go round the XCDs) and is the (b >> 3)-th tile of that XCD: row block
    // ((b >> 3) / ntn) * 8 + (b & 7), column tile (b >> 3) % ntn -- all column tiles of a row block on ONE XCD, so its A rows
    // cross the fabric once.  The tile space is padded to a multiple of 8 row blocks; tiles beyond ntm are skipped (round 3:
    // packed batches have any number of row blocks and used to fall back to b / ntn, every row block on up to ntn XCDs).
    const int ntiles = ((ntm + 7) & ~7) * ntn;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + 16 * w * 64));
    const int i = lane & 31, kg = lane >> 5;
    const int sww = (i >> 2) & 3;
    const char* ard = smem + (64 * wr + i) * 64;                    // Ah rows of this wave; Al at + X_W_BYTES
    const char* wrd = smem + 2 * X_W_BYTES + (64 * wc + i) * 64;    // Wh rows of this wave; Wl at + X_W_BYTES
    const int nk_all = K / XBK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kb0 = __builtin_amdgcn_readfirstlane((int)blockIdx.y * per);
    const int nk = __builtin_amdgcn_readfirstlane(kb0 + per < nk_all ? kb0 + per : nk_all);  // one past this split's last K step
    if (nsplit > 1) C += (size_t)blockIdx.y * (size_t)M * (size_t)N;

    // operand addressing of one tile.  DMA: wave w moves rows [16w, 16w+16) of Ah, Al, Wh, Wl: one instruction of 16 rows
    // x 64 B each.  gridDim.x is a multiple of 8 whenever it is smaller than the tile count, so tile & 7 is this workgroup's XCD.
    struct Tile { const char *ah, *al, *wh, *wl; unsigned a_voff, w_voff; int m0, n0; };
    auto make_tile = [&](int b) __attribute__((always_inline)) {
        const int mt = __builtin_amdgcn_readfirstlane(((b >> 3) / ntn) * 8 + (b & 7));
        const int nt = __builtin_amdgcn_readfirstlane((b >> 3) % ntn);
        Tile t;
        t.m0 = mt * GT;
        t.n0 = nt * GT;
        const int r = 16 * w + (lane >> 2);
        int am = t.m0 + r; if (am > M - 1) am = M - 1;
        int wn = t.n0 + r; if (wn > N - 1) wn = N - 1;
        const size_t chunk = (size_t)(((lane & 3) ^ ((r >> 2) & 3)) * 8);
        t.a_voff = (unsigned)(((size_t)(am - t.m0) * XBK + chunk) * 2);  // pair layout: row within the tile's 16-KiB block
        t.w_voff = wt ? (unsigned)(((size_t)r * XBK + chunk) * 2) : (unsigned)(((size_t)(wn - t.n0) * K + chunk) * 2);
        t.ah = reinterpret_cast<const char*>(Ah + (size_t)t.m0 * K);
        t.al = reinterpret_cast<const char*>(Al + (size_t)t.m0 * K);
        t.wh = reinterpret_cast<const char*>(Wh + (size_t)t.n0 * K);
        t.wl = reinterpret_cast<const char*>(Wl + (size_t)t.n0 * K);
        return t;
    };
    auto issue = [&](const Tile& t, int kb, int stg) __attribute__((always_inline)) {
        const unsigned so = lds0 + stg * XS_STAGE;
        const size_t ko = (size_t)kb * (XBK * 2);
        const size_t kow = wt ? (size_t)kb * X_W_BYTES : ko;  // tile layout: the next K step is the next 16-KiB block
        const size_t koa = (size_t)kb * X_W_BYTES;            // the activation pair is always in pair layout
        dma16s(t.ah + koa, t.a_voff, so);
        dma16s(t.al + koa, t.a_voff, so + X_W_BYTES);
        dma16s(t.wh + kow, t.w_voff, so + 2 * X_W_BYTES);
        dma16s(t.wl + kow, t.w_voff, so + 3 * X_W_BYTES);
    };

    auto valid_from = [&](int t) __attribute__((always_inline)) {  // the first tile t, t + gridDim.x, ... inside the matrix
        while (t < ntiles && ((t >> 3) / ntn) * 8 + (t & 7) >= ntm) t += (int)gridDim.x;
        return t;
    };
    int tile = valid_from(blockIdx.x);
    if (tile >= ntiles) return;
    Tile cur = make_tile(tile);
    issue(cur, kb0, 0);
    int stage = 0;
    for (;;) {
    const int next = valid_from(tile + (int)gridDim.x);
    Tile nxt = cur;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    for (int kb = kb0; kb < nk; ++kb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kb + 1 < nk) issue(cur, kb + 1, stage ^ 1);
        else
        if (next < ntiles) {  // last step: the other stage is free -> the next tile's first K stage
            nxt = make_tile(next);
            issue(nxt, kb0, stage ^ 1);
        }
        const char* as = ard + stage * XS_STAGE;
        const char* ws = wrd + stage * XS_STAGE;
#pragma unroll
        for (int m = 0; m < XBK / 16; ++m) {
            const int o = ((2 * m + kg) ^ sww) * 16;
            const bf16x8_t ah0 = *reinterpret_cast<const bf16x8_t*>(as + o);
            const bf16x8_t ah1 = *reinterpret_cast<const bf16x8_t*>(as + 32 * 64 + o);
            const bf16x8_t al0 = *reinterpret_cast<const bf16x8_t*>(as + X_W_BYTES + o);
            const bf16x8_t al1 = *reinterpret_cast<const bf16x8_t*>(as + X_W_BYTES + 32 * 64 + o);
            const bf16x8_t h0 = *reinterpret_cast<const bf16x8_t*>(ws + o);
            const bf16x8_t h1 = *reinterpret_cast<const bf16x8_t*>(ws + 32 * 64 + o);
            const bf16x8_t l0 = *reinterpret_cast<const bf16x8_t*>(ws + X_W_BYTES + o);
            const bf16x8_t l1 = *reinterpret_cast<const bf16x8_t*>(ws + X_W_BYTES + 32 * 64 + o);
            // same per-accumulator order as gemm_nt_x3_kernel (lo.hi, hi.lo, hi.hi): bit-identical sums
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, h0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, h1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, h0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, h1, acc11, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, l0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, l1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, l0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, l1, acc11, 0, 0, 0);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, h0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, h1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, h0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, h1, acc11, 0, 0, 0);
        }
        stage ^= 1;
    }

    const int m0 = cur.m0, n0 = cur.n0;
    // the C-store epilogue, instantiated twice: a tile that lies wholly inside C (every tile of the encoder shapes) carries no
    // bound tests and no exec-mask juggling around its 64 stores per lane
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        const int nb0 = n0 + 64 * wc + i, nb1 = nb0 + 32;
        float bias0 = 0.f, bias1 = 0.f;
        if (EPI != EPI_NONE) {
            if (FULL || nb0 < N) bias0 = bias[nb0];
            if (FULL || nb1 < N) bias1 = bias[nb1];
        }
        const int mbase = m0 + 64 * wr + 4 * kg;
        // EPI_BIAS_RESIDUAL: the residual of four accumulator rows (16 values per lane) is requested in one burst, behind a
        // scheduling barrier -- left alone, the compiler loads two values, waits, stores, loads two more (the fragment
        // registers are free here, but its scheduler keeps the pressure low), and the wave pays the round trip 32 times
        float rres[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int mr0 = mbase + (reg & 3) + 8 * (reg >> 2), mr1 = mr0 + 32;
            const bool in00 = FULL || (mr0 < M && nb0 < N), in01 = FULL || (mr0 < M && nb1 < N);
            const bool in10 = FULL || (mr1 < M && nb0 < N), in11 = FULL || (mr1 < M && nb1 < N);
            if (EPI == EPI_BIAS_RESIDUAL && (reg & 3) == 0) {  // (a second group in flight ahead of the stores spills: slower)
                if (Rl) {
                    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(R);
                    auto pr = [&](int m_, int n_) __attribute__((always_inline)) {
                        const size_t at = pair_index((size_t)m_, n_, N);
                        return __uint_as_float((unsigned)Rh[at] << 16) + __uint_as_float((unsigned)Rl[at] << 16);
                    };
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ma = mbase + u + 8 * (reg >> 2), mb = ma + 32;
                        rres[4 * u + 0] = (FULL || (ma < M && nb0 < N)) ? pr(ma, nb0) : 0.f;
                        rres[4 * u + 1] = (FULL || (ma < M && nb1 < N)) ? pr(ma, nb1) : 0.f;
                        rres[4 * u + 2] = (FULL || (mb < M && nb0 < N)) ? pr(mb, nb0) : 0.f;
                        rres[4 * u + 3] = (FULL || (mb < M && nb1 < N)) ? pr(mb, nb1) : 0.f;
                    }
                } else
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ma = mbase + u + 8 * (reg >> 2), mb = ma + 32;
                    rres[4 * u + 0] = (FULL || (ma < M && nb0 < N)) ? R[(size_t)ma * N + nb0] : 0.f;
                    rres[4 * u + 1] = (FULL || (ma < M && nb1 < N)) ? R[(size_t)ma * N + nb1] : 0.f;
                    rres[4 * u + 2] = (FULL || (mb < M && nb0 < N)) ? R[(size_t)mb * N + nb0] : 0.f;
                    rres[4 * u + 3] = (FULL || (mb < M && nb1 < N)) ? R[(size_t)mb * N + nb1] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float v00 = acc00[reg] + bias0, v01 = acc01[reg] + bias1, v10 = acc10[reg] + bias0, v11 = acc11[reg] + bias1;
            if (EPI == EPI_BIAS_GELU) { v00 = gelu_erf(v00); v01 = gelu_erf(v01); v10 = gelu_erf(v10); v11 = gelu_erf(v11); }
            if (EPI == EPI_BIAS_QUICKGELU) { v00 = quick_gelu(v00); v01 = quick_gelu(v01); v10 = quick_gelu(v10); v11 = quick_gelu(v11); }
            if (EPI == EPI_BIAS_RESIDUAL) {
                v00 += rres[4 * (reg & 3) + 0];
                v01 += rres[4 * (reg & 3) + 1];
                v10 += rres[4 * (reg & 3) + 2];
                v11 += rres[4 * (reg & 3) + 3];
            }
            if (SPLIT_OUT) {  // N is even (checked by the host): a pair of columns is in or out together
                store_split_pair(v00, in00, Ch, Cl, pair_index(mr0, nb0 & ~1, N), lane);
                store_split_pair(v01, in01, Ch, Cl, pair_index(mr0, nb1 & ~1, N), lane);
                store_split_pair(v10, in10, Ch, Cl, pair_index(mr1, nb0 & ~1, N), lane);
                store_split_pair(v11, in11, Ch, Cl, pair_index(mr1, nb1 & ~1, N), lane);
            } else {
                if (in00) C[(size_t)mr0 * N + nb0] = v00;
                if (in01) C[(size_t)mr0 * N + nb1] = v01;
                if (in10) C[(size_t)mr1 * N + nb0] = v10;
                if (in11) C[(size_t)mr1 * N + nb1] = v11;
            }
        }
    };
    if ((SPLIT_OUT || EPI != EPI_BIAS_RESIDUAL || MQ_GEMM_F32_VIA_LDS) && m0 + GT <= M && n0 + GT <= N && (N & 7) == 0) {
        // Round 3 for SPLIT outputs (FFN1 + GELU 2.90 -> 2.83 ms), round 5 for fp32 outputs without a residual (above; in round 3,
        // before the tile layouts, the detour cost them 2-6 %): the C store of a tile that lies wholly inside C goes through LDS, one 32 x 32 accumulator at a time, so that
        // a lane ends up with EIGHT consecutive columns of one row: 16-byte stores (two for fp32, one each for the hi and the
        // lo halves of a split output) and 16-byte residual loads, instead of 4-byte ones -- a quarter of the memory
        // instructions, and the split pair needs no DPP exchange.  Bias and activation are applied in the accumulator layout
        // (one column per lane), the residual after the transposition: per element the operations and their order are those
        // of the direct epilogue, so the results keep their bits.  Scratch: the LDS stage the K loop has just finished with
        // (the other one may be receiving the next tile's first K step), 4 KiB per wave, 16-byte groups XOR-swizzled by the row.
        __syncthreads();  // every wave has read its last operands from that stage
        float* Ot = reinterpret_cast<float*>(smem + (stage ^ 1) * XS_STAGE) + w * 1024;
        const int nb0 = n0 + 64 * wc + i, nb1 = nb0 + 32;
        const float bias0 = EPI != EPI_NONE ? bias[nb0] : 0.f, bias1 = EPI != EPI_NONE ? bias[nb1] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x16& acc = a == 0 ? acc00 : a == 1 ? acc01 : a == 2 ? acc10 : acc11;
            const float bs = (a & 1) ? bias1 : bias0;
            // the residual of this accumulator's two row groups, in the TRANSPOSED layout (8 consecutive columns of one row per
            // lane), requested before the trip through LDS so that its round trip overlaps it (round 5: requested after the LDS
            // reads, each accumulator paid it in full -- the reason fp32 outputs with a residual avoided this path)
            float4 ru[2], rv[2];
            if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = 16 * it + (lane >> 2), g = lane & 3;
                    if (Rl) {
                        const size_t rp = pair_index((size_t)(m0 + 64 * wr + 32 * (a >> 1) + row), n0 + 64 * wc + 32 * (a & 1) + 8 * g, N);
                        ru[it] = __builtin_bit_cast(float4, *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(R) + rp));
                        rv[it] = __builtin_bit_cast(float4, *reinterpret_cast<const uint4*>(Rl + rp));
                    } else {
                        const size_t at = (size_t)(m0 + 64 * wr + 32 * (a >> 1) + row) * N + (n0 + 64 * wc + 32 * (a & 1) + 8 * g);
                        ru[it] = *reinterpret_cast<const float4*>(R + at);
                        rv[it] = *reinterpret_cast<const float4*>(R + at + 4);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                float v = acc[reg] + bs;
                if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EPI_BIAS_QUICKGELU) v = quick_gelu(v);
                Ot[row * 32 + ((((i >> 3) ^ (row & 3)) << 3) | (i & 7))] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads what it has just written (LDS is in order per wave)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = 16 * it + (lane >> 2), g = lane & 3;
                const float* src = Ot + row * 32 + ((g ^ (row & 3)) << 3);
                float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
                const size_t at = (size_t)(m0 + 64 * wr + 32 * (a >> 1) + row) * N + (n0 + 64 * wc + 32 * (a & 1) + 8 * g);
                if (EPI == EPI_BIAS_RESIDUAL) {
                    float4 xu, xv;
                    if (Rl) {
                        const uint4 h4 = __builtin_bit_cast(uint4, ru[it]), l4 = __builtin_bit_cast(uint4, rv[it]);
                        auto two = [](unsigned hw, unsigned lw, float& e0, float& e1) __attribute__((always_inline)) {
                            e0 = __uint_as_float(hw << 16) + __uint_as_float(lw << 16);
                            e1 = __uint_as_float(hw & 0xFFFF0000u) + __uint_as_float(lw & 0xFFFF0000u);
                        };
                        two(h4.x, l4.x, xu.x, xu.y); two(h4.y, l4.y, xu.z, xu.w);
                        two(h4.z, l4.z, xv.x, xv.y); two(h4.w, l4.w, xv.z, xv.w);
                    } else {
                        xu = ru[it];
                        xv = rv[it];
                    }
                    u.x += xu.x; u.y += xu.y; u.z += xu.z; u.w += xu.w;
                    v.x += xv.x; v.y += xv.y; v.z += xv.z; v.w += xv.w;
                }
                if (SPLIT_OUT) {
                    bf16x8_t hi8, lo8;
                    split8(u, v, hi8, lo8);
                    const size_t pt = pair_index((size_t)(m0 + 64 * wr + 32 * (a >> 1) + row), n0 + 64 * wc + 32 * (a & 1) + 8 * g, N);
                    *reinterpret_cast<uint4*>(Ch + pt) = __builtin_bit_cast(uint4, hi8);
                    *reinterpret_cast<uint4*>(Cl + pt) = __builtin_bit_cast(uint4, lo8);
                } else {
                    *reinterpret_cast<float4*>(C + at) = u;
                    *reinterpret_cast<float4*>(C + at + 4) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next accumulator overwrites the scratch
        }
    } else if (m0 + GT <= M && n0 + GT <= N) epilogue(std::true_type{}); else epilogue(std::false_type{});
    if (next >= ntiles) break;
    tile = next;
    cur = nxt;
    }  // tiles
}

#include "gemm_x3w.inc"

template <int EPI>
__global__ __launch_bounds__(1024) void gemm_nt_x3_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wh,
                                                          const unsigned short* __restrict__ Wl, const float* __restrict__ bias,
                                                          const float* __restrict__ R, float* __restrict__ C, int M, int N, int K,
                                                          int ntm, int ntn, int wt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int mt = __builtin_amdgcn_readfirstlane((((int)blockIdx.x >> 3) / ntn) * 8 + ((int)blockIdx.x & 7));  // see gemm_nt_x3s_kernel
    const int nt = __builtin_amdgcn_readfirstlane(((int)blockIdx.x >> 3) % ntn);
    if (mt >= ntm) return;  // padding of the tile space to 8 row blocks
    const int m0 = mt * GT, n0 = nt * GT;

    // DMA: wave w moves rows [16w, 16w+16) of A (two instructions of 8 rows x 128 B) and of Wh, Wl (one
    // instruction of 16 rows x 64 B each)
    unsigned a_voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = 16 * w + 8 * u + (lane >> 3);
        int am = m0 + r; if (am > M - 1) am = M - 1;
        a_voff[u] = (unsigned)(((size_t)(am - m0) * K + (size_t)(((lane & 7) ^ ((r >> 1) & 7)) * 4)) * 4);
    }
    unsigned w_voff;
    {
        const int r = 16 * w + (lane >> 2);
        int wn = n0 + r; if (wn > N - 1) wn = N - 1;
        const size_t chunk = (size_t)(((lane & 3) ^ ((r >> 2) & 3)) * 8);
        w_voff = wt ? (unsigned)(((size_t)r * XBK + chunk) * 2) : (unsigned)(((size_t)(wn - n0) * K + chunk) * 2);
    }
    const char* const abase = reinterpret_cast<const char*>(A + (size_t)m0 * K);
    const char* const whbase = reinterpret_cast<const char*>(Wh + (size_t)n0 * K);
    const char* const wlbase = reinterpret_cast<const char*>(Wl + (size_t)n0 * K);
    const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + 16 * w * 128));
    const unsigned lds_wh = __builtin_amdgcn_readfirstlane(lds_addr_of(smem + X_A_BYTES + 16 * w * 64));
    const unsigned lds_wl = lds_wh + X_W_BYTES;
    auto issue = [&](int kb, int stg) __attribute__((always_inline)) {
        const unsigned so = stg * X_STAGE;
        dma16s(abase + (size_t)kb * (XBK * 4), a_voff[0], lds_a + so);
        dma16s(abase + (size_t)kb * (XBK * 4), a_voff[1], lds_a + so + 1024);
        const size_t kow = wt ? (size_t)kb * X_W_BYTES : (size_t)kb * (XBK * 2);
        dma16s(whbase + kow, w_voff, lds_wh + so);
        dma16s(wlbase + kow, w_voff, lds_wl + so);
    };

    const int i = lane & 31, kg = lane >> 5;
    const int swa = (i >> 1) & 7, sww = (i >> 2) & 3;
    const char* ard = smem + (64 * wr + i) * 128;
    const char* hrd = smem + X_A_BYTES + (64 * wc + i) * 64;
    const char* lrd = hrd + X_W_BYTES;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nk = K / XBK;
    issue(0, 0);
    int stage = 0;
    for (int kb = 0; kb < nk; ++kb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kb + 1 < nk) issue(kb + 1, stage ^ 1);
        const char* as = ard + stage * X_STAGE;
        const char* hs = hrd + stage * X_STAGE;
        const char* ls = lrd + stage * X_STAGE;
#pragma unroll
        for (int m = 0; m < XBK / 16; ++m) {
            const int wo = ((2 * m + kg) ^ sww) * 16;
            const bf16x8_t h0 = *reinterpret_cast<const bf16x8_t*>(hs + wo);
            const bf16x8_t h1 = *reinterpret_cast<const bf16x8_t*>(hs + 32 * 64 + wo);
            const bf16x8_t l0 = *reinterpret_cast<const bf16x8_t*>(ls + wo);
            const bf16x8_t l1 = *reinterpret_cast<const bf16x8_t*>(ls + 32 * 64 + wo);
            const int ca = 4 * m + 2 * kg;
            const int ao0 = (ca ^ swa) * 16, ao1 = ((ca + 1) ^ swa) * 16;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float4 u = *reinterpret_cast<const float4*>(as + a * 32 * 128 + ao0);
                const float4 v = *reinterpret_cast<const float4*>(as + a * 32 * 128 + ao1);
                bf16x8_t ah, al;
                split8(u, v, ah, al);
                if (a == 0) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h1, acc01, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l1, acc01, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h1, acc01, 0, 0, 0);
                } else {
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h1, acc11, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l1, acc11, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h1, acc11, 0, 0, 0);
                }
            }
        }
        stage ^= 1;
    }

    // epilogue (same C/D map as gemm_nt_kernel)
    const int nb0 = n0 + 64 * wc + i, nb1 = nb0 + 32;
    float bias0 = 0.f, bias1 = 0.f;
    if (EPI != EPI_NONE) {
        if (nb0 < N) bias0 = bias[nb0];
        if (nb1 < N) bias1 = bias[nb1];
    }
    const int mbase = m0 + 64 * wr + 4 * kg;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int mr0 = mbase + (reg & 3) + 8 * (reg >> 2), mr1 = mr0 + 32;
        float v00 = acc00[reg] + bias0, v01 = acc01[reg] + bias1, v10 = acc10[reg] + bias0, v11 = acc11[reg] + bias1;
        if (EPI == EPI_BIAS_GELU) { v00 = gelu_erf(v00); v01 = gelu_erf(v01); v10 = gelu_erf(v10); v11 = gelu_erf(v11); }
        if (EPI == EPI_BIAS_QUICKGELU) { v00 = quick_gelu(v00); v01 = quick_gelu(v01); v10 = quick_gelu(v10); v11 = quick_gelu(v11); }
        if (mr0 < M) {
            if (nb0 < N) { if (EPI == EPI_BIAS_RESIDUAL) v00 += R[(size_t)mr0 * N + nb0]; C[(size_t)mr0 * N + nb0] = v00; }
            if (nb1 < N) { if (EPI == EPI_BIAS_RESIDUAL) v01 += R[(size_t)mr0 * N + nb1]; C[(size_t)mr0 * N + nb1] = v01; }
        }
        if (mr1 < M) {
            if (nb0 < N) { if (EPI == EPI_BIAS_RESIDUAL) v10 += R[(size_t)mr1 * N + nb0]; C[(size_t)mr1 * N + nb0] = v10; }
            if (nb1 < N) { if (EPI == EPI_BIAS_RESIDUAL) v11 += R[(size_t)mr1 * N + nb1]; C[(size_t)mr1 * N + nb1] = v11; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm family: one wave per row, row held in registers (C <= 1024)
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXPER = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// normalises v[] (this lane's elements c = lane + 64 t, t < nper) and stores
// out (fp32 row, may be null) and/or oh, ol (the row's split pair, may be null; C even then)
// (oh, ol: the WHOLE pair arrays -- pair layout -- and `prow` the row's index in them)
__device__ __forceinline__ void ln_store(float (&v)[LN_MAXPER], int C, int lane, const float* __restrict__ g,
                                         const float* __restrict__ b, float eps, float* __restrict__ out,
                                         unsigned short* __restrict__ oh = nullptr, unsigned short* __restrict__ ol = nullptr,
                                         size_t prow = 0) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; if (c < C) s += v[t]; }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; if (c < C) { const float d = v[t] - mean; q += d * d; } }
    const float var = wave_sum(q) / (float)C;
    const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) {
        const int c = lane + 64 * t;
        const float y = (c < C) ? (v[t] - mean) * inv * g[c] + b[c] : 0.f;
        if (out && c < C) out[c] = y;
        if (oh && 64 * t < C) store_split_pair(y, c < C, oh, ol, pair_index(prow, c & ~1, C), lane);  // wave-uniform guard
    }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ Y,
                                                        unsigned short* __restrict__ Yh, unsigned short* __restrict__ Yl, int M, int C,
                                                        float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float v[LN_MAXPER];
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; v[t] = (c < C) ? X[(size_t)row * C + c] : 0.f; }
    ln_store(v, C, lane, g, b, eps, Y ? Y + (size_t)row * C : nullptr, Yh, Yl, (size_t)row);
}

// layernorm_kernel's split output with FULL-LINE stores.  In the pair layout a row's 32 columns of one K block are 64 bytes, and
// rows r .. r + 3 (this workgroup's four waves) are 256 consecutive bytes: the four rows' (hi, lo) values go through LDS as packed
// words and leave as 16-byte pieces, sixteen lanes covering two whole 128-byte lines of one array -- layernorm_kernel's own pair
// stores are 4 bytes per lane in 64-byte runs 16 KiB apart.  The arithmetic (and every bit of the output) is ln_store's.
constexpr int LNP_STRIDE = 64 * LN_MAXPER + 32;   // words per row in LDS: rows 32 banks apart
__global__ __launch_bounds__(256) void layernorm_pairs_kernel(const float* __restrict__ X, const float* __restrict__ g,
                                                              const float* __restrict__ b, float* __restrict__ Y,
                                                              unsigned short* __restrict__ Yh, unsigned short* __restrict__ Yl, int M, int C,
                                                              float eps) {
    __shared__ __attribute__((aligned(16))) unsigned words[4 * LNP_STRIDE];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 4, row = row0 + w;
    if (row < M) {   // wave-uniform
        float v[LN_MAXPER];
#pragma unroll
        for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; v[t] = (c < C) ? X[(size_t)row * C + c] : 0.f; }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; if (c < C) s += v[t]; }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; if (c < C) { const float d = v[t] - mean; q += d * d; } }
        const float var = wave_sum(q) / (float)C;
        const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int t = 0; t < LN_MAXPER; ++t) {
            const int c = lane + 64 * t;
            if (c < C) {
                const float y = (v[t] - mean) * inv * g[c] + b[c];
                if (Y) Y[(size_t)row * C + c] = y;
                words[w * LNP_STRIDE + c] = split_bits(y);
            }
        }
    }
    __syncthreads();
    // piece p of the workgroup: array (hi / lo), K block, row of the four, eight columns
    const int per_array = (C >> 5) * 16;
    for (int p = threadIdx.x; p < 2 * per_array; p += 256) {
        const int arr = p >= per_array, rem = arr ? p - per_array : p;
        const int cb = rem >> 4, r = (rem >> 2) & 3, part = rem & 3;
        if (row0 + r >= M) continue;
        const uint4* src = reinterpret_cast<const uint4*>(words + r * LNP_STRIDE + cb * 32 + part * 8);
        const uint4 a = src[0], c4 = src[1];
        uint4 o;
        if (arr) { o.x = (a.x >> 16) | (a.y & 0xFFFF0000u); o.y = (a.z >> 16) | (a.w & 0xFFFF0000u); o.z = (c4.x >> 16) | (c4.y & 0xFFFF0000u); o.w = (c4.z >> 16) | (c4.w & 0xFFFF0000u); }
        else { o.x = (a.x & 0xFFFFu) | (a.y << 16); o.y = (a.z & 0xFFFFu) | (a.w << 16); o.z = (c4.x & 0xFFFFu) | (c4.y << 16); o.w = (c4.z & 0xFFFFu) | (c4.w << 16); }
        *reinterpret_cast<uint4*>((arr ? Yl : Yh) + pair_index((size_t)(row0 + r), cb * 32 + part * 8, C)) = o;
    }
}

// BertEmbeddings (meerqat/models/bert.py:153-214): (word[id] + type[tt]) + pos[t], then LayerNorm
__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const long long* __restrict__ ids, const long long* __restrict__ tts,
                                                            const float* __restrict__ word, const float* __restrict__ pos,
                                                            const float* __restrict__ type, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ out,
                                                            unsigned short* __restrict__ oh, unsigned short* __restrict__ ol, int M,
                                                            int L, int H, float eps, const int* __restrict__ pos_ids = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const long long id = ids[row];
    const long long tt = tts ? tts[row] : 0;
    const int t_pos = pos_ids ? pos_ids[row] : row % L;  // packed rows carry their position inside their sequence
    float v[LN_MAXPER];
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) {
        const int c = lane + 64 * t;
        v[t] = (c < H) ? (word[(size_t)id * H + c] + type[(size_t)tt * H + c]) + pos[(size_t)t_pos * H + c] : 0.f;
    }
    ln_store(v, H, lane, g, b, eps, out + (size_t)row * H, oh, ol, (size_t)row);
}

// CLIPVisionEmbeddings + pre_layrnorm: token 0 = class embedding, token 1+p = patch embedding p; + position
__global__ __launch_bounds__(256) void clip_assemble_ln_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                               const float* __restrict__ pos, const float* __restrict__ g,
                                                               const float* __restrict__ b, float* __restrict__ out, int M,
                                                               int T, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const int bi = row / T, t_pos = row % T;
    const float* src = t_pos == 0 ? cls : pe + ((size_t)bi * (T - 1) + (t_pos - 1)) * H;
    float v[LN_MAXPER];
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) {
        const int c = lane + 64 * t;
        v[t] = (c < H) ? src[c] + pos[(size_t)t_pos * H + c] : 0.f;
    }
    ln_store(v, H, lane, g, b, eps, out + (size_t)row * H);
}

// pixels [B,Cn,S,S] -> rows [B*G*G, Cn*P*P] in (c, ph, pw) order = flattened conv weight order
__global__ void clip_patchify_kernel(const float* __restrict__ px, float* __restrict__ out, int B, int Cn, int S, int P) {
    const int G = S / P;
    const size_t total4 = (size_t)B * Cn * S * S / 4;
    const size_t e4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e4 >= total4) return;
    const size_t e = e4 * 4;  // output element index, 4 consecutive pw
    const int KP = Cn * P * P;
    const size_t rowi = e / KP;
    const int col = (int)(e - rowi * KP);
    const int c = col / (P * P), ph = (col / P) % P, pw = col % P;
    const int bi = (int)(rowi / (G * G)), gy = (int)(rowi % (G * G)) / G, gx = (int)(rowi % (G * G)) % G;
    const float4 v = *reinterpret_cast<const float4*>(px + (((size_t)bi * Cn + c) * S + gy * P + ph) * S + gx * P + pw);
    *reinterpret_cast<float4*>(out + e) = v;
}

constexpr int DH = 64;  // attention head size


// ------------------------------------------------------------------------------------------------
// MFMA attention (fp32): one workgroup (4 waves) per (sequence, head, block of 128 queries).
// S^T = K . Q^T with K as the A operand, so each lane column is ONE query: the softmax over keys is a
// per-lane reduction + one exchange with lane^32.  The probabilities then feed the P.V MFMAs as the
// B operand straight from the accumulator registers (lanes 0-31 hold keys = 0..3 mod 8, lanes 32-63
// keys = 4..7 mod 8, which is exactly a 2-deep k pair), with V^T rows as the A operand.
// ------------------------------------------------------------------------------------------------
template <int NKT, bool MULTI>  // key tiles of 32 per key block; MULTI: sequences longer than one block loop over blocks (online softmax)
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, const long long* __restrict__ mask,
                                                             float* __restrict__ out, unsigned short* __restrict__ out_h,
                                                             unsigned short* __restrict__ out_l, int L, int heads, float scale,
                                                             int causal, const int* __restrict__ cu = nullptr,
                                                             const int* __restrict__ seq_ids = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = 32 * NKT;
    float* Ks = reinterpret_cast<float*>(smem);        // [NK][65]
    float* Vs = Ks + NK * 65;                           // [NK][64]
    float* addm = Vs + NK * 64;                         // [NK] 0 or -inf
    // dense batch: sequence bi = rows [bi L, bi L + L).  Packed batch (cu != NULL): this launch covers the sequences
    // listed in seq_ids (one length class); sequence bi = rows [cu[bi], cu[bi + 1]) of the packed token matrix, every key
    // is real (no mask), and L is the class's longest length (grid / wave count only).
    const int h = blockIdx.x % heads;
    const int bi = seq_ids ? seq_ids[blockIdx.x / heads] : (int)(blockIdx.x / heads);
    const size_t row0 = cu ? (size_t)cu[bi] : (size_t)bi * L;
    if (cu) {
        L = cu[bi + 1] - cu[bi];
        if ((int)blockIdx.y * 128 >= L) return;  // block-uniform: no query of this sequence in the block
    }
    const int H = heads * DH, ld = 3 * H;
    const float* base = qkv + row0 * ld + h * DH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthr = blockDim.x;  // 64 * min(4, ceil(queries of this block / 32)) threads: no wave without queries
    const int i = lane & 31, kh = lane >> 5;
    const int qrow = blockIdx.y * 128 + 32 * w + i;   // this lane's query
    float qreg[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) qreg[kk] = (qrow < L) ? base[(size_t)qrow * ld + 2 * kk + kh] : 0.f;

    // Key blocks of NK keys with the online softmax (running maximum m_run, denominator den, rescaled accumulators);
    // a sequence that fits one block (L <= NK: every shipped config but ECA's text + faces + image) takes the loop once
    // and computes exactly what the single-block kernel did.
    float m_run = -INFINITY, den = 0.f;
    f32x16 oacc0 = {0}, oacc1 = {0};
    for (int kb0 = 0; kb0 < (MULTI ? L : 1); kb0 += NK) {  // !MULTI: exactly one pass, known at compile time
    if (kb0) __syncthreads();  // the previous block's K / V are dead
    for (int e = tid; e < NK * (DH / 4); e += nthr) {
        const int j = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
        float4 kf = make_float4(0.f, 0.f, 0.f, 0.f), vf = kf;
        if (kb0 + j < L) {
            kf = *reinterpret_cast<const float4*>(base + (size_t)(kb0 + j) * ld + H + c4);
            vf = *reinterpret_cast<const float4*>(base + (size_t)(kb0 + j) * ld + 2 * H + c4);
        }
        float* kd = Ks + j * 65 + c4;
        kd[0] = kf.x; kd[1] = kf.y; kd[2] = kf.z; kd[3] = kf.w;
        *reinterpret_cast<float4*>(Vs + j * 64 + c4) = vf;
    }
    for (int j = tid; j < NK; j += nthr)
        addm[j] = (kb0 + j < L && (!mask || mask[row0 + kb0 + j] != 0)) ? 0.f : -INFINITY;
    __syncthreads();

    f32x16 sacc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) sacc[t] = (f32x16){0};
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            const float a = Ks[(32 * t + i) * 65 + 2 * kk + kh];
            sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[kk], sacc[t], 0, 0, 0);
        }
    }
    // sacc[t][reg] = <k_key, q_query>, key = 32 t + (reg&3) + 8 (reg>>2) + 4 kh, query = this lane's column
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = 32 * t + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            float sv = sacc[t][reg] * scale + addm[key];
            if (causal && kb0 + key > qrow) sv = -INFINITY;   // CLIP text tower: a token attends to itself and the past
            sacc[t][reg] = sv;
            mx = fmaxf(mx, sv);
        }
    }
    mx = fmaxf(fmaxf(mx, __shfl_xor(mx, 32)), m_run);
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float p = (sacc[t][reg] == -INFINITY) ? 0.f : expf(sacc[t][reg] - mx);
            sacc[t][reg] = p;
            bsum += p;
        }
    }
    bsum += __shfl_xor(bsum, 32);
    if (kb0) {  // rescale what the earlier blocks accumulated to the new maximum
        const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - mx);
        den *= alpha;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { oacc0[reg] *= alpha; oacc1[reg] *= alpha; }
    }
    den += bsum;
    m_run = mx;

#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = 32 * t + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            const float p = sacc[t][reg];
            oacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + i], p, oacc0, 0, 0, 0);
            oacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + 32 + i], p, oacc1, 0, 0, 0);
        }
    }
    }  // key blocks
    // oacc{0,1}[reg] = sum_key p V[key][d], d = 32 dt + (reg&3) + 8 (reg>>2) + 4 kh, for this lane's query.
    // Transpose through LDS (K/V are dead) so that every query row is stored as 256 contiguous bytes.
    __syncthreads();
    float* Ot = reinterpret_cast<float*>(smem) + w * (32 * 65);   // [32 queries][65]
    const float inv = 1.0f / den;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int d = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        Ot[i * 65 + d] = oacc0[reg] * inv;
        Ot[i * 65 + 32 + d] = oacc1[reg] * inv;
    }
    __syncthreads();
    for (int e = lane; e < 32 * 64; e += 64) {
        const int r = e >> 6, c = e & 63;  // c = lane: lanes l, l^1 hold adjacent columns
        const int qr = blockIdx.y * 128 + 32 * w + r;
        const size_t at = (row0 + qr) * H + h * DH;
        const float val = Ot[r * 65 + c];
        if (out && qr < L) out[at + c] = val;
        if (out_h) store_split_pair(val, qr < L, out_h, out_l, pair_index(row0 + qr, h * DH + (c & ~1), H), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 attention: the same softmax(q k^T * scale + mask) v on the bf16 matrix pipe at fp32-class accuracy.
// K and V^T are staged in LDS as (hi, lo) bf16 pairs, Q is split in registers, and every product is the three-term
// sum lo.hi + hi.lo + hi.hi (fp32 accumulate), as in the split-bf16 GEMM: 12 MFMAs of 32 cycles per 32-key tile for
// S and 12 for P.V instead of 32 + 32 fp32 MFMAs of 64 cycles.  S^T = K . Q^T keeps one query per lane column
// (softmax = per-lane reduction + one lane^32 exchange).  For P.V the probabilities are the B operand: lane
// (query j, half kg) needs P for 8 CONSECUTIVE keys, while the accumulator layout gave it keys {0-3, 8-11, ...} + 4 kh:
// four values per 16-key group are exchanged with lane^32, then split into (hi, lo).
// LDS: Kh, Kl [NK][64] bf16 (16-byte chunks swizzled by (key >> 1) & 7), Vth, Vtl [64][NK] bf16 (chunks swizzled by
// d & (NK/8 - 1)), mask row.
// ------------------------------------------------------------------------------------------------
// WIDE: up to 8 waves (256 queries) per workgroup -- a whole packed sequence of 129..256 tokens stages its K / V once
// instead of once per 128-query block (224 VGPRs: two waves per SIMD fit).
template <int NKT, bool MULTI, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? 512 : 256) void attention_x3_kernel(const float* __restrict__ qkv, const long long* __restrict__ mask,
                                                           float* __restrict__ out, unsigned short* __restrict__ out_h,
                                                           unsigned short* __restrict__ out_l, int L, int heads, float scale,
                                                           int causal, const int* __restrict__ cu = nullptr,
                                                           const int* __restrict__ seq_ids = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NK = 32 * NKT, NCH = NK / 8;           // 16-byte chunks (8 keys) per V^T row
    constexpr int K_BYTES = NK * 128, V_BYTES = 64 * NK * 2;
    char* Kh = smem;
    char* Kl = Kh + K_BYTES;
    char* Vth = Kl + K_BYTES;
    char* Vtl = Vth + V_BYTES;
    float* addm = reinterpret_cast<float*>(Vtl + V_BYTES);  // [NK] 0 or -inf
    // dense batch: sequence bi = rows [bi L, bi L + L).  Packed batch (cu != NULL): this launch covers the sequences
    // listed in seq_ids (one length class); sequence bi = rows [cu[bi], cu[bi + 1]) of the packed token matrix, every key
    // is real (no mask), and L is the class's longest length (grid / wave count only).
    const int h = blockIdx.x % heads;
    const int bi = seq_ids ? seq_ids[blockIdx.x / heads] : (int)(blockIdx.x / heads);
    const size_t row0 = cu ? (size_t)cu[bi] : (size_t)bi * L;
    if (cu) {
        L = cu[bi + 1] - cu[bi];
        if ((int)blockIdx.y * (int)(blockDim.x >> 1) >= L) return;  // block-uniform: no query of this sequence in the block
    }
    const int H = heads * DH, ld = 3 * H;
    const float* base = qkv + row0 * ld + h * DH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthr = blockDim.x;
    const int i = lane & 31, kg = lane >> 5;
    const int qpb = nthr >> 1;                         // queries per workgroup: 32 per wave
    const bool wave_active = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * qpb + 32 * w)) < L;
    // key tiles that hold at least one key of this sequence (block-uniform).  A packed sequence of 150 tokens in the
    // 256-key instantiation has 5 live tiles of 8: the dead ones would only multiply zeros (their probabilities are exactly
    // 0 and their V rows are staged as zeros), so skipping their staging, MFMAs and exponentials changes no bit.
    const int nt_live = MULTI ? NKT : ((L + 31) / 32 < NKT ? (L + 31) / 32 : NKT);
    const int qrow = blockIdx.y * qpb + 32 * w + i;   // this lane's query
    bf16x8_t qh[4], ql[4];                            // d in [16 m + 8 kg, + 8)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
        if (qrow < L) {
            const float* qp = base + (size_t)qrow * ld + 16 * m + 8 * kg;
            u = *reinterpret_cast<const float4*>(qp);
            v = *reinterpret_cast<const float4*>(qp + 4);
        }
        split8(u, v, qh[m], ql[m]);
    }

    // key blocks of NK keys with the online softmax (see attention_mfma_kernel): one pass when L <= NK
    const float scale2 = scale * 1.4426950408889634f;  // softmax(x) = 2^(x log2 e - max) / sum: one multiply folded into the scale
    float m_run = -INFINITY, den = 0.f;
    f32x16 oacc0 = {0}, oacc1 = {0};
    for (int kb0 = 0; kb0 < (MULTI ? L : 1); kb0 += NK) {  // !MULTI: exactly one pass, known at compile time
    if (kb0) __syncthreads();  // the previous block's K / V are dead
    // K / V staging, one item = 4 consecutive keys x 4 consecutive d: K rows leave as 8-byte (hi, lo) pieces as before; V^T
    // gets, for each of the 4 d, the 4 keys as ONE 8-byte piece per (hi, lo) instead of four 2-byte stores (keys 4 kq .. + 3
    // share a 16-byte chunk), and the conversions run on packed pairs
    for (int e0 = tid; e0 < nt_live * 8 * (DH / 4); e0 += 2 * nthr) {
        // two items per round, all sixteen 16-byte loads issued before the first conversion: ONE round trip to HBM / L2 for
        // the K / V of a 100-key sequence instead of two (the kernel is bound by these round trips, not by its arithmetic:
        // a variant that read q / k / v as ready-made bf16 pairs -- no conversion at all -- measured 3 % SLOWER, round 3)
        float4 kf2[2][4], vf2[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = e0 + it * nthr;
            const int kq = e / (DH / 4), c4 = (e % (DH / 4)) * 4, j0 = 4 * kq;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kf2[it][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                vf2[it][u] = kf2[it][u];
                if (e < nt_live * 8 * (DH / 4) && kb0 + j0 + u < L) {
                    kf2[it][u] = *reinterpret_cast<const float4*>(base + (size_t)(kb0 + j0 + u) * ld + H + c4);
                    vf2[it][u] = *reinterpret_cast<const float4*>(base + (size_t)(kb0 + j0 + u) * ld + 2 * H + c4);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
        const int e = e0 + it * nthr;
        if (e >= nt_live * 8 * (DH / 4)) break;
        const int kq = e / (DH / 4), c4 = (e % (DH / 4)) * 4, j0 = 4 * kq;
        const float4* kf = kf2[it];
        const float4* vf = vf2[it];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // K row j0 + u: four consecutive d -> 8 bytes inside chunk c4 / 8
            const int j = j0 + u;
            uint2 h2, l2;
            split4(kf[u].x, kf[u].y, kf[u].z, kf[u].w, h2, l2);
            const int koff = j * 128 + (((c4 >> 3) ^ ((j >> 1) & 7)) * 16) + (c4 & 7) * 2;
            *reinterpret_cast<uint2*>(Kh + koff) = h2;
            *reinterpret_cast<uint2*>(Kl + koff) = l2;
        }
        const float vv[4][4] = {{vf[0].x, vf[1].x, vf[2].x, vf[3].x}, {vf[0].y, vf[1].y, vf[2].y, vf[3].y},
                                {vf[0].z, vf[1].z, vf[2].z, vf[3].z}, {vf[0].w, vf[1].w, vf[2].w, vf[3].w}};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {  // V^T row c4 + jj, columns j0 .. j0 + 3
            const int d = c4 + jj;
            uint2 h2, l2;
            split4(vv[jj][0], vv[jj][1], vv[jj][2], vv[jj][3], h2, l2);
            const int voff = d * (NK * 2) + (((j0 >> 3) ^ (d & (NCH - 1))) * 16) + (j0 & 7) * 2;
            *reinterpret_cast<uint2*>(Vth + voff) = h2;
            *reinterpret_cast<uint2*>(Vtl + voff) = l2;
        }
        }
    }
    for (int j = tid; j < NK; j += nthr)
        addm[j] = (kb0 + j < L && (!mask || mask[row0 + kb0 + j] != 0)) ? 0.f : -INFINITY;
    __syncthreads();

    if (wave_active) {  // wave-uniform: a wave without queries (short sequence in a wide workgroup) only helps staging
    f32x16 sacc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) sacc[t] = (f32x16){0};
    const int ksw = (i >> 1) & 7;
    // Tiles in PAIRS: the two accumulators of a pair alternate, so no MFMA waits for the one issued just before it (a
    // dependent 32x32x16 MFMA cannot issue back to back); every accumulator still sees lo.hi, hi.lo, hi.hi for m = 0 .. 3 in
    // that order, so the sums keep their bits.
#pragma unroll
    for (int t = 0; t < NKT; t += 2) {
        if (t >= nt_live) break;
        const bool two = t + 1 < nt_live;  // block-uniform
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ko = (32 * t + i) * 128 + (((2 * m + kg) ^ ksw) * 16);
            const bf16x8_t kh0 = *reinterpret_cast<const bf16x8_t*>(Kh + ko);
            const bf16x8_t kl0 = *reinterpret_cast<const bf16x8_t*>(Kl + ko);
            if (two) {
                const bf16x8_t kh1 = *reinterpret_cast<const bf16x8_t*>(Kh + ko + 32 * 128);
                const bf16x8_t kl1 = *reinterpret_cast<const bf16x8_t*>(Kl + ko + 32 * 128);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh[m], sacc[t], 0, 0, 0);
                sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qh[m], sacc[t + 1], 0, 0, 0);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh0, ql[m], sacc[t], 0, 0, 0);
                sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh1, ql[m], sacc[t + 1], 0, 0, 0);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh0, qh[m], sacc[t], 0, 0, 0);
                sacc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh1, qh[m], sacc[t + 1], 0, 0, 0);
            } else {
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qh[m], sacc[t], 0, 0, 0);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh0, ql[m], sacc[t], 0, 0, 0);
                sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh0, qh[m], sacc[t], 0, 0, 0);
            }
        }
    }
    // sacc[t][reg] = <k_key, q_query>, key = 32 t + (reg&3) + 8 (reg>>2) + 4 kg, query = this lane's column
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt_live) break;
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
            // registers 4 a4 .. 4 a4 + 3 hold four CONSECUTIVE keys: their additive mask entries are one 16-byte LDS read
            const int key0 = 32 * t + 8 * a4 + 4 * kg;
            const float4 am = *reinterpret_cast<const float4*>(addm + key0);
            const float amv[4] = {am.x, am.y, am.z, am.w};
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const int reg = 4 * a4 + b4;
                float sv = sacc[t][reg] * scale2 + amv[b4];   // scores in units of log2 e: the exponentials below are v_exp_f32
                if (causal && kb0 + key0 + b4 > qrow) sv = -INFINITY;
                sacc[t][reg] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    }
    mx = fmaxf(fmaxf(mx, __shfl_xor(mx, 32)), m_run);
    // a masked score is -inf and 2^(-inf - max) is exactly 0; only a row with NO live key so far (max = -inf) needs a guard,
    // taken once per lane instead of once per score: subtracting 0 keeps every -inf, hence every probability 0 as before
    const float mxs = (mx == -INFINITY) ? 0.f : mx;
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt_live) break;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float p = __builtin_amdgcn_exp2f(sacc[t][reg] - mxs);
            sacc[t][reg] = p;
            bsum += p;
        }
    }
    bsum += __shfl_xor(bsum, 32);
    if (kb0) {  // rescale what the earlier blocks accumulated to the new maximum
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - mx);
        den *= alpha;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { oacc0[reg] *= alpha; oacc1[reg] *= alpha; }
    }
    den += bsum;
    m_run = mx;

    const int vsw0 = i & (NCH - 1);  // d = i and d = 32 + i swizzle alike when NCH <= 32
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t >= nt_live) break;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // keys 32 t + 16 g + 8 kg + (0..7): k 0..3 sit in the lower lane's registers 8g+4kg+r, k 4..7 in the upper's
            float p8[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // keys +0..3 are held by the lower lane (its registers 8g + r), keys +4..7 by the upper lane (8g + 4 + r):
                // v_permlane32_swap exchanges the upper half of the first operand with the lower half of the second, which is
                // exactly this regrouping (one VALU instruction instead of select + ds_bpermute + two selects)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[t][8 * g + r]),
                                                                 __float_as_uint(sacc[t][8 * g + 4 + r]), false, false);
                p8[r] = __uint_as_float(sw[0]);
                p8[4 + r] = __uint_as_float(sw[1]);
            }
            bf16x8_t ph, pl;
            split8(make_float4(p8[0], p8[1], p8[2], p8[3]), make_float4(p8[4], p8[5], p8[6], p8[7]), ph, pl);
            const int ck = 4 * t + 2 * g + kg;
            {   // both 32-row halves of d, their accumulators alternating (see the note at the S products)
                const int vo0 = i * (NK * 2) + ((ck ^ vsw0) * 16), vo1 = vo0 + 32 * (NK * 2);
                const bf16x8_t vh0 = *reinterpret_cast<const bf16x8_t*>(Vth + vo0);
                const bf16x8_t vl0 = *reinterpret_cast<const bf16x8_t*>(Vtl + vo0);
                const bf16x8_t vh1 = *reinterpret_cast<const bf16x8_t*>(Vth + vo1);
                const bf16x8_t vl1 = *reinterpret_cast<const bf16x8_t*>(Vtl + vo1);
                oacc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0, ph, oacc0, 0, 0, 0);
                oacc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1, ph, oacc1, 0, 0, 0);
                oacc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0, pl, oacc0, 0, 0, 0);
                oacc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1, pl, oacc1, 0, 0, 0);
                oacc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh0, ph, oacc0, 0, 0, 0);
                oacc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh1, ph, oacc1, 0, 0, 0);
            }
        }
    }
    }  // wave_active
    }  // key blocks
    // oacc{0,1}[reg] = sum_key p V[key][d], d = 32 dt + (reg&3) + 8 (reg>>2) + 4 kg, for this lane's query.
    // Transpose through LDS (K/V are dead) so that every query row is stored as 256 contiguous bytes.
    __syncthreads();
    float* Ot = reinterpret_cast<float*>(smem) + w * (32 * 65);   // [32 queries][65]
    const float inv = 1.0f / den;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int d = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
        Ot[i * 65 + d] = oacc0[reg] * inv;
        Ot[i * 65 + 32 + d] = oacc1[reg] * inv;
    }
    __syncthreads();
    // a lane takes EIGHT consecutive d of one query row: two 16-byte stores for the fp32 output, one 16-byte store each for
    // the hi and the lo halves of the split output (round 2 stored 4 bytes per lane and instruction: 32 stores and 32 DPP /
    // pack sequences per lane instead of 4 + 4)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3), c0 = 8 * (lane & 7);
        const int qr = blockIdx.y * qpb + 32 * w + r;
        const size_t at = (row0 + qr) * H + h * DH + c0;
        const float* src = Ot + r * 65 + c0;
        const float4 u = make_float4(src[0], src[1], src[2], src[3]), v = make_float4(src[4], src[5], src[6], src[7]);
        if (qr < L) {
            if (out) {
                *reinterpret_cast<float4*>(out + at) = u;
                *reinterpret_cast<float4*>(out + at + 4) = v;
            }
            if (out_h) {
                bf16x8_t hi8, lo8;
                split8(u, v, hi8, lo8);
                const size_t pt = pair_index(row0 + qr, h * DH + c0, H);
                *reinterpret_cast<uint4*>(out_h + pt) = __builtin_bit_cast(uint4, hi8);
                *reinterpret_cast<uint4*>(out_l + pt) = __builtin_bit_cast(uint4, lo8);
            }
        }
    }
}

// out[g] = init[g] + x[g][0] + x[g][1] + ... (IntermediateLinearFusion: the face embeddings of one example are summed
// into the projected text vector, meerqat/models/mm.py:838-843)
__global__ __launch_bounds__(256) void sum_groups_kernel(const float* __restrict__ x, const float* __restrict__ init,
                                                         float* __restrict__ out, int G, int n, int H) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)G * H) return;
    const size_t g = e / H, c = e % H;
    float acc = init ? init[e] : 0.f;
    for (int j = 0; j < n; ++j) acc += x[(g * n + j) * H + c];
    out[e] = acc;
}

// out[e] = bias[e % N] + part[0][e] + part[1][e] + ... (the partial sums of a split-K GEMM, in split order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            float* __restrict__ out, int S, size_t mn, int N) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= mn) return;
    float acc = part[e];
    for (int s2 = 1; s2 < S; ++s2) acc += part[(size_t)s2 * mn + e];
    out[e] = bias ? acc + bias[e % N] : acc;
}

// CLIP text tower input: token embedding + position embedding (no LayerNorm, no token types)
__global__ __launch_bounds__(256) void clip_text_embed_kernel(const long long* __restrict__ ids, const float* __restrict__ tok,
                                                              const float* __restrict__ pos, float* __restrict__ out, int M,
                                                              int L, int H, const int* __restrict__ pos_ids) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const long long id = ids[row];
    const int t_pos = pos_ids ? pos_ids[row] : row % L;  // packed token matrix: the position travels with the token
    for (int c = lane; c < H; c += 64) out[(size_t)row * H + c] = tok[(size_t)id * H + c] + pos[(size_t)t_pos * H + c];
}

// CLIP text pooling: the hidden state at the end-of-text token, through the final LayerNorm (LayerNorm is row-wise,
// so normalising only the pooled row equals HF's "normalise everything, then gather").  eos_token_id == 2 is the
// legacy config of the published checkpoints: the EOT token is the LARGEST id of the sequence (first occurrence);
// otherwise the first position holding eos_token_id (position 0 if there is none, like torch's argmax of zeros).
__global__ __launch_bounds__(256) void clip_eos_pool_ln_kernel(const float* __restrict__ X, const long long* __restrict__ ids,
                                                               long long eos, const float* __restrict__ g,
                                                               const float* __restrict__ b, float* __restrict__ out, int B,
                                                               int L, int H, float eps) {
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (seq >= B) return;
    long long best = -0x7fffffffffffffffLL - 1;
    int at = 0x7fffffff;
    for (int j = lane; j < L; j += 64) {
        const long long id = ids[(size_t)seq * L + j];
        const long long val = (eos == 2) ? (long long)(int)id : (long long)(id == eos);   // HF casts the ids to int32
        if (val > best) { best = val; at = j; }   // ascending j per lane: keeps the first occurrence
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(at, o);
        if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    const size_t row = (size_t)seq * L + at;
    float v[LN_MAXPER];
#pragma unroll
    for (int t = 0; t < LN_MAXPER; ++t) { const int c = lane + 64 * t; v[t] = (c < H) ? X[row * H + c] : 0.f; }
    ln_store(v, H, lane, g, b, eps, out + (size_t)seq * H);
}

// Workgroups of a split-bf16 GEMM launch: PERSISTENT by default -- one workgroup per CU (a multiple of 8) walks the tiles and
// requests the next tile's first K stage during the current tile's last step; MQ_GEMM_WGS=n overrides the count, 0 = one
// workgroup per output tile.  Round 2 measured the two launches equal (976 vs 975, 740 vs 731, 949 vs 941, 1093 vs 1092 executed
// TFLOP/s) and kept the plain one; with the operands stored tile by tile (round 3) the persistent launch is ahead on every
// shape (tools/bench_gemm_shapes.py: QKV 1.982 -> 1.972 ms, out-proj 0.725 -> 0.699, FFN1 2.616 -> 2.586, FFN2 2.357 -> 2.324) and
// on whole forwards (DPR 2048 x 100 104.0 -> 102.4 ms, pad-to-256 139.2 -> 137.8 ms).
// MQ_GEMM_OPT_WIDE (mq_gemm_set_option; initial value from MQ_GEMM_WIDE in the environment, read once): 1 = the eight-wave
// 128 x 64 kernel (gemm_x3w.inc, the default), 0 = gemm_nt_x3s_kernel (16 waves of 64 x 64).  Bit-identical results either way.
std::atomic<int> g_gemm_opt[MQ_GEMM_OPT_COUNT];
std::once_flag g_gemm_opt_once;
void gemm_opt_init() {
    std::call_once(g_gemm_opt_once, [] {
        const char* e = getenv("MQ_GEMM_WIDE");
        g_gemm_opt[MQ_GEMM_OPT_WIDE].store(e ? (atoi(e) != 0) : 1);
        e = getenv("MQ_GEMM_STAGGER");
        g_gemm_opt[MQ_GEMM_OPT_STAGGER].store(e ? atoi(e) : 0);
    });
}

int gemm_persistent_wgs() {
    static const int wgs = [] {
        const char* e = getenv("MQ_GEMM_WGS");
        if (e) return atoi(e) / 8 * 8;
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n / 8 * 8;
    }();
    return wgs;
}

}  // namespace

extern "C" {

int mq_gemm_nt_f32(const float* A_dev, const float* W_dev, const float* bias_dev, const float* residual_dev, float* C_dev,
                   int M, int N, int K, int epilogue, void* stream) {
    if (M == 0 || N == 0) return MQ_OK;
    if (!A_dev || !W_dev || !C_dev || M < 0 || N < 0 || K <= 0 || (K % GBK) != 0) return MQ_EINVAL;
    if (epilogue < EPI_NONE || epilogue > EPI_BIAS_RESIDUAL) return MQ_EINVAL;
    if (epilogue != EPI_NONE && !bias_dev) return MQ_EINVAL;
    if (epilogue == EPI_BIAS_RESIDUAL && !residual_dev) return MQ_EINVAL;
    if (((uintptr_t)A_dev | (uintptr_t)W_dev) & 15) return MQ_EINVAL;
    const int ntm = (M + GT - 1) / GT, ntn = (N + GT - 1) / GT;
    const dim3 grid((unsigned)(ntm * ntn)), block(1024);
    hipStream_t st = (hipStream_t)stream;
#define MQ_LAUNCH(E)                                                                                                  \
    case E:                                                                                                           \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, G_LDS_BYTES, gemm_nt_kernel<E>); \
        hipLaunchKernelGGL(gemm_nt_kernel<E>, grid, block, G_LDS_BYTES, st, A_dev, W_dev, bias_dev, residual_dev, C_dev, M, N, K, ntm, ntn); \
        break;
    switch (epilogue) {
        MQ_LAUNCH(EPI_NONE)
        MQ_LAUNCH(EPI_BIAS)
        MQ_LAUNCH(EPI_BIAS_GELU)
        MQ_LAUNCH(EPI_BIAS_QUICKGELU)
        MQ_LAUNCH(EPI_BIAS_RESIDUAL)
    }
#undef MQ_LAUNCH
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_split_bf16_f32(const float* src_dev, int64_t n, uint16_t* hi_dev, uint16_t* lo_dev, void* stream) {
    if (n == 0) return MQ_OK;
    if (!src_dev || !hi_dev || !lo_dev || n < 0) return MQ_EINVAL;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_dev, (long long)n,
                       (unsigned short*)hi_dev, (unsigned short*)lo_dev);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int64_t mq_split_bf16_tiled_elems(int N, int K) { return (int64_t)((N + GT - 1) / GT) * GT * (int64_t)K; }

int mq_split_bf16_tiled_f32(const float* W_dev, int N, int K, uint16_t* hi_dev, uint16_t* lo_dev, void* stream) {
    if (N == 0) return MQ_OK;
    if (!W_dev || !hi_dev || !lo_dev || N < 0 || K <= 0 || (K % XBK) != 0) return MQ_EINVAL;
    if (((uintptr_t)hi_dev | (uintptr_t)lo_dev) & 15) return MQ_EINVAL;
    const long long quads = (long long)mq_split_bf16_tiled_elems(N, K) / 4;
    hipLaunchKernelGGL(split_bf16_tiled_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W_dev, N, K,
                       (unsigned short*)hi_dev, (unsigned short*)lo_dev);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_gemm_nt_bf16x3_f32(const float* A_dev, const uint16_t* Wh_dev, const uint16_t* Wl_dev, const float* bias_dev,
                          const float* residual_dev, float* C_dev, int M, int N, int K, int epilogue, void* stream) {
    const int wt = (epilogue & MQ_GEMM_W_TILED) ? 1 : 0;
    epilogue &= ~MQ_GEMM_W_TILED;
    if (M == 0 || N == 0) return MQ_OK;
    if (!A_dev || !Wh_dev || !Wl_dev || !C_dev || M < 0 || N < 0 || K <= 0 || (K % XBK) != 0) return MQ_EINVAL;
    if (epilogue < EPI_NONE || epilogue > EPI_BIAS_RESIDUAL) return MQ_EINVAL;
    if (epilogue != EPI_NONE && !bias_dev) return MQ_EINVAL;
    if (epilogue == EPI_BIAS_RESIDUAL && !residual_dev) return MQ_EINVAL;
    if (((uintptr_t)A_dev | (uintptr_t)Wh_dev | (uintptr_t)Wl_dev) & 15) return MQ_EINVAL;
    const int ntm = (M + GT - 1) / GT, ntn = (N + GT - 1) / GT;
    const dim3 grid((unsigned)(((ntm + 7) & ~7) * ntn)), block(1024);  // tile space padded to 8 row blocks (XCD placement)
    hipStream_t st = (hipStream_t)stream;
#define MQ_LAUNCH(E)                                                                                                  \
    case E:                                                                                                           \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, X_LDS_BYTES, gemm_nt_x3_kernel<E>); \
        hipLaunchKernelGGL(gemm_nt_x3_kernel<E>, grid, block, X_LDS_BYTES, st, A_dev, (const unsigned short*)Wh_dev,    \
                           (const unsigned short*)Wl_dev, bias_dev, residual_dev, C_dev, M, N, K, ntm, ntn, wt);       \
        break;
    switch (epilogue) {
        MQ_LAUNCH(EPI_NONE)
        MQ_LAUNCH(EPI_BIAS)
        MQ_LAUNCH(EPI_BIAS_GELU)
        MQ_LAUNCH(EPI_BIAS_QUICKGELU)
        MQ_LAUNCH(EPI_BIAS_RESIDUAL)
    }
#undef MQ_LAUNCH
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

static int gemm_x3s_launch(const uint16_t* Ah_dev, const uint16_t* Al_dev, const uint16_t* Wh_dev, const uint16_t* Wl_dev,
                           const float* bias_dev, const float* residual_dev, float* C_dev, uint16_t* Ch_dev, uint16_t* Cl_dev,
                           int M, int N, int K, int epilogue, int nsplit, void* stream, const uint16_t* residual_lo_dev = nullptr) {
    const int wt = (epilogue & MQ_GEMM_W_TILED) ? 1 : 0;
    epilogue &= ~MQ_GEMM_W_TILED;
    if (M == 0 || N == 0) return MQ_OK;
    if (!Ah_dev || !Al_dev || !Wh_dev || !Wl_dev || M < 0 || N < 0 || K <= 0 || (K % XBK) != 0) return MQ_EINVAL;
    if ((!C_dev) == (!Ch_dev) || (!Ch_dev != !Cl_dev)) return MQ_EINVAL;  // exactly one output form
    if (epilogue < EPI_NONE || epilogue > EPI_BIAS_RESIDUAL) return MQ_EINVAL;
    if (epilogue != EPI_NONE && !bias_dev) return MQ_EINVAL;
    if (epilogue == EPI_BIAS_RESIDUAL && !residual_dev) return MQ_EINVAL;
    if (((uintptr_t)Ah_dev | (uintptr_t)Al_dev | (uintptr_t)Wh_dev | (uintptr_t)Wl_dev) & 15) return MQ_EINVAL;
    if (Ch_dev && (N & 31)) return MQ_EUNSUPPORTED;  // pair layout: 32-column tiles
    const int ntm = (M + GT - 1) / GT, ntn = (N + GT - 1) / GT;
    // MQ_GEMM_WGS=n: persistent launch, n workgroups walk the tiles (default: one workgroup per tile)
    const int persist = gemm_persistent_wgs();
    const int ntiles = ((ntm + 7) & ~7) * ntn;  // tile space padded to 8 row blocks (XCD placement, see the kernel)
    const dim3 grid((unsigned)(persist > 0 && ntiles > persist ? persist : ntiles), (unsigned)nsplit), block(1024);
    hipStream_t st = (hipStream_t)stream;
    gemm_opt_init();
    const bool wide = g_gemm_opt[MQ_GEMM_OPT_WIDE].load() != 0 && (N & 3) == 0;  // its epilogue moves four consecutive columns per lane
    const dim3 block_w(512);
#define MQ_LAUNCHW(E, S, P)                                                                                            \
    {                                                                                                                 \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, XW_LDS_BYTES, gemm_nt_x3w_kernel<E, S, P>); \
        hipLaunchKernelGGL((gemm_nt_x3w_kernel<E, S, P>), grid, block_w, XW_LDS_BYTES, st, (const unsigned short*)Ah_dev,  \
                           (const unsigned short*)Al_dev, (const unsigned short*)Wh_dev, (const unsigned short*)Wl_dev, \
                           bias_dev, residual_dev, C_dev, (unsigned short*)Ch_dev, (unsigned short*)Cl_dev, M, N, K, ntm, ntn, wt, nsplit, \
                           (const unsigned short*)residual_lo_dev, g_gemm_opt[MQ_GEMM_OPT_STAGGER].load()); \
    }
#define MQ_LAUNCH2(E, S)                                                                                              \
    if (wide && E == EPI_BIAS_RESIDUAL && residual_lo_dev) {                                                          \
        MQ_LAUNCHW(E, S, (E == EPI_BIAS_RESIDUAL))                                                                     \
    } else if (wide) {                                                                                                \
        MQ_LAUNCHW(E, S, false)                                                                                        \
    } else {                                                                                                          \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, XS_LDS_BYTES, gemm_nt_x3s_kernel<E, S>); \
        hipLaunchKernelGGL((gemm_nt_x3s_kernel<E, S>), grid, block, XS_LDS_BYTES, st, (const unsigned short*)Ah_dev,    \
                           (const unsigned short*)Al_dev, (const unsigned short*)Wh_dev, (const unsigned short*)Wl_dev, \
                           bias_dev, residual_dev, C_dev, (unsigned short*)Ch_dev, (unsigned short*)Cl_dev, M, N, K, ntm, ntn, wt, nsplit, \
                           (const unsigned short*)residual_lo_dev); \
    }
#define MQ_LAUNCH(E)                                                                                                  \
    case E:                                                                                                           \
        if (Ch_dev) MQ_LAUNCH2(E, true) else MQ_LAUNCH2(E, false)                                                     \
        break;
    switch (epilogue) {
        MQ_LAUNCH(EPI_NONE)
        MQ_LAUNCH(EPI_BIAS)
        MQ_LAUNCH(EPI_BIAS_GELU)
        MQ_LAUNCH(EPI_BIAS_QUICKGELU)
        MQ_LAUNCH(EPI_BIAS_RESIDUAL)
    }
#undef MQ_LAUNCH
#undef MQ_LAUNCH2
#undef MQ_LAUNCHW
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_gemm_set_option(int which, int value) {
    if (which < 0 || which >= MQ_GEMM_OPT_COUNT) return MQ_EINVAL;
    gemm_opt_init();
    return g_gemm_opt[which].exchange(value);
}

int mq_gemm_nt_bf16x3s_f32(const uint16_t* Ah_dev, const uint16_t* Al_dev, const uint16_t* Wh_dev, const uint16_t* Wl_dev,
                           const float* bias_dev, const float* residual_dev, float* C_dev, uint16_t* Ch_dev, uint16_t* Cl_dev,
                           int M, int N, int K, int epilogue, void* stream) {
    return gemm_x3s_launch(Ah_dev, Al_dev, Wh_dev, Wl_dev, bias_dev, residual_dev, C_dev, Ch_dev, Cl_dev, M, N, K, epilogue, 1, stream);
}

int mq_gemm_nt_bf16x3s_respair_f32(const uint16_t* Ah_dev, const uint16_t* Al_dev, const uint16_t* Wh_dev, const uint16_t* Wl_dev,
                                   const float* bias_dev, const uint16_t* Rh_dev, const uint16_t* Rl_dev, float* C_dev, uint16_t* Ch_dev,
                                   uint16_t* Cl_dev, int M, int N, int K, int epilogue, void* stream) {
    if ((epilogue & ~MQ_GEMM_W_TILED) != EPI_BIAS_RESIDUAL || !Rh_dev || !Rl_dev || (N & 31)) return MQ_EINVAL;
    if (((uintptr_t)Rh_dev | (uintptr_t)Rl_dev) & 15) return MQ_EINVAL;
    return gemm_x3s_launch(Ah_dev, Al_dev, Wh_dev, Wl_dev, bias_dev, reinterpret_cast<const float*>(Rh_dev), C_dev, Ch_dev, Cl_dev, M, N, K,
                           epilogue, 1, stream, Rl_dev);
}

int mq_gemm_nt_bf16x3s_splitk_f32(const uint16_t* Ah_dev, const uint16_t* Al_dev, const uint16_t* Wh_dev, const uint16_t* Wl_dev,
                                  const float* bias_dev, float* C_dev, int M, int N, int K, int w_tiled, int nsplit,
                                  float* partials_dev, void* stream) {
    if (M == 0 || N == 0) return MQ_OK;
    if (!C_dev || !partials_dev || nsplit < 1 || K <= 0 || (K % XBK) != 0 || M < 0 || N < 0) return MQ_EINVAL;
    const int nk = K / XBK;
    if (nsplit > nk) nsplit = nk;
    const int per = (nk + nsplit - 1) / nsplit;
    nsplit = (nk + per - 1) / per;  // no empty split
    const int rc = gemm_x3s_launch(Ah_dev, Al_dev, Wh_dev, Wl_dev, nullptr, nullptr, partials_dev, nullptr, nullptr, M, N, K,
                                   EPI_NONE | (w_tiled ? MQ_GEMM_W_TILED : 0), nsplit, stream);
    if (rc != MQ_OK) return rc;
    const size_t mn = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials_dev, bias_dev,
                       C_dev, nsplit, mn, N);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_layernorm_f32(const float* X_dev, const float* gamma_dev, const float* beta_dev, float* Y_dev, int M, int C, float eps,
                     void* stream) {
    if (!Y_dev) return M == 0 ? MQ_OK : MQ_EINVAL;
    return mq_layernorm_split_f32(X_dev, gamma_dev, beta_dev, Y_dev, nullptr, nullptr, M, C, eps, stream);
}

int mq_layernorm_split_f32(const float* X_dev, const float* gamma_dev, const float* beta_dev, float* Y_dev, uint16_t* Yh_dev,
                           uint16_t* Yl_dev, int M, int C, float eps, void* stream) {
    if (M == 0) return MQ_OK;
    if (!X_dev || !gamma_dev || !beta_dev || M < 0 || C <= 0) return MQ_EINVAL;
    if ((!Y_dev && !Yh_dev) || (!Yh_dev != !Yl_dev)) return MQ_EINVAL;
    if (C > 64 * LN_MAXPER || (Yh_dev && (C & 31))) return MQ_EUNSUPPORTED;
    static const bool via_lds = [] { const char* e = getenv("MQ_LN_PAIR_VIA_LDS"); return !e || atoi(e) != 0; }();
    if (Yh_dev && via_lds)
        hipLaunchKernelGGL(layernorm_pairs_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X_dev, gamma_dev,
                           beta_dev, Y_dev, (unsigned short*)Yh_dev, (unsigned short*)Yl_dev, M, C, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X_dev, gamma_dev,
                           beta_dev, Y_dev, (unsigned short*)Yh_dev, (unsigned short*)Yl_dev, M, C, eps);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_bert_embed_ln_f32(const int64_t* input_ids_dev, const int64_t* token_type_ids_dev, const float* word_dev,
                         const float* pos_dev, const float* type_dev, const float* gamma_dev, const float* beta_dev,
                         float* out_dev, int B, int L, int H, float eps, void* stream) {
    return mq_bert_embed_ln_split_f32(input_ids_dev, token_type_ids_dev, word_dev, pos_dev, type_dev, gamma_dev, beta_dev, out_dev,
                                      nullptr, nullptr, B, L, H, eps, stream);
}

int mq_bert_embed_ln_split_f32(const int64_t* input_ids_dev, const int64_t* token_type_ids_dev, const float* word_dev,
                               const float* pos_dev, const float* type_dev, const float* gamma_dev, const float* beta_dev,
                               float* out_dev, uint16_t* out_h_dev, uint16_t* out_l_dev, int B, int L, int H, float eps,
                               void* stream) {
    if (B == 0 || L == 0) return MQ_OK;
    if (!input_ids_dev || !word_dev || !pos_dev || !type_dev || !gamma_dev || !beta_dev || !out_dev || B < 0 || L < 0 || H <= 0)
        return MQ_EINVAL;
    if (!out_h_dev != !out_l_dev) return MQ_EINVAL;
    if (H > 64 * LN_MAXPER || (out_h_dev && (H & 31))) return MQ_EUNSUPPORTED;
    const int M = B * L;
    hipLaunchKernelGGL(bert_embed_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)input_ids_dev, (const long long*)token_type_ids_dev, word_dev, pos_dev, type_dev,
                       gamma_dev, beta_dev, out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, M, L, H, eps);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_bert_embed_ln_packed_f32(const int64_t* input_ids_dev, const int64_t* token_type_ids_dev, const int32_t* position_ids_dev,
                                const float* word_dev, const float* pos_dev, const float* type_dev, const float* gamma_dev,
                                const float* beta_dev, float* out_dev, uint16_t* out_h_dev, uint16_t* out_l_dev, int T, int H,
                                float eps, void* stream) {
    if (T == 0) return MQ_OK;
    if (!input_ids_dev || !position_ids_dev || !word_dev || !pos_dev || !type_dev || !gamma_dev || !beta_dev || !out_dev || T < 0 ||
        H <= 0)
        return MQ_EINVAL;
    if (!out_h_dev != !out_l_dev) return MQ_EINVAL;
    if (H > 64 * LN_MAXPER || (out_h_dev && (H & 31))) return MQ_EUNSUPPORTED;
    hipLaunchKernelGGL(bert_embed_ln_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)input_ids_dev, (const long long*)token_type_ids_dev, word_dev, pos_dev, type_dev,
                       gamma_dev, beta_dev, out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, T, 1, H, eps,
                       (const int*)position_ids_dev);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_attention_f32(const float* qkv_dev, const int64_t* attention_mask_dev, float* out_dev, int B, int L, int heads,
                     int head_dim, float scale, void* stream) {
    return mq_attention_causal_f32(qkv_dev, attention_mask_dev, out_dev, B, L, heads, head_dim, scale, 0, stream);
}

int mq_attention_causal_f32(const float* qkv_dev, const int64_t* attention_mask_dev, float* out_dev, int B, int L, int heads,
                            int head_dim, float scale, int causal, void* stream) {
    if (!out_dev) return (B == 0 || L == 0) ? MQ_OK : MQ_EINVAL;
    return mq_attention_split_f32(qkv_dev, attention_mask_dev, out_dev, nullptr, nullptr, B, L, heads, head_dim, scale, causal, 0,
                                  stream);
}

int mq_attention_split_f32(const float* qkv_dev, const int64_t* attention_mask_dev, float* out_dev, uint16_t* out_h_dev,
                           uint16_t* out_l_dev, int B, int L, int heads, int head_dim, float scale, int causal, int bf16x3,
                           void* stream) {
    if (B == 0 || L == 0) return MQ_OK;
    if (!qkv_dev || (!out_dev && !out_h_dev) || (!out_h_dev != !out_l_dev) || B < 0 || L < 0 || heads <= 0) return MQ_EINVAL;
    if (head_dim != DH) return MQ_EUNSUPPORTED;  // any L: sequences beyond 256 keys loop over key blocks
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(B * heads), (unsigned)((L + 127) / 128));
    const unsigned nthr = 64u * (unsigned)(L >= 97 ? 4 : (L + 31) / 32);  // one wave per 32 queries (short sequences: fewer waves)
#define MQ_ATT(NKT, MULTI)                                                                                                   \
    {                                                                                                                 \
        const size_t lds = (size_t)(32 * NKT) * (65 + 64 + 1) * 4 > (size_t)4 * 32 * 65 * 4 ? (size_t)(32 * NKT) * (65 + 64 + 1) * 4 : (size_t)4 * 32 * 65 * 4; \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, mq_detail::LDS_PER_CU, attention_mfma_kernel<NKT, MULTI>); \
        hipLaunchKernelGGL((attention_mfma_kernel<NKT, MULTI>), grid, dim3(nthr), lds, st, qkv_dev, (const long long*)attention_mask_dev,    \
                           out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, L, heads, scale, causal ? 1 : 0); \
    }
#define MQ_ATT3(NKT, MULTI)                                                                                                  \
    {                                                                                                                 \
        const size_t need = (size_t)(32 * NKT) * 128 * 2 + (size_t)64 * (32 * NKT) * 2 * 2 + (size_t)(32 * NKT) * 4;    \
        const size_t lds = need > (size_t)4 * 32 * 65 * 4 ? need : (size_t)4 * 32 * 65 * 4;                           \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, mq_detail::LDS_PER_CU, attention_x3_kernel<NKT, MULTI>); \
        hipLaunchKernelGGL((attention_x3_kernel<NKT, MULTI>), grid, dim3(nthr), lds, st, qkv_dev, (const long long*)attention_mask_dev,    \
                           out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, L, heads, scale, causal ? 1 : 0); \
    }
    if (bf16x3) {
        if (((uintptr_t)qkv_dev & 15) || (heads * DH) % 4) return MQ_EINVAL;  // 16-byte query loads
        if (L <= 64) MQ_ATT3(2, false)
        else if (L <= 128) MQ_ATT3(4, false)
        else if (L <= 256) MQ_ATT3(8, false)
        else MQ_ATT3(8, true)
    } else {
        if (L <= 64) MQ_ATT(2, false)
        else if (L <= 128) MQ_ATT(4, false)
        else if (L <= 256) MQ_ATT(8, false)
        else MQ_ATT(8, true)
    }
#undef MQ_ATT
#undef MQ_ATT3
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_attention_packed_f32(const float* qkv_dev, const int32_t* cu_seqlens_dev, const int32_t* seq_ids_dev, int n_seqs,
                            int max_len, float* out_dev, uint16_t* out_h_dev, uint16_t* out_l_dev, int heads, int head_dim,
                            float scale, int causal, int bf16x3, void* stream) {
    if (n_seqs == 0 || max_len == 0) return MQ_OK;
    if (!qkv_dev || !cu_seqlens_dev || !seq_ids_dev || (!out_dev && !out_h_dev) || (!out_h_dev != !out_l_dev) || n_seqs < 0 ||
        max_len < 0 || heads <= 0)
        return MQ_EINVAL;
    if (head_dim != DH) return MQ_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int L = max_len;
    const dim3 grid((unsigned)(n_seqs * heads), (unsigned)((L + 127) / 128));
    const unsigned nthr = 64u * (unsigned)(L >= 97 ? 4 : (L + 31) / 32);
#define MQ_ATTP(NKT, MULTI)                                                                                                  \
    {                                                                                                                 \
        const size_t lds = (size_t)(32 * NKT) * (65 + 64 + 1) * 4 > (size_t)4 * 32 * 65 * 4 ? (size_t)(32 * NKT) * (65 + 64 + 1) * 4 : (size_t)4 * 32 * 65 * 4; \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, mq_detail::LDS_PER_CU, attention_mfma_kernel<NKT, MULTI>); \
        hipLaunchKernelGGL((attention_mfma_kernel<NKT, MULTI>), grid, dim3(nthr), lds, st, qkv_dev, (const long long*)nullptr,    \
                           out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, L, heads, scale, causal ? 1 : 0, \
                           (const int*)cu_seqlens_dev, (const int*)seq_ids_dev); \
    }
#define MQ_ATTP3(NKT, MULTI)                                                                                                  \
    {                                                                                                                 \
        const size_t need = (size_t)(32 * NKT) * 128 * 2 + (size_t)64 * (32 * NKT) * 2 * 2 + (size_t)(32 * NKT) * 4;    \
        const size_t lds = need > (size_t)4 * 32 * 65 * 4 ? need : (size_t)4 * 32 * 65 * 4;                           \
        MQ_DYNAMIC_LDS_WITH(ENC_HIP, mq_detail::LDS_PER_CU, attention_x3_kernel<NKT, MULTI>); \
        hipLaunchKernelGGL((attention_x3_kernel<NKT, MULTI>), grid, dim3(nthr), lds, st, qkv_dev, (const long long*)nullptr,    \
                           out_dev, (unsigned short*)out_h_dev, (unsigned short*)out_l_dev, L, heads, scale, causal ? 1 : 0, \
                           (const int*)cu_seqlens_dev, (const int*)seq_ids_dev); \
    }
    if (bf16x3) {
        if (((uintptr_t)qkv_dev & 15) || (heads * DH) % 4) return MQ_EINVAL;
        if (L <= 64) MQ_ATTP3(2, false)
        else if (L <= 128) MQ_ATTP3(4, false)
        else if (L <= 256) {
            // one workgroup of ceil(L / 32) <= 8 waves per (sequence, head): K / V staged once for all its queries
            const size_t need = (size_t)256 * 128 * 2 + (size_t)64 * 256 * 2 * 2 + (size_t)256 * 4;
            const size_t ot = (size_t)8 * 32 * 65 * 4;
            const size_t lds = need > ot ? need : ot;
            MQ_DYNAMIC_LDS_WITH(ENC_HIP, mq_detail::LDS_PER_CU, attention_x3_kernel<8, false, true>);
            hipLaunchKernelGGL((attention_x3_kernel<8, false, true>), dim3((unsigned)(n_seqs * heads), 1u), dim3(64u * (unsigned)((L + 31) / 32)),
                               lds, st, qkv_dev, (const long long*)nullptr, out_dev, (unsigned short*)out_h_dev,
                               (unsigned short*)out_l_dev, L, heads, scale, causal ? 1 : 0, (const int*)cu_seqlens_dev,
                               (const int*)seq_ids_dev);
        }
        else MQ_ATTP3(8, true)
    } else {
        if (L <= 64) MQ_ATTP(2, false)
        else if (L <= 128) MQ_ATTP(4, false)
        else if (L <= 256) MQ_ATTP(8, false)
        else MQ_ATTP(8, true)
    }
#undef MQ_ATTP
#undef MQ_ATTP3
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_sum_groups_f32(const float* x_dev, const float* init_dev, float* out_dev, int G, int n, int H, void* stream) {
    if (G == 0 || H == 0) return MQ_OK;
    if (!x_dev || !out_dev || G < 0 || n < 0 || H < 0) return MQ_EINVAL;
    const size_t total = (size_t)G * H;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, init_dev,
                       out_dev, G, n, H);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_clip_text_embed_f32(const int64_t* input_ids_dev, const float* token_emb_dev, const float* pos_emb_dev, float* out_dev,
                           int B, int L, int H, void* stream) {
    if (B == 0 || L == 0) return MQ_OK;
    if (!input_ids_dev || !token_emb_dev || !pos_emb_dev || !out_dev || B < 0 || L < 0 || H <= 0) return MQ_EINVAL;
    const int M = B * L;
    hipLaunchKernelGGL(clip_text_embed_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)input_ids_dev, token_emb_dev, pos_emb_dev, out_dev, M, L, H, (const int*)nullptr);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_clip_text_embed_packed_f32(const int64_t* input_ids_dev, const int32_t* position_ids_dev, const float* token_emb_dev,
                                  const float* pos_emb_dev, float* out_dev, int T, int H, void* stream) {
    if (T == 0) return MQ_OK;
    if (!input_ids_dev || !position_ids_dev || !token_emb_dev || !pos_emb_dev || !out_dev || T < 0 || H <= 0) return MQ_EINVAL;
    hipLaunchKernelGGL(clip_text_embed_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)input_ids_dev, token_emb_dev, pos_emb_dev, out_dev, T, 1, H,
                       (const int*)position_ids_dev);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_clip_eos_pool_ln_f32(const float* hidden_dev, const int64_t* input_ids_dev, int64_t eos_token_id, const float* gamma_dev,
                            const float* beta_dev, float* out_dev, int B, int L, int H, float eps, void* stream) {
    if (B == 0) return MQ_OK;
    if (!hidden_dev || !input_ids_dev || !gamma_dev || !beta_dev || !out_dev || B < 0 || L <= 0 || H <= 0) return MQ_EINVAL;
    if (H > 64 * LN_MAXPER) return MQ_EUNSUPPORTED;
    hipLaunchKernelGGL(clip_eos_pool_ln_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, hidden_dev,
                       (const long long*)input_ids_dev, (long long)eos_token_id, gamma_dev, beta_dev, out_dev, B, L, H, eps);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_clip_patchify_f32(const float* pixels_dev, float* patches_dev, int B, int channels, int image_size, int patch_size,
                         void* stream) {
    if (B == 0) return MQ_OK;
    if (!pixels_dev || !patches_dev || B < 0 || channels <= 0 || image_size <= 0 || patch_size <= 0) return MQ_EINVAL;
    if (image_size % patch_size || patch_size % 4) return MQ_EUNSUPPORTED;
    const size_t total4 = (size_t)B * channels * image_size * image_size / 4;
    hipLaunchKernelGGL(clip_patchify_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pixels_dev,
                       patches_dev, B, channels, image_size, patch_size);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

int mq_clip_assemble_ln_f32(const float* patch_emb_dev, const float* class_emb_dev, const float* pos_emb_dev,
                            const float* gamma_dev, const float* beta_dev, float* out_dev, int B, int tokens, int H, float eps,
                            void* stream) {
    if (B == 0) return MQ_OK;
    if (!patch_emb_dev || !class_emb_dev || !pos_emb_dev || !gamma_dev || !beta_dev || !out_dev || B < 0 || tokens < 2 || H <= 0)
        return MQ_EINVAL;
    if (H > 64 * LN_MAXPER) return MQ_EUNSUPPORTED;
    const int M = B * tokens;
    hipLaunchKernelGGL(clip_assemble_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, patch_emb_dev,
                       class_emb_dev, pos_emb_dev, gamma_dev, beta_dev, out_dev, M, tokens, H, eps);
    ENC_HIP(hipGetLastError());
    return MQ_OK;
}

}  // extern "C"
