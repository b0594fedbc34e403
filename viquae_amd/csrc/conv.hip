// conv.hip -- the convolution side of the ArcFace r50 face encoder (meerqat/image/face_recognition.py:55-61: insightface
// arcface_torch `get_model('r50')` = IResNet-50) and its face alignment (:44-52: cv2.warpAffine to 112 x 112), for gfx950.
//
// A convolution runs here as  C[B*Ho*Wo, Cout] = A[B*Ho*Wo, KH*KW*Cin] . W[Cout, KH*KW*Cin]^T  on the split-bf16 GEMM of
// encoder.hip (mq_gemm_nt_bf16x3s_f32: three bf16 MFMA products per fp32 product, fp32-class accuracy, fused bias / residual
// epilogues), activations NHWC so that C IS the next layer's input.  This file makes A:
//   im2col_split_kernel   gathers the KH x KW x Cin patch of every output pixel from an fp32 activation, applies the layer's
//                         elementwise PRE-operations on the way -- PReLU (per-channel slope) of the producing layer, then the
//                         BatchNorm that stands IN FRONT of the convolution in an IBasicBlock (bn1: per-channel scale / shift,
//                         exact at the zero-padded border, where folding the shift into the convolution's bias is not) -- and
//                         writes the (hi, lo) bf16 pair in the GEMM's PAIR LAYOUT ([row / 256][col / 32][256][32]): the GEMM
//                         streams it by LDS-DMA and converts nothing.  BatchNorms BEHIND a convolution are folded into its
//                         weights and bias on the host (exact).
//   warp_affine_kernel    cv2.warpAffine(image, M, (112, 112), borderValue=0) -- OpenCV's fixed-point bilinear remap as
//                         published (INTER_BITS = 5, AB_BITS = 10, weights scaled to 2^15) -- fused with ToTensor + Normalize(0.5,
//                         0.5): uint8 H x W x 3 -> fp32 [3, 112, 112] in [-1, 1].
//   conv3x3_x3s_kernel    (round 4, second half) the 3 x 3 convolutions of the blocks as IMPLICIT GEMMs: the patch matrix is never
//                         written, the GEMM's LDS-DMA gathers the taps from the input's pair; PReLU / shortcut / the next BatchNorm in the
//                         epilogue.  The explicit path (im2col + GEMM: ~150 MB of patches per face, as long as the 12.6 GFLOP of matrix
//                         work) remains for the strided 1 x 1 downsamples and the head, and as the check of the implicit one.
//   stem_conv_kernel      the 3 -> 64 stem (K = 27) as a direct fp32 convolution fused with PReLU.
// Neither OpenCV, scikit-image nor arcface_torch is vendored by the reference or installable here: this path is "parity unpinned"
// (oracle/arcface.py restates the published algorithms).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/meerqat_hip.h"
#include "launch_attr.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// element (row, col) of a pair [M, K] (encoder.hip, PAIR LAYOUT)
__device__ __forceinline__ size_t pair_index(size_t row, int col, int K) {
    return ((row >> 8) * (size_t)(K >> 5) + (size_t)(col >> 5)) * (256 * 32) + ((row & 255) << 5) + (size_t)(col & 31);
}

__device__ __forceinline__ float pre_op(float x, int c, const float* __restrict__ slope, const float* __restrict__ scale,
                                        const float* __restrict__ shift) {
    if (slope) x = x >= 0.f ? x : x * slope[c];               // nn.PReLU(C)
    if (scale) x = __builtin_fmaf(x, scale[c], shift[c]);     // BatchNorm2d in eval mode: x * a + b (a, b precomputed on the host)
    return x;
}

// One workgroup = 64 rows x one 32-column block of A; thread t = columns 8 (t & 3) ... + 7 of row (t >> 2): a wave stores 16 rows x 64
// bytes = ONE contiguous KiB of each array of the pair (the pair layout keeps a [256 rows][32 columns] block contiguous), and the four
// lanes of a row read 128 contiguous bytes of one input pixel (NHWC, C a multiple of 8: the 8 columns are 8 channels of one tap).
// First version: one thread per 8 columns in row-major order of A -- 64-byte store segments 16 KiB apart and three 64-bit divisions
// per thread: 95 us per face of the 170 (rocprofv3); this one 58 us (7.7 k faces/s instead of 5.9 k).
__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ x, int B, int H, int W, int C, int nchw, int KH,
                                                           int KW, int stride, int pad, int Ho, int Wo, const float* __restrict__ slope,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           unsigned short* __restrict__ Ah, unsigned short* __restrict__ Al, int Kpad) {
    const unsigned M = (unsigned)B * Ho * Wo;
    const unsigned row = (blockIdx.z * 65535u + blockIdx.y) * 64u + (threadIdx.x >> 2);  // gridDim.y <= 65535: z carries the rest
    if (row >= M) return;
    const int col0 = (int)blockIdx.x * 32 + (int)(threadIdx.x & 3) * 8;
    const unsigned hw = (unsigned)Ho * Wo;
    const int b = (int)(row / hw);
    const int pix = (int)(row - (unsigned)b * hw);
    const int ho = pix / Wo, wo = pix - ho * Wo;
    const int Ktrue = KH * KW * C;
    float v[8];
    if (!nchw && (C & 7) == 0 && col0 + 8 <= Ktrue) {
        // the 8 columns are 8 consecutive channels of ONE tap
        const int tap = col0 / C, c0 = col0 - tap * C;
        const int kh = tap / KW, kw = tap - kh * KW;
        const int hi = ho * stride + kh - pad, wi = wo * stride + kw - pad;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
            const float4* p = reinterpret_cast<const float4*>(x + (((size_t)b * H + hi) * W + wi) * C + c0);
            const float4 u = p[0], w4 = p[1];
            const float t[8] = {u.x, u.y, u.z, u.w, w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pre_op(t[j], c0 + j, slope, scale, shift);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;  // zero padding comes AFTER the pre-operations (the conv pads its own input)
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = col0 + j;
            float t = 0.f;
            if (col < Ktrue) {
                const int tap = col / C, c = col - tap * C;
                const int kh = tap / KW, kw = tap - kh * KW;
                const int hi = ho * stride + kh - pad, wi = wo * stride + kw - pad;
                if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
                    const size_t at = nchw ? (((size_t)b * C + c) * H + hi) * W + wi : (((size_t)b * H + hi) * W + wi) * C + c;
                    t = pre_op(x[at], c, slope, scale, shift);
                }
            }
            v[j] = t;
        }
    }
    bf16x8_t h8, l8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        h8[j] = h;
        l8[j] = (__bf16)(v[j] - (float)h);
    }
    const size_t at = pair_index(row, col0, Kpad);
    *reinterpret_cast<bf16x8_t*>(Ah + at) = h8;
    *reinterpret_cast<bf16x8_t*>(Al + at) = l8;
}

// ---- 3 x 3 convolution as an IMPLICIT GEMM (round 4) ----------------------------------------------------------------------
// C[M = B Ho Wo, N = Cout] = A[M, 9 C] . W[N, 9 C]^T with A never written: the K step (tap (kh, kw), 32 channels) of an output pixel is
// the 64 contiguous bytes [pixel (b, ho s + kh - 1, wo s + kw - 1)][32 channels] of the INPUT pair (PAIR LAYOUT over [B H W, C] keeps
// them contiguous), so the LDS-DMA that fills the GEMM's A stage takes a per-lane source address -- the input pixel, or a page of
// zeros at the padded border -- and everything behind it is gemm_nt_x3s_kernel (encoder.hip): 64-byte tile rows, 16-byte chunks
// XOR-swizzled on the source, hi / lo fragments, products lo.hi, hi.lo, hi.hi per 16 k on v_mfma_f32_32x32x16_bf16, K in (kh, kw,
// c) order -- the accumulation order of the explicit path, so the two give the same bits.
// The input pair holds the convolution's input AFTER its elementwise pre-operations (PReLU, the BatchNorm in front): they are
// applied by the kernel that PRODUCES the tensor -- this kernel's epilogues:
//   EP_PRELU_PAIR      v = prelu(acc + bias)                       -> pair [M, N]            (IBasicBlock.conv1 + bn2 + prelu)
//   EP_RESIDUAL_AFFINE v = acc + bias + R;  Y = v (fp32, the next shortcut);  pair = v * scale + shift (the next block's bn1; optional)
// Tiles: 16 waves, each 64 rows x WN columns; RG row groups x 16 / RG column groups: 256 x 256 (RG 4, WN 64), 512 x 128 (8, 64),
// 256 x 128 (4, 32), 512 x 64 (8, 32) -- the 64- / 128-channel stages no longer pay for 256-column tiles.
constexpr int EP_PRELU_PAIR = 0, EP_RESIDUAL_AFFINE = 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long)(lds_char*)p; }
// one 1-KiB LDS-DMA piece, LDS destination lds_dst (wave-uniform) + lane * 16: wave-uniform base + per-lane byte offset ...
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// ... or a 64-bit source address per lane (the gather)
__device__ __forceinline__ void dma16v(const void* lane_ptr, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_ptr), "s"(lds_dst) : "memory");
}

struct ConvArgs {
    const unsigned short *Xh, *Xl;  // input pair, PAIR LAYOUT over [B * H * W, C]
    const unsigned short *Wh, *Wl;  // weights, tile layout [N / 256][9 C / 32][256][32] (mq_split_bf16_tiled_f32), K = (kh, kw, c)
    const float* bias;              // [N]
    const float* slope;             // EP_PRELU_PAIR: PReLU slope [N]
    const float* R;                 // EP_RESIDUAL_AFFINE: residual fp32 [M, N]
    const float *scale, *shift;     // EP_RESIDUAL_AFFINE: affine of the pair output, or null = no pair output
    float* Y;                       // EP_RESIDUAL_AFFINE: fp32 [M, N]
    unsigned short *Ph, *Pl;        // pair output, PAIR LAYOUT over [M, N]
    const void* zeros;              // >= 16 bytes of zeros: the padded border
    int H, W, C, N, stride, Ho, Wo, M, ntm, ntn;
    int cmajor;  // K steps in (channel block, tap) order instead of (tap, channel block): see mq_conv3x3_pair_f32
};

template <int RG, int WN, int EP>
__global__ __launch_bounds__(1024) void conv3x3_x3s_kernel(const ConvArgs a) {
    constexpr int CG = 16 / RG, MT = 64 * RG, NT = WN * CG, NF = WN / 32, AP = MT / 256;
    constexpr int A_BYTES = MT * 64, W_BYTES = NT * 64, STAGE = 2 * A_BYTES + 2 * W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / CG, wc = w % CG;
    // workgroup b runs on XCD b & 7.  Each XCD takes a CONTIGUOUS range of row blocks (per = ceil(ntm / 8) of them) and all
    // column tiles of a row block: the tiles its CUs hold at one time are neighbours in the image, so the halo rows two row blocks
    // share (2 of a 512-pixel tile's 4.6 image rows at 112 x 112) and a row block's A rows are fetched into that XCD's L2 once
    const int b = (int)blockIdx.x;
    const int per = (a.ntm + 7) >> 3;
    const int mt = __builtin_amdgcn_readfirstlane((b & 7) * per + (b >> 3) / a.ntn);
    const int nt = __builtin_amdgcn_readfirstlane((b >> 3) % a.ntn);
    if ((b >> 3) / a.ntn >= per || mt >= a.ntm) return;
    const int m0 = mt * MT, n0 = nt * NT;
    const int CB = a.C >> 5, nk = 9 * CB;

    // ---- the gather: wave w fills rows [16 (w + 16 j), + 16) of the A stage, lane = (row, 16-byte slot)
    unsigned tapmask[AP];
    int pbase[AP];
    const unsigned slot = (unsigned)(lane & 3);
    unsigned chunk_bytes[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int r = 16 * (w + 16 * j) + (lane >> 2);
        const int m = m0 + r;
        const unsigned hw = (unsigned)(a.Ho * a.Wo);
        const unsigned mm = m < a.M ? (unsigned)m : 0u;
        const unsigned bi = mm / hw, pix = mm - bi * hw;
        const int ho = (int)(pix / (unsigned)a.Wo), wo = (int)(pix - (unsigned)ho * (unsigned)a.Wo);
        const int ih0 = ho * a.stride - 1, iw0 = wo * a.stride - 1;
        unsigned mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = ih0 + t / 3, iw = iw0 + t % 3;
            if (m < a.M && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) mask |= 1u << t;
        }
        tapmask[j] = mask;
        pbase[j] = ((int)bi * a.H + ih0) * a.W + iw0;
        chunk_bytes[j] = (slot ^ (unsigned)((r >> 2) & 3)) * 16u;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    // weights: piece w of NT / 16 (16 rows x 64 B), rows n0 % 256 + 16 w ... of the K step's 16-KiB block
    const int rW = 16 * w + (lane >> 2);
    const unsigned w_voff = (unsigned)(((n0 & 255) + rW) * 64) + (slot ^ (unsigned)((rW >> 2) & 3)) * 16u;
    const char* wh0 = reinterpret_cast<const char*>(a.Wh) + (size_t)(n0 >> 8) * (size_t)nk * 16384;
    const char* wl0 = reinterpret_cast<const char*>(a.Wl) + (size_t)(n0 >> 8) * (size_t)nk * 16384;
    const char* xh = reinterpret_cast<const char*>(a.Xh);
    const char* xl = reinterpret_cast<const char*>(a.Xl);
    const char* zp = reinterpret_cast<const char*>(a.zeros);

    int tap = 0, cb = 0;  // of the NEXT K step to request
    auto issue = [&](int kb, int stg) __attribute__((always_inline)) {
        const unsigned so = lds0 + (unsigned)(stg * STAGE);
        const int dp = (tap / 3) * a.W + (tap % 3);
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const bool ok = (tapmask[j] >> tap) & 1u;
            const unsigned p = (unsigned)(pbase[j] + dp);
            const size_t off = ((size_t)(p >> 8) * (size_t)CB + (size_t)cb) * 16384 + (size_t)((p & 255u) * 64u + chunk_bytes[j]);
            const unsigned dst = so + (unsigned)(16 * (w + 16 * j) * 64);
            dma16v(ok ? xh + off : zp, dst);
            dma16v(ok ? xl + off : zp, dst + A_BYTES);
        }
        if (w < NT / 16) {
            const unsigned dst = so + 2 * A_BYTES + (unsigned)(16 * w * 64);
            const size_t kw_ = (size_t)__builtin_amdgcn_readfirstlane(tap * CB + cb) * 16384;  // the weights' K order is (tap, channel) either way
            dma16s(wh0 + kw_, w_voff, dst);
            dma16s(wl0 + kw_, w_voff, dst + W_BYTES);
        }
        if (a.cmajor) { if (++tap == 9) { tap = 0; ++cb; } }
        else if (++cb == CB) { cb = 0; ++tap; }
    };

    const int i = lane & 31, kg = lane >> 5;
    const int sww = (i >> 2) & 3;
    const char* ard = smem + (64 * wr + i) * 64;                   // Ah rows of this wave; Al at + A_BYTES
    const char* wrd = smem + 2 * A_BYTES + (WN * wc + i) * 64;     // Wh rows of this wave; Wl at + W_BYTES
    f32x16 acc[2][NF];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NF; ++y) acc[x][y] = (f32x16){0};

    issue(0, 0);
    int stage = 0;
    for (int kb = 0; kb < nk; ++kb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kb + 1 < nk) issue(kb + 1, stage ^ 1);
        const char* as = ard + stage * STAGE;
        const char* ws = wrd + stage * STAGE;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int o = ((2 * m + kg) ^ sww) * 16;
            bf16x8_t ah[2], al[2], wh[NF], wl[NF];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ah[x] = *reinterpret_cast<const bf16x8_t*>(as + x * 32 * 64 + o);
                al[x] = *reinterpret_cast<const bf16x8_t*>(as + A_BYTES + x * 32 * 64 + o);
            }
#pragma unroll
            for (int y = 0; y < NF; ++y) {
                wh[y] = *reinterpret_cast<const bf16x8_t*>(ws + y * 32 * 64 + o);
                wl[y] = *reinterpret_cast<const bf16x8_t*>(ws + W_BYTES + y * 32 * 64 + o);
            }
            // per accumulator: lo.hi, hi.lo, hi.hi -- the order of gemm_nt_x3s_kernel
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NF; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[x], wh[y], acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NF; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[x], wl[y], acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NF; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[x], wh[y], acc[x][y], 0, 0, 0);
        }
        stage ^= 1;
    }

    // ---- epilogue: one 32 x 32 accumulator at a time through LDS (4 KiB per wave), so that a lane ends up with eight consecutive
    // columns of one row: 16-byte loads and stores.  Bias (and PReLU) in the accumulator layout (one column per lane); residual,
    // affine and split after the transposition: per element the operations of the explicit path, in its order.
    __syncthreads();  // every wave has read its last operands
    float* Ot = reinterpret_cast<float*>(smem) + w * 1024;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NF; ++y) {
            const int col = n0 + WN * wc + 32 * y + i;
            const float bs = a.bias[col];
            const float sl = EP == EP_PRELU_PAIR ? a.slope[col] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                float v = acc[x][y][reg] + bs;
                if (EP == EP_PRELU_PAIR) v = v >= 0.f ? v : v * sl;
                Ot[row * 32 + ((((i >> 3) ^ (row & 3)) << 3) | (i & 7))] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = 16 * it + (lane >> 2), g = lane & 3;
                const float* src = Ot + row * 32 + ((g ^ (row & 3)) << 3);
                float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
                const int gm = m0 + 64 * wr + 32 * x + row, gn = n0 + WN * wc + 32 * y + 8 * g;
                if (gm < a.M) {
                    bool pair_out = true;
                    if (EP == EP_RESIDUAL_AFFINE) {
                        const size_t at = (size_t)gm * a.N + gn;
                        const float4 ru = *reinterpret_cast<const float4*>(a.R + at), rv = *reinterpret_cast<const float4*>(a.R + at + 4);
                        u.x += ru.x; u.y += ru.y; u.z += ru.z; u.w += ru.w;
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        *reinterpret_cast<float4*>(a.Y + at) = u;
                        *reinterpret_cast<float4*>(a.Y + at + 4) = v;
                        pair_out = a.scale != nullptr;
                        if (pair_out) {
                            const float4 su = *reinterpret_cast<const float4*>(a.scale + gn), sv = *reinterpret_cast<const float4*>(a.scale + gn + 4);
                            const float4 tu = *reinterpret_cast<const float4*>(a.shift + gn), tv = *reinterpret_cast<const float4*>(a.shift + gn + 4);
                            u.x = __builtin_fmaf(u.x, su.x, tu.x); u.y = __builtin_fmaf(u.y, su.y, tu.y);
                            u.z = __builtin_fmaf(u.z, su.z, tu.z); u.w = __builtin_fmaf(u.w, su.w, tu.w);
                            v.x = __builtin_fmaf(v.x, sv.x, tv.x); v.y = __builtin_fmaf(v.y, sv.y, tv.y);
                            v.z = __builtin_fmaf(v.z, sv.z, tv.z); v.w = __builtin_fmaf(v.w, sv.w, tv.w);
                        }
                    }
                    if (pair_out) {
                        const float t[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
                        bf16x8_t h8, l8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const __bf16 h = (__bf16)t[e];
                            h8[e] = h;
                            l8[e] = (__bf16)(t[e] - (float)h);
                        }
                        const size_t pt = pair_index((size_t)gm, gn, a.N);
                        *reinterpret_cast<bf16x8_t*>(a.Ph + pt) = h8;
                        *reinterpret_cast<bf16x8_t*>(a.Pl + pt) = l8;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next accumulator overwrites the scratch
        }
}

// ---- stride-1 3 x 3 convolution with the taps served from ONE LDS-resident input patch (round 4, last) ---------------------------------
// conv3x3_x3s_kernel gathers, for each of the nine taps of a channel block, the tile's rows again -- nine passes of A through L2 -> LDS.
// With few output channels there is little work per pass: a 512 x 64 tile moves 73.7 KB per K step for 0.88 us of MFMA work, and
// the 64-channel stage runs at the rate those bytes arrive (750 / 580 executed TFLOP/s against 1216 at 14 x 14 x 256).  Here a tile
// is 256 CONSECUTIVE output pixels x 64 channels and the pixels its nine taps read are 256 + 2 W + 2 consecutive input pixels (stride
// 1: input pixel = output pixel + (kh - 1) W + (kw - 1)): that patch is loaded ONCE per 32-channel block (two LDS buffers, the next
// block's patch arriving while this one is used) and tap (kh, kw) reads its A fragments kh W + kw rows further down -- the 64-byte rows and
// their XOR swizzle (keyed by the patch row) make any row offset conflict-free.  Pixels a tap must not see (image borders: the patch is
// linear in the pixel index, so w = 0 / W - 1 wrap into the neighbouring image row) are zeroed in the fragment registers from a 9-bit
// mask per lane.  Weights: one 8-KiB stage per tap in a four-stage ring, requested three taps ahead; every wave's DMA count per
// step is constant (zero-page requests past the end), so the in-order vmcnt immediates are compile-time.  16 waves of 32 x 32.
// Same products in the same order as conv3x3_x3s_kernel with MQ_CONV_K_CHANNEL_MAJOR: the same bits.
constexpr int PT_MT = 256, PT_NT = 64, PT_PR = 512, PT_WS = 4;
constexpr int PT_PATCH = PT_PR * 128;            // one patch buffer: hi rows then lo rows, 64 bytes each
constexpr int PT_WSTAGE = PT_NT * 128;           // one weight stage: hi rows then lo rows
constexpr int PT_LDS = 2 * PT_PATCH + PT_WS * PT_WSTAGE;   // 160 KiB

template <int EP>
__global__ __launch_bounds__(1024) void conv3x3_patch_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int b = (int)blockIdx.x;
    const int per = (a.ntm + 7) >> 3;
    const int mt = __builtin_amdgcn_readfirstlane((b & 7) * per + (b >> 3) / a.ntn);
    const int nt = __builtin_amdgcn_readfirstlane((b >> 3) % a.ntn);
    if ((b >> 3) / a.ntn >= per || mt >= a.ntm) return;
    const int m0 = mt * PT_MT, n0 = nt * PT_NT;
    const int CB = a.C >> 5, nk = 9 * CB, W = a.W;
    const int P = a.M;                       // stride 1: as many input pixels as output pixels
    const int prows = PT_MT + 2 * W + 2;     // patch rows in use
    const int i = lane & 31, kg = lane >> 5;

    // this lane's output pixel (MFMA row i of the wave's 32 rows) and the taps it may read
    unsigned mask = 0;
    {
        const int m = m0 + 32 * wr + i;
        const unsigned hw = (unsigned)(a.H * W), mm = m < a.M ? (unsigned)m : 0u;
        const unsigned pix = mm % hw;
        const int h = (int)(pix / (unsigned)W), wq = (int)(pix - (unsigned)h * (unsigned)W);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = h + t / 3 - 1, iw = wq + t % 3 - 1;
            if (m < a.M && ih >= 0 && ih < a.H && iw >= 0 && iw < W) mask |= 1u << t;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const char* xh = reinterpret_cast<const char*>(a.Xh);
    const char* xl = reinterpret_cast<const char*>(a.Xl);
    const char* zp = reinterpret_cast<const char*>(a.zeros);
    // patch pieces of this wave: rows 16 (w + 16 j) + (lane >> 2), j = 0, 1; slot lane & 3 takes source chunk slot ^ key(row)
    auto issue_patch = [&](int cb, bool real, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = 16 * (w + 16 * j) + (lane >> 2);
            const int p = m0 - W - 1 + rho;
            const bool ok = real && rho < prows && p >= 0 && p < P;
            const unsigned pu = (unsigned)p;
            const size_t off = ((size_t)(pu >> 8) * (size_t)CB + (size_t)cb) * 16384 +
                               (size_t)((pu & 255u) * 64u + (((unsigned)(lane & 3) ^ (unsigned)((rho >> 2) & 3)) * 16u));
            const unsigned dst = lds0 + (unsigned)(buf * PT_PATCH + 16 * (w + 16 * j) * 64);
            dma16v(ok ? xh + off : zp, dst);
            dma16v(ok ? xl + off : zp, dst + PT_PR * 64);
        }
    };
    // weight piece of waves 0-7: array w >> 2 (hi, lo), rows 16 (w & 3) ... of the tap's 64-row stage
    const int wrow = 16 * (w & 3) + (lane >> 2);
    const unsigned w_voff = (unsigned)(((n0 & 255) + wrow) * 64) + (((unsigned)(lane & 3) ^ (unsigned)((wrow >> 2) & 3)) * 16u);
    const char* wbase = reinterpret_cast<const char*>((w >> 2) ? a.Wl : a.Wh) + (size_t)(n0 >> 8) * (size_t)nk * 16384;
    auto issue_w = [&](int step) __attribute__((always_inline)) {  // step = cb * 9 + tap; weights' K order is (tap, channel block)
        const unsigned dst = lds0 + (unsigned)(2 * PT_PATCH + (step & (PT_WS - 1)) * PT_WSTAGE + (w >> 2) * (PT_NT * 64) + (w & 3) * 1024);
        if (step < nk) {
            const int cb = step / 9, tap = step - cb * 9;
            dma16s(wbase + (size_t)__builtin_amdgcn_readfirstlane(tap * CB + cb) * 16384, w_voff, dst);
        } else {
            dma16v(zp, dst);
        }
    };

    f32x16 acc = {0};
    issue_patch(0, true, 0);
    if (w < 8) { issue_w(0); issue_w(1); issue_w(2); }
    const int sww = (i >> 2) & 3;
    const char* wrd0 = smem + 2 * PT_PATCH + (32 * wc + i) * 64;
    for (int cb = 0; cb < CB; ++cb) {
        const char* pbuf = smem + (cb & 1) * PT_PATCH;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = cb * 9 + tap;
            // in-order counter: everything up to this step's weight stage (and, at tap 0, this block's patch) has landed when at
            // most the younger requests are outstanding: two weight pieces, plus the next patch's four between taps 0 and 3
            if (w < 8) {
                if (tap == 1 || tap == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else if (tap == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tap == 0) issue_patch(cb + 1, cb + 1 < CB, (cb + 1) & 1);
            if (w < 8) issue_w(step + 3);
            const int off = (tap / 3) * W + (tap % 3);
            const int rho = 32 * wr + i + off;
            const int key = (rho >> 2) & 3;
            const char* ap = pbuf + rho * 64;
            const char* wp = wrd0 + (step & (PT_WS - 1)) * PT_WSTAGE;
            const bool ok = (mask >> tap) & 1u;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = 2 * m + kg;
                bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(ap + ((c ^ key) << 4));
                bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(ap + PT_PR * 64 + ((c ^ key) << 4));
                const bf16x8_t wh = *reinterpret_cast<const bf16x8_t*>(wp + ((c ^ sww) << 4));
                const bf16x8_t wl = *reinterpret_cast<const bf16x8_t*>(wp + PT_NT * 64 + ((c ^ sww) << 4));
                if (!ok) {
                    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                    ah = __builtin_bit_cast(bf16x8_t, z);
                    al = __builtin_bit_cast(bf16x8_t, z);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero-page requests past the end still target LDS
    __syncthreads();

    // ---- epilogue: conv3x3_x3s_kernel's, for one 32 x 32 accumulator
    float* Ot = reinterpret_cast<float*>(smem) + w * 1024;
    {
        const int col = n0 + 32 * wc + i;
        const float bs = a.bias[col];
        const float sl = EP == EP_PRELU_PAIR ? a.slope[col] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kg;
            float v = acc[reg] + bs;
            if (EP == EP_PRELU_PAIR) v = v >= 0.f ? v : v * sl;
            Ot[row * 32 + ((((i >> 3) ^ (row & 3)) << 3) | (i & 7))] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = 16 * it + (lane >> 2), g = lane & 3;
            const float* src = Ot + row * 32 + ((g ^ (row & 3)) << 3);
            float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
            const int gm = m0 + 32 * wr + row, gn = n0 + 32 * wc + 8 * g;
            if (gm < a.M) {
                bool pair_out = true;
                if (EP == EP_RESIDUAL_AFFINE) {
                    const size_t at = (size_t)gm * a.N + gn;
                    const float4 ru = *reinterpret_cast<const float4*>(a.R + at), rv = *reinterpret_cast<const float4*>(a.R + at + 4);
                    u.x += ru.x; u.y += ru.y; u.z += ru.z; u.w += ru.w;
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    *reinterpret_cast<float4*>(a.Y + at) = u;
                    *reinterpret_cast<float4*>(a.Y + at + 4) = v;
                    pair_out = a.scale != nullptr;
                    if (pair_out) {
                        const float4 su = *reinterpret_cast<const float4*>(a.scale + gn), sv = *reinterpret_cast<const float4*>(a.scale + gn + 4);
                        const float4 tu = *reinterpret_cast<const float4*>(a.shift + gn), tv = *reinterpret_cast<const float4*>(a.shift + gn + 4);
                        u.x = __builtin_fmaf(u.x, su.x, tu.x); u.y = __builtin_fmaf(u.y, su.y, tu.y);
                        u.z = __builtin_fmaf(u.z, su.z, tu.z); u.w = __builtin_fmaf(u.w, su.w, tu.w);
                        v.x = __builtin_fmaf(v.x, sv.x, tv.x); v.y = __builtin_fmaf(v.y, sv.y, tv.y);
                        v.z = __builtin_fmaf(v.z, sv.z, tv.z); v.w = __builtin_fmaf(v.w, sv.w, tv.w);
                    }
                }
                if (pair_out) {
                    const float t[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
                    bf16x8_t h8, l8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 hb = (__bf16)t[e];
                        h8[e] = hb;
                        l8[e] = (__bf16)(t[e] - (float)hb);
                    }
                    const size_t pt = pair_index((size_t)gm, gn, a.N);
                    *reinterpret_cast<bf16x8_t*>(a.Ph + pt) = h8;
                    *reinterpret_cast<bf16x8_t*>(a.Pl + pt) = l8;
                }
            }
        }
    }
}

#define CONV_HIP(call)                                  \
    do {                                                \
        hipError_t _e = (call);                         \
        if (_e != hipSuccess) return MQ_EHIP;           \
    } while (0)

template <int RG, int WN, int EP>
int launch_conv3x3(ConvArgs a, hipStream_t st) {
    constexpr int CG = 16 / RG, MT = 64 * RG, NT = WN * CG;
    constexpr int LDS = 2 * (2 * MT * 64 + 2 * NT * 64);
    static_assert(LDS >= 16 * 4096, "the epilogue's scratch: 4 KiB per wave");
    a.ntm = (a.M + MT - 1) / MT;
    a.ntn = a.N / NT;
    const int ntiles = ((a.ntm + 7) >> 3) * 8 * a.ntn;
    MQ_DYNAMIC_LDS_WITH(CONV_HIP, LDS, conv3x3_x3s_kernel<RG, WN, EP>);
    hipLaunchKernelGGL((conv3x3_x3s_kernel<RG, WN, EP>), dim3((unsigned)ntiles), dim3(1024), LDS, st, a);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

template <int EP>
int launch_conv3x3_patch(ConvArgs a, hipStream_t st) {
    a.ntm = (a.M + PT_MT - 1) / PT_MT;
    a.ntn = a.N / PT_NT;
    const int ntiles = ((a.ntm + 7) >> 3) * 8 * a.ntn;
    MQ_DYNAMIC_LDS_WITH(CONV_HIP, PT_LDS, conv3x3_patch_kernel<EP>);
    hipLaunchKernelGGL((conv3x3_patch_kernel<EP>), dim3((unsigned)ntiles), dim3(1024), PT_LDS, st, a);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

template <int EP>
int launch_conv3x3_variant(const ConvArgs& a, int variant, hipStream_t st) {
    switch (variant) {
        case MQ_CONV_TILE_PATCH_256x64:
            // the patch is linear in the pixel index: stride 1, and 256 + 2 W + 2 rows must fit the 512-row buffer
            return (a.stride != 1 || a.N % 64 || 2 * a.W + 2 > PT_PR - PT_MT || !a.cmajor) ? MQ_EINVAL : launch_conv3x3_patch<EP>(a, st);
        case MQ_CONV_TILE_256x256: return a.N % 256 ? MQ_EINVAL : launch_conv3x3<4, 64, EP>(a, st);
        case MQ_CONV_TILE_512x128: return a.N % 128 ? MQ_EINVAL : launch_conv3x3<8, 64, EP>(a, st);
        case MQ_CONV_TILE_256x128: return a.N % 128 ? MQ_EINVAL : launch_conv3x3<4, 32, EP>(a, st);
        case MQ_CONV_TILE_512x64: return a.N % 64 ? MQ_EINVAL : launch_conv3x3<8, 32, EP>(a, st);
    }
    return MQ_EINVAL;
}

// ---- the stem: 3 -> 64 channels, 3 x 3, stride 1, padding 1 (K = 27: no matrix-pipe shape) -- direct fp32 convolution on the vector ALU.
// Through im2col + GEMM + the elementwise pass that makes the first block's input pair, the stem moved 4.2 GB per 328 faces for 43
// MFLOP per face (1.46 ms of a 15-ms forward); here the NCHW pixels are read once and only what the network reads next is written:
//   P  = split(prelu(conv + bias) * scale + shift)   the first block's conv1 input pair (bn1 applied), PAIR LAYOUT over [B H W, 64]
//   D  = split(prelu(conv + bias)) at even (h, w)    the A operand of the first block's strided 1 x 1 downsample, [B H/2 W/2, 64]
// A thread = 4 horizontally adjacent pixels x 8 output channels (32 accumulators; the 27 x 64 weights sit in LDS, one 32-byte read
// per tap serves 4 x 8 multiply-adds); fp32 fused multiply-adds in (kh, kw, c) order: fp32-exact products, one rounding per term.
__global__ __launch_bounds__(256, 4) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                        const float* __restrict__ slope, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, unsigned short* __restrict__ Ph,
                                                        unsigned short* __restrict__ Pl, unsigned short* __restrict__ Dh,
                                                        unsigned short* __restrict__ Dl, int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) float ws[27 * 64];
    for (int e = threadIdx.x; e < 27 * 64; e += 256) ws[e] = wt[e];
    __syncthreads();
    const int cg = threadIdx.x & 7, ch0 = cg * 8;
    const unsigned qpr = (unsigned)W >> 2, qpi = qpr * (unsigned)H;  // quads per row / per image
    const unsigned q = blockIdx.x * 32u + (threadIdx.x >> 3);
    if (q >= (unsigned)B * qpi) return;
    const unsigned b = q / qpi, rem = q - b * qpi;
    const int h = (int)(rem / qpr), w0 = (int)(rem - (unsigned)h * qpr) * 4;
    float acc[4][8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + ch0), b1 = *reinterpret_cast<const float4*>(bias + ch0 + 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
            acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
        }
    }
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * 3 * plane;
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = h + kh - 1;
        const bool rok = ih >= 0 && ih < H;
        float in[3][6];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int iw = w0 + j - 1;
                in[c][j] = (rok && iw >= 0 && iw < W) ? xb[(size_t)c * plane + (size_t)ih * W + iw] : 0.f;
            }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* wp = ws + ((kh * 3 + kw) * 3 + c) * 64 + ch0;
                const float4 u = *reinterpret_cast<const float4*>(wp), v = *reinterpret_cast<const float4*>(wp + 4);
                const float wv[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[p][j] = __builtin_fmaf(in[c][p + kw], wv[j], acc[p][j]);
                asm volatile("" ::: "memory");  // one tap's weights live at a time (hoisted, the 27 reads need 216 registers)
            }
    }
    float sl[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sl[j] = slope[ch0 + j]; sc[j] = scale[ch0 + j]; sh[j] = shift[ch0 + j]; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        bf16x8_t h8, l8, dh8, dl8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[p][j];
            v = v >= 0.f ? v : v * sl[j];
            const __bf16 dhi = (__bf16)v;
            dh8[j] = dhi;
            dl8[j] = (__bf16)(v - (float)dhi);
            const float t = __builtin_fmaf(v, sc[j], sh[j]);
            const __bf16 hi = (__bf16)t;
            h8[j] = hi;
            l8[j] = (__bf16)(t - (float)hi);
        }
        const int wq = w0 + p;
        const size_t m = ((size_t)b * H + h) * W + wq;
        const size_t at = pair_index(m, ch0, 64);
        *reinterpret_cast<bf16x8_t*>(Ph + at) = h8;
        *reinterpret_cast<bf16x8_t*>(Pl + at) = l8;
        if (Dh && !(h & 1) && !(wq & 1)) {
            const size_t md = ((size_t)b * (H >> 1) + (h >> 1)) * (W >> 1) + (wq >> 1);
            const size_t ad = pair_index(md, ch0, 64);
            *reinterpret_cast<bf16x8_t*>(Dh + ad) = dh8;
            *reinterpret_cast<bf16x8_t*>(Dl + ad) = dl8;
        }
    }
}

// ---- cv2.warpAffine (INTER_LINEAR, BORDER_CONSTANT 0), as published in OpenCV's imgwarp.cpp ----
constexpr int INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS, AB_BITS = 10, AB_SCALE = 1 << AB_BITS;
constexpr int INTER_REMAP_COEF_BITS = 15, INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS;

__device__ __forceinline__ int cv_round_sat(double v) {  // saturate_cast<int>(double) = cvRound: round half to even
    const double r = __builtin_rint(v);
    return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (int)0x80000000 : (int)r);
}

// the four int16 weights of fractional offsets (fx, fy) / 32: initInterTab2D's fixed-point table for INTER_LINEAR
__device__ __forceinline__ void bilinear_weights(int fx, int fy, int w[4]) {
    const float ax = (float)fx * (1.f / INTER_TAB_SIZE), ay = (float)fy * (1.f / INTER_TAB_SIZE);
    const float tx[2] = {1.f - ax, ax}, ty[2] = {1.f - ay, ay};
    int isum = 0;
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const float vv = ty[k1] * tx[k2];
            int q = (int)__builtin_rintf(vv * INTER_REMAP_COEF_SCALE);  // saturate_cast<short>(float): cvRound, then clamp
            q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
            w[k1 * 2 + k2] = q;
            isum += q;
        }
    if (isum != INTER_REMAP_COEF_SCALE) {
        // the published fix-up: the difference goes to the largest (too small a sum) or smallest (too large) of the 2 x 2 weights
        const int diff = isum - INTER_REMAP_COEF_SCALE;
        int mk = 0;
        for (int k = 1; k < 4; ++k) {
            if (diff < 0 ? w[k] > w[mk] : w[k] < w[mk]) mk = k;
        }
        w[mk] -= diff;
    }
}

// one thread per destination pixel; Minv = the INVERTED 2 x 3 matrix (double, as cv::warpAffine computes it), per face
__global__ __launch_bounds__(256) void warp_affine_kernel(const unsigned char* __restrict__ images, const long long* __restrict__ offsets,
                                                          const int* __restrict__ hw, const int* __restrict__ face_image,
                                                          const double* __restrict__ Minv, int nfaces, int size, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nfaces * size * size) return;
    const int f = e / (size * size), p = e - f * size * size, y = p / size, x = p - y * size;
    const int img = face_image[f];
    const unsigned char* src = images + offsets[img];
    const int H = hw[2 * img], W = hw[2 * img + 1];
    const double* M = Minv + 6 * f;
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
    const int adelta = cv_round_sat(M[0] * x * AB_SCALE), bdelta = cv_round_sat(M[3] * x * AB_SCALE);
    const int X0 = cv_round_sat((M[1] * y + M[2]) * AB_SCALE) + round_delta;
    const int Y0 = cv_round_sat((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
    int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);  // saturate_cast<short>
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    int w[4];
    bilinear_weights(X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1), w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = sy + dy, xx = sx + dx;
                const int s = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)src[((size_t)yy * W + xx) * 3 + c] : 0;  // borderValue 0
                acc += s * w[dy * 2 + dx];
            }
        int px = (acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS;  // FixedPtCast<int, uchar, 15>
        px = px < 0 ? 0 : (px > 255 ? 255 : px);
        // ToTensor (x / 255) then Normalize(0.5, 0.5): two IEEE operations each, as torchvision computes them
        const float t = (float)px / 255.0f;
        out[((size_t)f * 3 + c) * size * size + (size_t)y * size + x] = (t - 0.5f) / 0.5f;
    }
}

}  // namespace

extern "C" {

int mq_im2col_split_f32(const float* x_dev, int B, int H, int W, int C, int nchw, int KH, int KW, int stride, int pad,
                        const float* prelu_slope_dev, const float* scale_dev, const float* shift_dev, uint16_t* Ah_dev,
                        uint16_t* Al_dev, int Kpad, void* stream) {
    if (B == 0) return MQ_OK;
    if (!x_dev || !Ah_dev || !Al_dev || B < 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return MQ_EINVAL;
    if ((scale_dev == nullptr) != (shift_dev == nullptr)) return MQ_EINVAL;
    if (Kpad % 32 || Kpad < KH * KW * C) return MQ_EINVAL;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return MQ_EINVAL;
    const size_t M = (size_t)B * Ho * Wo;
    if (M >= 0x7FFFFFFFull) return MQ_EUNSUPPORTED;
    // x = 32-column blocks (<= 784 for the 7 x 7 x 512 head), (y, z) = 64-row tiles
    const size_t rt = (M + 63) / 64;
    const unsigned gy = (unsigned)(rt < 65535 ? rt : 65535), gz = (unsigned)((rt + 65534) / 65535);
    hipLaunchKernelGGL(im2col_split_kernel, dim3((unsigned)(Kpad / 32), gy, gz), dim3(256), 0, (hipStream_t)stream, x_dev, B, H, W, C,
                       nchw, KH, KW, stride, pad, Ho, Wo, prelu_slope_dev, scale_dev, shift_dev, (unsigned short*)Ah_dev,
                       (unsigned short*)Al_dev, Kpad);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

int mq_conv3x3_pair_f32(const uint16_t* Xh_dev, const uint16_t* Xl_dev, int B, int H, int W, int C, int stride, const uint16_t* Wh_dev,
                        const uint16_t* Wl_dev, int N, const float* bias_dev, const float* prelu_slope_dev, const float* residual_dev,
                        const float* scale_dev, const float* shift_dev, float* Y_dev, uint16_t* Ph_dev, uint16_t* Pl_dev,
                        const void* zeros_dev, int tile, void* stream) {
    if (B == 0) return MQ_OK;
    if (!Xh_dev || !Xl_dev || !Wh_dev || !Wl_dev || !bias_dev || !zeros_dev || B < 0 || H <= 0 || W <= 0) return MQ_EINVAL;
    if (C <= 0 || C % 32 || N <= 0 || N % 64 || (stride != 1 && stride != 2)) return MQ_EINVAL;
    const bool prelu = prelu_slope_dev != nullptr;
    if (prelu && (residual_dev || Y_dev || scale_dev || shift_dev || !Ph_dev || !Pl_dev)) return MQ_EINVAL;  // EP_PRELU_PAIR
    if (!prelu && (!residual_dev || !Y_dev)) return MQ_EINVAL;                                               // EP_RESIDUAL_AFFINE
    if ((scale_dev == nullptr) != (shift_dev == nullptr)) return MQ_EINVAL;
    if (!prelu && scale_dev && (!Ph_dev || !Pl_dev)) return MQ_EINVAL;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const size_t M = (size_t)B * Ho * Wo, P = (size_t)B * H * W;
    if (M >= 0x7FFFFFFFull || P >= 0x7FFFFFFFull) return MQ_EUNSUPPORTED;
    ConvArgs a;
    a.Xh = Xh_dev; a.Xl = Xl_dev; a.Wh = Wh_dev; a.Wl = Wl_dev; a.bias = bias_dev; a.slope = prelu_slope_dev; a.R = residual_dev;
    a.scale = scale_dev; a.shift = shift_dev; a.Y = Y_dev; a.Ph = Ph_dev; a.Pl = Pl_dev; a.zeros = zeros_dev;
    a.H = H; a.W = W; a.C = C; a.N = N; a.stride = stride; a.Ho = Ho; a.Wo = Wo; a.M = (int)M; a.ntm = a.ntn = 0;
    a.cmajor = (tile & MQ_CONV_K_CHANNEL_MAJOR) ? 1 : 0;
    tile &= ~MQ_CONV_K_CHANNEL_MAJOR;
    if (tile == MQ_CONV_TILE_AUTO) {
        // the widest tile the channel count fills; a 256-column tile only while it still gives every CU a workgroup
        if (N % 256 == 0 && ((M + 255) / 256) * (size_t)(N / 256) >= 192) tile = MQ_CONV_TILE_256x256;
        else if (N % 128 == 0) tile = ((M + 511) / 512) * (size_t)(N / 128) >= 192 ? MQ_CONV_TILE_512x128 : MQ_CONV_TILE_256x128;
        else tile = MQ_CONV_TILE_512x64;
    }
    return prelu ? launch_conv3x3_variant<EP_PRELU_PAIR>(a, tile, (hipStream_t)stream)
                 : launch_conv3x3_variant<EP_RESIDUAL_AFFINE>(a, tile, (hipStream_t)stream);
}

int mq_stem_conv3x3_f32(const float* x_dev, int B, int H, int W, const float* wt_dev, const float* bias_dev, const float* prelu_slope_dev,
                        const float* scale_dev, const float* shift_dev, uint16_t* Ph_dev, uint16_t* Pl_dev, uint16_t* Dh_dev,
                        uint16_t* Dl_dev, void* stream) {
    if (B == 0) return MQ_OK;
    if (!x_dev || !wt_dev || !bias_dev || !prelu_slope_dev || !scale_dev || !shift_dev || !Ph_dev || !Pl_dev) return MQ_EINVAL;
    if ((Dh_dev == nullptr) != (Dl_dev == nullptr) || B < 0 || H <= 0 || W <= 0) return MQ_EINVAL;
    if ((W & 3) || (Dh_dev && (H & 1))) return MQ_EUNSUPPORTED;
    const size_t quads = (size_t)B * H * (W / 4);
    if ((quads + 31) / 32 > 0x7FFFFFFFull) return MQ_EUNSUPPORTED;
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)((quads + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x_dev, wt_dev, bias_dev,
                       prelu_slope_dev, scale_dev, shift_dev, (unsigned short*)Ph_dev, (unsigned short*)Pl_dev, (unsigned short*)Dh_dev,
                       (unsigned short*)Dl_dev, B, H, W);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

int mq_warp_affine_faces_f32(const uint8_t* images_dev, const int64_t* offsets_dev, const int32_t* hw_dev, const int32_t* face_image_dev,
                             const double* minv_dev, int nfaces, int size, float* out_dev, void* stream) {
    if (nfaces == 0) return MQ_OK;
    if (!images_dev || !offsets_dev || !hw_dev || !face_image_dev || !minv_dev || !out_dev || nfaces < 0 || size <= 0) return MQ_EINVAL;
    const int total = nfaces * size * size;
    hipLaunchKernelGGL(warp_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images_dev,
                       (const long long*)offsets_dev, (const int*)hw_dev, (const int*)face_image_dev, minv_dev, nfaces, size, out_dev);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

}  // extern "C"
