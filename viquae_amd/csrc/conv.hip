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
// An explicit im2col costs HBM traffic the implicit form would not (9x the activation for a 3 x 3 kernel): ~150 MB per face
// over the 50 convolutions, about as long as the 12.6 GFLOP of matrix work at the GEMM's rate (next step = the
// gather inside the GEMM's LDS-DMA addresses).  Neither OpenCV, scikit-image nor arcface_torch is vendored by the reference or
// installable here: this path is "parity unpinned" (oracle/arcface.py restates the published algorithms).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/meerqat_hip.h"

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
    const unsigned row = blockIdx.y * 64u + (threadIdx.x >> 2);
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
    if (M >= 0x7FFFFFFFull || (M + 63) / 64 > 65535ull * 1024ull) return MQ_EUNSUPPORTED;
    // x = 32-column blocks (<= 784 for the 7 x 7 x 512 head), y = 64-row tiles (gridDim.y <= 65535: y carries the rest in z ... not
    // needed: 256 faces x 12544 pixels / 64 = 50176 tiles)
    if ((M + 63) / 64 > 65535ull) return MQ_EUNSUPPORTED;
    hipLaunchKernelGGL(im2col_split_kernel, dim3((unsigned)(Kpad / 32), (unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x_dev, B, H, W, C,
                       nchw, KH, KW, stride, pad, Ho, Wo, prelu_slope_dev, scale_dev, shift_dev, (unsigned short*)Ah_dev,
                       (unsigned short*)Al_dev, Kpad);
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
