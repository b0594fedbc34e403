// diag.hip -- measurement aid, not on the product path: the bf16 matrix-pipe rate this MI355X SUSTAINS with nothing else going
// on.  bench.py times it next to the screening scan so that the roofline discussion has the ceiling of the same box, same run:
// v_mfma_f32_32x32x16_bf16 on register operands (no LDS, no memory in the loop), four accumulator chains per wave, 16 waves per
// CU on every CU.  With zero operands the chip holds ~2.39 GHz (2.4 PFLOP/s: the nominal dense peak); with N(0,1)-like random
// bf16 operands the power management settles at ~1.78 GHz (1.75 PFLOP/s) although the pipe is 93 % busy -- the ceiling of any
// bf16 GEMM-shaped kernel on such data, before it moves a single operand (profiles/r02_mfma_peak.txt).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/meerqat_hip.h"

namespace {
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// random sign and mantissa, exponents 2^-3 .. 2^1 (the bulk of N(0,1) samples)
__device__ __forceinline__ bf16x8_t operand(unsigned seed, int random_operands) {
    union { bf16x8_t v; unsigned short s[8]; } u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned h = mix(seed * 8u + i);
        const unsigned e = 124u + (h & 3u) + ((h >> 2) & 1u);
        u.s[i] = random_operands ? (unsigned short)(((h >> 16) & 0x8000u) | (e << 7) | ((h >> 8) & 0x7Fu)) : (unsigned short)0;
    }
    return u.v;
}

__global__ __launch_bounds__(1024) void mfma_loop_kernel(int iters, int random_operands, float* __restrict__ out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operand(tid * 8u + i, random_operands);
        b[i] = operand(tid * 8u + 4 + i, random_operands);
    }
    f32x16_t c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 16 MFMAs: the 2 x 2 sub-tile pattern of the scans
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + 1) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + 1) & 3], b[u], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + 1) & 3], b[(u + 1) & 3], c3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (out) out[tid] = s;
}

// What v_mfma_f32_32x32x16_bf16 RETURNS for a 32 x 32 tile of dot products over dp columns, accumulated like the screening
// scan accumulates them: one MFMA per 16 columns, in column order, each taking the previous result as its C operand.  Lane l
// feeds row l & 31, columns 16 s + 8 (l >> 5) .. + 7 of step s; C/D: out[row 8 (reg >> 2) + (reg & 3) + 4 (l >> 5)][column l & 31].
__global__ __launch_bounds__(64) void mfma_dot_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int dp,
                                                      float* __restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    f32x16_t c = {0};
    for (int k0 = 0; k0 < dp; k0 += 16) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(A + (size_t)i * dp + k0 + 8 * kg);
        const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(B + (size_t)i * dp + k0 + 8 * kg);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) out[(8 * (reg >> 2) + (reg & 3) + 4 * kg) * 32 + i] = c[reg];
}
}  // namespace

extern "C" {
int mq_diag_mfma_bf16_dot(const uint16_t* A_dev, const uint16_t* B_dev, int dp, float* out_dev, void* stream) {
    if (!A_dev || !B_dev || !out_dev || dp <= 0 || dp % 16 != 0) return MQ_EINVAL;
    hipLaunchKernelGGL(mfma_dot_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A_dev, B_dev, dp, out_dev);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}

int mq_diag_mfma_bf16_loop(int iters, int random_operands, int workgroups, float* out_dev, void* stream) {
    if (iters < 0 || workgroups <= 0 || !out_dev) return MQ_EINVAL;
    hipLaunchKernelGGL(mfma_loop_kernel, dim3((unsigned)workgroups), dim3(1024), 0, (hipStream_t)stream, iters, random_operands, out_dev);
    return hipGetLastError() == hipSuccess ? MQ_OK : MQ_EHIP;
}
}
