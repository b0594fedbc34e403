// How fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16, by the number of independent accumulators and by where the B
// operand lives (VGPR through the builtin, VGPR / AGPR through an asm statement)?  Cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_bench.hip -o tools/probes/mfma_rate_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_A(acc, xa, qb) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(xa), "a"(qb))
#define MF_V(acc, xa, qb) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(xa), "v"(qb))
#define MF_VV(acc, xa, qb) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(xa), "v"(qb))
#define MF_NA(acc, xa, qb) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(xa), "a"(qb))

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, int iters, unsigned seed) {
    union { bf16x8_t v; unsigned u[4]; } a, b;
    for (int e = 0; e < 4; ++e) {
        unsigned h = seed * 2654435761u + (unsigned)(threadIdx.x * 4 + e) * 40503u;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        a.u[e] = seed ? ((h & 0x807F807Fu) | 0x3F003F00u) : 0u;
        b.u[e] = seed ? (((h * 31u) & 0x807F807Fu) | 0x3F003F00u) : 0u;
    }
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8_t bb = b.v;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 96; ++m) {
            if (MODE == 0) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, bb, acc[m % NACC], 0, 0, 0);
            if (MODE == 1) MF_V(acc[m % NACC], a.v, bb);
            if (MODE == 2) MF_A(acc[m % NACC], a.v, bb);
            if (MODE == 3) MF_NA(acc[m % NACC], a.v, bb);
            if (MODE == 4) MF_VV(acc[m % NACC], a.v, bb);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0]), "+a"(acc[1]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 1.2345f) out[1] = 1;
}

template <int MODE, int NACC>
void run(unsigned long long* out, int threads, unsigned seed) {
    const int iters = 200;
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate_kernel<MODE, NACC>), dim3(256), dim3(threads), 0, 0, out, iters, seed);
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    }
    const char* names[] = {"builtin (VGPR B)", "asm, B in VGPR, acc AGPR", "asm, B in AGPR, acc AGPR", "asm + s_nop 1, B in AGPR", "asm, all VGPR"};
    printf("%-28s accumulators %d, waves/SIMD %d, operands %s: %.1f memtime ticks per MFMA\n", names[MODE], NACC, threads / 256, seed ? "random" : "zero",
           (double)h / (iters * 96.0));
    fflush(stdout);
}


// One wave per SIMD: NM MFMAs per "tile" in CH dependent chains, one ds_read_b128 behind every RD-th MFMA requested DIST MFMAs ahead of
// its use (ring of DIST fragments), B operands in registers.
template <int CH, int RD, int DIST>
__global__ __launch_bounds__(256) void lds_mfma_kernel(unsigned long long* out, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 144 * 1024 / 4; e += 256) reinterpret_cast<unsigned*>(smem)[e] = (e * 2654435761u) & 0x3F803F80u;
    __syncthreads();
    union { bf16x8_t v; unsigned u[4]; } b;
    for (int e = 0; e < 4; ++e) {
        unsigned h = seed * 2654435761u + (unsigned)(threadIdx.x * 4 + e) * 40503u;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        b.u[e] = ((h * 31u) & 0x807F807Fu) | 0x3F003F00u;
    }
    const int i = lane & 31, kg = lane >> 5, sw = (i >> 1) & 7;
    unsigned xs[4];
    for (int m = 0; m < 4; ++m) xs[m] = (unsigned)(i * 128 + (((2 * m + kg) ^ sw) * 16));
    f32x16 acc[2] = {{0}, {0}};
    bf16x8_t bb = b.v;
    constexpr int NMF = 96;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned base = (unsigned)((it % 3) * 49152);
        bf16x8_t xf[DIST];
#define LDX(g) (*reinterpret_cast<const bf16x8_t*>(smem + base + xs[(g) & 3] + ((((g) / RD) % 48) >> 2) * 4096))
#pragma unroll
        for (int g = 0; g < DIST; ++g) xf[g] = LDX(g * RD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NMF; ++g) {
            MF_NA(acc[g % CH], xf[(g / RD) % DIST], bb);
            __builtin_amdgcn_sched_barrier(0);
            if (g % RD == RD - 1 && g + RD * DIST - (RD - 1) < NMF) xf[(g / RD) % DIST] = LDX(g + RD * DIST - (RD - 1));
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LDX
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0]), "+a"(acc[1]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc[0][0] + acc[1][1] == 1.2345f) out[1] = 1;
}

template <int CH, int RD, int DIST>
void run_lds(unsigned long long* out) {
    const int iters = 200;
    unsigned long long h = 0;
    hipFuncSetAttribute((const void*)lds_mfma_kernel<CH, RD, DIST>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((lds_mfma_kernel<CH, RD, DIST>), dim3(256), dim3(256), 144 * 1024, 0, out, iters, 7u);
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    }
    printf("chains %d, one ds_read_b128 per %d MFMAs, requested %2d MFMAs ahead: %.1f ticks per MFMA  (%s)\n", CH, RD, RD * DIST, (double)h / (iters * 96.0),
           hipGetLastError() == hipSuccess ? "ok" : "ERR");
    fflush(stdout);
}


// What hides behind an MFMA of one wave per SIMD?  After every MFMA on acc0: NV plain VALU ops (v_max3 on VGPRs) and NR reads of the
// OTHER accumulator's registers (v_accvgpr_read of acc1, which no MFMA in flight writes).
template <int NV, int NR>
__global__ __launch_bounds__(256) void filler_kernel(unsigned long long* out, int iters, unsigned seed) {
    union { bf16x8_t v; unsigned u[4]; } a, b;
    for (int e = 0; e < 4; ++e) {
        unsigned h = seed * 2654435761u + (unsigned)(threadIdx.x * 4 + e) * 40503u;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        a.u[e] = (h & 0x807F807Fu) | 0x3F003F00u;
        b.u[e] = ((h * 31u) & 0x807F807Fu) | 0x3F003F00u;
    }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int e = 0; e < 16; ++e) acc1[e] = (float)(threadIdx.x + e);
    asm volatile("" : "+a"(acc1));
    float x0 = (float)threadIdx.x, x1 = 1.f, x2 = 2.f, r = 0.f;
    bf16x8_t aa = a.v, bb = b.v;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 96; ++g) {
            MF_NA(acc0, aa, bb);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
#pragma unroll
            for (int v = 0; v < NR; ++v) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc1[(g + v) & 15])); asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(t)); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc0));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc0[0] + x0 + r == 1.2345f) out[1] = 1;
}
template <int NV, int NR>
void run_filler(unsigned long long* out) {
    const int iters = 200;
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((filler_kernel<NV, NR>), dim3(256), dim3(256), 0, 0, out, iters, 7u);
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    }
    printf("behind every MFMA: %d v_max3 + %d (v_accvgpr_read of the other accumulator + v_max): %.1f ticks per MFMA\n", NV, NR, (double)h / (iters * 96.0));
    fflush(stdout);
}


// Price of a wave-uniform branch behind every MFMA (one wave per SIMD): `flag` = 0: the branch is TAKEN (skips a block), 1: it falls through.
__global__ __launch_bounds__(256) void branch_kernel(unsigned long long* out, int iters, unsigned seed, int flag) {
    union { bf16x8_t v; unsigned u[4]; } a, b;
    for (int e = 0; e < 4; ++e) {
        unsigned h = seed * 2654435761u + (unsigned)(threadIdx.x * 4 + e) * 40503u;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        a.u[e] = (h & 0x807F807Fu) | 0x3F003F00u;
        b.u[e] = ((h * 31u) & 0x807F807Fu) | 0x3F003F00u;
    }
    f32x16 acc0 = {0};
    bf16x8_t aa = a.v, bb = b.v;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 96; ++g) {
            MF_NA(acc0, aa, bb);
            if (flag) asm volatile("s_nop 0" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc0));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc0[0] == 1.2345f) out[1] = 1;
}
void run_branch(unsigned long long* out, int flag) {
    const int iters = 200;
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(branch_kernel, dim3(256), dim3(256), 0, 0, out, iters, 7u, flag);
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    }
    printf("a wave-uniform branch behind every MFMA, %s: %.1f ticks per MFMA\n", flag ? "falling through" : "taken", (double)h / (iters * 96.0));
    fflush(stdout);
}

int main() {
    unsigned long long* out;
    hipMalloc(&out, 64);
    run<0, 2>(out, 256, 0); run<0, 2>(out, 256, 7);
    run<0, 4>(out, 256, 7); run<0, 1>(out, 256, 7);
    run<1, 2>(out, 256, 7); run<2, 2>(out, 256, 7); run<3, 2>(out, 256, 7); run<4, 2>(out, 256, 7);
    run<2, 4>(out, 256, 7); run<3, 4>(out, 256, 7);
    run_branch(out, 0); run_branch(out, 1);
    run_filler<0, 0>(out); run_filler<2, 0>(out); run_filler<4, 0>(out); run_filler<6, 0>(out); run_filler<8, 0>(out); run_filler<0, 1>(out); run_filler<0, 2>(out); run_filler<0, 3>(out); run_filler<2, 2>(out);
    run_lds<1, 1, 8>(out); run_lds<2, 1, 8>(out); run_lds<1, 2, 4>(out); run_lds<2, 2, 4>(out); run_lds<1, 1, 4>(out); run_lds<1, 1, 12>(out); run_lds<2, 2, 8>(out);
    return 0;
}
