// store_rate.hip -- how fast can ONE compute unit store?  (round 6, behind the GEMM epilogue question: a 256 x 256 fp32 tile = 256 KiB
// left a CU in ~9 us whatever the epilogue looked like.)  Each workgroup (512 threads) writes `per_wg` bytes as full 1-KiB wave
// stores (global_store_dwordx4, consecutive lanes = consecutive 16 bytes), `wgs` workgroups at once; variants: plain / non-temporal.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float4* out, size_t per_wg_vec, int reps) {
    float4* base = out + (size_t)blockIdx.x * per_wg_vec;
    const float4 v = {1.f * threadIdx.x, 2.f, 3.f, 4.f};
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i < per_wg_vec; i += 512) {
            if (MODE == 0) base[i] = v;
            else if (MODE == 1) { typedef float f4 __attribute__((ext_vector_type(4))); f4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, reinterpret_cast<f4*>(base + i)); }
            else if (MODE == 2) { float* p = reinterpret_cast<float*>(base + i); p[0] = v.x; }  // 4-byte stores of a quarter of the bytes
        }
}

// the GEMM epilogue's pattern: a wave instruction stores 8 rows x 128 B (8 lanes per row), rows `ld` floats apart; a workgroup of 8
// waves walks 256 x 256 fp32 tiles (wave w: rows 128 (w >> 2) + 32 rb + 8 t + (lane >> 3), columns 64 (w & 3) + 32 cb + 4 (lane & 7))
template <int VEC>  // 4: dwordx4 (8 lanes per 128-byte row), 2: dwordx2 (16 lanes per row: two instructions per 8 rows)
__global__ __launch_bounds__(512) void tile_store_kernel(float* out, int ld, int tiles_per_wg, int ntn) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int tile = blockIdx.x + it * gridDim.x;
        const size_t m0 = (size_t)(tile / ntn) * 256, n0 = (size_t)(tile % ntn) * 256;
        for (int a = 0; a < 8; ++a) {
            const int rb = a >> 1, cb = a & 1;
            for (int t = 0; t < 4; ++t) {
                if (VEC == 4) {
                    const size_t row = m0 + 128 * (w >> 2) + 32 * rb + 8 * t + (lane >> 3), col = n0 + 64 * (w & 3) + 32 * cb + 4 * (lane & 7);
                    *reinterpret_cast<float4*>(out + row * ld + col) = float4{1.f * lane, 2.f, 3.f, 4.f};
                } else {
                    for (int h = 0; h < 2; ++h) {
                        const size_t row = m0 + 128 * (w >> 2) + 32 * rb + 8 * t + 4 * h + (lane >> 4), col = n0 + 64 * (w & 3) + 32 * cb + 2 * (lane & 15);
                        *reinterpret_cast<float2*>(out + row * ld + col) = float2{1.f * lane, 2.f};
                    }
                }
            }
        }
    }
}

int main() {
    const size_t per_wg = 8u << 20;  // 8 MiB per workgroup and repetition: far beyond any cache
    int ncu = 256;
    float4* buf;
    hipMalloc(&buf, per_wg * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {1, 8, 64, 256}) {
            const int reps = wgs == 256 ? 4 : 8;
            hipLaunchKernelGGL(store_kernel<0>, dim3(wgs), dim3(512), 0, 0, buf, per_wg / 16, 1);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(wgs), dim3(512), 0, 0, buf, per_wg / 16, reps);
            else hipLaunchKernelGGL(store_kernel<1>, dim3(wgs), dim3(512), 0, 0, buf, per_wg / 16, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)per_wg * wgs * reps;
            printf("%s stores, %3d workgroups: %.2f ms, %.1f GB/s total, %.1f GB/s per workgroup\n", mode ? "non-temporal" : "plain", wgs, ms,
                   bytes / ms / 1e6, bytes / ms / 1e6 / wgs);
        }
    (void)ncu;
    // tile pattern: M x 2304 fp32 (the QKV output), 7200 tiles
    const int ld = 2304, ntn = 9, M = 204800;
    float* cbuf;
    hipMalloc(&cbuf, (size_t)M * ld * 4);
    for (int vec : {4, 2})
        for (int wgs : {1, 8, 64, 256}) {
            const int tiles_per_wg = 7200 / 256;
            hipEventRecord(e0);
            if (vec == 4) hipLaunchKernelGGL(tile_store_kernel<4>, dim3(wgs), dim3(512), 0, 0, cbuf, ld, tiles_per_wg, ntn);
            else hipLaunchKernelGGL(tile_store_kernel<2>, dim3(wgs), dim3(512), 0, 0, cbuf, ld, tiles_per_wg, ntn);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 262144.0 * tiles_per_wg * wgs;
            printf("tile pattern dwordx%d, %3d workgroups: %.3f ms, %.1f GB/s total, %.1f GB/s per workgroup = %.2f us per 256-KiB tile\n", vec, wgs, ms,
                   bytes / ms / 1e6, bytes / ms / 1e6 / wgs, 262144.0 / (bytes / ms / 1e6 / wgs) / 1e3);
        }
    return 0;
}
