// micro-benchmark: does VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) s += acc[t][v];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int WAVES>
void run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 / WAVES * (WAVES > 4 ? 1 : 1);
    hipLaunchKernelGGL((k<NV, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 4.0 * iters;                       // MFMAs per wave
    printf("valu/mfma %2d waves/CU %d: %8.1f us  -> %.1f clk per MFMA per SIMD-wave-slot (2.4 GHz)\n", NV, WAVES, ms * 1e3,
           ms * 1e-3 * 2.4e9 / mf / (WAVES > 4 ? WAVES / 4 : 1));
    (void)blocks;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    run<0, 4>(d, iters); run<4, 4>(d, iters); run<8, 4>(d, iters); run<12, 4>(d, iters); run<16, 4>(d, iters); run<24, 4>(d, iters);
    run<0, 8>(d, iters); run<8, 8>(d, iters); run<16, 8>(d, iters);
    return 0;
}
