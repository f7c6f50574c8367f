// micro-benchmark: how should G workgroups add their private (E)-element f32 blocks into ONE f64 arena block?
//   (a) f64 device atomics straight from registers (same element order in every workgroup / rotated by workgroup),
//   (b) f32 partial slab (G x E) + a second launch that sums the slab in f64 (today's dw_reduce scheme).
// This decides how the fused wide backward (SA2 / SA3 weight gradients) leaves its accumulators.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -w atomic_reduce.hip -o atomic_reduce.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_atomic(double* arena, int E, int rotate, float v) {
    const int per = E / 256;                                   // elements per thread
    const int rot = rotate ? (blockIdx.x * 977) % per : 0;
    for (int i = 0; i < per; ++i) {
        const int j = (i + rot) % per;
        unsafeAtomicAdd(arena + (size_t)j * 256 + threadIdx.x, (double)(v + j));
    }
}

__global__ __launch_bounds__(256) void k_atomic_f32(float* arena, int E, int rotate, float v) {
    const int per = E / 256;
    const int rot = rotate ? (blockIdx.x * 977) % per : 0;
    for (int i = 0; i < per; ++i) {
        const int j = (i + rot) % per;
        unsafeAtomicAdd(arena + (size_t)j * 256 + threadIdx.x, v + j);
    }
}

__global__ __launch_bounds__(256) void k_slab(float* slab, int E, float v) {
    const int per = E / 256;
    float* p = slab + (size_t)blockIdx.x * E;
    for (int i = 0; i < per; ++i) p[(size_t)i * 256 + threadIdx.x] = v + i;
}

__global__ __launch_bounds__(256) void k_reduce(const float* slab, double* arena, int E, int G, int chunk) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int g0 = blockIdx.y * chunk, g1 = min(G, g0 + chunk);
    double s = 0.0;
    for (int g = g0; g < g1; ++g) s += (double)slab[(size_t)g * E + e];
    unsafeAtomicAdd(arena + e, s);
}

template <class F>
float time_us(F f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    double* arena; float* slab; float* arena32;
    hipMalloc(&arena, 1 << 22); hipMalloc(&arena32, 1 << 22); hipMalloc(&slab, (size_t)1024 * 131072 * 4);
    hipMemset(arena, 0, 1 << 22); hipMemset(arena32, 0, 1 << 22);
    const int Gs[] = {128, 256, 512, 1024};
    const int Es[] = {16384, 32768, 65536, 131072};
    for (int E : Es) for (int G : Gs) {
        if ((size_t)G * E > (size_t)1024 * 131072) continue;
        const float a0 = time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(G), dim3(256), 0, 0, arena, E, 0, 1.f); });
        const float a1 = time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(G), dim3(256), 0, 0, arena, E, 1, 1.f); });
        const float a2 = time_us([&] { hipLaunchKernelGGL(k_atomic_f32, dim3(G), dim3(256), 0, 0, arena32, E, 1, 1.f); });
        const float s0 = time_us([&] { hipLaunchKernelGGL(k_slab, dim3(G), dim3(256), 0, 0, slab, E, 1.f); });
        const float s1 = time_us([&] {
            hipLaunchKernelGGL(k_slab, dim3(G), dim3(256), 0, 0, slab, E, 1.f);
            hipLaunchKernelGGL(k_reduce, dim3(E / 256, (G + 15) / 16), dim3(256), 0, 0, slab, arena, E, G, 16);
        });
        printf("G %4d E %6d (%5.1f M adds): f64 atomics %7.1f us, rotated %7.1f us, f32 atomics rotated %7.1f us | slab write %6.1f us, slab + reduce %6.1f us\n",
               G, E, (double)G * E * 1e-6, a0, a1, a2, s0, s1);
    }
    return 0;
}
