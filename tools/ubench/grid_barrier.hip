// micro-benchmark: what does a grid-wide barrier cost on MI355X (8 XCDs, one L2 each)?  Decides whether the M = 256 tail of an
// encoder pass (FC1 -> BN -> FC2 -> BN -> head L1 -> L2 -> L3: seven dependent launches of 7 - 17 us for 0.9 GFLOP) could
// run as ONE persistent launch with barriers between its phases.
//   (a) hand-written sense-reversing barrier: agent-scope atomic arrive + spin on a generation word (release / acquire),
//   (b) cooperative_groups grid.sync() under hipLaunchCooperativeKernel,
// each with every workgroup writing and then reading 4 KB of the other workgroups' data per phase (so the fences have
// something to publish).
// build: hipcc -O3 --offload-arch=gfx950 -w grid_barrier.hip -o grid_barrier.bin
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_RELEASE);                                  // agent scope: publish this workgroup's stores
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, unsigned* sync, int phases, float* out) {
    cg::grid_group grid = cg::this_grid();
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int p = 0; p < phases; ++p) {
        float* mine = buf + ((size_t)(p & 1) * nb + blockIdx.x) * 1024;
        for (int i = threadIdx.x; i < 1024; i += 256) mine[i] = acc + i + p;
        if (MODE == 0) grid_barrier(sync, sync + 32, nb);
        else if (MODE == 1) grid.sync();
        const float* other = buf + ((size_t)(p & 1) * nb + (blockIdx.x + 1 + p) % nb) * 1024;
        for (int i = threadIdx.x; i < 1024; i += 256) acc += other[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
float run(int blocks, int phases, float* buf, unsigned* sync, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void* args[] = {&buf, &sync, &phases, &out};
    auto launch = [&] {
        hipMemsetAsync(sync, 0, 256, 0);
        if (MODE == 1) hipLaunchCooperativeKernel((void*)k<1>, dim3(blocks), dim3(256), args, 0, 0);
        else hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, buf, sync, phases, out);
    };
    launch(); launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.f;                                                            // us per launch
}

int main() {
    float *buf, *out; unsigned* sync;
    hipMalloc(&buf, (size_t)2 * 1024 * 1024 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&sync, 256);
    for (int blocks : {64, 128, 256, 512}) {
        const float base = run<2>(blocks, 64, buf, sync, out);
        const float a = run<0>(blocks, 64, buf, sync, out), a2 = run<0>(blocks, 8, buf, sync, out);
        const float c = run<1>(blocks, 64, buf, sync, out), c2 = run<1>(blocks, 8, buf, sync, out);
        printf("%4d workgroups: no barrier %6.1f us / 64 phases | atomic barrier %6.2f us each | grid.sync %6.2f us each\n", blocks, base,
               (a - a2) / 56.f, (c - c2) / 56.f);
    }
    return 0;
}
