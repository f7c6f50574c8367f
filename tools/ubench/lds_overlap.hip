// micro-benchmark: does a workgroup with MORE THAN 64 KB of LDS keep its LDS to itself on MI355X (160 KB per CU)?
// A "writer" kernel (BYTES of LDS: static array, or dynamic with hipFuncAttributeMaxDynamicSharedMemorySize) keeps filling its
// whole allocation with a pattern; a "victim" kernel on a second stream (12 KB of LDS) writes its own pattern once, then re-reads
// it for a while and counts every word that changed.  Found in round 5: the furthest-point-sampling kernel of step N + 1 picked
// wrong points in ~1 % of the steps when it ran beside the split-bf16 wide-tile GEMMs (79 - 80 KB static LDS) of step N.
// build: hipcc -O3 --offload-arch=gfx950 -w lds_overlap.hip -o lds_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BYTES, bool DYN>
__global__ __launch_bounds__(256, 2) void writer(int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned stat[DYN ? 4 : BYTES / 4];
    extern __shared__ __attribute__((aligned(16))) unsigned dyn[];
    unsigned* lds = DYN ? dyn : stat;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < BYTES / 4; i += 256) lds[i] = 0xAB000000u + (unsigned)it;
        __syncthreads();
        acc += lds[(threadIdx.x * 37 + it) % (BYTES / 4)];
        __syncthreads();
    }
    if (acc == 12345u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void victim(int iters, unsigned* bad) {
    __shared__ unsigned lds[3072];
    for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = 0x51000000u + (unsigned)i;
    __syncthreads();
    unsigned n = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 3072; i += 256) n += lds[i] != 0x51000000u + (unsigned)i;
        __builtin_amdgcn_s_sleep(8);
    }
    if (n) atomicAdd(bad, n);
}

template <int BYTES, bool DYN>
static void run(const char* what) {
    unsigned *bad, *sink;
    hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    hipMemset(bad, 0, 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    if (DYN) hipFuncSetAttribute(reinterpret_cast<const void*>(writer<BYTES, DYN>), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL((writer<BYTES, DYN>), dim3(512), dim3(256), DYN ? BYTES : 0, s1, 200, sink);
        hipLaunchKernelGGL(victim, dim3(512), dim3(256), 0, s2, 2000, bad);
    }
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(writer<BYTES, DYN>));
    printf("%-44s static LDS %6zu B: victim words changed: %u  (%s)\n", what, fa.sharedSizeBytes, h, hipGetErrorString(hipGetLastError()));
}

int main() {
    run<60 * 1024, false>("writer 60 KB static");
    run<64 * 1024, false>("writer 64 KB static");
    run<65 * 1024, false>("writer 65 KB static");
    run<80 * 1024, false>("writer 80 KB static");
    run<80 * 1024, true>("writer 80 KB dynamic + attribute");
    run<150 * 1024, false>("writer 150 KB static");
    run<150 * 1024, true>("writer 150 KB dynamic + attribute");
    return 0;
}
