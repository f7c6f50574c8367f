// micro-benchmark: wavefront A issues a chain of v_mfma_f32_32x32x2_f32 (4 accumulators) while wavefront B on the SAME SIMD runs
// an "epilogue" -- a stream of (a) v_fma_f32, (b) v_pk_fma_f32, (c) ds_write_b32, (d) buffer_store_dword -- how long does each side
// take against running alone?  (The streaming kernels keep 2 wavefronts per SIMD: while one is in its MFMA loop the other is in
// its epilogue.)
// build: hipcc -O3 --offload-arch=gfx950 -w mfma_corun.hip -o mfma_corun.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE_B: 0 none (B idle), 1 v_fma, 2 v_pk_fma, 3 ds_write_b32, 4 global store dword; MODE_A: 1 = MFMA chain, 0 = idle
template <int MODE_A, int MODE_B>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int n_mfma, int n_b, float a, float b) {
    __shared__ float lds[512 * 4];
    const int wave = threadIdx.x >> 6;                     // waves w and w + 4 share a SIMD
    long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < 4) {
        if (MODE_A) {
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
            for (int it = 0; it < n_mfma / 4; ++it)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) s += acc[t][v];
        }
    } else if (MODE_B == 5) {                             // v_fma_f32 at raised wave priority
        __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x + i;
        for (int it = 0; it < n_b / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        for (int i = 0; i < 8; ++i) s += x[i];
        __builtin_amdgcn_s_setprio(0);
    } else if (MODE_B == 6) {                             // stores at raised wave priority
        __builtin_amdgcn_s_setprio(3);
        float* o = out + 1024 * 1024 + (size_t)blockIdx.x * 65536 + threadIdx.x;
        for (int it = 0; it < n_b / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[(it & 15) * 4096 + i * 512] = a + it;
            __asm__ volatile("" ::: "memory");
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (MODE_B == 1) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x + i;
        for (int it = 0; it < n_b / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        for (int i = 0; i < 8; ++i) s += x[i];
    } else if (MODE_B == 2) {
        f32x2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f32x2{a + threadIdx.x + i, b};
        const f32x2 aa = {a, a}, bb = {b, b};
        for (int it = 0; it < n_b / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], aa, bb);
        for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    } else if (MODE_B == 3) {
        for (int it = 0; it < n_b / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lds[threadIdx.x + 512 * (i & 3)] = a + it;
            __asm__ volatile("" ::: "memory");
        }
        s = lds[threadIdx.x];
    } else if (MODE_B == 4) {
        float* o = out + 1024 * 1024 + (size_t)blockIdx.x * 65536 + threadIdx.x;
        for (int it = 0; it < n_b / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[(it & 15) * 4096 + i * 512] = a + it;
            __asm__ volatile("" ::: "memory");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// same wavefront: NF filler instructions of kind F between consecutive MFMAs (1 = v_fma_f32, 3 = ds_write_b32, 4 = global store
// dword, 7 = global load dword, 8 = ds_read_b128); 4 wavefronts per workgroup = one per SIMD
template <int F, int NF>
__global__ __launch_bounds__(256) void ks(float* out, long long* cyc, int n_mfma, float a, float b) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 8];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x + i;
    float* o = out + 1024 * 1024 + (size_t)blockIdx.x * 65536 + threadIdx.x;
    float ld = 0.f;
    float4 l4 = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n_mfma / 4; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (F == 1) x[i & 7] = __builtin_fmaf(x[i & 7], a, b);
                if (F == 3) lds[threadIdx.x + 256 * (i & 7)] = a + it;
                if (F == 4) o[(it & 15) * 4096 + (t * NF + i) * 256] = a + it;
                if (F == 7) ld += __builtin_nontemporal_load(o + (it & 15) * 4096 + (t * NF + i) * 256);
                if (F == 8) { const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x + 64 * i + it) & 255) * 4); l4.x += q.x; l4.y += q.w; }
            }
            if (F == 3 || F == 4) __asm__ volatile("" ::: "memory");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = ld + l4.x + l4.y + lds[threadIdx.x];
    for (int t = 0; t < 4; ++t) for (int v = 0; v < 16; ++v) s += acc[t][v];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int F, int NF>
void run_same(const char* what, float* d, long long* c, int n_mfma) {
    hipLaunchKernelGGL((ks<F, NF>), dim3(256), dim3(256), 0, 0, d, c, n_mfma, 1.0f, 0.5f);
    hipLaunchKernelGGL((ks<F, NF>), dim3(256), dim3(256), 0, 0, d, c, n_mfma, 1.0f, 0.5f);
    hipDeviceSynchronize();
    static long long h[256 * 4];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double A = 0;
    for (int i = 0; i < 1024; ++i) A += h[i];
    printf("same wavefront: %d x %-22s per MFMA: %6.1f cyc per MFMA\n", NF, what, A / 1024 / n_mfma);
}

template <int MA, int MB>
void run(const char* what, float* d, long long* c, int n_mfma, int n_b) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MA, MB>), dim3(256), dim3(512), 0, 0, d, c, n_mfma, n_b, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MA, MB>), dim3(256), dim3(512), 0, 0, d, c, n_mfma, n_b, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double A = 0, B = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? A : B) += h[b * 8 + w];
    A /= 1024; B /= 1024;
    printf("%-44s A: %8.0f cyc (%5.1f per MFMA) | B: %8.0f cyc (%5.1f per instruction) | %7.1f us -> %.2f GHz\n", what, A,
           MA ? A / n_mfma : 0.0, B, MB ? B / n_b : 0.0, ms * 1e3, (A > B ? A : B) / (ms * 1e-3) * 1e-9);
}

int main() {
    float* d; long long* c;
    hipMalloc(&d, (size_t)(1024 * 1024 + 256 * 65536) * 4); hipMalloc(&c, 256 * 8 * 8);
    const int NM = 4096, NB = 4096;
    run<1, 0>("MFMA chain alone", d, c, NM, NB);
    run<0, 1>("v_fma_f32 alone", d, c, NM, NB);
    run<0, 2>("v_pk_fma_f32 alone", d, c, NM, NB);
    run<0, 3>("ds_write_b32 alone", d, c, NM, NB);
    run<0, 4>("global_store_dword alone", d, c, NM, NB);
    run<1, 1>("MFMA chain + v_fma_f32", d, c, NM, NB);
    run<1, 2>("MFMA chain + v_pk_fma_f32", d, c, NM, NB);
    run<1, 3>("MFMA chain + ds_write_b32", d, c, NM, NB);
    run<1, 4>("MFMA chain + global_store_dword", d, c, NM, NB);
    run<1, 1>("MFMA chain + 512 v_fma_f32", d, c, NM, 512);
    run<1, 5>("MFMA chain + 512 v_fma_f32 at s_setprio 3", d, c, NM, 512);
    run<1, 5>("MFMA chain + v_fma_f32 at s_setprio 3", d, c, NM, NB);
    run<1, 6>("MFMA chain + 512 stores at s_setprio 3", d, c, NM, 512);
    run<1, 4>("MFMA chain + 512 stores", d, c, NM, 512);
    run_same<1, 0>("(nothing)", d, c, NM);
    run_same<1, 2>("v_fma_f32", d, c, NM); run_same<1, 4>("v_fma_f32", d, c, NM); run_same<1, 8>("v_fma_f32", d, c, NM);
    run_same<3, 1>("ds_write_b32", d, c, NM); run_same<3, 2>("ds_write_b32", d, c, NM); run_same<3, 4>("ds_write_b32", d, c, NM);
    run_same<8, 1>("ds_read_b128", d, c, NM); run_same<8, 2>("ds_read_b128", d, c, NM); run_same<8, 4>("ds_read_b128", d, c, NM);
    run_same<4, 1>("global store dword", d, c, NM); run_same<4, 2>("global store dword", d, c, NM);
    run_same<7, 1>("global load dword", d, c, NM); run_same<7, 2>("global load dword", d, c, NM);
    return 0;
}
