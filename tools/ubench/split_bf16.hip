// micro-benchmark / accuracy probe (NOT used by the library): an FP32 layer GEMM Z = X . W^T (rows x 64 -> 128, the shape of SA1 layer 3)
//   (a) on v_mfma_f32_32x32x2_f32 (what libgaddpg does: exact f32 products, 157 TFLOP/s peak),
//   (b) as SPLIT bf16 MFMAs: every operand = hi + mid + lo bf16 terms (8 + 8 + 8 significand bits, residuals exact in f32),
//       the 6 products of weight >= 2^-16 accumulated in the f32 accumulator by v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate),
//   (c) the 3-product variant (hi.hi + hi.mid + mid.hi: ~16 significand bits),
// same streaming structure (W staged once in LDS, 32-row slabs per wavefront, operands straight into MFMA registers), timing per
// launch and error against an f64 host reference.  DESIGN.md "where the next factor would have to come from".
// build: hipcc -O3 --offload-arch=gfx950 -w split_bf16.hip -o split_bf16.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define K 64
#define N 128

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {      // {bf16(hi), bf16(lo)}, round to nearest even
    unsigned r;
    __asm__("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 8 floats -> hi / mid / lo bf16x8 (packed pairs); the residuals are exact f32 subtractions
__device__ __forceinline__ void split8(const float* v, u32x4& H, u32x4& M, u32x4& L) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = v[2 * p], b = v[2 * p + 1];
        const unsigned h = cvt_pk(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        H[p] = h; M[p] = m; L[p] = cvt_pk(sa, sb);
    }
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 u) { return *reinterpret_cast<bf16x8*>(&u); }

__device__ __forceinline__ int acc_row(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

// (a) f32 MFMA
__global__ __launch_bounds__(512, 2) void k_f32(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Z, int rows) {
    constexpr int PW = K + 4;
    __shared__ __attribute__((aligned(16))) float Ws[N * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int u = tid; u < N * K / 4; u += 512) {
        const int n = u / (K / 4), c = (u % (K / 4)) * 4;
        *reinterpret_cast<float4*>(Ws + n * PW + c) = *reinterpret_cast<const float4*>(W + n * K + c);
    }
    __syncthreads();
    const int n_slabs = (rows + 31) >> 5;
    for (int slab = wave * gridDim.x + blockIdx.x; slab < n_slabs; slab += gridDim.x * 8) {
        const int r = min(slab * 32 + l31, rows - 1);
        float4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const float4*>(X + (size_t)r * K + 8 * j + 4 * half);
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const float4*>(Ws + (t * 32 + l31) * PW + 8 * j + 4 * half);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b[t].w, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int rr = slab * 32 + acc_row(v, half);
                if (rr < rows) Z[(size_t)rr * N + t * 32 + l31] = acc[t][v];
            }
    }
}

// (b) / (c) split bf16: NP = 6 or 3 products
template <int NP>
__global__ __launch_bounds__(512, 2) void k_split(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Z, int rows) {
    constexpr int PB = K * 2 + 16;                        // bytes per row of a bf16 W plane (16-byte pad)
    __shared__ __attribute__((aligned(16))) unsigned char Wp[3 * N * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int u = tid; u < N * K / 8; u += 512) {          // W: three bf16 planes, once per workgroup
        const int n = u / (K / 8), c = (u % (K / 8)) * 8;
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(W + n * K + c);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(W + n * K + c + 4);
        u32x4 H, M, L;
        split8(v, H, M, L);
        *reinterpret_cast<u32x4*>(Wp + (0 * N + n) * PB + c * 2) = H;
        *reinterpret_cast<u32x4*>(Wp + (1 * N + n) * PB + c * 2) = M;
        *reinterpret_cast<u32x4*>(Wp + (2 * N + n) * PB + c * 2) = L;
    }
    __syncthreads();
    const int n_slabs = (rows + 31) >> 5;
    for (int slab = wave * gridDim.x + blockIdx.x; slab < n_slabs; slab += gridDim.x * 8) {
        const int r = min(slab * 32 + l31, rows - 1);
        u32x4 AH[4], AM[4], AL[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {                      // lane (row, half): X[row][16 s + 8 half .. + 7]
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(X + (size_t)r * K + 16 * s + 8 * half);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(X + (size_t)r * K + 16 * s + 8 * half + 4);
            split8(v, AH[s], AM[s], AL[s]);
        }
        f32x16 acc[4], acn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[t][v] = 0.f; acn[t][v] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (NP == 12 && (s & 1)) {                     // odd steps accumulate the NEGATED products in a second accumulator set
#pragma unroll
                for (int p = 0; p < 4; ++p) { AH[s][p] ^= 0x80008000u; AM[s][p] ^= 0x80008000u; AL[s][p] ^= 0x80008000u; }
            }
            f32x16 (&ac)[4] = (NP == 12 && (s & 1)) ? acn : acc;
            u32x4 BH[4], BM[4], BL[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = (t * 32 + l31) * PB + (16 * s + 8 * half) * 2;
                BH[t] = *reinterpret_cast<const u32x4*>(Wp + 0 * N * PB + off);
                BM[t] = *reinterpret_cast<const u32x4*>(Wp + 1 * N * PB + off);
                if (NP >= 6) BL[t] = *reinterpret_cast<const u32x4*>(Wp + 2 * N * PB + off);
            }
            // smallest terms first
            if (NP >= 6) {
#pragma unroll
                for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AL[s]), as_bf(BH[t]), ac[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AH[s]), as_bf(BL[t]), ac[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AM[s]), as_bf(BM[t]), ac[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AM[s]), as_bf(BH[t]), ac[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AH[s]), as_bf(BM[t]), ac[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(AH[s]), as_bf(BH[t]), ac[t], 0, 0, 0);
        }
        if (NP == 12) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] -= acn[t][v];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int rr = slab * 32 + acc_row(v, half);
                if (rr < rows) Z[(size_t)rr * N + t * 32 + l31] = acc[t][v];
            }
    }
}

template <typename F>
float time_us(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 50.f;
}

int main() {
    const int rows = 213034;
    std::vector<float> hX((size_t)rows * K), hW(N * K), hZ((size_t)rows * N);
    srand(7);
    for (auto& v : hX) v = fmaxf(0.f, (float)rand() / RAND_MAX * 4.f - 1.5f);          // relu-like activations
    for (auto& v : hW) v = ((float)rand() / RAND_MAX - 0.5f) * 0.25f;
    float *X, *W, *Z;
    hipMalloc(&X, hX.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&Z, hZ.size() * 4);
    hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    const int sample = 4096;
    std::vector<double> ref((size_t)sample * N);
    double scale = 0;
    for (int i = 0; i < sample; ++i) {
        const int r = (int)((long long)i * rows / sample);
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hX[(size_t)r * K + k] * hW[n * K + k];
            ref[(size_t)i * N + n] = s;
            scale = fmax(scale, fabs(s));
        }
    }
    auto check = [&](const char* what, float us) {
        hipMemcpy(hZ.data(), Z, hZ.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0, ssum = 0;
        for (int i = 0; i < sample; ++i) {
            const int r = (int)((long long)i * rows / sample);
            for (int n = 0; n < N; ++n) {
                const double e = fabs(hZ[(size_t)r * N + n] - ref[(size_t)i * N + n]);
                emax = fmax(emax, e); esum += e; ssum += hZ[(size_t)r * N + n] - ref[(size_t)i * N + n];
            }
        }
        printf("%-46s %7.1f us per launch   error vs f64: max %.3e  mean |e| %.3e  mean SIGNED %+.3e   (of max |z| = %.3f: %.2e / %.2e / %+.2e)\n",
               what, us, emax, esum / ((double)sample * N), ssum / ((double)sample * N), scale, emax / scale,
               esum / ((double)sample * N) / scale, ssum / ((double)sample * N) / scale);
    };
    hipMemset(Z, 0, hZ.size() * 4);
    float us = time_us([&] { hipLaunchKernelGGL(k_f32, dim3(256), dim3(512), 0, 0, X, W, Z, rows); });
    check("f32 MFMA (v_mfma_f32_32x32x2_f32)", us);
    hipMemset(Z, 0, hZ.size() * 4);
    us = time_us([&] { hipLaunchKernelGGL(k_split<6>, dim3(256), dim3(512), 0, 0, X, W, Z, rows); });
    check("split bf16, 6 products (24 significand bits)", us);
    hipMemset(Z, 0, hZ.size() * 4);
    us = time_us([&] { hipLaunchKernelGGL(k_split<3>, dim3(256), dim3(512), 0, 0, X, W, Z, rows); });
    check("split bf16, 3 products (16 significand bits)", us);
    hipMemset(Z, 0, hZ.size() * 4);
    us = time_us([&] { hipLaunchKernelGGL(k_split<12>, dim3(256), dim3(512), 0, 0, X, W, Z, rows); });
    check("split bf16, 6 products, +/- accumulator pair", us);
    return 0;
}
