// Shared host/device helpers for libgaddpg (gfx950 only: wave64, no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gaddpg.h"

#define GAD_WAVE 64

void gad_set_error(const char* fmt, ...);

#define GAD_REQUIRE(cond, code, ...)     \
    do {                                 \
        if (!(cond)) {                   \
            gad_set_error(__VA_ARGS__);  \
            return (code);               \
        }                                \
    } while (0)

#define GAD_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            gad_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return GAD_ERR_LAUNCH;                                                      \
        }                                                                               \
    } while (0)

static inline int gad_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
#ifdef __HIPCC__
__device__ __forceinline__ int gad_cdiv_dev(int a, int b) { return (a + b - 1) / b; }
#endif

void gad_geometry_set_option(const char* name, int value, int* found);

#ifdef __HIPCC__
// squared distance with the evaluation order pinned to the oracle's (oracle/pn2_ref.c sqdist):
// every product and sum individually rounded, no FMA contraction.
__device__ __forceinline__ float gad_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    // hipcc contracts a*b+c into an FMA by default, and the __fmul_rn / __fadd_rn of its headers are plain operators compiled
    // with that default: only operators written under this pragma stay individually rounded
#pragma clang fp contract(off)
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// |p|^2 with the same pinned order (FPS skip rule, oracle/pn2_ref.c: (m0 + m1) + m2)
__device__ __forceinline__ float gad_sqnorm(float x, float y, float z) {
#pragma clang fp contract(off)
    float xx = x * x, yy = y * y, zz = z * z;
    return (xx + yy) + zz;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// f64 / f32 device-scope atomic adds (hardware global_atomic_add_f64 / _f32 on gfx950)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
#endif
