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

extern const char* g_gad_last_kernel;     // geometry.hip: name of the kernel family the last GEMM entry point launched
#define GAD_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        if ((name)[0] == 'g' && (name)[4] == '_') g_gad_last_kernel = name;             \
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
// the timing slot armed by gad_timing_slot() for the NEXT launch of the calling thread (NULL if none); consumed once
unsigned long long* gad_take_timing_slot();
// the same, carrying the wavefront issue priority gad_stream_priority() gave `stream` in bits 0..1 of the pointer (KTimer
// strips them; the slot itself is 8-byte aligned): kernels that construct a KTimer raise their wavefronts with s_setprio
unsigned long long* gad_take_timing_slot(void* stream);
int gad_take_grid_rows();       // rows the caller expects to be live (0: unknown) -- sizes the grid of the next tile launch

// Train-mode BatchNorm finalisation of a layer (internal argument blocks of bn_finalize_kernel / bn_bwd_coef_kernel /
// the pool finalisation; not part of the C ABI).
struct gad_bn_fin {
    const double* stat_sum;    // (GAD_STAT_REPLICAS, stat_stride) accumulators, this layer's first channel
    const double* stat_sq;
    int32_t stat_stride;
    double count;              // rows behind the statistics (padded duplicates included)
    const float* gamma;
    const float* beta;
    float eps;
    float momentum;
    float* running_mean;       // nullable (pass overlapped with another pass of the same network)
    float* running_var;
    float* scale;              // outputs: scale = gamma*istd, shift = beta - mean*scale
    float* shift;
    float* mean;               // nullable
    float* istd;               // nullable
};
struct gad_bn_bwd {
    const double* dbeta;       // (GAD_STAT_REPLICAS, stat_stride)
    const double* dgamma;
    int32_t stat_stride;
    double count;
    const float* mean;         // saved by the forward pass
    const float* istd;
    double* gacc_gamma;        // nullable
    double* gacc_beta;
    int32_t accumulate;
};

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

// Start / end stamps of every wavefront of a launch (include/gaddpg.h: gad_timing_slot) on the constant-rate wall clock,
// written with plain 8-byte stores into the wavefront's own entry of the slot (same-address atomics from ~2000 workgroups
// cost ~25 ns each: a 25 us kernel took 200 us with an atomic min/max pair).  The host takes min(start) / max(end): the
// quantity a profiler reports as the dispatch duration, measurable inside an untraced, multi-stream run.
struct KTimer {
    unsigned long long* p;
    __device__ __forceinline__ explicit KTimer(unsigned long long* q0) : p(nullptr) {
        const unsigned pr = (unsigned)(reinterpret_cast<size_t>(q0) & 3);          // workgroup-uniform (kernel argument)
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
        unsigned long long* q = reinterpret_cast<unsigned long long*>(reinterpret_cast<size_t>(q0) & ~(size_t)7);
        if (q && (threadIdx.x & 63) == 0) {
            const unsigned blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if (w < GAD_TIMING_WAVES) { p = q + 2 * (size_t)w; p[0] = (unsigned long long)wall_clock64(); }
        }
    }
    __device__ __forceinline__ ~KTimer() {
        if (p) p[1] = (unsigned long long)wall_clock64();
    }
};

// Train-mode BatchNorm finalisation of ONE channel from the replicated f64 statistics (gad_bn_finalize's arithmetic; also
// evaluated by the pool finalisation for its channel slice).  `writer` (exactly one thread of the grid per channel)
// publishes the vectors the backward pass reads and applies the running-statistics momentum update.
__device__ __forceinline__ void gad_bn_fin_channel(const gad_bn_fin& b, int c, bool writer, float& sc, float& sh) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < GAD_STAT_REPLICAS; ++r) {
        s1 += b.stat_sum[(size_t)r * b.stat_stride + c];
        s2 += b.stat_sq[(size_t)r * b.stat_stride + c];
    }
    const double mean = s1 / b.count;
    double var = s2 / b.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)b.eps));
    sc = b.gamma[c] * istd;
    sh = b.beta[c] - (float)mean * sc;
    if (writer) {
        b.scale[c] = sc;
        b.shift[c] = sh;
        if (b.mean) b.mean[c] = (float)mean;
        if (b.istd) b.istd[c] = istd;
        if (b.running_mean) b.running_mean[c] = (1.f - b.momentum) * b.running_mean[c] + b.momentum * (float)mean;
        if (b.running_var) {
            const double unbiased = b.count > 1.0 ? var * b.count / (b.count - 1.0) : var;
            b.running_var[c] = (1.f - b.momentum) * b.running_var[c] + b.momentum * (float)unbiased;
        }
    }
}

// BatchNorm-backward coefficients of ONE channel (gad_bn_bwd_coef's arithmetic): dZ = P*dY - w*(Q + S*z).
// `writer` (one thread of the grid per channel, only when b.accumulate) adds dgamma / dbeta to the gradient arena.
__device__ __forceinline__ void gad_bn_bwd_channel(const gad_bn_bwd& b, const float* scale, int c, bool writer, float& P,
                                                   float& Q, float& S) {
    double db = 0.0, dg = 0.0;
#pragma unroll
    for (int r = 0; r < GAD_STAT_REPLICAS; ++r) {
        db += b.dbeta[(size_t)r * b.stat_stride + c];
        dg += b.dgamma[(size_t)r * b.stat_stride + c];
    }
    const double sc = scale[c], is = b.istd[c], mu = b.mean[c];
    P = (float)sc;
    Q = (float)(sc * (db - mu * is * dg) / b.count);
    S = (float)(sc * is * dg / b.count);
    if (writer && b.accumulate) {
        if (b.gacc_gamma) b.gacc_gamma[c] += dg;
        if (b.gacc_beta) b.gacc_beta[c] += db;
    }
}
#endif
