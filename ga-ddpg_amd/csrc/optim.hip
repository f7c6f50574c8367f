// Optimiser / bookkeeping kernels over flat parameter buffers.
// Reference arithmetic replaced: torch.optim.Adam.step x4 (core/utils.py:969-970,993-994,
// 183-237), clip_grad_norm_ (core/ddpg.py:141), soft/half-soft/half-hard target updates
// (core/utils.py:750-774), module_max_param / module_max_gradient (core/utils.py:92-108).
// One launch per optimiser instead of ~10 small kernels per parameter tensor; the updated value is
// also mirrored into the packed, padded compute layout the GEMMs read (no separate repack pass).
#include "common.hpp"

__global__ __launch_bounds__(256) void grad_from_arena_kernel(const double* __restrict__ gacc,
                                                              const int32_t* __restrict__ m2p, int n,
                                                              float* __restrict__ grad, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = m2p[i];
    const float g = j >= 0 ? (float)gacc[j] : 0.f;
    grad[i] = accumulate ? grad[i] + g : g;
}

// the same conversion + the sum of squares of the resulting gradient (clip_grad_norm_'s total norm): one launch fewer on the
// critic phase's chain than gad_grad_from_arena followed by gad_sumsq
__global__ __launch_bounds__(256) void grad_from_arena_sumsq_kernel(const double* __restrict__ gacc, const int32_t* __restrict__ m2p, int n,
                                                                    float* __restrict__ grad, int accumulate, double* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int j = m2p[i];
        const float g0 = j >= 0 ? (float)gacc[j] : 0.f;
        const float g = accumulate ? grad[i] + g0 : g0;
        grad[i] = g;
        s += (double)g * (double)g;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f64(out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int gad_grad_from_arena_sumsq(const double* gacc, const int32_t* m2p, int n, float* grad, int accumulate, double* sumsq,
                                         void* stream) {
    GAD_REQUIRE(gacc && m2p && grad && sumsq, GAD_ERR_NULL, "grad_from_arena_sumsq: null pointer");
    if (n <= 0) return GAD_OK;
    int gx = gad_cdiv(n, 1024);
    gx = gx < 1 ? 1 : gx;
    hipLaunchKernelGGL(grad_from_arena_sumsq_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, gacc, m2p, n, grad, accumulate, sumsq);
    GAD_CHECK_LAUNCH("grad_from_arena_sumsq");
    return GAD_OK;
}

extern "C" int gad_grad_from_arena(const double* gacc, const int32_t* m2p, int n, float* grad, int accumulate,
                                   void* stream) {
    GAD_REQUIRE(gacc && m2p && grad, GAD_ERR_NULL, "grad_from_arena: null pointer");
    if (n <= 0) return GAD_OK;
    hipLaunchKernelGGL(grad_from_arena_kernel, dim3(gad_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gacc, m2p, n,
                       grad, accumulate);
    GAD_CHECK_LAUNCH("grad_from_arena");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int n, double* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const double v = g[i]; s += v * v; }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f64(out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int gad_sumsq(const float* grad, int n, double* out, void* stream) {
    GAD_REQUIRE(grad && out, GAD_ERR_NULL, "sumsq: null pointer");
    if (n <= 0) return GAD_OK;
    int gx = gad_cdiv(n, 256 * 8);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(sumsq_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, grad, n, out);
    GAD_CHECK_LAUNCH("sumsq");
    return GAD_OK;
}

// out[s] = max |x| over segment s.  Non-negative floats order like their bit patterns.
__global__ __launch_bounds__(256) void absmax_segments_kernel(const float* __restrict__ x,
                                                              const int32_t* __restrict__ seg_off,
                                                              float* __restrict__ out) {
    const int s = blockIdx.y;
    const int lo = seg_off[s], hi = seg_off[s + 1];
    float m = 0.f;
    for (int i = lo + blockIdx.x * 256 + threadIdx.x; i < hi; i += gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(out + s), __float_as_uint(m));
}

extern "C" int gad_absmax_segments(const float* x, const int32_t* seg_off, int n_seg, float* out, void* stream) {
    GAD_REQUIRE(x && seg_off && out, GAD_ERR_NULL, "absmax_segments: null pointer");
    if (n_seg <= 0) return GAD_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * n_seg, st);
    (void)e;
    hipLaunchKernelGGL(absmax_segments_kernel, dim3(64, n_seg), dim3(256), 0, st, x, seg_off, out);
    GAD_CHECK_LAUNCH("absmax_segments");
    return GAD_OK;
}

// hyper: {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, sqrt(1-beta2^t), grad_scale}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const uint8_t* __restrict__ active,
                                                   const int32_t* __restrict__ m2p, float* __restrict__ packed, int n,
                                                   const float* __restrict__ hyper,
                                                   const double* __restrict__ clip_sumsq, float clip_max) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (active && !active[i]) return;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5],
                sbc2 = hyper[6];
    float coef = hyper[7];
    if (clip_sumsq) {
        const float c = clip_max / ((float)sqrt(*clip_sumsq) + 1e-6f);     // torch clip_grad_norm_
        coef *= c < 1.f ? c : 1.f;
    }
    float g = grad[i] * coef;
    if (clip_sumsq) grad[i] = g;                                           // torch scales .grad in place
    const float pv = p[i];
    g = fmaf(wd, pv, g);
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sbc2 + eps;
    const float np = pv - (lr / bc1) * (mi / denom);
    p[i] = np;
    if (packed) { const int j = m2p[i]; if (j >= 0) packed[j] = np; }
}

extern "C" int gad_adam_step(float* p, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* active,
                             const int32_t* m2p, float* packed, int n, const float* hyper, const double* clip_sumsq,
                             float clip_max, void* stream) {
    GAD_REQUIRE(p && grad && exp_avg && exp_avg_sq && hyper, GAD_ERR_NULL, "adam_step: null pointer");
    GAD_REQUIRE(!packed || m2p, GAD_ERR_NULL, "adam_step: packed mirror needs m2p");
    if (n <= 0) return GAD_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(gad_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p,
                       const_cast<float*>(grad), exp_avg, exp_avg_sq, active, m2p, packed, n, hyper, clip_sumsq,
                       clip_max);
    GAD_CHECK_LAUNCH("adam_step");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ t, const float* __restrict__ s,
                                                     const uint8_t* __restrict__ sel, const int32_t* __restrict__ m2p,
                                                     float* __restrict__ packed, int n, float tau, int hard_enable) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = sel ? sel[i] : 1;
    float nv;
    if (k == 1) nv = t[i] * (1.f - tau) + s[i] * tau;
    else if (k == 2 && hard_enable) nv = s[i];
    else return;
    t[i] = nv;
    if (packed) { const int j = m2p[i]; if (j >= 0) packed[j] = nv; }
}

extern "C" int gad_polyak(float* target, const float* source, const uint8_t* sel, const int32_t* m2p,
                          float* target_packed, int n, float tau, int hard_enable, void* stream) {
    GAD_REQUIRE(target && source, GAD_ERR_NULL, "polyak: null pointer");
    GAD_REQUIRE(!target_packed || m2p, GAD_ERR_NULL, "polyak: packed mirror needs m2p");
    if (n <= 0) return GAD_OK;
    hipLaunchKernelGGL(polyak_kernel, dim3(gad_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, target, source, sel, m2p,
                       target_packed, n, tau, hard_enable);
    GAD_CHECK_LAUNCH("polyak");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// One launch for an optimiser phase: up to GAD_MAX_OPTIM_JOBS flat buffers (blockIdx.y = job), each optionally
//   gradient arena (f64, packed order) -> .grad (f32, master order)           [gad_grad_from_arena]
//   Adam step with in-kernel clip_grad_norm_ scaling, packed mirror            [gad_adam_step]
//   polyak / hard update of a target network FROM THE UPDATED value, mirror    [gad_polyak]
//   max |parameter| / max |gradient| into log slots (non-negative float bits)  [gad_absmax_segments]
//   a BatchNorm num_batches_tracked bump.
// The step used to end in ~12 launches of 5-10 us each on its critical chain (core/agent.py:192-259: optimize, target
// updates, log_stat); the arithmetic per element is unchanged.
// ------------------------------------------------------------------------------------------------
struct OptimJobs { gad_optim_job j[GAD_MAX_OPTIM_JOBS]; };

__global__ __launch_bounds__(256) void optim_jobs_kernel(OptimJobs jobs) {
    const gad_optim_job& J = jobs.j[blockIdx.y];
    const int n = J.n;
    if (blockIdx.x == 0 && threadIdx.x < J.counter_n && J.counter) J.counter[threadIdx.x] += J.counter_add;
    float lr = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, wd = 0.f, bc1 = 1.f, sbc2 = 1.f, coef = 1.f;
    const bool adam = J.hyper != nullptr;
    if (adam) {
        lr = J.hyper[0]; b1 = J.hyper[1]; b2 = J.hyper[2]; eps = J.hyper[3]; wd = J.hyper[4]; bc1 = J.hyper[5]; sbc2 = J.hyper[6];
        coef = J.hyper[7];
        if (J.clip_sumsq) {
            const float c = J.clip_max / ((float)sqrt(*J.clip_sumsq) + 1e-6f);     // torch clip_grad_norm_
            coef *= c < 1.f ? c : 1.f;
        }
    }
    float amax_p = 0.f, amax_g = 0.f;
    // one element: everything it reads arrives in registers (g, pv, m, v, the packed index jm, the arena value ga, the
    // target's value tv / selector k / packed index jt), everything it writes leaves through the flags -- so the body is the
    // same arithmetic for the scalar tail and for the 4-wide main loop
    struct Out { float g, pv, m, v, tv; bool w_g, w_p, w_t; };
    auto elem = [&](float g, float pv, float m, float v, int jm, double ga64, bool act, float tv, int k) {
        Out o;
        o.w_g = false; o.w_p = false; o.w_t = false;
        if (J.gacc) {
            const float ga = jm >= 0 ? (float)ga64 : 0.f;
            g = J.accumulate ? g + ga : ga;
            o.w_g = true;
        }
        if (adam && act) {
            g *= coef;
            if (J.clip_sumsq) o.w_g = true;                                        // torch scales .grad in place
            float gw = fmaf(wd, pv, g);
            const float mi = b1 * m + (1.f - b1) * gw;
            const float vi = b2 * v + (1.f - b2) * gw * gw;
            m = mi; v = vi;
            const float denom = sqrtf(vi) / sbc2 + eps;
            pv = pv - (lr / bc1) * (mi / denom);
            o.w_p = true;
        }
        if (J.target) {
            if (k == 1) { tv = tv * (1.f - J.tau) + pv * J.tau; o.w_t = true; }
            else if (k == 2 && J.hard_enable) { tv = pv; o.w_t = true; }
        }
        o.g = g; o.pv = pv; o.m = m; o.v = v; o.tv = tv;
        amax_p = fmaxf(amax_p, fabsf(pv));
        amax_g = fmaxf(amax_g, fabsf(g));
        return o;
    };
    // main loop: four consecutive elements per thread, every stream a 16-byte access (the launch is latency-bound: with one
    // 4-byte element per thread 2 M parameters took 38 us = 1.6 TB/s of its 28 bytes per parameter)
    const int n4 = n >> 2;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
        const int i = q << 2;
        int4 jm4 = make_int4(-1, -1, -1, -1);
        if (J.m2p) jm4 = *reinterpret_cast<const int4*>(J.m2p + i);
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), p4 = g4, m4 = g4, v4 = g4, t4 = g4;
        if (J.grad && !(J.gacc && !J.accumulate)) g4 = *reinterpret_cast<const float4*>(J.grad + i);
        if (J.p) p4 = *reinterpret_cast<const float4*>(J.p + i);
        if (adam) { m4 = *reinterpret_cast<const float4*>(J.exp_avg + i); v4 = *reinterpret_cast<const float4*>(J.exp_avg_sq + i); }
        double ga[4] = {0.0, 0.0, 0.0, 0.0};
        const int jm[4] = {jm4.x, jm4.y, jm4.z, jm4.w};
        if (J.gacc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ga[e] = J.gacc[jm[e] >= 0 ? jm[e] : 0];
        }
        uchar4 act4 = make_uchar4(1, 1, 1, 1);
        if (J.active) act4 = *reinterpret_cast<const uchar4*>(J.active + i);
        int k4[4] = {1, 1, 1, 1};
        int4 jt4 = make_int4(-1, -1, -1, -1);
        if (J.target) {
            t4 = *reinterpret_cast<const float4*>(J.target + i);
            if (J.target_sel) { const uchar4 s4 = *reinterpret_cast<const uchar4*>(J.target_sel + i); k4[0] = s4.x; k4[1] = s4.y; k4[2] = s4.z; k4[3] = s4.w; }
            if (J.target_packed) jt4 = *reinterpret_cast<const int4*>(J.target_m2p + i);
        }
        const float gi[4] = {g4.x, g4.y, g4.z, g4.w}, pi[4] = {p4.x, p4.y, p4.z, p4.w}, mi[4] = {m4.x, m4.y, m4.z, m4.w};
        const float vi[4] = {v4.x, v4.y, v4.z, v4.w}, ti[4] = {t4.x, t4.y, t4.z, t4.w};
        const bool ai[4] = {act4.x != 0, act4.y != 0, act4.z != 0, act4.w != 0};
        const int jt[4] = {jt4.x, jt4.y, jt4.z, jt4.w};
        Out o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = elem(gi[e], pi[e], mi[e], vi[e], jm[e], ga[e], ai[e], ti[e], k4[e]);
        // (the write flags of the four elements agree except for `active` and the target selector: per-element stores there)
        if (J.grad && (o[0].w_g || o[1].w_g || o[2].w_g || o[3].w_g))
            *reinterpret_cast<float4*>(J.grad + i) = make_float4(o[0].w_g ? o[0].g : gi[0], o[1].w_g ? o[1].g : gi[1], o[2].w_g ? o[2].g : gi[2], o[3].w_g ? o[3].g : gi[3]);
        if (adam && (o[0].w_p || o[1].w_p || o[2].w_p || o[3].w_p)) {
            *reinterpret_cast<float4*>(J.exp_avg + i) = make_float4(o[0].m, o[1].m, o[2].m, o[3].m);
            *reinterpret_cast<float4*>(J.exp_avg_sq + i) = make_float4(o[0].v, o[1].v, o[2].v, o[3].v);
            *reinterpret_cast<float4*>(J.p + i) = make_float4(o[0].pv, o[1].pv, o[2].pv, o[3].pv);
            if (J.packed) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (o[e].w_p && jm[e] >= 0) J.packed[jm[e]] = o[e].pv;
            }
        }
        if (J.target && (o[0].w_t || o[1].w_t || o[2].w_t || o[3].w_t)) {
            *reinterpret_cast<float4*>(J.target + i) = make_float4(o[0].tv, o[1].tv, o[2].tv, o[3].tv);
            if (J.target_packed) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (o[e].w_t && jt[e] >= 0) J.target_packed[jt[e]] = o[e].tv;
            }
        }
    }
    // scalar tail (n % 4 elements), first workgroup
    for (int i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int jm = J.m2p ? J.m2p[i] : -1;
        const float g0 = (J.grad && !(J.gacc && !J.accumulate)) ? J.grad[i] : 0.f;
        const float p0 = J.p ? J.p[i] : 0.f;
        const float m0 = adam ? J.exp_avg[i] : 0.f, v0 = adam ? J.exp_avg_sq[i] : 0.f;
        const double ga = (J.gacc && jm >= 0) ? J.gacc[jm] : 0.0;
        const bool act = !(J.active && !J.active[i]);
        const float t0 = J.target ? J.target[i] : 0.f;
        const int k = J.target ? (J.target_sel ? J.target_sel[i] : 1) : 0;
        const Out o = elem(g0, p0, m0, v0, jm, ga, act, t0, k);
        if (o.w_g && J.grad) J.grad[i] = o.g;
        if (o.w_p) {
            J.exp_avg[i] = o.m; J.exp_avg_sq[i] = o.v; J.p[i] = o.pv;
            if (J.packed && jm >= 0) J.packed[jm] = o.pv;
        }
        if (o.w_t) {
            J.target[i] = o.tv;
            if (J.target_packed) { const int jt = J.target_m2p[i]; if (jt >= 0) J.target_packed[jt] = o.tv; }
        }
    }
    // statistics: one atomic per workgroup that had elements, spread over GAD_ABSMAX_SLOTS addresses (a same-address
    // device atomic costs ~25 ns: 4096 wavefronts on one slot were 0.1 ms)
    if ((J.absmax_p || J.absmax_grad) && (int)(blockIdx.x * 1024) < n) {
        __shared__ float red[2][4];
        amax_p = wave_max(amax_p);
        amax_g = wave_max(amax_g);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = amax_p; red[1][threadIdx.x >> 6] = amax_g; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int slot = blockIdx.x % GAD_ABSMAX_SLOTS;
            // look before the atomic (a relaxed device-scope load): once a few workgroups have posted, most find their
            // maximum already covered and skip the same-address atomic
            auto post = [&](float* base, float v) {
                unsigned* p = reinterpret_cast<unsigned int*>(base) + slot;
                const unsigned bits = __float_as_uint(v);
                if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
            };
            if (J.absmax_p) post(J.absmax_p, fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
            if (J.absmax_grad) post(J.absmax_grad, fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
        }
    }
}

extern "C" int gad_optim_jobs(const gad_optim_job* host_jobs, int n_jobs, void* stream) {
    GAD_REQUIRE(host_jobs && n_jobs >= 1 && n_jobs <= GAD_MAX_OPTIM_JOBS, GAD_ERR_SHAPE, "optim_jobs: 1..%d jobs", GAD_MAX_OPTIM_JOBS);
    OptimJobs jobs;
    int nmax = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const gad_optim_job& J = host_jobs[k];
        GAD_REQUIRE(J.n >= 0, GAD_ERR_SHAPE, "optim_jobs: job %d: n", k);
        GAD_REQUIRE(!J.hyper || (J.p && J.grad && J.exp_avg && J.exp_avg_sq), GAD_ERR_NULL, "optim_jobs: job %d: Adam buffers", k);
        GAD_REQUIRE(!J.gacc || (J.m2p && J.grad), GAD_ERR_NULL, "optim_jobs: job %d: arena conversion needs m2p and grad", k);
        GAD_REQUIRE(!J.packed || J.m2p, GAD_ERR_NULL, "optim_jobs: job %d: packed mirror needs m2p", k);
        GAD_REQUIRE(!J.target || (J.p && (!J.target_packed || J.target_m2p)), GAD_ERR_NULL, "optim_jobs: job %d: target update", k);
        GAD_REQUIRE(!J.counter || (J.counter_n >= 0 && J.counter_n <= 256), GAD_ERR_SHAPE, "optim_jobs: job %d: counters", k);
        // the main loop moves four elements per access: float / int32 streams 16-byte aligned, the uint8 masks 4-byte aligned
        const void* p16[] = {J.p, J.grad, J.exp_avg, J.exp_avg_sq, J.m2p, J.target, J.target_m2p};
        for (const void* q : p16) GAD_REQUIRE(((uintptr_t)q & 15) == 0, GAD_ERR_SHAPE, "optim_jobs: job %d: float / int32 buffers must be 16-byte aligned", k);
        GAD_REQUIRE((((uintptr_t)J.active | (uintptr_t)J.target_sel) & 3) == 0, GAD_ERR_SHAPE, "optim_jobs: job %d: mask buffers must be 4-byte aligned", k);
        jobs.j[k] = J;
        nmax = J.n > nmax ? J.n : nmax;
    }
    // one element per thread (memory-latency-bound: 1024 looping workgroups took 25 us for 1.4 M parameters, one round of
    // 5.5 k workgroups 10 us); the statistics' atomics are per workgroup WITH elements and spread over 8 slots
    int gx = gad_cdiv(nmax, 1024);                     // four elements per thread
    gx = gx < 1 ? 1 : (gx > 16384 ? 16384 : gx);
    hipLaunchKernelGGL(optim_jobs_kernel, dim3(gx, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    GAD_CHECK_LAUNCH("optim_jobs");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ p, const int32_t* __restrict__ m2p,
                                                          int n, float* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = m2p[i];
    if (j >= 0) packed[j] = p[i];
}

extern "C" int gad_pack_params(const float* p, const int32_t* m2p, int n, float* packed, void* stream) {
    GAD_REQUIRE(p && m2p && packed, GAD_ERR_NULL, "pack_params: null pointer");
    if (n <= 0) return GAD_OK;
    hipLaunchKernelGGL(pack_params_kernel, dim3(gad_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, m2p, n, packed);
    GAD_CHECK_LAUNCH("pack_params");
    return GAD_OK;
}


// ------------------------------------------------------------------------------------------------
// replay minibatch gather (include/gaddpg.h section F).  HBM-bound: 2 x B x 16.5 KB of cloud rows, copied as
// 8-byte pairs (a cloud row is 4120 floats: 8-byte but not 16-byte aligned); blockIdx.y = sample, the first
// workgroup of each sample also moves its scalar / short-vector fields.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void replay_gather_kernel(gad_replay_gather_args a) {
    const int b = blockIdx.y;
    const long long i = a.idx[b], n = a.nxt[b];
    const int pairs = a.cloud_elems / 2;
    const float2* s0 = reinterpret_cast<const float2*>(a.point_state + (size_t)i * a.cloud_elems);
    const float2* s1 = reinterpret_cast<const float2*>(a.point_state + (size_t)n * a.cloud_elems);
    float2* d0 = reinterpret_cast<float2*>(a.out_point + (size_t)b * a.cloud_elems);
    float2* d1 = reinterpret_cast<float2*>(a.out_next_point + (size_t)b * a.cloud_elems);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < pairs; p += gridDim.x * 256) {
        d0[p] = s0[p];
        if (a.out_next_point) d1[p] = s1[p];
    }
    if (blockIdx.x != 0) return;
    const int t = threadIdx.x;
    if (t < 6) { a.out_action[b * 6 + t] = a.action[i * 6 + t]; a.out_expert_action[b * 6 + t] = a.expert_action[i * 6 + t]; }
    if (t < 7) a.out_goal[b * 7 + t] = a.goal[i * 7 + t];
    if (t == 8) a.out_reward[b] = a.reward[i];
    if (t == 9) a.out_return[b] = a.returns[i];
    if (t == 10) a.out_mask[b] = a.terminal[i];
    if (t == 11) a.out_expert_flag[b] = a.expert_flags[i];
    if (t == 12) a.out_perturb_flag[b] = a.perturb_flags[i];
    if (t == 13) {
        const float tm = a.timestep[a.end[b]] + 1.f - a.timestep[i];     // remaining steps of the episode
        a.out_time[b] = tm;
        a.out_time_m1[b] = tm - 1.f;
    }
}

extern "C" int gad_replay_gather(const gad_replay_gather_args* a, void* stream) {
    GAD_REQUIRE(a && a->idx && a->nxt && a->end && a->point_state && a->out_point, GAD_ERR_NULL, "replay_gather: null pointer");
    GAD_REQUIRE(a->action && a->expert_action && a->goal && a->reward && a->returns && a->terminal && a->timestep &&
                a->expert_flags && a->perturb_flags, GAD_ERR_NULL, "replay_gather: null source");
    GAD_REQUIRE(a->out_action && a->out_expert_action && a->out_goal && a->out_reward && a->out_return && a->out_mask &&
                a->out_time && a->out_time_m1 && a->out_expert_flag && a->out_perturb_flag, GAD_ERR_NULL, "replay_gather: null output");
    GAD_REQUIRE(a->B >= 1 && a->cloud_elems >= 2 && a->cloud_elems % 2 == 0, GAD_ERR_SHAPE, "replay_gather: bad shape");
    hipLaunchKernelGGL(replay_gather_kernel, dim3(4, a->B), dim3(256), 0, (hipStream_t)stream, *a);
    GAD_CHECK_LAUNCH("replay_gather");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// one launch clears up to six buffers (statistics, gradient arenas, scatter targets of a backward pass): each of
// them used to be its own fill kernel at the head / in the middle of the pass's dependency chain
// ------------------------------------------------------------------------------------------------
struct ZeroSegs { void* p[6]; long long n[6]; };   // n: bytes (multiples of 4)

__global__ __launch_bounds__(256) void zero_buffers_kernel(ZeroSegs z) {
    const int sgm = blockIdx.y;
    char* base = static_cast<char*>(z.p[sgm]);
    const long long bytes = z.n[sgm];
    if (!base || bytes <= 0) return;
    const long long head = ((16 - (reinterpret_cast<size_t>(base) & 15)) & 15);          // bytes up to 16-byte alignment
    const long long h = head < bytes ? head : bytes;
    const long long body = (bytes - h) / 16;
    const long long stride = (long long)gridDim.x * 256;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    float4* b16 = reinterpret_cast<float4*>(base + h);
    for (long long i = t; i < body; i += stride) b16[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < h / 4) reinterpret_cast<float*>(base)[t] = 0.f;
    const long long tail0 = h + body * 16;
    if (t < (bytes - tail0) / 4) reinterpret_cast<float*>(base + tail0)[t] = 0.f;
}

extern "C" int gad_zero_buffers(void* p0, long long n0, void* p1, long long n1, void* p2, long long n2, void* p3,
                                long long n3, void* p4, long long n4, void* p5, long long n5, void* stream) {
    ZeroSegs z;
    void* ps[6] = {p0, p1, p2, p3, p4, p5};
    const long long ns[6] = {n0, n1, n2, n3, n4, n5};
    int last = -1;
    long long nmax = 0;
    for (int i = 0; i < 6; ++i) {
        GAD_REQUIRE(ns[i] >= 0 && ns[i] % 4 == 0, GAD_ERR_SHAPE, "zero_buffers: byte count %d must be a non-negative multiple of 4", i);
        GAD_REQUIRE((reinterpret_cast<size_t>(ps[i]) & 3) == 0, GAD_ERR_SHAPE, "zero_buffers: buffer %d is not 4-byte aligned", i);
        z.p[i] = ps[i]; z.n[i] = ps[i] ? ns[i] : 0;
        if (z.n[i] > 0) { last = i; nmax = z.n[i] > nmax ? z.n[i] : nmax; }
    }
    if (last < 0) return GAD_OK;
    long long blocks = (nmax / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(zero_buffers_kernel, dim3((unsigned)blocks, last + 1), dim3(256), 0, (hipStream_t)stream, z);
    GAD_CHECK_LAUNCH("zero_buffers");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// one launch copies up to GAD_COPY_MAX_SEGS device buffers (gaddpg.h: gad_copy_buffers): a device-resident minibatch adopted
// into the step's static input buffers used to be one runtime copy dispatch per key (23 copyBuffer dispatches per step)
// ------------------------------------------------------------------------------------------------
struct CopySegs { gad_copy_seg s[GAD_COPY_MAX_SEGS]; };

__global__ __launch_bounds__(256) void copy_buffers_kernel(CopySegs z) {
    const gad_copy_seg sg = z.s[blockIdx.y];
    char* dst = static_cast<char*>(sg.dst);
    const char* src = static_cast<const char*>(sg.src);
    const long long bytes = sg.bytes;
    if (!dst || !src || bytes <= 0) return;
    const long long stride = (long long)gridDim.x * 256;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool al16 = ((reinterpret_cast<size_t>(dst) | reinterpret_cast<size_t>(src)) & 15) == 0;
    const long long body = al16 ? bytes / 16 : 0;
    if (sg.add == 0.f) {
        const float4* s16 = reinterpret_cast<const float4*>(src);
        float4* d16 = reinterpret_cast<float4*>(dst);
        for (long long i = t; i < body; i += stride) d16[i] = s16[i];
        for (long long i = body * 4 + t; i < bytes / 4; i += stride)
            reinterpret_cast<float*>(dst)[i] = reinterpret_cast<const float*>(src)[i];
    } else {
        for (long long i = t; i < bytes / 4; i += stride)
            reinterpret_cast<float*>(dst)[i] = reinterpret_cast<const float*>(src)[i] + sg.add;
    }
}

extern "C" int gad_copy_buffers(const gad_copy_seg* host_segs, int n_segs, void* stream) {
    GAD_REQUIRE(host_segs || n_segs == 0, GAD_ERR_NULL, "copy_buffers: NULL segment table");
    GAD_REQUIRE(n_segs >= 0 && n_segs <= GAD_COPY_MAX_SEGS, GAD_ERR_SHAPE, "copy_buffers: %d segments (at most %d)", n_segs,
                GAD_COPY_MAX_SEGS);
    CopySegs z = {};
    long long nmax = 0;
    for (int i = 0; i < n_segs; ++i) {
        const gad_copy_seg& g = host_segs[i];
        GAD_REQUIRE(g.bytes >= 0 && g.bytes % 4 == 0, GAD_ERR_SHAPE, "copy_buffers: byte count of segment %d must be a non-negative multiple of 4", i);
        GAD_REQUIRE(((reinterpret_cast<size_t>(g.dst) | reinterpret_cast<size_t>(g.src)) & 3) == 0, GAD_ERR_SHAPE,
                    "copy_buffers: segment %d is not 4-byte aligned", i);
        z.s[i] = g;
        if (g.dst && g.src && g.bytes > nmax) nmax = g.bytes;
    }
    if (nmax == 0) return GAD_OK;
    long long blocks = (nmax / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(copy_buffers_kernel, dim3((unsigned)blocks, n_segs), dim3(256), 0, (hipStream_t)stream, z);
    GAD_CHECK_LAUNCH("copy_buffers");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// split-bf16 mirrors of packed weight matrices (include/gaddpg.h: gad_split_weights).  One thread per (row n, column pair
// k, k + 1): hi / mid / lo by round-to-nearest-even conversions of the exact residuals, forward mirror as packed pairs
// (coalesced 4-byte stores), transposed mirror as 2-byte stores (the matrices are a few hundred KB: the launch is ~4 us).
// A value is negated where its reduction index (k forward, n transposed) lies in an odd block of 16.
// ------------------------------------------------------------------------------------------------
struct SplitLayers { gad_split_layer l[GAD_MAX_SPLIT_LAYERS]; int tile0[GAD_MAX_SPLIT_LAYERS + 1]; int n_layers; };

__device__ __forceinline__ unsigned split_cvt_pk_bf16(float lo, float hi) {      // {bf16(hi) << 16 | bf16(lo)}, RNE
    unsigned r;
    __asm__("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// One workgroup per 32 (rows n) x 32 (columns k) tile of one layer (tiles of all layers in one flat grid).  Thread (n = t >> 3,
// k4 = (t & 7) * 4) loads four consecutive k of row n (16 bytes, coalesced), splits them, stores 8 bytes per plane of the forward
// mirror, and parks the twelve bf16 values in LDS as [plane][k][n]; after the barrier thread (k = t >> 3, n4 = (t & 7) * 4) writes
// 8 bytes per plane of the transposed mirror (coalesced along n).  (First version: 2-byte scattered stores, 13 us per launch on
// the step's critical chain behind each optimiser phase.)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ packed, SplitLayers L, uint16_t* __restrict__ out) {
    __shared__ uint16_t tile[3][32][36];
    int li = 0;
    while (li + 1 < L.n_layers && (int)blockIdx.x >= L.tile0[li + 1]) ++li;
    const gad_split_layer& y = L.l[li];
    const int tk = y.Ks >> 5;
    const int tile_id = blockIdx.x - L.tile0[li];
    const int n0 = (tile_id / tk) * 32, k0 = (tile_id % tk) * 32;
    const long long plane = (long long)y.n_out * y.Ks;
    const int t = threadIdx.x;
    {
        const int n = n0 + (t >> 3), k = k0 + (t & 7) * 4;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < y.n_out) w = *reinterpret_cast<const float4*>(packed + y.w_off + (size_t)n * y.Kp + k);
        unsigned v[3][2];
        {
            const unsigned H = split_cvt_pk_bf16(w.x, w.y);
            const float ra = w.x - __uint_as_float(H << 16), rb = w.y - __uint_as_float(H & 0xffff0000u);
            const unsigned M = split_cvt_pk_bf16(ra, rb);
            v[0][0] = H; v[1][0] = M; v[2][0] = split_cvt_pk_bf16(ra - __uint_as_float(M << 16), rb - __uint_as_float(M & 0xffff0000u));
        }
        {
            const unsigned H = split_cvt_pk_bf16(w.z, w.w);
            const float ra = w.z - __uint_as_float(H << 16), rb = w.w - __uint_as_float(H & 0xffff0000u);
            const unsigned M = split_cvt_pk_bf16(ra, rb);
            v[0][1] = H; v[1][1] = M; v[2][1] = split_cvt_pk_bf16(ra - __uint_as_float(M << 16), rb - __uint_as_float(M & 0xffff0000u));
        }
        const unsigned sk = ((k >> 4) & 1) ? 0x80008000u : 0u;       // forward: reduction index k
        const unsigned sn = ((n >> 4) & 1) ? 0x8000u : 0u;           // transposed: reduction index n
        const int nl = t >> 3, kl = (t & 7) * 4;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (n < y.n_out)
                *reinterpret_cast<uint2*>(out + y.fwd_off + p * plane + (size_t)n * y.Ks + k) = make_uint2(v[p][0] ^ sk, v[p][1] ^ sk);
            tile[p][kl + 0][nl] = (uint16_t)((v[p][0] & 0xffffu) ^ sn);
            tile[p][kl + 1][nl] = (uint16_t)((v[p][0] >> 16) ^ sn);
            tile[p][kl + 2][nl] = (uint16_t)((v[p][1] & 0xffffu) ^ sn);
            tile[p][kl + 3][nl] = (uint16_t)((v[p][1] >> 16) ^ sn);
        }
    }
    __syncthreads();
    {
        const int kl = t >> 3, nl = (t & 7) * 4;
        const int k = k0 + kl, n = n0 + nl;
        if (n < y.n_out) {                                           // (n_out is a multiple of 4 for every mirrored layer: checked by the host)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint2*>(out + y.t_off + p * plane + (size_t)k * y.n_out + n) = *reinterpret_cast<const uint2*>(&tile[p][kl][nl]);
        }
    }
}

extern "C" int gad_split_weights(const float* packed, const gad_split_layer* host_layers, int n_layers, uint16_t* out, void* stream) {
    GAD_REQUIRE(packed && host_layers && out, GAD_ERR_NULL, "split_weights: null pointer");
    GAD_REQUIRE(n_layers >= 1 && n_layers <= GAD_MAX_SPLIT_LAYERS, GAD_ERR_SHAPE, "split_weights: 1..%d layers", GAD_MAX_SPLIT_LAYERS);
    SplitLayers L;
    int tiles = 0;
    for (int i = 0; i < n_layers; ++i) {
        const gad_split_layer& y = host_layers[i];
        GAD_REQUIRE(y.n_out > 0 && y.n_out % 4 == 0 && y.Ks > 0 && y.Ks % 32 == 0 && y.Ks <= y.Kp && y.Kp % 4 == 0 && y.w_off % 4 == 0 &&
                    y.fwd_off % 4 == 0 && y.t_off % 4 == 0 && y.w_off >= 0 && y.fwd_off >= 0 && y.t_off >= 0, GAD_ERR_SHAPE,
                    "split_weights: layer %d: Ks must be a multiple of 32 and <= Kp; n_out, Kp and the offsets multiples of 4", i);
        L.l[i] = y;
        L.tile0[i] = tiles;
        tiles += gad_cdiv(y.n_out, 32) * (y.Ks / 32);
    }
    L.tile0[n_layers] = tiles;
    L.n_layers = n_layers;
    hipLaunchKernelGGL(split_weights_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, packed, L, out);
    GAD_CHECK_LAUNCH("split_weights");
    return GAD_OK;
}
