// Small per-layer kernels around the GEMMs: train-mode BatchNorm finalisation (forward and
// backward coefficients), segment max-pool with arg-max, pooled-gradient statistics.
// Reference arithmetic replaced: torch BatchNorm2d/1d (train) and F.max_pool2d(kernel=[1,nsample])
// inside upstream _PointnetSAModuleBase.forward / reference core/networks.py:84-91.
#include "common.hpp"

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ssum,
                                                          const double* __restrict__ ssq, int stride,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C, double count,
                                                          float eps, float momentum, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mean_o,
                                                          float* __restrict__ istd_o) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < GAD_STAT_REPLICAS; ++r) { s1 += ssum[(size_t)r * stride + c]; s2 += ssq[(size_t)r * stride + c]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * istd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (mean_o) mean_o[c] = (float)mean;
    if (istd_o) istd_o[c] = istd;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

extern "C" int gad_bn_finalize(const double* stat_sum, const double* stat_sq, int stat_stride, const float* gamma,
                               const float* beta, int C, double count, float eps, float momentum,
                               float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                               float* istd, void* stream) {
    GAD_REQUIRE(stat_sum && stat_sq && gamma && beta && scale && shift, GAD_ERR_NULL, "bn_finalize: null pointer");
    GAD_REQUIRE(C >= 1 && count >= 1.0, GAD_ERR_SHAPE, "bn_finalize: bad shape");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stat_sum,
                       stat_sq, stat_stride, gamma, beta, C, count, eps, momentum, running_mean, running_var, scale, shift, mean,
                       istd);
    GAD_CHECK_LAUNCH("bn_finalize");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void bn_eval_affine_kernel(const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ rmean,
                                                             const float* __restrict__ rvar, int C, float eps,
                                                             float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

extern "C" int gad_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, int C, float eps, float* scale, float* shift,
                                  void* stream) {
    GAD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, GAD_ERR_NULL, "bn_eval_affine: null pointer");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, C, eps, scale, shift);
    GAD_CHECK_LAUNCH("bn_eval_affine");
    return GAD_OK;
}

// running statistics update from SAVED batch statistics (mean, 1/sqrt(var+eps)) -- used when the pass that produced
// them ran concurrently with another pass of the same network on a second stream and the momentum updates have to be
// applied afterwards in the reference's order.  count[c] = rows behind channel c's statistics.
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ istd,
                                                                const float* __restrict__ count, int C, float eps,
                                                                float momentum, float* __restrict__ rmean,
                                                                float* __restrict__ rvar) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double is = istd[c], n = count[c];
    double var = 1.0 / (is * is) - (double)eps;
    if (var < 0.0) var = 0.0;
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
}

extern "C" int gad_bn_running_update(const float* mean, const float* istd, const float* count, int C, float eps,
                                     float momentum, float* running_mean, float* running_var, void* stream) {
    GAD_REQUIRE(mean && istd && count && running_mean && running_var, GAD_ERR_NULL, "bn_running_update: null pointer");
    if (C <= 0) return GAD_OK;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, istd, count,
                       C, eps, momentum, running_mean, running_var);
    GAD_CHECK_LAUNCH("bn_running_update");
    return GAD_OK;
}

// out[g][c] = max over the group's rows of relu(scale*z+shift); arg-max = first maximal row
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ z, int z_pitch, int C,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const int32_t* __restrict__ off, long long total,
                                                           float* __restrict__ out, int32_t* __restrict__ argmax) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int g = (int)(q / C), c = (int)(q - (long long)g * C);
    const int r0 = off[g], r1 = off[g + 1];
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    float best = -1.f;
    int arg = r0;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {            // four independent loads in flight per thread (the loop is load-latency bound)
        const float* p = z + (size_t)r * z_pitch + c;
        const float z0 = p[0], z1 = p[z_pitch], z2 = p[2 * (size_t)z_pitch], z3 = p[3 * (size_t)z_pitch];
        const float y0 = fmaxf(fmaf(z0, sc, sh), 0.f), y1 = fmaxf(fmaf(z1, sc, sh), 0.f);
        const float y2 = fmaxf(fmaf(z2, sc, sh), 0.f), y3 = fmaxf(fmaf(z3, sc, sh), 0.f);
        if (y0 > best) { best = y0; arg = r; }
        if (y1 > best) { best = y1; arg = r + 1; }
        if (y2 > best) { best = y2; arg = r + 2; }
        if (y3 > best) { best = y3; arg = r + 3; }
    }
    for (; r < r1; ++r) {
        const float y = fmaxf(fmaf(z[(size_t)r * z_pitch + c], sc, sh), 0.f);
        if (y > best) { best = y; arg = r; }
    }
    out[q] = best < 0.f ? 0.f : best;
    if (argmax) argmax[q] = arg;
}

extern "C" int gad_segment_pool(const float* z, int z_pitch, int C, const float* scale, const float* shift,
                                const int32_t* grp_off, int G, float* out, int32_t* argmax, void* stream) {
    GAD_REQUIRE(z && grp_off && out, GAD_ERR_NULL, "segment_pool: null pointer");
    const long long total = (long long)G * C;
    if (total == 0) return GAD_OK;
    hipLaunchKernelGGL(segment_pool_kernel, dim3(gad_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, z, z_pitch,
                       C, scale, shift, grp_off, total, out, argmax);
    GAD_CHECK_LAUNCH("segment_pool");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ z, int z_pitch, int C,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu, long long total,
                                                         float* __restrict__ out, int out_pitch) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int r = (int)(q / C), c = (int)(q - (long long)r * C);
    float v = z[(size_t)r * z_pitch + c];
    if (scale) v = fmaf(v, scale[c], shift[c]);
    if (relu) v = fmaxf(v, 0.f);
    out[(size_t)r * out_pitch + c] = v;
}

extern "C" int gad_affine_act(const float* z, int z_pitch, int rows, int C, const float* scale, const float* shift,
                              int relu, float* out, int out_pitch, void* stream) {
    GAD_REQUIRE(z && out, GAD_ERR_NULL, "affine_act: null pointer");
    const long long total = (long long)rows * C;
    if (total == 0) return GAD_OK;
    hipLaunchKernelGGL(affine_act_kernel, dim3(gad_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, z, z_pitch, C,
                       scale, shift, relu, total, out, out_pitch);
    GAD_CHECK_LAUNCH("affine_act");
    return GAD_OK;
}

// dbeta[c] += sum_g dout[g][c]*[y*>0],  dgamma[c] += sum_g dout[g][c]*[y*>0]*xhat*  (* = arg-max row)
__global__ __launch_bounds__(256) void pool_bwd_stats_kernel(const float* __restrict__ dout,
                                                             const int32_t* __restrict__ argmax, int G, int C,
                                                             const float* __restrict__ z, int z_pitch,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ istd,
                                                             double* __restrict__ dbeta,
                                                             double* __restrict__ dgamma, int stride) {
    const int cpb = C < 256 ? C : 256;          // channels per block (C is a multiple of 32)
    const int gl = 256 / cpb;                   // groups processed side by side
    const int c = blockIdx.x * cpb + threadIdx.x % cpb;
    if (c >= C) return;
    const int gstride = gridDim.y * gl;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = istd[c];
    float sb = 0.f, sg = 0.f;
    for (int g = blockIdx.y * gl + threadIdx.x / cpb; g < G; g += gstride) {
        const float v = dout[(size_t)g * C + c];
        const int r = argmax[(size_t)g * C + c];
        const float zp = z[(size_t)r * z_pitch + c];
        if (fmaf(zp, sc, sh) > 0.f) { sb += v; sg = fmaf(v, (zp - mu) * is, sg); }
    }
    const int rep = blockIdx.y % GAD_STAT_REPLICAS;
    atomic_add_f64(dbeta + (size_t)rep * stride + c, (double)sb);
    atomic_add_f64(dgamma + (size_t)rep * stride + c, (double)sg);
}

extern "C" int gad_pool_bwd_stats(const float* dout, const int32_t* argmax, int G, int C, const float* z,
                                  int z_pitch, const float* scale, const float* shift, const float* mean,
                                  const float* istd, double* dbeta, double* dgamma, int stat_stride, void* stream) {
    GAD_REQUIRE(dout && argmax && z && scale && shift && mean && istd && dbeta && dgamma, GAD_ERR_NULL,
                "pool_bwd_stats: null pointer");
    GAD_REQUIRE(C % 32 == 0 && (C <= 256 ? 256 % C == 0 : C % 256 == 0), GAD_ERR_SHAPE, "pool_bwd_stats: C=%d", C);
    if (G == 0) return GAD_OK;
    const int cpb = C < 256 ? C : 256, gl = 256 / cpb;
    int gy = gad_cdiv(G, gl * 4);
    if (gy > 512) gy = 512;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(C / cpb, gy), dim3(256), 0, (hipStream_t)stream, dout, argmax, G, C,
                       z, z_pitch, scale, shift, mean, istd, dbeta, dgamma, stat_stride);
    GAD_CHECK_LAUNCH("pool_bwd_stats");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const double* __restrict__ dbeta,
                                                          const double* __restrict__ dgamma, int stride,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ istd, int C, double count,
                                                          float* __restrict__ P, float* __restrict__ Q,
                                                          float* __restrict__ S, double* __restrict__ gg,
                                                          double* __restrict__ gb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double db = 0.0, dg = 0.0;
    for (int r = 0; r < GAD_STAT_REPLICAS; ++r) { db += dbeta[(size_t)r * stride + c]; dg += dgamma[(size_t)r * stride + c]; }
    const double sc = scale[c], is = istd[c], mu = mean[c];
    P[c] = (float)sc;
    Q[c] = (float)(sc * (db - mu * is * dg) / count);
    S[c] = (float)(sc * is * dg / count);
    if (gg) gg[c] += dg;
    if (gb) gb[c] += db;
}

extern "C" int gad_bn_bwd_coef(const double* dbeta, const double* dgamma, int stat_stride, const float* scale, const float* mean,
                               const float* istd, int C, double count, float* coefP, float* coefQ, float* coefS,
                               double* gacc_gamma, double* gacc_beta, void* stream) {
    GAD_REQUIRE(dbeta && dgamma && scale && mean && istd && coefP && coefQ && coefS, GAD_ERR_NULL, "bn_bwd_coef: null pointer");
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dbeta, dgamma, stat_stride, scale,
                       mean, istd, C, count, coefP, coefQ, coefS, gacc_gamma, gacc_beta);
    GAD_CHECK_LAUNCH("bn_bwd_coef");
    return GAD_OK;
}
