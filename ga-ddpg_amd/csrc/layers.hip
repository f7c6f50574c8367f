// Small per-layer kernels around the GEMMs: train-mode BatchNorm finalisation (forward and
// backward coefficients), segment max-pool with arg-max, pooled-gradient statistics.
// Reference arithmetic replaced: torch BatchNorm2d/1d (train) and F.max_pool2d(kernel=[1,nsample])
// inside upstream _PointnetSAModuleBase.forward / reference core/networks.py:84-91.
#include "common.hpp"
#include <string.h>

__global__ __launch_bounds__(256) void bn_finalize_kernel(gad_bn_fin b, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sc, sh;
    gad_bn_fin_channel(b, c, true, sc, sh);
}

extern "C" int gad_bn_finalize(const double* stat_sum, const double* stat_sq, int stat_stride, const float* gamma,
                               const float* beta, int C, double count, float eps, float momentum,
                               float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                               float* istd, void* stream) {
    GAD_REQUIRE(stat_sum && stat_sq && gamma && beta && scale && shift, GAD_ERR_NULL, "bn_finalize: null pointer");
    GAD_REQUIRE(C >= 1 && count >= 1.0, GAD_ERR_SHAPE, "bn_finalize: bad shape");
    gad_bn_fin b;
    b.stat_sum = stat_sum; b.stat_sq = stat_sq; b.stat_stride = stat_stride; b.count = count; b.gamma = gamma; b.beta = beta;
    b.eps = eps; b.momentum = momentum; b.running_mean = running_mean; b.running_var = running_var; b.scale = scale;
    b.shift = shift; b.mean = mean; b.istd = istd;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, b, C);
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

// Segment max-pool with arg-max: out[g][c] = max over the group's rows of relu(scale*z+shift), arg-max = FIRST maximal
// row (torch max_pool2d's tie rule over upstream's slot order).  A group's rows are contiguous (CSR), a row is C
// consecutive floats: one wavefront per group, lane = (row phase, 16-byte channel quad), so every load is a coalesced
// 16-byte access of consecutive rows; the row phases of a quad are folded with wavefront shuffles
// (value first, then the smaller row index) -- north_star's "wavefront shuffle reductions for the max-pool".
// QPR = quads per row handled per lane-row = min(C/4, 64); lanes cover RPW = 64/QPR rows per step and NQ = C/(4*QPR)
// quads each.
template <int QPR, int NQ>
__global__ __launch_bounds__(256) void segment_pool_kernel(const float* __restrict__ z, int z_pitch,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const int32_t* __restrict__ off, int G,
                                                           float* __restrict__ out, int32_t* __restrict__ argmax,
                                                           unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int C = 4 * QPR * NQ, RPW = 64 / QPR;
    __shared__ __attribute__((aligned(16))) float sv[C], tv[C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < C; i += 256) {
        float sc = 1.f, sh = 0.f;
        if (scale) { sc = scale[i]; sh = shift[i]; }
        sv[i] = sc; tv[i] = sh;
    }
    __syncthreads();
    const int sub = lane / QPR, q = lane % QPR;
    float4 s4[NQ], t4[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        s4[k] = *reinterpret_cast<const float4*>(sv + 4 * (q + k * QPR));
        t4[k] = *reinterpret_cast<const float4*>(tv + 4 * (q + k * QPR));
    }
    for (int g = blockIdx.x * 4 + wave; g < G; g += gridDim.x * 4) {
        const int r0 = off[g], r1 = off[g + 1];
        float best[NQ][4];
        int arg[NQ][4];
#pragma unroll
        for (int k = 0; k < NQ; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { best[k][e] = -1.f; arg[k][e] = r0; }
        auto upd = [&](int k, int r, const float4& zz) {
            const float y0 = fmaxf(fmaf(zz.x, s4[k].x, t4[k].x), 0.f), y1 = fmaxf(fmaf(zz.y, s4[k].y, t4[k].y), 0.f);
            const float y2 = fmaxf(fmaf(zz.z, s4[k].z, t4[k].z), 0.f), y3 = fmaxf(fmaf(zz.w, s4[k].w, t4[k].w), 0.f);
            if (y0 > best[k][0]) { best[k][0] = y0; arg[k][0] = r; }
            if (y1 > best[k][1]) { best[k][1] = y1; arg[k][1] = r; }
            if (y2 > best[k][2]) { best[k][2] = y2; arg[k][2] = r; }
            if (y3 > best[k][3]) { best[k][3] = y3; arg[k][3] = r; }
        };
        int r = r0 + sub;
        for (; r + 3 * RPW < r1; r += 4 * RPW) {             // four rows per lane in flight
            float4 za[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NQ; ++k)
                    za[u][k] = *reinterpret_cast<const float4*>(z + (size_t)(r + u * RPW) * z_pitch + 4 * (q + k * QPR));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NQ; ++k) upd(k, r + u * RPW, za[u][k]);
        }
        for (; r < r1; r += RPW) {
#pragma unroll
            for (int k = 0; k < NQ; ++k)
                upd(k, r, *reinterpret_cast<const float4*>(z + (size_t)r * z_pitch + 4 * (q + k * QPR)));
        }
        // fold the RPW row phases of every quad: larger value wins, on equal values the smaller row index
#pragma unroll
        for (int o = 32; o >= QPR; o >>= 1) {
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ob = __shfl_xor(best[k][e], o, 64);
                    const int oa = __shfl_xor(arg[k][e], o, 64);
                    const bool take = ob > best[k][e] || (ob == best[k][e] && oa < arg[k][e]);
                    best[k][e] = take ? ob : best[k][e];
                    arg[k][e] = take ? oa : arg[k][e];
                }
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const size_t o4 = (size_t)g * C + 4 * (q + k * QPR);
                *reinterpret_cast<float4*>(out + o4) = make_float4(fmaxf(best[k][0], 0.f), fmaxf(best[k][1], 0.f),
                                                                  fmaxf(best[k][2], 0.f), fmaxf(best[k][3], 0.f));
                if (argmax) *reinterpret_cast<int4*>(argmax + o4) = make_int4(arg[k][0], arg[k][1], arg[k][2], arg[k][3]);
            }
        }
    }
}

extern "C" int gad_segment_pool(const float* z, int z_pitch, int C, const float* scale, const float* shift,
                                const int32_t* grp_off, int G, float* out, int32_t* argmax, void* stream) {
    unsigned long long* ts = gad_take_timing_slot(stream);
    GAD_REQUIRE(z && grp_off && out, GAD_ERR_NULL, "segment_pool: null pointer");
    GAD_REQUIRE((scale == nullptr) == (shift == nullptr), GAD_ERR_NULL, "segment_pool: scale and shift come together");
    GAD_REQUIRE(z_pitch % 4 == 0, GAD_ERR_SHAPE, "segment_pool: row pitch %d must be a multiple of 4", z_pitch);
    if (G <= 0 || C <= 0) return GAD_OK;
    int gx = gad_cdiv(G, 4);
    if (gx > 2048) gx = 2048;
#define LAUNCH_POOL(QPR, NQ)                                                                                      \
    hipLaunchKernelGGL((segment_pool_kernel<QPR, NQ>), dim3(gx), dim3(256), 0, (hipStream_t)stream, z, z_pitch, scale, \
                       shift, grp_off, G, out, argmax, ts)
    switch (C) {
        case 8: LAUNCH_POOL(2, 1); break;
        case 16: LAUNCH_POOL(4, 1); break;
        case 32: LAUNCH_POOL(8, 1); break;
        case 64: LAUNCH_POOL(16, 1); break;
        case 128: LAUNCH_POOL(32, 1); break;
        case 256: LAUNCH_POOL(64, 1); break;
        case 512: LAUNCH_POOL(64, 2); break;
        case 1024: LAUNCH_POOL(64, 4); break;
        default:
            GAD_REQUIRE(false, GAD_ERR_SHAPE, "segment_pool: C=%d (supported: 8, 16, 32, 64, 128, 256, 512, 1024)", C);
    }
#undef LAUNCH_POOL
    GAD_CHECK_LAUNCH("segment_pool");
    return GAD_OK;
}

// Finish of the max-pool that gad_gemm_fwd folded into the pooled layer's epilogue (gad_gemm_fwd_args.pool_key): per
// (group, channel) the packed key holds max over the group's rows of sgn(gamma) * z and the first row attaining it.
// This kernel (1) finalises the layer's train-mode BatchNorm for its 64-channel slice from the f64 statistics -- the
// arithmetic of gad_bn_finalize; the workgroups with blockIdx.x == 0 publish scale / shift / mean / istd and apply the
// running-statistics update -- or takes scale / shift as given (eval mode), (2) decodes the keys: zmax = sgn * value,
// out = relu(scale * zmax + shift), arg-max = the key's row where out > 0, else the group's FIRST row (every row ties at
// 0 after the ReLU: torch's max_pool2d keeps the first), and (3) resets the keys to 0 for the next pass.
// Thread layout: 16 channel quads x 16 groups per step; a group's 64-channel slice is 512 contiguous key bytes.
__device__ __forceinline__ float pool_unord(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

__global__ __launch_bounds__(256) void pool_finalize_kernel(unsigned long long* __restrict__ key, int C, int G,
                                                            const int32_t* __restrict__ grp_off, gad_bn_fin bn,
                                                            float* __restrict__ out, int32_t* __restrict__ argmax,
                                                            float* __restrict__ zmax) {
    __shared__ __attribute__((aligned(16))) float sv[64], tv[64], gv[64];
    const int tid = threadIdx.x, c0 = blockIdx.y * 64;
    if (tid < 64 && c0 + tid < C) {
        const int c = c0 + tid;
        float sc, sh;
        if (bn.stat_sum) gad_bn_fin_channel(bn, c, blockIdx.x == 0, sc, sh);
        else { sc = bn.scale[c]; sh = bn.shift[c]; }
        sv[tid] = sc; tv[tid] = sh; gv[tid] = bn.gamma[c] < 0.f ? -1.f : 1.f;
    }
    __syncthreads();
    const int q = tid & 15, gl = tid >> 4;
    const int c = c0 + 4 * q;
    if (c >= C) return;
    const float4 s4 = *reinterpret_cast<const float4*>(sv + 4 * q), t4 = *reinterpret_cast<const float4*>(tv + 4 * q);
    const float4 n4 = *reinterpret_cast<const float4*>(gv + 4 * q);
    const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {t4.x, t4.y, t4.z, t4.w}, sg[4] = {n4.x, n4.y, n4.z, n4.w};
    for (int g = blockIdx.x * 16 + gl; g < G; g += gridDim.x * 16) {
        const int r0 = grp_off[g];
        unsigned long long* kp = key + (size_t)g * C + c;
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(kp), k23 = *reinterpret_cast<const ulonglong2*>(kp + 2);
        const unsigned long long k[4] = {k01.x, k01.y, k23.x, k23.y};
        float y[4], zm[4];
        int a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool any = k[e] != 0ull;                       // (a group without rows cannot occur: max(cnt, 1) rows each)
            const float z = any ? sg[e] * pool_unord((unsigned)(k[e] >> 32)) : 0.f;
            const float yy = any ? fmaxf(fmaf(z, sc[e], sh[e]), 0.f) : 0.f;
            y[e] = yy; zm[e] = z;
            a[e] = (any && yy > 0.f && sc[e] != 0.f) ? (int)(0xffffffffu - (unsigned)(k[e] & 0xffffffffull)) : r0;
        }
        const size_t o4 = (size_t)g * C + c;
        *reinterpret_cast<float4*>(out + o4) = make_float4(y[0], y[1], y[2], y[3]);
        if (argmax) *reinterpret_cast<int4*>(argmax + o4) = make_int4(a[0], a[1], a[2], a[3]);
        if (zmax) *reinterpret_cast<float4*>(zmax + o4) = make_float4(zm[0], zm[1], zm[2], zm[3]);
        const ulonglong2 zero = {0ull, 0ull};
        *reinterpret_cast<ulonglong2*>(kp) = zero;
        *reinterpret_cast<ulonglong2*>(kp + 2) = zero;
    }
}

extern "C" int gad_pool_finalize(uint64_t* key, int C, int G, const int32_t* grp_off, const double* stat_sum,
                                 const double* stat_sq, int stat_stride, double count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 float* scale, float* shift, float* mean, float* istd, float* out, int32_t* argmax,
                                 float* zmax, void* stream) {
    GAD_REQUIRE(key && grp_off && gamma && scale && shift && out, GAD_ERR_NULL, "pool_finalize: null pointer");
    GAD_REQUIRE(C >= 4 && C % 4 == 0, GAD_ERR_SHAPE, "pool_finalize: C=%d must be a positive multiple of 4", C);
    GAD_REQUIRE(!stat_sum || (stat_sq && beta && count >= 1.0), GAD_ERR_NULL, "pool_finalize: statistics need stat_sq, beta, count");
    if (G <= 0) return GAD_OK;
    gad_bn_fin b;
    b.stat_sum = stat_sum; b.stat_sq = stat_sq; b.stat_stride = stat_stride; b.count = count; b.gamma = gamma; b.beta = beta;
    b.eps = eps; b.momentum = momentum; b.running_mean = running_mean; b.running_var = running_var; b.scale = scale;
    b.shift = shift; b.mean = mean; b.istd = istd;
    int gx = gad_cdiv(G, 16);                     // one 16-group step per workgroup (four steps per workgroup: -1.2 % steps/s --
                                                  // the launch is latency-bound, more workgroups = more loads in flight)
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(pool_finalize_kernel, dim3(gx, gad_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(key), C, G, grp_off, b, out, argmax, zmax);
    GAD_CHECK_LAUNCH("pool_finalize");
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

// batched transpose through a 32 x 33 LDS tile: coalesced 128-byte row segments on both sides
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C,
                                                                int src_pitch, long long src_batch, int dst_pitch, long long dst_batch) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const float* s = src + (size_t)blockIdx.z * src_batch;
    float* d = dst + (size_t)blockIdx.z * dst_batch;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty + 8 * u, j = j0 + tx;
        if (i < R && j < C) tile[ty + 8 * u][tx] = s[(size_t)i * src_pitch + j];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = j0 + ty + 8 * u, i = i0 + tx;
        if (i < R && j < C) d[(size_t)j * dst_pitch + i] = tile[tx][ty + 8 * u];
    }
}

extern "C" int gad_transpose_batched(const float* src, float* dst, int B, int R, int C, int src_pitch, long long src_batch,
                                     int dst_pitch, long long dst_batch, void* stream) {
    GAD_REQUIRE(src && dst, GAD_ERR_NULL, "transpose_batched: null pointer");
    GAD_REQUIRE(B >= 0 && R >= 0 && C >= 0 && src_pitch >= C && dst_pitch >= R && B <= 65535, GAD_ERR_SHAPE,
                "transpose_batched: B=%d R=%d C=%d pitches %d / %d", B, R, C, src_pitch, dst_pitch);
    if (B == 0 || R == 0 || C == 0) return GAD_OK;
    GAD_REQUIRE(gad_cdiv(R, 32) <= 65535, GAD_ERR_SHAPE, "transpose_batched: R=%d too large", R);
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(gad_cdiv(C, 32), gad_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, src, dst,
                       R, C, src_pitch, src_batch, dst_pitch, dst_batch);
    GAD_CHECK_LAUNCH("transpose_batched");
    return GAD_OK;
}

// dbeta[c] += sum_g dout[g][c]*[y*>0],  dgamma[c] += sum_g dout[g][c]*[y*>0]*xhat*  (* = arg-max row)
__global__ __launch_bounds__(256) void pool_bwd_stats_kernel(float* __restrict__ dout,
                                                             const int32_t* __restrict__ argmax, int G, int C,
                                                             const float* __restrict__ z, int z_pitch,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ istd,
                                                             double* __restrict__ dbeta,
                                                             double* __restrict__ dgamma, int stride, int mask,
                                                             const float* __restrict__ zmax) {
    const int cpb = C < 256 ? C : 256;          // channels per block (C is a multiple of 32)
    const int gl = 256 / cpb;                   // groups processed side by side
    const int c = blockIdx.x * cpb + threadIdx.x % cpb;
    if (c >= C) return;
    const int gstride = gridDim.y * gl;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = istd[c];
    float sb = 0.f, sg = 0.f;
    // four groups per round: their loads are independent and in flight together (one group per round left every thread
    // waiting a full memory latency per 8 bytes: 11.7 us for SA1's 8 MB)
    const bool gather = !(zmax && !(sc == 0.f && z && argmax));
    for (int g0 = blockIdx.y * gl + threadIdx.x / cpb; g0 < G; g0 += 4 * gstride) {
        float v[4], zp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * gstride;
            const size_t o = (size_t)(g < G ? g : g0) * C + c;
            v[u] = dout[o];
            // the winner's raw value, saved by gad_pool_finalize: no gather.  A channel with scale == 0 (gamma == 0) is constant
            // over the rows: every row ties and the arg-max is the group's FIRST row (torch's max_pool2d), not the key's row
            // whose value zmax holds -- there the routed row's value is gathered, so that dgamma sees the same x_hat as the reference
            zp[u] = gather ? z[(size_t)argmax[o] * z_pitch + c] : zmax[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * gstride;
            if (g >= G) break;
            if (fmaf(zp[u], sc, sh) > 0.f) { sb += v[u]; sg = fmaf(v[u], (zp[u] - mu) * is, sg); }
            else if (mask) dout[(size_t)g * C + c] = 0.f;
        }
    }
    const int rep = blockIdx.y % GAD_STAT_REPLICAS;
    atomic_add_f64(dbeta + (size_t)rep * stride + c, (double)sb);
    atomic_add_f64(dgamma + (size_t)rep * stride + c, (double)sg);
}

extern "C" int gad_pool_bwd_stats(float* dout, const int32_t* argmax, int G, int C, const float* z,
                                  int z_pitch, const float* scale, const float* shift, const float* mean,
                                  const float* istd, double* dbeta, double* dgamma, int stat_stride, int mask_in_place,
                                  const float* zmax, void* stream) {
    GAD_REQUIRE(dout && (zmax || (argmax && z)) && scale && shift && mean && istd && dbeta && dgamma, GAD_ERR_NULL,
                "pool_bwd_stats: null pointer");
    GAD_REQUIRE(C % 32 == 0 && (C <= 256 ? 256 % C == 0 : C % 256 == 0), GAD_ERR_SHAPE, "pool_bwd_stats: C=%d", C);
    if (G == 0) return GAD_OK;
    const int cpb = C < 256 ? C : 256, gl = 256 / cpb;
    int gy = gad_cdiv(G, gl * 4);
    if (gy > 512) gy = 512;                       // (1024 / 2048 workgroups: -0.5 / -1 % steps/s -- more same-address atomics)
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(C / cpb, gy), dim3(256), 0, (hipStream_t)stream, dout, argmax, G, C,
                       z, z_pitch, scale, shift, mean, istd, dbeta, dgamma, stat_stride, mask_in_place, zmax);
    GAD_CHECK_LAUNCH("pool_bwd_stats");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(gad_bn_bwd b, const float* __restrict__ scale, int C,
                                                          float* __restrict__ P, float* __restrict__ Q,
                                                          float* __restrict__ S) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float p, q, sv;
    gad_bn_bwd_channel(b, scale, c, true, p, q, sv);
    P[c] = p; Q[c] = q; S[c] = sv;
}

extern "C" int gad_bn_bwd_coef(const double* dbeta, const double* dgamma, int stat_stride, const float* scale, const float* mean,
                               const float* istd, int C, double count, float* coefP, float* coefQ, float* coefS,
                               double* gacc_gamma, double* gacc_beta, void* stream) {
    GAD_REQUIRE(dbeta && dgamma && scale && mean && istd && coefP && coefQ && coefS, GAD_ERR_NULL, "bn_bwd_coef: null pointer");
    gad_bn_bwd b;
    b.dbeta = dbeta; b.dgamma = dgamma; b.stat_stride = stat_stride; b.count = count; b.mean = mean; b.istd = istd;
    b.gacc_gamma = gacc_gamma; b.gacc_beta = gacc_beta; b.accumulate = 1;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(gad_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, b, scale, C, coefP, coefQ,
                       coefS);
    GAD_CHECK_LAUNCH("bn_bwd_coef");
    return GAD_OK;
}
