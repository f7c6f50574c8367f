// Layer-GEMM family for the per-point shared MLPs, the FC head and the actor/critic heads: FP32 results throughout, products either
// on v_mfma_f32_32x32x2_f32 or -- the default for the set-abstraction layers since round 5 -- as split-bf16 term products with f32
// accumulation on v_mfma_f32_32x32x16_bf16 (section "split-bf16 arithmetic" below; library option "mfma_split").
//
// Reference arithmetic replaced: the 1x1 Conv2d / Linear + BatchNorm(train) + ReLU chains that
// upstream build_shared_mlp and reference core/networks.py:84-91,280-300,339-351 run through
// cuDNN/cuBLAS, plus their autograd backward.  Nothing is materialised in the (B,C,npoint,nsample)
// layout: rows are the DE-DUPLICATED (group, point) pairs (include/gaddpg.h section B), weighted by
// their multiplicity in the BatchNorm statistics, which is exactly equivalent to the reference's
// padded neighbourhoods (duplicates change neither a max-pool nor its gradient).
//
// Hardware mapping (gfx950): v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD), 4 wavefronts
// per workgroup, operands staged k-major in LDS ([k][row], pitch = tile+1 or tile+4 so both the
// transposing ds_write_b32 and the fragment ds_read_b32 are bank-conflict free), next K-tile's
// global loads issued before the current tile's MFMAs (register-staged software pipeline).
// Producers (BN affine + ReLU, neighbourhood gather, BN-backward dZ) are fused into the operand load
// and written BRANCH-FREE (clamped addresses + selects): every 16-byte load of a tile is issued
// before the first wait, instead of one s_waitcnt per predicated load.
// BatchNorm statistics: registers -> __shfl_xor -> LDS across wavefronts -> f64 device atomics
// into one of GAD_STAT_REPLICAS accumulators (blockIdx % replicas) to bound same-address contention.
#include "common.hpp"
#include <type_traits>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// cache policy of the streaming kernels' big output stores (buffer-store aux bits: 0 plain, 2 nt, 16 sc1 = write-through)
#ifndef GAD_STREAM_STORE_AUX
#define GAD_STREAM_STORE_AUX 0
#endif

#define KT 32   // reduction tile

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4sel(bool c, float4 a, float4 b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

typedef unsigned gad_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    const gad_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
}
// buffer descriptors address 32-bit UNSIGNED byte offsets: a row tensor may reach 2 GiB (the host-side route predicates bound
// rows x pitch x 4 by 2^31 inclusive, fits_i32_bytes) -- record counts are formed in unsigned arithmetic
#define GAD_BUF_MAX ((int)0x80000000u)
__device__ __forceinline__ int gad_nbytes(int rows, int row_bytes) { return (int)((unsigned)rows * (unsigned)row_bytes); }


// ------------------------------------------------------------------------------------------------
// operand producers.  Every producer is split in two so that global loads overlap the MFMAs:
//   *_raw    issues the 16-byte loads of the NEXT tile into registers (no arithmetic on them),
//   *_finish runs after the current tile's MFMAs, right before the LDS store: BatchNorm affine /
//            ReLU / BatchNorm-backward with the per-channel vectors read from LDS.
// ------------------------------------------------------------------------------------------------
#define VMAX 1024            // max channels of one layer side (per-channel vectors staged in LDS)

struct XSrc {   // how a layer's INPUT row r, column c is produced (gad_gemm_fwd_args subset)
    int mode;
    const float* zin; int zin_pitch; int c_in;
    const float* scale; const float* shift; int relu;
    const float* extra; int ones_col;
    const float* src_xyz; const float* ctr_xyz; const float* feat; int feat_c;
    const float* action; int act_c; int gps;
    const int32_t* row_pt; const int32_t* row_grp;
    int affine;          // ACT input: a per-channel affine (scale / shift) is applied
    const float* pre_W; int pre_Kp;   // mode 2: the ACT input is RECOMPUTED from the gathered first layer (its packed weights)
    gad_bn_fin bn;       // input layer's BatchNorm finalised in this launch's prologue (bn.stat_sum == NULL: scale / shift as given)
};

static XSrc make_xsrc(const gad_gemm_fwd_args& a) {
    XSrc x;
    x.mode = a.mode; x.zin = a.zin; x.zin_pitch = a.zin_pitch; x.c_in = a.c_in;
    x.scale = a.scale; x.shift = a.shift; x.relu = a.relu; x.extra = a.extra; x.ones_col = a.ones_col;
    x.src_xyz = a.src_xyz; x.ctr_xyz = a.ctr_xyz; x.feat = a.feat; x.feat_c = a.feat_c;
    x.action = a.action; x.act_c = a.act_c; x.gps = a.grp_per_sample > 0 ? a.grp_per_sample : 1;
    x.row_pt = a.row_pt; x.row_grp = a.row_grp;
    x.affine = (x.scale && x.shift) ? 1 : 0;
    x.pre_W = a.pre_W; x.pre_Kp = a.pre_Kp;
    x.bn.stat_sum = a.in_stat_sum; x.bn.stat_sq = a.in_stat_sq; x.bn.stat_stride = a.in_stat_stride; x.bn.count = a.in_count;
    x.bn.gamma = a.in_gamma; x.bn.beta = a.in_beta; x.bn.eps = a.in_eps; x.bn.momentum = a.in_momentum;
    x.bn.running_mean = a.in_running_mean; x.bn.running_var = a.in_running_var;
    x.bn.scale = const_cast<float*>(a.scale); x.bn.shift = const_cast<float*>(a.shift); x.bn.mean = a.in_mean; x.bn.istd = a.in_istd;
    return x;
}

// per-channel affine of an ACT input -> LDS
template <int NT>
__device__ __forceinline__ void stage_affine(float* sv, float* tv, const XSrc& x, int off, int n) {
    for (int i = threadIdx.x; i < n; i += NT) { sv[i] = x.scale[off + i]; tv[i] = x.shift[off + i]; }
}

// diagnostic build (-DGAD_X_PHASES=1, tools/ubench_phases.py): the streaming forward kernel writes each wavefront's cycles per
// phase (wait + prefetch | MFMA loop | epilogue) into the launch's timing slot instead of the start / end stamps
#ifdef GAD_X_PHASES
#define GAD_PH_STAMP(t) t = (long long)__builtin_readcyclecounter()
#else
#define GAD_PH_STAMP(t)
#endif
// diagnostic build (-DGAD_W_PHASES, tools/ubench_wphases.py): the wide-tile forward / dX kernels write, per workgroup, the wall
// clock (100 MHz) at the end of each part of their first row tile -- prologue | first tile staged | K loop | stores + statistics |
// column atomics -- into the upper half of the launch's timing slot (entries 8192 + workgroup)
#ifdef GAD_W_PHASES
#define GAD_WPH_DECL long long wph_[6] = {0, 0, 0, 0, 0, 0}; wph_[0] = (long long)wall_clock64()
#define GAD_WPH(i) do { if (wph_[i] == 0) wph_[i] = (long long)wall_clock64(); } while (0)
#define GAD_WPH_STORE(ts)                                                                                                     \
    do {                                                                                                                      \
        unsigned long long* q_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<size_t>(ts) & ~(size_t)7);            \
        const unsigned blk_ = blockIdx.x + gridDim.x * blockIdx.y;                                                            \
        if (q_ && threadIdx.x == 0 && blk_ < 4096) {                                                                          \
            auto c_ = [&](int i) { const long long d_ = wph_[i] ? wph_[i] - wph_[0] : 0; return (unsigned long long)(d_ > 65535 ? 65535 : d_); }; \
            q_[2 * (8192 + blk_)] = (unsigned long long)wph_[0];                                                              \
            q_[2 * (8192 + blk_) + 1] = c_(1) | (c_(2) << 16) | (c_(3) << 32) | (c_(4) << 48);                                \
            q_[2 * (12288 + blk_)] = c_(5);                                                                                    \
        }                                                                                                                     \
    } while (0)
// ... and, summed over the K-tiles of the first row tile, every wavefront's shader cycles in: fragments + MFMAs issued | next tile
// staged + loads issued | barrier wait (entries 4096 + 8 * workgroup + wavefront: {mfma << 32 | stage, barrier})
#define GAD_WKL_DECL long long wkl_[4] = {0, 0, 0, 0}; long long wkt_ = 0, wkp_ = 0
#define GAD_WKL(i)                                                                   \
    do {                                                                             \
        wkt_ = (long long)__builtin_readcyclecounter();                              \
        if (i > 0) wkl_[i] += wkt_ - wkp_;                                           \
        wkp_ = wkt_;                                                                 \
    } while (0)
#define GAD_WKL_STORE(ts)                                                                                                     \
    do {                                                                                                                      \
        unsigned long long* q_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<size_t>(ts) & ~(size_t)7);            \
        const unsigned blk_ = blockIdx.x + gridDim.x * blockIdx.y;                                                            \
        if (q_ && (threadIdx.x & 63) == 0 && blk_ < 512) {                                                                    \
            const unsigned e_ = 4096 + 8 * blk_ + (threadIdx.x >> 6);                                                         \
            q_[2 * e_] = ((unsigned long long)wkl_[1] << 32) | (unsigned long long)(wkl_[2] & 0xffffffffu);                   \
            q_[2 * e_ + 1] = (unsigned long long)wkl_[3];                                                                     \
        }                                                                                                                     \
    } while (0)
#else
#define GAD_WPH_DECL
#define GAD_WPH(i)
#define GAD_WPH_STORE(ts)
#define GAD_WKL_DECL
#define GAD_WKL(i)
#define GAD_WKL_STORE(ts)
#endif
struct XRaw { float4 a; float4 s; };     // a: the 16 raw bytes; s: special columns (tail tiles only)

// number of leading "bulk" columns served by aligned 16-byte loads
__device__ __forceinline__ int x_bulk(const XSrc& x) { return x.mode == 0 ? x.c_in : x.feat_c; }

// ACT input: columns [0,c_in) = act(scale*z+shift) of the previous layer's raw output (c_in % 4 == 0),
//            column c_in = extra[r] (optional), column ones_col = 1, everything else 0.
// GATHER input (packed order, features first): [feat[pt] (feat_c) | src_xyz[pt]-ctr_xyz[grp] (3) |
//            action[grp/gps] (act_c) | 0..].
// `tail` (wave-uniform): this K-tile reaches beyond the bulk columns.  `pt`: point index of the row.
template <int XM>
__device__ __forceinline__ XRaw x_raw(const XSrc& x, int r, bool valid, int zoff, int c, bool tail, int pt, int grp_pre = -1) {
    XRaw o;
    o.s = f4zero();
    const int rr = valid ? r : 0;
    if (XM == 0) {
        const int cc = c < x.c_in ? c : x.c_in - 4;
        o.a = ldg4(x.zin + (size_t)rr * x.zin_pitch + zoff + cc);
        if (tail) {
            float e = 0.f;
            if (x.extra) e = x.extra[rr];
            o.s.x = (c + 0 == x.c_in && x.extra) ? e : (c + 0 == x.ones_col ? 1.f : 0.f);
            o.s.y = (c + 1 == x.c_in && x.extra) ? e : (c + 1 == x.ones_col ? 1.f : 0.f);
            o.s.z = (c + 2 == x.c_in && x.extra) ? e : (c + 2 == x.ones_col ? 1.f : 0.f);
            o.s.w = (c + 3 == x.c_in && x.extra) ? e : (c + 3 == x.ones_col ? 1.f : 0.f);
        }
    } else {
        const int cc = c < x.feat_c ? c : x.feat_c - 4;
        o.a = ldg4(x.feat + (size_t)pt * x.feat_c + cc);
        if (tail) {
            const int grp = grp_pre >= 0 ? grp_pre : x.row_grp[rr];
            const float* p = x.src_xyz + (size_t)pt * 3;
            float q0 = p[0], q1 = p[1], q2 = p[2];
            if (x.ctr_xyz) {
                const float* cp = x.ctr_xyz + (size_t)grp * 3;
                q0 = __fsub_rn(q0, cp[0]); q1 = __fsub_rn(q1, cp[1]); q2 = __fsub_rn(q2, cp[2]);
            }
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            const int t0 = c - x.feat_c;
            if (x.action) {
                const float* ap = x.action + (size_t)(grp / x.gps) * x.act_c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ai = t0 + j - 3;
                    const int aic = ai < 0 ? 0 : (ai >= x.act_c ? x.act_c - 1 : ai);
                    const float av = ap[aic];
                    a[j] = (ai >= 0 && ai < x.act_c) ? av : 0.f;
                }
            }
            float* spv = &o.s.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j;
                spv[j] = t == 0 ? q0 : (t == 1 ? q1 : (t == 2 ? q2 : a[j]));
            }
        }
    }
    return o;
}

// sv/tv: the layer-input BatchNorm scale/shift of this group's channels, staged in LDS
template <int XM>
__device__ __forceinline__ float4 x_finish(const XSrc& x, const XRaw& raw, bool valid, int c, const float* sv,
                                           const float* tv) {
    const int bulk = XM == 0 ? x.c_in : x.feat_c;
    const bool inside = c < bulk;
    float4 v = raw.a;
    if (XM == 0) {
        const int cc = inside ? c : x.c_in - 4;
        if (x.affine) {
            const float4 s = *reinterpret_cast<const float4*>(sv + cc), t = *reinterpret_cast<const float4*>(tv + cc);
            v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
        }
        if (x.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    return f4sel(valid, f4sel(inside, v, raw.s), f4zero());
}

__device__ __forceinline__ void stage_vec(float* dst, const float* src, int off, int n, float fill) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src ? src[off + i] : fill;
}

struct DzSrc {   // gad_dz_src on the device
    const float* z; int z_pitch; const float* scale; const float* shift; int relu;
    const float* P; const float* Q; const float* S; const float* row_w;
    int gmode; const float* G; int g_pitch; const int32_t* argmax; const float* dout;
    const int32_t* row_grp; int c;
    int coef;            // BatchNorm backward applies (P/Q/S given, or formed here from bw)
    int premasked;       // the ReLU mask is already in G / dout
    gad_bn_bwd bw;       // bw.dbeta != NULL: P, Q, S are formed by this launch from the layer's f64 sums
};

static DzSrc make_dzsrc(const gad_dz_src& d) {
    DzSrc s;
    s.z = d.z; s.z_pitch = d.z_pitch; s.scale = d.scale; s.shift = d.shift; s.relu = d.relu;
    s.P = d.coefP; s.Q = d.coefQ; s.S = d.coefS; s.row_w = d.row_w;
    s.gmode = d.gmode; s.G = d.G; s.g_pitch = d.g_pitch; s.argmax = d.argmax; s.dout = d.dout;
    s.row_grp = d.row_grp; s.c = d.c;
    s.premasked = d.premasked;
    s.bw.dbeta = d.bn_dbeta; s.bw.dgamma = d.bn_dgamma; s.bw.stat_stride = d.bn_stride; s.bw.count = d.bn_count;
    s.bw.mean = d.bn_mean; s.bw.istd = d.bn_istd; s.bw.gacc_gamma = d.gacc_gamma; s.bw.gacc_beta = d.gacc_beta;
    s.bw.accumulate = (d.gacc_gamma || d.gacc_beta) ? 1 : 0;
    s.coef = (d.coefP != nullptr || d.bn_dbeta != nullptr) ? 1 : 0;
    return s;
}

// P, Q, S of channel ch (identity when the layer has no BatchNorm)
// `writer`: this thread is the launch's one thread for channel ch that may add dgamma / dbeta to the gradient arena
__device__ __forceinline__ void dz_coef(const DzSrc& d, int ch, float& P, float& Q, float& S, bool writer = false) {
    if (d.bw.dbeta) gad_bn_bwd_channel(d.bw, d.scale, ch, writer, P, Q, S);
    else if (d.P) { P = d.P[ch]; Q = d.Q[ch]; S = d.S[ch]; }
    else { P = 1.f; Q = 0.f; S = 0.f; }
}
__device__ __forceinline__ bool first_workgroup() { return blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0; }

struct DzRaw { float4 z; float4 g; int4 a; };

// scalar (unaligned) evaluation of dZ[r][n..n+3], used for the tiny last layers of the heads
__device__ __forceinline__ float4 dz_scalar4(const DzSrc& d, int r, bool valid, int off, int n, int nmax) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nn = n + j;
        const bool ok = valid && nn < nmax;
        const int rr = ok ? r : 0;
        const int ch = off + (ok ? nn : 0);
        float z = 0.f;
        if (d.z) z = d.z[(size_t)rr * d.z_pitch + ch];
        float g;
        if (d.gmode == 0) {
            g = d.G[(size_t)rr * d.g_pitch + ch];
        } else {
            const int grp = d.row_grp[rr];
            g = d.argmax[(size_t)grp * d.c + ch] == r ? d.dout[(size_t)grp * d.c + ch] : 0.f;
        }
        if (d.relu && !d.premasked) {
            const float y = d.scale ? fmaf(z, d.scale[ch], d.shift[ch]) : z;
            g = y > 0.f ? g : 0.f;
        }
        if (d.P) {
            const float w = d.row_w ? d.row_w[rr] : 1.f;
            g = d.P[ch] * g - w * fmaf(d.S[ch], z, d.Q[ch]);
        }
        v[j] = ok ? g : 0.f;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// raw loads of dZ's ingredients for row r, channels off+n..+3 (VEC: everything 16-byte aligned)
template <bool VEC>
__device__ __forceinline__ DzRaw dz_raw(const DzSrc& d, int r, bool valid, int off, int n, int nmax, int grp) {
    DzRaw o;
    o.z = f4zero(); o.a = make_int4(0, 0, 0, 0);
    if (!VEC) { o.g = dz_scalar4(d, r, valid, off, n, nmax); return o; }
    const int rr = valid ? r : 0;
    const int ch = off + (n < nmax ? n : 0);
    if (d.z) o.z = ldg4(d.z + (size_t)rr * d.z_pitch + ch);
    if (d.gmode == 0) {
        o.g = ldg4(d.G + (size_t)rr * d.g_pitch + ch);
    } else {
        o.a = *reinterpret_cast<const int4*>(d.argmax + (size_t)grp * d.c + ch);
        o.g = ldg4(d.dout + (size_t)grp * d.c + ch);
    }
    return o;
}

// vec: LDS copies of {scale, shift, P, Q, S} of this group's channels (VM floats each; VM = VMAX unless the kernel sizes its LDS for narrower layers)
template <bool VEC, int VM = VMAX>
__device__ __forceinline__ float4 dz_finish(const DzSrc& d, const DzRaw& raw, int r, bool valid, int n, int nmax, float w,
                                            const float* vec) {
    if (!VEC) return raw.g;
    const bool inside = n < nmax;
    const int nn = inside ? n : 0;
    float4 g = raw.g;
    const float4 z = raw.z;
    if (d.gmode != 0) {
        g.x = raw.a.x == r ? g.x : 0.f; g.y = raw.a.y == r ? g.y : 0.f;
        g.z = raw.a.z == r ? g.z : 0.f; g.w = raw.a.w == r ? g.w : 0.f;
    }
    if (d.relu && !d.premasked) {
        float4 y = z;
        if (d.scale) {
            const float4 s = *reinterpret_cast<const float4*>(vec + nn), t = *reinterpret_cast<const float4*>(vec + VM + nn);
            y.x = fmaf(z.x, s.x, t.x); y.y = fmaf(z.y, s.y, t.y); y.z = fmaf(z.z, s.z, t.z); y.w = fmaf(z.w, s.w, t.w);
        }
        g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f;
        g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
    }
    if (d.coef) {
        const float4 P = *reinterpret_cast<const float4*>(vec + 2 * VM + nn);
        const float4 Q = *reinterpret_cast<const float4*>(vec + 3 * VM + nn);
        const float4 S = *reinterpret_cast<const float4*>(vec + 4 * VM + nn);
        g.x = P.x * g.x - w * fmaf(S.x, z.x, Q.x); g.y = P.y * g.y - w * fmaf(S.y, z.y, Q.y);
        g.z = P.z * g.z - w * fmaf(S.z, z.z, Q.z); g.w = P.w * g.w - w * fmaf(S.w, z.w, Q.w);
    }
    return f4sel(valid && inside, g, f4zero());
}

template <int VM = VMAX>
__device__ __forceinline__ void stage_dz_vecs(float* vec, const DzSrc& d, int off, int n) {
    if (!d.premasked) {
        stage_vec(vec, d.scale, off, n, 1.f);
        stage_vec(vec + VM, d.shift, off, n, 0.f);
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        float P, Q, S;
        dz_coef(d, off + i, P, Q, S, first_workgroup());
        vec[2 * VM + i] = P; vec[3 * VM + i] = Q; vec[4 * VM + i] = S;
    }
}

static bool dz_vectorizable(const gad_dz_src& d, const int32_t* off, const int32_t* n_out, int ng) {
    bool ok = (d.z == nullptr || d.z_pitch % 4 == 0) && (d.gmode != 0 || d.g_pitch % 4 == 0) && (d.gmode == 0 || d.c % 4 == 0);
    for (int i = 0; i < ng; ++i) ok = ok && off[i] % 4 == 0 && n_out[i] % 4 == 0 && n_out[i] <= VMAX;
    return ok;
}

// ------------------------------------------------------------------------------------------------
// LDS tile helpers.  Tiles are k-major: T[kk][i], kk in [0,KT), i in [0,DIM).
//   transposing store (source contiguous along kk): pitch DIM+1, four ds_write_b32
//   direct store      (source contiguous along i) : pitch DIM+4, one ds_write_b128
// ------------------------------------------------------------------------------------------------
template <int DIM> struct PitchT { static constexpr int v = DIM + 1; };
template <int DIM> struct PitchD { static constexpr int v = DIM + 4; };

template <int DIM> __device__ __forceinline__ void unit_T(int u, int& i, int& kk) { i = u >> 3; kk = (u & 7) << 2; }
template <int DIM> __device__ __forceinline__ void unit_D(int u, int& kk, int& i) { kk = u / (DIM / 4); i = (u % (DIM / 4)) << 2; }

template <int DIM> __device__ __forceinline__ void store_T(float* t, int i, int kk, float4 v) {
    constexpr int P = PitchT<DIM>::v;
    t[(kk + 0) * P + i] = v.x; t[(kk + 1) * P + i] = v.y; t[(kk + 2) * P + i] = v.z; t[(kk + 3) * P + i] = v.w;
}
template <int DIM> __device__ __forceinline__ void store_D(float* t, int kk, int i, float4 v) {
    constexpr int P = PitchD<DIM>::v;
    *reinterpret_cast<float4*>(t + kk * P + i) = v;
}

// One K-tile of MFMAs.  The operand fragments are read from LDS one CHUNK (4 k-steps) ahead of the MFMAs that
// consume them: written naively (read, wait, MFMA) hipcc emits `ds_read; s_waitcnt lgkmcnt(0); v_mfma` per step and
// a lone wavefront per SIMD (the small-M layers) spends as long waiting for LDS as issuing MFMAs.
template <int TM, int TN, int PA, int PB>
__device__ __forceinline__ void mfma_ktile(const float* __restrict__ As, const float* __restrict__ Bs, int am0,
                                           int bn0, int lane, f32x16 (&acc)[TM][TN]) {
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int CH = 4, NCH = KT / 2 / CH;
    const float* ap = As + half * PA + am0 + l31;
    const float* bp = Bs + half * PB + bn0 + l31;
    float a[2][CH][TM], b[2][CH][TN];
    auto fetch = [&](int c, int buf) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[buf][s][tm] = ap[(2 * (c * CH + s)) * PA + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[buf][s][tn] = bp[(2 * (c * CH + s)) * PB + tn * 32];
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) fetch(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);       // keep the next chunk's ds_reads ABOVE this chunk's MFMAs
#pragma unroll
        for (int s = 0; s < CH; ++s)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][s][tm], b[c & 1][s][tn], acc[tm][tn], 0, 0, 0);
    }
}

__device__ __forceinline__ int acc_row(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------------
// Segment max-pool folded into the epilogue of the pooled layer's GEMM (upstream _PointnetSAModuleBase.forward:
// F.max_pool2d over the nsample axis of relu(bn(conv)); reference call site core/networks.py:66-81).
// The pooled quantity is y = relu(scale * z + shift) with scale = gamma * istd: istd > 0, so the SIGN of scale is the sign
// of gamma -- known before the layer's statistics exist -- and y is monotone in z.  The epilogue therefore reduces the RAW
// output: max of sgn * z per (group, channel), sgn = +-1 from gamma; gad_pool_finalize applies the affine + ReLU to the
// winner once the statistics are in (bit-identical to pooling y: a correctly rounded fma is monotone).
// A group's rows are contiguous (CSR).  The output tile goes through LDS once so that a lane owns ONE column and walks the
// rows in order: every lane of the wavefront sees the same rows, so the segment structure (where a group ends) is
// wave-uniform -- scalar branches, group ids read with v_readlane -- and only the running maximum is per-lane work
// (compare + two selects per row).  A group that lies inside the walked row range is written with ONE plain 8-byte store
// per channel; only the groups cut by the range's ends (<= 2 per range) need a packed 64-bit atomic maximum --
// key = (order-preserving bits of the value) << 32 | ~row, so among equal values the smaller row wins (the arg-max trick
// of fps_kernel).  key 0 = "no row yet".  (First version: per-lane scan of the accumulator registers with an atomic per
// segment -- 3-4 M atomics per SA1 / SA2 launch cost 20-35 us, profiles/README.md round 3.)
// Arg-max rule vs the reference ("first maximal y"): identical unless two DIFFERENT raw values of a group round to the
// same y (then the larger raw value wins here) -- gad_pool_finalize handles y <= 0 (every row ties at 0: first row).
// ------------------------------------------------------------------------------------------------
struct PoolEpi {
    unsigned long long* key;       // (groups, C) packed keys, NULL: no pooling
    const int32_t* row_grp;        // (rows) group of every row
    const float* gamma;            // (C) BatchNorm weight of the pooled layer (its sign picks max / min)
    int C;
};

__device__ __forceinline__ unsigned pool_ord(float f) {
    const unsigned u = __float_as_uint(f);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}

// partial (wave-uniform): the group continues outside the rows this wavefront walks
__device__ __forceinline__ void pool_put(unsigned long long* keycol, int g, int C, float v, int row, bool partial) {
    const unsigned long long k = ((unsigned long long)pool_ord(v) << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);
    if (partial) atomicMax(keycol + (size_t)g * C, k);
    else keycol[(size_t)g * C] = k;
}

// running state of one lane's column: the open group (uniform), whether it began before the walked range (uniform), the
// best value so far and its row
struct PoolRun { int g; int part; float v; int row; };

// rows [i, e) of the LDS tile belong to the open group: running maximum of sgn * z, strict > (the first maximum stays),
// for NC columns per lane (zt + c * cstride).  i, e, row_base are wave-uniform: the loads of a run are independent of the
// compares and pipeline freely.
template <int NC>
__device__ __forceinline__ void pool_rows(const float* zt, int pitch, int cstride, int i, int e, int row_base, const float (&sgn)[NC],
                                          float (&best)[NC], int (&brow)[NC]) {
    int r = i;
    for (; r + 2 <= e; r += 2) {
        float z0[NC], z1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { z0[c] = zt[r * pitch + c * cstride] * sgn[c]; z1[c] = zt[(r + 1) * pitch + c * cstride] * sgn[c]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bool t = z0[c] > best[c]; best[c] = t ? z0[c] : best[c]; brow[c] = t ? row_base + r : brow[c];
            t = z1[c] > best[c]; best[c] = t ? z1[c] : best[c]; brow[c] = t ? row_base + r + 1 : brow[c];
        }
    }
    if (r < e) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float z0 = zt[r * pitch + c * cstride] * sgn[c];
            const bool t = z0 > best[c]; best[c] = t ? z0 : best[c]; brow[c] = t ? row_base + r : brow[c];
        }
    }
}

// One wavefront walks 32 consecutive rows (row_base ..) of its LDS tile, lane = column, carrying the open group across
// calls (streaming kernel: a contiguous range of slabs per wavefront).  gl: lane i < 32 holds the group of row i (-1 past
// the live rows).  A group that closes here is complete unless c.part says it began before the range.
__device__ __forceinline__ void pool_scan32(const float* zt, int pitch, int gl, int row_base, float sgn, PoolRun& c,
                                            unsigned long long* keycol, int C) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(gl, 1, 64);
    const bool st = lane == 0 ? gl != c.g : gl != prev;
    const unsigned m = (unsigned)__ballot(st);                  // bit i: row i opens a group (c.g is wave-uniform)
    int i = 0;
    while (i < 32) {                                            // wave-uniform control flow throughout
        if ((m >> i) & 1u) {
            if (c.g >= 0) pool_put(keycol, c.g, C, c.v, c.row, c.part != 0);
            c.g = __builtin_amdgcn_readlane(gl, i); c.part = 0; c.v = -INFINITY; c.row = row_base + i;
        }
        const unsigned rest = i < 31 ? m >> (i + 1) : 0u;
        const int e = rest ? i + 1 + __builtin_ctz(rest) : 32;
        if (c.g >= 0) {
            const float sg1[1] = {sgn};
            float bv[1] = {c.v};
            int br[1] = {c.row};
            pool_rows<1>(zt, pitch, 0, i, e, row_base, sg1, bv, br);
            c.v = bv[0]; c.row = br[0];
        }
        i = e;
    }
}

// A 64-row tile in LDS, lane = column (NC columns per lane, cstride apart), the tile's groups dealt to the workgroup's
// wavefronts (piece k -> wavefront k % nw): every piece is closed by the wavefront that walked it; only the pieces cut by
// the tile's first / last row are partial.
// gl: lane i holds the group of tile row i (-1 past the live rows); n_live = live rows of the tile (>= 1); g_before /
// g_after: the groups of the rows just outside the tile (-1: none) -- a piece that shares its group with them is partial.
template <int NC>
__device__ __forceinline__ void pool_tile64(const float* zt, int pitch, int cstride, int gl, int row0, int n_live,
                                            const float (&sgn)[NC], int wave, int nw, int g_before, int g_after,
                                            unsigned long long* keycol, int C) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(gl, 1, 64);
    const bool st = (lane == 0 || gl != prev) && lane < n_live;
    unsigned long long m = __ballot(st);
    int k = 0;
    while (m) {                                                 // wave-uniform
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        const int e = m ? __builtin_ctzll(m) : n_live;
        if ((k++ % nw) != wave) continue;
        const int g = __builtin_amdgcn_readlane(gl, i);
        float best[NC];
        int brow[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { best[c] = -INFINITY; brow[c] = row0 + i; }
        pool_rows<NC>(zt, pitch, cstride, i, e, row0, sgn, best, brow);
        const bool partial = (i == 0 && g == g_before) || (e == n_live && g == g_after);
#pragma unroll
        for (int c = 0; c < NC; ++c) pool_put(keycol + c * cstride, g, C, best[c], brow[c], partial);
    }
}

struct Groups {
    int n; int aoff[GAD_MAX_GROUPS]; int woff[GAD_MAX_GROUPS]; int ooff[GAD_MAX_GROUPS]; int nout[GAD_MAX_GROUPS];
};

// per-column partial sums held by lanes 0..31 of each wavefront -> one f64 atomic per column and block
template <int WM, int WN, int TN>
__device__ __forceinline__ void block_column_atomics(float* red, const float (&c0)[TN], const float (&c1)[TN], int lane,
                                                     int wm, int wn, int col0, int col_limit, double* out0,
                                                     double* out1) {
    constexpr int BN = WN * TN * 32;
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const float s0 = c0[tn] + __shfl_xor(c0[tn], 32, 64);
        const float s1 = c1[tn] + __shfl_xor(c1[tn], 32, 64);
        if (lane < 32) {
            const int cl = wn * TN * 32 + tn * 32 + lane;
            red[wm * BN + cl] = s0;
            red[(WM + wm) * BN + cl] = s1;
        }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < BN && col0 + tid < col_limit) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s0 += red[w * BN + tid]; s1 += red[(WM + w) * BN + tid]; }
        atomic_add_f64(out0 + col0 + tid, (double)s0);
        atomic_add_f64(out1 + col0 + tid, (double)s1);
    }
}

// ------------------------------------------------------------------------------------------------
// forward:  zout[r][n] = sum_k X[r][k] * W[n][k]      (+ weighted BatchNorm statistics)
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int XM, bool POOL>
__global__ __launch_bounds__(256) void gemm_fwd_kernel(XSrc x, Groups gr, const int32_t* __restrict__ n_rows_dev,
                                                        int n_rows_static, const float* __restrict__ row_w,
                                                        const float* __restrict__ W, int Kp,
                                                        float* __restrict__ zout, int zout_pitch,
                                                        double* __restrict__ stat_sum,
                                                        double* __restrict__ stat_sq, int stat_stride, PoolEpi pe,
                                                        unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchT<BM>::v, PB = PitchT<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    constexpr int TILE = KT * PA + KT * PB;
    constexpr int SM = ((TILE + 3) & ~3) + 2 * BM + 2 * VMAX;
    static_assert(TILE >= 2 * WM * BN, "reduction scratch must fit in the tile LDS");
    __shared__ __attribute__((aligned(16))) float smem[SM];
    float* As = smem;
    float* Bs = smem + KT * PA;
    float* wS = smem + ((TILE + 3) & ~3);
    int32_t* ptS = reinterpret_cast<int32_t*>(wS + BM);
    float* sv = wS + 2 * BM;
    float* tv = sv + VMAX;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int zoff = gr.aoff[g], n_out = gr.nout[g], ooff = gr.ooff[g];
    const float* Wg = W + gr.woff[g];
    const int n0 = blockIdx.y * BN;
    if (n0 >= n_out) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    if ((int)(blockIdx.x * BM) >= n_rows) return;          // idle block (grid sized for the worst case)
    const int nk = (Kp + KT - 1) / KT;
    const int bulk = XM == 0 ? x.c_in : x.feat_c;

    float csum[TN], csq[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { csum[t] = 0.f; csq[t] = 0.f; }

    int row0 = blockIdx.x * BM;
    XRaw ra[UA];
    float4 rb[UB];
    auto load_tile = [&](int kt) {
        const int k0 = kt * KT;
        const bool tail = k0 + KT > bulk;
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
            const int r = row0 + i;
            const bool ok = r < n_rows && (k0 + kk < Kp);
            ra[it] = x_raw<XM>(x, r, ok, zoff, k0 + kk, tail, XM == 1 ? ptS[i] : 0);
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int j, kk; unit_T<BN>(it * 256 + tid, j, kk);
            const int n = n0 + j;
            const bool ok = n < n_out && (k0 + kk < Kp);
            rb[it] = ldg4(Wg + (size_t)(ok ? n : 0) * Kp + (ok ? k0 + kk : 0));
        }
    };
    auto store_tile = [&](int kt) {
        const int k0 = kt * KT;
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
            const int r = row0 + i;
            const bool ok = r < n_rows && (k0 + kk < Kp);
            store_T<BM>(As, i, kk, x_finish<XM>(x, ra[it], ok, k0 + kk, sv, tv));
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int j, kk; unit_T<BN>(it * 256 + tid, j, kk);
            const bool ok = (n0 + j) < n_out && (k0 + kk < Kp);
            store_T<BN>(Bs, j, kk, f4sel(ok, rb[it], f4zero()));
        }
    };
    // ACT input: the first K-tile's loads need neither the row weights nor the BatchNorm vectors -- issue them first so
    // the three global-load latencies of the prologue (tile, vectors, weights) overlap instead of chaining
    bool preloaded = false;
    if (XM == 0) { load_tile(0); preloaded = true; }
    if (XM == 0 && x.affine) stage_affine<256>(sv, tv, x, zoff, x.c_in);
    for (; row0 < n_rows; row0 += gridDim.x * BM) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
        if (tid < BM) {
            const int r = row0 + tid;
            wS[tid] = r < n_rows ? (row_w ? row_w[r] : 1.f) : 0.f;
            if (XM == 1) ptS[tid] = r < n_rows ? x.row_pt[r] : 0;
        }
        // fused max-pool: the row -> group map of this tile and of its two outside neighbours, the column's sign -- issued
        // here so that their latency sits under the K loop
        int pgl = -1, pg_before = -1, pg_after = -1;
        float psgn = 1.f;
        if (POOL) {
            const int r = row0 + lane;
            pgl = r < n_rows ? pe.row_grp[r] : -1;
            pg_before = row0 > 0 ? pe.row_grp[row0 - 1] : -1;
            pg_after = row0 + BM < n_rows ? pe.row_grp[row0 + BM] : -1;
            psgn = pe.gamma[n0 + lane] < 0.f ? -1.f : 1.f;
        }
        __syncthreads();                                   // wS / ptS / sv / tv visible
        if (!preloaded) load_tile(0);
        preloaded = false;
        for (int kt = 0; kt < nk; ++kt) {
            store_tile(kt);
            __syncthreads();
            if (kt + 1 < nk) load_tile(kt + 1);
            mfma_ktile<TM, TN, PA, PB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
            __syncthreads();
        }
        const int l31 = lane & 31, half = lane >> 5;
        if (POOL) {                                          // (64 x 64 tiles, one group, n_out % 64 == 0: the launcher checks)
            static_assert(!POOL || (BM == 64 && BN == 64 && TM == 1 && TN == 1 && TILE >= 64 * 65), "pooled epilogue: 64 x 64 tiles");
            float* zt = smem;                                // the operand tiles are done with (barrier at the end of the K loop)
#pragma unroll
            for (int v = 0; v < 16; ++v) zt[(wm * 32 + acc_row(v, half)) * 65 + wn * 32 + l31] = acc[0][0][v];
            __syncthreads();
            // lane = column; the tile's groups dealt to the four wavefronts
            {
                const float sg1[1] = {psgn};
                pool_tile64<1>(zt + lane, 65, 0, pgl, row0, min(64, n_rows - row0), sg1, __builtin_amdgcn_readfirstlane(wave), 4,
                               __builtin_amdgcn_readfirstlane(pg_before), __builtin_amdgcn_readfirstlane(pg_after),
                               pe.key + n0 + lane, pe.C);
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * TN * 32 + tn * 32 + l31;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int il = wm * TM * 32 + tm * 32 + acc_row(v, half);
                    const int r = row0 + il;
                    const float zv = acc[tm][tn][v];
                    if (zout && r < n_rows && n < n_out) zout[(size_t)r * zout_pitch + ooff + n] = zv;
                    const float w = wS[il];
                    s1 = fmaf(w, zv, s1);
                    s2 = fmaf(w * zv, zv, s2);
                }
            csum[tn] += s1;
            csq[tn] += s2;
        }
        __syncthreads();   // wS / ptS reuse
    }
    if (stat_sum) {
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        block_column_atomics<WM, WN, TN>(smem, csum, csq, lane, wm, wn, n0, n_out,
                                         stat_sum + (size_t)rep * stat_stride + ooff,
                                         stat_sq + (size_t)rep * stat_stride + ooff);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- split-bf16 arithmetic (library option "mfma_split": a mask of kernel families, include/gaddpg.h GAD_SPLIT_*) ----
// An f32 value = hi + mid + lo, three bf16 terms of 8 significand bits each (the residuals are exact f32 subtractions); a product of
// two such values to 24 bits = the six term products of weight >= 2^-16, each exact in the f32 accumulator of
// v_mfma_f32_32x32x16_bf16 (16x the rate of v_mfma_f32_32x32x2_f32).  The bf16 MFMA's adder truncates toward -inf (a -1e-8
// relative bias, tools/ubench/split_bf16.hip): the products of every second block of 16 along the reduction index are
// NEGATED (one operand's sign) and summed into a second accumulator; result = plain - negated, which cancels the bias and
// leaves a smaller random error than the f32 MFMA's (profiles/r04_split_bf16_ubench.txt).  The cancellation is exact only
// where the two accumulators sit in the same binade; what is left (~2e-11 of max |z| per element) has a sign that depends on
// the output COLUMN (which half of a column's weights is larger), so a sum over the rows of one column -- BatchNorm statistics,
// dbeta -- would add it coherently.  The kernels whose outputs are summed over rows therefore also negate the A operand of
// every ODD ROW and take (negated - plain) there: the residual changes sign from row to row and averages out
// (profiles/r05_split_families.txt).
// Range: |x| must stay below 3.39e38 (bf16(x) must not round to infinity); an infinite operand gives NaN where the f32
// path gives +-inf (both non-finite).
typedef __bf16 gad_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned gad_cvt_pk_bf16(float lo, float hi) {     // {bf16(hi), bf16(lo)}, round to nearest even
    unsigned r;
    __asm__("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void gad_split2(float a, float b, unsigned& H, unsigned& M, unsigned& L) {
    H = gad_cvt_pk_bf16(a, b);
    const f32x2 r = f32x2{a, b} - f32x2{__uint_as_float(H << 16), __uint_as_float(H & 0xffff0000u)};
    M = gad_cvt_pk_bf16(r.x, r.y);
    const f32x2 q = r - f32x2{__uint_as_float(M << 16), __uint_as_float(M & 0xffff0000u)};
    L = gad_cvt_pk_bf16(q.x, q.y);
}
__device__ __forceinline__ gad_bf16x8 gad_as_bf16x8(gad_u32x4 u) { return *reinterpret_cast<gad_bf16x8*>(&u); }

// family mask (GAD_SPLIT_*; 1 = all): which GEMM families multiply as split-bf16 MFMAs.  LIBRARY default 0 = v_mfma_f32_32x32x2_f32
// throughout (round 6, ADVICE r05: a caller of the C ABI gets the reference's arithmetic unless it asks for something else); the
// Python package opts in explicitly when it loads the library (ga-ddpg_amd/hip.py OPTION_DEFAULTS: GAD_SPLIT_ALL -- the per-family
// float64 gates of tests/test_gpu_split_families.py, the range / specials tests and every oracle-facing gate in both modes are green
// on MI355X), GAD_OPT_mfma_split=0 keeps the f32 MFMA.  Precondition of the split form: finite operands below 3.39e38 in magnitude
// (bf16(x) must not round to infinity; an infinite operand gives NaN where the f32 path gives +-inf).
static int g_opt_mfma_split = 0;
static bool split_on(int family) { return g_opt_mfma_split == GAD_SPLIT_ALL || (g_opt_mfma_split & family) != 0; }

// Wide-tile kernels in split form (gemm_fwd_wide / gemm_dx_wide with SP): a 64 x 128 block tile per 4-wavefront workgroup,
// K-tile 32.  Both operand tiles live in LDS as three bf16 planes of 64-byte rows (32 reduction indices), double-buffered:
//   stage = A planes 3 x 64 rows | B planes 3 x 128 rows = 36 KB;
// a row's four 16-byte chunks are XOR-swizzled by (row >> 2) & 3, so the fragment reads -- lane (row l31, half) takes chunk
// 2 s + half of k16-step s: one ds_read_b128 = the 8 consecutive reduction indices v_mfma_f32_32x32x16_bf16 wants per lane --
// are conflict-free without padding (every 16-lane group of the instruction covers 16 rows that differ mod 16), and so are
// the staging stores (ds_write_b64 for the A side, split in registers; ds_write_b128 for the B side, copied from the
// pre-split weight mirror).  k16-step 0 of a K-tile accumulates into `acc`, step 1 -- whose mirror values are stored negated --
// into `an`.
namespace spw {
constexpr int APL = 64 * 64, BPL = 128 * 64, STAGE = 3 * (APL + BPL);       // bytes
struct BRegs { gad_u32x4 r[2][3]; };
// this thread's B chunks of K-tile kt: rows (tid >> 2) + 64 u of the block's 128 mirror rows, chunk tid & 3
__device__ __forceinline__ void load_b(BRegs& b, __amdgpu_buffer_rsrc_t rs, const int (&vb)[2], int plane_bytes, int kt) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) b.r[u][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vb[u], p * plane_bytes + kt * 64, 0);
}
__device__ __forceinline__ void store_b(unsigned char* stage, const BRegs& b, int tid) {
    const int row = tid >> 2;
    unsigned char* q = stage + 3 * APL + row * 64 + ((((tid & 3) ^ ((row >> 2) & 3))) << 4);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<gad_u32x4*>(q + p * BPL + u * (64 * 64)) = b.r[u][p];
}
// four consecutive reduction indices (c4 = (tid & 7) * 4 of the K-tile) of A row `row` -> the three planes (8 bytes each)
// sg: 0x80000000 for an odd row (its values enter negated, the epilogue takes negated - plain), else 0
__device__ __forceinline__ void store_a4(unsigned char* stage, int row, int tid, float4 v, unsigned sg) {
    unsigned h0, m0, l0, h1, m1, l1;
    gad_split2(__uint_as_float(__float_as_uint(v.x) ^ sg), __uint_as_float(__float_as_uint(v.y) ^ sg), h0, m0, l0);
    gad_split2(__uint_as_float(__float_as_uint(v.z) ^ sg), __uint_as_float(__float_as_uint(v.w) ^ sg), h1, m1, l1);
    unsigned char* q = stage + row * 64 + (((((tid & 7) >> 1) ^ ((row >> 2) & 3))) << 4) + ((tid & 1) << 3);
    *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(q + APL) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(q + 2 * APL) = make_uint2(l0, l1);
}
// MFMAs of one staged K-tile for the wavefront's 32 x 64 tile: arow / brow = byte offsets of the lane's A row and first B row
// inside the stage, fo = ((half ^ swizzle) << 4) -- the lane's chunk of k16-step 0; step 1's is fo ^ 32
__device__ __forceinline__ void ktile(const unsigned char* stage, int arow, int brow, int fo, f32x16 (&acc)[2], f32x16 (&an)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int off = fo ^ (s << 5);
        const unsigned char* ap = stage + arow + off;
        const unsigned char* bp = stage + 3 * APL + brow + off;
        const gad_u32x4 AH = *reinterpret_cast<const gad_u32x4*>(ap), AM = *reinterpret_cast<const gad_u32x4*>(ap + APL);
        const gad_u32x4 AL = *reinterpret_cast<const gad_u32x4*>(ap + 2 * APL);
        gad_u32x4 BH[2], BM[2], BL[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            BH[t] = *reinterpret_cast<const gad_u32x4*>(bp + t * (32 * 64));
            BM[t] = *reinterpret_cast<const gad_u32x4*>(bp + BPL + t * (32 * 64));
            BL[t] = *reinterpret_cast<const gad_u32x4*>(bp + 2 * BPL + t * (32 * 64));
        }
        // the six products of weight >= 2^-16, smallest first
#define GAD_SPW(A, B)                                                                                                          \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                        \
            if (s) an[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[t]), an[t], 0, 0, 0);     \
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[t]), acc[t], 0, 0, 0);     \
        }
        GAD_SPW(AL, BH) GAD_SPW(AH, BL) GAD_SPW(AM, BM) GAD_SPW(AM, BH) GAD_SPW(AH, BM) GAD_SPW(AH, BH)
#undef GAD_SPW
    }
}
}  // namespace spw

// ------------------------------------------------------------------------------------------------
// wide-tile forward for the MID-SIZE layers (SA2 / SA3 layers 2 and 3: 8e3 - 3e4 rows, K 128 - 256, 128 - 512 outputs).
// The 64 x 64 kernel above gives every wavefront ONE 32 x 32 accumulator: 16 MFMAs (0.45 us) per K-tile against ~1 us of
// global-load latency, a transposing LDS store of four ds_write_b32 per staged float4 and ~10 vector instructions per MFMA.
// Here a workgroup owns 64 rows x 128 columns, a wavefront 32 x 64 (two accumulators sharing the A fragment):
//   * 32 MFMAs (0.85 us) per K-tile and wavefront cover the load latency of the next tile (register-staged, issued right
//     after the previous LDS write), LDS is double-buffered: ONE barrier per K-tile;
//   * operands sit row-major in LDS, [row][32 + 4] floats: the staged float4 goes out as one ds_write_b128 (eight lanes
//     fill one row: conflict-free), fragments come in as one ds_read_b128 per four MFMA steps (k visited as 8j + 4h + i, the
//     streaming kernels' order: the 36-float pitch spreads 16 rows over all 64 banks);
//   * relu(scale * z + shift) is applied once per staged element, not per fragment read.
// XM = 1: the first layer of SA2 / SA3, whose input rows are GATHERED ([feat[pt] | src_xyz[pt] - ctr_xyz[grp]], features a
// multiple of 32 wide, already activated): the K loop runs over the feature columns only, the three coordinate columns
// are a rank-3 update of the accumulators in the epilogue (z += dx . W[n][feat_c .. feat_c + 2]) instead of a fifth,
// almost empty K-tile.
// Measured alone (tools/diag_gemm.py, B = 256 shapes): see profiles/README.md round 3.
// ------------------------------------------------------------------------------------------------
// SP: the products as split-bf16 MFMAs (namespace spw above): A is split while it is staged, B comes from the layer's forward
// weight mirror (wsp: three planes of wsp_plane bf16, rows of wsp_pitch), same tiles, prologue and epilogue.
template <int XM, bool POOL, bool SP = false>
__global__ __launch_bounds__(256, 2) void gemm_fwd_wide_kernel(XSrc x, const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                               const float* __restrict__ row_w, const float* __restrict__ W,
                                                               int Kp, int n_out, float* __restrict__ zout, int zout_pitch,
                                                               double* __restrict__ stat_sum, double* __restrict__ stat_sq,
                                                               int stat_stride, PoolEpi pe, const uint16_t* __restrict__ wsp,
                                                               int wsp_pitch, int wsp_plane, unsigned long long* __restrict__ ts) {
    KTimer kt_(ts);
    GAD_WPH_DECL;
    GAD_WKL_DECL;
    constexpr int BM = 64, BN = 128, P = KT + 4, STAGE = SP ? spw::STAGE / 4 : (BM + BN) * P, VM = 512;
    static_assert(2 * STAGE >= 64 * 129, "the pooled epilogue's tile lives in the operand buffers");
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 2 * VM + BM + 4 * BM];
    float* sv = smem + 2 * STAGE;
    float* tv = sv + VM;
    float* wS = tv + VM;
    float* dxS = wS + BM;                                // XM = 1: [row][4] = src_xyz[pt] - ctr_xyz[grp]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    if ((int)(blockIdx.x * BM) >= n_rows) return;
    const int nk = (XM == 0 ? Kp : x.feat_c) / KT;
    const int c4 = (tid & 7) * 4;                        // this thread's 16-byte column chunk inside a K-tile
    const int ur = tid >> 3;                             // ... and row (A: rows ur, ur + 32; B: ur, +32, +64, +96)
    // operands and output through buffer descriptors: a 32-bit byte offset per lane (set up once per row tile) + a scalar
    // offset per access instead of 64-bit pointer arithmetic per load / store; the output descriptor is bounded by the live
    // rows (a NULL zout: zero records), so stores of rows past them -- or of a pass that keeps no raw output -- are dropped
    const __amdgpu_buffer_rsrc_t ar_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(XM == 0 ? x.zin : x.feat), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, GAD_BUF_MAX, 0x00020000);
    const int zo_pitch4 = zout_pitch * 4;
    const __amdgpu_buffer_rsrc_t zo_ = __builtin_amdgcn_make_buffer_rsrc(zout ? zout : const_cast<float*>(W), 0, zout ? gad_nbytes(n_rows, zo_pitch4) : 0, 0x00020000);
    int vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vb[u] = ((n0 + ur + 32 * u) * Kp + c4) * 4;
    // SP: the weight mirror's rows n0 + (tid >> 2) + 64 u, 16-byte chunk tid & 3 of a K-tile's 64 bytes
    const __amdgpu_buffer_rsrc_t ws_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SP ? wsp : reinterpret_cast<const uint16_t*>(W)), 0, GAD_BUF_MAX, 0x00020000);
    int vbs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) vbs[u] = ((n0 + (tid >> 2) + 64 * u) * wsp_pitch + (tid & 3) * 8) * 2;
    unsigned char* const smem_b = reinterpret_cast<unsigned char*>(smem);
    const int sp_fo = ((half ^ ((l31 >> 2) & 3)) << 4);
    if (XM == 0) {
        if (x.bn.stat_sum) {
            // the input layer's train-mode BatchNorm finalised here (no gad_bn_finalize launch between the two GEMMs): every
            // workgroup forms scale / shift of the K input channels from the f64 statistics -- 16 loads per channel, in flight
            // beside the first K-tile's operand loads below -- and the first workgroup publishes them for the backward pass
            const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
            for (int i = tid; i < Kp; i += 256) {
                float sc, sh;
                gad_bn_fin_channel(x.bn, i, writer, sc, sh);
                sv[i] = sc; tv[i] = sh;
            }
        } else {
            stage_affine<256>(sv, tv, x, 0, Kp);
        }
    }

    float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
    for (int row0 = blockIdx.x * BM; row0 < n_rows; row0 += gridDim.x * BM) {
        f32x16 acc[2];
        f32x16 an[2];                                    // SP: the negated accumulators (odd k16-steps)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[t][v] = 0.f; if (SP) an[t][v] = 0.f; }
        int pgl = -1, pg_before = -1, pg_after = -1;
        float psgn[2] = {1.f, 1.f};
        if (POOL) {
            const int r = row0 + lane;
            pgl = r < n_rows ? pe.row_grp[r] : -1;
            pg_before = row0 > 0 ? pe.row_grp[row0 - 1] : -1;
            pg_after = row0 + BM < n_rows ? pe.row_grp[row0 + BM] : -1;
            psgn[0] = pe.gamma[n0 + lane] < 0.f ? -1.f : 1.f;
            psgn[1] = pe.gamma[n0 + 64 + lane] < 0.f ? -1.f : 1.f;
        }
        // rows past the live count are clamped to the last live row (never stored, weight 0 in the statistics)
        const int ra0 = min(row0 + ur, max(n_rows - 1, 0)), ra1 = min(row0 + ur + 32, max(n_rows - 1, 0));
        const int va0 = ((XM == 0 ? ra0 * x.zin_pitch : x.row_pt[ra0] * x.feat_c) + c4) * 4;
        const int va1 = ((XM == 0 ? ra1 * x.zin_pitch : x.row_pt[ra1] * x.feat_c) + c4) * 4;
        // two register sets: the global loads of a K-tile are issued TWO tiles before its LDS write (one tile of MFMAs is
        // ~0.85 us, less than the memory latency under load: with one set every K-tile's barrier waited for its loads)
        float4 ra2[2][2], rb2[2][SP ? 1 : 4];
        spw::BRegs rbs[SP ? 2 : 1];
        auto load_regs = [&](int kt, auto setc) {
            constexpr int S = decltype(setc)::value;
            const int k0 = kt * (KT * 4);
            ra2[S][0] = buf_ld4(ar_, va0, k0); ra2[S][1] = buf_ld4(ar_, va1, k0);
            if (SP) { spw::load_b(rbs[SP ? S : 0], ws_, vbs, wsp_plane * 2, kt); return; }
#pragma unroll
            for (int u = 0; u < (SP ? 1 : 4); ++u) rb2[S][u] = buf_ld4(wr_, vb[u], k0);
        };
        auto write_lds = [&](int kt, auto setc) {
            constexpr int S = decltype(setc)::value;
            const float4* ra = ra2[S];
            const float4* rb = rb2[S];
            float* As = smem + (kt & 1) * STAGE;
            float* Bs = As + BM * P;
            float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), t4 = f4zero();
            if (XM == 0) {
                s4 = *reinterpret_cast<const float4*>(sv + kt * KT + c4);
                t4 = *reinterpret_cast<const float4*>(tv + kt * KT + c4);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 v = ra[u];
                if (XM == 0) {
                    v.x = fmaxf(fmaf(v.x, s4.x, t4.x), 0.f); v.y = fmaxf(fmaf(v.y, s4.y, t4.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, s4.z, t4.z), 0.f); v.w = fmaxf(fmaf(v.w, s4.w, t4.w), 0.f);
                }
                if (SP) spw::store_a4(smem_b + (kt & 1) * spw::STAGE, ur + 32 * u, tid, v, (ur & 1) ? 0x80000000u : 0u);
                else *reinterpret_cast<float4*>(As + (ur + 32 * u) * P + c4) = v;
            }
            if (SP) { spw::store_b(smem_b + (kt & 1) * spw::STAGE, rbs[SP ? S : 0], tid); return; }
#pragma unroll
            for (int u = 0; u < (SP ? 1 : 4); ++u) *reinterpret_cast<float4*>(Bs + (ur + 32 * u) * P + c4) = rb[u];
        };
        const std::integral_constant<int, 0> S0;
        const std::integral_constant<int, 1> S1;
        load_regs(0, S0);
        if (nk > 1) load_regs(1, S1);
        __syncthreads();                                 // sv / tv visible; the previous row tile's LDS reads (wS, zt) are done
        if (tid < BM) {
            const int r = row0 + tid;
            wS[tid] = r < n_rows ? (row_w ? row_w[r] : 1.f) : 0.f;
            if (XM == 1) {                               // the row's recentred coordinates (rounded as the oracle rounds them)
                const int rr = min(r, max(n_rows - 1, 0));
                const float* p = x.src_xyz + (size_t)x.row_pt[rr] * 3;
                float q0 = p[0], q1 = p[1], q2 = p[2];
                if (x.ctr_xyz) {
                    const float* cp = x.ctr_xyz + (size_t)x.row_grp[rr] * 3;
                    q0 = __fsub_rn(q0, cp[0]); q1 = __fsub_rn(q1, cp[1]); q2 = __fsub_rn(q2, cp[2]);
                }
                *reinterpret_cast<float4*>(dxS + 4 * tid) = make_float4(q0, q1, q2, 0.f);
            }
        }
        GAD_WPH(1);
        write_lds(0, S0);
        if (nk > 2) load_regs(2, S0);
        __syncthreads();
        GAD_WPH(2);
        auto ktile = [&](int kt, auto nxtc) {             // MFMAs of tile kt; tile kt + 1 (register set nxtc) -> LDS; loads of kt + 3
            if (SP) {
                GAD_WKL(0);
                spw::ktile(smem_b + (kt & 1) * spw::STAGE, (wm * 32 + l31) * 64, (wn * 64 + l31) * 64, sp_fo, acc, an);
                GAD_WKL(1);
                if (kt + 1 < nk) write_lds(kt + 1, nxtc);
                if (kt + 3 < nk) load_regs(kt + 3, nxtc);
                GAD_WKL(2);
                __syncthreads();
                GAD_WKL(3);
                return;
            }
            const float* As = smem + (kt & 1) * STAGE + (wm * 32 + l31) * P + 4 * half;
            const float* Bs = smem + (kt & 1) * STAGE + BM * P + (wn * 64 + l31) * P + 4 * half;
            float4 a4 = *reinterpret_cast<const float4*>(As);
            float4 b0 = *reinterpret_cast<const float4*>(Bs), b1 = *reinterpret_cast<const float4*>(Bs + 32 * P);
#pragma unroll
            for (int j = 0; j < KT / 8; ++j) {
                float4 an = a4, bn0 = b0, bn1 = b1;
                if (j + 1 < KT / 8) {                    // next k group's fragments land under this group's MFMAs
                    an = *reinterpret_cast<const float4*>(As + 8 * (j + 1));
                    bn0 = *reinterpret_cast<const float4*>(Bs + 8 * (j + 1));
                    bn1 = *reinterpret_cast<const float4*>(Bs + 32 * P + 8 * (j + 1));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
                a4 = an; b0 = bn0; b1 = bn1;
            }
            if (kt + 1 < nk) write_lds(kt + 1, nxtc);    // the other LDS buffer: its readers passed the previous barrier
            if (kt + 3 < nk) load_regs(kt + 3, nxtc);
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {             // (nk is even: K is a multiple of 64 here)
            ktile(kt, S1);
            if (kt + 1 < nk) ktile(kt + 1, S0);
        }
        GAD_WPH(3);
        if (SP) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = (v & 1) ? an[t][v] - acc[t][v] : acc[t][v] - an[t][v];   // (odd rows entered negated)
        }
        if (XM == 1) {                                   // the three coordinate columns: z += dx . W[n][feat_c .. feat_c + 2]
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float* wx = W + (size_t)(n0 + wn * 64 + t * 32 + l31) * Kp + x.feat_c;
                const float w0 = wx[0], w1 = wx[1], w2 = wx[2];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float4 q = *reinterpret_cast<const float4*>(dxS + 4 * (wm * 32 + acc_row(v, half)));
                    acc[t][v] = fmaf(q.x, w0, fmaf(q.y, w1, fmaf(q.z, w2, acc[t][v])));
                }
            }
        }
        if (POOL) {
            float* zt = smem;                            // [64][129]: every wavefront is past the K loop's last barrier
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) zt[(wm * 32 + acc_row(v, half)) * 129 + wn * 64 + t * 32 + l31] = acc[t][v];
            __syncthreads();
            const int gb = __builtin_amdgcn_readfirstlane(pg_before), ga = __builtin_amdgcn_readfirstlane(pg_after);
            pool_tile64<2>(zt + lane, 129, 64, pgl, row0, min(64, n_rows - row0), psgn, wave, 4, gb, ga, pe.key + n0 + lane, pe.C);
        }
        if (SP) {
            // stores first, then the statistics from ONE batch of weight reads.  (Interleaved as in the FP32 path below, hipcc
            // packs the two column tiles into v_pk_* pairs and refills the weight registers piecemeal -- ds_read_b96 / b32 / b64
            // into registers whose stores are still in flight; with two workgroups per CU that form produced wrong sums of
            // squares for the first column tile in 29 of 40 launches on an MI355X, this one in 0 of 40: tools/diag_split_sq.py.)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = n0 + wn * 64 + t * 32 + l31;
            const int vzo = (4 * half * zout_pitch + n) * 4;
#pragma unroll
            for (int v = 0; v < 16; ++v)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][v]), zo_, vzo, (row0 + wm * 32 + (v & 3) + 8 * (v >> 2)) * zo_pitch4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float w = wS[wm * 32 + acc_row(v, half)];
                const float zv = acc[t][v];
                s1 = fmaf(w, zv, s1);
                s2 = fmaf(w * zv, zv, s2);
            }
            csum[t] += s1;
            csq[t] += s2;
        }
        } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = n0 + wn * 64 + t * 32 + l31;
            const int vzo = (4 * half * zout_pitch + n) * 4;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int il = wm * 32 + acc_row(v, half);
                const float zv = acc[t][v];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zv), zo_, vzo, (row0 + wm * 32 + (v & 3) + 8 * (v >> 2)) * zo_pitch4, 0);
                const float w = wS[il];
                s1 = fmaf(w, zv, s1);
                s2 = fmaf(w * zv, zv, s2);
            }
            csum[t] += s1;
            csq[t] += s2;
        }
        }
        // (the next row tile's first barrier orders these wS / zt reads before its writes)
        GAD_WPH(4);
    }
    if (stat_sum) {
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        block_column_atomics<2, 2, 2>(smem, csum, csq, lane, wm, wn, n0, n_out, stat_sum + (size_t)rep * stat_stride,
                                      stat_sq + (size_t)rep * stat_stride);
    }
#ifdef GAD_W_PHASES
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GAD_WPH(5);
    GAD_WPH_STORE(ts);
    GAD_WKL_STORE(ts);
#endif
}

static int g_opt_fwd_wide = 1;
static int g_opt_dx_wide = 1, g_opt_dw_wide = 1;
static int g_opt_bwd_fused = 1;
static int g_opt_fwd_bn_prologue = 1;           // 0: routes with a BatchNorm-finalising prologue launch gad_bn_finalize instead (A/B)
static int g_opt_bwd_wide = 0;                  // fused wide backward: 0 off (default: slower in the step), 1 SA2 and SA3 shapes, 2 only layers with >= 16384 rows (SA2)
static int g_opt_bwd_wide_slab = 4;             // most partial-dW elements (millions) a fused launch may write: bounds its workgroups per k block
static int g_opt_dw_wide_wgs = 256;        // workgroups a wide-tile dW launch aims for (its partial slab = this x 128 x 128 floats)
// the wide-tile kernel covers: ACT input, one group, K = the channel count itself (a multiple of 32, no bias / extra
// column), outputs a multiple of 128
static bool fits_i32_bytes(long long rows, int pitch_a, int pitch_b, int pitch_c, int pitch_d);
static bool fwd_wideable(const gad_gemm_fwd_args& a) {
    if (!g_opt_fwd_wide || a.n_groups != 1 || a.zin_off[0] != 0 || a.w_off[0] != 0 || a.out_off[0] != 0) return false;
    // (gathered input: the point-major feature tensor has at most as many rows as there are (group, point) rows upstream)
    if (!fits_i32_bytes(a.n_rows, a.mode == 1 ? a.feat_c : a.zin_pitch, a.zout ? a.zout_pitch : 0, 0, 0)) return false;
    if (a.n_rows < 2048 || a.n_out[0] % 128 != 0 || a.ones_col >= 0) return false;
    if (a.mode == 2) return false;
    if (a.mode == 1)                                     // gathered first layer: features a multiple of 32, + 3 coordinates
        return g_opt_fwd_wide != 2 && !a.pool_key && a.act_c == 0 && a.feat_c % 32 == 0 && a.feat_c >= 32 && a.feat_c <= 512 &&
               a.Kp == ((a.feat_c + 3 + 7) & ~7);
    if (a.Kp % 32 != 0 || a.Kp > 512 || a.Kp != a.c_in) return false;
    return !a.extra && a.scale && a.shift && a.relu;
}
// the wide-tile kernels address their row tensors through buffer descriptors with 32-bit byte offsets: a launch whose largest
// row tensor reaches 2 GiB takes the 64 x 64 tile kernels (64-bit pointer arithmetic) instead
static bool fits_i32_bytes(long long rows, int pitch_a, int pitch_b, int pitch_c, int pitch_d) {
    int p = pitch_a > pitch_b ? pitch_a : pitch_b;
    p = p > pitch_c ? p : pitch_c;
    p = p > pitch_d ? p : pitch_d;
    return rows * (long long)p * 4 <= (1ll << 31);
}

// ------------------------------------------------------------------------------------------------
// streaming forward for the wide-and-shallow SA1 layers (rows ~ 2e5, Kp <= 64, n_out <= 128):
// HBM-bound (SA1 layer 3: 192 B/row moved vs 16 kFLOP/row), so the kernel is built around bytes in
// flight rather than around a block tile:
//   * W (<= 34 KB) is staged ONCE per workgroup in LDS, row-major [n][Kp+4] (ds_read_b128, conflict-free);
//   * every wavefront owns 32-row slabs: lane (r = lane%32, h = lane/32) loads X[r][8j+4h..+3] as 16-byte
//     loads straight into the MFMA A-operand registers (no LDS round trip, no workgroup barrier in the
//     loop) -- the reduction index is visited in the order k = 8j+4h+i, identically for A and B;
//   * the next slab's loads are issued before the current slab's MFMAs (register double buffer) and the
//     grid is persistent at 2 wavefronts per SIMD, so loads, MFMAs and the epilogue stores of different
//     wavefronts overlap.
// ------------------------------------------------------------------------------------------------
// NOTE (measured, tools/ubench/mfma_valu.hip): on gfx950 the f32 MFMA shares the vector ALU -- every VALU
// instruction issued between MFMAs adds its ~4 clocks to the 64 of the MFMA, also with 2 wavefronts per SIMD.
// So the loop below is written for a minimal VALU instruction count: packed (2-wide) f32 math for the BatchNorm
// affine and the statistics, buffer stores whose row offset lives in an SGPR (no per-store address arithmetic),
// clamped row indices instead of per-element selects.
// XM = 2 / 3 (round 4): the 64-channel ACT input is not read but RECOMPUTED per slab from the gathered rows of
// the stage's first layer (packed K = 8 / 16: XM - 1 k groups) -- z1^T = W1 . X^T with the operands of the first layer's own
// MFMAs swapped, so the accumulator registers of lane (row, half) ARE this kernel's A fragments (channels 8j + 4 half + i) and
// every product and sum is the one the first layer's launch made (bit-equal z1): 8 / 16 extra MFMAs per slab instead of
// 8 KB of z1 from HBM.  The first layer's launch still provides the BatchNorm statistics (and z1 for a pass that is
// back-propagated).
template <int KJ, int TN, int XM, bool POOL, bool SP = false>
__global__ __launch_bounds__(512, 2) void gemm_fwd_stream_kernel(XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                                  int n_rows_static, const float* __restrict__ row_w,
                                                                  const float* __restrict__ W, float* __restrict__ zout,
                                                                  double* __restrict__ stat_sum,
                                                                  double* __restrict__ stat_sq, int stat_stride, PoolEpi pe,
                                                                  unsigned long long* __restrict__ ts) {
#ifdef GAD_X_PHASES
    const long long kt_start = __builtin_readcyclecounter();
#else
    KTimer kt(ts);
#endif
    constexpr int KP = 8 * KJ, NO = 32 * TN, PW = KP + 4;
    constexpr bool RE = XM >= 2;                         // recomputed ACT input
    constexpr int PK = RE ? XM - 1 : 0;                  // k groups of the recomputed first layer
    constexpr bool ACT = XM == 0 || RE, GATHER = XM == 1 || RE;
    constexpr int NG = RE ? PK : KJ;                     // k groups a slab's loads cover
    static_assert(!RE || (KJ == 8 && !POOL), "recomputed input: 64 channels");
    // SP (opt-in split-bf16 arithmetic, ACT input with 64 channels): W lives in LDS as three bf16 planes [plane][n][64 (+ 8 pad)] and a
    // lane's slab columns are 16 s + 8 half + 0 .. 7 (the A layout of v_mfma_f32_32x32x16_bf16) instead of 8 j + 4 half + 0 .. 3
    static_assert(!SP || (XM == 0 && KJ == 8), "split-bf16: ACT input, 64 channels");
    constexpr int PB = 2 * KP + 16;                      // bytes per row of a bf16 plane
    __shared__ __attribute__((aligned(16))) float Ws[SP ? 3 * NO * PB / 4 : NO * PW];
    __shared__ __attribute__((aligned(16))) float sv[KP], tv[KP];
    __shared__ float red[2 * 8 * NO];
    __shared__ float pz[POOL ? 8 * 32 * 64 : 1];        // fused max-pool: a 32-row x 64-column tile per wavefront
    static_assert(!POOL || TN == 4, "pooled epilogue: 128 output columns");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int n_slabs = (n_rows + 31) >> 5;
#if GAD_X_PHASES == 2
    long long st1, st2, st3, st4;
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GAD_PH_STAMP(st1);                                   // the live row count has arrived
#endif
    // 8 wavefronts per workgroup, one workgroup per CU: wavefronts w and w+4 share a SIMD (cyclic placement).  Slabs
    // are dealt wave-major (w * gridDim + block), so the left-over slabs of the last round go to w = 0..3 first and
    // every SIMD ends up with the same slab count +-1 (dealing block-major left whole CUs a round short: -20 %).
    int stride = gridDim.x * 8;
    int slab = wave * gridDim.x + blockIdx.x;            // wave-uniform (SGPR)
    int slab_end = n_slabs;
    if (POOL) {
        // fused max-pool: every wavefront walks a CONTIGUOUS range of slabs, so a group's running maximum is carried in
        // registers from slab to slab and only the two groups cut by the range's ends need atomics (same slab count per
        // wavefront +-1 as the dealing above)
        const int nw = gridDim.x * 8, base = n_slabs / nw, rem = n_slabs - base * nw;
        slab_end = slab * base + min(slab, rem) + base + (slab < rem ? 1 : 0);
        slab = slab * base + min(slab, rem);
        stride = 1;
    }
    // output through a buffer descriptor: rows >= n_rows fall outside num_records and are dropped by the hardware
    const bool store_z = zout != nullptr;                // (a pass that is never back-propagated keeps only the pooled maxima)
    const __amdgpu_buffer_rsrc_t zrsrc =
        __builtin_amdgcn_make_buffer_rsrc(store_z ? zout : const_cast<float*>(W), 0, store_z ? gad_nbytes(n_rows, NO * 4) : 0, 0x00020000);
    const int zlane = (4 * half * NO + l31) * 4;         // byte offset of (row 4*half, column l31)
    // fused max-pool: in the scan a lane owns columns lane and 64 + lane; +-1 = the sign of their BatchNorm weight
    float psgn[2];
    PoolRun prun[2];
    const int range0 = slab * 32;
    int g_after = -1;                                    // group of the first row after the range (-1: none)
    if (POOL && slab < slab_end) {
        const int g0 = pe.row_grp[range0];               // (range0 < n_rows: the range has rows)
        const int part0 = range0 > 0 && pe.row_grp[range0 - 1] == g0;
        if (slab_end * 32 < n_rows) g_after = pe.row_grp[slab_end * 32];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            psgn[p] = pe.gamma[p * 64 + lane] < 0.f ? -1.f : 1.f;
            prun[p].g = g0; prun[p].part = part0; prun[p].v = -INFINITY; prun[p].row = range0;
        }
    }

    f32x2 csum[TN], csq[TN];                             // .x/.y: even / odd accumulator rows, summed at the end
#pragma unroll
    for (int t = 0; t < TN; ++t) { csum[t] = f32x2{0.f, 0.f}; csq[t] = f32x2{0.f, 0.f}; }

    XRaw ra[NG], rn[NG];
    float4 w1f[RE ? 2 : 1][RE ? PK : 1];                 // recomputed input: W1[32 b + l31][8 j + 4 half .. + 3], loop-invariant
    if (RE) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < PK; ++j) w1f[b][j] = ldg4(x.pre_W + (size_t)(32 * b + l31) * x.pre_Kp + 8 * j + 4 * half);
    }
    int pt_nxt = 0;
    auto load_pt = [&](int sl) {          // point index of the lane's row in slab sl (gather input only)
        const int r = min(sl * 32 + l31, max(n_rows - 1, 0));
        return (GATHER && sl < n_slabs) ? x.row_pt[r] : 0;
    };
    auto load_slab = [&](int sl, int pt, XRaw (&dst)[NG]) {
        const int r = min(sl < n_slabs ? sl * 32 + l31 : 0, max(n_rows - 1, 0));      // clamped: ragged rows repeat the last row
        if (GATHER) {
            // SA1's gathered rows [f (4) | x_j - c_i (3) | action (0 / 6)]: everything a row needs in FIVE loads (features,
            // point, centre as 12-byte loads, action as two), then each (k group, half) picks its four columns.  x_raw per
            // k group issued 11 scalar gathers per lane, most of them for columns the lane does not feed: the layer was
            // bound by the texture path (64 cache lines per gather instruction), not by its 54 MB of output.
            struct F3 { float x, y, z; } __attribute__((packed, aligned(4)));
            const float4 f = ldg4(x.feat + (size_t)pt * 4);
            const F3 p = *reinterpret_cast<const F3*>(x.src_xyz + (size_t)pt * 3);
            const int grp = x.row_grp[r];
            float q0 = p.x, q1 = p.y, q2 = p.z;
            if (x.ctr_xyz) {
                const F3 c = *reinterpret_cast<const F3*>(x.ctr_xyz + (size_t)grp * 3);
                q0 = __fsub_rn(q0, c.x); q1 = __fsub_rn(q1, c.y); q2 = __fsub_rn(q2, c.z);
            }
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f;
            if (x.action) {                                                  // (fwd_streamable: act_c == 6)
                const float* ap = x.action + (size_t)(grp / x.gps) * 6;
                const F3 u = *reinterpret_cast<const F3*>(ap), v = *reinterpret_cast<const F3*>(ap + 3);
                a0 = u.x; a1 = u.y; a2 = u.z; a3 = v.x; a4 = v.y; a5 = v.z;
            }
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                dst[j].a = f;
                if (j == 0) dst[j].s = make_float4(q0, q1, q2, a0);          // columns 4 .. 7 (half 1; half 0 takes the features)
                else dst[j].s = half ? make_float4(a5, 0.f, 0.f, 0.f) : make_float4(a1, a2, a3, a4);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j)
            dst[j] = x_raw<GATHER ? 1 : 0>(x, r, true, 0, SP ? 16 * (j >> 1) + 8 * half + 4 * (j & 1) : 8 * j + 4 * half, GATHER, pt);
    };
    auto lds_b = [&](int j, float4 (&b4)[TN]) {
#pragma unroll
        for (int t = 0; t < TN; ++t) b4[t] = *reinterpret_cast<const float4*>(Ws + (t * 32 + l31) * PW + 8 * j + 4 * half);
    };
    {   // the first slab's loads are in flight while W and the input layer's affine are staged
        const int pt0 = load_pt(slab);
        load_slab(slab, pt0, ra);
        pt_nxt = load_pt(slab + stride);
    }
    {   // stage W: all 16-byte loads in flight before the first LDS store; the input layer's per-channel affine (published vectors,
        // or its BatchNorm finalised here from the statistic replicas -- see gemm_fwd_wide_kernel) is fetched UNDER the W loads:
        // one global round trip for the whole prologue instead of two
        constexpr int UNITS = NO * KJ * 2, UW = (UNITS + 511) / 512;
        float4 wr[UW];
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid;
            wr[it] = ldg4(W + (size_t)(u < UNITS ? u : 0) * 4);
        }
#if GAD_X_PHASES == 2
        GAD_PH_STAMP(st2);                               // first slab + W loads issued
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GAD_PH_STAMP(st3);                               // ... and arrived
#endif
        if (ACT && tid < KP) {                            // (KP <= 64 < 512 threads: one channel per thread)
            float sc, sh;
            if (x.bn.stat_sum) gad_bn_fin_channel(x.bn, tid, blockIdx.x == 0, sc, sh);
            else { sc = x.scale[tid]; sh = x.shift[tid]; }
            sv[tid] = sc; tv[tid] = sh;
        }
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid;
            const int n = u / (KJ * 2), c = (u % (KJ * 2)) * 4;
            if (SP) {
                unsigned h0, m0, l0, h1, m1, l1;
                gad_split2(wr[it].x, wr[it].y, h0, m0, l0);
                gad_split2(wr[it].z, wr[it].w, h1, m1, l1);
                unsigned char* wp = reinterpret_cast<unsigned char*>(Ws) + n * PB + c * 2;
                if (u < UNITS) {
                    *reinterpret_cast<uint2*>(wp) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(wp + NO * PB) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(wp + 2 * NO * PB) = make_uint2(l0, l1);
                }
            } else if (u < UNITS) *reinterpret_cast<float4*>(Ws + n * PW + c) = wr[it];
        }
    }
    __syncthreads();
#if GAD_X_PHASES == 2
    GAD_PH_STAMP(st4);
    if (ts && lane == 0) {
        auto c16 = [&](long long t) { const long long d = (t - kt_start) >> 2; return (unsigned long long)(d > 65535 ? 65535 : d); };
        ts[2 * (blockIdx.x * 8 + wave)] = c16(st1) | (c16(st2) << 16) | (c16(st3) << 32) | (c16(st4) << 48);
    }
#endif

#ifdef GAD_X_PHASES
    long long tA = 0, tB = 0, tC = 0, ph_top = 0, ph_mfma = 0, ph_epi = 0;
    int ph_n = 0;
    long long tP, tS;
    GAD_PH_STAMP(tP);
    tS = (long long)kt_start;
#endif
    for (; slab < slab_end; slab += stride) {
        load_slab(slab + stride, pt_nxt, rn);
        pt_nxt = load_pt(slab + 2 * stride);
        // the 16 rows this lane owns in the accumulator layout: 4 runs of 4 consecutive rows -> 4 aligned loads
        float4 w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r0 = slab * 32 + 8 * q + 4 * half;
            w4[q] = row_w ? ldg4(row_w + (r0 + 3 < n_rows_static ? r0 : 0)) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        if (slab * 32 + 32 > n_rows) {                   // ragged last slab (wave-uniform branch): zero the missing rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* wq = &w4[q].x;
#pragma unroll
                for (int e = 0; e < 4; ++e) wq[e] = (slab * 32 + 8 * q + 4 * half + e < n_rows) ? wq[e] : 0.f;
            }
        }
        int gl = -1;                                     // fused max-pool: lane i (both halves) holds the group of slab row i
        if (POOL) { const int rr = slab * 32 + l31; gl = rr < n_rows ? pe.row_grp[rr] : -1; }
        __asm__ volatile("" ::: "memory");        // keep the (loop-invariant) LDS reads of W inside the loop: registers
        f32x16 acc[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        float4 bn[TN];
        if (!SP) lds_b(0, bn);
        GAD_PH_STAMP(tA);
        if (SP) {
            const unsigned char* wp = reinterpret_cast<const unsigned char*>(Ws) + l31 * PB + 16 * half;
            constexpr int NS = SP ? KJ / 2 : 1;           // 16-channel steps
            gad_u32x4 AH[NS], AM[NS], AL[NS];
#pragma unroll
            for (int sg = 0; sg < NS; ++sg) {
                unsigned ah[4], am[4], al[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {           // relu(scale * z + shift) of the lane's 8 channels, then hi / mid / lo
                    const int c = 16 * sg + 8 * half + 4 * e;
                    const float4 s4 = *reinterpret_cast<const float4*>(sv + c), t4 = *reinterpret_cast<const float4*>(tv + c);
                    const float4 zr = ra[SP ? 2 * sg + e : 0].a;
                    const float a0 = __builtin_fmaxf(fmaf(zr.x, s4.x, t4.x), 0.f), a1 = __builtin_fmaxf(fmaf(zr.y, s4.y, t4.y), 0.f);
                    const float a2 = __builtin_fmaxf(fmaf(zr.z, s4.z, t4.z), 0.f), a3 = __builtin_fmaxf(fmaf(zr.w, s4.w, t4.w), 0.f);
                    gad_split2(a0, a1, ah[2 * e], am[2 * e], al[2 * e]);
                    gad_split2(a2, a3, ah[2 * e + 1], am[2 * e + 1], al[2 * e + 1]);
                }
                // the bf16 MFMA's adder truncates toward -inf (a -1e-8 relative bias, tools/ubench/split_bf16.hip): odd steps
                // accumulate the NEGATED products into a second accumulator pair, the difference of the two cancels it
                const unsigned sgn = ((sg ^ l31) & 1) ? 0x80008000u : 0u;        // (and the whole of an odd row: see the header)
                AH[sg] = gad_u32x4{ah[0] ^ sgn, ah[1] ^ sgn, ah[2] ^ sgn, ah[3] ^ sgn};
                AM[sg] = gad_u32x4{am[0] ^ sgn, am[1] ^ sgn, am[2] ^ sgn, am[3] ^ sgn};
                AL[sg] = gad_u32x4{al[0] ^ sgn, al[1] ^ sgn, al[2] ^ sgn, al[3] ^ sgn};
            }
#pragma unroll
            for (int pp = 0; pp < TN / 2; ++pp) {        // two 32-column tiles at a time: (plain, negated) accumulator pairs
                f32x16 an[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 16; ++v) an[i][v] = 0.f;
#pragma unroll
                for (int sg = 0; sg < NS; ++sg) {
                    gad_u32x4 BH[2], BM[2], BL[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned char* q = wp + (2 * pp + i) * 32 * PB + 32 * sg;
                        BH[i] = *reinterpret_cast<const gad_u32x4*>(q);
                        BM[i] = *reinterpret_cast<const gad_u32x4*>(q + NO * PB);
                        BL[i] = *reinterpret_cast<const gad_u32x4*>(q + 2 * NO * PB);
                    }
                    // the six products of weight >= 2^-16, smallest first
#define GAD_SPLIT_MFMA(A, B)                                                                                                   \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
                        if (sg & 1) an[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A[sg]), gad_as_bf16x8(B[i]), an[i], 0, 0, 0); \
                        else acc[2 * pp + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A[sg]), gad_as_bf16x8(B[i]), acc[2 * pp + i], 0, 0, 0); \
                    }
                    GAD_SPLIT_MFMA(AL, BH) GAD_SPLIT_MFMA(AH, BL) GAD_SPLIT_MFMA(AM, BM)
                    GAD_SPLIT_MFMA(AM, BH) GAD_SPLIT_MFMA(AH, BM) GAD_SPLIT_MFMA(AH, BH)
#undef GAD_SPLIT_MFMA
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[2 * pp + i][v] = (v & 1) ? an[i][v] - acc[2 * pp + i][v] : acc[2 * pp + i][v] - an[i][v];
            }
        }
        f32x16 zacc[RE ? 2 : 1];
        if (RE) {                   // z1^T tile: channels 32 b + (accumulator row), rows = lanes; the first layer's own products
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) zacc[b][v] = 0.f;
#pragma unroll
            for (int j = 0; j < PK; ++j) {
                const float4 xb = x_finish<1>(x, ra[j], true, 8 * j + 4 * half, sv, tv);
#pragma unroll
                for (int b = 0; b < 2; ++b) zacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[b][j].x, xb.x, zacc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) zacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[b][j].y, xb.y, zacc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) zacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[b][j].z, xb.z, zacc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) zacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[b][j].w, xb.w, zacc[b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < (SP ? 0 : KJ); ++j) {
            float4 b4[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) b4[t] = bn[t];
            float4 a4;
            if (ACT) {              // relu(scale * z + shift): two packed FMAs + four max
                const float4 s4 = *reinterpret_cast<const float4*>(sv + 8 * j + 4 * half);
                const float4 t4 = *reinterpret_cast<const float4*>(tv + 8 * j + 4 * half);
                float4 zr;
                if (RE) zr = make_float4(zacc[j >> 2][4 * (j & 3)], zacc[j >> 2][4 * (j & 3) + 1], zacc[j >> 2][4 * (j & 3) + 2],
                                         zacc[j >> 2][4 * (j & 3) + 3]);
                else zr = ra[RE ? 0 : j].a;
                const f32x2 lo = f32x2{zr.x, zr.y} * f32x2{s4.x, s4.y} + f32x2{t4.x, t4.y};
                const f32x2 hi = f32x2{zr.z, zr.w} * f32x2{s4.z, s4.w} + f32x2{t4.z, t4.w};
                a4 = make_float4(__builtin_fmaxf(lo.x, 0.f), __builtin_fmaxf(lo.y, 0.f), __builtin_fmaxf(hi.x, 0.f),
                                 __builtin_fmaxf(hi.y, 0.f));
            } else {
                a4 = x_finish<1>(x, ra[RE ? 0 : j], true, 8 * j + 4 * half, sv, tv);
            }
            if (j + 1 < KJ) lds_b(j + 1, bn);      // next group's fragments land under this group's MFMAs
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, acc[t], 0, 0, 0);
        }
        GAD_PH_STAMP(tB);
        // epilogue: raw layer output (SGPR row offset + immediate column offset: no address arithmetic) and the
        // weighted BatchNorm partial sums, two accumulator rows per packed instruction
        const int zrow = slab * 32 * NO * 4;
        if (POOL) {
            // 64 columns at a time through this wavefront's LDS tile (in-order LDS: no barrier), then lane = column
            float* zt = pz + wave * (32 * 64);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    zt[acc_row(v, half) * 64 + l31] = acc[2 * p][v];
                    zt[acc_row(v, half) * 64 + 32 + l31] = acc[2 * p + 1][v];
                }
                pool_scan32(zt + lane, 64, gl, slab * 32, psgn[p], prun[p], pe.key + p * 64 + lane, pe.C);
            }
        }
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const int rb = ((v & 3) + 8 * (v >> 2)) * NO * 4;
            const f32x2 wr = f32x2{(&w4[v >> 2].x)[v & 3], (&w4[v >> 2].x)[(v & 3) + 1]};
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const f32x2 zv = f32x2{acc[t][v], acc[t][v + 1]};
                if (store_z) {                           // (NULL zout: statistics / pooled maxima only; rows past n_rows are dropped)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][v]), zrsrc, zlane + t * 128, zrow + rb, GAD_STREAM_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][v + 1]), zrsrc, zlane + t * 128, zrow + rb + NO * 4, GAD_STREAM_STORE_AUX);
                }
                const f32x2 wz = wr * zv;
                csum[t] += wz;
                csq[t] += wz * zv;
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) ra[j] = rn[j];
#ifdef GAD_X_PHASES
        GAD_PH_STAMP(tC);
        ph_top += tA - tP; ph_mfma += tB - tA; ph_epi += tC - tB; tP = tC; ++ph_n;
#endif
    }
#ifdef GAD_X_PHASES
    if (ts && lane == 0) {
        const unsigned w = blockIdx.x * 8 + wave;
        if (GAD_X_PHASES != 2) ts[2 * w] = ((unsigned long long)ph_mfma << 32) | (unsigned long long)(ph_epi & 0xffffffffu);
        ts[2 * w + 1] = ((unsigned long long)ph_top << 32) | ((unsigned long long)(tP - tS) & 0xffffff00u) | (unsigned long long)ph_n;
    }
#endif
    if (POOL && range0 < slab_end * 32) {                // the group still open at the end of the range
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (prun[p].g >= 0)
                pool_put(pe.key + p * 64 + lane, prun[p].g, pe.C, prun[p].v, prun[p].row, prun[p].part != 0 || prun[p].g == g_after);
    }
    if (stat_sum) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float c0 = csum[t].x + csum[t].y, c1 = csq[t].x + csq[t].y;
            const float s0 = c0 + __shfl_xor(c0, 32, 64);
            const float s1 = c1 + __shfl_xor(c1, 32, 64);
            if (lane < 32) { red[wave * NO + t * 32 + lane] = s0; red[(8 + wave) * NO + t * 32 + lane] = s1; }
        }
        __syncthreads();
        if (tid < NO) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { s0 += red[w * NO + tid]; s1 += red[(8 + w) * NO + tid]; }
            const int rep = blockIdx.x % GAD_STAT_REPLICAS;
            atomic_add_f64(stat_sum + (size_t)rep * stat_stride + tid, (double)s0);
            atomic_add_f64(stat_sq + (size_t)rep * stat_stride + tid, (double)s1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinny forward for the small-M layers (FC head and actor/critic heads: M = batch rows <= 1024, K up to 1032).
// A 64x64-tile launch has M/64 * N/64 = 32..64 workgroups for 256 CUs and walks K serially: 33 K-tiles of
// (global load -> LDS -> barrier -> 16 MFMA -> barrier) at ~1.2 us each (load latency, nothing to overlap with).
// Here every workgroup owns one 32x32 output tile and its 8 wavefronts split K: each wavefront issues ALL the
// 16-byte loads of its K slice (A and B operand straight in MFMA register layout, k visited as 8j+4h+i like the
// streaming kernel, no LDS staging, no barrier) before its first MFMA, so the load latency is paid about once;
// the partial tiles are summed through LDS.  256 rows x 1024 -> 512: 37.8 us -> see tools/diag_gemm.py.
// ------------------------------------------------------------------------------------------------
#define SK_NW 8          // wavefronts per workgroup (K split); 16 -> 128-VGPR budget -> spills, 2x slower
#define SK_CH 6          // 8-wide k groups per register chunk (17 x 1 = the whole K share in one round of loads: measured 2 % slower)
#define SK_SP 36         // pitch (floats) of the wavefront-private operand stage of the skinny forward kernel
#define SK_KCAP (8 * SK_NW * SK_CH * SK_MAXCH)   // most input channels of a skinny launch (per-channel affine staged in LDS)
#define SK_MAXCH 3       // chunks per wavefront: K <= 8 * SK_NW * SK_CH * SK_MAXCH = 1152
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_fwd_skinny_kernel(XSrc x, Groups gr, int n_rows,
                                                                     const float* __restrict__ row_w,
                                                                     const float* __restrict__ W, int Kp,
                                                                     float* __restrict__ zout, int zout_pitch,
                                                                     double* __restrict__ stat_sum,
                                                                     double* __restrict__ stat_sq, int stat_stride, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    __shared__ __attribute__((aligned(16))) float stA[NW * 32 * SK_SP], stW[NW * 32 * SK_SP];
    __shared__ __attribute__((aligned(16))) float sv[SK_KCAP], tv[SK_KCAP];
    float* const part = stA;                                  // partial tiles reuse the operand stage: wavefront w's 4 KB inside its own 4.5 KB
    static_assert(32 * SK_SP >= 16 * 64, "partial tile must fit the wavefront's stage");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int g = blockIdx.z;
    const int zoff = gr.aoff[g], n_out = gr.nout[g], ooff = gr.ooff[g];
    // x = column tile, y = row tile: workgroups go to the 8 XCDs round-robin in x-fastest order, so the workgroups that read
    // one 32-column slice of W share an L2 (each XCD pulls 1/8 of the weights instead of all of them)
    const int n0 = blockIdx.x * 32, row0 = blockIdx.y * 32;
    if (n0 >= n_out) return;                                  // workgroup-uniform
    const float* Wg = W + gr.woff[g] + (size_t)min(n0 + l31, n_out - 1) * Kp + 4 * half;   // clamped: extra columns unused
    const int r = min(row0 + l31, max(n_rows - 1, 0));                // clamped: extra rows are neither stored nor counted
    const int nj = Kp >> 3;
    const int per = (nj + NW - 1) / NW;
    const int j0 = wave * per, j1 = min(nj, j0 + per);        // this wavefront's k groups (wave-uniform)

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // Operand blocks of four k groups (32 columns) go through a wavefront-private LDS stage: loaded COALESCED (8 lanes per
    // 128-byte row segment, 8 rows per instruction: 8-16 cache lines instead of the 32 a lane-per-row 16-byte load touches
    // -- these launches were bound by the texture path, not by the number of load rounds), written row-major with pitch 36
    // and read back as the MFMA fragments.  Blocks that reach beyond the bulk columns (bias / extra column) or the
    // wavefront's last groups take the lane-per-row loads.
    float* const myA = stA + wave * (32 * SK_SP);
    float* const myW = stW + wave * (32 * SK_SP);
    const int rsub = lane >> 3, chunk = lane & 7;
    const float* pa[4];
    const float* pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pa[i] = x.zin + (size_t)min(row0 + rsub + 8 * i, max(n_rows - 1, 0)) * x.zin_pitch + zoff + 4 * chunk;
        pw[i] = W + gr.woff[g] + (size_t)min(n0 + rsub + 8 * i, n_out - 1) * Kp + 4 * chunk;
    }
    auto staged = [&](int jb) { return jb + 4 <= j1 && 8 * (jb + 4) <= x.c_in; };       // wave-uniform
    float4 ga[4], gw[4];
    auto load_blk = [&](int jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ga[i] = ldg4(pa[i] + 8 * jb); gw[i] = ldg4(pw[i] + 8 * jb); }
    };
    if (j0 < j1 && staged(j0)) load_blk(j0);
    // the operand loads above are in flight while the input layer's per-channel affine is staged in LDS
    if (x.bn.stat_sum) {                                      // the input layer's BatchNorm is finalised here (one group, zoff == 0)
        const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
        for (int i = tid; i < x.c_in; i += 64 * NW) {
            float sc, sh;
            gad_bn_fin_channel(x.bn, i, writer, sc, sh);
            sv[i] = sc;
            tv[i] = sh;
        }
    } else if (x.affine) stage_affine<64 * NW>(sv, tv, x, zoff, x.c_in);
    __syncthreads();
    for (int jb = j0; jb < j1; jb += 4) {
        if (staged(jb)) {                                     // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(myA + (rsub + 8 * i) * SK_SP + 4 * chunk) = ga[i];
                *reinterpret_cast<float4*>(myW + (rsub + 8 * i) * SK_SP + 4 * chunk) = gw[i];
            }
            if (jb + 4 < j1 && staged(jb + 4)) load_blk(jb + 4);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                XRaw raw;
                raw.a = *reinterpret_cast<const float4*>(myA + l31 * SK_SP + 8 * u + 4 * half);
                raw.s = f4zero();
                const float4 a4 = x_finish<0>(x, raw, true, 8 * (jb + u) + 4 * half, sv, tv);
                const float4 b4 = *reinterpret_cast<const float4*>(myW + l31 * SK_SP + 8 * u + 4 * half);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            const int je = min(jb + 4, j1);
            XRaw ra[4];
            float4 rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(jb + u, nj - 1);
                const int col = 8 * j + 4 * half;
                ra[u] = x_raw<0>(x, r, true, zoff, col, col + 4 > x.c_in, 0);
                rb[u] = ldg4(Wg + 8 * j);
            }
            if (jb + 4 < j1 && staged(jb + 4)) load_blk(jb + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (jb + u < je) {                            // wave-uniform
                    const float4 a4 = x_finish<0>(x, ra[u], true, 8 * (jb + u) + 4 * half, sv, tv);
                    const float4 b4 = rb[u];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) part[wave * (32 * SK_SP) + v * 64 + lane] = acc[v];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] += part[w * (32 * SK_SP) + v * 64 + lane];
    const int n = n0 + l31;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int rr = row0 + acc_row(v, half);
        const float zv = acc[v];
        const bool live = rr < n_rows;
        if (live && n < n_out) zout[(size_t)rr * zout_pitch + ooff + n] = zv;
        const float w = live ? (row_w ? row_w[rr] : 1.f) : 0.f;
        s1 = fmaf(w, zv, s1);
        s2 = fmaf(w * zv, zv, s2);
    }
    if (stat_sum) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32 && n < n_out) {
            const int rep = blockIdx.y % GAD_STAT_REPLICAS;
            atomic_add_f64(stat_sum + (size_t)rep * stat_stride + ooff + n, (double)s1);
            atomic_add_f64(stat_sq + (size_t)rep * stat_stride + ooff + n, (double)s2);
        }
    }
}

#define GAD_GX_CAP 2048           // most row tiles a tile launch spreads over workgroups (the rest by its grid-stride loop)
static int g_opt_skinny_nw = SK_NW;    // wavefronts per skinny workgroup: 8, or 4 (A/B: a quarter of a CU's registers and < 60 KB of LDS, so a workgroup fits beside the side lanes' wide kernels)
static int g_opt_fwd_skinny = 1, g_opt_dx_skinny = 1, g_opt_dx_stream = 1, g_opt_dw_skinny = 1, g_opt_dw_stream = 1;
static bool fwd_skinny(const gad_gemm_fwd_args& a) {
    return g_opt_fwd_skinny && a.mode == 0 && !a.n_rows_dev && a.n_rows <= 1024 && a.Kp <= SK_KCAP;
}


static int g_opt_fwd_stream = 1;
#define DW_STREAM_SPLITS 256
static int g_opt_bwd_stream_wgs = DW_STREAM_SPLITS;   // persistent workgroups (= partial dW blocks) of the fused SA1 backward: A/B, <= DW_STREAM_SPLITS
static int g_opt_fwd_stream_wgs = 256;         // persistent workgroups of the streaming forward's layers 2 / 3 (A/B: fewer leave CUs to the sibling pass's small launches)
static int g_opt_fwd_stream_l1_wgs = 256;      // persistent workgroups of the streaming forward's gathered first layer (A/B: 512 = two per CU)

extern "C" int gad_set_option(const char* name, int value) {
    GAD_REQUIRE(name, GAD_ERR_NULL, "set_option: null name");
    if (!strcmp(name, "fwd_stream")) { g_opt_fwd_stream = value; return GAD_OK; }
    if (!strcmp(name, "bwd_stream_wgs")) { g_opt_bwd_stream_wgs = (value > 0 && value <= DW_STREAM_SPLITS) ? value : DW_STREAM_SPLITS; return GAD_OK; }
    if (!strcmp(name, "fwd_stream_wgs")) { g_opt_fwd_stream_wgs = value > 0 ? value : 256; return GAD_OK; }
    if (!strcmp(name, "fwd_stream_l1_wgs")) { g_opt_fwd_stream_l1_wgs = value > 0 ? value : 256; return GAD_OK; }
    if (!strcmp(name, "fwd_wide")) { g_opt_fwd_wide = value; return GAD_OK; }
    if (!strcmp(name, "dx_wide")) { g_opt_dx_wide = value; return GAD_OK; }
    if (!strcmp(name, "dw_wide")) { g_opt_dw_wide = value; return GAD_OK; }
    if (!strcmp(name, "bwd_fused")) { g_opt_bwd_fused = value; return GAD_OK; }
    if (!strcmp(name, "bwd_wide")) { g_opt_bwd_wide = value; return GAD_OK; }
    if (!strcmp(name, "fwd_bn_prologue")) { g_opt_fwd_bn_prologue = value; return GAD_OK; }
    if (!strcmp(name, "bwd_wide_slab")) { g_opt_bwd_wide_slab = value > 0 ? value : 4; return GAD_OK; }
    if (!strcmp(name, "mfma_split")) { g_opt_mfma_split = value; return GAD_OK; }
    if (!strcmp(name, "dw_wide_wgs")) { g_opt_dw_wide_wgs = value > 0 ? value : 256; return GAD_OK; }
    if (!strcmp(name, "skinny_nw")) { g_opt_skinny_nw = value == 4 ? 4 : SK_NW; return GAD_OK; }
    if (!strcmp(name, "fwd_skinny")) { g_opt_fwd_skinny = value; return GAD_OK; }
    if (!strcmp(name, "dx_skinny")) { g_opt_dx_skinny = value; return GAD_OK; }
    if (!strcmp(name, "dx_stream")) { g_opt_dx_stream = value; return GAD_OK; }
    if (!strcmp(name, "dw_skinny")) { g_opt_dw_skinny = value; return GAD_OK; }
    if (!strcmp(name, "dw_stream")) { g_opt_dw_stream = value; return GAD_OK; }
    int found = 0;
    gad_geometry_set_option(name, value, &found);                                     // geometry.hip: "bq_cells"
    if (found) return GAD_OK;
    GAD_REQUIRE(false, GAD_ERR_SHAPE, "set_option: unknown option '%s'", name);
    return GAD_OK;
}

// the streaming kernel covers: one group, no bias / extra column, Kp in {16, 64}, n_out in {64, 128}
static bool fwd_streamable(const gad_gemm_fwd_args& a) {
    if (!g_opt_fwd_stream) return false;
    if (a.pool_key && !(a.mode == 0 && a.n_out[0] == 128)) return false;          // pooled instantiation: 64 -> 128 only
    if (a.n_groups != 1 || a.zin_off[0] != 0 || a.w_off[0] != 0 || a.out_off[0] != 0) return false;
    if (a.n_rows < 32768 || (a.n_out[0] != 64 && a.n_out[0] != 128) || (a.zout && a.zout_pitch != a.n_out[0])) return false;
    if ((long long)a.n_rows * a.n_out[0] * 4 > (1ll << 31)) return false;              // buffer-descriptor byte offsets
    if (a.mode == 0) return a.Kp == 64 && a.c_in == 64 && a.ones_col < 0 && !a.extra && (a.scale && a.shift) && a.relu;
    const bool sa1_rows = a.feat_c == 4 && (a.action ? a.act_c == 6 : a.act_c == 0);
    if (a.mode == 2)                                                             // 64-channel input recomputed from SA1's gathered rows
        return a.Kp == 64 && a.c_in == 64 && a.n_out[0] == 64 && !a.pool_key && a.ones_col < 0 && !a.extra && a.scale && a.shift && a.relu &&
               a.pre_W && (a.pre_Kp == 8 || a.pre_Kp == 16) && sa1_rows && a.feat_c + 3 + a.act_c <= a.pre_Kp;
    // SA1's gathered rows: [f (4) | dx (3)] (policy encoder, Kp 8) or [f (4) | dx (3) | action (6)] (value encoder, Kp 16)
    return (a.Kp == 16 || a.Kp == 8) && a.feat_c == 4 && (a.action ? a.act_c == 6 : a.act_c == 0) && a.feat_c + 3 + a.act_c <= a.Kp;
}

static Groups make_groups(int n, const int32_t* a, const int32_t* w, const int32_t* o, const int32_t* no) {
    Groups g;
    g.n = n;
    for (int i = 0; i < GAD_MAX_GROUPS; ++i) {
        g.aoff[i] = (i < n && a) ? a[i] : 0; g.woff[i] = (i < n && w) ? w[i] : 0;
        g.ooff[i] = (i < n && o) ? o[i] : 0; g.nout[i] = i < n ? no[i] : 0;
    }
    return g;
}

static int max_nout(const Groups& g) { int m = 0; for (int i = 0; i < g.n; ++i) m = g.nout[i] > m ? g.nout[i] : m; return m; }

static int check_input(const gad_gemm_fwd_args& a, const char* who) {
    if (a.mode == 2) {
        GAD_REQUIRE(a.src_xyz && a.row_pt && a.row_grp && a.feat && a.pre_W, GAD_ERR_NULL, "%s: recomputed input needs the gather inputs and pre_W", who);
        GAD_REQUIRE(a.act_c == 0 || a.action, GAD_ERR_NULL, "%s: action", who);
        GAD_REQUIRE(fwd_streamable(a), GAD_ERR_SHAPE, "%s: mode 2 (recomputed 64-channel input) is a streaming-kernel form: SA1 rows, "
                    "Kp = c_in = n_out = 64, rows >= 32768, pre_Kp 8 / 16, scale / shift, relu", who);
        return GAD_OK;
    }
    if (a.mode == 0) {
        GAD_REQUIRE(a.zin && a.c_in % 4 == 0 && a.c_in >= 4 && a.c_in <= VMAX && a.zin_pitch % 4 == 0, GAD_ERR_SHAPE,
                    "%s: ACT input needs 4 <= c_in <= %d, c_in and pitch multiples of 4", who, VMAX);
        for (int i = 0; i < a.n_groups; ++i)
            GAD_REQUIRE(a.zin_off[i] % 4 == 0, GAD_ERR_SHAPE, "%s: zin_off must be a multiple of 4", who);
    } else {
        GAD_REQUIRE(a.src_xyz && a.row_pt && a.row_grp && a.feat, GAD_ERR_NULL, "%s: gather inputs", who);
        GAD_REQUIRE(a.feat_c % 4 == 0 && a.feat_c >= 4, GAD_ERR_SHAPE, "%s: feat_c must be a positive multiple of 4", who);
        GAD_REQUIRE(a.act_c == 0 || a.action, GAD_ERR_NULL, "%s: action", who);
    }
    return GAD_OK;
}

extern "C" int gad_gemm_fwd(const gad_gemm_fwd_args* a, void* stream) {
    unsigned long long* ts = gad_take_timing_slot(stream);
    const int rows_hint = gad_take_grid_rows();
    GAD_REQUIRE(a && a->W && (a->zout || a->pool_key || a->stat_sum), GAD_ERR_NULL, "gemm_fwd: null pointer");
    GAD_REQUIRE(a->n_groups >= 1 && a->n_groups <= GAD_MAX_GROUPS, GAD_ERR_SHAPE, "gemm_fwd: n_groups");
    GAD_REQUIRE(a->Kp % 8 == 0 && a->Kp >= 8, GAD_ERR_SHAPE, "gemm_fwd: Kp=%d must be a multiple of 8", a->Kp);
    if (int e = check_input(*a, "gemm_fwd")) return e;
    if (a->n_rows <= 0) return GAD_OK;
    XSrc x = make_xsrc(*a);
    Groups gr = make_groups(a->n_groups, a->zin_off, a->w_off, a->out_off, a->n_out);
    const int nmax = max_nout(gr);
    hipStream_t st = (hipStream_t)stream;
    const int rows = a->n_rows;
    PoolEpi pe;
    pe.key = reinterpret_cast<unsigned long long*>(a->pool_key); pe.row_grp = a->pool_row_grp; pe.gamma = a->pool_gamma;
    pe.C = a->n_out[0];
    if (pe.key) {
        GAD_REQUIRE(pe.row_grp && pe.gamma, GAD_ERR_NULL, "gemm_fwd: fused max-pool needs pool_row_grp and pool_gamma");
        GAD_REQUIRE(a->mode == 0 && a->n_groups == 1 && a->out_off[0] == 0 && a->n_out[0] % 64 == 0 && rows % 4 == 0,
                    GAD_ERR_SHAPE, "gemm_fwd: fused max-pool needs an ACT input, one group, n_out %% 64 == 0, rows %% 4 == 0 (got n_out=%d rows=%d)",
                    a->n_out[0], rows);
    }
    const int grid_rows = (a->n_rows_dev && rows_hint > 0 && rows_hint < rows) ? rows_hint : rows;   // gad_grid_rows_hint
    if (a->in_stat_sum) {
        GAD_REQUIRE(a->mode != 1 && a->n_groups == 1 && a->zin_off[0] == 0 && a->in_stat_sq && a->in_gamma && a->in_beta && a->scale && a->shift,
                    GAD_ERR_NULL, "gemm_fwd: input-layer BatchNorm block needs an ACT input, one group, in_stat_sq, in_gamma, in_beta, scale, shift");
        const bool has_prologue = fwd_wideable(*a) || (fwd_streamable(*a) && a->mode != 1) || (!pe.key && fwd_skinny(*a));
        if (!(has_prologue && g_opt_fwd_bn_prologue)) {               // this route reads scale / shift as given: finalise first
            if (int e = gad_bn_finalize(a->in_stat_sum, a->in_stat_sq, a->in_stat_stride, a->in_gamma, a->in_beta, a->c_in, a->in_count,
                                        a->in_eps, a->in_momentum, a->in_running_mean, a->in_running_var, const_cast<float*>(a->scale),
                                        const_cast<float*>(a->shift), a->in_mean, a->in_istd, stream)) return e;
            x.bn.stat_sum = nullptr;
        }
    }
    if (!pe.key && fwd_skinny(*a)) {
        // (a NULL zout -- statistics / pooled maxima only -- is honoured by the streaming, wide-tile and 64 x 64 kernels; this one stores unguarded)
        GAD_REQUIRE(a->zout, GAD_ERR_NULL, "gemm_fwd: zout == NULL is not supported for small-M (skinny) shapes");
        if (g_opt_skinny_nw == 4)
            hipLaunchKernelGGL(gemm_fwd_skinny_kernel<4>, dim3(gad_cdiv(nmax, 32), gad_cdiv(rows, 32), gr.n), dim3(64 * 4), 0, st, x,
                               gr, rows, a->row_w, a->W, a->Kp, a->zout, a->zout_pitch, a->stat_sum, a->stat_sq, a->stat_stride, ts);
        else
            hipLaunchKernelGGL(gemm_fwd_skinny_kernel<SK_NW>, dim3(gad_cdiv(nmax, 32), gad_cdiv(rows, 32), gr.n), dim3(64 * SK_NW), 0, st, x,
                               gr, rows, a->row_w, a->W, a->Kp, a->zout, a->zout_pitch, a->stat_sum, a->stat_sq, a->stat_stride, ts);
        GAD_CHECK_LAUNCH("gemm_fwd(skinny)");
        return GAD_OK;
    }
#define LAUNCH_FWD2(WM, WN, TM, TN, XM, POOL)                                                                  \
    do {                                                                                                   \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                                \
        int gx = gad_cdiv(grid_rows, BM); if (gx > GAD_GX_CAP) gx = GAD_GX_CAP;                                        \
        hipLaunchKernelGGL((gemm_fwd_kernel<WM, WN, TM, TN, XM, POOL>), dim3(gx, gad_cdiv(nmax, BN), gr.n), dim3(256), \
                           0, st, x, gr, a->n_rows_dev, rows, a->row_w, a->W, a->Kp, a->zout, a->zout_pitch, \
                           a->stat_sum, a->stat_sq, a->stat_stride, pe, ts);                               \
    } while (0)
#define LAUNCH_FWD(WM, WN, TM, TN) do { if (a->mode == 0) LAUNCH_FWD2(WM, WN, TM, TN, 0, false); else LAUNCH_FWD2(WM, WN, TM, TN, 1, false); } while (0)
    if (fwd_streamable(*a)) {
        const int slabs = gad_cdiv(rows, 32);
        // one 8-wavefront workgroup per CU; the gathered first layer (K <= 16: ~100 registers, latency-bound on its five dependent
        // loads per row) may run more (option "fwd_stream_l1_wgs")
        const int gcap = a->mode == 1 ? g_opt_fwd_stream_l1_wgs : g_opt_fwd_stream_wgs;
        int gx = gad_cdiv(slabs, 8); if (gx > gcap) gx = gcap;
#define LAUNCH_STREAM(KJ, TN, XM, POOL, ...)                                                               \
        hipLaunchKernelGGL((gemm_fwd_stream_kernel<KJ, TN, XM, POOL, ##__VA_ARGS__>), dim3(gx), dim3(512), 0, st, x, a->n_rows_dev, rows, \
                           a->row_w, a->W, a->zout, a->stat_sum, a->stat_sq, a->stat_stride, pe, ts)
        if (split_on(GAD_SPLIT_FWD_STREAM) && a->mode == 0) {             // split-bf16 products (the kernel splits W itself)
            if (pe.key) LAUNCH_STREAM(8, 4, 0, true, true);
            else if (a->n_out[0] == 64) LAUNCH_STREAM(8, 2, 0, false, true);
            else LAUNCH_STREAM(8, 4, 0, false, true);
        } else if (pe.key) {
            LAUNCH_STREAM(8, 4, 0, true);                                  // (fwd_streamable: ACT input, 128 outputs)
        } else {
            if (a->mode == 2) { if (a->pre_Kp == 8) LAUNCH_STREAM(8, 2, 2, false); else LAUNCH_STREAM(8, 2, 3, false); }
            else if (a->mode == 0) { if (a->n_out[0] == 64) LAUNCH_STREAM(8, 2, 0, false); else LAUNCH_STREAM(8, 4, 0, false); }
            else if (a->Kp == 8) { if (a->n_out[0] == 64) LAUNCH_STREAM(1, 2, 1, false); else LAUNCH_STREAM(1, 4, 1, false); }
            else { if (a->n_out[0] == 64) LAUNCH_STREAM(2, 2, 1, false); else LAUNCH_STREAM(2, 4, 1, false); }
        }
#undef LAUNCH_STREAM
        if (split_on(GAD_SPLIT_FWD_STREAM) && a->mode == 0) GAD_CHECK_LAUNCH("gemm_fwd(stream split)"); else GAD_CHECK_LAUNCH("gemm_fwd(stream)");
        return GAD_OK;
    }
    if (fwd_wideable(*a)) {
        int gx = gad_cdiv(grid_rows, 64); if (gx > GAD_GX_CAP) gx = GAD_GX_CAP;
        const dim3 grid(gx, a->n_out[0] / 128);
        // split-bf16 form: family bit set and the call carries the forward weight mirror of exactly the columns the K loop covers
        const bool sp = split_on(GAD_SPLIT_FWD_WIDE) && a->W_split && a->W_split_pitch == (a->mode == 1 ? a->feat_c : a->Kp) &&
                        a->W_split_plane >= a->n_out[0] * a->W_split_pitch;
#define LAUNCH_WIDE(XM, POOL, SP)                                                                                              \
        hipLaunchKernelGGL((gemm_fwd_wide_kernel<XM, POOL, SP>), grid, dim3(256), 0, st, x, a->n_rows_dev, rows, a->row_w, a->W, a->Kp, \
                           a->n_out[0], a->zout, a->zout_pitch, a->stat_sum, a->stat_sq, a->stat_stride, pe, a->W_split, a->W_split_pitch, \
                           a->W_split_plane, ts)
        if (a->mode == 1) { if (sp) LAUNCH_WIDE(1, false, true); else LAUNCH_WIDE(1, false, false); }
        else if (pe.key) { if (sp) LAUNCH_WIDE(0, true, true); else LAUNCH_WIDE(0, true, false); }
        else { if (sp) LAUNCH_WIDE(0, false, true); else LAUNCH_WIDE(0, false, false); }
#undef LAUNCH_WIDE
        if (sp) GAD_CHECK_LAUNCH("gemm_fwd(wide split)"); else GAD_CHECK_LAUNCH("gemm_fwd(wide)");
        return GAD_OK;
    }
    // 64 x 64 tiles throughout: with K <= 1024 these launches are prologue/epilogue-bound, more and smaller
    // workgroups win over the 128-wide tiles at every shape of the step (tools/diag_gemm.py)
    if (pe.key) LAUNCH_FWD2(2, 2, 1, 1, 0, true);
    else if (nmax <= 32) LAUNCH_FWD(4, 1, 1, 1); else LAUNCH_FWD(2, 2, 1, 1);
#undef LAUNCH_FWD
#undef LAUNCH_FWD2
    GAD_CHECK_LAUNCH("gemm_fwd");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// backward wrt the layer input:  gout[r][k] = sum_n dZ[r][n] * W[n][k]
// ------------------------------------------------------------------------------------------------
struct DxEpi {
    int mode; float* gout; int gout_pitch; int k_valid;
    const float* zprev; int zprev_pitch; const float* ps; const float* pt; const float* pm; const float* pi;
    double* dbeta; double* dgamma; int stat_stride; int store_masked;
    float* dfeat; int feat_c; const int32_t* row_pt; const int32_t* row_grp; double* daction; int act_c; int gps;
};

template <int WM, int WN, int TM, int TN, bool VEC, int VM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_dx_kernel(DzSrc d, Groups gr, const int32_t* __restrict__ n_rows_dev,
                                                       int n_rows_static, const float* __restrict__ W, int Kp,
                                                       DxEpi e, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchT<BM>::v, PB = PitchD<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    constexpr int OFFB = (KT * PA + 3) & ~3;
    constexpr int TILE = OFFB + KT * PB;
    constexpr int SM = TILE + 3 * BM + (VEC ? 5 * VM : 4);
    static_assert(TILE >= 2 * WM * BN, "reduction scratch must fit in the tile LDS");
    __shared__ __attribute__((aligned(16))) float smem[SM];
    float* As = smem;
    float* Bs = smem + OFFB;
    int32_t* ptS = reinterpret_cast<int32_t*>(smem + TILE);
    int32_t* grS = ptS + BM;
    float* wS = reinterpret_cast<float*>(grS + BM);
    float* vec = wS + BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], n_out = gr.nout[g], goff = gr.ooff[g];
    const float* Wg = W + gr.woff[g];
    const int k0out = blockIdx.y * BN;
    if (k0out >= e.k_valid) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    if ((int)(blockIdx.x * BM) >= n_rows) return;
    const int nk = gad_cdiv_dev(n_out, KT);
    if (VEC) stage_dz_vecs<VM>(vec, d, doff, n_out);
    const bool need_grp = e.mode == 1 || d.gmode != 0;

    float cb[TN], cg[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { cb[t] = 0.f; cg[t] = 0.f; }

    for (int row0 = blockIdx.x * BM; row0 < n_rows; row0 += gridDim.x * BM) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
        if (tid < BM) {
            const int r = row0 + tid;
            const bool ok = r < n_rows;
            wS[tid] = (ok && d.row_w) ? d.row_w[r] : 1.f;
            if (e.mode == 1) ptS[tid] = ok ? e.row_pt[r] : 0;
            if (need_grp) grS[tid] = ok ? (e.mode == 1 ? e.row_grp[r] : d.row_grp[r]) : 0;
        }
        __syncthreads();

        DzRaw ra[UA];
        float4 rb[UB];
        auto load_tile = [&](int kt) {
            const int nb = kt * KT;
#pragma unroll
            for (int it = 0; it < UA; ++it) {
                int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
                const int r = row0 + i;
                ra[it] = dz_raw<VEC>(d, r, r < n_rows, doff, nb + kk, n_out, need_grp ? grS[i] : 0);
            }
#pragma unroll
            for (int it = 0; it < UB; ++it) {
                int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
                const int n = nb + kk, k = k0out + j;
                const bool ok = n < n_out && k < Kp;
                rb[it] = ldg4(Wg + (size_t)(ok ? n : 0) * Kp + (ok ? k : 0));
            }
        };
        auto store_tile = [&](int kt) {
            const int nb = kt * KT;
#pragma unroll
            for (int it = 0; it < UA; ++it) {
                int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
                const int r = row0 + i;
                store_T<BM>(As, i, kk, dz_finish<VEC, VM>(d, ra[it], r, r < n_rows, nb + kk, n_out, wS[i], vec));
            }
#pragma unroll
            for (int it = 0; it < UB; ++it) {
                int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
                const bool ok = (nb + kk) < n_out && (k0out + j) < Kp;
                store_D<BN>(Bs, kk, j, f4sel(ok, rb[it], f4zero()));
            }
        };
        load_tile(0);
        for (int kt = 0; kt < nk; ++kt) {
            store_tile(kt);
            __syncthreads();
            if (kt + 1 < nk) load_tile(kt + 1);
            mfma_ktile<TM, TN, PA, PB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
            __syncthreads();
        }
        const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int k = k0out + wn * TN * 32 + tn * 32 + l31;
            const bool kok = k < e.k_valid;
            float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
            const bool stats = e.dbeta != nullptr && kok;
            if (stats) { sc = e.ps[goff + k]; sh = e.pt[goff + k]; mu = e.pm[goff + k]; is = e.pi[goff + k]; }
            float sb = 0.f, sg = 0.f;
            int run_smp = -1;
            double run_sum = 0.0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int il = wm * TM * 32 + tm * 32 + acc_row(v, half);
                    const int r = row0 + il;
                    if (r >= n_rows || !kok) continue;
                    const float gv = acc[tm][tn][v];
                    if (e.mode == 0) {
                        float outv = gv;
                        if (stats) {
                            const float zp = e.zprev[(size_t)r * e.zprev_pitch + goff + k];
                            if (fmaf(zp, sc, sh) > 0.f) { sb += gv; sg = fmaf(gv, (zp - mu) * is, sg); }
                            else if (e.store_masked) outv = 0.f;
                        }
                        e.gout[(size_t)r * e.gout_pitch + goff + k] = outv;
                    } else {
                        // gather-layer columns: [feat (feat_c) | xyz (3) | action (act_c)]
                        if (k < e.feat_c) {
                            if (e.dfeat) atomic_add_f32(e.dfeat + (size_t)ptS[il] * e.feat_c + k, gv);
                        } else if (k >= e.feat_c + 3 && k < e.feat_c + 3 + e.act_c) {
                            // per-sample action gradient: ~875 rows of a sample add into the same 6 addresses.  The rows a
                            // lane holds are consecutive runs of a tile, almost always of ONE sample: accumulate the run
                            // in a register and issue one f64 atomic per run instead of one per row
                            if (e.daction) {
                                const int smp = grS[il] / e.gps;
                                if (smp != run_smp) {
                                    if (run_smp >= 0) atomic_add_f64(e.daction + (size_t)run_smp * e.act_c + (k - e.feat_c - 3), run_sum);
                                    run_smp = smp; run_sum = 0.0;
                                }
                                run_sum += (double)gv;
                            }
                        }
                    }
                }
            if (run_smp >= 0) atomic_add_f64(e.daction + (size_t)run_smp * e.act_c + (k - e.feat_c - 3), run_sum);
            cb[tn] += sb; cg[tn] += sg;
        }
        __syncthreads();
    }
    if (e.dbeta) {
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        block_column_atomics<WM, WN, TN>(smem, cb, cg, lane, wm, wn, k0out, e.k_valid,
                                         e.dbeta + (size_t)rep * e.stat_stride + goff,
                                         e.dgamma + (size_t)rep * e.stat_stride + goff);
    }
}

// ------------------------------------------------------------------------------------------------
// streaming dX for the SA1 layers (rows ~ 2e5, n_out in {64,128} reduced onto 64 input channels): the backward twin
// of gemm_fwd_stream_kernel.  HBM-bound (1 KB per row: z and dY in, dY out, z_prev for the BatchNorm-backward sums).
//   * W^T staged once per workgroup in LDS ([k][n_out+4]: B fragments are conflict-free ds_read_b128);
//   * the A operand dZ[r][8j+4h..+3] is produced in registers from 16-byte loads of z and dY (or of the pooled
//     arg-max / gradient pair), four k-groups per register chunk, the next chunk (or the next slab's first chunk)
//     in flight during the current chunk's MFMAs;
//   * epilogue: dY of the previous layer through buffer stores (SGPR row offsets) + that layer's dbeta / dgamma sums;
//     its z_prev loads are issued before the last chunk's MFMAs.
// ------------------------------------------------------------------------------------------------
template <int NJ, int GM>
__global__ __launch_bounds__(512, 2) void gemm_dx_stream_kernel(DzSrc d, const int32_t* __restrict__ n_rows_dev,
                                                                 int n_rows_static, const float* __restrict__ W, DxEpi e, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int NO = 8 * NJ, PW = NO + 4, NCH = NJ / 4;
    __shared__ __attribute__((aligned(16))) float Wt[64 * PW];
    __shared__ __attribute__((aligned(16))) float vec[3 * NO];          // P | Q | S (the ReLU mask is already in dY: premasked)
    __shared__ float red[2 * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    {   // W (n_out x 64, row-major) -> W^T in LDS
        constexpr int UW = NO * 16 / 512;
        float4 wr[UW];
#pragma unroll
        for (int it = 0; it < UW; ++it) wr[it] = ldg4(W + (size_t)(it * 512 + tid) * 4);
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid;
            const int n = u >> 4, k4 = (u & 15) * 4;
            Wt[(k4 + 0) * PW + n] = wr[it].x; Wt[(k4 + 1) * PW + n] = wr[it].y;
            Wt[(k4 + 2) * PW + n] = wr[it].z; Wt[(k4 + 3) * PW + n] = wr[it].w;
        }
    }
    for (int i = tid; i < NO; i += 512) {
        float P, Q, S;
        dz_coef(d, i, P, Q, S, first_workgroup());
        vec[i] = P; vec[NO + i] = Q; vec[2 * NO + i] = S;
    }
    __syncthreads();
    // previous layer's BatchNorm vectors of this lane's two output columns
    float ps[2], pt[2], pm[2], pi[2];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
        const int k = tk * 32 + l31;
        ps[tk] = e.ps[k]; pt[tk] = e.pt[k]; pm[tk] = e.pm[k]; pi[tk] = e.pi[k];
    }
    const int n_slabs = (n_rows + 31) >> 5;
    const int stride = gridDim.x * 8;
    int slab = wave * gridDim.x + blockIdx.x;                 // wave-uniform; same dealing as the forward kernel
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(e.gout, 0, gad_nbytes(n_rows, 64 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t zrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.zprev), 0, gad_nbytes(n_rows_static, 64 * 4), 0x00020000);
    const int glane = (4 * half * 64 + l31) * 4;

    float sb[2] = {0.f, 0.f}, sg[2] = {0.f, 0.f};
    float4 rz[2][4], rg[2][4];
    int4 ra[2][4];
    auto row_of = [&](int sl) { return min(sl < n_slabs ? sl * 32 + l31 : 0, max(n_rows - 1, 0)); };
    auto load_chunk = [&](int r, int grp, int c, int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned n = 8 * (4 * c + u) + 4 * half;
            rz[buf][u] = ldg4(d.z + ((unsigned)r * NO + n));
            if (GM == 0) {
                rg[buf][u] = ldg4(d.G + ((unsigned)r * NO + n));
            } else {
                ra[buf][u] = *reinterpret_cast<const int4*>(d.argmax + ((unsigned)grp * NO + n));
                rg[buf][u] = ldg4(d.dout + ((unsigned)grp * NO + n));
            }
        }
    };
    int r_cur = row_of(slab);
    int grp_cur = GM == 1 ? d.row_grp[r_cur] : 0;
    load_chunk(r_cur, grp_cur, 0, 0);
    for (; slab < n_slabs; slab += stride) {
        const int r_nxt = row_of(slab + stride);
        const int grp_nxt = GM == 1 ? d.row_grp[r_nxt] : 0;
        const float wrow = d.row_w ? d.row_w[r_cur] : 1.f;
        __asm__ volatile("" ::: "memory");            // keep loop-invariant LDS reads inside the loop (registers)
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        float zp[2][16];
        const int zrow = slab * 32 * 64 * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) load_chunk(r_cur, grp_cur, c + 1, (c + 1) & 1);
            else {
                load_chunk(r_nxt, grp_nxt, 0, 0);
#pragma unroll
                for (int v = 0; v < 16; ++v)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        zp[t][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            zrsrc, glane + t * 128, zrow + ((v & 3) + 8 * (v >> 2)) * 256, 0));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = 8 * (4 * c + u) + 4 * half;
                const float4 z = rz[c & 1][u];
                float4 g = rg[c & 1][u];
                if (GM == 1) {
                    const int4 a = ra[c & 1][u];
                    const int rr = slab * 32 + l31;                 // true row (clamped rows never match: rr >= n_rows > any arg-max)
                    g.x = a.x == rr ? g.x : 0.f; g.y = a.y == rr ? g.y : 0.f;
                    g.z = a.z == rr ? g.z : 0.f; g.w = a.w == rr ? g.w : 0.f;
                }
                const float4 P = *reinterpret_cast<const float4*>(vec + n);
                const float4 Q = *reinterpret_cast<const float4*>(vec + NO + n);
                const float4 S = *reinterpret_cast<const float4*>(vec + 2 * NO + n);
                float4 a4;
                a4.x = P.x * g.x - wrow * fmaf(S.x, z.x, Q.x); a4.y = P.y * g.y - wrow * fmaf(S.y, z.y, Q.y);
                a4.z = P.z * g.z - wrow * fmaf(S.z, z.z, Q.z); a4.w = P.w * g.w - wrow * fmaf(S.w, z.w, Q.w);
                const float4 b0 = *reinterpret_cast<const float4*>(Wt + l31 * PW + n);
                const float4 b1 = *reinterpret_cast<const float4*>(Wt + (32 + l31) * PW + n);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
            }
        }
        // epilogue: dY of the previous layer + its BatchNorm-backward sums (rows past n_rows: the store is dropped by
        // the buffer bounds only for the voffset part, so ragged slabs are predicated; their sums are masked)
        const bool full = slab * 32 + 32 <= n_rows;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = acc_row(v, half);
            const bool live = slab * 32 + row < n_rows;
            const int rb = zrow + ((v & 3) + 8 * (v >> 2)) * 256;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float gv = acc[t][v];
                const float zv = live ? zp[t][v] : 0.f;       // rows past n_rows hold whatever the allocation held (NaN x 0 = NaN)
                const bool act = fmaf(zv, ps[t], pt[t]) > 0.f && live;
                const float ga = act ? gv : 0.f;
                // the previous layer's consumers take dY with its ReLU mask applied (premasked)
                if (full) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ga), grsrc, glane + t * 128, rb, GAD_STREAM_STORE_AUX);
                else if (live) e.gout[(size_t)(slab * 32 + row) * 64 + t * 32 + l31] = ga;
                sb[t] += ga;
                sg[t] = fmaf(ga, (zv - pm[t]) * pi[t], sg[t]);
            }
        }
        r_cur = r_nxt;
        grp_cur = grp_nxt;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float s0 = sb[t] + __shfl_xor(sb[t], 32, 64);
        const float s1 = sg[t] + __shfl_xor(sg[t], 32, 64);
        if (lane < 32) { red[wave * 64 + t * 32 + lane] = s0; red[(8 + wave) * 64 + t * 32 + lane] = s1; }
    }
    __syncthreads();
    if (tid < 64) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { s0 += red[w * 64 + tid]; s1 += red[(8 + w) * 64 + tid]; }
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        atomic_add_f64(e.dbeta + (size_t)rep * e.stat_stride + tid, (double)s0);
        atomic_add_f64(e.dgamma + (size_t)rep * e.stat_stride + tid, (double)s1);
    }
}


// ------------------------------------------------------------------------------------------------
// Fused backward of an SA1 layer (rows ~ 2e5, 64 input channels, 64 / 128 output channels): dX AND dW in one pass over
// z / dY / z_prev.  The two streaming kernels above read the same 55 - 110 MB tensors from two streams at once and slow each
// other down far below what either reaches alone (layer 3: 67 + 83 us alone, 119 + 168 us side by side).  Here a workgroup
// of 8 wavefronts splits into 4 PRODUCERS and 4 CONSUMERS (one of each per SIMD, so the MFMA pipe of a SIMD alternates
// between them) and walks quads of four 32-row slabs:
//   producer p takes slab 4*quad + p exactly as gemm_dx_stream_kernel does (16-byte loads of z and dY, dZ in registers,
//     W^T fragments from LDS -> dY_prev, ReLU-masked, stored; the previous layer's BatchNorm-backward sums) and ALSO
//     writes its dZ values, row-major, into an LDS exchange buffer: the first half of the output channels, barrier, the
//     second half, barrier (two buffers: while one half is being written the consumers read the other);
//   consumer q owns one 32 x 32 tile of dW per channel half -- (channel tile, input half) -- in persistent accumulators
//     (2 x 16 VGPRs) and adds dZ^T . relu(bn(z_prev)) of the quad's slabs to it: dZ^T fragments are 4-byte LDS reads of
//     the exchange buffer (lane = channel: conflict-free), the activation fragments are rebuilt from z_prev (L1 / L2 hits:
//     the producers read the same rows).  n_out = 64 has one tile per half and input half, so two consumers share a tile
//     and take two slabs each.
// The producers' global operands come through a ring of 8 steps (8 channels each) filled 7 steps ahead, across slab
// boundaries (one producer per SIMD: nothing else hides the HBM latency; a scheduling barrier per step keeps the compiler
// from sinking the loads to their uses), their per-row metadata a whole slab ahead, their LDS operands one step ahead.
// No dW traffic leaves the workgroup before its end: each consumer then stores its tiles into the workgroup's partial
// block(s); dw_reduce sums the blocks (f64).  A first version that had every wavefront do both products on its own slab
// and add 32 x 32 tiles into a dW block in LDS (ds_add_f32) ran at 365 us for layer 3: LDS float atomics retire ~1 lane
// per clock, and re-reading dZ in the accumulator layout cost another 130 us.
// ------------------------------------------------------------------------------------------------
template <int NJ, int GM>
__global__ __launch_bounds__(512, 2) void gemm_bwd_stream_kernel(DzSrc d, const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                                  const float* __restrict__ W, DxEpi e, float* __restrict__ partial,
                                                                  unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int NO = 8 * NJ, PW = NO + 4, NCH = NJ / 4, HCH = NCH / 2;    // chunks of 32 channels; per half
    constexpr int HC = NO / 2, HP = HC + 4;                                 // channels per half, exchange-buffer pitch
    constexpr int NCT = HC / 32;                                            // dW channel tiles per half (2 or 1)
    __shared__ __attribute__((aligned(16))) float Wt[64 * PW];
    __shared__ __attribute__((aligned(16))) float vec[3 * NO];          // P | Q | S
    __shared__ float red[2 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float dzb[2 * 4 * 32 * HP];  // [half][slab of the quad][row][channel of the half]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    {   // W (n_out x 64, row-major) -> W^T in LDS
        constexpr int UW = NO * 16 / 512;
        float4 wr[UW];
#pragma unroll
        for (int it = 0; it < UW; ++it) wr[it] = ldg4(W + (size_t)(it * 512 + tid) * 4);
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid;
            const int n = u >> 4, k4 = (u & 15) * 4;
            Wt[(k4 + 0) * PW + n] = wr[it].x; Wt[(k4 + 1) * PW + n] = wr[it].y;
            Wt[(k4 + 2) * PW + n] = wr[it].z; Wt[(k4 + 3) * PW + n] = wr[it].w;
        }
    }
    for (int i = tid; i < NO; i += 512) {
        float P, Q, S;
        dz_coef(d, i, P, Q, S, first_workgroup());
        vec[i] = P; vec[NO + i] = Q; vec[2 * NO + i] = S;
    }
    __syncthreads();
    const int n_slabs = (n_rows + 31) >> 5;
    const int n_quads = (n_slabs + 3) >> 2;
    const __amdgpu_buffer_rsrc_t zrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.zprev), 0, gad_nbytes(n_rows, 64 * 4), 0x00020000);   // rows past n_rows read as 0
    const int glane = (4 * half * 64 + l31) * 4;

    if (wave < 4) {
        // ---------------------------------------------------------------- producer: dX of slab 4 * quad + wave
        float ps[2], pt[2], pm[2], pi[2];
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
            const int k = tk * 32 + l31;
            ps[tk] = e.ps[k]; pt[tk] = e.pt[k]; pm[tk] = e.pm[k]; pi[tk] = e.pi[k];
        }
        const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(e.gout, 0, gad_nbytes(n_rows, 64 * 4), 0x00020000);
        float sb[2] = {0.f, 0.f}, sg[2] = {0.f, 0.f};
        // operand ring: 8 steps of 8 output channels each (z, dY [, arg-max] of this lane's row: 16-byte loads), filled 7
        // steps ahead of their use and across slab boundaries -- with one producer per SIMD nothing else hides the HBM
        // latency of these loads (a two-chunk double buffer left ~2 us of stall per 32-channel chunk)
        constexpr int NU = NO / 8, RING = 8, AHEAD = RING - 1;
        float4 rz[RING], rg[RING];
        int4 ra[RING];
        auto row_of = [&](int sl) { return min(sl < n_slabs ? sl * 32 + l31 : 0, max(n_rows - 1, 0)); };
        auto load_u = [&](int r, int grp, int uu, int buf) {
            const unsigned n = 8 * uu + 4 * half;
            rz[buf] = ldg4(d.z + ((unsigned)r * NO + n));
            if (GM == 0) {
                rg[buf] = ldg4(d.G + ((unsigned)r * NO + n));
            } else {
                ra[buf] = *reinterpret_cast<const int4*>(d.argmax + ((unsigned)grp * NO + n));
                rg[buf] = ldg4(d.dout + ((unsigned)grp * NO + n));
            }
        };
        int quad = blockIdx.x;
        int slab = quad * 4 + wave;
        // row index / group / weight of this lane's row: fetched one slab (two for the group, which the ring's addresses
        // need mid-slab) before their use -- loads return in order, so waiting for a young load drains the whole ring
        int r_cur = row_of(slab), r_nxt = row_of(slab + 4 * gridDim.x);
        int grp_cur = GM == 1 ? d.row_grp[r_cur] : 0, grp_nxt = GM == 1 ? d.row_grp[r_nxt] : 0;
        float wrow = d.row_w ? d.row_w[r_cur] : 1.f;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) load_u(r_cur, grp_cur, i, i);
        float* const mybuf = dzb + (wave * 32 + l31) * HP + 4 * half;
        // LDS operands of a step (dZ coefficients of its 4 channels, W^T fragments): read one step ahead
        float4 P, Q, S, nb0, nb1;
        auto lds_operands = [&](int uu) {
            const int n = 8 * uu + 4 * half;
            P = *reinterpret_cast<const float4*>(vec + n);
            Q = *reinterpret_cast<const float4*>(vec + NO + n);
            S = *reinterpret_cast<const float4*>(vec + 2 * NO + n);
            nb0 = *reinterpret_cast<const float4*>(Wt + l31 * PW + n);
            nb1 = *reinterpret_cast<const float4*>(Wt + (32 + l31) * PW + n);
        };
        lds_operands(0);
        for (; quad < n_quads; quad += gridDim.x) {
            slab = quad * 4 + wave;
            const int r_nn = row_of(slab + 8 * gridDim.x);
            const int grp_nn = GM == 1 ? d.row_grp[r_nn] : 0;
            const float w_nxt = d.row_w ? d.row_w[r_nxt] : 1.f;
            const bool row_live = slab * 32 + l31 < n_rows;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
            float zp[2][16];
            const int zrow = (slab < n_slabs ? slab : 0) * 32 * 64 * 4;     // (a slab past the end: nothing is kept of it)
#pragma unroll
            for (int uu = 0; uu < NU; ++uu) {
                if (uu + AHEAD < NU) load_u(r_cur, grp_cur, uu + AHEAD, (uu + AHEAD) % RING);
                else load_u(r_nxt, grp_nxt, uu + AHEAD - NU, (uu + AHEAD) % RING);
                if (uu == 0) {                                             // z_prev for the epilogue: a whole slab ahead of its use
#pragma unroll
                    for (int v = 0; v < 16; ++v)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            zp[t][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                zrsrc, glane + t * 128, zrow + ((v & 3) + 8 * (v >> 2)) * 256, 0));
                }
                __builtin_amdgcn_sched_barrier(0);                         // (the scheduler would sink the loads to their uses)
                float* const hb = mybuf + (uu / (NU / 2)) * (4 * 32 * HP) + (uu % (NU / 2)) * 8;
                {
                    const float4 z = rz[uu % RING];
                    float4 g = rg[uu % RING];
                    if (GM == 1) {
                        const int4 a = ra[uu % RING];
                        const int rr = slab * 32 + l31;
                        g.x = a.x == rr ? g.x : 0.f; g.y = a.y == rr ? g.y : 0.f;
                        g.z = a.z == rr ? g.z : 0.f; g.w = a.w == rr ? g.w : 0.f;
                    }
                    float4 a4;
                    a4.x = P.x * g.x - wrow * fmaf(S.x, z.x, Q.x); a4.y = P.y * g.y - wrow * fmaf(S.y, z.y, Q.y);
                    a4.z = P.z * g.z - wrow * fmaf(S.z, z.z, Q.z); a4.w = P.w * g.w - wrow * fmaf(S.w, z.w, Q.w);
                    if (!row_live) a4 = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the end add nothing to dW
                    *reinterpret_cast<float4*>(hb) = a4;
                    const float4 b0 = nb0, b1 = nb1;
                    lds_operands((uu + 1) % NU);                           // the next step's, in flight under this step's MFMAs
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
                }
                if ((uu + 1) % (NU / 2) == 0) __syncthreads();             // this half of the quad's dZ is in the buffer
            }
            // epilogue: dY of the previous layer (ReLU-masked) + its BatchNorm-backward sums (the consumers are at work
            // on the second half meanwhile)
            const bool full = slab * 32 + 32 <= n_rows;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = acc_row(v, half);
                const bool live = slab * 32 + row < n_rows;
                const int rb = zrow + ((v & 3) + 8 * (v >> 2)) * 256;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float gv = acc[t][v];
                    const float zv = zp[t][v];
                    const bool act = fmaf(zv, ps[t], pt[t]) > 0.f && live;
                    const float ga = act ? gv : 0.f;
                    if (full) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ga), grsrc, glane + t * 128, rb, GAD_STREAM_STORE_AUX);
                    else if (live) e.gout[(size_t)(slab * 32 + row) * 64 + t * 32 + l31] = ga;
                    sb[t] += ga;
                    sg[t] = fmaf(ga, (zv - pm[t]) * pi[t], sg[t]);
                }
            }
            r_cur = r_nxt; r_nxt = r_nn;
            grp_cur = grp_nxt; grp_nxt = grp_nn;
            wrow = w_nxt;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float s0 = sb[t] + __shfl_xor(sb[t], 32, 64);
            const float s1 = sg[t] + __shfl_xor(sg[t], 32, 64);
            if (lane < 32) { red[wave * 64 + t * 32 + lane] = s0; red[(4 + wave) * 64 + t * 32 + lane] = s1; }
        }
    } else {
        // ---------------------------------------------------------------- consumer: dW tiles (channel tile, input half b)
        const int q = wave - 4;
        const int b = q & 1;
        const int tcl = NCT == 2 ? (q >> 1) : 0;                            // channel tile within the half
        const int s0 = NCT == 2 ? 0 : 2 * (q >> 1);                         // slabs of the quad this wavefront takes
        constexpr int NS = NCT == 2 ? 4 : 2;
        const float psb = e.ps[32 * b + l31], ptb = e.pt[32 * b + l31];
        f32x16 aw[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 16; ++v) aw[h][v] = 0.f;
        // relu(bn(z_prev)) of the quad's slabs in the accumulator layout, kept for both channel halves; the next quad's
        // rows are requested as soon as the second half has used a slab's values (a whole exchange phase ahead)
        float yq[NS][16];
        auto load_y = [&](int slab, int s) {
            const int zrow = (slab < n_slabs ? slab : 0) * 32 * 64 * 4;     // (past the end: the producers wrote dZ = 0)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                yq[s][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zrsrc, glane + b * 128, zrow + ((v & 3) + 8 * (v >> 2)) * 256, 0));
        };
        const float* const rbase = dzb + (4 * half) * HP + 32 * tcl + l31;
#pragma unroll
        for (int s = 0; s < NS; ++s) load_y(blockIdx.x * 4 + s0 + s, s);
        for (int quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();                                            // half h of this quad is in buffer h
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float* rb = rbase + (h * 4 + s0 + s) * (32 * HP);
                    float dz[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v) dz[v] = rb[((v & 3) + 8 * (v >> 2)) * HP];
                    if (h == 0) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) yq[s][v] = fmaxf(fmaf(yq[s][v], psb, ptb), 0.f);
                    }
#pragma unroll
                    for (int v = 0; v < 16; ++v) aw[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[v], yq[s][v], aw[h], 0, 0, 0);
                    if (h == 1) load_y((quad + gridDim.x) * 4 + s0 + s, s);
                }
            }
        }
        // the workgroup's partial dW block(s): n_out = 64 -> two consumers share a tile, each writes its own block
        float* pout = partial + (size_t)(NCT == 2 ? blockIdx.x : 2 * blockIdx.x + (q >> 1)) * NO * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 16; ++v) pout[(32 * (NCT * h + tcl) + acc_row(v, half)) * 64 + 32 * b + l31] = aw[h][v];
    }
    __syncthreads();
    if (tid < 64) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s0 += red[w * 64 + tid]; s1 += red[(4 + w) * 64 + tid]; }
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        atomic_add_f64(e.dbeta + (size_t)rep * e.stat_stride + tid, (double)s0);
        atomic_add_f64(e.dgamma + (size_t)rep * e.stat_stride + tid, (double)s1);
    }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 form of the SA1 streaming backward (option "mfma_split": GAD_SPLIT_BWD_STREAM): gemm_bwd_stream_kernel (DW: dX and dW
// of one layer from one pass, 4 producer + 4 consumer wavefronts) and gemm_dx_stream_kernel (!DW: eight producers, no exchange).
//   producer: a lane owns row l31 of its slab and, per k16-step s, the 8 output channels 16 s + 8 half .. + 7 (two 16-byte
//     loads of z and dY through the operand ring); dZ = P*dY - w*(Q + S*z) is split into hi / mid / lo in registers -- the A
//     operand of six v_mfma_f32_32x32x16_bf16 per 32-column tile of dX against the TRANSPOSED weight mirror (LDS: three planes
//     [k][n_out (+ 8 pad)] copied from gad_split_weights' output, whose odd 16-blocks of n are stored negated: odd steps
//     accumulate into a second accumulator pair).  DW: the same hi / mid / lo registers go to the consumers TRANSPOSED: the two
//     lanes of a row pair exchange halves (DPP quad_perm + v_perm_b32) so that each holds {row 2j, row 2j + 1} of four of the
//     eight channels, and store 4 bytes per channel and plane into the exchange buffer [half][plane][slab][channel][32 rows]
//     (64-byte rows, 16-byte chunks XOR-swizzled by (channel >> 2) & 3: conflict-free ds_write_b32 and ds_read_b128).
//   consumer: dW tile (32 channels x 32 inputs) += dZ^T . y over the quad's slabs: A fragments = three ds_read_b128 of 8
//     consecutive rows of its channel, B fragments = relu(bn(z_prev)) of rows 16 st + 8 half .. + 7 loaded straight from
//     global memory (one value per lane and row, as the f32 kernel does), split once per slab and kept for both channel
//     halves; the rows 16 .. 31 of every slab carry NEGATED y and accumulate into a second accumulator pair.
// Results, partial-block layout, barrier structure and the epilogue are those of the f32 kernels.
// ------------------------------------------------------------------------------------------------
template <int NJ, int GM, bool DW>
__global__ __launch_bounds__(512, 2) void gemm_bwd_stream_split_kernel(DzSrc d, const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                                        const uint16_t* __restrict__ wsp, int wsp_plane, DxEpi e,
                                                                        float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int NO = 8 * NJ, NS = NO / 16;                                // output channels, k16-steps
    constexpr int HC = NO / 2, NCT = HC / 32;                               // channels per exchange half, dW channel tiles per half
    constexpr int RB = 2 * NO + 16, WPL = 64 * RB;                          // bytes: mirror row in LDS (+ 16 pad), one plane
    constexpr int XSL = HC * 64, XPL = 4 * XSL, XBUF = 3 * XPL;             // bytes: exchange slab, plane (4 slabs), buffer (3 planes)
    __shared__ __attribute__((aligned(16))) unsigned char WtS[3 * WPL];
    __shared__ __attribute__((aligned(16))) float vec[3 * NO];              // P | Q | S
    __shared__ float red[2 * 8 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char dzb[DW ? 2 * XBUF : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    {   // transposed weight mirror (3 planes x 64 rows x NO bf16, row pitch NO) -> LDS rows of RB bytes
        constexpr int CPR = NO / 8, UNITS = 3 * 64 * CPR, UW = (UNITS + 511) / 512;      // 16-byte chunks
        gad_u32x4 wr[UW];
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid, uc = u < UNITS ? u : 0;
            const int pl = uc / (64 * CPR), rem = uc % (64 * CPR);
            wr[it] = *reinterpret_cast<const gad_u32x4*>(wsp + (size_t)pl * wsp_plane + (size_t)rem * 8);
        }
#pragma unroll
        for (int it = 0; it < UW; ++it) {
            const int u = it * 512 + tid;
            const int pl = u / (64 * CPR), rem = u % (64 * CPR), k = rem / CPR, c = rem % CPR;
            if (u < UNITS) *reinterpret_cast<gad_u32x4*>(WtS + pl * WPL + k * RB + c * 16) = wr[it];
        }
    }
    for (int i = tid; i < NO; i += 512) {
        float P, Q, S;
        dz_coef(d, i, P, Q, S, first_workgroup());
        vec[i] = P; vec[NO + i] = Q; vec[2 * NO + i] = S;
    }
    __syncthreads();
    const int n_slabs = (n_rows + 31) >> 5;
    const int n_quads = (n_slabs + 3) >> 2;
    const __amdgpu_buffer_rsrc_t zrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.zprev), 0, gad_nbytes(n_rows, 64 * 4), 0x00020000);   // rows past n_rows read as 0
    const int glane = (4 * half * 64 + l31) * 4;
    const int nprod = DW ? 4 : 8;                                           // producer wavefronts per workgroup

    if (!DW || wave < 4) {
        // ---------------------------------------------------------------- producer: dX of one slab per round
        float ps[2], pt[2], pm[2], pi[2];
#pragma unroll
        for (int tk = 0; tk < 2; ++tk) {
            const int k = tk * 32 + l31;
            ps[tk] = e.ps[k]; pt[tk] = e.pt[k]; pm[tk] = e.pm[k]; pi[tk] = e.pi[k];
        }
        const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(e.gout, 0, gad_nbytes(n_rows, 64 * 4), 0x00020000);
        float sb[2] = {0.f, 0.f}, sg[2] = {0.f, 0.f};
        // operand ring: entries of 4 channels (z, dY [, arg-max]: 16-byte loads); entry uu covers channels
        // 16 (uu >> 1) + 8 half + 4 (uu & 1) .. + 3, so a k16-step consumes an even / odd pair; filled AHEAD entries ahead, across slabs
        // The pooled source (GM = 1) reads its gradient / arg-max pair per GROUP (~26 consecutive rows share one: L1 / L2 hits), so that
        // pair rides a short ring of its own (RINGG = 4, one step ahead) -- with it in the long ring the kernel spilled.
        constexpr int NU = NO / 8, RING = 8, AHEAD = 6, RINGG = GM == 1 ? 4 : RING, AHEADG = GM == 1 ? 2 : AHEAD;
        static_assert(NU % RING == 0 && NU % RINGG == 0, "a slab's entries must fill the rings a whole number of times");
        float4 rz[RING], rg[RINGG];
        int4 ra[RINGG];
        auto row_of = [&](int sl) { return min(sl < n_slabs ? sl * 32 + l31 : 0, max(n_rows - 1, 0)); };
        auto load_z = [&](int r, int uu) {
            const unsigned n = 16 * (uu >> 1) + 8 * half + 4 * (uu & 1);
            rz[uu % RING] = ldg4(d.z + ((unsigned)r * NO + n));
            if (GM == 0) rg[uu % RINGG] = ldg4(d.G + ((unsigned)r * NO + n));
        };
        auto load_g = [&](int grp, int uu) {              // (GM = 1)
            const unsigned n = 16 * (uu >> 1) + 8 * half + 4 * (uu & 1);
            ra[uu % RINGG] = *reinterpret_cast<const int4*>(d.argmax + ((unsigned)grp * NO + n));
            rg[uu % RINGG] = ldg4(d.dout + ((unsigned)grp * NO + n));
        };
        // slabs: DW: quad q = blockIdx.x + i * gridDim.x, slab 4 q + wave; !DW: the wave-major dealing of gemm_dx_stream_kernel
        const int round_stride = DW ? 4 * gridDim.x : 8 * gridDim.x;
        int slab = DW ? blockIdx.x * 4 + wave : wave * gridDim.x + blockIdx.x;
        const int slab_limit = DW ? n_quads * 4 : n_slabs;
        int r_cur = row_of(slab), r_nxt = row_of(slab + round_stride);
        int grp_cur = GM == 1 ? d.row_grp[r_cur] : 0, grp_nxt = GM == 1 ? d.row_grp[r_nxt] : 0;
        float wrow = d.row_w ? d.row_w[r_cur] : 1.f;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) load_z(r_cur, i);
        if (GM == 1) {
#pragma unroll
            for (int i = 0; i < AHEADG; ++i) load_g(grp_cur, i);
        }
        // exchange store position of this lane: row pair l31 >> 1 (4 bytes) of channel (c0 & (HC - 1)) + 2 i + (l31 & 1)
        const unsigned psel = (l31 & 1) ? 0x03020706u : 0x05040100u;        // v_perm_b32 selector: {even row | odd row} of the lane's channels
        const unsigned rsg = (l31 & 1) ? 0x80000000u : 0u;                  // sign of this lane's row in the products (header of the split section)
        const int xq = l31 >> 3, xin = ((l31 >> 1) & 3) * 4;                // 16-byte chunk (before the swizzle), byte inside it
        const unsigned char* const wb = WtS + l31 * RB + 16 * half;
        for (; slab < slab_limit; slab += round_stride) {
            const int r_nn = row_of(slab + 2 * round_stride);
            const int grp_nn = GM == 1 ? d.row_grp[r_nn] : 0;
            const float w_nxt = d.row_w ? d.row_w[r_nxt] : 1.f;
            const bool row_live = slab * 32 + l31 < n_rows;
            f32x16 acc[2], an[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) { acc[t][v] = 0.f; an[t][v] = 0.f; }
            float zp[2][16];
            const int zrow = (slab < n_slabs ? slab : 0) * 32 * 64 * 4;     // (a slab past the end: nothing is kept of it)
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    const int uu = 2 * s2 + ee + AHEAD;
                    if (uu < NU) load_z(r_cur, uu); else load_z(r_nxt, uu - NU);
                    if (GM == 1) {
                        const int ug = 2 * s2 + ee + AHEADG;
                        if (ug < NU) load_g(grp_cur, ug); else load_g(grp_nxt, ug - NU);
                    }
                }
                if (s2 == 0) {                                              // z_prev for the epilogue: a whole slab ahead of its use
#pragma unroll
                    for (int v = 0; v < 16; ++v)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            zp[t][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                zrsrc, glane + t * 128, zrow + ((v & 3) + 8 * (v >> 2)) * 256, 0));
                }
                __builtin_amdgcn_sched_barrier(0);                          // (the scheduler would sink the loads to their uses)
                // B fragments of this step (transposed mirror rows l31, 32 + l31; channels 16 s + 8 half .. + 7)
                gad_u32x4 BH[2], BM[2], BL[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned char* q = wb + t * (32 * RB) + 32 * s2;
                    BH[t] = *reinterpret_cast<const gad_u32x4*>(q);
                    BM[t] = *reinterpret_cast<const gad_u32x4*>(q + WPL);
                    BL[t] = *reinterpret_cast<const gad_u32x4*>(q + 2 * WPL);
                }
                unsigned ah[4], am[4], al[4];
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    const int uu = 2 * s2 + ee;
                    const int n = 16 * s2 + 8 * half + 4 * ee;
                    const float4 P = *reinterpret_cast<const float4*>(vec + n);
                    const float4 Q = *reinterpret_cast<const float4*>(vec + NO + n);
                    const float4 S = *reinterpret_cast<const float4*>(vec + 2 * NO + n);
                    const float4 z = rz[uu % RING];
                    float4 g = rg[uu % RINGG];
                    if (GM == 1) {
                        const int4 a = ra[uu % RINGG];
                        const int rr = slab * 32 + l31;
                        g.x = a.x == rr ? g.x : 0.f; g.y = a.y == rr ? g.y : 0.f;
                        g.z = a.z == rr ? g.z : 0.f; g.w = a.w == rr ? g.w : 0.f;
                    }
                    float4 a4;
                    a4.x = P.x * g.x - wrow * fmaf(S.x, z.x, Q.x); a4.y = P.y * g.y - wrow * fmaf(S.y, z.y, Q.y);
                    a4.z = P.z * g.z - wrow * fmaf(S.z, z.z, Q.z); a4.w = P.w * g.w - wrow * fmaf(S.w, z.w, Q.w);
                    if (!row_live) a4 = make_float4(0.f, 0.f, 0.f, 0.f);       // rows past the end add nothing to dW
                    // (odd rows enter negated -- here and, through the exchange, in the consumers, which negate y of odd rows)
                    gad_split2(__uint_as_float(__float_as_uint(a4.x) ^ rsg), __uint_as_float(__float_as_uint(a4.y) ^ rsg), ah[2 * ee], am[2 * ee], al[2 * ee]);
                    gad_split2(__uint_as_float(__float_as_uint(a4.z) ^ rsg), __uint_as_float(__float_as_uint(a4.w) ^ rsg), ah[2 * ee + 1], am[2 * ee + 1], al[2 * ee + 1]);
                }
                const gad_u32x4 AH = {ah[0], ah[1], ah[2], ah[3]}, AM = {am[0], am[1], am[2], am[3]}, AL = {al[0], al[1], al[2], al[3]};
                if (DW) {
                    // transposed hand-over: channel cl = (16 s + 8 half) % HC + 2 i + (l31 & 1), rows {l31 & ~1, l31 | 1}
                    unsigned char* hb = dzb + (s2 / (NS / 2)) * XBUF + wave * XSL + xin;
                    const int cbase = (16 * s2 + 8 * half) % HC + (l31 & 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int cl = cbase + 2 * i;
                        unsigned char* q = hb + cl * 64 + ((xq ^ ((cl >> 2) & 3)) << 4);
                        const unsigned own[3] = {ah[i], am[i], al[i]};
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own[pl], 0xB1, 0xf, 0xf, false);   // lane ^ 1
                            *reinterpret_cast<unsigned*>(q + pl * XPL) = __builtin_amdgcn_perm(nbr, own[pl], psel);
                        }
                    }
                }
                // the six products of weight >= 2^-16, smallest first; odd steps (negated mirror values) -> the second pair
#define GAD_SPB(A, B)                                                                                                          \
                _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                \
                    if (s2 & 1) an[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[t]), an[t], 0, 0, 0);   \
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[t]), acc[t], 0, 0, 0); \
                }
                GAD_SPB(AL, BH) GAD_SPB(AH, BL) GAD_SPB(AM, BM) GAD_SPB(AM, BH) GAD_SPB(AH, BM) GAD_SPB(AH, BH)
#undef GAD_SPB
                if (DW && (s2 + 1) % (NS / 2) == 0) __syncthreads();        // this half of the quad's dZ is in the buffer
            }
            // epilogue: dY of the previous layer (ReLU-masked) + its BatchNorm-backward sums
            const bool full = slab * 32 + 32 <= n_rows;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = acc_row(v, half);
                const bool live = slab * 32 + row < n_rows;
                const int rb = zrow + ((v & 3) + 8 * (v >> 2)) * 256;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float gv = (v & 1) ? an[t][v] - acc[t][v] : acc[t][v] - an[t][v];
                    const float zv = live ? zp[t][v] : 0.f;
                    const bool act = fmaf(zv, ps[t], pt[t]) > 0.f && live;
                    const float ga = act ? gv : 0.f;
                    if (full) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ga), grsrc, glane + t * 128, rb, GAD_STREAM_STORE_AUX);
                    else if (live) e.gout[(size_t)(slab * 32 + row) * 64 + t * 32 + l31] = ga;
                    sb[t] += ga;
                    sg[t] = fmaf(ga, (zv - pm[t]) * pi[t], sg[t]);
                }
            }
            r_cur = r_nxt; r_nxt = r_nn;
            grp_cur = grp_nxt; grp_nxt = grp_nn;
            wrow = w_nxt;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float s0 = sb[t] + __shfl_xor(sb[t], 32, 64);
            const float s1 = sg[t] + __shfl_xor(sg[t], 32, 64);
            if (lane < 32) { red[wave * 64 + t * 32 + lane] = s0; red[(8 + wave) * 64 + t * 32 + lane] = s1; }
        }
    } else {
        // ---------------------------------------------------------------- consumer: dW tiles (channel tile, input half b)
        const int q = wave - 4;
        const int b = q & 1;
        const int tcl = NCT == 2 ? (q >> 1) : 0;                            // channel tile within the half
        const int s0 = NCT == 2 ? 0 : 2 * (q >> 1);                         // slabs of the quad this wavefront takes
        constexpr int NSC = NCT == 2 ? 4 : 2;
        const float psb = e.ps[32 * b + l31], ptb = e.pt[32 * b + l31];
        f32x16 aw[2], awn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 16; ++v) { aw[h][v] = 0.f; awn[h][v] = 0.f; }
        // z_prev of the quad's slabs: lane (input column 32 b + l31, half) holds rows 16 st + 8 half + i of a slab -- the B layout
        const int ylane = (8 * half * 64 + 32 * b + l31) * 4;
        float yr[NSC][16];
        gad_u32x4 YH[NSC][2], YM[NSC][2], YL[NSC][2];
        auto load_y = [&](int slab, int sl) {
            const int zrow = (slab < n_slabs ? slab : 0) * 32 * 64 * 4;     // (past the end: the producers wrote dZ = 0)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                yr[sl][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zrsrc, ylane, zrow + (16 * (v >> 3) + (v & 7)) * 256, 0));
        };
        const int cl = 32 * tcl + l31;
        const unsigned char* const rbase = dzb + cl * 64;
        const int fo = ((half ^ ((cl >> 2) & 3)) << 4);
#pragma unroll
        for (int sl = 0; sl < NSC; ++sl) load_y(blockIdx.x * 4 + s0 + sl, sl);
        for (int quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();                                            // half h of this quad is in buffer h
#pragma unroll
                for (int sl = 0; sl < NSC; ++sl) {
                    if (h == 0) {                                           // relu(bn(z_prev)) -> hi / mid / lo, rows 16 .. 31 negated
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            unsigned yh[4], ym[4], yl[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                float y0 = fmaxf(fmaf(yr[sl][8 * st + 2 * i], psb, ptb), 0.f), y1 = fmaxf(fmaf(yr[sl][8 * st + 2 * i + 1], psb, ptb), 0.f);
                                // dZ of the odd rows arrives negated: undo it on y; rows 16 .. 31 are negated for the accumulator pair
                                if (st) y0 = -y0; else y1 = -y1;
                                gad_split2(y0, y1, yh[i], ym[i], yl[i]);
                            }
                            YH[sl][st] = gad_u32x4{yh[0], yh[1], yh[2], yh[3]};
                            YM[sl][st] = gad_u32x4{ym[0], ym[1], ym[2], ym[3]};
                            YL[sl][st] = gad_u32x4{yl[0], yl[1], yl[2], yl[3]};
                        }
                    }
                    const unsigned char* rb = rbase + h * XBUF + (s0 + sl) * XSL;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int off = fo ^ (st << 5);
                        const gad_u32x4 AH = *reinterpret_cast<const gad_u32x4*>(rb + off);
                        const gad_u32x4 AM = *reinterpret_cast<const gad_u32x4*>(rb + XPL + off);
                        const gad_u32x4 AL = *reinterpret_cast<const gad_u32x4*>(rb + 2 * XPL + off);
#define GAD_SPC(A, B)                                                                                                          \
                        if (st) awn[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[sl][st]), awn[h], 0, 0, 0); \
                        else aw[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A), gad_as_bf16x8(B[sl][st]), aw[h], 0, 0, 0);
                        GAD_SPC(AL, YH) GAD_SPC(AH, YL) GAD_SPC(AM, YM) GAD_SPC(AM, YH) GAD_SPC(AH, YM) GAD_SPC(AH, YH)
#undef GAD_SPC
                    }
                    if (h == 1) load_y((quad + gridDim.x) * 4 + s0 + sl, sl);
                }
            }
        }
        // the workgroup's partial dW block(s): n_out = 64 -> two consumers share a tile, each writes its own block
        float* pout = partial + (size_t)(NCT == 2 ? blockIdx.x : 2 * blockIdx.x + (q >> 1)) * NO * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 16; ++v) pout[(32 * (NCT * h + tcl) + acc_row(v, half)) * 64 + 32 * b + l31] = aw[h][v] - awn[h][v];
    }
    __syncthreads();
    if (tid < 64) {
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < nprod; ++w) { s0 += red[w * 64 + tid]; s1 += red[(8 + w) * 64 + tid]; }
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        atomic_add_f64(e.dbeta + (size_t)rep * e.stat_stride + tid, (double)s0);
        atomic_add_f64(e.dgamma + (size_t)rep * e.stat_stride + tid, (double)s1);
    }
}

// ------------------------------------------------------------------------------------------------
// wide-tile dX for the MID-SIZE layers (the backward twin of gemm_fwd_wide_kernel): a workgroup owns 64 rows x 128 input
// channels of gout, a wavefront 32 x 64.  A operand: dZ[r][n] = P*dY - w*(Q + S*z) formed once per staged element from
// 16-byte loads of z and dY (or of the pooled arg-max / gradient pair; the ReLU mask is already in dY: premasked), stored
// row-major ([row][32 + 4]: one ds_read_b128 per four MFMA steps); B operand: W[n][k] kept AS STORED in LDS ([n][128 + 4], one
// ds_write_b128 per staged float4), its fragments are conflict-free ds_read_b32 across k of row n = 8j+4h+i.  (Round 3 wrote
// the W tile TRANSPOSED -- four ds_write_b32 per float4, 6 - 8 % bank-conflict cycles, 10 - 14 % LDS waits -- to read it back
// with ds_read_b128: round 4 measured the stored layout 18 % faster, 20 - 37 us -> 16 - 30 us per launch, +2.3 % steps/s.)
// Epilogue: dY of the previous layer masked by its ReLU (store_masked) + that layer's BatchNorm-backward sums.
// ------------------------------------------------------------------------------------------------
// SC = 1: scatter epilogue of the gathered first layers (SA2 / SA3: dX columns = the feature channels of the row's point):
// float atomics into dfeat[row_pt[r]][k], nothing stored per row, no BatchNorm in front
// SP: split-bf16 products (namespace spw): dZ is split while it is staged, the B side is the layer's TRANSPOSED weight mirror
// (rows = input channels k, the reduction index n contiguous: wsp, pitch = n_out).
template <int GM, int SC, bool SP = false>
__global__ __launch_bounds__(256, 2) void gemm_dx_wide_kernel(DzSrc d, const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                              const float* __restrict__ W, int Kp, int n_out, DxEpi e,
                                                              const uint16_t* __restrict__ wsp, int wsp_plane,
                                                              unsigned long long* __restrict__ ts) {
    KTimer kt_(ts);
    constexpr int BM = 64, BN = 128, P = KT + 4, PB = BN + 4, STAGE = SP ? spw::STAGE / 4 : BM * P + KT * PB, VM = 512;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 3 * VM + BM];
    __shared__ int32_t ptS[BM];                          // SC: the tile rows' points
    float* vP = smem + 2 * STAGE;                        // P | Q | S of this layer's channels
    float* wS = vP + 3 * VM;
    int32_t* grS = reinterpret_cast<int32_t*>(wS);       // (pooled source: the rows' groups share the slot with the weights: see below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int k0out = blockIdx.y * BN;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    if ((int)(blockIdx.x * BM) >= n_rows) return;
    const int nk = n_out / KT;
    // every global operand through a buffer descriptor: one 32-bit byte offset per lane and tile + a wave-uniform scalar offset
    // per access instead of a 64-bit pointer multiply-add per load / store (on this part every VALU instruction is MFMA issue
    // time: the epilogue's 64 address computations per tile were a third of the vector instructions of the N = 128 layers)
    const int gpitch = GM == 0 ? d.g_pitch : d.c;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.z), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GM == 0 ? d.G : d.dout), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(GM == 0 ? d.row_grp : d.argmax), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, GAD_BUF_MAX, 0x00020000);
    // previous layer's raw output / the gradient this launch writes: bounded by the live rows (reads past them give 0,
    // stores past them are dropped)
    const int zp_pitch4 = SC ? 0 : e.zprev_pitch * 4, go_pitch4 = SC ? 0 : e.gout_pitch * 4;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SC ? W : e.zprev), 0, SC ? 0 : gad_nbytes(n_rows, zp_pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(SC ? const_cast<float*>(W) : e.gout, 0, SC ? 0 : gad_nbytes(n_rows, go_pitch4), 0x00020000);
    const int c4 = (tid & 7) * 4, ur = tid >> 3;         // A staging: 16-byte chunk of the K-tile, row (ur, ur + 32)
    const int bk4 = (tid & 31) * 4, bn = tid >> 5;       // B staging: W rows bn, +8, +16, +24 of the K-tile, k chunk bk4
    int vw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vw[u] = ((bn + 8 * u) * Kp + k0out + bk4) * 4;
    // SP: rows k0out + (tid >> 2) + 64 u of the transposed mirror, 16-byte chunk tid & 3 of a K-tile's 32 output channels
    const __amdgpu_buffer_rsrc_t ws_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SP ? wsp : reinterpret_cast<const uint16_t*>(W)), 0, GAD_BUF_MAX, 0x00020000);
    int vbs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) vbs[u] = ((k0out + (tid >> 2) + 64 * u) * n_out + (tid & 3) * 8) * 2;
    unsigned char* const smem_b = reinterpret_cast<unsigned char*>(smem);
    const int sp_fo = ((half ^ ((l31 >> 2) & 3)) << 4);
    for (int i = tid; i < n_out; i += 256) {
        float Pc, Qc, Sc;
        dz_coef(d, i, Pc, Qc, Sc, first_workgroup());
        vP[i] = Pc; vP[VM + i] = Qc; vP[2 * VM + i] = Sc;
    }
    (void)grS;
    // previous layer's BatchNorm vectors of this lane's two output columns
    float ps[2] = {0.f, 0.f}, pt[2] = {0.f, 0.f}, pm[2] = {0.f, 0.f}, pi[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = k0out + wn * 64 + t * 32 + l31;
        if (!SC) { ps[t] = e.ps[k]; pt[t] = e.pt[k]; pm[t] = e.pm[k]; pi[t] = e.pi[k]; }
    }
    float cb[2] = {0.f, 0.f}, cg[2] = {0.f, 0.f};
    for (int row0 = blockIdx.x * BM; row0 < n_rows; row0 += gridDim.x * BM) {
        f32x16 acc[2];
        f32x16 an[2];                                    // SP: the negated accumulators (odd k16-steps)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[t][v] = 0.f; if (SP) an[t][v] = 0.f; }
        // this thread's two staged rows (clamped past the live count: their dZ is zeroed through the weight / validity)
        int vz[2], vg[2];
        float wrow[2];
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = row0 + ur + 32 * u;
            live[u] = r < n_rows;
            const int rr = live[u] ? r : max(n_rows - 1, 0);
            wrow[u] = d.row_w ? d.row_w[rr] : 1.f;
            vz[u] = (rr * d.z_pitch + c4) * 4;
            vg[u] = ((GM == 1 ? d.row_grp[rr] : rr) * gpitch + c4) * 4;
        }
        float4 rz[2], rg[2], rb[SP ? 1 : 4];
        gad_u32x4 ra[2];
        spw::BRegs rbs;
        auto load_regs = [&](int kt) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                rz[u] = buf_ld4(zr, vz[u], kt * (KT * 4));
                rg[u] = buf_ld4(gr_, vg[u], kt * (KT * 4));
                if (GM == 1) ra[u] = __builtin_amdgcn_raw_buffer_load_b128(ar, vg[u], kt * (KT * 4), 0);
            }
            if (SP) { spw::load_b(rbs, ws_, vbs, wsp_plane * 2, kt); return; }
#pragma unroll
            for (int u = 0; u < (SP ? 1 : 4); ++u) rb[u] = buf_ld4(wr, vw[u], kt * (KT * 4) * Kp);
        };
        auto write_lds = [&](int kt) {
            float* As = smem + (kt & 1) * STAGE;
            float* Bs = As + BM * P;
            const int nb = kt * KT + c4;
            const float4 P4 = *reinterpret_cast<const float4*>(vP + nb), Q4 = *reinterpret_cast<const float4*>(vP + VM + nb);
            const float4 S4 = *reinterpret_cast<const float4*>(vP + 2 * VM + nb);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 g = rg[u];
                const float4 z = rz[u];
                if (GM == 1) {
                    const unsigned r = row0 + ur + 32 * u;   // the true row (a clamped row never matches an arg-max)
                    g.x = ra[u].x == r ? g.x : 0.f; g.y = ra[u].y == r ? g.y : 0.f;
                    g.z = ra[u].z == r ? g.z : 0.f; g.w = ra[u].w == r ? g.w : 0.f;
                }
                const float w = wrow[u];
                float4 v;
                v.x = P4.x * g.x - w * fmaf(S4.x, z.x, Q4.x); v.y = P4.y * g.y - w * fmaf(S4.y, z.y, Q4.y);
                v.z = P4.z * g.z - w * fmaf(S4.z, z.z, Q4.z); v.w = P4.w * g.w - w * fmaf(S4.w, z.w, Q4.w);
                if (!live[u]) v = f4zero();
                if (SP) spw::store_a4(smem_b + (kt & 1) * spw::STAGE, ur + 32 * u, tid, v, (ur & 1) ? 0x80000000u : 0u);
                else *reinterpret_cast<float4*>(As + (ur + 32 * u) * P + c4) = v;
            }
            if (SP) { spw::store_b(smem_b + (kt & 1) * spw::STAGE, rbs, tid); return; }
#pragma unroll
            for (int u = 0; u < (SP ? 1 : 4); ++u)           // W[n][k..k+3] as stored: Bs[n][k], one ds_write_b128 (no transposing stores)
                *reinterpret_cast<float4*>(Bs + (bn + 8 * u) * PB + bk4) = rb[u];
        };
        load_regs(0);
        __syncthreads();                                 // vP visible; the previous row tile's LDS reads are done
        if (SC && tid < BM) ptS[tid] = e.row_pt[min(row0 + tid, max(n_rows - 1, 0))];     // (read after the K loop's barriers)
        write_lds(0);
        if (nk > 1) load_regs(1);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (SP) {
                spw::ktile(smem_b + (kt & 1) * spw::STAGE, (wm * 32 + l31) * 64, (wn * 64 + l31) * 64, sp_fo, acc, an);
                if (kt + 1 < nk) write_lds(kt + 1);
                if (kt + 2 < nk) load_regs(kt + 2);
                __syncthreads();
                continue;
            }
            const float* As = smem + (kt & 1) * STAGE + (wm * 32 + l31) * P + 4 * half;
            // B fragments: row n = 8 j + 4 h + i of the stored tile, this lane's column -- conflict-free ds_read_b32 across k
            const float* Bs = smem + (kt & 1) * STAGE + BM * P + (4 * half) * PB + wn * 64 + l31;
            float4 a4 = *reinterpret_cast<const float4*>(As);
            float4 b0 = make_float4(Bs[0], Bs[PB], Bs[2 * PB], Bs[3 * PB]);
            float4 b1 = make_float4(Bs[32], Bs[PB + 32], Bs[2 * PB + 32], Bs[3 * PB + 32]);
#pragma unroll
            for (int j = 0; j < KT / 8; ++j) {
                float4 an = a4, bn0 = b0, bn1 = b1;
                if (j + 1 < KT / 8) {
                    an = *reinterpret_cast<const float4*>(As + 8 * (j + 1));
                    const float* bq = Bs + 8 * (j + 1) * PB;
                    bn0 = make_float4(bq[0], bq[PB], bq[2 * PB], bq[3 * PB]);
                    bn1 = make_float4(bq[32], bq[PB + 32], bq[2 * PB + 32], bq[3 * PB + 32]);
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b1.w, acc[1], 0, 0, 0);
                a4 = an; b0 = bn0; b1 = bn1;
            }
            if (kt + 1 < nk) write_lds(kt + 1);
            if (kt + 2 < nk) load_regs(kt + 2);
            __syncthreads();
        }
        if (SP) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = (v & 1) ? an[t][v] - acc[t][v] : acc[t][v] - an[t][v];   // (odd rows entered negated)
        }
        // epilogue: dY of the previous layer (ReLU-masked) + its BatchNorm-backward sums; all z_prev loads first
        if (SC) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = k0out + wn * 64 + t * 32 + l31;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int il = wm * 32 + acc_row(v, half);
                    if (row0 + il < n_rows) atomic_add_f32(e.dfeat + (size_t)ptS[il] * e.feat_c + k, acc[t][v]);
                }
            }
            continue;
        }
        // (a row past the live count: z_prev reads 0 through the bounded descriptor, its accumulator row is exactly 0 -- its dZ
        // was zeroed -- and the store is dropped)
        const int rb0 = row0 + wm * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kcol = k0out + wn * 64 + t * 32 + l31;
            const int vzp = (4 * half * e.zprev_pitch + kcol) * 4, vgo = (4 * half * e.gout_pitch + kcol) * 4;
            float zp[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) zp[v] = buf_ld(pr, vzp, (rb0 + (v & 3) + 8 * (v >> 2)) * zp_pitch4);
            const float npm = -pm[t] * pi[t];
            float sb = 0.f, sg = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float ga = fmaf(zp[v], ps[t], pt[t]) > 0.f ? acc[t][v] : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ga), or_, vgo, (rb0 + (v & 3) + 8 * (v >> 2)) * go_pitch4, 0);
                sb += ga;
                sg = fmaf(ga, fmaf(zp[v], pi[t], npm), sg);
            }
            cb[t] += sb; cg[t] += sg;
        }
    }
    if (SC) return;
    const int rep = blockIdx.x % GAD_STAT_REPLICAS;
    block_column_atomics<2, 2, 2>(smem, cb, cg, lane, wm, wn, k0out, e.k_valid, e.dbeta + (size_t)rep * e.stat_stride,
                                  e.dgamma + (size_t)rep * e.stat_stride);
}

// ------------------------------------------------------------------------------------------------
// dX of the value encoder's SA1 first layer, of which only the ACTION columns are wanted (the gradient of Q(s, pi(s)) with
// respect to pi: reference core/ddpg.py:160-177): daction[sample][a] = sum over the sample's rows of dZ[r][:] . W[:][c0 + a].
// A 64-channel dot product per row and action component: HBM-bound (z and dY: 512 B per row), no MFMA.  16 lanes per row
// (one coalesced 256-byte row per quarter wavefront), a contiguous row range per wavefront, so the rows of a sample (~830,
// consecutive) are summed in registers and leave through six f64 atomics per sample and wavefront.  (The 64 x 64 tile
// kernel spent 63 us here forming all 16 input columns with MFMAs at 4 % of their peak.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dx_action_stream_kernel(DzSrc d, const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                               const float* __restrict__ W, int Kp, int c0, DxEpi e,
                                                               unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int n_waves = gridDim.x * 4;
    const int gw = blockIdx.x * 4 + (tid >> 6);
    int chunk = (n_rows + n_waves - 1) / n_waves;
    chunk = (chunk + 3) & ~3;
    const int r0 = gw * chunk, r1 = min(n_rows, r0 + chunk);
    if (r0 >= r1) return;
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
    float4 P, Q, S;
    dz_coef(d, c4 + 0, P.x, Q.x, S.x); dz_coef(d, c4 + 1, P.y, Q.y, S.y);
    dz_coef(d, c4 + 2, P.z, Q.z, S.z); dz_coef(d, c4 + 3, P.w, Q.w, S.w);
    float4 wa[6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
        wa[a] = make_float4(W[(size_t)(c4 + 0) * Kp + c0 + a], W[(size_t)(c4 + 1) * Kp + c0 + a], W[(size_t)(c4 + 2) * Kp + c0 + a],
                            W[(size_t)(c4 + 3) * Kp + c0 + a]);
    double run[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int cur = -1;                                                    // the open sample (wave-uniform)
    auto flush = [&]() {
        if (cur >= 0 && lane == 0)
#pragma unroll
            for (int a = 0; a < 6; ++a) atomic_add_f64(e.daction + (size_t)cur * 6 + a, run[a]);
#pragma unroll
        for (int a = 0; a < 6; ++a) run[a] = 0.0;
    };
    for (int rb = r0; rb < r1; rb += 4) {
        const int r = rb + sub;
        const bool live = r < r1;
        const int rr = live ? r : r1 - 1;
        const float4 z = ldg4(d.z + (size_t)rr * d.z_pitch + c4);
        const float4 g = ldg4(d.G + (size_t)rr * d.g_pitch + c4);
        const float w = d.row_w ? d.row_w[rr] : 1.f;
        const int smp = e.row_grp[rr] / e.gps;
        float4 dz;
        dz.x = P.x * g.x - w * fmaf(S.x, z.x, Q.x); dz.y = P.y * g.y - w * fmaf(S.y, z.y, Q.y);
        dz.z = P.z * g.z - w * fmaf(S.z, z.z, Q.z); dz.w = P.w * g.w - w * fmaf(S.w, z.w, Q.w);
        float p[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            p[a] = fmaf(dz.x, wa[a].x, fmaf(dz.y, wa[a].y, fmaf(dz.z, wa[a].z, dz.w * wa[a].w)));
            p[a] = live ? p[a] : 0.f;
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) p[a] += __shfl_xor(p[a], m, 16);        // the row's 16 lanes
        }
        const int s0 = __builtin_amdgcn_readlane(smp, 0), s1 = __builtin_amdgcn_readlane(smp, 16);
        const int s2 = __builtin_amdgcn_readlane(smp, 32), s3 = __builtin_amdgcn_readlane(smp, 48);
        if (s0 == s1 && s1 == s2 && s2 == s3) {                      // wave-uniform: the four rows are of one sample
            if (s0 != cur) { flush(); cur = s0; }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float t = p[a] + __shfl_xor(p[a], 16, 64);
                t += __shfl_xor(t, 32, 64);
                run[a] += (double)t;
            }
        } else {                                                     // a sample boundary inside the four rows: row by row
            flush();
            cur = -1;
            if ((lane & 15) == 0 && live)
#pragma unroll
                for (int a = 0; a < 6; ++a) atomic_add_f64(e.daction + (size_t)smp * 6 + a, (double)p[a]);
        }
    }
    flush();
}

// the action-gradient layer: scatter epilogue that wants nothing but daction, G source, 64 channels, 6 action components
static bool dx_action_streamable(const gad_gemm_dx_args& a, bool vec) {
    if (!g_opt_dx_stream || !vec || a.n_groups != 1 || a.dz_off[0] != 0 || a.w_off[0] != 0) return false;
    if (a.epilogue != 1 || a.dfeat || !a.daction || a.act_c != 6 || !a.row_grp || a.n_rows < 32768 || a.n_out[0] != 64) return false;
    const gad_dz_src& d = a.dz;
    return d.gmode == 0 && d.G && d.z && d.z_pitch % 4 == 0 && d.g_pitch % 4 == 0 && d.premasked && d.coefP && d.coefQ && d.coefS &&
           a.feat_c + 3 + a.act_c <= a.Kp;
}

static bool dx_wideable(const gad_gemm_dx_args& a, bool vec) {
    if (!g_opt_dx_wide || !vec || a.n_groups != 1 || a.dz_off[0] != 0 || a.w_off[0] != 0 || a.gout_off[0] != 0) return false;
    if (a.n_rows < 2048 || a.k_valid % 128 != 0 || a.k_valid > a.Kp) return false;
    if (a.n_out[0] % 32 != 0 || a.n_out[0] < 32 || a.n_out[0] > 512) return false;
    if (a.epilogue == 1) {                               // scatter into the points' feature gradients (SA2 / SA3 first layers)
        if (g_opt_dx_wide == 2 || !a.dfeat || a.daction || !a.row_pt || a.k_valid != a.feat_c || a.prev_dbeta) return false;
    } else if (!a.prev_dbeta || !a.store_masked || !(a.zprev && a.prev_scale && a.prev_shift && a.prev_mean && a.prev_istd && a.prev_dgamma)) return false;
    const gad_dz_src& d = a.dz;
    if (!d.z || d.z_pitch % 4 != 0 || !d.relu || !d.premasked || !(d.coefP && d.coefQ && d.coefS)) return false;
    if (!fits_i32_bytes(a.n_rows, d.z_pitch, d.gmode == 0 ? d.g_pitch : d.c, a.epilogue == 0 ? a.gout_pitch : 0, a.epilogue == 0 ? a.zprev_pitch : 0)) return false;
    return d.gmode == 0 ? (d.g_pitch % 4 == 0 && d.G) : (d.c % 4 == 0);
}

static bool dx_streamable(const gad_gemm_dx_args& a, bool vec) {
    if (!g_opt_dx_stream || !vec || a.n_groups != 1 || a.dz_off[0] != 0 || a.w_off[0] != 0 || a.gout_off[0] != 0) return false;
    if (a.n_rows < 32768 || a.epilogue != 0 || a.k_valid != 64 || a.Kp != 64 || a.gout_pitch != 64) return false;
    if (a.n_out[0] != 64 && a.n_out[0] != 128) return false;
    if (!a.prev_dbeta || a.zprev_pitch != 64 || !a.store_masked) return false;
    const gad_dz_src& d = a.dz;
    const bool coef = (d.coefP && d.coefQ && d.coefS);
    if (!d.z || d.z_pitch != a.n_out[0] || !d.scale || !d.relu || !d.premasked || !coef) return false;
    if (d.gmode == 0 ? d.g_pitch != a.n_out[0] : d.c != a.n_out[0]) return false;
    return (long long)a.n_rows * 128 * 4 <= (1ll << 31);
}

// skinny dX for the small-M layers (same idea as gemm_fwd_skinny_kernel): one 32x32 tile of gout per workgroup, the 8
// wavefronts split the reduction over the layer's output channels n.  A operand = dZ[r][8j+4h..+3] (16-byte loads of
// z and G, BatchNorm-backward applied in registers), B operand = W[8j+4h+i][k0+lane%32]: four 4-byte loads per group
// of 8 channels, each a fully coalesced 128-byte row segment -- no transposed weight copy needed.
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_dx_skinny_kernel(DzSrc d, Groups gr, int n_rows,
                                                                    const float* __restrict__ W, int Kp, DxEpi e, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    __shared__ __attribute__((aligned(16))) float stZ[NW * 32 * SK_SP], stG[NW * 32 * SK_SP];
    float* const part = stZ;                                  // partial tiles reuse the wavefront's own operand stage
    __shared__ __attribute__((aligned(16))) float vec[5 * VMAX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], n_out = gr.nout[g], goff = gr.ooff[g];
    const int k0 = blockIdx.x * 32, row0 = blockIdx.y * 32;      // (x = column tile: see gemm_fwd_skinny_kernel)
    for (int i = tid; i < n_out; i += 64 * NW) {
        vec[i] = d.scale ? d.scale[doff + i] : 1.f;
        vec[VMAX + i] = d.shift ? d.shift[doff + i] : 0.f;
        float P, Q, S;
        dz_coef(d, doff + i, P, Q, S, first_workgroup());
        vec[2 * VMAX + i] = P; vec[3 * VMAX + i] = Q; vec[4 * VMAX + i] = S;
    }
    __syncthreads();
    const int r = min(row0 + l31, max(n_rows - 1, 0));                // clamped: extra rows are not stored
    const int k = min(k0 + l31, Kp - 1);
    const float* Wg = W + gr.woff[g] + k;
    const float wrow = d.row_w ? d.row_w[r] : 1.f;
    const int nj = (n_out + 7) >> 3;
    const int per = (nj + NW - 1) / NW;
    const int j0 = wave * per, j1 = min(nj, j0 + per);

    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // z and dY blocks of four channel groups go through a wavefront-private LDS stage, loaded coalesced (see
    // gemm_fwd_skinny_kernel); the W fragments are coalesced as they are.  Two blocks per loop trip: static register sets.
    float* const myZ = stZ + wave * (32 * SK_SP);
    float* const myG = stG + wave * (32 * SK_SP);
    const int rsub = lane >> 3, chunk = lane & 7;
    const float* pz[4];
    const float* pg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t rr = (size_t)min(row0 + rsub + 8 * i, max(n_rows - 1, 0));
        pz[i] = d.z ? d.z + rr * d.z_pitch + doff + 4 * chunk : nullptr;
        pg[i] = d.G + rr * d.g_pitch + doff + 4 * chunk;
    }
    auto staged = [&](int jb) { return d.z != nullptr && jb + 4 <= j1 && 8 * (jb + 4) <= n_out; };     // wave-uniform
    float4 gz[4], gg[4], rb[2][4];
    auto load_blk = [&](int jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { gz[i] = ldg4(pz[i] + 8 * jb); gg[i] = ldg4(pg[i] + 8 * jb); }
    };
    auto load_w = [&](int jb, auto bufc) {
        constexpr int B = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = 8 * min(jb + u, nj - 1) + 4 * half;
            const int nc = n < n_out ? n : 0;                 // clamped; dz_finish zeroes n >= n_out
            const float* wp = Wg + (size_t)nc * Kp;
            rb[B][u] = make_float4(wp[0], wp[Kp], wp[2 * (size_t)Kp], wp[3 * (size_t)Kp]);
        }
    };
    auto block = [&](int jb, auto bufc) {
        constexpr int B = decltype(bufc)::value;
        const bool more = jb + 4 < j1;
        float4 rz[4], rg[4];
        const bool st = staged(jb);
        if (st) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(myZ + (rsub + 8 * i) * SK_SP + 4 * chunk) = gz[i];
                *reinterpret_cast<float4*>(myG + (rsub + 8 * i) * SK_SP + 4 * chunk) = gg[i];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = 8 * min(jb + u, nj - 1) + 4 * half;
                const int nc = n < n_out ? n : 0;
                rz[u] = d.z ? ldg4(d.z + (size_t)r * d.z_pitch + doff + nc) : f4zero();
                rg[u] = ldg4(d.G + (size_t)r * d.g_pitch + doff + nc);
            }
        }
        if (more) {
            if (staged(jb + 4)) load_blk(jb + 4);
            load_w(jb + 4, std::integral_constant<int, B ^ 1>{});
        }
        if (st) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rz[u] = *reinterpret_cast<const float4*>(myZ + l31 * SK_SP + 8 * u + 4 * half);
                rg[u] = *reinterpret_cast<const float4*>(myG + l31 * SK_SP + 8 * u + 4 * half);
            }
        }
        const int je = min(jb + 4, j1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (jb + u < je) {                                // wave-uniform
                DzRaw raw;
                raw.z = rz[u]; raw.g = rg[u]; raw.a = make_int4(0, 0, 0, 0);
                const float4 a4 = dz_finish<true>(d, raw, r, true, 8 * (jb + u) + 4 * half, n_out, wrow, vec);
                const float4 b4 = rb[B][u];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        }
        if (st) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (j0 < j1) {
        if (staged(j0)) load_blk(j0);
        load_w(j0, std::integral_constant<int, 0>{});
    }
    for (int jb = j0; jb < j1; jb += 8) {
        block(jb, std::integral_constant<int, 0>{});
        if (jb + 4 < j1) block(jb + 4, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) part[wave * (32 * SK_SP) + v * 64 + lane] = acc[v];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] += part[w * (32 * SK_SP) + v * 64 + lane];
    const int kk = k0 + l31;
    const bool kok = kk < e.k_valid;
    const bool stats = e.dbeta != nullptr && kok;
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (stats) { sc = e.ps[goff + kk]; sh = e.pt[goff + kk]; mu = e.pm[goff + kk]; is = e.pi[goff + kk]; }
    float sb = 0.f, sg = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int rr = row0 + acc_row(v, half);
        if (rr >= n_rows || !kok) continue;
        const float gv = acc[v];
        float outv = gv;
        if (stats) {
            const float zp = e.zprev[(size_t)rr * e.zprev_pitch + goff + kk];
            if (fmaf(zp, sc, sh) > 0.f) { sb += gv; sg = fmaf(gv, (zp - mu) * is, sg); }
            else if (e.store_masked) outv = 0.f;
        }
        e.gout[(size_t)rr * e.gout_pitch + goff + kk] = outv;
    }
    if (e.dbeta) {
        sb += __shfl_xor(sb, 32, 64);
        sg += __shfl_xor(sg, 32, 64);
        if (lane < 32 && kok) {
            const int rep = blockIdx.y % GAD_STAT_REPLICAS;
            atomic_add_f64(e.dbeta + (size_t)rep * e.stat_stride + goff + kk, (double)sb);
            atomic_add_f64(e.dgamma + (size_t)rep * e.stat_stride + goff + kk, (double)sg);
        }
    }
}

// A route whose kernels read P / Q / S per lane (no cooperative prologue) cannot form the BatchNorm-backward coefficients
// itself: gad_bn_bwd_coef runs first, into the caller's coefP / Q / S scratch, and the block is dropped.
static int coef_fallback(gad_dz_src& dz, void* stream) {
    if (!dz.bn_dbeta) return GAD_OK;
    GAD_REQUIRE(dz.coefP && dz.coefQ && dz.coefS && dz.scale && dz.bn_dgamma && dz.bn_mean && dz.bn_istd, GAD_ERR_NULL,
                "BatchNorm-backward block on a route without a prologue needs coefP / Q / S scratch");
    if (int e = gad_bn_bwd_coef(dz.bn_dbeta, dz.bn_dgamma, dz.bn_stride, dz.scale, dz.bn_mean, dz.bn_istd, dz.c, dz.bn_count,
                                const_cast<float*>(dz.coefP), const_cast<float*>(dz.coefQ), const_cast<float*>(dz.coefS), dz.gacc_gamma,
                                dz.gacc_beta, stream)) return e;
    dz.bn_dbeta = nullptr; dz.bn_dgamma = nullptr; dz.gacc_gamma = nullptr; dz.gacc_beta = nullptr;
    return GAD_OK;
}

extern "C" int gad_gemm_dx(const gad_gemm_dx_args* a, void* stream) {
    unsigned long long* ts = gad_take_timing_slot(stream);
    const int rows_hint = gad_take_grid_rows();
    GAD_REQUIRE(a && a->W, GAD_ERR_NULL, "gemm_dx: null pointer");
    GAD_REQUIRE(a->n_groups >= 1 && a->n_groups <= GAD_MAX_GROUPS, GAD_ERR_SHAPE, "gemm_dx: n_groups");
    GAD_REQUIRE(a->Kp % 8 == 0, GAD_ERR_SHAPE, "gemm_dx: Kp must be a multiple of 8");
    GAD_REQUIRE(a->epilogue == 1 || a->gout, GAD_ERR_NULL, "gemm_dx: gout");
    GAD_REQUIRE(a->dz.gmode == 0 ? a->dz.G != nullptr : (a->dz.argmax && a->dz.dout && a->dz.row_grp), GAD_ERR_NULL,
                "gemm_dx: gradient source");
    if (a->n_rows <= 0) return GAD_OK;
    DzSrc d = make_dzsrc(a->dz);
    Groups gr = make_groups(a->n_groups, a->dz_off, a->w_off, a->gout_off, a->n_out);
    DxEpi e;
    e.mode = a->epilogue; e.gout = a->gout; e.gout_pitch = a->gout_pitch; e.k_valid = a->k_valid;
    e.zprev = a->zprev; e.zprev_pitch = a->zprev_pitch; e.ps = a->prev_scale; e.pt = a->prev_shift;
    e.pm = a->prev_mean; e.pi = a->prev_istd; e.dbeta = a->prev_dbeta; e.dgamma = a->prev_dgamma;
    e.stat_stride = a->stat_stride;
    e.store_masked = (a->store_masked && a->prev_dbeta) ? 1 : 0;
    e.dfeat = a->dfeat; e.feat_c = a->feat_c; e.row_pt = a->row_pt; e.row_grp = a->row_grp;
    e.daction = a->daction; e.act_c = a->act_c; e.gps = a->grp_per_sample > 0 ? a->grp_per_sample : 1;
    GAD_REQUIRE(e.mode == 0 || (e.row_pt && e.row_grp), GAD_ERR_NULL, "gemm_dx: scatter epilogue needs row maps");
    GAD_REQUIRE(!e.dbeta || (e.zprev && e.ps && e.pt && e.pm && e.pi && e.dgamma), GAD_ERR_NULL, "gemm_dx: prev BN stats inputs");
    hipStream_t st = (hipStream_t)stream;
    const int rows = a->n_rows, kv = a->k_valid;
    const int grid_rows = (a->n_rows_dev && rows_hint > 0 && rows_hint < rows) ? rows_hint : rows;   // gad_grid_rows_hint
    const bool vec = dz_vectorizable(a->dz, a->dz_off, a->n_out, a->n_groups);
    if (dx_streamable(*a, vec)) {
        const int slabs = gad_cdiv(rows, 32);
        int gx = gad_cdiv(slabs, 8); if (gx > 256) gx = 256;
        if (split_on(GAD_SPLIT_BWD_STREAM) && a->W_split_t && a->W_split_t_pitch == a->n_out[0] && a->W_split_t_plane >= 64 * a->n_out[0]) {
#define LAUNCH_DXSS(NJ, GM) hipLaunchKernelGGL((gemm_bwd_stream_split_kernel<NJ, GM, false>), dim3(gx), dim3(512), 0, st, d, a->n_rows_dev, rows, \
                                                a->W_split_t, a->W_split_t_plane, e, nullptr, ts)
            if (a->n_out[0] == 128) { if (a->dz.gmode == 0) LAUNCH_DXSS(16, 0); else LAUNCH_DXSS(16, 1); }
            else { if (a->dz.gmode == 0) LAUNCH_DXSS(8, 0); else LAUNCH_DXSS(8, 1); }
#undef LAUNCH_DXSS
            GAD_CHECK_LAUNCH("gemm_dx(stream split)");
            return GAD_OK;
        }
#define LAUNCH_DXS(NJ, GM) hipLaunchKernelGGL((gemm_dx_stream_kernel<NJ, GM>), dim3(gx), dim3(512), 0, st, d, a->n_rows_dev, rows, a->W, e, ts)
        if (a->n_out[0] == 128) { if (a->dz.gmode == 0) LAUNCH_DXS(16, 0); else LAUNCH_DXS(16, 1); }
        else { if (a->dz.gmode == 0) LAUNCH_DXS(8, 0); else LAUNCH_DXS(8, 1); }
#undef LAUNCH_DXS
        GAD_CHECK_LAUNCH("gemm_dx(stream)");
        return GAD_OK;
    }
    if (dx_action_streamable(*a, vec)) {
        if (a->dz.bn_dbeta) {                            // per-lane coefficients: formed by gad_bn_bwd_coef first
            gad_dz_src dz2 = a->dz;
            if (int e2 = coef_fallback(dz2, stream)) return e2;
            d = make_dzsrc(dz2);
        }
        hipLaunchKernelGGL(dx_action_stream_kernel, dim3(1024), dim3(256), 0, st, d, a->n_rows_dev, rows, a->W, a->Kp, a->feat_c + 3, e, ts);
        GAD_CHECK_LAUNCH("gemm_dx(action stream)");
        return GAD_OK;
    }
    if (dx_wideable(*a, vec)) {
        int gx = gad_cdiv(grid_rows, 64); if (gx > GAD_GX_CAP) gx = GAD_GX_CAP;
        const dim3 grid(gx, kv / 128);
        // split-bf16 form: family bit set, the call carries the transposed weight mirror, the reduction (n_out) is whole K-tiles
        const bool sp = split_on(GAD_SPLIT_DX_WIDE) && a->W_split_t && a->W_split_t_pitch == a->n_out[0] && a->n_out[0] % 32 == 0 &&
                        a->W_split_t_plane >= kv * a->n_out[0];
#define LAUNCH_DXW(GM, SC, SP)                                                                                                 \
        hipLaunchKernelGGL((gemm_dx_wide_kernel<GM, SC, SP>), grid, dim3(256), 0, st, d, a->n_rows_dev, rows, a->W, a->Kp, a->n_out[0], e, \
                           a->W_split_t, a->W_split_t_plane, ts)
        if (a->epilogue == 1 && a->dz.gmode == 0) { if (sp) LAUNCH_DXW(0, 1, true); else LAUNCH_DXW(0, 1, false); }
        else if (a->dz.gmode == 0) { if (sp) LAUNCH_DXW(0, 0, true); else LAUNCH_DXW(0, 0, false); }
        else { if (sp) LAUNCH_DXW(1, 0, true); else LAUNCH_DXW(1, 0, false); }
#undef LAUNCH_DXW
        if (sp) GAD_CHECK_LAUNCH("gemm_dx(wide split)"); else GAD_CHECK_LAUNCH("gemm_dx(wide)");
        return GAD_OK;
    }
    int nmax_dx = 0;
    for (int i = 0; i < a->n_groups; ++i) nmax_dx = a->n_out[i] > nmax_dx ? a->n_out[i] : nmax_dx;
    if (g_opt_dx_skinny && vec && e.mode == 0 && a->dz.gmode == 0 && !a->n_rows_dev && rows <= 1024 &&
        nmax_dx <= VMAX) {
        if (g_opt_skinny_nw == 4)
            hipLaunchKernelGGL(gemm_dx_skinny_kernel<4>, dim3(gad_cdiv(kv, 32), gad_cdiv(rows, 32), gr.n), dim3(64 * 4), 0, st, d, gr,
                               rows, a->W, a->Kp, e, ts);
        else
            hipLaunchKernelGGL(gemm_dx_skinny_kernel<SK_NW>, dim3(gad_cdiv(kv, 32), gad_cdiv(rows, 32), gr.n), dim3(64 * SK_NW), 0, st, d, gr,
                               rows, a->W, a->Kp, e, ts);
        GAD_CHECK_LAUNCH("gemm_dx(skinny)");
        return GAD_OK;
    }
#define LAUNCH_DX2(WM, WN, TM, TN, V)                                                                    \
    do {                                                                                                 \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                              \
        int gx = gad_cdiv(grid_rows, BM); if (gx > GAD_GX_CAP) gx = GAD_GX_CAP;                                      \
        if (nmax_dx <= 512)                                                                              \
            hipLaunchKernelGGL((gemm_dx_kernel<WM, WN, TM, TN, V, 512>), dim3(gx, gad_cdiv(kv, BN), gr.n), dim3(256), 0, \
                               st, d, gr, a->n_rows_dev, rows, a->W, a->Kp, e, ts);                          \
        else                                                                                             \
            hipLaunchKernelGGL((gemm_dx_kernel<WM, WN, TM, TN, V, VMAX>), dim3(gx, gad_cdiv(kv, BN), gr.n), dim3(256), 0, \
                               st, d, gr, a->n_rows_dev, rows, a->W, a->Kp, e, ts);                          \
    } while (0)
#define LAUNCH_DX(WM, WN, TM, TN) do { if (vec) LAUNCH_DX2(WM, WN, TM, TN, true); else LAUNCH_DX2(WM, WN, TM, TN, false); } while (0)
    // 64 x 64 tiles (or 128 x 32 for narrow outputs): measured best on the whole step, also for the 2e5-row SA1
    // layers (128 x 64: -3.7 % steps/s, 128 x 128: -16 %)
    if (kv <= 32) LAUNCH_DX(4, 1, 1, 1); else LAUNCH_DX(2, 2, 1, 1);
#undef LAUNCH_DX
#undef LAUNCH_DX2
    GAD_CHECK_LAUNCH("gemm_dx");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// backward wrt the weights:  gacc[n][k] += sum_r dZ[r][n] * X[r][k]
// split over rows: every (tile, split) block writes its partial tile to the caller's workspace, a
// second kernel sums the splits in f64.  Without a workspace: f64 atomics.
// ------------------------------------------------------------------------------------------------
// VM: stride of the per-channel vectors in LDS (>= the widest layer side).  512 instead of VMAX = 1024 brings the
// workgroup from 46 KB to 32 KB of LDS: four instead of three resident workgroups per CU (the register budget allows four).
template <int WM, int WN, int TM, int TN, int XM, bool VEC, int VM>
__global__ __launch_bounds__(256) void gemm_dw_kernel(DzSrc d, XSrc x, Groups gr,
                                                       const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                       int Kp, int k_used, int n_ktiles, double* __restrict__ gacc,
                                                       float* __restrict__ partial, long long group_stride, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchD<BM>::v, PB = PitchD<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    constexpr int TILE = KT * PA + KT * PB;
    constexpr int SM = TILE + (VEC ? 5 * VM : 4) + 2 * VM;
    __shared__ __attribute__((aligned(16))) float smem[SM];
    float* As = smem;
    float* Bs = smem + KT * PA;
    float* vec = smem + TILE;
    float* sv = vec + (VEC ? 5 * VM : 4);
    float* tv = sv + VM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], zoff = gr.ooff[g], n_out = gr.nout[g];
    const int tile_n = blockIdx.x / n_ktiles, tile_k = blockIdx.x % n_ktiles;
    const int n0 = tile_n * BM, k0 = tile_k * BN;
    if (n0 >= n_out) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    int chunk = gad_cdiv_dev(n_rows, (int)gridDim.y);
    chunk = (chunk + KT - 1) / KT * KT;
    const int r_begin = blockIdx.y * chunk;
    const int r_end = min(r_begin + chunk, n_rows);
    if (r_begin >= r_end) return;     // the reducer skips the same splits (same chunk arithmetic)
    if (VEC) stage_dz_vecs<VM>(vec, d, doff, n_out);
    if (XM == 0 && x.affine) stage_affine<256>(sv, tv, x, zoff, x.c_in);
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    DzRaw ra[UA];
    float wa[UA];
    XRaw rb[UB];
    const int bulk = XM == 0 ? x.c_in : x.feat_c;
    const bool tail = k0 + BN > bulk;
    // row -> group / point indices are needed to ADDRESS the pooled-gradient and gather loads: fetched one K-tile
    // ahead of the data they address, so a tile's load phase is one memory latency, not two chained ones
    int gq[UA], pq[UB], xq[UB];
    auto load_idx = [&](int rb0) {
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int kk, i; unit_D<BM>(it * 256 + tid, kk, i);
            const int r = rb0 + kk;
            gq[it] = d.gmode != 0 ? d.row_grp[r < r_end ? r : 0] : 0;
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
            const int r = rb0 + kk;
            const int rr = r < r_end ? r : 0;
            pq[it] = XM == 1 ? x.row_pt[rr] : 0;
            xq[it] = (XM == 1 && tail) ? x.row_grp[rr] : -1;
        }
    };
    auto load_tile = [&](int rb0) {
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int kk, i; unit_D<BM>(it * 256 + tid, kk, i);
            const int r = rb0 + kk;
            const bool ok = r < r_end;
            const int rr = ok ? r : 0;
            wa[it] = d.row_w ? d.row_w[rr] : 1.f;
            ra[it] = dz_raw<VEC>(d, r, ok, doff, n0 + i, n_out, gq[it]);
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
            const int r = rb0 + kk;
            const bool ok = r < r_end && (k0 + j < Kp);
            rb[it] = x_raw<XM>(x, r, ok, zoff, k0 + j, tail, pq[it], xq[it]);
        }
        load_idx(rb0 + KT);
    };
    auto store_tile = [&](int rb0) {
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int kk, i; unit_D<BM>(it * 256 + tid, kk, i);
            const int r = rb0 + kk;
            store_D<BM>(As, kk, i, dz_finish<VEC, VM>(d, ra[it], r, r < r_end, n0 + i, n_out, wa[it], vec));
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
            const int r = rb0 + kk;
            const bool ok = r < r_end && (k0 + j < Kp);
            store_D<BN>(Bs, kk, j, x_finish<XM>(x, rb[it], ok, k0 + j, sv, tv));
        }
    };
    load_idx(r_begin);
    load_tile(r_begin);
    for (int rb0 = r_begin; rb0 < r_end; rb0 += KT) {
        store_tile(rb0);
        __syncthreads();
        if (rb0 + KT < r_end) load_tile(rb0 + KT);
        mfma_ktile<TM, TN, PA, PB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
        __syncthreads();
    }
    const int l31 = lane & 31, half = lane >> 5;
    float* pout = partial ? partial + (size_t)g * group_stride + (size_t)blockIdx.y * n_out * Kp : nullptr;
    double* aout = gacc + gr.woff[g];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int k = k0 + wn * TN * 32 + tn * 32 + l31;
        if (k >= k_used) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = n0 + wm * TM * 32 + tm * 32 + acc_row(v, half);
                if (n >= n_out) continue;
                if (pout) pout[(size_t)n * Kp + k] = acc[tm][tn][v];
                else atomic_add_f64(aout + (size_t)n * Kp + k, (double)acc[tm][tn][v]);
            }
    }
}

#define DW_RED_CHUNK 16
// all_active: every split wrote its partial tile (the streaming kernels deal row units to the splits round-robin); else the
// splits are contiguous row chunks and those past the live rows were not written
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ partial, long long group_stride,
                                                        Groups gr, const int32_t* __restrict__ n_rows_dev,
                                                        int n_rows_static, int splits, int Kp, int k_used,
                                                        double* __restrict__ gacc, int all_active = 0) {
    const int g = blockIdx.z;
    const int n_out = gr.nout[g];
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)n_out * Kp) return;
    const int k = (int)(e % Kp);
    if (k >= k_used) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    int chunk = (n_rows + splits - 1) / splits;
    chunk = (chunk + KT - 1) / KT * KT;
    const int active = all_active ? splits : (chunk > 0 ? (n_rows + chunk - 1) / chunk : 0);
    const int s0 = blockIdx.y * DW_RED_CHUNK, s1 = min(s0 + DW_RED_CHUNK, active);
    if (s0 >= s1) return;
    const float* p = partial + (size_t)g * group_stride + e;
    const size_t stride = (size_t)n_out * Kp;
    double s = 0.0;
#pragma unroll 4
    for (int sidx = s0; sidx < s1; ++sidx) s += (double)p[(size_t)sidx * stride];
    atomic_add_f64(gacc + gr.woff[g] + e, s);
}

// skinny dW for the small-M layers (FC head, actor / critic heads: <= 1024 rows to reduce over, outputs up to 1024 x 1032).
// One 32x32 tile of dW per workgroup, the 8 wavefronts split the ROWS.  Both MFMA operands are indexed [row][channel]
// in memory with the reduction index (rows) leading, so with the k = 8j+4h+i visiting order lane (channel = lane%32)
// needs four consecutive rows of its channel: four coalesced 4-byte loads each for z, dY (-> dZ in registers, per-lane
// BatchNorm constants) and for the layer input (-> act(scale*z+shift), or the bias / extra column).  No LDS staging,
// all loads of a wavefront's rows in flight before its first MFMA; partial tiles summed through LDS, then f64 atomics
// straight into the gradient arena (no split-K workspace, no reduce launch).
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_dw_skinny_kernel(DzSrc d, XSrc x, Groups gr, int n_rows, int Kp,
                                                                    int k_used, double* __restrict__ gacc, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    __shared__ float part[NW * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], zoff = gr.ooff[g], n_out = gr.nout[g];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    if (n0 >= n_out) return;
    // A side: this lane's output channel n and its BatchNorm-backward constants
    const int n = n0 + l31;
    const bool n_ok = n < n_out;
    const int ch = doff + (n_ok ? n : 0);
    const float dsc = (d.scale && !d.premasked) ? d.scale[ch] : 1.f, dsh = (d.scale && !d.premasked) ? d.shift[ch] : 0.f;
    float dP, dQ, dS;
    dz_coef(d, ch, dP, dQ, dS);
    // B side: this lane's input column k: 0 = activation column, 1 = extra column, 2 = bias (ones) column, 3 = padding
    const int k = k0 + l31;
    const int kind = k < x.c_in ? 0 : ((k == x.c_in && x.extra) ? 1 : (k == x.ones_col ? 2 : 3));
    const int kc = zoff + (kind == 0 ? k : 0);
    const float xs = (kind == 0 && x.scale) ? x.scale[kc] : 1.f, xt = (kind == 0 && x.shift) ? x.shift[kc] : 0.f;

    const int units = (n_rows + 7) >> 3;
    const int per = (units + NW - 1) / NW;
    const int u0 = wave * per, u1 = min(units, u0 + per);
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    constexpr int CHU = 4;                                   // 8-row units per register chunk
    for (int c0 = u0; c0 < u1; c0 += CHU) {
        float rz[CHU][4], rg[CHU][4], rx[CHU][4], rw[CHU][4];
#pragma unroll
        for (int u = 0; u < CHU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * (c0 + u) + 4 * half + i;
                const int rr = min(r, max(n_rows - 1, 0));
                rz[u][i] = d.z ? d.z[(size_t)rr * d.z_pitch + ch] : 0.f;
                rg[u][i] = d.G[(size_t)rr * d.g_pitch + ch];
                rw[u][i] = d.row_w ? d.row_w[rr] : 1.f;
                rx[u][i] = kind == 0 ? x.zin[(size_t)rr * x.zin_pitch + kc] : (kind == 1 ? x.extra[rr] : (kind == 2 ? 1.f : 0.f));
            }
#pragma unroll
        for (int u = 0; u < CHU; ++u) {
            if (c0 + u >= u1) break;                         // wave-uniform
            float a4[4], b4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * (c0 + u) + 4 * half + i;
                const bool live = r < n_rows;
                const float z = rz[u][i];
                float gq = rg[u][i];
                if (d.relu && !d.premasked) gq = fmaf(z, dsc, dsh) > 0.f ? gq : 0.f;
                if (d.coef) gq = dP * gq - rw[u][i] * fmaf(dS, z, dQ);
                a4[i] = (live && n_ok) ? gq : 0.f;
                float xv = rx[u][i];
                if (kind == 0) { xv = fmaf(xv, xs, xt); if (x.relu) xv = fmaxf(xv, 0.f); }
                b4[i] = live ? xv : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b4[i], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) part[(wave * 16 + v) * 64 + lane] = acc[v];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] += part[(w * 16 + v) * 64 + lane];
    if (k >= k_used) return;
    double* aout = gacc + gr.woff[g];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int nn = n0 + acc_row(v, half);
        if (nn < n_out) atomic_add_f64(aout + (size_t)nn * Kp + k, (double)acc[v]);
    }
}

// Streaming dW for the SA1 layers (>= 32k de-duplicated rows, 64 input channels, 64 or 128 output channels).
// dW[n][k] = sum_r dZ[r][n] * X[r][k]: the reduction index is the leading index of both operands in memory, so a lane that
// owns output channel n (A side) / input channel k (B side) feeds the MFMA straight from coalesced 4-byte global loads --
// no LDS tiles, no barriers in the loop, BatchNorm constants per lane in registers (as in the skinny kernel above).
// Work split: a workgroup of 8 wavefronts owns one split of the rows (the dw_reduce geometry: blockIdx.x = split);
// for 128 output channels wavefronts 0-3 take output channels 0-63 and 4-7 take 64-127, each set dealing the split's
// 8-row units among its four members.  Every wavefront keeps a 64 x 64 block of dW in 64 accumulator registers
// (2 x 2 MFMA tiles: each loaded operand is used twice), loads unit u+1 while computing unit u, and the four partial
// blocks are summed through LDS into the split's partial tile.
template <int NG, int GM>
__global__ __launch_bounds__(512) void gemm_dw_stream_kernel(DzSrc d, XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                             int n_rows_static, int splits, float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int NO = 64 * NG, KP = 64, RW = 8 / NG;              // RW: wavefronts that share the rows of one column set
    __shared__ float red[8 * 16 * 64];                             // one accumulator tile of every wavefront
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int cset = wave / RW, wr = wave % RW;                    // column set, rank among the wavefronts of that set
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    // 8-row units are dealt to the workgroups ROUND-ROBIN (unit = (k * gridDim.x + blockIdx.x) * RW + rank): at any time
    // the whole grid reads one contiguous window of rows and that window sweeps the arrays front to back -- the order in
    // which the dX kernel of the same layer, running beside this one on the main stream, walks them too, so the second
    // reader of z / dY finds the lines in the L2 / Infinity Cache instead of HBM (contiguous chunks per workgroup touched
    // all of the array at once).  Every workgroup writes its partial tile (zeros without rows): dw_reduce all_active.
    const int r_begin = 0, r_end = n_rows;
    (void)splits;

    // per-lane constants: A side (dZ) output channels n = 64 cset + 32 j + l31, B side (X) input channels k = 32 b + l31
    float dP[2], dQ[2], dS[2], xs[2], xt[2];          // (the ReLU mask is already in dY: premasked)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = 64 * cset + 32 * j + l31;
        dz_coef(d, ch, dP[j], dQ[j], dS[j]);
        xs[j] = x.scale[32 * j + l31]; xt[j] = x.shift[32 * j + l31];
    }
    // Addressing of the main loop: raw buffer loads whose per-lane byte offset is a CONSTANT (channel block + the
    // half-wave's 4-row stagger) and whose row position lives in the scalar offset -- no vector ALU work per load.
    const unsigned lane_n = 64u * cset + l31;
    const int zpitch4 = d.z_pitch * 4, xpitch4 = x.zin_pitch * 4, gpitch4 = (GM == 0 ? d.g_pitch : d.c) * 4;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.z), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GM == 0 ? d.G : d.dout), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(GM == 0 ? d.row_grp : d.argmax), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x.zin), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.row_w ? d.row_w : d.z), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(GM == 0 ? d.argmax : d.row_grp), 0, GAD_BUF_MAX, 0x00020000);
    const int vl_z = 4 * half * zpitch4 + (int)lane_n * 4, vl_x = 4 * half * xpitch4 + l31 * 4;
    const int vl_g = (GM == 0 ? 4 * half * gpitch4 : 0) + (int)lane_n * 4, vl_r = 4 * half * 4;

    struct Unit { float z[4][2], g[4][2], xv[4][2], w[4]; int a[4][2]; };
    auto ldf = [](__amdgpu_buffer_rsrc_t r, int vo, int so) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); };
    auto load = [&](Unit& u, int unit) {                           // a unit with all eight rows inside the split
        const int rs = r_begin + 8 * unit;                         // scalar
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u.w[i] = d.row_w ? ldf(wr_, vl_r, (rs + i) * 4) : 1.f;
            int vg = vl_g;
            if (GM == 1) vg = (int)__builtin_amdgcn_raw_buffer_load_b32(pr, vl_r, (rs + i) * 4, 0) * gpitch4 + vl_g;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u.z[i][j] = ldf(zr, vl_z + 128 * j, (rs + i) * zpitch4);
                u.g[i][j] = ldf(gr_, vg + 128 * j, GM == 0 ? (rs + i) * gpitch4 : 0);
                if (GM == 1) u.a[i][j] = (int)__builtin_amdgcn_raw_buffer_load_b32(ar, vg + 128 * j, 0, 0);
                u.xv[i][j] = ldf(xr, vl_x + 128 * j, (rs + i) * xpitch4);
            }
        }
    };
    auto load_tail = [&](Unit& u, int unit) {                      // the ragged last unit: clamped rows, plain loads
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(r_begin + 8 * unit + 4 * half + i, r_end - 1);
            u.w[i] = d.row_w ? d.row_w[r] : 1.f;
            const size_t go = (GM == 0 ? (size_t)r * d.g_pitch : (size_t)d.row_grp[r] * d.c) + lane_n;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u.z[i][j] = d.z[(size_t)r * d.z_pitch + lane_n + 32 * j];
                u.g[i][j] = (GM == 0 ? d.G : d.dout)[go + 32 * j];
                if (GM == 1) u.a[i][j] = d.argmax[go + 32 * j];
                u.xv[i][j] = x.zin[(size_t)r * x.zin_pitch + l31 + 32 * j];
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][b][v] = 0.f;
    auto compute = [&](const Unit& u, int unit, auto tail) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r_begin + 8 * unit + 4 * half + i;
            const bool live = !decltype(tail)::value || r < r_end;
            float a2[2], b2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float z = u.z[i][j];
                float gq = u.g[i][j];
                if (GM == 1) gq = u.a[i][j] == r ? gq : 0.f;
                gq = dP[j] * gq - u.w[i] * fmaf(dS[j], z, dQ[j]);
                a2[j] = live ? gq : 0.f;
                const float xv = fmaxf(fmaf(u.xv[i][j], xs[j], xt[j]), 0.f);
                b2[j] = live ? xv : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], b2[b], acc[j][b], 0, 0, 0);
        }
    };
    const std::integral_constant<bool, false> full;
    const std::integral_constant<bool, true> part;
    const int nfull = (r_end - r_begin) >> 3;                      // units with all eight rows live
    const int ustep = gridDim.x * RW;
    Unit u0, u1;
    int un = blockIdx.x * RW + wr;
    if (un < nfull) load(u0, un);
    while (un < nfull) {
        if (un + ustep < nfull) load(u1, un + ustep);
        compute(u0, un, full);
        un += ustep;
        if (un >= nfull) break;
        if (un + ustep < nfull) load(u0, un + ustep);
        compute(u1, un, full);
        un += ustep;
    }
    if (((r_end - r_begin) & 7) != 0 && (nfull / RW) % gridDim.x == blockIdx.x && wr == nfull % RW) {   // the ragged last unit
        load_tail(u0, nfull);
        compute(u0, nfull, part);
    }
    // the set's RW partial 64 x 64 blocks -> one: tile (j, b) of every wavefront through LDS, summed by wavefront (j, b) of the set
    float* pout = partial + (size_t)blockIdx.x * NO * KP;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = t >> 1, b = t & 1;
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(wave * 16 + v) * 64 + lane] = acc[j][b][v];
        __syncthreads();
        if (wr == t % RW) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < RW; ++w) sum += red[((cset * RW + w) * 16 + v) * 64 + lane];
                const int n = 64 * cset + 32 * j + acc_row(v, half), k = 32 * b + l31;
                pout[(size_t)n * KP + k] = sum;
            }
        }
        __syncthreads();
    }
}

// Streaming dW for SA1 layer 1, whose input rows are GATHERED: [feat[pt] | src_xyz[pt] - ctr_xyz[grp] | action[sample] | 0]
// (<= 32 columns, no BatchNorm in front).  Same scheme as above with one B tile: lane c of the B side produces input column
// c of its rows from per-lane constants (which array, which stride, which component), the A side is the adjacent-pair
// mapping (accumulator row i of tile j = output channel 2 i + j, one 8-byte load per row for z and for dY).
__global__ __launch_bounds__(512) void gemm_dw_gather_stream_kernel(DzSrc d, XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                                    int n_rows_static, int splits, int Kp, int k_used,
                                                                    float inv_gps, float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt(ts);
    constexpr int RW = 8;
    typedef unsigned gad_u32x2 __attribute__((ext_vector_type(2)));
    __shared__ float red[8 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int r_begin = 0, r_end = n_rows;         // units dealt round-robin over the grid (see gemm_dw_stream_kernel)
    (void)splits;

    const unsigned lane_n = 2u * l31;
    float dP[2], dQ[2], dS[2];                        // (premasked dY)
#pragma unroll
    for (int j = 0; j < 2; ++j) dz_coef(d, lane_n + j, dP[j], dQ[j], dS[j]);
    // B side: what input column k = l31 is made of
    const int k = l31, fc = x.feat_c;
    const int kind = k < fc ? 0 : (k < fc + 3 ? 1 : ((x.action && k < fc + 3 + x.act_c) ? 2 : 3));
    const float* base1 = kind == 0 ? x.feat : (kind == 1 ? x.src_xyz : (kind == 2 ? x.action : x.feat));
    const int stride1 = kind == 0 ? fc : (kind == 1 ? 3 : (kind == 2 ? x.act_c : 0));
    const int off1 = kind == 0 ? k : (kind == 1 ? k - fc : (kind == 2 ? k - fc - 3 : 0));
    const bool sub_ctr = kind == 1 && x.ctr_xyz != nullptr;
    const float* base2 = sub_ctr ? x.ctr_xyz + off1 : x.src_xyz;   // lanes without a centre read a valid address, unused

    const int zpitch4 = d.z_pitch * 4, gpitch4 = d.g_pitch * 4;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.z), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.G), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.row_w ? d.row_w : d.z), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t ptr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(x.row_pt), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t grpr = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(x.row_grp), 0, GAD_BUF_MAX, 0x00020000);
    const int vl_z = 4 * half * zpitch4 + (int)lane_n * 4, vl_g = 4 * half * gpitch4 + (int)lane_n * 4, vl_r = 4 * half * 4;

    struct Unit { float z[4][2], g[4][2], w[4], x1[4], x2[4]; };
    auto gather = [&](Unit& u, int i, int pt, int grp) {
        int idx = pt;
        if (kind == 2) {                                           // sample = grp / gps (gps need not be a power of two)
            int q = (int)((float)grp * inv_gps);
            const int rem = grp - q * x.gps;
            q += rem >= x.gps ? 1 : (rem < 0 ? -1 : 0);
            idx = q;
        }
        u.x1[i] = base1[(size_t)idx * stride1 + off1];
        u.x2[i] = base2[sub_ctr ? (size_t)grp * 3 : 0];
    };
    auto load = [&](Unit& u, int unit) {
        const int rs = r_begin + 8 * unit;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pt = (int)__builtin_amdgcn_raw_buffer_load_b32(ptr_, vl_r, (rs + i) * 4, 0);
            const int grp = (int)__builtin_amdgcn_raw_buffer_load_b32(grpr, vl_r, (rs + i) * 4, 0);
            u.w[i] = d.row_w ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr_, vl_r, (rs + i) * 4, 0)) : 1.f;
            const gad_u32x2 zz = __builtin_amdgcn_raw_buffer_load_b64(zr, vl_z, (rs + i) * zpitch4, 0);
            const gad_u32x2 gg = __builtin_amdgcn_raw_buffer_load_b64(gr_, vl_g, (rs + i) * gpitch4, 0);
            u.z[i][0] = __uint_as_float(zz.x); u.z[i][1] = __uint_as_float(zz.y);
            u.g[i][0] = __uint_as_float(gg.x); u.g[i][1] = __uint_as_float(gg.y);
            gather(u, i, pt, grp);
        }
    };
    auto load_tail = [&](Unit& u, int unit) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(r_begin + 8 * unit + 4 * half + i, r_end - 1);
            u.w[i] = d.row_w ? d.row_w[r] : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u.z[i][j] = d.z[(size_t)r * d.z_pitch + lane_n + j];
                u.g[i][j] = d.G[(size_t)r * d.g_pitch + lane_n + j];
            }
            gather(u, i, x.row_pt[r], x.row_grp[r]);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    auto compute = [&](const Unit& u, int unit, auto tail) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r_begin + 8 * unit + 4 * half + i;
            const bool live = !decltype(tail)::value || r < r_end;
            float xv = sub_ctr ? __fsub_rn(u.x1[i], u.x2[i]) : u.x1[i];
            xv = (kind != 3 && live) ? xv : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float z = u.z[i][j];
                const float gq = dP[j] * u.g[i][j] - u.w[i] * fmaf(dS[j], z, dQ[j]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? gq : 0.f, xv, acc[j], 0, 0, 0);
            }
        }
    };
    const std::integral_constant<bool, false> full;
    const std::integral_constant<bool, true> part;
    const int nfull = (r_end - r_begin) >> 3;
    const int ustep = gridDim.x * RW;
    Unit u0, u1;
    int un = blockIdx.x * RW + wave;
    if (un < nfull) load(u0, un);
    while (un < nfull) {
        if (un + ustep < nfull) load(u1, un + ustep);
        compute(u0, un, full);
        un += ustep;
        if (un >= nfull) break;
        if (un + ustep < nfull) load(u0, un + ustep);
        compute(u1, un, full);
        un += ustep;
    }
    if (((r_end - r_begin) & 7) != 0 && (nfull / RW) % gridDim.x == blockIdx.x && wave == nfull % RW) {
        load_tail(u0, nfull);
        compute(u0, nfull, part);
    }
    float* pout = partial + (size_t)blockIdx.x * 64 * Kp;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(wave * 16 + v) * 64 + lane] = acc[j][v];
        __syncthreads();
        if ((wave >> 2) == j) {                                    // four wavefronts per tile, four accumulator rows each
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const int v = 4 * (wave & 3) + vv;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < RW; ++w) sum += red[(w * 16 + v) * 64 + lane];
                const int n = 2 * acc_row(v, half) + j;
                if (l31 < Kp) pout[(size_t)n * Kp + l31] = l31 < k_used ? sum : 0.f;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// wide-tile dW for the MID-SIZE layers: dW[n][k] = sum_r dZ[r][n] * X[r][k] is a tiny output (128 x 128 ... 512 x 256)
// behind a long reduction (8e3 - 3e4 rows).  The 64 x 64 kernel re-reads every row of dZ and X once per output tile (4 - 32
// times through L2: 112 MB of traffic for 42 MB of operands) and keeps one accumulator per wavefront.  Here a workgroup owns
// a 128 x 128 block of dW for its chunk of rows -- the whole matrix for the 128-wide layers -- a wavefront 64 x 64 (four
// accumulators: each staged operand value feeds two MFMAs), K-tile = 32 rows, LDS double-buffered with one barrier per
// K-tile, operands stored as loaded ([row][128 + 4]: one ds_write_b128 per staged float4, fragments are conflict-free
// ds_read_b32 of consecutive channels).  dZ = P*dY - w*(Q + S*z) and X = relu(scale*z_in + shift) are formed once per
// staged element.  Partial tiles go to the caller's workspace, dw_reduce sums the row chunks in f64.
// XM = 1 (first layer of SA2 / SA3: gathered input [feat[pt] | src_xyz[pt] - ctr_xyz[grp]]): the X side is feat[pt] as
// stored (row -> point indices fetched one K-tile ahead of the rows they address); the three coordinate columns of dW are
// sums of dZ * dx accumulated with vector FMAs while dZ is staged (12 per float4) by the workgroups of the first k block.
// ------------------------------------------------------------------------------------------------
template <int XM, int GM>
__global__ __launch_bounds__(256, 2) void gemm_dw_wide_kernel(DzSrc d, XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                              int n_rows_static, int Kp, int n_out, int tiles_k,
                                                              float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt_(ts);
    constexpr int BT = 128, P = BT + 4, STAGE = 2 * KT * P, VM = 512;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 5 * VM];
    float* vP = smem + 2 * STAGE;                        // P | Q | S of the dZ channels, scale | shift of the input channels
    float* sv = vP + 3 * VM;
    float* tv = sv + VM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = (blockIdx.x / tiles_k) * BT, k0 = (blockIdx.x % tiles_k) * BT;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    int chunk = gad_cdiv_dev(n_rows, (int)gridDim.y);
    chunk = (chunk + KT - 1) / KT * KT;                  // == dw_reduce_kernel's split geometry
    const int r_begin = blockIdx.y * chunk, r_end = min(r_begin + chunk, n_rows);
    if (r_begin >= r_end) return;                        // the reducer skips the same splits
    for (int i = tid; i < BT; i += 256) {
        float Pc, Qc, Sc;
        dz_coef(d, n0 + i, Pc, Qc, Sc, blockIdx.y == 0 && k0 == 0);
        vP[i] = Pc; vP[VM + i] = Qc; vP[2 * VM + i] = Sc;
        if (XM == 0) { sv[i] = x.scale[k0 + i]; tv[i] = x.shift[k0 + i]; }
    }
    const bool coords = XM == 1 && k0 == 0;              // this workgroup also forms dW[n][feat_c .. feat_c + 2]
    float wx[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wx[j][0] = 0.f; wx[j][1] = 0.f; wx[j][2] = 0.f; }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    // staging: a K-tile is 32 rows x 128 channels per side = 1024 float4: thread -> rows sr + 8 u (u < 4), channels c4 .. c4 + 3
    const int c4 = (tid & 31) * 4, sr = tid >> 5;
    const int gpitch = GM == 0 ? d.g_pitch : d.c;
    float4 rz[4], rg[4], rx[4];
    int4 ra[4];
    float rw[4], rd[4][3];
    int pq[4], gq[4];                                    // XM = 1: point / group of the NEXT tile's rows
    int gd[4];                                           // GM = 1: group of the NEXT tile's rows (pooled gradient source)
    auto load_idx = [&](int rb0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + sr + 8 * u;
            const int rr = r < r_end ? r : r_end - 1;
            pq[u] = XM == 1 ? x.row_pt[rr] : 0;
            gq[u] = (XM == 1 && x.ctr_xyz) ? x.row_grp[rr] : 0;
            gd[u] = GM == 1 ? d.row_grp[rr] : 0;
        }
    };
    auto load_regs = [&](int rb0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + sr + 8 * u;
            const int rr = r < r_end ? r : r_end - 1;
            rw[u] = d.row_w ? d.row_w[rr] : 1.f;
            rz[u] = ldg4(d.z + (size_t)rr * d.z_pitch + n0 + c4);
            if (GM == 0) {
                rg[u] = ldg4(d.G + (size_t)rr * gpitch + n0 + c4);
            } else {
                const int grp = gd[u];                   // (fetched one K-tile ahead: one memory round trip here, not two)
                ra[u] = *reinterpret_cast<const int4*>(d.argmax + (size_t)grp * gpitch + n0 + c4);
                rg[u] = ldg4(d.dout + (size_t)grp * gpitch + n0 + c4);
            }
            if (XM == 0) {
                rx[u] = ldg4(x.zin + (size_t)rr * x.zin_pitch + k0 + c4);
            } else {
                rx[u] = ldg4(x.feat + (size_t)pq[u] * x.feat_c + k0 + c4);
                if (coords) {
                    const float* p = x.src_xyz + (size_t)pq[u] * 3;
                    float q0 = p[0], q1 = p[1], q2 = p[2];
                    if (x.ctr_xyz) {
                        const float* cp = x.ctr_xyz + (size_t)gq[u] * 3;
                        q0 = __fsub_rn(q0, cp[0]); q1 = __fsub_rn(q1, cp[1]); q2 = __fsub_rn(q2, cp[2]);
                    }
                    rd[u][0] = q0; rd[u][1] = q1; rd[u][2] = q2;
                }
            }
        }
        if (XM == 1 || GM == 1) load_idx(rb0 + KT);
    };
    auto write_lds = [&](int it, int rb0) {
        float* As = smem + (it & 1) * STAGE;
        float* Bs = As + KT * P;
        const float4 Pv = *reinterpret_cast<const float4*>(vP + c4), Qv = *reinterpret_cast<const float4*>(vP + VM + c4);
        const float4 Sv = *reinterpret_cast<const float4*>(vP + 2 * VM + c4);
        float4 s4 = f4zero(), t4 = f4zero();
        if (XM == 0) { s4 = *reinterpret_cast<const float4*>(sv + c4); t4 = *reinterpret_cast<const float4*>(tv + c4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + sr + 8 * u;
            float4 g = rg[u];
            const float4 z = rz[u];
            if (GM == 1) {
                g.x = ra[u].x == r ? g.x : 0.f; g.y = ra[u].y == r ? g.y : 0.f;
                g.z = ra[u].z == r ? g.z : 0.f; g.w = ra[u].w == r ? g.w : 0.f;
            }
            const float w = rw[u];
            float4 a, b;
            a.x = Pv.x * g.x - w * fmaf(Sv.x, z.x, Qv.x); a.y = Pv.y * g.y - w * fmaf(Sv.y, z.y, Qv.y);
            a.z = Pv.z * g.z - w * fmaf(Sv.z, z.z, Qv.z); a.w = Pv.w * g.w - w * fmaf(Sv.w, z.w, Qv.w);
            if (XM == 0) {
                b.x = fmaxf(fmaf(rx[u].x, s4.x, t4.x), 0.f); b.y = fmaxf(fmaf(rx[u].y, s4.y, t4.y), 0.f);
                b.z = fmaxf(fmaf(rx[u].z, s4.z, t4.z), 0.f); b.w = fmaxf(fmaf(rx[u].w, s4.w, t4.w), 0.f);
            } else {
                b = rx[u];
            }
            if (r >= r_end) { a = f4zero(); b = f4zero(); }
            if (coords) {
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    wx[0][dd] = fmaf(a.x, rd[u][dd], wx[0][dd]); wx[1][dd] = fmaf(a.y, rd[u][dd], wx[1][dd]);
                    wx[2][dd] = fmaf(a.z, rd[u][dd], wx[2][dd]); wx[3][dd] = fmaf(a.w, rd[u][dd], wx[3][dd]);
                }
            }
            *reinterpret_cast<float4*>(As + (sr + 8 * u) * P + c4) = a;
            *reinterpret_cast<float4*>(Bs + (sr + 8 * u) * P + c4) = b;
        }
    };
    if (XM == 1 || GM == 1) load_idx(r_begin);
    load_regs(r_begin);
    __syncthreads();                                     // vP / sv / tv visible
    write_lds(0, r_begin);
    if (r_begin + KT < r_end) load_regs(r_begin + KT);
    __syncthreads();
    int it = 0;
    for (int rb0 = r_begin; rb0 < r_end; rb0 += KT, ++it) {
        const float* As = smem + (it & 1) * STAGE + half * P + wm * 64 + l31;
        const float* Bs = smem + (it & 1) * STAGE + KT * P + half * P + wn * 64 + l31;
#pragma unroll
        for (int s = 0; s < KT / 2; ++s) {
            const float a0 = As[2 * s * P], a1 = As[2 * s * P + 32];
            const float b0 = Bs[2 * s * P], b1 = Bs[2 * s * P + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (rb0 + KT < r_end) write_lds(it + 1, rb0 + KT);
        if (rb0 + 2 * KT < r_end) load_regs(rb0 + 2 * KT);
        __syncthreads();
    }
    float* pout = partial + (size_t)blockIdx.y * n_out * Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = n0 + wm * 64 + a * 32 + acc_row(v, half), k = k0 + wn * 64 + b * 32 + l31;
                pout[(size_t)n * Kp + k] = acc[a][b][v];
            }
    if (coords) {                                        // eight staging rows per channel quad -> one sum (the tile LDS is free)
        float* red = smem;                               // [8][128][3]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) red[(sr * BT + c4 + j) * 3 + dd] = wx[j][dd];
        __syncthreads();
        for (int i = tid; i < BT * 3; i += 256) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[q * BT * 3 + i];
            pout[(size_t)(n0 + i / 3) * Kp + x.feat_c + i % 3] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wide-tile dW in split-bf16 form (option "mfma_split", GAD_SPLIT_DW_WIDE): the same 128 x 128 block per workgroup, 64 x 64
// per wavefront, K-tile = 32 rows.  The reduction index (rows) is the SLOW index of both operands in memory, while
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction indices per lane: so a thread stages a 4-row x 4-channel patch
// (rows 4 rq .. + 3 of the K-tile, channels 4 cq .. + 3; rq = tid & 7: a wavefront's loads cover 8 rows x 128 contiguous
// bytes), forms dZ / the activated input in registers, splits ROW PAIRS of one channel into hi / mid / lo and stores
// 8 bytes (four rows) per channel and plane into LDS laid out [plane][channel][32 rows] (64-byte rows, 16-byte chunks
// XOR-swizzled by (channel >> 2) & 3: fragments are conflict-free ds_read_b128 of 8 consecutive rows).  Rows 16 .. 31 of every
// K-tile carry NEGATED dZ and accumulate into the second accumulator set (the bf16 MFMA's truncation bias cancels in the
// difference).  One workgroup per CU (eight accumulators per wavefront), LDS double-buffered, one barrier per K-tile.
// ------------------------------------------------------------------------------------------------
template <int XM, int GM>
__global__ __launch_bounds__(256, 1) void gemm_dw_wide_split_kernel(DzSrc d, XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                                    int n_rows_static, int Kp, int n_out, int tiles_k,
                                                                    float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt_(ts);
    constexpr int BT = 128, OPL = BT * 64, STAGE = 6 * OPL, VM = 512;      // bytes: one operand plane, one stage (A planes | B planes)
    __shared__ __attribute__((aligned(16))) unsigned char smem_b[2 * STAGE];
    __shared__ __attribute__((aligned(16))) float vP[5 * VM];               // P | Q | S of the dZ channels, scale | shift of the input channels
    float* sv = vP + 3 * VM;
    float* tv = sv + VM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int n0 = (blockIdx.x / tiles_k) * BT, k0 = (blockIdx.x % tiles_k) * BT;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    int chunk = gad_cdiv_dev(n_rows, (int)gridDim.y);
    chunk = (chunk + KT - 1) / KT * KT;                  // == dw_reduce_kernel's split geometry
    const int r_begin = blockIdx.y * chunk, r_end = min(r_begin + chunk, n_rows);
    if (r_begin >= r_end) return;                        // the reducer skips the same splits
    for (int i = tid; i < BT; i += 256) {
        float Pc, Qc, Sc;
        dz_coef(d, n0 + i, Pc, Qc, Sc, blockIdx.y == 0 && k0 == 0);
        vP[i] = Pc; vP[VM + i] = Qc; vP[2 * VM + i] = Sc;
        if (XM == 0) { sv[i] = x.scale[k0 + i]; tv[i] = x.shift[k0 + i]; }
    }
    const bool coords = XM == 1 && k0 == 0;              // this workgroup also forms dW[n][feat_c .. feat_c + 2]
    float wx[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wx[j][0] = 0.f; wx[j][1] = 0.f; wx[j][2] = 0.f; }
    f32x16 acc[2][2], an[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[a][b][v] = 0.f; an[a][b][v] = 0.f; }
    // staging patch of this thread: rows 4 rq + u (u < 4) of the K-tile, channels c4 .. c4 + 3
    const int rq = tid & 7, c4 = (tid >> 3) * 4;
    const float sgn = rq >= 4 ? -1.f : 1.f;              // rows 16 .. 31: negated dZ
    const int gpitch = GM == 0 ? d.g_pitch : d.c;
    float4 rz[4], rg[4], rx[4];
    int4 ra[4];
    float rw[4], rd[4][3];
    int pq[4], gq[4], gd[4];                             // point / group of the NEXT tile's rows (gathered input, pooled gradient)
    auto load_idx = [&](int rb0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + 4 * rq + u;
            const int rr = r < r_end ? r : r_end - 1;
            pq[u] = XM == 1 ? x.row_pt[rr] : 0;
            gq[u] = (XM == 1 && x.ctr_xyz) ? x.row_grp[rr] : 0;
            gd[u] = GM == 1 ? d.row_grp[rr] : 0;
        }
    };
    auto load_regs = [&](int rb0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + 4 * rq + u;
            const int rr = r < r_end ? r : r_end - 1;
            rw[u] = d.row_w ? d.row_w[rr] : 1.f;
            rz[u] = ldg4(d.z + (size_t)rr * d.z_pitch + n0 + c4);
            if (GM == 0) {
                rg[u] = ldg4(d.G + (size_t)rr * gpitch + n0 + c4);
            } else {
                const int grp = gd[u];
                ra[u] = *reinterpret_cast<const int4*>(d.argmax + (size_t)grp * gpitch + n0 + c4);
                rg[u] = ldg4(d.dout + (size_t)grp * gpitch + n0 + c4);
            }
            if (XM == 0) {
                rx[u] = ldg4(x.zin + (size_t)rr * x.zin_pitch + k0 + c4);
            } else {
                rx[u] = ldg4(x.feat + (size_t)pq[u] * x.feat_c + k0 + c4);
                if (coords) {
                    const float* p = x.src_xyz + (size_t)pq[u] * 3;
                    float q0 = p[0], q1 = p[1], q2 = p[2];
                    if (x.ctr_xyz) {
                        const float* cp = x.ctr_xyz + (size_t)gq[u] * 3;
                        q0 = __fsub_rn(q0, cp[0]); q1 = __fsub_rn(q1, cp[1]); q2 = __fsub_rn(q2, cp[2]);
                    }
                    rd[u][0] = q0; rd[u][1] = q1; rd[u][2] = q2;
                }
            }
        }
        if (XM == 1 || GM == 1) load_idx(rb0 + KT);
    };
    // [plane][channel][rows]: this thread's 8 bytes (rows 4 rq .. + 3) of channel c4 + j
    const int wr_off = c4 * 64 + (rq & 1) * 8, wr_q = rq >> 1;
    auto write_lds = [&](int it, int rb0) {
        unsigned char* As = smem_b + (it & 1) * STAGE;
        unsigned char* Bs = As + 3 * OPL;
        const float4 Pv = *reinterpret_cast<const float4*>(vP + c4), Qv = *reinterpret_cast<const float4*>(vP + VM + c4);
        const float4 Sv = *reinterpret_cast<const float4*>(vP + 2 * VM + c4);
        float4 s4 = f4zero(), t4 = f4zero();
        if (XM == 0) { s4 = *reinterpret_cast<const float4*>(sv + c4); t4 = *reinterpret_cast<const float4*>(tv + c4); }
        float av[4][4], bv[4][4];                        // [row u][channel j]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb0 + 4 * rq + u;
            float4 g = rg[u];
            const float4 z = rz[u];
            if (GM == 1) {
                g.x = ra[u].x == r ? g.x : 0.f; g.y = ra[u].y == r ? g.y : 0.f;
                g.z = ra[u].z == r ? g.z : 0.f; g.w = ra[u].w == r ? g.w : 0.f;
            }
            const float w = rw[u];
            float4 a, b;
            a.x = Pv.x * g.x - w * fmaf(Sv.x, z.x, Qv.x); a.y = Pv.y * g.y - w * fmaf(Sv.y, z.y, Qv.y);
            a.z = Pv.z * g.z - w * fmaf(Sv.z, z.z, Qv.z); a.w = Pv.w * g.w - w * fmaf(Sv.w, z.w, Qv.w);
            if (XM == 0) {
                b.x = fmaxf(fmaf(rx[u].x, s4.x, t4.x), 0.f); b.y = fmaxf(fmaf(rx[u].y, s4.y, t4.y), 0.f);
                b.z = fmaxf(fmaf(rx[u].z, s4.z, t4.z), 0.f); b.w = fmaxf(fmaf(rx[u].w, s4.w, t4.w), 0.f);
            } else {
                b = rx[u];
            }
            if (r >= r_end) { a = f4zero(); b = f4zero(); }
            if (coords) {
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    wx[0][dd] = fmaf(a.x, rd[u][dd], wx[0][dd]); wx[1][dd] = fmaf(a.y, rd[u][dd], wx[1][dd]);
                    wx[2][dd] = fmaf(a.z, rd[u][dd], wx[2][dd]); wx[3][dd] = fmaf(a.w, rd[u][dd], wx[3][dd]);
                }
            }
            av[u][0] = a.x * sgn; av[u][1] = a.y * sgn; av[u][2] = a.z * sgn; av[u][3] = a.w * sgn;
            bv[u][0] = b.x; bv[u][1] = b.y; bv[u][2] = b.z; bv[u][3] = b.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = wr_off + j * 64 + (((wr_q ^ (((c4 + j) >> 2) & 3))) << 4);
            unsigned h0, m0, l0, h1, m1, l1;
            gad_split2(av[0][j], av[1][j], h0, m0, l0);
            gad_split2(av[2][j], av[3][j], h1, m1, l1);
            *reinterpret_cast<uint2*>(As + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(As + OPL + off) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(As + 2 * OPL + off) = make_uint2(l0, l1);
            gad_split2(bv[0][j], bv[1][j], h0, m0, l0);
            gad_split2(bv[2][j], bv[3][j], h1, m1, l1);
            *reinterpret_cast<uint2*>(Bs + off) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(Bs + OPL + off) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(Bs + 2 * OPL + off) = make_uint2(l0, l1);
        }
    };
    if (XM == 1 || GM == 1) load_idx(r_begin);
    load_regs(r_begin);
    __syncthreads();                                     // vP / sv / tv visible
    write_lds(0, r_begin);
    if (r_begin + KT < r_end) load_regs(r_begin + KT);
    __syncthreads();
    // fragment addressing: channel rows wm * 64 + a * 32 + l31 (A) / wn * 64 + b * 32 + l31 (B); chunk 2 st + half, swizzled
    const int fo = ((half ^ ((l31 >> 2) & 3)) << 4);
    const int arow = (wm * 64 + l31) * 64, brow = 3 * OPL + (wn * 64 + l31) * 64;
    int it = 0;
    for (int rb0 = r_begin; rb0 < r_end; rb0 += KT, ++it) {
        const unsigned char* st = smem_b + (it & 1) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int off = fo ^ (s2 << 5);
            gad_u32x4 A[2][3], B[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    A[a][pl] = *reinterpret_cast<const gad_u32x4*>(st + arow + a * (32 * 64) + pl * OPL + off);
                    B[a][pl] = *reinterpret_cast<const gad_u32x4*>(st + brow + a * (32 * 64) + pl * OPL + off);
                }
            // the six products of weight >= 2^-16, smallest first: (A plane, B plane) = (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
#define GAD_SPD(PA, PB_)                                                                                                       \
            _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) {                      \
                if (s2) an[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A[a][PA]), gad_as_bf16x8(B[b][PB_]), an[a][b], 0, 0, 0); \
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gad_as_bf16x8(A[a][PA]), gad_as_bf16x8(B[b][PB_]), acc[a][b], 0, 0, 0); \
            }
            GAD_SPD(2, 0) GAD_SPD(0, 2) GAD_SPD(1, 1) GAD_SPD(1, 0) GAD_SPD(0, 1) GAD_SPD(0, 0)
#undef GAD_SPD
        }
        if (rb0 + KT < r_end) write_lds(it + 1, rb0 + KT);
        if (rb0 + 2 * KT < r_end) load_regs(rb0 + 2 * KT);
        __syncthreads();
    }
    float* pout = partial + (size_t)blockIdx.y * n_out * Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = n0 + wm * 64 + a * 32 + acc_row(v, half), k = k0 + wn * 64 + b * 32 + l31;
                pout[(size_t)n * Kp + k] = acc[a][b][v] - an[a][b][v];
            }
    if (coords) {                                        // 8 row quads per channel quad -> one sum (the tile LDS is free)
        float* red = reinterpret_cast<float*>(smem_b);   // [8][128][3]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) red[(rq * BT + c4 + j) * 3 + dd] = wx[j][dd];
        __syncthreads();
        for (int i = tid; i < BT * 3; i += 256) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[q * BT * 3 + i];
            pout[(size_t)(n0 + i / 3) * Kp + x.feat_c + i % 3] = sum;
        }
    }
}

// the wide-tile dW covers: ACT input with BatchNorm + ReLU in front, one group, 128-multiples on both sides, no bias / extra column
static bool dw_wideable(const gad_gemm_dw_args& a, int k_used, bool vec) {
    const gad_gemm_fwd_args& in = a.in;
    const gad_dz_src& d = a.dz;
    if (!g_opt_dw_wide || !vec || in.n_groups != 1 || in.zin_off[0] != 0 || a.dz_off[0] != 0) return false;
    if (in.n_rows < 2048 || in.n_out[0] % 128 != 0 || in.n_out[0] > 512 || in.ones_col >= 0) return false;
    if (in.mode == 1) {                                  // gathered first layer: features a multiple of 128, + 3 coordinates
        if (g_opt_dw_wide == 2 || d.gmode != 0 || in.act_c != 0 || in.feat_c % 128 != 0 || in.feat_c > 512) return false;
        if (in.Kp != ((in.feat_c + 3 + 7) & ~7) || k_used != in.feat_c + 3) return false;
    } else {
        if (in.Kp % 128 != 0 || in.Kp > 512 || k_used != in.Kp) return false;
        if (in.c_in != in.Kp || !in.scale || !in.shift || !in.relu || in.extra) return false;
    }
    if (!d.z || d.z_pitch % 4 != 0 || !d.relu || !d.premasked || !(d.coefP && d.coefQ && d.coefS)) return false;
    if (d.gmode == 0 ? (d.g_pitch % 4 != 0 || !d.G) : (d.c % 4 != 0)) return false;
    return a.partial != nullptr && a.row_splits <= 0;
}

static bool dw_gather_streamable(const gad_gemm_dw_args& a, int k_used) {
    const gad_gemm_fwd_args& in = a.in;
    const gad_dz_src& d = a.dz;
    if (!g_opt_dw_stream || in.mode != 1 || in.n_groups != 1 || in.Kp > 32 || in.n_out[0] != 64 || k_used > in.Kp) return false;
    if (in.zin_off[0] != 0 || a.dz_off[0] != 0 || in.n_rows < 32768 || d.gmode != 0 || !d.G) return false;
    if (!d.z || !d.scale || !d.relu || !d.premasked || !(d.coefP && d.coefQ && d.coefS)) return false;
    if (in.feat_c + 3 + (in.action ? in.act_c : 0) > in.Kp) return false;
    if (!a.partial || 256ll * 64 * in.Kp > a.partial_elems) return false;
    return a.row_splits <= 0;
}

static bool dw_streamable(const gad_gemm_dw_args& a, int k_used) {
    const gad_gemm_fwd_args& in = a.in;
    const gad_dz_src& d = a.dz;
    if (!g_opt_dw_stream || in.mode != 0 || in.n_groups != 1 || in.Kp != 64 || in.c_in != 64 || k_used != 64) return false;
    if (in.zin_off[0] != 0 || a.dz_off[0] != 0 || (in.n_out[0] != 64 && in.n_out[0] != 128)) return false;   // w_off: arena offset, dw_reduce applies it
    if (in.n_rows < 32768 || !in.scale || !in.shift || !in.relu || in.extra || in.ones_col >= 0) return false;
    if (!d.z || !d.scale || !d.relu || !d.premasked || !(d.coefP && d.coefQ && d.coefS)) return false;
    if (d.gmode != 0 && d.c != in.n_out[0]) return false;
    if (!a.partial || (long long)DW_STREAM_SPLITS * in.n_out[0] * 64 > a.partial_elems) return false;
    return a.row_splits <= 0;
}

extern "C" int gad_gemm_dw(const gad_gemm_dw_args* a, void* stream) {
    unsigned long long* ts = gad_take_timing_slot(stream);
    GAD_REQUIRE(a && a->gacc, GAD_ERR_NULL, "gemm_dw: null pointer");
    const gad_gemm_fwd_args& in = a->in;
    GAD_REQUIRE(in.n_groups >= 1 && in.n_groups <= GAD_MAX_GROUPS, GAD_ERR_SHAPE, "gemm_dw: n_groups");
    GAD_REQUIRE(in.Kp % 8 == 0, GAD_ERR_SHAPE, "gemm_dw: Kp must be a multiple of 8");
    if (int e = check_input(in, "gemm_dw")) return e;
    GAD_REQUIRE(a->dz.gmode == 0 ? a->dz.G != nullptr : (a->dz.argmax && a->dz.dout && a->dz.row_grp), GAD_ERR_NULL,
                "gemm_dw: gradient source");
    if (in.n_rows <= 0) return GAD_OK;
    XSrc x = make_xsrc(in);
    DzSrc d = make_dzsrc(a->dz);
    // group g: dz channel offset dz_off[g], input channel offset zin_off[g], weights at w_off[g]
    Groups gr = make_groups(in.n_groups, a->dz_off, in.w_off, in.zin_off, in.n_out);
    const int nmax = max_nout(gr);
    int k_used = in.mode == 0 ? in.c_in + (in.extra ? 1 : 0) : in.feat_c + 3 + in.act_c;
    if (in.ones_col >= k_used) k_used = in.ones_col + 1;
    if (k_used > in.Kp) k_used = in.Kp;
    hipStream_t st = (hipStream_t)stream;
    const int rows = in.n_rows;
    const bool vec = dz_vectorizable(a->dz, a->dz_off, in.n_out, in.n_groups);
    const bool skinny_route = g_opt_dw_skinny && in.mode == 0 && a->dz.gmode == 0 && !in.n_rows_dev && rows <= 1024;
    if (a->dz.bn_dbeta && (skinny_route || dw_gather_streamable(*a, k_used) || dw_streamable(*a, k_used))) {
        gad_dz_src dz2 = a->dz;                          // these kernels read P / Q / S per lane: gad_bn_bwd_coef first
        if (int e2 = coef_fallback(dz2, stream)) return e2;
        d = make_dzsrc(dz2);
    }
    if (skinny_route) {
        if (g_opt_skinny_nw == 4)
            hipLaunchKernelGGL(gemm_dw_skinny_kernel<4>, dim3(gad_cdiv(nmax, 32), gad_cdiv(k_used, 32), gr.n), dim3(64 * 4), 0, st, d, x,
                               gr, rows, in.Kp, k_used, a->gacc, ts);
        else
            hipLaunchKernelGGL(gemm_dw_skinny_kernel<SK_NW>, dim3(gad_cdiv(nmax, 32), gad_cdiv(k_used, 32), gr.n), dim3(64 * SK_NW), 0, st, d, x,
                               gr, rows, in.Kp, k_used, a->gacc, ts);
        GAD_CHECK_LAUNCH("gemm_dw(skinny)");
        return GAD_OK;
    }
    if (dw_gather_streamable(*a, k_used)) {
        const int splits = 256;
        const int gps = in.grp_per_sample > 0 ? in.grp_per_sample : 1;
        hipLaunchKernelGGL(gemm_dw_gather_stream_kernel, dim3(splits), dim3(512), 0, st, d, x, in.n_rows_dev, rows, splits, in.Kp, k_used,
                           1.0f / (float)gps, a->partial, ts);
        GAD_CHECK_LAUNCH("gemm_dw(gather stream)");
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)nmax * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), gr.n), dim3(256), 0,
                           st, a->partial, (long long)splits * nmax * in.Kp, gr, in.n_rows_dev, rows, splits, in.Kp, k_used, a->gacc, 1);
        GAD_CHECK_LAUNCH("dw_reduce");
        return GAD_OK;
    }
    if (dw_streamable(*a, k_used)) {
        const int splits = DW_STREAM_SPLITS;
        float* part = a->partial;
        if (in.n_out[0] == 64) {
            if (a->dz.gmode == 0) hipLaunchKernelGGL((gemm_dw_stream_kernel<1, 0>), dim3(splits), dim3(512), 0, st, d, x, in.n_rows_dev, rows, splits, part, ts);
            else                  hipLaunchKernelGGL((gemm_dw_stream_kernel<1, 1>), dim3(splits), dim3(512), 0, st, d, x, in.n_rows_dev, rows, splits, part, ts);
        } else {
            if (a->dz.gmode == 0) hipLaunchKernelGGL((gemm_dw_stream_kernel<2, 0>), dim3(splits), dim3(512), 0, st, d, x, in.n_rows_dev, rows, splits, part, ts);
            else                  hipLaunchKernelGGL((gemm_dw_stream_kernel<2, 1>), dim3(splits), dim3(512), 0, st, d, x, in.n_rows_dev, rows, splits, part, ts);
        }
        GAD_CHECK_LAUNCH("gemm_dw(stream)");
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)nmax * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), gr.n), dim3(256), 0,
                           st, part, (long long)splits * nmax * in.Kp, gr, in.n_rows_dev, rows, splits, in.Kp, k_used, a->gacc, 1);
        GAD_CHECK_LAUNCH("dw_reduce");
        return GAD_OK;
    }
    if (dw_wideable(*a, k_used, vec)) {
        const int tn_ = in.n_out[0] / 128, tk_ = (in.mode == 1 ? in.feat_c : in.Kp) / 128;
        int splits = gad_cdiv(g_opt_dw_wide_wgs, tn_ * tk_);             // ~one workgroup per CU
        const int by_rows = gad_cdiv(rows, 4 * KT);
        if (splits > by_rows) splits = by_rows;
        if (splits < 1) splits = 1;
        if ((long long)splits * in.n_out[0] * in.Kp <= a->partial_elems && split_on(GAD_SPLIT_DW_WIDE)) {
            // split-bf16 products: both operands are formed and split by the kernel itself (no weight mirror involved)
#define LAUNCH_DWS(XM, GM)                                                                                                     \
            hipLaunchKernelGGL((gemm_dw_wide_split_kernel<XM, GM>), dim3(tn_ * tk_, splits), dim3(256), 0, st, d, x, in.n_rows_dev, rows, \
                               in.Kp, in.n_out[0], tk_, a->partial, ts)
            if (in.mode == 1) LAUNCH_DWS(1, 0); else if (a->dz.gmode == 0) LAUNCH_DWS(0, 0); else LAUNCH_DWS(0, 1);
#undef LAUNCH_DWS
            GAD_CHECK_LAUNCH("gemm_dw(wide split)");
            hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)nmax * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), gr.n), dim3(256),
                               0, st, a->partial, (long long)splits * nmax * in.Kp, gr, in.n_rows_dev, rows, splits, in.Kp, k_used, a->gacc);
            GAD_CHECK_LAUNCH("dw_reduce");
            return GAD_OK;
        }
        if ((long long)splits * in.n_out[0] * in.Kp <= a->partial_elems) {
            if (in.mode == 1)
                hipLaunchKernelGGL((gemm_dw_wide_kernel<1, 0>), dim3(tn_ * tk_, splits), dim3(256), 0, st, d, x, in.n_rows_dev, rows, in.Kp,
                                   in.n_out[0], tk_, a->partial, ts);
            else if (a->dz.gmode == 0)
                hipLaunchKernelGGL((gemm_dw_wide_kernel<0, 0>), dim3(tn_ * tk_, splits), dim3(256), 0, st, d, x, in.n_rows_dev, rows, in.Kp,
                                   in.n_out[0], tk_, a->partial, ts);
            else
                hipLaunchKernelGGL((gemm_dw_wide_kernel<0, 1>), dim3(tn_ * tk_, splits), dim3(256), 0, st, d, x, in.n_rows_dev, rows, in.Kp,
                                   in.n_out[0], tk_, a->partial, ts);
            GAD_CHECK_LAUNCH("gemm_dw(wide)");
            hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)nmax * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), gr.n), dim3(256),
                               0, st, a->partial, (long long)splits * nmax * in.Kp, gr, in.n_rows_dev, rows, splits, in.Kp, k_used, a->gacc);
            GAD_CHECK_LAUNCH("dw_reduce");
            return GAD_OK;
        }
    }
    long long group_stride = 0;
#define LAUNCH_DW4(WM, WN, TM, TN, XM, V, VM)                                                              \
    hipLaunchKernelGGL((gemm_dw_kernel<WM, WN, TM, TN, XM, V, VM>), dim3(tn_ * tk_, splits, gr.n), dim3(256), 0, st, d, \
                       x, gr, in.n_rows_dev, rows, in.Kp, k_used, tk_, a->gacc, part, group_stride, ts)
#define LAUNCH_DW3(WM, WN, TM, TN, XM, V)                                                                  \
    do { if (narrow) LAUNCH_DW4(WM, WN, TM, TN, XM, V, 512); else LAUNCH_DW4(WM, WN, TM, TN, XM, V, VMAX); } while (0)
#define LAUNCH_DW(WM, WN, TM, TN)                                                                          \
    do {                                                                                                   \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                                \
        const int tn_ = gad_cdiv(nmax, BM), tk_ = gad_cdiv(k_used, BN);                                    \
        splits = a->row_splits;                                                                            \
        if (splits <= 0) {                                                                                 \
            splits = gad_cdiv(512, tn_ * tk_ * gr.n);                     /* 256 / 1024 / 2048 measured slower */                                                    \
            const int by_rows = gad_cdiv(rows, 8 * KT);                                                    \
            if (splits > by_rows) splits = by_rows;                                                        \
            if (splits < 1) splits = 1;                                                                    \
        }                                                                                                  \
        group_stride = (long long)splits * nmax * in.Kp;                                                   \
        if (part && (splits == 1 || group_stride * gr.n > a->partial_elems)) part = nullptr; \
        if (in.mode == 0) { if (vec) LAUNCH_DW3(WM, WN, TM, TN, 0, true); else LAUNCH_DW3(WM, WN, TM, TN, 0, false); } \
        else              { if (vec) LAUNCH_DW3(WM, WN, TM, TN, 1, true); else LAUNCH_DW3(WM, WN, TM, TN, 1, false); } \
    } while (0)
    int splits = 1;
    float* part = a->partial;
    const bool narrow = nmax <= 512 && (in.mode != 0 || in.c_in <= 512);   // per-channel vectors fit a 512-float stride
    if (nmax <= 32) {
        LAUNCH_DW(1, 4, 1, 1);      // 32 x 128
    } else if (k_used <= 32) {
        LAUNCH_DW(4, 1, 1, 1);      // 128 x 32
    } else {
        LAUNCH_DW(2, 2, 1, 1);      // 64 x 64
    }
#undef LAUNCH_DW
#undef LAUNCH_DW3
#undef LAUNCH_DW4
    GAD_CHECK_LAUNCH("gemm_dw");
    if (part) {
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)nmax * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), gr.n),
                           dim3(256), 0, st, part,
                           group_stride, gr, in.n_rows_dev, rows, splits, in.Kp, k_used, a->gacc);
        GAD_CHECK_LAUNCH("dw_reduce");
    }
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// FUSED wide-tile backward for the MID-SIZE layers (SA2 / SA3; round 4): dX AND dW of one layer from ONE pass over dZ.
// The separate kernels (gemm_dx_wide / gemm_dw_wide) each form dZ = P*dY - w*(Q + S*z) from their own loads of z and dY
// (1 + K/128 times per element for dX, once per 128-wide k block for dW), and each launch is only a few K-tiles deep per
// workgroup: barrier and load latency, not MFMA issue, is what they spend their time on (0.17 - 0.25 of the FP32-MFMA peak
// inside the step).  Here a workgroup owns 64-row tiles x ONE 64-wide block of input channels (grid.y = K / 64) and walks the
// layer's N output channels in steps of 64: every staged dZ tile [64 rows x 64 channels] and W tile [64 channels x 64 k]
// feeds 32 dX MFMAs (32 rows x 32 k per wavefront, reduction over the 64 channels) AND 32 dW MFMAs (32 channels x 32 k per
// wavefront, reduction over the tile's 64 rows) -- twice the MFMA work per barrier, per staged byte and per dZ formation.
//   * dZ tile: row-major [row][64 + 4] -- the dX A fragments are ds_read_b128 along the channels (k = 8j+4h+i order), the dW
//     A fragments (dZ^T) are conflict-free ds_read_b32 across the channels of rows 2s + h;
//   * W tile as stored ([n][64 + 4]): the dX B fragments are ds_read_b32 across k of row n = 8j+4h+i -- no transposing store;
//   * the dW B operand -- the layer INPUT X[row][k], k = this lane's column -- never touches LDS: 32 coalesced 4-byte loads
//     per lane and row tile (relu(scale * z_prev + shift) with the lane's own scale / shift, or the gathered feature), held
//     in 32 registers and reused by every channel step of the tile;
//   * the dW accumulators (N / 64 per wavefront: the wavefront pair (w >> 1) splits each 64-channel step by halves) persist
//     over the workgroup's row tiles and leave as ONE partial block per workgroup; dw_reduce sums the blocks in f64.  (f64
//     atomics straight into the arena are SLOWER than slab + reduce on this part: 5.6 vs 3.5 us per million partial
//     elements, tools/ubench/atomic_reduce.hip -> profiles/r04_atomic_reduce.txt.)
//   * dX epilogue as in gemm_dx_wide: dY of the previous layer masked by its ReLU + that layer's BatchNorm-backward sums, or
//     (L1: the gathered first layers) float atomics into the points' feature gradients; L1 also forms the three coordinate
//     columns of dW with vector FMAs while dZ is staged (k block 0 only).
// One LDS buffer + register prefetch (two barriers per 64 MFMAs): 36 KB of LDS and <= 128 VGPRs of accumulators leave room
// for 3 - 4 workgroups per CU, which is what hides the barriers here.
// ------------------------------------------------------------------------------------------------
template <int NIT, int GM, int L1>
__global__ __launch_bounds__(256, NIT >= 8 ? 1 : 2) void gemm_bwd_wide_kernel(DzSrc d, XSrc x, const int32_t* __restrict__ n_rows_dev,
                                                               int n_rows_static, const float* __restrict__ W, int Kp, DxEpi e,
                                                               float* __restrict__ partial, unsigned long long* __restrict__ ts) {
    KTimer kt_(ts);
    constexpr int N = NIT * 64, P = 68, TILE = 64 * P;
    __shared__ __attribute__((aligned(16))) float smem[2 * TILE + 3 * N + 64 * 4];
    __shared__ int32_t ptS[64];
    float* As = smem;
    float* Ws = smem + TILE;
    float* vP = smem + 2 * TILE;                          // P | Q | S of the layer's N channels
    float* relS = vP + 3 * N;                             // L1: [row][4] = src_xyz[pt] - ctr_xyz[grp]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // dX: row half / k half;  dW: channel half of the step / k half
    const int l31 = lane & 31, half = lane >> 5;
    const int k0 = blockIdx.y * 64;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int n_tiles = (n_rows + 63) >> 6;
    const int c4 = (tid & 15) * 4, sr = tid >> 4;         // staging: 16-byte chunk of the 64-wide tile, rows sr + 16 u
    for (int i = tid; i < N; i += 256) {
        float Pc, Qc, Sc;
        dz_coef(d, i, Pc, Qc, Sc, first_workgroup());
        vP[i] = Pc; vP[N + i] = Qc; vP[2 * N + i] = Sc;
    }
    // every global operand goes through a buffer descriptor: 32-bit per-lane byte offsets (one VGPR each, set up once per
    // tile) + a wave-uniform scalar offset per access, instead of a 64-bit pointer pair per unrolled load
    const int gpitch = GM == 0 ? d.g_pitch : d.c;
    // z, dY and the layer input are bounded by the LIVE rows: a read past them returns 0, so a dead row's dZ is exactly 0
    // (its weight is 0 as well) without a select per element -- on this part every VALU instruction is MFMA issue time
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.z), 0, gad_nbytes(n_rows, d.z_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t gr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GM == 0 ? d.G : d.dout), 0,
                                                                         GM == 0 ? gad_nbytes(n_rows, gpitch * 4) : GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(GM == 0 ? d.row_grp : d.argmax), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, GAD_BUF_MAX, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L1 ? x.feat : x.zin), 0,
                                                                        L1 ? GAD_BUF_MAX : gad_nbytes(n_rows, x.zin_pitch * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sr_ = __builtin_amdgcn_make_buffer_rsrc(partial, 0, GAD_BUF_MAX, 0x00020000);
    // previous layer's raw output / the gradient this launch writes: bounded by the live rows (reads past them give 0,
    // stores past them are dropped)
    const int zp_pitch4 = L1 ? 0 : e.zprev_pitch * 4, go_pitch4 = L1 ? 0 : e.gout_pitch * 4;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L1 ? W : e.zprev), 0, L1 ? 0 : gad_nbytes(n_rows, zp_pitch4), 0x00020000);
    const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(L1 ? const_cast<float*>(W) : e.gout, 0, L1 ? 0 : gad_nbytes(n_rows, go_pitch4), 0x00020000);
    const int kx = k0 + wn * 32 + l31;                    // this lane's input channel (dW column, dX column)
    float xs = 1.f, xt = 0.f, ps = 0.f, pt = 0.f, pm = 0.f, pi = 0.f;
    if (!L1) { xs = x.scale[kx]; xt = x.shift[kx]; ps = e.ps[kx]; pt = e.pt[kx]; pm = e.pm[kx]; pi = e.pi[kx]; }
    const bool coords = L1 && k0 == 0;
    float wx[NIT][3];                                     // coords: channel it * 64 + lane, rows wave * 16 .. + 15 of every tile
    if (L1) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { wx[it][0] = 0.f; wx[it][1] = 0.f; wx[it][2] = 0.f; }
    }
    f32x16 accw[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int v = 0; v < 16; ++v) accw[it][v] = 0.f;
    float cb = 0.f, cg = 0.f;
    int vw[4];                                            // W staging offsets: rows sr + 16 u of the step's 64 channels
#pragma unroll
    for (int u = 0; u < 4; ++u) vw[u] = ((sr + 16 * u) * Kp + k0 + c4) * 4;
    const int xpitch4 = L1 ? x.feat_c * 4 : x.zin_pitch * 4;
    const int vx = (half * (L1 ? 0 : x.zin_pitch) + kx) * 4;             // X fragment lane offset (row 2 s + half, column kx)
    const int vzp = (4 * half * (L1 ? 0 : e.zprev_pitch) + kx) * 4, vgo = (4 * half * (L1 ? 0 : e.gout_pitch) + kx) * 4;
    // this thread's four staged rows of a tile (offsets, weights) and the staging registers; the NEXT tile's first step is
    // loaded under the current tile's last MFMAs and epilogue (cross-tile prefetch: a workgroup has 1 - 3 tiles, so the
    // staging latency in front of every tile was a third of its time)
    int vz[4], vg[4];
    float wrow[4];
    float4 rz[4], rg[4], rb[4];
    gad_u32x4 ra[4];
    auto meta = [&](int row0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = row0 + sr + 16 * u;
            const bool live = r < n_rows;
            const int rr = live ? r : max(n_rows - 1, 0);
            const float w = d.row_w ? d.row_w[rr] : 1.f;
            wrow[u] = live ? w : 0.f;
            vz[u] = (r * d.z_pitch + c4) * 4;             // (the true row: past the live rows the bounded descriptor reads 0)
            vg[u] = ((GM == 1 ? d.row_grp[rr] : r) * gpitch + c4) * 4;
        }
    };
    auto load_regs = [&](int it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rz[u] = buf_ld4(zr, vz[u], it * 256);
            rg[u] = buf_ld4(gr_, vg[u], it * 256);
            if (GM == 1) ra[u] = __builtin_amdgcn_raw_buffer_load_b128(ar, vg[u], it * 256, 0);
            rb[u] = buf_ld4(wr, vw[u], it * 256 * Kp);
        }
    };
    if ((int)blockIdx.x < n_tiles) { meta(blockIdx.x << 6); load_regs(0); }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile << 6;
        __syncthreads();                                  // vP visible; the previous tile's LDS reads (As, Ws, ptS, relS) are done
        if (L1 && tid < 64) {
            const int r = min(row0 + tid, max(n_rows - 1, 0));
            const int p = x.row_pt[r];
            ptS[tid] = p;
            const float* q = x.src_xyz + (size_t)p * 3;
            float q0 = q[0], q1 = q[1], q2 = q[2];
            if (x.ctr_xyz) {
                const float* cp = x.ctr_xyz + (size_t)x.row_grp[r] * 3;
                q0 = __fsub_rn(q0, cp[0]); q1 = __fsub_rn(q1, cp[1]); q2 = __fsub_rn(q2, cp[2]);
            }
            *reinterpret_cast<float4*>(relS + 4 * tid) = make_float4(q0, q1, q2, 0.f);
        }
        if (L1) __syncthreads();
        // ---- the dW B operand: X[row0 + 2 s + half][kx], s = 0 .. 31, straight into registers (activated below)
        float xf[32];
#pragma unroll
        for (int sidx = 0; sidx < 32; ++sidx) {
            if (L1) {
                xf[sidx] = buf_ld(xr, ptS[2 * sidx + half] * xpitch4 + vx, 0);
            } else {
                xf[sidx] = buf_ld(xr, vx, (row0 + 2 * sidx) * xpitch4);     // (past the live rows: 0; their dZ is 0 as well)
            }
        }
        f32x16 accx;
#pragma unroll
        for (int v = 0; v < 16; ++v) accx[v] = 0.f;
        float zp[16];
        const int rb0 = row0 + wm * 32;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it > 0) __syncthreads();                  // the previous step's fragment reads are done
            {   // registers -> LDS: dZ formed once per staged element
                const int nb = it * 64 + c4;
                const float4 P4 = *reinterpret_cast<const float4*>(vP + nb), Q4 = *reinterpret_cast<const float4*>(vP + N + nb);
                const float4 S4 = *reinterpret_cast<const float4*>(vP + 2 * N + nb);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 g = rg[u];
                    const float4 z = rz[u];
                    if (GM == 1) {
                        const unsigned r = row0 + sr + 16 * u;   // the true row (a clamped row never matches an arg-max)
                        g.x = ra[u].x == r ? g.x : 0.f; g.y = ra[u].y == r ? g.y : 0.f;
                        g.z = ra[u].z == r ? g.z : 0.f; g.w = ra[u].w == r ? g.w : 0.f;
                    }
                    const float w = wrow[u];
                    float4 v;
                    v.x = P4.x * g.x - w * fmaf(S4.x, z.x, Q4.x); v.y = P4.y * g.y - w * fmaf(S4.y, z.y, Q4.y);
                    v.z = P4.z * g.z - w * fmaf(S4.z, z.z, Q4.z); v.w = P4.w * g.w - w * fmaf(S4.w, z.w, Q4.w);
                    *reinterpret_cast<float4*>(As + (sr + 16 * u) * P + c4) = v;
                    *reinterpret_cast<float4*>(Ws + (sr + 16 * u) * P + c4) = rb[u];
                }
            }
            if (it + 1 < NIT) {
                load_regs(it + 1);
            } else {
                const int nxt = tile + gridDim.x;         // next tile's first step (wave-uniform branch)
                if (nxt < n_tiles) { meta(nxt << 6); load_regs(0); }
                if (!L1) {                                // this tile's z_prev for the epilogue: in flight under the last MFMAs
#pragma unroll
                    for (int v = 0; v < 16; ++v) zp[v] = buf_ld(pr, vzp, (rb0 + (v & 3) + 8 * (v >> 2)) * zp_pitch4);
                }
            }
            if (it == 0) {                                // activate the X fragments once per tile (their loads have had a whole staging pass)
                if (!L1) {
#pragma unroll
                    for (int sidx = 0; sidx < 32; ++sidx) xf[sidx] = fmaxf(fmaf(xf[sidx], xs, xt), 0.f);
                }
            }
            __syncthreads();
            if (coords) {                                 // dW[n][feat_c .. + 2] += dZ[r][n] * rel[r][0 .. 2]: lane = channel, 16 rows per wavefront
                const float* ap = As + (wave * 16) * P + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = ap[r * P];
                    const float4 q = *reinterpret_cast<const float4*>(relS + 4 * (wave * 16 + r));
                    wx[it][0] = fmaf(a, q.x, wx[it][0]); wx[it][1] = fmaf(a, q.y, wx[it][1]); wx[it][2] = fmaf(a, q.z, wx[it][2]);
                }
            }
            // ---- dX: 32 rows x 32 k of this wavefront, reduction over the step's 64 channels
            {
                const float* ap = As + (wm * 32 + l31) * P + 4 * half;
                const float* bp = Ws + (4 * half) * P + wn * 32 + l31;
                float4 a4 = *reinterpret_cast<const float4*>(ap);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 an = a4;
                    if (j + 1 < 8) an = *reinterpret_cast<const float4*>(ap + 8 * (j + 1));
                    const float b0 = bp[(8 * j + 0) * P], b1 = bp[(8 * j + 1) * P], b2 = bp[(8 * j + 2) * P], b3 = bp[(8 * j + 3) * P];
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, accx, 0, 0, 0);
                    accx = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, accx, 0, 0, 0);
                    a4 = an;
                }
            }
            // ---- dW: 32 channels (half wm of the step) x 32 k, reduction over the tile's 64 rows
            {
                const float* ap = As + half * P + wm * 32 + l31;
#pragma unroll
                for (int s8 = 0; s8 < 4; ++s8) {
                    float a[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = ap[(2 * (8 * s8 + q)) * P];
#pragma unroll
                    for (int q = 0; q < 8; ++q) accw[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], xf[8 * s8 + q], accw[it], 0, 0, 0);
                }
            }
        }
        // ---- dX epilogue
        if (L1) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int il = wm * 32 + acc_row(v, half);
                if (row0 + il < n_rows) atomic_add_f32(e.dfeat + (size_t)ptS[il] * e.feat_c + kx, accx[v]);
            }
        } else {
            // (a row past the live count: z_prev reads 0, its accumulator row is exactly 0 -- dZ = 0 -- and the store is dropped)
            const float npm = -pm * pi;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float ga = fmaf(zp[v], ps, pt) > 0.f ? accx[v] : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ga), or_, vgo, (rb0 + (v & 3) + 8 * (v >> 2)) * go_pitch4, 0);
                cb += ga;
                cg = fmaf(ga, fmaf(zp[v], pi, npm), cg);
            }
        }
    }
    // ---- this workgroup's partial dW block (zeros if it had no tile: dw_reduce sums every block)
    float* pout = partial + (size_t)blockIdx.x * N * Kp;
    {
        const int vs = (4 * half * Kp + kx) * 4, sbase = (blockIdx.x * N + wm * 32) * Kp * 4;       // (< 2^31: the slab is <= 36 MB)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accw[it][v]), sr_, vs, sbase + (it * 64 + (v & 3) + 8 * (v >> 2)) * Kp * 4, 0);
    }
    __syncthreads();                                      // the tile loop's last LDS reads are done: As / Ws are free
    if (coords) {
        float* red = smem;                                // [4 wavefronts = row quarters][N][3]
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) red[(wave * N + it * 64 + lane) * 3 + dd] = wx[it][dd];
        __syncthreads();
        for (int i = tid; i < N * 3; i += 256)
            pout[(size_t)(i / 3) * Kp + x.feat_c + i % 3] = red[i] + red[N * 3 + i] + red[2 * N * 3 + i] + red[3 * N * 3 + i];
        __syncthreads();
    }
    if (!L1) {
        const int rep = blockIdx.x % GAD_STAT_REPLICAS;
        const float c0[1] = {cb}, c1[1] = {cg};
        block_column_atomics<2, 2, 1>(smem, c0, c1, lane, wm, wn, k0, e.k_valid, e.dbeta + (size_t)rep * e.stat_stride,
                                      e.dgamma + (size_t)rep * e.stat_stride);
    }
}

// workgroups per k block of a fused wide launch = partial dW blocks it writes (static: the reduce launch recomputes it)
static int bwd_wide_splits(const gad_gemm_dx_args& ax) {
    const long long per = (long long)ax.n_out[0] * ax.Kp;
    long long g = (long long)g_opt_bwd_wide_slab * 1000000ll / per;
    const int tiles = gad_cdiv(ax.n_rows, 64);
    const int kb = ax.k_valid / 64;
    if (g * kb > 1024) g = 1024 / kb;                    // (no more workgroups than fit the chip at once)
    if (g > tiles) g = tiles;
    if (g < 16) g = 16 < tiles ? 16 : tiles;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------
// dX and dW of one layer in one call.  The SA1 layers the two streaming kernels cover take the fused kernel
// (gemm_bwd_stream_kernel); everything else runs gad_gemm_dw then gad_gemm_dx on the same stream.
// ------------------------------------------------------------------------------------------------
// the fused wide backward covers what gemm_dx_wide + gemm_dw_wide cover together: one group, BatchNorm + ReLU on both
// sides (premasked gradients), N in {128, 256, 512}, K a multiple of 64; the gathered first layers with their scatter epilogue
static bool bwd_wideable(const gad_gemm_dx_args& ax, const gad_gemm_dw_args& aw, bool vec) {
    const gad_gemm_fwd_args& in = aw.in;
    if (!g_opt_bwd_wide || !vec || ax.n_groups != 1 || in.n_groups != 1) return false;
    if (ax.dz_off[0] != 0 || ax.w_off[0] != 0 || ax.gout_off[0] != 0 || in.zin_off[0] != 0 || aw.dz_off[0] != 0) return false;
    const int N = ax.n_out[0];
    if (N != 128 && N != 256 && N != 512) return false;
    if (ax.n_rows < 2048 || (g_opt_bwd_wide == 2 && ax.n_rows < 16384) || ax.n_rows != in.n_rows || ax.n_rows_dev != in.n_rows_dev) return false;
    if (in.n_out[0] != N || in.Kp != ax.Kp || in.ones_col >= 0 || !aw.partial || !aw.gacc || !ax.W) return false;
    if (ax.k_valid % 64 != 0 || ax.k_valid < 64 || (long long)bwd_wide_splits(ax) * N * ax.Kp > aw.partial_elems) return false;   // (workspace of another route)
    const gad_dz_src& d = ax.dz;
    if (!d.z || d.z_pitch % 4 != 0 || !d.relu || !d.premasked || !(d.coefP && d.coefQ && d.coefS)) return false;
    if (d.gmode == 0 ? (d.g_pitch % 4 != 0 || !d.G) : (d.c % 4 != 0 || !d.argmax || !d.dout || !d.row_grp)) return false;
    // both descriptions are of the same layer
    if (d.z != aw.dz.z || d.gmode != aw.dz.gmode || d.G != aw.dz.G || d.dout != aw.dz.dout || d.argmax != aw.dz.argmax ||
        d.row_w != aw.dz.row_w || d.coefP != aw.dz.coefP) return false;
    if (ax.epilogue == 1) {                              // gathered first layer: scatter into the points' feature gradients
        if (in.mode != 1 || d.gmode != 0 || !ax.dfeat || ax.daction || !ax.row_pt || ax.prev_dbeta || in.act_c != 0) return false;
        if (in.feat_c % 64 != 0 || in.feat_c > 512 || ax.k_valid != in.feat_c || ax.feat_c != in.feat_c) return false;
        return in.Kp == ((in.feat_c + 3 + 7) & ~7) && in.row_pt == ax.row_pt;
    }
    if (in.mode != 0 || !ax.prev_dbeta || !ax.store_masked || !ax.gout) return false;
    if (!(ax.zprev && ax.prev_scale && ax.prev_shift && ax.prev_mean && ax.prev_istd && ax.prev_dgamma)) return false;
    if (in.Kp % 64 != 0 || in.Kp > 512 || in.c_in != in.Kp || ax.k_valid != in.Kp || !in.scale || !in.shift || !in.relu || in.extra) return false;
    return in.zin == ax.zprev && in.scale == ax.prev_scale && in.shift == ax.prev_shift && in.zin_pitch == ax.zprev_pitch;
}

static bool bwd_streamable(const gad_gemm_dx_args* ax, const gad_gemm_dw_args* aw) {
    const gad_gemm_fwd_args& in = aw->in;
    if (!g_opt_bwd_fused) return false;
    const bool vec = dz_vectorizable(ax->dz, ax->dz_off, ax->n_out, ax->n_groups);
    int k_used = in.mode == 0 ? in.c_in + (in.extra ? 1 : 0) : in.feat_c + 3 + in.act_c;
    bool fused = dx_streamable(*ax, vec) && dw_streamable(*aw, k_used);
    // the two descriptions must be of the same layer: same dZ source, and dW's input = the layer dX feeds
    return fused && ax->dz.z == aw->dz.z && ax->dz.gmode == aw->dz.gmode && ax->dz.G == aw->dz.G && ax->dz.dout == aw->dz.dout &&
           ax->dz.argmax == aw->dz.argmax && ax->dz.row_w == aw->dz.row_w && ax->n_rows == in.n_rows && ax->n_rows_dev == in.n_rows_dev &&
           ax->n_out[0] == in.n_out[0] && in.zin == ax->zprev && in.scale == ax->prev_scale && in.shift == ax->prev_shift &&
           in.zin_pitch == 64 && ax->gout && ax->prev_dgamma;
}

// the reduce launch of a fused backward call whose aw->row_splits was GAD_DW_REDUCE_LATER (include/gaddpg.h): sums the partial
// dW blocks that call left in aw->partial into the arena.  A layer gad_gemm_bwd does not fuse has reduced already: no-op.
extern "C" int gad_gemm_dw_reduce(const gad_gemm_dx_args* ax, const gad_gemm_dw_args* aw, void* stream) {
    GAD_REQUIRE(ax && aw, GAD_ERR_NULL, "gemm_dw_reduce: null pointer");
    const gad_gemm_fwd_args& in = aw->in;
    if (ax->n_rows <= 0) return GAD_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = dz_vectorizable(ax->dz, ax->dz_off, ax->n_out, ax->n_groups);
    const bool streamable = bwd_streamable(ax, aw);
    if (!streamable && bwd_wideable(*ax, *aw, vec)) {
        const int splits = bwd_wide_splits(*ax);
        const int k_used = in.mode == 1 ? in.feat_c + 3 : in.Kp;
        Groups gr = make_groups(1, aw->dz_off, in.w_off, in.zin_off, in.n_out);
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)in.n_out[0] * in.Kp, 256), gad_cdiv(splits, DW_RED_CHUNK), 1), dim3(256), 0,
                           st, aw->partial, (long long)splits * in.n_out[0] * in.Kp, gr, in.n_rows_dev, ax->n_rows, splits, in.Kp, k_used,
                           aw->gacc, 1);
        GAD_CHECK_LAUNCH("dw_reduce");
        return GAD_OK;
    }
    if (streamable) {
        const int wgs = g_opt_bwd_stream_wgs;
        const int splits = ax->n_out[0] == 64 ? 2 * wgs : wgs;
        Groups gr = make_groups(1, aw->dz_off, in.w_off, in.zin_off, in.n_out);
        hipLaunchKernelGGL(dw_reduce_kernel, dim3(gad_cdiv((long long)in.n_out[0] * 64, 256), gad_cdiv(splits, DW_RED_CHUNK), 1), dim3(256), 0, st,
                           aw->partial, (long long)splits * in.n_out[0] * 64, gr, in.n_rows_dev, ax->n_rows, splits, 64, 64, aw->gacc, 1);
        GAD_CHECK_LAUNCH("dw_reduce");
    }
    return GAD_OK;
}

extern "C" int gad_gemm_bwd(const gad_gemm_dx_args* ax, const gad_gemm_dw_args* aw, void* stream) {
    GAD_REQUIRE(ax && aw, GAD_ERR_NULL, "gemm_bwd: null pointer");
    const gad_gemm_fwd_args& in = aw->in;
    const bool later = aw->row_splits == GAD_DW_REDUCE_LATER;          // the caller launches gad_gemm_dw_reduce itself
    const bool vec0 = dz_vectorizable(ax->dz, ax->dz_off, ax->n_out, ax->n_groups);
    const bool streamable = bwd_streamable(ax, aw);                    // (the SA1 shapes: the streaming kernel has precedence)
    if (!streamable && bwd_wideable(*ax, *aw, vec0)) {
        unsigned long long* ts = gad_take_timing_slot(stream);
        (void)gad_take_grid_rows();
        if (ax->n_rows <= 0) return GAD_OK;
        const int N = ax->n_out[0], splits = bwd_wide_splits(*ax);
        GAD_REQUIRE((long long)splits * N * in.Kp <= aw->partial_elems, GAD_ERR_SHAPE, "gemm_bwd(wide): partial workspace too small (%lld floats needed)",
                    (long long)splits * N * in.Kp);
        DzSrc d = make_dzsrc(ax->dz);
        XSrc x = make_xsrc(in);
        DxEpi e;
        e.mode = ax->epilogue; e.gout = ax->gout; e.gout_pitch = ax->gout_pitch; e.k_valid = ax->k_valid;
        e.zprev = ax->zprev; e.zprev_pitch = ax->zprev_pitch; e.ps = ax->prev_scale; e.pt = ax->prev_shift;
        e.pm = ax->prev_mean; e.pi = ax->prev_istd; e.dbeta = ax->prev_dbeta; e.dgamma = ax->prev_dgamma;
        e.stat_stride = ax->stat_stride; e.store_masked = 1;
        e.dfeat = ax->dfeat; e.feat_c = ax->feat_c; e.row_pt = ax->row_pt; e.row_grp = ax->row_grp; e.daction = nullptr; e.act_c = 0; e.gps = 1;
        hipStream_t st = (hipStream_t)stream;
        const dim3 grid(splits, ax->k_valid / 64);
#define LAUNCH_BWW(NIT, GM, L1) hipLaunchKernelGGL((gemm_bwd_wide_kernel<NIT, GM, L1>), grid, dim3(256), 0, st, d, x, ax->n_rows_dev, ax->n_rows, ax->W, ax->Kp, e, aw->partial, ts)
        const bool l1 = ax->epilogue == 1, pooled = ax->dz.gmode != 0;
        if (N == 128) { if (l1) LAUNCH_BWW(2, 0, 1); else if (pooled) LAUNCH_BWW(2, 1, 0); else LAUNCH_BWW(2, 0, 0); }
        else if (N == 256) { if (l1) LAUNCH_BWW(4, 0, 1); else if (pooled) LAUNCH_BWW(4, 1, 0); else LAUNCH_BWW(4, 0, 0); }
        else { if (l1) LAUNCH_BWW(8, 0, 1); else if (pooled) LAUNCH_BWW(8, 1, 0); else LAUNCH_BWW(8, 0, 0); }
#undef LAUNCH_BWW
        GAD_CHECK_LAUNCH("gemm_bwd(wide)");
        return later ? GAD_OK : gad_gemm_dw_reduce(ax, aw, stream);
    }
    if (!streamable) {
        gad_gemm_dw_args aw2 = *aw;
        if (later) aw2.row_splits = 0;
        if (int e = gad_gemm_dw(&aw2, stream)) return e;
        return gad_gemm_dx(ax, stream);
    }
    unsigned long long* ts = gad_take_timing_slot(stream);
    (void)gad_take_grid_rows();
    GAD_REQUIRE(aw->gacc && ax->W, GAD_ERR_NULL, "gemm_bwd: null pointer");
    if (ax->n_rows <= 0) return GAD_OK;
    DzSrc d = make_dzsrc(ax->dz);
    DxEpi e;
    e.mode = 0; e.gout = ax->gout; e.gout_pitch = ax->gout_pitch; e.k_valid = ax->k_valid;
    e.zprev = ax->zprev; e.zprev_pitch = ax->zprev_pitch; e.ps = ax->prev_scale; e.pt = ax->prev_shift;
    e.pm = ax->prev_mean; e.pi = ax->prev_istd; e.dbeta = ax->prev_dbeta; e.dgamma = ax->prev_dgamma;
    e.stat_stride = ax->stat_stride; e.store_masked = 1;
    e.dfeat = nullptr; e.feat_c = 0; e.row_pt = nullptr; e.row_grp = nullptr; e.daction = nullptr; e.act_c = 0; e.gps = 1;
    hipStream_t st = (hipStream_t)stream;
    const int rows = ax->n_rows, wgs = g_opt_bwd_stream_wgs;
    const int splits = ax->n_out[0] == 64 ? 2 * wgs : wgs;            // partial dW blocks the kernel writes (see its header)
    GAD_REQUIRE((long long)splits * in.n_out[0] * 64 <= aw->partial_elems, GAD_ERR_SHAPE, "gemm_bwd: partial workspace too small");
    if (split_on(GAD_SPLIT_BWD_STREAM) && ax->W_split_t && ax->W_split_t_pitch == ax->n_out[0] && ax->W_split_t_plane >= 64 * ax->n_out[0]) {
#define LAUNCH_BWSS(NJ, GM) hipLaunchKernelGGL((gemm_bwd_stream_split_kernel<NJ, GM, true>), dim3(wgs), dim3(512), 0, st, d, ax->n_rows_dev, rows, \
                                                ax->W_split_t, ax->W_split_t_plane, e, aw->partial, ts)
        if (ax->n_out[0] == 128) { if (ax->dz.gmode == 0) LAUNCH_BWSS(16, 0); else LAUNCH_BWSS(16, 1); }
        else { if (ax->dz.gmode == 0) LAUNCH_BWSS(8, 0); else LAUNCH_BWSS(8, 1); }
#undef LAUNCH_BWSS
        GAD_CHECK_LAUNCH("gemm_bwd(stream split)");
        return later ? GAD_OK : gad_gemm_dw_reduce(ax, aw, stream);
    }
#define LAUNCH_BWS(NJ, GM) hipLaunchKernelGGL((gemm_bwd_stream_kernel<NJ, GM>), dim3(wgs), dim3(512), 0, st, d, ax->n_rows_dev, rows, ax->W, e, aw->partial, ts)
    if (ax->n_out[0] == 128) { if (ax->dz.gmode == 0) LAUNCH_BWS(16, 0); else LAUNCH_BWS(16, 1); }
    else { if (ax->dz.gmode == 0) LAUNCH_BWS(8, 0); else LAUNCH_BWS(8, 1); }
#undef LAUNCH_BWS
    GAD_CHECK_LAUNCH("gemm_bwd(stream)");
    return later ? GAD_OK : gad_gemm_dw_reduce(ax, aw, stream);
}
