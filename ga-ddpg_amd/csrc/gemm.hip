// FP32-MFMA GEMM family for the per-point shared MLPs, the FC head and the actor/critic heads.
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
// global loads issued before the current tile's MFMAs (register-staged software pipeline),
// producers (BN affine + ReLU, neighbourhood gather, BN-backward dZ) fused into the operand load,
// BatchNorm statistics / parameter gradients reduced in registers then f64 device atomics.
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KT 32   // reduction tile

// ------------------------------------------------------------------------------------------------
// operand producers
// ------------------------------------------------------------------------------------------------
struct XSrc {   // how a layer's INPUT row r, column c is produced (gad_gemm_fwd_args subset)
    int mode;
    const float* zin; int zin_pitch; int c_in;
    const float* scale; const float* shift; int relu;
    const float* extra; int ones_col;
    const float* src_xyz; const float* ctr_xyz; const float* feat; int feat_c;
    const float* action; int act_c; int gps;
    const int32_t* row_pt; const int32_t* row_grp;
};

static XSrc make_xsrc(const gad_gemm_fwd_args& a) {
    XSrc x;
    x.mode = a.mode; x.zin = a.zin; x.zin_pitch = a.zin_pitch; x.c_in = a.c_in;
    x.scale = a.scale; x.shift = a.shift; x.relu = a.relu; x.extra = a.extra; x.ones_col = a.ones_col;
    x.src_xyz = a.src_xyz; x.ctr_xyz = a.ctr_xyz; x.feat = a.feat; x.feat_c = a.feat_c;
    x.action = a.action; x.act_c = a.act_c; x.gps = a.grp_per_sample > 0 ? a.grp_per_sample : 1;
    x.row_pt = a.row_pt; x.row_grp = a.row_grp;
    return x;
}

__device__ __forceinline__ float x_elem(const XSrc& x, int r, int zoff, int c, int pt, int grp) {
    if (x.mode == 0) {
        if (c < x.c_in) {
            float v = x.zin[(size_t)r * x.zin_pitch + zoff + c];
            if (x.scale) v = fmaf(v, x.scale[zoff + c], x.shift[zoff + c]);
            if (x.relu) v = fmaxf(v, 0.f);
            return v;
        }
        if (c == x.c_in && x.extra) return x.extra[r];
        return c == x.ones_col ? 1.f : 0.f;
    }
    if (c < 3) {
        const float p = x.src_xyz[(size_t)pt * 3 + c];
        return x.ctr_xyz ? __fsub_rn(p, x.ctr_xyz[(size_t)grp * 3 + c]) : p;
    }
    c -= 3;
    if (c < x.feat_c) return x.feat[(size_t)pt * x.feat_c + c];
    c -= x.feat_c;
    if (c < x.act_c) return x.action[(size_t)(grp / x.gps) * x.act_c + c];
    return (c + 3 + x.feat_c) == x.ones_col ? 1.f : 0.f;
}

// four consecutive input columns c..c+3 of row r (c % 4 == 0)
__device__ __forceinline__ float4 x_load4(const XSrc& x, int r, bool valid, int zoff, int c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!valid) return v;
    if (x.mode == 0 && c + 3 < x.c_in) {
        v = *reinterpret_cast<const float4*>(x.zin + (size_t)r * x.zin_pitch + zoff + c);
        if (x.scale) {
            const float4 s = *reinterpret_cast<const float4*>(x.scale + zoff + c);
            const float4 t = *reinterpret_cast<const float4*>(x.shift + zoff + c);
            v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y);
            v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
        }
        if (x.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        return v;
    }
    int pt = 0, grp = 0;
    if (x.mode != 0) { pt = x.row_pt[r]; grp = x.row_grp[r]; }
    v.x = x_elem(x, r, zoff, c + 0, pt, grp);
    v.y = x_elem(x, r, zoff, c + 1, pt, grp);
    v.z = x_elem(x, r, zoff, c + 2, pt, grp);
    v.w = x_elem(x, r, zoff, c + 3, pt, grp);
    return v;
}

struct DzSrc {   // gad_dz_src on the device
    const float* z; int z_pitch; const float* scale; const float* shift; int relu;
    const float* P; const float* Q; const float* S; const float* row_w;
    int gmode; const float* G; int g_pitch; const int32_t* argmax; const float* dout;
    const int32_t* row_grp; int c;
};

static DzSrc make_dzsrc(const gad_dz_src& d) {
    DzSrc s;
    s.z = d.z; s.z_pitch = d.z_pitch; s.scale = d.scale; s.shift = d.shift; s.relu = d.relu;
    s.P = d.coefP; s.Q = d.coefQ; s.S = d.coefS; s.row_w = d.row_w;
    s.gmode = d.gmode; s.G = d.G; s.g_pitch = d.g_pitch; s.argmax = d.argmax; s.dout = d.dout;
    s.row_grp = d.row_grp; s.c = d.c;
    return s;
}

// dZ[r][n] for one element; `off` = channel offset of the group inside the layer
__device__ __forceinline__ float dz_elem(const DzSrc& d, int r, int off, int n, int nmax) {
    if (n >= nmax) return 0.f;
    const int ch = off + n;
    float z = 0.f;
    if (d.z) z = d.z[(size_t)r * d.z_pitch + ch];
    float g;
    if (d.gmode == 0) {
        g = d.G[(size_t)r * d.g_pitch + ch];
    } else {
        const int grp = d.row_grp[r];
        g = d.argmax[(size_t)grp * d.c + ch] == r ? d.dout[(size_t)grp * d.c + ch] : 0.f;
    }
    if (d.relu) {
        const float y = d.scale ? fmaf(z, d.scale[ch], d.shift[ch]) : z;
        if (!(y > 0.f)) g = 0.f;
    }
    if (d.P) {
        const float w = d.row_w ? d.row_w[r] : 1.f;
        g = d.P[ch] * g - w * fmaf(d.S[ch], z, d.Q[ch]);
    }
    return g;
}

__device__ __forceinline__ float4 dz_load4(const DzSrc& d, int r, bool valid, int off, int n, int nmax) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!valid) return v;
    v.x = dz_elem(d, r, off, n + 0, nmax);
    v.y = dz_elem(d, r, off, n + 1, nmax);
    v.z = dz_elem(d, r, off, n + 2, nmax);
    v.w = dz_elem(d, r, off, n + 3, nmax);
    return v;
}

// ------------------------------------------------------------------------------------------------
// LDS tile helpers.  Tiles are k-major: T[kk][i], kk in [0,KT), i in [0,DIM).
//   transposing store (source contiguous along kk): pitch DIM+1, four ds_write_b32
//   direct store      (source contiguous along i) : pitch DIM+4, one ds_write_b128
// ------------------------------------------------------------------------------------------------
template <int DIM> struct PitchT { static constexpr int v = DIM + 1; };
template <int DIM> struct PitchD { static constexpr int v = DIM + 4; };

// units of a (DIM x KT) tile whose source is contiguous along kk: u -> (i = u/8, kk4 = (u%8)*4)
template <int DIM> __device__ __forceinline__ void unit_T(int u, int& i, int& kk) { i = u >> 3; kk = (u & 7) << 2; }
// units of a (KT x DIM) tile whose source is contiguous along i: u -> (kk = u/(DIM/4), i4)
template <int DIM> __device__ __forceinline__ void unit_D(int u, int& kk, int& i) { kk = u / (DIM / 4); i = (u % (DIM / 4)) << 2; }

template <int DIM> __device__ __forceinline__ void store_T(float* t, int i, int kk, float4 v) {
    constexpr int P = PitchT<DIM>::v;
    t[(kk + 0) * P + i] = v.x; t[(kk + 1) * P + i] = v.y; t[(kk + 2) * P + i] = v.z; t[(kk + 3) * P + i] = v.w;
}
template <int DIM> __device__ __forceinline__ void store_D(float* t, int kk, int i, float4 v) {
    constexpr int P = PitchD<DIM>::v;
    *reinterpret_cast<float4*>(t + kk * P + i) = v;
}

// one K-tile of MFMAs: acc[tm][tn] += A^T-tile x B-tile
template <int TM, int TN, int PA, int PB>
__device__ __forceinline__ void mfma_ktile(const float* __restrict__ As, const float* __restrict__ Bs, int am0,
                                           int bn0, int lane, f32x16 (&acc)[TM][TN]) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int s = 0; s < KT / 2; ++s) {
        float a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = As[(2 * s + half) * PA + am0 + tm * 32 + l31];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[(2 * s + half) * PB + bn0 + tn * 32 + l31];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
}

// accumulator element v of a 32x32 tile -> row inside the tile (column is lane & 31)
__device__ __forceinline__ int acc_row(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

struct Groups {
    int n; int aoff[GAD_MAX_GROUPS]; int woff[GAD_MAX_GROUPS]; int ooff[GAD_MAX_GROUPS]; int nout[GAD_MAX_GROUPS];
};

// ------------------------------------------------------------------------------------------------
// forward:  zout[r][n] = sum_k X[r][k] * W[n][k]      (+ weighted BatchNorm statistics)
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_fwd_kernel(XSrc x, Groups gr, const int32_t* __restrict__ n_rows_dev,
                                                        int n_rows_static, const float* __restrict__ row_w,
                                                        const float* __restrict__ W, int Kp,
                                                        float* __restrict__ zout, int zout_pitch,
                                                        double* __restrict__ stat_sum,
                                                        double* __restrict__ stat_sq) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchT<BM>::v, PB = PitchT<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    __shared__ __attribute__((aligned(16))) float smem[KT * PA + KT * PB + BM];
    float* As = smem;
    float* Bs = smem + KT * PA;
    float* wS = Bs + KT * PB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int zoff = gr.aoff[g], n_out = gr.nout[g], ooff = gr.ooff[g];
    const float* Wg = W + gr.woff[g];
    const int n0 = blockIdx.y * BN;
    if (n0 >= n_out) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int nk = Kp / KT + ((Kp % KT) ? 1 : 0);

    float csum[TN], csq[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { csum[t] = 0.f; csq[t] = 0.f; }

    for (int row0 = blockIdx.x * BM; row0 < n_rows; row0 += gridDim.x * BM) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

        float4 ra[UA], rb[UB];
        auto load_tile = [&](int kt) {
            const int k0 = kt * KT;
#pragma unroll
            for (int it = 0; it < UA; ++it) {
                int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
                const int r = row0 + i;
                ra[it] = (k0 + kk < Kp) ? x_load4(x, r, r < n_rows, zoff, k0 + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < UB; ++it) {
                int j, kk; unit_T<BN>(it * 256 + tid, j, kk);
                const int n = n0 + j;
                rb[it] = (n < n_out && k0 + kk < Kp)
                             ? *reinterpret_cast<const float4*>(Wg + (size_t)n * Kp + k0 + kk)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        load_tile(0);
        if (tid < BM) {
            const int r = row0 + tid;
            wS[tid] = r < n_rows ? (row_w ? row_w[r] : 1.f) : 0.f;
        }
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int it = 0; it < UA; ++it) { int i, kk; unit_T<BM>(it * 256 + tid, i, kk); store_T<BM>(As, i, kk, ra[it]); }
#pragma unroll
            for (int it = 0; it < UB; ++it) { int j, kk; unit_T<BN>(it * 256 + tid, j, kk); store_T<BN>(Bs, j, kk, rb[it]); }
            __syncthreads();
            if (kt + 1 < nk) load_tile(kt + 1);
            mfma_ktile<TM, TN, PA, PB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
            __syncthreads();
        }
        // epilogue: store + statistics
        const int l31 = lane & 31, half = lane >> 5;
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
                    if (r < n_rows && n < n_out) zout[(size_t)r * zout_pitch + ooff + n] = zv;
                    const float w = wS[il];
                    s1 = fmaf(w, zv, s1);
                    s2 = fmaf(w * zv, zv, s2);
                }
            csum[tn] += s1;
            csq[tn] += s2;
        }
        __syncthreads();   // wS reuse
    }
    if (stat_sum) {
        const int l31 = lane & 31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            float s1 = csum[tn] + __shfl_xor(csum[tn], 32, 64);
            float s2 = csq[tn] + __shfl_xor(csq[tn], 32, 64);
            const int n = n0 + wn * TN * 32 + tn * 32 + l31;
            if (lane < 32 && n < n_out) {
                atomic_add_f64(stat_sum + ooff + n, (double)s1);
                atomic_add_f64(stat_sq + ooff + n, (double)s2);
            }
        }
    }
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

extern "C" int gad_gemm_fwd(const gad_gemm_fwd_args* a, void* stream) {
    GAD_REQUIRE(a && a->W && a->zout, GAD_ERR_NULL, "gemm_fwd: null pointer");
    GAD_REQUIRE(a->n_groups >= 1 && a->n_groups <= GAD_MAX_GROUPS, GAD_ERR_SHAPE, "gemm_fwd: n_groups");
    GAD_REQUIRE(a->Kp % 8 == 0 && a->Kp >= 8, GAD_ERR_SHAPE, "gemm_fwd: Kp=%d must be a multiple of 8", a->Kp);
    GAD_REQUIRE(a->mode == 1 || (a->zin && a->c_in % 4 == 0 && a->zin_pitch % 4 == 0), GAD_ERR_SHAPE,
                "gemm_fwd: ACT input needs c_in, pitch multiples of 4");
    GAD_REQUIRE(a->mode == 0 || (a->src_xyz && a->row_pt && a->row_grp), GAD_ERR_NULL, "gemm_fwd: gather inputs");
    if (a->n_rows <= 0) return GAD_OK;
    XSrc x = make_xsrc(*a);
    Groups gr = make_groups(a->n_groups, a->zin_off, a->w_off, a->out_off, a->n_out);
    const int nmax = max_nout(gr);
    hipStream_t st = (hipStream_t)stream;
    const int rows = a->n_rows;
#define LAUNCH_FWD(WM, WN, TM, TN)                                                                         \
    do {                                                                                                   \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                                \
        int gx = gad_cdiv(rows, BM); if (gx > 2048) gx = 2048;                                             \
        hipLaunchKernelGGL((gemm_fwd_kernel<WM, WN, TM, TN>), dim3(gx, gad_cdiv(nmax, BN), gr.n), dim3(256), \
                           0, st, x, gr, a->n_rows_dev, rows, a->row_w, a->W, a->Kp, a->zout, a->zout_pitch, \
                           a->stat_sum, a->stat_sq);                                                       \
    } while (0)
    if (rows <= 1024) {
        if (nmax <= 32) LAUNCH_FWD(4, 1, 1, 1); else LAUNCH_FWD(2, 2, 1, 1);     // many small tiles: fill the CUs
    } else if (nmax <= 64) {
        LAUNCH_FWD(4, 1, 1, 2);                                                   // 128 x 64
    } else {
        LAUNCH_FWD(2, 2, 2, 2);                                                   // 128 x 128
    }
#undef LAUNCH_FWD
    GAD_CHECK_LAUNCH("gemm_fwd");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// backward wrt the layer input:  gout[r][k] = sum_n dZ[r][n] * W[n][k]
// ------------------------------------------------------------------------------------------------
struct DxEpi {
    int mode; float* gout; int gout_pitch; int k_valid;
    const float* zprev; int zprev_pitch; const float* ps; const float* pt; const float* pm; const float* pi;
    double* dbeta; double* dgamma;
    float* dfeat; int feat_c; const int32_t* row_pt; const int32_t* row_grp; float* daction; int act_c; int gps;
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_dx_kernel(DzSrc d, Groups gr, const int32_t* __restrict__ n_rows_dev,
                                                       int n_rows_static, const float* __restrict__ W, int Kp,
                                                       DxEpi e) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchT<BM>::v, PB = PitchD<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    __shared__ __attribute__((aligned(16))) float smem[KT * PA + 3 + KT * PB + 2 * BM];
    float* As = smem;
    float* Bs = smem + ((KT * PA + 3) & ~3);
    int32_t* ptS = reinterpret_cast<int32_t*>(Bs + KT * PB);
    int32_t* grS = ptS + BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], n_out = gr.nout[g], goff = gr.ooff[g];
    const float* Wg = W + gr.woff[g];
    const int k0out = blockIdx.y * BN;
    if (k0out >= e.k_valid) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    const int nk = gad_cdiv_dev(n_out, KT);

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
        float4 ra[UA], rb[UB];
        auto load_tile = [&](int kt) {
            const int nb = kt * KT;
#pragma unroll
            for (int it = 0; it < UA; ++it) {
                int i, kk; unit_T<BM>(it * 256 + tid, i, kk);
                const int r = row0 + i;
                ra[it] = dz_load4(d, r, r < n_rows, doff, nb + kk, n_out);
            }
#pragma unroll
            for (int it = 0; it < UB; ++it) {
                int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
                const int n = nb + kk, k = k0out + j;
                rb[it] = (n < n_out && k < Kp) ? *reinterpret_cast<const float4*>(Wg + (size_t)n * Kp + k)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        load_tile(0);
        if (e.mode == 1 && tid < BM) {
            const int r = row0 + tid;
            ptS[tid] = r < n_rows ? e.row_pt[r] : 0;
            grS[tid] = r < n_rows ? e.row_grp[r] : 0;
        }
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int it = 0; it < UA; ++it) { int i, kk; unit_T<BM>(it * 256 + tid, i, kk); store_T<BM>(As, i, kk, ra[it]); }
#pragma unroll
            for (int it = 0; it < UB; ++it) { int kk, j; unit_D<BN>(it * 256 + tid, kk, j); store_D<BN>(Bs, kk, j, rb[it]); }
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
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int il = wm * TM * 32 + tm * 32 + acc_row(v, half);
                    const int r = row0 + il;
                    if (r >= n_rows || !kok) continue;
                    const float gv = acc[tm][tn][v];
                    if (e.mode == 0) {
                        e.gout[(size_t)r * e.gout_pitch + goff + k] = gv;
                        if (stats) {
                            const float zp = e.zprev[(size_t)r * e.zprev_pitch + goff + k];
                            if (fmaf(zp, sc, sh) > 0.f) { sb += gv; sg = fmaf(gv, (zp - mu) * is, sg); }
                        }
                    } else {
                        const int c = k - 3;
                        if (c >= 0 && c < e.feat_c) {
                            if (e.dfeat) atomic_add_f32(e.dfeat + (size_t)ptS[il] * e.feat_c + c, gv);
                        } else if (c >= e.feat_c && c < e.feat_c + e.act_c) {
                            if (e.daction) atomic_add_f32(e.daction + (size_t)(grS[il] / e.gps) * e.act_c + (c - e.feat_c), gv);
                        }
                    }
                }
            cb[tn] += sb; cg[tn] += sg;
        }
        __syncthreads();
    }
    if (e.dbeta) {
        const int l31 = lane & 31;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            float s1 = cb[tn] + __shfl_xor(cb[tn], 32, 64);
            float s2 = cg[tn] + __shfl_xor(cg[tn], 32, 64);
            const int k = k0out + wn * TN * 32 + tn * 32 + l31;
            if (lane < 32 && k < e.k_valid) {
                atomic_add_f64(e.dbeta + goff + k, (double)s1);
                atomic_add_f64(e.dgamma + goff + k, (double)s2);
            }
        }
    }
}

extern "C" int gad_gemm_dx(const gad_gemm_dx_args* a, void* stream) {
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
    e.dfeat = a->dfeat; e.feat_c = a->feat_c; e.row_pt = a->row_pt; e.row_grp = a->row_grp;
    e.daction = a->daction; e.act_c = a->act_c; e.gps = a->grp_per_sample > 0 ? a->grp_per_sample : 1;
    GAD_REQUIRE(e.mode == 0 || (e.row_pt && e.row_grp), GAD_ERR_NULL, "gemm_dx: scatter epilogue needs row maps");
    GAD_REQUIRE(!e.dbeta || (e.zprev && e.ps && e.pt && e.pm && e.pi && e.dgamma), GAD_ERR_NULL, "gemm_dx: prev BN stats inputs");
    hipStream_t st = (hipStream_t)stream;
    const int rows = a->n_rows, kv = a->k_valid;
#define LAUNCH_DX(WM, WN, TM, TN)                                                                        \
    do {                                                                                                 \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                              \
        int gx = gad_cdiv(rows, BM); if (gx > 2048) gx = 2048;                                           \
        hipLaunchKernelGGL((gemm_dx_kernel<WM, WN, TM, TN>), dim3(gx, gad_cdiv(kv, BN), gr.n), dim3(256), 0, \
                           st, d, gr, a->n_rows_dev, rows, a->W, a->Kp, e);                              \
    } while (0)
    if (rows <= 1024) {
        if (kv <= 32) LAUNCH_DX(4, 1, 1, 1); else LAUNCH_DX(2, 2, 1, 1);
    } else if (kv <= 32) {
        LAUNCH_DX(4, 1, 1, 1);
    } else if (kv <= 64) {
        LAUNCH_DX(4, 1, 1, 2);
    } else {
        LAUNCH_DX(2, 2, 2, 2);
    }
#undef LAUNCH_DX
    GAD_CHECK_LAUNCH("gemm_dx");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// backward wrt the weights:  gacc[n][k] += sum_r dZ[r][n] * X[r][k]     (f64 atomics, split rows)
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_dw_kernel(DzSrc d, XSrc x, Groups gr,
                                                       const int32_t* __restrict__ n_rows_dev, int n_rows_static,
                                                       int Kp, int k_used, int n_ktiles, double* __restrict__ gacc) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PA = PitchD<BM>::v, PB = PitchD<BN>::v;
    constexpr int UA = BM * 8 / 256, UB = BN * 8 / 256;
    __shared__ __attribute__((aligned(16))) float smem[KT * PA + KT * PB];
    float* As = smem;
    float* Bs = smem + KT * PA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = blockIdx.z;
    const int doff = gr.aoff[g], zoff = gr.ooff[g], n_out = gr.nout[g];
    double* out = gacc + gr.woff[g];
    const int tile_n = blockIdx.x / n_ktiles, tile_k = blockIdx.x % n_ktiles;
    const int n0 = tile_n * BM, k0 = tile_k * BN;
    if (n0 >= n_out) return;
    const int n_rows = n_rows_dev ? min(*n_rows_dev, n_rows_static) : n_rows_static;
    int chunk = gad_cdiv_dev(n_rows, (int)gridDim.y);
    chunk = (chunk + KT - 1) / KT * KT;
    const int r_begin = blockIdx.y * chunk;
    const int r_end = min(r_begin + chunk, n_rows);
    if (r_begin >= r_end) return;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    float4 ra[UA], rb[UB];
    auto load_tile = [&](int rb0) {
#pragma unroll
        for (int it = 0; it < UA; ++it) {
            int kk, i; unit_D<BM>(it * 256 + tid, kk, i);
            const int r = rb0 + kk;
            ra[it] = dz_load4(d, r, r < r_end, doff, n0 + i, n_out);
        }
#pragma unroll
        for (int it = 0; it < UB; ++it) {
            int kk, j; unit_D<BN>(it * 256 + tid, kk, j);
            const int r = rb0 + kk;
            rb[it] = (k0 + j < Kp) ? x_load4(x, r, r < r_end, zoff, k0 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_tile(r_begin);
    for (int rb0 = r_begin; rb0 < r_end; rb0 += KT) {
#pragma unroll
        for (int it = 0; it < UA; ++it) { int kk, i; unit_D<BM>(it * 256 + tid, kk, i); store_D<BM>(As, kk, i, ra[it]); }
#pragma unroll
        for (int it = 0; it < UB; ++it) { int kk, j; unit_D<BN>(it * 256 + tid, kk, j); store_D<BN>(Bs, kk, j, rb[it]); }
        __syncthreads();
        if (rb0 + KT < r_end) load_tile(rb0 + KT);
        mfma_ktile<TM, TN, PA, PB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
        __syncthreads();
    }
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int k = k0 + wn * TN * 32 + tn * 32 + l31;
        if (k >= k_used) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = n0 + wm * TM * 32 + tm * 32 + acc_row(v, half);
                if (n < n_out) atomic_add_f64(out + (size_t)n * Kp + k, (double)acc[tm][tn][v]);
            }
    }
}

extern "C" int gad_gemm_dw(const gad_gemm_dw_args* a, void* stream) {
    GAD_REQUIRE(a && a->gacc, GAD_ERR_NULL, "gemm_dw: null pointer");
    const gad_gemm_fwd_args& in = a->in;
    GAD_REQUIRE(in.n_groups >= 1 && in.n_groups <= GAD_MAX_GROUPS, GAD_ERR_SHAPE, "gemm_dw: n_groups");
    GAD_REQUIRE(in.Kp % 8 == 0, GAD_ERR_SHAPE, "gemm_dw: Kp must be a multiple of 8");
    GAD_REQUIRE(in.mode == 1 || (in.zin && in.c_in % 4 == 0 && in.zin_pitch % 4 == 0), GAD_ERR_SHAPE, "gemm_dw: ACT input");
    GAD_REQUIRE(a->dz.gmode == 0 ? a->dz.G != nullptr : (a->dz.argmax && a->dz.dout && a->dz.row_grp), GAD_ERR_NULL,
                "gemm_dw: gradient source");
    if (in.n_rows <= 0) return GAD_OK;
    XSrc x = make_xsrc(in);
    DzSrc d = make_dzsrc(a->dz);
    // group g: dz channel offset dz_off[g], input channel offset zin_off[g], weights at w_off[g]
    Groups gr = make_groups(in.n_groups, a->dz_off, in.w_off, in.zin_off, in.n_out);
    const int nmax = max_nout(gr);
    // number of real input columns (the rest of Kp is zero padding: skip its atomics)
    int k_used = in.mode == 0 ? in.c_in + (in.extra ? 1 : 0) : 3 + in.feat_c + in.act_c;
    if (in.ones_col >= k_used) k_used = in.ones_col + 1;
    if (k_used > in.Kp) k_used = in.Kp;
    hipStream_t st = (hipStream_t)stream;
    const int rows = in.n_rows;
#define LAUNCH_DW(WM, WN, TM, TN)                                                                          \
    do {                                                                                                   \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                                \
        const int tn_ = gad_cdiv(nmax, BM), tk_ = gad_cdiv(k_used, BN);                                    \
        int splits = a->row_splits;                                                                        \
        if (splits <= 0) {                                                                                 \
            splits = gad_cdiv(1024, tn_ * tk_ * gr.n);                                                     \
            const int by_rows = gad_cdiv(rows, 4 * KT);                                                    \
            if (splits > by_rows) splits = by_rows;                                                        \
            if (splits < 1) splits = 1;                                                                    \
        }                                                                                                  \
        hipLaunchKernelGGL((gemm_dw_kernel<WM, WN, TM, TN>), dim3(tn_ * tk_, splits, gr.n), dim3(256), 0, st, \
                           d, x, gr, in.n_rows_dev, rows, in.Kp, k_used, tk_, a->gacc);                    \
    } while (0)
    if (nmax <= 32) {
        if (k_used <= 32) LAUNCH_DW(1, 4, 1, 1); else LAUNCH_DW(1, 4, 1, 1);   // 32 x 128
    } else if (k_used <= 32) {
        LAUNCH_DW(4, 1, 1, 1);                                                  // 128 x 32
    } else {
        LAUNCH_DW(2, 2, 1, 1);                                                  // 64 x 64
    }
#undef LAUNCH_DW
    GAD_CHECK_LAUNCH("gemm_dw");
    return GAD_OK;
}
