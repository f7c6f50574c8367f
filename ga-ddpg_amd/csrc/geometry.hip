// PointNet++ geometry operators for gfx950: furthest point sampling, ball query, grouping (+grad),
// gather (+grad), and the compaction of ball-query output into de-duplicated neighbourhood rows.
// Replaces the CUDA kernels of the reference's external `pointnet2_ops._ext` (sampling_gpu.cu,
// ball_query_gpu.cu, group_points_gpu.cu; reached from reference core/networks.py:66-81).
// Design (MI355X-first, not a translation): wave64 ballot/prefix-popcount compaction for the ball
// query (one wavefront per centroid instead of one thread), register-resident clouds with a
// shuffle arg-max for FPS (one workgroup per cloud, no global scratch), 16-byte coalesced
// streams for the materialising group kernels.
#include "common.hpp"
#include <type_traits>

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";
static thread_local unsigned long long* g_timing_slot = nullptr;

unsigned long long* gad_take_timing_slot() {
    unsigned long long* p = g_timing_slot;
    g_timing_slot = nullptr;
    return p;
}

// wavefront issue priority per stream (gaddpg.h: gad_stream_priority): a small table, written before the steps start
static void* g_prio_stream[8];
static int g_prio_val[8];
static int g_prio_n = 0;
extern "C" int gad_stream_priority(void* stream, int prio) {
    GAD_REQUIRE(prio >= 0 && prio <= 3, GAD_ERR_SHAPE, "stream_priority: priority %d outside 0..3", prio);
    for (int i = 0; i < g_prio_n; ++i)
        if (g_prio_stream[i] == stream) { g_prio_val[i] = prio; return GAD_OK; }
    if (prio == 0) return GAD_OK;
    GAD_REQUIRE(g_prio_n < 8, GAD_ERR_SHAPE, "stream_priority: more than 8 prioritised streams");
    g_prio_stream[g_prio_n] = stream;
    g_prio_val[g_prio_n++] = prio;
    return GAD_OK;
}
unsigned long long* gad_take_timing_slot(void* stream) {
    size_t v = reinterpret_cast<size_t>(gad_take_timing_slot());
    for (int i = 0; i < g_prio_n; ++i)
        if (g_prio_stream[i] == stream) v |= (size_t)g_prio_val[i];
    return reinterpret_cast<unsigned long long*>(v);
}

extern "C" int gad_timing_slot(void* slot) {
    GAD_REQUIRE((reinterpret_cast<size_t>(slot) & 7) == 0, GAD_ERR_SHAPE, "timing_slot: the slot must be 8-byte aligned");
    g_timing_slot = static_cast<unsigned long long*>(slot);
    return GAD_OK;
}

// Grid-size hint for the next tiled gad_gemm_fwd / gad_gemm_dx launch of this thread (see gaddpg.h)
static thread_local int g_grid_rows = 0;
int gad_take_grid_rows() {
    const int r = g_grid_rows;
    g_grid_rows = 0;
    return r;
}
extern "C" int gad_grid_rows_hint(const int32_t* rows_host, void* /*stream: unused, plan calls pass one*/) {
    g_grid_rows = rows_host ? *rows_host : 0;
    return GAD_OK;
}

extern "C" int gad_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}
void gad_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gad_last_error(void) { return g_err; }
extern "C" int gad_abi_version(void) { return 11; }

const char* g_gad_last_kernel = "";
extern "C" const char* gad_last_kernel(void) { return g_gad_last_kernel; }

// ------------------------------------------------------------------------------------------------
// furthest point sampling
// ------------------------------------------------------------------------------------------------
// Arg-max candidates are packed as (float bits of the distance) << 32 | ~key: distances are >= +0, so their bit patterns
// order like the values, and among equal distances the SMALLER key must win (see the tie rule below); 0 = "no candidate".
// pack(a) > pack(b)  <=>  a.v > b.v || (a.v == b.v && a.key < b.key), the upstream comparison.
__device__ __forceinline__ unsigned long long fps_pack(float v, unsigned key) {
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(unsigned)(~key);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long fps_dpp_max(unsigned long long v) {      // lanes without a source keep v
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
    return o > v ? o : v;
}
// wavefront maximum (uniform result): DPP row shifts / row broadcasts instead of six ds_bpermute round trips
__device__ __forceinline__ unsigned long long fps_wave_max(unsigned long long v) {
    v = fps_dpp_max<0x111, 0xf>(v); v = fps_dpp_max<0x112, 0xf>(v); v = fps_dpp_max<0x114, 0xf>(v); v = fps_dpp_max<0x118, 0xf>(v);
    v = fps_dpp_max<0x142, 0xa>(v);                                // row_bcast:15 -> rows 1, 3
    v = fps_dpp_max<0x143, 0xc>(v);                                // row_bcast:31 -> rows 2, 3: lane 63 holds the maximum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// wavefront maximum of a float / minimum of an unsigned (uniform results), same DPP ladder
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float fps_dpp_fmax(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    return o > v ? o : v;
}
__device__ __forceinline__ float fps_wave_fmax(float v) {
    v = fps_dpp_fmax<0x111, 0xf>(v); v = fps_dpp_fmax<0x112, 0xf>(v); v = fps_dpp_fmax<0x114, 0xf>(v); v = fps_dpp_fmax<0x118, 0xf>(v);
    v = fps_dpp_fmax<0x142, 0xa>(v);
    v = fps_dpp_fmax<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// the same ladder as ONE instruction per step (v_max_f32 with a DPP source; a lane without a source is not written and keeps
// its own value).  hipcc turns the select form above into v_mov / v_mov_dpp / v_cmp / v_cndmask + hazard s_nops: 4x the issue
// slots on the one dependent chain every arg-max round waits for.  (s_nop 1: VALU write -> DPP read of the same register.)
__device__ __forceinline__ float fps_wave_fmax_asm(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// v_min_f32 / v_max3_f32 as they are (no canonicalising v_max x, x in front: the operands are never NaN here -- distances of
// finite coordinates, the -1 sentinel, 1e10)
__device__ __forceinline__ float fps_vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fps_vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fps_settle(float v) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned fps_dpp_umin(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned fps_wave_umin(unsigned v) {
    v = fps_dpp_umin<0x111, 0xf>(v); v = fps_dpp_umin<0x112, 0xf>(v); v = fps_dpp_umin<0x114, 0xf>(v); v = fps_dpp_umin<0x118, 0xf>(v);
    v = fps_dpp_umin<0x142, 0xa>(v);
    v = fps_dpp_umin<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

#ifdef GAD_FPS_PHASES
// diagnostic build (tools/ubench_fps.py --phases): cycles wavefront 0 of workgroup 0 spends in each part of an arg-max round
__device__ unsigned long long gad_fps_phase[8];
#define FPS_T(i) do { const unsigned long long t1_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) gad_fps_phase[i] += t1_ - t0_; t0_ = __builtin_readcyclecounter(); } while (0)
extern "C" int gad_fps_phase_read(unsigned long long* out8, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(gad_fps_phase), sizeof(z));
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(gad_fps_phase), z, sizeof(z));
    return 0;
}
#else
#define FPS_T(i) do { } while (0)
#endif
// One workgroup (WAVES wavefronts) per cloud; thread t keeps points k = s*T + t (s < NPL) in
// registers.  The cloud is also staged in LDS so the coordinates of the last pick are a broadcast
// LDS read.  Tie rule == the upstream block reduction: "thread" t = k mod tie_bs keeps its lowest k
// (strict >), and the pairwise tree `v2 > v1 ? i2 : i1` over strides bs/2..1 lets the candidate with
// the smallest BIT-REVERSED thread id win among equal values (slot t beats slot t+s at every level).
// (the library is built without packed-f32 instructions -- csrc/Makefile; this kernel opts back in: its packed
// operands are the points' long-lived register pairs and SCALAR pick coordinates, never a pair fresh from an LDS read, and the
// distance update is half of its round)
template <int NPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((target("packed-fp32-ops"))) void fps_kernel(const float* __restrict__ xyz, int N, int M,
                                                          int tie_bits, int32_t* __restrict__ idx,
                                                          float* __restrict__ new_xyz) {
    constexpr int T = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sp = lds;                                   // N*3 coordinates
    unsigned long long* red = reinterpret_cast<unsigned long long*>(lds + ((N * 3 + 3) & ~3));   // 2 x WAVES packed maxima
    int* pk = reinterpret_cast<int*>(lds + ((N * 3 + 3) & ~3) + 64);                             // the M picks (written out at the end)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = tid; i < N * 3; i += T) sp[i] = p[i];
    __syncthreads();
    int old = 0;
    if (tid == 0 && M > 0) pk[0] = 0;
    auto key_of = [&](int k) {
        const unsigned t = (unsigned)k & ((1u << tie_bits) - 1u);              // k mod tie_bs
        const unsigned rev = tie_bits ? (__brev(t) >> (32 - tie_bits)) : 0u;
        return (rev << 16) | (unsigned)k;
    };
#ifdef GAD_FPS_PHASES
    unsigned long long t0_ = __builtin_readcyclecounter();
#endif
    auto publish = [&](int j, unsigned long long best) {           // cross-wavefront maximum, the pick, its output
        if (WAVES > 1) {                                           // one barrier per pick: the exchange buffer alternates
            unsigned long long* r = red + (j & 1) * WAVES;
            if ((tid & 63) == 0) r[tid >> 6] = best;
            __syncthreads();
            FPS_T(4);
            best = r[0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) best = r[w] > best ? r[w] : best;
        }
        const unsigned bkey = best ? ~(unsigned)best : 0u;         // no candidate at all (every point skipped): index 0
        const int pick = (int)(bkey & 0xFFFFu);
        // the pick goes to LDS, the outputs are written once all M are known: a global store (and the LDS read of the pick's
        // coordinates for it) in here sat on wavefront 0's path to the next barrier -- 150 of a round's ~1300 cycles
        if (tid == 0) pk[j] = pick;
        FPS_T(5);
        return pick;
    };
    if constexpr (NPL % 2 == 0) {
        // ---- fast path (round 5).  The thread's points live as packed PAIRS (v_pk_*_f32 operands as they are: the first version
        // re-packed scalar arrays every round and the 16-point instantiation ran with 248 + 256 registers and 772 bytes of
        // scratch per lane), a skipped / absent point is a point whose temp is -1 (never updated -- min(d, -1) = -1 -- and never
        // a candidate), and the arg-max needs no 64-bit compare: the thread visits its points in ASCENDING KEY ORDER with a strict
        // '>' on the float distance (the first maximum = the smallest key stays), the wavefront takes the float maximum with a
        // DPP ladder and then asks which lanes hold it: one lane -> its key by v_readlane, several (exact ties) -> the smallest
        // key among them.  Key order inside a thread: key = (bit-reversed (k mod tie_bs) << 16) | k, k = s * T + tid: for
        // tie_bs <= T the keys ascend with s; for tie_bs = 2 T (T = 256, N >= 512) the even s come first.
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        constexpr int H = NPL / 2;
        f32x2_t qx[H], qy[H], qz[H], tm[H];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = (2 * h + u) * T + tid;
                const int kk = k < N ? k : N - 1;
                // (through a v_mov each: the packed instructions below must not be the first readers of registers an LDS read has just
                // filled -- DESIGN.md section 5; a scalar instruction in that position is safe)
                const float x = fps_settle(sp[kk * 3 + 0]), y = fps_settle(sp[kk * 3 + 1]), z = fps_settle(sp[kk * 3 + 2]);
                // upstream: `if (mag <= 1e-3) continue;` compares the FLOAT mag with the DOUBLE literal: the float nearest to 0.001
                // (0x3A83126F = 0.00100000005) is > 0.001 and therefore NOT skipped -- in float terms "skip iff mag < 1e-3f"
                const bool skip = k >= N || gad_sqnorm(x, y, z) < 1e-3f;
                qx[h][u] = x; qy[h][u] = y; qz[h][u] = z;
                tm[h][u] = skip ? -1.f : 1e10f;
            }
        const int ratio = (1 << tie_bits) / T;              // (tie_bs <= 512: at most 8 for one wavefront per cloud)
        for (int j = 1; j < M; ++j) {
            const float* po = sp + ((old << 1) + old);
            // the pick's coordinates as wave-uniform SCALARS (v_readfirstlane of the broadcast LDS read)
            const float x1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(po[0])));
            const float y1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(po[1])));
            const float z1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(po[2])));
            {
                // two points per instruction, each operation individually rounded like the scalar gad_sqdist (contraction off: the
                // indices must stay bit-identical to upstream's)
#pragma clang fp contract(off)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const f32x2_t dx = qx[h] - x1, dy = qy[h] - y1, dz = qz[h] - z1;
                    const f32x2_t xx = dx * dx, yy = dy * dy, zz = dz * dz;
                    const f32x2_t dd = (xx + yy) + zz;
                    tm[h][0] = fps_vmin(dd[0], tm[h][0]);
                    tm[h][1] = fps_vmin(dd[1], tm[h][1]);
                }
            }
            FPS_T(0);
            // the thread's maximum by a v_max3 tree -- the wavefront maximum below starts from it at once -- and, in its shadow,
            // WHICH of the thread's points holds it: the first in ascending key order (upstream's strict '>' keeps the smallest key
            // among equal distances), found by walking the points in DESCENDING key order with independent compares
            float bd;
            if constexpr (NPL == 2) {
                bd = fps_vmax3(tm[0][0], tm[0][1], -1.f);
            } else {
                float lv[NPL / 2];
#pragma unroll
                for (int h = 0; h < H; h += 1) lv[h] = h == 0 ? fps_vmax3(tm[0][0], tm[0][1], -1.f) : fps_vmax3(tm[h][0], tm[h][1], lv[h - 1]);
                bd = lv[H - 1];
            }
            const float dmax = fps_wave_fmax_asm(bd);
            FPS_T(2);
            int bs = 0;
            // key order: R = tie_bs / T (upstream's block vs this workgroup); key's leading part is the bit-reversed (s mod R),
            // then k: so s = m, m + R, m + 2 R, ... for m in bit-reversed counting order -- walked backwards here
            auto visit = [&](auto rc) {
                constexpr int R = decltype(rc)::value;
                constexpr int LR = R == 1 ? 0 : (R == 2 ? 1 : (R == 4 ? 2 : 3));
#pragma unroll
                for (int c = R - 1; c >= 0; --c) {
                    const int m = LR == 0 ? 0 : (int)(__builtin_bitreverse32((unsigned)c) >> (32 - (LR ? LR : 1)));
                    if (m < NPL) {
#pragma unroll
                        for (int q = (NPL - 1 - m) / R; q >= 0; --q) { const int s = m + q * R; bs = tm[s >> 1][s & 1] == bd ? s : bs; }
                    }
                }
            };
            if (ratio <= 1) visit(std::integral_constant<int, 1>());
            else if (ratio == 2) visit(std::integral_constant<int, 2>());
            else if (ratio == 4) visit(std::integral_constant<int, 4>());
            else visit(std::integral_constant<int, 8>());
            FPS_T(1);
            unsigned long long best = 0ull;
            if (dmax >= 0.f) {
                const unsigned bk = key_of(bs * T + tid);
                const unsigned long long tied = __ballot(bd == dmax);
                unsigned kb;
                if (__builtin_popcountll(tied) == 1) kb = (unsigned)__builtin_amdgcn_readlane((int)bk, __builtin_ctzll(tied));
                else kb = fps_wave_umin(bd == dmax ? bk : 0xffffffffu);
                best = fps_pack(dmax, kb);
            }
            FPS_T(3);
            old = publish(j, best);
        }
    } else {
        float px[NPL], py[NPL], pz[NPL], tmp[NPL];
        unsigned key[NPL];
        unsigned valid = 0;
#pragma unroll
        for (int s = 0; s < NPL; ++s) {
            const int k = s * T + tid;
            px[s] = py[s] = pz[s] = 0.f;
            tmp[s] = 1e10f;
            key[s] = 0;
            if (k < N) {
                px[s] = sp[k * 3 + 0];
                py[s] = sp[k * 3 + 1];
                pz[s] = sp[k * 3 + 2];
                const float mag = gad_sqnorm(px[s], py[s], pz[s]);
                if (!(mag < 1e-3f)) valid |= 1u << s;              // (the float-vs-double-literal compare: see the fast path)
                key[s] = key_of(k);
            }
        }
        for (int j = 1; j < M; ++j) {
            const float x1 = sp[old * 3 + 0], y1 = sp[old * 3 + 1], z1 = sp[old * 3 + 2];
            unsigned long long best = 0ull;
#pragma unroll
            for (int s = 0; s < NPL; ++s) {
                if (valid & (1u << s)) {
                    const float d = gad_sqdist(px[s], py[s], pz[s], x1, y1, z1);
                    const float d2 = d < tmp[s] ? d : tmp[s];
                    tmp[s] = d2;
                    const unsigned long long c = fps_pack(d2, key[s]);
                    best = c > best ? c : best;
                }
            }
            old = publish(j, fps_wave_max(best));
        }
    }
    __syncthreads();
    for (int j = tid; j < M; j += T) {
        const int pick = pk[j];
        idx[(size_t)b * M + j] = pick;
        if (new_xyz) {
            float* o = new_xyz + ((size_t)b * M + j) * 3;
            o[0] = sp[pick * 3 + 0]; o[1] = sp[pick * 3 + 1]; o[2] = sp[pick * 3 + 2];
        }
    }
}

static int g_opt_fps_cfg = 0;     // points per thread x wavefronts for 1024 < N <= 4096: 0 = 16 x 4, 1 = 8 x 8, 2 = 4 x 16 (A/B)
static int fps_tie_bits(int n) {  // log2 of upstream opt_n_threads(n) = pow2 <= min(n, 512)
    int bits = 0;
    while ((2 << bits) <= n && (2 << bits) <= 512) ++bits;
    return bits;
}

extern "C" int gad_furthest_point_sampling(const float* xyz, int B, int N, int M, int32_t* idx,
                                           float* new_xyz, void* stream) {
    GAD_REQUIRE(xyz && idx, GAD_ERR_NULL, "fps: null pointer");
    GAD_REQUIRE(B >= 0 && N >= 1 && M >= 0 && N <= 16384 && N <= 65535, GAD_ERR_SHAPE, "fps: unsupported shape B=%d N=%d M=%d", B, N, M);
    // (upstream samples without replacement from N points: npoint > N repeats index 0 there; the pick buffer below is sized by M)
    GAD_REQUIRE(M <= N, GAD_ERR_SHAPE, "fps: M=%d picks from N=%d points", M, N);
    if (B == 0 || M == 0) return GAD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int tie = fps_tie_bits(N);
    const size_t lds = (size_t)(((N * 3 + 3) & ~3) + 64 + M) * sizeof(float);      // coordinates, exchange slots, picks
    // one workgroup holds its whole cloud in LDS: 160 KB per CU bounds N (N = M: 10 220 points); beyond 64 KB the kernel's dynamic
    // LDS limit has to be raised first (such a launch used to fail at hipGetLastError)
    GAD_REQUIRE(lds <= 160 * 1024, GAD_ERR_SHAPE, "fps: N=%d, M=%d need %zu bytes of LDS per cloud (the CU has 163840)", N, M, lds);
#define FPS_LAUNCH(PPL, WV)                                                                                                   \
    do {                                                                                                                      \
        auto kern = fps_kernel<PPL, WV>;                                                                                      \
        if (lds > 64 * 1024 &&                                                                                                \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            gad_set_error("fps: cannot raise the dynamic LDS limit to %zu bytes", lds);                                       \
            return GAD_ERR_LAUNCH;                                                                                            \
        }                                                                                                                     \
        hipLaunchKernelGGL(kern, dim3(B), dim3(64 * WV), lds, st, xyz, N, M, tie, idx, new_xyz);                              \
    } while (0)
    if (N <= 64) {
        FPS_LAUNCH(1, 1);
    } else if (N <= 1024) {
        if (g_opt_fps_cfg == 3) FPS_LAUNCH(16, 1);
        else if (g_opt_fps_cfg == 4) FPS_LAUNCH(8, 2);
        else FPS_LAUNCH(4, 4);
    } else if (N <= 4096) {
        if (g_opt_fps_cfg == 3) FPS_LAUNCH(64, 1);
        else if (g_opt_fps_cfg == 4) FPS_LAUNCH(32, 2);
        else if (g_opt_fps_cfg == 1) FPS_LAUNCH(8, 8);
        else if (g_opt_fps_cfg == 2) FPS_LAUNCH(4, 16);
        else FPS_LAUNCH(16, 4);
    } else {
        FPS_LAUNCH(16, 16);
    }
#undef FPS_LAUNCH
    GAD_CHECK_LAUNCH("fps");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// ball query: one wavefront per centroid, ballot + prefix popcount keeps ascending index order
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_ball_query(const float* __restrict__ p, int N, float cx, float cy,
                                               float cz, float r2, int nsample, int lane,
                                               int32_t* __restrict__ out) {
    int cnt = 0, first = 0;
    for (int base = 0; base < N && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool in = false;
        if (k < N) {
            const float d2 = gad_sqdist(cx, cy, cz, p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
            in = d2 < r2;
        }
        const unsigned long long mask = __ballot(in);
        if (mask) {
            if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
            const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (in && slot < nsample) out[slot] = k;
            cnt += __popcll(mask);
        }
    }
    cnt = cnt < nsample ? cnt : nsample;
    for (int s = cnt + lane; s < nsample; s += 64) out[s] = first;   // pad with the first hit (0 if none)
    return cnt;
}

// Radius search for LARGE clouds (N > 1024: BASELINE configs[3], 512 centroids x 4096 points x 128 clouds = 2.7e8
// distance tests).  The one-wavefront-per-centroid scan above is bound by VALU issue: ~15 instructions per 64 tests,
// each lane re-loading its point for every centroid.  Here a workgroup stages its cloud once in LDS (structure of
// arrays) and every wavefront scans it for SIXTEEN centroids at a time, two per packed instruction (v_pk_add/mul_f32
// keep the pinned evaluation order ((dx*dx)+(dy*dy))+(dz*dz) with every operation rounded: fp contraction is off in
// this function), i.e. ~5 instructions per 64 tests instead of ~15.  Results are identical to the scan above (same
// predicate, same ascending compaction).  Measured at configs[3]: ball query 198 -> 134 us, query_and_group 224 -> 179 us;
// the floor of this brute-force formulation is ~40 us of packed arithmetic (a cell list would be the next step).
typedef float gad_f32x2 __attribute__((ext_vector_type(2)));
#define BQ_CPW 16                                  // centroids per wavefront
#define BQ_MAXN 4096                               // points staged in LDS (48 KB)

__device__ __forceinline__ gad_f32x2 bq_sqdist2(gad_f32x2 cx, gad_f32x2 cy, gad_f32x2 cz, float x, float y, float z) {
#pragma clang fp contract(off)
    const gad_f32x2 dx = cx - gad_f32x2{x, x}, dy = cy - gad_f32x2{y, y}, dz = cz - gad_f32x2{z, z};
    const gad_f32x2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// idx rows + counts of the 4 * BQ_CPW centroids [m0, m0 + 4 * BQ_CPW) of one cloud; lists: LDS, 4 * BQ_CPW x nsample ints
__device__ __forceinline__ void bq_tile_scan(const float* __restrict__ xs, const float* __restrict__ ys,
                                             const float* __restrict__ zs, int N, const float* __restrict__ ctr, int m0,
                                             int M, float r2, int nsample, int32_t* __restrict__ lists,
                                             int (&cnt)[BQ_CPW]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gad_f32x2 cx[BQ_CPW / 2], cy[BQ_CPW / 2], cz[BQ_CPW / 2];
    int first[BQ_CPW];
#pragma unroll
    for (int j = 0; j < BQ_CPW; ++j) {
        const int m = min(m0 + wave * BQ_CPW + j, M - 1);        // clamped: surplus slots repeat the last centroid, not written
        const float* c = ctr + (size_t)m * 3;
        cx[j >> 1][j & 1] = c[0]; cy[j >> 1][j & 1] = c[1]; cz[j >> 1][j & 1] = c[2];
        cnt[j] = 0; first[j] = 0;
    }
    for (int base = 0; base < N; base += 64) {
        const int k = base + lane;
        const bool live = k < N;
        // lanes past the end of the cloud get a far-away point: the plain predicate d2 < r2 is then false for them and the
        // ballot is a single v_cmp (no extra masking per centroid)
        const float x = live ? xs[k] : 3.0e18f, y = live ? ys[k] : 3.0e18f, z = live ? zs[k] : 3.0e18f;
#pragma unroll
        for (int p = 0; p < BQ_CPW / 2; ++p) {
            const gad_f32x2 d2 = bq_sqdist2(cx[p], cy[p], cz[p], x, y, z);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * p + e;
                const bool in = d2[e] < r2;
                const unsigned long long mask = __ballot(in);
                if (mask) {                                   // wave-uniform; ~3 of 4 (centroid, chunk) pairs have no hit
                    if (cnt[j] == 0) first[j] = base + __ffsll((long long)mask) - 1;
                    const int slot = cnt[j] + __popcll(mask & ((1ull << lane) - 1ull));
                    if (in && slot < nsample) lists[(wave * BQ_CPW + j) * nsample + slot] = k;
                    cnt[j] += __popcll(mask);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BQ_CPW; ++j) {
        cnt[j] = cnt[j] < nsample ? cnt[j] : nsample;
        for (int s2 = cnt[j] + lane; s2 < nsample; s2 += 64) lists[(wave * BQ_CPW + j) * nsample + s2] = first[j];
    }
}

// mode 0: ball query only (idx, cnt);  mode 1: QueryAndGroup(use_xyz=True) output as well
__global__ __launch_bounds__(256) void ball_query_tiled_kernel(const float* __restrict__ new_xyz,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ feat, int C, int N, int M,
                                                               float r2, int S, int32_t* __restrict__ idx,
                                                               int32_t* __restrict__ cnt_out, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float bq_lds[];
    float* xs = bq_lds;
    float* ys = xs + N;
    float* zs = ys + N;
    int32_t* lists = reinterpret_cast<int32_t*>(zs + N);          // (4 * BQ_CPW) x S
    const int b = blockIdx.y, m0 = blockIdx.x * 4 * BQ_CPW;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N; i += 256) { xs[i] = p[i * 3 + 0]; ys[i] = p[i * 3 + 1]; zs[i] = p[i * 3 + 2]; }
    __syncthreads();
    int cnt[BQ_CPW];
    bq_tile_scan(xs, ys, zs, N, new_xyz + (size_t)b * M * 3, m0, M, r2, S, lists, cnt);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // a wavefront reads back only its own lists
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)M * S;
#pragma unroll
    for (int j = 0; j < BQ_CPW; ++j) {
        const int m = m0 + wave * BQ_CPW + j;
        if (m >= M) break;                                        // wave-uniform
        const size_t g = (size_t)b * M + m;
        const int32_t* my = lists + (wave * BQ_CPW + j) * S;
        if (lane == 0 && cnt_out) cnt_out[g] = cnt[j];
        const float* c = new_xyz + g * 3;
        const float ccx = c[0], ccy = c[1], ccz = c[2];
        float* o = out ? out + ((size_t)b * (3 + C)) * plane + (size_t)m * S : nullptr;
        for (int s2 = lane; s2 < S; s2 += 64) {
            const int k = my[s2];
            idx[g * S + s2] = k;
            if (o) {
                o[0 * plane + s2] = __fsub_rn(xs[k], ccx);
                o[1 * plane + s2] = __fsub_rn(ys[k], ccy);
                o[2 * plane + s2] = __fsub_rn(zs[k], ccz);
                const float* f = feat + (size_t)b * C * N + k;
                for (int ch = 0; ch < C; ++ch) o[(3 + ch) * plane + s2] = f[(size_t)ch * N];
            }
        }
    }
}

// ---- cell-list radius search (configs[3] regime: thousands of points, a radius that is a small fraction of the cloud) ----
// The brute-force tile scan above tests every (centroid, point) pair: 4096 tests per centroid at configs[3] for ~17 hits.
// Here one 1024-thread workgroup sorts its cloud ONCE into a uniform grid held in LDS (cell edge >= 1.001 x radius, so
// every in-radius point lies in the 3x3x3 cells around the centroid's cell: ~110 candidates), then each wavefront takes
// four centroids at a time (16 lanes each): the candidates of the nine (y,z) cell rows -- three x-adjacent cells are one
// contiguous run of the sorted cloud -- get the exact pinned-order distance test, and hits set bit k of the centroid's
// N-bit LDS bitmap.  The bitmap is then read back in ASCENDING point order (one 64-bit word per lane, popcount prefix
// scan), which is the reference's result order (first nsample hits by index) whatever order the cells were visited in:
// indices, counts and the grouped tensor are bit-identical to the scans above.  Cells are only an acceleration
// structure: a centroid outside the cloud's bounding box is clamped to the border cell, which still covers every
// candidate it can have, and the exact predicate d2 < r2 decides.
#define BQC_THREADS 1024
#define BQC_WAVES 16
#define BQC_GMAX 12                                // cells per axis; 12^3 = 1728
#define BQC_MAXCELLS 1728
#define BQC_MB 256                                 // centroids per workgroup

struct BqcGrid { float lox, loy, loz, ivx, ivy, ivz; int nx, ny, nz; };

// cells along one axis: edge >= h (= 1.001 r).  Only an acceleration structure, so approximate arithmetic is fine as
// long as the edge never drops below the radius: the 1e-3 (1e-4) slack covers the ~1e-7 error of v_rcp_f32
__device__ __forceinline__ int bqc_axis(float lo, float hi, float inv_h, float& inv) {
    const float ext = hi - lo;
    int n = (int)(ext * inv_h) + 1;
    inv = inv_h;
    if (!(n <= BQC_GMAX)) { n = BQC_GMAX; inv = (float)BQC_GMAX * __builtin_amdgcn_rcpf(ext * 1.0001f); }   // coarser cells
    return n < 1 ? 1 : n;
}
__device__ __forceinline__ int bqc_cell1(float p, float lo, float inv, int n) {
    const int c = (int)floorf((p - lo) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// DPP cross-lane helpers (row = 16 lanes): no LDS traffic, unlike __shfl (ds_bpermute)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int row_incl_scan(int x) {             // inclusive prefix sum inside each 16-lane row
    x += dpp_i<0x111>(0, x); x += dpp_i<0x112>(0, x); x += dpp_i<0x114>(0, x); x += dpp_i<0x118>(0, x);   // row_shr:1,2,4,8
    return x;
}
__device__ __forceinline__ int wave_incl_scan(int x) {            // inclusive prefix sum over the wavefront
    x = row_incl_scan(x);
    x += dpp_i<0x142, 0xa>(0, x);                                  // row_bcast:15 -> rows 1, 3
    x += dpp_i<0x143, 0xc>(0, x);                                  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ float row_max_f(float v) {             // lane 15 of each row: the row maximum
    v = fmaxf(v, __int_as_float(dpp_i<0x111>(__float_as_int(v), __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x112>(__float_as_int(v), __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x114>(__float_as_int(v), __float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<0x118>(__float_as_int(v), __float_as_int(v))));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {            // every lane gets the wavefront maximum
    const int i = __float_as_int(row_max_f(v));
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 15)), __int_as_float(__builtin_amdgcn_readlane(i, 31))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 47)), __int_as_float(__builtin_amdgcn_readlane(i, 63))));
}

#define BQC_QUADS (BQC_MB / (4 * BQC_WAVES))       // passes of four centroids per wavefront
#define BQC_CPIPE 4                                // feature channels the grouped-output phase keeps in flight per lane
// LDS map (bytes; the cloud arrays are sized for BQ_MAXN so that every offset is an instruction immediate)
#define BQC_L_SY 16384
#define BQC_L_SZ 32768
#define BQC_L_SIDX 49152                           // u16[4096]  sorted slot -> point index
#define BQC_L_SPOS 57344                           // u16[4096]  point index -> sorted slot
#define BQC_L_BM 65536                             // u64[waves][4][64]  per-centroid bitmaps
#define BQC_L_TAB 98304                            // u32[waves][16][9]  (candidate run: first sorted slot | length << 16)
#define BQC_L_CSTART 107520                        // i32[BQC_MAXCELLS + 4]
#define BQC_L_RED 114448                           // f32[128]
#define BQC_L_LISTS 114960                         // i32[waves][4][SP]; the build's scatter cursors live here first

// SP: list stride (nsample <= SP).  The workgroup's 256 centroids: wavefront w takes quads q = w + 16 qi (qi < 4) of
// four consecutive centroids; lane group sub = lane / 16 works on centroid 4q + sub.
template <int SP>
__global__ __launch_bounds__(BQC_THREADS) void ball_query_cells_kernel(const float* __restrict__ new_xyz,
                                                                       const float* __restrict__ xyz,
                                                                       const float* __restrict__ feat, int C, int N, int M,
                                                                       float radius, int S, int32_t* __restrict__ idx,
                                                                       int32_t* __restrict__ cnt_out, float* __restrict__ out, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float bq_lds[];
    char* const L = reinterpret_cast<char*>(bq_lds);
    float* const sx = reinterpret_cast<float*>(L);
    float* const sy = reinterpret_cast<float*>(L + BQC_L_SY);
    float* const sz = reinterpret_cast<float*>(L + BQC_L_SZ);
    unsigned short* const sidx = reinterpret_cast<unsigned short*>(L + BQC_L_SIDX);
    unsigned short* const spos = reinterpret_cast<unsigned short*>(L + BQC_L_SPOS);
    unsigned long long* const bm = reinterpret_cast<unsigned long long*>(L + BQC_L_BM);
    unsigned* const tab = reinterpret_cast<unsigned*>(L + BQC_L_TAB);
    int32_t* const cstart = reinterpret_cast<int32_t*>(L + BQC_L_CSTART);
    float* const red = reinterpret_cast<float*>(L + BQC_L_RED);
    int32_t* const lists = reinterpret_cast<int32_t*>(L + BQC_L_LISTS);
    int32_t* const cursor = lists;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int sub = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.y, m0 = blockIdx.x * BQC_MB;
    const int m_end = min(M, m0 + BQC_MB);
    const float* p = xyz + (size_t)b * N * 3;

    // lane c16 = lane % 16 prepares centroid 4 * (wave + 16 * (c16 / 4)) + c16 % 4 (the four lane rows in unison); its
    // coordinates are fetched now so that the latency hides behind the build
    const int pm = m0 + 4 * (wave + BQC_WAVES * (l16 >> 2)) + (l16 & 3);
    float pcx, pcy, pcz;
    {
        const float* c = new_xyz + ((size_t)b * M + min(pm, M - 1)) * 3;
        pcx = c[0]; pcy = c[1]; pcz = c[2];
    }
    if (dbg & 8) return;
    // ---- build: bounding box -> grid -> counting sort -------------------------------------------------------------
    float px[4], py[4], pz[4];
    float hi[6] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};          // max of (x, y, z, -x, -y, -z)
    // thread t owns points 4t .. 4t+3: three 16-byte loads when the cloud is 16-byte aligned (N % 4 == 0)
    const bool vec = (N & 3) == 0 && (reinterpret_cast<size_t>(xyz) & 15) == 0;
    if (vec) {
        if (4 * tid < N) {
            const float4* p4 = reinterpret_cast<const float4*>(p) + 3 * tid;
            const float4 a = p4[0], bq = p4[1], cq = p4[2];
            px[0] = a.x; py[0] = a.y; pz[0] = a.z; px[1] = a.w; py[1] = bq.x; pz[1] = bq.y;
            px[2] = bq.z; py[2] = bq.w; pz[2] = cq.x; px[3] = cq.y; py[3] = cq.z; pz[3] = cq.w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * tid + u;
            if (k < N) { px[u] = p[k * 3 + 0]; py[u] = p[k * 3 + 1]; pz[u] = p[k * 3 + 2]; }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (4 * tid + u < N) {
            hi[0] = fmaxf(hi[0], px[u]); hi[1] = fmaxf(hi[1], py[u]); hi[2] = fmaxf(hi[2], pz[u]);
            hi[3] = fmaxf(hi[3], -px[u]); hi[4] = fmaxf(hi[4], -py[u]); hi[5] = fmaxf(hi[5], -pz[u]);
        }
    }
    for (int c = tid; c < BQC_MAXCELLS + 4; c += BQC_THREADS) cstart[c] = 0;
    for (int i = tid; i < BQC_WAVES * 4 * 64; i += BQC_THREADS) bm[i] = 0ull;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        hi[a] = wave_max_f(hi[a]);
        if (lane == 0) red[a * BQC_WAVES + wave] = hi[a];
    }
    __syncthreads();
    if (dbg & 16) return;
#pragma unroll
    for (int a = 0; a < 6; ++a)                                    // 16 per-wavefront values = one lane row
        hi[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row_max_f(red[a * BQC_WAVES + l16])), 15));
    BqcGrid g;
    const float inv_h = __builtin_amdgcn_rcpf(radius * 1.001f);
    g.lox = -hi[3]; g.loy = -hi[4]; g.loz = -hi[5];
    g.nx = bqc_axis(g.lox, hi[0], inv_h, g.ivx);
    g.ny = bqc_axis(g.loy, hi[1], inv_h, g.ivy);
    g.nz = bqc_axis(g.loz, hi[2], inv_h, g.ivz);
    const int ncell = g.nx * g.ny * g.nz;

    int cell[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = 4 * tid + u;
        cell[u] = 0;
        if (k < N) {
            cell[u] = (bqc_cell1(pz[u], g.loz, g.ivz, g.nz) * g.ny + bqc_cell1(py[u], g.loy, g.ivy, g.ny)) * g.nx +
                      bqc_cell1(px[u], g.lox, g.ivx, g.nx);
            atomicAdd(&cstart[cell[u] + 1], 1);
        }
    }
    __syncthreads();
    if (dbg & 32) return;
    {   // inclusive scan of cstart[0 .. ncell] in place: two entries per thread (2048 >= BQC_MAXCELLS + 1)
        int32_t* wtot = reinterpret_cast<int32_t*>(red) + 6 * BQC_WAVES;
        const int e0 = 2 * tid;
        const int2 a = *reinterpret_cast<const int2*>(cstart + min(e0, BQC_MAXCELLS + 2));
        const int a0 = e0 <= ncell ? a.x : 0, a1 = e0 + 1 <= ncell ? a.y : 0;
        const int incl = wave_incl_scan(a0 + a1);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        const int winc = row_incl_scan(wtot[l16]);                 // 16 wavefront totals = one lane row
        const int base = wave == 0 ? 0 : __shfl(winc, wave - 1, 64);
        const int excl = base + incl - (a0 + a1);
        if (e0 <= ncell) { cstart[e0] = excl + a0; cursor[e0] = excl + a0; }
        if (e0 + 1 <= ncell) { cstart[e0 + 1] = excl + a0 + a1; cursor[e0 + 1] = excl + a0 + a1; }
    }
    __syncthreads();
    if (dbg & 64) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = 4 * tid + u;
        if (k < N) {
            const int t = atomicAdd(&cursor[cell[u]], 1);
            sx[t] = px[u]; sy[t] = py[u]; sz[t] = pz[u];
            sidx[t] = (unsigned short)k; spos[k] = (unsigned short)t;
        }
    }
    __syncthreads();

    // ---- candidate runs of this wavefront's 16 centroids: nine (y, z) cell rows, three x-adjacent cells = one run ----
    {
        const int cx = bqc_cell1(pcx, g.lox, g.ivx, g.nx), cy = bqc_cell1(pcy, g.loy, g.ivy, g.ny),
                  cz = bqc_cell1(pcz, g.loz, g.ivz, g.nz);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1) + 1;
        const int nyx = g.ny * g.nx, rb0 = (cz * g.ny + cy) * g.nx;
        unsigned* te = tab + (wave * 16 + l16) * 9;
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int dy = (rr % 3) - 1, dz = (rr / 3) - 1;
            const bool ok = pm < m_end && (unsigned)(cy + dy) < (unsigned)g.ny && (unsigned)(cz + dz) < (unsigned)g.nz;
            const int rb = ok ? rb0 + dz * nyx + dy * g.nx : 0;
            const int beg = cstart[rb + x0], end = cstart[rb + x1];
            if (sub == 0) te[rr] = ok ? (unsigned)beg | ((unsigned)(end - beg) << 16) : 0u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // from here on a wavefront touches only its own LDS rows
    __builtin_amdgcn_wave_barrier();

    // ---- query: four centroids per wavefront pass, 16 lanes each ----------------------------------------------------
    const float r2 = radius * radius;
    // the centroid of (pass qi, lane group sub) was prepared by lane 4 qi + sub
    float qx[BQC_QUADS], qy[BQC_QUADS], qz[BQC_QUADS];
#pragma unroll
    for (int qi = 0; qi < BQC_QUADS; ++qi) {
        qx[qi] = __shfl(pcx, 4 * qi + sub, 64); qy[qi] = __shfl(pcy, 4 * qi + sub, 64); qz[qi] = __shfl(pcz, 4 * qi + sub, 64);
    }
    const unsigned bmrow = BQC_L_BM + (unsigned)(wave * 4 + sub) * 512u;        // LDS byte address of this lane group's bitmap
    unsigned long long* const bw = bm + (size_t)(wave * 4 + sub) * 64;
    int32_t* const mylist = lists + (size_t)(wave * 4 + sub) * SP;
    const size_t plane = (size_t)M * S;
    const bool pipe = SP == 64 && C <= BQC_CPIPE;
    float pf[4][BQC_CPIPE];
    int pend_q = -1;
    // candidate slot t of the sorted cloud: exact test, hit -> bit k of the bitmap (a miss ORs in 0; slots past the end of a
    // run read whatever follows -- in bounds of the LDS allocation -- and are masked by `valid`)
    auto mark = [&](unsigned k, bool hit) {
        atomicOr(reinterpret_cast<unsigned*>(L + (bmrow | ((k >> 3) & 0x1fcu))), hit ? 1u << (k & 31) : 0u);
    };
    auto probe = [&](unsigned t, bool valid, float ccx, float ccy, float ccz) {
        const float x = sx[t], y = sy[t], z = sz[t];
        const unsigned k = sidx[t];
        mark(k, valid & (gad_sqdist(ccx, ccy, ccz, x, y, z) < r2));
    };
#pragma unroll
    for (int qi = 0; qi < BQC_QUADS; ++qi) {
        const int q = wave + qi * BQC_WAVES;
        if (m0 + 4 * q >= m_end) break;                            // wave-uniform
        const float ccx = qx[qi], ccy = qy[qi], ccz = qz[qi];
        if (!(dbg & 2)) {
            const unsigned* te = tab + (wave * 16 + qi * 4 + sub) * 9;
            const unsigned short* te16 = reinterpret_cast<const unsigned short*>(te);
            unsigned beg[9], len[9];
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) { beg[rr] = te16[2 * rr]; len[rr] = te16[2 * rr + 1]; }
            // the first 16 slots of all nine runs (a run holds ~12 candidates at configs[3]): every LDS read is issued
            // before the first bitmap atomic, which the compiler will not move loads across
            float tx[9], ty[9], tz[9];
            unsigned tk[9];
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) {
                const unsigned t = beg[rr] + (unsigned)l16;
                tx[rr] = sx[t]; ty[rr] = sy[t]; tz[rr] = sz[t]; tk[rr] = sidx[t];
            }
#pragma unroll
            for (int rr = 0; rr < 9; ++rr)
                mark(tk[rr], ((unsigned)l16 < len[rr]) & (gad_sqdist(ccx, ccy, ccz, tx[rr], ty[rr], tz[rr]) < r2));
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) {
                if (__builtin_amdgcn_ballot_w64(len[rr] > 16u) != 0ull)                // wave-uniform
                    for (unsigned off = 16; __builtin_amdgcn_ballot_w64(off < len[rr]) != 0ull; off += 16)
                        probe(beg[rr] + (unsigned)l16 + off, off + (unsigned)l16 < len[rr], ccx, ccy, ccz);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (dbg & 4) continue;

        // ---- bitmaps -> ascending index lists, the four centroids side by side: lane l16 owns bits [256 l16, 256 l16 + 256)
        unsigned long long w[4];
        {
            const ulonglong2 w01 = *reinterpret_cast<const ulonglong2*>(bw + 4 * l16);
            const ulonglong2 w23 = *reinterpret_cast<const ulonglong2*>(bw + 4 * l16 + 2);
            w[0] = w01.x; w[1] = w01.y; w[2] = w23.x; w[3] = w23.y;
            *reinterpret_cast<ulonglong2*>(bw + 4 * l16) = ulonglong2{0ull, 0ull};
            *reinterpret_cast<ulonglong2*>(bw + 4 * l16 + 2) = ulonglong2{0ull, 0ull};
        }
        const int nb = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
        const int incl = row_incl_scan(nb);
        const int total = __shfl(incl, lane | 15, 64);
        int pos = incl - nb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned long long ww = w[i];
            while (ww != 0ull && pos < S) {
                mylist[pos++] = 256 * l16 + 64 * i + (__ffsll((long long)ww) - 1);
                ww &= ww - 1ull;
            }
        }
        const int cnt = total < S ? total : S;
        if (cnt < S) {                                             // pad with the first hit (0 if the ball is empty)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int first = total > 0 ? mylist[0] : 0;
            for (int s2 = cnt + l16; s2 < S; s2 += 16) mylist[s2] = first;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- write-out: one centroid per 64-lane sweep.  Fast path (nsample <= 64, <= BQC_CPIPE feature channels): the
        // feature loads of this pass stay in flight across the NEXT pass's scan and are stored after it
        if (cnt_out && l16 == 0 && m0 + 4 * q + sub < m_end) cnt_out[(size_t)b * M + m0 + 4 * q + sub] = cnt;
        if (pipe) {
            const bool act = lane < S;
            if (out && pend_q >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                      // phase C of the previous pass
                    const int mj = m0 + 4 * pend_q + j;            // uniform
                    float* oc = out + ((size_t)b * (3 + C) + 3) * plane + (size_t)mj * S;
                    if (mj < m_end) {
#pragma unroll
                        for (int ch = 0; ch < BQC_CPIPE; ++ch) {
                            if (ch < C && act) oc[lane] = pf[j][ch];
                            oc += plane;
                        }
                    }
                }
            }
            unsigned kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                          // phase A: indices and the feature loads of all four
                const bool okj = act && (m0 + 4 * q + j < m_end);
                kk[j] = okj ? (unsigned)(lists + (size_t)(wave * 4 + j) * SP)[lane] : 0u;
                if (out) {
                    const float* fc = feat + (size_t)b * C * N;
#pragma unroll
                    for (int ch = 0; ch < BQC_CPIPE; ++ch) {
                        if (ch < C) pf[j][ch] = fc[kk[j]];
                        fc += N;
                    }
                }
            }
            pend_q = q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                          // phase B: indices + recentred coordinates (LDS only)
                const int mj = m0 + 4 * q + j;
                if (mj >= m_end) break;
                const size_t gi = (size_t)b * M + mj;
                if (act) {
                    (idx + gi * S)[lane] = (int)kk[j];
                    if (out) {
                        const float jx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccx), 16 * j));
                        const float jy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccy), 16 * j));
                        const float jz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ccz), 16 * j));
                        float* orow = out + ((size_t)b * (3 + C)) * plane + (size_t)mj * S;
                        const unsigned t = spos[kk[j]];
                        orow[lane] = __fsub_rn(sx[t], jx);
                        (orow + plane)[lane] = __fsub_rn(sy[t], jy);
                        (orow + 2 * plane)[lane] = __fsub_rn(sz[t], jz);
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int mj = m0 + 4 * q + j;
                if (mj >= m_end) break;
                const size_t gi = (size_t)b * M + mj;
                const int32_t* my = lists + (size_t)(wave * 4 + j) * SP;
                const float jx = __shfl(ccx, 16 * j, 64), jy = __shfl(ccy, 16 * j, 64), jz = __shfl(ccz, 16 * j, 64);
                float* o = out ? out + ((size_t)b * (3 + C)) * plane + (size_t)mj * S : nullptr;
                for (int s2 = lane; s2 < S; s2 += 64) {
                    const int k = my[s2];
                    idx[gi * S + s2] = k;
                    if (o) {
                        const int t = spos[k];
                        o[0 * plane + s2] = __fsub_rn(sx[t], jx);
                        o[1 * plane + s2] = __fsub_rn(sy[t], jy);
                        o[2 * plane + s2] = __fsub_rn(sz[t], jz);
                        const float* f = feat + (size_t)b * C * N + k;
                        for (int ch = 0; ch < C; ++ch) o[(3 + ch) * plane + s2] = f[(size_t)ch * N];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                           // the lists are rewritten by the next pass
    }
    if (pipe && out && pend_q >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mj = m0 + 4 * pend_q + j;
            float* oc = out + ((size_t)b * (3 + C) + 3) * plane + (size_t)mj * S;
            if (mj < m_end) {
#pragma unroll
                for (int ch = 0; ch < BQC_CPIPE; ++ch) {
                    if (ch < C && lane < S) oc[lane] = pf[j][ch];
                    oc += plane;
                }
            }
        }
    }
}

static int g_opt_bq_cells = 1;
void gad_geometry_set_option(const char* name, int value, int* found) {
    if (!strcmp(name, "bq_cells")) { g_opt_bq_cells = value; *found = 1; }
    if (!strcmp(name, "fps_cfg")) { g_opt_fps_cfg = value; *found = 1; }
}
static bool bq_use_cells(int N, int nsample, float radius) {
    return g_opt_bq_cells && N > 1024 && N <= BQ_MAXN && radius > 0.f && radius < 1.0e18f && nsample <= 128;
}
static int bq_launch_cells(const float* new_xyz, const float* xyz, const float* feat, int B, int C, int N, int M, float radius,
                           int nsample, int32_t* idx, int32_t* cnt, float* out, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_cells_kernel<64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ball_query_cells_kernel<128>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const dim3 grid(gad_cdiv(M, BQC_MB), B);
    if (nsample <= 64)
        hipLaunchKernelGGL(ball_query_cells_kernel<64>, grid, dim3(BQC_THREADS), BQC_L_LISTS + BQC_WAVES * 4 * 64 * 4, st, new_xyz,
                           xyz, feat, C, N, M, radius, nsample, idx, cnt, out, g_opt_bq_cells);
    else
        hipLaunchKernelGGL(ball_query_cells_kernel<128>, grid, dim3(BQC_THREADS), BQC_L_LISTS + BQC_WAVES * 4 * 128 * 4, st, new_xyz,
                           xyz, feat, C, N, M, radius, nsample, idx, cnt, out, g_opt_bq_cells);
    return 0;
}

static bool bq_use_tiled(int N, int nsample) { return N > 1024 && N <= BQ_MAXN && nsample <= 256; }
static int bq_launch_tiled(const float* new_xyz, const float* xyz, const float* feat, int B, int C, int N, int M, float radius,
                           int nsample, int32_t* idx, int32_t* cnt, float* out, hipStream_t st) {
    const size_t lds = (size_t)3 * N * sizeof(float) + (size_t)4 * BQ_CPW * nsample * sizeof(int32_t);
    hipLaunchKernelGGL(ball_query_tiled_kernel, dim3(gad_cdiv(M, 4 * BQ_CPW), B), dim3(256), lds, st, new_xyz, xyz, feat, C, N, M,
                       radius * radius, nsample, idx, cnt, out);
    return 0;
}

__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz, int G, int N,
                                                         int M, float r2, int nsample,
                                                         int32_t* __restrict__ idx,
                                                         int32_t* __restrict__ cnt_out) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const int b = g / M;
    const float* c = new_xyz + (size_t)g * 3;
    const int cnt = wave_ball_query(xyz + (size_t)b * N * 3, N, c[0], c[1], c[2], r2, nsample, lane,
                                    idx + (size_t)g * nsample);
    if (lane == 0 && cnt_out) cnt_out[g] = cnt;
}

extern "C" int gad_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius,
                              int nsample, int32_t* idx, int32_t* cnt, void* stream) {
    GAD_REQUIRE(new_xyz && xyz && idx, GAD_ERR_NULL, "ball_query: null pointer");
    GAD_REQUIRE(B >= 0 && N >= 1 && M >= 0 && nsample >= 1, GAD_ERR_SHAPE, "ball_query: bad shape");
    const int G = B * M;
    if (G == 0) return GAD_OK;
    if (bq_use_cells(N, nsample, radius)) {
        bq_launch_cells(new_xyz, xyz, nullptr, B, 0, N, M, radius, nsample, idx, cnt, nullptr, (hipStream_t)stream);
        GAD_CHECK_LAUNCH("ball_query(cells)");
        return GAD_OK;
    }
    if (bq_use_tiled(N, nsample)) {
        bq_launch_tiled(new_xyz, xyz, nullptr, B, 0, N, M, radius, nsample, idx, cnt, nullptr, (hipStream_t)stream);
        GAD_CHECK_LAUNCH("ball_query(tiled)");
        return GAD_OK;
    }
    hipLaunchKernelGGL(ball_query_kernel, dim3(gad_cdiv(G, 4)), dim3(256), 0, (hipStream_t)stream, new_xyz,
                       xyz, G, N, M, radius * radius, nsample, idx, cnt);
    GAD_CHECK_LAUNCH("ball_query");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// materialising group / gather operators (reference-API layouts, channel-major)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ pts,
                                                           const int32_t* __restrict__ idx, int C, int N,
                                                           int MS, long long total,
                                                           float* __restrict__ out) {
    // 4 consecutive outputs per thread: one 16-B idx load, one 16-B store
    const long long q4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q4 >= total) return;
    const long long bc = q4 / MS;            // (b*C + c); MS % 4 == 0 is guaranteed by the host
    const int ms = (int)(q4 - bc * MS);
    const int b = (int)(bc / C);
    const int4 ii = *reinterpret_cast<const int4*>(idx + (size_t)b * MS + ms);
    const float* src = pts + (size_t)bc * N;
    float4 v = make_float4(src[ii.x], src[ii.y], src[ii.z], src[ii.w]);
    *reinterpret_cast<float4*>(out + q4) = v;
}

__global__ __launch_bounds__(256) void group_points_scalar_kernel(const float* __restrict__ pts,
                                                                  const int32_t* __restrict__ idx, int C,
                                                                  int N, int MS, long long total,
                                                                  float* __restrict__ out) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const long long bc = q / MS;
    const int ms = (int)(q - bc * MS);
    const int b = (int)(bc / C);
    out[q] = pts[(size_t)bc * N + idx[(size_t)b * MS + ms]];
}

extern "C" int gad_group_points(const float* points, const int32_t* idx, int B, int C, int N, int M, int S,
                                float* out, void* stream) {
    GAD_REQUIRE(points && idx && out, GAD_ERR_NULL, "group_points: null pointer");
    const long long total = (long long)B * C * M * S;
    if (total == 0) return GAD_OK;
    const int MS = M * S;
    if (MS % 4 == 0) {
        hipLaunchKernelGGL(group_points_kernel, dim3(gad_cdiv(total / 4, 256)), dim3(256), 0,
                           (hipStream_t)stream, points, idx, C, N, MS, total, out);
    } else {
        hipLaunchKernelGGL(group_points_scalar_kernel, dim3(gad_cdiv(total, 256)), dim3(256), 0,
                           (hipStream_t)stream, points, idx, C, N, MS, total, out);
    }
    GAD_CHECK_LAUNCH("group_points");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(const float* __restrict__ go,
                                                                const int32_t* __restrict__ idx, int C,
                                                                int N, int MS, long long total,
                                                                float* __restrict__ gp) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const long long bc = q / MS;
    const int ms = (int)(q - bc * MS);
    const int b = (int)(bc / C);
    atomic_add_f32(gp + (size_t)bc * N + idx[(size_t)b * MS + ms], go[q]);
}

extern "C" int gad_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                     int S, float* grad_points, void* stream) {
    GAD_REQUIRE(grad_out && idx && grad_points, GAD_ERR_NULL, "group_points_grad: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if ((long long)B * C * N > 0) { hipError_t me = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)B * C * N, st); (void)me; }
    const long long total = (long long)B * C * M * S;
    if (total == 0) return GAD_OK;
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(gad_cdiv(total, 256)), dim3(256), 0, st, grad_out, idx,
                       C, N, M * S, total, grad_points);
    GAD_CHECK_LAUNCH("group_points_grad");
    return GAD_OK;
}

extern "C" int gad_gather_points(const float* points, const int32_t* idx, int B, int C, int N, int M,
                                 float* out, void* stream) {
    // gather == group with S = 1
    return gad_group_points(points, idx, B, C, N, M, 1, out, stream);
}
extern "C" int gad_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                      float* grad_points, void* stream) {
    return gad_group_points_grad(grad_out, idx, B, C, N, M, 1, grad_points, stream);
}

// QueryAndGroup(use_xyz=True) in one pass: a wavefront owns a centroid, keeps its idx row in LDS,
// then streams the (3+C) grouped channels out as contiguous S-float runs.
__global__ __launch_bounds__(256) void query_and_group_kernel(const float* __restrict__ new_xyz,
                                                              const float* __restrict__ xyz,
                                                              const float* __restrict__ feat, int G, int C,
                                                              int N, int M, float r2, int S,
                                                              int32_t* __restrict__ idx,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) int32_t sidx[];   // 4 waves x S
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + w;
    if (g >= G) return;
    const int b = g / M, m = g - b * M;
    const float* c = new_xyz + (size_t)g * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* my = sidx + w * S;
    wave_ball_query(p, N, cx, cy, cz, r2, S, lane, my);
    // LDS writes of this wave are read back by the same wave only
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    int32_t* gi = idx + (size_t)g * S;
    const size_t plane = (size_t)M * S;
    float* o = out + ((size_t)b * (3 + C)) * plane + (size_t)m * S;
    for (int s = lane; s < S; s += 64) {
        const int k = my[s];
        gi[s] = k;
        o[0 * plane + s] = __fsub_rn(p[k * 3 + 0], cx);
        o[1 * plane + s] = __fsub_rn(p[k * 3 + 1], cy);
        o[2 * plane + s] = __fsub_rn(p[k * 3 + 2], cz);
        const float* f = feat + (size_t)b * C * N + k;
        for (int ch = 0; ch < C; ++ch) o[(3 + ch) * plane + s] = f[(size_t)ch * N];
    }
}

extern "C" int gad_query_and_group(const float* new_xyz, const float* xyz, const float* features, int B,
                                   int C, int N, int M, float radius, int nsample, int32_t* idx, float* out,
                                   void* stream) {
    GAD_REQUIRE(new_xyz && xyz && idx && out && (features || C == 0), GAD_ERR_NULL, "query_and_group: null pointer");
    GAD_REQUIRE(nsample >= 1 && nsample <= 4096, GAD_ERR_SHAPE, "query_and_group: nsample out of range");
    const int G = B * M;
    if (G == 0) return GAD_OK;
    if (bq_use_cells(N, nsample, radius)) {
        bq_launch_cells(new_xyz, xyz, features, B, C, N, M, radius, nsample, idx, nullptr, out, (hipStream_t)stream);
        GAD_CHECK_LAUNCH("query_and_group(cells)");
        return GAD_OK;
    }
    if (bq_use_tiled(N, nsample)) {
        bq_launch_tiled(new_xyz, xyz, features, B, C, N, M, radius, nsample, idx, nullptr, out, (hipStream_t)stream);
        GAD_CHECK_LAUNCH("query_and_group(tiled)");
        return GAD_OK;
    }
    hipLaunchKernelGGL(query_and_group_kernel, dim3(gad_cdiv(G, 4)), dim3(256), sizeof(int32_t) * 4 * nsample,
                       (hipStream_t)stream, new_xyz, xyz, features, G, C, N, M, radius * radius, nsample, idx,
                       out);
    GAD_CHECK_LAUNCH("query_and_group");
    return GAD_OK;
}

// ------------------------------------------------------------------------------------------------
// fused-path geometry: point layout conversion and row compaction
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_points_kernel(const float* __restrict__ ps, int C4, int NP,
                                                          int skip, int N, long long total,
                                                          float* __restrict__ xyz, float* __restrict__ feat) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int b = (int)(q / N), j = (int)(q - (long long)b * N);
    const float* src = ps + (size_t)b * C4 * NP + skip + j;
    const float x = src[0], y = src[(size_t)NP], z = src[(size_t)2 * NP];
    const float f = C4 > 3 ? src[(size_t)3 * NP] : 0.f;
    xyz[q * 3 + 0] = x; xyz[q * 3 + 1] = y; xyz[q * 3 + 2] = z;
    *reinterpret_cast<float4*>(feat + q * 4) = make_float4(x, y, z, f);
}

extern "C" int gad_prep_points(const float* point_state, int B, int C4, int NP, int skip, float* xyz,
                               float* feat, void* stream) {
    GAD_REQUIRE(point_state && xyz && feat, GAD_ERR_NULL, "prep_points: null pointer");
    GAD_REQUIRE(C4 >= 3 && NP > skip && skip >= 0, GAD_ERR_SHAPE, "prep_points: bad shape");
    const int N = NP - skip;
    const long long total = (long long)B * N;
    if (total == 0) return GAD_OK;
    hipLaunchKernelGGL(prep_points_kernel, dim3(gad_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       point_state, C4, NP, skip, N, total, xyz, feat);
    GAD_CHECK_LAUNCH("prep_points");
    return GAD_OK;
}

// exclusive scan of max(cnt,1) over G groups by one 1024-thread workgroup, 4096 groups per pass: every thread takes four
// consecutive counts (one 16-byte load, coalesced), a wavefront scan by ds_bpermute-free shuffles, the sixteen wavefront totals
// through LDS.  (First version: every thread walked its own G / 1024 consecutive counts -- 256-byte strides between lanes -- and a
// 20-barrier Hillis-Steele pass: 101 us for the 65536 groups of configs[3]'s first module.)
__global__ __launch_bounds__(1024) void rows_scan_kernel(const int32_t* __restrict__ cnt, int G,
                                                         int32_t* __restrict__ off,
                                                         int32_t* __restrict__ n_rows) {
    __shared__ int wtot[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run = 0;                                          // groups before this pass (uniform)
    int pass = 0;
    const bool vec = (G & 3) == 0 && ((reinterpret_cast<uintptr_t>(cnt) | reinterpret_cast<uintptr_t>(off)) & 15) == 0;
    for (int base = 0; base < G; base += 4096, ++pass) {
        const int g0 = base + 4 * tid;
        int c[4];
        if (g0 + 3 < G && vec) {
            const int4 v = *reinterpret_cast<const int4*>(cnt + g0);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = g0 + i < G ? cnt[g0 + i] : 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = g0 + i < G ? max(c[i], 1) : 0;
        const int s = c[0] + c[1] + c[2] + c[3];
        int inc = s;                                      // inclusive scan over the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wtot[pass & 1][wave] = inc;
        __syncthreads();                                  // (the buffer alternates: one barrier per pass)
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = wtot[pass & 1][w];
            before += w < wave ? t : 0;
            total += t;
        }
        int o0 = run + before + inc - s;
        if (g0 + 3 < G && vec) {
            *reinterpret_cast<int4*>(off + g0) = make_int4(o0, o0 + c[0], o0 + c[0] + c[1], o0 + c[0] + c[1] + c[2]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (g0 + i < G) off[g0 + i] = o0; o0 += c[i]; }
        }
        run += total;
    }
    if (tid == 0) { off[G] = run; *n_rows = run; }
}

__global__ __launch_bounds__(256) void rows_fill_kernel(const int32_t* __restrict__ idx,
                                                        const int32_t* __restrict__ cnt,
                                                        const int32_t* __restrict__ off, int G, int M, int Nsrc,
                                                        int nsample, int32_t* __restrict__ row_pt,
                                                        int32_t* __restrict__ row_grp,
                                                        float* __restrict__ row_w) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int lane = threadIdx.x & 63;
    const int n = max(cnt[g], 1);
    const int base = off[g];
    const int b = g / M;
    for (int s = lane; s < n; s += 64) {
        row_pt[base + s] = b * Nsrc + idx[(size_t)g * nsample + s];
        row_grp[base + s] = g;
        row_w[base + s] = s == 0 ? (float)(nsample - n + 1) : 1.f;
    }
}

extern "C" int gad_rows_from_ball_query(const int32_t* idx, const int32_t* cnt, int G, int M, int Nsrc,
                                        int nsample, int32_t* grp_off, int32_t* row_pt, int32_t* row_grp,
                                        float* row_w, int32_t* n_rows, void* stream) {
    GAD_REQUIRE(idx && cnt && grp_off && row_pt && row_grp && row_w && n_rows, GAD_ERR_NULL, "rows: null pointer");
    GAD_REQUIRE(G >= 1 && M >= 1, GAD_ERR_SHAPE, "rows: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, G, grp_off, n_rows);
    hipLaunchKernelGGL(rows_fill_kernel, dim3(gad_cdiv(G, 4)), dim3(256), 0, st, idx, cnt, grp_off, G, M, Nsrc,
                       nsample, row_pt, row_grp, row_w);
    GAD_CHECK_LAUNCH("rows_from_ball_query");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void rows_group_all_kernel(int G, int P, int32_t* __restrict__ off,
                                                             int32_t* __restrict__ row_pt,
                                                             int32_t* __restrict__ row_grp,
                                                             float* __restrict__ row_w,
                                                             int32_t* __restrict__ n_rows) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < G * P) { row_pt[q] = q; row_grp[q] = q / P; row_w[q] = 1.f; }
    if (q <= G) off[q] = q * P;
    if (q == 0) *n_rows = G * P;
}

extern "C" int gad_rows_group_all(int G, int P, int32_t* grp_off, int32_t* row_pt, int32_t* row_grp,
                                  float* row_w, int32_t* n_rows, void* stream) {
    GAD_REQUIRE(grp_off && row_pt && row_grp && row_w && n_rows, GAD_ERR_NULL, "rows_group_all: null pointer");
    hipLaunchKernelGGL(rows_group_all_kernel, dim3(gad_cdiv((long long)G * P + 1, 256)), dim3(256), 0,
                       (hipStream_t)stream, G, P, grp_off, row_pt, row_grp, row_w, n_rows);
    GAD_CHECK_LAUNCH("rows_group_all");
    return GAD_OK;
}
