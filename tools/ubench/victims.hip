// victims for tools/diag_fps_corun.py: kernels that only check that their own state survives while other kernels run on the
// same CUs.  lds: 12 KB of LDS written once, re-read; regs: 48 VGPRs of known values, re-checked; both count changed words.
// build: hipcc -O3 --offload-arch=gfx950 -w -shared -fPIC victims.hip -o libvictims.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void victim_lds(int iters, unsigned* bad) {
    __shared__ unsigned lds[3072];
    for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = 0x51000000u + (unsigned)i;
    __syncthreads();
    unsigned n = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < 3072; i += 256) n += lds[i] != 0x51000000u + (unsigned)i;
        __builtin_amdgcn_s_sleep(8);
    }
    if (n) atomicAdd(bad, n);
}

__global__ __launch_bounds__(256) void victim_regs(int iters, unsigned* bad) {
    unsigned r[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) r[i] = 0x77000000u + threadIdx.x * 64u + (unsigned)i;
    unsigned n = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            asm volatile("" : "+v"(r[i]));          // keep every value in its register across the loop
            n += r[i] != 0x77000000u + threadIdx.x * 64u + (unsigned)i;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (n) atomicAdd(bad, n);
}

// fp: a deterministic float recurrence (packed f32 adds / multiplies, min, a DPP wave maximum every 16 steps), one result per thread
__global__ __launch_bounds__(256) void victim_fp(int iters, float* out) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {0.001f * threadIdx.x + 0.5f, 0.002f * threadIdx.x + 0.25f}, m = {1e10f, 1e10f};
    const f2 c = {0.37f + 0.001f * blockIdx.x, 0.73f};
    float w = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma clang fp contract(off)
        const f2 d = a - c, q = d * d + c * d;
        m[0] = q[0] < m[0] ? q[0] : m[0];
        m[1] = q[1] < m[1] ? q[1] : m[1];
        a = a * 0.999f + q * 0.001f;
        if ((it & 15) == 15) {
            float v = a[0] + a[1];
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) { const float o = __shfl_xor(v, s, 64); v = o > v ? o : v; }
            w += v * 1e-3f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + m[0] + m[1] + w;
}

extern "C" int victim_fp_launch(int blocks, int iters, float* out, void* stream) {
    hipLaunchKernelGGL(victim_fp, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, out);
    return (int)hipGetLastError();
}

// mix: the instruction kinds of the furthest-point-sampling round, each checked against a value known in closed form.
//   bit 0: DPP wave maximum (row_shr 1/2/4/8, row_bcast 15/31) + v_readlane 63;   bit 1: bystander registers (16 floats held across
//   the loop) changed;   bit 2: packed-f32 distance + v_min chain differs from the scalar evaluation;   bit 3: ballot / readlane of
//   a lane-dependent value;   bit 4: LDS broadcast read
template <int RED>
__global__ __launch_bounds__(256) void victim_mix(int iters, unsigned* flags, unsigned* detail) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ float lds[3072];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3072; i += 256) lds[i] = __fmaf_rn(0.001f, (float)i, 0.25f);
    __syncthreads();
    float keep[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) keep[i] = 100.f + tid + 1000.f * i;
    f2 q[2] = {{0.1f + 0.001f * tid, 0.2f}, {0.3f, 0.4f + 0.002f * tid}};
    f2 tmin[2] = {{1e10f, 1e10f}, {1e10f, 1e10f}};
    float smin[4] = {1e10f, 1e10f, 1e10f, 1e10f}, rmin[4] = {1e10f, 1e10f, 1e10f, 1e10f};
    unsigned err = 0;
    for (int it = 0; it < iters; ++it) {
        // (0) DPP wave maximum of v = hash(lane, it): expected by a butterfly over ds_bpermute
        const float v = (float)(((lane * 2654435761u + it * 40503u) >> 8) & 0xffff);
        float d = v, dm;
        if (RED == 0) {
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                         "s_nop 1" : "+v"(d));
            dm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
        } else if (RED == 2) {
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1" : "+v"(d));
            const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 15)), a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 31));
            const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 47)), a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
            dm = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
        } else {
            dm = -1.f;
        }
        float e = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const float o = __shfl_xor(e, s, 64); e = o > e ? o : e; }
        if (RED != 1 && dm != e) err |= 1u;
        // (1) bystanders
#pragma unroll
        for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(keep[i])); if (keep[i] != 100.f + tid + 1000.f * i) err |= 2u; }
        // (2) packed distance update vs scalar
        const float cx = lds[(it * 3) % 3000], cy = lds[(it * 3) % 3000 + 1];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 dx = q[h] - cx, dy = q[h] - cy;
                const f2 dd = dx * dx + dy * dy;
                float r0, r1;
                asm("v_min_f32 %0, %1, %2" : "=v"(r0) : "v"(dd[0]), "v"(tmin[h][0]));
                asm("v_min_f32 %0, %1, %2" : "=v"(r1) : "v"(dd[1]), "v"(tmin[h][1]));
                tmin[h][0] = r0; tmin[h][1] = r1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float sx = q[h][u] - cx, sy = q[h][u] - cy;
                    const float sd = sx * sx + sy * sy;
                    smin[2 * h + u] = sd < smin[2 * h + u] ? sd : smin[2 * h + u];
                }
            }
        }
        if (tmin[0][0] != smin[0] || tmin[0][1] != smin[1] || tmin[1][0] != smin[2] || tmin[1][1] != smin[3]) err |= 4u;
        {   // the same distances from the values the LDS words are KNOWN to hold (no LDS read involved): which side is wrong?
#pragma clang fp contract(off)
            const float ex = __fmaf_rn(0.001f, (float)((it * 3) % 3000), 0.25f), ey = __fmaf_rn(0.001f, (float)((it * 3) % 3000 + 1), 0.25f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sx = q[k >> 1][k & 1] - ex, sy = q[k >> 1][k & 1] - ey;
                const float sd = sx * sx + sy * sy;
                rmin[k] = sd < rmin[k] ? sd : rmin[k];
            }
            if (tmin[0][0] != rmin[0] || tmin[0][1] != rmin[1] || tmin[1][0] != rmin[2] || tmin[1][1] != rmin[3]) err |= 32u;   // packed path wrong
            if (smin[0] != rmin[0] || smin[1] != rmin[1] || smin[2] != rmin[2] || smin[3] != rmin[3]) err |= 64u;               // scalar path wrong
        }
        // (3) ballot / readlane
        const unsigned long long b = __ballot(((lane + it) & 3) == 0);
        unsigned long long eb = 0x1111111111111111ull;
        eb = (eb << ((4 - (it & 3)) & 3)) | (eb >> (64 - ((4 - (it & 3)) & 3)) * (((4 - (it & 3)) & 3) != 0));
        if (b != eb) err |= 8u;
        if (__builtin_amdgcn_readlane(lane * 7 + it, (it * 5) & 63) != ((it * 5) & 63) * 7 + it) err |= 8u;
        // (4) LDS broadcast
        if (cx != __fmaf_rn(0.001f, (float)((it * 3) % 3000), 0.25f)) err |= 16u;
    }
    if (err) { atomicOr(flags, err); atomicAdd(detail + (lane >> 4), 1u); }
}

extern "C" int victim_mix_launch(int blocks, int iters, unsigned* flags, unsigned* detail, void* stream, int red) {
    if (red == 1) hipLaunchKernelGGL(victim_mix<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, flags, detail);
    else if (red == 2) hipLaunchKernelGGL(victim_mix<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, flags, detail);
    else hipLaunchKernelGGL(victim_mix<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, flags, detail);
    return (int)hipGetLastError();
}

// pk: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 against v_add_f32 / v_mul_f32 / v_fma_f32 on the same operands, every iteration.
// counts[op * 8 + half * 4 + row]: mismatching (thread, iteration) pairs; op 0 add, 1 mul, 2 fma; half 0 lo, 1 hi; row = lane / 16
__global__ __launch_bounds__(256) void victim_pk(int iters, unsigned* counts, float* sample) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    f2 a = {0.5f + 0.001f * tid, 1.5f - 0.002f * tid}, b = {0.75f, 1.25f + 0.003f * tid};
    unsigned cnt[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) cnt[i] = 0;
    const int row = lane >> 4;
    for (int it = 0; it < iters; ++it) {
        f2 ra, rm, rf;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(ra) : "v"(a), "v"(b));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(rm) : "v"(a), "v"(b));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(rf) : "v"(a), "v"(b), "v"(ra));
        float sa[2], sm[2], sf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sa[u]) : "v"(a[u]), "v"(b[u]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sm[u]) : "v"(a[u]), "v"(b[u]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sf[u]) : "v"(a[u]), "v"(b[u]), "v"(sa[u]));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ra[u] != sa[u]) { if (!cnt[0 * 8 + u * 4 + row]) { sample[0] = ra[u]; sample[1] = sa[u]; sample[2] = a[u]; sample[3] = b[u]; } cnt[0 * 8 + u * 4 + row]++; }
            if (rm[u] != sm[u]) { if (!cnt[1 * 8 + u * 4 + row]) { sample[4] = rm[u]; sample[5] = sm[u]; sample[6] = a[u]; sample[7] = b[u]; } cnt[1 * 8 + u * 4 + row]++; }
            if (rf[u] != sf[u]) cnt[2 * 8 + u * 4 + row]++;
        }
        a[0] = a[0] * 0.9990234375f + 0.0009765625f; a[1] = a[1] * 0.99951171875f + 0.00048828125f;
        b[0] += 0.0001220703125f; b[1] -= 0.0001220703125f;
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) if (cnt[i]) atomicAdd(counts + i, cnt[i]);
}

extern "C" int victim_pk_launch(int blocks, int iters, unsigned* counts, float* sample, void* stream) {
    hipLaunchKernelGGL(victim_pk, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, counts, sample);
    return (int)hipGetLastError();
}

// ldsuse: an LDS read consumed by the very next VALU instruction after its s_waitcnt.  Variants (counts[var * 4 + row]):
//   0 uniform address (broadcast) -> v_add_f32;  1 per-lane address -> v_add_f32;  2 uniform address, s_nop 7 before the v_add_f32;
//   3 uniform ds_read2_b32 -> v_pk_add_f32 with op_sel broadcast (the form hipcc emits for float2 - scalar);  4 uniform ds_read_b32 ->
//   v_mov_b32 first, then the add
__global__ __launch_bounds__(256) void victim_ldsuse(int iters, unsigned* counts) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.0f + (float)i;
    __syncthreads();
    unsigned cnt[5] = {0, 0, 0, 0, 0};
    const float c = 0.5f;
    for (int it = 0; it < iters; ++it) {
        const int iu = (it * 7) & 2047;                 // uniform index (changes every iteration: a stale register holds another value)
        const int il = (iu + lane * 3) & 4095;          // per-lane index
        const unsigned au = (unsigned)(uintptr_t)(lds + iu) & 0xffffu, al = (unsigned)(uintptr_t)(lds + il) & 0xffffu;
        float l0, r0, l1, r1, l2, r2, l4, r4, t4;
        asm volatile("ds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %1, %0, %3" : "=&v"(l0), "=v"(r0) : "v"(au), "v"(c) : "memory");
        asm volatile("ds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %1, %0, %3" : "=&v"(l1), "=v"(r1) : "v"(al), "v"(c) : "memory");
        asm volatile("ds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\tv_add_f32 %1, %0, %3" : "=&v"(l2), "=v"(r2) : "v"(au), "v"(c) : "memory");
        f2 l3, r3, q = {10.f, 20.f};
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %3, %0 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                     : "=&v"(l3), "=v"(r3) : "v"(au), "v"(q) : "memory");
        asm volatile("ds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %2, %0\n\tv_add_f32 %1, %2, %4" : "=&v"(l4), "=v"(r4), "=&v"(t4) : "v"(au), "v"(c) : "memory");
        const float eu = 1.0f + (float)iu, el = 1.0f + (float)il;
        cnt[0] += r0 != eu + c; cnt[1] += r1 != el + c; cnt[2] += r2 != eu + c;
        cnt[3] += (r3[0] != 10.f - eu) || (r3[1] != 20.f - eu);
        cnt[4] += r4 != eu + c;
    }
#pragma unroll
    for (int v = 0; v < 5; ++v) if (cnt[v]) atomicAdd(counts + v * 4 + row, cnt[v]);
}

extern "C" int victim_ldsuse_launch(int blocks, int iters, unsigned* counts, void* stream) {
    hipLaunchKernelGGL(victim_ldsuse, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, counts);
    return (int)hipGetLastError();
}

// ldsuse2: the compiler's form of the failing sequence.  counts[var * 4 + row]:
//   0 ds_read2_b32 whose destination pair starts at its own address register, consumed by v_pk_add_f32 right after s_waitcnt
//   1 the same with a separate address register;  2 separate address, eight independent VALU instructions between the read and
//   its s_waitcnt;  3 as 2 plus s_nop 3 after the s_waitcnt;  4 as 0 with the consumer a plain v_sub_f32
__global__ __launch_bounds__(256) void victim_ldsuse2(int iters, unsigned* counts) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.0f + (float)i;
    __syncthreads();
    unsigned cnt[5] = {0, 0, 0, 0, 0};
    f2 q = {10.f + tid, 20.f + tid};
    float junk = (float)tid;
    for (int it = 0; it < iters; ++it) {
        const int iu = (it * 7) & 2047;
        const unsigned au = (unsigned)(uintptr_t)(lds + iu) & 0xffffu;
        const float eu = 1.0f + (float)iu;
        f2 r0, r1, r2, r3;
        float r4;
        asm volatile("v_mov_b32 v100, %1\n\tds_read2_b32 v[100:101], v100 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %2, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r0) : "s"(au), "v"(q) : "v100", "v101", "memory");
        asm volatile("v_mov_b32 v102, %1\n\tds_read2_b32 v[100:101], v102 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %2, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r1) : "s"(au), "v"(q) : "v100", "v101", "v102", "memory");
        asm volatile("v_mov_b32 v102, %1\n\tds_read2_b32 v[100:101], v102 offset1:1\n\t"
                     "v_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\tv_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\t"
                     "v_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\tv_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %2, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "s"(au), "v"(q), "v"(junk) : "v100", "v101", "v102", "memory");
        asm volatile("v_mov_b32 v102, %1\n\tds_read2_b32 v[100:101], v102 offset1:1\n\t"
                     "v_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\tv_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\t"
                     "v_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\tv_add_f32 %3, 1.0, %3\n\tv_mul_f32 %3, 0.5, %3\n\t"
                     "s_waitcnt lgkmcnt(0)\n\ts_nop 3\n\t"
                     "v_pk_add_f32 %0, %2, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r3) : "s"(au), "v"(q), "v"(junk) : "v100", "v101", "v102", "memory");
        asm volatile("v_mov_b32 v100, %1\n\tds_read2_b32 v[100:101], v100 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_sub_f32 %0, %2, v100" : "=v"(r4) : "s"(au), "v"(q[0]) : "v100", "v101", "memory");
        cnt[0] += (r0[0] != q[0] - eu) || (r0[1] != q[1] - eu);
        cnt[1] += (r1[0] != q[0] - eu) || (r1[1] != q[1] - eu);
        cnt[2] += (r2[0] != q[0] - eu) || (r2[1] != q[1] - eu);
        cnt[3] += (r3[0] != q[0] - eu) || (r3[1] != q[1] - eu);
        cnt[4] += r4 != q[0] - eu;
    }
#pragma unroll
    for (int v = 0; v < 5; ++v) if (cnt[v]) atomicAdd(counts + v * 4 + row, cnt[v]);
}

extern "C" int victim_ldsuse2_launch(int blocks, int iters, unsigned* counts, void* stream) {
    hipLaunchKernelGGL(victim_ldsuse2, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, counts);
    return (int)hipGetLastError();
}

extern "C" int victim_launch(int kind, int blocks, int iters, unsigned* bad, void* stream) {
    if (kind == 0) hipLaunchKernelGGL(victim_lds, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, bad);
    else hipLaunchKernelGGL(victim_regs, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, bad);
    return (int)hipGetLastError();
}
