// Head-output losses of the update step, value + gradient in one pass (no autograd graph).
// Reference arithmetic replaced: core/loss.py:17-31 (goal_pred_loss, pose_bc_loss) with the pose
// math of core/utils.py:814-958, the TD3 target / masked smooth-L1 of core/ddpg.py:61-88,119-130,
// the actor term of core/ddpg.py:169-177 and the tanh squashing of core/networks.py:353-371.
// One workgroup handles the whole local batch (B rows): the masked means couple all rows.
#include "common.hpp"

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul3(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

// gripper control points (reference core/utils.py:819-824); the goal loss uses them rotated by
// rotZ(pi/2) in float64 then cast to float32 (:826-827), which leaves x' = x*cos(pi/2) ~ 3e-18.
__device__ __forceinline__ V3 control_point(int p, bool rotz) {
    const float sx = (p == 2 || p == 4) ? 0.053f : ((p == 3 || p == 5) ? -0.053f : 0.f);
    const float sz = p < 2 ? 0.f : (p < 4 ? 0.075f : 0.105f);
    if (!rotz) return v3(sx, 0.f, sz);
    return v3((float)((double)sx * 6.123233995736766e-17), -sx, sz);
}

// qrot (core/utils.py:940-958): v + 2*(w*(u x v) + u x (u x v))
__device__ __forceinline__ V3 quat_rot(float w, V3 u, V3 v) {
    const V3 uv = cross3(u, v);
    const V3 uuv = cross3(u, uv);
    return add3(v, mul3(2.f, add3(mul3(w, uv), uuv)));
}

// sum over the 6 control points of sum_xyz |P(q,t) - P(qg,tg)| and its gradient wrt (q, t)
__device__ __forceinline__ float goal_point_loss(const float* q, const float* t, const float* qg, const float* tg,
                                                 float* gq, float* gt) {
    const V3 u = v3(q[1], q[2], q[3]), ug = v3(qg[1], qg[2], qg[3]);
    float loss = 0.f, gw = 0.f;
    V3 gu = v3(0, 0, 0), gtt = v3(0, 0, 0);
    for (int p = 0; p < 6; ++p) {
        const V3 v = control_point(p, true);
        const V3 a = add3(quat_rot(q[0], u, v), v3(t[0], t[1], t[2]));
        const V3 b = add3(quat_rot(qg[0], ug, v), v3(tg[0], tg[1], tg[2]));
        const V3 d = sub3(a, b);
        loss += fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
        const V3 s = v3(sgn(d.x), sgn(d.y), sgn(d.z));
        const V3 uv = cross3(u, v);
        gw += 2.f * dot3(s, uv);
        // d/du of 2w(u x v) + 2 u x (u x v), contracted with s
        gu = add3(gu, mul3(2.f, add3(mul3(q[0], cross3(v, s)), add3(cross3(uv, s), cross3(v, cross3(s, u))))));
        gtt = add3(gtt, s);
    }
    gq[0] = gw; gq[1] = gu.x; gq[2] = gu.y; gq[3] = gu.z;
    gt[0] = gtt.x; gt[1] = gtt.y; gt[2] = gtt.z;
    return loss;
}

// head aux output x (7 raw) -> [normalize(x[:4]), x[4:]]   (F.normalize eps 1e-12)
__device__ __forceinline__ float unit_quat(const float* x, float* q) {
    const float n = fmaxf(sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]), 1e-12f);
    for (int i = 0; i < 4; ++i) q[i] = x[i] / n;
    return n;
}
// gradient through the normalisation: (g - q (q.g)) / n
__device__ __forceinline__ void unit_quat_bwd(const float* q, float n, const float* gq, float* gx) {
    const float d = q[0] * gq[0] + q[1] * gq[1] + q[2] * gq[2] + q[3] * gq[3];
    for (int i = 0; i < 4; ++i) gx[i] = (gq[i] - q[i] * d) / n;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(256) void critic_loss_kernel(const float* __restrict__ out9,
                                                          const float* __restrict__ tgt9,
                                                          const float* __restrict__ reward,
                                                          const float* __restrict__ done,
                                                          const float* __restrict__ perturb,
                                                          const float* __restrict__ ret,
                                                          const float* __restrict__ goal, int B, float gamma,
                                                          int critic_aux, const float* __restrict__ inv_n,
                                                          float* __restrict__ y_out, float* __restrict__ aux_norm,
                                                          float* __restrict__ g9, float* __restrict__ scalars) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float nk = 0.f, ng = 0.f;
    for (int i = tid; i < B; i += 256) { nk += perturb[i] < 1.f ? 1.f : 0.f; ng += ret[i] > 0.f ? 1.f : 0.f; }
    nk = block_sum(nk, red);
    ng = block_sum(ng, red);
    const float inv_k = inv_n ? inv_n[0] : 1.f / nk;             // 0/0 -> NaN like torch's empty mean
    const float inv_g = inv_n ? inv_n[1] : 1.f / (ng * 6.f);
    float lq = 0.f, la = 0.f;
    for (int i = tid; i < B; i += 256) {
        const float* o = out9 + (size_t)i * 9;
        float* g = g9 + (size_t)i * 9;
        const float y = reward[i] + (1.f - done[i]) * gamma * fminf(tgt9[(size_t)i * 9], tgt9[(size_t)i * 9 + 1]);
        y_out[i] = y;
        const bool keep = perturb[i] < 1.f;
        for (int h = 0; h < 2; ++h) {
            const float d = o[h] - y, ad = fabsf(d);
            if (keep) lq += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
            g[h] = keep ? (ad < 1.f ? d : sgn(d)) * inv_k : 0.f;
        }
        float q[4], gq[4], gt[3], gx[4];
        const float n = unit_quat(o + 2, q);
        if (aux_norm) {
            for (int c = 0; c < 4; ++c) aux_norm[(size_t)i * 7 + c] = q[c];
            for (int c = 0; c < 3; ++c) aux_norm[(size_t)i * 7 + 4 + c] = o[6 + c];
        }
        if (critic_aux && ret[i] > 0.f) {
            la += goal_point_loss(q, o + 6, goal + (size_t)i * 7, goal + (size_t)i * 7 + 4, gq, gt);
            unit_quat_bwd(q, n, gq, gx);
            for (int c = 0; c < 4; ++c) g[2 + c] = gx[c] * inv_g;
            for (int c = 0; c < 3; ++c) g[6 + c] = gt[c] * inv_g;
        } else {
            for (int c = 0; c < 7; ++c) g[2 + c] = 0.f;
        }
    }
    lq = block_sum(lq, red);
    la = block_sum(la, red);
    if (tid == 0) {
        scalars[0] = lq * inv_k;
        scalars[1] = critic_aux ? la * inv_g : 0.f;
        scalars[2] = nk;
        scalars[3] = ng;
    }
}

extern "C" int gad_critic_loss(const float* out9, const float* tgt_out9, const float* reward, const float* done,
                               const float* perturb_flag, const float* ret, const float* goal, int B, float gamma,
                               int critic_aux, const float* inv_n, float* y, float* aux_norm, float* g_out9,
                               float* scalars, void* stream) {
    GAD_REQUIRE(out9 && tgt_out9 && reward && done && perturb_flag && ret && goal && y && g_out9 && scalars,
                GAD_ERR_NULL, "critic_loss: null pointer");
    GAD_REQUIRE(B >= 1, GAD_ERR_SHAPE, "critic_loss: B");
    hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out9, tgt_out9, reward, done,
                       perturb_flag, ret, goal, B, gamma, critic_aux, inv_n, y, aux_norm, g_out9, scalars);
    GAD_CHECK_LAUNCH("critic_loss");
    return GAD_OK;
}

// pi = tanh(mean)*scale + bias, aux = [normalize(extra[:4]), extra[4:]]
__global__ __launch_bounds__(256) void policy_outputs_kernel(const float* __restrict__ pol13, int B, int pitch,
                                                             const float* __restrict__ ascale, const float* __restrict__ abias,
                                                             float* __restrict__ pi, float* __restrict__ aux) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float* o = pol13 + (size_t)i * pitch;
    for (int c = 0; c < 6; ++c) {
        const float t = tanhf(o[c]) * ascale[c];
        pi[(size_t)i * 6 + c] = abias ? t + abias[c] : t;        // (two roundings, as torch evaluates tanh(mean) * scale + bias)
    }
    if (aux) {
        float q[4];
        unit_quat(o + 6, q);
        for (int c = 0; c < 4; ++c) aux[(size_t)i * 7 + c] = q[c];
        for (int c = 0; c < 3; ++c) aux[(size_t)i * 7 + 4 + c] = o[10 + c];
    }
}

extern "C" int gad_policy_outputs(const float* pol13, int B, int pitch, const float* action_scale, const float* action_bias,
                                  float* pi, float* aux_norm, void* stream) {
    GAD_REQUIRE(pol13 && action_scale && pi, GAD_ERR_NULL, "policy_outputs: null pointer");
    GAD_REQUIRE(pitch >= 6 && (!aux_norm || pitch >= 13), GAD_ERR_SHAPE,
                "policy_outputs: head pitch %d (6 mean columns, + 7 aux columns when aux_norm is requested)", pitch);
    if (B <= 0) return GAD_OK;
    hipLaunchKernelGGL(policy_outputs_kernel, dim3(gad_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, pol13, B, pitch,
                       action_scale, action_bias, pi, aux_norm);
    GAD_CHECK_LAUNCH("policy_outputs");
    return GAD_OK;
}

// GaussianPolicy.forward + sample (reference core/networks.py:339-371) from the raw head outputs
// head (B, pitch) = [mean (6) | extra (extra_dim) | log_std (6)]:
//   log_std = clamp(log_std, -10, 2);  x = mean + exp(log_std) * eps;  y = squash ? tanh(x) : x;
//   action = y * scale + bias;  mean_sq = squash ? tanh(mean) * scale + bias : mean;
//   log_prob = sum_c [ -eps^2/2 - log_std - log(sqrt(2 pi)) - log(scale * (1 - y^2) + 1e-6) ]
//   extra = extra_dim == 7 ? [normalize(extra[:4]), extra[4:]] : extra
__global__ __launch_bounds__(256) void policy_sample_kernel(const float* __restrict__ head, int B, int pitch, int extra_dim,
                                                            const float* __restrict__ eps,
                                                            const float* __restrict__ ascale,
                                                            const float* __restrict__ abias, int squash,
                                                            float* __restrict__ mean_sq, float* __restrict__ log_std,
                                                            float* __restrict__ log_prob, float* __restrict__ action,
                                                            float* __restrict__ extra) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float* o = head + (size_t)i * pitch;
    const float* ls = o + 6 + extra_dim;
    float lp = 0.f;
    for (int c = 0; c < 6; ++c) {
        const float l = fminf(fmaxf(ls[c], -10.f), 2.f);
        const float e = eps ? eps[(size_t)i * 6 + c] : 0.f;
        const float x = o[c] + expf(l) * e;
        const float sc = ascale ? ascale[c] : 1.f, bi = abias ? abias[c] : 0.f;
        const float y = squash ? tanhf(x) : x;
        if (log_std) log_std[(size_t)i * 6 + c] = l;
        if (action) action[(size_t)i * 6 + c] = squash ? y * sc + bi : x;
        if (mean_sq) mean_sq[(size_t)i * 6 + c] = squash ? tanhf(o[c]) * sc + bi : o[c];
        lp += -0.5f * e * e - l - 0.91893853320467274f - logf(sc * (1.f - y * y) + 1e-6f);
    }
    if (log_prob) log_prob[i] = lp;
    if (extra) {
        if (extra_dim == 7) {
            float q[4];
            unit_quat(o + 6, q);
            for (int c = 0; c < 4; ++c) extra[(size_t)i * 7 + c] = q[c];
            for (int c = 0; c < 3; ++c) extra[(size_t)i * 7 + 4 + c] = o[10 + c];
        } else {
            for (int c = 0; c < extra_dim; ++c) extra[(size_t)i * extra_dim + c] = o[6 + c];
        }
    }
}

extern "C" int gad_policy_sample(const float* head, int B, int pitch, int extra_dim, const float* eps,
                                 const float* action_scale, const float* action_bias, int squash, float* mean_sq,
                                 float* log_std, float* log_prob, float* action, float* extra, void* stream) {
    GAD_REQUIRE(head, GAD_ERR_NULL, "policy_sample: null pointer");
    GAD_REQUIRE(extra_dim >= 0 && pitch >= 12 + extra_dim, GAD_ERR_SHAPE,
                "policy_sample: pitch %d < 6 + extra_dim (%d) + 6", pitch, extra_dim);
    if (B <= 0) return GAD_OK;
    hipLaunchKernelGGL(policy_sample_kernel, dim3(gad_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, head, B, pitch,
                       extra_dim, eps, action_scale, action_bias, squash, mean_sq, log_std, log_prob, action, extra);
    GAD_CHECK_LAUNCH("policy_sample");
    return GAD_OK;
}

// control points moved by R = Rz(th) Ry(el) Rx(az) and translation: value of sum_p sum_xyz |.| and grad
__device__ __forceinline__ float bc_point_loss(const float* a, const float* e, float* ga) {
    // rotation matrices for prediction a and expert e (core/utils.py:890-910)
    float Ra[9], Re[9], dAz[9], dEl[9], dTh[9];
    for (int k = 0; k < 2; ++k) {
        const float* s = k == 0 ? a : e;
        const float cx = cosf(s[3]), sx = sinf(s[3]), cy = cosf(s[4]), sy = sinf(s[4]), cz = cosf(s[5]), sz = sinf(s[5]);
        float* R = k == 0 ? Ra : Re;
        // Rz*Ry*Rx
        R[0] = cz * cy; R[1] = cz * sy * sx - sz * cx; R[2] = cz * sy * cx + sz * sx;
        R[3] = sz * cy; R[4] = sz * sy * sx + cz * cx; R[5] = sz * sy * cx - cz * sx;
        R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
        if (k == 0) {
            dAz[0] = 0.f; dAz[1] = cz * sy * cx + sz * sx;  dAz[2] = -cz * sy * sx + sz * cx;
            dAz[3] = 0.f; dAz[4] = sz * sy * cx - cz * sx;  dAz[5] = -sz * sy * sx - cz * cx;
            dAz[6] = 0.f; dAz[7] = cy * cx;                 dAz[8] = -cy * sx;
            dEl[0] = -cz * sy; dEl[1] = cz * cy * sx; dEl[2] = cz * cy * cx;
            dEl[3] = -sz * sy; dEl[4] = sz * cy * sx; dEl[5] = sz * cy * cx;
            dEl[6] = -cy;      dEl[7] = -sy * sx;     dEl[8] = -sy * cx;
            dTh[0] = -sz * cy; dTh[1] = -sz * sy * sx - cz * cx; dTh[2] = -sz * sy * cx + cz * sx;
            dTh[3] = cz * cy;  dTh[4] = cz * sy * sx - sz * cx;  dTh[5] = cz * sy * cx + sz * sx;
            dTh[6] = 0.f;      dTh[7] = 0.f;                     dTh[8] = 0.f;
        }
    }
    float loss = 0.f;
    for (int c = 0; c < 6; ++c) ga[c] = 0.f;
    for (int p = 0; p < 6; ++p) {
        const V3 v = control_point(p, false);
        float s[3];
        for (int r = 0; r < 3; ++r) {
            const float pa = Ra[r * 3] * v.x + Ra[r * 3 + 1] * v.y + Ra[r * 3 + 2] * v.z + a[r];
            const float pe = Re[r * 3] * v.x + Re[r * 3 + 1] * v.y + Re[r * 3 + 2] * v.z + e[r];
            const float d = pa - pe;
            loss += fabsf(d);
            s[r] = sgn(d);
            ga[r] += s[r];
        }
        for (int r = 0; r < 3; ++r) {
            ga[3] += s[r] * (dAz[r * 3] * v.x + dAz[r * 3 + 1] * v.y + dAz[r * 3 + 2] * v.z);
            ga[4] += s[r] * (dEl[r * 3] * v.x + dEl[r * 3 + 1] * v.y + dEl[r * 3 + 2] * v.z);
            ga[5] += s[r] * (dTh[r * 3] * v.x + dTh[r * 3 + 1] * v.y + dTh[r * 3 + 2] * v.z);
        }
    }
    return loss;
}

__global__ __launch_bounds__(256) void actor_loss_kernel(const float* __restrict__ pol13, const float* __restrict__ pi,
                                                         const float* __restrict__ expert_action,
                                                         const float* __restrict__ expert_flag,
                                                         const float* __restrict__ ret, const float* __restrict__ goal,
                                                         int B, int pitch, float bc_scale, int policy_aux,
                                                         const float* __restrict__ ascale,
                                                         const double* __restrict__ g_pi_critic,
                                                         const float* __restrict__ inv_n, float* __restrict__ g13,
                                                         float* __restrict__ scalars) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float ne = 0.f, ng = 0.f;
    for (int i = tid; i < B; i += 256) { ne += expert_flag[i] >= 1.f ? 1.f : 0.f; ng += ret[i] > 0.f ? 1.f : 0.f; }
    ne = block_sum(ne, red);
    ng = block_sum(ng, red);
    const float inv_e = inv_n ? inv_n[0] : 1.f / (ne * 6.f);
    const float inv_g = inv_n ? inv_n[1] : 1.f / (ng * 6.f);
    float lb = 0.f, la = 0.f;
    for (int i = tid; i < B; i += 256) {
        const float* o = pol13 + (size_t)i * pitch;
        float* g = g13 + (size_t)i * pitch;
        const float* p = pi + (size_t)i * 6;
        float gpi[6];
        for (int c = 0; c < 6; ++c) gpi[c] = g_pi_critic ? (float)g_pi_critic[(size_t)i * 6 + c] : 0.f;
        if (expert_flag[i] >= 1.f) {
            float ga[6];
            lb += bc_point_loss(p, expert_action + (size_t)i * 6, ga);
            for (int c = 0; c < 6; ++c) gpi[c] += ga[c] * inv_e * bc_scale;
        }
        for (int c = 0; c < 6; ++c) {                      // pi = tanh(m)*scale
            const float th = p[c] / ascale[c];
            g[c] = gpi[c] * ascale[c] * (1.f - th * th);
        }
        if (policy_aux && ret[i] > 0.f) {                  // policy_aux implies pitch >= 13 (checked by the host entry)
            float q[4], gq[4], gt[3], gx[4];
            const float n = unit_quat(o + 6, q);
            la += goal_point_loss(q, o + 10, goal + (size_t)i * 7, goal + (size_t)i * 7 + 4, gq, gt);
            unit_quat_bwd(q, n, gq, gx);
            for (int c = 0; c < 4; ++c) g[6 + c] = gx[c] * inv_g;
            for (int c = 0; c < 3; ++c) g[10 + c] = gt[c] * inv_g;
        } else {
            for (int c = 6; c < pitch; ++c) g[c] = 0.f;
        }
    }
    lb = block_sum(lb, red);
    la = block_sum(la, red);
    if (tid == 0) {
        scalars[0] = lb * inv_e * bc_scale;
        scalars[1] = policy_aux ? la * inv_g : 0.f;
        scalars[2] = ne;
        scalars[3] = ng;
    }
}

extern "C" int gad_actor_loss(const float* pol13, const float* pi, const float* expert_action,
                              const float* expert_flag, const float* ret, const float* goal, int B, int pitch,
                              float bc_scale, int policy_aux, const float* action_scale, const double* g_pi_critic,
                              const float* inv_n, float* g_pol13, float* scalars, void* stream) {
    GAD_REQUIRE(pol13 && pi && expert_action && expert_flag && ret && goal && action_scale && g_pol13 && scalars,
                GAD_ERR_NULL, "actor_loss: null pointer");
    GAD_REQUIRE(B >= 1, GAD_ERR_SHAPE, "actor_loss: B");
    GAD_REQUIRE(pitch >= 6 && (!policy_aux || pitch >= 13), GAD_ERR_SHAPE,
                "actor_loss: head pitch %d (6 mean columns, + 7 aux columns with policy_aux)", pitch);
    hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pol13, pi, expert_action,
                       expert_flag, ret, goal, B, pitch, bc_scale, policy_aux, action_scale, g_pi_critic, inv_n, g_pol13,
                       scalars);
    GAD_CHECK_LAUNCH("actor_loss");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void actor_critic_loss_kernel(const float* __restrict__ out9,
                                                                const float* __restrict__ expert_flag,
                                                                const float* __restrict__ ret, int B, float ratio,
                                                                const float* __restrict__ inv_n,
                                                                float* __restrict__ g9, float* __restrict__ scalars) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float nk = 0.f;
    for (int i = tid; i < B; i += 256) nk += (expert_flag[i] >= 1.f && ret[i] > 0.f) ? 0.f : 1.f;
    nk = block_sum(nk, red);
    const float inv = inv_n ? inv_n[0] : 1.f / nk;
    float l = 0.f;
    for (int i = tid; i < B; i += 256) {
        const float q1 = out9[(size_t)i * 9], q2 = out9[(size_t)i * 9 + 1];
        float* g = g9 + (size_t)i * 9;
        for (int c = 0; c < 9; ++c) g[c] = 0.f;
        if (!(expert_flag[i] >= 1.f && ret[i] > 0.f)) {
            l += fminf(q1, q2);
            if (q1 < q2) g[0] = -ratio * inv;
            else if (q2 < q1) g[1] = -ratio * inv;
            else { g[0] = -0.5f * ratio * inv; g[1] = g[0]; }
        }
    }
    l = block_sum(l, red);
    if (tid == 0) { scalars[0] = -ratio * l * inv; scalars[1] = nk; }
}

extern "C" int gad_actor_critic_loss(const float* out9, const float* expert_flag, const float* ret, int B, float ratio,
                                     const float* inv_n, float* g_out9, float* scalars, void* stream) {
    GAD_REQUIRE(out9 && expert_flag && ret && g_out9 && scalars, GAD_ERR_NULL, "actor_critic_loss: null pointer");
    GAD_REQUIRE(B >= 1, GAD_ERR_SHAPE, "actor_critic_loss: B");
    hipLaunchKernelGGL(actor_critic_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out9, expert_flag, ret, B,
                       ratio, inv_n, g_out9, scalars);
    GAD_CHECK_LAUNCH("actor_critic_loss");
    return GAD_OK;
}

// local mask counts of a minibatch (the numbers the masked means divide by): [kept, goal rows, expert rows,
// rows not (expert & reward)] as doubles -- a data-parallel run all-reduces them (ga_ddpg_amd/parallel.py)
__global__ __launch_bounds__(256) void mask_counts_kernel(const float* __restrict__ ret, const float* __restrict__ expert,
                                                          const float* __restrict__ perturb, int B,
                                                          double* __restrict__ out4) {
    __shared__ float red[4];
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < B; i += 256) {
        const bool reward = ret[i] > 0.f, exp = expert[i] >= 1.f;
        c[0] += perturb[i] < 1.f ? 1.f : 0.f;
        c[1] += reward ? 1.f : 0.f;
        c[2] += exp ? 1.f : 0.f;
        c[3] += (reward && exp) ? 0.f : 1.f;
    }
    for (int k = 0; k < 4; ++k) {
        const float s = block_sum(c[k], red);
        if (threadIdx.x == 0) out4[k] = (double)s;
    }
}

extern "C" int gad_mask_counts(const float* ret, const float* expert_flag, const float* perturb_flag, int B, double* out4,
                               void* stream) {
    GAD_REQUIRE(ret && expert_flag && perturb_flag && out4, GAD_ERR_NULL, "mask_counts: null pointer");
    GAD_REQUIRE(B >= 1, GAD_ERR_SHAPE, "mask_counts: B");
    hipLaunchKernelGGL(mask_counts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ret, expert_flag, perturb_flag, B, out4);
    GAD_CHECK_LAUNCH("mask_counts");
    return GAD_OK;
}

__global__ __launch_bounds__(256) void target_noise_kernel(const float* __restrict__ pi, const float* __restrict__ u,
                                                           int n, float level, int normal, float* __restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int c = q % 6;
    // uniform draw u in [0,1): (u*3-6)*level -- reference quirk: always negative;  normal draw u ~ N(0,1): u*level/2
    float d = normal ? u[q] * level / 2.f : (u[q] * 3.f - 6.f) * level;
    if (c >= 3) d *= 5.f;
    else d = fminf(fmaxf(d, -0.01f), 0.01f);
    out[q] = pi[q] + d;
}

extern "C" int gad_target_noise(const float* pi, const float* u, int B, float level, int normal, float* out,
                                void* stream) {
    GAD_REQUIRE(pi && u && out, GAD_ERR_NULL, "target_noise: null pointer");
    if (B <= 0) return GAD_OK;
    hipLaunchKernelGGL(target_noise_kernel, dim3(gad_cdiv(B * 6, 256)), dim3(256), 0, (hipStream_t)stream, pi, u, B * 6,
                       level, normal, out);
    GAD_CHECK_LAUNCH("target_noise");
    return GAD_OK;
}
