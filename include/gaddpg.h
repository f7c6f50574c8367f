/*
 * gaddpg.h -- C ABI of libgaddpg.so: the MI355X (gfx950) implementation of GA-DDPG's
 * PointNet++-encoded actor-critic update step.
 *
 * Boundary rules (all entry points):
 *   - extern "C", plain device pointers + sizes, no torch / C++ types.  Every pointer is a DEVICE
 *     pointer unless named host_*.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - outputs are caller-allocated; no hidden allocation, no host synchronisation, re-entrant,
 *     ordered only by `stream` (same contract as the reference extension, which launches on
 *     at::cuda::getCurrentCUDAStream() without syncing).
 *   - return value: 0 = GAD_OK, <0 = gad_status error (no exceptions cross the ABI).  The Python
 *     binding turns a non-zero status into RuntimeError, mirroring the TORCH_CHECK -> RuntimeError
 *     behaviour of the reference extension's CHECK_* macros.
 *
 * Section A replaces the C++/CUDA extension `pointnet2_ops._ext` that reference core/networks.py:10
 * and core/utils.py:32 reach through pointnet2_ops.pointnet2_utils (upstream bindings.cpp:
 * furthest_point_sampling, gather_points[_grad], ball_query, group_points[_grad]).
 * Sections B-E are the fused update-step path that sits behind core/networks.py:65-92,182-371,
 * core/ddpg.py:119-185, core/agent.py:127-139,192-259, core/loss.py:17-31 and core/utils.py:750-774.
 */
#ifndef GADDPG_H
#define GADDPG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GAD_OK = 0,
    GAD_ERR_NULL = -1,      /* required pointer is NULL                                   */
    GAD_ERR_SHAPE = -2,     /* size / shape outside what the kernels support              */
    GAD_ERR_LAUNCH = -3,    /* hipGetLastError() != hipSuccess after a launch             */
    GAD_ERR_UNSUPPORTED = -4
} gad_status;

int gad_abi_version(void);                 /* bumped on any signature change or new entry point (2: gad_set_option,
                                            * gad_bn_running_update, gad_replay_gather; 3: head pitch in gad_policy_outputs /
                                            * gad_actor_loss, noise type in gad_target_noise, gad_policy_sample;
                                            * 4: max-pool fused into the pooled layer's GEMM (pool_key fields, nullable
                                            * zout, gad_pool_finalize, zmax in gad_pool_bwd_stats); the deferred-BatchNorm
                                            * fields and the slab / graph switches of version 3 are gone;
                                            * 5: gad_gemm_bwd, gad_optim_jobs, gad_last_kernel; 6: gad_gemm_dw_reduce,
                                            * GAD_DW_REDUCE_LATER; 7: trailing BatchNorm blocks of gad_gemm_fwd_args (in_*)
                                            * and gad_dz_src (bn_*, gacc_*), GAD_STAT_REPLICAS 8 -> 4;
                                            * 8: split-bf16 weight mirrors (gad_split_weights; W_split* of gad_gemm_fwd_args,
                                            * W_split_t* of gad_gemm_dx_args), option "mfma_split" as a family mask;
                                            * 9: gad_transpose_batched; 10: gad_stream_priority;
                                            * 11: step replay (section H: gad_plan_*), gad_copy_buffers,
                                            * action_bias in gad_policy_outputs)                                      */
/* diagnostics: which kernel family the last gad_gemm_fwd / _dx / _dw / _bwd call routed to ("gemm_fwd(stream)",
 * "gemm_dx(wide)", "gemm_bwd(stream)", "gemm_dw" = generic tile kernel, ...); bench.py labels its per-kernel table
 * with it instead of restating the routing rules.                                                  */
const char* gad_last_kernel(void);
const char* gad_last_error(void);          /* thread-local description of the last <0   */
/* Kernel-selection switches for A/B diagnostics (defaults in brackets).  "fwd_stream" [1]: route the wide and
 * shallow SA1 forward layers to the streaming kernel instead of the tiled one; "dx_stream" [1]: the same for their dX; "fwd_skinny" / "dx_skinny" / "dw_skinny" [1]: route
 * the small-M (<= 1024 rows) forward / dX / dW layers to the split-K kernels; "skinny_nw" [8]: wavefronts per workgroup of
 * those kernels (8, or 4 = the small-footprint instantiation: A/B in profiles/r05_skinny_footprint.txt); "fwd_stream_wgs" [256] /
 * "fwd_stream_l1_wgs" [256] / "bwd_stream_wgs" [256]: persistent workgroups of the streaming forward (SA1 layers 2 / 3; the gathered
 * layer 1) and of the fused SA1 backward (= its partial dW blocks).  Returns GAD_ERR_SHAPE for an
 * unknown name.  Not part of the numerical contract: both settings satisfy the same parity tests.
 * "mfma_split" [0; the Python package sets GAD_SPLIT_ALL when it loads the library -- its accuracy gates are green on hardware,
 * DESIGN.md section 5 -- a plain C caller gets the f32 MFMA unless it opts in]: the arithmetic of the layer GEMMs' products.
 * Precondition of the non-zero settings: finite operands with |x| < 3.39e38.
 * 0: v_mfma_f32_32x32x2_f32 throughout.  Non-zero: a mask of kernel families that form every FP32 product from split-bf16
 * terms on v_mfma_f32_32x32x16_bf16 with f32 accumulation (GAD_SPLIT_* below; 1 = every family that has the form).  A launch
 * takes the split form only if its family's bit is set AND the call carries the weight mirror it needs (W_split /
 * W_split_t; the streaming SA1 kernels split W themselves); otherwise it runs the FP32-MFMA kernel.                     */
#define GAD_SPLIT_ALL 1
#define GAD_SPLIT_FWD_STREAM 2
#define GAD_SPLIT_FWD_WIDE 4
#define GAD_SPLIT_DX_WIDE 8
#define GAD_SPLIT_DW_WIDE 16
#define GAD_SPLIT_BWD_STREAM 32
#define GAD_SPLIT_DW_STREAM 64
int gad_set_option(const char* name, int value);

/* In-kernel launch timing (diagnostics / bench.py roofline): arm `slot` -- device memory, GAD_TIMING_WAVES x {start, end}
 * uint64 pairs, starts initialised to a large value, ends to 0 -- for the NEXT gad_gemm_fwd / gad_gemm_dx / gad_gemm_dw /
 * gad_segment_pool call of this thread.  Every wavefront of that launch stores its own start / end wall-clock stamp (ticks of
 * gad_wall_clock_khz()) into its entry; max(end) - min(start) is the dispatch duration a profiler would report, without
 * tracing and regardless of what other streams do.  Wavefronts beyond GAD_TIMING_WAVES are not recorded.  NULL disarms. */
#define GAD_TIMING_WAVES 16384
int gad_timing_slot(void* slot);
/* Wavefront issue priority of the launches a stream carries (ABI 10).  prio 0..3 (0 = the hardware default; 0 also removes the
 * stream from the table, 8 entries): gad_gemm_fwd / _dx / _dw / _bwd / gad_segment_pool launches on `stream` raise their
 * wavefronts with s_setprio at their first instruction, so that where they share a SIMD with the wavefronts of launches on
 * other streams (the step's weight-gradient lanes, the next step's prefetch) the arbiter issues theirs first.  This is per
 * wavefront, inside the CU -- unlike a HIP stream priority it does not change the queue mapping.  Numerics unaffected.
 * Process-wide; call it before the steps start (not thread-safe against concurrent launches).                              */
int gad_stream_priority(void* stream, int prio);
int gad_wall_clock_khz(void);              /* rate of that clock (hipDeviceAttributeWallClockRate), 0 if unavailable    */

/* Grid-size hint: the layers of SA1 / SA2 run over de-duplicated rows whose count lives on the device (n_rows_dev); the host
 * only knows the worst case (n_rows, 20-40x larger), and a grid sized for it is mostly workgroups that exit at once (~1 ns
 * each: 3-6 us of an SA2 launch).  *rows_host (a HOST int, read at call time) = the number of live rows the caller expects
 * for the NEXT tiled gad_gemm_fwd / gad_gemm_dx launch of this thread; the grid is sized for it.  It is a hint only: the
 * kernels walk the rows in a grid-stride loop bounded by the device count, so any value gives the same result.           */
int gad_grid_rows_hint(const int32_t* rows_host, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A. pointnet2_ops._ext operator parity (materialising, reference API shapes)
 * ------------------------------------------------------------------------------------------- */

/* furthest_point_sampling(xyz (B,N,3) f32, npoint) -> idx (B,M) i32.  Start index 0, points with
 * |p|^2 <= 1e-3 are never selected/updated, arg-max ties resolved as the upstream block reduction
 * does.  new_xyz (B,M,3) is optional (fused gather_operation of the xyz rows).    */
int gad_furthest_point_sampling(const float* xyz, int B, int N, int M, int32_t* idx,
                                float* new_xyz /*nullable*/, void* stream);

/* gather_points(points (B,C,N), idx (B,M)) -> out (B,C,M);  _grad scatters back (zero-fills).  */
int gad_gather_points(const float* points, const int32_t* idx, int B, int C, int N, int M,
                      float* out, void* stream);
int gad_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                           float* grad_points, void* stream);

/* ball_query(new_xyz (B,M,3), xyz (B,N,3), radius, nsample) -> idx (B,M,nsample) i32: first
 * nsample indices with d2 < radius^2 in ascending order, padded with the first hit, zeros if none.
 * cnt (B,M) (optional) receives the number of distinct hits kept (<= nsample).                  */
int gad_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius,
                   int nsample, int32_t* idx, int32_t* cnt /*nullable*/, void* stream);

/* group_points(points (B,C,N), idx (B,M,S)) -> out (B,C,M,S);  _grad: atomic scatter-add into
 * grad_points (B,C,N), which the call zero-fills first (upstream torch::zeros + atomicAdd).      */
int gad_group_points(const float* points, const int32_t* idx, int B, int C, int N, int M, int S,
                     float* out, void* stream);
int gad_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                          int S, float* grad_points, void* stream);

/* QueryAndGroup in one pass (ball_query + group xyz + recentre + group features + concat):
 * out (B,3+C,M,S) exactly as pointnet2_utils.QueryAndGroup(use_xyz=True) returns it; idx as above.
 * This is the HBM-bound "config 4a" kernel of BASELINE.md.                                       */
int gad_query_and_group(const float* new_xyz, const float* xyz, const float* features /*(B,C,N)*/,
                        int B, int C, int N, int M, float radius, int nsample, int32_t* idx,
                        float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * B. geometry for the fused set-abstraction path (de-duplicated neighbourhood rows)
 * ------------------------------------------------------------------------------------------- */

/* point_state (B,C4,NP) channel-major f32 (rows x,y,z,flag; reference replay layout) ->
 * xyz (B,N,3) and point-major features feat (B*N,4) = [x,y,z,flag], N = NP - skip.              */
int gad_prep_points(const float* point_state, int B, int C4, int NP, int skip, float* xyz,
                    float* feat, void* stream);

/* Compact ball-query output into unique (group, point) rows.  Group g = b*M+m keeps its
 * max(cnt,1) distinct hits; the first hit carries weight nsample-cnt+1 (the padded duplicates).
 * grp_off (G+1) exclusive offsets, row_pt global point index b*Nsrc+j, row_grp = g, row_w weight,
 * n_rows device scalar = grp_off[G].  Deterministic order (g ascending, slot ascending).         */
int gad_rows_from_ball_query(const int32_t* idx, const int32_t* cnt, int G, int M, int Nsrc,
                             int nsample, int32_t* grp_off, int32_t* row_pt, int32_t* row_grp,
                             float* row_w, int32_t* n_rows, void* stream);
/* GroupAll: rows = all points, group = sample (pts_per_group points each), weight 1.            */
int gad_rows_group_all(int G, int pts_per_group, int32_t* grp_off, int32_t* row_pt,
                       int32_t* row_grp, float* row_w, int32_t* n_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C. layer kernels (FP32 MFMA GEMMs with fused producers / epilogues)
 *
 * Weights use the PACKED layout: W (n_out, Kp) row-major, Kp = K rounded up to 8, an optional bias
 * stored as column `ones_col`, zero padding after it.  Activations are row-major (rows, channels).
 * ------------------------------------------------------------------------------------------- */

#define GAD_MAX_GROUPS 3
/* BatchNorm statistic accumulators are replicated: a block adds into replica blockIdx.x % GAD_STAT_REPLICAS (bounds
 * same-address atomic contention); gad_bn_finalize / gad_bn_bwd_coef and the consumers' prologues sum the replicas.
 * 4 since ABI 7 (8 before: every consumer workgroup now reads them; 4 measured +0.5 %, 2 measured -0.2 %).       */
#ifndef GAD_STAT_REPLICAS
#define GAD_STAT_REPLICAS 4
#endif

typedef struct {
    /* rows */
    const int32_t* n_rows_dev; /* device scalar with the live row count, or NULL -> n_rows       */
    int32_t n_rows;            /* static row count (upper bound when n_rows_dev is given)        */
    const float* row_w;        /* (rows) multiplicity weights, NULL -> 1                         */
    /* input mode: 0 = ACT (act(scale*z+shift) of a previous layer's raw output), 1 = GATHER,
     * 2 = ACT recomputed from the gathered first layer (pre_W below)                              */
    int32_t mode;
    const float* zin;          /* ACT: (rows, zin_pitch)                                         */
    int32_t zin_pitch;
    int32_t c_in;              /* ACT: channels read from zin;  GATHER: 3 + feat_c + act_c       */
    const float* scale;        /* ACT: per-channel affine (NULL -> identity)                     */
    const float* shift;
    int32_t relu;              /* ACT: apply max(.,0) after the affine                           */
    const float* extra;        /* ACT: optional extra input column (rows) placed at index c_in   */
    int32_t ones_col;          /* column holding 1.0 (bias), or -1                               */
    /* GATHER: row r = [feat[pt] (feat_c, multiple of 4), src_xyz[pt]-ctr_xyz[grp] (3),
     *                  action[grp/grp_per_sample] (act_c)]  -- PACKED column order (features first so
     *                  they are 16-byte aligned); the host's master->packed map permutes the
     *                  reference's [xyz, features] conv-weight columns accordingly                  */
    const float* src_xyz;      /* (points,3)                                                     */
    const float* ctr_xyz;      /* (groups,3) or NULL (GroupAll: no recentring)                   */
    const float* feat;         /* (points, feat_c) point-major                                   */
    int32_t feat_c;
    const float* action;       /* (samples, act_c) or NULL                                       */
    int32_t act_c;
    int32_t grp_per_sample;
    const int32_t* row_pt;
    const int32_t* row_grp;
    /* groups (blockIdx.z): independent GEMMs sharing the row set (e.g. the three critic trunks) */
    int32_t n_groups;
    int32_t zin_off[GAD_MAX_GROUPS]; /* channel offset into zin (ACT)                            */
    int32_t w_off[GAD_MAX_GROUPS];   /* element offset into W                                    */
    int32_t out_off[GAD_MAX_GROUPS]; /* channel offset into zout / stats                         */
    int32_t n_out[GAD_MAX_GROUPS];
    const float* W;
    int32_t Kp;
    /* outputs */
    float* zout;               /* (rows, zout_pitch) raw pre-activation output (bias included); nullable when the max-pool
                                * is fused (pool_key): a pass that is never back-propagated keeps only the pooled maxima */
    int32_t zout_pitch;
    double* stat_sum;          /* per output channel sum_r w*z and sum_r w*z^2 (f64 atomics),    */
    double* stat_sq;           /*   NULL -> no statistics                                        */
    int32_t stat_stride;       /* elements between the GAD_STAT_REPLICAS replicas of the sums     */
    /* segment max-pool fused into the epilogue (the pooled layer of a set-abstraction stage): per (group, channel) the
     * packed maximum of sgn(gamma) * z over the group's rows, finished by gad_pool_finalize (which also resets the keys).
     * One group, n_out a multiple of 64, rows a multiple of 4.  pool_key == NULL: not used.                            */
    uint64_t* pool_key;        /* (groups, n_out) keys, all 0 before the launch                   */
    const int32_t* pool_row_grp; /* (rows) group index of every row (contiguous runs: CSR order)  */
    const float* pool_gamma;   /* (n_out) BatchNorm weight of THIS layer                          */
    /* train-mode BatchNorm finalisation of the INPUT layer folded into this launch (round 4; in_stat_sum == NULL: not used --
     * scale / shift are read as given).  The launch forms scale / shift of its c_in input channels from that layer's f64
     * statistics (the arithmetic of gad_bn_finalize) in every workgroup's prologue; its FIRST workgroup also publishes them to
     * `scale` / `shift` (written although declared const above), in_mean / in_istd, and applies the running-statistics update.
     * A route that has no such prologue runs gad_bn_finalize on `stream` first: results do not depend on the route.         */
    const double* in_stat_sum; /* (GAD_STAT_REPLICAS, in_stat_stride) sums of the input layer, its first channel */
    const double* in_stat_sq;
    int32_t in_stat_stride;
    double in_count;           /* rows behind the statistics (padded duplicates included)         */
    const float* in_gamma;
    const float* in_beta;
    float in_eps;
    float in_momentum;
    float* in_running_mean;    /* nullable                                                        */
    float* in_running_var;
    float* in_mean;            /* published for the backward pass; nullable                       */
    float* in_istd;
    /* mode 2 (round 4): the 64-channel ACT input is NOT read from zin but recomputed, per 32-row slab, from the gathered rows
     * of the stage's first layer (src_xyz / ctr_xyz / feat / action / row_pt / row_grp as in mode 1) and that layer's packed
     * weights -- the same products in the same order as that layer's own launch, so the values are bit-equal to what it
     * stored -- then activated with scale / shift (or the in_* block) as in mode 0.  A streaming-kernel form only: SA1's
     * rows (feat_c 4, act_c 0 / 6), Kp = c_in = n_out = 64, n_rows >= 32768; anything else is GAD_ERR_SHAPE.            */
    const float* pre_W;        /* (64, pre_Kp) packed weights of the recomputed layer                */
    int32_t pre_Kp;            /* 8 or 16                                                             */
    /* split-bf16 mirror of W (ABI 8; NULL: the launch multiplies on the FP32 MFMA whatever "mfma_split" says): three bf16
     * planes hi | mid | lo written by gad_split_weights, plane p at W_split + p * W_split_plane, row n at + n * W_split_pitch,
     * the first W_split_pitch input columns of every row.  One group only.                                              */
    const uint16_t* W_split;
    int32_t W_split_pitch;     /* bf16 elements per row (= columns mirrored: Kp, or feat_c for a gathered first layer)    */
    int32_t W_split_plane;     /* bf16 elements per plane                                                                  */
} gad_gemm_fwd_args;

int gad_gemm_fwd(const gad_gemm_fwd_args* host_args, void* stream);

/* train-mode BatchNorm finalisation: mean/var from the f64 sums over `count` rows (duplicates
 * included), scale = gamma*istd, shift = beta - mean*scale, running stats momentum update
 * (unbiased variance), saves mean/istd for the backward pass.                                   */
int gad_bn_finalize(const double* stat_sum, const double* stat_sq, int stat_stride, const float* gamma,
                    const float* beta, int C, double count, float eps, float momentum,
                    float* running_mean /*nullable*/, float* running_var /*nullable*/,
                    float* scale, float* shift, float* mean, float* istd, void* stream);
/* eval-mode: scale/shift from the running statistics */
/* momentum update of the running statistics from SAVED batch statistics (mean, istd = 1/sqrt(var_biased+eps)) of a
 * pass whose gad_bn_finalize ran with running_mean = running_var = NULL (a pass overlapped with another pass of the
 * same network on a second stream: the updates are then applied in the reference's order).  count: (C) rows behind
 * each channel's statistics.                                                                                      */
int gad_bn_running_update(const float* mean, const float* istd, const float* count, int C, float eps,
                          float momentum, float* running_mean, float* running_var, void* stream);

int gad_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, int C, float eps, float* scale, float* shift,
                       void* stream);

/* segment max-pool over each group's rows of act(scale*z+shift): out (G,C) point-major,
 * argmax (G,C) = global row index of the first maximum.                                          */
int gad_segment_pool(const float* z, int z_pitch, int C, const float* scale, const float* shift,
                     const int32_t* grp_off, int G, float* out, int32_t* argmax, void* stream);

/* Finish of the max-pool folded into the pooled layer's GEMM epilogue (gad_gemm_fwd_args.pool_key): finalises that
 * layer's train-mode BatchNorm from its f64 statistics (stat_sum != NULL: same arithmetic and outputs as gad_bn_finalize;
 * running_mean / running_var nullable) or takes scale / shift as given (stat_sum == NULL: eval mode), then per (group,
 * channel): zmax = the winning raw value, out = relu(scale * zmax + shift), argmax = the winner's global row where
 * out > 0, else the group's first row (grp_off[g]); the keys are reset to 0.  argmax / zmax nullable.
 * Replaces upstream's F.max_pool2d over the nsample axis (reference call site core/networks.py:66-81).              */
int gad_pool_finalize(uint64_t* key, int C, int G, const int32_t* grp_off, const double* stat_sum /*nullable*/,
                      const double* stat_sq, int stat_stride, double count, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                      float* mean /*nullable*/, float* istd /*nullable*/, float* out, int32_t* argmax, float* zmax,
                      void* stream);

/* apply act(scale*z+shift) elementwise -> out (rows,C) (used at API boundaries only)            */
int gad_affine_act(const float* z, int z_pitch, int rows, int C, const float* scale,
                   const float* shift, int relu, float* out, int out_pitch, void* stream);

/* batched 2-D transpose at the operator boundary: dst[b][j][i] = src[b][i][j] for i < R, j < C; rows of src / dst are
 * src_pitch / dst_pitch floats apart, batches src_batch / dst_batch floats (upstream's (B, C, N) channel-major tensors
 * <-> the point-major rows the kernels read: replaces tensor.transpose(1, 2).contiguous() at the pointnet2_ops facade,
 * reference call sites pointnet2_modules.py forward: features (B, C, N) in, new_features (B, C', npoint) out).          */
int gad_transpose_batched(const float* src, float* dst, int B, int R, int C, int src_pitch, long long src_batch,
                          int dst_pitch, long long dst_batch, void* stream);

/* backward source of dY for a layer: dense G (rows,C) or routed from the pooled gradient        */
typedef struct {
    const float* z;            /* (rows, z_pitch) raw output of THIS layer                       */
    int32_t z_pitch;
    const float* scale;        /* this layer's forward affine (mask = scale*z+shift > 0); NULL = identity */
    const float* shift;
    int32_t relu;              /* 0: no activation after this layer (dY = G)                     */
    const float* coefP;        /* BN backward: dZ = P*dY - w*(Q + S*z); NULL -> dZ = dY          */
    const float* coefQ;
    const float* coefS;
    const float* row_w;
    int32_t gmode;             /* 0 dense G, 1 pooled (argmax routing)                           */
    const float* G;            /* dense: (rows, g_pitch)                                         */
    int32_t g_pitch;
    const int32_t* argmax;     /* pooled: (groups, C) global row index                           */
    const float* dout;         /* pooled: (groups, C)                                            */
    const int32_t* row_grp;    /* pooled: (rows)                                                 */
    int32_t c;                 /* channels of this layer (n_out)                                 */
    int32_t premasked;         /* the ReLU mask is already applied to G / dout (dX epilogue with store_masked,
                                * gad_pool_bwd_stats with mask_in_place): `relu`, scale and shift are not needed   */
    /* BatchNorm-backward coefficients formed by the consuming launch itself (round 4; bn_dbeta == NULL: coefP / Q / S are
     * read as given).  Every workgroup of a dX / dW launch forms P, Q, S of the layer's channels from the f64 sums
     * (the arithmetic of gad_bn_bwd_coef: needs `scale`) in its prologue -- no gad_bn_bwd_coef launch in front of it; the
     * launch whose block carries gacc_gamma / gacc_beta adds dgamma / dbeta to the gradient arena (first workgroup; give
     * them to exactly ONE of the launches that consume a layer).  A route without such a prologue runs gad_bn_bwd_coef on
     * `stream` first (coefP / Q / S must then point at scratch for its output).                                          */
    const double* bn_dbeta;    /* (GAD_STAT_REPLICAS, bn_stride) sums of this layer, its first channel */
    const double* bn_dgamma;
    int32_t bn_stride;
    double bn_count;
    const float* bn_mean;
    const float* bn_istd;
    double* gacc_gamma;        /* nullable */
    double* gacc_beta;
} gad_dz_src;

/* pooled-gradient statistics for the BN that feeds a segment pool: dbeta/dgamma f64 sums        */
/* mask_in_place != 0: dout[g][c] is also overwritten with its ReLU-masked value (0 where the arg-max row's
 * activation is not positive), so that the layer's dX / dW can take it as `premasked`.             */
/* zmax (nullable): (G,C) raw value of every arg-max row as saved by gad_pool_finalize -- read instead of gathering
 * z[argmax] (argmax / z may then be NULL).                                                          */
int gad_pool_bwd_stats(float* dout, const int32_t* argmax, int G, int C, const float* z,
                       int z_pitch, const float* scale, const float* shift, const float* mean,
                       const float* istd, double* dbeta, double* dgamma, int stat_stride,
                       int mask_in_place, const float* zmax, void* stream);

/* BN backward coefficients from (dbeta,dgamma): P,Q,S above; also accumulates dgamma/dbeta into
 * the f64 gradient arena slots gacc_gamma/gacc_beta (nullable).                                 */
int gad_bn_bwd_coef(const double* dbeta, const double* dgamma, int stat_stride, const float* scale,
                    const float* mean, const float* istd, int C, double count, float* coefP,
                    float* coefQ, float* coefS, double* gacc_gamma, double* gacc_beta,
                    void* stream);

typedef struct {
    const int32_t* n_rows_dev;
    int32_t n_rows;
    gad_dz_src dz;
    int32_t n_groups;
    int32_t dz_off[GAD_MAX_GROUPS];  /* channel offset of the group inside dz (z, G, coef)       */
    int32_t w_off[GAD_MAX_GROUPS];
    int32_t n_out[GAD_MAX_GROUPS];
    int32_t gout_off[GAD_MAX_GROUPS];/* channel offset into gout                                  */
    int32_t accumulate;              /* groups add into the same gout columns (shared input)      */
    const float* W;
    int32_t Kp;
    int32_t k_valid;                 /* input channels to produce (columns >= k_valid are skipped)*/
    /* epilogue 0: store gout (rows, gout_pitch)                                                  */
    int32_t epilogue;
    float* gout;
    int32_t gout_pitch;
    /* epilogue 0 + statistics for the PREVIOUS layer's BN backward (nullable)                    */
    const float* zprev; int32_t zprev_pitch;
    const float* prev_scale; const float* prev_shift; const float* prev_mean; const float* prev_istd;
    double* prev_dbeta; double* prev_dgamma; int32_t stat_stride;
    int32_t store_masked;            /* with prev statistics: gout receives the ReLU-masked gradient (dY where the previous
                                      * layer's activation is positive, else 0) -> its consumer's dz is `premasked`    */
    /* epilogue 1: gather-layer scatter (packed column order): columns [0,feat_c) atomically added to
     * dfeat[row_pt], columns [feat_c+3, feat_c+3+act_c) to daction[row_grp/grp_per_sample] (f64:
     * the per-sample action gradient is a sum of many cancelling terms)                            */
    float* dfeat; int32_t feat_c; const int32_t* row_pt; const int32_t* row_grp;
    double* daction; int32_t act_c; int32_t grp_per_sample;
    /* split-bf16 mirror of W^T (ABI 8; NULL: FP32 MFMA): planes hi | mid | lo of W_split_t_plane bf16 elements, row k (input
     * channel) at + k * W_split_t_pitch, columns = the layer's n_out output channels (gad_split_weights).  One group only. */
    const uint16_t* W_split_t;
    int32_t W_split_t_pitch;
    int32_t W_split_t_plane;
} gad_gemm_dx_args;

int gad_gemm_dx(const gad_gemm_dx_args* host_args, void* stream);

typedef struct {
    gad_gemm_fwd_args in;      /* describes how the layer's INPUT rows are produced (W/zout unused) */
    gad_dz_src dz;
    int32_t dz_off[GAD_MAX_GROUPS];
    double* gacc;              /* f64 gradient arena, packed layout; group g at gacc + in.w_off[g]; ADDED to */
    int32_t row_splits;        /* rows are divided over this many blocks (0 -> auto)               */
    float* partial;            /* caller workspace for the per-split partial tiles (NULL -> f64 atomics) */
    int64_t partial_elems;     /* capacity of `partial` in floats                                   */
} gad_gemm_dw_args;

int gad_gemm_dw(const gad_gemm_dw_args* host_args, void* stream);

/* dX and dW of ONE layer in one call: `dx` and `dw` describe the same layer (same gradient source; dw->in = the layer
 * dx writes the gradient of).  The SA1 layers of the update step (>= 32768 rows, 64 input channels, 64 / 128 outputs,
 * BatchNorm+ReLU on both sides: reference core/networks.py:29-51 through pointnet2's SharedMLP) run as one streaming
 * pass that reads z / dY / z_prev once for both products; any other layer runs gad_gemm_dw then gad_gemm_dx on
 * `stream`.  Results are those of the two separate calls (dW: f32 partial sums per workgroup, f64 across them).
 * Round 4: the mid-size layers (SA2 / SA3: >= 2048 rows, 128 / 256 / 512 outputs, K a multiple of 64, incl. the gathered
 * first layers with their scatter epilogue) are fused as well (gemm_bwd_wide_kernel): one staged dZ tile feeds the dX and
 * the dW product.  The fused kernels leave one partial dW block per workgroup in dw->partial and a reduce launch sums
 * them into the arena: on `stream` right behind the kernel, or -- dw->row_splits == GAD_DW_REDUCE_LATER -- by the caller's
 * own gad_gemm_dw_reduce(dx, dw, other_stream) once `other_stream` waits for the kernel, which takes the reduce off the
 * dX chain (the workspace must then stay untouched until that launch has run).                                          */
#define GAD_DW_REDUCE_LATER (-2)
int gad_gemm_bwd(const gad_gemm_dx_args* dx, const gad_gemm_dw_args* dw, void* stream);
/* the deferred reduce of a gad_gemm_bwd(dx, dw) call made with GAD_DW_REDUCE_LATER (same argument blocks); a no-op for a
 * layer gad_gemm_bwd does not fuse (its dW was reduced by that call).                                                     */
int gad_gemm_dw_reduce(const gad_gemm_dx_args* dx, const gad_gemm_dw_args* dw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * D. heads' losses (forward value + gradient wrt head outputs in one pass)
 * ------------------------------------------------------------------------------------------- */

/* Critic phase (core/ddpg.py:61-88,119-130): TD3 target y = r + (1-done)*gamma*min(q1t,q2t),
 * masked smooth-L1 on both Q heads, control-point L1 on the quaternion-normalised aux head.
 *   out9 (B,9) raw head outputs [q1, q2, aux7];  tgt_out9 (B,9) target-critic raw outputs.
 *   g_out9 (B,9) receives dLoss/d(out9).  scalars[0]=critic_loss, [1]=critic_grasp_aux_loss,
 *   [2]=#kept, [3]=#goal rows.  inv_n_* (nullable device scalars) override 1/count with a
 *   globally reduced value for data-parallel runs.                                               */
int gad_critic_loss(const float* out9, const float* tgt_out9, const float* reward,
                    const float* done, const float* perturb_flag, const float* ret,
                    const float* goal, int B, float gamma, int critic_aux, const float* inv_n,
                    float* y, float* aux_norm, float* g_out9, float* scalars, void* stream);

/* Actor phase (core/agent.py:127-139, core/ddpg.py:169-177, core/loss.py): outputs of the policy
 * head pol13 (B,13) = [mean6 raw, extra7 raw].  pi = tanh(mean)*scale + bias (core/networks.py:329-337, 359: action_bias =
 * (high + low) / 2 of the action space; ABI 11: nullable `action_bias`, NULL = symmetric bounds).  bc_loss over expert rows
 * scaled by bc_scale, goal aux over return>0 rows, and (optional) the gradient of
 * -ratio*mean(min(q1_pi,q2_pi)) wrt pi arriving as g_pi_critic (B,6) (already scaled).
 *   g_pol13 receives dLoss/d(pol13); scalars[0]=bc_loss (scaled), [1]=policy_grasp_aux_loss.     */
/* `pitch` = floats per row of the head output / gradient buffers: 6 + extra_pred_dim (13 with policy_aux, 7
 * without: reference core/agent.py:31-36); aux_norm / policy_aux need pitch >= 13.                */
int gad_policy_outputs(const float* pol13, int B, int pitch, const float* action_scale, const float* action_bias /*nullable*/,
                       float* pi, float* aux_norm /*nullable*/, void* stream);
int gad_actor_loss(const float* pol13, const float* pi, const float* expert_action,
                   const float* expert_flag, const float* ret, const float* goal, int B, int pitch,
                   float bc_scale, int policy_aux, const float* action_scale,
                   const double* g_pi_critic /*nullable*/, const float* inv_n, float* g_pol13,
                   float* scalars, void* stream);
/* GaussianPolicy.forward + sample (core/networks.py:339-371) on the raw head outputs head (B,pitch) =
 * [mean 6 | extra extra_dim | log_std 6]: log_std (B,6) clamped to [-10,2]; x = mean + exp(log_std)*eps with the
 * caller's N(0,1) draw eps (B,6) (NULL -> 0); squash != 0: action = tanh(x)*scale+bias, mean_sq = tanh(mean)*scale+bias;
 * log_prob (B) = sum_c N(x; mean, std).log_prob - log(scale*(1-tanh(x)^2)+1e-6); extra (B,extra_dim) with the
 * quaternion part normalised when extra_dim == 7.  Every output is nullable.                      */
int gad_policy_sample(const float* head, int B, int pitch, int extra_dim, const float* eps /*nullable*/,
                      const float* action_scale /*nullable: 1*/, const float* action_bias /*nullable: 0*/, int squash,
                      float* mean_sq, float* log_std, float* log_prob, float* action, float* extra, void* stream);
/* -ratio * mean over rows NOT (expert & return>0) of min(q1,q2): value + dLoss/d(out9[:, :2])    */
int gad_actor_critic_loss(const float* out9, const float* expert_flag, const float* ret, int B,
                          float ratio, const float* inv_n, float* g_out9, float* scalars,
                          void* stream);
/* local mask counts of a minibatch as doubles: out4 = [#(perturb < 1), #(return > 0), #(expert >= 1), #not(expert & reward)]
 * (core/agent.py:224-229): what the masked means divide by; a data-parallel run sums them over the ranks.            */
int gad_mask_counts(const float* ret, const float* expert_flag, const float* perturb_flag, int B, double* out4,
                    void* stream);
/* TD3 target-policy smoothing (core/utils.py:568-576 + core/ddpg.py:80-82, quirk preserved):
 * normal == 0 (noise_type "uniform"): u ~ U[0,1): a = pi + clamp3(((u*3-6)*level) * [1,1,1,5,5,5])
 * normal != 0 (any other noise_type):  u ~ N(0,1): a = pi + clamp3((u*level/2) * [1,1,1,5,5,5])    */
int gad_target_noise(const float* pi, const float* u, int B, float level, int normal, float* out,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * E. optimiser / bookkeeping over flat parameter buffers
 * ------------------------------------------------------------------------------------------- */

/* grad[i] = (float) gacc[m2p[i]] for the n master parameters (m2p[i] < 0 -> 0); accumulate != 0
 * adds to the existing grad instead (a second backward into the same .grad, as autograd does).   */
int gad_grad_from_arena(const double* gacc, const int32_t* m2p, int n, float* grad, int accumulate,
                        void* stream);
/* the same conversion + the sum of squares of the resulting gradient atomically added to *sumsq (f64; zero it first):
 * gad_grad_from_arena followed by gad_sumsq in one launch (the critic phase: clip_grad_norm_'s total norm)                 */
int gad_grad_from_arena_sumsq(const double* gacc, const int32_t* m2p, int n, float* grad, int accumulate, double* sumsq,
                              void* stream);
/* sum of squares of grad[0..n) into *out (f64, atomically accumulated; zero it first)           */
int gad_sumsq(const float* grad, int n, double* out, void* stream);
/* max |x| over segments: out[s] = max |x[seg_off[s] .. seg_off[s+1])|                           */
int gad_absmax_segments(const float* x, const int32_t* seg_off, int n_seg, float* out,
                        void* stream);
/* Adam with L2 weight decay folded into the gradient (torch.optim.Adam, amsgrad=False), over a
 * flat buffer; `hyper` is a device array {lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2,
 * grad_scale}; active[i]==0 skips an element (parameters that never receive a gradient);
 * clip_sumsq/clip_max (nullable) apply clip_grad_norm_ on the fly:
 *   g *= min(1, clip_max / (sqrt(*clip_sumsq) + 1e-6)).
 * Updated values are written to the master buffer p and mirrored into the packed compute buffer
 * packed[m2p[i]].                                                                               */
int gad_adam_step(float* p, const float* grad, float* exp_avg, float* exp_avg_sq,
                  const uint8_t* active, const int32_t* m2p, float* packed, int n,
                  const float* hyper, const double* clip_sumsq, float clip_max, void* stream);
/* target <- (1-tau)*target + tau*source where sel[i]==1, target <- source where sel[i]==2        */
int gad_polyak(float* target, const float* source, const uint8_t* sel, const int32_t* m2p,
               float* target_packed, int n, float tau, int hard_enable, void* stream);
/* One launch for a whole optimiser phase: each job is one flat buffer and, optionally and in this order per element:
 * .grad <- gradient arena (gad_grad_from_arena), Adam step with clip scaling and packed mirror (gad_adam_step; hyper ==
 * NULL: none), target-network update FROM THE UPDATED parameter (gad_polyak; target == NULL: none), max |p| / max |grad|
 * atomically maximised into absmax_p / absmax_grad (GAD_ABSMAX_SLOTS floats each, float bits, zeroed by the caller,
 * who takes the maximum of the slots), and counter[0..counter_n) += counter_add (BatchNorm num_batches_tracked).
 * Same arithmetic as the single-purpose entry points.  Alignment: p, grad, exp_avg, exp_avg_sq, m2p, target, target_m2p 16 bytes;
 * active, target_sel 4 bytes (four elements per access; GAD_ERR_SHAPE otherwise).                                       */
#define GAD_MAX_OPTIM_JOBS 4
#define GAD_ABSMAX_SLOTS 8
typedef struct {
    int32_t n;
    float* p; float* grad; float* exp_avg; float* exp_avg_sq; const uint8_t* active; const int32_t* m2p; float* packed;
    const double* gacc; int32_t accumulate;
    const float* hyper; const double* clip_sumsq; float clip_max;
    float* target; const uint8_t* target_sel; const int32_t* target_m2p; float* target_packed; float tau; int32_t hard_enable;
    float* absmax_p; float* absmax_grad;
    int64_t* counter; int32_t counter_n; int32_t counter_add;
} gad_optim_job;
int gad_optim_jobs(const gad_optim_job* host_jobs, int n_jobs, void* stream);

/* packed[m2p[i]] = p[i] (refresh the compute layout after an external parameter change)          */
int gad_pack_params(const float* p, const int32_t* m2p, int n, float* packed, void* stream);

/* Split-bf16 mirrors of packed weight matrices (ABI 8; the arithmetic behind option "mfma_split": every FP32 product is
 * formed as the six leading bf16 x bf16 term products of hi + mid + lo splits -- 24 significand bits -- accumulated in f32 by
 * v_mfma_f32_32x32x16_bf16).  For each layer, from its packed f32 weights W (n_out, Kp) at packed + w_off, the first Ks
 * columns (Ks a multiple of 32):
 *   forward mirror   out + fwd_off: planes hi | mid | lo of n_out * Ks bf16, row n, column k  (reduction index k contiguous)
 *   transposed mirror out + t_off : planes hi | mid | lo of Ks * n_out bf16, row k, column n  (reduction index n contiguous)
 * (n_out, Kp and the three offsets multiples of 4: the kernel moves 8-byte groups of four bf16)
 * hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid) (round to nearest even; the residuals are exact), and every
 * value whose REDUCTION index lies in an odd block of 16 is stored NEGATED: the kernels add those blocks' products into a
 * second accumulator and subtract it, which cancels the bf16 MFMA adder's truncation bias.  Call it whenever
 * `packed` changes (after gad_pack_params / gad_adam_step / gad_optim_jobs) for networks whose launches carry W_split*. */
#define GAD_MAX_SPLIT_LAYERS 16
typedef struct {
    int32_t w_off;             /* element offset of the layer's weights in `packed`                 */
    int32_t n_out;
    int32_t Kp;                /* row pitch of the packed weights                                   */
    int32_t Ks;                /* leading columns mirrored (multiple of 32, <= Kp)                  */
    int64_t fwd_off;           /* bf16-element offsets into `out`                                   */
    int64_t t_off;
} gad_split_layer;
int gad_split_weights(const float* packed, const gad_split_layer* host_layers, int n_layers, uint16_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * F. replay minibatch gather (the step before the path: reference core/replay_memory.py:109-127,251-272 run as a
 *    device-side index gather over a GPU-resident float32 mirror of the buffer; the index arithmetic -- draw,
 *    successor index, episode end -- stays on the host and is the reference's).  One launch fills the update
 *    step's input buffers: out_*[b] = src[idx[b]], next cloud = point_state[nxt[b]],
 *    time[b] = timestep[end[b]] + 1 - timestep[idx[b]], time_m1 = time - 1.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B;
    int32_t cloud_elems;                 /* floats per transition of point_state (4 * 1030), multiple of 2 */
    const int64_t* idx; const int64_t* nxt; const int64_t* end;           /* (B) device index vectors */
    const float* point_state;            /* (cap, cloud_elems) */
    const float* action; const float* expert_action;                      /* (cap, 6) */
    const float* goal;                   /* (cap, 7) */
    const float* reward; const float* returns; const float* terminal; const float* timestep;
    const float* expert_flags; const float* perturb_flags;                /* (cap) */
    float* out_point; float* out_next_point;                              /* (B, cloud_elems) */
    float* out_action; float* out_expert_action; float* out_goal;
    float* out_reward; float* out_return; float* out_mask; float* out_time; float* out_time_m1;
    float* out_expert_flag; float* out_perturb_flag;
} gad_replay_gather_args;

int gad_replay_gather(const gad_replay_gather_args* host_args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * G. housekeeping
 * ------------------------------------------------------------------------------------------- */

/* Clears up to six device buffers in one launch (pN may be NULL; nN = bytes, multiples of 4).  Replaces the
 * tensor.zero_() calls the reference's autograd does implicitly (fresh .grad / statistics buffers every backward,
 * core/agent.py:261-280 set_mode -> zero_grad): gradient arenas, BatchNorm-backward statistics and the scatter
 * targets of one backward pass. */
int gad_zero_buffers(void* p0, long long n0, void* p1, long long n1, void* p2, long long n2, void* p3,
                     long long n3, void* p4, long long n4, void* p5, long long n5, void* stream);

/* Copies up to GAD_COPY_MAX_SEGS device buffers in one launch (bytes: multiples of 4, 4-byte aligned; a segment with a NULL
 * pointer or 0 bytes is skipped).  add != 0: the segment is float32 data and dst[i] = src[i] + add (the step's "remaining time
 * minus one" vector of the TD target, reference core/ddpg.py:100-104, is formed while the time vector is copied).  Replaces the
 * key-by-key tensor copies with which a device-resident minibatch was adopted into the step's static input buffers
 * (reference core/agent.py:211-240 prepare_data: one .cuda() per key). */
#define GAD_COPY_MAX_SEGS 16
typedef struct {
    void* dst;
    const void* src;
    long long bytes;
    float add;
    int32_t reserved_;
} gad_copy_seg;
int gad_copy_buffers(const gad_copy_seg* host_segs, int n_segs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H. step replay (ABI 11)
 *
 * The update step of reference core/ddpg.py:146-185 / core/agent.py:211-259 is ~240 launches of the entry points above on four
 * streams.  The reference's host walks its step from Python, one torch call after the other
 * (core/train_test_offline.py:117-126: sample, update_parameters, read the losses -- every iteration); a host that does the
 * same over this ABI pays one foreign call per launch.  A gad_plan is the recorded list -- launches with their argument words,
 * device copies / clears, stream forks and joins -- replayed by ONE call: gad_plan_run walks it in C and enqueues every item on
 * the stream of its lane.  Nothing is captured or re-ordered (this is not a HIP graph: measured slower on this part); the
 * launches are exactly those the item-by-item host would have made, in the same order on every stream.
 *
 *   - lanes: small integers naming the streams of a run; gad_plan_run takes the table lane -> hipStream_t.
 *   - argument words: one uint64 per argument of the entry point, the trailing `stream` excluded, with its kind
 *     (GAD_ARG_*: I32 = int / int32_t, I64 = pointers and long long, F32 = float bits in the low half, F64 = double bits).
 *     gad_plan_add_call checks count and kinds against the entry point's real signature and fails with GAD_ERR_SHAPE on a
 *     mismatch.  Pointers to host argument blocks (gad_gemm_fwd_args, gad_optim_job[], ...) are stored as pointers: the
 *     caller keeps the blocks alive and may edit them between runs.
 *   - gad_plan_patch rewrites one word of an item (a scalar that changes from step to step, the pinned block of this
 *     step's ring slot, an event handle); 0 in the event word of a record / wait-event item makes it a no-op.
 *   - gad_plan_arm_timing: gad_timing_slot for ONE launch item of the next run (consumed by it).
 *   - a plan is not thread-safe; two plans may run from two threads.  gad_plan_add_* return the item's index (>= 0).
 * ------------------------------------------------------------------------------------------- */
#define GAD_ARG_I32 0
#define GAD_ARG_I64 1
#define GAD_ARG_F32 2
#define GAD_ARG_F64 3
#define GAD_PLAN_MAX_LANES 16
typedef struct gad_plan gad_plan;
int gad_plan_create(gad_plan** out);
int gad_plan_destroy(gad_plan* plan);
int gad_plan_size(const gad_plan* plan);
int gad_plan_add_call(gad_plan* plan, const char* entry, const uint64_t* words, const uint8_t* kinds, int n_words, int lane);
/* `waiter_lane` waits for everything enqueued so far on `signal_lane` (plan-owned event: record + hipStreamWaitEvent) */
int gad_plan_add_wait(gad_plan* plan, int waiter_lane, int signal_lane);
/* caller-owned hipEvent_t (nullable, patchable word 0): record on / make wait the lane's stream */
int gad_plan_add_record(gad_plan* plan, int lane, void* event);
int gad_plan_add_wait_event(gad_plan* plan, int lane, void* event);
int gad_plan_add_memset(gad_plan* plan, void* dst, long long bytes, int lane);                 /* hipMemsetAsync(dst, 0, bytes) */
int gad_plan_add_memcpy(gad_plan* plan, void* dst, const void* src, long long bytes, int lane); /* hipMemcpyAsync, any direction;
                                                                                                 * words: dst, src, bytes     */
int gad_plan_patch(gad_plan* plan, int item, int word, uint64_t value);
int gad_plan_arm_timing(gad_plan* plan, int item, void* slot);
/* enqueue items [first, first + count) (count < 0: to the end); no host synchronisation.  On failure: the status of the item
 * that failed, gad_last_error() names it. */
int gad_plan_run(gad_plan* plan, void* const* streams, int n_streams, int first, int count);
int gad_plan_entry_count(void);               /* the entry points a plan can replay (every one above that takes a stream) */
const char* gad_plan_entry_name(int i);

#ifdef __cplusplus
}
#endif
#endif /* GADDPG_H */
