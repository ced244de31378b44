/*
 * dagnet_hip.h -- C ABI of libdagnet_hip.so: the MI355X (gfx950) implementation of
 * DAG-Net's per-timestep VRNN rollout (VRNN / A-VRNN / DAG-Net, forward + best-of-K sample).
 *
 * The reference (alexmonti19/dagnet) is pure Python/PyTorch and has no FFI; the boundary it
 * fixes is the set of nn.Module methods listed below.  Each entry point cites the reference
 * interface it sits under.  The Python modules in dagnet_amd/ (same class names, signatures
 * and state-dict keys as the reference) are the only intended callers; INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors, fp32 unless
 *     stated); pointers named h_* are HOST pointers; the caller owns every buffer.
 *   - rows: N = total agents of the batch, scene s owns rows scene_ptr[s]..scene_ptr[s+1]-1
 *     (== the reference's seq_start_end[s] = (start,end)); tensors are time-major [T,N,C].
 *   - every function returns 0 on success, a negative dagnet_status otherwise;
 *     dagnet_last_error() returns a thread-local message for the last failure.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No call synchronises
 *     the host except dagnet_model_create / dagnet_plan_create (which upload small tables).
 */
#ifndef DAGNET_HIP_H
#define DAGNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DAGNET_OK = 0,
    DAGNET_ERR_INVALID = -1,   /* bad argument / unsupported configuration            */
    DAGNET_ERR_MISSING = -2,   /* a state-dict tensor the model needs was not passed   */
    DAGNET_ERR_HIP = -3,       /* a HIP runtime call or kernel launch failed           */
    DAGNET_ERR_CAPACITY = -4   /* scene too large for one workgroup tile (LDS budget)  */
} dagnet_status;

/* DAGNET_MODEL_GRAPH: a stand-alone GAT/GCN module (rnn_dim = nin, h_dim = nout, keys "graph.*") */
enum { DAGNET_MODEL_VRNN = 0, DAGNET_MODEL_AVRNN = 1, DAGNET_MODEL_DAGNET = 2, DAGNET_MODEL_GRAPH = 3 };
enum { DAGNET_GRAPH_NONE = 0, DAGNET_GRAPH_GAT = 1, DAGNET_GRAPH_GCN = 2 };
enum { DAGNET_ADJ_FULL = 0, DAGNET_ADJ_DISTSIM = 1, DAGNET_ADJ_KNN = 2 };

/* Mirrors the `args` namespace the reference constructors read
 * (models/dagnet/model.py:11-29, models/avrnn/model.py:12-27, models/baseline/model.py:8-15). */
typedef struct {
    int32_t model;          /* DAGNET_MODEL_*                                         */
    int32_t x_dim, h_dim, z_dim, rnn_dim, n_layers;
    int32_t g_dim;          /* DAG-Net only                                            */
    int32_t n_max_agents;   /* DAG-Net only: d_dim = 2*n_max_agents                    */
    int32_t graph_model;    /* DAGNET_GRAPH_*                                          */
    int32_t graph_hid;      /* F                                                       */
    int32_t n_heads;        /* P (GAT)                                                 */
    int32_t adjacency_type; /* DAGNET_ADJ_*                                            */
    int32_t top_k_neigh;    /* kNN adjacency                                           */
    float alpha;            /* GAT LeakyReLU slope                                     */
    double sigma;           /* distance-similarity bandwidth (fp64 as in the reference)*/
} dagnet_config;

/* One state-dict entry: reference key name (e.g. "rnn.weight_ih_l0"), host fp32 data,
 * row-major, shape[ndim].  BatchNorm "num_batches_tracked" may be omitted. */
typedef struct {
    const char *name;
    const float *h_data;
    int32_t ndim;
    int64_t shape[4];
} dagnet_tensor;

typedef struct dagnet_model dagnet_model;   /* packed weights on one device            */
typedef struct dagnet_plan dagnet_plan;     /* scene->tile tables for one batch layout */

const char *dagnet_last_error(void);
const char *dagnet_version(void);

/* Replaces: DAGNet.__init__/AVRNN.__init__/VRNN.__init__ + load_state_dict
 * (models/dagnet/model.py:11-116, models/avrnn/model.py:12-84, models/baseline/model.py:8-63).
 * Re-packs the weights into MFMA fragment order on the current device. */
int dagnet_model_create(const dagnet_config *cfg, const dagnet_tensor *tensors, int32_t n_tensors,
                        dagnet_model **out);
void dagnet_model_destroy(dagnet_model *m);

/* Replaces the per-call Python loops over seq_start_end
 * (models/dagnet/model.py:145-147,207-209,265-267; models/utils/adjacency_matrix.py:24,52,89).
 * h_scene_ptr: HOST int32[S+1]; K = number of samples rolled out together (rows = N*K). */
int dagnet_plan_create(const dagnet_model *m, const int32_t *h_scene_ptr, int32_t n_scenes,
                       int32_t n_samples, dagnet_plan **out);
void dagnet_plan_destroy(dagnet_plan *p);
/* rows per tile / number of tiles / LDS bytes per workgroup of the kernel dagnet_sample() runs on this plan (reporting) */
int dagnet_plan_info(const dagnet_plan *p, int32_t *tile_rows, int32_t *n_tiles, int64_t *lds_bytes);
/* which sample() kernel the plan dispatches to: 2 = register-resident (rollout_v2.hip), 3 = wide goal head (batched GEMMs,
 * wide_head.hip + gemm_f32.hip) + register-resident step kernel, 0 = generic LDS-tiled, 1 = generic with goal-width
 * buffers in global scratch (g_dim too large for the LDS); -1 = tables-only plan */
int dagnet_plan_kernel(const dagnet_plan *p);

/* Replaces: K calls of DAGNet.sample / AVRNN.sample / VRNN.sample
 * (models/dagnet/model.py:196-270, models/avrnn/model.py:163-215, models/baseline/model.py:135-170)
 * as they are issued by evaluate() (models/dagnet/evaluate_bsk.py:113-114).
 *   d_h0      [L,N,R]   hidden state after the observed steps (shared by the K samples)
 *   d_x0      [N,2]     last observed absolute position        (A-VRNN, DAG-Net; else NULL)
 *   d_g0      int32[N]  last observed goal index               (DAG-Net; else NULL)
 *   d_eps     [K,T,N,Z] injected U[0,1) noise, or NULL -> Philox stream from `seed`
 *   d_q       [K,T,N,G] injected Exp(1) variates for the goal draw (DAG-Net): goal = argmax_g w_g / q_g, torch.multinomial's
 *                       exponential race; or NULL -> one uniform per (sample, step, row) from the Philox stream of `seed` and
 *                       the inverse of the cumulative weights (the same categorical distribution, csrc/rng.h rng_draw_u)
 *   d_force_goals int32[K,T,N] or NULL: overrides the categorical draw (tests)
 *   d_out     [K,T,N,2] predicted displacements (the reference's `samples`)
 *   d_out_goals int32[K,T,N] or NULL: sampled goal index per step (DAG-Net)
 * A NaN in d_h0 comes out as the reference's arithmetic would let it: NaN for that row (VRNN), and with a graph NaN for
 * the row at step 0 and for the whole batch from step 1 on (the reference's attention / adjacency products are dense over
 * the block-diagonal batch).  A plan whose sample() works in buffers of the plan (dagnet_plan_kernel() 1 or 3) refuses a
 * call while an earlier call on ANOTHER stream is still running (DAGNET_ERR_INVALID); calls on one stream are ordered. */
int dagnet_sample(const dagnet_model *m, const dagnet_plan *p, int32_t T,
                  const float *d_h0, const float *d_x0, const int32_t *d_g0,
                  const float *d_eps, const float *d_q, const int32_t *d_force_goals,
                  uint64_t seed, float *d_out, int32_t *d_out_goals, void *stream);

/* Replaces: DAGNet.forward / AVRNN.forward / VRNN.forward in eval mode
 * (models/dagnet/model.py:138-194, models/avrnn/model.py:93-143, models/baseline/model.py:70-115).
 * The plan must have been created with n_samples = 1.
 *   d_traj     [T,N,2] absolute positions (DAG-Net; else NULL)
 *   d_traj_rel [T,N,2] relative coords
 *   d_goals    int32[T,N] goal indices (DAG-Net; else NULL)
 *   d_adj      adjacency per timestep (A-VRNN, DAG-Net), either dense [T_adj,N,N]
 *              (adj_dense_ld = N) or compact per-scene blocks [T_adj, sum n_s^2] (adj_dense_ld = 0);
 *              timestep t of the loop reads slice t.
 *   d_eps      [T-1,N,Z] injected noise for steps 1..T-1, or NULL -> Philox
 *   d_losses   double[3]: KLD, NLL, CE accumulated as the reference does (mean over rows per
 *              step for KLD/NLL, sum for CE)
 *   d_h_out    [L,N,R]
 * A NaN in d_traj_rel / d_traj makes the three losses NaN, and d_h_out NaN for the row (VRNN) or the batch (graph models).
 * The losses are summed in buffers of the plan: a call that arrives on another stream than the previous call on this plan is
 * ordered behind it on the device (it waits for that call's completion event), so calls on one plan never overlap - use one
 * plan per stream where they should.  Streams under graph capture are not tracked.            */
int dagnet_forward(const dagnet_model *m, const dagnet_plan *p, int32_t T,
                   const float *d_traj, const float *d_traj_rel, const int32_t *d_goals,
                   const float *d_adj, int64_t adj_dense_ld, const float *d_eps, uint64_t seed,
                   double *d_losses, float *d_h_out, void *stream);

/* Replaces: GAT.forward / GCN.forward on block-diagonal adjacency
 * (models/dagnet/gat/gat_model.py:18-22, gat_layers.py:25-40, gcn/gcn_model.py:14-19,
 * gcn_layers.py:20-26), eval-mode BatchNorm.  `which` = 0 hidden-state graph
 * (graph_hiddens / graph), 1 = goal graph (graph_goals).  d_x [N,nin] -> d_out [N,nout].
 * d_adj as in dagnet_forward (single timestep). */
int dagnet_graph_forward(const dagnet_model *m, const dagnet_plan *p, int32_t which,
                         const float *d_x, const float *d_adj, int64_t adj_dense_ld,
                         float *d_out, void *stream);

/* Replaces: adjs_fully_connected_pred / adjs_distance_sim_pred / adjs_knn_sim_pred and
 * compute_adjs / compute_adjs_distsim / compute_adjs_knnsim + block_diag_irregular
 * (models/utils/adjacency_matrix.py:21-28,49-58,86-103,108-152; models/utils/utils.py:290-302).
 * Scene layout comes from the plan.  d_pos [T,N,2] (ignored for DAGNET_ADJ_FULL);
 * output dense [T,N,N] (dense_ld = N, zero-filled off the blocks by this call)
 * or compact [T, sum n_s^2] (dense_ld = 0; size per timestep = dagnet_plan_compact_adj_size). */
int dagnet_adjacency(const dagnet_plan *p, int32_t adjacency_type, double sigma, int32_t top_k,
                     int32_t T, const float *d_pos, float *d_out, int64_t dense_ld, void *stream);
int64_t dagnet_plan_compact_adj_size(const dagnet_plan *p);

/* Replaces: relative_to_abs + average/final_displacement_error(mode='raw') + evaluate_helper
 * (models/utils/utils.py:8-21,76-111; models/dagnet/evaluate_bsk.py:71-82,115-121).
 *   d_samples_rel [K,T,N,2], d_start [N,2], d_gt [T,N,2]
 *   d_ade_raw/d_fde_raw [K,N]   per-sample per-agent errors (outputs)
 *   d_scene_best  [S,2]         per scene: min over K of the scene-summed ADE / FDE (output)
 *   d_sums double[2]            sum over scenes of d_scene_best (fp64, fixed order)        */
int dagnet_best_of_k(const dagnet_plan *p, int32_t K, int32_t T, const float *d_samples_rel,
                     const float *d_start, const float *d_gt, float *d_ade_raw, float *d_fde_raw,
                     float *d_scene_best, double *d_sums, void *stream);

/* Replaces: relative_to_abs (models/utils/utils.py:8-21): d_out[t] = d_start + cumsum_t(d_rel),
 * all [T,N,2] / [N,2]. */
int dagnet_relative_to_abs(int32_t T, int32_t N, const float *d_rel, const float *d_start,
                           float *d_out, void *stream);

/* Replaces: evaluate_helper (models/dagnet/evaluate_bsk.py:71-82) for two error arrays at once:
 * d_scene_best[s] = (min_k sum_{agents of s} ade[k,.], min_k sum fde[k,.]); d_sums = fp64 sums over scenes. */
int dagnet_scene_min_sum(const dagnet_plan *p, int32_t K, const float *d_ade_raw, const float *d_fde_raw,
                         float *d_scene_best, double *d_sums, void *stream);

/* Replaces: average_displacement_error / final_displacement_error with mode='raw'
 * (models/utils/utils.py:76-111) on absolute coordinates d_pred, d_gt [T,N,2]:
 * d_ade[n] = sum_t |gt - pred|_2, d_fde[n] = |gt_T - pred_T|_2. */
int dagnet_displacement_errors(int32_t T, int32_t N, const float *d_pred, const float *d_gt,
                               float *d_ade, float *d_fde, void *stream);

/* Replaces: sample_multinomial's torch.multinomial(probs, 1) (models/utils/utils.py:200-207):
 * d_out[n] = argmax_g d_weights[n,g] / q[n,g] with q ~ Exp(1) taken from d_q [N,G]; when d_q is NULL, one uniform u per row from
 * the Philox stream of `seed` and d_out[n] = min { g : w_0 + .. + w_g > u (w_0 + .. + w_{G-1}) } (negative weights count as 0;
 * a column of weight 0 is never drawn) - the same distribution. */
int dagnet_categorical(int32_t N, int32_t G, const float *d_weights, const float *d_q, uint64_t seed,
                       int32_t *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Training-step operators.  The reference's training step is torch autograd over
 * DAGNet.forward / AVRNN.forward / VRNN.forward in train() mode (models/dagnet/train_bsk.py:259-291,
 * models/dagnet/model.py:138-194).  These are the forward and hand-written backward kernels of every
 * layer of that pass; dagnet_amd/_ops.py binds each pair into one torch.autograd.Function.
 * All matrices: device fp32, row-major, explicit row stride (ld*, in floats).  Every function returns a
 * dagnet_status and launches on `stream` without synchronising.
 * ------------------------------------------------------------------------------------------------- */

/* Replaces: nn.Linear forward/backward (F.linear and its autograd; models/dagnet/model.py:60-116) and the
 * x @ W products of the graph layers (gat_layers.py:26, gcn_layers.py:21).
 * C[M,N] = act([C +] A[M,K] op(B) + bias + gather[gidx[m]*gs_g + n*gs_n]); op(B) = B[N,K]^T (transb 0) or B[K,N]
 * (transb 1).  act: 0 none, 1 LeakyReLU(slope), 2 tanh, 3 ReLU.  One-hot inputs are the gather term. */
int dagnet_op_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, int32_t transb, const float *bias, float *C,
                   int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, float slope, int32_t accumulate,
                   const float *gather, const int32_t *gidx, int64_t gs_g, int64_t gs_n, void *stream);
/* Weight gradients: C[N,K] (+)= A[M,N]^T B[M,K], deterministic two-pass reduction over the rows; dbias (nullable):
 * [N] (+)= column sums of A from the same pass (the bias gradient).  workspace: dagnet_op_gemm_tn_workspace(M, N, K) floats. */
int64_t dagnet_op_gemm_tn_workspace(int32_t M, int32_t N, int32_t K);
int dagnet_op_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float *dbias, int32_t M,
                      int32_t N, int32_t K, float *workspace, int32_t accumulate, void *stream);
/* The same product accumulated over many calls (the time steps of one backward pass) with ONE final reduction: _partial adds
 * this call's split partial sums to `part` (first != 0 overwrites), _finish reduces them into C (+)= and dbias (+)=.
 * B = NULL with bidx != NULL: one-hot second operand generated from the indices.  M, N, K must not change between calls;
 * part: dagnet_op_gemm_tn_workspace(M, N, K) floats owned by the caller for the whole accumulation. */
int dagnet_op_gemm_tn_partial(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, int32_t want_bias,
                              int32_t M, int32_t N, int32_t K, float *part, int32_t first, void *stream);
int dagnet_op_gemm_tn_finish(const float *part, float *C, int64_t ldc, float *dbias, int32_t M, int32_t N, int32_t K,
                             int32_t accumulate, void *stream);
/* dP = dY * act'(.) through the saved layer output Y (backward of LeakyReLU / ReLU / tanh). */
int dagnet_op_act_bwd(int64_t M, int32_t N, const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp,
                      int32_t act, float slope, void *stream);
/* Replaces: the autograd of `act(F.linear(x, W, b))` from the gradient of the layer's OUTPUT down to the gradient of its input
 * (models/dagnet/model.py:60-116 under loss.backward(), train_bsk.py:288): dP[M,K] = dY * act'(Y) (Y = the saved output, K = the
 * layer's width) and C[M,N] = [C +] dP op(B) in one pass over dY and Y; dP stays for the weight-gradient product. */
int dagnet_op_gemm_dact(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp, int32_t act, float slope,
                        const float *B, int64_t ldb, int32_t transb, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                        int32_t accumulate, void *stream);
/* dst (+)= src for an [M,N] block (concatenations and their backward). */
int dagnet_op_copy2d(int64_t M, int32_t N, const float *src, int64_t lds, float *dst, int64_t ldd, int32_t accumulate,
                     void *stream);
/* Gradient of the G weight columns behind a one-hot input: dW[n][g] (+)= sum_{m: gidx[m] = g} dP[m][n] (a weight-gradient
 * GEMM whose second operand is generated from the indices).  workspace: dagnet_op_gemm_tn_workspace(M, N, G) floats. */
int dagnet_op_onehot_grad(int64_t M, int32_t N, int32_t G, const float *dP, int64_t ldp, const int32_t *gidx, float *dW,
                          int64_t ldw, float *workspace, int32_t accumulate, void *stream);
/* Replaces: nn.Softmax(dim=-1) in dec_goal / lg_goals (models/dagnet/model.py:66,100) and its backward. */
int dagnet_op_softmax_fwd(int64_t M, int32_t N, const float *X, int64_t ldx, float *Y, int64_t ldy, void *stream);
int dagnet_op_softmax_bwd(int64_t M, int32_t N, const float *Y, int64_t ldy, const float *dY, int64_t ldd, float *dX, int64_t ldx,
                          void *stream);
/* Replaces: the gate arithmetic of nn.GRU (models/dagnet/model.py:116,185); gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh
 * [M,3R] come from dagnet_op_gemm.  gates [M,3R] keeps r, z, n for the backward. */
int dagnet_op_gru_fwd(int64_t M, int32_t R, const float *gi, const float *gh, const float *h, float *hn, float *gates, void *stream);
/* Replaces: the two-layer blocks phi_x / phi_z / enc / prior / dec (models/dagnet/model.py:60-116: Linear, LeakyReLU, Linear,
 * LeakyReLU) applied to a concatenated input that is never materialised: y1 = act1([x_0 | x_1 | x_2 | one-hot(gidx)] W1^T + b1),
 * y2 = act2(y1 W2^T + b2).  x, ldx, K, cols: HOST arrays of npieces (<= 3) device pointers, row strides, widths and first
 * columns in W1; gidx (nullable): the one-hot piece at column gcol.  Both outputs are written (the backward pass reads y1).
 * 64 / 64 features with >= 2048 rows run as ONE kernel (csrc/mlp_fused.hip); other shapes layer by layer. */
int dagnet_op_mlp2_fwd(int64_t M, int32_t npieces, const float *const *x, const int64_t *ldx, const int32_t *K, const int32_t *cols,
                       const float *W1, int64_t ldw1, const float *b1, int32_t N1, const int32_t *gidx, int32_t gcol, int32_t act1,
                       float slope1, const float *W2, int64_t ldw2, const float *b2, int32_t N2, int32_t act2, float slope2, float *y1,
                       int64_t ldy1, float *y2, int64_t ldy2, void *stream);
/* dagnet_op_mlp2_fwd with optional layers: W2 = NULL stops after y1 (lg_hiddens / lg: a Linear over [graph output | h],
 * models/dagnet/model.py:96,190); Wm, Wl != NULL append the two heads that read y2, mu = y2 Wm^T + bm and lv = y2 Wl^T + bl, [M,Z]
 * each (enc_mean / enc_logvar, prior_*, dec_*: model.py:75-76,83-84,92-93), as a third product of the same kernel. */
int dagnet_op_mlp_chain_fwd(int64_t M, int32_t npieces, const float *const *x, const int64_t *ldx, const int32_t *K, const int32_t *cols,
                            const float *W1, int64_t ldw1, const float *b1, int32_t N1, const int32_t *gidx, int32_t gcol, int32_t act1,
                            float slope1, const float *W2, int64_t ldw2, const float *b2, int32_t N2, int32_t act2, float slope2, float *y1,
                            int64_t ldy1, float *y2, int64_t ldy2, const float *Wm, int64_t ldwm, const float *bm, const float *Wl,
                            int64_t ldwl, const float *bl, int32_t Z, float *mu, int64_t ldmu, float *lv, int64_t ldlv, void *stream);
/* Replaces: autograd's input gradients of lg_hiddens / lg (a Linear with 64 outputs and no activation over [graph output | h],
 * models/dagnet/model.py:96,190): dxA (+)= dy W[:, cA : cA + KA], dxB (+)= dy W[:, cB : cB + KB] (dxB NULL: one piece) from one pass
 * over dy [M,64].  Returns 1 (nothing written) when the shape has no fused kernel (>= 2048 rows, widths multiples of 4 up to 64,
 * 16-byte rows); switched by bit 2 of dagnet_op_set_fused. */
int dagnet_op_linear_pieces_bwd(int64_t M, const float *dy, int64_t lddy, const float *W, int64_t ldw, float *dxA, int64_t lddxA,
                                int32_t cA, int32_t KA, int32_t accA, float *dxB, int64_t lddxB, int32_t cB, int32_t KB, int32_t accB,
                                void *stream);
/* Selects the fused training kernels at run time: bit 0 = dagnet_op_gru_layer_fwd's one-kernel form, bit 1 = dagnet_op_mlp_chain_fwd's,
 * bit 2 = dagnet_op_mlp2_bwd's (a cleared bit makes the entry point take its layer-by-layer form / return 1); bit 3 = split
 * products on the 16-bit matrix pipe, fp32 in / out / accumulation, fp32-MFMA-class error: three bf16 terms per value and six
 * products in the wide layers (more than 96 outputs) of dagnet_op_gemm; two fp16 terms per value and three products in the wide
 * layers of the g_dim = 960 goal head behind dagnet_sample / dagnet_forward (operands of order one or below by construction;
 * DAGNET_WIDE_BF16X3 in the environment keeps three bf16 terms there) and in every dense and graph product of the
 * register-resident rollout kernels behind dagnet_sample (csrc/rollout_v2.hip; a model with a weight beyond fp16's range keeps
 * the fp32-MFMA kernels, which are also what a cleared bit selects); bit 4 (never on by default; needs bit 3) = two fp16 terms
 * in dagnet_op_gemm's wide layers too - fp16 overflows at 65 504 and carries an absolute representation error of up to 3e-8 per
 * operand value below 0.125, so this is for tests and microbenchmarks of that form; mask < 0 restores the default (bits 0-3 on, minus
 * DAGNET_NO_GRU_FUSED / DAGNET_NO_MLP_FUSED / DAGNET_NO_MLP_BWD_FUSED / DAGNET_NO_BF16X3).  -> the previous mask.
 * No counterpart in the reference: a switch for A/B comparisons and tests. */
int dagnet_op_set_fused(int32_t mask);
/* Replaces: autograd's backward of such a block (and of its heads) between the gradients that arrive at it - dmu, dlv [M,Z] at the
 * heads (dy2 = NULL), or dy2 [M,64] at the block output - and the gradients of its input pieces: dp2 = (dmu Wm + dlv Wl | dy2) *
 * act2'(y2), dp1 = (dp2 W2) * act1'(y1) (both written: the layers' weight gradients are products with them), dxA / dxB (+)= dp1
 * W1[:, cA : cA + KA] / [:, cB : cB + KB] (NULL: not wanted; acc: add to what is there).  Returns 1 (nothing written) when the
 * shape has no fused kernel (csrc/mlp_fused.hip: 64 / 64 features, >= 2048 rows, Z in {1, 2, 16, 32}, 16-byte rows). */
int dagnet_op_mlp2_bwd(int64_t M, const float *dy2, int64_t lddy2, const float *dmu, int64_t lddmu, const float *dlv, int64_t lddlv,
                       int32_t Z, const float *Wm, int64_t ldwm, const float *Wl, int64_t ldwl, const float *y2, int64_t ldy2,
                       const float *y1, int64_t ldy1, const float *W2, int64_t ldw2, int32_t act2, float slope2, const float *W1,
                       int64_t ldw1, int32_t act1, float slope1, float *dp2, int64_t lddp2, float *dp1, int64_t lddp1, float *dxA,
                       int64_t lddxA, int32_t cA, int32_t KA, int32_t accA, float *dxB, int64_t lddxB, int32_t cB, int32_t KB,
                       int32_t accB, void *stream);
/* Replaces: one layer of nn.GRU at seq_len = 1 from its inputs (models/dagnet/model.py:109-110,:186): gi = [x0 | x1] W_ih^T + b_ih,
 * gh = h W_hh^T + b_hh and the gate arithmetic in one call (x1 = NULL, K1 = 0: a single input piece).  gi, gh [M,3R], hn [M,R],
 * gates [M,3R] are contiguous.  rnn_dim 64 with 64-wide pieces and >= 2048 rows runs as ONE kernel (csrc/gru_fused.hip) that
 * leaves gi untouched and writes only columns 2R..3R of gh - all that dagnet_op_gru_bwd reads; other shapes fill both.
 * pack (nullable): the layer's weights in the kernel's LDS order, dagnet_op_gru_pack_floats(R, K0, K1) floats written by
 * dagnet_op_gru_pack - valid until the weights change (a training step packs once for all its time steps); without it every
 * workgroup gathers the image itself. */
int dagnet_op_gru_layer_fwd(int64_t M, int32_t R, const float *x0, int64_t ldx0, int32_t K0, const float *x1, int64_t ldx1, int32_t K1,
                            const float *h, int64_t ldh, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                            int64_t ldwhh, const float *bhh, float *gi, float *gh, float *hn, float *gates, const float *pack,
                            void *stream);
int64_t dagnet_op_gru_pack_floats(int32_t R, int32_t K0, int32_t K1);
int dagnet_op_gru_pack(int32_t R, int32_t K0, int32_t K1, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                       int64_t ldwhh, const float *bhh, float *out, void *stream);
int dagnet_op_gru_bwd(int64_t M, int32_t R, const float *dhn, const float *gates, const float *gh, const float *h, float *dgi,
                      float *dgh, float *dh, void *stream);
/* Replaces: autograd's backward of that layer step (models/dagnet/model.py:109-110,:186: nn.GRU's gate gradients and the input
 * gradients of its two products) as ONE kernel for rnn_dim 64 with 64-wide pieces and >= 2048 rows: from dhn [M,R] (gradient of
 * h'), the saved gates [M,3R], gh [M,3R] (columns 2R..3R) and h to the gate gradients dgi, dgh [M,3R] (contiguous; the
 * weight-gradient products read them) and dx0 / dx1 (+)= dgi W_ih[:, piece], dh (+)= dhn z + dgh W_hh (NULL: not wanted; acc: add
 * to what is there).  pack: dagnet_op_gru_pack_bwd_floats(R, K0, K1) floats written by dagnet_op_gru_pack_bwd, valid until the
 * weights change.  Returns 1 (nothing written) when the shape has no fused kernel, without a pack, or with bit 0 of
 * dagnet_op_set_fused cleared / DAGNET_NO_GRU_BWD_FUSED set: the caller then runs dagnet_op_gru_bwd and the products. */
int64_t dagnet_op_gru_pack_bwd_floats(int32_t R, int32_t K0, int32_t K1);
int dagnet_op_gru_pack_bwd(int32_t R, int32_t K0, int32_t K1, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh,
                           float *out, void *stream);
int dagnet_op_gru_layer_bwd(int64_t M, int32_t R, int32_t K0, int32_t K1, const float *dhn, int64_t lddhn, const float *gates,
                            const float *gh, const float *h, int64_t ldh, const float *pack, float *dgi, float *dgh, float *dx0,
                            int64_t lddx0, int32_t acc0, float *dx1, int64_t lddx1, int32_t acc1, float *dh, int64_t lddh, int32_t acch,
                            void *stream);
/* Replaces: _reparameterize + _kld (models/dagnet/model.py:118-128): z = eps exp(logvar_e / 2) + mean_e, kld_row[m] =
 * per-row KLD (the caller takes the mean with dagnet_op_sum); backward from dz and the scalar gradient g_kld[0] * scale. */
int dagnet_op_vae_fwd(int64_t M, int32_t Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                      const float *eps, float *z, float *kld_row, void *stream);
int dagnet_op_vae_bwd(int64_t M, int32_t Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                      const float *eps, const float *dz, const float *g_kld, float scale, float *dmu_e, float *dlv_e,
                      float *dmu_p, float *dlv_p, void *stream);
/* Replaces: _nll_gauss (models/dagnet/model.py:130-135), per row, and its backward. */
int dagnet_op_nll_fwd(int64_t M, int32_t X, const float *mu, const float *lv, const float *x, float *nll_row, void *stream);
int dagnet_op_nll_bwd(int64_t M, int32_t X, const float *mu, const float *lv, const float *x, const float *g, float scale,
                      float *dmu, float *dlv, void *stream);
/* Replaces: cross_entropy -= sum(g_t * g_combined) (models/dagnet/model.py:162): ce_row[m] = -p[m][goal[m]]. */
int dagnet_op_ce_fwd(int64_t M, const float *p, int64_t ldp, const int32_t *goal, float *ce_row, void *stream);
int dagnet_op_ce_bwd(int64_t M, int32_t G, const int32_t *goal, const float *g, float *dp, int64_t ldp, void *stream);
/* The backward of ce_row = -softmax(x)[m][goal[m]] (lg_goals' softmax followed by the probability "cross entropy", model.py:100,162) in
 * one pass from the saved probabilities p: dx[m][c] = g[0] p[m][c] (p[m][goal[m]] - [c == goal[m]]). */
int dagnet_op_softmax_ce_bwd(int64_t M, int32_t G, const float *p, int64_t ldp, const int32_t *goal, const float *g, float *dx, int64_t ldx,
                             void *stream);
/* out[0] (+)= scale * sum(x) in fp64, fixed order (loss scalars). */
int dagnet_op_sum(int64_t n, const float *x, float scale, float *out, int32_t accumulate, void *stream);
/* Column sums over the rows (bias gradients): out[c] (+)= sum_m X[m][c]; workspace: dagnet_op_col_workspace(M, N) floats. */
int64_t dagnet_op_col_workspace(int64_t M, int32_t N);
int dagnet_op_colsum(int64_t M, int32_t N, const float *X, int64_t ldx, float *out, int32_t accumulate, float *workspace,
                     void *stream);
/* No counterpart in the reference: dagnet_op_colsum for callers that cannot allocate or pass a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels and result as dagnet_op_colsum; the
 * split partial sums live in a static device array of the library (4 MiB per GPU, held until the process ends), so N is
 * limited to 1024 columns (DAGNET_ERR_CAPACITY beyond).  Calls no hipMalloc / hipFree and does not synchronise.  All calls
 * on one GPU share that array: calls that overlap on one GPU (different streams) race. */
int dagnet_op_colsum_static(int64_t M, int32_t N, const float *X, int64_t ldx, float *out, int32_t accumulate, void *stream);
/* Replaces: nn.BatchNorm1d in train() mode followed by tanh (gat_model.py:21-22, gcn_model.py:15-19): batch statistics
 * over all M rows, running statistics updated in place (momentum, unbiased variance).  stat [2N] keeps mean | invstd,
 * tmp [2N] and sums [2N] are scratch. */
int dagnet_op_bn_fwd(int64_t M, int32_t N, const float *X, int64_t ldx, const float *gamma, const float *beta, float eps,
                     float momentum, int32_t tanh_out, float *Y, int64_t ldy, float *stat, float *run_mean, float *run_var,
                     float *tmp, float *workspace, void *stream);
int dagnet_op_bn_bwd(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *dOut,
                     int64_t ldd, const float *Out, int64_t ldo, int32_t tanh_out, float *dX, int64_t ldg, float *dgamma,
                     float *dbeta, int32_t accumulate, float *sums, float *workspace, void *stream);
/* Replaces: the same BatchNorm when the batch is sharded over ranks (data-parallel training): the reference's
 * nn.BatchNorm1d takes its statistics over ALL rows of the batch (gat_model.py:21, gcn_model.py:16,18), so the shards exchange
 * mergeable moments.  The stages between which the caller runs its collectives:
 *   dagnet_op_bn_moments  -> mom [3N] = count | mean | sum of squared deviations of this shard's M rows (M = 0: zeros);
 *   (all-gather mom over the ranks: parts [n_parts][3N], rank order)
 *   dagnet_op_bn_merge    -> stat [2N] = mean | invstd over all rows (fp64 merge in rank order: every rank gets the same bits),
 *                            running statistics updated as nn.BatchNorm1d does with the whole batch, total[0] = all rows;
 *   dagnet_op_bn_apply    -> Y = (X - mean) invstd gamma + beta (+ tanh);
 *   dagnet_op_bn_bwd_sums -> sums [2N] = sum g | sum g xhat over this shard (= its share of dbeta | dgamma);
 *   (all-reduce(SUM) sums over the ranks)
 *   dagnet_op_bn_bwd_apply-> dX from the all-reduced sums and Mtot = all rows.
 * tmp [2N] scratch; workspace: dagnet_op_col_workspace(M, N) floats. */
int dagnet_op_bn_moments(int64_t M, int32_t N, const float *X, int64_t ldx, float *mom, float *tmp, float *workspace, void *stream);
int dagnet_op_bn_merge(int32_t N, int32_t n_parts, const float *parts, float eps, float momentum, float *stat, float *run_mean,
                       float *run_var, float *total, void *stream);
int dagnet_op_bn_apply(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *beta,
                       int32_t tanh_out, float *Y, int64_t ldy, void *stream);
int dagnet_op_bn_bwd_sums(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *dOut, int64_t ldd,
                          const float *Out, int64_t ldo, int32_t tanh_out, float *sums, float *workspace, void *stream);
int dagnet_op_bn_bwd_apply(int64_t M, int64_t Mtot, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma,
                           const float *dOut, int64_t ldd, const float *Out, int64_t ldo, int32_t tanh_out, const float *sums, float *dX,
                           int64_t ldg, void *stream);
/* Replaces: GraphAttentionLayer.forward (gat_layers.py:25-40) after Wh = x W, per scene of the plan, for H heads of width F
 * stacked along the columns of Wh; a [H][2F].  adj: compact block-diagonal (adj_ld 0) or dense [N,N] (adj_ld N).
 * Backward: dWh and da_part [S][H*2F], every scene's contribution to the gradient of a (sum its columns with
 * dagnet_op_colsum). */
int dagnet_op_gat_attn_fwd(const dagnet_plan *p, const float *adj, int64_t adj_ld, const float *Wh, int64_t ldw, const float *a,
                           int32_t H, int32_t F, float alpha, int32_t elu, float *Out, int64_t ldo, void *stream);
int dagnet_op_gat_attn_bwd(const dagnet_plan *p, const float *adj, int64_t adj_ld, const float *Wh, int64_t ldw, const float *a,
                           int32_t H, int32_t F, float alpha, int32_t elu, const float *Out, int64_t ldo, const float *dOut,
                           int64_t ldd, float *dWh, int64_t ldg, float *da_part, void *stream);
/* Replaces: torch.spmm(adj, support) + bias of GraphConvolution.forward (gcn_layers.py:20-26) per scene; transpose = 1
 * multiplies by adj^T (the backward). */
int dagnet_op_adj_matmul(const dagnet_plan *p, int32_t F, const float *adj, int64_t adj_ld, const float *X, int64_t ldx,
                         const float *bias, int32_t transpose, float *Y, int64_t ldy, void *stream);
/* Replaces: the disposition tensor d of DAGNet.forward (models/dagnet/model.py:141-147) for one timestep: out[r] = flattened
 * xy of the agents of r's scene, zero padded to D columns. */
int dagnet_op_disposition(const dagnet_plan *p, int32_t D, const float *pos, float *out, int64_t ldo, void *stream);
/* Replaces: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (models/dagnet/train_bsk.py:289-290).
 * dagnet_op_sumsq: acc[0] += sum g^2 (fp64).  dagnet_op_adam: g *= min(1, clip / (sqrt(sumsq[0]) + 1e-6)) (clip <= 0: off),
 * then the Adam update of p with moments m, v at step `step` (1-based). */
int dagnet_op_sumsq(int64_t n, const float *g, double *acc, void *stream);
int dagnet_op_adam(int64_t n, float *p, float *g, float *m, float *v, const double *sumsq, float clip, float lr, float beta1,
                   float beta2, float eps, int32_t step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DAGNET_HIP_H */
