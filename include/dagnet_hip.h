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
/* rows per workgroup tile / number of tiles chosen for the plan (for reporting) */
int dagnet_plan_info(const dagnet_plan *p, int32_t *tile_rows, int32_t *n_tiles, int64_t *lds_bytes);
/* which sample() kernel the plan dispatches to: 2 = register-resident (rollout_v2.hip), 0 = generic LDS-tiled,
 * 1 = generic with goal-width buffers in global scratch (g_dim too large for the LDS); -1 = tables-only plan */
int dagnet_plan_kernel(const dagnet_plan *p);

/* Replaces: K calls of DAGNet.sample / AVRNN.sample / VRNN.sample
 * (models/dagnet/model.py:196-270, models/avrnn/model.py:163-215, models/baseline/model.py:135-170)
 * as they are issued by evaluate() (models/dagnet/evaluate_bsk.py:113-114).
 *   d_h0      [L,N,R]   hidden state after the observed steps (shared by the K samples)
 *   d_x0      [N,2]     last observed absolute position        (A-VRNN, DAG-Net; else NULL)
 *   d_g0      int32[N]  last observed goal index               (DAG-Net; else NULL)
 *   d_eps     [K,T,N,Z] injected U[0,1) noise, or NULL -> Philox stream from `seed`
 *   d_q       [K,T,N,G] injected Exp(1) variates for the goal draw (DAG-Net), or NULL
 *   d_force_goals int32[K,T,N] or NULL: overrides the categorical draw (tests)
 *   d_out     [K,T,N,2] predicted displacements (the reference's `samples`)
 *   d_out_goals int32[K,T,N] or NULL: sampled goal index per step (DAG-Net)            */
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
 *   d_h_out    [L,N,R]                                                                   */
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
 * d_out[n] = argmax_g d_weights[n,g] / q[n,g] with q ~ Exp(1) taken from d_q [N,G] or, when
 * d_q is NULL, from the Philox stream of `seed`. */
int dagnet_categorical(int32_t N, int32_t G, const float *d_weights, const float *d_q, uint64_t seed,
                       int32_t *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DAGNET_HIP_H */
