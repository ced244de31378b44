// Shared host/device parameter blocks for the rollout kernel.
#pragma once
#include <stdint.h>

#define DN_THREADS 256          // 4 wavefronts of 64 per workgroup
#define DN_WAVES 4
#define DN_MAX_LAYERS 4
#define DN_MAX_SEG 3
#define DN_FCAP 16              // max graph_hid handled by the attention-head accumulators

// One dense layer as the MFMA GEMM sees it: out^T[n][m] = sum_seg W_seg[n][:] . in_seg^T[:][m].
// Weights of a segment are packed per (n-tile of 32, k-group of 8) as 64 lanes x float4:
//   lane l, element i  ->  W[nt*32 + (l&31)][kg*8 + 2*i + (l>>5)]      (zero padded)
struct DnLin {
    int32_t w[DN_MAX_SEG];      // float offset of the packed segment in the weight buffer
    int32_t kg[DN_MAX_SEG];     // number of k-groups (ceil(K/8)) of the segment
    int32_t nseg;
    int32_t n;                  // true output width
    int32_t nt;                 // ceil(n/32)
    int32_t bias;               // float offset of bias[nt*32] (zero padded), -1 = none
    int32_t gather;             // float offset of one-hot table [G][nt*32] (W[:,col0+g]), -1 = none
};

// Small layers evaluated on the VALU: raw row-major W[n][k], b[n]
struct DnRaw {
    int32_t w, b, n, k;
};

struct DnGraph {
    int32_t kind;               // DAGNET_GRAPH_*
    int32_t nin, nout;
    // GAT
    DnLin heads;                // nin -> P*F, no bias
    int32_t a_heads;            // raw [P][2F]
    DnLin out;                  // P*F -> nout, no bias
    int32_t a_out;              // raw [2*nout]
    int32_t bn1;                // raw [4][nout or F]: gamma, beta, mean, sqrt(var+eps)
    // GCN
    DnLin gc1;                  // nin -> F (bias applied after aggregation)
    DnLin gc2;                  // F -> nout
    int32_t b1, b2;             // raw biases [F], [nout]  (-1 = none)
    int32_t bn2;                // raw [4][nout]
};

struct DnGru {
    DnLin r, z;                 // segs: inputs..., then h   (bias = b_i + b_h)
    DnLin ni;                   // segs: inputs              (bias = b_in)
    DnLin nh;                   // seg : h                   (bias = b_hn)
};

struct DnWeights {
    DnRaw phi_x0;               // X -> H on VALU (K = x_dim is tiny)
    DnLin phi_x2, phi_z0, phi_z2;
    DnLin enc0, enc2, enc_ml;
    DnLin prior0, prior2, prior_ml;
    DnLin dec0, dec2;
    DnRaw dec_mean, dec_logvar; // H -> X on VALU
    DnGru gru[DN_MAX_LAYERS];
    DnGraph g_hid;              // graph_hiddens / graph
    DnLin blend;                // lg_hiddens / lg : [h_g ; h] -> R
    DnLin dec_goal0, dec_goal2; // DAG-Net
    DnGraph g_goal;
    DnLin lg_goals;
};

struct DnDims {
    int32_t model, graph, adj_type, knn_k;
    int32_t X, H, Z, G, R, L, F, P, D;
    float alpha;
    double sigma;
};

// LDS layout (float offsets); row stride of every [feature][row] buffer is M.
struct DnLds {
    int32_t M;                  // rows per tile (multiple of 32)
    int32_t AS;                 // adjacency row stride = max agents of a scene in the plan
    int32_t h[DN_MAX_LAYERS];
    int32_t t1, t2, t3, px, pz; // wide temporaries
    int32_t d;                  // disposition [pad8(D)][M]
    int32_t wh, x2;             // graph temporaries [pad32(P*F)][M]
    int32_t decg, gg, gcomb;    // goal buffers [pad32(G)][M]  (alias t2,t3,px,pz)
    int32_t adj, att;           // [M][AS]
    int32_t s12;                // [2][P][M]
    int32_t red;                // [8][M] reduction scratch
    int32_t xabs, dm, xt;       // [2][M], [X][M], [X][M]
    int32_t ri;                 // int region: sstart[M], sn[M], grow[M], gidx[M], gprev[M], krow[M], sid[M]
    int32_t total;              // floats
};

struct DnRun {
    const float *wp;            // packed weights
    const int32_t *scene_ptr;   // device [S+1]
    const int32_t *tile_u0;     // device [n_tiles] first scene-sample unit of the tile
    const int32_t *tile_nu;     // device [n_tiles] number of units
    int32_t S, N, K, T, n_tiles;
    // sample
    const float *h0, *x0;
    const int32_t *g0;
    const float *eps, *q;
    const int32_t *force_goals;
    uint64_t seed;
    float *out;
    int32_t *out_goals;
    // forward
    const float *traj, *traj_rel;
    const int32_t *goals;
    const float *adj;           // per-timestep adjacency
    int64_t adj_ld;             // N for dense, 0 for compact
    int64_t adj_tstride;        // floats between timesteps
    const int64_t *adj_off;     // device [S]: offset of the scene's block (dense: s0*N+s0, compact: prefix of n^2)
    double *loss_part;          // [n_tiles][T][3] partial sums
    float *h_out;
    // graph-only mode
    const float *gx;
    float *gout;
    int32_t gwhich;
    float *gscratch;            // big-G plans: [gblocks][gslab] floats of global scratch, else NULL
    int64_t gslab;
    int32_t gblocks;
    long long *prof;            // DN_PROFILE builds: per-phase cycle counters of tile 0 (else unused)
};

struct DnParams {
    DnDims d;
    DnLds l;
    DnWeights w;
    DnRun r;
};

// operands of dn_launch_nan_guard (aux_kernels.hip): 32-bit words, rows `cols` words apart, `planes` leading planes
struct DnNanSrc {
    const uint32_t *p;        // NULL = absent
    int64_t plane_stride;     // words between planes ([L] or [T] leading dimension)
    int32_t planes, cols;     // rows are `cols` words apart
};
struct DnNanDst {
    uint32_t *p;
    int64_t plane_stride;
    int64_t planes;
    int32_t cols;
};
