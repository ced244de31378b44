// Parameter block of the register-resident rollout kernel (rollout_v2.hip).
//
// Weight segments are packed for 32x32x2 fp32 MFMA chains whose k index is the feature index of
// an accumulator tile: for output tile nt, input tile kt, group j (0..3), lane l (h = l>>5, n = l&31)
//     float4 { W[nt*32 + n][kt*32 + 8*j + 4*h + i] , i = 0..3 }      at  ((nt*KT + kt)*4 + j)*64 + l
// so that MFMA number 4*j+i of the (nt,kt) block multiplies register 4*j+i of the input tile,
// which holds feature (i + 8*j) + 4*h of that tile in lane half h.
#pragma once
#include <stdint.h>

struct V2Lin {
    int32_t w[3];      // float offsets of up to 3 chain-packed segments
    int32_t bias;      // [NT*32] zero padded, -1 = none
    int32_t gather;    // one-hot table [G][NT*32], -1 = none
};

// GCN models reuse the slots (api.hip v2_gcn, v2_common.h gcn<>): heads = W1, fold = [2][32] layer-1 scale | shift,
// out = W2, bn = [2][NT*32] layer-2 scale | shift.
struct V2Gat {
    int32_t heads;     // chain-packed [1 tile out (P*F = 32)][KT]       (S->T product)
    int32_t fold;      // chain-packed [1 tile out][KT]: features 0..P-1 = W.a_src, P..2P-1 = W.a_dst
    int32_t out;       // chain-packed [NT][1]
    int32_t fold_out;  // chain-packed [1][1]: feature 0 = W_out.a_src, 1 = W_out.a_dst
    int32_t bn;        // raw [4][NT*32]: gamma, beta, mean, sqrt(var+eps)  (zero padded per row)
    int32_t nout;
};

struct V2Gru {
    V2Lin r, z;        // w[0..] input segments then hidden segment
    V2Lin ni, nh;
};

struct V2Weights {
    int32_t phi_x0_w, phi_x0_b;      // raw [H][2], [H]
    V2Lin phi_x2, phi_z0, phi_z2;
    V2Lin prior0, prior2, prior_ml;
    V2Lin dec0, dec2;
    int32_t dec_mean_w, dec_mean_b;  // raw [2][H], [2]
    V2Gru gru[2];
    V2Gat g_hid;
    V2Lin blend;
    V2Lin dec_goal0, dec_goal2;
    V2Gat g_goal;
    V2Lin lg_goals;
    // teacher-forced pass only (forward_v2.hip)
    V2Lin enc0, enc2, enc_ml;
    int32_t dec_logvar_w, dec_logvar_b;   // raw [2][H], [2]
};

struct V2Run {
    const float *wp;                 // chain-packed weights
    const int32_t *scene_ptr, *tile_u0, *tile_nu;
    int32_t S, N, K, T, n_tiles;
    int32_t G, D, adj_type;
    float alpha;
    double sigma;
    const float *h0, *x0;
    const int32_t *g0;
    const float *eps, *q;
    const int32_t *force_goals;
    uint64_t seed;
    float *out;
    int32_t *out_goals;
    // model 3 (goal head outside the kernel): steps [t_begin, t_end) of T, state per (sample, row), written back
    int32_t t_begin, t_end;
    float *h_out, *x_out;
    // teacher-forced pass (forward_v2.hip): inputs [T,N,.], adjacency per step (dense ld = N or compact ld = 0), per-tile
    // loss partials [n_tiles][3] (KLD, NLL, CE row sums)
    const float *traj, *traj_rel;
    const int32_t *goals;
    const float *adj;
    int64_t adj_ld, adj_tstride;
    const int64_t *adj_off;
    double *loss_part;
    long long *prof;                 // DN_PROFILE builds: per-phase cycle counters of tile 0
    float *dbg;                      // debug builds: tile dumps of the first step of tile 0 (else NULL)
};
