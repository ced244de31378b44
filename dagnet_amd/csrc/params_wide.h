// Wide goal head (g_dim in the hundreds, SDD-shaped DAG-Net): the goal head of DAGNet.sample
// (models/dagnet/model.py:215-235) runs layer by layer over all K*N rows of the batch, as batched fp32 MFMA GEMMs
// (gemm_f32.hip) plus small per-scene attention kernels (wide_head.hip); the rest of the step runs in the
// register-resident kernel (rollout_v2.hip, model 3).
#pragma once
#include <stdint.h>

struct WideWeights {            // float offsets into the model's weight buffer; all matrices row-major [out][ld]
    int32_t g0_w, g0_ld;        // dec_goal.0 columns [d | h]          [H][pad4(D+R)]
    int32_t g0_b;               // [H]
    int32_t g0_gather, g0_gld;  // dec_goal.0 one-hot columns, by goal  [G][g0_gld]
    int32_t g2_w, g2_ld, g2_b;  // dec_goal.2                           [G][pad4(H)], [G]
    int32_t hf_w, hf_ld;        // graph_goals heads + folded logits    [P*F + 2P][pad4(G)]
    int32_t out_w, out_ld;      // graph_goals.out_att.W^T, rows scaled by the folded BatchNorm scale   [G][pad4(P*F)]
    int32_t fold_out;           // [2][P*F]: W_out . a_src, W_out . a_dst
    int32_t bn;                 // [G]: folded eval BatchNorm shift (the bias of the out GEMM)
    int32_t lg_w, lg_ld, lg_b;  // lg_goals.0                           [G][pad4(2G)], [G]
};

struct WideBuffers {            // workspace of one plan (rows = K*N)
    float *A0;                  // [rows][a0_ld]   [d | h_last]
    float *T1;                  // [rows][H]
    float *A1;                  // [rows][2*Gp]    [softmax(dec_goal) | graph_goals]   (Gp = pad4(G))
    float *B;                   // [rows][Gp]      the lg_goals logits
    float *HS;                  // [rows][hs_ld]   heads (P*F) + folded logits (2P); then X3 = out-layer attention over X2
    float *X2;                  // [rows][P*F]
    float *S2;                  // [rows][2]
    float *adj;                 // [K][compact]    block-diagonal adjacency of every sample
    float *h;                   // [L][rows][R]    hidden state per (sample, row)
    float *pos;                 // [rows][2]
    int32_t *goal;              // [rows]
    int64_t a0_ld, hs_ld, Gp;
};
