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
    int32_t gather;    // one-hot table [G][NT*32] with the bias folded into every row, -1 = none
};

// GCN models reuse the slots (api.hip v2_gcn, v2_common.h gcn<>): heads = W1, fold = [2][32] layer-1 scale | shift,
// out = W2, bn = [2][NT*32] layer-2 scale | shift.
struct V2Gat {
    int32_t heads;     // chain-packed [1 tile out (P*F = 32)][KT]       (S->T product)
    int32_t fold;      // [64]: a_src | a_dst of the 4 heads, entry 8p+f = a_p[f] (resp. a_p[F+f]), times log2(e)
    int32_t out;       // chain-packed [NT][1]
    int32_t fold_out;  // [64]: W_out.a_src | W_out.a_dst over the 32 head features, times log2(e)
    int32_t bn;        // [2][NT*32]: GAT row 0 = folded BatchNorm shift (the scale sits in `out`), row 1 unused
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

// ---------------------------------------------------------------------------------------------
// Compile-time layout.  For one kernel instantiation every segment size is known (goal-wide dims are padded to
// V2_GT tiles, the disposition to its tile count), so the offsets of V2Weights RELATIVE to the start of the v2
// region are constants: the rollout kernel uses them as immediates (no table loads on the critical path, no
// SGPRs held for offsets).  v2_layout() mirrors the order of api.hip v2_pack(); dagnet_model_create() compares the
// packer's table with it and refuses the model on any difference.
// model codes as in rollout_v2.hip: 0 VRNN, 1 A-VRNN, 2 DAG-Net, 3 DAG-Net with the goal head outside, 4 DAG-Net with
// two disposition tiles.  gm: 1 GAT, 2 GCN.
// ---------------------------------------------------------------------------------------------
#define V2_GT 3     // goal tiles (g_dim <= 96)

// Split-bf16 packing (bf = 1; v2_common.h mm<..., BF = true>, api.hip Packer::chain3): every fp32 weight is three bf16 terms
// (8 + 8 + 8 significant bits by truncation: hi + mid + lo IS the weight), and a 32 x 32 block is
//     two half blocks (k-half p = 0, 1) of [term 0..2][lane 64] 8 x bf16 = W_term[32*nt + n][32*kt + 8*(2p + s/4) + 4*h + s%4], s = 0..7
// - the operand of v_mfma_f32_32x32x16_bf16 whose 8 k-slots of lane half h are registers 8p .. 8p+7 of an input tile -
// i.e. 2 x 768 dwords where the fp32 block has 1024.  Dense layers are streamed [kt][p][nt] (half an input tile is split once
// and feeds all output tiles); GRU gates and the graph layers' per-tile products are one output tile per call: [nt][kt][p].
#define V2_BLOCK_F32 1024
#define V2_BLOCK_BF3 1536
#define V2_HALF_BF3 768

struct V2Layout {
    V2Weights off;      // relative float offsets
    V2Weights used;     // 1 where `off` holds an offset this instantiation packs
    int32_t total;      // floats in the v2 region
};

constexpr V2Layout v2_layout(int model, int nl, int zd, int gm, int bf = 0) {
    V2Layout L{};
    int32_t cur = 0;
    const int BS = bf ? V2_BLOCK_BF3 : V2_BLOCK_F32;
    const bool dag = model >= 2, graph = model != 0, head = model == 2 || model == 4;
    const int DT = (model == 3 || model == 4) ? 2 : 1, GT = V2_GT, GP = V2_GT * 32;
    const int zt2 = (2 * zd + 31) / 32;
    auto take = [&cur](int32_t &o, int32_t &u, int n) {
        cur = (cur + 3) & ~3;
        o = cur;
        u = 1;
        cur += n;
    };
    // dense layer: chain segments [nt][k_i] (k_i = 0: absent), bias [nt*32], optional one-hot table [GP][nt*32]
    auto lin = [&](V2Lin &o, V2Lin &u, int nt, int k0, int k1, int k2, bool gather) {
        o.bias = -1;
        o.gather = -1;
        if (k0) take(o.w[0], u.w[0], nt * k0 * BS);
        if (k1) take(o.w[1], u.w[1], nt * k1 * BS);
        if (k2) take(o.w[2], u.w[2], nt * k2 * BS);
        take(o.bias, u.bias, nt * 32);
        if (gather) take(o.gather, u.gather, GP * nt * 32);
    };
    auto raw = [&](int32_t &w, int32_t &uw, int32_t &b, int32_t &ub, int nw, int nb) {
        take(w, uw, nw);
        take(b, ub, nb);
    };
    auto graphnet = [&](V2Gat &o, V2Gat &u, int kt, int nt) {
        if (gm == 2) {      // GCN: heads = W1, out = W2, fold / bn = folded scale | shift of the two layers
            take(o.heads, u.heads, kt * BS);
            take(o.out, u.out, nt * BS);
            take(o.fold, u.fold, 64);
            take(o.bn, u.bn, 2 * nt * 32);
        } else {
            take(o.heads, u.heads, kt * BS);
            take(o.fold, u.fold, 64);
            take(o.out, u.out, nt * BS);
            take(o.fold_out, u.fold_out, 64);
            take(o.bn, u.bn, 2 * nt * 32);
        }
    };
    raw(L.off.phi_x0_w, L.used.phi_x0_w, L.off.phi_x0_b, L.used.phi_x0_b, 128, 64);
    raw(L.off.dec_mean_w, L.used.dec_mean_w, L.off.dec_mean_b, L.used.dec_mean_b, 128, 2);
    lin(L.off.phi_x2, L.used.phi_x2, 2, 2, 0, 0, false);
    lin(L.off.phi_z0, L.used.phi_z0, 2, (zd + 31) / 32, 0, 0, false);
    lin(L.off.phi_z2, L.used.phi_z2, 2, 2, 0, 0, false);
    if (dag) {
        lin(L.off.prior0, L.used.prior0, 2, 2, 0, 0, true);
        lin(L.off.dec0, L.used.dec0, 2, DT, 2, 2, true);
    } else {
        lin(L.off.prior0, L.used.prior0, 2, 2, 0, 0, false);
        lin(L.off.dec0, L.used.dec0, 2, 2, 2, 0, false);
    }
    lin(L.off.prior2, L.used.prior2, 2, 2, 0, 0, false);
    lin(L.off.prior_ml, L.used.prior_ml, zt2, 2, 0, 0, false);
    lin(L.off.enc0, L.used.enc0, 2, 2, 2, 0, dag);
    lin(L.off.enc2, L.used.enc2, 2, 2, 0, 0, false);
    lin(L.off.enc_ml, L.used.enc_ml, zt2, 2, 0, 0, false);
    raw(L.off.dec_logvar_w, L.used.dec_logvar_w, L.off.dec_logvar_b, L.used.dec_logvar_b, 128, 2);
    lin(L.off.dec2, L.used.dec2, 2, 2, 0, 0, false);
    for (int l = 0; l < nl; ++l) {
        const int kx = l == 0 ? 4 : 2;
        lin(L.off.gru[l].r, L.used.gru[l].r, 2, kx, 2, 0, false);
        lin(L.off.gru[l].z, L.used.gru[l].z, 2, kx, 2, 0, false);
        lin(L.off.gru[l].ni, L.used.gru[l].ni, 2, kx, 0, 0, false);
        lin(L.off.gru[l].nh, L.used.gru[l].nh, 2, 2, 0, 0, false);
    }
    if (graph) {
        graphnet(L.off.g_hid, L.used.g_hid, 2, 2);
        lin(L.off.blend, L.used.blend, 2, 2, 2, 0, false);
    }
    if (head) {
        lin(L.off.dec_goal0, L.used.dec_goal0, 2, DT, 2, 0, true);
        lin(L.off.dec_goal2, L.used.dec_goal2, GT, 2, 0, 0, false);
        graphnet(L.off.g_goal, L.used.g_goal, GT, GT);
        lin(L.off.lg_goals, L.used.lg_goals, GT, GT, GT, 0, false);
    }
    L.total = (cur + 3) & ~3;
    return L;
}

struct V2Run {
    const float *wp;                 // chain-packed weights: start of the v2 region (compile-time relative offsets, v2_layout())
                                     // or, for model 3, of the whole buffer (absolute offsets from the packer's table)
    const int32_t *scene_ptr, *tile_u0, *tile_nu;
    int32_t S, N, K, T, n_tiles;
    int32_t G, D, adj_type;
    int32_t top_k;                   // kNN adjacency (adj_type 2): neighbours kept per row
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
