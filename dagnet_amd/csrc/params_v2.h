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

// Split packing (bf = 1; v2_common.h mm<..., BF = true>, api.hip Packer::chain_split): the dense products of the rollout kernels
// run on the 16-bit matrix pipe with every fp32 value as TWO fp16 terms (11 + 11 significant bits, round to nearest: hi = fp16(v),
// lo = fp16(v - hi)) and three products (hi hi, hi lo, lo hi; the dropped lo lo is below 2^-22 of the product) accumulated in
// fp32 - measured error of an 8-layer chain against fp64: 2.4e-6 max / 1.4e-7 mean, the fp32 MFMA's 2.2e-6 / 1.1e-7
// (tools/ubench/split_bf16.hip, profiles/r03_split_ubench.txt; three bf16 terms / six products, the first form of this round:
// 1.1e-6, at 1.65 x the time).  fp16 keeps fp32's range only up to 65504: a layer INPUT beyond that overflows (activations of
// these models are O(1) .. O(100)); values below 6e-5 are fp16 subnormals, kept by the matrix pipe (absolute error 3e-8).
// A 32 x 32 weight block is two half blocks (k-half p = 0, 1) of
//     [term hi, lo][lane 64] 8 x fp16 = W_term[32*nt + n][32*kt + 8*(2p + s/4) + 4*h + s%4], s = 0..7
// - the operand of v_mfma_f32_32x32x16_f16 whose 8 k-slots of lane half h are registers 8p .. 8p+7 of an input tile - i.e. 2 x 512
// dwords, the size of the fp32 block.  Dense layers are streamed [kt][p][nt] (half an input tile is split once and feeds all
// output tiles); the graph layers' per-tile products are one output tile per call: [nt][kt][p].
#define V2_BLOCK_F32 1024
#define V2_BLOCK_SPLIT 1024
#define V2_HALF_SPLIT 512

struct V2Layout {
    V2Weights off;      // relative float offsets
    V2Weights used;     // 1 where `off` holds an offset this instantiation packs
    int32_t total;      // floats in the v2 region
};

// Order of the region: first every MFMA-packed weight segment in the order in which one time step of the rollout kernel
// consumes them (goal head, prior, phi_z, decoder, phi_x, GRU gates nh / r / ni / z per layer, hidden graph, blend) - so the
// kernel's weight stream is ONE ascending sequence of blocks per step and a prefetch of any depth simply runs ahead into the
// next segment (v2_common.h mm<>) - then the segments only the teacher-forced pass reads, then all vectors (biases, one-hot
// tables, attention vectors, the raw phi_x.0 / dec_mean / dec_logvar layers).
constexpr V2Layout v2_layout(int model, int nl, int zd, int gm, int bf = 0) {
    V2Layout L{};
    int32_t cur = 0;
    const int BS = bf ? V2_BLOCK_SPLIT : V2_BLOCK_F32;
    const bool dag = model >= 2, graph = model != 0, head = model == 2 || model == 4;
    const int DT = (model == 3 || model == 4) ? 2 : 1, GT = V2_GT, GP = V2_GT * 32;
    const int zt2 = (2 * zd + 31) / 32;
    V2Weights &o = L.off, &u = L.used;
    auto take = [&cur](int32_t &off, int32_t &used, int n) {
        cur = (cur + 3) & ~3;
        off = cur;
        used = 1;
        cur += n;
    };
    auto seg = [&](int32_t &off, int32_t &used, int nt, int kt) { take(off, used, nt * kt * BS); };
    // ---- the weight stream of one rollout step
    if (head) {
        seg(o.dec_goal0.w[0], u.dec_goal0.w[0], 2, DT);
        seg(o.dec_goal0.w[1], u.dec_goal0.w[1], 2, 2);
        seg(o.dec_goal2.w[0], u.dec_goal2.w[0], GT, 2);
        seg(o.lg_goals.w[0], u.lg_goals.w[0], GT, GT);
        seg(o.g_goal.heads, u.g_goal.heads, 1, GT);
        seg(o.g_goal.out, u.g_goal.out, GT, 1);
        seg(o.lg_goals.w[1], u.lg_goals.w[1], GT, GT);
    }
    seg(o.prior0.w[0], u.prior0.w[0], 2, 2);
    seg(o.prior2.w[0], u.prior2.w[0], 2, 2);
    seg(o.prior_ml.w[0], u.prior_ml.w[0], zt2, 2);
    seg(o.phi_z0.w[0], u.phi_z0.w[0], 2, (zd + 31) / 32);
    seg(o.phi_z2.w[0], u.phi_z2.w[0], 2, 2);
    if (dag) {
        seg(o.dec0.w[0], u.dec0.w[0], 2, DT);
        seg(o.dec0.w[1], u.dec0.w[1], 2, 2);
        seg(o.dec0.w[2], u.dec0.w[2], 2, 2);
    } else {
        seg(o.dec0.w[0], u.dec0.w[0], 2, 2);
        seg(o.dec0.w[1], u.dec0.w[1], 2, 2);
    }
    seg(o.dec2.w[0], u.dec2.w[0], 2, 2);
    seg(o.phi_x2.w[0], u.phi_x2.w[0], 2, 2);
    for (int l = 0; l < nl; ++l) {
        const int kx = l == 0 ? 4 : 2;
        if (bf) {       // two dense layers with six output tiles: r.w[0] = [n | r | z] rows of W_ih, r.w[1] = [r | z | n] rows of W_hh
            seg(o.gru[l].r.w[0], u.gru[l].r.w[0], 6, kx);
            seg(o.gru[l].r.w[1], u.gru[l].r.w[1], 6, 2);
        } else {
            seg(o.gru[l].nh.w[0], u.gru[l].nh.w[0], 2, 2);
            seg(o.gru[l].r.w[0], u.gru[l].r.w[0], 2, kx);
            seg(o.gru[l].r.w[1], u.gru[l].r.w[1], 2, 2);
            seg(o.gru[l].ni.w[0], u.gru[l].ni.w[0], 2, kx);
            seg(o.gru[l].z.w[0], u.gru[l].z.w[0], 2, kx);
            seg(o.gru[l].z.w[1], u.gru[l].z.w[1], 2, 2);
        }
    }
    if (graph) {
        seg(o.g_hid.heads, u.g_hid.heads, 1, 2);
        seg(o.g_hid.out, u.g_hid.out, 2, 1);
        seg(o.blend.w[0], u.blend.w[0], 2, 2);
        seg(o.blend.w[1], u.blend.w[1], 2, 2);
    }
    // ---- teacher-forced pass only (forward_v2.hip)
    seg(o.enc0.w[0], u.enc0.w[0], 2, 2);
    seg(o.enc0.w[1], u.enc0.w[1], 2, 2);
    seg(o.enc2.w[0], u.enc2.w[0], 2, 2);
    seg(o.enc_ml.w[0], u.enc_ml.w[0], zt2, 2);
    // ---- vectors: bias [nt*32] (+ one-hot table [GP][nt*32]) per dense layer, in the order above
    auto vec = [&](V2Lin &lo, V2Lin &lu, int nt, bool gather) {
        lo.gather = -1;
        take(lo.bias, lu.bias, nt * 32);
        if (gather) take(lo.gather, lu.gather, GP * nt * 32);
    };
    auto gvec = [&](V2Gat &go, V2Gat &gu, int nt) {
        take(go.fold, gu.fold, 64);
        if (gm != 2) take(go.fold_out, gu.fold_out, 64);
        take(go.bn, gu.bn, 2 * nt * 32);
    };
    if (head) {
        vec(o.dec_goal0, u.dec_goal0, 2, true);
        vec(o.dec_goal2, u.dec_goal2, GT, false);
        vec(o.lg_goals, u.lg_goals, GT, false);
        gvec(o.g_goal, u.g_goal, GT);
    }
    vec(o.prior0, u.prior0, 2, dag);
    vec(o.prior2, u.prior2, 2, false);
    vec(o.prior_ml, u.prior_ml, zt2, false);
    vec(o.phi_z0, u.phi_z0, 2, false);
    vec(o.phi_z2, u.phi_z2, 2, false);
    vec(o.dec0, u.dec0, 2, dag);
    vec(o.dec2, u.dec2, 2, false);
    vec(o.phi_x2, u.phi_x2, 2, false);
    for (int l = 0; l < nl; ++l) {
        vec(o.gru[l].nh, u.gru[l].nh, 2, false);
        vec(o.gru[l].r, u.gru[l].r, 2, false);
        vec(o.gru[l].ni, u.gru[l].ni, 2, false);
        vec(o.gru[l].z, u.gru[l].z, 2, false);
    }
    if (graph) {
        gvec(o.g_hid, u.g_hid, 2);
        vec(o.blend, u.blend, 2, false);
    }
    vec(o.enc0, u.enc0, 2, dag);
    vec(o.enc2, u.enc2, 2, false);
    vec(o.enc_ml, u.enc_ml, zt2, false);
    take(o.phi_x0_w, u.phi_x0_w, 128);
    take(o.phi_x0_b, u.phi_x0_b, 64);
    take(o.dec_mean_w, u.dec_mean_w, 128);
    take(o.dec_mean_b, u.dec_mean_b, 2);
    take(o.dec_logvar_w, u.dec_logvar_w, 128);
    take(o.dec_logvar_b, u.dec_logvar_b, 2);
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
