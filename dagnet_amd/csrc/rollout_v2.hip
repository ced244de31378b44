// Register-resident rollout kernel for the standard DAG-Net / A-VRNN / VRNN shapes
// (x_dim 2, h_dim = rnn_dim = 64, z_dim 16|32, n_layers 1|2, GAT with n_heads*graph_hid = 32,
// g_dim <= 96, scenes of <= 32 agents, adjacency type 0|1).  Anything else runs on the generic
// LDS-tiled kernel in rollout.hip.
//
// One WAVE owns a tile of <= 32 rows made of whole scene-samples and keeps their entire state
// in registers for all T steps; there is no workgroup barrier and no activation traffic at all:
//   * an activation tile is 32 features x 32 rows in the accumulator layout of
//     v_mfma_f32_32x32x2_f32 ("S": lane = row, register r = feature (r&3)+8(r>>2)+4(lane>>5)),
//     which is exactly the B operand the next layer's MFMA needs, so dense layers chain
//     accumulator -> operand with weights streamed from L2 in a matching packed order;
//   * swapping the operands of the same MFMA gives the product transposed ("T": lane = feature,
//     register = row), which is the A operand of the per-scene attention aggregation
//     out^T = Wh^T . alpha^T, so graph attention runs on the matrix cores too and its softmax
//     weights are built per lane (each lane owns one row of alpha);
//   * softmax / categorical draw / GRU gates / reparameterisation are per-lane VALU epilogues.
// Per-wave LDS (a few hundred bytes) is only used to broadcast positions and attention logits.
//
// Reference behaviour: models/dagnet/model.py:196-270, models/avrnn/model.py:163-215,
// models/baseline/model.py:135-170, models/dagnet/gat/gat_layers.py:25-40, gat_model.py:18-22,
// models/utils/adjacency_matrix.py:21-28,49-58, models/utils/utils.py:200-207.
#include "v2_common.h"

namespace v2 {

// MODEL: 0 VRNN, 1 A-VRNN, 2 DAG-Net, 4 DAG-Net with 17..32 agents per scene (two disposition tiles), 3 DAG-Net whose goal head runs outside this kernel (wide_head.hip: the goal of the
// step is read from the per-sample goal state; state is per (sample, row) and written back after the last step).
// NL: GRU layers.  ZT: 32 -> z_dim 32, 16 -> z_dim 16.
// GM: 1 = GAT (4 heads x 8), 2 = GCN (graph_hid <= 32).
// BF: dense and graph products on the 16-bit matrix pipe (two fp16 terms per fp32 value, three products: v2_common.h
// mm<..., BF>); weights from the split region of the model (params_v2.h).
#ifndef DN_V2_OCC
#define DN_V2_OCC 2       // waves per SIMD the register budget is cut for
#endif
template <int MODEL, int NL, int ZD, int GM, bool BF>
__global__ __launch_bounds__(64, DN_V2_OCC) void rollout_v2_kernel(const V2Weights *__restrict__ Wp, const V2Run R) {
    // segment offsets are compile-time constants of the instantiation (params_v2.h v2_layout; checked against the packer's
    // table when the model is created): immediates, no table loads on the critical path, no SGPRs held for them
    // (model 3 keeps the packer's table: its one-hot tables have g_dim rows, in the hundreds)
    static constexpr V2Layout LO = v2_layout(MODEL, NL, ZD, GM, BF);
    const V2Weights &W = MODEL == 3 ? *Wp : LO.off;
    __shared__ float xch[32 * 4 + 64];
    __shared__ float gfold[GM == 1 ? 2 * GAT_FOLD_FLOATS : 1];      // attention vectors of the hidden / goal GAT (v2_common.h gat_fold_to_lds)
    // DAG-Net: hidden state parked in LDS while the goal head needs the registers ([tile][reg][lane], conflict free)
    // also (NL == 2): the top layer's state is parked while GRU layer 0 runs (its 4 input tiles + 2 state tiles + gate
    // temporaries do not fit beside it; the compiler would spill ~140 registers to scratch memory instead)
    // split-product kernels: every layer's state has its slot (a GRU layer parks its state there while it multiplies its inputs)
    __shared__ float stash[(MODEL == 2 || MODEL == 4 || BF) ? NL * 32 * 64 : (NL == 2 ? 32 * 64 : 1)];
    __shared__ float gbias[NL * GRU_BIAS_FLOATS];      // GRU gate biases (v2_common.h gru_bias_to_lds)
    float *pos = xch + 128;                 // [32][2] current absolute positions of the tile rows
    const int lane = threadIdx.x, lh = lane >> 5, lc = lane & 31;
#ifdef DN_V2_STAGGER_SLOT
    {   // same idea, deterministic: the wave in an odd hardware slot of its SIMD (HW_ID bits 3:0) starts half a step late;
        // only the first round of waves needs it, a successor starts when its predecessor ends and inherits the offset
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);
        if ((slot & 1u) && blockIdx.x < 2048u)
            for (unsigned i = 0; i < DN_V2_STAGGER_SLOT; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
#ifdef DN_V2_STAGGER
    {   // de-synchronise the two waves of a SIMD: identical programs started together stay in lockstep and then want the
        // MFMA pipe, and later the VALU, at the same time; a pseudo-random start delay of up to ~one step breaks that
        const unsigned d = (blockIdx.x * 2654435761u >> 20) % DN_V2_STAGGER;
        for (unsigned i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    WBuf wp;
    wp.init(R.wp);
    constexpr bool DAG = MODEL >= 2, GRAPH = MODEL != 0, HEAD = MODEL == 2 || MODEL == 4, EXT = MODEL == 3;
    constexpr int GT = V2_GT;               // goal tiles (g_dim <= 96)
    constexpr int DT = (EXT || MODEL == 4) ? 2 : 1;         // disposition tiles (2 * n_max_agents <= 32 * DT)
    const int t_begin = EXT ? R.t_begin : 0, t_end = EXT ? R.t_end : R.T;

    // ---- which rows does this wave own?  (uniform scalar walk over the tile's scene-samples)
    int s0 = lc, n = 1, grow = -1, krow = 0;      // padding lanes: a one-agent scene of their own, never written
    {
        const int u0 = R.tile_u0[blockIdx.x], nu = R.tile_nu[blockIdx.x];
        int row = 0;
        for (int i = 0; i < nu; ++i) {
            const int u = u0 + i;
            const int k = u / R.S, s = u - k * R.S;
            const int b = R.scene_ptr[s], cnt = R.scene_ptr[s + 1] - b;
            if (lc >= row && lc < row + cnt) { s0 = row; n = cnt; grow = b + (lc - row); krow = k; }
            row += cnt;
        }
    }
    const bool live = grow >= 0;
    const int64_t NK = R.N;
    const uint32_t rngrow = (uint32_t)(krow * R.N + (live ? grow : 0));
    // state row: the K samples share h0/x0/g0 ([.., N, ..]) unless the state is per sample ([.., K*N, ..])
    const int64_t HN = EXT ? (int64_t)R.N * R.K : (int64_t)R.N;
    const int64_t srow = EXT ? (int64_t)krow * R.N + grow : (int64_t)grow;

    // ---- state
    T16 h[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = *reinterpret_cast<const float4 *>(R.h0 + ((int64_t)l * HN + srow) * 64 + nt * 32 + 8 * a + 4 * lh);
                h[l][nt][4 * a + 0] = v.x; h[l][nt][4 * a + 1] = v.y; h[l][nt][4 * a + 2] = v.z; h[l][nt][4 * a + 3] = v.w;
            }
    float xa = 0.f, ya = 0.f;
    int gi = 0;
    if (GRAPH && live) { xa = R.x0[srow * 2]; ya = R.x0[srow * 2 + 1]; }
    if (DAG && live) gi = min(max(R.g0[srow], 0), R.G - 1);
    unsigned amask = 0;

#pragma unroll
    for (int l = 0; l < NL; ++l) gru_bias_to_lds(gbias + l * GRU_BIAS_FLOATS, wp, W.gru[l], lane);
    if constexpr (GRAPH && GM == 1) gat_fold_to_lds(gfold, wp, W.g_hid, lane);
    if constexpr (HEAD && GM == 1) gat_fold_to_lds(gfold + GAT_FOLD_FLOATS, wp, W.g_goal, lane);
    const int first_off = HEAD ? W.dec_goal0.w[0] : W.prior0.w[0];
    WBlk pre = wload<BF>(wp, first_off, lane);
#ifdef DN_PROFILE
    long long pstamp = clock64();
#define PH(i)                                                                     \
    do {                                                                          \
        if (R.prof && blockIdx.x == 0 && lane == 0) {                             \
            const long long now_ = clock64();                                     \
            R.prof[i] += now_ - pstamp;                                           \
            pstamp = now_;                                                        \
        }                                                                         \
    } while (0)
#else
#define PH(i)
#endif
#ifdef DN_DEBUG_V2
#define DBG(stage, arr, NTILES)                                                                   \
    if (R.dbg && t == 0 && blockIdx.x == 0) {                                                     \
        for (int nt_ = 0; nt_ < NTILES; ++nt_)                                                    \
            for (int r_ = 0; r_ < 16; ++r_) R.dbg[(((stage)*4 + nt_) * 64 + lane) * 16 + r_] = arr[nt_][r_]; \
    }
#else
#define DBG(stage, arr, NTILES)
#endif
    const int lane0 = lane, s00 = s0, n0 = n;
    for (int t = t_begin; t < t_end; ++t) {
        // Lane predicates (v_cmp results) live in SGPR pairs; if the compiler hoists the ~150 loop-invariant ones
        // out of the time loop they spill.  Re-deriving the lane ids from an opaque copy once per step keeps
        // them local to the step.
        int lane = lane0, s0 = s00, n = n0;
        asm volatile("" : "+v"(lane), "+v"(s0), "+v"(n));
        const int lh = lane >> 5, lc = lane & 31;
        // Same idea for what depends on the row and the step: left alone, the compiler turns every [k][t][row] address into
        // a 64-bit per-lane induction pointer and the first Philox round of every counter into a loop invariant, and then
        // spills them to scratch (reloaded with a full memory round trip in the middle of the goal draw).  Recomputing
        // from opaque copies costs a few VALU instructions per step.
        int tq = t;
        asm volatile("" : "+s"(tq));
        uint32_t rrow = rngrow;
        int krow_t = krow, grow_t = grow;
        asm volatile("" : "+v"(rrow), "+v"(krow_t), "+v"(grow_t));
        const int64_t row_t = ((int64_t)krow_t * R.T + tq) * NK + (live ? grow_t : 0);      // row of [K][T][N] arrays, this step
        auto build_mask = [&]() {
            // rows j = FR(r)+4*lh of the tile that are in this row's scene and have adjacency > 0
            __syncthreads();
            if (lh == 0) { pos[lc * 2] = xa; pos[lc * 2 + 1] = ya; }
            __syncthreads();
            unsigned mk = 0;
            if (R.adj_type == 2) {
                // kNN (adjacency_matrix.py:86-103): column j is on iff it is among the min(top_k, n) nearest of this row, the row
                // itself included; ranked by the squared distance in fp64, ties by index - the order of the other kernels
                // (aux_kernels.hip dn_adjacency_kernel, rollout_body.h).  The two lanes of a row hold 16 candidates each: pass p
                // finds the p-th smallest key (d2, j) of the 32 - each half the smallest of its own that is greater than the
                // previous one, then the smaller of the two halves' - and the last one found is the threshold.
                // (the squared distances are recomputed from the positions in LDS in every pass: holding 16 doubles across the
                // passes costs 32 registers, which the kernel does not have - the compiler then spills around this branch on
                // the paths that never take it)
                auto d2 = [&](int r) -> double {
                    const int j = FR(r) + 4 * lh;
                    const float2 pj = *reinterpret_cast<const float2 *>(pos + j * 2);
                    const double dx = (double)xa - (double)pj.x, dy = (double)ya - (double)pj.y;
                    // spelled out (product rounded, then one fma): every pass must get the SAME bits for a candidate, or the
                    // "greater than the previous key" test stops excluding it - left to the compiler, one inlined copy was
                    // contracted differently from another
                    return ((j >= s0) & (j < s0 + n)) ? __builtin_fma(dx, dx, __dmul_rn(dy, dy)) : (double)INFINITY;
                };
                const int kk = min(R.top_k, n);
                double pd = -1.0;
                int pj = -1;
                for (int p = 0; p < R.top_k && p < 32; ++p) {
                    double cd = (double)INFINITY;
                    int cj = 64;
    #pragma unroll 4
                    for (int r = 0; r < 16; ++r) {
                        const int j = FR(r) + 4 * lh;
                        const double dr = d2(r);
                        const bool gt = (dr > pd) | ((dr == pd) & (j > pj));
                        const bool lt = (dr < cd) | ((dr == cd) & (j < cj));
                        if (gt & lt) { cd = dr; cj = j; }
                    }
                    const double od = __shfl_xor(cd, 32);
                    const int oj = __shfl_xor(cj, 32);
                    if ((od < cd) | ((od == cd) & (oj < cj))) { cd = od; cj = oj; }
                    if (p < kk) { pd = cd; pj = cj; }       // a scene smaller than top_k has all its columns on after n passes
                }
    #pragma unroll 4
                for (int r = 0; r < 16; ++r) {
                    const int j = FR(r) + 4 * lh;
                    const double dr = d2(r);
                    const bool on = (j >= s0) & (j < s0 + n) & ((dr < pd) | ((dr == pd) & (j <= pj)));
                    mk |= on ? (1u << r) : 0u;
                }
                amask = mk;
                return;
            }
            const double lim = 103.97207708399179 * R.sigma, lim2 = lim * lim;      // 150 ln 2 = 103.972077...
            const bool nodist = R.adj_type != 1;
            // adjacency > 0  <=>  float32(exp(-d/sigma)) > 0  <=>  exp(-d/sigma) > 2^-150 (the largest value that still rounds to
            // zero)  <=>  d < 150 ln2 sigma: a squared-distance compare, decided in fp64 as the reference's distances are.  The
            // squared distance in fp32 is within 1e-6 relative of the fp64 one, so unless some pair of the tile lands in the band
            // lim^2 (1 +- 4e-6) - the scenes would have to be ~125 sigma wide - the fp32 compare IS the fp64 compare, at a quarter of
            // the cost (16 pairs x 7 fp64 instructions per lane and step); the wave falls back to fp64 as a whole if any lane is in doubt.
            const float lim2f = (float)lim2, lo2 = lim2f * (1.0f - 4e-6f), hi2 = lim2f * (1.0f + 4e-6f);
            unsigned mk32 = 0;
            bool doubt = false;
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = FR(r) + 4 * lh;
                const float2 pj = *reinterpret_cast<const float2 *>(pos + j * 2);
                const float dx = xa - pj.x, dy = ya - pj.y, d2 = dx * dx + dy * dy;
                const bool in = (j >= s0) & (j < s0 + n);
                doubt |= in & (d2 > lo2) & (d2 < hi2);
                mk32 |= (in & (nodist | (d2 < lim2f))) ? (1u << r) : 0u;
            }
#ifdef DN_V2_MASK64
            doubt = true;
#endif
            if (nodist || !__any(doubt)) {
                amask = mk32;
                return;
            }
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = FR(r) + 4 * lh;
                const float2 pj = *reinterpret_cast<const float2 *>(pos + j * 2);
                const double dx = (double)xa - (double)pj.x, dy = (double)ya - (double)pj.y;
                const bool on = (j >= s0) & (j < s0 + n) & (dx * dx + dy * dy < lim2);
                mk |= on ? (1u << r) : 0u;
            }
            amask = mk;
        };
        // GCN: this lane's row of the adjacency, from the positions build_mask() left in LDS
        // (adjacency_matrix.py:21-28,49-58: 1 inside the scene, or exp(-d / sigma) with d in fp64; exp itself in fp32 here)
        auto adjval = [&](int r) -> float {
            const int j = FR(r) + 4 * lh;
            const bool in = (j >= s0) & (j < s0 + n);
            float v = 1.0f;
            if (R.adj_type == 2) return (amask >> r) & 1u ? 1.0f : 0.f;        // kNN: build_mask() holds the row's 0 / 1 entries
            if (R.adj_type == 1) {
                const float2 pj = *reinterpret_cast<const float2 *>(pos + j * 2);
                const double dx = (double)xa - (double)pj.x, dy = (double)ya - (double)pj.y;
                v = fexp(-(float)sqrt(dx * dx + dy * dy) * (float)(1.0 / R.sigma));
            }
            return in ? v : 0.f;
        };
        T16 d[DT];
        auto build_disposition = [&]() {    // flattened xy of the scene's agents, zero padded
            int s0 = s00, ln = lane0;       // opaque per call: the two calls of a step must not share (and spill) their LDS addresses
            asm volatile("" : "+v"(s0), "+v"(ln));
            const int lh = ln >> 5;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = dt * 32 + FR(r) + 4 * lh, j = f >> 1;
                    const float pv = pos[min(s0 + j, 31) * 2 + (f & 1)];
                    d[dt][r] = ((f < R.D) & (j < n)) ? pv : 0.f;
                }
        };
        if constexpr (EXT) build_mask();    // PRE-update positions; the goal of this step is already in the goal state
        if constexpr (HEAD) {
            if constexpr (NL == 2) {        // h[0] is not read until the GRU: park it in LDS, the goal head needs the registers
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stash[((2 + nt) * 16 + r) * 64 + lane] = h[0][nt][r];
            }
            build_mask();                   // PRE-update positions (models/dagnet/model.py:215-220)
            build_disposition();
            PH(0);
            // ---------------------------------------------------------------- goal head
            T16 tg[2];
            start<2>(tg, wp, W.dec_goal0, gi, lh);
            pre = mm<2, DT, false, BF>(tg, wp, W.dec_goal0.w[0], d, lane, pre, W.dec_goal0.w[1]);
            pre = mm<2, 2, false, BF>(tg, wp, W.dec_goal0.w[1], h[NL - 1], lane, pre, W.dec_goal2.w[0]);
            activate<2, 2>(tg);
            // the top-layer hidden state is not read again until the prior: park it as well
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) stash[(nt * 16 + r) * 64 + lane] = h[NL - 1][nt][r];
            T16 decg[GT];
            start<GT>(decg, wp, W.dec_goal2, 0, lh);
            pre = mm<GT, 2, false, BF>(decg, wp, W.dec_goal2.w[0], tg, lane, pre, W.lg_goals.w[0]);
            activate<0, GT>(decg);
            softmax_feat<GT>(decg);
            PH(1);
            // lg_goals = W1.dec_g + W2.g_graph: the dec_g half runs before the graph so that dec_g is dead
            // once the graph has consumed it (peak registers: gc + gg instead of dec_g + gg + gc)
            T16 gc[GT];
            start<GT>(gc, wp, W.lg_goals, 0, lh);
            pre = mm<GT, GT, false, BF>(gc, wp, W.lg_goals.w[0], decg, lane, pre, W.g_goal.heads);
            T16 gg[GT];
            if constexpr (GM == 2) pre = gcn<GT, GT, BF>(decg, gg, wp, W.g_goal, adjval, lane, pre, W.lg_goals.w[1]);
            else pre = gat<GT, GT, BF>(decg, gg, wp, W.g_goal, amask, R.alpha, xch, gfold + GAT_FOLD_FLOATS, lane, pre, W.lg_goals.w[1], [&](int i) { PH(i); });
            PH(2);
            pre = mm<GT, GT, false, BF>(gc, wp, W.lg_goals.w[1], gg, lane, pre, W.prior0.w[0]);
            activate<0, GT>(gc);
            // softmax left unnormalised: gc = exp2(logit - max), p = gc * inv
            const float pscale = softmax_feat_unnorm<GT>(gc) * 1.44269504088896340736f;
            // categorical draw: argmax_g exp(p_g) / q_g, q ~ Exp(1) (torch.multinomial on exp(probabilities), utils.py:200-207),
            // evaluated as argmax_g  p_g log2(e) - log2(q_g)  (a monotone map of the same ratio: one v_log instead of
            // v_exp + v_rcp per feature, no normalisation pass).  With the Philox stream q = -ln(u), u = n 2^-24, so
            // log2(q) = log2(24 - log2(n)) + log2(ln 2), and the constant drops out of the argmax.  Branch-free (selects
            // only): features are visited in increasing order within a lane, so a strict '>' keeps the first maximum;
            // out-of-range features score -inf.
            float best = -INFINITY;
            int bi = 0x7fffffff;
            const int Gm1 = R.G - 1;
            const int64_t qrow = row_t * R.G;
            if (R.q) {          // injected q (parity with the reference's torch.multinomial): the exponential race
#pragma unroll
                for (int nt = 0; nt < GT; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int f0 = nt * 32 + 8 * a + 4 * lh;
                        float lq[4];          // log2(q) up to a constant; loads are clamped into the row instead of predicated
#pragma unroll
                        for (int i = 0; i < 4; ++i) lq[i] = __builtin_amdgcn_logf(R.q[qrow + min(f0 + i, Gm1)]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int f = f0 + i;
                            float v = gc[nt][4 * a + i] * pscale - lq[i];
                            v = f <= Gm1 ? v : -INFINITY;
                            const bool take = v > best;
                            best = take ? v : best;
                            bi = take ? f : bi;
                        }
                    }
                const float ov = xhalf(best);
                const int oi = xhalf(bi);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            } else {
                // in-kernel stream: one uniform per row and step, inverse of the cumulative weights (rng.h rng_draw_u): one Philox
                // block, G exp2 and ~3 G additions / selects where the race took G / 4 blocks and 2 G logarithms - the draw
                // was 1 500 of the step's 8 700 VALU instructions.  Order of the sum: the groups of four consecutive features
                // a lane half holds, ((w0 + w1) + w2) + w3, then the 8 GT groups in feature order (both halves add the same
                // numbers in the same order, so they agree on the group; the half that holds it resolves the feature).
                constexpr int NG = 4 * GT;          // groups per half
                float gs[NG];
#pragma unroll
                for (int nt = 0; nt < GT; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int f0 = nt * 32 + 8 * a + 4 * lh;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float w = __builtin_amdgcn_exp2f(gc[nt][4 * a + i] * pscale);      // exp(p), p the softmax probability
                            gc[nt][4 * a + i] = f0 + i <= Gm1 ? w : 0.f;
                        }
                        gs[nt * 4 + a] = ((gc[nt][4 * a] + gc[nt][4 * a + 1]) + gc[nt][4 * a + 2]) + gc[nt][4 * a + 3];
                    }
                float pre[2 * NG];                  // inclusive prefix over the groups in feature order: k = 2 (4 nt + a) + half
                float run = 0.f;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float og = xhalf(gs[g]);
                    run += lh ? og : gs[g];
                    pre[2 * g] = run;
                    run += lh ? gs[g] : og;
                    pre[2 * g + 1] = run;
                }
                const float target = rng_draw_u(R.seed, rrow, (uint32_t)tq) * run;
                int ks = 0;                         // groups that end at or below the target: the draw lies in group ks
                float base = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * NG - 1; ++k) {
                    const bool below = pre[k] <= target;
                    ks += below ? 1 : 0;
                    base = below ? pre[k] : base;
                }
                const int mine = ks >> 1;           // the group's index in its half
                float w0 = 0.f, w1 = 0.f, w2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < GT; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const bool hit = mine == nt * 4 + a;
                        w0 = hit ? gc[nt][4 * a] : w0;
                        w1 = hit ? gc[nt][4 * a + 1] : w1;
                        w2 = hit ? gc[nt][4 * a + 2] : w2;
                    }
                const float c0 = base + w0, c1 = c0 + w1, c2 = c1 + w2;
                const int in = (c0 <= target ? 1 : 0) + (c1 <= target ? 1 : 0) + (c2 <= target ? 1 : 0);
                const int f = (mine >> 2) * 32 + (mine & 3) * 8 + (ks & 1) * 4 + in;
                const int of = xhalf(f);
                bi = (ks & 1) == lh ? f : of;
            }
            if (R.force_goals && live) bi = min(max(R.force_goals[row_t], 0), R.G - 1);
            gi = min(bi, R.G - 1);
            if (R.out_goals && live && lh == 0) R.out_goals[row_t] = gi;
        }
        PH(3);
        if constexpr (HEAD) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[NL - 1][nt][r] = stash[(nt * 16 + r) * 64 + lane];
        }
        // -------------------------------------------------------------------- prior
        T16 ta[2], tb[2];
        DBG(7, h[NL - 1], 2);
        start<2>(ta, wp, W.prior0, gi, lh);
        pre = mm<2, 2, false, BF>(ta, wp, W.prior0.w[0], h[NL - 1], lane, pre, W.prior2.w[0]);
        activate<1, 2>(ta);
        DBG(0, ta, 2);
        start<2>(tb, wp, W.prior2, 0, lh);
        pre = mm<2, 2, false, BF>(tb, wp, W.prior2.w[0], ta, lane, pre, W.prior_ml.w[0]);
        activate<1, 2>(tb);
        constexpr int MLT = ZD == 32 ? 2 : 1;      // [mean ; logvar] = 2*z_dim features
        T16 ml[MLT];
        start<MLT>(ml, wp, W.prior_ml, 0, lh);
        pre = mm<MLT, 2, false, BF>(ml, wp, W.prior_ml.w[0], tb, lane, pre, W.phi_z0.w[0]);
        activate<0, MLT>(ml);
        DBG(1, tb, 2);
        DBG(2, ml, MLT);
        PH(4);
        // -------------------------------------------------------------------- z = eps * exp(0.5 logvar) + mean
        T16 z[1];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int f0 = 8 * a + 4 * lh;        // latent index of registers 4a..4a+3
            float e4[4] = {0.f, 0.f, 0.f, 0.f};
            if (f0 < ZD) {
                if (R.eps) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        e4[i] = live ? R.eps[row_t * ZD + f0 + i] : 0.f;
                } else {
                    const uint4 u = philox4x32(make_uint4(rrow, (uint32_t)tq, (uint32_t)(f0 >> 2), 0u),
                                                  make_uint2((uint32_t)R.seed, (uint32_t)(R.seed >> 32)));
                    e4[0] = (float)(u.x >> 8) * (1.0f / 16777216.0f);
                    e4[1] = (float)(u.y >> 8) * (1.0f / 16777216.0f);
                    e4[2] = (float)(u.z >> 8) * (1.0f / 16777216.0f);
                    e4[3] = (float)(u.w >> 8) * (1.0f / 16777216.0f);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * a + i;
                float v = 0.f;
                if constexpr (ZD == 32) {
                    v = e4[i] * fexp(0.5f * ml[MLT - 1][r]) + ml[0][r];
                } else {
                    if (a < 2) v = e4[i] * fexp(0.5f * ml[0][(r + 8) & 15]) + ml[0][r];     // logvar = features 16..31
                }
                z[0][r] = v;
            }
        }
        DBG(3, z, 1);
        PH(5);
        // -------------------------------------------------------------------- phi_z, decoder
        start<2>(ta, wp, W.phi_z0, 0, lh);
        pre = mm<2, 1, false, BF>(ta, wp, W.phi_z0.w[0], z, lane, pre, W.phi_z2.w[0]);
        activate<1, 2>(ta);
        T16 pz[2];
        start<2>(pz, wp, W.phi_z2, 0, lh);
        pre = mm<2, 2, false, BF>(pz, wp, W.phi_z2.w[0], ta, lane, pre, W.dec0.w[0]);
        activate<1, 2>(pz);
        DBG(4, pz, 2);
        start<2>(ta, wp, W.dec0, gi, lh);
        if (DAG) {
            // rebuilt from the positions in LDS (unchanged since the start of the step) rather than kept live
            build_disposition();
            pre = mm<2, DT, false, BF>(ta, wp, W.dec0.w[0], d, lane, pre, W.dec0.w[1]);
            pre = mm<2, 2, false, BF>(ta, wp, W.dec0.w[1], pz, lane, pre, W.dec0.w[2]);
            pre = mm<2, 2, false, BF>(ta, wp, W.dec0.w[2], h[NL - 1], lane, pre, W.dec2.w[0]);
        } else {
            pre = mm<2, 2, false, BF>(ta, wp, W.dec0.w[0], pz, lane, pre, W.dec0.w[1]);
            pre = mm<2, 2, false, BF>(ta, wp, W.dec0.w[1], h[NL - 1], lane, pre, W.dec2.w[0]);
        }
        activate<1, 2>(ta);
        start<2>(tb, wp, W.dec2, 0, lh);
        pre = mm<2, 2, false, BF>(tb, wp, W.dec2.w[0], ta, lane, pre, W.phi_x2.w[0]);
        activate<1, 2>(tb);
        DBG(5, ta, 2);
        DBG(6, tb, 2);
        // dec_mean: 2 outputs over 64 features, each half sums its 32 features then the halves combine
        float dmx = 0.f, dmy = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int fs = nt * 32 + 8 * a;
                const float4 w0 = wp.ld4(lh * 16, W.dec_mean_w + fs);
                const float4 w1 = wp.ld4(lh * 16, W.dec_mean_w + 64 + fs);
                dmx += w0.x * tb[nt][4 * a] + w0.y * tb[nt][4 * a + 1] + w0.z * tb[nt][4 * a + 2] + w0.w * tb[nt][4 * a + 3];
                dmy += w1.x * tb[nt][4 * a] + w1.y * tb[nt][4 * a + 1] + w1.z * tb[nt][4 * a + 2] + w1.w * tb[nt][4 * a + 3];
            }
        dmx = hsum(dmx);
        dmy = hsum(dmy);
        dmx += wp.ld1(0, W.dec_mean_b);
        dmy += wp.ld1(0, W.dec_mean_b + 1);
        if (live && lh == 0)
            *reinterpret_cast<float2 *>(R.out + row_t * 2) = make_float2(dmx, dmy);
        PH(6);
        // -------------------------------------------------------------------- phi_x of the prediction
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int fs = nt * 32 + 8 * a;
                const float4 wa = wp.ld4(lh * 32, W.phi_x0_w + fs * 2);          // W[f0..f0+1][0..1]
                const float4 wb = wp.ld4(lh * 32, W.phi_x0_w + fs * 2 + 4);      // W[f0+2..f0+3][0..1]
                const float4 bb = wp.ld4(lh * 16, W.phi_x0_b + fs);
                ta[nt][4 * a + 0] = lrelu(wa.x * dmx + wa.y * dmy + bb.x, 0.01f);
                ta[nt][4 * a + 1] = lrelu(wa.z * dmx + wa.w * dmy + bb.y, 0.01f);
                ta[nt][4 * a + 2] = lrelu(wb.x * dmx + wb.y * dmy + bb.z, 0.01f);
                ta[nt][4 * a + 3] = lrelu(wb.z * dmx + wb.w * dmy + bb.w, 0.01f);
            }
        T16 xin[4];
        {
            T16 px[2];
            start<2>(px, wp, W.phi_x2, 0, lh);
            pre = mm<2, 2, false, BF>(px, wp, W.phi_x2.w[0], ta, lane, pre, BF ? W.gru[0].r.w[0] : W.gru[0].nh.w[0]);
            activate<1, 2>(px);
            xin[0] = px[0]; xin[1] = px[1]; xin[2] = pz[0]; xin[3] = pz[1];
        }
        PH(7);
        // -------------------------------------------------------------------- GRU
        if constexpr (HEAD && NL == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[0][nt][r] = stash[((2 + nt) * 16 + r) * 64 + lane];
        }
        const int after_gru = GRAPH ? W.g_hid.heads : first_off;
        if constexpr (NL == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) stash[(nt * 16 + r) * 64 + lane] = h[1][nt][r];
        }
        // (slot of h[l] in the stash: tiles 2l, 2l+1 counted from the top layer down - the top layer's state sits at 0)
        pre = gru<4, true, BF>(xin, h[0], wp, W.gru[0], lane, pre, NL == 2 ? (BF ? W.gru[1].r.w[0] : W.gru[1].nh.w[0]) : after_gru, gbias,
                               BF ? stash + (NL == 2 ? 2 * 16 * 64 : 0) : nullptr);
        if constexpr (NL == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[1][nt][r] = stash[(nt * 16 + r) * 64 + lane];
            pre = gru<2, true, BF>(h[0], h[1], wp, W.gru[1], lane, pre, after_gru, gbias + GRU_BIAS_FLOATS, BF ? stash : nullptr);
        }
        PH(8);
        // -------------------------------------------------------------------- graph refinement of h[-1]
        if (MODEL == 1) {       // A-VRNN: adjacency from the UPDATED positions (models/avrnn/model.py:201-209)
            xa += dmx;
            ya += dmy;
            build_mask();
        }
        if (GRAPH) {
            T16 hg[2];
            if constexpr (GM == 2) pre = gcn<2, 2, BF>(h[NL - 1], hg, wp, W.g_hid, adjval, lane, pre, W.blend.w[0]);
            else pre = gat<2, 2, BF>(h[NL - 1], hg, wp, W.g_hid, amask, R.alpha, xch, gfold, lane, pre, W.blend.w[0]);
            start<2>(ta, wp, W.blend, 0, lh);
            pre = mm<2, 2, false, BF>(ta, wp, W.blend.w[0], hg, lane, pre, W.blend.w[1]);
            pre = mm<2, 2, false, BF>(ta, wp, W.blend.w[1], h[NL - 1], lane, pre, first_off);
            activate<0, 2>(ta);
            h[NL - 1][0] = ta[0];
            h[NL - 1][1] = ta[1];
        }
        if (DAG) {
            xa += dmx;
            ya += dmy;
        }
        PH(9);
    }
    if constexpr (EXT) {    // write the state back (each wave touches only its own rows: in place is safe)
        if (live) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        *reinterpret_cast<float4 *>(R.h_out + ((int64_t)l * HN + srow) * 64 + nt * 32 + 8 * a + 4 * lh) =
                            make_float4(h[l][nt][4 * a], h[l][nt][4 * a + 1], h[l][nt][4 * a + 2], h[l][nt][4 * a + 3]);
            if (lh == 0) *reinterpret_cast<float2 *>(R.x_out + srow * 2) = make_float2(xa, ya);
        }
    }
}

}   // namespace v2

// This file is compiled twice (Makefile): DN_V2_BF = 0 -> the fp32-MFMA kernels (dn_launch_rollout_v2), DN_V2_BF = 1 -> the
// split-product ones (dn_launch_rollout_v2_bf3); one translation unit each so that the two sets build in parallel.
#ifndef DN_V2_BF
#define DN_V2_BF 0
#endif

template <int MODEL, int NL, int ZD, int GM>
static hipError_t launch_one(const V2Weights *d_w, const V2Run &r, hipStream_t st) {
    hipLaunchKernelGGL((v2::rollout_v2_kernel<MODEL, NL, ZD, GM, DN_V2_BF != 0>), dim3(r.n_tiles), dim3(64), 0, st, d_w, r);
    return hipGetLastError();
}

// (model, GRU layers, z_dim, graph) combinations with an instantiation: the shapes of the shipped checkpoints
// (basketball: DAG-Net L2/Z32, A-VRNN L1/Z32, VRNN L2/Z32; SDD: A-VRNN L1/Z16, VRNN L2/Z16) plus A-VRNN L2/Z32, DAG-Net L1/Z32 and L2/Z16, model 3
// (goal head outside) and the GCN variants of the graph models.  graph: 0 none (VRNN), 1 GAT, 2 GCN.
#ifdef DN_V2_ONLY      // kernel experiments (tools/variants.sh): the DAG-Net / A-VRNN / VRNN basketball instantiations only
#define V2_INSTANCES V2_CASE(2, 2, 32, 1) V2_CASE(1, 1, 32, 1) V2_CASE(0, 2, 32, 1) V2_CASE(2, 1, 32, 2) V2_CASE(2, 2, 32, 2)
#else
#define V2_INSTANCES                                                                                                   \
    V2_CASE(2, 2, 32, 1) V2_CASE(1, 1, 32, 1) V2_CASE(0, 2, 32, 1) V2_CASE(1, 1, 16, 1) V2_CASE(0, 2, 16, 1) V2_CASE(1, 2, 32, 1) \
    V2_CASE(3, 2, 32, 1) V2_CASE(4, 2, 32, 1) V2_CASE(2, 2, 32, 2) V2_CASE(1, 1, 32, 2) V2_CASE(1, 1, 16, 2) V2_CASE(1, 2, 32, 2) \
    V2_CASE(2, 1, 32, 1) V2_CASE(2, 2, 16, 1) V2_CASE(4, 1, 32, 1) V2_CASE(4, 2, 16, 1) V2_CASE(3, 1, 32, 1) V2_CASE(3, 2, 16, 1) \
    V2_CASE(2, 1, 32, 2) V2_CASE(2, 2, 16, 2)
#endif

// returns hipErrorInvalidValue when the combination has no instantiation; d_w: DEVICE pointer to the model's V2Weights table
#if DN_V2_BF
extern "C" hipError_t dn_launch_rollout_v2_bf3(int model, int n_layers, int z_dim, int graph, const V2Weights *d_w, const V2Run *r,
                                               hipStream_t st) {
#else
extern "C" hipError_t dn_launch_rollout_v2(int model, int n_layers, int z_dim, int graph, const V2Weights *d_w, const V2Run *r,
                                           hipStream_t st) {
#endif
    const int gm = graph == 2 ? 2 : 1;
#define V2_CASE(M, L, Z, G) \
    if (model == M && n_layers == L && z_dim == Z && gm == G) return launch_one<M, L, Z, G>(d_w, *r, st);
    V2_INSTANCES
#undef V2_CASE
    return hipErrorInvalidValue;
}

#if !DN_V2_BF
extern "C" int dn_v2_has(int model, int n_layers, int z_dim, int graph) {
    const int gm = graph == 2 ? 2 : 1;
#define V2_CASE(M, L, Z, G) \
    if (model == M && n_layers == L && z_dim == Z && gm == G) return 1;
    V2_INSTANCES
#undef V2_CASE
    return 0;
}
#endif
