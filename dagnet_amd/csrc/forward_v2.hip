// Register-resident teacher-forced pass: VRNN.forward / AVRNN.forward / DAGNet.forward in eval mode
// (models/baseline/model.py:70-133, models/avrnn/model.py:93-161, models/dagnet/model.py:138-194) for the shapes
// rollout_v2.hip covers.  Same scheme (one wave owns <= 32 rows of whole scenes, state in registers for all steps,
// chained fp32 MFMA layers, graph attention on the matrix cores); differences to the sample kernel:
//   * inputs are teacher forced (x_t, positions, goals from the batch), the latent comes from the ENCODER,
//   * the adjacency mask of step t is read from the caller's adjacency (dense or compact block-diagonal),
//   * per-row KLD / NLL / CE are accumulated in fp64 and reduced per tile (dn_loss_finalize_kernel adds the tiles).
#include "v2_common.h"

namespace v2 {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODEL, int NL, int ZD, int GM>
__global__ __launch_bounds__(64, 2) void forward_v2_kernel(const V2Weights *__restrict__ Wp, const V2Run R) {
    // compile-time segment offsets relative to the v2 region, as in rollout_v2.hip (model 3 reads the packer's table)
    static constexpr V2Layout LO = v2_layout(MODEL, NL, ZD, GM);
    const V2Weights &W = MODEL == 3 ? *Wp : LO.off;
    __shared__ float xch[32 * 4 + 64];
    __shared__ float gfold[GM == 1 ? 2 * GAT_FOLD_FLOATS : 1];      // attention vectors of the hidden / goal GAT (v2_common.h gat_fold_to_lds)
    __shared__ float stash[(MODEL == 2 || MODEL == 4) ? NL * 32 * 64 : (NL == 2 ? 32 * 64 : 1)];
    float *pos = xch + 128;
    const int lane = threadIdx.x, lc = lane & 31;
    WBuf wp;
    wp.init(R.wp);
    // MODEL 3: DAG-Net whose goal head (and its CE term) runs outside (wide_head.hip); steps [t_begin, t_end) per launch,
    // hidden state read from / written back to h_out, loss partials accumulated over the launches
    constexpr bool DAG = MODEL >= 2, HEAD = MODEL == 2 || MODEL == 4, EXT = MODEL == 3, GRAPH = MODEL != 0;
    constexpr int GT = V2_GT;
    constexpr int DT = (EXT || MODEL == 4) ? 2 : 1;
    constexpr int MLT = ZD == 32 ? 2 : 1;
    if constexpr (GRAPH && GM == 1) gat_fold_to_lds(gfold, wp, W.g_hid, lane);
    if constexpr (HEAD && GM == 1) gat_fold_to_lds(gfold + GAT_FOLD_FLOATS, wp, W.g_goal, lane);

    // ---- rows of this wave (K = 1: unit = scene)
    int s0 = lc, n = 1, grow = -1, sid = 0;
    {
        const int u0 = R.tile_u0[blockIdx.x], nu = R.tile_nu[blockIdx.x];
        int row = 0;
        for (int i = 0; i < nu; ++i) {
            const int s = u0 + i;
            const int b = R.scene_ptr[s], cnt = R.scene_ptr[s + 1] - b;
            if (lc >= row && lc < row + cnt) { s0 = row; n = cnt; grow = b + (lc - row); sid = s; }
            row += cnt;
        }
    }
    const bool live = grow >= 0;
    const int64_t NK = R.N;
    const int64_t arow_off = (GRAPH && live) ? R.adj_off[sid] + (int64_t)(lc - s0) * (R.adj_ld ? R.adj_ld : (int64_t)n) : 0;

    T16 h[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) zero(h[l]);
    if constexpr (EXT) {
        const int lh0 = lane >> 5;
        if (live) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float4 v = *reinterpret_cast<const float4 *>(R.h_out + ((int64_t)l * NK + grow) * 64 + nt * 32 + 8 * a + 4 * lh0);
                        h[l][nt][4 * a] = v.x; h[l][nt][4 * a + 1] = v.y; h[l][nt][4 * a + 2] = v.z; h[l][nt][4 * a + 3] = v.w;
                    }
        }
    }
    double kld_acc = 0.0, nll_acc = 0.0, ce_acc = 0.0;
    unsigned amask = 0;

    const int first_off = HEAD ? W.dec_goal0.w[0] : W.phi_x2.w[0];
    const int t_begin = EXT ? R.t_begin : 1, t_end = EXT ? R.t_end : R.T;
    WBlk pre = wload(wp, first_off, lane);
    const int lane0 = lane, s00 = s0, n0 = n;
    for (int t = t_begin; t < t_end; ++t) {
        int lane = lane0, s0 = s00, n = n0;
        asm volatile("" : "+v"(lane), "+v"(s0), "+v"(n));
        const int lh = lane >> 5, lc = lane & 31;
        const int64_t trow = (int64_t)t * NK + (live ? grow : 0);
        const float2 xt = live ? *reinterpret_cast<const float2 *>(R.traj_rel + trow * 2) : make_float2(0.f, 0.f);
        if constexpr (GRAPH) {
            // adjacency mask of the step: bit r <-> tile row j = FR(r) + 4*lh is in this row's scene and adj[i][j] > 0
            const float *arow = R.adj + (int64_t)t * R.adj_tstride + arow_off;
            unsigned mk = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = FR(r) + 4 * lh;
                const bool in = (j >= s0) & (j < s0 + n) & live;
                const float v = arow[in ? j - s0 : 0];
                const bool on = in ? v > 0.f : (j == lc);        // padding lanes: a one-row scene of their own
                mk |= on ? (1u << r) : 0u;
            }
            amask = mk;
        }
        // GCN: the adjacency VALUES of the step (same addressing as the mask above)
        auto adjval = [&](int r) -> float {
            const int j = FR(r) + 4 * lh;
            const bool in = (j >= s0) & (j < s0 + n) & live;
            const float v = (R.adj + (int64_t)t * R.adj_tstride + arow_off)[in ? j - s0 : 0];
            return in ? v : (j == lc ? 1.0f : 0.f);
        };
        int gt = 0;
        T16 d[DT];
        auto build_disposition = [&]() {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = dt * 32 + FR(r) + 4 * lh, j = f >> 1;
                    const float pv = pos[min(s0 + j, 31) * 2 + (f & 1)];
                    d[dt][r] = ((f < R.D) & (j < n)) ? pv : 0.f;
                }
        };
        if constexpr (DAG) {
            const float2 pa = live ? *reinterpret_cast<const float2 *>(R.traj + trow * 2) : make_float2(0.f, 0.f);
            __syncthreads();
            if (lh == 0) { pos[lc * 2] = pa.x; pos[lc * 2 + 1] = pa.y; }
            __syncthreads();
            gt = live ? min(max(R.goals[trow], 0), R.G - 1) : 0;
        }
        if constexpr (HEAD) {
            const int gprev = live ? min(max(R.goals[trow - NK], 0), R.G - 1) : 0;
            if constexpr (NL == 2) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stash[((2 + nt) * 16 + r) * 64 + lane] = h[0][nt][r];
            }
            build_disposition();
            // ---------------------------------------------------------------- goal head (teacher forced: g_{t-1} in, CE on g_t)
            T16 tg[2];
            zero(tg);
            pre = mm<2, DT, false>(tg, wp, W.dec_goal0.w[0], d, lane, pre, W.dec_goal0.w[1]);
            pre = mm<2, 2, false>(tg, wp, W.dec_goal0.w[1], h[NL - 1], lane, pre, W.dec_goal2.w[0]);
            finish<2, 2>(tg, wp, W.dec_goal0, gprev, lh);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) stash[(nt * 16 + r) * 64 + lane] = h[NL - 1][nt][r];
            T16 decg[GT];
            zero(decg);
            pre = mm<GT, 2, false>(decg, wp, W.dec_goal2.w[0], tg, lane, pre, W.lg_goals.w[0]);
            finish<0, GT>(decg, wp, W.dec_goal2, 0, lh);
            softmax_feat<GT>(decg);
            T16 gc[GT];
            zero(gc);
            pre = mm<GT, GT, false>(gc, wp, W.lg_goals.w[0], decg, lane, pre, W.g_goal.heads);
            T16 gg[GT];
            if constexpr (GM == 2) pre = gcn<GT, GT>(decg, gg, wp, W.g_goal, adjval, lane, pre, W.lg_goals.w[1]);
            else pre = gat<GT, GT>(decg, gg, wp, W.g_goal, amask, R.alpha, xch, gfold + GAT_FOLD_FLOATS, lane, pre, W.lg_goals.w[1]);
            pre = mm<GT, GT, false>(gc, wp, W.lg_goals.w[1], gg, lane, pre, W.phi_x2.w[0]);
            finish<0, GT>(gc, wp, W.lg_goals, 0, lh);
            softmax_feat<GT>(gc);
            // cross_entropy -= sum(g_t * g_combined)   (probabilities, models/dagnet/model.py:162)
            float pv = 0.f;
#pragma unroll
            for (int nt = 0; nt < GT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) pv += (nt * 32 + FR(r) + 4 * lh == gt) ? gc[nt][r] : 0.f;
            pv += xhalf(pv);
            ce_acc -= (double)pv;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[NL - 1][nt][r] = stash[(nt * 16 + r) * 64 + lane];
        }
        // -------------------------------------------------------------------- phi_x(x_t)
        T16 ta[2], tb[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int fs = nt * 32 + 8 * a;
                const float4 wa = wp.ld4(lh * 32, W.phi_x0_w + fs * 2);
                const float4 wb = wp.ld4(lh * 32, W.phi_x0_w + fs * 2 + 4);
                const float4 bb = wp.ld4(lh * 16, W.phi_x0_b + fs);
                ta[nt][4 * a + 0] = lrelu(wa.x * xt.x + wa.y * xt.y + bb.x, 0.01f);
                ta[nt][4 * a + 1] = lrelu(wa.z * xt.x + wa.w * xt.y + bb.y, 0.01f);
                ta[nt][4 * a + 2] = lrelu(wb.x * xt.x + wb.y * xt.y + bb.z, 0.01f);
                ta[nt][4 * a + 3] = lrelu(wb.z * xt.x + wb.w * xt.y + bb.w, 0.01f);
            }
        T16 px[2];
        zero(px);
        pre = mm<2, 2, false>(px, wp, W.phi_x2.w[0], ta, lane, pre, W.enc0.w[0]);
        finish<1, 2>(px, wp, W.phi_x2, 0, lh);
        // -------------------------------------------------------------------- encoder
        zero(ta);
        pre = mm<2, 2, false>(ta, wp, W.enc0.w[0], px, lane, pre, W.enc0.w[1]);
        pre = mm<2, 2, false>(ta, wp, W.enc0.w[1], h[NL - 1], lane, pre, W.enc2.w[0]);
        finish<1, 2>(ta, wp, W.enc0, gt, lh);
        zero(tb);
        pre = mm<2, 2, false>(tb, wp, W.enc2.w[0], ta, lane, pre, W.enc_ml.w[0]);
        finish<1, 2>(tb, wp, W.enc2, 0, lh);
        T16 eml[MLT];
        zero(eml);
        pre = mm<MLT, 2, false>(eml, wp, W.enc_ml.w[0], tb, lane, pre, W.prior0.w[0]);
        finish<0, MLT>(eml, wp, W.enc_ml, 0, lh);
        // -------------------------------------------------------------------- prior
        zero(ta);
        pre = mm<2, 2, false>(ta, wp, W.prior0.w[0], h[NL - 1], lane, pre, W.prior2.w[0]);
        finish<1, 2>(ta, wp, W.prior0, gt, lh);
        zero(tb);
        pre = mm<2, 2, false>(tb, wp, W.prior2.w[0], ta, lane, pre, W.prior_ml.w[0]);
        finish<1, 2>(tb, wp, W.prior2, 0, lh);
        T16 pml[MLT];
        zero(pml);
        pre = mm<MLT, 2, false>(pml, wp, W.prior_ml.w[0], tb, lane, pre, W.phi_z0.w[0]);
        finish<0, MLT>(pml, wp, W.prior_ml, 0, lh);
        // -------------------------------------------------------------------- KLD (models/dagnet/model.py:123-128), z from the encoder
        T16 z[1];
        {
            float e = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int f0 = 8 * a + 4 * lh;
                float e4[4] = {0.f, 0.f, 0.f, 0.f};
                if (f0 < ZD) {
                    if (R.eps) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) e4[i] = live ? R.eps[(trow - NK) * ZD + f0 + i] : 0.f;
                    } else {
                        const uint4 u = philox4x32(make_uint4((uint32_t)(live ? grow : 0), (uint32_t)t, (uint32_t)(f0 >> 2), 0u),
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
                        const float me = eml[0][r], le = eml[MLT - 1][r], mp = pml[0][r], lp = pml[MLT - 1][r];
                        v = e4[i] * fexp(0.5f * le) + me;
                        const float dm = me - mp;
                        e += (lp - le) + fexp(le - lp) + dm * dm * frcp(fexp(lp));
                    } else {
                        if (a < 2) {
                            const float me = eml[0][r], le = eml[0][(r + 8) & 15], mp = pml[0][r], lp = pml[0][(r + 8) & 15];
                            v = e4[i] * fexp(0.5f * le) + me;
                            const float dm = me - mp;
                            e += (lp - le) + fexp(le - lp) + dm * dm * frcp(fexp(lp));
                        }
                    }
                    z[0][r] = v;
                }
            }
            e += xhalf(e);
            kld_acc += 0.5 * (double)(e - (float)ZD);
        }
        // -------------------------------------------------------------------- phi_z, decoder, NLL
        zero(ta);
        pre = mm<2, 1, false>(ta, wp, W.phi_z0.w[0], z, lane, pre, W.phi_z2.w[0]);
        finish<1, 2>(ta, wp, W.phi_z0, 0, lh);
        T16 pz[2];
        zero(pz);
        pre = mm<2, 2, false>(pz, wp, W.phi_z2.w[0], ta, lane, pre, W.dec0.w[0]);
        finish<1, 2>(pz, wp, W.phi_z2, 0, lh);
        zero(ta);
        if constexpr (DAG) {
            build_disposition();
            pre = mm<2, DT, false>(ta, wp, W.dec0.w[0], d, lane, pre, W.dec0.w[1]);
            pre = mm<2, 2, false>(ta, wp, W.dec0.w[1], pz, lane, pre, W.dec0.w[2]);
            pre = mm<2, 2, false>(ta, wp, W.dec0.w[2], h[NL - 1], lane, pre, W.dec2.w[0]);
        } else {
            pre = mm<2, 2, false>(ta, wp, W.dec0.w[0], pz, lane, pre, W.dec0.w[1]);
            pre = mm<2, 2, false>(ta, wp, W.dec0.w[1], h[NL - 1], lane, pre, W.dec2.w[0]);
        }
        finish<1, 2>(ta, wp, W.dec0, gt, lh);
        zero(tb);
        pre = mm<2, 2, false>(tb, wp, W.dec2.w[0], ta, lane, pre, W.gru[0].nh.w[0]);
        finish<1, 2>(tb, wp, W.dec2, 0, lh);
        {
            float mx_ = 0.f, my_ = 0.f, lx_ = 0.f, ly_ = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int fs = nt * 32 + 8 * a;
                    const float4 w0 = wp.ld4(lh * 16, W.dec_mean_w + fs), w1 = wp.ld4(lh * 16, W.dec_mean_w + 64 + fs);
                    const float4 v0 = wp.ld4(lh * 16, W.dec_logvar_w + fs), v1 = wp.ld4(lh * 16, W.dec_logvar_w + 64 + fs);
                    const float t0 = tb[nt][4 * a], t1 = tb[nt][4 * a + 1], t2 = tb[nt][4 * a + 2], t3 = tb[nt][4 * a + 3];
                    mx_ += w0.x * t0 + w0.y * t1 + w0.z * t2 + w0.w * t3;
                    my_ += w1.x * t0 + w1.y * t1 + w1.z * t2 + w1.w * t3;
                    lx_ += v0.x * t0 + v0.y * t1 + v0.z * t2 + v0.w * t3;
                    ly_ += v1.x * t0 + v1.y * t1 + v1.z * t2 + v1.w * t3;
                }
            mx_ += xhalf(mx_); my_ += xhalf(my_); lx_ += xhalf(lx_); ly_ += xhalf(ly_);
            mx_ += wp.ld1(0, W.dec_mean_b); my_ += wp.ld1(0, W.dec_mean_b + 1);
            lx_ += wp.ld1(0, W.dec_logvar_b); ly_ += wp.ld1(0, W.dec_logvar_b + 1);
            // NLL of the row (models/dagnet/model.py:130-135)
            const float dx = xt.x - mx_, dy = xt.y - my_;
            const float e = dx * dx * frcp(fexp(lx_)) + lx_ + dy * dy * frcp(fexp(ly_)) + ly_ + 2.0f * 1.8378770664093453f;
            nll_acc += 0.5 * (double)e;
        }
        // -------------------------------------------------------------------- GRU
        T16 xin[4];
        xin[0] = px[0]; xin[1] = px[1]; xin[2] = pz[0]; xin[3] = pz[1];
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
        pre = gru<4>(xin, h[0], wp, W.gru[0], lane, pre, NL == 2 ? W.gru[1].nh.w[0] : after_gru);
        if constexpr (NL == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[1][nt][r] = stash[(nt * 16 + r) * 64 + lane];
            pre = gru<2>(h[0], h[1], wp, W.gru[1], lane, pre, after_gru);
        }
        // -------------------------------------------------------------------- graph refinement of h[-1]
        if constexpr (GRAPH) {
            T16 hg[2];
            if constexpr (GM == 2) pre = gcn<2, 2>(h[NL - 1], hg, wp, W.g_hid, adjval, lane, pre, W.blend.w[0]);
            else pre = gat<2, 2>(h[NL - 1], hg, wp, W.g_hid, amask, R.alpha, xch, gfold, lane, pre, W.blend.w[0]);
            zero(ta);
            pre = mm<2, 2, false>(ta, wp, W.blend.w[0], hg, lane, pre, W.blend.w[1]);
            pre = mm<2, 2, false>(ta, wp, W.blend.w[1], h[NL - 1], lane, pre, first_off);
            finish<0, 2>(ta, wp, W.blend, 0, lh);
            h[NL - 1][0] = ta[0];
            h[NL - 1][1] = ta[1];
        }
    }
    // ---- outputs: h [L,N,R] and the tile's loss partials (each row's sums sit in both lane halves: count half 0)
    const int lh = lane >> 5;
    if (live) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    *reinterpret_cast<float4 *>(R.h_out + ((int64_t)l * NK + grow) * 64 + nt * 32 + 8 * a + 4 * lh) =
                        make_float4(h[l][nt][4 * a], h[l][nt][4 * a + 1], h[l][nt][4 * a + 2], h[l][nt][4 * a + 3]);
    }
    const bool count = live && lh == 0;
    const double k = wave_sum_d(count ? kld_acc : 0.0), nl = wave_sum_d(count ? nll_acc : 0.0), c = wave_sum_d(count ? ce_acc : 0.0);
    if (lane == 0) {
        double *lp = R.loss_part + (int64_t)blockIdx.x * 3;
        if (EXT) { lp[0] += k; lp[1] += nl; }       // one launch per step: accumulate (zeroed by the host; CE comes from outside)
        else { lp[0] = k; lp[1] = nl; lp[2] = c; }
    }
}

}   // namespace v2

template <int MODEL, int NL, int ZD, int GM>
static hipError_t launch_fwd(const V2Weights *d_w, const V2Run &r, hipStream_t st) {
    hipLaunchKernelGGL((v2::forward_v2_kernel<MODEL, NL, ZD, GM>), dim3(r.n_tiles), dim3(64), 0, st, d_w, r);
    return hipGetLastError();
}

#define V2F_INSTANCES                                                                                                  \
    V2_CASE(2, 2, 32, 1) V2_CASE(1, 1, 32, 1) V2_CASE(0, 2, 32, 1) V2_CASE(1, 1, 16, 1) V2_CASE(0, 2, 16, 1) V2_CASE(1, 2, 32, 1) \
    V2_CASE(3, 2, 32, 1) V2_CASE(4, 2, 32, 1) V2_CASE(2, 2, 32, 2) V2_CASE(1, 1, 32, 2) V2_CASE(1, 1, 16, 2) V2_CASE(1, 2, 32, 2) \
    V2_CASE(2, 1, 32, 1) V2_CASE(2, 2, 16, 1) V2_CASE(4, 1, 32, 1) V2_CASE(4, 2, 16, 1) V2_CASE(3, 1, 32, 1) V2_CASE(3, 2, 16, 1) \
    V2_CASE(2, 1, 32, 2) V2_CASE(2, 2, 16, 2)

extern "C" hipError_t dn_launch_forward_v2(int model, int n_layers, int z_dim, int graph, const V2Weights *d_w, const V2Run *r,
                                           hipStream_t st) {
    const int gm = graph == 2 ? 2 : 1;
#define V2_CASE(M, L, Z, G) \
    if (model == M && n_layers == L && z_dim == Z && gm == G) return launch_fwd<M, L, Z, G>(d_w, *r, st);
    V2F_INSTANCES
#undef V2_CASE
    return hipErrorInvalidValue;
}
