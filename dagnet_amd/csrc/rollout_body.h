// Body of the generic LDS-tiled rollout kernel; included twice by rollout.hip:
//   DN_BIG 0 -> every activation buffer in LDS                       (namespace dn_lds, kernel dagnet_rollout_kernel)
//   DN_BIG 1 -> goal-width buffers in an L2-resident global slab      (namespace dn_big, kernel dagnet_rollout_big_kernel)
namespace DN_NS {

struct DnCtx {
    const DnDims &d;
    const DnLds &l;
    const DnWeights &w;
    const DnRun &r;
    float *gs;        // DN_BIG: this workgroup's slab of global scratch (buffers whose offset is >= l.total)
};

// Activation buffers are addressed by a float offset.  Offsets below l.total are LDS; in the DN_BIG build the
// goal-width buffers (g_dim in the hundreds: 3 x pad32(G) x M floats do not fit the LDS) live in a per-workgroup
// slab of global memory that stays L2 resident, addressed by the offsets from l.total upwards.
#if DN_BIG
__device__ __forceinline__ float *dn_mem(const DnCtx &P, int i) { return i < P.l.total ? smem + i : P.gs + (i - P.l.total); }
#define MEM(i) (*dn_mem(P, (i)))
#else
#define MEM(i) smem[(i)]
#endif

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---------------------------------------------------------------------------------------------
// MFMA GEMM over LDS activations.  Every (n-tile, m-tile) pair of 32x32 outputs is owned by one
// wave; lane l of the result holds column m = l&31 and features (r&3)+8*(r>>2)+4*(l>>5).
// ---------------------------------------------------------------------------------------------
struct GemmIn {
    int off[DN_MAX_SEG];
};

// k-groups of all segments of a layer as one sequence: g -> (weight fragment pointer, LDS offset)
struct KSeq {
    const float4 *w[DN_MAX_SEG];
    int b[DN_MAX_SEG];
    int c1, c2, total;
    __device__ __forceinline__ const float4 *wptr(int g) const {
        return g < c1 ? w[0] + g * 64 : (g < c2 ? w[1] + (g - c1) * 64 : w[2] + (g - c2) * 64);
    }
    __device__ __forceinline__ int boff(int g, int M) const {
        return g < c1 ? b[0] + g * 8 * M : (g < c2 ? b[1] + (g - c1) * 8 * M : b[2] + (g - c2) * 8 * M);
    }
};

__device__ __forceinline__ KSeq make_kseq(const float *wp, const DnLin &L, const int *in_off, int nt, int lane) {
    KSeq q;
    int cum = 0;
#pragma unroll
    for (int s = 0; s < DN_MAX_SEG; ++s) {
        const bool on = s < L.nseg;
        const int kg = on ? L.kg[s] : 0;
        q.w[s] = reinterpret_cast<const float4 *>(wp + (on ? L.w[s] : L.w[0])) + (size_t)nt * kg * 64 + lane;
        q.b[s] = on ? in_off[s] : 0;
        cum += kg;
        if (s == 0) q.c1 = cum;
        if (s == 1) q.c2 = cum;
    }
    q.total = cum;
    return q;
}

#define DN_CHG 4  // same for the three weight streams of a GRU layer
#define DN_CH 8   // k-groups per prefetch chunk: 8 x float4 of weights in flight per accumulator stream

template <class Pre, class Epi>
__device__ __forceinline__ void gemm(const DnCtx &P, const DnLin &L, const GemmIn &in, Pre pre, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lm = lane & 31, lh = lane >> 5;
    const int M = P.l.M, MT = M >> 5;
    const int npairs = L.nt * MT;
    for (int pair = wave; pair < npairs; pair += DN_WAVES) {
        const int nt = pair / MT, mt = pair - nt * MT;
        const int m = mt * 32 + lm;
        const KSeq q = make_kseq(P.r.wp, L, in.off, nt, lane);
        const int bl = lh * M + m;
        float4 a0[DN_CH], a1[DN_CH];
        float add[16];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        auto loadw = [&](float4(&a)[DN_CH], int c) {
#pragma unroll
            for (int j = 0; j < DN_CH; ++j) {
                const int g = min(c * DN_CH + j, q.total - 1);
                a[j] = *q.wptr(g);
            }
        };
        auto compute = [&](const float4(&a)[DN_CH], int c) {
#pragma unroll
            for (int j = 0; j < DN_CH; ++j) {
                const int g = c * DN_CH + j;
                if (g < q.total) {
                    const int bo = q.boff(g, M) + bl;
                    const float b0 = MEM(bo), b1 = MEM(bo + 2 * M), b2 = MEM(bo + 4 * M), b3 = MEM(bo + 6 * M);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b3, acc, 0, 0, 0);
                }
            }
        };
        const int nch = (q.total + DN_CH - 1) / DN_CH;
        loadw(a0, 0);
        // epilogue addends (bias, one-hot gather) are fetched while the first weights are in flight
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = pre(nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, m);
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) loadw(a1, c + 1);
            compute(a0, c);
            if (c + 1 < nch) {
                if (c + 2 < nch) loadw(a0, c + 2);
                compute(a1, c + 1);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            epi(f, m, acc[r] + add[r]);
        }
    }
}

// standard epilogue: bias (+ one-hot column gather) + activation -> LDS buffer
// ACT: 0 none, 1 LeakyReLU(0.01), 2 ReLU
template <int ACT>
__device__ __forceinline__ void linear(const DnCtx &P, const DnLin &L, const GemmIn &in, int out_off, const int *gsel) {
    const float *__restrict__ wp = P.r.wp;
    const int M = P.l.M;
    const int ld = L.nt * 32;
    gemm(
        P, L, in,
        [&](int f, int m) {
            float v = 0.f;
            if (L.bias >= 0) v += wp[L.bias + f];
            if (L.gather >= 0) v += wp[L.gather + gsel[m] * ld + f];
            return v;
        },
        [&](int f, int m, float v) {
            if (ACT == 1) v = lrelu(v, 0.01f);
            if (ACT == 2) v = fmaxf(v, 0.f);
            MEM(out_off + f * M + m) = v;
        });
}

// Fused GRU layer: 4 accumulators (r, z, n_input, n_hidden) per output tile, gates in the epilogue.
// k-group sequence = input segments then the hidden segment; the n gate switches accumulator there.
__device__ __forceinline__ void gru_layer(const DnCtx &P, const DnGru &G, const GemmIn &xin, int nxseg, int h_off, int out_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lm = lane & 31, lh = lane >> 5;
    const int M = P.l.M, MT = M >> 5;
    const float *__restrict__ wp = P.r.wp;
    const int npairs = G.r.nt * MT;
    int in_off[DN_MAX_SEG];
#pragma unroll
    for (int s = 0; s < DN_MAX_SEG; ++s) in_off[s] = s < nxseg ? xin.off[s] : h_off;
    for (int pair = wave; pair < npairs; pair += DN_WAVES) {
        const int nt = pair / MT, mt = pair - nt * MT;
        const int m = mt * 32 + lm;
        const int bl = lh * M + m;
        const KSeq qr = make_kseq(wp, G.r, in_off, nt, lane);
        const KSeq qz = make_kseq(wp, G.z, in_off, nt, lane);
        const KSeq qi = make_kseq(wp, G.ni, in_off, nt, lane);       // input part only
        const int kx = qi.total;                                      // k-groups before the hidden segment
        const float4 *wh = reinterpret_cast<const float4 *>(wp + G.nh.w[0]) + (size_t)nt * G.nh.kg[0] * 64 + lane;
        float4 cr0[DN_CHG], cz0[DN_CHG], cn0[DN_CHG], cr1[DN_CHG], cz1[DN_CHG], cn1[DN_CHG];
        f32x16 ar, az, ai, ah;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ar[r] = 0.f; az[r] = 0.f; ai[r] = 0.f; ah[r] = 0.f; }
        auto loadw = [&](float4(&cr)[DN_CHG], float4(&cz)[DN_CHG], float4(&cn)[DN_CHG], int c) {
#pragma unroll
            for (int j = 0; j < DN_CHG; ++j) {
                const int g = min(c * DN_CHG + j, qr.total - 1);
                cr[j] = *qr.wptr(g);
                cz[j] = *qz.wptr(g);
                cn[j] = g < kx ? *qi.wptr(g) : wh[(g - kx) * 64];
            }
        };
        auto compute = [&](const float4(&cr)[DN_CHG], const float4(&cz)[DN_CHG], const float4(&cn)[DN_CHG], int c) {
#pragma unroll
            for (int j = 0; j < DN_CHG; ++j) {
                const int g = c * DN_CHG + j;
                if (g < qr.total) {
                    const int bo = qr.boff(g, M) + bl;
                    const float b0 = MEM(bo), b1 = MEM(bo + 2 * M), b2 = MEM(bo + 4 * M), b3 = MEM(bo + 6 * M);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[j].x, b0, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_32x32x2f32(cz[j].x, b0, az, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[j].y, b1, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_32x32x2f32(cz[j].y, b1, az, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[j].z, b2, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_32x32x2f32(cz[j].z, b2, az, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_32x32x2f32(cr[j].w, b3, ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_32x32x2f32(cz[j].w, b3, az, 0, 0, 0);
                    if (g < kx) {
                        ai = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].x, b0, ai, 0, 0, 0);
                        ai = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].y, b1, ai, 0, 0, 0);
                        ai = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].z, b2, ai, 0, 0, 0);
                        ai = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].w, b3, ai, 0, 0, 0);
                    } else {
                        ah = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].x, b0, ah, 0, 0, 0);
                        ah = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].y, b1, ah, 0, 0, 0);
                        ah = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].z, b2, ah, 0, 0, 0);
                        ah = __builtin_amdgcn_mfma_f32_32x32x2f32(cn[j].w, b3, ah, 0, 0, 0);
                    }
                }
            }
        };
        const int nch = (qr.total + DN_CHG - 1) / DN_CHG;
        loadw(cr0, cz0, cn0, 0);
        float br[16], bz[16], bi[16], bh[16], hold[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            br[r] = wp[G.r.bias + f];
            bz[r] = wp[G.z.bias + f];
            bi[r] = wp[G.ni.bias + f];
            bh[r] = wp[G.nh.bias + f];
            hold[r] = MEM(h_off + f * M + m);
        }
        for (int c = 0; c < nch; c += 2) {
            if (c + 1 < nch) loadw(cr1, cz1, cn1, c + 1);
            compute(cr0, cz0, cn0, c);
            if (c + 1 < nch) {
                if (c + 2 < nch) loadw(cr0, cz0, cn0, c + 2);
                compute(cr1, cz1, cn1, c + 1);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float rg = sigmoidf_(ar[r] + br[r]);
            const float zg = sigmoidf_(az[r] + bz[r]);
            const float ng = tanhf((ai[r] + bi[r]) + rg * (ah[r] + bh[r]));
            MEM(out_off + f * M + m) = (1.0f - zg) * ng + zg * hold[r];
        }
    }
}

__device__ __forceinline__ void copy_rows(const DnCtx &P, int dst, int src, int nfloats) {
    for (int i = threadIdx.x; i < nfloats; i += DN_THREADS) MEM(dst + i) = MEM(src + i);
}

// ---------------------------------------------------------------------------------------------
// Per-scene graph layers on the tile.  Row m belongs to the scene whose rows are
// [ss[m], ss[m]+sn[m]) of the tile; A[m][jl] (stride AS) is the adjacency of m to scene member jl.
// ---------------------------------------------------------------------------------------------
struct RowInfo {
    const int *ss, *sn;
};

// in_off [nin_pad][M] -> out_off [pad32(nout)][M]; wh2_off: scratch [pad32(nout)][M] (!= in, != out)
__device__ __forceinline__ void graph_gat(const DnCtx &P, const DnGraph &Gr, const RowInfo &ri, int in_off, int wh2_off, int out_off) {
    const int M = P.l.M, AS = P.l.AS;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const int F = P.d.F, NP = P.d.P;
    const float alpha = P.d.alpha;
    const float *__restrict__ wp = P.r.wp;
    const float *A = smem + P.l.adj;
    const int n = ri.sn[m], s0 = ri.ss[m];
    GemmIn in;
    in.off[0] = in_off;
    // Wh for all heads: [P*F][M]
    linear<0>(P, Gr.heads, in, P.l.wh, nullptr);
    __syncthreads();
    // s1[p][m] = a[:F].Wh_m ; s2[p][m] = a[F:].Wh_m
    for (int p = sl; p < NP; p += nsl) {
        float s1 = 0.f, s2 = 0.f;
        for (int f = 0; f < F; ++f) {
            const float v = MEM(P.l.wh + (p * F + f) * M + m);
            s1 += wp[Gr.a_heads + p * 2 * F + f] * v;
            s2 += wp[Gr.a_heads + p * 2 * F + F + f] * v;
        }
        MEM(P.l.s12 + p * M + m) = s1;
        MEM(P.l.s12 + (NP + p) * M + m) = s2;
    }
    __syncthreads();
    // attention heads -> x2 = elu(softmax(e) @ Wh)
    for (int p = sl; p < NP; p += nsl) {
        const float s1 = MEM(P.l.s12 + p * M + m);
        float mx = -INFINITY;
        for (int jl = 0; jl < n; ++jl) {
            float e = lrelu(s1 + MEM(P.l.s12 + (NP + p) * M + s0 + jl), alpha);
            e = A[m * AS + jl] > 0.f ? e : -9e15f;
            mx = fmaxf(mx, e);
        }
        float acc[DN_FCAP];
#pragma unroll
        for (int f = 0; f < DN_FCAP; ++f) acc[f] = 0.f;
        float sum = 0.f;
        for (int jl = 0; jl < n; ++jl) {
            float e = lrelu(s1 + MEM(P.l.s12 + (NP + p) * M + s0 + jl), alpha);
            e = A[m * AS + jl] > 0.f ? e : -9e15f;
            const float w = expf(e - mx);
            sum += w;
#pragma unroll
            for (int f = 0; f < DN_FCAP; ++f)
                if (f < F) acc[f] += w * MEM(P.l.wh + (p * F + f) * M + s0 + jl);
        }
#pragma unroll
        for (int f = 0; f < DN_FCAP; ++f)
            if (f < F) {
                float v = n > 0 ? acc[f] / sum : 0.f;
                v = v > 0.f ? v : expm1f(v);
                MEM(P.l.x2 + (p * F + f) * M + m) = v;
            }
    }
    // zero the K padding rows of x2 (P*F .. pad8)
    {
        const int pf = NP * F, pf8 = (pf + 7) & ~7;
        for (int f = pf + sl; f < pf8; f += nsl) MEM(P.l.x2 + f * M + m) = 0.f;
    }
    __syncthreads();
    // out_att: Wh2 = x2 @ W_out  [nout][M]
    in.off[0] = P.l.x2;
    linear<0>(P, Gr.out, in, wh2_off, nullptr);
    __syncthreads();
    const int nout = Gr.nout;
    {   // partial dot products a_out . Wh2_m, split over the slices
        float s1 = 0.f, s2 = 0.f;
        for (int c = sl; c < nout; c += nsl) {
            const float v = MEM(wh2_off + c * M + m);
            s1 += wp[Gr.a_out + c] * v;
            s2 += wp[Gr.a_out + nout + c] * v;
        }
        MEM(P.l.red + sl * M + m) = s1;
        MEM(P.l.red + (nsl + sl) * M + m) = s2;
    }
    __syncthreads();
    if (sl == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < nsl; ++i) {
            s1 += MEM(P.l.red + i * M + m);
            s2 += MEM(P.l.red + (nsl + i) * M + m);
        }
        MEM(P.l.s12 + m) = s1;
        MEM(P.l.s12 + M + m) = s2;
    }
    __syncthreads();
    // un-normalised attention weights of row m -> att[m][jl]; every slice computes the row max
    {
        const float s1 = MEM(P.l.s12 + m);
        float mx = -INFINITY;
        for (int jl = 0; jl < n; ++jl) {
            float e = lrelu(s1 + MEM(P.l.s12 + M + s0 + jl), alpha);
            e = A[m * AS + jl] > 0.f ? e : -9e15f;
            mx = fmaxf(mx, e);
        }
        for (int jl = sl; jl < n; jl += nsl) {
            float e = lrelu(s1 + MEM(P.l.s12 + M + s0 + jl), alpha);
            e = A[m * AS + jl] > 0.f ? e : -9e15f;
            MEM(P.l.att + m * AS + jl) = expf(e - mx);
        }
    }
    __syncthreads();
    {
        float sum = 0.f;
        for (int jl = 0; jl < n; ++jl) sum += MEM(P.l.att + m * AS + jl);
        const float *bn = wp + Gr.bn1;
        const int npad = Gr.out.nt * 32;
        for (int c0 = sl * 4; c0 < npad; c0 += nsl * 4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int jl = 0; jl < n; ++jl) {
                const float w = MEM(P.l.att + m * AS + jl);
                const int vo = wh2_off + c0 * M + s0 + jl;
                a0 += w * MEM(vo);
                a1 += w * MEM(vo + M);
                a2 += w * MEM(vo + 2 * M);
                a3 += w * MEM(vo + 3 * M);
            }
            const float av[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i;
                float v = 0.f;
                if (c < nout && n > 0) {
                    v = av[i] / sum;
                    v = (v - bn[2 * nout + c]) / bn[3 * nout + c] * bn[c] + bn[nout + c];
                    v = tanhf(v);
                }
                MEM(out_off + c * M + m) = v;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void graph_gcn(const DnCtx &P, const DnGraph &Gr, const RowInfo &ri, int in_off, int wh2_off, int out_off) {
    const int M = P.l.M, AS = P.l.AS;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const int F = P.d.F;
    const float *__restrict__ wp = P.r.wp;
    const float *A = smem + P.l.adj;
    const int n = ri.sn[m], s0 = ri.ss[m];
    GemmIn in;
    in.off[0] = in_off;
    linear<0>(P, Gr.gc1, in, P.l.wh, nullptr);
    __syncthreads();
    {
        const float *bn = wp + Gr.bn1;
        const int f8 = (F + 7) & ~7;
        for (int f = sl; f < f8; f += nsl) {
            float v = 0.f;
            if (f < F) {
                for (int jl = 0; jl < n; ++jl) v += A[m * AS + jl] * MEM(P.l.wh + f * M + s0 + jl);
                if (Gr.b1 >= 0) v += wp[Gr.b1 + f];
                v = (v - bn[2 * F + f]) / bn[3 * F + f] * bn[f] + bn[F + f];
            }
            MEM(P.l.x2 + f * M + m) = v;
        }
    }
    __syncthreads();
    in.off[0] = P.l.x2;
    linear<0>(P, Gr.gc2, in, wh2_off, nullptr);
    __syncthreads();
    {
        const int nout = Gr.nout;
        const float *bn = wp + Gr.bn2;
        const int npad = Gr.gc2.nt * 32;
        for (int c = sl; c < npad; c += nsl) {
            float v = 0.f;
            if (c < nout) {
                for (int jl = 0; jl < n; ++jl) v += A[m * AS + jl] * MEM(wh2_off + c * M + s0 + jl);
                if (Gr.b2 >= 0) v += wp[Gr.b2 + c];
                v = (v - bn[2 * nout + c]) / bn[3 * nout + c] * bn[c] + bn[nout + c];
                v = tanhf(v);
            }
            MEM(out_off + c * M + m) = v;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void graph_apply(const DnCtx &P, const DnGraph &Gr, const RowInfo &ri, int in_off, int wh2_off, int out_off) {
    if (Gr.kind == 1)
        graph_gat(P, Gr, ri, in_off, wh2_off, out_off);
    else
        graph_gcn(P, Gr, ri, in_off, wh2_off, out_off);
}

// adjacency of the tile from positions pos[c][row] (LDS, stride M) -> A[m][jl]
__device__ __forceinline__ void build_adjacency(const DnCtx &P, const RowInfo &ri, int pos_off) {
    const int M = P.l.M, AS = P.l.AS;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const int n = ri.sn[m], s0 = ri.ss[m];
    float *A = smem + P.l.adj;
    const int type = P.d.adj_type;
    if (type == 0) {
        for (int jl = sl; jl < n; jl += nsl) A[m * AS + jl] = 1.0f;
    } else {
        const double xi = (double)MEM(pos_off + m), yi = (double)MEM(pos_off + M + m);
        if (type == 1) {
            const double sigma = P.d.sigma;
            for (int jl = sl; jl < n; jl += nsl) {
                const double dx = xi - (double)MEM(pos_off + s0 + jl), dy = yi - (double)MEM(pos_off + M + s0 + jl);
                const double d = sqrt(dx * dx + dy * dy);
                A[m * AS + jl] = (float)exp(-d / sigma);
            }
        } else {
            const int k = P.d.knn_k < n ? P.d.knn_k : n;
            for (int jl = sl; jl < n; jl += nsl) {
                const double dx = xi - (double)MEM(pos_off + s0 + jl), dy = yi - (double)MEM(pos_off + M + s0 + jl);
                const double d = dx * dx + dy * dy;   // rank by squared distance == rank by distance
                int rank = 0;
                for (int j2 = 0; j2 < n; ++j2) {
                    const double ex = xi - (double)MEM(pos_off + s0 + j2), ey = yi - (double)MEM(pos_off + M + s0 + j2);
                    const double d2 = ex * ex + ey * ey;
                    rank += (d2 < d || (d2 == d && j2 < jl)) ? 1 : 0;
                }
                A[m * AS + jl] = rank < k ? 1.0f : 0.0f;
            }
        }
    }
}

// adjacency of the tile from a caller-provided tensor (forward / graph modes)
__device__ __forceinline__ void load_adjacency(const DnCtx &P, const RowInfo &ri, const int *sid, const float *adj_t) {
    const int M = P.l.M, AS = P.l.AS;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const int n = ri.sn[m], s0 = ri.ss[m];
    if (n == 0) return;
    const int64_t ld = P.r.adj_ld > 0 ? P.r.adj_ld : (int64_t)n;
    const float *blk = adj_t + P.r.adj_off[sid[m]] + (int64_t)(m - s0) * ld;
    float *A = smem + P.l.adj;
    for (int jl = sl; jl < n; jl += nsl) A[m * AS + jl] = blk[jl];
}

// row softmax over the first G features of buf[g][m], in place; 4 barriers
__device__ __forceinline__ void softmax_rows(const DnCtx &P, int buf, int G) {
    const int M = P.l.M;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    float mx = -INFINITY;
    for (int g = sl; g < G; g += nsl) mx = fmaxf(mx, MEM(buf + g * M + m));
    MEM(P.l.red + sl * M + m) = mx;
    __syncthreads();
    mx = MEM(P.l.red + m);
    for (int i = 1; i < nsl; ++i) mx = fmaxf(mx, MEM(P.l.red + i * M + m));
    float sum = 0.f;
    for (int g = sl; g < G; g += nsl) {
        const float e = expf(MEM(buf + g * M + m) - mx);
        MEM(buf + g * M + m) = e;
        sum += e;
    }
    __syncthreads();
    MEM(P.l.red + sl * M + m) = sum;
    __syncthreads();
    sum = 0.f;
    for (int i = 0; i < nsl; ++i) sum += MEM(P.l.red + i * M + m);
    for (int g = sl; g < G; g += nsl) MEM(buf + g * M + m) = MEM(buf + g * M + m) / sum;
    __syncthreads();
}

// VALU linear for tiny K or N: out[n][m] = act(b[n] + sum_k W[n][k] in[k][m])
template <int ACT>
__device__ __forceinline__ void small_linear(const DnCtx &P, const DnRaw &L, int in_off, int out_off, int npad) {
    const int M = P.l.M;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const float *__restrict__ wp = P.r.wp;
    for (int nn = sl; nn < npad; nn += nsl) {
        float v = 0.f;
        if (nn < L.n) {
            for (int k = 0; k < L.k; ++k) v += wp[L.w + nn * L.k + k] * MEM(in_off + k * M + m);
            v += wp[L.b + nn];
            if (ACT == 1) v = lrelu(v, 0.01f);
        }
        MEM(out_off + nn * M + m) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void run_tile(const DnCtx &P, const int tile) {
    const DnDims &D = P.d;
    const DnLds &L = P.l;
    const DnWeights &W = P.w;
    const DnRun &R = P.r;
    const int M = L.M;
    const int tid = threadIdx.x;
    const int m = tid % M, sl = tid / M, nsl = DN_THREADS / M;
    const bool dag = D.model == 2, avr = D.model == 1;
    const bool has_graph = D.model != 0;
    int *ri_ss = reinterpret_cast<int *>(smem + L.ri);
    int *ri_sn = ri_ss + M;
    int *ri_grow = ri_sn + M;      // base row (0..N) of tile row, -1 for padding rows
    int *ri_gidx = ri_grow + M;    // current goal index
    int *ri_gprev = ri_gidx + M;   // previous goal (forward mode)
    int *ri_k = ri_gprev + M;      // sample index of the row
    int *ri_sid = ri_k + M;        // scene id
    RowInfo ri{ri_ss, ri_sn};

    // ---- zero LDS, build the row tables ----
    for (int i = tid; i < L.total; i += DN_THREADS) MEM(i) = 0.f;
    __syncthreads();
    if (tid < M) {
        ri_grow[tid] = -1;
    }
    __syncthreads();
    if (tid == 0) {
        const int u0 = R.tile_u0[tile], nu = R.tile_nu[tile];
        int row = 0;
        for (int i = 0; i < nu; ++i) {
            const int u = u0 + i;
            const int k = u / R.S, s = u - k * R.S;
            const int b = R.scene_ptr[s], n = R.scene_ptr[s + 1] - b;
            for (int a = 0; a < n && row + a < M; ++a) {
                ri_ss[row + a] = row;
                ri_sn[row + a] = n;
                ri_grow[row + a] = b + a;
                ri_k[row + a] = k;
                ri_sid[row + a] = s;
            }
            row += n;
        }
    }
    __syncthreads();
    const int grow = ri_grow[m];           // -1 on padding rows
    const int krow = ri_k[m];
    const bool live = grow >= 0;
    const int64_t NK = (int64_t)R.N;

    // ---- initial state ----
    if (MODE == MODE_SAMPLE) {
        for (int l = 0; l < D.L; ++l)
            for (int i = tid; i < M * D.R; i += DN_THREADS) {
                const int mm = i / D.R, f = i - mm * D.R;
                const int g = ri_grow[mm];
                if (g >= 0) MEM(L.h[l] + f * M + mm) = R.h0[((int64_t)l * R.N + g) * D.R + f];
            }
        if (sl == 0 && live) {
            if (has_graph) {
                MEM(L.xabs + m) = R.x0[grow * 2 + 0];
                MEM(L.xabs + M + m) = R.x0[grow * 2 + 1];
            }
            if (dag) ri_gidx[m] = min(max(R.g0[grow], 0), D.G - 1);
        }
    } else if (MODE == MODE_GRAPH) {
        const DnGraph &Gr = R.gwhich == 0 ? W.g_hid : W.g_goal;
        const int in_off = R.gwhich == 0 ? L.h[0] : L.decg;
        for (int i = tid; i < M * Gr.nin; i += DN_THREADS) {
            const int mm = i / Gr.nin, f = i - mm * Gr.nin;
            const int g = ri_grow[mm];
            if (g >= 0) MEM(in_off + f * M + mm) = R.gx[(int64_t)g * Gr.nin + f];
        }
        load_adjacency(P, ri, ri_sid, R.adj);
        __syncthreads();
        const int wh2 = R.gwhich == 0 ? L.t2 : L.gcomb;
        const int out = R.gwhich == 0 ? L.t3 : L.gg;
        graph_apply(P, Gr, ri, in_off, wh2, out);
        for (int i = tid; i < M * Gr.nout; i += DN_THREADS) {
            const int mm = i / Gr.nout, f = i - mm * Gr.nout;
            const int g = ri_grow[mm];
            if (g >= 0) R.gout[(int64_t)g * Gr.nout + f] = MEM(out + f * M + mm);
        }
        return;
    }
    __syncthreads();

    const int hl = L.h[D.L - 1];
    const int t_begin = MODE == MODE_FORWARD ? 1 : 0;
    double kld_acc = 0.0, nll_acc = 0.0, ce_acc = 0.0;   // thread-private partials (forward)

#ifdef DN_PROFILE
    long long pstamp = clock64();
#define PH(i)                                                                     \
    do {                                                                          \
        if (R.prof && tile == 0 && tid == 0) {                                    \
            const long long now_ = clock64();                                     \
            R.prof[i] += now_ - pstamp;                                           \
            pstamp = now_;                                                        \
        }                                                                         \
    } while (0)
#else
#define PH(i)
#endif
    for (int t = t_begin; t < R.T; ++t) {
        GemmIn in;
        // ------------------------------------------------------------------ inputs of the step
        if (MODE == MODE_FORWARD) {
            if (sl == 0 && live) {
                for (int c = 0; c < D.X; ++c) MEM(L.xt + c * M + m) = R.traj_rel[((int64_t)t * R.N + grow) * D.X + c];
                if (dag) {
                    MEM(L.xabs + m) = R.traj[((int64_t)t * R.N + grow) * 2 + 0];
                    MEM(L.xabs + M + m) = R.traj[((int64_t)t * R.N + grow) * 2 + 1];
                    ri_gprev[m] = min(max(R.goals[(int64_t)(t - 1) * R.N + grow], 0), D.G - 1);
                    ri_gidx[m] = min(max(R.goals[(int64_t)t * R.N + grow], 0), D.G - 1);
                }
            }
            if (has_graph) load_adjacency(P, ri, ri_sid, R.adj + (int64_t)t * R.adj_tstride);
            __syncthreads();
        }
        if (dag) {
            // disposition d_t: flattened xy of the scene's agents, zero padded to pad8(D)
            const int dp = (D.D + 7) & ~7;
            const int n = ri_sn[m], s0 = ri_ss[m];
            for (int c = sl; c < dp; c += nsl) {
                const int j = c >> 1;
                MEM(L.d + c * M + m) = (c < D.D && j < n) ? MEM(L.xabs + (c & 1) * M + s0 + j) : 0.f;
            }
            if (MODE == MODE_SAMPLE) build_adjacency(P, ri, L.xabs);    // PRE-update positions
            __syncthreads();
            PH(0);

            // -------------------------------------------------------------- goal head
            in.off[0] = L.d; in.off[1] = hl;
            linear<2>(P, W.dec_goal0, in, L.t1, MODE == MODE_FORWARD ? ri_gprev : ri_gidx);
            __syncthreads();
            in.off[0] = L.t1;
            linear<0>(P, W.dec_goal2, in, L.decg, nullptr);
            __syncthreads();
            softmax_rows(P, L.decg, D.G);
            PH(1);
            graph_apply(P, W.g_goal, ri, L.decg, L.gcomb, L.gg);
            PH(2);
            in.off[0] = L.decg; in.off[1] = L.gg;
            linear<0>(P, W.lg_goals, in, L.gcomb, nullptr);
            __syncthreads();
            softmax_rows(P, L.gcomb, D.G);
            if (MODE == MODE_FORWARD) {
                if (sl == 0 && live) ce_acc -= (double)MEM(L.gcomb + ri_gidx[m] * M + m);
            } else {
                // categorical draw.  Injected q: argmax_g exp(p_g) / q_g (first maximum wins) - torch.multinomial's exponential race.
                // In-kernel stream: one uniform per row and step, inverse of the cumulative weights (rng.h rng_draw_u): slice sl
                // sums the chunk [sl len, (sl + 1) len) of the features; the chunk whose cumulative range holds u * total
                // resolves the feature (the LAST slice whose range starts at or below the target wins).
                float best = -1.f;
                int bi = 0;
                if (R.q) {
                    if (live) {
                        const int64_t qb = (((int64_t)krow * R.T + t) * NK + grow) * D.G;
                        for (int g = sl; g < D.G; g += nsl) {
                            const float v = expf(MEM(L.gcomb + g * M + m)) / R.q[qb + g];
                            if (v > best) { best = v; bi = g; }
                        }
                    }
                    MEM(L.red + sl * M + m) = best;
                    reinterpret_cast<int *>(smem + L.red)[(nsl + sl) * M + m] = bi;
                    __syncthreads();
                } else {
                    const int len = (D.G + nsl - 1) / nsl, g0 = min(sl * len, D.G), g1 = min(g0 + len, D.G);
                    float mysum = 0.f;
                    if (live)
                        for (int g = g0; g < g1; ++g) mysum += expf(MEM(L.gcomb + g * M + m));
                    MEM(L.red + sl * M + m) = mysum;
                    __syncthreads();
                    float before = 0.f, total = 0.f;
                    for (int i = 0; i < nsl; ++i) {
                        const float v = MEM(L.red + i * M + m);
                        before += i < sl ? v : 0.f;
                        total += v;
                    }
                    __syncthreads();                 // L.red is rewritten below
                    int cand = -1;                   // this slice's answer if the target lies at or beyond its range's start
                    if (live && g0 < g1) {
                        const float target = rng_draw_u(R.seed, (uint32_t)(krow * R.N + grow), (uint32_t)t) * total;
                        if (before <= target) {
                            float c = before;
                            cand = g1 - 1;
                            for (int g = g0; g < g1; ++g) {
                                c += expf(MEM(L.gcomb + g * M + m));
                                if (c > target) { cand = g; break; }
                            }
                        }
                    }
                    MEM(L.red + sl * M + m) = (float)cand;                       // the slices' ranges increase with sl: the largest candidate is the draw
                    reinterpret_cast<int *>(smem + L.red)[(nsl + sl) * M + m] = cand;
                    best = (float)cand;
                    bi = max(cand, 0);
                    __syncthreads();
                }
                if (sl == 0 && live) {
                    for (int i = 1; i < nsl; ++i) {
                        const float v = MEM(L.red + i * M + m);
                        const int gi = reinterpret_cast<int *>(smem + L.red)[(nsl + i) * M + m];
                        if (v > best || (v == best && gi < bi)) { best = v; bi = gi; }
                    }
                    if (R.force_goals) bi = min(max(R.force_goals[((int64_t)krow * R.T + t) * NK + grow], 0), D.G - 1);
                    ri_gidx[m] = bi;
                    if (R.out_goals) R.out_goals[((int64_t)krow * R.T + t) * NK + grow] = bi;
                }
            }
            __syncthreads();
            PH(3);
        }

        // ------------------------------------------------------------------ encoder (forward)
        if (MODE == MODE_FORWARD) {
            small_linear<1>(P, W.phi_x0, L.xt, L.t1, W.phi_x2.kg[0] * 8);
            __syncthreads();
            in.off[0] = L.t1;
            linear<1>(P, W.phi_x2, in, L.px, nullptr);
            __syncthreads();
            in.off[0] = L.px; in.off[1] = hl;
            linear<1>(P, W.enc0, in, L.t1, ri_gidx);
            __syncthreads();
            in.off[0] = L.t1;
            linear<1>(P, W.enc2, in, L.t2, nullptr);
            __syncthreads();
            in.off[0] = L.t2;
            linear<0>(P, W.enc_ml, in, L.t3, nullptr);      // rows [0,Z) mean, [Z,2Z) logvar
            __syncthreads();
        }
        // ------------------------------------------------------------------ prior
        in.off[0] = hl;
        linear<1>(P, W.prior0, in, L.t1, ri_gidx);
        __syncthreads();
        in.off[0] = L.t1;
        linear<1>(P, W.prior2, in, L.t2, nullptr);
        __syncthreads();
        in.off[0] = L.t2;
        linear<0>(P, W.prior_ml, in, L.t1, nullptr);
        __syncthreads();
        PH(4);
        // ------------------------------------------------------------------ latent sample
        {
            const int zp = (D.Z + 7) & ~7;
            const int src = MODE == MODE_FORWARD ? L.t3 : L.t1;
            if (MODE == MODE_FORWARD && live) {
                // KLD of row m, features split over the slices (models/dagnet/model.py:123-128)
                float e = 0.f;
                for (int z = sl; z < D.Z; z += nsl) {
                    const float mue = MEM(L.t3 + z * M + m), lve = MEM(L.t3 + (D.Z + z) * M + m);
                    const float mup = MEM(L.t1 + z * M + m), lvp = MEM(L.t1 + (D.Z + z) * M + m);
                    const float dm_ = mue - mup;
                    e += (lvp - lve) + expf(lve - lvp) + dm_ * dm_ / expf(lvp);
                }
                if (sl == 0) e -= (float)D.Z;
                kld_acc += 0.5 * (double)e;
            }
            // src (t1 / t3) and destination (t2) are distinct buffers: no staging needed
            for (int z = sl; z < zp; z += nsl) {
                float v = 0.f;
                if (z < D.Z && live) {
                    float ev;
                    if (MODE == MODE_FORWARD)
                        ev = R.eps ? R.eps[((int64_t)(t - 1) * R.N + grow) * D.Z + z] : rng_uniform(R.seed, grow, t, z);
                    else
                        ev = R.eps ? R.eps[(((int64_t)krow * R.T + t) * NK + grow) * D.Z + z]
                                   : rng_uniform(R.seed, (uint32_t)(krow * R.N + grow), t, z);
                    v = ev * expf(0.5f * MEM(src + (D.Z + z) * M + m)) + MEM(src + z * M + m);
                }
                MEM(L.t2 + z * M + m) = v;
            }
            __syncthreads();
        }
        PH(5);
        // ------------------------------------------------------------------ phi_z, decoder
        in.off[0] = L.t2;
        linear<1>(P, W.phi_z0, in, L.t1, nullptr);
        __syncthreads();
        in.off[0] = L.t1;
        linear<1>(P, W.phi_z2, in, L.pz, nullptr);
        __syncthreads();
        if (dag) { in.off[0] = L.d; in.off[1] = L.pz; in.off[2] = hl; }
        else     { in.off[0] = L.pz; in.off[1] = hl; }
        linear<1>(P, W.dec0, in, L.t1, ri_gidx);
        __syncthreads();
        in.off[0] = L.t1;
        linear<1>(P, W.dec2, in, L.t2, nullptr);
        __syncthreads();
        small_linear<0>(P, W.dec_mean, L.t2, L.dm, D.X);
        if (MODE == MODE_FORWARD) small_linear<0>(P, W.dec_logvar, L.t2, L.dm + D.X * M, D.X);
        __syncthreads();
        PH(6);
        if (MODE == MODE_SAMPLE) {
            if (live)
                for (int c = sl; c < D.X; c += nsl)
                    R.out[(((int64_t)krow * R.T + t) * NK + grow) * D.X + c] = MEM(L.dm + c * M + m);
            // phi_x of the predicted displacement
            small_linear<1>(P, W.phi_x0, L.dm, L.t1, W.phi_x2.kg[0] * 8);
            __syncthreads();
            in.off[0] = L.t1;
            linear<1>(P, W.phi_x2, in, L.px, nullptr);
            __syncthreads();
        } else {
            if (sl == 0 && live) {
                // NLL of row m (models/dagnet/model.py:130-135)
                float e = 0.f;
                for (int c = 0; c < D.X; ++c) {
                    const float mu = MEM(L.dm + c * M + m), lv = MEM(L.dm + (D.X + c) * M + m);
                    const float dx = MEM(L.xt + c * M + m) - mu;
                    e += dx * dx / expf(lv) + lv;
                }
                e += (float)D.X * 1.8378770664093453f;   // log(2*pi)
                nll_acc += 0.5 * (double)e;
            }
        }
        PH(7);
        // ------------------------------------------------------------------ GRU
        for (int l = 0; l < D.L; ++l) {
            GemmIn xi;
            int nx;
            if (l == 0) { xi.off[0] = L.px; xi.off[1] = L.pz; nx = 2; }
            else        { xi.off[0] = L.h[l - 1]; nx = 1; }
            gru_layer(P, W.gru[l], xi, nx, L.h[l], L.t3);
            __syncthreads();
            copy_rows(P, L.h[l], L.t3, W.gru[l].r.nt * 32 * M);
            __syncthreads();
        }
        PH(8);
        // ------------------------------------------------------------------ graph refinement
        if (MODE == MODE_SAMPLE && avr) {
            // A-VRNN builds the adjacency from the UPDATED positions (models/avrnn/model.py:201-209)
            if (sl == 0) {
                MEM(L.xabs + m) += MEM(L.dm + m);
                MEM(L.xabs + M + m) += MEM(L.dm + M + m);
            }
            __syncthreads();
            build_adjacency(P, ri, L.xabs);
            __syncthreads();
        }
        if (has_graph) {
            graph_apply(P, W.g_hid, ri, hl, L.t2, L.t3);
            in.off[0] = L.t3; in.off[1] = hl;
            linear<0>(P, W.blend, in, L.t2, nullptr);
            __syncthreads();
            copy_rows(P, hl, L.t2, W.blend.nt * 32 * M);
            __syncthreads();
        }
        PH(9);
        if (MODE == MODE_SAMPLE && dag) {
            if (sl == 0) {
                MEM(L.xabs + m) += MEM(L.dm + m);
                MEM(L.xabs + M + m) += MEM(L.dm + M + m);
            }
            __syncthreads();
        }
    }

    // ---- outputs of forward mode ----
    if (MODE == MODE_FORWARD) {
        for (int l = 0; l < D.L; ++l)
            for (int i = tid; i < M * D.R; i += DN_THREADS) {
                const int mm = i / D.R, f = i - mm * D.R;
                const int g = ri_grow[mm];
                if (g >= 0) R.h_out[((int64_t)l * R.N + g) * D.R + f] = MEM(L.h[l] + f * M + mm);
            }
        // deterministic in-tile reduction of the thread-private loss partials
        __syncthreads();
        double *red = reinterpret_cast<double *>(smem);           // 3*256 doubles = 6 KB <= total (make_layout)
        red[tid] = kld_acc;
        red[DN_THREADS + tid] = nll_acc;
        red[2 * DN_THREADS + tid] = ce_acc;
        __syncthreads();
        if (tid < 3) {
            double s = 0.0;
            for (int i = 0; i < DN_THREADS; ++i) s += red[tid * DN_THREADS + i];
            R.loss_part[(int64_t)tile * 3 + tid] = s;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(DN_THREADS, 1) void DN_KERNEL(const DnStatic *__restrict__ ps, const DnRun run) {
    const DnCtx P{ps->d, ps->l, ps->w, run, run.gscratch ? run.gscratch + (int64_t)blockIdx.x * run.gslab : nullptr};
    for (int tile = blockIdx.x; tile < run.n_tiles; tile += gridDim.x) {
        __syncthreads();
        run_tile<MODE>(P, tile);
    }
}

}   // namespace DN_NS
#undef MEM
#ifdef PH
#undef PH
#endif
