// Host side of libdagnet_hip.so: weight re-packing into MFMA fragment order, scene->tile planning,
// LDS layout, argument checks and the extern "C" entry points declared in include/dagnet_hip.h.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dagnet_hip.h"
#include "params.h"
#include "params_v2.h"
#include "params_wide.h"

extern "C" hipError_t dn_launch_rollout(int mode, const void *d_static, const DnRun *run, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t dn_launch_adjacency(int type, double sigma, int top_k, const int32_t *scene_ptr, const int32_t *row_scene,
                                          const int64_t *blk_off, int N, int T, const float *pos, float *out, int64_t ld,
                                          int64_t tstride, hipStream_t st);
extern "C" hipError_t dn_launch_best_of_k(const int32_t *scene_ptr, int S, int N, int K, int T, const float *rel,
                                          const float *start, const float *gt, float *ade, float *fde, float *scene_best,
                                          double *sums, hipStream_t st);
extern "C" hipError_t dn_launch_rollout_v2(int model, int n_layers, int z_dim, int graph, const V2Weights *w, const V2Run *r, hipStream_t st);
extern "C" hipError_t dn_launch_rollout_v2_bf3(int model, int n_layers, int z_dim, int graph, const V2Weights *w, const V2Run *r, hipStream_t st);
extern "C" int dn_v2_has(int model, int n_layers, int z_dim, int graph);
extern "C" hipError_t dn_launch_forward_v2(int model, int n_layers, int z_dim, int graph, const V2Weights *w, const V2Run *r, hipStream_t st);
extern "C" hipError_t dn_launch_loss_finalize(int n_tiles, int N, const double *part, double *losses, hipStream_t st);
extern "C" hipError_t dn_launch_nan_rows(int64_t rows, const DnNanSrc *src, const DnNanDst *dst, double *losses, void *marker,
                                         hipStream_t st);
extern "C" hipError_t dn_launch_nan_spread(const void *marker, int marker_is_double, void *base, int chunks, int64_t chunk_stride,
                                           int64_t len, hipStream_t st);
extern "C" hipError_t dn_launch_abs_errors(int T, int N, const float *pred, const float *gt, float *ade, float *fde, hipStream_t st);
extern "C" hipError_t dn_launch_scene_min_sum(const int32_t *scene_ptr, int S, int N, int K, const float *ade, const float *fde,
                                              float *scene_best, double *sums, hipStream_t st);
extern "C" hipError_t dn_launch_rel_to_abs(int T, int N, const float *rel, const float *start, float *out, hipStream_t st);
extern "C" hipError_t dn_launch_categorical(int N, int G, const float *w, const float *q, uint64_t seed, int32_t *out, hipStream_t st);

struct WideStep {                     // wide_head.hip
    const float *wp;
    WideWeights w;
    WideBuffers b;
    const int32_t *scene_ptr, *row_scene;
    const int64_t *off_compact;
    int64_t compact;
    int N, K, S, T, t;
    int D, R, H, G, L, P, F;
    int adj_type, top_k;
    double sigma;
    float alpha;
    const float *q;
    const int32_t *force_goals;
    uint64_t seed;
    int32_t *out_goals;
    const float *ext_pos;
    const int32_t *ext_goal_prev;
    const int32_t *ext_goal_t;
    const float *ext_adj;
    int64_t ext_adj_ld;
    const int64_t *ext_adj_off;
    float *ce_acc;
};
extern "C" hipError_t dn_wide_ce_finish(int64_t rows, const float *ce_acc, double *loss_part, hipStream_t st);
extern "C" hipError_t dn_wide_init_state(int N, int K, int L, int R, int G, const float *h0, const float *x0, const int32_t *g0,
                                         float *h, float *pos, int32_t *goal, hipStream_t st);
extern "C" hipError_t dn_wide_goal_head(const WideStep *sp, hipStream_t st);

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail(DAGNET_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));      \
    } while (0)

static inline int pad8(int x) { return (x + 7) & ~7; }
static inline int pad32(int x) { return (x + 31) & ~31; }

struct dagnet_model {
    dagnet_config cfg;
    DnDims dims;
    DnWeights w;
    float *d_wp = nullptr;
    size_t wp_floats = 0;
    int device = 0;
    bool v2 = false;            // the register-resident kernel (rollout_v2.hip) covers this model
    // two v2 regions: [0] fp32 MFMA blocks (the teacher-forced kernels, and the rollout when split products are switched off),
    // [1] split fp16 blocks (the rollout kernels' default)
    V2Weights w2[2];            // absolute offsets into d_wp (read by model 3 only: wide goal head)
    V2Weights *d_w2[2] = {nullptr, nullptr};  // device copies of w2
    int v2_base[2] = {0, 0};    // start of the v2 regions in d_wp: the v2 kernels use compile-time offsets relative to them
    bool split_ok = true;       // every weight fits fp16's range: the split region is usable
    // wide goal head (wide_head.hip) + register-resident kernel model 3 for the rest of the step
    bool wide = false;
    WideWeights ww;
};

struct dagnet_plan {
    int S = 0, N = 0, K = 1, M = 0, AS = 1, n_tiles = 0, device = 0;
    DnLds lds;
    int32_t *d_scene_ptr = nullptr, *d_tile_u0 = nullptr, *d_tile_nu = nullptr, *d_row_scene = nullptr;
    int64_t *d_off_dense = nullptr, *d_off_compact = nullptr;
    double *d_loss_part = nullptr;
    void *d_static = nullptr;   // {DnDims, DnLds, DnWeights} prefix of DnParams for this (model, plan)
    int64_t compact_size = 0;
    const dagnet_model *owner = nullptr;
    // 32-row tiles for the register-resident kernel (only when the model and every scene qualify)
    int32_t *d_tile2_u0 = nullptr, *d_tile2_nu = nullptr;
    int n_tiles2 = 0;
    // big-G plans (rollout_body.h with DN_BIG): per-workgroup global slabs for the goal-width buffers
    bool big = false;
    float *d_gscratch = nullptr;
    int64_t gslab = 0;
    int gblocks = 0;
    // wide goal head: layer-by-layer workspace over all K*N rows (one hipMalloc, carved into WideBuffers)
    bool wide = false;
    float *d_wide = nullptr;
    WideBuffers wb;
    // calls that write the plan's own buffers (loss partials, wide / big-G workspace) must not overlap: the last such call's
    // completion event and stream (plan_enter / plan_leave)
    mutable std::mutex mu;
    mutable hipEvent_t busy = nullptr;
    mutable hipStream_t busy_stream = nullptr;
    mutable bool armed = false;
};

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
namespace {

struct Tensor {
    const float *p;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

typedef std::function<float(int, int)> WFn;   // (output n, input k) -> weight

struct Packer {
    std::vector<float> buf;
    std::map<std::string, Tensor> t;
    std::string missing;

    const Tensor *get(const std::string &name, std::initializer_list<int64_t> shape) {
        auto it = t.find(name);
        if (it == t.end()) {
            if (missing.empty()) missing = name;
            return nullptr;
        }
        if (it->second.shape != std::vector<int64_t>(shape)) {
            if (missing.empty()) {
                missing = name + " (shape mismatch: got [";
                for (auto s : it->second.shape) missing += std::to_string(s) + ",";
                missing += "] want [";
                for (auto s : shape) missing += std::to_string(s) + ",";
                missing += "])";
            }
            return nullptr;
        }
        return &it->second;
    }
    bool has(const std::string &name) const { return t.count(name) != 0; }

    int align() {
        while (buf.size() % 4) buf.push_back(0.f);
        return (int)buf.size();
    }
    // fragment-ordered segment: [nt][kg][lane 64][4]
    int seg(const WFn &W, int N, int K) {
        const int nt = (N + 31) / 32, kg = (K + 7) / 8;
        const int off = align();
        buf.resize(off + (size_t)nt * kg * 256, 0.f);
        for (int a = 0; a < nt; ++a)
            for (int g = 0; g < kg; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int n = a * 32 + (l & 31), k = g * 8 + 2 * i + (l >> 5);
                        buf[off + (((size_t)a * kg + g) * 64 + l) * 4 + i] = (n < N && k < K) ? W(n, k) : 0.f;
                    }
        return off;
    }
    // MFMA-chain order (params_v2.h): [nt][kt][j 0..3][lane 64][i 0..3] = W[nt*32+n][kt*32+8j+4h+i]
    int chain(const WFn &W, int N, int K) {
        const int nt = (N + 31) / 32, kt = (K + 31) / 32;
        const int off = align();
        buf.resize(off + (size_t)nt * kt * 1024, 0.f);
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < kt; ++b)
                for (int j = 0; j < 4; ++j)
                    for (int l = 0; l < 64; ++l)
                        for (int i = 0; i < 4; ++i) {
                            const int n = a * 32 + (l & 31), k = b * 32 + 8 * j + 4 * (l >> 5) + i;
                            buf[off + ((((size_t)a * kt + b) * 4 + j) * 64 + l) * 4 + i] = (n < N && k < K) ? W(n, k) : 0.f;
                        }
        return off;
    }
    // The same blocks for the 16-bit matrix pipe (params_v2.h, v2_common.h mm<..., BF = true>): every weight as two fp16 terms,
    // rounded to nearest (hi = fp16(w), lo = fp16(w - hi): 22 significant bits); a half block (output tile nt, input tile kt,
    // k-half p) = [term][lane 64][8 fp16], slot s of lane (n, h) = W[nt*32+n][kt*32 + 8*(2p + s/4) + 4h + s%4]; 512 dwords.
    // kt_major: half blocks ordered [kt][p][nt] (dense layers: half an input tile is split once for all output tiles),
    // else [nt][kt][p] (one output tile per product).
    bool bf = false;            // v2 region being packed: chain_v2() emits split blocks
    int chain_split(const WFn &W, int N, int K, bool kt_major) {
        const int nt = (N + 31) / 32, kt = (K + 31) / 32;
        const int off = align();
        buf.resize(off + (size_t)nt * kt * V2_BLOCK_SPLIT, 0.f);
        uint16_t *dst = reinterpret_cast<uint16_t *>(buf.data() + off);
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < kt; ++b)
                for (int p = 0; p < 2; ++p) {
                    const size_t half = kt_major ? ((size_t)b * 2 + p) * nt + a : ((size_t)a * kt + b) * 2 + p;
                    for (int l = 0; l < 64; ++l)
                        for (int sl = 0; sl < 8; ++sl) {
                            const int n = a * 32 + (l & 31), k = b * 32 + 8 * (2 * p + sl / 4) + 4 * (l >> 5) + sl % 4;
                            float v = (n < N && k < K) ? W(n, k) : 0.f;
                            for (int t = 0; t < 2; ++t) {
                                const _Float16 hv = (_Float16)v;          // round to nearest even; beyond 65504: +-inf (refused below)
                                uint16_t bits;
                                memcpy(&bits, &hv, 2);
                                if ((bits & 0x7c00u) == 0x7c00u) split_overflow = true;
                                v -= (float)hv;
                                dst[(half * V2_HALF_SPLIT + ((size_t)t * 64 + l) * 4) * 2 + sl] = bits;
                            }
                        }
                }
        return off;
    }
    bool split_overflow = false;      // a weight beyond the fp16 range (or not finite): the model keeps the fp32 kernels
    // the v2 regions' dense segments: fp32 chain order or split fp16 blocks, as the region being packed says
    int chain_v2(const WFn &W, int N, int K, bool kt_major) { return bf ? chain_split(W, N, K, kt_major) : chain(W, N, K); }
    // plain row-major [N][ld] (ld >= K, zero padded): operands of the batched GEMMs (gemm_f32.hip)
    int rowmajor(const WFn &W, int N, int K, int ld) {
        const int off = align();
        buf.resize(off + (size_t)N * ld, 0.f);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) buf[off + (size_t)n * ld + k] = W(n, k);
        return off;
    }
    int vec(const std::function<float(int)> &f, int n, int npad) {
        const int off = align();
        buf.resize(off + npad, 0.f);
        for (int i = 0; i < n; ++i) buf[off + i] = f(i);
        return off;
    }
    int gather(const WFn &W, int N, int G, int Gpad = 0) {      // Gpad: rows allocated (zero beyond G)
        const int ld = pad32(N);
        const int off = align();
        buf.resize(off + (size_t)std::max(G, Gpad) * ld, 0.f);
        for (int g = 0; g < G; ++g)
            for (int n = 0; n < N; ++n) buf[off + (size_t)g * ld + n] = W(n, g);
        return off;
    }
};

struct SegSpec {
    int col0, K;
    int kpad = 0;      // chain packing only: pad the segment with zero columns up to this width
};

// torch nn.Linear "name": weight [N][Ktot], bias [N]
DnLin pack_linear(Packer &pk, const std::string &name, int N, int Ktot, std::vector<SegSpec> segs, int gcol0 = -1, int G = 0) {
    DnLin L;
    memset(&L, 0, sizeof(L));
    L.n = N;
    L.nt = (N + 31) / 32;
    L.bias = L.gather = -1;
    const Tensor *W = pk.get(name + ".weight", {N, Ktot});
    const Tensor *B = pk.get(name + ".bias", {N});
    if (!W || !B) return L;
    const float *w = W->p;
    L.nseg = (int)segs.size();
    for (int s = 0; s < L.nseg; ++s) {
        const int c0 = segs[s].col0;
        L.w[s] = pk.seg([=](int n, int k) { return w[(size_t)n * Ktot + c0 + k]; }, N, segs[s].K);
        L.kg[s] = (segs[s].K + 7) / 8;
    }
    const float *b = B->p;
    L.bias = pk.vec([=](int n) { return b[n]; }, N, L.nt * 32);
    if (gcol0 >= 0) L.gather = pk.gather([=](int n, int g) { return w[(size_t)n * Ktot + gcol0 + g]; }, N, G);
    return L;
}

// two heads stacked: rows [0,Z) from a, [Z,2Z) from b   (mean / logvar)
DnLin pack_pair(Packer &pk, const std::string &a, const std::string &b, int Z, int K) {
    DnLin L;
    memset(&L, 0, sizeof(L));
    L.n = 2 * Z;
    L.nt = (2 * Z + 31) / 32;
    L.bias = L.gather = -1;
    const Tensor *Wa = pk.get(a + ".weight", {Z, K}), *Wb = pk.get(b + ".weight", {Z, K});
    const Tensor *Ba = pk.get(a + ".bias", {Z}), *Bb = pk.get(b + ".bias", {Z});
    if (!Wa || !Wb || !Ba || !Bb) return L;
    const float *wa = Wa->p, *wb = Wb->p, *ba = Ba->p, *bb = Bb->p;
    L.nseg = 1;
    L.w[0] = pk.seg([=](int n, int k) { return n < Z ? wa[(size_t)n * K + k] : wb[(size_t)(n - Z) * K + k]; }, 2 * Z, K);
    L.kg[0] = (K + 7) / 8;
    L.bias = pk.vec([=](int n) { return n < Z ? ba[n] : bb[n - Z]; }, 2 * Z, L.nt * 32);
    return L;
}

DnRaw pack_raw_linear(Packer &pk, const std::string &name, int N, int K) {
    DnRaw r;
    memset(&r, 0, sizeof(r));
    r.n = N;
    r.k = K;
    const Tensor *W = pk.get(name + ".weight", {N, K});
    const Tensor *B = pk.get(name + ".bias", {N});
    if (!W || !B) return r;
    const float *w = W->p, *b = B->p;
    r.w = pk.vec([=](int i) { return w[i]; }, N * K, N * K);
    r.b = pk.vec([=](int i) { return b[i]; }, N, N);
    return r;
}

int pack_bn(Packer &pk, const std::string &name, int n) {
    const Tensor *g = pk.get(name + ".weight", {n}), *b = pk.get(name + ".bias", {n});
    const Tensor *mu = pk.get(name + ".running_mean", {n}), *var = pk.get(name + ".running_var", {n});
    if (!g || !b || !mu || !var) return -1;
    const float *gp = g->p, *bp = b->p, *mp = mu->p, *vp = var->p;
    return pk.vec(
        [=](int i) {
            const int c = i % n;
            switch (i / n) {
                case 0: return gp[c];
                case 1: return bp[c];
                case 2: return mp[c];
                default: return sqrtf(vp[c] + 1e-5f);
            }
        },
        4 * n, 4 * n);
}

DnLin empty_lin() {
    DnLin L;
    memset(&L, 0, sizeof(L));
    L.bias = L.gather = -1;
    return L;
}

DnGraph pack_graph(Packer &pk, const std::string &prefix, int kind, int nin, int nout, int F, int P) {
    DnGraph g;
    memset(&g, 0, sizeof(g));
    g.kind = kind;
    g.nin = nin;
    g.nout = nout;
    g.heads = g.out = g.gc1 = g.gc2 = empty_lin();
    g.b1 = g.b2 = -1;
    if (kind == DAGNET_GRAPH_GAT) {
        std::vector<const float *> Wh(P), ah(P);
        for (int p = 0; p < P; ++p) {
            const Tensor *W = pk.get(prefix + ".attention_" + std::to_string(p) + ".W", {nin, F});
            const Tensor *a = pk.get(prefix + ".attention_" + std::to_string(p) + ".a", {2 * F, 1});
            if (!W || !a) return g;
            Wh[p] = W->p;
            ah[p] = a->p;
        }
        g.heads.n = P * F;
        g.heads.nt = (P * F + 31) / 32;
        g.heads.nseg = 1;
        g.heads.w[0] = pk.seg([=](int n, int k) { return Wh[n / F][(size_t)k * F + (n % F)]; }, P * F, nin);
        g.heads.kg[0] = (nin + 7) / 8;
        g.a_heads = pk.vec([=](int i) { return ah[i / (2 * F)][i % (2 * F)]; }, P * 2 * F, P * 2 * F);
        const Tensor *Wo = pk.get(prefix + ".out_att.W", {P * F, nout});
        const Tensor *ao = pk.get(prefix + ".out_att.a", {2 * nout, 1});
        if (!Wo || !ao) return g;
        const float *wo = Wo->p, *aop = ao->p;
        g.out.n = nout;
        g.out.nt = (nout + 31) / 32;
        g.out.nseg = 1;
        g.out.w[0] = pk.seg([=](int n, int k) { return wo[(size_t)k * nout + n]; }, nout, P * F);
        g.out.kg[0] = (P * F + 7) / 8;
        g.a_out = pk.vec([=](int i) { return aop[i]; }, 2 * nout, 2 * nout);
        g.bn1 = pack_bn(pk, prefix + ".bn1", nout);
    } else if (kind == DAGNET_GRAPH_GCN) {
        const Tensor *W1 = pk.get(prefix + ".gc1.weight", {nin, F});
        const Tensor *W2 = pk.get(prefix + ".gc2.weight", {F, nout});
        if (!W1 || !W2) return g;
        const float *w1 = W1->p, *w2 = W2->p;
        g.gc1.n = F;
        g.gc1.nt = (F + 31) / 32;
        g.gc1.nseg = 1;
        g.gc1.w[0] = pk.seg([=](int n, int k) { return w1[(size_t)k * F + n]; }, F, nin);
        g.gc1.kg[0] = (nin + 7) / 8;
        g.gc2.n = nout;
        g.gc2.nt = (nout + 31) / 32;
        g.gc2.nseg = 1;
        g.gc2.w[0] = pk.seg([=](int n, int k) { return w2[(size_t)k * nout + n]; }, nout, F);
        g.gc2.kg[0] = (F + 7) / 8;
        if (pk.has(prefix + ".gc1.bias")) {
            const Tensor *b = pk.get(prefix + ".gc1.bias", {F});
            if (b) { const float *bp = b->p; g.b1 = pk.vec([=](int i) { return bp[i]; }, F, F); }
        }
        if (pk.has(prefix + ".gc2.bias")) {
            const Tensor *b = pk.get(prefix + ".gc2.bias", {nout});
            if (b) { const float *bp = b->p; g.b2 = pk.vec([=](int i) { return bp[i]; }, nout, nout); }
        }
        g.bn1 = pack_bn(pk, prefix + ".bn1", F);
        g.bn2 = pack_bn(pk, prefix + ".bn2", nout);
    }
    return g;
}

DnGru pack_gru(Packer &pk, int l, int R, int H) {
    DnGru g;
    g.r = g.z = g.ni = g.nh = empty_lin();
    const int Kin = l == 0 ? 2 * H : R;
    const std::string sl = std::to_string(l);
    const Tensor *Wi = pk.get("rnn.weight_ih_l" + sl, {3 * R, Kin});
    const Tensor *Wh = pk.get("rnn.weight_hh_l" + sl, {3 * R, R});
    const Tensor *Bi = pk.get("rnn.bias_ih_l" + sl, {3 * R});
    const Tensor *Bh = pk.get("rnn.bias_hh_l" + sl, {3 * R});
    if (!Wi || !Wh || !Bi || !Bh) return g;
    const float *wi = Wi->p, *wh = Wh->p, *bi = Bi->p, *bh = Bh->p;
    std::vector<SegSpec> xs;
    if (l == 0) xs = {{0, H}, {H, H}};
    else xs = {{0, R}};
    auto mk = [&](DnLin &L, int gate, bool with_x, bool with_h) {
        L.n = R;
        L.nt = (R + 31) / 32;
        L.nseg = 0;
        if (with_x)
            for (auto s : xs) {
                const int c0 = s.col0;
                L.w[L.nseg] = pk.seg([=](int n, int k) { return wi[(size_t)(gate * R + n) * Kin + c0 + k]; }, R, s.K);
                L.kg[L.nseg] = (s.K + 7) / 8;
                ++L.nseg;
            }
        if (with_h) {
            L.w[L.nseg] = pk.seg([=](int n, int k) { return wh[(size_t)(gate * R + n) * R + k]; }, R, R);
            L.kg[L.nseg] = (R + 7) / 8;
            ++L.nseg;
        }
        L.bias = pk.vec(
            [=](int n) { return (with_x ? bi[gate * R + n] : 0.f) + (with_h ? bh[gate * R + n] : 0.f); }, R, L.nt * 32);
    };
    mk(g.r, 0, true, true);
    mk(g.z, 1, true, true);
    mk(g.ni, 2, true, false);
    mk(g.nh, 2, false, true);
    return g;
}

// ---- register-resident kernel (rollout_v2.hip): same tensors, MFMA-chain packing -------------
// Every function packs in two parts, in the order of params_v2.h v2_layout(): part 0 = one weight segment (they are laid out
// in the order a rollout step consumes them), part 1 = the layer's vectors (bias, one-hot table, attention vectors).
// Npad / Gpad / SegSpec.kpad: sizes the segment is padded to (zeros), so that the layout of an instantiation does not
// depend on g_dim or n_max_agents (params_v2.h v2_layout)
// logits: the layer feeds a softmax over its N outputs and nothing else - weights and bias are scaled by log2(e) (the
// kernels' softmax is exp2(x - max)) and the bias of the padded outputs is -inf, so they come out as exactly 0 without a mask.
struct V2LinSpec {
    std::string name;
    int N, Ktot;
    std::vector<SegSpec> segs;
    int gcol0 = -1, G = 0, Npad = 0, Gpad = 0;
    bool logits = false;
};
void v2_lin_init(V2Lin &L) {
    L.w[0] = L.w[1] = L.w[2] = 0;
    L.bias = L.gather = -1;
}
// weight segment s of the layer
void v2_lin_w(Packer &pk, V2Lin &L, const V2LinSpec &sp, int s) {
    const Tensor *W = pk.get(sp.name + ".weight", {sp.N, sp.Ktot});
    if (!W) return;
    const float *w = W->p;
    const int N = sp.N, Ktot = sp.Ktot, Np = std::max(sp.N, sp.Npad);
    const float sc = sp.logits ? 1.44269504088896340736f : 1.0f;
    const int c0 = sp.segs[s].col0, Ks = sp.segs[s].K;
    L.w[s] = pk.chain_v2([=](int n, int k) { return (n < N && k < Ks) ? sc * w[(size_t)n * Ktot + c0 + k] : 0.f; }, Np, std::max(Ks, sp.segs[s].kpad), true);
}
void v2_lin_v(Packer &pk, V2Lin &L, const V2LinSpec &sp) {
    const Tensor *W = pk.get(sp.name + ".weight", {sp.N, sp.Ktot});
    const Tensor *B = pk.get(sp.name + ".bias", {sp.N});
    if (!W || !B) return;
    const float *w = W->p, *b = B->p;
    const int N = sp.N, Ktot = sp.Ktot, Np = std::max(sp.N, sp.Npad), gcol0 = sp.gcol0;
    const float sc = sp.logits ? 1.44269504088896340736f : 1.0f;
    if (sp.logits) L.bias = pk.vec([=](int n) { return n < N ? sc * b[n] : -INFINITY; }, pad32(Np), pad32(Np));
    else L.bias = pk.vec([=](int n) { return b[n]; }, N, pad32(Np));
    // one-hot input: row g of the table = column gcol0 + g of the weight PLUS the bias (v2_common.h start<> loads one row)
    if (gcol0 >= 0) L.gather = pk.gather([=](int n, int g) { return w[(size_t)n * Ktot + gcol0 + g] + b[n]; }, N, sp.G, sp.Gpad);
}

// two heads stacked: rows [0,Z) from a, [Z,2Z) from b   (mean / logvar)
void v2_pair(Packer &pk, V2Lin &L, const std::string &a, const std::string &b, int Z, int K, int part) {
    const Tensor *Wa = pk.get(a + ".weight", {Z, K}), *Wb = pk.get(b + ".weight", {Z, K});
    const Tensor *Ba = pk.get(a + ".bias", {Z}), *Bb = pk.get(b + ".bias", {Z});
    if (!Wa || !Wb || !Ba || !Bb) return;
    const float *wa = Wa->p, *wb = Wb->p, *ba = Ba->p, *bb = Bb->p;
    if (part == 0) L.w[0] = pk.chain_v2([=](int n, int k) { return n < Z ? wa[(size_t)n * K + k] : wb[(size_t)(n - Z) * K + k]; }, 2 * Z, K, true);
    else L.bias = pk.vec([=](int n) { return n < Z ? ba[n] : bb[n - Z]; }, 2 * Z, pad32(2 * Z));
}

// pad: nin and nout are padded to this many features (the goal graph: V2_GT tiles whatever g_dim is)
static const double kLog2e = 1.4426950408889634074;
static const double kTanhScale = 2.0 * kLog2e, kSigmScale = -kLog2e;   // v2_common.h tanh2() / sigm2()
// part 0: heads, 1: out (both weight segments), 2: vectors
void v2_gat(Packer &pk, V2Gat &g, const std::string &prefix, int nin, int nout, int F, int P, int pad, int part) {
    g.nout = nout;
    std::vector<const float *> Wh(P), ah(P);
    for (int p = 0; p < P; ++p) {
        const Tensor *W = pk.get(prefix + ".attention_" + std::to_string(p) + ".W", {nin, F});
        const Tensor *a = pk.get(prefix + ".attention_" + std::to_string(p) + ".a", {2 * F, 1});
        if (!W || !a) return;
        Wh[p] = W->p;
        ah[p] = a->p;
    }
    const Tensor *Wo = pk.get(prefix + ".out_att.W", {P * F, nout});
    const Tensor *ao = pk.get(prefix + ".out_att.a", {2 * nout, 1});
    if (!Wo || !ao) return;
    const float *wo = Wo->p, *aop = ao->p;
    const int ninp = std::max(nin, pad), noutp = std::max(nout, pad);
    const Tensor *ga = pk.get(prefix + ".bn1.weight", {nout}), *be = pk.get(prefix + ".bn1.bias", {nout});
    const Tensor *mu = pk.get(prefix + ".bn1.running_mean", {nout}), *var = pk.get(prefix + ".bn1.running_var", {nout});
    if (!ga || !be || !mu || !var) return;
    const float *gp = ga->p, *bp = be->p, *mp = mu->p, *vp = var->p;
    if (part == 0) {
        g.heads = pk.chain_v2([=](int n, int k) { return k < nin ? Wh[n / F][(size_t)k * F + (n % F)] : 0.f; }, P * F, ninp, false);
        return;
    }
    if (part == 1) {
        // eval-mode BatchNorm y = x * scale + shift and the factor 2 log2(e) of the kernels' tanh2() (v2_common.h): the scale
        // goes into the out_att weights (the attention sums over rows, the scale is per feature), the shift into g.bn
        g.out = pk.chain_v2(
            [=](int n, int k) { return n < nout ? (float)(kTanhScale * (gp[n] / sqrtf(vp[n] + 1e-5f)) * wo[(size_t)k * nout + n]) : 0.f; },
            noutp, P * F, false);
        return;
    }
    // attention vectors of the heads for the per-lane logit reduction (v2_common.h gat<>): entry 8p+f = a_p[f] (source half)
    // and 32 + 8p+f = a_p[F+f] (destination half), in the log2 domain: the kernels' softmax is exp2(e - max)
    g.fold = pk.vec([=](int i) { const int q = i & 31; return (float)((double)ah[q / F][(i >> 5) * F + (q % F)] * kLog2e); }, 64, 64);
    // a_out . (x2 W_out) == x2 . (W_out a_out): the out head's two logit dot products over its 32 input features
    g.fold_out = pk.vec(
        [=](int i) {
            const int q = i >> 5, k = i & 31;
            double acc = 0.0;
            for (int n = 0; n < nout; ++n) acc += (double)wo[(size_t)k * nout + n] * (double)aop[q * nout + n];
            return (float)(acc * kLog2e);
        },
        64, 64);
    const int ld = pad32(noutp);
    g.bn = pk.vec(
        [=](int c) {
            if (c >= nout) return 0.0f;
            const float sc = gp[c] / sqrtf(vp[c] + 1e-5f);
            return (float)(kTanhScale * (bp[c] - mp[c] * sc));
        },
        ld, 2 * ld);        // second row unused (params_v2.h keeps one table size for GAT and GCN)
}

// GCN in the V2Gat slots (v2_common.h gcn<>): heads = W1, fold = layer-1 scale|shift, out = W2, bn = layer-2 scale|shift;
// bias and eval BatchNorm folded: y = agg * sc + ((b - mean) * sc + beta), sc = gamma / sqrt(var + eps)
void v2_gcn(Packer &pk, V2Gat &g, const std::string &prefix, int nin, int nout, int F, int pad, int part) {
    g.nout = nout;
    const Tensor *W1 = pk.get(prefix + ".gc1.weight", {nin, F}), *W2 = pk.get(prefix + ".gc2.weight", {F, nout});
    if (!W1 || !W2) return;
    const float *w1 = W1->p, *w2 = W2->p;
    const float *b1 = nullptr, *b2 = nullptr;
    if (pk.has(prefix + ".gc1.bias")) { const Tensor *b = pk.get(prefix + ".gc1.bias", {F}); if (b) b1 = b->p; }
    if (pk.has(prefix + ".gc2.bias")) { const Tensor *b = pk.get(prefix + ".gc2.bias", {nout}); if (b) b2 = b->p; }
    const int ninp = std::max(nin, pad), noutp = std::max(nout, pad);
    if (part == 0) {
        g.heads = pk.chain_v2([=](int n, int k) { return k < nin ? w1[(size_t)k * F + n] : 0.f; }, F, ninp, false);
        return;
    }
    if (part == 1) {
        g.out = pk.chain_v2([=](int n, int k) { return n < nout ? w2[(size_t)k * nout + n] : 0.f; }, noutp, F, false);
        return;
    }
    auto fold = [&](const std::string &bn, int n, const float *bias, int ld, double act) {
        const Tensor *ga = pk.get(bn + ".weight", {n}), *be = pk.get(bn + ".bias", {n});
        const Tensor *mu = pk.get(bn + ".running_mean", {n}), *var = pk.get(bn + ".running_var", {n});
        if (!ga || !be || !mu || !var) return -1;
        const float *gp = ga->p, *bp = be->p, *mp = mu->p, *vp = var->p;
        return pk.vec(
            [=](int i) {
                const int c = i % ld, row = i / ld;
                if (c >= n) return 0.0f;
                const float sc = gp[c] / sqrtf(vp[c] + 1e-5f);
                return (float)(act * (row == 0 ? sc : ((bias ? bias[c] : 0.f) - mp[c]) * sc + bp[c]));
            },
            2 * ld, 2 * ld);
    };
    g.fold = fold(prefix + ".bn1", F, b1, 32, 1.0);
    g.bn = fold(prefix + ".bn2", nout, b2, pad32(noutp), kTanhScale);       // layer 2 ends in tanh2()
}

// part 0: the six weight segments in the order the kernels' GRU consumes them (nh, r.x, r.h, ni, z.x, z.h); part 1: biases.
// fp32 blocks are [nt][kt] (one output tile per product); split fp16 ones [kt][p][nt] (both output tiles per product:
// v2_common.h gru<>, BF).
void v2_gru(Packer &pk, V2Gru &g, int l, int R, int H, int part) {
    const int Kin = l == 0 ? 2 * H : R;
    const std::string sl = std::to_string(l);
    const Tensor *Wi = pk.get("rnn.weight_ih_l" + sl, {3 * R, Kin});
    const Tensor *Wh = pk.get("rnn.weight_hh_l" + sl, {3 * R, R});
    const Tensor *Bi = pk.get("rnn.bias_ih_l" + sl, {3 * R});
    const Tensor *Bh = pk.get("rnn.bias_hh_l" + sl, {3 * R});
    if (!Wi || !Wh || !Bi || !Bh) return;
    const float *wi = Wi->p, *wh = Wh->p, *bi = Bi->p, *bh = Bh->p;
    // gate pre-activations come out already scaled for sigm2() / tanh2() (v2_common.h): r, z by -log2(e); both halves of
    // the candidate (W_in x + b_in and W_hn h + b_hn, the latter multiplied by r in between) by 2 log2(e)
    if (pk.bf && part == 0) {
        // split-product kernels (v2_common.h gru<>, BF): two dense layers with 3 R outputs each - W_ih with its gates in the order
        // n, r, z and W_hh in the order r, z, n - so that an input tile is split once for all gates
        const int gx[3] = {2, 0, 1}, gh[3] = {0, 1, 2};
        auto scale = [](int gate) { return gate == 2 ? kTanhScale : kSigmScale; };
        g.r.w[0] = pk.chain_v2([=](int n, int k) { const int gate = gx[n / R]; return (float)(scale(gate) * wi[(size_t)(gate * R + n % R) * Kin + k]); },
                               3 * R, Kin, true);
        g.r.w[1] = pk.chain_v2([=](int n, int k) { const int gate = gh[n / R]; return (float)(scale(gate) * wh[(size_t)(gate * R + n % R) * R + k]); },
                               3 * R, R, true);
        return;
    }
    auto mk = [&](V2Lin &L, int gate, bool with_x, bool with_h) {
        int s = 0;
        const double a = gate == 2 ? kTanhScale : kSigmScale;
        if (part == 0) {
            if (with_x) L.w[s++] = pk.chain_v2([=](int n, int k) { return (float)(a * wi[(size_t)(gate * R + n) * Kin + k]); }, R, Kin, true);
            if (with_h) L.w[s++] = pk.chain_v2([=](int n, int k) { return (float)(a * wh[(size_t)(gate * R + n) * R + k]); }, R, R, true);
        } else {
            L.bias = pk.vec([=](int n) { return (float)(a * ((with_x ? bi[gate * R + n] : 0.f) + (with_h ? bh[gate * R + n] : 0.f))); }, R, pad32(R));
        }
    };
    mk(g.nh, 2, false, true);
    mk(g.r, 0, true, true);
    mk(g.ni, 2, true, false);
    mk(g.z, 1, true, true);
}

bool v2_eligible(const dagnet_config &c) {
    if (c.model < 0 || c.model > 2) return false;
    if (c.x_dim != 2 || c.h_dim != 64 || c.rnn_dim != 64) return false;
    if (!dn_v2_has(c.model, c.n_layers, c.z_dim, c.model == DAGNET_MODEL_VRNN ? 0 : c.graph_model)) return false;
    if (c.model != DAGNET_MODEL_VRNN) {
        if (c.graph_model == DAGNET_GRAPH_GAT && (c.n_heads != 4 || c.graph_hid != 8)) return false;
        if (c.graph_model == DAGNET_GRAPH_GCN && c.graph_hid > 32) return false;
    }
    if (c.model == DAGNET_MODEL_DAGNET && (c.g_dim > 96 || c.n_max_agents > 32)) return false;
    if (c.model == DAGNET_MODEL_DAGNET && c.n_max_agents > 16 && !dn_v2_has(4, c.n_layers, c.z_dim, c.graph_model)) return false;
    return true;
}

static inline int pad4(int x) { return (x + 3) & ~3; }

// DAG-Net with a goal head too wide for one wave's registers (g_dim > 96): wide_head.hip + rollout_v2 model 3
bool wide_eligible(const dagnet_config &c) {
    if (c.model != DAGNET_MODEL_DAGNET) return false;
    if (c.x_dim != 2 || c.h_dim != 64 || c.rnn_dim != 64) return false;
    if (!dn_v2_has(3, c.n_layers, c.z_dim, DAGNET_GRAPH_GAT)) return false;
    if (c.graph_model != DAGNET_GRAPH_GAT || c.n_heads != 4 || c.graph_hid != 8) return false;
    if (c.g_dim <= 96 || (c.g_dim & 3) || c.n_max_agents > 32) return false;
    return true;
}

void wide_pack(Packer &pk, const dagnet_config &c, WideWeights &W) {
    memset(&W, 0, sizeof(W));
    const int H = c.h_dim, R = c.rnn_dim, G = c.g_dim, Dd = 2 * c.n_max_agents, F = c.graph_hid, P = c.n_heads, PF = P * F;
    const Tensor *W0 = pk.get("dec_goal.0.weight", {H, Dd + R + G}), *B0 = pk.get("dec_goal.0.bias", {H});
    const Tensor *W2 = pk.get("dec_goal.2.weight", {G, H}), *B2 = pk.get("dec_goal.2.bias", {G});
    const Tensor *Wl = pk.get("lg_goals.0.weight", {G, 2 * G}), *Bl = pk.get("lg_goals.0.bias", {G});
    const Tensor *Wo = pk.get("graph_goals.out_att.W", {PF, G}), *ao = pk.get("graph_goals.out_att.a", {2 * G, 1});
    if (!W0 || !B0 || !W2 || !B2 || !Wl || !Bl || !Wo || !ao) return;
    std::vector<const float *> Wh(P), ah(P);
    for (int p = 0; p < P; ++p) {
        const Tensor *Wp_ = pk.get("graph_goals.attention_" + std::to_string(p) + ".W", {G, F});
        const Tensor *ap_ = pk.get("graph_goals.attention_" + std::to_string(p) + ".a", {2 * F, 1});
        if (!Wp_ || !ap_) return;
        Wh[p] = Wp_->p;
        ah[p] = ap_->p;
    }
    const int Ktot = Dd + R + G;
    const float *w0 = W0->p, *b0 = B0->p, *w2 = W2->p, *b2 = B2->p, *wl = Wl->p, *bl = Bl->p, *wo = Wo->p, *aop = ao->p;
    W.g0_ld = pad4(Dd + R);
    W.g0_w = pk.rowmajor([=](int n, int k) { return w0[(size_t)n * Ktot + k]; }, H, Dd + R, W.g0_ld);
    W.g0_b = pk.vec([=](int n) { return b0[n]; }, H, H);
    W.g0_gld = H;
    W.g0_gather = pk.rowmajor([=](int g, int n) { return w0[(size_t)n * Ktot + Dd + R + g]; }, G, H, H);
    W.g2_ld = pad4(H);
    W.g2_w = pk.rowmajor([=](int n, int k) { return w2[(size_t)n * H + k]; }, G, H, W.g2_ld);
    W.g2_b = pk.vec([=](int n) { return b2[n]; }, G, G);
    // heads: row p*F+f = W_p[:, f]; folded logits: row PF+p = W_p a_p[:F], row PF+P+p = W_p a_p[F:]  (x.(W a) == (x W).a)
    W.hf_ld = pad4(G);
    W.hf_w = pk.rowmajor(
        [=](int n, int k) {
            if (n < PF) return Wh[n / F][(size_t)k * F + (n % F)];
            const int p = (n - PF) % P, dst = (n - PF) / P;
            double acc = 0.0;
            for (int f = 0; f < F; ++f) acc += (double)Wh[p][(size_t)k * F + f] * (double)ah[p][dst * F + f];
            return (float)acc;
        },
        PF + 2 * P, G, W.hf_ld);
    const Tensor *g = pk.get("graph_goals.bn1.weight", {G}), *b = pk.get("graph_goals.bn1.bias", {G});
    const Tensor *mu = pk.get("graph_goals.bn1.running_mean", {G}), *var = pk.get("graph_goals.bn1.running_var", {G});
    if (!g || !b || !mu || !var) return;
    const float *gp = g->p, *bp = b->p, *mp = mu->p, *vp = var->p;
    // out layer: the attention is applied to its 32-wide input first (wide_head.hip attn_mix_kernel), so eval BatchNorm folds
    // into the dense layer: rows of Wout^T scaled by gamma / sqrt(var + eps), shift as the bias, tanh in the GEMM epilogue
    W.out_ld = pad4(PF);
    W.out_w = pk.rowmajor([=](int n, int k) { return wo[(size_t)k * G + n] * (gp[n] / sqrtf(vp[n] + 1e-5f)); }, G, PF, W.out_ld);
    W.fold_out = pk.vec(
        [=](int i) {
            const int dst = i / PF, k = i % PF;
            double acc = 0.0;
            for (int n = 0; n < G; ++n) acc += (double)wo[(size_t)k * G + n] * (double)aop[dst * G + n];
            return (float)acc;
        },
        2 * PF, 2 * PF);
    W.bn = pk.vec([=](int i) { return bp[i] - mp[i] * (gp[i] / sqrtf(vp[i] + 1e-5f)); }, G, G);     // shift only
    W.lg_ld = 2 * pad4(G);
    const int Gp = pad4(G);
    W.lg_w = pk.rowmajor([=](int n, int k) { return k < Gp ? (k < G ? wl[(size_t)n * 2 * G + k] : 0.f)
                                                            : (k - Gp < G ? wl[(size_t)n * 2 * G + G + (k - Gp)] : 0.f); },
                         G, 2 * Gp, W.lg_ld);
    W.lg_b = pk.vec([=](int n) { return bl[n]; }, G, G);
}

void v2_pack(Packer &pk, const dagnet_config &c, V2Weights &W, bool ext = false) {
    memset(&W, 0, sizeof(W));
    const int H = 64, R = 64, Z = c.z_dim, G = c.g_dim, Dd = 2 * c.n_max_agents;
    const bool dag = c.model == DAGNET_MODEL_DAGNET, graph = c.model != DAGNET_MODEL_VRNN;
    const bool head = dag && !ext, gcn = c.graph_model == DAGNET_GRAPH_GCN;
    const int GP = V2_GT * 32;                       // goal-wide dims are padded to V2_GT tiles
    const int DP = (ext || Dd > 32) ? 64 : 32;       // disposition tiles (model 3 multiplies two whatever n_max_agents is)
    const std::string hid = dag ? "graph_hiddens" : "graph";
    V2LinSpec dec_goal0{"dec_goal.0", H, Dd + R + G, {{0, Dd, DP}, {Dd, R}}, Dd + R, G, 0, GP};
    V2LinSpec dec_goal2{"dec_goal.2", G, H, {{0, H}}, -1, 0, GP, 0, true};
    V2LinSpec lg_goals{"lg_goals.0", G, 2 * G, {{0, G, GP}, {G, G, GP}}, -1, 0, GP, 0, true};
    V2LinSpec prior0 = dag ? V2LinSpec{"prior.0", H, G + R, {{G, R}}, 0, G, 0, GP} : V2LinSpec{"prior.0", H, R, {{0, R}}};
    V2LinSpec prior2{"prior.2", H, H, {{0, H}}};
    V2LinSpec phi_z0{"phi_z.0", H, Z, {{0, Z}}}, phi_z2{"phi_z.2", H, H, {{0, H}}};
    V2LinSpec dec0 = dag ? V2LinSpec{"dec.0", H, Dd + G + H + R, {{0, Dd, DP}, {Dd + G, H}, {Dd + G + H, R}}, Dd, G, 0, GP}
                         : V2LinSpec{"dec.0", H, H + R, {{0, H}, {H, R}}};
    V2LinSpec dec2{"dec.2", H, H, {{0, H}}}, phi_x2{"phi_x.2", H, H, {{0, H}}};
    V2LinSpec blend{dag ? "lg_hiddens" : "lg", R, 2 * R, {{0, R}, {R, R}}};
    V2LinSpec enc0 = dag ? V2LinSpec{"enc.0", H, H + G + R, {{0, H}, {H + G, R}}, H, G, 0, GP} : V2LinSpec{"enc.0", H, H + R, {{0, H}, {H, R}}};
    V2LinSpec enc2{"enc.2", H, H, {{0, H}}};
    for (V2Lin *L : {&W.phi_x2, &W.phi_z0, &W.phi_z2, &W.prior0, &W.prior2, &W.prior_ml, &W.dec0, &W.dec2, &W.blend, &W.dec_goal0, &W.dec_goal2,
                     &W.lg_goals, &W.enc0, &W.enc2, &W.enc_ml})
        v2_lin_init(*L);
    for (int l = 0; l < 2; ++l)
        for (V2Lin *L : {&W.gru[l].r, &W.gru[l].z, &W.gru[l].ni, &W.gru[l].nh}) v2_lin_init(*L);
    auto graphnet = [&](V2Gat &g, const std::string &prefix, int nin, int nout, int pad, int part) {
        if (gcn) v2_gcn(pk, g, prefix, nin, nout, c.graph_hid, pad, part);
        else v2_gat(pk, g, prefix, nin, nout, c.graph_hid, c.n_heads, pad, part);
    };
    // ---- the weight stream of one rollout step, in the order it is consumed (params_v2.h v2_layout)
    if (head) {
        v2_lin_w(pk, W.dec_goal0, dec_goal0, 0);
        v2_lin_w(pk, W.dec_goal0, dec_goal0, 1);
        v2_lin_w(pk, W.dec_goal2, dec_goal2, 0);
        v2_lin_w(pk, W.lg_goals, lg_goals, 0);
        graphnet(W.g_goal, "graph_goals", G, G, GP, 0);
        graphnet(W.g_goal, "graph_goals", G, G, GP, 1);
        v2_lin_w(pk, W.lg_goals, lg_goals, 1);
    }
    v2_lin_w(pk, W.prior0, prior0, 0);
    v2_lin_w(pk, W.prior2, prior2, 0);
    v2_pair(pk, W.prior_ml, "prior_mean", "prior_logvar", Z, H, 0);
    v2_lin_w(pk, W.phi_z0, phi_z0, 0);
    v2_lin_w(pk, W.phi_z2, phi_z2, 0);
    for (size_t s = 0; s < dec0.segs.size(); ++s) v2_lin_w(pk, W.dec0, dec0, (int)s);
    v2_lin_w(pk, W.dec2, dec2, 0);
    v2_lin_w(pk, W.phi_x2, phi_x2, 0);
    for (int l = 0; l < c.n_layers; ++l) v2_gru(pk, W.gru[l], l, R, H, 0);
    if (graph) {
        graphnet(W.g_hid, hid, R, R, 0, 0);
        graphnet(W.g_hid, hid, R, R, 0, 1);
        v2_lin_w(pk, W.blend, blend, 0);
        v2_lin_w(pk, W.blend, blend, 1);
    }
    // ---- teacher-forced pass only (forward_v2.hip)
    v2_lin_w(pk, W.enc0, enc0, 0);
    v2_lin_w(pk, W.enc0, enc0, 1);
    v2_lin_w(pk, W.enc2, enc2, 0);
    v2_pair(pk, W.enc_ml, "enc_mean", "enc_logvar", Z, H, 0);
    // ---- vectors
    if (head) {
        v2_lin_v(pk, W.dec_goal0, dec_goal0);
        v2_lin_v(pk, W.dec_goal2, dec_goal2);
        v2_lin_v(pk, W.lg_goals, lg_goals);
        graphnet(W.g_goal, "graph_goals", G, G, GP, 2);
    }
    v2_lin_v(pk, W.prior0, prior0);
    v2_lin_v(pk, W.prior2, prior2);
    v2_pair(pk, W.prior_ml, "prior_mean", "prior_logvar", Z, H, 1);
    v2_lin_v(pk, W.phi_z0, phi_z0);
    v2_lin_v(pk, W.phi_z2, phi_z2);
    v2_lin_v(pk, W.dec0, dec0);
    v2_lin_v(pk, W.dec2, dec2);
    v2_lin_v(pk, W.phi_x2, phi_x2);
    for (int l = 0; l < c.n_layers; ++l) v2_gru(pk, W.gru[l], l, R, H, 1);
    if (graph) {
        graphnet(W.g_hid, hid, R, R, 0, 2);
        v2_lin_v(pk, W.blend, blend);
    }
    v2_lin_v(pk, W.enc0, enc0);
    v2_lin_v(pk, W.enc2, enc2);
    v2_pair(pk, W.enc_ml, "enc_mean", "enc_logvar", Z, H, 1);
    DnRaw px0 = pack_raw_linear(pk, "phi_x.0", H, 2), dm = pack_raw_linear(pk, "dec_mean", 2, H), dl = pack_raw_linear(pk, "dec_logvar", 2, H);
    W.phi_x0_w = px0.w; W.phi_x0_b = px0.b;
    W.dec_mean_w = dm.w; W.dec_mean_b = dm.b;
    W.dec_logvar_w = dl.w; W.dec_logvar_b = dl.b;
}


// The packer's table against the compile-time layout of the instantiation that will run (params_v2.h v2_layout):
// every offset the layout marks as used must equal base + the layout's relative offset.  Returns a description of the
// first difference, or an empty string.
std::string v2_check_layout(const V2Weights &W, int base, int total, int model_code, int nl, int zd, int gm, int bf) {
    const V2Layout L = v2_layout(model_code, nl, zd, gm, bf);
    const int32_t *a = reinterpret_cast<const int32_t *>(&W), *o = reinterpret_cast<const int32_t *>(&L.off),
                  *u = reinterpret_cast<const int32_t *>(&L.used);
    for (size_t i = 0; i < sizeof(V2Weights) / sizeof(int32_t); ++i)
        if (u[i] && a[i] != base + o[i])
            return "v2 layout: table entry " + std::to_string(i) + " is at " + std::to_string(a[i] - base) + ", layout says " +
                   std::to_string(o[i]);
    if (total != L.total) return "v2 layout: region holds " + std::to_string(total) + " floats, layout says " + std::to_string(L.total);
    return "";
}

}   // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" const char *dagnet_last_error(void) { return g_err; }
extern "C" const char *dagnet_version(void) { return "dagnet_hip 0.3 (gfx950, split fp16x2 / bf16x3 32x32x16 + fp32 32x32x2 MFMA)"; }

extern "C" int dagnet_model_create(const dagnet_config *cfg, const dagnet_tensor *tensors, int32_t n_tensors, dagnet_model **out) {
    if (!cfg || !out || (!tensors && n_tensors > 0)) return fail(DAGNET_ERR_INVALID, "null argument");
    *out = nullptr;
    const dagnet_config &c = *cfg;
    if (c.model < 0 || c.model > 3) return fail(DAGNET_ERR_INVALID, "unknown model %d", c.model);
    if (c.x_dim < 1 || c.h_dim < 1 || c.z_dim < 1 || c.rnn_dim < 1) return fail(DAGNET_ERR_INVALID, "dims must be positive");
    if (c.x_dim != 2 && c.model != DAGNET_MODEL_VRNN && c.model != DAGNET_MODEL_GRAPH)
        return fail(DAGNET_ERR_INVALID, "graph models need x_dim == 2 (positions are xy), got %d", c.x_dim);
    if (c.n_layers < 1 || c.n_layers > DN_MAX_LAYERS) return fail(DAGNET_ERR_INVALID, "n_layers must be 1..%d", DN_MAX_LAYERS);
    const bool graph = c.model != DAGNET_MODEL_VRNN;
    if (graph) {
        if (c.graph_model != DAGNET_GRAPH_GAT && c.graph_model != DAGNET_GRAPH_GCN)
            return fail(DAGNET_ERR_INVALID, "graph_model must be gat or gcn");
        if (c.graph_hid < 1 || c.graph_hid > DN_FCAP) return fail(DAGNET_ERR_INVALID, "graph_hid must be 1..%d", DN_FCAP);
        if (c.graph_model == DAGNET_GRAPH_GAT && c.n_heads < 1) return fail(DAGNET_ERR_INVALID, "n_heads must be >= 1");
        if (c.adjacency_type < 0 || c.adjacency_type > 2) return fail(DAGNET_ERR_INVALID, "adjacency_type must be 0,1,2");
        if (c.adjacency_type == 2 && c.top_k_neigh < 1)
            return fail(DAGNET_ERR_INVALID, "Using KNN-similarity but top_k_neigh is not specified");
    }
    if (c.model == DAGNET_MODEL_DAGNET && (c.g_dim < 1 || c.n_max_agents < 1))
        return fail(DAGNET_ERR_INVALID, "DAG-Net needs g_dim and n_max_agents");

    Packer pk;
    for (int i = 0; i < n_tensors; ++i) {
        Tensor t;
        t.p = tensors[i].h_data;
        for (int d = 0; d < tensors[i].ndim; ++d) t.shape.push_back(tensors[i].shape[d]);
        if (!tensors[i].name || !t.p) return fail(DAGNET_ERR_INVALID, "tensor %d has a null name or data", i);
        pk.t[tensors[i].name] = t;
    }
    dagnet_model *m = new dagnet_model();
    m->cfg = c;
    DnDims &D = m->dims;
    memset(&D, 0, sizeof(D));
    D.model = c.model;
    D.graph = graph ? c.graph_model : 0;
    D.adj_type = c.adjacency_type;
    D.knn_k = c.top_k_neigh;
    D.X = c.x_dim; D.H = c.h_dim; D.Z = c.z_dim; D.R = c.rnn_dim; D.L = c.n_layers;
    D.G = c.model == DAGNET_MODEL_DAGNET ? c.g_dim : 0;
    D.D = c.model == DAGNET_MODEL_DAGNET ? 2 * c.n_max_agents : 0;
    D.F = graph ? c.graph_hid : 0;
    D.P = graph && c.graph_model == DAGNET_GRAPH_GAT ? c.n_heads : 0;
    D.sigma = c.sigma;
    D.alpha = c.alpha;
    const int X = D.X, H = D.H, Z = D.Z, R = D.R, G = D.G, Dd = D.D;
    const bool dag = c.model == DAGNET_MODEL_DAGNET;

    DnWeights &W = m->w;
    memset(&W, 0, sizeof(W));
    if (c.model == DAGNET_MODEL_GRAPH) {
        // stand-alone GAT / GCN module: rnn_dim = nin, h_dim = nout, state-dict prefix "graph"
        W.g_hid = pack_graph(pk, "graph", D.graph, R, H, D.F, D.P);
    } else {
    W.phi_x0 = pack_raw_linear(pk, "phi_x.0", H, X);
    W.phi_x2 = pack_linear(pk, "phi_x.2", H, H, {{0, H}});
    W.phi_z0 = pack_linear(pk, "phi_z.0", H, Z, {{0, Z}});
    W.phi_z2 = pack_linear(pk, "phi_z.2", H, H, {{0, H}});
    if (dag) {
        W.enc0 = pack_linear(pk, "enc.0", H, H + G + R, {{0, H}, {H + G, R}}, H, G);
        W.prior0 = pack_linear(pk, "prior.0", H, G + R, {{G, R}}, 0, G);
        W.dec0 = pack_linear(pk, "dec.0", H, Dd + G + H + R, {{0, Dd}, {Dd + G, H}, {Dd + G + H, R}}, Dd, G);
    } else {
        W.enc0 = pack_linear(pk, "enc.0", H, H + R, {{0, H}, {H, R}});
        W.prior0 = pack_linear(pk, "prior.0", H, R, {{0, R}});
        W.dec0 = pack_linear(pk, "dec.0", H, H + R, {{0, H}, {H, R}});
    }
    W.enc2 = pack_linear(pk, "enc.2", H, H, {{0, H}});
    W.enc_ml = pack_pair(pk, "enc_mean", "enc_logvar", Z, H);
    W.prior2 = pack_linear(pk, "prior.2", H, H, {{0, H}});
    W.prior_ml = pack_pair(pk, "prior_mean", "prior_logvar", Z, H);
    W.dec2 = pack_linear(pk, "dec.2", H, H, {{0, H}});
    W.dec_mean = pack_raw_linear(pk, "dec_mean", X, H);
    W.dec_logvar = pack_raw_linear(pk, "dec_logvar", X, H);
    for (int l = 0; l < D.L; ++l) W.gru[l] = pack_gru(pk, l, R, H);
    if (graph) {
        W.g_hid = pack_graph(pk, dag ? "graph_hiddens" : "graph", D.graph, R, R, D.F, D.P);
        W.blend = pack_linear(pk, dag ? "lg_hiddens" : "lg", R, 2 * R, {{0, R}, {R, R}});
    }
    if (dag) {
        W.dec_goal0 = pack_linear(pk, "dec_goal.0", H, Dd + R + G, {{0, Dd}, {Dd, R}}, Dd + R, G);
        W.dec_goal2 = pack_linear(pk, "dec_goal.2", G, H, {{0, H}});
        W.g_goal = pack_graph(pk, "graph_goals", D.graph, G, G, D.F, D.P);
        W.lg_goals = pack_linear(pk, "lg_goals.0", G, 2 * G, {{0, G}, {G, G}});
    }
    }
    m->v2 = v2_eligible(c);
    m->wide = !m->v2 && wide_eligible(c) && !getenv("DAGNET_NO_WIDE");
    if ((m->v2 || m->wide) && pk.missing.empty()) {
        for (int bf = 0; bf < 2; ++bf) {
            m->v2_base[bf] = pk.align();
            pk.bf = bf != 0;
            v2_pack(pk, c, m->w2[bf], m->wide);
            pk.bf = false;
            if (bf) m->split_ok = !pk.split_overflow;
            if (pk.missing.empty() && !m->wide) {      // (model 3 reads the table: its one-hot tables have g_dim rows)
                const int code = dag && Dd > 32 ? 4 : c.model;
                const std::string bad = v2_check_layout(m->w2[bf], m->v2_base[bf], pk.align() - m->v2_base[bf], code, c.n_layers, c.z_dim,
                                                        c.graph_model == DAGNET_GRAPH_GCN ? 2 : 1, bf);
                if (!bad.empty()) {
                    const int rc = fail(DAGNET_ERR_INVALID, "internal error, %s", bad.c_str());
                    delete m;
                    return rc;
                }
            }
        }
        if (m->wide) wide_pack(pk, c, m->ww);
    }
    if (!pk.missing.empty()) {
        const int rc = fail(DAGNET_ERR_MISSING, "state dict entry missing or mis-shaped: %s", pk.missing.c_str());
        delete m;
        return rc;
    }
    pk.align();
    m->wp_floats = pk.buf.size();
    hipError_t e = hipGetDevice(&m->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_wp), m->wp_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->d_wp, pk.buf.data(), m->wp_floats * sizeof(float), hipMemcpyHostToDevice);
    for (int bf = 0; bf < 2 && e == hipSuccess && (m->v2 || m->wide); ++bf) {
        e = hipMalloc(reinterpret_cast<void **>(&m->d_w2[bf]), sizeof(V2Weights));
        if (e == hipSuccess) e = hipMemcpy(m->d_w2[bf], &m->w2[bf], sizeof(V2Weights), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (m->d_wp) (void)hipFree(m->d_wp);
        for (int bf = 0; bf < 2; ++bf)
            if (m->d_w2[bf]) (void)hipFree(m->d_w2[bf]);
        delete m;
        return fail(DAGNET_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return DAGNET_OK;
}

extern "C" void dagnet_model_destroy(dagnet_model *m) {
    if (!m) return;
    if (m->d_wp) (void)hipFree(m->d_wp);
    for (int bf = 0; bf < 2; ++bf)
        if (m->d_w2[bf]) (void)hipFree(m->d_w2[bf]);
    delete m;
}

// LDS layout for tile height M; returns total floats
// big: the three goal-width buffers get offsets >= total, i.e. they live in the workgroup's global slab
static DnLds make_layout(const DnDims &D, int M, int AS, bool big = false) {
    DnLds L;
    memset(&L, 0, sizeof(L));
    L.M = M;
    L.AS = AS;
    const int TW = std::max(std::max(pad32(D.H), pad32(D.R)), pad32(2 * D.Z));
    const int Hp = pad32(D.H), Rp = pad32(D.R);
    int off = 0;
    auto take = [&](int floats) {
        const int o = off;
        off += (floats + 3) & ~3;
        return o;
    };
    for (int l = 0; l < D.L; ++l) L.h[l] = take(Rp * M);
    L.t1 = take(TW * M);
    const bool graph = D.model != 0, dag = D.model == 2;
    if (dag) L.d = take(pad8(D.D) * M);
    if (graph) {
        const int gw = pad32(std::max(D.P * D.F, D.F));
        L.wh = take(gw * M);
        L.x2 = take(gw * M);
        L.adj = take(M * AS);
        L.att = take(M * AS);
        L.s12 = take(2 * std::max(D.P, 1) * M);
    }
    L.red = take(2 * DN_THREADS);
    L.xabs = take(2 * M);
    L.dm = take(2 * D.X * M);
    L.xt = take(D.X * M);
    L.ri = take(7 * M);
    const int u0 = off;
    L.t2 = u0;
    L.t3 = u0 + TW * M;
    L.px = u0 + 2 * TW * M;
    L.pz = L.px + Hp * M;
    int usize = (2 * TW + 2 * Hp) * M;
    if (dag && !big) {
        const int Gp = pad32(D.G);
        L.decg = u0;
        L.gg = u0 + Gp * M;
        L.gcomb = u0 + 2 * Gp * M;
        usize = std::max(usize, 3 * Gp * M);
    }
    off += usize;
    L.total = std::max(off, 3 * DN_THREADS * 2);   // room for the fp64 loss reduction at offset 0
    if (dag && big) {
        const int Gp = pad32(D.G);
        L.decg = L.total;
        L.gg = L.total + Gp * M;
        L.gcomb = L.total + 2 * Gp * M;
    }
    return L;
}

static const int64_t kLdsBudgetBytes = 160 * 1024;

extern "C" int dagnet_plan_create(const dagnet_model *m, const int32_t *h_scene_ptr, int32_t S, int32_t K, dagnet_plan **out) {
    if (!h_scene_ptr || !out) return fail(DAGNET_ERR_INVALID, "null argument");
    *out = nullptr;
    if (S < 1 || K < 1) return fail(DAGNET_ERR_INVALID, "need at least one scene and one sample");
    if (h_scene_ptr[0] != 0) return fail(DAGNET_ERR_INVALID, "scene_ptr[0] must be 0");
    int nmax = 0;
    for (int s = 0; s < S; ++s) {
        const int n = h_scene_ptr[s + 1] - h_scene_ptr[s];
        if (n < 1) return fail(DAGNET_ERR_INVALID, "scene %d is empty or scene_ptr is not increasing", s);
        nmax = std::max(nmax, n);
    }
    const int N = h_scene_ptr[S];
    if ((int64_t)N * K > 0x7fffffffLL) return fail(DAGNET_ERR_INVALID, "N*K exceeds 2^31 rows; shard the batch");
    if (m && m->dims.model == DAGNET_MODEL_DAGNET && 2 * nmax > m->dims.D)
        return fail(DAGNET_ERR_INVALID, "scene with %d agents exceeds n_max_agents=%d", nmax, m->dims.D / 2);
    // tile height: 64 rows when it fits the LDS, else 32; scenes must fit a tile
    int M = 0;
    DnLds L;
    const int cand[4] = {64, 32, 128, 256};
    memset(&L, 0, sizeof(L));
    for (int i = 0; m && i < 4 && !M; ++i) {
        if (cand[i] < nmax) continue;
        L = make_layout(m->dims, cand[i], nmax);
        if ((int64_t)L.total * 4 <= kLdsBudgetBytes) M = cand[i];
    }
    bool big = false;
    if (m && !M && m->dims.model == DAGNET_MODEL_DAGNET) {
        // goal-width buffers do not fit the LDS (g_dim in the hundreds): keep them in a global slab per workgroup
        const int cand_big[3] = {64, 32, 128};
        for (int i = 0; i < 3 && !M; ++i) {
            if (cand_big[i] < nmax) continue;
            L = make_layout(m->dims, cand_big[i], nmax, true);
            if ((int64_t)L.total * 4 <= kLdsBudgetBytes) { M = cand_big[i]; big = true; }
        }
    }
    if (m && !M)
        return fail(DAGNET_ERR_CAPACITY,
                    "no tile height fits: largest scene has %d agents, model needs %lld B of LDS at the smallest usable tile "
                    "(budget %lld B)",
                    nmax, (long long)make_layout(m->dims, std::max(32, pad32(nmax)), nmax).total * 4, (long long)kLdsBudgetBytes);

    std::vector<int32_t> u0, nu, row_scene(N);
    std::vector<int64_t> offd(S), offc(S);
    int64_t acc = 0;
    for (int s = 0; s < S; ++s) {
        const int b = h_scene_ptr[s], n = h_scene_ptr[s + 1] - b;
        for (int i = 0; i < n; ++i) row_scene[b + i] = s;
        offd[s] = (int64_t)b * N + b;
        offc[s] = acc;
        acc += (int64_t)n * n;
    }
    const int64_t units = (int64_t)S * K;
    int64_t u = 0;
    while (m && u < units) {
        int rows = 0, cnt = 0;
        while (u + cnt < units) {
            const int s = (int)((u + cnt) % S);
            const int n = h_scene_ptr[s + 1] - h_scene_ptr[s];
            if (rows + n > M) break;
            rows += n;
            ++cnt;
        }
        u0.push_back((int32_t)u);
        nu.push_back(cnt);
        u += cnt;
    }
    std::vector<int32_t> u2, nu2;
    if (m && (m->v2 || m->wide) && nmax <= 32) {
        int64_t v = 0;
        while (v < units) {
            int rows = 0, cnt = 0;
            while (v + cnt < units) {
                const int s = (int)((v + cnt) % S);
                const int n = h_scene_ptr[s + 1] - h_scene_ptr[s];
                if (rows + n > 32) break;
                rows += n;
                ++cnt;
            }
            u2.push_back((int32_t)v);
            nu2.push_back(cnt);
            v += cnt;
        }
    }
    dagnet_plan *p = new dagnet_plan();
    p->S = S; p->N = N; p->K = K; p->M = M; p->AS = nmax; p->lds = L;
    p->owner = m;
    p->big = big;
    p->n_tiles = (int)u0.size();
    p->compact_size = acc;
    hipError_t e = hipGetDevice(&p->device);
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes);
        if (e == hipSuccess) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up(reinterpret_cast<void **>(&p->d_scene_ptr), h_scene_ptr, sizeof(int32_t) * (S + 1));
    p->n_tiles2 = (int)u2.size();
    if (!u2.empty()) {
        up(reinterpret_cast<void **>(&p->d_tile2_u0), u2.data(), sizeof(int32_t) * u2.size());
        up(reinterpret_cast<void **>(&p->d_tile2_nu), nu2.data(), sizeof(int32_t) * nu2.size());
    }
    if (m) {
        up(reinterpret_cast<void **>(&p->d_tile_u0), u0.data(), sizeof(int32_t) * u0.size());
        up(reinterpret_cast<void **>(&p->d_tile_nu), nu.data(), sizeof(int32_t) * nu.size());
    }
    up(reinterpret_cast<void **>(&p->d_row_scene), row_scene.data(), sizeof(int32_t) * N);
    up(reinterpret_cast<void **>(&p->d_off_dense), offd.data(), sizeof(int64_t) * S);
    up(reinterpret_cast<void **>(&p->d_off_compact), offc.data(), sizeof(int64_t) * S);
    if (m && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_loss_part), sizeof(double) * 3 * std::max(p->n_tiles, p->n_tiles2));
    if (big && e == hipSuccess) {
        p->gslab = (int64_t)3 * pad32(m->dims.G) * M;
        p->gblocks = std::min(p->n_tiles, 1024);
        e = hipMalloc(reinterpret_cast<void **>(&p->d_gscratch), sizeof(float) * p->gslab * p->gblocks);
    }
    if (m) {
        DnParams P;
        memset(&P, 0, sizeof(P));
        P.d = m->dims;
        P.l = L;
        P.w = m->w;
        up(&p->d_static, &P, offsetof(DnParams, r));
    }
    if (m && m->wide && nmax <= 32 && e == hipSuccess) {
        // layer-by-layer workspace of the wide goal head, carved from one allocation (64-float aligned pieces)
        const DnDims &D = m->dims;
        const int64_t rows = (int64_t)N * K, Gp = (D.G + 3) & ~3, PF = D.P * D.F;
        WideBuffers &b = p->wb;
        memset(&b, 0, sizeof(b));
        b.a0_ld = m->ww.g0_ld;
        b.hs_ld = (PF + 2 * D.P + 3) & ~3;
        b.Gp = Gp;
        int64_t off = 0;
        auto take = [&](int64_t floats) {
            const int64_t o = off;
            off += (floats + 63) & ~(int64_t)63;
            return o;
        };
        const int64_t oA0 = take(rows * b.a0_ld), oT1 = take(rows * D.H), oA1 = take(rows * 2 * Gp), oB = take(rows * Gp),
                      oHS = take(rows * b.hs_ld), oX2 = take(rows * PF), oS2 = take(rows * 2), oadj = take((int64_t)K * acc),
                      oh = take((int64_t)D.L * rows * D.R), opos = take(rows * 2), ogoal = take(rows);
        e = hipMalloc(reinterpret_cast<void **>(&p->d_wide), sizeof(float) * off);
        if (e == hipSuccess) e = hipMemset(p->d_wide, 0, sizeof(float) * off);
        if (e == hipSuccess) {
            float *w = p->d_wide;
            b.A0 = w + oA0; b.T1 = w + oT1; b.A1 = w + oA1; b.B = w + oB; b.HS = w + oHS; b.X2 = w + oX2; b.S2 = w + oS2;
            b.adj = w + oadj; b.h = w + oh; b.pos = w + opos;
            b.goal = reinterpret_cast<int32_t *>(w + ogoal);
            p->wide = true;
        }
    }
    if (e != hipSuccess) {
        dagnet_plan_destroy(p);
        return fail(DAGNET_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return DAGNET_OK;
}

extern "C" void dagnet_plan_destroy(dagnet_plan *p) {
    if (!p) return;
    void *ptrs[] = {p->d_scene_ptr, p->d_tile_u0, p->d_tile_nu, p->d_row_scene, p->d_off_dense, p->d_off_compact, p->d_loss_part, p->d_static, p->d_tile2_u0, p->d_tile2_nu, p->d_gscratch, p->d_wide};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (p->busy) (void)hipEventDestroy(p->busy);
    delete p;
}

extern "C" int dagnet_plan_kernel(const dagnet_plan *p);
extern "C" int dagnet_plan_info(const dagnet_plan *p, int32_t *tile_rows, int32_t *n_tiles, int64_t *lds_bytes) {
    if (!p) return fail(DAGNET_ERR_INVALID, "null plan");
    // geometry of the kernel dagnet_sample() will run on this plan (dagnet_plan_kernel)
    const int kind = dagnet_plan_kernel(p);
    if (kind == 2 || kind == 3) {       // register-resident: one wave per tile of <= 32 rows; static LDS of the wave
        const dagnet_model *m = p->owner;
        const int64_t stash = m->dims.model == DAGNET_MODEL_DAGNET && kind == 2 ? (int64_t)m->dims.L * 32 * 64
                              : (m->dims.L == 2 ? 32 * 64 : 1);
        if (tile_rows) *tile_rows = 32;
        if (n_tiles) *n_tiles = p->n_tiles2;
        if (lds_bytes) *lds_bytes = (stash + 32 * 4 + 64) * 4;
        return DAGNET_OK;
    }
    if (tile_rows) *tile_rows = p->M;
    if (n_tiles) *n_tiles = p->n_tiles;
    if (lds_bytes) *lds_bytes = (int64_t)p->lds.total * 4;
    return DAGNET_OK;
}

// template instance of the register-resident kernels: DAG-Net with more than 16 agents per scene needs two disposition tiles
static int v2_model_code(const dagnet_model *m) { return m->dims.model == DAGNET_MODEL_DAGNET && m->dims.D > 32 ? 4 : m->dims.model; }

static int check_plan(const dagnet_model *m, const dagnet_plan *p) {
    if (!p->d_static || p->n_tiles < 1) return fail(DAGNET_ERR_INVALID, "plan was created without a model (tables only)");
    if (p->owner != m) return fail(DAGNET_ERR_INVALID, "plan was created for another model");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev != p->device || dev != m->device)
        return fail(DAGNET_ERR_INVALID, "model/plan belong to device %d/%d but device %d is current", m->device, p->device, dev);
    return DAGNET_OK;
}

#ifdef DN_PROFILE
static long long *g_prof = nullptr;
extern "C" void dagnet_debug_set_prof(long long *d_prof) { g_prof = d_prof; }
#endif

static void fill_common(DnParams &P, const dagnet_model *m, const dagnet_plan *p) {
    memset(&P, 0, sizeof(P));
#ifdef DN_PROFILE
    P.r.prof = g_prof;
#endif
    P.d = m->dims;
    P.l = p->lds;
    P.w = m->w;
    P.r.wp = m->d_wp;
    P.r.scene_ptr = p->d_scene_ptr;
    P.r.tile_u0 = p->d_tile_u0;
    P.r.tile_nu = p->d_tile_nu;
    P.r.S = p->S;
    P.r.N = p->N;
    P.r.K = p->K;
    P.r.n_tiles = p->n_tiles;
    P.r.gscratch = p->d_gscratch;
    P.r.gslab = p->gslab;
    P.r.gblocks = p->gblocks;
#ifdef DN_PROFILE
    P.r.prof = g_prof;
#endif
}


// ---------------------------------------------------------------------------------------------
// Hazard guards
// ---------------------------------------------------------------------------------------------
// A plan's workspace (loss partials of the teacher-forced pass, state and layer buffers of the wide goal head, goal slabs of big-G
// plans) is written by the calls that use it, so two such calls on one plan may not be in flight at once.  Calls on the same
// stream order themselves; a call on ANOTHER stream while the previous one has not completed is refused here instead of
// corrupting both results.  Streams under graph capture are not tracked (an event recorded into a capture cannot be queried).
static bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

static int plan_enter(const dagnet_plan *p, hipStream_t st, const char *what) {
    // A call that writes buffers of the plan while an earlier one on ANOTHER stream may still be running is ordered behind it on
    // the device (hipStreamWaitEvent on the earlier call's completion event).  Round 2 refused such a call after a host-side
    // hipEventQuery - which also refused programs that HAD ordered the two streams with an event of their own, whenever the GPU
    // was simply not done yet (timing-dependent failures of valid programs).
    (void)what;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->armed || p->busy_stream == st || stream_capturing(st)) return DAGNET_OK;
    HIP_TRY(hipStreamWaitEvent(st, p->busy, 0));
    return DAGNET_OK;
}

static int plan_leave(const dagnet_plan *p, hipStream_t st) {
    std::lock_guard<std::mutex> lk(p->mu);
    if (stream_capturing(st)) return DAGNET_OK;
    if (!p->busy) HIP_TRY(hipEventCreateWithFlags(&p->busy, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->busy, st));
    p->busy_stream = st;
    p->armed = true;
    return DAGNET_OK;
}

// NaN in the state a call starts from (h0 / x0, or the observed trajectory): poison what the reference would poison
// (aux_kernels.hip dn_nan_rows_kernel / dn_nan_spread_kernel).  Runs after the call's kernels on the same stream.
static DnNanSrc nan_src(const void *p, int64_t plane_stride, int planes, int cols) {
    return DnNanSrc{static_cast<const uint32_t *>(p), plane_stride, planes, cols};
}
static DnNanDst nan_dst(void *p, int64_t plane_stride, int64_t planes, int cols) {
    return DnNanDst{static_cast<uint32_t *>(p), plane_stride, planes, cols};
}

// The rollout kernels' dense products: split (two fp16 terms per fp32 value on the 16-bit matrix pipe, params_v2.h) unless switched off -
// DAGNET_NO_BF16X3 in the environment or bit 3 of dagnet_op_set_fused, the switch of the wide GEMM's split products.
static int fused_mask();
static int rollout_bf3() { return (fused_mask() >> 3) & 1; }

static int sample_impl(const dagnet_model *m, const dagnet_plan *p, int32_t T, const float *d_h0, const float *d_x0,
                       const int32_t *d_g0, const float *d_eps, const float *d_q, const int32_t *d_force_goals, uint64_t seed,
                       float *d_out, int32_t *d_out_goals, void *stream) {
    if (!m || !p) return fail(DAGNET_ERR_INVALID, "null argument");
    if (m->v2 && p->n_tiles2 > 0 && !getenv("DAGNET_FORCE_GENERIC")) {
        const int bf = rollout_bf3() && m->split_ok;
        V2Run r;
        memset(&r, 0, sizeof(r));
        r.wp = m->d_wp + m->v2_base[bf];      // rollout_v2: compile-time offsets relative to the v2 region
        r.scene_ptr = p->d_scene_ptr; r.tile_u0 = p->d_tile2_u0; r.tile_nu = p->d_tile2_nu;
        r.S = p->S; r.N = p->N; r.K = p->K; r.T = T; r.n_tiles = p->n_tiles2;
        r.G = m->dims.G; r.D = m->dims.D; r.adj_type = m->dims.adj_type; r.top_k = m->dims.knn_k;
        r.alpha = m->dims.alpha; r.sigma = m->dims.sigma;
        r.h0 = d_h0; r.x0 = d_x0; r.g0 = d_g0; r.eps = d_eps; r.q = d_q; r.force_goals = d_force_goals;
        r.seed = seed; r.out = d_out; r.out_goals = d_out_goals;
#ifdef DN_PROFILE
        if (getenv("DAGNET_V2_DUMP")) r.dbg = reinterpret_cast<float *>(g_prof);
        else r.prof = g_prof;
#endif
        HIP_TRY((bf ? dn_launch_rollout_v2_bf3 : dn_launch_rollout_v2)(v2_model_code(m), m->dims.L, m->dims.Z, m->dims.graph, m->d_w2[bf], &r,
                                                                       static_cast<hipStream_t>(stream)));
        return DAGNET_OK;
    }
    if (m->wide && p->wide && !getenv("DAGNET_FORCE_GENERIC")) {
        // wide goal head: per step, the goal head layer by layer over all K*N rows, then one step of the
        // register-resident kernel (model 3) with the drawn goals; state lives in the plan's workspace
        hipStream_t st = static_cast<hipStream_t>(stream);
        const DnDims &D = m->dims;
        const int bf = rollout_bf3() && m->split_ok;
        HIP_TRY(dn_wide_init_state(p->N, p->K, D.L, D.R, D.G, d_h0, d_x0, d_g0, p->wb.h, p->wb.pos, p->wb.goal, st));
        WideStep ws;
        memset(&ws, 0, sizeof(ws));
        ws.wp = m->d_wp; ws.w = m->ww; ws.b = p->wb;
        ws.scene_ptr = p->d_scene_ptr; ws.row_scene = p->d_row_scene; ws.off_compact = p->d_off_compact; ws.compact = p->compact_size;
        ws.N = p->N; ws.K = p->K; ws.S = p->S; ws.T = T;
        ws.D = D.D; ws.R = D.R; ws.H = D.H; ws.G = D.G; ws.L = D.L; ws.P = D.P; ws.F = D.F;
        ws.adj_type = D.adj_type; ws.top_k = D.knn_k; ws.sigma = D.sigma; ws.alpha = D.alpha;
        ws.q = d_q; ws.force_goals = d_force_goals; ws.seed = seed; ws.out_goals = d_out_goals;
        V2Run r;
        memset(&r, 0, sizeof(r));
        r.wp = m->d_wp;                    // model 3: absolute offsets from the table
        r.scene_ptr = p->d_scene_ptr; r.tile_u0 = p->d_tile2_u0; r.tile_nu = p->d_tile2_nu;
        r.S = p->S; r.N = p->N; r.K = p->K; r.T = T; r.n_tiles = p->n_tiles2;
        r.G = D.G; r.D = D.D; r.adj_type = D.adj_type; r.top_k = D.knn_k; r.alpha = D.alpha; r.sigma = D.sigma;
        r.h0 = p->wb.h; r.x0 = p->wb.pos; r.g0 = p->wb.goal; r.eps = d_eps; r.seed = seed; r.out = d_out;
        r.h_out = p->wb.h; r.x_out = p->wb.pos;
        for (int t = 0; t < T; ++t) {
            ws.t = t;
            HIP_TRY(dn_wide_goal_head(&ws, st));
            r.t_begin = t; r.t_end = t + 1;
            HIP_TRY((bf ? dn_launch_rollout_v2_bf3 : dn_launch_rollout_v2)(3, D.L, D.Z, D.graph, m->d_w2[bf], &r, st));
        }
        return DAGNET_OK;
    }
    DnParams P;
    fill_common(P, m, p);
    P.r.T = T;
    P.r.h0 = d_h0; P.r.x0 = d_x0; P.r.g0 = d_g0;
    P.r.eps = d_eps; P.r.q = d_q; P.r.force_goals = d_force_goals;
    P.r.seed = seed;
    P.r.out = d_out; P.r.out_goals = d_out_goals;
    HIP_TRY(dn_launch_rollout(0, p->d_static, &P.r, (size_t)p->lds.total * 4, static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_sample(const dagnet_model *m, const dagnet_plan *p, int32_t T, const float *d_h0, const float *d_x0,
                             const int32_t *d_g0, const float *d_eps, const float *d_q, const int32_t *d_force_goals, uint64_t seed,
                             float *d_out, int32_t *d_out_goals, void *stream) {
    if (!m || !p) return fail(DAGNET_ERR_INVALID, "null argument");
    if (T < 0) return fail(DAGNET_ERR_INVALID, "T must be >= 0");
    if (T == 0) return DAGNET_OK;          // nothing to roll out (the output tensor is empty)
    if (!d_h0 || !d_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (m->dims.model == DAGNET_MODEL_GRAPH) return fail(DAGNET_ERR_INVALID, "stand-alone graph modules have no sample()");
    if (m->dims.model != DAGNET_MODEL_VRNN && !d_x0) return fail(DAGNET_ERR_INVALID, "x_abs_start is required for graph models");
    if (m->dims.model == DAGNET_MODEL_DAGNET && !d_g0) return fail(DAGNET_ERR_INVALID, "g_start is required for DAG-Net");
    if (m->dims.model == DAGNET_MODEL_DAGNET && 2 * p->AS > m->dims.D) return fail(DAGNET_ERR_INVALID, "plan was made for another model");
    if (int rc = check_plan(m, p)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the register-resident kernel keeps all state in registers; the wide-head and big-G paths work in the plan's buffers
    const bool owns = !(m->v2 && p->n_tiles2 > 0) && (p->wide || p->big);
    if (owns)
        if (int rc = plan_enter(p, st, "dagnet_sample")) return rc;
    if (int rc = sample_impl(m, p, T, d_h0, d_x0, d_g0, d_eps, d_q, d_force_goals, seed, d_out, d_out_goals, stream)) return rc;
    const bool graph = m->dims.model != DAGNET_MODEL_VRNN;
    const int64_t plane = (int64_t)p->N * 2;                      // out is [K][T][N][2]
    // (h0 only: the start positions enter through the adjacency alone, where the reference's own `adj > 0` mask drops a NaN)
    const DnNanSrc src[2] = {nan_src(d_h0, (int64_t)p->N * m->dims.R, m->dims.L, m->dims.R), nan_src(nullptr, 0, 0, 0)};
    if (!graph) {               // independent rows: the row's own trajectory, every step
        const DnNanDst dst = nan_dst(d_out, plane, (int64_t)p->K * T, 2);
        HIP_TRY(dn_launch_nan_rows(p->N, src, &dst, nullptr, nullptr, st));
    } else {                    // step 0 of the row; from step 1 on the graph has spread it over the batch
        const DnNanDst dst = nan_dst(d_out, plane * T, p->K, 2);
        float *marker = T > 1 ? d_out + plane : nullptr;
        HIP_TRY(dn_launch_nan_rows(p->N, src, &dst, nullptr, marker, st));
        if (T > 1) HIP_TRY(dn_launch_nan_spread(marker, 0, marker, p->K, plane * T, plane * (T - 1), st));
    }
    if (owns) return plan_leave(p, st);
    return DAGNET_OK;
}

static int set_adj(DnParams &P, const dagnet_plan *p, const float *d_adj, int64_t adj_dense_ld) {
    if (!d_adj) return fail(DAGNET_ERR_INVALID, "adjacency is required for graph models");
    if (adj_dense_ld != 0 && adj_dense_ld != p->N) return fail(DAGNET_ERR_INVALID, "adj_dense_ld must be 0 (compact) or N");
    P.r.adj = d_adj;
    P.r.adj_ld = adj_dense_ld;
    P.r.adj_tstride = adj_dense_ld ? (int64_t)p->N * p->N : p->compact_size;
    P.r.adj_off = adj_dense_ld ? p->d_off_dense : p->d_off_compact;
    return DAGNET_OK;
}

static int forward_impl(const dagnet_model *m, const dagnet_plan *p, int32_t T, const float *d_traj, const float *d_traj_rel,
                        const int32_t *d_goals, const float *d_adj, int64_t adj_dense_ld, const float *d_eps, uint64_t seed,
                        double *d_losses, float *d_h_out, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    DnParams P;
    fill_common(P, m, p);
    if (m->dims.model != DAGNET_MODEL_VRNN) {
        const int rc = set_adj(P, p, d_adj, adj_dense_ld);
        if (rc) return rc;
    }
    if (m->wide && p->wide && !getenv("DAGNET_FORCE_GENERIC")) {
        // g_dim in the hundreds: per step the goal head layer by layer over all N rows (wide_head.hip, ending in the CE term),
        // then one step of the register-resident teacher-forced kernel (model 3).  h_out doubles as the hidden state.
        const DnDims &D = m->dims;
        const int64_t N = p->N;
        HIP_TRY(hipMemsetAsync(d_h_out, 0, sizeof(float) * D.L * N * D.R, st));
        HIP_TRY(hipMemsetAsync(p->d_loss_part, 0, sizeof(double) * 3 * p->n_tiles2, st));
        float *ce_acc = p->wb.pos;           // the positions state is unused in the teacher-forced pass ([rows][2] >= [rows])
        HIP_TRY(hipMemsetAsync(ce_acc, 0, sizeof(float) * N, st));
        WideStep ws;
        memset(&ws, 0, sizeof(ws));
        ws.wp = m->d_wp; ws.w = m->ww; ws.b = p->wb;
        ws.b.h = d_h_out;
        ws.scene_ptr = p->d_scene_ptr; ws.row_scene = p->d_row_scene; ws.off_compact = p->d_off_compact; ws.compact = 0;
        ws.N = p->N; ws.K = 1; ws.S = p->S; ws.T = T;
        ws.D = D.D; ws.R = D.R; ws.H = D.H; ws.G = D.G; ws.L = D.L; ws.P = D.P; ws.F = D.F;
        ws.adj_type = D.adj_type; ws.top_k = D.knn_k; ws.sigma = D.sigma; ws.alpha = D.alpha;
        ws.ext_adj_ld = P.r.adj_ld; ws.ext_adj_off = P.r.adj_off; ws.ce_acc = ce_acc;
        V2Run r;
        memset(&r, 0, sizeof(r));
        r.wp = m->d_wp;
        r.scene_ptr = p->d_scene_ptr; r.tile_u0 = p->d_tile2_u0; r.tile_nu = p->d_tile2_nu;
        r.S = p->S; r.N = p->N; r.K = 1; r.T = T; r.n_tiles = p->n_tiles2;
        r.G = D.G; r.D = D.D; r.adj_type = D.adj_type; r.top_k = D.knn_k; r.alpha = D.alpha; r.sigma = D.sigma;
        r.eps = d_eps; r.seed = seed; r.h_out = d_h_out;
        r.traj = d_traj; r.traj_rel = d_traj_rel; r.goals = d_goals;
        r.adj = P.r.adj; r.adj_ld = P.r.adj_ld; r.adj_tstride = P.r.adj_tstride; r.adj_off = P.r.adj_off;
        r.loss_part = p->d_loss_part;
        for (int t = 1; t < T; ++t) {
            ws.t = t;
            ws.ext_pos = d_traj + (int64_t)t * N * 2;
            ws.ext_goal_prev = d_goals + (int64_t)(t - 1) * N;
            ws.ext_goal_t = d_goals + (int64_t)t * N;
            ws.ext_adj = P.r.adj + (int64_t)t * P.r.adj_tstride;
            HIP_TRY(dn_wide_goal_head(&ws, st));
            r.t_begin = t; r.t_end = t + 1;
            HIP_TRY(dn_launch_forward_v2(3, D.L, D.Z, D.graph, m->d_w2[0], &r, st));
        }
        HIP_TRY(dn_wide_ce_finish(N, ce_acc, p->d_loss_part, st));
        HIP_TRY(dn_launch_loss_finalize(p->n_tiles2, p->N, p->d_loss_part, d_losses, st));
        return DAGNET_OK;
    }
    if (m->v2 && p->n_tiles2 > 0 && !getenv("DAGNET_FORCE_GENERIC")) {
        // register-resident teacher-forced pass (forward_v2.hip); GAT consumes only the adjacency's support
        V2Run r;
        memset(&r, 0, sizeof(r));
        r.wp = m->d_wp + m->v2_base[0];   // compile-time offsets relative to the (fp32) v2 region
        r.scene_ptr = p->d_scene_ptr; r.tile_u0 = p->d_tile2_u0; r.tile_nu = p->d_tile2_nu;
        r.S = p->S; r.N = p->N; r.K = 1; r.T = T; r.n_tiles = p->n_tiles2;
        r.G = m->dims.G; r.D = m->dims.D; r.adj_type = m->dims.adj_type; r.top_k = m->dims.knn_k; r.alpha = m->dims.alpha; r.sigma = m->dims.sigma;
        r.eps = d_eps; r.seed = seed; r.h_out = d_h_out;
        r.traj = d_traj; r.traj_rel = d_traj_rel; r.goals = d_goals;
        r.adj = P.r.adj; r.adj_ld = P.r.adj_ld; r.adj_tstride = P.r.adj_tstride; r.adj_off = P.r.adj_off;
        r.loss_part = p->d_loss_part;
        HIP_TRY(dn_launch_forward_v2(v2_model_code(m), m->dims.L, m->dims.Z, m->dims.graph, m->d_w2[0], &r, st));
        HIP_TRY(dn_launch_loss_finalize(p->n_tiles2, p->N, p->d_loss_part, d_losses, st));
        return DAGNET_OK;
    }
    P.r.T = T;
    P.r.traj = d_traj; P.r.traj_rel = d_traj_rel; P.r.goals = d_goals;
    P.r.eps = d_eps;
    P.r.seed = seed;
    P.r.loss_part = p->d_loss_part;
    P.r.h_out = d_h_out;
    HIP_TRY(dn_launch_rollout(1, p->d_static, &P.r, (size_t)p->lds.total * 4, st));
    HIP_TRY(dn_launch_loss_finalize(p->n_tiles, p->N, p->d_loss_part, d_losses, st));
    return DAGNET_OK;
}

extern "C" int dagnet_forward(const dagnet_model *m, const dagnet_plan *p, int32_t T, const float *d_traj, const float *d_traj_rel,
                              const int32_t *d_goals, const float *d_adj, int64_t adj_dense_ld, const float *d_eps, uint64_t seed,
                              double *d_losses, float *d_h_out, void *stream) {
    if (!m || !p || !d_traj_rel || !d_losses || !d_h_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (p->K != 1) return fail(DAGNET_ERR_INVALID, "forward needs a plan with n_samples == 1");
    if (T < 1) return fail(DAGNET_ERR_INVALID, "T must be >= 1");
    if (m->dims.model == DAGNET_MODEL_GRAPH) return fail(DAGNET_ERR_INVALID, "stand-alone graph modules have no forward pass");
    const bool graph = m->dims.model != DAGNET_MODEL_VRNN;
    if (m->dims.model == DAGNET_MODEL_DAGNET && (!d_traj || !d_goals)) return fail(DAGNET_ERR_INVALID, "traj and goals are required for DAG-Net");
    if (int rc = check_plan(m, p)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = plan_enter(p, st, "dagnet_forward")) return rc;      // every path sums the losses in the plan's partials
    if (int rc = forward_impl(m, p, T, d_traj, d_traj_rel, d_goals, d_adj, adj_dense_ld, d_eps, seed, d_losses, d_h_out, stream)) return rc;
    // the losses are means over all rows: NaN for a NaN anywhere; h of the row (VRNN) or of the whole batch (graph)
    const int64_t pl = (int64_t)p->N * 2, hp = (int64_t)p->N * m->dims.R;
    const DnNanSrc src[2] = {nan_src(d_traj_rel, pl, T, 2), nan_src(graph ? d_traj : nullptr, pl, T, 2)};
    const DnNanDst dst = graph ? nan_dst(nullptr, 0, 0, 0) : nan_dst(d_h_out, hp, m->dims.L, m->dims.R);
    HIP_TRY(dn_launch_nan_rows(p->N, src, &dst, d_losses, nullptr, st));
    if (graph) HIP_TRY(dn_launch_nan_spread(d_losses, 1, d_h_out, 1, 0, hp * m->dims.L, st));
    return plan_leave(p, st);
}

extern "C" int dagnet_graph_forward(const dagnet_model *m, const dagnet_plan *p, int32_t which, const float *d_x, const float *d_adj,
                                    int64_t adj_dense_ld, float *d_out, void *stream) {
    if (!m || !p || !d_x || !d_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (p->K != 1) return fail(DAGNET_ERR_INVALID, "graph_forward needs a plan with n_samples == 1");
    if (m->dims.model == DAGNET_MODEL_VRNN) return fail(DAGNET_ERR_INVALID, "VRNN has no graph");
    if (which != 0 && !(which == 1 && m->dims.model == DAGNET_MODEL_DAGNET)) return fail(DAGNET_ERR_INVALID, "no such graph: %d", which);
    if (int rc = check_plan(m, p)) return rc;
    DnParams P;
    fill_common(P, m, p);
    const int rc = set_adj(P, p, d_adj, adj_dense_ld);
    if (rc) return rc;
    P.r.gx = d_x;
    P.r.gout = d_out;
    P.r.gwhich = which;
    HIP_TRY(dn_launch_rollout(2, p->d_static, &P.r, (size_t)p->lds.total * 4, static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_adjacency(const dagnet_plan *p, int32_t adjacency_type, double sigma, int32_t top_k, int32_t T, const float *d_pos,
                                float *d_out, int64_t dense_ld, void *stream) {
    if (!p || !d_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (adjacency_type < 0 || adjacency_type > 2) return fail(DAGNET_ERR_INVALID, "adjacency_type must be 0,1,2");
    if (adjacency_type != 0 && !d_pos) return fail(DAGNET_ERR_INVALID, "positions are required");
    if (adjacency_type == 2 && top_k < 1) return fail(DAGNET_ERR_INVALID, "top_k must be >= 1");
    if (dense_ld != 0 && dense_ld != p->N) return fail(DAGNET_ERR_INVALID, "dense_ld must be 0 (compact) or N");
    if (T < 1) return DAGNET_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t tstride = dense_ld ? (int64_t)p->N * p->N : p->compact_size;
    if (dense_ld) HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(float) * tstride * T, st));
    HIP_TRY(dn_launch_adjacency(adjacency_type, sigma, top_k, p->d_scene_ptr, p->d_row_scene, dense_ld ? p->d_off_dense : p->d_off_compact,
                                p->N, T, d_pos, d_out, dense_ld, tstride, st));
    return DAGNET_OK;
}

extern "C" int dagnet_plan_kernel(const dagnet_plan *p) {
    if (p && p->wide && p->owner && p->owner->wide && !getenv("DAGNET_FORCE_GENERIC")) return 3;
    if (!p || !p->owner) return -1;
    if (p->owner->v2 && p->n_tiles2 > 0 && !getenv("DAGNET_FORCE_GENERIC")) return 2;
    return p->big ? 1 : 0;
}

extern "C" int64_t dagnet_plan_compact_adj_size(const dagnet_plan *p) { return p ? p->compact_size : -1; }

extern "C" int dagnet_best_of_k(const dagnet_plan *p, int32_t K, int32_t T, const float *d_samples_rel, const float *d_start,
                                const float *d_gt, float *d_ade_raw, float *d_fde_raw, float *d_scene_best, double *d_sums, void *stream) {
    if (!p || !d_samples_rel || !d_start || !d_gt || !d_ade_raw || !d_fde_raw || !d_scene_best || !d_sums)
        return fail(DAGNET_ERR_INVALID, "null argument");
    if (K < 1 || T < 1) return fail(DAGNET_ERR_INVALID, "K and T must be >= 1");
    HIP_TRY(dn_launch_best_of_k(p->d_scene_ptr, p->S, p->N, K, T, d_samples_rel, d_start, d_gt, d_ade_raw, d_fde_raw, d_scene_best, d_sums,
                                static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_relative_to_abs(int32_t T, int32_t N, const float *d_rel, const float *d_start, float *d_out, void *stream) {
    if (!d_rel || !d_start || !d_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (T < 0 || N < 0) return fail(DAGNET_ERR_INVALID, "negative size");
    HIP_TRY(dn_launch_rel_to_abs(T, N, d_rel, d_start, d_out, static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_categorical(int32_t N, int32_t G, const float *d_weights, const float *d_q, uint64_t seed, int32_t *d_out,
                                  void *stream) {
    if (!d_weights || !d_out) return fail(DAGNET_ERR_INVALID, "null argument");
    if (N < 0 || G < 1) return fail(DAGNET_ERR_INVALID, "need G >= 1");
    HIP_TRY(dn_launch_categorical(N, G, d_weights, d_q, seed, d_out, static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_displacement_errors(int32_t T, int32_t N, const float *d_pred, const float *d_gt, float *d_ade, float *d_fde,
                                          void *stream) {
    if (!d_pred || !d_gt || !d_ade || !d_fde) return fail(DAGNET_ERR_INVALID, "null argument");
    if (T < 1 || N < 0) return fail(DAGNET_ERR_INVALID, "need T >= 1");
    HIP_TRY(dn_launch_abs_errors(T, N, d_pred, d_gt, d_ade, d_fde, static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

extern "C" int dagnet_scene_min_sum(const dagnet_plan *p, int32_t K, const float *d_ade_raw, const float *d_fde_raw, float *d_scene_best,
                                    double *d_sums, void *stream) {
    if (!p || !d_ade_raw || !d_fde_raw || !d_scene_best || !d_sums) return fail(DAGNET_ERR_INVALID, "null argument");
    if (K < 1) return fail(DAGNET_ERR_INVALID, "K must be >= 1");
    HIP_TRY(dn_launch_scene_min_sum(p->d_scene_ptr, p->S, p->N, K, d_ade_raw, d_fde_raw, d_scene_best, d_sums,
                                    static_cast<hipStream_t>(stream)));
    return DAGNET_OK;
}

// ---------------------------------------------------------------------------------------------
// training-step operators (row T): thin checked entry points over gemm_f32.hip / train_ops.hip
// ---------------------------------------------------------------------------------------------
extern "C" hipError_t dn_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, float *C,
                              int64_t ldc, int M, int N, int K, int act, float slope, int accumulate, const float *gather,
                              const int32_t *gidx, int64_t gs_g, int64_t gs_n, hipStream_t st);
extern "C" int64_t dn_gemm_tn_workspace(int M, int N, int K);
extern "C" hipError_t dn_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int M, int N, int K,
                                 float *part, int accumulate, hipStream_t st);
extern "C" hipError_t dn_gemm_tn_idx(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, float *C, int64_t ldc,
                                     float *dbias, int M, int N, int K, float *part, int accumulate, hipStream_t st);
extern "C" hipError_t dn_gemm_tn_staged(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, float *C,
                                        int64_t ldc, float *dbias, int want_bias, int M, int N, int K, float *part, int accumulate,
                                        int stages, int part_accumulate, hipStream_t st);
extern "C" hipError_t dn_t_act_bwd(int64_t, int, const float *, int64_t, const float *, int64_t, float *, int64_t, int, float, hipStream_t);
extern "C" hipError_t dn_t_copy2d(int64_t, int, const float *, int64_t, float *, int64_t, int, hipStream_t);
extern "C" hipError_t dn_t_softmax_fwd(int64_t, int, const float *, int64_t, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_softmax_bwd(int64_t, int, const float *, int64_t, const float *, int64_t, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_gru_fwd(int64_t, int, const float *, const float *, const float *, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_gru_bwd(int64_t, int, const float *, const float *, const float *, const float *, float *, float *, float *,
                                   hipStream_t);
extern "C" hipError_t dn_t_vae_fwd(int64_t, int, const float *, const float *, const float *, const float *, const float *, float *, float *,
                                   hipStream_t);
extern "C" hipError_t dn_t_vae_bwd(int64_t, int, const float *, const float *, const float *, const float *, const float *, const float *,
                                   const float *, float, float *, float *, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_nll_fwd(int64_t, int, const float *, const float *, const float *, float *, hipStream_t);
extern "C" hipError_t dn_t_nll_bwd(int64_t, int, const float *, const float *, const float *, const float *, float, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_ce_fwd(int64_t, const float *, int64_t, const int32_t *, float *, hipStream_t);
extern "C" hipError_t dn_t_ce_bwd(int64_t, int, const int32_t *, const float *, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_sum(int64_t, const float *, float, float *, int, hipStream_t);
extern "C" int64_t dn_t_col_workspace(int64_t M, int N);
extern "C" hipError_t dn_t_colsum(int64_t, int, const float *, int64_t, float *, int, float *, hipStream_t);
extern "C" hipError_t dn_t_bn_fwd(int64_t, int, const float *, int64_t, const float *, const float *, float, float, int, float *, int64_t,
                                  float *, float *, float *, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_bn_bwd(int64_t, int, const float *, int64_t, const float *, const float *, const float *, int64_t, const float *,
                                  int64_t, int, float *, int64_t, float *, float *, int, float *, float *, hipStream_t);
extern "C" int64_t dn_t_gat_lds_bytes(int n_max, int F, int backward);
extern "C" hipError_t dn_t_gat_attn_fwd(const int32_t *, const int64_t *, int, int, int64_t, const float *, const float *, int64_t, const float *, int,
                                        int, float, int, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_gat_attn_bwd(const int32_t *, const int64_t *, int, int, int64_t, const float *, const float *, int64_t, const float *, int,
                                        int, float, int, const float *, int64_t, const float *, int64_t, float *, int64_t, float *,
                                        hipStream_t);
extern "C" hipError_t dn_t_adj_matmul(const int32_t *, const int64_t *, const int32_t *, int, int64_t, int, int64_t, const float *, const float *,
                                      int64_t, const float *, int, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_disposition(const int32_t *, const int64_t *, const int32_t *, int, int64_t, int, const float *, float *, int64_t,
                                       hipStream_t);
extern "C" hipError_t dn_t_sumsq(int64_t, const float *, double *, hipStream_t);
extern "C" hipError_t dn_t_adam(int64_t, float *, float *, float *, float *, const double *, float, float, float, float, float, float, float,
                                hipStream_t);

#define OP_ST static_cast<hipStream_t>(stream)
#define OP_NEED(cond, msg)                                        \
    do {                                                          \
        if (!(cond)) return fail(DAGNET_ERR_INVALID, "%s", msg);  \
    } while (0)

extern "C" int dagnet_op_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, int32_t transb, const float *bias, float *C,
                              int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, float slope, int32_t accumulate,
                              const float *gather, const int32_t *gidx, int64_t gs_g, int64_t gs_n, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || N == 0) return DAGNET_OK;
    OP_NEED(A && B && C, "null operand");
    OP_NEED(lda >= K && ldc >= N && ldb >= (transb ? N : K), "row stride smaller than the row");
    OP_NEED(!gather || gidx, "gather without indices");
    HIP_TRY(dn_gemm(A, lda, B, ldb, transb, bias, C, ldc, M, N, K, act, slope, accumulate, gather, gidx, gs_g, gs_n, OP_ST));
    return DAGNET_OK;
}
extern "C" hipError_t dn_gemm_skinny_dact(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *P, int64_t ldp, int dact,
                                          float dslope, const float *B, int64_t ldb, int transb, float *C, int64_t ldc, int M, int N,
                                          int K, int accumulate, hipStream_t st);
extern "C" hipError_t dn_t_act_bwd(int64_t M, int N, const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp,
                                   int act, float slope, hipStream_t st);
// dP = dY * act'(Y) and C = [C +] dP op(B) in one pass where the shape allows (gemm_skinny.hip), else as the two steps
extern "C" int dagnet_op_gemm_dact(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp, int32_t act,
                                   float slope, const float *B, int64_t ldb, int32_t transb, float *C, int64_t ldc, int32_t M, int32_t N,
                                   int32_t K, int32_t accumulate, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || K == 0) return DAGNET_OK;
    OP_NEED(dY && Y && dP && B && C, "null operand");
    OP_NEED(ldd >= K && ldy >= K && ldp >= K && ldc >= N && ldb >= (transb ? N : K), "row stride smaller than the row");
    OP_NEED(act >= 0 && act <= 3, "unknown activation");
    static const bool off = getenv("DAGNET_NO_DACT") != nullptr;
    const hipError_t e = off ? hipErrorNotSupported
                             : dn_gemm_skinny_dact(dY, ldd, Y, ldy, dP, ldp, act, slope, B, ldb, transb, C, ldc, M, N, K, accumulate, OP_ST);
    if (e == hipErrorNotSupported) {
        HIP_TRY(dn_t_act_bwd(M, K, dY, ldd, Y, ldy, dP, ldp, act, slope, OP_ST));
        if (N > 0) HIP_TRY(dn_gemm(dP, ldp, B, ldb, transb, nullptr, C, ldc, M, N, K, 0, 0.f, accumulate, nullptr, nullptr, 0, 0, OP_ST));
        return DAGNET_OK;
    }
    HIP_TRY(e);
    return DAGNET_OK;
}
extern "C" int64_t dagnet_op_gemm_tn_workspace(int32_t M, int32_t N, int32_t K) { return dn_gemm_tn_workspace(M, N, K); }
extern "C" int dagnet_op_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float *dbias, int32_t M,
                                 int32_t N, int32_t K, float *workspace, int32_t accumulate, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (N == 0 || K == 0) return DAGNET_OK;
    OP_NEED(A && B && C && workspace, "null operand");
    OP_NEED(lda >= N && ldb >= K && ldc >= K, "row stride smaller than the row");
    HIP_TRY(dn_gemm_tn_idx(A, lda, B, ldb, nullptr, C, ldc, dbias, M, N, K, workspace, accumulate, OP_ST));
    return DAGNET_OK;
}
// Weight gradient of one layer accumulated over many calls (time steps) with ONE final reduction: stage 1 adds this call's
// split partial sums to `part` (first != 0: overwrites them), dagnet_op_gemm_tn_finish reduces them into C / dbias.
// B = NULL with bidx: one-hot second operand (dagnet_op_onehot_grad).  part: dagnet_op_gemm_tn_workspace(M, N, K) floats,
// owned by the caller for the whole accumulation; M, N, K must be the same in every call.
extern "C" int dagnet_op_gemm_tn_partial(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, int32_t want_bias,
                                         int32_t M, int32_t N, int32_t K, float *part, int32_t first, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (N == 0 || K == 0) return DAGNET_OK;
    OP_NEED(A && (B || bidx) && part, "null operand");
    HIP_TRY(dn_gemm_tn_staged(A, lda, B, ldb, bidx, nullptr, 0, nullptr, want_bias, M, N, K, part, 0, 1, first ? 0 : 1, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_gemm_tn_finish(const float *part, float *C, int64_t ldc, float *dbias, int32_t M, int32_t N, int32_t K,
                                        int32_t accumulate, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (N == 0 || K == 0) return DAGNET_OK;
    OP_NEED(part && C && ldc >= K, "bad argument");
    HIP_TRY(dn_gemm_tn_staged(nullptr, 0, nullptr, 0, nullptr, C, ldc, dbias, dbias != nullptr, M, N, K, const_cast<float *>(part),
                              accumulate, 2, 0, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_act_bwd(int64_t M, int32_t N, const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp,
                                 int32_t act, float slope, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && (M * N == 0 || (dY && Y && dP)), "bad argument");
    HIP_TRY(dn_t_act_bwd(M, N, dY, ldd, Y, ldy, dP, ldp, act, slope, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_copy2d(int64_t M, int32_t N, const float *src, int64_t lds, float *dst, int64_t ldd, int32_t accumulate,
                                void *stream) {
    OP_NEED(M >= 0 && N >= 0 && (M * N == 0 || (src && dst)), "bad argument");
    HIP_TRY(dn_t_copy2d(M, N, src, lds, dst, ldd, accumulate, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_onehot_grad(int64_t M, int32_t N, int32_t G, const float *dP, int64_t ldp, const int32_t *gidx, float *dW,
                                     int64_t ldw, float *workspace, int32_t accumulate, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && G >= 0 && M <= 0x7fffffffLL && (M * N == 0 || (dP && gidx && dW && workspace)), "bad argument");
    if (M * N == 0 || G == 0) return DAGNET_OK;
    // dW[n][g] = sum_m dP[m][n] * onehot(gidx[m])[g]: the weight-gradient GEMM with a generated one-hot operand
    HIP_TRY(dn_gemm_tn_idx(dP, ldp, nullptr, 0, gidx, dW, ldw, nullptr, (int)M, N, G, workspace, accumulate, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_softmax_fwd(int64_t M, int32_t N, const float *X, int64_t ldx, float *Y, int64_t ldy, void *stream) {
    OP_NEED(M >= 0 && N >= 1 && (M == 0 || (X && Y)), "bad argument");
    HIP_TRY(dn_t_softmax_fwd(M, N, X, ldx, Y, ldy, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_softmax_bwd(int64_t M, int32_t N, const float *Y, int64_t ldy, const float *dY, int64_t ldd, float *dX, int64_t ldx,
                                     void *stream) {
    OP_NEED(M >= 0 && N >= 1 && (M == 0 || (Y && dY && dX)), "bad argument");
    HIP_TRY(dn_t_softmax_bwd(M, N, Y, ldy, dY, ldd, dX, ldx, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_gru_fwd(int64_t M, int32_t R, const float *gi, const float *gh, const float *h, float *hn, float *gates,
                                 void *stream) {
    OP_NEED(M >= 0 && R >= 1 && (M == 0 || (gi && gh && h && hn && gates)), "bad argument");
    HIP_TRY(dn_t_gru_fwd(M, R, gi, gh, h, hn, gates, OP_ST));
    return DAGNET_OK;
}
// Which of the fused training kernels are in use (bit 0: GRU layer forward, 1: dense blocks forward, 2: dense blocks backward).
// All on unless DAGNET_NO_GRU_FUSED / DAGNET_NO_MLP_FUSED / DAGNET_NO_MLP_BWD_FUSED is set; dagnet_op_set_fused changes it at run
// time (the tests compare a training step with and without them).
static int fused_default() {
    return (getenv("DAGNET_NO_GRU_FUSED") ? 0 : 1) | (getenv("DAGNET_NO_MLP_FUSED") ? 0 : 2) | (getenv("DAGNET_NO_MLP_BWD_FUSED") ? 0 : 4) |
           (getenv("DAGNET_NO_BF16X3") ? 0 : 8);
}
static std::atomic<int> g_fused{-1};
static int fused_mask() {
    int m = g_fused.load(std::memory_order_relaxed);
    if (m < 0) {
        m = fused_default();
        g_fused.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" int dn_gemm_set_bf16x3(int on);      // gemm_f32.hip: bit 3, split-bf16 products in the wide (N > 96) layers
extern "C" int dagnet_op_set_fused(int32_t mask) {
    const int before = fused_mask();
    // bit 4 (never part of the default): two fp16 terms in EVERY wide layer of dagnet_op_gemm - tests and microbenchmarks of that form
    const int now = mask < 0 ? fused_default() : (mask & 31);
    g_fused.store(now, std::memory_order_relaxed);
    dn_gemm_set_bf16x3(((now >> 3) & 1) ? (((now >> 4) & 1) ? 2 : 1) : 0);
    return before;
}
extern "C" hipError_t dn_gru_fused_fwd(int64_t M, int R, int npieces, const float *x0, int64_t ldx0, const float *x1, int64_t ldx1,
                                       const float *h, int64_t ldh, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                                       int64_t ldwhh, const float *bhh, float *hn, int64_t ldhn, float *gates, int64_t ldg, float *gh,
                                       int64_t ldgh, const float *pack, hipStream_t st);
extern "C" int64_t dn_gru_pack_floats(int npieces);
extern "C" hipError_t dn_gru_pack(int R, int npieces, const float *Wih, int64_t ldwih, const float *bih, const float *Whh, int64_t ldwhh,
                                  const float *bhh, float *out, hipStream_t st);
// -> floats of the packed image of a GRU layer that dagnet_op_gru_layer_fwd can take (0: this shape has no fused kernel)
extern "C" int64_t dagnet_op_gru_pack_floats(int32_t R, int32_t K0, int32_t K1) {
    return (R == 64 && K0 == 64 && (K1 == 0 || K1 == 64)) ? dn_gru_pack_floats(K1 ? 2 : 1) : 0;
}
extern "C" int dagnet_op_gru_pack(int32_t R, int32_t K0, int32_t K1, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                                  int64_t ldwhh, const float *bhh, float *out, void *stream) {
    OP_NEED(dagnet_op_gru_pack_floats(R, K0, K1) > 0 && Wih && Whh && bih && bhh && out, "no fused GRU kernel for this shape");
    HIP_TRY(dn_gru_pack(R, K1 ? 2 : 1, Wih, ldwih, bih, Whh, ldwhh, bhh, out, OP_ST));
    return DAGNET_OK;
}
// One GRU layer step from its inputs.  The fused kernel (csrc/gru_fused.hip: rnn_dim 64, one or two 64-wide input pieces, >= 2048
// rows) leaves gi untouched and writes only the n-gate columns of gh; any other shape runs as products + cell and fills both.
extern "C" int dagnet_op_gru_layer_fwd(int64_t M, int32_t R, const float *x0, int64_t ldx0, int32_t K0, const float *x1, int64_t ldx1,
                                       int32_t K1, const float *h, int64_t ldh, const float *Wih, int64_t ldwih, const float *bih,
                                       const float *Whh, int64_t ldwhh, const float *bhh, float *gi, float *gh, float *hn, float *gates,
                                       const float *pack, void *stream) {
    OP_NEED(M >= 0 && R >= 1 && K0 >= 1 && K1 >= 0 && M < 0x7fffffffLL, "bad argument");
    if (M == 0) return DAGNET_OK;
    OP_NEED(x0 && (K1 == 0 || x1) && h && Wih && Whh && gi && gh && hn && gates, "null operand");
    const bool no_fused = !(fused_mask() & 1);
    if (!no_fused && K0 == 64 && (K1 == 0 || K1 == 64)) {
        const hipError_t e = dn_gru_fused_fwd(M, R, K1 ? 2 : 1, x0, ldx0, x1, ldx1, h, ldh, Wih, ldwih, bih, Whh, ldwhh, bhh, hn, R, gates,
                                              3 * R, gh, 3 * R, pack, OP_ST);
        if (e == hipSuccess) return DAGNET_OK;
        if (e != hipErrorNotSupported) HIP_TRY(e);
    }
    HIP_TRY(dn_gemm(x0, ldx0, Wih, ldwih, 0, K1 ? nullptr : bih, gi, 3 * R, (int)M, 3 * R, K0, 0, 0.f, 0, nullptr, nullptr, 0, 0, OP_ST));
    if (K1)
        HIP_TRY(dn_gemm(x1, ldx1, Wih + K0, ldwih, 0, bih, gi, 3 * R, (int)M, 3 * R, K1, 0, 0.f, 1, nullptr, nullptr, 0, 0, OP_ST));
    HIP_TRY(dn_gemm(h, ldh, Whh, ldwhh, 0, bhh, gh, 3 * R, (int)M, 3 * R, R, 0, 0.f, 0, nullptr, nullptr, 0, 0, OP_ST));
    OP_NEED(ldh == R, "the unfused GRU cell needs a contiguous hidden state");
    HIP_TRY(dn_t_gru_fwd(M, R, gi, gh, h, hn, gates, OP_ST));
    return DAGNET_OK;
}
extern "C" int64_t dn_gru_pack_bwd_floats(int npieces);
extern "C" hipError_t dn_gru_pack_bwd(int R, int npieces, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh, float *out,
                                      hipStream_t st);
extern "C" hipError_t dn_gru_fused_bwd(int64_t M, int R, int npieces, const float *dhn, int64_t lddhn, const float *gates, int64_t ldg,
                                       const float *gh, int64_t ldgh, const float *h, int64_t ldh, const float *pack, float *dgi,
                                       int64_t lddgi, float *dgh, int64_t lddgh, float *dx0, int64_t lddx0, int acc0, float *dx1,
                                       int64_t lddx1, int acc1, float *dh, int64_t lddh, int acch, hipStream_t st);
// -> floats of the packed image dagnet_op_gru_layer_bwd takes (0: this shape has no fused backward kernel)
extern "C" int64_t dagnet_op_gru_pack_bwd_floats(int32_t R, int32_t K0, int32_t K1) {
    return (R == 64 && K0 == 64 && (K1 == 0 || K1 == 64)) ? dn_gru_pack_bwd_floats(K1 ? 2 : 1) : 0;
}
extern "C" int dagnet_op_gru_pack_bwd(int32_t R, int32_t K0, int32_t K1, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh,
                                      float *out, void *stream) {
    OP_NEED(dagnet_op_gru_pack_bwd_floats(R, K0, K1) > 0 && Wih && Whh && out, "no fused GRU backward kernel for this shape");
    HIP_TRY(dn_gru_pack_bwd(R, K1 ? 2 : 1, Wih, ldwih, Whh, ldwhh, out, OP_ST));
    return DAGNET_OK;
}
// Backward of one GRU layer step in one kernel (csrc/gru_fused.hip gru_fused_bwd_kernel): gate gradients dgi / dgh [M,3R] (written
// for the weight-gradient products) and the input gradients dx0 / dx1 (+)= dgi W_ih[:, piece], dh (+)= dhn z + dgh W_hh (NULL: not
// wanted).  -> 1 (nothing written) when the shape has no fused kernel or bit 0 of dagnet_op_set_fused is cleared.
extern "C" int dagnet_op_gru_layer_bwd(int64_t M, int32_t R, int32_t K0, int32_t K1, const float *dhn, int64_t lddhn, const float *gates,
                                       const float *gh, const float *h, int64_t ldh, const float *pack, float *dgi, float *dgh, float *dx0,
                                       int64_t lddx0, int32_t acc0, float *dx1, int64_t lddx1, int32_t acc1, float *dh, int64_t lddh,
                                       int32_t acch, void *stream) {
    OP_NEED(M >= 0 && R >= 1 && M < 0x7fffffffLL, "bad argument");
    static const bool off = getenv("DAGNET_NO_GRU_BWD_FUSED") != nullptr;
    if (off || !(fused_mask() & 1) || !pack || !(K0 == 64 && (K1 == 0 || K1 == 64))) return 1;
    if (M == 0) return DAGNET_OK;
    OP_NEED(dhn && gates && gh && h && dgi && dgh, "null operand");
    const hipError_t e = dn_gru_fused_bwd(M, R, K1 ? 2 : 1, dhn, lddhn, gates, 3 * R, gh, 3 * R, h, ldh, pack, dgi, 3 * R, dgh, 3 * R, dx0,
                                          lddx0, acc0, dx1, lddx1, acc1, dh, lddh, acch, OP_ST);
    if (e == hipErrorNotSupported) return 1;
    HIP_TRY(e);
    return DAGNET_OK;
}
// y1 = act1([x_0 | x_1 | x_2 | one-hot(gidx)] W1^T + b1) [M,N1], y2 = act2(y1 W2^T + b2) [M,N2]; piece s sits at columns cols[s] of W1,
// the one-hot block at gcol.  One kernel for the 64 / 64 blocks of the models (csrc/mlp_fused.hip), layer by layer otherwise.
extern "C" int dagnet_op_mlp2_fwd(int64_t M, int32_t npieces, const float *const *x, const int64_t *ldx, const int32_t *K,
                                  const int32_t *cols, const float *W1, int64_t ldw1, const float *b1, int32_t N1, const int32_t *gidx,
                                  int32_t gcol, int32_t act1, float slope1, const float *W2, int64_t ldw2, const float *b2, int32_t N2,
                                  int32_t act2, float slope2, float *y1, int64_t ldy1, float *y2, int64_t ldy2, void *stream) {
    OP_NEED(W2 && y2 && N2 >= 1, "null operand");
    return dagnet_op_mlp_chain_fwd(M, npieces, x, ldx, K, cols, W1, ldw1, b1, N1, gidx, gcol, act1, slope1, W2, ldw2, b2, N2, act2, slope2, y1,
                                   ldy1, y2, ldy2, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, stream);
}
extern "C" hipError_t dn_mlp_chain_fused(int64_t M, int npieces, const float *const *x, const int64_t *ldx, const int32_t *K,
                                         const int32_t *cols, const float *W1, int64_t ldw1, const float *b1, int N1, const int32_t *gidx,
                                         int gcol, int act1, float slope1, const float *W2, int64_t ldw2, const float *b2, int N2, int act2,
                                         float slope2, float *y1, int64_t ldy1, float *y2, int64_t ldy2, const float *Wm, int64_t ldwm,
                                         const float *bm, const float *Wl, int64_t ldwl, const float *bl, int Z, float *mu, int64_t ldmu,
                                         float *lv, int64_t ldlv, hipStream_t st);
// dagnet_op_mlp2_fwd with optional layers: W2 == NULL stops after y1; Wm / Wl != NULL append the two heads mu = y2 Wm^T + bm and
// lv = y2 Wl^T + bl ([M,Z] each, no activation).
extern "C" int dagnet_op_mlp_chain_fwd(int64_t M, int32_t npieces, const float *const *x, const int64_t *ldx, const int32_t *K,
                                       const int32_t *cols, const float *W1, int64_t ldw1, const float *b1, int32_t N1, const int32_t *gidx,
                                       int32_t gcol, int32_t act1, float slope1, const float *W2, int64_t ldw2, const float *b2, int32_t N2,
                                       int32_t act2, float slope2, float *y1, int64_t ldy1, float *y2, int64_t ldy2, const float *Wm,
                                       int64_t ldwm, const float *bm, const float *Wl, int64_t ldwl, const float *bl, int32_t Z, float *mu,
                                       int64_t ldmu, float *lv, int64_t ldlv, void *stream) {
    OP_NEED(M >= 0 && M < 0x7fffffffLL && npieces >= 1 && npieces <= 3 && N1 >= 1 && x && ldx && K && cols, "bad argument");
    if (M == 0) return DAGNET_OK;
    OP_NEED(W1 && y1 && (!W2 || (y2 && N2 >= 1)), "null operand");
    OP_NEED((!Wm && !Wl) || (W2 && Wm && Wl && mu && lv && Z >= 1), "heads need both weight matrices, both outputs and the second layer");
    for (int i = 0; i < npieces; ++i) OP_NEED(x[i] && K[i] >= 1, "null or empty input piece");
    const bool no_fused = !(fused_mask() & 2);
    if (!no_fused) {
        const hipError_t e = dn_mlp_chain_fused(M, npieces, x, ldx, K, cols, W1, ldw1, b1, N1, gidx, gcol, act1, slope1, W2, ldw2, b2, N2, act2,
                                                slope2, y1, ldy1, y2, ldy2, Wm, ldwm, bm, Wl, ldwl, bl, Z, mu, ldmu, lv, ldlv, OP_ST);
        if (e == hipSuccess) return DAGNET_OK;
        if (e != hipErrorNotSupported) HIP_TRY(e);
    }
    for (int i = 0; i < npieces; ++i) {
        const bool fin = i == npieces - 1;
        HIP_TRY(dn_gemm(x[i], ldx[i], W1 + cols[i], ldw1, 0, fin ? b1 : nullptr, y1, ldy1, (int)M, N1, K[i], fin ? act1 : 0, slope1, i > 0,
                        fin && gidx ? W1 + gcol : nullptr, fin ? gidx : nullptr, 1, ldw1, OP_ST));
    }
    if (!W2) return DAGNET_OK;
    HIP_TRY(dn_gemm(y1, ldy1, W2, ldw2, 0, b2, y2, ldy2, (int)M, N2, N1, act2, slope2, 0, nullptr, nullptr, 0, 0, OP_ST));
    if (Wm) {
        HIP_TRY(dn_gemm(y2, ldy2, Wm, ldwm, 0, bm, mu, ldmu, (int)M, Z, N2, 0, 0.f, 0, nullptr, nullptr, 0, 0, OP_ST));
        HIP_TRY(dn_gemm(y2, ldy2, Wl, ldwl, 0, bl, lv, ldlv, (int)M, Z, N2, 0, 0.f, 0, nullptr, nullptr, 0, 0, OP_ST));
    }
    return DAGNET_OK;
}
extern "C" hipError_t dn_mlp2_bwd_fused(int64_t M, const float *dy2, int64_t lddy2, const float *dmu, int64_t lddmu, const float *dlv,
                                        int64_t lddlv, int Z, const float *Wm, int64_t ldwm, const float *Wl, int64_t ldwl,
                                        const float *y2, int64_t ldy2, const float *y1, int64_t ldy1, const float *W2, int64_t ldw2,
                                        int act2, float slope2, const float *W1, int64_t ldw1, int act1, float slope1, float *dp2,
                                        int64_t lddp2, float *dp1, int64_t lddp1, float *dxA, int64_t lddxA, int cA, int KA, int accA,
                                        float *dxB, int64_t lddxB, int cB, int KB, int accB, hipStream_t st);
// Backward of a two-layer block from the gradients at its heads (dy2 = NULL) or at its output: dpre2, dpre1 (the operands of the
// layers' weight gradients) and the input gradients of up to two pieces.  -> DAGNET_OK, or 1 when this shape has no fused kernel
// (nothing written: the caller runs the backward layer by layer).
extern "C" int dagnet_op_mlp2_bwd(int64_t M, const float *dy2, int64_t lddy2, const float *dmu, int64_t lddmu, const float *dlv,
                                  int64_t lddlv, int32_t Z, const float *Wm, int64_t ldwm, const float *Wl, int64_t ldwl, const float *y2,
                                  int64_t ldy2, const float *y1, int64_t ldy1, const float *W2, int64_t ldw2, int32_t act2, float slope2,
                                  const float *W1, int64_t ldw1, int32_t act1, float slope1, float *dp2, int64_t lddp2, float *dp1,
                                  int64_t lddp1, float *dxA, int64_t lddxA, int32_t cA, int32_t KA, int32_t accA, float *dxB,
                                  int64_t lddxB, int32_t cB, int32_t KB, int32_t accB, void *stream) {
    OP_NEED(M >= 0, "bad argument");
    if (!(fused_mask() & 4) || M == 0) return 1;
    const hipError_t e = dn_mlp2_bwd_fused(M, dy2, lddy2, dmu, lddmu, dlv, lddlv, Z, Wm, ldwm, Wl, ldwl, y2, ldy2, y1, ldy1, W2, ldw2, act2,
                                           slope2, W1, ldw1, act1, slope1, dp2, lddp2, dp1, lddp1, dxA, lddxA, cA, KA, accA, dxB, lddxB, cB,
                                           KB, accB, OP_ST);
    if (e == hipErrorNotSupported) return 1;
    HIP_TRY(e);
    return DAGNET_OK;
}
extern "C" hipError_t dn_linear_pieces_bwd_fused(int64_t M, const float *dy, int64_t lddy, const float *W, int64_t ldw, float *dxA,
                                                 int64_t lddxA, int cA, int KA, int accA, float *dxB, int64_t lddxB, int cB, int KB, int accB,
                                                 hipStream_t st);
// Input gradients of one Linear (64 outputs, no activation) over two input pieces from one pass over dy.  -> DAGNET_OK, or 1 when
// the shape has no fused kernel (nothing written).
extern "C" int dagnet_op_linear_pieces_bwd(int64_t M, const float *dy, int64_t lddy, const float *W, int64_t ldw, float *dxA, int64_t lddxA,
                                           int32_t cA, int32_t KA, int32_t accA, float *dxB, int64_t lddxB, int32_t cB, int32_t KB,
                                           int32_t accB, void *stream) {
    OP_NEED(M >= 0, "bad argument");
    if (!(fused_mask() & 4) || M == 0) return 1;
    const hipError_t e = dn_linear_pieces_bwd_fused(M, dy, lddy, W, ldw, dxA, lddxA, cA, KA, accA, dxB, lddxB, cB, KB, accB, OP_ST);
    if (e == hipErrorNotSupported) return 1;
    HIP_TRY(e);
    return DAGNET_OK;
}
extern "C" int dagnet_op_gru_bwd(int64_t M, int32_t R, const float *dhn, const float *gates, const float *gh, const float *h, float *dgi,
                                 float *dgh, float *dh, void *stream) {
    OP_NEED(M >= 0 && R >= 1 && (M == 0 || (dhn && gates && gh && h && dgi && dgh && dh)), "bad argument");
    HIP_TRY(dn_t_gru_bwd(M, R, dhn, gates, gh, h, dgi, dgh, dh, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_vae_fwd(int64_t M, int32_t Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                                 const float *eps, float *z, float *kld_row, void *stream) {
    OP_NEED(M >= 0 && Z >= 1 && (M == 0 || (mu_e && lv_e && mu_p && lv_p && eps && z && kld_row)), "bad argument");
    HIP_TRY(dn_t_vae_fwd(M, Z, mu_e, lv_e, mu_p, lv_p, eps, z, kld_row, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_vae_bwd(int64_t M, int32_t Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                                 const float *eps, const float *dz, const float *g_kld, float scale, float *dmu_e, float *dlv_e,
                                 float *dmu_p, float *dlv_p, void *stream) {
    OP_NEED(M >= 0 && Z >= 1 && (M == 0 || (mu_e && lv_e && mu_p && lv_p && eps && dmu_e && dlv_e && dmu_p && dlv_p)), "bad argument");
    HIP_TRY(dn_t_vae_bwd(M, Z, mu_e, lv_e, mu_p, lv_p, eps, dz, g_kld, scale, dmu_e, dlv_e, dmu_p, dlv_p, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_nll_fwd(int64_t M, int32_t X, const float *mu, const float *lv, const float *x, float *nll_row, void *stream) {
    OP_NEED(M >= 0 && X >= 1 && (M == 0 || (mu && lv && x && nll_row)), "bad argument");
    HIP_TRY(dn_t_nll_fwd(M, X, mu, lv, x, nll_row, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_nll_bwd(int64_t M, int32_t X, const float *mu, const float *lv, const float *x, const float *g, float scale,
                                 float *dmu, float *dlv, void *stream) {
    OP_NEED(M >= 0 && X >= 1 && (M == 0 || (mu && lv && x && g && dmu && dlv)), "bad argument");
    HIP_TRY(dn_t_nll_bwd(M, X, mu, lv, x, g, scale, dmu, dlv, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_ce_fwd(int64_t M, const float *p, int64_t ldp, const int32_t *goal, float *ce_row, void *stream) {
    OP_NEED(M >= 0 && (M == 0 || (p && goal && ce_row)), "bad argument");
    HIP_TRY(dn_t_ce_fwd(M, p, ldp, goal, ce_row, OP_ST));
    return DAGNET_OK;
}
extern "C" hipError_t dn_t_softmax_ce_bwd(int64_t M, int G, const float *p, int64_t ldp, const int32_t *goal, const float *g, float *dx,
                                          int64_t ldx, hipStream_t st);
extern "C" int dagnet_op_softmax_ce_bwd(int64_t M, int32_t G, const float *p, int64_t ldp, const int32_t *goal, const float *g, float *dx,
                                        int64_t ldx, void *stream) {
    OP_NEED(M >= 0 && G >= 1 && (M == 0 || (p && goal && g && dx)), "bad argument");
    HIP_TRY(dn_t_softmax_ce_bwd(M, G, p, ldp, goal, g, dx, ldx, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_ce_bwd(int64_t M, int32_t G, const int32_t *goal, const float *g, float *dp, int64_t ldp, void *stream) {
    OP_NEED(M >= 0 && G >= 1 && (M == 0 || (goal && g && dp)), "bad argument");
    HIP_TRY(dn_t_ce_bwd(M, G, goal, g, dp, ldp, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_sum(int64_t n, const float *x, float scale, float *out, int32_t accumulate, void *stream) {
    OP_NEED(n >= 0 && out && (n == 0 || x), "bad argument");
    HIP_TRY(dn_t_sum(n, x, scale, out, accumulate, OP_ST));
    return DAGNET_OK;
}
extern "C" int64_t dagnet_op_col_workspace(int64_t M, int32_t N) { return dn_t_col_workspace(M, N); }
extern "C" int dagnet_op_colsum(int64_t M, int32_t N, const float *X, int64_t ldx, float *out, int32_t accumulate, float *workspace,
                                void *stream) {
    OP_NEED(M >= 0 && N >= 0 && (N == 0 || (out && workspace && (M == 0 || X))), "bad argument");
    HIP_TRY(dn_t_colsum(M, N, X, ldx, out, accumulate, workspace, OP_ST));
    return DAGNET_OK;
}
extern "C" int dn_t_colsum_static_max_cols(void);
extern "C" hipError_t dn_t_colsum_static(int64_t, int, const float *, int64_t, float *, int, hipStream_t);
extern "C" int dagnet_op_colsum_static(int64_t M, int32_t N, const float *X, int64_t ldx, float *out, int32_t accumulate, void *stream) {
    OP_NEED(M >= 0 && N >= 0 && (N == 0 || (out && (M == 0 || X))), "bad argument");
    if (N > dn_t_colsum_static_max_cols())
        return fail(DAGNET_ERR_CAPACITY, "dagnet_op_colsum_static: %d columns exceed the static scratch (at most %d); use dagnet_op_colsum "
                    "with a caller workspace", N, dn_t_colsum_static_max_cols());
    HIP_TRY(dn_t_colsum_static(M, N, X, ldx, out, accumulate, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_fwd(int64_t M, int32_t N, const float *X, int64_t ldx, const float *gamma, const float *beta, float eps,
                                float momentum, int32_t tanh_out, float *Y, int64_t ldy, float *stat, float *run_mean, float *run_var,
                                float *tmp, float *workspace, void *stream) {
    OP_NEED(M >= 1 && N >= 1 && X && gamma && beta && Y && stat && tmp && workspace, "bad argument");
    OP_NEED((run_mean == nullptr) == (run_var == nullptr), "running_mean and running_var go together");
    HIP_TRY(dn_t_bn_fwd(M, N, X, ldx, gamma, beta, eps, momentum, tanh_out, Y, ldy, stat, run_mean, run_var, tmp, workspace, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_bwd(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *dOut,
                                int64_t ldd, const float *Out, int64_t ldo, int32_t tanh_out, float *dX, int64_t ldg, float *dgamma,
                                float *dbeta, int32_t accumulate, float *sums, float *workspace, void *stream) {
    OP_NEED(M >= 1 && N >= 1 && X && stat && gamma && dOut && dX && dgamma && dbeta && sums && workspace && (!tanh_out || Out),
            "bad argument");
    HIP_TRY(dn_t_bn_bwd(M, N, X, ldx, stat, gamma, dOut, ldd, Out, ldo, tanh_out, dX, ldg, dgamma, dbeta, accumulate, sums, workspace, OP_ST));
    return DAGNET_OK;
}
extern "C" hipError_t dn_t_bn_moments(int64_t, int, const float *, int64_t, float *, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_bn_merge(int, int, const float *, float, float, float *, float *, float *, float *, hipStream_t);
extern "C" hipError_t dn_t_bn_apply(int64_t, int, const float *, int64_t, const float *, const float *, const float *, int, float *, int64_t, hipStream_t);
extern "C" hipError_t dn_t_bn_bwd_sums(int64_t, int, const float *, int64_t, const float *, const float *, int64_t, const float *, int64_t, int,
                                       float *, float *, hipStream_t);
extern "C" hipError_t dn_t_bn_bwd_apply(int64_t, int64_t, int, const float *, int64_t, const float *, const float *, const float *, int64_t,
                                        const float *, int64_t, int, const float *, float *, int64_t, hipStream_t);
extern "C" int dagnet_op_bn_moments(int64_t M, int32_t N, const float *X, int64_t ldx, float *mom, float *tmp, float *workspace, void *stream) {
    OP_NEED(M >= 0 && N >= 1 && mom && tmp && workspace && (M == 0 || X), "bad argument");
    HIP_TRY(dn_t_bn_moments(M, N, X, ldx, mom, tmp, workspace, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_merge(int32_t N, int32_t n_parts, const float *parts, float eps, float momentum, float *stat, float *run_mean,
                                  float *run_var, float *total, void *stream) {
    OP_NEED(N >= 1 && n_parts >= 1 && parts && stat, "bad argument");
    OP_NEED((run_mean == nullptr) == (run_var == nullptr), "running_mean and running_var go together");
    HIP_TRY(dn_t_bn_merge(N, n_parts, parts, eps, momentum, stat, run_mean, run_var, total, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_apply(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *beta,
                                  int32_t tanh_out, float *Y, int64_t ldy, void *stream) {
    OP_NEED(M >= 0 && N >= 1 && stat && gamma && beta && (M == 0 || (X && Y)), "bad argument");
    HIP_TRY(dn_t_bn_apply(M, N, X, ldx, stat, gamma, beta, tanh_out, Y, ldy, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_bwd_sums(int64_t M, int32_t N, const float *X, int64_t ldx, const float *stat, const float *dOut, int64_t ldd,
                                     const float *Out, int64_t ldo, int32_t tanh_out, float *sums, float *workspace, void *stream) {
    OP_NEED(M >= 0 && N >= 1 && stat && sums && workspace && (M == 0 || (X && dOut && (!tanh_out || Out))), "bad argument");
    HIP_TRY(dn_t_bn_bwd_sums(M, N, X, ldx, stat, dOut, ldd, Out, ldo, tanh_out, sums, workspace, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_bn_bwd_apply(int64_t M, int64_t Mtot, int32_t N, const float *X, int64_t ldx, const float *stat, const float *gamma,
                                      const float *dOut, int64_t ldd, const float *Out, int64_t ldo, int32_t tanh_out, const float *sums,
                                      float *dX, int64_t ldg, void *stream) {
    OP_NEED(M >= 0 && Mtot >= M && Mtot >= 1 && N >= 1 && stat && gamma && sums && (M == 0 || (X && dOut && dX && (!tanh_out || Out))), "bad argument");
    HIP_TRY(dn_t_bn_bwd_apply(M, Mtot, N, X, ldx, stat, gamma, dOut, ldd, Out, ldo, tanh_out, sums, dX, ldg, OP_ST));
    return DAGNET_OK;
}
static int op_plan(const dagnet_plan *p) {
    if (!p) return fail(DAGNET_ERR_INVALID, "null plan");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev != p->device)
        return fail(DAGNET_ERR_INVALID, "plan belongs to device %d but device %d is current", p->device, dev);
    return DAGNET_OK;
}
// adj_ld: 0 = compact block-diagonal adjacency (dagnet_adjacency layout), N = dense [N, N]
static int op_adj(const dagnet_plan *p, int64_t adj_ld, const int64_t **off) {
    if (adj_ld != 0 && adj_ld != p->N) return fail(DAGNET_ERR_INVALID, "adj_ld must be 0 (compact) or N");
    *off = adj_ld ? p->d_off_dense : p->d_off_compact;
    return DAGNET_OK;
}
extern "C" int dagnet_op_gat_attn_fwd(const dagnet_plan *p, const float *adj, int64_t adj_ld, const float *Wh, int64_t ldw, const float *a, int32_t H,
                                      int32_t F, float alpha, int32_t elu, float *Out, int64_t ldo, void *stream) {
    if (int rc = op_plan(p)) return rc;
    OP_NEED(adj && Wh && a && Out && H >= 1 && F >= 1, "bad argument");
    if (dn_t_gat_lds_bytes(p->AS, F, 0) > 64 * 1024)
        return fail(DAGNET_ERR_CAPACITY, "graph attention with %d agents x %d features does not fit the LDS tile of the training kernels", p->AS, F);
    const int64_t *off;
    if (int rc = op_adj(p, adj_ld, &off)) return rc;
    HIP_TRY(dn_t_gat_attn_fwd(p->d_scene_ptr, off, p->S, p->AS, adj_ld, adj, Wh, ldw, a, H, F, alpha, elu, Out, ldo, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_gat_attn_bwd(const dagnet_plan *p, const float *adj, int64_t adj_ld, const float *Wh, int64_t ldw, const float *a, int32_t H,
                                      int32_t F, float alpha, int32_t elu, const float *Out, int64_t ldo, const float *dOut, int64_t ldd,
                                      float *dWh, int64_t ldg, float *da_part, void *stream) {
    if (int rc = op_plan(p)) return rc;
    OP_NEED(adj && Wh && a && dOut && dWh && da_part && (!elu || Out) && H >= 1 && F >= 1, "bad argument");
    if (dn_t_gat_lds_bytes(p->AS, F, 1) > 64 * 1024)
        return fail(DAGNET_ERR_CAPACITY, "graph attention backward with %d agents x %d features does not fit the LDS tile of the training kernels", p->AS, F);
    const int64_t *off;
    if (int rc = op_adj(p, adj_ld, &off)) return rc;
    HIP_TRY(dn_t_gat_attn_bwd(p->d_scene_ptr, off, p->S, p->AS, adj_ld, adj, Wh, ldw, a, H, F, alpha, elu, Out, ldo, dOut, ldd, dWh,
                              ldg, da_part, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_adj_matmul(const dagnet_plan *p, int32_t F, const float *adj, int64_t adj_ld, const float *X, int64_t ldx, const float *bias,
                                    int32_t transpose, float *Y, int64_t ldy, void *stream) {
    if (int rc = op_plan(p)) return rc;
    OP_NEED(adj && X && Y && F >= 1, "bad argument");
    const int64_t *off;
    if (int rc = op_adj(p, adj_ld, &off)) return rc;
    HIP_TRY(dn_t_adj_matmul(p->d_scene_ptr, off, p->d_row_scene, p->S, p->N, F, adj_ld, adj, X, ldx, bias, transpose, Y, ldy, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_disposition(const dagnet_plan *p, int32_t D, const float *pos, float *out, int64_t ldo, void *stream) {
    if (int rc = op_plan(p)) return rc;
    OP_NEED(pos && out && D >= 2 * p->AS, "bad argument (D must hold 2 * the largest scene)");
    HIP_TRY(dn_t_disposition(p->d_scene_ptr, p->d_off_compact, p->d_row_scene, p->S, p->N, D, pos, out, ldo, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_sumsq(int64_t n, const float *g, double *acc, void *stream) {
    OP_NEED(n >= 0 && acc && (n == 0 || g), "bad argument");
    HIP_TRY(dn_t_sumsq(n, g, acc, OP_ST));
    return DAGNET_OK;
}
extern "C" int dagnet_op_adam(int64_t n, float *p, float *g, float *m, float *v, const double *sumsq, float clip, float lr, float beta1,
                              float beta2, float eps, int32_t step, void *stream) {
    OP_NEED(n >= 0 && step >= 1 && (n == 0 || (p && g && m && v)) && (clip <= 0.f || sumsq), "bad argument");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    HIP_TRY(dn_t_adam(n, p, g, m, v, sumsq, clip, lr, beta1, beta2, eps, bc1, bc2, OP_ST));
    return DAGNET_OK;
}
