// Two dense layers of the training forward as ONE kernel: y1 = act1([x0 | x1 | x2 | one-hot(g)] W1^T + b1), y2 = act2(y1 W2^T + b2)
// with 64 hidden and 64 output features (the phi_x / phi_z / enc / prior / dec blocks of models/dagnet/model.py:60-116: Linear,
// LeakyReLU, Linear, LeakyReLU).  Layer by layer this is one launch per input piece (each re-reading and re-writing y1) plus one
// for the second layer; here a wave owns 32-row tiles end to end (the skinny-GEMM scheme of gemm_skinny.hip): the pieces' rows
// come straight from HBM as MFMA operands, and the first layer's accumulators (lane = row, register 4 a + i of tile nt <->
// feature 32 nt + 8 a + 4 lh + i) ARE the second layer's operand quads (quad g = 4 nt + a, element i) - y1 goes from the
// matrix cores back into them without leaving the registers.  Both outputs are written (the backward pass reads y1 too).
//
// MODE 1 stops after the first layer (lg_hiddens: two pieces, no activation); MODE 2 appends the two heads that read y2 (enc_mean /
// enc_logvar, prior_*, dec_*: 64 -> Z each, no activation) as a third product on the y2 accumulators, written to mu and lv.
//
// Pieces: up to three; a piece is either a multiple of 8 columns wide with 16-byte rows (float4 operand loads) or - piece 0 only -
// at most 4 columns of any layout (x_t and the disposition, 2 columns: scalar loads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>

namespace mlpf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_RELU = 3 };

__device__ f32x4 mf_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float mf_load1(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void mf_store4(f32x4 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void mf_store1(float data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");

__device__ __forceinline__ i32x4 make_rsrc(const void *base, int64_t bytes) {
    union { const void *p; int w[2]; } u;
    u.p = base;
    return i32x4{u.w[0], u.w[1], (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000};
}

__device__ __forceinline__ float act_apply(int act, float v, float slope) {
    if (act == ACT_LRELU) return fmaxf(v, v * slope);
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// A weight image of COUNT quads, a thread taking quads tid, tid + 256, ...: ALL of the thread's loads are issued before its LDS
// writes.  Written as `for (e = tid; e < n; e += 256) lds[e] = f(W[..])` the loops below compiled to one load in flight per
// iteration (load, s_waitcnt vmcnt(0), ds_write): 16 to 64 serial L2 round trips in front of every workgroup's first tile.
// load(q) is called unconditionally with q clamped to the image.
template <int COUNT, class L>
__device__ __forceinline__ void stage_quads(float *dst, int tid, L load) {
    constexpr int IT = (COUNT + 255) / 256;
    f32x4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) v[it] = load(min(tid + 256 * it, COUNT - 1));
#pragma unroll
    for (int it = 0; it < IT; ++it)
        if (COUNT % 256 == 0 || tid + 256 * it < COUNT) *reinterpret_cast<f32x4 *>(dst + (tid + 256 * it) * 4) = v[it];
}
// quad q of a transposed image [NT][8][64][4]: element i = W[k = 8 g + 4 (l >> 5) + i][c + n], n = 32 nt + (l & 31) < K (else 0);
// lanes are consecutive n: every load covers whole 128-byte rows of W
__device__ __forceinline__ f32x4 quad_t(const float *W, int64_t ld, int c, int K, int q) {
    const int l = q & 63, g = (q >> 6) & 7, nt = q >> 9, n = nt * 32 + (l & 31);
    const float *src = W + (int64_t)(8 * g + 4 * (l >> 5)) * ld + c + min(n, K - 1);
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = src[(int64_t)i * ld];
    if (n >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
    return v;
}

struct Args {
    const float *x0, *x1, *x2; int64_t ldx0, ldx1, ldx2; int K0, c0, c1, c2;      // pieces and their first columns in W1
    const float *W1; int64_t ldw1; const float *b1;                                // [64, Ktot]
    const int32_t *gidx; int gcol;                                                 // one-hot piece: y1 += W1[:, gcol + gidx[m]]
    const float *W2; int64_t ldw2; const float *b2;                                // [64, 64]
    float *y1; int64_t ldy1; float *y2; int64_t ldy2;
    int M, act1, act2; float slope1, slope2;
    // heads (MODE 2): mu = y2 Wm^T + bm, lv = y2 Wl^T + bl, Z columns each
    const float *Wm, *Wl; int64_t ldwm, ldwl; const float *bm, *bl; int Z; float *mu, *lv; int64_t ldmu, ldlv;
};

// G0, G1, G2: k-groups of 8 per piece (0: piece absent; G0 = 1 with K0 <= 4 is the narrow form).
constexpr int mlp_wpc(int KG) { return KG <= 4 ? 4 : (KG <= 8 ? 3 : 2); }      // workgroups per CU the registers allow

// MODE: 0 = two layers, 1 = the first layer only, 2 = two layers + heads (NT3 = 32-wide tiles of the stacked heads, 2 Z <= 32 NT3)
template <int G0, int G1, int G2, bool NARROW, bool GATHER, int MODE, int NT3>
__global__ __launch_bounds__(256, mlp_wpc(G0 + G1 + G2)) void mlp2_kernel(const Args p) {
    constexpr int KG = G0 + G1 + G2;
    extern __shared__ float lds[];           // w1 [2][KG][64][4] | w2 [2][8][64][4] | w3 [NT3][8][64][4] | b1, b2, b3 [64] each
    float *w1 = lds, *w2 = w1 + 2 * KG * 256, *w3 = w2 + (MODE != 1 ? 2 * 8 * 256 : 0), *bb = w3 + (MODE == 2 ? NT3 * 8 * 256 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // ---- weight images: quad (tile nt, group g, lane l) = W[nt * 32 + (l & 31)][first column of g + 4 (l >> 5) .. + 3]
    stage_quads<2 * KG * 64>(w1, tid, [&](int e) {
        const int l = e & 63, g = (e >> 6) % KG, nt = (e >> 6) / KG;
        const int col = (g < G0 ? p.c0 + 8 * g : g < G0 + G1 ? p.c1 + 8 * (g - G0) : p.c2 + 8 * (g - G0 - G1)) + 4 * (l >> 5);
        const float *src = p.W1 + (int64_t)(nt * 32 + (l & 31)) * p.ldw1 + col;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (!NARROW || g > 0 || 4 * (l >> 5) + i < p.K0) ? src[i] : 0.f;
        return v;
    });
    if constexpr (MODE != 1) {
        stage_quads<2 * 8 * 64>(w2, tid, [&](int e) {
            const int l = e & 63, g = (e >> 6) & 7, nt = e >> 9;
            const float *src = p.W2 + (int64_t)(nt * 32 + (l & 31)) * p.ldw2 + 8 * g + 4 * (l >> 5);
            return f32x4{src[0], src[1], src[2], src[3]};
        });
    }
    if constexpr (MODE == 2) {               // the two heads stacked along the output features: [mean | logvar | zeros]
        stage_quads<NT3 * 8 * 64>(w3, tid, [&](int e) {
            const int l = e & 63, g = (e >> 6) & 7, nt = e >> 9, n = nt * 32 + (l & 31), nc = min(n, 2 * p.Z - 1);
            const float *src = (nc < p.Z ? p.Wm + (int64_t)nc * p.ldwm : p.Wl + (int64_t)(nc - p.Z) * p.ldwl) + 8 * g + 4 * (l >> 5);
            f32x4 v = f32x4{src[0], src[1], src[2], src[3]};
            if (n >= 2 * p.Z) v = f32x4{0.f, 0.f, 0.f, 0.f};
            return v;
        });
    }
    if (tid < 64) {
        bb[tid] = p.b1 ? p.b1[tid] : 0.f;
        bb[64 + tid] = (MODE != 1 && p.b2) ? p.b2[tid] : 0.f;
        float b3 = 0.f;
        if (MODE == 2 && tid < 2 * p.Z) b3 = tid < p.Z ? (p.bm ? p.bm[tid] : 0.f) : (p.bl ? p.bl[tid - p.Z] : 0.f);
        bb[128 + tid] = b3;
    }
    __syncthreads();

    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const i32x4 r0 = make_rsrc(p.x0, (int64_t)p.M * p.ldx0 * 4);
    const i32x4 r1 = make_rsrc(G1 ? p.x1 : p.x0, (int64_t)p.M * (G1 ? p.ldx1 : p.ldx0) * 4);
    const i32x4 r2 = make_rsrc(G2 ? p.x2 : p.x0, (int64_t)p.M * (G2 ? p.ldx2 : p.ldx0) * 4);
    const i32x4 ry1 = make_rsrc(p.y1, (int64_t)p.M * p.ldy1 * 4), ry2 = make_rsrc(MODE != 1 ? p.y2 : p.y1, (int64_t)p.M * (MODE != 1 ? p.ldy2 : p.ldy1) * 4);
    const i32x4 rmu = make_rsrc(MODE == 2 ? p.mu : p.y1, MODE == 2 ? (int64_t)p.M * p.ldmu * 4 : 0), rlv = make_rsrc(MODE == 2 ? p.lv : p.y1, MODE == 2 ? (int64_t)p.M * p.ldlv * 4 : 0);
    const i32x4 rw = make_rsrc(p.W1, 0x7fffffffLL);
    const int ld04 = (int)p.ldx0 * 4, ld14 = (int)p.ldx1 * 4, ld24 = (int)p.ldx2 * 4, ldy14 = (int)p.ldy1 * 4, ldy24 = (int)p.ldy2 * 4;

    f32x4 cur[KG], nxt[KG];
    auto load_rows = [&](f32x4 (&d)[KG], int t) {
        const int row = t * 32 + lr;             // rows past M lie beyond the buffers: the range check returns 0
        if constexpr (NARROW) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float u = mf_load1(r0, row * ld04 + 4 * min(i, p.K0 - 1), 0, 0);
                v[i] = (lh == 0 && i < p.K0) ? u : 0.f;
            }
            d[0] = v;
        } else {
#pragma unroll
            for (int g = 0; g < G0; ++g) d[g] = mf_load4(r0, row * ld04 + 16 * lh + g * 32, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < G1; ++g) d[G0 + g] = mf_load4(r1, row * ld14 + 16 * lh + g * 32, 0, 0);
#pragma unroll
        for (int g = 0; g < G2; ++g) d[G0 + G1 + g] = mf_load4(r2, row * ld24 + 16 * lh + g * 32, 0, 0);
    };
    load_rows(cur, tile);
    while (true) {
        const int ntile = tile + stride;
        const bool more = ntile < n_tiles;
        if (more) load_rows(nxt, ntile);
        const int row = tile * 32 + lr;
        // one-hot piece: the 32 weights of this lane's (row, features) - loaded here, consumed after the MFMAs
        float gpre[GATHER ? 2 : 1][16];
        if (GATHER) {
            const int gvoff = (p.gcol + p.gidx[min(row, p.M - 1)]) * 4, ldw4 = (int)p.ldw1 * 4;
#pragma unroll
            for (int nt = 0; nt < (GATHER ? 2 : 1); ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gpre[nt][r] = mf_load1(rw, gvoff + (nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ldw4, 0, 0);
        }
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(w1 + ((nt * KG + g) * 64 + lane) * 4);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], cur[g][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], cur[g][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], cur[g][2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], cur[g][3], acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // at most two weight quads in flight
        }
        // ---- first epilogue: y1 quads in operand order (quad 4 nt + a = features 32 nt + 8 a + 4 lh .. + 3)
        f32x4 yq[8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int f0 = nt * 32 + 8 * a + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bb + f0);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[nt][4 * a + i] + b[i];
                    if (GATHER) v[i] += gpre[nt][4 * a + i];
                    v[i] = act_apply(p.act1, v[i], p.slope1);
                }
                yq[4 * nt + a] = v;
                mf_store4(v, ry1, row * ldy14 + f0 * 4, 0, 0);
            }
        if constexpr (MODE != 1) {
        // ---- second layer
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(w2 + ((nt * 8 + g) * 64 + lane) * 4);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], yq[g][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], yq[g][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], yq[g][2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], yq[g][3], acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int f0 = nt * 32 + 8 * a + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bb + 64 + f0);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_apply(p.act2, acc[nt][4 * a + i] + b[i], p.slope2);
                mf_store4(v, ry2, row * ldy24 + f0 * 4, 0, 0);
                yq[4 * nt + a] = v;
            }
        }
        if constexpr (MODE == 2) {
            // ---- heads: one product over the stacked [mean | logvar] rows; the quads of a feature range go to mu or to lv
            f32x16 ah[NT3];
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ah[nt][r] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int nt = 0; nt < NT3; ++nt) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(w3 + ((nt * 8 + g) * 64 + lane) * 4);
                    ah[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], yq[g][0], ah[nt], 0, 0, 0);
                    ah[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], yq[g][1], ah[nt], 0, 0, 0);
                    ah[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], yq[g][2], ah[nt], 0, 0, 0);
                    ah[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], yq[g][3], ah[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int ldmu4 = (int)p.ldmu * 4, ldlv4 = (int)p.ldlv * 4;
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int f0 = nt * 32 + 8 * a + 4 * lh;
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(bb + 128 + f0);
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ah[nt][4 * a + i] + b[i];
                    if ((p.Z & 3) == 0) {
                        if (f0 < p.Z) mf_store4(v, rmu, row * ldmu4 + f0 * 4, 0, 0);
                        else if (f0 < 2 * p.Z) mf_store4(v, rlv, row * ldlv4 + (f0 - p.Z) * 4, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int f = f0 + i;
                            if (f < p.Z) mf_store1(v[i], rmu, row * ldmu4 + f * 4, 0, 0);
                            else if (f < 2 * p.Z) mf_store1(v[i], rlv, row * ldlv4 + (f - p.Z) * 4, 0, 0);
                        }
                    }
                }
        }
        if (!more) break;
        tile = ntile;
#pragma unroll
        for (int g = 0; g < KG; ++g) cur[g] = nxt[g];
    }
}

template <int G0, int G1, int G2, bool NARROW, int MODE, int NT3>
static hipError_t launch(const Args &a, hipStream_t st) {
    constexpr int KG = G0 + G1 + G2;
    const size_t lds = ((size_t)2 * KG * 256 + (MODE != 1 ? 2 * 8 * 256 : 0) + (MODE == 2 ? NT3 * 8 * 256 : 0) + 192) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(mlp_wpc(KG), (160 * 1024) / (int64_t)lds));
    const int64_t blocks = std::min<int64_t>((tiles + 3) / 4, 256 * per_cu);
    if (a.gidx) hipLaunchKernelGGL((mlp2_kernel<G0, G1, G2, NARROW, true, MODE, NT3>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((mlp2_kernel<G0, G1, G2, NARROW, false, MODE, NT3>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace mlpf

// Layers present: W2 == NULL -> the first layer only; Wm != NULL -> the two heads on y2 as well.
// -> hipErrorNotSupported for anything but: 64 hidden and output features, >= 2048 rows, pieces as described above in one of the
// instantiated width combinations, 16-byte aligned output rows (the caller then runs the block layer by layer).
extern "C" hipError_t dn_mlp_chain_fused(int64_t M, int npieces, const float *const *x, const int64_t *ldx, const int32_t *K,
                                         const int32_t *cols, const float *W1, int64_t ldw1, const float *b1, int N1, const int32_t *gidx,
                                         int gcol, int act1, float slope1, const float *W2, int64_t ldw2, const float *b2, int N2, int act2,
                                         float slope2, float *y1, int64_t ldy1, float *y2, int64_t ldy2, const float *Wm, int64_t ldwm,
                                         const float *bm, const float *Wl, int64_t ldwl, const float *bl, int Z, float *mu, int64_t ldmu,
                                         float *lv, int64_t ldlv, hipStream_t st) {
    const int mode = !W2 ? 1 : (Wm ? 2 : 0);
    if (N1 != 64 || (mode != 1 && N2 != 64) || M < 2048 || npieces < 1 || npieces > 3) return hipErrorNotSupported;
    if (mode == 1 && Wm) return hipErrorNotSupported;
    if ((ldy1 & 3) || ldy1 < 64 || (reinterpret_cast<uintptr_t>(y1) & 15) || M * ldy1 * 4 >= 0x7fffffffLL) return hipErrorNotSupported;
    if (mode != 1 && ((ldy2 & 3) || ldy2 < 64 || (reinterpret_cast<uintptr_t>(y2) & 15) || M * ldy2 * 4 >= 0x7fffffffLL)) return hipErrorNotSupported;
    if (gcol < 0 || ldw1 > 4096) return hipErrorNotSupported;
    int nt3 = 1;
    if (mode == 2) {
        if (!Wl || !mu || !lv || Z < 1 || 2 * Z > 64 || ldmu < Z || ldlv < Z || M * std::max(ldmu, ldlv) * 4 >= 0x7fffffffLL) return hipErrorNotSupported;
        if ((Z & 3) == 0 && ((ldmu & 3) || (ldlv & 3) || ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(lv)) & 15)))
            return hipErrorNotSupported;
        nt3 = (2 * Z + 31) / 32;
    }
    int G[3] = {0, 0, 0};
    const bool narrow = K[0] <= 4;
    for (int i = 0; i < npieces; ++i) {
        if (!x[i] || M * ldx[i] * 4 >= 0x7fffffffLL) return hipErrorNotSupported;
        if (i == 0 && narrow) {
            G[0] = 1;
            continue;
        }
        if (K[i] % 8 || K[i] < 8 || K[i] > 64 || (ldx[i] & 3) || ldx[i] < K[i] || (reinterpret_cast<uintptr_t>(x[i]) & 15)) return hipErrorNotSupported;
        G[i] = K[i] / 8;
    }
    mlpf::Args a{x[0], npieces > 1 ? x[1] : nullptr, npieces > 2 ? x[2] : nullptr, ldx[0], npieces > 1 ? ldx[1] : 0, npieces > 2 ? ldx[2] : 0,
                 K[0], cols[0], npieces > 1 ? cols[1] : 0, npieces > 2 ? cols[2] : 0, W1, ldw1, b1, gidx, gcol, W2, ldw2, b2, y1, ldy1, y2, ldy2,
                 (int)M, act1, act2, slope1, slope2, Wm, Wl, ldwm, ldwl, bm, bl, Z, mu, lv, ldmu, ldlv};
#define MLP_CASE(A, B, C, NARROW_, MODE_, NT3_)                                                                    \
    if (G[0] == A && G[1] == B && G[2] == C && narrow == NARROW_ && mode == MODE_ && nt3 == NT3_)                 \
        return mlpf::launch<A, B, C, NARROW_, MODE_, NT3_>(a, st);
    MLP_CASE(1, 0, 0, true, 0, 1)      // phi_x: x_t
    MLP_CASE(2, 0, 0, false, 0, 1)     // phi_z, z_dim 16
    MLP_CASE(4, 0, 0, false, 0, 1)     // phi_z, z_dim 32
    MLP_CASE(8, 0, 0, false, 0, 1)     // prior: h
    MLP_CASE(8, 8, 0, false, 0, 1)     // enc: phi_x, h;  A-VRNN / VRNN dec: phi_z, h
    MLP_CASE(1, 8, 8, true, 0, 1)      // DAG-Net dec: disposition, phi_z, h
    MLP_CASE(8, 8, 0, false, 1, 1)     // lg_hiddens / lg: graph output, h (one layer)
    MLP_CASE(8, 0, 0, false, 2, 1)     // prior + heads, z_dim 16
    MLP_CASE(8, 0, 0, false, 2, 2)     // prior + heads, z_dim 32
    MLP_CASE(8, 8, 0, false, 2, 1)     // enc + heads, z_dim 16; A-VRNN / VRNN dec + heads (x_dim 2)
    MLP_CASE(8, 8, 0, false, 2, 2)     // enc + heads, z_dim 32
    MLP_CASE(1, 8, 8, true, 2, 1)      // DAG-Net dec + heads (x_dim 2)
#undef MLP_CASE
    return hipErrorNotSupported;
}

extern "C" hipError_t dn_mlp2_fused(int64_t M, int npieces, const float *const *x, const int64_t *ldx, const int32_t *K, const int32_t *cols,
                                    const float *W1, int64_t ldw1, const float *b1, int N1, const int32_t *gidx, int gcol, int act1,
                                    float slope1, const float *W2, int64_t ldw2, const float *b2, int N2, int act2, float slope2, float *y1,
                                    int64_t ldy1, float *y2, int64_t ldy2, hipStream_t st) {
    if (!W2) return hipErrorNotSupported;
    return dn_mlp_chain_fused(M, npieces, x, ldx, K, cols, W1, ldw1, b1, N1, gidx, gcol, act1, slope1, W2, ldw2, b2, N2, act2, slope2, y1, ldy1,
                              y2, ldy2, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, st);
}

// =================================================================================================
// Backward of a two-layer block (and its heads) as ONE kernel: everything between the gradients that arrive at the block -
// dmu, dlv at the heads, or dy2 at the block's output - and the gradients of its input pieces:
//     d y2 = dmu Wm + dlv Wl            (heads)          dpre2 = d y2 * act2'(y2)       -> written (weight gradient of layer 2)
//     d y1 = dpre2 W2                                    dpre1 = d y1 * act1'(y1)       -> written (weight gradient of layer 1)
//     dx_s (+)= dpre1 W1[:, c_s : c_s + K_s]             for up to two input pieces that need a gradient
// Layer by layer these are 4 - 6 launches that pass d y2, dpre2, d y1, dpre1 through HBM between them.  The same register trick as
// in the forward pass carries the chain: a product's accumulators (lane = row, 4 consecutive features per register quad) are the
// next product's operand quads; the transposed weight matrices sit in LDS in MFMA operand order.
// =================================================================================================
namespace mlpf {

struct BArgs {
    const float *dy2; int64_t lddy2;                       // G3 == 0: gradient of the block output [M, 64]
    const float *dmu, *dlv; int64_t lddmu, lddlv; int Z;   // heads: [M, Z] each
    const float *Wm, *Wl; int64_t ldwm, ldwl;              // [Z, 64]
    const float *y2, *y1; int64_t ldy2, ldy1;              // saved outputs [M, 64]
    const float *W2; int64_t ldw2;                         // [64, 64]
    const float *W1; int64_t ldw1; int cA, cB, KA, KB;     // [64, Ktot]; columns and widths of the pieces that need a gradient
    float *dp2, *dp1; int64_t lddp2, lddp1;                // [M, 64] each
    float *dxA, *dxB; int64_t lddxA, lddxB; int accA, accB;
    int M, act1, act2; float slope1, slope2;
};

__device__ __forceinline__ float act_grad(int act, float y, float slope) {
    if (act == ACT_LRELU) return y > 0.f ? 1.f : slope;
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_TANH) return 1.f - y * y;
    return 1.f;
}

// NTA, NTB: 32-column tiles of the two pieces that get a gradient (0: absent).  G3: 0 = dy2 is given; 4 / 8 = heads with
// 2 Z = 8 G3 stacked rows (Z = 16 / 32); 1 = narrow heads (2 Z <= 4: the decoder's 2 + 2 outputs, scalar loads).
template <int NTA, int NTB, int G3>
__global__ __launch_bounds__(256, 3) void mlp2_bwd_kernel(const BArgs p) {
    constexpr int G3A = G3 ? G3 : 1;
    extern __shared__ float lds[];           // w3t [2][G3][64][4] | w2t [2][8][64][4] | wAt [NTA][8][64][4] | wBt [NTB][8][64][4]
    float *w3t = lds, *w2t = w3t + (G3 ? 2 * G3 * 256 : 0), *wAt = w2t + 2 * 8 * 256, *wBt = wAt + NTA * 8 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // ---- transposed weight images: quad (tile nt, group g, lane l), element i = W[k = 8 g + 4 (l >> 5) + i][n = 32 nt + (l & 31)]
    if constexpr (G3 != 0) {
        stage_quads<2 * G3 * 64>(w3t, tid, [&](int q) {
            const int l = q & 63, g = (q >> 6) % G3, nt = (q >> 6) / G3, n = nt * 32 + (l & 31);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 8 * g + 4 * (l >> 5) + i, kc = min(k, 2 * p.Z - 1);
                const float *src = kc < p.Z ? p.Wm + (int64_t)kc * p.ldwm : p.Wl + (int64_t)(kc - p.Z) * p.ldwl;
                const float w = src[n];
                v[i] = k < 2 * p.Z ? w : 0.f;
            }
            return v;
        });
    }
    stage_quads<2 * 8 * 64>(w2t, tid, [&](int q) { return quad_t(p.W2, p.ldw2, 0, 64, q); });
    if constexpr (NTA != 0) stage_quads<(NTA ? NTA : 1) * 8 * 64>(wAt, tid, [&](int q) { return quad_t(p.W1, p.ldw1, p.cA, p.KA, q); });
    if constexpr (NTB != 0) stage_quads<(NTB ? NTB : 1) * 8 * 64>(wBt, tid, [&](int q) { return quad_t(p.W1, p.ldw1, p.cB, p.KB, q); });
    __syncthreads();

    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * 4;
    const i32x4 rdy = make_rsrc(G3 ? p.y2 : p.dy2, (int64_t)p.M * (G3 ? p.ldy2 : p.lddy2) * 4);
    const i32x4 rdm = make_rsrc(G3 ? p.dmu : p.y2, (int64_t)p.M * (G3 ? p.lddmu : p.ldy2) * 4);
    const i32x4 rdl = make_rsrc(G3 ? p.dlv : p.y2, (int64_t)p.M * (G3 ? p.lddlv : p.ldy2) * 4);
    const i32x4 ry2 = make_rsrc(p.y2, (int64_t)p.M * p.ldy2 * 4), ry1 = make_rsrc(p.y1, (int64_t)p.M * p.ldy1 * 4);
    const i32x4 rp2 = make_rsrc(p.dp2, (int64_t)p.M * p.lddp2 * 4), rp1 = make_rsrc(p.dp1, (int64_t)p.M * p.lddp1 * 4);
    const i32x4 rxa = make_rsrc(NTA ? p.dxA : p.dp1, (int64_t)p.M * (NTA ? p.lddxA : p.lddp1) * 4);
    const i32x4 rxb = make_rsrc(NTB ? p.dxB : p.dp1, (int64_t)p.M * (NTB ? p.lddxB : p.lddp1) * 4);
    const int ldy24 = (int)p.ldy2 * 4, ldy14 = (int)p.ldy1 * 4, ldp24 = (int)p.lddp2 * 4, ldp14 = (int)p.lddp1 * 4;

    for (int tile = (int)blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        const int row = tile * 32 + lr;          // rows past M: loads return 0 and stores are dropped (range check)
        f32x4 gq[G3A], yq[8], zq[8];
        // ---- operands of the first product and the saved outputs (same lanes, same features as the accumulators)
        if constexpr (G3 == 1) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = mf_load1(rdm, (row * (int)p.lddmu + min(i, p.Z - 1)) * 4, 0, 0);
                const float b = mf_load1(rdl, (row * (int)p.lddlv + min(max(i - p.Z, 0), p.Z - 1)) * 4, 0, 0);
                v[i] = lh == 0 ? (i < p.Z ? a : (i < 2 * p.Z ? b : 0.f)) : 0.f;
            }
            gq[0] = v;
        } else if constexpr (G3 > 1) {
#pragma unroll
            for (int g = 0; g < G3 / 2; ++g) {
                gq[g] = mf_load4(rdm, row * (int)p.lddmu * 4 + 16 * lh + g * 32, 0, 0);
                gq[G3 / 2 + g] = mf_load4(rdl, row * (int)p.lddlv * 4 + 16 * lh + g * 32, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            yq[q] = mf_load4(ry2, row * ldy24 + 16 * lh + q * 32, 0, 0);
            zq[q] = mf_load4(ry1, row * ldy14 + 16 * lh + q * 32, 0, 0);
        }
        f32x16 acc[2];
        f32x4 pq[8];
        if constexpr (G3 == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) pq[q] = mf_load4(rdy, row * (int)p.lddy2 * 4 + 16 * lh + q * 32, 0, 0);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
            for (int g = 0; g < G3A; ++g) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(w3t + ((nt * G3A + g) * 64 + lane) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], gq[g][i], acc[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) pq[q][i] = acc[q >> 2][4 * (q & 3) + i];
        }
        // ---- dpre2 = d y2 * act2'(y2)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pq[q][i] *= act_grad(p.act2, yq[q][i], p.slope2);
            mf_store4(pq[q], rp2, row * ldp24 + 16 * lh + q * 32, 0, 0);
        }
        // ---- d y1 = dpre2 W2, dpre1 = d y1 * act1'(y1)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(w2t + ((nt * 8 + g) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], pq[g][i], acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pq[q][i] = acc[q >> 2][4 * (q & 3) + i] * act_grad(p.act1, zq[q][i], p.slope1);
            mf_store4(pq[q], rp1, row * ldp14 + 16 * lh + q * 32, 0, 0);
        }
        // ---- input gradients of the pieces
        auto piece = [&](auto nt_tag, const float *wt, const i32x4 rx, const int ldx4, const int K, const int accum) {
            constexpr int NTP = decltype(nt_tag)::value;
            f32x16 ax[NTP ? NTP : 1];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    if (accum && nt * 32 + 8 * a + 4 * lh < K) c = mf_load4(rx, row * ldx4 + (nt * 32 + 8 * a + 4 * lh) * 4, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ax[nt][4 * a + i] = c[i];
                }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wt + ((nt * 8 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ax[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], pq[g][i], ax[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int f0 = nt * 32 + 8 * a + 4 * lh;
                    if (f0 < K) mf_store4(f32x4{ax[nt][4 * a], ax[nt][4 * a + 1], ax[nt][4 * a + 2], ax[nt][4 * a + 3]}, rx, row * ldx4 + f0 * 4, 0, 0);
                }
        };
        if constexpr (NTA > 0) piece(std::integral_constant<int, NTA>{}, wAt, rxa, (int)p.lddxA * 4, p.KA, p.accA);
        if constexpr (NTB > 0) piece(std::integral_constant<int, NTB>{}, wBt, rxb, (int)p.lddxB * 4, p.KB, p.accB);
    }
}

template <int NTA, int NTB, int G3>
static hipError_t launch_bwd(const BArgs &a, hipStream_t st) {
    const size_t lds = ((size_t)(G3 ? 2 * G3 * 256 : 0) + 2 * 8 * 256 + (NTA + NTB) * 8 * 256) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    const int64_t blocks = std::min<int64_t>((tiles + 3) / 4, 256 * 3);
    hipLaunchKernelGGL((mlp2_bwd_kernel<NTA, NTB, G3>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace mlpf

// -> hipErrorNotSupported when no instantiation fits (the caller runs the backward layer by layer; nothing has been written).
// heads: dmu / dlv / Wm / Wl / Z given and dy2 NULL; else dy2 given.  Pieces A, B: dx NULL = no gradient wanted.  Piece widths are
// multiples of 4 up to 64 with 16-byte rows; all [M, 64] matrices have 16-byte rows.
extern "C" hipError_t dn_mlp2_bwd_fused(int64_t M, const float *dy2, int64_t lddy2, const float *dmu, int64_t lddmu, const float *dlv,
                                        int64_t lddlv, int Z, const float *Wm, int64_t ldwm, const float *Wl, int64_t ldwl,
                                        const float *y2, int64_t ldy2, const float *y1, int64_t ldy1, const float *W2, int64_t ldw2,
                                        int act2, float slope2, const float *W1, int64_t ldw1, int act1, float slope1, float *dp2,
                                        int64_t lddp2, float *dp1, int64_t lddp1, float *dxA, int64_t lddxA, int cA, int KA, int accA,
                                        float *dxB, int64_t lddxB, int cB, int KB, int accB, hipStream_t st) {
    if (M < 2048 || !y2 || !y1 || !W2 || !W1 || !dp2 || !dp1) return hipErrorNotSupported;
    const bool heads = dy2 == nullptr;
    if (heads && (!dmu || !dlv || !Wm || !Wl || Z < 1)) return hipErrorNotSupported;
    if (dxB && !dxA) return hipErrorNotSupported;
    auto ok64 = [&](const void *q, int64_t ld) {
        return q && !(ld & 3) && ld >= 64 && !(reinterpret_cast<uintptr_t>(q) & 15) && M * ld * 4 < 0x7fffffffLL;
    };
    if (!ok64(y2, ldy2) || !ok64(y1, ldy1) || !ok64(dp2, lddp2) || !ok64(dp1, lddp1) || (!heads && !ok64(dy2, lddy2))) return hipErrorNotSupported;
    auto okp = [&](const float *q, int64_t ld, int K) {
        return !(K & 3) && K >= 4 && K <= 64 && !(ld & 3) && ld >= K && !(reinterpret_cast<uintptr_t>(q) & 15) && M * ld * 4 < 0x7fffffffLL;
    };
    if (dxA && !okp(dxA, lddxA, KA)) return hipErrorNotSupported;
    if (dxB && !okp(dxB, lddxB, KB)) return hipErrorNotSupported;
    int g3 = 0;
    if (heads) {
        if (M * std::max(lddmu, lddlv) * 4 >= 0x7fffffffLL) return hipErrorNotSupported;
        if (2 * Z <= 4) g3 = 1;
        else if (Z == 16 || Z == 32) {
            if ((lddmu & 3) || (lddlv & 3) || ((reinterpret_cast<uintptr_t>(dmu) | reinterpret_cast<uintptr_t>(dlv)) & 15)) return hipErrorNotSupported;
            g3 = Z / 4;
        } else return hipErrorNotSupported;
    }
    const int nta = dxA ? (KA + 31) / 32 : 0, ntb = dxB ? (KB + 31) / 32 : 0;
    mlpf::BArgs a{dy2, lddy2, dmu, dlv, lddmu, lddlv, Z, Wm, Wl, ldwm, ldwl, y2, y1, ldy2, ldy1, W2, ldw2, W1, ldw1, cA, cB, KA, KB,
                  dp2, dp1, lddp2, lddp1, dxA, dxB, lddxA, lddxB, accA, accB, (int)M, act1, act2, slope1, slope2};
#define MLPB_CASE(A, B, G)                                                         \
    if (nta == A && ntb == B && g3 == G) return mlpf::launch_bwd<A, B, G>(a, st);
    MLPB_CASE(0, 0, 0) MLPB_CASE(1, 0, 0) MLPB_CASE(2, 0, 0) MLPB_CASE(2, 2, 0)
    MLPB_CASE(0, 0, 1) MLPB_CASE(2, 0, 1) MLPB_CASE(2, 2, 1)
    MLPB_CASE(0, 0, 4) MLPB_CASE(2, 0, 4) MLPB_CASE(2, 2, 4)
    MLPB_CASE(0, 0, 8) MLPB_CASE(2, 0, 8) MLPB_CASE(2, 2, 8)
#undef MLPB_CASE
    return hipErrorNotSupported;
}

// Input gradients of ONE linear layer over two input pieces (lg_hiddens / lg: [graph output | h] -> 64, no activation):
// dxA (+)= dy W[:, cA : cA + KA], dxB (+)= dy W[:, cB : cB + KB] from one pass over dy (two products, each reading dy, before).
namespace mlpf {

struct PArgs {
    const float *dy; int64_t lddy;                         // [M, 64]
    const float *W; int64_t ldw; int cA, cB, KA, KB;       // [64, Ktot]
    float *dxA, *dxB; int64_t lddxA, lddxB; int accA, accB;
    int M;
};

template <int NTA, int NTB>
__global__ __launch_bounds__(256, 4) void linear_pieces_bwd_kernel(const PArgs p) {
    extern __shared__ float lds[];           // wAt [NTA][8][64][4] | wBt [NTB][8][64][4]
    float *wAt = lds, *wBt = wAt + NTA * 8 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    stage_quads<NTA * 8 * 64>(wAt, tid, [&](int q) { return quad_t(p.W, p.ldw, p.cA, p.KA, q); });
    if constexpr (NTB != 0) stage_quads<(NTB ? NTB : 1) * 8 * 64>(wBt, tid, [&](int q) { return quad_t(p.W, p.ldw, p.cB, p.KB, q); });
    __syncthreads();
    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * 4;
    const i32x4 rdy = make_rsrc(p.dy, (int64_t)p.M * p.lddy * 4);
    const i32x4 rxa = make_rsrc(p.dxA, (int64_t)p.M * p.lddxA * 4);
    const i32x4 rxb = make_rsrc(NTB ? p.dxB : p.dxA, (int64_t)p.M * (NTB ? p.lddxB : p.lddxA) * 4);
    for (int tile = (int)blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        const int row = tile * 32 + lr;
        f32x4 pq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pq[q] = mf_load4(rdy, row * (int)p.lddy * 4 + 16 * lh + q * 32, 0, 0);
        auto piece = [&](auto nt_tag, const float *wt, const i32x4 rx, const int ldx4, const int K, const int accum) {
            constexpr int NTP = decltype(nt_tag)::value;
            f32x16 ax[NTP ? NTP : 1];
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    if (accum && nt * 32 + 8 * a + 4 * lh < K) c = mf_load4(rx, row * ldx4 + (nt * 32 + 8 * a + 4 * lh) * 4, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ax[nt][4 * a + i] = c[i];
                }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wt + ((nt * 8 + g) * 64 + lane) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ax[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i], pq[g][i], ax[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int f0 = nt * 32 + 8 * a + 4 * lh;
                    if (f0 < K) mf_store4(f32x4{ax[nt][4 * a], ax[nt][4 * a + 1], ax[nt][4 * a + 2], ax[nt][4 * a + 3]}, rx, row * ldx4 + f0 * 4, 0, 0);
                }
        };
        piece(std::integral_constant<int, NTA>{}, wAt, rxa, (int)p.lddxA * 4, p.KA, p.accA);
        if constexpr (NTB > 0) piece(std::integral_constant<int, NTB>{}, wBt, rxb, (int)p.lddxB * 4, p.KB, p.accB);
    }
}

}  // namespace mlpf

// -> hipErrorNotSupported (nothing written) unless: 64 output features, >= 2048 rows, piece widths multiples of 4 up to 64, 16-byte rows.
extern "C" hipError_t dn_linear_pieces_bwd_fused(int64_t M, const float *dy, int64_t lddy, const float *W, int64_t ldw, float *dxA,
                                                 int64_t lddxA, int cA, int KA, int accA, float *dxB, int64_t lddxB, int cB, int KB, int accB,
                                                 hipStream_t st) {
    if (M < 2048 || !dy || !W || !dxA) return hipErrorNotSupported;
    if ((lddy & 3) || lddy < 64 || (reinterpret_cast<uintptr_t>(dy) & 15) || M * lddy * 4 >= 0x7fffffffLL) return hipErrorNotSupported;
    auto okp = [&](const float *q, int64_t ld, int K) {
        return !(K & 3) && K >= 4 && K <= 64 && !(ld & 3) && ld >= K && !(reinterpret_cast<uintptr_t>(q) & 15) && M * ld * 4 < 0x7fffffffLL;
    };
    if (!okp(dxA, lddxA, KA) || (dxB && !okp(dxB, lddxB, KB))) return hipErrorNotSupported;
    const int nta = (KA + 31) / 32, ntb = dxB ? (KB + 31) / 32 : 0;
    mlpf::PArgs a{dy, lddy, W, ldw, cA, cB, KA, KB, dxA, dxB, lddxA, lddxB, accA, accB, (int)M};
    const int64_t tiles = (M + 31) / 32;
    const int64_t blocks = std::min<int64_t>((tiles + 3) / 4, 256 * 4);
    const size_t lds = (size_t)(nta + ntb) * 8 * 256 * sizeof(float);
#define LPB_CASE(A, B)                                                                                                        \
    if (nta == A && ntb == B) {                                                                                               \
        hipLaunchKernelGGL((mlpf::linear_pieces_bwd_kernel<A, B>), dim3((unsigned)blocks), dim3(256), lds, st, a);            \
        return hipGetLastError();                                                                                             \
    }
    LPB_CASE(2, 2) LPB_CASE(2, 0) LPB_CASE(1, 0) LPB_CASE(2, 1) LPB_CASE(1, 2) LPB_CASE(1, 1)
#undef LPB_CASE
    return hipErrorNotSupported;
}
