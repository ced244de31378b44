// One GRU layer step of the training pass as ONE kernel (nn.GRU, seq_len = 1, gate order r, z, n; models/dagnet/model.py:109-110,
// :186 and its autograd): h' = (1 - z) n + z h with
//   r = sigm(W_ir x + b_ir + W_hr h + b_hr), z = sigm(W_iz x + b_iz + W_hz h + b_hz), n = tanh(W_in x + b_in + r (W_hn h + b_hn)).
//
// Layer by layer this is two or three 64 -> 192 products, each writing a [M, 192] matrix, and a cell kernel that reads both back:
// ~1 900 bytes per row move through HBM for 147 kFLOP per row (6 GFLOP at 40 960 rows - this is the MFMA-heaviest piece of a
// training step), four launches of ~45 us each.  Here
//   * both weight matrices sit in LDS in MFMA operand order for the life of the workgroup (147 KB: one workgroup per CU);
//   * a wave owns 32-row tiles end to end: x and h rows come straight from HBM as MFMA operands (the skinny-GEMM scheme of
//     gemm_skinny.hip; the h operand registers are, element for element, the h values the blend needs), r / z accumulate both
//     products in one accumulator, the gates are finished in registers and only what the backward pass reads leaves the chip:
//     h', the gates [r | z | n] and W_hn h + b_hn (into columns 128..191 of the layer's gh matrix, where gru_bwd_kernel expects it);
//   * the rows of the wave's next tile are requested before the current tile's 576 (384) MFMAs.
// Rows are 16-byte aligned with strides that are multiples of 4; the input is one or two pieces of 64 columns, rnn_dim = 64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

namespace gruf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ f32x4 gf_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void gf_store4(f32x4 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ i32x4 make_rsrc(const void *base, int64_t bytes) {
    union { const void *p; int w[2]; } u;
    u.p = base;
    return i32x4{u.w[0], u.w[1], (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000};
}

// hardware exp2 / rcp (1 ulp each; the forms the rollout kernels and gemm_f32.hip's tanh epilogue use): the library expf, the IEEE
// division and tanhf cost ~70 VALU instructions per gate element, which a wave cannot hide behind its MFMAs
__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * 2.8853900817779268f)); }

// Copy of a packed weight image into LDS, 12 quads per lane in flight at a time (the plain `for (e) lds[e] = pack[e]` loop compiled
// to one load in flight per iteration: 18 - 36 serial L2 round trips in front of every workgroup's first tile).
template <int NQ, int NTHR>
__device__ __forceinline__ void copy_image(float *lds, const float *pack, int tid) {
    constexpr int IT = (NQ + NTHR - 1) / NTHR, CH = 12;
#pragma unroll
    for (int c0 = 0; c0 < IT; c0 += CH) {
        f32x4 v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (c0 + u < IT) v[u] = *reinterpret_cast<const f32x4 *>(pack + (int64_t)min(tid + NTHR * (c0 + u), NQ - 1) * 4);
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (c0 + u < IT && (NQ % NTHR == 0 || tid + NTHR * (c0 + u) < NQ)) *reinterpret_cast<f32x4 *>(lds + (tid + NTHR * (c0 + u)) * 4) = v[u];
    }
}

struct Args {
    const float *x0; int64_t ldx0;     // [M, 64]
    const float *x1; int64_t ldx1;     // [M, 64] second piece of the input (NP = 2)
    const float *h; int64_t ldh;       // [M, 64]
    const float *Wih; int64_t ldwih;   // [192, 64 NP]
    const float *Whh; int64_t ldwhh;   // [192, 64]
    const float *bih, *bhh;            // [192]
    float *hn; int64_t ldhn;           // [M, 64]
    float *gates; int64_t ldg;         // [M, 192]  r | z | n
    float *gh; int64_t ldgh;           // [M, 192]  only columns 128..191 are written: W_hn h + b_hn
    int M;
    const float *pack;                 // not null: the LDS image (gru_pack_kernel), copied instead of rebuilt by every workgroup
};

constexpr int R = 64;                  // rnn_dim = width of every input piece

// the LDS image of a layer: wi [6][KGX][64][4] | wh [6][8][64][4] | b_r (input + hidden), b_z (both), b_in, b_hn [64] each
template <int KGX>
__device__ __forceinline__ void build_image(float *img, int tid, int nthr, const float *Wih, int64_t ldwih, const float *Whh,
                                            int64_t ldwhh, const float *bih, const float *bhh) {
    float *wi = img, *wh = wi + 6 * KGX * 256, *bb = wh + 6 * 8 * 256;
    for (int e = tid; e < 6 * KGX * 64; e += nthr) {
        const int l = e & 63, g = (e >> 6) % KGX, nt = (e >> 6) / KGX;
        *reinterpret_cast<f32x4 *>(wi + e * 4) =
            *reinterpret_cast<const f32x4 *>(Wih + (int64_t)(nt * 32 + (l & 31)) * ldwih + g * 8 + 4 * (l >> 5));
    }
    for (int e = tid; e < 6 * 8 * 64; e += nthr) {
        const int l = e & 63, g = (e >> 6) & 7, nt = e >> 9;
        *reinterpret_cast<f32x4 *>(wh + e * 4) =
            *reinterpret_cast<const f32x4 *>(Whh + (int64_t)(nt * 32 + (l & 31)) * ldwhh + g * 8 + 4 * (l >> 5));
    }
    for (int f = tid; f < R; f += nthr) {
        bb[f] = bih[f] + bhh[f];
        bb[R + f] = bih[R + f] + bhh[R + f];
        bb[2 * R + f] = bih[2 * R + f];
        bb[3 * R + f] = bhh[2 * R + f];
    }
}

template <int KGX>
__global__ __launch_bounds__(256) void gru_pack_kernel(float *out, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh,
                                                       const float *bih, const float *bhh) {
    build_image<KGX>(out, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x), Wih, ldwih, Whh, ldwhh, bih, bhh);
}

// NP: pieces of the input (1 or 2).  WAVES: waves per workgroup (one workgroup per CU): 4 = one wave per SIMD with the next tile's
// rows prefetched into registers of their own, 8 = two waves per SIMD (256 registers each, no prefetch: the other wave covers).
template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void gru_fused_fwd_kernel(const Args p) {
    constexpr int NTHR = 64 * WAVES;
    constexpr bool PF = WAVES == 4;
    constexpr int KGX = 8 * NP;              // k-groups of 8 of the input product
    extern __shared__ float lds[];           // wi [6][KGX][64][4] | wh [6][8][64][4] | br, bz, bin, bhn [64] each
    float *wi = lds, *wh = wi + 6 * KGX * 256, *bb = wh + 6 * 8 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // ---- weight images: quad (tile nt, group g, lane l) = W[nt * 32 + (l & 31)][8 g + 4 (l >> 5) .. + 3]
    if (p.pack) {               // packed once per training step: a straight, coalesced copy
        constexpr int NQ = (6 * KGX * 256 + 6 * 8 * 256 + 4 * R) / 4;
        copy_image<NQ, NTHR>(lds, p.pack, tid);
    } else {
        build_image<KGX>(lds, tid, NTHR, p.Wih, p.ldwih, p.Whh, p.ldwhh, p.bih, p.bhh);
    }
    __syncthreads();

    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * WAVES;
    int tile = (int)blockIdx.x * WAVES + wave;
    if (tile >= n_tiles) return;
    const i32x4 rx0 = make_rsrc(p.x0, (int64_t)p.M * p.ldx0 * 4), rx1 = make_rsrc(NP == 2 ? p.x1 : p.x0, (int64_t)p.M * (NP == 2 ? p.ldx1 : p.ldx0) * 4);
    const i32x4 rh = make_rsrc(p.h, (int64_t)p.M * p.ldh * 4), rhn = make_rsrc(p.hn, (int64_t)p.M * p.ldhn * 4);
    const i32x4 rg = make_rsrc(p.gates, (int64_t)p.M * p.ldg * 4), rgh = make_rsrc(p.gh, (int64_t)p.M * p.ldgh * 4);
    const int ldx04 = (int)p.ldx0 * 4, ldx14 = (int)p.ldx1 * 4, ldh4 = (int)p.ldh * 4;
    const int ldhn4 = (int)p.ldhn * 4, ldg4 = (int)p.ldg * 4, ldgh4 = (int)p.ldgh * 4;

    f32x4 cx[KGX], ch[8], nx[PF ? KGX : 1], nh[PF ? 8 : 1];
    auto load_rows = [&](auto &dx, auto &dh, int t) {
        const int row = t * 32 + lr;             // rows past M: beyond the buffers, the range check returns 0
#pragma unroll
        for (int g = 0; g < 8; ++g) dx[g] = gf_load4(rx0, row * ldx04 + 16 * lh + g * 32, 0, 0);
        if constexpr (NP == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) dx[8 + g] = gf_load4(rx1, row * ldx14 + 16 * lh + g * 32, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) dh[g] = gf_load4(rh, row * ldh4 + 16 * lh + g * 32, 0, 0);
    };
    load_rows(cx, ch, tile);
    while (true) {
        const int ntile = tile + stride;
        const bool more = ntile < n_tiles;
        if constexpr (PF) {
            if (more) load_rows(nx, nh, ntile);
        }
        f32x16 acc[6], ahn[2];               // r0 r1 z0 z1 (input + hidden), n0 n1 (input part) | hidden part of n
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ahn[t][r] = 0.f;
        // One step = one k-group of 8 against all six gate tiles: the six weight quads of step s + 1 are read from LDS while step s
        // multiplies, and consecutive MFMAs go to different accumulators (four in a row on one accumulator wait for each other:
        // 108 instead of 64 clocks per MFMA with a single wave per SIMD).
        auto wquads = [&](f32x4 (&w)[6], int sidx) {
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                w[nt] = sidx < KGX ? *reinterpret_cast<const f32x4 *>(wi + ((nt * KGX + sidx) * 64 + lane) * 4)
                                   : *reinterpret_cast<const f32x4 *>(wh + ((nt * 8 + (sidx - KGX)) * 64 + lane) * 4);
        };
        f32x4 w0[6], w1[6];
        wquads(w0, 0);
#pragma unroll
        for (int sidx = 0; sidx < KGX + 8; ++sidx) {
            f32x4 (&wc)[6] = (sidx & 1) ? w1 : w0;
            f32x4 (&wn)[6] = (sidx & 1) ? w0 : w1;
            if (sidx + 1 < KGX + 8) wquads(wn, sidx + 1);
            const f32x4 b = sidx < KGX ? cx[sidx < KGX ? sidx : 0] : ch[sidx < KGX ? 0 : sidx - KGX];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    if (sidx >= KGX && nt >= 4) ahn[nt - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nt][i], b[i], ahn[nt - 4], 0, 0, 0);
                    else acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nt][i], b[i], acc[nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- gates: lane = row, register 4 a + i of tile t2 <-> feature f = 32 t2 + 8 a + 4 lh + i, which is element i of the
        // h operand quad g = 4 t2 + a
        const int row = tile * 32 + lr;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int f0 = 32 * t2 + 8 * a + 4 * lh;
                const f32x4 b_r = *reinterpret_cast<const f32x4 *>(bb + f0), b_z = *reinterpret_cast<const f32x4 *>(bb + R + f0);
                const f32x4 b_n = *reinterpret_cast<const f32x4 *>(bb + 2 * R + f0), b_h = *reinterpret_cast<const f32x4 *>(bb + 3 * R + f0);
                const f32x4 hq = ch[4 * t2 + a];
                f32x4 vr, vz, vn, vg, vh;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = sigm(acc[t2][4 * a + i] + b_r[i]);
                    const float z = sigm(acc[2 + t2][4 * a + i] + b_z[i]);
                    const float ghn = ahn[t2][4 * a + i] + b_h[i];
                    const float n = tanh_fast(acc[4 + t2][4 * a + i] + b_n[i] + r * ghn);
                    vr[i] = r; vz[i] = z; vn[i] = n; vg[i] = ghn;
                    vh[i] = (1.0f - z) * n + z * hq[i];
                }
                gf_store4(vh, rhn, row * ldhn4 + f0 * 4, 0, 0);
                gf_store4(vr, rg, row * ldg4 + f0 * 4, 0, 0);
                gf_store4(vz, rg, row * ldg4 + (R + f0) * 4, 0, 0);
                gf_store4(vn, rg, row * ldg4 + (2 * R + f0) * 4, 0, 0);
                gf_store4(vg, rgh, row * ldgh4 + (2 * R + f0) * 4, 0, 0);
            }
        if (!more) break;
        tile = ntile;
        if constexpr (PF) {
#pragma unroll
            for (int g = 0; g < KGX; ++g) cx[g] = nx[g];
#pragma unroll
            for (int g = 0; g < 8; ++g) ch[g] = nh[g];
        } else {
            load_rows(cx, ch, tile);
        }
    }
}

template <int NP, int WAVES>
static hipError_t launch(const Args &a, hipStream_t st) {
    const size_t lds = ((size_t)6 * 8 * NP * 256 + 6 * 8 * 256 + 4 * R) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    const int64_t blocks = std::min<int64_t>((tiles + WAVES - 1) / WAVES, cus);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_fused_fwd_kernel<NP, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gru_fused_fwd_kernel<NP, WAVES>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the same layer step as ONE kernel.  Layer by layer it is the gate-gradient kernel (train_ops.hip gru_bwd_kernel:
// reads dh', the gates, W_hn h + b_hn and h, writes the [M,192] gate gradients of both products) followed by two or three
// K = 192 input-gradient products that read those matrices back (gemm_f32.hip, ~55 us each at 81 920 rows): ~590 MB and four
// launches per layer step.  Here a wave takes a 32-row tile from dh' to the input gradients: the gate gradients are computed in
// the registers that are the MFMA operands of the two products (lane = row, k = 8 g + 4 lh + i as in the forward kernel), written
// out once for the deferred weight-gradient products, and multiplied with W_ih / W_hh images that stay in LDS:
//     dn = dh' (1 - z), dz = dh' (h - n);  dpn = dn (1 - n^2), dpr = dpn ghn r (1 - r), dpz = dz z (1 - z)
//     dgi = [dpr | dpz | dpn],  dgh = [dpr | dpz | dpn r];  dx_p (+)= dgi W_ih[:, 64 p : 64 p + 64],  dh (+)= dh' z + dgh W_hh
// ---------------------------------------------------------------------------------------------------------------------------
struct BArgs {
    const float *dhn; int64_t lddhn;   // [M, 64]  gradient of h'
    const float *gates; int64_t ldg;   // [M, 192] r | z | n
    const float *gh; int64_t ldgh;     // [M, 192] columns 128..191 = W_hn h + b_hn
    const float *h; int64_t ldh;       // [M, 64]
    const float *pack;                 // dn_gru_pack_bwd image
    float *dgi; int64_t lddgi;         // [M, 192]
    float *dgh; int64_t lddgh;         // [M, 192]
    float *dx0; int64_t lddx0; int acc0;      // [M, 64] (null: not wanted)
    float *dx1; int64_t lddx1; int acc1;
    float *dh; int64_t lddh; int acch;
    int M;
};

constexpr int KGB = 24;                // k-groups of 8 over the 192 gate columns

// image: wi [2 NP][24][64][4] | wh [2][24][64][4]; quad (nt, g, l) = W[8 g + 4 (l >> 5) + 0..3][32 nt + (l & 31)]
template <int NP>
__global__ __launch_bounds__(256) void gru_pack_bwd_kernel(float *out, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh) {
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x), nthr = (int)(gridDim.x * blockDim.x);
    constexpr int NI = 2 * NP * KGB * 64, NH = 2 * KGB * 64;
    for (int e = tid; e < NI + NH; e += nthr) {
        const bool hid = e >= NI;
        const int q = hid ? e - NI : e;
        const int l = q & 63, g = (q >> 6) % KGB, nt = (q >> 6) / KGB;
        const float *W = hid ? Whh : Wih;
        const int64_t ld = hid ? ldwhh : ldwih;
        const int k = 8 * g + 4 * (l >> 5), n = 32 * nt + (l & 31);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = W[(int64_t)(k + i) * ld + n];
        *reinterpret_cast<f32x4 *>(out + (int64_t)e * 4) = v;
    }
}

template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void gru_fused_bwd_kernel(const BArgs p) {
    constexpr int NTHR = 64 * WAVES;
    constexpr int NTX = 2 * NP;
    extern __shared__ float lds[];
    float *wi = lds, *wh = lds + NTX * KGB * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    {
        constexpr int NQ = (NTX * KGB * 256 + 2 * KGB * 256) / 4;
        copy_image<NQ, NTHR>(lds, p.pack, tid);
    }
    __syncthreads();
    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * WAVES;
    // tiles are dealt round-robin over the workgroups first, so that a last partial round spreads over all CUs
    int tile = (int)blockIdx.x + (int)gridDim.x * wave;
    if (tile >= n_tiles) return;
    const i32x4 rd = make_rsrc(p.dhn, (int64_t)p.M * p.lddhn * 4), rg = make_rsrc(p.gates, (int64_t)p.M * p.ldg * 4);
    const i32x4 rgh = make_rsrc(p.gh, (int64_t)p.M * p.ldgh * 4), rh = make_rsrc(p.h, (int64_t)p.M * p.ldh * 4);
    const i32x4 rdgi = make_rsrc(p.dgi, (int64_t)p.M * p.lddgi * 4), rdgh = make_rsrc(p.dgh, (int64_t)p.M * p.lddgh * 4);
    const i32x4 rx0 = make_rsrc(p.dx0 ? p.dx0 : p.dgi, p.dx0 ? (int64_t)p.M * p.lddx0 * 4 : 0);
    const i32x4 rx1 = make_rsrc(p.dx1 ? p.dx1 : p.dgi, p.dx1 ? (int64_t)p.M * p.lddx1 * 4 : 0);
    const i32x4 rdh = make_rsrc(p.dh ? p.dh : p.dgi, p.dh ? (int64_t)p.M * p.lddh * 4 : 0);
    const int ldd4 = (int)p.lddhn * 4, ldg4 = (int)p.ldg * 4, ldgh4 = (int)p.ldgh * 4, ldh4 = (int)p.ldh * 4;
    const int lddgi4 = (int)p.lddgi * 4, lddgh4 = (int)p.lddgh * 4;
    const int lddx04 = (int)p.lddx0 * 4, lddx14 = (int)p.lddx1 * 4, lddh4 = (int)p.lddh * 4;

    for (; tile < n_tiles; tile += stride) {
        const int row = tile * 32 + lr;          // rows past M: the range check returns 0 / drops the store
        // ---- gate gradients, two k-groups at a time (12 quads in flight; the other wave of the SIMD covers their latency)
        f32x4 gr[8], gz[8], gn[8], gnr[8], dhz[8];
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) {
            f32x4 d[2], hq[2], r[2], z[2], n[2], gq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int off = 16 * lh + (2 * gp + u) * 32;
                d[u] = gf_load4(rd, row * ldd4 + off, 0, 0);
                hq[u] = gf_load4(rh, row * ldh4 + off, 0, 0);
                r[u] = gf_load4(rg, row * ldg4 + off, 0, 0);
                z[u] = gf_load4(rg, row * ldg4 + 256 + off, 0, 0);
                n[u] = gf_load4(rg, row * ldg4 + 512 + off, 0, 0);
                gq[u] = gf_load4(rgh, row * ldgh4 + 512 + off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gj = 2 * gp + u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dd = d[u][i], rr = r[u][i], zz = z[u][i], nn = n[u][i];
                    const float dn = dd * (1.0f - zz), dz = dd * (hq[u][i] - nn);
                    const float dpn = dn * (1.0f - nn * nn);
                    gr[gj][i] = dpn * gq[u][i] * rr * (1.0f - rr);
                    gz[gj][i] = dz * zz * (1.0f - zz);
                    gn[gj][i] = dpn;
                    gnr[gj][i] = dpn * rr;
                    dhz[gj][i] = dd * zz;
                }
                const int off = 16 * lh + gj * 32;
                gf_store4(gr[gj], rdgi, row * lddgi4 + off, 0, 0);
                gf_store4(gz[gj], rdgi, row * lddgi4 + 256 + off, 0, 0);
                gf_store4(gn[gj], rdgi, row * lddgi4 + 512 + off, 0, 0);
                gf_store4(gr[gj], rdgh, row * lddgh4 + off, 0, 0);
                gf_store4(gz[gj], rdgh, row * lddgh4 + 256 + off, 0, 0);
                gf_store4(gnr[gj], rdgh, row * lddgh4 + 512 + off, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- one pair of 32-column output tiles of a product: acc[t] = W image tiles (nt0, nt0 + 1) x gate gradients.  The two
        // accumulators alternate (four MFMAs in a row on one accumulator wait for each other); the weight quads of group g + 1 are
        // read from LDS while group g multiplies.
        auto product = [&](f32x16 (&acc)[2], const float *img, int nt0, bool hidden) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
            f32x4 w0[2], w1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) w0[t] = *reinterpret_cast<const f32x4 *>(img + (((nt0 + t) * KGB + 0) * 64 + lane) * 4);
#pragma unroll
            for (int g = 0; g < KGB; ++g) {
                f32x4 (&wc)[2] = (g & 1) ? w1 : w0;
                f32x4 (&wn)[2] = (g & 1) ? w0 : w1;
                if (g + 1 < KGB) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) wn[t] = *reinterpret_cast<const f32x4 *>(img + (((nt0 + t) * KGB + g + 1) * 64 + lane) * 4);
                }
                const f32x4 b = g < 8 ? gr[g & 7] : g < 16 ? gz[g & 7] : (hidden ? gnr[g & 7] : gn[g & 7]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[0][i], b[i], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[1][i], b[i], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // lane = row, register 4 a + i of tile t <-> column 32 t + 8 a + 4 lh + i of the 64-wide destination
        auto put = [&](const f32x16 (&acc)[2], i32x4 rsrc, int ld4, int accumulate, const f32x4 *extra) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int voff = row * ld4 + (32 * t + 8 * a + 4 * lh) * 4;
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[t][4 * a + i];
                    if (extra) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += extra[4 * t + a][i];
                    }
                    if (accumulate) {
                        const f32x4 c = gf_load4(rsrc, voff, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += c[i];
                    }
                    gf_store4(v, rsrc, voff, 0, 0);
                }
        };
        f32x16 acc[2];
        if (p.dh) {              // dh = dh' z + dgh W_hh: column 32 t + 8 a + 4 lh + i is element i of this lane's quad gj = 4 t + a
            product(acc, wh, 0, true);
            put(acc, rdh, lddh4, p.acch, dhz);
        }
        if (p.dx0) {
            product(acc, wi, 0, false);
            put(acc, rx0, lddx04, p.acc0, nullptr);
        }
        if constexpr (NP == 2) {
            if (p.dx1) {
                product(acc, wi, 2, false);
                put(acc, rx1, lddx14, p.acc1, nullptr);
            }
        }
    }
}

template <int NP, int WAVES>
static hipError_t launch_bwd(const BArgs &a, hipStream_t st) {
    const size_t lds = ((size_t)2 * NP * KGB * 256 + 2 * KGB * 256) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    const int64_t blocks = std::min<int64_t>(tiles, cus);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_fused_bwd_kernel<NP, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gru_fused_bwd_kernel<NP, WAVES>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, a);
    return hipGetLastError();
}

}  // namespace gruf

// -> hipErrorNotSupported when the shapes or alignments are not the ones above (the caller then runs the layer as products + cell).
extern "C" hipError_t dn_gru_fused_fwd(int64_t M, int R, int npieces, const float *x0, int64_t ldx0, const float *x1, int64_t ldx1,
                                       const float *h, int64_t ldh, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                                       int64_t ldwhh, const float *bhh, float *hn, int64_t ldhn, float *gates, int64_t ldg, float *gh,
                                       int64_t ldgh, const float *pack, hipStream_t st) {
    if (R != gruf::R || npieces < 1 || npieces > 2 || M < 2048 || !bih || !bhh) return hipErrorNotSupported;
    const int64_t lim = 0x7fffffffLL;
    const int64_t lds[] = {ldx0, npieces == 2 ? ldx1 : ldx0, ldh, ldwih, ldwhh, ldhn, ldg, ldgh};
    for (int64_t ld : lds)
        if ((ld & 3) || ld < 64) return hipErrorNotSupported;
    if (ldwih < 64 * npieces || ldg < 192 || ldgh < 192) return hipErrorNotSupported;
    if (M * std::max({ldx0, ldx1, ldh, ldhn, ldg, ldgh}) * 4 >= lim) return hipErrorNotSupported;
    const void *ptrs[] = {x0, npieces == 2 ? x1 : x0, h, Wih, Whh, hn, gates, gh, pack ? pack : h};
    for (const void *q : ptrs)
        if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return hipErrorNotSupported;
    gruf::Args a{x0, ldx0, x1, ldx1, h, ldh, Wih, ldwih, Whh, ldwhh, bih, bhh, hn, ldhn, gates, ldg, gh, ldgh, (int)M, pack};
    // one wave per SIMD with the next tile prefetched; two waves per SIMD without prefetch (WAVES = 8) measured 134 against 112 us
    return npieces == 2 ? gruf::launch<2, 4>(a, st) : gruf::launch<1, 4>(a, st);
}

// floats of a layer's packed image for `npieces` 64-wide input pieces
extern "C" int64_t dn_gru_pack_floats(int npieces) { return (int64_t)6 * 8 * npieces * 256 + 6 * 8 * 256 + 4 * gruf::R; }
// out = the LDS image of the layer (valid until its weights change): built once per training step instead of by every workgroup
// of every time step (147 KB gathered 16 bytes at a time from rows 256 / 512 bytes apart)
extern "C" hipError_t dn_gru_pack(int R, int npieces, const float *Wih, int64_t ldwih, const float *bih, const float *Whh, int64_t ldwhh,
                                  const float *bhh, float *out, hipStream_t st) {
    if (R != gruf::R || npieces < 1 || npieces > 2 || !bih || !bhh || (ldwih & 3) || (ldwhh & 3) || ldwih < 64 * npieces || ldwhh < 64)
        return hipErrorNotSupported;
    const void *ptrs[] = {Wih, Whh, out};
    for (const void *q : ptrs)
        if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return hipErrorNotSupported;
    if (npieces == 2) hipLaunchKernelGGL(gruf::gru_pack_kernel<16>, dim3(36), dim3(256), 0, st, out, Wih, ldwih, Whh, ldwhh, bih, bhh);
    else hipLaunchKernelGGL(gruf::gru_pack_kernel<8>, dim3(24), dim3(256), 0, st, out, Wih, ldwih, Whh, ldwhh, bih, bhh);
    return hipGetLastError();
}

// ---- backward of the layer step (gru_fused_bwd_kernel).  -> hipErrorNotSupported when the shape is not the fused one.
extern "C" int64_t dn_gru_pack_bwd_floats(int npieces) { return (int64_t)2 * npieces * gruf::KGB * 256 + 2 * gruf::KGB * 256; }
extern "C" hipError_t dn_gru_pack_bwd(int R, int npieces, const float *Wih, int64_t ldwih, const float *Whh, int64_t ldwhh, float *out,
                                      hipStream_t st) {
    if (R != gruf::R || npieces < 1 || npieces > 2 || !Wih || !Whh || !out || ldwih < 64 * npieces || ldwhh < 64 ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return hipErrorNotSupported;
    if (npieces == 2) hipLaunchKernelGGL(gruf::gru_pack_bwd_kernel<2>, dim3(36), dim3(256), 0, st, out, Wih, ldwih, Whh, ldwhh);
    else hipLaunchKernelGGL(gruf::gru_pack_bwd_kernel<1>, dim3(24), dim3(256), 0, st, out, Wih, ldwih, Whh, ldwhh);
    return hipGetLastError();
}
extern "C" hipError_t dn_gru_fused_bwd(int64_t M, int R, int npieces, const float *dhn, int64_t lddhn, const float *gates, int64_t ldg,
                                       const float *gh, int64_t ldgh, const float *h, int64_t ldh, const float *pack, float *dgi,
                                       int64_t lddgi, float *dgh, int64_t lddgh, float *dx0, int64_t lddx0, int acc0, float *dx1,
                                       int64_t lddx1, int acc1, float *dh, int64_t lddh, int acch, hipStream_t st) {
    if (R != gruf::R || npieces < 1 || npieces > 2 || M < 2048 || !pack) return hipErrorNotSupported;
    if (npieces == 1 && dx1) return hipErrorNotSupported;
    const int64_t lim = 0x7fffffffLL;
    const int64_t lds[] = {lddhn, ldg, ldgh, ldh, lddgi, lddgh, dx0 ? lddx0 : 64, dx1 ? lddx1 : 64, dh ? lddh : 64};
    int64_t ldmax = 0;
    for (int64_t ld : lds) {
        if ((ld & 3) || ld < 64) return hipErrorNotSupported;
        ldmax = std::max(ldmax, ld);
    }
    if (ldg < 192 || ldgh < 192 || lddgi < 192 || lddgh < 192 || M * ldmax * 4 >= lim) return hipErrorNotSupported;
    const void *ptrs[] = {dhn, gates, gh, h, pack, dgi, dgh, dx0 ? dx0 : dgi, dx1 ? dx1 : dgi, dh ? dh : dgi};
    for (const void *q : ptrs)
        if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return hipErrorNotSupported;
    gruf::BArgs a{dhn, lddhn, gates, ldg, gh, ldgh, h, ldh, pack, dgi, lddgi, dgh, lddgh, dx0, lddx0, acc0, dx1, lddx1, acc1, dh, lddh, acch, (int)M};
    static const int waves = getenv("DAGNET_GRU_BWD_WAVES") ? atoi(getenv("DAGNET_GRU_BWD_WAVES")) : 8;
    if (npieces == 2) return waves == 4 ? gruf::launch_bwd<2, 4>(a, st) : gruf::launch_bwd<2, 8>(a, st);
    return waves == 4 ? gruf::launch_bwd<1, 4>(a, st) : gruf::launch_bwd<1, 8>(a, st);
}
