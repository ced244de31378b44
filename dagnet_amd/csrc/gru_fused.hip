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
        for (int e = tid; e < NQ; e += NTHR)
            *reinterpret_cast<f32x4 *>(lds + e * 4) = *reinterpret_cast<const f32x4 *>(p.pack + e * 4);
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
