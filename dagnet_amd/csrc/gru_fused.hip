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

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

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
};

constexpr int R = 64;                  // rnn_dim = width of every input piece

// NP: pieces of the input (1 or 2)
template <int NP>
__global__ __launch_bounds__(256, 1) void gru_fused_fwd_kernel(const Args p) {
    constexpr int KGX = 8 * NP;              // k-groups of 8 of the input product
    extern __shared__ float lds[];           // wi [6][KGX][64][4] | wh [6][8][64][4] | br, bz, bin, bhn [64] each
    float *wi = lds, *wh = wi + 6 * KGX * 256, *bb = wh + 6 * 8 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // ---- weight images: quad (tile nt, group g, lane l) = W[nt * 32 + (l & 31)][8 g + 4 (l >> 5) .. + 3]
    for (int e = tid; e < 6 * KGX * 64; e += 256) {
        const int l = e & 63, g = (e >> 6) % KGX, nt = (e >> 6) / KGX;
        *reinterpret_cast<f32x4 *>(wi + e * 4) =
            *reinterpret_cast<const f32x4 *>(p.Wih + (int64_t)(nt * 32 + (l & 31)) * p.ldwih + g * 8 + 4 * (l >> 5));
    }
    for (int e = tid; e < 6 * 8 * 64; e += 256) {
        const int l = e & 63, g = (e >> 6) & 7, nt = e >> 9;
        *reinterpret_cast<f32x4 *>(wh + e * 4) =
            *reinterpret_cast<const f32x4 *>(p.Whh + (int64_t)(nt * 32 + (l & 31)) * p.ldwhh + g * 8 + 4 * (l >> 5));
    }
    if (tid < R) {
        bb[tid] = p.bih[tid] + p.bhh[tid];
        bb[R + tid] = p.bih[R + tid] + p.bhh[R + tid];
        bb[2 * R + tid] = p.bih[2 * R + tid];
        bb[3 * R + tid] = p.bhh[2 * R + tid];
    }
    __syncthreads();

    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const i32x4 rx0 = make_rsrc(p.x0, (int64_t)p.M * p.ldx0 * 4), rx1 = make_rsrc(NP == 2 ? p.x1 : p.x0, (int64_t)p.M * (NP == 2 ? p.ldx1 : p.ldx0) * 4);
    const i32x4 rh = make_rsrc(p.h, (int64_t)p.M * p.ldh * 4), rhn = make_rsrc(p.hn, (int64_t)p.M * p.ldhn * 4);
    const i32x4 rg = make_rsrc(p.gates, (int64_t)p.M * p.ldg * 4), rgh = make_rsrc(p.gh, (int64_t)p.M * p.ldgh * 4);
    const int ldx04 = (int)p.ldx0 * 4, ldx14 = (int)p.ldx1 * 4, ldh4 = (int)p.ldh * 4;
    const int ldhn4 = (int)p.ldhn * 4, ldg4 = (int)p.ldg * 4, ldgh4 = (int)p.ldgh * 4;

    f32x4 cx[KGX], ch[8], nx[KGX], nh[8];
    auto load_rows = [&](f32x4 (&dx)[KGX], f32x4 (&dh)[8], int t) {
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
        if (more) load_rows(nx, nh, ntile);
        f32x16 acc[6], ahn[2];               // r0 r1 z0 z1 (input + hidden), n0 n1 (input part) | hidden part of n
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ahn[t][r] = 0.f;
#pragma unroll
        for (int g = 0; g < KGX; ++g)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wi + ((nt * KGX + g) * 64 + lane) * 4);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], cx[g][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], cx[g][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], cx[g][2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], cx[g][3], acc[nt], 0, 0, 0);
                if (nt & 1) __builtin_amdgcn_sched_barrier(0);      // at most two weight quads in flight
            }
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wh + ((nt * 8 + g) * 64 + lane) * 4);
                if (nt < 4) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], ch[g][0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], ch[g][1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], ch[g][2], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], ch[g][3], acc[nt], 0, 0, 0);
                } else {
                    ahn[nt - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], ch[g][0], ahn[nt - 4], 0, 0, 0);
                    ahn[nt - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], ch[g][1], ahn[nt - 4], 0, 0, 0);
                    ahn[nt - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], ch[g][2], ahn[nt - 4], 0, 0, 0);
                    ahn[nt - 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], ch[g][3], ahn[nt - 4], 0, 0, 0);
                }
                if (nt & 1) __builtin_amdgcn_sched_barrier(0);
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
                    const float n = tanhf(acc[4 + t2][4 * a + i] + b_n[i] + r * ghn);
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
#pragma unroll
        for (int g = 0; g < KGX; ++g) cx[g] = nx[g];
#pragma unroll
        for (int g = 0; g < 8; ++g) ch[g] = nh[g];
    }
}

template <int NP>
static hipError_t launch(const Args &a, hipStream_t st) {
    const size_t lds = ((size_t)6 * 8 * NP * 256 + 6 * 8 * 256 + 4 * R) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    const int64_t blocks = std::min<int64_t>((tiles + 3) / 4, cus);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gru_fused_fwd_kernel<NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gru_fused_fwd_kernel<NP>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace gruf

// -> hipErrorNotSupported when the shapes or alignments are not the ones above (the caller then runs the layer as products + cell).
extern "C" hipError_t dn_gru_fused_fwd(int64_t M, int R, int npieces, const float *x0, int64_t ldx0, const float *x1, int64_t ldx1,
                                       const float *h, int64_t ldh, const float *Wih, int64_t ldwih, const float *bih, const float *Whh,
                                       int64_t ldwhh, const float *bhh, float *hn, int64_t ldhn, float *gates, int64_t ldg, float *gh,
                                       int64_t ldgh, hipStream_t st) {
    if (R != gruf::R || npieces < 1 || npieces > 2 || M < 2048 || !bih || !bhh) return hipErrorNotSupported;
    const int64_t lim = 0x7fffffffLL;
    const int64_t lds[] = {ldx0, npieces == 2 ? ldx1 : ldx0, ldh, ldwih, ldwhh, ldhn, ldg, ldgh};
    for (int64_t ld : lds)
        if ((ld & 3) || ld < 64) return hipErrorNotSupported;
    if (ldwih < 64 * npieces || ldg < 192 || ldgh < 192) return hipErrorNotSupported;
    if (M * std::max({ldx0, ldx1, ldh, ldhn, ldg, ldgh}) * 4 >= lim) return hipErrorNotSupported;
    const void *ptrs[] = {x0, npieces == 2 ? x1 : x0, h, Wih, Whh, hn, gates, gh};
    for (const void *q : ptrs)
        if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return hipErrorNotSupported;
    gruf::Args a{x0, ldx0, x1, ldx1, h, ldh, Wih, ldwih, Whh, ldwhh, bih, bhh, hn, ldhn, gates, ldg, gh, ldgh, (int)M};
    return npieces == 2 ? gruf::launch<2>(a, st) : gruf::launch<1>(a, st);
}
