// Tall-and-skinny fp32 MFMA product for the layer-by-layer training step: C[M,N] = act([C +] A[M,K] op(B) + bias + gather)
// with M in the tens of thousands and N, K <= 192 / 128 (nn.Linear layers of width 64 / 90 / 192 and their input gradients;
// models/dagnet/model.py:60-116 and its autograd).
//
// Such a product moves 4 (M K + M N) bytes for 2 M N K flops - about 16 flop/byte at N = K = 64: it is bound by streaming A
// and C through HBM, and the LDS-tiled gemm_nt_kernel (gemm_f32.hip) reaches only ~2 TB/s on it because every 64-row tile
// re-stages the (tiny) weight matrix and its A rows through LDS behind two barriers.  Here
//   * the weight matrix is converted ONCE per workgroup into MFMA operand order in LDS ([n-tile][k-group][lane][4]:
//     conflict-free ds_read_b128, one per 4 MFMAs) and stays there;
//   * every WAVE owns 32-row tiles on its own (no barrier after the prologue): the lane of row r loads its row of A straight
//     from global memory as float4s that ARE the second MFMA operand (k = 8 g + 4 h + i for MFMA i of group g in lane half
//     h - the same permutation the weights are stored with), and the accumulator layout (lane = row, 4 consecutive features
//     per register quad) stores to C as float4s;
//   * the loads of a wave's next tile are issued before the MFMAs of the current one; with 4 waves per SIMD (N, K <= 64) a
//     wave usually has one tile and the chip simply has every row in flight at once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

namespace skinny {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_RELU = 3 };

struct Args {
    const float *A; int64_t lda;      // [M, K]
    const float *W; int64_t ldw;      // [N, K], or [K, N] when transb
    const float *bias;                // [N] or null
    float *C; int64_t ldc;            // [M, N]
    int M, N, K;
    float slope;
    const float *gather; const int32_t *gidx; int64_t gs_g, gs_n;      // C[m][n] += gather[gidx[m] * gs_g + n * gs_n]
    int act, accumulate, transb;
    // DACT form (the input-gradient product of a layer with an activation): the first operand is dY * act'(Y), computed on
    // load from the layer's saved output Y [M, K]; the product dP = dY * act'(Y) is also written out (the weight-gradient
    // product reads it later).  dact = ACT_* of the layer.
    const float *Y; int64_t ldy;
    float *P; int64_t ldp;
    int dact; float dslope;
};

__device__ __forceinline__ float act_apply(int act, float v, float slope) {
    if (act == ACT_LRELU) return fmaxf(v, v * slope);
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer accesses declared on the LLVM intrinsics (as in v2_common.h): one 4-SGPR descriptor per matrix, a 32-bit per-lane
// byte offset and an immediate - no 64-bit address pair per access (those made the first version of this kernel spill), and
// accesses past the end of the buffer return 0 / are dropped by the hardware range check.
__device__ f32x4 sk_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float sk_load1(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void sk_store4(f32x4 data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void sk_store1(float data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
// a quad of columns c0 .. c0+3 of a row whose matrix has `cols` columns: whole quads as one 16-byte store, the one quad that
// straddles the last column element by element - columns >= cols belong to the caller (a column slice of a wider buffer)
__device__ __forceinline__ void sk_store_quad(f32x4 v, i32x4 rsrc, int voff, int c0, int cols) {
    if (c0 + 4 <= cols) {
        sk_store4(v, rsrc, voff, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (c0 + i < cols) sk_store1(v[i], rsrc, voff + 4 * i, 0, 0);
    }
}

__device__ __forceinline__ i32x4 make_rsrc(const void *base, int64_t bytes) {
    union { const void *p; int w[2]; } u;
    u.p = base;
    return i32x4{u.w[0], u.w[1], (int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes), 0x00020000};
}

// NT: 32-wide output tiles per wave (N <= 32 NT).  KG: k-groups of 8 (K <= 8 KG).
// Rows of A and C are 16-byte aligned with padded strides (lda >= 4 ceil(K / 4), ldc >= 4 ceil(N / 4)): float4 accesses.
// WPS: workgroups per CU the register budget is cut for (512 / WPS registers per lane).
// GATHER: the one-hot table term is present (its loads need registers of their own: they are issued before the MFMAs).
template <int NT, int KG, int WPS, bool GATHER, bool DACT = false>
__global__ __launch_bounds__(256, WPS) void skinny_kernel(const Args p) {
    extern __shared__ float wimg[];          // [NT][KG][64 lanes][4] weights, then [NT*32] bias (zero padded)
    float *bimg = wimg + NT * KG * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int n_tiles = (p.M + 31) >> 5;
    const int stride = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;

    const i32x4 ra = make_rsrc(p.A, (int64_t)p.M * p.lda * 4), rc = make_rsrc(p.C, (int64_t)p.M * p.ldc * 4);
    const i32x4 rg = make_rsrc(p.gather, 0x7fffffffLL);
    const int lda4 = (int)p.lda * 4, ldc4 = (int)p.ldc * 4;
    // element i of group g (k = 8 g + 4 h + i) is valid iff 8 g + i < kq; anything else is zeroed by a select: the padding
    // of a row is uninitialised memory, and a NaN there must not meet the zero weight of the padded k.  Rows past M read
    // past the end of the buffer (0) and their stores are dropped by the same range check.
    const int kq = p.K - 4 * lh;
    constexpr bool PF = KG <= 12;            // wider A tiles (K = 192: the GRU gates' input gradient) do not fit twice
    f32x4 cur[KG], nxt[PF ? KG : 1];
    f32x4 ynx[DACT ? KG : 1];                // DACT: the saved outputs of the tile in flight (same lanes, same k as its dY)
    const i32x4 ry = make_rsrc(DACT ? p.Y : p.A, (int64_t)p.M * (DACT ? p.ldy : p.lda) * 4);
    const i32x4 rp = make_rsrc(DACT ? (const float *)p.P : p.A, (int64_t)p.M * (DACT ? p.ldp : p.lda) * 4);
    const int ldy4 = (int)p.ldy * 4, ldp4 = (int)p.ldp * 4;
    auto load_rows = [&](f32x4 (&dst)[KG], int t) {
        const int voff = (t * 32 + lr) * lda4 + 16 * lh;
#pragma unroll
        for (int g = 0; g < KG; ++g) dst[g] = sk_load4(ra, voff + g * 32, 0, 0);
        if constexpr (DACT) {
            const int yoff = (t * 32 + lr) * ldy4 + 16 * lh;
#pragma unroll
            for (int g = 0; g < KG; ++g) ynx[g] = sk_load4(ry, yoff + g * 32, 0, 0);
        }
    };
    // dY -> dY * act'(Y) in place, and out to P (columns < K only)
    auto dact = [&](f32x4 (&dst)[KG], int t) {
        if constexpr (DACT) {
            const int poff = (t * 32 + lr) * ldp4 + 16 * lh;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = ynx[g][i];
                    const float d = p.dact == ACT_LRELU ? (y > 0.f ? 1.f : p.dslope) : p.dact == ACT_RELU ? (y > 0.f ? 1.f : 0.f)
                                    : p.dact == ACT_TANH ? 1.f - y * y : 1.f;
                    dst[g][i] *= d;
                }
                sk_store_quad(dst[g], rp, poff + g * 32, 8 * g + 4 * lh, p.K);
            }
        }
    };
    auto mask = [&](f32x4 (&dst)[KG]) {
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[g][i] = (8 * g + i < kq) ? dst[g][i] : 0.f;
    };
    // ---- weight image (once per workgroup): quad e = (nt, g, lane l) holds W[n = 32 nt + (l & 31)][k = 8 g + 4 (l >> 5) + 0..3];
    // a thread's quads advance by four k-groups per iteration (no division by KG in the loop).  Order of the requests: the
    // image's quads (L2 hits), then the first tile's rows (HBM), then the image is written to LDS - loads return in order, so
    // the image is complete while the rows are still on their way, and the rows' latency overlaps the LDS writes and the barrier
    // instead of starting behind them (a tile past the end reads zeros through the range check).
    constexpr int QI = (NT * KG * 64 + 255) / 256;
    f32x4 wv[QI];
    {
        const bool quads = !p.transb && !(p.ldw & 3) && !(reinterpret_cast<uintptr_t>(p.W) & 15);
        int g = wave, nt = 0;
        while (g >= KG) { g -= KG; ++nt; }
        auto image = [&](auto load) {
#pragma unroll
            for (int q = 0; q < QI; ++q) {
                const int n = nt * 32 + lr, k = g * 8 + 4 * lh;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (nt < NT && n < p.N && k < p.K) {
                    v = load(n, k);
#pragma unroll
                    for (int i = 1; i < 4; ++i)
                        if (k + i >= p.K) v[i] = 0.f;
                }
                wv[q] = v;
                g += 4;
                while (g >= KG) { g -= KG; ++nt; }
            }
        };
        if (quads) {
            image([&](int n, int k) { return *reinterpret_cast<const f32x4 *>(p.W + (int64_t)n * p.ldw + k); });      // k + 3 < ldw
        } else if (p.transb) {
            image([&](int n, int k) {                                   // lanes = consecutive n: coalesced rows of W [K, N]
                const float *w = p.W + (int64_t)k * p.ldw + n;
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = k + i < p.K ? w[(int64_t)i * p.ldw] : 0.f;
                return v;
            });
        } else {
            image([&](int n, int k) {
                const float *w = p.W + (int64_t)n * p.ldw + k;
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = k + i < p.K ? w[i] : 0.f;
                return v;
            });
        }
    }
    load_rows(cur, tile);
#pragma unroll
    for (int q = 0; q < QI; ++q)
        if (tid + 256 * q < NT * KG * 64) *reinterpret_cast<f32x4 *>(wimg + (tid + 256 * q) * 4) = wv[q];
    if (tid < NT * 32) bimg[tid] = (p.bias && tid < p.N) ? p.bias[tid] : 0.f;
    __syncthreads();
    if (tile >= n_tiles) return;
    while (true) {
        const int ntile = tile + stride;
        const bool more = ntile < n_tiles;
        dact(cur, tile);                     // consumes ynx (the current tile's Y) before the next tile's loads overwrite it
        if constexpr (PF) {
            if (more) load_rows(nxt, ntile);
        }
        mask(cur);
        const int row = tile * 32 + lr;
        const int coff = row * ldc4 + 16 * lh;
        f32x16 acc[NT];
        if (p.accumulate) {                  // the accumulating form reads C before the MFMAs
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 c = sk_load4(rc, coff + nt * 128 + a * 32, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[nt][4 * a + i] = c[i];
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        }
        // one-hot table term: element loads issued here, consumed after the MFMAs (their latency hides behind them)
        float gpre[GATHER ? NT : 1][16];
        if (GATHER) {
            const int gvoff = (int)((int64_t)p.gidx[min(row, p.M - 1)] * p.gs_g * 4), gsn4 = (int)(p.gs_n * 4);
#pragma unroll
            for (int nt = 0; nt < (GATHER ? NT : 1); ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gpre[nt][r] = sk_load1(rg, gvoff + min(nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, p.N - 1) * gsn4, 0, 0);
        }
        // one output tile after the other: the stores of tile nt leave while the MFMAs of tile nt + 1 run (with all tiles' MFMAs
        // first, every wave of the chip computed and then stored at the same time, and neither phase hid the other)
#ifdef DN_SK_GOUTER
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wimg + ((nt * KG + g) * 64 + lane) * 4);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], cur[g][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], cur[g][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], cur[g][2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], cur[g][3], acc[nt], 0, 0, 0);
                if (((nt + g * NT) & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#ifndef DN_SK_GOUTER
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wimg + ((nt * KG + g) * 64 + lane) * 4);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], cur[g][0], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], cur[g][1], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], cur[g][2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], cur[g][3], acc[nt], 0, 0, 0);
                if ((g & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // at most two weight quads in flight
            }
#endif
            // ---- epilogue: lane = row, register 4a + i <-> column nt*32 + 8a + 4h + i; columns < N only (a quad that straddles N is
            // stored element by element); rows past M are dropped by the range check
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n0 = nt * 32 + 8 * a + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bimg + n0);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[nt][4 * a + i] + b[i];
                    if (GATHER) v[i] += gpre[nt][4 * a + i];
                }
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = act_apply(p.act, v[i], p.slope);
                }
                sk_store_quad(v, rc, coff + nt * 128 + a * 32, n0, p.N);
            }
        }
        if (!more) break;
        tile = ntile;
        if constexpr (PF) {
#pragma unroll
            for (int g = 0; g < KG; ++g) cur[g] = nxt[g];
        } else {
            load_rows(cur, tile);
        }
    }
}

template <int NT, int KG, int WPS, bool GATHER, bool DACT = false>
static hipError_t launch(const Args &a, hipStream_t st) {
    const size_t lds = ((size_t)NT * KG * 256 + NT * 32) * sizeof(float);
    const int64_t tiles = ((int64_t)a.M + 31) / 32;
    // as many workgroups per CU as registers and LDS allow (measured at N = K = 64, M = 81920: 1 / 2 / 3 per CU = 27 / 23 / 20 us:
    // the kernel is bound by the latency of a wave's tile, not by issue slots)
    static const int per_cu_env = getenv("DAGNET_SKINNY_WGS_PER_CU") ? atoi(getenv("DAGNET_SKINNY_WGS_PER_CU")) : 8;
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(WPS, per_cu_env), (160 * 1024) / (int64_t)lds));
    const int64_t blocks = std::min<int64_t>((tiles + 3) / 4, 256 * per_cu);
    hipError_t e = hipSuccess;
    if (lds > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(skinny_kernel<NT, KG, WPS, GATHER, DACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((skinny_kernel<NT, KG, WPS, GATHER, DACT>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace skinny

// -> hipErrorNotSupported when the shape is not a skinny one or its rows are not padded to 16 bytes (the caller then takes
// the LDS-tiled kernel).  Padding contract: lda >= 4 ceil(K/4), ldc >= 4 ceil(N/4), both multiples of 4, bases 16-byte aligned;
// only columns < N of C (and < K of dP) are written: C may be a column slice of a wider buffer.
extern "C" hipError_t dn_gemm_skinny(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, float *C,
                                     int64_t ldc, int M, int N, int K, int act, float slope, int accumulate, const float *gather,
                                     const int32_t *gidx, int64_t gs_g, int64_t gs_n, hipStream_t st) {
    // K > 96: only the GRU gates' input gradient (K = 192 -> 64 columns).  Alone, the accumulating form is faster here than in the
    // LDS-tiled kernel (46 against 57 us; 43 against 38 without accumulation), but inside the training step's graph, beside the
    // kernels of the side streams, the step is 1 % slower with it (112.9 against 111.9 ms): off unless DAGNET_SKINNY_K192=1
    static const bool k192 = getenv("DAGNET_SKINNY_K192") && atoi(getenv("DAGNET_SKINNY_K192")) != 0;
    if (M < 2048 || N > 192 || N < 1 || K < 1 || K > 192 || (K > 96 && (N > 64 || !k192 || gather))) return hipErrorNotSupported;
    if ((int64_t)M * lda * 4 >= 0x7fffffffLL || (int64_t)M * ldc * 4 >= 0x7fffffffLL) return hipErrorNotSupported;     // 32-bit buffer offsets
    if (gather && (gs_g < 0 || gs_n < 0 || gs_g > 4096 || gs_n > 4096)) return hipErrorNotSupported;      // 32-bit table offsets (indices < 2^16)
    if ((lda & 3) || (ldc & 3) || lda < ((K + 3) & ~3) || ldc < ((N + 3) & ~3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
        (reinterpret_cast<uintptr_t>(C) & 15))
        return hipErrorNotSupported;
    skinny::Args a{A, lda, B, ldb, bias, C, ldc, M, N, K, slope, gather, gidx, gs_g, gs_n, act, accumulate, transb, nullptr, 0, nullptr, 0, 0, 0.f};
    const int nt = (N + 31) / 32, kg = (K + 7) / 8;
    if (gather && nt > 3) return hipErrorNotSupported;
#define SK_CASE(NTV, KGV, WPS)                                                                       \
    if (nt <= NTV && kg <= KGV) {                                                                    \
        if constexpr (NTV <= 3) {                                                                    \
            if (gather) return skinny::launch<NTV, KGV, 2, true>(a, st);                               \
        }                                                                                            \
        return skinny::launch<NTV, KGV, WPS, false>(a, st);                                          \
    }
    // (n-tiles, k-groups, workgroups per CU the registers allow: accumulators + two A tiles; 2 with the table term)
    SK_CASE(1, 4, 4) SK_CASE(1, 8, 4) SK_CASE(1, 12, 4)
    SK_CASE(2, 4, 4) SK_CASE(2, 8, 4) SK_CASE(2, 12, 3)
    SK_CASE(3, 4, 4) SK_CASE(3, 8, 3) SK_CASE(3, 12, 3)
    SK_CASE(6, 4, 3) SK_CASE(6, 8, 2) SK_CASE(6, 12, 2)
    if (nt <= 2 && kg <= 24) return skinny::launch<2, 24, 3, false>(a, st);
#undef SK_CASE
    return hipErrorNotSupported;
}

// C[M,N] = [C +] (dY * act'(Y)) op(B), and P = dY * act'(Y): the input-gradient product of a layer y = act(x W^T + b) taken
// straight from the gradient of its OUTPUT (dY [M,K], Y [M,K] the saved output, K = the layer's width), which replaces a pass
// of its own over dY and Y (train_ops.hip act_bwd_kernel) plus the re-read of its result.  Same shape rules as dn_gemm_skinny,
// for all three matrices of the first operand; -> hipErrorNotSupported otherwise (the caller runs the two steps separately).
extern "C" hipError_t dn_gemm_skinny_dact(const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *P, int64_t ldp, int dact,
                                          float dslope, const float *B, int64_t ldb, int transb, float *C, int64_t ldc, int M, int N,
                                          int K, int accumulate, hipStream_t st) {
    if (M < 2048 || N > 96 || K > 96 || N < 1 || K < 1) return hipErrorNotSupported;
    const int64_t lim = 0x7fffffffLL;
    if ((int64_t)M * ldd * 4 >= lim || (int64_t)M * ldy * 4 >= lim || (int64_t)M * ldp * 4 >= lim || (int64_t)M * ldc * 4 >= lim)
        return hipErrorNotSupported;
    const int k4 = (K + 3) & ~3;
    if ((ldd & 3) || (ldy & 3) || (ldp & 3) || (ldc & 3) || ldd < k4 || ldy < k4 || ldp < k4 || ldc < ((N + 3) & ~3)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(C)) & 15)
        return hipErrorNotSupported;
    skinny::Args a{dY, ldd, B, ldb, nullptr, C, ldc, M, N, K, 0.f, nullptr, nullptr, 0, 0, skinny::ACT_NONE, accumulate, transb,
                   Y, ldy, P, ldp, dact, dslope};
    const int nt = (N + 31) / 32, kg = (K + 7) / 8;
#define SK_DCASE(NTV, KGV, WPS) \
    if (nt <= NTV && kg <= KGV) return skinny::launch<NTV, KGV, WPS, false, true>(a, st);
    // registers: accumulators + dY of two tiles + Y of one
    SK_DCASE(1, 4, 4) SK_DCASE(1, 8, 4) SK_DCASE(1, 12, 3)
    SK_DCASE(2, 4, 4) SK_DCASE(2, 8, 3) SK_DCASE(2, 12, 2)
    SK_DCASE(3, 4, 4) SK_DCASE(3, 8, 3) SK_DCASE(3, 12, 2)
#undef SK_DCASE
    return hipErrorNotSupported;
}
