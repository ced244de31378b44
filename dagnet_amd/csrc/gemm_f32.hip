// fp32 MFMA GEMMs for the layer-by-layer paths (wide goal head of SDD-shaped DAG-Net, training step).
//
//   NT:  C[M,N] = act(A[M,K] . W[N,K]^T + bias[N])            nn.Linear forward        (models/dagnet/model.py:60-113)
//   NN:  C[M,N] = A[M,K] . B[K,N]                             x @ W of the graph layers, input gradients dY . W
//   TN:  C[N,K] = sum_m A[m,N] . B[m,K]   (split over m)      weight gradients
//
// 256 threads = 4 waves per workgroup; every wave owns TM x TN accumulator tiles of v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulation).  Operands are staged through LDS in k-chunks of 32, double buffered,
// register-staged global loads (float4 per lane, 128 B per 8 lanes).  The k index inside a chunk is permuted
// (MFMA number 4j+i of a chunk consumes k = 8j + 4h + i in lane half h) so that each lane fetches its operands as
// one ds_read_b128 per 4 MFMAs; any permutation is legal because both operands use the same one.
// Row stride of the LDS tiles is 36 floats: 16 consecutive rows x 16 B then cover all 64 banks exactly once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

namespace gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;        // k-chunk
constexpr int LDT = BK + 4;   // LDS row stride (floats)

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_RELU = 3 };

struct NtArgs {
    const float *A; int64_t lda;      // [M, K]
    const float *W; int64_t ldw;      // [N, K]
    const float *bias;                // [N] or null
    float *C; int64_t ldc;            // [M, N]
    int M, N, K;
    float slope;                      // LeakyReLU slope
    // optional one-hot input folded into the epilogue: C[m][n] += gather[gidx[m] * gs_g + n * gs_n]
    const float *gather; const int32_t *gidx; int64_t gs_g, gs_n;
    int act;                          // ACT_*
    int accumulate;                   // C = act(C + A W^T + ...)
};

__device__ __forceinline__ float act_apply(int act, float v, float slope) {
    if (act == ACT_LRELU) return fmaxf(v, v * slope);
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// Wave grid WM x WN (WM*WN == 4), per-wave TM x TN tiles of 32x32.  Tile of the workgroup: (WM*TM*32) x (WN*TN*32).
// VECA / VECB: float4 staging loads of that operand (needs a 16-byte aligned base, a row stride and an extent along the
// load direction that are multiples of 4); else scalar loads.
// TRANSB: the second operand is B[K, N] row-major (ldw = its row stride) instead of W[N, K].
template <int WM, int WN, int TM, int TN, bool VECA, bool VECB, bool TRANSB>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NtArgs p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LA = BM * 8 / 256, LB = BN * 8 / 256;      // float4 loads per thread and chunk
    static_assert(LA >= 1 && LB >= 1, "tile too small for 256 threads");
    extern __shared__ float smem[];
    float *sA = smem;                       // [2][BM][LDT]
    float *sB = smem + 2 * BM * LDT;        // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware tile order: workgroup w runs on XCD w % 8 (round-robin dispatch), each XCD with its own L2.  Give every
    // XCD a contiguous range of logical tiles, column blocks fastest, so that the nbn workgroups that read the same A
    // rows run back to back on ONE XCD (measured before: 4x the algorithmic HBM reads on the 1920 -> 960 layer, every
    // XCD fetching every A row block).  W is small enough to stay resident in every L2.
    // Every workgroup walks a strided list of tiles of its XCD's range and prefetches the first chunk of the next tile while
    // it computes the last chunk of the current one (one tile per workgroup: all workgroups of a CU load, then all compute).
    const int nbn = (p.N + BN - 1) / BN;
    const int64_t total = (int64_t)((p.M + BM - 1) / BM) * nbn, per_xcd = (total + 7) / 8;
    const int64_t xbeg = (int64_t)(blockIdx.x & 7) * per_xcd, xend = min(xbeg + per_xcd, total);
    const int64_t tstride = gridDim.x >> 3;
    int64_t tile = xbeg + (blockIdx.x >> 3);
    if (tile >= xend) return;
    int m0 = (int)(tile / nbn) * BM, n0 = (int)(tile % nbn) * BN;

    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;          // this thread's (row, k) inside a 32-row slab of a chunk
    f32x4 ga[LA], gb[LB];
    // 4 consecutive elements of row r starting at column c of a [nr, nc] matrix, zero outside
    auto ld4 = [&](const float *base, int64_t ld, int r, int nr, int c, int nc, bool VEC) {
        f32x4 v = {0, 0, 0, 0};
        if (r < nr) {
            const float *q = base + (int64_t)r * ld + c;
            if (VEC) {
                if (c < nc) v = *reinterpret_cast<const f32x4 *>(q);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c + i < nc) v[i] = q[i];
            }
        }
        return v;
    };
    constexpr int BN4 = BN / 4;
    auto gload = [&](int m0, int n0, int k0) {
#pragma unroll
        for (int i = 0; i < LA; ++i) ga[i] = ld4(p.A, p.lda, m0 + lrow + 32 * i, p.M, k0 + lc4, p.K, VECA);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (TRANSB) {      // element e of the [32 k][BN n] slab: 4 consecutive n of one k
                const int e = tid + 256 * i, kr = e / BN4, c4 = (e % BN4) * 4;
                gb[i] = ld4(p.W, p.ldw, k0 + kr, p.K, n0 + c4, p.N, VECB);
            } else {
                gb[i] = ld4(p.W, p.ldw, n0 + lrow + 32 * i, p.N, k0 + lc4, p.K, VECB);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4 *>(sA + (buf * BM + lrow + 32 * i) * LDT + lc4) = ga[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (TRANSB) {
                const int e = tid + 256 * i, kr = e / BN4, c4 = (e % BN4) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) sB[(buf * BN + c4 + c) * LDT + kr] = gb[i][c];
            } else {
                *reinterpret_cast<f32x4 *>(sB + (buf * BN + lrow + 32 * i) * LDT + lc4) = gb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(m0, n0, 0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int64_t ntile = tile + tstride;
        const bool more = ntile < xend;
        const int nm0 = more ? (int)(ntile / nbn) * BM : 0, nn0 = more ? (int)(ntile % nbn) * BN : 0;
        for (int kc = 0; kc < nk; ++kc) {
            const bool last = kc + 1 == nk, pf = !last || more;          // uniform over the workgroup
            if (!last) gload(m0, n0, (kc + 1) * BK);
            else if (more) gload(nm0, nn0, 0);
            const float *a_base = sA + (buf * BM + (wm * TM) * 32 + lr) * LDT + 4 * lh;
            const float *b_base = sB + (buf * BN + (wn * TN) * 32 + lr) * LDT + 4 * lh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const f32x4 *>(a_base + a * 32 * LDT + 8 * j);
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[b] = *reinterpret_cast<const f32x4 *>(b_base + b * 32 * LDT + 8 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][i], fb[b][i], acc[a][b], 0, 0, 0);
            }
            if (pf) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }

        // epilogue: lane = output column, register r <-> row (r&3) + 8(r>>2) + 4h of the 32x32 tile
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + (wn * TN + b) * 32 + lr;
            const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int rbase = m0 + (wm * TM + a) * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < p.M && col < p.N) {
                        float v = acc[a][b][r] + bv;
                        if (p.gather) v += p.gather[(int64_t)p.gidx[row] * p.gs_g + (int64_t)col * p.gs_n];
                        float *dst = p.C + (int64_t)row * p.ldc + col;
                        if (p.accumulate) v += *dst;
                        *dst = act_apply(p.act, v, p.slope);
                    }
                    acc[a][b][r] = 0.f;
                }
            }
        }
        if (!more) break;
        tile = ntile;
        m0 = nm0;
        n0 = nn0;
    }
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 tiles, software-pipelined variant for the wide layers (N > 96, K a multiple of 32, float4-aligned operands,
// W[N, K]): the layer that dominates the SDD-shaped goal head (1920 -> 960 over K * N rows).
//
// Same tile, LDS layout and MFMA order as gemm_nt_kernel<2, 2, 2, 2>, but the staging around the MFMAs is branch-free and
// ordered so that one wave keeps the MFMA pipe busy by itself:
//   * raw buffer loads (one descriptor per operand and tile, 32-bit per-lane offset fixed for the whole kernel, the k position in
//     the scalar offset): no address arithmetic and no bounds branches per load - rows past the end of the matrix are cut off
//     by the descriptor's range check and read as 0;
//   * chunk q + 1 (loaded during chunk q - 1, long since landed) is written to the other LDS buffer and chunk q + 2 is requested
//     at the TOP of chunk q, in front of its 64 MFMAs; the operand fragments of k-group j + 1 are read while group j multiplies.
//     What is left exposed per chunk is one barrier.
// (The previous kernel issued its loads under 16 exec-masked branches, which also kept the compiler from moving anything across
// them; two workgroups per CU then ran in convoy - both loading, both multiplying - at 0.72 of the fp32 MFMA peak.)
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 gp_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void gp_store1(float data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");

__device__ __forceinline__ i32x4 gp_rsrc(const float *base, int64_t bytes) {
    union { const float *p; int w[2]; } u;
    u.p = base;
    const int64_t lim = bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes);
    return i32x4{__builtin_amdgcn_readfirstlane(u.w[0]), __builtin_amdgcn_readfirstlane(u.w[1]), __builtin_amdgcn_readfirstlane((int)lim),
                 0x00020000};
}

// tanh on the hardware exp2 / rcp: (1 - e) / (1 + e) with e = exp(-2|v|) from 0.1 up (absolute error ~1e-7 = relative below 1e-6
// there), the odd series v (1 - v^2/3 + 2 v^4/15) below (truncation 5e-8 relative at 0.1) - the plain 1 - 2 / (1 + exp(2v)) form of
// round 2 cancelled near zero (relative error 1e-3 at |v| = 1e-4, where the other epilogues' tanhf is exact: ADVICE r2)
__device__ __forceinline__ float tanh_hw(float v) {
    const float a = fabsf(v), v2 = v * v;
    const float e = __builtin_amdgcn_exp2f(a * -2.8853900817779268f);
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    const float small = v * fmaf(v2, fmaf(v2, 0.13333334f, -0.33333334f), 1.0f);
    return a < 0.1f ? small : copysignf(big, v);
}

// BKP: k-chunk (32 or 16), WPC: workgroups per CU the registers are budgeted for (LDS: 2 * 256 * (BKP + 4) * 4 bytes each)
//
// BF3 (k-chunks of 16): the same products on the bf16 matrix pipe.  Both operands are split where they are staged into LDS, each
// fp32 value into three bf16 terms by truncation (v_and / v_sub / v_perm: 8 + 8 + 8 significant bits, so hi + mid + lo is the
// value exactly), and a chunk is six v_mfma_f32_32x32x16_bf16 per accumulator tile - hi.hi, hi.mid, mid.hi, mid.mid, hi.lo,
// lo.hi; the three dropped products are below 2^-24 of the largest - with fp32 accumulation: 24 x 32 clocks per wave and chunk
// where the fp32 form takes 32 x 64.  Measured on a chain of layers against fp64 (tools/ubench/split_bf16.hip,
// profiles/r02_split_bf16.txt): half the error of v_mfma_f32_32x32x2_f32.  A non-finite input comes out as NaN (inf - inf
// in the split) where the fp32 form would carry the infinity.
// LDS: [buffer 2][term 3][row 128][16 bf16 + pad] per operand, row stride 12 dwords (16 rows x 16 B cover the 64 banks once).
//
// SPLIT = 2 (k-chunks of 16): two fp16 terms per value, rounded to nearest (hi = fp16(v), lo = fp16(v - hi): v_cvt_pk_f16_f32 +
// v_fma_mixlo/hi_f16, three instructions per pair - see v2_common.h split2 for how the compiler is made to select them), THREE
// v_mfma_f32_32x32x16_f16 per accumulator tile and chunk (lo.hi, hi.lo, hi.hi), two LDS planes per operand (48 KB: three
// workgroups per CU).  What it gives up against the three-bf16 form: range - fp16 overflows at 65 504 and the lo term is a
// subnormal below |v| = 0.125, so a value carries an ABSOLUTE representation error of up to 2^-25 = 3e-8 instead of fp32's
// relative 6e-8.  Taken only where the caller says its operands are of order one or below (the goal head of the evaluation
// path: ReLU / softmax / tanh outputs against weights of order 1 / sqrt(K)): dn_gemm_nt_gather; dn_gemm keeps three bf16 terms.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int BF3_RS = 12;      // dwords per row of a term plane
constexpr size_t split_lds(int terms, int bm = 128, int bn = 128) { return (size_t)2 * terms * (bm + bn) * BF3_RS * 4; }      // bytes: two operands, two buffers
// SPLIT: 0 = fp32 MFMA, 3 = three bf16 terms / six products, 2 = two fp16 terms / three products
// BM: rows of the workgroup tile (128, or 256 for the split forms: every wave then owns 128 x 64 - 24 KB of operands staged per
// 2 x the MFMAs of the 128-row tile, and half the W requests and barriers per MFMA)
// BN: columns of the workgroup tile (128 = four waves, 256 = eight waves: 512 threads, 256 x 256 with BM = 256 - A is then fetched
// by 4 instead of 8 column blocks and 32 KB are staged per 192 MFMAs where the 256 x 128 tile stages 24 KB per 96)
template <int BKP, int WPC, int SPLIT = 0, int BM = 128, int BN = 128>
__global__ __launch_bounds__((BN / 64) * 128, WPC) void gemm_nt_pipe_kernel(const NtArgs p) {
    constexpr int WN = BN / 64, NTHR = WN * 128;                   // waves along N (two along M), threads
    constexpr bool BF3 = SPLIT != 0;
    constexpr int NPL = SPLIT ? SPLIT : 3;
    constexpr int PLA = BM * BF3_RS, PLB = BN * BF3_RS;            // dwords: one term plane of A / of W
    constexpr int OPA = NPL * PLA, OPB = NPL * PLB;                // dwords: one operand buffer
    static_assert(!BF3 || BKP == 16, "the split forms take one 16-deep k block per chunk");
    static_assert(BF3 || (BM == 128 && BN == 128), "the fp32 form has the 128 x 128 tile only");
    constexpr int TM = BM / 64, TN = 2;
    constexpr int LDP = BKP + 4;                 // LDS row stride (floats): 16 rows x 16 B cover the 64 banks once for 36 and for 20
    constexpr int TPR = BKP / 4, RPP = NTHR / TPR, NL = BN / RPP, NLA = BM / RPP;      // loader: threads per row, rows per pass, passes (W / A)
    extern __shared__ float smem[];
    float *sA = smem;                                             // [2][BM][LDP]
    float *sB = smem + (BF3 ? 2 * OPA : 2 * BM * LDP);            // [2][BN][LDP]
    unsigned *uA = reinterpret_cast<unsigned *>(sA), *uB = reinterpret_cast<unsigned *>(sB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int nbn = (p.N + BN - 1) / BN;
    const int64_t total = (int64_t)((p.M + BM - 1) / BM) * nbn, per_xcd = (total + 7) / 8;
    const int64_t xbeg = (int64_t)(blockIdx.x & 7) * per_xcd, xend = min(xbeg + per_xcd, total);
    const int64_t tstride = gridDim.x >> 3;
    int64_t tile = xbeg + (blockIdx.x >> 3);
    if (tile >= xend) return;
    const int nk = p.K / BKP;

    // loader: this thread's float4 of row tid / TPR + RPP i, k columns 4 (tid % TPR) .. + 3 of a chunk
    // split forms: a thread writes 8 bytes per term plane at row * 12 dwords + 2 * (tid % 4), and a ds_write_b64 goes out 32 lanes at
    // a time - eight consecutive rows put rows 0 / 5, 1 / 6, 2 / 7 on the same banks (60 = 5 * 12 wraps onto row 0: SQ_LDS_BANK_CONFLICT
    // was a third of the LDS's active cycles); eight rows of ONE parity tile the 64 banks exactly, so lanes 0-31 take the even rows
    // of the wave's 16 and lanes 32-63 the odd ones
    const int lrow = BF3 ? wave * 16 + ((lane >> 2) & 7) * 2 + (lane >> 5) : tid / TPR, lc4 = (tid % TPR) * 4;
    const int lda4 = (int)p.lda * 4, ldw4 = (int)p.ldw * 4;
    const int voa = lrow * lda4 + lc4 * 4, vow = lrow * ldw4 + lc4 * 4;
    f32x4 ga[NLA], gb[NL];
    int64_t ltile = tile;       // tile and chunk the loader is at
    int lkc = 0;
    i32x4 ra, rw;
    auto set_desc = [&](int64_t t) {
        if (t < xend) {
            const int m0 = (int)(t / nbn) * BM, n0 = (int)(t % nbn) * BN;
            ra = gp_rsrc(p.A + (int64_t)m0 * p.lda, (int64_t)(p.M - m0) * lda4);
            rw = gp_rsrc(p.W + (int64_t)n0 * p.ldw, (int64_t)(p.N - n0) * ldw4);
        } else {                // past this workgroup's last tile: empty buffers, every load returns 0 without touching memory
            ra = gp_rsrc(p.A, 0);
            rw = gp_rsrc(p.W, 0);
        }
    };
    auto gload = [&]() {        // request the loader's chunk, then step the loader
        const int so = lkc * (BKP * 4);
#pragma unroll
        for (int i = 0; i < NLA; ++i) ga[i] = gp_load4(ra, voa + i * RPP * lda4, so, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) gb[i] = gp_load4(rw, vow + i * RPP * ldw4, so, 0);
        if (++lkc == nk) {
            lkc = 0;
            ltile += tstride;
            set_desc(ltile);
        }
    };
    // four consecutive k of one row -> 4 bf16 / fp16 (8 bytes) in each of the term planes
    auto split_store = [&](unsigned *dst, f32x4 v, int pl) {
        if constexpr (SPLIT == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            float m1 = -1.0f;
            asm volatile("" : "+s"(m1));
            unsigned hi[2], lo[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f2 x = {v[2 * j], v[2 * j + 1]};
                const h2 h = __builtin_convertvector(x, h2);
                hi[j] = __builtin_bit_cast(unsigned, h);
                h2 l;
                _Float16 l0 = (_Float16)__builtin_fmaf((float)h[0], m1, x[0]);
                asm("" : "+v"(l0));
                l[0] = l0;
                l[1] = (_Float16)__builtin_fmaf((float)h[1], m1, x[1]);
                lo[j] = __builtin_bit_cast(unsigned, l);
            }
            *reinterpret_cast<uint2 *>(dst) = uint2{hi[0], hi[1]};
            *reinterpret_cast<uint2 *>(dst + pl) = uint2{lo[0], lo[1]};
            return;
        }
        float e[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            unsigned u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = __builtin_bit_cast(unsigned, e[k]);
            const uint2 pk = {__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
            *reinterpret_cast<uint2 *>(dst + t * pl) = pk;
            if (t < 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] -= __builtin_bit_cast(float, u[k] & 0xffff0000u);
            }
        }
    };
    auto sstore = [&](int buf) {
        if constexpr (BF3) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) split_store(uA + buf * OPA + (lrow + RPP * i) * BF3_RS + (lc4 >> 1), ga[i], PLA);
#pragma unroll
            for (int i = 0; i < NL; ++i) split_store(uB + buf * OPB + (lrow + RPP * i) * BF3_RS + (lc4 >> 1), gb[i], PLB);
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4 *>(sA + (buf * BM + lrow + RPP * i) * LDP + lc4) = ga[i];
#pragma unroll
            for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4 *>(sB + (buf * BN + lrow + RPP * i) * LDP + lc4) = gb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    set_desc(ltile);
    gload();
    sstore(0);
    gload();                    // chunk 1 (or the next tile's chunk 0, or zeros) waits in registers
    __syncthreads();
    int buf = 0, kc = 0;
    int m0 = (int)(tile / nbn) * BM, n0 = (int)(tile % nbn) * BN;
    while (true) {
      if constexpr (SPLIT == 2) {
        const unsigned *a_base = uA + buf * OPA + ((wm * TM) * 32 + lr) * BF3_RS + 4 * lh;
        const unsigned *b_base = uB + buf * OPB + ((wn * TN) * 32 + lr) * BF3_RS + 4 * lh;
        f16x8 fa[2][2], fb[2][TN];        // W fragments of the chunk, A fragments of two row tiles (the one in use, the next)
        auto rd = [](const unsigned *q) { return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(q)); };
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[0][t] = rd(a_base + t * PLA);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[t][b] = rd(b_base + t * PLB + b * 32 * BF3_RS);
        }
        sstore(buf ^ 1);        // chunk q + 1, split on the way
        gload();                // chunk q + 2
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};      // small products first
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (a + 1 < TM) {
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[(a + 1) & 1][t] = rd(a_base + t * PLA + (a + 1) * 32 * BF3_RS);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a & 1][PA[q]], fb[PB[q]][b], acc[a][b], 0, 0, 0);
        }
      } else if constexpr (BF3) {
        // this chunk's operands, in the order of their first use: 8 k of one row per lane and term
        const unsigned *a_base = uA + buf * OPA + ((wm * TM) * 32 + lr) * BF3_RS + 4 * lh;
        const unsigned *b_base = uB + buf * OPB + ((wn * TN) * 32 + lr) * BF3_RS + 4 * lh;
        bf16x8 fa[3][TM], fb[3][TN];
        auto rd = [](const unsigned *q) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(q)); };
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[t][a] = rd(a_base + t * PLA + a * 32 * BF3_RS);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[t][b] = rd(b_base + t * PLB + b * 32 * BF3_RS);
        }
        sstore(buf ^ 1);        // chunk q + 1, split on the way
        gload();                // chunk q + 2
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][a], fb[PB[q]][b], acc[a][b], 0, 0, 0);
      } else {
        sstore(buf ^ 1);        // chunk q + 1 -> the buffer that chunk q - 1 was read from (all waves passed the barrier since)
        gload();                // chunk q + 2
        const float *a_base = sA + (buf * BM + (wm * TM) * 32 + lr) * LDP + 4 * lh;
        const float *b_base = sB + (buf * BN + (wn * TN) * 32 + lr) * LDP + 4 * lh;
        f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[0][a] = *reinterpret_cast<const f32x4 *>(a_base + a * 32 * LDP);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[0][b] = *reinterpret_cast<const f32x4 *>(b_base + b * 32 * LDP);
#pragma unroll
        for (int j = 0; j < BKP / 8; ++j) {
            if (j < BKP / 8 - 1) {
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[(j + 1) & 1][a] = *reinterpret_cast<const f32x4 *>(a_base + a * 32 * LDP + 8 * (j + 1));
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[(j + 1) & 1][b] = *reinterpret_cast<const f32x4 *>(b_base + b * 32 * LDP + 8 * (j + 1));
            }
            // keep the reads of group j + 1 in FRONT of group j's 16 MFMAs (left alone, the scheduler sinks them to two MFMAs
            // before their use and the wave then waits out the LDS latency in every group)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1][a][i], fb[j & 1][b][i], acc[a][b], 0, 0, 0);
        }
      }
        if (++kc == nk) {
            // epilogue: lane = output column, register r <-> row (r&3) + 8(r>>2) + 4h of the 32x32 tile
            if (!p.gather && !p.accumulate) {
                // plain layer: one buffer descriptor per 32-row slab, the row and column of an element in the 32-bit offset
                // (rows past M and columns past N fall outside the descriptor's range: dropped by the hardware), one
                // v_add per address; tanh through exp2 / rcp (absolute error ~1e-7)
                const int ldc4 = (int)p.ldc * 4;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int row0 = m0 + (wm * TM + a) * 32;
                    const i32x4 rc = gp_rsrc(p.C + (int64_t)row0 * p.ldc, (int64_t)(p.M - row0) * ldc4);
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const int col = n0 + (wn * TN + b) * 32 + lr;
                        const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
                        const int vo = col < p.N ? 4 * lh * ldc4 + col * 4 : (int)0x80000000u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[a][b][r] + bv;
                            if (p.act == ACT_LRELU) v = fmaxf(v, v * p.slope);
                            else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                            else if (p.act == ACT_TANH)
                                v = tanh_hw(v);
                            gp_store1(v, rc, vo + ((r & 3) + 8 * (r >> 2)) * ldc4, 0, 0);
                            acc[a][b][r] = 0.f;
                        }
                    }
                }
            } else {
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + (wn * TN + b) * 32 + lr;
                const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int rbase = m0 + (wm * TM + a) * 32 + 4 * lh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < p.M && col < p.N) {
                            float v = acc[a][b][r] + bv;
                            if (p.gather) v += p.gather[(int64_t)p.gidx[row] * p.gs_g + (int64_t)col * p.gs_n];
                            float *dst = p.C + (int64_t)row * p.ldc + col;
                            if (p.accumulate) v += *dst;
                            *dst = act_apply(p.act, v, p.slope);
                        }
                        acc[a][b][r] = 0.f;
                    }
                }
            }
            }
            kc = 0;
            tile += tstride;
            if (tile >= xend) break;
            m0 = (int)(tile / nbn) * BM;
            n0 = (int)(tile % nbn) * BN;
        }
        __syncthreads();
        buf ^= 1;
    }
}

struct TnArgs {
    const float *A; int64_t lda;      // [M, N]  (e.g. dY)
    const float *B; int64_t ldb;      // [M, K]  (e.g. X)
    float *C; int64_t ldc;            // [splits][N, K] partial sums, split stride = N*ldc
    int M, N, K;
    int rows_per_split;               // multiple of 32
    const int32_t *bidx;              // not null: B is one-hot, B[m][k] = (bidx[m] == k)   (gradients of one-hot inputs)
    float *bias_part;                 // not null: [splits][N] column sums of A over this split's rows (bias gradient)
    int part_accumulate;              // add to the partial sums already in C / bias_part (several calls, one final reduction)
};

// C_split[n][k] = sum over this split's rows m of A[m][n] * B[m][k].  Workgroup tile 64 x 64 (2x2 waves, 1 tile each),
// reduction chunks of 32 rows; LDS tiles are [m][n] / [m][k] as loaded, operands fetched as ds_read_b32.
__global__ __launch_bounds__(256, 4) void gemm_tn_kernel(const TnArgs p) {
    __shared__ float sA[2][32][64 + 1], sB[2][32][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int nbk = (p.K + 63) / 64;
    const int n0 = (blockIdx.x / nbk) * 64, k0 = (blockIdx.x % nbk) * 64;
    const int mbeg = blockIdx.y * p.rows_per_split, mend = min(p.M, mbeg + p.rows_per_split);
    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;          // 16 rows x 16 float4 per pass, 2 passes per 32-row chunk
    const bool vecA = (p.lda & 3) == 0, vecB = (p.ldb & 3) == 0;
    f32x4 ga[2], gb[2];
    f32x4 bsum = {0, 0, 0, 0};                               // this thread's share of the column sums of A
    const bool want_bias = p.bias_part != nullptr && k0 == 0;
    auto ld4 = [&](const float *base, int64_t ld, int row, int col, int ncol, bool vec) {
        f32x4 v = {0, 0, 0, 0};
        if (row < mend) {
            const float *q = base + (int64_t)row * ld + col;
            if (vec && col + 3 < ncol) v = *reinterpret_cast<const f32x4 *>(q);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (col + i < ncol) v[i] = q[i];
            }
        }
        return v;
    };
    auto gload = [&](int m) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ga[i] = ld4(p.A, p.lda, m + lrow + 16 * i, n0 + lc4, p.N, vecA);
            if (want_bias) bsum += ga[i];
            if (p.bidx) {
                const int row = m + lrow + 16 * i;
                const int hot = row < mend ? p.bidx[row] - (k0 + lc4) : -1;
                gb[i] = f32x4{hot == 0 ? 1.f : 0.f, hot == 1 ? 1.f : 0.f, hot == 2 ? 1.f : 0.f, hot == 3 ? 1.f : 0.f};
            } else {
                gb[i] = ld4(p.B, p.ldb, m + lrow + 16 * i, k0 + lc4, p.K, vecB);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sA[buf][lrow + 16 * i][lc4 + c] = ga[i][c];
                sB[buf][lrow + 16 * i][lc4 + c] = gb[i][c];
            }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = (mend - mbeg + 31) / 32;
    if (nchunk > 0) {
        gload(mbeg);
        sstore(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(mbeg + (c + 1) * 32);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[buf][2 * kk + lh][wm * 32 + lr], sB[buf][2 * kk + lh][wn * 32 + lr], acc, 0, 0, 0);
        if (c + 1 < nchunk) sstore(buf ^ 1);
        __syncthreads();
    }
    if (want_bias) {       // 16 row lanes x 64 columns -> column sums (the loop above ended with a barrier: sA is free)
#pragma unroll
        for (int c = 0; c < 4; ++c) sA[0][lrow][lc4 + c] = bsum[c];
        __syncthreads();
        if (tid < 64 && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += sA[0][r][tid];
            float *bp = p.bias_part + (int64_t)blockIdx.y * p.N + n0 + tid;
            *bp = p.part_accumulate ? *bp + t : t;
        }
    }
    float *C = p.C + (int64_t)blockIdx.y * p.N * p.ldc;
    const int col = k0 + wn * 32 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = n0 + wm * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
        if (row < p.N && col < p.K) {
            float *dst = C + (int64_t)row * p.ldc + col;
            *dst = p.part_accumulate ? *dst + acc[r] : acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same product without LDS in the main loop.  The 32x32x2 MFMA wants, from lane (lr, lh), one element of row m + lh of each
// operand; WHICH column a lane feeds only decides where its result lands.  So lane lr takes the NT consecutive floats at column
// NT * lr of A (one dwordxNT buffer load: a wave-load covers two whole rows) and the KT floats at column KT * lr of B, and output
// tile (cn, ck) holds the columns {NT * i + cn} x {KT * j + ck}.  Every wave streams its own row range straight from HBM into MFMA
// operands - no staging stores, no barriers - with the loads of the next block of 2 * P rows in flight under the current block's
// MFMAs.  Each load has its own descriptor (base and remaining bytes advance in SGPRs), so the hardware range check zeroes whatever
// lies beyond the wave's rows and the loop needs no tail code.  The four waves of a workgroup then add their tiles through LDS
// in a fixed order and write one partial like gemm_tn_kernel does.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ float gd_load1(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ f32x2 gd_load2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ f32x3 gd_load3(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");

template <int NV>
__device__ __forceinline__ void gd_load(float (&d)[NV], const i32x4 r, const int vo) {
    static_assert(NV == 1 || NV == 2 || NV == 3 || NV == 4 || NV == 6, "columns per lane");
    if constexpr (NV == 1) d[0] = gd_load1(r, vo, 0, 0);
    if constexpr (NV == 2) { const f32x2 t = gd_load2(r, vo, 0, 0); d[0] = t[0]; d[1] = t[1]; }
    if constexpr (NV == 3) { const f32x3 t = gd_load3(r, vo, 0, 0); d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; }
    if constexpr (NV == 4) { const f32x4 t = gp_load4(r, vo, 0, 0); d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3]; }
    if constexpr (NV == 6) {
        const f32x3 t = gd_load3(r, vo, 0, 0), u = gd_load3(r, vo + 12, 0, 0);
        d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = u[0]; d[4] = u[1]; d[5] = u[2];
    }
}

constexpr int tnd_wpc(int NT, int KT) {           // register budget in workgroups per CU: accumulators + two operand blocks + addresses
    const int regs = 16 * NT * KT + 16 * (NT + KT) + 24;
    return regs <= 128 ? 4 : (regs <= 168 ? 3 : (regs <= 256 ? 2 : 1));
}

// NH = 2: the output columns are split in two halves of 32 * NT; waves 0-1 take the first and waves 2-3 the second half over the
// SAME rows, so B is fetched from HBM once and the second fetch is served by the cache (N = 192: the GRU's gate columns).
template <int NT, int KT, int NH>
__global__ __launch_bounds__(256, tnd_wpc(NT, KT)) void gemm_tn_direct_kernel(const TnArgs p) {
    constexpr int P = 8;                                    // row pairs per operand block
    constexpr int NP = 32 * NT * NH, KP = 32 * KT, WS = 4 / NH;   // WS: waves that share a column half (they split the rows)
    extern __shared__ float tsm[];                          // [NP][KP] tile sums, then [WS][NP] bias sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int half = wave / WS, sub = wave % WS, nh = p.N / NH;     // nh: columns per half
    const int mbeg = blockIdx.x * p.rows_per_split, mend = min(p.M, mbeg + p.rows_per_split);
    const int q = ((p.rows_per_split + WS - 1) / WS + 1) & ~1;      // rows per wave (even)
    const int w0 = __builtin_amdgcn_readfirstlane(min(mend, mbeg + sub * q));
    const int w1 = __builtin_amdgcn_readfirstlane(min(mend, w0 + q));
    const int64_t lda4 = p.lda * 4, ldb4 = p.ldb * 4;
    const int voa = NT * lr < nh ? (int)(lh * lda4) + (half * nh + NT * lr) * 4 : (int)0x80000000;     // a lane without columns reads zeros
    const int vob = KT * lr < p.K ? (int)(lh * ldb4) + KT * lr * 4 : (int)0x80000000;
    const bool want_bias = p.bias_part != nullptr;
    float a0[P][NT], b0[P][KT], a1[P][NT], b1[P][KT];
    f32x16 acc[NT][KT];
    float bs[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        bs[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    auto load = [&](float (&a)[P][NT], float (&b)[P][KT], const int row) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int r = row + 2 * i;
            gd_load<NT>(a[i], gp_rsrc(p.A + (int64_t)r * p.lda, (int64_t)(w1 - r) * lda4), voa);
            gd_load<KT>(b[i], gp_rsrc(p.B + (int64_t)r * p.ldb, (int64_t)(w1 - r) * ldb4), vob);
        }
    };
    auto mult = [&](const float (&a)[P][NT], const float (&b)[P][KT]) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
#pragma unroll
            for (int cn = 0; cn < NT; ++cn) {
#pragma unroll
                for (int ck = 0; ck < KT; ++ck) acc[cn][ck] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][cn], b[i][ck], acc[cn][ck], 0, 0, 0);
                if (want_bias) bs[cn] += a[i][cn];
            }
        }
    };
    load(a0, b0, w0);
    for (int row = w0; row < w1; row += 4 * P) {
        load(a1, b1, row + 2 * P);
        mult(a0, b0);
        load(a0, b0, row + 4 * P);
        mult(a1, b1);
    }
    // the waves of a column half add their tiles in wave order
    for (int w = 0; w < WS; ++w) {
        if (sub == w) {
#pragma unroll
            for (int cn = 0; cn < NT; ++cn)
#pragma unroll
                for (int ck = 0; ck < KT; ++ck)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int loc = NT * (4 * lh + (r & 3) + 8 * (r >> 2)) + cn, k = KT * lr + ck;
                        float *d = tsm + (half * nh + loc) * KP + k;
                        if (loc < nh) *d = w == 0 ? acc[cn][ck][r] : *d + acc[cn][ck][r];    // rows beyond a half belong to the next one
                    }
        }
        __syncthreads();
    }
    float *C = p.C + (int64_t)blockIdx.x * p.N * p.ldc;
    for (int i = tid; i < p.N * KP; i += 256) {
        const int n = i / KP, k = i % KP;
        if (k < p.K) {
            float *dst = C + (int64_t)n * p.ldc + k;
            *dst = p.part_accumulate ? *dst + tsm[i] : tsm[i];
        }
    }
    if (want_bias) {
        __syncthreads();
#pragma unroll
        for (int cn = 0; cn < NT; ++cn) {
            const float t = bs[cn] + __shfl_xor(bs[cn], 32);
            if (lh == 0 && NT * lr < nh) tsm[sub * NP + half * nh + NT * lr + cn] = t;
        }
        __syncthreads();
        if (tid < p.N) {
            float t = tsm[tid];
#pragma unroll
            for (int w = 1; w < WS; ++w) t += tsm[w * NP + tid];
            float *bp = p.bias_part + (int64_t)blockIdx.x * p.N + tid;
            *bp = p.part_accumulate ? *bp + t : t;
        }
    }
}

// columns per lane for an operand (or operand half) of `n` columns (0: the direct kernel does not take it)
static int tnd_cols(int n, int64_t ld) {
    const int v = (n + 31) / 32;
    if (v < 1 || v > 3 || n % v != 0) return 0;
    if (v == 2 && (ld & 1)) return 0;
    return v;
}

// X(NT, KT, NH): the instantiations (NH = 2 above 96 output columns)
#define DN_TND_LIST X(1, 1, 1) X(1, 2, 1) X(1, 3, 1) X(2, 1, 1) X(2, 2, 1) X(2, 3, 1) X(3, 1, 1) X(3, 2, 1) X(3, 3, 1) \
                    X(2, 1, 2) X(2, 2, 2) X(3, 1, 2) X(3, 2, 2)
static size_t tnd_lds(int nt, int kt, int nh) { return sizeof(float) * std::max(32 * nt * nh * 32 * kt, 4 * 32 * nt * nh); }

// workgroups of one instantiation that are resident at once (what its registers and LDS allow on this device): the number of splits,
// so that one round of workgroups covers the rows; 0 = no such instantiation
static int tnd_resident(int nt, int kt, int nh) {
#define X(NT_, KT_, NH_)                                                                                                        \
    if (nt == NT_ && kt == KT_ && nh == NH_) {                                                                                \
        static const int n = [] {                                                                                             \
            int per_cu = 0, dev = 0, cus = 256;                                                                               \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_tn_direct_kernel<NT_, KT_, NH_>, 256,              \
                                                             tnd_lds(NT_, KT_, NH_)) != hipSuccess || per_cu < 1)             \
                per_cu = tnd_wpc(NT_, KT_);                                                                                   \
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; \
            return std::min(per_cu, 4) * (cus > 0 ? cus : 256);                                                               \
        }();                                                                                                                  \
        return n;                                                                                                             \
    }
    DN_TND_LIST
#undef X
    return 0;
}

static bool launch_tn_direct(const TnArgs &a, int splits, hipStream_t st) {
    const int nhalves = a.N > 96 ? 2 : 1;
    if (a.N % nhalves) return false;
    const int nt = tnd_cols(a.N / nhalves, a.lda), kt = tnd_cols(a.K, a.ldb);
    if (!nt || !kt || a.bidx || a.lda * 4 * 2 >= 0x7fffffffLL || a.ldb * 4 * 2 >= 0x7fffffffLL) return false;
    if (nhalves == 2 && nt == 2 && ((a.N / 2) & 1)) return false;
    if (nt == 2 && (reinterpret_cast<uintptr_t>(a.A) & 7)) return false;
    if (kt == 2 && (reinterpret_cast<uintptr_t>(a.B) & 7)) return false;
#define X(NT_, KT_, NH_)                                                                                                        \
    if (nt == NT_ && kt == KT_ && nhalves == NH_) {                                                                           \
        hipLaunchKernelGGL((gemm_tn_direct_kernel<NT_, KT_, NH_>), dim3(splits), dim3(256), tnd_lds(NT_, KT_, NH_), st, a);   \
        return true;                                                                                                          \
    }
    DN_TND_LIST
#undef X
    return false;
}

// out[r][c] (+)= sum_s part[s][r][c]   (deterministic second pass of the split reduction; out has row stride ldc)
// block = 32 outputs x 32 split lanes: coalesced 128-byte reads, fixed summation order
__global__ __launch_bounds__(1024) void reduce_splits_kernel(const float *part, int splits, int N, int K, float *out, int64_t ldc,
                                                             int accumulate) {
    __shared__ float red[32][33];
    const int lo = threadIdx.x & 31, lg = threadIdx.x >> 5;
    const int64_t n = (int64_t)N * K;
    const int64_t i = (int64_t)blockIdx.x * 32 + lo;
    float s = 0.f;
    if (i < n)
        for (int k = lg; k < splits; k += 32) s += part[k * n + i];
    red[lg][lo] = s;
    __syncthreads();
    if (lg == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][lo];
        float *dst = out + (i / K) * ldc + (i % K);
        *dst = accumulate ? *dst + t : t;
    }
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_nt_cfg(const NtArgs &a, bool veca, bool vecb, bool transb, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    const int64_t tiles = (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (tiles <= 0) return hipSuccess;
    // 8 XCDs x equal ranges of tiles (the kernel drops the padding); at most as many workgroups as are resident at once
    // (LDS-limited, <= 4 per CU), each walking several tiles.  Fewer, longer-lived workgroups are slower (measured 3 / 2 / 1
    // per CU: 233 / 245 / 284 ms per training step against 232): bytes in flight come from resident workgroups.
    const int64_t resident = 256 * std::min<int64_t>(4, (160 * 1024) / (int64_t)lds);
    const int64_t blocks = std::min((tiles + 7) / 8 * 8, resident);
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 g((unsigned)blocks), t(256);
#define DN_NT_CASE(VA, VB, TB)                                                                                        \
    if (veca == VA && vecb == VB && transb == TB) {                                                                   \
        hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, VA, VB, TB>), g, t, lds, st, a);                           \
        return hipGetLastError();                                                                                     \
    }
    DN_NT_CASE(true, true, false) DN_NT_CASE(true, true, true) DN_NT_CASE(true, false, false) DN_NT_CASE(true, false, true)
    DN_NT_CASE(false, true, false) DN_NT_CASE(false, true, true) DN_NT_CASE(false, false, false) DN_NT_CASE(false, false, true)
#undef DN_NT_CASE
    return hipErrorInvalidValue;
}

// split products for the wide layers: on unless DAGNET_NO_BF16X3 is set; dn_gemm_set_bf16x3 switches at run time (tests):
// 0 = fp32 MFMA, 1 = three bf16 terms (two fp16 terms where the caller vouches for the operands' range), 2 = two fp16 terms in
// every wide layer (tests and microbenchmarks of that form through dagnet_op_gemm)
static int g_bf16x3 = -1;
static bool bf16x3_on() {
    if (g_bf16x3 < 0) g_bf16x3 = getenv("DAGNET_NO_BF16X3") ? 0 : 1;
    return g_bf16x3 != 0;
}

#ifndef DN_H2_WPC
#define DN_H2_WPC 3
#endif
static hipError_t launch_nt_pipe(const NtArgs &a, hipStream_t st, bool fp16x2_ok) {
    constexpr int BM = 128, BN = 128;
    const int64_t tiles = (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static const bool no_h2 = getenv("DAGNET_WIDE_BF16X3") != nullptr;
    if (bf16x3_on() && ((fp16x2_ok && !no_h2) || g_bf16x3 == 2)) {
        static const int bm = getenv("DAGNET_H2_BM") ? atoi(getenv("DAGNET_H2_BM")) : 256;
        // 256 x 128 tiles (74 KB of LDS, two workgroups per CU) where the k loop is long: 1920 -> 960 298 against 281-291 TFLOP/s at
        // 122 880 rows; the short layers (K = 32, 64) are bound by writing C and lose with the larger tile (126 against 114 us)
        static const int bn = getenv("DAGNET_H2_BN") ? atoi(getenv("DAGNET_H2_BN")) : 256;
        if (bm == 256 && bn == 256 && a.M >= 4096 && a.K >= 256 && a.N >= 512) {       // 256 x 256, eight waves, 98 KB of LDS: one workgroup per CU
            constexpr size_t LDS = split_lds(2, 256, 256);
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_pipe_kernel<16, 1, 2, 256, 256>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
            if (attr != hipSuccess) return attr;
            const int64_t t = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
            const int64_t blocks = std::min<int64_t>((t + 7) / 8 * 8, 256);
            hipLaunchKernelGGL((gemm_nt_pipe_kernel<16, 1, 2, 256, 256>), dim3((unsigned)blocks), dim3(512), LDS, st, a);
            return hipGetLastError();
        }
        if (bm == 256 && a.M >= 4096 && a.K >= 256) {
            constexpr size_t LDS = split_lds(2, 256);
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_pipe_kernel<16, 2, 2, 256>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
            if (attr != hipSuccess) return attr;
            const int64_t t256 = (int64_t)((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
            const int64_t blocks = std::min<int64_t>((t256 + 7) / 8 * 8, 256 * 2);
            hipLaunchKernelGGL((gemm_nt_pipe_kernel<16, 2, 2, 256>), dim3((unsigned)blocks), dim3(256), LDS, st, a);
            return hipGetLastError();
        }
        constexpr size_t LDS = split_lds(2);          // 48 KB of LDS: three workgroups per CU
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_pipe_kernel<16, DN_H2_WPC, 2>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (attr != hipSuccess) return attr;
        const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, 256 * DN_H2_WPC);
        hipLaunchKernelGGL((gemm_nt_pipe_kernel<16, DN_H2_WPC, 2>), dim3((unsigned)blocks), dim3(256), LDS, st, a);
        return hipGetLastError();
    }
    // k-chunks of 16 with three workgroups per CU (41 KB of LDS each) measured 4 % faster on the 1920 -> 960 layer than chunks
    // of 32 with two: 127.5 against 122.3 TFLOP/s at 122 880 rows (tools/ubench/gemm_bench.py)
    if (bf16x3_on()) {          // 72 KB of LDS: two workgroups per CU
        constexpr size_t BF3_LDS = split_lds(3);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_pipe_kernel<16, 2, 3>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF3_LDS);
        if (attr != hipSuccess) return attr;
        const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, 256 * 2);
        hipLaunchKernelGGL((gemm_nt_pipe_kernel<16, 2, 3>), dim3((unsigned)blocks), dim3(256), BF3_LDS, st, a);
        return hipGetLastError();
    }
    const size_t lds = (size_t)2 * (BM + BN) * 20 * sizeof(float);
    const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, 256 * 3);
    hipLaunchKernelGGL((gemm_nt_pipe_kernel<16, 3>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace gemm

// General entry: C = act([C +] A op(B) + bias + gather), op(B) = W[N,K]^T (transb = 0) or B[K,N] (transb = 1).
// An operand that is 16-byte aligned with a row stride and a load-direction extent that are multiples of 4 takes float4
// staging loads, anything else scalar loads.  Tiles: 128x128 for wide outputs, 128x96 / 64x64 / 128x32 for the narrow layers,
// which are bound by streaming A and C rather than by the MFMAs (small tiles = more workgroups in flight per CU).
extern "C" hipError_t dn_gemm_skinny(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, float *C,
                                     int64_t ldc, int M, int N, int K, int act, float slope, int accumulate, const float *gather,
                                     const int32_t *gidx, int64_t gs_g, int64_t gs_n, hipStream_t st);
// fp16x2_ok: the caller vouches for operands of order one or below (see gemm_nt_pipe_kernel, SPLIT = 2)
static hipError_t gemm_impl(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, float *C,
                            int64_t ldc, int M, int N, int K, int act, float slope, int accumulate, const float *gather,
                            const int32_t *gidx, int64_t gs_g, int64_t gs_n, bool fp16x2_ok, hipStream_t st) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (act < gemm::ACT_NONE || act > gemm::ACT_RELU) return hipErrorInvalidValue;
    {   // tall-and-skinny shapes (the training step's layers): weights resident in LDS, A and C streamed by the waves themselves
        static const bool off = getenv("DAGNET_NO_SKINNY") != nullptr;
        if (!off) {
            const hipError_t e = dn_gemm_skinny(A, lda, B, ldb, transb, bias, C, ldc, M, N, K, act, slope, accumulate, gather, gidx, gs_g, gs_n, st);
            if (e != hipErrorNotSupported) return e;
        }
    }
    const bool veca = !((lda & 3) || (K & 3) || (reinterpret_cast<uintptr_t>(A) & 15));
    const bool vecb = !((ldb & 3) || ((transb ? N : K) & 3) || (reinterpret_cast<uintptr_t>(B) & 15));
    gemm::NtArgs a{A, lda, B, ldb, bias, C, ldc, M, N, K, slope, gather, gidx, gs_g, gs_n, act, accumulate};
    if (N > 128 && N <= 192) return gemm::launch_nt_cfg<2, 2, 1, 3>(a, veca, vecb, transb != 0, st);   // 64 x 192: the GRU gate columns in one tile, A read once
    if (N > 96) {                                                                             // 128 x 128
        static const bool no_pipe = getenv("DAGNET_NO_PIPE") != nullptr;
        if (!no_pipe && veca && vecb && !transb && K % 16 == 0 && lda * 4 * 128 < 0x7fffffffLL && ldb * 4 * 128 < 0x7fffffffLL &&
            ldc * 4 * 128 < 0x7fffffffLL)
            return gemm::launch_nt_pipe(a, st, fp16x2_ok);
        return gemm::launch_nt_cfg<2, 2, 2, 2>(a, veca, vecb, transb != 0, st);
    }
    // 128 x 96: g_dim = 90 in one column tile
    if (N > 64) return gemm::launch_nt_cfg<4, 1, 1, 3>(a, veca, vecb, transb != 0, st);
    if (N > 32) return gemm::launch_nt_cfg<2, 2, 1, 1>(a, veca, vecb, transb != 0, st);     // 64 x 64 (measured: 7 % of a training step over 128 x 64)
    return gemm::launch_nt_cfg<4, 1, 1, 1>(a, veca, vecb, transb != 0, st);                 // 128 x 32
}

extern "C" hipError_t dn_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, int transb, const float *bias, float *C,
                              int64_t ldc, int M, int N, int K, int act, float slope, int accumulate, const float *gather,
                              const int32_t *gidx, int64_t gs_g, int64_t gs_n, hipStream_t st) {
    return gemm_impl(A, lda, B, ldb, transb, bias, C, ldc, M, N, K, act, slope, accumulate, gather, gidx, gs_g, gs_n, false, st);
}

extern "C" int dn_gemm_set_bf16x3(int on) {
    const int before = gemm::bf16x3_on() ? 1 : 0;
    gemm::g_bf16x3 = on < 0 ? (getenv("DAGNET_NO_BF16X3") ? 0 : 1) : (on > 2 ? 1 : on);
    return before;
}

extern "C" hipError_t dn_gemm_nt_gather(const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, float *C,
                                        int64_t ldc, int M, int N, int K, int act, float slope, const float *gather,
                                        const int32_t *gidx, int64_t ldg, hipStream_t st) {
    // the goal head of the evaluation path (wide_head.hip) is this entry's only caller: ReLU / softmax / tanh outputs against
    // weights of order 1 / sqrt(K) - the wide layers may take two fp16 terms
    return gemm_impl(A, lda, W, ldw, 0, bias, C, ldc, M, N, K, act, slope, 0, gather, gidx, ldg, 1, true, st);
}

extern "C" hipError_t dn_gemm_nt(const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, float *C, int64_t ldc,
                                 int M, int N, int K, int act, float slope, hipStream_t st) {
    return dn_gemm(A, lda, W, ldw, 0, bias, C, ldc, M, N, K, act, slope, 0, nullptr, nullptr, 0, 0, st);
}

// C[N,K] (+)= A[M,N]^T B[M,K], deterministic two-pass split over M.  `part` is a workspace of
// dn_gemm_tn_workspace(M, N, K) floats.
// splits over the rows: enough workgroups for 256 CUs x 4 resident ones (the kernel is bound by load latency, not by its MFMAs;
// 2048 splits gained 10 % on the products and lost it again in the reduction of their partial sums), at least 128 rows each
extern "C" int64_t dn_gemm_tn_splits(int M, int N, int K) {
    const int64_t blocks = (int64_t)((N + 63) / 64) * ((K + 63) / 64);
    int64_t s = (1024 + blocks - 1) / blocks;
    {   // shapes of the direct kernel: one workgroup per split, as many as are resident at once
        static const bool off = getenv("DAGNET_NO_TN_DIRECT") != nullptr;
        const int nh = N > 96 ? 2 : 1, nt = (N / nh + 31) / 32, kt = (K + 31) / 32;
        const int res = off ? 0 : gemm::tnd_resident(nt, kt, nh);
        if (res > 0) s = res;
    }
    const int64_t cap = (M + 127) / 128;
    if (s > cap) s = cap;
    return s < 1 ? 1 : (s > 1024 ? 1024 : s);
}
extern "C" int64_t dn_gemm_tn_workspace(int M, int N, int K) { return dn_gemm_tn_splits(M, N, K) * ((int64_t)N * K + N); }
// dbias (nullable): [N] (+)= column sums of A, computed by the same pass over A (the bias gradient of the layer).
// stages: 1 = split partial sums only (part_accumulate: add to the partials of earlier calls with the same M, N, K),
//         2 = final reduction of the partials into C / dbias only, 3 = both.
extern "C" hipError_t dn_gemm_tn_staged(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, float *C,
                                        int64_t ldc, float *dbias, int want_bias, int M, int N, int K, float *part, int accumulate,
                                        int stages, int part_accumulate, hipStream_t st) {
    if (N <= 0 || K <= 0) return hipSuccess;
    const int splits = (int)dn_gemm_tn_splits(M, N, K);
    int rows = (M + splits - 1) / splits;
    rows = (rows + 31) & ~31;
    if (rows < 32) rows = 32;
    float *bias_part = want_bias ? part + (int64_t)splits * N * K : nullptr;
    if (stages & 1) {
        gemm::TnArgs a{A, lda, B, ldb, part, K, M, N, K, rows, bidx, bias_part, part_accumulate};
        const int blocks = ((N + 63) / 64) * ((K + 63) / 64);
        static const bool no_direct = getenv("DAGNET_NO_TN_DIRECT") != nullptr;
        if (no_direct || !gemm::launch_tn_direct(a, splits, st))
            hipLaunchKernelGGL(gemm::gemm_tn_kernel, dim3(blocks, splits), dim3(256), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (stages & 2) {
        const int64_t n = (int64_t)N * K;
        hipLaunchKernelGGL(gemm::reduce_splits_kernel, dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, st, part, splits, N, K, C, ldc,
                           accumulate);
        if (want_bias && dbias)
            hipLaunchKernelGGL(gemm::reduce_splits_kernel, dim3((unsigned)((N + 31) / 32)), dim3(1024), 0, st, bias_part, splits, 1, N,
                               dbias, (int64_t)N, accumulate);
    }
    return hipGetLastError();
}

extern "C" hipError_t dn_gemm_tn_idx(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *bidx, float *C, int64_t ldc,
                                     float *dbias, int M, int N, int K, float *part, int accumulate, hipStream_t st) {
    return dn_gemm_tn_staged(A, lda, B, ldb, bidx, C, ldc, dbias, dbias != nullptr, M, N, K, part, accumulate, 3, 0, st);
}

extern "C" hipError_t dn_gemm_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, int M, int N, int K,
                                 float *part, int accumulate, hipStream_t st) {
    return dn_gemm_tn_idx(A, lda, B, ldb, nullptr, C, ldc, nullptr, M, N, K, part, accumulate, st);
}
