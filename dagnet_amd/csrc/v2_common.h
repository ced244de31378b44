// Building blocks shared by the register-resident kernels (rollout_v2.hip: sample, forward_v2.hip: teacher-forced pass):
// weight streaming through one buffer descriptor, chained MFMA layers, per-lane softmax / attention, GAT, GRU.
// See the header comment of rollout_v2.hip for the layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "params_v2.h"
#include "rng.h"

typedef float T16 __attribute__((ext_vector_type(16)));

#define FR(r) (((r) & 3) + 8 * ((r) >> 2))
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// raw buffer loads declared straight on the LLVM intrinsics: clang 22's
// __builtin_amdgcn_raw_buffer_load_b128 collapses to a single dword here (all four lanes of the
// result alias element 0), these forms select buffer_load_dwordx4 / _dword as expected.
__device__ f32x4 dn_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float dn_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

#ifndef DN_V2_DEPTH
#define DN_V2_DEPTH 1     // weight blocks in flight ahead of the MFMAs (1 block = 16 MFMAs = 1024 cycles of cover)
#endif

namespace v2 {

// All weight traffic goes through ONE buffer descriptor (4 SGPRs): per-lane part of the address is a
// single VGPR (voff, bytes), the per-use part a scalar offset.  With plain pointers the compiler
// hoists ~150 loop-invariant 64-bit per-lane addresses out of the time loop and spills them.
struct WBuf {
    i32x4 rs;
    __device__ __forceinline__ void init(const float *p) {
        union { const float *p; int w[2]; } u;
        u.p = p;
        rs = i32x4{u.w[0], u.w[1], 0x7fffffff, 0x00020000};
    }
    __device__ __forceinline__ float4 ld4(int voff_bytes, int soff_floats) const {
        asm volatile("" : "+s"(soff_floats));     // keep scalar offsets out of loop-invariant hoisting (SGPR spills)
        const f32x4 v = dn_buffer_load_f32x4(rs, voff_bytes, soff_floats * 4, 0);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    // the scalar offset in BYTES, already opaque (wload_half: a compile-time segment offset then costs one s_mov, not s_mov + s_lshl)
    __device__ __forceinline__ float4 ld4b(int voff_bytes, int soff_bytes) const {
        const f32x4 v = dn_buffer_load_f32x4(rs, voff_bytes, soff_bytes, 0);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ float ld1(int voff_bytes, int soff_floats) const {
        return dn_buffer_load_f32(rs, voff_bytes, soff_floats * 4, 0);
    }
};

// Hardware transcendental units (v_exp_f32 / v_rcp_f32 / v_log_f32, 1 ulp) instead of the ~10-instruction
// libm expansions: the per-lane epilogues are pure VALU time next to the MFMA stream.  Relative error
// <= ~2e-7 * (1 + |x|), two orders below the 1e-4 parity tolerance (checked by tests/test_gpu_parity.py).
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
// lrelu: 0 < s < 1: max(v, s v)
__device__ __forceinline__ float lrelu(float v, float s) { return __builtin_amdgcn_fmed3f(v, v * s, INFINITY); }
// sigmoid / tanh of pre-activations that arrive ALREADY scaled (api.hip folds the factors into the packed weights, biases
// and BatchNorm tables): sigm2(v) = sigmoid(x) for v = -log2(e) x, tanh2(v) = tanh(x) for v = 2 log2(e) x.  No clamp:
// exp2 overflows to +inf, whose reciprocal is 0, and underflows to 0 - both limits come out exactly.
__device__ __forceinline__ float sigm2(float v) { return frcp(1.0f + __builtin_amdgcn_exp2f(v)); }
__device__ __forceinline__ float tanh2(float v) { return 1.0f - 2.0f * frcp(1.0f + __builtin_amdgcn_exp2f(v)); }
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }
__device__ __forceinline__ int xhalf(int v) { return __shfl_xor(v, 32); }
// max / sum over the two lane halves, the result in both: one v_permlane32_swap (gfx950, a VALU instruction) instead of a
// ds_bpermute round trip through LDS in the middle of a dependent chain (softmax: max, then sum)
__device__ __forceinline__ float hmax(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // r[0] = lower half's v, r[1] = upper half's, in every lane
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float hsum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int N>
__device__ __forceinline__ void zero(T16 (&x)[N]) {
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = 0.f;
}

// Weights in flight.  fp32 (BF = false): one (n-tile, k-tile) block = 4 x float4 per lane = the A (or B) operands of 16
// v_mfma_f32_32x32x2_f32, in v[0..3].  Split fp16 (BF = true, params_v2.h): the unit is a HALF block - one output tile x one
// k-half (16 input features) x two fp16 terms = 2 x 16 bytes per lane, the weight operands of 3 v_mfma_f32_32x32x16_f16 -
// and V2_BF_DEPTH of them are in flight: v[2*d + term] = half block d of the consumer's stream.
#ifndef V2_BF_DEPTH
#define V2_BF_DEPTH 3
#endif
#ifdef DN_V2_NOFENCE      // experiment: leave the order of splits, loads and MFMAs of the split products to the compiler
#define V2_FENCE()
#else
#define V2_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
struct WBlk {
    float4 v[2 * V2_BF_DEPTH > 4 ? 2 * V2_BF_DEPTH : 4];
};
// the two quads of one half block into slot d
__device__ __forceinline__ void wload_half(WBlk &b, int d, const WBuf &wp, int off, int lane) {
    int ob = off * 4;
    asm volatile("" : "+s"(ob));      // keep the scalar offset out of loop-invariant hoisting (SGPR spills)
#pragma unroll
    for (int j = 0; j < 2; ++j) b.v[2 * d + j] = wp.ld4b(lane * 16 + j * 1024, ob);
}
template <bool BF = false>
__device__ __forceinline__ WBlk wload(const WBuf &wp, int off, int lane) {
    WBlk b;
    if constexpr (BF) {
#pragma unroll
        for (int d = 0; d < V2_BF_DEPTH; ++d) wload_half(b, d, wp, off + d * V2_HALF_SPLIT, lane);
    } else {
        asm volatile("" : "+s"(off));
#pragma unroll
        for (int j = 0; j < 4; ++j) b.v[j] = wp.ld4(lane * 16 + j * 1024, off);
    }
    return b;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f16x8 as_h8(float4 v) { return __builtin_bit_cast(f16x8, v); }

// Registers 8p .. 8p+7 of an S- or T-layout tile as two fp16 terms, rounded to nearest: hi = fp16(v), lo = fp16(v - hi) (the
// difference is exact in fp32) - three instructions per pair of values: v_cvt_pk_f16_f32 for the two hi terms, then v_fma_mixlo_f16 /
// v_fma_mixhi_f16, which read one half of the packed hi as fp16, compute hi * (-1) + v in fp32 and write the rounded fp16 into one
// half of the packed lo.  The compiler selects them for fptrunc(fma(fpext(hi), m, v)) only if (a) the multiplier is not a
// constant it can fold (v - hi as an fsub becomes two v_cvt_f32_f16 + a v_pk_add_f32 + a v_cvt_pk_f16_f32: five instructions per
// pair, DN_V2_SPLIT_SUB) - hence the opaque -1.0 in an SGPR - and (b) the SLP vectoriser does not pair the two fmas into one
// v_pk_fma_f32 - hence the opaque copy of the low half between them.  1 215 fewer VALU instructions per wave-step of the DAG-Net
// kernel (10 024 -> 8 774), 42.1 -> 40.1 ms per launch.
// (The same two instructions written as INLINE ASM were tried first and are gone: inline asm is invisible to the compiler's hazard
// recognizer, which pads a VALU write that an MFMA reads next with s_nop - 531 of them per step in the compiler-visible form, 367
// with the asm - and ONE instantiation - DAG-Net, one GRU layer, GCN - produced trajectories 1e-2 off from the second step on.)
// t[term] = the 8 k-slots of one v_mfma_f32_32x32x16_f16 operand (slot s = register 8p + s: the order the weights are packed in).
// The values are read through an opaque copy: left alone, the compiler recognises the split of a tile that several products
// read (h, phi_z, the GRU's inputs) as a common subexpression and keeps its registers alive from the first use to the last -
// the register file does not have them (the first version spilled ~100 registers per lane).
struct XSplit {
    float4 t[2];
};
// -1.0 as a value the compiler cannot fold: a weak constant-memory variable (it may not assume a weak symbol's initialiser),
// read once per kernel into an SGPR.  (An opaque copy of the literal per call, `asm volatile("" : "+s"(m))`, cost an s_mov and
// a hazard s_nop behind every inline asm: ~150 of each per wave-step.)
__attribute__((weak)) __constant__ float v2_minus_one = -1.0f;
__device__ __forceinline__ float split_m1() { return v2_minus_one; }
__device__ __forceinline__ void split2(const T16 &x, int p, XSplit &s) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned pk[2][4];
#ifndef DN_V2_SPLIT_SUB
    const float m1 = split_m1();
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v0 = x[8 * p + 2 * j], v1 = x[8 * p + 2 * j + 1];
#ifndef DN_V2_SPLIT_NOCOPY
        asm volatile("" : "+v"(v0), "+v"(v1));
#endif
        const f2 v = {v0, v1};
        const h2 hi = __builtin_convertvector(v, h2);
        pk[0][j] = __builtin_bit_cast(unsigned, hi);
#ifndef DN_V2_SPLIT_SUB
        h2 lo;
        _Float16 l0 = (_Float16)__builtin_fmaf((float)hi[0], m1, v0);
#ifndef DN_V2_NO_SLP
        // (built with the SLP vectoriser: keep it from pairing the two fmas.  The Makefile builds this object without it and
        // defines DN_V2_NO_SLP - the opaque copy reads a register that v_fma_mixlo_f16 has just written in part, which costs
        // a wait state each: 300 s_nop per wave-step)
        asm("" : "+v"(l0));
#endif
        lo[0] = l0;
        lo[1] = (_Float16)__builtin_fmaf((float)hi[1], m1, v1);
#else
        const f2 r = v - __builtin_convertvector(hi, f2);
        const h2 lo = __builtin_convertvector(r, h2);
#endif
        pk[1][j] = __builtin_bit_cast(unsigned, lo);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
        s.t[t] = make_float4(__uint_as_float(pk[t][0]), __uint_as_float(pk[t][1]), __uint_as_float(pk[t][2]), __uint_as_float(pk[t][3]));
}

// out[nt] += W(nt,kt) . x[kt]     TR = false: S <- S (weights are the A operand)
//                                 TR = true : T <- S (weights are the B operand: out = x^T W^T)
// Weight blocks are software pipelined across the whole step: `pre` holds this call's first block(s) (already in flight,
// issued by the previous call), `next_off` is the start of whatever consumes weights next; the return value is its first
// block(s), in flight.
// BF ("split"): the products run on the 16-bit matrix pipe - x and W as two fp16 terms each, three products (params_v2.h), fp32
// accumulation: 6 MFMAs of 32 clocks per 32 x 32 block where the fp32 form takes 16 of 64.  The stream is ordered [kt][k-half p][nt]:
// half an input tile is split once (8 registers) and feeds all output tiles, then the other half.
template <int NT, int KT, bool TR, bool BF = false>
__device__ __forceinline__ WBlk mm(T16 (&out)[NT], const WBuf &wp, int off, const T16 (&x)[KT], int lane, WBlk pre,
                                   int next_off) {
    constexpr int NB = NT * KT;
    if constexpr (BF) {
        constexpr int NH = 2 * NB, D = V2_BF_DEPTH;      // half blocks of this call (>= 2)
        WBlk q = pre;
        XSplit xs;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int kt = i / (2 * NT), p = (i / NT) % 2, nt = i % NT, d = i % D;
            if (nt == 0) split2(x[kt], p, xs);
            V2_FENCE();
            const float4 wh = q.v[2 * d], wl = q.v[2 * d + 1];
            // the two small products first
            if (!TR) {
                out[nt] = MFMA_H(as_h8(wl), as_h8(xs.t[0]), out[nt]);
                out[nt] = MFMA_H(as_h8(wh), as_h8(xs.t[1]), out[nt]);
                out[nt] = MFMA_H(as_h8(wh), as_h8(xs.t[0]), out[nt]);
            } else {
                out[nt] = MFMA_H(as_h8(xs.t[0]), as_h8(wl), out[nt]);
                out[nt] = MFMA_H(as_h8(xs.t[1]), as_h8(wh), out[nt]);
                out[nt] = MFMA_H(as_h8(xs.t[0]), as_h8(wh), out[nt]);
            }
            V2_FENCE();
            // the slot just consumed takes half block i + D of this call, or the next consumer's
            wload_half(q, d, wp, i + D < NH ? off + (i + D) * V2_HALF_SPLIT : next_off + (i + D - NH) * V2_HALF_SPLIT, lane);
        }
        // slot d now holds the next consumer's half block (NH + d' ...) - rotate so that its half block 0 sits in slot 0
        if constexpr (NH % D != 0) {
            WBlk r;
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int j = 0; j < 2; ++j) r.v[2 * d + j] = q.v[2 * ((d + NH) % D) + j];
            return r;
        }
        return q;
    } else {
#if DN_V2_DEPTH == 2
    WBlk q0 = pre, q1, q2;
    q1 = wload(wp, NB > 1 ? off + 1024 : next_off, lane);
#else
    WBlk q0 = pre, q1;
#endif
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#if DN_V2_DEPTH == 2
        if (b + 2 <= NB) q2 = wload(wp, b + 2 < NB ? off + (b + 2) * 1024 : next_off, lane);
#else
        q1 = wload(wp, b + 1 < NB ? off + (b + 1) * 1024 : next_off, lane);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const int nt = b / KT, kt = b % KT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a = q0.v[j];
            if (!TR) {
                out[nt] = MFMA(a.x, x[kt][4 * j + 0], out[nt]);
                out[nt] = MFMA(a.y, x[kt][4 * j + 1], out[nt]);
                out[nt] = MFMA(a.z, x[kt][4 * j + 2], out[nt]);
                out[nt] = MFMA(a.w, x[kt][4 * j + 3], out[nt]);
            } else {
                out[nt] = MFMA(x[kt][4 * j + 0], a.x, out[nt]);
                out[nt] = MFMA(x[kt][4 * j + 1], a.y, out[nt]);
                out[nt] = MFMA(x[kt][4 * j + 2], a.z, out[nt]);
                out[nt] = MFMA(x[kt][4 * j + 3], a.w, out[nt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        q0 = q1;
#if DN_V2_DEPTH == 2
        q1 = q2;
#endif
    }
    return q0;
    }
}

// acc += A . B for two register tiles on the 16-bit matrix pipe: A (already split: a[k-half]) supplies lane = output feature,
// slots = the 32 contraction rows; b (split here) lane = output column.  The graph layers' per-scene aggregation out^T = Wh^T alpha^T.
__device__ __forceinline__ void split_tile(const T16 &x, XSplit (&s)[2]) {
    split2(x, 0, s[0]);
    split2(x, 1, s[1]);
}
__device__ __forceinline__ void agg_split(T16 &acc, const XSplit (&a)[2], const T16 &b) {
    XSplit bs[2];
    split_tile(b, bs);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        acc = MFMA_H(as_h8(a[kh].t[1]), as_h8(bs[kh].t[0]), acc);
        acc = MFMA_H(as_h8(a[kh].t[0]), as_h8(bs[kh].t[1]), acc);
        acc = MFMA_H(as_h8(a[kh].t[0]), as_h8(bs[kh].t[0]), acc);
    }
}

// x[nt][r] += v[nt*32 + feature(r, h)]   for a vector stored [NT*32]
template <int NT>
__device__ __forceinline__ void add_vec(T16 (&x)[NT], const WBuf &wp, int off, int voff, int lh) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 b = wp.ld4(voff + lh * 16, off + nt * 32 + 8 * a);
            x[nt][4 * a + 0] += b.x;
            x[nt][4 * a + 1] += b.y;
            x[nt][4 * a + 2] += b.z;
            x[nt][4 * a + 3] += b.w;
        }
}

// ACT: 0 none, 1 LeakyReLU(0.01), 2 ReLU
template <int ACT, int NT>
__device__ __forceinline__ void activate(T16 (&x)[NT]) {
#ifdef DN_V2_NO_PKLRELU
    if (ACT == 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[nt][r] = lrelu(x[nt][r], 0.01f);
        return;
    }
#endif
    if (ACT == 1) {         // LeakyReLU(0.01) = max(v, 0.01 v): the products as v_pk_mul_f32 (two per instruction), then one v_max each
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2 v = {x[nt][r], x[nt][r + 1]};
                const f2 m = v * f2{0.01f, 0.01f};
                x[nt][r] = __builtin_amdgcn_fmed3f(v.x, m.x, INFINITY);
                x[nt][r + 1] = __builtin_amdgcn_fmed3f(v.y, m.y, INFINITY);
            }
    } else if (ACT != 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[nt][r] = fmaxf(x[nt][r], 0.f);
    }
}
// The same vector as the INITIAL value of the layer's accumulators instead of an addition after its MFMAs: the loads are
// issued before the layer's first MFMA block (the scheduler lifts them up to the end of the previous layer's MFMAs, so the
// round trip to L2 overlaps that layer's activation), and the NT*16 additions per lane disappear.
template <int NT>
__device__ __forceinline__ void start(T16 (&x)[NT], const WBuf &wp, const V2Lin &L, int gsel, int lh) {
    const bool g = L.gather >= 0;
    const int off = g ? L.gather : L.bias, voff = (g ? gsel * (NT * 32 * 4) : 0) + lh * 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 b = wp.ld4(voff, off + nt * 32 + 8 * a);
            x[nt][4 * a + 0] = b.x;
            x[nt][4 * a + 1] = b.y;
            x[nt][4 * a + 2] = b.z;
            x[nt][4 * a + 3] = b.w;
        }
}

// Bias of a dense layer, or - for a layer with a one-hot input - row gsel of its table (api.hip v2_linear: the
// bias is folded into every row of the table), so either way ONE set of NT*4 independent loads per layer.
template <int ACT, int NT>
__device__ __forceinline__ void finish(T16 (&x)[NT], const WBuf &wp, const V2Lin &L, int gsel, int lh) {
    const bool g = L.gather >= 0;
    add_vec<NT>(x, wp, g ? L.gather : L.bias, g ? gsel * (NT * 32 * 4) : 0, lh);
    activate<ACT, NT>(x);
}

// softmax over the features of an S-layout row whose logits arrive in the log2 domain with -inf in the padded features
// (api.hip v2_linear, logits = true): no mask, exp2(x - max) with no multiply; padded features come out as exactly 0
template <int NT>
__device__ __forceinline__ void softmax_feat(T16 (&x)[NT]) {
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, x[nt][r]);
    mx = hmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[nt][r] = __builtin_amdgcn_exp2f(x[nt][r] - mx);
            sum += x[nt][r];
        }
    sum = hsum(sum);
    const float inv = frcp(sum);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[nt][r] = x[nt][r] * inv;
}

// the same, left unnormalised: x = exp2(logit - max); returns 1 / sum (the goal draw folds it into its score scale)
template <int NT>
__device__ __forceinline__ float softmax_feat_unnorm(T16 (&x)[NT]) {
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, x[nt][r]);
    mx = hmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[nt][r] = __builtin_amdgcn_exp2f(x[nt][r] - mx);
            sum += x[nt][r];
        }
    return frcp(hsum(sum));
}

// attention weights of this lane's row over the 16 tile rows its half owns:
// alpha[r] = softmax_j( mask ? LeakyReLU(s1 + s2[j]) : -inf ),  j = FR(r) + 4*lh
// s1, s2 arrive scaled by log2(e) (api.hip v2_gat folds the factor into the logit weights; LeakyReLU commutes with a
// positive factor), so the softmax is exp2(e - max) with no multiply; mb[r] = 0 / -inf is the mask as an addend, built once
// per graph call instead of a bit test per head and row.  (The reference masks with -9e15: exp() of either is exactly 0.)
__device__ __forceinline__ void attn_mask_bias(float (&mb)[16], unsigned amask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mb[r] = ((amask >> r) & 1u) ? 0.f : -INFINITY;
}
// NORM = false: alpha is left unnormalised (exp2(e - max)) and 1 / sum is returned, for the caller to apply to the
// aggregated result - a factor per row commutes with the aggregation, and the MFMAs need not wait for sum -> rcp -> multiply
template <bool NORM = true>
__device__ __forceinline__ float attn_weights(T16 &alpha, float s1, const float (&s2)[16], const float (&mb)[16], float slope) {
    float e[16];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        e[r] = lrelu(s1 + s2[r], slope) + mb[r];
        mx = fmaxf(mx, e[r]);
    }
    mx = hmax(mx);          // finite: a row is always its own neighbour, in one of the two halves
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        e[r] = __builtin_amdgcn_exp2f(e[r] - mx);
        sum += e[r];
    }
    sum = hsum(sum);
    const float inv = frcp(sum);
#pragma unroll
    for (int r = 0; r < 16; ++r) alpha[r] = NORM ? e[r] * inv : e[r];
    return inv;
}

// Sum over the 8 consecutive lanes of a head (lanes 8p .. 8p+7), the result in all 8: three DPP adds, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    // no contraction: with it, v = a * b turns the add into v_fmac(dpp(v), a, b) behind a separate v_mov_b32_dpp - three
    // instructions (v_mul, v_mov_dpp, v_fmac) where v_mul + v_add_f32_dpp do
#pragma clang fp contract(off)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float red8(float v) {
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    return dpp_add<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its group of 8, i.e. the other quad's sum
}

// Attention vectors of one GAT for the per-lane logit arithmetic of gat<> (api.hip v2_gat: fold | fold_out, 128 floats):
// [a_src 32 | a_dst 32 | w1 32 | w2 32], copied to LDS once per wave
constexpr int GAT_FOLD_FLOATS = 128;
__device__ __forceinline__ void gat_fold_to_lds(float *dst, const WBuf &wp, const V2Gat &Gw, int lane) {
    dst[lane] = wp.ld1(lane * 4, Gw.fold);
    dst[64 + lane] = wp.ld1(lane * 4, Gw.fold_out);
}

// Dense GAT (P heads of F features with P*F = 32 -> out_att -> BatchNorm(eval) -> tanh) on the tile.
// x: KT input tiles (S).  out: NT tiles (S).  xch: per-wave LDS scratch of >= 32*4 floats.  gf: gat_fold_to_lds() image.
// The attention logits s1[p] = a_src[p].Wh_row, s2[p] = a_dst[p].Wh_row are VALU work on the transposed head product
// (lane = head feature: multiply by the lane's entry of a, sum the 8 lanes of the head with DPP), and the out head's
// t = (x2 W_out).a_out = x2.(W_out a_out) a 32-term dot product per lane - not 32-wide MFMA tiles with 8 (resp. 2) live
// output features: 5 + 2 blocks of 16 MFMAs per step of the DAG-Net kernel for ~400 VALU instructions (fp32 MFMAs and
// VALU share the SIMD's FMA lanes, DESIGN.md 5, so an MFMA tile that is 3/4 padding is pure loss).
struct NoProf {
    __device__ __forceinline__ void operator()(int) const {}
};
// ph: stage timer of the diagnostic build (rollout_v2.hip PH(), tools/phase_profile.py); a no-op otherwise
template <int KT, int NT, bool BF = false, class PF = NoProf>
__device__ __forceinline__ WBlk gat(const T16 (&x)[KT], T16 (&out)[NT], const WBuf &wp, const V2Gat &Gw, unsigned amask,
                                    float slope, float *xch, const float *gf, int lane, WBlk pre, int next_off, PF ph = PF()) {
    const int lh = lane >> 5, lc = lane & 31;
    // the mask addends are rebuilt per call (16 selects): with the same mask in both graphs of a step the compiler otherwise
    // keeps the 16 values of the goal graph's call alive - in scratch memory - until the hidden graph's
    asm volatile("" : "+v"(amask));
    // Wh for all heads, transposed: lane = head feature, register = tile row
    T16 whT[1];
    zero(whT);
    const float a_src = gf[lc], a_dst = gf[32 + lc];
    constexpr int BS = BF ? V2_BLOCK_SPLIT : V2_BLOCK_F32;
    pre = mm<1, KT, true, BF>(whT, wp, Gw.heads, x, lane, pre, Gw.out);
    ph(10);
    // stage by stage over all 32 values: a DPP read needs two wait states behind the VALU write of its source, which the next
    // values' instructions fill (value by value the compiler padded every step of every chain with an s_nop)
    float u1[16], u2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u1[r] = whT[0][r] * a_src;
        u2[r] = whT[0][r] * a_dst;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { u1[r] = dpp_add<0xB1>(u1[r]); u2[r] = dpp_add<0xB1>(u2[r]); }      // quad_perm [1,0,3,2]
#pragma unroll
    for (int r = 0; r < 16; ++r) { u1[r] = dpp_add<0x4E>(u1[r]); u2[r] = dpp_add<0x4E>(u2[r]); }      // quad_perm [2,3,0,1]
#pragma unroll
    for (int r = 0; r < 16; ++r) { u1[r] = dpp_add<0x141>(u1[r]); u2[r] = dpp_add<0x141>(u2[r]); }    // row_half_mirror
    // register r = 4a + i of half lh is tile row i + 8a + 4 lh: the first lane of every head writes its 16 rows of s1, every
    // lane picks up its own row's four values, then the same 128 floats carry s2 (read by everybody in the head loop)
    __syncthreads();
    if ((lc & 7) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4 *>(xch + (lc >> 3) * 32 + 8 * a + 4 * lh) = make_float4(u1[4 * a], u1[4 * a + 1], u1[4 * a + 2], u1[4 * a + 3]);
    }
    __syncthreads();
    float s1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) s1[p] = xch[p * 32 + lc];
    __syncthreads();
    if ((lc & 7) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4 *>(xch + (lc >> 3) * 32 + 8 * a + 4 * lh) = make_float4(u2[4 * a], u2[4 * a + 1], u2[4 * a + 2], u2[4 * a + 3]);
    }
    __syncthreads();
    float mb[16];
    attn_mask_bias(mb, amask);
    ph(11);
    T16 x2[1];
    zero(x2);
    float inv[4];
    XSplit whs[2];
    if constexpr (BF) split_tile(whT[0], whs);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float s2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s2[r] = xch[p * 32 + FR(r) + 4 * lh];
        T16 alpha;
        inv[p] = attn_weights<false>(alpha, s1[p], s2, mb, slope);
        if constexpr (BF) {
            // the product of ALL 32 head features with head p's attention weights; its registers 4p .. 4p+3 are head p's features
            // 8p .. 8p+7 of this lane's row (the other twelve belong to other heads' attention and are dropped) - six 32-clock
            // MFMAs per head where the masked fp32 form took sixteen of 64
            T16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            agg_split(dp, whs, alpha);
#pragma unroll
            for (int i = 0; i < 4; ++i) x2[0][4 * p + i] = dp[4 * p + i];
        } else {
            const bool mine = (lc >> 3) == p;     // this lane's feature belongs to head p
#pragma unroll
            for (int r = 0; r < 16; ++r) x2[0] = MFMA(mine ? whT[0][r] : 0.f, alpha[r], x2[0]);
        }
    }
    ph(12);
    // head p's features 8p .. 8p+7 are registers 4p .. 4p+3 of this lane's row: its softmax denominator is applied here
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = x2[0][r] * inv[r >> 2];
        x2[0][r] = v > 0.f ? v : fexp(v) - 1.0f;     // ELU
    }
    // out_att logits: this half's 16 features of the row, then the other half's
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 w1 = *reinterpret_cast<const float4 *>(gf + 64 + 8 * a + 4 * lh);
        const float4 w2 = *reinterpret_cast<const float4 *>(gf + 96 + 8 * a + 4 * lh);
        t1 += x2[0][4 * a] * w1.x + x2[0][4 * a + 1] * w1.y + x2[0][4 * a + 2] * w1.z + x2[0][4 * a + 3] * w1.w;
        t2 += x2[0][4 * a] * w2.x + x2[0][4 * a + 1] * w2.y + x2[0][4 * a + 2] * w2.z + x2[0][4 * a + 3] * w2.w;
    }
    t1 = hsum(t1);
    t2 = hsum(t2);
    __syncthreads();
    if (lh == 0) xch[lc] = t2;
    __syncthreads();
    float s2o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s2o[r] = xch[FR(r) + 4 * lh];
    T16 alpha;
    const float invo = attn_weights<false>(alpha, t1, s2o, mb, slope);
    ph(13);
    // eval BatchNorm folded away (api.hip v2_gat): its scale (and tanh2's factor) sits in the out_att weights - a factor per
    // output feature commutes with the aggregation over rows - and its shift is added with the softmax denominator
    zero(out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {      // one output tile at a time: Wh2^T tile -> aggregation (keeps 16, not NT*16, registers live)
        T16 wh2T[1];
        zero(wh2T);
        pre = mm<1, 1, true, BF>(wh2T, wp, Gw.out + nt * BS, x2, lane, pre, nt + 1 < NT ? Gw.out + (nt + 1) * BS : next_off);
        if constexpr (BF) {
            XSplit w2s[2];
            split_tile(wh2T[0], w2s);
            agg_split(out[nt], w2s, alpha);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nt] = MFMA(wh2T[0][r], alpha[r], out[nt]);
        }
    }
    ph(14);
    // tanh; padded features (weights and shift 0) stay 0
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 sh = wp.ld4(lh * 16, Gw.bn + nt * 32 + 8 * a);
            const float b4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) out[nt][4 * a + i] = tanh2(out[nt][4 * a + i] * invo + b4[i]);
        }
    ph(15);
    return pre;
}

// Dense GCN (gcn_model.py:14-19: gc1 -> BatchNorm -> gc2 -> BatchNorm -> tanh, gc(x) = adj @ (x W) + b, gcn_layers.py:20-26)
// on the tile, with the same transposed-product trick as gat<>: (x W)^T is the A operand and this lane's row of the
// adjacency (adjv(r) = adj[row][tile row FR(r) + 4*lh], 0 outside the scene) the B operand of the aggregation.
// V2Gat fields as used here: heads = W1 [1 tile (F <= 32)][KT], fold = [2][32] scale | shift of layer 1 (bias and eval
// BatchNorm folded, zero beyond F), out = W2 [NT][1], bn = [2][NT*32] scale | shift of layer 2.
template <int KT, int NT, bool BF = false, class AdjFn = void>
__device__ __forceinline__ WBlk gcn(const T16 (&x)[KT], T16 (&out)[NT], const WBuf &wp, const V2Gat &Gw, AdjFn adjv, int lane,
                                    WBlk pre, int next_off) {
    const int lh = lane >> 5;
    constexpr int BS = BF ? V2_BLOCK_SPLIT : V2_BLOCK_F32;
    T16 whT[1];
    zero(whT);
    pre = mm<1, KT, true, BF>(whT, wp, Gw.heads, x, lane, pre, Gw.out);
    float av[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) av[r] = adjv(r);
    T16 x1[1], avt;
    zero(x1);
#pragma unroll
    for (int r = 0; r < 16; ++r) avt[r] = av[r];
    if constexpr (BF) {
        XSplit whs[2];
        split_tile(whT[0], whs);
        agg_split(x1[0], whs, avt);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) x1[0] = MFMA(whT[0][r], av[r], x1[0]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 sc = wp.ld4(lh * 16, Gw.fold + 8 * a), sh = wp.ld4(lh * 16, Gw.fold + 32 + 8 * a);
        x1[0][4 * a + 0] = x1[0][4 * a + 0] * sc.x + sh.x; x1[0][4 * a + 1] = x1[0][4 * a + 1] * sc.y + sh.y;
        x1[0][4 * a + 2] = x1[0][4 * a + 2] * sc.z + sh.z; x1[0][4 * a + 3] = x1[0][4 * a + 3] * sc.w + sh.w;
    }
    zero(out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        T16 wh2T[1];
        zero(wh2T);
        pre = mm<1, 1, true, BF>(wh2T, wp, Gw.out + nt * BS, x1, lane, pre, nt + 1 < NT ? Gw.out + (nt + 1) * BS : next_off);
        if constexpr (BF) {
            XSplit w2s[2];
            split_tile(wh2T[0], w2s);
            agg_split(out[nt], w2s, avt);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nt] = MFMA(wh2T[0][r], av[r], out[nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int fs = nt * 32 + 8 * a;
            const float4 sc = wp.ld4(lh * 16, Gw.bn + fs), sh = wp.ld4(lh * 16, Gw.bn + NT * 32 + fs);
            const float g4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) out[nt][4 * a + i] = tanh2(out[nt][4 * a + i] * g4[i] + b4[i]);
        }
    return pre;
}

// Gate biases of one GRU layer for the register-resident kernels: [nh | r | ni | z][64] floats.  The rollout kernel keeps
// them in LDS (copied once per wave by gru_bias_to_lds): a gate's bias is needed right after its last MFMA, where a load
// from L2 is a fully exposed round trip (16 of them per step with two layers), an LDS read a tenth of that.
constexpr int GRU_BIAS_FLOATS = 4 * 64;
__device__ __forceinline__ void gru_bias_to_lds(float *dst, const WBuf &wp, const V2Gru &Gw, int lane) {
    dst[0 * 64 + lane] = wp.ld1(lane * 4, Gw.nh.bias);
    dst[1 * 64 + lane] = wp.ld1(lane * 4, Gw.r.bias);
    dst[2 * 64 + lane] = wp.ld1(lane * 4, Gw.ni.bias);
    dst[3 * 64 + lane] = wp.ld1(lane * 4, Gw.z.bias);
}

// One GRU layer.  xin: KX input tiles; h: 2 tiles (R = 64), updated in place.
// LB: the layer's biases are in LDS at lb (gru_bias_to_lds); else they are loaded from the weight buffer.
template <int KX, bool LB = false, bool BF = false>
__device__ __forceinline__ WBlk gru(const T16 (&xin)[KX], T16 (&h)[2], const WBuf &wp, const V2Gru &Gw, int lane, WBlk pre,
                                    int next_off, const float *lb = nullptr, float *park = nullptr) {
    const int lh = lane >> 5;
    auto bias4 = [&](int gate, int goff, int f) -> float4 {      // 4 consecutive features f + 4*lh .. of gate `gate`
        if constexpr (LB) return *reinterpret_cast<const float4 *>(lb + gate * 64 + f + 4 * lh);
        else return wp.ld4(lh * 16, goff + f);
    };
    // a gate's bias is the initial value of its accumulator (read from LDS / L2 while the previous gate's MFMAs run), not
    // an addition after its last MFMA
    auto init = [&](T16 &acc, int gate, int goff, int nt) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 b = bias4(gate, goff, nt * 32 + 8 * a);
            acc[4 * a + 0] = b.x; acc[4 * a + 1] = b.y; acc[4 * a + 2] = b.z; acc[4 * a + 3] = b.w;
        }
    };
    if constexpr (BF) {
        // split products: the GRU as two dense layers with six output tiles each, so that every input tile is split once -
        //   X phase: [ni | r | z] += W_ih x      (g[0..5], stream Gw.r.w[0]: rows n, r, z of weight_ih, [kt][p][6 tiles])
        //   H phase: [r | z | nh] += W_hh h      (g[2..7], stream Gw.r.w[1]: rows r, z, n of weight_hh)
        // h is not read in the X phase: with `park` (16 x 64 floats per tile of LDS, [tile][reg][lane]) it waits there.
        T16 g[8];
        init(g[0], 2, Gw.ni.bias, 0);
        init(g[1], 2, Gw.ni.bias, 1);
        init(g[2], 1, Gw.r.bias, 0);
        init(g[3], 1, Gw.r.bias, 1);
        init(g[4], 3, Gw.z.bias, 0);
        init(g[5], 3, Gw.z.bias, 1);
        if (park) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) park[(nt * 16 + r) * 64 + lane] = h[nt][r];
        }
        pre = mm<6, KX, false, true>(reinterpret_cast<T16(&)[6]>(g[0]), wp, Gw.r.w[0], xin, lane, pre, Gw.r.w[1]);
        if (park) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[nt][r] = park[(nt * 16 + r) * 64 + lane];
        }
        init(g[6], 0, Gw.nh.bias, 0);
        init(g[7], 0, Gw.nh.bias, 1);
        pre = mm<6, 2, false, true>(reinterpret_cast<T16(&)[6]>(g[2]), wp, Gw.r.w[1], h, lane, pre, next_off);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ng = tanh2(g[nt][r] + sigm2(g[2 + nt][r]) * g[6 + nt][r]);
                const float zg = sigm2(g[4 + nt][r]);
                h[nt][r] = zg * (h[nt][r] - ng) + ng;        // (1 - z) n + z h
            }
        return pre;
    }
    T16 hn[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        // gates one after the other so that only one accumulator plus two 16-register results are live:
        //   nh -> r -> t = r*(W_hn h + b_hn) -> n = tanh(W_in x + b_in + t) -> z -> h' = (1-z) n + z h
        // weights of a gate are packed per output tile: offset of tile nt = nt * KT * 1024 floats
        constexpr int BS = BF ? V2_BLOCK_SPLIT : V2_BLOCK_F32;
        const int ox = nt * KX * BS, oh = nt * 2 * BS;
        T16 acc[1], tt, ng;
        init(acc[0], 0, Gw.nh.bias, nt);
        pre = mm<1, 2, false, BF>(acc, wp, Gw.nh.w[0] + oh, h, lane, pre, Gw.r.w[0] + ox);
        tt = acc[0];
        init(acc[0], 1, Gw.r.bias, nt);
        pre = mm<1, KX, false, BF>(acc, wp, Gw.r.w[0] + ox, xin, lane, pre, Gw.r.w[1] + oh);
        pre = mm<1, 2, false, BF>(acc, wp, Gw.r.w[1] + oh, h, lane, pre, Gw.ni.w[0] + ox);
#pragma unroll
        for (int r = 0; r < 16; ++r) tt[r] *= sigm2(acc[0][r]);
        init(acc[0], 2, Gw.ni.bias, nt);
        pre = mm<1, KX, false, BF>(acc, wp, Gw.ni.w[0] + ox, xin, lane, pre, Gw.z.w[0] + ox);
#pragma unroll
        for (int r = 0; r < 16; ++r) ng[r] = tanh2(acc[0][r] + tt[r]);
        init(acc[0], 3, Gw.z.bias, nt);
        pre = mm<1, KX, false, BF>(acc, wp, Gw.z.w[0] + ox, xin, lane, pre, Gw.z.w[1] + oh);
        pre = mm<1, 2, false, BF>(acc, wp, Gw.z.w[1] + oh, h, lane, pre, nt == 0 ? Gw.nh.w[0] + 2 * BS : next_off);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float zg = sigm2(acc[0][r]);
            hn[nt][r] = zg * (h[nt][r] - ng[r]) + ng[r];        // (1 - z) n + z h
        }
    }
    h[0] = hn[0];
    h[1] = hn[1];
    return pre;
}

}   // namespace v2
