// Wide goal head of DAGNet.sample (models/dagnet/model.py:215-235) for g_dim in the hundreds, one layer at a time over
// all K*N rows of the batch.  Dense layers are fp32 MFMA GEMMs (gemm_f32.hip); this file holds the row-wise and
// per-scene kernels between them and the per-step driver.  Row r = k*N + g (sample k, batch row g).
//
//   A0 = [d_t | h_last]                                   wide_goal_in_kernel
//   T1 = relu(A0 W0^T + b0 + W0[:, goal_prev])            GEMM (gather epilogue)
//   A1[:, :G] = softmax(T1 W2^T + b2)                     GEMM + wide_softmax_kernel
//   HS = A1[:, :G] [heads | folded logits]^T              GEMM (N = P*F + 2P)
//   X2 = elu(per-head attention), S2 = X2 fold_out^T      wide_attn_heads_kernel
//   X3 = out-layer attention applied to X2 (32 wide)      wide_attn_mix_kernel
//   A1[:, G:] = tanh(bn(X3 Wout^T))                       GEMM (BatchNorm folded into Wout / bias, tanh epilogue)
//   B  = A1 Wlg^T + blg                                   GEMM (K = 2G)
//   goal = argmax_g exp(softmax(B)_g) / q_g               wide_draw_kernel
// Teacher-forced pass (DAGNet.forward, models/dagnet/model.py:151-162): same layers with positions, previous goals and
// adjacency taken from the batch, ending in ce_kernel (ce -= softmax(B)[g_t]) instead of the draw.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "params_wide.h"
#include "rng.h"

extern "C" hipError_t dn_gemm_nt_gather(const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, float *C,
                                        int64_t ldc, int M, int N, int K, int act, float slope, const float *gather,
                                        const int32_t *gidx, int64_t ldg, hipStream_t st);
extern "C" hipError_t dn_launch_adjacency(int type, double sigma, int top_k, const int32_t *scene_ptr, const int32_t *row_scene,
                                          const int64_t *blk_off, int N, int T, const float *pos, float *out, int64_t ld,
                                          int64_t tstride, hipStream_t st);

namespace wide {

enum { ACT_NONE = 0, ACT_TANH = 2, ACT_RELU = 3 };    // gemm_f32.hip

// state of every (sample, row): the K samples start from the same h / position / goal
__global__ void init_state_kernel(int N, int K, int L, int R, int G, const float *__restrict__ h0, const float *__restrict__ x0,
                                  const int32_t *__restrict__ g0, float *__restrict__ h, float *__restrict__ pos,
                                  int32_t *__restrict__ goal) {
    const int64_t rows = (int64_t)N * K;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)L * rows * R;
    if (idx < total) {
        const int f = (int)(idx % R);
        const int64_t lr = idx / R;
        const int64_t r = lr % rows;
        const int l = (int)(lr / rows);
        h[idx] = h0[((int64_t)l * N + r % N) * R + f];
    }
    if (idx < rows) {
        const int g = (int)(idx % N);
        pos[idx * 2] = x0[g * 2];
        pos[idx * 2 + 1] = x0[g * 2 + 1];
        goal[idx] = min(max(g0[g], 0), G - 1);
    }
}

// A0[r] = [disposition of r's scene (flattened xy of its agents, zero padded to D) | h_last[r] | 0 pad]
__global__ void goal_in_kernel(int N, int K, int D, int R, int64_t ld, const int32_t *__restrict__ scene_ptr,
                               const int32_t *__restrict__ row_scene, const float *__restrict__ pos, const float *__restrict__ hl,
                               float *__restrict__ A0) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * K * ld) return;
    const int c = (int)(idx % ld);
    const int64_t r = idx / ld;
    float v = 0.f;
    if (c < D) {
        const int g = (int)(r % N);
        const int64_t kb = r - g;
        const int s = row_scene[g];
        const int b = scene_ptr[s], n = scene_ptr[s + 1] - b;
        const int j = c >> 1;
        if (j < n) v = pos[(kb + b + j) * 2 + (c & 1)];
    } else if (c < D + R) {
        v = hl[r * R + (c - D)];
    }
    A0[idx] = v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// in-place softmax over the first G columns of every row; one wave per row.  Rows of up to 1024 columns are read once
// (float4 pieces held in registers between the three passes).
__global__ void softmax_kernel(int64_t rows, int G, int64_t ld, float *__restrict__ x) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float *p = x + r * ld;
    constexpr float LOG2E = 1.4426950408889634f;
    if ((G & 3) == 0 && (ld & 3) == 0 && G <= 1024) {
        float4 v[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            v[i] = c4 < (G >> 2) ? *reinterpret_cast<const float4 *>(p + 4 * c4) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = __builtin_amdgcn_exp2f((v[i].x - mx) * LOG2E);
            v[i].y = __builtin_amdgcn_exp2f((v[i].y - mx) * LOG2E);
            v[i].z = __builtin_amdgcn_exp2f((v[i].z - mx) * LOG2E);
            v[i].w = __builtin_amdgcn_exp2f((v[i].w - mx) * LOG2E);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        s = wave_sum(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < (G >> 2)) *reinterpret_cast<float4 *>(p + 4 * c4) = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        }
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < G; c += 64) mx = fmaxf(mx, p[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < G; c += 64) s += expf(p[c] - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < G; c += 64) p[c] = expf(p[c] - mx) * inv;
}

// attention heads of graph_goals (gat_layers.py:25-40 with concat=True), one thread per (row, head):
//   e_ij = leakyrelu(s_src_i + s_dst_j), masked by adj > 0, softmax over the scene, h'_i = elu(sum_j a_ij Wh_j)
// and the folded logits of the out head: S2[i] = [X2_i . (W_out a_src), X2_i . (W_out a_dst)]
template <int P, int F>
__global__ void attn_heads_kernel(int N, int K, const int32_t *__restrict__ scene_ptr, const int32_t *__restrict__ row_scene,
                                  const int64_t *__restrict__ off_compact, int64_t compact, int64_t adj_ld, const float *__restrict__ adj,
                                  const float *__restrict__ HS, int64_t hs_ld, const float *__restrict__ fold_out, float alpha,
                                  float *__restrict__ X2, float *__restrict__ S2) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < (int64_t)N * K * P;
    const int p = (int)(idx % P);
    const int64_t r = live ? idx / P : 0;
    const int g = (int)(r % N), k = (int)(r / N);
    const int s = row_scene[g];
    const int b = scene_ptr[s], n = scene_ptr[s + 1] - b;
    const int64_t r0 = (int64_t)k * N + b;
    // adjacency block of the scene: compact (row stride n) or dense [N,N] (row stride adj_ld; off_compact then holds b*N+b)
    const float *arow = adj + (int64_t)k * compact + off_compact[s] + (int64_t)(g - b) * (adj_ld ? adj_ld : (int64_t)n);
    const float ssrc = HS[r * hs_ld + P * F + p];
    float mx = -INFINITY;
    for (int j = 0; j < n; ++j)
        if (arow[j] > 0.f) {
            const float e = ssrc + HS[(r0 + j) * hs_ld + P * F + P + p];
            mx = fmaxf(mx, fmaxf(e, e * alpha));
        }
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    float sum = 0.f;
    for (int j = 0; j < n; ++j)
        if (arow[j] > 0.f) {
            const float *hj = HS + (r0 + j) * hs_ld;
            const float e = ssrc + hj[P * F + P + p];
            const float w = expf(fmaxf(e, e * alpha) - mx);
            sum += w;
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += w * hj[p * F + f];
        }
    const float inv = 1.0f / sum;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float v = acc[f] * inv;
        v = v > 0.f ? v : expm1f(v);
        if (live) X2[r * (P * F) + p * F + f] = v;
        s1 += v * fold_out[p * F + f];
        s2 += v * fold_out[P * F + p * F + f];
    }
#pragma unroll
    for (int o = 1; o < P; o <<= 1) {     // the P heads of a row sit in consecutive lanes (P is a power of two)
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (live && p == 0) {
        S2[r * 2] = s1;
        S2[r * 2 + 1] = s2;
    }
}

// out head of graph_goals (concat=False, gat_model.py:18-22): the attention of the out layer is applied BEFORE its dense
// layer -- sum_j a_ij (X2_j Wout) == (sum_j a_ij X2_j) Wout -- so the per-scene kernel mixes the PF-wide (32) head
// outputs instead of the g_dim-wide products, and the GEMM that follows writes tanh(bn(.)) straight into A1[:, G:]
// (BatchNorm scale folded into Wout's rows, shift as the bias, tanh in the epilogue).  One workgroup per (sample, scene).
__global__ __launch_bounds__(256) void attn_mix_kernel(int N, int S, const int32_t *__restrict__ scene_ptr,
                                                       const int64_t *__restrict__ off_compact, int64_t compact, int64_t adj_ld,
                                                       const float *__restrict__ adj, const float *__restrict__ S2, float alpha,
                                                       const float *__restrict__ X2, int PF, float *__restrict__ X3, int64_t ld3) {
    __shared__ float att[32][33];
    const int u = blockIdx.x, k = u / S, s = u - k * S;
    const int b = scene_ptr[s], n = scene_ptr[s + 1] - b;      // n <= 32 (checked by the host)
    const int64_t r0 = (int64_t)k * N + b;
    const float *ablk = adj + (int64_t)k * compact + off_compact[s];
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const float v = S2[(r0 + i) * 2] + S2[(r0 + j) * 2 + 1];
        att[i][j] = ablk[(int64_t)i * (adj_ld ? adj_ld : (int64_t)n) + j] > 0.f ? fmaxf(v, v * alpha) : -INFINITY;
    }
    __syncthreads();
    if (tid < n) {
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, att[tid][j]);
        float sum = 0.f;
        for (int j = 0; j < n; ++j) {
            const float w = att[tid][j] == -INFINITY ? 0.f : expf(att[tid][j] - mx);
            att[tid][j] = w;
            sum += w;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < n; ++j) att[tid][j] *= inv;
    }
    __syncthreads();
    for (int e = tid; e < n * PF; e += 256) {
        const int i = e / PF, f = e - i * PF;
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += att[i][j] * X2[(r0 + j) * PF + f];
        X3[(r0 + i) * ld3 + f] = acc;
    }
}

// categorical draw of the goal (utils.py:200-207 via multinomial == argmax w/q, q ~ Exp(1)); one wave per row.
// The reference feeds w = exp(softmax probabilities) to multinomial (models/dagnet/model.py:232-233).  The race is run in the
// log domain like the register-resident kernel's (rollout_v2.hip): argmax_c  p_c * log2(e) - log2(q_c), with
// log2(q) = log2(24 - log2(n)) + const for the in-kernel stream's q = -ln(n / 2^24) - no division, hardware exp2 / log2 only.
__global__ void draw_kernel(int N, int K, int T, int t, int G, const float *__restrict__ LG, int64_t ld, const float *__restrict__ q,
                            uint64_t seed, const int32_t *__restrict__ force_goals, int32_t *__restrict__ goal,
                            int32_t *__restrict__ out_goals) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= (int64_t)N * K) return;
    const int lane = threadIdx.x & 63;
    const int g = (int)(r % N), k = (int)(r / N);
    const float *p = LG + r * ld;
    constexpr float LOG2E = 1.4426950408889634f;
    const int64_t slot = ((int64_t)k * T + t) * N + g;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (!q && (G & 3) == 0 && (ld & 3) == 0 && G <= 1024) {
        // in-kernel stream: columns 4c..4c+3 are the four words of ONE Philox block (rng.h rng_bits: counter j >> 2, word j & 3),
        // so a lane takes four consecutive columns per block; its (at most 4) float4 pieces of the row stay in registers
        float4 x[4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            x[i] = c4 < (G >> 2) ? *reinterpret_cast<const float4 *>(p + 4 * c4) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            mx = fmaxf(fmaxf(mx, fmaxf(x[i].x, x[i].y)), fmaxf(x[i].z, x[i].w));
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i].x = __builtin_amdgcn_exp2f((x[i].x - mx) * LOG2E);
            x[i].y = __builtin_amdgcn_exp2f((x[i].y - mx) * LOG2E);
            x[i].z = __builtin_amdgcn_exp2f((x[i].z - mx) * LOG2E);
            x[i].w = __builtin_amdgcn_exp2f((x[i].w - mx) * LOG2E);
            s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
        }
        s = wave_sum(s);
        const float pscale = LOG2E / s;
        // in-kernel stream: one uniform per row and step, inverse of the cumulative weights w_c = exp(p_c) (rng.h rng_draw_u) in
        // column order - piece i of lane l holds columns 4 (64 i + l) .. + 3, so the order is piece-major, lanes within a piece
        float ps[4], excl[4], tot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = lane + 64 * i < (G >> 2);
            x[i].x = on ? __builtin_amdgcn_exp2f(x[i].x * pscale) : 0.f;
            x[i].y = on ? __builtin_amdgcn_exp2f(x[i].y * pscale) : 0.f;
            x[i].z = on ? __builtin_amdgcn_exp2f(x[i].z * pscale) : 0.f;
            x[i].w = on ? __builtin_amdgcn_exp2f(x[i].w * pscale) : 0.f;
            ps[i] = ((x[i].x + x[i].y) + x[i].z) + x[i].w;
            float inc = ps[i];                          // inclusive scan over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float up = __shfl_up(inc, o);
                inc += lane >= o ? up : 0.f;
            }
            excl[i] = tot + (inc - ps[i]);
            tot += __shfl(inc, 63);
        }
        const float target = rng_draw_u(seed, (uint32_t)r, (uint32_t)t) * tot;
        // every piece that starts at or below the target proposes the column where its own cumulative sum passes it (its last
        // column if it never does); starts increase with the column, so the largest proposal is the draw
        bi = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < (G >> 2) && excl[i] <= target) {
                const float c0 = excl[i] + x[i].x, c1 = c0 + x[i].y, c2 = c1 + x[i].z;
                bi = max(bi, 4 * c4 + (c0 <= target ? 1 : 0) + (c1 <= target ? 1 : 0) + (c2 <= target ? 1 : 0));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bi = max(bi, __shfl_xor(bi, o));
        best = INFINITY;            // the reduction below keeps this lane's bi (all lanes hold the same)
    } else {
        float mx = -INFINITY;
        for (int c = lane; c < G; c += 64) mx = fmaxf(mx, p[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < G; c += 64) s += __builtin_amdgcn_exp2f((p[c] - mx) * LOG2E);
        s = wave_sum(s);
        const float pscale = LOG2E / s;
        for (int c = lane; c < G; c += 64) {
            const float qq = q ? q[slot * G + c] : rng_exp1(seed, (uint32_t)r, (uint32_t)t, (uint32_t)c);
            const float v = __builtin_amdgcn_exp2f((p[c] - mx) * LOG2E) * pscale - __builtin_amdgcn_logf(qq);
            if (v > best) { best = v; bi = c; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        if (force_goals) bi = force_goals[slot];
        bi = min(max(bi, 0), G - 1);
        goal[r] = bi;
        if (out_goals) out_goals[slot] = bi;
    }
}

// teacher-forced pass: ce_acc[r] -= softmax(LG[r])[goal_t[r]]   (models/dagnet/model.py:162), one wave per row
__global__ void ce_kernel(int64_t rows, int G, const float *__restrict__ LG, int64_t ld, const int32_t *__restrict__ goal_t,
                          float *__restrict__ ce_acc) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *p = LG + r * ld;
    float mx = -INFINITY;
    for (int c = lane; c < G; c += 64) mx = fmaxf(mx, p[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < G; c += 64) s += expf(p[c] - mx);
    s = wave_sum(s);
    if (lane == 0) {
        const int g = min(max(goal_t[r], 0), G - 1);
        ce_acc[r] -= expf(p[g] - mx) / s;
    }
}

// loss_part[2] += sum_r ce_acc[r]  (fp64, fixed order, one workgroup)
__global__ __launch_bounds__(1024) void ce_finish_kernel(int64_t rows, const float *__restrict__ ce_acc, double *__restrict__ loss_part) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 1024) s += (double)ce_acc[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[2] += red[0];
}

}   // namespace wide

struct WideStep {
    const float *wp;                  // model weight buffer
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
    // teacher-forced pass (ce_acc != NULL): K = 1, positions / previous goals / goals of the step and the adjacency come
    // from the batch; the head ends in the CE term instead of a draw
    const float *ext_pos;             // [N,2] positions of the step
    const int32_t *ext_goal_prev;     // [N]
    const int32_t *ext_goal_t;        // [N]
    const float *ext_adj;             // adjacency of the step (compact, or dense with ext_adj_ld = N)
    int64_t ext_adj_ld;
    const int64_t *ext_adj_off;       // block offsets matching ext_adj's layout
    float *ce_acc;                    // [N]
};

#define W_TRY(x)                           \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)

extern "C" hipError_t dn_wide_init_state(int N, int K, int L, int R, int G, const float *h0, const float *x0, const int32_t *g0,
                                         float *h, float *pos, int32_t *goal, hipStream_t st) {
    const int64_t total = (int64_t)L * N * K * R;
    hipLaunchKernelGGL(wide::init_state_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, N, K, L, R, G, h0, x0, g0, h,
                       pos, goal);
    return hipGetLastError();
}

// goal head of step s.t for all K*N rows: reads the state (h of the last layer, positions, previous goals), writes the new
// goals into the goal state (and out_goals)
extern "C" hipError_t dn_wide_goal_head(const WideStep *sp, hipStream_t st) {
    const WideStep &s = *sp;
    const int64_t rows64 = (int64_t)s.N * s.K;
    if (rows64 > 0x7fffffffLL) return hipErrorInvalidValue;
    const int rows = (int)rows64;
    const int PF = s.P * s.F;
    const int64_t Gp = s.b.Gp;
    const float *hl = s.b.h + (int64_t)(s.L - 1) * rows * s.R;
    if (s.P != 4 || s.F != 8) return hipErrorInvalidValue;
    const bool teacher = s.ce_acc != nullptr;
    const float *pos = teacher ? s.ext_pos : s.b.pos;
    const int32_t *gprev = teacher ? s.ext_goal_prev : s.b.goal;
    const float *adj = teacher ? s.ext_adj : s.b.adj;
    const int64_t adj_ld = teacher ? s.ext_adj_ld : 0;
    const int64_t *adj_off = teacher ? s.ext_adj_off : s.off_compact;
    // adjacency of every sample from the PRE-update positions (models/dagnet/model.py:215-220)
    if (!teacher)
        W_TRY(dn_launch_adjacency(s.adj_type, s.sigma, s.top_k, s.scene_ptr, s.row_scene, s.off_compact, s.N, s.K, s.b.pos, s.b.adj, 0,
                                  s.compact, st));
    {
        const int64_t total = rows64 * s.b.a0_ld;
        hipLaunchKernelGGL(wide::goal_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s.N, s.K, s.D, s.R, s.b.a0_ld,
                           s.scene_ptr, s.row_scene, pos, hl, s.b.A0);
        W_TRY(hipGetLastError());
    }
    W_TRY(dn_gemm_nt_gather(s.b.A0, s.b.a0_ld, s.wp + s.w.g0_w, s.w.g0_ld, s.wp + s.w.g0_b, s.b.T1, s.H, rows, s.H, (int)s.b.a0_ld,
                            wide::ACT_RELU, 0.f, s.wp + s.w.g0_gather, gprev, s.w.g0_gld, st));
    W_TRY(dn_gemm_nt_gather(s.b.T1, s.H, s.wp + s.w.g2_w, s.w.g2_ld, s.wp + s.w.g2_b, s.b.A1, 2 * Gp, rows, s.G, s.w.g2_ld,
                            wide::ACT_NONE, 0.f, nullptr, nullptr, 0, st));
    hipLaunchKernelGGL(wide::softmax_kernel, dim3((unsigned)((rows64 + 3) / 4)), dim3(256), 0, st, rows64, s.G, 2 * Gp, s.b.A1);
    W_TRY(hipGetLastError());
    W_TRY(dn_gemm_nt_gather(s.b.A1, 2 * Gp, s.wp + s.w.hf_w, s.w.hf_ld, nullptr, s.b.HS, s.b.hs_ld, rows, PF + 2 * s.P, (int)Gp,
                            wide::ACT_NONE, 0.f, nullptr, nullptr, 0, st));
    {
        const int64_t total = rows64 * s.P;
        hipLaunchKernelGGL((wide::attn_heads_kernel<4, 8>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s.N, s.K,
                           s.scene_ptr, s.row_scene, adj_off, s.compact, adj_ld, adj, s.b.HS, s.b.hs_ld, s.wp + s.w.fold_out,
                           s.alpha, s.b.X2, s.b.S2);
        W_TRY(hipGetLastError());
    }
    // X3 reuses the HS buffer (dead after attn_heads): [rows][hs_ld], first PF columns
    hipLaunchKernelGGL(wide::attn_mix_kernel, dim3((unsigned)(s.K * s.S)), dim3(256), 0, st, s.N, s.S, s.scene_ptr, adj_off, s.compact,
                       adj_ld, adj, s.b.S2, s.alpha, s.b.X2, PF, s.b.HS, s.b.hs_ld);
    W_TRY(hipGetLastError());
    W_TRY(dn_gemm_nt_gather(s.b.HS, s.b.hs_ld, s.wp + s.w.out_w, s.w.out_ld, s.wp + s.w.bn, s.b.A1 + Gp, 2 * Gp, rows, s.G, PF,
                            wide::ACT_TANH, 0.f, nullptr, nullptr, 0, st));
    W_TRY(dn_gemm_nt_gather(s.b.A1, 2 * Gp, s.wp + s.w.lg_w, s.w.lg_ld, s.wp + s.w.lg_b, s.b.B, Gp, rows, s.G, (int)(2 * Gp),
                            wide::ACT_NONE, 0.f, nullptr, nullptr, 0, st));
    if (teacher)
        hipLaunchKernelGGL(wide::ce_kernel, dim3((unsigned)((rows64 + 3) / 4)), dim3(256), 0, st, rows64, s.G, s.b.B, Gp, s.ext_goal_t,
                           s.ce_acc);
    else
        hipLaunchKernelGGL(wide::draw_kernel, dim3((unsigned)((rows64 + 3) / 4)), dim3(256), 0, st, s.N, s.K, s.T, s.t, s.G, s.b.B, Gp,
                           s.q, s.seed, s.force_goals, s.b.goal, s.out_goals);
    return hipGetLastError();
}

extern "C" hipError_t dn_wide_ce_finish(int64_t rows, const float *ce_acc, double *loss_part, hipStream_t st) {
    hipLaunchKernelGGL(wide::ce_finish_kernel, dim3(1), dim3(1024), 0, st, rows, ce_acc, loss_part);
    return hipGetLastError();
}
