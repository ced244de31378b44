// Row-wise, per-scene and optimizer kernels of the training step (row T: models/dagnet/train_bsk.py:259-291 and the
// avrnn / baseline analogues).  The differentiable forward() runs layer by layer over all N rows of the batch:
// dense layers are the fp32 MFMA GEMMs of gemm_f32.hip, everything between them lives here, each op with its
// hand-written backward.  Host side: dagnet_amd/_ops.py (torch.autograd.Function wrappers that only allocate).
//
// All matrices are row-major fp32 with an explicit row stride (ld*).  Scene structure comes from a dagnet_plan
// (scene_ptr, row -> scene, offsets of the compact block-diagonal adjacency).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

namespace tops {

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_RELU = 3 };

__device__ __forceinline__ int64_t gid() { return (int64_t)blockIdx.x * blockDim.x + threadIdx.x; }

// ---------------------------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------------------------
// dPre = dY * act'(pre), expressed through the saved output Y (slope > 0 keeps the sign)
__global__ void act_bwd_kernel(int64_t M, int N, const float *__restrict__ dY, int64_t ldd, const float *__restrict__ Y, int64_t ldy,
                               float *__restrict__ dP, int64_t ldp, int act, float slope) {
    const int64_t i = gid();
    if (i >= M * N) return;
    const int64_t r = i / N;
    const int c = (int)(i - r * N);
    const float y = Y[r * ldy + c], g = dY[r * ldd + c];
    float d = 1.f;
    if (act == ACT_LRELU) d = y > 0.f ? 1.f : slope;
    else if (act == ACT_RELU) d = y > 0.f ? 1.f : 0.f;
    else if (act == ACT_TANH) d = 1.f - y * y;
    dP[r * ldp + c] = g * d;
}

__global__ void copy2d_kernel(int64_t M, int N, const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd,
                              int accumulate) {
    const int64_t i = gid();
    if (i >= M * N) return;
    const int64_t r = i / N;
    const int c = (int)(i - r * N);
    const float v = src[r * lds + c];
    float *d = dst + r * ldd + c;
    *d = accumulate ? *d + v : v;
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

// one wave per row
__global__ void softmax_fwd_kernel(int64_t M, int N, const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= M) return;
    const int lane = threadIdx.x & 63;
    const float *x = X + r * ldx;
    float mx = -INFINITY;
    for (int c = lane; c < N; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < N; c += 64) s += expf(x[c] - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < N; c += 64) Y[r * ldy + c] = expf(x[c] - mx) * inv;
}

// Rows of at most 128 columns with 16-byte aligned rows (g_dim = 90 padded to a stride of 92): HALF a wave per row, one float4
// per lane, the row held in registers between the passes (the kernels above spend a whole wave and three passes on 90 numbers).
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ void softmax_fwd4_kernel(int64_t M, int N, const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int c = (threadIdx.x & 31) * 4;
    const bool on = r < M && c < N;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (on) v = *reinterpret_cast<const float4 *>(X + r * ldx + c);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c + i >= N) e[i] = -INFINITY;
    const float mx = half_max(fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3])));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = expf(e[i] - mx);
        s += e[i];
    }
    const float inv = 1.0f / half_sum(s);
    if (on) *reinterpret_cast<float4 *>(Y + r * ldy + c) = make_float4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
}
__global__ void softmax_bwd4_kernel(int64_t M, int N, const float *__restrict__ Y, int64_t ldy, const float *__restrict__ dY, int64_t ldd,
                                    float *__restrict__ dX, int64_t ldx) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int c = (threadIdx.x & 31) * 4;
    const bool on = r < M && c < N;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f), g = y;
    if (on) {
        y = *reinterpret_cast<const float4 *>(Y + r * ldy + c);
        g = *reinterpret_cast<const float4 *>(dY + r * ldd + c);
    }
    float ye[4] = {y.x, y.y, y.z, y.w}, ge[4] = {g.x, g.y, g.z, g.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (c + i >= N) ye[i] = 0.f, ge[i] = 0.f;        // the row's padding is uninitialised
        s += ge[i] * ye[i];
    }
    s = half_sum(s);
    if (on) *reinterpret_cast<float4 *>(dX + r * ldx + c) = make_float4(ye[0] * (ge[0] - s), ye[1] * (ge[1] - s), ye[2] * (ge[2] - s), ye[3] * (ge[3] - s));
}

// dX = Y * (dY - sum(dY * Y))
__global__ void softmax_bwd_kernel(int64_t M, int N, const float *__restrict__ Y, int64_t ldy, const float *__restrict__ dY, int64_t ldd,
                                   float *__restrict__ dX, int64_t ldx) {
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= M) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < N; c += 64) s += dY[r * ldd + c] * Y[r * ldy + c];
    s = wave_sum(s);
    for (int c = lane; c < N; c += 64) dX[r * ldx + c] = Y[r * ldy + c] * (dY[r * ldd + c] - s);
}

// ---------------------------------------------------------------------------------------------
// GRU cell (nn.GRU gate order r, z, n):  gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh  come from the GEMMs
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void gru_fwd_kernel(int64_t M, int R, const float *__restrict__ gi, const float *__restrict__ gh, const float *__restrict__ h,
                               float *__restrict__ hn, float *__restrict__ gates) {      // gates [M][3R]: r, z, n
    const int64_t i = gid();
    if (i >= M * R) return;
    const int64_t m = i / R;
    const int f = (int)(i - m * R);
    const float *a = gi + m * 3 * R, *b = gh + m * 3 * R;
    const float r = sigm(a[f] + b[f]), z = sigm(a[R + f] + b[R + f]);
    const float n = tanhf(a[2 * R + f] + r * b[2 * R + f]);
    hn[i] = (1.0f - z) * n + z * h[i];
    float *g = gates + m * 3 * R;
    g[f] = r; g[R + f] = z; g[2 * R + f] = n;
}

// -> dgi, dgh [M][3R], dh [M][R] (the direct z * dhn path only; the W_hh path is a GEMM on dgh)
__global__ void gru_bwd_kernel(int64_t M, int R, const float *__restrict__ dhn, const float *__restrict__ gates,
                               const float *__restrict__ gh, const float *__restrict__ h, float *__restrict__ dgi,
                               float *__restrict__ dgh, float *__restrict__ dh) {
    const int64_t i = gid();
    if (i >= M * R) return;
    const int64_t m = i / R;
    const int f = (int)(i - m * R);
    const float *g = gates + m * 3 * R;
    const float r = g[f], z = g[R + f], n = g[2 * R + f];
    const float ghn = gh[m * 3 * R + 2 * R + f];
    const float d = dhn[i];
    const float dn = d * (1.0f - z), dz = d * (h[i] - n);
    const float dpn = dn * (1.0f - n * n);
    const float dpr = dpn * ghn * r * (1.0f - r);
    const float dpz = dz * z * (1.0f - z);
    float *a = dgi + m * 3 * R, *b = dgh + m * 3 * R;
    a[f] = dpr; a[R + f] = dpz; a[2 * R + f] = dpn;
    b[f] = dpr; b[R + f] = dpz; b[2 * R + f] = dpn * r;
    dh[i] = d * z;
}

// ---------------------------------------------------------------------------------------------
// latent sample + losses (models/dagnet/model.py:118-135); per-row partial sums, reduced by sum_kernel
// ---------------------------------------------------------------------------------------------
// z = eps * exp(0.5 lv_e) + mu_e ;  kld_row = 0.5 * sum_z (lv_p - lv_e + (exp(lv_e) + (mu_e - mu_p)^2) / exp(lv_p) - 1)
// ZP (power of two >= Z, <= 64) consecutive lanes share a row: coalesced loads, shuffle reduction of the row sum
__global__ __launch_bounds__(256) void vae_fwd_kernel(int64_t M, int Z, int ZP, const float *__restrict__ mu_e,
                                                      const float *__restrict__ lv_e, const float *__restrict__ mu_p,
                                                      const float *__restrict__ lv_p, const float *__restrict__ eps,
                                                      float *__restrict__ z, float *__restrict__ kld_row) {
    const int c = threadIdx.x % ZP;
    const int64_t m = (int64_t)blockIdx.x * (256 / ZP) + threadIdx.x / ZP;
    float acc = 0.f;
    if (m < M && c < Z) {
        const int64_t i = m * Z + c;
        const float me = mu_e[i], le = lv_e[i], mp = mu_p[i], lp = lv_p[i];
        z[i] = eps[i] * expf(0.5f * le) + me;
        const float d = me - mp;
        acc = (lp - le) + (expf(le) + d * d) / expf(lp) - 1.0f;
    }
    for (int o = ZP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (m < M && c == 0) kld_row[m] = 0.5f * acc;
}

__global__ void vae_bwd_kernel(int64_t M, int Z, const float *__restrict__ mu_e, const float *__restrict__ lv_e,
                               const float *__restrict__ mu_p, const float *__restrict__ lv_p, const float *__restrict__ eps,
                               const float *__restrict__ dz, const float *__restrict__ g_kld, float scale, float *__restrict__ dmu_e,
                               float *__restrict__ dlv_e, float *__restrict__ dmu_p, float *__restrict__ dlv_p) {
    const int64_t i = gid();
    if (i >= M * Z) return;
    const float g = g_kld ? g_kld[0] * scale : 0.f;
    const float me = mu_e[i], le = lv_e[i], mp = mu_p[i], lp = lv_p[i];
    const float d = me - mp, ip = 1.0f / expf(lp), ee = expf(le);
    const float dzz = dz ? dz[i] : 0.f;
    dmu_e[i] = dzz + g * d * ip;
    dlv_e[i] = dzz * eps[i] * 0.5f * expf(0.5f * le) + g * 0.5f * (ee * ip - 1.0f);
    dmu_p[i] = -g * d * ip;
    dlv_p[i] = g * 0.5f * (1.0f - (ee + d * d) * ip);
}

// nll_row = 0.5 * sum_c ((x - mu)^2 / exp(lv) + lv + log(2 pi))
__global__ void nll_fwd_kernel(int64_t M, int X, const float *__restrict__ mu, const float *__restrict__ lv, const float *__restrict__ x,
                               float *__restrict__ nll_row) {
    const int64_t m = gid();
    if (m >= M) return;
    float acc = 0.f;
    for (int c = 0; c < X; ++c) {
        const int64_t i = m * X + c;
        const float d = x[i] - mu[i];
        acc += d * d / expf(lv[i]) + lv[i] + 1.8378770664093453f;
    }
    nll_row[m] = 0.5f * acc;
}

__global__ void nll_bwd_kernel(int64_t M, int X, const float *__restrict__ mu, const float *__restrict__ lv, const float *__restrict__ x,
                               const float *__restrict__ g_nll, float scale, float *__restrict__ dmu, float *__restrict__ dlv) {
    const int64_t i = gid();
    if (i >= M * X) return;
    const float g = g_nll[0] * scale;
    const float d = x[i] - mu[i], iv = 1.0f / expf(lv[i]);
    dmu[i] = -g * d * iv;
    dlv[i] = g * 0.5f * (1.0f - d * d * iv);
}

// ce_row = -p[m][goal[m]]   (the reference sums g_t * probabilities, not log-probabilities: model.py:162)
__global__ void ce_fwd_kernel(int64_t M, const float *__restrict__ p, int64_t ldp, const int32_t *__restrict__ goal,
                              float *__restrict__ ce_row) {
    const int64_t m = gid();
    if (m >= M) return;
    ce_row[m] = -p[m * ldp + goal[m]];
}

// backward of rows[m] = -softmax(x)[m][goal[m]] in one pass: dx[m][c] = g p[m][c] (p[m][goal] - [c == goal])   (the one-hot gradient
// of the probabilities and the softmax backward over it were two kernels and a [M, G] matrix of zeros)
__global__ void softmax_ce_bwd_kernel(int64_t M, int G, const float *__restrict__ p, int64_t ldp, const int32_t *__restrict__ goal,
                                      const float *__restrict__ g_ce, float *__restrict__ dx, int64_t ldx) {
    const int64_t i = gid();
    if (i >= M * G) return;
    const int64_t m = i / G;
    const int c = (int)(i - m * G);
    const int gl = goal[m];
    const float pg = p[m * ldp + gl];
    dx[m * ldx + c] = g_ce[0] * p[m * ldp + c] * (pg - (c == gl ? 1.f : 0.f));
}

__global__ void ce_bwd_kernel(int64_t M, int G, const int32_t *__restrict__ goal, const float *__restrict__ g_ce, float *__restrict__ dp,
                              int64_t ldp) {
    const int64_t i = gid();
    if (i >= M * G) return;
    const int64_t m = i / G;
    const int c = (int)(i - m * G);
    dp[m * ldp + c] = c == goal[m] ? -g_ce[0] : 0.f;
}

// out[0] (+)= scale * sum of x[0..n) in fp64, one workgroup, fixed order
__global__ __launch_bounds__(1024) void sum_kernel(int64_t n, const float *__restrict__ x, float scale, float *__restrict__ out, int accumulate) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)((accumulate ? (double)out[0] : 0.0) + red[0] * (double)scale);
}

// ---------------------------------------------------------------------------------------------
// column reductions over the rows (bias gradients, BatchNorm statistics): two deterministic stages
// ---------------------------------------------------------------------------------------------
// stage 1: grid (ceil(N/64), splits), block 256 = 64 columns x 4 row lanes; part[split][q][N]
// MODE 0: q0 = sum x                                   (bias gradient, BN mean)
// MODE 1: q0 = sum (x - mu)^2                          (BN variance, second pass)
// MODE 2: q0 = sum g, q1 = sum g * xhat, g = dOut * (1 - out^2) [tanh] or dOut;  xhat = (x - mu) * invstd
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(int64_t M, int N, int rows_per_split, const float *__restrict__ X, int64_t ldx,
                                                         const float *__restrict__ stat, const float *__restrict__ dOut, int64_t ldd,
                                                         const float *__restrict__ Out, int64_t ldo, int tanh_out,
                                                         float *__restrict__ part) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    float q0 = 0.f, q1 = 0.f;
    if (c < N) {
        const float mu = MODE >= 1 ? stat[c] : 0.f, is = MODE == 2 ? stat[N + c] : 0.f;
        // four rows in flight per thread: the loop is a chain of dependent adds fed by loads the compiler will not batch by itself
#pragma unroll 4
        for (int64_t m = m0 + rl; m < m1; m += 4) {
            if (MODE == 0) q0 += X[m * ldx + c];
            else if (MODE == 1) { const float d = X[m * ldx + c] - mu; q0 += d * d; }
            else {
                float g = dOut[m * ldd + c];
                if (tanh_out) { const float o = Out[m * ldo + c]; g *= 1.0f - o * o; }
                q0 += g;
                q1 += g * (X[m * ldx + c] - mu) * is;
            }
        }
    }
    red[0][rl][threadIdx.x & 63] = q0;
    red[1][rl][threadIdx.x & 63] = q1;
    __syncthreads();
    if (rl == 0 && c < N) {
        const int l = threadIdx.x;
        part[((int64_t)blockIdx.y * 2 + 0) * N + c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        part[((int64_t)blockIdx.y * 2 + 1) * N + c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}

// The same for 16-byte rows (N and the row strides multiples of 4, aligned bases): a thread owns four columns and every fourth
// row lane-group, a wave-load covers four rows of 64 columns, 16 rows in flight per wave and operand.  (The scalar form keeps
// 3 KB in flight per wave: the BatchNorm backward statistics ran at 1.2 TB/s.)
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce4_kernel(int64_t M, int N, int rows_per_split, const float *__restrict__ X, int64_t ldx,
                                                          const float *__restrict__ stat, const float *__restrict__ dOut, int64_t ldd,
                                                          const float *__restrict__ Out, int64_t ldo, int tanh_out,
                                                          float *__restrict__ part) {
    __shared__ float red[2][16][64 + 4];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        // N need not be a multiple of 4 (g_dim = 90 in rows of 92): the columns of a straddling quad past N read the row padding
        // into sums of their own that are never written; the statistics are fetched element by element
        float4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
        if (MODE >= 1) mu = make_float4(stat[c], c + 1 < N ? stat[c + 1] : 0.f, c + 2 < N ? stat[c + 2] : 0.f, c + 3 < N ? stat[c + 3] : 0.f);
        if (MODE == 2)
            is = make_float4(stat[N + c], c + 1 < N ? stat[N + c + 1] : 0.f, c + 2 < N ? stat[N + c + 2] : 0.f, c + 3 < N ? stat[N + c + 3] : 0.f);
#pragma unroll 4
        for (int64_t m = m0 + rl; m < m1; m += 16) {
            const float4 x = *reinterpret_cast<const float4 *>(X + m * ldx + c);
            if (MODE == 0) { q0.x += x.x; q0.y += x.y; q0.z += x.z; q0.w += x.w; }
            else if (MODE == 1) {
                const float d0 = x.x - mu.x, d1 = x.y - mu.y, d2 = x.z - mu.z, d3 = x.w - mu.w;
                q0.x += d0 * d0; q0.y += d1 * d1; q0.z += d2 * d2; q0.w += d3 * d3;
            } else {
                float4 g = *reinterpret_cast<const float4 *>(dOut + m * ldd + c);
                if (tanh_out) {
                    const float4 o = *reinterpret_cast<const float4 *>(Out + m * ldo + c);
                    g.x *= 1.0f - o.x * o.x; g.y *= 1.0f - o.y * o.y; g.z *= 1.0f - o.z * o.z; g.w *= 1.0f - o.w * o.w;
                }
                q0.x += g.x; q0.y += g.y; q0.z += g.z; q0.w += g.w;
                q1.x += g.x * (x.x - mu.x) * is.x; q1.y += g.y * (x.y - mu.y) * is.y;
                q1.z += g.z * (x.z - mu.z) * is.z; q1.w += g.w * (x.w - mu.w) * is.w;
            }
        }
    }
    *reinterpret_cast<float4 *>(&red[0][rl][cl * 4]) = q0;
    *reinterpret_cast<float4 *>(&red[1][rl][cl * 4]) = q1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int q = threadIdx.x >> 6, l = threadIdx.x & 63, col = blockIdx.x * 64 + l;
        if (col < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[q][r][l];
            part[((int64_t)blockIdx.y * 2 + q) * N + col] = t;
        }
    }
}

// BatchNorm forward statistics in ONE pass over X (mean and sum of squared deviations), for 16-byte rows: part[split] = [mean_s | M2_s]
// of the split's rows.  A thread sums (x - K) and (x - K)^2 with K = its own first value (deviations of the size of the spread: no
// E[x^2] - mean^2 cancellation), turns them into (mean, M2) and the 16 row lanes merge pairwise with the parallel-variance formula
// M2 = M2_a + M2_b + d^2 n_a n_b / n; bn_finish_stat_kernel merges the splits in fp64.  (Two passes + three small launches before.)
__global__ __launch_bounds__(256) void bn_moments_kernel(int64_t M, int N, int rows_per_split, const float *__restrict__ X, int64_t ldx,
                                                         float *__restrict__ part) {
    __shared__ float smean[16][64 + 4], sm2[16][64 + 4], scnt[16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_split, m1 = min(M, m0 + rows_per_split);
    float4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, k = {0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    if (c < N && m0 + rl < m1) {
        k = *reinterpret_cast<const float4 *>(X + (m0 + rl) * ldx + c);
#pragma unroll 4
        for (int64_t m = m0 + rl; m < m1; m += 16) {
            const float4 x = *reinterpret_cast<const float4 *>(X + m * ldx + c);
            const float d0 = x.x - k.x, d1 = x.y - k.y, d2 = x.z - k.z, d3 = x.w - k.w;
            s1.x += d0; s1.y += d1; s1.z += d2; s1.w += d3;
            s2.x += d0 * d0; s2.y += d1 * d1; s2.z += d2 * d2; s2.w += d3 * d3;
            ++cnt;
        }
    }
    const float inv = cnt ? 1.0f / (float)cnt : 0.f;
    const float4 mean = {k.x + s1.x * inv, k.y + s1.y * inv, k.z + s1.z * inv, k.w + s1.w * inv};
    const float4 m2 = {s2.x - s1.x * s1.x * inv, s2.y - s1.y * s1.y * inv, s2.z - s1.z * s1.z * inv, s2.w - s1.w * s1.w * inv};
    *reinterpret_cast<float4 *>(&smean[rl][cl * 4]) = mean;
    *reinterpret_cast<float4 *>(&sm2[rl][cl * 4]) = m2;
    if (cl == 0) scnt[rl] = (float)cnt;          // the same for all columns of a row lane
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x, col = blockIdx.x * 64 + l;
        if (col < N) {
            float n = scnt[0], mu = smean[0][l], q = sm2[0][l];
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const float nb = scnt[r];
                if (nb > 0.f) {
                    const float nn = n + nb, d = smean[r][l] - mu;
                    mu += d * (nb / nn);
                    q += sm2[r][l] + d * d * (n * nb / nn);
                    n = nn;
                }
            }
            part[((int64_t)blockIdx.y * 2 + 0) * N + col] = mu;
            part[((int64_t)blockIdx.y * 2 + 1) * N + col] = q;
        }
    }
}

// merge of the splits' (mean, M2) in fp64 and everything bn_stat_kernel does: stat = [mean | invstd], running statistics
__global__ __launch_bounds__(256) void bn_finish_stat_kernel(int N, int64_t M, int splits, int rows_per_split, const float *__restrict__ part,
                                                             float eps, float momentum, float *__restrict__ stat,
                                                             float *__restrict__ run_mean, float *__restrict__ run_var) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= N) return;
    double n = 0.0, mu = 0.0, q = 0.0;
    for (int k = lane; k < splits; k += 64) {
        const int64_t m0 = (int64_t)k * rows_per_split;
        const double nb = (double)(min(M, m0 + rows_per_split) - m0);
        if (nb <= 0.0) continue;
        const double mb = part[((int64_t)k * 2 + 0) * N + c], qb = part[((int64_t)k * 2 + 1) * N + c];
        const double nn = n + nb, d = mb - mu;
        mu += d * (nb / nn);
        q += qb + d * d * (n * nb / nn);
        n = nn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double nb = __shfl_xor(n, o), mb = __shfl_xor(mu, o), qb = __shfl_xor(q, o);
        const double nn = n + nb;
        if (nn > 0.0) {
            const double d = mb - mu;
            // symmetric form: both partners of an exchange arrive at the same triple
            mu = (n * mu + nb * mb) / nn;
            q = q + qb + d * d * (n * nb / nn);
        }
        n = nn;
    }
    if (lane == 0) {
        const float mean = (float)mu, var_sum = (float)q;
        const float var = var_sum / (float)M;
        stat[c] = mean;
        stat[N + c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
            const float unb = M > 1 ? var_sum / (float)(M - 1) : var;
            run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
        }
    }
}

// Synchronised BatchNorm (data-parallel training: nn.BatchNorm1d of the reference sees the rows of the WHOLE batch,
// gat_model.py:21, gcn_model.py:16,18).  A shard's statistics as a mergeable triple per column: mom = [count | mean | M2].
// Same merge of the splits as bn_finish_stat_kernel.
__global__ __launch_bounds__(256) void bn_moments_finish_kernel(int N, int64_t M, int splits, int rows_per_split, const float *__restrict__ part,
                                                                float *__restrict__ mom) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= N) return;
    double n = 0.0, mu = 0.0, q = 0.0;
    for (int k = lane; k < splits; k += 64) {
        const int64_t m0 = (int64_t)k * rows_per_split;
        const double nb = (double)(min(M, m0 + rows_per_split) - m0);
        if (nb <= 0.0) continue;
        const double mb = part[((int64_t)k * 2 + 0) * N + c], qb = part[((int64_t)k * 2 + 1) * N + c];
        const double nn = n + nb, d = mb - mu;
        mu += d * (nb / nn);
        q += qb + d * d * (n * nb / nn);
        n = nn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double nb = __shfl_xor(n, o), mb = __shfl_xor(mu, o), qb = __shfl_xor(q, o);
        const double nn = n + nb;
        if (nn > 0.0) {
            const double d = mb - mu;
            mu = (n * mu + nb * mb) / nn;
            q = q + qb + d * d * (n * nb / nn);
        }
        n = nn;
    }
    if (lane == 0) {
        mom[c] = (float)M;
        mom[N + c] = (float)mu;
        mom[2 * N + c] = (float)q;
    }
}
// the two-pass fallback's (mean, sum of squared deviations) as the same triple
__global__ void bn_moments_pack_kernel(int N, int64_t M, const float *__restrict__ mean, const float *__restrict__ var_sum, float *__restrict__ mom) {
    const int c = (int)gid();
    if (c >= N) return;
    mom[c] = (float)M;
    mom[N + c] = mean[c];
    mom[2 * N + c] = var_sum[c];
}
// Merge of the shards' triples, in shard order and fp64 (every rank holds the same gathered list, so every rank arrives at the
// same bits), then what bn_stat_kernel does with the merged statistics: stat = [mean | invstd], running statistics with the
// unbiased variance over ALL rows.  total (nullable): the merged row count, for the backward pass.
__global__ void bn_merge_kernel(int N, int n_parts, const float *__restrict__ parts, float eps, float momentum, float *__restrict__ stat,
                                float *__restrict__ run_mean, float *__restrict__ run_var, float *__restrict__ total) {
    const int c = (int)gid();
    if (c >= N) return;
    double n = 0.0, mu = 0.0, q = 0.0;
    for (int k = 0; k < n_parts; ++k) {
        const float *pk = parts + (int64_t)k * 3 * N;
        const double nb = pk[c];
        if (nb <= 0.0) continue;
        const double nn = n + nb, d = (double)pk[N + c] - mu;
        mu += d * (nb / nn);
        q += (double)pk[2 * N + c] + d * d * (n * nb / nn);
        n = nn;
    }
    const float mean = (float)mu, var_sum = (float)q;
    const float var = n > 0.0 ? var_sum / (float)n : 0.f;
    stat[c] = mean;
    stat[N + c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
        const float unb = n > 1.0 ? var_sum / (float)(n - 1.0) : var;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
    }
    if (total && c == 0) total[0] = (float)n;
}

// stage 2: out_q[c] (+)= scale * sum_split part[split][q][c]; one wave per column, lanes stride over the splits
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// extra0 / extra1 (nullable): the two sums are also added into (or stored to) a second pair of vectors - the BatchNorm backward's
// dbeta / dgamma, which were two copy launches of their own
__global__ __launch_bounds__(256) void col_finish_kernel(int N, int splits, const float *__restrict__ part, float scale,
                                                         float *__restrict__ out0, float *__restrict__ out1, int accumulate,
                                                         float *__restrict__ extra0, float *__restrict__ extra1, int extra_accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = lane; k < splits; k += 64) {
        s0 += (double)part[((int64_t)k * 2 + 0) * N + c];
        if (out1) s1 += (double)part[((int64_t)k * 2 + 1) * N + c];
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) {
        if (out0) out0[c] = (accumulate ? out0[c] : 0.f) + (float)(s0 * scale);
        if (out1) out1[c] = (accumulate ? out1[c] : 0.f) + (float)(s1 * scale);
        if (extra0) extra0[c] = (extra_accumulate ? extra0[c] : 0.f) + (float)(s0 * scale);
        if (extra1) extra1[c] = (extra_accumulate ? extra1[c] : 0.f) + (float)(s1 * scale);
    }
}

// BatchNorm1d in train mode (+ optional tanh): stat = [mean | invstd]; running statistics follow nn.BatchNorm1d
// (momentum 0.1, unbiased variance).  var_sum = sum (x - mean)^2.
__global__ void bn_stat_kernel(int N, int64_t M, const float *__restrict__ mean, const float *__restrict__ var_sum, float eps,
                               float momentum, float *__restrict__ stat, float *__restrict__ run_mean, float *__restrict__ run_var) {
    const int c = (int)gid();
    if (c >= N) return;
    const float var = var_sum[c] / (float)M;
    stat[c] = mean[c];
    stat[N + c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[c];
        const float unb = M > 1 ? var_sum[c] / (float)(M - 1) : var;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
    }
}

__global__ void bn_apply_kernel(int64_t M, int N, const float *__restrict__ X, int64_t ldx, const float *__restrict__ stat,
                                const float *__restrict__ gamma, const float *__restrict__ beta, int tanh_out, float *__restrict__ Y,
                                int64_t ldy) {
    const int64_t i = gid();
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int c = (int)(i - m * N);
    const float v = (X[m * ldx + c] - stat[c]) * stat[N + c] * gamma[c] + beta[c];
    Y[m * ldy + c] = tanh_out ? tanhf(v) : v;
}

// dX = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  sums = [sum g | sum g xhat]
__global__ void bn_bwd_kernel(int64_t M, int N, const float *__restrict__ X, int64_t ldx, const float *__restrict__ stat,
                              const float *__restrict__ gamma, const float *__restrict__ dOut, int64_t ldd,
                              const float *__restrict__ Out, int64_t ldo, int tanh_out, const float *__restrict__ sums,
                              float *__restrict__ dX, int64_t ldg, int64_t Mtot) {
    // Mtot: rows the statistics (and `sums`) were taken over - M, or the rows of ALL shards with synchronised statistics
    const int64_t i = gid();
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int c = (int)(i - m * N);
    float g = dOut[m * ldd + c];
    if (tanh_out) { const float o = Out[m * ldo + c]; g *= 1.0f - o * o; }
    const float xhat = (X[m * ldx + c] - stat[c]) * stat[N + c];
    const float inv_m = 1.0f / (float)Mtot;
    dX[m * ldg + c] = gamma[c] * stat[N + c] * (g - sums[c] * inv_m - xhat * sums[N + c] * inv_m);
}

// ---------------------------------------------------------------------------------------------
// per-scene graph kernels on the compact block-diagonal adjacency
// ---------------------------------------------------------------------------------------------
struct Scenes {
    const int32_t *scene_ptr;
    const int64_t *off;        // start of every scene's block: compact offsets, or b * N + b for a dense [N, N] matrix
    int S;
    int64_t ld;                // 0: compact blocks (row stride = the scene's agent count); else the dense row stride N
};

#define GAT_T 64
// Global -> LDS staging for the single-wave attention workgroups: the loads of up to U elements per lane are issued back to
// back and only then written to LDS.  Written as `for (e) lds[e] = f(global[e])`, the compiler keeps ONE load in flight per
// iteration (load, s_waitcnt vmcnt(0), ds_write, next load): ten serial HBM round trips for a 10 x 64 scene slab, which was
// most of these kernels' 35 - 70 us.  load(e) must be safe for every e < total (it is called with e clamped, unconditionally,
// so that no load sits behind a branch); store(e, v) runs for e < total only.
struct GatV3 { float a, b, c; };
template <int U, class V, class L, class S>
__device__ __forceinline__ void gat_stage(int total, int tid, L load, S store) {
    for (int e0 = tid; e0 < total; e0 += GAT_T * U) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load(min(e0 + u * GAT_T, total - 1));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u * GAT_T < total) store(e0 + u * GAT_T, v[u]);
    }
}
// e / d for 0 <= e < 2^16 with a divisor that is only known at run time (agents of the scene, head width): one v_mul_hi instead of
// the ~35-instruction unsigned division - these kernels index everything by (row, feature) or (row, row) pairs of a flat e
struct FastDiv {
    uint32_t m;
    int d;
    __device__ explicit FastDiv(int d_) : m(d_ > 1 ? 0xffffffffu / (uint32_t)d_ + 1u : 0u), d(d_) {}
    __device__ __forceinline__ int operator()(int e) const { return d > 1 ? (int)__umulhi((uint32_t)e, m) : e; }
};
// GAT attention of one layer (gat_layers.py:25-40) for H heads of width F stacked along the columns of Wh:
//   s1_i = Wh_i . a[:F], s2_j = Wh_j . a[F:], e_ij = leakyrelu(s1_i + s2_j) where adj_ij > 0, att = softmax_j, out = att Wh (elu)
// grid (S, H); dynamic LDS: Wh [n][F] | s1 [n] | s2 [n] | att [n][n].  Pair (i,j) and (row, feature) loops are spread over
// the 64 threads of ONE wave per (scene, head) (single-wave workgroups: barriers cost nothing, many scenes per CU); only the O(n) row normalisations run on n threads.
// Rows of Wh / dO sit in LDS with a stride FP = 4 ceil(F / 4), zero padded: the n^2 F loops (d att = dO . Wh^T, dWh = att^T dO,
// out = att Wh) then move float4s - one ds_read_b128 per four multiply-adds instead of two ds_read_b32 per one.
__device__ __forceinline__ void gat_logits(int n, int F, int FP, int64_t rld, const float *wh, const float *ah, const float *ablk,
                                           float alpha, float *s1, float *s2, float *att, int tid) {
    {   // rows of wh and both halves of ah are zero-padded to FP: float4 dot products
        const float4 *wh4 = reinterpret_cast<const float4 *>(wh), *ah4 = reinterpret_cast<const float4 *>(ah);
        const int F4 = FP >> 2;
        for (int e = tid; e < 2 * n; e += GAT_T) {
            const int i = e >> 1, w = e & 1;
            float u = 0.f;
            for (int q = 0; q < F4; ++q) {
                const float4 x = wh4[i * F4 + q], y = ah4[w * F4 + q];
                u += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
            }
            (w ? s2 : s1)[i] = u;
        }
    }
    __syncthreads();
    const FastDiv dn(n);
    gat_stage<4, float>(n * n, tid, [&](int e) { const int i = dn(e); return ablk[i * rld + (e - i * n)]; },
                        [&](int e, float adjv) {
                            const int i = dn(e), j = e - i * n;
                            const float v = s1[i] + s2[j];
                            att[e] = adjv > 0.f ? fmaxf(v, v * alpha) : -INFINITY;
                        });
    __syncthreads();
    for (int i = tid; i < n; i += GAT_T) {
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, att[i * n + j]);
        float sum = 0.f;
        for (int j = 0; j < n; ++j) {
            const float w = att[i * n + j] == -INFINITY ? 0.f : expf(att[i * n + j] - mx);
            att[i * n + j] = w;
            sum += w;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < n; ++j) att[i * n + j] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(GAT_T) void gat_attn_fwd_kernel(Scenes sc, const float *__restrict__ adj, const float *__restrict__ Wh,
                                                          int64_t ldw, const float *__restrict__ a, int F, float alpha, int elu,
                                                          float *__restrict__ Out, int64_t ldo) {
    extern __shared__ __align__(16) float sm[];
    const int s = blockIdx.x, hd = blockIdx.y;
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b, FP = (F + 3) & ~3, F4 = FP >> 2;
    float *wh = sm, *ah = wh + n * FP, *s1 = ah + 2 * FP, *s2 = s1 + n, *att = s2 + n;      // ah: [a[:F] | 0.. | a[F:] | 0..], FP each
    const float *ablk = adj + sc.off[s];
    const int64_t rld = sc.ld ? sc.ld : n;
    const int tid = threadIdx.x;
    const FastDiv dFP(FP), dF4(F4);
    gat_stage<4, float>(2 * FP, tid,                           // the attention vector: read F times per row below
                        [&](int e) { const int w = e >= FP, f = min(e - w * FP, F - 1); return a[(int64_t)hd * 2 * F + w * F + f]; },
                        [&](int e, float v) { const int w = e >= FP; ah[e] = e - w * FP < F ? v : 0.f; });
    gat_stage<12, float>(n * FP, tid,
                         [&](int e) { const int i = dFP(e), f = min(e - i * FP, F - 1); return Wh[(int64_t)(b + i) * ldw + hd * F + f]; },
                         [&](int e, float v) { const int i = dFP(e); wh[e] = e - i * FP < F ? v : 0.f; });
    __syncthreads();
    gat_logits(n, F, FP, rld, wh, ah, ablk, alpha, s1, s2, att, tid);
    const float4 *wh4 = reinterpret_cast<const float4 *>(wh);
    for (int e = tid; e < n * F4; e += GAT_T) {
        const int i = dF4(e), q = e - i * F4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < n; ++j) {
            const float w = att[i * n + j];
            const float4 v = wh4[j * F4 + q];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = 4 * q + c;
            if (f < F) {
                float v = o[c];
                if (elu) v = v > 0.f ? v : expm1f(v);
                Out[(int64_t)(b + i) * ldo + hd * F + f] = v;
            }
        }
    }
}

// backward of the above.  dynamic LDS: Wh [n][FP] | dO [n][FP] | s1 | s2 | ds1 | ds2 | dot [n] each | att [n][n] | dpre [n][n]
// da_part [S][H x 2F]: this scene's contribution to the gradient of a; the caller sums the column (same-address atomics
// over thousands of scenes serialise in the L2: 260 us per call before, and the sum order was not fixed).
__global__ __launch_bounds__(GAT_T) void gat_attn_bwd_kernel(Scenes sc, const float *__restrict__ adj, const float *__restrict__ Wh,
                                                          int64_t ldw, const float *__restrict__ a, int F, float alpha, int elu,
                                                          const float *__restrict__ Out, int64_t ldo, const float *__restrict__ dOut,
                                                          int64_t ldd, float *__restrict__ dWh, int64_t ldg, float *__restrict__ da_part) {
    extern __shared__ __align__(16) float sm[];
    const int s = blockIdx.x, hd = blockIdx.y;
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b, FP = (F + 3) & ~3, F4 = FP >> 2;
    float *wh = sm, *dO = wh + n * FP, *ah = dO + n * FP, *s1 = ah + 2 * FP, *s2 = s1 + n, *ds1 = s2 + n, *ds2 = ds1 + n, *dot = ds2 + n;
    float *att = dot + n, *dpre = att + n * n;
    const float *ablk = adj + sc.off[s];
    const int64_t rld = sc.ld ? sc.ld : n;
    const int tid = threadIdx.x;
    const FastDiv dFP(FP), dF4(F4), dn(n);
    gat_stage<4, float>(2 * FP, tid,
                        [&](int e) { const int w = e >= FP, f = min(e - w * FP, F - 1); return a[(int64_t)hd * 2 * F + w * F + f]; },
                        [&](int e, float v) { const int w = e >= FP; ah[e] = e - w * FP < F ? v : 0.f; });
    gat_stage<6, GatV3>(n * FP, tid,
                        [&](int e) {
                            const int ii = dFP(e), f = min(e - ii * FP, F - 1);
                            const int64_t row = b + ii;
                            const int col = hd * F + f;
                            return GatV3{Wh[row * ldw + col], dOut[row * ldd + col], elu ? Out[row * ldo + col] : 1.f};
                        },
                        [&](int e, GatV3 v) {
                            const int ii = dFP(e);
                            const bool in = e - ii * FP < F;
                            float g = v.b;
                            if (elu) g *= v.c > 0.f ? 1.f : v.c + 1.f;
                            wh[e] = in ? v.a : 0.f;
                            dO[e] = in ? g : 0.f;
                        });
    __syncthreads();
    gat_logits(n, F, FP, rld, wh, ah, ablk, alpha, s1, s2, att, tid);
    const float4 *wh4 = reinterpret_cast<const float4 *>(wh), *dO4 = reinterpret_cast<const float4 *>(dO);
    // d att_ij = dO_i . Wh_j
    for (int e = tid; e < n * n; e += GAT_T) {
        const int i = dn(e), j = e - i * n;
        float d = 0.f;
        for (int q = 0; q < F4; ++q) {
            const float4 u = dO4[i * F4 + q], v = wh4[j * F4 + q];
            d += (u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w);
        }
        dpre[e] = d;
    }
    __syncthreads();
    for (int i = tid; i < n; i += GAT_T) {
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += att[i * n + j] * dpre[i * n + j];
        dot[i] = acc;
    }
    __syncthreads();
    // d e_ij = att_ij (d att_ij - sum_k att_ik d att_ik), through the leaky relu
    for (int e = tid; e < n * n; e += GAT_T) {
        const int i = dn(e), j = e - i * n;
        const float v = s1[i] + s2[j];
        dpre[e] = att[e] * (dpre[e] - dot[i]) * (v > 0.f ? 1.f : alpha);
    }
    __syncthreads();
    for (int e = tid; e < 2 * n; e += GAT_T) {
        const int i = e >> 1;
        float acc = 0.f;
        if (e & 1) { for (int k = 0; k < n; ++k) acc += dpre[k * n + i]; ds2[i] = acc; }
        else       { for (int k = 0; k < n; ++k) acc += dpre[i * n + k]; ds1[i] = acc; }
    }
    __syncthreads();
    for (int e = tid; e < n * F4; e += GAT_T) {
        const int j = dF4(e), q = e - j * F4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < n; ++i) {
            const float w = att[i * n + j];
            const float4 v = dO4[i * F4 + q];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        const float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = 4 * q + c;
            if (f < F) dWh[(int64_t)(b + j) * ldg + hd * F + f] = o[c] + ds1[j] * ah[f] + ds2[j] * ah[FP + f];
        }
    }
    for (int f = tid; f < 2 * F; f += GAT_T) {
        const float *ds = f < F ? ds1 : ds2;
        const int ff = f < F ? f : f - F;
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += ds[i] * wh[i * FP + ff];
        da_part[((int64_t)s * gridDim.y + hd) * 2 * F + f] = acc;
    }
}

// All H heads of a scene in ONE single-wave workgroup (H > 1: the hidden and goal graphs' 4 x 8 heads).  The per-(scene, head)
// kernels above are bound by the latency chain of a workgroup's life (scene_ptr -> block offset -> Wh, then eight short phases),
// not by arithmetic: 4 x 8192 workgroups of a few hundred FMAs each ran 35 us forward / 69 us backward per call.  Here the
// scene's [n][H F] slab of Wh is read once (whole rows, coalesced), every phase runs over (head, ...) jointly - the same
// number of barriers for four heads as for one - and a quarter of the workgroups is dispatched.
// dynamic LDS forward: wh [n][HF] | s1 [H][n] | s2 [H][n] | att [H][n][n]
__device__ __forceinline__ void gat_logits_heads(int n, int H, int F, int64_t rld, const float *wh, const float *a, const float *ablk,
                                                 float alpha, float *s1, float *s2, float *att, int tid) {
    const int HF = H * F, nn = n * n;
    const FastDiv d2n(2 * n), dnn(nn), dn(n);
    for (int e = tid; e < H * 2 * n; e += GAT_T) {
        const int hd = d2n(e), r = e - hd * 2 * n, i = r >> 1, w = r & 1;
        const float *ah = a + (int64_t)hd * 2 * F + w * F;
        float u = 0.f;
        for (int f = 0; f < F; ++f) u += wh[i * HF + hd * F + f] * ah[f];
        (w ? s2 : s1)[hd * n + i] = u;
    }
    __syncthreads();
    gat_stage<8, float>(H * nn, tid,
                        [&](int e) { const int hd = dnn(e), ij = e - hd * nn, i = dn(ij); return ablk[i * rld + (ij - i * n)]; },
                        [&](int e, float adjv) {
                            const int hd = dnn(e), ij = e - hd * nn, i = dn(ij), j = ij - i * n;
                            const float v = s1[hd * n + i] + s2[hd * n + j];
                            att[e] = adjv > 0.f ? fmaxf(v, v * alpha) : -INFINITY;
                        });
    __syncthreads();
    for (int e = tid; e < H * n; e += GAT_T) {
        float *row = att + (int64_t)e * n;          // e = hd * n + i
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, row[j]);
        float sum = 0.f;
        for (int j = 0; j < n; ++j) {
            const float w = row[j] == -INFINITY ? 0.f : expf(row[j] - mx);
            row[j] = w;
            sum += w;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < n; ++j) row[j] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(GAT_T) void gat_attn_fwd_heads_kernel(Scenes sc, const float *__restrict__ adj, const float *__restrict__ Wh,
                                                                int64_t ldw, const float *__restrict__ a, int H, int F, float alpha, int elu,
                                                                float *__restrict__ Out, int64_t ldo) {
    extern __shared__ float sm[];
    const int s = blockIdx.x;
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b, HF = H * F, nn = n * n;
    float *wh = sm, *s1 = wh + n * HF, *s2 = s1 + H * n, *att = s2 + H * n, *as = att + H * nn;
    const float *ablk = adj + sc.off[s];
    const int64_t rld = sc.ld ? sc.ld : n;
    const int tid = threadIdx.x;
    const FastDiv dHF(HF), dF(F);
    gat_stage<2, float>(H * 2 * F, tid, [&](int e) { return a[e]; }, [&](int e, float v) { as[e] = v; });      // the heads' attention vectors
    gat_stage<8, float>(n * HF, tid, [&](int e) { const int i = dHF(e); return Wh[(int64_t)(b + i) * ldw + (e - i * HF)]; },
                        [&](int e, float v) { wh[e] = v; });
    __syncthreads();
    gat_logits_heads(n, H, F, rld, wh, as, ablk, alpha, s1, s2, att, tid);
    for (int e = tid; e < n * HF; e += GAT_T) {
        const int i = dHF(e), c = e - i * HF, hd = dF(c);
        const float *arow = att + hd * nn + i * n;
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += arow[j] * wh[j * HF + c];
        if (elu) acc = acc > 0.f ? acc : expm1f(acc);
        Out[(int64_t)(b + i) * ldo + c] = acc;
    }
}

// dynamic LDS backward: wh [n][HF] | dO [n][HF] | s1 | s2 | ds1 | ds2 | dot [H][n] each | att [H][n][n] | dpre [H][n][n]
__global__ __launch_bounds__(GAT_T) void gat_attn_bwd_heads_kernel(Scenes sc, const float *__restrict__ adj, const float *__restrict__ Wh,
                                                                int64_t ldw, const float *__restrict__ a, int H, int F, float alpha, int elu,
                                                                const float *__restrict__ Out, int64_t ldo, const float *__restrict__ dOut,
                                                                int64_t ldd, float *__restrict__ dWh, int64_t ldg,
                                                                float *__restrict__ da_part) {
    extern __shared__ float sm[];
    const int s = blockIdx.x;
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b, HF = H * F, nn = n * n, Hn = H * n;
    float *wh = sm, *dO = wh + n * HF, *s1 = dO + n * HF, *s2 = s1 + Hn, *ds1 = s2 + Hn, *ds2 = ds1 + Hn, *dot = ds2 + Hn;
    float *att = dot + Hn, *dpre = att + H * nn, *as = dpre + H * nn;
    const float *ablk = adj + sc.off[s];
    const int64_t rld = sc.ld ? sc.ld : n;
    const int tid = threadIdx.x;
    const FastDiv dHF(HF), dF(F), dnn(nn), dn(n), d2F(2 * F);
    gat_stage<2, float>(H * 2 * F, tid, [&](int e) { return a[e]; }, [&](int e, float v) { as[e] = v; });
    gat_stage<6, GatV3>(n * HF, tid,
                        [&](int e) {
                            const int i = dHF(e), c = e - i * HF;
                            const int64_t row = b + i;
                            return GatV3{Wh[row * ldw + c], dOut[row * ldd + c], elu ? Out[row * ldo + c] : 1.f};
                        },
                        [&](int e, GatV3 v) {
                            float g = v.b;
                            if (elu) g *= v.c > 0.f ? 1.f : v.c + 1.f;
                            wh[e] = v.a;
                            dO[e] = g;
                        });
    __syncthreads();
    gat_logits_heads(n, H, F, rld, wh, as, ablk, alpha, s1, s2, att, tid);
    // d att_ij = dO_i . Wh_j (per head)
    for (int e = tid; e < H * nn; e += GAT_T) {
        const int hd = dnn(e), ij = e - hd * nn, i = dn(ij), j = ij - i * n;
        float d = 0.f;
        for (int f = 0; f < F; ++f) d += dO[i * HF + hd * F + f] * wh[j * HF + hd * F + f];
        dpre[e] = d;
    }
    __syncthreads();
    for (int e = tid; e < Hn; e += GAT_T) {         // e = hd * n + i
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc += att[e * n + j] * dpre[e * n + j];
        dot[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < H * nn; e += GAT_T) {
        const int hd = dnn(e), ij = e - hd * nn, i = dn(ij), j = ij - i * n;
        const float v = s1[hd * n + i] + s2[hd * n + j];
        dpre[e] = att[e] * (dpre[e] - dot[hd * n + i]) * (v > 0.f ? 1.f : alpha);
    }
    __syncthreads();
    for (int e = tid; e < 2 * Hn; e += GAT_T) {
        const int hi = e >> 1, hd = dn(hi), i = hi - hd * n;
        const float *dp = dpre + hd * nn;
        float acc = 0.f;
        if (e & 1) { for (int k = 0; k < n; ++k) acc += dp[k * n + i]; ds2[hi] = acc; }
        else       { for (int k = 0; k < n; ++k) acc += dp[i * n + k]; ds1[hi] = acc; }
    }
    __syncthreads();
    for (int e = tid; e < n * HF; e += GAT_T) {
        const int j = dHF(e), c = e - j * HF, hd = dF(c), f = c - hd * F;
        const float *ah = as + hd * 2 * F;
        float acc = ds1[hd * n + j] * ah[f] + ds2[hd * n + j] * ah[F + f];
        const float *ap = att + hd * nn;
        for (int i = 0; i < n; ++i) acc += ap[i * n + j] * dO[i * HF + c];
        dWh[(int64_t)(b + j) * ldg + c] = acc;
    }
    for (int e = tid; e < H * 2 * F; e += GAT_T) {
        const int hd = d2F(e), f = e - hd * 2 * F;
        const float *ds = (f < F ? ds1 : ds2) + hd * n;
        const int ff = f < F ? f : f - F;
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += ds[i] * wh[i * HF + hd * F + ff];
        da_part[((int64_t)s * H + hd) * 2 * F + f] = acc;
    }
}

// GCN aggregation (gcn_layers.py:20-26): Y_i = sum_j adj_ij X_j (+ bias);  transpose: Y_j = sum_i adj_ij X_i (backward)
__global__ void adj_matmul_kernel(Scenes sc, const int32_t *__restrict__ row_scene, int64_t Nrows, int F, const float *__restrict__ adj,
                                  const float *__restrict__ X, int64_t ldx, const float *__restrict__ bias, int transpose,
                                  float *__restrict__ Y, int64_t ldy) {
    const int64_t idx = gid();
    if (idx >= Nrows * F) return;
    const int64_t r = idx / F;
    const int f = (int)(idx - r * F);
    const int s = row_scene[r];
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b, i = (int)(r - b);
    const float *ablk = adj + sc.off[s];
    const int64_t rld = sc.ld ? sc.ld : n;
    float acc = bias ? bias[f] : 0.f;
    for (int j = 0; j < n; ++j) acc += (transpose ? ablk[j * rld + i] : ablk[i * rld + j]) * X[(int64_t)(b + j) * ldx + f];
    Y[r * ldy + f] = acc;
}

// disposition d_t (models/dagnet/model.py:141-147): row r gets the flattened xy of its scene's agents, zero padded to D
__global__ void disposition_kernel(Scenes sc, const int32_t *__restrict__ row_scene, int64_t Nrows, int D, const float *__restrict__ pos,
                                   float *__restrict__ out, int64_t ldo) {
    const int64_t idx = gid();
    if (idx >= Nrows * D) return;
    const int64_t r = idx / D;
    const int c = (int)(idx - r * D);
    const int s = row_scene[r];
    const int b = sc.scene_ptr[s], n = sc.scene_ptr[s + 1] - b;
    const int j = c >> 1;
    out[r * ldo + c] = j < n ? pos[(int64_t)(b + j) * 2 + (c & 1)] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// optimizer: clip_grad_norm_ + Adam (torch.optim.Adam defaults), no host synchronisation
// ---------------------------------------------------------------------------------------------
// acc[0] += sum g^2 (fp64); one workgroup per call keeps the order fixed
__global__ __launch_bounds__(1024) void sumsq_kernel(int64_t n, const float *__restrict__ g, double *__restrict__ acc) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[0] += red[0];
}

// coef = min(1, clip / (sqrt(sumsq) + 1e-6)) (torch.nn.utils.clip_grad_norm_); the gradient is scaled in place
__global__ void adam_kernel(int64_t n, float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            const double *__restrict__ sumsq, float clip, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const int64_t i = gid();
    if (i >= n) return;
    float coef = 1.f;
    if (clip > 0.f) coef = fminf(1.0f, clip / ((float)sqrt(sumsq[0]) + 1e-6f));
    const float gi = g[i] * coef;
    g[i] = gi;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * mi / denom;
}

}   // namespace tops

// ---------------------------------------------------------------------------------------------
// launchers (called from api.hip, which validates arguments and owns the C ABI)
// ---------------------------------------------------------------------------------------------
#define T_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)

extern "C" hipError_t dn_t_act_bwd(int64_t M, int N, const float *dY, int64_t ldd, const float *Y, int64_t ldy, float *dP, int64_t ldp,
                                   int act, float slope, hipStream_t st) {
    if (M * N == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::act_bwd_kernel, T_GRID(M * N), 0, st, M, N, dY, ldd, Y, ldy, dP, ldp, act, slope);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_copy2d(int64_t M, int N, const float *src, int64_t lds, float *dst, int64_t ldd, int accumulate, hipStream_t st) {
    if (M * N == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::copy2d_kernel, T_GRID(M * N), 0, st, M, N, src, lds, dst, ldd, accumulate);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_softmax_fwd(int64_t M, int N, const float *X, int64_t ldx, float *Y, int64_t ldy, hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int n4 = (N + 3) & ~3;
    if (N <= 128 && !(ldx & 3) && !(ldy & 3) && ldx >= n4 && ldy >= n4 && !((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15))
        hipLaunchKernelGGL(tops::softmax_fwd4_kernel, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, st, M, N, X, ldx, Y, ldy);
    else
        hipLaunchKernelGGL(tops::softmax_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, M, N, X, ldx, Y, ldy);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_softmax_bwd(int64_t M, int N, const float *Y, int64_t ldy, const float *dY, int64_t ldd, float *dX, int64_t ldx,
                                       hipStream_t st) {
    if (M == 0) return hipSuccess;
    const int n4 = (N + 3) & ~3;
    if (N <= 128 && !(ldy & 3) && !(ldd & 3) && !(ldx & 3) && ldy >= n4 && ldd >= n4 && ldx >= n4 &&
        !((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(dX)) & 15))
        hipLaunchKernelGGL(tops::softmax_bwd4_kernel, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, st, M, N, Y, ldy, dY, ldd, dX, ldx);
    else
        hipLaunchKernelGGL(tops::softmax_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, M, N, Y, ldy, dY, ldd, dX, ldx);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_gru_fwd(int64_t M, int R, const float *gi, const float *gh, const float *h, float *hn, float *gates,
                                   hipStream_t st) {
    if (M * R == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::gru_fwd_kernel, T_GRID(M * R), 0, st, M, R, gi, gh, h, hn, gates);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_gru_bwd(int64_t M, int R, const float *dhn, const float *gates, const float *gh, const float *h, float *dgi,
                                   float *dgh, float *dh, hipStream_t st) {
    if (M * R == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::gru_bwd_kernel, T_GRID(M * R), 0, st, M, R, dhn, gates, gh, h, dgi, dgh, dh);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_vae_fwd(int64_t M, int Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                                   const float *eps, float *z, float *kld_row, hipStream_t st) {
    if (M == 0) return hipSuccess;
    if (Z > 64) return hipErrorInvalidValue;
    int ZP = 1;
    while (ZP < Z) ZP <<= 1;
    const int rows_per_block = 256 / ZP;
    hipLaunchKernelGGL(tops::vae_fwd_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, M, Z, ZP, mu_e,
                       lv_e, mu_p, lv_p, eps, z, kld_row);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_vae_bwd(int64_t M, int Z, const float *mu_e, const float *lv_e, const float *mu_p, const float *lv_p,
                                   const float *eps, const float *dz, const float *g_kld, float scale, float *dmu_e, float *dlv_e,
                                   float *dmu_p, float *dlv_p, hipStream_t st) {
    if (M * Z == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::vae_bwd_kernel, T_GRID(M * Z), 0, st, M, Z, mu_e, lv_e, mu_p, lv_p, eps, dz, g_kld, scale, dmu_e, dlv_e, dmu_p, dlv_p);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_nll_fwd(int64_t M, int X, const float *mu, const float *lv, const float *x, float *nll_row, hipStream_t st) {
    if (M == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::nll_fwd_kernel, T_GRID(M), 0, st, M, X, mu, lv, x, nll_row);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_nll_bwd(int64_t M, int X, const float *mu, const float *lv, const float *x, const float *g, float scale,
                                   float *dmu, float *dlv, hipStream_t st) {
    if (M * X == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::nll_bwd_kernel, T_GRID(M * X), 0, st, M, X, mu, lv, x, g, scale, dmu, dlv);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_ce_fwd(int64_t M, const float *p, int64_t ldp, const int32_t *goal, float *ce_row, hipStream_t st) {
    if (M == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::ce_fwd_kernel, T_GRID(M), 0, st, M, p, ldp, goal, ce_row);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_softmax_ce_bwd(int64_t M, int G, const float *p, int64_t ldp, const int32_t *goal, const float *g, float *dx,
                                          int64_t ldx, hipStream_t st) {
    if (M * G == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::softmax_ce_bwd_kernel, T_GRID(M * G), 0, st, M, G, p, ldp, goal, g, dx, ldx);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_ce_bwd(int64_t M, int G, const int32_t *goal, const float *g, float *dp, int64_t ldp, hipStream_t st) {
    if (M * G == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::ce_bwd_kernel, T_GRID(M * G), 0, st, M, G, goal, g, dp, ldp);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_sum(int64_t n, const float *x, float scale, float *out, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(tops::sum_kernel, dim3(1), dim3(1024), 0, st, n, x, scale, out, accumulate);
    return hipGetLastError();
}

static int col_splits(int64_t M) {
    int64_t s = (M + 255) / 256;          // 256 rows per workgroup: enough workgroups to fill the chip on narrow matrices
    return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}
// floats of workspace the column reductions need for a matrix with N columns
extern "C" int64_t dn_t_col_workspace(int64_t M, int N) { return (int64_t)col_splits(M) * 2 * N; }

template <int MODE>
static hipError_t col_reduce(int64_t M, int N, const float *X, int64_t ldx, const float *stat, const float *dOut, int64_t ldd,
                             const float *Out, int64_t ldo, int tanh_out, float *part, float scale, float *out0, float *out1,
                             int accumulate, hipStream_t st, float *extra0 = nullptr, float *extra1 = nullptr, int extra_accumulate = 0) {
    const int splits = col_splits(M);
    const int rows = (int)((M + splits - 1) / splits);
    // 16-byte rows: aligned base, row stride a multiple of 4 that covers the last quad (N itself may be 90)
    const int64_t n4 = (N + 3) & ~3;
    auto al = [n4](const float *q, int64_t ld) { return !q || (((reinterpret_cast<uintptr_t>(q) & 15) == 0) && (ld & 3) == 0 && ld >= n4); };
    if (al(X, ldx) && (MODE < 2 || (al(dOut, ldd) && (!tanh_out || al(Out, ldo)))))
        hipLaunchKernelGGL((tops::col_reduce4_kernel<MODE>), dim3((N + 63) / 64, splits), dim3(256), 0, st, M, N, rows, X, ldx, stat, dOut,
                           ldd, Out, ldo, tanh_out, part);
    else
        hipLaunchKernelGGL((tops::col_reduce_kernel<MODE>), dim3((N + 63) / 64, splits), dim3(256), 0, st, M, N, rows, X, ldx, stat, dOut,
                           ldd, Out, ldo, tanh_out, part);
    hipLaunchKernelGGL(tops::col_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, splits, part, scale, out0, out1, accumulate, extra0,
                       extra1, extra_accumulate);
    return hipGetLastError();
}

// out[c] (+)= sum_m X[m][c]
extern "C" hipError_t dn_t_colsum(int64_t M, int N, const float *X, int64_t ldx, float *out, int accumulate, float *part, hipStream_t st) {
    if (N == 0) return hipSuccess;
    return col_reduce<0>(M, N, X, ldx, nullptr, nullptr, 0, nullptr, 0, 0, part, 1.0f, out, nullptr, accumulate, st);
}

// ---------------------------------------------------------------------------------------------
// dagnet_op_colsum_static: dn_t_colsum with the split partial sums in a file-scope device array instead of a caller
// workspace, for callers that cannot allocate (HIP graph capture, tight serving loops).  Every GPU holds its own copy of
// the array until the process ends; the host-side name `scratch` is only a handle for hipGetSymbolAddress, the kernels get
// the per-device address.  Calls that overlap on one GPU (two streams) race on the array.
// ---------------------------------------------------------------------------------------------
#define DN_STATIC_COLS 1024                                   // widest matrix the static variant accepts
#define DN_STATIC_FLOATS (512 * 2 * DN_STATIC_COLS)           // col_splits() <= 512 splits x 2 quantities x columns = 4 MiB
static __device__ __align__(256) float scratch[DN_STATIC_FLOATS];

#define DN_MAX_DEVICES 64
static float *g_scratch_ptr[DN_MAX_DEVICES];                  // per device: address of that GPU's copy of `scratch`
static std::mutex g_scratch_mu;

extern "C" int dn_t_colsum_static_max_cols(void) { return DN_STATIC_COLS; }

// -> hipErrorInvalidValue for N > DN_STATIC_COLS (the caller reports the limit); no allocation, no host synchronisation
extern "C" hipError_t dn_t_colsum_static(int64_t M, int N, const float *X, int64_t ldx, float *out, int accumulate, hipStream_t st) {
    if (N == 0) return hipSuccess;
    if (N > DN_STATIC_COLS || (int64_t)col_splits(M) * 2 * N > DN_STATIC_FLOATS) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= DN_MAX_DEVICES) return hipErrorInvalidDevice;
    float *part;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[dev]) {                            // first call on this device, with this device current
            void *p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess) return e;
            g_scratch_ptr[dev] = static_cast<float *>(p);
        }
        part = g_scratch_ptr[dev];
    }
    // every partial the second stage reads is written by the first stage of the same call: nothing to clear between calls
    return col_reduce<0>(M, N, X, ldx, nullptr, nullptr, 0, nullptr, 0, 0, part, 1.0f, out, nullptr, accumulate, st);
}

// BatchNorm1d forward in train mode (+ tanh).  stat [2N] receives mean | invstd (saved for the backward);
// tmp [2N] scratch; running statistics are updated in place when given.
extern "C" hipError_t dn_t_bn_fwd(int64_t M, int N, const float *X, int64_t ldx, const float *gamma, const float *beta, float eps,
                                  float momentum, int tanh_out, float *Y, int64_t ldy, float *stat, float *run_mean, float *run_var,
                                  float *tmp, float *part, hipStream_t st) {
    if (M == 0 || N == 0) return hipSuccess;
    if (((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx & 3) == 0 && ldx >= ((N + 3) & ~3)) {      // 16-byte rows: one pass
        const int splits = col_splits(M);
        const int rows = (int)((M + splits - 1) / splits);
        hipLaunchKernelGGL(tops::bn_moments_kernel, dim3((N + 63) / 64, splits), dim3(256), 0, st, M, N, rows, X, ldx, part);
        hipLaunchKernelGGL(tops::bn_finish_stat_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, M, splits, rows, part, eps, momentum, stat,
                           run_mean, run_var);
        hipLaunchKernelGGL(tops::bn_apply_kernel, T_GRID(M * N), 0, st, M, N, X, ldx, stat, gamma, beta, tanh_out, Y, ldy);
        return hipGetLastError();
    }
    hipError_t e = col_reduce<0>(M, N, X, ldx, nullptr, nullptr, 0, nullptr, 0, 0, part, 1.0f / (float)M, tmp, nullptr, 0, st);
    if (e != hipSuccess) return e;
    e = col_reduce<1>(M, N, X, ldx, tmp, nullptr, 0, nullptr, 0, 0, part, 1.0f, tmp + N, nullptr, 0, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tops::bn_stat_kernel, T_GRID(N), 0, st, N, M, tmp, tmp + N, eps, momentum, stat, run_mean, run_var);
    hipLaunchKernelGGL(tops::bn_apply_kernel, T_GRID(M * N), 0, st, M, N, X, ldx, stat, gamma, beta, tanh_out, Y, ldy);
    return hipGetLastError();
}

// -> dX, dgamma (+)=, dbeta (+)=.  sums [2N] scratch.
extern "C" hipError_t dn_t_bn_bwd(int64_t M, int N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *dOut,
                                  int64_t ldd, const float *Out, int64_t ldo, int tanh_out, float *dX, int64_t ldg, float *dgamma,
                                  float *dbeta, int accumulate, float *sums, float *part, hipStream_t st) {
    if (M == 0 || N == 0) return hipSuccess;
    hipError_t e = col_reduce<2>(M, N, X, ldx, stat, dOut, ldd, Out, ldo, tanh_out, part, 1.0f, sums, sums + N, 0, st, dbeta, dgamma, accumulate);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tops::bn_bwd_kernel, T_GRID(M * N), 0, st, M, N, X, ldx, stat, gamma, dOut, ldd, Out, ldo, tanh_out, sums, dX, ldg, M);
    return hipGetLastError();
}

// ---- synchronised BatchNorm in stages (the collectives between them belong to the caller) -------------------------------------
// stage 1: this shard's mom = [count | mean | M2] per column
extern "C" hipError_t dn_t_bn_moments(int64_t M, int N, const float *X, int64_t ldx, float *mom, float *tmp, float *part, hipStream_t st) {
    if (N == 0) return hipSuccess;
    if (M == 0) return hipMemsetAsync(mom, 0, sizeof(float) * 3 * N, st);          // an empty shard: count 0
    if (((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx & 3) == 0 && ldx >= ((N + 3) & ~3)) {
        const int splits = col_splits(M);
        const int rows = (int)((M + splits - 1) / splits);
        hipLaunchKernelGGL(tops::bn_moments_kernel, dim3((N + 63) / 64, splits), dim3(256), 0, st, M, N, rows, X, ldx, part);
        hipLaunchKernelGGL(tops::bn_moments_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, M, splits, rows, part, mom);
        return hipGetLastError();
    }
    hipError_t e = col_reduce<0>(M, N, X, ldx, nullptr, nullptr, 0, nullptr, 0, 0, part, 1.0f / (float)M, tmp, nullptr, 0, st);
    if (e != hipSuccess) return e;
    e = col_reduce<1>(M, N, X, ldx, tmp, nullptr, 0, nullptr, 0, 0, part, 1.0f, tmp + N, nullptr, 0, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tops::bn_moments_pack_kernel, T_GRID(N), 0, st, N, M, tmp, tmp + N, mom);
    return hipGetLastError();
}
// stage 2: merge the gathered triples of all shards -> stat, running statistics, total row count
extern "C" hipError_t dn_t_bn_merge(int N, int n_parts, const float *parts, float eps, float momentum, float *stat, float *run_mean,
                                    float *run_var, float *total, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::bn_merge_kernel, T_GRID(N), 0, st, N, n_parts, parts, eps, momentum, stat, run_mean, run_var, total);
    return hipGetLastError();
}
// stage 3: y = (x - mean) invstd gamma + beta (+ tanh) with the merged statistics
extern "C" hipError_t dn_t_bn_apply(int64_t M, int N, const float *X, int64_t ldx, const float *stat, const float *gamma, const float *beta,
                                    int tanh_out, float *Y, int64_t ldy, hipStream_t st) {
    if (M == 0 || N == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::bn_apply_kernel, T_GRID(M * N), 0, st, M, N, X, ldx, stat, gamma, beta, tanh_out, Y, ldy);
    return hipGetLastError();
}
// backward stage 1: this shard's sums = [sum g | sum g xhat] (g = dOut (1 - Out^2) with tanh): its share of dbeta | dgamma
extern "C" hipError_t dn_t_bn_bwd_sums(int64_t M, int N, const float *X, int64_t ldx, const float *stat, const float *dOut, int64_t ldd,
                                       const float *Out, int64_t ldo, int tanh_out, float *sums, float *part, hipStream_t st) {
    if (N == 0) return hipSuccess;
    if (M == 0) return hipMemsetAsync(sums, 0, sizeof(float) * 2 * N, st);
    return col_reduce<2>(M, N, X, ldx, stat, dOut, ldd, Out, ldo, tanh_out, part, 1.0f, sums, sums + N, 0, st);
}
// backward stage 2: dX from the sums over ALL shards' rows (Mtot of them)
extern "C" hipError_t dn_t_bn_bwd_apply(int64_t M, int64_t Mtot, int N, const float *X, int64_t ldx, const float *stat, const float *gamma,
                                        const float *dOut, int64_t ldd, const float *Out, int64_t ldo, int tanh_out, const float *sums,
                                        float *dX, int64_t ldg, hipStream_t st) {
    if (M == 0 || N == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::bn_bwd_kernel, T_GRID(M * N), 0, st, M, N, X, ldx, stat, gamma, dOut, ldd, Out, ldo, tanh_out, sums, dX, ldg, Mtot);
    return hipGetLastError();
}

extern "C" int64_t dn_t_gat_lds_bytes(int n_max, int F, int backward) {
    const int64_t n = n_max, FP = (F + 3) & ~3;          // rows of Wh / dO are padded to float4s in LDS
    return sizeof(float) * ((backward ? 2 * n * FP + 5 * n + 2 * n * n : n * FP + 2 * n + n * n) + 2 * FP);     // + the head's attention vector
}
extern "C" hipError_t dn_t_gat_attn_fwd(const int32_t *scene_ptr, const int64_t *off_compact, int S, int n_max, int64_t adj_ld, const float *adj,
                                        const float *Wh, int64_t ldw, const float *a, int H, int F, float alpha, int elu, float *Out,
                                        int64_t ldo, hipStream_t st) {
    if (S == 0) return hipSuccess;
    const int64_t lds = dn_t_gat_lds_bytes(n_max, F, 0);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    tops::Scenes sc{scene_ptr, off_compact, S, adj_ld};
    if (H > 1 && H * lds <= 40 * 1024) {        // all heads of a scene in one workgroup (at least 4 workgroups per CU)
        hipLaunchKernelGGL(tops::gat_attn_fwd_heads_kernel, dim3(S), dim3(GAT_T), (size_t)(H * lds), st, sc, adj, Wh, ldw, a, H, F, alpha, elu,
                           Out, ldo);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(tops::gat_attn_fwd_kernel, dim3(S, H), dim3(GAT_T), (size_t)lds, st, sc, adj, Wh, ldw, a, F, alpha, elu, Out, ldo);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_gat_attn_bwd(const int32_t *scene_ptr, const int64_t *off_compact, int S, int n_max, int64_t adj_ld, const float *adj,
                                        const float *Wh, int64_t ldw, const float *a, int H, int F, float alpha, int elu, const float *Out,
                                        int64_t ldo, const float *dOut, int64_t ldd, float *dWh, int64_t ldg, float *da_part, hipStream_t st) {
    if (S == 0) return hipSuccess;
    const int64_t lds = dn_t_gat_lds_bytes(n_max, F, 1);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    tops::Scenes sc{scene_ptr, off_compact, S, adj_ld};
    if (H > 1 && H * lds <= 40 * 1024) {
        hipLaunchKernelGGL(tops::gat_attn_bwd_heads_kernel, dim3(S), dim3(GAT_T), (size_t)(H * lds), st, sc, adj, Wh, ldw, a, H, F, alpha, elu,
                           Out, ldo, dOut, ldd, dWh, ldg, da_part);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(tops::gat_attn_bwd_kernel, dim3(S, H), dim3(GAT_T), (size_t)lds, st, sc, adj, Wh, ldw, a, F, alpha, elu, Out, ldo, dOut,
                       ldd, dWh, ldg, da_part);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_adj_matmul(const int32_t *scene_ptr, const int64_t *off_compact, const int32_t *row_scene, int S, int64_t Nrows,
                                      int F, int64_t adj_ld, const float *adj, const float *X, int64_t ldx, const float *bias, int transpose, float *Y,
                                      int64_t ldy, hipStream_t st) {
    if (Nrows * F == 0) return hipSuccess;
    tops::Scenes sc{scene_ptr, off_compact, S, adj_ld};
    hipLaunchKernelGGL(tops::adj_matmul_kernel, T_GRID(Nrows * F), 0, st, sc, row_scene, Nrows, F, adj, X, ldx, bias, transpose, Y, ldy);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_disposition(const int32_t *scene_ptr, const int64_t *off_compact, const int32_t *row_scene, int S, int64_t Nrows,
                                       int D, const float *pos, float *out, int64_t ldo, hipStream_t st) {
    if (Nrows * D == 0) return hipSuccess;
    tops::Scenes sc{scene_ptr, off_compact, S, 0};
    hipLaunchKernelGGL(tops::disposition_kernel, T_GRID(Nrows * D), 0, st, sc, row_scene, Nrows, D, pos, out, ldo);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_sumsq(int64_t n, const float *g, double *acc, hipStream_t st) {
    hipLaunchKernelGGL(tops::sumsq_kernel, dim3(1), dim3(1024), 0, st, n, g, acc);
    return hipGetLastError();
}
extern "C" hipError_t dn_t_adam(int64_t n, float *p, float *g, float *m, float *v, const double *sumsq, float clip, float lr, float b1,
                                float b2, float eps, float bc1, float bc2, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tops::adam_kernel, T_GRID(n), 0, st, n, p, g, m, v, sumsq, clip, lr, b1, b2, eps, bc1, bc2);
    return hipGetLastError();
}
