// (round 3: + MODE 4 = fp16 two terms, three products)
// What would the rollout's dense layers cost, and how exact would they be, on split-bf16 products instead of the fp32 MFMA?
// (the fp32 v_mfma_f32_32x32x2_f32 takes 64 clocks and blocks vector issue; the bf16 32x32x16 form takes 32.)
// One wave = one 32-row tile running a chain of L layers x <- relu(W x), W 64 x 64 in LDS, activations register-resident in
// the accumulator layout (lane = row, register 4a+i of tile nt = feature 32nt + 8a + 4lh + i) which is also the next product's
// operand layout in all three modes:
//   MODE 0: 64 v_mfma_f32_32x32x2_f32 per layer (what mm<> in v2_common.h does).
//   MODE 3: x and W split into three bf16 terms, six products (hh, hm, mh, hl, lh, mm): 48 v_mfma_f32_32x32x16_bf16.
//   MODE 2: two terms, three products (hh, hl, lh): 24 of them.
// Prints clocks per layer per wave at 1 and 2 waves per SIMD and the error of an 8-layer chain against fp64.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/split_bf16 tools/ubench/split_bf16.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// LDS images.  fp32: [mt 2][q 8][lane 64] float4, q = 4 nt_in + a, element i = W[32mt + m][32nt_in + 8a + 4lh + i].
// bf16: [term 3][mt 2][kb 4][lane 64] 8 x bf16, kb = 2 nt_in + p, slot j = W[32mt + m][32nt_in + 8(2p + j/4) + 4lh + j%4].
constexpr int F32_IMG = 2 * 8 * 64 * 4;       // floats
constexpr int BF_IMG = 3 * 2 * 4 * 64 * 4;    // dwords

template <int MODE, int SPLIT>
__global__ __launch_bounds__(512) void chain(const float *__restrict__ img, const float *__restrict__ X, float *__restrict__ Y,
                                             int layers, unsigned long long *ticks) {
    extern __shared__ float lds[];
    constexpr int IMG = MODE == 0 ? F32_IMG : BF_IMG;      // (MODE 4 uses two of the three term planes)
    for (int i = threadIdx.x; i < IMG; i += blockDim.x) lds[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int row = lane & 31, lh = lane >> 5;
    const float *xr = X + ((size_t)wave * 32 + row) * 64;
    f32x16 x[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) x[nt][4 * a + i] = xr[32 * nt + 8 * a + 4 * lh + i];
    bf16x8 xs[3][4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        f32x16 acc[2] = {{0}, {0}};
        if constexpr (MODE == 0) {
            const float4 *w = reinterpret_cast<const float4 *>(lds);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 w0 = w[(0 * 8 + q) * 64 + lane], w1 = w[(1 * 8 + q) * 64 + lane];
                const float wa[4] = {w0.x, w0.y, w0.z, w0.w}, wb[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float b = x[q >> 2][4 * (q & 3) + i];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i], b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i], b, acc[1], 0, 0, 0);
                }
            }
        } else if constexpr (MODE == 4) {
            // fp16, two terms (11 + 11 significant bits, round to nearest), three products (hh, hl, lh)
            const uint4 *w = reinterpret_cast<const uint4 *>(lds);
            uint4 xh[4], xl[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                unsigned ph[4], pl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f2 v = {x[kb >> 1][8 * (kb & 1) + 2 * j], x[kb >> 1][8 * (kb & 1) + 2 * j + 1]};
                    const h2 hi = __builtin_convertvector(v, h2);
                    const f2 r = v - __builtin_convertvector(hi, f2);
                    const h2 lo = __builtin_convertvector(r, h2);
                    ph[j] = __builtin_bit_cast(unsigned, hi);
                    pl[j] = __builtin_bit_cast(unsigned, lo);
                }
                xh[kb] = uint4{ph[0], ph[1], ph[2], ph[3]};
                xl[kb] = uint4{pl[0], pl[1], pl[2], pl[3]};
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const f16x8 w0h = __builtin_bit_cast(f16x8, w[((0 * 2 + 0) * 4 + kb) * 64 + lane]), w1h = __builtin_bit_cast(f16x8, w[((0 * 2 + 1) * 4 + kb) * 64 + lane]);
                const f16x8 w0l = __builtin_bit_cast(f16x8, w[((1 * 2 + 0) * 4 + kb) * 64 + lane]), w1l = __builtin_bit_cast(f16x8, w[((1 * 2 + 1) * 4 + kb) * 64 + lane]);
                const f16x8 bh = __builtin_bit_cast(f16x8, xh[kb]), bl = __builtin_bit_cast(f16x8, xl[kb]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, bh, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, bl, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, bl, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, bh, acc[1], 0, 0, 0);
            }
        } else {
            constexpr int NT = MODE;                 // terms kept of x and of W
            const uint4 *w = reinterpret_cast<const uint4 *>(lds);
            // SPLIT false (timing only): the operands of the first layer are reused, the conversions drop out
            if (SPLIT || l == 0)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if constexpr (SPLIT == 2) {
                    // truncating split from full-rate integer / fp32 instructions only: and, sub, perm (8 + 8 + 8 bits, exact)
                    unsigned pk[NT][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v0 = x[kb >> 1][8 * (kb & 1) + 2 * j], v1 = x[kb >> 1][8 * (kb & 1) + 2 * j + 1];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const unsigned u0 = __builtin_bit_cast(unsigned, v0), u1 = __builtin_bit_cast(unsigned, v1);
                            pk[t][j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
                            if (t + 1 < NT) {
                                v0 -= __builtin_bit_cast(float, u0 & 0xffff0000u);
                                v1 -= __builtin_bit_cast(float, u1 & 0xffff0000u);
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const uint4 q = {pk[t][0], pk[t][1], pk[t][2], pk[t][3]};
                        xs[t][kb] = __builtin_bit_cast(bf16x8, q);
                    }
                } else {
                    f32x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = x[kb >> 1][8 * (kb & 1) + j];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        xs[t][kb] = __builtin_convertvector(v, bf16x8);
                        v -= __builtin_convertvector(xs[t][kb], f32x8);
                    }
                }
            }
            // smallest products first: term index sum s = tw + tx descending, only s <= NT - 1 kept
#pragma unroll
            for (int s = NT - 1; s >= 0; --s)
#pragma unroll
                for (int tw = 0; tw <= s; ++tw) {
                    const int tx = s - tw;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const bf16x8 w0 = __builtin_bit_cast(bf16x8, w[((tw * 2 + 0) * 4 + kb) * 64 + lane]);
                        const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[((tw * 2 + 1) * 4 + kb) * 64 + lane]);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xs[tx][kb], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xs[tx][kb], acc[1], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[nt][r] = fmaxf(acc[nt][r], 0.f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float *yr = Y + ((size_t)wave * 32 + row) * 64;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) yr[32 * nt + 8 * a + 4 * lh + i] = x[nt][4 * a + i];
    // the oldest wave of a SIMD is served first, so one wave's span flatters: report first start to last end of the block
    if (lane == 0 && blockIdx.x == 0) {
        ticks[2 * (threadIdx.x >> 6)] = t0;
        ticks[2 * (threadIdx.x >> 6) + 1] = t1;
    }
}

static unsigned short bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int MODE, int SPLIT = 1>
void run(const char *name, const std::vector<float> &W, const std::vector<float> &X, const std::vector<double> &ref, int rows_ref) {
    std::vector<float> img(MODE == 0 ? F32_IMG : BF_IMG);
    if (MODE == 0) {
        for (int mt = 0; mt < 2; ++mt)
            for (int q = 0; q < 8; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i)
                        img[((mt * 8 + q) * 64 + lane) * 4 + i] =
                            W[(32 * mt + (lane & 31)) * 64 + 32 * (q >> 2) + 8 * (q & 3) + 4 * (lane >> 5) + i];
    } else if (MODE == 4) {
        _Float16 *h = reinterpret_cast<_Float16 *>(img.data());
        for (int mt = 0; mt < 2; ++mt)
            for (int kb = 0; kb < 4; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        float v = W[(32 * mt + (lane & 31)) * 64 + 32 * (kb >> 1) + 8 * (2 * (kb & 1) + j / 4) + 4 * (lane >> 5) + j % 4];
                        for (int t = 0; t < 2; ++t) {
                            const _Float16 b = (_Float16)v;
                            h[((((t * 2 + mt) * 4 + kb) * 64 + lane) * 8) + j] = b;
                            v -= (float)b;
                        }
                    }
    } else {
        unsigned short *h = reinterpret_cast<unsigned short *>(img.data());
        for (int mt = 0; mt < 2; ++mt)
            for (int kb = 0; kb < 4; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        float v = W[(32 * mt + (lane & 31)) * 64 + 32 * (kb >> 1) + 8 * (2 * (kb & 1) + j / 4) + 4 * (lane >> 5) + j % 4];
                        for (int t = 0; t < 3; ++t) {
                            const unsigned short b = bf16_rne(v);
                            h[((((t * 2 + mt) * 4 + kb) * 64 + lane) * 8) + j] = b;
                            v -= bf16_f(b);
                        }
                    }
    }
    float *dimg, *dX, *dY;
    unsigned long long *ticks;
    const int max_blocks = 512, nrows = max_blocks * 4 * 32;
    (void)hipMalloc(&dimg, img.size() * 4);
    (void)hipMalloc(&dX, (size_t)nrows * 64 * 4);
    (void)hipMalloc(&dY, (size_t)nrows * 64 * 4);
    (void)hipMalloc(&ticks, 16 * 8);
    (void)hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < nrows; off += rows_ref)
        (void)hipMemcpy(dX + (size_t)off * 64, X.data(), (size_t)rows_ref * 64 * 4, hipMemcpyHostToDevice);
    const size_t lds = img.size() * 4;
    // accuracy: 8 layers, first wave tiles against fp64
    hipLaunchKernelGGL((chain<MODE, SPLIT>), dim3(rows_ref / 128), dim3(256), lds, 0, dimg, dX, dY, 8, ticks);
    std::vector<float> Y((size_t)rows_ref * 64);
    (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, ymax = 0, esum = 0;
    for (size_t i = 0; i < Y.size(); ++i) {
        const double e = fabs((double)Y[i] - ref[i]);
        emax = fmax(emax, e);
        esum += e;
        ymax = fmax(ymax, fabs(ref[i]));
    }
    if (SPLIT) printf("%-26s 8-layer chain vs fp64: max |err| %.3e  mean %.3e  (max |y| %.3f)\n", name, emax, esum / Y.size(), ymax);
    const int layers = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        // one block per CU, 4 * wps waves in it: every SIMD holds exactly wps waves
        hipLaunchKernelGGL((chain<MODE, SPLIT>), dim3(256), dim3(256 * wps), lds, 0, dimg, dX, dY, layers, ticks);
        hipLaunchKernelGGL((chain<MODE, SPLIT>), dim3(256), dim3(256 * wps), lds, 0, dimg, dX, dY, layers, ticks);
        unsigned long long tt[16], t = 0, lo = ~0ull;
        (void)hipMemcpy(tt, ticks, sizeof(tt), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4 * wps; ++w) {
            lo = tt[2 * w] < lo ? tt[2 * w] : lo;
            t = tt[2 * w + 1] > t ? tt[2 * w + 1] : t;
        }
        t -= lo;
        printf("%-26s %d wave(s) per SIMD: %7.0f clocks per layer per wave, %7.0f per layer per SIMD-tile\n", name, wps,
               (double)t / layers, (double)t / layers / wps);
    }
    (void)hipFree(dimg); (void)hipFree(dX); (void)hipFree(dY); (void)hipFree(ticks);
}

int main() {
    const int rows = 512;
    std::vector<float> W(64 * 64), X((size_t)rows * 64);
    srand(7);
    auto gauss = []() {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
    };
    for (auto &w : W) w = (float)(gauss() * sqrt(2.0 / 64.0));
    for (auto &x : X) x = (float)gauss();
    std::vector<double> cur(X.begin(), X.end()), nxt(cur.size());
    for (int l = 0; l < 8; ++l) {
        for (int r = 0; r < rows; ++r)
            for (int o = 0; o < 64; ++o) {
                double s = 0;
                for (int k = 0; k < 64; ++k) s += (double)W[o * 64 + k] * cur[(size_t)r * 64 + k];
                nxt[(size_t)r * 64 + o] = s > 0 ? s : 0;
            }
        cur.swap(nxt);
    }
    run<0>("fp32 32x32x2", W, X, cur, rows);
    run<3>("bf16 x3 (six products)", W, X, cur, rows);
    run<2>("bf16 x2 (three products)", W, X, cur, rows);
    run<3, 2>("bf16 x3, truncating split", W, X, cur, rows);
    run<3, 0>("bf16 x3 without the split", W, X, cur, rows);
    run<4>("fp16 x2 (three products)", W, X, cur, rows);
    return 0;
}
