// Three ways to split 16 fp32 registers into three truncated bf16 terms (v2_common.h split3): clocks per 16-value split and
// exactness (hi + mid + lo == value, bit for bit).
//   A: v_perm + 2 v_and + 2 v_sub per pair and term (the first form)
//   B: v_perm + 2 v_and + 1 v_pk_add_f32 (packed subtract)
//   C: v_perm + 2 v_dot2c_f32_bf16 (value -= hi as a dot product with {-1, 0} / {0, -1}: no v_and)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/split_forms tools/ubench/split_forms.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int FORM>
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned (&pk)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
        pk[t] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        if (t < 2) {
            if (FORM == 0) {
                v0 -= __uint_as_float(u0 & 0xffff0000u);
                v1 -= __uint_as_float(u1 & 0xffff0000u);
            } else if (FORM == 1) {
                f2 v = {v0, v1};
                f2 h = {__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)};
                v = v - h;
                v0 = v.x; v1 = v.y;
            } else {
                const unsigned m0 = 0x0000bf80u, m1 = 0xbf800000u;
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v0) : "v"(pk[t]), "v"(m0));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v1) : "v"(pk[t]), "v"(m1));
            }
        }
    }
}

template <int FORM>
__global__ void check(const float *x, unsigned *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned pk[3];
    split_pair<FORM>(x[2 * i], x[2 * i + 1], pk);
    for (int t = 0; t < 3; ++t) out[3 * i + t] = pk[t];
}

template <int FORM>
__global__ __launch_bounds__(256) void timing(const float *x, float *y, int iters, unsigned long long *ticks) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = x[threadIdx.x * 16 + r];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned pk[3];
            split_pair<FORM>(v[2 * j], v[2 * j + 1], pk);
            acc ^= pk[0] ^ pk[1] ^ pk[2];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(__float_as_uint(v[r]) ^ (acc & 0xff));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    y[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc);
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t0; ticks[1] = t1; }
}

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    srand(3);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        const int e = 127 + (rand() % 60) - 40;                 // exponents 2^-40 .. 2^19
        u = (u & 0x807fffffu) | ((unsigned)e << 23);
        if (i % 97 == 0) u = (u & 0x80000000u) | (rand() % 3 == 0 ? 0u : (1u << 23) | (u & 0xffff));     // zeros and tiny values
        memcpy(&x[i], &u, 4);
    }
    float *dx, *dy; unsigned *dout; unsigned long long *dt;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, 1 << 22); (void)hipMalloc(&dout, (size_t)n / 2 * 3 * 4); (void)hipMalloc(&dt, 16);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> o((size_t)n / 2 * 3);
    const char *names[3] = {"A perm + 2 and + 2 sub", "B perm + 2 and + pk_add", "C perm + 2 dot2c"};
    for (int form = 0; form < 3; ++form) {
        if (form == 0) hipLaunchKernelGGL(check<0>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
        if (form == 1) hipLaunchKernelGGL(check<1>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
        if (form == 2) hipLaunchKernelGGL(check<2>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
        (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n / 2; ++i)
            for (int h = 0; h < 2; ++h) {
                double s = 0;
                for (int t = 0; t < 3; ++t) {
                    unsigned w = h ? (o[3 * i + t] & 0xffff0000u) : (o[3 * i + t] << 16);
                    float f; memcpy(&f, &w, 4);
                    s += (double)f;
                }
                if ((float)s != x[2 * i + h] || s != (double)x[2 * i + h]) ++bad;
            }
        const int iters = 20000;
        unsigned long long tk[2];
        for (int rep = 0; rep < 2; ++rep) {
            if (form == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(256), 0, 0, dx, dy, iters, dt);
            if (form == 1) hipLaunchKernelGGL(timing<1>, dim3(256), dim3(256), 0, 0, dx, dy, iters, dt);
            if (form == 2) hipLaunchKernelGGL(timing<2>, dim3(256), dim3(256), 0, 0, dx, dy, iters, dt);
            (void)hipMemcpy(tk, dt, 16, hipMemcpyDeviceToHost);
        }
        printf("%-26s inexact values: %ld of %d   clocks per 16-value split (one wave per SIMD): %.1f\n", names[form], bad, n,
               (double)(tk[1] - tk[0]) / iters);
    }
    return 0;
}
