// micro-benchmark: fp64 FMA issue rate and table-exp throughput on gfx950 (tuning aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__constant__ double TAB[64];

__device__ __forceinline__ double exp_tab(double t, const double* tab) {
    const double MAGIC = 6755399441055744.0;
    const double u = fma(t, 9.23324826168936567683e+01, MAGIC);
    const int nn = __double2loint(u);
    const double kd = u - MAGIC;
    double r = fma(kd, -1.08304246932675596327e-02, t);
    r = fma(kd, -2.98158582698529328128e-12, r);
    double p = fma(r, 8.3333333333333333333e-3, 4.1666666666666666667e-2);
    p = fma(p, r, 1.6666666666666666667e-1);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(tab[nn & 63] * p, nn >> 6);
}

template <int CH>
__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double a, double b) {
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, bool LDS_ABK>
__global__ void __launch_bounds__(256) exp_kernel(double* out, int iters, double a, double b) {
    __shared__ double tab[64];
    __shared__ double abk[64 * 4];
    if (threadIdx.x < 64) { tab[threadIdx.x] = TAB[threadIdx.x]; }
    abk[threadIdx.x] = 1e-3 * threadIdx.x;
    __syncthreads();
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0;
    const double X1 = threadIdx.x * 1e-2, X2 = X1 * X1;
    for (int i = 0; i < iters; i += CH) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            double t;
            if (LDS_ABK) { const int j = (i + c) & 63; t = fma(abk[4 * j + 2], X2, fma(abk[4 * j + 1], X1, abk[4 * j])); }
            else t = fma(a, X1, b * (i + c));
            acc[c] += exp_tab(t - 3.0, tab);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    std::vector<double> tab(64);
    for (int j = 0; j < 64; ++j) tab[j] = std::exp2(j / 64.0);
    hipMemcpyToSymbol(HIP_SYMBOL(TAB), tab.data(), 64 * 8);
    double* out; hipMalloc(&out, 8ull * 256 * 256 * 32);
    const int iters = 1 << 16;
    for (int bpc : {1, 2, 4, 8}) {
        const int grid = 256 * bpc;       // bpc workgroups (4 waves each) per CU -> bpc waves per SIMD
        double ms = time_ms([&] { fma_kernel<4><<<grid, 256>>>(out, iters, 0.999, 1e-3); });
        double inst = (double)grid * 4 * iters * 4;            // wave-instructions
        printf("fma x4 chains, %d waves/SIMD: %.3f ms  -> %.2f cycles/wave-inst/SIMD @2.4GHz, %.1f TFLOP/s\n", bpc, ms,
               ms * 1e-3 * 2.4e9 * 1024 / inst, inst * 64 * 2 / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { exp_kernel<2, false><<<grid, 256>>>(out, iters, 0.5, 1e-4); });
        double terms = (double)grid * 4 * iters;               // wave-terms
        printf("exp x2 chains (no abk), %d waves/SIMD: %.3f ms -> %.1f cycles/wave-term/SIMD\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { exp_kernel<2, true><<<grid, 256>>>(out, iters, 0.5, 1e-4); });
        printf("exp x2 chains + LDS abk, %d waves/SIMD: %.3f ms -> %.1f cycles/wave-term/SIMD\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { exp_kernel<4, true><<<grid, 256>>>(out, iters, 0.5, 1e-4); });
        printf("exp x4 chains + LDS abk, %d waves/SIMD: %.3f ms -> %.1f cycles/wave-term/SIMD\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
    }
    return 0;
}
