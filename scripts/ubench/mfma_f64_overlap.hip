// micro-benchmark: does v_mfma_f64_16x16x4_f64 overlap with fp64 VALU work on gfx950? (tuning aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = VALU only, 1 = MFMA only, 2 = both (independent streams)
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    double a = threadIdx.x * 1e-3 + 1.0, b = a + 0.5, c = a + 0.25, d = a + 0.125;
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double ma = a, mb = b;
    for (int i = 0; i < iters; ++i) {
        if (MODE != 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {       // 32 independent-ish fp64 FMAs
                a = fma(a, 0.999, 1e-3); b = fma(b, 0.999, 1e-3); c = fma(c, 0.999, 1e-3); d = fma(d, 0.999, 1e-3);
            }
        }
        if (MODE != 0) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[3];
}

template <typename K>
double run(K kern, double* out, int bpc) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096;
    kern<<<256 * bpc, 256>>>(out, 16); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<256 * bpc, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.0e9 / ((double)bpc * iters);   // cycles (2.0 GHz) per iteration per SIMD (bpc waves per SIMD)
}

int main() {
    double* out; hipMalloc(&out, 8ull * 256 * 8 * 256);
    for (int bpc : {1, 2, 4, 8}) {
        const double v = run(k<0>, out, bpc), m = run(k<1>, out, bpc), both = run(k<2>, out, bpc);
        printf("%d waves/SIMD: per iteration per wave-slot: 32 FMAs %.1f cyc | 2 MFMA f64 16x16x4 %.1f cyc | both %.1f cyc (sum %.1f, max %.1f)\n",
               bpc, v, m, both, v + m, v > m ? v : m);
    }
    return 0;
}
