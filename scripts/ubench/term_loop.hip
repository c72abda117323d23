// micro-benchmark of the f2v term loop variants (tuning aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ double TABG[2048];
struct AB { double a, b; };

// VAR 0: 2048-entry table, cubic (current)   VAR 1: same but table index forced to lane-constant (no conflicts)
// VAR 2: 64-entry table replicated 32x (conflict-free), degree 5   VAR 3: no table at all (multiply by constant)
template <int VAR>
__device__ __forceinline__ double exp_v(double t, const double* tab, int lane) {
    const double MAGIC = 6755399441055744.0;
    if (VAR == 2) {
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
        return ldexp(tab[((nn & 63) << 5) | (lane & 31)] * p, nn >> 6);
    }
    const double u = fma(t, 2954.639443740597, MAGIC);
    const int nn = __double2loint(u);
    const double kd = u - MAGIC;
    double r = fma(kd, -0.0003384507717577858, t);
    r = fma(kd, -1.1323470770733885e-20, r);
    double p = fma(r, 1.6666666666666666667e-1, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double T;
    if (VAR == 0) T = tab[nn & 2047];
    else if (VAR == 1) T = tab[(lane * 33) & 2047];
    else T = 1.0000001;
    return ldexp(T * p, nn >> 11);
}

template <int VAR>
__global__ void __launch_bounds__(256) loop_kernel(double* out, int iters) {
    __shared__ double tab[2048];
    __shared__ AB ab[4][64];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = TABG[i];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ab[wid][lane].a = -1.0 - 1e-2 * lane; ab[wid][lane].b = 1e-3 * lane;
    __syncthreads();
    const AB* sh = ab[wid];
    const double X1 = lane * 0.137 - 4.0, C = -0.01 * X1 * X1;
    double acc0 = 0, acc1 = 0;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < 64; j += 2) {
            const AB r0 = sh[j], r1 = sh[j + 1];
            acc0 += exp_v<VAR>(fma(r0.b, X1, r0.a) + C, tab, lane);
            acc1 += exp_v<VAR>(fma(r1.b, X1, r1.a) + C, tab, lane);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
}

template <typename F>
double time_ms(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    std::vector<double> tab(2048);
    for (int j = 0; j < 2048; ++j) tab[j] = std::exp2(j / 2048.0);
    hipMemcpyToSymbol(HIP_SYMBOL(TABG), tab.data(), 2048 * 8);
    double* out; hipMalloc(&out, 8ull * 256 * 256 * 8);
    const int iters = 1024;
    for (int bpc : {4, 5}) {
        const int grid = 256 * bpc;
        const double terms = (double)grid * 4 * iters * 64;
        double ms;
        ms = time_ms([&] { loop_kernel<0><<<grid, 256>>>(out, iters); });
        printf("%d waves/SIMD  var0 2048-table cubic      : %.3f ms -> %.1f cyc/wave-term (2.4GHz units)\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { loop_kernel<1><<<grid, 256>>>(out, iters); });
        printf("%d waves/SIMD  var1 table, no conflicts   : %.3f ms -> %.1f\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { loop_kernel<2><<<grid, 256>>>(out, iters); });
        printf("%d waves/SIMD  var2 64x32 replicated deg5 : %.3f ms -> %.1f\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { loop_kernel<3><<<grid, 256>>>(out, iters); });
        printf("%d waves/SIMD  var3 no table read         : %.3f ms -> %.1f\n", bpc, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
    }
    return 0;
}
