// micro-benchmark of the fused f2v term loop (tuning aid, not part of the product): where do the cycles go?
//   REC   0 = (a,b) record from LDS (uniform ds_read_b128)   1 = record from registers (no LDS read)
//   TAB   0 = random table gather (ds_read_b64)   1 = conflict-free gather   2 = no table read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ double TABG[2048];
struct AB { double a, b; };

template <int TAB>
__device__ __forceinline__ double acc_v(double acc, double t, double magic, const double* tab, int lane) {
    const double u = fma(t, 2954.639443740597, magic);
    const int nn = __double2loint(u);
    const double kd = u - magic;
    double r = fma(kd, -0.0003384507717577858, t);
    r = fma(kd, -1.1323470770733885e-20, r);
    double p = fma(r, 1.6666666666666666667e-1, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double T;
    if (TAB == 0) T = tab[nn & 2047];
    else if (TAB == 1) T = tab[(lane * 33 + (nn & 1)) & 2047];
    else T = __hiloint2double(0x3ff00000 + (nn & 2047), nn);
    return fma(ldexp(T, nn >> 11), p, acc);
}

template <int REC, int TAB>
__global__ void __launch_bounds__(256) loop_kernel(double* out, int iters) {
    __shared__ double tab[2048];
    __shared__ AB ab[4][64];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = TABG[i];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ab[wid][lane].a = -1.0 - 1e-2 * lane; ab[wid][lane].b = 1e-3 * lane;
    __syncthreads();
    const AB* sh = ab[wid];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const double X1 = (h & 0xffff) * (20.0 / 65536) - 10.0;
    const double magic = 6755399441055744.0 + (double)(int)(-30.0 * X1 * X1);
    double acc0 = 0, acc1 = 0;
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < 64; j += 2) {
            AB r0, r1;
            if (REC == 0) { r0 = sh[j]; r1 = sh[j + 1]; }
            else { r0.a = -1.0 - 1e-2 * j; r0.b = 1e-3 * j + it; r1.a = -1.01 - 1e-2 * j; r1.b = 1e-3 * j + it; }
            acc0 = acc_v<TAB>(acc0, fma(r0.b, X1, r0.a), magic, tab, lane);
            acc1 = acc_v<TAB>(acc1, fma(r1.b, X1, r1.a), magic, tab, lane);
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
    const int iters = 512;
    for (int bpc : {1, 2, 4, 5, 7}) {
        const int grid = 256 * bpc;
        const double terms = (double)grid * 4 * iters * 64;
        double ms;
#define RUN(R, T, name) ms = time_ms([&] { loop_kernel<R, T><<<grid, 256>>>(out, iters); }); \
        printf("%d waves/SIMD  %-40s: %.3f ms -> %.1f units/wave-term\n", bpc, name, ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        RUN(0, 0, "lds record + random table gather")
        RUN(0, 1, "lds record + conflict-free gather")
        RUN(0, 2, "lds record, no table")
        RUN(1, 0, "reg record + random table gather")
        RUN(1, 2, "reg record, no table (pure VALU)")
    }
    return 0;
}
