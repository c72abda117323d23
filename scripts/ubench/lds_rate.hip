// micro-benchmark: LDS cycles per wave-level read instruction on gfx950 (tuning aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    __shared__ double tab[2048];
    __shared__ double4 rec[4][64];
    for (int i = threadIdx.x; i < 2048; i += 256) tab[i] = i;
    reinterpret_cast<double*>(rec)[threadIdx.x] = threadIdx.x;
    reinterpret_cast<double*>(rec)[threadIdx.x + 256] = threadIdx.x;
    __syncthreads();
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    double acc = 0;
    const double2* r2 = reinterpret_cast<const double2*>(rec[wid]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            h = h * 1664525u + 1013904223u;
            if (MODE == 0) { const double2 r = r2[j]; acc += r.x + r.y; }                 // uniform ds_read_b128
            else if (MODE == 1) acc += tab[(h >> 8) & 2047];                               // random ds_read_b64
            else if (MODE == 2) acc += tab[(j * 64 + lane) & 2047];                        // conflict-free ds_read_b64
            else if (MODE == 3) acc += reinterpret_cast<const double*>(rec[wid])[j];       // uniform ds_read_b64
            else if (MODE == 4) { const double2 r = r2[(j + (lane >> 5) * 32) & 63]; acc += r.x + r.y; }  // two addresses (split round)
            else acc += (double)(h & 3);                                                   // no LDS (ALU baseline)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K>
double run(K kern, double* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 256;
    kern<<<256 * 7, 256>>>(out, 4); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<256 * 7, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.0e9 / (7.0 * 4 * iters * 64);     // cycles (at 2.0 GHz) per wave-instruction per CU
}

int main() {
    double* out; hipMalloc(&out, 8ull * 256 * 7 * 256);
    const double base = run(k<5>, out);
    printf("ALU-only loop: %.2f CU-cycles per iteration\n", base);
    printf("uniform ds_read_b128      : %.2f CU-cycles/instr (total, incl. ALU)\n", run(k<0>, out));
    printf("random ds_read_b64 gather : %.2f\n", run(k<1>, out));
    printf("conflict-free ds_read_b64 : %.2f\n", run(k<2>, out));
    printf("uniform ds_read_b64       : %.2f\n", run(k<3>, out));
    printf("two-address ds_read_b128  : %.2f\n", run(k<4>, out));
    return 0;
}
