// micro-benchmark + accuracy check of the "floor" form of the f2v term loop against the shipped one (tuning aid, not part of
// the product).  Shipped: t = a + b x; u = t / step + magic; kd = u - magic; r = t - kd step; cubic in r  (9 fp64 operations per
// term).  Floor form: records pre-divided by the step, s = a' + b' x; u = s + magic with the fp64 rounding mode set to
// round-down for that one addition, so that the low word of u is floor(s) exactly; f = v_fract_f64(s) = s - floor(s) is the
// polynomial argument (8 fp64 operations per term).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lifted-hybrid-variational-inference_amd/csrc -I include scripts/ubench/term_floor.hip -o scripts/ubench/term_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "fastmath.hpp"

using namespace lhvi;

struct AB2 { double a, b; };

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) shipped_kernel(const AB2* __restrict__ rec, const double* __restrict__ xs, double kx, double* __restrict__ out, int iters) {
    __shared__ AB2 sh_all[4][64];
    __shared__ double sh_tab[EXP_TAB_N];
    load_exp_table(sh_tab);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    AB2* sh = sh_all[wid];
    sh[lane] = rec[(blockIdx.x * 4 + wid) % 64 * 64 + lane];
    __syncthreads();
    const double X1 = xs[(blockIdx.x * 256 + threadIdx.x) % 4096], C = kx * X1 * X1;
    double total = 0.0;
    for (int it = 0; it < iters; ++it) {
        const ExpShift sft = exp_shift(C);
        double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        for (int j = 0; j < 64; j += 4) {
            const AB2 r0 = sh[j], r1 = sh[j + 1], r2 = sh[j + 2], r3 = sh[j + 3];
            acc0 = exp_accumulate(acc0, fma(r0.b, X1, r0.a), sft.magic, sh_tab);
            acc1 = exp_accumulate(acc1, fma(r1.b, X1, r1.a), sft.magic, sh_tab);
            acc2 = exp_accumulate(acc2, fma(r2.b, X1, r2.a), sft.magic, sh_tab);
            acc3 = exp_accumulate(acc3, fma(r3.b, X1, r3.a), sft.magic, sh_tab);
        }
        total += ((acc0 + acc2) + (acc1 + acc3)) * sft.scale;
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) floor_kernel(const AB2* __restrict__ rec, const double* __restrict__ xs, double kx, double* __restrict__ out, int iters) {
    __shared__ AB2 sh_all[4][64];
    __shared__ double sh_tab[EXP_TAB_N];
    load_exp_table(sh_tab);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    AB2* sh = sh_all[wid];
    {
        AB2 r = rec[(blockIdx.x * 4 + wid) % 64 * 64 + lane];
        r.a *= LHVI_EXP_INV_STEP; r.b *= LHVI_EXP_INV_STEP;
        sh[lane] = r;
    }
    __syncthreads();
    const double X1 = xs[(blockIdx.x * 256 + threadIdx.x) % 4096], C = kx * X1 * X1;
    double total = 0.0;
    for (int it = 0; it < iters; ++it) {
        const ExpShiftFloor sft = exp_shift_floor(C);
        double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        round_down_on();
        for (int j = 0; j < 64; j += 4) {
            const AB2 r0 = sh[j], r1 = sh[j + 1], r2 = sh[j + 2], r3 = sh[j + 3];
            acc0 = exp_accumulate_floor(acc0, fma(r0.b, X1, r0.a), sft.magic, sh_tab);
            acc1 = exp_accumulate_floor(acc1, fma(r1.b, X1, r1.a), sft.magic, sh_tab);
            acc2 = exp_accumulate_floor(acc2, fma(r2.b, X1, r2.a), sft.magic, sh_tab);
            acc3 = exp_accumulate_floor(acc3, fma(r3.b, X1, r3.a), sft.magic, sh_tab);
        }
        round_down_off();
        total += ((acc0 + acc2) + (acc1 + acc3)) * sft.scale;
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

// the two halves of the trick on their own: floor(s) from the low word of RD(s + magic), and s - floor(s) from v_fract
__global__ void consistency_kernel(const double* __restrict__ s, int n, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double MAGIC = 844424930131968.0;             // 1.5 * 2^49: ulp 1/8, floor(s) from bit 3 of the low word up
    round_down_on();
    double u;
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(u) : "v"(s[i]), "v"(MAGIC));
    round_down_off();
    const int nn = __double2loint(u) >> 3;
    const double f = __builtin_amdgcn_fract(s[i]);
    const double fl = floor(s[i]);
    if ((double)nn != fl || !(f >= 0.0 && f < 1.0) || fabs((fl + f) - s[i]) > 1e-9 * fabs(s[i]) + 1e-300) atomicAdd(bad, 1);
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
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::uniform_real_distribution<double> ud(-10.0, 10.0);
    std::vector<AB2> rec(64 * 64);
    std::vector<double> xs(4096);
    for (auto& r : rec) { r.a = 3.0 * nd(rng) - 2.0; r.b = 1.5 * nd(rng); }
    for (auto& x : xs) x = ud(rng);
    const double kx = -0.37;
    AB2* d_rec; double *d_x, *d_out;
    hipMalloc(&d_rec, rec.size() * sizeof(AB2)); hipMalloc(&d_x, xs.size() * 8);
    hipMemcpy(d_rec, rec.data(), rec.size() * sizeof(AB2), hipMemcpyHostToDevice);
    hipMemcpy(d_x, xs.data(), xs.size() * 8, hipMemcpyHostToDevice);
    const int grid = 256 * 7;
    hipMalloc(&d_out, (size_t)grid * 256 * 8);
    // consistency on values around integers, ties, negatives, huge
    {
        std::vector<double> s;
        for (int k = -3000000; k <= 3000000; k += 9973) for (double e : {0.0, 0.5, -0.5, 1e-9, -1e-9, 0.4999999999, 0.25, 2.220446049250313e-16 * k})
            s.push_back(k + e);
        for (int i = 0; i < 2000000; ++i) s.push_back(2.0e6 * nd(rng));
        double* d_s; int* d_bad; hipMalloc(&d_s, s.size() * 8); hipMalloc(&d_bad, 4); hipMemset(d_bad, 0, 4);
        hipMemcpy(d_s, s.data(), s.size() * 8, hipMemcpyHostToDevice);
        consistency_kernel<<<(s.size() + 255) / 256, 256>>>(d_s, (int)s.size(), d_bad);
        int bad; hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        printf("consistency: %zu values, %d with low word of RD(s + magic) != floor(s) or fract(s) off\n", s.size(), bad);
    }
    // accuracy of one pass against long double
    std::vector<double> o1((size_t)grid * 256), o2(o1.size());
    shipped_kernel<<<grid, 256>>>(d_rec, d_x, kx, d_out, 1); hipMemcpy(o1.data(), d_out, o1.size() * 8, hipMemcpyDeviceToHost);
    floor_kernel<<<grid, 256>>>(d_rec, d_x, kx, d_out, 1); hipMemcpy(o2.data(), d_out, o2.size() * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, m1 = 0, m2 = 0;
    const size_t N = 64 * 256;
    for (size_t i = 0; i < N; ++i) {
        const int blk = (int)(i / 256), tid = (int)(i % 256), wid = tid >> 6;
        const AB2* r = &rec[(size_t)((blk * 4 + wid) % 64) * 64];
        const long double X = xs[i % 4096];
        long double ref = 0;
        for (int j = 0; j < 64; ++j) ref += expl((long double)r[j].a + (long double)r[j].b * X + (long double)kx * X * X);
        const double r1 = (double)fabsl((o1[i] - ref) / ref), r2 = (double)fabsl((o2[i] - ref) / ref);
        e1 = fmax(e1, r1); e2 = fmax(e2, r2); m1 += r1 / N; m2 += r2 / N;
    }
    printf("relative error of the 64-term sums: shipped max %.3g mean %.3g   floor form max %.3g mean %.3g\n", e1, m1, e2, m2);
    const int iters = 512;
    for (int rep = 0; rep < 2; ++rep) {
        const double terms = (double)grid * 4 * iters * 64;
        double ms = time_ms([&] { shipped_kernel<<<grid, 256>>>(d_rec, d_x, kx, d_out, iters); });
        printf("shipped loop: %.3f ms -> %.1f cycles per wave-term per SIMD (2.4 GHz units)\n", ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
        ms = time_ms([&] { floor_kernel<<<grid, 256>>>(d_rec, d_x, kx, d_out, iters); });
        printf("floor form  : %.3f ms -> %.1f\n", ms, ms * 1e-3 * 2.4e9 * 1024 / terms);
    }
    return 0;
}
