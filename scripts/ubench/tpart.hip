// micro-benchmark (tuning aid, not part of the product): the integral-point part of a heavy f2v edge,
//   S_t = sum_j exp(a_j + b_j x_t),  x_t = x_0 + t h,  t < 32, j < 64,
// (A) the way the kernel does it today -- lane = (half, t), 32 table exponentials per lane, one fold -- against
// (B) lane = j: G_{t+1,j} = G_{t,j} r_j with r_j = exp(b_j h) (the grid is uniform), and a reduce-scatter over the
//     lanes that leaves S_t in the lanes that own t: v_permlane32_swap / v_permlane16_swap for lane bits 5 and 4 (one swap
//     per dword moves two values, no selects), bank-masked DPP for bits 3 and 2, quad permutes for bits 1 and 0.
// Prints cycles per edge for both and the largest relative difference of the sums.
// build: hipcc -O3 --offload-arch=gfx950 -I../../lifted-hybrid-variational-inference_amd/csrc -I../../include tpart.hip -o tpart
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "fastmath.hpp"

using namespace lhvi;

struct Rec { double a, b; };

template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_into(double old, double src) {      // lanes of the enabled banks take src[permuted], the rest keep old
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}

// x kept by the lanes whose bit is 0, y by the others; returns own-kept + partner's copy of the same value
__device__ __forceinline__ double fold32(double x, double y) {
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold16(double x, double y) {
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold8(double x, double y) {             // lane ^ 8 = row_ror:8; bit 3 = banks 2, 3
    const double t = dpp_into<0x128, 0x3>(y, x);      // bit3 = 0: partner's x, else own y
    const double u = dpp_into<0x128, 0xc>(x, y);      // bit3 = 1: partner's y, else own x
    return t + u;
}
__device__ __forceinline__ double fold4(double x, double y) {             // lane ^ 4: bit 2 = banks 1, 3
    const double t = dpp_into<0x104, 0x5>(y, x);      // row_shl:4 (read lane + 4) into banks 0, 2
    const double u = dpp_into<0x114, 0xa>(x, y);      // row_shr:4 (read lane - 4) into banks 1, 3
    return t + u;
}
__device__ __forceinline__ double fold2(double x, double y, bool bit1) {  // lane ^ 2 = quad_perm [2,3,0,1]
    const double send = bit1 ? x : y, keep = bit1 ? y : x;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), 0x4e, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), 0x4e, 0xf, 0xf, false);
    return keep + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fold1(double x) {                       // lane ^ 1 = quad_perm [1,0,3,2]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xb1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xb1, 0xf, 0xf, false);
    return x + __hiloint2double(hi, lo);
}

// the lane's value ends up being S_t for t = 8 (2 b1 + b2) + 4 b3 + 2 b4 + b5  (b_i = bit i of the lane id)
__device__ __forceinline__ int owned_t(int lane) {
    return 8 * (2 * ((lane >> 1) & 1) + ((lane >> 2) & 1)) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);
}

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) tpart_kernel(const Rec* __restrict__ recs, int nrec,
                                                                                             double* __restrict__ out, int edges, double x0, double h) {
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    __shared__ Rec sh_rec[4][64];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + wid;
    Rec* sh = sh_rec[wid];
    double check = 0.0;
    for (int e = 0; e < edges; ++e) {
        const Rec r = recs[(size_t)((wave * 17 + e) % nrec) * 64 + lane];
        double res;
        int t;
        if (MODE == 0) {
            __builtin_amdgcn_wave_barrier();
            sh[lane] = r;
            __builtin_amdgcn_wave_barrier();
            t = lane & 31;
            const int sub = lane >> 5;
            const double X1 = x0 + t * h;
            const ExpShift sft = exp_shift(0.0);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const Rec* base = sh + 32 * sub;
            for (int j = 0; j < 32; j += 4) {
                a0 = exp_accumulate(a0, fma(base[j].b, X1, base[j].a), sft.magic, sh_tab);
                a1 = exp_accumulate(a1, fma(base[j + 1].b, X1, base[j + 1].a), sft.magic, sh_tab);
                a2 = exp_accumulate(a2, fma(base[j + 2].b, X1, base[j + 2].a), sft.magic, sh_tab);
                a3 = exp_accumulate(a3, fma(base[j + 3].b, X1, base[j + 3].a), sft.magic, sh_tab);
            }
            res = ((a0 + a2) + (a1 + a3)) * sft.scale;
            res += __shfl_xor(res, 32);
        } else {
            double g = exp_core(fma(r.b, x0, r.a), sh_tab);
            const double q = exp_core(r.b * h, sh_tab);
            double z[4];
#pragma unroll
            for (int bt = 0; bt < 4; ++bt) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = g; g *= q; }
                const double w0 = fold32(v[0], v[1]), w1 = fold32(v[2], v[3]), w2 = fold32(v[4], v[5]), w3 = fold32(v[6], v[7]);
                const double u0 = fold16(w0, w1), u1 = fold16(w2, w3);
                z[bt] = fold8(u0, u1);
            }
            const double y0 = fold4(z[0], z[1]), y1 = fold4(z[2], z[3]);
            res = fold1(fold2(y0, y1, (lane >> 1) & 1));
            t = owned_t(lane);
        }
        const double lg = log_table(res, sh_log);
        if (MODE == 0 ? lane < 32 : !(lane & 1)) out[(size_t)wave * 32 + t] = lg + check;
        check += 1e-30 * lg;
    }
}

int main() {
    const int nrec = 4096, edges = 400;
    std::vector<Rec> h_recs((size_t)nrec * 64);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0); };
    for (auto& r : h_recs) { r.a = -60.0 + 120.0 * rnd(); r.b = -5.0 + 10.0 * rnd(); }
    Rec* d_recs; double* d_out[2];
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 7, waves = blocks * 4;
    hipMalloc(&d_recs, h_recs.size() * sizeof(Rec));
    hipMemcpy(d_recs, h_recs.data(), h_recs.size() * sizeof(Rec), hipMemcpyHostToDevice);
    for (int m = 0; m < 2; ++m) hipMalloc(&d_out[m], (size_t)waves * 32 * sizeof(double));
    const double x0 = -10.0, hh = 20.0 / 31.0;
    float ms[2];
    for (int m = 0; m < 2; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (m == 0) hipLaunchKernelGGL(tpart_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_recs, nrec, d_out[0], edges, x0, hh);
            else hipLaunchKernelGGL(tpart_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_recs, nrec, d_out[1], edges, x0, hh);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        hipEventElapsedTime(&ms[m], a, b);
    }
    std::vector<double> o0((size_t)waves * 32), o1(o0.size());
    hipMemcpy(o0.data(), d_out[0], o0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), d_out[1], o1.size() * 8, hipMemcpyDeviceToHost);
    double maxd = 0.0;
    for (size_t i = 0; i < o0.size(); ++i) maxd = fmax(maxd, fabs(o0[i] - o1[i]));
    const double clk = 2.0e9;       // sustained clock under this load
    for (int m = 0; m < 2; ++m)
        printf("%s: %.3f ms, %.0f cycles per edge per SIMD-resident wave set (7 waves/SIMD)\n", m == 0 ? "direct    " : "recurrence",
               ms[m], ms[m] * 1e-3 * clk / edges / 7.0);
    printf("max |log S_direct - log S_recurrence| = %.3e\n", maxd);
    return 0;
}
