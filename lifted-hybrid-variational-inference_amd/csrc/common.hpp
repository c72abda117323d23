// common.hpp -- launch helpers and small device utilities shared by the lhvi kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/lhvi.h"

namespace lhvi {

constexpr int WAVE = 64;          // CDNA4 wavefront
constexpr int BLOCK = 256;        // default workgroup: 4 waves = one per SIMD

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { g_last_hip_error = (int)err; return LHVI_E_LAUNCH; }
    return LHVI_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned grid_for(int64_t work, int block = BLOCK) {
    int64_t g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g);
}

// canonical edge of e (identity when the graph has no repeated clusters in a factor scope)
__device__ __forceinline__ int canon(const int32_t* __restrict__ edge_canon, int e) {
    return edge_canon ? edge_canon[e] : e;
}

__device__ __forceinline__ bool is_hidden(double value) { return value != value; }

struct double2_ { double x, y; };

__device__ __forceinline__ double2 ld2(const double* p, int64_t i) {
    return reinterpret_cast<const double2*>(p)[i];
}
__device__ __forceinline__ void st2(double* p, int64_t i, double a, double b) {
    reinterpret_cast<double2*>(p)[i] = make_double2(a, b);
}

// ---- wavefront reductions on the DPP data path (no LDS crossbar round trips) --------------------------------------
// The classic gfx9 sequence: two quad permutes, two row rotations, row_bcast:15, row_bcast:31 leave the total of the
// wave in lane 63 (and the totals of lanes 0-31 / 32-63 in lanes 31 / 63 before the last step); v_readlane broadcasts.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
struct SumOp { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct MaxOp { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };

template <class Op>
__device__ __forceinline__ double dpp_reduce_rows32(double v, Op op) {      // lanes 31 / 63 hold the half-wave results
    v = op(dpp_move<0xb1>(v), v);      // quad_perm [1,0,3,2]
    v = op(dpp_move<0x4e>(v), v);      // quad_perm [2,3,0,1]
    v = op(dpp_move<0x124>(v), v);     // row_ror:4
    v = op(dpp_move<0x128>(v), v);     // row_ror:8
    v = op(dpp_move<0x142>(v), v);     // row_bcast:15
    return v;
}
template <class Op>
__device__ __forceinline__ double dpp_row_reduce(double v, Op op) {         // every lane gets its 16-lane row's result
    v = op(dpp_move<0xb1>(v), v);
    v = op(dpp_move<0x4e>(v), v);
    v = op(dpp_move<0x124>(v), v);
    return op(dpp_move<0x128>(v), v);
}
template <class Op>
__device__ __forceinline__ double dpp_wave_reduce(double v, Op op) {        // every lane gets the wave-wide result
    v = dpp_reduce_rows32(v, op);
    v = op(dpp_move<0x143>(v), v);     // row_bcast:31
    return readlane_f64(v, 63);
}
template <class Op>
__device__ __forceinline__ double dpp_half_reduce(double v, Op op, int lane) {   // result of the lane's own 32-lane half
    v = dpp_reduce_rows32(v, op);
    const double a = readlane_f64(v, 31), b = readlane_f64(v, 63);
    return lane < 32 ? a : b;
}

// log(x) for x > 0 (denormals and +inf included): x = m * 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s),
// s = (m - 1)/(m + 1), |s| <= 0.1716, odd series through s^19 (next term < 3e-17 relative).  ~35 fp64 operations against
// ~95 for the ocml routine (which carries double-double intermediates for < 1 ulp); measured max error 2 ulp
// (tests/test_gpu_pbp.py::test_device_log_accuracy).  Used where a kernel takes one log per output point.
// sqrt(x), x > 0 normal: v_rsq_f64 seed (about 24 good bits) + one coupled Newton step + a residual correction
__device__ __forceinline__ double sqrt_pos(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    return fma(fma(-g, g, x), h, g);
}

// 1 / x for finite normal x != 0: v_rcp_f64 seed + two Newton steps (< 1 ulp); for divisors shared by many divisions
__device__ __forceinline__ double rcp_newton(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// A double constant materialised in a scalar register pair right where it is used.  Left to itself the compiler keeps
// every fp64 literal of a polynomial in a VGPR pair for the whole kernel (v_fmac wants its addend in a VGPR), which
// costs the persistent kernels a wave of occupancy; scalar moves are free next to fp64 VALU work.
template <unsigned long long BITS>
__device__ __forceinline__ double scalar_const() {
    int lo, hi;
    asm("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((int)(BITS & 0xffffffffull)), "i"((int)(BITS >> 32)));
    return __hiloint2double(hi, lo);
}
#define LHVI_SCONST(x) scalar_const<__builtin_bit_cast(unsigned long long, (double)(x))>()

__device__ __forceinline__ double log_pos(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    const bool low = m < LHVI_SCONST(0.70710678118654752440);
    m = low ? m + m : m;
    e -= low ? 1 : 0;
    const double num = m - 1.0, den = m + 1.0;
    double rc = __builtin_amdgcn_rcp(den);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    double s = num * rc;
    s = fma(fma(-den, s, num), rc, s);
    const double z = s * s;
    double p = LHVI_SCONST(1.0 / 19.0);
    p = fma(p, z, LHVI_SCONST(1.0 / 17.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 15.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 13.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 11.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 9.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 7.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 5.0));
    p = fma(p, z, LHVI_SCONST(1.0 / 3.0));
    const double ed = (double)e;
    double r = fma(ed, LHVI_SCONST(1.90821492927058770002e-10), 2.0 * (s * z * p));    // ln2 low part
    r += 2.0 * s;
    r = fma(ed, LHVI_SCONST(6.93147180369123816490e-01), r);                            // ln2 high part (21 trailing zero bits)
    return x == __builtin_huge_val() ? x : r;
}

}  // namespace lhvi
