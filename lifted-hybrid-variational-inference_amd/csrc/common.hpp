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

}  // namespace lhvi
