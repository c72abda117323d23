// halo.hip -- pack / unpack of the owner-computes exchange of the edge-sharded particle sweep (gfx950 only).
//
// Variables are partitioned over the ranks; a rank computes every message whose TARGET it owns.  For a factor cut by the
// partition, the f -> v message towards an owned variable needs the partner's v -> f message (EPBPLogVersion.py:250-258:
// message[(rv, f)], one value per particle of rv) and the partner's particles, which the rank regenerates itself from the
// partner's proposal q[rv] (EPBPLogVersion.py:83-101) with the sampler keyed by the variable's global id.  So what travels per
// sweep is, per peer, a block of v -> f rows (of the cut edges whose variable is owned here) followed by a block of proposals
// (of the owned variables that are ghosts there) -- packed by one launch into the send buffer of the sweep's ONE all_to_all and
// scattered by one launch out of the receive buffer.  Both ends list the rows in ascending (global factor, position) and the
// proposals in ascending global variable id, so no index travels.
#include "common.hpp"

namespace lhvi {

// one wavefront per row: lanes copy the row's `width` doubles (n for a continuous variable, its number of states otherwise)
template <bool PACK>
__global__ void __launch_bounds__(BLOCK) halo_rows_kernel(int n, int n_rows, const int32_t* __restrict__ row_edge,
                                                          const int64_t* __restrict__ row_off, const int32_t* __restrict__ row_width,
                                                          double* __restrict__ v2f, double* __restrict__ buf) {
    const int64_t row = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / WAVE;      // (64-bit: n_rows * 64 passes 2^31 from 33 M rows on)
    const int lane = threadIdx.x & (WAVE - 1);
    if (row >= n_rows) return;
    const int64_t e = row_edge[row], off = row_off[row];
    const int w = row_width[row];
    for (int j = lane; j < w; j += WAVE) {
        if (PACK) buf[off + j] = v2f[e * n + j];
        else v2f[e * n + j] = buf[off + j];
    }
}

template <bool PACK>
__global__ void __launch_bounds__(BLOCK) halo_q_kernel(int n_q, const int32_t* __restrict__ q_var, const int64_t* __restrict__ q_off,
                                                       double* __restrict__ q, double* __restrict__ buf) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_q) return;
    const int64_t v = q_var[i], off = q_off[i];
    if (PACK) { buf[off] = q[2 * v]; buf[off + 1] = q[2 * v + 1]; }
    else { q[2 * v] = buf[off]; q[2 * v + 1] = buf[off + 1]; }
}

template <bool PACK>
static int halo(double* v2f, int n, int n_rows, const int32_t* row_edge, const int64_t* row_off, const int32_t* row_width,
                double* q, int n_q, const int32_t* q_var, const int64_t* q_off, double* buf, void* stream) {
    if (n <= 0 || n_rows < 0 || n_q < 0) return LHVI_E_ARG;
    if (n_rows > 0 && (!v2f || !row_edge || !row_off || !row_width || !buf)) return LHVI_E_ARG;
    if (n_q > 0 && (!q || !q_var || !q_off || !buf)) return LHVI_E_ARG;
    if (n_rows > 0)
        hipLaunchKernelGGL(halo_rows_kernel<PACK>, dim3(grid_for((int64_t)n_rows * WAVE)), dim3(BLOCK), 0, as_stream(stream), n, n_rows,
                           row_edge, row_off, row_width, v2f, buf);
    if (n_q > 0)
        hipLaunchKernelGGL(halo_q_kernel<PACK>, dim3(grid_for(n_q)), dim3(BLOCK), 0, as_stream(stream), n_q, q_var, q_off, q, buf);
    return check_launch();
}

}  // namespace lhvi

using namespace lhvi;

extern "C" {

int lhvi_pbp_halo_pack(const double* v2f, int32_t n, int32_t n_rows, const int32_t* row_edge, const int64_t* row_off,
                       const int32_t* row_width, const double* q, int32_t n_q, const int32_t* q_var, const int64_t* q_off,
                       double* out, void* stream) {
    return halo<true>(const_cast<double*>(v2f), n, n_rows, row_edge, row_off, row_width, const_cast<double*>(q), n_q, q_var, q_off, out, stream);
}

int lhvi_pbp_halo_unpack(const double* in, int32_t n, int32_t n_rows, const int32_t* row_edge, const int64_t* row_off,
                         const int32_t* row_width, double* v2f, int32_t n_q, const int32_t* q_var, const int64_t* q_off,
                         double* q, void* stream) {
    return halo<false>(v2f, n, n_rows, row_edge, row_off, row_width, q, n_q, q_var, q_off, const_cast<double*>(in), stream);
}

}  // extern "C"
