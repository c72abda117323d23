// color.hip -- colour refinement (colour passing) half-rounds on the GPU, integer only.
//
// Reference semantics: CompressedGraphWithObs.py:47-76 (SuperRV.split_by_structure),
// :152-175 (SuperF.split_by_structure); CompressedGraphSorted.py:40-61,90-116 use the same signatures.
//   factor signature = (old factor colour, tuple of its scope's rv colours, sorted iff potential.symmetric)
//   rv signature     = (old rv colour, sorted multiset of the colours of its incident factors)
// Only the induced *partition* matters (cluster objects have no stable ids in the reference), so a new
// colour is the dense rank of the signature.
//
// Exactness: signatures are reduced to two independent 64-bit fingerprints (h1, h2).  Items are ranked by
// h1 (radix sort); if two adjacent items agree on h1 but not on h2 a collision flag is raised and the
// host retries with another seed, so a wrong merge needs a simultaneous 128-bit collision.  The rv
// fingerprint is a *sum* of per-neighbour mixes: integer addition commutes, so the multiset needs no
// sorting and the atomics below are deterministic.
//
// Bound: HBM (gathers + a 64-bit radix sort), ~120 B per edge per round.
#include "common.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace lhvi {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr uint64_t SEED1 = 0x243F6A8885A308D3ull, SEED2 = 0x13198A2E03707344ull;

__global__ void __launch_bounds__(BLOCK) factor_sig_kernel(lhvi_graph_t g, const uint8_t* __restrict__ symmetric,
                                                          const int32_t* __restrict__ rv_color,
                                                          const int32_t* __restrict__ f_color, uint64_t seed,
                                                          uint64_t* __restrict__ h1, uint64_t* __restrict__ h2,
                                                          uint32_t* __restrict__ idx) {
    int f = blockIdx.x * BLOCK + threadIdx.x;
    if (f >= g.F) return;
    const int base = g.fac_ptr[f];
    const int a = min(g.fac_ptr[f + 1] - base, LHVI_MAX_ARITY);
    int32_t c[LHVI_MAX_ARITY];
#pragma unroll
    for (int p = 0; p < LHVI_MAX_ARITY; ++p) c[p] = p < a ? rv_color[g.edge_var[base + p]] : -1;
    if (symmetric && symmetric[f]) {            // tuple(sorted(...)) for symmetric potentials
#pragma unroll
        for (int i = 1; i < LHVI_MAX_ARITY; ++i)
#pragma unroll
            for (int j = LHVI_MAX_ARITY - 1; j >= i; --j)
                if (j < a && c[j] < c[j - 1]) { int32_t t = c[j]; c[j] = c[j - 1]; c[j - 1] = t; }
    }
    uint64_t a1 = mix64((uint64_t)(uint32_t)f_color[f] ^ seed ^ SEED1);
    uint64_t a2 = mix64((uint64_t)(uint32_t)f_color[f] * 0x100000001B3ull + seed + SEED2);
#pragma unroll
    for (int p = 0; p < LHVI_MAX_ARITY; ++p) {
        if (p < a) {
            a1 = mix64(a1 * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)c[p] + 1);
            a2 = mix64(a2 ^ (((uint64_t)(uint32_t)c[p] + 0x51ull) * 0xD6E8FEB86659FD93ull));
        }
    }
    h1[f] = a1; h2[f] = a2; idx[f] = (uint32_t)f;
}

__global__ void __launch_bounds__(BLOCK) rv_sig_init_kernel(int V, const int32_t* __restrict__ rv_color, uint64_t seed,
                                                           uint64_t* __restrict__ h1, uint64_t* __restrict__ h2,
                                                           uint32_t* __restrict__ idx) {
    int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    h1[v] = mix64((uint64_t)(uint32_t)rv_color[v] ^ seed ^ SEED2);
    h2[v] = mix64(((uint64_t)(uint32_t)rv_color[v] + seed) * 0xA24BAED4963EE407ull + SEED1);
    idx[v] = (uint32_t)v;
}

// Add the incident factors' colour mixes into the variable's two fingerprints.  Integer sums commute, so any order
// gives the same bits: a thread walks the CSR row of a variable with up to HUB_DEGREE edges, a wavefront shares the row
// of a hub (template variables of relational models have thousands of edges).  No atomics: 20 M 64-bit atomic adds on
// a 10 M-edge graph cost 3 ms per round, the segmented sums 0.3.
constexpr int HUB_DEGREE = LHVI_HUB_DEGREE;

__device__ __forceinline__ void sig_terms(const lhvi_graph_t& g, const int32_t* __restrict__ f_color, uint64_t seed, int k,
                                          uint64_t& a, uint64_t& b) {
    const uint64_t c = (uint64_t)(uint32_t)f_color[g.edge_fac[g.var_edge[k]]];
    a += mix64(c ^ seed ^ SEED1);
    b += mix64((c + 0x7Full) * 0xC2B2AE3D27D4EB4Full + seed);
}

__global__ void __launch_bounds__(BLOCK) rv_sig_accum_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                            uint64_t seed, uint64_t* __restrict__ h1,
                                                            uint64_t* __restrict__ h2) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= g.V) return;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo > HUB_DEGREE) return;
    uint64_t a = 0, b = 0;
    for (int k = lo; k < hi; ++k) sig_terms(g, f_color, seed, k, a, b);
    h1[v] += a; h2[v] += b;
}

__global__ void __launch_bounds__(BLOCK) rv_sig_accum_hub_kernel(lhvi_graph_t g, const int32_t* __restrict__ f_color,
                                                                uint64_t seed, uint64_t* __restrict__ h1,
                                                                uint64_t* __restrict__ h2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= (g.hub_vars ? g.n_hubs : g.V)) return;
    const int v = g.hub_vars ? g.hub_vars[i] : i;          // without a hub list: scan every variable
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= HUB_DEGREE) return;
    uint64_t a = 0, b = 0;
    for (int k = lo + lane; k < hi; k += 64) sig_terms(g, f_color, seed, k, a, b);
    for (int off = 32; off > 0; off >>= 1) {
        a += ((uint64_t)(uint32_t)__shfl_xor((int)(a >> 32), off) << 32) | (uint32_t)__shfl_xor((int)a, off);
        b += ((uint64_t)(uint32_t)__shfl_xor((int)(b >> 32), off) << 32) | (uint32_t)__shfl_xor((int)b, off);
    }
    if (lane == 0) { h1[v] += a; h2[v] += b; }
}

__global__ void __launch_bounds__(BLOCK) flag_kernel(int n, const uint64_t* __restrict__ key_sorted,
                                                    const uint32_t* __restrict__ idx_sorted,
                                                    const uint64_t* __restrict__ h2, int32_t* __restrict__ flag,
                                                    int32_t* __restrict__ result) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    int fl = 1;
    if (i > 0 && key_sorted[i] == key_sorted[i - 1]) {
        fl = 0;
        if (h2[idx_sorted[i]] != h2[idx_sorted[i - 1]]) result[1] = 1;   // h1 collision: host retries
    }
    flag[i] = fl;
}

__global__ void __launch_bounds__(BLOCK) scatter_rank_kernel(int n, const uint32_t* __restrict__ idx_sorted,
                                                            const int32_t* __restrict__ rank,
                                                            int32_t* __restrict__ color_out,
                                                            int32_t* __restrict__ result) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    color_out[idx_sorted[i]] = rank[i] - 1;
    if (i == n - 1) result[0] = rank[i];
}

struct Workspace {
    uint64_t *h1, *h1_sorted, *h2;
    uint32_t *idx, *idx_sorted;
    int32_t *flag, *rank;
    void* temp;
    size_t temp_bytes;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t temp_bytes_for(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                              (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>(), (hipStream_t)0);
    return align_up(a > b ? a : b);
}

static size_t carve(Workspace* w, void* base, size_t n) {
    size_t off = 0;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
    w->h1 = (uint64_t*)take(n * 8); w->h1_sorted = (uint64_t*)take(n * 8); w->h2 = (uint64_t*)take(n * 8);
    w->idx = (uint32_t*)take(n * 4); w->idx_sorted = (uint32_t*)take(n * 4);
    w->flag = (int32_t*)take(n * 4); w->rank = (int32_t*)take(n * 4);
    w->temp_bytes = temp_bytes_for(n);
    w->temp = take(w->temp_bytes);
    return off;
}

static int rank_and_scatter(Workspace& w, int n, int32_t* color_out, int32_t* result, hipStream_t st) {
    size_t tb = w.temp_bytes;
    if (rocprim::radix_sort_pairs(w.temp, tb, w.h1, w.h1_sorted, w.idx, w.idx_sorted, (size_t)n, 0, 64, st) != hipSuccess)
        return LHVI_E_LAUNCH;
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, w.h1_sorted, w.idx_sorted, w.h2, w.flag, result);
    tb = w.temp_bytes;
    if (rocprim::inclusive_scan(w.temp, tb, w.flag, w.rank, (size_t)n, rocprim::plus<int32_t>(), st) != hipSuccess)
        return LHVI_E_LAUNCH;
    hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, n, w.idx_sorted, w.rank, color_out, result);
    return check_launch();
}

}  // namespace lhvi

using namespace lhvi;

extern "C" {

size_t lhvi_color_workspace_bytes(const lhvi_graph_t* g) {
    if (!g) return 0;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    return carve(&w, nullptr, n < 1 ? 1 : n);
}

// n_colors_out: device int32[2] = {number of colours, collision flag (retry with another seed if 1)}
int lhvi_color_refine_factors(const lhvi_graph_t* g, const uint8_t* symmetric, const int32_t* rv_color,
                              const int32_t* f_color, int32_t* f_color_out, int32_t* n_colors_out,
                              void* ws, size_t ws_bytes, void* stream) {
    if (!g || !rv_color || !f_color || !f_color_out || !n_colors_out || !ws) return LHVI_E_ARG;
    if (ws_bytes < lhvi_color_workspace_bytes(g)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(n_colors_out, 0, 2 * sizeof(int32_t), st) != hipSuccess) return LHVI_E_LAUNCH;
    if (g->F == 0) return LHVI_OK;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    carve(&w, ws, n);
    const uint64_t seed = 0;
    hipLaunchKernelGGL(factor_sig_kernel, dim3(grid_for(g->F)), dim3(BLOCK), 0, st, *g, symmetric, rv_color, f_color,
                       seed, w.h1, w.h2, w.idx);
    if (int rc = check_launch()) return rc;
    return rank_and_scatter(w, g->F, f_color_out, n_colors_out, st);
}

int lhvi_color_refine_rvs(const lhvi_graph_t* g, const int32_t* f_color, const int32_t* rv_color,
                          int32_t* rv_color_out, int32_t* n_colors_out, void* ws, size_t ws_bytes, void* stream) {
    if (!g || !rv_color || !f_color || !rv_color_out || !n_colors_out || !ws) return LHVI_E_ARG;
    if (ws_bytes < lhvi_color_workspace_bytes(g)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(n_colors_out, 0, 2 * sizeof(int32_t), st) != hipSuccess) return LHVI_E_LAUNCH;
    if (g->V == 0) return LHVI_OK;
    Workspace w;
    size_t n = (size_t)(g->V > g->F ? g->V : g->F);
    carve(&w, ws, n);
    const uint64_t seed = 0;
    hipLaunchKernelGGL(rv_sig_init_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, g->V, rv_color, seed, w.h1, w.h2, w.idx);
    if (g->nnz > 0)
    {
        hipLaunchKernelGGL(rv_sig_accum_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, *g, f_color, seed, w.h1, w.h2);
        const int64_t nh = g->hub_vars ? g->n_hubs : g->V;
        if (nh > 0)
            hipLaunchKernelGGL(rv_sig_accum_hub_kernel, dim3(grid_for(nh * 64)), dim3(BLOCK), 0, st, *g, f_color, seed, w.h1, w.h2);
    }
    if (int rc = check_launch()) return rc;
    return rank_and_scatter(w, g->V, rv_color_out, n_colors_out, st);
}

}  // extern "C"
