// pbp.hip -- particle belief propagation sweep (EPBP / HybridLBP, log space) for gfx950.
//
// Reference semantics: EPBPLogVersion.py:30-215,225-289, HybridLBPLogVersion.py:44-236.
// Log messages are tabulated per edge in HBM:
//     f2v[e][0..n)  at the variable's particles      f2v[e][n..n+T)  at its integral points
//     v2f[e][0..n)  at the variable's particles
// Kernels of a sweep:
//   pbp_v2f_kernel            one wavefront per variable, lane = particle; the incident f2v rows are read coalesced (8n
//                             bytes each), leave-one-out sums in rv.nb order, DPP mean (and, rarely, max) for
//                             log_message_balance.  HBM bound (16n B per edge).
//   pbp_proposal_kernel       one wavefront per variable: T-point moments per incident edge (row reductions), site update
//                             rule, Gaussian product.
//   pbp_resample_uniq_kernel  Philox / Box-Muller particles keyed by the variable's global id + first-occurrence mask.
//   pbp_f2v_heavy_kernel      continuous x continuous edges (99 % of the terms): persistent waves over 128-byte edge
//                             descriptors; partner particle, potential and incoming message folded into (a_j, b_j) records in
//                             LDS; sum_j exp(a_j + b_j x + k x^2) per output point with a table-driven fused exponential at
//                             the particles, and by recurrence along the uniform integral-point grid + a lane
//                             reduce-scatter at the integral points.  fp64-VALU / LDS bound.
//   pbp_f2v_light_kernel      HybridQuadratic edges with a binary or observed discrete side (no staging).
//   pbp_f2v_fast_kernel       every other quadratic-family edge;  pbp_f2v_generic_kernel: any potential / arity.
#include "common.hpp"
#include "potential.hpp"
#include "fastmath.hpp"
#include "cq.hpp"

namespace lhvi {

__device__ __forceinline__ double wave_sum(double v) { return dpp_wave_reduce(v, SumOp()); }
__device__ __forceinline__ double wave_max(double v) { return dpp_wave_reduce(v, MaxOp()); }
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// The log-term of a padding record of the f -> v term loops.  The evaluated term is exp(a + b x + C) with C the output point's own
// constant (kx x^2, plus the target's message in the joint routes), and kx > 0 occurs (MLN formulas with negative weights), so
// the padding must stay below the underflow threshold (-745) for any C a model can produce: -6e5 leaves 1.2e5 of headroom inside
// the table exponential's argument range (|t| < 2^31 ln2 / 2048 = 7.26e5, fastmath.hpp).
constexpr double PAD_LOG_TERM = -6.0e5;

// variable range of the per-variable kernels: [var_lo, var_hi) when set, else every variable
__host__ __device__ __forceinline__ int var_first(const lhvi_pbp_t& s) { return s.var_hi > s.var_lo ? s.var_lo : 0; }
__host__ __device__ __forceinline__ int var_limit(const lhvi_graph_t& g, const lhvi_pbp_t& s) { return s.var_hi > s.var_lo ? s.var_hi : g.V; }

// EPBP.norm_pdf (EPBPLogVersion.py:49-53): sig is a standard deviation
__device__ __forceinline__ double norm_pdf_std(double x, double mu, double sig) {
    const double u = (x - mu) / sig;
    return exp(-u * u * 0.5) / (2.506628274631 * sig);
}

// log(important_weight(x, rv)) (EPBP:156-163, HLBP:173-180).  For an interior particle the reference evaluates
// log(1 / max(N(x; mu, sd), 1e-200)); with N = exp(-u^2/2) / (2.5066 sd) that is min(u^2/2 + log(2.5066 sd), -log 1e-200),
// which needs one division per particle instead of exp + two divisions + log (log_sd_term is per variable).
__device__ __forceinline__ double log_importance(const lhvi_graph_t& g, const lhvi_pbp_t& s, int v, int d, double x,
                                                 double mu, double rsd /* 1 / sd */, double log_norm) {
    const double LOG_1E200 = 460.51701859880916;       // -log(1e-200)
    if (g.dom_cont[d]) {
        if (x == g.dom_lo[d] || x == g.dom_hi[d]) return -LOG_1E200;
    } else {
        if (!(s.flags & LHVI_PBP_EPBP_DISCRETE)) return 0.0;
        const int b = g.dom_ptr[d];
        const int ns = g.dom_ptr[d + 1] - b;
        if (x == g.dom_val[b] || (ns > 1 && x == g.dom_val[b + 1])) return -LOG_1E200;
    }
    const double u = (x - mu) * rsd;
    return fmin(u * u * 0.5 + log_norm, LOG_1E200);
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) pbp_uniq_kernel(int V, int n, const double* __restrict__ particles,
                                                        const int32_t* __restrict__ np, uint8_t* __restrict__ uniq) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)V * n) return;
    const int v = (int)(i / n), j = (int)(i % n);
    int u = j < np[v];
    const double x = particles[i];
    const double* row = particles + (int64_t)v * n;
    for (int k = 0; k < j && u; ++k)
        if (row[k] == x) u = 0;
    uniq[i] = (uint8_t)u;
}

// n <= 64: one wavefront per variable, lane = particle; the row lives in registers and is broadcast lane by lane
__global__ void __launch_bounds__(BLOCK) pbp_uniq_wave_kernel(int V, int n, const double* __restrict__ particles,
                                                             const int32_t* __restrict__ np, uint8_t* __restrict__ uniq) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (v >= V) return;
    const int cnt = np[v];
    const double x = lane < n ? particles[(int64_t)v * n + lane] : 0.0;
    int u = lane < cnt;
    for (int k = 0; k < cnt - 1; ++k) {
        const double xk = __shfl(x, k);
        if (k < lane && xk == x) u = 0;
    }
    if (lane < n) uniq[(int64_t)v * n + lane] = (uint8_t)u;
}

// ---------------------------------------------------------------------------------------------
// One wavefront per variable, lane = particle.  Every incident f2v row is read once (coalesced 8n bytes), the
// per-particle total T_j = sum_k c_k m_k[j] is formed in rv.nb order, and edge k's message is T_j - m_k[j] + log w_j.
// Log messages are additive, so "total minus own" costs an absolute error of a few ulp of |T| (~1e-13), far inside
// the stated tolerance; it replaces the reference's O(deg^2) re-summation (EPBP:165-174) by O(deg).
// The first V2F_CACHE rows stay in registers (compile-time indices only: a runtime-indexed register array made
// hipcc 7.2 emit an out-of-range s_set_gpr_idx store); their loads are issued back to back so that a degree-4
// variable has four 512-byte rows in flight per wave instead of one (eight slots were measured 5 % slower: scalar-register
// spills; rows beyond the cache are read a second time, out of L2).
constexpr int V2F_CACHE = 4;

// HALO: the build that also writes a cut edge's row into a shard's send buffer (s.halo_off / s.halo_buf); the other one carries
// none of that code (the check alone cost the single-GPU launch 11 %: 1.66 -> 1.84 ms)
template <bool HALO>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) pbp_v2f_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                       double* __restrict__ v2f) {
    const int lane = threadIdx.x & 63;
    int v, np, deg, d, lo;
    int e4[V2F_CACHE] = {0, 0, 0, 0};
    const bool records = s.v2f_wide && (s.flags & LHVI_PBP_V2F_RECORDS);
    if (records) {
        // one 32-byte record per variable (one scalar load), then the rows: walked through the graph -- id -> var_value / np /
        // var_ptr -> var_edge -> rows -- a wave waits out four dependent global round trips, and with every wave slot taken the
        // launch lasts as long as those chains (1.79 ms on the headline graph at 0.65 of the HBM roof in algorithmic bytes)
        const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
        if (item >= s.n_v2f_wide) return;
        const int32_t* rec = s.v2f_wide + 8 * (int64_t)item;
        v = rec[0]; deg = rec[1]; np = rec[2]; d = rec[3];
        e4[0] = rec[4]; e4[1] = rec[5]; e4[2] = rec[6]; e4[3] = rec[7];
        lo = (deg > V2F_CACHE || np > WAVE) ? g.var_ptr[v] : 0;          // (the row's place in var_edge: only rows the record does not cover)
    } else {
        if (s.v2f_wide) {                                  // the caller's list: hidden variables with more than four particles
            const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
            if (item >= s.n_v2f_wide) return;
            v = s.v2f_wide[item];
        } else {
            v = __builtin_amdgcn_readfirstlane(var_first(s) + blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
            if (v >= var_limit(g, s)) return;
        }
        if (!is_hidden(g.var_value[v])) return;
        np = s.np[v];
        lo = g.var_ptr[v];
        deg = g.var_ptr[v + 1] - lo;
        d = g.var_dom[v];
    }
    const int n = s.n, S = s.n + s.T;
    const bool lifted = g.edge_count != nullptr;
    const int nchunk = (np + 63) / 64;
    // per-variable constants through the short routines (a few ulp from the libm ones; same value in every lane)
    const double mu = s.q[2 * v], sd = sqrt_pos(s.q[2 * v + 1]), rsd = rcp_newton(sd);
    const double log_norm = log_pos(2.506628274631 * sd);
    for (int c = 0; c < nchunk; ++c) {
        const int j = c * 64 + lane;
        const bool valid = j < np;
        int ecache[V2F_CACHE];
        double row[V2F_CACHE];
#pragma unroll
        for (int k = 0; k < V2F_CACHE; ++k) ecache[k] = records ? e4[k] : (k < deg ? g.var_edge[lo + k] : 0);
#pragma unroll
        for (int k = 0; k < V2F_CACHE; ++k) row[k] = (k < deg && valid) ? f2v[(int64_t)ecache[k] * S + j] : 0.0;
        const double x = valid ? s.particles[(int64_t)v * n + j] : 0.0;
        const bool uq = valid && s.uniq[(int64_t)v * n + j];
        double total = 0.0;
#pragma unroll
        for (int k = 0; k < V2F_CACHE; ++k)
            if (k < deg) total += lifted ? row[k] * g.edge_count[ecache[k]] : row[k];
        for (int k = V2F_CACHE; k < deg; ++k) {
            const int e = g.var_edge[lo + k];
            const double m = valid ? f2v[(int64_t)e * S + j] : 0.0;
            total += lifted ? m * g.edge_count[e] : m;
        }
        if (s.bslot && valid) {
            // edges of this variable that live on other ranks.  The ranks' sums are added in ascending rank order, this rank's at
            // its own position (like the proposal's), so that the total is the same bits on every rank and under either
            // exchange: rows of every peer (all-to-all), or one row with the finished total (LHVI_PBP_BOUNDARY_TOTALS)
            const int bs = s.bslot[v];
            if (bs >= 0) {
                if (s.flags & LHVI_PBP_BOUNDARY_TOTALS) {
                    total = s.recv[s.brow_off[s.brow_ptr[bs]] + j];
                } else {
                    const double own = total;
                    total = 0.0;
                    bool own_done = false;
                    for (int r = s.brow_ptr[bs]; r < s.brow_ptr[bs + 1]; ++r) {
                        if (!own_done && s.brow_peer[r] > s.rank) { total += own; own_done = true; }
                        total += s.recv[s.brow_off[r] + j];
                    }
                    if (!own_done) total += own;
                }
            }
        }
        const double logw = valid ? log_importance(g, s, v, d, x, mu, rsd, log_norm) : 0.0;
        const int cnt1 = __builtin_popcountll(__ballot(uq));        // distinct particles: the same for every incident edge
        const double rcnt = rcp_newton((double)cnt1);
        auto emit = [&](int e, double m) {
            // ground: sum over nb != f; lifted: own factor keeps count-1 copies (HLBP:182-191) -> total - m either way
            const double res = (total - m) + logw;
            // owner-computes shards: the row of a cut edge goes to its place in the send buffer as it is formed (s.halo_off)
            const int64_t halo = HALO ? s.halo_off[e] : -1;
            if (nchunk == 1) {
                // log_message_balance over the distinct keys (EPBP:204-215)
                const double tot = wave_sum(uq ? res : 0.0);
                const double mean = tot * rcnt;
                double shift = mean;
                // max - mean > max_log_value  <=>  some distinct particle exceeds mean + max_log_value: one ballot
                // decides, and the max reduction runs only in that (rare) case
                if (__ballot(uq && (res - mean > s.max_log_value)))
                    shift = wave_max(uq ? res : -__builtin_huge_val()) - s.max_log_value;
                if (valid) {
                    v2f[(int64_t)e * n + j] = res - shift;
                    if (HALO && halo >= 0) s.halo_buf[halo + j] = res - shift;
                }
            } else if (valid) {
                v2f[(int64_t)e * n + j] = res;       // balanced below once every chunk is written
            }
        };
#pragma unroll
        for (int k = 0; k < V2F_CACHE; ++k)
            if (k < deg) emit(ecache[k], row[k]);
        for (int k = V2F_CACHE; k < deg; ++k) {       // high-degree tail: second touch of the row comes out of L2
            const int e = g.var_edge[lo + k];
            emit(e, valid ? f2v[(int64_t)e * S + j] : 0.0);
        }
    }
    if (nchunk > 1) {
        for (int k = 0; k < deg; ++k) {
            const int e = g.var_edge[lo + k];
            double lsum = 0.0, lmax = -__builtin_huge_val();
            int lcnt = 0;
            for (int c = 0; c < nchunk; ++c) {
                const int j = c * 64 + lane;
                if (j < np && s.uniq[(int64_t)v * n + j]) {
                    const double r = v2f[(int64_t)e * n + j];
                    lsum += r; lmax = fmax(lmax, r); ++lcnt;
                }
            }
            const double mean = wave_sum(lsum) / (double)wave_sum_i(lcnt);
            const double mx = wave_max(lmax);
            const double shift = (mx - mean > s.max_log_value) ? mx - s.max_log_value : mean;
            const int64_t halo = HALO ? s.halo_off[e] : -1;
            for (int c = 0; c < nchunk; ++c) {
                const int j = c * 64 + lane;
                if (j < np) {
                    const double val = v2f[(int64_t)e * n + j] - shift;
                    v2f[(int64_t)e * n + j] = val;
                    if (HALO && halo >= 0) s.halo_buf[halo + j] = val;
                }
            }
        }
    }
}

// Template variables of relational models touch hundreds or thousands of factors (a topic of the paper-popularity model: one
// factor per paper).  One wavefront walking such a row is a chain of a few hundred dependent row loads, and the launch lasts as
// long as its longest chain.  Here a workgroup shares the row: wavefront w takes the entries k = w (mod 4), four row loads in
// flight each; the four partial totals are added in wavefront order (fixed, so the result is deterministic; it differs from the
// one-wave sum by rounding, like any other summation order), then every wavefront emits the messages of its own entries.
template <bool HALO>
__global__ void __launch_bounds__(BLOCK) pbp_v2f_hub_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                           double* __restrict__ v2f) {
    __shared__ double part[BLOCK / WAVE][WAVE];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int v = s.v2f_hub[blockIdx.x];
    const int n = s.n, S = s.n + s.T;
    const int np = s.np[v];                                 // <= 64 (the caller's list)
    const int lo = g.var_ptr[v], deg = g.var_ptr[v + 1] - lo;
    const bool lifted = g.edge_count != nullptr;
    const bool valid = lane < np;
    constexpr int NW = BLOCK / WAVE;
    double acc = 0.0;
    int k = wid;
    for (; k + 3 * NW < deg; k += 4 * NW) {
        int e[4];
        double m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = g.var_edge[lo + k + u * NW];
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = valid ? f2v[(int64_t)e[u] * S + lane] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += lifted ? m[u] * g.edge_count[e[u]] : m[u];
    }
    for (; k < deg; k += NW) {
        const int e = g.var_edge[lo + k];
        const double m = valid ? f2v[(int64_t)e * S + lane] : 0.0;
        acc += lifted ? m * g.edge_count[e] : m;
    }
    part[wid][lane] = acc;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += part[w][lane];
    const int d = g.var_dom[v];
    const double mu = s.q[2 * v], sd = sqrt_pos(s.q[2 * v + 1]), rsd = rcp_newton(sd);
    const double log_norm = log_pos(2.506628274631 * sd);
    const double x = valid ? s.particles[(int64_t)v * n + lane] : 0.0;
    const bool uq = valid && s.uniq[(int64_t)v * n + lane];
    const double logw = valid ? log_importance(g, s, v, d, x, mu, rsd, log_norm) : 0.0;
    const double rcnt = rcp_newton((double)__builtin_popcountll(__ballot(uq)));
    for (k = wid; k < deg; k += NW) {
        const int e = g.var_edge[lo + k];
        const double m = valid ? f2v[(int64_t)e * S + lane] : 0.0;      // second touch: out of L2
        const double res = (total - m) + logw;
        const double mean = wave_sum(uq ? res : 0.0) * rcnt;
        double shift = mean;
        if (__ballot(uq && (res - mean > s.max_log_value))) shift = wave_max(uq ? res : -__builtin_huge_val()) - s.max_log_value;
        if (valid) {
            v2f[(int64_t)e * n + lane] = res - shift;
            const int64_t halo = HALO ? s.halo_off[e] : -1;         // (owner-computes shards: the copy for the send buffer)
            if (HALO && halo >= 0) s.halo_buf[halo + lane] = res - shift;
        }
    }
}

// Variables with at most four particles -- the binary variables of a hybrid model, the boolean atoms of an MLN -- would
// leave 60 of a wavefront's lanes idle in the kernel above, and every such wave still costs its chain of dependent loads.
// Here a wavefront serves sixteen of them, four lanes each (lane & 3 = particle); the balance step's mean and max are quad
// reductions.  Same expressions per message as above (total minus own; mean over the distinct particles; max - 700 rule).
template <bool HALO>
__global__ void __launch_bounds__(BLOCK) pbp_v2f_narrow_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                              double* __restrict__ v2f) {
    const int lane = threadIdx.x & 63;
    const int64_t slot = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * 16 + (lane >> 2);
    const int j = lane & 3;
    const int v = slot < s.n_v2f_narrow ? s.v2f_narrow[slot] : -1;
    const int n = s.n, S = s.n + s.T;
    const bool on = v >= 0 && is_hidden(g.var_value[v]);
    const int np = on ? s.np[v] : 0;
    const bool valid = j < np;
    const int lo = on ? g.var_ptr[v] : 0, deg = on ? g.var_ptr[v + 1] - lo : 0;
    const bool lifted = g.edge_count != nullptr;
    double total = 0.0;
    for (int k = 0; k < deg; ++k) {
        const int e = g.var_edge[lo + k];
        const double m = valid ? f2v[(int64_t)e * S + j] : 0.0;
        total += lifted ? m * g.edge_count[e] : m;
    }
    double logw = 0.0;
    bool uq = false;
    if (valid) {
        const int d = g.var_dom[v];
        const double sd = sqrt_pos(s.q[2 * v + 1]);
        logw = log_importance(g, s, v, d, s.particles[(int64_t)v * n + j], s.q[2 * v], rcp_newton(sd), log_pos(2.506628274631 * sd));
        uq = s.uniq[(int64_t)v * n + j] != 0;
    }
    auto quad_sum = [](double x) { x += dpp_move<0xb1>(x); return x + dpp_move<0x4e>(x); };
    auto quad_max = [](double x) { x = fmax(x, dpp_move<0xb1>(x)); return fmax(x, dpp_move<0x4e>(x)); };
    const double rcnt = rcp_newton(fmax(quad_sum(uq ? 1.0 : 0.0), 1.0));     // (the kernel above multiplies by the same reciprocal)
    // every lane of the wave runs the longest row of its sixteen variables (the reductions are wave-wide instructions)
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
    for (int k = 0; k < maxdeg; ++k) {
        const bool live = k < deg;
        const int e = live ? g.var_edge[lo + k] : 0;
        const double m = (live && valid) ? f2v[(int64_t)e * S + j] : 0.0;
        const double res = (total - m) + logw;
        const double mean = quad_sum(uq ? res : 0.0) * rcnt;
        const double mx = quad_max(uq ? res : -__builtin_huge_val());
        const double shift = (mx - mean > s.max_log_value) ? mx - s.max_log_value : mean;
        if (live && valid) {
            v2f[(int64_t)e * n + j] = res - shift;
            const int64_t halo = HALO ? s.halo_off[e] : -1;
            if (HALO && halo >= 0) s.halo_buf[halo + j] = res - shift;
        }
    }
}

// The same for variables with 5-16 / 17-32 particles (the particle counts of the reference's demos): a lane group of W = 16 / 32
// lanes per variable, four / two variables per wavefront; the balance step's mean and max are the row / half-wave DPP reductions,
// whose steps inside a group are the ones the one-variable kernel's wave reduction performs on those lanes -- same bits.
template <int W, bool HALO>
__global__ void __launch_bounds__(BLOCK) pbp_v2f_packed_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                              double* __restrict__ v2f, const int32_t* __restrict__ list, int count) {
    constexpr int G = WAVE / W;
    const int lane = threadIdx.x & 63;
    const int64_t slot = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * G + lane / W;
    const int j = lane % W;
    const int v = slot < count ? list[slot] : -1;
    const int n = s.n, S = s.n + s.T;
    const bool on = v >= 0 && is_hidden(g.var_value[v]);
    const int np = on ? s.np[v] : 0;
    const bool valid = j < np;
    const int lo = on ? g.var_ptr[v] : 0, deg = on ? g.var_ptr[v + 1] - lo : 0;
    const bool lifted = g.edge_count != nullptr;
    double total = 0.0;
    for (int k = 0; k < deg; ++k) {
        const int e = g.var_edge[lo + k];
        const double m = valid ? f2v[(int64_t)e * S + j] : 0.0;
        total += lifted ? m * g.edge_count[e] : m;
    }
    double logw = 0.0;
    bool uq = false;
    if (valid) {
        const int d = g.var_dom[v];
        const double sd = sqrt_pos(s.q[2 * v + 1]);
        logw = log_importance(g, s, v, d, s.particles[(int64_t)v * n + j], s.q[2 * v], rcp_newton(sd), log_pos(2.506628274631 * sd));
        uq = s.uniq[(int64_t)v * n + j] != 0;
    }
    // (a row reduction leaves every lane with the row's sum in ITS OWN order of additions; the one-variable kernel's wave reduction
    // ends with lane 15's, so lane 15's is the one every lane of the group takes: row_newbcast:15)
    auto group_sum = [&](double x) { return W == 16 ? dpp_move<0x15F>(dpp_row_reduce(x, SumOp())) : dpp_half_reduce(x, SumOp(), lane); };
    auto group_max = [&](double x) { return W == 16 ? dpp_row_reduce(x, MaxOp()) : dpp_half_reduce(x, MaxOp(), lane); };
    const uint64_t mine = (W == 16 ? 0xffffull : 0xffffffffull) << (lane / W * W);
    const double rcnt = rcp_newton(fmax((double)__builtin_popcountll(__ballot(uq) & mine), 1.0));
    // every lane of the wave runs the longest row of its variables (the reductions are wave-wide instructions)
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off >= W; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
    for (int k = 0; k < maxdeg; ++k) {
        const bool live = k < deg;
        const int e = live ? g.var_edge[lo + k] : 0;
        const double m = (live && valid) ? f2v[(int64_t)e * S + j] : 0.0;
        const double res = (total - m) + logw;
        const double mean = group_sum(uq ? res : 0.0) * rcnt;
        double shift = mean;
        // max - mean > max_log_value  <=>  some distinct particle exceeds mean + max_log_value (as in pbp_v2f_kernel)
        if (__ballot(uq && (res - mean > s.max_log_value))) {
            const double mx = group_max(uq ? res : -__builtin_huge_val());
            if (mx - mean > s.max_log_value) shift = mx - s.max_log_value;
        }
        if (live && valid) {
            v2f[(int64_t)e * n + j] = res - shift;
            const int64_t halo = HALO ? s.halo_off[e] : -1;
            if (HALO && halo >= 0) s.halo_buf[halo + j] = res - shift;
        }
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int state_index(const lhvi_graph_t& g, int v, double x) {
    const int d = g.var_dom[v];
    if (g.dom_cont[d]) return 0;
    for (int i = g.dom_ptr[d]; i < g.dom_ptr[d + 1]; ++i)
        if (g.dom_val[i] == x) return i - g.dom_ptr[d];
    return (int)x;
}

// message_f_to_rv(x, f, rv, sample) for any arity / potential kind (EPBP:176-194; HLBP:193-215):
// sequential mixed-radix walk over the joint particles of the other arguments, last argument fastest.
// INTERP = false: the build without the formula interpreter (lhvi_pots_t.interpreted == 0)
template <bool INTERP = true>
__device__ double f2v_point_generic(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_pbp_t& s,
                                    const double* __restrict__ v2f, const double* __restrict__ partner_particles,
                                    int e, double x, int xi) {
    const int n = s.n;
    const int f = g.edge_fac[e], base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base;
    const int pos = e - base, tv = g.edge_var[e];
    const int pot = g.fac_pot[f], kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    int cnt[LHVI_MAX_ARITY], var[LHVI_MAX_ARITY], ce[LHVI_MAX_ARITY], it[LHVI_MAX_ARITY], ix[LHVI_MAX_ARITY];
    bool fixed[LHVI_MAX_ARITY], withmsg[LHVI_MAX_ARITY];
    double xs[LHVI_MAX_ARITY];
#pragma unroll
    for (int a = 0; a < LHVI_MAX_ARITY; ++a) {
        cnt[a] = 1; var[a] = 0; ce[a] = 0; it[a] = 0; ix[a] = 0; fixed[a] = true; withmsg[a] = false; xs[a] = 0.0;
        if (a < arity) {
            var[a] = g.edge_var[base + a];
            ce[a] = canon(g.edge_canon, base + a);
            if (a == pos) { xs[a] = x; ix[a] = xi; }
            else {
                const double val = g.var_value[var[a]];
                if (is_hidden(val)) { cnt[a] = s.np[var[a]]; fixed[a] = false; withmsg[a] = var[a] != tv; }
                else { xs[a] = val; ix[a] = state_index(g, var[a], val); }
            }
        }
    }
    double res = 0.0;
    for (;;) {
        double m = 0.0;
#pragma unroll
        for (int a = 0; a < LHVI_MAX_ARITY; ++a) {
            if (a < arity && !fixed[a]) {
                xs[a] = partner_particles[(int64_t)var[a] * n + it[a]];
                ix[a] = it[a];
                if (withmsg[a]) m += v2f[(int64_t)ce[a] * n + it[a]];
            }
        }
        res += pot_times_exp<INTERP>(kind, par, xs, ix, m);
        int a = arity - 1;
        while (a >= 0) {
            if (!fixed[a] && ++it[a] < cnt[a]) break;
            it[a] = 0;
            --a;
        }
        if (a < 0) break;
    }
    return res > 0.0 ? log(res) : -700.0;
}

// Edge classes of the f -> v half sweep.  FAST edges have a term of the form
//     log phi + m_j = a_j + b_j * X1 + k_j * X2 + C        (j = partner particle, X1/X2/C = per output point)
//   (1) continuous target, log phi quadratic in it (Gaussian / Quadratic / LinearGaussian / XY / HybridQuadratic with
//       the discrete partner): X1 = x, X2 = x^2, C = 0, (a, b, k) = coefficients given the partner particle + message;
//   (2) discrete target of a HybridQuadratic(1 disc, 1 cont): X1 = b_d, X2 = A_d, C = c_d, (a, b, k) = (m_j, y_j, y_j^2).
// Conditionally quadratic MLN formulas (cq.hpp; with LHVI_PBP_CQ set): resolved against the evidence they are edges of
// class 1 / 2 again (served by the heavy / light kernels through descriptors that carry the resolved coefficients) or,
// with a hidden discrete and a hidden continuous partner or two hidden continuous partners, edges of class 4 (EDGE_CQ).
// Everything else (tables, other MLN formulas, arity != 2) takes the GENERIC kernel.
enum { EDGE_SKIP = 0, EDGE_FAST_CONT = 1, EDGE_FAST_DISC = 2, EDGE_GENERIC = 3, EDGE_CQ = 4 };

// s.np / s.n / s.flags are all the classification reads of `s`
__device__ __forceinline__ int classify_edge(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_pbp_t& s, int e) {
    if (canon(g.edge_canon, e) != e) return EDGE_SKIP;
    const int tv = g.edge_var[e];
    if (!is_hidden(g.var_value[tv])) return EDGE_SKIP;
    const int f = g.edge_fac[e], base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base, pos = e - base;
    if ((s.flags & LHVI_PBP_CQ) && s.np && pots.kind[g.fac_pot[f]] == LHVI_POT_MLN) {
        CqInfo ci;
        const int route = cq_analyze(g, pots, s.np, s.n, e, ci);
        if (route == CQ_NONE) return EDGE_GENERIC;
        const int dom = g.var_dom[tv];
        const int T = g.dom_cont[dom] ? g.dom_ptr[dom + 1] - g.dom_ptr[dom] : 0;
        if (s.np[tv] + T > 128 || s.np[tv] > 64) return EDGE_GENERIC;      // the descriptor kernels' output-point limits
        if (route == CQ_MIX || route == CQ_JOINT) return EDGE_CQ;
        return route == CQ_LIGHT2 ? EDGE_FAST_DISC : EDGE_FAST_CONT;
    }
    if (arity != 2) return EDGE_GENERIC;
    const int pv = g.edge_var[base + (1 - pos)];
    if (pv == tv) return EDGE_GENERIC;
    const int pot = g.fac_pot[f], kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    const bool tcont = g.dom_cont[g.var_dom[tv]] != 0, pcont = g.dom_cont[g.var_dom[pv]] != 0;
    if (kind == LHVI_POT_HYBRID_QUADRATIC) {
        if ((int)par[0] != 1 || (int)par[1] != 1) return EDGE_GENERIC;
        if (pos == 1 && tcont && !pcont) return EDGE_FAST_CONT;
        if (pos == 0 && !tcont && pcont) return EDGE_FAST_DISC;
        return EDGE_GENERIC;
    }
    Quad2 q;
    if (tcont && quad2_of(kind, par, 0, q)) return EDGE_FAST_CONT;
    return EDGE_GENERIC;
}

// Static description of a FAST edge, built once per run (lhvi_pbp_describe) so that the persistent kernel fetches
// everything it needs about an edge with two scalar loads instead of a chain of dependent gathers.
struct FastDesc {
    int32_t e, tv, pv, pce;        // edge, target variable, partner variable, partner's canonical edge
    int32_t cls, pos, kind, nj;    // edge class, target position, potential kind, partner particle count (1 = observed)
    int32_t np, T, gb, par_off;    // target particle count, target grid size, grid base in dom_val, offset into pots.param
    double pval;                   // partner evidence value (NaN = hidden)
    int32_t pad[2];                // [0] light-kernel type, [1] 1 = the target's integral points are a uniform grid
    // class 1 with a constant x^2 coefficient (kind != HYBRID_QUADRATIC): the potential resolved for this edge's target
    // position, log phi(x, y) = kx x^2 + (ay y + by) y + c + (axy y + bx) x with x = target, y = partner
    double ay, by, c, axy, bx, kx;
    double pad2[2];                // uniform grid: first point and spacing (x_t = x0 + t h)
};
static_assert(sizeof(FastDesc) == LHVI_PBP_DESC_BYTES, "FastDesc is part of the ABI (LHVI_PBP_DESC_BYTES)");

__device__ __forceinline__ FastDesc make_fast_desc(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_pbp_t& s, int e) {
    FastDesc d;
    d.e = e;
    d.cls = classify_edge(g, pots, s, e);
    d.tv = g.edge_var[e];
    const int f = g.edge_fac[e], base = g.fac_ptr[f];
    d.pos = e - base;
    const int pot = g.fac_pot[f];
    d.kind = pots.kind[pot];
    d.par_off = pots.off[pot];
    const int dom = g.var_dom[d.tv];
    d.np = s.np[d.tv];
    d.gb = g.dom_ptr[dom];
    d.T = (d.cls == EDGE_FAST_CONT) ? g.dom_ptr[dom + 1] - d.gb : 0;
    d.pad[0] = d.pad[1] = 0;
    d.ay = d.by = d.c = d.axy = d.bx = d.kx = d.pad2[0] = d.pad2[1] = 0.0;
    d.pv = 0; d.pce = 0; d.pval = 0.0; d.nj = 1;
    const bool cq = d.kind == LHVI_POT_MLN;               // (an MLN edge is on the fast list only through its conditional-quadratic view)
    CqInfo ci;
    if (cq) {
        cq_analyze(g, pots, s.np, s.n, e, ci);
        // the descriptor of the kernel that serves the resolved edge: heavy (word 6 != HYBRID_QUADRATIC, word 14 == 0) or
        // light (word 6 == HYBRID_QUADRATIC, word 14 = type); partner = the one hidden partner left, if any
        if (ci.route == CQ_HEAVY) {
            d.kind = LHVI_POT_QUADRATIC;
            d.pv = ci.yv; d.pce = ci.yce; d.nj = ci.ny; d.pval = ci.yval;
            d.ay = ci.ay[0]; d.by = ci.by[0]; d.c = ci.c[0]; d.axy = ci.axy[0]; d.bx = ci.bx[0]; d.kx = ci.kx[0];
        } else if (ci.route == CQ_LIGHT1 || ci.route == CQ_LIGHT2) {
            d.kind = LHVI_POT_HYBRID_QUADRATIC;
            const bool t1 = ci.route == CQ_LIGHT1;
            d.pv = t1 ? ci.zv : ci.yv; d.pce = t1 ? ci.zce : ci.yce; d.nj = t1 ? ci.nz : ci.ny;
            d.pval = __builtin_nan("");
            d.ay = ci.kx[0]; d.by = ci.bx[0]; d.c = ci.c[0];
            if (ci.S > 1) { d.axy = ci.kx[1]; d.bx = ci.bx[1]; d.kx = ci.c[1]; }
            d.pad[0] = t1 ? 1 : 2;
        }
    } else {
        const int pe = base + (1 - d.pos);
        d.pv = g.edge_var[pe];
        d.pce = canon(g.edge_canon, pe);
        d.pval = g.var_value[d.pv];
        d.nj = is_hidden(d.pval) ? s.np[d.pv] : 1;
    }
    // uniform integral-point grid (the reference's Domain default is a linspace): x_t = x0 + t h to a few ulp.  The heavy
    // kernel then tabulates exp(a + b x_t) along t by multiplication instead of one exponential per point
    if (d.cls == EDGE_FAST_CONT && d.T >= 2) {
        const double x0 = g.dom_val[d.gb], xl = g.dom_val[d.gb + d.T - 1];
        const double h = (xl - x0) / (double)(d.T - 1);
        const double tol = 1.8e-15 * fmax(fabs(x0), fabs(xl));
        bool uniform = h > 0.0 && h < __builtin_huge_val();
        for (int t = 0; t < d.T && uniform; ++t) uniform = fabs(g.dom_val[d.gb + t] - fma((double)t, h, x0)) <= tol;
        if (uniform) { d.pad[1] = 1; d.pad2[0] = x0; d.pad2[1] = h; }
    }
    Quad2 q;
    if (!cq && d.cls == EDGE_FAST_CONT && d.kind != LHVI_POT_HYBRID_QUADRATIC && quad2_of(d.kind, pots.param + d.par_off, 0, q)) {
        if (d.pos == 0) { d.ay = q.a11; d.by = q.b1; d.axy = q.axy; d.bx = q.b0; d.kx = q.a00; }
        else            { d.ay = q.a00; d.by = q.b0; d.axy = q.axy; d.bx = q.b1; d.kx = q.a11; }
        d.c = q.c;
    }
    // HybridQuadratic(1 discrete, 1 continuous) edges whose discrete side has at most two points (binary variables,
    // or an observed discrete partner): log phi = A_s x^2 + b_s x + c_s per discrete point s, resolved here as
    // (A_0, b_0, c_0, A_1, b_1, c_1) for the light kernel.  pad[0]: 1 = continuous target / discrete partner,
    // 2 = discrete target / continuous partner, 0 = not a light edge
    if (!cq && d.kind == LHVI_POT_HYBRID_QUADRATIC && (d.cls == EDGE_FAST_CONT || d.cls == EDGE_FAST_DISC)) {
        const double* par = pots.param + d.par_off;
        const int nst = (int)par[2];
        const bool cont_target = d.cls == EDGE_FAST_CONT;
        const int dv = cont_target ? d.pv : d.tv;                 // the discrete variable of the factor
        const int npt = cont_target ? d.nj : d.np;
        const bool ok = npt <= 2 && (cont_target ? d.np + d.T <= 128 : (d.nj <= 64 && d.T == 0));
        if (ok) {
            double* co = &d.ay;
            for (int k = 0; k < 2; ++k) {
                int st = 0;
                if (k < npt) {
                    const double val = is_hidden(g.var_value[dv]) ? g.dom_val[g.dom_ptr[g.var_dom[dv]] + k] : g.var_value[dv];
                    st = (int)val;
                }
                st = st < 0 ? 0 : (st >= nst ? nst - 1 : st);
                co[3 * k] = par[3 + st]; co[3 * k + 1] = par[3 + nst + st]; co[3 * k + 2] = par[3 + 2 * nst + st];
            }
            d.pad[0] = cont_target ? 1 : 2;
        }
    }
    return d;
}

__global__ void __launch_bounds__(BLOCK) pbp_describe_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                            const int32_t* __restrict__ edges, int count,
                                                            FastDesc* __restrict__ out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < count) out[i] = make_fast_desc(g, pots, s, edges[i]);
}

// wave-private LDS hand-off: DS operations of one wavefront execute in order, so only the compiler needs fencing
#define LHVI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// LDS record of one partner particle: (a_j, b_j) as 16 bytes (one ds_read_b128 per term); the x^2 coefficient is
//   MODE_CONST  the same for every j (continuous x continuous potentials): folded into the per-point constant K*x^2
//   MODE_VARK   per j (HybridQuadratic with a discrete partner, nj = #states): read from a second LDS array
//   MODE_DISC   b_j^2 (discrete target of a HybridQuadratic): t = a_j + b_j (X1 + b_j X2) + C
enum { MODE_CONST = 0, MODE_VARK = 1, MODE_DISC = 2 };
struct AB { double a, b; };

template <int MODE>
__device__ __forceinline__ double fast_term(double acc, const AB* __restrict__ sh, const double* __restrict__ shk,
                                            const double* __restrict__ tab, int j, double X1, double X2, double magic) {
    const AB r = sh[j];
    double t;
    if (MODE == MODE_CONST) t = fma(r.b, X1, r.a);                     // + C = k * x^2 through `magic`
    else if (MODE == MODE_VARK) t = fma(shk[j], X2, fma(r.b, X1, r.a));
    else t = fma(r.b, fma(r.b, X2, X1), r.a);
    return exp_accumulate(acc, t, magic, tab);
}

// every lane walks `jn` consecutive records starting at its own base (full rounds: the same base for all lanes; the
// last partial round: one base per lane group) -> scalar loop control, LDS addresses with immediate offsets, two
// (or four) independent exp chains per iteration
template <int MODE, int UNROLL = 2>
__device__ __forceinline__ double fast_accumulate_uniform(const AB* __restrict__ sh, const double* __restrict__ shk,
                                                          const double* __restrict__ tab, int jn_, double X1, double X2, double C) {
    const int jn = __builtin_amdgcn_readfirstlane(jn_);
    const ExpShift sft = exp_shift(C);
    double acc0 = 0.0, acc1 = 0.0;
    int j = 0;
    if (UNROLL == 4) {                  // four chains: fewer LDS round trips per term; worth its registers in the heavy kernel only
        double acc2 = 0.0, acc3 = 0.0;
        for (; j + 4 <= jn; j += 4) {
#ifdef LHVI_HEAVY_STAGED
            if (MODE == MODE_CONST) {
                const AB r0 = sh[j], r1 = sh[j + 1], r2 = sh[j + 2], r3 = sh[j + 3];
                exp_accumulate4(acc0, acc1, acc2, acc3, fma(r0.b, X1, r0.a), fma(r1.b, X1, r1.a), fma(r2.b, X1, r2.a),
                                fma(r3.b, X1, r3.a), sft.magic, tab);
                continue;
            }
#endif
            acc0 = fast_term<MODE>(acc0, sh, shk, tab, j, X1, X2, sft.magic);
            acc1 = fast_term<MODE>(acc1, sh, shk, tab, j + 1, X1, X2, sft.magic);
            acc2 = fast_term<MODE>(acc2, sh, shk, tab, j + 2, X1, X2, sft.magic);
            acc3 = fast_term<MODE>(acc3, sh, shk, tab, j + 3, X1, X2, sft.magic);
        }
        acc0 += acc2; acc1 += acc3;
    }
    for (; j + 2 <= jn; j += 2) {
        acc0 = fast_term<MODE>(acc0, sh, shk, tab, j, X1, X2, sft.magic);
        acc1 = fast_term<MODE>(acc1, sh, shk, tab, j + 1, X1, X2, sft.magic);
    }
    if (j < jn) acc0 = fast_term<MODE>(acc0, sh, shk, tab, j, X1, X2, sft.magic);
    return (acc0 + acc1) * sft.scale;
}

#define LHVI_PIN1(a) do { if (PIN) asm volatile("" : "+v"(a)); } while (0)
#define LHVI_PIN2(a, b) do { if (PIN) asm volatile("" : "+v"(a), "+v"(b)); } while (0)
// The MODE_CONST loop on records pre-divided by the table step (a_j / step, b_j / step): exp_accumulate_floor, one fp64 operation
// less per term (fastmath.hpp).  The fp64 rounding mode is round-down between the two mode writes.
template <int UNROLL = 4, bool PIN = false>
__device__ __forceinline__ double fast_accumulate_floor(const AB* __restrict__ sh, const double* __restrict__ tab, int jn_, double X1,
                                                        double C) {
    const int jn = __builtin_amdgcn_readfirstlane(jn_);
    // The compiler may move plain fp64 arithmetic across the two mode writes (they are volatile asm, the arithmetic is not): the
    // shift's constants of an unrolled NEXT call computed inside this call's round-down window, the closing additions after it --
    // differently in different instantiations, so the last bit of a sum (a few percent of a sum that is denormal) depends on the
    // kernel it was formed in.  PIN: empty volatile asm statements pin both ends; what is computed under which rounding mode is then
    // the same wherever the call is inlined.  The few-particle kernels pin (their instantiations -- lane groups of 8 to 32 lanes,
    // one or two particles per lane -- serve the same edge depending on lhvi_pbp_t.n and must give it the same bits); the heavy
    // kernel, a single instantiation, does not: the pins cost it 0.7 % (scripts/diag/pin_ab.sh).
    LHVI_PIN1(C);
    ExpShiftFloor sft = exp_shift_floor(C);
    LHVI_PIN2(sft.magic, sft.scale);
    double acc0 = 0.0, acc1 = 0.0;
    int j = 0;
    round_down_on();
    if (UNROLL == 4) {
        double acc2 = 0.0, acc3 = 0.0;
        for (; j + 4 <= jn; j += 4) {
            const AB r0 = sh[j], r1 = sh[j + 1], r2 = sh[j + 2], r3 = sh[j + 3];
            acc0 = exp_accumulate_floor(acc0, fma(r0.b, X1, r0.a), sft.magic, tab);
            acc1 = exp_accumulate_floor(acc1, fma(r1.b, X1, r1.a), sft.magic, tab);
            acc2 = exp_accumulate_floor(acc2, fma(r2.b, X1, r2.a), sft.magic, tab);
            acc3 = exp_accumulate_floor(acc3, fma(r3.b, X1, r3.a), sft.magic, tab);
        }
        // the remainder: term j + k stays with accumulator k and the accumulators are folded at the very end, as if the list were
        // padded with zero terms to a multiple of four -- so a list that IS padded (the few-particle kernel runs every edge of a
        // wavefront to the longest list among them) gives every edge the bits it would get alone, whatever shares its wavefront
        if (j < jn) { const AB r0 = sh[j]; acc0 = exp_accumulate_floor(acc0, fma(r0.b, X1, r0.a), sft.magic, tab); }
        if (j + 1 < jn) { const AB r1 = sh[j + 1]; acc1 = exp_accumulate_floor(acc1, fma(r1.b, X1, r1.a), sft.magic, tab); }
        if (j + 2 < jn) { const AB r2 = sh[j + 2]; acc2 = exp_accumulate_floor(acc2, fma(r2.b, X1, r2.a), sft.magic, tab); }
        acc0 += acc2; acc1 += acc3;
        LHVI_PIN2(acc0, acc1);
        round_down_off();
        return (acc0 + acc1) * sft.scale;
    }
    for (; j + 2 <= jn; j += 2) {
        const AB r0 = sh[j], r1 = sh[j + 1];
        acc0 = exp_accumulate_floor(acc0, fma(r0.b, X1, r0.a), sft.magic, tab);
        acc1 = exp_accumulate_floor(acc1, fma(r1.b, X1, r1.a), sft.magic, tab);
    }
    if (j < jn) { const AB r0 = sh[j]; acc0 = exp_accumulate_floor(acc0, fma(r0.b, X1, r0.a), sft.magic, tab); }
    LHVI_PIN2(acc0, acc1);
    round_down_off();
    return (acc0 + acc1) * sft.scale;
}

// FAST edges: persistent kernel, one wavefront per edge at a time (each wave strides over the work list).
// Partner coefficients are staged in wave-private LDS in tiles of 64; each lane owns one output point per round and
// accumulates sum_j exp(.).  A final partial round splits the partner range over idle lanes and folds the partial
// sums with shuffles, so n + T = 96 points on 64 lanes still keep every lane busy.
// This kernel serves every fast edge the heavy kernel below does not take (per-state x^2 coefficient, discrete target,
// more than 64 partner particles or more than 128 output points): few terms per edge, general in every respect.
__global__ void __launch_bounds__(BLOCK) pbp_f2v_fast_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                            const double* __restrict__ v2f, double* __restrict__ f2v,
                                                            const FastDesc* __restrict__ descs,
                                                            const double* __restrict__ param) {
    __shared__ AB sh_all[BLOCK / WAVE][WAVE];
    __shared__ double shk_all[BLOCK / WAVE][WAVE];
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    AB* sh = sh_all[wid];
    double* shk = shk_all[wid];
    const int nitems = s.fast_edges ? s.n_fast : g.E;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    const int n = s.n, S = s.n + s.T;
    for (int item = blockIdx.x * (BLOCK / WAVE) + wid; item < nitems; item += nwaves) {
        FastDesc d;
        if (descs) d = descs[item];                       // wave-uniform address: scalar loads
        else d = make_fast_desc(g, pots, s, s.fast_edges ? s.fast_edges[item] : item);
        if (d.cls != EDGE_FAST_CONT && d.cls != EDGE_FAST_DISC) continue;
        const double* __restrict__ par = param + d.par_off;
        const int np = d.np, npts = d.np + d.T, nj = d.nj;
        const bool partner_hidden = is_hidden(d.pval);
        double* out = f2v + (int64_t)d.e * S;

        // issue every global load of this edge up front (one exposed round trip instead of one per phase):
        // partner particle + message of the first tile, and this lane's output point of the first two rounds
        double y0 = d.pval, m0 = 0.0;
        if (partner_hidden && lane < nj) { y0 = s.old_particles[(int64_t)d.pv * n + lane]; m0 = v2f[(int64_t)d.pce * n + lane]; }
        double xr[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rem = npts - 64 * r;
            if (rem > 0) {
                int lw = 6;
                if (rem <= 32) { lw = 0; while ((1 << lw) < rem) ++lw; }
                const int pl = lane & ((1 << lw) - 1), pp = 64 * r + pl;
                if (pl < rem) xr[r] = pp < np ? s.particles[(int64_t)d.tv * n + pp] : g.dom_val[d.gb + pp - np];
            }
        }

        // x^2 coefficient: constant per edge unless the partner is the discrete argument of a HybridQuadratic
        const int mode = d.cls == EDGE_FAST_DISC ? MODE_DISC : (d.kind == LHVI_POT_HYBRID_QUADRATIC ? MODE_VARK : MODE_CONST);
        double kconst = 0.0;
        if (mode == MODE_CONST) { Quad2 q; quad2_of(d.kind, par, 0, q); kconst = d.pos == 0 ? q.a00 : q.a11; }
        auto stage = [&](int j0, int jn, double y, double m) {
            LHVI_WAVE_SYNC();
            {
                // records past the tile are padded with a term that underflows to exactly 0 (exp(-800)), so that every
                // lane group can run the same uniform loop over ceil(jn / split) records without per-lane bounds
                AB r;
                r.a = PAD_LOG_TERM; r.b = 0.0;
                double kk = 0.0;
                if (lane < jn) {
                    if (d.cls == EDGE_FAST_CONT) {
                        Quad2 q;
                        quad2_of(d.kind, par, (d.kind == LHVI_POT_HYBRID_QUADRATIC) ? (int)y : 0, q);
                        if (d.pos == 0) { r.a = (q.a11 * y + q.b1) * y + q.c + m; r.b = q.axy * y + q.b0; kk = q.a00; }
                        else            { r.a = (q.a00 * y + q.b0) * y + q.c + m; r.b = q.axy * y + q.b1; kk = q.a11; }
                    } else { r.a = m; r.b = y; }
                }
                if (mode == MODE_CONST) { r.a *= LHVI_EXP_INV_STEP; r.b *= LHVI_EXP_INV_STEP; }     // floor form: records in units of the table step
                sh[lane] = r;
                if (mode == MODE_VARK) shk[lane] = kk;
            }
            LHVI_WAVE_SYNC();
        };
        const bool single_tile = nj <= 64;
        if (single_tile) stage(0, nj, y0, m0);             // staged once per edge, reused by every round

        for (int p0 = 0; p0 < npts; p0 += 64) {
            const int rem = npts - p0;
            int lw = 6;                                   // log2(width): width = 64 for full rounds, else next pow2 >= rem
            if (rem <= 32) { lw = 0; while ((1 << lw) < rem) ++lw; }
            const int width = 1 << lw, split = 64 >> lw, sub = lane >> lw, pl = lane & (width - 1);
            const int p = p0 + pl;
            const bool valid = pl < rem;
            double xv = 0.0;
            if (p0 == 0) xv = xr[0];
            else if (p0 == 64) xv = xr[1];
            else if (valid) xv = p < np ? s.particles[(int64_t)d.tv * n + p] : g.dom_val[d.gb + p - np];
            double X1 = 0.0, X2 = 0.0, C = 0.0;
            if (valid) {
                if (d.cls == EDGE_FAST_CONT) { X1 = xv; X2 = xv * xv; C = kconst * X2; }
                else {
                    const int nst = (int)par[2];
                    const int st = (int)xv;                                  // HybridQuadratic indexes by the state value
                    X2 = par[3 + st]; X1 = par[3 + nst + st]; C = par[3 + 2 * nst + st];
                }
            }
            double acc = 0.0;
            for (int j0 = 0; j0 < ((s.flags & LHVI_PBP_SKIP_TERMS) ? 0 : nj); j0 += 64) {     // flag 16: tuning aid, skips the term loop
                const int jn = min(64, nj - j0);
                if (!single_tile) {
                    double y = d.pval, m = 0.0;
                    if (partner_hidden && lane < jn) { y = s.old_particles[(int64_t)d.pv * n + j0 + lane]; m = v2f[(int64_t)d.pce * n + j0 + lane]; }
                    stage(j0, jn, y, m);
                }
                // lane group `sub` owns records [sub * chunk, (sub + 1) * chunk); split * chunk <= 64 always
                const int chunk = (jn + split - 1) >> (6 - lw);
                const AB* base = sh + sub * chunk;
                const double* basek = shk + sub * chunk;
                if (mode == MODE_CONST) acc += fast_accumulate_floor<2>(base, sh_tab, chunk, X1, C);
                else if (mode == MODE_VARK) acc += fast_accumulate_uniform<MODE_VARK>(base, basek, sh_tab, chunk, X1, X2, C);
                else acc += fast_accumulate_uniform<MODE_DISC>(base, basek, sh_tab, chunk, X1, X2, C);
            }
            for (int off = width; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
            if (valid && sub == 0) out[p < np ? p : n + (p - np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
        }
    }
}

// HEAVY edges = the bulk of the work: continuous target, constant x^2 coefficient (continuous x continuous
// quadratic-family potential, or an observed partner), nj <= 64 partner particles, at most 128 output points.  Same
// arithmetic as the general kernel above in MODE_CONST, but (a) only this mode, which fits 7 waves per SIMD, and
// (b) software-pipelined: while edge k is in its term loops the loads of edge k+1 are already in flight (its descriptor
// was fetched one edge earlier still), so a wave never sits out a memory round trip between two term loops.
// Work distribution of the persistent f2v kernels.  With a ticket word (lhvi_pbp_t.f2v_ticket) the waves claim chunks of
// WORK_CHUNK consecutive list entries with one atomic add each -- the next chunk is claimed when a chunk is entered and
// its index read (v_readfirstlane of the returned value) only when the chunk is used up, so the atomic's round trip hides
// behind WORK_CHUNK edges -- and a workgroup that reaches its CU late (another kernel's workgroups held the slot) simply
// finds less left to claim.  Without a ticket every wave strides over the list: a late workgroup then still owes its
// full static share.
#ifndef LHVI_PBP_TAIL_PER_WAVE
#define LHVI_PBP_TAIL_PER_WAVE 0            // entries per wave of a part's tail zone (claimed one at a time); 0: chunks to the end.
                                            // Measured on the 8-rank rehearsal (profiles/r05_experiments.md): 0 -> 0.875 ms per heavy launch,
                                            // 2 -> 0.948, 8 -> 1.20: a claim per entry costs more than the shorter tail returns
#endif
#if LHVI_PBP_TAIL_PER_WAVE > 0
template <int WORK_CHUNK, int WAVES_PER_BLOCK = BLOCK / WAVE>
struct WorkCursor {
    uint32_t* ticket;       // nullptr: static striding
    int item, limit, left, stride, pending, lo, body, mid;
    // A claim always advances the counter by WORK_CHUNK.  The first `body` positions of a part are real entries, claimed
    // WORK_CHUNK at a time; behind them every claim stands for ONE entry of the part's tail zone [mid, limit): a wave that
    // comes late then owes one entry, not a whole chunk -- a launch ends with the tail of a single entry per wave (what
    // matters when a shard's list gives a wave only a few chunks), for one atomic per entry on ~2 entries per wave only.
    __device__ __forceinline__ int claim(int lane) const {
        int v = 0;
        if (lane == 0) v = (int)atomicAdd(ticket, (uint32_t)WORK_CHUNK);
        return v;                                           // a position, valid in lane 0
    }
    __device__ __forceinline__ int entry_of(int pos) const { return pos < body ? lo + pos : mid + (pos - body) / WORK_CHUNK; }
    __device__ __forceinline__ int chunk_of(int pos) const { return pos < body ? WORK_CHUNK - 1 : 0; }       // entries left after the first
    // with tickets the list is cut into one contiguous range per XCD (workgroup i runs on XCD i mod 8, and each XCD has
    // its own L2: its waves then walk one region of the descriptors, particles and messages), each with its own counter
    __device__ __forceinline__ bool start(uint32_t* base, int nitems, int lane) {
        stride = gridDim.x * WAVES_PER_BLOCK;
        left = 0; pending = 0; lo = 0; limit = nitems; body = 0; mid = 0;
        ticket = base;
        if (ticket) {
            const int parts = min((int)gridDim.x, LHVI_PBP_TICKET_COUNTERS);
            const int part = blockIdx.x % parts;
            const int per = ((nitems + parts - 1) / parts + WORK_CHUNK - 1) / WORK_CHUNK * WORK_CHUNK;
            lo = min(part * per, nitems);
            limit = min(lo + per, nitems);
            const int waves = (gridDim.x + parts - 1) / parts * WAVES_PER_BLOCK;          // waves that draw from this counter
            const int tail = min(limit - lo, LHVI_PBP_TAIL_PER_WAVE * waves);
            body = (limit - lo - tail) / WORK_CHUNK * WORK_CHUNK;
            mid = lo + body;
            ticket = base + part;
            const int first = __builtin_amdgcn_readfirstlane(claim(lane));
            item = entry_of(first);
            left = chunk_of(first);
            pending = claim(lane);
        } else {
            item = blockIdx.x * WAVES_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        }
        return item < limit;
    }
    __device__ __forceinline__ int next() const {           // the entry after `item` (>= limit: none)
        if (!ticket) return item + stride;
        return left > 0 ? item + 1 : entry_of(__builtin_amdgcn_readfirstlane(pending));
    }
    __device__ __forceinline__ void advance(int nxt, int lane) {     // move to `nxt` = next()
        if (ticket) {
            if (left > 0) --left;
            else { left = chunk_of(__builtin_amdgcn_readfirstlane(pending)); pending = claim(lane); }
        }
        item = nxt;
    }
};
#else
// (no tail zone: the cursor of rounds 2-4, without the position arithmetic of the form above)
template <int WORK_CHUNK, int WAVES_PER_BLOCK = BLOCK / WAVE>
struct WorkCursor {
    uint32_t* ticket;       // nullptr: static striding
    int item, limit, left, stride, pending, lo;
    __device__ __forceinline__ int claim(int lane) const {
        int v = 0;
        if (lane == 0) v = lo + (int)atomicAdd(ticket, (uint32_t)WORK_CHUNK);
        return v;                                           // valid in lane 0
    }
    // with tickets the list is cut into one contiguous range per XCD (workgroup i runs on XCD i mod 8, and each XCD has
    // its own L2: its waves then walk one region of the descriptors, particles and messages), each with its own counter
    __device__ __forceinline__ bool start(uint32_t* base, int nitems, int lane) {
        stride = gridDim.x * WAVES_PER_BLOCK;
        left = 0; pending = 0; lo = 0; limit = nitems;
        ticket = base;
        if (ticket) {
            const int parts = min((int)gridDim.x, LHVI_PBP_TICKET_COUNTERS);
            const int part = blockIdx.x % parts;
            const int per = ((nitems + parts - 1) / parts + WORK_CHUNK - 1) / WORK_CHUNK * WORK_CHUNK;
            lo = min(part * per, nitems);
            limit = min(lo + per, nitems);
            ticket = base + part;
            item = __builtin_amdgcn_readfirstlane(claim(lane));
            pending = claim(lane);
            left = WORK_CHUNK - 1;
        } else {
            item = blockIdx.x * WAVES_PER_BLOCK + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        }
        return item < limit;
    }
    __device__ __forceinline__ int next() const {           // the entry after `item` (>= limit: none)
        if (!ticket) return item + stride;
        return left > 0 ? item + 1 : __builtin_amdgcn_readfirstlane(pending);
    }
    __device__ __forceinline__ void advance(int nxt, int lane) {     // move to `nxt` = next()
        if (ticket) {
            if (left > 0) --left;
            else { left = WORK_CHUNK - 1; pending = claim(lane); }
        }
        item = nxt;
    }
};
#endif

struct HeavyData { double y, m, x0, x1; };

// launch shape of the heavy kernel (tuning knobs; the defaults are what ships)
#ifndef LHVI_HEAVY_BLOCK
#define LHVI_HEAVY_BLOCK 256
#endif
#ifndef LHVI_HEAVY_WAVES
#define LHVI_HEAVY_WAVES 7
#endif
#ifndef LHVI_HEAVY_UNROLL
#define LHVI_HEAVY_UNROLL 4
#endif
#ifndef LHVI_HEAVY_FLOOR
#define LHVI_HEAVY_FLOOR 1          // term loops in the floor form (exp_accumulate_floor); 0: the round-to-nearest form
#endif
constexpr int HEAVY_BLOCK = LHVI_HEAVY_BLOCK;

// ---- integral points on a uniform grid: sum_j exp(a_j + b_j x_t) for t < 32 with lane = partner particle j ------------
// G_{t+1,j} = G_{t,j} * exp(b_j h) replaces the exponential per (t, j) by one multiplication; the sums over j are then a
// reduce-scatter over the lanes that leaves S_t in the lanes owning t.  One fold per lane-id bit: x is the value kept by
// the lanes whose bit is 0, y by the others, and the result is own + partner's copy of the kept value.  Bits 5 and 4:
// v_permlane32_swap / v_permlane16_swap (gfx950; a swap per dword moves both values, no selects); bits 3 and 2:
// bank-masked DPP row rotations / shifts; bits 1 and 0: quad permutes.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_into(double old, double src) {     // enabled banks take src[permuted], the rest keep old
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fold_bit5(double x, double y) {
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold_bit4(double x, double y) {
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold_bit3(double x, double y) {        // lane ^ 8 = row_ror:8; bit 3 set = banks 2, 3
    return dpp_into<0x128, 0x3>(y, x) + dpp_into<0x128, 0xc>(x, y);
}
__device__ __forceinline__ double fold_bit2(double x, double y) {        // lane ^ 4: row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3
    return dpp_into<0x104, 0x5>(y, x) + dpp_into<0x114, 0xa>(x, y);
}
__device__ __forceinline__ double fold_bit1(double x, double y, bool bit1) {      // lane ^ 2 = quad_perm [2,3,0,1]
    const double send = bit1 ? x : y, keep = bit1 ? y : x;
    return keep + dpp_move<0x4e>(send);
}
__device__ __forceinline__ double fold_bit0(double x) { return x + dpp_move<0xb1>(x); }   // lane ^ 1 = quad_perm [1,0,3,2]

// the point whose sum a lane holds after the six folds of 4 batches of 8 consecutive points
__device__ __forceinline__ int grid_owned_point(int lane) {
    return 8 * (2 * ((lane >> 1) & 1) + ((lane >> 2) & 1)) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);
}

// g = this lane's G at the first of 32 consecutive grid points (advanced by 32 steps on return), q = its ratio
__device__ __forceinline__ double grid_sums32(double& g, double q, int lane) {
    double z[4];
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = g; g *= q; }
        const double u0 = fold_bit4(fold_bit5(v[0], v[1]), fold_bit5(v[2], v[3]));
        const double u1 = fold_bit4(fold_bit5(v[4], v[5]), fold_bit5(v[6], v[7]));
        z[bt] = fold_bit3(u0, u1);
    }
    return fold_bit0(fold_bit1(fold_bit2(z[0], z[1]), fold_bit2(z[2], z[3]), (lane >> 1) & 1));
}
// The same reduce-scatter inside lane groups of W = 32 / 16 lanes (the few-particle kernel: lane = partner particle of its group's
// edge).  W = 32: folds over bits 4..0, every lane ends with the sum of ONE of 32 consecutive points; W = 16: bits 3..0, two points
// per lane (s0: points 0-15, s1: points 16-31 of the batch).  small_grid_point<W>(lane, k) names the point of sum k.
__device__ __forceinline__ double fold_bit0_xy(double x, double y, bool bit0) {   // lane ^ 1 = quad_perm [1,0,3,2]
    const double send = bit0 ? x : y, keep = bit0 ? y : x;
    return keep + dpp_move<0xb1>(send);
}
template <int W, typename Step>
__device__ __forceinline__ void small_grid_sums32(Step&& step /* the lane's value at the next grid point */, int lane, double& s0, double& s1) {
    double z[4];
    const bool b1 = (lane >> 1) & 1, b0 = lane & 1;
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = step();
        if (W == 32) {
            const double u0 = fold_bit3(fold_bit4(v[0], v[1]), fold_bit4(v[2], v[3]));
            const double u1 = fold_bit3(fold_bit4(v[4], v[5]), fold_bit4(v[6], v[7]));
            z[bt] = fold_bit2(u0, u1);
        } else {
            const double u0 = fold_bit2(fold_bit3(v[0], v[1]), fold_bit3(v[2], v[3]));
            const double u1 = fold_bit2(fold_bit3(v[4], v[5]), fold_bit3(v[6], v[7]));
            z[bt] = fold_bit1(u0, u1, b1);
        }
    }
    if (W == 32) { s0 = fold_bit0_xy(fold_bit1(z[0], z[1], b1), fold_bit1(z[2], z[3], b1), b0); s1 = 0.0; }
    else { s0 = fold_bit0_xy(z[0], z[1], b0); s1 = fold_bit0_xy(z[2], z[3], b0); }
}
template <int W>
__device__ __forceinline__ int small_grid_point(int lane, int k) {
    if (W == 32) return 8 * (((lane >> 1) & 1) + 2 * (lane & 1)) + ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1);
    return 16 * k + 8 * (lane & 1) + ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1) + 4 * ((lane >> 1) & 1);
}
// Lane groups whose width is NOT a power of two (10, 12, 20 lanes: six, five, three edges per wavefront for the particle counts
// the reference's demos run -- a 16- / 32-lane group would idle 6 / 4 / 12 of its lanes): the folds stay inside the quad (lane ^ 1,
// and lane ^ 2 when the width is a multiple of four, so that no fold leaves its group), which leaves P = W / 2 or W / 4 partial
// sums per point; those go through wave-private LDS, eight points at a time, and the lane that OWNS a point (point p of a batch
// of 32 belongs to lane p % W of its group, as its (p / W)-th) adds them up in the order of the partial's index.  The grouping
// depends on the lane's place in its group only: an edge gets the same bits whatever shares its wavefront.
template <int W>
struct SmallGeom {
    static constexpr int G = WAVE / W;                                  // edges per wavefront
    static constexpr bool POW2 = W == 16 || W == 32;                    // (the butterfly networks of small_grid_sums32; 8 lanes take the LDS form)
    static constexpr int L = POW2 ? 0 : (W % 4 == 0 ? 2 : 1);           // folds before the partials go through LDS (W even)
    static constexpr int OWN = (8 + W - 1) / W;                         // points of a chunk of eight that a lane can own (W < 8: two)
    static constexpr int P = W >> L;                                    // partial sums per point
    static constexpr int R = (32 + W - 1) / W;                          // points of a batch of 32 that a lane owns
    static constexpr int BUF_AB = POW2 ? 0 : G * 4 * P;                 // the partials of eight points, in 16-byte units
};
template <int W, typename Step, typename Emit>
__device__ __forceinline__ void small_grid_sums_lds(Step&& step /* the lane's value at the next grid point */, int lane, int grp, int gl, bool lane_ok,
                                                    double* __restrict__ buf, Emit&& emit /* (owns a point of the chunk, the point, its sum) */) {
    using Geo = SmallGeom<W>;
    constexpr int L = Geo::L, P = Geo::P;
    const bool b0 = lane & 1, b1 = (lane >> 1) & 1;
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = step();
        static_assert(L >= 1, "an even width: the fold over lane ^ 1 stays inside the group");
        {
            double w[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) w[m] = fold_bit0_xy(v[2 * m], v[2 * m + 1], b0);       // point 2 m + b0 of the chunk
            if (L == 1) {
                if (lane_ok) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) buf[(grp * 8 + 2 * m + (int)b0) * P + (gl >> 1)] = w[m];
                }
            } else {
                const double u0 = fold_bit1(w[0], w[1], b1), u1 = fold_bit1(w[2], w[3], b1);   // points 4 k + 2 b1 + b0
                if (lane_ok) {
                    buf[(grp * 8 + 2 * (int)b1 + (int)b0) * P + (gl >> 2)] = u0;
                    buf[(grp * 8 + 4 + 2 * (int)b1 + (int)b0) * P + (gl >> 2)] = u1;
                }
            }
        }
        LHVI_WAVE_SYNC();
        // the points of this chunk that the lane owns (point p of the batch belongs to lane p % W: one per chunk for W >= 8, up to two below)
        int first = gl - (8 * bt) % W;
        if (first < 0) first += W;
#pragma unroll
        for (int o = 0; o < Geo::OWN; ++o) {
            const int off = first + o * W;
            const bool has = off < 8;
            const double* __restrict__ row = buf + (grp * 8 + (has ? off : 0)) * P;
            double sum = row[0];
#pragma unroll
            for (int k = 1; k < P; ++k) sum += row[k];
            emit(has, 8 * bt + off, sum);
        }
        LHVI_WAVE_SYNC();
    }
}
constexpr int GRID_MIN_NJ = 24;         // fewer partner particles: the direct loop is cheaper than 32 multiplications + 6 folds
constexpr double GRID_MAX_EXPONENT = 600.0;
constexpr int GRID_MAX_T = 128;         // batches of 32 points; the recurrence carries ~t ulp (the reference's RGM domain has 100 points, Demo/Data/RGM/Generator.py:16)
__device__ __forceinline__ bool grid_eligible(int uniform_grid, int nj, int T) { return uniform_grid && nj >= GRID_MIN_NJ && T <= GRID_MAX_T; }

__device__ __forceinline__ int round_log2_width(int rem) {      // 64 lanes for a full round, else next pow2 >= rem
    int lw = 6;
    if (rem <= 32) { lw = 0; while ((1 << lw) < rem) ++lw; }
    return lw;
}

__device__ __forceinline__ HeavyData heavy_fetch(const FastDesc& d, const lhvi_graph_t& g, const lhvi_pbp_t& s,
                                                 const double* __restrict__ v2f, int lane) {
    HeavyData h;
    // edges whose integral points go through the grid recurrence (or its fallback) fetch their particles only
    const int n = s.n, np = d.np, npts = grid_eligible(d.pad[1], d.nj, d.T) ? d.np : d.np + d.T;
    h.y = d.pval; h.m = 0.0; h.x0 = 0.0; h.x1 = 0.0;
    if (is_hidden(d.pval) && lane < d.nj) { h.y = s.old_particles[(int64_t)d.pv * n + lane]; h.m = v2f[(int64_t)d.pce * n + lane]; }
    {
        const int pl = lane & ((1 << round_log2_width(npts)) - 1);
        if (pl < npts) h.x0 = pl < np ? s.particles[(int64_t)d.tv * n + pl] : g.dom_val[d.gb + pl - np];
    }
    if (npts > 64) {
        const int rem = npts - 64, pl = lane & ((1 << round_log2_width(rem)) - 1), pp = 64 + pl;
        if (pl < rem) h.x1 = pp < np ? s.particles[(int64_t)d.tv * n + pp] : g.dom_val[d.gb + pp - np];
    }
    return h;
}

__global__ void __launch_bounds__(HEAVY_BLOCK) __attribute__((amdgpu_waves_per_eu(LHVI_HEAVY_WAVES, LHVI_HEAVY_WAVES))) pbp_f2v_heavy_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f,
                                                             double* __restrict__ f2v, const FastDesc* __restrict__ descs,
                                                             int nitems, uint32_t* __restrict__ stats) {
    __shared__ AB sh_all[HEAVY_BLOCK / WAVE][WAVE];
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    AB* sh = sh_all[wid];
    // (descs is a kernel argument of its own so that its restrict qualifier holds and the descriptors come through the
    // scalar cache: s_load does not take part in vmcnt, which the prefetched vector loads rely on)
    const int last = nitems - 1;
    const int n = s.n, S = s.n + s.T;
    WorkCursor<8, HEAVY_BLOCK / WAVE> cur;
    if (!cur.start(s.f2v_ticket, nitems, lane)) return;
    int n_grid = 0, n_direct = 0;          // edges of this wave whose integral points went through the recurrence / fell back
    // pipeline: the loads of edge k+1 are issued as soon as edge k has been staged into LDS (its registers are free
    // then, so nothing has to be rotated) and stay in flight through the term loops of edge k; `dn` is the full
    // descriptor of edge k+1, fetched one edge ahead, `d` the seven words the arithmetic of edge k needs
    struct { int32_t e, nj, np, T, gb, grid; double ay, by, c, axy, bx, kx, gx0, gh; } d;
    FastDesc dn = descs[cur.item];
    HeavyData h = heavy_fetch(dn, g, s, v2f, lane);
    for (;;) {
        d.e = dn.e; d.nj = dn.nj; d.np = dn.np; d.T = dn.T; d.gb = dn.gb; d.grid = dn.pad[1];
        d.ay = dn.ay; d.by = dn.by; d.c = dn.c; d.axy = dn.axy; d.bx = dn.bx; d.kx = dn.kx; d.gx0 = dn.pad2[0]; d.gh = dn.pad2[1];
        const int nxt = cur.next();
        const bool more = nxt < cur.limit;
        dn = descs[__builtin_amdgcn_readfirstlane(min(nxt, last))];
        const int np = d.np, nj = d.nj;
        double* out = f2v + (int64_t)d.e * S;
        const double kconst = d.kx;
        // integral points by recurrence along the uniform grid (below) when every exponent a_j + b_j x + kx x^2 stays far
        // inside the double range over the whole grid (then neither form under- or overflows and they agree to rounding)
        const bool eligible = grid_eligible(d.grid, nj, d.T);
        bool grid_path = eligible && !(s.flags & (LHVI_PBP_SKIP_TERMS | LHVI_PBP_NO_GRID));
        AB mine;
        mine.a = PAD_LOG_TERM; mine.b = 0.0;                     // padding: underflows to exactly 0 whatever the point's own constant adds
        LHVI_WAVE_SYNC();
        {
            if (lane < nj) {
                const double y = h.y;
                mine.a = (d.ay * y + d.by) * y + d.c + h.m;
                mine.b = d.axy * y + d.bx;
            }
#if LHVI_HEAVY_FLOOR
            AB scaled;                                     // the term loops read the records in units of the table step
            scaled.a = mine.a * LHVI_EXP_INV_STEP; scaled.b = mine.b * LHVI_EXP_INV_STEP;
            sh[lane] = scaled;
#else
            sh[lane] = mine;
#endif
        }
        LHVI_WAVE_SYNC();
        if (grid_path) {
            const double X = fmax(fabs(d.gx0), fabs(fma((double)(d.T - 1), d.gh, d.gx0)));
            const double bound = fma(fabs(mine.b) + fabs(kconst) * X, X, fabs(mine.a));
            grid_path = __ballot(lane < nj && !(bound < GRID_MAX_EXPONENT)) == 0;
            if (grid_path) ++n_grid; else ++n_direct;
        }
        const int npts = eligible ? np : np + d.T;          // eligible: the integral points are handled after the particle rounds
        const double x0 = h.x0, x1 = h.x1;
        if (more) h = heavy_fetch(dn, g, s, v2f, lane);
        if (grid_path) {
            double gv = exp_core(fma(mine.b, d.gx0, mine.a), sh_tab);
            const double q = exp_core(mine.b * d.gh, sh_tab);
#ifdef LHVI_DIAG_SITE_MOMENTS          // timing aid (never defined in the product build): what forming the 'simple' rule's site in this epilogue would cost
            double mz = 0.0, ma = 0.0, mb = 0.0;
#endif
            for (int t0 = 0; t0 < d.T; t0 += 32) {
                const double sum = grid_sums32(gv, q, lane);
                const int t = t0 + grid_owned_point(lane);
                if (t < d.T && !(lane & 1)) {
                    const double xt = fma((double)t, d.gh, d.gx0);
                    out[n + t] = sum > 0.0 ? fma(kconst * xt, xt, log_table(sum, sh_log)) : -700.0;
#ifdef LHVI_DIAG_SITE_MOMENTS
                    const double w = sum * exp_core(kconst * xt * xt, sh_tab);      // = exp(message at x_t)
                    mz += w; ma += w * xt; mb += w * (xt * xt);
#endif
                }
            }
#ifdef LHVI_DIAG_SITE_MOMENTS
            {
                mz = wave_sum(mz); ma = wave_sum(ma); mb = wave_sum(mb);
                const double rz = rcp_newton(mz);
                const double mu = ma * rz;
                double sig = mb * rz - mu * mu;
                sig = fmax(sig, s.var_threshold * 4.0);
                // (stored only under a condition no run meets: the arithmetic has to happen, the site array stays the proposal kernel's)
                if (lane == 0 && mz < -1.0) { out[0] = mu; out[1] = sig; }
            }
#endif
        } else if (eligible) {
            // an exponent too close to the double range somewhere on the grid: the direct form, points fetched here
#pragma nounroll
            for (int t0 = 0; t0 < d.T; t0 += 64) {
                const int rem = d.T - t0;
                const int lw = round_log2_width(rem);
                const int width = 1 << lw, split = 64 >> lw, sub = lane >> lw, pl = lane & (width - 1);
                const bool valid = pl < rem;
                const double X1 = valid ? g.dom_val[d.gb + t0 + pl] : 0.0, C = kconst * X1 * X1;
                const int chunk = (s.flags & LHVI_PBP_SKIP_TERMS) ? 0 : (nj + split - 1) >> (6 - lw);
#if LHVI_HEAVY_FLOOR
                double acc = fast_accumulate_floor<LHVI_HEAVY_UNROLL>(sh + sub * chunk, sh_tab, chunk, X1, C);
#else
                double acc = fast_accumulate_uniform<MODE_CONST, LHVI_HEAVY_UNROLL>(sh + sub * chunk, nullptr, sh_tab, chunk, X1, 0.0, C);
#endif
                for (int off = width; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
                if (valid && sub == 0) out[n + t0 + pl] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
            }
        }
#pragma nounroll
        for (int r = 0; r < 2; ++r) {
            const int rem = npts - 64 * r;
            if (rem <= 0) break;
            const int lw = round_log2_width(rem);
            const int width = 1 << lw, split = 64 >> lw, sub = lane >> lw, pl = lane & (width - 1);
            const int p = 64 * r + pl;
            const bool valid = pl < rem;
            const double xv = r == 0 ? x0 : x1;
            const double X1 = valid ? xv : 0.0, C = kconst * X1 * X1;
            const int chunk = (s.flags & LHVI_PBP_SKIP_TERMS) ? 0 : (nj + split - 1) >> (6 - lw);   // flag 16: tuning aid, skips the term loop
#if LHVI_HEAVY_FLOOR
            double acc = fast_accumulate_floor<LHVI_HEAVY_UNROLL>(sh + sub * chunk, sh_tab, chunk, X1, C);
#else
            double acc = fast_accumulate_uniform<MODE_CONST, LHVI_HEAVY_UNROLL>(sh + sub * chunk, nullptr, sh_tab, chunk, X1, 0.0, C);
#endif
            for (int off = width; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
            if (valid && sub == 0) out[p < np ? p : n + (p - np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
        }
        if (!more) break;
        cur.advance(nxt, lane);
    }
    if (stats && lane == 0) { atomicAdd(stats, (uint32_t)n_grid); atomicAdd(stats + 1, (uint32_t)n_direct); }
}

// HEAVY-class edges with FEW particles (target and partner both at most PPL * W, W = 8 ... 32 lanes, PPL = 1 or 2 particles per
// lane: the particle counts of the reference's demos; the launcher picks by lhvi_pbp_t.n), floor(64 / W) edges per wavefront.  With one edge per wavefront such an edge keeps 64 lanes busy for a dozen terms and then
// waits out the latencies of its own loads, LDS hand-off and stores (gfx950 counts loads and stores in one in-order counter, so
// the wait for the next edge's operands includes this edge's store acknowledgements): ~10 us per edge whatever it holds.  Here a
// lane group of W lanes owns an edge: lane = partner particle while the records are staged (wave-private LDS, one block of W
// records per group), then lane = output point in rounds of W points (particles first, then the integral points); every lane runs
// the same term loop over its group's records (padded to the longest list of the wave with terms that underflow to 0), so no
// cross-lane reduction is needed at all, and every load, store and wait is shared by 64 / W edges.  Same term arithmetic as the
// heavy kernel's direct form (fast_accumulate_floor): an edge gets the same message from either kernel up to the order of the sum
// when the heavy kernel splits a short round across lane groups.
#ifndef LHVI_SMALL_HOIST
#define LHVI_SMALL_HOIST 2          // 0: round 4's loads (a global round trip per descriptor piece and per round); 1: two round trips per step;
#endif                              // 2: + the next step's descriptor touched a step ahead (scripts/diag/small_hoist.sh: 2.71 / 2.55 / 2.51 ms at n = 16)
#ifndef LHVI_SMALL_GRID
#define LHVI_SMALL_GRID 1         // integral points on a uniform grid by the recurrence along the grid (0: one exponential per point and particle)
#endif
#ifndef LHVI_SMALL_PAD
#define LHVI_SMALL_PAD 1          // a group's block of W 16-byte records starts one record further than W records after the one before it:
#endif                            // unpadded, record j of every group lies in the same four banks
#ifndef LHVI_SMALL_WAVES
#define LHVI_SMALL_WAVES 6
#endif
#ifndef LHVI_SMALL_WAVES10
#define LHVI_SMALL_WAVES10 5      // the 10-lane groups (six edges per wavefront) spill 12 words per lane at 80 registers
#endif
#ifndef LHVI_SMALL_WAVES2
#define LHVI_SMALL_WAVES2 5       // two particles per lane: two sets of record / recurrence values live (90-96 registers, no scratch)
#endif
// PPL = 2: a lane holds TWO partner particles (j and j + W) and serves two rounds of output points per W particles -- a group of W
// lanes then owns an edge with up to 2 W particles on both sides, so twice as many edges share a wavefront (n = 20: six instead of
// three, n = 32: four instead of two).  The direct rounds cost the same per edge (a round of W points over 2 W records instead of
// half as many rounds of 2 W points); the grid recurrence adds a lane's two values before the folds, which are then paid once
// for twice the edges, and so are the descriptor fetch, the staging and every wait.  Same record order, same term loop: the
// particle part of a message keeps its bits.
template <int W, int PPL = 1>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(PPL == 2 ? LHVI_SMALL_WAVES2 : W == 10 ? LHVI_SMALL_WAVES10 : LHVI_SMALL_WAVES, 8))) pbp_f2v_small_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f,
                                                            double* __restrict__ f2v, const FastDesc* __restrict__ descs, int nitems) {
    using Geo = SmallGeom<W>;
    constexpr int G = Geo::G;                                // edges per wavefront
    constexpr int GS = PPL * W + LHVI_SMALL_PAD;              // records between two groups' blocks (padded: see LHVI_SMALL_PAD)
    // (a group width that is not a power of two: the partial sums of the grid recurrence use the records' space once the
    // direct rounds are through with them -- DS operations of a wavefront execute in order)
    constexpr int SH_AB = G * GS > Geo::BUF_AB ? G * GS : Geo::BUF_AB;
    __shared__ AB sh_all[BLOCK / WAVE][SH_AB];
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // (a width that does not divide 64 leaves 64 - G W lanes without an edge: they follow the last lane of the last group,
    // store nothing and write nothing to LDS; the folds of the recurrence never leave a quad, and G W is a multiple of four)
    const bool lane_ok = lane < G * W;
    const int grp = lane_ok ? lane / W : G - 1, gl = lane_ok ? lane % W : W - 1;
    AB* sh = sh_all[wid];
    const AB* mine_recs = sh + grp * GS;
    const int n = s.n, S = s.n + s.T;
    const int nsteps = (nitems + G - 1) / G;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    for (int step = blockIdx.x * (BLOCK / WAVE) + wid; step < nsteps; step += nwaves) {
        const int idx = step * G + grp;
        const bool live = idx < nitems && lane_ok;
#if LHVI_SMALL_HOIST
        // One round trip for the descriptor (all 128 bytes at once, whatever the branches below use of it), one for everything it
        // points to: the partner's particles and message and the output points of the first three rounds (all of them for
        // n + T <= 3 W), loaded without branches from addresses clamped into their rows.  A round that loads its own points, or a
        // descriptor read field by field where the branches need it, costs a dependent global round trip each -- five or six per
        // step, which seven waves per SIMD do not cover.
        union { FastDesc d; int4 q[sizeof(FastDesc) / 16]; } u;
        {
            const int4* dp = reinterpret_cast<const int4*>(descs + min(idx, nitems - 1));
#pragma unroll
            for (int k = 0; k < (int)(sizeof(FastDesc) / 16); ++k) u.q[k] = dp[k];
        }
#if LHVI_SMALL_HOIST >= 2
        // the next step's descriptor (one 128-byte line, streamed from HBM) is touched now, so that its read at the head of the
        // next step finds it in the cache; the word is consumed at the end of this step
        const int touch = reinterpret_cast<const int*>(descs + min((step + nwaves) * G + grp, nitems - 1))[0];
#endif
        const FastDesc& d = u.d;
        const int e = d.e, tv = d.tv, nj = d.nj, np = d.np, T = d.T, gb = d.gb;
        const double pval = d.pval, kx = d.kx;
        const int npts = np + T;
        double yl[PPL], ml[PPL];
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            const int jl = min(gl + pp * W, nj - 1);
            yl[pp] = s.old_particles[(int64_t)d.pv * n + jl]; ml[pp] = v2f[(int64_t)d.pce * n + jl];
        }
        double xs[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = r * W + gl;
            const double* src = p < np ? s.particles + ((int64_t)tv * n + p) : g.dom_val + (gb + max(min(p - np, T - 1), 0));
            xs[r] = *src;
        }
        const bool hidden_partner = is_hidden(pval);
#else
        static_assert(PPL == 1, "the round-4 loads know one particle per lane");
        const FastDesc& d = descs[live ? idx : nitems - 1];  // (a group past the end repeats the last entry and stores nothing)
        const int e = d.e, tv = d.tv, nj = d.nj, np = d.np, T = d.T, gb = d.gb;
        const double pval = d.pval, kx = d.kx;
        const int npts = np + T;
        // staging: lane = partner particle of its group's edge
        const bool hidden_partner = is_hidden(pval);
        double yl[1] = {pval}, ml[1] = {0.0};
        if (gl < nj && hidden_partner) { yl[0] = s.old_particles[(int64_t)d.pv * n + gl]; ml[0] = v2f[(int64_t)d.pce * n + gl]; }
#endif
        double ua[PPL], ub[PPL];
        AB rec[PPL];
#pragma unroll
        for (int pp = 0; pp < PPL; ++pp) {
            const double y = hidden_partner ? yl[pp] : pval, m = hidden_partner ? ml[pp] : 0.0;
            ua[pp] = PAD_LOG_TERM; ub[pp] = 0.0;                   // padding: underflows to exactly 0 whatever the point's own constant adds
            if (gl + pp * W < nj) {
                ua[pp] = (d.ay * y + d.by) * y + d.c + m;
                ub[pp] = d.axy * y + d.bx;
            }
            rec[pp].a = ua[pp] * LHVI_EXP_INV_STEP; rec[pp].b = ub[pp] * LHVI_EXP_INV_STEP;      // (records in units of the table step: floor form)
        }
#if LHVI_SMALL_GRID
        // Integral points on a uniform grid: the recurrence of the heavy kernel inside the lane group (small_grid_sums32), when every
        // exponent of the edge stays far inside the double range over the whole grid -- a property of the edge alone, so an edge
        // gets the same bits whatever shares its wavefront.  A group that fails the test takes the direct rounds for all its points.
        const double gx0 = d.pad2[0], gh = d.pad2[1];
        bool gok = false;
        if (d.pad[1] && T <= GRID_MAX_T && !(s.flags & (LHVI_PBP_SKIP_TERMS | LHVI_PBP_NO_GRID))) {
            const double X = fmax(fabs(gx0), fabs(fma((double)(T - 1), gh, gx0)));
            bool mine_bad = false;
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) {
                const double bound = fma(fabs(ub[pp]) + fabs(kx) * X, X, fabs(ua[pp]));
                mine_bad |= gl + pp * W < nj && !(bound < GRID_MAX_EXPONENT);
            }
            const uint64_t bad = __ballot(mine_bad);
            gok = ((bad >> (grp * W)) & ((1ull << W) - 1)) == 0;
        }
        const int lim = gok ? np : npts;                           // points of this group's edge that the direct rounds serve
        const bool any_grid = __ballot(gok && live) != 0;
#else
        const int lim = npts;
#endif
        // the longest record list and the most output points of the wave's edges (wave-uniform loop bounds)
        int jmax = 0, pmax = 0;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            jmax = max(jmax, __builtin_amdgcn_readlane(nj, k * W));
            pmax = max(pmax, __builtin_amdgcn_readlane(lim, k * W));
        }
        if (s.flags & LHVI_PBP_SKIP_TERMS) jmax = 0;
        double* out = f2v + (int64_t)e * S;
        LHVI_WAVE_SYNC();
        if (lane_ok) {
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) sh[grp * GS + gl + pp * W] = rec[pp];
        }
        LHVI_WAVE_SYNC();
#if LHVI_SMALL_HOIST
        // the three prefetched rounds written out: each waits for ITS points only (loads return in order), not -- as a loop whose
        // later rounds load their own points makes the compiler assume -- for everything in flight, the previous round's stores included
        // (the three loads were issued together and arrive together: taking all of them here costs nothing, and no later round then
        // waits on the memory counter, which would include the stores of the rounds before it)
        asm volatile("" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (r * W < pmax) {
                const int p = r * W + gl;
                const bool valid = live && p < lim;
                const double x = valid ? xs[r] : 0.0;
                const double acc = fast_accumulate_floor<4, true>(mine_recs, sh_tab, jmax, x, kx * x * x);
                if (valid) out[p < np ? p : n + (p - np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
            }
        }
#pragma unroll 1
        for (int p0 = 3 * W; p0 < pmax; p0 += W) {
#else
#pragma unroll 1
        for (int p0 = 0; p0 < pmax; p0 += W) {
#endif
            const int p = p0 + gl;
            const bool valid = live && p < lim;
            double x = 0.0;
            if (valid) x = p < np ? s.particles[(int64_t)tv * n + p] : g.dom_val[gb + p - np];
            const double acc = fast_accumulate_floor<4, true>(mine_recs, sh_tab, jmax, x, kx * x * x);
            if (valid) out[p < np ? p : n + (p - np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
        }
#if LHVI_SMALL_GRID
        if (any_grid) {
            double gv[PPL], q[PPL];
#pragma unroll
            for (int pp = 0; pp < PPL; ++pp) { gv[pp] = exp_core(fma(ub[pp], gx0, ua[pp]), sh_tab); q[pp] = exp_core(ub[pp] * gh, sh_tab); }
            auto next_value = [&]() {                              // the lane's particles' sum at the next grid point
                double v = gv[0];
                gv[0] *= q[0];
                if constexpr (PPL == 2) { v += gv[1]; gv[1] *= q[1]; }
                return v;
            };
            int tmax = 0;
#pragma unroll
            for (int k = 0; k < G; ++k) tmax = max(tmax, __builtin_amdgcn_readlane(gok ? T : 0, k * W));
#pragma unroll 1
            for (int t0 = 0; t0 < tmax; t0 += 32) {
                if constexpr (Geo::POW2) {
                    double sum[2];
                    small_grid_sums32<W>(next_value, lane, sum[0], sum[1]);
#pragma unroll
                    for (int k = 0; k < (W == 16 ? 2 : 1); ++k) {
                        const int t = t0 + small_grid_point<W>(gl, k);
                        if (live && gok && t < T) {
                            const double xt = fma((double)t, gh, gx0);
                            out[n + t] = sum[k] > 0.0 ? fma(kx * xt, xt, log_table(sum[k], sh_log)) : -700.0;
                        }
                    }
                } else {
                    if constexpr (Geo::R >= 4) {
                        // a lane owns a point in (almost) every chunk of eight: the logarithm and the store right there
                        small_grid_sums_lds<W>(next_value, lane, grp, gl, lane_ok, reinterpret_cast<double*>(sh), [&](bool has, int p, double sum) {
                            const int t = t0 + p;
                            if (has && live && gok && t < T) {
                                const double xt = fma((double)t, gh, gx0);
                                out[n + t] = sum > 0.0 ? fma(kx * xt, xt, log_table(sum, sh_log)) : -700.0;
                            }
                        });
                    } else {
                        // fewer owned points than chunks: the sums are kept and go through the logarithm together
                        double sum[Geo::R];
#pragma unroll
                        for (int r = 0; r < Geo::R; ++r) sum[r] = 0.0;
                        small_grid_sums_lds<W>(next_value, lane, grp, gl, lane_ok, reinterpret_cast<double*>(sh), [&](bool has, int p, double v) {
#pragma unroll
                            for (int r = 0; r < Geo::R; ++r) sum[r] = (has && p >= r * W && p < (r + 1) * W) ? v : sum[r];
                        });
#pragma unroll
                        for (int r = 0; r < Geo::R; ++r) {
                            const int p = r * W + gl, t = t0 + p;
                            if (live && gok && p < 32 && t < T) {
                                const double xt = fma((double)t, gh, gx0);
                                out[n + t] = sum[r] > 0.0 ? fma(kx * xt, xt, log_table(sum[r], sh_log)) : -700.0;
                            }
                        }
                    }
                }
            }
        }
#endif
#if LHVI_SMALL_HOIST >= 2
        asm volatile("" :: "v"(touch));
#endif
    }
}

// LIGHT edges: HybridQuadratic(1 discrete, 1 continuous) with a binary (or observed) discrete side -- the edges between
// the continuous and the binary variables of the benchmark.  A handful of terms per output point, so the general
// kernel's staging / splitting / shuffling is all overhead; here nothing goes through LDS but the two tables:
//   type 1 (continuous target):  lane = output point, log sum_s exp(A_s x^2 + b_s x + c_s + m_s), s = partner states
//   type 2 (discrete target):    lane = partner particle j, one wave reduction per target state p of
//                                exp(A_p y_j^2 + b_p y_j + c_p + m_j)
// Same software pipeline as the heavy kernel (descriptor one edge ahead, vector loads in flight during the arithmetic).
struct LightData { double a, b, m0, m1; };   // type 1: own points of round 0 / 1 + the messages of the partner's (at most
                                             // two) states;  type 2: partner particle and its message

__device__ __forceinline__ LightData light_fetch(const FastDesc& d, const lhvi_graph_t& g, const lhvi_pbp_t& s,
                                                 const double* __restrict__ v2f, int lane) {
    LightData h;
    h.a = 0.0; h.b = 0.0; h.m0 = 0.0; h.m1 = 0.0;
    const int n = s.n;
    if (d.pad[0] == 1) {
        const int np = d.np, npts = d.np + d.T;
        // wave-uniform, fetched with the rest of the edge (one edge ahead) instead of at the head of its arithmetic
        if (is_hidden(d.pval)) { h.m0 = v2f[(int64_t)d.pce * n]; if (d.nj > 1) h.m1 = v2f[(int64_t)d.pce * n + 1]; }
        if (lane < npts) h.a = lane < np ? s.particles[(int64_t)d.tv * n + lane] : g.dom_val[d.gb + lane - np];
        const int pp = 64 + lane;
        if (pp < npts) h.b = pp < np ? s.particles[(int64_t)d.tv * n + pp] : g.dom_val[d.gb + pp - np];
    } else if (lane < d.nj) {
        h.a = d.pval;
        if (is_hidden(d.pval)) { h.a = s.old_particles[(int64_t)d.pv * n + lane]; h.b = v2f[(int64_t)d.pce * n + lane]; }
    }
    return h;
}

__global__ void __launch_bounds__(BLOCK) pbp_f2v_light_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f,
                                                             double* __restrict__ f2v, const FastDesc* __restrict__ descs,
                                                             int nitems) {
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int lane = threadIdx.x & 63;
    const int last = nitems - 1;
    const int n = s.n, S = s.n + s.T;
    // (static striding: this kernel's entries are short, a late workgroup owes little, and neighbouring waves on neighbouring
    // entries measured 15 % faster than chunks of 64 through the ticket)
    WorkCursor<64> work;
    if (!work.start(nullptr, nitems, lane)) return;
    struct { int32_t e, type, nj, np, T, pce; double pval, A0, b0, c0, A1, b1, c1; } d;
    FastDesc dn = descs[work.item];
    LightData h = light_fetch(dn, g, s, v2f, lane);
    for (;;) {
        d.e = dn.e; d.type = dn.pad[0]; d.nj = dn.nj; d.np = dn.np; d.T = dn.T; d.pce = dn.pce; d.pval = dn.pval;
        d.A0 = dn.ay; d.b0 = dn.by; d.c0 = dn.c; d.A1 = dn.axy; d.b1 = dn.bx; d.c1 = dn.kx;
        const int nxt = work.next();
        const bool more = nxt < work.limit;
        dn = descs[__builtin_amdgcn_readfirstlane(min(nxt, last))];
        double* out = f2v + (int64_t)d.e * S;
        const LightData cur = h;
        if (d.type == 1) {
            const double m0 = cur.m0, m1 = cur.m1;       // messages of the (at most two) partner states
            if (more) h = light_fetch(dn, g, s, v2f, lane);
            const int npts = d.np + d.T;
#pragma nounroll
            for (int r = 0; r < 2; ++r) {
                const int p = 64 * r + lane;
                if (64 * r >= npts) break;
                const double x = r == 0 ? cur.a : cur.b;
                double acc = exp_core(fma(x, fma(x, d.A0, d.b0), d.c0 + m0), sh_tab);
                if (d.nj > 1) acc += exp_core(fma(x, fma(x, d.A1, d.b1), d.c1 + m1), sh_tab);
                if (p < npts) out[p < d.np ? p : n + (p - d.np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
            }
        } else {
            if (more) h = light_fetch(dn, g, s, v2f, lane);
            const double y = cur.a, m = cur.b;
            double res = wave_sum(lane < d.nj ? exp_core(fma(y, fma(y, d.A0, d.b0), d.c0 + m), sh_tab) : 0.0);
            if (d.np > 1) {
                const double sum1 = wave_sum(lane < d.nj ? exp_core(fma(y, fma(y, d.A1, d.b1), d.c1 + m), sh_tab) : 0.0);
                if (lane == 1) res = sum1;
            }
            if (lane < d.np) out[lane] = res > 0.0 ? log_table(res, sh_log) : -700.0;
        }
        if (!more) break;
        work.advance(nxt, lane);
    }
}

// PAIRS: the two light edges of one HybridQuadratic(1 discrete, 1 continuous) factor served by one list entry.  The light
// kernel above is bound by memory-level parallelism -- one edge in flight per wave plus one prefetched -- and the two edges
// of such a factor share the potential's per-state coefficients but read different rows (the continuous variable's new
// particles for the message to it; its old particles and its v -> f row for the message to the discrete variable).  One
// 128-byte descriptor per factor therefore doubles the bytes a wave keeps in flight at the same scalar-register budget and
// halves the descriptor traffic.  Same expressions per edge as the light kernel (bit-identical messages).
struct PairDesc {
    int32_t e_c, e_d;            // edge to the continuous / to the discrete variable; -1: that variable is observed (no message)
    int32_t v_c, v_d;
    int32_t np_c, T, gb, ns;     // particles of v_c (hidden), its grid size / base in dom_val, live states of v_d (1 when observed)
    double val_c, val_d;         // evidence values, NaN = hidden
    double A0, b0, c0, A1, b1, c1;   // log phi = A_s x^2 + b_s x + c_s for the (at most two) live states s of v_d
    int32_t mc, md;              // rows of v2f holding the discrete variable's (mc) / the continuous variable's (md) message to the factor
    int32_t pad[6];
};
static_assert(sizeof(PairDesc) == LHVI_PBP_DESC_BYTES, "PairDesc is part of the ABI (LHVI_PBP_DESC_BYTES)");

struct PairData { double x0, x1, m0, m1, y, mj; };
#ifndef LHVI_PAIR_AHEAD
#define LHVI_PAIR_AHEAD 1          // (two entries ahead costs a wave slot to the scalar registers: 1.12 -> 1.18 ms, scripts/diag/pair_ahead.sh)
#endif

__device__ __forceinline__ PairData pair_fetch(const PairDesc& d, const lhvi_graph_t& g, const lhvi_pbp_t& s,
                                               const double* __restrict__ v2f, int lane) {
    PairData h;
    h.x0 = 0.0; h.x1 = 0.0; h.m0 = 0.0; h.m1 = 0.0; h.y = d.val_c; h.mj = 0.0;
    const int n = s.n;
    if (d.e_c >= 0) {
        const int np = d.np_c, npts = d.np_c + d.T;
        if (lane < npts) h.x0 = lane < np ? s.particles[(int64_t)d.v_c * n + lane] : g.dom_val[d.gb + lane - np];
        const int pp = 64 + lane;
        if (pp < npts) h.x1 = pp < np ? s.particles[(int64_t)d.v_c * n + pp] : g.dom_val[d.gb + pp - np];
        if (is_hidden(d.val_d)) { h.m0 = v2f[(int64_t)d.mc * n]; if (d.ns > 1) h.m1 = v2f[(int64_t)d.mc * n + 1]; }
    }
    if (d.e_d >= 0 && is_hidden(d.val_c) && lane < d.np_c) {
        h.y = s.old_particles[(int64_t)d.v_c * n + lane];
        h.mj = v2f[(int64_t)d.md * n + lane];
    }
    return h;
}

__global__ void __launch_bounds__(BLOCK) pbp_f2v_pair_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f,
                                                            double* __restrict__ f2v, const PairDesc* __restrict__ descs,
                                                            int nitems) {
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int lane = threadIdx.x & 63;
    const int last = nitems - 1;
    const int n = s.n, S = s.n + s.T;
    // static striding, descriptors TWO entries ahead: an iteration is short (a few hundred instructions), so a descriptor requested at
    // its head and needed at once for the next entry's loads -- as in the heavy kernel, whose iterations are fifty times longer --
    // would put a scalar load's full latency (the list streams from HBM) into every iteration
    const int stride = gridDim.x * (BLOCK / WAVE);
    int item = blockIdx.x * (BLOCK / WAVE) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (item >= nitems) return;
    struct { int32_t e_c, e_d, np_c, T, ns; double val_c, A0, b0, c0, A1, b1, c1; } d;
    PairDesc dn = descs[item];
#if LHVI_PAIR_AHEAD >= 2
    PairDesc dn2 = descs[__builtin_amdgcn_readfirstlane(min(item + stride, last))];
#endif
    PairData h = pair_fetch(dn, g, s, v2f, lane);
    for (;;) {
        d.e_c = dn.e_c; d.e_d = dn.e_d; d.np_c = dn.np_c; d.T = dn.T; d.ns = dn.ns; d.val_c = dn.val_c;
        d.A0 = dn.A0; d.b0 = dn.b0; d.c0 = dn.c0; d.A1 = dn.A1; d.b1 = dn.b1; d.c1 = dn.c1;
        const bool more = item + stride < nitems;
#if LHVI_PAIR_AHEAD >= 2
        dn = dn2;
        dn2 = descs[__builtin_amdgcn_readfirstlane(min(item + 2 * stride, last))];
#else
        dn = descs[__builtin_amdgcn_readfirstlane(min(item + stride, last))];
#endif
        const PairData cur = h;
        if (more) h = pair_fetch(dn, g, s, v2f, lane);
        if (d.e_c >= 0) {
            // message to the continuous variable: lane = output point, log sum over the discrete variable's live states
            double* out = f2v + (int64_t)d.e_c * S;
            const int npts = d.np_c + d.T;
#pragma nounroll
            for (int r = 0; r < 2; ++r) {
                const int p = 64 * r + lane;
                if (64 * r >= npts) break;
                const double x = r == 0 ? cur.x0 : cur.x1;
                double acc = exp_core(fma(x, fma(x, d.A0, d.b0), d.c0 + cur.m0), sh_tab);
                if (d.ns > 1) acc += exp_core(fma(x, fma(x, d.A1, d.b1), d.c1 + cur.m1), sh_tab);
                if (p < npts) out[p < d.np_c ? p : n + (p - d.np_c)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
            }
        }
        if (d.e_d >= 0) {
            // message to the discrete variable: lane = particle of the continuous variable, one wave reduction per state
            double* out = f2v + (int64_t)d.e_d * S;
            const int nj = is_hidden(d.val_c) ? d.np_c : 1;
            const double y = cur.y, m = cur.mj;
            double res = wave_sum(lane < nj ? exp_core(fma(y, fma(y, d.A0, d.b0), d.c0 + m), sh_tab) : 0.0);
            if (d.ns > 1) {
                const double sum1 = wave_sum(lane < nj ? exp_core(fma(y, fma(y, d.A1, d.b1), d.c1 + m), sh_tab) : 0.0);
                if (lane == 1) res = sum1;
            }
            if (lane < d.ns) out[lane] = res > 0.0 ? log_table(res, sh_log) : -700.0;
        }
        if (!more) break;
        item += stride;
    }
}

#ifndef LHVI_PAIR_SMALL
#define LHVI_PAIR_SMALL 1
#endif
// PAIRS with FEW particles (every variable of the run has at most W = 16, 20 or 32: lhvi_pbp_t.n <= W): floor(64 / W) list entries per
// wavefront, a lane group of W lanes each -- the one-entry-per-wave kernel above keeps 48 of 64 lanes busy for two exponentials
// and then waits for its own loads.  Same expressions per output point, and the sums over the continuous variable's particles
// run through the same reduction network at the same positions inside the group (dpp_row_reduce / dpp_reduce_rows32 are the first
// stages of wave_sum, whose later stages add the zeros of the empty rows; a group of 20 lanes, which is no row of that network,
// spells the same tree out over shuffled pair sums): the same bits as the kernel above.  Descriptor and operands in two round
// trips, as in pbp_f2v_small_kernel.
template <int W>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8)))
pbp_f2v_pair_small_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f, double* __restrict__ f2v,
                          const PairDesc* __restrict__ descs, int nitems) {
    constexpr int G = WAVE / W;
    static_assert(W % 2 == 0, "pairs of lanes (lane ^ 1) stay inside a group");
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int lane = threadIdx.x & 63;
    // (a width that does not divide 64 -- 20 lanes: three entries per wavefront -- leaves 64 - G W lanes without an entry: they follow
    // the last lane of the last group and store nothing)
    const bool lane_ok = lane < G * W;
    const int grp = lane_ok ? lane / W : G - 1, gl = lane_ok ? lane % W : W - 1;
    const int first_lane = grp * W;
    const int n = s.n, S = s.n + s.T;
    const int nsteps = (nitems + G - 1) / G;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    for (int step = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); step < nsteps; step += nwaves) {
        const int idx = step * G + grp;
        const bool live = idx < nitems && lane_ok;
        union { PairDesc d; int4 q[sizeof(PairDesc) / 16]; } u;
        {
            const int4* dp = reinterpret_cast<const int4*>(descs + min(idx, nitems - 1));
#pragma unroll
            for (int k = 0; k < (int)(sizeof(PairDesc) / 16); ++k) u.q[k] = dp[k];
        }
        const int touch = reinterpret_cast<const int*>(descs + min((step + nwaves) * G + grp, nitems - 1))[0];   // (the next step's line)
        const PairDesc& d = u.d;
        const bool to_c = live && d.e_c >= 0, to_d = live && d.e_d >= 0;
        const int np = d.np_c, T = d.T, npts = d.np_c + d.T;
        // everything the descriptor points to, without branches (addresses clamped into their rows; unused values dropped)
        double xs[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = r * W + gl;
            const double* src = p < np ? s.particles + ((int64_t)d.v_c * n + p) : g.dom_val + (d.gb + max(min(p - np, T - 1), 0));
            xs[r] = *src;
        }
        const int mc = max(d.mc, 0), md = max(d.md, 0), jl = min(gl, max(np - 1, 0));
        const double m0l = v2f[(int64_t)mc * n], m1l = v2f[(int64_t)mc * n + (d.ns > 1 ? 1 : 0)];
        const double yl = s.old_particles[(int64_t)d.v_c * n + jl], mjl = v2f[(int64_t)md * n + jl];
        asm volatile("" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]));
        const bool hid_d = is_hidden(d.val_d), hid_c = is_hidden(d.val_c);
        const double m0 = (to_c && hid_d) ? m0l : 0.0, m1 = (to_c && hid_d && d.ns > 1) ? m1l : 0.0;
        // ---- message to the continuous variable: lane = output point, log sum over the discrete variable's live states
        int pmax = 0;
#pragma unroll
        for (int k = 0; k < G; ++k) pmax = max(pmax, __builtin_amdgcn_readlane(to_c ? npts : 0, k * W));
        double* out_c = f2v + (int64_t)max(d.e_c, 0) * S;
        auto point = [&](int p, double x) {
            double acc = exp_core(fma(x, fma(x, d.A0, d.b0), d.c0 + m0), sh_tab);
            if (d.ns > 1) acc += exp_core(fma(x, fma(x, d.A1, d.b1), d.c1 + m1), sh_tab);
            if (to_c && p < npts) out_c[p < np ? p : n + (p - np)] = acc > 0.0 ? log_table(acc, sh_log) : -700.0;
        };
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r * W < pmax) point(r * W + gl, xs[r]);
#pragma unroll 1
        for (int p0 = 3 * W; p0 < pmax; p0 += W) {
            const int p = p0 + gl;
            double x = 0.0;
            if (to_c && p < npts) x = p < np ? s.particles[(int64_t)d.v_c * n + p] : g.dom_val[d.gb + p - np];
            point(p, x);
        }
        // ---- message to the discrete variable: lane = particle of the continuous variable, one group reduction per state
        if (__ballot(to_d)) {
            const int nj = hid_c ? np : 1;
            const double y = hid_c ? yl : d.val_c, m = hid_c ? mjl : 0.0;
            auto group_sum = [&](double x) {
                if constexpr (W == 16) return dpp_move<0x15F>(dpp_row_reduce(x, SumOp()));      // lane 15's sum: the one wave_sum hands on (each lane adds in its own order)
                else if constexpr (W == 32) {
                    x = dpp_reduce_rows32(x, SumOp());
                    return lane < 32 ? readlane_f64(x, 31) : readlane_f64(x, 63);
                } else {
                    // a group that is no row of the DPP network: the SAME tree, spelled out.  wave_sum's lane 63 (and the 16- / 32-lane
                    // forms' lanes 15 / 31) hold  rows added in order, a row = (Q0 + Q1) + (Q2 + Q3), a quad Q = (v0 + v1) + (v2 + v3);
                    // the lanes beyond the particles hold zeros there, and adding a zero changes nothing -- so the tree over the live
                    // pairs alone has the same bits.  Pair sums by lane ^ 1 (group bases are even), gathered by shuffle.
                    const double pr = x + dpp_move<0xb1>(x);
                    double p[W / 2];
#pragma unroll
                    for (int k = 0; k < W / 2; ++k) p[k] = __shfl(pr, first_lane + 2 * k);
                    double total = 0.0;
#pragma unroll
                    for (int r0 = 0; r0 < W / 2; r0 += 8) {                 // a row of sixteen lanes = eight pairs
                        double q[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int a = r0 + 2 * k;
                            q[k] = a + 1 < W / 2 ? p[a] + p[a + 1] : (a < W / 2 ? p[a] : 0.0);
                        }
                        const double lo = r0 + 2 < W / 2 ? q[0] + q[1] : q[0];
                        const double hi = r0 + 6 < W / 2 ? q[2] + q[3] : q[2];
                        const double row = r0 + 4 < W / 2 ? lo + hi : lo;
                        total = r0 == 0 ? row : total + row;
                    }
                    return total;
                }
            };
            double res = group_sum((to_d && gl < nj) ? exp_core(fma(y, fma(y, d.A0, d.b0), d.c0 + m), sh_tab) : 0.0);
            const double sum1 = group_sum((to_d && d.ns > 1 && gl < nj) ? exp_core(fma(y, fma(y, d.A1, d.b1), d.c1 + m), sh_tab) : 0.0);
            if (d.ns > 1 && gl == 1) res = sum1;
            if (to_d && gl < d.ns) f2v[(int64_t)d.e_d * S + gl] = res > 0.0 ? log_table(res, sh_log) : -700.0;
        }
        asm volatile("" :: "v"(touch));
    }
}

// CQ edges (cq.hpp): factors whose log potential is quadratic in two continuous arguments for every state of a discrete one --
// the ternary formulas of the reference's hybrid MLNs, x[0] * eq_op(x[1], x[2]) -- with more than one hidden partner:
//   type 1 (MIX)    continuous target x, hidden discrete partner z (S states), continuous partner y (hidden, or folded):
//                   exp(message) = sum_s sum_j exp(a_sj + b_sj x + k_s x^2),  a_sj = (ay_s y_j + by_s) y_j + c_s + m_y[j] + m_z[s]
//                   -- S heavy-style sums over the same staged particles, one per state, added before the log
//   type 2 (JOINT)  discrete target with S states, two hidden continuous partners x (lane = particle i) and y (staged):
//                   exp(message)[s] = sum_i sum_j exp(a_sj + b_sj x_i + (k_s x_i^2 + m_x[i])) -- the heavy kernel's particle round
//                   with the lane's own message folded into its per-point constant, then a wave reduction over i
// Same term loop as the heavy kernel (fast_accumulate_floor: 8 fp64 + 2 int32 VALU instructions per term).
struct CqDesc {
    int32_t e, tv, type, S;          // edge, target variable, 1 = MIX / 2 = JOINT, coefficient sets
    int32_t np, T, gb, yv;           // target particles (JOINT: = S states), grid points, grid base; staged partner's variable
    int32_t yce, ny, zv, zce;        // staged partner's v2f row and particles (1: observed / absent); z: variable and v2f row (MIX: -1 = none)
    int32_t nz, pad;                 // MIX: states of z; JOINT: particles of the lane-side partner;  pad: 1 = the target's integral points are a uniform grid
    double yval;                     // NaN: the staged partner is hidden
    double coef[LHVI_CQ_MAX_STATES][6];   // per set (ay, by, c, axy, bx, kx)
};
static_assert(sizeof(CqDesc) == 2 * LHVI_PBP_DESC_BYTES, "CqDesc is part of the ABI (2 * LHVI_PBP_DESC_BYTES)");

__global__ void __launch_bounds__(BLOCK) pbp_describe_cq_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                               const int32_t* __restrict__ edges, int count,
                                                               CqDesc* __restrict__ out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const int e = edges[i];
    CqInfo ci;
    const int route = cq_analyze(g, pots, s.np, s.n, e, ci);
    CqDesc d;
    d.e = e; d.tv = g.edge_var[e];
    d.type = route == CQ_MIX ? 1 : (route == CQ_JOINT ? 2 : 0);       // 0: not a CQ edge -- the kernel skips the entry
    d.S = ci.S;
    const int dom = g.var_dom[d.tv];
    d.np = s.np[d.tv];
    d.gb = g.dom_ptr[dom];
    d.T = g.dom_cont[dom] ? g.dom_ptr[dom + 1] - d.gb : 0;
    d.yv = ci.yv; d.yce = ci.yce; d.ny = ci.ny; d.zv = ci.zv; d.zce = ci.zce; d.nz = ci.nz; d.pad = 0;
    if (d.type == 1 && d.T >= 2) {                        // uniform integral-point grid (as in make_fast_desc): word 13 = 1
        const double x0 = g.dom_val[d.gb], xl = g.dom_val[d.gb + d.T - 1];
        const double h = (xl - x0) / (double)(d.T - 1);
        const double tol = 1.8e-15 * fmax(fabs(x0), fabs(xl));
        bool uniform = h > 0.0 && h < __builtin_huge_val();
        for (int t = 0; t < d.T && uniform; ++t) uniform = fabs(g.dom_val[d.gb + t] - fma((double)t, h, x0)) <= tol;
        d.pad = uniform ? 1 : 0;
    }
    d.yval = ci.yval;
    for (int k = 0; k < LHVI_CQ_MAX_STATES; ++k) {
        const bool on = d.type != 0 && k < ci.S;
        d.coef[k][0] = on ? ci.ay[k] : 0.0; d.coef[k][1] = on ? ci.by[k] : 0.0; d.coef[k][2] = on ? ci.c[k] : 0.0;
        d.coef[k][3] = on ? ci.axy[k] : 0.0; d.coef[k][4] = on ? ci.bx[k] : 0.0; d.coef[k][5] = on ? ci.kx[k] : 0.0;
    }
    out[i] = d;
}

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) pbp_f2v_cq_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ v2f,
                                                          double* __restrict__ f2v, const CqDesc* __restrict__ descs,
                                                          int nitems) {
    __shared__ AB sh_all[BLOCK / WAVE][WAVE];
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    AB* sh = sh_all[wid];
    const int n = s.n, S = s.n + s.T;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    for (int item = blockIdx.x * (BLOCK / WAVE) + wid; item < nitems; item += nwaves) {
        const CqDesc& d = descs[item];                      // wave-uniform address: scalar loads
        const int type = d.type, ny = d.ny, nst = d.S;
        if (type == 0) continue;
        double* out = f2v + (int64_t)d.e * S;
        // the staged partner's particles and message
        double y = d.yval, my = 0.0;
        if (is_hidden(d.yval)) { y = 0.0; if (lane < ny) { y = s.old_particles[(int64_t)d.yv * n + lane]; my = v2f[(int64_t)d.yce * n + lane]; } }
        if (type == 1) {
            const int np = d.np;
            double mz = 0.0;                                  // the discrete partner's message at its states (lane = state)
            if (d.zce >= 0 && lane < nst) mz = v2f[(int64_t)d.zce * n + lane];
            // integral points by the uniform-grid recurrence (as in the heavy kernel), once per state, when every exponent of
            // every state stays far inside the double range over the whole grid; otherwise they join the direct rounds
            bool grid_path = grid_eligible(d.pad, ny, d.T) && d.T <= 64 && !(s.flags & (LHVI_PBP_SKIP_TERMS | LHVI_PBP_NO_GRID));   // (two batches of sums per lane here)
            double gx0 = 0.0, gh = 0.0;
            if (grid_path) {
                gx0 = g.dom_val[d.gb];
                gh = (g.dom_val[d.gb + d.T - 1] - gx0) / (double)(d.T - 1);
                const double X = fmax(fabs(gx0), fabs(fma((double)(d.T - 1), gh, gx0)));
                for (int st = 0; st < nst; ++st) {
                    const double a = (d.coef[st][0] * y + d.coef[st][1]) * y + d.coef[st][2] + my + readlane_f64(mz, st);
                    const double b = d.coef[st][3] * y + d.coef[st][4];
                    const double bound = fma(fabs(b) + fabs(d.coef[st][5]) * X, X, fabs(a));
                    if (__ballot(lane < ny && !(bound < GRID_MAX_EXPONENT))) grid_path = false;
                }
            }
            const int npts = grid_path ? np : np + d.T;       // output points of the direct rounds
            // this lane's output points of the (at most two) rounds
            double xr[2] = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int rem = npts - 64 * r;
                if (rem > 0) {
                    const int pl = lane & ((1 << round_log2_width(rem)) - 1), pp = 64 * r + pl;
                    if (pl < rem) xr[r] = pp < np ? s.particles[(int64_t)d.tv * n + pp] : g.dom_val[d.gb + pp - np];
                }
            }
            double tot[2] = {0.0, 0.0};
            double totg[2] = {0.0, 0.0};                      // grid path: sums at this lane's point of each batch of 32 points
            for (int st = 0; st < nst; ++st) {
                const double ay = d.coef[st][0], by = d.coef[st][1], c = d.coef[st][2], axy = d.coef[st][3], bx = d.coef[st][4],
                             kx = d.coef[st][5];
                const double ms = readlane_f64(mz, st);
                AB mine;
                mine.a = PAD_LOG_TERM; mine.b = 0.0;             // padding: underflows to exactly 0 whatever the point's own constant adds
                if (lane < ny) { mine.a = (ay * y + by) * y + c + my + ms; mine.b = axy * y + bx; }
                LHVI_WAVE_SYNC();
                {
                    AB scaled;                             // the term loops read the records in units of the table step
                    scaled.a = mine.a * LHVI_EXP_INV_STEP; scaled.b = mine.b * LHVI_EXP_INV_STEP;
                    sh[lane] = scaled;
                }
                LHVI_WAVE_SYNC();
                if (grid_path) {
                    double gv = exp_core(fma(mine.b, gx0, mine.a), sh_tab);
                    const double q = exp_core(mine.b * gh, sh_tab);
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) {
                        if (32 * tb < d.T) {
                            const double sum = grid_sums32(gv, q, lane);
                            const double xt = fma((double)(32 * tb + grid_owned_point(lane)), gh, gx0);
                            totg[tb] = fma(sum, exp_core(kx * xt * xt, sh_tab), totg[tb]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int rem = npts - 64 * r;
                    if (rem > 0) {
                        const int lw = round_log2_width(rem);
                        const int width = 1 << lw, split = 64 >> lw, sub = lane >> lw, pl = lane & (width - 1);
                        const double X1 = pl < rem ? xr[r] : 0.0, C = kx * X1 * X1;
                        const int chunk = (ny + split - 1) >> (6 - lw);
                        double acc = fast_accumulate_floor<4>(sh + sub * chunk, sh_tab, chunk, X1, C);
                        for (int off = width; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
                        tot[r] += acc;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int rem = npts - 64 * r;
                if (rem > 0) {
                    const int lw = round_log2_width(rem);
                    const int sub = lane >> lw, pl = lane & ((1 << lw) - 1), p = 64 * r + pl;
                    if (pl < rem && sub == 0) out[p < np ? p : n + (p - np)] = tot[r] > 0.0 ? log_table(tot[r], sh_log) : -700.0;
                }
            }
            if (grid_path) {
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    const int t = 32 * tb + grid_owned_point(lane);
                    if (t < d.T && !(lane & 1)) out[n + t] = totg[tb] > 0.0 ? log_table(totg[tb], sh_log) : -700.0;
                }
            }
        } else {
            const int nx = d.nz;
            double x = 0.0, mx = 0.0;
            if (lane < nx) { x = s.old_particles[(int64_t)d.zv * n + lane]; mx = v2f[(int64_t)d.zce * n + lane]; }
            double res = 0.0;
            for (int st = 0; st < nst; ++st) {
                const double ay = d.coef[st][0], by = d.coef[st][1], c = d.coef[st][2], axy = d.coef[st][3], bx = d.coef[st][4],
                             kx = d.coef[st][5];
                AB mine;
                mine.a = PAD_LOG_TERM; mine.b = 0.0;
                if (lane < ny) { mine.a = (ay * y + by) * y + c + my; mine.b = axy * y + bx; }
                mine.a *= LHVI_EXP_INV_STEP; mine.b *= LHVI_EXP_INV_STEP;      // (records in units of the table step)
                LHVI_WAVE_SYNC();
                sh[lane] = mine;
                LHVI_WAVE_SYNC();
                double acc = fast_accumulate_floor<4>(sh, sh_tab, ny, x, fma(kx * x, x, mx));
                acc = wave_sum(lane < nx ? acc : 0.0);
                if (lane == st) res = acc;
            }
            if (lane < nst) out[lane] = res > 0.0 ? log_table(res, sh_log) : -700.0;
        }
    }
}

// test hook: y[i] = exp_core(x[i])
__global__ void __launch_bounds__(BLOCK) debug_exp_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    __shared__ double sh_tab[EXP_TAB_N];
    load_exp_table(sh_tab);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) y[i] = exp_core(x[i], sh_tab);
}

// test hook: y[i] = log_pos(x[i])
__global__ void __launch_bounds__(BLOCK) debug_log_kernel(const double* __restrict__ x, double* __restrict__ y, int64_t n, int which) {
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) y[i] = which == 0 ? log_table(x[i], sh_log) : log_pos(x[i]);
}

// test hook: y[i] = exp(x[i] + c[i]) through the accumulating form of the f2v term loop
__global__ void __launch_bounds__(BLOCK) debug_exp_acc_kernel(const double* __restrict__ x, const double* __restrict__ c,
                                                             double* __restrict__ y, int64_t n) {
    __shared__ double sh_tab[EXP_TAB_N];
    load_exp_table(sh_tab);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const ExpShift sft = exp_shift(c[i]);
        y[i] = exp_accumulate(0.0, x[i], sft.magic, sh_tab) * sft.scale;
    }
}

// test hook: the same through the floor form (s = x / step formed here, as the staging code does for a record)
__global__ void __launch_bounds__(BLOCK) debug_exp_acc_floor_kernel(const double* __restrict__ x, const double* __restrict__ c,
                                                                   double* __restrict__ y, int64_t n) {
    __shared__ double sh_tab[EXP_TAB_N];
    load_exp_table(sh_tab);
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const ExpShiftFloor sft = exp_shift_floor(c[i]);
        const double sv = x[i] * LHVI_EXP_INV_STEP;
        round_down_on();
        const double acc = exp_accumulate_floor(0.0, sv, sft.magic, sh_tab);
        round_down_off();
        y[i] = acc * sft.scale;
    }
}

__global__ void __launch_bounds__(BLOCK) pbp_classify_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s, uint8_t* __restrict__ cls) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e < g.E) cls[e] = (uint8_t)classify_edge(g, pots, s, e);
}

// GENERIC edges: lane = output point, sequential joint loop per lane.  A wave serves 64 >> pts_log2 edges at once:
// 2^pts_log2 lanes per edge (the host picks the smallest power of two that covers the largest point count in the work
// list, so the 2-point messages of discrete x discrete table factors pack 32 edges into a wave); edges with more than
// 64 points loop.
template <bool INTERP>
__global__ void __launch_bounds__(BLOCK) pbp_f2v_generic_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                               const double* __restrict__ v2f, double* __restrict__ f2v,
                                                               int pts_log2) {
    const int lane = threadIdx.x & 63;
    const int per_wave = 64 >> pts_log2;                    // edges per wave
    const int slot = lane >> pts_log2, pl = lane & ((1 << pts_log2) - 1);
    const int nitems = s.generic_edges ? s.n_generic : g.E;
    const int ngroups = (nitems + per_wave - 1) / per_wave;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    for (int grp = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6); grp < ngroups; grp += nwaves) {
        const int item = grp * per_wave + slot;
        if (item >= nitems) continue;
        const int e = s.generic_edges ? s.generic_edges[item] : item;
        if (classify_edge(g, pots, s, e) != EDGE_GENERIC) continue;
        const int tv = g.edge_var[e];
        const int n = s.n, S = s.n + s.T;
        const int d = g.var_dom[tv];
        const int np = s.np[tv];
        const int gb = g.dom_ptr[d];
        const int T = g.dom_cont[d] ? g.dom_ptr[d + 1] - gb : 0;
        const int npts = np + T;
        double* out = f2v + (int64_t)e * S;
        for (int p = pl; p < npts; p += (1 << pts_log2)) {
            const double x = p < np ? s.particles[(int64_t)tv * n + p] : g.dom_val[gb + p - np];
            const int xi = p < np ? p : p - np;
            out[p < np ? p : n + (p - np)] = f2v_point_generic<INTERP>(g, pots, s, v2f, s.old_particles, e, x, xi);
        }
    }
}

// belief_rv(x) = sum_f message_f_to_rv(x, f, rv, sample) at arbitrary points (EPBP:196-202; HLBP:313-317)
__global__ void __launch_bounds__(BLOCK) pbp_belief_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                          const double* __restrict__ v2f, int nq,
                                                          const int32_t* __restrict__ qvar, int npts,
                                                          const double* __restrict__ x, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)nq * npts) return;
    const int v = qvar[i / npts];
    const double xv = x[i];
    const int xi = state_index(g, v, xv);
    double res = 0.0;
    for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) {
        const int e = g.var_edge[k];
        const double m = f2v_point_generic(g, pots, s, v2f, s.particles, e, xv, xi);
        res += g.edge_count ? m * g.edge_count[e] : m;    // a lifted edge stands for `count` ground factors
    }
    out[i] = res;
}

// message_f_to_rv(x, f, rv, sample) for explicit (edge, point) pairs: the building block of belief_rv_query when the
// caller walks a GROUND variable's factors itself (HLBP:313-317)
__global__ void __launch_bounds__(BLOCK) pbp_edge_points_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                               const double* __restrict__ v2f, int nq,
                                                               const int32_t* __restrict__ qedge, int npts,
                                                               const double* __restrict__ x, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)nq * npts) return;
    const int e = qedge[i / npts];
    const double xv = x[i];
    out[i] = f2v_point_generic(g, pots, s, v2f, s.particles, e, xv, state_index(g, g.edge_var[e], xv));
}

// ---------------------------------------------------------------------------------------------
// map(rv) for every query row at once (EPBP.map EPBP:377-394; HLBP.map HLBP:405-424).  A row is either a variable of the
// solver's graph (qvar: its incident edges, count-weighted on a lifted graph) or an explicit list of (edge, multiplicity)
// pairs (qptr / qedge / qmult: a GROUND variable's factors on the lifted graph, belief_rv_query HLBP:313-317).
struct QueryRows {
    const int32_t* qvar; const int64_t* qptr; const int32_t* qedge; const double* qmult;
};

__device__ double query_log_belief(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_pbp_t& s, const double* __restrict__ v2f,
                                   const QueryRows& rows, int64_t i, int v, double x) {
    double res = 0.0;
    if (rows.qptr) {
        for (int64_t k = rows.qptr[i]; k < rows.qptr[i + 1]; ++k) {
            const int e = rows.qedge[k];
            res += f2v_point_generic(g, pots, s, v2f, s.particles, e, x, state_index(g, g.edge_var[e], x)) * rows.qmult[k];
        }
    } else {
        const int xi = state_index(g, v, x);
        for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) {
            const int e = g.var_edge[k];
            const double m = f2v_point_generic(g, pots, s, v2f, s.particles, e, x, xi);
            res += g.edge_count ? m * g.edge_count[e] : m;
        }
    }
    return res;
}

// One thread per row runs the whole of scipy.optimize.fminbound (SciPy's `_minimize_scalar_bounded`: Brent's golden-section /
// parabolic-interpolation minimiser on [lo, hi], what the reference's map() calls with its defaults xtol = 1e-5, maxfun = 500)
// on f(x) = -belief_rv(x): the same decisions in the same order, in IEEE double without contraction, so that every row steps
// through the reference's iterates.  A discrete row returns the first state with the largest belief (`max` over a dict in
// insertion order).  nfev [nq] (optional) receives the number of function evaluations.
__global__ void __launch_bounds__(BLOCK) pbp_map_brent_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                             const double* __restrict__ v2f, int64_t nq, QueryRows rows,
                                                             const int32_t* __restrict__ row_var,
                                                             double xatol_third, int maxfun, double* __restrict__ xout,
                                                             double* __restrict__ fout, int32_t* __restrict__ nfev) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nq) return;
    const int v = row_var[i];                      // a variable of g whose domain the row searches
    const int d = g.var_dom[v];
    if (!g.dom_cont[d]) {
        double best = 0.0, bestv = -__builtin_huge_val();
        int evals = 0;
        for (int k = g.dom_ptr[d]; k < g.dom_ptr[d + 1]; ++k, ++evals) {
            const double val = query_log_belief(g, pots, s, v2f, rows, i, v, g.dom_val[k]);
            if (k == g.dom_ptr[d] || val > bestv) { bestv = val; best = g.dom_val[k]; }
        }
        xout[i] = best;
        if (fout) fout[i] = bestv;
        if (nfev) nfev[i] = evals;
        return;
    }
    const double sqrt_eps = 0x1.fda324be34921p-27;         // sqrt(2.2e-16)
    const double golden_mean = 0x1.8722191a02d60p-2;       // 0.5 * (3 - sqrt(5))
    double a = g.dom_lo[d], b = g.dom_hi[d];
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = -query_log_belief(g, pots, s, v2f, rows, i, v, x);
    int num = 1;
    double fu = __builtin_huge_val();
    double ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol_third;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {                               // parabolic fit
            golden = false;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) {
                    const double dm = xm - xf;
                    const double si = (dm > 0.0 ? 1.0 : (dm < 0.0 ? -1.0 : 0.0)) + (dm == 0.0 ? 1.0 : 0.0);
                    rat = tol1 * si;
                }
            } else {
                golden = true;
            }
        }
        if (golden) {
            e = xf >= xm ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        const double si = (rat > 0.0 ? 1.0 : (rat < 0.0 ? -1.0 : 0.0)) + (rat == 0.0 ? 1.0 : 0.0);
        x = xf + si * fmax(fabs(rat), tol1);
        fu = -query_log_belief(g, pots, s, v2f, rows, i, v, x);
        ++num;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol_third;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) break;
    }
    xout[i] = xf;
    if (fout) fout[i] = -fx;
    if (nfev) nfev[i] = num;
}

// The normaliser of EPBP.belief (EPBP:325-328: scipy.integrate.quad of e ** belief_rv over [lo - 20, hi + 20]) for every query
// row at once, a thread each: QUADPACK's 21-point Gauss-Kronrod rule (dqk21, with its error estimate) inside the globally adaptive
// bisection of dqage -- the interval with the largest error estimate is halved until the summed estimate meets
// max(epsabs, epsrel |result|), at most `limit` intervals.  scipy's quad (dqagse) runs the same rule and the same bisection and
// adds the epsilon-algorithm extrapolation, so both answers lie within the requested tolerance (1.49e-8) of the integral:
// they agree to ~1e-7 relative (tests/test_gpu_pbp.py pins 1e-6).  status: 0 converged, 1 interval limit reached,
// 3 the integrand overflowed (e ** belief_rv = inf: the reference raises OverflowError there).
constexpr int QUAD_LIMIT = 50;

struct Gk21 { double result, abserr, resabs, resasc; bool overflow; };

template <typename F>
__device__ Gk21 gk21(F&& f, double a, double b) {
    const double xgk[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
                            0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
                            0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                            0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
                            0.294392862701460198131126603103866, 0.148874338981631210884826001129720, 0.0};
    const double wgk[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
                            0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
                            0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                            0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
                            0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
                            0.149445554002916905664936468389821};
    const double wg[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
                          0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
                          0.295524224714752870173815619188769};
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308;
    const double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double fv1[10], fv2[10];
    const double fc = f(centr);
    double resg = 0.0, resk = wgk[10] * fc, resabs = fabs(resk);
    for (int j = 0; j < 5; ++j) {
        const int jtw = 2 * j + 1;
        const double absc = hlgth * xgk[jtw];
        const double f1 = f(centr - absc), f2 = f(centr + absc);
        fv1[jtw] = f1; fv2[jtw] = f2;
        resg += wg[j] * (f1 + f2);
        resk += wgk[jtw] * (f1 + f2);
        resabs += wgk[jtw] * (fabs(f1) + fabs(f2));
    }
    for (int j = 0; j < 5; ++j) {
        const int jtwm1 = 2 * j;
        const double absc = hlgth * xgk[jtwm1];
        const double f1 = f(centr - absc), f2 = f(centr + absc);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        resk += wgk[jtwm1] * (f1 + f2);
        resabs += wgk[jtwm1] * (fabs(f1) + fabs(f2));
    }
    const double reskh = resk * 0.5;
    double resasc = wgk[10] * fabs(fc - reskh);
    for (int j = 0; j < 10; ++j) resasc += wgk[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    Gk21 r;
    r.result = resk * hlgth;
    r.resabs = resabs * dhlgth;
    r.resasc = resasc * dhlgth;
    r.abserr = fabs((resk - resg) * hlgth);
    if (r.resasc != 0.0 && r.abserr != 0.0) r.abserr = r.resasc * fmin(1.0, pow(200.0 * r.abserr / r.resasc, 1.5));
    if (r.resabs > uflow / (50.0 * epmach)) r.abserr = fmax((epmach * 50.0) * r.resabs, r.abserr);
    r.overflow = !(fabs(resk) < __builtin_huge_val());
    return r;
}

__global__ void __launch_bounds__(BLOCK) pbp_quad_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                        const double* __restrict__ v2f, int64_t nq, QueryRows rows,
                                                        const int32_t* __restrict__ row_var, const double* __restrict__ lo,
                                                        const double* __restrict__ hi, double epsabs, double epsrel,
                                                        double* __restrict__ zout, double* __restrict__ errout,
                                                        int32_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nq) return;
    const int v = row_var[i];
    auto f = [&](double x) { return exp(query_log_belief(g, pots, s, v2f, rows, i, v, x)); };
    double alist[QUAD_LIMIT], blist[QUAD_LIMIT], rlist[QUAD_LIMIT], elist[QUAD_LIMIT];
    Gk21 r = gk21(f, lo[i], hi[i]);
    int n = 1, st = 0;
    alist[0] = lo[i]; blist[0] = hi[i]; rlist[0] = r.result; elist[0] = r.abserr;
    double area = r.result, errsum = r.abserr;
    bool overflow = r.overflow;
    double errbnd = fmax(epsabs, epsrel * fabs(area));
    const bool done0 = (r.abserr <= errbnd && r.abserr != r.resabs) || r.abserr == 0.0;
    if (!done0 && !overflow) {
        st = 1;
        while (n < QUAD_LIMIT) {
            int m = 0;
            for (int k = 1; k < n; ++k) if (elist[k] > elist[m]) m = k;
            const double a1 = alist[m], b2 = blist[m], mid = 0.5 * (a1 + b2);
            const Gk21 r1 = gk21(f, a1, mid), r2 = gk21(f, mid, b2);
            overflow = overflow || r1.overflow || r2.overflow;
            if (overflow) break;
            errsum += (r1.abserr + r2.abserr) - elist[m];
            area += (r1.result + r2.result) - rlist[m];
            blist[m] = mid; rlist[m] = r1.result; elist[m] = r1.abserr;
            alist[n] = mid; blist[n] = b2; rlist[n] = r2.result; elist[n] = r2.abserr;
            ++n;
            errbnd = fmax(epsabs, epsrel * fabs(area));
            if (errsum <= errbnd) { st = 0; break; }
        }
    }
    double total = 0.0;
    for (int k = 0; k < n; ++k) total += rlist[k];
    zout[i] = overflow ? __builtin_nan("") : total;
    if (errout) errout[i] = errsum;
    if (status) status[i] = overflow ? 3 : st;
}

// ---------------------------------------------------------------------------------------------
// gaussian_division (EPBP:43-47)
__device__ __forceinline__ void gdiv(double a0, double a1, double b0, double b1, double& mu, double& sig) {
    sig = a1 * b1 / (b1 - a1);
    mu = (a0 * (b1 + sig) - b0 * sig) / b1;
}

// update_proposal (EPBP:83-154; HLBP:100-171): one wavefront per continuous hidden variable.  With T <= 32 integral
// points the wave works on four incident edges at once (16 lanes each), with T <= 64 on two, otherwise on one; the
// T-point moments are DPP reductions inside the lane group.  Each group accumulates the information-form sum of its edges; the groups
// are folded at the end (summation order differs from the reference's edge order by rounding only).
// ph_out != nullptr: sharded mode -- write the local information-form sums instead of q (lhvi_pbp_proposal_partial)
// EP: compiled with / without the cavity branch -- the 'simple' rule needs fewer registers (the launcher picks by s.flags)
template <bool EP>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(7, 8))) pbp_proposal_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                            double* __restrict__ eta, double* __restrict__ q,
                                                            double* __restrict__ ph_out) {
    const int lane = threadIdx.x & 63;
    const int n = s.n, S = s.n + s.T;
    int v, gb, T, lo, hi, slot = -1;
    int e4[4] = {0, 0, 0, 0};                              // the first four incident edges (descriptor path)
    const bool listed = s.prop_desc != nullptr;
    if (listed) {
        // one 32-byte record per hidden continuous variable instead of the chain var_value / var_dom -> dom_* / var_ptr ->
        // var_edge: one scalar load, then the message loads; no wave is spent on an observed or discrete variable
        const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
        if (item >= s.n_prop_desc) return;
        const int32_t* rec = s.prop_desc + 8 * (int64_t)item;
        v = rec[0]; gb = rec[2]; T = rec[3];
        if (rec[1] < 0) {
            // a slice of a hub variable's row: -rec[1] entries from position rec[4] of the row; the slice's information-form
            // sums go to slot rec[5] of prop_partial and pbp_proposal_hub_finish_kernel adds the slices up
            slot = rec[5];
            lo = g.var_ptr[v] + rec[4];
            hi = lo - rec[1];
        } else {
            // (the row's position in var_edge is needed only when one pass of four edge groups does not cover it)
            lo = (rec[1] > 4 || T > 32 || g.edge_count) ? g.var_ptr[v] : 0;       // (a lifted graph sums the row's counts below)
            hi = lo + rec[1];
            e4[0] = rec[4]; e4[1] = rec[5]; e4[2] = rec[6]; e4[3] = rec[7];
        }
    } else {
        v = __builtin_amdgcn_readfirstlane(var_first(s) + blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
        if (v >= var_limit(g, s)) return;
        const int d = g.var_dom[v];
        if (!is_hidden(g.var_value[v]) || !g.dom_cont[d]) return;
        gb = g.dom_ptr[d]; T = g.dom_ptr[d + 1] - gb;
        lo = g.var_ptr[v]; hi = g.var_ptr[v + 1];
    }
    double total = 0.0;
    if (s.var_degree) total = s.var_degree[v];
    else {
        const int r0 = slot >= 0 ? g.var_ptr[v] : lo, r1 = slot >= 0 ? g.var_ptr[v + 1] : hi;      // (a slice: the whole row's total)
        if (g.edge_count) for (int k = r0; k < r1; ++k) total += g.edge_count[g.var_edge[k]];
        else total = (double)(r1 - r0);
    }
    const double min_sig = total * s.var_threshold;
    const double q0 = s.q[2 * v], q1 = s.q[2 * v + 1];
    // lane groups of 16 / 32 / 64 lanes, one incident edge per group and pass: with T <= 32 integral points a degree-4
    // variable is done in one pass (two points per lane) and the moment sums are 4-step row reductions
    const int width = T <= 32 ? 16 : (T <= 64 ? 32 : 64);
    const int groups = 64 / width;
    const int grp = lane / width, tl = lane % width;
    double pm = 0.0, ps = 0.0;
    for (int k0 = lo; k0 < hi; k0 += groups) {
        const int k = k0 + grp;
        const bool live = k < hi;
        int e;
        if (listed && slot < 0 && k0 == lo && groups == 4) {           // first pass of four groups: the edges are in the record
            e = grp == 0 ? e4[0] : (grp == 1 ? e4[1] : (grp == 2 ? e4[2] : e4[3]));
            if (!live) e = e4[0];
        } else {
            e = live ? g.var_edge[k] : g.var_edge[k0];
        }
        const double* msg = f2v + (int64_t)e * S + n;
        const double b0 = eta[2 * e], b1 = eta[2 * e + 1];
        const bool use_cav = EP && !(q1 >= b1);
        double c0 = 0.0, c1 = 1.0;
        if (use_cav) gdiv(q0, q1, b0, b1, c0, c1);
        const double csd = sqrt(c1);
        double z = 0.0, a = 0.0, b = 0.0;
        for (int t = tl; t < T; t += width) {
            const double xg = g.dom_val[gb + t];
            double w = exp(msg[t]);
            if (use_cav) w = w * norm_pdf_std(xg, c0, csd);
            z += w; a += w * xg; b += w * (xg * xg);
        }
        if (width == 16) { z = dpp_row_reduce(z, SumOp()); a = dpp_row_reduce(a, SumOp()); b = dpp_row_reduce(b, SumOp()); }
        else if (width == 32) { z = dpp_half_reduce(z, SumOp(), lane); a = dpp_half_reduce(a, SumOp(), lane); b = dpp_half_reduce(b, SumOp(), lane); }
        else { z = wave_sum(z); a = wave_sum(a); b = wave_sum(b); }
        const double rz = rcp_newton(z);                  // z = 0 (every weight underflowed) -> nan -> the test below fails
        double mu = a * rz;
        double sig = b * rz - mu * mu;
        if (use_cav) { const double m0 = mu, m1 = sig; gdiv(m0, m1, c0, c1, mu, sig); }
        if (0.0 < sig && sig < __builtin_huge_val()) {
            sig = fmax(sig, min_sig);
            if (live && tl == 0) { eta[2 * e] = mu; eta[2 * e + 1] = sig; }
        } else {
            mu = b0; sig = b1;
        }
        if (live) {
            const double p = rcp_newton(sig);
            if (g.edge_count) { const double c = g.edge_count[e]; ps += p * c; pm += p * mu * c; }
            else { ps += p; pm += p * mu; }
        }
    }
    for (int off = width; off < 64; off <<= 1) { ps += __shfl_xor(ps, off); pm += __shfl_xor(pm, off); }
    if (ph_out) { if (lane == 0) { ph_out[2 * v] = ps; ph_out[2 * v + 1] = pm; } return; }
    if (slot >= 0) { if (lane == 0) { s.prop_partial[2 * slot] = ps; s.prop_partial[2 * slot + 1] = pm; } return; }
    ps = 1.0 / ps;
    if (lane == 0) { q[2 * v] = ps * pm; q[2 * v + 1] = ps; }
}

// q of the hub variables from their slices' sums, slices in row order (lhvi_pbp_t.prop_hub)
__global__ void __launch_bounds__(BLOCK) pbp_proposal_hub_finish_kernel(lhvi_pbp_t s, double* __restrict__ q) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= s.n_prop_hub) return;
    const int v = s.prop_hub[4 * i], first = s.prop_hub[4 * i + 1], count = s.prop_hub[4 * i + 2];
    double ps = 0.0, pm = 0.0;
    for (int k = 0; k < count; ++k) { ps += s.prop_partial[2 * (first + k)]; pm += s.prop_partial[2 * (first + k) + 1]; }
    ps = 1.0 / ps;
    q[2 * v] = ps * pm; q[2 * v + 1] = ps;
}

// q[v] from the local information-form sums plus the other ranks' (lhvi_pbp_proposal_finish)
__global__ void __launch_bounds__(BLOCK) pbp_proposal_finish_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ ph,
                                                                   double* __restrict__ q) {
    const int v = var_first(s) + blockIdx.x * BLOCK + threadIdx.x;
    if (v >= var_limit(g, s)) return;
    if (!is_hidden(g.var_value[v]) || !g.dom_cont[g.var_dom[v]]) return;
    double ps = ph[2 * v], pm = ph[2 * v + 1];
    if (s.bslot) {
        // boundary variable: add the ranks' information-form sums in ascending rank order (own sum at its own position),
        // so that every replica of the variable computes bit-identical q and therefore draws identical particles
        const int bs = s.bslot[v];
        if (bs >= 0 && (s.flags & LHVI_PBP_BOUNDARY_TOTALS)) {
            const double* row = s.recv + s.brow_off[s.brow_ptr[bs]] + s.n;     // the owner's finished sums (lhvi_pbp_boundary_reduce)
            ps = row[0]; pm = row[1];
        } else if (bs >= 0) {
            const double own_s = ps, own_m = pm;
            ps = 0.0; pm = 0.0;
            bool own_done = false;
            for (int r = s.brow_ptr[bs]; r < s.brow_ptr[bs + 1]; ++r) {
                if (!own_done && s.brow_peer[r] > s.rank) { ps += own_s; pm += own_m; own_done = true; }
                const double* row = s.recv + s.brow_off[r] + s.n;       // continuous variable: [n sums | 2 site sums]
                ps += row[0]; pm += row[1];
            }
            if (!own_done) { ps += own_s; pm += own_m; }
        }
    }
    ps = 1.0 / ps;
    q[2 * v] = ps * pm; q[2 * v + 1] = ps;
}

// one wavefront per boundary variable: row = [sum_local c * f2v[e][j] for j < n | ph[v]], written to each of the
// variable's slots of the send buffer (one per peer rank that also owns edges of it)
__global__ void __launch_bounds__(BLOCK) pbp_boundary_pack_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                                 const double* __restrict__ ph, int nb,
                                                                 const int32_t* __restrict__ bvars, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
    if (i >= nb) return;
    const int v = bvars[i];
    const int n = s.n, S = s.n + s.T;
    const int np = s.np[v];
    const int r0 = s.brow_ptr[i], r1 = s.brow_ptr[i + 1];
    // row width: n + 2 for a continuous variable (particle sums | site sums), np for a discrete one (its states only;
    // discrete variables have no proposal to exchange)
    const int W = g.dom_cont[g.var_dom[v]] ? n + 2 : np;
    for (int j = lane; j < W; j += 64) {
        double val = 0.0;
        if (j < np) {
            for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) {
                const int e = g.var_edge[k];
                const double m = f2v[(int64_t)e * S + j];
                val += g.edge_count ? m * g.edge_count[e] : m;
            }
        } else if (j >= n) val = ph[2 * v + (j - n)];
        for (int r = r0; r < r1; ++r) out[s.brow_off[r] + j] = val;
    }
}

// owner side of the reduce-to-owner exchange: one wavefront per owned boundary variable adds its sources (the ranks' rows in
// ascending rank order, host-listed) and writes the total to each of its destinations (own total row, one send row per replica)
__global__ void __launch_bounds__(BLOCK) pbp_boundary_reduce_kernel(int n_items, const int32_t* __restrict__ width,
                                                                   const int32_t* __restrict__ src_ptr, const int64_t* __restrict__ src_off,
                                                                   const int32_t* __restrict__ dst_ptr, const int64_t* __restrict__ dst_off,
                                                                   const double* __restrict__ in, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
    if (i >= n_items) return;
    const int W = width[i];
    for (int j = lane; j < W; j += 64) {
        double total = 0.0;
        for (int r = src_ptr[i]; r < src_ptr[i + 1]; ++r) total += in[src_off[r] + j];
        for (int r = dst_ptr[i]; r < dst_ptr[i + 1]; ++r) out[dst_off[r] + j] = total;
    }
}

// ---------------------------------------------------------------------------------------------
// Batched queries.  belief_rv(x) = sum over the variable's factors of message_f_to_rv(x)
// (EPBP:196-202): lhvi_pbp_f2v run with the query points in the place of the target particles tabulates every message
// at n points per variable; this kernel adds the rows up per variable (count-weighted on a lifted graph).
__global__ void __launch_bounds__(BLOCK) pbp_var_sum_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                           double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int n = s.n, S = s.n + s.T;
    if (i >= (int64_t)g.V * n) return;
    const int v = (int)(i / n), j = (int)(i % n);
    double acc = 0.0;
    if (is_hidden(g.var_value[v]) && j < s.np[v])
        for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) {
            const int e = g.var_edge[k];
            const double m = f2v[(int64_t)e * S + j];
            acc += g.edge_count ? m * g.edge_count[e] : m;
        }
    out[i] = acc;
}

// One step of the batched MAP search: per continuous hidden variable, the best of its n tabulated points and a new
// uniform grid of n points on the bracket around it [x_(i-1), x_(i+1)] (clamped to the previous grid's ends).
// Discrete variables keep their states as points; best[v] = the argmax point, best_val[v] = its log-belief.
__global__ void __launch_bounds__(BLOCK) pbp_refine_grid_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ logb,
                                                               double* __restrict__ x, double* __restrict__ best,
                                                               double* __restrict__ best_val) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= g.V) return;
    const int n = s.n;
    if (!is_hidden(g.var_value[v])) { best[v] = g.var_value[v]; best_val[v] = 0.0; return; }
    const int cnt = s.np[v];
    double* xr = x + (int64_t)v * n;
    const double* br = logb + (int64_t)v * n;
    int arg = 0;
    for (int j = 1; j < cnt; ++j)
        if (br[j] > br[arg]) arg = j;                      // first maximum, like the reference's argmax over a list
    best[v] = xr[arg];
    best_val[v] = br[arg];
    if (!g.dom_cont[g.var_dom[v]] || cnt < 3) return;
    const double lo = xr[arg > 0 ? arg - 1 : 0], hi = xr[arg < cnt - 1 ? arg + 1 : cnt - 1];
    const double step = (hi - lo) / (double)(cnt - 1);
    for (int j = 0; j < cnt; ++j) xr[j] = j == cnt - 1 ? hi : lo + step * j;
}

// uniform n-point grid on every continuous hidden variable's domain (discrete: the states), the start of the search
__global__ void __launch_bounds__(BLOCK) pbp_domain_grid_kernel(lhvi_graph_t g, lhvi_pbp_t s, double* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int n = s.n;
    if (i >= (int64_t)g.V * n) return;
    const int v = (int)(i / n), j = (int)(i % n);
    const int d = g.var_dom[v], cnt = s.np[v];
    double val = 0.0;
    if (is_hidden(g.var_value[v]) && j < cnt) {
        if (g.dom_cont[d]) val = j == cnt - 1 ? g.dom_hi[d] : g.dom_lo[d] + (g.dom_hi[d] - g.dom_lo[d]) / (double)(cnt - 1) * j;
        else val = g.dom_val[g.dom_ptr[d] + j];
    }
    x[i] = val;
}

// initial_proposal (EPBP:72-81; HLBP:89-98)
__global__ void __launch_bounds__(BLOCK) pbp_init_kernel(lhvi_graph_t g, lhvi_pbp_t s, double* __restrict__ eta,
                                                        double* __restrict__ q) {
    const int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= g.V) return;
    if (!is_hidden(g.var_value[v])) return;
    if (!g.dom_cont[g.var_dom[v]] && !(s.flags & LHVI_PBP_EPBP_DISCRETE)) return;
    q[2 * v] = 0.0; q[2 * v + 1] = 5.0;
    double total = 0.0;
    if (s.var_degree) total = s.var_degree[v];
    else for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) total += g.edge_count ? g.edge_count[g.var_edge[k]] : 1.0;
    for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) {
        const int e = g.var_edge[k];
        eta[2 * e] = 0.0; eta[2 * e + 1] = 5.0 * total;
    }
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG: Philox4x32-10 keyed by (seed), counter = (variable gid, particle j, iteration)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// cos(2 pi t), t in [0, 1): fold to w = distance to the nearest half turn's quarter, |2 pi w| <= pi/2, even series to a^22
// (remainder < 2e-17); the argument reduction is exact because t is
__device__ __forceinline__ double cos_turns(double t) {
    const double tt = fabs(t - rint(t));                        // [0, 0.5]
    const bool flip = tt > 0.25;
    const double w = flip ? 0.5 - tt : tt;                      // [0, 0.25]
    const double a = w * 6.283185307179586477;
    const double z = a * a;
    double p = LHVI_SCONST(-1.0 / 1124000727777607680000.0);    // -1/22!
    p = fma(p, z, LHVI_SCONST(1.0 / 2432902008176640000.0));    //  1/20!
    p = fma(p, z, LHVI_SCONST(-1.0 / 6402373705728000.0));      // -1/18!
    p = fma(p, z, LHVI_SCONST(1.0 / 20922789888000.0));         //  1/16!
    p = fma(p, z, LHVI_SCONST(-1.0 / 87178291200.0));           // -1/14!
    p = fma(p, z, LHVI_SCONST(1.0 / 479001600.0));              //  1/12!
    p = fma(p, z, LHVI_SCONST(-1.0 / 3628800.0));               // -1/10!
    p = fma(p, z, LHVI_SCONST(1.0 / 40320.0));                  //  1/8!
    p = fma(p, z, LHVI_SCONST(-1.0 / 720.0));                   // -1/6!
    p = fma(p, z, LHVI_SCONST(1.0 / 24.0));                     //  1/4!
    p = fma(p, z, -0.5);
    p = fma(p, z, 1.0);
    return flip ? -p : p;
}

// Two standard normal draws from one Philox block (key = seed, counter = (variable gid, block, iteration)): Box-Muller on two
// 53-bit uniforms gives r cos(2 pi u2) and r sin(2 pi u2).  Particle j of a variable takes block (j & 31) | (j >> 6 << 5) and the
// cosine (bit 5 of j clear) or the sine (set): particles j and j + 32 share a block, so that a wavefront can draw for two variables
// at once -- 32 blocks each -- whichever two they are (pbp_resample_uniq_kernel).
// log / sqrt / cos are the short routines above (absolute error < 1e-15 on z): a sampler needs reproducibility -- every
// rank runs this same code, so replicas of a boundary variable still draw identical particles -- not the last ulp.
// `logtab` = LDS copy of the log table (load_log_table), or nullptr for the table-free series
__device__ __forceinline__ void philox_normal_pair(uint64_t seed, uint64_t gid, uint32_t block, uint32_t iteration,
                                                   const LogRec* __restrict__ logtab, double& zc, double& zs) {
    uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), block, iteration};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t r0 = ((uint64_t)c[0] << 32) | c[1], r1 = ((uint64_t)c[2] << 32) | c[3];
    const double u1 = ((double)(r0 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(r1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double l = logtab ? log_table(u1, logtab) : log_pos(u1);
    const double rad = sqrt_pos(-2.0 * l);
    zc = rad * cos_turns(u2);
    zs = rad * cos_turns(u2 - 0.25);                          // sin(2 pi t) = cos(2 pi (t - 1/4))
}

__device__ __forceinline__ double philox_normal(uint64_t seed, uint64_t gid, uint32_t j, uint32_t iteration,
                                                const LogRec* __restrict__ logtab) {
    double zc, zs;
    philox_normal_pair(seed, gid, (j & 31u) | ((j >> 6) << 5), iteration, logtab, zc, zs);
    return (j & 32u) ? zs : zc;
}

// generate_sample (EPBP:61-70): clip(normal(q.mu, sqrt(q.var)), lo, hi); discrete rvs: the domain states
__global__ void __launch_bounds__(BLOCK) pbp_resample_kernel(lhvi_graph_t g, lhvi_pbp_t s, const int64_t* __restrict__ gid,
                                                            uint64_t seed, uint32_t iteration, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int n = s.n;
    if (i >= (int64_t)g.V * n) return;
    const int v = (int)(i / n), j = (int)(i % n);
    if (j >= s.np[v]) return;
    const int d = g.var_dom[v];
    if (!g.dom_cont[d]) { out[i] = g.dom_val[g.dom_ptr[d] + j]; return; }
    const double z = philox_normal(seed, gid ? (uint64_t)gid[v] : (uint64_t)v, (uint32_t)j, iteration, nullptr);
    const double x = s.q[2 * v] + sqrt(s.q[2 * v + 1]) * z;
    out[i] = fmin(fmax(x, g.dom_lo[d]), g.dom_hi[d]);
}

static int device_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

// grid of a wave-per-item persistent kernel: enough blocks for `per_cu` per CU, never more than the items need
static unsigned persistent_grid(int64_t items, int per_cu) {
    static const int cus = device_cus();
    const int64_t want = (items + BLOCK / WAVE - 1) / (BLOCK / WAVE), cap = (int64_t)cus * per_cu;
    return (unsigned)(want < cap ? want : cap);
}

static int blocks_per_cu(const void* kernel, int block = BLOCK) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 4;
    return nb > 8 ? 8 : nb;
}

// generate_sample + the first-occurrence mask in one pass (n <= 64): one wavefront per variable, lane = particle;
// the row never leaves registers and is broadcast lane by lane with v_readlane (exact equality test, like the dict keys)
constexpr int UNIQ_HASH_BITS = 14;

// LISTED: the caller's records (all hidden continuous: the paths for observed and discrete variables compile away)
template <bool LISTED>
__global__ void __launch_bounds__(BLOCK) pbp_resample_uniq_kernel(lhvi_graph_t g, lhvi_pbp_t s, const int64_t* __restrict__ gid,
                                                                 uint64_t seed, uint32_t iteration, double* __restrict__ out,
                                                                 uint8_t* __restrict__ uniq) {
    __shared__ LogRec sh_log[LOG_TAB_N];
    __shared__ uint32_t sh_bits[BLOCK / WAVE][2 << (UNIQ_HASH_BITS - 5)];      // per wave: the bitset, then its "hit twice" twin
    load_log_table(sh_log);
    for (int i = threadIdx.x; i < (BLOCK / WAVE) * (2 << (UNIQ_HASH_BITS - 5)); i += BLOCK) (&sh_bits[0][0])[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t* bits = sh_bits[threadIdx.x >> 6];
    uint32_t* twice = bits + (1 << (UNIQ_HASH_BITS - 5));
    const int n = s.n;
    const int nwaves = gridDim.x * (BLOCK / WAVE);
    // persistent waves (the table is loaded once per block), TWO variables per step: the Philox blocks and the Box-Muller radius are
    // the expensive part of a draw and a block yields two normals, so lanes 0-31 run the 32 blocks of the first variable, lanes
    // 32-63 those of the second, and one v_permlane32_swap per dword turns (cosines, sines) into the two variables' rows.
    // The pairs are the caller's list of hidden continuous variables two by two (s.resample_vars: nothing else is touched),
    // or neighbours of the variable range.
    constexpr bool listed = LISTED;
    const int vfirst = var_first(s), vend = var_limit(g, s);
    const int nvars = listed ? s.n_resample_vars : vend - vfirst;
    const int nitems = (nvars + 1) >> 1;
    constexpr int HALF_BITS = UNIQ_HASH_BITS - 1;           // each of the two rows of a step hashes into its own half of the bitsets
    // what a step needs to know about its two variables, fetched one step ahead (scalar loads: one record each from the caller's
    // list, or the chain np -> var_dom -> dom_cont / dom_lo / dom_hi) so that no step starts with a memory round trip
    struct Pair { int vv[2], cnt[2]; bool cont[2]; double lo[2], hi[2], mu[2], var[2]; };
    auto fetch = [&](int item) {
        Pair p;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * item + h;
            p.vv[h] = -1; p.cnt[h] = 0; p.cont[h] = false; p.lo[h] = 0.0; p.hi[h] = 0.0; p.mu[h] = 0.0; p.var[h] = 1.0;
            if (k < nvars) {
                if (listed) {
                    const int32_t* rec = s.resample_vars + 8 * (int64_t)k;
                    p.vv[h] = rec[0]; p.cnt[h] = rec[1]; p.cont[h] = true;
                    p.lo[h] = __hiloint2double(rec[3], rec[2]); p.hi[h] = __hiloint2double(rec[5], rec[4]);
                } else {
                    p.vv[h] = vfirst + k;
                    p.cnt[h] = s.np[p.vv[h]];
                    const int d = g.var_dom[p.vv[h]];
                    p.cont[h] = p.cnt[h] > 0 && g.dom_cont[d];
                    if (p.cont[h]) { p.lo[h] = g.dom_lo[d]; p.hi[h] = g.dom_hi[d]; }
                }
            }
        }
        return p;
    };
    auto fetch_q = [&](Pair& p) {       // second level: the proposals (their addresses come out of the first)
#pragma unroll
        for (int h = 0; h < 2; ++h) if (p.cont[h]) { p.mu[h] = s.q[2 * p.vv[h]]; p.var[h] = s.q[2 * p.vv[h] + 1]; }
    };
    int item = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
    if (item >= nitems) return;
    Pair ahead = fetch(item);
    fetch_q(ahead);
    for (; item < nitems; item += nwaves) {
        int vv[2], cnt[2];
        bool cont[2];
        double lo[2], hi[2], mu[2], sd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            vv[h] = ahead.vv[h]; cnt[h] = ahead.cnt[h]; cont[h] = ahead.cont[h]; lo[h] = ahead.lo[h]; hi[h] = ahead.hi[h];
            mu[h] = ahead.mu[h]; sd[h] = sqrt_pos(ahead.var[h]);
        }
        const bool more = item + nwaves < nitems;
        if (more) ahead = fetch(item + nwaves);
        double row[2] = {0.0, 0.0};
        if (cont[0] || cont[1]) {
            // (a half whose own variable needs no draw repeats the other's blocks: the swap below then leaves that row intact)
            const int mine = (lane >> 5) ? (cont[1] ? vv[1] : vv[0]) : (cont[0] ? vv[0] : vv[1]);
            double zc, zs;
#ifdef LHVI_DIAG_NO_PHILOX                                     // timing aid (scripts/diag/resample_time.py): never defined in the product build
            zc = 1e-3 * lane + mine; zs = -zc;
#else
            philox_normal_pair(seed, gid ? (uint64_t)gid[mine] : (uint64_t)mine, (uint32_t)(lane & 31), iteration, sh_log, zc, zs);
#endif
            // v_permlane32_swap a, b: a's upper half <-> b's lower half: a = (cos | sin) of the first variable, b of the second
            auto plo = __builtin_amdgcn_permlane32_swap(__double2loint(zc), __double2loint(zs), false, false);
            auto phi = __builtin_amdgcn_permlane32_swap(__double2hiint(zc), __double2hiint(zs), false, false);
            row[0] = __hiloint2double(phi[0], plo[0]);
            row[1] = __hiloint2double(phi[1], plo[1]);
        }
        if (more) fetch_q(ahead);                              // (the records asked for at the top of the step have arrived by now)
        // the rows themselves
        int xlo[2] = {0, 0}, xhi[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = vv[h], cn = cnt[h];
            if (v < 0) continue;
            if (!LISTED && cn == 0) {                              // observed: no particles; the mask row is all zero
                if (lane < n) uniq[(int64_t)v * n + lane] = 0;
            } else if (!LISTED && !cont[h]) {                      // discrete: the particles are the (distinct) states, no draw, no duplicates
                const int d = g.var_dom[v];
                if (lane < cn) out[(int64_t)v * n + lane] = g.dom_val[g.dom_ptr[d] + lane];
                if (lane < n) uniq[(int64_t)v * n + lane] = (uint8_t)(lane < cn);
            } else if (lane < cn) {
                // + 0.0: no -0, so that bitwise equality below is numeric equality
                const double x = fmin(fmax(fma(sd[h], row[h], mu[h]), lo[h]), hi[h]) + 0.0;
                out[(int64_t)v * n + lane] = x;
                xlo[h] = __double2loint(x); xhi[h] = __double2hiint(x);
            }
        }
#ifdef LHVI_DIAG_NO_UNIQ                                           // timing aid, as above
#pragma unroll
        for (int h = 0; h < 2; ++h) if (cont[h] && lane < n) uniq[(int64_t)vv[h] * n + lane] = (uint8_t)(lane < cnt[h]);
        continue;
#endif
        if (LISTED && (s.flags & LHVI_PBP_NO_UNIQ)) continue;      // particles only (ghost variables: nobody reads their masks here)
        if (!(cont[0] || cont[1])) continue;
        // first-occurrence masks of the (up to two) drawn rows, in step so that the LDS round trips are paid once.  Pass 1 only
        // asks "can two live particles be equal at all?": every lane sets the bit its low word hashes to in a wave-private LDS
        // bitset (8 Kbit per row) with a returning atomic OR; equal particles always meet in the same bit, different ones do with
        // probability 64^2 / 2 / 8192 = 25 % per variable (a false alarm costs the exact pass, nothing else).  The exact pass
        // compares the 64-bit patterns particle by particle and in practice runs for the variables with draws clipped to a bound.
        // Each lane clears its own words afterwards.
        // Two independent hashes into the same bitset.  The first is decisive: of two equal particles the lane whose returning
        // atomic OR executes second finds the bit set (whichever lane that is -- no ordering between lanes or between the two
        // instructions is assumed).  The second only rejects false alarms, and order-independently: a lane that finds its second
        // bit already set records it in the "twice" bitset, and after the wave has synchronised every lane asks whether its
        // second bit was hit more than once -- for equal particles it always was.  Suspect = first bit found set AND second bit
        // hit twice: different particles for ~1.5 % of the variables (one hash: 25 %)
        uint32_t *word[2], *word2[2], *tword2[2];
        uint32_t bit[2], bit2[2], old[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t hsh = (((uint32_t)xlo[h] * 0x9E3779B1u) >> (32 - HALF_BITS)) | ((uint32_t)h << HALF_BITS);
            const uint32_t hsh2 = ((((uint32_t)xhi[h] * 0x85EBCA6Bu) ^ ((uint32_t)xlo[h] * 0xC2B2AE35u)) >> (32 - HALF_BITS)) | ((uint32_t)h << HALF_BITS);
            word[h] = bits + (hsh >> 5); word2[h] = bits + (hsh2 >> 5); tword2[h] = twice + (hsh2 >> 5);
            bit[h] = 1u << (hsh & 31); bit2[h] = 1u << (hsh2 & 31);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (cont[h] && lane < cnt[h]) {
                old[h] = atomicOr(word[h], bit[h]);
                if (atomicOr(word2[h], bit2[h]) & bit2[h]) atomicOr(tword2[h], bit2[h]);
            }
        LHVI_WAVE_SYNC();
        uint64_t dup[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) dup[h] = __ballot(cont[h] && lane < cnt[h] && (old[h] & bit[h]) && (*tword2[h] & bit2[h]));
        LHVI_WAVE_SYNC();
#pragma unroll
        for (int h = 0; h < 2; ++h) if (cont[h] && lane < cnt[h]) { *word[h] = 0; *word2[h] = 0; *tword2[h] = 0; }
        LHVI_WAVE_SYNC();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!cont[h]) continue;
            const int cn = cnt[h];
            const uint64_t live = cn >= 64 ? ~0ull : ((1ull << cn) - 1);
            uint64_t dp = dup[h];
            if (dp) {
                // every group of equal particles contains a suspect lane (the one whose atomics came later), so walking the
                // suspects' values covers all groups: one step per distinct suspect value -- in practice the two domain bounds
                uint64_t todo = dp & live;
                dp = 0;
                while (todo) {
                    const int k = __builtin_ctzll(todo);
                    const int klo = __builtin_amdgcn_readlane(xlo[h], k), khi = __builtin_amdgcn_readlane(xhi[h], k);
                    const uint64_t same = __ballot(xlo[h] == klo && xhi[h] == khi) & live;    // every lane holding this value
                    dp |= same & (same - 1);                                                 // all but its first occurrence
                    todo &= ~same;
                }
            }
            const int u = (lane < cn) && !((dp >> lane) & 1);
            if (lane < n) uniq[(int64_t)vv[h] * n + lane] = (uint8_t)u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The three per-variable steps of a sweep in ONE pass over a variable's incident f -> v rows, for hidden continuous variables
// with few particles (np <= W = 16 or 32: the particle counts of the reference's demos, Demo/RGM/demo.py:20,
// RGMKLDivergence.py:54): message_rv_to_f + log_message_balance (EPBP:165-174,204-215; HLBP:182-191), update_proposal
// (EPBP:83-154; HLBP:100-171) and generate_sample + the first-occurrence mask (EPBP:61-70).  With one kernel per step such a
// variable's rows are fetched by three launches that each keep a handful of lanes busy and then wait out their own loads; here
// a lane group of W lanes owns the variable through all three (64 / W variables per wavefront), the new proposal never leaves
// registers between the second and the third, and the sweep has two launches and two passes over the rows less.
//   step 1 = the body of pbp_v2f_packed_kernel<W> (lane = particle);
//   step 2 = the body of pbp_proposal_kernel: an incident edge's T integral points over PW lanes (PW = 16 for T <= 32, 32 for
//            T <= 64: the widths that kernel picks), W / PW edges of the variable per pass, the information-form sums kept
//            per "edge index mod 64 / PW" and folded in that kernel's order;
//   step 3 = the draw of pbp_resample_uniq_kernel (Philox block j, cosine branch, for particle j < 32) and the exact
//            first-occurrence mask within the lane group.
// Same expressions, same reduction networks at the same lane positions, same order of additions: the same bits as the three
// kernels (tests/test_gpu_pbp.py::test_fused_variable_kernel_equals_the_three_kernels).  Records (lhvi_pbp_t.fused_desc):
// eight 32-bit words per variable -- 0 variable  1 incident edges  2 grid base in dom_val  3 T  4-5 dom_lo  6-7 dom_hi.
#ifndef LHVI_FUSED_WAVES
#define LHVI_FUSED_WAVES 4
#endif
constexpr int FUSED_CH = 8;          // edges of a row whose loads are in flight together
#ifndef LHVI_FUSED_NB
#define LHVI_FUSED_NB 2          // (scripts/diag/fused_batch.sh: 2 passes in flight at 4 waves/SIMD beat 4 passes, which spill)
#endif
template <int W, int PW, bool EP, bool R16>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LHVI_FUSED_WAVES, 8)))
pbp_var_fused_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v, double* __restrict__ v2f, double* __restrict__ eta,
                     double* __restrict__ q, const int64_t* __restrict__ gid, uint64_t seed, uint32_t iteration,
                     double* __restrict__ out, uint8_t* __restrict__ uniq, const int32_t* __restrict__ list, int count) {
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    __syncthreads();
    constexpr int G = WAVE / W, SL = W / PW;                 // variables per wavefront; edges of a variable per proposal pass
    const int lane = threadIdx.x & 63;
    const int64_t slot = ((int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6)) * G + lane / W;
    const int j = lane % W;
    const bool on = slot < count;
    // R16 (LHVI_PBP_FUSED_RECORDS16): records of sixteen words -- the eight above, then 8 particles (np)  9 var_ptr[v]  10-15 the
    // first six incident edges: the rows hang on ONE load behind the record instead of record -> np / var_ptr -> var_edge -> rows
    const int32_t* rec = list + (R16 ? 16 : 8) * (on ? slot : 0);
    int4 ra, rb;
    int2 rc = make_int2(0, 0);
    int e_rec = 0;                                             // lane j < 6: edge j of the row, straight from the record
    if (R16) {
        const int4* r4 = reinterpret_cast<const int4*>(rec);
        ra = r4[0]; rb = r4[1]; rc = reinterpret_cast<const int2*>(rec)[4];
        e_rec = rec[10 + min(lane % W, 5)];
    } else {
        ra = make_int4(rec[0], rec[1], rec[2], rec[3]); rb = make_int4(rec[4], rec[5], rec[6], rec[7]);
    }
    const int v = on ? ra.x : 0;
    const int deg = on ? ra.y : 0, gb = ra.z, T = on ? ra.w : 0;
    const double dlo = __hiloint2double(rb.y, rb.x), dhi = __hiloint2double(rb.w, rb.z);
    const int n = s.n, S = s.n + s.T;
    const int np = on ? (R16 ? rc.x : s.np[v]) : 0;
    const bool valid = j < np;
    const int lo = on ? (R16 ? rc.y : g.var_ptr[v]) : 0;
    const bool lifted = g.edge_count != nullptr;
    const double q0 = on ? s.q[2 * v] : 0.0, q1 = on ? s.q[2 * v + 1] : 1.0;
    // every lane of the wave runs the longest row of its variables (the reductions are wave-wide instructions)
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off >= W; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
    // A row is walked in chunks of FUSED_CH edges (one chunk for the rows of the benchmark and of the reference's models): lane j
    // of the group fetches edge j of the chunk, the ids are handed round with shuffles, and the row loads of ALL the chunk's edges
    // are issued before the first one is used.  Walked edge by edge -- id, then row, then the next id -- a variable costs 2 deg
    // dependent global round trips in step 1 and as many again in step 2, and the kernel's time is those latencies (2.6 ms at
    // n = 16 on the headline graph whatever its occupancy).  The arithmetic is the same expressions in the same order on the same
    // values as in the three kernels.
    constexpr int CH = FUSED_CH;
    static_assert(W >= CH && CH % SL == 0, "lane j of a group holds edge j of the chunk");
    const int first_lane = lane / W * W;
    const int nch = (maxdeg + CH - 1) / CH;                    // (wave-uniform, like maxdeg)
    int my_e = 0, held = -1;
    auto hold_chunk = [&](int c) {                             // lane j < CH: edge c CH + j of its variable's row
        if (c == held) return;
        held = c;
        if (R16 && c == 0) {                                   // the first six edges travel in the record
            my_e = j < deg ? (j < 6 ? e_rec : (j < CH ? g.var_edge[lo + j] : 0)) : 0;
            return;
        }
        my_e = (j < CH && c * CH + j < deg) ? g.var_edge[lo + c * CH + j] : 0;
    };
    // ---- step 1: v -> f (pbp_v2f_packed_kernel<W>)
    {
        double total = 0.0;
        int ee[CH];
        double mm[CH];
        auto load_rows = [&](int c) {
            hold_chunk(c);
#pragma unroll
            for (int i = 0; i < CH; ++i) ee[i] = __shfl(my_e, first_lane + i);
#pragma unroll
#ifdef LHVI_DIAG_FUSED_SLOT_ROWS          // timing aid (never defined in the product build): the rows of a variable read as if the table were in slot order
            for (int i = 0; i < CH; ++i) mm[i] = f2v[(int64_t)(lo + min(c * CH + i, max(deg - 1, 0))) * S + (valid ? j : 0)];
#else
            for (int i = 0; i < CH; ++i) mm[i] = f2v[(int64_t)ee[i] * S + (valid ? j : 0)];        // (beyond the row: edge 0's row, never used)
#endif
        };
        for (int c = 0; c < nch; ++c) {
            load_rows(c);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const double m = valid ? mm[i] : 0.0;
                if (c * CH + i < deg) total += lifted ? m * g.edge_count[ee[i]] : m;
            }
        }
        double logw = 0.0;
        bool uq = false;
        if (valid) {
            const int d = g.var_dom[v];
            const double sd = sqrt_pos(q1);
            logw = log_importance(g, s, v, d, s.particles[(int64_t)v * n + j], q0, rcp_newton(sd), log_pos(2.506628274631 * sd));
            uq = s.uniq[(int64_t)v * n + j] != 0;
        }
        auto group_sum = [&](double x) { return W == 16 ? dpp_move<0x15F>(dpp_row_reduce(x, SumOp())) : dpp_half_reduce(x, SumOp(), lane); };
        auto group_max = [&](double x) { return W == 16 ? dpp_row_reduce(x, MaxOp()) : dpp_half_reduce(x, MaxOp(), lane); };
        const uint64_t mine = (W == 16 ? 0xffffull : 0xffffffffull) << (lane / W * W);
        const double rcnt = rcp_newton(fmax((double)__builtin_popcountll(__ballot(uq) & mine), 1.0));
        for (int c = 0; c < nch; ++c) {
            if (nch > 1) load_rows(c);                          // (a one-chunk row still holds its messages)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int k = c * CH + i;
                if (k >= maxdeg) break;
                const bool live = k < deg;
                const double m = (live && valid) ? mm[i] : 0.0;
                const double res = (total - m) + logw;
                const double mean = group_sum(uq ? res : 0.0) * rcnt;
                double shift = mean;
                if (__ballot(uq && (res - mean > s.max_log_value))) {
                    const double mx = group_max(uq ? res : -__builtin_huge_val());
                    if (mx - mean > s.max_log_value) shift = mx - s.max_log_value;
                }
                if (live && valid) v2f[(int64_t)ee[i] * n + j] = res - shift;
            }
        }
    }
    // ---- step 2: the proposal (pbp_proposal_kernel<EP>)
    double total = 0.0;
    if (lifted) for (int k = 0; k < deg; ++k) total += g.edge_count[g.var_edge[lo + k]];
    else total = (double)deg;
    const double min_sig = total * s.var_threshold;
    const int sub = j / PW, tl = j % PW;
    double ps[4] = {0.0, 0.0, 0.0, 0.0}, pm[4] = {0.0, 0.0, 0.0, 0.0};      // sums of the edges with index = i (mod 64 / PW), slot `sub`'s share
    // an edge's T integral points over PW lanes: at most two per lane (T <= 2 PW by the lists' definition, include/lhvi.h)
    const int t0 = min(tl, max(T - 1, 0)), t1 = min(tl + PW, max(T - 1, 0));           // (clamped: the loads need no branch)
    const double xg0 = g.dom_val[gb + t0], xg1 = g.dom_val[gb + t1];
    constexpr int NB = LHVI_FUSED_NB;                          // passes (SL edges of a variable each) whose loads are in flight together
    for (int c = 0; c < nch; ++c) {
        hold_chunk(c);
#pragma unroll
        for (int p0 = 0; p0 < CH / SL; p0 += NB) {
            if (c * CH + p0 * SL >= maxdeg) break;
            int ee[NB];
            double b0[NB], b1[NB], g0[NB], g1[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int kc = (p0 + i) * SL + sub;               // position in the chunk
                ee[i] = __shfl(my_e, first_lane + min(kc, CH - 1));
                if (kc >= CH || c * CH + kc >= deg) ee[i] = 0;   // (beyond the row: edge 0's, loaded and dropped)
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
#ifdef LHVI_DIAG_FUSED_SLOT_ROWS
                const int ks = min(c * CH + (p0 + i) * SL + sub, max(deg - 1, 0));
                const double* msg = f2v + (int64_t)(lo + ks) * S + n;
                b0[i] = eta[2 * (lo + ks)]; b1[i] = eta[2 * (lo + ks) + 1];
#else
                const double* msg = f2v + (int64_t)ee[i] * S + n;
                b0[i] = eta[2 * ee[i]]; b1[i] = eta[2 * ee[i] + 1];
#endif
                g0[i] = msg[t0]; g1[i] = msg[t1];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (p0 + i >= CH / SL || c * CH + (p0 + i) * SL >= maxdeg) break;
                const int k = c * CH + (p0 + i) * SL + sub;
                const bool live = k < deg;
                const int e = ee[i];
                const double e0 = on ? b0[i] : 0.0, e1 = on ? b1[i] : 1.0;
                const bool use_cav = EP && !(q1 >= e1);
                double c0 = 0.0, c1 = 1.0;
                if (use_cav) gdiv(q0, q1, e0, e1, c0, c1);
                const double csd = sqrt(c1);
                double z = 0.0, a = 0.0, b = 0.0;
                if (on && tl < T) {
                    double w = exp(g0[i]);
                    if (use_cav) w = w * norm_pdf_std(xg0, c0, csd);
                    z += w; a += w * xg0; b += w * (xg0 * xg0);
                }
                if (on && tl + PW < T) {
                    double w = exp(g1[i]);
                    if (use_cav) w = w * norm_pdf_std(xg1, c0, csd);
                    z += w; a += w * xg1; b += w * (xg1 * xg1);
                }
                if (PW == 16) { z = dpp_row_reduce(z, SumOp()); a = dpp_row_reduce(a, SumOp()); b = dpp_row_reduce(b, SumOp()); }
                else { z = dpp_half_reduce(z, SumOp(), lane); a = dpp_half_reduce(a, SumOp(), lane); b = dpp_half_reduce(b, SumOp(), lane); }
                const double rz = rcp_newton(z);
                double mu = a * rz;
                double sig = b * rz - mu * mu;
                if (use_cav) { const double m0 = mu, m1 = sig; gdiv(m0, m1, c0, c1, mu, sig); }
                if (0.0 < sig && sig < __builtin_huge_val()) {
                    sig = fmax(sig, min_sig);
                    if (live && tl == 0) { eta[2 * e] = mu; eta[2 * e + 1] = sig; }
                } else {
                    mu = e0; sig = e1;
                }
                if (live) {
                    const double p = rcp_newton(sig);
                    const int slot4 = k & (WAVE / PW - 1);             // the lane group of pbp_proposal_kernel this edge would fall to
                    const double cnt = lifted ? g.edge_count[e] : 1.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r == slot4) {
                            if (lifted) { ps[r] += p * cnt; pm[r] += p * mu * cnt; }
                            else { ps[r] += p; pm[r] += p * mu; }
                        }
                }
            }
        }
    }
    // that kernel's fold of its lane groups: PW = 16: (g0 + g1) + (g2 + g3); PW = 32: g0 + g1
    double fs, fm;
    if (PW == 16 && SL == 1) { fs = (ps[0] + ps[1]) + (ps[2] + ps[3]); fm = (pm[0] + pm[1]) + (pm[2] + pm[3]); }
    else if (PW == 16) {
        // slot 0 holds g0 and g2, slot 1 (sixteen lanes on) g1 and g3
        const double s01 = (sub == 0 ? ps[0] : ps[1]), s23 = (sub == 0 ? ps[2] : ps[3]);
        const double m01 = (sub == 0 ? pm[0] : pm[1]), m23 = (sub == 0 ? pm[2] : pm[3]);
        fs = (s01 + __shfl_xor(s01, 16)) + (s23 + __shfl_xor(s23, 16));
        fm = (m01 + __shfl_xor(m01, 16)) + (m23 + __shfl_xor(m23, 16));
    } else { fs = ps[0] + ps[1]; fm = pm[0] + pm[1]; }
    fs = 1.0 / fs;
    const double mu_new = fs * fm, var_new = fs;
    if (on && j == 0) { q[2 * v] = mu_new; q[2 * v + 1] = var_new; }
    // ---- step 3: the new particles and their first-occurrence mask (pbp_resample_uniq_kernel)
    // (the values every lane of the group needs are the ones its first lane formed: that lane wrote q)
    const double mu_b = __shfl(mu_new, first_lane), sd_b = sqrt_pos(__shfl(var_new, first_lane));
    double x = 0.0;
    if (valid) {
        double zc, zs;
        philox_normal_pair(seed, gid ? (uint64_t)gid[v] : (uint64_t)v, (uint32_t)j, iteration, sh_log, zc, zs);
        x = fmin(fmax(fma(sd_b, zc, mu_b), dlo), dhi) + 0.0;      // + 0.0: no -0, so that equality below is equality of the patterns
        out[(int64_t)v * n + j] = x;
    }
    int u = valid;
    for (int k = 0; k + 1 < W; ++k) {
        const double xk = __shfl(x, first_lane + k);
        if (k < j && k < np && xk == x) u = 0;
    }
    if (on && j < n) uniq[(int64_t)v * n + j] = (uint8_t)u;
}

static int validate_pbp(const lhvi_graph_t* g, const lhvi_pbp_t* s) {
    if (!g || !s) return LHVI_E_ARG;
    if (g->V < 0 || g->E < 0 || s->n <= 0 || s->T < 0) return LHVI_E_ARG;
    if (g->V > 0 && (!g->var_ptr || !g->var_edge || !g->var_value || !g->var_dom || !g->dom_cont || !g->dom_ptr ||
                     !s->particles || !s->np)) return LHVI_E_ARG;
    if (s->var_hi > s->var_lo ? (s->var_lo < 0 || s->var_hi > g->V) : (s->var_lo != 0 || s->var_hi != 0)) return LHVI_E_ARG;
    return LHVI_OK;
}

}  // namespace lhvi

using namespace lhvi;

template <bool HALO>
static int launch_v2f(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, void* stream) {
    if (s->v2f_wide || s->v2f_narrow) {
        // the caller's split of the hidden variables: one wavefront per variable / sixteen variables per wavefront
        if (!s->v2f_wide || !s->v2f_narrow || s->n_v2f_wide < 0 || s->n_v2f_narrow < 0 || s->bslot || s->var_hi > s->var_lo) return LHVI_E_ARG;
        if (s->n_v2f_wide > 0)
            hipLaunchKernelGGL(pbp_v2f_kernel<HALO>, dim3(grid_for((int64_t)s->n_v2f_wide * WAVE)), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, v2f);
        if (s->v2f_hub && s->n_v2f_hub > 0)
            hipLaunchKernelGGL(pbp_v2f_hub_kernel<HALO>, dim3((unsigned)s->n_v2f_hub), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, v2f);
        if (s->n_v2f_narrow > 0)
            hipLaunchKernelGGL(pbp_v2f_narrow_kernel<HALO>, dim3(grid_for(((int64_t)s->n_v2f_narrow + 15) / 16 * WAVE)), dim3(BLOCK), 0,
                               as_stream(stream), *g, *s, f2v, v2f);
        if (s->n_v2f_mid16 < 0 || s->n_v2f_mid32 < 0 || (s->n_v2f_mid16 > 0 && !s->v2f_mid16) || (s->n_v2f_mid32 > 0 && !s->v2f_mid32)) return LHVI_E_ARG;
        if (s->n_v2f_mid16 > 0)
            hipLaunchKernelGGL((pbp_v2f_packed_kernel<16, HALO>), dim3(grid_for(((int64_t)s->n_v2f_mid16 + 3) / 4 * WAVE)), dim3(BLOCK), 0,
                               as_stream(stream), *g, *s, f2v, v2f, s->v2f_mid16, s->n_v2f_mid16);
        if (s->n_v2f_mid32 > 0)
            hipLaunchKernelGGL((pbp_v2f_packed_kernel<32, HALO>), dim3(grid_for(((int64_t)s->n_v2f_mid32 + 1) / 2 * WAVE)), dim3(BLOCK), 0,
                               as_stream(stream), *g, *s, f2v, v2f, s->v2f_mid32, s->n_v2f_mid32);
        return check_launch();
    }
    hipLaunchKernelGGL(pbp_v2f_kernel<HALO>, dim3(grid_for((int64_t)(var_limit(*g, *s) - var_first(*s)) * WAVE)), dim3(BLOCK), 0, as_stream(stream),
                       *g, *s, f2v, v2f);
    return check_launch();
}

template <int W>
static void launch_f2v_pair_small(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* v2f, double* f2v, int cus, int spare, void* stream) {
    static const int per_cu = blocks_per_cu((const void*)pbp_f2v_pair_small_kernel<W>);
    constexpr int PER_BLOCK = (BLOCK / WAVE) * (WAVE / W);                // entries per workgroup and step
    hipLaunchKernelGGL(pbp_f2v_pair_small_kernel<W>, dim3(min((s->n_pair + PER_BLOCK - 1) / PER_BLOCK, max(cus * per_cu - spare, 1))), dim3(BLOCK), 0,
                       as_stream(stream), *g, *s, v2f, f2v, reinterpret_cast<const PairDesc*>(s->pair_desc), s->n_pair);
}

template <int W, int PPL = 1>
static void launch_f2v_small(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* v2f, double* f2v, const void* desc, int nitems,
                             int cus, int share, int spare, void* stream) {
    static const int per_cu = blocks_per_cu((const void*)pbp_f2v_small_kernel<W, PPL>);
    constexpr int PER_BLOCK = (BLOCK / WAVE) * (WAVE / W);                // edges per workgroup and step
    hipLaunchKernelGGL((pbp_f2v_small_kernel<W, PPL>), dim3(min((nitems + PER_BLOCK - 1) / PER_BLOCK, max(cus * max(per_cu - share, 1) - spare, 1))),
                       dim3(BLOCK), 0, as_stream(stream), *g, *s, v2f, f2v, reinterpret_cast<const FastDesc*>(desc), nitems);
}

extern "C" {

int lhvi_pbp_uniq(const lhvi_graph_t* g, int32_t n, const double* particles, const int32_t* np, uint8_t* uniq, void* stream) {
    if (!g || n <= 0 || !particles || !np || !uniq) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    if (n <= WAVE)
        hipLaunchKernelGGL(pbp_uniq_wave_kernel, dim3(grid_for((int64_t)g->V * WAVE)), dim3(BLOCK), 0, as_stream(stream),
                           g->V, n, particles, np, uniq);
    else
        hipLaunchKernelGGL(pbp_uniq_kernel, dim3(grid_for((int64_t)g->V * n)), dim3(BLOCK), 0, as_stream(stream), g->V, n,
                           particles, np, uniq);
    return check_launch();
}

int lhvi_pbp_classify(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, uint8_t* edge_class, void* stream) {
    if (!g || !pots || !edge_class) return LHVI_E_ARG;
    if (s && (s->flags & LHVI_PBP_CQ) && (!s->np || s->n <= 0)) return LHVI_E_ARG;
    if (g->E == 0) return LHVI_OK;
    lhvi_pbp_t none = {};
    hipLaunchKernelGGL(pbp_classify_kernel, dim3(grid_for(g->E)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, s ? *s : none, edge_class);
    return check_launch();
}

int lhvi_pbp_describe(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const int32_t* edges, int32_t count,
                      void* desc_out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || count < 0) return LHVI_E_ARG;
    if (count == 0) return LHVI_OK;
    if (!edges || !desc_out) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_describe_kernel, dim3(grid_for(count)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s, edges, count,
                       reinterpret_cast<FastDesc*>(desc_out));
    return check_launch();
}

int lhvi_pbp_describe_cq(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const int32_t* edges, int32_t count,
                         void* desc_out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || count < 0) return LHVI_E_ARG;
    if (count == 0) return LHVI_OK;
    if (!edges || !desc_out) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_describe_cq_kernel, dim3(grid_for(count)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s, edges, count,
                       reinterpret_cast<CqDesc*>(desc_out));
    return check_launch();
}

int lhvi_debug_exp(const double* x, double* y, int64_t n, void* stream) {
    if (!x || !y || n < 0) return LHVI_E_ARG;
    if (n == 0) return LHVI_OK;
    hipLaunchKernelGGL(debug_exp_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, as_stream(stream), x, y, n);
    return check_launch();
}

int lhvi_debug_log(const double* x, double* y, int64_t n, int32_t which, void* stream) {
    if (!x || !y || n < 0 || which < 0 || which > 1) return LHVI_E_ARG;
    if (n == 0) return LHVI_OK;
    hipLaunchKernelGGL(debug_log_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, as_stream(stream), x, y, n, which);
    return check_launch();
}

int lhvi_debug_exp_acc_floor(const double* x, const double* c, double* y, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!x || !c || !y))) return LHVI_E_ARG;
    if (n == 0) return LHVI_OK;
    hipLaunchKernelGGL(debug_exp_acc_floor_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, as_stream(stream), x, c, y, n);
    return check_launch();
}

int lhvi_debug_exp_acc(const double* x, const double* c, double* y, int64_t n, void* stream) {
    if (!x || !c || !y || n < 0) return LHVI_E_ARG;
    if (n == 0) return LHVI_OK;
    hipLaunchKernelGGL(debug_exp_acc_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, as_stream(stream), x, c, y, n);
    return check_launch();
}

int lhvi_pbp_v2f(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !v2f || !s->uniq || !s->q) return LHVI_E_ARG;
    if (s->halo_off && !s->halo_buf) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    return s->halo_off ? launch_v2f<true>(g, s, f2v, v2f, stream) : launch_v2f<false>(g, s, f2v, v2f, stream);
}

int lhvi_pbp_f2v(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, double* f2v, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || !f2v || !s->old_particles) return LHVI_E_ARG;
    if ((s->flags & LHVI_PBP_CQ) && (!s->fast_edges || !s->generic_edges)) return LHVI_E_ARG;   // (conditional-quadratic routing needs the work lists)
    if (s->n_small16 < 0 || s->n_small32 < 0 || (s->n_small16 > 0 && !s->small16_desc) || (s->n_small32 > 0 && !s->small32_desc)) return LHVI_E_ARG;
    if (g->E == 0) return LHVI_OK;
    // persistent grids sized from the measured residency: CUs x resident workgroups per CU, every wave strides over its list
    const int nfast = s->fast_edges ? s->n_fast : g->E, ngen = s->generic_edges ? s->n_generic : g->E;
    static const int cus = device_cus();
    static const int fast_per_cu = blocks_per_cu((const void*)pbp_f2v_fast_kernel);
    static const int heavy_per_cu = blocks_per_cu((const void*)pbp_f2v_heavy_kernel, HEAVY_BLOCK);
    static const int light_per_cu = blocks_per_cu((const void*)pbp_f2v_light_kernel);
    static const int gen_per_cu = blocks_per_cu((const void*)pbp_f2v_generic_kernel<true>);
    static const int gen_slim_per_cu = blocks_per_cu((const void*)pbp_f2v_generic_kernel<false>);
    const int share = (s->flags & LHVI_PBP_SHARE_CUS) ? 1 : 0;          // a workgroup per CU left to the kernels of another stream
    const int heavy_blocks = max(heavy_per_cu - share, 1), side_blocks = 8;
    // LEAVE_ROOM (sharded runs): every persistent grid stays cus/8 workgroups short of filling the device, so a collective's
    // copy kernels on another stream can become resident while these waves run (no kernel here ever waits on another
    // workgroup, so a full device could only delay such a kernel, never block it -- but a delayed collective is an exposed one)
    const int spare = (s->flags & LHVI_PBP_LEAVE_ROOM) ? max(cus / 8, 1) : 0;
    // work ticket of the heavy kernel: reset on this stream right before the launch.  Chunks of 8 pay when every wave gets
    // several of them; a short list (a small graph, the interior part of a shard) keeps one entry per wave and strides
    const bool run_heavy = !(s->flags & LHVI_PBP_SKIP_FAST) && s->heavy_desc && s->n_heavy > 0 && !(s->flags & LHVI_PBP_SKIP_HEAVY);
    constexpr int HWPB = HEAVY_BLOCK / WAVE;
    const int heavy_grid = min((s->n_heavy + HWPB - 1) / HWPB, max(cus * heavy_blocks - spare * (BLOCK / WAVE) / HWPB, 1));
    lhvi_pbp_t sh = *s;
    if (sh.f2v_ticket && (int64_t)s->n_heavy < (int64_t)heavy_grid * HWPB * 8 * 4) sh.f2v_ticket = nullptr;
    if (s->f2v_ticket && run_heavy && hipMemsetAsync(s->f2v_ticket, 0, LHVI_PBP_TICKET_WORDS * sizeof(uint32_t), as_stream(stream)) != hipSuccess)
        return LHVI_E_LAUNCH;
    if (!(s->flags & LHVI_PBP_SKIP_FAST)) {
        if (s->heavy_desc && s->n_heavy > 0 && !(s->flags & LHVI_PBP_SKIP_HEAVY))
            hipLaunchKernelGGL(pbp_f2v_heavy_kernel, dim3(heavy_grid), dim3(HEAVY_BLOCK), 0, as_stream(stream),
                               *g, sh, v2f, f2v, reinterpret_cast<const FastDesc*>(s->heavy_desc), s->n_heavy,
                               s->f2v_ticket ? s->f2v_ticket + LHVI_PBP_TICKET_COUNTERS : (uint32_t*)nullptr);
        if (!(s->flags & LHVI_PBP_SKIP_HEAVY)) {
            // lane groups as narrow as the particle count allows: 10 / 12 lanes, or 8 with two particles per lane, for the small16 list;
            // 10 / 12 / 16 lanes with two particles per lane for small32
            // (no variable holds more than s->n particles; LHVI_PBP_POW2_GROUPS keeps the 16- / 32-lane kernels)
            const bool narrow = !(s->flags & LHVI_PBP_POW2_GROUPS);
            if (s->small16_desc && s->n_small16 > 0) {
                // (measured, scripts/diag/narrow_groups.sh + profiles/r05_experiments.md item 19: ten lanes beat two particles per lane in
                // groups of five at n = 10 -- 1.42 against 1.46 ms, the five-lane build needs 107 registers and 34 KB of LDS -- and lose to
                // groups of six at n = 12: 1.64 against 1.54 ms)
                if (narrow && s->n <= 10) launch_f2v_small<10>(g, s, v2f, f2v, s->small16_desc, s->n_small16, cus, share, spare, stream);
                else if (narrow && s->n <= 12) launch_f2v_small<6, 2>(g, s, v2f, f2v, s->small16_desc, s->n_small16, cus, share, spare, stream);
                else if (narrow) launch_f2v_small<8, 2>(g, s, v2f, f2v, s->small16_desc, s->n_small16, cus, share, spare, stream);   // 13-16 particles: eight edges per wavefront
                else launch_f2v_small<16>(g, s, v2f, f2v, s->small16_desc, s->n_small16, cus, share, spare, stream);
            }
            if (s->small32_desc && s->n_small32 > 0) {
                // (two particles per lane: groups of 10 / 12 / 16 lanes for up to 20 / 24 / 32 particles)
                if (narrow && s->n <= 20) launch_f2v_small<10, 2>(g, s, v2f, f2v, s->small32_desc, s->n_small32, cus, share, spare, stream);
                else if (narrow && s->n <= 24) launch_f2v_small<12, 2>(g, s, v2f, f2v, s->small32_desc, s->n_small32, cus, share, spare, stream);
                else if (narrow) launch_f2v_small<16, 2>(g, s, v2f, f2v, s->small32_desc, s->n_small32, cus, share, spare, stream);
                else launch_f2v_small<32>(g, s, v2f, f2v, s->small32_desc, s->n_small32, cus, share, spare, stream);
            }
        }
        if (s->pair_desc && s->n_pair > 0 && !(s->flags & (LHVI_PBP_SKIP_LIGHT | LHVI_PBP_WIDE_PAIRS)) && s->n <= 32 && LHVI_PAIR_SMALL) {
            // (every variable has at most s->n particles: four / three / two entries per wavefront.  Groups of 10 / 12 lanes for n <= 10 / 12
            // were measured too: 0.52 / 0.51 ms against 0.47-0.51 in groups of 16 -- the shuffles of their summation tree cost what six
            // or five entries per wavefront save; groups of 20 lanes for n <= 20: 0.65 against 0.77 ms in groups of 32)
            const bool narrow_pairs = !(s->flags & LHVI_PBP_POW2_GROUPS);
            if (s->n <= 16) launch_f2v_pair_small<16>(g, s, v2f, f2v, cus, spare, stream);
            else if (narrow_pairs && s->n <= 20) launch_f2v_pair_small<20>(g, s, v2f, f2v, cus, spare, stream);
            else launch_f2v_pair_small<32>(g, s, v2f, f2v, cus, spare, stream);
        } else if (s->pair_desc && s->n_pair > 0 && !(s->flags & LHVI_PBP_SKIP_LIGHT)) {
            static const int pair_per_cu = blocks_per_cu((const void*)pbp_f2v_pair_kernel);
            hipLaunchKernelGGL(pbp_f2v_pair_kernel, dim3(min((s->n_pair + 3) / 4, max(cus * min(pair_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0,
                               as_stream(stream), *g, *s, v2f, f2v, reinterpret_cast<const PairDesc*>(s->pair_desc), s->n_pair);
        } else if (s->light_desc && s->n_light > 0 && !(s->flags & LHVI_PBP_SKIP_LIGHT))
            hipLaunchKernelGGL(pbp_f2v_light_kernel, dim3(min((s->n_light + 3) / 4, max(cus * min(light_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0, as_stream(stream),
                               *g, *s, v2f, f2v, reinterpret_cast<const FastDesc*>(s->light_desc), s->n_light);
        if (nfast > 0 && !(s->flags & LHVI_PBP_SKIP_LIGHT))
            hipLaunchKernelGGL(pbp_f2v_fast_kernel, dim3(min((nfast + 3) / 4, max(cus * min(fast_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0, as_stream(stream),
                               *g, *pots, *s, v2f, f2v, reinterpret_cast<const FastDesc*>(s->fast_desc), pots->param);
    }
    if (s->cq_desc && s->n_cq > 0 && !(s->flags & LHVI_PBP_SKIP_CQ)) {
        static const int cq_per_cu = blocks_per_cu((const void*)pbp_f2v_cq_kernel);
        hipLaunchKernelGGL(pbp_f2v_cq_kernel, dim3(min((s->n_cq + 3) / 4, max(cus * min(cq_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, v2f, f2v, reinterpret_cast<const CqDesc*>(s->cq_desc), s->n_cq);
    }
    if (!(s->flags & LHVI_PBP_SKIP_GENERIC) && ngen > 0) {
        int pts_log2 = s->generic_edges ? s->generic_pts_log2 : 6;
        if (pts_log2 < 0 || pts_log2 > 6) pts_log2 = 6;
        const int groups = (ngen + (64 >> pts_log2) - 1) / (64 >> pts_log2);
        if (pots->interpreted == 0)
            hipLaunchKernelGGL(pbp_f2v_generic_kernel<false>, dim3(min((groups + 3) / 4, max(cus * min(gen_slim_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0,
                               as_stream(stream), *g, *pots, *s, v2f, f2v, pts_log2);
        else
            hipLaunchKernelGGL(pbp_f2v_generic_kernel<true>, dim3(min((groups + 3) / 4, max(cus * min(gen_per_cu, side_blocks) - spare, 1))), dim3(BLOCK), 0,
                               as_stream(stream), *g, *pots, *s, v2f, f2v, pts_log2);
    }
    return check_launch();
}

int lhvi_pbp_proposal(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* eta, double* q, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !eta || !q) return LHVI_E_ARG;
    if (s->prop_desc && (s->n_prop_desc < 0 || s->var_hi > s->var_lo)) return LHVI_E_ARG;    // the list replaces the variable range
    if (s->n_prop_hub < 0 || (s->n_prop_hub > 0 && (!s->prop_desc || !s->prop_hub || !s->prop_partial))) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    if (s->prop_desc) {
        if (s->n_prop_desc == 0) return LHVI_OK;
        const dim3 grid(grid_for((int64_t)s->n_prop_desc * WAVE));
        if (s->flags & LHVI_PBP_EP)
            hipLaunchKernelGGL(pbp_proposal_kernel<true>, grid, dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, eta, q, (double*)nullptr);
        else
            hipLaunchKernelGGL(pbp_proposal_kernel<false>, grid, dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, eta, q, (double*)nullptr);
        if (s->n_prop_hub > 0)
            hipLaunchKernelGGL(pbp_proposal_hub_finish_kernel, dim3(grid_for(s->n_prop_hub)), dim3(BLOCK), 0, as_stream(stream), *s, q);
        return check_launch();
    }
    if (s->flags & LHVI_PBP_EP)
        hipLaunchKernelGGL(pbp_proposal_kernel<true>, dim3(grid_for((int64_t)(var_limit(*g, *s) - var_first(*s)) * WAVE)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, f2v, eta, q, (double*)nullptr);
    else
        hipLaunchKernelGGL(pbp_proposal_kernel<false>, dim3(grid_for((int64_t)(var_limit(*g, *s) - var_first(*s)) * WAVE)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, f2v, eta, q, (double*)nullptr);
    return check_launch();
}

int lhvi_pbp_proposal_partial(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* eta, double* ph, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !eta || !ph || !s->q || s->prop_desc) return LHVI_E_ARG;      // (sharded runs address variables by range)
    if (g->V == 0) return LHVI_OK;
    if (s->flags & LHVI_PBP_EP)
        hipLaunchKernelGGL(pbp_proposal_kernel<true>, dim3(grid_for((int64_t)(var_limit(*g, *s) - var_first(*s)) * WAVE)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, f2v, eta, (double*)nullptr, ph);
    else
        hipLaunchKernelGGL(pbp_proposal_kernel<false>, dim3(grid_for((int64_t)(var_limit(*g, *s) - var_first(*s)) * WAVE)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, f2v, eta, (double*)nullptr, ph);
    return check_launch();
}

int lhvi_pbp_proposal_finish(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* ph, double* q, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!ph || !q || (s->bslot && (!s->recv || !s->brow_ptr || !s->brow_off || !s->brow_peer))) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_proposal_finish_kernel, dim3(grid_for(var_limit(*g, *s) - var_first(*s))), dim3(BLOCK), 0, as_stream(stream),
                       *g, *s, ph, q);
    return check_launch();
}

int lhvi_pbp_boundary_pack(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, const double* ph, int32_t nb,
                           const int32_t* bvars, double* out /* send buffer */, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (nb < 0) return LHVI_E_ARG;
    if (nb == 0) return LHVI_OK;
    if (!f2v || !ph || !bvars || !out || !s->brow_ptr || !s->brow_off) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_boundary_pack_kernel, dim3(grid_for((int64_t)nb * WAVE)), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, ph,
                       nb, bvars, out);
    return check_launch();
}

int lhvi_pbp_boundary_reduce(int32_t n_items, const int32_t* width, const int32_t* src_ptr, const int64_t* src_off,
                             const int32_t* dst_ptr, const int64_t* dst_off, const double* in, double* out, void* stream) {
    if (n_items < 0) return LHVI_E_ARG;
    if (n_items == 0) return LHVI_OK;
    if (!width || !src_ptr || !src_off || !dst_ptr || !dst_off || !in || !out) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_boundary_reduce_kernel, dim3(grid_for((int64_t)n_items * WAVE)), dim3(BLOCK), 0, as_stream(stream), n_items,
                       width, src_ptr, src_off, dst_ptr, dst_off, in, out);
    return check_launch();
}

int lhvi_pbp_init(const lhvi_graph_t* g, const lhvi_pbp_t* s, double* eta, double* q, double* f2v, double* v2f, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!eta || !q || !f2v || !v2f) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    if (g->E > 0) {
        if (hipMemsetAsync(f2v, 0, sizeof(double) * (size_t)g->E * (s->n + s->T), st) != hipSuccess) return LHVI_E_LAUNCH;
        if (hipMemsetAsync(v2f, 0, sizeof(double) * (size_t)g->E * s->n, st) != hipSuccess) return LHVI_E_LAUNCH;
    }
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_init_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, st, *g, *s, eta, q);
    return check_launch();
}

int lhvi_pbp_resample(const lhvi_graph_t* g, const lhvi_pbp_t* s, const int64_t* var_gid, uint64_t seed, uint32_t iteration,
                      double* particles_out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!particles_out || !s->q) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_resample_kernel, dim3(grid_for((int64_t)g->V * s->n)), dim3(BLOCK), 0, as_stream(stream), *g, *s,
                       var_gid, seed, iteration, particles_out);
    return check_launch();
}

int lhvi_pbp_edge_points(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f,
                         int32_t nq, const int32_t* qedge, int32_t npts, const double* x, double* out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || nq < 0 || npts < 0) return LHVI_E_ARG;
    if (nq == 0 || npts == 0) return LHVI_OK;
    if (!qedge || !x || !out) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_edge_points_kernel, dim3(grid_for((int64_t)nq * npts)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s,
                       v2f, nq, qedge, npts, x, out);
    return check_launch();
}

int lhvi_pbp_resample_uniq(const lhvi_graph_t* g, const lhvi_pbp_t* s, const int64_t* var_gid, uint64_t seed, uint32_t iteration,
                           double* particles_out, uint8_t* uniq_out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!particles_out || !uniq_out || !s->q) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    if (s->n > WAVE) {            // general n: two passes over every variable
        if (s->var_hi > s->var_lo) return LHVI_E_ARG;
        if (int rc = lhvi_pbp_resample(g, s, var_gid, seed, iteration, particles_out, stream)) return rc;
        return lhvi_pbp_uniq(g, s->n, particles_out, s->np, uniq_out, stream);
    }
    if (s->resample_vars && (s->n_resample_vars < 0 || s->var_hi > s->var_lo)) return LHVI_E_ARG;      // the list replaces the range
    const int64_t nvars = s->resample_vars ? s->n_resample_vars : var_limit(*g, *s) - var_first(*s);
    if (nvars == 0) return LHVI_OK;
    if (s->resample_vars)
        hipLaunchKernelGGL(pbp_resample_uniq_kernel<true>, dim3(persistent_grid((nvars + 1) / 2, 8)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, var_gid, seed, iteration, particles_out, uniq_out);
    else
        hipLaunchKernelGGL(pbp_resample_uniq_kernel<false>, dim3(persistent_grid((nvars + 1) / 2, 8)), dim3(BLOCK), 0,
                           as_stream(stream), *g, *s, var_gid, seed, iteration, particles_out, uniq_out);
    return check_launch();
}

int lhvi_pbp_var_sum(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !out) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_var_sum_kernel, dim3(grid_for((int64_t)g->V * s->n)), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, out);
    return check_launch();
}

int lhvi_pbp_domain_grid(const lhvi_graph_t* g, const lhvi_pbp_t* s, double* x, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!x || !g->dom_lo || !g->dom_hi || !g->dom_val) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_domain_grid_kernel, dim3(grid_for((int64_t)g->V * s->n)), dim3(BLOCK), 0, as_stream(stream), *g, *s, x);
    return check_launch();
}

int lhvi_pbp_refine_grid(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* logb, double* x, double* best, double* best_val,
                         void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!logb || !x || !best || !best_val) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_refine_grid_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, as_stream(stream), *g, *s, logb, x, best, best_val);
    return check_launch();
}

int lhvi_pbp_belief_points(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f,
                           int32_t nq, const int32_t* qvar, int32_t npts, const double* x, double* out, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || nq < 0 || npts < 0) return LHVI_E_ARG;
    if (nq == 0 || npts == 0) return LHVI_OK;
    if (!qvar || !x || !out) return LHVI_E_ARG;
    hipLaunchKernelGGL(pbp_belief_kernel, dim3(grid_for((int64_t)nq * npts)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s,
                       v2f, nq, qvar, npts, x, out);
    return check_launch();
}

int lhvi_pbp_map_brent(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, int64_t nq,
                       const int32_t* row_var, const int64_t* qptr, const int32_t* qedge, const double* qmult, double xtol,
                       int32_t maxfun, double* xout, double* fout, int32_t* nfev, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || nq < 0 || maxfun < 1 || !(xtol > 0.0)) return LHVI_E_ARG;
    if (nq == 0) return LHVI_OK;
    if (!row_var || !xout || !g->dom_lo || !g->dom_hi || !g->dom_val) return LHVI_E_ARG;
    if (qptr && (!qedge || !qmult)) return LHVI_E_ARG;
    QueryRows rows{row_var, qptr, qedge, qmult};
    hipLaunchKernelGGL(pbp_map_brent_kernel, dim3(grid_for(nq)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s, v2f, nq, rows,
                       row_var, xtol / 3.0, maxfun, xout, fout, nfev);
    return check_launch();
}

int lhvi_pbp_var_fused(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, double* eta, double* q,
                       const int64_t* var_gid, uint64_t seed, uint32_t iteration, double* particles_out, uint8_t* uniq_out,
                       const int32_t* desc, int32_t n16, int32_t n32_t32, int32_t n32_t64, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !v2f || !eta || !q || !particles_out || !uniq_out || !s->uniq || !s->q || !g->dom_lo || !g->dom_hi || !g->dom_val) return LHVI_E_ARG;
    if (n16 < 0 || n32_t32 < 0 || n32_t64 < 0 || s->n > 32 || s->bslot || s->var_hi > s->var_lo || particles_out == s->particles) return LHVI_E_ARG;
    if ((int64_t)n16 + n32_t32 + n32_t64 == 0) return LHVI_OK;
    if (!desc) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    const bool ep = (s->flags & LHVI_PBP_EP) != 0;
#define LHVI_FUSED(W, PW, list, cnt)                                                                                                  \
    if ((cnt) > 0) {                                                                                                                  \
        const dim3 grid(grid_for(((int64_t)(cnt) + WAVE / (W) - 1) / (WAVE / (W)) * WAVE));                                          \
        if (ep && r16) hipLaunchKernelGGL((pbp_var_fused_kernel<W, PW, true, true>), grid, dim3(BLOCK), 0, st, *g, *s, f2v, v2f, eta, q, var_gid, \
                                   seed, iteration, particles_out, uniq_out, list, cnt);                                             \
        else if (ep) hipLaunchKernelGGL((pbp_var_fused_kernel<W, PW, true, false>), grid, dim3(BLOCK), 0, st, *g, *s, f2v, v2f, eta, q, var_gid, \
                                   seed, iteration, particles_out, uniq_out, list, cnt);                                             \
        else if (r16) hipLaunchKernelGGL((pbp_var_fused_kernel<W, PW, false, true>), grid, dim3(BLOCK), 0, st, *g, *s, f2v, v2f, eta, q, var_gid,   \
                                seed, iteration, particles_out, uniq_out, list, cnt);                                                \
        else hipLaunchKernelGGL((pbp_var_fused_kernel<W, PW, false, false>), grid, dim3(BLOCK), 0, st, *g, *s, f2v, v2f, eta, q, var_gid,   \
                                seed, iteration, particles_out, uniq_out, list, cnt);                                                \
    }
    const bool r16 = (s->flags & LHVI_PBP_FUSED_RECORDS16) != 0;
    const int64_t rw = r16 ? 16 : 8;
    LHVI_FUSED(16, 16, desc, n16)
    LHVI_FUSED(32, 16, desc + rw * (int64_t)n16, n32_t32)
    LHVI_FUSED(32, 32, desc + rw * ((int64_t)n16 + n32_t32), n32_t64)
#undef LHVI_FUSED
    return check_launch();
}

int lhvi_pbp_quad(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, int64_t nq,
                  const int32_t* row_var, const int64_t* qptr, const int32_t* qedge, const double* qmult, const double* lo,
                  const double* hi, double epsabs, double epsrel, double* z, double* abserr, int32_t* status, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || nq < 0 || !(epsabs >= 0.0) || !(epsrel >= 0.0)) return LHVI_E_ARG;
    if (nq == 0) return LHVI_OK;
    if (!row_var || !lo || !hi || !z) return LHVI_E_ARG;
    if (qptr && (!qedge || !qmult)) return LHVI_E_ARG;
    QueryRows rows{row_var, qptr, qedge, qmult};
    hipLaunchKernelGGL(pbp_quad_kernel, dim3(grid_for(nq)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, *s, v2f, nq, rows, row_var,
                       lo, hi, epsabs, epsrel, z, abserr, status);
    return check_launch();
}

}  // extern "C"
