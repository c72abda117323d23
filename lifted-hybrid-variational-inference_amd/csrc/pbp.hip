// pbp.hip -- particle belief propagation sweep (EPBP / HybridLBP, log space) for gfx950.
//
// Reference semantics: EPBPLogVersion.py:30-215,225-289, HybridLBPLogVersion.py:44-236.
// Log messages are tabulated per edge in HBM:
//     f2v[e][0..n)  at the variable's particles      f2v[e][n..n+T)  at its integral points
//     v2f[e][0..n)  at the variable's particles
// Kernels (one launch each per sweep):
//   pbp_v2f_kernel       one wavefront per variable, lane = particle; rows of the incident f2v messages are read
//                        coalesced (8n bytes each), leave-one-out sums in rv.nb order, wave-shuffle mean/max for
//                        log_message_balance.  HBM bound (16n B per edge).
//   pbp_f2v_kernel       one wavefront per edge, lane = output point (new particles + integral points).  The
//                        partner's particles and its incoming log message are folded into per-particle
//                        coefficients (alpha, beta, kappa) staged in LDS, so the inner loop over joint particles is
//                        two FMAs + one fp64 exp.  fp64-VALU bound ((n+T)*n exps per edge).
//   pbp_proposal_kernel  one wavefront per variable: T-point moments per incident edge (shuffle reductions), site
//                        update rule, Gaussian product.
#include "common.hpp"
#include "potential.hpp"

namespace lhvi {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// EPBP.norm_pdf (EPBPLogVersion.py:49-53): sig is a standard deviation
__device__ __forceinline__ double norm_pdf_std(double x, double mu, double sig) {
    const double u = (x - mu) / sig;
    return exp(-u * u * 0.5) / (2.506628274631 * sig);
}

// log(important_weight(x, rv)) (EPBP:156-163, HLBP:173-180)
__device__ __forceinline__ double log_importance(const lhvi_graph_t& g, const lhvi_pbp_t& s, int v, int d, double x) {
    if (g.dom_cont[d]) {
        if (x == g.dom_lo[d] || x == g.dom_hi[d]) return log(1e-200);
    } else {
        if (!(s.flags & LHVI_PBP_EPBP_DISCRETE)) return 0.0;
        const int b = g.dom_ptr[d];
        const int ns = g.dom_ptr[d + 1] - b;
        if (x == g.dom_val[b] || (ns > 1 && x == g.dom_val[b + 1])) return log(1e-200);
    }
    const double mu = s.q[2 * v], sd = sqrt(s.q[2 * v + 1]);
    const double p = norm_pdf_std(x, mu, sd);
    return log(1.0 / fmax(p, 1e-200));
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

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) pbp_v2f_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                       double* __restrict__ v2f) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (v >= g.V) return;
    if (!is_hidden(g.var_value[v])) return;
    const int n = s.n, S = s.n + s.T;
    const int np = s.np[v];
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    const int d = g.var_dom[v];
    const bool lifted = g.edge_count != nullptr;
    const int nchunk = (np + 63) / 64;
    for (int k = lo; k < hi; ++k) {
        const int e = g.var_edge[k];
        const double own_c = lifted ? g.edge_count[e] - 1.0 : 0.0;
        double lsum = 0.0, lmax = -__builtin_huge_val();
        int lcnt = 0;
        double keep = 0.0;
        for (int c = 0; c < nchunk; ++c) {
            const int j = c * 64 + lane;
            double res = 0.0;
            if (j < np) {
                for (int kk = lo; kk < hi; ++kk) {
                    if (kk == k) continue;
                    const int e2 = g.var_edge[kk];
                    const double m = f2v[(int64_t)e2 * S + j];
                    res += lifted ? m * g.edge_count[e2] : m;
                }
                const double x = s.particles[(int64_t)v * n + j];
                res = res + log_importance(g, s, v, d, x);
                if (lifted) res = res + f2v[(int64_t)e * S + j] * own_c;
                if (s.uniq[(int64_t)v * n + j]) { lsum += res; lmax = fmax(lmax, res); ++lcnt; }
                if (nchunk > 1) v2f[(int64_t)e * n + j] = res;
            }
            keep = res;
        }
        // log_message_balance over the distinct keys (EPBP:204-215)
        const double tot = wave_sum(lsum);
        const double mx = wave_max(lmax);
        const int cnt = wave_sum_i(lcnt);
        const double mean = tot / (double)cnt;
        const double shift = (mx - mean > s.max_log_value) ? mx - s.max_log_value : mean;
        if (nchunk == 1) {
            if (lane < np) v2f[(int64_t)e * n + lane] = keep - shift;
        } else {
            for (int c = 0; c < nchunk; ++c) {
                const int j = c * 64 + lane;
                if (j < np) v2f[(int64_t)e * n + j] -= shift;
            }
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
        res += pot_times_exp(kind, par, xs, ix, m);
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

// exp for the f2v inner loop: same reduction and degree-11 polynomial as the ocml routine (|r| <= ln2/2,
// < 1 ulp), with the range checks replaced by one clamp -- the argument is log(phi) + log-message, and anything
// below -745 underflows to zero either way; overflow saturates to +inf through ldexp like exp() does.
__device__ __forceinline__ double exp_core(double t) {
    t = fmax(t, -745.2);
    const double k = rint(t * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, t);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.5052108385441718775e-8;
    p = fma(p, r, 2.7557319223985890653e-7);
    p = fma(p, r, 2.7557319223985890653e-6);
    p = fma(p, r, 2.4801587301587301587e-5);
    p = fma(p, r, 1.9841269841269841270e-4);
    p = fma(p, r, 1.3888888888888888889e-3);
    p = fma(p, r, 8.3333333333333333333e-3);
    p = fma(p, r, 4.1666666666666666667e-2);
    p = fma(p, r, 1.6666666666666666667e-1);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// Edge classes of the f -> v half sweep.  FAST edges have a term of the form
//     log phi + m_j = a_j + b_j * X1 + k_j * X2 + C        (j = partner particle, X1/X2/C = per output point)
//   (1) continuous target, log phi quadratic in it (Gaussian / Quadratic / LinearGaussian / XY / HybridQuadratic with
//       the discrete partner): X1 = x, X2 = x^2, C = 0, (a, b, k) = coefficients given the partner particle + message;
//   (2) discrete target of a HybridQuadratic(1 disc, 1 cont): X1 = b_d, X2 = A_d, C = c_d, (a, b, k) = (m_j, y_j, y_j^2).
// Everything else (tables, MLN formulas, arity != 2) takes the GENERIC kernel.
enum { EDGE_SKIP = 0, EDGE_FAST_CONT = 1, EDGE_FAST_DISC = 2, EDGE_GENERIC = 3 };

__device__ __forceinline__ int classify_edge(const lhvi_graph_t& g, const lhvi_pots_t& pots, int e) {
    if (canon(g.edge_canon, e) != e) return EDGE_SKIP;
    const int tv = g.edge_var[e];
    if (!is_hidden(g.var_value[tv])) return EDGE_SKIP;
    const int f = g.edge_fac[e], base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base, pos = e - base;
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

// FAST edges: one wavefront (= one 64-thread workgroup) per edge.  Partner coefficients are staged in LDS in
// tiles of 64; each lane owns one output point per round and accumulates sum_j exp(.) with two independent chains.
// A final partial round splits the partner range over idle lanes and folds the partial sums with shuffles, so
// n + T = 96 points on 64 lanes still keeps every lane busy.
__global__ void __launch_bounds__(WAVE) pbp_f2v_fast_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                           const double* __restrict__ v2f, double* __restrict__ f2v) {
    __shared__ double sh_a[WAVE], sh_b[WAVE], sh_k[WAVE];
    const int e = blockIdx.x;
    const int lane = threadIdx.x;
    const int cls = classify_edge(g, pots, e);
    if (cls != EDGE_FAST_CONT && cls != EDGE_FAST_DISC) return;
    const int tv = g.edge_var[e];
    const int n = s.n, S = s.n + s.T;
    const int d = g.var_dom[tv];
    const int np = s.np[tv];
    const int gb = g.dom_ptr[d];
    const int T = (cls == EDGE_FAST_CONT) ? g.dom_ptr[d + 1] - gb : 0;
    const int npts = np + T;
    double* out = f2v + (int64_t)e * S;
    const int f = g.edge_fac[e], base = g.fac_ptr[f], pos = e - base;
    const int pot = g.fac_pot[f], kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    const int pe = base + (1 - pos);
    const int pv = g.edge_var[pe];
    const int pce = canon(g.edge_canon, pe);
    const double pval = g.var_value[pv];
    const bool partner_hidden = is_hidden(pval);
    const int nj = partner_hidden ? s.np[pv] : 1;

    for (int p0 = 0; p0 < npts; p0 += 64) {
        const int rem = npts - p0;
        int width = 64;
        if (rem <= 32) { width = 1; while (width < rem) width <<= 1; }
        const int split = 64 / width, sub = lane / width, pl = lane % width;
        const int p = p0 + pl;
        const bool valid = pl < rem;
        double X1 = 0.0, X2 = 0.0, C = 0.0;
        if (valid) {
            if (cls == EDGE_FAST_CONT) {
                X1 = p < np ? s.particles[(int64_t)tv * n + p] : g.dom_val[gb + p - np];
                X2 = X1 * X1;
            } else {
                const int nst = (int)par[2];
                const int st = (int)s.particles[(int64_t)tv * n + p];     // HybridQuadratic indexes by the state value
                X2 = par[3 + st]; X1 = par[3 + nst + st]; C = par[3 + 2 * nst + st];
            }
        }
        double acc0 = 0.0, acc1 = 0.0;
        for (int j0 = 0; j0 < nj; j0 += 64) {
            const int jn = min(64, nj - j0);
            __syncthreads();
            if (lane < jn) {
                const int j = j0 + lane;
                double y = pval, m = 0.0;
                if (partner_hidden) { y = s.old_particles[(int64_t)pv * n + j]; m = v2f[(int64_t)pce * n + j]; }
                double a, b, k;
                if (cls == EDGE_FAST_CONT) {
                    Quad2 q;
                    quad2_of(kind, par, (kind == LHVI_POT_HYBRID_QUADRATIC) ? (int)y : 0, q);
                    if (pos == 0) { a = (q.a11 * y + q.b1) * y + q.c + m; b = q.axy * y + q.b0; k = q.a00; }
                    else          { a = (q.a00 * y + q.b0) * y + q.c + m; b = q.axy * y + q.b1; k = q.a11; }
                } else { a = m; b = y; k = y * y; }
                sh_a[lane] = a; sh_b[lane] = b; sh_k[lane] = k;
            }
            __syncthreads();
            const int chunk = (jn + split - 1) / split;
            const int jb = sub * chunk, je = min(jn, jb + chunk);
            int j = jb;
            for (; j + 1 < je; j += 2) {
                const double t0 = fma(sh_k[j], X2, fma(sh_b[j], X1, sh_a[j])) + C;
                const double t1 = fma(sh_k[j + 1], X2, fma(sh_b[j + 1], X1, sh_a[j + 1])) + C;
                acc0 += exp_core(t0);
                acc1 += exp_core(t1);
            }
            if (j < je) acc0 += exp_core(fma(sh_k[j], X2, fma(sh_b[j], X1, sh_a[j])) + C);
        }
        double acc = acc0 + acc1;
        for (int off = width; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        if (valid && sub == 0) out[p < np ? p : n + (p - np)] = acc > 0.0 ? log(acc) : -700.0;
    }
}

// GENERIC edges: one wavefront per edge, lane = output point, sequential joint loop per lane.
__global__ void __launch_bounds__(WAVE) pbp_f2v_generic_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_pbp_t s,
                                                              const double* __restrict__ v2f, double* __restrict__ f2v) {
    const int e = blockIdx.x;
    const int lane = threadIdx.x;
    if (classify_edge(g, pots, e) != EDGE_GENERIC) return;
    const int tv = g.edge_var[e];
    const int n = s.n, S = s.n + s.T;
    const int d = g.var_dom[tv];
    const int np = s.np[tv];
    const int gb = g.dom_ptr[d];
    const int T = g.dom_cont[d] ? g.dom_ptr[d + 1] - gb : 0;
    const int npts = np + T;
    double* out = f2v + (int64_t)e * S;
    for (int p = lane; p < npts; p += 64) {
        const double x = p < np ? s.particles[(int64_t)tv * n + p] : g.dom_val[gb + p - np];
        const int xi = p < np ? p : p - np;
        out[p < np ? p : n + (p - np)] = f2v_point_generic(g, pots, s, v2f, s.old_particles, e, x, xi);
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

// ---------------------------------------------------------------------------------------------
// gaussian_division (EPBP:43-47)
__device__ __forceinline__ void gdiv(double a0, double a1, double b0, double b1, double& mu, double& sig) {
    sig = a1 * b1 / (b1 - a1);
    mu = (a0 * (b1 + sig) - b0 * sig) / b1;
}

// update_proposal (EPBP:83-154; HLBP:100-171): one wavefront per continuous hidden variable
__global__ void __launch_bounds__(BLOCK) pbp_proposal_kernel(lhvi_graph_t g, lhvi_pbp_t s, const double* __restrict__ f2v,
                                                            double* __restrict__ eta, double* __restrict__ q) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (v >= g.V) return;
    const int d = g.var_dom[v];
    if (!is_hidden(g.var_value[v]) || !g.dom_cont[d]) return;
    const int n = s.n, S = s.n + s.T;
    const int gb = g.dom_ptr[d], T = g.dom_ptr[d + 1] - gb;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    double total = 0.0;
    for (int k = lo; k < hi; ++k) total += g.edge_count ? g.edge_count[g.var_edge[k]] : 1.0;
    const double min_sig = total * s.var_threshold;
    const double q0 = q[2 * v], q1 = q[2 * v + 1];
    double pm = 0.0, ps = 0.0;
    for (int k = lo; k < hi; ++k) {
        const int e = g.var_edge[k];
        const double* msg = f2v + (int64_t)e * S + n;
        const double b0 = eta[2 * e], b1 = eta[2 * e + 1];
        const bool use_cav = (s.flags & LHVI_PBP_EP) && !(q1 >= b1);
        double c0 = 0.0, c1 = 1.0;
        if (use_cav) gdiv(q0, q1, b0, b1, c0, c1);
        const double csd = sqrt(c1);
        double z = 0.0, a = 0.0, b = 0.0;
        for (int t = lane; t < T; t += 64) {
            const double xg = g.dom_val[gb + t];
            double w = exp(msg[t]);
            if (use_cav) w = w * norm_pdf_std(xg, c0, csd);
            z += w; a += w * xg; b += w * (xg * xg);
        }
        z = wave_sum(z); a = wave_sum(a); b = wave_sum(b);
        double mu = a / z;
        double sig = b / z - mu * mu;
        if (use_cav) { const double m0 = mu, m1 = sig; gdiv(m0, m1, c0, c1, mu, sig); }
        if (0.0 < sig && sig < __builtin_huge_val()) {
            sig = fmax(sig, min_sig);
            if (lane == 0) { eta[2 * e] = mu; eta[2 * e + 1] = sig; }
        } else {
            mu = b0; sig = b1;
        }
        const double p = 1.0 / sig;
        if (g.edge_count) { const double c = g.edge_count[e]; ps += p * c; pm += p * mu * c; }
        else { ps += p; pm += p * mu; }
    }
    ps = 1.0 / ps;
    if (lane == 0) { q[2 * v] = ps * pm; q[2 * v + 1] = ps; }
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
    for (int k = g.var_ptr[v]; k < g.var_ptr[v + 1]; ++k) total += g.edge_count ? g.edge_count[g.var_edge[k]] : 1.0;
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

__device__ __forceinline__ double philox_normal(uint64_t seed, uint64_t gid, uint32_t j, uint32_t iteration) {
    uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), j, iteration};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t r0 = ((uint64_t)c[0] << 32) | c[1], r1 = ((uint64_t)c[2] << 32) | c[3];
    const double u1 = ((double)(r0 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(r1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
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
    const double z = philox_normal(seed, gid ? (uint64_t)gid[v] : (uint64_t)v, (uint32_t)j, iteration);
    const double x = s.q[2 * v] + sqrt(s.q[2 * v + 1]) * z;
    out[i] = fmin(fmax(x, g.dom_lo[d]), g.dom_hi[d]);
}

static int validate_pbp(const lhvi_graph_t* g, const lhvi_pbp_t* s) {
    if (!g || !s) return LHVI_E_ARG;
    if (g->V < 0 || g->E < 0 || s->n <= 0 || s->T < 0) return LHVI_E_ARG;
    if (g->V > 0 && (!g->var_ptr || !g->var_edge || !g->var_value || !g->var_dom || !g->dom_cont || !g->dom_ptr ||
                     !s->particles || !s->np)) return LHVI_E_ARG;
    return LHVI_OK;
}

}  // namespace lhvi

using namespace lhvi;

extern "C" {

int lhvi_pbp_uniq(const lhvi_graph_t* g, int32_t n, const double* particles, const int32_t* np, uint8_t* uniq, void* stream) {
    if (!g || n <= 0 || !particles || !np || !uniq) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_uniq_kernel, dim3(grid_for((int64_t)g->V * n)), dim3(BLOCK), 0, as_stream(stream), g->V, n,
                       particles, np, uniq);
    return check_launch();
}

int lhvi_pbp_v2f(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !v2f || !s->uniq || !s->q) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_v2f_kernel, dim3(grid_for((int64_t)g->V * WAVE)), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, v2f);
    return check_launch();
}

int lhvi_pbp_f2v(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, double* f2v, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!pots || !v2f || !f2v || !s->old_particles) return LHVI_E_ARG;
    if (g->E == 0) return LHVI_OK;
    if (!(s->flags & LHVI_PBP_SKIP_FAST))
        hipLaunchKernelGGL(pbp_f2v_fast_kernel, dim3(g->E), dim3(WAVE), 0, as_stream(stream), *g, *pots, *s, v2f, f2v);
    if (!(s->flags & LHVI_PBP_SKIP_GENERIC))
        hipLaunchKernelGGL(pbp_f2v_generic_kernel, dim3(g->E), dim3(WAVE), 0, as_stream(stream), *g, *pots, *s, v2f, f2v);
    return check_launch();
}

int lhvi_pbp_proposal(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* eta, double* q, void* stream) {
    if (int rc = validate_pbp(g, s)) return rc;
    if (!f2v || !eta || !q) return LHVI_E_ARG;
    if (g->V == 0) return LHVI_OK;
    hipLaunchKernelGGL(pbp_proposal_kernel, dim3(grid_for((int64_t)g->V * WAVE)), dim3(BLOCK), 0, as_stream(stream), *g, *s, f2v, eta, q);
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

}  // extern "C"
