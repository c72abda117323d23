// vi.hip -- mixture-of-products variational step (VarInference / LiftedVarInference) for gfx950.
//
// Reference semantics: VarInference.py:26-195,249-287, LiftedVarInference.py:28-199.
// One launch of lhvi_vi_grad =
//   vi_var_kernel      thread per (variable, k): (N-1) E_k[log b_v] terms and their mu / var / category gradients
//   vi_factor_kernel   thread per (factor, k): tensor-product Gauss-Hermite / categorical expectation of
//                      F = log(phi + 1e-100) - log(b_f + 1e-100), per-edge gradient partials
//   vi_gather_kernel   thread per (variable, k): sums its edges' partials in rv.nb order (no atomics -> deterministic),
//                      softmax-Jacobian projection of the category gradient
//   vi_factor_cc_kernel the same for pairwise factors over continuous / observed variables (the bulk of a Gaussian model)
//   vi_weights_*       g_w and the free energy: two-stage reduction of the per-variable / per-factor expectations
// fp64-VALU bound and tiny after lifting.
#include "common.hpp"
#include "potential.hpp"
#include "fastmath.hpp"

namespace lhvi {

constexpr int VI_MAX_D = 32;   // max states of a discrete variable handled by the pinned expectation

__device__ __forceinline__ bool v_cont(const lhvi_graph_t& g, int v) { return g.dom_cont[g.var_dom[v]] != 0; }
__device__ __forceinline__ int v_nstates(const lhvi_graph_t& g, int v) { const int d = g.var_dom[v]; return g.dom_ptr[d + 1] - g.dom_ptr[d]; }
__device__ __forceinline__ const double* v_states(const lhvi_graph_t& g, int v) { return g.dom_val + g.dom_ptr[g.var_dom[v]]; }

__device__ __forceinline__ int vi_state_index(const lhvi_graph_t& g, int v, double x) {
    if (v_cont(g, v)) return 0;
    const double* s = v_states(g, v);
    const int n = v_nstates(g, v);
    for (int i = 0; i < n; ++i) if (s[i] == x) return i;
    return (int)x;
}

// C2FVarInference (C2FVarInference.py:120-136,253-261): an evidence cluster whose members' values differ is a Gaussian
// observation N(value, variance) -- obs_var[v] > 0 marks it.  It is integrated with T quadrature nodes like a hidden
// continuous variable, multiplies every mixture component of a belief by its pdf, and owns no parameters.
__device__ __forceinline__ bool is_gobs(const lhvi_vi_t& p, int v) { return p.obs_var != nullptr && p.obs_var[v] > 0.0; }

// VarInference.norm_pdf (VI:26-30): the normaliser is 2.5066 * var (sic)
__device__ __forceinline__ double norm_pdf_var(double x, double mu, double var) {
    const double u = x - mu;
    return exp(-u * u * 0.5 / var) / (2.506628274631 * var);
}

// rvs_belief (VI:336-353) over `m` variables
__device__ double rvs_belief(const lhvi_graph_t& g, const lhvi_vi_t& p, const double* x, const int* idx, const int* vars, int m) {
    for (int i = 0; i < m; ++i) {
        const double val = g.var_value[vars[i]];
        if (!is_hidden(val) && !is_gobs(p, vars[i]) && x[i] != val) return 0.0;
    }
    double s = 0.0;
    for (int k = 0; k < p.K; ++k) {
        double b = p.w[k];
        for (int i = 0; i < m; ++i) {
            const int v = vars[i];
            if (!is_hidden(g.var_value[v])) {
                if (is_gobs(p, v)) b *= norm_pdf_var(x[i], g.var_value[v], p.obs_var[v]);
                continue;
            }
            if (v_cont(g, v)) { const double* e = p.eta_c + ((int64_t)v * p.K + k) * 2; b *= norm_pdf_var(x[i], e[0], e[1]); }
            else b *= p.eta_d[((int64_t)v * p.K + k) * p.Dmax + idx[i]];
        }
        s += b;
    }
    return s;
}

// node t of variable v's axis under component k (the (is_continuous, eta) argument of expectation(), VI:40-55)
struct Node { double x, w; int idx; };
__device__ __forceinline__ int axis_len(const lhvi_graph_t& g, const lhvi_vi_t& p, int v) {
    if (!is_hidden(g.var_value[v])) return is_gobs(p, v) ? p.T : 1;
    return v_cont(g, v) ? p.T : v_nstates(g, v);
}
__device__ __forceinline__ Node axis_node(const lhvi_graph_t& g, const lhvi_vi_t& p, int v, int k, int t) {
    Node nd;
    const double val = g.var_value[v];
    if (!is_hidden(val) && is_gobs(p, v)) { nd.x = sqrt(2 * p.obs_var[v]) * p.gh_x[t] + val; nd.w = p.gh_w[t]; nd.idx = 0; }
    else if (!is_hidden(val)) { nd.x = val; nd.w = 1.0; nd.idx = vi_state_index(g, v, val); }
    else if (v_cont(g, v)) {
        const double* e = p.eta_c + ((int64_t)v * p.K + k) * 2;
        nd.x = sqrt(2 * e[1]) * p.gh_x[t] + e[0]; nd.w = p.gh_w[t]; nd.idx = 0;
    } else { nd.x = v_states(g, v)[t]; nd.w = p.eta_d[((int64_t)v * p.K + k) * p.Dmax + t]; nd.idx = t; }
    return nd;
}

__device__ double F_of(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_vi_t& p, int f, const double* x, const int* idx,
                       const int* vars, int arity) {
    const int pot = g.fac_pot[f];
    const double phi = pot_value(pots.kind[pot], pots.param + pots.off[pot], x, idx);
    return log(phi + 1e-100) - log(rvs_belief(g, p, x, idx, vars, arity) + 1e-100);
}

// ---- variable terms ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) vi_var_kernel(lhvi_graph_t g, lhvi_vi_t p, double* __restrict__ rvterm,
                                                      double* __restrict__ g_c, double* __restrict__ g_d) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)g.V * p.K) return;
    const int v = (int)(i / p.K), k = (int)(i % p.K);
    double N = (double)(g.var_ptr[v + 1] - g.var_ptr[v]);          // ground graph: the degree (sum of ones is exact)
    if (p.var_N) N = p.var_N[v];                                   // (the caller's sum: a hub cluster's row has thousands of entries)
    else if (g.edge_count) {
        N = 0.0;
        for (int j = g.var_ptr[v]; j < g.var_ptr[v + 1]; ++j) N += g.edge_count[g.var_edge[j]];
    }
    const double Mv = g.var_mult ? g.var_mult[v] : 1.0;
    const bool hid = is_hidden(g.var_value[v]);
    const bool cont = v_cont(g, v);
    const int n = axis_len(g, p, v);
    const double* e = p.eta_c + ((int64_t)v * p.K + k) * 2;
    double E = 0.0, Em = 0.0, Ev = 0.0;
    for (int t = 0; t < n; ++t) {
        const Node nd = axis_node(g, p, v, k, t);
        const double R = (N - 1) * log(rvs_belief(g, p, &nd.x, &nd.idx, &v, 1) + 1e-100);
        E += nd.w * R;
        if (hid && cont) {
            Em += nd.w * (R * (nd.x - e[0]));
            Ev += nd.w * (R * ((nd.x - e[0]) * (nd.x - e[0]) - e[1]));
        }
        if (hid && !cont) g_d[((int64_t)v * p.K + k) * p.Dmax + t] = -R;     // no weight: one entry per state (VI:139-141)
    }
    rvterm[i] = Mv * E;
    g_c[i * 2] = (hid && cont) ? -Em / e[1] : 0.0;
    g_c[i * 2 + 1] = (hid && cont) ? -Ev / (2 * e[1] * e[1]) : 0.0;
    if (!(hid && !cont)) for (int d = 0; d < p.Dmax; ++d) g_d[((int64_t)v * p.K + k) * p.Dmax + d] = 0.0;
    else for (int d = n; d < p.Dmax; ++d) g_d[((int64_t)v * p.K + k) * p.Dmax + d] = 0.0;
}

// expectation over the other slots with slot `pos` pinned to state d (gradient_category_tau, VI:133-160)
template <int MAXA>
__device__ double pinned_expectation(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_vi_t& p, int f, int base,
                                     int arity, const int* vars, int pos, int d, int k) {
    const int tv = vars[pos];
    const int Dt = v_nstates(g, tv);
    const double* tvals = v_states(g, tv);
    double x[MAXA];
    int idx[MAXA], nx[MAXA], nw[MAXA];
    int64_t totx = 1, totw = 1;
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        x[a] = 0.0; idx[a] = 0; nx[a] = 1; nw[a] = 1;
        if (a < arity && a != pos) {
            const int v = vars[a];
            if (is_hidden(g.var_value[v])) {
                if (p.quirks) { nx[a] = Dt; nw[a] = v_cont(g, v) ? 2 : v_nstates(g, v); }   // VI:147-150
                else { nx[a] = axis_len(g, p, v); nw[a] = nx[a]; }
            } else if (is_gobs(p, v)) { nx[a] = p.T; nw[a] = p.T; }                       // C2FVI:221-223: a proper axis
            totx *= nx[a]; totw *= nw[a];
        }
    }
    const int64_t cnt = totx < totw ? totx : totw;     // zip(product(xs), product(ws)) truncates to the shorter
    double E = 0.0;
    for (int64_t i = 0; i < cnt; ++i) {
        int64_t rx = i, rw = i;
        double w = 1.0;
#pragma unroll
        for (int a = MAXA - 1; a >= 0; --a) {
            if (a < arity && a != pos) {
                const int v = vars[a];
                const int ixs = (int)(rx % nx[a]); rx /= nx[a];
                const int iws = (int)(rw % nw[a]); rw /= nw[a];
                const double val = g.var_value[v];
                if (!is_hidden(val) && is_gobs(p, v)) { const Node nd = axis_node(g, p, v, k, ixs); x[a] = nd.x; idx[a] = 0; w *= p.gh_w[iws]; }
                else if (!is_hidden(val)) { x[a] = val; idx[a] = vi_state_index(g, v, val); }
                else if (p.quirks) {
                    x[a] = tvals[ixs]; idx[a] = vi_state_index(g, v, x[a]);
                    w *= v_cont(g, v) ? p.eta_c[((int64_t)v * p.K + k) * 2 + iws] : p.eta_d[((int64_t)v * p.K + k) * p.Dmax + iws];
                } else {
                    const Node nd = axis_node(g, p, v, k, ixs);
                    x[a] = nd.x; idx[a] = nd.idx; w *= nd.w;
                }
            }
        }
        x[pos] = tvals[d]; idx[pos] = d;
        E += w * F_of(g, pots, p, f, x, idx, vars, arity);
    }
    return E;
}

// ---- factor terms ------------------------------------------------------------------------------------------
// Pairwise factors whose two (distinct) variables are continuous or observed -- every factor of the Gaussian relational
// models -- take the kernel below: same arithmetic in the same order as the general kernel, minus the mixed-radix
// machinery for arity <= 6 and discrete axes (which costs the general kernel 256 VGPRs and a wave of occupancy).
__device__ __forceinline__ bool vi_is_cc(const lhvi_graph_t& g, const lhvi_pots_t& pots, int f) {
    const int base = g.fac_ptr[f];
    if (g.fac_ptr[f + 1] - base != 2) return false;
    const int kind = pots.kind[g.fac_pot[f]];
    if (kind != LHVI_POT_GAUSSIAN && kind != LHVI_POT_QUADRATIC && kind != LHVI_POT_LINEAR_GAUSSIAN && kind != LHVI_POT_XY) return false;
    const int v0 = g.edge_var[base], v1 = g.edge_var[base + 1];
    if (v0 == v1) return false;
    return (!is_hidden(g.var_value[v0]) || v_cont(g, v0)) && (!is_hidden(g.var_value[v1]) || v_cont(g, v1));
}

// log phi(x0, x1) for the four kinds above: the same expressions, in the same order, as pot_eval (potential.hpp)
__device__ __forceinline__ double pot_log_cc(int kind, const double* __restrict__ par, double x0, double x1) {
    if (kind == LHVI_POT_GAUSSIAN) {           // n = 2: mu = par[1..2], P = par[3..6] row-major
        const double d0 = x0 - par[1], d1 = x1 - par[2];
        double q = 0.0;
        q += (0.0 + d0 * par[3] + d1 * par[5]) * d0;
        q += (0.0 + d0 * par[4] + d1 * par[6]) * d1;
        return -0.5 * q;
    }
    if (kind == LHVI_POT_QUADRATIC) {          // A = par[1..4], b = par[5..6], c = par[7]
        double res = 0.0;
        res += x0 * (0.0 + par[1] * x0 + par[2] * x1);
        res += x1 * (0.0 + par[3] * x0 + par[4] * x1);
        res += par[5] * x0;
        res += par[6] * x1;
        return res + par[7];
    }
    if (kind == LHVI_POT_LINEAR_GAUSSIAN) { const double d = x1 - par[0] * x0; return -(d * d) * 0.5 / par[1]; }
    return -par[0] * x0 * x1 * 0.5 / par[1];   // XY
}

__device__ __forceinline__ double norm_pdf_var_fast(double x, double mu, double var, const double* __restrict__ tab) {
    const double u = x - mu;
    return exp_core(-u * u * 0.5 / var, tab) / (2.506628274631 * var);
}

__device__ __forceinline__ void vi_factor_cc(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_vi_t& p, double* __restrict__ ef,
                                             double* __restrict__ pe_c, double* __restrict__ pe_d, int64_t i,
                                             const double* __restrict__ sh_tab, const LogRec* __restrict__ sh_log) {
    const int f = (int)(i / p.K), k = (int)(i % p.K);
    if (!vi_is_cc(g, pots, f)) return;
    const int base = g.fac_ptr[f];
    const int v0 = g.edge_var[base], v1 = g.edge_var[base + 1];
    const double val0 = g.var_value[v0], val1 = g.var_value[v1];
    const bool h0 = is_hidden(val0), h1 = is_hidden(val1);
    const bool go0 = !h0 && is_gobs(p, v0), go1 = !h1 && is_gobs(p, v1);        // Gaussian observations: an axis, no parameters
    const bool a0 = h0 || go0, a1 = h1 || go1;
    const double* e0 = p.eta_c + ((int64_t)v0 * p.K + k) * 2;
    const double* e1 = p.eta_c + ((int64_t)v1 * p.K + k) * 2;
    const double mu0 = go0 ? val0 : e0[0], var0 = go0 ? p.obs_var[v0] : e0[1], mu1 = go1 ? val1 : e1[0], var1 = go1 ? p.obs_var[v1] : e1[1];
    const double s0 = a0 ? sqrt(2 * var0) : 0.0, s1 = a1 ? sqrt(2 * var1) : 0.0;
    const int n0 = a0 ? p.T : 1, n1 = a1 ? p.T : 1;
    const int pot = g.fac_pot[f];
    const int kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    double E = 0.0, Em0 = 0.0, Ev0 = 0.0, Em1 = 0.0, Ev1 = 0.0;
    for (int t0 = 0; t0 < n0; ++t0) {
        const double x0 = a0 ? s0 * p.gh_x[t0] + mu0 : val0, w0 = a0 ? p.gh_w[t0] : 1.0;
        for (int t1 = 0; t1 < n1; ++t1) {
            const double x1 = a1 ? s1 * p.gh_x[t1] + mu1 : val1, w1 = a1 ? p.gh_w[t1] : 1.0;
            const double w = 1.0 * w0 * w1;
            const double phi = exp_core(pot_log_cc(kind, par, x0, x1), sh_tab);
            double b = 0.0;                                   // rvs_belief: the mixture at (x0, x1), evidence agrees by construction
            for (int kk = 0; kk < p.K; ++kk) {
                double t = p.w[kk];
                if (h0) { const double* e = p.eta_c + ((int64_t)v0 * p.K + kk) * 2; t *= norm_pdf_var_fast(x0, e[0], e[1], sh_tab); }
                else if (go0) t *= norm_pdf_var_fast(x0, mu0, var0, sh_tab);
                if (h1) { const double* e = p.eta_c + ((int64_t)v1 * p.K + kk) * 2; t *= norm_pdf_var_fast(x1, e[0], e[1], sh_tab); }
                else if (go1) t *= norm_pdf_var_fast(x1, mu1, var1, sh_tab);
                b += t;
            }
            const double F = log_table(phi + 1e-100, sh_log) - log_table(b + 1e-100, sh_log);
            E += w * F;
            if (h0) { Em0 += w * (F * (x0 - mu0)); Ev0 += w * (F * ((x0 - mu0) * (x0 - mu0) - var0)); }
            if (h1) { Em1 += w * (F * (x1 - mu1)); Ev1 += w * (F * ((x1 - mu1) * (x1 - mu1) - var1)); }
        }
    }
    ef[i] = (g.fac_mult ? g.fac_mult[f] : 1.0) * E;
    const double c0 = g.edge_count ? g.edge_count[base] : 1.0, c1 = g.edge_count ? g.edge_count[base + 1] : 1.0;
    double* o0 = pe_c + ((int64_t)base * p.K + k) * 2;
    double* o1 = pe_c + ((int64_t)(base + 1) * p.K + k) * 2;
    o0[0] = h0 ? c0 * Em0 / var0 : 0.0; o0[1] = h0 ? c0 * Ev0 / (2 * var0 * var0) : 0.0;
    o1[0] = h1 ? c1 * Em1 / var1 : 0.0; o1[1] = h1 ? c1 * Ev1 / (2 * var1 * var1) : 0.0;
    // (no pe_d rows: the gather reads them for hidden DISCRETE variables only, and neither argument is one)
}

__global__ void __launch_bounds__(BLOCK) vi_factor_cc_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_vi_t p, double* __restrict__ ef,
                                                            double* __restrict__ pe_c, double* __restrict__ pe_d,
                                                            const int32_t* __restrict__ list, int n_list) {
    // grid-stride over (factor, k): the exp / log tables are copied into LDS once per block; the ~9 transcendental
    // evaluations per quadrature node then cost ~12 and ~20 instructions instead of ocml's ~25 and ~95
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int64_t n = (int64_t)(list ? n_list : g.F) * p.K;
    for (int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * BLOCK)
        vi_factor_cc(g, pots, p, ef, pe_c, pe_d, list ? (int64_t)list[j / p.K] * p.K + j % p.K : j, sh_tab, sh_log);
}


// ---- round 4: the factor terms from per-axis tables ------------------------------------------------------------------------
// On the tensor-product grid of expectation() (VI:40-55) the mixture belief factorises along the axes:
//   b(x[t_0], .., x[t_{a-1}]) = sum_k w_k * prod_i comp_i,k(t_i),   comp_i,k(t) = pdf of argument i under component k at ITS node t
// so the K * a * |axis| component values are computed ONCE per (factor, k) and every grid node costs K * a multiplications
// instead of K * a exponentials (same products in the same order as rvs_belief).

// log(phi + 1e-100) for phi = exp(v): v itself while exp(v) > 1e-84 (the 1e-100 is below half an ulp of phi then)
__device__ __forceinline__ double log_phi_eps(double v, const double* __restrict__ sh_tab, const LogRec* __restrict__ sh_log) {
    return v > -190.0 ? v : log_table(exp_core(v, sh_tab) + 1e-100, sh_log);
}

// VarInference.norm_pdf (VI:26-30) with the two divisions by the variance replaced by its reciprocal, taken once per (variable,
// component): exp(-(x - mu)^2 / 2 * inv) * (inv / 2.5066...)
__device__ __forceinline__ double norm_pdf_inv(double x, double mu, double inv, double scale, const double* __restrict__ tab) {
    const double u = x - mu;
    return exp_core(-(u * u) * 0.5 * inv, tab) * scale;
}

// Fast path for pairwise continuous / observed factors, T quadrature points known at compile time (T <= TT): the nodes of both
// axes, their 2 * T pdfs per mixture component and the T * T belief accumulators live in registers.
template <int TT>
__device__ __forceinline__ void vi_factor_cc_tab(const lhvi_graph_t& g, const lhvi_pots_t& pots, const lhvi_vi_t& p, double* __restrict__ ef,
                                                 double* __restrict__ pe_c, double* __restrict__ pe_d, int f, int k,
                                                 const double* __restrict__ sh_tab, const LogRec* __restrict__ sh_log) {
    const int64_t i = (int64_t)f * p.K + k;
    const int base = g.fac_ptr[f];
    const int v0 = g.edge_var[base], v1 = g.edge_var[base + 1];
    const double val0 = g.var_value[v0], val1 = g.var_value[v1];
    const bool h0 = is_hidden(val0), h1 = is_hidden(val1);
    const bool go0 = !h0 && is_gobs(p, v0), go1 = !h1 && is_gobs(p, v1);
    const bool a0 = h0 || go0, a1 = h1 || go1;
    const double* e0 = p.eta_c + ((int64_t)v0 * p.K + k) * 2;
    const double* e1 = p.eta_c + ((int64_t)v1 * p.K + k) * 2;
    const double mu0 = go0 ? val0 : e0[0], var0 = go0 ? p.obs_var[v0] : e0[1], mu1 = go1 ? val1 : e1[0], var1 = go1 ? p.obs_var[v1] : e1[1];
    const double s0 = a0 ? sqrt(2 * var0) : 0.0, s1 = a1 ? sqrt(2 * var1) : 0.0;
    const int n0 = a0 ? p.T : 1, n1 = a1 ? p.T : 1;
    double x0[TT], x1[TT], w0[TT], w1[TT], b[TT][TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double gx = t < p.T ? p.gh_x[t] : 0.0, gw = t < p.T ? p.gh_w[t] : 0.0;
        x0[t] = a0 ? s0 * gx + mu0 : val0; w0[t] = a0 ? gw : 1.0;
        x1[t] = a1 ? s1 * gx + mu1 : val1; w1[t] = a1 ? gw : 1.0;
#pragma unroll
        for (int u = 0; u < TT; ++u) b[t][u] = 0.0;
    }
    for (int kk = 0; kk < p.K; ++kk) {                                  // rvs_belief: b += ((w_kk * pdf0) * pdf1), VI:336-353
        const double* c0 = p.eta_c + ((int64_t)v0 * p.K + kk) * 2;
        const double* c1 = p.eta_c + ((int64_t)v1 * p.K + kk) * 2;
        const double m0 = go0 ? mu0 : c0[0], r0 = go0 ? var0 : c0[1], m1 = go1 ? mu1 : c1[0], r1 = go1 ? var1 : c1[1];
        const double i0 = a0 ? 1.0 / r0 : 0.0, i1 = a1 ? 1.0 / r1 : 0.0;
        const double sc0 = i0 * (1.0 / 2.506628274631), sc1 = i1 * (1.0 / 2.506628274631);
        double q0[TT], q1[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            q0[t] = (a0 && t < n0) ? norm_pdf_inv(x0[t], m0, i0, sc0, sh_tab) : 1.0;
            q1[t] = (a1 && t < n1) ? norm_pdf_inv(x1[t], m1, i1, sc1, sh_tab) : 1.0;
        }
        const double wk = p.w[kk];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double t0 = a0 ? wk * q0[t] : wk;
#pragma unroll
            for (int u = 0; u < TT; ++u) b[t][u] += a1 ? t0 * q1[u] : t0;
        }
    }
    const int pot = g.fac_pot[f];
    const int kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    double E = 0.0, Em0 = 0.0, Ev0 = 0.0, Em1 = 0.0, Ev1 = 0.0;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        if (t >= n0) continue;
#pragma unroll
        for (int u = 0; u < TT; ++u) {
            if (u >= n1) continue;
            const double w = 1.0 * w0[t] * w1[u];
            const double F = log_phi_eps(pot_log_cc(kind, par, x0[t], x1[u]), sh_tab, sh_log) - log_table(b[t][u] + 1e-100, sh_log);
            E += w * F;
            if (h0) { Em0 += w * (F * (x0[t] - mu0)); Ev0 += w * (F * ((x0[t] - mu0) * (x0[t] - mu0) - var0)); }
            if (h1) { Em1 += w * (F * (x1[u] - mu1)); Ev1 += w * (F * ((x1[u] - mu1) * (x1[u] - mu1) - var1)); }
        }
    }
    ef[i] = (g.fac_mult ? g.fac_mult[f] : 1.0) * E;
    const double c0 = g.edge_count ? g.edge_count[base] : 1.0, c1 = g.edge_count ? g.edge_count[base + 1] : 1.0;
    double* o0 = pe_c + ((int64_t)base * p.K + k) * 2;
    double* o1 = pe_c + ((int64_t)(base + 1) * p.K + k) * 2;
    o0[0] = h0 ? c0 * Em0 / var0 : 0.0; o0[1] = h0 ? c0 * Ev0 / (2 * var0 * var0) : 0.0;
    o1[0] = h1 ? c1 * Em1 / var1 : 0.0; o1[1] = h1 ? c1 * Ev1 / (2 * var1 * var1) : 0.0;
    // (no pe_d rows: the gather reads them for hidden DISCRETE variables only, and neither argument is one)
}

template <int TT>
__global__ void __launch_bounds__(BLOCK) vi_factor_cc_tab_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_vi_t p, double* __restrict__ ef,
                                                                double* __restrict__ pe_c, double* __restrict__ pe_d,
                                                                const int32_t* __restrict__ list, int n_list) {
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int64_t n = (int64_t)(list ? n_list : g.F) * p.K;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const int f = list ? list[i / p.K] : (int)(i / p.K), k = (int)(i % p.K);
        if (!list && !vi_is_cc(g, pots, f)) continue;
        vi_factor_cc_tab<TT>(g, pots, p, ef, pe_c, pe_d, f, k, sh_tab, sh_log);
    }
}

// General factors (discrete axes, MLN formulas, arity <= 6): a GROUP of L lanes per (factor, k) instead of a thread.  The lanes
// share the per-axis tables of this (factor, k) in LDS -- node positions, node weights, the K component values per node -- and
// split the grid nodes (and the evaluations of the pinned expectations of gradient_category_tau, VI:133-160) among themselves;
// partial sums are folded with xor shuffles inside the group.  A model with a few thousand factors of up to 3^5 nodes (robot
// mapping) then runs on the whole device instead of on a hundred long threads, and a large graph spends K * a multiplications per
// node on its belief instead of K * a exponentials.  Eligible: sum of the axis lengths <= VI_GRP_SLOTS, K * that <= VI_GRP_COMP
// (host: lhvi/vi.py builds the lists; legacy callers without lists keep the thread-per-factor kernels).
constexpr int VI_GRP_L = 8;
#ifndef LHVI_VI_GRP_BLOCK
#define LHVI_VI_GRP_BLOCK 256
#endif
#ifndef LHVI_VI_GRP_WAVES
#define LHVI_VI_GRP_WAVES 2
#endif
constexpr int VI_GRP_BLOCK = LHVI_VI_GRP_BLOCK;
constexpr int VI_GRP_SLOTS = LHVI_VI_GROUP_SLOTS;      // 24
constexpr int VI_GRP_COMP = LHVI_VI_GROUP_COMP;        // 48
constexpr int VI_GRP_PAR = 2 * VI_GRP_COMP;            // (1 / var, 1 / (2.5066 var)) per (argument, component): K * arity <= K * S <= 48

// q = r / d, r = r % d for 0 <= r < 2^14 and 1 <= d <= 64 (grid nodes of a factor, axis lengths): a float multiply by the
// reciprocal instead of the ~40-instruction integer division -- (r + 0.5) / d lies at least 0.5 / d = 2^-7 away from the next
// integer while the float product is off by less than 2^-9
__device__ __forceinline__ int small_divmod(int& r, int d, float rcp_d) {
    const int q = (int)(((float)r + 0.5f) * rcp_d);
    const int rem = r - q * d;
    r = q;
    return rem;
}

template <int L>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// log(phi(x) + 1e-100)
template <bool INTERP = true>
__device__ __forceinline__ double pot_log_eps(int kind, const double* __restrict__ par, const double* x, const int* idx,
                                              const double* __restrict__ sh_tab, const LogRec* __restrict__ sh_log) {
    bool is_log;
    const double v = pot_eval<INTERP>(kind, par, x, idx, is_log);
    return is_log ? log_phi_eps(v, sh_tab, sh_log) : log_table(v + 1e-100, sh_log);
}

// rvs_belief (VI:336-353) at an arbitrary point (the pinned expectations leave the grid: under the reference's quirk a continuous
// neighbour is evaluated at the TARGET's state values)
template <bool INTERP = true, class Stack>
__device__ __forceinline__ double pot_log_eps_on(int kind, const double* __restrict__ par, const double* x, const int* idx,
                                                 const double* __restrict__ sh_tab, const LogRec* __restrict__ sh_log, Stack& st) {
    bool is_log;
    const double v = pot_eval_on<INTERP>(kind, par, x, idx, is_log, st);
    return is_log ? log_phi_eps(v, sh_tab, sh_log) : log_table(v + 1e-100, sh_log);
}

template <int MAXA>
__device__ __forceinline__ double belief_direct(const lhvi_graph_t& g, const lhvi_vi_t& p, const double* x, const int* idx, const int* vars,
                                                int arity, const bool* hid, const bool* cont, const bool* axis, const double* __restrict__ sp,
                                                const double* __restrict__ sh_tab) {
    // (evidence agrees by construction: the pinned expectation puts an observed argument at its value)
    double s = 0.0;
    for (int kk = 0; kk < p.K; ++kk) {
        double b = p.w[kk];
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a >= arity || !axis[a]) continue;
            const int v = vars[a];
            if (!hid[a]) b *= norm_pdf_inv(x[a], g.var_value[v], sp[2 * (a * p.K + kk)], sp[2 * (a * p.K + kk) + 1], sh_tab);
            else if (cont[a]) b *= norm_pdf_inv(x[a], p.eta_c[((int64_t)v * p.K + kk) * 2], sp[2 * (a * p.K + kk)], sp[2 * (a * p.K + kk) + 1], sh_tab);
            else b *= p.eta_d[((int64_t)v * p.K + kk) * p.Dmax + idx[a]];
        }
        s += b;
    }
    return s;
}

// INTERP = false: the build for potential tables without an interpreted formula (lhvi_pots_t.interpreted == 0: every formula of the
// graph is evaluated through its conditional-quadratic block) -- no bytecode loop, no evaluation stack
template <int MAXA, int L, bool INTERP = true>
__global__ void __launch_bounds__(VI_GRP_BLOCK) __attribute__((amdgpu_waves_per_eu(LHVI_VI_GRP_WAVES, 8))) vi_factor_group_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_vi_t p, double* __restrict__ ef,
                                                                      double* __restrict__ pe_c, double* __restrict__ pe_d,
                                                                      const int32_t* __restrict__ list, int n_list) {
    constexpr int GROUPS = VI_GRP_BLOCK / L;
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    __shared__ double sh_x[GROUPS][VI_GRP_SLOTS];
    __shared__ double sh_w[GROUPS][VI_GRP_SLOTS];
    __shared__ double sh_c[GROUPS][VI_GRP_COMP];
    __shared__ double sh_p[GROUPS][VI_GRP_PAR];
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int grp = threadIdx.x / L, gl = threadIdx.x % L;
    double* sx = sh_x[grp]; double* sw = sh_w[grp]; double* sc = sh_c[grp]; double* sp = sh_p[grp];
    // persistent: the tables above are loaded once per workgroup; groups stride over the (factor, k) items (whole groups leave
    // together, so the shuffles below stay inside a group)
    for (int64_t item = (int64_t)blockIdx.x * GROUPS + grp; item < (int64_t)n_list * p.K; item += (int64_t)gridDim.x * GROUPS) {
    const int f = list[item / p.K], k = (int)(item % p.K);
    const int64_t i = (int64_t)f * p.K + k;
    const int base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base;
    int vars[MAXA], len[MAXA], off[MAXA], fix[MAXA];
    bool hid[MAXA], cont[MAXA], axis[MAXA];
    double mu[MAXA], var[MAXA];
    int S = 0, G = 1;
    // what the grid of this factor looks like depends on the graph and the evidence pattern only: the caller's per-edge records
    // (lhvi_vi_t.edge_axis: variable, axis length | flags, state index of an observed value) replace four dependent loads per slot
    const int4* __restrict__ ax = reinterpret_cast<const int4*>(p.edge_axis);
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        if (ax) {
            const int4 r = a < arity ? ax[base + a] : make_int4(0, 1, 0, 0);
            vars[a] = r.x; len[a] = r.y & 0xffff; fix[a] = r.z;
            hid[a] = a < arity && ((r.y >> 16) & 1); cont[a] = a < arity && ((r.y >> 17) & 1);
            axis[a] = a < arity && (hid[a] || ((r.y >> 18) & 1));
        } else {
            vars[a] = a < arity ? g.edge_var[base + a] : 0;
            const double val = g.var_value[vars[a]];
            hid[a] = a < arity && is_hidden(val);
            cont[a] = a < arity && v_cont(g, vars[a]);
            axis[a] = a < arity && (hid[a] || is_gobs(p, vars[a]));                 // takes part in the belief
            len[a] = a < arity ? axis_len(g, p, vars[a]) : 1;
            fix[a] = (a < arity && !is_hidden(val)) ? vi_state_index(g, vars[a], val) : 0;
        }
        off[a] = S;
        if (a < arity) { S += len[a]; G *= len[a]; }
        const double* e = p.eta_c + ((int64_t)vars[a] * p.K + k) * 2;
        mu[a] = (hid[a] && cont[a]) ? e[0] : 0.0;
        var[a] = (hid[a] && cont[a]) ? e[1] : 1.0;
    }
    // ---- per-axis tables of this (factor, k).  First 1 / var and 1 / (2.5066 var) of every (continuous argument, component):
    // the only divisions of the belief; then the nodes (under component k) and every component's value at them
    float rlen[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) rlen[a] = __builtin_amdgcn_rcpf((float)len[a]);
    for (int t = gl; t < arity * p.K; t += L) {
        const int a = t / p.K, kk = t - a * p.K;
        int v = 0; bool h = false, c = false, ax = false;
#pragma unroll
        for (int b = 0; b < MAXA; ++b) if (b == a) { v = vars[b]; h = hid[b]; c = cont[b]; ax = axis[b]; }
        double inv = 0.0;
        if (h && c) inv = 1.0 / p.eta_c[((int64_t)v * p.K + kk) * 2 + 1];
        else if (!h && ax) inv = 1.0 / p.obs_var[v];
        sp[2 * t] = inv; sp[2 * t + 1] = inv * (1.0 / 2.506628274631);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = gl; s < S; s += L) {
        int a = 0;
#pragma unroll
        for (int b = 1; b < MAXA; ++b) if (b < arity && s >= off[b]) a = b;
        int v = 0, o = 0; bool h = false, c = false, ax = false;
#pragma unroll
        for (int b = 0; b < MAXA; ++b) if (b == a) { v = vars[b]; o = off[b]; h = hid[b]; c = cont[b]; ax = axis[b]; }
        const int t = s - o;
        const Node nd = axis_node(g, p, v, k, t);
        sx[s] = nd.x; sw[s] = nd.w;
        for (int kk = 0; kk < p.K; ++kk) {
            double cv = 1.0;
            if (h && c) cv = norm_pdf_inv(nd.x, p.eta_c[((int64_t)v * p.K + kk) * 2], sp[2 * (a * p.K + kk)], sp[2 * (a * p.K + kk) + 1], sh_tab);
            else if (h) cv = p.eta_d[((int64_t)v * p.K + kk) * p.Dmax + t];
            else if (ax) cv = norm_pdf_inv(nd.x, g.var_value[v], sp[2 * (a * p.K + kk)], sp[2 * (a * p.K + kk) + 1], sh_tab);
            sc[kk * S + s] = cv;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int pot = g.fac_pot[f];
    const int kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    // ---- the expectation over the grid: E_k[F], E_k[F (x - mu)], E_k[F ((x - mu)^2 - var)]
    double x[MAXA];
    int idx[MAXA], it[MAXA];
    double E = 0.0, Em[MAXA], Ev[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) { Em[a] = 0.0; Ev[a] = 0.0; x[a] = 0.0; idx[a] = 0; it[a] = 0; }
    for (int node = gl; node < G; node += L) {
        int r = node;
        double w = 1.0;
#pragma unroll
        for (int a = MAXA - 1; a >= 0; --a) if (a < arity) it[a] = small_divmod(r, len[a], rlen[a]);
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a >= arity) continue;
            x[a] = sx[off[a] + it[a]];
            w *= sw[off[a] + it[a]];
            idx[a] = hid[a] ? (cont[a] ? 0 : it[a]) : fix[a];
        }
        double b = 0.0;
        for (int kk = 0; kk < p.K; ++kk) {
            double t = p.w[kk];
#pragma unroll
            for (int a = 0; a < MAXA; ++a) if (axis[a]) t *= sc[kk * S + off[a] + it[a]];
            b += t;
        }
        const double F = pot_log_eps<INTERP>(kind, par, x, idx, sh_tab, sh_log) - log_table(b + 1e-100, sh_log);
        E += w * F;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (hid[a] && cont[a]) {
                Em[a] += w * (F * (x[a] - mu[a]));
                Ev[a] += w * (F * ((x[a] - mu[a]) * (x[a] - mu[a]) - var[a]));
            }
        }
    }
    E = group_sum<L>(E);
    if (gl == 0) ef[i] = (g.fac_mult ? g.fac_mult[f] : 1.0) * E;
    // ---- per-edge partials; only the first position of a variable in the scope contributes (f.nb.index(rv), LVI:112,146)
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        if (a >= arity) continue;
        const int e = base + a, v = vars[a];
        bool first = true;
#pragma unroll
        for (int b = 0; b < MAXA; ++b) if (b < a && vars[b] == v) first = false;
        const double c = g.edge_count ? g.edge_count[e] : 1.0;
        double c0 = 0.0, c1 = 0.0;
        if (hid[a] && cont[a]) {                                           // (uniform over the group)
            const double sm = group_sum<L>(Em[a]), sv = group_sum<L>(Ev[a]);
            if (first) { c0 = c * sm / var[a]; c1 = c * sv / (2 * var[a] * var[a]); }
        }
        if (gl == 0) {
            pe_c[((int64_t)e * p.K + k) * 2] = c0;
            pe_c[((int64_t)e * p.K + k) * 2 + 1] = c1;
        }
        const bool pinned = hid[a] && first && !cont[a];
        const int Dt = pinned ? len[a] : 0;
        if (pinned) {
            // gradient_category_tau (VI:133-160): the other slots integrated with slot a pinned to each of its states in turn
            const double* tvals = v_states(g, v);
            int nx[MAXA], nw[MAXA];
            int64_t totx = 1, totw = 1;
#pragma unroll
            for (int b = 0; b < MAXA; ++b) {
                nx[b] = 1; nw[b] = 1;
                if (b < arity && b != a) {
                    if (hid[b]) {
                        if (p.quirks) { nx[b] = Dt; nw[b] = cont[b] ? 2 : len[b]; }          // VI:147-150
                        else { nx[b] = len[b]; nw[b] = len[b]; }
                    } else if (axis[b]) { nx[b] = p.T; nw[b] = p.T; }                        // Gaussian observation: a proper axis
                    totx *= nx[b]; totw *= nw[b];
                }
            }
            const int cnt = (int)(totx < totw ? totx : totw);     // zip(product(xs), product(ws)) truncates to the shorter
            for (int d = 0; d < Dt; ++d) {
                double acc = 0.0;
                for (int j = gl; j < cnt; j += L) {
                    int rx = j, rw = j;
                    double w = 1.0;
#pragma unroll
                    for (int b = MAXA - 1; b >= 0; --b) {
                        if (b < arity && b != a) {
                            int ixs, iws;
                            if (cnt <= (1 << 14)) {
                                ixs = small_divmod(rx, nx[b], __builtin_amdgcn_rcpf((float)nx[b]));
                                iws = small_divmod(rw, nw[b], __builtin_amdgcn_rcpf((float)nw[b]));
                            } else { ixs = rx % nx[b]; rx /= nx[b]; iws = rw % nw[b]; rw /= nw[b]; }
                            if (!hid[b] && axis[b]) { x[b] = sx[off[b] + ixs]; idx[b] = 0; w *= p.gh_w[iws]; }
                            else if (!hid[b]) { x[b] = g.var_value[vars[b]]; idx[b] = fix[b]; }
                            else if (p.quirks) {
                                x[b] = tvals[ixs]; idx[b] = vi_state_index(g, vars[b], x[b]);
                                w *= cont[b] ? p.eta_c[((int64_t)vars[b] * p.K + k) * 2 + iws] : p.eta_d[((int64_t)vars[b] * p.K + k) * p.Dmax + iws];
                            } else { x[b] = sx[off[b] + ixs]; idx[b] = cont[b] ? 0 : ixs; w *= sw[off[b] + ixs]; }
                        }
                    }
#pragma unroll
                    for (int b = 0; b < MAXA; ++b) if (b == a) { x[b] = tvals[d]; idx[b] = d; }
                    const double bel = belief_direct<MAXA>(g, p, x, idx, vars, arity, hid, cont, axis, sp, sh_tab);
                    acc += w * (pot_log_eps<INTERP>(kind, par, x, idx, sh_tab, sh_log) - log_table(bel + 1e-100, sh_log));
                }
                acc = group_sum<L>(acc);
                if (gl == 0) pe_d[((int64_t)e * p.K + k) * p.Dmax + d] = c * acc;
            }
        }
        // a repeated variable's later positions contribute zeros; rows of other kinds of variables and states beyond the
        // variable's own are never read (vi_gather_kernel)
        if (gl == 0 && hid[a] && !cont[a]) for (int d = Dt; d < len[a]; ++d) pe_d[((int64_t)e * p.K + k) * p.Dmax + d] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // the next item overwrites this group's tables
    __builtin_amdgcn_wave_barrier();
    }
}

// ---- tiny grids: a thread per (factor, k) -----------------------------------------------------------------------------------
// Most formulas of a grounded hybrid MLN have one or two hidden arguments left once the evidence is in: grids of 1 .. 18 nodes
// (paper-popularity: 21 % of the factors are fully observed, 30 % have two nodes).  Eight lanes and a set of LDS tables per such
// item spend their time in the item's chain of dependent loads; here a thread walks its item alone.  The grid nodes and the
// points of the pinned expectations (gradient_category_tau, VI:133-160) form ONE list of points with one evaluation site --
// decode the point, belief as the sum over components of the product of the per-argument pdfs (1 / var hoisted per argument and
// component), log phi - log belief, accumulate -- so the formula interpreter and the pdf code exist once in the kernel.
// Same products in the same order as the group kernel; the points are summed in index order (the group kernel: strided over its
// lanes, then a butterfly), so the two agree to rounding.
// Eligible (host: lhvi/vi.py): arity <= 3, at most LHVI_VI_TINY_NODES grid nodes, K <= VI_TINY_K, axis records present.
constexpr int VI_TINY_K = 2;
constexpr int VI_TINY_PAR = 3072;      // doubles of parameter rows kept in LDS (24 KB: two workgroups per CU still fit)
#ifndef LHVI_VI_TINY_WAVES
#define LHVI_VI_TINY_WAVES 2
#endif
#ifndef LHVI_VI_TINY_SLIM_WAVES
#define LHVI_VI_TINY_SLIM_WAVES 3          // the build without the interpreter (measured: 2 waves 0.53 ms, 3 waves 0.50, 4 waves -- 304 B of scratch -- 0.64 on the scaled cfg 3)
#endif
#ifndef LHVI_VI_TINY_HOIST
#define LHVI_VI_TINY_HOIST 1
#endif
constexpr int VI_TINY_PAR_SLIM = 1024;     // parameter rows it keeps in LDS (8 KB); larger tables take the general build

template <bool INTERP>
struct TinyStack { using type = MlnLdsStack<BLOCK>; };
template <>
struct TinyStack<false> { using type = MlnNoStack; };

// INTERP = false: the build for potential tables without an interpreted formula (lhvi_pots_t.interpreted == 0) -- no bytecode
// loop and no stack columns in LDS (24 KB less per workgroup)
template <bool INTERP>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(INTERP ? LHVI_VI_TINY_WAVES : LHVI_VI_TINY_SLIM_WAVES, 8)))
vi_factor_tiny_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_vi_t p, double* __restrict__ ef, double* __restrict__ pe_c,
                      double* __restrict__ pe_d, const int32_t* __restrict__ list, int n_list) {
    __shared__ double sh_tab[EXP_TAB_N];
    __shared__ LogRec sh_log[LOG_TAB_N];
    __shared__ double sh_stack[INTERP ? MLN_STACK * BLOCK : 1];      // the formula interpreter's stack: a column per thread (in registers it is
    typename TinyStack<INTERP>::type stack;                          // a dynamically indexed array: 0.81 ms instead of 0.70 on the scaled cfg 3)
    if constexpr (INTERP) stack.base = sh_stack + threadIdx.x;
    // the parameter rows of all potentials (for an MLN formula: its program) in LDS: the interpreter fetches an opcode per step, each
    // a dependent load -- 7 to 21 global round trips per evaluation otherwise
    __shared__ double sh_par[INTERP ? VI_TINY_PAR : VI_TINY_PAR_SLIM];
    for (int t = threadIdx.x; t < p.tiny_par_words; t += BLOCK) sh_par[t] = pots.param[t];     // (the launcher checks the size)
    load_log_table(sh_log);
    load_exp_table(sh_tab);
    const int4* __restrict__ recs = reinterpret_cast<const int4*>(p.edge_axis);
    for (int64_t item = (int64_t)blockIdx.x * BLOCK + threadIdx.x; item < (int64_t)n_list * p.K; item += (int64_t)gridDim.x * BLOCK) {
        const int f = list[item / p.K], k = (int)(item % p.K);
        const int base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base;
        int av[3], alen[3], afix[3], adom[3];
        bool hid[3], cont[3], axis[3], gauss[3];      // gauss: integrated with Gauss-Hermite nodes (hidden continuous / Gaussian observation)
        double mu[3], var[3], sd[3], val[3], inv[3][VI_TINY_K];
#if LHVI_VI_TINY_HOIST
        double mk[3][VI_TINY_K];                      // every component's mean of a Gauss-Hermite argument: read per point, component and
#endif                                                // argument inside the loop below, it is a dependent global load each time
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int4 r = a < arity ? recs[base + a] : make_int4(0, 1, 0, 0);
            av[a] = r.x; alen[a] = r.y & 0xffff; afix[a] = r.z; adom[a] = r.w;
            hid[a] = a < arity && ((r.y >> 16) & 1); cont[a] = a < arity && ((r.y >> 17) & 1);
            const bool gobs = a < arity && ((r.y >> 18) & 1);
            axis[a] = hid[a] || gobs;
            gauss[a] = (hid[a] && cont[a]) || gobs;
            val[a] = a < arity ? g.var_value[av[a]] : 0.0;
            mu[a] = 0.0; var[a] = 1.0; sd[a] = 0.0;
#pragma unroll
            for (int kk = 0; kk < VI_TINY_K; ++kk) inv[a][kk] = 0.0;
#if LHVI_VI_TINY_HOIST
#pragma unroll
            for (int kk = 0; kk < VI_TINY_K; ++kk) mk[a][kk] = val[a];
#endif
            if (hid[a] && cont[a]) {
                const double* e = p.eta_c + (int64_t)av[a] * p.K * 2;
                mu[a] = e[2 * k]; var[a] = e[2 * k + 1]; sd[a] = sqrt(2 * var[a]);
#pragma unroll
                for (int kk = 0; kk < VI_TINY_K; ++kk) if (kk < p.K) {
                    inv[a][kk] = 1.0 / e[2 * kk + 1];
#if LHVI_VI_TINY_HOIST
                    mk[a][kk] = e[2 * kk];
#endif
                }
            } else if (gobs) {
                const double ov = p.obs_var[av[a]];
                mu[a] = val[a]; sd[a] = sqrt(2 * ov);
#pragma unroll
                for (int kk = 0; kk < VI_TINY_K; ++kk) inv[a][kk] = 1.0 / ov;
            }
        }
        const int pot = g.fac_pot[f];
        const int kind = pots.kind[pot];
        const double* par = sh_par + pots.off[pot];
        // the pinned positions: hidden discrete arguments at their first position in the scope (f.nb.index(rv), LVI:112,146);
        // npin[a] = points of position a's pinned expectations = states * min(prod nx, prod nw) (zip truncates to the shorter)
        const int G = alen[0] * alen[1] * alen[2];
        int npin[3], cntp[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bool first = a < arity;
#pragma unroll
            for (int b = 0; b < 3; ++b) if (b < a && av[b] == av[a]) first = false;
            int totx = 1, totw = 1;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b >= arity || b == a) continue;
                if (hid[b]) { totx *= p.quirks ? alen[a] : alen[b]; totw *= p.quirks ? (cont[b] ? 2 : alen[b]) : alen[b]; }   // VI:147-150
                else if (axis[b]) { totx *= p.T; totw *= p.T; }                                    // Gaussian observation: a proper axis
            }
            cntp[a] = totx < totw ? totx : totw;
            npin[a] = (first && hid[a] && !cont[a]) ? alen[a] * cntp[a] : 0;
            if (a < arity && hid[a] && !cont[a] && !first)                // a repeated variable's later positions contribute zeros
                for (int d = 0; d < alen[a]; ++d) pe_d[((int64_t)(base + a) * p.K + k) * p.Dmax + d] = 0.0;
        }
        const int total = G + npin[0] + npin[1] + npin[2];
        const double wk0 = p.w[0], wk1 = p.K > 1 ? p.w[1] : 0.0;
        double x[3];
        int idx[3];
        double E = 0.0, Em[3] = {0.0, 0.0, 0.0}, Ev[3] = {0.0, 0.0, 0.0}, acc = 0.0;
        for (int pt = 0; pt < total; ++pt) {
            // ---- decode the point
            int pa = -1, d = 0, j = pt, cnt = 1, Dt = 1;
            if (pt >= G) {
                int q = pt - G;
                pa = 0;
                if (q >= npin[0]) { q -= npin[0]; pa = 1; if (q >= npin[1]) { q -= npin[1]; pa = 2; } }
                cnt = pa == 0 ? cntp[0] : pa == 1 ? cntp[1] : cntp[2];
                Dt = pa == 0 ? alen[0] : pa == 1 ? alen[1] : alen[2];
                d = q / cnt; j = q - d * cnt;
            }
            const double* tvals = g.dom_val + (pa == 1 ? adom[1] : pa == 2 ? adom[2] : adom[0]);
            int rx = j, rw = j;
            double wb[3];
#pragma unroll
            for (int b = 2; b >= 0; --b) {
                wb[b] = 1.0; x[b] = 0.0; idx[b] = 0;
                if (b >= arity) continue;
                if (b == pa) { x[b] = tvals[d]; idx[b] = d; continue; }
                int nx = alen[b], nw = alen[b];
                const bool quirk = pa >= 0 && hid[b] && p.quirks;
                if (quirk) { nx = Dt; nw = cont[b] ? 2 : alen[b]; }
                const int ixs = small_divmod(rx, nx, __builtin_amdgcn_rcpf((float)nx));
                const int iws = pa >= 0 ? small_divmod(rw, nw, __builtin_amdgcn_rcpf((float)nw)) : ixs;
                if (quirk) {
                    x[b] = tvals[ixs];
                    idx[b] = 0;                                   // vi_state_index from the axis record: first matching state, else (int) x
                    if (!cont[b]) {
                        idx[b] = (int)x[b];
                        for (int t = alen[b] - 1; t >= 0; --t) if (g.dom_val[adom[b] + t] == x[b]) idx[b] = t;
                    }
                    wb[b] = cont[b] ? p.eta_c[((int64_t)av[b] * p.K + k) * 2 + iws] : p.eta_d[((int64_t)av[b] * p.K + k) * p.Dmax + iws];
                } else if (gauss[b]) { x[b] = sd[b] * p.gh_x[ixs] + mu[b]; wb[b] = p.gh_w[iws]; }
                else if (hid[b]) { x[b] = g.dom_val[adom[b] + ixs]; idx[b] = ixs; wb[b] = p.eta_d[((int64_t)av[b] * p.K + k) * p.Dmax + iws]; }
                else { x[b] = val[b]; idx[b] = afix[b]; }
            }
            // (the grid multiplies its weights in ascending position order, the pinned expectation in descending order)
            const double w = pa < 0 ? ((1.0 * wb[0]) * wb[1]) * wb[2] : ((1.0 * wb[2]) * wb[1]) * wb[0];
            // ---- rvs_belief (VI:336-353) at the point
            double bel = 0.0;
#pragma unroll
            for (int kk = 0; kk < VI_TINY_K; ++kk) {
                if (kk >= p.K) continue;
                double t = kk == 0 ? wk0 : wk1;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (!axis[b]) continue;
                    if (gauss[b]) {
#if LHVI_VI_TINY_HOIST
                        const double m = mk[b][kk];
#else
                        const double m = hid[b] ? p.eta_c[((int64_t)av[b] * p.K + kk) * 2] : val[b];
#endif
                        t *= norm_pdf_inv(x[b], m, inv[b][kk], inv[b][kk] * (1.0 / 2.506628274631), sh_tab);
                    } else t *= p.eta_d[((int64_t)av[b] * p.K + kk) * p.Dmax + idx[b]];
                }
                bel += t;
            }
            const double F = pot_log_eps_on<INTERP>(kind, par, x, idx, sh_tab, sh_log, stack) - log_table(bel + 1e-100, sh_log);
            if (pa < 0) {
                E += w * F;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (hid[a] && cont[a]) {
                        Em[a] += w * (F * (x[a] - mu[a]));
                        Ev[a] += w * (F * ((x[a] - mu[a]) * (x[a] - mu[a]) - var[a]));
                    }
                }
            } else {
                acc += w * F;
                if (j == cnt - 1) {
                    const int e = base + pa;
                    pe_d[((int64_t)e * p.K + k) * p.Dmax + d] = (g.edge_count ? g.edge_count[e] : 1.0) * acc;
                    acc = 0.0;
                }
            }
        }
        ef[(int64_t)f * p.K + k] = (g.fac_mult ? g.fac_mult[f] : 1.0) * E;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a >= arity) continue;
            bool first = true;
#pragma unroll
            for (int b = 0; b < 3; ++b) if (b < a && av[b] == av[a]) first = false;
            const int e = base + a;
            const double c = g.edge_count ? g.edge_count[e] : 1.0;
            double c0 = 0.0, c1 = 0.0;
            if (hid[a] && cont[a] && first) { c0 = c * Em[a] / var[a]; c1 = c * Ev[a] / (2 * var[a] * var[a]); }
            pe_c[((int64_t)e * p.K + k) * 2] = c0;
            pe_c[((int64_t)e * p.K + k) * 2 + 1] = c1;
        }
    }
}

// compiled for scopes of up to 3 and up to MAXA variables: every per-slot array lives in registers, and the three-slot
// build (all of the reference's MLN templates but robot mapping) needs half of them
template <int MAXA>
__global__ void __launch_bounds__(BLOCK) vi_factor_kernel(lhvi_graph_t g, lhvi_pots_t pots, lhvi_vi_t p, double* __restrict__ ef,
                                                         double* __restrict__ pe_c, double* __restrict__ pe_d,
                                                         const int32_t* __restrict__ list, int n_list) {
    const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= (int64_t)(list ? n_list : g.F) * p.K) return;
    const int f = list ? list[j / p.K] : (int)(j / p.K), k = (int)(j % p.K);
    const int64_t i = (int64_t)f * p.K + k;
    const int base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base;
    if (!list) {
        if (vi_is_cc(g, pots, f)) return;                  // served by vi_factor_cc_kernel
        if (MAXA == 3 ? arity > 3 : arity <= 3) return;    // the other build's factors
    }
    int vars[MAXA], len[MAXA], it[MAXA], idx[MAXA];
    double x[MAXA], wt[MAXA], Em[MAXA], Ev[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        vars[a] = a < arity ? g.edge_var[base + a] : 0;
        len[a] = a < arity ? axis_len(g, p, vars[a]) : 1;
        it[a] = 0; idx[a] = 0; x[a] = 0.0; wt[a] = 1.0; Em[a] = 0.0; Ev[a] = 0.0;
    }
    double E = 0.0;
    for (;;) {
        double w = 1.0;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a < arity) { const Node nd = axis_node(g, p, vars[a], k, it[a]); x[a] = nd.x; idx[a] = nd.idx; wt[a] = nd.w; w *= nd.w; }
        }
        const double F = F_of(g, pots, p, f, x, idx, vars, arity);
        E += w * F;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a < arity && is_hidden(g.var_value[vars[a]]) && v_cont(g, vars[a])) {
                const double* e = p.eta_c + ((int64_t)vars[a] * p.K + k) * 2;
                Em[a] += w * (F * (x[a] - e[0]));
                Ev[a] += w * (F * ((x[a] - e[0]) * (x[a] - e[0]) - e[1]));
            }
        }
        int a = arity - 1;
        for (; a >= 0; --a) {
            bool carry = true;
#pragma unroll
            for (int b = 0; b < MAXA; ++b) if (b == a) { if (++it[b] < len[b]) carry = false; else it[b] = 0; }
            if (!carry) break;
        }
        if (a < 0) break;
    }
    ef[i] = (g.fac_mult ? g.fac_mult[f] : 1.0) * E;
    // per-edge partials; only the first position of a variable in the scope contributes (f.nb.index(rv), LVI:112,146)
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        if (a >= arity) continue;
        const int e = base + a, v = vars[a];
        double c0 = 0.0, c1 = 0.0;
        bool first = true;
#pragma unroll
        for (int b = 0; b < MAXA; ++b) if (b < a && vars[b] == v) first = false;
        const bool hid = is_hidden(g.var_value[v]);
        const double c = g.edge_count ? g.edge_count[e] : 1.0;
        if (hid && first && v_cont(g, v)) {
            const double* et = p.eta_c + ((int64_t)v * p.K + k) * 2;
            c0 = c * Em[a] / et[1];
            c1 = c * Ev[a] / (2 * et[1] * et[1]);
        }
        pe_c[((int64_t)e * p.K + k) * 2] = c0;
        pe_c[((int64_t)e * p.K + k) * 2 + 1] = c1;
        if (hid && !v_cont(g, v)) {
            const int D = v_nstates(g, v);
            for (int d = 0; d < D; ++d)
                pe_d[((int64_t)e * p.K + k) * p.Dmax + d] = first ? c * pinned_expectation<MAXA>(g, pots, p, f, base, arity, vars, a, d, k) : 0.0;
        }
    }
}

// ---- gather per variable + projection ----------------------------------------------------------------------
// thread per (variable, k) up to VI_HUB_DEGREE incident edges; a wavefront per (variable, k) beyond (the template
// variables of a relational model touch thousands of factors: one thread walking them serialises the whole launch)
constexpr int VI_HUB_DEGREE = LHVI_HUB_DEGREE;
__global__ void __launch_bounds__(BLOCK) vi_gather_kernel(lhvi_graph_t g, lhvi_vi_t p, const double* __restrict__ pe_c,
                                                         const double* __restrict__ pe_d, double* __restrict__ g_c,
                                                         double* __restrict__ g_d) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)g.V * p.K) return;
    const int v = (int)(i / p.K), k = (int)(i % p.K);
    if (!is_hidden(g.var_value[v])) return;
    if (g.var_ptr[v + 1] - g.var_ptr[v] > VI_HUB_DEGREE) return;          // hubs: vi_gather_hub_kernel
    if (v_cont(g, v)) {
        double a = g_c[i * 2], b = g_c[i * 2 + 1];
        for (int j = g.var_ptr[v]; j < g.var_ptr[v + 1]; ++j) {
            const int e = g.var_edge[j];
            a -= pe_c[((int64_t)e * p.K + k) * 2];
            b -= pe_c[((int64_t)e * p.K + k) * 2 + 1];
        }
        g_c[i * 2] = a; g_c[i * 2 + 1] = b;
    } else {
        const int D = v_nstates(g, v);
        double* row = g_d + i * p.Dmax;
        const double* eta = p.eta_d + i * p.Dmax;
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            double acc = row[d];
            for (int j = g.var_ptr[v]; j < g.var_ptr[v + 1]; ++j) acc -= pe_d[((int64_t)g.var_edge[j] * p.K + k) * p.Dmax + d];
            row[d] = acc;
            s += acc * eta[d];
        }
        for (int d = 0; d < D; ++d) row[d] = eta[d] * (row[d] - s);      // eta * (g - sum(g * eta)) (VI:160)
    }
}

__global__ void __launch_bounds__(BLOCK) vi_gather_hub_kernel(lhvi_graph_t g, lhvi_vi_t p, const double* __restrict__ pe_c,
                                                             const double* __restrict__ pe_d, double* __restrict__ g_c,
                                                             double* __restrict__ g_d) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6);
    if (w >= (int64_t)(g.hub_vars ? g.n_hubs : g.V) * p.K) return;
    const int v = g.hub_vars ? g.hub_vars[w / p.K] : (int)(w / p.K), k = (int)(w % p.K);
    const int64_t i = (int64_t)v * p.K + k;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= VI_HUB_DEGREE || !is_hidden(g.var_value[v])) return;
    if (v_cont(g, v)) {
        double a = 0.0, b = 0.0;
        for (int j = lo + lane; j < hi; j += 64) {
            const int e = g.var_edge[j];
            a += pe_c[((int64_t)e * p.K + k) * 2];
            b += pe_c[((int64_t)e * p.K + k) * 2 + 1];
        }
        a = dpp_wave_reduce(a, SumOp()); b = dpp_wave_reduce(b, SumOp());
        if (lane == 0) { g_c[i * 2] -= a; g_c[i * 2 + 1] -= b; }
    } else {
        const int D = v_nstates(g, v);
        double* row = g_d + i * p.Dmax;
        const double* eta = p.eta_d + i * p.Dmax;
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            double acc = 0.0;
            for (int j = lo + lane; j < hi; j += 64) acc += pe_d[((int64_t)g.var_edge[j] * p.K + k) * p.Dmax + d];
            acc = row[d] - dpp_wave_reduce(acc, SumOp());
            if (lane == 0) row[d] = acc;
            s += acc * eta[d];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) for (int d = 0; d < D; ++d) row[d] = eta[d] * (row[d] - s);
    }
}

// ---- mixture-weight gradient and free energy: two-stage reduction in a fixed order (deterministic) ------------
// stage 1: block b sums a contiguous slice of the concatenated (rvterm | ef) rows for every k -> partial[b][k]
constexpr int VI_RED_BLOCKS = 512;
__global__ void __launch_bounds__(BLOCK) vi_weights_partial_kernel(int64_t nv, int64_t nf, lhvi_vi_t p, const double* __restrict__ rvterm,
                                                                  const double* __restrict__ ef, double* __restrict__ partial) {
    __shared__ double red[BLOCK];
    const int64_t n = nv + nf;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int k = 0; k < p.K; ++k) {
        double acc = 0.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) acc += i < nv ? rvterm[i * p.K + k] : ef[(i - nv) * p.K + k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * p.K + k] = red[0];
        __syncthreads();
    }
}

// stage 2: one workgroup folds the partials (ascending block order) and applies the softmax-Jacobian projection
__global__ void __launch_bounds__(BLOCK) vi_weights_kernel(int nblocks, lhvi_vi_t p, const double* __restrict__ partial,
                                                          double* __restrict__ g_w, double* __restrict__ fe) {
    __shared__ double red[BLOCK];
    __shared__ double gw[64];
    for (int k = 0; k < p.K; ++k) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += BLOCK) acc += partial[(int64_t)i * p.K + k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) gw[k] = -red[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double dot = 0.0;
        for (int k = 0; k < p.K; ++k) dot += gw[k] * p.w[k];
        for (int k = 0; k < p.K; ++k) g_w[k] = p.w[k] * (gw[k] - dot);     // w * (g - sum(g * w)) (VI:90)
        fe[0] = dot;                                                         // free energy = sum_k w_k gw_k (VI:162-195)
    }
}

// ---- utils.log_likelihood (utils.py:6-15) on flat arrays: -sum_f log phi_f(x), -inf as soon as a factor vanishes ------
__global__ void __launch_bounds__(BLOCK) loglik_factor_kernel(lhvi_graph_t g, lhvi_pots_t pots, const double* __restrict__ x,
                                                             double* __restrict__ term) {
    const int f = blockIdx.x * BLOCK + threadIdx.x;
    if (f >= g.F) return;
    const int base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base;
    double xs[LHVI_MAX_ARITY];
    int idx[LHVI_MAX_ARITY];
#pragma unroll
    for (int a = 0; a < LHVI_MAX_ARITY; ++a) {
        xs[a] = 0.0; idx[a] = 0;
        if (a < arity) { const int v = g.edge_var[base + a]; xs[a] = x[v]; idx[a] = vi_state_index(g, v, xs[a]); }
    }
    const int pot = g.fac_pot[f];
    const double phi = pot_value(pots.kind[pot], pots.param + pots.off[pot], xs, idx);
    term[f] = phi == 0.0 ? -__builtin_huge_val() : log(phi);
}

__global__ void __launch_bounds__(BLOCK) loglik_partial_kernel(int64_t n, const double* __restrict__ term, double* __restrict__ partial) {
    __shared__ double red[BLOCK];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += BLOCK) acc += term[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(BLOCK) loglik_final_kernel(int nblocks, const double* __restrict__ partial, double* __restrict__ out) {
    __shared__ double red[BLOCK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += BLOCK) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    // a vanishing factor makes the sum -inf; the reference then returns -inf (not +inf) -- keep its sign convention
    if (threadIdx.x == 0) out[0] = red[0] == -__builtin_huge_val() ? -__builtin_huge_val() : -red[0];
}

// one ADAM step of one element (VI:255-287).  Contraction is off inside, so that the per-array kernel below and
// the fused update kernel give the same bits whatever the compiler does around them.
struct AdamConst { double lr, b1, b2, eps, c1, c2; };
__device__ __forceinline__ double adam_one(double theta, double& m, double& s, double gr, const AdamConst& o) {
#pragma clang fp contract(off)
    m = m * o.b1 + (1 - o.b1) * gr;                       // the reference's expressions, one rounding per operation
    s = s * o.b2 + (1 - o.b2) * gr * gr;
    return theta - (o.lr * (m / o.c1)) / (sqrt(s / o.c2) + o.eps);
}

__global__ void __launch_bounds__(BLOCK) adam_kernel(double* __restrict__ theta, double* __restrict__ m, double* __restrict__ s,
                                                    const double* __restrict__ grad, int64_t count, double c1, double c2, double lr,
                                                    double b1, double b2, double eps, int clip_stride, double clip_min) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const AdamConst o = {lr, b1, b2, eps, c1, c2};
    double mi = m[i], si = s[i];
    double th = adam_one(theta[i], mi, si, grad[i], o);
    m[i] = mi; s[i] = si;
    if (clip_stride > 0 && (i % clip_stride) == clip_stride - 1 && th < clip_min) th = clip_min;
    theta[i] = th;
}

__global__ void __launch_bounds__(BLOCK) softmax_rows_kernel(const double* __restrict__ tau, double* __restrict__ out, int64_t rows,
                                                            int cols, int stride) {
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= rows) return;
    double z = 0.0;
    for (int c = 0; c < cols; ++c) z += exp(tau[r * stride + c]);       // e ** x / sum (VI:32-38): no max shift
    for (int c = 0; c < cols; ++c) out[r * stride + c] = exp(tau[r * stride + c]) / z;
}

// ADAM_update body for all three parameter arrays in one launch (VI:255-287): thread i < K updates w_tau[i] (thread 0 then
// forms w = softmax(w_tau) -- it waits for nobody: K <= 64 elements are its own loop); the next V * K threads own one
// (variable, k) each: a hidden continuous variable's (mu, var) with var clipped at var_min, or a hidden discrete variable's
// category logits followed by that row's softmax.  Rows of observed variables never move (the Python path multiplies their
// gradients by a zero mask: m and s stay 0 and the step is 0).
__global__ void __launch_bounds__(BLOCK) vi_update_kernel(lhvi_graph_t g, lhvi_vi_t p, lhvi_vi_opt_t o, double c1, double c2) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const AdamConst ac = {o.lr, o.b1, o.b2, o.eps, c1, c2};
    if (i == 0) {
        double z = 0.0;
        for (int k = 0; k < p.K; ++k) {
            o.w_tau[k] = adam_one(o.w_tau[k], o.m_w[k], o.s_w[k], o.g_w[k], ac);
            z += exp(o.w_tau[k]);
        }
        for (int k = 0; k < p.K; ++k) o.w[k] = exp(o.w_tau[k]) / z;       // e ** x / sum (VI:32-38): no max shift
    }
    if (i >= (int64_t)g.V * p.K) return;
    const int v = (int)(i / p.K);
    if (!is_hidden(g.var_value[v])) return;
    if (v_cont(g, v)) {
        double* th = o.eta_c + i * 2;
        th[0] = adam_one(th[0], o.m_c[i * 2], o.s_c[i * 2], o.g_c[i * 2], ac);
        const double var = adam_one(th[1], o.m_c[i * 2 + 1], o.s_c[i * 2 + 1], o.g_c[i * 2 + 1], ac);
        th[1] = var < o.var_min ? o.var_min : var;
    } else {
        const int D = v_nstates(g, v);
        double* tau = o.tau_d + i * p.Dmax;
        double* eta = o.eta_d + i * p.Dmax;
        double z = 0.0;
        for (int d = 0; d < D; ++d) {
            tau[d] = adam_one(tau[d], o.m_d[i * p.Dmax + d], o.s_d[i * p.Dmax + d], o.g_d[i * p.Dmax + d], ac);
            z += exp(tau[d]);
        }
        for (int d = 0; d < D; ++d) eta[d] = exp(tau[d]) / z;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace lhvi

using namespace lhvi;

extern "C" {

size_t lhvi_vi_workspace_bytes(const lhvi_graph_t* g, const lhvi_vi_t* p) {
    if (!g || !p) return 0;
    const size_t K = (size_t)p->K, D = (size_t)p->Dmax;
    return align256((size_t)g->V * K * 8) + align256((size_t)g->F * K * 8) + align256((size_t)g->E * K * 16) +
           align256((size_t)g->E * K * D * 8) + align256((size_t)VI_RED_BLOCKS * K * 8) + 256;
}

int lhvi_vi_grad(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_vi_t* p, double* g_w, double* g_c, double* g_d,
                 double* fe, void* ws, size_t ws_bytes, void* stream) {
    if (!g || !pots || !p || !g_w || !g_c || !g_d || !fe || !ws) return LHVI_E_ARG;
    if (p->K <= 0 || p->K > 64 || p->T <= 0 || p->Dmax <= 0 || p->Dmax > VI_MAX_D) return LHVI_E_UNSUPPORTED;
    if (ws_bytes < lhvi_vi_workspace_bytes(g, p)) return LHVI_E_ARG;
    if (!p->gh_x || !p->gh_w || !p->w || !p->eta_c || !p->eta_d) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    char* base = (char*)ws;
    const size_t K = (size_t)p->K;
    double* rvterm = (double*)base; base += align256((size_t)g->V * K * 8);
    double* ef = (double*)base; base += align256((size_t)g->F * K * 8);
    double* pe_c = (double*)base; base += align256((size_t)g->E * K * 16);
    double* pe_d = (double*)base; base += align256((size_t)g->E * K * (size_t)p->Dmax * 8);
    double* partial = (double*)base;
    if (g->V > 0)
        hipLaunchKernelGGL(vi_var_kernel, dim3(grid_for((int64_t)g->V * p->K)), dim3(BLOCK), 0, st, *g, *p, rvterm, g_c, g_d);
    if (g->F > 0 && p->fac_list) {
        // the caller's split of the factors (lhvi_vi_t.fac_list): [pairwise continuous | tiny grids | group kernel, arity <= 3 | group kernel,
        // arity 4..6 | thread-per-factor kernels for what fits neither]
        const int32_t* l = p->fac_list;
        const bool slim = pots->interpreted == 0;          // no formula of this graph goes through the bytecode interpreter
        if (p->n_cc < 0 || p->n_tiny < 0 || p->n_grp3 < 0 || p->n_grp6 < 0 || p->n_rest3 < 0 || p->n_rest6 < 0 ||
            (int64_t)p->n_cc + p->n_tiny + p->n_grp3 + p->n_grp6 + p->n_rest3 + p->n_rest6 != g->F) return LHVI_E_ARG;
        if (p->n_cc > 0) {
            const int64_t want = ((int64_t)p->n_cc * p->K + BLOCK - 1) / BLOCK;
            const dim3 grid((unsigned)(want < 2048 ? want : 2048));
            if (p->T <= 3)
                hipLaunchKernelGGL(vi_factor_cc_tab_kernel<3>, grid, dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_cc);
            else if (p->T <= 5)
                hipLaunchKernelGGL(vi_factor_cc_tab_kernel<5>, grid, dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_cc);
            else
                hipLaunchKernelGGL(vi_factor_cc_kernel, grid, dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_cc);
        }
        l += p->n_cc;
        if (p->n_tiny > 0) {
            if (!p->edge_axis || p->K > VI_TINY_K || p->tiny_par_words <= 0 || p->tiny_par_words > VI_TINY_PAR) return LHVI_E_ARG;
            // (4 096 workgroups, not the 512 resident ones: the hardware's dispatch balances the uneven items; 0.68 vs 0.76 ms)
            if (slim && p->tiny_par_words <= VI_TINY_PAR_SLIM)
                hipLaunchKernelGGL(vi_factor_tiny_kernel<false>, dim3(min(grid_for((int64_t)p->n_tiny * p->K), 4096u)), dim3(BLOCK), 0, st,
                                   *g, *pots, *p, ef, pe_c, pe_d, l, p->n_tiny);
            else
                hipLaunchKernelGGL(vi_factor_tiny_kernel<true>, dim3(min(grid_for((int64_t)p->n_tiny * p->K), 4096u)), dim3(BLOCK), 0, st,
                                   *g, *pots, *p, ef, pe_c, pe_d, l, p->n_tiny);
        }
        l += p->n_tiny;
        constexpr int GPB = VI_GRP_BLOCK / VI_GRP_L;
        if (p->n_grp3 > 0) {
            const dim3 grid(min(grid_for((int64_t)p->n_grp3 * p->K, GPB), 4096u));
            if (slim) hipLaunchKernelGGL((vi_factor_group_kernel<3, VI_GRP_L, false>), grid, dim3(VI_GRP_BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_grp3);
            else hipLaunchKernelGGL((vi_factor_group_kernel<3, VI_GRP_L, true>), grid, dim3(VI_GRP_BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_grp3);
        }
        l += p->n_grp3;
        if (p->n_grp6 > 0) {
            const dim3 grid(min(grid_for((int64_t)p->n_grp6 * p->K, GPB), 4096u));
            if (slim) hipLaunchKernelGGL((vi_factor_group_kernel<LHVI_MAX_ARITY, VI_GRP_L, false>), grid, dim3(VI_GRP_BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_grp6);
            else hipLaunchKernelGGL((vi_factor_group_kernel<LHVI_MAX_ARITY, VI_GRP_L, true>), grid, dim3(VI_GRP_BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_grp6);
        }
        l += p->n_grp6;
        if (p->n_rest3 > 0)
            hipLaunchKernelGGL(vi_factor_kernel<3>, dim3(grid_for((int64_t)p->n_rest3 * p->K)), dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_rest3);
        l += p->n_rest3;
        if (p->n_rest6 > 0)
            hipLaunchKernelGGL(vi_factor_kernel<LHVI_MAX_ARITY>, dim3(grid_for((int64_t)p->n_rest6 * p->K)), dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d, l, p->n_rest6);
    } else if (g->F > 0) {
        {
            const int64_t want = ((int64_t)g->F * p->K + BLOCK - 1) / BLOCK;
            hipLaunchKernelGGL(vi_factor_cc_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d,
                               (const int32_t*)nullptr, 0);
        }
        hipLaunchKernelGGL(vi_factor_kernel<3>, dim3(grid_for((int64_t)g->F * p->K)), dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d,
                           (const int32_t*)nullptr, 0);
        hipLaunchKernelGGL(vi_factor_kernel<LHVI_MAX_ARITY>, dim3(grid_for((int64_t)g->F * p->K)), dim3(BLOCK), 0, st, *g, *pots, *p, ef, pe_c, pe_d,
                           (const int32_t*)nullptr, 0);
    }
    if (g->V > 0)
    {
        hipLaunchKernelGGL(vi_gather_kernel, dim3(grid_for((int64_t)g->V * p->K)), dim3(BLOCK), 0, st, *g, *p, pe_c, pe_d, g_c, g_d);
        const int64_t nh = g->hub_vars ? g->n_hubs : g->V;
        if (nh > 0)
            hipLaunchKernelGGL(vi_gather_hub_kernel, dim3(grid_for(nh * p->K * WAVE)), dim3(BLOCK), 0, st, *g, *p, pe_c, pe_d, g_c, g_d);
    }
    const int64_t nred = (int64_t)g->V + g->F;
    const int nblocks = (int)(nred < VI_RED_BLOCKS ? (nred > 0 ? nred : 1) : VI_RED_BLOCKS);
    hipLaunchKernelGGL(vi_weights_partial_kernel, dim3(nblocks), dim3(BLOCK), 0, st, (int64_t)g->V, (int64_t)g->F, *p, rvterm, ef, partial);
    hipLaunchKernelGGL(vi_weights_kernel, dim3(1), dim3(BLOCK), 0, st, nblocks, *p, partial, g_w, fe);
    return check_launch();
}

int lhvi_vi_adam_run(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_vi_t* p, const lhvi_vi_opt_t* o, int32_t iterations,
                     double* fe_log, void* ws, size_t ws_bytes, void* stream) {
    if (!g || !pots || !p || !o || iterations < 0) return LHVI_E_ARG;
    if (!o->w_tau || !o->w || !o->eta_c || !o->tau_d || !o->eta_d || !o->m_w || !o->s_w || !o->m_c || !o->s_c || !o->m_d || !o->s_d ||
        !o->g_w || !o->g_c || !o->g_d || !o->fe) return LHVI_E_ARG;
    if (o->w != p->w || o->eta_c != p->eta_c || o->eta_d != p->eta_d) return LHVI_E_ARG;      // the step must see what it updates
    hipStream_t st = as_stream(stream);
    for (int i = 0; i < iterations; ++i) {
        // the free energy the reference logs after update i - 1 is the one this gradient pass computes (same parameters)
        double* fe = (fe_log && i > 0) ? fe_log + (i - 1) : o->fe;
        if (int rc = lhvi_vi_grad(g, pots, p, o->g_w, o->g_c, o->g_d, fe, ws, ws_bytes, stream)) return rc;
        const int t = o->t + i + 1;
        const double c1 = 1 - pow(o->b1, (double)t), c2 = 1 - pow(o->b2, (double)t);
        const int64_t n = (int64_t)g->V * p->K;
        hipLaunchKernelGGL(vi_update_kernel, dim3(grid_for(n > 1 ? n : 1)), dim3(BLOCK), 0, st, *g, *p, *o, c1, c2);
    }
    if (fe_log && iterations > 0)
        if (int rc = lhvi_vi_grad(g, pots, p, o->g_w, o->g_c, o->g_d, fe_log + (iterations - 1), ws, ws_bytes, stream)) return rc;
    return check_launch();
}

size_t lhvi_log_likelihood_workspace_bytes(const lhvi_graph_t* g) {
    return g ? align256((size_t)g->F * 8) + align256((size_t)VI_RED_BLOCKS * 8) + 256 : 0;
}

int lhvi_log_likelihood(const lhvi_graph_t* g, const lhvi_pots_t* pots, const double* x, double* out, void* ws, size_t ws_bytes,
                        void* stream) {
    if (!g || !pots || !x || !out || !ws) return LHVI_E_ARG;
    if (ws_bytes < lhvi_log_likelihood_workspace_bytes(g)) return LHVI_E_ARG;
    hipStream_t st = as_stream(stream);
    double* term = (double*)ws;
    double* partial = (double*)((char*)ws + align256((size_t)g->F * 8));
    const int nblocks = g->F < VI_RED_BLOCKS ? (g->F > 0 ? g->F : 1) : VI_RED_BLOCKS;
    if (g->F > 0)
        hipLaunchKernelGGL(loglik_factor_kernel, dim3(grid_for(g->F)), dim3(BLOCK), 0, st, *g, *pots, x, term);
    hipLaunchKernelGGL(loglik_partial_kernel, dim3(nblocks), dim3(BLOCK), 0, st, (int64_t)g->F, term, partial);
    hipLaunchKernelGGL(loglik_final_kernel, dim3(1), dim3(BLOCK), 0, st, nblocks, partial, out);
    return check_launch();
}

int lhvi_adam_step(double* theta, double* m, double* s, const double* grad, int64_t count, int32_t t, double lr, double b1,
                   double b2, double eps, int32_t clip_stride, double clip_min, void* stream) {
    if (count < 0 || t < 1) return LHVI_E_ARG;
    if (count == 0) return LHVI_OK;
    if (!theta || !m || !s || !grad) return LHVI_E_ARG;
    const double c1 = 1 - pow(b1, (double)t), c2 = 1 - pow(b2, (double)t);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(BLOCK), 0, as_stream(stream), theta, m, s, grad, count, c1, c2, lr,
                       b1, b2, eps, clip_stride, clip_min);
    return check_launch();
}

int lhvi_softmax_rows(const double* tau, double* out, int64_t rows, int32_t cols, int32_t stride, void* stream) {
    if (rows < 0 || cols <= 0 || stride < cols) return LHVI_E_ARG;
    if (rows == 0) return LHVI_OK;
    if (!tau || !out) return LHVI_E_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid_for(rows)), dim3(BLOCK), 0, as_stream(stream), tau, out, rows, cols, stride);
    return check_launch();
}

}  // extern "C"
