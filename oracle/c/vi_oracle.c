/*
 * vi_oracle.c -- CPU restatement of the reference's mixture variational step on flat arrays.
 * TEST INFRASTRUCTURE ONLY.  Reference: VarInference.py:26-195,249-287 and LiftedVarInference.py:28-199
 * (SURVEY.md Appendix A.4).  Pinned against tests/golden/vi_*.npz (oracle/capture_vi.py).
 *
 * Parameters: w[K]; eta_c[V][K][2] = (mu, var) of continuous hidden variables; eta_d[V][K][Dmax] = category
 * probabilities of discrete hidden variables.  Lifted multipliers: var_mult = len(rv.rvs), fac_mult = len(f.factors),
 * edge_count = rv.count[f] (NULL pointers = ground graph, all 1).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

typedef struct {
    int32_t K, T, Dmax, quirks;
    const double *gh_x, *gh_w, *w, *eta_c, *eta_d;
    /* C2FVarInference.py:120-136,253-261: an evidence cluster whose members' values differ is a Gaussian observation
     * N(value, variance): obs_var[v] > 0 marks it (NULL: none).  It is integrated with T quadrature nodes like a hidden
     * continuous variable, multiplies every mixture component of a belief by its pdf, and has no parameters of its own. */
    const double *obs_var;
} ovi_t;

#define MAXN 64

typedef struct { int n; double x[MAXN], w[MAXN]; int idx[MAXN]; } axis_t;

static int hidden(double v) { return v != v; }
static int gobs(const ovi_t *p, int v) { return p->obs_var && p->obs_var[v] > 0.0; }
static int is_cont(const ograph_t *g, int v) { return g->dom_cont[g->var_dom[v]]; }
static int nstates(const ograph_t *g, int v) { int d = g->var_dom[v]; return g->dom_ptr[d + 1] - g->dom_ptr[d]; }
static const double *states(const ograph_t *g, int v) { return g->dom_val + g->dom_ptr[g->var_dom[v]]; }

static int state_index(const ograph_t *g, int v, double x) {
    if (is_cont(g, v)) return 0;
    const double *s = states(g, v);
    for (int i = 0; i < nstates(g, v); ++i) if (s[i] == x) return i;
    return (int)x;
}

/* VarInference.norm_pdf (VI:26-30): note the normaliser 2.5066 * var (sic) */
static double norm_pdf_var(double x, double mu, double var) {
    double u = x - mu;
    return pow(M_E, -u * u * 0.5 / var) / (2.506628274631 * var);
}

/* rvs_belief (VI:336-353): sum_k w_k prod_i comp_{k,i}(x_i); 0 if x disagrees with evidence */
static double rvs_belief(const ograph_t *g, const ovi_t *p, const double *x, const int *idx, const int *vars, int m) {
    double b[32];
    for (int k = 0; k < p->K; ++k) b[k] = p->w[k];
    for (int i = 0; i < m; ++i) {
        int v = vars[i];
        if (!hidden(g->var_value[v])) {
            if (gobs(p, v)) { for (int k = 0; k < p->K; ++k) b[k] *= norm_pdf_var(x[i], g->var_value[v], p->obs_var[v]); }
            else if (x[i] != g->var_value[v]) return 0.0;
        }
        else if (is_cont(g, v)) {
            for (int k = 0; k < p->K; ++k) { const double *e = p->eta_c + ((long)v * p->K + k) * 2; b[k] *= norm_pdf_var(x[i], e[0], e[1]); }
        } else {
            for (int k = 0; k < p->K; ++k) b[k] *= p->eta_d[((long)v * p->K + k) * p->Dmax + idx[i]];
        }
    }
    double s = 0.0;
    for (int k = 0; k < p->K; ++k) s += b[k];
    return s;
}

/* the (is_continuous, eta) argument of expectation() for variable v under component k (VI:64-70) */
static void axis_of(const ograph_t *g, const ovi_t *p, int v, int k, axis_t *a) {
    if (!hidden(g->var_value[v]) && gobs(p, v)) {
        a->n = p->T;
        for (int t = 0; t < p->T; ++t) { a->x[t] = sqrt(2 * p->obs_var[v]) * p->gh_x[t] + g->var_value[v]; a->w[t] = p->gh_w[t]; a->idx[t] = 0; }
    }
    else if (!hidden(g->var_value[v])) { a->n = 1; a->x[0] = g->var_value[v]; a->w[0] = 1.0; a->idx[0] = state_index(g, v, a->x[0]); }
    else if (is_cont(g, v)) {
        const double *e = p->eta_c + ((long)v * p->K + k) * 2;
        a->n = p->T;
        for (int t = 0; t < p->T; ++t) { a->x[t] = sqrt(2 * e[1]) * p->gh_x[t] + e[0]; a->w[t] = p->gh_w[t]; a->idx[t] = 0; }
    } else {
        a->n = nstates(g, v);
        for (int d = 0; d < a->n; ++d) { a->x[d] = states(g, v)[d]; a->w[d] = p->eta_d[((long)v * p->K + k) * p->Dmax + d]; a->idx[d] = d; }
    }
}

/* F_f(x) = log(phi(x) + 1e-100) - log(b_f(x) + 1e-100) */
static double F_of(const ograph_t *g, const ovi_t *p, int f, const double *x, const int *idx) {
    int base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base, pot = g->fac_pot[f];
    int vars[MAX_ARITY];
    for (int a = 0; a < arity; ++a) vars[a] = g->edge_var[base + a];
    double phi = oracle_potential(g->pot_kind[pot], g->pot_param + g->pot_off[pot], arity, x, idx);
    return log(phi + 1e-100) - log(rvs_belief(g, p, x, idx, vars, arity) + 1e-100);
}

/* tensor-product expectation over the factor's scope under component k of: 1, and per slot (x_i - mu), ((x_i-mu)^2 - var).
 * out_E = E_k[F]; out_m[a], out_v[a] = E_k[F (x_a - mu_a,k)], E_k[F ((x_a - mu_a,k)^2 - var_a,k)] for continuous hidden slots */
static void factor_expectations(const ograph_t *g, const ovi_t *p, int f, int k, double *out_E, double *out_m, double *out_v) {
    int base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base;
    axis_t ax[MAX_ARITY];
    int it[MAX_ARITY];
    double x[MAX_ARITY];
    int idx[MAX_ARITY];
    for (int a = 0; a < arity; ++a) { axis_of(g, p, g->edge_var[base + a], k, &ax[a]); it[a] = 0; out_m[a] = 0.0; out_v[a] = 0.0; }
    double E = 0.0;
    for (;;) {
        double w = 1.0;
        for (int a = 0; a < arity; ++a) { x[a] = ax[a].x[it[a]]; idx[a] = ax[a].idx[it[a]]; w *= ax[a].w[it[a]]; }
        double F = F_of(g, p, f, x, idx);
        E += w * F;
        for (int a = 0; a < arity; ++a) {
            int v = g->edge_var[base + a];
            if (hidden(g->var_value[v]) && is_cont(g, v)) {
                const double *e = p->eta_c + ((long)v * p->K + k) * 2;
                out_m[a] += w * (F * (x[a] - e[0]));
                out_v[a] += w * (F * ((x[a] - e[0]) * (x[a] - e[0]) - e[1]));
            }
        }
        int a = arity - 1;
        while (a >= 0) { if (++it[a] < ax[a].n) break; it[a] = 0; --a; }
        if (a < 0) break;
    }
    *out_E = E;
}

/* expectation over the OTHER slots with slot `pos` pinned to state d (gradient_category_tau, VI:133-160).
 * quirks=1 reproduces VI:147-150: every other hidden slot is integrated over the TARGET's domain values with the
 * other variable's eta row as weights, and zip(product(xs), product(ws)) pairs the two products positionally. */
static double pinned_expectation(const ograph_t *g, const ovi_t *p, int f, int pos, int d, int k) {
    int base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base;
    int tv = g->edge_var[base + pos];
    int Dt = nstates(g, tv);
    const double *tvals = states(g, tv);
    double x[MAX_ARITY];
    int idx[MAX_ARITY];
    if (!p->quirks) {
        axis_t ax[MAX_ARITY];
        int it[MAX_ARITY];
        for (int a = 0; a < arity; ++a) {
            it[a] = 0;
            if (a == pos) { ax[a].n = 1; ax[a].x[0] = tvals[d]; ax[a].w[0] = 1.0; ax[a].idx[0] = d; }
            else axis_of(g, p, g->edge_var[base + a], k, &ax[a]);
        }
        double E = 0.0;
        for (;;) {
            double w = 1.0;
            for (int a = 0; a < arity; ++a) { x[a] = ax[a].x[it[a]]; idx[a] = ax[a].idx[it[a]]; w *= ax[a].w[it[a]]; }
            E += w * F_of(g, p, f, x, idx);
            int a = arity - 1;
            while (a >= 0) { if (++it[a] < ax[a].n) break; it[a] = 0; --a; }
            if (a < 0) break;
        }
        return E;
    }
    /* quirk mode */
    int nx[MAX_ARITY], nw[MAX_ARITY], oth[MAX_ARITY], m = 0;
    double wl[MAX_ARITY][MAXN], xl[MAX_ARITY][MAXN];
    long totx = 1, totw = 1;
    for (int a = 0; a < arity; ++a) {
        if (a == pos) continue;
        int v = g->edge_var[base + a];
        oth[m] = a;
        if (!hidden(g->var_value[v]) && gobs(p, v)) {     /* C2FVI:221-223: (True, (value, variance)) */
            axis_t ax;
            axis_of(g, p, v, k, &ax);
            nx[m] = nw[m] = ax.n;
            for (int t = 0; t < ax.n; ++t) { xl[m][t] = ax.x[t]; wl[m][t] = ax.w[t]; }
        }
        else if (!hidden(g->var_value[v])) { nx[m] = 1; nw[m] = 1; wl[m][0] = 1.0; }
        else {
            nx[m] = Dt;
            if (is_cont(g, v)) { nw[m] = 2; wl[m][0] = p->eta_c[((long)v * p->K + k) * 2]; wl[m][1] = p->eta_c[((long)v * p->K + k) * 2 + 1]; }
            else { nw[m] = nstates(g, v); for (int s = 0; s < nw[m]; ++s) wl[m][s] = p->eta_d[((long)v * p->K + k) * p->Dmax + s]; }
        }
        totx *= nx[m]; totw *= nw[m];
        ++m;
    }
    long cnt = totx < totw ? totx : totw;
    double E = 0.0;
    for (long i = 0; i < cnt; ++i) {
        long rx = i, rw = i;
        double w = 1.0;
        int ixs[MAX_ARITY], iws[MAX_ARITY];
        for (int j = m - 1; j >= 0; --j) { ixs[j] = (int)(rx % nx[j]); rx /= nx[j]; iws[j] = (int)(rw % nw[j]); rw /= nw[j]; }
        for (int j = 0; j < m; ++j) {
            int a = oth[j], v = g->edge_var[base + a];
            w *= wl[j][iws[j]];
            if (!hidden(g->var_value[v]) && gobs(p, v)) { x[a] = xl[j][ixs[j]]; idx[a] = 0; }
            else if (!hidden(g->var_value[v])) { x[a] = g->var_value[v]; idx[a] = state_index(g, v, x[a]); }
            else { x[a] = tvals[ixs[j]]; idx[a] = state_index(g, v, x[a]); }
        }
        x[pos] = tvals[d]; idx[pos] = d;
        E += w * F_of(g, p, f, x, idx);
    }
    return E;
}

static int first_pos(const ograph_t *g, int f, int v) {
    for (int e = g->fac_ptr[f]; e < g->fac_ptr[f + 1]; ++e) if (g->edge_var[e] == v) return e - g->fac_ptr[f];
    return -1;
}

/* gradient_w_tau, gradient_mu_var, gradient_category_tau, free_energy for all variables (VI:57-195; LVI:59-199) */
void oracle_vi_grad(const ograph_t *g, const ovi_t *p, double *g_w, double *g_c, double *g_d, double *fe) {
    const int K = p->K;
    double gw[32], energy = 0.0;
    for (int k = 0; k < K; ++k) gw[k] = 0.0;
    memset(g_c, 0, sizeof(double) * (size_t)g->V * K * 2);
    memset(g_d, 0, sizeof(double) * (size_t)g->V * K * p->Dmax);
    /* variable terms */
    for (int v = 0; v < g->V; ++v) {
        double N = 0.0;   /* rv.N = ground degree of a member */
        for (int j = g->var_ptr[v]; j < g->var_ptr[v + 1]; ++j) N += g->edge_count ? g->edge_count[g->var_edge[j]] : 1.0;
        double Mv = g->var_mult ? g->var_mult[v] : 1.0;
        for (int k = 0; k < K; ++k) {
            axis_t ax;
            axis_of(g, p, v, k, &ax);
            double E = 0.0, Em = 0.0, Ev = 0.0;
            const double *e = p->eta_c + ((long)v * K + k) * 2;
            for (int t = 0; t < ax.n; ++t) {
                double R = (N - 1) * log(rvs_belief(g, p, &ax.x[t], &ax.idx[t], &v, 1) + 1e-100);
                E += ax.w[t] * R;
                if (hidden(g->var_value[v]) && is_cont(g, v)) {
                    Em += ax.w[t] * (R * (ax.x[t] - e[0]));
                    Ev += ax.w[t] * (R * ((ax.x[t] - e[0]) * (ax.x[t] - e[0]) - e[1]));
                }
            }
            gw[k] -= Mv * E;
            energy -= Mv * p->w[k] * E;
            if (hidden(g->var_value[v])) {
                if (is_cont(g, v)) {
                    g_c[((long)v * K + k) * 2] -= Em / e[1];
                    g_c[((long)v * K + k) * 2 + 1] -= Ev / (2 * e[1] * e[1]);
                } else {
                    for (int d = 0; d < ax.n; ++d)
                        g_d[((long)v * K + k) * p->Dmax + d] -= (N - 1) * log(rvs_belief(g, p, &ax.x[d], &ax.idx[d], &v, 1) + 1e-100);
                }
            }
        }
    }
    /* factor terms */
    for (int f = 0; f < g->F; ++f) {
        double Mf = g->fac_mult ? g->fac_mult[f] : 1.0;
        int base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base;
        for (int k = 0; k < K; ++k) {
            double E, Em[MAX_ARITY], Ev[MAX_ARITY];
            factor_expectations(g, p, f, k, &E, Em, Ev);
            gw[k] -= Mf * E;
            energy -= Mf * p->w[k] * E;
            for (int a = 0; a < arity; ++a) {
                int e = base + a, v = g->edge_var[e];
                if (!hidden(g->var_value[v])) continue;
                if (first_pos(g, f, v) != a) continue;          /* f.nb.index(rv): first position only */
                double c = g->edge_count ? g->edge_count[g->edge_canon ? g->edge_canon[e] : e] : 1.0;
                if (is_cont(g, v)) {
                    const double *et = p->eta_c + ((long)v * K + k) * 2;
                    g_c[((long)v * K + k) * 2] -= c * Em[a] / et[1];
                    g_c[((long)v * K + k) * 2 + 1] -= c * Ev[a] / (2 * et[1] * et[1]);
                } else {
                    for (int d = 0; d < nstates(g, v); ++d)
                        g_d[((long)v * K + k) * p->Dmax + d] -= c * pinned_expectation(g, p, f, a, d, k);
                }
            }
        }
    }
    /* softmax-Jacobian projections (VI:90,160) */
    double dot = 0.0;
    for (int k = 0; k < K; ++k) dot += gw[k] * p->w[k];
    for (int k = 0; k < K; ++k) g_w[k] = p->w[k] * (gw[k] - dot);
    for (int v = 0; v < g->V; ++v) {
        if (!hidden(g->var_value[v]) || is_cont(g, v)) continue;
        int D = nstates(g, v);
        for (int k = 0; k < K; ++k) {
            double *row = g_d + ((long)v * K + k) * p->Dmax;
            const double *eta = p->eta_d + ((long)v * K + k) * p->Dmax;
            double s = 0.0;
            for (int d = 0; d < D; ++d) s += row[d] * eta[d];
            for (int d = 0; d < D; ++d) row[d] = eta[d] * (row[d] - s);
        }
    }
    *fe = energy;
}

/* one ADAM step (VI:255-287) on a flat parameter array */
void oracle_adam_step(double *theta, double *m, double *s, const double *grad, long count, int t, double lr,
                      double b1, double b2, double eps, int clip_stride, double clip_min) {
    for (long i = 0; i < count; ++i) {
        m[i] = m[i] * b1 + (1 - b1) * grad[i];
        s[i] = s[i] * b2 + (1 - b2) * grad[i] * grad[i];
        double th = theta[i] - (lr * (m[i] / (1 - pow(b1, t)))) / (sqrt(s[i] / (1 - pow(b2, t))) + eps);
        if (clip_stride > 0 && i % clip_stride == clip_stride - 1 && th < clip_min) th = clip_min;
        theta[i] = th;
    }
}
