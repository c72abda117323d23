/*
 * pbp_oracle.c -- CPU restatement of the reference's particle BP sweep (EPBP / HybridLBP) on flat arrays.
 * TEST INFRASTRUCTURE ONLY (checker for tests/smoke, reported cpu_baseline in bench.py).
 *
 * Reference: EPBPLogVersion.py:30-215,225-289 and HybridLBPLogVersion.py:44-236,430-536
 * (restated in SURVEY.md Appendix A.3).  Log messages are tabulated per edge:
 *   f2v[e][0..n)   at the variable's current particles,  f2v[e][n..n+T) at its integral points,
 *   v2f[e][0..n)   at the variable's particles.
 * Pinned against tests/golden/pbp_*.npz (captured from the reference by oracle/capture_pbp.py).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"

static int hidden(double v) { return v != v; }
static int canon(const ograph_t *g, int e) { return g->edge_canon ? g->edge_canon[e] : e; }

/* EPBP.norm_pdf (EPBPLogVersion.py:49-53): sig is a standard deviation here */
static double norm_pdf_std(double x, double mu, double sig) {
    double u = (x - mu) / sig;
    return exp(-u * u * 0.5) / (2.506628274631 * sig);
}

/* dict-key collapse of a sample (EPBPLogVersion.py:236-242,253-256): first occurrence of each value */
void oracle_pbp_uniq(const ograph_t *g, int n, const double *particles, const int32_t *np_, uint8_t *uniq) {
    for (int v = 0; v < g->V; ++v)
        for (int j = 0; j < n; ++j) {
            int u = j < np_[v];
            for (int i = 0; i < j && u; ++i)
                if (particles[(long)v * n + i] == particles[(long)v * n + j]) u = 0;
            uniq[(long)v * n + j] = (uint8_t)u;
        }
}

/* important_weight (EPBP:156-163, HLBP:173-180) */
static double important_weight(const ograph_t *g, const opbp_t *s, int v, double x) {
    int d = g->var_dom[v];
    if (g->dom_cont[d]) {
        if (x == g->dom_lo[d] || x == g->dom_hi[d]) return 1e-200;
    } else {
        if (!(s->flags & 2u)) return 1.0;                      /* HLBP: continuous rvs only */
        const double *vals = g->dom_val + g->dom_ptr[d];       /* EPBP: tests domain.values[0|1] on every hidden rv */
        int ns = g->dom_ptr[d + 1] - g->dom_ptr[d];
        if (x == vals[0] || (ns > 1 && x == vals[1])) return 1e-200;
    }
    double mu = s->q[2 * v], sd = sqrt(s->q[2 * v + 1]);
    double p = norm_pdf_std(x, mu, sd);
    return 1.0 / (p > 1e-200 ? p : 1e-200);
}

/* message_rv_to_f for all particles of all (rv, f), then log_message_balance (EPBP:165-174,204-215; HLBP:182-191,225-236) */
void oracle_pbp_v2f(const ograph_t *g, const opbp_t *s, const double *f2v, double *v2f) {
    const int n = s->n, S = s->n + s->T;
    for (int v = 0; v < g->V; ++v) {
        if (!hidden(g->var_value[v])) continue;
        const int np_ = s->np[v];
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k) {
            const int e = g->var_edge[k];
            double *out = v2f + (long)e * n;
            for (int j = 0; j < np_; ++j) {
                double x = s->particles[(long)v * n + j];
                double res = 0.0;
                for (int kk = g->var_ptr[v]; kk < g->var_ptr[v + 1]; ++kk) {
                    if (kk == k) continue;
                    int e2 = g->var_edge[kk];
                    double m = f2v[(long)e2 * S + j];
                    res += g->edge_count ? m * g->edge_count[e2] : m;
                }
                res = res + log(important_weight(g, s, v, x));
                if (g->edge_count) res = res + f2v[(long)e * S + j] * (g->edge_count[e] - 1.0);
                out[j] = res;
            }
            /* balance over the distinct keys: mean is exact in the reference (statistics.mean) -> compensated sum */
            double sum = 0.0, comp = 0.0, mx = -INFINITY;
            int cnt = 0;
            for (int j = 0; j < np_; ++j) {
                if (!s->uniq[(long)v * n + j]) continue;
                double y = out[j], t = sum + y;
                comp += fabs(sum) >= fabs(y) ? (sum - t) + y : (y - t) + sum;
                sum = t;
                if (y > mx) mx = y;
                ++cnt;
            }
            double mean = (sum + comp) / cnt;
            double shift = (mx - mean > s->max_log_value) ? mx - s->max_log_value : mean;
            for (int j = 0; j < np_; ++j) out[j] -= shift;
        }
    }
}

static int state_index(const ograph_t *g, int v, double x) {
    int d = g->var_dom[v];
    if (g->dom_cont[d]) return 0;
    for (int i = g->dom_ptr[d]; i < g->dom_ptr[d + 1]; ++i)
        if (g->dom_val[i] == x) return i - g->dom_ptr[d];
    return (int)x;
}

/* message_f_to_rv(x, f, rv, sample) (EPBP:176-194; HLBP:193-215) for the edge e = (f, rv) at point (x, xi) */
static double f2v_point(const ograph_t *g, const opbp_t *s, const double *v2f, const double *partner_particles,
                        int e, double x, int xi) {
    const int n = s->n;
    const int f = g->edge_fac[e], base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base;
    const int pos = e - base, tv = g->edge_var[e];
    const int pot = g->fac_pot[f], kind = g->pot_kind[pot];
    const double *par = g->pot_param + g->pot_off[pot];
    int cnt[MAX_ARITY], var[MAX_ARITY], withmsg[MAX_ARITY], fixed[MAX_ARITY], ce[MAX_ARITY];
    double xs[MAX_ARITY];
    int ix[MAX_ARITY], it[MAX_ARITY];
    for (int a = 0; a < arity; ++a) {
        var[a] = g->edge_var[base + a];
        ce[a] = canon(g, base + a);
        it[a] = 0;
        if (a == pos) { cnt[a] = 1; fixed[a] = 1; withmsg[a] = 0; xs[a] = x; ix[a] = xi; }
        else if (hidden(g->var_value[var[a]])) {
            cnt[a] = s->np[var[a]]; fixed[a] = 0;
            withmsg[a] = var[a] != tv;       /* HLBP: a repeated cluster iterates its particles without a message */
        } else { cnt[a] = 1; fixed[a] = 1; withmsg[a] = 0; xs[a] = g->var_value[var[a]]; ix[a] = state_index(g, var[a], xs[a]); }
    }
    double res = 0.0;
    for (;;) {
        double m = 0.0;
        for (int a = 0; a < arity; ++a) {
            if (fixed[a]) continue;
            xs[a] = partner_particles[(long)var[a] * n + it[a]];
            ix[a] = it[a];
            if (withmsg[a]) m += v2f[(long)ce[a] * n + it[a]];
        }
        res += oracle_potential(kind, par, arity, xs, ix) * pow(M_E, m);
        int a = arity - 1;                     /* itertools.product order: last argument fastest */
        while (a >= 0) {
            if (!fixed[a] && ++it[a] < cnt[a]) break;
            it[a] = 0;
            --a;
        }
        if (a < 0) break;
    }
    return res > 0 ? log(res) : -700.0;
}

/* the f -> rv half sweep at the new particles and the integral points (EPBP:275-285; HLBP:518-528) */
void oracle_pbp_f2v(const ograph_t *g, const opbp_t *s, const double *v2f, double *f2v) {
    const int n = s->n, S = s->n + s->T;
#pragma omp parallel for schedule(dynamic, 64)
    for (int e = 0; e < g->E; ++e) {
        if (canon(g, e) != e) continue;
        const int v = g->edge_var[e];
        if (!hidden(g->var_value[v])) continue;
        const int d = g->var_dom[v];
        double *out = f2v + (long)e * S;
        for (int j = 0; j < s->np[v]; ++j)
            out[j] = f2v_point(g, s, v2f, s->old_particles, e, s->particles[(long)v * n + j], j);
        if (g->dom_cont[d]) {
            int T = g->dom_ptr[d + 1] - g->dom_ptr[d];
            for (int t = 0; t < T; ++t)
                out[n + t] = f2v_point(g, s, v2f, s->old_particles, e, g->dom_val[g->dom_ptr[d] + t], t);
        }
    }
}

/* belief_rv(x) = sum_f message_f_to_rv(x, f, rv, sample) (EPBP:196-202) for query variables at npts points each */
void oracle_pbp_belief_points(const ograph_t *g, const opbp_t *s, const double *v2f, int nq, const int32_t *qvar,
                              int npts, const double *x, double *out) {
    for (int qi = 0; qi < nq; ++qi) {
        int v = qvar[qi];
        for (int p = 0; p < npts; ++p) {
            double xv = x[(long)qi * npts + p], res = 0.0;
            int xi = state_index(g, v, xv);
            for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k) {
                /* HybridLBP.belief_rv_query (HLBP:313-317) walks the GROUND rv's factors, so a lifted edge is
                   added count times */
                double m = f2v_point(g, s, v2f, s->particles, g->var_edge[k], xv, xi);
                res += g->edge_count ? m * g->edge_count[g->var_edge[k]] : m;
            }
            out[(long)qi * npts + p] = res;
        }
    }
}

/* message_f_to_rv(x, f, rv, sample) for explicit (edge, point) pairs (HLBP.belief_rv_query sums these, HLBP:313-317) */
void oracle_pbp_edge_points(const ograph_t *g, const opbp_t *s, const double *v2f, int nq, const int32_t *qedge,
                            int npts, const double *x, double *out) {
    for (int qi = 0; qi < nq; ++qi)
        for (int p = 0; p < npts; ++p) {
            double xv = x[(long)qi * npts + p];
            int e = qedge[qi];
            out[(long)qi * npts + p] = f2v_point(g, s, v2f, s->particles, e, xv, state_index(g, g->edge_var[e], xv));
        }
}

/* gaussian_division (EPBP:43-47) */
static void gdiv(double a0, double a1, double b0, double b1, double *mu, double *sig) {
    *sig = a1 * b1 / (b1 - a1);
    *mu = (a0 * (b1 + *sig) - b0 * *sig) / b1;
}

/* eta_approximation_simple (EPBP:103-121) with optional cavity weighting (eta_approximation, EPBP:123-154) */
static void moments(const double *msg, const double *grid, int T, int use_cavity, double cmu, double csd,
                    double *mu, double *sig) {
    double w[4096], z = 0.0, a = 0.0, b = 0.0;
    for (int t = 0; t < T; ++t) {
        w[t] = pow(M_E, msg[t]);
        if (use_cavity) w[t] = w[t] * norm_pdf_std(grid[t], cmu, csd);
    }
    for (int t = 0; t < T; ++t) z += w[t];
    for (int t = 0; t < T; ++t) { a += w[t] * grid[t]; b += w[t] * (grid[t] * grid[t]); }
    *mu = a / z;
    *sig = b / z - *mu * *mu;
}

/* update_proposal (EPBP:83-101; HLBP:100-118) */
void oracle_pbp_proposal(const ograph_t *g, const opbp_t *s, const double *f2v, double *eta, double *q) {
    const int n = s->n, S = s->n + s->T;
    for (int v = 0; v < g->V; ++v) {
        int d = g->var_dom[v];
        if (!hidden(g->var_value[v]) || !g->dom_cont[d]) continue;
        const double *grid = g->dom_val + g->dom_ptr[d];
        int T = g->dom_ptr[d + 1] - g->dom_ptr[d];
        double total = 0.0;
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k)
            total += g->edge_count ? g->edge_count[g->var_edge[k]] : 1.0;
        double min_sig = total * s->var_threshold;
        double pm = 0.0, ps = 0.0;
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k) {
            int e = g->var_edge[k];
            const double *msg = f2v + (long)e * S + n;
            double mu, sig;
            double a0 = q[2 * v], a1 = q[2 * v + 1], b0 = eta[2 * e], b1 = eta[2 * e + 1];
            if (!(s->flags & 1u) || a1 >= b1) {
                moments(msg, grid, T, 0, 0, 0, &mu, &sig);
            } else {
                double c0, c1, m0, m1;
                gdiv(a0, a1, b0, b1, &c0, &c1);
                moments(msg, grid, T, 1, c0, sqrt(c1), &m0, &m1);
                gdiv(m0, m1, c0, c1, &mu, &sig);
            }
            if (0 < sig && sig < INFINITY) {
                sig = sig > min_sig ? sig : min_sig;
                eta[2 * e] = mu; eta[2 * e + 1] = sig;
            } else {
                mu = eta[2 * e]; sig = eta[2 * e + 1];
            }
            double c = g->edge_count ? g->edge_count[e] : 1.0;
            double p = 1.0 / sig;
            if (g->edge_count) { ps += p * c; pm += p * mu * c; }   /* HLBP.gaussian_product (HLBP:44-54) */
            else               { ps += p;     pm += p * mu; }       /* EPBP.gaussian_product (EPBP:30-41) */
        }
        ps = 1.0 / ps;
        q[2 * v] = ps * pm; q[2 * v + 1] = ps;
    }
}

/* initial_proposal + zero messages (EPBP:72-81,233-242; HLBP:89-98,449-458) */
void oracle_pbp_init(const ograph_t *g, const opbp_t *s, double *eta, double *q, double *f2v, double *v2f) {
    const int n = s->n, S = s->n + s->T;
    memset(f2v, 0, sizeof(double) * (size_t)g->E * S);
    memset(v2f, 0, sizeof(double) * (size_t)g->E * n);
    for (int v = 0; v < g->V; ++v) {
        if (!hidden(g->var_value[v])) continue;
        int d = g->var_dom[v];
        if (!g->dom_cont[d] && !(s->flags & 2u)) continue;         /* HLBP: continuous only */
        q[2 * v] = 0.0; q[2 * v + 1] = 5.0;
        double total = 0.0;
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k)
            total += g->edge_count ? g->edge_count[g->var_edge[k]] : 1.0;
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k) {
            int e = g->var_edge[k];
            eta[2 * e] = 0.0; eta[2 * e + 1] = 5.0 * total;
        }
    }
}
