/*
 * gabp_oracle.c -- CPU restatement of the reference's Gaussian BP sweep on flat arrays.
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / reported baseline.  The product (liblhvi.so + lhvi/) never links or calls this.
 * Pinned against golden vectors captured from the reference itself (tests/golden/gauss_*.json,
 * produced by oracle/capture_golden.py).
 *
 * Layout: edges are (factor, position) incidences, factor-major; var_ptr/var_edge is the CSR of a
 * variable's incident edges in rv.nb order.  A message is (mu, var); var = NaN stands for the
 * reference's None variance, a (NaN, NaN) pair for a None message.
 *
 * Compile with -ffp-contract=off so that a*b+c rounds twice like CPython.
 */
#include <math.h>
#include <stdint.h>
#include "oracle.h"


static int hidden(double v) { return v != v; }

/* GaBP.message_rv_to_f (GaBP.py:20-35) / GaLBP.message_rv_to_f (GaLBP.py:21-39), for every (rv, f) */
void oracle_gabp_v2f(const ograph_t *g, const double *f2v, double *v2f) {
    for (int v = 0; v < g->V; ++v) {
        for (int k = g->var_ptr[v]; k < g->var_ptr[v + 1]; ++k) {
            int e = g->var_edge[k];
            if (!hidden(g->var_value[v])) { v2f[2 * e] = NAN; v2f[2 * e + 1] = NAN; continue; }
            double mu = 0.0, sig = 0.0;
            for (int j = g->var_ptr[v]; j < g->var_ptr[v + 1]; ++j) {
                int ej = g->var_edge[j];
                double nb_mu = f2v[2 * ej], nb_sig = f2v[2 * ej + 1];
                if (g->edge_count) {                       /* lifted: every nb, own factor count-1 times */
                    double count = g->edge_count[ej];
                    if (j == k) count = count - 1.0;
                    if (nb_sig != nb_sig) mu -= nb_mu * count;
                    else { double p = 1.0 / nb_sig; mu += p * nb_mu * count; sig += p * count; }
                } else {
                    if (j == k) continue;                  /* ground: nb != f */
                    if (nb_sig != nb_sig) mu -= nb_mu;
                    else { double p = 1.0 / nb_sig; mu += p * nb_mu; sig += p; }
                }
            }
            sig = 1.0 / sig;
            mu = sig * mu;
            v2f[2 * e] = mu; v2f[2 * e + 1] = sig;
        }
    }
}

/* GaBP.message_f_to_rv (GaBP.py:37-138) for every edge whose variable is hidden */
void oracle_gabp_f2v(const ograph_t *g, const double *v2f, double *f2v) {
    for (int f = 0; f < g->F; ++f) {
        int base = g->fac_ptr[f], arity = g->fac_ptr[f + 1] - base;
        int pot = g->fac_pot[f], kind = g->pot_kind[pot];
        const double *par = g->pot_param + g->pot_off[pot];
        for (int pos = 0; pos < arity; ++pos) {
            int e = base + pos;
            if (g->edge_canon && g->edge_canon[e] != e) continue;
            if (!hidden(g->var_value[g->edge_var[e]])) continue;
            double mu = 0.0, sig = INFINITY;
            if (kind == POT_X2) {
                double h = par[0], s = par[1];
                if (h != 0.0) { mu = 0.0; sig = s / h; }
            } else if (arity == 2 && (kind == POT_GAUSSIAN || kind == POT_LINEAR_GAUSSIAN || kind == POT_XY)) {
                int pe = base + (1 - pos);
                int pc = g->edge_canon ? g->edge_canon[pe] : pe;
                double y = g->var_value[g->edge_var[pe]];
                int ph = hidden(y);
                double u = ph ? v2f[2 * pc] : 0.0, s2 = ph ? v2f[2 * pc + 1] : 0.0;
                if (kind == POT_GAUSSIAN && (int)par[0] == 2) {
                    const double *m = par + 1, *a = par + 7;   /* a = sig ** -1 (np.matrix inverse) */
                    double a1, a2, a3, u1, u2;
                    if (pos == 1) { a1 = a[0]; a2 = a[1]; a3 = a[3]; u1 = m[0]; u2 = m[1]; }
                    else          { a1 = a[3]; a2 = a[1]; a3 = a[0]; u1 = m[1]; u2 = m[0]; }
                    if (ph) {
                        double a4 = 1.0 / s2;
                        double temp = a3 * (a4 + a1) - a2 * a2;
                        mu = a2 * a4 * (u1 - u) / temp + u2;
                        sig = 1.0 / (a3 - a2 * a2 / (a4 + a1));
                    } else {
                        mu = -u2 - a2 * (y - u1) / a3;
                        sig = 1.0 / a3;
                    }
                } else if (kind == POT_LINEAR_GAUSSIAN) {
                    double h = par[0], s1 = par[1];
                    if (h != 0.0) {
                        if (ph) {
                            if (pos == 0) { mu = u / h; sig = (s1 + s2) / (h * h); }
                            else          { mu = u * h; sig = s1 + s2 * (h * h); }
                        } else {
                            if (pos == 0) { mu = y / h; sig = s1 / (h * h); }
                            else          { mu = h * y; sig = s1; }
                        }
                    }
                } else if (kind == POT_XY) {
                    double h = par[0], s1 = par[1];
                    if (h != 0.0) {
                        if (ph) { mu = 2.0 * s1 * u / (h * s2); sig = -4.0 * (s1 * s1) / (h * h * s2); }
                        else    { mu = h * y / (2.0 * s1); sig = NAN; }
                    }
                }
            }
            f2v[2 * e] = mu; f2v[2 * e + 1] = sig;
        }
    }
}

/* GaBP.get_belief_params (GaBP.py:187-200) / GaLBP.map (GaLBP.py:201-217) */
void oracle_gabp_marginals(const ograph_t *g, const double *f2v, double *mu_var) {
    for (int v = 0; v < g->V; ++v) {
        double val = g->var_value[v];
        if (!hidden(val)) { mu_var[2 * v] = val; mu_var[2 * v + 1] = 0.0; continue; }
        double mu = 0.0, sig = 0.0;
        for (int j = g->var_ptr[v]; j < g->var_ptr[v + 1]; ++j) {
            int ej = g->var_edge[j];
            double nb_mu = f2v[2 * ej], nb_sig = f2v[2 * ej + 1];
            double count = g->edge_count ? g->edge_count[ej] : 1.0;
            if (g->edge_count) {
                if (nb_sig != nb_sig) mu -= nb_mu * count;
                else { double p = 1.0 / nb_sig; mu += p * nb_mu * count; sig += p * count; }
            } else {
                if (nb_sig != nb_sig) mu -= nb_mu;
                else { double p = 1.0 / nb_sig; mu += p * nb_mu; sig += p; }
            }
        }
        sig = 1.0 / sig;
        mu_var[2 * v] = sig * mu; mu_var[2 * v + 1] = sig;
    }
}

/* GaBP.run (GaBP.py:140-169): init (0,1); `iterations` sweeps; the last one skips f->rv */
void oracle_gabp_run(const ograph_t *g, double *f2v, double *v2f, int iterations) {
    for (int e = 0; e < g->E; ++e) { f2v[2 * e] = 0.0; f2v[2 * e + 1] = 1.0; v2f[2 * e] = 0.0; v2f[2 * e + 1] = 1.0; }
    for (int i = 0; i < iterations; ++i) {
        oracle_gabp_v2f(g, f2v, v2f);
        if (i < iterations - 1) oracle_gabp_f2v(g, v2f, f2v);
    }
}
