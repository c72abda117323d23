/*
 * potential_oracle.c -- phi(x) for every potential class of the reference, on the flat parameter rows
 * written by lhvi/potentials.py::device_spec.  TEST INFRASTRUCTURE ONLY.
 *
 * Each branch restates the `get` of one class in /root/reference/Potential.py or MLNPotential.py.
 * x[i] are the argument values, idx[i] the state index of a discrete argument (tables are indexed by state).
 */
#include <math.h>
#include "oracle.h"

/* postfix interpreter for traced MLN formulas (opcodes of lhvi/expr.py) */
static double mln_formula(const double *code, int ncode, const double *x) {
    double st[16];
    int sp = 0;
    for (int i = 0; i < ncode; ++i) {
        int op = (int)code[2 * i];
        double val = code[2 * i + 1];
        switch (op) {
            case 0: st[sp++] = x[(int)val]; break;
            case 1: st[sp++] = val; break;
            case 7: st[sp - 1] = -st[sp - 1]; break;
            case 8: st[sp - 1] = st[sp - 1] * st[sp - 1]; break;
            case 15: st[sp - 1] = fabs(st[sp - 1]); break;
            default: {
                double b = st[--sp], a = st[--sp], r = 0.0;
                switch (op) {
                    case 2: r = a + b; break;
                    case 3: r = a - b; break;
                    case 4: r = a * b; break;
                    case 5: r = a / b; break;
                    case 6: r = pow(a, b); break;
                    case 9: r = a == b; break;
                    case 10: r = a != b; break;
                    case 11: r = a < b; break;
                    case 12: r = a <= b; break;
                    case 13: r = a > b; break;
                    case 14: r = a >= b; break;
                }
                st[sp++] = r;
            }
        }
    }
    return st[0];
}

/* QuadraticPotential.get (Potential.py:84-94): e ** (x.(A x) + b.x + c) */
static double quadratic(const double *A, const double *b, double c, int n, const double *x) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) {
        double row = 0.0;
        for (int j = 0; j < n; ++j) row += A[i * n + j] * x[j];
        res += x[i] * row;
    }
    double lin = 0.0;
    for (int i = 0; i < n; ++i) lin += b[i] * x[i];
    res = res + lin;
    res += c;
    return pow(M_E, res);
}

double oracle_potential(int kind, const double *par, int arity, const double *x, const int *idx) {
    switch (kind) {
        case POT_TABLE: { /* TablePotential.get (Potential.py:20-21): table[parameters] */
            int nd = (int)par[0], off = 0;
            for (int i = 0; i < nd; ++i) off = off * (int)par[1 + i] + idx[i];
            return par[1 + nd + off];
        }
        case POT_GAUSSIAN: { /* GaussianPotential.get (Potential.py:51-54): pow(e, -0.5 * (x-mu) prec (x-mu)^T) */
            int n = (int)par[0];
            const double *mu = par + 1, *P = par + 1 + n;
            double d[MAX_ARITY], q = 0.0;
            for (int i = 0; i < n; ++i) d[i] = x[i] - mu[i];
            for (int j = 0; j < n; ++j) {
                double row = 0.0;
                for (int i = 0; i < n; ++i) row += d[i] * P[i * n + j];
                q += row * d[j];
            }
            return pow(M_E, -0.5 * q);
        }
        case POT_QUADRATIC: {
            int n = (int)par[0];
            return quadratic(par + 1, par + 1 + n * n, par[1 + n * n + n], n, x);
        }
        case POT_HYBRID_QUADRATIC: { /* HybridQuadraticPotential.get (Potential.py:295-305): args = [x_d, x_c] */
            int Nd = (int)par[0], Nc = (int)par[1], cfg = 0, ncfg = 1;
            for (int i = 0; i < Nd; ++i) { cfg = cfg * (int)par[2 + i] + (int)x[i]; ncfg *= (int)par[2 + i]; }
            const double *A = par + 2 + Nd + (long)cfg * Nc * Nc;
            const double *b = par + 2 + Nd + (long)ncfg * Nc * Nc + (long)cfg * Nc;
            double c = par[2 + Nd + (long)ncfg * Nc * Nc + (long)ncfg * Nc + cfg];
            return quadratic(A, b, c, Nc, x + Nd);
        }
        case POT_LINEAR_GAUSSIAN: { /* Potential.py:317-318 */
            double d = x[1] - par[0] * x[0];
            return exp(-(d * d) * 0.5 / par[1]);
        }
        case POT_X2: /* Potential.py:347-348 */
            return exp(-par[0] * (x[0] * x[0]) * 0.5 / par[1]);
        case POT_XY: /* Potential.py:377-378 */
            return exp(-par[0] * x[0] * x[1] * 0.5 / par[1]);
        case POT_MLN: /* MLNPotential.get (MLNPotential.py:36-37): e ** (formula(x) * w) */
            return pow(M_E, mln_formula(par + 3, (int)par[1], x) * par[0]);      /* row = [w, ncode, cq_off, program, ...]: the oracle interprets the program */
        case POT_MLN_HARD: /* MLNHardPotential.get (MLNPotential.py:48-49) */
            return mln_formula(par + 3, (int)par[1], x) > 0 ? 1.0 : 0.0;
        case POT_IMAGE_NODE: { /* Potential.py:406-408 */
            double u = (x[0] - x[1] - par[0]) / par[1];
            return exp(-u * u * 0.5) / (2.506628274631 * par[1]);
        }
        case POT_IMAGE_EDGE: { /* Potential.py:419-424 */
            double d = fabs(x[0] - x[1]);
            if (d > par[2]) return d * par[0] + par[3];
            return d * par[0] + pow(M_E, -d / par[1]);
        }
    }
    (void)arity;
    return NAN;
}
