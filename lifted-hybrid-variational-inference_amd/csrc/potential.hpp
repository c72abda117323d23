// potential.hpp -- device evaluators of the reference's potential classes (Potential.py, MLNPotential.py).
//
// Parameter rows are written by lhvi/potentials.py::device_spec:
//   TABLE            [nd, d0..d(nd-1), table (row-major)]                     Potential.py:15-21
//   GAUSSIAN         [n, mu[n], prec[n*n], (sig**-1)[n*n]]                    Potential.py:35-54
//   QUADRATIC        [n, A[n*n], b[n], c]                                     Potential.py:69-94
//   HYBRID_QUADRATIC [Nd, Nc, dims[Nd], A[cfg][Nc*Nc], b[cfg][Nc], c[cfg]]    Potential.py:273-305  (args = [x_d.., x_c..])
//   LINEAR_GAUSSIAN  [coeff, sig]   exp(-(x1 - coeff x0)^2 / (2 sig))          Potential.py:311-318
//   X2               [coeff, sig]   exp(-coeff x0^2 / (2 sig))                 Potential.py:341-348
//   XY               [coeff, sig]   exp(-coeff x0 x1 / (2 sig))                Potential.py:371-378
//   MLN / MLN_HARD   [w, ncode, cq_off, (op, val) * ncode, cq block]          MLNPotential.py:30-49, lhvi/expr.py
//                    cq block at par + cq_off (cq_off = 0: none): [CQ_MAGIC, arity, Nd, Nc, role[arity], dims[Nd], coef[ncfg][6]]
//                    -- the formula per joint discrete state as a polynomial of degree <= 2 in its (at most two) continuous
//                    arguments, weight folded in: evaluated by cq_log_phi, a lookup and six multiply-adds, no bytecode
//   IMAGE_NODE       [mu, sig]; IMAGE_EDGE [distant, scaling, max_threshold, v]  Potential.py:400-424
//
// Evaluators return log(phi) for the exponential-family kinds (so the caller can fold the incoming log
// messages into a single exp) and phi itself otherwise; `is_log` says which.
#pragma once
#include "common.hpp"

namespace lhvi {

constexpr int MLN_STACK = 12;

// the evaluation stack of the formula interpreter: registers (a dynamically indexed array: every access is a select chain /
// waterfall) or a column of LDS (slot i of thread t at base[i * stride]: two instructions per access)
struct MlnRegStack {
    double v[MLN_STACK];
    __device__ __forceinline__ double get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, double x) { v[i] = x; }
};
template <int STRIDE>
struct MlnLdsStack {
    double* base;
    __device__ __forceinline__ double get(int i) const { return base[i * STRIDE]; }
    __device__ __forceinline__ void set(int i, double x) { base[i * STRIDE] = x; }
};

template <class Stack>
__device__ __forceinline__ double mln_formula_on(const double* __restrict__ code, int ncode, const double* x, Stack& st) {
    int sp = 0;
    for (int i = 0; i < ncode; ++i) {
        const int op = (int)code[2 * i];
        const double val = code[2 * i + 1];
        if (op == 0) { st.set(sp++, x[(int)val]); }
        else if (op == 1) { st.set(sp++, val); }
        else if (op == 7) { st.set(sp - 1, -st.get(sp - 1)); }
        else if (op == 8) { const double a = st.get(sp - 1); st.set(sp - 1, a * a); }
        else if (op == 15) { st.set(sp - 1, fabs(st.get(sp - 1))); }
        else {
            const double b = st.get(--sp), a = st.get(--sp);
            double r = 0.0;
            switch (op) {
                case 2: r = a + b; break;
                case 3: r = a - b; break;
                case 4: r = a * b; break;
                case 5: r = a / b; break;
                case 6: r = pow(a, b); break;
                case 9: r = (a == b) ? 1.0 : 0.0; break;
                case 10: r = (a != b) ? 1.0 : 0.0; break;
                case 11: r = (a < b) ? 1.0 : 0.0; break;
                case 12: r = (a <= b) ? 1.0 : 0.0; break;
                case 13: r = (a > b) ? 1.0 : 0.0; break;
                case 14: r = (a >= b) ? 1.0 : 0.0; break;
            }
            st.set(sp++, r);
        }
    }
    return st.get(0);
}

// log phi of an MLN formula through its conditional-quadratic block b = [CQ_MAGIC, arity, Nd, Nc, role[arity], dims[Nd],
// coef[ncfg][6]] (lhvi/expr.py::cq_block): cfg = mixed-radix index of the discrete arguments' states (first one most
// significant), (u, v) = the continuous arguments in argument order, log phi = (a00 u + axy v + b0) u + (a11 v + b1) v + c.
// Straight-line: no opcode fetch, no evaluation stack.  idx[a] = state index of a discrete argument.
__device__ __forceinline__ double cq_log_phi(const double* __restrict__ b, const double* x, const int* idx) {
    const int arity = (int)b[1], Nd = (int)b[2];
    const double* __restrict__ role = b + 4;
    const double* __restrict__ dims = role + arity;
    int cfg = 0;
    double uv[2] = {0.0, 0.0};
#pragma unroll
    for (int a = 0; a < LHVI_MAX_ARITY; ++a) {
        if (a >= arity) break;
        const int r = (int)role[a];
        if (r >= 0) cfg = cfg * (int)dims[r] + idx[a];
        else if (r == -1) uv[0] = x[a];
        else uv[1] = x[a];
    }
    const double* __restrict__ c = dims + Nd + 6 * cfg;
    const double u = uv[0], v = uv[1];
    return fma(fma(c[0], u, fma(c[1], v, c[3])), u, fma(fma(c[2], v, c[4]), v, c[5]));
}

__device__ __forceinline__ double mln_formula(const double* __restrict__ code, int ncode, const double* x) {
    MlnRegStack st;
    return mln_formula_on(code, ncode, x, st);
}

__device__ __forceinline__ double quad_form(const double* __restrict__ A, const double* __restrict__ b, double c,
                                            int n, const double* x) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) {
        double row = 0.0;
        for (int j = 0; j < n; ++j) row += A[i * n + j] * x[j];
        res += x[i] * row;
    }
    for (int i = 0; i < n; ++i) res += b[i] * x[i];
    return res + c;
}

// value of the potential at the joint assignment x (idx = state indices of discrete arguments).
// INTERP = false: the build for graphs without an interpreted formula (every MLN row carries a conditional-quadratic block, no
// hard formula: lhvi_pots_t.interpreted == 0) -- the bytecode loop and its evaluation stack are not compiled in.
template <bool INTERP = true, class Stack>
__device__ __forceinline__ double pot_eval_on(int kind, const double* __restrict__ par, const double* x, const int* idx,
                                              bool& is_log, Stack& st) {
    is_log = true;
    switch (kind) {
        case LHVI_POT_TABLE: {
            is_log = false;
            const int nd = (int)par[0];
            int off = 0;
            for (int i = 0; i < nd; ++i) off = off * (int)par[1 + i] + idx[i];
            return par[1 + nd + off];
        }
        case LHVI_POT_GAUSSIAN: {
            const int n = (int)par[0];
            const double* mu = par + 1;
            const double* P = par + 1 + n;
            double q = 0.0;
            for (int j = 0; j < n; ++j) {
                double row = 0.0;
                for (int i = 0; i < n; ++i) row += (x[i] - mu[i]) * P[i * n + j];
                q += row * (x[j] - mu[j]);
            }
            return -0.5 * q;
        }
        case LHVI_POT_QUADRATIC: {
            const int n = (int)par[0];
            return quad_form(par + 1, par + 1 + n * n, par[1 + n * n + n], n, x);
        }
        case LHVI_POT_HYBRID_QUADRATIC: {
            const int Nd = (int)par[0], Nc = (int)par[1];
            int cfg = 0, ncfg = 1;
            for (int i = 0; i < Nd; ++i) { cfg = cfg * (int)par[2 + i] + (int)x[i]; ncfg *= (int)par[2 + i]; }
            const double* A = par + 2 + Nd + cfg * Nc * Nc;
            const double* b = par + 2 + Nd + ncfg * Nc * Nc + cfg * Nc;
            const double c = par[2 + Nd + ncfg * Nc * Nc + ncfg * Nc + cfg];
            return quad_form(A, b, c, Nc, x + Nd);
        }
        case LHVI_POT_LINEAR_GAUSSIAN: {
            const double d = x[1] - par[0] * x[0];
            return -(d * d) * 0.5 / par[1];
        }
        case LHVI_POT_X2: return -par[0] * (x[0] * x[0]) * 0.5 / par[1];
        case LHVI_POT_XY: return -par[0] * x[0] * x[1] * 0.5 / par[1];
        case LHVI_POT_MLN: {
            const int cq = (int)par[2];
            if (cq) return cq_log_phi(par + cq, x, idx);
            if (INTERP) return mln_formula_on(par + 3, (int)par[1], x, st) * par[0];
            break;
        }
        case LHVI_POT_MLN_HARD:
            is_log = false;
            if (INTERP) return mln_formula_on(par + 3, (int)par[1], x, st) > 0.0 ? 1.0 : 0.0;
            break;
        case LHVI_POT_IMAGE_NODE: {
            is_log = false;
            const double u = (x[0] - x[1] - par[0]) / par[1];
            return exp(-u * u * 0.5) / (2.506628274631 * par[1]);
        }
        case LHVI_POT_IMAGE_EDGE: {
            is_log = false;
            const double d = fabs(x[0] - x[1]);
            return d * par[0] + (d > par[2] ? par[3] : exp(-d / par[1]));
        }
    }
    is_log = false;
    return NAN;
}

struct MlnNoStack {        // the INTERP = false builds: nothing to hold
    __device__ __forceinline__ double get(int) const { return 0.0; }
    __device__ __forceinline__ void set(int, double) {}
};

template <bool INTERP = true>
__device__ __forceinline__ double pot_eval(int kind, const double* __restrict__ par, const double* x, const int* idx,
                                           bool& is_log) {
    if (INTERP) {
        MlnRegStack st;
        return pot_eval_on<true>(kind, par, x, idx, is_log, st);
    }
    MlnNoStack st;
    return pot_eval_on<false>(kind, par, x, idx, is_log, st);
}

// phi(x) * exp(m): one exp for the exponential-family kinds
template <bool INTERP = true>
__device__ __forceinline__ double pot_times_exp(int kind, const double* __restrict__ par, const double* x,
                                                const int* idx, double m) {
    bool is_log;
    const double v = pot_eval<INTERP>(kind, par, x, idx, is_log);
    return is_log ? exp(v + m) : v * exp(m);
}

// phi(x) itself (variational path needs log(phi + 1e-100))
__device__ __forceinline__ double pot_value(int kind, const double* __restrict__ par, const double* x, const int* idx) {
    bool is_log;
    const double v = pot_eval(kind, par, x, idx, is_log);
    return is_log ? exp(v) : v;
}

// Bivariate quadratic view of a pairwise potential: log phi(x0, x1) = a00 x0^2 + axy x0 x1 + a11 x1^2 + b0 x0 + b1 x1 + c.
// Returns false when the kind is not of that family.  For HYBRID_QUADRATIC (1 discrete + 1 continuous) the
// coefficients depend on the discrete state `d` (argument 0).
struct Quad2 { double a00, axy, a11, b0, b1, c; };

__device__ __forceinline__ bool quad2_of(int kind, const double* __restrict__ par, int d, Quad2& q) {
    switch (kind) {
        case LHVI_POT_GAUSSIAN: {
            if ((int)par[0] != 2) return false;
            const double m0 = par[1], m1 = par[2];
            const double p00 = par[3], p01 = par[4], p10 = par[5], p11 = par[6];
            const double s = 0.5 * (p01 + p10);
            q.a00 = -0.5 * p00; q.a11 = -0.5 * p11; q.axy = -s;
            q.b0 = p00 * m0 + s * m1; q.b1 = s * m0 + p11 * m1;
            q.c = -0.5 * (p00 * m0 * m0 + 2.0 * s * m0 * m1 + p11 * m1 * m1);
            return true;
        }
        case LHVI_POT_QUADRATIC: {
            if ((int)par[0] != 2) return false;
            q.a00 = par[1]; q.axy = par[2] + par[3]; q.a11 = par[4]; q.b0 = par[5]; q.b1 = par[6]; q.c = par[7];
            return true;
        }
        case LHVI_POT_LINEAR_GAUSSIAN: {
            const double h = par[0], is = 0.5 / par[1];
            q.a00 = -h * h * is; q.axy = 2.0 * h * is; q.a11 = -is; q.b0 = 0.0; q.b1 = 0.0; q.c = 0.0;
            return true;
        }
        case LHVI_POT_XY: {
            q.a00 = 0.0; q.a11 = 0.0; q.axy = -par[0] * 0.5 / par[1]; q.b0 = 0.0; q.b1 = 0.0; q.c = 0.0;
            return true;
        }
        case LHVI_POT_HYBRID_QUADRATIC: {
            if ((int)par[0] != 1 || (int)par[1] != 1) return false;
            const int nst = (int)par[2];
            q.a00 = 0.0; q.axy = 0.0; q.b0 = 0.0;
            q.a11 = par[3 + d]; q.b1 = par[3 + nst + d]; q.c = par[3 + 2 * nst + d];
            return true;
        }
    }
    return false;
}

}  // namespace lhvi
