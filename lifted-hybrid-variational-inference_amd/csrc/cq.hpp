// cq.hpp -- conditionally quadratic potentials on the f -> v half sweep (gfx950).
//
// The reference's hybrid Markov-logic formulas are polynomials of degree <= 2 in their continuous arguments for every
// assignment of their discrete ones: eq_op(a, b) = -(a - b) ** 2 (MLNPotential.py:26-27), so
// x[0] * eq_op(x[1], x[2]) (Demo/Data/HMLN/GeneratorPaperPopularity.py:28-40) and x[0] * eq_op(x[1], const)
// (GeneratorRobotMapping.py:60-75) are, per value of the boolean x[0], quadratic forms -- the family the heavy / light
// kernels of pbp.hip serve.  lhvi/expr.py::cq_block evaluates the traced formula symbolically per joint discrete state and
// appends the coefficient table behind the potential's bytecode (row header [w, ncode, cq_off]; the block sits at par + cq_off):
//     [CQ_MAGIC, arity, Nd, Nc, role[arity], dims[Nd], coef[ncfg][6]]
//     role[a] >= 0: index among the discrete arguments;  role[a] = -1 - i: the i-th continuous argument (i < Nc <= 2)
//     coef[cfg] = (a00, axy, a11, b0, b1, c):  log phi = a00 u^2 + axy u v + a11 v^2 + b0 u + b1 v + c,  (u, v) = the
//     continuous arguments in argument order, cfg = mixed-radix index of the discrete states (first one most significant)
// cq_analyze() resolves one edge of such a factor against the evidence (observed arguments are substituted, hidden ones
// become the target / the partners) and says which kernel serves it:
//     CQ_HEAVY   continuous target, at most one hidden partner, which is continuous      -> pbp_f2v_heavy_kernel
//     CQ_LIGHT1  continuous target, one hidden discrete partner with <= 2 states        -> pbp_f2v_light / _pair_kernel
//     CQ_LIGHT2  discrete target (<= 2 states), one hidden continuous partner           -> pbp_f2v_light / _pair_kernel
//     CQ_MIX     continuous target, hidden discrete AND hidden continuous partner       -> pbp_f2v_cq_kernel
//     CQ_JOINT   discrete target, two hidden continuous partners (n^2 joint particles)  -> pbp_f2v_cq_kernel
// Anything else (repeated variables in the scope -- HLBP:193-215 treats the second occurrence specially --, several hidden
// discrete partners, more than LHVI_CQ_MAX_STATES states) stays with the generic kernel.
#pragma once
#include "common.hpp"

namespace lhvi {

constexpr double CQ_MAGIC = 17233.0;
constexpr int LHVI_CQ_MAX_STATES = 4;

enum { CQ_NONE = 0, CQ_HEAVY = 1, CQ_LIGHT1 = 2, CQ_LIGHT2 = 3, CQ_MIX = 4, CQ_JOINT = 5 };

struct CqView {
    int arity, Nd, Nc;
    const double* role;
    const double* dims;
    const double* coef;
};

// the conditional-quadratic block of potential row `par` of `len` doubles (MLN kinds only), if it has one
__device__ __forceinline__ bool cq_view(int kind, const double* __restrict__ par, int len, CqView& v) {
    if (kind != LHVI_POT_MLN) return false;
    const int head = (int)par[2];                       // offset of the block in the row, 0 = none (lhvi/mln.py::device_spec)
    if (head < 3 || len < head + 4 || par[head] != CQ_MAGIC) return false;
    const double* b = par + head;
    v.arity = (int)b[1]; v.Nd = (int)b[2]; v.Nc = (int)b[3];
    v.role = b + 4;
    v.dims = v.role + v.arity;
    v.coef = v.dims + v.Nd;
    return v.Nc >= 1 && v.Nc <= 2 && v.arity <= LHVI_MAX_ARITY;
}

struct CqInfo {
    int route;                 // CQ_*
    int S;                     // coefficient sets: states of the hidden discrete partner (MIX / LIGHT1) or of the target (JOINT / LIGHT2); 1 for HEAVY
    int yv, yce, ny;           // the staged continuous partner (HEAVY / MIX / JOINT) or the particle side of LIGHT2: variable, v2f row, particles
    double yval;               // NaN when that partner is hidden; HEAVY / MIX without a hidden continuous partner: 0 (nothing left to substitute)
    int zv, zce, nz;           // MIX / LIGHT1: the hidden discrete partner; JOINT: the lane-side continuous partner
    // per set: log phi = kx x^2 + (ay y + by) y + c + (axy y + bx) x, x = target (JOINT: lane-side partner), y = staged partner;
    // LIGHT1 / LIGHT2: log phi = kx x^2 + bx x + c in the one hidden continuous argument
    double ay[LHVI_CQ_MAX_STATES], by[LHVI_CQ_MAX_STATES], c[LHVI_CQ_MAX_STATES], axy[LHVI_CQ_MAX_STATES],
           bx[LHVI_CQ_MAX_STATES], kx[LHVI_CQ_MAX_STATES];
};

__device__ __forceinline__ int cq_state_index(const lhvi_graph_t& g, int v, double x) {
    const int d = g.var_dom[v];
    for (int i = g.dom_ptr[d]; i < g.dom_ptr[d + 1]; ++i)
        if (g.dom_val[i] == x) return i - g.dom_ptr[d];
    return -1;
}

// Resolve edge e.  np[] = particle counts (lhvi_pbp_t.np), n = particle slots.  Returns info.route (CQ_NONE: generic).
__device__ inline int cq_analyze(const lhvi_graph_t& g, const lhvi_pots_t& pots, const int32_t* __restrict__ np, int n,
                                 int e, CqInfo& o) {
    o.route = CQ_NONE;
    const int f = g.edge_fac[e], base = g.fac_ptr[f], arity = g.fac_ptr[f + 1] - base, pos = e - base;
    const int pot = g.fac_pot[f];
    CqView v;
    if (!cq_view(pots.kind[pot], pots.param + pots.off[pot], pots.off[pot + 1] - pots.off[pot], v)) return CQ_NONE;
    if (v.arity != arity) return CQ_NONE;
    const int tv = g.edge_var[e];
    // scope walk: no variable twice; classify every argument
    int cvar[2] = {-1, -1}, cedge[2] = {-1, -1};
    bool chid[2] = {false, false};
    double cval[2] = {0.0, 0.0};
    int tcont = -1;                       // continuous index of the target, or -1 when it is discrete
    int cfg = 0;                          // mixed-radix index with the target's / the hidden partner's digit left at 0
    int stride_t = 0, stride_z = 0, zv = -1, zce = -1, nz = 0;
    for (int a = 0; a < arity; ++a) {
        const int va = g.edge_var[base + a];
        if (canon(g.edge_canon, base + a) != base + a) return CQ_NONE;
        for (int b = 0; b < a; ++b)
            if (g.edge_var[base + b] == va) return CQ_NONE;
        const double val = g.var_value[va];
        const int role = (int)v.role[a];
        if (role < 0) {
            const int ci = -1 - role;
            cvar[ci] = va; cedge[ci] = base + a; chid[ci] = is_hidden(val); cval[ci] = val;
            if (!g.dom_cont[g.var_dom[va]]) return CQ_NONE;
            if (a == pos) tcont = ci;
        } else {
            if (g.dom_cont[g.var_dom[va]]) return CQ_NONE;
            int stride = 1;
            for (int k = role + 1; k < v.Nd; ++k) stride *= (int)v.dims[k];
            const int nst = g.dom_ptr[g.var_dom[va] + 1] - g.dom_ptr[g.var_dom[va]];
            if (nst != (int)v.dims[role]) return CQ_NONE;
            if (a == pos) { stride_t = stride; }
            else if (is_hidden(val)) {
                if (zv >= 0) return CQ_NONE;                 // two hidden discrete partners: generic
                zv = va; zce = base + a; nz = nst; stride_z = stride;
            } else {
                const int st = cq_state_index(g, va, val);
                if (st < 0) return CQ_NONE;
                cfg += st * stride;
            }
        }
    }
    if (!is_hidden(g.var_value[tv])) return CQ_NONE;
    auto raw = [&](int c, double (&r)[6]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = v.coef[6 * c + k];
    };
    o.yv = 0; o.yce = 0; o.ny = 1; o.yval = 0.0; o.zv = 0; o.zce = -1; o.nz = 0;
    if (tcont >= 0) {
        // ---- continuous target ------------------------------------------------------------------------------------
        const int co = 1 - tcont;
        const bool other = v.Nc == 2;
        const bool yhid = other && chid[co];
        o.S = zv >= 0 ? nz : 1;
        if (o.S > LHVI_CQ_MAX_STATES) return CQ_NONE;
        for (int s = 0; s < o.S; ++s) {
            double r[6];
            raw(cfg + s * stride_z, r);
            double kx, bx, ay, by;
            if (tcont == 0) { kx = r[0]; bx = r[3]; ay = r[2]; by = r[4]; } else { kx = r[2]; bx = r[4]; ay = r[0]; by = r[3]; }
            double axy = r[1], c = r[5];
            if (other && !yhid) {                            // observed continuous partner: substitute its value
                const double y = cval[co];
                c += (ay * y + by) * y; bx += axy * y; ay = 0.0; by = 0.0; axy = 0.0;
            }
            o.kx[s] = kx; o.bx[s] = bx; o.ay[s] = ay; o.by[s] = by; o.axy[s] = axy; o.c[s] = c;
        }
        if (yhid) { o.yv = cvar[co]; o.yce = cedge[co]; o.ny = np[cvar[co]]; o.yval = __builtin_nan(""); }
        if (o.ny > 64) return CQ_NONE;
        if (zv < 0) { o.route = CQ_HEAVY; return o.route; }
        o.zv = zv; o.zce = zce; o.nz = nz;
        if (!yhid && nz <= 2) { o.route = CQ_LIGHT1; return o.route; }
        o.route = CQ_MIX;
        return o.route;
    }
    // ---- discrete target ----------------------------------------------------------------------------------------------
    if (zv >= 0) return CQ_NONE;                             // a hidden discrete partner besides a discrete target: generic
    const int ns = g.dom_ptr[g.var_dom[tv] + 1] - g.dom_ptr[g.var_dom[tv]];
    if (ns > LHVI_CQ_MAX_STATES) return CQ_NONE;
    o.S = ns;
    const int nhid = (chid[0] ? 1 : 0) + ((v.Nc == 2 && chid[1]) ? 1 : 0);
    if (nhid == 0) return CQ_NONE;                           // a table in disguise: a handful of terms, generic
    for (int s = 0; s < ns; ++s) {
        double r[6];
        raw(cfg + s * stride_t, r);
        if (nhid == 2) {                                     // lane side x = first continuous argument, staged y = second
            o.kx[s] = r[0]; o.bx[s] = r[3]; o.ay[s] = r[2]; o.by[s] = r[4]; o.axy[s] = r[1]; o.c[s] = r[5];
        } else if (chid[0]) {                                // one hidden continuous argument: fold the other (if any)
            const double y = v.Nc == 2 ? cval[1] : 0.0;
            o.kx[s] = r[0]; o.bx[s] = r[3] + r[1] * y; o.c[s] = r[5] + (r[2] * y + r[4]) * y;
            o.ay[s] = 0.0; o.by[s] = 0.0; o.axy[s] = 0.0;
        } else {
            const double y = cval[0];
            o.kx[s] = r[2]; o.bx[s] = r[4] + r[1] * y; o.c[s] = r[5] + (r[0] * y + r[3]) * y;
            o.ay[s] = 0.0; o.by[s] = 0.0; o.axy[s] = 0.0;
        }
    }
    if (nhid == 2) {
        o.zv = cvar[0]; o.zce = cedge[0]; o.nz = np[cvar[0]];
        o.yv = cvar[1]; o.yce = cedge[1]; o.ny = np[cvar[1]]; o.yval = __builtin_nan("");
        if (o.ny > 64 || o.nz > 64) return CQ_NONE;
        o.route = CQ_JOINT;
        return o.route;
    }
    const int h = chid[0] ? 0 : 1;
    o.yv = cvar[h]; o.yce = cedge[h]; o.ny = np[cvar[h]]; o.yval = __builtin_nan("");
    if (ns > 2 || o.ny > 64) return CQ_NONE;
    o.route = CQ_LIGHT2;
    return o.route;
}

}  // namespace lhvi
