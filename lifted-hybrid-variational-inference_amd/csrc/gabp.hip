// gabp.hip -- Gaussian belief propagation sweep (ground GaBP and lifted GaLBP) for gfx950.
//
// Reference semantics: GaBP.py:20-169,187-200 and GaLBP.py:21-217 (restated in SURVEY.md Appendix A.1/A.2).
// Messages are (mu, var) pairs of fp64; var = NaN encodes the reference's `None` (pure linear term).
// The schedule is flooding: each half sweep reads only the other half's buffer, so both kernels are
// embarrassingly parallel and bound by HBM bandwidth (76 B/edge/sweep algorithmic).
//
// Floating point: contraction is disabled in this file so a*b+c is rounded twice like CPython does;
// sums run in rv.nb order.  That makes the ground sweep bit-identical to the reference on the fixtures.
#include "common.hpp"

#pragma clang fp contract(off)

namespace lhvi {

__global__ void __launch_bounds__(BLOCK) gabp_init_kernel(int64_t n2, double* __restrict__ a, double* __restrict__ b) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n2) { st2(a, i, 0.0, 1.0); st2(b, i, 0.0, 1.0); }
}

// One thread per variable-CSR slot k = (variable v, incident edge e).  The d threads of a degree-d
// variable sit in adjacent lanes and walk the same d incoming messages, so after the first touch the
// reads are same-address broadcasts out of L1.  Leave-one-out is a direct sum (no total-minus-own
// cancellation), in rv.nb order like GaBP.py:23-29 / GaLBP.py:24-34.
// The Gaussian sweep keeps the reference's summation order (bit-identical messages on the fixtures) for every variable
// with up to GABP_HUB_DEGREE incident factors and only switches to wave-parallel sums beyond, where the direct
// leave-one-out sum is quadratic in the degree.
constexpr int GABP_HUB_DEGREE = 512;
constexpr int GABP_BATCH = 4;            // messages in flight per thread while it walks a variable's row

__global__ void __launch_bounds__(BLOCK) gabp_v2f_kernel(lhvi_graph_t g, const double* __restrict__ f2v,
                                                        double* __restrict__ v2f) {
    int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= g.nnz) return;
    const int e = g.var_edge[k];
    const int v = g.slot_var ? g.slot_var[k] : g.edge_var[e];      // contiguous copy instead of a gather through e
    if (!is_hidden(g.var_value[v])) {          // observed rv sends nothing (returns None)
        st2(v2f, e, NAN, NAN);
        return;
    }
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (g.hub_vars && hi - lo > GABP_HUB_DEGREE) return;   // hubs: gabp_v2f_hub_kernel (the direct sum is O(deg^2))
    double H = 0.0, P = 0.0;
    // GABP_BATCH messages are fetched before any of them is used (the loads are independent, the sums are not: they stay in
    // rv.nb order); a long row is a chain of L2 round trips otherwise
    for (int j0 = lo; j0 < hi; j0 += GABP_BATCH) {
        int ej[GABP_BATCH];
        double2 mm[GABP_BATCH];
        double cc[GABP_BATCH];
#pragma unroll
        for (int i = 0; i < GABP_BATCH; ++i) {
            const int j = j0 + i < hi ? j0 + i : hi - 1;
            ej[i] = g.var_edge[j];
        }
#pragma unroll
        for (int i = 0; i < GABP_BATCH; ++i) {
            mm[i] = ld2(f2v, ej[i]);
            cc[i] = g.edge_count ? g.edge_count[ej[i]] : 1.0;
        }
#pragma unroll
        for (int i = 0; i < GABP_BATCH; ++i) {
            const int j = j0 + i;
            if (j >= hi) break;
            double c = cc[i];
            if (g.edge_count) {                 // lifted: own factor enters count-1 times
                if (j == k) c -= 1.0;
            } else if (j == k) continue;        // ground: own factor skipped
            const double2 m = mm[i];
            if (m.y != m.y) {                   // var is None: linear term only
                H -= g.edge_count ? m.x * c : m.x;
            } else {
                const double p = 1.0 / m.y;
                if (g.edge_count) { H += p * m.x * c; P += p * c; }
                else              { H += p * m.x;     P += p; }
            }
        }
    }
    const double var = 1.0 / P;
    st2(v2f, e, var * H, var);
}

// Variables with more than GABP_HUB_DEGREE incident factors (the template variables of a relational model): one wavefront
// per variable forms the information-form total once and every slot subtracts one copy of its own factor's term
// ("total minus own": O(deg) instead of O(deg^2); costs a relative error of ~deg * 1e-16 in the hub's messages, which
// the direct sum of the low-degree path avoids).
__global__ void __launch_bounds__(BLOCK) gabp_v2f_hub_kernel(lhvi_graph_t g, const double* __restrict__ f2v,
                                                            double* __restrict__ v2f) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= g.n_hubs) return;
    const int v = g.hub_vars[i];
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= GABP_HUB_DEGREE) return;
    if (!is_hidden(g.var_value[v])) {
        for (int k = lo + lane; k < hi; k += 64) st2(v2f, g.var_edge[k], NAN, NAN);
        return;
    }
    double H = 0.0, P = 0.0;
    for (int k = lo + lane; k < hi; k += 64) {
        const int e = g.var_edge[k];
        const double c = g.edge_count ? g.edge_count[e] : 1.0;
        const double2 m = ld2(f2v, e);
        if (m.y != m.y) H -= m.x * c;
        else { const double p = 1.0 / m.y; H += p * m.x * c; P += p * c; }
    }
    H = dpp_wave_reduce(H, SumOp()); P = dpp_wave_reduce(P, SumOp());
    for (int k = lo + lane; k < hi; k += 64) {
        const int e = g.var_edge[k];
        const double2 m = ld2(f2v, e);
        double h = H, p = P;
        if (m.y != m.y) h += m.x;                                  // take one copy of the own term back out
        else { const double q = 1.0 / m.y; h -= q * m.x; p -= q; }
        const double var = 1.0 / p;
        st2(v2f, e, var * h, var);
    }
}

// closed forms of GaBP.message_f_to_rv (GaBP.py:37-138).  (u, s): partner's v->f message when the
// partner is hidden; y: its evidence value otherwise.
__device__ __forceinline__ double2 f2v_closed_form(int kind, const double* __restrict__ par, int arity, int pos,
                                                   bool partner_hidden, double u, double s, double y) {
    const double INF = __builtin_huge_val();
    if (kind == LHVI_POT_X2) {
        const double h = par[0], sg = par[1];
        if (h == 0.0) return make_double2(0.0, INF);
        return make_double2(0.0, sg / h);
    }
    if (arity != 2) return make_double2(0.0, INF);
    if (kind == LHVI_POT_GAUSSIAN) {
        if ((int)par[0] != 2) return make_double2(0.0, INF);
        const double* mu = par + 1;
        const double* a = par + 1 + 2 + 4;       // sig ** -1 as computed by np.matrix (GaBP.py:44)
        double a1, a2, a3, u1, u2;
        if (pos == 1) { a1 = a[0]; a2 = a[1]; a3 = a[3]; u1 = mu[0]; u2 = mu[1]; }
        else          { a1 = a[3]; a2 = a[1]; a3 = a[0]; u1 = mu[1]; u2 = mu[0]; }
        if (partner_hidden) {
            const double a4 = 1.0 / s;
            const double temp = a3 * (a4 + a1) - a2 * a2;
            const double m = a2 * a4 * (u1 - u) / temp + u2;
            const double v = 1.0 / (a3 - a2 * a2 / (a4 + a1));
            return make_double2(m, v);
        }
        return make_double2(-u2 - a2 * (y - u1) / a3, 1.0 / a3);
    }
    if (kind == LHVI_POT_LINEAR_GAUSSIAN) {
        const double h = par[0], s1 = par[1];
        if (h == 0.0) return make_double2(0.0, INF);
        const double h2 = h * h;
        if (partner_hidden) {
            if (pos == 0) return make_double2(u / h, (s1 + s) / h2);
            return make_double2(u * h, s1 + s * h2);
        }
        if (pos == 0) return make_double2(y / h, s1 / h2);
        return make_double2(h * y, s1);
    }
    if (kind == LHVI_POT_XY) {
        const double h = par[0], s1 = par[1];
        if (h == 0.0) return make_double2(0.0, INF);
        if (partner_hidden) {
            const double m = 2.0 * s1 * u / (h * s);
            const double v = -4.0 * (s1 * s1) / (h * h * s);
            return make_double2(m, v);
        }
        return make_double2(h * y / (2.0 * s1), NAN);   // (mu, None)
    }
    return make_double2(0.0, INF);
}

// One thread per edge (factor-major, so a factor's two edges are adjacent lanes and the partner's
// message is a neighbouring 16-byte load).
__global__ void __launch_bounds__(BLOCK) gabp_f2v_kernel(lhvi_graph_t g, lhvi_pots_t pots,
                                                        const double* __restrict__ v2f, double* __restrict__ f2v) {
    int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= g.E) return;
    if (canon(g.edge_canon, e) != e) return;     // alias of a repeated cluster: the canonical edge owns the message
    // message to an observed rv is never produced (GaBP.py:39-40); edge_value = the variable's value, per edge
    if (!is_hidden(g.edge_value ? g.edge_value[e] : g.var_value[g.edge_var[e]])) return;
    const int f = g.edge_fac[e];
    const int base = g.fac_ptr[f];
    const int arity = g.fac_ptr[f + 1] - base;
    const int pos = e - base;
    const int pot = g.fac_pot[f];
    const int kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    bool partner_hidden = false;
    double u = 0.0, s = 0.0, y = 0.0;
    if (arity == 2) {
        const int pe = base + (1 - pos);
        y = g.edge_value ? g.edge_value[pe] : g.var_value[g.edge_var[pe]];
        partner_hidden = is_hidden(y);
        if (partner_hidden) {
            const double2 m = ld2(v2f, canon(g.edge_canon, pe));
            u = m.x; s = m.y;
        }
    }
    const double2 out = f2v_closed_form(kind, par, arity, pos, partner_hidden, u, s, y);
    st2(f2v, e, out.x, out.y);
}

// Per-variable product of all incoming messages (GaBP.py:187-200, GaLBP.py:201-217).
__global__ void __launch_bounds__(BLOCK) gabp_marginal_kernel(lhvi_graph_t g, const double* __restrict__ f2v,
                                                             double* __restrict__ out) {
    int v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= g.V) return;
    const double val = g.var_value[v];
    if (!is_hidden(val)) { st2(out, v, val, 0.0); return; }
    if (g.hub_vars && g.var_ptr[v + 1] - g.var_ptr[v] > GABP_HUB_DEGREE) return;      // gabp_marginal_hub_kernel
    double H = 0.0, P = 0.0;
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (g.edge_count) {
        for (int j = lo; j < hi; ++j) {
            const int ej = g.var_edge[j];
            const double2 m = ld2(f2v, ej);
            const double c = g.edge_count[ej];
            if (m.y != m.y) H -= m.x * c;
            else { const double p = 1.0 / m.y; H += p * m.x * c; P += p * c; }
        }
    } else {
        // ground graph: four gathers (and their reciprocals) in flight per step, the additions in rv.nb order as before -- a
        // thread of a 100-entry row used to pay one memory round trip and one division per entry, one after the other
        int j = lo;
        for (; j + 4 <= hi; j += 4) {
            const double2 m0 = ld2(f2v, g.var_edge[j]), m1 = ld2(f2v, g.var_edge[j + 1]), m2 = ld2(f2v, g.var_edge[j + 2]),
                          m3 = ld2(f2v, g.var_edge[j + 3]);
            const double p0 = 1.0 / m0.y, p1 = 1.0 / m1.y, p2 = 1.0 / m2.y, p3 = 1.0 / m3.y;
#define LHVI_MARG_STEP(m, p) { const bool none = (m).y != (m).y; const double h = H + (none ? -(m).x : (p) * (m).x), q = P + (p); \
                               H = h; P = none ? P : q; }
            LHVI_MARG_STEP(m0, p0) LHVI_MARG_STEP(m1, p1) LHVI_MARG_STEP(m2, p2) LHVI_MARG_STEP(m3, p3)
        }
        for (; j < hi; ++j) { const double2 m = ld2(f2v, g.var_edge[j]); const double p = 1.0 / m.y; LHVI_MARG_STEP(m, p) }
#undef LHVI_MARG_STEP
    }
    const double var = 1.0 / P;
    st2(out, v, var * H, var);
}

__global__ void __launch_bounds__(BLOCK) gabp_marginal_hub_kernel(lhvi_graph_t g, const double* __restrict__ f2v,
                                                                 double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= g.n_hubs) return;
    const int v = g.hub_vars[i];
    if (g.var_ptr[v + 1] - g.var_ptr[v] <= GABP_HUB_DEGREE) return;
    if (!is_hidden(g.var_value[v])) return;                 // written by the thread-per-variable kernel
    double H = 0.0, P = 0.0;
    for (int j = g.var_ptr[v] + lane; j < g.var_ptr[v + 1]; j += 64) {
        const int ej = g.var_edge[j];
        const double c = g.edge_count ? g.edge_count[ej] : 1.0;
        const double2 m = ld2(f2v, ej);
        if (m.y != m.y) H -= m.x * c;
        else { const double p = 1.0 / m.y; H += p * m.x * c; P += p * c; }
    }
    H = dpp_wave_reduce(H, SumOp()); P = dpp_wave_reduce(P, SumOp());
    const double var = 1.0 / P;
    if (lane == 0) st2(out, v, var * H, var);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pull form of the sweep: ONE kernel per iteration, messages kept in variable-CSR ("slot") order.
// For pairwise / unary factors the f -> v message of an edge is a closed form of ONE v -> f message (the partner's), so
// it never has to exist in memory: slot k's thread evaluates the closed form for its own incoming edge (one 16-byte gather
// of the partner's previous v -> f message, the only random access of the sweep), parks it in LDS, and after a barrier
// every slot of the variable sums the row out of LDS -- in rv.nb order, leave-one-out as a direct sum, the same expressions
// as gabp_f2v_kernel + gabp_v2f_kernel, hence the same bits.  A row that crosses the block's slot range recomputes the
// outside entries.  Per slot and sweep: 16 B gathered + 16 B stored in order + 12 B of plan and CSR instead of two random 16-byte
// accesses, two in-order ones and two launches.
struct PullPlan { const int32_t* pslot; const int32_t* info; const double* count; };

__device__ __forceinline__ double2 pull_incoming(const lhvi_graph_t& g, const lhvi_pots_t& pots, const PullPlan& pl,
                                                 const double* __restrict__ vprev, int j) {
    const int info = pl.info[j];
    const int code = info & 3, pot = info >> 2;
    const int kind = pots.kind[pot];
    const double* par = pots.param + pots.off[pot];
    const int arity = code == 0 ? 1 : (code == 3 ? 3 : 2), pos = code == 2 ? 1 : 0;
    bool partner_hidden = false;
    double u = 0.0, sv = 0.0, y = 0.0;
    if (arity == 2) {
        const int ps = pl.pslot[j];
        partner_hidden = ps >= 0;
        if (partner_hidden) { const double2 m = ld2(vprev, ps); u = m.x; sv = m.y; }
        else y = g.var_value[-1 - ps];              // observed partner: pslot holds -1 - (its variable)
    }
    return f2v_closed_form(kind, par, arity, pos, partner_hidden, u, sv, y);
}

__global__ void __launch_bounds__(BLOCK) gabp_pull_kernel(lhvi_graph_t g, lhvi_pots_t pots, PullPlan pl,
                                                         const double* __restrict__ vprev, double* __restrict__ vnext, int first) {
    // LDS holds the f -> v messages of every slot of every (non-hub) row that touches this block's slot range: the rows
    // straddling the block's ends are staged whole, so the sums below never leave LDS
    constexpr int CAP = BLOCK + 2 * GABP_HUB_DEGREE;
    __shared__ double2 sh[CAP];
    const int k0 = blockIdx.x * BLOCK;
    const int kend = min(k0 + BLOCK, g.nnz);
    int lo_ext = k0, hi_ext = kend;
    {
        const int vf = g.slot_var[k0], vl = g.slot_var[kend - 1];
        const int a = g.var_ptr[vf], b = g.var_ptr[vl + 1];
        if (g.var_ptr[vf + 1] - a <= GABP_HUB_DEGREE) lo_ext = a;         // (a hub row is served by gabp_pull_hub_kernel)
        if (b - g.var_ptr[vl] <= GABP_HUB_DEGREE) hi_ext = b;
    }
    for (int j = lo_ext + threadIdx.x; j < hi_ext; j += BLOCK) {
        // first sweep: every f -> v message still is its initial value (0, 1) (GaBP.py:143-150)
        const int vj = g.slot_var[j];
        // (slots of a hub row are never read from LDS: gabp_pull_hub_kernel serves the whole row)
        if (g.n_hubs > 0 && g.var_ptr[vj + 1] - g.var_ptr[vj] > GABP_HUB_DEGREE) continue;
        const bool hid = is_hidden(g.var_value[vj]);
        const double2 m = (hid && !first) ? pull_incoming(g, pots, pl, vprev, j) : make_double2(0.0, 1.0);
        // staged in information form (p * mu, p), p = 1 / var -- the products every slot of the row would form anyway, so
        // the row sums below are additions only; a `None` variance (NaN) keeps (mu, NaN)
        const double p = 1.0 / m.y;
        sh[j - lo_ext] = (m.y != m.y) ? m : make_double2(p * m.x, p);
    }
    __syncthreads();
    const int k = k0 + threadIdx.x;
    if (k >= g.nnz) return;
    const int v = g.slot_var[k];
    if (!is_hidden(g.var_value[v])) { st2(vnext, k, NAN, NAN); return; }
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo > GABP_HUB_DEGREE) return;                      // gabp_pull_hub_kernel
    double H = 0.0, P = 0.0;
    for (int j = lo; j < hi; ++j) {
        double c = pl.count ? pl.count[j] : 1.0;
        if (pl.count) { if (j == k) c -= 1.0; }
        else if (j == k) continue;
        const double2 m = sh[j - lo_ext];
        if (m.y != m.y) H -= pl.count ? m.x * c : m.x;
        else if (pl.count) { H += m.x * c; P += m.y * c; }
        else               { H += m.x;     P += m.y; }
    }
    const double var = 1.0 / P;
    st2(vnext, k, var * H, var);
}

// Round 4: the same sweep with the per-slot dependent loads folded away.  The kernel above runs at the same speed whether the partner
// gather is random or sequential (measured: 0.295 vs 0.277 ms at 10 M slots, profiles/r04_experiments.md): it is bound by its
// chains of dependent loads -- slot_var -> var_value / var_ptr, info -> pots.kind -> pots.off -> parameters -- not by bytes.  Here
// one 16-byte record per slot (lhvi_gabp_plan_t.rec: partner slot, 4 * potential + position code, position in the row | row
// length << 10 | hidden << 20 | hub row << 21) carries everything the slot needs to know about the graph, and the potential
// parameters sit in LDS (up to GABP_LDS_POTS potentials; more: read from global memory as before).  Chain: record -> partner's
// message -> arithmetic.  Same expressions in the same order: the same bits.
constexpr int GABP_LDS_POTS = 32;
constexpr int GABP_POT_WORDS = 12;            // par[0 .. 10] (the Gaussian closed form reads up to par[10]) + the kind
#ifndef LHVI_GABP_ROW_DIRECT
#define LHVI_GABP_ROW_DIRECT 32
#endif
#ifndef LHVI_GABP_ROW_CHUNK
#define LHVI_GABP_ROW_CHUNK 8
#endif
constexpr int GABP_ROW_DIRECT = LHVI_GABP_ROW_DIRECT;   // rows up to this length are summed entry by entry in the reference's order
constexpr int GABP_ROW_CHUNK = LHVI_GABP_ROW_CHUNK;     // longer rows (ground graphs): sums of eight-entry chunks as intermediate results

template <bool LDS_POTS>
__device__ __forceinline__ double2 pull_incoming_rec(const lhvi_graph_t& g, const lhvi_pots_t& pots, const int4 r,
                                                     const double* __restrict__ vprev, const double* __restrict__ sh_par /* LDS, or the plan's pot_words */) {
    const int code = r.y & 3, pot = r.y >> 2;
    // the potential as twelve words -- par[0 .. 10] and the kind -- in LDS, or in the plan's table in global memory (one load
    // behind the record instead of pots.kind -> pots.off -> parameters)
    const double* par = sh_par + (int64_t)pot * GABP_POT_WORDS;
    const int kind = (int)par[GABP_POT_WORDS - 1];
    const int arity = code == 0 ? 1 : (code == 3 ? 3 : 2), pos = code == 2 ? 1 : 0;
    bool partner_hidden = false;
    double u = 0.0, sv = 0.0, y = 0.0;
    if (arity == 2) {
        partner_hidden = r.x >= 0;
        if (partner_hidden) { const double2 m = ld2(vprev, r.x); u = m.x; sv = m.y; }
        else y = g.var_value[-1 - r.x];
    }
    return f2v_closed_form(kind, par, arity, pos, partner_hidden, u, sv, y);
}

#ifndef LHVI_GABP_PULL_WAVES
#define LHVI_GABP_PULL_WAVES 8         // 64 registers: 0.557 / 0.514 / 0.472 ms on the Kalman-filter graph at 4 / 6 / 8 waves per SIMD (scripts/diag/pull_regs.sh)
#endif
template <bool LDS_POTS>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LHVI_GABP_PULL_WAVES, 8))) gabp_pull_rec_kernel(lhvi_graph_t g, lhvi_pots_t pots, const int4* __restrict__ rec,
                                                             const int2* __restrict__ seg, const double* __restrict__ pot_words,
                                                             const double* __restrict__ count, const double* __restrict__ vprev,
                                                             double* __restrict__ vnext, int first) {
    // a workgroup serves one SEGMENT of the slot order (lhvi_gabp_plan_t.seg): the rows that start inside one window of BLOCK
    // slots, whole -- at most BLOCK - 1 + GABP_HUB_DEGREE slots, no row cut, no hub row inside -- so nothing is staged twice
    // (workgroups of fixed 256-slot ranges staged the rows straddling their ends whole: 1.5x the gathers and closed forms on a
    // graph with rows of 66 entries)
    constexpr int CAP = BLOCK + GABP_HUB_DEGREE;
    constexpr int IT = CAP / BLOCK;                         // slots of a segment a thread serves (the same ones in every pass)
    __shared__ double2 sh[CAP];
    __shared__ double2 shc[CAP / GABP_ROW_CHUNK + 1];       // sums of eight-entry chunks of the long rows (ground graphs)
    __shared__ double2 sht[CAP / (GABP_ROW_DIRECT + 1) + 1]; // their totals: long rows start more than GABP_ROW_DIRECT slots apart
    __shared__ double sh_par[LDS_POTS ? GABP_LDS_POTS * GABP_POT_WORDS : 1];
    const int2 sg = seg[blockIdx.x];
    const int lo_ext = sg.x, hi_ext = sg.y;
    // a segment longer than the LDS stage (a plan not built by the rule of lhvi_gabp_plan_t.seg) is left unswept rather than
    // written past the stage; the whole workgroup takes this exit, ahead of every barrier
    if (hi_ext - lo_ext > CAP || hi_ext < lo_ext || lo_ext < 0 || hi_ext > g.nnz) return;
    // Round 5: a thread's slot records are fetched together, ahead of everything else, and kept in registers for all passes (each
    // pass used to read its record again -- a dependent L2 round trip in front of every pass, three more of them on a graph whose
    // rows are long), and the partners' messages of all its slots are in flight together (the second slot's gather used to wait
    // for the first slot's closed form).  Same expressions in the same order: the same bits.
    int4 r[IT];
    bool in[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int j = lo_ext + threadIdx.x + it * BLOCK;
        in[it] = j < hi_ext;
        r[it] = rec[in[it] ? j : lo_ext];
    }
    if (LDS_POTS) {
        for (int i = threadIdx.x; i < pots.P * GABP_POT_WORDS; i += BLOCK) sh_par[i] = pot_words[i];
        __syncthreads();
    }
    const double* __restrict__ pw = LDS_POTS ? sh_par : pot_words;
    double2 pm[IT];                                         // the partner's previous v -> f message (a hidden partner of a pairwise factor)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const bool want = in[it] && !first && !((r[it].z >> 21) & 1) && ((r[it].z >> 20) & 1) && r[it].x >= 0 && (r[it].y & 3) != 0 && (r[it].y & 3) != 3;
        pm[it] = make_double2(0.0, 0.0);
        if (want) pm[it] = ld2(vprev, r[it].x);
    }
    int long_rows = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (!in[it] || ((r[it].z >> 21) & 1)) continue;
        long_rows |= ((r[it].z >> 10) & 1023) > GABP_ROW_DIRECT;
        const bool hid = (r[it].z >> 20) & 1;
        double2 m = make_double2(0.0, 1.0);
        if (hid && !first) {
            const int code = r[it].y & 3, pot = r[it].y >> 2;
            const double* par = pw + (int64_t)pot * GABP_POT_WORDS;
            const int kind = (int)par[GABP_POT_WORDS - 1];
            const int arity = code == 0 ? 1 : (code == 3 ? 3 : 2), pos = code == 2 ? 1 : 0;
            const bool partner_hidden = arity == 2 && r[it].x >= 0;
            const double y = (arity == 2 && r[it].x < 0) ? g.var_value[-1 - r[it].x] : 0.0;
            m = f2v_closed_form(kind, par, arity, pos, partner_hidden, partner_hidden ? pm[it].x : 0.0, partner_hidden ? pm[it].y : 0.0, y);
        }
        const double p = 1.0 / m.y;
        // staged as the entry's CONTRIBUTION (h, p) to the row sums -- (p mu, p), or (-mu, 0) for a `None` variance (GaBP.py:27-33: a
        // linear term only) -- so that the sums below are additions (times the count on a lifted graph) without a case distinction
        sh[threadIdx.x + it * BLOCK] = (m.y != m.y) ? make_double2(-m.x, 0.0) : make_double2(p * m.x, p);
    }
    long_rows = __syncthreads_or(long_rows);                    // (the staging barrier; most blocks of most graphs have no long row)
    if (!count && long_rows) {
        // rows of more than GABP_ROW_DIRECT entries: one thread per chunk of eight entries (the last chunk takes the remainder)
        // adds its chunk up in rv.nb order.  Chunk starts lie at least eight apart, so (start >> 3) names a chunk uniquely.
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int rz = r[it].z;
            const int len = (rz >> 10) & 1023, pos = rz & 1023;
            if (!in[it] || ((rz >> 21) & 1) || len <= GABP_ROW_DIRECT || (pos % GABP_ROW_CHUNK) != 0) continue;
            const int nc = len / GABP_ROW_CHUNK, q = pos / GABP_ROW_CHUNK;
            if (q >= nc) continue;
            const int end = q == nc - 1 ? len - pos : GABP_ROW_CHUNK;
            const double2* __restrict__ e = sh + (threadIdx.x + it * BLOCK);
            // the first eight entries are read together (every chunk has them), the last chunk's remainder behind them
            double2 c[GABP_ROW_CHUNK];
#pragma unroll
            for (int i = 0; i < GABP_ROW_CHUNK; ++i) c[i] = e[i];
            double H = 0.0, P = 0.0;
#pragma unroll
            for (int i = 0; i < GABP_ROW_CHUNK; ++i) { H += c[i].x; P += c[i].y; }
            for (int i = GABP_ROW_CHUNK; i < end; ++i) { H += e[i].x; P += e[i].y; }
            shc[(threadIdx.x + it * BLOCK) / GABP_ROW_CHUNK] = make_double2(H, P);
        }
        __syncthreads();
        // ... and one thread per long row adds its chunks up: a slot's leave-one-out sum is then (total - its own chunk) + the
        // other entries of that chunk -- 1 + 8 LDS reads whatever the row's length (it used to walk all the row's chunks: a
        // Kalman-filter row of 66 entries cost every one of its slots 16 reads)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int rz = r[it].z;
            const int len = (rz >> 10) & 1023;
            if (!in[it] || ((rz >> 21) & 1) || len <= GABP_ROW_DIRECT || (rz & 1023) != 0) continue;
            const double2* __restrict__ cs = shc + (threadIdx.x + it * BLOCK) / GABP_ROW_CHUNK;
            const int nc = len / GABP_ROW_CHUNK;
            double H = 0.0, P = 0.0;
            int c = 0;
            for (; c + 4 <= nc; c += 4) {                       // four reads in flight, added in the chunks' order
                const double2 c0 = cs[c], c1 = cs[c + 1], c2 = cs[c + 2], c3 = cs[c + 3];
                H += c0.x; P += c0.y; H += c1.x; P += c1.y; H += c2.x; P += c2.y; H += c3.x; P += c3.y;
            }
            for (; c < nc; ++c) { H += cs[c].x; P += cs[c].y; }
            sht[(threadIdx.x + it * BLOCK) / (GABP_ROW_DIRECT + 1)] = make_double2(H, P);
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
    if (!in[it]) continue;
    const int k = lo_ext + threadIdx.x + it * BLOCK;
    const int rz = r[it].z;
    if (!((rz >> 20) & 1)) { st2(vnext, k, NAN, NAN); continue; }
    const int lo = k - (rz & 1023), hi = lo + ((rz >> 10) & 1023);
    double H = 0.0, P = 0.0;
    if (count) {
        // lifted graph (GaLBP.py:24-34): every entry times rv.count[f], the slot's own factor with count - 1; four entries in flight
        const double2* __restrict__ row = sh + (lo - lo_ext);
        const double* __restrict__ cnt = count + lo;
        const int n = hi - lo, own = k - lo;
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            const double2 m0 = row[j], m1 = row[j + 1], m2 = row[j + 2], m3 = row[j + 3];
            const double c0 = cnt[j], c1 = cnt[j + 1], c2 = cnt[j + 2], c3 = cnt[j + 3];
#define LHVI_CNT_STEP(m, c, jj) { const double cc = (jj) == own ? (c) - 1.0 : (c); H += (m).x * cc; P += (m).y * cc; }
            LHVI_CNT_STEP(m0, c0, j) LHVI_CNT_STEP(m1, c1, j + 1) LHVI_CNT_STEP(m2, c2, j + 2) LHVI_CNT_STEP(m3, c3, j + 3)
        }
        for (; j < n; ++j) { const double2 m = row[j]; const double c = cnt[j]; LHVI_CNT_STEP(m, c, j) }
#undef LHVI_CNT_STEP
    } else {
        const double2* __restrict__ row = sh + (lo - lo_ext);
        const int n = hi - lo, own = k - lo;
        if (n <= GABP_ROW_DIRECT) {
            // the row in rv.nb order, the slot's own entry left out (GaBP.py:23-33): the reference's additions in its order,
            // four LDS reads in flight (a thread of a long row used to wait out one LDS round trip per entry)
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                const double2 m0 = row[j], m1 = row[j + 1], m2 = row[j + 2], m3 = row[j + 3];
#define LHVI_ROW_STEP(m, jj) { const double h = H + (m).x, q = P + (m).y; const bool take = (jj) != own; H = take ? h : H; P = take ? q : P; }
                LHVI_ROW_STEP(m0, j) LHVI_ROW_STEP(m1, j + 1) LHVI_ROW_STEP(m2, j + 2) LHVI_ROW_STEP(m3, j + 3)
            }
            for (; j < n; ++j) { const double2 m = row[j]; LHVI_ROW_STEP(m, j) }
        } else {
            // a long row: the row's total less the slot's own chunk, plus that chunk's other entries (the chunks' sums and the
            // total are intermediate results: O(8) additions per slot instead of O(len); differs from the reference's
            // left-to-right sum by rounding only: asserted at 1e-13 on the fixtures)
            const int nc = n / GABP_ROW_CHUNK, q = min(own / GABP_ROW_CHUNK, nc - 1);
            const double2 tot = sht[(lo - lo_ext) / (GABP_ROW_DIRECT + 1)], mine = shc[(lo - lo_ext) / GABP_ROW_CHUNK + q];
            H = tot.x - mine.x; P = tot.y - mine.y;
            const int b0 = q * GABP_ROW_CHUNK, b1 = q == nc - 1 ? n : b0 + GABP_ROW_CHUNK;
            // (the chunk's first eight entries read together, the last chunk's remainder behind them)
            double2 c[GABP_ROW_CHUNK];
#pragma unroll
            for (int i = 0; i < GABP_ROW_CHUNK; ++i) c[i] = row[b0 + i];
#pragma unroll
            for (int i = 0; i < GABP_ROW_CHUNK; ++i) LHVI_ROW_STEP(c[i], b0 + i)
            for (int j = b0 + GABP_ROW_CHUNK; j < b1; ++j) { const double2 m = row[j]; LHVI_ROW_STEP(m, j) }
        }
#undef LHVI_ROW_STEP
    }
    const double var = 1.0 / P;
    st2(vnext, k, var * H, var);
    }
}

__global__ void __launch_bounds__(BLOCK) gabp_pull_hub_kernel(lhvi_graph_t g, lhvi_pots_t pots, PullPlan pl,
                                                             const double* __restrict__ vprev, double* __restrict__ vnext, int first) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= g.n_hubs) return;
    const int v = g.hub_vars[i];
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= GABP_HUB_DEGREE || !is_hidden(g.var_value[v])) return;     // observed rows: written by gabp_pull_kernel
    double H = 0.0, P = 0.0;
    // pass 1: the row's total; every slot's incoming message (one random gather + the closed form) is parked in the slot's own
    // output cell, which the same lane reads back in pass 2 -- one gather per slot and sweep instead of two
    for (int j = lo + lane; j < hi; j += 64) {
        const double c = pl.count ? pl.count[j] : 1.0;
        const double2 m = first ? make_double2(0.0, 1.0) : pull_incoming(g, pots, pl, vprev, j);
        st2(vnext, j, m.x, m.y);
        if (m.y != m.y) H -= m.x * c;
        else { const double p = 1.0 / m.y; H += p * m.x * c; P += p * c; }
    }
    H = dpp_wave_reduce(H, SumOp()); P = dpp_wave_reduce(P, SumOp());
    for (int j = lo + lane; j < hi; j += 64) {
        const double2 m = ld2(vnext, j);
        double h = H, p = P;
        if (m.y != m.y) h += m.x;
        else { const double q = 1.0 / m.y; h -= q * m.x; p -= q; }
        const double var = 1.0 / p;
        st2(vnext, j, var * h, var);
    }
}


// the same with the slot records (and the potential parameters in LDS): a slot's incoming message costs the record, the partner's
// message and arithmetic -- no walk through info -> pots.kind -> pots.off -> parameters per slot
template <bool LDS_POTS>
__global__ void __launch_bounds__(BLOCK) gabp_pull_hub_rec_kernel(lhvi_graph_t g, lhvi_pots_t pots, const int4* __restrict__ rec,
                                                                 const double* __restrict__ pot_words, const double* __restrict__ count,
                                                                 const double* __restrict__ vprev, double* __restrict__ vnext, int first) {
    __shared__ double sh_par[LDS_POTS ? GABP_LDS_POTS * GABP_POT_WORDS : 1];
    if (LDS_POTS) {
        for (int i = threadIdx.x; i < pots.P * GABP_POT_WORDS; i += BLOCK) sh_par[i] = pot_words[i];
        __syncthreads();
    }
    const double* __restrict__ pw = LDS_POTS ? sh_par : pot_words;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= g.n_hubs) return;
    const int v = g.hub_vars[i];
    const int lo = g.var_ptr[v], hi = g.var_ptr[v + 1];
    if (hi - lo <= GABP_HUB_DEGREE) return;
    if (!is_hidden(g.var_value[v])) {                          // an observed variable sends nothing (GaBP.py:39-40): NaN rows
        for (int j = lo + lane; j < hi; j += 64) st2(vnext, j, NAN, NAN);
        return;
    }
    double H = 0.0, P = 0.0;
    for (int j = lo + lane; j < hi; j += 64) {
        const double c = count ? count[j] : 1.0;
        const double2 m = first ? make_double2(0.0, 1.0) : pull_incoming_rec<LDS_POTS>(g, pots, rec[j], vprev, pw);
        st2(vnext, j, m.x, m.y);
        if (m.y != m.y) H -= m.x * c;
        else { const double p = 1.0 / m.y; H += p * m.x * c; P += p * c; }
    }
    H = dpp_wave_reduce(H, SumOp()); P = dpp_wave_reduce(P, SumOp());
    for (int j = lo + lane; j < hi; j += 64) {
        const double2 m = ld2(vnext, j);
        double h = H, p = P;
        if (m.y != m.y) h += m.x;
        else { const double q = 1.0 / m.y; h -= q * m.x; p -= q; }
        const double var = 1.0 / p;
        st2(vnext, j, var * h, var);
    }
}

// slot order -> edge order (the layout of lhvi_gabp_v2f / _f2v / _marginals and of the solvers' message views)
__global__ void __launch_bounds__(BLOCK) gabp_unpack_kernel(lhvi_graph_t g, const double* __restrict__ vslot, double* __restrict__ v2f) {
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= g.nnz) return;
    const double2 m = ld2(vslot, k);
    st2(v2f, g.var_edge[k], m.x, m.y);
}

static int validate(const lhvi_graph_t* g) {
    if (!g) return LHVI_E_ARG;
    if (g->V < 0 || g->F < 0 || g->E < 0 || g->nnz < 0) return LHVI_E_ARG;
    if (g->E > 0 && (!g->fac_ptr || !g->edge_var || !g->edge_fac || !g->var_ptr || !g->var_edge ||
                     !g->fac_pot || !g->var_value)) return LHVI_E_ARG;
    return LHVI_OK;
}

}  // namespace lhvi

using namespace lhvi;

extern "C" {

int lhvi_gabp_init(const lhvi_graph_t* g, double* f2v, double* v2f, void* stream) {
    if (int rc = validate(g)) return rc;
    if (g->E == 0) return LHVI_OK;
    if (!f2v || !v2f) return LHVI_E_ARG;
    hipLaunchKernelGGL(gabp_init_kernel, dim3(grid_for(g->E)), dim3(BLOCK), 0, as_stream(stream), (int64_t)g->E, f2v, v2f);
    return check_launch();
}

int lhvi_gabp_v2f(const lhvi_graph_t* g, const double* f2v, double* v2f, void* stream) {
    if (int rc = validate(g)) return rc;
    if (g->nnz == 0) return LHVI_OK;
    if (!f2v || !v2f) return LHVI_E_ARG;
    hipLaunchKernelGGL(gabp_v2f_kernel, dim3(grid_for(g->nnz)), dim3(BLOCK), 0, as_stream(stream), *g, f2v, v2f);
    if (g->hub_vars && g->n_hubs > 0)
        hipLaunchKernelGGL(gabp_v2f_hub_kernel, dim3(grid_for((int64_t)g->n_hubs * 64)), dim3(BLOCK), 0, as_stream(stream), *g, f2v, v2f);
    return check_launch();
}

int lhvi_gabp_f2v(const lhvi_graph_t* g, const lhvi_pots_t* pots, const double* v2f, double* f2v, void* stream) {
    if (int rc = validate(g)) return rc;
    if (!pots || (g->F > 0 && (!pots->kind || !pots->off))) return LHVI_E_ARG;
    if (g->E == 0) return LHVI_OK;
    if (!f2v || !v2f) return LHVI_E_ARG;
    hipLaunchKernelGGL(gabp_f2v_kernel, dim3(grid_for(g->E)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, v2f, f2v);
    return check_launch();
}

int lhvi_gabp_run(const lhvi_graph_t* g, const lhvi_pots_t* pots, double* f2v, double* v2f, int iterations, void* stream) {
    if (iterations < 0) return LHVI_E_ARG;
    if (int rc = lhvi_gabp_init(g, f2v, v2f, stream)) return rc;
    for (int i = 0; i < iterations; ++i) {
        if (int rc = lhvi_gabp_v2f(g, f2v, v2f, stream)) return rc;
        if (i < iterations - 1)
            if (int rc = lhvi_gabp_f2v(g, pots, v2f, f2v, stream)) return rc;
    }
    return LHVI_OK;
}

size_t lhvi_gabp_pull_workspace_bytes(const lhvi_graph_t* g) {
    return g ? (size_t)2 * (size_t)(g->nnz > 0 ? g->nnz : 1) * 16 : 0;
}

static int validate_plan(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan) {
    if (int rc = validate(g)) return rc;
    if (!pots || !plan || (g->F > 0 && (!pots->kind || !pots->off))) return LHVI_E_ARG;
    if (g->nnz > 0 && (!plan->pslot || !plan->info || !g->slot_var || !g->hub_vars)) return LHVI_E_ARG;
    return LHVI_OK;
}

int lhvi_gabp_pull(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, const double* v_prev,
                   double* v_next, int first, void* stream) {
    if (int rc = validate_plan(g, pots, plan)) return rc;
    if (g->nnz == 0) return LHVI_OK;
    if (!v_next || (!first && !v_prev) || v_prev == v_next) return LHVI_E_ARG;
    PullPlan pl;
    pl.pslot = plan->pslot; pl.info = plan->info; pl.count = plan->count;
    const bool records = plan->rec && plan->pot_words && plan->seg && plan->n_seg >= 0;
    if (records && plan->n_seg == 0) {}                        // (every row is a hub row)
    else if (records && pots->P <= GABP_LDS_POTS)
        hipLaunchKernelGGL(gabp_pull_rec_kernel<true>, dim3((unsigned)plan->n_seg), dim3(BLOCK), 0, as_stream(stream), *g, *pots,
                           reinterpret_cast<const int4*>(plan->rec), reinterpret_cast<const int2*>(plan->seg), plan->pot_words, plan->count,
                           v_prev, v_next, first);
    else if (records)
        hipLaunchKernelGGL(gabp_pull_rec_kernel<false>, dim3((unsigned)plan->n_seg), dim3(BLOCK), 0, as_stream(stream), *g, *pots,
                           reinterpret_cast<const int4*>(plan->rec), reinterpret_cast<const int2*>(plan->seg), plan->pot_words, plan->count,
                           v_prev, v_next, first);
    else
        hipLaunchKernelGGL(gabp_pull_kernel, dim3(grid_for(g->nnz)), dim3(BLOCK), 0, as_stream(stream), *g, *pots, pl, v_prev, v_next, first);
    if (g->hub_vars && g->n_hubs > 0 && plan->n_hub_rows != 0) {
        const dim3 hgrid(grid_for((int64_t)g->n_hubs * 64));
        if (records && pots->P <= GABP_LDS_POTS)
            hipLaunchKernelGGL(gabp_pull_hub_rec_kernel<true>, hgrid, dim3(BLOCK), 0, as_stream(stream), *g, *pots,
                               reinterpret_cast<const int4*>(plan->rec), plan->pot_words, plan->count, v_prev, v_next, first);
        else if (records)
            hipLaunchKernelGGL(gabp_pull_hub_rec_kernel<false>, hgrid, dim3(BLOCK), 0, as_stream(stream), *g, *pots,
                               reinterpret_cast<const int4*>(plan->rec), plan->pot_words, plan->count, v_prev, v_next, first);
        else
            hipLaunchKernelGGL(gabp_pull_hub_kernel, hgrid, dim3(BLOCK), 0, as_stream(stream), *g, *pots, pl, v_prev, v_next, first);
    }
    return check_launch();
}

int lhvi_gabp_run_pull(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, double* f2v, double* v2f,
                       int iterations, void* ws, size_t ws_bytes, void* stream) {
    if (iterations < 0) return LHVI_E_ARG;
    if (int rc = validate_plan(g, pots, plan)) return rc;
    if (int rc = lhvi_gabp_init(g, f2v, v2f, stream)) return rc;
    if (iterations == 0 || g->nnz == 0) return LHVI_OK;
    if (!ws || ws_bytes < lhvi_gabp_pull_workspace_bytes(g)) return LHVI_E_ARG;
    double* a = (double*)ws;
    double* b = a + (size_t)2 * g->nnz;
    for (int i = 0; i < iterations; ++i) {
        if (int rc = lhvi_gabp_pull(g, pots, plan, a, b, i == 0, stream)) return rc;
        double* t = a; a = b; b = t;
    }
    // a = v -> f of the last sweep, b = of the one before: the f -> v messages the reference ends with (its last sweep skips
    // that half, GaBP.py:161) are the closed forms of b
    const dim3 grid(grid_for(g->nnz));
    if (iterations >= 2) {
        hipLaunchKernelGGL(gabp_unpack_kernel, grid, dim3(BLOCK), 0, as_stream(stream), *g, b, v2f);
        if (int rc = lhvi_gabp_f2v(g, pots, v2f, f2v, stream)) return rc;
    }
    hipLaunchKernelGGL(gabp_unpack_kernel, grid, dim3(BLOCK), 0, as_stream(stream), *g, a, v2f);
    return check_launch();
}

// A whole run -- `iterations` pull sweeps, the conversion to edge order and the marginals -- recorded once as a hipGraph and
// replayed with one launch: on a template-sized graph (cfg 2: 20 k edges, 23 launches of a few microseconds each) the launches
// are the run.  The handle owns nothing but the executable graph; every buffer stays the caller's and is baked in.
struct GabpGraph { hipGraph_t graph; hipGraphExec_t exec; };

int lhvi_gabp_graph_create(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, double* f2v, double* v2f,
                           double* mu_var, int iterations, void* ws, size_t ws_bytes, void** handle_out) {
    if (!handle_out || !mu_var) return LHVI_E_ARG;
    *handle_out = nullptr;
    hipStream_t cap;
    if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) return LHVI_E_LAUNCH;
    int rc = LHVI_OK;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipStreamDestroy(cap); return LHVI_E_LAUNCH; }
    rc = lhvi_gabp_run_pull(g, pots, plan, f2v, v2f, iterations, ws, ws_bytes, cap);
    if (rc == LHVI_OK) rc = lhvi_gabp_marginals(g, f2v, mu_var, cap);
    const hipError_t end = hipStreamEndCapture(cap, &graph);
    (void)hipStreamDestroy(cap);
    if (rc != LHVI_OK || end != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); return rc != LHVI_OK ? rc : LHVI_E_LAUNCH; }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(graph); return LHVI_E_LAUNCH; }
    GabpGraph* h = new GabpGraph{graph, exec};
    *handle_out = h;
    return LHVI_OK;
}

int lhvi_gabp_graph_launch(void* handle, void* stream) {
    if (!handle) return LHVI_E_ARG;
    if (hipGraphLaunch(static_cast<GabpGraph*>(handle)->exec, as_stream(stream)) != hipSuccess) { g_last_hip_error = (int)hipGetLastError(); return LHVI_E_LAUNCH; }
    return LHVI_OK;
}

int lhvi_gabp_graph_destroy(void* handle) {
    if (!handle) return LHVI_OK;
    GabpGraph* h = static_cast<GabpGraph*>(handle);
    (void)hipGraphExecDestroy(h->exec);
    (void)hipGraphDestroy(h->graph);
    delete h;
    return LHVI_OK;
}

int lhvi_gabp_marginals(const lhvi_graph_t* g, const double* f2v, double* mu_var, void* stream) {
    if (int rc = validate(g)) return rc;
    if (g->V == 0) return LHVI_OK;
    if (!f2v || !mu_var) return LHVI_E_ARG;
    hipLaunchKernelGGL(gabp_marginal_kernel, dim3(grid_for(g->V)), dim3(BLOCK), 0, as_stream(stream), *g, f2v, mu_var);
    if (g->hub_vars && g->n_hubs > 0)
        hipLaunchKernelGGL(gabp_marginal_hub_kernel, dim3(grid_for((int64_t)g->n_hubs * 64)), dim3(BLOCK), 0, as_stream(stream), *g, f2v, mu_var);
    return check_launch();
}

}  // extern "C"
