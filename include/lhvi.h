/*
 * lhvi.h -- C ABI of the MI355X (gfx950) message-passing library `liblhvi.so`.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (leodd/Lifted-Hybrid-Variational-Inference) has no FFI: its boundary is the Python method surface
 * of the solver classes.  Each entry point below replaces the *body* of one of those methods and
 * cites it; the Python classes in lifted-hybrid-variational-inference_amd/lhvi/ keep the reference's
 * signatures and call these through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C, no torch / C++ types; every pointer inside the structs and every array argument is a
 *     DEVICE pointer (hipMalloc'd or a torch tensor's data_ptr()); the structs themselves live on the host.
 *   - all calls are asynchronous on the caller's `hipStream_t` (passed as void*; NULL = default stream),
 *     allocate nothing, keep no global state, and are re-entrant per stream.
 *   - return 0 on success, a negative LHVI_E_* code otherwise; nothing throws across the boundary.
 *   - all floating point is IEEE fp64, like the reference (Python floats).
 *   - a Gaussian message is two doubles (mu, var); var = NaN encodes the reference's `None`
 *     variance (pure linear term, GaBP.py:126), var = +Inf the vacuous message (GaBP.py:138).
 */
#ifndef LHVI_H
#define LHVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LHVI_ABI_VERSION 11  /* 2: lhvi_graph_t gained edge_value / slot_var / hub_vars, lhvi_pbp_t the heavy / light descriptor lists, 128-byte descriptors;
                              * 3: lhvi_pbp_t gained var_lo / var_hi;  4: f2v_ticket;  5: prop_desc;  6: lhvi_vi_t gained obs_var, lhvi_gabp_plan_t;  7: lhvi_pbp_t gained pair_desc;
                              * 8: lhvi_pbp_t gained cq_desc / n_cq, lhvi_pbp_classify takes the particle state, lhvi_pbp_describe_cq; the colour
                              *    refinement calls take a method and return four result words; lhvi_vi_t gained var_N; lhvi_vi_opt_t, lhvi_vi_adam_run;
                              *    lhvi_gabp_plan_t.n_hub_rows, lhvi_gabp_graph_*; lhvi_pbp_t gained v2f_wide / v2f_narrow / v2f_hub / v2f_mid16 / v2f_mid32, prop_hub / prop_partial, resample_vars, small16_desc / small32_desc; 16 ticket words; lhvi_pbp_boundary_reduce;
                              * 9: lhvi_vi_t gained fac_list / n_cc / n_tiny / n_grp3 / n_grp6 / n_rest3 / n_rest6 / edge_axis; lhvi_color_first_members, lhvi_color_segment_sums, lhvi_pbp_halo_pack / _unpack; lhvi_gabp_plan_t.rec;
                              * 10: lhvi_pbp_t gained halo_off / halo_buf, LHVI_PBP_NO_UNIQ, edge_canon may name rows beyond E; lhvi_pbp_map_brent, lhvi_pbp_quad;
                              * 11: LHVI_PBP_V2F_RECORDS (v2f_wide as 8-word records), LHVI_PBP_WIDE_PAIRS, LHVI_PBP_SHARE_CUS */
#define LHVI_MAX_ARITY 6

/* error codes */
#define LHVI_OK 0
#define LHVI_E_ARG (-1)        /* null pointer / negative size / inconsistent struct */
#define LHVI_E_LAUNCH (-2)     /* hipLaunch failed; see lhvi_last_hip_error() */
#define LHVI_E_UNSUPPORTED (-3)/* e.g. arity > LHVI_MAX_ARITY, n not supported */
#define LHVI_E_NODEVICE (-4)   /* no HIP device visible */

/* potential kinds (pot_kind[]); parameter layouts are documented in csrc/potential.hpp */
#define LHVI_POT_GENERIC 0
#define LHVI_POT_TABLE 1
#define LHVI_POT_GAUSSIAN 2
#define LHVI_POT_QUADRATIC 3
#define LHVI_POT_HYBRID_QUADRATIC 4
#define LHVI_POT_LINEAR_GAUSSIAN 5
#define LHVI_POT_X2 6
#define LHVI_POT_XY 7
#define LHVI_POT_MLN 8
#define LHVI_POT_MLN_HARD 9
#define LHVI_POT_IMAGE_NODE 10
#define LHVI_POT_IMAGE_EDGE 11

/* Flat factor graph (ground or lifted).  Built by lhvi/flat.py from Graph / CompressedGraph objects
 * (Graph.py:137-209, CompressedGraphWithObs.py:178-271).  Edges are (factor, argument) incidences,
 * factor-major: the edges of factor f are fac_ptr[f] .. fac_ptr[f+1]-1 in argument order. */
typedef struct lhvi_graph {
    int32_t V, F, E, nnz;       /* nnz = length of var_edge (== E unless a lifted factor repeats a cluster) */
    const int32_t* fac_ptr;     /* [F+1] */
    const int32_t* edge_var;    /* [E] variable of edge e */
    const int32_t* edge_fac;    /* [E] factor of edge e */
    const int32_t* edge_canon;  /* [E] canonical edge of the (factor, variable) pair, or NULL if all e: the ROW of the message arrays that
                                 * holds the pair's messages.  An edge with edge_canon[e] != e is served by no kernel (its messages are
                                 * another row's).  (ABI 10) edge_canon[e] >= E is allowed for the particle sweep: a row of the caller's
                                 * v -> f array BEHIND the graph's own E rows, where a message computed elsewhere arrives in place
                                 * (the ghost edges of the owner-computes split, lhvi/dist.py::OwnerPlan.ghost_rows) */
    const int32_t* var_ptr;     /* [V+1] */
    const int32_t* var_edge;    /* [nnz] canonical edge ids in rv.nb order */
    const double* edge_count;   /* [E] lifted multiplicity rv.count[f], or NULL (ground: all 1) */
    const int32_t* fac_pot;     /* [F] index into the potential table */
    const double* var_value;    /* [V] evidence value, NaN = hidden */
    const int32_t* var_dom;     /* [V] domain id */
    const double* var_mult;     /* [V] |cluster| (len(rv.rvs)), or NULL */
    const double* fac_mult;     /* [F] |cluster| (len(f.factors)), or NULL */
    int32_t D;                  /* number of domains */
    const int32_t* dom_cont;    /* [D] 1 = continuous */
    const double* dom_lo;       /* [D] */
    const double* dom_hi;       /* [D] */
    const int32_t* dom_ptr;     /* [D+1] into dom_val */
    const double* dom_val;      /* discrete: the states; continuous: the integral points */
    /* optional denormalised copies that turn random gathers of the Gaussian sweep into contiguous reads (NULL = gather) */
    const double* edge_value;   /* [E] var_value[edge_var[e]] */
    const int32_t* slot_var;    /* [nnz] variable of CSR slot k (the v with var_ptr[v] <= k < var_ptr[v+1]) */
    /* optional: the variables with more than LHVI_HUB_DEGREE incident edges (template variables of relational models).
     * Kernels that walk a variable's CSR row give these a wavefront each; without the list (NULL) the hub kernels
     * scan all V variables for them (colour refinement, variational gather) or the row is walked by one thread
     * (Gaussian v2f / marginals, which keep the reference's summation order up to 512 edges and use the list beyond:
     * direct O(deg^2) leave-one-out sums). */
    const int32_t* hub_vars;    /* [n_hubs] ascending variable ids */
    int32_t n_hubs;
} lhvi_graph_t;
#define LHVI_HUB_DEGREE 64

/* Potential table: one row per distinct (potential object, scope domains). */
typedef struct lhvi_pots {
    int32_t P;
    const int32_t* kind;        /* [P] LHVI_POT_* */
    const int32_t* off;         /* [P+1] into param */
    const double* param;
    int32_t interpreted;        /* (ABI 10) how many rows are formulas the device evaluates by interpreting their bytecode: MLN rows without a
                                 * conditional-quadratic block (cq_off = 0) and MLN_HARD rows.  0 (every formula the reference ships): the
                                 * kernels that evaluate general potentials run the builds compiled without the interpreter.  A caller that
                                 * does not know passes -1 (treated as "some") */
} lhvi_pots_t;

int lhvi_version(void);
const char* lhvi_strerror(int code);
int lhvi_last_hip_error(void);       /* hipError_t of the last failed launch on this thread */
int lhvi_device_count(void);

/* ---- Gaussian BP (GaBP.py / GaLBP.py) ----------------------------------------------------------
 * f2v / v2f: [E][2] doubles (mu, var), indexed by edge id. */

/* all messages (mu,var) = (0,1): GaBP.py:142-145, GaLBP.py:154-157 */
int lhvi_gabp_init(const lhvi_graph_t* g, double* f2v, double* v2f, void* stream);
/* variable -> factor half sweep: GaBP.message_rv_to_f GaBP.py:20-35, GaLBP.py:21-39 */
int lhvi_gabp_v2f(const lhvi_graph_t* g, const double* f2v, double* v2f, void* stream);
/* factor -> variable half sweep: GaBP.message_f_to_rv GaBP.py:37-138, GaLBP.py:41-142 */
int lhvi_gabp_f2v(const lhvi_graph_t* g, const lhvi_pots_t* pots, const double* v2f, double* f2v, void* stream);
/* `iterations` flooding sweeps; the last one skips f2v: GaBP.run GaBP.py:140-169, GaLBP.run GaLBP.py:159-181 */
int lhvi_gabp_run(const lhvi_graph_t* g, const lhvi_pots_t* pots, double* f2v, double* v2f, int iterations, void* stream);
/* Pull form of the Gaussian sweep (one launch per iteration; replaces the pair lhvi_gabp_v2f + lhvi_gabp_f2v of
 * GaBP.run GaBP.py:152-165 / GaLBP.run GaLBP.py:160-177 for graphs whose factors are unary or pairwise -- every factor
 * GaBP.message_f_to_rv knows a closed form for, GaBP.py:37-138).  Messages live in variable-CSR ("slot") order,
 * slot k = (variable slot_var[k], edge var_edge[k]); the f -> v message of a slot is recomputed from the partner's
 * previous v -> f message instead of being stored.  The caller builds the plan once per graph:
 *   pslot[k]  slot of the partner argument's (canonical) edge when the partner variable is hidden; -1 - (partner variable)
 *             when it is observed (its value is read from var_value); unused for codes 0 and 3
 *   info[k]   4 * potential index + code; code 0 = unary factor, 1 / 2 = pairwise factor with this variable at
 *             position 0 / 1, 3 = any other arity (vacuous message (0, Inf), GaBP.py:138)
 *   count[k]  lifted graphs: rv.count[f] of the slot (GaLBP.py:24-34); NULL on a ground graph
 * Results equal the two-kernel path bit for bit (same expressions, same summation order). */
typedef struct lhvi_gabp_plan {
    const int32_t* pslot;
    const int32_t* info;
    const double* count;
    int32_t n_hub_rows;     /* variables with more than 512 incident edges (served by the wave-parallel hub kernel); 0 skips that
                             * launch, -1 = not counted (the kernel is launched whenever the graph lists hub_vars) */
    const int32_t* rec;     /* (ABI 9) [nnz][4] or NULL: per slot {pslot[k], info[k], position of k in its row | row length << 10 |
                             * variable hidden << 20 | row longer than 512 << 21, 0}: with it a slot needs no lookup through
                             * slot_var / var_ptr / var_value (GaBP.message_rv_to_f, GaBP.py:20-35, reads rv.nb and rv.value) */
    const double* pot_words;/* (ABI 9) [P][12], required with rec: per potential its first eleven parameters (zero padded; the closed forms
                             * of GaBP.message_f_to_rv, GaBP.py:37-138, read at most par[10]) and its kind as a double -- one record
                             * instead of the walk pots.kind -> pots.off -> pots.param */
    const int32_t* seg;     /* (ABI 9) [n_seg][2], required with rec: the slot order cut into segments [lo, hi) of whole rows -- the rows
                             * of at most 512 entries that start inside one window of 256 slots, up to the next row of more than 512
                             * entries (those belong to the hub kernel) -- one workgroup each.  BOUND: hi - lo <= 768 (the kernel
                             * stages a segment whole in LDS: window - 1 + 512 slots at most, window <= 256); a longer segment is
                             * left unswept */
    int32_t n_seg;
} lhvi_gabp_plan_t;
size_t lhvi_gabp_pull_workspace_bytes(const lhvi_graph_t* g);
/* one sweep: v_next[k] = message_rv_to_f of slot k given the f -> v messages implied by v_prev (first != 0: given the
 * initial messages (0, 1); v_prev is not read).  v_prev, v_next: [nnz][2], distinct. */
int lhvi_gabp_pull(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, const double* v_prev,
                   double* v_next, int first, void* stream);
/* lhvi_gabp_run through the pull form: same f2v [E][2] / v2f [E][2] (edge order) on return.  ws: caller-owned scratch of
 * lhvi_gabp_pull_workspace_bytes(g) bytes. */
int lhvi_gabp_run_pull(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, double* f2v, double* v2f,
                       int iterations, void* ws, size_t ws_bytes, void* stream);

/* The whole of GaBP.run + the marginals (lhvi_gabp_run_pull, then lhvi_gabp_marginals into mu_var [V][2]) recorded once as a
 * hipGraph over the caller's buffers and replayed with ONE launch: on template-sized graphs (BASELINE cfg 2: 20 k edges, 23
 * launches) the launches are the run.  The handle owns only the executable graph; rebuild it when the graph, the plan, a
 * buffer or `iterations` changes.  Evidence values are read from g->var_value at replay time, so new evidence in the same
 * buffer needs no rebuild. */
int lhvi_gabp_graph_create(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_gabp_plan_t* plan, double* f2v, double* v2f,
                           double* mu_var, int iterations, void* ws, size_t ws_bytes, void** handle_out);
int lhvi_gabp_graph_launch(void* handle, void* stream);
int lhvi_gabp_graph_destroy(void* handle);

/* per-variable product of incoming messages -> mu_var [V][2]; evidence rows get (value, 0):
 * GaBP.get_belief_params GaBP.py:187-200, GaLBP.map GaLBP.py:201-217 */
int lhvi_gabp_marginals(const lhvi_graph_t* g, const double* f2v, double* mu_var, void* stream);

/* utils.log_likelihood (utils.py:6-15) of a full assignment x [V] on flat arrays: out[0] = -sum_f log phi_f(x_scope),
 * or -inf when some factor is 0 at x (the reference's convention).  Discrete arguments are matched to their states. */
size_t lhvi_log_likelihood_workspace_bytes(const lhvi_graph_t* g);
int lhvi_log_likelihood(const lhvi_graph_t* g, const lhvi_pots_t* pots, const double* x, double* out, void* ws, size_t ws_bytes,
                        void* stream);

/* ---- Particle BP (EPBPLogVersion.py / HybridLBPLogVersion.py) -----------------------------------
 * Log messages are tabulated per edge: f2v [E][n+T] (first n = at the variable's particles, next T = at
 * the variable's integral points), v2f [E][n]. */

#define LHVI_PBP_EP 1u            /* proposal_approximation == 'EP' (else 'simple') */
#define LHVI_PBP_EPBP_DISCRETE 2u /* EPBP applies importance weights to discrete rvs too (EPBP.py:157) */
#define LHVI_PBP_SKIP_FAST 4u     /* lhvi_pbp_f2v: launch neither quadratic-family kernel (profiling aid) */
#define LHVI_PBP_SKIP_GENERIC 8u  /* lhvi_pbp_f2v: do not launch the generic-potential kernel (profiling aid) */
#define LHVI_PBP_SKIP_TERMS 16u   /* lhvi_pbp_f2v: the quadratic-family kernels skip their term loops -- results are
                                   * meaningless; isolates the per-edge load/store cost when tuning */
#define LHVI_PBP_SKIP_HEAVY 32u   /* lhvi_pbp_f2v: do not launch the continuous x continuous (heavy_desc) kernel (profiling aid) */
#define LHVI_PBP_SKIP_LIGHT 64u   /* lhvi_pbp_f2v: do not launch the kernels of the remaining fast edges (light_desc and fast_edges;
                                   * profiling aid) */
#define LHVI_PBP_LEAVE_ROOM 256u /* lhvi_pbp_f2v: the persistent kernels launch cus/8 workgroups fewer than fill the device, so that
                                   * another stream's kernels (RCCL's copy kernels of an overlapped exchange) find free slots at
                                   * any time instead of waiting for a persistent workgroup to retire.  Set by sharded runs. */
#define LHVI_PBP_CQ 512u         /* route MLN factors whose formula is conditionally quadratic (lhvi/expr.py::cq_block: a polynomial of degree <= 2
                                   * in the continuous arguments for every state of the discrete ones, e.g. x[0] * eq_op(x[1], x[2]),
                                   * Demo/Data/HMLN/GeneratorPaperPopularity.py:28-40) to the quadratic-family kernels instead of the
                                   * generic interpreter kernel: set it for lhvi_pbp_classify / _describe / _describe_cq AND lhvi_pbp_f2v */
#define LHVI_PBP_SKIP_CQ 1024u   /* lhvi_pbp_f2v: do not launch the kernel of the cq_desc list (profiling aid) */
#define LHVI_PBP_BOUNDARY_TOTALS 2048u /* sharded runs: a boundary variable's one listed row of s->recv holds the finished total over all ranks
                                        * (lhvi_pbp_boundary_reduce); without it the rows are the peers' sums and the kernels add them */
#define LHVI_PBP_NO_UNIQ 4096u   /* lhvi_pbp_resample_uniq with a resample_vars list: draw the particles only, leave uniq_out untouched (ghost variables of
                                  * the owner-computes split: their first-occurrence masks are read by nobody on this rank) */
#define LHVI_PBP_V2F_RECORDS 8192u /* v2f_wide holds one 8-word record per variable instead of its id: 0 variable  1 incident edges  2 particles (np)
                                    * 3 domain  4-7 the first four incident edges, in row order (a shorter row: its last edge repeated).  The
                                    * kernel then reaches the f -> v rows after ONE dependent load instead of three (id -> var_ptr / np -> var_edge) */
#define LHVI_PBP_WIDE_PAIRS 16384u /* lhvi_pbp_f2v: the pair_desc list always through the one-entry-per-wavefront kernel, also when n <= 32 would let
                                   * four / two entries share a wavefront (testing aid: the two kernels give the same bits) */
#define LHVI_PBP_SHARE_CUS 32768u /* lhvi_pbp_f2v: the persistent grids of the long kernels (heavy_desc, small16 / small32) take one workgroup per CU less than
                                   * fits, so that the short kernels of the same half sweep, launched by a second call on ANOTHER stream (the pair /
                                   * light / cq / generic lists: disjoint rows of f2v), find a wave slot and LDS on every CU and run beside them */
#define LHVI_PBP_NO_GRID 128u    /* lhvi_pbp_f2v: integral points always by the direct form (one exponential per term), never by the
                                   * uniform-grid recurrence (testing / profiling aid) */
#define LHVI_PBP_FUSED_RECORDS16 131072u /* lhvi_pbp_var_fused: desc holds SIXTEEN 32-bit words per variable (64-byte aligned) -- words 0-7 as documented
                                   * there, then 8 particles of the variable (s->np[v])  9 g->var_ptr[v]  10-15 its first six incident edges
                                   * (var_edge[var_ptr[v] + 0 .. 5]; unused ones 0) -- so that a variable's rows hang on one load behind its record */
#define LHVI_PBP_POW2_GROUPS 65536u /* lhvi_pbp_f2v: the small16 / small32 lists always through lane groups of 16 / 32 lanes (four / two edges per
                                   * wavefront), also when s->n would let up to eight edges share one (narrower groups, two particles per lane; testing / profiling aid) */

typedef struct lhvi_pbp {
    int32_t n;                  /* particle slots per variable */
    int32_t T;                  /* integral-point slots per variable (max over domains) */
    uint32_t flags;
    double var_threshold;       /* EPBP 3, HybridLBP 5 */
    double max_log_value;       /* 700 */
    const double* particles;    /* [V][n] current sample */
    const double* old_particles;/* [V][n] sample the v2f messages were computed on */
    const int32_t* np;          /* [V] valid particles: n (continuous hidden), #states (discrete hidden), 0 (observed) */
    const uint8_t* uniq;        /* [V][n] 1 = first occurrence of that value among the variable's particles */
    const double* q;            /* [V][2] proposal (mu, var) */
    /* optional work lists for lhvi_pbp_f2v (from lhvi_pbp_classify); NULL = classify every edge on the fly */
    const int32_t* fast_edges;  /* [n_fast] edges whose log phi is quadratic in the target (LDS-staged kernel) */
    int32_t n_fast;
    const int32_t* generic_edges; /* [n_generic] all other edges with a hidden target */
    int32_t n_generic;
    int32_t generic_pts_log2;   /* ceil(log2(max output points of a generic edge)), clamped to [0, 6]: lanes per edge */
    const void* fast_desc;      /* [n_fast][LHVI_PBP_DESC_BYTES] from lhvi_pbp_describe, or NULL (built on the fly) */
    const void* heavy_desc;     /* [n_heavy][LHVI_PBP_DESC_BYTES] descriptors of the edges served by the specialised kernel:
                                 * class 1, constant x^2 coefficient (kind != HYBRID_QUADRATIC), nj <= 64, and either np + T <= 128 or a uniform
                                 * grid (word 15) of T <= 128 points with nj >= 24 and np <= 128; disjoint from fast_edges */
    int32_t n_heavy;
    const void* light_desc;     /* [n_light] descriptors with word 14 != 0: HybridQuadratic edges with a binary (or observed)
                                 * discrete side, served by their own kernel; disjoint from fast_edges and heavy_desc */
    int32_t n_light;
    /* edge-sharded runs only (all NULL on a single GPU).  A boundary variable (edges on several ranks) owns one row per
     * peer rank in the exchange buffers; row r of the send buffer and row r of the receive buffer belong to the same
     * (variable, peer) because both ends list their shared variables in ascending global id.  A row of a continuous
     * variable is n + 2 doubles: [sum over the sender's local edges of count * f2v[e][j], j < n | sum of count/var, sum of
     * count*mu/var of its sites]; a row of a discrete variable is np doubles (the sums at its states; it has no proposal).
     * Rows are packed back to back, peer-major; brow_off gives each row's element offset in either buffer. */
    const int32_t* bslot;       /* [V] index of the variable in the boundary list, -1 for interior / observed variables */
    const int32_t* brow_ptr;    /* [nb+1] CSR over exchange rows per boundary variable, rows in ascending peer rank */
    const int64_t* brow_off;    /* [rows] element offset of the row (the same offsets address the send and the receive buffer) */
    const int32_t* brow_peer;   /* [rows] peer rank of each listed row */
    const double* recv;         /* received rows of this sweep, packed like the send buffer */
    int32_t rank;               /* this rank (fixes the summation order of the proposals so that all replicas agree bit for bit) */
    const double* var_degree;   /* [V] global number of incoming messages (sum of counts over ALL ranks' edges) */
    /* variable range of the per-variable entry points (lhvi_pbp_v2f, _proposal, _proposal_partial, _proposal_finish,
     * _resample_uniq): they work on [var_lo, var_hi) when var_hi > var_lo, on every variable when both are 0.  A sharded
     * run numbers its interior variables first and sweeps them while the boundary rows are in flight. */
    int32_t var_lo, var_hi;
    /* optional: LHVI_PBP_TICKET_WORDS 32-bit words of device memory for lhvi_pbp_f2v.  When set, the persistent heavy kernel
     * cuts its work list into one contiguous range per XCD (each XCD has its own L2) and hands each range out in chunks of
     * consecutive entries through an atomic counter, which the call resets on its stream: a wave's descriptors are then
     * neighbours in memory, and a workgroup dispatched late finds less left to do instead of owing a full static share
     * (13.4 -> 12.0 ms per launch on the 10 M-edge benchmark).  NULL: every wave strides over the list.  Words 8 and 9 come
     * back with the launch's grid-recurrence statistics (LHVI_PBP_TICKET_COUNTERS). */
    uint32_t* f2v_ticket;
    /* optional, lhvi_pbp_proposal only: one record of eight 32-bit words per hidden CONTINUOUS variable, host-built --
     *   0 variable   1 degree (entries of its var_edge row)   2 grid base in dom_val   3 T   4-7 the first four incident
     *   edges var_edge[var_ptr[v] + k] (unused: repeat the first) --
     * so that a wave starts from one scalar load instead of the chain var_value / var_dom -> dom_ptr / var_ptr -> var_edge,
     * and no wave is launched for an observed or discrete variable.  NULL: the kernel walks the graph arrays. */
    const int32_t* prop_desc;
    int32_t n_prop_desc;
    /* optional, lhvi_pbp_f2v only: the light edges again, one LHVI_PBP_DESC_BYTES-byte record per FACTOR (both edges of a
     * HybridQuadratic(1 discrete, 1 continuous) factor share its per-state coefficients), host-built from light_desc --
     *   words 0 e_c  1 e_d  (edge to the continuous / discrete variable, -1 = that variable is observed)   2 v_c  3 v_d
     *   4 np_c  5 T  6 grid base  7 live states of v_d   8-9 val_c  10-11 val_d (doubles, NaN = hidden)
     *   12-23 A0 b0 c0 A1 b1 c1 (doubles)   24 / 25 rows of v2f with the discrete / continuous variable's message
     * When set, one kernel over these records replaces the light kernel (twice the bytes in flight per wave, half the
     * descriptor traffic; same messages bit for bit).  NULL: light_desc is used.  With n <= 16 / n <= 32 particles four / two records
     * share a wavefront (same bits again; LHVI_PBP_WIDE_PAIRS keeps one record per wavefront). */
    const void* pair_desc;
    int32_t n_pair;
    /* optional, lhvi_pbp_f2v only: [n_cq][2 * LHVI_PBP_DESC_BYTES] records from lhvi_pbp_describe_cq for the edges of class 4
     * (conditionally quadratic factors with a hidden discrete and a hidden continuous partner, or two hidden continuous
     * partners of a discrete target); served by their own kernel. */
    const void* cq_desc;
    int32_t n_cq;
    /* optional, lhvi_pbp_v2f only (single-GPU runs: not with bslot or a variable range): the hidden variables split by particle
     * count -- v2f_wide: more than four particles, one wavefront each; v2f_narrow: at most four (binary variables, boolean
     * atoms), sixteen per wavefront.  Both or neither; together they must list every hidden variable once.  NULL: one wavefront
     * per variable of the range.  With LHVI_PBP_V2F_RECORDS in flags, v2f_wide is an array of 8-word records (see the flag). */
    const int32_t* v2f_wide;
    int32_t n_v2f_wide;
    const int32_t* v2f_narrow;
    int32_t n_v2f_narrow;
    const int32_t* v2f_hub;     /* optional third part of the split: variables with 5..64 particles and more than 64 incident edges
                                 * (template variables of relational models), a workgroup each; such variables are then NOT in
                                 * v2f_wide */
    int32_t n_v2f_hub;
    const int32_t* v2f_mid16;   /* optional further parts of the split (with v2f_wide / v2f_narrow): variables with 5-16 and with 17-32 particles */
    int32_t n_v2f_mid16;        /* and at most 64 incident edges, four / two per wavefront; such variables are then NOT in v2f_wide */
    const int32_t* v2f_mid32;
    int32_t n_v2f_mid32;
    /* optional, lhvi_pbp_proposal with prop_desc only: variables with long rows (template variables of relational models) listed in
     * prop_desc by SLICES of their var_edge row, a wavefront per slice instead of one per variable.  A slice record has word 1 =
     * -(entries in the slice), word 4 = position of its first entry in the row, word 5 = its slot in prop_partial (words 6-7
     * unused); prop_hub lists each such variable once as (variable, first slot, slices, 0) with its slices in consecutive slots in
     * row order, and a second small kernel adds the slices' information-form sums in that order.  prop_partial: [slots][2]
     * doubles of scratch. */
    const int32_t* prop_hub;
    int32_t n_prop_hub;
    double* prop_partial;
    /* optional, lhvi_pbp_resample_uniq only (n <= 64, not with a variable range): one record of eight 32-bit words per hidden
     * CONTINUOUS variable -- 0 variable   1 its particle count np   2-3 dom_lo   4-5 dom_hi (doubles)   6-7 unused.  The call then
     * draws for these only, two per wavefront, and touches no other row of particles_out / uniq_out -- the rows of discrete and
     * observed variables never change, so the caller fills them once (one call without the list per particle buffer).  NULL:
     * every variable of the range, neighbours two by two.  The draws do not depend on which form is used. */
    const int32_t* resample_vars;
    int32_t n_resample_vars;
    /* optional, lhvi_pbp_f2v only: heavy-class descriptors (same rows as heavy_desc would hold, and NOT in heavy_desc) of the edges
     * whose target AND partner have at most 16 / at most 32 particles (nj <= 16 and np <= 16; the rest with nj <= 32 and np <= 32);
     * any number of integral points.  Served four / two edges per wavefront by their own kernel -- six / ten / eight for the small16
     * list when s->n <= 10 / 12 / 16, six / five / four for the small32 list when s->n <= 20 / 24 / 32 (lane groups as narrow as
     * the particle count allows, two particles per lane beyond 10; no variable may then hold more than s->n particles;
     * LHVI_PBP_POW2_GROUPS keeps four / two): with the particle counts of the
     * reference's demos (10-20) an edge per wavefront is bound by its own latencies, not by its terms.  Integral points on a uniform
     * grid (descriptor word 15) are tabulated by the recurrence along the grid inside the lane group, like the heavy kernel's (to
     * rounding the same values as the direct form; LHVI_PBP_NO_GRID forces that one).  NULL: such edges stay in heavy_desc.  Skipped
     * with LHVI_PBP_SKIP_HEAVY. */
    const void* small16_desc;
    int32_t n_small16;
    const void* small32_desc;
    int32_t n_small32;
    /* (ABI 10) optional, lhvi_pbp_v2f only -- the owner-computes split of a sharded sweep (lhvi/dist.py::OwnerRunner): halo_off [E],
     * per edge the element offset in halo_buf that its v -> f row is ALSO written to as it is formed (the send buffer of the
     * sweep's all_to_all: no pack pass re-reads the rows), or -1.  NULL: no copies. */
    const int64_t* halo_off;
    double* halo_buf;
} lhvi_pbp_t;

#define LHVI_PBP_DESC_BYTES 128
#define LHVI_PBP_TICKET_WORDS 16     /* the work counters, then two words of statistics of the last lhvi_pbp_f2v call */
#define LHVI_PBP_TICKET_COUNTERS 8   /* words 0-7: one work counter per XCD range; word 8: heavy edges whose integral points were tabulated by the
                                      * uniform-grid recurrence; word 9: eligible edges that failed its range guard and took the direct form */
/* static per-edge descriptors of the fast work list: lets the persistent f2v kernels fetch everything about an edge with
 * scalar loads.  Must be rebuilt when np / the graph / the potentials change.  Layout (32-bit words unless noted):
 *   0 edge   1 target variable   2 partner variable   3 partner's canonical edge   4 class (lhvi_pbp_classify)
 *   5 target position   6 potential kind   7 nj = partner particle count (1 = observed)   8 np = target particle count
 *   9 T = target integral points   10 grid base in dom_val   11 offset into pots.param   12-13 partner value (double,
 *   NaN = hidden)   14 light-kernel type (0 none, 1 continuous target / discrete partner, 2 discrete target /
 *   continuous partner)   15 1 = the target's integral points are a uniform grid (x_t = x0 + t h to a few ulp)   16-27 six doubles: the potential resolved for this edge -- class 1 with a constant
 *   x^2 coefficient: log phi = kx x^2 + (ay y + by) y + c + (axy y + bx) x as (ay, by, c, axy, bx, kx), x = target;
 *   light edges: (A_0, b_0, c_0, A_1, b_1, c_1) of the discrete side's two points   28-31 two doubles (x0, h) of a uniform grid.
 * The host builds heavy_desc / light_desc / fast_desc by splitting the rows on words 4, 6, 7, 8 + 9 and 14. */
int lhvi_pbp_describe(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const int32_t* edges, int32_t count,
                      void* desc_out, void* stream);
/* descriptors of the class-4 edges (2 * LHVI_PBP_DESC_BYTES each).  Layout (32-bit words unless noted):
 *   0 edge   1 target variable   2 type (1 = continuous target, hidden discrete partner z + continuous partner y: the message is
 *   log sum_s sum_j exp(a_sj + b_sj x + k_s x^2); 2 = discrete target, two hidden continuous partners x, y: S sums over the n^2
 *   joint particles)   3 S = coefficient sets (states of z / of the target)   4 np   5 T   6 grid base   7 variable of y
 *   8 v2f row of y   9 particles of y (1 = observed or absent)   10 variable of z (type 2: of x)   11 v2f row of z (type 1: -1 =
 *   none; type 2: of x)   12 states of z (type 2: particles of x)   14-15 value of an observed y (double, NaN = hidden)
 *   16-63 S x six doubles (ay, by, c, axy, bx, kx): log phi_s = kx x^2 + (ay y + by) y + c + (axy y + bx) x
 * EPBP.message_f_to_rv on such factors: EPBP.py:176-194 with MLNPotential.get MLNPotential.py:36-37. */
int lhvi_pbp_describe_cq(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const int32_t* edges, int32_t count,
                         void* desc_out, void* stream);
/* Batched queries (extension; EPBP.belief_rv EPBP:196-202 for every variable at once).  Tabulate the messages at n query
 * points per variable by running lhvi_pbp_f2v with s->particles = the query points [V][n] and s->old_particles = the
 * current sample into a scratch f2v buffer, then
 *   lhvi_pbp_var_sum      out[v][j] = sum over the variable's edges of count * f2v[e][j]   (the log-belief at point j)
 *   lhvi_pbp_domain_grid  x[v][:] = uniform n-point grid on the domain (discrete: the states)
 *   lhvi_pbp_refine_grid  best[v] / best_val[v] = argmax point and value of logb[v][:]; x[v][:] := uniform grid on the
 *                         bracket around it (one step of the batched MAP search, EPBP.map EPBP:377-394 uses fminbound) */
int lhvi_pbp_var_sum(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* out, void* stream);
int lhvi_pbp_domain_grid(const lhvi_graph_t* g, const lhvi_pbp_t* s, double* x, void* stream);
int lhvi_pbp_refine_grid(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* logb, double* x, double* best, double* best_val,
                         void* stream);
/* test hook: y[i] = the f2v kernel's exp(x[i]) */
int lhvi_debug_exp(const double* x, double* y, int64_t n, void* stream);
/* test hook: y[i] = exp(x[i] + c[i]) through the accumulating form the f2v term loop uses (c = the per-point constant) */
int lhvi_debug_exp_acc(const double* x, const double* c, double* y, int64_t n, void* stream);
/* the same value through the floor form of the term loop (heavy and cq kernels: records pre-divided by the table step, floor by a
 * round-down addition, csrc/fastmath.hpp::exp_accumulate_floor) */
int lhvi_debug_exp_acc_floor(const double* x, const double* c, double* y, int64_t n, void* stream);
/* test hook: y[i] = log(x[i]), x > 0; which = 0 the table-driven log of the f2v epilogue, 1 the series log */
int lhvi_debug_log(const double* x, double* y, int64_t n, int32_t which, void* stream);

/* edge_class[e]: 0 = no message (observed target / alias edge), 1|2 = quadratic-family (continuous | discrete target),
 * 3 = generic potential, 4 = conditionally quadratic with two hidden partners (only with LHVI_PBP_CQ in s->flags; `s` may be
 * NULL otherwise -- it is read for flags, n and np only).  Static per (graph, evidence, potentials, particle counts); the
 * host turns it into the work lists above. */
int lhvi_pbp_classify(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, uint8_t* edge_class, void* stream);

/* uniq[v][j] = no i<j with particles[v][i] == particles[v][j]  (dict-key collapse, EPBP.py:236-242) */
int lhvi_pbp_uniq(const lhvi_graph_t* g, int32_t n, const double* particles, const int32_t* np, uint8_t* uniq, void* stream);
/* EPBP.message_rv_to_f + important_weight + log_message_balance: EPBP.py:156-174,204-215; HLBP.py:173-191,225-236 */
int lhvi_pbp_v2f(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, void* stream);
/* EPBP.message_f_to_rv at the new particles + integral points: EPBP.py:176-194,275-285; HLBP.py:193-215.
 * Up to four kernels on `stream`, one per work list of `s` (heavy_desc, light_desc, fast_edges, generic_edges); the SKIP
 * flags select among them.  With s->f2v_ticket set the call first resets those words on `stream` (a 32-byte memset). */
int lhvi_pbp_f2v(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, double* f2v, void* stream);
/* update_proposal (sites eta [E][2] in/out, q [V][2] in/out): EPBP.py:83-154; HLBP.py:100-171 */
int lhvi_pbp_proposal(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* eta, double* q, void* stream);
/* Sharded form of lhvi_pbp_proposal: `partial` updates the local sites and writes, per variable, the information-form sum
 * over its LOCAL edges ph[v] = (sum c/var, sum c*mu/var); `boundary_pack` writes every boundary variable's row into each of
 * its slots of the send buffer; after the all_to_all, lhvi_pbp_v2f reads the received rows through s->recv, and `finish`
 * forms q[v] from ph[v] plus the received sums in rank order -- together they equal lhvi_pbp_proposal on the whole graph. */
int lhvi_pbp_proposal_partial(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* eta, double* ph, void* stream);
int lhvi_pbp_proposal_finish(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* ph, double* q, void* stream);
int lhvi_pbp_boundary_pack(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, const double* ph, int32_t nb,
                           const int32_t* bvars, double* send, void* stream);
/* Reduce-to-owner form of the exchange (a boundary variable present on k ranks costs 2 (k - 1) rows instead of k (k - 1)): every
 * such variable has one owner among its ranks; the others send it their row, the owner adds the rows in ascending rank order
 * (its own at its position -- the order lhvi_pbp_v2f / _proposal_finish use for the all-to-all form, so both forms give the same
 * bits) and sends the total back.  This call is the owner's sum: item i adds the `width[i]` doubles at in + src_off[r],
 * r in [src_ptr[i], src_ptr[i+1]), and stores the total at out + dst_off[r], r in [dst_ptr[i], dst_ptr[i+1]).  Afterwards
 * lhvi_pbp_v2f / _proposal_finish run with LHVI_PBP_BOUNDARY_TOTALS: each boundary variable lists ONE row (brow_ptr / brow_off into
 * s->recv) holding the total over all ranks. */
int lhvi_pbp_boundary_reduce(int32_t n_items, const int32_t* width, const int32_t* src_ptr, const int64_t* src_off,
                             const int32_t* dst_ptr, const int64_t* dst_off, const double* in, double* out, void* stream);
/* initial_proposal: q=(0,5), sites (0, 5*deg): EPBP.py:72-81; HLBP.py:89-98 */
int lhvi_pbp_init(const lhvi_graph_t* g, const lhvi_pbp_t* s, double* eta, double* q, double* f2v, double* v2f, void* stream);
/* generate_sample with a counter-based device RNG: EPBP.py:61-70.  Philox4x32-10, key = seed, counter = (variable gid, block,
 * iteration); a block yields two normals by Box-Muller, particle j takes block (j & 31) | (j >> 6 << 5) and the cosine (bit 5 of j
 * clear) or the sine.  var_gid may be NULL (gid = local index).  Parity runs inject host particles instead. */
int lhvi_pbp_resample(const lhvi_graph_t* g, const lhvi_pbp_t* s, const int64_t* var_gid, uint64_t seed, uint32_t iteration, double* particles_out, void* stream);
/* message_f_to_rv(x, f, rv, sample) for explicit (edge, point) pairs: qedge [nq] edge ids, x [nq][npts], out [nq][npts].
 * HybridLBP.belief_rv_query (HLBP.py:313-317) sums these over a ground variable's factors. */
int lhvi_pbp_edge_points(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f,
                         int32_t nq, const int32_t* qedge, int32_t npts, const double* x, double* out, void* stream);
/* lhvi_pbp_resample followed by lhvi_pbp_uniq on the fresh particles, fused into one pass when n <= 64 */
int lhvi_pbp_resample_uniq(const lhvi_graph_t* g, const lhvi_pbp_t* s, const int64_t* var_gid, uint64_t seed, uint32_t iteration,
                           double* particles_out, uint8_t* uniq_out, void* stream);
/* belief_rv(x) = sum_f message_f_to_rv(x, f, rv, sample) at arbitrary points: EPBP.py:196-202; HLBP.py:313-317.
 * qvar [nq] variable ids, x [nq][npts], out [nq][npts]; uses s->particles as the partners' sample. */
int lhvi_pbp_belief_points(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f,
                           int32_t nq, const int32_t* qvar, int32_t npts, const double* x, double* out, void* stream);
/* map(rv) of nq query rows in ONE launch: EPBP.map EPBP.py:377-394, HybridLBP.map HLBP.py:405-424.  Every continuous row runs
 * scipy.optimize.fminbound (Brent's bounded minimiser, the reference's call with its defaults xtol = 1e-5, maxfun = 500) on
 * f(x) = -belief_rv(x) over the row's domain [dom_lo, dom_hi], decision for decision, in IEEE double without contraction; a
 * discrete row returns its first state with the largest belief.  row_var [nq]: the variable of g whose domain the row searches.
 * qptr == NULL: the row's belief is belief_rv of row_var[i] (its incident edges, count-weighted on a lifted graph).
 * qptr [nq + 1] / qedge / qmult: the row's belief is sum_k qmult[k] * message_f_to_rv(x, edge qedge[k]) for k in
 * [qptr[i], qptr[i+1]) -- a GROUND variable's factors on the lifted graph (belief_rv_query HLBP.py:313-317).
 * xout [nq] the minimiser, fout [nq] (optional) the log-belief there, nfev [nq] (optional) function evaluations used. */
/* The three per-variable steps of a sweep fused for hidden continuous variables with few particles (s->n <= 32: the particle
 * counts of the reference's demos): message_rv_to_f + log_message_balance (EPBP.py:165-174,204-215; HLBP.py:182-191), then
 * update_proposal (EPBP.py:83-154; HLBP.py:100-171), then generate_sample with the counter-based sampler + the first-occurrence
 * mask (EPBP.py:61-70) -- one pass over a variable's incident f -> v rows instead of three launches' worth.  Same bits as
 * lhvi_pbp_v2f + lhvi_pbp_proposal + lhvi_pbp_resample_uniq on those variables (the caller runs those three on the others).
 * desc: records of eight 32-bit words per variable -- 0 variable  1 incident edges (<= 64)  2 grid base in dom_val  3 T
 * 4-5 dom_lo  6-7 dom_hi (doubles) -- in three blocks: n16 variables with np <= 16 and T <= 32, then n32_t32 with 16 < np <= 32
 * and T <= 32, then n32_t64 with np <= 32 and 32 < T <= 64.  Reads s->particles / s->uniq / s->q (the current sample and
 * proposal), writes v2f rows, eta, q, particles_out (must not be s->particles) and uniq_out rows of the listed variables. */
int lhvi_pbp_var_fused(const lhvi_graph_t* g, const lhvi_pbp_t* s, const double* f2v, double* v2f, double* eta, double* q,
                       const int64_t* var_gid, uint64_t seed, uint32_t iteration, double* particles_out, uint8_t* uniq_out,
                       const int32_t* desc, int32_t n16, int32_t n32_t32, int32_t n32_t64, void* stream);
/* The normaliser of EPBP.belief for nq query rows in ONE launch: EPBP.py:325-328 calls scipy.integrate.quad on e ** belief_rv over
 * [lo, hi] (= the domain widened by 20 on both sides).  A thread per row runs QUADPACK's 21-point Gauss-Kronrod rule (dqk21 and
 * its error estimate) inside dqage's globally adaptive bisection (at most 50 intervals) until the summed error estimate meets
 * max(epsabs, epsrel |z|); scipy's quad (dqagse) adds an extrapolation step to the same rule and bisection, so both lie within
 * the tolerance of the integral (scipy's default request: 1.49e-8 for both).  Rows as for lhvi_pbp_map_brent.  z [nq] the integral
 * (NaN where e ** belief_rv overflowed: the reference raises OverflowError there), abserr [nq] (optional) the error estimate,
 * status [nq] (optional) 0 converged / 1 interval limit / 3 overflow. */
int lhvi_pbp_quad(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, int64_t nq,
                  const int32_t* row_var, const int64_t* qptr, const int32_t* qedge, const double* qmult, const double* lo,
                  const double* hi, double epsabs, double epsrel, double* z, double* abserr, int32_t* status, void* stream);
int lhvi_pbp_map_brent(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_pbp_t* s, const double* v2f, int64_t nq,
                       const int32_t* row_var, const int64_t* qptr, const int32_t* qedge, const double* qmult, double xtol,
                       int32_t maxfun, double* xout, double* fout, int32_t* nfev, void* stream);

/* ---- owner-computes exchange of the edge-sharded sweep (lhvi/dist.py::OwnerRunner; csrc/halo.hip) ----------------------------
 * The reference is one process; these two calls move its `message[(rv, f)]` tables (EPBPLogVersion.py:250-258) and proposals
 * `q[rv]` (EPBPLogVersion.py:83-101) of the factors cut by a variable partition between ranks: pack fills the send buffer of the
 * sweep's one all_to_all, unpack scatters the receive buffer.  Row i: `row_width[i]` doubles of v2f row `row_edge[i]` (v2f is
 * [E][n]) at element offset `row_off[i]` of the buffer; proposal i: the two doubles of q row `q_var[i]` at `q_off[i]`. */
int lhvi_pbp_halo_pack(const double* v2f, int32_t n, int32_t n_rows, const int32_t* row_edge, const int64_t* row_off,
                       const int32_t* row_width, const double* q, int32_t n_q, const int32_t* q_var, const int64_t* q_off,
                       double* out, void* stream);
int lhvi_pbp_halo_unpack(const double* in, int32_t n, int32_t n_rows, const int32_t* row_edge, const int64_t* row_off,
                         const int32_t* row_width, double* v2f, int32_t n_q, const int32_t* q_var, const int64_t* q_off,
                         double* q, void* stream);

/* ---- Mixture variational inference (VarInference.py / LiftedVarInference.py) --------------------- */

typedef struct lhvi_vi {
    int32_t K;                  /* mixture components */
    int32_t T;                  /* Gauss-Hermite points */
    int32_t Dmax;               /* max #states of a discrete variable (row stride of eta_d) */
    int32_t quirks;             /* 1 = reproduce VarInference.py:147-150 */
    const double* gh_x;         /* [T] hermgauss nodes */
    const double* gh_w;         /* [T] weights / sqrt(pi) */
    const double* w;            /* [K] softmax(w_tau) */
    const double* eta_c;        /* [V][K][2] (mu, var) for continuous hidden variables */
    const double* eta_d;        /* [V][K][Dmax] category probabilities for discrete hidden variables */
    const double* obs_var;      /* [V] or NULL.  C2FVarInference.py:120-136,253-261: obs_var[v] > 0 makes the evidence cluster v
                                 * a Gaussian observation N(var_value[v], obs_var[v]) -- T quadrature nodes in every expectation,
                                 * its pdf in every belief, no parameters; 0 = exact evidence (or not evidence at all) */
    const double* var_N;        /* [V] or NULL: rv.N = number of incident ground factors (the sum of the row's edge_count on a
                                 * lifted graph, LiftedVarInference.py:64-67).  NULL: every (variable, k) thread sums its row itself,
                                 * which serialises on the template variables of a relational model (thousands of entries) */
    /* (ABI 9) the caller's split of the factors among the kernels of expectation() (VarInference.py:40-55), or NULL: a permutation
     * of the factor ids in six segments, in this order --
     *   n_cc    pairwise factors over two distinct continuous / observed variables with a Gaussian / quadratic / linear-Gaussian /
     *           XY potential (thread per (factor, k), per-axis pdf tables in registers);
     *   n_tiny  other factors of arity <= 3 whose grid has at most LHVI_VI_TINY_NODES nodes, K <= 2 (thread per (factor, k) walking
     *           the grid nodes and the points of the pinned expectations as one list; needs edge_axis and tiny_par_words);
     *   n_grp3  other factors of arity <= 3 whose axis lengths sum to <= LHVI_VI_GROUP_SLOTS with K * that <= LHVI_VI_GROUP_COMP
     *           (8 lanes per (factor, k), per-axis tables in LDS);  n_grp6: the same for arity 4 .. LHVI_MAX_ARITY;
     *   n_rest3 / n_rest6  whatever fits neither (thread per (factor, k), arity <= 3 / 4 .. LHVI_MAX_ARITY).
     * NULL: every kernel classifies the factors itself (thread-per-factor kernels only). */
    const int32_t* fac_list;
    int32_t n_cc, n_grp3, n_grp6, n_rest3, n_rest6;
    int32_t n_tiny;            /* (sits between n_cc and n_grp3 in the list) */
    int32_t tiny_par_words;    /* pots.off[P], the length of pots.param in doubles: the tiny-grid kernel keeps the parameter rows (an MLN
                                * formula's program) in LDS; n_tiny > 0 needs 0 < tiny_par_words <= 3072 */
    /* (ABI 9) [E][4] or NULL: per edge {variable, axis length | hidden << 16 | continuous << 17 | Gaussian observation << 18,
     * state index of the observed value (0 unless observed and discrete), offset of the variable's states in dom_val} -- the shape
     * of the factor's quadrature grid, which
     * depends on the graph and the evidence pattern only; NULL: the group kernels derive it per (factor, k) */
    const int32_t* edge_axis;
} lhvi_vi_t;
#define LHVI_VI_GROUP_SLOTS 24
#define LHVI_VI_GROUP_COMP 48
#define LHVI_VI_TINY_NODES 32

/* state of the optimiser for lhvi_vi_adam_run: the arrays ADAM_update (VarInference.py:249-287) reads and writes.  w, eta_c and
 * eta_d must be the arrays the lhvi_vi_t passed alongside points to (the step must see what it updates). */
typedef struct lhvi_vi_opt {
    double* w_tau;              /* [K] mixture logits */
    double* w;                  /* [K] softmax(w_tau) */
    double* eta_c;              /* [V][K][2] */
    double* tau_d;              /* [V][K][Dmax] category logits */
    double* eta_d;              /* [V][K][Dmax] softmax(tau_d) over each hidden discrete variable's states */
    double *m_w, *s_w, *m_c, *s_c, *m_d, *s_d;   /* ADAM moments, shaped like w_tau / eta_c / tau_d */
    double *g_w, *g_c, *g_d;    /* gradient scratch, same shapes */
    double* fe;                 /* [1] scratch for the free energy of passes that are not logged */
    double lr, b1, b2, eps;     /* VarInference.py:252-254: 0.9, 0.999, 1e-8 */
    double var_min;             /* var_threshold: variances are clipped from below (VarInference.py:11,277) */
    int32_t t;                  /* updates done before this call (bias correction uses t + i + 1) */
} lhvi_vi_opt_t;

/* gradient_w_tau / gradient_mu_var / gradient_category_tau / free_energy: VarInference.py:57-195,
 * LiftedVarInference.py:59-199.  Outputs: g_w [K] (already softmax-projected), g_c [V][K][2],
 * g_d [V][K][Dmax] (already projected), fe [1].  ws: workspace of lhvi_vi_workspace_bytes() bytes. */
size_t lhvi_vi_workspace_bytes(const lhvi_graph_t* g, const lhvi_vi_t* p);
int lhvi_vi_grad(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_vi_t* p,
                 double* g_w, double* g_c, double* g_d, double* fe, void* ws, size_t ws_bytes, void* stream);
/* `iterations` rounds of ADAM_update (VarInference.py:249-300) enqueued back to back, no host work in between: per round one
 * lhvi_vi_grad and ONE update launch for all three parameter arrays (ADAM step, variance clip, both softmaxes).  fe_log: device
 * [iterations] or NULL -- fe_log[i] = the free energy after update i + 1, which is what the next round's gradient pass computes
 * anyway (one extra pass after the last update).  Results equal the per-array calls below bit for bit. */
int lhvi_vi_adam_run(const lhvi_graph_t* g, const lhvi_pots_t* pots, const lhvi_vi_t* p, const lhvi_vi_opt_t* o, int32_t iterations,
                     double* fe_log, void* ws, size_t ws_bytes, void* stream);
/* ADAM_update body: VarInference.py:255-287.  theta/m/s/g are flat arrays of `count` doubles;
 * clip_stride>0 clamps every element with index % clip_stride == clip_stride-1 to >= clip_min (variances). */
int lhvi_adam_step(double* theta, double* m, double* s, const double* g, int64_t count, int32_t t,
                   double lr, double b1, double b2, double eps, int32_t clip_stride, double clip_min, void* stream);
/* softmax over rows of `cols` valid entries (stride `stride`): VarInference.py:32-38 */
int lhvi_softmax_rows(const double* tau, double* out, int64_t rows, int32_t cols, int32_t stride, void* stream);

/* ---- Colour refinement (CompressedGraphWithObs.py / CompressedGraphSorted.py) --------------------
 * One half-round each; colours are dense int32 ids (the rank of the item's 64-bit signature fingerprint among the distinct
 * fingerprints).  ws: lhvi_color_workspace_bytes(g) bytes.
 * result: device int32[4] = {number of colours, fingerprint collision (two items agree on the first 64-bit fingerprint and
 * differ on the second: retry), table overflow (method 0 only: repeat the call with method 1), 0}.
 * method 0: the signature kernel finds-or-inserts each fingerprint in a 1 M-slot hash table and only the distinct keys are
 *           sorted (the answer has few of them: ~30 k on a 10 M-edge relational graph);
 * method 1: radix sort of all items' fingerprints.  Both give the same colours. */
#define LHVI_COLOR_HASH 0
#define LHVI_COLOR_SORT 1
size_t lhvi_color_workspace_bytes(const lhvi_graph_t* g);
/* SuperF.split_by_structure CGWO.py:152-175: signature = (old colour, nb rv colours [sorted iff symmetric]) */
int lhvi_color_refine_factors(const lhvi_graph_t* g, const uint8_t* pot_symmetric_per_factor, const int32_t* rv_color,
                              const int32_t* f_color, int32_t* f_color_out, int32_t* result,
                              void* ws, size_t ws_bytes, int32_t method, void* stream);
/* SuperRV.split_by_structure CGWO.py:47-76: signature = (old colour, sorted multiset of nb factor colours) */
int lhvi_color_refine_rvs(const lhvi_graph_t* g, const int32_t* f_color, const int32_t* rv_color,
                          int32_t* rv_color_out, int32_t* result, void* ws, size_t ws_bytes, int32_t method, void* stream);

/* the representative of every cluster: first_out[c] = smallest i with color[i] == c, or n when colour c has no member
 * (SuperRV.update_nb CompressedGraphWithObs.py:41-45 and SuperF.update_nb :148-150 read the neighbourhood of
 * `next(iter(self.rvs))` / `next(iter(self.factors))` -- any member; this build always takes the first).  color [n] on the
 * device, values in [0, n_colors). */
int lhvi_color_first_members(const int32_t* color, int32_t n, int32_t n_colors, int32_t* first_out, void* stream);

/* sums_out[s] = values[offsets[s]] + values[offsets[s] + 1] + ... in index order (one running sum per segment, like
 * SuperRV.get_value CompressedGraphWithObs.py:24-28 over a cluster's observed members); offsets [n_segments + 1], ascending. */
int lhvi_color_segment_sums(const double* values, const int64_t* offsets, int32_t n_segments, double* sums_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LHVI_H */
