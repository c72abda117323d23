/* oracle.h -- shared structs of the CPU oracle (TEST INFRASTRUCTURE ONLY; see ../README.md). */
#ifndef LHVI_ORACLE_H
#define LHVI_ORACLE_H
#include <stdint.h>

#define MAX_ARITY 6

#define POT_GENERIC 0
#define POT_TABLE 1
#define POT_GAUSSIAN 2
#define POT_QUADRATIC 3
#define POT_HYBRID_QUADRATIC 4
#define POT_LINEAR_GAUSSIAN 5
#define POT_X2 6
#define POT_XY 7
#define POT_MLN 8
#define POT_MLN_HARD 9
#define POT_IMAGE_NODE 10
#define POT_IMAGE_EDGE 11

typedef struct {
    int32_t V, F, E, nnz;
    const int32_t *fac_ptr, *edge_var, *edge_fac, *edge_canon, *var_ptr, *var_edge;
    const double *edge_count; /* NULL on a ground graph */
    const int32_t *fac_pot;
    const double *var_value;
    const int32_t *pot_kind, *pot_off;
    const double *pot_param;
    /* fields below are used by the particle / variational oracles only */
    const int32_t *var_dom;
    const double *var_mult, *fac_mult;
    const int32_t *dom_cont;
    const double *dom_lo, *dom_hi;
    const int32_t *dom_ptr;
    const double *dom_val;
} ograph_t;

typedef struct {
    int32_t n, T;
    uint32_t flags; /* 1 = EP proposal, 2 = EPBP discrete importance-weight quirk */
    double var_threshold, max_log_value;
    const double *particles, *old_particles;
    const int32_t *np;
    const uint8_t *uniq;
    const double *q;
} opbp_t;

double oracle_potential(int kind, const double *par, int arity, const double *x, const int *idx);

#endif
