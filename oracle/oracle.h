/*
 * oracle.h - CPU restatement (TEST INFRASTRUCTURE, not product) of the reference's MIQP solve path.
 *
 * The arithmetic of the reference path lives in IBM ILOG CPLEX 12.10 + OPL (util/deps.bzl:69-95),
 * a proprietary dependency that is absent from /root/reference and from this image, so the
 * reference cannot be built here ("unbuildable": there is no oracle/_ref).  This oracle restates
 *   (1) the OPL model (the .mod files of cplexmodel/) row by row   (raw_model.c - sizes + constraint evaluator),
 *   (2) the same model in disjunctive form and
 *   (3) the published algorithm class CPLEX applies to it: branch and bound over the binaries with a convex QP
 *       relaxation per node (solve.c: dense-row Riccati primal-dual interior point, array open list with dives;
 *       the branching order is the one of the device solver).  inst.c reads ModelParameters records and OPL .dat files.
 * It is pinned against the reference's own known answers (tests/test_oracle_golden.py):
 *   K1 sizes 12361/1240/340/29834, K2 objective 9.57603, K3 solution vector
 *   (test/cplex_wrapper_test.cc:283-456, :857-876), K5 cplexmodel.dat + modelRun.txt feasibility.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library.
 */
#ifndef MIQP_ORACLE_H
#define MIQP_ORACLE_H

#include "../include/miqp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXC 4
#define ORC_NZMAX (8 * ORC_MAXC)

typedef struct oinst {
  int N, C, R, E, O, L, K, NP; /* NP = number of car pairs */
  double ts, vmin, vmax, amin, amax, jmin, jmax, max_slack, w_slack, w_slack_obs, vm;
  double gap, tilim;
  double *safety, *safety_slack; /* [N] */
  double *W;                     /* [C*8] px vx ax py vy ay jx jy */
  double *wb, *rad;              /* [C] */
  double *x0;                    /* [C*6] */
  double *ref;                   /* [C*N*6] */
  double *acc_lim, *jerk_lim;    /* [C*R*4] minx maxx miny maxy */
  int *init_region;              /* [C] 1-based */
  int *possible;                 /* [C*R] */
  double *frac;                  /* [R*4] */
  double *poly[6];               /* [R*3] SINT_UB SINT_LB COSS_UB COSS_LB KAPPA_MAX KAPPA_MIN */
  int *env_off;                  /* [E+1] edge offsets */
  double *env_edges;             /* [4*env_off[E]] x1 y1 x2 y2 */
  double *obs_edges;             /* [O*N*L*4] */
  int *obs_soft;                 /* [O] */
} oinst;

enum { OP_SINT_UB = 0, OP_SINT_LB, OP_COSS_UB, OP_COSS_LB, OP_KMAX, OP_KMIN };

/* ---- construction ---- */
oinst* orc_from_params(const miqp_model_params_c* p, int round_decimals /* <0: none */);
oinst* orc_from_dat(const char* path, char* err, int errlen);
void orc_free(oinst* I);

/* ---- raw OPL model (raw_model.c) ---- */
typedef struct { int rows, bin, cont, nnz; } orc_sizes;
orc_sizes orc_raw_sizes(const oinst* I);
/* max violation of every raw big-M row for a complete assignment (tolerance check of golden vectors);
 * returns max violation, writes the objective value */
double orc_raw_eval(const oinst* I, const miqp_raw_results_c* r, const double* slack_real, double* objective,
                    char* worst, int worstlen);

/* ---- solve ---- */
typedef struct {
  double gap;        /* relative MIP gap, <0: take the instance's */
  double time_limit; /* seconds, <=0: take the instance's */
  long long max_nodes;
  int verbose;
} orc_opts;

int orc_solve(const oinst* I, const orc_opts* o, miqp_raw_results_c* res, miqp_solution_properties_c* props);

/* continuous QP with every binary of `fixed` asserted (all alternatives whose binary is 0 are enforced);
 * used to pin the QP machinery against K3.  Returns 0 when feasible. */
int orc_solve_fixed(const oinst* I, const miqp_raw_results_c* fixed, miqp_raw_results_c* res, double* objective,
                    int* iters);

/* result buffers */
miqp_raw_results_c* orc_results_alloc(int C, int N, int R, int E, int O, int L);
void orc_results_free(miqp_raw_results_c* r);

#ifdef __cplusplus
}
#endif
#endif
