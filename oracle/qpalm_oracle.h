/*
 * qpalm_oracle.h -- CPU restatement of the QPALM semismooth-Newton path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle of the repository: a plain-C restatement of the reference's
 * CHOLMOD/Schur path (Benny44/QPALM src/qpalm.c, iteration.c, newton.c, linesearch.c,
 * termination.c, lin_alg.c, scaling.c, solver_interface.c) on top of a dense, natural-order,
 * no-pivot LDL^T (what CHOLMOD is configured to produce: solver_interface.c:523-541).
 *
 * It is NOT product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (qpalm_amd/, include/qpalm_gfx950.h) never links or calls it.
 *
 * Pinning: every golden vector the reference's own tests hold for this path is checked in
 * tests/test_oracle_golden.py (fixtures in tests/golden/reference_tests.json).  The reference
 * itself cannot be built here (CHOLMOD/LADEL submodules are empty), so update/downdate,
 * per-iteration iterates and the line-search step have no reference-side pin: "parity unpinned"
 * for those.
 */
#ifndef QPALM_ORACLE_H
#define QPALM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t oq_int;   /* c_int is 64-bit in every shipped configuration (CMakeLists.txt:53) */
typedef double  oq_float; /* c_float (include/global_opts.h:61) */

/* status codes: include/constants.h:30-37 */
#define OQ_SOLVED 1
#define OQ_DUAL_TERMINATED 2
#define OQ_MAX_ITER_REACHED (-2)
#define OQ_PRIMAL_INFEASIBLE (-3)
#define OQ_DUAL_INFEASIBLE (-4)
#define OQ_TIME_LIMIT_REACHED (-5)
#define OQ_UNSOLVED (-10)
#define OQ_ERROR 0
#define OQ_INFTY 1e20

/* field order == QPALMSettings, include/types.h:119-150 */
typedef struct {
  oq_int   max_iter;
  oq_int   inner_max_iter;
  oq_float eps_abs;
  oq_float eps_rel;
  oq_float eps_abs_in;
  oq_float eps_rel_in;
  oq_float rho;
  oq_float eps_prim_inf;
  oq_float eps_dual_inf;
  oq_float theta;
  oq_float delta;
  oq_float sigma_max;
  oq_float sigma_init;
  oq_int   proximal;
  oq_float gamma_init;
  oq_float gamma_upd;
  oq_float gamma_max;
  oq_int   scaling;
  oq_int   nonconvex;
  oq_int   verbose;
  oq_int   print_iter;
  oq_int   warm_start;
  oq_int   reset_newton_iter;
  oq_int   enable_dual_termination;
  oq_float dual_objective_limit;
  oq_float time_limit;
  oq_int   ordering;
  oq_int   factorization_method;
  oq_int   max_rank_update;
  oq_float max_rank_update_fraction;
} oq_settings;

/* field order == QPALMInfo with PROFILING, include/types.h:76-95 */
typedef struct {
  oq_int   iter;
  oq_int   iter_out;
  char     status[32];
  oq_int   status_val;
  oq_float pri_res_norm;
  oq_float dua_res_norm;
  oq_float dua2_res_norm;
  oq_float objective;
  oq_float dual_objective;
  oq_float setup_time;
  oq_float solve_time;
  oq_float run_time;
} oq_info;

/* CSC matrix; stype 0 = unsymmetric, -1 = symmetric with only the lower triangle used
 * (cholmod_sparse semantics at the reference's call sites). */
typedef struct {
  oq_int    nrow, ncol, nzmax;
  oq_int   *p, *i;
  oq_float *x;
  int       stype;
} oq_sparse;

typedef struct oq_workspace oq_workspace;

/* per-iteration trace buffers (caller-owned, may be NULL).  One record per executed loop
 * iteration that did not terminate (src/qpalm.c:484 loop body). */
typedef struct {
  oq_int    cap;       /* capacity in records */
  oq_int    len;       /* records written */
  oq_int   *kind;      /* 0 newton step, 1 outer update, 2 forced outer update (inner_max_iter) */
  oq_int   *fact;      /* 0 none, 1 refactor Q+A'SA, 2 rank update/downdate, 3 factor Q only */
  oq_int   *nb_active, *nb_enter, *nb_leave;
  oq_float *tau, *gamma, *pri_res_norm, *dua_res_norm, *dua2_res_norm;
  oq_float *x;         /* cap*n, scaled iterate after the iteration */
  oq_float *y;         /* cap*m */
  oq_float *d;         /* cap*n, Newton direction (newton iterations) */
  oq_int   *active;    /* cap*m, solver->active_constraints after the iteration */
} oq_trace;

/* ---- API mirroring include/qpalm.h:43-138 ------------------------------------------------ */
void          oq_set_default_settings(oq_settings *s);
oq_workspace *oq_setup(oq_int n, oq_int m,
                       const oq_int *Qp, const oq_int *Qi, const oq_float *Qx,
                       const oq_int *Ap, const oq_int *Ai, const oq_float *Ax,
                       const oq_float *q, oq_float c,
                       const oq_float *bmin, const oq_float *bmax,
                       const oq_settings *settings);
void oq_warm_start(oq_workspace *w, const oq_float *x_ws, const oq_float *y_ws);
void oq_solve(oq_workspace *w);
void oq_update_settings(oq_workspace *w, const oq_settings *s);
void oq_update_bounds(oq_workspace *w, const oq_float *bmin, const oq_float *bmax);
void oq_update_q(oq_workspace *w, const oq_float *q);
void oq_cleanup(oq_workspace *w);
void oq_set_trace(oq_workspace *w, oq_trace *t);

/* accessors (the reference's tests read QPALMWorkspace fields directly) */
const oq_info *oq_get_info(const oq_workspace *w);
const oq_float *oq_get_solution_x(const oq_workspace *w);
const oq_float *oq_get_solution_y(const oq_workspace *w);
oq_float *oq_get_vec(oq_workspace *w, const char *name, oq_int *len); /* "x","y","Ax",... */
oq_int   *oq_get_ivec(oq_workspace *w, const char *name, oq_int *len);/* "active","enter","leave" */
oq_float  oq_get_scalar(const oq_workspace *w, const char *name);     /* "gamma","tau","c",... */
void      oq_set_scalar(oq_workspace *w, const char *name, oq_float v);
oq_int    oq_get_counter(const oq_workspace *w, const char *name);    /* "n_refactor",... */
const oq_settings *oq_get_settings(const oq_workspace *w);
void      oq_get_matrix(oq_workspace *w, const char *name, oq_int *nrow, oq_int *ncol,
                        oq_int **p, oq_int **i, oq_float **x);       /* "A","Q","At_sqrt_sigma" */
const oq_float *oq_get_factor(const oq_workspace *w, const oq_float **D, oq_int *ld);
int       oq_set_perm(oq_workspace *w, const oq_int *perm, oq_int n); /* sparse-storage mode: factorise P H P' (perm[new] = old); 0 = accepted */
oq_int    oq_sparse_levels(const oq_workspace *w);                    /* height of the elimination tree after the first factorisation */
/* KKT path, operation by operation (solver_interface.h:82-126; workspaces set up with factorization_method = FACTORIZE_KKT) */
void oq_kkt_form_and_factor(oq_workspace *w);
void oq_kkt_update_entering_constraints(oq_workspace *w);
void oq_kkt_update_leaving_constraints(oq_workspace *w);
void oq_kkt_solve(oq_workspace *w);
const oq_float *oq_get_kkt_factor(const oq_workspace *w, const oq_float **D, oq_int *ld);

/* ---- boundary functions, include/solver_interface.h ------------------------------------- */
void oq_mat_vec(const oq_sparse *A, const oq_float *x, oq_float *y);       /* y may alias x */
void oq_mat_tpose_vec(const oq_sparse *A, const oq_float *x, oq_float *y); /* y may alias x */
void oq_mat_inf_norm_cols(const oq_sparse *M, oq_float *E);
void oq_mat_inf_norm_rows(const oq_sparse *M, oq_float *E);
void oq_ldlchol(const oq_sparse *M, oq_workspace *w);
void oq_ldlcholQAtsigmaA(oq_workspace *w);
void oq_ldlupdate_entering_constraints(oq_workspace *w);
void oq_ldldowndate_leaving_constraints(oq_workspace *w);
void oq_ldlupdate_sigma_changed(oq_workspace *w);
void oq_ldlsolveLD_neg_dphi(oq_workspace *w);

/* ---- algorithm steps (exposed so that tests can drive them one by one) ------------------- */
void     oq_compute_residuals(oq_workspace *w);
oq_int   oq_check_termination(oq_workspace *w);
void     oq_set_active_constraints(oq_workspace *w);
void     oq_set_entering_leaving_constraints(oq_workspace *w);
void     oq_newton_set_direction(oq_workspace *w);
oq_float oq_exact_linesearch(oq_workspace *w);
void     oq_update_primal_iterate(oq_workspace *w);
void     oq_update_sigma(oq_workspace *w);
void     oq_scale_data(oq_workspace *w);

/* ---- dense LDL^T kernels on raw arrays (column-major, unit lower L, ld = leading dim) ---- */
/* H (lower triangle, column-major, ld) is overwritten by L (strict lower) ; D gets the pivots. */
int  oq_rand_sequence(unsigned int seed, int count, int *out); /* restated glibc rand() (nonconvex.c:42, B9) */
void oq_dense_ldl_factor(oq_int n, oq_float *H, oq_int ld, oq_float *D);
void oq_dense_ldl_solve(oq_int n, const oq_float *L, oq_int ld, const oq_float *D, oq_float *b);
/* LDL' <- LDL' + sign * w w'  (w is destroyed), Davis & Hager method C1 as used by CHOLMOD */
void oq_dense_ldl_rank1(oq_int n, oq_float *L, oq_int ld, oq_float *D, oq_float *w, int update);

/* ---- lin_alg.c known-answer surface ------------------------------------------------------ */
oq_float oq_vec_prod(const oq_float *a, const oq_float *b, size_t n);
oq_float oq_vec_norm_inf(const oq_float *a, size_t n);
void oq_vec_set_scalar(oq_float *a, oq_float sc, size_t n);
void oq_vec_self_mult_scalar(oq_float *a, oq_float sc, size_t n);
void oq_vec_add_scaled(const oq_float *a, const oq_float *b, oq_float *c, oq_float sc, size_t n);
void oq_vec_mult_add_scaled(oq_float *a, const oq_float *b, oq_float sc1, oq_float sc2, size_t n);
void oq_vec_ew_recipr(const oq_float *a, oq_float *b, size_t n);
void oq_vec_ew_max_vec(const oq_float *a, const oq_float *b, oq_float *c, size_t n);
void oq_vec_ew_min_vec(const oq_float *a, const oq_float *b, oq_float *c, size_t n);
void oq_vec_ew_mid_vec(const oq_float *a, const oq_float *lo, const oq_float *hi, oq_float *c, size_t n);
void oq_vec_ew_prod(const oq_float *a, const oq_float *b, oq_float *c, size_t n);
void oq_vec_ew_div(const oq_float *a, const oq_float *b, oq_float *c, size_t n);
void oq_vec_ew_sqrt(const oq_float *a, oq_float *b, size_t n);

#ifdef __cplusplus
}
#endif
#endif
