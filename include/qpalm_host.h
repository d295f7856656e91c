/*
 * qpalm_host.h -- the reference's public C API (include/qpalm.h:43-138) and workspace types
 * (include/types.h) re-declared for the gfx950 backend, i.e. the third branch next to
 * USE_LADEL / USE_CHOLMOD of include/types.h:17-32.  Field ORDER and types of QPALMSettings,
 * QPALMInfo, QPALMData, QPALMScaling, QPALMSolution, QPALMSolver and QPALMWorkspace equal the
 * reference's (ABI witnessed by interfaces/python/qpalm.py:15-187) so that front-ends that read
 * workspace fields directly (interfaces/mex/qpalm_mex.c:303-351) keep working.
 *
 * The host stays in C; every numerical step runs in the HIP kernels behind include/qpalm_gfx950.h.
 * After each API call the host mirrors (x, y, Ax, Qx, sigma, ..., info, solution) are refreshed
 * from HBM.
 */
#ifndef QPALM_HOST_H
#define QPALM_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double  c_float; /* include/global_opts.h:61 */
typedef int64_t c_int;   /* include/global_opts.h:31-39 with -DDLONG (CMakeLists.txt:53) */

#define QPALM_SOLVED (1)
#define QPALM_DUAL_TERMINATED (2)
#define QPALM_MAX_ITER_REACHED (-2)
#define QPALM_PRIMAL_INFEASIBLE (-3)
#define QPALM_DUAL_INFEASIBLE (-4)
#define QPALM_TIME_LIMIT_REACHED (-5)
#define QPALM_UNSOLVED (-10)
#define QPALM_ERROR (0)
#define QPALM_NULL 0
#define QPALM_INFTY ((c_float)1e20)
#define FACTORIZE_KKT 0
#define FACTORIZE_SCHUR 1
#define FACTORIZE_KKT_OR_SCHUR 2
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

/* backend typedef block (the reference's include/types.h:17-32 for this backend).
 * solver_sparse keeps cholmod_sparse's member order (interfaces/python/qpalm.py:29-44). */
typedef struct {
  size_t nrow, ncol, nzmax;
  void *p, *i, *nz, *x, *z; /* p, i: c_int[] ; x: c_float[] */
  int stype, itype, xtype, dtype, sorted, packed;
} solver_sparse;
typedef struct {
  size_t nrow, ncol, nzmax, d;
  void *x, *z;
  int xtype, dtype;
} solver_dense;
typedef struct qpalm_gfx950_state solver_factor; /* opaque: device context + resident batch */
typedef void solver_symbolics;
typedef struct { int status; } solver_common;

typedef struct { c_float x; size_t i; } array_element;

typedef struct { c_float *x, *y; } QPALMSolution;
typedef struct QPALM_TIMER QPALMTimer;
typedef struct { c_float *D, *Dinv, *E, *Einv, c, cinv; } QPALMScaling;

typedef struct {
  c_int iter, iter_out;
  char status[32];
  c_int status_val;
  c_float pri_res_norm, dua_res_norm, dua2_res_norm, objective, dual_objective;
  c_float setup_time, solve_time, run_time;
} QPALMInfo;

typedef struct {
  size_t n, m;
  solver_sparse *Q, *A;
  c_float *q, c, *bmin, *bmax;
} QPALMData;

typedef struct {
  c_int max_iter, inner_max_iter;
  c_float eps_abs, eps_rel, eps_abs_in, eps_rel_in, rho, eps_prim_inf, eps_dual_inf, theta, delta, sigma_max, sigma_init;
  c_int proximal;
  c_float gamma_init, gamma_upd, gamma_max;
  c_int scaling, nonconvex, verbose, print_iter, warm_start, reset_newton_iter, enable_dual_termination;
  c_float dual_objective_limit, time_limit;
  c_int ordering, factorization_method, max_rank_update;
  c_float max_rank_update_fraction;
} QPALMSettings;

typedef struct {
  c_int factorization_method;
  solver_sparse *kkt, *kkt_full, *At;
  c_int *first_row_A;
  c_float *first_elem_A;
  solver_factor *LD;
  solver_symbolics *sym;
  solver_factor *LD_Q;
  solver_symbolics *sym_Q;
  solver_dense *E_temp, *D_temp, *neg_dphi, *rhs_kkt, *sol_kkt, *d, *Ad, *Qd, *yh, *Atyh;
  c_int first_factorization, reset_newton;
  c_int *active_constraints, *active_constraints_old;
  c_int nb_active_constraints;
  c_int *enter;
  c_int nb_enter;
  c_int *leave;
  c_int nb_leave;
  solver_dense *At_scale;
  solver_sparse *At_sqrt_sigma;
} QPALMSolver;

typedef struct {
  QPALMData *data;
  c_float *x, *y, *Ax, *Qx, *Aty, *x_prev;
  c_int initialized;
  c_float *temp_m, *temp_n, *sigma, *sigma_inv;
  c_float sqrt_sigma_max;
  c_int nb_sigma_changed;
  c_float gamma;
  c_int gamma_maxed;
  c_float *Axys, *z, *pri_res, *pri_res_in, *yh, *Atyh, *df, *x0, *xx0, *dphi, *neg_dphi, *dphi_prev, *d;
  c_float tau;
  c_float *Qd, *Ad, *sqrt_sigma;
  c_float sqrt_delta, eta, beta;
  c_float *delta, *alpha, *temp_2m, *delta2, *delta_alpha;
  array_element *s;
  c_int *index_L, *index_P, *index_J;
  c_float eps_pri, eps_dua, eps_dua_in, eps_abs_in, eps_rel_in;
  c_float *delta_y, *Atdelta_y;
  c_float *delta_x, *Qdelta_x, *Adelta_x;
  c_float *D_temp, *E_temp;
  QPALMSolver *solver;
  QPALMSettings *settings;
  QPALMScaling *scaling;
  QPALMSolution *solution;
  QPALMInfo *info;
  QPALMTimer *timer;
} QPALMWorkspace;

/* ---- include/qpalm.h:43-138 ---------------------------------------------------------------- */
void qpalm_set_default_settings(QPALMSettings *settings);
QPALMWorkspace *qpalm_setup(const QPALMData *data, const QPALMSettings *settings);
void qpalm_warm_start(QPALMWorkspace *work, c_float *x_warm_start, c_float *y_warm_start);
void qpalm_solve(QPALMWorkspace *work);
void qpalm_update_settings(QPALMWorkspace *work, const QPALMSettings *settings);
void qpalm_update_bounds(QPALMWorkspace *work, const c_float *bmin, const c_float *bmax);
void qpalm_update_q(QPALMWorkspace *work, const c_float *q);
void qpalm_cleanup(QPALMWorkspace *work);

/* ---- include/solver_interface.h (same names and argument meaning) --------------------------- */
void mat_vec(solver_sparse *A, solver_dense *x, solver_dense *y, solver_common *c);
void mat_tpose_vec(solver_sparse *A, solver_dense *x, solver_dense *y, solver_common *c);
void mat_inf_norm_cols(solver_sparse *M, c_float *E);
void mat_inf_norm_rows(solver_sparse *M, c_float *E);
void qpalm_set_factorization_method(QPALMWorkspace *work, solver_common *c);
void ldlchol(solver_sparse *M, QPALMWorkspace *work, solver_common *c);
void ldlcholQAtsigmaA(QPALMWorkspace *work, solver_common *c);
void ldlupdate_entering_constraints(QPALMWorkspace *work, solver_common *c);
void ldldowndate_leaving_constraints(QPALMWorkspace *work, solver_common *c);
void ldlupdate_sigma_changed(QPALMWorkspace *work, solver_common *c);
void ldlsolveLD_neg_dphi(QPALMWorkspace *work, solver_common *c);
/* the KKT set (solver_interface.h:82,89,97,106,126; the USE_LADEL branch of the reference): workspaces whose
 * settings->factorization_method was FACTORIZE_KKT at qpalm_setup.  On this backend the KKT matrix is a dense (n+m) x (n+m)
 * panel on the device, so qpalm_form_kkt / qpalm_reform_kkt both (re)build it for solver->active_constraints; the
 * factorisation that newton.c:36,44 asks LADEL for right after them is qpalm_kkt_factorize (no reference name: LADEL's
 * ladel_factorize_*_with_diag is called directly there).  kkt_solve fills work->d (and the device's sol_kkt). */
void qpalm_form_kkt(QPALMWorkspace *work);
void qpalm_reform_kkt(QPALMWorkspace *work);
void qpalm_kkt_factorize(QPALMWorkspace *work);
void kkt_update_entering_constraints(QPALMWorkspace *work, solver_common *c);
void kkt_update_leaving_constraints(QPALMWorkspace *work, solver_common *c);
void kkt_solve(QPALMWorkspace *work, solver_common *c);

/* helpers for callers that built their matrices with cholmod_allocate_sparse / allocate_dense */
solver_sparse *qpalm_sparse_alloc(size_t nrow, size_t ncol, size_t nzmax, int stype);
void qpalm_sparse_free(solver_sparse **A);
const char *qpalm_backend_error(void);

#ifdef __cplusplus
}
#endif
#endif
