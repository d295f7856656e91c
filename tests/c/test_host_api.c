/*
 * test_host_api.c -- the reference's C test-suites (tests/src/test_basic_qp.c, test_degen_hess.c,
 * test_update.c, test_prim_inf_qp.c, test_dua_inf_qp.c, test_solver_interface.c) restated against
 * include/qpalm_host.h, i.e. through qpalm_setup()/qpalm_solve()/QPALMWorkspace exactly as a user of
 * the reference would call them.  Problem data and expected values come from the golden fixture
 * (golden_data.h is generated from tests/golden/reference_tests.json by tests/test_host_c_api.py).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/qpalm_host.h"
#include "golden_data.h"

static int g_fail = 0, g_checks = 0;
static int g_verbose = 0; /* the reference's suites keep the default verbose = TRUE: every suite runs with both settings */
#define CHECK(cond) do { g_checks++; if (!(cond)) { g_fail++; printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)
#define CHECK_NEAR(a, b, tol) do { g_checks++; if (!(fabs((a) - (b)) <= (tol))) { g_fail++; printf("FAIL %s:%d: %s=%.15g vs %.15g (tol %g)\n", __FILE__, __LINE__, #a, (double)(a), (double)(b), (double)(tol)); } } while (0)

static QPALMData *make_data(const golden_problem *g) {
  QPALMData *d = (QPALMData *)calloc(1, sizeof(QPALMData));
  d->n = g->n; d->m = g->m; d->c = 0;
  d->A = qpalm_sparse_alloc(g->m, g->n, g->nnzA ? g->nnzA : 1, 0);
  d->Q = qpalm_sparse_alloc(g->n, g->n, g->nnzQ ? g->nnzQ : 1, -1);
  memcpy(d->A->p, g->Ap, (g->n + 1) * sizeof(c_int)); memcpy(d->A->i, g->Ai, g->nnzA * sizeof(c_int)); memcpy(d->A->x, g->Ax, g->nnzA * sizeof(c_float));
  memcpy(d->Q->p, g->Qp, (g->n + 1) * sizeof(c_int)); memcpy(d->Q->i, g->Qi, g->nnzQ * sizeof(c_int)); memcpy(d->Q->x, g->Qx, g->nnzQ * sizeof(c_float));
  d->q = (c_float *)malloc(g->n * sizeof(c_float)); memcpy(d->q, g->q, g->n * sizeof(c_float));
  d->bmin = (c_float *)malloc(g->m * sizeof(c_float)); memcpy(d->bmin, g->bmin, g->m * sizeof(c_float));
  d->bmax = (c_float *)malloc(g->m * sizeof(c_float)); memcpy(d->bmax, g->bmax, g->m * sizeof(c_float));
  return d;
}
static void free_data(QPALMData *d) { qpalm_sparse_free(&d->A); qpalm_sparse_free(&d->Q); free(d->q); free(d->bmin); free(d->bmax); free(d); }

/* suite_basic_qp (tests/src/test_basic_qp.c:90-427) */
static void basic_defaults(QPALMSettings *s) {
  qpalm_set_default_settings(s);
  s->max_rank_update_fraction = 1.0; s->verbose = g_verbose;
  s->eps_abs = 1e-6; s->eps_rel = 1e-6; s->gamma_init = 1e1;
}
static void check_basic_solution(QPALMWorkspace *work) {
  CHECK(work->info->status_val == QPALM_SOLVED);
  for (int i = 0; i < 4; i++) CHECK_NEAR(work->solution->x[i], basic_qp_solution[i], fabs(1e-5 * basic_qp_solution[i]));
}
static void suite_basic_qp(void) {
  QPALMData *data = make_data(&golden_basic_qp);
  QPALMSettings s;
  QPALMWorkspace *work;
  basic_defaults(&s); work = qpalm_setup(data, &s); CHECK(work != QPALM_NULL); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.scaling = 0; work = qpalm_setup(data, &s); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.proximal = FALSE; s.scaling = 2; work = qpalm_setup(data, &s); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.proximal = FALSE; s.scaling = 0; work = qpalm_setup(data, &s); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  { /* test_basic_qp_warm_start */
    c_float x[4] = {2.0, -60.0, -3380.0, -6.0}, y[5] = {0.0, 0.0, -23.0, -0.014, 0.0};
    basic_defaults(&s); s.scaling = 2; s.warm_start = TRUE;
    work = qpalm_setup(data, &s); qpalm_warm_start(work, x, y); qpalm_solve(work);
    CHECK(work->info->iter < 12); check_basic_solution(work); qpalm_cleanup(work);
  }
  { /* test_basic_qp_warm_start_resolve: identical to 1e-15, same iteration count */
    c_float x[4], y[5], xs[4], ys[5];
    basic_defaults(&s); work = qpalm_setup(data, &s);
    memcpy(x, work->x, sizeof x); memcpy(y, work->y, sizeof y);
    qpalm_solve(work); CHECK(work->info->status_val == QPALM_SOLVED);
    memcpy(xs, work->solution->x, sizeof xs); memcpy(ys, work->solution->y, sizeof ys);
    c_int iter = work->info->iter;
    qpalm_warm_start(work, x, y); qpalm_solve(work);
    CHECK(work->info->iter == iter);
    for (int i = 0; i < 4; i++) CHECK_NEAR(work->solution->x[i], xs[i], 1e-15);
    for (int i = 0; i < 5; i++) CHECK_NEAR(work->solution->y[i], ys[i], 1e-15);
    qpalm_cleanup(work);
  }
  basic_defaults(&s); s.max_iter = 1; work = qpalm_setup(data, &s); qpalm_solve(work); CHECK(work->info->status_val == QPALM_MAX_ITER_REACHED); qpalm_cleanup(work);
  basic_defaults(&s); s.eps_abs = 1e-8; s.eps_rel = 1e-8; s.inner_max_iter = 2; s.max_iter = 10;
  work = qpalm_setup(data, &s); qpalm_solve(work); CHECK(work->info->status_val == QPALM_MAX_ITER_REACHED); qpalm_cleanup(work);
  basic_defaults(&s); s.sigma_max = 1e3; work = qpalm_setup(data, &s); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.time_limit = 0.01 * 1e-3; work = qpalm_setup(data, &s); qpalm_solve(work); CHECK(work->info->status_val == QPALM_TIME_LIMIT_REACHED); qpalm_cleanup(work);
  /* the reference runs this suite in every factorization mode under LADEL (test_basic_qp.c:410-427): FACTORIZE_KKT here */
  basic_defaults(&s); s.factorization_method = FACTORIZE_KKT; work = qpalm_setup(data, &s); CHECK(work != QPALM_NULL);
  CHECK(work->solver->factorization_method == FACTORIZE_KKT); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.factorization_method = FACTORIZE_KKT; s.proximal = FALSE; s.scaling = 0;
  work = qpalm_setup(data, &s); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  basic_defaults(&s); s.factorization_method = FACTORIZE_SCHUR; work = qpalm_setup(data, &s);
  CHECK(work->solver->factorization_method == FACTORIZE_SCHUR); qpalm_solve(work); check_basic_solution(work); qpalm_cleanup(work);
  /* test_basic_qp_dual_objective (test_basic_qp.c:334-349) */
  basic_defaults(&s); s.enable_dual_termination = TRUE; work = qpalm_setup(data, &s); CHECK(work != QPALM_NULL); qpalm_solve(work);
  check_basic_solution(work); CHECK_NEAR(work->info->objective, work->info->dual_objective, 1e-5); qpalm_cleanup(work);
  /* test_basic_qp_dual_early_termination (test_basic_qp.c:351-362) */
  basic_defaults(&s); s.enable_dual_termination = TRUE; s.dual_objective_limit = -1000000000.0;
  work = qpalm_setup(data, &s); qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_DUAL_TERMINATED); CHECK(work->info->iter_out == 0); qpalm_cleanup(work);
  free_data(data);
}

/* suite_degen_hess (tests/src/test_degen_hess.c:95-105) */
static void suite_degen_hess(void) {
  QPALMData *data = make_data(&golden_degen_hess);
  QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.max_rank_update_fraction = 1.0; s.verbose = g_verbose;
  QPALMWorkspace *work = qpalm_setup(data, &s);
  qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_SOLVED);
  CHECK_NEAR(work->solution->x[0], 5.5, 1e-5); CHECK_NEAR(work->solution->x[1], 5, 1e-5); CHECK_NEAR(work->solution->x[2], -10, 1e-5);
  qpalm_cleanup(work); free_data(data);
}

/* suite_prim_inf_qp / suite_dua_inf_qp */
static void suite_infeasible(void) {
  for (int which = 0; which < 2; which++) {
    QPALMData *data = make_data(which ? &golden_dua_inf_qp : &golden_prim_inf_qp);
    const int prox[4] = {1, 1, 0, 0}, scal[4] = {2, 0, 2, 0};
    for (int k = 0; k < 4; k++) {
      QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.max_rank_update_fraction = 1.0; s.verbose = g_verbose;
      s.proximal = prox[k]; s.scaling = scal[k];
      QPALMWorkspace *work = qpalm_setup(data, &s); qpalm_solve(work);
      CHECK(work->info->status_val == (which ? QPALM_DUAL_INFEASIBLE : QPALM_PRIMAL_INFEASIBLE));
      qpalm_cleanup(work);
    }
    free_data(data);
  }
}

/* suite_update (tests/src/test_update.c:91-148): three tests on one workspace */
static void suite_update(void) {
  QPALMData *data = make_data(&golden_update);
  QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.scaling = 2; s.proximal = TRUE; s.verbose = g_verbose;
  QPALMWorkspace *work = qpalm_setup(data, &s);
  qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_SOLVED); CHECK_NEAR(work->solution->x[0], -0.1, 1e-5); CHECK_NEAR(work->solution->x[1], 0.3, 1e-5);
  s.gamma_init *= 0.1; s.theta = 0.9; s.proximal = TRUE; s.scaling = 10;
  qpalm_update_settings(work, &s); CHECK(work->info->status_val != QPALM_ERROR);
  qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_SOLVED); CHECK_NEAR(work->solution->x[0], -0.1, 1e-5); CHECK_NEAR(work->solution->x[1], 0.3, 1e-5);
  data->bmin[0] = 0.0; data->bmax[1] = 1.5;
  qpalm_update_bounds(work, data->bmin, data->bmax); qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_SOLVED); CHECK_NEAR(work->solution->x[0], 0.0, 1e-5); CHECK_NEAR(work->solution->x[1], 0.15, 1e-5);
  data->bmin[0] = -1; data->bmax[1] = 3; qpalm_update_bounds(work, data->bmin, data->bmax);
  data->q[0] = -0.5; data->q[1] = -0.75; qpalm_update_q(work, data->q); qpalm_solve(work);
  CHECK(work->info->status_val == QPALM_SOLVED); CHECK_NEAR(work->solution->x[0], 0.02, 1e-5); CHECK_NEAR(work->solution->x[1], 0.18, 1e-5);
  /* error handling (tests/src/test_error_handling.c:99-134) */
  s.max_iter = -10; qpalm_update_settings(work, &s); CHECK(work->info->status_val == QPALM_ERROR);
  qpalm_cleanup(work);
  s.max_iter = -1; CHECK(qpalm_setup(data, &s) == QPALM_NULL);
  free_data(data);
}

/* suite_nonconvex (tests/src/test_nonconvex_qp.c:117-139): the three factorization modes */
static void suite_nonconvex(void) {
  QPALMData *data = make_data(&golden_nonconvex_qp);
  const c_int methods[3] = {FACTORIZE_KKT_OR_SCHUR, FACTORIZE_KKT, FACTORIZE_SCHUR};
  for (int k = 0; k < 3; k++) {
    QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.nonconvex = TRUE; s.scaling = FALSE;
    s.max_rank_update_fraction = 1.0; s.factorization_method = methods[k]; s.verbose = g_verbose;
    QPALMWorkspace *work = qpalm_setup(data, &s);
    CHECK(work != QPALM_NULL);
    if (!work) continue;
    CHECK(work->settings->proximal == TRUE && work->settings->gamma_max == work->settings->gamma_init);
    qpalm_solve(work);
    CHECK(work->info->status_val == QPALM_SOLVED);
    CHECK_NEAR(work->gamma, 1.0 / 0.0021544347, 1e-1 * 1.0 / 0.0021544347); /* inverse of the lowest eigenvalue */
    CHECK(1 / work->gamma > 0.0021544347);                                   /* the eigenvalue is under-approximated */
    qpalm_cleanup(work);
  }
  free_data(data);
}

/* suite_solver (tests/src/test_solver_interface.c:106-160) */
static void suite_solver(void) {
  QPALMData *data = make_data(&golden_solver_interface);
  QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.verbose = 0;
  QPALMWorkspace *work = qpalm_setup(data, &s);
  solver_common common, *c = &common;
  const double TOL = 1e-8;
  work->Qd[0] = 1.1; work->Qd[1] = -0.5; work->Ad[0] = 1.1; work->Ad[1] = -0.5; work->Ad[2] = 20;
  mat_vec(data->A, work->solver->Qd, work->solver->Ad, c);
  CHECK_NEAR(work->Ad[0], 0.1, TOL); CHECK_NEAR(work->Ad[1], 1.3, TOL); CHECK_NEAR(work->Ad[2], 5.5, TOL);
  mat_vec(data->Q, work->solver->Qd, work->solver->Qd, c); /* aliased */
  CHECK_NEAR(work->Qd[0], 1.6, TOL); CHECK_NEAR(work->Qd[1], -2.1, TOL);
  work->Qd[0] = 1.1; work->Qd[1] = -0.5;
  mat_tpose_vec(data->Q, work->solver->Qd, work->solver->Qd, c);
  CHECK_NEAR(work->Qd[0], 1.6, TOL); CHECK_NEAR(work->Qd[1], -2.1, TOL);
  work->Ad[0] = 1.1; work->Ad[1] = -0.5; work->Ad[2] = 20;
  mat_tpose_vec(data->A, work->solver->Ad, work->solver->Qd, c);
  CHECK_NEAR(work->Qd[0], 99.6, TOL); CHECK_NEAR(work->Qd[1], 0.2, TOL);
  c_float cols[2], rows[3];
  mat_inf_norm_cols(data->A, cols); mat_inf_norm_rows(data->A, rows);
  CHECK_NEAR(cols[0], 5.0, TOL); CHECK_NEAR(cols[1], 4.0, TOL); CHECK_NEAR(rows[0], 2.0, TOL); CHECK_NEAR(rows[1], 4.0, TOL); CHECK_NEAR(rows[2], 5.0, TOL);
  /* test_ldlchol: the USER's Q, with and without the proximal term */
  work->settings->proximal = FALSE; work->dphi[0] = -1.0; work->dphi[1] = -2.0;
  ldlchol(data->Q, work, c); ldlsolveLD_neg_dphi(work, c);
  CHECK_NEAR(work->d[0], 4.0, TOL); CHECK_NEAR(work->d[1], 3.0, TOL);
  work->settings->proximal = TRUE; work->gamma = 1e3;
  ldlchol(data->Q, work, c); ldlsolveLD_neg_dphi(work, c);
  CHECK_NEAR(work->d[0], 3.989028924198480, TOL); CHECK_NEAR(work->d[1], 2.993017953122679, TOL);
  qpalm_cleanup(work); free_data(data);
}

/* the KKT set of solver_interface.h:82-126 through the host names (no reference test exists for them: LADEL-only, driven from
 * newton.c).  Known answer without active constraints: K = diag(Q + I/gamma, I), so kkt_solve must reproduce test_ldlchol's
 * d; with constraint 0 added by kkt_update_entering_constraints the solution must satisfy the Schur form of the same system,
 * (Q + I/gamma + sigma_0 a_0 a_0') d = -dphi, and kkt_update_leaving_constraints must bring the first answer back. */
static void suite_kkt(void) {
  QPALMData *data = make_data(&golden_solver_interface);
  QPALMSettings s; qpalm_set_default_settings(&s); s.eps_abs = 1e-6; s.eps_rel = 1e-6; s.verbose = 0; s.scaling = 0;
  s.factorization_method = 0; /* FACTORIZE_KKT */
  QPALMWorkspace *work = qpalm_setup(data, &s);
  solver_common common, *c = &common;
  const double TOL = 1e-8;
  const size_t m = data->m;
  work->settings->proximal = TRUE; work->gamma = 1e3; work->dphi[0] = -1.0; work->dphi[1] = -2.0;
  for (size_t k = 0; k < m; k++) { work->solver->active_constraints[k] = 0; work->sigma_inv[k] = 0.5; }
  work->solver->nb_enter = 0; work->solver->nb_leave = 0;
  qpalm_form_kkt(work); qpalm_kkt_factorize(work); kkt_solve(work, c);
  CHECK_NEAR(work->d[0], 3.989028924198480, TOL); CHECK_NEAR(work->d[1], 2.993017953122679, TOL);
  /* row 0 of A = (1, 2) enters (golden_solver_interface: A = [[1,2],[3,4],[5,0]]) */
  work->solver->active_constraints[0] = 1; work->solver->enter[0] = 0; work->solver->nb_enter = 1;
  kkt_update_entering_constraints(work, c); kkt_solve(work, c);
  {
    const double a0[2] = {1.0, 2.0}, sig = 2.0, g = 1e-3;
    const double ad = a0[0] * work->d[0] + a0[1] * work->d[1];
    const double r0 = (1.0 + g) * work->d[0] - 1.0 * work->d[1] + sig * a0[0] * ad, r1 = -1.0 * work->d[0] + (2.0 + g) * work->d[1] + sig * a0[1] * ad;
    CHECK_NEAR(r0, 1.0, TOL); CHECK_NEAR(r1, 2.0, TOL);
  }
  /* ... and leaves again; qpalm_reform_kkt + factorisation from scratch gives the same */
  work->solver->active_constraints[0] = 0; work->solver->leave[0] = 0; work->solver->nb_enter = 0; work->solver->nb_leave = 1;
  kkt_update_leaving_constraints(work, c); kkt_solve(work, c);
  CHECK_NEAR(work->d[0], 3.989028924198480, TOL); CHECK_NEAR(work->d[1], 2.993017953122679, TOL);
  work->solver->nb_leave = 0;
  qpalm_reform_kkt(work); qpalm_kkt_factorize(work); kkt_solve(work, c);
  CHECK_NEAR(work->d[0], 3.989028924198480, TOL); CHECK_NEAR(work->d[1], 2.993017953122679, TOL);
  qpalm_cleanup(work); free_data(data);
}

int main(void) {
  suite_solver();
  suite_kkt();
  for (g_verbose = 0; g_verbose < 2; g_verbose++) { /* 1 = the host-driven one-iteration-per-launch path with printing */
    suite_basic_qp();
    suite_degen_hess();
    suite_infeasible();
    suite_update();
    suite_nonconvex();
  }
  printf("%d checks, %d failures\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
