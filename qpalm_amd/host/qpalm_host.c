/*
 * qpalm_host.c -- the reference's C API (include/qpalm.h) implemented on the gfx950 backend.
 *
 * Plain C host code, as the north-star asks: it validates, allocates the QPALMWorkspace exactly like
 * qpalm_setup (src/qpalm.c:73-319), and forwards every numerical step to the HIP kernels through
 * the C ABI of include/qpalm_gfx950.h (a batch of one QP resident in HBM).  After each call the
 * workspace mirrors are refreshed so that code reading work->x, work->info, work->solver->... sees
 * what the reference would show.  There is no host-side numerical fallback.
 */
#include "../../include/qpalm_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/qpalm_gfx950.h"

struct qpalm_gfx950_state {
  qpg_ctx *ctx;
  qpg_batch *bt;
};

static char g_host_err[256];
const char *qpalm_backend_error(void) { return g_host_err[0] ? g_host_err : qpg_last_error(); }

#define STATE(work) ((work)->solver->LD)
#define BT(work) ((work)->solver->LD->bt)

static void *zalloc(size_t count, size_t size) { return calloc(count ? count : 1, size); }

solver_sparse *qpalm_sparse_alloc(size_t nrow, size_t ncol, size_t nzmax, int stype) {
  solver_sparse *A = (solver_sparse *)zalloc(1, sizeof(solver_sparse));
  A->nrow = nrow; A->ncol = ncol; A->nzmax = nzmax; A->stype = stype; A->sorted = 1; A->packed = 1;
  A->p = zalloc(ncol + 1, sizeof(c_int)); A->i = zalloc(nzmax, sizeof(c_int)); A->x = zalloc(nzmax, sizeof(c_float));
  return A;
}
void qpalm_sparse_free(solver_sparse **A) {
  if (!A || !*A) return;
  free((*A)->p); free((*A)->i); free((*A)->x); free(*A); *A = NULL;
}
static solver_sparse *sparse_copy(const solver_sparse *S) { /* cholmod copy_sparse, qpalm.c:141-143 */
  const c_int nz = ((const c_int *)S->p)[S->ncol];
  solver_sparse *A = qpalm_sparse_alloc(S->nrow, S->ncol, (size_t)(nz > 0 ? nz : 1), S->stype);
  memcpy(A->p, S->p, (S->ncol + 1) * sizeof(c_int));
  memcpy(A->i, S->i, (size_t)nz * sizeof(c_int));
  memcpy(A->x, S->x, (size_t)nz * sizeof(c_float));
  return A;
}
static solver_dense *dense_alloc(size_t n) { /* cholmod allocate_dense(n,1,n,REAL), zero filled */
  solver_dense *d = (solver_dense *)zalloc(1, sizeof(solver_dense));
  d->nrow = n; d->ncol = 1; d->nzmax = n; d->d = n; d->x = zalloc(n, sizeof(c_float));
  return d;
}
static void dense_free(solver_dense **d) { if (d && *d) { free((*d)->x); free(*d); *d = NULL; } }

static void set_status(QPALMInfo *info, c_int v) { /* src/util.c:61-105 */
  static const struct { c_int v; const char *s; } tab[] = {
      {QPALM_SOLVED, "solved"}, {QPALM_DUAL_TERMINATED, "dual terminated"}, {QPALM_PRIMAL_INFEASIBLE, "primal infeasible"},
      {QPALM_DUAL_INFEASIBLE, "dual infeasible"}, {QPALM_TIME_LIMIT_REACHED, "time limit exceeded"},
      {QPALM_MAX_ITER_REACHED, "maximum iterations reached"}, {QPALM_UNSOLVED, "unsolved"}, {QPALM_ERROR, "error"}};
  const char *s = "unrecognised status value";
  for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); k++) if (tab[k].v == v) s = tab[k].s;
  info->status_val = v;
  memset(info->status, 0, sizeof(info->status));
  strncpy(info->status, s, sizeof(info->status) - 1);
}

void qpalm_set_default_settings(QPALMSettings *settings) { qpg_set_default_settings((QPGSettings *)settings); }

/* ---- mirrors --------------------------------------------------------------------------------- */
static void getv(QPALMWorkspace *w, const char *name, c_float *dst, size_t len) {
  if (dst && len) qpg_batch_get_vector(BT(w), name, 0, dst, (qpg_int)len);
}
static void setv(QPALMWorkspace *w, const char *name, const c_float *src, size_t len) {
  if (src && len) qpg_batch_set_vector(BT(w), name, 0, src, (qpg_int)len);
}

static void pull(QPALMWorkspace *w) {
  const size_t n = w->data->n, m = w->data->m;
  getv(w, "x", w->x, n); getv(w, "y", w->y, m); getv(w, "Ax", w->Ax, m); getv(w, "Qx", w->Qx, n); getv(w, "Aty", w->Aty, n);
  getv(w, "x_prev", w->x_prev, n); getv(w, "x0", w->x0, n); getv(w, "sigma", w->sigma, m); getv(w, "sigma_inv", w->sigma_inv, m);
  getv(w, "sqrt_sigma", w->sqrt_sigma, m); getv(w, "Axys", w->Axys, m); getv(w, "z", w->z, m); getv(w, "pri_res", w->pri_res, m);
  getv(w, "pri_res_in", w->pri_res_in, m); getv(w, "yh", w->yh, m); getv(w, "Atyh", w->Atyh, n); getv(w, "df", w->df, n);
  getv(w, "dphi", w->dphi, n); getv(w, "dphi_prev", w->dphi_prev, n); getv(w, "d", w->d, n); getv(w, "Qd", w->Qd, n);
  getv(w, "Ad", w->Ad, m); getv(w, "delta_y", w->delta_y, m); getv(w, "delta_x", w->delta_x, n);
  getv(w, "delta", w->delta, 2 * m); getv(w, "alpha", w->alpha, 2 * m);
  getv(w, "At_scale", (c_float *)w->solver->At_scale->x, m);
  getv(w, "q", w->data->q, n); getv(w, "bmin", w->data->bmin, m); getv(w, "bmax", w->data->bmax, m);
  getv(w, "A_values", (c_float *)w->data->A->x, (size_t)((c_int *)w->data->A->p)[n]);
  if (w->scaling) {
    getv(w, "D", w->scaling->D, n); getv(w, "Dinv", w->scaling->Dinv, n); getv(w, "E", w->scaling->E, m); getv(w, "Einv", w->scaling->Einv, m);
  }
  getv(w, "solution_x", w->solution->x, n); getv(w, "solution_y", w->solution->y, m);
  if (m) {
    qpg_batch_get_ivector(BT(w), "active", 0, w->solver->active_constraints, (qpg_int)m);
    qpg_batch_get_ivector(BT(w), "active_old", 0, w->solver->active_constraints_old, (qpg_int)m);
    qpg_batch_get_ivector(BT(w), "enter", 0, w->solver->enter, (qpg_int)m);
    qpg_batch_get_ivector(BT(w), "leave", 0, w->solver->leave, (qpg_int)m);
  }
  QPGStats st;
  if (qpg_batch_get_stats(BT(w), 0, &st) == QPG_OK) {
    w->gamma = st.gamma; w->tau = st.tau; w->eta = st.eta; w->beta = st.beta;
    w->eps_pri = st.eps_pri; w->eps_dua = st.eps_dua; w->eps_dua_in = st.eps_dua_in;
    w->solver->nb_active_constraints = st.nb_active; w->solver->nb_enter = st.nb_enter; w->solver->nb_leave = st.nb_leave;
    if (w->scaling) { w->scaling->c = st.sc_c; w->scaling->cinv = 1.0 / st.sc_c; }
  }
  QPGInfo gi;
  if (qpg_batch_get_info(BT(w), 0, &gi) == QPG_OK) {
    const c_float setup_time = w->info->setup_time;
    memcpy(w->info, &gi, sizeof(QPALMInfo));
    w->info->setup_time = setup_time;
    w->info->run_time = setup_time + w->info->solve_time;
  }
}

/* ---- qpalm_setup (src/qpalm.c:73-319) ------------------------------------------------------------ */
QPALMWorkspace *qpalm_setup(const QPALMData *data, const QPALMSettings *settings) {
  g_host_err[0] = 0;
  if (!data) { snprintf(g_host_err, sizeof g_host_err, "Missing data"); return QPALM_NULL; }
  for (size_t j = 0; j < data->m; j++)
    if (data->bmin[j] > data->bmax[j]) { snprintf(g_host_err, sizeof g_host_err, "Data validation returned failure"); return QPALM_NULL; }
  if (!qpg_validate_settings((const QPGSettings *)settings)) { snprintf(g_host_err, sizeof g_host_err, "Settings validation returned failure"); return QPALM_NULL; }
  const size_t n = data->n, m = data->m;
  QPALMWorkspace *work = (QPALMWorkspace *)zalloc(1, sizeof(QPALMWorkspace));
  work->settings = (QPALMSettings *)malloc(sizeof(QPALMSettings));
  *work->settings = *settings;
  work->sqrt_delta = sqrt(settings->delta);
  work->gamma = settings->gamma_init;
  work->solver = (QPALMSolver *)zalloc(1, sizeof(QPALMSolver));
  work->data = (QPALMData *)zalloc(1, sizeof(QPALMData));
  work->data->n = n; work->data->m = m; work->data->c = data->c;
  work->data->bmin = (c_float *)zalloc(m, sizeof(c_float)); memcpy(work->data->bmin, data->bmin, m * sizeof(c_float));
  work->data->bmax = (c_float *)zalloc(m, sizeof(c_float)); memcpy(work->data->bmax, data->bmax, m * sizeof(c_float));
  work->data->q = (c_float *)zalloc(n, sizeof(c_float)); memcpy(work->data->q, data->q, n * sizeof(c_float));
  work->data->A = sparse_copy(data->A); work->data->A->stype = 0;
  work->data->Q = sparse_copy(data->Q);
#define VN(f) work->f = (c_float *)zalloc(n, sizeof(c_float))
#define VM(f) work->f = (c_float *)zalloc(m, sizeof(c_float))
  VN(x); VM(y); VM(Ax); VN(Qx); VN(x_prev); VN(Aty); VN(x0);
  VM(temp_m); VN(temp_n); VM(sigma); VM(sigma_inv); VM(z); VM(Axys); VM(pri_res); VM(pri_res_in); VN(df); VN(xx0); VN(dphi); VN(dphi_prev);
  VM(sqrt_sigma);
  work->delta = (c_float *)zalloc(2 * m, sizeof(c_float)); work->alpha = (c_float *)zalloc(2 * m, sizeof(c_float));
  work->delta2 = (c_float *)zalloc(2 * m, sizeof(c_float)); work->delta_alpha = (c_float *)zalloc(2 * m, sizeof(c_float));
  work->temp_2m = (c_float *)zalloc(2 * m, sizeof(c_float)); work->s = (array_element *)zalloc(2 * m, sizeof(array_element));
  work->index_L = (c_int *)zalloc(2 * m, sizeof(c_int)); work->index_P = (c_int *)zalloc(2 * m, sizeof(c_int)); work->index_J = (c_int *)zalloc(2 * m, sizeof(c_int));
  VM(delta_y); VN(Atdelta_y); VN(delta_x); VN(Qdelta_x); VM(Adelta_x);
#undef VN
#undef VM
  qpalm_set_factorization_method(work, NULL);
  if (settings->scaling) {
    work->scaling = (QPALMScaling *)zalloc(1, sizeof(QPALMScaling));
    work->scaling->D = (c_float *)zalloc(n, sizeof(c_float)); work->scaling->Dinv = (c_float *)zalloc(n, sizeof(c_float));
    work->scaling->E = (c_float *)zalloc(m, sizeof(c_float)); work->scaling->Einv = (c_float *)zalloc(m, sizeof(c_float));
    work->solver->E_temp = dense_alloc(m); work->E_temp = (c_float *)work->solver->E_temp->x;
    work->solver->D_temp = dense_alloc(n); work->D_temp = (c_float *)work->solver->D_temp->x;
  }
  work->solver->active_constraints = (c_int *)zalloc(m, sizeof(c_int));
  work->solver->active_constraints_old = (c_int *)zalloc(m, sizeof(c_int));
  work->solver->reset_newton = TRUE;
  work->solver->enter = (c_int *)zalloc(m, sizeof(c_int)); work->solver->leave = (c_int *)zalloc(m, sizeof(c_int));
  /* solver-side dense vectors aliased into the workspace (qpalm.c:278-290) */
  work->solver->neg_dphi = dense_alloc(n); work->neg_dphi = (c_float *)work->solver->neg_dphi->x;
  work->solver->d = dense_alloc(n); work->d = (c_float *)work->solver->d->x;
  work->solver->Qd = dense_alloc(n); work->Qd = (c_float *)work->solver->Qd->x;
  work->solver->Ad = dense_alloc(m); work->Ad = (c_float *)work->solver->Ad->x;
  work->solver->yh = dense_alloc(m); work->yh = (c_float *)work->solver->yh->x;
  work->solver->Atyh = dense_alloc(n); work->Atyh = (c_float *)work->solver->Atyh->x;
  work->solver->At_scale = dense_alloc(m);
  work->solution = (QPALMSolution *)zalloc(1, sizeof(QPALMSolution));
  work->solution->x = (c_float *)zalloc(n, sizeof(c_float)); work->solution->y = (c_float *)zalloc(m, sizeof(c_float));
  work->info = (QPALMInfo *)zalloc(1, sizeof(QPALMInfo));
  set_status(work->info, QPALM_UNSOLVED);
  /* the device side: context, one-QP batch, upload + Ruiz scaling on the GPU */
  solver_factor *stt = (solver_factor *)zalloc(1, sizeof(solver_factor));
  work->solver->LD = stt;
  const char *dev = getenv("QPALM_GFX950_DEVICE");
  int rc = qpg_ctx_create(dev ? atoi(dev) : 0, &stt->ctx);
  if (rc == QPG_OK) {
    const c_int nzA = ((c_int *)data->A->p)[n], nzQ = ((c_int *)data->Q->p)[n];
    rc = qpg_batch_create(stt->ctx, 1, (qpg_int)n, (qpg_int)m, nzA, nzQ, (const QPGSettings *)settings, &stt->bt);
    if (rc == QPG_OK)
      rc = qpg_batch_set_problem(stt->bt, 0, (const qpg_int *)data->Q->p, (const qpg_int *)data->Q->i, (const qpg_float *)data->Q->x,
                                 (const qpg_int *)data->A->p, (const qpg_int *)data->A->i, (const qpg_float *)data->A->x, data->q,
                                 data->c, data->bmin, data->bmax);
    if (rc == QPG_OK) rc = qpg_batch_setup(stt->bt);
  }
  if (rc != QPG_OK) {
    snprintf(g_host_err, sizeof g_host_err, "%s", qpg_last_error());
    qpalm_cleanup(work);
    return QPALM_NULL;
  }
  pull(work);
  if (settings->nonconvex) { /* set_settings_nonconvex ran on the device (nonconvex.c:171-183): mirror what it did to the settings */
    QPGStats gs;
    if (qpg_batch_get_stats(stt->bt, 0, &gs) == QPG_OK) {
      if (gs.nonconvex) {
        work->settings->proximal = TRUE;
        work->settings->gamma_init = 1 / (gs.lobpcg_lambda < 0 ? -gs.lobpcg_lambda : gs.lobpcg_lambda);
        work->settings->gamma_max = work->settings->gamma_init;
        work->gamma_maxed = TRUE;
      } else work->settings->nonconvex = FALSE;
    }
  }
  set_status(work->info, QPALM_UNSOLVED);
  return work;
}

void qpalm_warm_start(QPALMWorkspace *work, c_float *x_warm_start, c_float *y_warm_start) { /* qpalm.c:322-399 */
  qpg_batch_warm_start(BT(work), x_warm_start, y_warm_start);
  work->initialized = TRUE;
  const c_int status = work->info->status_val;
  pull(work);
  set_status(work->info, status);
}

static void print_header(void) { printf("\n                 QPALM (gfx950 backend)\n\nIter |   P. res   |   D. res   |  Stepsize  |  Objective \n"); }

void qpalm_solve(QPALMWorkspace *work) { /* qpalm.c:401-736 */
  if (work->settings->verbose) {
    /* host-driven, one iteration per launch, so that progress can be printed (util.c:107-206) */
    print_header();
    qpg_int left = 1;
    qpg_batch_begin_solve(BT(work)); /* a finished workspace starts a new solve (qpalm.c:409-424) */
    qpg_batch_iterate(BT(work), 1);
    while (qpg_batch_num_unfinished(BT(work), &left) == QPG_OK && left > 0) {
      QPGInfo gi; QPGStats st;
      qpg_batch_get_info(BT(work), 0, &gi); qpg_batch_get_stats(BT(work), 0, &st);
      if (gi.iter % (work->settings->print_iter > 0 ? work->settings->print_iter : 1) == 0)
        printf("%4ld | %.4e | %.4e | %.4e | %.4e \n", (long)gi.iter, gi.pri_res_norm, gi.dua_res_norm, st.tau, gi.objective);
      if (qpg_batch_iterate(BT(work), 1) != QPG_OK) break;
    }
  } else {
    qpg_batch_solve(BT(work));
  }
  pull(work);
  work->initialized = FALSE;
  if (work->settings->verbose) printf("\nQPALM finished: %s, %ld iterations\n", work->info->status, (long)work->info->iter);
}

void qpalm_update_settings(QPALMWorkspace *work, const QPALMSettings *settings) { /* qpalm.c:739-791 */
  if (qpg_batch_update_settings(BT(work), (const QPGSettings *)settings) != QPG_OK) {
    const char *e = qpg_last_error();
    if (strstr(e, "validation") || strstr(e, "Decreasing")) set_status(work->info, QPALM_ERROR);
    snprintf(g_host_err, sizeof g_host_err, "%s", e);
    return;
  }
  *work->settings = *settings;
  work->sqrt_delta = sqrt(settings->delta);
  const c_int status = work->info->status_val;
  if (settings->scaling && !work->scaling) {
    const size_t n = work->data->n, m = work->data->m;
    work->scaling = (QPALMScaling *)zalloc(1, sizeof(QPALMScaling));
    work->scaling->D = (c_float *)zalloc(n, sizeof(c_float)); work->scaling->Dinv = (c_float *)zalloc(n, sizeof(c_float));
    work->scaling->E = (c_float *)zalloc(m, sizeof(c_float)); work->scaling->Einv = (c_float *)zalloc(m, sizeof(c_float));
  }
  pull(work);
  set_status(work->info, status);
}

void qpalm_update_bounds(QPALMWorkspace *work, const c_float *bmin, const c_float *bmax) { /* qpalm.c:793-827 */
  if (qpg_batch_update_bounds(BT(work), bmin, bmax) != QPG_OK) { set_status(work->info, QPALM_ERROR); return; }
  const size_t m = work->data->m;
  getv(work, "bmin", work->data->bmin, m); getv(work, "bmax", work->data->bmax, m);
}

void qpalm_update_q(QPALMWorkspace *work, const c_float *q) { /* qpalm.c:829-871 */
  if (qpg_batch_update_q(BT(work), q) != QPG_OK) { set_status(work->info, QPALM_ERROR); return; }
  const c_int status = work->info->status_val;
  pull(work);
  set_status(work->info, status);
}

void qpalm_cleanup(QPALMWorkspace *work) { /* qpalm.c:874-1096 */
  if (!work) return;
  if (work->data) {
    qpalm_sparse_free(&work->data->A); qpalm_sparse_free(&work->data->Q);
    free(work->data->q); free(work->data->bmin); free(work->data->bmax); free(work->data);
  }
  if (work->scaling) { free(work->scaling->D); free(work->scaling->Dinv); free(work->scaling->E); free(work->scaling->Einv); free(work->scaling); }
  c_float *fv[] = {work->x, work->y, work->Ax, work->Qx, work->x_prev, work->Aty, work->x0, work->temp_m, work->temp_n, work->sigma,
                   work->sigma_inv, work->z, work->Axys, work->pri_res, work->pri_res_in, work->df, work->xx0, work->dphi,
                   work->dphi_prev, work->sqrt_sigma, work->delta, work->alpha, work->delta2, work->delta_alpha, work->temp_2m,
                   work->delta_y, work->Atdelta_y, work->delta_x, work->Qdelta_x, work->Adelta_x};
  for (size_t k = 0; k < sizeof(fv) / sizeof(fv[0]); k++) free(fv[k]);
  free(work->s); free(work->index_L); free(work->index_P); free(work->index_J);
  if (work->solver) {
    QPALMSolver *s = work->solver;
    if (s->LD) { if (s->LD->bt) qpg_batch_destroy(s->LD->bt); if (s->LD->ctx) qpg_ctx_destroy(s->LD->ctx); free(s->LD); }
    dense_free(&s->E_temp); dense_free(&s->D_temp); dense_free(&s->neg_dphi); dense_free(&s->d); dense_free(&s->Qd); dense_free(&s->Ad);
    dense_free(&s->yh); dense_free(&s->Atyh); dense_free(&s->At_scale);
    free(s->active_constraints); free(s->active_constraints_old); free(s->enter); free(s->leave);
    free(s);
  }
  if (work->solution) { free(work->solution->x); free(work->solution->y); free(work->solution); }
  free(work->settings); free(work->info); free(work);
}

/* ---- solver_interface.h ------------------------------------------------------------------------ */
/* mat_vec / mat_tpose_vec return void in the reference (solver_interface.h:33,47): a backend failure is made loud by
 * filling the result with NaN, a message on stderr and qpalm_backend_error() */
static void matvec_failed(solver_dense *y, size_t len) {
  snprintf(g_host_err, sizeof g_host_err, "mat_vec: %s", qpg_last_error());
  fprintf(stderr, "qpalm (gfx950 backend): %s\n", g_host_err);
  for (size_t k = 0; k < len; k++) ((c_float *)y->x)[k] = NAN;
}

static qpg_ctx *any_ctx(void) {
  static qpg_ctx *c = NULL;
  if (!c) { const char *dev = getenv("QPALM_GFX950_DEVICE"); if (qpg_ctx_create(dev ? atoi(dev) : 0, &c) != QPG_OK) c = NULL; }
  return c;
}

void mat_vec(solver_sparse *A, solver_dense *x, solver_dense *y, solver_common *c) { /* solver_interface.c:252-262 */
  (void)c;
  qpg_ctx *ctx = any_ctx();
  if (!ctx || qpg_sparse_matvec(ctx, (qpg_int)A->nrow, (qpg_int)A->ncol, (const qpg_int *)A->p, (const qpg_int *)A->i, (const qpg_float *)A->x,
                                A->stype, 0, (const qpg_float *)x->x, (qpg_float *)y->x) != QPG_OK)
    matvec_failed(y, A->stype ? A->ncol : A->nrow);
}
void mat_tpose_vec(solver_sparse *A, solver_dense *x, solver_dense *y, solver_common *c) { /* solver_interface.c:264-274 */
  (void)c;
  qpg_ctx *ctx = any_ctx();
  if (!ctx || qpg_sparse_matvec(ctx, (qpg_int)A->nrow, (qpg_int)A->ncol, (const qpg_int *)A->p, (const qpg_int *)A->i, (const qpg_float *)A->x,
                                A->stype, 1, (const qpg_float *)x->x, (qpg_float *)y->x) != QPG_OK)
    matvec_failed(y, A->ncol);
}
void mat_inf_norm_cols(solver_sparse *M, c_float *E) { /* solver_interface.c:276-292: host loop over the user's matrix */
  const c_int *Mp = (const c_int *)M->p; const c_float *Mx = (const c_float *)M->x;
  for (size_t j = 0; j < M->ncol; j++) { E[j] = 0.; for (c_int k = Mp[j]; k < Mp[j + 1]; k++) { c_float a = Mx[k] < 0 ? -Mx[k] : Mx[k]; if (a > E[j]) E[j] = a; } }
}
void mat_inf_norm_rows(solver_sparse *M, c_float *E) { /* solver_interface.c:294-314 */
  const c_int *Mp = (const c_int *)M->p, *Mi = (const c_int *)M->i; const c_float *Mx = (const c_float *)M->x;
  for (size_t j = 0; j < M->nrow; j++) E[j] = 0.;
  for (size_t j = 0; j < M->ncol; j++) for (c_int k = Mp[j]; k < Mp[j + 1]; k++) { c_float a = Mx[k] < 0 ? -Mx[k] : Mx[k]; if (a > E[Mi[k]]) E[Mi[k]] = a; }
}
/* solver_interface.c:20-75.  FACTORIZE_KKT and FACTORIZE_SCHUR are honoured.  FACTORIZE_KKT_OR_SCHUR: the reference compares
 * nnz(KKT) with an estimate of nnz(Q + A'A) for SPARSE factors; this backend keeps the factor as one dense panel per QP,
 * for which the n x n Schur panel is never larger than the (n+m) x (n+m) KKT panel, so the automatic choice is SCHUR
 * (as in the reference's CHOLMOD build, :72-74).  The backend batch is created with the same rule (qpg_batch_create). */
void qpalm_set_factorization_method(QPALMWorkspace *work, solver_common *c) {
  (void)c;
  work->solver->factorization_method = (work->settings->factorization_method == FACTORIZE_KKT) ? FACTORIZE_KKT : FACTORIZE_SCHUR;
}

static void push_solver_state(QPALMWorkspace *work) { /* what the boundary functions read from the workspace */
  const size_t n = work->data->n, m = work->data->m;
  setv(work, "dphi", work->dphi, n);
  qpg_batch_set_scalar(BT(work), "gamma", 0, work->gamma);
  qpg_batch_set_scalar(BT(work), "proximal", 0, (qpg_float)work->settings->proximal);
  if (m) {
    qpg_batch_set_ivector(BT(work), "active", 0, work->solver->active_constraints, (qpg_int)m);
    qpg_batch_set_ivector(BT(work), "enter", 0, work->solver->enter, (qpg_int)m);
    qpg_batch_set_ivector(BT(work), "leave", 0, work->solver->leave, (qpg_int)m);
    setv(work, "At_scale", (const c_float *)work->solver->At_scale->x, m);
  }
  qpg_batch_set_scalar(BT(work), "nb_enter", 0, (qpg_float)work->solver->nb_enter);
  qpg_batch_set_scalar(BT(work), "nb_leave", 0, (qpg_float)work->solver->nb_leave);
  qpg_batch_set_scalar(BT(work), "nb_sigma_changed", 0, (qpg_float)work->nb_sigma_changed);
}
void ldlchol(solver_sparse *M, QPALMWorkspace *work, solver_common *c) { /* solver_interface.c:319-370 */
  (void)c;
  push_solver_state(work);
  qpg_ldlchol_matrix(BT(work), 0, (qpg_int)M->ncol, (const qpg_int *)M->p, (const qpg_int *)M->i, (const qpg_float *)M->x);
}
void ldlcholQAtsigmaA(QPALMWorkspace *work, solver_common *c) { (void)c; push_solver_state(work); qpg_ldlcholQAtsigmaA(BT(work), 0); }
void ldlupdate_entering_constraints(QPALMWorkspace *work, solver_common *c) { (void)c; push_solver_state(work); qpg_ldlupdate_entering_constraints(BT(work), 0); }
void ldldowndate_leaving_constraints(QPALMWorkspace *work, solver_common *c) { (void)c; push_solver_state(work); qpg_ldldowndate_leaving_constraints(BT(work), 0); }
void ldlupdate_sigma_changed(QPALMWorkspace *work, solver_common *c) { (void)c; push_solver_state(work); qpg_ldlupdate_sigma_changed(BT(work), 0); }
/* ---- the KKT set (solver_interface.c:119-247), see include/qpalm_host.h ---------------------------------------------- */
static void push_kkt_state(QPALMWorkspace *work) {
  push_solver_state(work);
  if (work->data->m) setv(work, "sigma_inv", work->sigma_inv, work->data->m);
}
void qpalm_form_kkt(QPALMWorkspace *work) { push_kkt_state(work); qpg_kkt_form(BT(work), 0); }
void qpalm_reform_kkt(QPALMWorkspace *work) { push_kkt_state(work); qpg_kkt_form(BT(work), 0); }
void qpalm_kkt_factorize(QPALMWorkspace *work) { qpg_kkt_factorize(BT(work), 0); }
void kkt_update_entering_constraints(QPALMWorkspace *work, solver_common *c) { (void)c; push_kkt_state(work); qpg_kkt_update_entering_constraints(BT(work), 0); }
void kkt_update_leaving_constraints(QPALMWorkspace *work, solver_common *c) { (void)c; push_kkt_state(work); qpg_kkt_update_leaving_constraints(BT(work), 0); }
void kkt_solve(QPALMWorkspace *work, solver_common *c) { /* solver_interface.c:238-247: rhs = [-dphi; 0], d = sol_kkt[0 .. n) */
  (void)c;
  const size_t n = work->data->n;
  setv(work, "dphi", work->dphi, n);
  qpg_kkt_solve(BT(work), 0);
  getv(work, "d", work->d, n);
}
void ldlsolveLD_neg_dphi(QPALMWorkspace *work, solver_common *c) { /* solver_interface.c:505-519; d keeps its address */
  (void)c;
  const size_t n = work->data->n;
  setv(work, "dphi", work->dphi, n);
  qpg_ldlsolveLD_neg_dphi(BT(work), 0);
  getv(work, "d", work->d, n);
  for (size_t j = 0; j < n; j++) work->neg_dphi[j] = -work->dphi[j];
}
