/*
 * qpalm_iter.h -- the QPALM iteration as device code: one workgroup drives one QP through
 * scale_data (src/scaling.c:34-113), qpalm_warm_start (src/qpalm.c:322-399) and the loop body of
 * qpalm_solve (src/qpalm.c:484-711): compute_residuals (iteration.c:24-48), check_termination
 * (termination.c:19-240), the outer update (qpalm.c:515-645: update_sigma, gamma logic), and the
 * Newton step (newton.c:17-149, linesearch.c:14-120, iteration.c:213-229).
 *
 * Element-wise expressions are written in the reference's operation order (a + sc*b etc.) and this
 * translation unit is compiled with -ffp-contract=off, so they round exactly like the C reference.
 */
#ifndef QPALM_ITER_H
#define QPALM_ITER_H
#ifndef QP_NOFUSE
#define QP_NOFUSE 0 /* 1 (experiments): the Newton solve does not ride on the last update sweep */
#endif

#define QP_KIND_NEWTON 0
#define QP_KIND_OUTER 1
#define QP_KIND_FORCED 2
#define QP_KIND_TERMINATED 3

/* settings that set_settings_nonconvex (nonconvex.c:171-183) overrides per workspace: per-QP values for nonconvex QPs */
QPD int qp_prox(const qpg_settings &st, const qpg_scalars &s) { return (st.proximal != 0 || s.nc_flag != 0) ? 1 : 0; }
QPD double qp_gamma_init(const qpg_settings &st, const qpg_scalars &s) { return s.nc_flag ? s.nc_gamma : st.gamma_init; }
QPD double qp_gamma_max(const qpg_settings &st, const qpg_scalars &s) { return s.nc_flag ? s.nc_gamma : st.gamma_max; }

struct IterShared {
  QpShared S;
  qpg_scalars s;
  int kind, action, nL, pos;
  double a0, b0;
  double scan_a[QP_NW], scan_b[QP_NW];
};

struct QpArrays { /* per-QP views; base pointers are recomputed on use to keep SGPR pressure low */
  const qpg_view *V; int b, n, m;
  QPD const int *Ap() const { return V->Ap + (size_t)b * (V->n + 1); }
  QPD const int *Ai() const { return V->Ai + (size_t)b * V->nnzA; }
  QPD const int *Atp() const { return V->Atp + (size_t)b * (V->m + 1); }
  QPD const int *Ati() const { return V->Ati + (size_t)b * V->nnzA; }
  QPD const int *Atperm() const { return V->Atperm + (size_t)b * V->nnzA; }
  QPD const int *Qp() const { return V->Qp + (size_t)b * (V->n + 1); }
  QPD const int *Qi() const { return V->Qi + (size_t)b * V->nnzQ; }
  QPD const int *Qfp() const { return V->Qfp + (size_t)b * (V->n + 1); }
  QPD const int *Qfi() const { return V->Qfi + (size_t)b * V->nnzQf; }
  QPD const int *Qfperm() const { return V->Qfperm + (size_t)b * V->nnzQf; }
  QPD double *Ax() const { return V->Ax + (size_t)b * V->nnzA; }
  QPD double *Atx() const { return V->Atx + (size_t)b * V->nnzA; }
  QPD double *Atss() const { return V->Atss + (size_t)b * V->nnzA; }
  QPD double *Qx() const { return V->Qx + (size_t)b * V->nnzQ; }
  QPD double *Qfx() const { return V->Qfx + (size_t)b * V->nnzQf; }
  QPD double *q() const { return V->q + (size_t)b * V->n; }
  QPD double *x() const { return V->x + (size_t)b * V->n; }
  QPD double *Qxv() const { return V->Qxv + (size_t)b * V->n; }
  QPD double *Aty() const { return V->Aty + (size_t)b * V->n; }
  QPD double *x_prev() const { return V->x_prev + (size_t)b * V->n; }
  QPD double *x0() const { return V->x0 + (size_t)b * V->n; }
  QPD double *Atyh() const { return V->Atyh + (size_t)b * V->n; }
  QPD double *df() const { return V->df + (size_t)b * V->n; }
  QPD double *dphi() const { return V->dphi + (size_t)b * V->n; }
  QPD double *dphi_prev() const { return V->dphi_prev + (size_t)b * V->n; }
  QPD double *d() const { return V->d + (size_t)b * V->n; }
  QPD double *Qd() const { return V->Qd + (size_t)b * V->n; }
  QPD double *delta_x() const { return V->delta_x + (size_t)b * V->n; }
  QPD double *temp_n() const { return V->temp_n + (size_t)b * V->n; }
  QPD double *D() const { return V->D + (size_t)b * V->n; }
  QPD double *Dinv() const { return V->Dinv + (size_t)b * V->n; }
  QPD double *sol_x() const { return V->sol_x + (size_t)b * V->n; }
  QPD double *bmin() const { return V->bmin + (size_t)b * V->m; }
  QPD double *bmax() const { return V->bmax + (size_t)b * V->m; }
  QPD double *y() const { return V->y + (size_t)b * V->m; }
  QPD double *Axv() const { return V->Axv + (size_t)b * V->m; }
  QPD double *sigma() const { return V->sigma + (size_t)b * V->m; }
  QPD double *sigma_inv() const { return V->sigma_inv + (size_t)b * V->m; }
  QPD double *sqrt_sigma() const { return V->sqrt_sigma + (size_t)b * V->m; }
  QPD double *At_scale() const { return V->At_scale + (size_t)b * V->m; }
  QPD double *Axys() const { return V->Axys + (size_t)b * V->m; }
  QPD double *z() const { return V->z + (size_t)b * V->m; }
  QPD double *pri_res() const { return V->pri_res + (size_t)b * V->m; }
  QPD double *pri_res_in() const { return V->pri_res_in + (size_t)b * V->m; }
  QPD double *yh() const { return V->yh + (size_t)b * V->m; }
  QPD double *Ad() const { return V->Ad + (size_t)b * V->m; }
  QPD double *delta_y() const { return V->delta_y + (size_t)b * V->m; }
  QPD double *temp_m() const { return V->temp_m + (size_t)b * V->m; }
  QPD double *E() const { return V->E + (size_t)b * V->m; }
  QPD double *Einv() const { return V->Einv + (size_t)b * V->m; }
  QPD double *sol_y() const { return V->sol_y + (size_t)b * V->m; }
  QPD double *ls_key() const { return V->ls_key + (size_t)b * V->ls_stride; }
  QPD int *ls_idx() const { return V->ls_idx + (size_t)b * V->ls_stride; }
  QPD double *ls_delta() const { return V->ls_delta + (size_t)b * (2 * V->m); }
  QPD double *ls_alpha() const { return V->ls_alpha + (size_t)b * (2 * V->m); }
  QPD int *active() const { return V->active + (size_t)b * V->m; }
  QPD int *active_old() const { return V->active_old + (size_t)b * V->m; }
  QPD int *enter() const { return V->enter + (size_t)b * V->m; }
  QPD int *leave() const { return V->leave + (size_t)b * V->m; }
};

QPD QpArrays qp_arrays(const qpg_view &V, int b) {
  QpArrays a;
  a.V = &V; a.b = b;
  a.n = V.nq ? QP_UNIFORM(V.nq[b]) : V.n; a.m = V.mq ? QP_UNIFORM(V.mq[b]) : V.m; /* strides stay V.n / V.m */
  return a;
}

/* =============================================================================================
 * scale_data (src/scaling.c:34-113) and the derived copies (A' values, full-symmetric Q values)
 * =========================================================================================== */
QPN void dev_fill_derived(const qpg_view &V, const QpArrays &a, int b) {
  const int nzA = a.Ap()[a.n], nzQf = a.Qfp()[a.n];
  for (int k = threadIdx.x; k < nzA; k += QP_T) a.Atx()[k] = a.Ax()[a.Atperm()[k]];
  for (int k = threadIdx.x; k < nzQf; k += QP_T) a.Qfx()[k] = a.Qx()[a.Qfperm()[k]];
  __syncthreads();
}

QPN void dev_scale_data(const qpg_view &V, const QpArrays &a, int b, int nscale, IterShared &I) {
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double *D_temp = a.temp_n(), *E_temp = a.temp_m();
  for (int j = tid; j < n; j += QP_T) a.D()[j] = 1.0;
  for (int i = tid; i < m; i += QP_T) a.E()[i] = 1.0;
  __syncthreads();
  for (int it = 0; it < nscale; it++) {
    /* column / row infinity norms of A (solver_interface.c:276-314) */
    for (int j = tid; j < n; j += QP_T) {
      double e = 0.0;
      for (int k = a.Ap()[j]; k < a.Ap()[j + 1]; k++) e = qmax(qabs(a.Ax()[k]), e);
      e = e < 1e-12 ? 1.0 : e;            /* limit_scaling, scaling.c:25-31 */
      D_temp[j] = 1.0 / QP_SQRT(e);
    }
    for (int i = tid; i < m; i += QP_T) {
      double e = 0.0;
      for (int k = a.Atp()[i]; k < a.Atp()[i + 1]; k++) e = qmax(qabs(a.Ax()[a.Atperm()[k]]), e);
      e = e < 1e-12 ? 1.0 : e;
      E_temp[i] = 1.0 / QP_SQRT(e);
    }
    __syncthreads();
    for (int j = tid; j < n; j += QP_T) {   /* A <- diag(E_temp) A, then A <- A diag(D_temp) */
      const double dj = D_temp[j];
      for (int k = a.Ap()[j]; k < a.Ap()[j + 1]; k++) { double v = a.Ax()[k]; v *= E_temp[a.Ai()[k]]; v *= dj; a.Ax()[k] = v; }
      a.D()[j] = a.D()[j] * dj;
    }
    for (int i = tid; i < m; i += QP_T) a.E()[i] = a.E()[i] * E_temp[i];
    __syncthreads();
  }
  /* q <- D q ; Qx <- D Qx ; c = 1/max(1, ||Qx + q||inf) ; q <- c q   (scaling.c:83-89) */
  double vm[1] = {0.0}, vs[1] = {0.0};
  for (int j = tid; j < n; j += QP_T) {
    const double qj = a.D()[j] * a.q()[j];
    const double Qxj = a.D()[j] * a.Qxv()[j];
    a.q()[j] = qj; a.Qxv()[j] = Qxj;
    const double dp = Qxj + 1 * qj;
    a.dphi()[j] = dp;
    vm[0] = qmax(vm[0], qabs(dp));
  }
  block_reduce<1, 0>(I.S, vm, vs);
  const double c = 1 / qmax(1.0, vm[0]);
  for (int j = tid; j < n; j += QP_T) {
    a.q()[j] *= c;
    const double t = a.D()[j];
    for (int k = a.Qp()[j]; k < a.Qp()[j + 1]; k++) { double v = a.Qx()[k]; v *= t * a.D()[a.Qi()[k]]; v *= c; a.Qx()[k] = v; }
    a.Dinv()[j] = 1.0 / a.D()[j];
  }
  for (int i = tid; i < m; i += QP_T) {
    a.Einv()[i] = 1.0 / a.E()[i];
    a.bmin()[i] = a.E()[i] * a.bmin()[i];
    a.bmax()[i] = a.E()[i] * a.bmax()[i];
  }
  if (tid == 0) { I.s.sc_c = c; I.s.sc_cinv = 1.0 / c; }
  __syncthreads();
}

/* compute_objective, iteration.c:231-270 (the CPU groups terms by four; here a fixed tree) */
QPN double dev_objective(const qpg_view &V, const QpArrays &a, IterShared &I, double c0) {
  const qpg_settings &st = *V.settings;
  double vm[1] = {0.0}, vs[1] = {0.0};
  const double g = I.s.gamma;
  for (int j = threadIdx.x; j < a.n; j += QP_T) {
    if (qp_prox(st, I.s)) vs[0] += (0.5 * (a.Qxv()[j] - 1 / g * a.x()[j]) + a.q()[j]) * a.x()[j];
    else vs[0] += (0.5 * a.Qxv()[j] + a.q()[j]) * a.x()[j];
  }
  block_reduce<0, 1>(I.S, vm, vs);
  double obj = vs[0];
  if (I.s.has_scaling) obj *= I.s.sc_cinv;
  obj += c0;
  return obj;
}

/* qpalm_warm_start (qpalm.c:322-399) + initialize_sigma (iteration.c:50-84).
 * The (unscaled) warm-start vectors were placed in x / y by the host; has_x / has_y say which. */
QPN void dev_warm_start(const qpg_view &V, const QpArrays &a, int b, int has_x, int has_y, IterShared &I) {
  const qpg_settings &st = *V.settings;
  const int n = a.n, m = a.m, tid = threadIdx.x;
  if (tid == 0) I.s.gamma = qp_gamma_init(st, I.s);
  __syncthreads();
  if (has_x) {
    for (int j = tid; j < n; j += QP_T) {
      double xv = (has_x == 2) ? a.sol_x()[j] : a.x()[j]; /* 2: this QP's last stored solution (qpg_batch_warm_start_last) */
      if (I.s.has_scaling) xv = xv * a.Dinv()[j];
      a.x()[j] = xv; a.x0()[j] = xv; a.x_prev()[j] = xv;
    }
    __syncthreads();
    const double ginv = 1 / qp_gamma_init(st, I.s);
    const int prox = qp_prox(st, I.s);
    spmv_rows<8>(n, a.Qfp(), a.Qfi(), a.Qfx(), a.x(), [&](int r, double s) {
      a.Qd()[r] = s;
      a.Qxv()[r] = prox ? (s + ginv * a.x()[r]) : s;
    });
    spmv_rows<8>(m, a.Atp(), a.Ati(), a.Atx(), a.x(), [&](int r, double s) { a.Ad()[r] = s; a.Axv()[r] = s; });
    __syncthreads();
    const double obj = dev_objective(V, a, I, V.c0[b]);
    if (tid == 0) I.s.objective = obj;
  } else {
    for (int j = tid; j < n; j += QP_T) { a.x()[j] = 0.; a.x_prev()[j] = 0.; a.x0()[j] = 0.; a.Qxv()[j] = 0.; }
    for (int i = tid; i < m; i += QP_T) a.Axv()[i] = 0.;
    if (tid == 0) I.s.objective = 0.0;
  }
  if (has_y) {
    for (int i = tid; i < m; i += QP_T) {
      double yv = (has_y == 2) ? a.sol_y()[i] : a.y()[i];
      if (I.s.has_scaling) { yv = yv * a.Einv()[i]; yv *= I.s.sc_c; }
      a.y()[i] = yv;
    }
  } else {
    for (int i = tid; i < m; i += QP_T) a.y()[i] = 0.;
  }
  __syncthreads();
  /* initialize_sigma */
  double vm[1] = {0.0}, vs[3] = {0.0, 0.0, 0.0};
  for (int j = tid; j < n; j += QP_T) { vs[0] += a.x()[j] * a.Qxv()[j]; vs[1] += a.q()[j] * a.x()[j]; }
  for (int i = tid; i < m; i += QP_T) {
    const double ax = a.Axv()[i];
    const double mid = qmax(a.bmin()[i], qmin(ax, a.bmax()[i]));
    const double t = ax + (-1) * mid;
    vs[2] += t * t;
  }
  block_reduce<0, 3>(I.S, vm, vs);
  const double f = 0.5 * vs[0] + vs[1], dist2 = vs[2];
  const double sig = qmax(1e-4, qmin(st.sigma_init * qmax(1, qabs(f)) / qmax(1, 0.5 * dist2), 1e4));
  const double ssig = QP_SQRT(sig), sinv = 1.0 / sig;
  for (int i = tid; i < m; i += QP_T) {
    a.sigma()[i] = sig; a.sigma_inv()[i] = sinv; a.sqrt_sigma()[i] = ssig; a.At_scale()[i] = ssig;
    for (int k = a.Atp()[i]; k < a.Atp()[i + 1]; k++) a.Atss()[k] = a.Atx()[k] * ssig;
  }
  if (tid == 0) { I.s.sqrt_sigma_max = QP_SQRT(st.sigma_max); I.s.initialized = 1; }
  __syncthreads();
}

/* =============================================================================================
 * factorisation plumbing
 * =========================================================================================== */
template <int RPT>
QPN void dev_factor(const qpg_view &V, int n, double *L, double *Dg, char *lds, int64_t *tdbg) { dense_factor<RPT>(L, Dg, n, V.ld, lds, tdbg); }
/* RPT = rows of the factor per thread in the update sweep (registers); RPT == 0 is the large-factor form (more than
 * 4 QP_T rows: the running vectors live in HBM, dense_updown_big) */
template <int RPT>
QPN void dev_updown(const qpg_view &V, int b, int n, double *L, double *Dg, double *Wst, const int *up, int n_up,
                    const int *dn, int n_dn, QpShared &S, char *lds, int64_t *tdbg, double *fs = nullptr, int pre_jmin = -1) {
  const int *Atp = V.Atp + (size_t)b * (V.m + 1), *Ati = V.Ati + (size_t)b * V.nnzA;
  const double *Atss = V.Atss + (size_t)b * V.nnzA;
  /* how the sweeps sum a column's pivots (qp_pivot_mode, qpalm_device.h): the running pivot of the reference throughout for QPs whose factor can get
   * near-singular (nonconvex: indefinite; an LP or a Q with an empty diagonal: pivots down to 1 / gamma), else the guarded prefix tree */
  __syncthreads();
  if (threadIdx.x == 0) S.seq_ranks = qp_pivot_mode(V, b);
  __syncthreads();
  if constexpr (RPT == 0) dense_updown_big<16>(Atp, Ati, Atss, n, V.ld, L, Dg, Wst, up, n_up, dn, n_dn, &S, lds, tdbg, pre_jmin);
  else if constexpr (QP_K32(RPT)) {
    /* up to 32 ranks per sweep: the multi-pass form takes the ranks in sweeps of 32 as long as more than 16 are left (17..32 ranks:
     * one sweep over the panel instead of two), the 16-rank form the rest.  Bit-identical factors whichever form runs. */
    const int nr = n_up + n_dn;
    int n32 = 0;
    if (V.sweep_ranks >= 32 && pre_jmin < 0 && nr > 16) n32 = (nr % 32 > 16 || nr % 32 == 0) ? nr : nr - nr % 32;
    if (n32 > 0) dense_updown<1, 32>(Atp, Ati, Atss, n, V.ld, L, Dg, Wst, up, n_up, dn, n_dn, &S, lds, tdbg, -1, (n32 == nr) ? fs : nullptr, 0, n32);
    if (n32 < nr) dense_updown<RPT, QP_KSEL(RPT)>(Atp, Ati, Atss, n, V.ld, L, Dg, Wst, up, n_up, dn, n_dn, &S, lds, tdbg, pre_jmin, fs, n32, nr - n32);
  } else dense_updown<RPT, QP_KSEL(RPT)>(Atp, Ati, Atss, n, V.ld, L, Dg, Wst, up, n_up, dn, n_dn, &S, lds, tdbg, pre_jmin, fs);
}

/* =============================================================================================
 * update_sigma (iteration.c:86-145) + ldlupdate_sigma_changed (solver_interface.c:443-503)
 * =========================================================================================== */
/* First half of ldlupdate_sigma_changed (solver_interface.c:455-460,492): the changed rows (listed in a.enter()) get
 * At_scale <- sqrt(1 - 1/At_scale^2), then every column of At_sqrt_sigma is scaled by At_scale (1.0 = untouched).
 * One copy, used by the iteration loop and by the boundary operation qpg_ldlupdate_sigma_changed. */
QPN void dev_ldlupdate_sigma_scale(const QpArrays &a, int nchg) {
  const int m = a.m, tid = threadIdx.x;
  __syncthreads();
  for (int k = tid; k < nchg; k += QP_T) {
    const int row = a.enter()[k];
    double s = a.At_scale()[row];
    s = s * s;
    s = QP_SQRT(1 - 1 / s);
    a.At_scale()[row] = s;
  }
  __syncthreads();
  for (int k = tid; k < m; k += QP_T) {
    const double s = a.At_scale()[k];
    if (s != 1.0) for (int e = a.Atp()[k]; e < a.Atp()[k + 1]; e++) a.Atss()[e] *= s;
  }
  __syncthreads();
}

/* sparse factor: a rank-1 update walks the row's elimination-tree path (about `nlev` dependent steps); cheaper than rebuilding the factor while
 * 2 nchange nlev < n (qpalm_sparse.h: sp_updown) */
QPD bool sp_update_pays(int nchange, int nlev, int n) { return (long long)nchange * (long long)nlev * 2 < (long long)n; }
/* Part 1: new sigma, rescaled At_sqrt_sigma, list of changed rows (in a.enter()).  Returns the number
 * of rank-1 updates ldlupdate_sigma_changed has to apply (0: nothing to do or a refactorisation was
 * requested); the update itself runs at dev_solve's single linear-algebra site, then part 2. */
QPP int dev_update_sigma_pre(const qpg_view &V, const QpArrays &a, IterShared &I) {
  const qpg_settings &st = *V.settings;
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double vm[1] = {0.0}, vs[1] = {0.0};
  for (int i = tid; i < m; i += QP_T) vm[0] = qmax(vm[0], qabs(a.pri_res()[i]));
  block_reduce<1, 0>(I.S, vm, vs);
  const double prn = vm[0];
  int *changed_flag = a.ls_idx(); /* scratch */
  for (int k = tid; k < m; k += QP_T) {
    int chg = 0;
    if ((qabs(a.pri_res()[k]) > st.theta * qabs(a.pri_res_in()[k])) && a.active()[k]) {
      double mult_factor = qmax(1.0, st.delta * qabs(a.pri_res()[k]) / (prn + 1e-6));
      const double sigma_temp = mult_factor * a.sigma()[k];
      if (sigma_temp <= st.sigma_max) {
        if (a.sigma()[k] != sigma_temp) chg = 1;
        a.sigma()[k] = sigma_temp;
        a.sigma_inv()[k] = 1.0 / sigma_temp;
        mult_factor = QP_SQRT(mult_factor);
        a.sqrt_sigma()[k] = mult_factor * a.sqrt_sigma()[k];
        a.At_scale()[k] = mult_factor;
      } else {
        if (a.sigma()[k] != st.sigma_max) chg = 1;
        a.sigma()[k] = st.sigma_max;
        a.sigma_inv()[k] = 1.0 / st.sigma_max;
        a.At_scale()[k] = I.s.sqrt_sigma_max / a.sqrt_sigma()[k];
        a.sqrt_sigma()[k] = I.s.sqrt_sigma_max;
      }
    } else a.At_scale()[k] = 1.0;
    changed_flag[k] = chg;
    const double sc = a.At_scale()[k];
    if (sc != 1.0) for (int e = a.Atp()[k]; e < a.Atp()[k + 1]; e++) a.Atss()[e] *= sc;
  }
  __syncthreads();
  int nchg = 0, dummy = 0;
  block_compact2(I.S, m, [&](int i) { return changed_flag[i]; }, [&](int i) { return 0; }, a.enter(), a.leave(), nchg, dummy);
  if (tid == 0) { I.s.nb_sigma_changed = nchg; I.s.n_sigma_updates++; }
  __syncthreads();
  double thr = qmin(st.max_rank_update_fraction * (double)(n + m), 0.25 * (double)st.max_rank_update);
  if (V.offload && V.update_rank_threshold >= 0) thr = qmin(thr, (double)V.update_rank_threshold); /* coop mode: beyond its threshold the factor is rebuilt by many workgroups
                                                                        instead of updated by one (speed policy, same matrix) */
  int nupd = 0;
  if (V.sparse) {
    /* sparse factor (qpalm_sparse.h): ldlupdate_sigma_changed as the reference has it (solver_interface.c:443-503) -- rank-1 updates with the scaled rows along
     * their elimination-tree paths -- where walking the paths is cheaper than rebuilding (sp_update_pays, the rule of the entering / leaving rows); else the
     * factor is marked stale and the next Newton step refactorises.  (Through round 5 every change of sigma refactorised here.) */
    const bool pays = sp_update_pays(nchg, V.sp_nlev[a.b], n);
    if ((qp_prox(st, I.s) && I.s.gamma < qp_gamma_max(st, I.s)) || ((double)nchg > thr) || (nchg > 0 && !pays)) {
      if (tid == 0) I.s.reset_newton = 1;
    } else if (nchg > 0) {
      dev_ldlupdate_sigma_scale(a, nchg);
      nupd = nchg;
    }
    __syncthreads();
    return nupd;
  }
  if (V.kkt) {
    /* FACTORIZE_KKT (iteration.c:135-144, solver_interface.c:463-481): every branch that changes anything ends in
     * reset_newton = TRUE; the reference's rank-1 correction is applied at row pinv[row] (a variable's row) and is
     * overwritten by the refactorisation that reset_newton forces, so it is not restated */
    if (I.s.kkt_first || (qp_prox(st, I.s) && I.s.gamma < qp_gamma_max(st, I.s)) || nchg > 0) { if (tid == 0) I.s.reset_newton = 1; }
    __syncthreads();
    return 0;
  }
  if ((qp_prox(st, I.s) && I.s.gamma < qp_gamma_max(st, I.s)) || ((double)nchg > thr)) {
    if (tid == 0) I.s.reset_newton = 1;
  } else if (nchg == 0) {
  } else {
    dev_ldlupdate_sigma_scale(a, nchg);
    nupd = nchg;
  }
  __syncthreads();
  return nupd;
}
/* Part 2, after the rank updates: At_sqrt_sigma back to sqrt(sigma) scaling (solver_interface.c:498-502) */
QPP void dev_update_sigma_post(const QpArrays &a, IterShared &I, int nchg) {
  const int m = a.m, tid = threadIdx.x;
  for (int k = tid; k < m; k += QP_T) {
    const double s0 = a.At_scale()[k];
    const double s = 1.0 / s0;
    a.At_scale()[k] = s;
    if (s0 == 0.0) {
      /* sigma_k grew by one unit in the last place: sqrt(mult_factor) rounds to 1, the row's update vector sqrt(1 - 1/1) A_k is
       * exactly zero (a no-op sweep, as it should be) -- and scaling the zeroed row back by 1/0 gives 0 * inf = NaN in the
       * reference's CHOLMOD branch (solver_interface.c:492-502; its LADEL branch never scales At).  The row is rebuilt from A'
       * instead (found by the fresh-seed fuzz campaign of round 4, seed 204 case 155; the oracle restates the same repair). */
      const double ssig = a.sqrt_sigma()[k];
      for (int e = a.Atp()[k]; e < a.Atp()[k + 1]; e++) a.Atss()[e] = a.Atx()[e] * ssig;
    } else if (s != 1.0) for (int e = a.Atp()[k]; e < a.Atp()[k + 1]; e++) a.Atss()[e] *= s;
  }
  if (tid == 0) { I.s.n_rank1 += nchg; }
  __syncthreads();
}

/* update_gamma (iteration.c:147-156) */
QPP void dev_update_gamma(const qpg_view &V, const QpArrays &a, IterShared &I) {
  const qpg_settings &st = *V.settings;
  __syncthreads();
  if (I.s.gamma < qp_gamma_max(st, I.s)) {
    const double prev = I.s.gamma;
    const double g = qmin(prev * st.gamma_upd, qp_gamma_max(st, I.s));
    const double sc = 1 / g - 1 / prev;
    for (int j = threadIdx.x; j < a.n; j += QP_T) a.Qxv()[j] = a.Qxv()[j] + sc * a.x()[j];
    __syncthreads();
    if (threadIdx.x == 0) { I.s.gamma = g; I.s.reset_newton = 1; }
  }
  __syncthreads();
}

/* set_active_constraints + set_entering_leaving_constraints (newton.c:122-149) */
QPPH void dev_active_sets(const QpArrays &a, IterShared &I) {
  const int m = a.m;
  int cnt = 0;
  for (int i = threadIdx.x; i < m; i += QP_T) {
    const int act = ((a.Axys()[i] <= a.bmin()[i]) || (a.Axys()[i] >= a.bmax()[i])) ? 1 : 0;
    a.active()[i] = act;
    cnt += act;
  }
  cnt = block_isum(I.S, cnt);
  int ne = 0, nl = 0;
  block_compact2(I.S, m, [&](int i) { return a.active()[i] && !a.active_old()[i]; },
                 [&](int i) { return !a.active()[i] && a.active_old()[i]; }, a.enter(), a.leave(), ne, nl);
  if (threadIdx.x == 0) { I.s.nb_active = cnt; I.s.nb_enter = ne; I.s.nb_leave = nl; }
  __syncthreads();
}

/* boost_gamma (iteration.c:158-211); `ub` = Gershgorin bound of A' Sigma_active A, formed at dev_solve's
 * single linear-algebra site when there are active constraints */
QPP void dev_boost_gamma_apply(const qpg_view &V, const QpArrays &a, IterShared &I, double ub) {
  const qpg_settings &st = *V.settings;
  const double prev = I.s.gamma;
  double g;
  if (I.s.nb_active) {
    g = V.kkt ? 1e10 : qmax(qp_gamma_max(st, I.s), 1e14 / ub); /* iteration.c:173-176 under FACTORIZE_KKT */
    if (threadIdx.x == 0) I.s.gamma_maxed = 1;
  } else g = 1e12;
  __syncthreads();
  if (prev != g) {
    const double s1 = 1.0 / g - 1.0 / prev, s2 = I.s.tau / g - I.s.tau / prev;
    for (int j = threadIdx.x; j < a.n; j += QP_T) {
      a.Qxv()[j] = a.Qxv()[j] + s1 * a.x()[j];
      a.Qd()[j] = a.Qd()[j] + s2 * a.d()[j];
    }
    if (threadIdx.x == 0) I.s.reset_newton = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) { I.s.gamma = g; I.s.n_boost_gamma++; }
  __syncthreads();
}

/* store_solution (termination.c:242-252); B12: yh is rescaled in place */
QPP void dev_store_solution(const qpg_view &V, const QpArrays &a, int b, IterShared &I) {
  __syncthreads();
  if (I.s.has_scaling) {
    for (int j = threadIdx.x; j < a.n; j += QP_T) a.sol_x()[j] = a.x()[j] * a.D()[j];
    for (int i = threadIdx.x; i < a.m; i += QP_T) { const double v = a.yh()[i] * I.s.sc_cinv; a.yh()[i] = v; a.sol_y()[i] = v * a.E()[i]; }
  } else {
    for (int j = threadIdx.x; j < a.n; j += QP_T) a.sol_x()[j] = a.x()[j];
    for (int i = threadIdx.x; i < a.m; i += QP_T) a.sol_y()[i] = a.yh()[i];
  }
  const double obj = dev_objective(V, a, I, V.c0[b]);
  if (threadIdx.x == 0) I.s.objective = obj;
  __syncthreads();
}

/* =============================================================================================
 * exact_linesearch (linesearch.c:14-120): breakpoints -> LDS bitonic sort -> prefix scan
 * =========================================================================================== */
QPD bool ls_greater(double ka, int ia, double kb, int ib) { return (ka > kb) || (ka == kb && ia > ib); }

QPPH double dev_linesearch(const qpg_view &V, const QpArrays &a, IterShared &I, char *lds) {
  const qpg_settings &st = *V.settings;
  const int n = a.n, m = a.m, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double ginv = 1 / I.s.gamma;
  const int prox = qp_prox(st, I.s);
  __syncthreads();
  long long tl0 = QP_CLOCK();
  /* Qd = Q d (+ d / gamma), Ad = A d: d is gathered from LDS (8 n bytes of the block the sort uses afterwards) */
  const qp_lds_cdouble ds = spmv_stage_x(a.d(), n, lds, V.lds_bytes);
  __syncthreads();
  if (QP_UNIFORM((int)(ds != nullptr))) {
    spmv_rows<8>(n, a.Qfp(), a.Qfi(), a.Qfx(), ds, [&](int r, double s) { a.Qd()[r] = prox ? (s + ginv * ds[r]) : s; });
    spmv_rows<8>(m, a.Atp(), a.Ati(), a.Atx(), ds, [&](int r, double s) { a.Ad()[r] = s; });
  } else {
    spmv_rows<8>(n, a.Qfp(), a.Qfi(), a.Qfx(), (const double *)a.d(), [&](int r, double s) { a.Qd()[r] = prox ? (s + ginv * a.d()[r]) : s; });
    spmv_rows<8>(m, a.Atp(), a.Ati(), a.Atx(), (const double *)a.d(), [&](int r, double s) { a.Ad()[r] = s; });
  }
  __syncthreads();
  if (tid == 0) { const long long t = QP_CLOCK(); I.s.ticks_dbg[13] += t - tl0; tl0 = t; } /* 13: line-search SpMVs */
  double vm[1] = {0.0}, vs[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = tid; j < n; j += QP_T) { vs[0] += a.d()[j] * a.Qd()[j]; vs[1] += a.d()[j] * a.df()[j]; }
  /* delta, alpha, s = alpha/delta for the 2m breakpoints; J = L xor P sums */
  for (int i = tid; i < m; i += QP_T) {
    const double ss = a.sqrt_sigma()[i], sg = a.sigma()[i], ax = a.Axv()[i], yv = a.y()[i];
    const double tmp = ss * a.Ad()[i];
    const double dhi = tmp, dlo = tmp * -1;
    double t = ax + (-1) * a.bmin()[i]; t = sg * t; t = yv + 1 * t; const double alo = t / ss;
    t = a.bmax()[i] + (-1) * ax; t = sg * t; t = t + (-1) * yv; const double ahi = t / ss;
    const double slo = alo / dlo, shi = ahi / dhi;
    a.ls_delta()[i] = dlo; a.ls_delta()[m + i] = dhi; a.ls_alpha()[i] = alo; a.ls_alpha()[m + i] = ahi;
    a.ls_key()[i] = slo; a.ls_key()[m + i] = shi;
    const int Llo = slo > 0, Lhi = shi > 0, Plo = dlo > 0, Phi = dhi > 0;
    if ((Plo + Llo) == 1) { vs[2] += dlo * dlo; vs[3] += dlo * alo; }
    if ((Phi + Lhi) == 1) { vs[2] += dhi * dhi; vs[3] += dhi * ahi; }
  }
  block_reduce<0, 4>(I.S, vm, vs);
  const double eta = vs[0], beta = vs[1];
  const double a0 = eta + vs[2], b0 = beta - vs[3];
  /* compaction of L = {s > 0} into the sort buffer */
  int P2 = 1;
  while (P2 < 2 * m) P2 <<= 1;
  /* The sort buffer is LDS whenever 12 bytes per (power-of-two padded) breakpoint fit, else HBM.  The
   * code is instantiated once per case so that the pointers have a KNOWN address space: a run-time
   * select between the two makes them generic, and flat accesses to LDS made this phase 10x slower. */
  const bool in_lds = !V.ls_hbm && ((size_t)P2 * 12 <= (size_t)V.lds_bytes);
  int nL = 0, mypos = 0x7fffffff;
  double mytau = 0.0, ta = 0.0, tb = 0.0;
  auto sort_and_scan = [&](auto keys, auto idx, auto tiled) QP_ALWAYS_INLINE {
  {
    int base = 0;
    for (int e0 = 0; e0 < 2 * m; e0 += QP_T) {
      const int e = e0 + tid;
      const double kv = (e < 2 * m) ? a.ls_key()[e] : 0.0;
      const int p = (e < 2 * m) && (kv > 0);
      const unsigned long long bal = __ballot(p);
      const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pre = __popcll(bal & below);
      __syncthreads();
      if (lane == 0) I.S.ired[wid][2] = __popcll(bal);
      __syncthreads();
      int woff = 0, tot = 0;
      for (int w = 0; w < QP_NW; w++) { if (w < wid) woff += I.S.ired[w][2]; tot += I.S.ired[w][2]; }
      /* in-place compaction into a.ls_key() is safe: destination index <= source index and every
       * source of this chunk was read before the barrier above */
      if (p) { keys[base + woff + pre] = kv; idx[base + woff + pre] = e; }
      base += tot;
    }
    nL = base;
  }
  __syncthreads();
  int P = 1;
  while (P < nL) P <<= 1;
  if (P < 2) P = 2;
  for (int e = nL + tid; e < P; e += QP_T) { keys[e] = 1e300 * 1e300; idx[e] = 0x7fffffff; }
  __syncthreads();
  if (tid == 0) { const long long t = QP_CLOCK(); I.s.ticks_dbg[14] += t - tl0; tl0 = t; } /* 14: breakpoints + compaction */
  /* bitonic sort ascending by (key, idx) */
  if (decltype(tiled)::value) {
    /* the buffer is in HBM (more than lds_bytes / 12 breakpoints: m > 3200 or so): every compare-exchange distance below the tile size
     * stays inside a tile, so a tile (as many entries as the LDS holds) is loaded once, taken through ALL those steps in LDS and
     * stored once -- for 131 072 breakpoints 21 passes over the buffer instead of 153.  The network, hence the result, is the same. */
    int TS = 2;
    while ((size_t)(TS * 2) * 12 <= (size_t)V.lds_bytes && TS * 2 <= P && (V.ls_hbm < 2 || TS * 2 <= V.ls_hbm)) TS <<= 1; /* (ls_hbm >= 2: a test's cap on the tile) */
    double QP_LDS_AS *tk = QP_LDS_ARG(double, lds);
    int QP_LDS_AS *ti = QP_LDS_ARG(int, lds + (size_t)TS * 8);
    auto tile_pass = [&](const int kfrom, const int kto) QP_ALWAYS_INLINE {
      for (int t0 = 0; t0 < P; t0 += TS) {
        for (int e = tid; e < TS; e += QP_T) { tk[e] = keys[t0 + e]; ti[e] = idx[t0 + e]; }
        __syncthreads();
        for (int k = kfrom; k <= kto; k <<= 1) {
          const int j0 = ((k >> 1) < (TS >> 1)) ? (k >> 1) : (TS >> 1);
          for (int j = j0, lj = 31 - __builtin_clz(j0); j > 0; j >>= 1, lj--) {
            for (int e = tid; e < (TS >> 1); e += QP_T) {
              const int i = ((e >> lj) << (lj + 1)) + (e & (j - 1)), p = i + j;
              const bool up = (((t0 + i) & k) == 0);
              const double ki = tk[i], kp = tk[p];
              const int ii = ti[i], ip = ti[p];
              if (ls_greater(ki, ii, kp, ip) == up) { tk[i] = kp; tk[p] = ki; ti[i] = ip; ti[p] = ii; }
            }
            if (j > 32) __syncthreads(); else QP_WAVE_SYNC();
          }
          __syncthreads();
        }
        for (int e = tid; e < TS; e += QP_T) { keys[t0 + e] = tk[e]; idx[t0 + e] = ti[e]; }
        __syncthreads();
      }
    };
    tile_pass(2, TS);
    for (int k = TS << 1; k <= P; k <<= 1) {
      for (int j = k >> 1, lj = 31 - __builtin_clz(k >> 1); j >= TS; j >>= 1, lj--) {
        for (int e = tid; e < (P >> 1); e += QP_T) {
          const int i = ((e >> lj) << (lj + 1)) + (e & (j - 1)), p = i + j;
          const bool up = ((i & k) == 0);
          const double ki = keys[i], kp = keys[p];
          const int ii = idx[i], ip = idx[p];
          if (ls_greater(ki, ii, kp, ip) == up) { keys[i] = kp; keys[p] = ki; idx[i] = ip; idx[p] = ii; }
        }
        __syncthreads();
      }
      tile_pass(k, k);
    }
  } else
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1, lj = 31 - __builtin_clz(k >> 1); j > 0; j >>= 1, lj--) { /* j = 1 << lj: shifts, no integer division */
      for (int e = tid; e < (P >> 1); e += QP_T) {
        const int i = ((e >> lj) << (lj + 1)) + (e & (j - 1)), p = i + j;
        const bool up = ((i & k) == 0);
        const double ki = keys[i], kp = keys[p];
        const int ii = idx[i], ip = idx[p];
        if (ls_greater(ki, ii, kp, ip) == up) { keys[i] = kp; keys[p] = ki; idx[i] = ip; idx[p] = ii; }
      }
      /* partners closer than 64 stay inside the 128 elements a wavefront owns in each round */
      if (j > 32) __syncthreads(); else QP_WAVE_SYNC();
    }
    __syncthreads();
  }
  if (tid == 0) { const long long t = QP_CLOCK(); I.s.ticks_dbg[15] += t - tl0; tl0 = t; } /* 15: sort */
  /* running (a, b): element i contributes (+d^2, -d*alpha) if delta > 0 else (-d^2, +d*alpha) once
   * passed; tau = -b/a at the first sorted breakpoint where a*s + b > 0 (linesearch.c:90-118) */
  const int CH = (nL + QP_T - 1) / QP_T > 0 ? (nL + QP_T - 1) / QP_T : 1;
  const int e0 = tid * CH, e1 = (e0 + CH < nL) ? e0 + CH : nL;
  double la = 0.0, lb = 0.0;
  for (int e = e0; e < e1; e++) {
    const int iz = idx[e];
    const double dl = a.ls_delta()[iz], al = a.ls_alpha()[iz];
    if (dl > 0) { la += dl * dl; lb -= dl * al; } else { la -= dl * dl; lb += dl * al; }
  }
  /* exclusive scan of (la, lb) over threads */
  double ia = la, ib = lb;
  for (int o = 1; o < 64; o <<= 1) {
    const double ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
    if (lane >= o) { ia += ua; ib += ub; }
  }
  __syncthreads();
  if (lane == 63) { I.scan_a[wid] = ia; I.scan_b[wid] = ib; }
  __syncthreads();
  double wa = 0.0, wb = 0.0;
  for (int w = 0; w < QP_NW; w++) { if (w < wid) { wa += I.scan_a[w]; wb += I.scan_b[w]; } ta += I.scan_a[w]; tb += I.scan_b[w]; }
  double ea = __shfl_up(ia, 1), eb = __shfl_up(ib, 1); /* exclusive prefix inside the wavefront */
  if (lane == 0) { ea = 0.0; eb = 0.0; }
  double ra = a0 + (wa + ea), rb = b0 + (wb + eb); /* value before my first element */
  for (int e = e0; e < e1; e++) {
    if (mypos == 0x7fffffff && ra * keys[e] + rb > 0) { mypos = e; mytau = -rb / ra; }
    const int iz = idx[e];
    const double dl = a.ls_delta()[iz], al = a.ls_alpha()[iz];
    if (dl > 0) { ra = ra + dl * dl; rb = rb - dl * al; } else { ra = ra - dl * dl; rb = rb + dl * al; }
  }
  };
  if (in_lds) sort_and_scan((double *)lds, (int *)(lds + (size_t)P2 * 8), std::false_type());
  else sort_and_scan(a.ls_key() + 0, a.ls_idx() + 0, std::true_type()); /* keys are compacted in place */
  const int pos = block_imin(I.S, mypos);
  __syncthreads();
  if (pos == 0x7fffffff) { if (tid == 0) I.S.bc[0] = -(b0 + tb) / (a0 + ta); }
  else if (mypos == pos) I.S.bc[0] = mytau;
  __syncthreads();
  const double tau = I.S.bc[0];
  if (tid == 0) { I.s.eta = eta; I.s.beta = beta; I.nL = nL; }
  __syncthreads();
  return tau;
}

/* compute_dual_objective (iteration.c:272-299): rhs = Aty + q, solve with the resident factor of Q (LD_Q),
 * dual = -1/2 rhs' Q^{-1} rhs - sum_i (y_i > 0 ? y_i bmax_i : y_i bmin_i), unscaled by 1/c, plus the constant.
 * The two sums use the workgroup's fixed reduction tree (the CPU sums sequentially / in groups of four). */
QPP double dev_dual_objective(const qpg_view &V, const QpArrays &a, int b, const double *LQ, const double *DgQ, IterShared &I, char *lds) {
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double *rhs = V.dual_rhs + (size_t)b * n, *sol = a.temp_n();
  __syncthreads();
  for (int j = tid; j < n; j += QP_T) { const double r = a.Aty()[j] + 1.0 * a.q()[j]; rhs[j] = r; sol[j] = r; }
  __syncthreads();
  /* QP_CALL_BLOCK: see qpalm_device.h (ROCm 7.2 places live-range copies around this call ahead of the exec restore
   * of the loop above; the lane index of one shuffle step of the reduction below then held garbage on the hardware) */
  if (QP_CALL_BLOCK()) dense_solve(LQ, DgQ, n, V.ld, sol, lds, V.lds_bytes);
  double vm[1] = {0.0}, vs[2] = {0.0, 0.0};
  for (int j = tid; j < n; j += QP_T) vs[0] += rhs[j] * sol[j];
  for (int i = tid; i < m; i += QP_T) { const double yv = a.y()[i]; vs[1] += yv > 0 ? yv * a.bmax()[i] : yv * a.bmin()[i]; }
  block_reduce<0, 2>(I.S, vm, vs);
  double dobj = 0;
  dobj -= 0.5 * vs[0];
  dobj -= vs[1];
  if (I.s.has_scaling) dobj *= I.s.sc_cinv;
  dobj += V.c0[b];
  return dobj;
}

#include "qpalm_kkt.h"
#include "qpalm_sparse.h"

/* ... and with the sparse factor (round 6): LD_Q = the L D L' of Q alone on the pattern of the main factor (a superset of Q's own: the entries
 * outside it come out as exact zeros), in a second value array per slot; the same sums */
QPP double dev_dual_objective_sp(const qpg_view &V, const QpArrays &a, int b, const SpArrays &SQ, IterShared &I) {
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double *rhs = V.dual_rhs + (size_t)b * n, *sol = a.temp_n();
  __syncthreads();
  for (int j = tid; j < n; j += QP_T) { const double r = a.Aty()[j] + 1.0 * a.q()[j]; rhs[j] = r; sol[j] = r; }
  __syncthreads();
  if (QP_CALL_BLOCK()) sp_solve(n, SQ, sol);
  double vm[1] = {0.0}, vs[2] = {0.0, 0.0};
  for (int j = tid; j < n; j += QP_T) vs[0] += rhs[j] * sol[j];
  for (int i = tid; i < m; i += QP_T) { const double yv = a.y()[i]; vs[1] += yv > 0 ? yv * a.bmax()[i] : yv * a.bmin()[i]; }
  block_reduce<0, 2>(I.S, vm, vs);
  double dobj = 0;
  dobj -= 0.5 * vs[0];
  dobj -= vs[1];
  if (I.s.has_scaling) dobj *= I.s.sc_cinv;
  dobj += V.c0[b];
  return dobj;
}

/* =============================================================================================
 * the loop body of qpalm_solve (src/qpalm.c:484-711) for one QP; runs at most `budget` iterations
 * =========================================================================================== */
template <int RPT>
QPN void dev_solve(const qpg_view &V, int b, int slot, int budget, int fresh, IterShared &I, char *lds) {
  constexpr bool SPARSE = (RPT == QPG_RPT_SPARSE);
  const qpg_settings &st = *V.settings;
  QpArrays a = qp_arrays(V, b);
  const int n = a.n, m = a.m, tid = threadIdx.x;
  double *L = V.L + (size_t)slot * V.ld * V.nfac, *Dg = V.Dg + (size_t)slot * V.nfac, *Wst = V.Wst + (size_t)slot * V.wst_stride;
  double *LQ = V.LQ ? V.LQ + (SPARSE ? (size_t)slot * V.sp_nnzL : (size_t)slot * V.ld * V.nfac) : nullptr, *DgQ = V.DgQ ? V.DgQ + (size_t)slot * V.nfac : nullptr; /* (sparse: LQ = the values of LD_Q on the factor's pattern) */
  __syncthreads();
  if (tid == 0) {
    I.s = V.sc[b];
    /* qpalm_solve on a finished workspace starts over (src/qpalm.c:401-420 re-initialises its locals) */
    if (fresh && I.s.done) { I.s.done = 0; I.s.in_solve = 0; }
  }
  __syncthreads();
  if (I.s.done) return;
  const long long t_launch = QP_CLOCK();
  if (!I.s.in_solve) { /* qpalm.c:409-482 */
    if (tid == 0) {
      I.s.eps_abs_in = st.eps_abs_in; I.s.eps_rel_in = st.eps_rel_in;
      I.s.reset_newton = 1; I.s.gamma = qp_gamma_init(st, I.s); I.s.gamma_maxed = (0 || I.s.nc_flag);
    }
    for (int i = tid; i < m; i += QP_T) a.active_old()[i] = 0;
    __syncthreads();
    if (!I.s.initialized) dev_warm_start(V, a, b, 0, 0, I);
    if (tid == 0) {
      I.s.dual_objective = 0; /* QPALM_NULL (B8) unless dual termination is enabled */
      I.s.dual_pending = st.enable_dual_termination ? 1 : 0;
      I.s.iter = 0; I.s.iter_out = 0; I.s.prev_iter = 0; I.s.no_change = 0;
      I.s.eps_k_abs = st.eps_abs_in; I.s.eps_k_rel = st.eps_rel_in;
      I.s.in_solve = 1; I.s.solve_time = 0.0; I.s.slot = slot; I.s.pend_stage = 0; I.s.pend_clock = 0;
      I.s.n_refactor = 0; I.s.n_factor_Q = 0; I.s.n_sweeps = 0; I.s.n_rank1 = 0; I.s.n_solve = 0;
      I.s.n_sigma_updates = 0; I.s.n_boost_gamma = 0; I.s.n_fused_solve = 0; I.s.guard_redo = 0; I.s.n_guard_refactor = 0; I.s.guard_spent = 0;
      I.s.ticks_total = 0; I.s.ticks_factor = 0; I.s.ticks_update = 0; I.s.ticks_solve = 0; I.s.ticks_linesearch = 0; I.s.ticks_resid = 0;
      for (int k = 0; k < QPG_NDBG; k++) I.s.ticks_dbg[k] = 0;
      I.s.ticks_dbg[QPG_CNT_PLACEMENT] = I.S.placement;
    }
    __syncthreads();
  }
  const int scal = I.s.has_scaling, prox = qp_prox(st, I.s);
  int executed = 0;
  while (true) {
    /* At most ONE linear-algebra operation per pass, so form_schur / factor / update each have a
     * single call site (one copy of those loop nests in the kernel):
     *   1 refactor Q + A' Sigma_act A   2 update entering / downdate leaving   3 factor Q (+ I/gamma)
     *   4 update for changed sigma       5 Gershgorin bound for boost_gamma      6 boost_gamma without bound
     *   7 factor of Q alone into the second slot (LD_Q of qpalm.c:466-467, dual termination) */
    int la = 0, n_sig = 0, action = 0, nchange = 0, kind = QP_KIND_NEWTON;
    double gam = I.s.gamma;
    long long tr0 = 0;
    __syncthreads();
    const bool dual_init = (QP_UNIFORM(I.s.dual_pending) != 0); /* workgroup-uniform: keep the branch scalar */
    const bool resume = (QP_UNIFORM(I.s.pend_stage) != 0); /* coop mode: the host has done this iteration's factorisation / solve */
    const bool redo = !resume && (QP_UNIFORM(I.s.guard_redo) != 0); /* the Newton step of the last pass is taken again with a fresh factorisation (the guard below) */
    if (redo) {
      la = V.kkt ? 8 : 1; action = 1; kind = QP_KIND_NEWTON; nchange = 0; /* the active sets and nb_enter / nb_leave of the step stay (they are read again, B4) */
      __syncthreads();
      if (tid == 0) { I.s.guard_redo = 0; I.s.guard_spent = 2; } /* 2: this pass is the redone one (read again after its line search) */
      __syncthreads();
    } else
    if (resume) {
      la = I.s.pend_la; action = I.s.pend_action; kind = I.s.pend_kind; nchange = I.s.pend_nchange; gam = I.s.pend_gam;
      if (la == 4) { n_sig = nchange; nchange = 0; }
      __syncthreads();
      /* coop mode, one-launch update sweep (co_updown_persist): a workgroup that timed out waiting for a table left its mark and the factor half
       * updated.  Nothing iterates on that: the mark is cleared and the factor rebuilt -- a Newton step is taken again with a fresh factorisation
       * (the guard's path below), a sigma update leaves reset_newton set. */
      const bool sweep_died = (la == 2 || la == 4) && V.co_flags != nullptr && QP_UNIFORM(V.co_flags[(size_t)b * 4 + 1]) != 0;
      __syncthreads();
      if (sweep_died && tid == 0) { V.co_flags[(size_t)b * 4 + 1] = 0; I.s.n_guard_refactor++; if (la == 4) I.s.reset_newton = 1; }
      if (sweep_died && la == 2 && kind == QP_KIND_NEWTON) {
        if (tid == 0) { I.s.pend_stage = 0; I.s.guard_redo = 1; }
        __syncthreads();
        continue;
      }
      if (tid == 0) {
        I.s.pend_stage = 0;
        /* the host's multi-workgroup kernels ran between the suspension and this launch: their time belongs to the solve */
        if (I.s.pend_clock != 0 && t_launch > I.s.pend_clock) I.s.solve_time += (double)(t_launch - I.s.pend_clock) * 1e-8;
        I.s.pend_clock = 0;
      }
      __syncthreads();
    } else
    if (dual_init) la = 7;
    else {
    if (I.s.iter >= st.max_iter) { /* qpalm.c:712-735 */
      dev_store_solution(V, a, b, I);
      if (tid == 0) { I.s.status = QPG_MAX_ITER_REACHED; I.s.done = 1; I.s.initialized = 0; I.s.in_solve = 0; }
      break;
    }
    if (executed >= budget) break;
    executed++;
    QP_OPAQUE(a.b);
    tr0 = QP_CLOCK();
    /* ---- dx-dependent quantities of is_dual_infeasible (termination.c:190-203) -------------- */
    double vm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vs[4] = {0, 0, 0, 0};
    for (int j = tid; j < n; j += QP_T) {
      const double dx = a.x()[j] + (-1) * a.x_prev()[j];
      a.delta_x()[j] = dx;
      const double t = scal ? a.D()[j] * dx : dx;
      vm[0] = qmax(vm[0], qabs(t));
      vs[0] += t * t;
    }
    block_reduce<1, 1>(I.S, vm, vs);
    const double eps_dinf_norm_Ddx = st.eps_dual_inf * vm[0], dxdx = vs[0];
    /* ---- compute_residuals, m part (iteration.c:26-35) + norms -------------------------------- */
    for (int k = 0; k < 8; k++) vm[k] = 0.0;
    for (int k = 0; k < 4; k++) vs[k] = 0.0;
    double viol = 0.0;
    /* yh also goes into the dynamic LDS block (free in this phase) when it fits: A' yh then gathers it from there -- the third of a
     * column's three dependent round trips (pointer -> index / value -> yh[index]) becomes an LDS read */
    const bool ys_lds = QP_UNIFORM((int)(QP_SPMV_LDS && (size_t)m * sizeof(double) <= (size_t)V.lds_bytes)) != 0;
    double QP_LDS_AS *ys = (double QP_LDS_AS *)lds;
    for (int i = tid; i < m; i += QP_T) {
      const double yv = a.y()[i], ax = a.Axv()[i];
      double t = yv * a.sigma_inv()[i];
      const double axys = ax + 1 * t;
      const double zz = qmax(a.bmin()[i], qmin(axys, a.bmax()[i]));
      const double pr = ax + (-1) * zz;
      t = pr * a.sigma()[i];
      const double yhv = yv + 1 * t;
      a.Axys()[i] = axys; a.z()[i] = zz; a.pri_res()[i] = pr; a.yh()[i] = yhv;
      if (ys_lds) ys[i] = yhv;
      const double dy = yhv + (-1) * yv;
      a.delta_y()[i] = dy;
      if (scal) {
        const double Ei = a.E()[i], Einv = a.Einv()[i];
        vm[0] = qmax(vm[0], qabs(Einv * pr));
        vm[1] = qmax(vm[1], qabs(Einv * ax));          /* B1: only the Einv.*Ax half */
        vm[2] = qmax(vm[2], qabs(Ei * dy));
        vs[0] += (a.bmax()[i] < Ei * QPG_INFTY) ? a.bmax()[i] * qmax(dy, 0) : 0;
        vs[0] += (a.bmin()[i] > -Ei * QPG_INFTY) ? a.bmin()[i] * qmin(dy, 0) : 0;
        const double adx = Einv * a.Ad()[i];
        if ((a.bmax()[i] < Ei * QPG_INFTY && adx >= eps_dinf_norm_Ddx) || (a.bmin()[i] > -Ei * QPG_INFTY && adx <= -eps_dinf_norm_Ddx)) viol = 1.0;
      } else {
        vm[0] = qmax(vm[0], qabs(pr));
        vm[1] = qmax(vm[1], qmax(qabs(ax), qabs(zz)));
        vm[2] = qmax(vm[2], qabs(dy));
        vs[0] += (a.bmax()[i] < QPG_INFTY) ? a.bmax()[i] * qmax(dy, 0) : 0;
        vs[0] += (a.bmin()[i] > -QPG_INFTY) ? a.bmin()[i] * qmin(dy, 0) : 0;
        const double adx = a.Ad()[i];
        if ((a.bmax()[i] < QPG_INFTY && adx >= eps_dinf_norm_Ddx) || (a.bmin()[i] > -QPG_INFTY && adx <= -eps_dinf_norm_Ddx)) viol = 1.0;
      }
    }
    vm[3] = viol;
    __syncthreads();
    QP_OPAQUE(a.b);
    /* ---- Atyh = A' yh (iteration.c:45) ------------------------------------------------------- */
    if (ys_lds) spmv_rows<16>(n, a.Ap(), a.Ai(), a.Ax(), (qp_lds_cdouble)ys, [&](int r, double s) { a.Atyh()[r] = s; });
    else spmv_rows<16>(n, a.Ap(), a.Ai(), a.Ax(), (const double *)a.yh(), [&](int r, double s) { a.Atyh()[r] = s; });
    __syncthreads();
    /* ---- df, dphi (iteration.c:37-47) + dual residual norms (termination.c:61-129) ---------- */
    gam = I.s.gamma;
    const double mginv = -1 / gam, tg = -I.s.tau / gam;
    for (int j = tid; j < n; j += QP_T) {
      const double atyh = a.Atyh()[j];
      double dfv = a.Qxv()[j] + 1 * a.q()[j];
      if (prox) dfv = dfv + mginv * a.x0()[j];
      const double dphi = dfv + 1 * atyh;
      a.df()[j] = dfv; a.dphi()[j] = dphi;
      const double Dinv = scal ? a.Dinv()[j] : 1.0;
      double r1, r2;
      if (prox) {
        const double xx0 = a.x()[j] + (-1) * a.x0()[j];
        const double tn = dphi + mginv * xx0;
        r1 = scal ? Dinv * tn : tn;
        r2 = scal ? Dinv * dphi : dphi;
      } else { r1 = scal ? Dinv * dphi : dphi; r2 = r1; }
      vm[4] = qmax(vm[4], qabs(r1));
      vm[5] = qmax(vm[5], qabs(r2));
      const double nq = scal ? Dinv * a.Qxv()[j] : a.Qxv()[j], nqq = scal ? Dinv * a.q()[j] : a.q()[j], na = scal ? Dinv * atyh : atyh;
      vm[6] = qmax(vm[6], qmax(qabs(nq), qmax(qabs(nqq), qabs(na))));
      double atdy = atyh + (-1) * a.Aty()[j];
      if (scal) atdy = Dinv * atdy;
      vm[7] = qmax(vm[7], qabs(atdy));
      const double dx = a.delta_x()[j];
      if (prox) { const double tq = a.Qd()[j] + tg * a.d()[j]; vs[1] += dx * tq; } else vs[1] += a.Qd()[j] * dx;
      vs[2] += a.q()[j] * dx;
    }
    block_reduce<8, 3>(I.S, vm, vs);
    if (tid == 0) I.s.ticks_dbg[12] += QP_CLOCK() - tr0;
    /* ---- decision: check_termination + the branch of qpalm.c:488-676 ------------------------- */
    if (tid == 0) {
      qpg_scalars &s = I.s;
      s.pri_res_norm = vm[0];
      s.dua_res_norm = vm[4]; s.dua2_res_norm = vm[5];
      if (scal) { s.dua_res_norm *= s.sc_cinv; s.dua2_res_norm *= s.sc_cinv; }
      s.eps_pri = st.eps_abs + st.eps_rel * vm[1];
      s.norm_Ax_z = vm[1];
      double mx = vm[6];
      if (scal) mx *= s.sc_cinv;
      s.eps_dua = st.eps_abs + st.eps_rel * mx;
      s.eps_dua_in = s.eps_abs_in + s.eps_rel_in * mx;
      int kind = QP_KIND_NEWTON;
      const double eps_pinf = st.eps_prim_inf * vm[2];
      int prim_inf = 0, dual_inf = 0;
      if (eps_pinf != 0) prim_inf = (vm[7] <= eps_pinf) && (vs[0] <= -eps_pinf);
      if (eps_dinf_norm_Ddx != 0 && vm[3] == 0.0) {
        const double dxQdx = vs[1], qdx = vs[2], e2 = st.eps_dual_inf * st.eps_dual_inf;
        if (scal) dual_inf = (dxQdx <= -s.sc_c * e2 * dxdx) || ((dxQdx <= s.sc_c * e2 * dxdx) && (qdx <= -s.sc_c * eps_dinf_norm_Ddx));
        else dual_inf = (dxQdx <= -e2 * dxdx) || ((dxQdx <= e2 * dxdx) && (qdx <= -eps_dinf_norm_Ddx));
      }
      if ((s.pri_res_norm < s.eps_pri) && (s.dua_res_norm < s.eps_dua)) { s.status = QPG_SOLVED; kind = QP_KIND_TERMINATED; }
      else if (prim_inf) { s.status = QPG_PRIMAL_INFEASIBLE; kind = QP_KIND_TERMINATED; }
      else if (dual_inf) { s.status = QPG_DUAL_INFEASIBLE; kind = QP_KIND_TERMINATED; }
      else if ((s.dua2_res_norm <= s.eps_dua_in) || (s.no_change == 3)) kind = QP_KIND_OUTER;
      else if (s.iter == s.prev_iter + (int)st.inner_max_iter) kind = QP_KIND_FORCED;
      else kind = QP_KIND_NEWTON;
      I.kind = kind;
    }
    __syncthreads();
    QP_OPAQUE(a.b);
    kind = I.kind;
    if (kind == QP_KIND_TERMINATED) {
      const int status = I.s.status;
      if (status == QPG_SOLVED) dev_store_solution(V, a, b, I);
      else if (status == QPG_PRIMAL_INFEASIBLE) {
        if (scal) for (int i = tid; i < m; i += QP_T) { double v = a.delta_y()[i] * I.s.sc_cinv; a.delta_y()[i] = a.E()[i] * v; }
      } else {
        if (scal) for (int j = tid; j < n; j += QP_T) a.delta_x()[j] = a.D()[j] * a.delta_x()[j];
      }
      __syncthreads();
      if (tid == 0) { I.s.done = 1; I.s.initialized = 0; I.s.in_solve = 0; I.s.last_kind = QP_KIND_TERMINATED; }
      break;
    }
    if (kind == QP_KIND_OUTER || kind == QP_KIND_FORCED) { /* qpalm.c:585-660 */
      if (tid == 0) I.s.no_change = 0;
      __syncthreads();
      if (I.s.iter_out > 0 && I.s.pri_res_norm > I.s.eps_pri) { n_sig = dev_update_sigma_pre(V, a, I); if (n_sig > 0) la = 4; }
      if (kind == QP_KIND_OUTER) {
        for (int i = tid; i < m; i += QP_T) a.y()[i] = a.yh()[i];
        for (int j = tid; j < n; j += QP_T) a.Aty()[j] = a.Atyh()[j];
        __syncthreads();
        if (st.enable_dual_termination) { /* qpalm.c:543-583 */
          double dobj;
          if constexpr (SPARSE) { SpArrays SQ = sp_arrays(V, b, slot, DgQ, lds); SQ.Lx = LQ; dobj = dev_dual_objective_sp(V, a, b, SQ, I); }
          else dobj = dev_dual_objective(V, a, b, LQ, DgQ, I, lds);
          if (tid == 0) I.s.dual_objective = dobj;
          __syncthreads();
          if (QP_UNIFORM((int)(dobj > st.dual_objective_limit))) { /* same value in every lane: scalar branch */
            dev_store_solution(V, a, b, I);
            if (tid == 0) { I.s.status = QPG_DUAL_TERMINATED; I.s.done = 1; I.s.initialized = 0; I.s.in_solve = 0; I.s.last_kind = QP_KIND_TERMINATED; }
            break;
          }
        }
        if (tid == 0) {
          I.s.eps_abs_in = qmax(st.eps_abs, st.rho * I.s.eps_abs_in);
          I.s.eps_rel_in = qmax(st.eps_rel, st.rho * I.s.eps_rel_in);
        }
        __syncthreads();
        if (I.s.nc_flag) { /* qpalm.c:586-611: the proximal point and the tolerances move only when the subproblem is feasible enough */
          const double eps_k = I.s.eps_k_abs + I.s.eps_k_rel * I.s.norm_Ax_z; /* B1: the Einv.*Ax half only when scaled */
          if (I.s.pri_res_norm < eps_k) {
            for (int j = tid; j < n; j += QP_T) a.x0()[j] = a.x()[j];
            __syncthreads();
            if (tid == 0) {
              I.s.eps_k_abs = qmax(st.eps_abs, st.rho * I.s.eps_k_abs);
              I.s.eps_k_rel = qmax(st.eps_rel, st.rho * I.s.eps_k_rel);
            }
          }
          __syncthreads();
        } else
        if (prox) { /* qpalm.c:612-630 (convex) */
          if (!I.s.gamma_maxed && I.s.iter_out > 0 && I.s.nb_enter == 0 && I.s.nb_leave == 0 && I.s.pri_res_norm < I.s.eps_pri) {
            for (int i = tid; i < m; i += QP_T) { const double t = a.y()[i] / a.sigma()[i]; a.Axys()[i] = a.Axv()[i] + 1 * t; } /* B3 */
            __syncthreads();
            dev_active_sets(a, I);
            if (I.s.nb_enter == 0 && I.s.nb_leave == 0) la = (I.s.nb_active && !V.kkt) ? 5 : 6; /* boost_gamma */
            else dev_update_gamma(V, a, I);
          } else dev_update_gamma(V, a, I);
          for (int j = tid; j < n; j += QP_T) a.x0()[j] = a.x()[j];
        }
      } else if (prox) { /* qpalm.c:647-660 */
        dev_update_gamma(V, a, I);
        if (!I.s.nc_flag) for (int j = tid; j < n; j += QP_T) a.x0()[j] = a.x()[j];
      }
      for (int i = tid; i < m; i += QP_T) a.pri_res_in()[i] = a.pri_res()[i];
      __syncthreads();
      if (tid == 0) { I.s.iter_out++; I.s.prev_iter = I.s.iter; I.s.last_kind = kind; I.s.last_fact = 0; }
    } else { /* Newton step, qpalm.c:662-668 -> update_primal_iterate (iteration.c:213-229) */
      if (tid == 0) {
        if (I.s.nb_enter + I.s.nb_leave) I.s.no_change = 0; else I.s.no_change++;
        if (((I.s.iter % (int)st.reset_newton_iter) + (int)st.reset_newton_iter) % (int)st.reset_newton_iter == 0) I.s.reset_newton = 1;
      }
      __syncthreads();
      dev_active_sets(a, I);
      /* newton_set_direction, SCHUR branch (newton.c:96-113) */
      nchange = I.s.nb_enter + I.s.nb_leave;
      const double thr = qmin(st.max_rank_update_fraction * (double)(n + m), (double)st.max_rank_update);
      if (V.kkt) { /* newton.c:32-53 */
        action = (I.s.kkt_first || I.s.reset_newton || ((double)nchange > thr)) ? 1 : (nchange ? 2 : 0);
        la = 8;
      } else
      if ((I.s.reset_newton && I.s.nb_active) || ((double)nchange > thr) ||
          (V.update_rank_threshold >= 0 && I.s.nb_active && nchange > V.update_rank_threshold)) action = 1;
      else if (I.s.nb_active) action = nchange ? 2 : 0;
      else action = 3;
      if (!V.kkt) la = action;
    }
    } /* !dual_init */
    QP_OPAQUE(a.b);
    if (V.offload && !resume && !V.kkt && (la == 1 || la == 3 || la == 7 || (kind == QP_KIND_NEWTON && la == 0) || (V.offload >= 2 && (la == 2 || la == 4)))) {
      /* coop mode: hand the factorisation or the rank update (and the Newton solve that follows it) to the host's multi-workgroup
       * kernels.  With offload == 1 the rank updates (la == 2, 4) stay on this workgroup and only the solve after them goes to the host */
      if (kind == QP_KIND_NEWTON && la != 7) {
        for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1; /* ldlsolveLD_neg_dphi's right-hand side */
      }
      __syncthreads();
      if (tid == 0) {
        I.s.pend_stage = 1; I.s.pend_la = (la == 1 || la == 3 || la == 7 || la == 2 || la == 4) ? la : 0; I.s.pend_action = action; I.s.pend_kind = kind;
        I.s.pend_nchange = (la == 4) ? n_sig : nchange; I.s.pend_gam = gam;
      }
      __syncthreads();
      break;
    }
    const long long t0 = QP_CLOCK();
    double gersh_ub = 0.0;
    if constexpr (SPARSE) {
      /* sparse factor (qpalm_sparse.h): rows entering / leaving the active set are rank-1 updates along their elimination-tree
       * paths where that pays, a refactorisation otherwise; changed penalties likewise (la == 4: ldlupdate_sigma_changed as path updates with the scaled
       * rows, dev_update_sigma_pre decides) */
      const SpArrays SP = sp_arrays(V, b, slot, Dg, lds);
      if (la == 2 && !sp_update_pays(nchange, SP.nlev, n)) { la = 1; action = 1; } /* a chain-like tree: refactorising is cheaper than walking it per row */
      if (la == 2) sp_updown(V, b, n, SP, a.enter(), I.s.nb_enter, a.leave(), I.s.nb_leave);
      else if (la == 4) sp_updown(V, b, n, SP, a.enter(), n_sig, a.leave(), 0); /* ldlupdate_sigma_changed: the rows listed in enter[], scaled by dev_ldlupdate_sigma_scale */
      else if (la == 1 || la == 3) sp_factor(V, b, n, SP, la == 1, prox != 0, gam);
      else if (la == 5) gersh_ub = sp_gershgorin(V, b, n, SP, I.S);
      else if (la == 7) { /* qpalm.c:459-468: LD_Q = the factor of Q alone (no A' Sigma A, no I / gamma) and the dual objective of the starting point */
        SpArrays SQ = SP; SQ.Lx = LQ; SQ.Dg = DgQ;
        sp_factor(V, b, n, SQ, false, false, gam);
        const double dobj = dev_dual_objective_sp(V, a, b, SQ, I);
        if (tid == 0) { I.s.dual_objective = dobj; I.s.dual_pending = 0; }
        __syncthreads();
        continue;
      }
    } else
    if (resume) {
      if (la == 7) { /* LD_Q is ready (qpalm.c:459-468) */
        const double dobj = dev_dual_objective(V, a, b, LQ, DgQ, I, lds);
        if (tid == 0) { I.s.dual_objective = dobj; I.s.dual_pending = 0; }
        __syncthreads();
        continue;
      }
    } else
    if (la == 1 || la == 3 || la == 5 || la == 7) {
      double *Lt = (la == 7) ? LQ : L, *Dt = (la == 7) ? DgQ : Dg;
      gersh_ub = form_schur(V, b, n, Lt, la == 5, la == 1 || la == 5, (la == 1 || la == 3) && (prox != 0), gam, I.S, lds);
      if (tid == 0) I.s.ticks_dbg[3] += QP_CLOCK() - t0;
      if (la != 5) dev_factor<RPT>(V, n, Lt, Dt, lds, I.s.ticks_dbg);
      if (dual_init) { /* qpalm.c:459-468: LD_Q is ready, the dual objective of the starting point (scalar branch) */
        const double dobj = dev_dual_objective(V, a, b, LQ, DgQ, I, lds);
        if (tid == 0) { I.s.dual_objective = dobj; I.s.dual_pending = 0; }
        __syncthreads();
        continue;
      }
    } else if (la == 8) {
      kkt_newton<RPT>(&V, b, L, Dg, Wst, &I, lds, action, I.s.nb_enter, I.s.nb_leave, QP_KKT_SOLVE | QP_KKT_REFINE);
    } else if (la == 2 || la == 4) {
      const int n_up = (la == 2) ? I.s.nb_enter : n_sig, n_dn = (la == 2) ? I.s.nb_leave : 0;
      /* a Newton step solves right after the update: its forward substitution rides on the last sweep */
      double *fs = nullptr;
      constexpr bool FUSED = (RPT > 0) && !QP_NOFUSE; /* the large-factor sweep keeps the solve separate */
      if (FUSED && la == 2 && !V.offload) {
        for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1; /* ldlsolveLD_neg_dphi's right-hand side */
        fs = a.d();
        __syncthreads();
      }
      dev_updown<RPT>(V, b, n, L, Dg, Wst, a.enter(), n_up, a.leave(), n_dn, I.S, lds, I.s.ticks_dbg, fs);
      if (V.offload && !resume && !V.kkt && la == 2 && kind == QP_KIND_NEWTON) { /* coop mode: the solve with the updated factor goes to the host */
        for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
        __syncthreads();
        if (tid == 0) { I.s.pend_stage = 1; I.s.pend_la = 0; I.s.pend_action = action; I.s.pend_kind = kind; I.s.pend_nchange = nchange; I.s.pend_gam = gam; }
        __syncthreads();
        break;
      }
    }
    const long long t1 = QP_CLOCK();
    QP_OPAQUE(a.b);
    if (la == 4) {
      dev_update_sigma_post(a, I, n_sig);
      if (tid == 0) { I.s.n_sweeps = (int)I.s.ticks_dbg[QPG_CNT_SWEEPS]; I.s.ticks_update += t1 - t0; }
    } else if (la == 5 || la == 6) dev_boost_gamma_apply(V, a, I, gersh_ub);
    if (kind == QP_KIND_NEWTON) {
      /* ldlsolveLD_neg_dphi (solver_interface.c:505-519) */
      if constexpr (SPARSE) {
        for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
        sp_solve(n, sp_arrays(V, b, slot, Dg, lds), a.d());
      } else
      if (!V.kkt && !resume) {
        const bool fused = (RPT > 0) && !QP_NOFUSE && (action == 2) && !V.offload;
        if (!fused) for (int j = tid; j < n; j += QP_T) a.d()[j] = a.dphi()[j] * -1;
        __syncthreads();
        dense_solve(L, Dg, n, V.ld, a.d(), lds, V.lds_bytes, I.s.ticks_dbg, fused ? 2 : 0); /* 2: d already holds L^{-1} (-dphi) */
      }
      const long long t2 = QP_CLOCK();
      for (int i = tid; i < m; i += QP_T) a.active_old()[i] = a.active()[i];
      if (tid == 0) {
        I.s.reset_newton = 0;
        I.s.kkt_first = 0;
        if (action == 1) { I.s.n_refactor++; I.s.ticks_factor += t1 - t0; }
        if (action == 3) { I.s.n_factor_Q++; I.s.ticks_factor += t1 - t0; }
        if (action == 2) { if (!V.kkt) I.s.n_rank1 += nchange; I.s.n_sweeps = (int)I.s.ticks_dbg[QPG_CNT_SWEEPS]; I.s.ticks_update += t1 - t0; }
        I.s.n_solve++; I.s.ticks_solve += t2 - t1;
        if (!V.kkt && (RPT > 0) && !QP_NOFUSE && action == 2 && !V.offload) I.s.n_fused_solve++;
        I.s.last_fact = action;
      }
      QP_OPAQUE(a.b);
      const double tau = dev_linesearch(V, a, I, lds);
      const long long t3 = QP_CLOCK();
      QP_OPAQUE(a.b);
      /* NOT in the reference (stated deviation, restated by the oracle: oq_update_primal_iterate): a direction out of an UPDATED factor that is
       * not finite (eta = d'(Q + I/gamma)d or beta = d'df is not: a pivot went through zero inside an update sweep) is not stepped along: the
       * pass is taken again with a fresh factorisation.  The reference iterates on NaN to max_iter there (solver_interface.c:357-368 looks at
       * c->status only when !DLONG).  Fuzz case 701 / 114 (LP, sigma up to 1e9 against 1 / gamma = 1e-7).  Free on healthy steps: two compares
       * on scalars the line search has.  (A second trigger -- an update that leaves a pivot <= 0 in a convex QP -- was built in round 6 and taken
       * out: on LPs whose H is singular by construction it fires at every step, 80 000 times in campaign 701, and moves healthy trajectories.) */
      {
        const int gs = QP_UNIFORM(I.s.guard_spent); /* 0: armed, 1: spent, 2: this pass is the redone step */
        const double ge = I.s.eta, gb = I.s.beta;
        const bool bad = !(qabs(ge) <= 1.7976931348623157e308) || !(qabs(gb) <= 1.7976931348623157e308);
        const int what = ((action == 0 || action == 2 || gs == 2) && gs != 1 && bad) ? ((gs == 2) ? 2 : 1) : ((gs == 2) ? 3 : 0);
        const int w = QP_UNIFORM(what);
        if (w != 0) {
          __syncthreads();
          /* 1: redo with a fresh factorisation.  2: the fresh factorisation gives a non-finite direction too -- nothing to repair (a NaN right-hand side, an H that is
           * singular by construction): the step is taken as it is, like the reference's, and the guard stays off for the rest of this solve.  3: the redone step is fine: re-arm */
          if (tid == 0) {
            if (w == 1) { I.s.guard_redo = 1; I.s.n_guard_refactor++; I.s.ticks_linesearch += t3 - t2; }
            else I.s.guard_spent = (w == 2) ? 1 : 0;
          }
          __syncthreads();
          if (w == 1) continue;
        }
      }
      /* iteration.c:219-228 */
      for (int j = tid; j < n; j += QP_T) {
        const double xv = a.x()[j];
        a.x_prev()[j] = xv;
        a.dphi_prev()[j] = a.dphi()[j];
        a.x()[j] = xv + tau * a.d()[j];
        const double qd = a.Qd()[j] * tau;
        a.Qd()[j] = qd;
        a.Qxv()[j] = a.Qxv()[j] + 1 * qd;
      }
      for (int i = tid; i < m; i += QP_T) {
        const double ad = a.Ad()[i] * tau;
        a.Ad()[i] = ad;
        a.Axv()[i] = a.Axv()[i] + 1 * ad;
      }
      if (tid == 0) { I.s.tau = tau; I.s.last_kind = QP_KIND_NEWTON; I.s.ticks_linesearch += t3 - t2; }
    }
    __syncthreads();
    if (tid == 0) {
      I.s.iter++;
      /* time limit (qpalm.c:680-710): wall_clock64 ticks at 100 MHz */
      const double elapsed = (double)(QP_CLOCK() - t_launch) * 1e-8;
      if (I.s.setup_time + I.s.solve_time + elapsed > st.time_limit) I.kind = -1; else I.kind = 0;
    }
    __syncthreads();
    if (I.kind == -1) {
      if (tid == 0) { I.s.iter--; }
      __syncthreads();
      dev_store_solution(V, a, b, I);
      if (tid == 0) { I.s.status = QPG_TIME_LIMIT_REACHED; I.s.done = 1; I.s.initialized = 0; I.s.in_solve = 0; }
      break;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const long long t_exit = QP_CLOCK();
    I.s.solve_time += (double)(t_exit - t_launch) * 1e-8;
    I.s.ticks_total += t_exit - t_launch;
    I.s.pend_clock = I.s.pend_stage ? t_exit : 0; /* suspended for the host (coop mode): the clock keeps running */
    V.sc[b] = I.s;
  }
  __syncthreads();
}

#endif
