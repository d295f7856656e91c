/*
 * qpalm_oracle.c -- CPU restatement of the QPALM CHOLMOD/Schur path.  TEST INFRASTRUCTURE ONLY.
 * See qpalm_oracle.h for the rules of use.  Every function cites the reference file:line it
 * restates (paths relative to Benny44/QPALM).  Third-party arithmetic that is absent from the
 * reference tree (SuiteSparse/CHOLMOD, submodule `suitesparse`, branch master, SHA unpinned,
 * .gitmodules:1-4) is restated from its published algorithms:
 *   - sdmult / transpose / scale / aat / add / submatrix : CHOLMOD User Guide semantics at the
 *     argument values the reference passes;
 *   - factorize : simplicial, natural order, no pivoting LDL^T (T. Davis, "Algorithm 849: a
 *     concise sparse Cholesky factorization package", up-looking row order), applied to the
 *     dense lower triangle -- skipping structural zeros does not change any partial sum, so the
 *     dense recurrence below is the same arithmetic;
 *   - updown : Davis & Hager, "Multiple-rank modifications of a sparse Cholesky factorization"
 *     (SIAM J. Matrix Anal. Appl. 2001), method C1, one rank at a time per column which is
 *     bit-identical to CHOLMOD's interleaved multi-rank sweep;
 *   - solve(LDLt) : column-oriented forward solve, diagonal solve, column-oriented backward.
 */
#define _POSIX_C_SOURCE 200809L
#include "qpalm_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define OQ_MAX(a, b) (((a) > (b)) ? (a) : (b)) /* include/global_opts.h:180-190 */
#define OQ_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define OQ_ABS(x) (((x) < 0) ? -(x) : (x))
#define OQ_MOD(a, b) ((((a) % (b)) + (b)) % (b))
static size_t nz1(size_t k) { return k ? k : 1; }

typedef struct { oq_float x; size_t i; } oq_array_element; /* include/types.h:37-40 */

typedef struct {
  oq_float *L;   /* n x n column-major, strict lower part used, unit diagonal implicit */
  oq_float *D;   /* n pivots */
  int valid;
} oq_factor;

struct oq_workspace {
  oq_int n, m;
  /* data (deep copies, scaled in place: src/qpalm.c:128-144) */
  oq_sparse Q, A;
  oq_float *q, *bmin, *bmax, c;
  /* iterates (include/types.h:205-212) */
  oq_float *x, *y, *Ax, *Qx, *Aty, *x_prev, *x0;
  int initialized;
  /* workspace vectors (types.h:219-241) */
  oq_float *temp_m, *temp_n, *sigma, *sigma_inv, *sqrt_sigma;
  oq_float sqrt_sigma_max, gamma;
  oq_int nb_sigma_changed;
  int gamma_maxed;
  oq_float *Axys, *z, *pri_res, *pri_res_in, *yh, *Atyh, *df, *xx0, *dphi, *neg_dphi, *dphi_prev, *d;
  /* linesearch (types.h:249-265) */
  oq_float tau, sqrt_delta, eta, beta;
  oq_float *Qd, *Ad, *delta, *alpha, *temp_2m, *delta2, *delta_alpha;
  oq_array_element *s;
  oq_int *index_L, *index_P, *index_J;
  /* termination (types.h:273-277) */
  oq_float eps_pri, eps_dua, eps_dua_in, eps_abs_in, eps_rel_in;
  oq_float *delta_y, *Atdelta_y, *delta_x, *Qdelta_x, *Adelta_x;
  oq_float *D_temp, *E_temp;
  /* scaling (types.h:63-70) */
  int has_scaling;
  oq_float *D, *Dinv, *E, *Einv, sc_c, sc_cinv;
  /* solver (types.h:155-187) */
  oq_sparse At_sqrt_sigma;     /* n x m */
  oq_float *At_scale;
  oq_int *active, *active_old, *enter, *leave;
  oq_int nb_active, nb_enter, nb_leave;
  int reset_newton;
  oq_factor LD, LD_Q;
  /* KKT path (LADEL build, FACTORIZE_KKT; src/solver_interface.c:119-247, src/newton.c:22-95) */
  int kkt_mode;            /* solver->factorization_method == FACTORIZE_KKT */
  int first_factorization; /* types.h:176 */
  oq_factor LDK;           /* dense LDL' of the (n+m) KKT matrix, natural order */
  oq_int *kkt_state;       /* per constraint: 0 unit diagonal (inactive at the last (re)form), 1 active row present,
                              2 deleted by row_del (diagonal -1/sigma kept for the refinement mat-vec, :233-235) */
  oq_float *rhs_kkt, *sol_kkt, *kkt_tmp;
  oq_sparse At;            /* transpose of the (scaled) A: row k of A = column k (solver->At, qpalm.c:250) */
  oq_int n_row_add, n_row_del, n_refine, n_lobpcg_iter;
  oq_float lobpcg_lambda;
  oq_float *Hbuf; /* scratch n*n for forming Q + A'SA */
  oq_float *wbuf; /* scratch n for rank-1 vectors */
  int updown_block; /* > 1: updown_columns applies up to OQ_UPDOWN_BLOCK ranks per pass over L (bit-identical; cpu_baseline only) */
  oq_float *wblock; /* its OQ_UPDOWN_BLOCK x n vectors */
  /* sparse-storage mode of the Schur factor (round 5): the SAME factorisation and solves on a compressed-column L -- what CHOLMOD's
   * simplicial analyze / factorize / solve do -- for problems whose n^2 panel is out of reach (n = 20 000 .. 100 000).  Every entry
   * receives the operations of the dense routines above in the same order (structural zeros are skipped, and a skipped operation adds
   * an exact zero), so the dense mode pins it: tests/test_sparse_factor.py compares the two bit for bit.  In this mode every change
   * of the active set or of sigma refactorises in mode 2 (what pins the mode); in mode 1 the rows that enter or leave and the rows whose sigma
   * changed are applied as rank-1 updates along their elimination-tree paths where that is cheaper than rebuilding (the engine's rule). */
  int sparse_mode;
  oq_int sp_nlev;                                   /* height of the elimination tree (levels) */
  oq_int *sp_Lp, *sp_Li, *sp_Rp, *sp_Rk, *sp_Rpos; /* pattern of L by columns (strict lower, rows ascending) and by rows (columns ascending) */
  oq_float *sp_Lx, *sp_D, *sp_y;                    /* values on that pattern, pivots, dense work vector (zero outside of use) */
  oq_int *sp_perm, *sp_iperm;                       /* optional symmetric permutation of the factor: P H P' = L D L', perm[new] = old (oq_set_perm; NULL: natural ordering) */
  oq_float *sp_b;                                   /* permuted right-hand side of a solve */
  /* settings / solution / info */
  oq_settings settings;
  oq_float *sol_x, *sol_y;
  oq_info info;
  struct timespec tic;
  /* statistics */
  oq_int n_refactor, n_factor_Q, n_updown_calls, n_rank1, n_solve, n_sigma_updates, n_boost_gamma;
  oq_int last_fact;
  /* NOT in the reference (stated deviation, restated by the engine: qpalm_iter.h, dev_solve): a Newton step whose direction comes out of an
   * UPDATED factor and is not finite (eta or beta of the line search is not finite: a pivot went through zero inside an update) is taken again
   * with a fresh factorisation.  The reference iterates on (solver_interface.c:357-368 looks at c->status only when !DLONG); on such a case
   * its iterates are NaN to max_iter.  guard = 0 (oq_set_scalar "newton_guard") restates the reference without it. */
  int guard, guard_spent;
  oq_int n_guard_refactor;
  oq_trace *trace;
};

/* =========================================================================================
 * lin_alg.c restated (src/lin_alg.c:11-203)
 * ======================================================================================= */
static oq_float *vec_dup(const oq_float *a, size_t n) { /* lin_alg.c:11-22 */
  oq_float *b = (oq_float *)malloc((n ? n : 1) * sizeof(oq_float));
  for (size_t i = 0; i < n; i++) b[i] = a[i];
  return b;
}
static void vec_cp(const oq_float *a, oq_float *b, size_t n) { for (size_t i = 0; i < n; i++) b[i] = a[i]; } /* :24-30 */
static void ivec_cp(const oq_int *a, oq_int *b, size_t n) { for (size_t i = 0; i < n; i++) b[i] = a[i]; }   /* :32-38 */
static void ivec_set(oq_int *a, oq_int sc, size_t n) { for (size_t i = 0; i < n; i++) a[i] = sc; }          /* :48-54 */

void oq_vec_set_scalar(oq_float *a, oq_float sc, size_t n) { for (size_t i = 0; i < n; i++) a[i] = sc; }     /* :40-46 */
void oq_vec_self_mult_scalar(oq_float *a, oq_float sc, size_t n) { for (size_t i = 0; i < n; i++) a[i] *= sc; } /* :56-62 */

/* lin_alg.c:72-86 -- groups of four products are summed before being added to the total (B2) */
oq_float oq_vec_prod(const oq_float *a, const oq_float *b, size_t n) {
  oq_float prod = 0.0;
  size_t i = 0;
  if (n >= 4) {
    for (; i <= n - 4; i += 4)
      prod += (a[i] * b[i] + a[i + 1] * b[i + 1] + a[i + 2] * b[i + 2] + a[i + 3] * b[i + 3]);
  }
  for (; i < n; i++) prod += a[i] * b[i];
  return prod;
}
void oq_vec_ew_prod(const oq_float *a, const oq_float *b, oq_float *c, size_t n) { for (size_t i = 0; i < n; i++) c[i] = a[i] * b[i]; } /* :92-98 */
void oq_vec_ew_div(const oq_float *a, const oq_float *b, oq_float *c, size_t n) { for (size_t i = 0; i < n; i++) c[i] = a[i] / b[i]; }  /* :101-107 */
void oq_vec_add_scaled(const oq_float *a, const oq_float *b, oq_float *c, oq_float sc, size_t n) { /* :110-116 */
  for (size_t i = 0; i < n; i++) c[i] = a[i] + sc * b[i];
}
void oq_vec_mult_add_scaled(oq_float *a, const oq_float *b, oq_float sc1, oq_float sc2, size_t n) { /* :118-124 */
  for (size_t i = 0; i < n; i++) a[i] = sc1 * a[i] + sc2 * b[i];
}
/* lin_alg.c:126-163 -- max is order independent, so a plain loop gives the identical value */
oq_float oq_vec_norm_inf(const oq_float *a, size_t n) {
  oq_float mx = 0.0;
  for (size_t i = 0; i < n; i++) { oq_float s = OQ_ABS(a[i]); mx = s > mx ? s : mx; }
  return mx;
}
void oq_vec_ew_recipr(const oq_float *a, oq_float *b, size_t n) { for (size_t i = 0; i < n; i++) b[i] = (oq_float)1.0 / a[i]; } /* :165-171 */
void oq_vec_ew_max_vec(const oq_float *a, const oq_float *b, oq_float *c, size_t n) { for (size_t i = 0; i < n; i++) c[i] = OQ_MAX(a[i], b[i]); } /* :173-179 */
void oq_vec_ew_min_vec(const oq_float *a, const oq_float *b, oq_float *c, size_t n) { for (size_t i = 0; i < n; i++) c[i] = OQ_MIN(a[i], b[i]); } /* :181-187 */
void oq_vec_ew_mid_vec(const oq_float *a, const oq_float *lo, const oq_float *hi, oq_float *c, size_t n) { /* :189-195 */
  for (size_t i = 0; i < n; i++) c[i] = OQ_MAX(lo[i], OQ_MIN(a[i], hi[i]));
}
void oq_vec_ew_sqrt(const oq_float *a, oq_float *b, size_t n) { for (size_t i = 0; i < n; i++) b[i] = sqrt(a[i]); } /* :197-203 */

/* =========================================================================================
 * sparse helpers = CHOLMOD calls at the reference's call sites
 * ======================================================================================= */
static void sp_alloc(oq_sparse *S, oq_int nrow, oq_int ncol, oq_int nzmax, int stype) {
  S->nrow = nrow; S->ncol = ncol; S->nzmax = nzmax; S->stype = stype;
  S->p = (oq_int *)calloc((size_t)ncol + 1, sizeof(oq_int));
  S->i = (oq_int *)calloc((size_t)(nzmax ? nzmax : 1), sizeof(oq_int));
  S->x = (oq_float *)calloc((size_t)(nzmax ? nzmax : 1), sizeof(oq_float));
}
static void sp_free(oq_sparse *S) { free(S->p); free(S->i); free(S->x); S->p = S->i = NULL; S->x = NULL; }
static void sp_copy_from(oq_sparse *S, oq_int nrow, oq_int ncol, const oq_int *p, const oq_int *i,
                         const oq_float *x, int stype) { /* cholmod copy_sparse, qpalm.c:141-143 */
  oq_int nz = p[ncol];
  sp_alloc(S, nrow, ncol, nz, stype);
  memcpy(S->p, p, ((size_t)ncol + 1) * sizeof(oq_int));
  if (nz) { memcpy(S->i, i, (size_t)nz * sizeof(oq_int)); memcpy(S->x, x, (size_t)nz * sizeof(oq_float)); }
}
/* cholmod transpose(A, values=1): n x m CSC of A' with sorted columns (iteration.c:81) */
static void sp_transpose(const oq_sparse *A, oq_sparse *T) {
  oq_int m = A->nrow, n = A->ncol, nz = A->p[n];
  sp_alloc(T, n, m, nz, 0);
  oq_int *cnt = (oq_int *)calloc((size_t)m + 1, sizeof(oq_int));
  for (oq_int k = 0; k < nz; k++) cnt[A->i[k] + 1]++;
  for (oq_int r = 0; r < m; r++) cnt[r + 1] += cnt[r];
  memcpy(T->p, cnt, ((size_t)m + 1) * sizeof(oq_int));
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) {
      oq_int r = A->i[k], dst = cnt[r]++;
      T->i[dst] = j; T->x[dst] = A->x[k];
    }
  free(cnt);
}
/* cholmod scale(S, CHOLMOD_COL, A): A <- A diag(s) */
static void sp_scale_col(oq_sparse *A, const oq_float *s) {
  for (oq_int j = 0; j < A->ncol; j++) { oq_float t = s[j]; for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= t; }
}
/* CHOLMOD_ROW: A <- diag(s) A */
static void sp_scale_row(oq_sparse *A, const oq_float *s) {
  for (oq_int j = 0; j < A->ncol; j++) for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= s[A->i[k]];
}
/* CHOLMOD_SYM: A <- diag(s) A diag(s) */
static void sp_scale_sym(oq_sparse *A, const oq_float *s) {
  for (oq_int j = 0; j < A->ncol; j++) { oq_float t = s[j]; for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= t * s[A->i[k]]; }
}
/* CHOLMOD_SCALAR */
static void sp_scale_scalar(oq_sparse *A, oq_float t) { oq_int nz = A->p[A->ncol]; for (oq_int k = 0; k < nz; k++) A->x[k] *= t; }

/* cholmod sdmult(A, transpose, alpha=1, beta=0, X, Y) (solver_interface.c:252-274).
 * stype < 0: only entries with row >= col are read and the matrix is symmetric (B6). */
static void sdmult(const oq_sparse *A, int transpose, const oq_float *X, oq_float *Y) {
  oq_int ncol = A->ncol, nrow = A->nrow;
  if (A->stype == 0) {
    if (!transpose) {
      for (oq_int i = 0; i < nrow; i++) Y[i] = 0.0;
      for (oq_int j = 0; j < ncol; j++) {
        oq_float xj = X[j];
        for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) Y[A->i[k]] += A->x[k] * xj;
      }
    } else {
      for (oq_int j = 0; j < ncol; j++) {
        oq_float yj = 0.0;
        for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) yj += A->x[k] * X[A->i[k]];
        Y[j] = yj;
      }
    }
  } else {
    for (oq_int i = 0; i < nrow; i++) Y[i] = 0.0;
    for (oq_int j = 0; j < ncol; j++) {
      oq_float xj = X[j], yj = 0.0;
      for (oq_int k = A->p[j]; k < A->p[j + 1]; k++) {
        oq_int i = A->i[k];
        oq_float a = A->x[k];
        if (A->stype < 0 ? i < j : i > j) continue; /* ignored triangle */
        if (i == j) yj += a * xj;
        else { Y[i] += a * xj; yj += a * X[i]; }
      }
      Y[j] += yj;
    }
  }
}
void oq_mat_vec(const oq_sparse *A, const oq_float *x, oq_float *y) { /* solver_interface.c:252-262 */
  if (x != y) sdmult(A, 0, x, y);
  else { oq_float *x2 = vec_dup(x, (size_t)A->ncol); sdmult(A, 0, x2, y); free(x2); }
}
void oq_mat_tpose_vec(const oq_sparse *A, const oq_float *x, oq_float *y) { /* solver_interface.c:264-274 */
  if (x != y) sdmult(A, 1, x, y);
  else { oq_float *x2 = vec_dup(x, (size_t)A->nrow); sdmult(A, 1, x2, y); free(x2); }
}
void oq_mat_inf_norm_cols(const oq_sparse *M, oq_float *E) { /* solver_interface.c:276-292 */
  for (oq_int j = 0; j < M->ncol; j++) E[j] = 0.;
  for (oq_int j = 0; j < M->ncol; j++)
    for (oq_int k = M->p[j]; k < M->p[j + 1]; k++) E[j] = OQ_MAX(OQ_ABS(M->x[k]), E[j]);
}
void oq_mat_inf_norm_rows(const oq_sparse *M, oq_float *E) { /* solver_interface.c:294-314 */
  for (oq_int j = 0; j < M->nrow; j++) E[j] = 0.;
  for (oq_int j = 0; j < M->ncol; j++)
    for (oq_int k = M->p[j]; k < M->p[j + 1]; k++) { oq_int i = M->i[k]; E[i] = OQ_MAX(OQ_ABS(M->x[k]), E[i]); }
}

/* =========================================================================================
 * dense LDL^T (natural order, no pivoting, negative pivots accepted)
 * ======================================================================================= */
/* Up-looking (row by row) recurrence of the simplicial LDL^T:
 *   row k: y = H(0:k-1,k); for i = 0..k-1: yi = y[i]; y[i+1:k-1] -= L(i+1:k-1,i)*yi;
 *          l_ki = yi/d_i; d_k -= l_ki*yi.
 * H holds the lower triangle column-major; row k of H is read from H(k,0:k-1). */
void oq_dense_ldl_factor(oq_int n, oq_float *H, oq_int ld, oq_float *D) {
  oq_float *y = (oq_float *)malloc((size_t)(n ? n : 1) * sizeof(oq_float));
  for (oq_int k = 0; k < n; k++) {
    oq_float dk = H[k + k * ld];
    for (oq_int i = 0; i < k; i++) y[i] = H[k + i * ld];
    for (oq_int i = 0; i < k; i++) {
      oq_float yi = y[i];
      const oq_float *Li = H + i * ld;
      for (oq_int r = i + 1; r < k; r++) y[r] -= Li[r] * yi;
      oq_float lki = yi / D[i];
      dk -= lki * yi;
      H[k + i * ld] = lki;
    }
    D[k] = dk;
  }
  free(y);
}
/* cholmod solve(CHOLMOD_LDLt): L y = b (column oriented), y /= D, L' x = y (column dots) */
void oq_dense_ldl_solve(oq_int n, const oq_float *L, oq_int ld, const oq_float *D, oq_float *b) {
  for (oq_int j = 0; j < n; j++) {
    oq_float yj = b[j];
    const oq_float *Lj = L + j * ld;
    for (oq_int i = j + 1; i < n; i++) b[i] -= Lj[i] * yj;
  }
  for (oq_int j = 0; j < n; j++) b[j] /= D[j];
  for (oq_int j = n - 1; j >= 0; j--) {
    oq_float xj = b[j];
    const oq_float *Lj = L + j * ld;
    for (oq_int i = j + 1; i < n; i++) xj -= Lj[i] * b[i];
    b[j] = xj;
  }
}
/* Davis & Hager method C1 as coded in CHOLMOD's updown numeric kernel:
 *   a = alpha +/- w_j^2/d_j ; d_j <- d_j*a ; gamma = -/+ w_j/d_j ; d_j <- d_j/alpha ; alpha <- a
 *   for i > j: w_i -= w_j*l_ij ; l_ij -= gamma*w_i
 * (columns before the first nonzero of w are untouched). */
void oq_dense_ldl_rank1(oq_int n, oq_float *L, oq_int ld, oq_float *D, oq_float *w, int update) {
  oq_float alpha = 1.0;
  oq_int j0 = 0;
  while (j0 < n && w[j0] == 0.0) j0++;
#ifdef OQ_PIVOT_ENGINE
  /* TEST VARIANT ONLY (tests/fuzz_cases.py: oracle_variants, "pivot"): the same recurrence in the algebraically equal form the
   * device code uses (qpalm_amd/csrc/qpalm_dense.h, dense_updown: 1/alpha carried along, d_new = d + s w^2 / alpha, one reciprocal
   * per pivot instead of three divisions).  Like the -ffp-contract / -Ofast builds it answers one question: is the iteration count
   * of a case a property of the algorithm, or of how this recurrence is rounded? */
  {
    const oq_float sgn = update ? 1.0 : -1.0;
    oq_float ialpha = 1.0;
    for (oq_int j = j0; j < n; j++) {
      const oq_float wj = w[j], dprev = D[j];
      const oq_float pinc = sgn * wj * wj * ialpha;
      const oq_float dnew = dprev + pinc;
      const oq_float rdn = 1.0 / dnew, rdp = 1.0 / dprev;
      const oq_float gam = -sgn * wj * ialpha * rdn;
      if (wj != 0.0) { alpha = alpha * dnew * rdp; ialpha = ialpha * dprev * rdn; }
      D[j] = dnew;
      oq_float *Lj = L + j * ld;
      for (oq_int i = j + 1; i < n; i++) {
        oq_float wi = w[i] - wj * Lj[i];
        w[i] = wi;
        Lj[i] -= gam * wi;
      }
    }
    return;
  }
#endif
  for (oq_int j = j0; j < n; j++) {
    oq_float wj = w[j];
    oq_float dj = D[j];
    oq_float a, gam;
    if (update) { a = alpha + (wj * wj) / dj; dj *= a; gam = -wj / dj; }
    else        { a = alpha - (wj * wj) / dj; dj *= a; gam =  wj / dj; }
    dj /= alpha;
    alpha = a;
    D[j] = dj;
    oq_float *Lj = L + j * ld;
    for (oq_int i = j + 1; i < n; i++) {
      oq_float wi = w[i] - wj * Lj[i];
      w[i] = wi;
      Lj[i] -= gam * wi;
    }
  }
}

static void factor_alloc(oq_factor *F, oq_int n) {
  if (!F->L) { F->L = (oq_float *)calloc(nz1((size_t)n * (size_t)n), sizeof(oq_float)); F->D = (oq_float *)calloc((size_t)(n ? n : 1), sizeof(oq_float)); }
}
static void factor_free(oq_factor *F) { free(F->L); free(F->D); F->L = F->D = NULL; F->valid = 0; }

/* analyze + factorize_p(M, beta) (solver_interface.c:347-356): M symmetric, lower triangle used */
static void factor_sparse_lower(oq_workspace *w, const oq_sparse *M, oq_factor *F, int add_beta, oq_float beta) {
  oq_int n = w->n;
  factor_alloc(F, n);
  memset(F->L, 0, (size_t)(n * n) * sizeof(oq_float));
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = M->p[j]; k < M->p[j + 1]; k++) { oq_int i = M->i[k]; if (i >= j) F->L[i + j * n] += M->x[k]; }
  if (add_beta) for (oq_int j = 0; j < n; j++) F->L[j + j * n] += beta;
  oq_dense_ldl_factor(n, F->L, n, F->D);
  F->valid = 1;
}


/* =========================================================================================
 * nonconvex.c restated: lobpcg (:29-168), set_settings_nonconvex (:171-183).  gershgorin_max (:185-210) is above.
 * Third-party pieces absent from the reference tree, restated from their published definitions:
 *   - rand(): the reference seeds the start vector from the UNSEEDED C library generator (B9), i.e. glibc's default
 *     state (seed 1) in a fresh process.  glibc's TYPE_3 generator is restated (r_i = r_{i-3} + r_{i-31} on 32 bits,
 *     seeded by the Lehmer sequence 16807 r mod 2^31-1, first 310 outputs discarded, result >> 1; RAND_MAX = 2^31-1);
 *     tests check it against the C library of the build host.
 *   - LAPACKE_dsyev on the 2 x 2 and LAPACKE_dsygv (itype 1) on the 3 x 3 compressed pencil: eigenvalues ascending,
 *     eigenvectors C-normalised.  Restated with a cyclic Jacobi iteration on G^{-1} B G^{-T} (C = G G', Cholesky),
 *     which is what dsygv does up to its QR iteration; the sign of an eigenvector is LAPACK's choice and does not
 *     influence lambda (x -> -x flips w and p consistently).
 * ======================================================================================= */
typedef struct { int32_t r[34]; int f, b; } oq_glibc_rand;
static void oq_srand(oq_glibc_rand *g, unsigned int seed) {
  int32_t word = seed ? (int32_t)seed : 1;
  int32_t tab[344];
  tab[0] = word;
  for (int i = 1; i < 31; i++) {
    long hi = tab[i - 1] / 127773, lo = tab[i - 1] % 127773;
    long v = 16807 * lo - 2836 * hi;
    if (v < 0) v += 2147483647;
    tab[i] = (int32_t)v;
  }
  for (int i = 31; i < 34; i++) tab[i] = tab[i - 31];
  for (int i = 34; i < 344; i++) tab[i] = (int32_t)((uint32_t)tab[i - 31] + (uint32_t)tab[i - 3]);
  for (int i = 0; i < 34; i++) g->r[i] = tab[310 + i];
  g->f = 0; (void)g->b;
}
static int oq_rand(oq_glibc_rand *g) { /* o_k = o_{k-31} + o_{k-3}; the 34 newest values are kept in a ring */
  const int k = g->f;                                      /* slot of o_{k-34} -> becomes o_k */
  const uint32_t v = (uint32_t)g->r[(k + 3) % 34] + (uint32_t)g->r[(k + 31) % 34];
  g->r[k] = (int32_t)v;
  g->f = (k + 1) % 34;
  return (int)(v >> 1);
}
int oq_rand_sequence(unsigned int seed, int count, int *out) { /* for the test against the C library */
  oq_glibc_rand g; oq_srand(&g, seed);
  for (int i = 0; i < count; i++) out[i] = oq_rand(&g);
  return 0;
}

static oq_float vec_norm_two(const oq_float *a, size_t n) { return sqrt(oq_vec_prod(a, a, n)); } /* lin_alg.c:165-167 */

/* smallest eigenpair of the symmetric-definite pencil (B, C) of order dim (2 or 3), C = I for dim 2 */
static oq_float small_eig(int dim, oq_float B[3][3], oq_float Cm[3][3], oq_float y[3]) {
  oq_float G[3][3] = {{0}}, M[3][3], T[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int j = 0; j < dim; j++) { /* C = G G' */
    oq_float s = Cm[j][j];
    for (int k = 0; k < j; k++) s -= G[j][k] * G[j][k];
    G[j][j] = sqrt(s);
    for (int i = j + 1; i < dim; i++) { oq_float t = Cm[i][j]; for (int k = 0; k < j; k++) t -= G[i][k] * G[j][k]; G[i][j] = t / G[j][j]; }
  }
  for (int c = 0; c < dim; c++) /* T = G^{-1} B (forward substitution on every column) */
    for (int i = 0; i < dim; i++) { oq_float t = B[i][c]; for (int k = 0; k < i; k++) t -= G[i][k] * T[k][c]; T[i][c] = t / G[i][i]; }
  for (int r = 0; r < dim; r++) /* M = T G^{-T}: rows of T solved against G */
    for (int i = 0; i < dim; i++) { oq_float t = T[r][i]; for (int k = 0; k < i; k++) t -= G[i][k] * M[r][k]; M[r][i] = t / G[i][i]; }
  for (int i = 0; i < dim; i++) for (int j = 0; j < i; j++) { const oq_float a = 0.5 * (M[i][j] + M[j][i]); M[i][j] = a; M[j][i] = a; }
  for (int sweep = 0; sweep < 30; sweep++) { /* cyclic Jacobi */
    oq_float off = 0;
    for (int i = 0; i < dim; i++) for (int j = 0; j < i; j++) off += M[i][j] * M[i][j];
    if (off == 0.0) break;
    for (int p = 0; p < dim; p++)
      for (int q = p + 1; q < dim; q++) {
        if (M[p][q] == 0.0) continue;
        const oq_float theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
        const oq_float t = (theta >= 0 ? 1.0 : -1.0) / (OQ_ABS(theta) + sqrt(theta * theta + 1.0));
        const oq_float cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < dim; k++) { const oq_float a = M[k][p], b = M[k][q]; M[k][p] = cs * a - sn * b; M[k][q] = sn * a + cs * b; }
        for (int k = 0; k < dim; k++) { const oq_float a = M[p][k], b = M[q][k]; M[p][k] = cs * a - sn * b; M[q][k] = sn * a + cs * b; }
        for (int k = 0; k < dim; k++) { const oq_float a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
      }
  }
  int kmin = 0;
  for (int k = 1; k < dim; k++) if (M[k][k] < M[kmin][kmin]) kmin = k;
  for (int i = dim - 1; i >= 0; i--) { /* y = G^{-T} v */
    oq_float t = V[i][kmin];
    for (int k = i + 1; k < dim; k++) t -= G[k][i] * y[k];
    y[i] = t / G[i][i];
  }
  return M[kmin][kmin];
}

static oq_float lobpcg(oq_workspace *w) { /* nonconvex.c:29-168 with x == NULL */
  const size_t n = (size_t)w->n;
  const oq_sparse *A = &w->Q;
  oq_float *x = w->d, *Ax = w->Qd, *wv = w->neg_dphi, *Aw = w->Atyh, *p = w->temp_n, *Ap = w->xx0;
  oq_glibc_rand g; oq_srand(&g, 1);
  for (size_t i = 0; i < n; i++) x[i] = (oq_float)oq_rand(&g) / 2147483647;
  oq_vec_self_mult_scalar(x, 1.0 / vec_norm_two(x, n), n);
  oq_mat_vec(A, x, Ax);
  oq_float lambda = oq_vec_prod(x, Ax, n);
  oq_vec_add_scaled(Ax, x, wv, -lambda, n);
  oq_vec_add_scaled(wv, x, wv, -oq_vec_prod(x, wv, n), n);
  oq_vec_self_mult_scalar(wv, 1.0 / vec_norm_two(wv, n), n);
  oq_mat_vec(A, wv, Aw);
  oq_float xAw = oq_vec_prod(Aw, x, n), wAw = oq_vec_prod(Aw, wv, n);
  oq_float B[3][3] = {{lambda, xAw, 0}, {xAw, wAw, 0}, {0, 0, 0}}, Cm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, y[3] = {0, 0, 0};
  lambda = small_eig(2, B, Cm, y);
  for (size_t i = 0; i < n; i++) { p[i] = wv[i] * y[1]; Ap[i] = Aw[i] * y[1]; } /* vec_mult_scalar */
  oq_vec_add_scaled(p, x, x, y[0], n);
  oq_vec_add_scaled(Ap, Ax, Ax, y[0], n);
  w->n_lobpcg_iter = 0;
  for (size_t it = 0; it < 1000; it++) {
    oq_vec_add_scaled(Ax, x, wv, -lambda, n);
    if (oq_vec_norm_inf(wv, n) < 1e-5) { /* LOBPCG_TOL */
      const oq_float norm_w = vec_norm_two(wv, n);
      lambda -= sqrt(2) * norm_w + 1e-6;
      if (n <= 3) lambda -= 1e-6;
      break;
    }
    w->n_lobpcg_iter++;
    oq_vec_add_scaled(wv, x, wv, -oq_vec_prod(x, wv, n), n);
    oq_vec_self_mult_scalar(wv, 1.0 / vec_norm_two(wv, n), n);
    oq_mat_vec(A, wv, Aw);
    xAw = oq_vec_prod(Ax, wv, n);
    wAw = oq_vec_prod(wv, Aw, n);
    const oq_float p_norm_inv = 1.0 / vec_norm_two(p, n);
    oq_vec_self_mult_scalar(p, p_norm_inv, n);
    oq_vec_self_mult_scalar(Ap, p_norm_inv, n);
    const oq_float xAp = oq_vec_prod(Ax, p, n), wAp = oq_vec_prod(Aw, p, n), pAp = oq_vec_prod(Ap, p, n);
    const oq_float xp = oq_vec_prod(x, p, n), wp = oq_vec_prod(wv, p, n);
    oq_float B3[3][3] = {{lambda, xAw, xAp}, {xAw, wAw, wAp}, {xAp, wAp, pAp}}, C3[3][3] = {{1, 0, xp}, {0, 1, wp}, {xp, wp, 1.0}};
    lambda = small_eig(3, B3, C3, y);
    oq_vec_mult_add_scaled(p, wv, y[2], y[1], n);
    oq_vec_mult_add_scaled(Ap, Aw, y[2], y[1], n);
    oq_vec_mult_add_scaled(x, p, y[0], 1, n);
    oq_vec_mult_add_scaled(Ax, Ap, y[0], 1, n);
  }
  return lambda;
}

static void set_settings_nonconvex(oq_workspace *w) { /* nonconvex.c:171-183 */
  const oq_float lambda = lobpcg(w);
  w->lobpcg_lambda = lambda;
  if (lambda < 0) {
    w->settings.proximal = 1;
    w->settings.gamma_init = 1 / OQ_ABS(lambda);
    w->settings.gamma_max = w->settings.gamma_init;
    w->gamma_maxed = 1;
  } else w->settings.nonconvex = 0;
}

/* =========================================================================================
 * timers / status (src/util.c:61-105, :283-303)
 * ======================================================================================= */
static void tic(oq_workspace *w) { clock_gettime(CLOCK_MONOTONIC, &w->tic); }
static oq_float toc(oq_workspace *w) {
  struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
  return (oq_float)(t.tv_sec - w->tic.tv_sec) + 1e-9 * (oq_float)(t.tv_nsec - w->tic.tv_nsec);
}
static void update_status(oq_info *info, oq_int v) {
  info->status_val = v;
  const char *s;
  switch (v) {
    case OQ_SOLVED: s = "solved"; break;
    case OQ_DUAL_TERMINATED: s = "dual terminated"; break;
    case OQ_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
    case OQ_DUAL_INFEASIBLE: s = "dual infeasible"; break;
    case OQ_TIME_LIMIT_REACHED: s = "time limit exceeded"; break;
    case OQ_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
    case OQ_UNSOLVED: s = "unsolved"; break;
    case OQ_ERROR: s = "error"; break;
    default: s = "unrecognised status value"; break;
  }
  memset(info->status, 0, sizeof(info->status));
  strncpy(info->status, s, sizeof(info->status) - 1);
}

/* =========================================================================================
 * settings (src/qpalm.c:38-70, src/validate.c:18-221)
 * ======================================================================================= */
void oq_set_default_settings(oq_settings *s) {
  s->max_iter = 10000; s->inner_max_iter = 100; s->eps_abs = 1e-4; s->eps_rel = 1e-4;
  s->eps_abs_in = 1; s->eps_rel_in = 1; s->rho = 0.1; s->eps_prim_inf = 1e-5; s->eps_dual_inf = 1e-5;
  s->theta = 0.25; s->delta = 100; s->sigma_max = 1e9; s->sigma_init = 2e1; s->proximal = 1;
  s->gamma_init = 1e7; s->gamma_upd = 10; s->gamma_max = 1e7; s->scaling = 10; s->nonconvex = 0;
  s->verbose = 1; s->print_iter = 1; s->warm_start = 0; s->reset_newton_iter = 10000;
  s->enable_dual_termination = 0; s->dual_objective_limit = OQ_INFTY; s->time_limit = OQ_INFTY;
  s->ordering = 0; s->factorization_method = 2; s->max_rank_update = 160; s->max_rank_update_fraction = 0.1;
}
static int validate_settings(const oq_settings *s) {
  if (!s) return 0;
  if (s->max_iter <= 0 || s->inner_max_iter <= 0) return 0;
  if (s->eps_abs < 0 || s->eps_rel < 0 || (s->eps_rel == 0 && s->eps_abs == 0)) return 0;
  if (s->eps_abs_in < 0 || s->eps_rel_in < 0 || (s->eps_rel_in == 0 && s->eps_abs_in == 0)) return 0;
  if (s->rho <= 0 || s->rho >= 1) return 0;
  if (s->eps_prim_inf < 0 || s->eps_dual_inf < 0) return 0;
  if (s->theta > 1 || s->delta <= 1 || s->sigma_max <= 0) return 0;
  if (s->proximal != 0 && s->proximal != 1) return 0;
  if (s->gamma_init <= 0 || s->gamma_upd < 1 || s->gamma_max < s->gamma_init) return 0;
  if (s->scaling < 0) return 0;
  if (s->warm_start != 0 && s->warm_start != 1) return 0;
  if (s->verbose != 0 && s->verbose != 1) return 0;
  if (s->print_iter <= 0 || s->reset_newton_iter <= 0) return 0;
  if (s->enable_dual_termination != 0 && s->enable_dual_termination != 1) return 0;
  return 1;
}

/* =========================================================================================
 * scaling.c restated (src/scaling.c:25-113)
 * ======================================================================================= */
static void limit_scaling(oq_float *D, size_t n) { for (size_t i = 0; i < n; i++) D[i] = D[i] < 1e-12 ? 1.0 : D[i]; }

void oq_scale_data(oq_workspace *w) {
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_vec_set_scalar(w->D, 1, n);
  oq_vec_set_scalar(w->E, 1, m);
  for (oq_int it = 0; it < w->settings.scaling; it++) { /* Ruiz on A, scaling.c:47-81 */
    oq_mat_inf_norm_cols(&w->A, w->D_temp);
    oq_mat_inf_norm_rows(&w->A, w->E_temp);
    limit_scaling(w->D_temp, n);
    limit_scaling(w->E_temp, m);
    oq_vec_ew_sqrt(w->D_temp, w->D_temp, n);
    oq_vec_ew_sqrt(w->E_temp, w->E_temp, m);
    oq_vec_ew_recipr(w->D_temp, w->D_temp, n);
    oq_vec_ew_recipr(w->E_temp, w->E_temp, m);
    sp_scale_row(&w->A, w->E_temp);
    sp_scale_col(&w->A, w->D_temp);
    oq_vec_ew_prod(w->D, w->D_temp, w->D, n);
    oq_vec_ew_prod(w->E, w->E_temp, w->E, m);
  }
  /* Q <- c D Q D, q <- c D q (scaling.c:83-102); Qx is whatever the workspace holds (zeros at setup) */
  oq_vec_ew_prod(w->D, w->q, w->q, n);
  oq_vec_ew_prod(w->D, w->Qx, w->Qx, n);
  vec_cp(w->D, w->D_temp, n);
  oq_vec_add_scaled(w->Qx, w->q, w->dphi, 1, n);
  w->sc_c = 1 / OQ_MAX(1.0, oq_vec_norm_inf(w->dphi, n));
  oq_vec_self_mult_scalar(w->q, w->sc_c, n);
  sp_scale_sym(&w->Q, w->D_temp);
  sp_scale_scalar(&w->Q, w->sc_c);
  /* scaling.c:104-111 */
  oq_vec_ew_recipr(w->D, w->Dinv, n);
  oq_vec_ew_recipr(w->E, w->Einv, m);
  w->sc_cinv = (oq_float)1.0 / w->sc_c;
  oq_vec_ew_prod(w->E, w->bmin, w->bmin, m);
  oq_vec_ew_prod(w->E, w->bmax, w->bmax, m);
}

/* =========================================================================================
 * setup / cleanup (src/qpalm.c:73-319, :874-1096)
 * ======================================================================================= */
static oq_float *zalloc(size_t n) { return (oq_float *)calloc(n ? n : 1, sizeof(oq_float)); }
static oq_int *izalloc(size_t n) { return (oq_int *)calloc(n ? n : 1, sizeof(oq_int)); }

oq_workspace *oq_setup(oq_int n_, oq_int m_, const oq_int *Qp, const oq_int *Qi, const oq_float *Qx,
                       const oq_int *Ap, const oq_int *Ai, const oq_float *Ax, const oq_float *q, oq_float c,
                       const oq_float *bmin, const oq_float *bmax, const oq_settings *settings) {
  size_t n = (size_t)n_, m = (size_t)m_;
  for (size_t j = 0; j < m; j++) if (bmin[j] > bmax[j]) return NULL; /* validate.c:31-39 */
  if (!validate_settings(settings)) return NULL;
  oq_workspace *w = (oq_workspace *)calloc(1, sizeof(oq_workspace));
  if (!w) return NULL;
  w->guard = 1;
  tic(w);
  w->settings = *settings;
  w->sqrt_delta = sqrt(w->settings.delta);
  w->gamma = w->settings.gamma_init;
  w->n = n_; w->m = m_;
  w->bmin = vec_dup(bmin, m); w->bmax = vec_dup(bmax, m); w->q = vec_dup(q, n); w->c = c;
  sp_copy_from(&w->A, m_, n_, Ap, Ai, Ax, 0);
  sp_copy_from(&w->Q, n_, n_, Qp, Qi, Qx, -1);
  w->x = zalloc(n); w->y = zalloc(m); w->Ax = zalloc(m); w->Qx = zalloc(n); w->x_prev = zalloc(n);
  w->Aty = zalloc(n); w->x0 = zalloc(n); w->initialized = 0;
  w->temp_m = zalloc(m); w->temp_n = zalloc(n); w->sigma = zalloc(m); w->sigma_inv = zalloc(m);
  w->nb_sigma_changed = 0;
  w->z = zalloc(m); w->Axys = zalloc(m); w->pri_res = zalloc(m); w->pri_res_in = zalloc(m); w->df = zalloc(n);
  w->xx0 = zalloc(n); w->dphi = zalloc(n); w->dphi_prev = zalloc(n);
  w->sqrt_sigma = zalloc(m); w->delta = zalloc(2 * m); w->alpha = zalloc(2 * m); w->delta2 = zalloc(2 * m);
  w->delta_alpha = zalloc(2 * m); w->temp_2m = zalloc(2 * m);
  w->s = (oq_array_element *)calloc(nz1(2 * m), sizeof(oq_array_element));
  w->index_L = izalloc(2 * m); w->index_P = izalloc(2 * m); w->index_J = izalloc(2 * m);
  w->delta_y = zalloc(m); w->Atdelta_y = zalloc(n); w->delta_x = zalloc(n); w->Qdelta_x = zalloc(n); w->Adelta_x = zalloc(m);
  w->D_temp = zalloc(n); w->E_temp = zalloc(m);
  w->Hbuf = NULL; w->wbuf = zalloc(n);
  if (settings->scaling) {
    w->has_scaling = 1;
    w->D = zalloc(n); w->Dinv = zalloc(n); w->E = zalloc(m); w->Einv = zalloc(m);
    oq_scale_data(w);
  } else w->has_scaling = 0;
  w->active = izalloc(m); w->active_old = izalloc(m);
  ivec_set(w->active_old, 0, m);
  w->reset_newton = 1;
  w->enter = izalloc(m); w->leave = izalloc(m);
  w->neg_dphi = zalloc(n); w->d = zalloc(n); w->Qd = zalloc(n); w->Ad = zalloc(m); w->yh = zalloc(m); w->Atyh = zalloc(n);
  w->At_scale = zalloc(m);
  /* qpalm_set_factorization_method (solver_interface.c:20-75): the CHOLMOD build forces SCHUR (:72-74); under LADEL
   * the setting is honoured and KKT_OR_SCHUR applies an nnz-based rule.  The oracle takes KKT only when it is asked
   * for explicitly (FACTORIZE_KKT = 0); KKT_OR_SCHUR (2) keeps the CHOLMOD build's answer. */
  w->kkt_mode = (settings->factorization_method == 0);
  w->first_factorization = 1; /* qpalm.c:272 (set under USE_LADEL only; only the KKT path reads it here) */
  if (w->kkt_mode) {
    w->kkt_state = izalloc(m);
    w->rhs_kkt = zalloc(n + m); w->sol_kkt = zalloc(n + m); w->kkt_tmp = zalloc(n + m);
  }
  if (w->settings.nonconvex) set_settings_nonconvex(w); /* qpalm.c:293-296 */
  w->sol_x = zalloc(n); w->sol_y = zalloc(m);
  update_status(&w->info, OQ_UNSOLVED);
  w->info.solve_time = 0.0; w->info.run_time = 0.0;
  w->info.setup_time = toc(w);
  return w;
}

void oq_cleanup(oq_workspace *w) {
  if (w) free(w->wblock);
  if (!w) return;
  sp_free(&w->A); sp_free(&w->Q); if (w->At_sqrt_sigma.p) sp_free(&w->At_sqrt_sigma);
  oq_float **fv[] = {&w->q, &w->bmin, &w->bmax, &w->x, &w->y, &w->Ax, &w->Qx, &w->Aty, &w->x_prev, &w->x0,
    &w->temp_m, &w->temp_n, &w->sigma, &w->sigma_inv, &w->sqrt_sigma, &w->Axys, &w->z, &w->pri_res, &w->pri_res_in,
    &w->yh, &w->Atyh, &w->df, &w->xx0, &w->dphi, &w->neg_dphi, &w->dphi_prev, &w->d, &w->Qd, &w->Ad, &w->delta,
    &w->alpha, &w->temp_2m, &w->delta2, &w->delta_alpha, &w->delta_y, &w->Atdelta_y, &w->delta_x, &w->Qdelta_x,
    &w->Adelta_x, &w->D_temp, &w->E_temp, &w->D, &w->Dinv, &w->E, &w->Einv, &w->At_scale, &w->Hbuf, &w->wbuf,
    &w->sol_x, &w->sol_y};
  for (size_t k = 0; k < sizeof(fv) / sizeof(fv[0]); k++) { free(*fv[k]); *fv[k] = NULL; }
  free(w->s); free(w->index_L); free(w->index_P); free(w->index_J);
  free(w->active); free(w->active_old); free(w->enter); free(w->leave);
  factor_free(&w->LD); factor_free(&w->LD_Q); factor_free(&w->LDK);
  free(w->kkt_state); free(w->rhs_kkt); free(w->sol_kkt); free(w->kkt_tmp);
  if (w->At.p) sp_free(&w->At);
  free(w->sp_Lp); free(w->sp_Li); free(w->sp_Rp); free(w->sp_Rk); free(w->sp_Rpos); free(w->sp_Lx); free(w->sp_D); free(w->sp_y); free(w->sp_perm); free(w->sp_iperm); free(w->sp_b);
  free(w);
}

/* =========================================================================================
 * iteration.c restated
 * ======================================================================================= */
static oq_float compute_objective(oq_workspace *w) { /* iteration.c:231-270 (grouped by four, B2) */
  oq_float obj = 0;
  size_t n = (size_t)w->n, i = 0;
  const oq_float *Qx = w->Qx, *x = w->x, *q = w->q;
  if (w->settings.proximal) {
    oq_float g = w->gamma;
    if (n >= 4)
      for (; i <= n - 4; i += 4)
        obj += (0.5 * (Qx[i] - 1 / g * x[i]) + q[i]) * x[i] + (0.5 * (Qx[i + 1] - 1 / g * x[i + 1]) + q[i + 1]) * x[i + 1]
             + (0.5 * (Qx[i + 2] - 1 / g * x[i + 2]) + q[i + 2]) * x[i + 2] + (0.5 * (Qx[i + 3] - 1 / g * x[i + 3]) + q[i + 3]) * x[i + 3];
    for (; i < n; i++) obj += (0.5 * (Qx[i] - 1 / g * x[i]) + q[i]) * x[i];
  } else {
    if (n >= 4)
      for (; i <= n - 4; i += 4)
        obj += (0.5 * Qx[i] + q[i]) * x[i] + (0.5 * Qx[i + 1] + q[i + 1]) * x[i + 1]
             + (0.5 * Qx[i + 2] + q[i + 2]) * x[i + 2] + (0.5 * Qx[i + 3] + q[i + 3]) * x[i + 3];
    for (; i < n; i++) obj += (0.5 * Qx[i] + q[i]) * x[i];
  }
  if (w->has_scaling) obj *= w->sc_cinv;
  obj += w->c;
  return obj;
}

static oq_float compute_dual_objective(oq_workspace *w) { /* iteration.c:272-299 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_float dobj = 0;
  oq_vec_add_scaled(w->Aty, w->q, w->neg_dphi, 1.0, n);
  vec_cp(w->neg_dphi, w->D_temp, n);
  oq_dense_ldl_solve(w->n, w->LD_Q.L, w->n, w->LD_Q.D, w->D_temp);
  dobj -= 0.5 * oq_vec_prod(w->neg_dphi, w->D_temp, n);
  for (size_t i = 0; i < m; i++) dobj -= w->y[i] > 0 ? w->y[i] * w->bmax[i] : w->y[i] * w->bmin[i];
  if (w->has_scaling) dobj *= w->sc_cinv;
  dobj += w->c;
  return dobj;
}

void oq_compute_residuals(oq_workspace *w) { /* iteration.c:24-48 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_vec_ew_prod(w->y, w->sigma_inv, w->temp_m, m);
  oq_vec_add_scaled(w->Ax, w->temp_m, w->Axys, 1, m);
  oq_vec_ew_mid_vec(w->Axys, w->bmin, w->bmax, w->z, m);
  oq_vec_add_scaled(w->Ax, w->z, w->pri_res, -1, m);
  oq_vec_ew_prod(w->pri_res, w->sigma, w->temp_m, m);
  oq_vec_add_scaled(w->y, w->temp_m, w->yh, 1, m);
  oq_vec_add_scaled(w->Qx, w->q, w->df, 1, n);
  if (w->settings.proximal) oq_vec_add_scaled(w->df, w->x0, w->df, -1 / w->gamma, n);
  oq_mat_tpose_vec(&w->A, w->yh, w->Atyh);
  oq_vec_add_scaled(w->df, w->Atyh, w->dphi, 1, n);
}

static void initialize_sigma(oq_workspace *w) { /* iteration.c:50-84 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_float f = 0.5 * oq_vec_prod(w->x, w->Qx, n) + oq_vec_prod(w->q, w->x, n);
  oq_vec_ew_mid_vec(w->Ax, w->bmin, w->bmax, w->temp_m, m);
  oq_vec_add_scaled(w->Ax, w->temp_m, w->temp_m, -1, m);
  oq_float dist2 = oq_vec_prod(w->temp_m, w->temp_m, m);
  oq_vec_set_scalar(w->sigma, OQ_MAX(1e-4, OQ_MIN(w->settings.sigma_init * OQ_MAX(1, OQ_ABS(f)) / OQ_MAX(1, 0.5 * dist2), 1e4)), m);
  oq_vec_ew_recipr(w->sigma, w->sigma_inv, m);
  oq_vec_ew_sqrt(w->sigma, w->sqrt_sigma, m);
  w->sqrt_sigma_max = sqrt(w->settings.sigma_max);
  vec_cp(w->sqrt_sigma, w->At_scale, m);
  if (w->At_sqrt_sigma.p) sp_free(&w->At_sqrt_sigma);
  sp_transpose(&w->A, &w->At_sqrt_sigma);
  sp_scale_col(&w->At_sqrt_sigma, w->At_scale);
}

static int sparse_update_pays(const oq_workspace *w, oq_int nchange);
void oq_update_sigma(oq_workspace *w) { /* iteration.c:86-145 */
  size_t m = (size_t)w->m;
  const oq_settings *st = &w->settings;
  w->nb_sigma_changed = 0;
  w->n_sigma_updates++;
  oq_float *At_scalex = w->At_scale;
  oq_float pri_res_unscaled_norm = oq_vec_norm_inf(w->pri_res, m);
  oq_int *sigma_changed = w->enter; /* B4: enter[] reused as scratch */
  for (size_t k = 0; k < m; k++) {
    if ((OQ_ABS(w->pri_res[k]) > st->theta * OQ_ABS(w->pri_res_in[k])) && w->active[k]) { /* B13 */
      oq_float mult_factor = OQ_MAX(1.0, st->delta * OQ_ABS(w->pri_res[k]) / (pri_res_unscaled_norm + 1e-6));
      oq_float sigma_temp = mult_factor * w->sigma[k];
      if (sigma_temp <= st->sigma_max) {
        if (w->sigma[k] != sigma_temp) sigma_changed[w->nb_sigma_changed++] = (oq_int)k;
        w->sigma[k] = sigma_temp;
        w->sigma_inv[k] = 1.0 / sigma_temp;
        mult_factor = sqrt(mult_factor);
        w->sqrt_sigma[k] = mult_factor * w->sqrt_sigma[k];
        At_scalex[k] = mult_factor;
      } else {
        if (w->sigma[k] != st->sigma_max) sigma_changed[w->nb_sigma_changed++] = (oq_int)k;
        w->sigma[k] = st->sigma_max;
        w->sigma_inv[k] = 1.0 / st->sigma_max;
        At_scalex[k] = w->sqrt_sigma_max / w->sqrt_sigma[k];
        w->sqrt_sigma[k] = w->sqrt_sigma_max;
      }
    } else At_scalex[k] = 1.0;
  }
  sp_scale_col(&w->At_sqrt_sigma, w->At_scale);
  /* first_factorization exists only under USE_LADEL; the CHOLMOD branch reads the calloc'ed 0 */
  if (w->kkt_mode) {
    /* iteration.c:135-144 under FACTORIZE_KKT: either branch ends in reset_newton = TRUE.  The rank-1 path
     * (solver_interface.c:463-481) updates the factor at row pinv[row] -- a VARIABLE's row, not the constraint's --
     * and then forces reset_newton, so the next Newton step refactorises and the update has no effect on any iterate;
     * it is not restated. */
    if (w->first_factorization || (st->proximal && w->gamma < st->gamma_max) || w->nb_sigma_changed > 0 ||
        (w->nb_sigma_changed > OQ_MIN(st->max_rank_update_fraction * (w->n + w->m), 0.25 * st->max_rank_update)))
      w->reset_newton = 1;
    return;
  }
  if (w->sparse_mode) {
    /* sparse storage: the reference's rule (below) with one more condition -- the changed rows are applied as path updates only where walking
     * their elimination-tree paths is cheaper than rebuilding the factor (sparse_update_pays: the engine's rule, mode 1); in mode 2 every
     * change refactorises (what pins the mode against the dense one).  (Through round 5 every change of sigma refactorised in both modes.) */
    if ((st->proximal && w->gamma < st->gamma_max) ||
        (w->nb_sigma_changed > OQ_MIN(st->max_rank_update_fraction * (w->n + w->m), 0.25 * st->max_rank_update)) ||
        (w->nb_sigma_changed > 0 && !(w->sp_Lp && sparse_update_pays(w, w->nb_sigma_changed)))) w->reset_newton = 1;
    else if (w->nb_sigma_changed > 0) oq_ldlupdate_sigma_changed(w);
    return;
  }
  if ((st->proximal && w->gamma < st->gamma_max) ||
      (w->nb_sigma_changed > OQ_MIN(st->max_rank_update_fraction * (w->n + w->m), 0.25 * st->max_rank_update))) {
    w->reset_newton = 1;
  } else if (w->nb_sigma_changed == 0) {
    /* nothing */
  } else {
    oq_ldlupdate_sigma_changed(w);
  }
}

static void update_gamma(oq_workspace *w) { /* iteration.c:147-156 */
  if (w->gamma < w->settings.gamma_max) {
    oq_float prev = w->gamma;
    w->gamma = OQ_MIN(w->gamma * w->settings.gamma_upd, w->settings.gamma_max);
    w->reset_newton = 1;
    oq_vec_add_scaled(w->Qx, w->x, w->Qx, 1 / w->gamma - 1 / prev, (size_t)w->n);
  }
}

/* gershgorin_max over C = F F' with full storage (nonconvex.c:185-210 on the result of
 * cholmod aat, iteration.c:190).  C is formed densely here: C_ij = sum_t F_it F_jt, t ascending. */
static oq_float gershgorin_max_AtsigmaA(oq_workspace *w, const oq_int *fset, oq_int nf) {
  oq_int n = w->n;
  if (!w->Hbuf) w->Hbuf = zalloc((size_t)(n * n));
  oq_float *C = w->Hbuf;
  unsigned char *pat = (unsigned char *)calloc(nz1((size_t)n * (size_t)n), 1);
  memset(C, 0, (size_t)(n * n) * sizeof(oq_float));
  const oq_sparse *F = &w->At_sqrt_sigma;
  for (oq_int f = 0; f < nf; f++) {
    oq_int t = fset[f];
    for (oq_int a = F->p[t]; a < F->p[t + 1]; a++)
      for (oq_int b = F->p[t]; b < F->p[t + 1]; b++) {
        C[F->i[a] + F->i[b] * n] += F->x[a] * F->x[b];
        pat[F->i[a] + F->i[b] * n] = 1;
      }
  }
  oq_float ub = 0;
  for (oq_int j = 0; j < n; j++) {
    oq_float center = 0, radius = 0;
    for (oq_int i = 0; i < n; i++) {
      if (!pat[i + j * n]) continue;
      if (i == j) center = C[i + j * n]; else radius += OQ_ABS(C[i + j * n]);
    }
    w->temp_n[j] = center; w->neg_dphi[j] = radius;
    ub = (j == 0) ? center + radius : OQ_MAX(ub, center + radius);
  }
  free(pat);
  return ub;
}

static oq_float sparse_gershgorin(oq_workspace *w, const oq_int *fset, oq_int nf); /* sparse-storage mode, below */
static void boost_gamma(oq_workspace *w) { /* iteration.c:158-211 */
  size_t n = (size_t)w->n;
  oq_float prev = w->gamma;
  w->n_boost_gamma++;
  if (w->nb_active) {
    oq_int nb = 0;
    for (oq_int i = 0; i < w->m; i++) if (w->active[i]) w->enter[nb++] = i; /* B4 */
    if (w->kkt_mode) w->gamma = 1e10; /* iteration.c:173-176 */
    else w->gamma = OQ_MAX(w->settings.gamma_max, 1e14 / (w->sparse_mode ? sparse_gershgorin(w, w->enter, nb) : gershgorin_max_AtsigmaA(w, w->enter, nb)));
    w->gamma_maxed = 1;
  } else w->gamma = 1e12;
  if (prev != w->gamma) {
    oq_vec_add_scaled(w->Qx, w->x, w->Qx, 1.0 / w->gamma - 1.0 / prev, n);
    oq_vec_add_scaled(w->Qd, w->d, w->Qd, w->tau / w->gamma - w->tau / prev, n);
    w->reset_newton = 1;
  }
}

/* =========================================================================================
 * sparse-storage mode (see oq_workspace.sparse_mode)
 * ======================================================================================= */
static int cmp_int(const void *a, const void *b) { oq_int x = *(const oq_int *)a, y = *(const oq_int *)b; return (x > y) - (x < y); }
/* pattern of L for H = Q + A'A with ALL rows of A: struct(L_j) = struct(H_j) u U_{children} struct(L_c) \ {c}.  Natural ordering (the
 * reference's: solver_interface.c:530-540) unless the test handed over the engine's fill- / depth-reducing permutation (oq_set_perm):
 * then everything below -- pattern, values, work vectors -- lives in the permuted numbering, IP(i) = iperm[i]. */
#define IP(i) (w->sp_iperm ? w->sp_iperm[i] : (i))
#define PO(j) (w->sp_perm ? w->sp_perm[j] : (j))
static void sparse_analyze(oq_workspace *w) {
  oq_int n = w->n;
  const oq_sparse *Q = &w->Q, *A = &w->A;
  oq_sparse T; memset(&T, 0, sizeof T);
  sp_transpose(A, &T); /* n x m: column t = row t of A */
  oq_int *Lp = izalloc((size_t)n + 1), *mark = izalloc((size_t)n), *head = izalloc((size_t)n), *next = izalloc((size_t)n);
  size_t cap = (size_t)(Q->p[n] + A->p[n] + n + 16), nz = 0;
  oq_int *Li = (oq_int *)malloc(cap * sizeof(oq_int)), *col = izalloc((size_t)n);
  for (oq_int j = 0; j < n; j++) { mark[j] = -1; head[j] = -1; next[j] = -1; }
  oq_int *qp = NULL, *qi = NULL;
  if (w->sp_perm) { /* entry (r, c), r > c, of Q sits at (max, min) of (iperm[r], iperm[c]) */
    qp = izalloc((size_t)n + 1); qi = izalloc(nz1((size_t)Q->p[n]));
    for (int pass = 0; pass < 2; pass++) {
      for (oq_int c = 0; c < n; c++)
        for (oq_int k = Q->p[c]; k < Q->p[c + 1]; k++) {
          oq_int r = Q->i[k];
          if (r <= c) continue;
          oq_int a = IP(r), b2 = IP(c), lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
          if (pass == 0) qp[lo + 1]++; else qi[next[lo]++] = hi;
        }
      if (pass == 0) { for (oq_int j = 0; j < n; j++) { qp[j + 1] += qp[j]; next[j] = qp[j]; } }
    }
    for (oq_int j = 0; j < n; j++) next[j] = -1;
  }
  for (oq_int j = 0; j < n; j++) {
    oq_int cnt = 0;
    mark[j] = j;
    if (!w->sp_perm) {
      for (oq_int k = Q->p[j]; k < Q->p[j + 1]; k++) { oq_int i = Q->i[k]; if (i > j && mark[i] != j) { mark[i] = j; col[cnt++] = i; } }
    } else { /* the stored (lower) triangle of Q in the permuted numbering, bucketed by column (qp / qi below) */
      for (oq_int k = qp[j]; k < qp[j + 1]; k++) { oq_int i = qi[k]; if (mark[i] != j) { mark[i] = j; col[cnt++] = i; } }
    }
    for (oq_int p = A->p[PO(j)]; p < A->p[PO(j) + 1]; p++) {
      oq_int t = A->i[p];
      for (oq_int q = T.p[t]; q < T.p[t + 1]; q++) { oq_int i = IP(T.i[q]); if (i > j && mark[i] != j) { mark[i] = j; col[cnt++] = i; } }
    }
    for (oq_int c = head[j]; c >= 0; c = next[c])
      for (oq_int e = Lp[c]; e < Lp[c + 1]; e++) { oq_int i = Li[e]; if (i > j && mark[i] != j) { mark[i] = j; col[cnt++] = i; } }
    qsort(col, (size_t)cnt, sizeof(oq_int), cmp_int);
    if (nz + (size_t)cnt > cap) { cap = 2 * (nz + (size_t)cnt); Li = (oq_int *)realloc(Li, cap * sizeof(oq_int)); }
    for (oq_int e = 0; e < cnt; e++) Li[nz + (size_t)e] = col[e];
    nz += (size_t)cnt;
    Lp[j + 1] = (oq_int)nz;
    if (cnt) { next[j] = head[col[0]]; head[col[0]] = j; }
  }
  oq_int *Rp = izalloc((size_t)n + 1), *Rk = izalloc(nz), *Rpos = izalloc(nz), *cur = izalloc((size_t)n);
  for (size_t e = 0; e < nz; e++) Rp[Li[e] + 1]++;
  for (oq_int i = 0; i < n; i++) { Rp[i + 1] += Rp[i]; cur[i] = Rp[i]; }
  for (oq_int k = 0; k < n; k++)
    for (oq_int e = Lp[k]; e < Lp[k + 1]; e++) { oq_int dst = cur[Li[e]]++; Rk[dst] = k; Rpos[dst] = e; }
  { /* height of the elimination tree: parent(j) = first row index of column j (children have smaller indices than their parents) */
    oq_int *lev = izalloc((size_t)n), nlev = 0;
    for (oq_int j = 0; j < n; j++) {
      if (Lp[j + 1] > Lp[j]) { oq_int pj = Li[Lp[j]]; if (lev[pj] < lev[j] + 1) lev[pj] = lev[j] + 1; }
      if (lev[j] + 1 > nlev) nlev = lev[j] + 1;
    }
    w->sp_nlev = nlev;
    free(lev);
  }
  free(mark); free(head); free(next); free(col); free(cur); free(qp); free(qi); sp_free(&T);
  w->sp_Lp = Lp; w->sp_Li = Li; w->sp_Rp = Rp; w->sp_Rk = Rk; w->sp_Rpos = Rpos;
  w->sp_Lx = zalloc(nz); w->sp_D = zalloc((size_t)n); w->sp_y = zalloc((size_t)n); w->sp_b = zalloc((size_t)n);
}
static oq_int sparse_pos(const oq_workspace *w, oq_int i, oq_int j) { /* position of L(i, j), i > j */
  const oq_int *base = w->sp_Li + w->sp_Lp[j];
  const oq_int *hit = (const oq_int *)bsearch(&i, base, (size_t)(w->sp_Lp[j + 1] - w->sp_Lp[j]), sizeof(oq_int), cmp_int);
  return (oq_int)(hit - w->sp_Li);
}
/* H = tril(Q) (+ tril(F F') over the active rows) (+ beta I) on L's pattern, then oq_dense_ldl_factor's up-looking recurrence row by
 * row over the structural nonzeros only: the operations of oq_ldlcholQAtsigmaA / factor_sparse_lower, entry for entry */
static void sparse_factor(oq_workspace *w, int with_AtSA, int add_beta, oq_float beta) {
  oq_int n = w->n;
  if (!w->sp_Lp) sparse_analyze(w);
  oq_float *Lx = w->sp_Lx, *D = w->sp_D, *y = w->sp_y;
  memset(Lx, 0, (size_t)w->sp_Lp[n] * sizeof(oq_float));
  memset(D, 0, (size_t)n * sizeof(oq_float));
  if (with_AtSA) {
    const oq_sparse *F = &w->At_sqrt_sigma;
    for (oq_int t = 0; t < w->m; t++) {
      if (!w->active[t]) continue;
      for (oq_int b = F->p[t]; b < F->p[t + 1]; b++) {
        oq_float vb = F->x[b]; oq_int cb = IP(F->i[b]);
        for (oq_int a = F->p[t]; a < F->p[t + 1]; a++) {
          oq_int ra = IP(F->i[a]);
          if (ra == cb) D[cb] += F->x[a] * vb; else if (ra > cb) Lx[sparse_pos(w, ra, cb)] += F->x[a] * vb;
        }
      }
    }
  }
  const oq_sparse *Q = &w->Q;
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = Q->p[j]; k < Q->p[j + 1]; k++) {
      oq_int i = Q->i[k];
      if (i == j) D[IP(j)] = Q->x[k] + D[IP(j)];
      else if (i > j) { oq_int a = IP(i), b2 = IP(j); oq_int e = sparse_pos(w, a > b2 ? a : b2, a > b2 ? b2 : a); Lx[e] = Q->x[k] + Lx[e]; }
    }
  if (add_beta) for (oq_int j = 0; j < n; j++) D[j] += beta;
  for (oq_int k = 0; k < n; k++) { /* row k: its structural nonzeros are the columns Rk[Rp[k] .. Rp[k+1]), ascending */
    oq_float dk = D[k];
    for (oq_int r = w->sp_Rp[k]; r < w->sp_Rp[k + 1]; r++) y[w->sp_Rk[r]] = Lx[w->sp_Rpos[r]];
    for (oq_int r = w->sp_Rp[k]; r < w->sp_Rp[k + 1]; r++) {
      oq_int i = w->sp_Rk[r];
      oq_float yi = y[i];
      for (oq_int e = w->sp_Lp[i]; e < w->sp_Lp[i + 1]; e++) { oq_int rr = w->sp_Li[e]; if (rr >= k) break; y[rr] -= Lx[e] * yi; }
      oq_float lki = yi / D[i];
      dk -= lki * yi;
      Lx[w->sp_Rpos[r]] = lki;
    }
    for (oq_int r = w->sp_Rp[k]; r < w->sp_Rp[k + 1]; r++) y[w->sp_Rk[r]] = 0;
    D[k] = dk;
  }
}
/* oq_dense_ldl_rank1 on the compressed columns: the entries of a row of A form a clique of H, so the nonzeros of the vector -- those
 * it starts with and those it acquires -- lie on ONE path of the elimination tree, from the row's first column to the root; the dense
 * loop's columns off that path have w_j = 0 there (they only see d_j <- (d_j alpha) / alpha, which this form leaves out, as CHOLMOD's
 * updown does) */
static void sparse_rank1(oq_workspace *w, oq_int t, int update) {
  const oq_sparse *F = &w->At_sqrt_sigma;
  if (F->p[t + 1] <= F->p[t]) return;
  oq_float *v = w->sp_y, alpha = 1.0;
  oq_int j = w->n;
  for (oq_int k = F->p[t]; k < F->p[t + 1]; k++) { oq_int i = IP(F->i[k]); v[i] = F->x[k]; if (i < j) j = i; } /* the path starts at the row's first column */
  while (j >= 0) {
    oq_float wj = v[j], dj = w->sp_D[j], a, gam;
    if (update) { a = alpha + (wj * wj) / dj; dj *= a; gam = -wj / dj; }
    else        { a = alpha - (wj * wj) / dj; dj *= a; gam =  wj / dj; }
    dj /= alpha;
    alpha = a;
    w->sp_D[j] = dj;
    for (oq_int e = w->sp_Lp[j]; e < w->sp_Lp[j + 1]; e++) {
      oq_int i = w->sp_Li[e];
      oq_float wi = v[i] - wj * w->sp_Lx[e];
      v[i] = wi;
      w->sp_Lx[e] -= gam * wi;
    }
    v[j] = 0;
    j = (w->sp_Lp[j + 1] > w->sp_Lp[j]) ? w->sp_Li[w->sp_Lp[j]] : -1;
  }
}
/* the engine's rule (qpalm_sparse.h: sp_update_pays): walking nchange paths of at most nlev columns against refactorising n columns */
static int sparse_update_pays(const oq_workspace *w, oq_int nchange) { return w->sparse_mode == 1 && (long long)nchange * (long long)w->sp_nlev * 2 < (long long)w->n; }
static void sparse_solve(oq_workspace *w, oq_float *b_) { /* oq_dense_ldl_solve on the compressed columns (of P H P': x = P' (L D L')^-1 P b) */
  oq_int n = w->n;
  oq_float *b = b_;
  if (w->sp_perm) { b = w->sp_b; for (oq_int j = 0; j < n; j++) b[j] = b_[w->sp_perm[j]]; }
  for (oq_int j = 0; j < n; j++) { oq_float yj = b[j]; for (oq_int e = w->sp_Lp[j]; e < w->sp_Lp[j + 1]; e++) b[w->sp_Li[e]] -= w->sp_Lx[e] * yj; }
  for (oq_int j = 0; j < n; j++) b[j] /= w->sp_D[j];
  for (oq_int j = n - 1; j >= 0; j--) { oq_float xj = b[j]; for (oq_int e = w->sp_Lp[j]; e < w->sp_Lp[j + 1]; e++) xj -= w->sp_Lx[e] * b[w->sp_Li[e]]; b[j] = xj; }
  if (w->sp_perm) for (oq_int j = 0; j < n; j++) b_[w->sp_perm[j]] = b[j];
}
/* gershgorin_max_AtsigmaA without the n x n buffer: one column of C = F F' at a time in the dense work vector, summed over the touched
 * rows in ascending order (the dense routine walks i = 0 .. n-1 over its pattern map) */
static oq_float sparse_gershgorin(oq_workspace *w, const oq_int *fset, oq_int nf) {
  oq_int n = w->n;
  if (!w->sp_Lp) sparse_analyze(w);
  unsigned char *isact = (unsigned char *)calloc(nz1((size_t)w->m), 1);
  for (oq_int f = 0; f < nf; f++) isact[fset[f]] = 1;
  const oq_sparse *F = &w->At_sqrt_sigma, *A = &w->A;
  oq_float *c = w->sp_y, ub = 0;
  oq_int *touched = izalloc((size_t)n), *mark = izalloc((size_t)n);
  for (oq_int j = 0; j < n; j++) mark[j] = -1;
  for (oq_int j = 0; j < n; j++) {
    oq_int nt = 0;
    for (oq_int p = A->p[j]; p < A->p[j + 1]; p++) { /* the rows t with F_jt != 0 structurally, ascending */
      oq_int t = A->i[p];
      if (!isact[t]) continue;
      oq_float vb = 0;
      for (oq_int b = F->p[t]; b < F->p[t + 1]; b++) if (F->i[b] == j) vb = F->x[b];
      for (oq_int a = F->p[t]; a < F->p[t + 1]; a++) {
        oq_int i = F->i[a];
        if (mark[i] != j) { mark[i] = j; touched[nt++] = i; }
        c[i] += F->x[a] * vb;
      }
    }
    qsort(touched, (size_t)nt, sizeof(oq_int), cmp_int);
    oq_float center = 0, radius = 0;
    for (oq_int e = 0; e < nt; e++) { oq_int i = touched[e]; if (i == j) center = c[i]; else radius += OQ_ABS(c[i]); c[i] = 0; }
    w->temp_n[j] = center; w->neg_dphi[j] = radius;
    ub = (j == 0) ? center + radius : OQ_MAX(ub, center + radius);
  }
  free(isact); free(touched); free(mark);
  return ub;
}

/* =========================================================================================
 * solver_interface.c restated (CHOLMOD branch)
 * ======================================================================================= */
void oq_ldlchol(const oq_sparse *M, oq_workspace *w) { /* solver_interface.c:319-370 */
  if (w->sparse_mode && M == &w->Q) { sparse_factor(w, 0, (int)w->settings.proximal, 1.0 / w->gamma); return; }
  factor_sparse_lower(w, M, &w->LD, (int)w->settings.proximal, 1.0 / w->gamma);
}

void oq_ldlcholQAtsigmaA(oq_workspace *w) { /* solver_interface.c:372-405 */
  oq_int n = w->n, nb = 0;
  for (oq_int i = 0; i < w->m; i++) if (w->active[i]) w->enter[nb++] = i; /* B4: clobbers enter[] */
  if (w->sparse_mode) { sparse_factor(w, 1, (int)w->settings.proximal, 1.0 / w->gamma); return; }
  factor_alloc(&w->LD, n);
  oq_float *H = w->LD.L;
  memset(H, 0, (size_t)(n * n) * sizeof(oq_float));
  /* aat: C = F(:,f) F(:,f)', accumulated on its own with t ascending */
  const oq_sparse *F = &w->At_sqrt_sigma;
  for (oq_int f = 0; f < nb; f++) {
    oq_int t = w->enter[f];
    for (oq_int b = F->p[t]; b < F->p[t + 1]; b++) {
      oq_float vb = F->x[b]; oq_int cb = F->i[b];
      for (oq_int a = F->p[t]; a < F->p[t + 1]; a++) { oq_int ra = F->i[a]; if (ra >= cb) H[ra + cb * n] += F->x[a] * vb; }
    }
  }
  /* add: tril(Q) + tril(C) (stype re-tagged to Q's, :391-392) */
  const oq_sparse *Q = &w->Q;
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = Q->p[j]; k < Q->p[j + 1]; k++) { oq_int i = Q->i[k]; if (i >= j) H[i + j * n] = Q->x[k] + H[i + j * n]; }
  if (w->settings.proximal) { oq_float beta = 1.0 / w->gamma; for (oq_int j = 0; j < n; j++) H[j + j * n] += beta; }
  oq_dense_ldl_factor(n, H, n, w->LD.D);
  w->LD.valid = 1;
}

/* The same update for up to OQ_UPDOWN_BLOCK ranks in ONE pass over L, as cholmod_updown carries up to eight ranks per pass: column by column, rank
 * after rank inside a column.  Every entry of L, D and of the vectors receives exactly the operations of the rank-1 routine above in the
 * same order (rank r meets column j after ranks < r have met it and after it has itself met the columns < j), so the result is
 * BIT-IDENTICAL to applying the ranks one after the other -- tests/test_oracle_golden.py checks that -- while L is streamed once per
 * block instead of once per rank.  Used by bench.py's cpu_baseline leg (oq_set_scalar "updown_block"); the parity tests keep the scalar form. */
#define OQ_UPDOWN_BLOCK 8
static void dense_ldl_rankk(oq_int n, oq_float *L, oq_int ld, oq_float *D, oq_float *W, oq_int k, int update) {
  oq_float alpha[OQ_UPDOWN_BLOCK], wj[OQ_UPDOWN_BLOCK], gam[OQ_UPDOWN_BLOCK];
  oq_int j0 = n;
  for (oq_int r = 0; r < k; r++) {
    alpha[r] = 1.0;
    oq_int f = 0;
    while (f < n && W[r * n + f] == 0.0) f++;
    if (f < j0) j0 = f;
  }
  for (oq_int j = j0; j < n; j++) {
    oq_float dj = D[j];
    for (oq_int r = 0; r < k; r++) {
      oq_float a;
      wj[r] = W[r * n + j];
      if (wj[r] == 0.0 && alpha[r] == 1.0) { gam[r] = 0.0; continue; } /* above the rank's first nonzero: the rank-1 routine has not started yet (exact no-op) */
      if (update) { a = alpha[r] + (wj[r] * wj[r]) / dj; dj *= a; gam[r] = -wj[r] / dj; }
      else        { a = alpha[r] - (wj[r] * wj[r]) / dj; dj *= a; gam[r] =  wj[r] / dj; }
      dj /= alpha[r];
      alpha[r] = a;
    }
    D[j] = dj;
    oq_float *Lj = L + j * ld;
    /* kp streams + the column, independent across i: the compiler vectorises over i (same operations per entry, same order).  k is padded up to
     * 1, 2, 4 or 8 streams with zero vectors (w_j = 0, gamma = 0: exact no-ops on finite entries). */
#define OQ_RK_STEP(r) t = wp[r][i] - wj[r] * l; wp[r][i] = t; l -= gam[r] * t;
#define OQ_RK_LOOP(BODY) { _Pragma("GCC ivdep") for (oq_int i = j + 1; i < n; i++) { oq_float l = l_[i], t; BODY l_[i] = l; } }
    {
      oq_float *restrict l_ = Lj;
      oq_float *restrict wp[OQ_UPDOWN_BLOCK];
      for (int r = 0; r < OQ_UPDOWN_BLOCK; r++) wp[r] = W + (size_t)r * (size_t)n; /* (rows k .. kp-1 of W are zero: dense_ldl_rankk's caller clears kp rows) */
      for (oq_int r = k; r < OQ_UPDOWN_BLOCK; r++) { wj[r] = 0.0; gam[r] = 0.0; }
      if (k == 1) OQ_RK_LOOP(OQ_RK_STEP(0))
      else if (k == 2) OQ_RK_LOOP(OQ_RK_STEP(0) OQ_RK_STEP(1))
      else if (k <= 4) OQ_RK_LOOP(OQ_RK_STEP(0) OQ_RK_STEP(1) OQ_RK_STEP(2) OQ_RK_STEP(3))
      else OQ_RK_LOOP(OQ_RK_STEP(0) OQ_RK_STEP(1) OQ_RK_STEP(2) OQ_RK_STEP(3) OQ_RK_STEP(4) OQ_RK_STEP(5) OQ_RK_STEP(6) OQ_RK_STEP(7))
    }
#undef OQ_RK_STEP
#undef OQ_RK_LOOP
  }
}
static void updown_columns(oq_workspace *w, const oq_int *cols, oq_int ncols, int update) {
  /* submatrix(At_sqrt_sigma, :, cols) then updown(update, C, L) (solver_interface.c:415-421,433-439) */
  oq_int n = w->n;
  if (w->sparse_mode) { w->n_updown_calls++; for (oq_int c = 0; c < ncols; c++) { sparse_rank1(w, cols[c], update); w->n_rank1++; } return; }
  const oq_sparse *F = &w->At_sqrt_sigma;
  w->n_updown_calls++;
  if (w->updown_block > 1) {
    if (!w->wblock) w->wblock = (oq_float *)malloc((size_t)OQ_UPDOWN_BLOCK * (size_t)(n ? n : 1) * sizeof(oq_float));
    for (oq_int c0 = 0; c0 < ncols; c0 += OQ_UPDOWN_BLOCK) {
      const oq_int k = (ncols - c0 < OQ_UPDOWN_BLOCK) ? (ncols - c0) : OQ_UPDOWN_BLOCK;
      const oq_int kp = (k <= 2) ? k : ((k <= 4) ? 4 : OQ_UPDOWN_BLOCK); /* streams the kernel runs (zero vectors beyond k) */
      memset(w->wblock, 0, (size_t)kp * (size_t)n * sizeof(oq_float));
      for (oq_int r = 0; r < k; r++) {
        const oq_int t = cols[c0 + r];
        for (oq_int e = F->p[t]; e < F->p[t + 1]; e++) w->wblock[r * n + F->i[e]] = F->x[e];
      }
      dense_ldl_rankk(n, w->LD.L, n, w->LD.D, w->wblock, k, update);
      w->n_rank1 += k;
    }
    return;
  }
  for (oq_int c = 0; c < ncols; c++) {
    oq_int t = cols[c];
    memset(w->wbuf, 0, (size_t)n * sizeof(oq_float));
    for (oq_int k = F->p[t]; k < F->p[t + 1]; k++) w->wbuf[F->i[k]] = F->x[k];
    oq_dense_ldl_rank1(n, w->LD.L, n, w->LD.D, w->wbuf, update);
    w->n_rank1++;
  }
}
void oq_ldlupdate_entering_constraints(oq_workspace *w) { updown_columns(w, w->enter, w->nb_enter, 1); }   /* :407-423 */
void oq_ldldowndate_leaving_constraints(oq_workspace *w) { updown_columns(w, w->leave, w->nb_leave, 0); } /* :425-441 */

void oq_ldlupdate_sigma_changed(oq_workspace *w) { /* solver_interface.c:443-503 */
  oq_int *sigma_changed = w->enter;
  oq_float *At_scalex = w->At_scale;
  for (oq_int k = 0; k < w->nb_sigma_changed; k++) {
    oq_int row = sigma_changed[k];
    At_scalex[row] = At_scalex[row] * At_scalex[row];
    At_scalex[row] = sqrt(1 - 1 / At_scalex[row]); /* FACTORIZE_SCHUR */
  }
  sp_scale_col(&w->At_sqrt_sigma, w->At_scale);
  updown_columns(w, sigma_changed, w->nb_sigma_changed, 1);
  oq_int zeroed = 0;
  for (oq_int k = 0; k < w->m; k++) { if (At_scalex[k] == 0.0) zeroed++; At_scalex[k] = 1.0 / At_scalex[k]; }
  sp_scale_col(&w->At_sqrt_sigma, w->At_scale);
  if (zeroed) {
    /* NOT in the reference: a sigma that grew by one unit in the last place has sqrt(mult_factor) == 1, its update vector
     * sqrt(1 - 1/1) A_k is exactly zero (a no-op update) and the reference's CHOLMOD branch then scales the zeroed column back by
     * 1/0: 0 * inf = NaN in At_sqrt_sigma (solver_interface.c:498-502; the LADEL branch never scales At).  There is no reference
     * behaviour to restate on such a case; the column is rebuilt from A' and sqrt(sigma), as the engine does (qpalm_iter.h:
     * dev_update_sigma_post).  Found by the engine's fresh-seed fuzz campaign of round 4 (seed 204 case 155). */
    oq_sparse T;
    memset(&T, 0, sizeof T);
    sp_transpose(&w->A, &T);
    for (oq_int k = 0; k < w->m; k++)
      if (isinf(At_scalex[k]))
        for (oq_int e = T.p[k]; e < T.p[k + 1]; e++) w->At_sqrt_sigma.x[e] = T.x[e] * w->sqrt_sigma[k];
    sp_free(&T);
  }
}

void oq_ldlsolveLD_neg_dphi(oq_workspace *w) { /* solver_interface.c:505-519 */
  size_t n = (size_t)w->n;
  vec_cp(w->dphi, w->neg_dphi, n);
  oq_vec_self_mult_scalar(w->neg_dphi, -1, n);
  vec_cp(w->neg_dphi, w->d, n);
  if (w->sparse_mode) sparse_solve(w, w->d); else
  oq_dense_ldl_solve(w->n, w->LD.L, w->n, w->LD.D, w->d);
  w->n_solve++;
}

/* =========================================================================================
 * newton.c restated (SCHUR branch, newton.c:96-120)
 * ======================================================================================= */
void oq_set_active_constraints(oq_workspace *w) { /* newton.c:122-132 */
  w->nb_active = 0;
  for (oq_int i = 0; i < w->m; i++) {
    if ((w->Axys[i] <= w->bmin[i]) || (w->Axys[i] >= w->bmax[i])) { w->active[i] = 1; w->nb_active++; }
    else w->active[i] = 0;
  }
}
void oq_set_entering_leaving_constraints(oq_workspace *w) { /* newton.c:134-149 */
  oq_int ne = 0, nl = 0;
  for (oq_int i = 0; i < w->m; i++) {
    if (w->active[i] && !w->active_old[i]) w->enter[ne++] = i;
    if (!w->active[i] && w->active_old[i]) w->leave[nl++] = i;
  }
  w->nb_enter = ne; w->nb_leave = nl;
}

/* =========================================================================================
 * KKT path: qpalm_form_kkt / qpalm_reform_kkt, kkt_update_entering / leaving_constraints, kkt_solve
 * (src/solver_interface.c:119-247) and the KKT branch of newton_set_direction with its iterative refinement
 * (src/newton.c:22-95).  LADEL (github.com/Benny44/LADEL, branch master, SHA unpinned, .gitmodules:9-12) is absent from
 * the reference tree; its arithmetic is restated from the published algorithms on a DENSE (n+m) x (n+m) lower
 * triangle in natural order (LADEL's default AMD ordering only permutes the same quasi-definite system):
 *   - ladel_factorize(_advanced / _with_prior_basis)_with_diag : LDL' without pivoting of K + diag(1/gamma on the first n);
 *   - ladel_row_add / ladel_row_del : Davis & Hager, "Row modifications of a sparse Cholesky factorization"
 *     (SIAM J. Matrix Anal. Appl. 2005): with K = [K11 k12 K31'; k12' k22 k32'; K31 k32 K33] and row p = n+k,
 *       add:  L11 z = k12, l12 = D11^{-1} z, d22 = k22 - l12' z, l32 = (k32 - L31 z) / d22,
 *             L33 D33 L33' <- L33 D33 L33' - l32 d22 l32'   (a rank-1 update: d22 < 0 for a constraint row);
 *       del:  L33 D33 L33' <- L33 D33 L33' + l32 d22 l32',  l12 = 0, l32 = 0, d22 = 1;
 *   - ladel_dense_solve : L y = b, y /= D, L' x = y.
 * "parity unpinned": the reference's tests pin this path only through end-to-end solutions (tests/src/test_basic_qp.c
 * etc. run under the LADEL build, which CI uses: travis/buildTest.sh:48).
 * ======================================================================================= */
static void kkt_form_and_factor(oq_workspace *w) {
  const oq_int n = w->n, m = w->m, np = n + m;
  if (w->At.p) sp_free(&w->At);
  sp_transpose(&w->A, &w->At); /* n x m: column k = row k of the scaled A */
  factor_alloc(&w->LDK, np);
  oq_float *K = w->LDK.L;
  memset(K, 0, (size_t)np * (size_t)np * sizeof(oq_float));
  const oq_sparse *Q = &w->Q;
  for (oq_int j = 0; j < n; j++)
    for (oq_int k = Q->p[j]; k < Q->p[j + 1]; k++) { const oq_int i = Q->i[k]; if (i >= j) K[i + j * np] += Q->x[k]; }
  if (w->settings.proximal) for (oq_int j = 0; j < n; j++) K[j + j * np] += 1.0 / w->gamma; /* ladel_diag, newton.c:27-30 */
  for (oq_int k = 0; k < m; k++) {
    const oq_int p = n + k, cnt = w->At.p[k + 1] - w->At.p[k];
    if (w->active[k]) {
      w->kkt_state[k] = 1;
      for (oq_int e = w->At.p[k]; e < w->At.p[k + 1]; e++) K[p + w->At.i[e] * np] = w->At.x[e];
      K[p + p * np] = cnt ? -w->sigma_inv[k] : 1.0; /* empty rows of A get a unit diagonal (:169) */
    } else { w->kkt_state[k] = 0; K[p + p * np] = 1.0; }
  }
  oq_dense_ldl_factor(np, K, np, w->LDK.D);
  w->LDK.valid = 1;
}

static void kkt_row_add(oq_workspace *w, oq_int k) { /* ladel_row_add(LD, sym, n+k, kkt, n+k, -sigma_inv[k], c), :209-217 */
  const oq_int n = w->n, np = w->n + w->m, p = n + k;
  oq_float *L = w->LDK.L, *D = w->LDK.D, *z = w->kkt_tmp;
  for (oq_int j = 0; j < np; j++) z[j] = 0.0;
  for (oq_int e = w->At.p[k]; e < w->At.p[k + 1]; e++) z[w->At.i[e]] = w->At.x[e];
  for (oq_int j = 0; j < p; j++) { /* L11 z = k12, column oriented */
    const oq_float zj = z[j];
    if (zj == 0.0) continue;
    const oq_float *Lj = L + j * np;
    for (oq_int i = j + 1; i < p; i++) z[i] -= Lj[i] * zj;
  }
  oq_float d22 = -w->sigma_inv[k];
  for (oq_int j = 0; j < p; j++) { const oq_float l = z[j] / D[j]; d22 -= l * z[j]; L[p + j * np] = l; }
  oq_float *wv = w->rhs_kkt; /* scratch: the caller overwrites rhs_kkt afterwards */
  const oq_float sq = sqrt(OQ_ABS(d22));
  for (oq_int i = p + 1; i < np; i++) { /* l32 = -(L31 z) / d22  (k32 = 0: constraint rows do not couple) */
    oq_float acc = 0.0;
    for (oq_int j = 0; j < p; j++) acc += L[i + j * np] * z[j];
    const oq_float l = -acc / d22;
    L[i + p * np] = l;
    wv[i] = sq * l;
  }
  D[p] = d22;
  if (p + 1 < np) oq_dense_ldl_rank1(np - p - 1, L + (p + 1) + (p + 1) * np, np, D + p + 1, wv + p + 1, d22 < 0);
  w->kkt_state[k] = 1;
  w->n_row_add++;
}

static void kkt_row_del(oq_workspace *w, oq_int k) { /* ladel_row_del(LD, sym, n+k, c), :229 */
  const oq_int np = w->n + w->m, p = w->n + k;
  oq_float *L = w->LDK.L, *D = w->LDK.D, *wv = w->rhs_kkt;
  const oq_float d = D[p], sq = sqrt(OQ_ABS(d));
  for (oq_int i = p + 1; i < np; i++) { wv[i] = sq * L[i + p * np]; L[i + p * np] = 0.0; }
  for (oq_int j = 0; j < p; j++) L[p + j * np] = 0.0;
  D[p] = 1.0;
  if (p + 1 < np) oq_dense_ldl_rank1(np - p - 1, L + (p + 1) + (p + 1) * np, np, D + p + 1, wv + p + 1, d > 0);
  w->kkt_state[k] = 2;
  w->n_row_del++;
}

/* mat_vec(kkt, x, y) (+ the proximal diagonal added by hand, newton.c:58-59): kkt as the reference stores it, i.e.
 * columns truncated by nz[]: state 0 -> unit diagonal, 1 -> [A(k,:)'; -1/sigma_k], 2 -> diagonal -1/sigma_k only */
static void kkt_matvec(oq_workspace *w, const oq_float *x, oq_float *y) {
  const oq_int n = w->n, m = w->m;
  oq_mat_vec(&w->Q, x, y);
  if (w->settings.proximal) for (oq_int j = 0; j < n; j++) y[j] = 1 * y[j] + (1.0 / w->gamma) * x[j];
  for (oq_int k = 0; k < m; k++) {
    const oq_float xk = x[n + k];
    if (w->kkt_state[k] == 1) {
      oq_float acc = 0.0;
      for (oq_int e = w->At.p[k]; e < w->At.p[k + 1]; e++) { acc += w->At.x[e] * x[w->At.i[e]]; y[w->At.i[e]] += w->At.x[e] * xk; }
      const oq_int cnt = w->At.p[k + 1] - w->At.p[k];
      y[n + k] = acc + (cnt ? -w->sigma_inv[k] : 1.0) * xk;
    } else if (w->kkt_state[k] == 2) y[n + k] = -w->sigma_inv[k] * xk;
    else y[n + k] = xk;
  }
}

static void kkt_solve_refine(oq_workspace *w) { /* newton.c:47-95: kkt_solve + at most three refinement passes */
  const oq_int n = w->n, m = w->m, np = n + m;
  /* kkt_solve, solver_interface.c:238-247 */
  for (oq_int j = 0; j < n; j++) w->rhs_kkt[j] = w->dphi[j] * -1;
  for (oq_int k = 0; k < m; k++) w->rhs_kkt[n + k] = 0;
  vec_cp(w->rhs_kkt, w->sol_kkt, (size_t)np);
  oq_dense_ldl_solve(np, w->LDK.L, np, w->LDK.D, w->sol_kkt);
  vec_cp(w->sol_kkt, w->d, (size_t)n);
  w->n_solve++;
  /* iterative refinement, newton.c:57-90 (constants.h:101-103) */
  kkt_matvec(w, w->sol_kkt, w->rhs_kkt);
  oq_vec_self_mult_scalar(w->rhs_kkt, -1, (size_t)np);
  const oq_float ref_norm = OQ_MAX(oq_vec_norm_inf(w->rhs_kkt, (size_t)np), oq_vec_norm_inf(w->dphi, (size_t)n));
  oq_vec_mult_add_scaled(w->rhs_kkt, w->dphi, 1, -1, (size_t)n);
  oq_float res = oq_vec_norm_inf(w->rhs_kkt, (size_t)np);
  oq_int k = 0;
  while (k < 3 && res > OQ_MAX(1e-10 * ref_norm, 1e-12)) {
    k++;
    vec_cp(w->sol_kkt, w->temp_n, (size_t)n);
    vec_cp(w->sol_kkt + n, w->temp_m, (size_t)m);
    vec_cp(w->rhs_kkt, w->sol_kkt, (size_t)np);
    oq_dense_ldl_solve(np, w->LDK.L, np, w->LDK.D, w->sol_kkt);
    oq_vec_add_scaled(w->sol_kkt, w->d, w->d, 1, (size_t)n);
    oq_vec_mult_add_scaled(w->sol_kkt, w->temp_n, 1, 1, (size_t)n);
    oq_vec_mult_add_scaled(w->sol_kkt + n, w->temp_m, 1, 1, (size_t)m);
    kkt_matvec(w, w->sol_kkt, w->rhs_kkt);
    oq_vec_self_mult_scalar(w->rhs_kkt, -1, (size_t)np);
    oq_vec_mult_add_scaled(w->rhs_kkt, w->dphi, 1, -1, (size_t)n);
    res = oq_vec_norm_inf(w->rhs_kkt, (size_t)np);
    w->n_refine++;
  }
}
static void kkt_newton_direction(oq_workspace *w) { /* newton.c:22-95 */
  const oq_settings *st = &w->settings;
  if (w->first_factorization) {
    kkt_form_and_factor(w); w->first_factorization = 0; w->n_refactor++; w->last_fact = 1;
  } else if (w->reset_newton ||
             (w->nb_enter + w->nb_leave) > OQ_MIN(st->max_rank_update_fraction * (w->n + w->m), st->max_rank_update)) {
    kkt_form_and_factor(w); w->n_refactor++; w->last_fact = 1;
  } else {
    w->last_fact = 0;
    for (oq_int e = 0; e < w->nb_enter; e++) { kkt_row_add(w, w->enter[e]); w->last_fact = 2; }
    for (oq_int e = 0; e < w->nb_leave; e++) { kkt_row_del(w, w->leave[e]); w->last_fact = 2; }
  }
  kkt_solve_refine(w);
}

void oq_newton_set_direction(oq_workspace *w) { /* newton.c:17-120 */
  const oq_settings *st = &w->settings;
  oq_set_active_constraints(w);
  oq_set_entering_leaving_constraints(w);
  if (w->kkt_mode) {
    kkt_newton_direction(w);
    ivec_cp(w->active, w->active_old, (size_t)w->m);
    w->reset_newton = 0;
    return;
  }
  if ((w->reset_newton && w->nb_active) ||
      (w->nb_enter + w->nb_leave) > OQ_MIN(st->max_rank_update_fraction * (w->n + w->m), st->max_rank_update) ||
      (w->sparse_mode && w->nb_active && (w->nb_enter + w->nb_leave) > 0 && !(w->sp_Lp && sparse_update_pays(w, w->nb_enter + w->nb_leave)))) {
    /* (sparse mode: a changed active set refactorises unless the elimination tree is bushy enough for path updates to be cheaper) */
    oq_ldlcholQAtsigmaA(w); w->n_refactor++; w->last_fact = 1;
  } else if (w->nb_active) {
    w->last_fact = 0;
    if (w->nb_enter) { oq_ldlupdate_entering_constraints(w); w->last_fact = 2; }
    if (w->nb_leave) { oq_ldldowndate_leaving_constraints(w); w->last_fact = 2; }
  } else {
    oq_ldlchol(&w->Q, w); w->n_factor_Q++; w->last_fact = 3; /* B7 */
  }
  oq_ldlsolveLD_neg_dphi(w);
  ivec_cp(w->active, w->active_old, (size_t)w->m);
  w->reset_newton = 0;
}

/* =========================================================================================
 * linesearch.c restated
 * ======================================================================================= */
static int compare_elems(const void *a, const void *b) { /* linesearch.c:158-166 */
  oq_float f = ((const oq_array_element *)a)->x, s = ((const oq_array_element *)b)->x;
  if (f > s) return 1;
  if (f < s) return -1;
  return 0;
}
static oq_float vec_prod_ind(const oq_float *a, const oq_float *b, const oq_int *L, size_t n) { /* :145-156 */
  oq_float prod = 0.0;
  for (size_t i = 0; i < n; i++) if (L[i]) prod += a[i] * b[i];
  return prod;
}
oq_float oq_exact_linesearch(oq_workspace *w) { /* linesearch.c:14-120 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_mat_vec(&w->Q, w->d, w->Qd);
  if (w->settings.proximal) oq_vec_add_scaled(w->Qd, w->d, w->Qd, 1 / w->gamma, n);
  oq_mat_vec(&w->A, w->d, w->Ad);
  w->eta = oq_vec_prod(w->d, w->Qd, n);
  w->beta = oq_vec_prod(w->d, w->df, n);
  oq_vec_ew_prod(w->sqrt_sigma, w->Ad, w->temp_m, m);
  vec_cp(w->temp_m, w->delta + m, m);
  oq_vec_self_mult_scalar(w->temp_m, -1, m);
  vec_cp(w->temp_m, w->delta, m);
  oq_vec_add_scaled(w->Ax, w->bmin, w->temp_m, -1, m);
  oq_vec_ew_prod(w->sigma, w->temp_m, w->temp_m, m);
  oq_vec_add_scaled(w->y, w->temp_m, w->temp_m, 1, m);
  oq_vec_ew_div(w->temp_m, w->sqrt_sigma, w->temp_m, m);
  vec_cp(w->temp_m, w->alpha, m);
  oq_vec_add_scaled(w->bmax, w->Ax, w->temp_m, -1, m);
  oq_vec_ew_prod(w->sigma, w->temp_m, w->temp_m, m);
  oq_vec_add_scaled(w->temp_m, w->y, w->temp_m, -1, m);
  oq_vec_ew_div(w->temp_m, w->sqrt_sigma, w->temp_m, m);
  vec_cp(w->temp_m, w->alpha + m, m);
  oq_vec_ew_div(w->alpha, w->delta, w->temp_2m, m * 2);
  for (size_t i = 0; i < 2 * m; i++) { w->s[i].x = w->temp_2m[i]; w->s[i].i = i; }
  size_t nL = 0;
  for (size_t i = 0; i < m * 2; i++) {
    if (w->temp_2m[i] > 0) { w->index_L[i] = 1; nL++; } else w->index_L[i] = 0;
  }
  { size_t nb = 0; for (size_t i = 0; i < 2 * m; i++) if (w->index_L[i]) w->s[nb++] = w->s[i]; }
  for (size_t i = 0; i < m * 2; i++) w->index_P[i] = (w->delta[i] > 0) ? 1 : 0;
  for (size_t i = 0; i < m * 2; i++) w->index_J[i] = ((w->index_P[i] + w->index_L[i]) == 1) ? 1 : 0;
  oq_float a = w->eta + vec_prod_ind(w->delta, w->delta, w->index_J, m * 2);
  oq_float b = w->beta - vec_prod_ind(w->delta, w->alpha, w->index_J, m * 2);
  qsort(w->s, nL, sizeof(oq_array_element), compare_elems);
  if (nL == 0 || a * w->s[0].x + b > 0) return -b / a;
  size_t i = 0, iz;
  while (i < nL - 1) {
    iz = w->s[i].i;
    if (w->index_P[iz]) { a = a + w->delta[iz] * w->delta[iz]; b = b - w->delta[iz] * w->alpha[iz]; }
    else                { a = a - w->delta[iz] * w->delta[iz]; b = b + w->delta[iz] * w->alpha[iz]; }
    i++;
    if (a * w->s[i].x + b > 0) return -b / a;
  }
  iz = w->s[i].i;
  if (w->index_P[iz]) { a = a + w->delta[iz] * w->delta[iz]; b = b - w->delta[iz] * w->alpha[iz]; }
  else                { a = a - w->delta[iz] * w->delta[iz]; b = b + w->delta[iz] * w->alpha[iz]; }
  return -b / a;
}

void oq_update_primal_iterate(oq_workspace *w) { /* iteration.c:213-229 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_newton_set_direction(w);
  w->tau = oq_exact_linesearch(w);
  if (w->guard && !w->guard_spent && (w->last_fact == 0 || w->last_fact == 2) && (!isfinite(w->eta) || !isfinite(w->beta))) {
    /* NOT in the reference (oq_workspace::guard): the step is taken again with a fresh factorisation; the active sets and the
     * enter / leave counts of the step stay as they are (they are read again by the loop, B4) */
    w->n_guard_refactor++;
    if (w->kkt_mode) { kkt_form_and_factor(w); w->n_refactor++; w->last_fact = 1; kkt_solve_refine(w); }
    else { oq_ldlcholQAtsigmaA(w); w->n_refactor++; w->last_fact = 1; oq_ldlsolveLD_neg_dphi(w); }
    w->tau = oq_exact_linesearch(w);
    if (!isfinite(w->eta) || !isfinite(w->beta)) w->guard_spent = 1; /* the fresh factorisation gives a non-finite direction too (a NaN right-hand side, a singular H): nothing to repair, the guard is off for the rest of this solve */
  }
  vec_cp(w->x, w->x_prev, n);
  vec_cp(w->dphi, w->dphi_prev, n);
  oq_vec_add_scaled(w->x, w->d, w->x, w->tau, n);
  oq_vec_self_mult_scalar(w->Qd, w->tau, n);
  oq_vec_self_mult_scalar(w->Ad, w->tau, m);
  oq_vec_add_scaled(w->Qx, w->Qd, w->Qx, 1, n);
  oq_vec_add_scaled(w->Ax, w->Ad, w->Ax, 1, m);
}

/* =========================================================================================
 * termination.c restated
 * ======================================================================================= */
static void store_solution(oq_workspace *w) { /* termination.c:242-252 (B12: yh rescaled in place) */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  if (w->has_scaling) {
    oq_vec_ew_prod(w->x, w->D, w->sol_x, n);
    oq_vec_self_mult_scalar(w->yh, w->sc_cinv, m);
    oq_vec_ew_prod(w->yh, w->E, w->sol_y, m);
  } else { vec_cp(w->x, w->sol_x, n); vec_cp(w->yh, w->sol_y, m); }
  w->info.objective = compute_objective(w);
}
static void calculate_residuals_and_tolerances(oq_workspace *w) { /* termination.c:44-129 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  const oq_settings *st = &w->settings;
  if (w->has_scaling) { oq_vec_ew_prod(w->Einv, w->pri_res, w->temp_m, m); w->info.pri_res_norm = oq_vec_norm_inf(w->temp_m, m); }
  else w->info.pri_res_norm = oq_vec_norm_inf(w->pri_res, m);
  if (w->has_scaling) {
    if (st->proximal) {
      oq_vec_add_scaled(w->x, w->x0, w->xx0, -1, n);
      oq_vec_add_scaled(w->dphi, w->xx0, w->temp_n, -1 / w->gamma, n);
      oq_vec_ew_prod(w->Dinv, w->temp_n, w->temp_n, n);
      w->info.dua_res_norm = oq_vec_norm_inf(w->temp_n, n);
      oq_vec_ew_prod(w->Dinv, w->dphi, w->temp_n, n);
      w->info.dua2_res_norm = oq_vec_norm_inf(w->temp_n, n);
    } else {
      oq_vec_ew_prod(w->Dinv, w->dphi, w->temp_n, n);
      w->info.dua_res_norm = oq_vec_norm_inf(w->temp_n, n);
      w->info.dua2_res_norm = w->info.dua_res_norm;
    }
    w->info.dua_res_norm *= w->sc_cinv;
    w->info.dua2_res_norm *= w->sc_cinv;
  } else {
    if (st->proximal) {
      oq_vec_add_scaled(w->x, w->x0, w->xx0, -1, n);
      oq_vec_add_scaled(w->dphi, w->xx0, w->temp_n, -1 / w->gamma, n);
      w->info.dua_res_norm = oq_vec_norm_inf(w->temp_n, n);
      w->info.dua2_res_norm = oq_vec_norm_inf(w->dphi, n);
    } else { w->info.dua_res_norm = oq_vec_norm_inf(w->dphi, n); w->info.dua2_res_norm = w->info.dua_res_norm; }
  }
  if (w->has_scaling) { /* B1: only the Einv.*Ax half is normed */
    oq_vec_ew_prod(w->Einv, w->Ax, w->temp_2m, m);
    oq_vec_ew_prod(w->Einv, w->z, w->temp_2m + m, m);
    w->eps_pri = st->eps_abs + st->eps_rel * oq_vec_norm_inf(w->temp_2m, m);
  } else w->eps_pri = st->eps_abs + st->eps_rel * OQ_MAX(oq_vec_norm_inf(w->Ax, m), oq_vec_norm_inf(w->z, m));
  oq_float nQx, nq, nAtyh, mx;
  if (w->has_scaling) {
    oq_vec_ew_prod(w->Dinv, w->Qx, w->temp_n, n); nQx = oq_vec_norm_inf(w->temp_n, n);
    oq_vec_ew_prod(w->Dinv, w->q, w->temp_n, n); nq = oq_vec_norm_inf(w->temp_n, n);
    oq_vec_ew_prod(w->Dinv, w->Atyh, w->temp_n, n); nAtyh = oq_vec_norm_inf(w->temp_n, n);
  } else { nQx = oq_vec_norm_inf(w->Qx, n); nq = oq_vec_norm_inf(w->q, n); nAtyh = oq_vec_norm_inf(w->Atyh, n); }
  mx = OQ_MAX(nQx, OQ_MAX(nq, nAtyh));
  if (w->has_scaling) mx *= w->sc_cinv;
  w->eps_dua = st->eps_abs + st->eps_rel * mx;
  w->eps_dua_in = w->eps_abs_in + w->eps_rel_in * mx;
}
static int is_primal_infeasible(oq_workspace *w) { /* termination.c:136-182 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  oq_float eps_pinf_norm_Edy;
  oq_vec_add_scaled(w->yh, w->y, w->delta_y, -1, m);
  if (w->has_scaling) { oq_vec_ew_prod(w->E, w->delta_y, w->temp_m, m); eps_pinf_norm_Edy = w->settings.eps_prim_inf * oq_vec_norm_inf(w->temp_m, m); }
  else eps_pinf_norm_Edy = w->settings.eps_prim_inf * oq_vec_norm_inf(w->delta_y, m);
  if (eps_pinf_norm_Edy == 0) return 0;
  oq_vec_add_scaled(w->Atyh, w->Aty, w->Atdelta_y, -1, n);
  if (w->has_scaling) oq_vec_ew_prod(w->Dinv, w->Atdelta_y, w->Atdelta_y, n);
  oq_float oob = 0;
  if (w->has_scaling) {
    for (size_t i = 0; i < m; i++) {
      oob += (w->bmax[i] < w->E[i] * OQ_INFTY) ? w->bmax[i] * OQ_MAX(w->delta_y[i], 0) : 0;
      oob += (w->bmin[i] > -w->E[i] * OQ_INFTY) ? w->bmin[i] * OQ_MIN(w->delta_y[i], 0) : 0;
    }
  } else {
    for (size_t i = 0; i < m; i++) {
      oob += (w->bmax[i] < OQ_INFTY) ? w->bmax[i] * OQ_MAX(w->delta_y[i], 0) : 0;
      oob += (w->bmin[i] > -OQ_INFTY) ? w->bmin[i] * OQ_MIN(w->delta_y[i], 0) : 0;
    }
  }
  return (oq_vec_norm_inf(w->Atdelta_y, n) <= eps_pinf_norm_Edy) && (oob <= -eps_pinf_norm_Edy);
}
static int is_dual_infeasible(oq_workspace *w) { /* termination.c:184-240 */
  size_t n = (size_t)w->n, m = (size_t)w->m;
  const oq_settings *st = &w->settings;
  oq_float eps_dinf_norm_Ddx, dxQdx, dxdx;
  oq_vec_add_scaled(w->x, w->x_prev, w->delta_x, -1, n);
  if (w->has_scaling) {
    oq_vec_ew_prod(w->D, w->delta_x, w->temp_n, n);
    eps_dinf_norm_Ddx = st->eps_dual_inf * oq_vec_norm_inf(w->temp_n, n);
    dxdx = oq_vec_prod(w->temp_n, w->temp_n, n);
  } else { eps_dinf_norm_Ddx = st->eps_dual_inf * oq_vec_norm_inf(w->delta_x, n); dxdx = oq_vec_prod(w->delta_x, w->delta_x, n); }
  if (eps_dinf_norm_Ddx == 0) return 0;
  if (w->has_scaling) {
    oq_vec_ew_prod(w->Einv, w->Ad, w->Adelta_x, m);
    for (size_t k = 0; k < m; k++)
      if ((w->bmax[k] < w->E[k] * OQ_INFTY && w->Adelta_x[k] >= eps_dinf_norm_Ddx) ||
          (w->bmin[k] > -w->E[k] * OQ_INFTY && w->Adelta_x[k] <= -eps_dinf_norm_Ddx)) return 0;
  } else {
    for (size_t k = 0; k < m; k++)
      if ((w->bmax[k] < OQ_INFTY && w->Ad[k] >= eps_dinf_norm_Ddx) || (w->bmin[k] > -OQ_INFTY && w->Ad[k] <= -eps_dinf_norm_Ddx)) return 0;
  }
  if (st->proximal) { oq_vec_add_scaled(w->Qd, w->d, w->temp_n, -w->tau / w->gamma, n); dxQdx = oq_vec_prod(w->delta_x, w->temp_n, n); }
  else dxQdx = oq_vec_prod(w->Qd, w->delta_x, n);
  if (w->has_scaling)
    return (dxQdx <= -w->sc_c * st->eps_dual_inf * st->eps_dual_inf * dxdx) ||
           ((dxQdx <= w->sc_c * st->eps_dual_inf * st->eps_dual_inf * dxdx) && (oq_vec_prod(w->q, w->delta_x, n) <= -w->sc_c * eps_dinf_norm_Ddx));
  return (dxQdx <= -st->eps_dual_inf * st->eps_dual_inf * dxdx) ||
         ((dxQdx <= st->eps_dual_inf * st->eps_dual_inf * dxdx) && (oq_vec_prod(w->q, w->delta_x, n) <= -eps_dinf_norm_Ddx));
}
oq_int oq_check_termination(oq_workspace *w) { /* termination.c:19-42 */
  calculate_residuals_and_tolerances(w);
  if ((w->info.pri_res_norm < w->eps_pri) && (w->info.dua_res_norm < w->eps_dua)) {
    update_status(&w->info, OQ_SOLVED); store_solution(w); return 1;
  } else if (is_primal_infeasible(w)) {
    update_status(&w->info, OQ_PRIMAL_INFEASIBLE);
    if (w->has_scaling) { oq_vec_self_mult_scalar(w->delta_y, w->sc_cinv, (size_t)w->m); oq_vec_ew_prod(w->E, w->delta_y, w->delta_y, (size_t)w->m); }
    return 1;
  } else if (is_dual_infeasible(w)) {
    update_status(&w->info, OQ_DUAL_INFEASIBLE);
    if (w->has_scaling) oq_vec_ew_prod(w->D, w->delta_x, w->delta_x, (size_t)w->n);
    return 1;
  }
  return 0;
}

/* =========================================================================================
 * qpalm.c restated: warm_start (:322-399), solve (:401-736), update_* (:739-871)
 * ======================================================================================= */
void oq_warm_start(oq_workspace *w, const oq_float *x_ws, const oq_float *y_ws) {
  size_t n = (size_t)w->n, m = (size_t)w->m;
  w->gamma = w->settings.gamma_init;
  if (w->info.status_val != OQ_UNSOLVED) w->info.setup_time = 0;
  tic(w);
  if (x_ws != NULL) {
    vec_cp(x_ws, w->x, n);
    if (w->has_scaling) oq_vec_ew_prod(w->x, w->Dinv, w->x, n);
    vec_cp(w->x, w->x0, n);
    vec_cp(w->x, w->x_prev, n);
    vec_cp(w->x, w->neg_dphi, n);
    oq_mat_vec(&w->Q, w->neg_dphi, w->Qd);
    if (w->settings.proximal) oq_vec_add_scaled(w->Qd, w->x, w->Qx, 1 / w->settings.gamma_init, n);
    else vec_cp(w->Qd, w->Qx, n);
    oq_mat_vec(&w->A, w->neg_dphi, w->Ad);
    vec_cp(w->Ad, w->Ax, m);
    w->info.objective = compute_objective(w);
  } else {
    oq_vec_set_scalar(w->x, 0., n); oq_vec_set_scalar(w->x_prev, 0., n); oq_vec_set_scalar(w->x0, 0., n);
    oq_vec_set_scalar(w->Qx, 0., n); oq_vec_set_scalar(w->Ax, 0., m);
    w->info.objective = 0.0;
  }
  if (y_ws != NULL) {
    vec_cp(y_ws, w->y, m);
    if (w->has_scaling) { oq_vec_ew_prod(w->y, w->Einv, w->y, m); oq_vec_self_mult_scalar(w->y, w->sc_c, m); }
  } else oq_vec_set_scalar(w->y, 0., m);
  initialize_sigma(w);
  w->initialized = 1;
  w->info.setup_time += toc(w);
}

static void trace_record(oq_workspace *w, oq_int kind) {
  oq_trace *t = w->trace;
  if (!t || t->len >= t->cap) return;
  oq_int r = t->len++;
  if (t->kind) t->kind[r] = kind;
  if (t->fact) t->fact[r] = (kind == 0) ? w->last_fact : 0;
  if (t->nb_active) t->nb_active[r] = w->nb_active;
  if (t->nb_enter) t->nb_enter[r] = w->nb_enter;
  if (t->nb_leave) t->nb_leave[r] = w->nb_leave;
  if (t->tau) t->tau[r] = w->tau;
  if (t->gamma) t->gamma[r] = w->gamma;
  if (t->pri_res_norm) t->pri_res_norm[r] = w->info.pri_res_norm;
  if (t->dua_res_norm) t->dua_res_norm[r] = w->info.dua_res_norm;
  if (t->dua2_res_norm) t->dua2_res_norm[r] = w->info.dua2_res_norm;
  if (t->x) memcpy(t->x + r * w->n, w->x, (size_t)w->n * sizeof(oq_float));
  if (t->y) memcpy(t->y + r * w->m, w->y, (size_t)w->m * sizeof(oq_float));
  if (t->d) memcpy(t->d + r * w->n, w->d, (size_t)w->n * sizeof(oq_float));
  if (t->active) memcpy(t->active + r * w->m, w->active, (size_t)w->m * sizeof(oq_int));
}

static void finish_times(oq_workspace *w) {
  w->info.solve_time = toc(w);
  w->info.run_time = w->info.setup_time + w->info.solve_time;
}

void oq_solve(oq_workspace *w) {
  oq_settings *st = &w->settings;
  size_t n = (size_t)w->n, m = (size_t)w->m;
  w->eps_abs_in = st->eps_abs_in;
  w->eps_rel_in = st->eps_rel_in;
  w->reset_newton = 1;
  w->guard_spent = 0;
  w->gamma = st->gamma_init;
  w->gamma_maxed = (0 || st->nonconvex);
  ivec_set(w->active_old, 0, m);
  if (!w->initialized) oq_warm_start(w, NULL, NULL);
  tic(w);
  if (st->enable_dual_termination) { /* qpalm.c:459-472 */
    factor_sparse_lower(w, &w->Q, &w->LD_Q, 0, 0.0);
    w->info.dual_objective = compute_dual_objective(w);
  } else w->info.dual_objective = 0; /* QPALM_NULL, B8 */

  oq_int iter, iter_out = 0, prev_iter = 0;
  oq_float eps_k_abs = st->eps_abs_in, eps_k_rel = st->eps_rel_in, eps_k;
  oq_int no_change = 0;

  for (iter = 0; iter < st->max_iter; iter++) {
    oq_compute_residuals(w);
    if (oq_check_termination(w)) {
      w->info.iter = iter; w->info.iter_out = iter_out;
      finish_times(w);
      w->initialized = 0;
      return;
    } else if ((w->info.dua2_res_norm <= w->eps_dua_in) || (no_change == 3)) { /* qpalm.c:515, termination.c:254-256 */
      no_change = 0;
      if (iter_out > 0 && w->info.pri_res_norm > w->eps_pri) oq_update_sigma(w);
      vec_cp(w->yh, w->y, m);
      vec_cp(w->Atyh, w->Aty, n);
      if (st->enable_dual_termination) {
        w->info.dual_objective = compute_dual_objective(w);
        if (w->info.dual_objective > st->dual_objective_limit) {
          update_status(&w->info, OQ_DUAL_TERMINATED);
          store_solution(w);
          w->info.iter = iter; w->info.iter_out = iter_out;
          finish_times(w);
          w->initialized = 0;
          return;
        }
      }
      w->eps_abs_in = OQ_MAX(st->eps_abs, st->rho * w->eps_abs_in);
      w->eps_rel_in = OQ_MAX(st->eps_rel, st->rho * w->eps_rel_in);
      if (st->nonconvex) { /* qpalm.c:586-609 */
        if (w->has_scaling) {
          oq_vec_ew_prod(w->Einv, w->Ax, w->temp_2m, m);
          oq_vec_ew_prod(w->Einv, w->z, w->temp_2m + m, m);
          eps_k = eps_k_abs + eps_k_rel * oq_vec_norm_inf(w->temp_2m, m);
        } else eps_k = eps_k_abs + eps_k_rel * OQ_MAX(oq_vec_norm_inf(w->Ax, m), oq_vec_norm_inf(w->z, m));
        if (w->info.pri_res_norm < eps_k) {
          vec_cp(w->x, w->x0, n);
          eps_k_abs = OQ_MAX(st->eps_abs, st->rho * eps_k_abs);
          eps_k_rel = OQ_MAX(st->eps_rel, st->rho * eps_k_rel);
        }
      } else if (st->proximal) { /* qpalm.c:612-630 */
        if (!w->gamma_maxed && iter_out > 0 && w->nb_enter == 0 && w->nb_leave == 0 && w->info.pri_res_norm < w->eps_pri) {
          oq_vec_ew_div(w->y, w->sigma, w->temp_m, m); /* B3: divide here, multiply in compute_residuals */
          oq_vec_add_scaled(w->Ax, w->temp_m, w->Axys, 1, m);
          oq_set_active_constraints(w);
          oq_set_entering_leaving_constraints(w);
          if (w->nb_enter == 0 && w->nb_leave == 0) boost_gamma(w);
          else update_gamma(w);
        } else update_gamma(w);
        vec_cp(w->x, w->x0, n);
      }
      vec_cp(w->pri_res, w->pri_res_in, m);
      iter_out++;
      prev_iter = iter;
      trace_record(w, 1);
    } else if (iter == prev_iter + st->inner_max_iter) { /* qpalm.c:647-660 */
      no_change = 0;
      if (iter_out > 0 && w->info.pri_res_norm > w->eps_pri) oq_update_sigma(w);
      if (st->proximal) {
        update_gamma(w);
        if (!st->nonconvex) vec_cp(w->x, w->x0, n);
      }
      vec_cp(w->pri_res, w->pri_res_in, m);
      iter_out++;
      prev_iter = iter;
      trace_record(w, 2);
    } else { /* qpalm.c:662-676 */
      if (w->nb_enter + w->nb_leave) no_change = 0; else no_change++;
      if (OQ_MOD(iter, st->reset_newton_iter) == 0) w->reset_newton = 1; /* B10 */
      oq_update_primal_iterate(w);
      trace_record(w, 0);
    }
    { /* qpalm.c:680-710 */
      oq_float current_time = w->info.setup_time + toc(w);
      if (current_time > st->time_limit) {
        update_status(&w->info, OQ_TIME_LIMIT_REACHED);
        w->info.iter = iter; w->info.iter_out = iter_out;
        store_solution(w);
        finish_times(w);
        w->initialized = 0;
        return;
      }
    }
  }
  update_status(&w->info, OQ_MAX_ITER_REACHED); /* qpalm.c:712-735 */
  w->info.iter = iter; w->info.iter_out = iter_out;
  store_solution(w);
  finish_times(w);
  w->initialized = 0;
}

void oq_update_settings(oq_workspace *w, const oq_settings *s) { /* qpalm.c:739-791 */
  if (!validate_settings(s)) { update_status(&w->info, OQ_ERROR); return; }
  if (w->settings.scaling > s->scaling) { update_status(&w->info, OQ_ERROR); return; }
  else if (w->settings.scaling < s->scaling) {
    w->first_factorization = 1; /* qpalm.c:774 (USE_LADEL) */
    size_t n = (size_t)w->n, m = (size_t)w->m;
    if (!w->has_scaling) { /* the reference would dereference a NULL scaling struct here */
      w->has_scaling = 1; w->D = zalloc(n); w->Dinv = zalloc(n); w->E = zalloc(m); w->Einv = zalloc(m);
      oq_vec_set_scalar(w->D, 1, n); oq_vec_set_scalar(w->E, 1, m); w->sc_c = 1; w->sc_cinv = 1;
    }
    vec_cp(w->D, w->temp_n, n);
    vec_cp(w->E, w->temp_m, m);
    oq_float c_temp = w->sc_c;
    w->settings.scaling = s->scaling - w->settings.scaling;
    oq_scale_data(w);
    oq_vec_ew_prod(w->D, w->temp_n, w->D, n);
    oq_vec_ew_prod(w->E, w->temp_m, w->E, m);
    w->sc_c *= c_temp;
    oq_vec_ew_recipr(w->D, w->Dinv, n);
    oq_vec_ew_recipr(w->E, w->Einv, m);
    w->sc_cinv = 1 / w->sc_c;
  }
  w->settings = *s;
  w->sqrt_delta = sqrt(w->settings.delta);
}

void oq_update_bounds(oq_workspace *w, const oq_float *bmin, const oq_float *bmax) { /* qpalm.c:793-827 */
  size_t m = (size_t)w->m;
  if (bmin != NULL && bmax != NULL)
    for (size_t j = 0; j < m; j++) if (bmin[j] > bmax[j]) { update_status(&w->info, OQ_ERROR); return; }
  if (bmin != NULL) vec_cp(bmin, w->bmin, m);
  if (bmax != NULL) vec_cp(bmax, w->bmax, m);
  if (w->has_scaling) {
    if (bmin != NULL) oq_vec_ew_prod(w->E, w->bmin, w->bmin, m);
    if (bmax != NULL) oq_vec_ew_prod(w->E, w->bmax, w->bmax, m);
  }
}

void oq_update_q(oq_workspace *w, const oq_float *q) { /* qpalm.c:829-871 */
  size_t n = (size_t)w->n;
  vec_cp(q, w->q, n);
  if (w->has_scaling) {
    oq_vec_ew_prod(w->D, w->q, w->q, n);
    oq_float c_old = w->sc_c, c_ratio;
    if (w->settings.proximal) oq_vec_add_scaled(w->Qx, w->x, w->Qx, -1 / w->gamma, n);
    oq_vec_add_scaled(w->q, w->Qx, w->temp_n, w->sc_cinv, n);
    w->sc_c = 1 / OQ_MAX(1.0, oq_vec_norm_inf(w->temp_n, n));
    w->sc_cinv = 1 / w->sc_c;
    oq_vec_self_mult_scalar(w->q, w->sc_c, n);
    c_ratio = w->sc_c / c_old;
    sp_scale_scalar(&w->Q, w->sc_c / c_old);
    oq_vec_self_mult_scalar(w->Qx, c_ratio, n);
    if (w->settings.proximal) {
      w->gamma = w->settings.gamma_init;
      oq_vec_add_scaled(w->Qx, w->x, w->Qx, 1 / w->gamma, n);
    }
  }
}

void oq_set_trace(oq_workspace *w, oq_trace *t) { w->trace = t; if (t) t->len = 0; }

/* =========================================================================================
 * accessors
 * ======================================================================================= */
const oq_info *oq_get_info(const oq_workspace *w) { return &w->info; }
const oq_float *oq_get_solution_x(const oq_workspace *w) { return w->sol_x; }
const oq_float *oq_get_solution_y(const oq_workspace *w) { return w->sol_y; }
const oq_settings *oq_get_settings(const oq_workspace *w) { return &w->settings; }

oq_float *oq_get_vec(oq_workspace *w, const char *name, oq_int *len) {
  struct { const char *nm; oq_float *p; oq_int l; } tab[] = {
    {"x", w->x, w->n}, {"y", w->y, w->m}, {"Ax", w->Ax, w->m}, {"Qx", w->Qx, w->n}, {"Aty", w->Aty, w->n},
    {"x_prev", w->x_prev, w->n}, {"x0", w->x0, w->n}, {"sigma", w->sigma, w->m}, {"sigma_inv", w->sigma_inv, w->m},
    {"sqrt_sigma", w->sqrt_sigma, w->m}, {"Axys", w->Axys, w->m}, {"z", w->z, w->m}, {"pri_res", w->pri_res, w->m},
    {"pri_res_in", w->pri_res_in, w->m}, {"yh", w->yh, w->m}, {"Atyh", w->Atyh, w->n}, {"df", w->df, w->n},
    {"dphi", w->dphi, w->n}, {"neg_dphi", w->neg_dphi, w->n}, {"dphi_prev", w->dphi_prev, w->n}, {"d", w->d, w->n},
    {"Qd", w->Qd, w->n}, {"Ad", w->Ad, w->m}, {"delta", w->delta, 2 * w->m}, {"alpha", w->alpha, 2 * w->m},
    {"q", w->q, w->n}, {"bmin", w->bmin, w->m}, {"bmax", w->bmax, w->m}, {"D", w->D, w->n}, {"Dinv", w->Dinv, w->n},
    {"E", w->E, w->m}, {"Einv", w->Einv, w->m}, {"At_scale", w->At_scale, w->m}, {"delta_x", w->delta_x, w->n},
    {"delta_y", w->delta_y, w->m}, {"D_temp", w->D_temp, w->n}, {"E_temp", w->E_temp, w->m},
    {"temp_m", w->temp_m, w->m}, {"temp_n", w->temp_n, w->n}, {"xx0", w->xx0, w->n},
    {"sol_kkt", w->sol_kkt, w->kkt_mode ? w->n + w->m : 0}, {"rhs_kkt", w->rhs_kkt, w->kkt_mode ? w->n + w->m : 0}};
  for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); k++)
    if (!strcmp(tab[k].nm, name)) { if (len) *len = tab[k].l; return tab[k].p; }
  if (len) *len = 0;
  return NULL;
}
oq_int *oq_get_ivec(oq_workspace *w, const char *name, oq_int *len) {
  if (!strcmp(name, "active")) { *len = w->m; return w->active; }
  if (!strcmp(name, "active_old")) { *len = w->m; return w->active_old; }
  if (!strcmp(name, "enter")) { *len = w->nb_enter; return w->enter; }
  if (!strcmp(name, "leave")) { *len = w->nb_leave; return w->leave; }
  *len = 0; return NULL;
}
oq_float oq_get_scalar(const oq_workspace *w, const char *name) {
  if (!strcmp(name, "gamma")) return w->gamma;
  if (!strcmp(name, "tau")) return w->tau;
  if (!strcmp(name, "c")) return w->sc_c;
  if (!strcmp(name, "cinv")) return w->sc_cinv;
  if (!strcmp(name, "eta")) return w->eta;
  if (!strcmp(name, "beta")) return w->beta;
  if (!strcmp(name, "eps_pri")) return w->eps_pri;
  if (!strcmp(name, "eps_dua")) return w->eps_dua;
  if (!strcmp(name, "eps_dua_in")) return w->eps_dua_in;
  if (!strcmp(name, "eps_abs_in")) return w->eps_abs_in;
  if (!strcmp(name, "eps_rel_in")) return w->eps_rel_in;
  if (!strcmp(name, "sqrt_sigma_max")) return w->sqrt_sigma_max;
  if (!strcmp(name, "lobpcg_lambda")) return w->lobpcg_lambda;
  if (!strcmp(name, "gamma_init")) return w->settings.gamma_init;
  return NAN;
}
void oq_set_scalar(oq_workspace *w, const char *name, oq_float v) {
  if (!strcmp(name, "gamma")) w->gamma = v;
  else if (!strcmp(name, "tau")) w->tau = v;
  else if (!strcmp(name, "proximal")) w->settings.proximal = (oq_int)v;
  else if (!strcmp(name, "reset_newton")) w->reset_newton = (int)v;
  else if (!strcmp(name, "updown_block")) w->updown_block = (int)v; /* > 1: the blocked multi-rank form of updown_columns (same bits, L streamed once per eight ranks) */
  else if (!strcmp(name, "newton_guard")) w->guard = (v != 0); /* 0: the reference's behaviour, no guard against a non-finite Newton direction (oq_workspace::guard) */
  else if (!strcmp(name, "sparse_mode")) w->sparse_mode = (v != 0 && !w->kkt_mode) ? (int)v : 0; /* before the first solve; 1 = path
                                                         updates where they pay (the engine's rule), 2 = every change refactorises (what pins the mode against the dense one) */
  else if (!strcmp(name, "eps_abs_in")) w->eps_abs_in = v;
  else if (!strcmp(name, "eps_rel_in")) w->eps_rel_in = v;
  else if (!strcmp(name, "nb_sigma_changed")) w->nb_sigma_changed = (oq_int)v; /* op-level tests of ldlupdate_sigma_changed */
}
oq_int oq_get_counter(const oq_workspace *w, const char *name) {
  if (!strcmp(name, "n_refactor")) return w->n_refactor;
  if (!strcmp(name, "n_factor_Q")) return w->n_factor_Q;
  if (!strcmp(name, "n_updown_calls")) return w->n_updown_calls;
  if (!strcmp(name, "n_rank1")) return w->n_rank1;
  if (!strcmp(name, "n_solve")) return w->n_solve;
  if (!strcmp(name, "n_sigma_updates")) return w->n_sigma_updates;
  if (!strcmp(name, "n_boost_gamma")) return w->n_boost_gamma;
  if (!strcmp(name, "n_row_add")) return w->n_row_add;
  if (!strcmp(name, "n_row_del")) return w->n_row_del;
  if (!strcmp(name, "n_refine")) return w->n_refine;
  if (!strcmp(name, "n_guard_refactor")) return w->n_guard_refactor;
  if (!strcmp(name, "n_lobpcg_iter")) return w->n_lobpcg_iter;
  if (!strcmp(name, "nonconvex")) return w->settings.nonconvex;
  if (!strcmp(name, "kkt_mode")) return w->kkt_mode;
  if (!strcmp(name, "nb_active")) return w->nb_active;
  if (!strcmp(name, "nb_enter")) return w->nb_enter;
  if (!strcmp(name, "nb_leave")) return w->nb_leave;
  if (!strcmp(name, "nb_sigma_changed")) return w->nb_sigma_changed;
  if (!strcmp(name, "initialized")) return w->initialized;
  if (!strcmp(name, "n")) return w->n;
  if (!strcmp(name, "m")) return w->m;
  return -1;
}
void oq_get_matrix(oq_workspace *w, const char *name, oq_int *nrow, oq_int *ncol, oq_int **p, oq_int **i, oq_float **x) {
  oq_sparse *S = NULL;
  if (!strcmp(name, "A")) S = &w->A; else if (!strcmp(name, "Q")) S = &w->Q; else if (!strcmp(name, "At_sqrt_sigma")) S = &w->At_sqrt_sigma;
  if (!S || !S->p) { *nrow = *ncol = 0; *p = *i = NULL; *x = NULL; return; }
  *nrow = S->nrow; *ncol = S->ncol; *p = S->p; *i = S->i; *x = S->x;
}
/* ---- the KKT operations of solver_interface.h:82-126 one by one (workspaces set up with FACTORIZE_KKT), for op-level tests ---- */
void oq_kkt_form_and_factor(oq_workspace *w) { if (w->kkt_mode) { kkt_form_and_factor(w); w->first_factorization = 0; } } /* newton.c:32-45 */
void oq_kkt_update_entering_constraints(oq_workspace *w) { /* solver_interface.c:202-218 */
  if (w->kkt_mode) for (oq_int e = 0; e < w->nb_enter; e++) kkt_row_add(w, w->enter[e]);
}
void oq_kkt_update_leaving_constraints(oq_workspace *w) { /* solver_interface.c:220-236 */
  if (w->kkt_mode) for (oq_int e = 0; e < w->nb_leave; e++) kkt_row_del(w, w->leave[e]);
}
void oq_kkt_solve(oq_workspace *w) { /* solver_interface.c:238-247 */
  if (!w->kkt_mode) return;
  const oq_int n = w->n, m = w->m, np = n + m;
  for (oq_int j = 0; j < n; j++) w->rhs_kkt[j] = w->dphi[j] * -1;
  for (oq_int k = 0; k < m; k++) w->rhs_kkt[n + k] = 0;
  vec_cp(w->rhs_kkt, w->sol_kkt, (size_t)np);
  oq_dense_ldl_solve(np, w->LDK.L, np, w->LDK.D, w->sol_kkt);
  vec_cp(w->sol_kkt, w->d, (size_t)n);
}
const oq_float *oq_get_kkt_factor(const oq_workspace *w, const oq_float **D, oq_int *ld) {
  if (D) *D = w->LDK.D;
  if (ld) *ld = w->n + w->m;
  return w->LDK.L;
}
/* TEST INFRASTRUCTURE: the symmetric permutation the sparse-storage mode factorises under (the engine's, read back through
 * qpg_batch_sparse_perm by tests/test_sparse_factor.py), before the first solve; perm[new] = old */
int oq_set_perm(oq_workspace *w, const oq_int *perm, oq_int n) {
  if (!w || n != w->n || w->sp_Lp) return 1;
  oq_int *pm = izalloc(nz1((size_t)n)), *ip = izalloc(nz1((size_t)n));
  for (oq_int j = 0; j < n; j++) ip[j] = -1;
  for (oq_int j = 0; j < n; j++) { if (perm[j] < 0 || perm[j] >= n || ip[perm[j]] >= 0) { free(pm); free(ip); return 1; } pm[j] = perm[j]; ip[perm[j]] = j; }
  free(w->sp_perm); free(w->sp_iperm);
  w->sp_perm = pm; w->sp_iperm = ip;
  return 0;
}
oq_int oq_sparse_levels(const oq_workspace *w) { return w->sp_nlev; }
const oq_float *oq_get_factor(const oq_workspace *w_, const oq_float **D, oq_int *ld) {
  oq_workspace *w = (oq_workspace *)w_;
  if (w->sparse_mode && w->sp_Lp) { /* tests of the sparse-storage mode on small problems: the compressed columns spread out into the dense layout */
    oq_int n = w->n;
    factor_alloc(&w->LD, n);
    memset(w->LD.L, 0, (size_t)(n * n) * sizeof(oq_float));
    for (oq_int j = 0; j < n; j++) { w->LD.D[j] = w->sp_D[j]; for (oq_int e = w->sp_Lp[j]; e < w->sp_Lp[j + 1]; e++) w->LD.L[w->sp_Li[e] + j * n] = w->sp_Lx[e]; }
  }
  if (D) *D = w->LD.D;
  if (ld) *ld = w->n;
  return w->LD.L;
}
