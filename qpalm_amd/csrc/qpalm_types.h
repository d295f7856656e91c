/*
 * qpalm_types.h -- plain-data structures shared by the host C-ABI layer and the gfx950 kernels.
 *
 * Device layout (HBM): a "batch" holds B independent QPs of identical dimensions (n, m).  Every
 * named vector of the reference's QPALMWorkspace (include/types.h:197-314) is one [B][n] or [B][m]
 * fp64 array, QP-major, so that the workgroup that owns QP b streams contiguous, coalesced rows.
 * Indices are int32 on the device (n, m, nnz < 2^31); the C-ABI takes the reference's 64-bit c_int.
 */
#ifndef QPALM_TYPES_H_GFX950
#define QPALM_TYPES_H_GFX950

#include <stdint.h>

/* status codes, include/constants.h:30-37 */
#define QPG_SOLVED 1
#define QPG_DUAL_TERMINATED 2
#define QPG_MAX_ITER_REACHED (-2)
#define QPG_PRIMAL_INFEASIBLE (-3)
#define QPG_DUAL_INFEASIBLE (-4)
#define QPG_TIME_LIMIT_REACHED (-5)
#define QPG_UNSOLVED (-10)
#define QPG_ERROR 0
#define QPG_INFTY 1e20

/* == QPALMSettings (include/types.h:119-150), c_int = 64-bit */
typedef struct {
  int64_t max_iter, inner_max_iter;
  double eps_abs, eps_rel, eps_abs_in, eps_rel_in, rho, eps_prim_inf, eps_dual_inf, theta, delta, sigma_max,
      sigma_init;
  int64_t proximal;
  double gamma_init, gamma_upd, gamma_max;
  int64_t scaling, nonconvex, verbose, print_iter, warm_start, reset_newton_iter, enable_dual_termination;
  double dual_objective_limit, time_limit;
  int64_t ordering, factorization_method, max_rank_update;
  double max_rank_update_fraction;
} qpg_settings;

#define QPG_CU_KEYS 4096 /* 16 XCC ids x 256 (se, sh, cu) ids of HW_REG_HW_ID */
#define QPG_NDBG 22
#define QPG_CNT_SWEEP_ENTRIES 16 /* entries of L (doubles) the rank-update sweeps read AND wrote: sum of nnz(L[:, J0:]) */
#define QPG_CNT_SWEEPS 17        /* sweeps over the panel (<= K ranks each, K = 16 or 8 by instantiation) */
#define QPG_CNT_FACTOR_REREAD 18 /* entries of L re-read by the left-looking panel updates of the factorisation */
#define QPG_CNT_PLACEMENT 19     /* not a counter: where the workgroup ran (QPGStats.placement) */
#define QPG_CNT_SEQ_COLS 20      /* columns of the update sweeps whose pivots the per-column guard re-summed as running pivots (qp_rank_pivots) */
#define QPG_CNT_SWEEP_COLS 21    /* columns the diagonal-block recurrences of the update sweeps went through (the guard's denominator) */
/* dynamic LDS of a 512-thread workgroup: the update sweep's scratch (UpdownLds<2, 16>: 77 192 bytes); two workgroups plus
 * their static LDS fit the 160 KB of a CU */
#define QPG_LDS_DEFAULT 77824

/* per-QP scalar state: everything qpalm_solve keeps in locals or in QPALMWorkspace scalars
 * (src/qpalm.c:401-482, include/types.h:197-314), so that a solve can be suspended after any
 * iteration and resumed by a later launch. */
typedef struct {
  double gamma, tau, eta, beta;
  double eps_pri, eps_dua, eps_dua_in, eps_abs_in, eps_rel_in, eps_k_abs, eps_k_rel;
  double sqrt_sigma_max, sqrt_delta, sc_c, sc_cinv;
  double pri_res_norm, dua_res_norm, dua2_res_norm, objective, dual_objective;
  double setup_time, solve_time;
  /* nonconvex QPs (set_settings_nonconvex, nonconvex.c:171-183, changes settings PER WORKSPACE; a batch shares one settings
   * block, so the per-QP values live here): nc_flag = settings->nonconvex after LOBPCG (lambda < 0), nc_gamma = 1/|lambda|
   * = this QP's gamma_init = gamma_max (proximal forced on), lobpcg_lambda / lobpcg_iter for inspection,
   * norm_Ax_z = the norm of eps_pri (termination.c:92-100) reused by the nonconvex tolerance eps_k (qpalm.c:586-596) */
  double nc_gamma, lobpcg_lambda, norm_Ax_z;
  int32_t nc_flag, lobpcg_iter;
  int32_t iter, iter_out, prev_iter, no_change;
  int32_t status, done, initialized, gamma_maxed, reset_newton, in_solve;
  int32_t nb_active, nb_enter, nb_leave, nb_sigma_changed;
  int32_t last_kind, last_fact, slot, has_scaling;
  /* "coop" mode (one large QP, the linear algebra spread over many workgroups by the host, qpalm_capi.inc: coop_solve): the
   * iteration is suspended at its linear-algebra site.  pend_stage = 1: the host has to factorise (pend_la = 1 Q + A'SA, 3 Q only,
   * 7 LD_Q of the dual objective; 0 no factorisation) and, for a Newton step (pend_kind == 0, pend_la != 7), to solve for d */
  int32_t pend_stage, pend_la, pend_action, pend_kind, pend_nchange;
  int32_t seq_hint; /* set by the host at setup: Q has a column without a positive diagonal entry (an LP, a semidefinite Q): the factor's pivots can come down
                       to 1 / gamma -- its update sweeps sum the ranks' contributions to a pivot one after the other, like those of nonconvex QPs (dev_updown) */
  int32_t kkt_na; /* KKT mode: -1 = the factor slot holds the full (n+m) layout; na >= 0 = the COMPACT factor of the variables + the na constraints listed
                     in kkt_list (the factor is spread out only when a row addition / deletion or a read of the factor needs the full layout) */
  double pend_gam;
  int64_t pend_clock; /* device clock (100 MHz, the same counter in every launch) when the iteration was suspended: the time the host's
                         kernels take until it resumes is added to solve_time, so that run_time and time_limit see wall time (qpalm.c:680-723) */
  int32_t guard_redo, n_guard_refactor, guard_spent, guard_pad; /* guard_spent: a redone step came out non-finite as well (the right-hand side is not finite, or H is singular by construction): the guard is off for the rest of this solve */
  /* (the two below:) */ /* guard_redo: the Newton direction of the last pass was not finite on an UPDATED factor (a pivot went through zero inside a sweep):
                                           the next pass refactorises and solves again instead of stepping (dev_solve; the oracle restates the same guard); the count of such passes */
  int32_t dual_pending, kkt_first; /* kkt_first: solver->first_factorization (types.h:176), KKT path; dual_pending: the factor of Q and the initial dual objective (qpalm.c:459-468) are still to be computed */
  /* work counters (device side statistics for the roofline accounting in bench.py) */
  int32_t n_refactor, n_factor_Q, n_sweeps, n_rank1, n_solve, n_sigma_updates, n_boost_gamma, n_fused_solve; /* n_fused_solve: Newton solves whose forward substitution rode on the last update sweep (L streamed once less) */
  int64_t ticks_total, ticks_factor, ticks_update, ticks_solve, ticks_linesearch, ticks_resid;
  int64_t ticks_dbg[QPG_NDBG]; /* [0..15] fine-grained phase timers (100 MHz ticks), see QPGStats.ms_dbg;
                                  [16..] work counters written by the linear-algebra functions themselves (QPG_CNT_*) */
} qpg_scalars;

/* view of one batch in device memory; passed by value to the kernels */
typedef struct {
  int32_t B, n, m, ld, nnzA, nnzQ, nnzQf, nslots, lds_bytes, update_rank_threshold, ls_stride, wst_stride, place_panel_wave, narrow_rows;
  int32_t offload, sweep_ranks; /* sweep_ranks: most ranks one sweep of the rank update applies (16, the default, or 32 = the multi-pass form of dense_updown: bit-identical factors, measured slower).  coop mode: 1 = dev_solve suspends at its linear-algebra site for factorisations and Newton solves, 2 = for rank updates too */
  int32_t kkt_compact, kkt_pad; /* 1: KKT mode factorises the variables + ACTIVE constraints only and spreads the factor out (qpalm_kkt.h) */
  int32_t kkt, nfac; /* kkt != 0: FACTORIZE_KKT, the factor slots hold the (n+m) x (n+m) KKT panel; nfac = rows of a factor slot
                        (n, or n + m in KKT mode); ld = its leading dimension */
  /* problem data.  A: CSC m x n.  At: CSC of A' (n x m) with the permutation into A's entries.
   * Q: lower CSC.  Qf: both triangles (row == column compressed), with permutation into Q. */
  int32_t *Ap, *Ai, *Atp, *Ati, *Atperm, *Ainv, *Qp, *Qi, *Qfp, *Qfi, *Qfperm; /* Ainv: position in A' of every entry of A */
  double *Ax, *Atx, *Atss, *Qx, *Qfx;
  double *q, *bmin, *bmax, *c0;
  /* iterates / work vectors */
  double *x, *y, *Axv, *Qxv, *Aty, *x_prev, *x0;
  double *sigma, *sigma_inv, *sqrt_sigma, *At_scale;
  double *Axys, *z, *pri_res, *pri_res_in, *yh, *Atyh, *df, *dphi, *dphi_prev, *d, *Qd, *Ad;
  double *delta_y, *delta_x, *temp_n, *temp_m;
  double *D, *Dinv, *E, *Einv;
  double *ls_key;       /* [B][2m] line-search keys (global fallback / scratch) */
  int32_t *ls_idx;      /* [B][2m] */
  double *ls_delta, *ls_alpha; /* [B][2m] */
  int32_t *active, *active_old, *enter, *leave; /* [B][m] */
  double *sol_x, *sol_y;
  /* factor slots */
  double *L;   /* [nslots][ld*n] column-major, unit lower, strict lower part used */
  double *Dg;  /* [nslots][n] */
  double *LQ;  /* [nslots][ld*n] second resident factor: LD_Q = LDL' of Q for compute_dual_objective (qpalm.c:466-467);
                  allocated only when enable_dual_termination is set, else NULL */
  double *DgQ; /* [nslots][n] */
  double *dual_rhs; /* [B][n] Aty + q (the reference uses neg_dphi for it, iteration.c:276) */
  double *kkt_sol, *kkt_rhs, *kkt_tmp, *kkt_rhs2; /* [B][n+m] sol_kkt / rhs_kkt of the KKT path (qpalm.c:241-242) + scratch (kkt_rhs2: the compact right-hand side); NULL in Schur mode */
  int32_t *nq, *mq;   /* [B] per-QP dimensions (<= n, m: members of a mixed-size batch are padded to the batch strides); NULL = uniform */
  int32_t *kkt_list;  /* [B][m] the active constraints (ascending) the compact KKT factor was formed with (qpg_scalars.kkt_na) */
  int32_t *kkt_state; /* [B][m] 0 unit diagonal, 1 row present, 2 deleted by row_del (solver_interface.c:151-156,226-235) */
  double *Wst; /* [nslots][wst_stride]: staging of the rank-update vectors, dummy cells, exported tables (QPG_WST_STRIDE) */
  double *op_in, *op_out; /* [max(n,m)] scratch of the single-QP boundary operations */
  /* sparse factor (qpalm_sparse.h; NULL / 0 in the dense modes): symbolic arrays per QP with the batch's strides, values per slot */
  int32_t sparse, sp_nnzL;  /* sp_nnzL: stride of the entry arrays = the largest nnz(L) of the batch */
  int32_t *sp_Lp, *sp_Li;   /* [B][n+1], [B][sp_nnzL]: strict lower pattern of L by columns, rows ascending */
  int32_t *sp_Rp, *sp_Rk, *sp_Rpos; /* [B][n+1], [B][sp_nnzL] x 2: the same entries by rows: column k and position in column k's arrays, k ascending */
  int32_t *sp_levptr, *sp_levcol, *sp_nlev; /* [B][n+1], [B][n], [B]: level sets of the elimination tree */
  double *sp_Lx;            /* [nslots][sp_nnzL] */
  double *sp_wv;            /* [nslots][wavefronts][n] dense work vectors of the factorisation (zero outside of use) */
  double *co_tab;           /* coop mode, persistent update sweep: [B][co_tab_stride] the tables the owners of the diagonal blocks publish */
  int32_t *co_flags;        /* [B][4]: [0] blocks published so far, [1] != 0: a workgroup gave up waiting */
  int64_t co_tab_stride;
  int32_t *sp_perm;         /* [B][n] the factor is that of P H P': perm[new] = old (identity: natural ordering) */
  int32_t *sp_AtiP, *sp_QfiP, *sp_first; /* [B][nnzA] Ati, [B][nnzQf] Qfi in the factor's numbering; [B][m] first (smallest) such column of every row of A */
  double *sp_tmp;           /* [nslots][n] the permuted right-hand side of a solve */
  int32_t seq_mode;         /* context option "sequential_rank_sums": -1 = where the factor can get near-singular (nonconvex QPs, QPs flagged by seq_hint), 1 = always, 0 = never */
  int32_t ls_hbm;           /* tests: >= 1 = the line search keeps its sort buffer in HBM (the LDS-tiled sort) whatever m; >= 2: tiles of at most that many entries */
  int32_t sp_lds;           /* context option "sparse_lds" (default 1): columns are accumulated, and right-hand sides solved, in LDS where they fit (0: the HBM forms) */
  int32_t sp_gpw;           /* columns a wavefront factorises at a time (1, 2, 4 or 8 groups of 64 / sp_gpw lanes): sp_wv holds wavefronts x sp_gpw work vectors per slot */
  qpg_scalars *sc; /* [B] */
  qpg_settings *settings; /* [1] */
  int32_t *queue; /* [64 + QPG_CU_KEYS]: [0] work-queue head; [64 + key] workgroups that have arrived on compute unit `key` in this launch */
  int32_t *order; /* [B] the work queue hands out order[0], order[1], ...: the members by the kernel time of their PREVIOUS solve, longest first
                     (k_queue_order; identity before the first solve), so that the last QPs of a launch are short ones.  NULL: index order */
} qpg_view;

#define QPG_RPT_SPARSE 8 /* template argument of k_solve / dev_solve that selects the sparse factor (the dense instances use 0, 1, 2, 4) */
#define QPG_KMAX 16 /* ranks per sweep of the large-factor and coop-mode sweeps */
#define QPG_KWST 32 /* dense update vectors the staging area of a slot holds = most ranks per sweep of dense_updown */
#define QPG_DUMMY 4096 /* doubles per slot that masked-off rows load from / store to */
#define QPG_HSTASH 4096 /* doubles per slot of scratch behind the dummy area (coop mode: the state of a grid update) */
/* a slot of Wst: [QPG_KWST][rows] update vectors | QPG_DUMMY | QPG_HSTASH | [rows][QPG_KWST][2] exported (-w, -gamma) tables of a multi-pass sweep */
#define QPG_WST_STRIDE(rows) ((size_t)QPG_KWST * (rows) + QPG_DUMMY + QPG_HSTASH + (size_t)2 * QPG_KWST * (rows))

#endif
