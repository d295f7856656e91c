/*
 * qpalm_gfx950.h -- C ABI of the MI355X (gfx950) backend for the QPALM semismooth-Newton path.
 *
 * This is the drop-in boundary: plain C, `extern "C"`, caller-owned host buffers, int status codes,
 * no torch / C++ types.  It replaces what the reference reaches through its compile-time solver
 * backend (include/types.h:17-32 typedef block + include/solver_interface.h) and the API of
 * include/qpalm.h for a BATCH of independent QPs whose state lives in HBM.
 *
 * Reference interface replaced by each entry point (paths relative to Benny44/QPALM):
 *
 *   qpg_batch_create / qpg_batch_set_problem / qpg_batch_setup
 *        qpalm_setup                      include/qpalm.h:59-60,   src/qpalm.c:73-319
 *        (validate_data/validate_settings src/validate.c:18-221,  scale_data src/scaling.c:34-113)
 *   qpg_batch_warm_start                  qpalm_warm_start  include/qpalm.h:71-73, src/qpalm.c:322-399
 *   qpg_batch_solve / qpg_batch_iterate   qpalm_solve       include/qpalm.h:82,    src/qpalm.c:401-736
 *   qpg_batch_update_settings/bounds/q    qpalm_update_*    include/qpalm.h:95-126, src/qpalm.c:739-871
 *   qpg_batch_destroy                     qpalm_cleanup     include/qpalm.h:133,   src/qpalm.c:874-1096
 *   qpg_mat_vec / qpg_mat_tpose_vec       mat_vec / mat_tpose_vec          include/solver_interface.h:33,47
 *   qpg_ldlchol / qpg_ldlchol_matrix      ldlchol                           include/solver_interface.h:172
 *   qpg_sparse_matvec                     mat_vec / mat_tpose_vec on a caller-owned matrix
 *   qpg_ldlcholQAtsigmaA                  ldlcholQAtsigmaA                  include/solver_interface.h:185
 *   qpg_ldlupdate_entering_constraints    ldlupdate_entering_constraints    include/solver_interface.h:196
 *   qpg_ldldowndate_leaving_constraints   ldldowndate_leaving_constraints   include/solver_interface.h:207
 *   qpg_ldlupdate_sigma_changed           ldlupdate_sigma_changed           include/solver_interface.h:218
 *   qpg_ldlsolveLD_neg_dphi               ldlsolveLD_neg_dphi               include/solver_interface.h:228
 *   qpg_exact_linesearch                  exact_linesearch                  include/linesearch.h, src/linesearch.c:14-120
 *   qpg_compute_residuals                 compute_residuals                 include/iteration.h,  src/iteration.c:24-48
 *   qpg_set_active_constraints            set_active_constraints + set_entering_leaving_constraints src/newton.c:122-149
 *
 * All functions return QPG_OK (0) or a negative error code; qpg_last_error() gives the message.
 * There is NO CPU fallback: without a usable gfx950 device every entry point fails.
 */
#ifndef QPALM_GFX950_H
#define QPALM_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t qpg_int;   /* == c_int  (CMakeLists.txt:53 -DDLONG, include/global_opts.h:31-39) */
typedef double  qpg_float; /* == c_float (include/global_opts.h:61) */

#define QPG_OK 0
#define QPG_ERR_NO_DEVICE (-1)
#define QPG_ERR_INVALID (-2)
#define QPG_ERR_ALLOC (-3)
#define QPG_ERR_RUNTIME (-4)
#define QPG_ERR_UNSUPPORTED (-5)

/* == QPALMSettings, include/types.h:119-150 (same field order; ABI witnessed by
 *    interfaces/python/qpalm.py:49-80) */
typedef struct {
  qpg_int   max_iter;
  qpg_int   inner_max_iter;
  qpg_float eps_abs;
  qpg_float eps_rel;
  qpg_float eps_abs_in;
  qpg_float eps_rel_in;
  qpg_float rho;
  qpg_float eps_prim_inf;
  qpg_float eps_dual_inf;
  qpg_float theta;
  qpg_float delta;
  qpg_float sigma_max;
  qpg_float sigma_init;
  qpg_int   proximal;
  qpg_float gamma_init;
  qpg_float gamma_upd;
  qpg_float gamma_max;
  qpg_int   scaling;
  qpg_int   nonconvex;
  qpg_int   verbose;
  qpg_int   print_iter;
  qpg_int   warm_start;
  qpg_int   reset_newton_iter;
  qpg_int   enable_dual_termination;
  qpg_float dual_objective_limit;
  qpg_float time_limit;
  qpg_int   ordering;
  qpg_int   factorization_method;
  qpg_int   max_rank_update;
  qpg_float max_rank_update_fraction;
} QPGSettings;

/* == QPALMInfo with PROFILING, include/types.h:76-95 */
typedef struct {
  qpg_int   iter;
  qpg_int   iter_out;
  char      status[32];
  qpg_int   status_val;
  qpg_float pri_res_norm;
  qpg_float dua_res_norm;
  qpg_float dua2_res_norm;
  qpg_float objective;
  qpg_float dual_objective;
  qpg_float setup_time;
  qpg_float solve_time;
  qpg_float run_time;
} QPGInfo;

/* device-side work counters of one QP (for the roofline accounting) */
typedef struct {
  qpg_int n_refactor, n_factor_Q, n_sweeps, n_rank1, n_solve, n_sigma_updates, n_boost_gamma;
  qpg_int nb_active, nb_enter, nb_leave, last_kind, last_fact;
  qpg_float gamma, tau, eta, beta, eps_pri, eps_dua, eps_dua_in, sc_c;
  qpg_float ms_total, ms_factor, ms_update, ms_solve, ms_linesearch;
  qpg_float ms_dbg[16]; /* update: 0 staging, 1 panel wave busy, 2 last trailing wave busy, 7 sweep phases (wall);
                           factor: 3 form, 4 panel gemm, 5 block, 6 panel solve;
                           solve: 8 forward block, 9 forward rows below, 10 backward dots, 11 backward block; 12 SpMV+residuals;
                           line search: 13 SpMVs, 14 breakpoints + compaction, 15 sort (the scan is the rest) */
  qpg_int sweep_entries;         /* entries of L the rank-update sweeps touched (each read and written once): sum of nnz(L[:, J0:]) */
  qpg_int factor_reread_entries; /* entries of L re-read by the panel updates of the factorisations (beyond the compulsory write) */
  qpg_float lobpcg_lambda;       /* nonconvex QPs: the eigenvalue bound of lobpcg (nonconvex.c:29-168); gamma_init = gamma_max = 1/|lambda| */
  qpg_int placement;             /* where the QP's last solve ran: (XCC, SE, SH, CU) key << 16 | arrival index of the workgroup on
                                    that CU << 8 | panel wavefront << 4 | SIMD of wavefront 0 (qp_place_panel_wave) */
  qpg_int lobpcg_iter, nonconvex; /* LOBPCG iterations; settings->nonconvex of THIS QP after set_settings_nonconvex (:171-183) */
  qpg_int n_fused_solve;         /* of n_solve: solves whose forward substitution was done by the last update sweep (L read once, not twice) */
  qpg_int n_seq_columns;         /* columns of the update sweeps whose pivots the per-column guard re-summed as the reference's running pivot (a pivot shrank by 2^8 or more) */
  qpg_int n_sweep_columns;       /* ... out of this many columns of diagonal-block recurrences */
  qpg_int n_guard_refactor;      /* Newton steps redone with a fresh factorisation because the direction from an updated factor was not finite */
} QPGStats;

typedef struct qpg_ctx qpg_ctx;
typedef struct qpg_batch qpg_batch;

const char *qpg_last_error(void);
const char *qpg_backend_name(void); /* "gfx950-hip" for the shipped library */

void qpg_set_default_settings(QPGSettings *s);          /* src/qpalm.c:38-70 */
int  qpg_validate_settings(const QPGSettings *s);       /* 1 = valid, src/validate.c:43-221 */

int  qpg_ctx_create(int device, qpg_ctx **out);
void qpg_ctx_destroy(qpg_ctx *ctx);
/* Engine options (no reference counterpart; they change speed or placement, never what is computed):
 *   "max_slots"             resident factor panels = workgroups in flight (default 512)
 *   "lds_bytes"             dynamic LDS per 512-thread workgroup
 *   "update_rank_threshold" -1 = the reference's refactorise-or-update rule (newton.c:98-101), k >= 0: refactorise beyond k changed rows
 *   "small_workgroups"      1 = factors of at most 256 rows run on the 256-thread instance of the kernels (four workgroups per CU), and
 *                           batches of more than 2 max_slots QPs with factors of at most 192 rows on the 128-thread instance (seven per CU);
 *                           2 = the 128-thread instance wherever its 21.5 KB of LDS fit, 3 = never the 128-thread one, 0 = neither
 *   "narrow_rows"           1 = quarter-wavefront Schur assembly for short rows of A
 *   "ld_align"              leading dimension of the factor panels in doubles (16 = every column on a 128-byte line)
 *   "sweep_ranks"           16 (default) or 32 ranks per update sweep (32: the multi-pass sweep, bit-identical factors, slower)
 *   "kkt_compact"           1 = FACTORIZE_KKT factorises the variables + ACTIVE constraints only and spreads the factor out on demand
 *   "place_panel_wave"      0 / 1 / 2: SIMD placement of the sweeps' panel wavefronts (0 = the hardware's own)
 *   "sequential_rank_sums"  how an update sweep sums a column's pivots.  -1 (default) = a prefix tree, and any column in which a pivot
 *                           shrinks by 2^8 or more inside the sweep is summed again as the reference's running pivot (per-column guard); the running
 *                           pivot in every column for QPs whose factor can get near-singular (nonconvex; Q without a positive diagonal: LPs).
 *                           1 = the running pivot everywhere (-5 % on the benchmark); 0 = the unguarded tree (A/B runs only: loses eight digits on a
 *                           downdate into a numerically singular matrix)
 *   "queue_order"           1 (default) = a launch of more QPs than resident slots starts the members in descending order of the kernel time of
 *                           their previous solve (the launch's tail is then made of short solves); 0 = index order.  Never changes a result.
 *   "sparse_factor", "sparse_ordering"   the sparse L D L' and its ordering (qpg_batch_sparse_info / qpg_batch_sparse_perm below)
 *   "sparse_gpw"            columns of a level a wavefront of the sparse factorisation takes at a time: 1, 2, 4 or 8 groups of 64 / gpw lanes (default 8,
 *                           halved while the work vectors of all resident factors would exceed 4 GB).  Never changes a result.
 *   "sparse_lds"            1 (default) = the sparse factorisation accumulates a column, the sparse solves keep the right-hand side and the path
 *                           updates their work vector in LDS where they fit; 0 = the forms on vectors in HBM (same iterates bit for bit: A/B and
 *                           tests); >= 2 = LDS only for columns of at most that many entries (tests)
 *   "coop", "coop_workgroups", "coop_updates", "coop_rank_threshold"   one large QP on many workgroups
 * Environment: QPALM_HOST_THREADS = host threads of qpg_batch_set_problems (default: hardware threads, at most 24). */
int  qpg_ctx_set_option(qpg_ctx *ctx, const char *name, qpg_int value);

/* A batch = B QPs of dimensions up to (n, m) (equal for all members with qpg_batch_set_problem, smaller ones through
 * qpg_batch_set_problem_sized).  nnzA_max / nnzQ_max bound the entries of any member. */
int  qpg_batch_create(qpg_ctx *ctx, qpg_int B, qpg_int n, qpg_int m, qpg_int nnzA_max, qpg_int nnzQ_max,
                      const QPGSettings *settings, qpg_batch **out);
/* CSC, 64-bit indices as in cholmod_sparse; Q symmetric, only entries with row >= col are read
 * (stype = -1, B6).  Data is copied -- converted straight into the batch's host slab (huge pages, laid out as it is uploaded);
 * qpg_batch_create also starts the allocation of the batch's device memory on a thread of its own, qpg_batch_setup joins it. */
int  qpg_batch_set_problem(qpg_batch *bt, qpg_int idx, const qpg_int *Qp, const qpg_int *Qi, const qpg_float *Qx,
                           const qpg_int *Ap, const qpg_int *Ai, const qpg_float *Ax, const qpg_float *q,
                           qpg_float c, const qpg_float *bmin, const qpg_float *bmax);
/* A member smaller than the batch (n <= batch n, m <= batch m): size-bucketed batches, e.g. the Maros-Meszaros QPS set
 * streamed through interfaces/qps (BASELINE.json config 4).  The member keeps its own dimensions on the device; results
 * come back in the batch's [B][n] / [B][m] arrays, entries beyond the member's n / m are zero. */
int  qpg_batch_set_problem_sized(qpg_batch *bt, qpg_int idx, qpg_int n, qpg_int m, const qpg_int *Qp, const qpg_int *Qi,
                                 const qpg_float *Qx, const qpg_int *Ap, const qpg_int *Ai, const qpg_float *Ax,
                                 const qpg_float *q, qpg_float c, const qpg_float *bmin, const qpg_float *bmax);
/* qpg_batch_set_problem_sized for members first .. first + count - 1 in ONE call (entry k of every array = member first + k; n / m
 * NULL: every member has the batch's dimensions; c NULL: zero constants): the per-QP host work of qpalm_setup -- deep copies,
 * sorted CSC, the A' pattern (src/qpalm.c:128-144, iteration.c:81) -- runs on host threads (QPALM_HOST_THREADS; default: at most 24).  No reference counterpart (the reference sets up one QP per call). */
int  qpg_batch_set_problems(qpg_batch *bt, qpg_int first, qpg_int count, const qpg_int *n, const qpg_int *m,
                            const qpg_int *const *Qp, const qpg_int *const *Qi, const qpg_float *const *Qx,
                            const qpg_int *const *Ap, const qpg_int *const *Ai, const qpg_float *const *Ax,
                            const qpg_float *const *q, const qpg_float *c, const qpg_float *const *bmin, const qpg_float *const *bmax);
int  qpg_batch_setup(qpg_batch *bt);                               /* upload (one DMA per array from the host slab) + Ruiz scaling on device */
int  qpg_batch_warm_start(qpg_batch *bt, const qpg_float *x, const qpg_float *y); /* [B][n], [B][m] or NULL */
int  qpg_batch_warm_start_last(qpg_batch *bt);                     /* qpalm_warm_start(work, last x, last y) of every QP, from HBM (no host copy) */
int  qpg_batch_solve(qpg_batch *bt);                               /* run every QP to termination */
int  qpg_batch_iterate(qpg_batch *bt, qpg_int k);                  /* at most k more loop iterations each */
int  qpg_batch_begin_solve(qpg_batch *bt);                         /* start of a qpalm_solve driven by qpg_batch_iterate: finished QPs start over */
int  qpg_batch_last_solve_ms(qpg_batch *bt, float *ms);            /* HIP-event time of the last solve/iterate launch */
int  qpg_batch_num_unfinished(qpg_batch *bt, qpg_int *count);
/* how the batch runs: concurrent workgroups (= resident factor panels), threads per workgroup (512, or 256 for QPs whose
 * factor has at most 256 rows -- four workgroups per CU instead of two), dynamic LDS per workgroup.  No reference
 * counterpart (the reference runs one QP on one core); used by bench.py's per-phase bandwidth figures. */
int  qpg_batch_launch_shape(qpg_batch *bt, qpg_int *workgroups, qpg_int *threads, qpg_int *lds_bytes);
/* The sparse L D L' (round 5): replaces what solver_interface.c:319-370, 523-541 hands to cholmod_analyze / cholmod_factorize for
 * factors that are sparse or have more than 8192 rows (context option "sparse_factor": -1 automatic for > 8192 rows, 1 always,
 * 0 never; Schur path; dual termination keeps a second value array per slot for LD_Q, the factor of Q alone on the same pattern).  nnzL: entries of the member's strict lower triangle; device_bytes: the block that holds
 * the symbolic arrays of all members and the values of all resident factors.  QPG_ERR_UNSUPPORTED on a batch with dense factors.
 * Where its policy is not the reference's: rows that enter or leave the active set, and rows whose penalty changed
 * (ldlupdate_sigma_changed, solver_interface.c:443-503), are rank-1 updates along their elimination-tree paths only while
 * 2 x changed rows x tree height < n; beyond that the factor is rebuilt (on a chain-like tree -- a band under the natural ordering --
 * a path is the whole matrix).  QPGStats.n_refactor / n_rank1 then differ from the reference's split; the matrices factorised do not. */
int  qpg_batch_sparse_info(qpg_batch *bt, qpg_int idx, qpg_int *nnzL, qpg_int *device_bytes);
/* The ordering of member idx's sparse factor, P H P' = L D L': perm[new] = old (n entries), and the height of its elimination tree.
 * The reference configures CHOLMOD_NATURAL (solver_interface.c:530-540: identity); context option "sparse_ordering" = 1 orders by
 * nested dissection instead, -1 (default) does so where the natural tree is deep (a band: one column per level) and dissection makes it
 * at least four times shallower -- the factorisation and the solves run at the latency of the tree's height.  Changes rounding, not
 * what is computed. */
int  qpg_batch_sparse_perm(qpg_batch *bt, qpg_int idx, qpg_int *perm, qpg_int *levels);
int  qpg_batch_update_settings(qpg_batch *bt, const QPGSettings *s);
int  qpg_batch_update_bounds(qpg_batch *bt, const qpg_float *bmin, const qpg_float *bmax); /* [B][m] or NULL */
int  qpg_batch_update_q(qpg_batch *bt, const qpg_float *q);                              /* [B][n] */
int  qpg_batch_get_info(qpg_batch *bt, qpg_int idx, QPGInfo *out);
int  qpg_batch_get_stats(qpg_batch *bt, qpg_int idx, QPGStats *out);
int  qpg_batch_get_info_all(qpg_batch *bt, QPGInfo *out /* [B] */);   /* QPALMInfo of every QP (what the multi-GPU gather sends) */
int  qpg_batch_get_stats_all(qpg_batch *bt, QPGStats *out /* [B] */);
int  qpg_batch_get_solution(qpg_batch *bt, qpg_float *x, qpg_float *y);                  /* [B][n], [B][m] */
int  qpg_batch_get_vector(qpg_batch *bt, const char *name, qpg_int idx, qpg_float *out, qpg_int len);
int  qpg_batch_set_vector(qpg_batch *bt, const char *name, qpg_int idx, const qpg_float *in, qpg_int len);
int  qpg_batch_get_ivector(qpg_batch *bt, const char *name, qpg_int idx, qpg_int *out, qpg_int len);
int  qpg_batch_set_ivector(qpg_batch *bt, const char *name, qpg_int idx, const qpg_int *in, qpg_int len);
int  qpg_batch_set_scalar(qpg_batch *bt, const char *name, qpg_int idx, qpg_float v);   /* "gamma", ... */
int  qpg_batch_get_factor(qpg_batch *bt, qpg_int idx, qpg_float *L_colmajor, qpg_float *D, qpg_int n);
void qpg_batch_destroy(qpg_batch *bt);
/* raw device pointers (for harnesses that keep data resident, e.g. torch tensors via from_dlpack) */
int  qpg_batch_device_ptr(qpg_batch *bt, const char *name, void **ptr, size_t *bytes);
int  qpg_batch_sync(qpg_batch *bt);

/* ---- solver_interface.h surface, acting on QP `idx` of the batch (device-resident state) ---- */
int qpg_mat_vec(qpg_batch *bt, qpg_int idx, int which /* 'A' or 'Q' */, const qpg_float *x, qpg_float *y);
int qpg_mat_tpose_vec(qpg_batch *bt, qpg_int idx, int which, const qpg_float *x, qpg_float *y);
int qpg_ldlchol(qpg_batch *bt, qpg_int idx);               /* factor Q (+ I/gamma if proximal) */
/* ldlchol(M, work, c) for an arbitrary symmetric M given as CSC (lower triangle read), solver_interface.h:172 */
int qpg_ldlchol_matrix(qpg_batch *bt, qpg_int idx, qpg_int n, const qpg_int *Mp, const qpg_int *Mi, const qpg_float *Mx);
/* mat_vec / mat_tpose_vec for a caller-owned CSC matrix (stype 0, or -1/+1 = symmetric, one triangle stored) */
int qpg_sparse_matvec(qpg_ctx *ctx, qpg_int nrow, qpg_int ncol, const qpg_int *Ap, const qpg_int *Ai, const qpg_float *Ax,
                      int stype, int transpose, const qpg_float *x, qpg_float *y);
int qpg_ldlcholQAtsigmaA(qpg_batch *bt, qpg_int idx);
int qpg_ldlupdate_entering_constraints(qpg_batch *bt, qpg_int idx);
int qpg_ldldowndate_leaving_constraints(qpg_batch *bt, qpg_int idx);
int qpg_ldlupdate_sigma_changed(qpg_batch *bt, qpg_int idx);
int qpg_ldlsolveLD_neg_dphi(qpg_batch *bt, qpg_int idx);
int qpg_compute_residuals(qpg_batch *bt, qpg_int idx);
int qpg_set_active_constraints(qpg_batch *bt, qpg_int idx);
int qpg_exact_linesearch(qpg_batch *bt, qpg_int idx, qpg_float *tau);
/* ---- the KKT operations of solver_interface.h (batches created with factorization_method = FACTORIZE_KKT; QPG_ERR_UNSUPPORTED
 * otherwise).  State as in the reference: active_constraints, enter / leave lists with nb_enter / nb_leave, sigma_inv, gamma,
 * dphi; the (n+m) x (n+m) panel and sol_kkt / rhs_kkt live on the device ("L", "Dfac", "sol_kkt", "kkt_state"). ---- */
/* qpalm_form_kkt / qpalm_reform_kkt (solver_interface.h:82,89; solver_interface.c:119-200): K = [[Q + I/gamma, A_a'], [A_a, -Sigma_a^-1]]
 * for the current active set into the slot (lower triangle; inactive constraints = unit diagonal), not factorised */
int qpg_kkt_form(qpg_batch *bt, qpg_int idx);
/* what newton.c:36,44 calls LADEL for after (re)forming: LDL' of the matrix in the slot, natural order (ladel_factorize_*_with_diag) */
int qpg_kkt_factorize(qpg_batch *bt, qpg_int idx);
/* kkt_update_entering_constraints (solver_interface.h:97; .c:202-218): ladel_row_add for enter[0 .. nb_enter) */
int qpg_kkt_update_entering_constraints(qpg_batch *bt, qpg_int idx);
/* kkt_update_leaving_constraints (solver_interface.h:106; .c:220-236): ladel_row_del for leave[0 .. nb_leave) */
int qpg_kkt_update_leaving_constraints(qpg_batch *bt, qpg_int idx);
/* kkt_solve (solver_interface.h:126; .c:238-247): sol_kkt = K^-1 [-dphi; 0], d = sol_kkt[0 .. n) (no iterative refinement: that is newton.c's) */
int qpg_kkt_solve(qpg_batch *bt, qpg_int idx);
/* batched LDL^T solve of the current factors with right-hand side dphi (the kernel the metric's
 * "HBM GB/s on LDL" refers to): every QP of the batch, `reps` times, for benchmarking. */
int qpg_batch_ldlsolve_all(qpg_batch *bt, qpg_int reps, float *ms_per_rep);
/* diagnostic (tools/sweep_probe.py): every resident workgroup factorises Q + I/gamma and applies `reps` times a rank-`nranks`
 * update + the matching downdate (rows 0 .. nranks-1 of A), i.e. the sweeps of solver_interface.c:407-441 alone under full-chip
 * contention; the sweep's phase timers are left in QPGStats.ms_dbg, *ms = duration of the launch */
int qpg_batch_sweep_probe(qpg_batch *bt, qpg_int reps, qpg_int nranks, float *ms);
/* attainable HBM bandwidth of this device, measured with a plain copy kernel (best of `reps` copies of `bytes` bytes;
 * read + write counted): the yardstick quoted next to the 8 TB/s spec figure */
int qpg_ctx_hbm_copy_gbs(qpg_ctx *ctx, size_t bytes, qpg_int reps, float *gbs);
int qpg_ctx_hbm_read_gbs(qpg_ctx *ctx, size_t bytes, qpg_int reps, float *gbs); /* read-only stream (the LDL' solve only reads L) */
/* page-locked (DMA-able) host memory for the arrays handed over every step (bounds in, solutions out); zero-filled.
 * The reference has no counterpart: it runs where its data is. */
int qpg_host_alloc(qpg_ctx *ctx, size_t bytes, void **out);
int qpg_host_free(qpg_ctx *ctx, void *p);

#ifdef __cplusplus
}
#endif
#endif
