/*
 * cpu_bench.c -- TEST INFRASTRUCTURE: times the CPU oracle (qpalm_oracle.c) on a file of QPs with plain pthreads.
 *
 * This is bench.py's cpu_baseline leg (kind "port"): one oracle workspace per thread at a time, no Python in the loop.
 * Every QP is timed the way the reference times itself: info.run_time = setup_time + solve_time
 * (/root/reference/src/qpalm.c:493-495,721-723), here clock_gettime(CLOCK_MONOTONIC) around oq_setup + oq_solve;
 * oq_cleanup is outside the per-QP figure but inside the wall time the throughput is computed from.
 *
 *   cpu_bench <problems.bin> <threads> [passes] [sparse]      (sparse = 1: the oracle's sparse-storage mode, natural ordering: what the
 *                                                              reference hands to CHOLMOD for a sparse Schur complement)
 *
 * File layout (little endian, written by bench.py): int64 magic 0x5150424e, int64 count, oq_settings, then per QP
 *   int64 n, m, nnzQ, nnzA; Qp[n+1] Qi[nnzQ] (int64) Qx[nnzQ] (double); Ap[n+1] Ai[nnzA] Ax[nnzA]; q[n]; c; bmin[m]; bmax[m].
 * Prints one JSON object.  Never linked into or called by the product.
 */
#define _DEFAULT_SOURCE
#include <malloc.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qpalm_oracle.h"

typedef struct {
  oq_int n, m, nnzQ, nnzA;
  oq_int *Qp, *Qi, *Ap, *Ai;
  oq_float *Qx, *Ax, *q, *bmin, *bmax;
  oq_float c;
} problem;

static problem *g_prob;
static oq_int g_count, g_total;
static oq_settings g_settings;
static volatile oq_int g_next;
static int g_sparse;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static double g_sum_run, g_sum_solve;
static oq_int g_solved, g_iters;

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void *worker(void *arg) {
  (void)arg;
  double sum_run = 0.0, sum_solve = 0.0;
  oq_int solved = 0, iters = 0;
  for (;;) {
    pthread_mutex_lock(&g_lock);
    const oq_int k = g_next++;
    pthread_mutex_unlock(&g_lock);
    if (k >= g_total) break;
    const problem *p = &g_prob[k % g_count];
    const double t0 = now_s();
    oq_workspace *w = oq_setup(p->n, p->m, p->Qp, p->Qi, p->Qx, p->Ap, p->Ai, p->Ax, p->q, p->c, p->bmin, p->bmax, &g_settings);
    const double t1 = now_s();
    if (!w) continue;
    if (g_sparse) oq_set_scalar(w, "sparse_mode", 1);
    else oq_set_scalar(w, "updown_block", 8); /* up to eight ranks per pass over L, as cholmod_updown does (bit-identical to the rank-1 sweeps the parity tests use) */
    oq_solve(w);
    const double t2 = now_s();
    const oq_info *info = oq_get_info(w);
    if (info->status_val == OQ_SOLVED) solved++;
    iters += info->iter;
    sum_run += t2 - t0;
    sum_solve += t2 - t1;
    oq_cleanup(w);
  }
  pthread_mutex_lock(&g_lock);
  g_sum_run += sum_run; g_sum_solve += sum_solve; g_solved += solved; g_iters += iters;
  pthread_mutex_unlock(&g_lock);
  return NULL;
}

static int rd(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : 1; }

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: cpu_bench problems.bin threads [passes]\n"); return 2; }
  const int threads = atoi(argv[2]) > 0 ? atoi(argv[2]) : 1;
  const int passes = (argc > 3 && atoi(argv[3]) > 0) ? atoi(argv[3]) : 1;
  g_sparse = (argc > 4 && atoi(argv[4]) > 0) ? 1 : 0;
  /* keep the workspaces of successive QPs on the heap of the thread's arena: with the default thresholds every setup / cleanup
   * pair maps and unmaps its multi-megabyte arrays, and the threads then queue on the process's address-space lock */
  mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int64_t magic = 0, count = 0;
  if (rd(f, &magic, 8) || rd(f, &count, 8) || magic != 0x5150424e || count <= 0 || rd(f, &g_settings, sizeof g_settings)) {
    fprintf(stderr, "cpu_bench: bad header\n");
    return 2;
  }
  g_prob = (problem *)calloc((size_t)count, sizeof(problem));
  for (int64_t k = 0; k < count; k++) {
    problem *p = &g_prob[k];
    int64_t h[4];
    if (rd(f, h, sizeof h)) { fprintf(stderr, "cpu_bench: truncated file\n"); return 2; }
    p->n = h[0]; p->m = h[1]; p->nnzQ = h[2]; p->nnzA = h[3];
    p->Qp = (oq_int *)malloc((size_t)(p->n + 1) * 8); p->Qi = (oq_int *)malloc((size_t)(p->nnzQ + 1) * 8); p->Qx = (oq_float *)malloc((size_t)(p->nnzQ + 1) * 8);
    p->Ap = (oq_int *)malloc((size_t)(p->n + 1) * 8); p->Ai = (oq_int *)malloc((size_t)(p->nnzA + 1) * 8); p->Ax = (oq_float *)malloc((size_t)(p->nnzA + 1) * 8);
    p->q = (oq_float *)malloc((size_t)(p->n + 1) * 8); p->bmin = (oq_float *)malloc((size_t)(p->m + 1) * 8); p->bmax = (oq_float *)malloc((size_t)(p->m + 1) * 8);
    if (rd(f, p->Qp, (size_t)(p->n + 1) * 8) || rd(f, p->Qi, (size_t)p->nnzQ * 8) || rd(f, p->Qx, (size_t)p->nnzQ * 8) ||
        rd(f, p->Ap, (size_t)(p->n + 1) * 8) || rd(f, p->Ai, (size_t)p->nnzA * 8) || rd(f, p->Ax, (size_t)p->nnzA * 8) ||
        rd(f, p->q, (size_t)p->n * 8) || rd(f, &p->c, 8) || rd(f, p->bmin, (size_t)p->m * 8) || rd(f, p->bmax, (size_t)p->m * 8)) {
      fprintf(stderr, "cpu_bench: truncated file\n");
      return 2;
    }
  }
  fclose(f);
  g_count = count; g_total = count * passes; g_next = 0;
  pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
  const double t0 = now_s();
  for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, NULL);
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  const double wall = now_s() - t0;
  printf("{\"qps\": %.6f, \"wall_s\": %.6f, \"threads\": %d, \"count\": %lld, \"solved\": %lld, \"iter_mean\": %.3f, "
         "\"setup_plus_solve_s_per_qp\": %.6e, \"solve_s_per_qp\": %.6e}\n",
         (double)g_total / wall, wall, threads, (long long)g_total, (long long)g_solved, (double)g_iters / (double)g_total,
         g_sum_run / (double)g_total, g_sum_solve / (double)g_total);
  return g_solved == g_total ? 0 : 1;
}
