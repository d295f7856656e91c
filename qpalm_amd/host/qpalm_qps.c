/*
 * qpalm_qps.c -- QPS reader (include/qpalm_qps.h): host C, no numerics.  Own two-pass implementation (token scanner +
 * open-addressing name table); the behaviours it reproduces are cited in the header.
 */
#include "../../include/qpalm_qps.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QPS_INF 1e20 /* QPALM_INFTY, include/constants.h */

typedef struct { char *key; long val; char sign; } qps_slot;
typedef struct { qps_slot *slots; size_t cap, count; } qps_map;

static unsigned long qps_hash(const char *s) { unsigned long h = 1469598103934665603ul; while (*s) { h ^= (unsigned char)*s++; h *= 1099511628211ul; } return h; }
static void map_init(qps_map *m, size_t expect) { m->cap = 16; while (m->cap < 2 * expect + 8) m->cap <<= 1; m->slots = (qps_slot *)calloc(m->cap, sizeof(qps_slot)); m->count = 0; }
static void map_free(qps_map *m) { if (!m->slots) return; for (size_t i = 0; i < m->cap; i++) free(m->slots[i].key); free(m->slots); m->slots = NULL; }
static qps_slot *map_find(const qps_map *m, const char *key) {
  size_t i = qps_hash(key) & (m->cap - 1);
  while (m->slots[i].key) { if (!strcmp(m->slots[i].key, key)) return &m->slots[i]; i = (i + 1) & (m->cap - 1); }
  return NULL;
}
static void map_put(qps_map *m, const char *key, long val, char sign) {
  if (2 * (m->count + 1) > m->cap) { /* grow */
    qps_map big; big.cap = m->cap * 2; big.slots = (qps_slot *)calloc(big.cap, sizeof(qps_slot)); big.count = 0;
    for (size_t i = 0; i < m->cap; i++) if (m->slots[i].key) {
      size_t j = qps_hash(m->slots[i].key) & (big.cap - 1);
      while (big.slots[j].key) j = (j + 1) & (big.cap - 1);
      big.slots[j] = m->slots[i]; big.count++;
    }
    free(m->slots); *m = big;
  }
  size_t i = qps_hash(key) & (m->cap - 1);
  while (m->slots[i].key) i = (i + 1) & (m->cap - 1);
  m->slots[i].key = strdup(key); m->slots[i].val = val; m->slots[i].sign = sign; m->count++;
}

static int fail(char *err, size_t errlen, const char *fmt, const char *a, long line) {
  if (err && errlen) snprintf(err, errlen, fmt, a, line);
  return 1;
}
static double clamp_inf(double v) { return v > QPS_INF ? QPS_INF : (v < -QPS_INF ? -QPS_INF : v); }

enum { SEC_NONE, SEC_ROWS, SEC_COLUMNS, SEC_RHS, SEC_RANGES, SEC_BOUNDS, SEC_QUADOBJ, SEC_END };
static int section_of(const char *tok) {
  if (!strcmp(tok, "ROWS")) return SEC_ROWS;
  if (!strcmp(tok, "COLUMNS")) return SEC_COLUMNS;
  if (!strcmp(tok, "RHS")) return SEC_RHS;
  if (!strcmp(tok, "RANGES")) return SEC_RANGES;
  if (!strcmp(tok, "BOUNDS")) return SEC_BOUNDS;
  if (!strcmp(tok, "QUADOBJ")) return SEC_QUADOBJ;
  if (!strcmp(tok, "ENDATA")) return SEC_END;
  return SEC_NONE;
}
/* splits a data line into at most 6 tokens (in place) */
static int split(char *line, char *tok[6]) {
  int n = 0;
  char *p = line;
  while (*p && n < 6) {
    while (*p && isspace((unsigned char)*p)) p++;
    if (!*p) break;
    tok[n++] = p;
    while (*p && !isspace((unsigned char)*p)) p++;
    if (*p) *p++ = 0;
  }
  return n;
}
static int is_number(const char *s) { char *e; strtod(s, &e); return e != s && *e == 0; }

/* ---- the old fixed-column format (names may contain blanks).  The reference detects it by a ROWS line with a third token and
 * first rewrites the file to "<name>_copy.qps" with the blanks squeezed out of the names (interfaces/qps/src/qps_conversion.c:36-146,
 * called from qpalm_qps.c:104-108,731-741); here the same conversion is done line by line in memory (no file is written).  Fields
 * by column (1-based, MPS standard): 2-3 type, 5-12 name, 15-22 name, 25-36 number, 40-47 name, 50-61 number.  The reference
 * cuts the number fields at columns 29-37 / 53-61 (c_strcpy_limit(temp1, &line[27], 10) after one consumed character), which
 * drops leading digits of numbers that start at column 25; this reader takes the whole field.  A Fortran 'D' exponent is
 * accepted as 'E'.  out[] receives the squeezed fields in the order of the free format; returns their number. */
static int field(const char *line, size_t len, size_t c0, size_t c1, char *dst) { /* columns c0..c1 (1-based, inclusive), blanks removed */
  size_t n = 0;
  for (size_t c = c0; c <= c1 && c <= len; c++) {
    const char ch = line[c - 1];
    if (ch == '\n' || ch == '\r') break;
    if (!isspace((unsigned char)ch)) dst[n++] = ch;
  }
  dst[n] = 0;
  return n > 0;
}
static void fortran_exponent(char *s) { for (; *s; s++) if (*s == 'D' || *s == 'd') *s = 'E'; }
static int split_fixed(const char *line, int sec, char buf[6][40], char *tok[6]) {
  const size_t len = strlen(line);
  int n = 0;
  char f1[40], f2[40], f3[40], f4[40], f5[40], f6[40];
  const int h1 = field(line, len, 2, 3, f1), h2 = field(line, len, 5, 12, f2), h3 = field(line, len, 15, 22, f3), h4 = field(line, len, 25, 36, f4),
            h5 = field(line, len, 40, 47, f5), h6 = field(line, len, 50, 61, f6);
  fortran_exponent(f4); fortran_exponent(f6);
#define QPS_PUT(f) do { snprintf(buf[n], 40, "%s", f); tok[n] = buf[n]; n++; } while (0)
  if (sec == SEC_ROWS) {
    if (h1) QPS_PUT(f1);
    if (h2) QPS_PUT(f2);
  } else if (sec == SEC_BOUNDS) {
    if (h1) QPS_PUT(f1);
    if (h2) QPS_PUT(f2);
    if (h3) QPS_PUT(f3);
    if (h4) QPS_PUT(f4);
  }
  else { /* COLUMNS, RHS, RANGES, QUADOBJ: name name number [name number] */
    if (h2) QPS_PUT(f2);
    if (h3) QPS_PUT(f3);
    if (h4) QPS_PUT(f4);
    if (h5 && h6) { QPS_PUT(f5); QPS_PUT(f6); }
  }
#undef QPS_PUT
  return n;
}

typedef struct { long col; long row; double v; } qps_entry;

int qpalm_qps_read(const char *path, QPALMData **out, char *err, size_t errlen) {
  if (!path || !out) return fail(err, errlen, "qpalm_qps_read: NULL argument%s%ld", "", 0);
  FILE *fp = fopen(path, "r");
  if (!fp) return fail(err, errlen, "Could not open file %s%.0ld", path, 0);
  char line[512], objective[128] = "";
  char *tok[6];
  long lineno = 0;
  int sec = SEC_NONE, rc = 0;
  /* ---- pass 1: rows, columns (names, counts), free bounds ---------------------------------------------------- */
  qps_map rows, cols, freeb;
  map_init(&rows, 64); map_init(&cols, 64); map_init(&freeb, 16);
  long m_rows = 0, n = 0, nnzA = 0, nnzQ = 0;
  char prev_col[128] = "";
  char fbuf[6][40];
  int seen_name = 0, fixed = 0;
pass1:
  while (fgets(line, sizeof line, fp)) {
    lineno++;
    if (line[0] == '*' || line[0] == '\n' || line[0] == '\r') continue;
    if (!strchr(line, '\n') && !feof(fp)) { rc = fail(err, errlen, "Line too long in %s at line %ld", path, lineno); goto done1; }
    if (!isspace((unsigned char)line[0])) { /* section header */
      char head[64] = "";
      sscanf(line, "%63s", head);
      if (!strcmp(head, "NAME")) { seen_name = 1; continue; }
      sec = section_of(head);
      if (sec == SEC_NONE) { rc = fail(err, errlen, "Unknown section '%s' at line %ld", head, lineno); goto done1; }
      if (sec == SEC_END) break;
      continue;
    }
    int nt = fixed ? split_fixed(line, sec, fbuf, tok) : split(line, tok);
    if (!nt) continue;
    if (sec == SEC_ROWS) {
      if (nt != 2) {
        if (fixed) { rc = fail(err, errlen, "Malformed ROWS line in %s at line %ld", path, lineno); goto done1; }
        /* a third token = a name with a blank: the old fixed-column format (qpalm_qps.c:104-108); start over by columns */
        fixed = 1;
        map_free(&rows); map_free(&cols); map_free(&freeb);
        map_init(&rows, 64); map_init(&cols, 64); map_init(&freeb, 16);
        m_rows = 0; n = 0; nnzA = 0; nnzQ = 0; prev_col[0] = 0; objective[0] = 0; seen_name = 0; sec = SEC_NONE; lineno = 0;
        rewind(fp);
        goto pass1;
      }
      const char sgn = tok[0][0];
      if (sgn == 'N') { if (!objective[0]) snprintf(objective, sizeof objective, "%s", tok[1]); }
      else if (sgn == 'L' || sgn == 'G' || sgn == 'E') map_put(&rows, tok[1], m_rows++, sgn);
      else { rc = fail(err, errlen, "Unknown row type '%s' at line %ld", tok[0], lineno); goto done1; }
    } else if (sec == SEC_COLUMNS) {
      if (nt >= 3 && !strcmp(tok[1], "'MARKER'")) { rc = fail(err, errlen, "Integrality markers are not supported (%s line %ld)", path, lineno); goto done1; }
      if (nt != 3 && nt != 5) { rc = fail(err, errlen, "Malformed COLUMNS line in %s at line %ld", path, lineno); goto done1; }
      if (strcmp(tok[0], prev_col)) {
        /* the entries of a column must be contiguous: a name that comes back later would be counted twice */
        if (map_find(&cols, tok[0])) { rc = fail(err, errlen, "Column '%s' appears in two places (line %ld)", tok[0], lineno); goto done1; }
        map_put(&cols, tok[0], n++, ' '); snprintf(prev_col, sizeof prev_col, "%s", tok[0]);
      }
      for (int k = 1; k + 1 < nt; k += 2) if (strcmp(tok[k], objective)) nnzA++;
    } else if (sec == SEC_BOUNDS) {
      if (nt >= 2 && !strcmp(tok[0], "FR")) { const char *cname = tok[nt - 1]; if (!map_find(&freeb, cname)) map_put(&freeb, cname, 0, ' '); }
    } else if (sec == SEC_QUADOBJ) nnzQ++;
  }
done1:
  if (!rc && !seen_name) rc = fail(err, errlen, "Wrong file format. Expected first line to contain NAME problem_name (%s)%.0ld", path, 0);
  if (rc) { fclose(fp); map_free(&rows); map_free(&cols); map_free(&freeb); return rc; }
  const long n_bounds = n - (long)freeb.count, m = m_rows + n_bounds;
  QPALMData *d = (QPALMData *)calloc(1, sizeof(QPALMData));
  d->n = (size_t)n; d->m = (size_t)m; d->c = 0;
  d->q = (c_float *)calloc((size_t)(n ? n : 1), sizeof(c_float));
  d->bmin = (c_float *)calloc((size_t)(m ? m : 1), sizeof(c_float));
  d->bmax = (c_float *)calloc((size_t)(m ? m : 1), sizeof(c_float));
  d->A = qpalm_sparse_alloc((size_t)m, (size_t)n, (size_t)(nnzA + n_bounds + 1), 0);
  d->Q = qpalm_sparse_alloc((size_t)n, (size_t)n, (size_t)(nnzQ + 1), -1);
  c_int *Ap = (c_int *)d->A->p, *Ai = (c_int *)d->A->i, *Qp = (c_int *)d->Q->p, *Qi = (c_int *)d->Q->i;
  c_float *Ax = (c_float *)d->A->x, *Qx = (c_float *)d->Q->x;
  /* bound row of every variable (-1: free), in column order */
  long *brow = (long *)malloc((size_t)(n ? n : 1) * sizeof(long));
  {
    long next = m_rows;
    /* cols map values are the column indices; walk names through the map to test free bounds */
    for (size_t i = 0; i < cols.cap; i++) if (cols.slots[i].key) brow[cols.slots[i].val] = map_find(&freeb, cols.slots[i].key) ? -1 : 0;
    for (long j = 0; j < n; j++) if (brow[j] == 0) brow[j] = next++;
  }
  for (size_t i = 0; i < rows.cap; i++) if (rows.slots[i].key) { /* ROWS defaults (:283-296) */
    const long r = rows.slots[i].val;
    switch (rows.slots[i].sign) {
      case 'L': d->bmax[r] = 0; d->bmin[r] = -QPS_INF; break;
      case 'G': d->bmin[r] = 0; d->bmax[r] = QPS_INF; break;
      default: d->bmin[r] = 0; d->bmax[r] = 0; break;
    }
  }
  for (long k = m_rows; k < m; k++) { d->bmin[k] = 0; d->bmax[k] = QPS_INF; } /* default variable bounds [0, 1e20] (:298-302) */
  /* ---- pass 2: data ------------------------------------------------------------------------------------------ */
  rewind(fp);
  lineno = 0; sec = SEC_NONE;
  long cur_col = -1, elemA = 0, elemQ = 0, qcol_prev = 0;
  for (long j = 0; j <= n; j++) Ap[j] = 0;
  for (long j = 0; j <= n; j++) Qp[j] = 0;
  while (fgets(line, sizeof line, fp)) {
    lineno++;
    if (line[0] == '*' || line[0] == '\n' || line[0] == '\r') continue;
    if (!isspace((unsigned char)line[0])) {
      char head[64] = "";
      sscanf(line, "%63s", head);
      if (!strcmp(head, "NAME")) continue;
      sec = section_of(head);
      if (sec == SEC_END) break;
      continue;
    }
    int nt = fixed ? split_fixed(line, sec, fbuf, tok) : split(line, tok);
    if (!nt) continue;
    if (sec == SEC_COLUMNS) {
      const long col = map_find(&cols, tok[0])->val;
      if (col != cur_col) { /* a new column starts with its identity (bound) entry; the device layer sorts every column by row anyway
                             * (the reference's reader leaves the columns of A unsorted) */
        cur_col = col;
        Ap[col] = elemA;
        if (brow[col] >= 0) { Ai[elemA] = brow[col]; Ax[elemA] = 1; elemA++; }
      }
      for (int k = 1; k + 1 < nt; k += 2) {
        if (!is_number(tok[k + 1])) { rc = fail(err, errlen, "Bad number '%s' at line %ld", tok[k + 1], lineno); goto done2; }
        const double v = strtod(tok[k + 1], NULL);
        if (!strcmp(tok[k], objective)) d->q[col] = v;
        else {
          qps_slot *r = map_find(&rows, tok[k]);
          if (!r) { rc = fail(err, errlen, "Unknown row '%s' at line %ld", tok[k], lineno); goto done2; }
          Ai[elemA] = r->val; Ax[elemA] = clamp_inf(v); elemA++;
        }
      }
      Ap[col + 1] = elemA;
    } else if (sec == SEC_RHS || sec == SEC_RANGES) {
      /* "[setname] row value [row value]": the set name is optional (:143-150) */
      int k0 = (nt == 3 || nt == 5) ? 1 : 0;
      if (nt - k0 != 2 && nt - k0 != 4) { rc = fail(err, errlen, "Malformed RHS/RANGES line in %s at line %ld", path, lineno); goto done2; }
      for (int k = k0; k + 1 < nt; k += 2) {
        if (!is_number(tok[k + 1])) { rc = fail(err, errlen, "Bad number '%s' at line %ld", tok[k + 1], lineno); goto done2; }
        const double v = strtod(tok[k + 1], NULL);
        if (sec == SEC_RHS && !strcmp(tok[k], objective)) { d->c = -v; continue; }
        qps_slot *r = map_find(&rows, tok[k]);
        if (!r) { rc = fail(err, errlen, "Unknown row '%s' at line %ld", tok[k], lineno); goto done2; }
        const long row = r->val;
        if (sec == SEC_RHS) {
          switch (r->sign) {
            case 'L': d->bmax[row] = v; d->bmin[row] = -QPS_INF; break;
            case 'G': d->bmin[row] = v; break;
            default: d->bmin[row] = v; d->bmax[row] = v; break;
          }
        } else {
          switch (r->sign) {
            case 'L': d->bmin[row] = d->bmax[row] - v; break;
            case 'G': d->bmax[row] = d->bmin[row] + v; break;
            default: if (v >= 0) d->bmax[row] = d->bmin[row] + v; else d->bmin[row] = d->bmax[row] + v; break; /* MPS rule for E rows */
          }
        }
      }
    } else if (sec == SEC_BOUNDS) {
      /* "type [setname] column [value]" */
      const char *type = tok[0];
      if (strcmp(type, "UP") && strcmp(type, "LO") && strcmp(type, "FX") && strcmp(type, "FR") && strcmp(type, "MI") && strcmp(type, "PL")) {
        /* BV / LI / UI (integer variables) and anything else: the reference silently keeps the default bound (qpalm_qps.c:487-495),
         * i.e. solves another problem; refused here */
        rc = fail(err, errlen, "Unsupported bound type '%s' at line %ld (integer bounds BV/LI/UI are not part of a QP)", type, lineno); goto done2;
      }
      const int has_val = strcmp(type, "FR") && strcmp(type, "MI") && strcmp(type, "PL");
      const int need = has_val ? 3 : 2;
      if (nt != need && nt != need + 1) { rc = fail(err, errlen, "Malformed BOUNDS line in %s at line %ld", path, lineno); goto done2; }
      const char *cname = has_val ? tok[nt - 2] : tok[nt - 1];
      qps_slot *cslot = map_find(&cols, cname);
      if (!cslot) { rc = fail(err, errlen, "Unknown column '%s' at line %ld", cname, lineno); goto done2; }
      const long br = brow[cslot->val];
      if (!strcmp(type, "FR")) continue;
      if (br < 0) { rc = fail(err, errlen, "Bound on the free variable '%s' at line %ld", cname, lineno); goto done2; }
      double v = 0;
      if (has_val) { if (!is_number(tok[nt - 1])) { rc = fail(err, errlen, "Bad number '%s' at line %ld", tok[nt - 1], lineno); goto done2; } v = strtod(tok[nt - 1], NULL); }
      if (!strcmp(type, "UP")) d->bmax[br] = v;
      else if (!strcmp(type, "LO")) d->bmin[br] = v;
      else if (!strcmp(type, "FX")) { d->bmin[br] = v; d->bmax[br] = v; }
      else if (!strcmp(type, "MI")) d->bmin[br] = -QPS_INF;
      else if (!strcmp(type, "PL")) d->bmax[br] = QPS_INF;
    } else if (sec == SEC_QUADOBJ) {
      if (nt != 3 || !is_number(tok[2])) { rc = fail(err, errlen, "Malformed QUADOBJ line in %s at line %ld", path, lineno); goto done2; }
      qps_slot *c1 = map_find(&cols, tok[0]), *c2 = map_find(&cols, tok[1]);
      if (!c1 || !c2) { rc = fail(err, errlen, "Unknown column in QUADOBJ '%s' at line %ld", tok[0], lineno); goto done2; }
      const long col = c1->val, row = c2->val;
      if (col < qcol_prev) { rc = fail(err, errlen, "QUADOBJ columns out of order in %s at line %ld", path, lineno); goto done2; }
      for (; qcol_prev < col; qcol_prev++) Qp[qcol_prev + 1] = elemQ;
      Qi[elemQ] = row; Qx[elemQ] = clamp_inf(strtod(tok[2], NULL)); elemQ++;
      Qp[col + 1] = elemQ;
    }
  }
  /* columns that never appeared between others keep empty ranges; close the pointer arrays */
  for (long j = 1; j <= n; j++) if (Ap[j] < Ap[j - 1]) Ap[j] = Ap[j - 1];
  for (; qcol_prev < n; qcol_prev++) Qp[qcol_prev + 1] = elemQ;
  for (long j = 1; j <= n; j++) if (Qp[j] < Qp[j - 1]) Qp[j] = Qp[j - 1];
done2:
  fclose(fp);
  free(brow);
  map_free(&rows); map_free(&cols); map_free(&freeb);
  if (rc) { qpalm_qps_free_data(d); return rc; }
  *out = d;
  return 0;
}

void qpalm_qps_free_data(QPALMData *d) {
  if (!d) return;
  qpalm_sparse_free(&d->A); qpalm_sparse_free(&d->Q);
  free(d->q); free(d->bmin); free(d->bmax); free(d);
}

int qpalm_qps_read_settings(const char *path, QPALMSettings *s, char *err, size_t errlen) {
  qpalm_set_default_settings(s);
  FILE *fp = fopen(path, "r");
  if (!fp) return fail(err, errlen, "Could not open file %s%.0ld", path, 0);
  char line[256], name[128];
  for (int i = 0; i < 5; i++) if (!fgets(line, sizeof line, fp)) break; /* five header lines (:617-619) */
  double v;
  int rc = 0;
  while (fscanf(fp, "%127s %le", name, &v) == 2) {
#define QS_I(f) else if (!strcmp(name, #f)) s->f = (c_int)v
#define QS_F(f) else if (!strcmp(name, #f)) s->f = v
    if (0) {}
    QS_I(max_iter); QS_I(inner_max_iter); QS_F(eps_abs); QS_F(eps_rel); QS_F(eps_abs_in); QS_F(eps_rel_in); QS_F(rho);
    QS_F(eps_prim_inf); QS_F(eps_dual_inf); QS_F(theta); QS_F(delta); QS_F(sigma_max); QS_F(sigma_init); QS_I(proximal);
    QS_F(gamma_init); QS_F(gamma_upd); QS_F(gamma_max); QS_I(scaling); QS_I(nonconvex); QS_I(verbose); QS_I(print_iter);
    QS_I(warm_start); QS_I(reset_newton_iter); QS_I(enable_dual_termination); QS_F(dual_objective_limit); QS_F(time_limit);
    QS_I(ordering); QS_I(factorization_method); QS_I(max_rank_update); QS_F(max_rank_update_fraction);
    else { rc = fail(err, errlen, "Unrecognised setting: %s%.0ld", name, 0); break; }
#undef QS_I
#undef QS_F
  }
  fclose(fp);
  return rc;
}

#ifdef QPALM_QPS_MAIN
/* the reference's CLI (main :691-831): qpalm_qps problem.qps [settings.txt] */
int main(int argc, char *argv[]) {
  if (argc != 2 && argc != 3) { fprintf(stderr, "Wrong number of arguments. Correct usage is qpalm_qps problem.qps or qpalm_qps problem.qps settings.txt.\n"); return 1; }
  char err[256];
  QPALMData *data = NULL;
  if (qpalm_qps_read(argv[1], &data, err, sizeof err)) { fprintf(stderr, "%s\n", err); return 1; }
  printf("Reading successful.\n");
  QPALMSettings settings;
  if (argc == 3) { if (qpalm_qps_read_settings(argv[2], &settings, err, sizeof err)) { printf("%s\nUsing default settings instead\n", err); qpalm_set_default_settings(&settings); } }
  else qpalm_set_default_settings(&settings);
  QPALMWorkspace *work = qpalm_setup(data, &settings);
  if (!work) { fprintf(stderr, "qpalm_setup failed: %s\n", qpalm_backend_error()); qpalm_qps_free_data(data); return 1; }
  qpalm_solve(work);
  printf("Iter: %ld\n", (long)work->info->iter);
  printf("Runtime: %f seconds\n", work->info->run_time);
  printf("Status: %s, objective %.10e\n", work->info->status, work->info->objective);
  qpalm_cleanup(work);
  qpalm_qps_free_data(data);
  return 0;
}
#endif
