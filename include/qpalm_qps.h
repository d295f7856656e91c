/*
 * qpalm_qps.h -- QPS (free-format MPS with QUADOBJ) reader of the host layer: the front-end of BASELINE.json's config 4
 * (Maros-Meszaros QPS set).  Mirrors what the reference's CLI does before it calls qpalm_setup
 * (interfaces/qps/src/qpalm_qps.c:71-537 sizes + data passes, :610-689 settings file, main :691-831):
 *
 *   sections  ROWS (N/L/G/E), COLUMNS (one or two entries per line), RHS (objective RHS -> c = -value, :394-395),
 *             RANGES (L: bmin = bmax - r, G: bmax = bmin + r, :440-472), BOUNDS (UP, LO, FX set values; FR removes the
 *             bound row, :179-190,487-495), QUADOBJ (lower triangle, column major, :497-536);
 *   variable bounds become identity rows appended to A (rows m - n_bounds .. m - 1), default [0, 1e20] (:298-302), the
 *   identity entry FIRST in its column (:316-324: columns of A are not sorted by row); |values| are clamped to 1e20.
 * Deliberate supersets (the reference silently drops them, which changes the QP): RANGES on E rows (MPS rule: r >= 0 ->
 * [rhs, rhs + r], r < 0 -> [rhs + r, rhs]) and bound types MI (lower = -1e20) and PL (upper = 1e20).  BV / LI / UI and
 * integrality markers are rejected with an error instead of being ignored.
 *   Old fixed-column format (the form the Maros-Meszaros files are distributed in; names may contain blanks): detected like the
 * reference does, by a ROWS line with a third token (qpalm_qps.c:104-108), and then read by column position -- fields 2-3, 5-12,
 * 15-22, 25-36, 40-47, 50-61 -- with the blanks squeezed out of the names, which is what the reference's conversion to
 * "<file>_copy.qps" does (interfaces/qps/src/qps_conversion.c:36-146) before it reads the copy; here no file is written.
 * The reference cuts the numeric fields at columns 29-37 / 53-61; this reader takes the whole fields and accepts Fortran
 * 'D' exponents.  A column whose entries are not contiguous and lines longer than the line buffer are errors.
 */
#ifndef QPALM_QPS_H
#define QPALM_QPS_H

#include "qpalm_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Reads `path` into a freshly allocated QPALMData (A: CSC m x n with the bound rows, stype 0; Q: lower CSC, stype -1).
 * Returns 0, or non-zero with a message in err (if err != NULL).  Free with qpalm_qps_free_data. */
int  qpalm_qps_read(const char *path, QPALMData **out, char *err, size_t errlen);
void qpalm_qps_free_data(QPALMData *data);
/* Settings file of the reference's CLI (:610-689): five header lines are skipped, then "name value" pairs on top of the
 * defaults.  Returns 0, or non-zero for an unreadable file / unknown setting. */
int  qpalm_qps_read_settings(const char *path, QPALMSettings *settings, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
