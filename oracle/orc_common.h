/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64, -ffp-contract=off) of the reference Fortran
 * algorithms on the MOM6 split-RK2 hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this library; the product
 * (mom6_amd/) never does.
 *
 * PARITY UNPINNED: the reference (Fortran + FMS + netCDF) cannot be built in this
 * image without writing stand-ins for the FMS library, and the reference's own
 * tests hold no golden vectors for any routine on this path (SURVEY.md section 4 /
 * 8c: every regression test is differential).  This oracle is therefore a
 * line-by-line restatement, citing reference file:line for every block, validated
 * by the reference's own differential invariants (tests/test_oracle_*.py):
 * conservation, rotation (ROTATE_INDEX-style), dimensional rescaling by powers of
 * two (bit-identical, as .testing dim.* requires), layout (1 vs 2 tiles).
 * Two exceptions ARE pinned to numbers the reference itself holds:
 *  - orc_remap.c (MOM_remapping / ALE remapping): the known answers of remapping_unit_tests
 *    (src/ALE/MOM_remapping.F90:2072-2943), replayed by tests/test_remap_cpu.py, which also holds its PLM, PPM_H4 and
 *    sub-grid integration to the REAL reference code compiled from the dependency-free files (oracle/_ref, `make ref`);
 *  - the equation-of-state functions used by the pressure force: the check values of EOS_unit_tests
 *    (MOM_EOS.F90:2077-2079 WRIGHT, :2129-2131 LINEAR) --
 *    tests/test_oracle_cpu.py::test_equation_of_state_against_reference_known_answers.
 *
 * Index conventions follow include/mom6x.h (local 0-based compute indices,
 * capital I/J = east/north face or vertex of cell i/j).
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/mom6x.h"

#define IX2(d, i, j) ((size_t)((i) + (d)->ioff) + (size_t)((j) + (d)->joff) * (size_t)(d)->pitch)
#define IX3(d, i, j, k) (IX2(d, i, j) + (size_t)(k) * (size_t)(d)->slab)
#define GM(g, d, m) ((g) + (size_t)(m) * (size_t)(d)->slab)

static inline double orc_max(double a, double b) { return (a > b) ? a : b; }
static inline double orc_min(double a, double b) { return (a < b) ? a : b; }
/* Fortran sign(a,b): |a| with the sign of b */
static inline double orc_sign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

/* A row-temporary indexed by local i in [-halo-1, ni+halo): allocate `pitch` doubles and
 * bias the pointer so that tmp[i] is valid. */
static inline double *orc_row_alloc(const mom6x_dims *d) {
  double *p = (double *)calloc((size_t)d->pitch, sizeof(double));
  return p + d->ioff;
}
static inline void orc_row_free(const mom6x_dims *d, double *p) { free(p - d->ioff); }

#endif
