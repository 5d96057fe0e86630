// remap.hip -- MOM_remapping's remapping_core_h (OM4-era reconstructions PCM / PLM / PPM_H4, answer dates >= 20190101)
// and the remapping half of MOM_ALE on gfx950.
//
//   remapping_core_h :234, build_reconstructions_1d :410, intersect_src_tgt_grids :642, remap_src_to_sub_grid_om4 :845,
//   remap_src_to_sub_grid :962, remap_sub_to_tgt_grid_om4 :1103, average_value_ppoly :1391   (src/ALE/MOM_remapping.F90)
//   PLM_functions.F90, PPM_functions.F90, regrid_edge_values.F90 (bound_edge_values, check_discontinuous_edge_values,
//   edge_values_explicit_h4, end_value_h4), ALE_remap_tracers :760, ALE_remap_set_h_vel :882, ALE_remap_velocities :1089
//
// One thread per column, two kernels:
//  k_remap_recon   the reconstruction (edge values, PLM slope) of every source cell in ONE sweep over k with a register
//                  window (the reference makes a pass per stage); same k in every lane, so the column arrays (3-D scratch
//                  fields, [k][column]) are read once and written once, coalesced.
//  k_remap_apply   the reference builds the n0+n1+1 sub-cells of the two grids' intersection in arrays, integrates the
//                  reconstruction over each, corrects the thickest sub-cell of every source cell so that the cell's
//                  integral is conserved to the last bit, and sums the sub-cells of each target cell.  Per-thread arrays
//                  of that size would live in scratch memory; here the merge is STREAMED instead: the sub-cells of one
//                  source cell are replayed three times from a saved merge state (1: effective width and the thickest
//                  sub-cell, 2: the sum of the other sub-cells' integrals, 3: emission into the running target sums),
//                  so a thread carries O(1) state.  Every quantity is formed by the reference's expression in the
//                  reference's order, so the results are bit-identical to orc_remap.c (which keeps the arrays).
#include "mom6x_dev.h"
#include "eos_dev.h"
#include <cfloat>
#include <cstring>
#include <vector>

namespace {

struct View { size_t base, lev; };                       // element k (1-based) of a column: p[base + (k-1)*lev]
#define AT(p, v, k) (p)[(v).base + (size_t)((k) - 1) * (v).lev]

__device__ __forceinline__ double fsign(double a, double b) { return copysign(fabs(a), b); }   // Fortran sign()
__device__ __forceinline__ double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
__device__ __forceinline__ double dmin3(double a, double b, double c) { return dmin(dmin(a, b), c); }

enum { INTEGRATION_PCM = 0, INTEGRATION_PLM = 1, INTEGRATION_PPM = 3 };

// ---- PLM_functions.F90 --------------------------------------------------------------------------------------------
__device__ double PLM_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r) {   // :27-66
  const double sigma_r = u_r - u_c, sigma_l = u_c - u_l;
  const double sigma_c = 2.0 * (u_r - u_l) * (h_c / (h_l + 2.0 * h_c + h_r + h_neglect));
  const double u_min = dmin3(u_l, u_c, u_r), u_max = dmax3(u_l, u_c, u_r);
  double s;
  if ((sigma_l * sigma_r) > 0.0) s = fsign(dmin(fabs(sigma_c), 2. * dmin(u_c - u_min, u_max - u_c)), sigma_c);
  else s = 0.0;
  if (u_c - 0.5 * fabs(s) < u_min || u_c + 0.5 * fabs(s) > u_max) s = s * (1. - DBL_EPSILON);
  if (fabs(s) < 1.E-140) s = 0.;
  return s;
}
__device__ double PLM_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r) {   // :124-162
  const double almost_two = 2. * (1. - DBL_EPSILON);
  const double e_r = u_l + 0.5 * s_l, e_l = u_r - 0.5 * s_r;
  double slp = fabs(s_c);
  double edge = u_c - 0.5 * s_c;
  if ((edge - e_r) * (u_c - edge) < 0.) { edge = 0.5 * (edge + e_r); slp = dmin(slp, fabs(edge - u_c) * almost_two); }
  edge = u_c + 0.5 * s_c;
  if ((edge - u_c) * (e_l - edge) < 0.) { edge = 0.5 * (edge + e_l); slp = dmin(slp, fabs(edge - u_c) * almost_two); }
  return fsign(slp, s_c);
}
__device__ double PLM_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c) {   // :170-191
  const double hl = h_l + h_neglect, hc = h_c + h_neglect;
  const double left_edge = (u_l * hc + u_c * hl) / (hl + hc);
  return 2.0 * (u_c - left_edge);
}

// ---- regrid_edge_values.F90: end_value_h4 :634-747 ------------------------------------------------------------------
__device__ void end_value_h4(const double *dz, const double *u, double *Csys) {
  const double min_frac = 1.0e-6;
  double h1 = dz[0], h2 = dz[1], h3 = dz[2], h4 = dz[3];
  if ((h2 + h3) < min_frac * h1) h3 = min_frac * h1 - h2;
  if ((h3 + h4) < min_frac * h1) h4 = min_frac * h1 - h3;
  const double h12 = h1 + h2, h23 = h2 + h3, h34 = h3 + h4;
  const double h123 = h12 + h3, h234 = h2 + h34, h1234 = h12 + h34;
  const double I_denB3 = 1.0 / (h123 * h12 * h23);
  const double I_h12 = (h123 * h23) * I_denB3;
  const double I_h23 = (h12 * h123) * I_denB3;
  const double I_h123 = (h12 * h23) * I_denB3;
  const double I_denom = 1.0 / (h1234 * (h234 * h34));
  const double I_h234 = (h1234 * h34) * I_denom;
  const double I_h1234 = (h234 * h34) * I_denom;
  const double W11 = -h1 * (I_h1234 + I_h123 + I_h12);
  const double W21 = h1 * h12 * (I_h234 * I_h1234 + I_h23 * (I_h234 + I_h123));
  const double W31 = -h1 * h12 * h123 * I_denom;
  const double W12 = 2.0 * (I_h12 * (1.0 + (h1 + h12) * (I_h1234 + I_h123)) + h1 * I_h1234 * I_h123);
  const double W22 = -2.0 * ((h1 * h12 * I_h1234) * (I_h23 * (I_h234 + I_h123)) + (h1 + h12) * (I_h1234 * I_h234 + I_h23 * (I_h234 + I_h123)));
  const double W32 = 2.0 * ((h1 + h12) * h123 + h1 * h12) * I_denom;
  const double W13 = -3.0 * I_h12 * I_h123 * (1.0 + I_h1234 * ((h1 + h12) + h123));
  const double W23 = 3.0 * I_h23 * (I_h123 + I_h1234 * ((h1 + h12) + h123) * (I_h123 + I_h234));
  const double W33 = -3.0 * ((h1 + h12) + h123) * I_denom;
  const double W14 = 4.0 * I_h1234 * I_h123 * I_h12;
  const double W24 = -4.0 * I_h1234 * (I_h23 * (I_h123 + I_h234));
  const double W34 = 4.0 * I_denom;
  Csys[0] = ((u[0] + (W11 * (u[1] - u[0]))) + (W21 * (u[2] - u[1]))) + (W31 * (u[3] - u[2]));
  Csys[1] = ((W12 * (u[1] - u[0])) + (W22 * (u[2] - u[1]))) + (W32 * (u[3] - u[2]));
  Csys[2] = ((W13 * (u[1] - u[0])) + (W23 * (u[2] - u[1]))) + (W33 * (u[3] - u[2]));
  Csys[3] = ((W14 * (u[1] - u[0])) + (W24 * (u[2] - u[1]))) + (W34 * (u[3] - u[2]));
}

struct ReconArgs {
  int scheme;                 // the scheme after build_reconstructions_1d's small-n0 demotion
  int boundary_extrapolation;
  double h_neglect, h_neglect_edge;
  int n0;
};

// interior edge value of edge_values_explicit_h4 :246-279 between cells (h1,u1) and (h2,u2), with their outer neighbours
__device__ __forceinline__ double edge_h4(double h0, double h1, double h2, double h3, double u0, double u1, double u2, double u3, double hne) {
  const double hMinFrac = 1.e-5;
  if (h0 + h1 == 0.0 || h1 + h2 == 0.0 || h2 + h3 == 0.0) {
    const double h_min = hMinFrac * dmax(hne, (h0 + h1) + (h2 + h3));
    h0 = dmax(h_min, h0); h1 = dmax(h_min, h1); h2 = dmax(h_min, h2); h3 = dmax(h_min, h3);
  }
  const double I_h12 = 1.0 / (h1 + h2);
  const double I_den_et2 = 1.0 / (((h0 + h1) + h2) * (h0 + h1)), I_h012 = (h0 + h1) * I_den_et2;
  const double I_den_et3 = 1.0 / ((h1 + (h2 + h3)) * (h2 + h3)), I_h123 = (h2 + h3) * I_den_et3;
  const double et1 = (1.0 + (h1 * I_h012 + (h0 + h1) * I_h123)) * I_h12 * (h2 * (h2 + h3)) * u1 +
                     (1.0 + (h2 * I_h123 + (h2 + h3) * I_h012)) * I_h12 * (h1 * (h0 + h1)) * u2;
  const double et2 = (h1 * (h2 * (h2 + h3)) * I_den_et2) * (u1 - u0);
  const double et3 = (h2 * (h1 * (h0 + h1)) * I_den_et3) * (u2 - u3);
  return (et1 + (et2 + et3)) / ((h0 + h1) + (h2 + h3));
}
// bound_edge_values :39-101 for one cell (a, b: its edge values; m / p: the neighbour, or the cell itself at the ends)
__device__ __forceinline__ void bound_edges(double um, double uk, double up, double hm, double hk, double hp, double &a, double &b) {
  double slope_x_h = 0.0;
  if (((hm + hp) + 2.0 * hk) > 0.0) {
    const double sigma_l = (uk - um);
    const double sigma_c = (up - um) * (hk / ((hm + hp) + 2.0 * hk));
    const double sigma_r = (up - uk);
    if ((sigma_l * sigma_r) > 0.0) slope_x_h = fsign(dmin3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
  }
  if ((um - a) * (a - uk) < 0.0) a = uk - fsign(dmin(fabs(slope_x_h), fabs(a - uk)), slope_x_h);
  if ((up - b) * (b - uk) < 0.0) b = uk + fsign(dmin(fabs(slope_x_h), fabs(b - uk)), slope_x_h);
  a = dmax(dmin(a, dmax(um, uk)), dmin(um, uk));
  b = dmax(dmin(b, dmax(up, uk)), dmin(up, uk));
}
// the interior-cell part of PPM_limiter_standard :80-116
__device__ __forceinline__ void ppm_limit(double u_l, double u_c, double u_r, double &edge_l, double &edge_r) {
  if ((u_r - u_c) * (u_c - u_l) <= 0.0) {
    edge_l = u_c; edge_r = u_c;
  } else {
    const double expr1 = 3.0 * (edge_r - edge_l) * ((u_c - edge_l) + (u_c - edge_r));
    const double expr2 = (edge_r - edge_l) * (edge_r - edge_l);
    if (expr1 > expr2) {
      edge_l = u_c + 2.0 * (u_c - edge_r);
      edge_l = dmax(dmin(edge_l, dmax(u_l, u_c)), dmin(u_l, u_c));
    } else if (expr1 < -expr2) {
      edge_r = u_c + 2.0 * (u_c - edge_l);
      edge_r = dmax(dmin(edge_r, dmax(u_r, u_c)), dmin(u_r, u_c));
    }
  }
  if (fabs(edge_r - edge_l) < dmax(1.e-60, DBL_EPSILON * fabs(u_c))) { edge_l = u_c; edge_r = u_c; }
}

#ifndef RECON_RING
#define RECON_RING 4
#endif
#ifndef RECON_WAVES
#define RECON_WAVES 4
#endif
// build_reconstructions_1d :410-550 for one column.  h, u: the source column; E1, E2, C2: outputs (C2 only for PLM);
// Ucopy: a copy of u (the field itself may be overwritten by the remapped values).  All 1-based.
// The reference makes one pass over the column per stage (edge values, bounding, discontinuity check, limiter; first-guess
// slopes, monotonized slopes, coefficients).  Every stage only looks one or two cells away, so here ONE sweep carries a
// register window of the cells and of the intermediate values and finishes cell k-1 when cell k has been bounded: the
// column arrays are read once and the results written once, all with the same k in every lane (coalesced).
// IL: the two edge values of a cell next to each other in ONE array (E1[2 n], E1[2 n + 1]; E2 unused) -- what k_remap_merge reads:
// one 16-byte store and load per lane instead of two 8-byte ones.
// hst != 0: the column is a VELOCITY column and h holds the thicknesses of the CELLS: the thickness at the velocity point is formed
// where it is read, 0.5 * (h(cell) + h(cell + hst)) -- ALE_remap_set_h_vel's expression (MOM_ALE.F90:882; k_set_h_vel), so the bits
// are those of the h_u / h_v arrays it would have written and the remapping would have read back.
template <bool IL = false, bool PAIR = false>
__device__ void reconstruct_column(const ReconArgs &A, const double *__restrict__ h, const double *__restrict__ u, View vs,
                                   double *E1, double *E2, double *C2, double *Ucopy, View vw, int hst = 0) {
  const int N = A.n0;
#define H(k) (PAIR ? 0.5 * (AT(h, vs, k) + AT(h + hst, vs, k)) : AT(h, vs, k))
#define U(k) AT(u, vs, k)
#define e1(k) E1[(IL ? 2 : 1) * (vw.base + (size_t)((k) - 1) * vw.lev)]
#define e2(k) (IL ? E1 + 1 : E2)[(IL ? 2 : 1) * (vw.base + (size_t)((k) - 1) * vw.lev)]
#define c2(k) AT(C2, vw, k)
  if (A.scheme == MOM6X_REMAP_PCM) {                                          // PCM_functions.F90:16-35
    for (int k = 1; k <= N; k++) { const double v = U(k); e1(k) = v; e2(k) = v; if (Ucopy) AT(Ucopy, vw, k) = v; }
    return;
  }
  if (A.scheme == MOM6X_REMAP_PLM) {                                          // PLM_reconstruction :197-262 (N >= 2)
    const double almost_one = 1. - DBL_EPSILON, hn = A.h_neglect;
    // window at step k: cells k-1 (m), k (c), k+1 (p), k+2 (q); slopes slp(k-1..k+1), mslp(k-1..k)
    double um = U(1), uc = U(1), up = (N >= 2) ? U(2) : 0., hc = H(1), hp = (N >= 2) ? H(2) : 0.;
    double s_m = 0., s_c = 0., s_p = 0., ms_m = 0., ms_c = 0.;   // slp(k-1), slp(k), slp(k+1), mslp(k-1), mslp(k)
    // step k finishes cell k-1; cell 1 and N are written explicitly
    // prime: k = 1: slp(1) = 0, slp(2)
    if (N >= 3) s_p = PLM_slope_wa(H(1), H(2), H(3), hn, U(1), U(2), U(3));
    e1(1) = uc; e2(1) = uc; c2(1) = 0.;
    if (Ucopy) AT(Ucopy, vw, 1) = uc;
    for (int k = 2; k <= N - 1; k++) {
      // shift the window to cell k
      um = uc; uc = up; hc = hp; up = U(k + 1); hp = H(k + 1);
      s_m = s_c; s_c = s_p; ms_m = ms_c;
      s_p = (k + 1 <= N - 1) ? PLM_slope_wa(hc, hp, H(k + 2), hn, uc, up, U(k + 2)) : 0.;     // slp(k+1)
      ms_c = PLM_monotonized_slope(um, uc, up, s_m, s_c, s_p);                                   // mslp(k)
      if (Ucopy) AT(Ucopy, vw, k) = uc;
      if (k >= 3) {   // finish cell k-1 (an interior cell): needs mslp(k-1), mslp(k), slp(k)
        const double slope = ms_m, ukm = um;
        const double u_l = ukm - 0.5 * slope, u_r = ukm + 0.5 * slope;
        e1(k - 1) = u_l; e2(k - 1) = u_r;
        double p2 = (u_r - u_l);
        const double edge = p2 + u_l;
        const double e_r = uc - 0.5 * fsign(ms_c, s_c);
        if ((edge - ukm) * (e_r - edge) < 0.) p2 = p2 * almost_one;
        c2(k - 1) = p2;
      }
    }
    if (N >= 3) {   // finish cell N-1: mslp(N) = slp(N) = 0
      const double slope = ms_c, ukm = uc;
      const double u_l = ukm - 0.5 * slope, u_r = ukm + 0.5 * slope;
      e1(N - 1) = u_l; e2(N - 1) = u_r;
      double p2 = (u_r - u_l);
      const double edge = p2 + u_l;
      const double e_r = up - 0.5 * fsign(0., 0.);
      if ((edge - ukm) * (e_r - edge) < 0.) p2 = p2 * almost_one;
      c2(N - 1) = p2;
    }
    const double uN = U(N);
    e1(N) = uN; e2(N) = uN; c2(N) = 0.;
    if (Ucopy) AT(Ucopy, vw, N) = uN;
    if (A.boundary_extrapolation) {                                           // PLM_boundary_extrapolation :274-308
      double slope = -PLM_extrapolate_slope(H(2), H(1), hn, U(2), U(1));
      e1(1) = U(1) - 0.5 * slope; e2(1) = U(1) + 0.5 * slope;
      c2(1) = e2(1) - e1(1);
      slope = PLM_extrapolate_slope(H(N - 1), H(N), hn, U(N - 1), U(N));
      e1(N) = U(N) - 0.5 * slope; e2(N) = U(N) + 0.5 * slope;
      c2(N) = e2(N) - e1(N);
    }
    return;
  }
  // ---- PPM_H4 / PPM_IH4 (N >= 4): edge_values_explicit_h4 :213-348 or edge_values_implicit_h4 :473-630, then
  // PPM_limiter_standard :62-121
  const double hne = A.h_neglect_edge;
  const bool implicit = (A.scheme == MOM6X_REMAP_PPM_IH4);
  double edge_1, edge_2, edge_N, edge_N1;        // the two edge values at either end come from end_value_h4 :314-346
  {
    double dz[4], ut[4], C[4];
    for (int i = 1; i <= 4; i++) { dz[i - 1] = dmax(hne, H(i)); ut[i - 1] = U(i); }
    end_value_h4(dz, ut, C);
    edge_1 = C[0];
    edge_2 = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
    for (int i = 1; i <= 4; i++) { dz[i - 1] = dmax(hne, H(N + 1 - i)); ut[i - 1] = U(N + 1 - i); }
    end_value_h4(dz, ut, C);
    edge_N1 = C[0];
    edge_N = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
  }
  if (implicit) {
    // The N+1 interface values solve a diagonally dominant tridiagonal system (solve_diag_dominant_tridiag,
    // regrid_solvers.F90:246-280): the forward sweep leaves X(1..N) in the E1 column and c1(1..N) in the C2 column (free
    // for a PPM scheme), X(N+1) stays in a register; the backward sweep turns E1 into the final interface values, which
    // the limiter sweep below then reads one interface ahead of the cell it overwrites.
    double I_pivot = 1.0 / (1.0 + 0.0);                                       // tri_c(1) = 1, tri_u(1) = 0 :573-575
    double d1 = 1.0 * I_pivot, x_prev = edge_1 * I_pivot;
    c2(1) = 0.0 * I_pivot; e1(1) = x_prev;
    double h_a = H(1), u_a = U(1);
    for (int k = 2; k <= N; k++) {                                            // row k couples cells k-1 and k :528-552
      const double h_b = H(k), u_b = U(k);
      double h0 = dmax(h_a, hne), h1 = dmax(h_b, hne);
      if (fabs(h0) < 1.0e-12 * fabs(h1)) h0 = 1.0e-12 * h1;
      if (fabs(h1) < 1.0e-12 * fabs(h0)) h1 = 1.0e-12 * h0;
      const double I_h2 = 1.0 / ((h0 + h1) * (h0 + h1));
      const double alpha = (h1 * h1) * I_h2, beta = (h0 * h0) * I_h2, abmix = (h0 * h1) * I_h2;
      const double a = 2.0 * alpha * (alpha + 2.0 * beta + 3.0 * abmix);
      const double b = 2.0 * beta * (beta + 2.0 * alpha + 3.0 * abmix);
      const double Ac = 2.0 * abmix, R = a * u_a + b * u_b;
      const double denom_t1 = Ac + d1 * alpha;
      I_pivot = 1.0 / (denom_t1 + beta);
      d1 = denom_t1 * I_pivot;
      c2(k) = beta * I_pivot;
      x_prev = (R - alpha * x_prev) * I_pivot;
      e1(k) = x_prev;
      h_a = h_b; u_a = u_b;
    }
    I_pivot = 1.0 / (1.0 + d1 * 0.0);                                         // tri_c(N+1) = 1, tri_l(N+1) = 0 :608-613
    edge_N1 = (edge_N1 - 0.0 * x_prev) * I_pivot;
    double x_next = edge_N1;
    for (int k = N; k >= 1; k--) { x_next = e1(k) - c2(k) * x_next; e1(k) = x_next; }
    edge_1 = x_next;
  }
  // window at step k (bounding cell k, finishing cell k-1): cells k-2 (mm), k-1 (m), k (c), k+1 (p), k+2 (q)
  double umm = 0., um = 0., uc = 0., up = U(1), uq = U(2), hm = 0., hc = 0., hp = H(1), hq = H(2);
  double ed_c = 0., ed_p = edge_1;               // edge(k), edge(k+1)
  double a_m = 0., b_m = 0.;                     // bounded edge values of cell k-1 (a_m already through its pair check with k-2)
  // The column is read four cells ahead of its use (a ring of registers indexed by the level mod 4, the loop unrolled by
  // four so that the index is static and nothing is copied): a lane that asked for cell k+2 and used it in the same step
  // kept two loads in flight, and the sweep ran at the latency of a load per step.
  double ru[RECON_RING], rh[RECON_RING];
#pragma unroll
  for (int L = 3; L <= 2 + RECON_RING; L++) { ru[L % RECON_RING] = (L <= N) ? U(L) : 0.; rh[L % RECON_RING] = (L <= N) ? H(L) : 0.; }
  for (int k0 = 1; k0 <= N; k0 += RECON_RING)
#pragma unroll
  for (int kq = 0; kq < RECON_RING; kq++) {
    const int k = k0 + kq;
    if (k > N) break;
    umm = um; um = uc; uc = up; up = uq; hm = hc; hc = hp; hp = hq;
    if (k + 2 <= N) {
      uq = ru[(kq + 3) % RECON_RING]; hq = rh[(kq + 3) % RECON_RING];                      // level k + 2 (k0 = 1 mod RECON_RING)
      if (k + 2 + RECON_RING <= N) { ru[(kq + 3) % RECON_RING] = U(k + 2 + RECON_RING); rh[(kq + 3) % RECON_RING] = H(k + 2 + RECON_RING); }
    }
    ed_c = ed_p;
    const int e = k + 1;                         // the edge below cell k
    if (implicit) {
      ed_p = edge_N1;
      if (e <= N) { ed_p = e1(e); asm volatile("" : "+v"(ed_p)); }   // (its wait stays on this path: the explicit scheme's loads run ahead)
    }
    else if (e <= 2) ed_p = edge_2;
    else if (e >= N) ed_p = (e == N) ? edge_N : edge_N1;
    else ed_p = edge_h4(hm, hc, hp, hq, um, uc, up, uq, hne);      // i = e: cells e-2 .. e+1 = k-1 .. k+2
    if (Ucopy) AT(Ucopy, vw, k) = uc;
    double a = ed_c, b = ed_p;
    bound_edges((k > 1) ? um : uc, uc, (k < N) ? up : uc, (k > 1) ? hm : hc, hc, (k < N) ? hp : hc, a, b);
    if (k > 1) {
      // check_discontinuous_edge_values :132-150 for the pair (k-1, k)
      if ((a - b_m) * (uc - um) < 0.0) {
        double u0_avg = 0.5 * (b_m + a);
        u0_avg = dmax(dmin(u0_avg, dmax(um, uc)), dmin(um, uc));
        b_m = u0_avg; a = u0_avg;
      }
      // cell k-1 is complete
      if (k - 1 == 1) { e1(1) = um; e2(1) = um; }
      else {
        ppm_limit(umm, um, uc, a_m, b_m);
        e1(k - 1) = a_m; e2(k - 1) = b_m;
      }
    }
    a_m = a; b_m = b;
  }
  e1(N) = uc; e2(N) = uc;
  if (A.boundary_extrapolation) {                                             // PPM_boundary_extrapolation :166-296
    const double hn = A.h_neglect;
    {
      const double h0 = H(1), h1 = H(2), u0 = U(1), u1 = U(2);
      const double b = 4.0 * (u1 - e1(2)) + 2.0 * (u1 - e2(2));               // ppoly_coef(2,2) of PPM_reconstruction :49
      double u1_r = b * ((h0 + hn) / (h1 + hn));
      const double slope = 2.0 * (u1 - u0);
      if (fabs(u1_r) > fabs(slope)) u1_r = slope;
      double u0_r = e1(2);
      double u0_l = 3.0 * u0 + 0.5 * u1_r - 2.0 * u0_r;
      const double exp1 = (u0_r - u0_l) * (u0 - 0.5 * (u0_l + u0_r));
      const double exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
      if (exp1 > exp2) u0_l = 3.0 * u0 - 2.0 * u0_r;
      if (exp1 < -exp2) u0_r = 3.0 * u0 - 2.0 * u0_l;
      e1(1) = u0_l; e2(1) = u0_r;
    }
    {
      const double h0 = H(N - 1), h1 = H(N), u0 = U(N - 1), u1 = U(N);
      const double b = 4.0 * (u0 - e1(N - 1)) + 2.0 * (u0 - e2(N - 1));       // ppoly_coef(N-1,2)
      const double c = 3.0 * ((e2(N - 1) - u0) + (e1(N - 1) - u0));           // ppoly_coef(N-1,3)
      double u1_l = (b + 2 * c);
      u1_l = u1_l * ((h1 + hn) / (h0 + hn));
      const double slope = 2.0 * (u1 - u0);
      if (fabs(u1_l) > fabs(slope)) u1_l = slope;
      double u0_l = e2(N - 1);
      double u0_r = 3.0 * u1 - 0.5 * u1_l - 2.0 * u0_l;
      const double exp1 = (u0_r - u0_l) * (u1 - 0.5 * (u0_l + u0_r));
      const double exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
      if (exp1 > exp2) u0_l = 3.0 * u1 - 2.0 * u0_r;
      if (exp1 < -exp2) u0_r = 3.0 * u1 - 2.0 * u0_l;
      e1(N) = u0_l; e2(N) = u0_r;
    }
  }
#undef H
#undef U
#undef e1
#undef e2
#undef c2
}

// average_value_ppoly :1391-1494 for PCM / PLM / PPM; a_L, a_R, u_c, p2 = E(i0,1), E(i0,2), u0(i0), coefs(i0,2) [PLM]
__device__ __forceinline__ double average_value(int method, double a_L, double a_R, double u_c, double p2, double xa, double xb) {
  if (xb > xa) {
    if (method == INTEGRATION_PCM) return u_c;
    if (method == INTEGRATION_PLM) return (a_L + p2 * 0.5 * (xb + xa));
    const double mx = 0.5 * (xa + xb);
    const double a_c = 0.5 * ((u_c - a_L) + (u_c - a_R));
    if (mx < 0.5) {
      const double xa2b2ab = (xa * xa + xb * xb) + xa * xb;
      return a_L + ((a_R - a_L) * mx + a_c * (3. * (xb + xa) - 2. * xa2b2ab));
    }
    const double Ya = 1. - xa, Yb = 1. - xb, my = 0.5 * (Ya + Yb);
    const double Ya2b2ab = (Ya * Ya + Yb * Yb) + Ya * Yb;
    return a_R + ((a_L - a_R) * my + a_c * (3. * (Yb + Ya) - 2. * Ya2b2ab));
  }
  if (method == INTEGRATION_PCM) return u_c;       // ppoly0_coefs(i0,1) of PCM = u0(i0)
  const double Ya = 1. - xa;
  if (method == INTEGRATION_PLM) {
    if (xa < 0.5) return a_L + xa * (a_R - a_L);
    return a_R + Ya * (a_L - a_R);
  }
  const double a_c = 3. * ((u_c - a_L) + (u_c - a_R));
  if (xa < 0.5) return a_L + xa * ((a_R - a_L) + a_c * Ya);
  return a_R + Ya * ((a_L - a_R) + a_c * xa);
}

struct Merge {            // the running state of intersect_src_tgt_grids' loop :688-795
  double h0s, h1s;        // h0_supply, h1_supply
  double h1full;          // h1(i1), the whole width of the current target cell
  int i0, i1;
  bool src, tgt;          // src_has_volume, tgt_has_volume
};
enum { EV_SRC = 1, EV_TGT = 2 };

struct ApplyArgs {
  int n0, n1, method, om4, fb_sub, fb_tgt;
};

// one iteration of the loop: the new sub-cell's raw width dh (for h0_eff and the thickest-sub-cell test), its stored
// width h_sub, and which cell it closes
__device__ __forceinline__ int merge_step(Merge &m, const double *__restrict__ h0, View v0, const double *__restrict__ h1, View v1,
                                          int n0, int n1, double &dh, double &h_sub, double &eff) {
  dh = dmin(m.h0s, m.h1s);
  eff = dmin(dh, m.h0s);
  h_sub = dh;
  int ev;
  if (m.h0s <= m.h1s && m.src) { m.h1s = m.h1s - dh; ev = EV_SRC; }
  else if (m.h0s >= m.h1s && m.tgt) { m.h0s = m.h0s - dh; ev = EV_TGT; }
  else if (m.src) { h_sub = m.h0s; ev = EV_SRC; }
  else { h_sub = m.h1s; ev = EV_TGT; }
  if (ev == EV_SRC) {
    if (m.i0 < n0) { m.i0 = m.i0 + 1; m.h0s = AT(h0, v0, m.i0); }
    else { m.h0s = 0.; m.src = false; }
  } else {
    if (m.i1 < n1) { m.i1 = m.i1 + 1; m.h1s = AT(h1, v1, m.i1); m.h1full = m.h1s; }
    else { m.h1s = 0.; m.tgt = false; }
  }
  return ev;
}

struct Target {           // remap_sub_to_tgt_grid_om4 :1125-1160, one target cell at a time
  double dh, duh, umin, umax, ufirst;
  bool started;
};
__device__ __forceinline__ void tgt_reset(Target &t) { t.dh = 0.; t.duh = 0.; t.umin = 0.; t.umax = 0.; t.ufirst = 0.; t.started = false; }
__device__ __forceinline__ void tgt_feed(Target &t, double hs, double u, double uh, int fb) {
  if (!t.started) { t.ufirst = u; t.umin = u; t.umax = u; t.started = true; }
  if (fb) { t.umin = dmin(t.umin, u); t.umax = dmax(t.umax, u); }
  t.dh = t.dh + hs;
  t.duh = t.duh + uh;
}
__device__ __forceinline__ double tgt_close(Target &t, double h1, int fb) {
  double r;
  if (h1 > 0.) {
    r = t.duh / t.dh;
    if (fb) r = dmax(t.umin, dmin(t.umax, r));
  } else r = t.ufirst;
  tgt_reset(t);
  return r;
}

// remapping_core_h :234 for one column (after reconstruct_column).  u1 may alias the array the source values came from
// as long as `u0` points to a copy.
// CFG = 1: the OM4 switch set (PPM reconstruction, remap_src_to_sub_grid_om4, target values bounded, sub-cell values not) known at
// compile time -- the merge is issue-bound, and every sub-cell otherwise tests the method and the integrator again.
template <int CFG>
__device__ void apply_column(const ApplyArgs &A0, const double *__restrict__ h0, const double *__restrict__ u0, View v0,
                             const double *__restrict__ E1, const double *__restrict__ E2, const double *__restrict__ C2, View vw,
                             const double *__restrict__ h1, double *u1, View v1) {
  ApplyArgs A = A0;
  if (CFG == 1) { A.method = INTEGRATION_PPM; A.om4 = 1; A.fb_sub = 0; A.fb_tgt = 1; }   // (constants from here on)
  const int n0 = A.n0, n1 = A.n1, ns = n0 + n1 + 1, method = A.method;
  int last_thick = 0;
  for (int k = 1; k <= n0; k++) if (AT(h0, v0, k) > 0.) last_thick = k;          // i0_last_thick_cell :884-889
  Merge m; m.h0s = AT(h0, v0, 1); m.h1s = AT(h1, v1, 1); m.i0 = 1; m.i1 = 1; m.src = true; m.tgt = true;
  m.h1full = m.h1s;
  Target T; tgt_reset(T);
  int i_sub = 1;            // the index of the last sub-cell made
  // the source cell being integrated
  double a_L = AT(E1, vw, 1), a_R = AT(E2, vw, 1), u_c = AT(u0, v0, 1), p2 = (method == INTEGRATION_PLM) ? AT(C2, vw, 1) : 0.;
  double hsrc = AT(h0, v0, 1);
  // sub-cell 1 (zero width, top edge): :716-720, :893-894 / :1012-1030
  double u_s1 = A.om4 ? a_L : ((hsrc > 0.) ? average_value(method, a_L, a_R, u_c, p2, 0., 0.) : u_c);
  if (A.fb_sub && !A.om4) { u_s1 = dmax(u_s1, dmin(a_L, a_R)); u_s1 = dmin(u_s1, dmax(a_L, a_R)); }
  const double uh_s1 = A.om4 ? 0. : 0. * u_s1;
  tgt_feed(T, 0., u_s1, uh_s1, A.fb_tgt);
  double xa = 0., cum = 0., h0_eff_last = 0.;
  // xa, cum after the zero-width sub-cell: non-OM4 runs it through the loop (dh0_eff += 0, xb = 0 or, for a vanished
  // first cell, 1); OM4 starts the loop at sub-cell 2 with xa = 0
  if (!A.om4) { xa = (hsrc > 0.) ? dmin(1., 0. / hsrc) : 1.; }
  while (m.src) {
    const Merge s = m;
    const int i0 = m.i0;
    const double umin0 = dmin(a_L, a_R), umax0 = dmax(a_L, a_R);
    // ---- pass 1: the cell's sub-cells, its effective width h0_eff :722-747 and the thickest sub-cell :726-729.
    // The first NB sub-cells are also kept in registers: a source cell rarely has more, and then passes 2 and 3 work
    // from that buffer instead of replaying the merge.
    constexpr int NB = 4;
    double b_hs[NB], b_h1[NB]; int b_i1[NB]; bool b_has[NB], b_tev[NB];
    Merge t = s;
    int cnt = 0, imax = -1;
    double dh_max = 0., h0_eff = 0., hsub_imax = 0.;
    {
      int ev;
      do {
        const bool has = t.tgt;
        const int i1 = t.i1;
        const double h1f = t.h1full;
        double dh, hs, eff;
        ev = merge_step(t, h0, v0, h1, v1, n0, n1, dh, hs, eff);
        h0_eff = h0_eff + eff;
        if (dh >= dh_max) { imax = cnt; dh_max = dh; hsub_imax = hs; }
#pragma unroll
        for (int q = 0; q < NB; q++)
          if (q == cnt) { b_hs[q] = hs; b_h1[q] = h1f; b_i1[q] = i1; b_has[q] = has; b_tev[q] = (ev == EV_TGT); }
        cnt++;
      } while (ev != EV_SRC);
    }
    const double den = A.om4 ? h0_eff : hsrc;
    h0_eff_last = h0_eff;
    const bool adjust = (i0 <= last_thick) && (hsub_imax > 0.);
    double duh = 0.;
    if (i0 == 1) duh = duh + uh_s1;
    if (cnt <= NB) {
      // ---- passes 2 and 3 from the register buffer
      double b_u[NB], b_uh[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < cnt) {
          const double hs = b_hs[c];
          double u, uh;
          if (A.om4 && (i_sub + 1 + c) == ns) { u = AT(E2, vw, n0); uh = u * hs; }
          else {
            cum = cum + hs;
            double xb;
            if (den > 0.) { xb = dmin(1., cum / den); u = average_value(method, a_L, a_R, u_c, p2, xa, xb); }
            else { xb = 1.; u = u_c; }
            if (A.fb_sub) { u = dmax(u, umin0); u = dmin(u, umax0); }
            uh = hs * u;
            xa = xb;
          }
          b_u[c] = u; b_uh[c] = uh;
          if (c != imax) duh = duh + uh;
        }
      }
      const double uh_adj = u_c * hsrc - duh;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        if (c < cnt) {
          double uh = b_uh[c];
          if (adjust && c == imax) uh = uh_adj;
          if (b_has[c]) {
            tgt_feed(T, b_hs[c], b_u[c], uh, A.fb_tgt);
            if (b_tev[c]) AT(u1, v1, b_i1[c]) = tgt_close(T, b_h1[c], A.fb_tgt);
          }
        }
      }
      i_sub += cnt;
    } else {
      // ---- pass 2: sum of u*h over the sub-cells other than the thickest :939-957
      {
        t = s;
        double xa2 = xa, cum2 = cum;
        for (int c = 0; c < cnt; c++) {
          double dh, hs, eff;
          merge_step(t, h0, v0, h1, v1, n0, n1, dh, hs, eff);
          double u, uh;
          if (A.om4 && (i_sub + 1 + c) == ns) { u = AT(E2, vw, n0); uh = u * hs; }
          else {
            cum2 = cum2 + hs;
            double xb;
            if (den > 0.) { xb = dmin(1., cum2 / den); u = average_value(method, a_L, a_R, u_c, p2, xa2, xb); }
            else { xb = 1.; u = u_c; }
            if (A.fb_sub) { u = dmax(u, umin0); u = dmin(u, umax0); }
            uh = hs * u;
            xa2 = xb;
          }
          if (c != imax) duh = duh + uh;
        }
      }
      const double uh_adj = u_c * hsrc - duh;
      // ---- pass 3: emit the sub-cells into the target sums
      {
        t = s;
        for (int c = 0; c < cnt; c++) {
          const bool has_tgt = t.tgt;
          const int i1 = t.i1;
          const double h1f = t.h1full;
          double dh, hs, eff;
          const int ev = merge_step(t, h0, v0, h1, v1, n0, n1, dh, hs, eff);
          i_sub++;
          double u, uh;
          if (A.om4 && i_sub == ns) { u = AT(E2, vw, n0); uh = u * hs; }
          else {
            cum = cum + hs;
            double xb;
            if (den > 0.) { xb = dmin(1., cum / den); u = average_value(method, a_L, a_R, u_c, p2, xa, xb); }
            else { xb = 1.; u = u_c; }
            if (A.fb_sub) { u = dmax(u, umin0); u = dmin(u, umax0); }
            uh = hs * u;
            xa = xb;
          }
          if (adjust && c == imax) uh = uh_adj;
          if (has_tgt) {
            tgt_feed(T, hs, u, uh, A.fb_tgt);
            if (ev == EV_TGT) AT(u1, v1, i1) = tgt_close(T, h1f, A.fb_tgt);
          }
        }
      }
    }
    m = t;
    if (m.src) {          // the next source cell: "dh0_eff = 0 ; xa = 0" :932-934
      xa = 0.; cum = 0.;
      a_L = AT(E1, vw, m.i0); a_R = AT(E2, vw, m.i0); u_c = AT(u0, v0, m.i0); hsrc = m.h0s;
      if (method == INTEGRATION_PLM) p2 = AT(C2, vw, m.i0);
    }
  }
  // ---- the target column is deeper than the source column: the remaining sub-cells continue the last source cell
  {
    const double dd = A.om4 ? h0_eff_last : hsrc;   // h0_eff(n0) | h0(n0)
    const double umin0 = dmin(a_L, a_R), umax0 = dmax(a_L, a_R);
    while (m.tgt) {
      const int i1 = m.i1;
      const double h1f = m.h1full;
      double dh, hs, eff;
      merge_step(m, h0, v0, h1, v1, n0, n1, dh, hs, eff);
      i_sub++;
      double u, uh;
      if (A.om4 && i_sub == ns) { u = AT(E2, vw, n0); uh = u * hs; }
      else {
        cum = cum + hs;
        double xb;
        if (dd > 0.) { xb = dmin(1., cum / dd); u = average_value(method, a_L, a_R, u_c, p2, xa, xb); }
        else { xb = 1.; u = u_c; }
        if (A.fb_sub) { u = dmax(u, umin0); u = dmin(u, umax0); }
        uh = hs * u;
        xa = xb;
      }
      tgt_feed(T, hs, u, uh, A.fb_tgt);
      AT(u1, v1, i1) = tgt_close(T, h1f, A.fb_tgt);
    }
  }
}

struct Fields { double *p[8]; };

// the 3-D form: columns (i0..i1, j0..j1) with mask > 0; h_old / h_new / fields on the same staggering
template <bool IL, bool PAIR = false>
__global__ void __launch_bounds__(256, PAIR ? 2 : RECON_WAVES)
k_remap_recon(Dm d, const double *__restrict__ mask, ReconArgs A, const double *__restrict__ h_old, const double *__restrict__ f,
              double *E1, double *E2, double *C2, double *Ucopy, int i0, int i1, int j0, int j1, int hst) {
  const int i = I_BASE(i0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < i0 || i > i1 || j > j1) return;
  const size_t x = ix2(d, i, j);
  if (mask && !(mask[x] > 0.)) return;
  View v; v.base = x; v.lev = (size_t)d.slab;
  reconstruct_column<IL, PAIR>(A, h_old, f, v, E1, E2, C2, Ucopy, v, hst);
}
template <int CFG>
__global__ void __launch_bounds__(256)
k_remap_apply(Dm d, const double *__restrict__ mask, ApplyArgs A, const double *__restrict__ h_old, const double *__restrict__ h_new,
              const double *__restrict__ E1, const double *__restrict__ E2, const double *__restrict__ C2,
              const double *__restrict__ Ucopy, double *f, int i0, int i1, int j0, int j1) {
  const int i = I_BASE(i0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < i0 || i > i1 || j > j1) return;
  const size_t x = ix2(d, i, j);
  if (!(mask[x] > 0.)) return;
  View v; v.base = x; v.lev = (size_t)d.slab;
  apply_column<CFG>(A, h_old, Ucopy, v, E1, E2, C2, v, h_new, f, v);
}
// ---- the merge for the OM4 switch set, shared by the fields that live on one pair of grids --------------------------------
// REMAPPING_SCHEME = PPM_*, remap_src_to_sub_grid_om4 :845, target values bounded, sub-cell values not (what
// `k_remap_apply<1>` is compiled for), with the same results bit for bit.  k_remap_apply spends two thirds of a wavefront's
// life waiting for memory (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.64): every step of the merge needs h1 at an index that differs
// from lane to lane, and needs it at once.  Here
//  * the target column goes through a sliding window in LDS (MG_W levels per lane, the level MG_LEAD below the source cell
//    enters at the top of every iteration from a load issued one iteration earlier; a lane whose target index has left the
//    window reads global memory as before), and the source cell's values are loaded one iteration ahead, so a wavefront
//    waits at one place per source cell, for loads that are an iteration old;
//  * what the sub-cells of a source cell share between fields -- their widths, the positions xa, xb in the cell and the
//    polynomial weights that follow from them (two thirds of average_value_ppoly's arithmetic and its division) -- is formed
//    once and used by NF fields (T and S; tracers in pairs);
//  * a cell's sub-cells are kept as (width, two weights, form), from which a field's sub-cell mean costs four operations,
//    so the pass that sums u*h for the conservation fix and the pass that feeds the targets both evaluate it instead of
//    storing it; a cell with more than MG_NB sub-cells replays the merge from its saved start as k_remap_apply does.
// Three facts about the merge, for thicknesses >= 0, are used: (1) inside the source loop the sub-cells of a source cell are
// the closings of consecutive target cells followed by its own closing; a target cell closed by the 2nd, 3rd, ... of them
// lies inside the source cell and its whole width is the sub-cell's; (2) a sub-cell of positive width exists only in a
// source cell of positive thickness, which is then not below i0_last_thick_cell, so `adjust` is (hsub_imax > 0) alone;
// (3) the column's last sub-cell (u = E2(n0), :926-929) is the closing of source cell n0 when the targets are exhausted, or
// else the closing of target n1.
constexpr int MG_NT = 3, MG_W = 16, MG_LEAD = 8;
template <int NF> struct MergeFields { const double2 *E12[NF]; const double *Uc[NF]; double *out[NF]; };   // E12: (E1, E2) of a cell side by side
struct SubW { double p, q; int mode; };   // 0 / 1: xb > xa, left / right form; 2 / 3: a point, left / right; 4: u0(i0); 5: E2(n0)

// xa, xb of the next sub-cell (:900-912) and the weights of average_value_ppoly's PPM branches :1420-1437, :1470-1480
__device__ __forceinline__ SubW sub_weights(double &xa, double &cum, double hs, double den, bool last) {
  SubW w; w.p = 0.; w.q = 0.;
  if (last) { w.mode = 5; return w; }
  cum = cum + hs;
  if (den > 0.) {
    const double xb = dmin(1., cum / den);
    if (xb > xa) {
      const double mx = 0.5 * (xa + xb);
      if (mx < 0.5) { w.mode = 0; w.p = mx; w.q = 3. * (xb + xa) - 2. * ((xa * xa + xb * xb) + xa * xb); }
      else {
        const double Ya = 1. - xa, Yb = 1. - xb;
        w.mode = 1; w.p = 0.5 * (Ya + Yb); w.q = 3. * (Yb + Ya) - 2. * ((Ya * Ya + Yb * Yb) + Ya * Yb);
      }
    } else {
      const double Ya = 1. - xa;
      if (xa < 0.5) { w.mode = 2; w.p = xa; w.q = Ya; }
      else { w.mode = 3; w.p = Ya; w.q = xa; }
    }
    xa = xb;
  } else { w.mode = 4; xa = 1.; }
  return w;
}
// the same for the sub-cell that closes its source cell when it is not the column's last: cum >= den (the same sums, with
// hs >= dh in the last term), so xb = min(1, cum / den) is 1 without the division, which leaves the two right-hand forms
// with Yb = 0 (and x + 0 = x, x * 0 = 0 for the finite x >= 0 here)
__device__ __forceinline__ SubW sub_weights_closing(double &xa, double &cum, double hs, double den) {
  SubW w;
  cum = cum + hs;
  if (den > 0.) {
    const double Ya = 1. - xa;
    if (1. > xa) { w.mode = 1; w.p = 0.5 * Ya; w.q = 3. * Ya - 2. * (Ya * Ya); }
    else { w.mode = 3; w.p = Ya; w.q = xa; }
  } else { w.mode = 4; w.p = 0.; w.q = 0.; }
  xa = 1.;
  return w;
}
struct CellPoly { double aL, aR, uc, dLR, dRL, ac, ac3; };
__device__ __forceinline__ CellPoly cell_poly(double aL, double aR, double uc) {
  CellPoly c; c.aL = aL; c.aR = aR; c.uc = uc; c.dLR = aR - aL; c.dRL = aL - aR;
  const double s = (uc - aL) + (uc - aR);
  c.ac = 0.5 * s; c.ac3 = 3. * s;
  return c;
}
__device__ __forceinline__ double sub_mean(const SubW &w, const CellPoly &c) {
  const bool right = (w.mode & 1);
  const double B = right ? c.aR : c.aL, D = right ? c.dRL : c.dLR;
  if (w.mode < 2) return B + (D * w.p + c.ac * w.q);
  if (w.mode < 4) return B + w.p * (D + c.ac3 * w.q);
  return (w.mode == 4) ? c.uc : c.aR;
}
__device__ __forceinline__ double sub_mean_right(const SubW &w, const CellPoly &c) {   // modes 1, 3, 4, 5
  if (w.mode == 1) return c.aR + (c.dRL * w.p + c.ac * w.q);
  if (w.mode == 3) return c.aR + w.p * (c.dRL + c.ac3 * w.q);
  return (w.mode == 4) ? c.uc : c.aR;
}

template <int NF, bool PAIR = false>
__global__ void __launch_bounds__(256)
k_remap_merge(Dm d, const double *__restrict__ mask, const double *__restrict__ h0p, const double *__restrict__ h1p, MergeFields<NF> F,
              int i0, int i1, int j0, int j1, int hst) {
  __shared__ double win_all[4][MG_W][64];
  const int i = I_BASE(i0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < i0 || i > i1 || j > j1) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  if (!(mask[x] > 0.)) return;
  const int n = d.nk;
  double *win = &win_all[threadIdx.y][0][threadIdx.x];
#define WIN(L) win[((L) & (MG_W - 1)) * 64]
#define LEV(p, L) (p)[x + (size_t)((L) - 1) * slab]
  // (hst != 0: velocity columns on the cells' thicknesses, see reconstruct_column)
#define HLEV(p, L) (PAIR ? 0.5 * (LEV(p, L) + LEV((p) + hst, L)) : LEV(p, L))
  // the window holds the levels k + MG_LEAD - MG_W + 1 .. k + MG_LEAD of h1 during iteration k
  auto next_target = [&](int &it, double &h1s, double &h1full, bool &tgt, int k) {     // :765-771
    if (it < n) {
      it = it + 1;
      h1s = WIN(it);                   // (the slot exists whatever it holds)
      if ((unsigned)(it - (k + MG_LEAD - MG_W + 1)) >= (unsigned)MG_W) {
        h1s = HLEV(h1p, it);
        asm volatile("" : "+v"(h1s));  // (the wait for this load belongs here, not where the paths meet: there it would also
      }                                //  hold the lanes that read the window until this iteration's prefetches are back)
      h1full = h1s;
    } else { h1s = 0.; tgt = false; }
  };
  double h1s, h1_pf;
  {
    double t[MG_LEAD];
#pragma unroll
    for (int L = 1; L <= MG_LEAD; L++) t[L - 1] = (L <= n) ? HLEV(h1p, L) : 0.;
    h1_pf = (MG_LEAD + 1 <= n) ? HLEV(h1p, MG_LEAD + 1) : 0.;
#pragma unroll
    for (int L = 1; L <= MG_LEAD; L++) WIN(L) = t[L - 1];
    h1s = t[0];
  }
  double n_h0 = HLEV(h0p, 1), n_aL[NF], n_aR[NF], n_uc[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) { const double2 e = LEV(F.E12[f], 1); n_aL[f] = e.x; n_aR[f] = e.y; n_uc[f] = LEV(F.Uc[f], 1); }
  double h1full = h1s;
  int it = 1;
  bool tgt = true;
  // The running target cell (remap_sub_to_tgt_grid_om4 :1125-1160).  It always holds a sub-cell when a source cell begins:
  // the zero-width first one :893-894, later the closing sub-cell of the previous source cell.
  double Tdh = 0. + 0., Tduh[NF], Tmin[NF], Tmax[NF], Tfirst[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) { Tfirst[f] = n_aL[f]; Tmin[f] = n_aL[f]; Tmax[f] = n_aL[f]; Tduh[f] = 0. + 0.; }
  double xa = 0., cum = 0., h0_eff_last = 0.;
  CellPoly P[NF];
  // one more sub-cell of the running target / the first sub-cell of the next one / the target's value :1146-1158
  auto feed = [&](bool fresh, double hs, const double *u, const double *uh) {
    Tdh = (fresh ? 0. : Tdh) + hs;
#pragma unroll
    for (int f = 0; f < NF; f++) {
      if (fresh) { Tfirst[f] = u[f]; Tmin[f] = u[f]; Tmax[f] = u[f]; Tduh[f] = 0. + uh[f]; }
      else { Tmin[f] = dmin(Tmin[f], u[f]); Tmax[f] = dmax(Tmax[f], u[f]); Tduh[f] = Tduh[f] + uh[f]; }
    }
  };
  auto close = [&](int lev, double h1f) {
#pragma unroll
    for (int f = 0; f < NF; f++) {
      double r;
      if (h1f > 0.) { r = Tduh[f] / Tdh; r = dmax(Tmin[f], dmin(Tmax[f], r)); }
      else r = Tfirst[f];
      LEV(F.out[f], lev) = r;
    }
  };
  for (int k = 1; k <= n; k++) {
    // ---- what was asked for during the previous iteration arrives, the next requests leave
    const double hsrc = n_h0;
#pragma unroll
    for (int f = 0; f < NF; f++) P[f] = cell_poly(n_aL[f], n_aR[f], n_uc[f]);
    WIN(k + MG_LEAD) = h1_pf;
    if (k + 1 <= n) {
      n_h0 = HLEV(h0p, k + 1);
#pragma unroll
      for (int f = 0; f < NF; f++) { const double2 e = LEV(F.E12[f], k + 1); n_aL[f] = e.x; n_aR[f] = e.y; n_uc[f] = LEV(F.Uc[f], k + 1); }
    }
    if (k + MG_LEAD + 1 <= n) h1_pf = HLEV(h1p, k + MG_LEAD + 1);
    // ---- the cell's sub-cells (intersect_src_tgt_grids' loop :700-795): the target cells that end inside it, then its own
    // closing; their widths, the effective width h0_eff :722-747 and the thickest one :726-729
    double h0s = hsrc;
    const double s_h1s = h1s, s_h1full = h1full;
    const int s_it = it;
    double b_hs[MG_NT];
    int nt = 0, imax = -1;
    double dh_max = 0., h0_eff = 0., hsub_imax = 0.;
    bool more = tgt && !(h0s <= h1s);
#pragma unroll
    for (int c = 0; c < MG_NT; c++) {
      if (more) {
        const double dh = dmin(h0s, h1s);
        h0_eff = h0_eff + dh;
        if (dh >= dh_max) { imax = c; dh_max = dh; hsub_imax = dh; }
        b_hs[c] = dh; nt = c + 1;
        h0s = h0s - dh;
        next_target(it, h1s, h1full, tgt, k);
        more = tgt && !(h0s <= h1s);
      }
    }
    while (more) {           // more of them than the buffer holds
      const double dh = dmin(h0s, h1s);
      h0_eff = h0_eff + dh;
      if (dh >= dh_max) { imax = nt; dh_max = dh; hsub_imax = dh; }
      nt++;
      h0s = h0s - dh;
      next_target(it, h1s, h1full, tgt, k);
      more = tgt && !(h0s <= h1s);
    }
    double hs_l;
    {
      const double dh = dmin(h0s, h1s);
      hs_l = dh;
      if (h0s <= h1s) h1s = h1s - dh;
      else hs_l = h0s;
      h0_eff = h0_eff + dh;
      if (dh >= dh_max) { imax = nt; dh_max = dh; hsub_imax = hs_l; }
    }
    const bool has_last = tgt, col_last = (k == n) && !tgt, adjust = (hsub_imax > 0.);
    const double den = h0_eff;
    h0_eff_last = h0_eff;
    xa = 0.; cum = 0.;
    if (nt <= MG_NT) {
      SubW w[MG_NT], w_l;
#pragma unroll
      for (int c = 0; c < MG_NT; c++) if (c < nt) w[c] = sub_weights(xa, cum, b_hs[c], den, false);
      if (col_last) { w_l.mode = 5; w_l.p = 0.; w_l.q = 0.; }
      else w_l = sub_weights_closing(xa, cum, hs_l, den);
      double u[MG_NT][NF], u_l[NF], uh_adj[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) {
        double duh = 0. + 0.;
#pragma unroll
        for (int c = 0; c < MG_NT; c++)
          if (c < nt) { u[c][f] = sub_mean(w[c], P[f]); if (c != imax) duh = duh + b_hs[c] * u[c][f]; }
        u_l[f] = sub_mean_right(w_l, P[f]);
        if (nt != imax) duh = duh + hs_l * u_l[f];
        uh_adj[f] = P[f].uc * hsrc - duh;
      }
      double uh[NF];
#pragma unroll
      for (int c = 0; c < MG_NT; c++) {
        if (c < nt) {
#pragma unroll
          for (int f = 0; f < NF; f++) uh[f] = (adjust && c == imax) ? uh_adj[f] : b_hs[c] * u[c][f];
          feed(c > 0, b_hs[c], u[c], uh);
          close(s_it + c, (c == 0) ? s_h1full : b_hs[c]);
        }
      }
      if (has_last) {
#pragma unroll
        for (int f = 0; f < NF; f++) uh[f] = (adjust && nt == imax) ? uh_adj[f] : hs_l * u_l[f];
        feed(nt > 0, hs_l, u_l, uh);
      }
    } else {
      // the merge is replayed from the cell's start, once for the sum :939-957 ...
      double duh[NF], uh_adj[NF], u1[NF], uh1[NF];
#pragma unroll
      for (int f = 0; f < NF; f++) duh[f] = 0. + 0.;
      double r_h0s = hsrc, r_h1s = s_h1s, r_h1full = s_h1full, xa2 = 0., cum2 = 0.;
      int r_it = s_it;
      bool r_tgt = true;
      for (int c = 0; c <= nt; c++) {
        double hs = hs_l;
        SubW w;
        if (c < nt) {
          hs = dmin(r_h0s, r_h1s);
          r_h0s = r_h0s - hs;
          next_target(r_it, r_h1s, r_h1full, r_tgt, k);
          w = sub_weights(xa2, cum2, hs, den, false);
        } else if (col_last) { w.mode = 5; w.p = 0.; w.q = 0.; }
        else w = sub_weights_closing(xa2, cum2, hs, den);
#pragma unroll
        for (int f = 0; f < NF; f++) { const double t = hs * sub_mean(w, P[f]); if (c != imax) duh[f] = duh[f] + t; }
      }
#pragma unroll
      for (int f = 0; f < NF; f++) uh_adj[f] = P[f].uc * hsrc - duh[f];
      // ... and once to feed the targets
      r_h0s = hsrc; r_h1s = s_h1s; r_h1full = s_h1full; r_it = s_it; r_tgt = true;
      for (int c = 0; c <= nt; c++) {
        const int lev = r_it;
        const double h1f = r_h1full;
        double hs = hs_l;
        SubW w;
        if (c < nt) {
          hs = dmin(r_h0s, r_h1s);
          r_h0s = r_h0s - hs;
          next_target(r_it, r_h1s, r_h1full, r_tgt, k);
          w = sub_weights(xa, cum, hs, den, false);
        } else if (col_last) { w.mode = 5; w.p = 0.; w.q = 0.; }
        else w = sub_weights_closing(xa, cum, hs, den);
#pragma unroll
        for (int f = 0; f < NF; f++) { u1[f] = sub_mean(w, P[f]); uh1[f] = (adjust && c == imax) ? uh_adj[f] : hs * u1[f]; }
        if (c < nt || has_last) feed(c > 0, hs, u1, uh1);
        if (c < nt) close(lev, h1f);
      }
    }
  }
  // ---- the target column is deeper than the source column: the remaining sub-cells continue the last source cell
  bool fresh = false;
  while (tgt) {
    const int lev = it;
    const double h1f = h1full, hs = h1s;
    next_target(it, h1s, h1full, tgt, n);
    const SubW w = sub_weights(xa, cum, hs, h0_eff_last, !tgt);
    double u1[NF], uh1[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) { u1[f] = sub_mean(w, P[f]); uh1[f] = hs * u1[f]; }
    feed(fresh, hs, u1, uh1);
    close(lev, h1f);
    fresh = true;
  }
#undef WIN
#undef LEV
}

// the packed form of the unit tests: column c holds n0 | n1 values back to back; work arrays are [k][ncol]
__global__ void __launch_bounds__(64)
k_remap_packed(int ncol, ReconArgs R, ApplyArgs A, const double *__restrict__ h0, const double *__restrict__ u0,
               const double *__restrict__ h1, double *u1, double *E1, double *E2, double *C2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  View v0, v1, vw;
  v0.base = (size_t)c * A.n0; v0.lev = 1; v1.base = (size_t)c * A.n1; v1.lev = 1; vw.base = c; vw.lev = ncol;
  reconstruct_column(R, h0, u0, v0, E1, E2, C2, nullptr, vw);
  apply_column<0>(A, h0, u0, v0, E1, E2, C2, vw, h1, u1, v1);
}

__global__ void __launch_bounds__(256)
k_set_h_vel(Dm d, const double *__restrict__ G, const double *__restrict__ h_new, double *__restrict__ h_u, double *__restrict__ h_v) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni - 1 || j > d.nj - 1) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const bool do_u = (j >= 0) && (gm(G, d, MOM6X_G_mask2dCu)[x] > 0.), do_v = (i >= 0) && (gm(G, d, MOM6X_G_mask2dCv)[x] > 0.);
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double hc = h_new[c];
    if (do_u) h_u[c] = 0.5 * (hc + h_new[c + 1]);
    if (do_v) h_v[c] = 0.5 * (hc + h_new[c + d.pitch]);
  }
}

// bits of the context's device flag the regridding kernels raise (bit 1 is the NaN flag of continuity: left alone here)
enum { REGRID_FLAGS = 4 | 8 | 16 };

// ---- regridding: pieces shared by the coordinate generators.  A column lives in 3-D arrays ([k][slab], element k of
// array p at p[x + (k-1)*slab]): every loop has the same k in all lanes, so the accesses are coalesced.
#define LV(p, k) (p)[x + (size_t)((k) - 1) * slab]

// filtered_grid_motion MOM_regridding.F90:1105-1252 in two pieces, so that the column may live in global arrays or in
// registers: the constants of a column, and the displacement of one interface.
struct FilterConst { double sgn, zs, zd, wtd, Iwtd, dzwt, Idzwt, dInt_zs_zd, Aq; bool zero; };
// false: "z_old and z_new use different sign conventions" (a FATAL of the reference); F.zero: no motion at all (:1141-1143)
__device__ __forceinline__ bool filter_prepare(const mom6x_regrid_zstar_params &CS, double zOld1, double zOldB, double zNew1, double zNewB,
                                               FilterConst &F) {
  const double test = (zOldB - zOld1) * (zNewB - zNew1);
  F.zero = (test == 0.0);
  if (test < 0.0) return false;
  F.sgn = ((zOldB - zOld1) + (zNewB - zNew1) > 0.0) ? 1.0 : -1.0;
  F.zs = CS.depth_of_time_filter_shallow; F.zd = CS.depth_of_time_filter_deep;
  F.wtd = 1.0 - CS.old_grid_weight; F.Iwtd = 1.0 / F.wtd;
  F.dzwt = (F.zd - F.zs);
  F.Idzwt = 0.0; if (fabs(F.zd - F.zs) > 0.0) F.Idzwt = 1.0 / (F.zd - F.zs);
  F.dInt_zs_zd = 0.5 * (1.0 + F.Iwtd) * (F.zd - F.zs);
  F.Aq = 0.5 * (F.Iwtd - 1.0);
  return true;
}
__device__ __forceinline__ double filter_one(const FilterConst &F, double z_old_k, double z_new_k, double zOld1) {
  const double sgn = F.sgn, zs = F.zs, zd = F.zd, wtd = F.wtd, Iwtd = F.Iwtd, dzwt = F.dzwt, Idzwt = F.Idzwt, Aq = F.Aq;
  const double dz_tgt = sgn * (z_new_k - z_old_k);
  const double zr1 = sgn * (z_old_k - zOld1);
  double dz;
  if ((zr1 > zd) && (zr1 + wtd * dz_tgt > zd)) dz = sgn * wtd * dz_tgt;
  else if ((zr1 < zs) && (zr1 + dz_tgt < zs)) dz = sgn * dz_tgt;
  else {
    double Int_zd, Int_zs;
    if (zr1 >= zd) { Int_zd = Iwtd * (zd - zr1); Int_zs = Int_zd - F.dInt_zs_zd; }
    else if (zr1 <= zs) { Int_zs = (zs - zr1); Int_zd = F.dInt_zs_zd + (zs - zr1); }
    else {
      Int_zd = (zd - zr1) * (Iwtd * (0.5 * (zd + zr1) - zs) + 0.5 * (zd - zr1)) * Idzwt;
      Int_zs = (zs - zr1) * (0.5 * Iwtd * ((zr1 - zs)) + (zd - 0.5 * (zr1 + zs))) * Idzwt;
    }
    if (dz_tgt >= Int_zd) dz = sgn * ((zd - zr1) + wtd * (dz_tgt - Int_zd));
    else if (dz_tgt <= Int_zs) dz = sgn * ((zs - zr1) + (dz_tgt - Int_zs));
    else {
      double dz0, z0, F0;
      if (zr1 <= zs) { dz0 = zs - zr1; z0 = zs; F0 = dz_tgt - Int_zs; }
      else if (zr1 >= zd) { dz0 = zd - zr1; z0 = zd; F0 = dz_tgt - Int_zd; }
      else { dz0 = 0.0; z0 = zr1; F0 = dz_tgt; }
      const double Bq = (dzwt + 2.0 * Aq * (z0 - zs));
      dz = sgn * (dz0 + 2.0 * F0 * dzwt / (Bq + sqrt(Bq * Bq + 4.0 * Aq * F0 * dzwt)));
    }
  }
  return dz;
}
// the column in 3-D arrays: zNew comes in dzI and is replaced by dz_g, level by level
__device__ bool filtered_grid_motion_col(const mom6x_regrid_zstar_params &CS, int nz, const double *__restrict__ zOld, double *dzI,
                                         size_t x, size_t slab) {
  const double zOld1 = LV(zOld, 1);
  FilterConst F;
  if (!filter_prepare(CS, zOld1, LV(zOld, nz + 1), LV(dzI, 1), LV(dzI, nz + 1), F)) return false;
  if (F.zero) { for (int k = 1; k <= nz + 1; k++) LV(dzI, k) = 0.0; return true; }
  LV(dzI, 1) = 0.0;
  for (int k = 2; k <= nz + 1; k++) LV(dzI, k) = filter_one(F, LV(zOld, k), LV(dzI, k), zOld1);
  return true;
}

// adjust_interface_motion :1796-1857 (CS%nk == nk) on dz_int = dzI; false: "implied h<0 is larger than roundoff!"
__device__ bool adjust_interface_motion_col(double cs_min_thickness, int nk, const double *__restrict__ h_old, double *dzI, size_t x,
                                            size_t slab) {
  const double eps = DBL_EPSILON;
  double h_err = 0.;
  double dzk = LV(dzI, 1);
  for (int k = 1; k <= nk; k++) {
    const double hk = LV(h_old, k), dzn = LV(dzI, k + 1);
    h_err = h_err + dmax3(hk, fabs(dzk), fabs(dzn)) * eps;
    const double h_new = hk + (dzk - dzn);
    if (h_new < -3.0 * h_err) return false;
    dzk = dzn;
  }
  double dzn = LV(dzI, nk + 1);
  for (int k = nk; k >= 2; k--) {
    const double hk = LV(h_old, k);
    double dz = LV(dzI, k);
    double h_new = hk + (dz - dzn);
    if (h_new < cs_min_thickness) dz = (dzn - hk) + cs_min_thickness;
    h_new = hk + (dz - dzn);
    if (h_new < 0.) dz = (1. - eps) * (dzn - hk);
    h_new = hk + (dz - dzn);
    if (h_new < 0.) return false;
    LV(dzI, k) = dz;
    dzn = dz;
  }
  return true;
}

// calc_h_new_by_dz :1008-1037
__device__ void calc_h_new_col(int nz, const double *__restrict__ h, const double *__restrict__ dzI, double *__restrict__ h_new, size_t x,
                               size_t slab) {
  double dzk = LV(dzI, 1);
  for (int k = 1; k <= nz; k++) {
    const double dzn = LV(dzI, k + 1);
    LV(h_new, k) = dmax(0., LV(h, k) + (dzk - dzn));
    dzk = dzn;
  }
}

// ALE_regrid :518 -> regridding_main MOM_regridding.F90:862 (REGRIDDING_ZSTAR, no ice shelf, CS%nk == GV%ke): one thread per
// column of (isc-1..iec+1, jsc-1..jec+1); zOld lives in a 3-D scratch array and zNew in dzRegrid until the filter turns it
// into the interface displacement.  build_zstar_grid :1257-1366.
__global__ void __launch_bounds__(256)
k_regrid_zstar(Dm d, const double *__restrict__ G, mom6x_regrid_zstar_params CS, double Z_to_H, const double *__restrict__ res,
               const double *__restrict__ h, double *__restrict__ h_new, double *__restrict__ dzI, double *__restrict__ zOld,
               int *__restrict__ flag) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const int nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  if (gm(G, d, MOM6X_G_mask2dT)[x] == 0.) {                                           // :1300-1303, :1030-1033
    for (int k = 1; k <= nz; k++) { LV(h_new, k) = LV(h, k); LV(dzI, k) = 0.; }
    LV(dzI, nz + 1) = 0.;
    return;
  }
  const double depth = dmax((gm(G, d, MOM6X_G_bathyT)[x] + CS.Z_ref) * Z_to_H, 0.0);   // nom_depth_H :920-922
  double total = 0.0;
  for (int k = 1; k <= nz; k++) total = total + LV(h, k);                             // :1309-1312
  double zo = -depth;                                                                 // zOld :1315-1318
  LV(zOld, nz + 1) = zo;
  for (int k = nz; k >= 1; k--) { zo = zo + LV(h, k); LV(zOld, k) = zo; }
  // build_zstar_column coord_zlike.F90:86-142
  const double min_thickness = dmin(CS.min_thickness, total / (double)nz);
  const double eta = total - depth;
  const double stretching = total / (depth + 0.);
  double zn = eta;
  LV(dzI, 1) = zn;
  for (int k = 1; k <= nz; k++) {
    const double dh = stretching * res[k - 1] * Z_to_H;
    zn = zn - dh;
    LV(dzI, k + 1) = zn;
  }
  zn = -depth;
  LV(dzI, nz + 1) = zn;
  for (int k = nz; k >= 1; k--) {
    double zk = LV(dzI, k);
    if (zk < (zn + min_thickness)) { zk = zn + min_thickness; LV(dzI, k) = zk; }
    zn = zk;
  }
  if (!filtered_grid_motion_col(CS, nz, zOld, dzI, x, slab)) { atomicOr(flag, 4); return; }   // :1337
  if (!adjust_interface_motion_col(CS.min_thickness, nz, h, dzI, x, slab)) { atomicOr(flag, 8); return; }   // :1362
  calc_h_new_col(nz, h, dzI, h_new, x, slab);
}

// k_regrid_zstar with the column on chip (nk = NK): the thicknesses and the interface positions / displacements in registers,
// the old interface positions in LDS; the column is read once and h_new, dzInterface written once (3 words per cell-layer
// where the array version makes eight passes).  One wavefront per work-group.  Same operations in the same order.
template <int NKT>   // (NKT: mom6x_dev.h NK_OF / NK_EXACT -- the layer count itself, or a bound on it)
__global__ void __launch_bounds__(64)
k_regrid_zstar_cols(Dm d, const double *__restrict__ G, mom6x_regrid_zstar_params CS, double Z_to_H, const double *__restrict__ res,
                    const double *__restrict__ h, double *__restrict__ h_new, double *__restrict__ dzI, int *__restrict__ flag) {
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;   // (every register index below is a loop constant; slots beyond nk are skipped)
  extern __shared__ double rz_lds[];
  const int i = I_BASE(-1) + blockIdx.x * 64 + threadIdx.x;
  const int j = -1 + blockIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  double *zo_l = rz_lds + threadIdx.x;          // zOld(k) at zo_l[(k-1)*64]
  double hh[NK], dz[NK + 1];
#pragma unroll
  for (int k = 0; k < NK; k++) if (k < nk) hh[k] = h[x + (size_t)k * slab];
  if (gm(G, d, MOM6X_G_mask2dT)[x] == 0.) {
#pragma unroll
    for (int k = 0; k < NK; k++) if (k < nk) { h_new[x + (size_t)k * slab] = hh[k]; dzI[x + (size_t)k * slab] = 0.; }
    dzI[x + (size_t)nk * slab] = 0.;
    return;
  }
  const double depth = dmax((gm(G, d, MOM6X_G_bathyT)[x] + CS.Z_ref) * Z_to_H, 0.0);
  double total = 0.0;
#pragma unroll
  for (int k = 0; k < NK; k++) if (k < nk) total = total + hh[k];
  double zo = -depth;
  zo_l[nk * 64] = zo;
#pragma unroll
  for (int k = NK - 1; k >= 0; k--) if (k < nk) { zo = zo + hh[k]; zo_l[k * 64] = zo; }
  const double zOld1 = zo;
  const double min_thickness = dmin(CS.min_thickness, total / (double)nk);
  const double eta = total - depth;
  const double stretching = total / (depth + 0.);
  double zn = eta;
  dz[0] = zn;
#pragma unroll
  for (int k = 1; k <= NK; k++) if (k <= nk) { const double dh = stretching * res[k - 1] * Z_to_H; zn = zn - dh; dz[k] = zn; }
  zn = -depth;
#pragma unroll
  for (int k = NK; k >= 0; k--) {
    if (k > nk) continue;
    if (k == nk) { dz[k] = zn; continue; }          // the bottom interface sits on the bottom
    double zk = dz[k];
    if (zk < (zn + min_thickness)) { zk = zn + min_thickness; dz[k] = zk; }
    zn = zk;
  }
  FilterConst F;
  if (!filter_prepare(CS, zOld1, -depth, dz[0], -depth, F)) { atomicOr(flag, 4); return; }
  asm volatile("" ::: "memory");
  dz[0] = 0.0;
#pragma unroll
  for (int k = 1; k <= NK; k++) if (k <= nk) dz[k] = F.zero ? 0.0 : filter_one(F, zo_l[k * 64], dz[k], zOld1);
  // adjust_interface_motion :1796-1857
  {
    const double eps = DBL_EPSILON;
    double h_err = 0.;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NK; k++) if (k < nk) {
      h_err = h_err + dmax3(hh[k], fabs(dz[k]), fabs(dz[k + 1])) * eps;
      const double hn = hh[k] + (dz[k] - dz[k + 1]);
      if (hn < -3.0 * h_err) bad = true;
    }
#pragma unroll
    for (int k = NK - 1; k >= 1; k--) if (k < nk) {
      double hn = hh[k] + (dz[k] - dz[k + 1]);
      if (hn < CS.min_thickness) dz[k] = (dz[k + 1] - hh[k]) + CS.min_thickness;
      hn = hh[k] + (dz[k] - dz[k + 1]);
      if (hn < 0.) dz[k] = (1. - eps) * (dz[k + 1] - hh[k]);
      hn = hh[k] + (dz[k] - dz[k + 1]);
      if (hn < 0.) bad = true;
    }
    if (bad) { atomicOr(flag, 8); return; }
  }
#pragma unroll
  for (int k = 0; k <= NK; k++) {
    if (k < nk) h_new[x + (size_t)k * slab] = dmax(0., hh[k] + (dz[k] - dz[k + 1]));
    if (k <= nk) dzI[x + (size_t)k * slab] = dz[k];
  }
}

// ---- the density-following coordinates (REGRIDDING_RHO, REGRIDDING_HYCOM1) -------------------------------------------------
struct DensArgs {
  mom6x_regrid_rho_params CS;
  int form; double Rho_T0_S0, dRho_dT, dRho_dS, dRho_dp;   // tv%eqn_of_state
  double h_neglect;         // set_h_neglect :2602 (answer dates >= 20190101): GV%H_subroundoff, for cells and for edges
  double Z_to_H, H_to_RZ_g; // GV%Z_to_H; GV%H_to_RZ * GV%g_Earth
  int has_mid, has_mlt;
};
struct DensWork { double *zOld, *xT, *dens, *E1, *E2, *C2, *MP, *HN; };

// get_polynomial_coordinate regrid_interp.F90:376-509 (answer dates >= 20190101) on the edge values E1, E2 of the n0 source
// cells with widths hs (a 3-D array at `x`) and interfaces xg; a2, a3 of the cell that holds the target are re-formed the way
// the reconstruction that made E1, E2 writes its coefficients.  *bad: no cell holds the target (a FATAL of the reference).
__device__ double polynomial_coordinate(int scheme, int extrap, int N, const double *__restrict__ hs, const double *__restrict__ xg,
                                        const double *__restrict__ us, const DensWork &W, double target_value, size_t x, size_t slab,
                                        bool &bad) {
  if (target_value <= LV(W.E1, 1)) return LV(xg, 1);
  double e2m = LV(W.E2, 1);
  for (int k = 2; k <= N; k++) {
    const double e1 = LV(W.E1, k);
    if ((target_value >= e2m) && (target_value <= e1)) return LV(xg, k);
    e2m = LV(W.E2, k);
  }
  if (target_value >= e2m) return LV(xg, N + 1);
  int k_found = -1;
  for (int k = 1; k <= N; k++)
    if ((target_value > LV(W.E1, k)) && (target_value < LV(W.E2, k))) { k_found = k; break; }
  if (k_found == -1) { bad = true; return LV(xg, 1); }
  const double e1 = LV(W.E1, k_found), e2 = LV(W.E2, k_found);
  const double a1 = e1;
  double a2, a3 = 0.;
  if (scheme == MOM6X_INTERP_PLM) a2 = LV(W.C2, k_found);                   // PLM_reconstruction (with its almost_one factor)
  else if (scheme == MOM6X_INTERP_PPM_H4) {
    const double u = LV(us, k_found);
    if (extrap && (k_found == 1 || k_found == N)) { a2 = 6.0 * u - 4.0 * e1 - 2.0 * e2; a3 = 3.0 * (e2 + e1 - 2.0 * u); }   // PPM_boundary_extrapolation
    else { a2 = 4.0 * (u - e1) + 2.0 * (u - e2); a3 = 3.0 * ((e2 - u) + (e1 - u)); }                                        // PPM_reconstruction :46-52
  } else a2 = e2 - e1;                                                       // P1M_interpolation
  const double a4 = 0., a5 = 0., eps = 1e-6;
  double xi0 = 0.5;
  for (int iter = 1; iter <= 8; iter++) {
    const double numerator = (a1 - target_value) + xi0 * (a2 + xi0 * (a3 + xi0 * (a4 + a5 * xi0)));
    const double denominator = a2 + xi0 * (2. * a3 + xi0 * (3. * a4 + 4. * a5 * xi0));
    const double delta = -numerator / denominator;
    xi0 = xi0 + delta;
    if (xi0 < 0.0) { xi0 = 0.0; if (a2 == 0.0) xi0 = xi0 + eps; }
    if (xi0 > 1.0) { xi0 = 1.0; const double grad = a2 + (2. * a3 + (3. * a4 + 4. * a5)); if (grad == 0.0) xi0 = xi0 - eps; }
    if (fabs(delta) < 1e-12) break;
  }
  return LV(xg, k_found) + xi0 * LV(hs, k_found);
}

// build_and_interpolate_grid :331-358: regridding_set_ppolys :80 (P1M_H2, PLM, PPM_H4) + interpolate_grid :295.  Source: n0
// cells (hs, us = densities, xg = interfaces); result: the nk+1 new positions in X1 and the nk widths in W.HN.
__device__ bool build_and_interpolate_col(const DensArgs &A, int n0, const double *__restrict__ hs, const double *__restrict__ us,
                                          const double *__restrict__ xg, const double *__restrict__ tgt, int n1, double *X1,
                                          const DensWork &W, size_t x, size_t slab) {
  int scheme = A.CS.interp_scheme;
  const int extrap = A.CS.boundary_extrapolation;
  if (scheme == MOM6X_INTERP_PPM_H4 && n0 < 4) scheme = MOM6X_INTERP_P1M_H2;   // :154-170: too few cells for the h4 edges
  if (scheme == MOM6X_INTERP_P1M_H2) {
    // edge_values_explicit_h2 regrid_edge_values.F90:166-192
    double um = LV(us, 1), hm = LV(hs, 1);
    LV(W.E1, 1) = um;
    for (int k = 2; k <= n0; k++) {
      const double uk = LV(us, k), hk = LV(hs, k);
      double e;
      if (hm + hk == 0.0) e = 0.5 * (um + uk);
      else e = (um * hk + uk * hm) / (hm + hk);
      LV(W.E1, k) = e; LV(W.E2, k - 1) = e;
      um = uk; hm = hk;
    }
    LV(W.E2, n0) = um;
    // P1M_interpolation: bound_edge_values :39-101, average_discontinuous_edge_values :107-126
    for (int k = 1; k <= n0; k++) {
      const int km1 = (k - 1 > 1) ? k - 1 : 1, kp1 = (k + 1 < n0) ? k + 1 : n0;
      const double hl = LV(hs, km1), hc = LV(hs, k), hr = LV(hs, kp1), ul = LV(us, km1), uc = LV(us, k), ur = LV(us, kp1);
      double e1 = LV(W.E1, k), e2 = LV(W.E2, k);
      double slope_x_h = 0.0;
      if (((hl + hr) + 2.0 * hc) > 0.0) {
        const double sigma_l = (uc - ul);
        const double sigma_c = (ur - ul) * (hc / ((hl + hr) + 2.0 * hc));
        const double sigma_r = (ur - uc);
        if ((sigma_l * sigma_r) > 0.0) slope_x_h = fsign(dmin3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
      }
      if ((ul - e1) * (e1 - uc) < 0.0) e1 = uc - fsign(dmin(fabs(slope_x_h), fabs(e1 - uc)), slope_x_h);
      if ((ur - e2) * (e2 - uc) < 0.0) e2 = uc + fsign(dmin(fabs(slope_x_h), fabs(e2 - uc)), slope_x_h);
      e1 = dmax(dmin(e1, dmax(ul, uc)), dmin(ul, uc));
      e2 = dmax(dmin(e2, dmax(ur, uc)), dmin(ur, uc));
      LV(W.E1, k) = e1; LV(W.E2, k) = e2;
    }
    for (int k = 1; k <= n0 - 1; k++) {
      const double a = LV(W.E2, k), b = LV(W.E1, k + 1);
      if (a != b) { const double m = 0.5 * (a + b); LV(W.E2, k) = m; LV(W.E1, k + 1) = m; }
    }
    if (extrap) {   // P1M_boundary_extrapolation P1M_functions.F90:72-160
      double u0 = LV(us, 1), u1 = LV(us, 2);
      double slope = 2.0 * (u1 - u0);
      const double u0_r = u0 + 0.5 * slope;
      if ((u1 - u0) * (LV(W.E1, 2) - u0_r) < 0.0) slope = 2.0 * (LV(W.E1, 2) - u0);
      if (LV(hs, 1) != 0.0) LV(W.E1, 1) = u0 - 0.5 * slope; else LV(W.E1, 1) = u0;
      u0 = LV(us, n0 - 1); u1 = LV(us, n0);
      slope = 2.0 * (u1 - u0);
      const double u0_l = u1 - 0.5 * slope;
      if ((u1 - u0) * (u0_l - LV(W.E2, n0 - 1)) < 0.0) slope = 2.0 * (u1 - LV(W.E2, n0 - 1));
      if (LV(hs, n0) != 0.0) LV(W.E2, n0) = u1 + 0.5 * slope; else LV(W.E2, n0) = u1;
    }
  } else {
    ReconArgs R;
    R.scheme = (scheme == MOM6X_INTERP_PLM) ? MOM6X_REMAP_PLM : MOM6X_REMAP_PPM_H4;
    R.boundary_extrapolation = extrap; R.h_neglect = A.h_neglect; R.h_neglect_edge = A.h_neglect; R.n0 = n0;
    View v; v.base = x; v.lev = slab;
    reconstruct_column(R, hs, us, v, W.E1, W.E2, W.C2, nullptr, v);
  }
  // interpolate_grid :295-328
  bool bad = false;
  double xp = LV(xg, 1);
  LV(X1, 1) = xp;
  for (int k = 2; k <= n1; k++) {
    const double xk = polynomial_coordinate(scheme, extrap, n0, hs, xg, us, W, tgt[k - 1], x, slab, bad);
    LV(X1, k) = xk;
    LV(W.HN, k - 1) = xk - xp;
    xp = xk;
  }
  const double xb = LV(xg, n0 + 1);
  LV(X1, n1 + 1) = xb;
  LV(W.HN, n1) = xb - xp;
  return !bad;
}

// regridding_main :862 for REGRIDDING_RHO (HYCOM = false: build_rho_grid :1472 + build_rho_column coord_rho.F90:92) and
// REGRIDDING_HYCOM1 (HYCOM = true: build_grid_HyCOM1 :1638 + build_hycom1_column coord_hycom.F90:106, z positive downward),
// then calc_h_new_by_dz.  One thread per column of (isc-1..iec+1, jsc-1..jec+1).  h_new doubles as the array of non-vanished
// thicknesses until the end.
template <bool HYCOM>
__global__ void __launch_bounds__(256)
k_regrid_density(Dm d, const double *__restrict__ G, DensArgs A, const double *__restrict__ res, const double *__restrict__ tgt,
                 const double *__restrict__ mid, const double *__restrict__ mlt, const double *__restrict__ h,
                 const double *__restrict__ T, const double *__restrict__ S, double *h_new, double *dzI, DensWork W, int *flag) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const int nz = d.nk, nk = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double mT = gm(G, d, MOM6X_G_mask2dT)[x];
  if (HYCOM ? !(mT > 0.) : (mT == 0.)) {
    for (int k = 1; k <= nz; k++) { LV(h_new, k) = LV(h, k); LV(dzI, k) = 0.; }
    LV(dzI, nz + 1) = 0.;
    return;
  }
  const mom6x_regrid_rho_params &CS = A.CS;
  const double nominalDepth = dmax((gm(G, d, MOM6X_G_bathyT)[x] + CS.f.Z_ref) * A.Z_to_H, 0.0);
  if (!HYCOM) {
    double *HNV = h_new;
    // copy_finite_thicknesses coord_rho.F90:316-358
    int nout = 0, k_thickest = 1;
    double vanished = 0.0, thickest = LV(h, 1);
    for (int k = 1; k <= nz; k++) {
      const double hk = LV(h, k);
      LV(W.MP, k) = (double)nout;
      LV(HNV, k) = 0.;
      if (hk > CS.f.min_thickness) {
        nout = nout + 1;
        LV(W.MP, nout) = (double)k;
        LV(HNV, nout) = hk;
        if (hk > thickest) { thickest = hk; k_thickest = nout; }
      } else vanished = vanished + hk;
    }
    if (nout > 1) {
      LV(HNV, k_thickest) = LV(HNV, k_thickest) + vanished;
      double xs = 0.0;
      LV(W.xT, 1) = xs;
      for (int k = 1; k <= nout; k++) { xs = xs + LV(HNV, k); LV(W.xT, k + 1) = xs; }
      for (int k = 1; k <= nz; k++)
        LV(W.dens, k) = eos_density(A.form, A.Rho_T0_S0, A.dRho_dT, A.dRho_dS, A.dRho_dp, LV(T, k), LV(S, k), CS.ref_pressure);
      for (int k = 1; k <= nout; k++) LV(W.dens, k) = LV(W.dens, (int)LV(W.MP, k));   // (mapping(k) >= k: in place)
      if (!build_and_interpolate_col(A, nout, HNV, W.dens, W.xT, tgt, nk, dzI, W, x, slab)) { atomicOr(flag, 16); return; }
      // old_inflate_layers_1d :362-420
      int count = 0;
      for (int k = 1; k <= nk; k++) if (LV(W.HN, k) > CS.f.min_thickness) count = count + 1;
      if (count == 0) { for (int k = 1; k <= nk; k++) LV(W.HN, k) = CS.f.min_thickness; }
      else if (count != nk) {
        double correction = 0.0;
        for (int k = 1; k <= nk; k++) {
          const double hk = LV(W.HN, k);
          if (hk <= CS.f.min_thickness) { const double delta = CS.f.min_thickness - hk; correction = correction + delta; LV(W.HN, k) = hk + delta; }
        }
        double maxThickness = LV(W.HN, 1);
        int k_found = 1;
        for (int k = 1; k <= nk; k++) { const double hk = LV(W.HN, k); if (hk > maxThickness) { maxThickness = hk; k_found = k; } }
        LV(W.HN, k_found) = LV(W.HN, k_found) - correction;
      }
      // :146-150: thicknesses -> positions -> thicknesses
      double xa = 0.0;
      for (int k = 1; k <= nk; k++) { const double xb = xa + LV(W.HN, k); LV(W.HN, k) = xb - xa; xa = xb; }
    } else {
      for (int k = 1; k <= nk; k++) LV(W.HN, k) = LV(h, k);   // nz == CS%nk: "This keeps old behavior"
    }
    if (CS.integrate_downward_for_e) {
      double zn = 0., zo = 0.;
      LV(dzI, 1) = zn; LV(W.zOld, 1) = zo;
      for (int k = 1; k <= nk; k++) { zn = zn - LV(W.HN, k); LV(dzI, k + 1) = zn; zo = zo - LV(h, k); LV(W.zOld, k + 1) = zo; }
    } else {
      double zn = -nominalDepth, zo = -nominalDepth;
      LV(dzI, nk + 1) = zn; LV(W.zOld, nz + 1) = zo;
      for (int k = nk; k >= 1; k--) { zn = zn + LV(W.HN, k); LV(dzI, k) = zn; zo = zo + LV(h, k); LV(W.zOld, k) = zo; }
    }
    if (!filtered_grid_motion_col(CS.f, nz, W.zOld, dzI, x, slab)) { atomicOr(flag, 4); return; }
  } else {
    double zc = 0.0;
    LV(W.zOld, 1) = zc;   // z_col: downward from the surface
    for (int k = 1; k <= nz; k++) {
      const double zt = zc;
      zc = zc + LV(h, k);
      LV(W.zOld, k + 1) = zc;
      const double p_col = CS.ref_pressure + CS.compressibility_fraction * (0.5 * (zt + zc) * A.H_to_RZ_g - CS.ref_pressure);
      LV(W.dens, k) = eos_density(A.form, A.Rho_T0_S0, A.dRho_dT, A.dRho_dS, A.dRho_dp, LV(T, k), LV(S, k), p_col);
    }
    const double z_bot = zc;
    double rn = LV(W.dens, nz);
    for (int k = nz - 1; k >= 1; k--) { rn = dmin(LV(W.dens, k), rn); LV(W.dens, k) = rn; }   // monotonic, if not single valued
    if (!build_and_interpolate_col(A, nz, h, W.dens, W.zOld, tgt, nk, dzI, W, x, slab)) { atomicOr(flag, 16); return; }
    // at least as deep as the nominal z* grid, at most the bottom :193-201
    double nominal_z = 0.;
    const double stretching = z_bot / nominalDepth;
    for (int k = 2; k <= nk + 1; k++) {
      nominal_z = nominal_z + (A.Z_to_H * res[k - 2]) * stretching;
      double zk = dmax(LV(dzI, k), nominal_z);
      zk = dmin(zk, z_bot);
      LV(dzI, k) = zk;
    }
    if (A.has_mid || A.has_mlt) {   // :203-211
      double zm = LV(dzI, 1);
      for (int k = 2; k <= nk; k++) {
        double zk = LV(dzI, k);
        if (A.has_mid && A.has_mlt) zk = dmin3(zk, mid[k - 1], zm + mlt[k - 2]);
        else if (A.has_mid) zk = dmin(zk, mid[k - 1]);
        else zk = dmin(zk, zm + mlt[k - 2]);
        LV(dzI, k) = zk;
        zm = zk;
      }
    }
    if (!filtered_grid_motion_col(CS.f, nz, W.zOld, dzI, x, slab)) { atomicOr(flag, 4); return; }
    for (int k = 1; k <= nz + 1; k++) LV(dzI, k) = -LV(dzI, k);                        // :1713
    if (!adjust_interface_motion_col(CS.f.min_thickness, nz, h, dzI, x, slab)) { atomicOr(flag, 8); return; }
  }
  calc_h_new_col(nz, h, dzI, h_new, x, slab);
}

// convective_adjustment MOM_regridding.F90:1905-1967 over (isc-1..iec+1, jsc-1..jec+1): bubble passes until stratified
__global__ void __launch_bounds__(256)
k_convective_adjustment(Dm d, DensArgs A, double *h, double *T, double *S, double *dens) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const int nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  for (int k = 1; k <= nz; k++) LV(dens, k) = eos_density(A.form, A.Rho_T0_S0, A.dRho_dT, A.dRho_dS, A.dRho_dp, LV(T, k), LV(S, k), 0.);
  for (;;) {
    bool stratified = true;
    for (int k = 1; k <= nz - 1; k++) {
      const double r0 = LV(dens, k), r1 = LV(dens, k + 1);
      if (r0 > r1) {
        const double T0 = LV(T, k), T1 = LV(T, k + 1), S0 = LV(S, k), S1 = LV(S, k + 1), h0 = LV(h, k), h1 = LV(h, k + 1);
        LV(T, k) = T1; LV(T, k + 1) = T0; LV(S, k) = S1; LV(S, k + 1) = S0; LV(h, k) = h1; LV(h, k + 1) = h0;
        LV(dens, k) = eos_density(A.form, A.Rho_T0_S0, A.dRho_dT, A.dRho_dS, A.dRho_dp, T1, S1, 0.);
        LV(dens, k + 1) = eos_density(A.form, A.Rho_T0_S0, A.dRho_dT, A.dRho_dS, A.dRho_dp, T0, S0, 0.);
        stratified = false;
      }
    }
    if (stratified) break;
  }
}
#undef LV

int check_params(const mom6x_remapping_params *p, int n0, ReconArgs &R, ApplyArgs &A, int n1) {
  REQUIRE(p, MOM6X_EINVAL, "remapping: null parameters");
  REQUIRE(p->scheme == MOM6X_REMAP_PCM || p->scheme == MOM6X_REMAP_PLM || p->scheme == MOM6X_REMAP_PPM_H4 ||
          p->scheme == MOM6X_REMAP_PPM_IH4, MOM6X_EUNSUPPORTED,
          "MOM_remapping, build_reconstructions_1d: The selected remapping method is invalid");
  REQUIRE(p->answer_date >= 20190101, MOM6X_EUNSUPPORTED, "remapping: REMAPPING_ANSWER_DATE < 20190101 is not on the device path");
  REQUIRE(n0 >= 1 && n1 >= 1, MOM6X_EINVAL, "remapping: empty column");
  int scheme = p->scheme;                                   // :441-447
  if (n0 <= 1) scheme = MOM6X_REMAP_PCM;
  else if (n0 <= 3) scheme = std::min(scheme, (int)MOM6X_REMAP_PLM);
  else if (n0 <= 4) scheme = std::min(scheme, (int)MOM6X_REMAP_PPM_H4);
  R.scheme = scheme; R.boundary_extrapolation = p->boundary_extrapolation; R.h_neglect = p->h_neglect;
  R.h_neglect_edge = p->h_neglect_edge; R.n0 = n0;
  A.n0 = n0; A.n1 = n1; A.om4 = p->om4_remap_via_sub_cells; A.fb_sub = p->force_bounds_in_subcell; A.fb_tgt = p->force_bounds_in_target;
  A.method = (scheme == MOM6X_REMAP_PCM) ? INTEGRATION_PCM : (scheme == MOM6X_REMAP_PLM ? INTEGRATION_PLM : INTEGRATION_PPM);
  return MOM6X_OK;
}

// remap nf (1 or 2) 3-D fields that live on the same pair of grids, in place, on the points (i0..i1, j0..j1) where mask > 0
// hst != 0: h_old / h_new are the CELLS' thicknesses and the fields live at velocity points hst elements apart from their second cell
// (the OM4 switch set only: remap_fields_usable_on_cells)
static bool om4_switch_set(const mom6x_ctx *c, const mom6x_remapping_params *p) {
  ReconArgs R; ApplyArgs A;
  if (check_params(p, c->d.nk, R, A, c->d.nk)) return false;
  static const bool merge_off = [] { const char *e = getenv("MOM6X_REMAP_MERGE"); return e && !strcmp(e, "apply"); }();
  return (A.method == INTEGRATION_PPM && A.om4 && !A.fb_sub && A.fb_tgt) && !merge_off;
}
int remap_fields(mom6x_ctx *c, const mom6x_remapping_params *p, int mask_id, int i0, int i1, int j0, int j1, const double *h_old,
                 const double *h_new, double *const *f, int nf, int hst = 0) {
  const Dm d = c->d;
  ReconArgs R; ApplyArgs A;
  int rc = check_params(p, d.nk, R, A, d.nk);
  if (rc) return rc;
  // MOM6X_REMAP_MERGE=apply: the one-field streamed merge k_remap_apply for every switch set (the shared-field kernel is for OM4's)
  static const bool merge_off = [] { const char *e = getenv("MOM6X_REMAP_MERGE"); return e && !strcmp(e, "apply"); }();
  const bool om4_set = (A.method == INTEGRATION_PPM && A.om4 && !A.fb_sub && A.fb_tgt);
  const bool shared = om4_set && !merge_off;
  REQUIRE(shared || hst == 0, MOM6X_EUNSUPPORTED, "remap_fields: velocity columns on the cells' thicknesses need the OM4 switch set");
  if (!shared && nf > 1) {
    for (int m = 0; m < nf; m++) if ((rc = remap_fields(c, p, mask_id, i0, i1, j0, j1, h_old, h_new, f + m, 1))) return rc;
    return MOM6X_OK;
  }
  static const Scr work[2][4] = {{SCR_t0, SCR_t1, SCR_t3, SCR_t2}, {SCR_KE, SCR_q, SCR_absv, SCR_t2}};   // E1 (| E1 and E2 side by side), E2, Ucopy, C2 (PLM only)
  double *W[2][4];
  for (int m = 0; m < nf; m++)
    for (int a = 0; a < 4; a++) if ((rc = ctx_scratch(c, work[m][a], (shared && a == 0) ? 2 * d.nk : d.nk, &W[m][a]))) return rc;
  const dim3 b(64, 4, 1);
  const dim3 g = grid3(nxa(i1 - i0 + 1, i0), j1 - j0 + 1, 1, b);
  const double *mask = c->G + (size_t)mask_id * d.slab;
  for (int m = 0; m < nf; m++) {
    if (shared && hst) KLAUNCH(c, "k_remap_recon", (k_remap_recon<true, true>), g, b, d, mask, R, h_old, (const double *)f[m], W[m][0], W[m][1], W[m][3], W[m][2], i0, i1, j0, j1, hst);
    else if (shared) KLAUNCH(c, "k_remap_recon", k_remap_recon<true>, g, b, d, mask, R, h_old, (const double *)f[m], W[m][0], W[m][1], W[m][3], W[m][2], i0, i1, j0, j1, 0);
    else KLAUNCH(c, "k_remap_recon", k_remap_recon<false>, g, b, d, mask, R, h_old, (const double *)f[m], W[m][0], W[m][1], W[m][3], W[m][2], i0, i1, j0, j1, 0);
  }
  if (shared && nf == 2) {
    MergeFields<2> F;
    for (int m = 0; m < 2; m++) { F.E12[m] = (const double2 *)W[m][0]; F.Uc[m] = W[m][2]; F.out[m] = f[m]; }
    REQUIRE(hst == 0, MOM6X_EINVAL, "remap_fields: velocity columns go one field at a time");
    KLAUNCH(c, "k_remap_merge<2>", k_remap_merge<2>, g, b, d, mask, h_old, h_new, F, i0, i1, j0, j1, 0);
  } else if (shared) {
    MergeFields<1> F;
    F.E12[0] = (const double2 *)W[0][0]; F.Uc[0] = W[0][2]; F.out[0] = f[0];
    if (hst) KLAUNCH(c, "k_remap_merge<1>", (k_remap_merge<1, true>), g, b, d, mask, h_old, h_new, F, i0, i1, j0, j1, hst);
    else KLAUNCH(c, "k_remap_merge<1>", k_remap_merge<1>, g, b, d, mask, h_old, h_new, F, i0, i1, j0, j1, 0);
  } else if (om4_set)
    KLAUNCH(c, "k_remap_apply", k_remap_apply<1>, g, b, d, mask, A, h_old, h_new, (const double *)W[0][0], (const double *)W[0][1],
            (const double *)W[0][3], (const double *)W[0][2], f[0], i0, i1, j0, j1);
  else
    KLAUNCH(c, "k_remap_apply", k_remap_apply<0>, g, b, d, mask, A, h_old, h_new, (const double *)W[0][0], (const double *)W[0][1],
            (const double *)W[0][3], (const double *)W[0][2], f[0], i0, i1, j0, j1);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
int remap_field(mom6x_ctx *c, const mom6x_remapping_params *p, int mask_id, int i0, int i1, int j0, int j1, const double *h_old,
                const double *h_new, double *f) {
  return remap_fields(c, p, mask_id, i0, i1, j0, j1, h_old, h_new, &f, 1);
}

// ALE_PLM_edge_values, MOM_ALE.F90:1520-1577: one thread per column; slp(k-1), slp(k), slp(k+1) are carried in registers so
// that the column is read once (h, Q) and written once (Q_t, Q_b).
__global__ void __launch_bounds__(256)
k_plm_edge_values(Dm d, const double *__restrict__ h, const double *__restrict__ Q, int bdry_extrap, double hn,
                  double *__restrict__ Q_t, double *__restrict__ Q_b) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const int N = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
#define HK(k) h[x + (size_t)((k) - 1) * slab]
#define QK(k) Q[x + (size_t)((k) - 1) * slab]
  if (N >= 3) {
    double hm = HK(1), hc = HK(2), um = QK(1), uc = QK(2);
    double s_m = 0.0, s_c, s_p;
    {
      const double hp = HK(3), up = QK(3);
      s_c = PLM_slope_wa(hm, hc, hp, hn, um, uc, up);                                   // slp(2)
    }
    for (int k = 2; k <= N - 1; k++) {
      const double hp = HK(k + 1), up = QK(k + 1);
      s_p = (k + 1 <= N - 1) ? PLM_slope_wa(hc, hp, HK(k + 2), hn, uc, up, QK(k + 2)) : 0.;   // slp(k+1) (slp(N) = 0)
      const double mslp = PLM_monotonized_slope(um, uc, up, s_m, s_c, s_p);
      Q_t[x + (size_t)(k - 1) * slab] = uc - 0.5 * mslp;
      Q_b[x + (size_t)(k - 1) * slab] = uc + 0.5 * mslp;
      hm = hc; hc = hp; um = uc; uc = up; s_m = s_c; s_c = s_p;
    }
  }
  if (bdry_extrap && N >= 2) {
    double mslp = -PLM_extrapolate_slope(HK(2), HK(1), hn, QK(2), QK(1));
    Q_t[x] = QK(1) - 0.5 * mslp; Q_b[x] = QK(1) + 0.5 * mslp;
    mslp = PLM_extrapolate_slope(HK(N - 1), HK(N), hn, QK(N - 1), QK(N));
    Q_t[x + (size_t)(N - 1) * slab] = QK(N) - 0.5 * mslp; Q_b[x + (size_t)(N - 1) * slab] = QK(N) + 0.5 * mslp;
  } else {
    Q_t[x] = QK(1); Q_b[x] = QK(1);
    Q_t[x + (size_t)(N - 1) * slab] = QK(N); Q_b[x + (size_t)(N - 1) * slab] = QK(N);
  }
#undef HK
#undef QK
}
}  // namespace

// ALE_PLM_edge_values(CS, G, GV, h, Q, bdry_extrap, Q_t, Q_b), MOM_ALE.F90:1520 (answer dates >= 20190101)
extern "C" int mom6x_ALE_PLM_edge_values(mom6x_ctx *c, const double *h, const double *Q, int bdry_extrap, double *Q_t, double *Q_b) {
  REQUIRE(c && h && Q && Q_t && Q_b, MOM6X_EINVAL, "mom6x_ALE_PLM_edge_values: null argument");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b(64, 4, 1);
  KLAUNCH(c, "k_plm_edge_values", k_plm_edge_values, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, h, Q, bdry_extrap,
          c->GV.H_subroundoff, Q_t, Q_b);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// One field of TS_PPM_edge_values, MOM_ALE.F90:1581-1663 (answer dates >= 20190101): edge_values_implicit_h4 + PPM_reconstruction
// (+ PPM_boundary_extrapolation) of every column of (isc-1..iec+1, jsc-1..jec+1) -- the PPM_IH4 reconstruction of the remapping
// kernels with h_neglect = h_neglect_edge = GV%H_subroundoff and no land mask; Q_t, Q_b = ppol_E(:,1), ppol_E(:,2).
extern "C" int mom6x_ALE_PPM_edge_values(mom6x_ctx *c, const double *h, const double *Q, int bdry_extrap, double *Q_t, double *Q_b) {
  REQUIRE(c && h && Q && Q_t && Q_b, MOM6X_EINVAL, "mom6x_ALE_PPM_edge_values: null argument");
  REQUIRE(c->dims.nk >= 4, MOM6X_EUNSUPPORTED, "mom6x_ALE_PPM_edge_values: edge_values_implicit_h4 needs at least 4 layers");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *C2;
  int rc;
  if ((rc = ctx_scratch(c, SCR_KE, d.nk, &C2))) return rc;   // the tridiagonal solve's c1 column
  ReconArgs R;
  R.scheme = MOM6X_REMAP_PPM_IH4; R.boundary_extrapolation = bdry_extrap; R.h_neglect = c->GV.H_subroundoff;
  R.h_neglect_edge = c->GV.H_subroundoff; R.n0 = d.nk;
  const dim3 b(64, 4, 1);
  KLAUNCH(c, "k_remap_recon", k_remap_recon<false>, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, (const double *)nullptr, R, h, Q, Q_t, Q_b,
          C2, (double *)nullptr, -1, d.ni, -1, d.nj, 0);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

extern "C" int mom6x_ALE_remap_tracers(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old, const double *h_new,
                                       double *const *fields, int nfields) {
  REQUIRE(c && p && h_old && h_new && (fields || nfields == 0), MOM6X_EINVAL, "ALE_remap_tracers: null argument");
  HIPCHK(hipSetDevice(c->device));
  for (int m = 0; m < nfields; m++) REQUIRE(fields[m], MOM6X_EINVAL, "ALE_remap_tracers: null tracer array");
  for (int m = 0; m < nfields; m += 2) {     // the tracers share their grids: two at a time through one merge
    int rc = remap_fields(c, p, MOM6X_G_mask2dT, 0, c->d.ni - 1, 0, c->d.nj - 1, h_old, h_new, fields + m, std::min(2, nfields - m));
    if (rc) return rc;
  }
  return MOM6X_OK;
}

extern "C" int mom6x_ALE_remap_set_h_vel(mom6x_ctx *c, const double *h_new, double *h_u, double *h_v) {
  REQUIRE(c && h_new && h_u && h_v, MOM6X_EINVAL, "ALE_remap_set_h_vel: null argument");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b(64, 4, 1);
  KLAUNCH(c, "k_set_h_vel", k_set_h_vel, grid3(nxa(d.ni + 1, -1), d.nj + 1, nchunks(d.nk), b), b, d, c->G, h_new, h_u, h_v);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// The KE-conserving correction of ALE_remap_velocities :1166-1195 (REMAP_VEL_CONSERVE_KE with allow_preserve_variance): one thread
// per velocity column, three walks over k in the reference's order (u_bt and sum(h2); the two baroclinic KE integrals; the rescaling).
__global__ void __launch_bounds__(256)
k_remap_conserve_ke(Dm d, const double *__restrict__ G, int mask_plane, int i0, int i1, int j0, int j1, const double *__restrict__ h1,
                    const double *__restrict__ h2, const double *__restrict__ u_src, double *__restrict__ u_tgt, double H_subroundoff) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > i1 || j > j1) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  if (!(gm(G, d, mask_plane)[x] > 0.0)) return;
  const int nz = d.nk;
  double u_bt = 0.0, hsum = 0.0;
  for (int k = 0; k < nz; k++) { const size_t c = x + (size_t)k * slab; u_bt = u_bt + h2[c] * u_tgt[c]; hsum = hsum + h2[c]; }
  u_bt = u_bt / (hsum + H_subroundoff);
  double ke_c_src = 0.0, ke_c_tgt = 0.0;
  for (int k = 0; k < nz; k++) {
    const size_t c = x + (size_t)k * slab;
    const double a = u_src[c] - u_bt, b = u_tgt[c] - u_bt;
    ke_c_src = ke_c_src + h1[c] * (a * a);
    ke_c_tgt = ke_c_tgt + h2[c] * (b * b);
  }
  // (the 25 % cap on the amplification of the baroclinic part, :1182-1191)
  const double rescale_coef = (ke_c_src < 1.5625 * ke_c_tgt) ? sqrt(ke_c_src / ke_c_tgt) : 1.25;
  for (int k = 0; k < nz; k++) { const size_t c = x + (size_t)k * slab; u_tgt[c] = u_bt + rescale_coef * (u_tgt[c] - u_bt); }
}

static int remap_velocities(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old_u, const double *h_old_v,
                            const double *h_new_u, const double *h_new_v, double *u, double *v, bool conserve_ke) {
  REQUIRE(c && p && h_old_u && h_old_v && h_new_u && h_new_v && u && v, MOM6X_EINVAL, "ALE_remap_velocities: null argument");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const size_t n3 = (size_t)d.slab * d.nk;
  // The source column of the correction: allocated by the FIRST call with REMAP_VEL_CONSERVE_KE (a synchronising hipMalloc, once per
  // context; the entry point has no initialisation of its own), kept until the context goes.
  // ORDER: the reference masks the near-bottom velocities (mask_near_bottom_vel, BBL_h_vel_mask / h_vel_mask; MOM_ALE.F90:1194-1200)
  // AFTER the correction.  The device carries neither mask (both default to 0: nothing is masked); whoever adds them must apply them
  // after k_remap_conserve_ke, not inside remap_field.
  if (conserve_ke && !c->remap_src) HIPCHK(hipMalloc(&c->remap_src, n3 * sizeof(double)));
  const dim3 b(64, 4, 1);
  if (conserve_ke) HIPCHK(hipMemcpyAsync(c->remap_src, u, n3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  int rc = remap_field(c, p, MOM6X_G_mask2dCu, -1, d.ni - 1, 0, d.nj - 1, h_old_u, h_new_u, u);
  if (rc) return rc;
  if (conserve_ke)
    KLAUNCH(c, "k_remap_conserve_ke", k_remap_conserve_ke, grid3(d.ni + 1, d.nj, 1, b), b, d, c->G, (int)MOM6X_G_mask2dCu, -1, d.ni - 1, 0, d.nj - 1,
            h_old_u, h_new_u, (const double *)c->remap_src, u, c->GV.H_subroundoff);
  if (conserve_ke) HIPCHK(hipMemcpyAsync(c->remap_src, v, n3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  rc = remap_field(c, p, MOM6X_G_mask2dCv, 0, d.ni - 1, -1, d.nj - 1, h_old_v, h_new_v, v);
  if (rc) return rc;
  if (conserve_ke)
    KLAUNCH(c, "k_remap_conserve_ke", k_remap_conserve_ke, grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, (int)MOM6X_G_mask2dCv, 0, d.ni - 1, -1, d.nj - 1,
            h_old_v, h_new_v, (const double *)c->remap_src, v, c->GV.H_subroundoff);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

extern "C" int mom6x_ALE_remap_velocities(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old_u, const double *h_old_v,
                                          const double *h_new_u, const double *h_new_v, double *u, double *v) {
  return remap_velocities(c, p, h_old_u, h_old_v, h_new_u, h_new_v, u, v, false);
}
// ALE_remap_set_h_vel(h_old) + ALE_remap_set_h_vel(h_new) + ALE_remap_velocities as MOM.F90 calls them one after the other
// (ALE_regridding_and_remapping), from the CELLS' thicknesses: with OM4's switch set the thicknesses at the velocity points are
// formed where the remapping reads them (the same expression: the same bits) and the four h_u / h_v arrays are never written or
// read -- 6 words per cell-layer less.  Any other switch set: the arrays are made in work space and the call above follows.
extern "C" int mom6x_ALE_remap_velocities_from_h(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old, const double *h_new,
                                                 double *u, double *v) {
  REQUIRE(c && p && h_old && h_new && u && v, MOM6X_EINVAL, "ALE_remap_velocities: null argument");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  if (om4_switch_set(c, p)) {
    int rc = remap_fields(c, p, MOM6X_G_mask2dCu, -1, d.ni - 1, 0, d.nj - 1, h_old, h_new, &u, 1, 1);
    if (rc) return rc;
    return remap_fields(c, p, MOM6X_G_mask2dCv, 0, d.ni - 1, -1, d.nj - 1, h_old, h_new, &v, 1, d.pitch);
  }
  const size_t n3 = (size_t)d.slab * d.nk;
  if (!c->remap_hvel) { HIPCHK(hipMalloc(&c->remap_hvel, 4 * n3 * sizeof(double))); HIPCHK(hipMemsetAsync(c->remap_hvel, 0, 4 * n3 * sizeof(double), c->stream)); }
  double *hu0 = c->remap_hvel, *hv0 = hu0 + n3, *hu1 = hv0 + n3, *hv1 = hu1 + n3;
  int rc = mom6x_ALE_remap_set_h_vel(c, h_old, hu0, hv0);
  if (rc) return rc;
  if ((rc = mom6x_ALE_remap_set_h_vel(c, h_new, hu1, hv1))) return rc;
  return remap_velocities(c, p, hu0, hv0, hu1, hv1, u, v, false);
}
// ... with allow_preserve_variance = .true. and REMAP_VEL_CONSERVE_KE (CS%conserve_ke, MOM_ALE.F90:331): MOM.F90's call in the time step
extern "C" int mom6x_ALE_remap_velocities_conserve_ke(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old_u,
                                                      const double *h_old_v, const double *h_new_u, const double *h_new_v, double *u,
                                                      double *v) {
  return remap_velocities(c, p, h_old_u, h_old_v, h_new_u, h_new_v, u, v, true);
}

extern "C" int mom6x_ALE_regrid_zstar(mom6x_ctx *c, const mom6x_regrid_zstar_params *p, const double *coordinateResolution,
                                      const double *h, double *h_new, double *dzRegrid) {
  REQUIRE(c && p && coordinateResolution && h && h_new && dzRegrid, MOM6X_EINVAL, "ALE_regrid: null argument");
  REQUIRE(p->old_grid_weight >= 0.0 && p->old_grid_weight < 1.0, MOM6X_EINVAL, "ALE_regrid: old_grid_weight must be in [0, 1)");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  REQUIRE(d.halo >= 1, MOM6X_EINVAL, "ALE_regrid: one halo point needed");
  if (!c->regrid_res) HIPCHK(hipMalloc(&c->regrid_res, (size_t)d.nk * sizeof(double)));
  HIPCHK(hipMemcpyAsync(c->regrid_res, coordinateResolution, (size_t)d.nk * sizeof(double), hipMemcpyHostToDevice, c->stream));
  double *zOld;
  int rc = ctx_scratch(c, SCR_e, d.nk + 1, &zOld);
  if (rc) return rc;
  const dim3 b(64, 4, 1);
  if (d.nk <= COLS_NK_BOUND) {   // the layer counts the on-chip column kernel is built for
#define RZC(NKT) KLAUNCH_LDS(c, "k_regrid_zstar", (k_regrid_zstar_cols<NKT>), dim3((unsigned)((nxa(d.ni + 2, -1) + 63) / 64), (unsigned)(d.nj + 2), 1), dim3(64, 1, 1), \
                (size_t)(NK_OF(NKT) + 1) * 64 * sizeof(double), d, c->G, *p, c->GV.Z_to_H, (const double *)c->regrid_res, h, h_new, dzRegrid, c->flag)
    COLS_NK_DISPATCH(d.nk, RZC);
#undef RZC
  } else
  KLAUNCH(c, "k_regrid_zstar", k_regrid_zstar, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, c->G, *p, c->GV.Z_to_H,
          (const double *)c->regrid_res, h, h_new, dzRegrid, zOld, c->flag);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (flag & REGRID_FLAGS) { const int keep = flag & ~REGRID_FLAGS; HIPCHK(hipMemcpy(c->flag, &keep, sizeof(int), hipMemcpyHostToDevice)); }
  REQUIRE(!(flag & 4), MOM6X_EINVAL, "filtered_grid_motion: z_old and z_new use different sign conventions.");
  REQUIRE(!(flag & 8), MOM6X_EINVAL, "MOM_regridding: adjust_interface_motion() - implied h<0 is larger than roundoff!");
  return MOM6X_OK;
}

// REGRIDDING_RHO / REGRIDDING_HYCOM1: the host side the two entry points share
static int regrid_density(mom6x_ctx *c, bool hycom, const mom6x_regrid_rho_params *p, const mom6x_eos_params *eos,
                          const double *coordinateResolution, const double *target_density, const double *max_interface_depths,
                          const double *max_layer_thickness, const double *h, const double *T, const double *S, double *h_new,
                          double *dzRegrid) {
  REQUIRE(c && p && eos && target_density && h && T && S && h_new && dzRegrid && (!hycom || coordinateResolution), MOM6X_EINVAL,
          "ALE_regrid: null argument");
  REQUIRE(p->f.old_grid_weight >= 0.0 && p->f.old_grid_weight < 1.0, MOM6X_EINVAL, "ALE_regrid: old_grid_weight must be in [0, 1)");
  REQUIRE(p->interp_scheme == MOM6X_INTERP_P1M_H2 || p->interp_scheme == MOM6X_INTERP_PLM || p->interp_scheme == MOM6X_INTERP_PPM_H4,
          MOM6X_EUNSUPPORTED, "regrid_interp: INTERPOLATION_SCHEME must be P1M_H2, PLM or PPM_H4 on the device path");
  REQUIRE(eos->form >= MOM6X_EOS_LINEAR && eos->form <= MOM6X_EOS_ROQUET_SPV, MOM6X_EUNSUPPORTED, "ALE_regrid: every EQN_OF_STATE but TEOS10 is carried");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  REQUIRE(d.halo >= 1, MOM6X_EINVAL, "ALE_regrid: one halo point needed");
  REQUIRE(d.nk >= 2, MOM6X_EINVAL, "ALE_regrid: a density coordinate needs at least two layers");
  // host vectors -> one device buffer: res (nk) | target (nk+1) | max depths (nk+1) | max thickness (nk)
  const size_t nk = (size_t)d.nk, nv = 4 * nk + 2;
  if (!c->regrid_vec) HIPCHK(hipMalloc(&c->regrid_vec, nv * sizeof(double)));
  std::vector<double> hv(nv, 0.0);
  if (coordinateResolution) for (size_t k = 0; k < nk; k++) hv[k] = coordinateResolution[k];
  for (size_t k = 0; k <= nk; k++) hv[nk + k] = target_density[k];
  if (max_interface_depths) for (size_t k = 0; k <= nk; k++) hv[2 * nk + 1 + k] = max_interface_depths[k];
  if (max_layer_thickness) for (size_t k = 0; k < nk; k++) hv[3 * nk + 2 + k] = max_layer_thickness[k];
  HIPCHK(hipMemcpyAsync(c->regrid_vec, hv.data(), nv * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));   // (hv leaves scope)
  DensArgs A;
  A.CS = *p;
  A.form = eos->form; A.Rho_T0_S0 = eos->Rho_T0_S0; A.dRho_dT = eos->dRho_dT; A.dRho_dS = eos->dRho_dS; A.dRho_dp = eos->dRho_dp;
  A.h_neglect = c->GV.H_subroundoff; A.Z_to_H = c->GV.Z_to_H; A.H_to_RZ_g = c->GV.H_to_RZ * c->GV.g_Earth;
  A.has_mid = max_interface_depths ? 1 : 0; A.has_mlt = max_layer_thickness ? 1 : 0;
  DensWork W;
  int rc;
  if ((rc = ctx_scratch(c, SCR_e, d.nk + 1, &W.zOld)) || (rc = ctx_scratch(c, SCR_c1, d.nk + 1, &W.xT)) ||
      (rc = ctx_scratch(c, SCR_q, d.nk, &W.dens)) || (rc = ctx_scratch(c, SCR_t0, d.nk, &W.E1)) || (rc = ctx_scratch(c, SCR_t1, d.nk, &W.E2)) ||
      (rc = ctx_scratch(c, SCR_t2, d.nk, &W.C2)) || (rc = ctx_scratch(c, SCR_KE, d.nk, &W.MP)) || (rc = ctx_scratch(c, SCR_absv, d.nk, &W.HN)))
    return rc;
  const dim3 b(64, 4, 1);
  const dim3 g = grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b);
  const double *res = c->regrid_vec, *tgt = res + nk, *mid = tgt + nk + 1, *mlt = mid + nk + 1;
  if (hycom) KLAUNCH(c, "k_regrid_density<hycom1>", k_regrid_density<true>, g, b, d, c->G, A, res, tgt, mid, mlt, h, T, S, h_new, dzRegrid, W, c->flag);
  else KLAUNCH(c, "k_regrid_density<rho>", k_regrid_density<false>, g, b, d, c->G, A, res, tgt, mid, mlt, h, T, S, h_new, dzRegrid, W, c->flag);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (flag & REGRID_FLAGS) { const int keep = flag & ~REGRID_FLAGS; HIPCHK(hipMemcpy(c->flag, &keep, sizeof(int), hipMemcpyHostToDevice)); }
  REQUIRE(!(flag & 4), MOM6X_EINVAL, "filtered_grid_motion: z_old and z_new use different sign conventions.");
  REQUIRE(!(flag & 8), MOM6X_EINVAL, "MOM_regridding: adjust_interface_motion() - implied h<0 is larger than roundoff!");
  REQUIRE(!(flag & 16), MOM6X_EINVAL, "Could not find target coordinate in get_polynomial_coordinate. This is caused by an inconsistent interpolant (perhaps not monotonically increasing)");
  return MOM6X_OK;
}

extern "C" int mom6x_ALE_regrid_rho(mom6x_ctx *c, const mom6x_regrid_rho_params *p, const mom6x_eos_params *eos, const double *target_density,
                                    const double *h, const double *T, const double *S, double *h_new, double *dzRegrid) {
  return regrid_density(c, false, p, eos, nullptr, target_density, nullptr, nullptr, h, T, S, h_new, dzRegrid);
}

extern "C" int mom6x_ALE_regrid_hycom1(mom6x_ctx *c, const mom6x_regrid_rho_params *p, const mom6x_eos_params *eos,
                                       const double *coordinateResolution, const double *target_density, const double *max_interface_depths,
                                       const double *max_layer_thickness, const double *h, const double *T, const double *S, double *h_new,
                                       double *dzRegrid) {
  return regrid_density(c, true, p, eos, coordinateResolution, target_density, max_interface_depths, max_layer_thickness, h, T, S, h_new,
                        dzRegrid);
}

extern "C" int mom6x_ALE_convective_adjustment(mom6x_ctx *c, const mom6x_eos_params *eos, double *h, double *T, double *S) {
  REQUIRE(c && eos && h && T && S, MOM6X_EINVAL, "convective_adjustment: null argument");
  REQUIRE(eos->form >= MOM6X_EOS_LINEAR && eos->form <= MOM6X_EOS_ROQUET_SPV, MOM6X_EUNSUPPORTED, "convective_adjustment: every EQN_OF_STATE but TEOS10 is carried");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  REQUIRE(d.halo >= 1, MOM6X_EINVAL, "convective_adjustment: one halo point needed");
  DensArgs A;
  memset(&A, 0, sizeof(A));
  A.form = eos->form; A.Rho_T0_S0 = eos->Rho_T0_S0; A.dRho_dT = eos->dRho_dT; A.dRho_dS = eos->dRho_dS; A.dRho_dp = eos->dRho_dp;
  double *dens;
  int rc = ctx_scratch(c, SCR_q, d.nk, &dens);
  if (rc) return rc;
  const dim3 b(64, 4, 1);
  KLAUNCH(c, "k_convective_adjustment", k_convective_adjustment, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, A, h, T, S, dens);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

extern "C" int mom6x_remapping_core_h(mom6x_ctx *c, const mom6x_remapping_params *p, int ncol, int n0, const double *h0,
                                      const double *u0, int n1, const double *h1, double *u1) {
  REQUIRE(c && h0 && u0 && h1 && u1 && ncol >= 1, MOM6X_EINVAL, "remapping_core_h: null argument");
  HIPCHK(hipSetDevice(c->device));
  ReconArgs R; ApplyArgs A;
  int rc = check_params(p, n0, R, A, n1);
  if (rc) return rc;
  double *w = nullptr;
  const size_t per = (size_t)n0 * ncol;
  HIPCHK(hipMalloc(&w, 3 * per * sizeof(double)));
  KLAUNCH(c, "k_remap_packed", k_remap_packed, dim3((ncol + 63) / 64), dim3(64), ncol, R, A, h0, u0, h1, u1, w, w + per, w + 2 * per);
  hipError_t e = hipStreamSynchronize(c->stream);
  (void)hipFree(w);
  HIPCHK(e);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
