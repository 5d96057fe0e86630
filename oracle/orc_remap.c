/* orc_remap.c -- MOM_remapping's remapping_core_h with the OM4-era reconstruction functions PCM, PLM and PPM_H4
 * (answer dates >= 20190101), and the remapping half of MOM_ALE (ALE_remap_tracers, ALE_remap_set_h_vel,
 * ALE_remap_velocities).  ORACLE (test infrastructure only; see orc_common.h).
 *
 * PARITY PINNED: unlike the rest of the oracle, the reference holds known answers for these routines
 * (remapping_unit_tests, src/ALE/MOM_remapping.F90:2072-2943); tests/test_remap_cpu.py replays them.  The PLM and PCM
 * reconstructions, the PPM_H4 chain and the sub-grid integration are in addition held bit for bit to the reference's own
 * code, compiled where it lies from the files that need no FMS (PLM_functions.F90, PCM_functions.F90,
 * Recon1d_PPM_H4_2019.F90 + Recon1d_type.F90 + numerical_testing_type.F90): oracle/_ref, `make ref`.
 *
 * Arrays are 1-based inside this file (index 0 unused) so that the index arithmetic of intersect_src_tgt_grids and
 * the sub-cell loops reads exactly as in the reference. */
#include <math.h>
#include "orc_common.h"
#include <float.h>

static inline double fsign(double a, double b) { return copysign(fabs(a), b); }   /* Fortran sign() with signed zeros */
static inline double max3(double a, double b, double c) { return orc_max(orc_max(a, b), c); }
static inline double min3(double a, double b, double c) { return orc_min(orc_min(a, b), c); }

enum { INTEGRATION_PCM = 0, INTEGRATION_PLM = 1, INTEGRATION_PPM = 3 };

/* ---- PLM_functions.F90 ------------------------------------------------------------------------------------- */
static double PLM_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r) {   /* :27-66 */
  const double sigma_r = u_r - u_c, sigma_l = u_c - u_l;
  const double sigma_c = 2.0 * (u_r - u_l) * (h_c / (h_l + 2.0 * h_c + h_r + h_neglect));
  const double u_min = min3(u_l, u_c, u_r), u_max = max3(u_l, u_c, u_r);
  double s;
  if ((sigma_l * sigma_r) > 0.0) s = fsign(orc_min(fabs(sigma_c), 2. * orc_min(u_c - u_min, u_max - u_c)), sigma_c);
  else s = 0.0;
  if (u_c - 0.5 * fabs(s) < u_min || u_c + 0.5 * fabs(s) > u_max) s = s * (1. - DBL_EPSILON);
  if (fabs(s) < 1.E-140) s = 0.;
  return s;
}
static double PLM_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r) {   /* :124-162 */
  const double almost_two = 2. * (1. - DBL_EPSILON);
  const double e_r = u_l + 0.5 * s_l, e_l = u_r - 0.5 * s_r;
  double slp = fabs(s_c);
  double edge = u_c - 0.5 * s_c;
  if ((edge - e_r) * (u_c - edge) < 0.) { edge = 0.5 * (edge + e_r); slp = orc_min(slp, fabs(edge - u_c) * almost_two); }
  edge = u_c + 0.5 * s_c;
  if ((edge - u_c) * (e_l - edge) < 0.) { edge = 0.5 * (edge + e_l); slp = orc_min(slp, fabs(edge - u_c) * almost_two); }
  return fsign(slp, s_c);
}
static double PLM_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c) {   /* :170-191 */
  const double hl = h_l + h_neglect, hc = h_c + h_neglect;
  const double left_edge = (u_l * hc + u_c * hl) / (hl + hc);
  return 2.0 * (u_c - left_edge);
}
double orc_PLM_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r) {
  return PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r);
}
double orc_PLM_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r) {
  return PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r);
}
double orc_PLM_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c) {
  return PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c);
}
/* PLM_reconstruction :197-262 */
void orc_PLM_reconstruction(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double h_neglect) {
  const double almost_one = 1. - DBL_EPSILON;
  double *slp = (double *)calloc(N + 2, sizeof(double)), *mslp = (double *)calloc(N + 2, sizeof(double));
  for (int k = 2; k <= N - 1; k++) slp[k] = PLM_slope_wa(h[k - 1], h[k], h[k + 1], h_neglect, u[k - 1], u[k], u[k + 1]);
  slp[1] = 0.; slp[N] = 0.;
  for (int k = 2; k <= N - 1; k++) mslp[k] = PLM_monotonized_slope(u[k - 1], u[k], u[k + 1], slp[k - 1], slp[k], slp[k + 1]);
  mslp[1] = 0.; mslp[N] = 0.;
  E1[1] = u[1]; E2[1] = u[1]; C1[1] = u[1]; C2[1] = 0.;
  for (int k = 2; k <= N - 1; k++) {
    const double slope = mslp[k];
    const double u_l = u[k] - 0.5 * slope, u_r = u[k] + 0.5 * slope;
    E1[k] = u_l; E2[k] = u_r;
    C1[k] = u_l; C2[k] = (u_r - u_l);
    const double edge = C2[k] + C1[k];
    const double e_r = u[k + 1] - 0.5 * fsign(mslp[k + 1], slp[k + 1]);
    if ((edge - u[k]) * (e_r - edge) < 0.) C2[k] = C2[k] * almost_one;
  }
  E1[N] = u[N]; E2[N] = u[N]; C1[N] = u[N]; C2[N] = 0.;
  free(slp); free(mslp);
}
/* PLM_boundary_extrapolation :274-308 */
void orc_PLM_boundary_extrapolation(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double h_neglect) {
  double slope = -PLM_extrapolate_slope(h[2], h[1], h_neglect, u[2], u[1]);
  E1[1] = u[1] - 0.5 * slope; E2[1] = u[1] + 0.5 * slope;
  C1[1] = E1[1]; C2[1] = E2[1] - E1[1];
  slope = PLM_extrapolate_slope(h[N - 1], h[N], h_neglect, u[N - 1], u[N]);
  E1[N] = u[N] - 0.5 * slope; E2[N] = u[N] + 0.5 * slope;
  C1[N] = E1[N]; C2[N] = E2[N] - E1[N];
}

/* ---- regrid_edge_values.F90 -------------------------------------------------------------------------------- */
static const double hMinFrac = 1.e-5;   /* :25 */
/* bound_edge_values :39-101 (answer dates >= 20190101) */
static void bound_edge_values(int N, const double *h, const double *u, double *E1, double *E2, double h_neglect) {
  (void)h_neglect;
  for (int k = 1; k <= N; k++) {
    const int km1 = (k - 1 > 1) ? k - 1 : 1, kp1 = (k + 1 < N) ? k + 1 : N;
    double slope_x_h = 0.0;
    if (((h[km1] + h[kp1]) + 2.0 * h[k]) > 0.0) {
      const double sigma_l = (u[k] - u[km1]);
      const double sigma_c = (u[kp1] - u[km1]) * (h[k] / ((h[km1] + h[kp1]) + 2.0 * h[k]));
      const double sigma_r = (u[kp1] - u[k]);
      if ((sigma_l * sigma_r) > 0.0) slope_x_h = fsign(min3(fabs(sigma_l), fabs(sigma_c), fabs(sigma_r)), sigma_c);
    }
    if ((u[km1] - E1[k]) * (E1[k] - u[k]) < 0.0) E1[k] = u[k] - fsign(orc_min(fabs(slope_x_h), fabs(E1[k] - u[k])), slope_x_h);
    if ((u[kp1] - E2[k]) * (E2[k] - u[k]) < 0.0) E2[k] = u[k] + fsign(orc_min(fabs(slope_x_h), fabs(E2[k] - u[k])), slope_x_h);
    E1[k] = orc_max(orc_min(E1[k], orc_max(u[km1], u[k])), orc_min(u[km1], u[k]));
    E2[k] = orc_max(orc_min(E2[k], orc_max(u[kp1], u[k])), orc_min(u[kp1], u[k]));
  }
}
/* check_discontinuous_edge_values :132-150 */
static void check_discontinuous_edge_values(int N, const double *u, double *E1, double *E2) {
  for (int k = 1; k <= N - 1; k++) {
    if ((E1[k + 1] - E2[k]) * (u[k + 1] - u[k]) < 0.0) {
      double u0_avg = 0.5 * (E2[k] + E1[k + 1]);
      u0_avg = orc_max(orc_min(u0_avg, orc_max(u[k], u[k + 1])), orc_min(u[k], u[k + 1]));
      E2[k] = u0_avg; E1[k + 1] = u0_avg;
    }
  }
}
/* end_value_h4 :634-747; dz, u, Csys are 0-based arrays of 4 */
static void end_value_h4(const double *dz, const double *u, double *Csys) {
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
  const double I_h34 = (h1234 * h234) * I_denom;
  const double I_h234 = (h1234 * h34) * I_denom;
  const double I_h1234 = (h234 * h34) * I_denom;
  (void)I_h34;
  double Wt[3][4];
  Wt[0][0] = -h1 * (I_h1234 + I_h123 + I_h12);
  Wt[1][0] = h1 * h12 * (I_h234 * I_h1234 + I_h23 * (I_h234 + I_h123));
  Wt[2][0] = -h1 * h12 * h123 * I_denom;
  Wt[0][1] = 2.0 * (I_h12 * (1.0 + (h1 + h12) * (I_h1234 + I_h123)) + h1 * I_h1234 * I_h123);
  Wt[1][1] = -2.0 * ((h1 * h12 * I_h1234) * (I_h23 * (I_h234 + I_h123)) + (h1 + h12) * (I_h1234 * I_h234 + I_h23 * (I_h234 + I_h123)));
  Wt[2][1] = 2.0 * ((h1 + h12) * h123 + h1 * h12) * I_denom;
  Wt[0][2] = -3.0 * I_h12 * I_h123 * (1.0 + I_h1234 * ((h1 + h12) + h123));
  Wt[1][2] = 3.0 * I_h23 * (I_h123 + I_h1234 * ((h1 + h12) + h123) * (I_h123 + I_h234));
  Wt[2][2] = -3.0 * ((h1 + h12) + h123) * I_denom;
  Wt[0][3] = 4.0 * I_h1234 * I_h123 * I_h12;
  Wt[1][3] = -4.0 * I_h1234 * (I_h23 * (I_h123 + I_h234));
  Wt[2][3] = 4.0 * I_denom;
  Csys[0] = ((u[0] + (Wt[0][0] * (u[1] - u[0]))) + (Wt[1][0] * (u[2] - u[1]))) + (Wt[2][0] * (u[3] - u[2]));
  Csys[1] = ((Wt[0][1] * (u[1] - u[0])) + (Wt[1][1] * (u[2] - u[1]))) + (Wt[2][1] * (u[3] - u[2]));
  Csys[2] = ((Wt[0][2] * (u[1] - u[0])) + (Wt[1][2] * (u[2] - u[1]))) + (Wt[2][2] * (u[3] - u[2]));
  Csys[3] = ((Wt[0][3] * (u[1] - u[0])) + (Wt[1][3] * (u[2] - u[1]))) + (Wt[2][3] * (u[3] - u[2]));
}
/* edge_values_explicit_h4 :213-348 (answer dates >= 20190101); N >= 4 */
void orc_edge_values_explicit_h4(int N, const double *h, const double *u, double *E1, double *E2, double h_neglect) {
  for (int i = 3; i <= N - 1; i++) {
    double h0 = h[i - 2], h1 = h[i - 1], h2 = h[i], h3 = h[i + 1];
    if (h0 + h1 == 0.0 || h1 + h2 == 0.0 || h2 + h3 == 0.0) {
      const double h_min = hMinFrac * orc_max(h_neglect, (h0 + h1) + (h2 + h3));
      h0 = orc_max(h_min, h[i - 2]); h1 = orc_max(h_min, h[i - 1]); h2 = orc_max(h_min, h[i]); h3 = orc_max(h_min, h[i + 1]);
    }
    const double I_h12 = 1.0 / (h1 + h2);
    const double I_den_et2 = 1.0 / (((h0 + h1) + h2) * (h0 + h1)), I_h012 = (h0 + h1) * I_den_et2;
    const double I_den_et3 = 1.0 / ((h1 + (h2 + h3)) * (h2 + h3)), I_h123 = (h2 + h3) * I_den_et3;
    const double et1 = (1.0 + (h1 * I_h012 + (h0 + h1) * I_h123)) * I_h12 * (h2 * (h2 + h3)) * u[i - 1] +
                       (1.0 + (h2 * I_h123 + (h2 + h3) * I_h012)) * I_h12 * (h1 * (h0 + h1)) * u[i];
    const double et2 = (h1 * (h2 * (h2 + h3)) * I_den_et2) * (u[i - 1] - u[i - 2]);
    const double et3 = (h2 * (h1 * (h0 + h1)) * I_den_et3) * (u[i] - u[i + 1]);
    E1[i] = (et1 + (et2 + et3)) / ((h0 + h1) + (h2 + h3));
    E2[i - 1] = E1[i];
  }
  double dz[4], ut[4], C[4];
  for (int i = 1; i <= 4; i++) { dz[i - 1] = orc_max(h_neglect, h[i]); ut[i - 1] = u[i]; }
  end_value_h4(dz, ut, C);
  E1[1] = C[0];
  E2[1] = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
  E1[2] = E2[1];
  for (int i = 1; i <= 4; i++) { dz[i - 1] = orc_max(h_neglect, h[N + 1 - i]); ut[i - 1] = u[N + 1 - i]; }
  end_value_h4(dz, ut, C);
  E2[N] = C[0];
  E1[N] = C[0] + dz[0] * (C[1] + dz[0] * (C[2] + dz[0] * C[3]));
  E2[N - 1] = E1[N];
}

/* solve_diag_dominant_tridiag regrid_solvers.F90:246-280; 1-based arrays of N entries */
static void solve_diag_dominant_tridiag(const double *Al, const double *Ac, const double *Au, const double *R, double *X, int N) {
  double *c1 = (double *)calloc((size_t)N + 2, sizeof(double));
  double I_pivot = 1.0 / (Ac[1] + Au[1]);
  double d1 = Ac[1] * I_pivot;
  c1[1] = Au[1] * I_pivot;
  X[1] = R[1] * I_pivot;
  for (int k = 2; k <= N - 1; k++) {
    const double denom_t1 = Ac[k] + d1 * Al[k];
    I_pivot = 1.0 / (denom_t1 + Au[k]);
    d1 = denom_t1 * I_pivot;
    c1[k] = Au[k] * I_pivot;
    X[k] = (R[k] - Al[k] * X[k - 1]) * I_pivot;
  }
  I_pivot = 1.0 / (Ac[N] + d1 * Al[N]);
  X[N] = (R[N] - Al[N] * X[N - 1]) * I_pivot;
  for (int k = N - 1; k >= 1; k--) X[k] = X[k] - c1[k] * X[k + 1];
  free(c1);
}
/* edge_values_implicit_h4 :473-630 (answer dates >= 20190101); N >= 4 */
void orc_edge_values_implicit_h4(int N, const double *h, const double *u, double *E1, double *E2, double h_neglect) {
  double *w = (double *)calloc((size_t)(5 * (N + 3)), sizeof(double));
  double *tri_l = w, *tri_c = tri_l + N + 3, *tri_u = tri_c + N + 3, *tri_b = tri_u + N + 3, *tri_x = tri_b + N + 3;
  for (int i = 1; i <= N - 1; i++) {
    double h0 = orc_max(h[i], h_neglect), h1 = orc_max(h[i + 1], h_neglect);
    if (fabs(h0) < 1.0e-12 * fabs(h1)) h0 = 1.0e-12 * h1;
    if (fabs(h1) < 1.0e-12 * fabs(h0)) h1 = 1.0e-12 * h0;
    const double I_h2 = 1.0 / ((h0 + h1) * (h0 + h1));
    const double alpha = (h1 * h1) * I_h2, beta = (h0 * h0) * I_h2, abmix = (h0 * h1) * I_h2;
    const double a = 2.0 * alpha * (alpha + 2.0 * beta + 3.0 * abmix);
    const double b = 2.0 * beta * (beta + 2.0 * alpha + 3.0 * abmix);
    tri_c[i + 1] = 2.0 * abmix;
    tri_l[i + 1] = alpha; tri_u[i + 1] = beta;
    tri_b[i + 1] = a * u[i] + b * u[i + 1];
  }
  double dz[4], ut[4], C[4];
  for (int i = 1; i <= 4; i++) { dz[i - 1] = orc_max(h_neglect, h[i]); ut[i - 1] = u[i]; }
  end_value_h4(dz, ut, C);
  tri_b[1] = C[0]; tri_c[1] = 1.0; tri_u[1] = 0.0;
  for (int i = 1; i <= 4; i++) { dz[i - 1] = orc_max(h_neglect, h[N + 1 - i]); ut[i - 1] = u[N + 1 - i]; }
  end_value_h4(dz, ut, C);
  tri_b[N + 1] = C[0]; tri_c[N + 1] = 1.0; tri_l[N + 1] = 0.0;
  solve_diag_dominant_tridiag(tri_l, tri_c, tri_u, tri_b, tri_x, N + 1);
  E1[1] = tri_x[1];
  for (int i = 2; i <= N; i++) { E1[i] = tri_x[i]; E2[i - 1] = tri_x[i]; }
  E2[N] = tri_x[N + 1];
  free(w);
}

/* ---- PPM_functions.F90 ------------------------------------------------------------------------------------- */
/* PPM_limiter_standard :62-121 */
static void PPM_limiter_standard(int N, const double *h, const double *u, double *E1, double *E2, double h_neglect) {
  bound_edge_values(N, h, u, E1, E2, h_neglect);
  check_discontinuous_edge_values(N, u, E1, E2);
  for (int k = 2; k <= N - 1; k++) {
    const double u_l = u[k - 1], u_c = u[k], u_r = u[k + 1];
    double edge_l = E1[k], edge_r = E2[k];
    if ((u_r - u_c) * (u_c - u_l) <= 0.0) {
      edge_l = u_c; edge_r = u_c;
    } else {
      const double expr1 = 3.0 * (edge_r - edge_l) * ((u_c - edge_l) + (u_c - edge_r));
      const double expr2 = (edge_r - edge_l) * (edge_r - edge_l);
      if (expr1 > expr2) {
        edge_l = u_c + 2.0 * (u_c - edge_r);
        edge_l = orc_max(orc_min(edge_l, orc_max(u_l, u_c)), orc_min(u_l, u_c));
      } else if (expr1 < -expr2) {
        edge_r = u_c + 2.0 * (u_c - edge_l);
        edge_r = orc_max(orc_min(edge_r, orc_max(u_r, u_c)), orc_min(u_r, u_c));
      }
    }
    if (fabs(edge_r - edge_l) < orc_max(1.e-60, DBL_EPSILON * fabs(u_c))) { edge_l = u_c; edge_r = u_c; }
    E1[k] = edge_l; E2[k] = edge_r;
  }
  E1[1] = u[1]; E2[1] = u[1];
  E1[N] = u[N]; E2[N] = u[N];
}
/* PPM_reconstruction :25-55 */
void orc_PPM_reconstruction(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double *C3, double h_neglect) {
  PPM_limiter_standard(N, h, u, E1, E2, h_neglect);
  for (int k = 1; k <= N; k++) {
    const double edge_l = E1[k], edge_r = E2[k];
    C1[k] = edge_l;
    C2[k] = 4.0 * (u[k] - edge_l) + 2.0 * (u[k] - edge_r);
    C3[k] = 3.0 * ((edge_r - u[k]) + (edge_l - u[k]));
  }
}
/* PPM_boundary_extrapolation :166-296 */
void orc_PPM_boundary_extrapolation(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double *C3, double h_neglect) {
  int i0 = 1, i1 = 2;
  double h0 = h[i0], h1 = h[i1], u0 = u[i0], u1 = u[i1];
  double b = C2[i1];
  double u1_r = b * ((h0 + h_neglect) / (h1 + h_neglect));
  double slope = 2.0 * (u1 - u0);
  if (fabs(u1_r) > fabs(slope)) u1_r = slope;
  double u0_r = E1[i1];
  double u0_l = 3.0 * u0 + 0.5 * u1_r - 2.0 * u0_r;
  double exp1 = (u0_r - u0_l) * (u0 - 0.5 * (u0_l + u0_r));
  double exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
  if (exp1 > exp2) u0_l = 3.0 * u0 - 2.0 * u0_r;
  if (exp1 < -exp2) u0_r = 3.0 * u0 - 2.0 * u0_l;
  E1[i0] = u0_l; E2[i0] = u0_r;
  C1[i0] = u0_l; C2[i0] = 6.0 * u0 - 4.0 * u0_l - 2.0 * u0_r; C3[i0] = 3.0 * (u0_r + u0_l - 2.0 * u0);

  i0 = N - 1; i1 = N;
  h0 = h[i0]; h1 = h[i1]; u0 = u[i0]; u1 = u[i1];
  b = C2[i0];
  const double c = C3[i0];
  double u1_l = (b + 2 * c);
  u1_l = u1_l * ((h1 + h_neglect) / (h0 + h_neglect));
  slope = 2.0 * (u1 - u0);
  if (fabs(u1_l) > fabs(slope)) u1_l = slope;
  u0_l = E2[i0];
  u0_r = 3.0 * u1 - 0.5 * u1_l - 2.0 * u0_l;
  exp1 = (u0_r - u0_l) * (u1 - 0.5 * (u0_l + u0_r));
  exp2 = (u0_r - u0_l) * (u0_r - u0_l) / 6.0;
  if (exp1 > exp2) u0_l = 3.0 * u1 - 2.0 * u0_r;
  if (exp1 < -exp2) u0_r = 3.0 * u1 - 2.0 * u0_l;
  E1[i1] = u0_l; E2[i1] = u0_r;
  C1[i1] = u0_l; C2[i1] = 6.0 * u1 - 4.0 * u0_l - 2.0 * u0_r; C3[i1] = 3.0 * (u0_r + u0_l - 2.0 * u1);
}

/* ---- MOM_remapping.F90 ------------------------------------------------------------------------------------- */
/* build_reconstructions_1d :410-550 for PCM, PLM, PPM_H4, PPM_IH4 */
static int build_reconstructions_1d(const mom6x_remapping_params *CS, int n0, const double *h0, const double *u0, double *E1,
                                    double *E2, double *C1, double *C2, double *C3, int *iMethod) {
  for (int k = 0; k <= n0 + 1; k++) { E1[k] = 0.; E2[k] = 0.; C1[k] = 0.; C2[k] = 0.; C3[k] = 0.; }
  int scheme = CS->scheme;
  if (n0 <= 1) scheme = MOM6X_REMAP_PCM;
  else if (n0 <= 3) scheme = (scheme < MOM6X_REMAP_PLM) ? scheme : MOM6X_REMAP_PLM;
  else if (n0 <= 4) scheme = (scheme < MOM6X_REMAP_PPM_H4) ? scheme : MOM6X_REMAP_PPM_H4;
  switch (scheme) {
    case MOM6X_REMAP_PCM:
      for (int k = 1; k <= n0; k++) { C1[k] = u0[k]; E1[k] = u0[k]; E2[k] = u0[k]; }   /* PCM_functions.F90:16-35 */
      *iMethod = INTEGRATION_PCM;
      break;
    case MOM6X_REMAP_PLM:
      orc_PLM_reconstruction(n0, h0, u0, E1, E2, C1, C2, CS->h_neglect);
      if (CS->boundary_extrapolation) orc_PLM_boundary_extrapolation(n0, h0, u0, E1, E2, C1, C2, CS->h_neglect);
      *iMethod = INTEGRATION_PLM;
      break;
    case MOM6X_REMAP_PPM_H4:
    case MOM6X_REMAP_PPM_IH4:
      if (scheme == MOM6X_REMAP_PPM_IH4) orc_edge_values_implicit_h4(n0, h0, u0, E1, E2, CS->h_neglect_edge);
      else orc_edge_values_explicit_h4(n0, h0, u0, E1, E2, CS->h_neglect_edge);
      orc_PPM_reconstruction(n0, h0, u0, E1, E2, C1, C2, C3, CS->h_neglect);
      if (CS->boundary_extrapolation) orc_PPM_boundary_extrapolation(n0, h0, u0, E1, E2, C1, C2, C3, CS->h_neglect);
      *iMethod = INTEGRATION_PPM;
      break;
    default:
      return MOM6X_EUNSUPPORTED;
  }
  return MOM6X_OK;
}

/* intersect_src_tgt_grids :642-798 */
void orc_intersect_src_tgt_grids(int n0, const double *h0, int n1, const double *h1, double *h_sub, double *h0_eff, int *isrc_start,
                                 int *isrc_end, int *isrc_max, int *itgt_start, int *itgt_end, int *isub_src) {
  double h0_supply = h0[1], h1_supply = h1[1];
  int src_has_volume = 1, tgt_has_volume = 1;
  int i0 = 1, i1 = 1, i_start0 = 1, i_start1 = 1, i_max = 1;
  double dh_max = 0., dh0_eff = 0.;
  h_sub[1] = 0.;
  isrc_start[1] = 1; isrc_end[1] = 1; isrc_max[1] = 1; isub_src[1] = 1;
  for (int i_sub = 2; i_sub <= n0 + n1 + 1; i_sub++) {
    const double dh = orc_min(h0_supply, h1_supply);
    dh0_eff = dh0_eff + orc_min(dh, h0_supply);
    isub_src[i_sub] = i0;
    h_sub[i_sub] = dh;
    if (dh >= dh_max) { i_max = i_sub; dh_max = dh; }
    if (h0_supply <= h1_supply && src_has_volume) {
      h1_supply = h1_supply - dh;
      isrc_start[i0] = i_start0; isrc_end[i0] = i_sub; i_start0 = i_sub + 1;
      isrc_max[i0] = i_max; i_max = i_sub + 1; dh_max = 0.;
      h0_eff[i0] = dh0_eff;
      if (i0 < n0) { i0 = i0 + 1; h0_supply = h0[i0]; dh0_eff = 0.; }
      else { h0_supply = 0.; src_has_volume = 0; }
    } else if (h0_supply >= h1_supply && tgt_has_volume) {
      h0_supply = h0_supply - dh;
      itgt_start[i1] = i_start1; itgt_end[i1] = i_sub; i_start1 = i_sub + 1;
      if (i1 < n1) { i1 = i1 + 1; h1_supply = h1[i1]; }
      else { h1_supply = 0.; tgt_has_volume = 0; }
    } else if (src_has_volume) {
      h_sub[i_sub] = h0_supply;
      isrc_start[i0] = i_start0; isrc_end[i0] = i_sub; i_start0 = i_sub + 1;
      isrc_max[i0] = i_max; i_max = i_sub + 1; dh_max = 0.;
      h0_eff[i0] = dh0_eff;
      if (i0 < n0) { i0 = i0 + 1; h0_supply = h0[i0]; dh0_eff = 0.; }
      else { h0_supply = 0.; src_has_volume = 0; }
    } else if (tgt_has_volume) {
      h_sub[i_sub] = h1_supply;
      itgt_start[i1] = i_start1; itgt_end[i1] = i_sub; i_start1 = i_sub + 1;
      if (i1 < n1) { i1 = i1 + 1; h1_supply = h1[i1]; }
      else { h1_supply = 0.; tgt_has_volume = 0; }
    }
  }
}

/* average_value_ppoly :1391-1494 for PCM / PLM / PPM */
static double average_value_ppoly(const double *u0, const double *E1, const double *E2, const double *C1, const double *C2, int method,
                                  int i0, double xa, double xb) {
  double u_ave = 0.;
  if (xb > xa) {
    if (method == INTEGRATION_PCM) u_ave = u0[i0];
    else if (method == INTEGRATION_PLM) u_ave = (C1[i0] + C2[i0] * 0.5 * (xb + xa));
    else {
      const double mx = 0.5 * (xa + xb);
      const double a_L = E1[i0], a_R = E2[i0], u_c = u0[i0];
      const double a_c = 0.5 * ((u_c - a_L) + (u_c - a_R));
      if (mx < 0.5) {
        const double xa2b2ab = (xa * xa + xb * xb) + xa * xb;
        u_ave = a_L + ((a_R - a_L) * mx + a_c * (3. * (xb + xa) - 2. * xa2b2ab));
      } else {
        const double Ya = 1. - xa, Yb = 1. - xb, my = 0.5 * (Ya + Yb);
        const double Ya2b2ab = (Ya * Ya + Yb * Yb) + Ya * Yb;
        u_ave = a_R + ((a_L - a_R) * my + a_c * (3. * (Yb + Ya) - 2. * Ya2b2ab));
      }
    }
  } else {
    if (method == INTEGRATION_PCM) u_ave = C1[i0];
    else if (method == INTEGRATION_PLM) {
      const double a_L = E1[i0], a_R = E2[i0], Ya = 1. - xa;
      if (xa < 0.5) u_ave = a_L + xa * (a_R - a_L);
      else u_ave = a_R + Ya * (a_L - a_R);
    } else {
      const double a_L = E1[i0], a_R = E2[i0], u_c = u0[i0];
      const double a_c = 3. * ((u_c - a_L) + (u_c - a_R));
      const double Ya = 1. - xa;
      if (xa < 0.5) u_ave = a_L + xa * ((a_R - a_L) + a_c * Ya);
      else u_ave = a_R + Ya * ((a_L - a_R) + a_c * xa);
    }
  }
  return u_ave;
}

/* remap_src_to_sub_grid_om4 :845-959 (om4 != 0) / remap_src_to_sub_grid :962-1099 */
void orc_remap_src_to_sub_grid(int om4, int n0, const double *h0, const double *u0, const double *E1, const double *E2, const double *C1,
                               const double *C2, int n1, const double *h_sub, const double *h0_eff, const int *isrc_start,
                               const int *isrc_end, const int *isrc_max, const int *isub_src, int method, int force_bounds_in_subcell,
                               double *u_sub, double *uh_sub, double *u02_err) {
  double *u0_min = (double *)calloc(n0 + 2, sizeof(double)), *u0_max = (double *)calloc(n0 + 2, sizeof(double));
  int i0_last_thick_cell = 0;
  for (int i0 = 1; i0 <= n0; i0++) {
    u0_min[i0] = orc_min(E1[i0], E2[i0]); u0_max[i0] = orc_max(E1[i0], E2[i0]);
    if (h0[i0] > 0.) i0_last_thick_cell = i0;
  }
  double xa = 0., dh0_eff = 0.;
  *u02_err = 0.;
  int first = 1;
  if (om4) { uh_sub[1] = 0.; u_sub[1] = E1[1]; first = 2; }
  const int last = om4 ? n0 + n1 : n0 + n1 + 1;
  for (int i_sub = first; i_sub <= last; i_sub++) {
    const double dh = h_sub[i_sub];
    const int i0 = isub_src[i_sub];
    double xb;
    dh0_eff = dh0_eff + dh;
    const double hden = om4 ? h0_eff[i0] : h0[i0];
    if (hden > 0.) {
      xb = dh0_eff / hden;
      xb = orc_min(1., xb);
      u_sub[i_sub] = average_value_ppoly(u0, E1, E2, C1, C2, method, i0, xa, xb);
    } else {
      xb = 1.;
      u_sub[i_sub] = u0[i0];
    }
    if (force_bounds_in_subcell) {
      const double u_orig = u_sub[i_sub];
      u_sub[i_sub] = orc_max(u_sub[i_sub], u0_min[i0]);
      u_sub[i_sub] = orc_min(u_sub[i_sub], u0_max[i0]);
      *u02_err = *u02_err + dh * fabs(u_sub[i_sub] - u_orig);
    }
    uh_sub[i_sub] = dh * u_sub[i_sub];
    if (i_sub <= n0 + n1) {   /* (the non-OM4 code repeats the body once more for the last sub-cell, without this tail) */
      if (isub_src[i_sub + 1] != i0) { dh0_eff = 0.; xa = 0.; }
      else xa = xb;
    }
  }
  if (om4) {
    u_sub[n0 + n1 + 1] = E2[n0];
    uh_sub[n0 + n1 + 1] = E2[n0] * h_sub[n0 + n1 + 1];
  }
  for (int i0 = 1; i0 <= i0_last_thick_cell; i0++) {   /* adjust_thickest_subcell */
    const int i_max = isrc_max[i0];
    const double dh_max = h_sub[i_max];
    if (dh_max > 0.) {
      double duh = 0.;
      for (int i_sub = isrc_start[i0]; i_sub <= isrc_end[i0]; i_sub++)
        if (i_sub != i_max) duh = duh + uh_sub[i_sub];
      uh_sub[i_max] = u0[i0] * h0[i0] - duh;
      *u02_err = *u02_err + max3(fabs(uh_sub[i_max]), fabs(u0[i0] * h0[i0]), fabs(duh));
    }
  }
  free(u0_min); free(u0_max);
}

/* remap_sub_to_tgt_grid_om4 :1103-1163 */
void orc_remap_sub_to_tgt_grid_om4(int n0, int n1, const double *h1, const double *h_sub, const double *u_sub, const double *uh_sub,
                                   const int *itgt_start, const int *itgt_end, int force_bounds_in_target, double *u1, double *uh_err) {
  (void)n0;
  double u1min = 0., u1max = 0.;
  *uh_err = 0.;
  for (int i1 = 1; i1 <= n1; i1++) {
    if (h1[i1] > 0.) {
      double duh = 0., dh = 0.;
      int i_sub = itgt_start[i1];
      if (force_bounds_in_target) { u1min = u_sub[i_sub]; u1max = u_sub[i_sub]; }
      for (i_sub = itgt_start[i1]; i_sub <= itgt_end[i1]; i_sub++) {
        if (force_bounds_in_target) { u1min = orc_min(u1min, u_sub[i_sub]); u1max = orc_max(u1max, u_sub[i_sub]); }
        dh = dh + h_sub[i_sub];
        duh = duh + uh_sub[i_sub];
        *uh_err = *uh_err + orc_max(fabs(duh), fabs(uh_sub[i_sub])) * DBL_EPSILON;
      }
      u1[i1] = duh / dh;
      *uh_err = *uh_err + fabs(duh) * DBL_EPSILON;
      if (force_bounds_in_target) {
        const double u_orig = u1[i1];
        u1[i1] = orc_max(u1min, orc_min(u1max, u1[i1]));
        *uh_err = *uh_err + dh * fabs(u1[i1] - u_orig);
      }
    } else {
      u1[i1] = u_sub[itgt_start[i1]];
    }
  }
}

/* remapping_core_h :234-335 (OM4-era reconstruction branch).  h0, u0, h1, u1 are 0-based C arrays. */
int orc_remapping_core_h(const mom6x_remapping_params *CS, int n0, const double *h0c, const double *u0c, int n1, const double *h1c,
                         double *u1c, double *net_err) {
  if (CS->answer_date < 20190101) return MOM6X_EUNSUPPORTED;
  const int ns = n0 + n1 + 1;
  double *w = (double *)calloc((size_t)(7 * (n0 + 2) + 2 * (n1 + 2) + 3 * (ns + 2)), sizeof(double));
  int *iw = (int *)calloc((size_t)(3 * (n0 + 2) + 2 * (n1 + 2) + (ns + 2)), sizeof(int));
  double *h0 = w, *u0 = h0 + n0 + 2, *E1 = u0 + n0 + 2, *E2 = E1 + n0 + 2, *C1 = E2 + n0 + 2, *C2 = C1 + n0 + 2, *C3 = C2 + n0 + 2;
  /* h0_eff shares C3's successor */
  double *h1 = C3 + n0 + 2, *u1 = h1 + n1 + 2, *h_sub = u1 + n1 + 2, *uh_sub = h_sub + ns + 2, *u_sub = uh_sub + ns + 2;
  double *h0_eff = (double *)calloc((size_t)(n0 + 2), sizeof(double));
  int *isrc_start = iw, *isrc_end = isrc_start + n0 + 2, *isrc_max = isrc_end + n0 + 2, *itgt_start = isrc_max + n0 + 2,
      *itgt_end = itgt_start + n1 + 2, *isub_src = itgt_end + n1 + 2;
  for (int k = 1; k <= n0; k++) { h0[k] = h0c[k - 1]; u0[k] = u0c[k - 1]; }
  for (int k = 1; k <= n1; k++) h1[k] = h1c[k - 1];
  orc_intersect_src_tgt_grids(n0, h0, n1, h1, h_sub, h0_eff, isrc_start, isrc_end, isrc_max, itgt_start, itgt_end, isub_src);
  int iMethod = -999;
  int rc = build_reconstructions_1d(CS, n0, h0, u0, E1, E2, C1, C2, C3, &iMethod);
  double u02_err = 0., uh_err = 0.;
  if (rc == MOM6X_OK) {
    orc_remap_src_to_sub_grid(CS->om4_remap_via_sub_cells, n0, h0, u0, E1, E2, C1, C2, n1, h_sub, h0_eff, isrc_start, isrc_end, isrc_max,
                              isub_src, iMethod, CS->force_bounds_in_subcell, u_sub, uh_sub, &u02_err);
    orc_remap_sub_to_tgt_grid_om4(n0, n1, h1, h_sub, u_sub, uh_sub, itgt_start, itgt_end, CS->force_bounds_in_target, u1, &uh_err);
    uh_err = uh_err + u02_err;
    for (int k = 1; k <= n1; k++) u1c[k - 1] = u1[k];
    if (net_err) *net_err = uh_err;
  }
  free(w); free(iw); free(h0_eff);
  return rc;
}

/* `ncol` columns stored back to back */
int orc_remapping_core_h_cols(const mom6x_remapping_params *CS, int ncol, int n0, const double *h0, const double *u0, int n1,
                              const double *h1, double *u1) {
  for (int c = 0; c < ncol; c++) {
    int rc = orc_remapping_core_h(CS, n0, h0 + (size_t)c * n0, u0 + (size_t)c * n0, n1, h1 + (size_t)c * n1, u1 + (size_t)c * n1, NULL);
    if (rc) return rc;
  }
  return MOM6X_OK;
}

/* ---- MOM_ALE.F90 ------------------------------------------------------------------------------------------- */
static int remap_points(const mom6x_dims *d, const double *mask, int i_lo, int i_hi, int j_lo, int j_hi, const mom6x_remapping_params *CS,
                        const double *h_old, const double *h_new, double *f) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  int rc_all = MOM6X_OK;
#pragma omp parallel for schedule(static)
  for (int j = j_lo; j <= j_hi; j++) {
    double *h1 = (double *)malloc(sizeof(double) * 4 * (size_t)nz), *h2 = h1 + nz, *src = h2 + nz, *col = src + nz;
    for (int i = i_lo; i <= i_hi; i++) {
      const size_t x = IX2(d, i, j);
      if (!(mask[x] > 0.)) continue;
      for (int k = 0; k < nz; k++) { h1[k] = h_old[x + k * slab]; h2[k] = h_new[x + k * slab]; src[k] = f[x + k * slab]; }
      const int rc = orc_remapping_core_h(CS, nz, h1, src, nz, h2, col, NULL);
      if (rc) { rc_all = rc; continue; }
      for (int k = 0; k < nz; k++) f[x + k * slab] = col[k];
    }
    free(h1);
  }
  return rc_all;
}
/* ALE_remap_tracers :760-879 (no diagnostics, no conc_underflow) */
int orc_ALE_remap_tracers(const mom6x_dims *d, const double *G, const mom6x_remapping_params *CS, const double *h_old, const double *h_new,
                          double *const *fields, int nfields) {
  for (int m = 0; m < nfields; m++) {
    int rc = remap_points(d, GM(G, d, MOM6X_G_mask2dT), 0, d->ni - 1, 0, d->nj - 1, CS, h_old, h_new, fields[m]);
    if (rc) return rc;
  }
  return MOM6X_OK;
}
/* ALE_remap_set_h_vel :882-925 */
int orc_ALE_remap_set_h_vel(const mom6x_dims *d, const double *G, const double *h_new, double *h_u, double *h_v) {
  const double *mCu = GM(G, d, MOM6X_G_mask2dCu), *mCv = GM(G, d, MOM6X_G_mask2dCv);
  const size_t slab = (size_t)d->slab;
  const int st = d->pitch;
  for (int k = 0; k < d->nk; k++) {
    for (int j = 0; j < d->nj; j++) for (int I = -1; I < d->ni; I++) {
      const size_t x = IX2(d, I, j);
      if (mCu[x] > 0.) h_u[x + k * slab] = 0.5 * (h_new[x + k * slab] + h_new[x + 1 + k * slab]);
    }
    for (int J = -1; J < d->nj; J++) for (int i = 0; i < d->ni; i++) {
      const size_t x = IX2(d, i, J);
      if (mCv[x] > 0.) h_v[x + k * slab] = 0.5 * (h_new[x + k * slab] + h_new[x + st + k * slab]);
    }
  }
  return MOM6X_OK;
}
/* ALE_remap_velocities :1089-1300 (conserve_ke off, no near-bottom masking, no diagnostics) */
int orc_ALE_remap_velocities(const mom6x_dims *d, const double *G, const mom6x_remapping_params *CS, const double *h_old_u,
                             const double *h_old_v, const double *h_new_u, const double *h_new_v, double *u, double *v) {
  int rc = remap_points(d, GM(G, d, MOM6X_G_mask2dCu), -1, d->ni - 1, 0, d->nj - 1, CS, h_old_u, h_new_u, u);
  if (rc) return rc;
  return remap_points(d, GM(G, d, MOM6X_G_mask2dCv), 0, d->ni - 1, -1, d->nj - 1, CS, h_old_v, h_new_v, v);
}

/* The KE-conserving correction of ALE_remap_velocities (MOM_ALE.F90:1166-1195; REMAP_VEL_CONSERVE_KE with
 * allow_preserve_variance): the baroclinic part of the remapped column is rescaled so that its vertically integrated square equals
 * the source column's, by at most 25 %.  One velocity component: faces (i0..i1, j0..j1) with mask > 0. */
static void conserve_ke_points(const mom6x_dims *d, const double *mask, int i0, int i1, int j0, int j1, const double *h1, const double *h2,
                               const double *u_src, double *u_tgt, double H_subroundoff) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  for (int j = j0; j <= j1; j++) for (int i = i0; i <= i1; i++) {
    const size_t x = IX2(d, i, j);
    if (!(mask[x] > 0.0)) continue;
    double u_bt = 0.0, hsum = 0.0;
    for (int k = 0; k < nz; k++) u_bt = u_bt + h2[x + k * slab] * u_tgt[x + k * slab];
    for (int k = 0; k < nz; k++) hsum = hsum + h2[x + k * slab];          /* sum(h2(1:nz)) */
    u_bt = u_bt / (hsum + H_subroundoff);
    double ke_c_src = 0.0, ke_c_tgt = 0.0;
    for (int k = 0; k < nz; k++) {
      const double a = u_src[x + k * slab] - u_bt, b = u_tgt[x + k * slab] - u_bt;
      ke_c_src = ke_c_src + h1[x + k * slab] * (a * a);
      ke_c_tgt = ke_c_tgt + h2[x + k * slab] * (b * b);
    }
    double rescale_coef;
    if (ke_c_src < 1.5625 * ke_c_tgt) rescale_coef = sqrt(ke_c_src / ke_c_tgt);
    else rescale_coef = 1.25;
    for (int k = 0; k < nz; k++) u_tgt[x + k * slab] = u_bt + rescale_coef * (u_tgt[x + k * slab] - u_bt);
  }
}

/* ALE_remap_velocities with allow_preserve_variance and CS%conserve_ke (:1166-1195, :1240-1270) */
int orc_ALE_remap_velocities_conserve_ke(const mom6x_dims *d, const double *G, const mom6x_remapping_params *CS, double H_subroundoff,
                                         const double *h_old_u, const double *h_old_v, const double *h_new_u, const double *h_new_v,
                                         double *u, double *v) {
  const size_t n3 = (size_t)d->slab * d->nk;
  double *us = (double *)malloc(n3 * sizeof(double)), *vs = (double *)malloc(n3 * sizeof(double));
  memcpy(us, u, n3 * sizeof(double)); memcpy(vs, v, n3 * sizeof(double));
  int rc = orc_ALE_remap_velocities(d, G, CS, h_old_u, h_old_v, h_new_u, h_new_v, u, v);
  if (!rc) {
    conserve_ke_points(d, GM(G, d, MOM6X_G_mask2dCu), -1, d->ni - 1, 0, d->nj - 1, h_old_u, h_new_u, us, u, H_subroundoff);
    conserve_ke_points(d, GM(G, d, MOM6X_G_mask2dCv), 0, d->ni - 1, -1, d->nj - 1, h_old_v, h_new_v, vs, v, H_subroundoff);
  }
  free(us); free(vs);
  return rc;
}

/* ---- regridding: ALE_regrid :518 -> regridding_main MOM_regridding.F90:862 for REGRIDDING_ZSTAR ---------------------- */
/* build_zstar_column coord_zlike.F90:63-144 (no rigid top); zInterface is 1-based with nk+1 entries */
static void build_zstar_column(int nk, const double *coordinateResolution, double cs_min_thickness, double depth, double total_thickness,
                               double *zInterface, double z_scale) {
  const double min_thickness = orc_min(cs_min_thickness, total_thickness / (double)nk);
  const double eta = total_thickness - depth;
  const double stretching = total_thickness / (depth + 0.);
  zInterface[1] = eta;
  for (int k = 1; k <= nk; k++) {
    const double dh = stretching * coordinateResolution[k - 1] * z_scale;
    zInterface[k + 1] = zInterface[k] - dh;
  }
  zInterface[nk + 1] = -depth;
  for (int k = nk; k >= 1; k--)
    if (zInterface[k] < (zInterface[k + 1] + min_thickness)) zInterface[k] = zInterface[k + 1] + min_thickness;
}
/* filtered_grid_motion :1105-1252 (CS%nk == nk) */
static int filtered_grid_motion(const mom6x_regrid_zstar_params *CS, int nk, const double *z_old, const double *z_new, double *dz_g) {
  double sgn;
  const double test = (z_old[nk + 1] - z_old[1]) * (z_new[nk + 1] - z_new[1]);
  if (test < 0.0) return MOM6X_EINVAL;
  else if (test == 0.0) { for (int k = 1; k <= nk + 1; k++) dz_g[k] = 0.0; return MOM6X_OK; }
  else if ((z_old[nk + 1] - z_old[1]) + (z_new[nk + 1] - z_new[1]) > 0.0) sgn = 1.0;
  else sgn = -1.0;
  const double zs = CS->depth_of_time_filter_shallow, zd = CS->depth_of_time_filter_deep;
  const double wtd = 1.0 - CS->old_grid_weight, Iwtd = 1.0 / wtd;
  const double dzwt = (zd - zs);
  double Idzwt = 0.0; if (fabs(zd - zs) > 0.0) Idzwt = 1.0 / (zd - zs);
  const double dInt_zs_zd = 0.5 * (1.0 + Iwtd) * (zd - zs);
  const double Aq = 0.5 * (Iwtd - 1.0);
  dz_g[1] = 0.0;
  for (int k = 2; k <= nk + 1; k++) {
    const double z_old_k = z_old[k];
    const double dz_tgt = sgn * (z_new[k] - z_old_k);
    const double zr1 = sgn * (z_old_k - z_old[1]);
    if ((zr1 > zd) && (zr1 + wtd * dz_tgt > zd)) dz_g[k] = sgn * wtd * dz_tgt;
    else if ((zr1 < zs) && (zr1 + dz_tgt < zs)) dz_g[k] = sgn * dz_tgt;
    else {
      double Int_zd, Int_zs;
      if (zr1 >= zd) { Int_zd = Iwtd * (zd - zr1); Int_zs = Int_zd - dInt_zs_zd; }
      else if (zr1 <= zs) { Int_zs = (zs - zr1); Int_zd = dInt_zs_zd + (zs - zr1); }
      else {
        Int_zd = (zd - zr1) * (Iwtd * (0.5 * (zd + zr1) - zs) + 0.5 * (zd - zr1)) * Idzwt;
        Int_zs = (zs - zr1) * (0.5 * Iwtd * ((zr1 - zs)) + (zd - 0.5 * (zr1 + zs))) * Idzwt;
      }
      if (dz_tgt >= Int_zd) dz_g[k] = sgn * ((zd - zr1) + wtd * (dz_tgt - Int_zd));
      else if (dz_tgt <= Int_zs) dz_g[k] = sgn * ((zs - zr1) + (dz_tgt - Int_zs));
      else {
        double dz0, z0, F0;
        if (zr1 <= zs) { dz0 = zs - zr1; z0 = zs; F0 = dz_tgt - Int_zs; }
        else if (zr1 >= zd) { dz0 = zd - zr1; z0 = zd; F0 = dz_tgt - Int_zd; }
        else { dz0 = 0.0; z0 = zr1; F0 = dz_tgt; }
        const double Bq = (dzwt + 2.0 * Aq * (z0 - zs));
        dz_g[k] = sgn * (dz0 + 2.0 * F0 * dzwt / (Bq + sqrt(Bq * Bq + 4.0 * Aq * F0 * dzwt)));
      }
    }
  }
  return MOM6X_OK;
}
static int adjust_interface_motion(double cs_min_thickness, int nk, const double *h_old, double *dz_int);
/* regridding_main :862-987 (ZSTAR branch: build_zstar_grid :1257-1366) + calc_h_new_by_dz :1008-1037; dzRegrid has nk+1 levels */
int orc_ALE_regrid_zstar(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_regrid_zstar_params *CS,
                         const double *coordinateResolution, const double *h, double *h_new, double *dzRegrid) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT), *bathyT = GM(G, d, MOM6X_G_bathyT);
  double *zOld = (double *)calloc((size_t)(2 * (nz + 3)), sizeof(double)), *zNew = zOld + nz + 3;
  double *dz = (double *)calloc((size_t)(2 * (nz + 3)), sizeof(double)), *hc = dz + nz + 3;
  int rc = MOM6X_OK;
  for (size_t n = 0; n < slab * (size_t)(nz + 1); n++) dzRegrid[n] = 0.0;            /* ALE_regrid :539 */
  for (int j = -1; j <= d->nj; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
    if (mT[x] == 0.) {                                                                /* :1300-1303, :1030-1033 */
      for (int k = 0; k < nz; k++) h_new[x + k * slab] = h[x + k * slab];
      continue;
    }
    const double nominalDepth = orc_max((bathyT[x] + CS->Z_ref) * GV->Z_to_H, 0.0);   /* :920-922 */
    double totalThickness = 0.0;
    for (int k = 1; k <= nz; k++) totalThickness = totalThickness + h[x + (k - 1) * slab];
    zOld[nz + 1] = -nominalDepth;
    for (int k = nz; k >= 1; k--) zOld[k] = zOld[k + 1] + h[x + (k - 1) * slab];
    build_zstar_column(nz, coordinateResolution, CS->min_thickness, nominalDepth, totalThickness, zNew, GV->Z_to_H);
    rc = filtered_grid_motion(CS, nz, zOld, zNew, dz);
    if (rc) break;
    for (int k = 1; k <= nz; k++) hc[k] = h[x + (k - 1) * slab];
    rc = adjust_interface_motion(CS->min_thickness, nz, hc, dz);                       /* :1362 */
    if (rc) break;
    for (int k = 1; k <= nz + 1; k++) dzRegrid[x + (k - 1) * slab] = dz[k];
    for (int k = 1; k <= nz; k++) h_new[x + (k - 1) * slab] = orc_max(0., h[x + (k - 1) * slab] + (dz[k] - dz[k + 1]));
  }
  free(zOld); free(dz);
  return rc;
}

/* ==== the density-following coordinates: REGRIDDING_RHO and REGRIDDING_HYCOM1 ========================================== */
double orc_eos_density(const mom6x_eos_params *E, double T, double S, double p);   /* orc_dyn.c: calculate_density */

/* adjust_interface_motion MOM_regridding.F90:1796-1857 (CS%nk == nk); h_old is 1-based */
static int adjust_interface_motion(double cs_min_thickness, int nk, const double *h_old, double *dz_int) {
  const double eps = 2.220446049250313e-16;   /* epsilon(1.) */
  double h_total = 0., h_err = 0.;
  for (int k = 1; k <= nk; k++) {
    h_total = h_total + h_old[k];
    h_err = h_err + max3(h_old[k], fabs(dz_int[k]), fabs(dz_int[k + 1])) * eps;
    const double h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < -3.0 * h_err) return MOM6X_EINVAL;   /* "implied h<0 is larger than roundoff!" */
  }
  for (int k = nk; k >= 2; k--) {
    double h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < cs_min_thickness) dz_int[k] = (dz_int[k + 1] - h_old[k]) + cs_min_thickness;
    h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < 0.) dz_int[k] = (1. - eps) * (dz_int[k + 1] - h_old[k]);
    h_new = h_old[k] + (dz_int[k] - dz_int[k + 1]);
    if (h_new < 0.) return MOM6X_EINVAL;
  }
  return MOM6X_OK;
}

/* edge_values_explicit_h2 regrid_edge_values.F90:166-192 */
static void edge_values_explicit_h2(int N, const double *h, const double *u, double *E1, double *E2) {
  E1[1] = u[1]; E2[N] = u[N];
  for (int k = 2; k <= N; k++) {
    if (h[k - 1] + h[k] == 0.0) E1[k] = 0.5 * (u[k - 1] + u[k]);
    else E1[k] = (u[k - 1] * h[k] + u[k] * h[k - 1]) / (h[k - 1] + h[k]);
    E2[k - 1] = E1[k];
  }
}
/* average_discontinuous_edge_values :107-126 */
static void average_discontinuous_edge_values(int N, double *E1, double *E2) {
  for (int k = 1; k <= N - 1; k++) {
    if (E2[k] != E1[k + 1]) { const double a = 0.5 * (E2[k] + E1[k + 1]); E2[k] = a; E1[k + 1] = a; }
  }
}
/* P1M_interpolation P1M_functions.F90:25-56 and P1M_boundary_extrapolation :72-160 */
static void P1M_interpolation(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double h_neglect) {
  bound_edge_values(N, h, u, E1, E2, h_neglect);
  average_discontinuous_edge_values(N, E1, E2);
  for (int k = 1; k <= N; k++) { C1[k] = E1[k]; C2[k] = E2[k] - E1[k]; }
}
static void P1M_boundary_extrapolation(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2) {
  double u0 = u[1], u1 = u[2];
  double slope = 2.0 * (u1 - u0);
  const double u0_r = u0 + 0.5 * slope;
  if ((u1 - u0) * (E1[2] - u0_r) < 0.0) slope = 2.0 * (E1[2] - u0);
  if (h[1] != 0.0) E1[1] = u0 - 0.5 * slope; else E1[1] = u0;
  C1[1] = E1[1]; C2[1] = E2[1] - E1[1];
  u0 = u[N - 1]; u1 = u[N];
  slope = 2.0 * (u1 - u0);
  const double u0_l = u1 - 0.5 * slope;
  if ((u1 - u0) * (u0_l - E2[N - 1]) < 0.0) slope = 2.0 * (u1 - E2[N - 1]);
  if (h[N] != 0.0) E2[N] = u1 + 0.5 * slope; else E2[N] = u1;
  C1[N] = E1[N]; C2[N] = E2[N] - E1[N];
}
#define MOM6X_INTERP_P1M_H4 1   /* INTERPOLATION_P1M_H4: restated here only (the device path does not carry it) */
/* regridding_set_ppolys regrid_interp.F90:80-288 for P1M_H2, P1M_H4, PLM, PPM_H4; returns the degree (or < 0) */
static int regridding_set_ppolys(int scheme, int extrapolate, const double *dens, int n0, const double *h0, double *E1, double *E2,
                                 double *C1, double *C2, double *C3, double h_neglect, double h_neg_edge) {
  for (int k = 0; k <= n0 + 1; k++) { E1[k] = 0.; E2[k] = 0.; C1[k] = 0.; C2[k] = 0.; C3[k] = 0.; }
  switch (scheme) {
    case MOM6X_INTERP_PLM:
      orc_PLM_reconstruction(n0, h0, dens, E1, E2, C1, C2, h_neglect);
      if (extrapolate) orc_PLM_boundary_extrapolation(n0, h0, dens, E1, E2, C1, C2, h_neglect);
      return 1;
    case MOM6X_INTERP_PPM_H4:
      if (n0 >= 4) {
        orc_edge_values_explicit_h4(n0, h0, dens, E1, E2, h_neg_edge);
        orc_PPM_reconstruction(n0, h0, dens, E1, E2, C1, C2, C3, h_neglect);
        if (extrapolate) orc_PPM_boundary_extrapolation(n0, h0, dens, E1, E2, C1, C2, C3, h_neglect);
        return 2;
      }
      /* fall through: too few layers, the simplest continuous linear scheme */
    case MOM6X_INTERP_P1M_H2:
    case MOM6X_INTERP_P1M_H4:
      if (scheme == MOM6X_INTERP_P1M_H4 && n0 >= 4) orc_edge_values_explicit_h4(n0, h0, dens, E1, E2, h_neg_edge);
      else edge_values_explicit_h2(n0, h0, dens, E1, E2);
      P1M_interpolation(n0, h0, dens, E1, E2, C1, C2, h_neglect);
      if (extrapolate) P1M_boundary_extrapolation(n0, h0, dens, E1, E2, C1, C2);
      return 1;
    default:
      return -1;
  }
}
/* get_polynomial_coordinate :376-509 (answer dates >= 20190101); *err is set when no cell holds the target */
static double get_polynomial_coordinate(int N, const double *h, const double *x_g, const double *E1, const double *E2, const double *C1,
                                        const double *C2, const double *C3, double target_value, int degree, int *err) {
  const double eps = 1e-6;   /* NR_OFFSET */
  if (target_value <= E1[1]) return x_g[1];
  for (int k = 2; k <= N; k++)
    if ((target_value >= E2[k - 1]) && (target_value <= E1[k])) return x_g[k];
  if (target_value >= E2[N]) return x_g[N + 1];
  int k_found = -1;
  for (int k = 1; k <= N; k++)
    if ((target_value > E1[k]) && (target_value < E2[k])) { k_found = k; break; }
  if (k_found == -1) { *err = 1; return x_g[1]; }
  double a[6] = { 0., 0., 0., 0., 0., 0. };
  a[1] = C1[k_found]; a[2] = C2[k_found];
  if (degree >= 2) a[3] = C3[k_found];
  double xi0 = 0.5;
  for (int iter = 1; iter <= 8; iter++) {   /* NR_ITERATIONS */
    const double numerator = (a[1] - target_value) + xi0 * (a[2] + xi0 * (a[3] + xi0 * (a[4] + a[5] * xi0)));
    const double denominator = a[2] + xi0 * (2. * a[3] + xi0 * (3. * a[4] + 4. * a[5] * xi0));
    const double delta = -numerator / denominator;
    xi0 = xi0 + delta;
    if (xi0 < 0.0) { xi0 = 0.0; const double grad = a[2]; if (grad == 0.0) xi0 = xi0 + eps; }
    if (xi0 > 1.0) { xi0 = 1.0; const double grad = a[2] + (2. * a[3] + (3. * a[4] + 4. * a[5])); if (grad == 0.0) xi0 = xi0 - eps; }
    if (fabs(delta) < 1e-12) break;         /* NR_TOLERANCE */
  }
  return x_g[k_found] + xi0 * h[k_found];
}
/* build_and_interpolate_grid :331-358 = regridding_set_ppolys + interpolate_grid :295-328.  All arrays 1-based;
 * w: 5 * (n0 + 2) doubles of work space */
static int build_and_interpolate_grid(int scheme, int extrapolate, const double *dens, int n0, const double *h0, const double *x0,
                                      const double *target, int n1, double *h1, double *x1, double h_neglect, double h_neg_edge, double *w) {
  double *E1 = w, *E2 = w + (n0 + 2), *C1 = w + 2 * (n0 + 2), *C2 = w + 3 * (n0 + 2), *C3 = w + 4 * (n0 + 2);
  const int degree = regridding_set_ppolys(scheme, extrapolate, dens, n0, h0, E1, E2, C1, C2, C3, h_neglect, h_neg_edge);
  if (degree < 0) return MOM6X_EUNSUPPORTED;
  int err = 0;
  x1[1] = x0[1]; x1[n1 + 1] = x0[n0 + 1];
  for (int k = 2; k <= n1; k++) {
    x1[k] = get_polynomial_coordinate(n0, h0, x0, E1, E2, C1, C2, C3, target[k], degree, &err);
    h1[k - 1] = x1[k] - x1[k - 1];
  }
  h1[n1] = x1[n1 + 1] - x1[n1];
  return err ? MOM6X_EINVAL : MOM6X_OK;
}
/* copy_finite_thicknesses coord_rho.F90:316-358 and old_inflate_layers_1d :362-420 */
static void copy_finite_thicknesses(int nk, const double *h_in, double thresh, int *nout_, double *h_out, int *mapping) {
  int nout = 0, k_thickest = 1;
  double thickness_in_vanished = 0.0, thickest_h_out = h_in[1];
  for (int k = 1; k <= nk; k++) {
    mapping[k] = nout;
    h_out[k] = 0.;
    if (h_in[k] > thresh) {
      nout = nout + 1;
      mapping[nout] = k;
      h_out[nout] = h_in[k];
      if (h_out[nout] > thickest_h_out) { thickest_h_out = h_out[nout]; k_thickest = nout; }
    } else thickness_in_vanished = thickness_in_vanished + h_in[k];
  }
  *nout_ = nout;
  if (nout <= 1) return;
  h_out[k_thickest] = h_out[k_thickest] + thickness_in_vanished;
}
static void old_inflate_layers_1d(double min_thickness, int nk, double *h) {
  int count_nonzero_layers = 0;
  for (int k = 1; k <= nk; k++) if (h[k] > min_thickness) count_nonzero_layers = count_nonzero_layers + 1;
  if (count_nonzero_layers == nk) return;
  if (count_nonzero_layers == 0) { for (int k = 1; k <= nk; k++) h[k] = min_thickness; return; }
  double correction = 0.0;
  for (int k = 1; k <= nk; k++) {
    if (h[k] <= min_thickness) { const double delta = min_thickness - h[k]; correction = correction + delta; h[k] = h[k] + delta; }
  }
  double maxThickness = h[1];
  int k_found = 1;
  for (int k = 1; k <= nk; k++) if (h[k] > maxThickness) { maxThickness = h[k]; k_found = k; }
  h[k_found] = h[k_found] - correction;
}

/* convective_adjustment MOM_regridding.F90:1905-1967 over (isc-1..iec+1, jsc-1..jec+1) */
int orc_ALE_convective_adjustment(const mom6x_dims *d, const mom6x_eos_params *eos, double *h, double *T, double *S) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  double *dens = (double *)calloc((size_t)nz + 2, sizeof(double));
  for (int j = -1; j <= d->nj; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
    for (int k = 1; k <= nz; k++) dens[k] = orc_eos_density(eos, T[x + (k - 1) * slab], S[x + (k - 1) * slab], 0.);
    for (;;) {
      int stratified = 1;
      for (int k = 1; k <= nz - 1; k++) {
        const size_t c0 = x + (k - 1) * slab, c1 = x + k * slab;
        const double T0 = T[c0], T1 = T[c1], S0 = S[c0], S1 = S[c1], r0 = dens[k], r1 = dens[k + 1], h0 = h[c0], h1 = h[c1];
        if (r0 > r1) {
          T[c0] = T1; T[c1] = T0; S[c0] = S1; S[c1] = S0; h[c0] = h1; h[c1] = h0;
          dens[k] = orc_eos_density(eos, T[c0], S[c0], 0.);
          dens[k + 1] = orc_eos_density(eos, T[c1], S[c1], 0.);
          stratified = 0;
        }
      }
      if (stratified) break;
    }
  }
  free(dens);
  return MOM6X_OK;
}

static double set_h_neglect_2019(const mom6x_vgrid *GV) { return GV->H_subroundoff; }   /* set_h_neglect :2602, answer dates >= 20190101 */

/* regridding_main :862 for REGRIDDING_RHO: build_rho_grid :1472-1626 + build_rho_column coord_rho.F90:92-175 + calc_h_new_by_dz */
int orc_ALE_regrid_rho(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_regrid_rho_params *CS,
                       const mom6x_eos_params *eos, const double *target_density, const double *h, const double *T, const double *S,
                       double *h_new, double *dzRegrid) {
  const int nz = d->nk, nk = d->nk;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double h_neglect = set_h_neglect_2019(GV), h_neglect_edge = h_neglect;
  const size_t n = (size_t)nz + 3;
  double *buf = (double *)calloc(n * 16, sizeof(double));
  double *zOld = buf, *zNew = buf + n, *dz = buf + 2 * n, *hc = buf + 3 * n, *h_nv = buf + 4 * n, *dens = buf + 5 * n, *xTmp = buf + 6 * n;
  double *hn = buf + 7 * n, *x1 = buf + 8 * n, *w = buf + 9 * n, *tgt = buf + 14 * n, *dnv = buf + 15 * n;
  int *mapping = (int *)calloc(n, sizeof(int));
  for (int k = 1; k <= nk + 1; k++) tgt[k] = target_density[k - 1];
  int rc = MOM6X_OK;
  for (size_t q = 0; q < slab * (size_t)(nz + 1); q++) dzRegrid[q] = 0.0;
  for (int j = -1; j <= d->nj && !rc; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
    if (mT[x] == 0.) { for (int k = 0; k < nz; k++) h_new[x + k * slab] = h[x + k * slab]; continue; }   /* :1527-1530, :1030 */
    const double nominalDepth = orc_max((bathyT[x] + CS->f.Z_ref) * GV->Z_to_H, 0.0);
    for (int k = 1; k <= nz; k++) hc[k] = h[x + (k - 1) * slab];
    /* ---- build_rho_column */
    int count_nonzero_layers;
    copy_finite_thicknesses(nz, hc, CS->f.min_thickness, &count_nonzero_layers, h_nv, mapping);
    if (count_nonzero_layers > 1) {
      xTmp[1] = 0.0;
      for (int k = 1; k <= count_nonzero_layers; k++) xTmp[k + 1] = xTmp[k] + h_nv[k];
      for (int k = 1; k <= nz; k++) dens[k] = orc_eos_density(eos, T[x + (k - 1) * slab], S[x + (k - 1) * slab], CS->ref_pressure);
      for (int k = 1; k <= count_nonzero_layers; k++) dnv[k] = dens[mapping[k]];
      rc = build_and_interpolate_grid(CS->interp_scheme, CS->boundary_extrapolation, dnv, count_nonzero_layers, h_nv, xTmp, tgt, nk, hn, x1,
                                      h_neglect, h_neglect_edge, w);
      if (rc) break;
      old_inflate_layers_1d(CS->f.min_thickness, nk, hn);
      x1[1] = 0.0; for (int k = 1; k <= nk; k++) x1[k + 1] = x1[k] + hn[k];
      for (int k = 1; k <= nk; k++) hn[k] = x1[k + 1] - x1[k];
    } else {
      for (int k = 1; k <= nk; k++) hn[k] = hc[k];   /* nz == CS%nk: "This keeps old behavior" */
    }
    if (CS->integrate_downward_for_e) {
      zNew[1] = 0.; for (int k = 1; k <= nk; k++) zNew[k + 1] = zNew[k] - hn[k];
      zOld[1] = 0.; for (int k = 1; k <= nz; k++) zOld[k + 1] = zOld[k] - hc[k];
    } else {
      zNew[nk + 1] = -nominalDepth; for (int k = nk; k >= 1; k--) zNew[k] = zNew[k + 1] + hn[k];
      zOld[nz + 1] = -nominalDepth; for (int k = nz; k >= 1; k--) zOld[k] = zOld[k + 1] + hc[k];
    }
    rc = filtered_grid_motion(&CS->f, nz, zOld, zNew, dz);
    if (rc) break;
    for (int k = 1; k <= nz + 1; k++) dzRegrid[x + (k - 1) * slab] = dz[k];
    for (int k = 1; k <= nz; k++) h_new[x + (k - 1) * slab] = orc_max(0., hc[k] + (dz[k] - dz[k + 1]));
  }
  free(buf); free(mapping);
  return rc;
}

/* regridding_main :862 for REGRIDDING_HYCOM1: build_grid_HyCOM1 :1638-1726 + build_hycom1_column coord_hycom.F90:106-213 (no
 * "only improves") + calc_h_new_by_dz.  z is positive DOWNWARD here, as in the reference. */
int orc_ALE_regrid_hycom1(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_regrid_rho_params *CS,
                          const mom6x_eos_params *eos, const double *coordinateResolution, const double *target_density,
                          const double *max_interface_depths, const double *max_layer_thickness, const double *h, const double *T,
                          const double *S, double *h_new, double *dzRegrid) {
  const int nz = d->nk, nk = d->nk;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double h_neglect = set_h_neglect_2019(GV), h_neglect_edge = h_neglect;
  const double z_scale = GV->Z_to_H;
  const size_t n = (size_t)nz + 3;
  double *buf = (double *)calloc(n * 13, sizeof(double));
  double *z_col = buf, *z_new = buf + n, *dz = buf + 2 * n, *hc = buf + 3 * n, *rho = buf + 4 * n, *hn = buf + 5 * n, *w = buf + 6 * n;
  double *tgt = buf + 11 * n, *p_col = buf + 12 * n;
  for (int k = 1; k <= nk + 1; k++) tgt[k] = target_density[k - 1];
  int rc = MOM6X_OK;
  for (size_t q = 0; q < slab * (size_t)(nz + 1); q++) dzRegrid[q] = 0.0;
  for (int j = -1; j <= d->nj && !rc; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
    if (!(mT[x] > 0.)) { for (int k = 0; k < nz; k++) h_new[x + k * slab] = h[x + k * slab]; continue; }
    const double nominalDepth = orc_max((bathyT[x] + CS->f.Z_ref) * GV->Z_to_H, 0.0);
    for (int k = 1; k <= nz; k++) hc[k] = h[x + (k - 1) * slab];
    z_col[1] = 0.0;
    for (int k = 1; k <= nz; k++) {
      z_col[k + 1] = z_col[k] + hc[k];
      p_col[k] = CS->ref_pressure + CS->compressibility_fraction *
                 (0.5 * (z_col[k] + z_col[k + 1]) * (GV->H_to_RZ * GV->g_Earth) - CS->ref_pressure);
    }
    /* ---- build_hycom1_column */
    for (int k = 1; k <= nz; k++) rho[k] = orc_eos_density(eos, T[x + (k - 1) * slab], S[x + (k - 1) * slab], p_col[k]);
    for (int k = nz - 1; k >= 1; k--) rho[k] = orc_min(rho[k], rho[k + 1]);
    rc = build_and_interpolate_grid(CS->interp_scheme, CS->boundary_extrapolation, rho, nz, hc, z_col, tgt, nk, hn, z_new, h_neglect,
                                    h_neglect_edge, w);
    if (rc) break;
    double nominal_z = 0.;
    const double stretching = z_col[nz + 1] / nominalDepth;
    for (int k = 2; k <= nk + 1; k++) {
      nominal_z = nominal_z + (z_scale * coordinateResolution[k - 2]) * stretching;
      z_new[k] = orc_max(z_new[k], nominal_z);
      z_new[k] = orc_min(z_new[k], z_col[nz + 1]);
    }
    if (max_interface_depths && max_layer_thickness) {
      for (int k = 2; k <= nk; k++) z_new[k] = min3(z_new[k], max_interface_depths[k - 1], z_new[k - 1] + max_layer_thickness[k - 2]);
    } else if (max_interface_depths) {
      for (int k = 2; k <= nk; k++) z_new[k] = orc_min(z_new[k], max_interface_depths[k - 1]);
    } else if (max_layer_thickness) {
      for (int k = 2; k <= nk; k++) z_new[k] = orc_min(z_new[k], z_new[k - 1] + max_layer_thickness[k - 2]);
    }
    /* ---- back in build_grid_HyCOM1 */
    rc = filtered_grid_motion(&CS->f, nz, z_col, z_new, dz);
    if (rc) break;
    for (int k = 1; k <= nz + 1; k++) dz[k] = -dz[k];
    rc = adjust_interface_motion(CS->f.min_thickness, nz, hc, dz);
    if (rc) break;
    for (int k = 1; k <= nz + 1; k++) dzRegrid[x + (k - 1) * slab] = dz[k];
    for (int k = 1; k <= nz; k++) h_new[x + (k - 1) * slab] = orc_max(0., hc[k] + (dz[k] - dz[k + 1]));
  }
  free(buf);
  return rc;
}
