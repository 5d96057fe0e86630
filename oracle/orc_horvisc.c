/* orc_horvisc.c -- hor_visc_init and horizontal_viscosity (MOM_hor_visc.F90:2322-3290 / :266-2317).
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * Restated: LAPLACIAN and/or BIHARMONIC with background coefficients (KH, KH_VEL_SCALE, AH, AH_VEL_SCALE,
 * AH_TIME_SCALE), SMAGORINSKY_KH / _AH (+ BOUND_CORIOLIS_BIHARM), ADD_LES_VISCOSITY, BOUND_KH / BOUND_AH in the
 * "better" (:1226-1241, :1400-1410, :3025-3114) and the legacy (:1179-1183, :1382-1386) form,
 * USE_LAND_MASK_FOR_HVISC, NOSLIP.  Not restated (callers reject them): Leith, MEKE, GME, backscatter, anisotropy,
 * KH_SIN_LAT, 2-D background files, RE_AH, continuity thicknesses, OBCs, FrictWork.
 * The 2-D coefficient planes are kept in one block `P[ORC_HV_COUNT][slab]` (same layout as the device code). */
#include "orc_common.h"

enum {
  HV_dx2h = 0, HV_dy2h, HV_dx2q, HV_dy2q, HV_DX_dyT, HV_DY_dxT, HV_DX_dyBu, HV_DY_dxBu, HV_red_xx, HV_red_xy,
  HV_Kh_bg_xx, HV_Kh_bg_xy, HV_Kh_Max_xx, HV_Kh_Max_xy, HV_Lap2_xx, HV_Lap2_xy,
  HV_Idx2dyCu, HV_Idxdy2u, HV_Idx2dyCv, HV_Idxdy2v, HV_Ah_bg_xx, HV_Ah_bg_xy, HV_Ah_Max_xx, HV_Ah_Max_xy,
  HV_Bih_xx, HV_Bih_xy, HV_Bih2_xx, HV_Bih2_xy, HV_u0u, HV_u0v, HV_v0u, HV_v0v,
  HV_Lap3_xx, HV_Lap3_xy, HV_Bih6_xx, HV_Bih6_xy, HV_dF_dx, HV_dF_dy, ORC_HV_COUNT
};
int orc_hor_visc_nplanes(void) { return ORC_HV_COUNT; }
void orc_pass_var(const mom6x_dims *d, double *a, int stagger, int nk);   /* orc_rk2.c: the single-tile halo update */

#define PL(n) (P + (size_t)(n) * slab)
#define M(n) GM(G, d, MOM6X_G_##n)

static double max4(double a, double b, double c, double e) { return orc_max(orc_max(orc_max(a, b), c), e); }
static double min4(double a, double b, double c, double e) { return orc_min(orc_min(orc_min(a, b), c), e); }

/* hor_visc_init :2722-3114 (the parts that fill the 2-D planes) */
int orc_hor_visc_init(const mom6x_dims *d, const double *G, const mom6x_hor_visc_params *CS_in, double *P) {
  mom6x_hor_visc_params cs_ = *CS_in;   /* hor_visc_init :2465, :2495-2496, :2555-2571, :2624 */
  if (!cs_.Laplacian) { cs_.Smagorinsky_Kh = 0; cs_.bound_Kh = 0; cs_.better_bound_Kh = 0; }
  if (!cs_.biharmonic) { cs_.Smagorinsky_Ah = 0; cs_.bound_Ah = 0; cs_.better_bound_Ah = 0; }
  if (!cs_.Smagorinsky_Ah) cs_.bound_Coriolis = 0;
  if (!cs_.Laplacian) cs_.Leith_Kh = 0;     /* :2473 */
  if (!cs_.biharmonic) cs_.Leith_Ah = 0;    /* :2561 */
  const mom6x_hor_visc_params *CS = &cs_;
  const size_t slab = (size_t)d->slab;
  const int st = d->pitch;
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  memset(P, 0, sizeof(double) * slab * ORC_HV_COUNT);
  if (CS->no_slip && CS->biharmonic) return MOM6X_EINVAL;   /* :2723-2725 */
  if (!(CS->Laplacian || CS->biharmonic)) return MOM6X_OK;
  const double *dxBu = M(dxBu), *dyBu = M(dyBu), *IdxBu = M(IdxBu), *IdyBu = M(IdyBu), *dxT = M(dxT), *dyT = M(dyT);
  const double *IdxT = M(IdxT), *IdyT = M(IdyT), *IdxCu = M(IdxCu), *IdyCu = M(IdyCu), *IdxCv = M(IdxCv), *IdyCv = M(IdyCv);
  const double *IareaCu = M(IareaCu), *IareaCv = M(IareaCv), *dy_Cu = M(dy_Cu), *dyCu = M(dyCu), *dx_Cv = M(dx_Cv), *dxCv = M(dxCv);
  const double *fBu = M(CoriolisBu);
  const double Idt = 1.0 / CS->dt;
  if ((CS->Leith_Kh || CS->Leith_Ah) && CS->use_beta_in_Leith) {
    /* G%dF_dx, G%dF_dy: MOM_calculate_grad_Coriolis (MOM_shared_initialization.F90:91-122) over the computational domain,
     * then pass_vector(..., stagger=AGRID): a halo update of the two h-point planes */
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX2(d, i, j);
      double f1 = 0.5 * (fBu[x] + fBu[x - st]), f2 = 0.5 * (fBu[x - 1] + fBu[x - 1 - st]);
      PL(HV_dF_dx)[x] = IdxT[x] * (f1 - f2);
      f1 = 0.5 * (fBu[x] + fBu[x - 1]); f2 = 0.5 * (fBu[x - st] + fBu[x - 1 - st]);
      PL(HV_dF_dy)[x] = IdyT[x] * (f1 - f2);
    }
    orc_pass_var(d, PL(HV_dF_dx), 0, 1); orc_pass_var(d, PL(HV_dF_dy), 0, 1);
  }
  for (int J = js - 2; J <= Jeq + 1; J++) for (int I = is - 2; I <= Ieq + 1; I++) {   /* :2869-2889 */
    size_t x = IX2(d, I, J);
    PL(HV_dx2q)[x] = dxBu[x] * dxBu[x]; PL(HV_dy2q)[x] = dyBu[x] * dyBu[x];
    PL(HV_DX_dyBu)[x] = dxBu[x] * IdyBu[x]; PL(HV_DY_dxBu)[x] = dyBu[x] * IdxBu[x];
  }
  for (int j = js - 2; j <= Jeq + 2; j++) for (int i = is - 2; i <= Ieq + 2; i++) {   /* :2890-2893 */
    size_t x = IX2(d, i, j);
    PL(HV_dx2h)[x] = dxT[x] * dxT[x]; PL(HV_dy2h)[x] = dyT[x] * dyT[x];
    PL(HV_DX_dyT)[x] = dxT[x] * IdyT[x]; PL(HV_DY_dxT)[x] = dyT[x] * IdxT[x];
  }
  for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {         /* reduction_xx :2894-2908 */
    size_t x = IX2(d, i, j);
    double r = 1.0;
    if ((dy_Cu[x] > 0.0) && (dy_Cu[x] < dyCu[x]) && (dy_Cu[x] < dyCu[x] * r)) r = dy_Cu[x] / (dyCu[x]);
    if ((dy_Cu[x - 1] > 0.0) && (dy_Cu[x - 1] < dyCu[x - 1]) && (dy_Cu[x - 1] < dyCu[x - 1] * r)) r = dy_Cu[x - 1] / (dyCu[x - 1]);
    if ((dx_Cv[x] > 0.0) && (dx_Cv[x] < dxCv[x]) && (dx_Cv[x] < dxCv[x] * r)) r = dx_Cv[x] / (dxCv[x]);
    if ((dx_Cv[x - st] > 0.0) && (dx_Cv[x - st] < dxCv[x - st]) && (dx_Cv[x - st] < dxCv[x - st] * r)) r = dx_Cv[x - st] / (dxCv[x - st]);
    PL(HV_red_xx)[x] = r;
  }
  for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {           /* reduction_xy :2909-2923 */
    size_t x = IX2(d, I, J);
    double r = 1.0;
    if ((dy_Cu[x] > 0.0) && (dy_Cu[x] < dyCu[x]) && (dy_Cu[x] < dyCu[x] * r)) r = dy_Cu[x] / (dyCu[x]);
    if ((dy_Cu[x + st] > 0.0) && (dy_Cu[x + st] < dyCu[x + st]) && (dy_Cu[x + st] < dyCu[x + st] * r)) r = dy_Cu[x + st] / (dyCu[x + st]);
    if ((dx_Cv[x] > 0.0) && (dx_Cv[x] < dxCv[x]) && (dx_Cv[x] < dxCv[x] * r)) r = dx_Cv[x] / (dxCv[x]);
    if ((dx_Cv[x + 1] > 0.0) && (dx_Cv[x + 1] < dxCv[x + 1]) && (dx_Cv[x + 1] < dxCv[x + 1] * r)) r = dx_Cv[x + 1] / (dxCv[x + 1]);
    PL(HV_red_xy)[x] = r;
  }
  if (CS->Laplacian) {                                                                 /* :2924-2976 */
    const double Kh_Limit = 0.3 / (CS->dt * 4.0);
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      const double g2 = (2.0 * PL(HV_dx2h)[x] * PL(HV_dy2h)[x]) / (PL(HV_dx2h)[x] + PL(HV_dy2h)[x]);
      if (CS->Smagorinsky_Kh) PL(HV_Lap2_xx)[x] = CS->Smag_Lap_const * g2;
      if (CS->Leith_Kh) PL(HV_Lap3_xx)[x] = CS->Leith_Lap_const * (g2 * sqrt(g2));            /* :2900-2902 */
      PL(HV_Kh_bg_xx)[x] = orc_max(CS->Kh, CS->Kh_vel_scale * sqrt(g2));
      if (CS->bound_Kh && !CS->better_bound_Kh) {
        PL(HV_Kh_Max_xx)[x] = Kh_Limit * g2;
        PL(HV_Kh_bg_xx)[x] = orc_min(PL(HV_Kh_bg_xx)[x], PL(HV_Kh_Max_xx)[x]);
      }
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      size_t x = IX2(d, I, J);
      const double g2 = (2.0 * PL(HV_dx2q)[x] * PL(HV_dy2q)[x]) / (PL(HV_dx2q)[x] + PL(HV_dy2q)[x]);
      if (CS->Smagorinsky_Kh) PL(HV_Lap2_xy)[x] = CS->Smag_Lap_const * g2;
      if (CS->Leith_Kh) PL(HV_Lap3_xy)[x] = CS->Leith_Lap_const * (g2 * sqrt(g2));            /* :2925-2927 */
      PL(HV_Kh_bg_xy)[x] = orc_max(CS->Kh, CS->Kh_vel_scale * sqrt(g2));
      if (CS->bound_Kh && !CS->better_bound_Kh) {
        PL(HV_Kh_Max_xy)[x] = Kh_Limit * g2;
        PL(HV_Kh_bg_xy)[x] = orc_min(PL(HV_Kh_bg_xy)[x], PL(HV_Kh_Max_xy)[x]);
      }
    }
  }
  if (CS->biharmonic) {                                                                /* :2977-3024 */
    for (int j = js - 1; j <= Jeq + 1; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
      size_t x = IX2(d, I, j);
      PL(HV_Idx2dyCu)[x] = (IdxCu[x] * IdxCu[x]) * IdyCu[x];
      PL(HV_Idxdy2u)[x] = IdxCu[x] * (IdyCu[x] * IdyCu[x]);
    }
    for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, J);
      PL(HV_Idx2dyCv)[x] = (IdxCv[x] * IdxCv[x]) * IdyCv[x];
      PL(HV_Idxdy2v)[x] = IdxCv[x] * (IdyCv[x] * IdyCv[x]);
    }
    const double Ah_Limit = 0.3 / (CS->dt * 64.0);
    double BoundCorConst = 0.0;
    if (CS->Smagorinsky_Ah && CS->bound_Coriolis) BoundCorConst = 1.0 / (5.0 * (CS->bound_Cor_vel * CS->bound_Cor_vel));
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      const double g2 = (2.0 * PL(HV_dx2h)[x] * PL(HV_dy2h)[x]) / (PL(HV_dx2h)[x] + PL(HV_dy2h)[x]);
      if (CS->Smagorinsky_Ah) {
        PL(HV_Bih_xx)[x] = CS->Smag_bi_const * (g2 * g2);
        if (CS->bound_Coriolis) {
          const double fmax = max4(fabs(fBu[x - 1 - st]), fabs(fBu[x - st]), fabs(fBu[x - 1]), fabs(fBu[x]));
          PL(HV_Bih2_xx)[x] = (g2 * g2 * g2) * (fmax * BoundCorConst);
        }
      }
      if (CS->Leith_Ah) { const double g3 = g2 * sqrt(g2); PL(HV_Bih6_xx)[x] = CS->Leith_bi_const * (g3 * g3); }   /* :2984-2986 */
      PL(HV_Ah_bg_xx)[x] = orc_max(CS->Ah, CS->Ah_vel_scale * g2 * sqrt(g2));
      if (CS->Ah_time_scale > 0.) PL(HV_Ah_bg_xx)[x] = orc_max(PL(HV_Ah_bg_xx)[x], (g2 * g2) / CS->Ah_time_scale);
      if (CS->bound_Ah && !CS->better_bound_Ah) {
        PL(HV_Ah_Max_xx)[x] = Ah_Limit * (g2 * g2);
        PL(HV_Ah_bg_xx)[x] = orc_min(PL(HV_Ah_bg_xx)[x], PL(HV_Ah_Max_xx)[x]);
      }
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      size_t x = IX2(d, I, J);
      const double g2 = (2.0 * PL(HV_dx2q)[x] * PL(HV_dy2q)[x]) / (PL(HV_dx2q)[x] + PL(HV_dy2q)[x]);
      if (CS->Smagorinsky_Ah) {
        PL(HV_Bih_xy)[x] = CS->Smag_bi_const * (g2 * g2);
        if (CS->bound_Coriolis) PL(HV_Bih2_xy)[x] = (g2 * g2 * g2) * (fabs(fBu[x]) * BoundCorConst);
      }
      if (CS->Leith_Ah) { const double g3 = g2 * sqrt(g2); PL(HV_Bih6_xy)[x] = CS->Leith_bi_const * (g3 * g3); }   /* :3014-3016 */
      PL(HV_Ah_bg_xy)[x] = orc_max(CS->Ah, CS->Ah_vel_scale * g2 * sqrt(g2));
      if (CS->Ah_time_scale > 0.) PL(HV_Ah_bg_xy)[x] = orc_max(PL(HV_Ah_bg_xy)[x], (g2 * g2) / CS->Ah_time_scale);
      if (CS->bound_Ah && !CS->better_bound_Ah) {
        PL(HV_Ah_Max_xy)[x] = Ah_Limit * (g2 * g2);
        PL(HV_Ah_bg_xy)[x] = orc_min(PL(HV_Ah_bg_xy)[x], PL(HV_Ah_Max_xy)[x]);
      }
    }
  }
  const double *dx2h = PL(HV_dx2h), *dy2h = PL(HV_dy2h), *dx2q = PL(HV_dx2q), *dy2q = PL(HV_dy2q);
  const double *DX_dyT = PL(HV_DX_dyT), *DY_dxT = PL(HV_DY_dxT), *DX_dyBu = PL(HV_DX_dyBu), *DY_dxBu = PL(HV_DY_dxBu);
  if (CS->Laplacian && CS->better_bound_Kh) {                                          /* :3025-3048 */
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      const double denom = orc_max(
          (dy2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1]) * orc_max(IdyCu[x] * IareaCu[x], IdyCu[x - 1] * IareaCu[x - 1])),
          (dx2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st]) * orc_max(IdxCv[x] * IareaCv[x], IdxCv[x - st] * IareaCv[x - st])));
      PL(HV_Kh_Max_xx)[x] = 0.0;
      if (denom > 0.0) PL(HV_Kh_Max_xx)[x] = CS->bound_coef * 0.25 * Idt / denom;
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      size_t x = IX2(d, I, J);
      const double denom = orc_max(
          (dx2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x]) * orc_max(IdxCu[x] * IareaCu[x], IdxCu[x + st] * IareaCu[x + st])),
          (dy2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x]) * orc_max(IdyCv[x] * IareaCv[x], IdyCv[x + 1] * IareaCv[x + 1])));
      PL(HV_Kh_Max_xy)[x] = 0.0;
      if (denom > 0.0) PL(HV_Kh_Max_xy)[x] = CS->bound_coef * 0.25 * Idt / denom;
    }
  }
  if (CS->biharmonic && CS->better_bound_Ah) {                                         /* :3054-3110 */
    const double *Idxdy2u = PL(HV_Idxdy2u), *Idx2dyCu = PL(HV_Idx2dyCu), *Idxdy2v = PL(HV_Idxdy2v), *Idx2dyCv = PL(HV_Idx2dyCv);
    double *u0u = PL(HV_u0u), *u0v = PL(HV_u0v), *v0u = PL(HV_v0u), *v0v = PL(HV_v0v);
    for (int j = js - 1; j <= Jeq + 1; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
      size_t x = IX2(d, I, j);
      u0u[x] = ((Idxdy2u[x] * ((dy2h[x + 1] * DY_dxT[x + 1] * (IdyCu[x + 1] + IdyCu[x])) + (dy2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1])))) +
                (Idx2dyCu[x] * ((dx2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x])) + (dx2q[x - st] * DX_dyBu[x - st] * (IdxCu[x] + IdxCu[x - st])))));
      u0v[x] = ((Idxdy2u[x] * ((dy2h[x + 1] * DX_dyT[x + 1] * (IdxCv[x + 1] + IdxCv[x + 1 - st])) + (dy2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st])))) +
                (Idx2dyCu[x] * ((dx2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x])) + (dx2q[x - st] * DY_dxBu[x - st] * (IdyCv[x + 1 - st] + IdyCv[x - st])))));
    }
    for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, J);
      v0u[x] = ((Idxdy2v[x] * ((dy2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x])) + (dy2q[x - 1] * DX_dyBu[x - 1] * (IdxCu[x - 1 + st] + IdxCu[x - 1])))) +
                (Idx2dyCv[x] * ((dx2h[x + st] * DY_dxT[x + st] * (IdyCu[x + st] + IdyCu[x - 1 + st])) + (dx2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1])))));
      v0v[x] = ((Idxdy2v[x] * ((dy2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x])) + (dy2q[x - 1] * DY_dxBu[x - 1] * (IdyCv[x] + IdyCv[x - 1])))) +
                (Idx2dyCv[x] * ((dx2h[x + st] * DX_dyT[x + st] * (IdxCv[x + st] + IdxCv[x])) + (dx2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st])))));
    }
    for (int j = js - 1; j <= Jeq + 1; j++) for (int i = is - 1; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      const double denom = orc_max(
          (dy2h[x] * ((DY_dxT[x] * ((IdyCu[x] * u0u[x]) + (IdyCu[x - 1] * u0u[x - 1]))) + (DX_dyT[x] * ((IdxCv[x] * v0u[x]) + (IdxCv[x - st] * v0u[x - st])))) *
           orc_max(IdyCu[x] * IareaCu[x], IdyCu[x - 1] * IareaCu[x - 1])),
          (dx2h[x] * ((DY_dxT[x] * ((IdyCu[x] * u0v[x]) + (IdyCu[x - 1] * u0v[x - 1]))) + (DX_dyT[x] * ((IdxCv[x] * v0v[x]) + (IdxCv[x - st] * v0v[x - st])))) *
           orc_max(IdxCv[x] * IareaCv[x], IdxCv[x - st] * IareaCv[x - st])));
      PL(HV_Ah_Max_xx)[x] = 0.0;
      if (denom > 0.0) PL(HV_Ah_Max_xx)[x] = CS->bound_coef * 0.5 * Idt / denom;
    }
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      size_t x = IX2(d, I, J);
      const double denom = orc_max(
          (dx2q[x] * ((DX_dyBu[x] * ((u0u[x + st] * IdxCu[x + st]) + (u0u[x] * IdxCu[x]))) + (DY_dxBu[x] * ((v0u[x + 1] * IdyCv[x + 1]) + (v0u[x] * IdyCv[x])))) *
           orc_max(IdxCu[x] * IareaCu[x], IdxCu[x + st] * IareaCu[x + st])),
          (dy2q[x] * ((DX_dyBu[x] * ((u0v[x + st] * IdxCu[x + st]) + (u0v[x] * IdxCu[x]))) + (DY_dxBu[x] * ((v0v[x + 1] * IdyCv[x + 1]) + (v0v[x] * IdyCv[x])))) *
           orc_max(IdyCv[x] * IareaCv[x], IdyCv[x + 1] * IareaCv[x + 1])));
      PL(HV_Ah_Max_xy)[x] = 0.0;
      if (denom > 0.0) PL(HV_Ah_Max_xy)[x] = CS->bound_coef * 0.5 * Idt / denom;
    }
  }
  return MOM6X_OK;
}

/* horizontal_viscosity :266-2317, one layer at a time as in the reference */
int orc_horizontal_viscosity(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_hor_visc_params *CS_in,
                             const double *P, const double *u, const double *v, const double *h, double *diffu, double *diffv) {
  mom6x_hor_visc_params cs_ = *CS_in;   /* hor_visc_init :2465, :2495-2496, :2555-2571, :2624 */
  if (!cs_.Laplacian) { cs_.Smagorinsky_Kh = 0; cs_.bound_Kh = 0; cs_.better_bound_Kh = 0; }
  if (!cs_.biharmonic) { cs_.Smagorinsky_Ah = 0; cs_.bound_Ah = 0; cs_.better_bound_Ah = 0; }
  if (!cs_.Smagorinsky_Ah) cs_.bound_Coriolis = 0;
  if (!cs_.Laplacian) cs_.Leith_Kh = 0;     /* :2473 */
  if (!cs_.biharmonic) cs_.Leith_Ah = 0;    /* :2561 */
  const mom6x_hor_visc_params *CS = &cs_;
  const size_t slab = (size_t)d->slab;
  const int st = d->pitch, nz = d->nk;
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  if (!(CS->Laplacian || CS->biharmonic)) return MOM6X_OK;
  const double *IdxCu = M(IdxCu), *IdyCu = M(IdyCu), *IdxCv = M(IdxCv), *IdyCv = M(IdyCv), *IareaCu = M(IareaCu), *IareaCv = M(IareaCv);
  const double *mT = M(mask2dT), *mBu = M(mask2dBu), *mCu = M(mask2dCu), *mCv = M(mask2dCv);
  const double *dx2h = P + (size_t)HV_dx2h * slab, *dy2h = P + (size_t)HV_dy2h * slab, *dx2q = P + (size_t)HV_dx2q * slab, *dy2q = P + (size_t)HV_dy2q * slab;
  const double *DX_dyT = P + (size_t)HV_DX_dyT * slab, *DY_dxT = P + (size_t)HV_DY_dxT * slab;
  const double *DX_dyBu = P + (size_t)HV_DX_dyBu * slab, *DY_dxBu = P + (size_t)HV_DY_dxBu * slab;
  const double *red_xx = P + (size_t)HV_red_xx * slab, *red_xy = P + (size_t)HV_red_xy * slab;
  const double *Kh_bg_xx = P + (size_t)HV_Kh_bg_xx * slab, *Kh_bg_xy = P + (size_t)HV_Kh_bg_xy * slab;
  const double *Kh_Max_xx = P + (size_t)HV_Kh_Max_xx * slab, *Kh_Max_xy = P + (size_t)HV_Kh_Max_xy * slab;
  const double *Lap2_xx = P + (size_t)HV_Lap2_xx * slab, *Lap2_xy = P + (size_t)HV_Lap2_xy * slab;
  const double *Idx2dyCu = P + (size_t)HV_Idx2dyCu * slab, *Idxdy2u = P + (size_t)HV_Idxdy2u * slab;
  const double *Idx2dyCv = P + (size_t)HV_Idx2dyCv * slab, *Idxdy2v = P + (size_t)HV_Idxdy2v * slab;
  const double *Ah_bg_xx = P + (size_t)HV_Ah_bg_xx * slab, *Ah_bg_xy = P + (size_t)HV_Ah_bg_xy * slab;
  const double *Ah_Max_xx = P + (size_t)HV_Ah_Max_xx * slab, *Ah_Max_xy = P + (size_t)HV_Ah_Max_xy * slab;
  const double *Bih_xx = P + (size_t)HV_Bih_xx * slab, *Bih_xy = P + (size_t)HV_Bih_xy * slab;
  const double *Bih2_xx = P + (size_t)HV_Bih2_xx * slab, *Bih2_xy = P + (size_t)HV_Bih2_xy * slab;
  const double h_neglect = GV->H_subroundoff, h_neglect3 = h_neglect * h_neglect * h_neglect;
  const double *Lap3_xx = P + (size_t)HV_Lap3_xx * slab, *Lap3_xy = P + (size_t)HV_Lap3_xy * slab;
  const double *Bih6_xx = P + (size_t)HV_Bih6_xx * slab, *Bih6_xy = P + (size_t)HV_Bih6_xy * slab;
  const double *dF_dx = P + (size_t)HV_dF_dx * slab, *dF_dy = P + (size_t)HV_dF_dy * slab;
  const int leith = CS->Leith_Kh || CS->Leith_Ah;
  if (leith && d->halo < 3) return MOM6X_EINVAL;   /* "The minimum halo size is 3 when a Leith viscosity is being used." :550 */
  const double inv_PI3 = 1.0 / ((4.0 * atan(1.0)) * (4.0 * atan(1.0)) * (4.0 * atan(1.0))), inv_PI6 = inv_PI3 * inv_PI3;   /* :481-483 */
  const int legacy_bound = (CS->Smagorinsky_Kh || CS->Leith_Kh) && (CS->bound_Kh && !CS->better_bound_Kh);   /* :556-557 */
  const int smag = CS->Smagorinsky_Kh || CS->Smagorinsky_Ah, better = CS->better_bound_Ah || CS->better_bound_Kh;
  /* layers are independent (the reference: !$OMP parallel do over k, MOM_hor_visc.F90:690): every thread its own work planes */
#pragma omp parallel
  {
  double *w = (double *)calloc(slab * 25, sizeof(double));
  double *sh_xx = w, *sh_xy = w + slab, *h_u = w + 2 * slab, *h_v = w + 3 * slab, *Del2u = w + 4 * slab, *Del2v = w + 5 * slab;
  double *str_xx = w + 6 * slab, *str_xy = w + 7 * slab, *Shear = w + 8 * slab, *hrat = w + 9 * slab, *vbr = w + 10 * slab;
  double *dDel2vdx = w + 11 * slab, *dDel2udy = w + 12 * slab, *hq = w + 13 * slab, *Kh = w + 14 * slab, *Ah = w + 15 * slab;
  double *vort = w + 16 * slab, *vdx = w + 17 * slab, *vdy = w + 18 * slab, *D2q = w + 19 * slab, *divx = w + 20 * slab;
  double *ddx = w + 21 * slab, *ddy = w + 22 * slab, *gdh = w + 23 * slab, *gdq = w + 24 * slab;   /* (zero unless MODIFIED_LEITH) */
#pragma omp for schedule(static)
  for (int k = 0; k < nz; k++) {
    const double *uk = u + k * slab, *vk = v + k * slab, *hk = h + k * slab;
    for (int j = Jsq - 1; j <= Jeq + 2; j++) for (int i = Isq - 1; i <= Ieq + 2; i++) {   /* :724-731 */
      size_t x = IX2(d, i, j);
      const double dudx = DY_dxT[x] * ((IdyCu[x] * uk[x]) - (IdyCu[x - 1] * uk[x - 1]));
      const double dvdy = DX_dyT[x] * ((IdxCv[x] * vk[x]) - (IdxCv[x - st] * vk[x - st]));
      sh_xx[x] = dudx - dvdy;
      divx[x] = dudx + dvdy;                                                               /* :1029-1031 (same range) */
    }
    for (int J = js - 2; J <= Jeq + 1; J++) for (int I = is - 2; I <= Ieq + 1; I++) {       /* :733-737, :907-917 */
      size_t x = IX2(d, I, J);
      const double dvdx = DY_dxBu[x] * ((vk[x + 1] * IdyCv[x + 1]) - (vk[x] * IdyCv[x]));
      const double dudy = DX_dyBu[x] * ((uk[x + st] * IdxCu[x + st]) - (uk[x] * IdxCu[x]));
      if (CS->no_slip) sh_xy[x] = (2.0 - mBu[x]) * (dvdx + dudy);
      else sh_xy[x] = mBu[x] * (dvdx + dudy);
    }
    for (int j = js - 2; j <= je + 2; j++) for (int I = is - 2; I <= Ieq + 1; I++) {        /* :767-781 */
      size_t x = IX2(d, I, j);
      if (CS->use_land_mask) h_u[x] = 0.5 * (mT[x] * hk[x] + mT[x + 1] * hk[x + 1]);
      else h_u[x] = 0.5 * (hk[x] + hk[x + 1]);
    }
    for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 2; i <= ie + 2; i++) {
      size_t x = IX2(d, i, J);
      if (CS->use_land_mask) h_v[x] = 0.5 * (mT[x] * hk[x] + mT[x + st] * hk[x + st]);
      else h_v[x] = 0.5 * (hk[x] + hk[x + st]);
    }
    if (CS->biharmonic) {                                                                 /* :934-943 */
      for (int j = js - 1; j <= Jeq + 1; j++) for (int I = Isq - 1; I <= Ieq + 1; I++) {
        size_t x = IX2(d, I, j);
        Del2u[x] = Idx2dyCu[x] * ((dx2q[x] * sh_xy[x]) - (dx2q[x - st] * sh_xy[x - st])) +
                   Idxdy2u[x] * ((dy2h[x + 1] * sh_xx[x + 1]) - (dy2h[x] * sh_xx[x]));
      }
      for (int J = Jsq - 1; J <= Jeq + 1; J++) for (int i = is - 1; i <= Ieq + 1; i++) {
        size_t x = IX2(d, i, J);
        Del2v[x] = Idxdy2v[x] * ((dy2q[x] * sh_xy[x]) - (dy2q[x - 1] * sh_xy[x - 1])) -
                   Idx2dyCv[x] * ((dx2h[x + st] * sh_xx[x + st]) - (dx2h[x] * sh_xx[x]));
      }
    }
    if (leith) {   /* :961-1113 with js_vort = js_Kh-2 = js-3, je_vort = Jeq+2, is_vort = is-3, ie_vort = Ieq+2 (:547-552) */
      for (int J = js - 3; J <= Jeq + 2; J++) for (int I = is - 3; I <= Ieq + 2; I++) {   /* :730-733, :962-972 */
        size_t x = IX2(d, I, J);
        const double dvdx = DY_dxBu[x] * ((vk[x + 1] * IdyCv[x + 1]) - (vk[x] * IdyCv[x]));
        const double dudy = DX_dyBu[x] * ((uk[x + st] * IdxCu[x + st]) - (uk[x] * IdxCu[x]));
        if (CS->no_slip) vort[x] = (2.0 - mBu[x]) * (dvdx - dudy);
        else vort[x] = mBu[x] * (dvdx - dudy);
      }
      /* (these loops re-form DY_dxBu = G%dyBu * G%IdxBu from the grid, :991: not CS%DY_dxBu, which is zero beyond is-2..Ieq+1) */
      const double *dxBu = M(dxBu), *dyBu = M(dyBu), *IdxBu = M(IdxBu), *IdyBu = M(IdyBu);
      for (int J = js - 2; J <= je + 1; J++) for (int i = Isq - 1; i <= ie + 2; i++) {      /* :990-993 */
        size_t x = IX2(d, i, J);
        const double DYX = dyBu[x] * IdxBu[x];
        vdx[x] = DYX * ((vort[x] * IdyCu[x]) - (vort[x - 1] * IdyCu[x - 1]));
      }
      for (int j = Jsq - 1; j <= je + 2; j++) for (int I = is - 2; I <= ie + 1; I++) {      /* :995-998 */
        size_t x = IX2(d, I, j);
        const double DXY = dxBu[x] * IdyBu[x];
        vdy[x] = DXY * ((vort[x] * IdxCv[x]) - (vort[x - st] * IdxCv[x - st]));
      }
      for (int J = Jsq - 1; J <= je + 1; J++) for (int I = Isq - 1; I <= ie + 1; I++) {     /* :1017-1023 */
        size_t x = IX2(d, I, J);
        const double DYX = dyBu[x] * IdxBu[x], DXY = dxBu[x] * IdyBu[x];
        D2q[x] = DYX * ((vdx[x + 1] * IdyCv[x + 1]) - (vdx[x] * IdyCv[x])) +
                 DXY * ((vdy[x + st] * IdyCu[x + st]) - (vdy[x] * IdyCu[x]));
      }
      if (CS->modified_Leith) {                                                            /* :1026-1049 */
        for (int j = js - 1; j <= je + 1; j++) for (int I = Isq - 1; I <= ie + 1; I++) {
          size_t x = IX2(d, I, j);
          ddx[x] = IdxCu[x] * (divx[x + 1] - divx[x]);
        }
        for (int J = Jsq - 1; J <= je + 1; J++) for (int i = is - 1; i <= ie + 1; i++) {
          size_t x = IX2(d, i, J);
          ddy[x] = IdyCv[x] * (divx[x + st] - divx[x]);
        }
        for (int j = Jsq; j <= je + 1; j++) for (int i = Isq; i <= ie + 1; i++) {
          size_t x = IX2(d, i, j);
          const double a = 0.5 * (ddx[x] + ddx[x - 1]), b = 0.5 * (ddy[x] + ddy[x - st]);
          gdh[x] = sqrt((a * a) + (b * b));
        }
        for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
          size_t x = IX2(d, I, J);
          const double a = 0.5 * (ddx[x] + ddx[x + st]), b = 0.5 * (ddy[x] + ddy[x + 1]);
          gdq[x] = sqrt((a * a) + (b * b));
        }
      }
      if (CS->use_beta_in_Leith) {                                                         /* :1069-1076 */
        for (int J = js - 2; J <= Jeq + 1; J++) for (int i = is - 1; i <= ie + 1; i++) {
          size_t x = IX2(d, i, J);
          vdx[x] = vdx[x] + 0.5 * (dF_dx[x] + dF_dx[x + st]);
        }
        for (int j = js - 1; j <= je + 1; j++) for (int I = is - 2; I <= Ieq + 1; I++) {
          size_t x = IX2(d, I, j);
          vdy[x] = vdy[x] + 0.5 * (dF_dy[x] + dF_dy[x + 1]);
        }
      }
    }
    /* ---- h points (js_Kh..je_Kh = Jsq..je+1, is_Kh..ie_Kh = Isq..ie+1) */
    for (int j = Jsq; j <= je + 1; j++) for (int i = Isq; i <= ie + 1; i++) {
      size_t x = IX2(d, i, j);
      if (smag) {                                                                         /* :1112-1119 */
        const double sh_xx_sq = sh_xx[x] * sh_xx[x];
        const double sh_xy_sq = 0.25 * (((sh_xy[x - 1 - st] * sh_xy[x - 1 - st]) + (sh_xy[x] * sh_xy[x])) +
                                        ((sh_xy[x - 1] * sh_xy[x - 1]) + (sh_xy[x - st] * sh_xy[x - st])));
        Shear[x] = sqrt(sh_xx_sq + sh_xy_sq);
      }
      if (better) {                                                                       /* :1120-1125 */
        const double h_min = min4(h_u[x], h_u[x - 1], h_v[x], h_v[x - st]);
        hrat[x] = orc_min(1.0, h_min / (hk[x] + h_neglect));
      }
      if (CS->Laplacian) {                                                                /* :1126-1275 */
        double K = Kh_bg_xx[x];
        double vvm = 0.0;                                                                  /* :1104-1107, :1143-1146 */
        if (leith) {
          const double a = 0.5 * (vdx[x] + vdx[x - st]), b = 0.5 * (vdy[x] + vdy[x - 1]);
          vvm = sqrt((a * a) + (b * b)) + gdh[x];
        }
        if (CS->add_LES_viscosity) {
          if (CS->Smagorinsky_Kh) K = K + Lap2_xx[x] * Shear[x];
          if (CS->Leith_Kh) K = K + Lap3_xx[x] * vvm * inv_PI3;
        } else {
          if (CS->Smagorinsky_Kh) K = orc_max(K, Lap2_xx[x] * Shear[x]);
          if (CS->Leith_Kh) K = orc_max(K, Lap3_xx[x] * vvm * inv_PI3);
        }
        if (legacy_bound) K = orc_min(K, Kh_Max_xx[x]);
        K = orc_max(K, CS->Kh_bg_min);
        if (CS->better_bound_Kh && CS->better_bound_Ah) {
          vbr[x] = 1.0;
          const double Kh_max_here = hrat[x] * Kh_Max_xx[x];
          if (K >= Kh_max_here) { vbr[x] = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (CS->backscatter_underbound && (Kh_max_here > 0.0))) vbr[x] = 1.0 - K / Kh_max_here;
        } else if (CS->better_bound_Kh) {
          K = orc_min(K, hrat[x] * Kh_Max_xx[x]);
        }
        Kh[x] = K;
        str_xx[x] = -K * sh_xx[x];
      } else str_xx[x] = 0.0;
      if (CS->biharmonic) {                                                               /* :1283-1448 */
        double A = Ah_bg_xx[x];
        if (CS->Smagorinsky_Ah || CS->Leith_Ah) {                                          /* :1301-1385 */
          if (CS->Smagorinsky_Ah) {
            double AhSm;
            if (CS->bound_Coriolis) AhSm = Shear[x] * (Bih_xx[x] + Bih2_xx[x] * Shear[x]);
            else AhSm = Bih_xx[x] * Shear[x];
            A = orc_max(A, AhSm);
          }
          if (CS->Leith_Ah) {
            const double Del2vort_h = 0.25 * ((D2q[x] + D2q[x - 1 - st]) + (D2q[x - 1] + D2q[x - st]));
            const double AhLth = Bih6_xx[x] * fabs(Del2vort_h) * inv_PI6;
            A = orc_max(A, AhLth);
          }
          if (CS->bound_Ah && !CS->better_bound_Ah) A = orc_min(A, Ah_Max_xx[x]);
        }
        if (CS->better_bound_Ah) {
          if (CS->better_bound_Kh) A = orc_min(A, vbr[x] * hrat[x] * Ah_Max_xx[x]);
          else A = orc_min(A, hrat[x] * Ah_Max_xx[x]);
        }
        Ah[x] = A;
        const double d_del2u = (IdyCu[x] * Del2u[x]) - (IdyCu[x - 1] * Del2u[x - 1]);
        const double d_del2v = (IdxCv[x] * Del2v[x]) - (IdxCv[x - st] * Del2v[x - st]);
        const double d_str = A * ((DY_dxT[x] * d_del2u) - (DX_dyT[x] * d_del2v));
        str_xx[x] = str_xx[x] + d_str;
      }
    }
    /* ---- q points (js-1..Jeq, is-1..Ieq) */
    for (int J = js - 1; J <= Jeq; J++) for (int I = is - 1; I <= Ieq; I++) {
      size_t x = IX2(d, I, J);
      if (CS->biharmonic) {                                                               /* :1483-1486 */
        dDel2vdx[x] = DY_dxBu[x] * ((Del2v[x + 1] * IdyCv[x + 1]) - (Del2v[x] * IdyCv[x]));
        dDel2udy[x] = DX_dyBu[x] * ((Del2u[x + st] * IdxCu[x + st]) - (Del2u[x] * IdxCu[x]));
      }
      if (smag) {                                                                         /* :1513-1520 */
        const double sh_xy_sq = sh_xy[x] * sh_xy[x];
        const double sh_xx_sq = 0.25 * (((sh_xx[x] * sh_xx[x]) + (sh_xx[x + 1 + st] * sh_xx[x + 1 + st])) +
                                        ((sh_xx[x + st] * sh_xx[x + st]) + (sh_xx[x + 1] * sh_xx[x + 1])));
        Shear[x] = sqrt(sh_xy_sq + sh_xx_sq);
      }
      {                                                                                   /* :1521-1526 */
        const double h2uq = 4.0 * (h_u[x] * h_u[x + st]);
        const double h2vq = 4.0 * (h_v[x] * h_v[x + 1]);
        hq[x] = (2.0 * (h2uq * h2vq)) / (h_neglect3 + (h2uq + h2vq) * ((h_u[x] + h_u[x + st]) + (h_v[x] + h_v[x + 1])));
      }
      if (better) {
        const double h_min = min4(h_u[x], h_u[x + st], h_v[x], h_v[x + 1]);
        hrat[x] = orc_min(1.0, h_min / (hq[x] + h_neglect));
      }
      if (CS->no_slip && (mBu[x] < 0.5)) {                                                /* :1533-1566 */
        if ((mCu[x] + mCu[x + st]) + (mCv[x] + mCv[x + 1]) > 0.0) {
          const double hu = mCu[x] * h_u[x] + mCu[x + st] * h_u[x + st];
          const double hv = mCv[x] * h_v[x] + mCv[x + 1] * h_v[x + 1];
          if ((mCu[x] + mCu[x + st]) * (mCv[x] + mCv[x + 1]) == 0.0) {
            hq[x] = hu + hv;
            hrat[x] = 1.0;
          } else {
            hq[x] = 2.0 * (hu * hv) / ((hu + hv) + h_neglect);
            hrat[x] = orc_min(1.0, orc_min(hu, hv) / (hq[x] + h_neglect));
          }
        }
      }
      if (CS->Laplacian) {                                                                /* :1571-1692 */
        double K = Kh_bg_xy[x];
        if (CS->Smagorinsky_Kh) {
          if (CS->add_LES_viscosity) K = K + Lap2_xy[x] * Shear[x];
          else K = orc_max(K, Lap2_xy[x] * Shear[x]);
        }
        if (CS->Leith_Kh) {                                                                /* :1108-1111, :1587-1590, :1610-1620 */
          const double a = 0.5 * (vdx[x] + vdx[x + 1]), b = 0.5 * (vdy[x] + vdy[x + st]);
          const double vvm = sqrt((a * a) + (b * b)) + gdq[x];
          if (CS->add_LES_viscosity) K = K + Lap3_xy[x] * vvm * inv_PI3;
          else K = orc_max(K, Lap3_xy[x] * vvm * inv_PI3);
        }
        if (legacy_bound) K = orc_min(K, Kh_Max_xy[x]);
        K = orc_max(K, CS->Kh_bg_min);
        if (CS->better_bound_Kh && CS->better_bound_Ah) {
          vbr[x] = 1.0;
          const double Kh_max_here = hrat[x] * Kh_Max_xy[x];
          if (K >= Kh_max_here) { vbr[x] = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (CS->backscatter_underbound && (Kh_max_here > 0.0))) vbr[x] = 1.0 - K / Kh_max_here;
        } else if (CS->better_bound_Kh) {
          K = orc_min(K, hrat[x] * Kh_Max_xy[x]);
        }
        str_xy[x] = -K * sh_xy[x];
      } else str_xy[x] = 0.;
      if (CS->biharmonic) {                                                               /* :1714-1826 */
        double A = Ah_bg_xy[x];
        if (CS->Smagorinsky_Ah || CS->Leith_Ah) {                                          /* :1745-1773 */
          if (CS->Smagorinsky_Ah) {
            double AhSm;
            if (CS->bound_Coriolis) AhSm = Shear[x] * (Bih_xy[x] + Bih2_xy[x] * Shear[x]);
            else AhSm = Bih_xy[x] * Shear[x];
            A = orc_max(A, AhSm);
          }
          if (CS->Leith_Ah) A = orc_max(A, Bih6_xy[x] * fabs(D2q[x]) * inv_PI6);
          if (CS->bound_Ah && !CS->better_bound_Ah) A = orc_min(A, Ah_Max_xy[x]);
        }
        if (CS->better_bound_Ah) {
          if (CS->better_bound_Kh) A = orc_min(A, vbr[x] * hrat[x] * Ah_Max_xy[x]);
          else A = orc_min(A, hrat[x] * Ah_Max_xy[x]);
        }
        const double d_str = A * (dDel2vdx[x] + dDel2udy[x]);
        str_xy[x] = str_xy[x] + d_str;
      }
      if (CS->no_slip) str_xy[x] = str_xy[x] * (hq[x] * red_xy[x]);                       /* :1896-1906 */
      else str_xy[x] = str_xy[x] * (hq[x] * mBu[x] * red_xy[x]);
    }
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {           /* :1893-1895 */
      size_t x = IX2(d, i, j);
      str_xx[x] = str_xx[x] * (hk[x] * red_xx[x]);
    }
    for (int j = js; j <= je; j++) for (int I = Isq; I <= Ieq; I++) {                     /* :1910-1914 */
      size_t x = IX2(d, I, j);
      diffu[x + k * slab] = ((IdxCu[x] * ((dx2q[x - st] * str_xy[x - st]) - (dx2q[x] * str_xy[x])) +
                              IdyCu[x] * ((dy2h[x] * str_xx[x]) - (dy2h[x + 1] * str_xx[x + 1]))) * IareaCu[x]) / (h_u[x] + h_neglect);
    }
    for (int J = Jsq; J <= Jeq; J++) for (int i = is; i <= ie; i++) {                     /* :1927-1931 */
      size_t x = IX2(d, i, J);
      diffv[x + k * slab] = ((IdyCv[x] * ((dy2q[x - 1] * str_xy[x - 1]) - (dy2q[x] * str_xy[x])) -
                              IdxCv[x] * ((dx2h[x] * str_xx[x]) - (dx2h[x + st] * str_xx[x + st]))) * IareaCv[x]) / (h_v[x] + h_neglect);
    }
  }
  free(w);
  }   /* omp parallel */
  return MOM6X_OK;
}
