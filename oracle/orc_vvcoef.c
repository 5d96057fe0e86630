/* orc_vvcoef.c -- vertvisc_coef + find_coupling_coef (MOM_vert_friction.F90:1357-2310 / :2314-2925).
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * The branches restated: HARMONIC_VISC on/off (:1568-1617), BOTTOMDRAGLAW on/off (:1519-1525, :2497-2540),
 * KV_ML_INVZ2 (:2420-2436), visc%Kv_shear (:2439-2476).  Not restated (rejected by the callers): ice shelves,
 * OBCs, GL90, Kv_shear_Bu, dynamic / LOTW mixed-layer viscosities.  dz = H_to_Z*h (thickness_to_dz :892). */
#include "orc_common.h"

static void coef_dir(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_vertvisc_params *CS, int dir,
                     const double *u, const double *h, double dt, const double *Kv_bbl, const double *bbl_thick_in,
                     const double *Kv_shear, double *a_out, double *h_out) {
  const int nz = d->nk, st = dir ? d->pitch : 1;
  const size_t slab = (size_t)d->slab;
  const int a0 = dir ? 0 : -1, a1 = d->ni - 1, b0 = dir ? -1 : 0, b1 = d->nj - 1;
  const double *maskC = GM(G, d, dir ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double h_neglect = GV->H_subroundoff, dz_neglect = GV->dZ_subroundoff;
  const double a_cpl_max = 1.0e37 * GV->Z_to_H;
  double I_valBL = 0.0; if (CS->harm_BL_val > 0.0) I_valBL = 1.0 / CS->harm_BL_val;
  const double I_amax = (CS->answer_date < 20190101) ? (1.0e-10 * GV->H_to_Z) * dt : 0.0;   /* :2391-2395 */
  /* columns are independent (the reference: !$OMP parallel do over j, MOM_vert_friction.F90:1508) */
#pragma omp parallel
  {
  double *hvel = (double *)calloc(nz, sizeof(double)), *dz_vel = (double *)calloc(nz, sizeof(double));
  double *dz_harm = (double *)calloc(nz, sizeof(double)), *z_i = (double *)calloc(nz + 1, sizeof(double));
  double *a_cpl = (double *)calloc(nz + 1, sizeof(double));
#pragma omp for schedule(static)
  for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
    const size_t x = IX2(d, i, j), y = x + st;
    if (!(maskC[x] > 0.)) continue;                                   /* do_i :1514-1516 */
    double I_Hbbl = 1. / (CS->Hbbl + dz_neglect), kv_bbl = 0.0, bbl_thick = 0.0;
    if (CS->bottomdraglaw) {                                          /* :1519-1525 */
      kv_bbl = Kv_bbl[x];
      bbl_thick = bbl_thick_in[x] + dz_neglect;
      I_Hbbl = 1. / bbl_thick;
    }
    const double Dmin = orc_min(bathyT[x], bathyT[y]);
    double zh = 0., zcol0 = -bathyT[x], zcol1 = -bathyT[y];
    z_i[nz] = 0.;
    for (int k = nz - 1; k >= 0; k--) {                               /* :1568-1650 */
      const double h0 = h[x + k * slab], h1 = h[y + k * slab];
      const double dz0 = GV->H_to_Z * h0, dz1 = GV->H_to_Z * h1;
      const double h_harm = 2. * h0 * h1 / (h0 + h1 + h_neglect);
      const double h_arith = 0.5 * (h1 + h0);
      const double h_delta = h1 - h0;
      dz_harm[k] = 2. * dz0 * dz1 / (dz0 + dz1 + dz_neglect);
      const double dz_arith = 0.5 * (dz1 + dz0);
      const double uk = u[x + k * slab];
      if (CS->harmonic_visc) {
        hvel[k] = h_harm; dz_vel[k] = dz_harm[k];
        if (uk * h_delta < 0) {
          const double z2 = z_i[k + 1];
          const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
          hvel[k] = (1. - botfn) * h_harm + botfn * h_arith;
          dz_vel[k] = (1. - botfn) * dz_harm[k] + botfn * dz_arith;
        }
        z_i[k] = z_i[k + 1] + dz_harm[k] * I_Hbbl;
      } else {
        zcol0 = zcol0 + dz0; zcol1 = zcol1 + dz1;
        zh = zh + dz_harm[k];
        const double z_clear = orc_max(zcol0, zcol1) + Dmin;
        z_i[k] = orc_max(zh, z_clear) * I_Hbbl;
        hvel[k] = h_arith; dz_vel[k] = dz_arith;
        if (uk * h_delta > 0.) {
          if (zh * I_Hbbl < CS->harm_BL_val) {
            hvel[k] = h_harm; dz_vel[k] = dz_harm[k];
          } else {
            double z2_wt = 1.;
            if (zh * I_Hbbl < 2. * CS->harm_BL_val) z2_wt = orc_max(0., orc_min(1., zh * I_Hbbl * I_valBL - 1.));
            const double z2 = z2_wt * (orc_max(zh, z_clear) * I_Hbbl);
            const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
            hvel[k] = (1. - botfn) * h_arith + botfn * h_harm;
            dz_vel[k] = (1. - botfn) * dz_arith + botfn * dz_harm[k];
          }
        }
      }
    }
    /* find_coupling_coef(a_cpl, dz_vel, do_i, dz_harm, bbl_thick, kv_bbl, z_i, ...) :2314 */
    for (int K = 0; K <= nz; K++) a_cpl[K] = 0.0;
    const double hn = GV->dZ_subroundoff;   /* h_neglect of find_coupling_coef :2390 */
    double z_t = 0.0, I_Hmix = 0.0;
    if (CS->Kvml_invZ2 > 0.) { I_Hmix = 1. / (CS->Hmix + hn); z_t = hn * I_Hmix; }
    for (int K = 1; K < nz; K++) {          /* Fortran K = 2..nz : interface between layers K-1 and K (0-based k-1, k) */
      double Kv_tot = CS->Kv;
      if (CS->Kvml_invZ2 > 0.) {
        z_t = z_t + dz_harm[K - 1] * I_Hmix;
        Kv_tot = CS->Kv + CS->Kvml_invZ2 / ((z_t * z_t) * (1. + 0.09 * z_t * z_t * z_t * z_t * z_t * z_t));
      }
      if (Kv_shear) {
        const double Kv_add = 0.5 * (Kv_shear[x + K * slab] + Kv_shear[y + K * slab]);
        Kv_tot = Kv_tot + Kv_add;
      }
      if (CS->bottomdraglaw) {
        const double z2 = z_i[K];
        const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + (kv_bbl - CS->Kv) * botfn;
        const double dhc = 0.5 * (dz_vel[K] + dz_vel[K - 1]);
        double h_shear;
        if (dhc > bbl_thick) h_shear = ((1. - botfn) * dhc + botfn * bbl_thick) + hn;
        else h_shear = dhc + hn;
        a_cpl[K] = Kv_tot / (h_shear + (I_amax * Kv_tot));
      } else if (fabs(CS->Kv_extra_bbl) > 0.0) {
        const double z2 = z_i[K];
        const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + CS->Kv_extra_bbl * botfn;
        const double h_shear = 0.5 * (dz_vel[K] + dz_vel[K - 1] + hn);
        a_cpl[K] = Kv_tot / (h_shear + I_amax * Kv_tot);
      } else {
        const double h_shear = 0.5 * (dz_vel[K] + dz_vel[K - 1] + hn);
        a_cpl[K] = Kv_tot / (h_shear + I_amax * Kv_tot);
      }
    }
    if (CS->bottomdraglaw) {                /* :2543-2549 */
      const double dhc = dz_vel[nz - 1] * 0.5;
      a_cpl[nz] = kv_bbl / ((orc_min(dhc, bbl_thick) + hn) + I_amax * kv_bbl);
    } else if (fabs(CS->Kv_extra_bbl) > 0.0) {
      a_cpl[nz] = (CS->Kv + CS->Kv_extra_bbl) / ((0.5 * dz_vel[nz - 1] + hn) + I_amax * (CS->Kv + CS->Kv_extra_bbl));
    } else {
      a_cpl[nz] = CS->Kv / ((0.5 * dz_vel[nz - 1] + hn) + I_amax * CS->Kv);
    }
    for (int K = 0; K <= nz; K++) a_out[x + K * slab] = orc_min(a_cpl_max, a_cpl[K]);   /* :1863-1867 */
    for (int k = 0; k < nz; k++) h_out[x + k * slab] = hvel[k] + h_neglect;             /* :1868-1872 */
  }
  free(hvel); free(dz_vel); free(dz_harm); free(z_i); free(a_cpl);
  }   /* omp parallel */
}

int orc_vertvisc_coef(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_vertvisc_params *CS,
                      const double *u, const double *v, const double *h, double dt, const double *Kv_bbl_u,
                      const double *Kv_bbl_v, const double *bbl_thick_u, const double *bbl_thick_v, const double *Kv_shear,
                      double *a_u, double *a_v, double *h_u, double *h_v) {
  if (CS->bottomdraglaw && !(Kv_bbl_u && Kv_bbl_v && bbl_thick_u && bbl_thick_v)) return MOM6X_EINVAL;
  coef_dir(d, G, GV, CS, 0, u, h, dt, Kv_bbl_u, bbl_thick_u, Kv_shear, a_u, h_u);
  coef_dir(d, G, GV, CS, 1, v, h, dt, Kv_bbl_v, bbl_thick_v, Kv_shear, a_v, h_v);
  return MOM6X_OK;
}
