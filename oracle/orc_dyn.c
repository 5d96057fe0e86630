/*
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * CorAdCalc / gradKE          <- src/core/MOM_CoriolisAdv.F90:125-1052
 * PressureForce_FV_Bouss      <- src/core/MOM_PressureForce_FV.F90:947-2017 (layered, no-EOS path) with
 *   Set_pbce_Bouss            <- src/core/MOM_PressureForce_Montgomery.F90:649-748 (no-EOS branch)
 * vertvisc / vertvisc_remnant <- src/parameterizations/vertical/MOM_vert_friction.F90:557-1356
 *
 * Supported options (others return MOM6X_EUNSUPPORTED):
 *   CORIOLIS_SCHEME = SADOURNY75_ENERGY (default) | SADOURNY75_ENSTRO | ARAKAWA_HSU90 | ARAKAWA_LAMB81 | ARAKAWA_LAMB_BLEND |
 *   ROBUST_ENSTRO (with PV_ADV_SCHEME = PV_ADV_CENTERED | PV_ADV_UPWIND1), BOUND_CORIOLIS,
 *   NOSLIP, KE_SCHEME = KE_ARAKAWA (default) | KE_SIMPLE_GUDONOV | KE_GUDONOV; CORIOLIS_EN_DIS=False;
 *   no OBC, no Stokes drift; PGF: use_EOS=False, no p_atm, no tides/SAL, GFS_scale=1;
 *   vertvisc: with or without DIRECT_STRESS; no Stokes mixing / fpmix / GL90; optional Ray_u/Ray_v.
 */
#include "orc_common.h"

void orc_pass_var(const mom6x_dims *d, double *a, int stagger, int nk);

/* ------------------------------------------------------------------------------------------ */
/* CorAdCalc                                                                                   */
int orc_CorAdCalc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_coriolis_params *CS,
                  const double *u, const double *v, const double *h, const double *uh, const double *vh,
                  double *CAu, double *CAv) {
  if (CS->Coriolis_Scheme < MOM6X_SADOURNY75_ENERGY || CS->Coriolis_Scheme > MOM6X_AL_BLEND) return MOM6X_EINVAL;
  const int scheme = CS->Coriolis_Scheme;
  const int AH_like = (scheme == MOM6X_ARAKAWA_HSU90 || scheme == MOM6X_ARAKAWA_LAMB81 || scheme == MOM6X_AL_BLEND);
  const int AL_like = (scheme == MOM6X_ARAKAWA_LAMB81 || scheme == MOM6X_AL_BLEND);
  const double C1_24 = 1.0 / 24.0;
  const double eps_vel = 1.0e-10 * 1.0, h_tiny = GV->Angstrom_H;   /* :242-243 */
  const double *IdxCv = GM(G, d, MOM6X_G_IdxCv), *IdyCu = GM(G, d, MOM6X_G_IdyCu);
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk, st = d->pitch;
  const int Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT), *areaT = GM(G, d, MOM6X_G_areaT), *IareaT = GM(G, d, MOM6X_G_IareaT);
  const double *dyCv = GM(G, d, MOM6X_G_dyCv), *dxCu = GM(G, d, MOM6X_G_dxCu);
  const double *mBu = GM(G, d, MOM6X_G_mask2dBu), *IareaBu = GM(G, d, MOM6X_G_IareaBu), *fBu = GM(G, d, MOM6X_G_CoriolisBu);
  const double *IdxCu = GM(G, d, MOM6X_G_IdxCu), *IdyCv = GM(G, d, MOM6X_G_IdyCv);
  const double *areaCu = GM(G, d, MOM6X_G_areaCu), *areaCv = GM(G, d, MOM6X_G_areaCv);
  const double vol_neglect = GV->H_subroundoff * ((1e-4 * 1.0) * (1e-4 * 1.0));
  const double C1_12 = 1.0 / 12.0;
#define NEW2(x) double *x = (double *)calloc(slab, sizeof(double))
  NEW2(Area_h); NEW2(Area_q);
  const double *dy_Cu = GM(G, d, MOM6X_G_dy_Cu), *dx_Cv = GM(G, d, MOM6X_G_dx_Cv);
  /* CoriolisAdv_init :1119, :1158: ROBUST_ENSTRO switches En_Dis off; En_Dis with SADOURNY75_ENERGY switches the bound off */
  const int en_dis = CS->Coriolis_En_Dis && scheme != MOM6X_ROBUST_ENSTRO;
  const int bound_Coriolis = CS->bound_Coriolis && !(en_dis && scheme == MOM6X_SADOURNY75_ENERGY) && scheme != MOM6X_ROBUST_ENSTRO;

  for (int j = Jsq - 1; j <= Jeq + 2; j++) for (int i = Isq - 1; i <= Ieq + 2; i++) {
    size_t x = IX2(d, i, j); Area_h[x] = mT[x] * areaT[x];
  }
  for (int j = Jsq - 1; j <= Jeq + 1; j++) for (int i = Isq - 1; i <= Ieq + 1; i++) {
    size_t x = IX2(d, i, j);
    Area_q[x] = (Area_h[x] + Area_h[x + 1 + st]) + (Area_h[x + 1] + Area_h[x + st]);
  }

  /* layers are independent (the reference: !$OMP parallel do over k, MOM_CoriolisAdv.F90:305): every thread its own 2-d work arrays */
#pragma omp parallel
  {
  NEW2(dvdx); NEW2(dudy); NEW2(hArea_u); NEW2(hArea_v); NEW2(rel_vort); NEW2(abs_vort);
  NEW2(q); NEW2(a); NEW2(b); NEW2(c); NEW2(dd); NEW2(KE); NEW2(KEx); NEW2(KEy);
  NEW2(uh_min); NEW2(uh_max); NEW2(vh_min); NEW2(vh_max); NEW2(Ih_qa); NEW2(ep_u); NEW2(ep_v);
#pragma omp for schedule(static)
  for (int k = 0; k < nz; k++) {
    const double *uk = u + k * slab, *vk = v + k * slab, *hk = h + k * slab, *uhk = uh + k * slab, *vhk = vh + k * slab;
    double *CAuk = CAu + k * slab, *CAvk = CAv + k * slab;
    for (int j = Jsq - 1; j <= Jeq + 1; j++) for (int i = Isq - 1; i <= Ieq + 1; i++) { /* :314-317 */
      size_t x = IX2(d, i, j);
      dvdx[x] = (vk[x + 1] * dyCv[x + 1]) - (vk[x] * dyCv[x]);
      dudy[x] = (uk[x + st] * dxCu[x + st]) - (uk[x] * dxCu[x]);
    }
    for (int j = Jsq - 1; j <= Jeq + 1; j++) for (int i = Isq - 1; i <= Ieq + 2; i++) { /* :319-321 */
      size_t x = IX2(d, i, j);
      hArea_v[x] = 0.5 * ((Area_h[x] * hk[x]) + (Area_h[x + st] * hk[x + st]));
    }
    for (int j = Jsq - 1; j <= Jeq + 2; j++) for (int i = Isq - 1; i <= Ieq + 1; i++) { /* :322-324 */
      size_t x = IX2(d, i, j);
      hArea_u[x] = 0.5 * ((Area_h[x] * hk[x]) + (Area_h[x + 1] * hk[x + 1]));
    }
    for (int j = Jsq - 1; j <= Jeq + 1; j++) for (int i = Isq - 1; i <= Ieq + 1; i++) { /* :470-491 */
      size_t x = IX2(d, i, j);
      if (CS->no_slip) rel_vort[x] = (2.0 - mBu[x]) * (dvdx[x] - dudy[x]) * IareaBu[x];
      else rel_vort[x] = mBu[x] * (dvdx[x] - dudy[x]) * IareaBu[x];
      abs_vort[x] = fBu[x] + rel_vort[x];
      double hArea_q = (hArea_u[x] + hArea_u[x + st]) + (hArea_v[x] + hArea_v[x + 1]);
      double Ih_q = Area_q[x] / (hArea_q + vol_neglect);
      q[x] = abs_vort[x] * Ih_q;
      Ih_qa[x] = Ih_q;
    }
    if (scheme == MOM6X_ARAKAWA_HSU90) { /* :523-533 */
      for (int j = Jsq; j <= Jeq + 1; j++) {
        for (int i = is - 1; i <= Ieq; i++) {
          size_t x = IX2(d, i, j);
          a[x] = (q[x] + (q[x + 1] + q[x - st])) * C1_12;
          dd[x] = ((q[x] + q[x + 1 - st]) + q[x - st]) * C1_12;
        }
        for (int i = Isq; i <= Ieq; i++) {
          size_t x = IX2(d, i, j);
          b[x] = (q[x] + (q[x - 1] + q[x - st])) * C1_12;
          c[x] = ((q[x] + q[x - 1 - st]) + q[x - st]) * C1_12;
        }
      }
    }
    if (scheme == MOM6X_ARAKAWA_LAMB81) {   /* :534-542 */
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
        size_t x = IX2(d, i, j);
        a[x - 1] = (2.0 * (q[x] + q[x - 1 - st]) + (q[x - 1] + q[x - st])) * C1_24;
        dd[x - 1] = ((q[x] + q[x - 1 - st]) + 2.0 * (q[x - 1] + q[x - st])) * C1_24;
        b[x] = ((q[x] + q[x - 1 - st]) + 2.0 * (q[x - 1] + q[x - st])) * C1_24;
        c[x] = (2.0 * (q[x] + q[x - 1 - st]) + (q[x - 1] + q[x - st])) * C1_24;
        ep_u[x] = ((q[x] - q[x - 1 - st]) + (q[x - 1] - q[x - st])) * C1_24;
        ep_v[x] = (-(q[x] - q[x - 1 - st]) + (q[x - 1] - q[x - st])) * C1_24;
      }
    } else if (scheme == MOM6X_AL_BLEND) {   /* :543-588 */
      const double wt_lin_blend = orc_min(1.0, orc_max(CS->wt_lin_blend, 1e-16));   /* CoriolisAdv_init :1139 */
      double Fe_m2 = CS->F_eff_max_blend - 2.0;
      double rat_lin = 1.5 * Fe_m2 / orc_max(wt_lin_blend, 1.0e-16);
      if (CS->F_eff_max_blend <= 2.0) { Fe_m2 = -1.; rat_lin = -1.0; }
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
        size_t x = IX2(d, i, j);
        double min_Ihq = orc_min(orc_min(orc_min(Ih_qa[x - 1 - st], Ih_qa[x - st]), Ih_qa[x - 1]), Ih_qa[x]);
        double max_Ihq = orc_max(orc_max(orc_max(Ih_qa[x - 1 - st], Ih_qa[x - st]), Ih_qa[x - 1]), Ih_qa[x]);
        double rat_m1 = 1.0e15, AL_wt, Sad_wt;
        if (max_Ihq < 1.0e15 * min_Ihq) rat_m1 = max_Ihq / min_Ihq - 1.0;
        if (rat_m1 <= Fe_m2) AL_wt = 1.0;
        else if (rat_m1 < 1.5 * Fe_m2) AL_wt = 3.0 * Fe_m2 / rat_m1 - 2.0;
        else AL_wt = 0.0;
        if (rat_m1 <= 1.5 * Fe_m2) Sad_wt = 0.0;
        else if (rat_m1 <= rat_lin) Sad_wt = 1.0 - (1.5 * Fe_m2) / rat_m1;
        else if (rat_m1 < 2.0 * rat_lin) Sad_wt = 1.0 - (wt_lin_blend / rat_lin) * (rat_m1 - 2.0 * rat_lin);
        else Sad_wt = 1.0;
        a[x - 1] = Sad_wt * 0.25 * q[x - 1] + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q[x - 1] + AL_wt * q[x - st]) + 2.0 * (q[x] + q[x - 1 - st])) * C1_24;
        dd[x - 1] = Sad_wt * 0.25 * q[x - 1 - st] + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q[x - 1 - st] + AL_wt * q[x]) + 2.0 * (q[x - 1] + q[x - st])) * C1_24;
        b[x] = Sad_wt * 0.25 * q[x] + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q[x] + AL_wt * q[x - 1 - st]) + 2.0 * (q[x - 1] + q[x - st])) * C1_24;
        c[x] = Sad_wt * 0.25 * q[x - st] + (1.0 - Sad_wt) * (((2.0 - AL_wt) * q[x - st] + AL_wt * q[x - 1]) + 2.0 * (q[x] + q[x - 1 - st])) * C1_24;
        ep_u[x] = AL_wt * ((q[x] - q[x - 1 - st]) + (q[x - 1] - q[x - st])) * C1_24;
        ep_v[x] = AL_wt * (-(q[x] - q[x - 1 - st]) + (q[x - 1] - q[x - st])) * C1_24;
      }
    }
    if (en_dis) {   /* uh_center, vh_center :326-333 and the bracketing transports :590-635 */
      const double c1 = 1.0 - 1.5 * 0.5, c2 = 1.0 - 0.5, c3 = 2.0, slope = 0.5;
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = is - 1; i <= ie; i++) {
        size_t x = IX2(d, i, j);
        double uhc = 0.5 * ((dy_Cu[x] * 1.0) * uk[x]) * (hk[x] + hk[x + 1]);
        double uhm = uhk[x];
        if (dy_Cu[x] == 0.0) uhc = uhm;
        if (fabs(uhc) < 0.1 * fabs(uhm)) uhm = 10.0 * uhc;
        else if (fabs(uhc) > c1 * fabs(uhm)) {
          if (fabs(uhc) < c2 * fabs(uhm)) uhc = (3.0 * uhc + (1.0 - c2 * 3.0) * uhm);
          else if (fabs(uhc) <= c3 * fabs(uhm)) uhc = uhm;
          else uhc = slope * uhc + (1.0 - c3 * slope) * uhm;
        }
        if (uhc > uhm) { uh_min[x] = uhm; uh_max[x] = uhc; } else { uh_max[x] = uhm; uh_min[x] = uhc; }
      }
      for (int j = js - 1; j <= je; j++) for (int i = Isq; i <= Ieq + 1; i++) {
        size_t x = IX2(d, i, j);
        double vhc = 0.5 * ((dx_Cv[x] * 1.0) * vk[x]) * (hk[x] + hk[x + st]);
        double vhm = vhk[x];
        if (dx_Cv[x] == 0.0) vhc = vhm;
        if (fabs(vhc) < 0.1 * fabs(vhm)) vhm = 10.0 * vhc;
        else if (fabs(vhc) > c1 * fabs(vhm)) {
          if (fabs(vhc) < c2 * fabs(vhm)) vhc = (3.0 * vhc + (1.0 - c2 * 3.0) * vhm);
          else if (fabs(vhc) <= c3 * fabs(vhm)) vhc = vhm;
          else vhc = slope * vhc + (1.0 - c3 * slope) * vhm;
        }
        if (vhc > vhm) { vh_min[x] = vhm; vh_max[x] = vhc; } else { vh_max[x] = vhm; vh_min[x] = vhc; }
      }
    }
    /* gradKE :969-1052 */
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      if (CS->KE_Scheme == MOM6X_KE_ARAKAWA) {
        KE[x] = (((areaCu[x] * (uk[x] * uk[x])) + (areaCu[x - 1] * (uk[x - 1] * uk[x - 1]))) +
                 ((areaCv[x] * (vk[x] * vk[x])) + (areaCv[x - st] * (vk[x - st] * vk[x - st])))) * 0.25 * IareaT[x];
      } else if (CS->KE_Scheme == MOM6X_KE_SIMPLE_GUDONOV) {
        double up = 0.5 * (uk[x - 1] + fabs(uk[x - 1])), up2 = up * up;
        double um = 0.5 * (uk[x] - fabs(uk[x])), um2 = um * um;
        double vp = 0.5 * (vk[x - st] + fabs(vk[x - st])), vp2 = vp * vp;
        double vm = 0.5 * (vk[x] - fabs(vk[x])), vm2 = vm * vm;
        KE[x] = (orc_max(up2, um2) + orc_max(vp2, vm2)) * 0.5;
      } else {
        double up = 0.5 * (uk[x - 1] + fabs(uk[x - 1])), up2a = up * up * areaCu[x - 1];
        double um = 0.5 * (uk[x] - fabs(uk[x])), um2a = um * um * areaCu[x];
        double vp = 0.5 * (vk[x - st] + fabs(vk[x - st])), vp2a = vp * vp * areaCv[x - st];
        double vm = 0.5 * (vk[x] - fabs(vk[x])), vm2a = vm * vm * areaCv[x];
        KE[x] = (orc_max(um2a, up2a) + orc_max(vm2a, vp2a)) * 0.5 * IareaT[x];
      }
    }
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) { size_t x = IX2(d, i, j); KEx[x] = (KE[x + 1] - KE[x]) * IdxCu[x]; }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) { size_t x = IX2(d, i, j); KEy[x] = (KE[x + st] - KE[x]) * IdyCv[x]; }

    /* CAu :664-754 */
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX2(d, i, j);
      double ca;
      if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENERGY && en_dis) {   /* :665-684 energy dissipating biased scheme */
        double temp1, temp2;
        if (q[x] * uk[x] == 0.0) temp1 = q[x] * ((vh_max[x] + vh_max[x + 1]) + (vh_min[x] + vh_min[x + 1])) * 0.5;
        else if (q[x] * uk[x] < 0.0) temp1 = q[x] * (vh_max[x] + vh_max[x + 1]);
        else temp1 = q[x] * (vh_min[x] + vh_min[x + 1]);
        if (q[x - st] * uk[x] == 0.0) temp2 = q[x - st] * ((vh_max[x - st] + vh_max[x + 1 - st]) + (vh_min[x - st] + vh_min[x + 1 - st])) * 0.5;
        else if (q[x - st] * uk[x] < 0.0) temp2 = q[x - st] * (vh_max[x - st] + vh_max[x + 1 - st]);
        else temp2 = q[x - st] * (vh_min[x - st] + vh_min[x + 1 - st]);
        ca = 0.25 * IdxCu[x] * (temp1 + temp2);
      } else if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENERGY)
        ca = 0.25 * ((q[x] * (vhk[x + 1] + vhk[x])) + (q[x - st] * (vhk[x - st] + vhk[x + 1 - st]))) * IdxCu[x];
      else if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENSTRO)
        ca = 0.125 * (IdxCu[x] * (q[x] + q[x - st])) * ((vhk[x + 1] + vhk[x]) + (vhk[x - st] + vhk[x + 1 - st]));
      else if (AH_like)
        ca = (((a[x] * vhk[x + 1]) + (c[x] * vhk[x - st])) + ((b[x] * vhk[x]) + (dd[x] * vhk[x + 1 - st]))) * IdxCu[x];
      else {   /* ROBUST_ENSTRO :687-714 */
        double Heff1 = fabs(vhk[x] * IdxCv[x]) / (eps_vel + fabs(vk[x]));
        Heff1 = orc_max(Heff1, orc_min(hk[x], hk[x + st])); Heff1 = orc_min(Heff1, orc_max(hk[x], hk[x + st]));
        double Heff2 = fabs(vhk[x - st] * IdxCv[x - st]) / (eps_vel + fabs(vk[x - st]));
        Heff2 = orc_max(Heff2, orc_min(hk[x - st], hk[x])); Heff2 = orc_min(Heff2, orc_max(hk[x - st], hk[x]));
        double Heff3 = fabs(vhk[x + 1] * IdxCv[x + 1]) / (eps_vel + fabs(vk[x + 1]));
        Heff3 = orc_max(Heff3, orc_min(hk[x + 1], hk[x + 1 + st])); Heff3 = orc_min(Heff3, orc_max(hk[x + 1], hk[x + 1 + st]));
        double Heff4 = fabs(vhk[x + 1 - st] * IdxCv[x + 1 - st]) / (eps_vel + fabs(vk[x + 1 - st]));
        Heff4 = orc_max(Heff4, orc_min(hk[x + 1 - st], hk[x + 1])); Heff4 = orc_min(Heff4, orc_max(hk[x + 1 - st], hk[x + 1]));
        if (CS->PV_Adv_Scheme == MOM6X_PV_ADV_UPWIND1) {
          double VHeff = ((vhk[x] + vhk[x + 1 - st]) + (vhk[x - st] + vhk[x + 1]));
          double QVHeff = 0.5 * (((abs_vort[x] + abs_vort[x - st]) * VHeff) - ((abs_vort[x] - abs_vort[x - st]) * fabs(VHeff)));
          ca = (QVHeff / (h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3)))) * IdxCu[x];
        } else
          ca = 0.5 * (abs_vort[x] + abs_vort[x - st]) * ((vhk[x] + vhk[x + 1 - st]) + (vhk[x - st] + vhk[x + 1])) /
               (h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdxCu[x];
      }
      if (AL_like) ca = ca + ((ep_u[x] * uhk[x - 1]) - (ep_u[x + 1] * uhk[x + 1])) * IdxCu[x];   /* :716-721 */
      if (bound_Coriolis) {
        double fv1 = abs_vort[x] * vk[x + 1], fv2 = abs_vort[x] * vk[x];
        double fv3 = abs_vort[x - st] * vk[x + 1 - st], fv4 = abs_vort[x - st] * vk[x - st];
        double max_fv = orc_max(orc_max(orc_max(fv1, fv2), fv3), fv4), min_fv = orc_min(orc_min(orc_min(fv1, fv2), fv3), fv4);
        ca = orc_min(ca, max_fv); ca = orc_max(ca, min_fv);
      }
      CAuk[x] = ca - KEx[x];
    }
    /* CAv :775-884 */
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX2(d, i, j);
      double ca;
      if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENERGY && en_dis) {   /* :776-795 */
        double temp1, temp2;
        if (q[x - 1] * vk[x] == 0.0) temp1 = q[x - 1] * ((uh_max[x - 1] + uh_max[x - 1 + st]) + (uh_min[x - 1] + uh_min[x - 1 + st])) * 0.5;
        else if (q[x - 1] * vk[x] > 0.0) temp1 = q[x - 1] * (uh_max[x - 1] + uh_max[x - 1 + st]);
        else temp1 = q[x - 1] * (uh_min[x - 1] + uh_min[x - 1 + st]);
        if (q[x] * vk[x] == 0.0) temp2 = q[x] * ((uh_max[x] + uh_max[x + st]) + (uh_min[x] + uh_min[x + st])) * 0.5;
        else if (q[x] * vk[x] > 0.0) temp2 = q[x] * (uh_max[x] + uh_max[x + st]);
        else temp2 = q[x] * (uh_min[x] + uh_min[x + st]);
        ca = -0.25 * IdyCv[x] * (temp1 + temp2);
      } else if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENERGY)
        ca = -0.25 * ((q[x - 1] * (uhk[x - 1] + uhk[x - 1 + st])) + (q[x] * (uhk[x] + uhk[x + st]))) * IdyCv[x];
      else if (CS->Coriolis_Scheme == MOM6X_SADOURNY75_ENSTRO)
        ca = -0.125 * (IdyCv[x] * (q[x - 1] + q[x])) * ((uhk[x - 1] + uhk[x - 1 + st]) + (uhk[x] + uhk[x + st]));
      else if (AH_like)
        ca = -(((a[x - 1] * uhk[x - 1]) + (c[x + st] * uhk[x + st])) + ((b[x] * uhk[x]) + (dd[x - 1 + st] * uhk[x - 1 + st]))) * IdyCv[x];
      else {   /* ROBUST_ENSTRO :808-838 */
        double Heff1 = fabs(uhk[x] * IdyCu[x]) / (eps_vel + fabs(uk[x]));
        Heff1 = orc_max(Heff1, orc_min(hk[x], hk[x + 1])); Heff1 = orc_min(Heff1, orc_max(hk[x], hk[x + 1]));
        double Heff2 = fabs(uhk[x - 1] * IdyCu[x - 1]) / (eps_vel + fabs(uk[x - 1]));
        Heff2 = orc_max(Heff2, orc_min(hk[x - 1], hk[x])); Heff2 = orc_min(Heff2, orc_max(hk[x - 1], hk[x]));
        double Heff3 = fabs(uhk[x + st] * IdyCu[x + st]) / (eps_vel + fabs(uk[x + st]));
        Heff3 = orc_max(Heff3, orc_min(hk[x + st], hk[x + 1 + st])); Heff3 = orc_min(Heff3, orc_max(hk[x + st], hk[x + 1 + st]));
        double Heff4 = fabs(uhk[x - 1 + st] * IdyCu[x - 1 + st]) / (eps_vel + fabs(uk[x - 1 + st]));
        Heff4 = orc_max(Heff4, orc_min(hk[x - 1 + st], hk[x + st])); Heff4 = orc_min(Heff4, orc_max(hk[x - 1 + st], hk[x + st]));
        if (CS->PV_Adv_Scheme == MOM6X_PV_ADV_UPWIND1) {
          double UHeff = ((uhk[x] + uhk[x - 1 + st]) + (uhk[x - 1] + uhk[x + st]));
          double QUHeff = 0.5 * (((abs_vort[x] + abs_vort[x - 1]) * UHeff) - ((abs_vort[x] - abs_vort[x - 1]) * fabs(UHeff)));
          ca = -(QUHeff / (h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv[x]);
        } else
          ca = -(0.5 * (abs_vort[x] + abs_vort[x - 1]) * ((uhk[x] + uhk[x - 1 + st]) + (uhk[x - 1] + uhk[x + st])) /
                 (h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv[x]);
      }
      if (AL_like) ca = ca + ((ep_v[x] * vhk[x - st]) - (ep_v[x + st] * vhk[x + st])) * IdyCv[x];   /* :840-845 */
      if (bound_Coriolis) {
        double fu1 = -abs_vort[x] * uk[x + st], fu2 = -abs_vort[x] * uk[x];
        double fu3 = -abs_vort[x - 1] * uk[x - 1 + st], fu4 = -abs_vort[x - 1] * uk[x - 1];
        double max_fu = orc_max(orc_max(orc_max(fu1, fu2), fu3), fu4), min_fu = orc_min(orc_min(orc_min(fu1, fu2), fu3), fu4);
        ca = orc_min(ca, max_fu); ca = orc_max(ca, min_fu);
      }
      CAvk[x] = ca - KEy[x];
    }
  }
  double *all[] = { dvdx, dudy, hArea_u, hArea_v, rel_vort, abs_vort, q, a, b, c, dd, KE, KEx, KEy, uh_min, uh_max, vh_min, vh_max, Ih_qa, ep_u, ep_v };
  for (size_t m = 0; m < sizeof(all) / sizeof(all[0]); m++) free(all[m]);
  }   /* omp parallel */
  free(Area_h); free(Area_q);
  return MOM6X_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Equation of state pieces used by the pressure force                                           */
/* Wright (1997) constants: the reduced-range fit of EQN_OF_STATE = "WRIGHT" (MOM_EOS_Wright.F90:23-37) and "WRIGHT_REDUCED"
 * (MOM_EOS_Wright_red.F90:18-35), the full-range fit of "WRIGHT_FULL" (MOM_EOS_Wright_full.F90:18-35).  "WRIGHT" keeps the original
 * parenthesisation ("buggy_Wright"); WRIGHT_FULL and WRIGHT_REDUCED share the corrected expressions. */
typedef struct { double a0, a1, a2, b0, b1, b2, b3, b4, b5, c0, c1, c2, c3, c4, c5; } wright_set;
static const wright_set WC_RED = { 7.057924e-4, 3.480336e-7, -1.112733e-7, 5.790749e8, 3.516535e6, -4.002714e4, 2.084372e2, 5.944068e5,
                                   -9.643486e3, 1.704853e5, 7.904722e2, -7.984422, 5.140652e-2, -2.302158e2, -3.079464 };
static const wright_set WC_FULL = { 7.133718e-4, 2.724670e-7, -1.646582e-7, 5.613770e8, 3.600337e6, -3.727194e4, 1.660557e2, 6.844158e5,
                                    -8.389457e3, 1.609893e5, 8.427815e2, -6.931554, 3.869318e-2, -1.664201e2, -2.765195 };
static const wright_set *wright_of(int form) { return form == MOM6X_EOS_WRIGHT_FULL ? &WC_FULL : &WC_RED; }

static void wright_coefs(int form, double T, double S, double *al0, double *p0, double *lambda) {
  const wright_set *W = wright_of(form);
  if (form == MOM6X_EOS_WRIGHT) {   /* MOM_EOS_Wright.F90:91-93, :555-557 */
    *al0 = (W->a0 + W->a1 * T) + W->a2 * S;
    *p0 = (W->b0 + W->b4 * S) + T * (W->b1 + T * ((W->b2 + W->b3 * T)) + W->b5 * S);
    *lambda = (W->c0 + W->c4 * S) + T * (W->c1 + T * ((W->c2 + W->c3 * T)) + W->c5 * S);
  } else {                          /* MOM_EOS_Wright_full.F90:84-86, :550-552 (and _red) */
    *al0 = W->a0 + (W->a1 * T + W->a2 * S);
    *p0 = W->b0 + (W->b4 * S + T * (W->b1 + (T * (W->b2 + W->b3 * T) + W->b5 * S)));
    *lambda = W->c0 + (W->c4 * S + T * (W->c1 + (T * (W->c2 + W->c3 * T) + W->c5 * S)));
  }
}

/* EQN_OF_STATE = "UNESCO" (MOM_EOS_UNESCO.F90): UNESCO (1981) as refit by Jackett and McDougall (1995).  Rab: the S^a T^b term of the
 * one-atmosphere density (6 = power 1.5), Sabc: the S^a T^b p^c term of the secant bulk modulus; pressure in bar. */
static const double U_R00 = 999.842594, U_R01 = 6.793952e-2, U_R02 = -9.095290e-3, U_R03 = 1.001685e-4, U_R04 = -1.120083e-6, U_R05 = 6.536332e-9;
static const double U_R10 = 0.824493, U_R11 = -4.0899e-3, U_R12 = 7.6438e-5, U_R13 = -8.2467e-7, U_R14 = 5.3875e-9;
static const double U_R60 = -5.72466e-3, U_R61 = 1.0227e-4, U_R62 = -1.6546e-6, U_R20 = 4.8314e-4;
static const double U_S000 = 1.965933e4, U_S010 = 1.444304e2, U_S020 = -1.706103, U_S030 = 9.648704e-3, U_S040 = -4.190253e-5;
static const double U_S100 = 52.84855, U_S110 = -3.101089e-1, U_S120 = 6.283263e-3, U_S130 = -5.084188e-5;
static const double U_S600 = 3.886640e-1, U_S610 = 9.085835e-3, U_S620 = -4.619924e-4;
static const double U_S001 = 3.186519, U_S011 = 2.212276e-2, U_S021 = -2.984642e-4, U_S031 = 1.956415e-6;
static const double U_S101 = 6.704388e-3, U_S111 = -1.847318e-4, U_S121 = 2.059331e-7, U_S601 = 1.480266e-4;
static const double U_S002 = 2.102898e-4, U_S012 = -1.202016e-5, U_S022 = 1.394680e-7, U_S102 = -2.040237e-6, U_S112 = 6.128773e-8, U_S122 = 6.207323e-10;

static double unesco_sig0(double t1, double s1, double s12) {   /* :114-116 */
  return (t1 * (U_R01 + t1 * (U_R02 + t1 * (U_R03 + t1 * (U_R04 + t1 * U_R05)))) +
          s1 * ((U_R10 + t1 * (U_R11 + t1 * (U_R12 + t1 * (U_R13 + t1 * U_R14)))) + (s12 * (U_R60 + t1 * (U_R61 + t1 * U_R62)) + s1 * U_R20)));
}
static double unesco_ks(double t1, double s1, double s12, double p1) {   /* :120-124 */
  return (U_S000 + (t1 * (U_S010 + t1 * (U_S020 + t1 * (U_S030 + t1 * U_S040))) +
                    s1 * ((U_S100 + t1 * (U_S110 + t1 * (U_S120 + t1 * U_S130))) + s12 * (U_S600 + t1 * (U_S610 + t1 * U_S620))))) +
         p1 * ((U_S001 + (t1 * (U_S011 + t1 * (U_S021 + t1 * U_S031)) + s1 * ((U_S101 + t1 * (U_S111 + t1 * U_S121)) + s12 * U_S601))) +
               p1 * (U_S002 + (t1 * (U_S012 + t1 * U_S022) + s1 * (U_S102 + t1 * (U_S112 + t1 * U_S122)))));
}
static double unesco_density(double T, double S, double pressure) {   /* density_elem_UNESCO :95-128 */
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = orc_max(S, 0.0), s12 = sqrt(s1);
  const double rho0 = U_R00 + unesco_sig0(t1, s1, s12), ks = unesco_ks(t1, s1, s12, p1);
  return rho0 * ks / (ks - p1);
}
static double unesco_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :133-167 */
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = orc_max(S, 0.0), s12 = sqrt(s1);
  const double sig0 = unesco_sig0(t1, s1, s12), ks = unesco_ks(t1, s1, s12, p1);
  return ((U_R00 - rho_ref) * ks + (sig0 * ks + p1 * rho_ref)) / (ks - p1);
}
static void unesco_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :244-297 */
  const double p1 = pressure * 1.0e-5, t1 = T, s1 = orc_max(S, 0.0), s12 = sqrt(s1);
  const double rho0 = U_R00 + unesco_sig0(t1, s1, s12);
  const double drho0_dT = U_R01 + (t1 * (2.0 * U_R02 + t1 * (3.0 * U_R03 + t1 * (4.0 * U_R04 + t1 * (5.0 * U_R05)))) +
                                   s1 * (U_R11 + (t1 * (2.0 * U_R12 + t1 * (3.0 * U_R13 + t1 * (4.0 * U_R14))) + s12 * (U_R61 + t1 * (2.0 * U_R62)))));
  const double drho0_dS = U_R10 + (t1 * (U_R11 + t1 * (U_R12 + t1 * (U_R13 + t1 * U_R14))) +
                                   (1.5 * (s12 * (U_R60 + t1 * (U_R61 + t1 * U_R62))) + s1 * (2.0 * U_R20)));
  const double ks = unesco_ks(t1, s1, s12, p1);
  const double dks_dT = (U_S010 + (t1 * (2.0 * U_S020 + t1 * (3.0 * U_S030 + t1 * (4.0 * U_S040))) +
                                   s1 * ((U_S110 + t1 * (2.0 * U_S120 + t1 * (3.0 * U_S130))) + s12 * (U_S610 + t1 * (2.0 * U_S620))))) +
                        p1 * (((U_S011 + t1 * (2.0 * U_S021 + t1 * (3.0 * U_S031))) + s1 * (U_S111 + t1 * (2.0 * U_S121))) +
                              p1 * (U_S012 + t1 * (2.0 * U_S022) + s1 * (U_S112 + t1 * (2.0 * U_S122))));
  const double dks_dS = (U_S100 + (t1 * (U_S110 + t1 * (U_S120 + t1 * U_S130)) + 1.5 * (s12 * (U_S600 + t1 * (U_S610 + t1 * U_S620))))) +
                        p1 * ((U_S101 + t1 * (U_S111 + t1 * U_S121) + s12 * (1.5 * U_S601)) + p1 * (U_S102 + t1 * (U_S112 + t1 * U_S122)));
  const double I_denom = 1.0 / (ks - p1);
  *drho_dT = (ks * drho0_dT - dks_dT * ((rho0 * p1) * I_denom)) * I_denom;
  *drho_dS = (ks * drho0_dS - dks_dS * ((rho0 * p1) * I_denom)) * I_denom;
}

#define RQ_FN static

/* EQN_OF_STATE = "ROQUET_RHO" (alias "NEMO"): the polynomial of Roquet et al. (2015), MOM_EOS_Roquet_rho.F90.  EOSabc: the zs^a T^b p^c
 * term (zs = sqrt((S + 32) * 0.875 / 35.16504)), R0c: the reference profile's p^(c+1) term, ALP / BET: the T / zs derivatives'
 * coefficients -- the reference forms them as named constants from the published numbers (:12-155), and so do these macros, with
 * integer powers by repeated squaring as the compiler folds them. */
#define RQ_POW2(x) ((x) * (x))
#define RQ_POW3(x) ((x) * ((x) * (x)))
#define RQ_POW4(x) (((x) * (x)) * ((x) * (x)))
#define RQ_POW5(x) ((x) * (((x) * (x)) * ((x) * (x))))
#define RQ_POW6(x) (((x) * (x)) * (((x) * (x)) * ((x) * (x))))
#define RQ_Pa2kb (1.e-8)
#define RQ_rdeltaS (32.)
#define RQ_r1_S0 (0.875/35.16504)
#define RQ_I_Ts (0.025)
#define RQ_R00 (4.6494977072e+01*RQ_Pa2kb)
#define RQ_R01 (-5.2099962525*RQ_POW2(RQ_Pa2kb))
#define RQ_R02 (2.2601900708e-01*RQ_POW3(RQ_Pa2kb))
#define RQ_R03 (6.4326772569e-02*RQ_POW4(RQ_Pa2kb))
#define RQ_R04 (1.5616995503e-02*RQ_POW5(RQ_Pa2kb))
#define RQ_R05 (-1.7243708991e-03*RQ_POW6(RQ_Pa2kb))
#define RQ_EOS000 (8.0189615746e+02)
#define RQ_EOS100 (8.6672408165e+02)
#define RQ_EOS200 (-1.7864682637e+03)
#define RQ_EOS300 (2.0375295546e+03)
#define RQ_EOS400 (-1.2849161071e+03)
#define RQ_EOS500 (4.3227585684e+02)
#define RQ_EOS600 (-6.0579916612e+01)
#define RQ_EOS010 (2.6010145068e+01*RQ_I_Ts)
#define RQ_EOS110 (-6.5281885265e+01*RQ_I_Ts)
#define RQ_EOS210 (8.1770425108e+01*RQ_I_Ts)
#define RQ_EOS310 (-5.6888046321e+01*RQ_I_Ts)
#define RQ_EOS410 (1.7681814114e+01*RQ_I_Ts)
#define RQ_EOS510 (-1.9193502195*RQ_I_Ts)
#define RQ_EOS020 (-3.7074170417e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS120 (6.1548258127e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS220 (-6.0362551501e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS320 (2.9130021253e+01*RQ_POW2(RQ_I_Ts))
#define RQ_EOS420 (-5.4723692739*RQ_POW2(RQ_I_Ts))
#define RQ_EOS030 (2.1661789529e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS130 (-3.3449108469e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS230 (1.9717078466e+01*RQ_POW3(RQ_I_Ts))
#define RQ_EOS330 (-3.1742946532*RQ_POW3(RQ_I_Ts))
#define RQ_EOS040 (-8.3627885467*RQ_POW4(RQ_I_Ts))
#define RQ_EOS140 (1.1311538584e+01*RQ_POW4(RQ_I_Ts))
#define RQ_EOS240 (-5.3563304045*RQ_POW4(RQ_I_Ts))
#define RQ_EOS050 (5.4048723791e-01*RQ_POW5(RQ_I_Ts))
#define RQ_EOS150 (4.8169980163e-01*RQ_POW5(RQ_I_Ts))
#define RQ_EOS060 (-1.9083568888e-01*RQ_POW6(RQ_I_Ts))
#define RQ_EOS001 (1.9681925209e+01*RQ_Pa2kb)
#define RQ_EOS101 (-4.2549998214e+01*RQ_Pa2kb)
#define RQ_EOS201 (5.0774768218e+01*RQ_Pa2kb)
#define RQ_EOS301 (-3.0938076334e+01*RQ_Pa2kb)
#define RQ_EOS401 (6.6051753097*RQ_Pa2kb)
#define RQ_EOS011 (-1.3336301113e+01*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS111 (-4.4870114575*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS211 (5.0042598061*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS311 (-6.5399043664e-01*(RQ_I_Ts*RQ_Pa2kb))
#define RQ_EOS021 (6.7080479603*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS121 (3.5063081279*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS221 (-1.8795372996*(RQ_POW2(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS031 (-2.4649669534*(RQ_POW3(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS131 (-5.5077101279e-01*(RQ_POW3(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS041 (5.5927935970e-01*(RQ_POW4(RQ_I_Ts)*RQ_Pa2kb))
#define RQ_EOS002 (2.0660924175*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS102 (-4.9527603989*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS202 (2.5019633244*RQ_POW2(RQ_Pa2kb))
#define RQ_EOS012 (2.0564311499*(RQ_I_Ts*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS112 (-2.1311365518e-01*(RQ_I_Ts*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS022 (-1.2419983026*(RQ_POW2(RQ_I_Ts)*RQ_POW2(RQ_Pa2kb)))
#define RQ_EOS003 (-2.3342758797e-02*RQ_POW3(RQ_Pa2kb))
#define RQ_EOS103 (-1.8507636718e-02*RQ_POW3(RQ_Pa2kb))
#define RQ_EOS013 (3.7969820455e-01*(RQ_I_Ts*RQ_POW3(RQ_Pa2kb)))
#define RQ_ALP000 (RQ_EOS010)
#define RQ_ALP100 (RQ_EOS110)
#define RQ_ALP200 (RQ_EOS210)
#define RQ_ALP300 (RQ_EOS310)
#define RQ_ALP400 (RQ_EOS410)
#define RQ_ALP500 (RQ_EOS510)
#define RQ_ALP010 (2.*RQ_EOS020)
#define RQ_ALP110 (2.*RQ_EOS120)
#define RQ_ALP210 (2.*RQ_EOS220)
#define RQ_ALP310 (2.*RQ_EOS320)
#define RQ_ALP410 (2.*RQ_EOS420)
#define RQ_ALP020 (3.*RQ_EOS030)
#define RQ_ALP120 (3.*RQ_EOS130)
#define RQ_ALP220 (3.*RQ_EOS230)
#define RQ_ALP320 (3.*RQ_EOS330)
#define RQ_ALP030 (4.*RQ_EOS040)
#define RQ_ALP130 (4.*RQ_EOS140)
#define RQ_ALP230 (4.*RQ_EOS240)
#define RQ_ALP040 (5.*RQ_EOS050)
#define RQ_ALP140 (5.*RQ_EOS150)
#define RQ_ALP050 (6.*RQ_EOS060)
#define RQ_ALP001 (RQ_EOS011)
#define RQ_ALP101 (RQ_EOS111)
#define RQ_ALP201 (RQ_EOS211)
#define RQ_ALP301 (RQ_EOS311)
#define RQ_ALP011 (2.*RQ_EOS021)
#define RQ_ALP111 (2.*RQ_EOS121)
#define RQ_ALP211 (2.*RQ_EOS221)
#define RQ_ALP021 (3.*RQ_EOS031)
#define RQ_ALP121 (3.*RQ_EOS131)
#define RQ_ALP031 (4.*RQ_EOS041)
#define RQ_ALP002 (RQ_EOS012)
#define RQ_ALP102 (RQ_EOS112)
#define RQ_ALP012 (2.*RQ_EOS022)
#define RQ_ALP003 (RQ_EOS013)
#define RQ_BET000 (0.5*RQ_EOS100*RQ_r1_S0)
#define RQ_BET100 (RQ_EOS200*RQ_r1_S0)
#define RQ_BET200 (1.5*RQ_EOS300*RQ_r1_S0)
#define RQ_BET300 (2.0*RQ_EOS400*RQ_r1_S0)
#define RQ_BET400 (2.5*RQ_EOS500*RQ_r1_S0)
#define RQ_BET500 (3.0*RQ_EOS600*RQ_r1_S0)
#define RQ_BET010 (0.5*RQ_EOS110*RQ_r1_S0)
#define RQ_BET110 (RQ_EOS210*RQ_r1_S0)
#define RQ_BET210 (1.5*RQ_EOS310*RQ_r1_S0)
#define RQ_BET310 (2.0*RQ_EOS410*RQ_r1_S0)
#define RQ_BET410 (2.5*RQ_EOS510*RQ_r1_S0)
#define RQ_BET020 (0.5*RQ_EOS120*RQ_r1_S0)
#define RQ_BET120 (RQ_EOS220*RQ_r1_S0)
#define RQ_BET220 (1.5*RQ_EOS320*RQ_r1_S0)
#define RQ_BET320 (2.0*RQ_EOS420*RQ_r1_S0)
#define RQ_BET030 (0.5*RQ_EOS130*RQ_r1_S0)
#define RQ_BET130 (RQ_EOS230*RQ_r1_S0)
#define RQ_BET230 (1.5*RQ_EOS330*RQ_r1_S0)
#define RQ_BET040 (0.5*RQ_EOS140*RQ_r1_S0)
#define RQ_BET140 (RQ_EOS240*RQ_r1_S0)
#define RQ_BET050 (0.5*RQ_EOS150*RQ_r1_S0)
#define RQ_BET001 (0.5*RQ_EOS101*RQ_r1_S0)
#define RQ_BET101 (RQ_EOS201*RQ_r1_S0)
#define RQ_BET201 (1.5*RQ_EOS301*RQ_r1_S0)
#define RQ_BET301 (2.0*RQ_EOS401*RQ_r1_S0)
#define RQ_BET011 (0.5*RQ_EOS111*RQ_r1_S0)
#define RQ_BET111 (RQ_EOS211*RQ_r1_S0)
#define RQ_BET211 (1.5*RQ_EOS311*RQ_r1_S0)
#define RQ_BET021 (0.5*RQ_EOS121*RQ_r1_S0)
#define RQ_BET121 (RQ_EOS221*RQ_r1_S0)
#define RQ_BET031 (0.5*RQ_EOS131*RQ_r1_S0)
#define RQ_BET002 (0.5*RQ_EOS102*RQ_r1_S0)
#define RQ_BET102 (RQ_EOS202*RQ_r1_S0)
#define RQ_BET012 (0.5*RQ_EOS112*RQ_r1_S0)
#define RQ_BET003 (0.5*RQ_EOS103*RQ_r1_S0)

RQ_FN void roquet_parts(double T, double S, double pressure, double *zs_out, double *rhoTS0, double *rhoTS1, double *rhoTS2,
                        double *rhoTS3, double *rho0S0, double *rho00p) {   /* :216-241 */
  const double zt = T, zs = sqrt(fabs(S + RQ_rdeltaS) * RQ_r1_S0), zp = pressure;
  *rhoTS3 = RQ_EOS003 + (zs * RQ_EOS103 + zt * RQ_EOS013);
  *rhoTS2 = RQ_EOS002 + (zs * (RQ_EOS102 + zs * RQ_EOS202) + zt * (RQ_EOS012 + (zs * RQ_EOS112 + zt * RQ_EOS022)));
  *rhoTS1 = RQ_EOS001 + (zs * (RQ_EOS101 + zs * (RQ_EOS201 + zs * (RQ_EOS301 + zs * RQ_EOS401))) +
                         zt * (RQ_EOS011 + (zs * (RQ_EOS111 + zs * (RQ_EOS211 + zs * RQ_EOS311)) +
                                            zt * (RQ_EOS021 + (zs * (RQ_EOS121 + zs * RQ_EOS221) +
                                                               zt * (RQ_EOS031 + (zs * RQ_EOS131 + zt * RQ_EOS041)))))));
  *rhoTS0 = zt * (RQ_EOS010 +
                  (zs * (RQ_EOS110 + zs * (RQ_EOS210 + zs * (RQ_EOS310 + zs * (RQ_EOS410 + zs * RQ_EOS510)))) +
                   zt * (RQ_EOS020 + (zs * (RQ_EOS120 + zs * (RQ_EOS220 + zs * (RQ_EOS320 + zs * RQ_EOS420))) +
                                      zt * (RQ_EOS030 + (zs * (RQ_EOS130 + zs * (RQ_EOS230 + zs * RQ_EOS330)) +
                                                         zt * (RQ_EOS040 + (zs * (RQ_EOS140 + zs * RQ_EOS240) +
                                                                            zt * (RQ_EOS050 + (zs * RQ_EOS150 + zt * RQ_EOS060))))))))));
  *rho0S0 = RQ_EOS000 + zs * (RQ_EOS100 + zs * (RQ_EOS200 + zs * (RQ_EOS300 + zs * (RQ_EOS400 + zs * (RQ_EOS500 + zs * RQ_EOS600)))));
  *rho00p = zp * (RQ_R00 + zp * (RQ_R01 + zp * (RQ_R02 + zp * (RQ_R03 + zp * (RQ_R04 + zp * RQ_R05)))));
  *zs_out = zs;
}
RQ_FN double roquet_density(double T, double S, double pressure) {   /* density_elem_Roquet_rho :192-243 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  const double rhoTS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return rhoTS + p0;
}
RQ_FN double roquet_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :248-306 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  s0 = s0 - rho_ref;
  const double rhoTS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return rhoTS + p0;
}
RQ_FN void roquet_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :340-411 */
  const double zt = T, zs = sqrt(fabs(S + RQ_rdeltaS) * RQ_r1_S0), zp = pressure;
  const double dRdzt3 = RQ_ALP003;
  const double dRdzt2 = RQ_ALP002 + (zs * RQ_ALP102 + zt * RQ_ALP012);
  const double dRdzt1 = RQ_ALP001 + (zs * (RQ_ALP101 + zs * (RQ_ALP201 + zs * RQ_ALP301)) +
                                     zt * (RQ_ALP011 + (zs * (RQ_ALP111 + zs * RQ_ALP211) + zt * (RQ_ALP021 + (zs * RQ_ALP121 + zt * RQ_ALP031)))));
  const double dRdzt0 = RQ_ALP000 + (zs * (RQ_ALP100 + zs * (RQ_ALP200 + zs * (RQ_ALP300 + zs * (RQ_ALP400 + zs * RQ_ALP500)))) +
                                     zt * (RQ_ALP010 + (zs * (RQ_ALP110 + zs * (RQ_ALP210 + zs * (RQ_ALP310 + zs * RQ_ALP410))) +
                                                        zt * (RQ_ALP020 + (zs * (RQ_ALP120 + zs * (RQ_ALP220 + zs * RQ_ALP320)) +
                                                                           zt * (RQ_ALP030 + (zt * (RQ_ALP040 + (zs * RQ_ALP140 + zt * RQ_ALP050)) +
                                                                                              zs * (RQ_ALP130 + zs * RQ_ALP230))))))));
  *drho_dT = dRdzt0 + zp * (dRdzt1 + zp * (dRdzt2 + zp * dRdzt3));
  const double dRdzs3 = RQ_BET003;
  const double dRdzs2 = RQ_BET002 + (zs * RQ_BET102 + zt * RQ_BET012);
  const double dRdzs1 = RQ_BET001 + (zs * (RQ_BET101 + zs * (RQ_BET201 + zs * RQ_BET301)) +
                                     zt * (RQ_BET011 + (zs * (RQ_BET111 + zs * RQ_BET211) + zt * (RQ_BET021 + (zs * RQ_BET121 + zt * RQ_BET031)))));
  const double dRdzs0 = RQ_BET000 + (zs * (RQ_BET100 + zs * (RQ_BET200 + zs * (RQ_BET300 + zs * (RQ_BET400 + zs * RQ_BET500)))) +
                                     zt * (RQ_BET010 + (zs * (RQ_BET110 + zs * (RQ_BET210 + zs * (RQ_BET310 + zs * RQ_BET410))) +
                                                        zt * (RQ_BET020 + (zs * (RQ_BET120 + zs * (RQ_BET220 + zs * RQ_BET320)) +
                                                                           zt * (RQ_BET030 + (zt * (RQ_BET040 + (zs * RQ_BET140 + zt * RQ_BET050)) +
                                                                                              zs * (RQ_BET130 + zs * RQ_BET230))))))));
  *drho_dS = (dRdzs0 + zp * (dRdzs1 + zp * (dRdzs2 + zp * dRdzs3))) / zs;
}

/* EQN_OF_STATE = "ROQUET_SPV": the specific-volume polynomial of Roquet et al. (2015), MOM_EOS_Roquet_SpV.F90 -- the same structure
 * as ROQUET_RHO with SPVabc / V0c / ALP / BET (zs = sqrt((S + 24) * 0.875 / 35.16504)); density = 1 / spec_vol. */
#define RS_Pa2kb (1.e-8)
#define RS_rdeltaS (24.)
#define RS_r1_S0 (0.875/35.16504)
#define RS_I_Ts (0.025)
#define RS_V00 (-4.4015007269e-05*RS_Pa2kb)
#define RS_V01 (6.9232335784e-06*RQ_POW2(RS_Pa2kb))
#define RS_V02 (-7.5004675975e-07*RQ_POW3(RS_Pa2kb))
#define RS_V03 (1.7009109288e-08*RQ_POW4(RS_Pa2kb))
#define RS_V04 (-1.6884162004e-08*RQ_POW5(RS_Pa2kb))
#define RS_V05 (1.9613503930e-09*RQ_POW6(RS_Pa2kb))
#define RS_SPV000 (1.0772899069e-03)
#define RS_SPV100 (-3.1263658781e-04)
#define RS_SPV200 (6.7615860683e-04)
#define RS_SPV300 (-8.6127884515e-04)
#define RS_SPV400 (5.9010812596e-04)
#define RS_SPV500 (-2.1503943538e-04)
#define RS_SPV600 (3.2678954455e-05)
#define RS_SPV010 (-1.4949652640e-05*RS_I_Ts)
#define RS_SPV110 (3.1866349188e-05*RS_I_Ts)
#define RS_SPV210 (-3.8070687610e-05*RS_I_Ts)
#define RS_SPV310 (2.9818473563e-05*RS_I_Ts)
#define RS_SPV410 (-1.0011321965e-05*RS_I_Ts)
#define RS_SPV510 (1.0751931163e-06*RS_I_Ts)
#define RS_SPV020 (2.7546851539e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV120 (-3.6597334199e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV220 (3.4489154625e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV320 (-1.7663254122e-05*RQ_POW2(RS_I_Ts))
#define RS_SPV420 (3.5965131935e-06*RQ_POW2(RS_I_Ts))
#define RS_SPV030 (-1.6506828994e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV130 (2.4412359055e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV230 (-1.4606740723e-05*RQ_POW3(RS_I_Ts))
#define RS_SPV330 (2.3293406656e-06*RQ_POW3(RS_I_Ts))
#define RS_SPV040 (6.7896174634e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV140 (-8.7951832993e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV240 (4.4249040774e-06*RQ_POW4(RS_I_Ts))
#define RS_SPV050 (-7.2535743349e-07*RQ_POW5(RS_I_Ts))
#define RS_SPV150 (-3.4680559205e-07*RQ_POW5(RS_I_Ts))
#define RS_SPV060 (1.9041365570e-07*RQ_POW6(RS_I_Ts))
#define RS_SPV001 (-1.6889436589e-05*RS_Pa2kb)
#define RS_SPV101 (2.1106556158e-05*RS_Pa2kb)
#define RS_SPV201 (-2.1322804368e-05*RS_Pa2kb)
#define RS_SPV301 (1.7347655458e-05*RS_Pa2kb)
#define RS_SPV401 (-4.3209400767e-06*RS_Pa2kb)
#define RS_SPV011 (1.5355844621e-05*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV111 (2.0914122241e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV211 (-5.7751479725e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV311 (1.0767234341e-06*(RS_I_Ts*RS_Pa2kb))
#define RS_SPV021 (-9.6659393016e-06*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV121 (-7.0686982208e-07*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV221 (1.4488066593e-06*(RQ_POW2(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV031 (3.1134283336e-06*(RQ_POW3(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV131 (7.9562529879e-08*(RQ_POW3(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV041 (-5.6590253863e-07*(RQ_POW4(RS_I_Ts)*RS_Pa2kb))
#define RS_SPV002 (1.0500241168e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV102 (1.9600661704e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV202 (-2.1666693382e-06*RQ_POW2(RS_Pa2kb))
#define RS_SPV012 (-3.8541359685e-06*(RS_I_Ts*RQ_POW2(RS_Pa2kb)))
#define RS_SPV112 (1.0157632247e-06*(RS_I_Ts*RQ_POW2(RS_Pa2kb)))
#define RS_SPV022 (1.7178343158e-06*(RQ_POW2(RS_I_Ts)*RQ_POW2(RS_Pa2kb)))
#define RS_SPV003 (-4.1503454190e-07*RQ_POW3(RS_Pa2kb))
#define RS_SPV103 (3.5627020989e-07*RQ_POW3(RS_Pa2kb))
#define RS_SPV013 (-1.1293871415e-07*(RS_I_Ts*RQ_POW3(RS_Pa2kb)))
#define RS_ALP000 (RS_SPV010)
#define RS_ALP100 (RS_SPV110)
#define RS_ALP200 (RS_SPV210)
#define RS_ALP300 (RS_SPV310)
#define RS_ALP400 (RS_SPV410)
#define RS_ALP500 (RS_SPV510)
#define RS_ALP010 (2.*RS_SPV020)
#define RS_ALP110 (2.*RS_SPV120)
#define RS_ALP210 (2.*RS_SPV220)
#define RS_ALP310 (2.*RS_SPV320)
#define RS_ALP410 (2.*RS_SPV420)
#define RS_ALP020 (3.*RS_SPV030)
#define RS_ALP120 (3.*RS_SPV130)
#define RS_ALP220 (3.*RS_SPV230)
#define RS_ALP320 (3.*RS_SPV330)
#define RS_ALP030 (4.*RS_SPV040)
#define RS_ALP130 (4.*RS_SPV140)
#define RS_ALP230 (4.*RS_SPV240)
#define RS_ALP040 (5.*RS_SPV050)
#define RS_ALP140 (5.*RS_SPV150)
#define RS_ALP050 (6.*RS_SPV060)
#define RS_ALP001 (RS_SPV011)
#define RS_ALP101 (RS_SPV111)
#define RS_ALP201 (RS_SPV211)
#define RS_ALP301 (RS_SPV311)
#define RS_ALP011 (2.*RS_SPV021)
#define RS_ALP111 (2.*RS_SPV121)
#define RS_ALP211 (2.*RS_SPV221)
#define RS_ALP021 (3.*RS_SPV031)
#define RS_ALP121 (3.*RS_SPV131)
#define RS_ALP031 (4.*RS_SPV041)
#define RS_ALP002 (RS_SPV012)
#define RS_ALP102 (RS_SPV112)
#define RS_ALP012 (2.*RS_SPV022)
#define RS_ALP003 (RS_SPV013)
#define RS_BET000 (0.5*RS_SPV100*RS_r1_S0)
#define RS_BET100 (RS_SPV200*RS_r1_S0)
#define RS_BET200 (1.5*RS_SPV300*RS_r1_S0)
#define RS_BET300 (2.0*RS_SPV400*RS_r1_S0)
#define RS_BET400 (2.5*RS_SPV500*RS_r1_S0)
#define RS_BET500 (3.0*RS_SPV600*RS_r1_S0)
#define RS_BET010 (0.5*RS_SPV110*RS_r1_S0)
#define RS_BET110 (RS_SPV210*RS_r1_S0)
#define RS_BET210 (1.5*RS_SPV310*RS_r1_S0)
#define RS_BET310 (2.0*RS_SPV410*RS_r1_S0)
#define RS_BET410 (2.5*RS_SPV510*RS_r1_S0)
#define RS_BET020 (0.5*RS_SPV120*RS_r1_S0)
#define RS_BET120 (RS_SPV220*RS_r1_S0)
#define RS_BET220 (1.5*RS_SPV320*RS_r1_S0)
#define RS_BET320 (2.0*RS_SPV420*RS_r1_S0)
#define RS_BET030 (0.5*RS_SPV130*RS_r1_S0)
#define RS_BET130 (RS_SPV230*RS_r1_S0)
#define RS_BET230 (1.5*RS_SPV330*RS_r1_S0)
#define RS_BET040 (0.5*RS_SPV140*RS_r1_S0)
#define RS_BET140 (RS_SPV240*RS_r1_S0)
#define RS_BET050 (0.5*RS_SPV150*RS_r1_S0)
#define RS_BET001 (0.5*RS_SPV101*RS_r1_S0)
#define RS_BET101 (RS_SPV201*RS_r1_S0)
#define RS_BET201 (1.5*RS_SPV301*RS_r1_S0)
#define RS_BET301 (2.0*RS_SPV401*RS_r1_S0)
#define RS_BET011 (0.5*RS_SPV111*RS_r1_S0)
#define RS_BET111 (RS_SPV211*RS_r1_S0)
#define RS_BET211 (1.5*RS_SPV311*RS_r1_S0)
#define RS_BET021 (0.5*RS_SPV121*RS_r1_S0)
#define RS_BET121 (RS_SPV221*RS_r1_S0)
#define RS_BET031 (0.5*RS_SPV131*RS_r1_S0)
#define RS_BET002 (0.5*RS_SPV102*RS_r1_S0)
#define RS_BET102 (RS_SPV202*RS_r1_S0)
#define RS_BET012 (0.5*RS_SPV112*RS_r1_S0)
#define RS_BET003 (0.5*RS_SPV103*RS_r1_S0)
RQ_FN void roquet_spv_parts(double T, double S, double pressure, double *zs_out, double *svTS0, double *svTS1, double *svTS2,
                        double *svTS3, double *sv0S0, double *sv00p) {   /* :216-241 */
  const double zt = T, zs = sqrt(fabs(S + RS_rdeltaS) * RS_r1_S0), zp = pressure;
  *svTS3 = RS_SPV003 + (zs * RS_SPV103 + zt * RS_SPV013);
  *svTS2 = RS_SPV002 + (zs * (RS_SPV102 + zs * RS_SPV202) + zt * (RS_SPV012 + (zs * RS_SPV112 + zt * RS_SPV022)));
  *svTS1 = RS_SPV001 + (zs * (RS_SPV101 + zs * (RS_SPV201 + zs * (RS_SPV301 + zs * RS_SPV401))) +
                         zt * (RS_SPV011 + (zs * (RS_SPV111 + zs * (RS_SPV211 + zs * RS_SPV311)) +
                                            zt * (RS_SPV021 + (zs * (RS_SPV121 + zs * RS_SPV221) +
                                                               zt * (RS_SPV031 + (zs * RS_SPV131 + zt * RS_SPV041)))))));
  *svTS0 = zt * (RS_SPV010 +
                  (zs * (RS_SPV110 + zs * (RS_SPV210 + zs * (RS_SPV310 + zs * (RS_SPV410 + zs * RS_SPV510)))) +
                   zt * (RS_SPV020 + (zs * (RS_SPV120 + zs * (RS_SPV220 + zs * (RS_SPV320 + zs * RS_SPV420))) +
                                      zt * (RS_SPV030 + (zs * (RS_SPV130 + zs * (RS_SPV230 + zs * RS_SPV330)) +
                                                         zt * (RS_SPV040 + (zs * (RS_SPV140 + zs * RS_SPV240) +
                                                                            zt * (RS_SPV050 + (zs * RS_SPV150 + zt * RS_SPV060))))))))));
  *sv0S0 = RS_SPV000 + zs * (RS_SPV100 + zs * (RS_SPV200 + zs * (RS_SPV300 + zs * (RS_SPV400 + zs * (RS_SPV500 + zs * RS_SPV600)))));
  *sv00p = zp * (RS_V00 + zp * (RS_V01 + zp * (RS_V02 + zp * (RS_V03 + zp * (RS_V04 + zp * RS_V05)))));
  *zs_out = zs;
}
RQ_FN double roquet_spv(double T, double S, double pressure, double spv_ref, int anomaly) {   /* spec_vol_elem :191-248, _anomaly :253-315 */
  double zs, r0, r1, r2, r3, s0, p0;
  roquet_spv_parts(T, S, pressure, &zs, &r0, &r1, &r2, &r3, &s0, &p0);
  const double zp = pressure;
  if (anomaly) s0 = s0 - spv_ref;
  const double SV_TS = (r0 + s0) + zp * (r1 + zp * (r2 + zp * r3));
  return SV_TS + p0;
}
RQ_FN double roquet_spv_density(double T, double S, double pressure) {   /* density_elem_Roquet_SpV :318-330 */
  return 1.0 / roquet_spv(T, S, pressure, 0.0, 0);
}
RQ_FN double roquet_spv_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :335-348 */
  const double spv = roquet_spv(T, S, pressure, 1.0 / rho_ref, 1);
  return -((rho_ref * rho_ref) * spv / (rho_ref * spv + 1.0));
}
RQ_FN void roquet_spv_derivs(double T, double S, double pressure, double *dSV_dT, double *dSV_dS) {   /* calculate_specvol_derivs_elem_Roquet_SpV :352-423 */
  const double zt = T, zs = sqrt(fabs(S + RS_rdeltaS) * RS_r1_S0), zp = pressure;
  const double dRdzt3 = RS_ALP003;
  const double dRdzt2 = RS_ALP002 + (zs * RS_ALP102 + zt * RS_ALP012);
  const double dRdzt1 = RS_ALP001 + (zs * (RS_ALP101 + zs * (RS_ALP201 + zs * RS_ALP301)) +
                                     zt * (RS_ALP011 + (zs * (RS_ALP111 + zs * RS_ALP211) + zt * (RS_ALP021 + (zs * RS_ALP121 + zt * RS_ALP031)))));
  const double dRdzt0 = RS_ALP000 + (zs * (RS_ALP100 + zs * (RS_ALP200 + zs * (RS_ALP300 + zs * (RS_ALP400 + zs * RS_ALP500)))) +
                                     zt * (RS_ALP010 + (zs * (RS_ALP110 + zs * (RS_ALP210 + zs * (RS_ALP310 + zs * RS_ALP410))) +
                                                        zt * (RS_ALP020 + (zs * (RS_ALP120 + zs * (RS_ALP220 + zs * RS_ALP320)) +
                                                                           zt * (RS_ALP030 + (zt * (RS_ALP040 + (zs * RS_ALP140 + zt * RS_ALP050)) +
                                                                                              zs * (RS_ALP130 + zs * RS_ALP230))))))));
  *dSV_dT = dRdzt0 + zp * (dRdzt1 + zp * (dRdzt2 + zp * dRdzt3));
  const double dRdzs3 = RS_BET003;
  const double dRdzs2 = RS_BET002 + (zs * RS_BET102 + zt * RS_BET012);
  const double dRdzs1 = RS_BET001 + (zs * (RS_BET101 + zs * (RS_BET201 + zs * RS_BET301)) +
                                     zt * (RS_BET011 + (zs * (RS_BET111 + zs * RS_BET211) + zt * (RS_BET021 + (zs * RS_BET121 + zt * RS_BET031)))));
  const double dRdzs0 = RS_BET000 + (zs * (RS_BET100 + zs * (RS_BET200 + zs * (RS_BET300 + zs * (RS_BET400 + zs * RS_BET500)))) +
                                     zt * (RS_BET010 + (zs * (RS_BET110 + zs * (RS_BET210 + zs * (RS_BET310 + zs * RS_BET410))) +
                                                        zt * (RS_BET020 + (zs * (RS_BET120 + zs * (RS_BET220 + zs * RS_BET320)) +
                                                                           zt * (RS_BET030 + (zt * (RS_BET040 + (zs * RS_BET140 + zt * RS_BET050)) +
                                                                                              zs * (RS_BET130 + zs * RS_BET230))))))));
  *dSV_dS = (dRdzs0 + zp * (dRdzs1 + zp * (dRdzs2 + zp * dRdzs3))) / zs;
}
RQ_FN void roquet_spv_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :427-454 */
  double dSV_dT, dSV_dS;
  roquet_spv_derivs(T, S, pressure, &dSV_dT, &dSV_dS);
  const double rho = 1.0 / roquet_spv(T, S, pressure, 0.0, 0);
  *drho_dT = -dSV_dT * (rho * rho);
  *drho_dS = -dSV_dS * (rho * rho);
}
#undef RQ_FN

#define JK_FN static
#define JK_MAX(a, b) orc_max(a, b)

/* EQN_OF_STATE = "JACKETT_06": the 25-term rational function of Jackett et al. (2006), MOM_EOS_Jackett06.F90.  RNabc / RDabc: the
 * S^a T^b p^c term of the numerator / denominator (6 = power 1.5). */
#define JK_RN000 9.9984085444849347e+02
#define JK_RN001 1.1798263740430364e-06
#define JK_RN002 -2.5862187075154352e-16
#define JK_RN010 7.3471625860981584e+00
#define JK_RN020 -5.3211231792841769e-02
#define JK_RN021 9.8920219266399117e-12
#define JK_RN022 -3.2921414007960662e-20
#define JK_RN030 3.6492439109814549e-04
#define JK_RN100 2.5880571023991390e+00
#define JK_RN101 4.6996642771754730e-10
#define JK_RN110 -6.7168282786692355e-03
#define JK_RN200 1.9203202055760151e-03
#define JK_RD001 6.7103246285651894e-10
#define JK_RD010 7.2815210113327091e-03
#define JK_RD013 -9.1534417604289062e-30
#define JK_RD020 -4.4787265461983921e-05
#define JK_RD030 3.3851002965802430e-07
#define JK_RD032 -2.4461698007024582e-25
#define JK_RD040 1.3651202389758572e-10
#define JK_RD100 1.7632126669040377e-03
#define JK_RD110 -8.8066583251206474e-06
#define JK_RD130 -1.8832689434804897e-10
#define JK_RD600 5.7463776745432097e-06
#define JK_RD620 1.4716275472242334e-09
JK_FN void jackett_num_den(double T, double S, double pressure, double *num_STP, double *den) {   /* :94-102 */
  const double S1_2 = sqrt(JK_MAX(0.0, S)), T2 = T * T;
  *num_STP = (T * (JK_RN010 + T * (JK_RN020 + T * JK_RN030)) + S * (JK_RN100 + (T * JK_RN110 + S * JK_RN200))) +
             pressure * (JK_RN001 + ((T2 * JK_RN021 + S * JK_RN101) + pressure * (JK_RN002 + T2 * JK_RN022)));
  *den = 1.0 + ((T * (JK_RD010 + T * (JK_RD020 + T * (JK_RD030 + T * JK_RD040))) +
                 S * (JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (JK_RD600 + T2 * JK_RD620)))) +
                pressure * (JK_RD001 + pressure * T * (T2 * JK_RD032 + pressure * JK_RD013)));
}
JK_FN double jackett_density(double T, double S, double pressure) {   /* density_elem_Jackett06 :77-106 */
  double num_STP, den;
  jackett_num_den(T, S, pressure, &num_STP, &den);
  const double I_den = 1.0 / den;
  return (JK_RN000 + num_STP) * I_den;
}
JK_FN double jackett_density_anomaly(double T, double S, double pressure, double rho_ref) {   /* :111-144 */
  double num_STP, den;
  jackett_num_den(T, S, pressure, &num_STP, &den);
  const double I_den = 1.0 / den;
  const double rho0 = JK_RN000 - rho_ref * den;
  return (rho0 + num_STP) * I_den;
}
JK_FN void jackett_density_derivs(double T, double S, double pressure, double *drho_dT, double *drho_dS) {   /* :216-262 */
  const double S1_2 = sqrt(JK_MAX(0.0, S)), T2 = T * T;
  const double num = JK_RN000 + ((T * (JK_RN010 + T * (JK_RN020 + T * JK_RN030)) + S * (JK_RN100 + (T * JK_RN110 + S * JK_RN200))) +
                                 pressure * (JK_RN001 + ((T2 * JK_RN021 + S * JK_RN101) + pressure * (JK_RN002 + T2 * JK_RN022))));
  const double den = 1.0 + ((T * (JK_RD010 + T * (JK_RD020 + T * (JK_RD030 + T * JK_RD040))) +
                             S * (JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (JK_RD600 + T2 * JK_RD620)))) +
                            pressure * (JK_RD001 + pressure * T * (T2 * JK_RD032 + pressure * JK_RD013)));
  const double dnum_dT = ((JK_RN010 + T * (2. * JK_RN020 + T * (3. * JK_RN030))) + S * JK_RN110) +
                         pressure * T * (2. * JK_RN021 + pressure * (2. * JK_RN022));
  const double dnum_dS = (JK_RN100 + (T * JK_RN110 + S * (2. * JK_RN200))) + pressure * JK_RN101;
  const double dden_dT = ((JK_RD010 + T * ((2. * JK_RD020) + T * ((3. * JK_RD030) + T * (4. * JK_RD040)))) +
                          S * ((JK_RD110 + T2 * (3. * JK_RD130)) + S1_2 * T * (2. * JK_RD620))) +
                         (pressure * pressure) * (T2 * 3. * JK_RD032 + pressure * JK_RD013);
  const double dden_dS = JK_RD100 + (T * (JK_RD110 + T2 * JK_RD130) + S1_2 * (1.5 * JK_RD600 + T2 * (1.5 * JK_RD620)));
  const double I_denom2 = 1.0 / (den * den);
  *drho_dT = (dnum_dT * den - num * dden_dT) * I_denom2;
  *drho_dS = (dnum_dS * den - num * dden_dS) * I_denom2;
}
#undef JK_FN

/* calculate_density (no rho_ref): density_elem_linear MOM_EOS_linear.F90:66, density_elem_buggy_Wright :80-96, density_elem_Wright_full :73-89 */
static double eos_density(const mom6x_eos_params *E, double T, double S, double p) {
  if (E->form == MOM6X_EOS_LINEAR) return E->Rho_T0_S0 + E->dRho_dT * T + E->dRho_dS * S + E->dRho_dp * p;
  if (E->form == MOM6X_EOS_UNESCO) return unesco_density(T, S, p);
  if (E->form == MOM6X_EOS_ROQUET_RHO) return roquet_density(T, S, p);
  if (E->form == MOM6X_EOS_JACKETT06) return jackett_density(T, S, p);
  if (E->form == MOM6X_EOS_ROQUET_SPV) return roquet_spv_density(T, S, p);
  double al0, p0, lambda;
  wright_coefs(E->form, T, S, &al0, &p0, &lambda);
  return (p + p0) / (lambda + al0 * (p + p0));
}

/* calculate_density_derivs: linear :117-134, Wright :178-206, Wright_full / _red :177-202 */
static void eos_density_derivs(const mom6x_eos_params *E, double T, double S, double p, double *dRdT, double *dRdS) {
  if (E->form == MOM6X_EOS_LINEAR) { *dRdT = E->dRho_dT; *dRdS = E->dRho_dS; return; }
  if (E->form == MOM6X_EOS_UNESCO) { unesco_density_derivs(T, S, p, dRdT, dRdS); return; }
  if (E->form == MOM6X_EOS_ROQUET_RHO) { roquet_density_derivs(T, S, p, dRdT, dRdS); return; }
  if (E->form == MOM6X_EOS_JACKETT06) { jackett_density_derivs(T, S, p, dRdT, dRdS); return; }
  if (E->form == MOM6X_EOS_ROQUET_SPV) { roquet_spv_density_derivs(T, S, p, dRdT, dRdS); return; }
  const wright_set *W = wright_of(E->form);
  double al0, p0, lambda;
  wright_coefs(E->form, T, S, &al0, &p0, &lambda);
  if (E->form == MOM6X_EOS_WRIGHT) {
    double I_denom2 = 1.0 / (lambda + al0 * (p + p0));
    I_denom2 = I_denom2 * I_denom2;
    *dRdT = I_denom2 * (lambda * (W->b1 + T * (2.0 * W->b2 + 3.0 * W->b3 * T) + W->b5 * S) -
                        (p + p0) * ((p + p0) * W->a1 + (W->c1 + T * (W->c2 * 2.0 + W->c3 * 3.0 * T) + W->c5 * S)));
    *dRdS = I_denom2 * (lambda * (W->b4 + W->b5 * T) - (p + p0) * ((p + p0) * W->a2 + (W->c4 + W->c5 * T)));
  } else {
    const double den = (lambda + al0 * (p + p0));
    const double I_denom2 = 1.0 / (den * den);
    *dRdT = I_denom2 * (lambda * (W->b1 + (T * (2.0 * W->b2 + 3.0 * W->b3 * T) + W->b5 * S)) -
                        (p + p0) * ((p + p0) * W->a1 + (W->c1 + (T * (W->c2 * 2.0 + W->c3 * 3.0 * T) + W->c5 * S))));
    *dRdS = I_denom2 * (lambda * (W->b4 + W->b5 * T) - (p + p0) * ((p + p0) * W->a2 + (W->c4 + W->c5 * T)));
  }
}

/* exported for the known-answer tests against the reference's own EOS_unit_tests values (MOM_EOS.F90:2077-2079 WRIGHT,
 * :2129-2131 LINEAR) */
double orc_eos_density(const mom6x_eos_params *E, double T, double S, double p) { return eos_density(E, T, S, p); }
void orc_eos_density_derivs(const mom6x_eos_params *E, double T, double S, double p, double *dRdT, double *dRdS) {
  eos_density_derivs(E, T, S, p, dRdT, dRdS);
}

/* hWght and the four T/S interpolation weights of a face between columns L and R
 * (int_density_dz_linear :392-416 / int_density_dz_wright :566-583).  Returns hWght (> 0: weighted).  */
static double face_weights(int do_mw, int top_mw, double bathyL, double bathyR, double ztL, double ztR, double zbL, double zbR,
                           double sshL, double sshR, double dz_neglect, double *LL, double *LR, double *RR, double *RL) {
  double hWght = 0.0;
  if (do_mw) hWght = orc_max(orc_max(0., -bathyL - ztR), -bathyR - ztL);
  if (top_mw) hWght = orc_max(orc_max(hWght, zbR - sshL), zbL - sshR);
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_neglect, hR = (ztR - zbR) + dz_neglect;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    *LL = (hWght * hL + hR * hL) * iDenom; *LR = (hWght * hR) * iDenom;
    *RR = (hWght * hR + hR * hL) * iDenom; *RL = (hWght * hL) * iDenom;
  } else { *LL = 1.0; *LR = 0.0; *RR = 1.0; *RL = 0.0; }
  return hWght;
}

/* One column pair of int_density_dz_linear :386-430 (x) / :432-476 (y): the face integral intx_dpa|inty_dpa. */
static double face_int_linear(const mom6x_eos_params *E, double rho_ref, double G_e, double GxRho, int do_mw, int top_mw,
                              double TL, double SL, double TR, double SR, double ztL, double zbL, double ztR, double zbR,
                              double z0L, double z0R, double bathyL, double bathyR, double sshL, double sshR, double dz_neglect,
                              double dpaL, double dpaR) {
  const double C1_6 = 1.0 / 6.0, C1_90 = 1.0 / 90.0;
  double hWght = 0.0;
  if (do_mw) hWght = orc_max(orc_max(0., -bathyL - ztR), -bathyR - ztL);
  if (top_mw) hWght = orc_max(orc_max(hWght, zbR - sshL), zbL - sshR);
  if (hWght <= 0.0) {
    const double dzL = ztL - zbL, dzR = ztR - zbR;
    double p_ave = -GxRho * (0.5 * (ztL + zbL) - z0L);
    const double raL = (E->Rho_T0_S0 - rho_ref) + ((E->dRho_dT * TL + E->dRho_dS * SL) + E->dRho_dp * p_ave);
    p_ave = -GxRho * (0.5 * (ztR + zbR) - z0R);
    const double raR = (E->Rho_T0_S0 - rho_ref) + ((E->dRho_dT * TR + E->dRho_dS * SR) + E->dRho_dp * p_ave);
    return G_e * C1_6 * ((dzL * (2.0 * raL + raR)) + (dzR * (2.0 * raR + raL)));
  }
  double LL, LR, RR, RL;
  face_weights(do_mw, top_mw, bathyL, bathyR, ztL, ztR, zbL, zbR, sshL, sshR, dz_neglect, &LL, &LR, &RR, &RL);
  double intz[5];
  intz[0] = dpaL; intz[4] = dpaR;
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = (wt_L * LL) + (wt_R * RL), wtT_R = (wt_L * LR) + (wt_R * RR);
    const double dz = (wt_L * (ztL - zbL)) + (wt_R * (ztR - zbR));
    const double p_ave = -GxRho * ((wt_L * (0.5 * (ztL + zbL) - z0L)) + (wt_R * (0.5 * (ztR + zbR) - z0R)));
    const double rho_anom = (E->Rho_T0_S0 - rho_ref) +
                            ((E->dRho_dT * ((wtT_L * TL) + (wtT_R * TR)) + E->dRho_dS * ((wtT_L * SL) + (wtT_R * SR))) + E->dRho_dp * p_ave);
    intz[m - 1] = G_e * rho_anom * dz;
  }
  return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
}

/* One column pair of int_density_dz_wright :560-607 (x) / :609-653 (y). */
static double face_int_wright(int form, double rho_ref, double G_e, double GxRho, double I_Rho, int do_mw, int top_mw,
                              double TL, double SL, double TR, double SR, double ztL, double zbL, double ztR, double zbR,
                              double z0L, double z0R, double bathyL, double bathyR, double sshL, double sshR, double dz_neglect,
                              double dpaL, double dpaR) {
  const double C1_3 = 1.0 / 3.0, C1_7 = 1.0 / 7.0, C1_9 = 1.0 / 9.0, C1_90 = 1.0 / 90.0;
  double LL, LR, RR, RL;
  face_weights(do_mw, top_mw, bathyL, bathyR, ztL, ztR, zbL, zbR, sshL, sshR, dz_neglect, &LL, &LR, &RR, &RL);
  double al0L, p0L, lamL, al0R, p0R, lamR;
  wright_coefs(form, TL, SL, &al0L, &p0L, &lamL);
  wright_coefs(form, TR, SR, &al0R, &p0R, &lamR);
  double intz[5];
  intz[0] = dpaL; intz[4] = dpaR;
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = (wt_L * LL) + (wt_R * RL), wtT_R = (wt_L * LR) + (wt_R * RR);
    const double al0 = (wtT_L * al0L) + (wtT_R * al0R);
    const double p0 = (wtT_L * p0L) + (wtT_R * p0R);
    const double lambda = (wtT_L * lamL) + (wtT_R * lamR);
    const double dz = (wt_L * (ztL - zbL)) + (wt_R * (ztR - zbR));
    const double p_ave = -GxRho * ((wt_L * (0.5 * (ztL + zbL) - z0L)) + (wt_R * (0.5 * (ztR + zbR) - z0R)));
    const double I_al0 = 1.0 / al0;
    if (form == MOM6X_EOS_WRIGHT) {   /* MOM_EOS_Wright.F90:601-605 */
      const double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
      const double eps = 0.5 * GxRho * dz * I_Lzz, eps2 = eps * eps;
      intz[m - 1] = 1.0 * (G_e * dz * ((p0 + p_ave) * (I_Lzz * I_al0) - rho_ref) - 2.0 * eps *
                           I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2))));
    } else {                          /* MOM_EOS_Wright_full.F90:606-610 */
      const double I_Lzz = 1.0 / ((p0 + p_ave) + lambda * I_al0);
      const double eps = 0.5 * (GxRho * dz) * I_Lzz, eps2 = eps * eps;
      intz[m - 1] = 1.0 * ((G_e * dz) * ((p0 + p_ave) * (I_Lzz * I_al0) - rho_ref) - 2.0 * eps *
                           (I_Rho * (lambda * (I_al0 * I_al0))) * (eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2)))));
    }
  }
  return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
}

/* density_anomaly_elem_linear (MOM_EOS_linear.F90:74-84) / density_anomaly_elem_buggy_Wright (MOM_EOS_Wright.F90:101-129):
 * calculate_density(T, S, p, rho, EOS, dom, rho_ref=rho_ref) of the quadratures below. */
static double eos_density_anomaly(const mom6x_eos_params *E, double T, double S, double pressure, double rho_ref) {
  if (E->form == MOM6X_EOS_LINEAR)
    return (E->Rho_T0_S0 - rho_ref) + ((E->dRho_dT * T + E->dRho_dS * S) + E->dRho_dp * pressure);
  if (E->form == MOM6X_EOS_UNESCO) return unesco_density_anomaly(T, S, pressure, rho_ref);
  if (E->form == MOM6X_EOS_ROQUET_RHO) return roquet_density_anomaly(T, S, pressure, rho_ref);
  if (E->form == MOM6X_EOS_JACKETT06) return jackett_density_anomaly(T, S, pressure, rho_ref);
  if (E->form == MOM6X_EOS_ROQUET_SPV) return roquet_spv_density_anomaly(T, S, pressure, rho_ref);
  const wright_set *W = wright_of(E->form);   /* the same expression in all three Wright modules (_full.F90:108-119) */
  const double pa_000 = (W->b0 * (1.0 - W->a0 * rho_ref) - rho_ref * W->c0);
  const double al_TS = W->a1 * T + W->a2 * S;
  const double al0 = W->a0 + al_TS;
  const double p_TSp = pressure + (W->b4 * S + T * (W->b1 + (T * (W->b2 + W->b3 * T) + W->b5 * S)));
  const double lam_TS = W->c4 * S + T * (W->c1 + (T * (W->c2 + W->c3 * T) + W->c5 * S));
  return (pa_000 + (p_TSp - rho_ref * (p_TSp * al0 + (W->b0 * al_TS + lam_TS)))) / ((W->c0 + lam_TS) + al0 * (W->b0 + p_TSp));
}
double orc_eos_density_anomaly(const mom6x_eos_params *E, double T, double S, double p, double rho_ref) {
  return eos_density_anomaly(E, T, S, p, rho_ref);
}

double orc_PLM_slope_wa(double h_l, double h_c, double h_r, double h_neglect, double u_l, double u_c, double u_r);
double orc_PLM_monotonized_slope(double u_l, double u_c, double u_r, double s_l, double s_c, double s_r);
double orc_PLM_extrapolate_slope(double h_l, double h_c, double h_neglect, double u_l, double u_c);

/* ALE_PLM_edge_values, MOM_ALE.F90:1520-1577 (answer_date >= 20190101: h_neglect = GV%H_subroundoff) */
void orc_ALE_PLM_edge_values(const mom6x_dims *d, const mom6x_vgrid *GV, const double *h, const double *Q, int bdry_extrap,
                             double *Q_t, double *Q_b) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double h_neglect = GV->H_subroundoff;
#pragma omp parallel
  {
  double *slp = (double *)calloc((size_t)nz + 2, sizeof(double));
#pragma omp for schedule(static)
  for (int j = -1; j <= d->nj; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
#define Hk(k) h[x + (size_t)((k) - 1) * slab]
#define Qk(k) Q[x + (size_t)((k) - 1) * slab]
    slp[1] = 0.;
    for (int k = 2; k <= nz - 1; k++) slp[k] = orc_PLM_slope_wa(Hk(k - 1), Hk(k), Hk(k + 1), h_neglect, Qk(k - 1), Qk(k), Qk(k + 1));
    slp[nz] = 0.;
    for (int k = 2; k <= nz - 1; k++) {
      const double mslp = orc_PLM_monotonized_slope(Qk(k - 1), Qk(k), Qk(k + 1), slp[k - 1], slp[k], slp[k + 1]);
      Q_t[x + (size_t)(k - 1) * slab] = Qk(k) - 0.5 * mslp;
      Q_b[x + (size_t)(k - 1) * slab] = Qk(k) + 0.5 * mslp;
    }
    if (bdry_extrap && nz >= 2) {
      double mslp = -orc_PLM_extrapolate_slope(Hk(2), Hk(1), h_neglect, Qk(2), Qk(1));
      Q_t[x] = Qk(1) - 0.5 * mslp; Q_b[x] = Qk(1) + 0.5 * mslp;
      mslp = orc_PLM_extrapolate_slope(Hk(nz - 1), Hk(nz), h_neglect, Qk(nz - 1), Qk(nz));
      Q_t[x + (size_t)(nz - 1) * slab] = Qk(nz) - 0.5 * mslp; Q_b[x + (size_t)(nz - 1) * slab] = Qk(nz) + 0.5 * mslp;
    } else {
      Q_t[x] = Qk(1); Q_b[x] = Qk(1);
      Q_t[x + (size_t)(nz - 1) * slab] = Qk(nz); Q_b[x + (size_t)(nz - 1) * slab] = Qk(nz);
    }
#undef Hk
#undef Qk
  }
  free(slp);
  }
}

void orc_edge_values_implicit_h4(int N, const double *h, const double *u, double *E1, double *E2, double h_neglect);
void orc_PPM_reconstruction(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double *C3, double h_neglect);
void orc_PPM_boundary_extrapolation(int N, const double *h, const double *u, double *E1, double *E2, double *C1, double *C2, double *C3, double h_neglect);

/* One field of TS_PPM_edge_values, MOM_ALE.F90:1581-1663 (answer_date >= 20190101: h_neglect = h_neglect_edge = GV%H_subroundoff):
 * edge_values_implicit_h4, PPM_reconstruction and, with bdry_extrap, PPM_boundary_extrapolation of every column of
 * (isc-1..iec+1, jsc-1..jec+1); Q_t, Q_b = ppol_E(:,1), ppol_E(:,2).  GV%ke >= 4. */
void orc_ALE_PPM_edge_values(const mom6x_dims *d, const mom6x_vgrid *GV, const double *h, const double *Q, int bdry_extrap,
                             double *Q_t, double *Q_b) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double h_neglect = GV->H_subroundoff, h_neglect_edge = GV->H_subroundoff;
#pragma omp parallel
  {
  double *w = (double *)calloc((size_t)7 * (nz + 2), sizeof(double));
  double *hT = w, *tmp = hT + nz + 2, *E1 = tmp + nz + 2, *E2 = E1 + nz + 2, *C1 = E2 + nz + 2, *C2 = C1 + nz + 2, *C3 = C2 + nz + 2;
#pragma omp for schedule(static)
  for (int j = -1; j <= d->nj; j++) for (int i = -1; i <= d->ni; i++) {
    const size_t x = IX2(d, i, j);
    for (int k = 1; k <= nz; k++) { hT[k] = h[x + (size_t)(k - 1) * slab]; tmp[k] = Q[x + (size_t)(k - 1) * slab]; E1[k] = 0.0; E2[k] = 0.0; }
    orc_edge_values_implicit_h4(nz, hT, tmp, E1, E2, h_neglect_edge);
    orc_PPM_reconstruction(nz, hT, tmp, E1, E2, C1, C2, C3, h_neglect);
    if (bdry_extrap) orc_PPM_boundary_extrapolation(nz, hT, tmp, E1, E2, C1, C2, C3, h_neglect);
    for (int k = 1; k <= nz; k++) { Q_t[x + (size_t)(k - 1) * slab] = E1[k]; Q_b[x + (size_t)(k - 1) * slab] = E2[k]; }
  }
  free(w);
  }
}

/* One face of int_density_dz_generic_ppm: section 2 (x, MOM_density_integrals.F90:1075-1183) or 3 (y, :1186-1308): parabolic T, S
 * in the vertical from the top, mean and bottom values of the two columns (*t*, *m*, *b*). */
static double face_int_generic_ppm(const mom6x_eos_params *E, double rho_ref, double G_e, double GxRho, double mwT, double topT,
                                   double nvT, double h_nv, double dz_subroundoff, double TtL, double TbL, double TmL, double StL,
                                   double SbL, double SmL, double TtR, double TbR, double TmR, double StR, double SbR, double SmR,
                                   double ztL, double zbL, double ztR, double zbR, double z0L, double z0R, double bathyL, double bathyR,
                                   double sshL, double sshR, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  double hWght = mwT * orc_max(orc_max(0., -bathyL - ztR), -bathyR - ztL);
  const double hWghtTop = topT * orc_max(orc_max(0., zbR - sshL), zbL - sshR);
  hWght = orc_max(hWght, hWghtTop);
  if (((ztL - zbL) > h_nv) && ((ztR - zbR) > h_nv)) hWght = nvT * hWght;
  double Ttl, Tbl, Tml, Ttr, Tbr, Tmr, Stl, Sbl, Sml, Str, Sbr, Smr;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_subroundoff, hR = (ztR - zbR) + dz_subroundoff;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1. / (hWght * (hR + hL) + hL * hR);
    Ttl = ((hWght * hR) * TtR + (hWght * hL + hR * hL) * TtL) * iDenom;
    Tbl = ((hWght * hR) * TbR + (hWght * hL + hR * hL) * TbL) * iDenom;
    Tml = ((hWght * hR) * TmR + (hWght * hL + hR * hL) * TmL) * iDenom;
    Ttr = ((hWght * hL) * TtL + (hWght * hR + hR * hL) * TtR) * iDenom;
    Tbr = ((hWght * hL) * TbL + (hWght * hR + hR * hL) * TbR) * iDenom;
    Tmr = ((hWght * hL) * TmL + (hWght * hR + hR * hL) * TmR) * iDenom;
    Stl = ((hWght * hR) * StR + (hWght * hL + hR * hL) * StL) * iDenom;
    Sbl = ((hWght * hR) * SbR + (hWght * hL + hR * hL) * SbL) * iDenom;
    Sml = ((hWght * hR) * SmR + (hWght * hL + hR * hL) * SmL) * iDenom;
    Str = ((hWght * hL) * StL + (hWght * hR + hR * hL) * StR) * iDenom;
    Sbr = ((hWght * hL) * SbL + (hWght * hR + hR * hL) * SbR) * iDenom;
    Smr = ((hWght * hL) * SmL + (hWght * hR + hR * hL) * SmR) * iDenom;
  } else {
    Ttl = TtL; Tbl = TbL; Ttr = TtR; Tbr = TbR; Tml = TmL; Tmr = TmR;
    Stl = StL; Sbl = SbL; Str = StR; Sbr = SbR; Sml = SmL; Smr = SmR;
  }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
  for (int m = 2; m <= 4; m++) {
    const double w_left = 0.25 * (double)(5 - m), w_right = 1.0 - w_left;
    const double T_top = (w_left * Ttl) + (w_right * Ttr), T_mn = (w_left * Tml) + (w_right * Tmr), T_bot = (w_left * Tbl) + (w_right * Tbr);
    const double S_top = (w_left * Stl) + (w_right * Str), S_mn = (w_left * Sml) + (w_right * Smr), S_bot = (w_left * Sbl) + (w_right * Sbr);
    const double dz_x = (w_left * (ztL - zbL)) + (w_right * (ztR - zbR));
    double p15[6], r15[6];
    p15[1] = -GxRho * ((w_left * (ztL - z0L)) + (w_right * (ztR - z0R)));
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
    const double s6 = 3.0 * (2.0 * S_mn - (S_top + S_bot)), t6 = 3.0 * (2.0 * T_mn - (T_top + T_bot));
    for (int n = 1; n <= 5; n++) {
      const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
      const double S15 = wt_t * S_top + wt_b * (S_bot + s6 * wt_t);
      const double T15 = wt_t * T_top + wt_b * (T_bot + t6 * wt_t);
      r15[n] = eos_density_anomaly(E, T15, S15, p15[n], rho_ref);
    }
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

/* One face of int_density_dz_generic_pcm (EOS_QUADRATURE; MOM_density_integrals.F90:265-339 in x, :342-414 in y): layer-mean
 * T, S, interpolated across the face with (possibly thickness-weighted) weights. */
static double face_int_generic_pcm(const mom6x_eos_params *E, double rho_ref, double G_e, double GxRho, int do_mw, int top_mw,
                                   double nvT, double h_nv, double dz_neglect, double TL, double SL, double TR, double SR,
                                   double ztL, double zbL, double ztR, double zbR, double z0L, double z0R, double bathyL, double bathyR,
                                   double sshL, double sshR, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  double hWght = 0.0;
  if (do_mw) hWght = orc_max(orc_max(0., -bathyL - ztR), -bathyR - ztL);
  if (top_mw) hWght = orc_max(orc_max(hWght, zbR - sshL), zbL - sshR);
  if (((ztL - zbL) > h_nv) && ((ztR - zbR) > h_nv)) hWght = nvT * hWght;
  double hWt_LL, hWt_LR, hWt_RR, hWt_RL;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_neglect, hR = (ztR - zbR) + dz_neglect;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    hWt_LL = (hWght * hL + hR * hL) * iDenom; hWt_LR = (hWght * hR) * iDenom;
    hWt_RR = (hWght * hR + hR * hL) * iDenom; hWt_RL = (hWght * hL) * iDenom;
  } else { hWt_LL = 1.0; hWt_LR = 0.0; hWt_RR = 1.0; hWt_RL = 0.0; }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = (wt_L * hWt_LL) + (wt_R * hWt_RL), wtT_R = (wt_L * hWt_LR) + (wt_R * hWt_RR);
    const double dz_x = (wt_L * (ztL - zbL)) + (wt_R * (ztR - zbR));
    const double T15 = (wtT_L * TL) + (wtT_R * TR), S15 = (wtT_L * SL) + (wtT_R * SR);
    double p15[6], r15[6];
    p15[1] = -GxRho * ((wt_L * (ztL - z0L)) + (wt_R * (ztR - z0R)));
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
    for (int n = 1; n <= 5; n++) r15[n] = eos_density_anomaly(E, T15, S15, p15[n], rho_ref);
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

/* One face of int_density_dz_generic_plm: section 2 (x, MOM_density_integrals.F90:640-742) or 3 (y, :745-868).  L / R: the
 * two columns; *_t, *_b: top and bottom values of the layer; zt, zb: its interfaces; ssh: e(:,:,1). */
static double face_int_generic_plm(const mom6x_eos_params *E, double rho_ref, double G_e, double GxRho, double mwT, double topT,
                                   double nvT, double h_nv, double dz_subroundoff, double TtL, double TbL, double StL, double SbL,
                                   double TtR, double TbR, double StR, double SbR, double ztL, double zbL, double ztR, double zbR,
                                   double z0L, double z0R, double bathyL, double bathyR, double sshL, double sshR, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  double hWght = mwT * orc_max(orc_max(0., -bathyL - ztR), -bathyR - ztL);
  const double hWghtTop = topT * orc_max(orc_max(0., zbR - sshL), zbL - sshR);
  hWght = orc_max(hWght, hWghtTop);
  if (((ztL - zbL) > h_nv) && ((ztR - zbR) > h_nv)) hWght = nvT * hWght;
  double Ttl, Tbl, Ttr, Tbr, Stl, Sbl, Str, Sbr;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_subroundoff, hR = (ztR - zbR) + dz_subroundoff;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1. / (hWght * (hR + hL) + hL * hR);
    Ttl = ((hWght * hR) * TtR + (hWght * hL + hR * hL) * TtL) * iDenom;
    Ttr = ((hWght * hL) * TtL + (hWght * hR + hR * hL) * TtR) * iDenom;
    Tbl = ((hWght * hR) * TbR + (hWght * hL + hR * hL) * TbL) * iDenom;
    Tbr = ((hWght * hL) * TbL + (hWght * hR + hR * hL) * TbR) * iDenom;
    Stl = ((hWght * hR) * StR + (hWght * hL + hR * hL) * StL) * iDenom;
    Str = ((hWght * hL) * StL + (hWght * hR + hR * hL) * StR) * iDenom;
    Sbl = ((hWght * hR) * SbR + (hWght * hL + hR * hL) * SbL) * iDenom;
    Sbr = ((hWght * hL) * SbL + (hWght * hR + hR * hL) * SbR) * iDenom;
  } else {
    Ttl = TtL; Tbl = TbL; Ttr = TtR; Tbr = TbR;
    Stl = StL; Sbl = SbL; Str = StR; Sbr = SbR;
  }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
  for (int m = 2; m <= 4; m++) {
    const double w_left = 0.25 * (double)(5 - m), w_right = 1.0 - w_left;
    const double dz_x = (w_left * (ztL - zbL)) + (w_right * (ztR - zbR));
    double T15[6], S15[6], p15[6], r15[6];
    T15[1] = (w_left * Ttl) + (w_right * Ttr); T15[5] = (w_left * Tbl) + (w_right * Tbr);
    S15[1] = (w_left * Stl) + (w_right * Str); S15[5] = (w_left * Sbl) + (w_right * Sbr);
    p15[1] = -GxRho * ((w_left * (ztL - z0L)) + (w_right * (ztR - z0R)));
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
    for (int n = 2; n <= 4; n++) {
      const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
      S15[n] = wt_t * S15[1] + wt_b * S15[5];
      T15[n] = wt_t * T15[1] + wt_b * T15[5];
    }
    for (int n = 1; n <= 5; n++) r15[n] = eos_density_anomaly(E, T15[n], S15[n], p15[n], rho_ref);
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

/* ------------------------------------------------------------------------------------------ */
/* PressureForce_FV_Bouss :947-2017.  T == NULL: layered (no equation of state) path; else the use_EOS path with
 * analytic_int_density_dz (MOM_EOS.F90:1384) for EOS_LINEAR / EOS_WRIGHT and Set_pbce_Bouss's use_EOS branch.      */
int orc_PressureForce_FV_Bouss(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
                               const mom6x_pgf_params *CS, const double *Rlay, const double *g_prime,
                               const double *h, double *PFu, double *PFv, double *pbce, double *eta,
                               const double *T, const double *S, const mom6x_eos_params *EOS) {
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk, st = d->pitch;
  const int Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  const size_t slab = (size_t)d->slab;
  const double *bathyT = GM(G, d, MOM6X_G_bathyT), *IdxCu = GM(G, d, MOM6X_G_IdxCu), *IdyCv = GM(G, d, MOM6X_G_IdyCv);
  const double h_neglect = GV->H_subroundoff, dz_neglect = GV->dZ_subroundoff;
  const double I_Rho0 = 1.0 / GV->Rho0;
  const double GxRho0 = GV->g_Earth * GV->Rho0;
  const double rho_ref = CS->rho_ref;
  const double GxRho_ref = CS->rho_ref_bug ? GxRho0 : GV->g_Earth * rho_ref;
  const double Z_ref = CS->Z_ref;
  double *e = (double *)calloc(slab * (nz + 1), sizeof(double)), *pa = (double *)calloc(slab * (nz + 1), sizeof(double));
  double *dpa = (double *)calloc(slab * nz, sizeof(double)), *intz_dpa = (double *)calloc(slab * nz, sizeof(double));
  double *intx_pa = (double *)calloc(slab * (nz + 1), sizeof(double)), *inty_pa = (double *)calloc(slab * (nz + 1), sizeof(double));
  double *intx_dpa = (double *)calloc(slab * nz, sizeof(double)), *inty_dpa = (double *)calloc(slab * nz, sizeof(double));
  double *dz_geo = (double *)calloc(slab, sizeof(double));

  for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) e[IX2(d, i, j) + nz * slab] = -bathyT[IX2(d, i, j)];
#pragma omp parallel for schedule(static)
  for (int j = Jsq; j <= Jeq + 1; j++) for (int k = nz - 1; k >= 0; k--) for (int i = Isq; i <= Ieq + 1; i++) {
    size_t x = IX2(d, i, j);
    e[x + k * slab] = e[x + (k + 1) * slab] + h[x + k * slab] * GV->H_to_Z;
  }
  for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
    size_t x = IX2(d, i, j);
    pa[x] = GxRho_ref * (e[x] - Z_ref);
  }
  const int use_EOS = (T != NULL);
  if (use_EOS && !(EOS && S && (EOS->form >= MOM6X_EOS_LINEAR && EOS->form <= MOM6X_EOS_ROQUET_SPV) &&
                   (EOS->form < MOM6X_EOS_UNESCO || EOS->EOS_quadrature || EOS->Recon_Scheme) &&   /* no analytic integrals: MOM_EOS.F90:1495 */
                   (EOS->Recon_Scheme >= 0 && EOS->Recon_Scheme <= 2) && (EOS->Recon_Scheme != 2 || nz >= 4))) {
    free(e); free(pa); free(dpa); free(intz_dpa); free(intx_pa); free(inty_pa); free(intx_dpa); free(inty_dpa); free(dz_geo);
    return MOM6X_EUNSUPPORTED;
  }
  if (use_EOS && (EOS->Recon_Scheme >= 1 || EOS->EOS_quadrature)) {
    /* use_ALE with PRESSURE_RECONSTRUCTION_SCHEME = 1 (:1235-1236, :1287-1296): TS_PLM_edge_values, then
     * int_density_dz_generic_plm (MOM_density_integrals.F90:418-870) layer by layer; = 2 (:1237-1238, :1297-1303):
     * TS_PPM_edge_values, then int_density_dz_generic_ppm (:874-1310); without a reconstruction and with EOS_QUADRATURE
     * (int_density_dz :95-99): int_density_dz_generic_pcm (:108-416) */
    const int ppm = (EOS->Recon_Scheme == 2), pcm = (EOS->Recon_Scheme == 0);
    const double rho0_int = CS->rho_ref_bug ? rho_ref : GV->Rho0;    /* rho0_int_density :1134-1144 */
    const double G_e = GV->g_Earth, GxRho = G_e * rho0_int, C1_90 = 1.0 / 90.0;
    const double mwT = (EOS->MassWghtInterp & 1) ? 1. : 0., topT = ((EOS->MassWghtInterp >> 1) & 1) ? 1. : 0.;
    const double nvT = EOS->MassWghtInterpVanOnly ? 0. : 1.;
    const double h_nv = GV->H_to_Z * EOS->h_nonvanished;             /* dz_nonvanished :1128 */
    double *T_t = (double *)calloc(slab * nz, sizeof(double)), *T_b = (double *)calloc(slab * nz, sizeof(double));
    double *S_t = (double *)calloc(slab * nz, sizeof(double)), *S_b = (double *)calloc(slab * nz, sizeof(double));
    if (ppm) {
      orc_ALE_PPM_edge_values(d, GV, h, S, EOS->boundary_extrap, S_t, S_b);
      orc_ALE_PPM_edge_values(d, GV, h, T, EOS->boundary_extrap, T_t, T_b);
    } else if (!pcm) {
      orc_ALE_PLM_edge_values(d, GV, h, S, EOS->boundary_extrap, S_t, S_b);
      orc_ALE_PLM_edge_values(d, GV, h, T, EOS->boundary_extrap, T_t, T_b);
    } else {
      memcpy(S_t, S, slab * nz * sizeof(double)); memcpy(S_b, S, slab * nz * sizeof(double));
      memcpy(T_t, T, slab * nz * sizeof(double)); memcpy(T_b, T, slab * nz * sizeof(double));
    }
    const int do_mw = EOS->MassWghtInterp & 1, top_mw = (EOS->MassWghtInterp >> 1) & 1;
    double *z0 = (double *)calloc(slab, sizeof(double));
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {   /* Z_0p :1264-1276 (p_atm absent) */
      size_t x = IX2(d, i, j);
      z0[x] = EOS->use_SSH_in_Z0p ? e[x] : Z_ref;
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++) {
      const double *zt = e + k * slab, *zb = e + (k + 1) * slab;
      const double *Tt = T_t + k * slab, *Tb = T_b + k * slab, *St = S_t + k * slab, *Sb = S_b + k * slab;
      const double *Tm = T + k * slab, *Sm = S + k * slab;
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {   /* 1. vertical integrals :587-637 | :1047-1073 | :243-262 */
        size_t x = IX2(d, i, j), x3 = x + k * slab;
        const double dz = zt[x] - zb[x];
        double r5[6];
        double s6 = 0., t6 = 0.;
        if (ppm) {   /* curvature coefficient of the parabolas :1050-1052 */
          s6 = 3.0 * (2.0 * Sm[x] - (St[x] + Sb[x]));
          t6 = 3.0 * (2.0 * Tm[x] - (Tt[x] + Tb[x]));
        }
        for (int n = 1; n <= 5; n++) {
          const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
          const double p5 = -GxRho * ((zt[x] - z0[x]) - 0.25 * (double)(n - 1) * dz);
          double S5, T5;
          if (ppm) { S5 = wt_t * St[x] + wt_b * (Sb[x] + s6 * wt_t); T5 = wt_t * Tt[x] + wt_b * (Tb[x] + t6 * wt_t); }
          else if (pcm) { S5 = Sm[x]; T5 = Tm[x]; }
          else { S5 = wt_t * St[x] + wt_b * Sb[x]; T5 = wt_t * Tt[x] + wt_b * Tb[x]; }
          r5[n] = eos_density_anomaly(EOS, T5, S5, p5, rho_ref);
        }
        const double rho_anom = C1_90 * (7.0 * (r5[1] + r5[5]) + 32.0 * (r5[2] + r5[4]) + 12.0 * r5[3]);
        dpa[x3] = G_e * dz * rho_anom;
        intz_dpa[x3] = 0.5 * G_e * (dz * dz) * (rho_anom - C1_90 * (16.0 * (r5[4] - r5[2]) + 7.0 * (r5[5] - r5[1])));
      }
      for (int dir = 0; dir < 2; dir++) {                                              /* 2. x :640-742, 3. y :745-868 */
        const int s2 = dir ? st : 1;
        const int a0 = dir ? is : Isq, a1 = dir ? ie : Ieq, b0 = dir ? Jsq : js, b1 = dir ? Jeq : je;
        double *out = dir ? inty_dpa : intx_dpa;
        for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
          size_t x = IX2(d, i, j), y = x + s2;
          if (ppm)
            out[x + k * slab] = face_int_generic_ppm(EOS, rho_ref, G_e, GxRho, mwT, topT, nvT, h_nv, dz_neglect, Tt[x], Tb[x], Tm[x], St[x],
                                                     Sb[x], Sm[x], Tt[y], Tb[y], Tm[y], St[y], Sb[y], Sm[y], zt[x], zb[x], zt[y], zb[y],
                                                     z0[x], z0[y], bathyT[x], bathyT[y], e[x], e[y], dpa[x + k * slab], dpa[y + k * slab]);
          else if (pcm)
            out[x + k * slab] = face_int_generic_pcm(EOS, rho_ref, G_e, GxRho, do_mw, top_mw, nvT, h_nv, dz_neglect, Tm[x], Sm[x], Tm[y],
                                                     Sm[y], zt[x], zb[x], zt[y], zb[y], z0[x], z0[y], bathyT[x], bathyT[y], e[x], e[y],
                                                     dpa[x + k * slab], dpa[y + k * slab]);
          else
          out[x + k * slab] = face_int_generic_plm(EOS, rho_ref, G_e, GxRho, mwT, topT, nvT, h_nv, dz_neglect, Tt[x], Tb[x], St[x], Sb[x],
                                                   Tt[y], Tb[y], St[y], Sb[y], zt[x], zb[x], zt[y], zb[y], z0[x], z0[y], bathyT[x],
                                                   bathyT[y], e[x], e[y], dpa[x + k * slab], dpa[y + k * slab]);
        }
      }
      if (GV->Z_to_H != 1.0)
        for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) intz_dpa[IX2(d, i, j) + k * slab] *= GV->Z_to_H;
    }
    free(z0); free(T_t); free(T_b); free(S_t); free(S_b);
  } else if (use_EOS) {   /* :1289-1316 with int_density_dz -> analytic_int_density_dz */
    const double rho0_int = CS->rho_ref_bug ? rho_ref : GV->Rho0;    /* rho0_int_density :1134-1144 */
    const double G_e = GV->g_Earth, GxRho = G_e * rho0_int, I_Rho = 1.0 / rho0_int;
    const int do_mw = EOS->MassWghtInterp & 1, top_mw = (EOS->MassWghtInterp >> 1) & 1;
    const double C1_6 = 1.0 / 6.0, C1_3 = 1.0 / 3.0, C1_7 = 1.0 / 7.0, C1_9 = 1.0 / 9.0;
    double *z0 = (double *)calloc(slab, sizeof(double));
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {   /* Z_0p :1264-1276 (p_atm absent) */
      size_t x = IX2(d, i, j);
      z0[x] = EOS->use_SSH_in_Z0p ? e[x] : Z_ref;
    }
    for (int k = 0; k < nz; k++) {
      const double *zt = e + k * slab, *zb = e + (k + 1) * slab, *Tk = T + k * slab, *Sk = S + k * slab;
      for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
        size_t x = IX2(d, i, j), x3 = x + k * slab;
        const double dz = zt[x] - zb[x];
        const double p_ave = -GxRho * (0.5 * (zt[x] + zb[x]) - z0[x]);
        if (EOS->form == MOM6X_EOS_LINEAR) {   /* MOM_EOS_linear.F90:377-384 */
          const double rho_anom = (EOS->Rho_T0_S0 - rho_ref) + EOS->dRho_dT * Tk[x] + EOS->dRho_dS * Sk[x] + EOS->dRho_dp * p_ave;
          dpa[x3] = G_e * rho_anom * dz;
          intz_dpa[x3] = 0.5 * G_e * (rho_anom - C1_6 * EOS->dRho_dp * (GxRho * dz)) * (dz * dz);
        } else {                               /* MOM_EOS_Wright.F90:554-577 */
          double al0, p0, lambda;
          wright_coefs(EOS->form, Tk[x], Sk[x], &al0, &p0, &lambda);
          const double I_al0 = 1.0 / al0;
          if (EOS->form == MOM6X_EOS_WRIGHT) {
            const double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
            const double eps = 0.5 * GxRho * dz * I_Lzz, eps2 = eps * eps;
            const double rho_anom = (p0 + p_ave) * (I_Lzz * I_al0) - rho_ref;
            const double rem = I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2)));
            dpa[x3] = 1.0 * (G_e * rho_anom * dz - 2.0 * eps * rem);
            intz_dpa[x3] = 1.0 * (0.5 * G_e * rho_anom * (dz * dz) - dz * (1.0 + eps) * rem);
          } else {                           /* MOM_EOS_Wright_full.F90:550-572 */
            const double I_Lzz = 1.0 / ((p0 + p_ave) + lambda * I_al0);
            const double eps = 0.5 * (GxRho * dz) * I_Lzz, eps2 = eps * eps;
            const double rho_anom = (p0 + p_ave) * (I_Lzz * I_al0) - rho_ref;
            const double rem = (I_Rho * (lambda * (I_al0 * I_al0))) * (eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2))));
            dpa[x3] = 1.0 * ((G_e * rho_anom) * dz - 2.0 * eps * rem);
            intz_dpa[x3] = 1.0 * (0.5 * (G_e * rho_anom) * (dz * dz) - dz * ((1.0 + eps) * rem));
          }
        }
      }
      for (int dir = 0; dir < 2; dir++) {
        const int s2 = dir ? st : 1;
        const int a0 = dir ? is : Isq, a1 = dir ? ie : Ieq, b0 = dir ? Jsq : js, b1 = dir ? Jeq : je;
        double *out = dir ? inty_dpa : intx_dpa;
        for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
          size_t x = IX2(d, i, j), y = x + s2;
          if (EOS->form == MOM6X_EOS_LINEAR)
            out[x + k * slab] = face_int_linear(EOS, rho_ref, G_e, GxRho, do_mw, top_mw, Tk[x], Sk[x], Tk[y], Sk[y], zt[x], zb[x], zt[y],
                                                zb[y], z0[x], z0[y], bathyT[x], bathyT[y], e[x], e[y], dz_neglect, dpa[x + k * slab],
                                                dpa[y + k * slab]);
          else
            out[x + k * slab] = face_int_wright(EOS->form, rho_ref, G_e, GxRho, I_Rho, do_mw, top_mw, Tk[x], Sk[x], Tk[y], Sk[y], zt[x], zb[x], zt[y],
                                                zb[y], z0[x], z0[y], bathyT[x], bathyT[y], e[x], e[y], dz_neglect, dpa[x + k * slab],
                                                dpa[y + k * slab]);
        }
      }
      if (GV->Z_to_H != 1.0)
        for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) intz_dpa[IX2(d, i, j) + k * slab] *= GV->Z_to_H;
    }
    free(z0);
  } else
#pragma omp parallel
  {
  double *dz_geo = (double *)calloc(slab, sizeof(double));   /* (per thread; shadows the routine's plane) */
#pragma omp for schedule(static)
  for (int k = 0; k < nz; k++) { /* :1323-1333 */
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j), x3 = x + k * slab;
      dz_geo[x] = GV->g_Earth * GV->H_to_Z * h[x3];
      dpa[x3] = (Rlay[k] - rho_ref) * dz_geo[x];
      intz_dpa[x3] = 0.5 * (Rlay[k] - rho_ref) * dz_geo[x] * h[x3];
    }
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX2(d, i, j);
      intx_dpa[x + k * slab] = 0.5 * (Rlay[k] - rho_ref) * (dz_geo[x] + dz_geo[x + 1]);
    }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX2(d, i, j);
      inty_dpa[x + k * slab] = 0.5 * (Rlay[k] - rho_ref) * (dz_geo[x] + dz_geo[x + st]);
    }
  }
  free(dz_geo);
  }   /* omp parallel */
  /* (column recurrences: the j loop outside, as the reference's !$OMP parallel do over j) */
#pragma omp parallel for schedule(static)
  for (int j = Jsq; j <= Jeq + 1; j++) for (int k = 0; k < nz; k++) for (int i = Isq; i <= Ieq + 1; i++) {
    size_t x = IX2(d, i, j);
    pa[x + (k + 1) * slab] = pa[x + k * slab] + dpa[x + k * slab];
  }
  for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) { size_t x = IX2(d, i, j); intx_pa[x] = 0.5 * (pa[x] + pa[x + 1]); }
  for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) { size_t x = IX2(d, i, j); inty_pa[x] = 0.5 * (pa[x] + pa[x + st]); }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = Isq; i <= Ieq; i++) {
    size_t x = IX2(d, i, j);
    intx_pa[x + (k + 1) * slab] = intx_pa[x + k * slab] + intx_dpa[x + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = Jsq; j <= Jeq; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t x = IX2(d, i, j);
    inty_pa[x + (k + 1) * slab] = inty_pa[x + k * slab] + inty_dpa[x + k * slab];
  }
  /* PFu, PFv :1794-1813 */
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
    size_t x = IX2(d, i, j), x3 = x + k * slab, xb = x + (k + 1) * slab;
    PFu[x3] = (((pa[x3] * h[x3] + intz_dpa[x3]) - (pa[x3 + 1] * h[x3 + 1] + intz_dpa[x3 + 1])) +
               ((h[x3 + 1] - h[x3]) * intx_pa[x3] - (e[xb + 1] - e[xb]) * intx_dpa[x3] * GV->Z_to_H)) *
              ((2.0 * I_Rho0 * IdxCu[x]) / ((h[x3] + h[x3 + 1]) + h_neglect));
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
    size_t x = IX2(d, i, j), x3 = x + k * slab, xb = x + (k + 1) * slab;
    PFv[x3] = (((pa[x3] * h[x3] + intz_dpa[x3]) - (pa[x3 + st] * h[x3 + st] + intz_dpa[x3 + st])) +
               ((h[x3 + st] - h[x3]) * inty_pa[x3] - (e[xb + st] - e[xb]) * inty_dpa[x3] * GV->Z_to_H)) *
              ((2.0 * I_Rho0 * IdyCv[x]) / ((h[x3] + h[x3 + st]) + h_neglect));
  }
  if (pbce && use_EOS) { /* Set_pbce_Bouss, use_EOS without rho_star :704-733; Rho0 argument = rho0_set_pbce */
    const double Rho0_arg = CS->rho_ref_bug ? rho_ref : GV->Rho0;
    const double Rho0xG = Rho0_arg * GV->g_Earth, G_Rho0 = GV->g_Earth / GV->Rho0, GFS_scale = 1.0;
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      const double Ihtot = GV->H_to_Z / ((e[x] - e[x + nz * slab]) + dz_neglect);
      double press = -Rho0xG * (e[x] - Z_ref);
      const double rho_in_situ = eos_density(EOS, T[x], S[x], press);
      pbce[x] = G_Rho0 * (GFS_scale * rho_in_situ) * GV->H_to_Z;
      for (int k = 1; k < nz; k++) {
        size_t x3 = x + k * slab;
        press = -Rho0xG * (e[x3] - Z_ref);
        const double T_int = 0.5 * (T[x3 - slab] + T[x3]), S_int = 0.5 * (S[x3 - slab] + S[x3]);
        double dR_dT, dR_dS;
        eos_density_derivs(EOS, T_int, S_int, press, &dR_dT, &dR_dS);
        pbce[x3] = pbce[x3 - slab] + G_Rho0 * ((e[x3] - e[x + nz * slab]) * Ihtot) *
                   (dR_dT * (T[x3] - T[x3 - slab]) + dR_dS * (S[x3] - S[x3 - slab]));
      }
    }
  } else
  if (pbce) { /* Set_pbce_Bouss, not use_EOS :735-746 */
    for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) {
      size_t x = IX2(d, i, j);
      double Ihtot = 1.0 / ((e[x] - e[x + nz * slab]) + dz_neglect);
      pbce[x] = g_prime[0] * GV->H_to_Z;
      for (int k = 1; k < nz; k++)
        pbce[x + k * slab] = pbce[x + (k - 1) * slab] + (g_prime[k] * GV->H_to_Z) * ((e[x + k * slab] - e[x + nz * slab]) * Ihtot);
    }
  }
  if (eta) for (int j = Jsq; j <= Jeq + 1; j++) for (int i = Isq; i <= Ieq + 1; i++) eta[IX2(d, i, j)] = e[IX2(d, i, j)] * GV->Z_to_H;
  free(e); free(pa); free(dpa); free(intz_dpa); free(intx_pa); free(inty_pa); free(intx_dpa); free(inty_dpa); free(dz_geo);
  return MOM6X_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* vertvisc :557-1228 (one direction at a time; the reference does u then v with identical code) */
static void vertvisc_dir(const mom6x_dims *d, const double *maskC, int a0, int a1, int b0, int b1, double *u,
                         const double *a_u, const double *h_u, const double *Ray_u, const double *tau,
                         double dt, double dt_Rho0, double H_to_RZ, double *tau_bot, double Hmix_stress, const double *h,
                         int st, double h_neglect) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  /* columns are independent (the reference: !$OMP parallel do over j, MOM_vert_friction.F90:698) */
#pragma omp parallel
  {
  double *c1 = (double *)calloc((size_t)nz, sizeof(double));
#pragma omp for schedule(static)
  for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
    size_t x = IX2(d, i, j);
    if (maskC[x] > 0.) {
      double surface_stress = dt_Rho0 * (maskC[x] * tau[x]);
      if (Hmix_stress > 0.0) {   /* DIRECT_STRESS :707-720 / :958-971 */
        const double Hmix = Hmix_stress, I_Hmix = 1.0 / Hmix;
        surface_stress = 0.0;
        double zDS = 0.0;
        const double stress = dt_Rho0 * tau[x];
        for (int k = 0; k < nz; k++) {
          const double h_a = 0.5 * (h[x + k * slab] + h[x + st + k * slab]) + h_neglect;
          double hfr = 1.0; if ((zDS + h_a) > Hmix) hfr = (Hmix - zDS) / h_a;
          u[x + k * slab] = u[x + k * slab] + I_Hmix * hfr * stress;
          zDS = zDS + h_a; if (zDS >= Hmix) break;
        }
      }
      double Ray = Ray_u ? Ray_u[x] : 0.;
      double b_denom_1 = h_u[x] + dt * (Ray + a_u[x]);
      double b1 = 1.0 / (b_denom_1 + dt * a_u[x + slab]);
      double d1 = b_denom_1 * b1;
      u[x] = b1 * (h_u[x] * u[x] + surface_stress);
      for (int k = 1; k < nz; k++) {
        size_t x3 = x + k * slab;
        if (Ray_u) Ray = Ray_u[x3];
        c1[k] = dt * a_u[x3] * b1;
        b_denom_1 = h_u[x3] + dt * (Ray + a_u[x3] * d1);
        b1 = 1.0 / (b_denom_1 + dt * a_u[x3 + slab]);
        d1 = b_denom_1 * b1;
        u[x3] = (h_u[x3] * u[x3] + dt * a_u[x3] * u[x3 - slab]) * b1;
      }
      for (int k = nz - 2; k >= 0; k--) u[x + k * slab] = u[x + k * slab] + c1[k + 1] * u[x + (k + 1) * slab];
    }
    if (tau_bot) {
      tau_bot[x] = H_to_RZ * (u[x + (nz - 1) * slab] * a_u[x + nz * slab]);
      if (Ray_u) for (int k = 0; k < nz; k++) tau_bot[x] = tau_bot[x] + H_to_RZ * (Ray_u[x + k * slab] * u[x + k * slab]);
    }
  }
  free(c1);
  }   /* omp parallel */
}

int orc_vertvisc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, double *u, double *v,
                 const double *a_u, const double *a_v, const double *h_u, const double *h_v,
                 const double *Ray_u, const double *Ray_v, const double *taux, const double *tauy, double dt,
                 double *taux_bot, double *tauy_bot, double Hmix_stress, const double *h) {
  /* Hmix_stress > 0: DIRECT_STRESS with that HMIX_STRESS [H]; h = vertvisc's layer thicknesses */
  const double dt_Rho0 = dt / GV->H_to_RZ;
  if (Hmix_stress > 0.0 && !h) return MOM6X_EINVAL;
  vertvisc_dir(d, GM(G, d, MOM6X_G_mask2dCu), -1, d->ni - 1, 0, d->nj - 1, u, a_u, h_u, Ray_u, taux, dt, dt_Rho0, GV->H_to_RZ, taux_bot,
               Hmix_stress, h, 1, GV->H_subroundoff);
  vertvisc_dir(d, GM(G, d, MOM6X_G_mask2dCv), 0, d->ni - 1, -1, d->nj - 1, v, a_v, h_v, Ray_v, tauy, dt, dt_Rho0, GV->H_to_RZ, tauy_bot,
               Hmix_stress, h, d->pitch, GV->H_subroundoff);
  return MOM6X_OK;
}

/* vertvisc_remnant :1229-1356 */
static void remnant_dir(const mom6x_dims *d, const double *maskC, int a0, int a1, int b0, int b1, double *vr,
                        const double *a_u, const double *h_u, const double *Ray_u, double dt) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
#pragma omp parallel
  {
  double *c1 = (double *)calloc((size_t)nz, sizeof(double));
#pragma omp for schedule(static)
  for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
    size_t x = IX2(d, i, j);
    if (!(maskC[x] > 0.)) continue;
    double Ray = Ray_u ? Ray_u[x] : 0.;
    double b_denom_1 = h_u[x] + dt * (Ray + a_u[x]);
    double b1 = 1.0 / (b_denom_1 + dt * a_u[x + slab]);
    double d1 = b_denom_1 * b1;
    vr[x] = b1 * h_u[x];
    for (int k = 1; k < nz; k++) {
      size_t x3 = x + k * slab;
      if (Ray_u) Ray = Ray_u[x3];
      c1[k] = dt * a_u[x3] * b1;
      b_denom_1 = h_u[x3] + dt * (Ray + a_u[x3] * d1);
      b1 = 1.0 / (b_denom_1 + dt * a_u[x3 + slab]);
      d1 = b_denom_1 * b1;
      vr[x3] = (h_u[x3] + dt * a_u[x3] * vr[x3 - slab]) * b1;
    }
    for (int k = nz - 2; k >= 0; k--) vr[x + k * slab] = vr[x + k * slab] + c1[k + 1] * vr[x + (k + 1) * slab];
  }
  free(c1);
  }   /* omp parallel */
}

int orc_vertvisc_remnant(const mom6x_dims *d, const double *G, double *visc_rem_u, double *visc_rem_v,
                         const double *a_u, const double *a_v, const double *h_u, const double *h_v,
                         const double *Ray_u, const double *Ray_v, double dt) {
  remnant_dir(d, GM(G, d, MOM6X_G_mask2dCu), -1, d->ni - 1, 0, d->nj - 1, visc_rem_u, a_u, h_u, Ray_u, dt);
  remnant_dir(d, GM(G, d, MOM6X_G_mask2dCv), 0, d->ni - 1, -1, d->nj - 1, visc_rem_v, a_v, h_v, Ray_v, dt);
  return MOM6X_OK;
}
