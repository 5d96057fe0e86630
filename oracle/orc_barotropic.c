/*
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * btstep and friends restated from /root/reference/src/core/MOM_barotropic.F90, on the
 * default-flag path listed in SURVEY.md section 8(b.1):
 *   USE_BT_CONT_TYPE=T, or F (BT_cont NULL: find_face_areas :5146-5237, NONLINEAR_BT_CONTINUITY on or off), INTEGRAL_BT_CONTINUITY=F, LINEARIZED_BT_CORIOLIS=T,
 *   BT_NONLIN_STRESS=F, DYNAMIC_SURFACE_PRESSURE=F, BT_LINEAR_WAVE_DRAG=F, GRADUAL_BT_ICS=F, no OBC,
 *   no SAL/tides/filters, BT_USE_WIDE_HALOS=T with BTHALO=0 (wide halo == data halo),
 *   BAROTROPIC_ANSWER_DATE >= 20190101, eta_PF_start unassociated;
 * run-time selectable: BT_PROJECT_VELOCITY, SADOURNY, BT_STRONG_DRAG, VISC_REM_BT_WEIGHT_BUG,
 *   BT_USE_OLD_CORIOLIS_BRACKET_BUG, BT_USE_VISC_REM_U_UH0, CLIP_BT_VELOCITY, BOUND_BT_CORRECTION
 *   (with BT_cont bounds), BEBT, DTBT, DT_BT_FILTER, VEL_UNDERFLOW, G_BT_EXTRA.
 * Halo updates act on ONE tile: periodic wrap where re-entrant, untouched otherwise (what
 * mpp_update_domains does on a single PE).
 */
#include "orc_common.h"

/* ---- single-tile halo update (pass_var / pass_vector, MOM_domain_infra.F90:171-560) ------ */
/* stagger: 0 h, 1 u, 2 v, 3 q.  Symmetric memory: the u-point compute domain is I=-1..ni-1
 * (both ends owned), so only I<=-2 and I>=ni are halo points. */
void orc_pass_var(const mom6x_dims *d, double *a, int stagger, int nk) {
  const int xB = (stagger == 1 || stagger == 3), yB = (stagger == 2 || stagger == 3);
  const int w = d->halo, ni = d->ni, nj = d->nj;
  for (int k = 0; k < nk; k++) {
    double *p = a + (size_t)k * d->slab;
    if (d->reentrant_x) { /* computational rows only; the corners are filled by the y pass (as the 8-neighbour exchange does) */
      for (int j = -yB; j <= nj - 1; j++) {
        for (int i = -w - xB; i <= -1 - xB; i++) p[IX2(d, i, j)] = p[IX2(d, i + ni, j)];
        for (int i = ni; i <= ni - 1 + w; i++) p[IX2(d, i, j)] = p[IX2(d, i - ni, j)];
      }
    }
    if (d->reentrant_y) {
      const int ia = d->reentrant_x ? -w - xB : -xB, ib = d->reentrant_x ? ni - 1 + w : ni - 1;
      for (int i = ia; i <= ib; i++) {
        for (int j = -w - yB; j <= -1 - yB; j++) p[IX2(d, i, j)] = p[IX2(d, i, j + nj)];
        for (int j = nj; j <= nj - 1 + w; j++) p[IX2(d, i, j)] = p[IX2(d, i, j - nj)];
      }
    }
  }
}

typedef struct {   /* local_BT_cont_u_type / _v_type, MOM_barotropic.F90:352-392 */
  double FA_EE, FA_E0, FA_W0, FA_WW, uBT_WW, uBT_EE, uh_crvW, uh_crvE, uh_WW, uh_EE;
} btcl_t;

/* find_uhbt :4610-4629 (find_vhbt :4744 is the same function of the v-point fit) */
static inline double find_uhbt(double u, const btcl_t *B) {
  if (u == 0.0) return 0.0;
  else if (u < B->uBT_EE) return (u - B->uBT_EE) * B->FA_EE + B->uh_EE;
  else if (u < 0.0) return u * (B->FA_E0 + B->uh_crvE * (u * u));
  else if (u <= B->uBT_WW) return u * (B->FA_W0 + B->uh_crvW * (u * u));
  else return (u - B->uBT_WW) * B->FA_WW + B->uh_WW;
}

/* State of barotropic_CS that persists between calls (all HOST pitched arrays). */
typedef struct orc_bt_cs {
  double *frhatu, *frhatv;            /* 3-D */
  double *IDatu, *IDatv;              /* 2-D */
  double *ubtav, *vbtav, *eta_cor;    /* 2-D */
  double *q_D, *D_u_Cor, *D_v_Cor;    /* 2-D, LINEARIZED_BT_CORIOLIS */
} orc_bt_cs;

/* barotropic_init static fields: q_D, D_u_Cor, D_v_Cor (:5865-5896), IDatu/IDatv (:6146-6163). */
int orc_barotropic_init(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
                        const mom6x_barotropic_params *P, orc_bt_cs *CS) {
  const double *bathyT = GM(G, d, MOM6X_G_bathyT), *areaT = GM(G, d, MOM6X_G_areaT);
  const double *mT = GM(G, d, MOM6X_G_mask2dT), *mCu = GM(G, d, MOM6X_G_mask2dCu), *mCv = GM(G, d, MOM6X_G_mask2dCv);
  const double *fBu = GM(G, d, MOM6X_G_CoriolisBu);
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, st = d->pitch;
  const double Z_to_H = GV->Z_to_H, Mean_SL = P->Z_ref;
  memset(CS->q_D, 0, sizeof(double) * d->slab); memset(CS->D_u_Cor, 0, sizeof(double) * d->slab);
  memset(CS->D_v_Cor, 0, sizeof(double) * d->slab);
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    CS->D_u_Cor[c] = 0.5 * (orc_max(Mean_SL + bathyT[c + 1], 0.0) + orc_max(Mean_SL + bathyT[c], 0.0)) * Z_to_H;
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    CS->D_v_Cor[c] = 0.5 * (orc_max(Mean_SL + bathyT[c + st], 0.0) + orc_max(Mean_SL + bathyT[c], 0.0)) * Z_to_H;
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (mT[c] + mT[c + st] + mT[c + 1] + mT[c + 1 + st] > 0.) {
      CS->q_D[c] = 0.25 * (P->BT_Coriolis_scale * fBu[c]) *
          ((areaT[c] + areaT[c + 1 + st]) + (areaT[c + 1] + areaT[c + st])) /
          (Z_to_H * orc_max((((areaT[c] * orc_max(Mean_SL + bathyT[c], 0.0)) +
                              (areaT[c + 1 + st] * orc_max(Mean_SL + bathyT[c + 1 + st], 0.0))) +
                             ((areaT[c + 1] * orc_max(Mean_SL + bathyT[c + 1], 0.0)) +
                              (areaT[c + st] * orc_max(Mean_SL + bathyT[c + st], 0.0)))), GV->H_subroundoff));
    } else {
      CS->q_D[c] = 0.;
    }
  }
  orc_pass_var(d, CS->q_D, 3, 1); orc_pass_var(d, CS->D_u_Cor, 1, 1); orc_pass_var(d, CS->D_v_Cor, 2, 1);
  /* .not.nonlin_stress :6146-6163 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (mCu[c] > 0.) CS->IDatu[c] = mCu[c] * 2.0 / (Z_to_H * ((bathyT[c + 1] + bathyT[c]) + 2.0 * Mean_SL));
    else CS->IDatu[c] = 0.;
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (mCv[c] > 0.) CS->IDatv[c] = mCv[c] * 2.0 / (Z_to_H * ((bathyT[c + st] + bathyT[c]) + 2.0 * Mean_SL));
    else CS->IDatv[c] = 0.;
  }
  return MOM6X_OK;
}

/* btcalc :4360-4605.  h_u/h_v present (BT_THICK_SCHEME=FROM_BT_CONT) or, when NULL, `scheme` (MOM6X_BT_THICK_*): ARITHMETIC
 * :4448-4452, HYBRID :4453-4475, HARMONIC :4476-4483; FROM_BT_CONT without h_u is the HYBRID default selected by
 * may_use_default (:4419-4424). */
int orc_btcalc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h,
               const double *h_u, const double *h_v, orc_bt_cs *CS, int scheme) {
  const double *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double *mCu = GM(G, d, MOM6X_G_mask2dCu), *mCv = GM(G, d, MOM6X_G_mask2dCv);
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double h_neglect = GV->H_subroundoff, Z_to_H = GV->Z_to_H;
  double *hat = (double *)calloc((size_t)nz, sizeof(double));
  double *e = (double *)calloc((size_t)nz + 1, sizeof(double));
  for (int dir = 0; dir < 2; dir++) {
    const int st = dir ? d->pitch : 1;
    const double *hf = dir ? h_v : h_u, *mC = dir ? mCv : mCu;
    double *fr = dir ? CS->frhatv : CS->frhatu;
    const int a0 = dir ? is : is - 1, b0 = dir ? js - 1 : js;
    for (int j = b0; j <= je; j++) for (int i = a0; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      double hattot = 0.0;
      if (hf) {
        for (int k = 0; k < nz; k++) { hat[k] = hf[c + k * slab]; hattot = hattot + hat[k]; }
      } else if (scheme == MOM6X_BT_THICK_ARITHMETIC) {
        for (int k = 0; k < nz; k++) {
          hat[k] = 0.5 * (h[c + st + k * slab] + h[c + k * slab]);
          hattot = hattot + hat[k];
        }
      } else if (scheme == MOM6X_BT_THICK_HARMONIC) {
        for (int k = 0; k < nz; k++) {
          hat[k] = 2.0 * (h[c + st + k * slab] * h[c + k * slab]) / ((h[c + st + k * slab] + h[c + k * slab]) + h_neglect);
          hattot = hattot + hat[k];
        }
      } else { /* HYBRID, or FROM_BT_CONT with may_use_default */
        e[nz] = -0.5 * Z_to_H * (bathyT[c + st] + bathyT[c]);
        double D_shallow = -Z_to_H * orc_min(bathyT[c + st], bathyT[c]);
        for (int k = nz - 1; k >= 0; k--) {
          double hp = h[c + st + k * slab], hm = h[c + k * slab];
          e[k] = e[k + 1] + 0.5 * (hp + hm);
          double h_arith = 0.5 * (hp + hm);
          if (e[k + 1] >= D_shallow) {
            hat[k] = h_arith;
          } else {
            double h_harm = (hp * hm) / (h_arith + h_neglect);
            if (e[k] <= D_shallow) hat[k] = h_harm;
            else {
              double wt_arith = (e[k] - D_shallow) / (h_arith + h_neglect);
              hat[k] = wt_arith * h_arith + (1.0 - wt_arith) * h_harm;
            }
          }
          hattot = hattot + hat[k];
        }
      }
      double Ihattot = mC[c] / (hattot + h_neglect);
      for (int k = 0; k < nz; k++) fr[c + k * slab] = hat[k] * Ihattot;
    }
  }
  free(hat); free(e);
  return MOM6X_OK;
}

/* bt_mass_source :5243-5296 */
int orc_bt_mass_source(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h,
                       const double *eta, int set_cor, orc_bt_cs *CS) {
  const double *bathyT = GM(G, d, MOM6X_G_bathyT);
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t c = IX2(d, i, j);
    double eta_h = h[c] - bathyT[c] * GV->Z_to_H;
    for (int k = 1; k < d->nk; k++) eta_h = eta_h + h[c + (size_t)k * d->slab];
    double d_eta = eta_h - eta[c];
    if (set_cor) CS->eta_cor[c] = d_eta; else CS->eta_cor[c] = CS->eta_cor[c] + d_eta;
  }
  return MOM6X_OK;
}

/* set_dtbt :3509-3633 without BT_cont: find_face_areas(add_max=SSH_add) :5208-5219. */
static int orc_set_dtbt_ex(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
                 const mom6x_barotropic_params *P, const orc_bt_cs *CS, const double *pbce,
                 double gtot_est, int use_add_max, double SSH_add, double *dtbt, double *dtbt_max_out, const double *eta) {
  const double *bathyT = GM(G, d, MOM6X_G_bathyT), *dy_Cu = GM(G, d, MOM6X_G_dy_Cu), *dx_Cv = GM(G, d, MOM6X_G_dx_Cv);
  const double *IareaT = GM(G, d, MOM6X_G_IareaT), *IdxCu = GM(G, d, MOM6X_G_IdxCu), *IdyCv = GM(G, d, MOM6X_G_IdyCv);
  const double *f2 = GM(G, d, MOM6X_G_Coriolis2Bu);
  const int st = d->pitch, nz = d->nk;
  const size_t slab = (size_t)d->slab;
  double *Datu = (double *)calloc(slab, sizeof(double)), *Datv = (double *)calloc(slab, sizeof(double));
  for (int j = 0; j < d->nj; j++) for (int i = -1; i < d->ni; i++) {
    size_t c = IX2(d, i, j);
    if (use_add_max) {
      Datu[c] = dy_Cu[c] * GV->Z_to_H * orc_max(orc_max(bathyT[c + 1], bathyT[c]) + (P->Z_ref + SSH_add), 0.0);
    } else { /* find_face_areas without eta/add_max :5221-5236; with eta (NONLINEAR_BT_CONTINUITY, Boussinesq) :5171-5186 */
      double H1 = (bathyT[c] + P->Z_ref) * GV->Z_to_H, H2 = (bathyT[c + 1] + P->Z_ref) * GV->Z_to_H;
      if (eta) { H1 = bathyT[c] * GV->Z_to_H + eta[c]; H2 = bathyT[c + 1] * GV->Z_to_H + eta[c + 1]; }
      Datu[c] = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) Datu[c] = dy_Cu[c] * (2.0 * H1 * H2) / (H1 + H2);
    }
  }
  for (int j = -1; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t c = IX2(d, i, j);
    if (use_add_max) {
      Datv[c] = dx_Cv[c] * GV->Z_to_H * orc_max(orc_max(bathyT[c + st], bathyT[c]) + (P->Z_ref + SSH_add), 0.0);
    } else {
      double H1 = (bathyT[c] + P->Z_ref) * GV->Z_to_H, H2 = (bathyT[c + st] + P->Z_ref) * GV->Z_to_H;
      if (eta) { H1 = bathyT[c] * GV->Z_to_H + eta[c]; H2 = bathyT[c + st] * GV->Z_to_H + eta[c + st]; }
      Datv[c] = 0.0;
      if ((H1 > 0.0) && (H2 > 0.0)) Datv[c] = dx_Cv[c] * (2.0 * H1 * H2) / (H1 + H2);
    }
  }
  double dgeo_de = 1.0 + orc_max(0.0, P->G_extra - 0.0);
  double min_max_dt2 = 1.0e38;
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t c = IX2(d, i, j);
    double gE = gtot_est, gW = gtot_est, gN = gtot_est, gS = gtot_est;
    if (pbce) {
      gE = gW = gN = gS = 0.0;
      for (int k = 0; k < nz; k++) {
        double pb = pbce[c + k * slab];
        gE = gE + pb * CS->frhatu[c + k * slab];
        gW = gW + pb * CS->frhatu[c - 1 + k * slab];
        gN = gN + pb * CS->frhatv[c + k * slab];
        gS = gS + pb * CS->frhatv[c - st + k * slab];
      }
    }
    double Idt_max2 = 0.5 * (1.0 + 2.0 * P->bebt) * (IareaT[c] *
        (((gE * Datu[c] * IdxCu[c]) + (gW * Datu[c - 1] * IdxCu[c - 1])) +
         ((gN * Datv[c] * IdyCv[c]) + (gS * Datv[c - st] * IdyCv[c - st]))) +
        ((f2[c] + f2[c - 1 - st]) + (f2[c - 1] + f2[c - st])) * (P->BT_Coriolis_scale * P->BT_Coriolis_scale));
    if (Idt_max2 * min_max_dt2 > 1.0) min_max_dt2 = 1.0 / Idt_max2;
  }
  double dtbt_max = sqrt(min_max_dt2 / dgeo_de);
  *dtbt = P->dtbt_fraction * dtbt_max;
  if (dtbt_max_out) *dtbt_max_out = dtbt_max;
  free(Datu); free(Datv);
  return MOM6X_OK;
}

/* set_dtbt(..., gtot_est=, SSH_add=) as called from barotropic_init :5962 */
int orc_set_dtbt(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
                 const mom6x_barotropic_params *P, const orc_bt_cs *CS, const double *pbce,
                 double gtot_est, double SSH_add, double *dtbt, double *dtbt_max_out) {
  return orc_set_dtbt_ex(d, G, GV, P, CS, pbce, gtot_est, 1, SSH_add, dtbt, dtbt_max_out, NULL);
}
/* set_dtbt(G, GV, US, CS, pbce, eta=eta) as called from step_MOM_dyn_split_RK2 :667: without a BT_cont argument and
 * without (NONLINEAR_BT_CONTINUITY and eta) the face areas are ALWAYS find_face_areas(add_max=add_SSH) with add_SSH = 0
 * (:3576-3582: the third branch; the harmonic-mean form :5221-5236 is never reached from set_dtbt).  Updates P->dtbt. */
int orc_set_dtbt_pbce(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, mom6x_barotropic_params *P,
                      const orc_bt_cs *CS, const double *pbce) {
  double dtbt = 0.0;
  int rc = orc_set_dtbt_ex(d, G, GV, P, CS, pbce, 0.0, 1, 0.0, &dtbt, NULL, NULL);
  if (rc == MOM6X_OK) P->dtbt = dtbt;
  return rc;
}
/* ... without a BT_cont_type: eta enters the face areas when NONLINEAR_BT_CONTINUITY is set (:3577-3578) */
int orc_set_dtbt_pbce_eta(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, mom6x_barotropic_params *P,
                          const orc_bt_cs *CS, const double *pbce, const double *eta) {
  double dtbt = 0.0;
  const int nonlin = P->nonlinear_continuity && eta;   /* :3578 */
  int rc = orc_set_dtbt_ex(d, G, GV, P, CS, pbce, 0.0, nonlin ? 0 : 1, 0.0, &dtbt, NULL, nonlin ? eta : NULL);
  if (rc == MOM6X_OK) P->dtbt = dtbt;
  return rc;
}

/* set_local_BT_cont_types :4876-5003 (dt_baroclinic absent => dt = 1), halo = hs */
static void set_local_BT_cont_types(const mom6x_dims *d, const mom6x_BT_cont *BT, btcl_t *Bu, btcl_t *Bv, int hs) {
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1;
  const size_t n = (size_t)d->slab;
  const double C1_3 = 1.0 / 3.0;
  double *t[12];
  const double *src[12] = { BT->uBT_EE, BT->uBT_WW, BT->FA_u_EE, BT->FA_u_E0, BT->FA_u_W0, BT->FA_u_WW,
                            BT->vBT_NN, BT->vBT_SS, BT->FA_v_NN, BT->FA_v_N0, BT->FA_v_S0, BT->FA_v_SS };
  for (int m = 0; m < 12; m++) t[m] = (double *)calloc(n, sizeof(double));
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    for (int m = 0; m < 6; m++) t[m][c] = src[m][c];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    for (int m = 6; m < 12; m++) t[m][c] = src[m][c];
  }
  for (int m = 0; m < 6; m++) orc_pass_var(d, t[m], 1, 1);
  for (int m = 6; m < 12; m++) orc_pass_var(d, t[m], 2, 1);
#pragma omp parallel for schedule(static)
  for (int j = js - hs; j <= je + hs; j++) for (int i = is - hs - 1; i <= ie + hs; i++) {
    size_t c = IX2(d, i, j);
    btcl_t *B = &Bu[c];
    B->FA_EE = t[2][c]; B->FA_E0 = t[3][c]; B->FA_W0 = t[4][c]; B->FA_WW = t[5][c];
    B->uBT_EE = 1.0 * t[0][c]; B->uBT_WW = 1.0 * t[1][c];
    B->uh_EE = B->uBT_EE * (C1_3 * (2.0 * B->FA_E0 + B->FA_EE));
    B->uh_WW = B->uBT_WW * (C1_3 * (2.0 * B->FA_W0 + B->FA_WW));
    B->uh_crvE = 0.0; B->uh_crvW = 0.0;
    if (fabs(B->uBT_WW) > 0.0) B->uh_crvW = (C1_3 * (B->FA_WW - B->FA_W0)) / (B->uBT_WW * B->uBT_WW);
    if (fabs(B->uBT_EE) > 0.0) B->uh_crvE = (C1_3 * (B->FA_EE - B->FA_E0)) / (B->uBT_EE * B->uBT_EE);
  }
#pragma omp parallel for schedule(static)
  for (int j = js - hs - 1; j <= je + hs; j++) for (int i = is - hs; i <= ie + hs; i++) {
    size_t c = IX2(d, i, j);
    btcl_t *B = &Bv[c];   /* N <-> E, S <-> W */
    B->FA_EE = t[8][c]; B->FA_E0 = t[9][c]; B->FA_W0 = t[10][c]; B->FA_WW = t[11][c];
    B->uBT_EE = 1.0 * t[6][c]; B->uBT_WW = 1.0 * t[7][c];
    B->uh_EE = B->uBT_EE * (C1_3 * (2.0 * B->FA_E0 + B->FA_EE));
    B->uh_WW = B->uBT_WW * (C1_3 * (2.0 * B->FA_W0 + B->FA_WW));
    B->uh_crvE = 0.0; B->uh_crvW = 0.0;
    if (fabs(B->uBT_WW) > 0.0) B->uh_crvW = (C1_3 * (B->FA_WW - B->FA_W0)) / (B->uBT_WW * B->uBT_WW);
    if (fabs(B->uBT_EE) > 0.0) B->uh_crvE = (C1_3 * (B->FA_EE - B->FA_E0)) / (B->uBT_EE * B->uBT_EE);
  }
  for (int m = 0; m < 12; m++) free(t[m]);
}

/* find_face_areas :5146-5237 without add_max: from bathymetry + eta (Boussinesq branch :5171-5186) or from the bathymetry alone
 * (:5216-5233), over the computational domain and `hs` points beyond it */
static void find_face_areas(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_barotropic_params *P,
                            double *Datu, double *Datv, int hs, const double *eta) {
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, st = d->pitch;
  const double *dy_Cu = GM(G, d, MOM6X_G_dy_Cu), *dx_Cv = GM(G, d, MOM6X_G_dx_Cv), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double Z_to_H = GV->Z_to_H;
  if (hs < 0) hs = 0;
#pragma omp parallel for schedule(static)
  for (int j = js - hs; j <= je + hs; j++) for (int i = is - 1 - hs; i <= ie + hs; i++) {
    size_t c = IX2(d, i, j);
    double H1, H2;
    if (eta) { H1 = bathyT[c] * Z_to_H + eta[c]; H2 = bathyT[c + 1] * Z_to_H + eta[c + 1]; }
    else { H1 = (bathyT[c] + P->Z_ref) * Z_to_H; H2 = (bathyT[c + 1] + P->Z_ref) * Z_to_H; }
    Datu[c] = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Datu[c] = dy_Cu[c] * (2.0 * H1 * H2) / (H1 + H2);
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1 - hs; j <= je + hs; j++) for (int i = is - hs; i <= ie + hs; i++) {
    size_t c = IX2(d, i, j);
    double H1, H2;
    if (eta) { H1 = bathyT[c] * Z_to_H + eta[c]; H2 = bathyT[c + st] * Z_to_H + eta[c + st]; }
    else { H1 = (bathyT[c] + P->Z_ref) * Z_to_H; H2 = (bathyT[c + st] + P->Z_ref) * Z_to_H; }
    Datv[c] = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Datv[c] = dx_Cv[c] * (2.0 * H1 * H2) / (H1 + H2);
  }
}

#define F4(a, n, c) ((a)[(size_t)4 * (c) + ((n) - 1)])   /* f_4_u(n,I,j) */

/* btstep :455-2172 with btstep_timeloop :2175-2834 and the btloop_* helpers :2956-3384. */
int orc_btstep(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
               const mom6x_barotropic_params *P, orc_bt_cs *CS, int first_direction,
               const double *U_in, const double *V_in, const double *eta_in, double dt,
               const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
               const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
               double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
               const double *visc_rem_u, const double *visc_rem_v, const mom6x_BT_cont *BT_cont,
               const double *taux_bot, const double *tauy_bot,
               const double *uh0, const double *vh0, const double *u_uh0, const double *v_vh0,
               double *etaav, int *nstep_out) {
  /* USE_BT_CONT_TYPE = False (BT_cont not associated): the barotropic continuity equation is linear in the velocities with the
   * face areas Datu, Datv of find_face_areas :5146-5237 (NONLINEAR_BT_CONTINUITY = False: its last branch, from the bathymetry);
   * BOUND_BT_CORRECTION then bounds eta_cor by eta_cor_bound (below). */
  const int use_BT_cont = (BT_cont != NULL);
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk, st = d->pitch;
  const int isd = -d->halo, ied = d->ni - 1 + d->halo, jsd = -d->halo, jed = d->nj - 1 + d->halo;
  const size_t slab = (size_t)d->slab, n3 = slab * nz;
  const double *mCu = GM(G, d, MOM6X_G_mask2dCu), *mCv = GM(G, d, MOM6X_G_mask2dCv), *mT = GM(G, d, MOM6X_G_mask2dT);
  const double *IdxCu = GM(G, d, MOM6X_G_IdxCu), *IdyCv = GM(G, d, MOM6X_G_IdyCv);
  const double *IareaT = GM(G, d, MOM6X_G_IareaT), *areaT = GM(G, d, MOM6X_G_areaT);
  const double *dy_Cu = GM(G, d, MOM6X_G_dy_Cu), *dx_Cv = GM(G, d, MOM6X_G_dx_Cv);
  const double *dxT = GM(G, d, MOM6X_G_dxT), *dyT = GM(G, d, MOM6X_G_dyT), *bathyT = GM(G, d, MOM6X_G_bathyT);
  const double subroundoff = 1e-30;

  const double Idt = 1.0 / dt;
  const int find_etaav = (etaav != NULL), add_uh0 = (uh0 != NULL);
  if (add_uh0 && !(vh0 && u_uh0 && v_vh0)) return MOM6X_EINVAL;
  /* :766-768: the face areas recomputed from eta inside the loop read one more point */
  const int evolving_face_areas = (!use_BT_cont) && P->nonlinear_continuity && (P->nonlin_cont_update_period > 0);
  const int stencil = evolving_face_areas ? 2 : 1;
  const int num_cycles = (d->halo / stencil >= 1) ? d->halo / stencil : 1;   /* min((is-isdw)/stencil,(js-jsdw)/stencil), use_wide_halos */
  const int isvf = is - (num_cycles - 1) * stencil, ievf = ie + (num_cycles - 1) * stencil;
  const int jsvf = js - (num_cycles - 1) * stencil, jevf = je + (num_cycles - 1) * stencil;
  const int nstep = (int)ceil(dt / P->dtbt - 0.0001);
  if (nstep_out) *nstep_out = nstep;
  const double Instep = 1.0 / (double)nstep;
  const double dtbt = dt * Instep;

#define NEW2(x) double *x = (double *)calloc(slab, sizeof(double))
  NEW2(q); NEW2(DCor_u); NEW2(DCor_v);
  NEW2(gtot_E); NEW2(gtot_W); NEW2(gtot_N); NEW2(gtot_S);
  NEW2(eta); NEW2(eta_PF); NEW2(Cor_ref_u); NEW2(Cor_ref_v); NEW2(BT_force_u); NEW2(BT_force_v);
  NEW2(ubt); NEW2(vbt); NEW2(bt_rem_u); NEW2(bt_rem_v); NEW2(uhbt0); NEW2(vhbt0);
  NEW2(ubt_Cor); NEW2(vbt_Cor); NEW2(uhbt); NEW2(vhbt); NEW2(u_accel_bt); NEW2(v_accel_bt);
  NEW2(av_rem_u); NEW2(av_rem_v); NEW2(eta_src); NEW2(e_anom);
  NEW2(eta_sum); NEW2(eta_wtd); NEW2(ubt_wtd); NEW2(vbt_wtd); NEW2(ubt_trans); NEW2(vbt_trans);
  NEW2(ubt_prev); NEW2(vbt_prev); NEW2(eta_pred); NEW2(PFu); NEW2(PFv); NEW2(Cor_u); NEW2(Cor_v);
  double *wt_u = (double *)calloc(n3, sizeof(double)), *wt_v = (double *)calloc(n3, sizeof(double));
  double *f_4_u = (double *)calloc(4 * slab, sizeof(double)), *f_4_v = (double *)calloc(4 * slab, sizeof(double));
  btcl_t *BTCL_u = (btcl_t *)calloc(slab, sizeof(btcl_t)), *BTCL_v = (btcl_t *)calloc(slab, sizeof(btcl_t));
  NEW2(Datu); NEW2(Datv);
/* the transport of a face at the velocity u: find_uhbt of the BT_cont fit, or Datu * u (:1221, :2639, :3053-3056) */
#define UHBT(u, c) (use_BT_cont ? find_uhbt((u), &BTCL_u[c]) : Datu[c] * (u))
#define VHBT(v, c) (use_BT_cont ? find_uhbt((v), &BTCL_v[c]) : Datv[c] * (v))

  /* linearized_BT_PV :880-893 */
  for (int j = jsvf - 2; j <= jevf + 1; j++) for (int i = isvf - 2; i <= ievf + 1; i++) q[IX2(d, i, j)] = CS->q_D[IX2(d, i, j)];
  for (int j = jsvf - 1; j <= jevf + 1; j++) for (int i = isvf - 2; i <= ievf + 1; i++) DCor_u[IX2(d, i, j)] = CS->D_u_Cor[IX2(d, i, j)];
  for (int j = jsvf - 2; j <= jevf + 1; j++) for (int i = isvf - 1; i <= ievf + 1; i++) DCor_v[IX2(d, i, j)] = CS->D_v_Cor[IX2(d, i, j)];

  /* copy input arrays into their wide-halo counterparts :996-1001 */
#pragma omp parallel for schedule(static)
  for (int j = jsd; j <= jed; j++) for (int i = isd; i <= ied; i++) {
    size_t c = IX2(d, i, j);
    eta[c] = eta_in[c]; eta_PF[c] = eta_PF_in[c];
  }

  /* wt_u, wt_v :1011-1030 */
  for (int k = 0; k < nz; k++) for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX3(d, i, j, k);
    double visc_rem = orc_min(visc_rem_u[c], 1.);
    visc_rem = orc_max(visc_rem, 1. - 0.5 * Instep / (visc_rem + subroundoff));
    visc_rem = orc_max(visc_rem, 0.);
    wt_u[c] = CS->frhatu[c] * visc_rem;
  }
  for (int k = 0; k < nz; k++) for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX3(d, i, j, k);
    double visc_rem = orc_min(visc_rem_v[c], 1.);
    visc_rem = orc_max(visc_rem, 1. - 0.5 * Instep / (visc_rem + subroundoff));
    visc_rem = orc_max(visc_rem, 0.);
    wt_v[c] = CS->frhatv[c] * visc_rem;
  }
  if (!P->wt_uv_bug) { /* :1032-1059 */
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      double tot = wt_u[c];
      for (int k = 1; k < nz; k++) tot = tot + wt_u[c + k * slab];
      if (fabs(tot) > 0.0) tot = mCu[c] / tot;
      for (int k = 0; k < nz; k++) wt_u[c + k * slab] = wt_u[c + k * slab] * tot;
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      double tot = wt_v[c];
      for (int k = 1; k < nz; k++) tot = tot + wt_v[c + k * slab];
      if (fabs(tot) > 0.0) tot = mCv[c] / tot;
      for (int k = 0; k < nz; k++) wt_v[c + k * slab] = wt_v[c + k * slab] * tot;
    }
  }

  /* ubt_Cor, vbt_Cor :1064-1070 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    ubt_Cor[c] = ubt_Cor[c] + wt_u[c + k * slab] * U_Cor[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    vbt_Cor[c] = vbt_Cor[c] + wt_v[c + k * slab] * V_Cor[c + k * slab];
  }
  /* gtot :1077-1089 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    gtot_E[c] = gtot_E[c] + pbce[c + k * slab] * wt_u[c + k * slab];
    gtot_W[c + 1] = gtot_W[c + 1] + pbce[c + 1 + k * slab] * wt_u[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    gtot_N[c] = gtot_N[c] + pbce[c + k * slab] * wt_v[c + k * slab];
    gtot_S[c + st] = gtot_S[c + st] + pbce[c + st + k * slab] * wt_v[c + k * slab];
  }
  const double dgeo_de = 1.0 + P->G_extra;

  if (use_BT_cont) {
    set_local_BT_cont_types(d, BT_cont, BTCL_u, BTCL_v, 1 + ievf - ie);
  } else {   /* :1131-1136 find_face_areas(Datu, Datv, ..., halo = 1 [, eta]) :5146-5237, then pass_Dat_uv :846, :1465-1652 */
    find_face_areas(d, G, GV, P, Datu, Datv, 1, P->nonlinear_continuity ? eta : NULL);
    orc_pass_var(d, Datu, 1, 1); orc_pass_var(d, Datv, 2, 1);
  }

  if (add_uh0) { /* :1152-1227 */
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      uhbt[c] = uhbt[c] + uh0[c + k * slab];
      ubt[c] = ubt[c] + (P->visc_rem_u_uh0 ? wt_u[c + k * slab] : CS->frhatu[c + k * slab]) * u_uh0[c + k * slab];
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      vhbt[c] = vhbt[c] + vh0[c + k * slab];
      vbt[c] = vbt[c] + (P->visc_rem_u_uh0 ? wt_v[c + k * slab] : CS->frhatv[c + k * slab]) * v_vh0[c + k * slab];
    }
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      uhbt0[c] = uhbt[c] - UHBT(ubt[c], c);
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      vhbt0[c] = vhbt[c] - VHBT(vbt[c], c);
    }
  }

  /* btstep_ubt_from_layer :3388-3418 */
  memset(ubt, 0, sizeof(double) * slab); memset(vbt, 0, sizeof(double) * slab);
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    ubt[c] = ubt[c] + wt_u[c + k * slab] * U_in[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    vbt[c] = vbt[c] + wt_v[c + k * slab] * V_in[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) if (fabs(ubt[IX2(d, i, j)]) < P->vel_underflow) ubt[IX2(d, i, j)] = 0.0;
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) if (fabs(vbt[IX2(d, i, j)]) < P->vel_underflow) vbt[IX2(d, i, j)] = 0.0;
  memset(uhbt, 0, sizeof(double) * slab); memset(vhbt, 0, sizeof(double) * slab);

  /* BT_force :1259-1330 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (mCu[c] > 0.0) BT_force_u[c] = taux[c] * GV->RZ_to_H * CS->IDatu[c] * visc_rem_u[c];
    else BT_force_u[c] = 0.0;
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (mCv[c] > 0.0) BT_force_v[c] = tauy[c] * GV->RZ_to_H * CS->IDatv[c] * visc_rem_v[c];
    else BT_force_v[c] = 0.0;
  }
  if (taux_bot && tauy_bot) {
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      if (mCu[c] > 0.0) BT_force_u[c] = BT_force_u[c] - taux_bot[c] * GV->RZ_to_H * CS->IDatu[c];
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      if (mCv[c] > 0.0) BT_force_v[c] = BT_force_v[c] - tauy_bot[c] * GV->RZ_to_H * CS->IDatv[c];
    }
  }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    BT_force_u[c] = BT_force_u[c] + wt_u[c + k * slab] * bc_accel_u[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    BT_force_v[c] = BT_force_v[c] + wt_v[c + k * slab] * bc_accel_v[c + k * slab];
  }

  /* btstep_find_Cor :2836-2894 */
#pragma omp parallel for schedule(static)
  for (int j = jsvf - 1; j <= jevf; j++) for (int i = isvf - 1; i <= ievf + 1; i++) {
    size_t c = IX2(d, i, j);
    if (P->Sadourny) {
      F4(f_4_v, 1, c) = 1.0 * DCor_u[c - 1] * q[c - 1];
      F4(f_4_v, 2, c) = 1.0 * DCor_u[c] * q[c];
      F4(f_4_v, 4, c) = 1.0 * DCor_u[c + st] * q[c];
      F4(f_4_v, 3, c) = 1.0 * DCor_u[c - 1 + st] * q[c - 1];
    } else {
      F4(f_4_v, 1, c) = 1.0 * DCor_u[c - 1] * ((q[c] + q[c - 1 - st]) + q[c - 1]) / 3.0;
      F4(f_4_v, 2, c) = 1.0 * DCor_u[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
      F4(f_4_v, 4, c) = 1.0 * DCor_u[c + st] * (q[c] + (q[c - 1] + q[c + st])) / 3.0;
      F4(f_4_v, 3, c) = 1.0 * DCor_u[c - 1 + st] * ((q[c] + q[c - 1 + st]) + q[c - 1]) / 3.0;
    }
  }
#pragma omp parallel for schedule(static)
  for (int j = jsvf - 1; j <= jevf + 1; j++) for (int i = isvf - 1; i <= ievf; i++) {
    size_t c = IX2(d, i, j);
    if (P->Sadourny) {
      F4(f_4_u, 4, c) = 1.0 * DCor_v[c + 1] * q[c];
      F4(f_4_u, 3, c) = 1.0 * DCor_v[c] * q[c];
      F4(f_4_u, 1, c) = 1.0 * DCor_v[c - st] * q[c - st];
      F4(f_4_u, 2, c) = 1.0 * DCor_v[c + 1 - st] * q[c - st];
    } else {
      F4(f_4_u, 4, c) = 1.0 * DCor_v[c + 1] * (q[c] + (q[c + 1] + q[c - st])) / 3.0;
      F4(f_4_u, 3, c) = 1.0 * DCor_v[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
      F4(f_4_u, 1, c) = 1.0 * DCor_v[c - st] * ((q[c] + q[c - 1 - st]) + q[c - st]) / 3.0;
      F4(f_4_u, 2, c) = 1.0 * DCor_v[c + 1 - st] * ((q[c] + q[c + 1 - st]) + q[c - st]) / 3.0;
    }
  }

  /* pass_gtot, pass_ubt_Cor :1431-1442 */
  orc_pass_var(d, gtot_E, 0, 1); orc_pass_var(d, gtot_N, 0, 1); orc_pass_var(d, gtot_W, 0, 1); orc_pass_var(d, gtot_S, 0, 1);
  orc_pass_var(d, ubt_Cor, 1, 1); orc_pass_var(d, vbt_Cor, 2, 1);

  /* Cor_ref :1451-1461 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    Cor_ref_u[c] = (((F4(f_4_u, 4, c) * vbt_Cor[c + 1]) + (F4(f_4_u, 1, c) * vbt_Cor[c - st])) +
                    ((F4(f_4_u, 3, c) * vbt_Cor[c]) + (F4(f_4_u, 2, c) * vbt_Cor[c + 1 - st])));
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    Cor_ref_v[c] = -1.0 * (((F4(f_4_v, 1, c) * ubt_Cor[c - 1]) + (F4(f_4_v, 4, c) * ubt_Cor[c + st])) +
                           ((F4(f_4_v, 2, c) * ubt_Cor[c]) + (F4(f_4_v, 3, c) * ubt_Cor[c - 1 + st])));
  }

  /* av_rem, bt_rem :1473-1509 */
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    av_rem_u[c] = av_rem_u[c] + CS->frhatu[c + k * slab] * visc_rem_u[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int k = 0; k < nz; k++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    av_rem_v[c] = av_rem_v[c] + CS->frhatv[c + k * slab] * visc_rem_v[c + k * slab];
  }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (P->strong_drag) bt_rem_u[c] = mCu[c] * ((nstep * av_rem_u[c]) / (1.0 + (nstep - 1) * av_rem_u[c]));
    else { bt_rem_u[c] = 0.0; if (mCu[c] * av_rem_u[c] > 0.0) bt_rem_u[c] = mCu[c] * pow(av_rem_u[c], Instep); }
  }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    if (P->strong_drag) bt_rem_v[c] = mCv[c] * ((nstep * av_rem_v[c]) / (1.0 + (nstep - 1) * av_rem_v[c]));
    else { bt_rem_v[c] = 0.0; if (mCv[c] * av_rem_v[c] > 0.0) bt_rem_v[c] = mCv[c] * pow(av_rem_v[c], Instep); }
  }

  /* eta_src :1548-1587 */
  if (P->bound_BT_corr && !(use_BT_cont && P->BT_cont_bounds)) {
    /* :1582-1585 with eta_cor_bound as barotropic_init forms it :6164-6173 (find_face_areas(Datu, Datv, ..., 1) :5221-5236) */
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      double Dat[4];
      const size_t a_[4] = { c - 1, c, c, c - st }, b_[4] = { c, c + 1, c + st, c };   /* Datu(I-1,j), Datu(I,j), Datv(i,J), Datv(i,J-1) */
      const double len_[4] = { GM(G, d, MOM6X_G_dy_Cu)[c - 1], GM(G, d, MOM6X_G_dy_Cu)[c], GM(G, d, MOM6X_G_dx_Cv)[c], GM(G, d, MOM6X_G_dx_Cv)[c - st] };
      for (int q = 0; q < 4; q++) {
        double H1 = (bathyT[a_[q]] + P->Z_ref) * GV->Z_to_H, H2 = (bathyT[b_[q]] + P->Z_ref) * GV->Z_to_H;
        Dat[q] = 0.0;
        if ((H1 > 0.0) && (H2 > 0.0)) Dat[q] = len_[q] * (2.0 * H1 * H2) / (H1 + H2);
      }
      double eta_cor_bound = IareaT[c] * 0.1 * P->maxvel * ((Dat[0] + Dat[1]) + (Dat[2] + Dat[3]));
      if (fabs(CS->eta_cor[c]) > dt * eta_cor_bound) CS->eta_cor[c] = copysign(dt * eta_cor_bound, CS->eta_cor[c]);
    }
  } else if (P->bound_BT_corr) {
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      if (mT[c] > 0.0) {
        if (CS->eta_cor[c] > 0.0) {
          double u_max_cor = dxT[c] * (P->maxCFL_BT_cont * Idt), v_max_cor = dyT[c] * (P->maxCFL_BT_cont * Idt);
          double eta_cor_max = dt * (IareaT[c] *
              (((find_uhbt(u_max_cor, &BTCL_u[c]) + uhbt0[c]) - (find_uhbt(-u_max_cor, &BTCL_u[c - 1]) + uhbt0[c - 1])) +
               ((find_uhbt(v_max_cor, &BTCL_v[c]) + vhbt0[c]) - (find_uhbt(-v_max_cor, &BTCL_v[c - st]) + vhbt0[c - st]))));
          CS->eta_cor[c] = orc_min(CS->eta_cor[c], orc_max(0.0, eta_cor_max));
        } else {
          double Htot = bathyT[c] * GV->Z_to_H + eta[c];
          CS->eta_cor[c] = orc_max(CS->eta_cor[c], -orc_max(0.0, Htot));
        }
      }
    }
  }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    eta_src[c] = mT[c] * (Instep * CS->eta_cor[c]);
  }

  /* pass_eta_bt_rem, pass_force_hbt0_Cor_ref :1639-1644 */
  orc_pass_var(d, eta_PF, 0, 1); orc_pass_var(d, eta_src, 0, 1);
  orc_pass_var(d, bt_rem_u, 1, 1); orc_pass_var(d, bt_rem_v, 2, 1);
  orc_pass_var(d, BT_force_u, 1, 1); orc_pass_var(d, BT_force_v, 2, 1);
  if (add_uh0) { orc_pass_var(d, uhbt0, 1, 1); orc_pass_var(d, vhbt0, 2, 1); }
  orc_pass_var(d, Cor_ref_u, 1, 1); orc_pass_var(d, Cor_ref_v, 2, 1);

  /* filter weights :1726-1795 (1-based) */
  double dt_filt;
  if (P->dt_bt_filter >= 0.0) dt_filt = 0.5 * orc_max(0.0, orc_min(P->dt_bt_filter, 2.0 * dt));
  else dt_filt = 0.5 * orc_max(0.0, dt * orc_min(-P->dt_bt_filter, 2.0));
  const int nfilter = (int)ceil(dt_filt / dtbt);
  const int nt = nstep + nfilter;
  if (nt == 0) return MOM6X_EINVAL;
  double *wt_vel = (double *)calloc(nt + 2, sizeof(double)), *wt_eta = (double *)calloc(nt + 2, sizeof(double));
  double *wt_trans = (double *)calloc(nt + 2, sizeof(double)), *wt_accel = (double *)calloc(nt + 2, sizeof(double));
  double *wt_accel2 = (double *)calloc(nt + 2, sizeof(double));
  double sum_wt_vel = 0.0, sum_wt_eta = 0.0, sum_wt_accel = 0.0, sum_wt_trans = 0.0;
  for (int n = 1; n <= nt; n++) {
    if ((n == nstep) || (dt_filt - abs(n - nstep) * dtbt >= 0.0)) { wt_vel[n] = 1.0; wt_eta[n] = 1.0; }
    else if (dtbt + dt_filt - abs(n - nstep) * dtbt > 0.0) { wt_vel[n] = 1.0 + (dt_filt / dtbt) - abs(n - nstep); wt_eta[n] = wt_vel[n]; }
    else { wt_vel[n] = 0.0; wt_eta[n] = 0.0; }
    sum_wt_vel = sum_wt_vel + wt_vel[n]; sum_wt_eta = sum_wt_eta + wt_eta[n];
  }
  wt_trans[nt + 1] = 0.0; wt_accel[nt + 1] = 0.0;
  for (int n = nt; n >= 1; n--) {
    wt_trans[n] = wt_trans[n + 1] + wt_eta[n];
    wt_accel[n] = wt_accel[n + 1] + wt_vel[n];
    sum_wt_accel = sum_wt_accel + wt_accel[n]; sum_wt_trans = sum_wt_trans + wt_trans[n];
  }
  double I_sum_wt_vel = 1.0 / sum_wt_vel, I_sum_wt_accel = 1.0 / sum_wt_accel;
  double I_sum_wt_eta = 1.0 / sum_wt_eta, I_sum_wt_trans = 1.0 / sum_wt_trans;
  for (int n = 1; n <= nt; n++) {
    wt_vel[n] = wt_vel[n] * I_sum_wt_vel;
    wt_accel2[n] = wt_accel[n] * I_sum_wt_accel;
    wt_trans[n] = wt_trans[n] * I_sum_wt_trans;
    wt_accel[n] = wt_accel[n] * I_sum_wt_accel;
    wt_eta[n] = wt_eta[n] * I_sum_wt_eta;
  }
  I_sum_wt_vel = 1.0; I_sum_wt_eta = 1.0; I_sum_wt_accel = 1.0; I_sum_wt_trans = 1.0;
  (void)I_sum_wt_vel; (void)I_sum_wt_trans;

  /* ---------------- btstep_timeloop :2175-2834 ---------------- */
  double trans_wt1, trans_wt2;
  if (P->BT_project_velocity) { trans_wt1 = (1.0 + P->bebt); trans_wt2 = -P->bebt; }
  else { trans_wt1 = P->bebt; trans_wt2 = (1.0 - P->bebt); }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) { size_t c = IX2(d, i, j); CS->ubtav[c] = 0.0; uhbtav[c] = 0.0; ubt_wtd[c] = 0.0; }
#pragma omp parallel for schedule(static)
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) { size_t c = IX2(d, i, j); CS->vbtav[c] = 0.0; vhbtav[c] = 0.0; vbt_wtd[c] = 0.0; }

  int isv = is, iev = ie, jsv = js, jev = je;
  for (int n = 1; n <= nt; n++) {
    if (P->clip_velocity) { /* truncate_velocities :2918-2944 */
#pragma omp parallel for schedule(static)
      for (int j = jsv; j <= jev; j++) for (int i = isv - 1; i <= iev; i++) {
        size_t c = IX2(d, i, j);
        if ((ubt[c] * (dt * dy_Cu[c])) * IareaT[c + 1] < -P->CFL_trunc)
          ubt[c] = (-0.95 * P->CFL_trunc) * (areaT[c + 1] / (dt * dy_Cu[c]));
        else if ((ubt[c] * (dt * dy_Cu[c])) * IareaT[c] > P->CFL_trunc)
          ubt[c] = (0.95 * P->CFL_trunc) * (areaT[c] / (dt * dy_Cu[c]));
      }
#pragma omp parallel for schedule(static)
      for (int j = jsv - 1; j <= jev; j++) for (int i = isv; i <= iev; i++) {
        size_t c = IX2(d, i, j);
        if ((vbt[c] * (dt * dx_Cv[c])) * IareaT[c + st] < -P->CFL_trunc)
          vbt[c] = (-0.9 * P->CFL_trunc) * (areaT[c + st] / (dt * dx_Cv[c]));
        else if ((vbt[c] * (dt * dx_Cv[c])) * IareaT[c] > P->CFL_trunc)
          vbt[c] = (0.9 * P->CFL_trunc) * (areaT[c] / (dt * dx_Cv[c]));
      }
    }
    if ((iev - stencil < ie) || (jev - stencil < je)) {
      orc_pass_var(d, eta, 0, 1); orc_pass_var(d, ubt, 1, 1); orc_pass_var(d, vbt, 2, 1);
      isv = isvf; iev = ievf; jsv = jsvf; jev = jevf;
    } else {
      isv = isv + stencil; iev = iev - stencil; jsv = jsv + stencil; jev = jev - stencil;
    }
#pragma omp parallel for schedule(static)
    for (int j = jsv; j <= jev; j++) for (int i = isv - 2; i <= iev + 1; i++) ubt_prev[IX2(d, i, j)] = ubt[IX2(d, i, j)];
#pragma omp parallel for schedule(static)
    for (int j = jsv - 2; j <= jev + 1; j++) for (int i = isv; i <= iev; i++) vbt_prev[IX2(d, i, j)] = vbt[IX2(d, i, j)];

    /* :2539-2543 */
    if (evolving_face_areas && (n > 1) && ((n - 1) % P->nonlin_cont_update_period == 0))
      find_face_areas(d, G, GV, P, Datu, Datv, 1 + iev - ie, eta);

    if (!P->BT_project_velocity) { /* btloop_eta_predictor :2956-3018, use_BT_cont branch */
#pragma omp parallel for schedule(static)
      for (int j = jsv - 1; j <= jev + 1; j++) for (int i = isv - 2; i <= iev + 1; i++) {
        size_t c = IX2(d, i, j);
        uhbt[c] = UHBT(ubt[c], c) + uhbt0[c];
      }
#pragma omp parallel for schedule(static)
      for (int j = jsv - 2; j <= jev + 1; j++) for (int i = isv - 1; i <= iev + 1; i++) {
        size_t c = IX2(d, i, j);
        vhbt[c] = VHBT(vbt[c], c) + vhbt0[c];
      }
#pragma omp parallel for schedule(static)
      for (int j = jsv - 1; j <= jev + 1; j++) for (int i = isv - 1; i <= iev + 1; i++) {
        size_t c = IX2(d, i, j);
        eta_pred[c] = (eta[c] + eta_src[c]) + (dtbt * IareaT[c]) * ((uhbt[c - 1] - uhbt[c]) + (vhbt[c - st] - vhbt[c]));
      }
    }
    const int v_first = (((n + first_direction) % 2) == 1);
    { /* btloop_find_PF :3063-3111 */
      const double *eta_PF_BT = P->BT_project_velocity ? eta : eta_pred;
      int is_v, ie_v, js_u, je_u;
      if (v_first) { is_v = isv - 1; ie_v = iev + 1; js_u = jsv; je_u = jev; }
      else { is_v = isv; ie_v = iev; js_u = jsv - 1; je_u = jev + 1; }
#pragma omp parallel for schedule(static)
      for (int j = js_u; j <= je_u; j++) for (int i = isv - 1; i <= iev; i++) {
        size_t c = IX2(d, i, j);
        PFu[c] = (((eta_PF_BT[c] - eta_PF[c]) * gtot_E[c]) - ((eta_PF_BT[c + 1] - eta_PF[c + 1]) * gtot_W[c + 1])) * dgeo_de * IdxCu[c];
      }
#pragma omp parallel for schedule(static)
      for (int j = jsv - 1; j <= jev; j++) for (int i = is_v; i <= ie_v; i++) {
        size_t c = IX2(d, i, j);
        PFv[c] = (((eta_PF_BT[c] - eta_PF[c]) * gtot_N[c]) - ((eta_PF_BT[c + st] - eta_PF[c + st]) * gtot_S[c + st])) * dgeo_de * IdyCv[c];
      }
      if (find_etaav && (fabs(wt_accel2[n]) > 0.0))
#pragma omp parallel for schedule(static)
        for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
          size_t c = IX2(d, i, j);
          eta_sum[c] = eta_sum[c] + wt_accel2[n] * eta_PF_BT[c];
        }
    }
    for (int half = 0; half < 2; half++) {
      const int do_v = (half == 0) ? v_first : !v_first;
      if (do_v) { /* btloop_update_v :3209-3303 */
        int is_v, ie_v; const int Js_v = jsv - 1, Je_v = jev;
        int bracket_bug = 0;
        if (v_first) { is_v = isv - 1; ie_v = iev + 1; } else { is_v = isv; ie_v = iev; bracket_bug = P->use_old_coriolis_bracket_bug; }
#pragma omp parallel for schedule(static)
        for (int j = Js_v; j <= Je_v; j++) for (int i = is_v; i <= ie_v; i++) {
          size_t c = IX2(d, i, j);
          if (bracket_bug)
            Cor_v[c] = -1.0 * (((F4(f_4_v, 1, c) * ubt[c - 1]) + (F4(f_4_v, 2, c) * ubt[c])) +
                               ((F4(f_4_v, 4, c) * ubt[c + st]) + (F4(f_4_v, 3, c) * ubt[c - 1 + st]))) - Cor_ref_v[c];
          else
            Cor_v[c] = -1.0 * (((F4(f_4_v, 1, c) * ubt[c - 1]) + (F4(f_4_v, 4, c) * ubt[c + st])) +
                               ((F4(f_4_v, 2, c) * ubt[c]) + (F4(f_4_v, 3, c) * ubt[c - 1 + st]))) - Cor_ref_v[c];
        }
#pragma omp parallel for schedule(static)
        for (int j = Js_v; j <= Je_v; j++) for (int i = is_v; i <= ie_v; i++) {
          size_t c = IX2(d, i, j);
          vbt[c] = bt_rem_v[c] * (vbt[c] + dtbt * ((BT_force_v[c] + Cor_v[c]) + PFv[c]));
          if (fabs(vbt[c]) < P->vel_underflow) vbt[c] = 0.0;
          v_accel_bt[c] = v_accel_bt[c] + wt_accel[n] * (Cor_v[c] + PFv[c]);
        }
      } else { /* btloop_update_u :3306-3384 */
        int js_u, je_u; const int Is_u = isv - 1, Ie_u = iev;
        if (v_first) { js_u = jsv; je_u = jev; } else { js_u = jsv - 1; je_u = jev + 1; }
#pragma omp parallel for schedule(static)
        for (int j = js_u; j <= je_u; j++) for (int i = Is_u; i <= Ie_u; i++) {
          size_t c = IX2(d, i, j);
          Cor_u[c] = (((F4(f_4_u, 4, c) * vbt[c + 1]) + (F4(f_4_u, 1, c) * vbt[c - st])) +
                      ((F4(f_4_u, 3, c) * vbt[c]) + (F4(f_4_u, 2, c) * vbt[c + 1 - st]))) - Cor_ref_u[c];
          ubt[c] = bt_rem_u[c] * (ubt[c] + dtbt * ((BT_force_u[c] + Cor_u[c]) + PFu[c]));
          if (fabs(ubt[c]) < P->vel_underflow) ubt[c] = 0.0;
        }
#pragma omp parallel for schedule(static)
        for (int j = js_u; j <= je_u; j++) for (int i = Is_u; i <= Ie_u; i++) {
          size_t c = IX2(d, i, j);
          u_accel_bt[c] = u_accel_bt[c] + wt_accel[n] * (Cor_u[c] + PFu[c]);
        }
      }
    }
    /* transports, use_BT_cont branch :2624-2632 */
#pragma omp parallel for schedule(static)
    for (int j = jsv; j <= jev; j++) for (int i = isv - 1; i <= iev; i++) {
      size_t c = IX2(d, i, j);
      ubt_trans[c] = trans_wt1 * ubt[c] + trans_wt2 * ubt_prev[c];
      uhbt[c] = UHBT(ubt_trans[c], c) + uhbt0[c];
    }
#pragma omp parallel for schedule(static)
    for (int j = jsv - 1; j <= jev; j++) for (int i = isv; i <= iev; i++) {
      size_t c = IX2(d, i, j);
      vbt_trans[c] = trans_wt1 * vbt[c] + trans_wt2 * vbt_prev[c];
      vhbt[c] = VHBT(vbt_trans[c], c) + vhbt0[c];
    }
    /* running sums :2690-2700 */
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      CS->ubtav[c] = CS->ubtav[c] + wt_trans[n] * ubt_trans[c];
      uhbtav[c] = uhbtav[c] + wt_trans[n] * uhbt[c];
      ubt_wtd[c] = ubt_wtd[c] + wt_vel[n] * ubt[c];
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      CS->vbtav[c] = CS->vbtav[c] + wt_trans[n] * vbt_trans[c];
      vhbtav[c] = vhbtav[c] + wt_trans[n] * vhbt[c];
      vbt_wtd[c] = vbt_wtd[c] + wt_vel[n] * vbt[c];
    }
    /* eta corrector :2721-2727 */
#pragma omp parallel for schedule(static)
    for (int j = jsv; j <= jev; j++) for (int i = isv; i <= iev; i++) {
      size_t c = IX2(d, i, j);
      eta[c] = (eta[c] + eta_src[c]) + (dtbt * IareaT[c]) * ((uhbt[c - 1] - uhbt[c]) + (vhbt[c - st] - vhbt[c]));
      eta_wtd[c] = eta_wtd[c] + eta[c] * wt_eta[n];
    }
  }

  /* ---------------- after the time loop :1807-1913 ---------------- */
  if (find_etaav) for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) etaav[IX2(d, i, j)] = eta_sum[IX2(d, i, j)] * I_sum_wt_accel;
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
    size_t c = IX2(d, i, j);
    e_anom[c] = dgeo_de * (0.5 * (eta[c] + eta_in[c]) - eta_PF[c]);
  }
#pragma omp parallel for schedule(static)
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) eta_out[IX2(d, i, j)] = eta_wtd[IX2(d, i, j)] * I_sum_wt_eta;
  if (find_etaav) orc_pass_var(d, etaav, 0, 1);
  orc_pass_var(d, e_anom, 0, 1);
  orc_pass_var(d, CS->ubtav, 1, 1); orc_pass_var(d, CS->vbtav, 2, 1);
  orc_pass_var(d, uhbtav, 1, 1); orc_pass_var(d, vhbtav, 2, 1);

  /* btstep_layer_accel :3432-3504 */
  const double accel_underflow = P->vel_underflow * Idt;
  for (int k = 0; k < nz; k++) {
#pragma omp parallel for schedule(static)
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) {
      size_t c = IX2(d, i, j), c3 = c + k * slab;
      accel_layer_u[c3] = (u_accel_bt[c] - (((pbce[c3 + 1] - gtot_W[c + 1]) * e_anom[c + 1]) -
                                            ((pbce[c3] - gtot_E[c]) * e_anom[c])) * IdxCu[c]);
      if (fabs(accel_layer_u[c3]) < accel_underflow) accel_layer_u[c3] = 0.0;
    }
#pragma omp parallel for schedule(static)
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j), c3 = c + k * slab;
      accel_layer_v[c3] = (v_accel_bt[c] - (((pbce[c3 + st] - gtot_S[c + st]) * e_anom[c + st]) -
                                            ((pbce[c3] - gtot_N[c]) * e_anom[c])) * IdyCv[c]);
      if (fabs(accel_layer_v[c3]) < accel_underflow) accel_layer_v[c3] = 0.0;
    }
  }

  double *all2[] = { q, DCor_u, DCor_v, gtot_E, gtot_W, gtot_N, gtot_S, eta, eta_PF, Cor_ref_u, Cor_ref_v, BT_force_u,
                     BT_force_v, ubt, vbt, bt_rem_u, bt_rem_v, uhbt0, vhbt0, ubt_Cor, vbt_Cor, uhbt, vhbt, u_accel_bt,
                     v_accel_bt, av_rem_u, av_rem_v, eta_src, e_anom, eta_sum, eta_wtd, ubt_wtd, vbt_wtd, ubt_trans,
                     vbt_trans, ubt_prev, vbt_prev, eta_pred, PFu, PFv, Cor_u, Cor_v, wt_u, wt_v, f_4_u, f_4_v,
                     wt_vel, wt_eta, wt_trans, wt_accel, wt_accel2 };
  for (size_t m = 0; m < sizeof(all2) / sizeof(all2[0]); m++) free(all2[m]);
  free(BTCL_u); free(BTCL_v); free(Datu); free(Datv);
#undef UHBT
#undef VHBT
  return MOM6X_OK;
}
