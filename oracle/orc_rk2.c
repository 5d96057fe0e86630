/*
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * step_MOM_dyn_split_RK2 and the new-run part of initialize_dyn_split_RK2 restated from
 * /root/reference/src/core/MOM_dynamics_split_RK2.F90:294-1205, :1577-1650.
 *
 * Callees that are not on the ported hot path are represented by their OUTPUTS:
 *   vertvisc_coef (+ set_viscous_ML, thickness_to_dz)  -> coefficient sets coef[stage], stage = 0 (RK2 :609),
 *       1 (:738), 2 (:1003); the same set may be passed three times (coefficients frozen over the step)
 *   horizontal_viscosity (:886)                          -> optional replacement diffu/diffv arrays
 * No OBC, no waves/Stokes, FPMIX=False, p_surf_begin/end unassociated (eta_PF_start => NULL).
 */
#include "orc_common.h"

void orc_pass_var(const mom6x_dims *d, double *a, int stagger, int nk);

typedef struct orc_bt_cs orc_bt_cs;
int orc_continuity_PPM(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_continuity_params *CS,
                       int first_direction, const double *u, const double *v, const double *hin, double *h, double *uh,
                       double *vh, double dt, const double *uhbt, const double *vhbt, const double *visc_rem_u,
                       const double *visc_rem_v, double *u_cor, double *v_cor, const mom6x_BT_cont *BT, double *du_cor,
                       double *dv_cor);
int orc_btcalc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h, const double *h_u,
               const double *h_v, orc_bt_cs *CS, int scheme);
int orc_bt_mass_source(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h, const double *eta,
                       int set_cor, orc_bt_cs *CS);
int orc_set_dtbt_pbce(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, mom6x_barotropic_params *P,
                      const orc_bt_cs *CS, const double *pbce);
int orc_set_dtbt_pbce_eta(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, mom6x_barotropic_params *P,
                          const orc_bt_cs *CS, const double *pbce, const double *eta);
int orc_btstep(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_barotropic_params *P,
               orc_bt_cs *CS, int first_direction, const double *U_in, const double *V_in, const double *eta_in,
               double dt, const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
               const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
               double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav, double *vhbtav,
               const double *visc_rem_u, const double *visc_rem_v, const mom6x_BT_cont *BT_cont, const double *taux_bot,
               const double *tauy_bot, const double *uh0, const double *vh0, const double *u_uh0, const double *v_vh0,
               double *etaav, int *nstep_out);
int orc_CorAdCalc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_coriolis_params *CS,
                  const double *u, const double *v, const double *h, const double *uh, const double *vh, double *CAu,
                  double *CAv);
int orc_PressureForce_FV_Bouss(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_pgf_params *CS,
                               const double *Rlay, const double *g_prime, const double *h, double *PFu, double *PFv,
                               double *pbce, double *eta, const double *T, const double *S, const mom6x_eos_params *EOS);
int orc_vertvisc(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, double *u, double *v, const double *a_u,
                 const double *a_v, const double *h_u, const double *h_v, const double *Ray_u, const double *Ray_v,
                 const double *taux, const double *tauy, double dt, double *taux_bot, double *tauy_bot, double Hmix_stress,
                 const double *h);
int orc_vertvisc_remnant(const mom6x_dims *d, const double *G, double *visc_rem_u, double *visc_rem_v, const double *a_u,
                         const double *a_v, const double *h_u, const double *h_v, const double *Ray_u,
                         const double *Ray_v, double dt);

typedef struct { const double *a_u, *a_v, *h_u, *h_v, *Ray_u, *Ray_v; } orc_visc_coef;
int orc_horizontal_viscosity(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_hor_visc_params *CS,
                             const double *P, const double *u, const double *v, const double *h, double *diffu, double *diffv);
int orc_vertvisc_coef(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const mom6x_vertvisc_params *CS,
                      const double *u, const double *v, const double *h, double dt, const double *Kv_bbl_u,
                      const double *Kv_bbl_v, const double *bbl_thick_u, const double *bbl_thick_v, const double *Kv_shear,
                      double *a_u, double *a_v, double *h_u, double *h_v);

/* MOM_dyn_split_RK2_CS (RK2.F90:85-273): the arrays, all HOST pitched */
typedef struct orc_rk2_cs {
  double *CAu, *CAv, *CAu_pred, *CAv_pred, *PFu, *PFv, *diffu, *diffv, *visc_rem_u, *visc_rem_v;
  double *u_accel_bt, *v_accel_bt, *u_av, *v_av, *h_av, *pbce;            /* 3-D */
  double *eta, *eta_PF, *uhbt, *vhbt, *taux_bot, *tauy_bot;               /* 2-D */
  int CAu_pred_stored;
} orc_rk2_cs;

typedef struct orc_rk2_all {   /* everything step_MOM_dyn_split_RK2 reaches through CS% pointers */
  const mom6x_dims *d; const double *G; const mom6x_vgrid *GV;
  const mom6x_continuity_params *cont; mom6x_barotropic_params *bt; const mom6x_coriolis_params *cor;
  const mom6x_pgf_params *pgf; const mom6x_rk2_params *rk2; const double *Rlay, *g_prime;
  orc_rk2_cs *CS; orc_bt_cs *BTCS; const mom6x_BT_cont *BT_cont; int first_direction;
  const double *T, *S; const mom6x_eos_params *eos;   /* tv%T, tv%S, tv%eqn_of_state (NULL: layered) */
  /* vertvisc_CS + vertvisc_type inputs: when vv != NULL the three vertvisc_coef calls of the step are made here and
   * fill vv_a_u..vv_h_v (caller-allocated); otherwise the caller's coef[stage] sets are used. */
  const mom6x_vertvisc_params *vv;
  const double *Kv_bbl_u, *Kv_bbl_v, *bbl_thick_u, *bbl_thick_v, *Kv_shear, *Ray_u, *Ray_v;
  double *vv_a_u, *vv_a_v, *vv_h_u, *vv_h_v;
  /* hor_visc_CS: when hv != NULL horizontal_viscosity is called by the step (:886) and by the new-run initialisation
   * (:1601) with the coefficient planes hv_planes (orc_hor_visc_init); otherwise diffu/diffv stay as given. */
  const mom6x_hor_visc_params *hv; const double *hv_planes;
  double Hmix_stress;   /* > 0: DIRECT_STRESS with this HMIX_STRESS [H] in the two vertvisc calls of the step */
} orc_rk2_all;

/* initialize_dyn_split_RK2 :1577-1668, variable by variable: `have` = the MOM6X_RK2_HAVE_* bits of the restart variables the
 * file held (query_initialized true: CS->eta, diffu, diffv, u_av, v_av, CAu_pred, CAv_pred hold them); the others are formed as
 * the reference forms them.  have = 0 is the new run.  (barotropic_init's ubtav :6124-6135 is the caller's.) */
int orc_restart_fills_dyn_split_RK2(const orc_rk2_all *A, const double *u, const double *v, const double *h,
                                    double *uh, double *vh, double dt, int have) {
  const mom6x_dims *d = A->d; orc_rk2_cs *CS = A->CS;
  const size_t slab = (size_t)d->slab, n3 = slab * d->nk;
  const double *bathyT = GM(A->G, d, MOM6X_G_bathyT);
  if (!A->rk2->store_CAu) return MOM6X_EUNSUPPORTED;
  if (!(have & MOM6X_RK2_HAVE_ETA)) {   /* :1578-1590 */
    for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
      size_t x = IX2(d, i, j);
      CS->eta[x] = -A->GV->Z_to_H * bathyT[x];
    }
    for (int k = 0; k < d->nk; k++) for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
      size_t x = IX2(d, i, j);
      CS->eta[x] = CS->eta[x] + h[x + k * slab];
    }
  }
  if (!(have & MOM6X_RK2_HAVE_DIFFU) && A->hv) {   /* :1599-1606 */
    int rc_ = orc_horizontal_viscosity(d, A->G, A->GV, A->hv, A->hv_planes, u, v, h, CS->diffu, CS->diffv);
    if (rc_) return rc_;
  }
  if (!(have & MOM6X_RK2_HAVE_U2)) {   /* :1608-1614 */
    memcpy(CS->u_av, u, n3 * sizeof(double)); memcpy(CS->v_av, v, n3 * sizeof(double));
  }
  if (have & MOM6X_RK2_HAVE_CAU) {     /* :1617-1619 */
    CS->CAu_pred_stored = 1;
  } else {
    if ((have & MOM6X_RK2_HAVE_UH) && (have & MOM6X_RK2_HAVE_H2)) {   /* :1621-1628 */
      orc_pass_var(d, CS->h_av, 0, d->nk);
    } else {                           /* :1629-1636 */
      double *h_tmp = (double *)malloc(n3 * sizeof(double));
      memcpy(h_tmp, h, n3 * sizeof(double));
      int rc = orc_continuity_PPM(d, A->G, A->GV, A->cont, A->first_direction, CS->u_av, CS->v_av, h, h_tmp, uh, vh, dt,
                                  NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
      if (rc) { free(h_tmp); return rc; }
      orc_pass_var(d, h_tmp, 0, d->nk);
      for (size_t n = 0; n < n3; n++) CS->h_av[n] = 0.5 * (h[n] + h_tmp[n]);
      free(h_tmp);
    }
    orc_pass_var(d, CS->u_av, 1, d->nk); orc_pass_var(d, CS->v_av, 2, d->nk);
    orc_pass_var(d, uh, 1, d->nk); orc_pass_var(d, vh, 2, d->nk);
    int rc = orc_CorAdCalc(d, A->G, A->GV, A->cor, CS->u_av, CS->v_av, CS->h_av, uh, vh, CS->CAu_pred, CS->CAv_pred);
    if (rc) return rc;
    CS->CAu_pred_stored = 1;
  }
  if (have & (MOM6X_RK2_HAVE_U2 | MOM6X_RK2_HAVE_CAU)) {   /* :1670-1679 pass_av_h_uvh */
    orc_pass_var(d, CS->u_av, 1, d->nk); orc_pass_var(d, CS->v_av, 2, d->nk);
    orc_pass_var(d, CS->CAu_pred, 1, d->nk); orc_pass_var(d, CS->CAv_pred, 2, d->nk);
  }
  return MOM6X_OK;
}

/* the new-run branch of initialize_dyn_split_RK2 :1577-1650 */
int orc_initialize_dyn_split_RK2(const orc_rk2_all *A, const double *u, const double *v, const double *h,
                                 double *uh, double *vh, double dt) {
  return orc_restart_fills_dyn_split_RK2(A, u, v, h, uh, vh, dt, 0);
}

int orc_step_dyn_split_RK2(const orc_rk2_all *A, double *u_inst, double *v_inst, double *h, double *uh, double *vh,
                           double *uhtr, double *vhtr, double *eta_av, const double *taux, const double *tauy,
                           double dt, int calc_dtbt, const orc_visc_coef coef_in[3], const double *diffu_new,
                           const double *diffv_new) {
  orc_visc_coef coef[3];
  for (int s_ = 0; s_ < 3; s_++) {
    if (A->vv) { orc_visc_coef c_ = { A->vv_a_u, A->vv_a_v, A->vv_h_u, A->vv_h_v, A->Ray_u, A->Ray_v }; coef[s_] = c_; }
    else coef[s_] = coef_in[s_];
  }
#define VV_COEF(uu, vv_, dtt) do { if (A->vv) { int rc_ = orc_vertvisc_coef(A->d, A->G, A->GV, A->vv, uu, vv_, h, dtt, A->Kv_bbl_u, A->Kv_bbl_v, \
    A->bbl_thick_u, A->bbl_thick_v, A->Kv_shear, A->vv_a_u, A->vv_a_v, A->vv_h_u, A->vv_h_v); if (rc_) return rc_; } } while (0)
  const mom6x_dims *d = A->d; const double *G = A->G; const mom6x_vgrid *GV = A->GV; orc_rk2_cs *CS = A->CS;
  const mom6x_rk2_params *R = A->rk2;
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk;
  const int Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  const size_t slab = (size_t)d->slab, n3 = slab * nz;
  const double *mCu = GM(G, d, MOM6X_G_mask2dCu), *mCv = GM(G, d, MOM6X_G_mask2dCv);
  int rc;
#define NEW3(x) double *x = (double *)calloc(n3, sizeof(double))
  NEW3(up); NEW3(vp); NEW3(hp); NEW3(u_bc_accel); NEW3(v_bc_accel); NEW3(uh_in); NEW3(vh_in);
  double *eta_pred = (double *)calloc(slab, sizeof(double));
  double *u_av = CS->u_av, *v_av = CS->v_av, *h_av = CS->h_av, *eta = CS->eta;
  if (!R->BT_use_layer_fluxes) return MOM6X_EUNSUPPORTED;
  /* USE_BT_CONT_TYPE = False: CS%BT_cont is not associated (BT_cont_BT_thick :467-469 false) */
  const mom6x_BT_cont *BTc = R->no_BT_cont ? NULL : A->BT_cont;
  const double *taux_bot = R->split_bottom_stress ? CS->taux_bot : NULL;
  const double *tauy_bot = R->split_bottom_stress ? CS->tauy_bot : NULL;

  memcpy(hp, h, n3 * sizeof(double)); /* :421-425 */

  /* PFu = d/dx M(h,T,S); pbce = dM/deta  :503 */
  rc = orc_PressureForce_FV_Bouss(d, G, GV, A->pgf, A->Rlay, A->g_prime, h, CS->PFu, CS->PFv, CS->pbce, CS->eta_PF, A->T, A->S, A->eos);
  if (rc) return rc;
  if (!CS->CAu_pred_stored) { /* :552-557 */
    rc = orc_CorAdCalc(d, G, GV, A->cor, u_av, v_av, h_av, uh, vh, CS->CAu_pred, CS->CAv_pred);
    if (rc) return rc;
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :564-571 */
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX3(d, i, j, k);
      u_bc_accel[x] = (CS->CAu_pred[x] + CS->PFu[x]) + CS->diffu[x];
    }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX3(d, i, j, k);
      v_bc_accel[x] = (CS->CAv_pred[x] + CS->PFv[x]) + CS->diffv[x];
    }
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :591-598 */
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX3(d, i, j, k);
      up[x] = mCu[IX2(d, i, j)] * (u_inst[x] + dt * u_bc_accel[x]);
    }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX3(d, i, j, k);
      vp[x] = mCv[IX2(d, i, j)] * (v_inst[x] + dt * v_bc_accel[x]);
    }
  }
  /* vertvisc_coef(up, vp, h, dt) -> coef[0]; vertvisc_remnant :609-610 */
  VV_COEF(up, vp, dt);
  orc_vertvisc_remnant(d, G, CS->visc_rem_u, CS->visc_rem_v, coef[0].a_u, coef[0].a_v, coef[0].h_u, coef[0].h_v,
                       coef[0].Ray_u, coef[0].Ray_v, dt);
  orc_pass_var(d, eta, 0, 1); /* pass_eta :620 */
  orc_pass_var(d, CS->visc_rem_u, 1, nz); orc_pass_var(d, CS->visc_rem_v, 2, nz); /* pass_visc_rem :621 */

  /* BT_cont_BT_thick true (BT_cont%h_u, h_v allocated): btcalc is called after continuity :649-652; false: from h :627-628 */
  if (!BTc) {
    if (A->bt->bt_thick_scheme == MOM6X_BT_THICK_FROM_BT_CONT) return MOM6X_EINVAL;   /* barotropic_init :5589-5591 */
    orc_btcalc(d, G, GV, h, NULL, NULL, A->BTCS, A->bt->bt_thick_scheme);
  }
  orc_bt_mass_source(d, G, GV, h, eta, 1, A->BTCS);
  rc = orc_continuity_PPM(d, G, GV, A->cont, A->first_direction, u_inst, v_inst, h, hp, uh_in, vh_in, dt, NULL, NULL,
                          CS->visc_rem_u, CS->visc_rem_v, NULL, NULL, BTc, NULL, NULL);   /* :644-648 (BT_USE_LAYER_FLUXES) */
  if (rc) return rc;
  if (BTc) orc_btcalc(d, G, GV, h, BTc->h_u, BTc->h_v, A->BTCS, A->bt->bt_thick_scheme);
  if (calc_dtbt) { rc = orc_set_dtbt_pbce_eta(d, G, GV, A->bt, A->BTCS, CS->pbce, BTc ? NULL : eta); if (rc) return rc; } /* :659-668 */

  /* predictor btstep :673-676 */
  rc = orc_btstep(d, G, GV, A->bt, A->BTCS, A->first_direction, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy,
                  CS->pbce, CS->eta_PF, u_av, v_av, CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt,
                  CS->visc_rem_u, CS->visc_rem_v, BTc, taux_bot, tauy_bot, uh_in, vh_in, u_inst, v_inst, NULL, NULL);
  if (rc) return rc;

  const double dt_pred = dt * R->be; /* :679 */
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :681-694 */
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX3(d, i, j, k);
      vp[x] = mCv[IX2(d, i, j)] * (v_inst[x] + dt_pred * (v_bc_accel[x] + CS->v_accel_bt[x]));
    }
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX3(d, i, j, k);
      up[x] = mCu[IX2(d, i, j)] * (u_inst[x] + dt_pred * (u_bc_accel[x] + CS->u_accel_bt[x]));
    }
  }
  /* vertvisc_coef(up, vp, h, dt_pred) -> coef[1]; vertvisc :738-755 */
  VV_COEF(up, vp, dt_pred);
  orc_vertvisc(d, G, GV, up, vp, coef[1].a_u, coef[1].a_v, coef[1].h_u, coef[1].h_v, coef[1].Ray_u, coef[1].Ray_v, taux,
               tauy, dt_pred, CS->taux_bot, CS->tauy_bot, A->Hmix_stress, h);
  orc_vertvisc_remnant(d, G, CS->visc_rem_u, CS->visc_rem_v, coef[1].a_u, coef[1].a_v, coef[1].h_u, coef[1].h_v,
                       coef[1].Ray_u, coef[1].Ray_v, R->visc_rem_dt_bug ? dt_pred : dt); /* :763-767 */
  orc_pass_var(d, CS->visc_rem_u, 1, nz); orc_pass_var(d, CS->visc_rem_v, 2, nz); /* :769 */
  orc_pass_var(d, up, 1, nz); orc_pass_var(d, vp, 2, nz);                           /* pass_uvp :773 */

  /* uh = u_av * h ; hp = h + dt * div . uh  :779-781 */
  rc = orc_continuity_PPM(d, G, GV, A->cont, A->first_direction, up, vp, h, hp, uh, vh, dt, CS->uhbt, CS->vhbt,
                          CS->visc_rem_u, CS->visc_rem_v, u_av, v_av, BTc, NULL, NULL);
  if (rc) return rc;
  orc_pass_var(d, hp, 0, nz); orc_pass_var(d, u_av, 1, nz); orc_pass_var(d, v_av, 2, nz);   /* pass_hp_uv :785 */
  orc_pass_var(d, uh, 1, nz); orc_pass_var(d, vh, 2, nz);

#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++) { /* :808-810 */
    size_t x = IX3(d, i, j, k);
    h_av[x] = 0.5 * (h[x] + hp[x]);
  }
  /* corrector */
  orc_bt_mass_source(d, G, GV, hp, eta_pred, 0, A->BTCS); /* :820 */
  if (R->begw != 0.0) { /* :822-833 */
    for (int k = 0; k < nz; k++) for (int j = js - 1; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {
      size_t x = IX3(d, i, j, k);
      hp[x] = (1.0 - R->begw) * h[x] + R->begw * hp[x];
    }
    rc = orc_PressureForce_FV_Bouss(d, G, GV, A->pgf, A->Rlay, A->g_prime, hp, CS->PFu, CS->PFv, CS->pbce, CS->eta_PF, A->T, A->S, A->eos);
    if (rc) return rc;
  }
  if (BTc) orc_btcalc(d, G, GV, h, BTc->h_u, BTc->h_v, A->BTCS, A->bt->bt_thick_scheme); /* :864-867 */

  /* diffu = horizontal viscosity terms (u_av) :884-888 -> replaced arrays, if supplied */
  if (diffu_new) memcpy(CS->diffu, diffu_new, n3 * sizeof(double));
  if (diffv_new) memcpy(CS->diffv, diffv_new, n3 * sizeof(double));
  if (A->hv) {   /* :886 */
    rc = orc_horizontal_viscosity(d, G, GV, A->hv, A->hv_planes, CS->u_av, CS->v_av, CS->h_av, CS->diffu, CS->diffv);
    if (rc) return rc;
  }

  rc = orc_CorAdCalc(d, G, GV, A->cor, u_av, v_av, h_av, uh, vh, CS->CAu, CS->CAv); /* :893 */
  if (rc) return rc;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :900-907 */
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX3(d, i, j, k);
      u_bc_accel[x] = (CS->CAu[x] + CS->PFu[x]) + CS->diffu[x];
    }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX3(d, i, j, k);
      v_bc_accel[x] = (CS->CAv[x] + CS->PFv[x]) + CS->diffv[x];
    }
  }
  /* corrector btstep :939-942 */
  rc = orc_btstep(d, G, GV, A->bt, A->BTCS, A->first_direction, u_inst, v_inst, eta, dt, u_bc_accel, v_bc_accel, taux, tauy,
                  CS->pbce, CS->eta_PF, u_av, v_av, CS->u_accel_bt, CS->v_accel_bt, eta_pred, CS->uhbt, CS->vhbt,
                  CS->visc_rem_u, CS->visc_rem_v, BTc, taux_bot, tauy_bot, uh, vh, u_av, v_av, eta_av, NULL);
  if (rc) return rc;
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) eta[IX2(d, i, j)] = eta_pred[IX2(d, i, j)]; /* :946 */

#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :957-966 */
    for (int j = js; j <= je; j++) for (int i = Isq; i <= Ieq; i++) {
      size_t x = IX3(d, i, j, k);
      u_inst[x] = mCu[IX2(d, i, j)] * (u_inst[x] + dt * (u_bc_accel[x] + CS->u_accel_bt[x]));
    }
    for (int j = Jsq; j <= Jeq; j++) for (int i = is; i <= ie; i++) {
      size_t x = IX3(d, i, j, k);
      v_inst[x] = mCv[IX2(d, i, j)] * (v_inst[x] + dt * (v_bc_accel[x] + CS->v_accel_bt[x]));
    }
  }
  /* vertvisc_coef(u, v, h, dt) -> coef[2]; vertvisc; vertvisc_remnant :1003-1022 */
  VV_COEF(u_inst, v_inst, dt);
  orc_vertvisc(d, G, GV, u_inst, v_inst, coef[2].a_u, coef[2].a_v, coef[2].h_u, coef[2].h_v, coef[2].Ray_u, coef[2].Ray_v,
               taux, tauy, dt, CS->taux_bot, CS->tauy_bot, A->Hmix_stress, h);
  orc_vertvisc_remnant(d, G, CS->visc_rem_u, CS->visc_rem_v, coef[2].a_u, coef[2].a_v, coef[2].h_u, coef[2].h_v,
                       coef[2].Ray_u, coef[2].Ray_v, dt);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++) { /* :1025-1027 */
    size_t x = IX3(d, i, j, k);
    h_av[x] = h[x];
  }
  orc_pass_var(d, CS->visc_rem_u, 1, nz); orc_pass_var(d, CS->visc_rem_v, 2, nz); /* :1030 */
  orc_pass_var(d, u_inst, 1, nz); orc_pass_var(d, v_inst, 2, nz);                   /* pass_uv :1034 */

  /* uh = u_av * h ; h = h + dt * div . uh :1041-1043 */
  rc = orc_continuity_PPM(d, G, GV, A->cont, A->first_direction, u_inst, v_inst, h, h, uh, vh, dt, CS->uhbt, CS->vhbt,
                          CS->visc_rem_u, CS->visc_rem_v, u_av, v_av, NULL, NULL, NULL);
  if (rc) return rc;
  orc_pass_var(d, h, 0, nz); /* pass_h :1045 */
  orc_pass_var(d, u_av, 1, nz); orc_pass_var(d, v_av, 2, nz); orc_pass_var(d, uh, 1, nz); orc_pass_var(d, vh, 2, nz); /* :1053 */

#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) for (int j = js - 2; j <= je + 2; j++) for (int i = is - 2; i <= ie + 2; i++) { /* :1064-1066 */
    size_t x = IX3(d, i, j, k);
    h_av[x] = 0.5 * (h_av[x] + h[x]);
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nz; k++) { /* :1072-1079 */
    for (int j = js - 2; j <= je + 2; j++) for (int i = Isq - 2; i <= Ieq + 2; i++) {
      size_t x = IX3(d, i, j, k);
      uhtr[x] = uhtr[x] + uh[x] * dt;
    }
    for (int j = Jsq - 2; j <= Jeq + 2; j++) for (int i = is - 2; i <= ie + 2; i++) {
      size_t x = IX3(d, i, j, k);
      vhtr[x] = vhtr[x] + vh[x] * dt;
    }
  }
  if (R->store_CAu) { /* :1081-1090 */
    rc = orc_CorAdCalc(d, G, GV, A->cor, u_av, v_av, h_av, uh, vh, CS->CAu_pred, CS->CAv_pred);
    if (rc) return rc;
    CS->CAu_pred_stored = 1;
  } else {
    CS->CAu_pred_stored = 0;
  }
  free(up); free(vp); free(hp); free(u_bc_accel); free(v_bc_accel); free(uh_in); free(vh_in); free(eta_pred);
  return MOM6X_OK;
}
