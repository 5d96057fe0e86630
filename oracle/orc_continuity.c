/*
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * continuity_PPM restated from /root/reference/src/core/MOM_continuity_PPM.F90.
 * The zonal and meridional halves of the reference are textual mirror images
 * (zonal_* :519-1409 vs meridional_* :1412-2304, PPM_reconstruction_x :2307 vs _y :2442);
 * both are expressed here through one direction-generic implementation in which a face
 * with flat index f separates the "minus" cell f and the "plus" cell f+st (st = 1 for
 * zonal faces, st = pitch for meridional faces).  Row structure, the do_I masks and the
 * row-wide `domore` early exit of the Newton iteration are kept as in the reference.
 *
 * Supported flags: the default path of continuity_PPM_init (:2674-2754) plus
 * monotonic / simple_2nd / upwind_1st; aggress_adjust and vol_CFL (:612, :651-716, :938-945, :1019-1025 and their meridional twins);
 * OBC unassociated; por_face_area == 1.
 */
#include "orc_common.h"

typedef struct {
  int dir;            /* 0: zonal, 1: meridional */
  int st;             /* flat stride from the minus to the plus cell of a face */
  const double *Lface;   /* G%dy_Cu | G%dx_Cv */
  const double *IdT;     /* G%IdxT  | G%IdyT  */
  const double *dT;      /* G%dxT   | G%dyT   */
  const double *dC;      /* G%dxCu  | G%dyCv  */
  const double *maskC;   /* G%mask2dCu | G%mask2dCv */
  const double *IareaT;
  const double *mask2dT;
  const double *areaT;
  int vol_CFL;           /* CS%vol_CFL: CFL numbers and limits from cell areas / open face widths */
} dir_t;

static void dir_setup(dir_t *D, const mom6x_dims *d, const double *G, int dir, int vol_CFL) {
  D->dir = dir;
  D->vol_CFL = vol_CFL;
  D->areaT = GM(G, d, MOM6X_G_areaT);
  D->st = dir ? d->pitch : 1;
  D->Lface = GM(G, d, dir ? MOM6X_G_dx_Cv : MOM6X_G_dy_Cu);
  D->IdT = GM(G, d, dir ? MOM6X_G_IdyT : MOM6X_G_IdxT);
  D->dT = GM(G, d, dir ? MOM6X_G_dyT : MOM6X_G_dxT);
  D->dC = GM(G, d, dir ? MOM6X_G_dyCv : MOM6X_G_dxCu);
  D->maskC = GM(G, d, dir ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu);
  D->IareaT = GM(G, d, MOM6X_G_IareaT);
  D->mask2dT = GM(G, d, MOM6X_G_mask2dT);
}

/* mom6x_continuity_params.sum_order == MOM6X_SUM_TREE16_FMA (include/mom6x.h): fused multiply-adds at FIXED sites -- the Horner
 * chains a + CFL * (p + q * r) of *_flux_layer and *_flux_thickness, u + du * visc_rem, and the masked neighbours and edge values
 * of PPM_reconstruction_x/y -- the same sites as mom6_amd/csrc/continuity_wave.hip (struct Col<MAXL, true>).  Set by
 * orc_continuity_PPM for the duration of the call (read-only inside the OpenMP regions). */
static int orc_fma_mode = 0;
static inline double horner(double a, double CFL, double p, double q, double r) {
  return orc_fma_mode ? fma(CFL, fma(q, r, p), a) : a + CFL * (p + q * r);
}
static inline double vel_cor(double u, double du, double vrem) { return orc_fma_mode ? fma(du, vrem, u) : u + du * vrem; }
static inline double mul_add(double x, double y, double z) { return orc_fma_mode ? fma(x, y, z) : x * y + z; }   /* x*y + z */
static inline double add_mul(double z, double x, double y) { return orc_fma_mode ? fma(x, y, z) : z + x * y; }   /* z + x*y */

/* PPM_limit_pos, MOM_continuity_PPM.F90:2578-2616 */
static void ppm_limit_pos(double h_in, double *h_L, double *h_R, double h_min) {
  double curv = 3.0 * ((*h_L + *h_R) - 2.0 * h_in);
  if (curv > 0.0) { /* Only minima are limited. */
    double dh = *h_R - *h_L;
    if (fabs(dh) < curv) { /* The parabola's minimum is within the cell. */
      if (h_in <= h_min) {
        *h_L = h_in; *h_R = h_in;
      } else if (12.0 * curv * (h_in - h_min) < (curv * curv + 3.0 * (dh * dh))) {
        double scale = 12.0 * curv * (h_in - h_min) / (curv * curv + 3.0 * (dh * dh));
        *h_L = h_in + scale * (*h_L - h_in);
        *h_R = h_in + scale * (*h_R - h_in);
      }
    }
  }
}

/* PPM_limit_CW84, MOM_continuity_PPM.F90:2620-2657 */
static void ppm_limit_cw84(double h_i, double *h_L, double *h_R) {
  if ((*h_R - h_i) * (h_i - *h_L) <= 0.0) {
    *h_L = h_i; *h_R = h_i;
  } else {
    double RLdiff = *h_R - *h_L;
    double RLmean = 0.5 * (*h_R + *h_L);
    double FunFac = 6.0 * RLdiff * (h_i - RLmean);
    double RLdiff2 = RLdiff * RLdiff;
    if (FunFac > RLdiff2) *h_L = 3.0 * h_i - 2.0 * (*h_R);
    if (FunFac < -RLdiff2) *h_R = 3.0 * h_i - 2.0 * (*h_L);
  }
}

/* zonal_edge_thickness :433 / meridional_edge_thickness :476 with PPM_reconstruction_x :2307 /
 * PPM_reconstruction_y :2442 for one layer (2-D slab pointers).  (ish..ieh, jsh..jeh) are the
 * LB bounds; the reconstruction is done one cell beyond them in the sweep direction. */
static void edge_thickness_2d(const mom6x_dims *d, const dir_t *D, const mom6x_continuity_params *CS,
                              const double *h_in, double *h_L, double *h_R, double h_min,
                              int ish, int ieh, int jsh, int jeh, double *slp) {
  const int st = D->st;
  int isl, iel, jsl, jel;
  if (D->dir == 0) { isl = ish - 1; iel = ieh + 1; jsl = jsh; jel = jeh; }
  else             { isl = ish; iel = ieh; jsl = jsh - 1; jel = jeh + 1; }
  const double *m = D->mask2dT;

  if (CS->upwind_1st) { /* :459-463 */
    for (int j = jsl; j <= jel; j++) for (int i = isl; i <= iel; i++) {
      size_t c = IX2(d, i, j);
      h_L[c] = h_in[c]; h_R[c] = h_in[c];
    }
    return;
  }

  if (CS->simple_2nd) { /* :2360-2366 */
    for (int j = jsl; j <= jel; j++) for (int i = isl; i <= iel; i++) {
      size_t c = IX2(d, i, j);
      double h_im1 = mul_add(m[c - st], h_in[c - st], (1.0 - m[c - st]) * h_in[c]);
      double h_ip1 = mul_add(m[c + st], h_in[c + st], (1.0 - m[c + st]) * h_in[c]);
      h_L[c] = 0.5 * (h_im1 + h_in[c]);
      h_R[c] = 0.5 * (h_ip1 + h_in[c]);
    }
  } else {
    /* slopes :2367-2380 on the range extended by one in the sweep direction */
    int is2 = isl, ie2 = iel, js2 = jsl, je2 = jel;
    if (D->dir == 0) { is2 = isl - 1; ie2 = iel + 1; } else { js2 = jsl - 1; je2 = jel + 1; }
    for (int j = js2; j <= je2; j++) for (int i = is2; i <= ie2; i++) {
      size_t c = IX2(d, i, j);
      if ((m[c - st] * m[c] * m[c + st]) == 0.0) {
        slp[c] = 0.0;
      } else {
        double s = 0.5 * (h_in[c + st] - h_in[c - st]);
        double dMx = orc_max(orc_max(h_in[c + st], h_in[c - st]), h_in[c]) - h_in[c];
        double dMn = h_in[c] - orc_min(orc_min(h_in[c + st], h_in[c - st]), h_in[c]);
        slp[c] = orc_sign(1.0, s) * orc_min(fabs(s), 2.0 * orc_min(dMx, dMn));
      }
    }
    const double oneSixth = 1.0 / 6.0;
    for (int j = jsl; j <= jel; j++) for (int i = isl; i <= iel; i++) { /* :2396-2405 */
      size_t c = IX2(d, i, j);
      double h_im1 = mul_add(m[c - st], h_in[c - st], (1.0 - m[c - st]) * h_in[c]);
      double h_ip1 = mul_add(m[c + st], h_in[c + st], (1.0 - m[c + st]) * h_in[c]);
      h_L[c] = add_mul(0.5 * (h_im1 + h_in[c]), oneSixth, slp[c - st] - slp[c]);
      h_R[c] = add_mul(0.5 * (h_ip1 + h_in[c]), oneSixth, slp[c] - slp[c + st]);
    }
  }
  for (int j = jsl; j <= jel; j++) for (int i = isl; i <= iel; i++) { /* :2432-2436 */
    size_t c = IX2(d, i, j);
    if (CS->monotonic) ppm_limit_cw84(h_in[c], &h_L[c], &h_R[c]);
    else ppm_limit_pos(h_in[c], &h_L[c], &h_R[c], h_min);
  }
}

/* zonal_flux_layer :896-971 / merid_flux_layer :1787-1850 for one face.
 * f = flat 2-D index of the face (= its minus cell); h,hL,hR are one layer's slabs. */
static inline void flux_layer_face(const dir_t *D, size_t f, double u, const double *h,
                                   const double *hL, const double *hR, double dt,
                                   double visc_rem, double *uh, double *duhdu) {
  double CFL, curv_3, h_marg;
  const size_t p = f + (size_t)D->st;
  if (u > 0.0) {
    if (D->vol_CFL) CFL = (u * dt) * (D->Lface[f] * D->IareaT[f]);   /* :938 / :1832 */
    else CFL = u * dt * D->IdT[f];
    curv_3 = (hL[f] + hR[f]) - 2.0 * h[f];
    *uh = (D->Lface[f] * 1.0) * u * horner(hR[f], CFL, 0.5 * (hL[f] - hR[f]), curv_3, CFL - 1.5);
    h_marg = horner(hR[f], CFL, hL[f] - hR[f], 3.0 * curv_3, CFL - 1.0);
  } else if (u < 0.0) {
    if (D->vol_CFL) CFL = (-u * dt) * (D->Lface[f] * D->IareaT[p]);   /* :945 / :1840 */
    else CFL = -u * dt * D->IdT[p];
    curv_3 = (hL[p] + hR[p]) - 2.0 * h[p];
    *uh = (D->Lface[f] * 1.0) * u * horner(hL[p], CFL, 0.5 * (hR[p] - hL[p]), curv_3, CFL - 1.5);
    h_marg = horner(hL[p], CFL, hR[p] - hL[p], 3.0 * curv_3, CFL - 1.0);
  } else {
    *uh = 0.0;
    h_marg = 0.5 * (hL[p] + hR[f]);
  }
  *duhdu = (D->Lface[f] * 1.0) * h_marg * visc_rem;
}

/* mom6x_continuity_params.sum_order == MOM6X_SUM_TREE16 (include/mom6x.h): NOT the reference's order.  Sixteen partial
 * sums over k = q, q+16, q+32, ... (each started from +0.0 and walked in increasing k), then a balanced binary tree
 * in q order.  x[k*stride], k = 0..nk-1. */
static double tree16_sum(const double *x, size_t stride, int nk) {
  double s[16];
  for (int q = 0; q < 16; q++) {
    s[q] = 0.0;
    for (int k = q; k < nk; k += 16) s[q] = s[q] + x[(size_t)k * stride];
  }
  for (int w = 1; w < 16; w *= 2)
    for (int q = 0; q < 16; q += 2 * w) s[q] = s[q] + s[q + w];
  return s[0];
}

/* Row description: faces a = a0..a1 of row b. */
typedef struct { int a0, a1, b; } row_t;
static inline size_t row_face(const mom6x_dims *d, const dir_t *D, const row_t *R, int a) {
  (void)D;
  return IX2(d, a, R->b);
}

/* zonal_flux_adjust :1093-1242 / meridional_flux_adjust :1992-2140 for one row of faces.
 * visc_rem is (a, k) -> vr[k*pitch + a] in row-temporary storage (biased so a may be <0).
 * uh_3d may be NULL (as in the call from set_*_BT_cont). */
static void flux_adjust_row(const mom6x_dims *d, const dir_t *D, const mom6x_continuity_params *CS,
                            const row_t *R, const double *u, const double *h_in, const double *hL,
                            const double *hR, const double *uhbt_row, const double *uh_tot_0,
                            const double *duhdu_tot_0, double *du, const double *du_max_CFL,
                            const double *du_min_CFL, double dt, const double *vr,
                            const int *do_I_in, double *uh_3d) {
  const int nz = d->nk, P = d->pitch, max_itts = 20;
  const size_t slab = (size_t)d->slab;
  double *uh_aux = (double *)calloc((size_t)P * nz, sizeof(double)) + d->ioff;
  double *duhdu = (double *)calloc((size_t)P * nz, sizeof(double)) + d->ioff;
  double *uh_err = orc_row_alloc(d), *uh_err_best = orc_row_alloc(d), *duhdu_tot = orc_row_alloc(d);
  double *du_min = orc_row_alloc(d), *du_max = orc_row_alloc(d);
  int *do_I = (int *)calloc((size_t)P, sizeof(int)) + d->ioff;

  if (uh_3d) for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++)
    uh_aux[(size_t)k * P + a] = uh_3d[row_face(d, D, R, a) + k * slab];

  for (int a = R->a0; a <= R->a1; a++) {
    du[a] = 0.0; do_I[a] = do_I_in[a];
    du_max[a] = du_max_CFL[a]; du_min[a] = du_min_CFL[a];
    uh_err[a] = uh_tot_0[a] - uhbt_row[a]; duhdu_tot[a] = duhdu_tot_0[a];
    uh_err_best[a] = fabs(uh_err[a]);
  }

  for (int itt = 1; itt <= max_itts; itt++) {
    double tol_eta;
    if (itt <= 1) tol_eta = 1e-6 * CS->tol_eta;
    else if (itt == 2) tol_eta = 1e-4 * CS->tol_eta;
    else if (itt == 3) tol_eta = 1e-2 * CS->tol_eta;
    else tol_eta = CS->tol_eta;
    const double tol_vel = CS->tol_vel;

    for (int a = R->a0; a <= R->a1; a++) {
      if (uh_err[a] > 0.0) du_max[a] = du[a];
      else if (uh_err[a] < 0.0) du_min[a] = du[a];
      else do_I[a] = 0;
    }
    int domore = 0;
    for (int a = R->a0; a <= R->a1; a++) if (do_I[a]) {
      size_t f = row_face(d, D, R, a);
      if ((dt * orc_min(D->IareaT[f], D->IareaT[f + D->st]) * fabs(uh_err[a]) > tol_eta) ||
          (CS->better_iter && ((fabs(uh_err[a]) > tol_vel * duhdu_tot[a]) ||
                               (fabs(uh_err[a]) > uh_err_best[a])))) {
        /* Newton's method, provided it stays bounded; otherwise bisect. */
        double ddu = -uh_err[a] / duhdu_tot[a];
        double du_prev = du[a];
        du[a] = du[a] + ddu;
        if (fabs(ddu) < 1.0e-15 * fabs(du[a])) {
          do_I[a] = 0;
        } else if (ddu > 0.0) {
          if (du[a] >= du_max[a]) {
            du[a] = 0.5 * (du_prev + du_max[a]);
            if (du_max[a] - du_prev < 1.0e-15 * fabs(du[a])) do_I[a] = 0;
          }
        } else {
          if (du[a] <= du_min[a]) {
            du[a] = 0.5 * (du_prev + du_min[a]);
            if (du_prev - du_min[a] < 1.0e-15 * fabs(du[a])) do_I[a] = 0;
          }
        }
        if (do_I[a]) domore = 1;
      } else {
        do_I[a] = 0;
      }
    }
    if (!domore) break;

    if ((itt < max_itts) || uh_3d) {
      for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++) if (do_I[a]) {
        size_t f = row_face(d, D, R, a);
        double vrem = vr[(size_t)k * P + a];
        double u_new = vel_cor(u[f + k * slab], du[a], vrem);
        flux_layer_face(D, f, u_new, h_in + k * slab, hL + k * slab, hR + k * slab, dt, vrem,
                        &uh_aux[(size_t)k * P + a], &duhdu[(size_t)k * P + a]);
      }
    }

    if ((itt < max_itts) && CS->sum_order == MOM6X_SUM_TREE16) {
      /* (a face whose do_I is false re-sums unchanged layer values: the same bits) */
      for (int a = R->a0; a <= R->a1; a++) {
        uh_err[a] = tree16_sum(&uh_aux[a], (size_t)P, nz) - uhbt_row[a];
        duhdu_tot[a] = tree16_sum(&duhdu[a], (size_t)P, nz);
      }
      for (int a = R->a0; a <= R->a1; a++) uh_err_best[a] = orc_min(uh_err_best[a], fabs(uh_err[a]));
    } else if (itt < max_itts) {
      for (int a = R->a0; a <= R->a1; a++) { uh_err[a] = -uhbt_row[a]; duhdu_tot[a] = 0.0; }
      for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++) {
        uh_err[a] = uh_err[a] + uh_aux[(size_t)k * P + a];
        duhdu_tot[a] = duhdu_tot[a] + duhdu[(size_t)k * P + a];
      }
      for (int a = R->a0; a <= R->a1; a++) uh_err_best[a] = orc_min(uh_err_best[a], fabs(uh_err[a]));
    }
  }

  if (uh_3d) for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++)
    uh_3d[row_face(d, D, R, a) + k * slab] = uh_aux[(size_t)k * P + a];

  free(uh_aux - d->ioff); free(duhdu - d->ioff); free(do_I - d->ioff);
  orc_row_free(d, uh_err); orc_row_free(d, uh_err_best); orc_row_free(d, duhdu_tot);
  orc_row_free(d, du_min); orc_row_free(d, du_max);
}

/* set_zonal_BT_cont :1246-1409 / set_merid_BT_cont :2143-2304 for one row.
 * FA_m0/FA_mm/uBT_mm are the "from the minus side" planes (FA_u_W0, FA_u_WW, uBT_WW |
 * FA_v_S0, FA_v_SS, vBT_SS); FA_p0/FA_pp/uBT_pp the plus side (E | N). */
static void set_BT_cont_row(const mom6x_dims *d, const dir_t *D, const mom6x_continuity_params *CS,
                            const row_t *R, const double *u, const double *h_in, const double *hL,
                            const double *hR, double *FA_m0, double *FA_mm, double *uBT_mm,
                            double *FA_p0, double *FA_pp, double *uBT_pp,
                            const double *uh_tot_0, const double *duhdu_tot_0,
                            const double *du_max_CFL, const double *du_min_CFL, double dt,
                            const double *vr, const double *visc_rem_max, const int *do_I) {
  const int nz = d->nk, P = d->pitch;
  const size_t slab = (size_t)d->slab;
  const double Idt = 1.0 / dt, min_visc_rem = 0.1, CFL_min = 1e-6;
  double *du0 = orc_row_alloc(d), *duL = orc_row_alloc(d), *duR = orc_row_alloc(d);
  double *zeros = orc_row_alloc(d), *du_CFL = orc_row_alloc(d);
  double *FAmt_L = orc_row_alloc(d), *FAmt_R = orc_row_alloc(d), *FAmt_0 = orc_row_alloc(d);
  double *uhtot_L = orc_row_alloc(d), *uhtot_R = orc_row_alloc(d);
  double *t5[5] = {NULL, NULL, NULL, NULL, NULL};   /* sum_order TREE16: the layer values of the five column sums */
  if (CS->sum_order == MOM6X_SUM_TREE16)
    for (int q = 0; q < 5; q++) t5[q] = (double *)calloc((size_t)P * nz, sizeof(double)) + d->ioff;

  flux_adjust_row(d, D, CS, R, u, h_in, hL, hR, zeros, uh_tot_0, duhdu_tot_0, du0, du_max_CFL,
                  du_min_CFL, dt, vr, do_I, NULL);

  int domore = 0;
  for (int a = R->a0; a <= R->a1; a++) {
    size_t f = row_face(d, D, R, a);
    if (do_I[a]) domore = 1;
    du_CFL[a] = (CFL_min * Idt) * D->dC[f];
    duR[a] = orc_min(0.0, du0[a] - du_CFL[a]);
    duL[a] = orc_max(0.0, du0[a] + du_CFL[a]);
    FAmt_L[a] = 0.0; FAmt_R[a] = 0.0; FAmt_0[a] = 0.0; uhtot_L[a] = 0.0; uhtot_R[a] = 0.0;
  }
  if (!domore) {
    for (int a = R->a0; a <= R->a1; a++) {
      size_t f = row_face(d, D, R, a);
      FA_m0[f] = 0.0; FA_mm[f] = 0.0; FA_p0[f] = 0.0; FA_pp[f] = 0.0; uBT_mm[f] = 0.0; uBT_pp[f] = 0.0;
    }
    goto done;
  }

  const int tree = (CS->sum_order == MOM6X_SUM_TREE16);
  for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++) if (do_I[a]) {
    size_t f = row_face(d, D, R, a);
    double vrem = vr[(size_t)k * P + a], uk = u[f + k * slab];
    double visc_rem_lim = orc_max(vrem, min_visc_rem * visc_rem_max[a]);
    if (visc_rem_lim > 0.0) {
      if (tree) {   /* the recurrence below in exact arithmetic: "uk + duR*lim > -du_CFL*vrem" <=> duR > the quotient */
        duR[a] = orc_min(duR[a], -(uk + du_CFL[a] * vrem) / visc_rem_lim);
        duL[a] = orc_max(duL[a], -(uk - du_CFL[a] * vrem) / visc_rem_lim);
        continue;
      }
      if (uk + duR[a] * visc_rem_lim > -du_CFL[a] * vrem)
        duR[a] = -(uk + du_CFL[a] * vrem) / visc_rem_lim;
      if (uk + duL[a] * visc_rem_lim < du_CFL[a] * vrem)
        duL[a] = -(uk - du_CFL[a] * vrem) / visc_rem_lim;
    }
  }

  for (int k = 0; k < nz; k++) for (int a = R->a0; a <= R->a1; a++) if (do_I[a]) {
    size_t f = row_face(d, D, R, a);
    double vrem = vr[(size_t)k * P + a], uk = u[f + k * slab];
    double u_L = vel_cor(uk, duL[a], vrem), u_R = vel_cor(uk, duR[a], vrem), u_0 = vel_cor(uk, du0[a], vrem);
    double uh_0, uh_L, uh_R, duhdu_0, duhdu_L, duhdu_R;
    flux_layer_face(D, f, u_0, h_in + k * slab, hL + k * slab, hR + k * slab, dt, vrem, &uh_0, &duhdu_0);
    flux_layer_face(D, f, u_L, h_in + k * slab, hL + k * slab, hR + k * slab, dt, vrem, &uh_L, &duhdu_L);
    flux_layer_face(D, f, u_R, h_in + k * slab, hL + k * slab, hR + k * slab, dt, vrem, &uh_R, &duhdu_R);
    if (tree) {
      t5[0][(size_t)k * P + a] = duhdu_0; t5[1][(size_t)k * P + a] = duhdu_L; t5[2][(size_t)k * P + a] = duhdu_R;
      t5[3][(size_t)k * P + a] = uh_L; t5[4][(size_t)k * P + a] = uh_R;
      continue;
    }
    FAmt_0[a] = FAmt_0[a] + duhdu_0;
    FAmt_L[a] = FAmt_L[a] + duhdu_L;
    FAmt_R[a] = FAmt_R[a] + duhdu_R;
    uhtot_L[a] = uhtot_L[a] + uh_L;
    uhtot_R[a] = uhtot_R[a] + uh_R;
  }
  if (tree) for (int a = R->a0; a <= R->a1; a++) if (do_I[a]) {
    FAmt_0[a] = tree16_sum(&t5[0][a], (size_t)P, nz); FAmt_L[a] = tree16_sum(&t5[1][a], (size_t)P, nz);
    FAmt_R[a] = tree16_sum(&t5[2][a], (size_t)P, nz); uhtot_L[a] = tree16_sum(&t5[3][a], (size_t)P, nz);
    uhtot_R[a] = tree16_sum(&t5[4][a], (size_t)P, nz);
  }
  for (int a = R->a0; a <= R->a1; a++) {
    size_t f = row_face(d, D, R, a);
    if (do_I[a]) {
      double FA_0 = FAmt_0[a], FA_avg = FAmt_0[a];
      if ((duL[a] - du0[a]) != 0.0) FA_avg = uhtot_L[a] / (duL[a] - du0[a]);
      if (FA_avg > orc_max(FA_0, FAmt_L[a])) FA_avg = orc_max(FA_0, FAmt_L[a]);
      else if (FA_avg < orc_min(FA_0, FAmt_L[a])) FA_0 = FA_avg;
      FA_m0[f] = FA_0; FA_mm[f] = FAmt_L[a];
      if (fabs(FA_0 - FAmt_L[a]) <= 1e-12 * FA_0) uBT_mm[f] = 0.0;
      else uBT_mm[f] = (1.5 * (duL[a] - du0[a])) * ((FAmt_L[a] - FA_avg) / (FAmt_L[a] - FA_0));

      FA_0 = FAmt_0[a]; FA_avg = FAmt_0[a];
      if ((duR[a] - du0[a]) != 0.0) FA_avg = uhtot_R[a] / (duR[a] - du0[a]);
      if (FA_avg > orc_max(FA_0, FAmt_R[a])) FA_avg = orc_max(FA_0, FAmt_R[a]);
      else if (FA_avg < orc_min(FA_0, FAmt_R[a])) FA_0 = FA_avg;
      FA_p0[f] = FA_0; FA_pp[f] = FAmt_R[a];
      if (fabs(FAmt_R[a] - FA_0) <= 1e-12 * FA_0) uBT_pp[f] = 0.0;
      else uBT_pp[f] = (1.5 * (duR[a] - du0[a])) * ((FAmt_R[a] - FA_avg) / (FAmt_R[a] - FA_0));
    } else {
      FA_m0[f] = 0.0; FA_mm[f] = 0.0; FA_p0[f] = 0.0; FA_pp[f] = 0.0; uBT_mm[f] = 0.0; uBT_pp[f] = 0.0;
    }
  }
done:
  for (int q = 0; q < 5; q++) if (t5[q]) free(t5[q] - d->ioff);
  orc_row_free(d, du0); orc_row_free(d, duL); orc_row_free(d, duR); orc_row_free(d, zeros);
  orc_row_free(d, du_CFL); orc_row_free(d, FAmt_L); orc_row_free(d, FAmt_R); orc_row_free(d, FAmt_0);
  orc_row_free(d, uhtot_L); orc_row_free(d, uhtot_R);
}

/* zonal_flux_thickness :975-1089 / merid_flux_thickness :1866-1988 */
static void flux_thickness(const mom6x_dims *d, const dir_t *D, const double *u, const double *h,
                           const double *hL, const double *hR, double *h_u, double dt,
                           int ish, int ieh, int jsh, int jeh, int marginal, const double *visc_rem_u) {
  const int st = D->st;
  int a0, a1, b0, b1;   /* i-range, j-range of faces */
  if (D->dir == 0) { a0 = ish - 1; a1 = ieh; b0 = jsh; b1 = jeh; }
  else             { a0 = ish; a1 = ieh; b0 = jsh - 1; b1 = jeh; }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < d->nk; k++) for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
    size_t f = IX3(d, i, j, k), f2 = IX2(d, i, j), p = f + st;
    double CFL, curv_3, h_avg, h_marg, uf = u[f];
    if (uf > 0.0) {
      if (D->vol_CFL) CFL = (uf * dt) * (D->Lface[f2] * D->IareaT[f2]);   /* :1019 / :1917 */
      else CFL = uf * dt * D->IdT[f2];
      curv_3 = (hL[f] + hR[f]) - 2.0 * h[f];
      h_avg = horner(hR[f], CFL, 0.5 * (hL[f] - hR[f]), curv_3, CFL - 1.5);
      h_marg = horner(hR[f], CFL, hL[f] - hR[f], 3.0 * curv_3, CFL - 1.0);
    } else if (uf < 0.0) {
      if (D->vol_CFL) CFL = (-uf * dt) * (D->Lface[f2] * D->IareaT[f2 + st]);   /* :1025 / :1924 */
      else CFL = -uf * dt * D->IdT[f2 + st];
      curv_3 = (hL[p] + hR[p]) - 2.0 * h[p];
      h_avg = horner(hL[p], CFL, 0.5 * (hR[p] - hL[p]), curv_3, CFL - 1.5);
      h_marg = horner(hL[p], CFL, hR[p] - hL[p], 3.0 * curv_3, CFL - 1.0);
    } else {
      h_avg = 0.5 * (hL[p] + hR[f]);
      h_marg = 0.5 * (hL[p] + hR[f]);
    }
    h_u[f] = marginal ? h_marg : h_avg;
  }
#pragma omp parallel for schedule(static)
  for (int k = 0; k < d->nk; k++) for (int j = b0; j <= b1; j++) for (int i = a0; i <= a1; i++) {
    size_t f = IX3(d, i, j, k);
    if (visc_rem_u) h_u[f] = h_u[f] * (visc_rem_u[f] * 1.0);
    else h_u[f] = h_u[f] * 1.0;
  }
}

/* zonal_mass_flux :519-819 / meridional_mass_flux :1412-1711 */
/* ratio_max :2660-2671 */
static inline double ratio_max(double a, double b, double maxrat) {
  return (fabs(a) > fabs(maxrat * b)) ? maxrat : a / b;
}
/* dx_W, dx_E (dy_S, dy_N) of a face: :651-654 / :1544-1547 */
static inline void face_widths(const dir_t *D, size_t f, double *dx_W, double *dx_E) {
  const size_t p = f + (size_t)D->st;
  if (D->vol_CFL) {
    *dx_W = ratio_max(D->areaT[f], D->Lface[f], 1000.0 * D->dT[f]);
    *dx_E = ratio_max(D->areaT[p], D->Lface[f], 1000.0 * D->dT[p]);
  } else { *dx_W = D->dT[f]; *dx_E = D->dT[p]; }
}

static void mass_flux(const mom6x_dims *d, const dir_t *D, const mom6x_continuity_params *CS,
                      const double *u, const double *h_in, const double *hL, const double *hR,
                      double *uh, double dt, int ish, int ieh, int jsh, int jeh,
                      const double *uhbt, const double *visc_rem_u, double *u_cor,
                      double *FA_m0, double *FA_mm, double *uBT_mm, double *FA_p0, double *FA_pp,
                      double *uBT_pp, double *BT_h_u, int set_BT_cont, double *du_cor) {
  const int nz = d->nk, P = d->pitch;
  const size_t slab = (size_t)d->slab;
  const int use_visc_rem = (visc_rem_u != NULL);
  const double I_dt = 1.0 / dt;
  const double CFL_dt = CS->aggress_adjust ? I_dt : CS->CFL_limit_adjust / dt;   /* :610-612 */
  int b0, b1, a0, a1;
  if (D->dir == 0) { b0 = jsh; b1 = jeh; a0 = ish - 1; a1 = ieh; }
  else             { b0 = jsh - 1; b1 = jeh; a0 = ish; a1 = ieh; }

  if (du_cor) memset(du_cor, 0, sizeof(double) * slab); /* du_cor(:,:) = 0.0 */

  /* rows are independent (the reference: !$OMP parallel do over j, :615 / :1508): every thread has its own row temporaries */
#pragma omp parallel
  {
  double *duhdu = (double *)calloc((size_t)P * nz, sizeof(double)) + d->ioff;
  double *vr = (double *)calloc((size_t)P * nz, sizeof(double)) + d->ioff;
  double *du = orc_row_alloc(d), *du_min_CFL = orc_row_alloc(d), *du_max_CFL = orc_row_alloc(d);
  double *duhdu_tot_0 = orc_row_alloc(d), *uh_tot_0 = orc_row_alloc(d), *visc_rem_max = orc_row_alloc(d);
  double *uhbt_row = orc_row_alloc(d);
  int *do_I = (int *)calloc((size_t)P, sizeof(int)) + d->ioff;

  if (!use_visc_rem) for (int k = 0; k < nz; k++) for (int a = -d->ioff; a < P - d->ioff; a++) vr[(size_t)k * P + a] = 1.0;

#pragma omp for schedule(dynamic, 2)
  for (int b = b0; b <= b1; b++) {
    row_t R = { a0, a1, b };
    for (int a = a0; a <= a1; a++) do_I[a] = 1;
    for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
      size_t f = row_face(d, D, &R, a);
      if (use_visc_rem) vr[(size_t)k * P + a] = visc_rem_u[f + k * slab];
      flux_layer_face(D, f, u[f + k * slab], h_in + k * slab, hL + k * slab, hR + k * slab, dt,
                      vr[(size_t)k * P + a], &uh[f + k * slab], &duhdu[(size_t)k * P + a]);
    }

    if (uhbt || set_BT_cont) {
      if (use_visc_rem && CS->use_visc_rem_max) {
        for (int a = a0; a <= a1; a++) visc_rem_max[a] = 0.0;
        for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++)
          visc_rem_max[a] = orc_max(visc_rem_max[a], vr[(size_t)k * P + a]);
      } else {
        for (int a = a0; a <= a1; a++) visc_rem_max[a] = 1.0;
      }
      /* Set limits on du that will keep the CFL number between -1 and 1. :646-657 */
      for (int a = a0; a <= a1; a++) {
        size_t f = row_face(d, D, &R, a);
        double I_vrm = 0.0;
        if (visc_rem_max[a] > 0.0) I_vrm = 1.0 / visc_rem_max[a];
        double dx_W, dx_E;
        face_widths(D, f, &dx_W, &dx_E);
        du_max_CFL[a] = 2.0 * (CFL_dt * dx_W) * I_vrm;
        du_min_CFL[a] = -2.0 * (CFL_dt * dx_E) * I_vrm;
        uh_tot_0[a] = 0.0; duhdu_tot_0[a] = 0.0;
      }
      if (CS->sum_order == MOM6X_SUM_TREE16) {
        for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a);
          duhdu_tot_0[a] = tree16_sum(&duhdu[a], (size_t)P, nz);
          uh_tot_0[a] = tree16_sum(&uh[f], slab, nz);
        }
      } else
      for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
        size_t f = row_face(d, D, &R, a);
        duhdu_tot_0[a] = duhdu_tot_0[a] + duhdu[(size_t)k * P + a];
        uh_tot_0[a] = uh_tot_0[a] + uh[f + k * slab];
      }
      if (use_visc_rem && CS->aggress_adjust) { /* :664-678 / :1558-1572 */
        for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a), fk = f + k * slab;
          double dx_W, dx_E;
          face_widths(D, f, &dx_W, &dx_E);
          double vrem = vr[(size_t)k * P + a];
          double du_lim = 0.499 * ((dx_W * I_dt - u[fk]) + orc_min(0.0, u[fk - D->st]));
          if (du_max_CFL[a] * vrem > du_lim) du_max_CFL[a] = du_lim / vrem;
          du_lim = 0.499 * ((-dx_E * I_dt - u[fk]) + orc_max(0.0, u[fk + D->st]));
          if (du_min_CFL[a] * vrem < du_lim) du_min_CFL[a] = du_lim / vrem;
        }
      } else if (use_visc_rem) { /* :680-691 */
        for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a);
          double dx_W, dx_E;
          face_widths(D, f, &dx_W, &dx_E);
          double uk = u[f + k * slab], vrem = vr[(size_t)k * P + a];
          if (du_max_CFL[a] * vrem > dx_W * CFL_dt - uk * D->maskC[f])
            du_max_CFL[a] = (dx_W * CFL_dt - uk) / vrem;
          if (du_min_CFL[a] * vrem < -dx_E * CFL_dt - uk * D->maskC[f])
            du_min_CFL[a] = -(dx_E * CFL_dt + uk) / vrem;
        }
      } else if (CS->aggress_adjust) { /* :693-704 */
        for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a), fk = f + k * slab;
          double dx_W, dx_E;
          face_widths(D, f, &dx_W, &dx_E);
          du_max_CFL[a] = orc_min(du_max_CFL[a], 0.499 * ((dx_W * I_dt - u[fk]) + orc_min(0.0, u[fk - D->st])));
          du_min_CFL[a] = orc_max(du_min_CFL[a], 0.499 * ((-dx_E * I_dt - u[fk]) + orc_max(0.0, u[fk + D->st])));
        }
      } else { /* :705-716 */
        for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a);
          double dx_W, dx_E;
          face_widths(D, f, &dx_W, &dx_E);
          double uk = u[f + k * slab];
          du_max_CFL[a] = orc_min(du_max_CFL[a], dx_W * CFL_dt - uk);
          du_min_CFL[a] = orc_max(du_min_CFL[a], -(dx_E * CFL_dt + uk));
        }
      }
      for (int a = a0; a <= a1; a++) {
        du_max_CFL[a] = orc_max(du_max_CFL[a], 0.0);
        du_min_CFL[a] = orc_min(du_min_CFL[a], 0.0);
      }
      for (int a = a0; a <= a1; a++) do_I[a] = 1;

      if (uhbt) {
        for (int a = a0; a <= a1; a++) uhbt_row[a] = uhbt[row_face(d, D, &R, a)];
        flux_adjust_row(d, D, CS, &R, u, h_in, hL, hR, uhbt_row, uh_tot_0, duhdu_tot_0, du,
                        du_max_CFL, du_min_CFL, dt, vr, do_I, uh);
        if (u_cor) for (int k = 0; k < nz; k++) for (int a = a0; a <= a1; a++) {
          size_t f = row_face(d, D, &R, a);
          u_cor[f + k * slab] = vel_cor(u[f + k * slab], du[a], vr[(size_t)k * P + a]);
        }
        if (du_cor) for (int a = a0; a <= a1; a++) du_cor[row_face(d, D, &R, a)] = du[a];
      }
      if (set_BT_cont) {
        set_BT_cont_row(d, D, CS, &R, u, h_in, hL, hR, FA_m0, FA_mm, uBT_mm, FA_p0, FA_pp, uBT_pp,
                        uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL, dt, vr, visc_rem_max, do_I);
      }
    }
  }

  free(duhdu - d->ioff); free(vr - d->ioff); free(do_I - d->ioff);
  orc_row_free(d, du); orc_row_free(d, du_min_CFL); orc_row_free(d, du_max_CFL);
  orc_row_free(d, duhdu_tot_0); orc_row_free(d, uh_tot_0); orc_row_free(d, visc_rem_max);
  orc_row_free(d, uhbt_row);
  }   /* omp parallel */

  if (set_BT_cont && BT_h_u) { /* :802-812 */
    flux_thickness(d, D, (u_cor ? u_cor : u), h_in, hL, hR, BT_h_u, dt, ish, ieh, jsh, jeh,
                   CS->marginal_faces, visc_rem_u);
  }
}

/* continuity_zonal_convergence :348 / continuity_merdional_convergence :386 */
static void convergence(const mom6x_dims *d, const dir_t *D, double *h, const double *uh, double dt,
                        int ish, int ieh, int jsh, int jeh, const double *hin, double h_min) {
  const int st = D->st;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < d->nk; k++) for (int j = jsh; j <= jeh; j++) for (int i = ish; i <= ieh; i++) {
    size_t c = IX3(d, i, j, k), c2 = IX2(d, i, j);
    double h0 = hin ? hin[c] : h[c];
    h[c] = orc_max(h0 - dt * D->IareaT[c2] * (uh[c] - uh[c - st]), h_min);
  }
}

/* continuity_PPM, MOM_continuity_PPM.F90:86-194.  All arrays are HOST pitched arrays. */
int orc_continuity_PPM(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV,
                       const mom6x_continuity_params *CS, int first_direction,
                       const double *u, const double *v, const double *hin, double *h,
                       double *uh, double *vh, double dt,
                       const double *uhbt, const double *vhbt,
                       const double *visc_rem_u, const double *visc_rem_v,
                       double *u_cor, double *v_cor, const mom6x_BT_cont *BT,
                       double *du_cor, double *dv_cor) {
  if (CS->sum_order != MOM6X_SUM_REFERENCE && CS->sum_order != MOM6X_SUM_TREE16 && CS->sum_order != MOM6X_SUM_TREE16_FMA) return MOM6X_EINVAL;
  /* CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL: the device takes its thread-per-column kernels for these, whose column
   * sums run in the reference's order whatever sum_order says (include/mom6x.h) */
  mom6x_continuity_params CS_local = *CS;
  if (CS_local.aggress_adjust || CS_local.vol_CFL) CS_local.sum_order = MOM6X_SUM_REFERENCE;
  orc_fma_mode = (CS_local.sum_order == MOM6X_SUM_TREE16_FMA);   /* the sums of TREE16 + fused multiply-adds at the fixed sites */
  if (orc_fma_mode) CS_local.sum_order = MOM6X_SUM_TREE16;
  CS = &CS_local;
  if ((visc_rem_u != NULL) != (visc_rem_v != NULL)) return MOM6X_EINVAL;
  const size_t n3 = (size_t)d->slab * d->nk;
  double *h_W = (double *)calloc(n3, sizeof(double)), *h_E = (double *)calloc(n3, sizeof(double));
  double *slp = (double *)calloc((size_t)d->slab, sizeof(double));
  const double h_min = GV->Angstrom_H;
  dir_t DX, DY; dir_setup(&DX, d, G, 0, CS->vol_CFL); dir_setup(&DY, d, G, 1, CS->vol_CFL);
  const int x_first = ((first_direction % 2) == 0);
  int stencil = 3; if (CS->simple_2nd) stencil = 2; if (CS->upwind_1st) stencil = 1;
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1;
  const int set_BT = (BT != NULL);

  for (int pass = 0; pass < 2; pass++) {
    const int do_x = (pass == 0) ? x_first : !x_first;
    int ish = is, ieh = ie, jsh = js, jeh = je;
    if (pass == 0) { /* loop bounds that accommodate the subsequent advection in the other direction */
      if (do_x) { jsh = js - stencil; jeh = je + stencil; } else { ish = is - stencil; ieh = ie + stencil; }
    }
    const double *h_src = (pass == 0) ? hin : h;
    const dir_t *D = do_x ? &DX : &DY;
    /* (the reference threads its k loops: !$OMP parallel do, MOM_continuity_PPM.F90:370, :615; every thread its own slope plane) */
#pragma omp parallel
    {
      double *slp_t = (double *)calloc((size_t)d->slab, sizeof(double));
#pragma omp for schedule(static)
      for (int k = 0; k < d->nk; k++)
        edge_thickness_2d(d, D, CS, h_src + (size_t)k * d->slab, h_W + (size_t)k * d->slab,
                          h_E + (size_t)k * d->slab, 2.0 * GV->Angstrom_H, ish, ieh, jsh, jeh, slp_t);
      free(slp_t);
    }
    if (do_x)
      mass_flux(d, D, CS, u, h_src, h_W, h_E, uh, dt, ish, ieh, jsh, jeh, uhbt, visc_rem_u, u_cor,
                set_BT ? BT->FA_u_W0 : NULL, set_BT ? BT->FA_u_WW : NULL, set_BT ? BT->uBT_WW : NULL,
                set_BT ? BT->FA_u_E0 : NULL, set_BT ? BT->FA_u_EE : NULL, set_BT ? BT->uBT_EE : NULL,
                set_BT ? BT->h_u : NULL, set_BT, du_cor);
    else
      mass_flux(d, D, CS, v, h_src, h_W, h_E, vh, dt, ish, ieh, jsh, jeh, vhbt, visc_rem_v, v_cor,
                set_BT ? BT->FA_v_S0 : NULL, set_BT ? BT->FA_v_SS : NULL, set_BT ? BT->vBT_SS : NULL,
                set_BT ? BT->FA_v_N0 : NULL, set_BT ? BT->FA_v_NN : NULL, set_BT ? BT->vBT_NN : NULL,
                set_BT ? BT->h_v : NULL, set_BT, dv_cor);
    if (pass == 0) convergence(d, D, h, do_x ? uh : vh, dt, ish, ieh, jsh, jeh, hin, 0.0);
    else           convergence(d, D, h, do_x ? uh : vh, dt, ish, ieh, jsh, jeh, NULL, h_min);
  }
  free(h_W); free(h_E); free(slp);
  return MOM6X_OK;
}
