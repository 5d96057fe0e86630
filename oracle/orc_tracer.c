/*
 * ORACLE (test infrastructure only; see orc_common.h header).  PARITY UNPINNED.
 *
 * advect_tracer / advect_x / advect_y  <- src/tracer/MOM_tracer_advect.F90:53-350, :355-746, :748-1153
 * triDiagTS, triDiagTS_Eulerian        <- src/parameterizations/vertical/MOM_diabatic_aux.F90:394-488
 * tracer_vertdiff, _Eulerian (core)    <- src/tracer/MOM_tracer_diabatic.F90:25-420 (no surface/bottom flux,
 *                                         no sinking, no reservoir: the plain tridiagonal solve)
 *
 * Schemes: ADVECT_PLM (0), ADVECT_PPMH3 (1), ADVECT_PPM (2) (MOM_tracer_advect_schemes.F90:11-13).
 * No OBC, no diagnostics arrays, conc_underflow = 0, online mode (vol_prev absent).  The row flags
 * domore_u/domore_v and the layer flags domore_k are kept exactly as in the reference, because they decide
 * which rows are touched at all.
 */
#include "orc_common.h"
#include <float.h>

void orc_pass_var(const mom6x_dims *d, double *a, int stagger, int nk);

enum { ADVECT_PLM = 0, ADVECT_PPMH3 = 1, ADVECT_PPM = 2 };

static inline double max3(double a, double b, double c) { return orc_max(orc_max(a, b), c); }
static inline double min3(double a, double b, double c) { return orc_min(orc_min(a, b), c); }

/* One face flux: common to advect_x :546-607 and advect_y :951-1010.  T points at the tracer of the layer,
 * f = face (= minus cell) flat 2-D index, st = stride to the plus cell, maskC = mask2dCu|mask2dCv. */
static double face_flux(int scheme, const double *T, const double *maskC, size_t f, int st, double uhh, double CFL) {
  if (scheme == ADVECT_PPM || scheme == ADVECT_PPMH3) {
    const size_t up = (uhh >= 0.0) ? f : f + st;
    const double Tp = T[up + st], Tc = T[up], Tm = T[up - st];
    double aL, aR;
    if (scheme == ADVECT_PPMH3) {
      aL = (5. * Tc + (2. * Tm - Tp)) / 6.;
      aL = orc_max(orc_min(Tc, Tm), aL); aL = orc_min(orc_max(Tc, Tm), aL);
      aR = (5. * Tc + (2. * Tp - Tm)) / 6.;
      aR = orc_max(orc_min(Tc, Tp), aR); aR = orc_min(orc_max(Tc, Tp), aR);
    } else {
      /* PLM slopes :445-449 at up-1, up, up+1 */
      double sl[3];
      for (int q = -1; q <= 1; q++) {
        const size_t c = up + (size_t)((long)q * st);
        const double tp = T[c + st], tc = T[c], tm = T[c - st];
        const double dMx = max3(tp, tc, tm) - tc, dMn = tc - min3(tp, tc, tm);
        sl[q + 1] = maskC[c] * maskC[c - st] * orc_sign(min3(0.5 * fabs(tp - tm), 2.0 * dMx, 2.0 * dMn), tp - tm);
      }
      aL = 0.5 * ((Tm + Tc) + (sl[0] - sl[1]) / 3.);
      aR = 0.5 * ((Tc + Tp) + (sl[1] - sl[2]) / 3.);
    }
    const double dA = aR - aL, mA = 0.5 * (aR + aL);
    if (maskC[up] * maskC[up - st] * (Tp - Tc) * (Tc - Tm) <= 0.) { aL = Tc; aR = Tc; }
    else if (dA * (Tc - mA) > (dA * dA) / 6.) aL = (3. * Tc) - 2. * aR;
    else if (dA * (Tc - mA) < -(dA * dA) / 6.) aR = (3. * Tc) - 2. * aL;
    const double a6 = 6. * Tc - 3. * (aR + aL);
    if (uhh >= 0.0) return uhh * (aR - 0.5 * CFL * ((aR - aL) - a6 * (1. - 2. / 3. * CFL)));
    return uhh * (aL + 0.5 * CFL * ((aR - aL) + a6 * (1. - 2. / 3. * CFL)));
  } else { /* PLM :585-606 */
    const size_t c = (uhh >= 0.0) ? f : f + st;
    const double tp = T[c + st], tc = T[c], tm = T[c - st];
    const double dMx = max3(tp, tc, tm) - tc, dMn = tc - min3(tp, tc, tm);
    const double slope = maskC[c] * maskC[c - st] * orc_sign(min3(0.5 * fabs(tp - tm), 2.0 * dMx, 2.0 * dMn), tp - tm);
    if (uhh >= 0.0) return uhh * (tc + 0.5 * slope * (1. - CFL));
    return uhh * (tc - 0.5 * slope * (1. - CFL));
  }
}

/* The flux that occurs during this iteration and its CFL number :490-525 / :918-949 */
static void limit_flux(const double *uhr, const double *hprev, const double *areaT, size_t f, int st, double min_h,
                       double tiny_h, double *uhh, double *CFL, int *domore) {
  const double ur = uhr[f];
  if ((ur == 0.0) || ((ur < 0.0) && (hprev[f + st] <= tiny_h)) || ((ur > 0.0) && (hprev[f] <= tiny_h))) {
    *uhh = 0.0; *CFL = 0.0;
  } else if (ur < 0.0) {
    const double hup = hprev[f + st] - areaT[f + st] * min_h;
    const double hlos = orc_max(0.0, uhr[f + st]);
    if ((((hup - hlos) + ur) < 0.0) && ((0.5 * hup + ur) < 0.0)) {
      *uhh = min3(-0.5 * hup, -hup + hlos, 0.0);
      *domore = 1;
    } else *uhh = ur;
    *CFL = -(*uhh) / (hprev[f + st]);
  } else {
    const double hup = hprev[f] - areaT[f] * min_h;
    const double hlos = orc_max(0.0, -uhr[f - st]);
    if ((((hup - hlos) - ur) < 0.0) && ((0.5 * hup - ur) < 0.0)) {
      *uhh = max3(0.5 * hup, hup - hlos, 0.0);
      *domore = 1;
    } else *uhh = ur;
    *CFL = (*uhh) / (hprev[f]);
  }
}

/* advect_x :355-746 for layer k */
static void advect_x(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, double **Tr, const int *scheme, int ntr,
                     double *hprev3, double *uhr3, const double *uh_neglect, int *domore_u /* [nk][nrows] */, int is, int ie,
                     int js, int je, int k) {
  const size_t slab = (size_t)d->slab;
  const int nrows = d->nj + 2 * d->halo + 1, P = d->pitch;
  const double *areaT = GM(G, d, MOM6X_G_areaT), *mCu = GM(G, d, MOM6X_G_mask2dCu);
  const double min_h = 0.1 * GV->Angstrom_H, tiny_h = DBL_MIN, h_neglect = GV->H_subroundoff;
  double *hprev = hprev3 + k * slab, *uhr = uhr3 + k * slab;
  double *uhh = orc_row_alloc(d), *CFL = orc_row_alloc(d), *hlst = orc_row_alloc(d), *Ihnew = orc_row_alloc(d);
  double *flux = (double *)calloc((size_t)P * ntr, sizeof(double)) + d->ioff;
  int *do_i = (int *)calloc((size_t)P, sizeof(int)) + d->ioff;
  for (int j = js; j <= je; j++) {
    int *flag = &domore_u[k * nrows + (j + d->joff)];
    if (!*flag) continue;
    *flag = 0;
    for (int i = is - 1; i <= ie; i++) {
      int more = 0;
      limit_flux(uhr, hprev, areaT, IX2(d, i, j), 1, min_h, tiny_h, &uhh[i], &CFL[i], &more);
      if (more) *flag = 1;
    }
    for (int m = 0; m < ntr; m++) for (int i = is - 1; i <= ie; i++)
      flux[(size_t)m * P + i] = face_flux(scheme[m], Tr[m] + k * slab, mCu, IX2(d, i, j), 1, uhh[i], CFL[i]);
    for (int i = is - 1; i <= ie; i++) { /* :668-671 */
      size_t f = IX2(d, i, j);
      uhr[f] = uhr[f] - uhh[i];
      if (fabs(uhr[f]) < uh_neglect[f]) uhr[f] = 0.0;
    }
    for (int i = is; i <= ie; i++) { /* :672-685 */
      size_t c = IX2(d, i, j);
      if ((uhh[i] != 0.0) || (uhh[i - 1] != 0.0)) {
        do_i[i] = 1;
        hlst[i] = hprev[c];
        hprev[c] = hprev[c] - (uhh[i] - uhh[i - 1]);
        if (hprev[c] <= 0.0) do_i[i] = 0;
        else if (hprev[c] < h_neglect * areaT[c]) {
          hlst[i] = hlst[i] + (h_neglect * areaT[c] - hprev[c]);
          Ihnew[i] = 1.0 / (h_neglect * areaT[c]);
        } else Ihnew[i] = 1.0 / hprev[c];
      } else do_i[i] = 0;
    }
    for (int m = 0; m < ntr; m++) for (int i = is; i <= ie; i++) if (do_i[i]) { /* :704-711 */
      size_t c = IX2(d, i, j) + k * slab;
      if (Ihnew[i] > 0.0) Tr[m][c] = (Tr[m][c] * hlst[i] - (flux[(size_t)m * P + i] - flux[(size_t)m * P + i - 1])) * Ihnew[i];
    }
  }
  orc_row_free(d, uhh); orc_row_free(d, CFL); orc_row_free(d, hlst); orc_row_free(d, Ihnew);
  free(flux - d->ioff); free(do_i - d->ioff);
}

/* advect_y :748-1153 for layer k */
static void advect_y(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, double **Tr, const int *scheme, int ntr,
                     double *hprev3, double *vhr3, const double *vh_neglect, int *domore_v, int is, int ie, int js, int je,
                     int k) {
  const size_t slab = (size_t)d->slab;
  const int nrows = d->nj + 2 * d->halo + 1, st = d->pitch;
  const double *areaT = GM(G, d, MOM6X_G_areaT), *mCv = GM(G, d, MOM6X_G_mask2dCv);
  const double min_h = 0.1 * GV->Angstrom_H, tiny_h = DBL_MIN, h_neglect = GV->H_subroundoff;
  double *hprev = hprev3 + k * slab, *vhr = vhr3 + k * slab;
  double *vhh = (double *)calloc(slab, sizeof(double));
  double *flux = (double *)calloc(slab * ntr, sizeof(double));
  int *do_j_tr = (int *)calloc((size_t)nrows + 8, sizeof(int)) + d->joff + 4;
  int stencil = 1, usePLMslope = 0;
  for (int m = 0; m < ntr; m++) {
    if (scheme[m] == ADVECT_PLM || scheme[m] == ADVECT_PPM) usePLMslope = 1;
    if (scheme[m] == ADVECT_PPM) stencil = 2;
  }
  (void)usePLMslope;
  for (int j = js - 1; j <= je; j++) if (domore_v[k * nrows + (j + d->joff)])
    for (int j2 = 1 - stencil; j2 <= stencil; j2++) do_j_tr[j + j2] = 1;

  for (int j = js - 1; j <= je; j++) {
    int *flag = &domore_v[k * nrows + (j + d->joff)];
    if (*flag) {
      *flag = 0;
      for (int i = is; i <= ie; i++) {
        size_t f = IX2(d, i, j);
        double CFL; int more = 0;
        limit_flux(vhr, hprev, areaT, f, st, min_h, tiny_h, &vhh[f], &CFL, &more);
        if (more) *flag = 1;
        for (int m = 0; m < ntr; m++) flux[(size_t)m * slab + f] = face_flux(scheme[m], Tr[m] + k * slab, mCv, f, st, vhh[f], CFL);
      }
    } else {
      for (int i = is; i <= ie; i++) {
        size_t f = IX2(d, i, j);
        vhh[f] = 0.0;
        for (int m = 0; m < ntr; m++) flux[(size_t)m * slab + f] = 0.0;
      }
    }
  }
  for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) { /* :1073-1076 */
    size_t f = IX2(d, i, j);
    vhr[f] = vhr[f] - vhh[f];
    if (fabs(vhr[f]) < vh_neglect[f]) vhr[f] = 0.0;
  }
  for (int j = js; j <= je; j++) if (do_j_tr[j]) { /* :1080-1125 */
    for (int i = is; i <= ie; i++) {
      size_t c = IX2(d, i, j);
      if ((vhh[c] != 0.0) || (vhh[c - st] != 0.0)) {
        int do_i = 1;
        double hlst = hprev[c], Ihnew = 0.0;
        hprev[c] = orc_max(hprev[c] - (vhh[c] - vhh[c - st]), 0.0);
        if (hprev[c] <= 0.0) do_i = 0;
        else if (hprev[c] < h_neglect * areaT[c]) {
          hlst = hlst + (h_neglect * areaT[c] - hprev[c]);
          Ihnew = 1.0 / (h_neglect * areaT[c]);
        } else Ihnew = 1.0 / hprev[c];
        if (do_i) for (int m = 0; m < ntr; m++) {
          size_t c3 = c + k * slab;
          Tr[m][c3] = (Tr[m][c3] * hlst - (flux[(size_t)m * slab + c] - flux[(size_t)m * slab + c - st])) * Ihnew;
        }
      }
    }
  }
  free(vhh); free(flux); free(do_j_tr - d->joff - 4);
}

/* advect_tracer :53-350.  scheme[m] < 0 selects default_scheme.  x_first_in: -1 = from first_direction. */
int orc_advect_tracer(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, int first_direction, double dt_dyn,
                      int default_scheme, int useHuynhStencilBug, const double *h_end, const double *uhtr,
                      const double *vhtr, double dt, double **Tr, const int *scheme_in, int ntr, int x_first_in,
                      int max_iter_in, double *uhr_out, double *vhr_out, int *iters_out) {
  const int is = 0, ie = d->ni - 1, js = 0, je = d->nj - 1, nz = d->nk, w = d->halo, st = d->pitch;
  const size_t slab = (size_t)d->slab, n3 = slab * nz;
  const int nrows = d->nj + 2 * d->halo + 1;
  const double *areaT = GM(G, d, MOM6X_G_areaT);
  if (ntr == 0) return MOM6X_OK;
  int scheme[32], stencil = 2;
  if (ntr > 32) return MOM6X_EINVAL;
  for (int m = 0; m < ntr; m++) {
    scheme[m] = scheme_in[m] < 0 ? default_scheme : scheme_in[m];
    int sl = 2;
    if (scheme[m] == ADVECT_PPM) sl = 3;
    else if (scheme[m] == ADVECT_PPMH3) sl = useHuynhStencilBug ? 2 : 3;
    if (sl > stencil) stencil = sl;
  }
  if (w < stencil) return MOM6X_EINVAL;
  int x_first = ((first_direction % 2) == 0);
  int max_iter = 2 * (int)ceil(dt / dt_dyn) + 1;
  if (max_iter_in > 0) max_iter = max_iter_in;
  if (x_first_in >= 0) x_first = x_first_in;
  double *hprev = (double *)calloc(n3, sizeof(double)), *uhr = (double *)calloc(n3, sizeof(double)), *vhr = (double *)calloc(n3, sizeof(double));
  double *uh_neglect = (double *)calloc(slab, sizeof(double)), *vh_neglect = (double *)calloc(slab, sizeof(double));
  int *domore_u = (int *)calloc((size_t)nrows * nz, sizeof(int)), *domore_v = (int *)calloc((size_t)nrows * nz, sizeof(int));
  int *domore_k = (int *)calloc((size_t)nz, sizeof(int));
  for (int k = 0; k < nz; k++) { /* :178-206 */
    domore_k[k] = 1;
    for (int j = js; j <= je; j++) for (int i = is - 1; i <= ie; i++) uhr[IX3(d, i, j, k)] = uhtr[IX3(d, i, j, k)];
    for (int j = js - 1; j <= je; j++) for (int i = is; i <= ie; i++) vhr[IX3(d, i, j, k)] = vhtr[IX3(d, i, j, k)];
    for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) {
      size_t c = IX3(d, i, j, k), c2 = IX2(d, i, j);
      hprev[c] = orc_max(0.0, areaT[c2] * h_end[c] + ((uhr[c] - uhr[c - 1]) + (vhr[c] - vhr[c - st])));
      hprev[c] = hprev[c] + orc_max(0.0, 1.0e-13 * hprev[c] - areaT[c2] * h_end[c]);
    }
  }
  for (int j = -w; j <= d->nj - 1 + w; j++) for (int i = -w; i <= d->ni - 2 + w; i++) {
    size_t c = IX2(d, i, j);
    uh_neglect[c] = GV->H_subroundoff * orc_min(areaT[c], areaT[c + 1]);
  }
  for (int j = -w; j <= d->nj - 2 + w; j++) for (int i = -w; i <= d->ni - 1 + w; i++) {
    size_t c = IX2(d, i, j);
    vh_neglect[c] = GV->H_subroundoff * orc_min(areaT[c], areaT[c + st]);
  }
  int isv = is, iev = ie, jsv = js, jev = je, itt;
  for (itt = 1; itt <= max_iter; itt++) {
    if (isv > is - stencil) { /* :229-262 */
      orc_pass_var(d, uhr, 1, nz); orc_pass_var(d, vhr, 2, nz); orc_pass_var(d, hprev, 0, nz);
      for (int m = 0; m < ntr; m++) orc_pass_var(d, Tr[m], 0, nz);
      const int nsten_halo = w / stencil;
      isv = is - nsten_halo * stencil; jsv = js - nsten_halo * stencil;
      iev = ie + nsten_halo * stencil; jev = je + nsten_halo * stencil;
      if ((nsten_halo > 1) || (itt == 1)) {
        for (int k = 0; k < nz; k++) if (domore_k[k] > 0) {
          for (int j = jsv; j <= jev; j++) if (!domore_u[k * nrows + j + d->joff]) {
            for (int i = isv + stencil - 1; i <= iev - stencil; i++) if (uhr[IX3(d, i, j, k)] != 0.0) { domore_u[k * nrows + j + d->joff] = 1; break; }
          }
          for (int j = jsv + stencil - 1; j <= jev - stencil; j++) if (!domore_v[k * nrows + j + d->joff]) {
            for (int i = isv + stencil; i <= iev - stencil; i++) if (vhr[IX3(d, i, j, k)] != 0.0) { domore_v[k * nrows + j + d->joff] = 1; break; }
          }
          domore_k[k] = 0;
          for (int j = jsv; j <= jev; j++) if (domore_u[k * nrows + j + d->joff]) domore_k[k] = 1;
          for (int j = jsv + stencil - 1; j <= jev - stencil; j++) if (domore_v[k * nrows + j + d->joff]) domore_k[k] = 1;
        }
      }
    }
    isv = isv + stencil; iev = iev - stencil; jsv = jsv + stencil; jev = jev - stencil;
    if (x_first) {
      for (int k = 0; k < nz; k++) if (domore_k[k] > 0)
        advect_x(d, G, GV, Tr, scheme, ntr, hprev, uhr, uh_neglect, domore_u, isv, iev, jsv - stencil, jev + stencil, k);
      for (int k = 0; k < nz; k++) if (domore_k[k] > 0) {
        advect_y(d, G, GV, Tr, scheme, ntr, hprev, vhr, vh_neglect, domore_v, isv, iev, jsv, jev, k);
        domore_k[k] = 0;
        for (int j = jsv - stencil; j <= jev + stencil; j++) if (domore_u[k * nrows + j + d->joff]) domore_k[k] = 1;
        for (int j = jsv - 1; j <= jev; j++) if (domore_v[k * nrows + j + d->joff]) domore_k[k] = 1;
      }
    } else {
      for (int k = 0; k < nz; k++) if (domore_k[k] > 0)
        advect_y(d, G, GV, Tr, scheme, ntr, hprev, vhr, vh_neglect, domore_v, isv - stencil, iev + stencil, jsv, jev, k);
      for (int k = 0; k < nz; k++) if (domore_k[k] > 0) {
        advect_x(d, G, GV, Tr, scheme, ntr, hprev, uhr, uh_neglect, domore_u, isv, iev, jsv, jev, k);
        domore_k[k] = 0;
        for (int j = jsv; j <= jev; j++) if (domore_u[k * nrows + j + d->joff]) domore_k[k] = 1;
        for (int j = jsv - 1; j <= jev; j++) if (domore_v[k * nrows + j + d->joff]) domore_k[k] = 1;
      }
    }
    if (itt >= max_iter) break;
    if (isv > is - stencil) {
      int do_any = 0;
      for (int k = 0; k < nz; k++) do_any += domore_k[k];
      if (do_any == 0) break;
    }
  }
  if (iters_out) *iters_out = itt;
  if (uhr_out) memcpy(uhr_out, uhr, n3 * sizeof(double));
  if (vhr_out) memcpy(vhr_out, vhr, n3 * sizeof(double));
  free(hprev); free(uhr); free(vhr); free(uh_neglect); free(vh_neglect); free(domore_u); free(domore_v); free(domore_k);
  return MOM6X_OK;
}

/* triDiagTS :394-440: T,S solved together in the reference; one field at a time here (identical arithmetic). */
int orc_triDiagTS(const mom6x_dims *d, const double *hold, const double *ea, const double *eb, double *T,
                  double h_neglect) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  double *c1 = (double *)calloc((size_t)nz, sizeof(double));
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t x = IX2(d, i, j);
    double h_tr = hold[x] + h_neglect;
    double b1 = 1.0 / (h_tr + eb[x]);
    double d1 = h_tr * b1;
    T[x] = (b1 * h_tr) * T[x];
    for (int k = 1; k < nz; k++) {
      size_t c = x + k * slab;
      c1[k] = eb[c - slab] * b1;
      h_tr = hold[c] + h_neglect;
      double b_denom_1 = h_tr + d1 * ea[c];
      b1 = 1.0 / (b_denom_1 + eb[c]);
      d1 = b_denom_1 * b1;
      T[c] = b1 * (h_tr * T[c] + ea[c] * T[c - slab]);
    }
    for (int k = nz - 2; k >= 0; k--) T[x + k * slab] = T[x + k * slab] + c1[k + 1] * T[x + (k + 1) * slab];
  }
  free(c1);
  return MOM6X_OK;
}

/* triDiagTS_Eulerian :444-488: ent has nz+1 interfaces */
int orc_triDiagTS_Eulerian(const mom6x_dims *d, const double *hold, const double *ent, double *T, double h_neglect) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  double *c1 = (double *)calloc((size_t)nz, sizeof(double));
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t x = IX2(d, i, j);
    double h_tr = hold[x] + h_neglect;
    double b1 = 1.0 / (h_tr + ent[x + slab]);
    double d1 = h_tr * b1;
    T[x] = (b1 * h_tr) * T[x];
    for (int k = 1; k < nz; k++) {
      size_t c = x + k * slab;
      c1[k] = ent[c] * b1;
      h_tr = hold[c] + h_neglect;
      double b_denom_1 = h_tr + d1 * ent[c];
      b1 = 1.0 / (b_denom_1 + ent[c + slab]);
      d1 = b_denom_1 * b1;
      T[c] = b1 * (h_tr * T[c] + ent[c] * T[c - slab]);
    }
    for (int k = nz - 2; k >= 0; k--) T[x + k * slab] = T[x + k * slab] + c1[k + 1] * T[x + (k + 1) * slab];
  }
  free(c1);
  return MOM6X_OK;
}

/* tracer_vertdiff :25-217 (no sink_rate / btm_reservoir branch :181-214) and, with ea = ent(:,:,K),
 * eb = ent(:,:,K+1), tracer_vertdiff_Eulerian :224-420 (no-sink branch :382-414). */
int orc_tracer_vertdiff(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h_old,
                        const double *ea, const double *eb, double dt, double *tr, const double *sfc_flux,
                        const double *btm_flux, int convert_flux) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT);
  const double h_neglect = GV->H_subroundoff;
  if (nz == 1) return MOM6X_OK;
  double *c1 = (double *)calloc((size_t)nz, sizeof(double));
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t x = IX2(d, i, j);
    if (!(mT[x] > 0.0)) continue;
    double sfc_src = 0.0, btm_src = 0.0;
    if (sfc_flux) sfc_src = convert_flux ? (sfc_flux[x] * dt) * GV->RZ_to_H : sfc_flux[x];
    if (btm_flux) btm_src = convert_flux ? (btm_flux[x] * dt) * GV->RZ_to_H : btm_flux[x];
    double h_tr = h_old[x] + h_neglect;
    double b_denom_1 = h_tr + ea[x];
    double b1 = 1.0 / (b_denom_1 + eb[x]);
    double d1 = h_tr * b1;
    tr[x] = (b1 * h_tr) * tr[x] + b1 * sfc_src;
    for (int k = 1; k < nz - 1; k++) {
      size_t c = x + k * slab;
      c1[k] = eb[c - slab] * b1;
      h_tr = h_old[c] + h_neglect;
      b_denom_1 = h_tr + d1 * ea[c];
      b1 = 1.0 / (b_denom_1 + eb[c]);
      d1 = b_denom_1 * b1;
      tr[c] = b1 * (h_tr * tr[c] + ea[c] * tr[c - slab]);
    }
    {
      size_t c = x + (size_t)(nz - 1) * slab;
      c1[nz - 1] = eb[c - slab] * b1;
      h_tr = h_old[c] + h_neglect;
      b_denom_1 = h_tr + d1 * ea[c];
      b1 = 1.0 / (b_denom_1 + eb[c]);
      tr[c] = b1 * ((h_tr * tr[c] + btm_src) + ea[c] * tr[c - slab]);
    }
    for (int k = nz - 2; k >= 0; k--) tr[x + k * slab] = tr[x + k * slab] + c1[k + 1] * tr[x + (k + 1) * slab];
  }
  free(c1);
  return MOM6X_OK;
}


/* tracer_vertdiff WITH sink_rate (:123-179) and, with ea = ent(:,:,K), eb = ent(:,:,K+1), tracer_vertdiff_Eulerian's sinking branch
 * (:315-380): the sinking distances at the interfaces, limited so that characteristics do not cross within the step (or, with a
 * bottom reservoir, unlimited: what leaves the bottom layer is collected in btm_reservoir), then the tridiagonal solve with the
 * sinking flux on the lower diagonal.  btm_reservoir may be NULL (not present).  TEST INFRASTRUCTURE like the rest of oracle/. */
int orc_tracer_vertdiff_sink(const mom6x_dims *d, const double *G, const mom6x_vgrid *GV, const double *h_old,
                             const double *ea, const double *eb, double dt, double *tr, const double *sfc_flux,
                             const double *btm_flux, double *btm_reservoir, double sink_rate, int convert_flux) {
  const int nz = d->nk;
  const size_t slab = (size_t)d->slab;
  const double *mT = GM(G, d, MOM6X_G_mask2dT);
  const double h_neglect = GV->H_subroundoff;
  if (nz == 1) return MOM6X_OK;
  const double sink_dist = (dt * sink_rate) * GV->Z_to_H;
  double *c1 = (double *)calloc((size_t)nz + 1, sizeof(double));
  double *sink = (double *)calloc((size_t)nz + 2, sizeof(double));     /* sink[K], K = 0 .. nz (interface K is the top of layer K) */
  double *hmd = (double *)calloc((size_t)nz + 1, sizeof(double));      /* h_minus_dsink */
  for (int j = 0; j < d->nj; j++) for (int i = 0; i < d->ni; i++) {
    size_t x = IX2(d, i, j);
    double sfc_src = 0.0, btm_src = 0.0;
    if (sfc_flux) sfc_src = convert_flux ? (sfc_flux[x] * dt) * GV->RZ_to_H : sfc_flux[x];
    if (btm_flux) btm_src = convert_flux ? (btm_flux[x] * dt) * GV->RZ_to_H : btm_flux[x];
    if (btm_reservoir) {
      sink[nz] = sink_dist;
      for (int k = 1; k < nz; k++) { sink[k] = sink_dist; hmd[k] = h_old[x + k * slab]; }
    } else {
      sink[nz] = 0.0;
      for (int k = nz - 1; k >= 1; k--) {
        const double h = h_old[x + k * slab];
        if (sink[k + 1] >= sink_dist) {
          sink[k] = sink_dist;
          hmd[k] = h + (sink[k + 1] - sink[k]);
        } else if (sink[k + 1] + h < sink_dist) {
          sink[k] = sink[k + 1] + h;
          hmd[k] = 0.0;
        } else {
          sink[k] = sink_dist;
          hmd[k] = (h + sink[k + 1]) - sink[k];
        }
      }
    }
    sink[0] = 0.0; hmd[0] = (h_old[x] + sink[1]);
    if (!(mT[x] > 0.0)) continue;
    double b_denom_1 = hmd[0] + ea[x] + h_neglect;
    double b1 = 1.0 / (b_denom_1 + eb[x]);
    double d1 = b_denom_1 * b1;
    double h_tr = h_old[x] + h_neglect;
    tr[x] = (b1 * h_tr) * tr[x] + b1 * sfc_src;
    for (int k = 1; k < nz - 1; k++) {
      size_t c = x + k * slab;
      c1[k] = eb[c - slab] * b1;
      b_denom_1 = hmd[k] + d1 * (ea[c] + sink[k]) + h_neglect;
      b1 = 1.0 / (b_denom_1 + eb[c]);
      d1 = b_denom_1 * b1;
      h_tr = h_old[c] + h_neglect;
      tr[c] = b1 * (h_tr * tr[c] + (ea[c] + sink[k]) * tr[c - slab]);
    }
    {
      const int k = nz - 1;
      size_t c = x + (size_t)k * slab;
      c1[k] = eb[c - slab] * b1;
      b_denom_1 = hmd[k] + d1 * (ea[c] + sink[k]) + h_neglect;
      b1 = 1.0 / (b_denom_1 + eb[c]);
      h_tr = h_old[c] + h_neglect;
      tr[c] = b1 * ((h_tr * tr[c] + btm_src) + (ea[c] + sink[k]) * tr[c - slab]);
      if (btm_reservoir) btm_reservoir[x] = btm_reservoir[x] + (sink[nz] * tr[c]) * GV->H_to_RZ;
    }
    for (int k = nz - 2; k >= 0; k--) tr[x + k * slab] = tr[x + k * slab] + c1[k + 1] * tr[x + (k + 1) * slab];
  }
  free(c1); free(sink); free(hmd);
  return MOM6X_OK;
}
