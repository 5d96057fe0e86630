// continuity_dev.h -- device helpers shared by the continuity kernels (continuity.hip, continuity_lds.hip).
#pragma once
#include "mom6x_dev.h"

struct DirMetrics {
  const double *Lface, *IdT, *dT, *dC, *maskC, *IareaT, *mask2dT, *areaT;
  int vol_CFL;   // CONT_PPM_VOLUME_BASED_CFL (the thread-per-column kernels of continuity.hip only)
};

template <int DIR>
__device__ __forceinline__ DirMetrics dir_metrics(const double *G, const Dm &d) {
  DirMetrics D;
  D.Lface = gm(G, d, DIR ? MOM6X_G_dx_Cv : MOM6X_G_dy_Cu);
  D.IdT = gm(G, d, DIR ? MOM6X_G_IdyT : MOM6X_G_IdxT);
  D.dT = gm(G, d, DIR ? MOM6X_G_dyT : MOM6X_G_dxT);
  D.dC = gm(G, d, DIR ? MOM6X_G_dyCv : MOM6X_G_dxCu);
  D.maskC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu);
  D.IareaT = gm(G, d, MOM6X_G_IareaT);
  D.mask2dT = gm(G, d, MOM6X_G_mask2dT);
  D.areaT = gm(G, d, MOM6X_G_areaT);
  D.vol_CFL = 0;
  return D;
}

// PPM_limit_pos :2578-2616
__device__ __forceinline__ void ppm_limit_pos(double h_in, double &h_L, double &h_R, double h_min) {
  const double curv = 3.0 * ((h_L + h_R) - 2.0 * h_in);
  if (curv > 0.0) {
    const double dh = h_R - h_L;
    if (fabs(dh) < curv) {
      if (h_in <= h_min) {
        h_L = h_in; h_R = h_in;
      } else if (12.0 * curv * (h_in - h_min) < (curv * curv + 3.0 * (dh * dh))) {
        const double scale = 12.0 * curv * (h_in - h_min) / (curv * curv + 3.0 * (dh * dh));
        h_L = h_in + scale * (h_L - h_in);
        h_R = h_in + scale * (h_R - h_in);
      }
    }
  }
}

// PPM_limit_CW84 :2620-2657
__device__ __forceinline__ void ppm_limit_cw84(double h_i, double &h_L, double &h_R) {
  if ((h_R - h_i) * (h_i - h_L) <= 0.0) {
    h_L = h_i; h_R = h_i;
  } else {
    const double RLdiff = h_R - h_L;
    const double RLmean = 0.5 * (h_R + h_L);
    const double FunFac = 6.0 * RLdiff * (h_i - RLmean);
    const double RLdiff2 = RLdiff * RLdiff;
    if (FunFac > RLdiff2) h_L = 3.0 * h_i - 2.0 * h_R;
    if (FunFac < -RLdiff2) h_R = 3.0 * h_i - 2.0 * h_L;
  }
}

// Lin (1994) B2 limited slope at cell c (:2368-2378)
__device__ __forceinline__ double ppm_slope(const double *h, const double *m, size_t c, int st) {
  const double hm = h[c - st], h0 = h[c], hp = h[c + st];
  if ((m[c - st] * m[c] * m[c + st]) == 0.0) return 0.0;
  const double s = 0.5 * (hp - hm);
  const double dMx = dmax(dmax(hp, hm), h0) - h0;
  const double dMn = h0 - dmin(dmin(hp, hm), h0);
  return dsign(1.0, s) * dmin(fabs(s), 2.0 * dmin(dMx, dMn));
}

// Lin (1994) B2 limited slope (:2368-2378) from registers; mprod = product of the three masks.
__device__ __forceinline__ double slope3(double hm, double h0, double hp, double mprod) {
  if (mprod == 0.0) return 0.0;
  const double s = 0.5 * (hp - hm);
  const double dMx = dmax(dmax(hp, hm), h0) - h0;
  const double dMn = h0 - dmin(dmin(hp, hm), h0);
  return dsign(1.0, s) * dmin(fabs(s), 2.0 * dmin(dMx, dMn));
}

// PPM_reconstruction_x/y + limiter for one cell from its 5-point stencil hh[0..4] (cell = hh[2]).
// FMA (continuity_wave.hip, sum_order == MOM6X_SUM_TREE16_FMA): the masked neighbours and the edge values with fused multiply-adds.
template <bool FMA = false>
__device__ __forceinline__ void edge5(const double *hh, const double *mm, int scheme, int monotonic, double h_min,
                                      double &hl, double &hr, double &c3) {
  const double h0 = hh[2];
  if (scheme == 2) {
    hl = h0; hr = h0;
  } else {
    const double h_im1 = FMA ? fma(mm[1], hh[1], (1.0 - mm[1]) * h0) : mm[1] * hh[1] + (1.0 - mm[1]) * h0;
    const double h_ip1 = FMA ? fma(mm[3], hh[3], (1.0 - mm[3]) * h0) : mm[3] * hh[3] + (1.0 - mm[3]) * h0;
    if (scheme == 1) {
      hl = 0.5 * (h_im1 + h0);
      hr = 0.5 * (h_ip1 + h0);
    } else {
      const double oneSixth = 1.0 / 6.0;
      const double sm = slope3(hh[0], hh[1], hh[2], mm[0] * mm[1] * mm[2]);
      const double s0 = slope3(hh[1], hh[2], hh[3], mm[1] * mm[2] * mm[3]);
      const double sp = slope3(hh[2], hh[3], hh[4], mm[2] * mm[3] * mm[4]);
      hl = FMA ? fma(oneSixth, sm - s0, 0.5 * (h_im1 + h0)) : 0.5 * (h_im1 + h0) + oneSixth * (sm - s0);
      hr = FMA ? fma(oneSixth, s0 - sp, 0.5 * (h_ip1 + h0)) : 0.5 * (h_ip1 + h0) + oneSixth * (s0 - sp);
    }
    if (monotonic) ppm_limit_cw84(h0, hl, hr);
    else ppm_limit_pos(h0, hl, hr, h_min);
  }
  c3 = (hl + hr) - 2.0 * h0;
}

// zonal_flux_layer :896 / merid_flux_layer :1787 for one face of one layer.
// f = flat 3-D index of the face (= its minus cell), f2 = its 2-D index.
__device__ __forceinline__ void flux_layer(const DirMetrics &D, int st, size_t f, size_t f2, double u,
                                           const double *__restrict__ h, const double *__restrict__ hL,
                                           const double *__restrict__ hR, double dt, double visc_rem,
                                           double Lf, double &uh, double &duhdu) {
  double h_marg;
  if (u > 0.0) {
    const double CFL = D.vol_CFL ? (u * dt) * (D.Lface[f2] * D.IareaT[f2]) : u * dt * D.IdT[f2];   // :938 / :1832
    const double l = hL[f], r = hR[f];
    const double curv_3 = (l + r) - 2.0 * h[f];
    uh = Lf * u * (r + CFL * (0.5 * (l - r) + curv_3 * (CFL - 1.5)));
    h_marg = r + CFL * ((l - r) + 3.0 * curv_3 * (CFL - 1.0));
  } else if (u < 0.0) {
    const size_t p = f + st;
    const double CFL = D.vol_CFL ? (-u * dt) * (D.Lface[f2] * D.IareaT[f2 + st]) : -u * dt * D.IdT[f2 + st];   // :945 / :1840
    const double l = hL[p], r = hR[p];
    const double curv_3 = (l + r) - 2.0 * h[p];
    uh = Lf * u * (l + CFL * (0.5 * (r - l) + curv_3 * (CFL - 1.5)));
    h_marg = l + CFL * ((r - l) + 3.0 * curv_3 * (CFL - 1.0));
  } else {
    uh = 0.0;
    h_marg = 0.5 * (hL[f + st] + hR[f]);
  }
  duhdu = Lf * h_marg * visc_rem;
}

struct FluxArgs {
  const double *u, *h_in, *hL, *hR;
  double *uh;
  const double *uhbt;        // 2-D or null
  const double *visc_rem;    // 3-D or null
  double *u_cor;             // 3-D or null
  double *du_cor;            // 2-D or null
  // BT_cont planes for this direction ("m" = from the minus side: W|S, "p" = plus side: E|N)
  double *FA_m0, *FA_mm, *uBT_mm, *FA_p0, *FA_pp, *uBT_pp;
  int set_BT_cont;
  double dt, CFL_limit_adjust, tol_eta, tol_vel;
  int better_iter, use_visc_rem_max;
  int aggress_adjust, vol_CFL;   // CONT_PPM_AGGRESS_ADJUST, CONT_PPM_VOLUME_BASED_CFL: carried by k_mass_flux (continuity.hip) only
  int a0, a1, b0, b1;        // face index ranges (i-range, j-range)
};

