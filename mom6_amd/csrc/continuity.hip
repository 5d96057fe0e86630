// continuity.hip -- PPM layer continuity with barotropic flux reconciliation on gfx950.
//
// Replaces continuity_PPM and its callees (MOM_continuity_PPM.F90:86-2657).  The zonal and
// meridional halves of the reference are mirror images; one templated kernel set handles both
// with DIR = 0 (faces between cell f and f+1) or DIR = 1 (faces between f and f+pitch).  In
// both directions the lane index is i, so every access is coalesced along the contiguous axis.
//
// Kernels (all FP64, HBM-bandwidth bound):
//   k_edge<DIR>        PPM_reconstruction_x/y + PPM_limit_pos/CW84   (3-D, one thread per cell)
//   k_mass_flux<DIR>   zonal/meridional_mass_flux incl. flux_adjust Newton iteration and
//                      set_*_BT_cont                                  (2-D, one thread per face
//                      column; the k loops are walked sequentially in the reference's order so
//                      the column sums are bit-identical to the Fortran loop nests)
//   k_flux_thickness<DIR>  zonal/merid_flux_thickness                 (3-D)
//   k_convergence<DIR> continuity_zonal/merdional_convergence         (3-D)
#include "continuity_dev.h"
#include "continuity_lds.h"
#include <cstdlib>

namespace {

// MOM6X_MASSFLUX=legacy selects the thread-per-column kernels (k_edge, k_mass_flux, k_flux_thickness) instead
// of the LDS-resident kernel of continuity_lds.hip; both give bit-identical results (tests run both).
bool use_lds_path(int nk) {
  const char *e = getenv("MOM6X_MASSFLUX");
  if (e && !strcmp(e, "legacy")) return false;
  return mass_flux_lds_usable(nk);
}


// PPM_reconstruction_x :2307 / _y :2442 over cells (i0..i1, j0..j1) of every layer.
template <int DIR>
__global__ void __launch_bounds__(256)
k_edge(Dm d, const double *__restrict__ G, const double *__restrict__ h_in, double *__restrict__ h_L,
       double *__restrict__ h_R, double h_min, int scheme /*0 ppm,1 simple_2nd,2 upwind*/, int monotonic,
       int i0, int i1, int j0, int j1) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > i1 || j > j1) return;
  const int st = DIR ? d.pitch : 1;
  const double *m = gm(G, d, MOM6X_G_mask2dT);
  const size_t c2 = ix2(d, i, j);
  const double *h = h_in + (size_t)k * d.slab;
  const size_t c = c2 + (size_t)k * d.slab;
  const double h0 = h[c2];
  double hl, hr;
  if (scheme == 2) {
    hl = h0; hr = h0;
    h_L[c] = hl; h_R[c] = hr;
    return;
  }
  const double mm = m[c2 - st], mp = m[c2 + st];
  const double h_im1 = mm * h[c2 - st] + (1.0 - mm) * h0;
  const double h_ip1 = mp * h[c2 + st] + (1.0 - mp) * h0;
  if (scheme == 1) {
    hl = 0.5 * (h_im1 + h0);
    hr = 0.5 * (h_ip1 + h0);
  } else {
    const double oneSixth = 1.0 / 6.0;
    const double sm = ppm_slope(h, m, c2 - st, st);
    const double s0 = ppm_slope(h, m, c2, st);
    const double sp = ppm_slope(h, m, c2 + st, st);
    hl = 0.5 * (h_im1 + h0) + oneSixth * (sm - s0);
    hr = 0.5 * (h_ip1 + h0) + oneSixth * (s0 - sp);
  }
  if (monotonic) ppm_limit_cw84(h0, hl, hr);
  else ppm_limit_pos(h0, hl, hr, h_min);
  h_L[c] = hl; h_R[c] = hr;
}

struct ColIn {   // everything a face column needs
  const double *u, *h, *hL, *hR, *vr;   // vr may be null (visc_rem == 1)
  double dt;
  int nk; size_t slab;
};

// zonal_flux_adjust :1093-1242 / meridional_flux_adjust :1992-2140 for ONE face column.
// The reference iterates a whole row with a row-wide `domore`, but a face whose do_I is false is
// frozen, so iterating each face until its own do_I clears gives identical results.
// If uh_store != nullptr the layer transports are (re)written there (uh_3d present).
template <int DIR>
__device__ double flux_adjust(const DirMetrics &D, int st, const ColIn &C, size_t f2, double Lf,
                              double uhbt, double uh_tot_0, double duhdu_tot_0, double du_max_CFL,
                              double du_min_CFL, double tol_eta_cs, double tol_vel, int better_iter,
                              bool do_I, double *uh_store) {
  const int max_itts = 20;
  double du = 0.0, du_max = du_max_CFL, du_min = du_min_CFL;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0;
  double uh_err_best = fabs(uh_err);
  const double IareaMin = dmin(D.IareaT[f2], D.IareaT[f2 + st]);
  for (int itt = 1; itt <= max_itts; itt++) {
    double tol_eta;
    if (itt <= 1) tol_eta = 1e-6 * tol_eta_cs;
    else if (itt == 2) tol_eta = 1e-4 * tol_eta_cs;
    else if (itt == 3) tol_eta = 1e-2 * tol_eta_cs;
    else tol_eta = tol_eta_cs;

    if (uh_err > 0.0) du_max = du;
    else if (uh_err < 0.0) du_min = du;
    else do_I = false;

    if (do_I) {
      if ((C.dt * IareaMin * fabs(uh_err) > tol_eta) ||
          (better_iter && ((fabs(uh_err) > tol_vel * duhdu_tot) || (fabs(uh_err) > uh_err_best)))) {
        const double ddu = -uh_err / duhdu_tot;
        const double du_prev = du;
        du = du + ddu;
        if (fabs(ddu) < 1.0e-15 * fabs(du)) {
          do_I = false;
        } else if (ddu > 0.0) {
          if (du >= du_max) {
            du = 0.5 * (du_prev + du_max);
            if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
          }
        } else {
          if (du <= du_min) {
            du = 0.5 * (du_prev + du_min);
            if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
          }
        }
      } else {
        do_I = false;
      }
    }
    if (!do_I) break;

    if ((itt < max_itts) || uh_store) {
      double err = -uhbt, dtot = 0.0;
      for (int k = 0; k < C.nk; k++) {
        const size_t f = f2 + (size_t)k * C.slab;
        const double vrem = C.vr ? C.vr[f] : 1.0;
        const double u_new = C.u[f] + du * vrem;
        double uh, duhdu;
        flux_layer(D, st, f, f2, u_new, C.h, C.hL, C.hR, C.dt, vrem, Lf, uh, duhdu);
        if (uh_store) uh_store[f] = uh;
        err = err + uh;
        dtot = dtot + duhdu;
      }
      if (itt < max_itts) {
        uh_err = err; duhdu_tot = dtot;
        uh_err_best = dmin(uh_err_best, fabs(uh_err));
      }
    }
  }
  return du;
}

// zonal_mass_flux :519-819 / meridional_mass_flux :1412-1711: one thread per face column.
template <int DIR>
__global__ void __launch_bounds__(256)
k_mass_flux(Dm d, const double *__restrict__ G, FluxArgs A) {
  const int i = A.a0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = A.b0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > A.a1 || j > A.b1) return;
  const int st = DIR ? d.pitch : 1;
  DirMetrics D = dir_metrics<DIR>(G, d);
  D.vol_CFL = A.vol_CFL;
  const size_t f2 = ix2(d, i, j);
  const size_t slab = (size_t)d.slab;
  const int nk = d.nk;
  const double dt = A.dt;
  const double Lf = D.Lface[f2] * 1.0;   // G%dy_Cu * por_face_areaU (== 1)
  ColIn C; C.u = A.u; C.h = A.h_in; C.hL = A.hL; C.hR = A.hR; C.vr = A.visc_rem; C.dt = dt; C.nk = nk; C.slab = slab;
  const bool use_visc_rem = (A.visc_rem != nullptr);
  const bool need_adjust = (A.uhbt != nullptr) || A.set_BT_cont;
  const double I_dt = 1.0 / dt;
  const double CFL_dt = A.aggress_adjust ? I_dt : A.CFL_limit_adjust / dt;   // :610-612

  // First sweep: layer transports and the column sums (:615-668)
  double duhdu_tot_0 = 0.0, uh_tot_0 = 0.0, visc_rem_max = 0.0;
  double dx_W = D.dT[f2], dx_E = D.dT[f2 + st];
  if (A.vol_CFL) {   // :651-654 / :1544-1547 (ratio_max :2660)
    auto ratio_max = [](double a, double b, double maxrat) { return (fabs(a) > fabs(maxrat * b)) ? maxrat : a / b; };
    dx_W = ratio_max(D.areaT[f2], D.Lface[f2], 1000.0 * D.dT[f2]);
    dx_E = ratio_max(D.areaT[f2 + st], D.Lface[f2], 1000.0 * D.dT[f2 + st]);
  }
  for (int k = 0; k < nk; k++) {
    const size_t f = f2 + (size_t)k * slab;
    const double vrem = use_visc_rem ? A.visc_rem[f] : 1.0;
    double uh, duhdu;
    flux_layer(D, st, f, f2, A.u[f], A.h_in, A.hL, A.hR, dt, vrem, Lf, uh, duhdu);
    A.uh[f] = uh;
    duhdu_tot_0 = duhdu_tot_0 + duhdu;
    uh_tot_0 = uh_tot_0 + uh;
    visc_rem_max = dmax(visc_rem_max, vrem);
  }
  if (!need_adjust) return;
  if (!(use_visc_rem && A.use_visc_rem_max)) visc_rem_max = 1.0;

  // Limits on du that keep the CFL number between -1 and 1 (:646-723)
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  double du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
  double du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
  const double maskC = D.maskC[f2];
  if (A.aggress_adjust) {   // :664-678, :693-704 / :1558-1572, :1585-1596: the limits look at the neighbouring faces' velocities
    for (int k = 0; k < nk; k++) {
      const size_t f = f2 + (size_t)k * slab;
      const double uk = A.u[f];
      const double lim_max = 0.499 * ((dx_W * I_dt - uk) + dmin(0.0, A.u[f - st]));
      const double lim_min = 0.499 * ((-dx_E * I_dt - uk) + dmax(0.0, A.u[f + st]));
      if (use_visc_rem) {
        const double vrem = A.visc_rem[f];
        if (du_max_CFL * vrem > lim_max) du_max_CFL = lim_max / vrem;
        if (du_min_CFL * vrem < lim_min) du_min_CFL = lim_min / vrem;
      } else {
        du_max_CFL = dmin(du_max_CFL, lim_max);
        du_min_CFL = dmax(du_min_CFL, lim_min);
      }
    }
  } else if (use_visc_rem) {
    for (int k = 0; k < nk; k++) {
      const size_t f = f2 + (size_t)k * slab;
      const double uk = A.u[f], vrem = A.visc_rem[f];
      if (du_max_CFL * vrem > dx_W * CFL_dt - uk * maskC) du_max_CFL = (dx_W * CFL_dt - uk) / vrem;
      if (du_min_CFL * vrem < -dx_E * CFL_dt - uk * maskC) du_min_CFL = -(dx_E * CFL_dt + uk) / vrem;
    }
  } else {
    for (int k = 0; k < nk; k++) {
      const double uk = A.u[f2 + (size_t)k * slab];
      du_max_CFL = dmin(du_max_CFL, dx_W * CFL_dt - uk);
      du_min_CFL = dmax(du_min_CFL, -(dx_E * CFL_dt + uk));
    }
  }
  du_max_CFL = dmax(du_max_CFL, 0.0);
  du_min_CFL = dmin(du_min_CFL, 0.0);

  if (A.uhbt) {
    const double du = flux_adjust<DIR>(D, st, C, f2, Lf, A.uhbt[f2], uh_tot_0, duhdu_tot_0, du_max_CFL,
                                       du_min_CFL, A.tol_eta, A.tol_vel, A.better_iter, true, A.uh);
    if (A.u_cor) {
      for (int k = 0; k < nk; k++) {
        const size_t f = f2 + (size_t)k * slab;
        const double vrem = use_visc_rem ? A.visc_rem[f] : 1.0;
        A.u_cor[f] = A.u[f] + du * vrem;
      }
    }
    if (A.du_cor) A.du_cor[f2] = du;
  }

  if (A.set_BT_cont) {   // set_zonal_BT_cont :1246-1409 / set_merid_BT_cont :2143-2304
    const double Idt = 1.0 / dt, min_visc_rem = 0.1, CFL_min = 1e-6;
    const double du0 = flux_adjust<DIR>(D, st, C, f2, Lf, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                        A.tol_eta, A.tol_vel, A.better_iter, true, nullptr);
    const double du_CFL = (CFL_min * Idt) * D.dC[f2];
    double duR = dmin(0.0, du0 - du_CFL);
    double duL = dmax(0.0, du0 + du_CFL);
    for (int k = 0; k < nk; k++) {
      const size_t f = f2 + (size_t)k * slab;
      const double vrem = use_visc_rem ? A.visc_rem[f] : 1.0, uk = A.u[f];
      const double visc_rem_lim = dmax(vrem, min_visc_rem * visc_rem_max);
      if (visc_rem_lim > 0.0) {
        if (uk + duR * visc_rem_lim > -du_CFL * vrem) duR = -(uk + du_CFL * vrem) / visc_rem_lim;
        if (uk + duL * visc_rem_lim < du_CFL * vrem) duL = -(uk - du_CFL * vrem) / visc_rem_lim;
      }
    }
    double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
    for (int k = 0; k < nk; k++) {
      const size_t f = f2 + (size_t)k * slab;
      const double vrem = use_visc_rem ? A.visc_rem[f] : 1.0, uk = A.u[f];
      const double u_L = uk + duL * vrem, u_R = uk + duR * vrem, u_0 = uk + du0 * vrem;
      double uh_0, uh_L, uh_R, d_0, d_L, d_R;
      flux_layer(D, st, f, f2, u_0, A.h_in, A.hL, A.hR, dt, vrem, Lf, uh_0, d_0);
      flux_layer(D, st, f, f2, u_L, A.h_in, A.hL, A.hR, dt, vrem, Lf, uh_L, d_L);
      flux_layer(D, st, f, f2, u_R, A.h_in, A.hL, A.hR, dt, vrem, Lf, uh_R, d_R);
      FAmt_0 = FAmt_0 + d_0; FAmt_L = FAmt_L + d_L; FAmt_R = FAmt_R + d_R;
      uhtot_L = uhtot_L + uh_L; uhtot_R = uhtot_R + uh_R;
    }
    double FA_0 = FAmt_0, FA_avg = FAmt_0;
    if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
    if (FA_avg > dmax(FA_0, FAmt_L)) FA_avg = dmax(FA_0, FAmt_L);
    else if (FA_avg < dmin(FA_0, FAmt_L)) FA_0 = FA_avg;
    A.FA_m0[f2] = FA_0; A.FA_mm[f2] = FAmt_L;
    if (fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0) A.uBT_mm[f2] = 0.0;
    else A.uBT_mm[f2] = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

    FA_0 = FAmt_0; FA_avg = FAmt_0;
    if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
    if (FA_avg > dmax(FA_0, FAmt_R)) FA_avg = dmax(FA_0, FAmt_R);
    else if (FA_avg < dmin(FA_0, FAmt_R)) FA_0 = FA_avg;
    A.FA_p0[f2] = FA_0; A.FA_pp[f2] = FAmt_R;
    if (fabs(FAmt_R - FA_0) <= 1e-12 * FA_0) A.uBT_pp[f2] = 0.0;
    else A.uBT_pp[f2] = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));
  }
}

// zonal_flux_thickness :975-1089 / merid_flux_thickness :1866-1988
template <int DIR>
__global__ void __launch_bounds__(256)
k_flux_thickness(Dm d, const double *__restrict__ G, const double *__restrict__ u,
                 const double *__restrict__ h, const double *__restrict__ hL,
                 const double *__restrict__ hR, double *__restrict__ h_u, double dt, int marginal,
                 const double *__restrict__ visc_rem, int a0, int a1, int b0, int b1, int vol_CFL) {
  const int i = a0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = b0 + blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > a1 || j > b1) return;
  const int st = DIR ? d.pitch : 1;
  const double *IdT = gm(G, d, DIR ? MOM6X_G_IdyT : MOM6X_G_IdxT);
  const double *Lface = gm(G, d, DIR ? MOM6X_G_dx_Cv : MOM6X_G_dy_Cu), *IareaT = gm(G, d, MOM6X_G_IareaT);
  const size_t f2 = ix2(d, i, j), f = f2 + (size_t)k * d.slab, p = f + st;
  const double uf = u[f];
  double h_avg, h_marg;
  if (uf > 0.0) {
    const double CFL = vol_CFL ? (uf * dt) * (Lface[f2] * IareaT[f2]) : uf * dt * IdT[f2];   // :1019 / :1917
    const double curv_3 = (hL[f] + hR[f]) - 2.0 * h[f];
    h_avg = hR[f] + CFL * (0.5 * (hL[f] - hR[f]) + curv_3 * (CFL - 1.5));
    h_marg = hR[f] + CFL * ((hL[f] - hR[f]) + 3.0 * curv_3 * (CFL - 1.0));
  } else if (uf < 0.0) {
    const double CFL = vol_CFL ? (-uf * dt) * (Lface[f2] * IareaT[f2 + st]) : -uf * dt * IdT[f2 + st];   // :1025 / :1924
    const double curv_3 = (hL[p] + hR[p]) - 2.0 * h[p];
    h_avg = hL[p] + CFL * (0.5 * (hR[p] - hL[p]) + curv_3 * (CFL - 1.5));
    h_marg = hL[p] + CFL * ((hR[p] - hL[p]) + 3.0 * curv_3 * (CFL - 1.0));
  } else {
    h_avg = 0.5 * (hL[p] + hR[f]);
    h_marg = 0.5 * (hL[p] + hR[f]);
  }
  double hu = marginal ? h_marg : h_avg;
  if (visc_rem) hu = hu * (visc_rem[f] * 1.0);
  else hu = hu * 1.0;
  h_u[f] = hu;
}

// continuity_zonal_convergence :348 / continuity_merdional_convergence :386.  Column walk over KCHUNK layers: the 2-D plane
// IareaT is read once per chunk, not once per layer (one word per cell-layer less).  The step's time averages of the
// thicknesses that read the same arrays ride along (RK2.F90:808-810, :1025-1027, :1064-1066): `av_mode` 1: h_av = 0.5 * (h_old +
// h_new) with h_old = the routine's input thickness `av_src` (second direction of the predictor's call); 2: h_av = h_old (a copy
// kept by the first direction of the corrector's call, whose in-place update destroys h_old); 3: h_av = 0.5 * (h_av + h_new)
// (its second direction).  Own cells only; the halo frame of h_av is the caller's (k_h_av after the group pass).
template <int DIR>
__global__ void __launch_bounds__(256)
k_convergence(Dm d, const double *__restrict__ G, double *h, const double *__restrict__ uh, double dt,
              const double *hin, double h_min, int i0, int i1, int j0, int j1, int *flag, double *h_av, const double *av_src,
              int av_mode) {
  const int i = I_BASE(i0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < i0 || i > i1 || j > j1) return;
  const int st = DIR ? d.pitch : 1;
  const size_t c2 = ix2(d, i, j), slab = (size_t)d.slab;
  const double IareaT = gm(G, d, MOM6X_G_IareaT)[c2];
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool own = (i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1);
  bool bad = false;
  for (int k = k0; k < k1; k++) {
    const size_t c = c2 + (size_t)k * slab;
    const double h_old = hin[c];
    const double hn = h_old - dt * IareaT * (uh[c] - uh[c - st]);
    const double hnew = dmax(hn, h_min);
    h[c] = hnew;
    if (av_mode && own) {
      if (av_mode == 1) h_av[c] = 0.5 * (av_src[c] + hnew);
      else if (av_mode == 2) h_av[c] = h_old;
      else h_av[c] = 0.5 * (h_av[c] + hnew);
    }
    bad = bad || (hn != hn);
  }
  if (bad) atomicOr(flag, 1);   // a NaN has reached the thicknesses: MOM6X_ENUMERIC at the next mom6x_ctx_sync
}

template <int DIR>
int run_direction(mom6x_ctx *c, const double *u, const double *h_src, double *h, double *uh, double dt,
                  int ish, int ieh, int jsh, int jeh, const double *uhbt, const double *visc_rem,
                  double *u_cor, const mom6x_BT_cont *BT, double *du_cor, const double *hin_conv,
                  double h_min_conv, bool first_pass) {
  const Dm d = c->d;
  const mom6x_continuity_params &P = c->cont;
  const dim3 blk(64, 4, 1);
  // edge thicknesses: one cell beyond the LB bounds in the sweep direction
  int ei0 = ish, ei1 = ieh, ej0 = jsh, ej1 = jeh;
  if (DIR == 0) { ei0 = ish - 1; ei1 = ieh + 1; } else { ej0 = jsh - 1; ej1 = jeh + 1; }
  const int scheme = P.upwind_1st ? 2 : (P.simple_2nd ? 1 : 0);
  // CONT_PPM_AGGRESS_ADJUST / CONT_PPM_VOLUME_BASED_CFL (non-default, :2725-2733) are carried by the thread-per-column kernels
  // alone: their limits read the neighbouring faces' velocities, which the marching kernels do not stage.  Those kernels sum
  // columns in the reference's order, so under these switches sum_order has no effect (include/mom6x.h).
  const bool special = P.aggress_adjust || P.vol_CFL;
  const bool wave = !special && (P.sum_order == MOM6X_SUM_TREE16 || P.sum_order == MOM6X_SUM_TREE16_FMA);   // one wavefront row per face column, everything in registers
  const bool lds = !special && (wave || use_lds_path(d.nk));
  if (!lds)
    KLAUNCH(c, "k_edge<DIR>", k_edge<DIR>, grid3(ei1 - ei0 + 1, ej1 - ej0 + 1, d.nk, blk), blk, d, c->G, h_src,
                       c->hL, c->hR, 2.0 * c->GV.Angstrom_H, scheme, P.monotonic, ei0, ei1, ej0, ej1);
  FluxArgs A;
  memset(&A, 0, sizeof(A));
  A.u = u; A.h_in = h_src; A.hL = c->hL; A.hR = c->hR; A.uh = uh; A.uhbt = uhbt; A.visc_rem = visc_rem;
  A.u_cor = u_cor; A.du_cor = du_cor; A.set_BT_cont = (BT != nullptr);
  if (BT) {
    if (DIR == 0) { A.FA_m0 = BT->FA_u_W0; A.FA_mm = BT->FA_u_WW; A.uBT_mm = BT->uBT_WW;
                    A.FA_p0 = BT->FA_u_E0; A.FA_pp = BT->FA_u_EE; A.uBT_pp = BT->uBT_EE; }
    else          { A.FA_m0 = BT->FA_v_S0; A.FA_mm = BT->FA_v_SS; A.uBT_mm = BT->vBT_SS;
                    A.FA_p0 = BT->FA_v_N0; A.FA_pp = BT->FA_v_NN; A.uBT_pp = BT->vBT_NN; }
  }
  A.dt = dt; A.CFL_limit_adjust = P.CFL_limit_adjust; A.tol_eta = P.tol_eta; A.tol_vel = P.tol_vel;
  A.better_iter = P.better_iter; A.use_visc_rem_max = P.use_visc_rem_max;
  A.aggress_adjust = P.aggress_adjust; A.vol_CFL = P.vol_CFL;
  if (DIR == 0) { A.a0 = ish - 1; A.a1 = ieh; A.b0 = jsh; A.b1 = jeh; }
  else          { A.a0 = ish; A.a1 = ieh; A.b0 = jsh - 1; A.b1 = jeh; }
  if (du_cor) HIPCHK(hipMemsetAsync(du_cor, 0, sizeof(double) * d.slab, c->stream));
  double *BT_h = BT ? (DIR == 0 ? BT->h_u : BT->h_v) : nullptr;
  // A group pass of this call's inputs may still be in flight on the halo stream (start_group_pass by the caller:
  // the RK2 step starts pass_uvp / pass_uv / pass_visc_rem and calls continuity straight away).  The faces of the
  // tile's own rows (columns) only read owned points of u, visc_rem -- symmetric memory: the faces on the tile edge are
  // owned too -- so the first pass does them while the messages travel, waits (complete_group_pass), and then does the
  // rows (columns) in the halo, which are what the pass delivers.  A face's result does not depend on the launch it is in.
  const bool split = wave && first_pass && c->pass_pending;
  if (!split) halo_complete(c);
  if (lds) {
    LdsArgs E;
    E.h_min = 2.0 * c->GV.Angstrom_H; E.scheme = scheme; E.monotonic = P.monotonic;
    E.marginal = P.marginal_faces; E.h_face = BT_h; E.fma = (P.sum_order == MOM6X_SUM_TREE16_FMA);
    auto part = [&](int a0, int a1, int b0, int b1) -> int {
      if (a0 > a1 || b0 > b1) return MOM6X_OK;
      FluxArgs S = A;
      S.a0 = a0; S.a1 = a1; S.b0 = b0; S.b1 = b1;
      return wave ? mass_flux_wave(c, DIR, S, E) : mass_flux_lds(c, DIR, S, E);
    };
    auto pair = [&](int a0, int a1, int b0, int b1, int c0, int c1, int e0, int e1) -> int {
      if (!wave) { const int r1 = part(a0, a1, b0, b1); return r1 ? r1 : part(c0, c1, e0, e1); }
      FluxArgs S1 = A, S2 = A;
      S1.a0 = a0; S1.a1 = a1; S1.b0 = b0; S1.b1 = b1;
      S2.a0 = c0; S2.a1 = c1; S2.b0 = e0; S2.b1 = e1;
      return mass_flux_wave_pair(c, DIR, S1, S2, E);
    };
    int rc;
    if (!split) {
      rc = part(A.a0, A.a1, A.b0, A.b1);
    } else if (DIR == 0) {   // rows: own rows first, then the halo rows south and north of them
      const int o0 = A.b0 > 0 ? A.b0 : 0, o1 = A.b1 < d.nj - 1 ? A.b1 : d.nj - 1;
      rc = part(A.a0, A.a1, o0, o1);
      halo_complete(c);
      if (!rc) rc = pair(A.a0, A.a1, A.b0, o0 - 1, A.a0, A.a1, o1 + 1, A.b1);     // the two rims in ONE launch
    } else {                 // columns
      const int o0 = A.a0 > 0 ? A.a0 : 0, o1 = A.a1 < d.ni - 1 ? A.a1 : d.ni - 1;
      rc = part(o0, o1, A.b0, A.b1);
      halo_complete(c);
      if (!rc) rc = pair(A.a0, o0 - 1, A.b0, A.b1, o1 + 1, A.a1, A.b0, A.b1);
    }
    if (rc) return rc;
  } else {
    KLAUNCH(c, "k_mass_flux<DIR>", k_mass_flux<DIR>, grid3(A.a1 - A.a0 + 1, A.b1 - A.b0 + 1, 1, blk), blk, d, c->G, A);
    if (BT_h) {
      KLAUNCH(c, "k_flux_thickness<DIR>", k_flux_thickness<DIR>, grid3(A.a1 - A.a0 + 1, A.b1 - A.b0 + 1, d.nk, blk), blk,
                         d, c->G, (u_cor ? (const double *)u_cor : u), h_src, c->hL, c->hR, BT_h, dt,
                         P.marginal_faces, visc_rem, A.a0, A.a1, A.b0, A.b1, P.vol_CFL);
    }
  }
  if (first_pass || !c->cont_h_unused)   // (the second direction's new thicknesses are the routine's result -- unless nobody wants it)
  {
    // the time average of the thicknesses the RK2 step wants next to this convergence (c->cont_av_*; 0: none)
    int av_mode = 0;
    if (c->cont_av_kind == 1 && !first_pass) av_mode = 1;
    if (c->cont_av_kind == 2) av_mode = first_pass ? 2 : 3;
    KLAUNCH(c, "k_convergence<DIR>", k_convergence<DIR>, grid3(nxa(ieh - ish + 1, ish), jeh - jsh + 1, nchunks(d.nk), blk), blk, d, c->G,
                       h, uh, dt, hin_conv, h_min_conv, ish, ieh, jsh, jeh, c->flag, c->cont_av, c->cont_av_src, av_mode);
  }
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

}  // namespace

extern "C" int mom6x_continuity_init(mom6x_ctx *c, const mom6x_continuity_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_continuity_init: null argument");
  REQUIRE(p->sum_order == MOM6X_SUM_REFERENCE || p->sum_order == MOM6X_SUM_TREE16 || p->sum_order == MOM6X_SUM_TREE16_FMA, MOM6X_EINVAL,
          "continuity_PPM: sum_order must be MOM6X_SUM_REFERENCE (0), MOM6X_SUM_TREE16 (1) or MOM6X_SUM_TREE16_FMA (2)");
  REQUIRE(p->sum_order == MOM6X_SUM_REFERENCE || mass_flux_wave_usable(c->d.nk), MOM6X_EUNSUPPORTED,
          "continuity_PPM: sum_order = MOM6X_SUM_TREE16(_FMA) carries at most 128 layers; use MOM6X_SUM_REFERENCE");
  c->cont = *p;
  c->cont_init = true;
  return MOM6X_OK;
}

// Newton statistics of the mass-flux kernel (sum_order TREE16 only): out[0] = flux re-evaluations (whole column sweeps, counted
// per wavefront of four faces), out[1] = Newton solves (per wavefront), out[2] = solves repeated with the exact CFL limits,
// accumulated while collection is on.  `mode`: 1 switch collection on (and reset), 0 switch it off, -1 leave it as it is.  While
// it is on the kernel runs in a slower variant (the counters spill registers): for diagnostics, not for timed runs.
// out[0] / out[1] is the number bench.py prints as newton_evals_per_solve.  Synchronises the stream.
extern "C" int mom6x_continuity_stats(mom6x_ctx *c, int mode, unsigned long long *out3) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_continuity_stats: null context");
  const int reset = (mode == 1);
  if (mode == 1) c->cont_stats_on = true;
  if (mode == 0) c->cont_stats_on = false;
  HIPCHK(hipSetDevice(c->device));
  if (!c->cont_stats) {
    HIPCHK(hipMalloc(&c->cont_stats, 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(c->cont_stats, 0, 4 * sizeof(unsigned long long), c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (out3) HIPCHK(hipMemcpy(out3, c->cont_stats, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemsetAsync(c->cont_stats, 0, 4 * sizeof(unsigned long long), c->stream));
  return MOM6X_OK;
}

extern "C" int mom6x_continuity_PPM(mom6x_ctx *c, const double *u, const double *v, const double *hin,
                                    double *h, double *uh, double *vh, double dt, const double *uhbt,
                                    const double *vhbt, const double *visc_rem_u, const double *visc_rem_v,
                                    double *u_cor, double *v_cor, const mom6x_BT_cont *BT, double *du_cor,
                                    double *dv_cor) {
  REQUIRE(c && c->cont_init, MOM6X_EINVAL,
          "MOM_continuity_PPM: Module must be initialized before it is used.");
  REQUIRE(u && v && hin && h && uh && vh, MOM6X_EINVAL, "continuity_PPM: null mandatory array");
  REQUIRE((visc_rem_u != nullptr) == (visc_rem_v != nullptr), MOM6X_EINVAL,
          "MOM_continuity_PPM: Either both visc_rem_u and visc_rem_v or neither one must be present "
          "in call to continuity_PPM.");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const mom6x_continuity_params &P = c->cont;
  const double h_min = c->GV.Angstrom_H;
  int stencil = 3; if (P.simple_2nd) stencil = 2; if (P.upwind_1st) stencil = 1;
  REQUIRE(d.halo >= stencil, MOM6X_EINVAL, "continuity_PPM: halo smaller than the continuity stencil");
  const bool x_first = ((c->first_direction % 2) == 0);
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1;
  int rc;
  if (x_first) {
    rc = run_direction<0>(c, u, hin, h, uh, dt, is, ie, js - stencil, je + stencil, uhbt, visc_rem_u, u_cor, BT,
                          du_cor, hin, 0.0, true);
    if (rc) return rc;
    rc = run_direction<1>(c, v, h, h, vh, dt, is, ie, js, je, vhbt, visc_rem_v, v_cor, BT, dv_cor, h, h_min, false);
  } else {
    rc = run_direction<1>(c, v, hin, h, vh, dt, is - stencil, ie + stencil, js, je, vhbt, visc_rem_v, v_cor, BT,
                          dv_cor, hin, 0.0, true);
    if (rc) return rc;
    rc = run_direction<0>(c, u, h, h, uh, dt, is, ie, js, je, uhbt, visc_rem_u, u_cor, BT, du_cor, h, h_min, false);
  }
  return rc;
}
