// hor_visc.hip -- hor_visc_init and horizontal_viscosity on gfx950 (MOM_hor_visc.F90:2322-3290 / :266-2317).
//
// On the path: LAPLACIAN and/or BIHARMONIC with background coefficients, SMAGORINSKY_KH / _AH
// (+ BOUND_CORIOLIS_BIHARM), LEITH_KH / LEITH_AH (+ MODIFIED_LEITH, USE_BETA_IN_LEITH), ADD_LES_VISCOSITY,
// BOUND_KH / BOUND_AH ("better" and legacy form), USE_LAND_MASK_FOR_HVISC, NOSLIP.  Everything else of the module is rejected by mom6x_hor_visc_init's caller
// contract (include/mom6x.h).
//
// The reference works one layer at a time on 2-D temporaries.  Here the chain
//     (u, v) -> sh_xx, sh_xy -> Del2u, Del2v -> str_xx, str_xy -> diffu, diffv
// is either ONE kernel (k_hv_fused, the default: LDS tiles with a 3-cell halo, every stage computed on the whole tile, 20 -> ~7
// words per cell-layer, 4.75 ms per call at 1440 x 1080 x 75) or four 3-D kernels (column walk over KCHUNK layers with the 2-D
// coefficient planes in registers where it pays; 6.0 ms; MOM6X_HORVISC=legacy and every Leith configuration).  h_u, h_v, hq,
// Shear_mag, hrat_min, the viscosities and the strain derivatives are recomputed where they are needed instead of being stored.
#include <algorithm>
#include "mom6x_dev.h"

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);  // halo.hip

enum HV {
  HV_dx2h = 0, HV_dy2h, HV_dx2q, HV_dy2q, HV_DX_dyT, HV_DY_dxT, HV_DX_dyBu, HV_DY_dxBu, HV_red_xx, HV_red_xy,
  HV_Kh_bg_xx, HV_Kh_bg_xy, HV_Kh_Max_xx, HV_Kh_Max_xy, HV_Lap2_xx, HV_Lap2_xy,
  HV_Idx2dyCu, HV_Idxdy2u, HV_Idx2dyCv, HV_Idxdy2v, HV_Ah_bg_xx, HV_Ah_bg_xy, HV_Ah_Max_xx, HV_Ah_Max_xy,
  HV_Bih_xx, HV_Bih_xy, HV_Bih2_xx, HV_Bih2_xy, HV_u0u, HV_u0v, HV_v0u, HV_v0v,
  HV_Lap3_xx, HV_Lap3_xy, HV_Bih6_xx, HV_Bih6_xy, HV_dF_dx, HV_dF_dy, HV_COUNT
};

namespace {

inline dim3 blk2() { return dim3(64, 4, 1); }
#define MG(n) gm(G, d, MOM6X_G_##n)
#define PLN(n) (P + (size_t)(n) * slab)

__device__ __forceinline__ double dmax4(double a, double b, double c, double e) { return dmax(dmax(dmax(a, b), c), e); }
__device__ __forceinline__ double dmin4(double a, double b, double c, double e) { return dmin(dmin(dmin(a, b), c), e); }

__device__ __forceinline__ bool in_box(int i, int j, int i0, int i1, int j0, int j1) { return i >= i0 && i <= i1 && j >= j0 && j <= j1; }

// ---- hor_visc_init :2869-3024: metric products, reductions, background coefficients, Smagorinsky constants ------
__global__ void __launch_bounds__(256)
k_hv_init1(Dm d, const double *__restrict__ G, mom6x_hor_visc_params CS, double *__restrict__ P) {
  const int i = -d.halo + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -d.halo + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + d.halo || j > d.nj - 1 + d.halo) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1, Isq = -1, Ieq = ie, Jsq = -1, Jeq = je;
  const double *dxBu = MG(dxBu), *dyBu = MG(dyBu), *IdxBu = MG(IdxBu), *IdyBu = MG(IdyBu), *dxT = MG(dxT), *dyT = MG(dyT);
  const double *IdxT = MG(IdxT), *IdyT = MG(IdyT), *IdxCu = MG(IdxCu), *IdyCu = MG(IdyCu), *IdxCv = MG(IdxCv), *IdyCv = MG(IdyCv);
  const double *dy_Cu = MG(dy_Cu), *dyCu = MG(dyCu), *dx_Cv = MG(dx_Cv), *dxCv = MG(dxCv), *fBu = MG(CoriolisBu);
  double dx2q = 0., dy2q = 0., dx2h = 0., dy2h = 0.;
  if (in_box(i, j, is - 2, Ieq + 1, js - 2, Jeq + 1)) {
    dx2q = dxBu[x] * dxBu[x]; dy2q = dyBu[x] * dyBu[x];
    PLN(HV_dx2q)[x] = dx2q; PLN(HV_dy2q)[x] = dy2q;
    PLN(HV_DX_dyBu)[x] = dxBu[x] * IdyBu[x]; PLN(HV_DY_dxBu)[x] = dyBu[x] * IdxBu[x];
  }
  if (in_box(i, j, is - 2, Ieq + 2, js - 2, Jeq + 2)) {
    dx2h = dxT[x] * dxT[x]; dy2h = dyT[x] * dyT[x];
    PLN(HV_dx2h)[x] = dx2h; PLN(HV_dy2h)[x] = dy2h;
    PLN(HV_DX_dyT)[x] = dxT[x] * IdyT[x]; PLN(HV_DY_dxT)[x] = dyT[x] * IdxT[x];
  }
  if (in_box(i, j, Isq, Ieq + 1, Jsq, Jeq + 1)) {   // reduction_xx :2894-2908
    double r = 1.0;
    if ((dy_Cu[x] > 0.0) && (dy_Cu[x] < dyCu[x]) && (dy_Cu[x] < dyCu[x] * r)) r = dy_Cu[x] / (dyCu[x]);
    if ((dy_Cu[x - 1] > 0.0) && (dy_Cu[x - 1] < dyCu[x - 1]) && (dy_Cu[x - 1] < dyCu[x - 1] * r)) r = dy_Cu[x - 1] / (dyCu[x - 1]);
    if ((dx_Cv[x] > 0.0) && (dx_Cv[x] < dxCv[x]) && (dx_Cv[x] < dxCv[x] * r)) r = dx_Cv[x] / (dxCv[x]);
    if ((dx_Cv[x - st] > 0.0) && (dx_Cv[x - st] < dxCv[x - st]) && (dx_Cv[x - st] < dxCv[x - st] * r)) r = dx_Cv[x - st] / (dxCv[x - st]);
    PLN(HV_red_xx)[x] = r;
  }
  if (in_box(i, j, is - 1, Ieq, js - 1, Jeq)) {     // reduction_xy :2909-2923
    double r = 1.0;
    if ((dy_Cu[x] > 0.0) && (dy_Cu[x] < dyCu[x]) && (dy_Cu[x] < dyCu[x] * r)) r = dy_Cu[x] / (dyCu[x]);
    if ((dy_Cu[x + st] > 0.0) && (dy_Cu[x + st] < dyCu[x + st]) && (dy_Cu[x + st] < dyCu[x + st] * r)) r = dy_Cu[x + st] / (dyCu[x + st]);
    if ((dx_Cv[x] > 0.0) && (dx_Cv[x] < dxCv[x]) && (dx_Cv[x] < dxCv[x] * r)) r = dx_Cv[x] / (dxCv[x]);
    if ((dx_Cv[x + 1] > 0.0) && (dx_Cv[x + 1] < dxCv[x + 1]) && (dx_Cv[x + 1] < dxCv[x + 1] * r)) r = dx_Cv[x + 1] / (dxCv[x + 1]);
    PLN(HV_red_xy)[x] = r;
  }
  const bool h_box = in_box(i, j, is - 1, Ieq + 1, js - 1, Jeq + 1), q_box = in_box(i, j, is - 1, Ieq, js - 1, Jeq);
  if (CS.Laplacian) {                               // :2924-2976
    const double Kh_Limit = 0.3 / (CS.dt * 4.0);
    if (h_box) {
      const double g2 = (2.0 * dx2h * dy2h) / (dx2h + dy2h);
      if (CS.Smagorinsky_Kh) PLN(HV_Lap2_xx)[x] = CS.Smag_Lap_const * g2;
      if (CS.Leith_Kh) PLN(HV_Lap3_xx)[x] = CS.Leith_Lap_const * (g2 * sqrt(g2));   // :2900-2902
      double K = dmax(CS.Kh, CS.Kh_vel_scale * sqrt(g2));
      if (CS.bound_Kh && !CS.better_bound_Kh) { PLN(HV_Kh_Max_xx)[x] = Kh_Limit * g2; K = dmin(K, Kh_Limit * g2); }
      PLN(HV_Kh_bg_xx)[x] = K;
    }
    if (q_box) {
      const double g2 = (2.0 * dx2q * dy2q) / (dx2q + dy2q);
      if (CS.Smagorinsky_Kh) PLN(HV_Lap2_xy)[x] = CS.Smag_Lap_const * g2;
      if (CS.Leith_Kh) PLN(HV_Lap3_xy)[x] = CS.Leith_Lap_const * (g2 * sqrt(g2));   // :2925-2927
      double K = dmax(CS.Kh, CS.Kh_vel_scale * sqrt(g2));
      if (CS.bound_Kh && !CS.better_bound_Kh) { PLN(HV_Kh_Max_xy)[x] = Kh_Limit * g2; K = dmin(K, Kh_Limit * g2); }
      PLN(HV_Kh_bg_xy)[x] = K;
    }
  }
  if (CS.biharmonic) {                              // :2977-3024
    if (in_box(i, j, is - 2, Ieq + 1, js - 1, Jeq + 1)) {
      PLN(HV_Idx2dyCu)[x] = (IdxCu[x] * IdxCu[x]) * IdyCu[x];
      PLN(HV_Idxdy2u)[x] = IdxCu[x] * (IdyCu[x] * IdyCu[x]);
    }
    if (in_box(i, j, is - 1, Ieq + 1, js - 2, Jeq + 1)) {
      PLN(HV_Idx2dyCv)[x] = (IdxCv[x] * IdxCv[x]) * IdyCv[x];
      PLN(HV_Idxdy2v)[x] = IdxCv[x] * (IdyCv[x] * IdyCv[x]);
    }
    const double Ah_Limit = 0.3 / (CS.dt * 64.0);
    double BoundCorConst = 0.0;
    if (CS.Smagorinsky_Ah && CS.bound_Coriolis) BoundCorConst = 1.0 / (5.0 * (CS.bound_Cor_vel * CS.bound_Cor_vel));
    if (h_box) {
      const double g2 = (2.0 * dx2h * dy2h) / (dx2h + dy2h);
      if (CS.Smagorinsky_Ah) {
        PLN(HV_Bih_xx)[x] = CS.Smag_bi_const * (g2 * g2);
        if (CS.bound_Coriolis) {
          const double fmax = dmax4(fabs(fBu[x - 1 - st]), fabs(fBu[x - st]), fabs(fBu[x - 1]), fabs(fBu[x]));
          PLN(HV_Bih2_xx)[x] = (g2 * g2 * g2) * (fmax * BoundCorConst);
        }
      }
      if (CS.Leith_Ah) { const double g3 = g2 * sqrt(g2); PLN(HV_Bih6_xx)[x] = CS.Leith_bi_const * (g3 * g3); }   // :2984-2986
      double A = dmax(CS.Ah, CS.Ah_vel_scale * g2 * sqrt(g2));
      if (CS.Ah_time_scale > 0.) A = dmax(A, (g2 * g2) / CS.Ah_time_scale);
      if (CS.bound_Ah && !CS.better_bound_Ah) { PLN(HV_Ah_Max_xx)[x] = Ah_Limit * (g2 * g2); A = dmin(A, Ah_Limit * (g2 * g2)); }
      PLN(HV_Ah_bg_xx)[x] = A;
    }
    if (q_box) {
      const double g2 = (2.0 * dx2q * dy2q) / (dx2q + dy2q);
      if (CS.Smagorinsky_Ah) {
        PLN(HV_Bih_xy)[x] = CS.Smag_bi_const * (g2 * g2);
        if (CS.bound_Coriolis) PLN(HV_Bih2_xy)[x] = (g2 * g2 * g2) * (fabs(fBu[x]) * BoundCorConst);
      }
      if (CS.Leith_Ah) { const double g3 = g2 * sqrt(g2); PLN(HV_Bih6_xy)[x] = CS.Leith_bi_const * (g3 * g3); }   // :3014-3016
      double A = dmax(CS.Ah, CS.Ah_vel_scale * g2 * sqrt(g2));
      if (CS.Ah_time_scale > 0.) A = dmax(A, (g2 * g2) / CS.Ah_time_scale);
      if (CS.bound_Ah && !CS.better_bound_Ah) { PLN(HV_Ah_Max_xy)[x] = Ah_Limit * (g2 * g2); A = dmin(A, Ah_Limit * (g2 * g2)); }
      PLN(HV_Ah_bg_xy)[x] = A;
    }
  }
}

// ---- :3025-3048 (Kh_Max) and :3054-3070 (u0u, u0v, v0u, v0v) -- need the planes of k_hv_init1 at neighbours -----
__global__ void __launch_bounds__(256)
k_hv_init2(Dm d, const double *__restrict__ G, mom6x_hor_visc_params CS, double *__restrict__ P) {
  const int i = -d.halo + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -d.halo + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + d.halo || j > d.nj - 1 + d.halo) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1, Ieq = ie, Jeq = je;
  const double *IdxCu = MG(IdxCu), *IdyCu = MG(IdyCu), *IdxCv = MG(IdxCv), *IdyCv = MG(IdyCv), *IareaCu = MG(IareaCu), *IareaCv = MG(IareaCv);
  const double *dx2h = PLN(HV_dx2h), *dy2h = PLN(HV_dy2h), *dx2q = PLN(HV_dx2q), *dy2q = PLN(HV_dy2q);
  const double *DX_dyT = PLN(HV_DX_dyT), *DY_dxT = PLN(HV_DY_dxT), *DX_dyBu = PLN(HV_DX_dyBu), *DY_dxBu = PLN(HV_DY_dxBu);
  const double Idt = 1.0 / CS.dt;
  if (CS.Laplacian && CS.better_bound_Kh) {
    if (in_box(i, j, is - 1, Ieq + 1, js - 1, Jeq + 1)) {
      const double denom = dmax(
          (dy2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1]) * dmax(IdyCu[x] * IareaCu[x], IdyCu[x - 1] * IareaCu[x - 1])),
          (dx2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st]) * dmax(IdxCv[x] * IareaCv[x], IdxCv[x - st] * IareaCv[x - st])));
      double r = 0.0;
      if (denom > 0.0) r = CS.bound_coef * 0.25 * Idt / denom;
      PLN(HV_Kh_Max_xx)[x] = r;
    }
    if (in_box(i, j, is - 1, Ieq, js - 1, Jeq)) {
      const double denom = dmax(
          (dx2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x]) * dmax(IdxCu[x] * IareaCu[x], IdxCu[x + st] * IareaCu[x + st])),
          (dy2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x]) * dmax(IdyCv[x] * IareaCv[x], IdyCv[x + 1] * IareaCv[x + 1])));
      double r = 0.0;
      if (denom > 0.0) r = CS.bound_coef * 0.25 * Idt / denom;
      PLN(HV_Kh_Max_xy)[x] = r;
    }
  }
  if (CS.biharmonic && CS.better_bound_Ah) {
    const double *Idxdy2u = PLN(HV_Idxdy2u), *Idx2dyCu = PLN(HV_Idx2dyCu), *Idxdy2v = PLN(HV_Idxdy2v), *Idx2dyCv = PLN(HV_Idx2dyCv);
    if (in_box(i, j, is - 2, Ieq + 1, js - 1, Jeq + 1)) {
      PLN(HV_u0u)[x] = ((Idxdy2u[x] * ((dy2h[x + 1] * DY_dxT[x + 1] * (IdyCu[x + 1] + IdyCu[x])) + (dy2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1])))) +
                        (Idx2dyCu[x] * ((dx2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x])) + (dx2q[x - st] * DX_dyBu[x - st] * (IdxCu[x] + IdxCu[x - st])))));
      PLN(HV_u0v)[x] = ((Idxdy2u[x] * ((dy2h[x + 1] * DX_dyT[x + 1] * (IdxCv[x + 1] + IdxCv[x + 1 - st])) + (dy2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st])))) +
                        (Idx2dyCu[x] * ((dx2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x])) + (dx2q[x - st] * DY_dxBu[x - st] * (IdyCv[x + 1 - st] + IdyCv[x - st])))));
    }
    if (in_box(i, j, is - 1, Ieq + 1, js - 2, Jeq + 1)) {
      PLN(HV_v0u)[x] = ((Idxdy2v[x] * ((dy2q[x] * DX_dyBu[x] * (IdxCu[x + st] + IdxCu[x])) + (dy2q[x - 1] * DX_dyBu[x - 1] * (IdxCu[x - 1 + st] + IdxCu[x - 1])))) +
                        (Idx2dyCv[x] * ((dx2h[x + st] * DY_dxT[x + st] * (IdyCu[x + st] + IdyCu[x - 1 + st])) + (dx2h[x] * DY_dxT[x] * (IdyCu[x] + IdyCu[x - 1])))));
      PLN(HV_v0v)[x] = ((Idxdy2v[x] * ((dy2q[x] * DY_dxBu[x] * (IdyCv[x + 1] + IdyCv[x])) + (dy2q[x - 1] * DY_dxBu[x - 1] * (IdyCv[x] + IdyCv[x - 1])))) +
                        (Idx2dyCv[x] * ((dx2h[x + st] * DX_dyT[x + st] * (IdxCv[x + st] + IdxCv[x])) + (dx2h[x] * DX_dyT[x] * (IdxCv[x] + IdxCv[x - st])))));
    }
  }
}

// ---- :3071-3110: Ah_Max_xx, Ah_Max_xy from u0u .. v0v at neighbours ---------------------------------------------
__global__ void __launch_bounds__(256)
k_hv_init3(Dm d, const double *__restrict__ G, mom6x_hor_visc_params CS, double *__restrict__ P) {
  const int i = -d.halo + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -d.halo + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + d.halo || j > d.nj - 1 + d.halo) return;
  if (!(CS.biharmonic && CS.better_bound_Ah)) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1, Ieq = ie, Jeq = je;
  const double *IdxCu = MG(IdxCu), *IdyCu = MG(IdyCu), *IdxCv = MG(IdxCv), *IdyCv = MG(IdyCv), *IareaCu = MG(IareaCu), *IareaCv = MG(IareaCv);
  const double *dx2h = PLN(HV_dx2h), *dy2h = PLN(HV_dy2h), *dx2q = PLN(HV_dx2q), *dy2q = PLN(HV_dy2q);
  const double *DX_dyT = PLN(HV_DX_dyT), *DY_dxT = PLN(HV_DY_dxT), *DX_dyBu = PLN(HV_DX_dyBu), *DY_dxBu = PLN(HV_DY_dxBu);
  const double *u0u = PLN(HV_u0u), *u0v = PLN(HV_u0v), *v0u = PLN(HV_v0u), *v0v = PLN(HV_v0v);
  const double Idt = 1.0 / CS.dt;
  if (in_box(i, j, is - 1, Ieq + 1, js - 1, Jeq + 1)) {
    const double denom = dmax(
        (dy2h[x] * ((DY_dxT[x] * ((IdyCu[x] * u0u[x]) + (IdyCu[x - 1] * u0u[x - 1]))) + (DX_dyT[x] * ((IdxCv[x] * v0u[x]) + (IdxCv[x - st] * v0u[x - st])))) *
         dmax(IdyCu[x] * IareaCu[x], IdyCu[x - 1] * IareaCu[x - 1])),
        (dx2h[x] * ((DY_dxT[x] * ((IdyCu[x] * u0v[x]) + (IdyCu[x - 1] * u0v[x - 1]))) + (DX_dyT[x] * ((IdxCv[x] * v0v[x]) + (IdxCv[x - st] * v0v[x - st])))) *
         dmax(IdxCv[x] * IareaCv[x], IdxCv[x - st] * IareaCv[x - st])));
    double r = 0.0;
    if (denom > 0.0) r = CS.bound_coef * 0.5 * Idt / denom;
    PLN(HV_Ah_Max_xx)[x] = r;
  }
  if (in_box(i, j, is - 1, Ieq, js - 1, Jeq)) {
    const double denom = dmax(
        (dx2q[x] * ((DX_dyBu[x] * ((u0u[x + st] * IdxCu[x + st]) + (u0u[x] * IdxCu[x]))) + (DY_dxBu[x] * ((v0u[x + 1] * IdyCv[x + 1]) + (v0u[x] * IdyCv[x])))) *
         dmax(IdxCu[x] * IareaCu[x], IdxCu[x + st] * IareaCu[x + st])),
        (dy2q[x] * ((DX_dyBu[x] * ((u0v[x + st] * IdxCu[x + st]) + (u0v[x] * IdxCu[x]))) + (DY_dxBu[x] * ((v0v[x + 1] * IdyCv[x + 1]) + (v0v[x] * IdyCv[x])))) *
         dmax(IdyCv[x] * IareaCv[x], IdyCv[x + 1] * IareaCv[x + 1])));
    double r = 0.0;
    if (denom > 0.0) r = CS.bound_coef * 0.5 * Idt / denom;
    PLN(HV_Ah_Max_xy)[x] = r;
  }
}

// =================================================================================================================
// horizontal_viscosity, stage 1 :724-737, :907-917: sh_xx at h points (Isq-1..Ieq+2), sh_xy at q points (is-2..Ieq+1)
__global__ void __launch_bounds__(256)
k_hv_strain(Dm d, const double *__restrict__ G, const double *__restrict__ P, const double *__restrict__ u,
            const double *__restrict__ v, double *__restrict__ sh_xx, double *__restrict__ sh_xy, int no_slip) {
  const int i = I_BASE(-2) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -2 || i > d.ni + 1 || j > d.nj + 1) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_h = true;                                     // (Isq-1..Ieq+2, Jsq-1..Jeq+2) = (-2..ni+1, -2..nj+1)
  const bool do_q = (i <= d.ni) && (j <= d.nj);               // (is-2..Ieq+1, js-2..Jeq+1) = (-2..ni, -2..nj)
  const double DY_dxT = PLN(HV_DY_dxT)[x], DX_dyT = PLN(HV_DX_dyT)[x];
  const double IdyCu0 = MG(IdyCu)[x], IdyCum = MG(IdyCu)[x - 1], IdxCv0 = MG(IdxCv)[x], IdxCvm = MG(IdxCv)[x - st];
  double DY_dxBu = 0., DX_dyBu = 0., IdyCvp = 0., IdyCv0 = 0., IdxCup = 0., IdxCu0 = 0., mfac = 0.;
  if (do_q) {
    DY_dxBu = PLN(HV_DY_dxBu)[x]; DX_dyBu = PLN(HV_DX_dyBu)[x];
    IdyCvp = MG(IdyCv)[x + 1]; IdyCv0 = MG(IdyCv)[x]; IdxCup = MG(IdxCu)[x + st]; IdxCu0 = MG(IdxCu)[x];
    const double mBu = MG(mask2dBu)[x];
    mfac = no_slip ? (2.0 - mBu) : mBu;
  }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double u0 = u[c], v0 = v[c];
    if (do_h) {
      const double dudx = DY_dxT * ((IdyCu0 * u0) - (IdyCum * u[c - 1]));
      const double dvdy = DX_dyT * ((IdxCv0 * v0) - (IdxCvm * v[c - st]));
      sh_xx[c] = dudx - dvdy;
    }
    if (do_q) {
      const double dvdx = DY_dxBu * ((v[c + 1] * IdyCvp) - (v0 * IdyCv0));
      const double dudy = DX_dyBu * ((u[c + st] * IdxCup) - (u0 * IdxCu0));
      sh_xy[c] = mfac * (dvdx + dudy);
    }
  }
}

// stage 2 :934-943: Del2u on (Isq-1..Ieq+1, js-1..Jeq+1), Del2v on (is-1..Ieq+1, Jsq-1..Jeq+1)
__global__ void __launch_bounds__(256)
k_hv_del2(Dm d, const double *__restrict__ G, const double *__restrict__ P, const double *__restrict__ sh_xx,
          const double *__restrict__ sh_xy, double *__restrict__ Del2u, double *__restrict__ Del2v) {
  const int i = I_BASE(-2) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -2 || i > d.ni || j > d.nj) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_u = (j >= -1);                 // I = -2..ni, j = -1..nj
  const bool do_v = (i >= -1);                 // i = -1..ni, J = -2..nj
  const double Idx2dyCu = PLN(HV_Idx2dyCu)[x], Idxdy2u = PLN(HV_Idxdy2u)[x], Idx2dyCv = PLN(HV_Idx2dyCv)[x], Idxdy2v = PLN(HV_Idxdy2v)[x];
  const double dx2q0 = PLN(HV_dx2q)[x], dy2q0 = PLN(HV_dy2q)[x];
  const double dx2q_s = do_u ? PLN(HV_dx2q)[x - st] : 0., dy2q_w = do_v ? PLN(HV_dy2q)[x - 1] : 0.;
  const double dy2h0 = PLN(HV_dy2h)[x], dy2h_e = PLN(HV_dy2h)[x + 1], dx2h0 = PLN(HV_dx2h)[x], dx2h_n = PLN(HV_dx2h)[x + st];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double sxy = sh_xy[c], sxx = sh_xx[c];
    if (do_u) Del2u[c] = Idx2dyCu * ((dx2q0 * sxy) - (dx2q_s * sh_xy[c - st])) + Idxdy2u * ((dy2h_e * sh_xx[c + 1]) - (dy2h0 * sxx));
    if (do_v) Del2v[c] = Idxdy2v * ((dy2q0 * sxy) - (dy2q_w * sh_xy[c - 1])) - Idx2dyCv * ((dx2h_n * sh_xx[c + st]) - (dx2h0 * sxx));
  }
}

// G%dF_dx, G%dF_dy of MOM_calculate_grad_Coriolis (MOM_shared_initialization.F90:91-122) over the computational domain; the
// halo update that follows (pass_vector, AGRID) is the caller's.
__global__ void __launch_bounds__(256)
k_hv_grad_Coriolis(Dm d, const double *__restrict__ G, double *__restrict__ P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double *fBu = MG(CoriolisBu);
  double f1 = 0.5 * (fBu[x] + fBu[x - st]), f2 = 0.5 * (fBu[x - 1] + fBu[x - 1 - st]);
  PLN(HV_dF_dx)[x] = MG(IdxT)[x] * (f1 - f2);
  f1 = 0.5 * (fBu[x] + fBu[x - 1]); f2 = 0.5 * (fBu[x - st] + fBu[x - 1 - st]);
  PLN(HV_dF_dy)[x] = MG(IdyT)[x] * (f1 - f2);
}

// Leith viscosities, stage L1 :730-733, :961-972, :1029-1031: the vertical vorticity at q points (is-3..Ieq+2, js-3..Jeq+2)
// and, for MODIFIED_LEITH, the divergence at h points (Isq-1..Ieq+2, Jsq-1..Jeq+2).
__global__ void __launch_bounds__(256)
k_hv_vort(Dm d, const double *__restrict__ G, const double *__restrict__ P, const double *__restrict__ u,
          const double *__restrict__ v, double *__restrict__ vort, double *__restrict__ divx, int no_slip, int modified) {
  const int i = I_BASE(-3) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -3 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -3 || i > d.ni + 1 || j > d.nj + 1) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_h = modified && (i >= -2) && (j >= -2);
  // CS%DY_dxBu, CS%DX_dyBu are set on (is-2..Ieq+1, js-2..Jeq+1) only and zero beyond (hor_visc_init :2869): the
  // outermost ring of the vorticity is zero in the reference, and here
  const double DY_dxBu = PLN(HV_DY_dxBu)[x], DX_dyBu = PLN(HV_DX_dyBu)[x];
  const double IdyCvp = MG(IdyCv)[x + 1], IdyCv0 = MG(IdyCv)[x], IdxCup = MG(IdxCu)[x + st], IdxCu0 = MG(IdxCu)[x];
  const double mBu = MG(mask2dBu)[x], mfac = no_slip ? (2.0 - mBu) : mBu;
  double DY_dxT = 0., DX_dyT = 0., IdyCum = 0., IdxCvm = 0.;
  if (do_h) { DY_dxT = PLN(HV_DY_dxT)[x]; DX_dyT = PLN(HV_DX_dyT)[x]; IdyCum = MG(IdyCu)[x - 1]; IdxCvm = MG(IdxCv)[x - st]; }
  const double IdyCu0 = MG(IdyCu)[x], IdxCv0 = MG(IdxCv)[x];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double u0 = u[c], v0 = v[c];
    const double dvdx = DY_dxBu * ((v[c + 1] * IdyCvp) - (v0 * IdyCv0));
    const double dudy = DX_dyBu * ((u[c + st] * IdxCup) - (u0 * IdxCu0));
    vort[c] = mfac * (dvdx - dudy);
    if (do_h) {
      const double dudx = DY_dxT * ((IdyCu0 * u0) - (IdyCum * u[c - 1]));
      const double dvdy = DX_dyT * ((IdxCv0 * v0) - (IdxCvm * v[c - st]));
      divx[c] = dudx + dvdy;
    }
  }
}

// stage L2 :987-1111: from the vorticity (and the divergence) the three fields the viscosities need --
//   LD = Del2vort_q on (is_Kh-1..ie_Kh, js_Kh-1..je_Kh) = (-2..ni, -2..nj)
//   LH = grad_vort_mag_h + grad_div_mag_h on (-1..ni, -1..nj),  LQ = grad_vort_mag_q + grad_div_mag_q on (-1..ni-1, -1..nj-1).
// vort_xy_dx / vort_xy_dy are re-formed where they are used (two vorticities each); beta joins them after Del2vort_q has
// been taken (:1069-1076) -- every gradient LH and LQ look at lies inside the ranges the reference adds beta on.
__global__ void __launch_bounds__(256)
k_hv_leith(Dm d, const double *__restrict__ G, const double *__restrict__ P, const double *__restrict__ vort,
           const double *__restrict__ divx, double *__restrict__ LD, double *__restrict__ LH, double *__restrict__ LQ,
           int modified, int beta) {
  const int i = I_BASE(-2) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -2 || i > d.ni || j > d.nj) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_h = (i >= -1) && (j >= -1), do_q = do_h && (i <= d.ni - 1) && (j <= d.nj - 1);
  const double *IdyCu = MG(IdyCu), *IdxCv = MG(IdxCv), *IdyCv = MG(IdyCv), *IdxCu = MG(IdxCu);
  auto DYX = [&](size_t y) { return MG(dyBu)[y] * MG(IdxBu)[y]; };   // G%dyBu * G%IdxBu, as :991 re-forms it
  auto DXY = [&](size_t y) { return MG(dxBu)[y] * MG(IdyBu)[y]; };
  // vort_xy_dx(i,J) at the v points x, x+1, x-st, and vort_xy_dy(I,j) at the u points x, x+st, x-1: coefficients
  const double ax0 = DYX(x), ax0c = IdyCu[x], ax0w = IdyCu[x - 1];
  const double axE = DYX(x + 1), axEc = IdyCu[x + 1];
  const double axS = DYX(x - st), axSc = IdyCu[x - st], axSw = IdyCu[x - st - 1];
  const double ay0 = DXY(x), ay0c = IdxCv[x], ay0s = IdxCv[x - st];
  const double ayN = DXY(x + st), ayNc = IdxCv[x + st];
  const double ayW = DXY(x - 1), ayWc = IdxCv[x - 1], ayWs = IdxCv[x - 1 - st];
  const double IdyCv0 = IdyCv[x], IdyCvE = IdyCv[x + 1], IdyCu0 = IdyCu[x], IdyCuN = IdyCu[x + st];
  double bx0 = 0., bxE = 0., bxS = 0., by0 = 0., byN = 0., byW = 0.;
  if (beta && do_h) {
    const double *Fx = PLN(HV_dF_dx), *Fy = PLN(HV_dF_dy);
    bx0 = 0.5 * (Fx[x] + Fx[x + st]); bxE = 0.5 * (Fx[x + 1] + Fx[x + 1 + st]); bxS = 0.5 * (Fx[x - st] + Fx[x]);
    by0 = 0.5 * (Fy[x] + Fy[x + 1]); byN = 0.5 * (Fy[x + st] + Fy[x + st + 1]); byW = 0.5 * (Fy[x - 1] + Fy[x]);
  }
  const double IdxCu0 = IdxCu[x], IdxCuW = IdxCu[x - 1], IdxCuN2 = IdxCu[x + st];
  const double IdyCv00 = IdyCv[x], IdyCvS = IdyCv[x - st], IdyCvE2 = IdyCv[x + 1];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double w0 = vort[c], wW = vort[c - 1], wE = vort[c + 1], wS = vort[c - st], wN = vort[c + st];
    // gradients without beta
    const double vdx0 = ax0 * ((w0 * ax0c) - (wW * ax0w));                    // vort_xy_dx(i, J)
    const double vdxE = axE * ((wE * axEc) - (w0 * ax0c));                    // vort_xy_dx(i+1, J)
    const double vdy0 = ay0 * ((w0 * ay0c) - (wS * ay0s));                    // vort_xy_dy(I, j)
    const double vdyN = ayN * ((wN * ayNc) - (w0 * ay0c));                    // vort_xy_dy(I, j+1)
    LD[c] = ax0 * ((vdxE * IdyCvE) - (vdx0 * IdyCv0)) + ay0 * ((vdyN * IdyCuN) - (vdy0 * IdyCu0));   // :1017-1023
    if (do_h) {
      const double wSW = vort[c - st - 1];
      const double vdxS = axS * ((wS * axSc) - (wSW * axSw));                 // vort_xy_dx(i, J-1)
      const double vdyW = ayW * ((wW * ayWc) - (wSW * ayWs));                 // vort_xy_dy(I-1, j)
      double gdh = 0., gdq = 0.;
      if (modified) {                                                        // :1033-1049
        const double d0 = divx[c], dE = divx[c + 1], dW = divx[c - 1], dN = divx[c + st], dS = divx[c - st];
        const double ddx0 = IdxCu0 * (dE - d0), ddxW = IdxCuW * (d0 - dW);
        const double ddy0 = IdyCv00 * (dN - d0), ddyS = IdyCvS * (d0 - dS);
        const double a = 0.5 * (ddx0 + ddxW), b = 0.5 * (ddy0 + ddyS);
        gdh = sqrt((a * a) + (b * b));
        if (do_q) {
          const double dNE = divx[c + st + 1];
          const double ddxN = IdxCuN2 * (dNE - dN), ddyE = IdyCvE2 * (dNE - dE);
          const double a2 = 0.5 * (ddx0 + ddxN), b2 = 0.5 * (ddy0 + ddyE);
          gdq = sqrt((a2 * a2) + (b2 * b2));
        }
      }
      const double gx0 = beta ? vdx0 + bx0 : vdx0, gxS = beta ? vdxS + bxS : vdxS, gxE = beta ? vdxE + bxE : vdxE;
      const double gy0 = beta ? vdy0 + by0 : vdy0, gyW = beta ? vdyW + byW : vdyW, gyN = beta ? vdyN + byN : vdyN;
      {
        const double a = 0.5 * (gx0 + gxS), b = 0.5 * (gy0 + gyW);           // :1104-1107
        LH[c] = sqrt((a * a) + (b * b)) + gdh;
      }
      if (do_q) {
        const double a = 0.5 * (gx0 + gxE), b = 0.5 * (gy0 + gyN);           // :1108-1111
        LQ[c] = sqrt((a * a) + (b * b)) + gdq;
      }
    }
  }
}

// h_u, h_v :767-781 from the thicknesses and T-point masks of the two cells
__device__ __forceinline__ double hface2(double ha, double hb, double ma, double mb, int land_mask) {
  if (land_mask) return 0.5 * (ma * ha + mb * hb);
  return 0.5 * (ha + hb);
}
__device__ __forceinline__ double hface(const double *__restrict__ h, const double *__restrict__ mT, size_t c, size_t c2, int s,
                                        int land_mask) {
  return hface2(h[c], h[c + s], mT[c2], mT[c2 + s], land_mask);
}

// stage 3: str_xx at h points (Isq..Ieq+1, Jsq..Jeq+1) :1112-1448, :1893 and str_xy at q points (is-1..Ieq, js-1..Jeq)
// :1483-1826, :1896-1906 -- both already multiplied by the thickness and reduction factors.  Column walk: the ~40
// coefficient values of the two points stay in registers over KCHUNK layers.
__global__ void __launch_bounds__(256)
k_hv_stress(Dm d, const double *__restrict__ G, const double *__restrict__ P, mom6x_hor_visc_params CS,
            const double *__restrict__ h, const double *__restrict__ sh_xx, const double *__restrict__ sh_xy,
            const double *__restrict__ Del2u, const double *__restrict__ Del2v, double *__restrict__ str_xx,
            double *__restrict__ str_xy, double h_neglect, const double *__restrict__ LD, const double *__restrict__ LH,
            const double *__restrict__ LQ) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni || j > d.nj) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool smag = CS.Smagorinsky_Kh || CS.Smagorinsky_Ah, better = CS.better_bound_Ah || CS.better_bound_Kh;
  const bool legacy_bound = (CS.Smagorinsky_Kh || CS.Leith_Kh) && (CS.bound_Kh && !CS.better_bound_Kh);   // :556-557
  const bool lap = CS.Laplacian, bih = CS.biharmonic;
  const double pi_ = 4.0 * atan(1.0), inv_PI3 = 1.0 / (pi_ * pi_ * pi_), inv_PI6 = inv_PI3 * inv_PI3;            // :481-483
  const double Lap3_xx = CS.Leith_Kh ? PLN(HV_Lap3_xx)[x] : 0., Bih6_xx = CS.Leith_Ah ? PLN(HV_Bih6_xx)[x] : 0.;
  const bool do_q = (i <= d.ni - 1 && j <= d.nj - 1);
  const double h_neglect3 = h_neglect * h_neglect * h_neglect;
  const int lm = CS.use_land_mask;
  const double *mT = MG(mask2dT);
  // T-point masks of the 3 x 3 block the two points touch (only with the land mask)
  double m00 = 1., mE = 1., mW = 1., mN = 1., mS = 1., mNE = 1.;
  if (lm) { m00 = mT[x]; mE = mT[x + 1]; mW = mT[x - 1]; mN = mT[x + st]; mS = mT[x - st]; mNE = mT[x + st + 1]; }
  // h point
  const double red_xx = PLN(HV_red_xx)[x];
  const double Kh_bg_xx = lap ? PLN(HV_Kh_bg_xx)[x] : 0., Kh_Max_xx = lap ? PLN(HV_Kh_Max_xx)[x] : 0.;
  const double Lap2_xx = CS.Smagorinsky_Kh ? PLN(HV_Lap2_xx)[x] : 0.;
  const double Ah_bg_xx = bih ? PLN(HV_Ah_bg_xx)[x] : 0., Ah_Max_xx = bih ? PLN(HV_Ah_Max_xx)[x] : 0.;
  const double Bih_xx = CS.Smagorinsky_Ah ? PLN(HV_Bih_xx)[x] : 0., Bih2_xx = CS.bound_Coriolis ? PLN(HV_Bih2_xx)[x] : 0.;
  const double IdyCu0 = bih ? MG(IdyCu)[x] : 0., IdyCuW = bih ? MG(IdyCu)[x - 1] : 0.;
  const double IdxCv0 = bih ? MG(IdxCv)[x] : 0., IdxCvS = bih ? MG(IdxCv)[x - st] : 0.;
  const double DY_dxT = bih ? PLN(HV_DY_dxT)[x] : 0., DX_dyT = bih ? PLN(HV_DX_dyT)[x] : 0.;
  // q point
  double red_xy = 0., Kh_bg_xy = 0., Kh_Max_xy = 0., Lap2_xy = 0., Ah_bg_xy = 0., Ah_Max_xy = 0., Bih_xy = 0., Bih2_xy = 0.;
  double Lap3_xy = 0., Bih6_xy = 0.;
  double DY_dxBu = 0., DX_dyBu = 0., IdyCvE = 0., IdyCv0 = 0., IdxCuN = 0., IdxCu0 = 0., mBu = 0., mu0 = 0., mu1 = 0., mv0 = 0., mv1 = 0.;
  if (do_q) {
    red_xy = PLN(HV_red_xy)[x]; mBu = MG(mask2dBu)[x];
    if (lap) { Kh_bg_xy = PLN(HV_Kh_bg_xy)[x]; Kh_Max_xy = PLN(HV_Kh_Max_xy)[x]; }
    if (CS.Smagorinsky_Kh) Lap2_xy = PLN(HV_Lap2_xy)[x];
    if (CS.Leith_Kh) Lap3_xy = PLN(HV_Lap3_xy)[x];
    if (CS.Leith_Ah) Bih6_xy = PLN(HV_Bih6_xy)[x];
    if (bih) {
      Ah_bg_xy = PLN(HV_Ah_bg_xy)[x]; Ah_Max_xy = PLN(HV_Ah_Max_xy)[x];
      DY_dxBu = PLN(HV_DY_dxBu)[x]; DX_dyBu = PLN(HV_DX_dyBu)[x];
      IdyCvE = MG(IdyCv)[x + 1]; IdyCv0 = MG(IdyCv)[x]; IdxCuN = MG(IdxCu)[x + st]; IdxCu0 = MG(IdxCu)[x];
    }
    if (CS.Smagorinsky_Ah) Bih_xy = PLN(HV_Bih_xy)[x];
    if (CS.bound_Coriolis) Bih2_xy = PLN(HV_Bih2_xy)[x];
    if (CS.no_slip) { mu0 = MG(mask2dCu)[x]; mu1 = MG(mask2dCu)[x + st]; mv0 = MG(mask2dCv)[x]; mv1 = MG(mask2dCv)[x + 1]; }
  }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double h00 = h[c], hE = h[c + 1], hN = h[c + st];
    // face thicknesses shared by the two points: h_u(I,j), h_v(i,J)
    const double hu0 = hface2(h00, hE, m00, mE, lm), hv0 = hface2(h00, hN, m00, mN, lm);
    const double sxx = sh_xx[c], sxy = sh_xy[c];
    {   // ---- h point (always inside Isq..Ieq+1, Jsq..Jeq+1)
      double Shear = 0., hrat = 0., vbr = 0., sxx_out;
      if (smag) {
        const double sh_xx_sq = sxx * sxx;
        const double a = sh_xy[c - 1 - st], e = sh_xy[c - 1], f = sh_xy[c - st];
        const double sh_xy_sq = 0.25 * (((a * a) + (sxy * sxy)) + ((e * e) + (f * f)));
        Shear = sqrt(sh_xx_sq + sh_xy_sq);
      }
      if (better) {
        const double h_min = dmin4(hu0, hface2(h[c - 1], h00, mW, m00, lm), hv0, hface2(h[c - st], h00, mS, m00, lm));
        hrat = dmin(1.0, h_min / (h00 + h_neglect));
      }
      if (lap) {
        double K = Kh_bg_xx;
        const double vvm = CS.Leith_Kh ? LH[c] : 0.0;
        if (CS.add_LES_viscosity) {
          if (CS.Smagorinsky_Kh) K = K + Lap2_xx * Shear;
          if (CS.Leith_Kh) K = K + Lap3_xx * vvm * inv_PI3;
        } else {
          if (CS.Smagorinsky_Kh) K = dmax(K, Lap2_xx * Shear);
          if (CS.Leith_Kh) K = dmax(K, Lap3_xx * vvm * inv_PI3);
        }
        if (legacy_bound) K = dmin(K, Kh_Max_xx);
        K = dmax(K, CS.Kh_bg_min);
        if (CS.better_bound_Kh && CS.better_bound_Ah) {
          vbr = 1.0;
          const double Kh_max_here = hrat * Kh_Max_xx;
          if (K >= Kh_max_here) { vbr = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (CS.backscatter_underbound && (Kh_max_here > 0.0))) vbr = 1.0 - K / Kh_max_here;
        } else if (CS.better_bound_Kh) {
          K = dmin(K, hrat * Kh_Max_xx);
        }
        sxx_out = -K * sxx;
      } else sxx_out = 0.0;
      if (bih) {
        double A = Ah_bg_xx;
        if (CS.Smagorinsky_Ah || CS.Leith_Ah) {   // :1301-1385
          if (CS.Smagorinsky_Ah) {
            double AhSm;
            if (CS.bound_Coriolis) AhSm = Shear * (Bih_xx + Bih2_xx * Shear);
            else AhSm = Bih_xx * Shear;
            A = dmax(A, AhSm);
          }
          if (CS.Leith_Ah) {
            const double Del2vort_h = 0.25 * ((LD[c] + LD[c - 1 - st]) + (LD[c - 1] + LD[c - st]));
            A = dmax(A, Bih6_xx * fabs(Del2vort_h) * inv_PI6);
          }
          if (CS.bound_Ah && !CS.better_bound_Ah) A = dmin(A, Ah_Max_xx);
        }
        if (CS.better_bound_Ah) {
          if (CS.better_bound_Kh) A = dmin(A, vbr * hrat * Ah_Max_xx);
          else A = dmin(A, hrat * Ah_Max_xx);
        }
        const double d_del2u = (IdyCu0 * Del2u[c]) - (IdyCuW * Del2u[c - 1]);
        const double d_del2v = (IdxCv0 * Del2v[c]) - (IdxCvS * Del2v[c - st]);
        const double d_str = A * ((DY_dxT * d_del2u) - (DX_dyT * d_del2v));
        sxx_out = sxx_out + d_str;
      }
      str_xx[c] = sxx_out * (h00 * red_xx);
    }
    if (do_q) {   // ---- q point (is-1..Ieq, js-1..Jeq)
      double Shear = 0., hrat = 0., vbr = 0., sxy_out;
      if (smag) {
        const double sh_xy_sq = sxy * sxy;
        const double b = sh_xx[c + 1 + st], e = sh_xx[c + st], f = sh_xx[c + 1];
        const double sh_xx_sq = 0.25 * (((sxx * sxx) + (b * b)) + ((e * e) + (f * f)));
        Shear = sqrt(sh_xy_sq + sh_xx_sq);
      }
      const double hNE = h[c + st + 1];
      const double hu1 = hface2(hN, hNE, mN, mNE, lm), hv1 = hface2(hE, hNE, mE, mNE, lm);
      const double h2uq = 4.0 * (hu0 * hu1), h2vq = 4.0 * (hv0 * hv1);
      double hq = (2.0 * (h2uq * h2vq)) / (h_neglect3 + (h2uq + h2vq) * ((hu0 + hu1) + (hv0 + hv1)));
      if (better) {
        const double h_min = dmin4(hu0, hu1, hv0, hv1);
        hrat = dmin(1.0, h_min / (hq + h_neglect));
      }
      if (CS.no_slip && (mBu < 0.5)) {
        if ((mu0 + mu1) + (mv0 + mv1) > 0.0) {
          const double hu = mu0 * hu0 + mu1 * hu1;
          const double hv = mv0 * hv0 + mv1 * hv1;
          if ((mu0 + mu1) * (mv0 + mv1) == 0.0) {
            hq = hu + hv;
            hrat = 1.0;
          } else {
            hq = 2.0 * (hu * hv) / ((hu + hv) + h_neglect);
            hrat = dmin(1.0, dmin(hu, hv) / (hq + h_neglect));
          }
        }
      }
      if (lap) {
        double K = Kh_bg_xy;
        if (CS.Smagorinsky_Kh) {
          if (CS.add_LES_viscosity) K = K + Lap2_xy * Shear;
          else K = dmax(K, Lap2_xy * Shear);
        }
        if (CS.Leith_Kh) {   // :1610-1620
          if (CS.add_LES_viscosity) K = K + Lap3_xy * LQ[c] * inv_PI3;
          else K = dmax(K, Lap3_xy * LQ[c] * inv_PI3);
        }
        if (legacy_bound) K = dmin(K, Kh_Max_xy);
        K = dmax(K, CS.Kh_bg_min);
        if (CS.better_bound_Kh && CS.better_bound_Ah) {
          vbr = 1.0;
          const double Kh_max_here = hrat * Kh_Max_xy;
          if (K >= Kh_max_here) { vbr = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (CS.backscatter_underbound && (Kh_max_here > 0.0))) vbr = 1.0 - K / Kh_max_here;
        } else if (CS.better_bound_Kh) {
          K = dmin(K, hrat * Kh_Max_xy);
        }
        sxy_out = -K * sxy;
      } else sxy_out = 0.;
      if (bih) {
        double A = Ah_bg_xy;
        if (CS.Smagorinsky_Ah || CS.Leith_Ah) {   // :1745-1773
          if (CS.Smagorinsky_Ah) {
            double AhSm;
            if (CS.bound_Coriolis) AhSm = Shear * (Bih_xy + Bih2_xy * Shear);
            else AhSm = Bih_xy * Shear;
            A = dmax(A, AhSm);
          }
          if (CS.Leith_Ah) A = dmax(A, Bih6_xy * fabs(LD[c]) * inv_PI6);
          if (CS.bound_Ah && !CS.better_bound_Ah) A = dmin(A, Ah_Max_xy);
        }
        if (CS.better_bound_Ah) {
          if (CS.better_bound_Kh) A = dmin(A, vbr * hrat * Ah_Max_xy);
          else A = dmin(A, hrat * Ah_Max_xy);
        }
        const double dDel2vdx = DY_dxBu * ((Del2v[c + 1] * IdyCvE) - (Del2v[c] * IdyCv0));
        const double dDel2udy = DX_dyBu * ((Del2u[c + st] * IdxCuN) - (Del2u[c] * IdxCu0));
        const double d_str = A * (dDel2vdx + dDel2udy);
        sxy_out = sxy_out + d_str;
      }
      if (CS.no_slip) str_xy[c] = sxy_out * (hq * red_xy);
      else str_xy[c] = sxy_out * (hq * mBu * red_xy);
    }
  }
}

// stage 4 :1910-1931: diffu on (Isq..Ieq, js..je), diffv on (is..ie, Jsq..Jeq)
__global__ void __launch_bounds__(256)
k_hv_accel(Dm d, const double *__restrict__ G, const double *__restrict__ P, const double *__restrict__ h,
           const double *__restrict__ str_xx, const double *__restrict__ str_xy, double *__restrict__ diffu,
           double *__restrict__ diffv, int land_mask, double h_neglect) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < -1 || i > d.ni - 1 || j > d.nj - 1) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_u = (j >= 0), do_v = (i >= 0);
  const double *mT = MG(mask2dT);
  const double IdxCu = MG(IdxCu)[x], IdyCu = MG(IdyCu)[x], IareaCu = MG(IareaCu)[x];
  const double IdxCv = MG(IdxCv)[x], IdyCv = MG(IdyCv)[x], IareaCv = MG(IareaCv)[x];
  const double dx2q0 = PLN(HV_dx2q)[x], dy2q0 = PLN(HV_dy2q)[x];
  const double dx2q_s = do_u ? PLN(HV_dx2q)[x - st] : 0., dy2q_w = do_v ? PLN(HV_dy2q)[x - 1] : 0.;
  const double dy2h0 = PLN(HV_dy2h)[x], dy2h_e = PLN(HV_dy2h)[x + 1], dx2h0 = PLN(HV_dx2h)[x], dx2h_n = PLN(HV_dx2h)[x + st];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double sxy = str_xy[c], sxx = str_xx[c];
    if (do_u) {
      const double h_u = hface(h, mT, c, x, 1, land_mask);
      diffu[c] = ((IdxCu * ((dx2q_s * str_xy[c - st]) - (dx2q0 * sxy)) + IdyCu * ((dy2h0 * sxx) - (dy2h_e * str_xx[c + 1]))) * IareaCu) /
                 (h_u + h_neglect);
    }
    if (do_v) {
      const double h_v = hface(h, mT, c, x, st, land_mask);
      diffv[c] = ((IdyCv * ((dy2q_w * str_xy[c - 1]) - (dy2q0 * sxy)) - IdxCv * ((dx2h0 * sxx) - (dx2h_n * str_xx[c + st]))) * IareaCv) /
                 (h_v + h_neglect);
    }
  }
}


// ---- the four stages in ONE kernel (no Leith) ------------------------------------------------------------------------------
// A work-group owns a tile of HT_X x HT_Y points and walks a chunk of layers; a THREAD owns one point (its h point (i,j), u point
// (I,j), v point (i,J) and q point (I,J)) and keeps the ~30 coefficient values of that point in registers for the whole chunk --
// the register file (512 KB per CU) is the only on-chip store big enough for them.  Per layer everything goes through LDS:
//     u, v, h (ONE global load each per point, fetched a layer ahead) -> | -> sh_xx, sh_xy -> | -> Del2u, Del2v -> | -> str_xx,
//     str_xy -> | -> diffu, diffv (global)
// with three barriers (the next layer's u, v, h go into the second of two input buffers before the last one).  Every stage is
// computed on the whole tile; a result is only valid where its inputs were: the outputs of the last stage are good on the tile
// minus a frame of HT_H = 2 points (see below), and that is all that is stored.  The four metric planes del2 and accel read at neighbouring
// points (dx2q, dy2q, dy2h, dx2h) sit in LDS for the life of the work-group.  Arithmetic: the expressions of k_hv_strain,
// k_hv_del2, k_hv_stress and k_hv_accel, unchanged -- results are bit-identical with the four-kernel path
// (MOM6X_HORVISC=legacy; Leith always takes the four kernels).  3 reads (x the tile's halo overhead) + 2 writes per cell-layer
// instead of 20.
#define HT_H 2
// OM4: the switches of the OM4-class configuration (LAPLACIAN + BIHARMONIC with SMAGORINSKY_AH, both "better" bounds, land mask,
// free slip, no LES addition) known at COMPILE time: the generic kernel keeps ~170 values live because every option's
// coefficients are loaded and every branch is present; any other combination takes the generic instantiation.
#define OM4_better_bound_Kh 1
#define OM4_better_bound_Ah 1
#define OM4_Smagorinsky_Kh 0
#define OM4_Smagorinsky_Ah 1
#define OM4_no_slip 0
#define OM4_bound_Coriolis 0
#define OM4_bound_Ah 1
#define OM4_backscatter_underbound 1
#define OM4_add_LES_viscosity 0
#define OM4_use_land_mask 1
#define OM4_bound_Kh 1
#define OM4_biharmonic 1
#define OM4_Laplacian 1
#define HVF(f) (OM4 ? (OM4_##f != 0) : (CS.f != 0))
#ifndef HV_OM4_W4
#define HV_OM4_W4 0
#endif
#ifndef HV_TX
#define HV_TX 32
#endif
#ifndef HV_TY
#define HV_TY 24
#endif
#ifndef HV_WAVES   // wavefronts per SIMD the kernel is compiled for (0: from the tile size)
#define HV_WAVES 0
#endif
template <int HT_X, int HT_Y, bool OM4>
__global__ void __launch_bounds__(HT_X * HT_Y, HV_WAVES ? HV_WAVES : ((HT_X * HT_Y >= 1024 || (OM4 && HV_OM4_W4)) ? 4 : ((HT_X * HT_Y >= 768) ? 3 : 2)))   // 1024 threads: 4 wavefronts per SIMD (128 registers, 56 spilled); 768: 3 per SIMD (168 registers); 512: 2 per SIMD
k_hv_fused(Dm d, const double *__restrict__ G, const double *__restrict__ P, mom6x_hor_visc_params CS,
           const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ h,
           double *__restrict__ diffu, double *__restrict__ diffv, double h_neglect, int kc, int gx, int gy, int gz, int xcd_order) {
  constexpr int HT_LDW = HT_X + 2, HT_LDN = (HT_Y + 2) * HT_LDW;
  // Work-groups go to the eight XCDs round robin: with xcd_order XCD n walks a CONTIGUOUS run of tiles (x fastest), so that the
  // 128-byte lines two neighbouring tiles share (a tile row is 32 doubles at an offset of 28 n - 3: three lines for 224 useful
  // bytes) meet in ONE L2 instead of being fetched by two (as k_corad_lds, dyn_kernels.hip)
  int bid = (int)blockIdx.x;
  const int nb = gx * gy * gz;
  if (xcd_order) { const int per = (nb + 7) / 8; bid = (bid % 8) * per + bid / 8; }
  if (bid >= nb) return;
  const int bxi = bid % gx, byi = (bid / gx) % gy, bzi = bid / (gx * gy);
  extern __shared__ double lds[];
  double *s_xx = lds, *s_xy = lds + HT_LDN, *s_d2u = lds + 2 * HT_LDN, *s_d2v = lds + 3 * HT_LDN;
  double *s_txx = lds + 4 * HT_LDN, *s_txy = lds + 5 * HT_LDN;
  double *c_dx2q = lds + 6 * HT_LDN, *c_dy2q = lds + 7 * HT_LDN, *c_dy2h = lds + 8 * HT_LDN, *c_dx2h = lds + 9 * HT_LDN;
  double *s_in = lds + 10 * HT_LDN;                     // u, v, h of the layer: two buffers of three planes
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int i = -1 - HT_H + bxi * (HT_X - 2 * HT_H) + tx;
  const int j = -1 - HT_H + byi * (HT_Y - 2 * HT_H) + ty;
  const int st = d.pitch;
  const size_t slab = (size_t)d.slab;
  const int k0 = bzi * kc, k1 = min(k0 + kc, d.nk);
  const int l = (ty + 1) * HT_LDW + (tx + 1);            // this point in the LDS planes (a frame of one keeps +-1 reads inside)
  // points whose coefficient stencils stay inside the planes (halo 4): everything the valid outputs need lies in -3..ni+2 /
  // -3..nj+2; the fields themselves are also loaded one point further out (the neighbours of those points)
  const bool live = (i >= -3) && (i <= d.ni + 2) && (j >= -3) && (j <= d.nj + 2);
  const bool loadable = (i >= -4) && (i <= d.ni + 3) && (j >= -4) && (j <= d.nj + 3);
  const size_t x = live ? ix2(d, i, j) : ix2(d, 0, 0);
  const size_t xf = loadable ? ix2(d, i, j) : ix2(d, 0, 0);
  const bool out_u = live && tx >= HT_H && tx < HT_X - HT_H && ty >= HT_H && ty < HT_Y - HT_H && (i <= d.ni - 1) && (j >= 0) && (j <= d.nj - 1);
  const bool out_v = live && tx >= HT_H && tx < HT_X - HT_H && ty >= HT_H && ty < HT_Y - HT_H && (i >= 0) && (i <= d.ni - 1) && (j <= d.nj - 1);
  // Where a stage is good, from what it reads (the inputs are there on the whole tile, 0 .. HT_X-1 / 0 .. HT_Y-1):
  //   sh_xx (u, v to the west / south): >= 1          sh_xy (u, v to the north / east): <= HT-2
  //   Del2u, Del2v (sh_xy to the south / west, sh_xx to the east / north): 1 .. HT-2
  //   str_xx (sh_xy, Del2 to the west / south): 2 .. HT-2      str_xy (sh_xx, Del2 to the east / north): 1 .. HT-3
  //   diffu, diffv (str_xy to the south / west, str_xx to the east / north): 2 .. HT-3 -- a frame of two.
  // Stage 3 runs where stage 4 will look: str_xx at the outputs and one further east / north, str_xy one further west / south.
  const bool need_xx = live && tx >= HT_H && tx <= HT_X - HT_H && ty >= HT_H && ty <= HT_Y - HT_H;
  const bool need_xy = live && tx >= HT_H - 1 && tx < HT_X - HT_H && ty >= HT_H - 1 && ty < HT_Y - HT_H;
  const bool smag = HVF(Smagorinsky_Kh) || HVF(Smagorinsky_Ah), better = HVF(better_bound_Ah) || HVF(better_bound_Kh);
  const bool legacy_bound = HVF(Smagorinsky_Kh) && (HVF(bound_Kh) && !HVF(better_bound_Kh));   // :556-557 (no Leith here)
  const bool lap = HVF(Laplacian), bih = HVF(biharmonic);
  const double h_neglect3 = h_neglect * h_neglect * h_neglect;
  const int lm = HVF(use_land_mask);
  // ---- coefficients of this point, for all layers of the chunk
  c_dx2q[l] = PLN(HV_dx2q)[x]; c_dy2q[l] = PLN(HV_dy2q)[x]; c_dy2h[l] = PLN(HV_dy2h)[x]; c_dx2h[l] = PLN(HV_dx2h)[x];
  const double *mT = MG(mask2dT);
  double m00 = 1., mE = 1., mW = 1., mN = 1., mS = 1., mNE = 1.;
  if (lm) { m00 = mT[x]; mE = mT[x + 1]; mW = mT[x - 1]; mN = mT[x + st]; mS = mT[x - st]; mNE = mT[x + st + 1]; }
  const double DY_dxT = PLN(HV_DY_dxT)[x], DX_dyT = PLN(HV_DX_dyT)[x], DY_dxBu = PLN(HV_DY_dxBu)[x], DX_dyBu = PLN(HV_DX_dyBu)[x];
  const double IdyCu0 = MG(IdyCu)[x], IdyCuW = MG(IdyCu)[x - 1], IdxCv0 = MG(IdxCv)[x], IdxCvS = MG(IdxCv)[x - st];
  const double IdyCvE = MG(IdyCv)[x + 1], IdyCv0 = MG(IdyCv)[x], IdxCuN = MG(IdxCu)[x + st], IdxCu0 = MG(IdxCu)[x];
  const double mBu = MG(mask2dBu)[x], mfac = HVF(no_slip) ? (2.0 - mBu) : mBu;
  const double Idx2dyCu = PLN(HV_Idx2dyCu)[x], Idxdy2u = PLN(HV_Idxdy2u)[x], Idx2dyCv = PLN(HV_Idx2dyCv)[x], Idxdy2v = PLN(HV_Idxdy2v)[x];
  const double IareaCu = MG(IareaCu)[x], IareaCv = MG(IareaCv)[x];
  const double red_xx = PLN(HV_red_xx)[x], red_xy = PLN(HV_red_xy)[x];
  const double Kh_bg_xx = lap ? PLN(HV_Kh_bg_xx)[x] : 0., Kh_Max_xx = lap ? PLN(HV_Kh_Max_xx)[x] : 0.;
  const double Kh_bg_xy = lap ? PLN(HV_Kh_bg_xy)[x] : 0., Kh_Max_xy = lap ? PLN(HV_Kh_Max_xy)[x] : 0.;
  const double Lap2_xx = HVF(Smagorinsky_Kh) ? PLN(HV_Lap2_xx)[x] : 0., Lap2_xy = HVF(Smagorinsky_Kh) ? PLN(HV_Lap2_xy)[x] : 0.;
  const double Ah_bg_xx = bih ? PLN(HV_Ah_bg_xx)[x] : 0., Ah_Max_xx = bih ? PLN(HV_Ah_Max_xx)[x] : 0.;
  const double Ah_bg_xy = bih ? PLN(HV_Ah_bg_xy)[x] : 0., Ah_Max_xy = bih ? PLN(HV_Ah_Max_xy)[x] : 0.;
  const double Bih_xx = HVF(Smagorinsky_Ah) ? PLN(HV_Bih_xx)[x] : 0., Bih_xy = HVF(Smagorinsky_Ah) ? PLN(HV_Bih_xy)[x] : 0.;
  const double Bih2_xx = HVF(bound_Coriolis) ? PLN(HV_Bih2_xx)[x] : 0., Bih2_xy = HVF(bound_Coriolis) ? PLN(HV_Bih2_xy)[x] : 0.;
  double mu0 = 0., mu1 = 0., mv0 = 0., mv1 = 0.;
  if (HVF(no_slip)) { mu0 = MG(mask2dCu)[x]; mu1 = MG(mask2dCu)[x + st]; mv0 = MG(mask2dCv)[x]; mv1 = MG(mask2dCv)[x + 1]; }
  // the frame of the LDS planes is only ever read by points whose results are discarded: no initialisation needed, but the
  // stage planes of points that are not live must not hold NaN patterns that trap -- they cannot: nothing here traps
  __syncthreads();
  const double dx2q0 = c_dx2q[l], dx2q_s = c_dx2q[l - HT_LDW], dy2q0 = c_dy2q[l], dy2q_w = c_dy2q[l - 1];
  const double dy2h0 = c_dy2h[l], dy2h_e = c_dy2h[l + 1], dx2h0 = c_dx2h[l], dx2h_n = c_dx2h[l + HT_LDW];

  double un = u[xf + (size_t)k0 * slab], vn = v[xf + (size_t)k0 * slab], hn = h[xf + (size_t)k0 * slab];
  s_in[l] = un; s_in[HT_LDN + l] = vn; s_in[2 * HT_LDN + l] = hn;
  if (k0 + 1 < k1) { un = u[xf + (size_t)(k0 + 1) * slab]; vn = v[xf + (size_t)(k0 + 1) * slab]; hn = h[xf + (size_t)(k0 + 1) * slab]; }
  __syncthreads();
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double *su = s_in + ((k - k0) & 1) * 3 * HT_LDN, *sv = su + HT_LDN, *sh = su + 2 * HT_LDN;
    // ---- stage 1 :898-931: sh_xx at (i,j), sh_xy at (I,J)
    const double u0 = su[l], v0 = sv[l];
    const double h00 = sh[l], hE = sh[l + 1], hN = sh[l + HT_LDW], hNE = sh[l + HT_LDW + 1], hW = sh[l - 1], hS = sh[l - HT_LDW];
    double sxx, sxy;
    {
      const double dudx = DY_dxT * ((IdyCu0 * u0) - (IdyCuW * su[l - 1]));
      const double dvdy = DX_dyT * ((IdxCv0 * v0) - (IdxCvS * sv[l - HT_LDW]));
      sxx = dudx - dvdy;
      const double dvdx = DY_dxBu * ((sv[l + 1] * IdyCvE) - (v0 * IdyCv0));
      const double dudy = DX_dyBu * ((su[l + HT_LDW] * IdxCuN) - (u0 * IdxCu0));
      sxy = mfac * (dvdx + dudy);
    }
    s_xx[l] = sxx; s_xy[l] = sxy;
    __syncthreads();
    // ---- stage 2 :934-943: Del2u at (I,j), Del2v at (i,J)
    double d2u = 0., d2v = 0.;
    if (bih) {
      d2u = Idx2dyCu * ((dx2q0 * sxy) - (dx2q_s * s_xy[l - HT_LDW])) + Idxdy2u * ((dy2h_e * s_xx[l + 1]) - (dy2h0 * sxx));
      d2v = Idxdy2v * ((dy2q0 * sxy) - (dy2q_w * s_xy[l - 1])) - Idx2dyCv * ((dx2h_n * s_xx[l + HT_LDW]) - (dx2h0 * sxx));
      s_d2u[l] = d2u; s_d2v[l] = d2v;
    }
    __syncthreads();
    // ---- stage 3: str_xx at the h point :1112-1448, str_xy at the q point :1483-1826 (k_hv_stress) -- the expensive stage
    // (square roots, five divisions): only where stage 4 will look (its outputs' own points and one row / column around)
    const double hu0 = hface2(h00, hE, m00, mE, lm), hv0 = hface2(h00, hN, m00, mN, lm);
    if (need_xx) {
      double Shear = 0., hrat = 0., vbr = 0., sxx_out;
      if (smag) {
        const double sh_xx_sq = sxx * sxx;
        const double a = s_xy[l - 1 - HT_LDW], e = s_xy[l - 1], f = s_xy[l - HT_LDW];
        const double sh_xy_sq = 0.25 * (((a * a) + (sxy * sxy)) + ((e * e) + (f * f)));
        Shear = sqrt(sh_xx_sq + sh_xy_sq);
      }
      if (better) {
        const double h_min = dmin4(hu0, hface2(hW, h00, mW, m00, lm), hv0, hface2(hS, h00, mS, m00, lm));
        hrat = dmin(1.0, h_min / (h00 + h_neglect));
      }
      if (lap) {
        double K = Kh_bg_xx;
        if (HVF(add_LES_viscosity)) {
          if (HVF(Smagorinsky_Kh)) K = K + Lap2_xx * Shear;
        } else {
          if (HVF(Smagorinsky_Kh)) K = dmax(K, Lap2_xx * Shear);
        }
        if (legacy_bound) K = dmin(K, Kh_Max_xx);
        K = dmax(K, CS.Kh_bg_min);
        if (HVF(better_bound_Kh) && HVF(better_bound_Ah)) {
          vbr = 1.0;
          const double Kh_max_here = hrat * Kh_Max_xx;
          if (K >= Kh_max_here) { vbr = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (HVF(backscatter_underbound) && (Kh_max_here > 0.0))) vbr = 1.0 - K / Kh_max_here;
        } else if (HVF(better_bound_Kh)) {
          K = dmin(K, hrat * Kh_Max_xx);
        }
        sxx_out = -K * sxx;
      } else sxx_out = 0.0;
      if (bih) {
        double A = Ah_bg_xx;
        if (HVF(Smagorinsky_Ah)) {
          double AhSm;
          if (HVF(bound_Coriolis)) AhSm = Shear * (Bih_xx + Bih2_xx * Shear);
          else AhSm = Bih_xx * Shear;
          A = dmax(A, AhSm);
          if (HVF(bound_Ah) && !HVF(better_bound_Ah)) A = dmin(A, Ah_Max_xx);
        }
        if (HVF(better_bound_Ah)) {
          if (HVF(better_bound_Kh)) A = dmin(A, vbr * hrat * Ah_Max_xx);
          else A = dmin(A, hrat * Ah_Max_xx);
        }
        const double d_del2u = (IdyCu0 * d2u) - (IdyCuW * s_d2u[l - 1]);
        const double d_del2v = (IdxCv0 * d2v) - (IdxCvS * s_d2v[l - HT_LDW]);
        const double d_str = A * ((DY_dxT * d_del2u) - (DX_dyT * d_del2v));
        sxx_out = sxx_out + d_str;
      }
      s_txx[l] = sxx_out * (h00 * red_xx);
    }
    if (need_xy) {
      double Shear = 0., hrat = 0., vbr = 0., sxy_out;
      if (smag) {
        const double sh_xy_sq = sxy * sxy;
        const double b = s_xx[l + 1 + HT_LDW], e = s_xx[l + HT_LDW], f = s_xx[l + 1];
        const double sh_xx_sq = 0.25 * (((sxx * sxx) + (b * b)) + ((e * e) + (f * f)));
        Shear = sqrt(sh_xy_sq + sh_xx_sq);
      }
      const double hu1 = hface2(hN, hNE, mN, mNE, lm), hv1 = hface2(hE, hNE, mE, mNE, lm);
      const double h2uq = 4.0 * (hu0 * hu1), h2vq = 4.0 * (hv0 * hv1);
      double hq = (2.0 * (h2uq * h2vq)) / (h_neglect3 + (h2uq + h2vq) * ((hu0 + hu1) + (hv0 + hv1)));
      if (better) {
        const double h_min = dmin4(hu0, hu1, hv0, hv1);
        hrat = dmin(1.0, h_min / (hq + h_neglect));
      }
      if (HVF(no_slip) && (mBu < 0.5)) {
        if ((mu0 + mu1) + (mv0 + mv1) > 0.0) {
          const double hu = mu0 * hu0 + mu1 * hu1;
          const double hv = mv0 * hv0 + mv1 * hv1;
          if ((mu0 + mu1) * (mv0 + mv1) == 0.0) {
            hq = hu + hv;
            hrat = 1.0;
          } else {
            hq = 2.0 * (hu * hv) / ((hu + hv) + h_neglect);
            hrat = dmin(1.0, dmin(hu, hv) / (hq + h_neglect));
          }
        }
      }
      if (lap) {
        double K = Kh_bg_xy;
        if (HVF(Smagorinsky_Kh)) {
          if (HVF(add_LES_viscosity)) K = K + Lap2_xy * Shear;
          else K = dmax(K, Lap2_xy * Shear);
        }
        if (legacy_bound) K = dmin(K, Kh_Max_xy);
        K = dmax(K, CS.Kh_bg_min);
        if (HVF(better_bound_Kh) && HVF(better_bound_Ah)) {
          vbr = 1.0;
          const double Kh_max_here = hrat * Kh_Max_xy;
          if (K >= Kh_max_here) { vbr = 0.0; K = Kh_max_here; }
          else if ((K > 0.0) || (HVF(backscatter_underbound) && (Kh_max_here > 0.0))) vbr = 1.0 - K / Kh_max_here;
        } else if (HVF(better_bound_Kh)) {
          K = dmin(K, hrat * Kh_Max_xy);
        }
        sxy_out = -K * sxy;
      } else sxy_out = 0.;
      if (bih) {
        double A = Ah_bg_xy;
        if (HVF(Smagorinsky_Ah)) {
          double AhSm;
          if (HVF(bound_Coriolis)) AhSm = Shear * (Bih_xy + Bih2_xy * Shear);
          else AhSm = Bih_xy * Shear;
          A = dmax(A, AhSm);
          if (HVF(bound_Ah) && !HVF(better_bound_Ah)) A = dmin(A, Ah_Max_xy);
        }
        if (HVF(better_bound_Ah)) {
          if (HVF(better_bound_Kh)) A = dmin(A, vbr * hrat * Ah_Max_xy);
          else A = dmin(A, hrat * Ah_Max_xy);
        }
        const double dDel2vdx = DY_dxBu * ((s_d2v[l + 1] * IdyCvE) - (d2v * IdyCv0));
        const double dDel2udy = DX_dyBu * ((s_d2u[l + HT_LDW] * IdxCuN) - (d2u * IdxCu0));
        const double d_str = A * (dDel2vdx + dDel2udy);
        sxy_out = sxy_out + d_str;
      }
      if (HVF(no_slip)) s_txy[l] = sxy_out * (hq * red_xy);
      else s_txy[l] = sxy_out * (hq * mBu * red_xy);
    }
    if (k + 1 < k1) {   // the next layer's inputs into the other buffer (last read two barriers ago), and the layer after into flight
      double *nu = s_in + ((k + 1 - k0) & 1) * 3 * HT_LDN;
      nu[l] = un; nu[HT_LDN + l] = vn; nu[2 * HT_LDN + l] = hn;
      if (k + 2 < k1) { un = u[xf + (size_t)(k + 2) * slab]; vn = v[xf + (size_t)(k + 2) * slab]; hn = h[xf + (size_t)(k + 2) * slab]; }
    }
    __syncthreads();
    // ---- stage 4 :1910-1931: diffu at (I,j), diffv at (i,J)
    {
      const double txy = s_txy[l], txx = s_txx[l];
      if (out_u) {
        const double h_u = hface2(h00, hE, m00, mE, lm);
        diffu[c] = ((IdxCu0 * ((dx2q_s * s_txy[l - HT_LDW]) - (dx2q0 * txy)) + IdyCu0 * ((dy2h0 * txx) - (dy2h_e * s_txx[l + 1]))) * IareaCu) /
                   (h_u + h_neglect);
      }
      if (out_v) {
        const double h_v = hface2(h00, hN, m00, mN, lm);
        diffv[c] = ((IdyCv0 * ((dy2q_w * s_txy[l - 1]) - (dy2q0 * txy)) - IdxCv0 * ((dx2h0 * txx) - (dx2h_n * s_txx[l + HT_LDW]))) * IareaCv) /
                   (h_v + h_neglect);
      }
    }
    // (the next layer's stage 1 writes s_xx / s_xy, last read in stage 3 of this layer: a barrier lies in between;
    //  its stage 3 writes s_txx / s_txy, read here: two barriers lie in between)
  }
}

}  // namespace

void hor_visc_free(mom6x_ctx *c) { (void)hipFree(c->hv_planes); c->hv_planes = nullptr; }

extern "C" int mom6x_hor_visc_init(mom6x_ctx *c, const mom6x_hor_visc_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_hor_visc_init: null argument");
  REQUIRE(p->dt > 0.0, MOM6X_EINVAL, "hor_visc_init: DT must be positive");
  mom6x_hor_visc_params cs = *p;   // hor_visc_init :2465, :2495-2496, :2555-2571, :2624
  if (!cs.Laplacian) { cs.Smagorinsky_Kh = 0; cs.bound_Kh = 0; cs.better_bound_Kh = 0; }
  if (!cs.biharmonic) { cs.Smagorinsky_Ah = 0; cs.bound_Ah = 0; cs.better_bound_Ah = 0; }
  if (!cs.Smagorinsky_Ah) cs.bound_Coriolis = 0;
  if (!cs.Laplacian) cs.Leith_Kh = 0;     // :2473
  if (!cs.biharmonic) cs.Leith_Ah = 0;    // :2561
  REQUIRE(!(cs.no_slip && cs.biharmonic), MOM6X_EINVAL,
          "ERROR: NOSLIP and BIHARMONIC cannot be defined at the same time in MOM.");
  REQUIRE(c->dims.halo >= 3, MOM6X_EINVAL, "hor_visc_init: halo >= 3 required");
  HIPCHK(hipSetDevice(c->device));
  c->hv = cs;
  const Dm d = c->d;
  const size_t bytes = (size_t)HV_COUNT * d.slab * sizeof(double);
  if (!c->hv_planes) HIPCHK(hipMalloc(&c->hv_planes, bytes));
  HIPCHK(hipMemsetAsync(c->hv_planes, 0, bytes, c->stream));
  if (cs.Laplacian || cs.biharmonic) {
    const dim3 b = blk2();
    const dim3 g = grid3(d.ni + 2 * d.halo, d.nj + 2 * d.halo, 1, b);
    KLAUNCH(c, "k_hv_init1", k_hv_init1, g, b, d, c->G, cs, c->hv_planes);
    KLAUNCH(c, "k_hv_init2", k_hv_init2, g, b, d, c->G, cs, c->hv_planes);
    KLAUNCH(c, "k_hv_init3", k_hv_init3, g, b, d, c->G, cs, c->hv_planes);
    if ((cs.Leith_Kh || cs.Leith_Ah) && cs.use_beta_in_Leith) {   // G%dF_dx, G%dF_dy (+ their pass_vector)
      KLAUNCH(c, "k_hv_grad_Coriolis", k_hv_grad_Coriolis, grid3(d.ni, d.nj, 1, b), b, d, c->G, c->hv_planes);
      double *f[] = { c->hv_planes + (size_t)HV_dF_dx * d.slab, c->hv_planes + (size_t)HV_dF_dy * d.slab };
      const int stg[] = { 0, 0 }, nks[] = { 1, 1 };
      halo_wrap(c, f, stg, nks, 2);
    }
    HIPCHK(hipGetLastError());
  }
  c->hv_init = true;
  return MOM6X_OK;
}

extern "C" int mom6x_horizontal_viscosity(mom6x_ctx *c, const double *u, const double *v, const double *h, double *diffu,
                                          double *diffv) {
  REQUIRE(c && c->hv_init, MOM6X_EINVAL, "MOM_hor_visc: Module must be initialized before it is used.");
  REQUIRE(u && v && h && diffu && diffv, MOM6X_EINVAL, "horizontal_viscosity: null array");
  const mom6x_hor_visc_params &CS = c->hv;
  if (!(CS.Laplacian || CS.biharmonic)) return MOM6X_OK;
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *sh_xx, *sh_xy, *Del2u, *Del2v, *str_xx, *str_xy;
  int rc;
  if ((rc = ctx_scratch(c, SCR_t0, d.nk, &sh_xx)) || (rc = ctx_scratch(c, SCR_t1, d.nk, &sh_xy)) ||
      (rc = ctx_scratch(c, SCR_t2, d.nk, &Del2u)) || (rc = ctx_scratch(c, SCR_t3, d.nk, &Del2v)) ||
      (rc = ctx_scratch(c, SCR_q, d.nk, &str_xx)) || (rc = ctx_scratch(c, SCR_KE, d.nk, &str_xy)))
    return rc;
  const dim3 b = blk2();
  const double *P = c->hv_planes;
  // Default: the four stages in one LDS-tiled kernel (k_hv_fused: 3 reads + 2 writes per cell-layer instead of 20);
  // MOM6X_HORVISC=legacy: the four kernels.  Measured round 3 at 1440 x 1080 x 75 (bench switches): 4.75 ms on 32 x 16 tiles
  // against 6.0 ms for the four kernels.  The first version was SLOWER (7.5 ms; 6.8 on 64 x 16 tiles) and was shelved as
  // "instruction-bound": it had been compiled for four wavefronts per SIMD (128 registers) and kept 56 of its ~170 live values
  // in scratch memory.  At two wavefronts per SIMD nothing spills; a 1024-thread tile cannot have that (16 wavefronts per
  // work-group): the 64 x 16 tiles are gone.
  static const bool fused = [] { const char *e = getenv("MOM6X_HORVISC"); return !(e && !strcmp(e, "legacy")); }();
  if (fused && !(CS.Leith_Kh || CS.Leith_Ah) && d.halo >= 4) {
    // (layers per work-group: the whole column up to 80 layers -- the ~30 coefficient planes of a tile are then read once; 4.53 ms
    //  against 4.68 with 25-layer chunks at nk = 75)
    const int kc = (d.nk <= 80) ? d.nk : ((d.nk % 25 == 0) ? 25 : KCHUNK);
    // Tiles: 32 x 24 points (768 threads = 12 wavefronts, three per SIMD at <= 168 registers: the OM4-class instantiation has 150;
    // outputs on 28 x 20 = 73 % of the tile).  Round 3's 32 x 16 tiles (two per SIMD, 66 %: 3.57 ms against 2.49-2.82 at 1440 x 1080 x 75)
    // and the 64 x 16 tiles (1024 threads: four wavefronts per SIMD = 128 registers, 56 values in scratch) lost and are gone
    // (profiles/README.md has their numbers); so has the launch-order walk of the tiles.
    constexpr int TX = HV_TX, TY = HV_TY;
    const dim3 bt(TX, TY, 1);
    const int gx = (d.ni + 1 + (TX - 2 * HT_H) - 1) / (TX - 2 * HT_H), gy = (d.nj + 1 + (TY - 2 * HT_H) - 1) / (TY - 2 * HT_H), gz = (d.nk + kc - 1) / kc;
    const dim3 gt((unsigned)(((gx * gy * gz + 7) / 8) * 8), 1, 1);
    const size_t ldsb = (size_t)16 * (TY + 2) * (TX + 2) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {   // more than 64 KB of dynamic LDS has to be asked for
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hv_fused<TX, TY, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hv_fused<TX, TY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
      attr_set = true;
    }
    const bool om4 = CS.Laplacian && CS.biharmonic && !CS.Smagorinsky_Kh && CS.Smagorinsky_Ah && CS.better_bound_Kh &&
                     CS.better_bound_Ah && !CS.no_slip && !CS.bound_Coriolis && CS.bound_Ah && CS.bound_Kh && CS.backscatter_underbound &&
                     !CS.add_LES_viscosity && CS.use_land_mask;
    if (om4) {
      KLAUNCH_LDS(c, "k_hv_fused", (k_hv_fused<TX, TY, true>), gt, bt, ldsb, d, c->G, P, CS, u, v, h, diffu, diffv, c->GV.H_subroundoff, kc, gx, gy, gz, 1);
    } else {
      KLAUNCH_LDS(c, "k_hv_fused", (k_hv_fused<TX, TY, false>), gt, bt, ldsb, d, c->G, P, CS, u, v, h, diffu, diffv, c->GV.H_subroundoff, kc, gx, gy, gz, 1);
    }
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  KLAUNCH(c, "k_hv_strain", k_hv_strain, grid3(nxa(d.ni + 4, -2), d.nj + 4, nchunks(d.nk), b), b, d, c->G, P, u, v, sh_xx, sh_xy, CS.no_slip);
  if (CS.biharmonic)
    KLAUNCH(c, "k_hv_del2", k_hv_del2, grid3(nxa(d.ni + 3, -2), d.nj + 3, nchunks(d.nk), b), b, d, c->G, P, (const double *)sh_xx,
            (const double *)sh_xy, Del2u, Del2v);
  double *LD = nullptr, *LH = nullptr, *LQ = nullptr;
  if (CS.Leith_Kh || CS.Leith_Ah) {
    // the vorticity and the divergence borrow the stress arrays (dead until k_hv_stress writes them)
    double *vort = str_xx, *divx = str_xy;
    if ((rc = ctx_scratch(c, SCR_absv, d.nk, &LD)) || (rc = ctx_scratch(c, SCR_e, d.nk + 1, &LH)) || (rc = ctx_scratch(c, SCR_c1, d.nk, &LQ)))
      return rc;
    KLAUNCH(c, "k_hv_vort", k_hv_vort, grid3(nxa(d.ni + 5, -3), d.nj + 5, nchunks(d.nk), b), b, d, c->G, P, u, v, vort, divx, CS.no_slip,
            CS.modified_Leith);
    KLAUNCH(c, "k_hv_leith", k_hv_leith, grid3(nxa(d.ni + 3, -2), d.nj + 3, nchunks(d.nk), b), b, d, c->G, P, (const double *)vort,
            (const double *)divx, LD, LH, LQ, CS.modified_Leith, CS.use_beta_in_Leith);
    // k_hv_stress reads LD / LH / LQ and overwrites vort / divx: the stream orders them
  }
  KLAUNCH(c, "k_hv_stress", k_hv_stress, grid3(nxa(d.ni + 2, -1), d.nj + 2, nchunks(d.nk), b), b, d, c->G, P, CS, h, (const double *)sh_xx,
          (const double *)sh_xy, (const double *)Del2u, (const double *)Del2v, str_xx, str_xy, c->GV.H_subroundoff, (const double *)LD,
          (const double *)LH, (const double *)LQ);
  KLAUNCH(c, "k_hv_accel", k_hv_accel, grid3(nxa(d.ni + 1, -1), d.nj + 1, nchunks(d.nk), b), b, d, c->G, P, h, (const double *)str_xx,
          (const double *)str_xy, diffu, diffv, CS.use_land_mask, c->GV.H_subroundoff);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
