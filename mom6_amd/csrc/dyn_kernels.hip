// dyn_kernels.hip -- Coriolis/momentum advection, finite-volume pressure-gradient force and the
// vertical-viscosity tridiagonal solves on gfx950.
//
//   CorAdCalc + gradKE        <- MOM_CoriolisAdv.F90:125-1052
//   PressureForce_FV_Bouss    <- MOM_PressureForce_FV.F90:947-2017 (layered / no-EOS path)
//     + Set_pbce_Bouss        <- MOM_PressureForce_Montgomery.F90:649-748
//   vertvisc, vertvisc_remnant<- MOM_vert_friction.F90:557-1356
//
// Horizontal-stencil kernels are "column-walk" kernels: lane index = i (coalesced), each thread keeps
// the 2-D metric coefficients of its point in registers and walks KCHUNK layers, so the metric planes
// are read nk/KCHUNK times instead of nk times.  Vertical solves are one thread per column with
// sequential k (Thomas algorithm in the Schopf & Loughe form used by the reference).
#include "mom6x_dev.h"
#include "eos_dev.h"

namespace {

inline dim3 blk2() { return dim3(64, 4, 1); }

// ---------------------------------------------------------------------------------------------
// CorAdCalc, pass 1: potential vorticity q (and abs_vort) at vertices, kinetic energy at cells.
__global__ void __launch_bounds__(256)
k_corad_q(Dm d, const double *__restrict__ G, const double *__restrict__ u, const double *__restrict__ v,
          const double *__restrict__ h, double *__restrict__ q, double *__restrict__ absv, double *__restrict__ KE,
          int no_slip, int ke_scheme, double vol_neglect, double *__restrict__ Ihq) {
  const int i = I_BASE(-2) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni || j > d.nj) return;
  if (i < (-2)) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const double *mT = gm(G, d, MOM6X_G_mask2dT), *areaT = gm(G, d, MOM6X_G_areaT);
  // Area_h :239-241, Area_q :265-268
  const double A00 = mT[x] * areaT[x], A10 = mT[x + 1] * areaT[x + 1];
  const double A01 = mT[x + st] * areaT[x + st], A11 = mT[x + 1 + st] * areaT[x + 1 + st];
  const double Area_q = (A00 + A11) + (A10 + A01);
  const double dyCv0 = gm(G, d, MOM6X_G_dyCv)[x], dyCv1 = gm(G, d, MOM6X_G_dyCv)[x + 1];
  const double dxCu0 = gm(G, d, MOM6X_G_dxCu)[x], dxCu1 = gm(G, d, MOM6X_G_dxCu)[x + st];
  const double mBu = gm(G, d, MOM6X_G_mask2dBu)[x], IareaBu = gm(G, d, MOM6X_G_IareaBu)[x];
  const double fBu = gm(G, d, MOM6X_G_CoriolisBu)[x];
  const double vfac = no_slip ? (2.0 - mBu) : mBu;
  const bool do_KE = (i >= -1 && j >= -1);
  double aCu0 = 0, aCu1 = 0, aCv0 = 0, aCv1 = 0, IareaT = 0;
  if (do_KE) {
    aCu0 = gm(G, d, MOM6X_G_areaCu)[x]; aCu1 = gm(G, d, MOM6X_G_areaCu)[x - 1];
    aCv0 = gm(G, d, MOM6X_G_areaCv)[x]; aCv1 = gm(G, d, MOM6X_G_areaCv)[x - st];
    IareaT = gm(G, d, MOM6X_G_IareaT)[x];
  }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    const double u0 = u[c], v0 = v[c];
    const double dvdx = (v[c + 1] * dyCv1) - (v0 * dyCv0);
    const double dudy = (u[c + st] * dxCu1) - (u0 * dxCu0);
    const double h00 = h[c], h10 = h[c + 1], h01 = h[c + st], h11 = h[c + 1 + st];
    const double hAu0 = 0.5 * ((A00 * h00) + (A10 * h10));      // hArea_u(I,j)
    const double hAu1 = 0.5 * ((A01 * h01) + (A11 * h11));      // hArea_u(I,j+1)
    const double hAv0 = 0.5 * ((A00 * h00) + (A01 * h01));      // hArea_v(i,J)
    const double hAv1 = 0.5 * ((A10 * h10) + (A11 * h11));      // hArea_v(i+1,J)
    const double rel_vort = vfac * (dvdx - dudy) * IareaBu;
    const double abs_vort = fBu + rel_vort;
    const double hArea_q = (hAu0 + hAu1) + (hAv0 + hAv1);
    const double Ih_q = Area_q / (hArea_q + vol_neglect);
    q[c] = abs_vort * Ih_q;
    if (absv) absv[c] = abs_vort;
    if (Ihq) Ihq[c] = Ih_q;   // ARAKAWA_LAMB_BLEND weighs its three schemes by the spread of Ih_q around a cell (:550-573)
    if (do_KE) {
      const double um1 = u[c - 1], vm1 = v[c - st];
      double ke;
      if (ke_scheme == MOM6X_KE_ARAKAWA) {
        ke = (((aCu0 * (u0 * u0)) + (aCu1 * (um1 * um1))) + ((aCv0 * (v0 * v0)) + (aCv1 * (vm1 * vm1)))) * 0.25 * IareaT;
      } else if (ke_scheme == MOM6X_KE_SIMPLE_GUDONOV) {
        const double up = 0.5 * (um1 + fabs(um1)), up2 = up * up;
        const double um = 0.5 * (u0 - fabs(u0)), um2 = um * um;
        const double vp = 0.5 * (vm1 + fabs(vm1)), vp2 = vp * vp;
        const double vm = 0.5 * (v0 - fabs(v0)), vm2 = vm * vm;
        ke = (dmax(up2, um2) + dmax(vp2, vm2)) * 0.5;
      } else {
        const double up = 0.5 * (um1 + fabs(um1)), up2a = up * up * aCu1;
        const double um = 0.5 * (u0 - fabs(u0)), um2a = um * um * aCu0;
        const double vp = 0.5 * (vm1 + fabs(vm1)), vp2a = vp * vp * aCv1;
        const double vm = 0.5 * (v0 - fabs(v0)), vm2a = vm * vm * aCv0;
        ke = (dmax(um2a, up2a) + dmax(vm2a, vp2a)) * 0.5 * IareaT;
      }
      KE[c] = ke;
    }
  }
}

__device__ __forceinline__ double max4(double a, double b, double c, double d) { return dmax(dmax(dmax(a, b), c), d); }
__device__ __forceinline__ double min4(double a, double b, double c, double d) { return dmin(dmin(dmin(a, b), c), d); }

// CorAdCalc, pass 2: the accelerations CAu (I=-1..ni-1, j=0..nj-1) and CAv (i=0..ni-1, J=-1..nj-1).
// One layer of one point, shared by the two-kernel form (q, KE, abs_vort read from HBM: k_corad_acc) and the one-kernel form
// (from the work-group's LDS tile: k_corad_fused) through the accessors Q(di, dj), KEf(di, dj), AVf(di, dj).
struct CoradAcc {
  const double *u, *v, *uh, *vh, *h, *PFu, *PFv, *diffu, *diffv;
  double *CAu, *CAv, *u_bc, *v_bc;
  double IdxCu, IdyCv, Lv[4], Lu[4];
  int scheme, bound, en_dis;
  bool do_u, do_v;
  // ARAKAWA_LAMB_BLEND (:544-548), ROBUST_ENSTRO (:242-243; IdxCv / IdyCu of the four faces in Lv / Lu)
  double Fe_m2, rat_lin, wt_lin, eps_vel, h_tiny;
  int pv_upwind;
};
// The Arakawa & Lamb (1981) weights of one thickness cell from the potential vorticity at its four corners (NE = q(I,J),
// SW = q(I-1,J-1), NW = q(I-1,J), SE = q(I,J-1)): a(I-1,j), d(I-1,j), b(I,j), c(I,j), ep_u(i,j), ep_v(i,j)  :534-542, and their
// ARAKAWA_LAMB_BLEND form :543-588 (ih* = Ih_q at the same corners).
struct ALCell { double a, d, b, c, ep_u, ep_v; };
__device__ __forceinline__ ALCell al_cell(const CoradAcc &X, double qNE, double qSW, double qNW, double qSE, double ihNE,
                                          double ihSW, double ihNW, double ihSE) {
  const double C1_24 = 1.0 / 24.0;
  ALCell r;
  if (X.scheme == MOM6X_ARAKAWA_LAMB81) {
    r.a = (2.0 * (qNE + qSW) + (qNW + qSE)) * C1_24;
    r.d = ((qNE + qSW) + 2.0 * (qNW + qSE)) * C1_24;
    r.b = ((qNE + qSW) + 2.0 * (qNW + qSE)) * C1_24;
    r.c = (2.0 * (qNE + qSW) + (qNW + qSE)) * C1_24;
    r.ep_u = ((qNE - qSW) + (qNW - qSE)) * C1_24;
    r.ep_v = (-(qNE - qSW) + (qNW - qSE)) * C1_24;
  } else {
    const double min_Ihq = dmin(dmin(dmin(ihSW, ihSE), ihNW), ihNE), max_Ihq = dmax(dmax(dmax(ihSW, ihSE), ihNW), ihNE);
    double rat_m1 = 1.0e15, AL_wt, Sad_wt;
    if (max_Ihq < 1.0e15 * min_Ihq) rat_m1 = max_Ihq / min_Ihq - 1.0;
    if (rat_m1 <= X.Fe_m2) AL_wt = 1.0;
    else if (rat_m1 < 1.5 * X.Fe_m2) AL_wt = 3.0 * X.Fe_m2 / rat_m1 - 2.0;
    else AL_wt = 0.0;
    if (rat_m1 <= 1.5 * X.Fe_m2) Sad_wt = 0.0;
    else if (rat_m1 <= X.rat_lin) Sad_wt = 1.0 - (1.5 * X.Fe_m2) / rat_m1;
    else if (rat_m1 < 2.0 * X.rat_lin) Sad_wt = 1.0 - (X.wt_lin / X.rat_lin) * (rat_m1 - 2.0 * X.rat_lin);
    else Sad_wt = 1.0;
    r.a = Sad_wt * 0.25 * qNW + (1.0 - Sad_wt) * (((2.0 - AL_wt) * qNW + AL_wt * qSE) + 2.0 * (qNE + qSW)) * C1_24;
    r.d = Sad_wt * 0.25 * qSW + (1.0 - Sad_wt) * (((2.0 - AL_wt) * qSW + AL_wt * qNE) + 2.0 * (qNW + qSE)) * C1_24;
    r.b = Sad_wt * 0.25 * qNE + (1.0 - Sad_wt) * (((2.0 - AL_wt) * qNE + AL_wt * qSW) + 2.0 * (qNW + qSE)) * C1_24;
    r.c = Sad_wt * 0.25 * qSE + (1.0 - Sad_wt) * (((2.0 - AL_wt) * qSE + AL_wt * qNW) + 2.0 * (qNE + qSW)) * C1_24;
    r.ep_u = AL_wt * ((qNE - qSW) + (qNW - qSE)) * C1_24;
    r.ep_v = AL_wt * (-(qNE - qSW) + (qNW - qSE)) * C1_24;
  }
  return r;
}
// Heff of ROBUST_ENSTRO (:692-703, :813-824): the transport's own thickness, kept between the two cells' thicknesses
__device__ __forceinline__ double robust_heff(double tr, double Idl, double vel, double eps_vel, double hA, double hB) {
  double He = fabs(tr * Idl) / (eps_vel + fabs(vel));
  He = dmax(He, dmin(hA, hB));
  return dmin(He, dmax(hA, hB));
}
struct NoIhq { __device__ double operator()(int, int) const { return 0.0; } };
// ALL: with the three schemes only the two-kernel form runs (ROBUST_ENSTRO, ARAKAWA_LAMB81, ARAKAWA_LAMB_BLEND).  k_corad_fused
// compiles them OUT: merely present, never taken, they cost it 14 spilled registers (3.6 instead of 2.0 ms per call).
// LEAN: the default configuration known at compile time (SADOURNY75_ENERGY without BOUND_CORIOLIS and CORIOLIS_EN_DIS).
template <bool ALL, bool LEAN = false, class QF, class KF, class AF, class IF = NoIhq>
__device__ __forceinline__ void corad_acc_layer(const CoradAcc &X, size_t c, int st, const QF &Q, const KF &KEf, const AF &AVf,
                                                const IF &IH = NoIhq()) {
  const double *__restrict__ u = X.u, *__restrict__ v = X.v, *__restrict__ uh = X.uh, *__restrict__ vh = X.vh, *__restrict__ h = X.h;
  const double *__restrict__ PFu = X.PFu, *__restrict__ PFv = X.PFv, *__restrict__ diffu = X.diffu, *__restrict__ diffv = X.diffv;
  double *__restrict__ CAu = X.CAu, *__restrict__ CAv = X.CAv, *__restrict__ u_bc = X.u_bc, *__restrict__ v_bc = X.v_bc;
  const double IdxCu = X.IdxCu, IdyCv = X.IdyCv;
  const double *Lv = X.Lv, *Lu = X.Lu;
  const int scheme = LEAN ? (int)MOM6X_SADOURNY75_ENERGY : X.scheme, bound = LEAN ? 0 : X.bound, en_dis = LEAN ? 0 : X.en_dis;
  const bool do_u = X.do_u, do_v = X.do_v;
  const double C1_12 = 1.0 / 12.0;
  auto ihq = [&](int di, int dj) { return (scheme == MOM6X_AL_BLEND) ? IH(di, dj) : 0.0; };   // only the blend has (and reads) Ih_q
  // CORIOLIS_EN_DIS (:326-333, :590-635): the centred thickness transport of a face and the one the continuity solver
  // gave bracket the transport used by the energy-dissipating scheme; recomputed here for the four faces each point needs
  auto bracket = [](double Lf, double vel, double hsum, double hm_in, double &mn, double &mx) {
    const double c1 = 1.0 - 1.5 * 0.5, c2 = 1.0 - 0.5, c3 = 2.0, slope = 0.5;
    double uhc = 0.5 * ((Lf * 1.0) * vel) * hsum, uhm = hm_in;
    if (Lf == 0.0) uhc = uhm;
    if (fabs(uhc) < 0.1 * fabs(uhm)) uhm = 10.0 * uhc;
    else if (fabs(uhc) > c1 * fabs(uhm)) {
      if (fabs(uhc) < c2 * fabs(uhm)) uhc = (3.0 * uhc + (1.0 - c2 * 3.0) * uhm);
      else if (fabs(uhc) <= c3 * fabs(uhm)) uhc = uhm;
      else uhc = slope * uhc + (1.0 - c3 * slope) * uhm;
    }
    if (uhc > uhm) { mn = uhm; mx = uhc; } else { mx = uhm; mn = uhc; }
  };
    const double q00 = Q(0, 0);
    if (do_u) {
      const double q0m = Q(0, -1);
      double ca;
      if (scheme == MOM6X_SADOURNY75_ENERGY && en_dis) {   // :665-684
        double mn0, mx0, mn1, mx1, mn2, mx2, mn3, mx3;      // v faces (i,J), (i+1,J), (i,J-1), (i+1,J-1)
        bracket(Lv[0], v[c], h[c] + h[c + st], vh[c], mn0, mx0);
        bracket(Lv[1], v[c + 1], h[c + 1] + h[c + 1 + st], vh[c + 1], mn1, mx1);
        bracket(Lv[2], v[c - st], h[c - st] + h[c], vh[c - st], mn2, mx2);
        bracket(Lv[3], v[c + 1 - st], h[c + 1 - st] + h[c + 1], vh[c + 1 - st], mn3, mx3);
        const double uk = u[c];
        double temp1, temp2;
        if (q00 * uk == 0.0) temp1 = q00 * ((mx0 + mx1) + (mn0 + mn1)) * 0.5;
        else if (q00 * uk < 0.0) temp1 = q00 * (mx0 + mx1);
        else temp1 = q00 * (mn0 + mn1);
        if (q0m * uk == 0.0) temp2 = q0m * ((mx2 + mx3) + (mn2 + mn3)) * 0.5;
        else if (q0m * uk < 0.0) temp2 = q0m * (mx2 + mx3);
        else temp2 = q0m * (mn2 + mn3);
        ca = 0.25 * IdxCu * (temp1 + temp2);
      } else if (scheme == MOM6X_SADOURNY75_ENERGY) {
        ca = 0.25 * ((q00 * (vh[c + 1] + vh[c])) + (q0m * (vh[c - st] + vh[c + 1 - st]))) * IdxCu;
      } else if (scheme == MOM6X_SADOURNY75_ENSTRO) {
        ca = 0.125 * (IdxCu * (q00 + q0m)) * ((vh[c + 1] + vh[c]) + (vh[c - st] + vh[c + 1 - st]));
      } else if (!ALL || scheme == MOM6X_ARAKAWA_HSU90) {   // :523-533, :683-686
        const double a = (q00 + (Q(1, 0) + q0m)) * C1_12;
        const double dd = ((q00 + Q(1, -1)) + q0m) * C1_12;
        const double b = (q00 + (Q(-1, 0) + q0m)) * C1_12;
        const double cc = ((q00 + Q(-1, -1)) + q0m) * C1_12;
        ca = (((a * vh[c + 1]) + (cc * vh[c - st])) + ((b * vh[c]) + (dd * vh[c + 1 - st]))) * IdxCu;
      } else if (scheme == MOM6X_ROBUST_ENSTRO) {   // :687-714; Lv = IdxCv of the v faces (i,J), (i+1,J), (i,J-1), (i+1,J-1)
        const double Heff1 = robust_heff(vh[c], Lv[0], v[c], X.eps_vel, h[c], h[c + st]);
        const double Heff2 = robust_heff(vh[c - st], Lv[2], v[c - st], X.eps_vel, h[c - st], h[c]);
        const double Heff3 = robust_heff(vh[c + 1], Lv[1], v[c + 1], X.eps_vel, h[c + 1], h[c + 1 + st]);
        const double Heff4 = robust_heff(vh[c + 1 - st], Lv[3], v[c + 1 - st], X.eps_vel, h[c + 1 - st], h[c + 1]);
        const double av0 = AVf(0, 0), avm = AVf(0, -1);
        const double VHeff = ((vh[c] + vh[c + 1 - st]) + (vh[c - st] + vh[c + 1]));
        if (X.pv_upwind) {
          const double QVHeff = 0.5 * (((av0 + avm) * VHeff) - ((av0 - avm) * fabs(VHeff)));
          ca = (QVHeff / (X.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3)))) * IdxCu;
        } else
          ca = 0.5 * (av0 + avm) * VHeff / (X.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdxCu;
      } else {   // ARAKAWA_LAMB81 / ARAKAWA_LAMB_BLEND: the cells (i+1,j) and (i,j) either side of the face  :534-588, :683-686, :716-721
        const double q10 = Q(1, 0), q1m = Q(1, -1), qm0 = Q(-1, 0), qmm = Q(-1, -1);
        const ALCell E = al_cell(X, q10, q0m, q00, q1m, ihq(1, 0), ihq(0, -1), ihq(0, 0), ihq(1, -1));
        const ALCell W = al_cell(X, q00, qmm, qm0, q0m, ihq(0, 0), ihq(-1, -1), ihq(-1, 0), ihq(0, -1));
        ca = (((E.a * vh[c + 1]) + (W.c * vh[c - st])) + ((W.b * vh[c]) + (E.d * vh[c + 1 - st]))) * IdxCu;
        ca = ca + ((W.ep_u * uh[c - 1]) - (E.ep_u * uh[c + 1])) * IdxCu;
      }
      if (bound) {   // :734-747
        const double av0 = AVf(0, 0), avm = AVf(0, -1);
        const double fv1 = av0 * v[c + 1], fv2 = av0 * v[c], fv3 = avm * v[c + 1 - st], fv4 = avm * v[c - st];
        ca = dmin(ca, max4(fv1, fv2, fv3, fv4));
        ca = dmax(ca, min4(fv1, fv2, fv3, fv4));
      }
      const double cau = ca - (KEf(1, 0) - KEf(0, 0)) * IdxCu;
      CAu[c] = cau;
      if (u_bc) u_bc[c] = (cau + PFu[c]) + diffu[c];   // u_bc_accel of the RK2 step (:900-907) while CAu is at hand
    }
    if (do_v) {
      const double qm0 = Q(-1, 0);
      double ca;
      if (scheme == MOM6X_SADOURNY75_ENERGY && en_dis) {   // :776-795
        double mn0, mx0, mn1, mx1, mn2, mx2, mn3, mx3;      // u faces (I-1,j), (I-1,j+1), (I,j), (I,j+1)
        bracket(Lu[0], u[c - 1], h[c - 1] + h[c], uh[c - 1], mn0, mx0);
        bracket(Lu[1], u[c - 1 + st], h[c - 1 + st] + h[c + st], uh[c - 1 + st], mn1, mx1);
        bracket(Lu[2], u[c], h[c] + h[c + 1], uh[c], mn2, mx2);
        bracket(Lu[3], u[c + st], h[c + st] + h[c + 1 + st], uh[c + st], mn3, mx3);
        const double vk = v[c];
        double temp1, temp2;
        if (qm0 * vk == 0.0) temp1 = qm0 * ((mx0 + mx1) + (mn0 + mn1)) * 0.5;
        else if (qm0 * vk > 0.0) temp1 = qm0 * (mx0 + mx1);
        else temp1 = qm0 * (mn0 + mn1);
        if (q00 * vk == 0.0) temp2 = q00 * ((mx2 + mx3) + (mn2 + mn3)) * 0.5;
        else if (q00 * vk > 0.0) temp2 = q00 * (mx2 + mx3);
        else temp2 = q00 * (mn2 + mn3);
        ca = -0.25 * IdyCv * (temp1 + temp2);
      } else if (scheme == MOM6X_SADOURNY75_ENERGY) {
        ca = -0.25 * ((qm0 * (uh[c - 1] + uh[c - 1 + st])) + (q00 * (uh[c] + uh[c + st]))) * IdyCv;
      } else if (scheme == MOM6X_SADOURNY75_ENSTRO) {
        ca = -0.125 * (IdyCv * (qm0 + q00)) * ((uh[c - 1] + uh[c - 1 + st]) + (uh[c] + uh[c + st]));
      } else if (!ALL || scheme == MOM6X_ARAKAWA_HSU90) {
        // a(I-1,j), c(I,j+1), b(I,j), d(I-1,j+1)
        const double a_m = (qm0 + (q00 + Q(-1, -1))) * C1_12;
        const double c_p = ((Q(0, 1) + Q(-1, 0)) + q00) * C1_12;
        const double b_0 = (q00 + (qm0 + Q(0, -1))) * C1_12;
        const double d_mp = ((Q(-1, 1) + q00) + qm0) * C1_12;
        ca = -(((a_m * uh[c - 1]) + (c_p * uh[c + st])) + ((b_0 * uh[c]) + (d_mp * uh[c - 1 + st]))) * IdyCv;
      } else if (scheme == MOM6X_ROBUST_ENSTRO) {   // :808-838; Lu = IdyCu of the u faces (I-1,j), (I-1,j+1), (I,j), (I,j+1)
        const double Heff1 = robust_heff(uh[c], Lu[2], u[c], X.eps_vel, h[c], h[c + 1]);
        const double Heff2 = robust_heff(uh[c - 1], Lu[0], u[c - 1], X.eps_vel, h[c - 1], h[c]);
        const double Heff3 = robust_heff(uh[c + st], Lu[3], u[c + st], X.eps_vel, h[c + st], h[c + 1 + st]);
        const double Heff4 = robust_heff(uh[c - 1 + st], Lu[1], u[c - 1 + st], X.eps_vel, h[c - 1 + st], h[c + st]);
        const double av0 = AVf(0, 0), avm = AVf(-1, 0);
        const double UHeff = ((uh[c] + uh[c - 1 + st]) + (uh[c - 1] + uh[c + st]));
        if (X.pv_upwind) {
          const double QUHeff = 0.5 * (((av0 + avm) * UHeff) - ((av0 - avm) * fabs(UHeff)));
          ca = -(QUHeff / (X.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv);
        } else
          ca = -(0.5 * (av0 + avm) * UHeff / (X.h_tiny + ((Heff1 + Heff4) + (Heff2 + Heff3))) * IdyCv);
      } else {   // ARAKAWA_LAMB81 / ARAKAWA_LAMB_BLEND: the cells (i,j) and (i,j+1) either side of the face  :796-801, :840-845
        const double qmm = Q(-1, -1), q0m = Q(0, -1), q01 = Q(0, 1), qm1 = Q(-1, 1);
        const ALCell S = al_cell(X, q00, qmm, qm0, q0m, ihq(0, 0), ihq(-1, -1), ihq(-1, 0), ihq(0, -1));
        const ALCell N = al_cell(X, q01, qm0, qm1, q00, ihq(0, 1), ihq(-1, 0), ihq(-1, 1), ihq(0, 0));
        ca = -(((S.a * uh[c - 1]) + (N.c * uh[c + st])) + ((S.b * uh[c]) + (N.d * uh[c - 1 + st]))) * IdyCv;
        ca = ca + ((S.ep_v * vh[c - st]) - (N.ep_v * vh[c + st])) * IdyCv;
      }
      if (bound) {
        const double av0 = AVf(0, 0), avm = AVf(-1, 0);
        const double fu1 = -av0 * u[c + st], fu2 = -av0 * u[c], fu3 = -avm * u[c - 1 + st], fu4 = -avm * u[c - 1];
        ca = dmin(ca, max4(fu1, fu2, fu3, fu4));
        ca = dmax(ca, min4(fu1, fu2, fu3, fu4));
      }
      const double cav = ca - (KEf(0, 1) - KEf(0, 0)) * IdyCv;
      CAv[c] = cav;
      if (v_bc) v_bc[c] = (cav + PFv[c]) + diffv[c];
    }
}

__global__ void __launch_bounds__(256)
k_corad_acc(Dm d, const double *__restrict__ G, const double *__restrict__ u, const double *__restrict__ v,
            const double *__restrict__ uh, const double *__restrict__ vh, const double *__restrict__ q,
            const double *__restrict__ absv, const double *__restrict__ KE, double *__restrict__ CAu,
            double *__restrict__ CAv, int scheme, int bound, const double *__restrict__ h, int en_dis,
            const double *__restrict__ PFu, const double *__restrict__ PFv, const double *__restrict__ diffu,
            const double *__restrict__ diffv, double *__restrict__ u_bc, double *__restrict__ v_bc,
            double *__restrict__ uhtr, double *__restrict__ vhtr, double dt_tr, const double *__restrict__ Ihq, CoradAcc X0) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < (-1)) return;
  const int st = d.pitch;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  CoradAcc X = X0;   // the scheme's constants (Fe_m2, rat_lin, wt_lin, eps_vel, h_tiny, pv_upwind)
  X.u = u; X.v = v; X.uh = uh; X.vh = vh; X.h = h; X.PFu = PFu; X.PFv = PFv; X.diffu = diffu; X.diffv = diffv;
  X.CAu = CAu; X.CAv = CAv; X.u_bc = u_bc; X.v_bc = v_bc; X.scheme = scheme; X.bound = bound; X.en_dis = en_dis;
  X.do_u = (j >= 0); X.do_v = (i >= 0);
  X.IdxCu = gm(G, d, MOM6X_G_IdxCu)[x]; X.IdyCv = gm(G, d, MOM6X_G_IdyCv)[x];
  for (int n = 0; n < 4; n++) { X.Lv[n] = 0.; X.Lu[n] = 0.; }
  if (en_dis) {
    const double *dx_Cv = gm(G, d, MOM6X_G_dx_Cv), *dy_Cu = gm(G, d, MOM6X_G_dy_Cu);
    if (X.do_u) { X.Lv[0] = dx_Cv[x]; X.Lv[1] = dx_Cv[x + 1]; X.Lv[2] = dx_Cv[x - st]; X.Lv[3] = dx_Cv[x + 1 - st]; }
    if (X.do_v) { X.Lu[0] = dy_Cu[x - 1]; X.Lu[1] = dy_Cu[x - 1 + st]; X.Lu[2] = dy_Cu[x]; X.Lu[3] = dy_Cu[x + st]; }
  }
  if (scheme == MOM6X_ROBUST_ENSTRO) {
    const double *IdxCv = gm(G, d, MOM6X_G_IdxCv), *IdyCu = gm(G, d, MOM6X_G_IdyCu);
    if (X.do_u) { X.Lv[0] = IdxCv[x]; X.Lv[1] = IdxCv[x + 1]; X.Lv[2] = IdxCv[x - st]; X.Lv[3] = IdxCv[x + 1 - st]; }
    if (X.do_v) { X.Lu[0] = IdyCu[x - 1]; X.Lu[1] = IdyCu[x - 1 + st]; X.Lu[2] = IdyCu[x]; X.Lu[3] = IdyCu[x + st]; }
  }
  // uhtr = uhtr + uh*dt, vhtr = vhtr + vh*dt (RK2.F90:1072-1079) for the points of this kernel's box (-1..ni-1, -1..nj-1), whose
  // uh(I,j), vh(i,J) it reads anyway; k_uhtr does the ring around the box
  if (uhtr)
    for (int k = k0; k < k1; k++) {
      const size_t c = x + (size_t)k * slab;
      uhtr[c] = uhtr[c] + uh[c] * dt_tr;
      vhtr[c] = vhtr[c] + vh[c] * dt_tr;
    }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    corad_acc_layer<true>(X, c, st, [&](int di, int dj) { return q[c + di + dj * st]; }, [&](int di, int dj) { return KE[c + di + dj * st]; },
                          [&](int di, int dj) { return absv[c + di + dj * st]; }, [&](int di, int dj) { return Ihq[c + di + dj * st]; });
  }
}

// CorAdCalc in ONE kernel: the potential vorticity, the kinetic energy (and abs_vort) of a layer go from the threads that
// computed them to their neighbours through LDS instead of through HBM (2 writes + ~4.6 reads per cell-layer less).  A work-group
// of CF_X x CF_Y = 32 x 16 threads owns a tile of points; every thread evaluates k_corad_q's expressions for its own point,
// the accelerations are evaluated by the tile minus a frame of one point (q is read at -1..+1, KE at 0..+1): 30 x 14 outputs per
// tile.  Two LDS buffers alternate between layers: one barrier per layer.  Same expressions, same bits as k_corad_q + k_corad_acc.
#define CF_X 32
#ifndef CF_Y
#define CF_Y 16
#endif
#ifndef CF_MINW
#define CF_MINW 4
#endif
#define CF_LDW (CF_X + 2)
#define CF_LDN ((CF_Y + 2) * CF_LDW)
template <bool LEAN>
__global__ void __launch_bounds__(CF_X * CF_Y, CF_MINW)   // 4: two work-groups (16 wavefronts) per CU, at most 128 registers
k_corad_fused(Dm d, const double *__restrict__ G, const double *__restrict__ u, const double *__restrict__ v,
              const double *__restrict__ uh, const double *__restrict__ vh, double *__restrict__ CAu,
              double *__restrict__ CAv, int scheme, int bound, const double *__restrict__ h, int en_dis,
              const double *__restrict__ PFu, const double *__restrict__ PFv, const double *__restrict__ diffu,
              const double *__restrict__ diffv, double *__restrict__ u_bc, double *__restrict__ v_bc,
              double *__restrict__ uhtr, double *__restrict__ vhtr, double dt_tr, int no_slip, int ke_scheme, double vol_neglect,
              int kc, int gx, int gy, int gz, int xcd_order) {
  __shared__ double lds[2 * 3 * CF_LDN];
  const int tx = threadIdx.x, ty = threadIdx.y;
  // A 1-D grid of gx * gy * gz work-groups.  The hardware deals consecutive work-groups round-robin to the 8 XCDs (each with its
  // own L2).  A tile row of 32 doubles starts anywhere in a 128-byte line (the tiles advance by 30), so a tile touches the lines of
  // its x neighbours: with the plain order those neighbours sit on other XCDs and the shared lines come from HBM once per tile
  // (measured 18.2 words per cell-layer for ~12 algorithmic).  xcd_order: XCD n walks a CONTIGUOUS run of tiles (x fastest), so
  // the neighbour's lines are L2 hits.
  int b = (int)blockIdx.x;
  const int nb = gx * gy * gz;
  if (xcd_order) {
    const int per = (nb + 7) / 8;
    b = (b % 8) * per + b / 8;
  }
  if (b >= nb) return;   // (the grid is padded to a multiple of 8; the whole work-group leaves)
  const int bxi = b % gx, byi = (b / gx) % gy, bzi = b / (gx * gy);
  const int i = -2 + bxi * (CF_X - 2) + tx;
  const int j = -2 + byi * (CF_Y - 2) + ty;
  const int st = d.pitch;
  const size_t slab = (size_t)d.slab;
  const int k0 = bzi * kc, k1 = min(k0 + kc, d.nk);
  const int l = (ty + 1) * CF_LDW + (tx + 1);
  const bool live = (i <= d.ni) && (j <= d.nj);                 // the range of k_corad_q: (-2..ni, -2..nj)
  const size_t x = live ? ix2(d, i, j) : ix2(d, 0, 0);
  const bool out = live && tx >= 1 && tx <= CF_X - 2 && ty >= 1 && ty <= CF_Y - 2 && i <= d.ni - 1 && j <= d.nj - 1;   // (i, j >= -1)
  // ---- the coefficients of k_corad_q
  const double *mT = gm(G, d, MOM6X_G_mask2dT), *areaT = gm(G, d, MOM6X_G_areaT);
  const double A00 = mT[x] * areaT[x], A10 = mT[x + 1] * areaT[x + 1];
  const double A01 = mT[x + st] * areaT[x + st], A11 = mT[x + 1 + st] * areaT[x + 1 + st];
  const double Area_q = (A00 + A11) + (A10 + A01);
  const double dyCv0 = gm(G, d, MOM6X_G_dyCv)[x], dyCv1 = gm(G, d, MOM6X_G_dyCv)[x + 1];
  const double dxCu0 = gm(G, d, MOM6X_G_dxCu)[x], dxCu1 = gm(G, d, MOM6X_G_dxCu)[x + st];
  const double mBu = gm(G, d, MOM6X_G_mask2dBu)[x], IareaBu = gm(G, d, MOM6X_G_IareaBu)[x];
  const double fBu = gm(G, d, MOM6X_G_CoriolisBu)[x];
  const double vfac = no_slip ? (2.0 - mBu) : mBu;
  const bool do_KE = live && (i >= -1 && j >= -1);
  double aCu0 = 0, aCu1 = 0, aCv0 = 0, aCv1 = 0, IareaT = 0;
  if (do_KE) {
    aCu0 = gm(G, d, MOM6X_G_areaCu)[x]; aCu1 = gm(G, d, MOM6X_G_areaCu)[x - 1];
    aCv0 = gm(G, d, MOM6X_G_areaCv)[x]; aCv1 = gm(G, d, MOM6X_G_areaCv)[x - st];
    IareaT = gm(G, d, MOM6X_G_IareaT)[x];
  }
  // ---- and of k_corad_acc
  CoradAcc X;
  X.u = u; X.v = v; X.uh = uh; X.vh = vh; X.h = h; X.PFu = PFu; X.PFv = PFv; X.diffu = diffu; X.diffv = diffv;
  X.CAu = CAu; X.CAv = CAv; X.u_bc = u_bc; X.v_bc = v_bc; X.scheme = scheme; X.bound = bound; X.en_dis = en_dis;
  X.do_u = out && (j >= 0); X.do_v = out && (i >= 0);
  X.IdxCu = gm(G, d, MOM6X_G_IdxCu)[x]; X.IdyCv = gm(G, d, MOM6X_G_IdyCv)[x];
  for (int n = 0; n < 4; n++) { X.Lv[n] = 0.; X.Lu[n] = 0.; }
  if (!LEAN && en_dis && out) {
    const double *dx_Cv = gm(G, d, MOM6X_G_dx_Cv), *dy_Cu = gm(G, d, MOM6X_G_dy_Cu);
    if (X.do_u) { X.Lv[0] = dx_Cv[x]; X.Lv[1] = dx_Cv[x + 1]; X.Lv[2] = dx_Cv[x - st]; X.Lv[3] = dx_Cv[x + 1 - st]; }
    if (X.do_v) { X.Lu[0] = dy_Cu[x - 1]; X.Lu[1] = dy_Cu[x - 1 + st]; X.Lu[2] = dy_Cu[x]; X.Lu[3] = dy_Cu[x + st]; }
  }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    double *sq = lds + ((k - k0) & 1) * 3 * CF_LDN, *sk = sq + CF_LDN, *sa = sq + 2 * CF_LDN;
    double qv = 0.0, kev = 0.0, av = 0.0;
    if (live) {
      const double u0 = u[c], v0 = v[c];
      const double dvdx = (v[c + 1] * dyCv1) - (v0 * dyCv0);
      const double dudy = (u[c + st] * dxCu1) - (u0 * dxCu0);
      const double h00 = h[c], h10 = h[c + 1], h01 = h[c + st], h11 = h[c + 1 + st];
      const double hAu0 = 0.5 * ((A00 * h00) + (A10 * h10));      // hArea_u(I,j)
      const double hAu1 = 0.5 * ((A01 * h01) + (A11 * h11));      // hArea_u(I,j+1)
      const double hAv0 = 0.5 * ((A00 * h00) + (A01 * h01));      // hArea_v(i,J)
      const double hAv1 = 0.5 * ((A10 * h10) + (A11 * h11));      // hArea_v(i+1,J)
      const double rel_vort = vfac * (dvdx - dudy) * IareaBu;
      const double abs_vort = fBu + rel_vort;
      const double hArea_q = (hAu0 + hAu1) + (hAv0 + hAv1);
      const double Ih_q = Area_q / (hArea_q + vol_neglect);
      qv = abs_vort * Ih_q;
      av = abs_vort;
      if (do_KE) {
        const double um1 = u[c - 1], vm1 = v[c - st];
        if (ke_scheme == MOM6X_KE_ARAKAWA) {
          kev = (((aCu0 * (u0 * u0)) + (aCu1 * (um1 * um1))) + ((aCv0 * (v0 * v0)) + (aCv1 * (vm1 * vm1)))) * 0.25 * IareaT;
        } else if (ke_scheme == MOM6X_KE_SIMPLE_GUDONOV) {
          const double up = 0.5 * (um1 + fabs(um1)), up2 = up * up;
          const double um = 0.5 * (u0 - fabs(u0)), um2 = um * um;
          const double vp = 0.5 * (vm1 + fabs(vm1)), vp2 = vp * vp;
          const double vm = 0.5 * (v0 - fabs(v0)), vm2 = vm * vm;
          kev = (dmax(up2, um2) + dmax(vp2, vm2)) * 0.5;
        } else {
          const double up = 0.5 * (um1 + fabs(um1)), up2a = up * up * aCu1;
          const double um = 0.5 * (u0 - fabs(u0)), um2a = um * um * aCu0;
          const double vp = 0.5 * (vm1 + fabs(vm1)), vp2a = vp * vp * aCv1;
          const double vm = 0.5 * (v0 - fabs(v0)), vm2a = vm * vm * aCv0;
          kev = (dmax(um2a, up2a) + dmax(vm2a, vp2a)) * 0.5 * IareaT;
        }
      }
    }
    sq[l] = qv; sk[l] = kev; if (!LEAN && bound) sa[l] = av;
    __syncthreads();
    if (out) {
      if (uhtr) {   // :1072-1079 for the box (-1..ni-1, -1..nj-1): see k_corad_acc
        uhtr[c] = uhtr[c] + uh[c] * dt_tr;
        vhtr[c] = vhtr[c] + vh[c] * dt_tr;
      }
      corad_acc_layer<false, LEAN>(X, c, st, [&](int di, int dj) { return sq[l + di + dj * CF_LDW]; }, [&](int di, int dj) { return sk[l + di + dj * CF_LDW]; },
                             [&](int di, int dj) { return sa[l + di + dj * CF_LDW]; });
    }
    // (the layer after next writes this buffer again: the barrier of the next layer lies in between)
  }
}

// k_corad_fused<LEAN> with its INPUTS through LDS too (the default configuration only: SADOURNY75_ENERGY, no bound, no EN_DIS).
// k_corad_fused's threads each ask the vector memory unit for 22 values per layer, 9 of them their own point's (u, v, h, uh, vh and
// the four arrays of the folded u_bc_accel) and 13 a neighbour's, which a neighbouring thread asks for as well: at 8 wavefronts of
// 22 loads per layer and tile the L1's 64 bytes per clock are busy for 1.3 ms of the kernel's 2.8, and every load is used at once.
// Here a thread loads its own point's five values one layer AHEAD into registers (the next layer's requests are in flight during
// the whole of this one), hands them to the tile through LDS, and the neighbours' values are LDS reads; the tile's inputs one point
// beyond its last column / row (q needs them) are loaded the same way by designated threads.  Two barriers per layer: inputs ->
// q, KE -> accelerations; the input planes alternate between layers (the accelerations still read uh, vh when a fast wavefront
// writes the next layer's), q and KE need one buffer.  Same expressions, same bits.
__global__ void __launch_bounds__(CF_X * CF_Y, 4)
k_corad_lds(Dm d, const double *__restrict__ G, const double *__restrict__ u, const double *__restrict__ v,
            const double *__restrict__ uh, const double *__restrict__ vh, double *__restrict__ CAu, double *__restrict__ CAv,
            const double *__restrict__ h, const double *__restrict__ PFu, const double *__restrict__ PFv,
            const double *__restrict__ diffu, const double *__restrict__ diffv, double *__restrict__ u_bc, double *__restrict__ v_bc,
            double *__restrict__ uhtr, double *__restrict__ vhtr, double dt_tr, int no_slip, int ke_scheme, double vol_neglect,
            int kc, int gx, int gy, int gz, int xcd_order) {
  __shared__ double lds[12 * CF_LDN];
  const int tx = threadIdx.x, ty = threadIdx.y;
  int b = (int)blockIdx.x;
  const int nb = gx * gy * gz;
  if (xcd_order) {
    const int per = (nb + 7) / 8;
    b = (b % 8) * per + b / 8;
  }
  if (b >= nb) return;
  const int bxi = b % gx, byi = (b / gx) % gy, bzi = b / (gx * gy);
  const int i = -2 + bxi * (CF_X - 2) + tx;
  const int j = -2 + byi * (CF_Y - 2) + ty;
  const int st = d.pitch;
  const size_t slab = (size_t)d.slab;
  const int k0 = bzi * kc, k1 = min(k0 + kc, d.nk);
  const int l = (ty + 1) * CF_LDW + (tx + 1);
  const bool live = (i <= d.ni) && (j <= d.nj);                 // the range of k_corad_q: (-2..ni, -2..nj)
  const bool inb = (i <= d.ni + 1) && (j <= d.nj + 1);          // the points a live thread reads (halo >= 3: inside the arrays)
  const size_t x = inb ? ix2(d, i, j) : ix2(d, 0, 0);
  const bool out = live && tx >= 1 && tx <= CF_X - 2 && ty >= 1 && ty <= CF_Y - 2 && i <= d.ni - 1 && j <= d.nj - 1;   // (i, j >= -1)
  // the tile's inputs one point beyond its last column and row: (kind 1) v, h east of column CF_X-1, (2) u, h north of row
  // CF_Y-1, (3) h at the corner
  int eKind = 0, eL = 0, ei = 0, ej = 0;
  if (tx == CF_X - 1) { eKind = 1; ei = i + 1; ej = j; eL = (ty + 1) * CF_LDW + CF_X + 1; }
  else if (ty == CF_Y - 1) { eKind = 2; ei = i; ej = j + 1; eL = (CF_Y + 1) * CF_LDW + tx + 1; }
  else if (tx == 0 && ty == 0) { eKind = 2; ei = i + CF_X - 1; ej = j + CF_Y; eL = (CF_Y + 1) * CF_LDW + CF_X; }
  else if (tx == 1 && ty == 0) { eKind = 3; ei = i - 1 + CF_X; ej = j + CF_Y; eL = (CF_Y + 1) * CF_LDW + CF_X + 1; }
  if (ei > d.ni + 1 || ej > d.nj + 1) eKind = 0;
  const size_t xe = eKind ? ix2(d, ei, ej) : x;
  const double *eA = (eKind == 1) ? v : ((eKind == 2) ? u : h);
  // ---- the coefficients of k_corad_q
  const double *mT = gm(G, d, MOM6X_G_mask2dT), *areaT = gm(G, d, MOM6X_G_areaT);
  const size_t xq = live ? x : ix2(d, 0, 0);
  const double A00 = mT[xq] * areaT[xq], A10 = mT[xq + 1] * areaT[xq + 1];
  const double A01 = mT[xq + st] * areaT[xq + st], A11 = mT[xq + 1 + st] * areaT[xq + 1 + st];
  const double Area_q = (A00 + A11) + (A10 + A01);
  const double dyCv0 = gm(G, d, MOM6X_G_dyCv)[xq], dyCv1 = gm(G, d, MOM6X_G_dyCv)[xq + 1];
  const double dxCu0 = gm(G, d, MOM6X_G_dxCu)[xq], dxCu1 = gm(G, d, MOM6X_G_dxCu)[xq + st];
  const double mBu = gm(G, d, MOM6X_G_mask2dBu)[xq], IareaBu = gm(G, d, MOM6X_G_IareaBu)[xq];
  const double fBu = gm(G, d, MOM6X_G_CoriolisBu)[xq];
  const double vfac = no_slip ? (2.0 - mBu) : mBu;
  const bool do_KE = live && (i >= -1 && j >= -1) && tx >= 1 && ty >= 1;   // (the accelerations read KE of threads 1.. only)
  double aCu0 = 0, aCu1 = 0, aCv0 = 0, aCv1 = 0, IareaT = 0;
  if (do_KE) {
    aCu0 = gm(G, d, MOM6X_G_areaCu)[xq]; aCu1 = gm(G, d, MOM6X_G_areaCu)[xq - 1];
    aCv0 = gm(G, d, MOM6X_G_areaCv)[xq]; aCv1 = gm(G, d, MOM6X_G_areaCv)[xq - st];
    IareaT = gm(G, d, MOM6X_G_IareaT)[xq];
  }
  const bool do_u = out && (j >= 0), do_v = out && (i >= 0);
  const double IdxCu = gm(G, d, MOM6X_G_IdxCu)[xq], IdyCv = gm(G, d, MOM6X_G_IdyCv)[xq];
  double *sq = lds + 10 * CF_LDN, *sk = sq + CF_LDN;
  // the first layer's values
  double r_u = 0., r_v = 0., r_h = 0., r_uh = 0., r_vh = 0., r_eA = 0., r_eB = 0.;
  {
    const size_t c = x + (size_t)k0 * slab, ce = xe + (size_t)k0 * slab;
    if (inb) { r_u = u[c]; r_v = v[c]; r_h = h[c]; r_uh = uh[c]; r_vh = vh[c]; }
    if (eKind) { r_eA = eA[ce]; if (eKind != 3) r_eB = h[ce]; }
  }
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    double *su = lds + ((k - k0) & 1) * 5 * CF_LDN, *sv = su + CF_LDN, *sh = su + 2 * CF_LDN, *suh = su + 3 * CF_LDN, *svh = su + 4 * CF_LDN;
    su[l] = r_u; sv[l] = r_v; sh[l] = r_h; suh[l] = r_uh; svh[l] = r_vh;
    if (eKind == 1) { sv[eL] = r_eA; sh[eL] = r_eB; }
    else if (eKind == 2) { su[eL] = r_eA; sh[eL] = r_eB; }
    else if (eKind == 3) sh[eL] = r_eA;
    if (k + 1 < k1) {          // the next layer's requests leave before this layer's work
      const size_t cn = c + slab, ce = xe + (size_t)(k + 1) * slab;
      if (inb) { r_u = u[cn]; r_v = v[cn]; r_h = h[cn]; r_uh = uh[cn]; r_vh = vh[cn]; }
      if (eKind) { r_eA = eA[ce]; if (eKind != 3) r_eB = h[ce]; }
    }
    __syncthreads();
    double qv = 0.0, kev = 0.0;
    if (live) {
      const double u0 = su[l], v0 = sv[l];
      const double dvdx = (sv[l + 1] * dyCv1) - (v0 * dyCv0);
      const double dudy = (su[l + CF_LDW] * dxCu1) - (u0 * dxCu0);
      const double h00 = sh[l], h10 = sh[l + 1], h01 = sh[l + CF_LDW], h11 = sh[l + 1 + CF_LDW];
      const double hAu0 = 0.5 * ((A00 * h00) + (A10 * h10));      // hArea_u(I,j)
      const double hAu1 = 0.5 * ((A01 * h01) + (A11 * h11));      // hArea_u(I,j+1)
      const double hAv0 = 0.5 * ((A00 * h00) + (A01 * h01));      // hArea_v(i,J)
      const double hAv1 = 0.5 * ((A10 * h10) + (A11 * h11));      // hArea_v(i+1,J)
      const double rel_vort = vfac * (dvdx - dudy) * IareaBu;
      const double abs_vort = fBu + rel_vort;
      const double hArea_q = (hAu0 + hAu1) + (hAv0 + hAv1);
      const double Ih_q = Area_q / (hArea_q + vol_neglect);
      qv = abs_vort * Ih_q;
      if (do_KE) {
        const double um1 = su[l - 1], vm1 = sv[l - CF_LDW];
        if (ke_scheme == MOM6X_KE_ARAKAWA) {
          kev = (((aCu0 * (u0 * u0)) + (aCu1 * (um1 * um1))) + ((aCv0 * (v0 * v0)) + (aCv1 * (vm1 * vm1)))) * 0.25 * IareaT;
        } else if (ke_scheme == MOM6X_KE_SIMPLE_GUDONOV) {
          const double up = 0.5 * (um1 + fabs(um1)), up2 = up * up;
          const double um = 0.5 * (u0 - fabs(u0)), um2 = um * um;
          const double vp = 0.5 * (vm1 + fabs(vm1)), vp2 = vp * vp;
          const double vm = 0.5 * (v0 - fabs(v0)), vm2 = vm * vm;
          kev = (dmax(up2, um2) + dmax(vp2, vm2)) * 0.5;
        } else {
          const double up = 0.5 * (um1 + fabs(um1)), up2a = up * up * aCu1;
          const double um = 0.5 * (u0 - fabs(u0)), um2a = um * um * aCu0;
          const double vp = 0.5 * (vm1 + fabs(vm1)), vp2a = vp * vp * aCv1;
          const double vm = 0.5 * (v0 - fabs(v0)), vm2a = vm * vm * aCv0;
          kev = (dmax(um2a, up2a) + dmax(vm2a, vp2a)) * 0.5 * IareaT;
        }
      }
    }
    sq[l] = qv; sk[l] = kev;
    __syncthreads();
    if (out) {
      if (uhtr) {   // :1072-1079 for the box (-1..ni-1, -1..nj-1): see k_corad_acc
        uhtr[c] = uhtr[c] + suh[l] * dt_tr;
        vhtr[c] = vhtr[c] + svh[l] * dt_tr;
      }
      const double q00 = sq[l];
      if (do_u) {   // :646-650, :723-731 (SADOURNY75_ENERGY)
        const double q0m = sq[l - CF_LDW];
        const double ca = 0.25 * ((q00 * (svh[l + 1] + svh[l])) + (q0m * (svh[l - CF_LDW] + svh[l + 1 - CF_LDW]))) * IdxCu;
        const double cau = ca - (sk[l + 1] - sk[l]) * IdxCu;
        CAu[c] = cau;
        if (u_bc) u_bc[c] = (cau + PFu[c]) + diffu[c];
      }
      if (do_v) {   // :757-761, :847-855
        const double qm0 = sq[l - 1];
        const double ca = -0.25 * ((qm0 * (suh[l - 1] + suh[l - 1 + CF_LDW])) + (q00 * (suh[l] + suh[l + CF_LDW]))) * IdyCv;
        const double cav = ca - (sk[l + CF_LDW] - sk[l]) * IdyCv;
        CAv[c] = cav;
        if (v_bc) v_bc[c] = (cav + PFv[c]) + diffv[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// PressureForce_FV_Bouss, pass 1: interface heights bottom-up (:1200-1202) on (-1..ni, -1..nj).
__global__ void __launch_bounds__(256)
k_pgf_e(Dm d, const double *__restrict__ G, const double *__restrict__ h, double *__restrict__ e, double H_to_Z) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni || j > d.nj) return;
  if (i < (-1)) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  double ek = -gm(G, d, MOM6X_G_bathyT)[x];
  e[x + (size_t)d.nk * slab] = ek;
  for (int k = d.nk - 1; k >= 0; k--) {
    ek = ek + h[x + (size_t)k * slab] * H_to_Z;
    e[x + (size_t)k * slab] = ek;
  }
}

// pass 2: top-down pressure anomalies and the accelerations (:1323-1345, :1539-1552, :1794-1813),
// Set_pbce_Bouss (no-EOS :735-746) and eta (:1886).
__global__ void __launch_bounds__(256)
k_pgf_main(Dm d, const double *__restrict__ G, const double *__restrict__ h, const double *__restrict__ e,
           const double *__restrict__ Rlay, const double *__restrict__ g_prime, double *__restrict__ PFu,
           double *__restrict__ PFv, double *__restrict__ pbce, double *__restrict__ eta, double g_Earth,
           double H_to_Z, double Z_to_H, double rho_ref, double GxRho_ref, double Z_ref, double I_Rho0,
           double h_neglect, double dz_neglect, BcFold B) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni || j > d.nj) return;
  if (i < (-1)) return;
  const int st = d.pitch, nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const bool do_u = (i <= d.ni - 1) && (j >= 0) && (j <= d.nj - 1);
  const bool do_v = (j <= d.nj - 1) && (i >= 0) && (i <= d.ni - 1);
  const double e_top = e[x], e_bot = e[x + (size_t)nz * slab];
  if (eta) eta[x] = e_top * Z_to_H;
  const double Ihtot = 1.0 / ((e_top - e_bot) + dz_neglect);
  double pa0 = GxRho_ref * (e_top - Z_ref), pa1 = 0.0, pa2 = 0.0, intx_pa = 0.0, inty_pa = 0.0;
  double cu = 0.0, cv = 0.0;
  if (do_u) { pa1 = GxRho_ref * (e[x + 1] - Z_ref); intx_pa = 0.5 * (pa0 + pa1); cu = (2.0 * I_Rho0 * gm(G, d, MOM6X_G_IdxCu)[x]); }
  if (do_v) { pa2 = GxRho_ref * (e[x + st] - Z_ref); inty_pa = 0.5 * (pa0 + pa2); cv = (2.0 * I_Rho0 * gm(G, d, MOM6X_G_IdyCv)[x]); }
  double pb = 0.0;
  // bt_mass_source's eta_h (MOM_barotropic.F90:5268-5272: h summed from the top, less the depth) of the same h, while it passes
  const bool do_eh = (B.eta_h != nullptr) && i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1;
  double eta_h = 0.0;
  for (int k = 0; k < nz; k++) {
    const size_t c = x + (size_t)k * slab, cb = c + slab;
    const double R = Rlay[k] - rho_ref;
    const double h0 = h[c];
    if (do_eh) eta_h = (k == 0) ? (h0 - gm(G, d, MOM6X_G_bathyT)[x] * Z_to_H) : (eta_h + h0);
    const double dz0 = g_Earth * H_to_Z * h0;
    const double dpa0 = R * dz0, iz0 = 0.5 * R * dz0 * h0;
    const double eb0 = e[cb];
    if (do_u) {
      const double h1 = h[c + 1];
      const double dz1 = g_Earth * H_to_Z * h1;
      const double iz1 = 0.5 * R * dz1 * h1;
      const double intx_dpa = 0.5 * R * (dz0 + dz1);
      const double pf = (((pa0 * h0 + iz0) - (pa1 * h1 + iz1)) + ((h1 - h0) * intx_pa - (e[cb + 1] - eb0) * intx_dpa * Z_to_H)) *
                        (cu / ((h0 + h1) + h_neglect));
      PFu[c] = pf;
      if (B.u_bc) B.u_bc[c] = (B.CAu[c] + pf) + B.diffu[c];   // u_bc_accel of the predictor (RK2.F90:565-572) while PFu is at hand
      pa1 = pa1 + R * dz1;
      intx_pa = intx_pa + intx_dpa;
    }
    if (do_v) {
      const double h2 = h[c + st];
      const double dz2 = g_Earth * H_to_Z * h2;
      const double iz2 = 0.5 * R * dz2 * h2;
      const double inty_dpa = 0.5 * R * (dz0 + dz2);
      const double pf = (((pa0 * h0 + iz0) - (pa2 * h2 + iz2)) + ((h2 - h0) * inty_pa - (e[cb + st] - eb0) * inty_dpa * Z_to_H)) *
                        (cv / ((h0 + h2) + h_neglect));
      PFv[c] = pf;
      if (B.v_bc) B.v_bc[c] = (B.CAv[c] + pf) + B.diffv[c];
      pa2 = pa2 + R * dz2;
      inty_pa = inty_pa + inty_dpa;
    }
    pa0 = pa0 + dpa0;
    if (pbce) {
      if (k == 0) pb = g_prime[0] * H_to_Z;
      else pb = pb + (g_prime[k] * H_to_Z) * ((e[c] - e_bot) * Ihtot);
      pbce[c] = pb;
    }
  }
  if (do_eh) B.eta_h[x] = eta_h;
}

// DIRECT_STRESS (:707-720 / :958-971): the wind stress as a body force over the topmost HMIX_STRESS instead of a stress
// boundary condition.  The increment of layer k is added where the sweep picks the layer's velocity up.
struct DirectStress { double Hmix, I_Hmix, h_neglect; const double *h; };
struct DSWalk {
  bool on; double zDS, stress;
  __device__ __forceinline__ void start(const DirectStress &S, double dt_Rho0, double tau) { on = (S.Hmix > 0.0); zDS = 0.0; stress = dt_Rho0 * tau; }
  __device__ __forceinline__ double add(const DirectStress &S, double uk, size_t x3, int st) {
    if (!on) return uk;
    const double h_a = 0.5 * (S.h[x3] + S.h[x3 + st]) + S.h_neglect;
    double hfr = 1.0; if ((zDS + h_a) > S.Hmix) hfr = (S.Hmix - zDS) / h_a;
    uk = uk + S.I_Hmix * hfr * stress;
    zDS = zDS + h_a; if (zDS >= S.Hmix) on = false;
    return uk;
  }
};
static DirectStress direct_stress_of(const mom6x_ctx *c) {
  DirectStress S; S.Hmix = c->ds_Hmix; S.I_Hmix = (c->ds_Hmix > 0.0) ? 1.0 / c->ds_Hmix : 0.0; S.h_neglect = c->GV.H_subroundoff; S.h = c->ds_h;
  return S;
}

// ---------------------------------------------------------------------------------------------
// vertvisc :557-1228 (one direction); c1 is a 3-D scratch array.
template <int DIR>
__global__ void __launch_bounds__(256)
k_vertvisc(Dm d, const double *__restrict__ G, double *__restrict__ u, const double *__restrict__ a_u,
           const double *__restrict__ h_u, const double *__restrict__ Ray_u, const double *__restrict__ tau,
           double *__restrict__ c1, double dt, double dt_Rho0, double H_to_RZ, double *__restrict__ tau_bot, DirectStress S) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int nz = d.nk, st = DIR ? d.pitch : 1;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  if (mC > 0.) {
    const double surface_stress = (S.Hmix > 0.0) ? 0.0 : dt_Rho0 * (mC * tau[x]);
    DSWalk W; W.start(S, dt_Rho0, tau[x]);
    double Ray = Ray_u ? Ray_u[x] : 0.;
    double a_k = a_u[x], a_kp = a_u[x + slab];
    double hu = h_u[x];
    double b_denom_1 = hu + dt * (Ray + a_k);
    double b1 = 1.0 / (b_denom_1 + dt * a_kp);
    double d1 = b_denom_1 * b1;
    double uprev = b1 * (hu * W.add(S, u[x], x, st) + surface_stress);
    u[x] = uprev;
    for (int k = 1; k < nz; k++) {
      const size_t x3 = x + (size_t)k * slab;
      if (Ray_u) Ray = Ray_u[x3];
      a_k = a_kp; a_kp = a_u[x3 + slab];
      hu = h_u[x3];
      c1[x3] = dt * a_k * b1;
      b_denom_1 = hu + dt * (Ray + a_k * d1);
      b1 = 1.0 / (b_denom_1 + dt * a_kp);
      d1 = b_denom_1 * b1;
      uprev = (hu * W.add(S, u[x3], x3, st) + dt * a_k * uprev) * b1;
      u[x3] = uprev;
    }
    for (int k = nz - 2; k >= 0; k--) {
      const size_t x3 = x + (size_t)k * slab;
      uprev = u[x3] + c1[x3 + slab] * uprev;
      u[x3] = uprev;
    }
  }
  if (tau_bot) {
    double tb = H_to_RZ * (u[x + (size_t)(nz - 1) * slab] * a_u[x + (size_t)nz * slab]);
    if (Ray_u) for (int k = 0; k < nz; k++) tb = tb + H_to_RZ * (Ray_u[x + (size_t)k * slab] * u[x + (size_t)k * slab]);
    tau_bot[x] = tb;
  }
}

// vertvisc_remnant :1229-1356 (one direction)
template <int DIR>
__global__ void __launch_bounds__(256)
k_vertvisc_remnant(Dm d, const double *__restrict__ G, double *__restrict__ vr, const double *__restrict__ a_u,
                   const double *__restrict__ h_u, const double *__restrict__ Ray_u, double *__restrict__ c1, double dt) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  if (!(mC > 0.)) return;
  double Ray = Ray_u ? Ray_u[x] : 0.;
  double a_k = a_u[x], a_kp = a_u[x + slab];
  double hu = h_u[x];
  double b_denom_1 = hu + dt * (Ray + a_k);
  double b1 = 1.0 / (b_denom_1 + dt * a_kp);
  double d1 = b_denom_1 * b1;
  double prev = b1 * hu;
  vr[x] = prev;
  for (int k = 1; k < nz; k++) {
    const size_t x3 = x + (size_t)k * slab;
    if (Ray_u) Ray = Ray_u[x3];
    a_k = a_kp; a_kp = a_u[x3 + slab];
    hu = h_u[x3];
    c1[x3] = dt * a_k * b1;
    b_denom_1 = hu + dt * (Ray + a_k * d1);
    b1 = 1.0 / (b_denom_1 + dt * a_kp);
    d1 = b_denom_1 * b1;
    prev = (hu + dt * a_k * prev) * b1;
    vr[x3] = prev;
  }
  for (int k = nz - 2; k >= 0; k--) {
    const size_t x3 = x + (size_t)k * slab;
    prev = vr[x3] + c1[x3 + slab] * prev;
    vr[x3] = prev;
  }
}

// vertvisc_remnant with the column on chip (see k_vertvisc_cols): c1 and the un-substituted remnant stay in
// registers, 2 words read and 1 written per face-layer instead of 6.  No Ray_u (that goes through
// k_vertvisc_remnant); same operations in the same order.
// (NKT: mom6x_dev.h NK_OF / NK_EXACT -- the layer count itself, or a bound on it)
template <int DIR, int NKT>
__global__ void __launch_bounds__(64)
k_vertvisc_remnant_cols(Dm d, const double *__restrict__ G, double *__restrict__ vr, const double *__restrict__ a_u,
                        const double *__restrict__ h_u, double dt) {
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  if (!(mC > 0.)) return;
  constexpr int VV_G = 8, NG = (NK + VV_G - 1) / VV_G;
  double rr[NK], cu[NK];
  double q_a[2][VV_G], q_h[2][VV_G];
  auto fetch = [&](int g, int b) {
#pragma unroll
    for (int m = 0; m < VV_G; m++) {
      const int k = g * VV_G + m;
      if (k < NK && k < nk) { const size_t x3 = x + (size_t)k * slab; q_a[b][m] = a_u[x3 + slab]; q_h[b][m] = h_u[x3]; }
    }
  };
  double a_kp = a_u[x];
  double b1 = 0., d1 = 0., prev = 0.;
  fetch(0, 0);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) fetch(g + 1, (g + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < VV_G; m++) {
      const int k = g * VV_G + m;
      if (k < NK && k < nk) {
        const double a_k = a_kp; a_kp = q_a[g & 1][m];
        const double hu = q_h[g & 1][m];
        if (k == 0) {
          const double b_denom_1 = hu + dt * (0. + a_k);
          b1 = 1.0 / (b_denom_1 + dt * a_kp);
          d1 = b_denom_1 * b1;
          prev = b1 * hu;
        } else {
          cu[k] = dt * a_k * b1;
          const double b_denom_1 = hu + dt * (0. + a_k * d1);
          b1 = 1.0 / (b_denom_1 + dt * a_kp);
          d1 = b_denom_1 * b1;
          prev = (hu + dt * a_k * prev) * b1;
        }
        rr[k] = prev;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  vr[x + (size_t)(nk - 1) * slab] = prev;
#pragma unroll
  for (int k = NK - 2; k >= 0; k--) {
    if (k >= nk - 1) continue;
    prev = rr[k] + cu[k + 1] * prev;
    vr[x + (size_t)k * slab] = prev;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int mom6x_CoriolisAdv_init(mom6x_ctx *c, const mom6x_coriolis_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_CoriolisAdv_init: null argument");
  REQUIRE(p->Coriolis_Scheme >= MOM6X_SADOURNY75_ENERGY && p->Coriolis_Scheme <= MOM6X_AL_BLEND, MOM6X_EINVAL,
          "CoriolisAdv_init: Unrecognized setting of CORIOLIS_SCHEME");
  REQUIRE(p->KE_Scheme >= MOM6X_KE_ARAKAWA && p->KE_Scheme <= MOM6X_KE_GUDONOV, MOM6X_EINVAL, "CoriolisAdv: bad KE_SCHEME");
  REQUIRE(p->PV_Adv_Scheme == 0 || p->PV_Adv_Scheme == MOM6X_PV_ADV_CENTERED || p->PV_Adv_Scheme == MOM6X_PV_ADV_UPWIND1, MOM6X_EINVAL,
          "CoriolisAdv_init: PV_ADV_SCHEME is invalid");
  c->cor = *p;
  if (c->cor.Coriolis_Scheme == MOM6X_ROBUST_ENSTRO) { c->cor.Coriolis_En_Dis = 0; c->cor.bound_Coriolis = 0; }   // :1118, :1158
  // CoriolisAdv_init :1158: with CORIOLIS_EN_DIS and SADOURNY75_ENERGY the bound is always effectively off
  if (c->cor.Coriolis_En_Dis && c->cor.Coriolis_Scheme == MOM6X_SADOURNY75_ENERGY) c->cor.bound_Coriolis = 0;
  c->cor_init = true;
  return MOM6X_OK;
}

static inline dim3 gridk(int nx, int ny, int nk, dim3 b) {
  return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y, nchunks(nk));
}

extern "C" int mom6x_CorAdCalc(mom6x_ctx *c, const double *u, const double *v, const double *h, const double *uh,
                               const double *vh, double *CAu, double *CAv) {
  return CorAdCalc_bc(c, u, v, h, uh, vh, CAu, CAv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0);
}

// CorAdCalc, and -- for the RK2 step -- u_bc_accel = (CAu + PFu) + diffu (:900-907) formed where CAu is made
int CorAdCalc_bc(mom6x_ctx *c, const double *u, const double *v, const double *h, const double *uh, const double *vh, double *CAu,
                 double *CAv, const double *PFu, const double *PFv, const double *diffu, const double *diffv, double *u_bc,
                 double *v_bc, double *uhtr, double *vhtr, double dt_tr) {
  REQUIRE(c && c->cor_init, MOM6X_EINVAL, "MOM_CoriolisAdv: Module must be initialized before it is used.");
  REQUIRE(u && v && h && uh && vh && CAu && CAv, MOM6X_EINVAL, "CorAdCalc: null array");
  REQUIRE(c->dims.halo >= 3, MOM6X_EINVAL, "CorAdCalc: halo >= 3 required");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *q, *KE, *absv = nullptr;
  int rc;
  if ((rc = ctx_scratch(c, SCR_q, d.nk, &q))) return rc;
  if ((rc = ctx_scratch(c, SCR_KE, d.nk, &KE))) return rc;
  const int scheme = c->cor.Coriolis_Scheme;
  // the schemes of the default k_corad_fused; ROBUST_ENSTRO, ARAKAWA_LAMB81 and ARAKAWA_LAMB_BLEND take the two-kernel form
  const bool fusable = (scheme == MOM6X_SADOURNY75_ENERGY || scheme == MOM6X_SADOURNY75_ENSTRO || scheme == MOM6X_ARAKAWA_HSU90);
  if ((c->cor.bound_Coriolis || scheme == MOM6X_ROBUST_ENSTRO) && (rc = ctx_scratch(c, SCR_absv, d.nk, &absv))) return rc;
  double *Ihq = nullptr;
  if (scheme == MOM6X_AL_BLEND && (rc = ctx_scratch(c, SCR_t0, d.nk, &Ihq))) return rc;
  CoradAcc X0 = {};
  X0.Fe_m2 = c->cor.F_eff_max_blend - 2.0;                                            // :544-548
  X0.wt_lin = c->cor.wt_lin_blend < 1e-16 ? 1e-16 : (c->cor.wt_lin_blend > 1.0 ? 1.0 : c->cor.wt_lin_blend);   // :1139
  X0.rat_lin = 1.5 * X0.Fe_m2 / (X0.wt_lin > 1.0e-16 ? X0.wt_lin : 1.0e-16);
  if (c->cor.F_eff_max_blend <= 2.0) { X0.Fe_m2 = -1.; X0.rat_lin = -1.0; }
  X0.eps_vel = 1.0e-10 * 1.0; X0.h_tiny = c->GV.Angstrom_H;                              // :242-243
  X0.pv_upwind = (c->cor.PV_Adv_Scheme == MOM6X_PV_ADV_UPWIND1);
  const dim3 b = blk2();
  const double vol_neglect = c->GV.H_subroundoff * ((1e-4 * 1.0) * (1e-4 * 1.0));
  // (A barrier-free form -- every thread evaluating q at its own vertex and the one to the south, the western one by a lane
  //  shuffle -- was measured too: 200 registers, 7.1 ms per step against 5.9 for k_corad_fused and 6.5 for the two kernels.)
  static const bool two_kernels = [] { const char *e = getenv("MOM6X_CORAD"); return e && !strcmp(e, "legacy"); }();
  if (!two_kernels && fusable && d.halo >= 3) {   // q, KE, abs_vort through LDS (k_corad_fused); MOM6X_CORAD=legacy: through HBM
    const int kc = (d.nk % 25 == 0) ? 25 : ((d.nk >= KCHUNK) ? KCHUNK : d.nk);
    const dim3 bt(CF_X, CF_Y, 1);
    const int gx = (d.ni + 1 + (CF_X - 2) - 1) / (CF_X - 2), gy = (d.nj + 1 + (CF_Y - 2) - 1) / (CF_Y - 2), gz = (d.nk + kc - 1) / kc;
    constexpr int xcd_order = 1;   // (the launch-order walk of the tiles lost in round 4 and is gone: profiles/README.md)
    const dim3 gt((unsigned)(((gx * gy * gz + 7) / 8) * 8), 1, 1);
    // the default configuration has its own kernel (inputs, q and KE through LDS); every other one the generic k_corad_fused
    const bool lean = (scheme == MOM6X_SADOURNY75_ENERGY) && !c->cor.bound_Coriolis && !c->cor.Coriolis_En_Dis;
    if (lean)
      KLAUNCH(c, "k_corad_lds", k_corad_lds, gt, bt, d, c->G, u, v, uh, vh, CAu, CAv, h, PFu, PFv, diffu, diffv, u_bc, v_bc, uhtr, vhtr, dt_tr,
              c->cor.no_slip, c->cor.KE_Scheme, vol_neglect, kc, gx, gy, gz, xcd_order);
    else
      KLAUNCH(c, "k_corad_fused", k_corad_fused<false>, gt, bt, d, c->G, u, v, uh, vh, CAu, CAv, c->cor.Coriolis_Scheme, c->cor.bound_Coriolis, h,
              c->cor.Coriolis_En_Dis, PFu, PFv, diffu, diffv, u_bc, v_bc, uhtr, vhtr, dt_tr, c->cor.no_slip, c->cor.KE_Scheme, vol_neglect, kc, gx, gy, gz,
              xcd_order);
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  KLAUNCH(c, "k_corad_q", k_corad_q, gridk(nxa(d.ni + 3, -2), d.nj + 3, d.nk, b), b, d, c->G, u, v, h, q, absv, KE,
          c->cor.no_slip, c->cor.KE_Scheme, vol_neglect, Ihq);
  KLAUNCH(c, "k_corad_acc", k_corad_acc, gridk(nxa(d.ni + 1, -1), d.nj + 1, d.nk, b), b, d, c->G, u, v, uh, vh, q, absv, KE,
          CAu, CAv, c->cor.Coriolis_Scheme, c->cor.bound_Coriolis, h, c->cor.Coriolis_En_Dis, PFu, PFv, diffu, diffv, u_bc, v_bc,
          uhtr, vhtr, dt_tr, (const double *)Ihq, X0);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// ---------------------------------------------------------------------------------------------
// PressureForce_FV_Bouss with an equation of state (:1206, :1289-1316): analytic_int_density_dz
// (MOM_EOS.F90:1384) for EOS_LINEAR (MOM_EOS_linear.F90:275-475) and EOS_WRIGHT (MOM_EOS_Wright.F90:389-655),
// and the use_EOS branch of Set_pbce_Bouss (MOM_PressureForce_Montgomery.F90:704-733).
struct EosDev { int form; double Rho_T0_S0, dRho_dT, dRho_dS, dRho_dp; int do_mw, top_mw, ssh_z0; int van_only; double dz_nv; };

namespace {

// dpa and intz_dpa of one cell (the first loop of int_density_dz_linear :373-384 / _wright :554-577)
template <int FORM>
__device__ __forceinline__ void cell_int(const EosDev &E, double rho_ref, double G_e, double GxRho, double I_Rho, double T,
                                         double S, double zt, double zb, double z0, double &dpa, double &intz) {
  const double dz = zt - zb;
  const double p_ave = -GxRho * (0.5 * (zt + zb) - z0);
  if (FORM == MOM6X_EOS_LINEAR) {
    const double C1_6 = 1.0 / 6.0;
    const double rho_anom = (E.Rho_T0_S0 - rho_ref) + E.dRho_dT * T + E.dRho_dS * S + E.dRho_dp * p_ave;
    dpa = G_e * rho_anom * dz;
    intz = 0.5 * G_e * (rho_anom - C1_6 * E.dRho_dp * (GxRho * dz)) * (dz * dz);
  } else {
    const double C1_3 = 1.0 / 3.0, C1_7 = 1.0 / 7.0, C1_9 = 1.0 / 9.0;
    double al0, p0, lambda;
    wright_coefs<FORM>(T, S, al0, p0, lambda);
    const double I_al0 = 1.0 / al0;
    if (FORM == MOM6X_EOS_WRIGHT) {   // int_density_dz_wright, MOM_EOS_Wright.F90:554-577
      const double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
      const double eps = 0.5 * GxRho * dz * I_Lzz, eps2 = eps * eps;
      const double rho_anom = (p0 + p_ave) * (I_Lzz * I_al0) - rho_ref;
      const double rem = I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2)));
      dpa = 1.0 * (G_e * rho_anom * dz - 2.0 * eps * rem);
      intz = 1.0 * (0.5 * G_e * rho_anom * (dz * dz) - dz * (1.0 + eps) * rem);
    } else {                          // int_density_dz_wright_full / _red, MOM_EOS_Wright_full.F90:550-572
      const double I_Lzz = 1.0 / ((p0 + p_ave) + lambda * I_al0);
      const double eps = 0.5 * (GxRho * dz) * I_Lzz, eps2 = eps * eps;
      const double rho_anom = (p0 + p_ave) * (I_Lzz * I_al0) - rho_ref;
      const double rem = (I_Rho * (lambda * (I_al0 * I_al0))) * (eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2))));
      dpa = 1.0 * ((G_e * rho_anom) * dz - 2.0 * eps * rem);
      intz = 1.0 * (0.5 * (G_e * rho_anom) * (dz * dz) - dz * ((1.0 + eps) * rem));
    }
  }
}

// intx_dpa | inty_dpa of the face between columns L and R (:386-430 / :560-607)
template <int FORM>
__device__ __forceinline__ double face_int(const EosDev &E, double rho_ref, double G_e, double GxRho, double I_Rho, double TL,
                                           double SL, double TR, double SR, double ztL, double zbL, double ztR, double zbR,
                                           double z0L, double z0R, double bathyL, double bathyR, double sshL, double sshR,
                                           double dz_neglect, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  double hWght = 0.0;
  if (E.do_mw) hWght = dmax(dmax(0., -bathyL - ztR), -bathyR - ztL);
  if (E.top_mw) hWght = dmax(dmax(hWght, zbR - sshL), zbL - sshR);
  if (FORM == MOM6X_EOS_LINEAR && hWght <= 0.0) {
    const double C1_6 = 1.0 / 6.0;
    const double dzL = ztL - zbL, dzR = ztR - zbR;
    double p_ave = -GxRho * (0.5 * (ztL + zbL) - z0L);
    const double raL = (E.Rho_T0_S0 - rho_ref) + ((E.dRho_dT * TL + E.dRho_dS * SL) + E.dRho_dp * p_ave);
    p_ave = -GxRho * (0.5 * (ztR + zbR) - z0R);
    const double raR = (E.Rho_T0_S0 - rho_ref) + ((E.dRho_dT * TR + E.dRho_dS * SR) + E.dRho_dp * p_ave);
    return G_e * C1_6 * ((dzL * (2.0 * raL + raR)) + (dzR * (2.0 * raR + raL)));
  }
  double LL = 1.0, LR = 0.0, RR = 1.0, RL = 0.0;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_neglect, hR = (ztR - zbR) + dz_neglect;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    LL = (hWght * hL + hR * hL) * iDenom; LR = (hWght * hR) * iDenom;
    RR = (hWght * hR + hR * hL) * iDenom; RL = (hWght * hL) * iDenom;
  }
  double al0L = 0., p0L = 0., lamL = 0., al0R = 0., p0R = 0., lamR = 0.;
  if (FORM != MOM6X_EOS_LINEAR) { wright_coefs<FORM>(TL, SL, al0L, p0L, lamL); wright_coefs<FORM>(TR, SR, al0R, p0R, lamR); }
  double intz[5];
  intz[0] = dpaL; intz[4] = dpaR;
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = (wt_L * LL) + (wt_R * RL), wtT_R = (wt_L * LR) + (wt_R * RR);
    const double dz = (wt_L * (ztL - zbL)) + (wt_R * (ztR - zbR));
    const double p_ave = -GxRho * ((wt_L * (0.5 * (ztL + zbL) - z0L)) + (wt_R * (0.5 * (ztR + zbR) - z0R)));
    if (FORM == MOM6X_EOS_LINEAR) {
      const double rho_anom = (E.Rho_T0_S0 - rho_ref) +
                              ((E.dRho_dT * ((wtT_L * TL) + (wtT_R * TR)) + E.dRho_dS * ((wtT_L * SL) + (wtT_R * SR))) + E.dRho_dp * p_ave);
      intz[m - 1] = G_e * rho_anom * dz;
    } else {
      const double C1_3 = 1.0 / 3.0, C1_7 = 1.0 / 7.0, C1_9 = 1.0 / 9.0;
      const double al0 = (wtT_L * al0L) + (wtT_R * al0R);
      const double p0 = (wtT_L * p0L) + (wtT_R * p0R);
      const double lambda = (wtT_L * lamL) + (wtT_R * lamR);
      const double I_al0 = 1.0 / al0;
      if (FORM == MOM6X_EOS_WRIGHT) {   // MOM_EOS_Wright.F90:601-605
        const double I_Lzz = 1.0 / (p0 + (lambda * I_al0) + p_ave);
        const double eps = 0.5 * GxRho * dz * I_Lzz, eps2 = eps * eps;
        intz[m - 1] = 1.0 * (G_e * dz * ((p0 + p_ave) * (I_Lzz * I_al0) - rho_ref) - 2.0 * eps *
                             I_Rho * (lambda * (I_al0 * I_al0)) * eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2))));
      } else {                          // MOM_EOS_Wright_full.F90:606-610
        const double I_Lzz = 1.0 / ((p0 + p_ave) + lambda * I_al0);
        const double eps = 0.5 * (GxRho * dz) * I_Lzz, eps2 = eps * eps;
        intz[m - 1] = 1.0 * ((G_e * dz) * ((p0 + p_ave) * (I_Lzz * I_al0) - rho_ref) - 2.0 * eps *
                             (I_Rho * (lambda * (I_al0 * I_al0))) * (eps2 * (C1_3 + eps2 * (0.2 + eps2 * (C1_7 + C1_9 * eps2)))));
      }
    }
  }
  return C1_90 * (7.0 * (intz[0] + intz[4]) + 32.0 * (intz[1] + intz[3]) + 12.0 * intz[2]);
}


// ---- use_ALE with PRESSURE_RECONSTRUCTION_SCHEME = 1 (PressureForce_FV.F90:1235-1236, :1287-1296) ---------------------
// density_anomaly_elem_linear (MOM_EOS_linear.F90:74-84) / density_anomaly_elem_buggy_Wright (MOM_EOS_Wright.F90:101-129):
// calculate_density(..., rho_ref=rho_ref) of the quadratures of int_density_dz_generic_plm
template <int FORM>
__device__ __forceinline__ double density_anomaly(const EosDev &E, double T, double S, double pressure, double rho_ref) {
  if (FORM == MOM6X_EOS_LINEAR)
    return (E.Rho_T0_S0 - rho_ref) + ((E.dRho_dT * T + E.dRho_dS * S) + E.dRho_dp * pressure);
  if (FORM == MOM6X_EOS_UNESCO) return unesco::density_anomaly(T, S, pressure, rho_ref);
  if (FORM == MOM6X_EOS_ROQUET_RHO) return roquet::roquet_density_anomaly(T, S, pressure, rho_ref);
  if (FORM == MOM6X_EOS_JACKETT06) return jackett::jackett_density_anomaly(T, S, pressure, rho_ref);
  if (FORM == MOM6X_EOS_ROQUET_SPV) return roquet::roquet_spv_density_anomaly(T, S, pressure, rho_ref);
  typedef WC<FORM> W;   // the same expression in MOM_EOS_Wright.F90:119-128, _full.F90:108-119, _red.F90:108-119
  const double pa_000 = (W::b0 * (1.0 - W::a0 * rho_ref) - rho_ref * W::c0);
  const double al_TS = W::a1 * T + W::a2 * S;
  const double al0 = W::a0 + al_TS;
  const double p_TSp = pressure + (W::b4 * S + T * (W::b1 + (T * (W::b2 + W::b3 * T) + W::b5 * S)));
  const double lam_TS = W::c4 * S + T * (W::c1 + (T * (W::c2 + W::c3 * T) + W::c5 * S));
  return (pa_000 + (p_TSp - rho_ref * (p_TSp * al0 + (W::b0 * al_TS + lam_TS)))) / ((W::c0 + lam_TS) + al0 * (W::b0 + p_TSp));
}

// section 1 of int_density_dz_generic_plm (MOM_density_integrals.F90:587-637): dpa and intz_dpa of one cell by Boole's rule
// MODE 1: linear T, S between the edge values (int_density_dz_generic_plm); 2: parabolic through the edge values and the mean
// (int_density_dz_generic_ppm :1047-1073); 3: the layer mean (int_density_dz_generic_pcm :243-262)
template <int FORM, int MODE>
__device__ __forceinline__ void cell_int_plm(const EosDev &E, double rho_ref, double G_e, double GxRho, double Tt, double Tb,
                                             double St, double Sb, double zt, double zb, double z0, double &dpa, double &intz,
                                             double Tm = 0., double Sm = 0.) {
  const double C1_90 = 1.0 / 90.0;
  const double dz = zt - zb;
  double r5[6];
  double s6 = 0., t6 = 0.;
  if (MODE == 2) { s6 = 3.0 * (2.0 * Sm - (St + Sb)); t6 = 3.0 * (2.0 * Tm - (Tt + Tb)); }
#pragma unroll
  for (int n = 1; n <= 5; n++) {
    const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
    const double p5 = -GxRho * ((zt - z0) - 0.25 * (double)(n - 1) * dz);
    double S5, T5;
    if (MODE == 2) { S5 = wt_t * St + wt_b * (Sb + s6 * wt_t); T5 = wt_t * Tt + wt_b * (Tb + t6 * wt_t); }
    else if (MODE == 3) { S5 = Sm; T5 = Tm; }
    else { S5 = wt_t * St + wt_b * Sb; T5 = wt_t * Tt + wt_b * Tb; }
    r5[n] = density_anomaly<FORM>(E, T5, S5, p5, rho_ref);
  }
  const double rho_anom = C1_90 * (7.0 * (r5[1] + r5[5]) + 32.0 * (r5[2] + r5[4]) + 12.0 * r5[3]);
  dpa = G_e * dz * rho_anom;
  intz = 0.5 * G_e * (dz * dz) * (rho_anom - C1_90 * (16.0 * (r5[4] - r5[2]) + 7.0 * (r5[5] - r5[1])));
}

// sections 2 / 3 (:640-742 / :745-868): intx_dpa | inty_dpa of the face between columns L and R
template <int FORM>
__device__ __forceinline__ double face_int_plm(const EosDev &E, double rho_ref, double G_e, double GxRho, double dz_subroundoff,
                                               double TtL, double TbL, double StL, double SbL, double TtR, double TbR, double StR,
                                               double SbR, double ztL, double zbL, double ztR, double zbR, double z0L, double z0R,
                                               double bathyL, double bathyR, double sshL, double sshR, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  const double mwT = E.do_mw ? 1. : 0., topT = E.top_mw ? 1. : 0., nvT = E.van_only ? 0. : 1.;
  double hWght = mwT * dmax(dmax(0., -bathyL - ztR), -bathyR - ztL);
  const double hWghtTop = topT * dmax(dmax(0., zbR - sshL), zbL - sshR);
  hWght = dmax(hWght, hWghtTop);
  if (((ztL - zbL) > E.dz_nv) && ((ztR - zbR) > E.dz_nv)) hWght = nvT * hWght;
  double Ttl = TtL, Tbl = TbL, Ttr = TtR, Tbr = TbR, Stl = StL, Sbl = SbL, Str = StR, Sbr = SbR;
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
  }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double w_left = 0.25 * (double)(5 - m), w_right = 1.0 - w_left;
    const double dz_x = (w_left * (ztL - zbL)) + (w_right * (ztR - zbR));
    double T15[6], S15[6], p15[6], r15[6];
    T15[1] = (w_left * Ttl) + (w_right * Ttr); T15[5] = (w_left * Tbl) + (w_right * Tbr);
    S15[1] = (w_left * Stl) + (w_right * Str); S15[5] = (w_left * Sbl) + (w_right * Sbr);
    p15[1] = -GxRho * ((w_left * (ztL - z0L)) + (w_right * (ztR - z0R)));
#pragma unroll
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
#pragma unroll
    for (int n = 2; n <= 4; n++) {
      const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
      S15[n] = wt_t * S15[1] + wt_b * S15[5];
      T15[n] = wt_t * T15[1] + wt_b * T15[5];
    }
#pragma unroll
    for (int n = 1; n <= 5; n++) r15[n] = density_anomaly<FORM>(E, T15[n], S15[n], p15[n], rho_ref);
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

// sections 2 / 3 of int_density_dz_generic_ppm (:1075-1183 / :1186-1308): the face between columns L and R, T and S parabolic in
// the vertical through the (thickness-weighted) top, mean and bottom values
template <int FORM>
__device__ __forceinline__ double face_int_ppm(const EosDev &E, double rho_ref, double G_e, double GxRho, double dz_subroundoff,
                                               double TtL, double TbL, double TmL, double StL, double SbL, double SmL, double TtR,
                                               double TbR, double TmR, double StR, double SbR, double SmR, double ztL, double zbL,
                                               double ztR, double zbR, double z0L, double z0R, double bathyL, double bathyR,
                                               double sshL, double sshR, double dpaL, double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  const double mwT = E.do_mw ? 1. : 0., topT = E.top_mw ? 1. : 0., nvT = E.van_only ? 0. : 1.;
  double hWght = mwT * dmax(dmax(0., -bathyL - ztR), -bathyR - ztL);
  const double hWghtTop = topT * dmax(dmax(0., zbR - sshL), zbL - sshR);
  hWght = dmax(hWght, hWghtTop);
  if (((ztL - zbL) > E.dz_nv) && ((ztR - zbR) > E.dz_nv)) hWght = nvT * hWght;
  double Ttl = TtL, Tbl = TbL, Tml = TmL, Ttr = TtR, Tbr = TbR, Tmr = TmR;
  double Stl = StL, Sbl = SbL, Sml = SmL, Str = StR, Sbr = SbR, Smr = SmR;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_subroundoff, hR = (ztR - zbR) + dz_subroundoff;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1. / (hWght * (hR + hL) + hL * hR);
    const double wR = (hWght * hR), wLL = (hWght * hL + hR * hL), wL = (hWght * hL), wRR = (hWght * hR + hR * hL);
    Ttl = (wR * TtR + wLL * TtL) * iDenom; Tbl = (wR * TbR + wLL * TbL) * iDenom; Tml = (wR * TmR + wLL * TmL) * iDenom;
    Ttr = (wL * TtL + wRR * TtR) * iDenom; Tbr = (wL * TbL + wRR * TbR) * iDenom; Tmr = (wL * TmL + wRR * TmR) * iDenom;
    Stl = (wR * StR + wLL * StL) * iDenom; Sbl = (wR * SbR + wLL * SbL) * iDenom; Sml = (wR * SmR + wLL * SmL) * iDenom;
    Str = (wL * StL + wRR * StR) * iDenom; Sbr = (wL * SbL + wRR * SbR) * iDenom; Smr = (wL * SmL + wRR * SmR) * iDenom;
  }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double w_left = 0.25 * (double)(5 - m), w_right = 1.0 - w_left;
    const double T_top = (w_left * Ttl) + (w_right * Ttr), T_mn = (w_left * Tml) + (w_right * Tmr), T_bot = (w_left * Tbl) + (w_right * Tbr);
    const double S_top = (w_left * Stl) + (w_right * Str), S_mn = (w_left * Sml) + (w_right * Smr), S_bot = (w_left * Sbl) + (w_right * Sbr);
    const double dz_x = (w_left * (ztL - zbL)) + (w_right * (ztR - zbR));
    double p15[6], r15[6];
    p15[1] = -GxRho * ((w_left * (ztL - z0L)) + (w_right * (ztR - z0R)));
#pragma unroll
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
    const double s6 = 3.0 * (2.0 * S_mn - (S_top + S_bot)), t6 = 3.0 * (2.0 * T_mn - (T_top + T_bot));
#pragma unroll
    for (int n = 1; n <= 5; n++) {
      const double wt_t = 0.25 * (double)(5 - n), wt_b = 1.0 - wt_t;
      const double S15 = wt_t * S_top + wt_b * (S_bot + s6 * wt_t);
      const double T15 = wt_t * T_top + wt_b * (T_bot + t6 * wt_t);
      r15[n] = density_anomaly<FORM>(E, T15, S15, p15[n], rho_ref);
    }
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

// int_density_dz_generic_pcm (EOS_QUADRATURE), :265-339 / :342-414: layer-mean T, S carried across the face with the
// (possibly thickness-weighted) weights hWt_LL ... hWt_RL
template <int FORM>
__device__ __forceinline__ double face_int_pcm(const EosDev &E, double rho_ref, double G_e, double GxRho, double dz_neglect, double TL,
                                               double SL, double TR, double SR, double ztL, double zbL, double ztR, double zbR, double z0L,
                                               double z0R, double bathyL, double bathyR, double sshL, double sshR, double dpaL,
                                               double dpaR) {
  const double C1_90 = 1.0 / 90.0;
  const double nvT = E.van_only ? 0. : 1.;
  double hWght = 0.0;
  if (E.do_mw) hWght = dmax(dmax(0., -bathyL - ztR), -bathyR - ztL);
  if (E.top_mw) hWght = dmax(dmax(hWght, zbR - sshL), zbL - sshR);
  if (((ztL - zbL) > E.dz_nv) && ((ztR - zbR) > E.dz_nv)) hWght = nvT * hWght;
  double hWt_LL = 1.0, hWt_LR = 0.0, hWt_RR = 1.0, hWt_RL = 0.0;
  if (hWght > 0.) {
    const double hL = (ztL - zbL) + dz_neglect, hR = (ztR - zbR) + dz_neglect;
    const double q = (hL - hR) / (hL + hR);
    hWght = hWght * (q * q);
    const double iDenom = 1.0 / (hWght * (hR + hL) + hL * hR);
    hWt_LL = (hWght * hL + hR * hL) * iDenom; hWt_LR = (hWght * hR) * iDenom;
    hWt_RR = (hWght * hR + hR * hL) * iDenom; hWt_RL = (hWght * hL) * iDenom;
  }
  double intz[6];
  intz[1] = dpaL; intz[5] = dpaR;
#pragma unroll
  for (int m = 2; m <= 4; m++) {
    const double wt_L = 0.25 * (double)(5 - m), wt_R = 1.0 - wt_L;
    const double wtT_L = (wt_L * hWt_LL) + (wt_R * hWt_RL), wtT_R = (wt_L * hWt_LR) + (wt_R * hWt_RR);
    const double dz_x = (wt_L * (ztL - zbL)) + (wt_R * (ztR - zbR));
    const double T15 = (wtT_L * TL) + (wtT_R * TR), S15 = (wtT_L * SL) + (wtT_R * SR);
    double p15[6], r15[6];
    p15[1] = -GxRho * ((wt_L * (ztL - z0L)) + (wt_R * (ztR - z0R)));
#pragma unroll
    for (int n = 2; n <= 5; n++) p15[n] = p15[n - 1] + GxRho * 0.25 * dz_x;
#pragma unroll
    for (int n = 1; n <= 5; n++) r15[n] = density_anomaly<FORM>(E, T15, S15, p15[n], rho_ref);
    intz[m] = (G_e * dz_x * (C1_90 * (7.0 * (r15[1] + r15[5]) + 32.0 * (r15[2] + r15[4]) + 12.0 * r15[3])));
  }
  return C1_90 * (7.0 * (intz[1] + intz[5]) + 32.0 * (intz[2] + intz[4]) + 12.0 * intz[3]);
}

// One thread per (i,j) column, top-down like k_pgf_main; the integrals of the east and north neighbours are
// recomputed by this thread (no 3-D dpa / intz_dpa / intx_dpa arrays).  PLM: the T, S edge values of TS_PLM_edge_values
// (Tt, Tb, St, Sb) and the generic quadratures instead of the layer means and the analytic integrals.
template <int FORM, int MODE>   // MODE 0: analytic integrals; 1: PLM; 2: PPM; 3: layer means by quadrature (EOS_QUADRATURE)
__global__ void __launch_bounds__(256)
k_pgf_main_eos(Dm d, const double *__restrict__ G, const double *__restrict__ h, const double *__restrict__ e,
               const double *__restrict__ Tv, const double *__restrict__ Sv, const double *__restrict__ Tt,
               const double *__restrict__ Tb, const double *__restrict__ St, const double *__restrict__ Sb, EosDev E,
               double *__restrict__ PFu,
               double *__restrict__ PFv, double *__restrict__ pbce, double *__restrict__ eta, double g_Earth, double H_to_Z,
               double Z_to_H, double rho_ref, double GxRho_ref, double Z_ref, double Rho0, double rho0_alt, double h_neglect,
               double dz_neglect, BcFold B) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni || j > d.nj) return;
  if (i < (-1)) return;
  const int st = d.pitch, nz = d.nk;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const bool do_u = (i <= d.ni - 1) && (j >= 0) && (j <= d.nj - 1);
  const bool do_v = (j <= d.nj - 1) && (i >= 0) && (i <= d.ni - 1);
  const double *bathyT = gm(G, d, MOM6X_G_bathyT);
  const double I_Rho0 = 1.0 / Rho0;
  const double G_e = g_Earth, GxRho = G_e * rho0_alt, I_Rho = 1.0 / rho0_alt;   // rho0_int_density :1134-1144
  const double e_top = e[x], e_bot = e[x + (size_t)nz * slab];
  if (eta) eta[x] = e_top * Z_to_H;
  const double ssh0 = e_top, ssh1 = do_u ? e[x + 1] : 0.0, ssh2 = do_v ? e[x + st] : 0.0;
  const double z00 = E.ssh_z0 ? ssh0 : Z_ref, z01 = E.ssh_z0 ? ssh1 : Z_ref, z02 = E.ssh_z0 ? ssh2 : Z_ref;   // Z_0p :1264-1276
  const double b0 = bathyT[x], b1 = do_u ? bathyT[x + 1] : 0.0, b2 = do_v ? bathyT[x + st] : 0.0;
  double pa0 = GxRho_ref * (e_top - Z_ref), pa1 = 0.0, pa2 = 0.0, intx_pa = 0.0, inty_pa = 0.0;
  double cu = 0.0, cv = 0.0;
  if (do_u) { pa1 = GxRho_ref * (ssh1 - Z_ref); intx_pa = 0.5 * (pa0 + pa1); cu = (2.0 * I_Rho0 * gm(G, d, MOM6X_G_IdxCu)[x]); }
  if (do_v) { pa2 = GxRho_ref * (ssh2 - Z_ref); inty_pa = 0.5 * (pa0 + pa2); cv = (2.0 * I_Rho0 * gm(G, d, MOM6X_G_IdyCv)[x]); }
  // Set_pbce_Bouss, use_EOS, no rho_star :704-733 (Rho0 argument = rho0_set_pbce = rho0_alt)
  const double Rho0xG = rho0_alt * g_Earth, G_Rho0 = g_Earth / Rho0;
  const double Ihtot = H_to_Z / ((e_top - e_bot) + dz_neglect);
  double pb = 0.0, T_prev = 0.0, S_prev = 0.0;
  double zt0 = e_top, zt1 = ssh1, zt2 = ssh2;
  for (int k = 0; k < nz; k++) {
    const size_t c = x + (size_t)k * slab, cb = c + slab;
    const double h0 = h[c], T0 = Tv[c], S0 = Sv[c];
    const double zb0 = e[cb];
    double dpa0, iz0;
    double Tt0 = 0., Tb0 = 0., St0 = 0., Sb0 = 0.;
    if (MODE != 0) {
      if (MODE != 3) { Tt0 = Tt[c]; Tb0 = Tb[c]; St0 = St[c]; Sb0 = Sb[c]; }
      cell_int_plm<FORM, MODE>(E, rho_ref, G_e, GxRho, Tt0, Tb0, St0, Sb0, zt0, zb0, z00, dpa0, iz0, T0, S0);
    } else {
      cell_int<FORM>(E, rho_ref, G_e, GxRho, I_Rho, T0, S0, zt0, zb0, z00, dpa0, iz0);
    }
    if (Z_to_H != 1.0) iz0 = iz0 * Z_to_H;
    if (do_u) {
      const double h1 = h[c + 1], T1 = Tv[c + 1], S1 = Sv[c + 1], zb1 = e[cb + 1];
      double dpa1, iz1, intx_dpa;
      if (MODE != 0) {
        double Tt1 = 0., Tb1 = 0., St1 = 0., Sb1 = 0.;
        if (MODE != 3) { Tt1 = Tt[c + 1]; Tb1 = Tb[c + 1]; St1 = St[c + 1]; Sb1 = Sb[c + 1]; }
        cell_int_plm<FORM, MODE>(E, rho_ref, G_e, GxRho, Tt1, Tb1, St1, Sb1, zt1, zb1, z01, dpa1, iz1, T1, S1);
        if (MODE == 1)
          intx_dpa = face_int_plm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, Tt0, Tb0, St0, Sb0, Tt1, Tb1, St1, Sb1, zt0, zb0, zt1, zb1,
                                        z00, z01, b0, b1, ssh0, ssh1, dpa0, dpa1);
        else if (MODE == 2)
          intx_dpa = face_int_ppm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, Tt0, Tb0, T0, St0, Sb0, S0, Tt1, Tb1, T1, St1, Sb1, S1, zt0, zb0,
                                        zt1, zb1, z00, z01, b0, b1, ssh0, ssh1, dpa0, dpa1);
        else
          intx_dpa = face_int_pcm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, T0, S0, T1, S1, zt0, zb0, zt1, zb1, z00, z01, b0, b1, ssh0, ssh1,
                                        dpa0, dpa1);
      } else {
        cell_int<FORM>(E, rho_ref, G_e, GxRho, I_Rho, T1, S1, zt1, zb1, z01, dpa1, iz1);
        intx_dpa = face_int<FORM>(E, rho_ref, G_e, GxRho, I_Rho, T0, S0, T1, S1, zt0, zb0, zt1, zb1, z00, z01, b0, b1,
                                  ssh0, ssh1, dz_neglect, dpa0, dpa1);
      }
      if (Z_to_H != 1.0) iz1 = iz1 * Z_to_H;
      const double pf = (((pa0 * h0 + iz0) - (pa1 * h1 + iz1)) + ((h1 - h0) * intx_pa - (zb1 - zb0) * intx_dpa * Z_to_H)) *
                        (cu / ((h0 + h1) + h_neglect));
      PFu[c] = pf;
      if (B.u_bc) B.u_bc[c] = (B.CAu[c] + pf) + B.diffu[c];
      pa1 = pa1 + dpa1;
      intx_pa = intx_pa + intx_dpa;
      zt1 = zb1;
    }
    if (do_v) {
      const double h2 = h[c + st], T2 = Tv[c + st], S2 = Sv[c + st], zb2 = e[cb + st];
      double dpa2, iz2, inty_dpa;
      if (MODE != 0) {
        double Tt2 = 0., Tb2 = 0., St2 = 0., Sb2 = 0.;
        if (MODE != 3) { Tt2 = Tt[c + st]; Tb2 = Tb[c + st]; St2 = St[c + st]; Sb2 = Sb[c + st]; }
        cell_int_plm<FORM, MODE>(E, rho_ref, G_e, GxRho, Tt2, Tb2, St2, Sb2, zt2, zb2, z02, dpa2, iz2, T2, S2);
        if (MODE == 1)
          inty_dpa = face_int_plm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, Tt0, Tb0, St0, Sb0, Tt2, Tb2, St2, Sb2, zt0, zb0, zt2, zb2,
                                        z00, z02, b0, b2, ssh0, ssh2, dpa0, dpa2);
        else if (MODE == 2)
          inty_dpa = face_int_ppm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, Tt0, Tb0, T0, St0, Sb0, S0, Tt2, Tb2, T2, St2, Sb2, S2, zt0, zb0,
                                        zt2, zb2, z00, z02, b0, b2, ssh0, ssh2, dpa0, dpa2);
        else
          inty_dpa = face_int_pcm<FORM>(E, rho_ref, G_e, GxRho, dz_neglect, T0, S0, T2, S2, zt0, zb0, zt2, zb2, z00, z02, b0, b2, ssh0, ssh2,
                                        dpa0, dpa2);
      } else {
        cell_int<FORM>(E, rho_ref, G_e, GxRho, I_Rho, T2, S2, zt2, zb2, z02, dpa2, iz2);
        inty_dpa = face_int<FORM>(E, rho_ref, G_e, GxRho, I_Rho, T0, S0, T2, S2, zt0, zb0, zt2, zb2, z00, z02, b0, b2,
                                  ssh0, ssh2, dz_neglect, dpa0, dpa2);
      }
      if (Z_to_H != 1.0) iz2 = iz2 * Z_to_H;
      const double pf = (((pa0 * h0 + iz0) - (pa2 * h2 + iz2)) + ((h2 - h0) * inty_pa - (zb2 - zb0) * inty_dpa * Z_to_H)) *
                        (cv / ((h0 + h2) + h_neglect));
      PFv[c] = pf;
      if (B.v_bc) B.v_bc[c] = (B.CAv[c] + pf) + B.diffv[c];
      pa2 = pa2 + dpa2;
      inty_pa = inty_pa + inty_dpa;
      zt2 = zb2;
    }
    pa0 = pa0 + dpa0;
    if (pbce) {
      const double press = -Rho0xG * (zt0 - Z_ref);
      if (k == 0) {
        double rho_in_situ;
        if (FORM == MOM6X_EOS_LINEAR) rho_in_situ = E.Rho_T0_S0 + E.dRho_dT * T0 + E.dRho_dS * S0 + E.dRho_dp * press;
        else if (FORM == MOM6X_EOS_UNESCO) rho_in_situ = unesco::density(T0, S0, press);
        else if (FORM == MOM6X_EOS_ROQUET_RHO) rho_in_situ = roquet::roquet_density(T0, S0, press);
        else if (FORM == MOM6X_EOS_JACKETT06) rho_in_situ = jackett::jackett_density(T0, S0, press);
        else if (FORM == MOM6X_EOS_ROQUET_SPV) rho_in_situ = roquet::roquet_spv_density(T0, S0, press);
        else {
          rho_in_situ = wright_density<FORM>(T0, S0, press);
        }
        pb = G_Rho0 * (1.0 * rho_in_situ) * H_to_Z;
      } else {
        const double T_int = 0.5 * (T_prev + T0), S_int = 0.5 * (S_prev + S0);
        double dR_dT, dR_dS;
        if (FORM == MOM6X_EOS_LINEAR) { dR_dT = E.dRho_dT; dR_dS = E.dRho_dS; }
        else if (FORM == MOM6X_EOS_UNESCO) unesco::density_derivs(T_int, S_int, press, dR_dT, dR_dS);
        else if (FORM == MOM6X_EOS_ROQUET_RHO) roquet::roquet_density_derivs(T_int, S_int, press, &dR_dT, &dR_dS);
        else if (FORM == MOM6X_EOS_JACKETT06) jackett::jackett_density_derivs(T_int, S_int, press, &dR_dT, &dR_dS);
        else if (FORM == MOM6X_EOS_ROQUET_SPV) roquet::roquet_spv_density_derivs(T_int, S_int, press, &dR_dT, &dR_dS);
        else {
          typedef WC<FORM> W;
          double al0, p0, lambda;
          wright_coefs<FORM>(T_int, S_int, al0, p0, lambda);
          if (FORM == MOM6X_EOS_WRIGHT) {   // calculate_density_derivs_elem_buggy_Wright, MOM_EOS_Wright.F90:208-222
            double I_denom2 = 1.0 / (lambda + al0 * (press + p0));
            I_denom2 = I_denom2 * I_denom2;
            dR_dT = I_denom2 * (lambda * (W::b1 + T_int * (2.0 * W::b2 + 3.0 * W::b3 * T_int) + W::b5 * S_int) -
                                (press + p0) * ((press + p0) * W::a1 + (W::c1 + T_int * (W::c2 * 2.0 + W::c3 * 3.0 * T_int) + W::c5 * S_int)));
            dR_dS = I_denom2 * (lambda * (W::b4 + W::b5 * T_int) - (press + p0) * ((press + p0) * W::a2 + (W::c4 + W::c5 * T_int)));
          } else {                          // calculate_density_derivs_elem_Wright_full / _red, MOM_EOS_Wright_full.F90:192-200
            const double den = (lambda + al0 * (press + p0));
            const double I_denom2 = 1.0 / (den * den);
            dR_dT = I_denom2 * (lambda * (W::b1 + (T_int * (2.0 * W::b2 + 3.0 * W::b3 * T_int) + W::b5 * S_int)) -
                                (press + p0) * ((press + p0) * W::a1 + (W::c1 + (T_int * (W::c2 * 2.0 + W::c3 * 3.0 * T_int) + W::c5 * S_int))));
            dR_dS = I_denom2 * (lambda * (W::b4 + W::b5 * T_int) - (press + p0) * ((press + p0) * W::a2 + (W::c4 + W::c5 * T_int)));
          }
        }
        pb = pb + G_Rho0 * ((zt0 - e_bot) * Ihtot) * (dR_dT * (T0 - T_prev) + dR_dS * (S0 - S_prev));
      }
      pbce[c] = pb;
    }
    T_prev = T0; S_prev = S0;
    zt0 = zb0;
  }
}
}  // namespace

extern "C" int mom6x_PressureForce_set_tv(mom6x_ctx *c, const double *T, const double *S, const mom6x_eos_params *eos) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_PressureForce_set_tv: null ctx");
  if (!T) { c->tv_T = nullptr; c->tv_S = nullptr; return MOM6X_OK; }
  REQUIRE(S && eos, MOM6X_EINVAL, "mom6x_PressureForce_set_tv: tv%T without tv%S or tv%eqn_of_state");
  REQUIRE(eos->form >= MOM6X_EOS_LINEAR && eos->form <= MOM6X_EOS_ROQUET_SPV, MOM6X_EUNSUPPORTED,
          "PressureForce: EQN_OF_STATE must be LINEAR, WRIGHT, WRIGHT_FULL, WRIGHT_REDUCED, UNESCO, ROQUET_RHO (NEMO), JACKETT_06 or ROQUET_SPV");
  // analytic_int_density_dz, MOM_EOS.F90:1495-1496
  REQUIRE(eos->form < MOM6X_EOS_UNESCO || eos->EOS_quadrature || eos->Recon_Scheme, MOM6X_EUNSUPPORTED,
          "No analytic integration option is available with this EOS!");
  REQUIRE(eos->Recon_Scheme >= 0 && eos->Recon_Scheme <= 2, MOM6X_EINVAL,
          "PressureForce_FV_init: PRESSURE_RECONSTRUCTION_SCHEME must be 1 (PLM) or 2 (PPM), or 0 without RECONSTRUCT_FOR_PRESSURE");
  REQUIRE(eos->Recon_Scheme != 2 || c->dims.nk >= 4, MOM6X_EUNSUPPORTED,
          "PressureForce_FV: PRESSURE_RECONSTRUCTION_SCHEME = 2 (edge_values_implicit_h4) needs NK >= 4");
  c->tv_T = T; c->tv_S = S; c->eos = *eos;
  return MOM6X_OK;
}

extern "C" int mom6x_PressureForce_init(mom6x_ctx *c, const mom6x_pgf_params *p, const double *Rlay, const double *g_prime) {
  REQUIRE(c && p && Rlay && g_prime, MOM6X_EINVAL, "mom6x_PressureForce_init: null argument");
  HIPCHK(hipSetDevice(c->device));
  c->pgf = *p;
  const size_t n = (size_t)c->dims.nk * sizeof(double);
  if (!c->Rlay) { HIPCHK(hipMalloc(&c->Rlay, n)); HIPCHK(hipMalloc(&c->g_prime, n)); }
  HIPCHK(hipMemcpy(c->Rlay, Rlay, n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->g_prime, g_prime, n, hipMemcpyHostToDevice));
  c->pgf_init = true;
  return MOM6X_OK;
}

extern "C" int mom6x_PressureForce(mom6x_ctx *c, const double *h, double *PFu, double *PFv, double *pbce, double *eta) {
  REQUIRE(c && c->pgf_init, MOM6X_EINVAL, "MOM_PressureForce_FV_Bouss: Module must be initialized before it is used.");
  REQUIRE(h && PFu && PFv, MOM6X_EINVAL, "PressureForce: null array");
  HIPCHK(hipSetDevice(c->device));
  c->pgf_eta_h_written = false;
  const Dm d = c->d;
  double *e;
  int rc;
  if ((rc = ctx_scratch(c, SCR_e, d.nk + 1, &e))) return rc;
  const dim3 b = blk2();
  const mom6x_vgrid &GV = c->GV;
  const double GxRho0 = GV.g_Earth * GV.Rho0;
  const double GxRho_ref = c->pgf.rho_ref_bug ? GxRho0 : GV.g_Earth * c->pgf.rho_ref;
  KLAUNCH(c, "k_pgf_e", k_pgf_e, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, c->G, h, e, GV.H_to_Z);
  if (c->tv_T) {   // use_EOS = associated(tv%eqn_of_state) :1125
    EosDev E;
    E.form = c->eos.form; E.Rho_T0_S0 = c->eos.Rho_T0_S0; E.dRho_dT = c->eos.dRho_dT; E.dRho_dS = c->eos.dRho_dS;
    E.dRho_dp = c->eos.dRho_dp; E.do_mw = c->eos.MassWghtInterp & 1; E.top_mw = (c->eos.MassWghtInterp >> 1) & 1;
    E.ssh_z0 = c->eos.use_SSH_in_Z0p;
    E.van_only = c->eos.MassWghtInterpVanOnly; E.dz_nv = GV.H_to_Z * c->eos.h_nonvanished;   // dz_nonvanished :1128
    const double rho0_alt = c->pgf.rho_ref_bug ? c->pgf.rho_ref : GV.Rho0;   // rho0_int_density = rho0_set_pbce
    const dim3 g = grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b);
    // 0: analytic_int_density_dz; 1: TS_PLM_edge_values + int_density_dz_generic_plm; 2: TS_PPM_edge_values + ..._generic_ppm;
    // 3: EOS_QUADRATURE without a reconstruction: int_density_dz_generic_pcm (int_density_dz, MOM_density_integrals.F90:95-99)
    const int mode = c->eos.Recon_Scheme ? c->eos.Recon_Scheme : (c->eos.EOS_quadrature ? 3 : 0);
    double *Tt = nullptr, *Tb = nullptr, *St = nullptr, *Sb = nullptr;
    if (mode == 1 || mode == 2) {   // TS_PLM_edge_values (MOM_ALE.F90:1495) | TS_PPM_edge_values (:1581): S first, then T
      if ((rc = ctx_scratch(c, SCR_t0, d.nk, &Tt)) || (rc = ctx_scratch(c, SCR_t1, d.nk, &Tb)) ||
          (rc = ctx_scratch(c, SCR_t2, d.nk, &St)) || (rc = ctx_scratch(c, SCR_t3, d.nk, &Sb))) return rc;
      if (mode == 1) {
        if ((rc = mom6x_ALE_PLM_edge_values(c, h, c->tv_S, c->eos.boundary_extrap, St, Sb))) return rc;
        if ((rc = mom6x_ALE_PLM_edge_values(c, h, c->tv_T, c->eos.boundary_extrap, Tt, Tb))) return rc;
      } else {
        if ((rc = mom6x_ALE_PPM_edge_values(c, h, c->tv_S, c->eos.boundary_extrap, St, Sb))) return rc;
        if ((rc = mom6x_ALE_PPM_edge_values(c, h, c->tv_T, c->eos.boundary_extrap, Tt, Tb))) return rc;
      }
    }
#define PGF_EOS(F, P, NAME) KLAUNCH(c, NAME, (k_pgf_main_eos<F, P>), g, b, d, c->G, h, e, c->tv_T, c->tv_S, Tt, Tb, St, Sb, E, PFu, PFv,   \
                                    pbce, eta, GV.g_Earth, GV.H_to_Z, GV.Z_to_H, c->pgf.rho_ref, GxRho_ref, c->pgf.Z_ref, GV.Rho0, \
                                    rho0_alt, GV.H_subroundoff, GV.dZ_subroundoff, c->pgf_fold)
#define PGF_FORM(F, N) do { if (mode == 1) PGF_EOS(F, 1, "k_pgf_main_plm<" N ">"); else if (mode == 2) PGF_EOS(F, 2, "k_pgf_main_ppm<" N ">"); \
                            else if (mode == 3) PGF_EOS(F, 3, "k_pgf_main_pcm<" N ">"); else PGF_EOS(F, 0, "k_pgf_main_eos<" N ">"); } while (0)
    if (E.form == MOM6X_EOS_UNESCO) {   // quadratures only
      if (mode == 1) PGF_EOS(MOM6X_EOS_UNESCO, 1, "k_pgf_main_plm<unesco>");
      else if (mode == 2) PGF_EOS(MOM6X_EOS_UNESCO, 2, "k_pgf_main_ppm<unesco>");
      else PGF_EOS(MOM6X_EOS_UNESCO, 3, "k_pgf_main_pcm<unesco>");
    } else if (E.form == MOM6X_EOS_ROQUET_RHO) {
      if (mode == 1) PGF_EOS(MOM6X_EOS_ROQUET_RHO, 1, "k_pgf_main_plm<roquet_rho>");
      else if (mode == 2) PGF_EOS(MOM6X_EOS_ROQUET_RHO, 2, "k_pgf_main_ppm<roquet_rho>");
      else PGF_EOS(MOM6X_EOS_ROQUET_RHO, 3, "k_pgf_main_pcm<roquet_rho>");
    } else if (E.form == MOM6X_EOS_ROQUET_SPV) {
      if (mode == 1) PGF_EOS(MOM6X_EOS_ROQUET_SPV, 1, "k_pgf_main_plm<roquet_spv>");
      else if (mode == 2) PGF_EOS(MOM6X_EOS_ROQUET_SPV, 2, "k_pgf_main_ppm<roquet_spv>");
      else PGF_EOS(MOM6X_EOS_ROQUET_SPV, 3, "k_pgf_main_pcm<roquet_spv>");
    } else if (E.form == MOM6X_EOS_JACKETT06) {
      if (mode == 1) PGF_EOS(MOM6X_EOS_JACKETT06, 1, "k_pgf_main_plm<jackett06>");
      else if (mode == 2) PGF_EOS(MOM6X_EOS_JACKETT06, 2, "k_pgf_main_ppm<jackett06>");
      else PGF_EOS(MOM6X_EOS_JACKETT06, 3, "k_pgf_main_pcm<jackett06>");
    } else if (E.form == MOM6X_EOS_LINEAR) PGF_FORM(MOM6X_EOS_LINEAR, "linear");
    else if (E.form == MOM6X_EOS_WRIGHT_FULL) PGF_FORM(MOM6X_EOS_WRIGHT_FULL, "wright_full");
    else if (E.form == MOM6X_EOS_WRIGHT_REDUCED) PGF_FORM(MOM6X_EOS_WRIGHT_REDUCED, "wright_red");
    else PGF_FORM(MOM6X_EOS_WRIGHT, "wright");
#undef PGF_FORM
#undef PGF_EOS
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  KLAUNCH(c, "k_pgf_main", k_pgf_main, grid3(nxa(d.ni + 2, -1), d.nj + 2, 1, b), b, d, c->G, h, e, c->Rlay, c->g_prime, PFu, PFv,
          pbce, eta, GV.g_Earth, GV.H_to_Z, GV.Z_to_H, c->pgf.rho_ref, GxRho_ref, c->pgf.Z_ref, 1.0 / GV.Rho0,
          GV.H_subroundoff, GV.dZ_subroundoff, c->pgf_fold);
  c->pgf_eta_h_written = (c->pgf_fold.eta_h != nullptr);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// The velocity update of the RK2 step (:681-694 / :957-966), vertvisc (:557) and vertvisc_remnant (:1231)
// of one direction in ONE column sweep.  The three share the tridiagonal coefficients (b1, d1, c1 depend
// only on a, h, Ray and dt), so the fused kernel reads a_u and h_u once instead of twice, never writes
// and re-reads the un-diffused velocity, and keeps a single c1 array.  Each quantity goes through
// exactly the operations of the separate kernels, in the same order: results are bit-identical.
//   UPD: u_start = mask * (u_in + dtx * (u_bc + u_abt)) is formed on the fly (else u is updated in place)
//   REM: visc_rem is computed alongside (needs the same dt as the velocity solve)
template <int DIR, bool UPD, bool REM>
__global__ void __launch_bounds__(256)
k_vertvisc_fused(Dm d, const double *__restrict__ G, const double *u_in, const double *__restrict__ u_bc,
                 const double *__restrict__ u_abt, double dtx, double *u, double *__restrict__ vr,
                 const double *__restrict__ a_u, const double *__restrict__ h_u, const double *__restrict__ Ray_u,
                 const double *__restrict__ tau, double *__restrict__ c1, double dt, double dt_Rho0, double H_to_RZ,
                 double *__restrict__ tau_bot, DirectStress S) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int nz = d.nk, st = DIR ? d.pitch : 1;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  if (mC > 0.) {
    const double surface_stress = (S.Hmix > 0.0) ? 0.0 : dt_Rho0 * (mC * tau[x]);
    DSWalk W; W.start(S, dt_Rho0, tau[x]);
    double Ray = Ray_u ? Ray_u[x] : 0.;
    double a_k = a_u[x], a_kp = a_u[x + slab];
    double hu = h_u[x];
    double b_denom_1 = hu + dt * (Ray + a_k);
    double b1 = 1.0 / (b_denom_1 + dt * a_kp);
    double d1 = b_denom_1 * b1;
    double u0 = UPD ? mC * (u_in[x] + dtx * (u_bc[x] + u_abt[x])) : u_in[x];
    u0 = W.add(S, u0, x, st);
    double uprev = b1 * (hu * u0 + surface_stress);
    u[x] = uprev;
    double rprev = b1 * hu;
    if (REM) vr[x] = rprev;
#pragma unroll 4
    for (int k = 1; k < nz; k++) {
      const size_t x3 = x + (size_t)k * slab;
      if (Ray_u) Ray = Ray_u[x3];
      a_k = a_kp; a_kp = a_u[x3 + slab];
      hu = h_u[x3];
      u0 = UPD ? mC * (u_in[x3] + dtx * (u_bc[x3] + u_abt[x3])) : u_in[x3];
      u0 = W.add(S, u0, x3, st);
      c1[x3] = dt * a_k * b1;
      b_denom_1 = hu + dt * (Ray + a_k * d1);
      b1 = 1.0 / (b_denom_1 + dt * a_kp);
      d1 = b_denom_1 * b1;
      uprev = (hu * u0 + dt * a_k * uprev) * b1;
      u[x3] = uprev;
      if (REM) { rprev = (hu + dt * a_k * rprev) * b1; vr[x3] = rprev; }
    }
#pragma unroll 4
    for (int k = nz - 2; k >= 0; k--) {
      const size_t x3 = x + (size_t)k * slab;
      const double ck = c1[x3 + slab];
      uprev = u[x3] + ck * uprev;
      u[x3] = uprev;
      if (REM) { rprev = vr[x3] + ck * rprev; vr[x3] = rprev; }
    }
  } else if (UPD) {
    for (int k = 0; k < nz; k++) {
      const size_t x3 = x + (size_t)k * slab;
      u[x3] = mC * (u_in[x3] + dtx * (u_bc[x3] + u_abt[x3]));
    }
  }
  if (tau_bot) {
    double tb = H_to_RZ * (u[x + (size_t)(nz - 1) * slab] * a_u[x + (size_t)nz * slab]);
    if (Ray_u) for (int k = 0; k < nz; k++) tb = tb + H_to_RZ * (Ray_u[x + (size_t)k * slab] * u[x + (size_t)k * slab]);
    tau_bot[x] = tb;
  }
}

// k_vertvisc_fused with the whole column on chip: the forward sweep keeps c1 and the un-substituted velocity
// in registers (2*NK doubles; the 512-entry unified VGPR/AGPR file of a wave that runs alone on its SIMD)
// and the un-substituted remnant in LDS (NK*8 B per lane, lane-minor: conflict-free), so the backward sweep
// never goes back to HBM: 5 words read and 2 written per face-layer against 12.7 with the c1 / u / visc_rem
// round trips.  One wave per work-group; four work-groups (4 * 64 * NK * 8 B of LDS) fill a CU.  With one
// wave per SIMD nothing else hides the HBM latency, so the inputs are fetched VV_G layers ahead into a
// double buffer (3..5 * VV_G loads in flight per lane while the previous group is solved); the scheduling
// fences keep the compiler from hoisting every load of the unrolled column to the top (which spills).
// Same operations in the same order as k_vertvisc_fused: bit-identical.  Ray_u and the direct-stress
// option go through k_vertvisc_fused.
template <int DIR, bool UPD, bool REM, int NKT>
__global__ void __launch_bounds__(64)
k_vertvisc_cols(Dm d, const double *__restrict__ G, const double *u_in, const double *__restrict__ u_bc,
                const double *__restrict__ u_abt, double dtx, double *u, double *__restrict__ vr,
                const double *__restrict__ a_u, const double *__restrict__ h_u,
                const double *__restrict__ tau, double dt, double dt_Rho0, double H_to_RZ,
                double *__restrict__ tau_bot) {
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;
  extern __shared__ double vv_lds[];
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  double *rr = vv_lds + threadIdx.x;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  constexpr int VV_G = UPD ? 4 : 6;
  constexpr int NG = (NK + VV_G - 1) / VV_G;
  double uu[NK], cu[NK];
  double q_a[2][VV_G], q_h[2][VV_G], q_u[2][VV_G], q_b[2][VV_G], q_t[2][VV_G];
  auto fetch = [&](int g, int b) {
#pragma unroll
    for (int m = 0; m < VV_G; m++) {
      const int k = g * VV_G + m;
      if (k < NK && k < nk) {
        const size_t x3 = x + (size_t)k * slab;
        q_a[b][m] = a_u[x3 + slab];
        q_h[b][m] = h_u[x3];
        q_u[b][m] = u_in[x3];
        if (UPD) { q_b[b][m] = u_bc[x3]; q_t[b][m] = u_abt[x3]; }
      }
    }
  };
  if (mC > 0.) {
    const double surface_stress = dt_Rho0 * (mC * tau[x]);
    double a_kp = a_u[x];
    double b1 = 0., d1 = 0., uprev = 0., rprev = 0.;
    fetch(0, 0);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (g + 1 < NG) fetch(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < VV_G; m++) {
        const int k = g * VV_G + m;
        if (k < NK && k < nk) {
          const double a_k = a_kp; a_kp = q_a[g & 1][m];
          const double hu = q_h[g & 1][m];
          const double u0 = UPD ? mC * (q_u[g & 1][m] + dtx * (q_b[g & 1][m] + q_t[g & 1][m])) : q_u[g & 1][m];
          if (k == 0) {
            const double b_denom_1 = hu + dt * (0. + a_k);
            b1 = 1.0 / (b_denom_1 + dt * a_kp);
            d1 = b_denom_1 * b1;
            uprev = b1 * (hu * u0 + surface_stress);
            rprev = b1 * hu;
          } else {
            cu[k] = dt * a_k * b1;
            const double b_denom_1 = hu + dt * (0. + a_k * d1);
            b1 = 1.0 / (b_denom_1 + dt * a_kp);
            d1 = b_denom_1 * b1;
            uprev = (hu * u0 + dt * a_k * uprev) * b1;
            if (REM) rprev = (hu + dt * a_k * rprev) * b1;
          }
          uu[k] = uprev;
          if (REM) rr[k * 64] = rprev;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    u[x + (size_t)(nk - 1) * slab] = uprev;
    if (REM) vr[x + (size_t)(nk - 1) * slab] = rprev;
    const double u_bottom = uprev;   // (uu[nk - 1])
    asm volatile("" ::: "memory");   // the remnant comes back from LDS, not from 75 more live registers
#pragma unroll
    for (int k = NK - 2; k >= 0; k--) {
      if (k >= nk - 1) continue;
      const size_t x3 = x + (size_t)k * slab;
      const double ck = cu[k + 1];
      uprev = uu[k] + ck * uprev;
      u[x3] = uprev;
      if (REM) { rprev = rr[k * 64] + ck * rprev; vr[x3] = rprev; }
    }
    if (tau_bot) tau_bot[x] = H_to_RZ * (u_bottom * a_kp);
  } else {
    if (UPD) {
      for (int k = 0; k < nk; k++) {
        const size_t x3 = x + (size_t)k * slab;
        u[x3] = mC * (u_in[x3] + dtx * (u_bc[x3] + u_abt[x3]));
      }
    }
    if (tau_bot) tau_bot[x] = H_to_RZ * (u[x + (size_t)(nk - 1) * slab] * a_u[x + (size_t)nk * slab]);
  }
}

// The layer counts the on-chip column kernels are built for (anything deeper walks through HBM): 75 as such (the headline's), any
// other count up to the bound with uniform tests on the layer index (mom6x_dev.h COLS_NK_BOUND).
constexpr int VV_NK_BOUND = COLS_NK_BOUND;
#define VV_NK_DISPATCH(nk, CALL) COLS_NK_DISPATCH(nk, CALL)
constexpr int COEF_COLS_NK_BOUND = COLS_NK_BOUND;
#define COEF_NK_DISPATCH(nk, CALL) COLS_NK_DISPATCH(nk, CALL)
static bool vertvisc_cols_usable(int nk, const double *Ray, const DirectStress &S) {
  static const int mode = [] { const char *e = getenv("MOM6X_VERTVISC"); return (e && !strcmp(e, "walk")) ? 0 : 1; }();
  return mode && nk <= VV_NK_BOUND && !Ray && !(S.Hmix > 0.0);
}

template <int DIR>
static void launch_vertvisc_fused(mom6x_ctx *c, bool upd, bool rem, const double *u_in, const double *u_bc,
                                  const double *u_abt, double dtx, double *u, double *vr, const double *a, const double *h,
                                  const double *Ray, const double *tau, double *c1, double dt, double *tau_bot) {
  const Dm d = c->d;
  const dim3 b = blk2();
  const dim3 g = DIR ? grid3(d.ni, d.nj + 1, 1, b) : grid3(nxa(d.ni + 1, -1), d.nj, 1, b);
  const double dt_Rho0 = dt / c->GV.H_to_RZ, HR = c->GV.H_to_RZ;
  const char *nm = DIR ? "k_vertvisc_fused<1>" : "k_vertvisc_fused<0>";
  const DirectStress S = direct_stress_of(c);
  if (vertvisc_cols_usable(d.nk, Ray, S)) {
    const dim3 bc(64, 1, 1);
    const dim3 gc((unsigned)(((DIR ? d.ni : nxa(d.ni + 1, -1)) + 63) / 64), (unsigned)(DIR ? d.nj + 1 : d.nj), 1);
    const char *nc = DIR ? "k_vertvisc_cols<1>" : "k_vertvisc_cols<0>";
#define VVC(U, R, NKT) KLAUNCH_LDS(c, nc, (k_vertvisc_cols<DIR, U, R, NKT>), gc, bc, (rem ? (size_t)NK_OF(NKT) * 64 * sizeof(double) : (size_t)0), d, c->G, u_in, u_bc, u_abt, dtx, u, vr, a, h, tau, dt, dt_Rho0, HR, tau_bot)
#define VVC4(NKT) do { if (upd && rem) VVC(true, true, NKT); else if (upd) VVC(true, false, NKT); else if (rem) VVC(false, true, NKT); else VVC(false, false, NKT); } while (0)
    VV_NK_DISPATCH(d.nk, VVC4);
#undef VVC4
#undef VVC
    return;
  }
#define VVF(U, R) KLAUNCH(c, nm, (k_vertvisc_fused<DIR, U, R>), g, b, d, c->G, u_in, u_bc, u_abt, dtx, u, vr, a, h, Ray, tau, c1, dt, dt_Rho0, HR, tau_bot, S)
  if (upd && rem) VVF(true, true);
  else if (upd) VVF(true, false);
  else if (rem) VVF(false, true);
  else VVF(false, false);
#undef VVF
}

// [u = mask*(u_in + dtx*(u_bc + u_abt));] vertvisc(u, dt); [vertvisc_remnant(vr, dt)] -- see k_vertvisc_fused.
// u_bc == nullptr: no velocity update (u_in is ignored, u is updated in place); vr_u == nullptr: no remnant.
int vertvisc_fused(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc,
                   const double *u_abt, const double *v_abt, double dtx, double *u, double *v, const double *taux,
                   const double *tauy, double dt, double *taux_bot, double *tauy_bot, double *vr_u, double *vr_v) {
  REQUIRE(c && c->a_u, MOM6X_EINVAL, "MOM_vert_friction(visc): Module must be initialized before it is used.");
  HIPCHK(hipSetDevice(c->device));
  double *c1;
  int rc;
  if ((rc = ctx_scratch(c, SCR_c1, c->d.nk, &c1))) return rc;
  const bool upd = (u_bc != nullptr), rem = (vr_u != nullptr);
  launch_vertvisc_fused<0>(c, upd, rem, upd ? u_in : u, u_bc, u_abt, dtx, u, vr_u, c->a_u, c->h_u, c->Ray_u, taux, c1, dt, taux_bot);
  launch_vertvisc_fused<1>(c, upd, rem, upd ? v_in : v, v_bc, v_abt, dtx, v, vr_v, c->a_v, c->h_v, c->Ray_v, tauy, c1, dt, tauy_bot);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// One layer of the bottom-up walk of vertvisc_coef :1357 + find_coupling_coef :2314 (see k_vertvisc_coef): the thickness at the velocity
// point (CS%h_u) of layer K-1 (0-based k = K-1) and the coupling coefficient (CS%a_u) of the interface K below it, from the two cells'
// thicknesses, the velocity the upwinding looks at, and what the walk carries up from the bottom.  Shared by k_vertvisc_coef and
// k_vertvisc_coef_cols, so the two cannot differ in arithmetic.
struct CoefWalk {
  mom6x_vertvisc_params CS;
  double I_Hbbl, I_valBL, kv_bbl, bbl_thick, hn, H_to_Z, h_neglect, dz_neglect, a_cpl_max, I_amax, Dmin;
  double zh, zcol0, zcol1, z_i_below, dz_vel_below;
  __device__ __forceinline__ void init(const mom6x_vertvisc_params &CS_, double bathy0, double bathy1, double I_Hbbl_, double I_valBL_, double kv_bbl_,
                                       double bbl_thick_, double hn_, double H_to_Z_, double h_neglect_, double dz_neglect_, double a_cpl_max_,
                                       double I_amax_) {
    CS = CS_; I_Hbbl = I_Hbbl_; I_valBL = I_valBL_; kv_bbl = kv_bbl_; bbl_thick = bbl_thick_; hn = hn_; H_to_Z = H_to_Z_;
    h_neglect = h_neglect_; dz_neglect = dz_neglect_; a_cpl_max = a_cpl_max_; I_amax = I_amax_;
    Dmin = dmin(bathy0, bathy1);
    zh = 0.; zcol0 = -bathy0; zcol1 = -bathy1;
    z_i_below = 0.;          // z_i(k+1): the interface below the layer being worked on
    dz_vel_below = 0.;       // dz_vel(k+1)
  }
  __device__ __forceinline__ void layer(int K, int nz, double h0, double h1, double uk, double z_t, bool have_Kv_add, double Kv_add,
                                        double &h_u_out, double &a_out) {
    const double dz0 = H_to_Z * h0, dz1 = H_to_Z * h1;
    const double h_harm = 2. * h0 * h1 / (h0 + h1 + h_neglect);
    const double h_arith = 0.5 * (h1 + h0);
    const double h_delta = h1 - h0;
    const double dz_harm = 2. * dz0 * dz1 / (dz0 + dz1 + dz_neglect);
    const double dz_arith = 0.5 * (dz1 + dz0);
    double hvel, dz_vel, z_i_top;
    if (CS.harmonic_visc) {
      hvel = h_harm; dz_vel = dz_harm;
      if (uk * h_delta < 0) {
        const double z2 = z_i_below;
        const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        hvel = (1. - botfn) * h_harm + botfn * h_arith;
        dz_vel = (1. - botfn) * dz_harm + botfn * dz_arith;
      }
      z_i_top = z_i_below + dz_harm * I_Hbbl;
    } else {
      zcol0 = zcol0 + dz0; zcol1 = zcol1 + dz1;
      zh = zh + dz_harm;
      const double z_clear = dmax(zcol0, zcol1) + Dmin;
      z_i_top = dmax(zh, z_clear) * I_Hbbl;
      hvel = h_arith; dz_vel = dz_arith;
      if (uk * h_delta > 0.) {
        if (zh * I_Hbbl < CS.harm_BL_val) {
          hvel = h_harm; dz_vel = dz_harm;
        } else {
          double z2_wt = 1.;
          if (zh * I_Hbbl < 2. * CS.harm_BL_val) z2_wt = dmax(0., dmin(1., zh * I_Hbbl * I_valBL - 1.));
          const double z2 = z2_wt * (dmax(zh, z_clear) * I_Hbbl);
          const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
          hvel = (1. - botfn) * h_arith + botfn * h_harm;
          dz_vel = (1. - botfn) * dz_arith + botfn * dz_harm;
        }
      }
    }
    h_u_out = hvel + h_neglect;                                    // CS%h_u :1868-1872
    double a_cpl;
    if (K == nz) {                                                 // :2543-2558
      if (CS.bottomdraglaw) {
        const double dhc = dz_vel * 0.5;
        a_cpl = kv_bbl / ((dmin(dhc, bbl_thick) + hn) + I_amax * kv_bbl);
      } else if (fabs(CS.Kv_extra_bbl) > 0.0) {
        a_cpl = (CS.Kv + CS.Kv_extra_bbl) / ((0.5 * dz_vel + hn) + I_amax * (CS.Kv + CS.Kv_extra_bbl));
      } else {
        a_cpl = CS.Kv / ((0.5 * dz_vel + hn) + I_amax * CS.Kv);
      }
    } else {                                                       // :2418-2540, Fortran K+1 between layers k and k+1
      double Kv_tot = CS.Kv;
      if (CS.Kvml_invZ2 > 0.) Kv_tot = CS.Kv + CS.Kvml_invZ2 / ((z_t * z_t) * (1. + 0.09 * z_t * z_t * z_t * z_t * z_t * z_t));
      if (have_Kv_add) Kv_tot = Kv_tot + Kv_add;
      if (CS.bottomdraglaw) {
        const double z2 = z_i_below;
        const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + (kv_bbl - CS.Kv) * botfn;
        const double dhc = 0.5 * (dz_vel_below + dz_vel);
        double h_shear;
        if (dhc > bbl_thick) h_shear = ((1. - botfn) * dhc + botfn * bbl_thick) + hn;
        else h_shear = dhc + hn;
        a_cpl = Kv_tot / (h_shear + (I_amax * Kv_tot));
      } else if (fabs(CS.Kv_extra_bbl) > 0.0) {
        const double z2 = z_i_below;
        const double botfn = 1. / (1. + 0.09 * z2 * z2 * z2 * z2 * z2 * z2);
        Kv_tot = Kv_tot + CS.Kv_extra_bbl * botfn;
        const double h_shear = 0.5 * (dz_vel_below + dz_vel + hn);
        a_cpl = Kv_tot / (h_shear + I_amax * Kv_tot);
      } else {
        const double h_shear = 0.5 * (dz_vel_below + dz_vel + hn);
        a_cpl = Kv_tot / (h_shear + I_amax * Kv_tot);
      }
    }
    a_out = dmin(a_cpl_max, a_cpl);                                // CS%a_u :1863-1867
    z_i_below = z_i_top; dz_vel_below = dz_vel;
  }
};

// ---------------------------------------------------------------------------------------------
// vertvisc_coef :1357 + find_coupling_coef :2314 for one direction: one thread per face column.
// Everything the coupling coefficient of interface K needs (z_i(K), dz_vel of the layers above and below) is
// available while the column is walked bottom-up, so a_cpl is formed in the same sweep and no 3-D temporaries
// (hvel, dz_vel, dz_harm, z_i, a_cpl of the reference) exist.  The only top-down quantity is the mixed-layer
// coordinate z_t of KV_ML_INVZ2 (:2420-2436): when that option is on, a first top-down walk leaves z_t(K) in
// the a array, where the main sweep picks it up before overwriting it.
// dz = H_to_Z*h (thickness_to_dz, MOM_interface_heights.F90:892).
// MODE selects the velocity the upwinding looks at: 0: u as given; 1: the predictor estimate of :591-598,
// mask*(u + dtx*u_bc); 2: that of :681-694 / :957-966, mask*(u + dtx*(u_bc + u_abt)) -- formed on the fly with the
// reference's expression, so the step never has to write and re-read up/vp just for this routine.
template <int DIR, int MODE>
__global__ void __launch_bounds__(256)
k_vertvisc_coef(Dm d, const double *__restrict__ G, mom6x_vertvisc_params CS, const double *u, double *u_out,
                const double *__restrict__ u_bc, const double *__restrict__ u_abt, double dtx,
                const double *__restrict__ h, const double *__restrict__ Kv_bbl, const double *__restrict__ bbl_thick_in,
                const double *__restrict__ Kv_shear, double *__restrict__ a_out, double *__restrict__ h_out, double H_to_Z,
                double h_neglect, double dz_neglect, double a_cpl_max, double I_amax, LayerAccelSrc LA) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int nz = d.nk, st = DIR ? d.pitch : 1;
  const size_t x = ix2(d, i, j), y = x + st, slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  // MODE 3: accel_layer_u of btstep_layer_accel (MOM_barotropic.F90:3432-3504) formed here from pbce and the barotropic solver's
  // 2-D results (the expression of k_layer_accel, barotropic.hip) instead of being read as u_abt
  double la_e0 = 0., la_e1 = 0., la_g0 = 0., la_g1 = 0., la_a = 0., la_Idx = 0.;
  if (MODE == 3) {
    la_e0 = LA.e_anom[x]; la_e1 = LA.e_anom[y]; la_g0 = LA.g_own[x]; la_g1 = LA.g_nbr[y]; la_a = LA.a2d[x];
    la_Idx = gm(G, d, DIR ? MOM6X_G_IdyCv : MOM6X_G_IdxCu)[x];
  }
  auto abt = [&](size_t c) -> double {
    if (MODE != 3) return u_abt[c];
    double a = (la_a - (((LA.pbce[c + st] - la_g1) * la_e1) - ((LA.pbce[c] - la_g0) * la_e0)) * la_Idx);
    if (fabs(a) < LA.underflow) a = 0.0;
    return a;
  };
  if (!(mC > 0.)) {   // do_i :1514-1516
    if (u_out && MODE >= 2)   // (the velocity estimate the solve that follows starts from, see below: masked faces too)
      for (int k = 0; k < nz; k++) { const size_t c = x + (size_t)k * slab; u_out[c] = mC * (u[c] + dtx * (u_bc[c] + abt(c))); }
    return;
  }
  const double *bathyT = gm(G, d, MOM6X_G_bathyT);
  double I_valBL = 0.0; if (CS.harm_BL_val > 0.0) I_valBL = 1.0 / CS.harm_BL_val;
  double I_Hbbl = 1. / (CS.Hbbl + dz_neglect), kv_bbl = 0.0, bbl_thick = 0.0;
  if (CS.bottomdraglaw) {
    kv_bbl = Kv_bbl[x];
    bbl_thick = bbl_thick_in[x] + dz_neglect;
    I_Hbbl = 1. / bbl_thick;
  }
  const double hn = dz_neglect;   // h_neglect of find_coupling_coef :2390
  if (CS.Kvml_invZ2 > 0.) {       // z_t(K), K = 2..nz, top-down
    const double I_Hmix = 1. / (CS.Hmix + hn);
    double z_t = hn * I_Hmix;
    for (int K = 1; K < nz; K++) {
      const double dz0 = H_to_Z * h[x + (size_t)(K - 1) * slab], dz1 = H_to_Z * h[y + (size_t)(K - 1) * slab];
      z_t = z_t + (2. * dz0 * dz1 / (dz0 + dz1 + dz_neglect)) * I_Hmix;
      a_out[x + (size_t)K * slab] = z_t;
    }
  }
  CoefWalk W;
  W.init(CS, bathyT[x], bathyT[y], I_Hbbl, I_valBL, kv_bbl, bbl_thick, hn, H_to_Z, h_neglect, dz_neglect, a_cpl_max, I_amax);
  for (int k = nz - 1; k >= 0; k--) {
    const size_t c = x + (size_t)k * slab;
    const double h0 = h[c], h1 = h[c + st];
    double uk = u[c];
    if (MODE == 1) uk = mC * (uk + dtx * u_bc[c]);
    if (MODE >= 2) {
      uk = mC * (uk + dtx * (u_bc[c] + abt(c)));
      // u_out: this IS the velocity the RK2 step hands to vertvisc next (:681-694 / :957-966); written here, the solve reads one
      // array instead of three (u_out may be u itself: every thread reads and writes its own column only)
      if (u_out) u_out[c] = uk;
    }
    const int K = k + 1;   // the interface below this layer (bottom: K = nz)
    double z_t = 0.0, Kv_add = 0.0;
    if (K < nz) {
      if (CS.Kvml_invZ2 > 0.) z_t = a_out[x + (size_t)K * slab];
      if (Kv_shear) Kv_add = 0.5 * (Kv_shear[x + (size_t)K * slab] + Kv_shear[y + (size_t)K * slab]);
    }
    double hu, a;
    W.layer(K, nz, h0, h1, uk, z_t, Kv_shear != nullptr, Kv_add, hu, a);
    h_out[c] = hu;                                                 // CS%h_u :1868-1872
    a_out[x + (size_t)K * slab] = a;                               // CS%a_u :1863-1867
  }
  a_out[x] = dmin(a_cpl_max, 0.0);   // a_cpl(:,:,1) stays 0 without shelves / dynamic mixed-layer viscosity
}

#ifndef CC_GROUP
#define CC_GROUP 5
#endif
#ifndef CC_GROUP1
#define CC_GROUP1 8   // (MODE 1 holds no pbce pair and no velocity: 8 layers ahead, 1.53-1.58 against 1.60-1.62 ms per launch on average; profiles/r05_ab_vv.txt)
#endif
#ifndef CC_UG
#define CC_UG 8
#endif
// k_vertvisc_coef_cols, pass 1: the coefficients of the pair P of layer groups (2P and 2P+1, G layers each, counted from the bottom) go
// to their registers -- every index a constant.
template <int NK, int G, int P>
__device__ __forceinline__ void cc_file(double (&aa)[NK + 1], const double (&t_a)[2][G]) {
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int m = 0; m < G; m++) {
      const int k = NK - 1 - ((2 * P + b) * G + m);
      if (k >= 0) aa[k + 1] = t_a[b][m];
    }
}
template <int NK, int G>
__device__ __forceinline__ void cc_file_switch(int p, double (&aa)[NK + 1], const double (&t_a)[2][G]) {
  static_assert((NK + G - 1) / G <= 40, "cc_file_switch: at most 20 pairs of groups");
  switch (p) {
#define CC_CASE(P) case P: cc_file<NK, G, P>(aa, t_a); break;
    CC_CASE(0) CC_CASE(1) CC_CASE(2) CC_CASE(3) CC_CASE(4) CC_CASE(5) CC_CASE(6) CC_CASE(7) CC_CASE(8) CC_CASE(9)
    CC_CASE(10) CC_CASE(11) CC_CASE(12) CC_CASE(13) CC_CASE(14) CC_CASE(15) CC_CASE(16) CC_CASE(17) CC_CASE(18) CC_CASE(19)
#undef CC_CASE
    default: break;
  }
}

// vertvisc_coef + vertvisc [+ vertvisc_remnant] (MODE 3: :737-767 and :1002-1022 of the RK2 step) or vertvisc_coef + vertvisc_remnant
// (MODE 1: :602-610) in ONE kernel per direction.  k_vertvisc_coef writes a_u (NK+1 levels) and h_u for k_vertvisc_cols /
// k_vertvisc_remnant_cols to read straight back.  The coefficients are a bottom-up recurrence (z_i counts from the bottom) and the Thomas
// sweep runs top-down, so the column of coefficients has to wait on chip: a_u in registers, h_u in LDS -- and as the forward sweep consumes
// them it puts c1 into a_u's registers and the un-substituted remnant into h_u's LDS words, so the kernel holds what k_vertvisc_cols holds
// (2 NK doubles in the 512-entry register file of a wave alone on its SIMD, NK in LDS).
//   pass 1 (bottom-up): velocity estimate, h_u, a_u, the inputs fetched CC_G layers ahead.  The walk is a LOOP over pairs of groups (75
//     copies of the layer's ~300 instructions would not fit the instruction cache), but a_u can only live in registers if every index
//     is a constant: each pair leaves its coefficients in temporaries and a switch over the pair's number files them (cc_file<P>).
//     The velocity estimate goes to its place in the result array and comes back in pass 2 (150 more registers would spill:
//     measured, 2.1 against 1.75 ms); u may be u_in: a layer's inputs are read before its estimate is written.
//   pass 2 / 3: k_vertvisc_cols's sweeps from the chip.
// WRITE_COEF: a_u and h_u are also written (CS%a_u, CS%h_u stay what the step's LAST vertvisc_coef made them; the coefficients of the
// earlier stages are replaced before anybody can look, unless a vertvisc_remnant of their own follows).
// Words per face-layer: MODE 3: u, u_bc, pbce, h in; the estimate out and in; u [, visc_rem] out [; a_u, h_u out] = 8-10 where the
// pair moves 12-13; MODE 1: u, u_bc, h in, visc_rem out = 4 against 8.  The same expressions in the same order as the kernels it
// replaces (CoefWalk; the sweeps are copies): bit-identical.  KV_ML_INVZ2 (a top-down pre-pass through a_u), Rayleigh drag and the
// direct-stress option stay with the separate kernels.
template <int DIR, int MODE, bool REM, bool WRITE_COEF, int NKT>
__global__ void __launch_bounds__(64)
k_vertvisc_coef_cols(Dm d, const double *__restrict__ G, mom6x_vertvisc_params CS, const double *u_in, const double *__restrict__ u_bc,
                     double dtx, const double *__restrict__ h, const double *__restrict__ Kv_bbl, const double *__restrict__ bbl_thick_in,
                     const double *__restrict__ Kv_shear, double *__restrict__ a_out, double *h_out, double H_to_Z,
                     double h_neglect, double dz_neglect, double a_cpl_max, double I_amax, LayerAccelSrc LA,
                     double *u, double *__restrict__ vr, const double *__restrict__ tau, double dt, double dt_Rho0, double H_to_RZ,
                     double *__restrict__ tau_bot) {
  static_assert(MODE == 1 || MODE == 3, "k_vertvisc_coef_cols: MODE 1 (coefficients + remnant) or 3 (coefficients + solve)");
  static_assert(MODE == 3 || REM, "k_vertvisc_coef_cols: MODE 1 makes the remnant");
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;   // (NKT < 0: the arrays and unrolled loops have NK slots, the column nk <= NK layers)
  extern __shared__ double cc_lds[];
  // (DIR = 1: a block column's rows on ONE XCD, so that the row north of a face column -- the next work-group's own row -- meets it in
  //  that XCD's L2, was measured: 1.77-1.81 against 1.56 ms per launch; rows along blockIdx.y it is.)
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * 64 + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int st = DIR ? d.pitch : 1;
  const size_t x = ix2(d, i, j), y = x + st, slab = (size_t)d.slab;
  double *hh = cc_lds + threadIdx.x;               // h_u(k) at hh[k * 64], later the un-substituted remnant
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[x];
  // accel_layer_u of btstep_layer_accel formed here (k_vertvisc_coef, MODE 3)
  double la_e0 = 0., la_e1 = 0., la_g0 = 0., la_g1 = 0., la_a = 0., la_Idx = 0.;
  if (MODE == 3) {
    la_e0 = LA.e_anom[x]; la_e1 = LA.e_anom[y]; la_g0 = LA.g_own[x]; la_g1 = LA.g_nbr[y]; la_a = LA.a2d[x];
    la_Idx = gm(G, d, DIR ? MOM6X_G_IdyCv : MOM6X_G_IdxCu)[x];
  }
  auto abt_of = [&](double pb0, double pb1) -> double {
    double a = (la_a - (((pb1 - la_g1) * la_e1) - ((pb0 - la_g0) * la_e0)) * la_Idx);
    if (fabs(a) < LA.underflow) a = 0.0;
    return a;
  };
  if (!(mC > 0.)) {   // do_i :1514-1516: (MODE 3) the velocity estimate of the masked faces too; no coefficients, no solve
    if (MODE == 3) {
      double ul = 0.0;
      for (int k = 0; k < nk; k++) {
        const size_t c = x + (size_t)k * slab;
        ul = mC * (u_in[c] + dtx * (u_bc[c] + abt_of(LA.pbce[c], LA.pbce[c + st])));
        u[c] = ul;
      }
      if (tau_bot) tau_bot[x] = H_to_RZ * (ul * a_out[x + (size_t)nk * slab]);
    }
    return;
  }
  const double *bathyT = gm(G, d, MOM6X_G_bathyT);
  double I_valBL = 0.0; if (CS.harm_BL_val > 0.0) I_valBL = 1.0 / CS.harm_BL_val;
  double I_Hbbl = 1. / (CS.Hbbl + dz_neglect), kv_bbl = 0.0, bbl_thick = 0.0;
  if (CS.bottomdraglaw) {
    kv_bbl = Kv_bbl[x];
    bbl_thick = bbl_thick_in[x] + dz_neglect;
    I_Hbbl = 1. / bbl_thick;
  }
  CoefWalk W;
  W.init(CS, bathyT[x], bathyT[y], I_Hbbl, I_valBL, kv_bbl, bbl_thick, dz_neglect, H_to_Z, h_neglect, dz_neglect, a_cpl_max, I_amax);
  double aa[NK + 1];                               // a_u(K), then c1(k)
  constexpr bool EST_LDS = (MODE == 3) && WRITE_COEF;
  // ---- pass 1
  constexpr int CC_G = (MODE == 1) ? CC_GROUP1 : CC_GROUP;   // layers per group of the walk (MODE 1 holds no pbce pair)
  constexpr int NG = (NK + CC_G - 1) / CC_G, NP = (NG + 1) / 2;
  double q_u[2][CC_G], q_b[2][CC_G], q_p0[2][CC_G], q_p1[2][CC_G], q_h0[2][CC_G], q_h1[2][CC_G];
  double t_a[2][CC_G];
  double a_bot = 0.;
  auto fetch = [&](int g, const int b) {
#pragma unroll
    for (int m = 0; m < CC_G; m++) {
      const int k = NK - 1 - (g * CC_G + m);
      if (k >= 0) {
        // (NKT < 0: a slot below the column's bottom fetches the bottom layer once more instead of standing under a test on the
        //  layer index: the number of loads in flight stays a constant of the code and the waits for them stay partial)
        const size_t c = x + (size_t)(NK_EXACT(NKT) ? k : min(k, nk - 1)) * slab;
        q_u[b][m] = u_in[c]; q_b[b][m] = u_bc[c];
        if (MODE == 3) { q_p0[b][m] = LA.pbce[c]; q_p1[b][m] = LA.pbce[c + st]; }
        q_h0[b][m] = h[c]; q_h1[b][m] = h[c + st];
      }
    }
  };
  auto group = [&](int g, const int b) {
#pragma unroll
    for (int m = 0; m < CC_G; m++) {
      const int k = NK - 1 - (g * CC_G + m);
      if (k >= 0 && k < nk) {
        const size_t c = x + (size_t)k * slab;
        const double uk = (MODE == 3) ? mC * (q_u[b][m] + dtx * (q_b[b][m] + abt_of(q_p0[b][m], q_p1[b][m])))
                                      : mC * (q_u[b][m] + dtx * q_b[b][m]);
        const int K = k + 1;
        double Kv_add = 0.0;
        if (K < nk && Kv_shear) Kv_add = 0.5 * (Kv_shear[x + (size_t)K * slab] + Kv_shear[y + (size_t)K * slab]);
        double hu, a;
        W.layer(K, nk, q_h0[b][m], q_h1[b][m], uk, 0.0, Kv_shear != nullptr, Kv_add, hu, a);
        t_a[b][m] = a;
        if (k == nk - 1) a_bot = a;                  // a_u at the bottom interface (aa[nk]): the walk's first layer
        // between the passes: h_u in LDS and the estimate through the result array -- or, when h_u is written anyway, the estimate
        // in LDS and h_u back from its array (a word less)
        if (EST_LDS) hh[k * 64] = uk; else { hh[k * 64] = hu; if (MODE == 3) u[c] = uk; }
        if (WRITE_COEF) { h_out[c] = hu; a_out[x + (size_t)K * slab] = a; }
      }
      __builtin_amdgcn_sched_barrier(0);             // (one layer's temporaries at a time)
    }
  };
  fetch(0, 0);
#pragma unroll 1
  for (int p = 0; p < NP; p++) {
    if (2 * p + 1 < NG) fetch(2 * p + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    group(2 * p, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (2 * p + 2 < NG) fetch(2 * p + 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (2 * p + 1 < NG) group(2 * p + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    cc_file_switch<NK, CC_G>(p, aa, t_a);
  }
  aa[0] = dmin(a_cpl_max, 0.0);                    // a_cpl(:,:,1) stays 0 without shelves / dynamic mixed-layer viscosity
  if (WRITE_COEF) a_out[x] = aa[0];
  asm volatile("" ::: "memory");
  // ---- pass 2: the forward sweep of k_vertvisc_cols / k_vertvisc_remnant_cols from the chip; the velocity estimate comes back
  //      (written a column's walk ago: mostly from the L2) U_G layers ahead of the sweep
  const double surface_stress = (MODE == 3) ? dt_Rho0 * (mC * tau[x]) : 0.0;
  double b1 = 0., d1 = 0., uprev = 0., rprev = 0.;
  double uu[(MODE == 3) ? NK : 1];
  constexpr int U_G = CC_UG;
  const double *back = EST_LDS ? (const double *)h_out : (const double *)u;   // what comes back from memory: h_u or the estimate
  if (MODE == 3) {
#pragma unroll
    for (int k = 0; k < U_G && k < NK; k++) uu[k] = back[x + (size_t)min(k, nk - 1) * slab];
  }
#pragma unroll
  for (int k = 0; k < NK; k++) {
    // (NKT < 0: the loads are NOT under the test on the layer index -- a slot beyond the column reads the bottom layer again --, so
    //  that the memory counter of the loads in flight stays exact across the tests; with them inside, the compiler waits for every
    //  load at every join: 70 instead of 15 s_waitcnt vmcnt(0), and the kernel ran at half its speed)
    if (MODE == 3 && k + U_G < NK) uu[(MODE == 3) ? k + U_G : 0] = back[x + (size_t)min(k + U_G, nk - 1) * slab];
    if (k < nk) {
    const double a_k = aa[k], a_kp = aa[k + 1];
    const double from_lds = hh[k * 64], from_mem = (MODE == 3) ? uu[(MODE == 3) ? k : 0] : 0.0;
    const double hu = EST_LDS ? from_mem : from_lds;
    const double u0 = EST_LDS ? from_lds : from_mem;
    if (k == 0) {
      const double b_denom_1 = hu + dt * (0. + a_k);
      b1 = 1.0 / (b_denom_1 + dt * a_kp);
      d1 = b_denom_1 * b1;
      if (MODE == 3) uprev = b1 * (hu * u0 + surface_stress);
      rprev = b1 * hu;
    } else {
      aa[k] = dt * a_k * b1;                         // c1(k)
      const double b_denom_1 = hu + dt * (0. + a_k * d1);
      b1 = 1.0 / (b_denom_1 + dt * a_kp);
      d1 = b_denom_1 * b1;
      if (MODE == 3) uprev = (hu * u0 + dt * a_k * uprev) * b1;
      if (REM) rprev = (hu + dt * a_k * rprev) * b1;
    }
    if (MODE == 3) uu[(MODE == 3) ? k : 0] = uprev;
    if (REM) hh[k * 64] = rprev;
    }
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  if (MODE == 3) u[x + (size_t)(nk - 1) * slab] = uprev;
  if (REM) vr[x + (size_t)(nk - 1) * slab] = rprev;
  const double u_bot = uprev;   // uu[nk - 1]
  asm volatile("" ::: "memory");
  // ---- pass 3: back substitution
#pragma unroll
  for (int k = NK - 2; k >= 0; k--) {
    if (k >= nk - 1) continue;
    const size_t x3 = x + (size_t)k * slab;
    const double ck = aa[k + 1];
    if (MODE == 3) { uprev = uu[(MODE == 3) ? k : 0] + ck * uprev; u[x3] = uprev; }
    if (REM) { rprev = hh[k * 64] + ck * rprev; vr[x3] = rprev; }
  }
  if (MODE == 3 && tau_bot) tau_bot[x] = H_to_RZ * (u_bot * a_bot);
}

extern "C" int mom6x_vertvisc_init(mom6x_ctx *c, const mom6x_vertvisc_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_vertvisc_init: null argument");
  HIPCHK(hipSetDevice(c->device));
  c->vv = *p;
  const size_t n3 = (size_t)c->dims.slab * c->dims.nk, n3i = (size_t)c->dims.slab * (c->dims.nk + 1);
  if (!c->vv_a_u) {
    HIPCHK(hipMalloc(&c->vv_a_u, n3i * sizeof(double))); HIPCHK(hipMalloc(&c->vv_a_v, n3i * sizeof(double)));
    HIPCHK(hipMalloc(&c->vv_h_u, n3 * sizeof(double))); HIPCHK(hipMalloc(&c->vv_h_v, n3 * sizeof(double)));
  }
  HIPCHK(hipMemsetAsync(c->vv_a_u, 0, n3i * sizeof(double), c->stream)); HIPCHK(hipMemsetAsync(c->vv_a_v, 0, n3i * sizeof(double), c->stream));
  HIPCHK(hipMemsetAsync(c->vv_h_u, 0, n3 * sizeof(double), c->stream)); HIPCHK(hipMemsetAsync(c->vv_h_v, 0, n3 * sizeof(double), c->stream));
  c->a_u = c->vv_a_u; c->a_v = c->vv_a_v; c->h_u = c->vv_h_u; c->h_v = c->vv_h_v;
  c->vv_init = true;
  return MOM6X_OK;
}

extern "C" int mom6x_vertvisc_set_visc(mom6x_ctx *c, const double *Kv_bbl_u, const double *Kv_bbl_v, const double *bbl_thick_u,
                                       const double *bbl_thick_v, const double *Kv_shear, const double *Ray_u, const double *Ray_v) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_vertvisc_set_visc: null ctx");
  REQUIRE((Ray_u != nullptr) == (Ray_v != nullptr), MOM6X_EINVAL, "mom6x_vertvisc_set_visc: Ray_u and Ray_v come together");
  c->Kv_bbl_u = Kv_bbl_u; c->Kv_bbl_v = Kv_bbl_v; c->bbl_thick_u = bbl_thick_u; c->bbl_thick_v = bbl_thick_v;
  c->Kv_shear = Kv_shear; c->Ray_u = Ray_u; c->Ray_v = Ray_v;
  return MOM6X_OK;
}

extern "C" double *mom6x_vertvisc_field(mom6x_ctx *c, int which) {
  if (!c || !c->vv_init) return nullptr;
  double *t[] = { c->vv_a_u, c->vv_a_v, c->vv_h_u, c->vv_h_v };
  return (which >= 0 && which < 4) ? t[which] : nullptr;
}

// vertvisc_coef on u, v themselves (mode 0) or on the velocity estimates the RK2 step would hand over (modes 1, 2)
static int vertvisc_coef_launch(mom6x_ctx *c, int mode, const double *u, const double *v, const double *u_bc, const double *v_bc,
                                const double *u_abt, const double *v_abt, const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx,
                                const double *h, double dt, double *u_out, double *v_out);

int vertvisc_coef_upd(mom6x_ctx *c, int mode, const double *u, const double *v, const double *u_bc, const double *v_bc,
                      const double *u_abt, const double *v_abt, double dtx, const double *h, double dt, double *u_out, double *v_out) {
  LayerAccelSrc none;
  memset(&none, 0, sizeof(none));
  return vertvisc_coef_launch(c, mode, u, v, u_bc, v_bc, u_abt, v_abt, none, none, dtx, h, dt, u_out, v_out);
}

// mode 2 with the barotropic accelerations of the layers formed on the fly (LayerAccelSrc, mom6x_dev.h)
int vertvisc_coef_upd_la(mom6x_ctx *c, const double *u, const double *v, const double *u_bc, const double *v_bc,
                         const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx, const double *h, double dt, double *u_out,
                         double *v_out) {
  return vertvisc_coef_launch(c, 3, u, v, u_bc, v_bc, nullptr, nullptr, LAu, LAv, dtx, h, dt, u_out, v_out);
}

static int vertvisc_coef_launch(mom6x_ctx *c, int mode, const double *u, const double *v, const double *u_bc, const double *v_bc,
                                const double *u_abt, const double *v_abt, const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx,
                                const double *h, double dt, double *u_out, double *v_out) {
  REQUIRE(c && c->vv_init, MOM6X_EINVAL, "MOM_vert_friction(coef): Module must be initialized before it is used.");
  REQUIRE(u && v && h, MOM6X_EINVAL, "vertvisc_coef: null array");
  REQUIRE(!c->vv.bottomdraglaw || (c->Kv_bbl_u && c->Kv_bbl_v && c->bbl_thick_u && c->bbl_thick_v), MOM6X_EINVAL,
          "vertvisc_coef: BOTTOMDRAGLAW needs visc%Kv_bbl_u/v and visc%bbl_thick_u/v (mom6x_vertvisc_set_visc)");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b = blk2();
  const mom6x_vgrid &GV = c->GV;
  const double a_cpl_max = 1.0e37 * GV.Z_to_H;
  const double I_amax = (c->vv.answer_date < 20190101) ? (1.0e-10 * GV.H_to_Z) * dt : 0.0;
  const dim3 gu = grid3(nxa(d.ni + 1, -1), d.nj, 1, b), gv = grid3(d.ni, d.nj + 1, 1, b);
#define VVC(M)                                                                                                                  \
  KLAUNCH(c, "k_vertvisc_coef<0>", (k_vertvisc_coef<0, M>), gu, b, d, c->G, c->vv, u, u_out, u_bc, u_abt, dtx, h, c->Kv_bbl_u, c->bbl_thick_u, \
          c->Kv_shear, c->vv_a_u, c->vv_h_u, GV.H_to_Z, GV.H_subroundoff, GV.dZ_subroundoff, a_cpl_max, I_amax, LAu);           \
  KLAUNCH(c, "k_vertvisc_coef<1>", (k_vertvisc_coef<1, M>), gv, b, d, c->G, c->vv, v, v_out, v_bc, v_abt, dtx, h, c->Kv_bbl_v, c->bbl_thick_v, \
          c->Kv_shear, c->vv_a_v, c->vv_h_v, GV.H_to_Z, GV.H_subroundoff, GV.dZ_subroundoff, a_cpl_max, I_amax, LAv)
  if (mode == 0) { VVC(0); } else if (mode == 1) { VVC(1); } else if (mode == 2) { VVC(2); } else { VVC(3); }
#undef VVC
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// vertvisc_coef_upd_la + vertvisc_fused of the RK2 step as one kernel per direction (k_vertvisc_coef_cols) where it exists: 75 layers,
// no Rayleigh drag, no direct stress, no KV_ML_INVZ2.  MOM6X_VERTVISC=walk|pair: the two kernels.
bool vertvisc_coef_solve_usable(mom6x_ctx *c) {
  static const int mode = [] { const char *e = getenv("MOM6X_VERTVISC"); return (e && (!strcmp(e, "walk") || !strcmp(e, "pair"))) ? 0 : 1; }();
  return mode && c->vv_init && c->d.nk <= COEF_COLS_NK_BOUND && !c->Ray_u && !(direct_stress_of(c).Hmix > 0.0) && !(c->vv.Kvml_invZ2 > 0.0) &&
         c->a_u == c->vv_a_u && c->a_v == c->vv_a_v && c->h_u == c->vv_h_u && c->h_v == c->vv_h_v &&   // (the solve reads what vertvisc_coef writes)
         (!c->vv.bottomdraglaw || (c->Kv_bbl_u && c->Kv_bbl_v && c->bbl_thick_u && c->bbl_thick_v));
}
static int vertvisc_coef_cols_launch(mom6x_ctx *c, int mode, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc,
                                     const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx, const double *h, double dt_coef,
                                     double *u, double *v, const double *taux, const double *tauy, double dt, double *taux_bot, double *tauy_bot,
                                     double *vr_u, double *vr_v, bool keep_coef) {
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const mom6x_vgrid &GV = c->GV;
  const double a_cpl_max = 1.0e37 * GV.Z_to_H;
  const double I_amax = (c->vv.answer_date < 20190101) ? (1.0e-10 * GV.H_to_Z) * dt_coef : 0.0;
  const double dt_Rho0 = dt / GV.H_to_RZ, HR = GV.H_to_RZ;
  const dim3 bc(64, 1, 1);
  const dim3 gu((unsigned)((nxa(d.ni + 1, -1) + 63) / 64), (unsigned)d.nj, 1), gv((unsigned)((d.ni + 63) / 64), (unsigned)(d.nj + 1), 1);
  const bool rem = (vr_u != nullptr);
#define VCS(DIR, M, R, WC, NKT, g, uin, ubc, LA, uo, vro, tau, taub, Kb, bt, ao, ho)                                                  \
  KLAUNCH_LDS(c, DIR ? "k_vertvisc_coef_cols<1>" : "k_vertvisc_coef_cols<0>", (k_vertvisc_coef_cols<DIR, M, R, WC, NKT>), g, bc,       \
              (size_t)NK_OF(NKT) * 64 * sizeof(double), d, c->G, c->vv, uin, ubc,                                                       \
              dtx, h, Kb, bt, c->Kv_shear, ao, ho, GV.H_to_Z, GV.H_subroundoff, GV.dZ_subroundoff, a_cpl_max, I_amax, LA, uo, vro, tau, dt, dt_Rho0, HR, taub)
#define VCS2(M, R, WC, NKT) do { \
    VCS(0, M, R, WC, NKT, gu, u_in, u_bc, LAu, u, vr_u, taux, taux_bot, c->Kv_bbl_u, c->bbl_thick_u, c->vv_a_u, c->vv_h_u); \
    VCS(1, M, R, WC, NKT, gv, v_in, v_bc, LAv, v, vr_v, tauy, tauy_bot, c->Kv_bbl_v, c->bbl_thick_v, c->vv_a_v, c->vv_h_v); } while (0)
#define VCS_ALL(NKT) do {                                                                                                               \
    if (mode == 1) VCS2(1, true, false, NKT);   /* (nobody sees the coefficients of :602-609: :737 replaces them) */                    \
    else if (rem && keep_coef) VCS2(3, true, true, NKT);                                                                                \
    else if (rem) VCS2(3, true, false, NKT);                                                                                            \
    else if (keep_coef) VCS2(3, false, true, NKT);                                                                                      \
    else VCS2(3, false, false, NKT); } while (0)
  COEF_NK_DISPATCH(d.nk, VCS_ALL);
#undef VCS_ALL
#undef VCS2
#undef VCS
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
// :737-767 / :1002-1022: vertvisc_coef on the velocity estimate + vertvisc [+ vertvisc_remnant]; keep_coef: CS%a_u, CS%h_u are written
// (the step's LAST vertvisc_coef, or a vertvisc_remnant of its own follows)
int vertvisc_coef_solve_la(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc,
                           const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx, const double *h, double dt_coef,
                           double *u, double *v, const double *taux, const double *tauy, double dt, double *taux_bot, double *tauy_bot,
                           double *vr_u, double *vr_v, bool keep_coef) {
  REQUIRE(c && vertvisc_coef_solve_usable(c), MOM6X_EINVAL, "vertvisc_coef_solve: not usable in this configuration");
  REQUIRE(u_in && v_in && u_bc && v_bc && h && u && v && taux && tauy, MOM6X_EINVAL, "vertvisc_coef_solve: null array");
  REQUIRE((vr_u != nullptr) == (vr_v != nullptr), MOM6X_EINVAL, "vertvisc_coef_solve: visc_rem_u and visc_rem_v come together");
  return vertvisc_coef_cols_launch(c, 3, u_in, v_in, u_bc, v_bc, LAu, LAv, dtx, h, dt_coef, u, v, taux, tauy, dt, taux_bot, tauy_bot, vr_u, vr_v, keep_coef);
}
// :591-610: vertvisc_coef on mask * (u + dt * u_bc_accel) + vertvisc_remnant
int vertvisc_coef_remnant(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc, double dtx,
                          const double *h, double dt_coef, double *vr_u, double *vr_v, double dt, bool keep_coef) {
  REQUIRE(c && vertvisc_coef_solve_usable(c), MOM6X_EINVAL, "vertvisc_coef_remnant: not usable in this configuration");
  REQUIRE(!keep_coef, MOM6X_EUNSUPPORTED, "vertvisc_coef_remnant: the coefficients of this stage are not kept (call vertvisc_coef and vertvisc_remnant)");
  REQUIRE(u_in && v_in && u_bc && v_bc && h && vr_u && vr_v, MOM6X_EINVAL, "vertvisc_coef_remnant: null array");
  LayerAccelSrc none;
  memset(&none, 0, sizeof(none));
  return vertvisc_coef_cols_launch(c, 1, u_in, v_in, u_bc, v_bc, none, none, dtx, h, dt_coef, nullptr, nullptr, nullptr, nullptr, dt, nullptr, nullptr,
                                   vr_u, vr_v, keep_coef);
}

extern "C" int mom6x_vertvisc_coef(mom6x_ctx *c, const double *u, const double *v, const double *h, double dt) {
  return vertvisc_coef_upd(c, 0, u, v, nullptr, nullptr, nullptr, nullptr, 0.0, h, dt, nullptr, nullptr);
}

extern "C" int mom6x_vertvisc_set_coef(mom6x_ctx *c, const double *a_u, const double *a_v, const double *h_u,
                                       const double *h_v, const double *Ray_u, const double *Ray_v) {
  REQUIRE(c && a_u && a_v && h_u && h_v, MOM6X_EINVAL, "mom6x_vertvisc_set_coef: null mandatory array");
  REQUIRE((Ray_u != nullptr) == (Ray_v != nullptr), MOM6X_EINVAL, "mom6x_vertvisc_set_coef: Ray_u and Ray_v come together");
  c->a_u = a_u; c->a_v = a_v; c->h_u = h_u; c->h_v = h_v; c->Ray_u = Ray_u; c->Ray_v = Ray_v;
  return MOM6X_OK;
}

extern "C" int mom6x_vertvisc(mom6x_ctx *c, double *u, double *v, const double *taux, const double *tauy, double dt,
                              double *taux_bot, double *tauy_bot) {
  REQUIRE(c && c->a_u, MOM6X_EINVAL, "MOM_vert_friction(visc): Module must be initialized before it is used.");
  REQUIRE(u && v && taux && tauy, MOM6X_EINVAL, "vertvisc: null array");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *c1;
  int rc;
  if ((rc = ctx_scratch(c, SCR_c1, d.nk, &c1))) return rc;
  const dim3 b = blk2();
  const double dt_Rho0 = dt / c->GV.H_to_RZ;
  KLAUNCH(c, "k_vertvisc<0>", k_vertvisc<0>, grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, u, c->a_u, c->h_u, c->Ray_u, taux, c1, dt,
          dt_Rho0, c->GV.H_to_RZ, taux_bot, direct_stress_of(c));
  KLAUNCH(c, "k_vertvisc<1>", k_vertvisc<1>, grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, v, c->a_v, c->h_v, c->Ray_v, tauy, c1, dt,
          dt_Rho0, c->GV.H_to_RZ, tauy_bot, direct_stress_of(c));
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

extern "C" int mom6x_vertvisc_set_direct_stress(mom6x_ctx *c, double Hmix_stress, const double *h) {
  REQUIRE(c, MOM6X_EINVAL, "vertvisc_set_direct_stress: null context");
  REQUIRE(!(Hmix_stress > 0.0) || h, MOM6X_EINVAL, "vertvisc: DIRECT_STRESS needs the layer thicknesses");
  c->ds_Hmix = (Hmix_stress > 0.0) ? Hmix_stress : 0.0;
  c->ds_h = (Hmix_stress > 0.0) ? h : nullptr;
  return MOM6X_OK;
}

extern "C" int mom6x_vertvisc_remnant(mom6x_ctx *c, double *visc_rem_u, double *visc_rem_v, double dt) {
  REQUIRE(c && c->a_u, MOM6X_EINVAL, "MOM_vert_friction(remant): Module must be initialized before it is used.");
  REQUIRE(visc_rem_u && visc_rem_v, MOM6X_EINVAL, "vertvisc_remnant: null array");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  double *c1;
  int rc;
  if (vertvisc_cols_usable(d.nk, c->Ray_u, DirectStress{}) && !c->Ray_v) {
    const dim3 bc(64, 1, 1);
#define VRC(NKT) do {                                                                                                                   \
    KLAUNCH(c, "k_vertvisc_remnant_cols<0>", (k_vertvisc_remnant_cols<0, NKT>), dim3((unsigned)((nxa(d.ni + 1, -1) + 63) / 64), (unsigned)d.nj, 1), bc, \
            d, c->G, visc_rem_u, c->a_u, c->h_u, dt);                                                                                    \
    KLAUNCH(c, "k_vertvisc_remnant_cols<1>", (k_vertvisc_remnant_cols<1, NKT>), dim3((unsigned)((d.ni + 63) / 64), (unsigned)(d.nj + 1), 1), bc, \
            d, c->G, visc_rem_v, c->a_v, c->h_v, dt); } while (0)
    VV_NK_DISPATCH(d.nk, VRC);
#undef VRC
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  if ((rc = ctx_scratch(c, SCR_c1, d.nk, &c1))) return rc;
  const dim3 b = blk2();
  KLAUNCH(c, "k_vertvisc_remnant<0>", k_vertvisc_remnant<0>, grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, visc_rem_u, c->a_u, c->h_u,
          c->Ray_u, c1, dt);
  KLAUNCH(c, "k_vertvisc_remnant<1>", k_vertvisc_remnant<1>, grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, visc_rem_v, c->a_v, c->h_v,
          c->Ray_v, c1, dt);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}
