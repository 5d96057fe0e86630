// barotropic.hip -- the split-explicit barotropic solver (btstep) on gfx950.
//
// Replaces btstep / btstep_timeloop / btloop_* / btcalc / bt_mass_source / set_dtbt and the static
// part of barotropic_init (MOM_barotropic.F90), on the default-flag path of SURVEY.md 8(b.1)
// (USE_BT_CONT_TYPE=T, LINEARIZED_BT_CORIOLIS=T, BT_NONLIN_STRESS=F, no OBC/SAL/filters,
// BT_USE_WIDE_HALOS=T with BTHALO=0).
//
// Structure on the device:
//  * k_bt_col<DIR>: ONE pass over the 3-D inputs per direction.  One thread per face column walks k
//    in the reference's order and produces every 3-D -> 2-D reduction btstep needs (wt_u is never
//    stored): ubt_Cor, gtot_E/W, uhbt(uh0), ubt(u_uh0), ubt, BT_force, av_rem -> bt_rem.
//  * small 2-D kernels for the Coriolis weights f_4_u/v, Cor_ref, the BT_cont cubic fits
//    (set_local_BT_cont_types) as SoA planes, eta_src.
//  * the sub-cycle: 3 kernels per barotropic step (first velocity component, second component,
//    eta corrector fused with the next step's eta predictor).  The pressure force is evaluated inside
//    the velocity kernels; transports, running means and weighted sums are accumulated in the same
//    kernels, so each 2-D coefficient plane is read exactly once per sub-step.
//  * k_layer_accel: btstep_layer_accel (3-D write of accel_layer_u/v).
// All 2-D work planes live in one HBM block that is cleared with a single memset per call.
#include <algorithm>
#include <type_traits>
#include <vector>
#include <cmath>
#include "mom6x_dev.h"

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);  // halo.hip

enum BTW {   // 2-D work planes
  W_q = 0, W_DCor_u, W_DCor_v, W_gtot_E, W_gtot_W, W_gtot_N, W_gtot_S, W_eta, W_eta_PF,
  W_Cor_ref_u, W_Cor_ref_v, W_BT_force_u, W_BT_force_v, W_ubt, W_vbt, W_bt_rem_u, W_bt_rem_v,
  W_uhbt0, W_vhbt0, W_ubt_Cor, W_vbt_Cor, W_uhbt, W_vhbt, W_u_accel_bt, W_v_accel_bt,
  W_eta_src, W_e_anom, W_eta_sum, W_eta_wtd, W_ubt_wtd, W_vbt_wtd, W_eta_pred,
  W_uh0sum, W_vh0sum, W_ubt0, W_vbt0,
  W_f4u, W_f4u_2, W_f4u_3, W_f4u_4, W_f4v, W_f4v_2, W_f4v_3, W_f4v_4,
  W_BTCu,   // 10 planes
  W_BTCv = W_BTCu + 10,   // 10 planes
  W_BTtmp = W_BTCv + 10,  // 12 planes: halo-updated copies of the BT_cont arrays
  W_uhn = W_BTtmp + 12,   // find_uhbt(ubt) + uhbt0 / find_vhbt(vbt) + vhbt0 at the velocities of the last update: what the next
  W_vhn,                  //   sub-step's eta predictor needs, formed by the kernel that has the velocity and its fit planes at hand
  W_ubt2, W_vbt2, W_eta_pred2,   // k_bt_substep: the second copies of ubt, vbt, eta_pred (a sub-step reads one set and writes the other)
  W_COUNT
};

enum BTC { B_FA_EE = 0, B_FA_E0, B_FA_W0, B_FA_WW, B_uBT_WW, B_uBT_EE, B_crvW, B_crvE, B_uh_WW, B_uh_EE };

struct BTState {
  // persistent barotropic_CS members
  double *frhatu, *frhatv, *IDatu, *IDatv, *ubtav, *vbtav, *eta_cor, *q_D, *D_u_Cor, *D_v_Cor;
  double *work;   // W_COUNT planes
  int nstep_last;
  // btstep's "eta has dropped below bathyT" warnings (:2738-2745): [0] count, [1] claimed-by-the-first flag on the
  // device; the first offender's eta, bathyT, i, j next to them
  unsigned long long *warn;
  double *warn_info;
  // btstep_layer_accel deferred to the consumer (LayerAccelSrc, mom6x_dev.h)
  bool la_defer, la_pending;
  const double *la_pbce;
  double la_underflow;
  // btcalc deferred to btstep's column pass (inside the RK2 step): frhatu / frhatv = hf * (mask / sum(hf)) are formed there from
  // the face thicknesses and never written; fr_pending: frhatu / frhatv are NOT current, fr_hu / fr_hv are their source
  bool fr_defer, fr_pending;
  const double *fr_hu, *fr_hv;
};

namespace {

__device__ __forceinline__ double *wp(double *work, const Dm &d, int m) { return work + (size_t)m * d.slab; }

// find_uhbt :4610-4629 / find_vhbt :4744 on SoA fit planes
__device__ __forceinline__ double find_uhbt(double u, const double *__restrict__ B, size_t c, size_t slab) {
  if (u == 0.0) return 0.0;
  const double uEE = B[B_uBT_EE * slab + c];
  if (u < uEE) return (u - uEE) * B[B_FA_EE * slab + c] + B[B_uh_EE * slab + c];
  if (u < 0.0) return u * (B[B_FA_E0 * slab + c] + B[B_crvE * slab + c] * (u * u));
  const double uWW = B[B_uBT_WW * slab + c];
  if (u <= uWW) return u * (B[B_FA_W0 * slab + c] + B[B_crvW * slab + c] * (u * u));
  return (u - uWW) * B[B_FA_WW * slab + c] + B[B_uh_WW * slab + c];
}

// ---- barotropic_init static fields :5865-5896, :6146-6163 ---------------------------------
__global__ void k_bt_init_static(Dm d, const double *__restrict__ G, double Z_to_H, double Mean_SL,
                                 double cor_scale, double H_subroundoff, double *q_D, double *D_u_Cor,
                                 double *D_v_Cor, double *IDatu, double *IDatv) {
  const int i = -1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j);
  const double *bathyT = gm(G, d, MOM6X_G_bathyT), *areaT = gm(G, d, MOM6X_G_areaT), *mT = gm(G, d, MOM6X_G_mask2dT);
  const double *mCu = gm(G, d, MOM6X_G_mask2dCu), *mCv = gm(G, d, MOM6X_G_mask2dCv), *fBu = gm(G, d, MOM6X_G_CoriolisBu);
  if (j >= 0) {
    D_u_Cor[c] = 0.5 * (dmax(Mean_SL + bathyT[c + 1], 0.0) + dmax(Mean_SL + bathyT[c], 0.0)) * Z_to_H;
    if (mCu[c] > 0.) IDatu[c] = mCu[c] * 2.0 / (Z_to_H * ((bathyT[c + 1] + bathyT[c]) + 2.0 * Mean_SL));
    else IDatu[c] = 0.;
  }
  if (i >= 0) {
    D_v_Cor[c] = 0.5 * (dmax(Mean_SL + bathyT[c + st], 0.0) + dmax(Mean_SL + bathyT[c], 0.0)) * Z_to_H;
    if (mCv[c] > 0.) IDatv[c] = mCv[c] * 2.0 / (Z_to_H * ((bathyT[c + st] + bathyT[c]) + 2.0 * Mean_SL));
    else IDatv[c] = 0.;
  }
  if (mT[c] + mT[c + st] + mT[c + 1] + mT[c + 1 + st] > 0.) {
    q_D[c] = 0.25 * (cor_scale * fBu[c]) * ((areaT[c] + areaT[c + 1 + st]) + (areaT[c + 1] + areaT[c + st])) /
        (Z_to_H * dmax((((areaT[c] * dmax(Mean_SL + bathyT[c], 0.0)) +
                         (areaT[c + 1 + st] * dmax(Mean_SL + bathyT[c + 1 + st], 0.0))) +
                        ((areaT[c + 1] * dmax(Mean_SL + bathyT[c + 1], 0.0)) +
                         (areaT[c + st] * dmax(Mean_SL + bathyT[c + st], 0.0)))), H_subroundoff));
  } else {
    q_D[c] = 0.;
  }
}

// ---- btcalc :4360-4605 --------------------------------------------------------------------
// btcalc :4360 with BT_THICK_SCHEME = FROM_BT_CONT at nk = NK: the column of h_u is read ONCE into registers (NK
// independent loads in flight), summed in the reference's order and written back scaled -- 2 words per face-layer
// instead of 3.
template <int DIR, int NKT>   // (NKT: mom6x_dev.h NK_OF / NK_EXACT -- the layer count itself, or a bound on it)
__global__ void __launch_bounds__(256)
k_btcalc_cols(Dm d, const double *__restrict__ G, const double *__restrict__ hf, double *__restrict__ fr, double h_neglect) {
  constexpr int NK = NK_OF(NKT);
  const int nk = NK_EXACT(NKT) ? NK : d.nk;
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[c];
  double v[NK];
#pragma unroll
  for (int k = 0; k < NK; k++) if (k < nk) v[k] = hf[c + (size_t)k * slab];
  double hattot = 0.0;
#pragma unroll
  for (int k = 0; k < NK; k++) if (k < nk) hattot = hattot + v[k];
  const double Ihattot = mC / (hattot + h_neglect);
#pragma unroll
  for (int k = 0; k < NK; k++) if (k < nk) fr[c + (size_t)k * slab] = v[k] * Ihattot;
}

template <int DIR>
__global__ void __launch_bounds__(256)
k_btcalc(Dm d, const double *__restrict__ G, const double *__restrict__ h, const double *__restrict__ hf,
         double *__restrict__ fr, double h_neglect, double Z_to_H, int scheme) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int st = DIR ? d.pitch : 1, nz = d.nk;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[c];
  double hattot = 0.0;
  if (hf) {
    for (int k = 0; k < nz; k++) hattot = hattot + hf[c + k * slab];
    const double Ihattot = mC / (hattot + h_neglect);
    for (int k = 0; k < nz; k++) fr[c + k * slab] = hf[c + k * slab] * Ihattot;
  } else if (scheme == MOM6X_BT_THICK_ARITHMETIC || scheme == MOM6X_BT_THICK_HARMONIC) {   // :4448-4452, :4476-4483
    for (int k = 0; k < nz; k++) {
      const double hp = h[c + st + k * slab], hm = h[c + k * slab];
      const double hat = (scheme == MOM6X_BT_THICK_ARITHMETIC) ? 0.5 * (hp + hm) : 2.0 * (hp * hm) / ((hp + hm) + h_neglect);
      fr[c + k * slab] = hat;
      hattot = hattot + hat;
    }
    const double Ihattot = mC / (hattot + h_neglect);
    for (int k = 0; k < nz; k++) fr[c + k * slab] = fr[c + k * slab] * Ihattot;
  } else {   // HYBRID (or may_use_default) :4453-4475; hat is staged in fr and rescaled afterwards
    const double *bathyT = gm(G, d, MOM6X_G_bathyT);
    double e_below = -0.5 * Z_to_H * (bathyT[c + st] + bathyT[c]);
    const double D_shallow = -Z_to_H * dmin(bathyT[c + st], bathyT[c]);
    for (int k = nz - 1; k >= 0; k--) {
      const double hp = h[c + st + k * slab], hm = h[c + k * slab];
      const double e_k = e_below + 0.5 * (hp + hm);
      const double h_arith = 0.5 * (hp + hm);
      double hat;
      if (e_below >= D_shallow) {
        hat = h_arith;
      } else {
        const double h_harm = (hp * hm) / (h_arith + h_neglect);
        if (e_k <= D_shallow) hat = h_harm;
        else {
          const double wt_arith = (e_k - D_shallow) / (h_arith + h_neglect);
          hat = wt_arith * h_arith + (1.0 - wt_arith) * h_harm;
        }
      }
      fr[c + k * slab] = hat;
      hattot = hattot + hat;
      e_below = e_k;
    }
    const double Ihattot = mC / (hattot + h_neglect);
    for (int k = 0; k < nz; k++) fr[c + k * slab] = fr[c + k * slab] * Ihattot;
  }
}

// ---- bt_mass_source :5243-5296 ---------------------------------------------------------------
__global__ void k_bt_mass_source(Dm d, const double *__restrict__ G, const double *__restrict__ h,
                                 const double *__restrict__ eta, double *eta_cor, int set_cor, double Z_to_H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t c = ix2(d, i, j);
  double eta_h = h[c] - gm(G, d, MOM6X_G_bathyT)[c] * Z_to_H;
  for (int k = 1; k < d.nk; k++) eta_h = eta_h + h[c + (size_t)k * d.slab];
  const double d_eta = eta_h - eta[c];
  eta_cor[c] = set_cor ? d_eta : (eta_cor[c] + d_eta);
}

// ... with eta_h formed by the kernel that last walked the same h top-down (k_pgf_main inside the RK2 step)
__global__ void k_bt_mass_source_from(Dm d, const double *__restrict__ eta_h, const double *__restrict__ eta, double *eta_cor, int set_cor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t c = ix2(d, i, j);
  const double d_eta = eta_h[c] - eta[c];
  eta_cor[c] = set_cor ? d_eta : (eta_cor[c] + d_eta);
}

// ---- set_dtbt :3509-3633 (find_face_areas add_max branch :5208-5219) -------------------------
__global__ void k_set_dtbt(Dm d, const double *__restrict__ G, const double *__restrict__ pbce,
                           const double *__restrict__ frhatu, const double *__restrict__ frhatv,
                           double gtot_est, double Z_to_H, double zadd, int add_max, double bebt, double cor_scale2,
                           double *Idt_max2_out, const double *__restrict__ eta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *bathyT = gm(G, d, MOM6X_G_bathyT), *dy_Cu = gm(G, d, MOM6X_G_dy_Cu), *dx_Cv = gm(G, d, MOM6X_G_dx_Cv);
  const double *IdxCu = gm(G, d, MOM6X_G_IdxCu), *IdyCv = gm(G, d, MOM6X_G_IdyCv), *f2 = gm(G, d, MOM6X_G_Coriolis2Bu);
  double DatuE, DatuW, DatvN, DatvS;
  if (add_max) {   // find_face_areas(add_max=) :5208-5219
    DatuE = dy_Cu[c] * Z_to_H * dmax(dmax(bathyT[c + 1], bathyT[c]) + zadd, 0.0);
    DatuW = dy_Cu[c - 1] * Z_to_H * dmax(dmax(bathyT[c], bathyT[c - 1]) + zadd, 0.0);
    DatvN = dx_Cv[c] * Z_to_H * dmax(dmax(bathyT[c + st], bathyT[c]) + zadd, 0.0);
    DatvS = dx_Cv[c - st] * Z_to_H * dmax(dmax(bathyT[c], bathyT[c - st]) + zadd, 0.0);
  } else {         // find_face_areas without eta / add_max :5221-5236 (zadd = G%Z_ref); with eta (NONLINEAR_BT_CONTINUITY) :5171-5186
    auto Hc = [&](size_t x) { return eta ? (bathyT[x] * Z_to_H + eta[x]) : ((bathyT[x] + zadd) * Z_to_H); };
    const double H0 = Hc(c), HE = Hc(c + 1), HW = Hc(c - 1);
    const double HN = Hc(c + st), HS = Hc(c - st);
    DatuE = 0.0; if ((H0 > 0.0) && (HE > 0.0)) DatuE = dy_Cu[c] * (2.0 * H0 * HE) / (H0 + HE);
    DatuW = 0.0; if ((HW > 0.0) && (H0 > 0.0)) DatuW = dy_Cu[c - 1] * (2.0 * HW * H0) / (HW + H0);
    DatvN = 0.0; if ((H0 > 0.0) && (HN > 0.0)) DatvN = dx_Cv[c] * (2.0 * H0 * HN) / (H0 + HN);
    DatvS = 0.0; if ((HS > 0.0) && (H0 > 0.0)) DatvS = dx_Cv[c - st] * (2.0 * HS * H0) / (HS + H0);
  }
  double gE = gtot_est, gW = gtot_est, gN = gtot_est, gS = gtot_est;
  if (pbce) {
    gE = gW = gN = gS = 0.0;
    for (int k = 0; k < d.nk; k++) {
      const double pb = pbce[c + k * slab];
      gE = gE + pb * frhatu[c + k * slab];
      gW = gW + pb * frhatu[c - 1 + k * slab];
      gN = gN + pb * frhatv[c + k * slab];
      gS = gS + pb * frhatv[c - st + k * slab];
    }
  }
  Idt_max2_out[c] = 0.5 * (1.0 + 2.0 * bebt) * (gm(G, d, MOM6X_G_IareaT)[c] *
      (((gE * DatuE * IdxCu[c]) + (gW * DatuW * IdxCu[c - 1])) + ((gN * DatvN * IdyCv[c]) + (gS * DatvS * IdyCv[c - st]))) +
      ((f2[c] + f2[c - 1 - st]) + (f2[c - 1] + f2[c - st])) * cor_scale2);
}

// ---- the single 3-D -> 2-D pass of btstep :1011-1330, :1473-1509 -----------------------------
struct ColArgs {
  const double *visc_rem, *frhat, *U_Cor, *pbce, *uh0, *u_uh0, *U_in, *bc_accel, *tau, *tau_bot, *IDat;
  const double *hf; double h_neglect;   // HF: frhat is not stored -- btcalc's expressions on the face thicknesses hf instead
  double *ubt_Cor, *gtot_m /*E|N at cell c*/, *gtot_p /*W|S at cell c+st*/, *uh0sum, *ubt0, *ubt, *BT_force, *bt_rem;
  double Instep, RZ_to_H, vel_underflow;
  int nstep, wt_uv_bug, visc_rem_u_uh0, strong_drag;
};

// HF: btcalc (:4360, BT_THICK_SCHEME = FROM_BT_CONT) folded in -- a first sweep sums the face thicknesses of the column, and
// frhat(k) = hf(k) * (mask / (sum(hf) + h_neglect)) is formed where it is used (the same two operations on the same operands) and
// never goes to memory: btcalc's two launches per btstep are gone, and the sweeps' re-reads of hf come from the cache hierarchy as
// those of visc_rem do.  (The column in registers, as k_btcalc_cols has it, spills here: 75 more doubles next to 7 input streams.)
template <int DIR, bool HF>
__global__ void __launch_bounds__(256)
k_bt_col(Dm d, const double *__restrict__ G, ColArgs A) {
  const int i = I_BASE((DIR ? 0 : -1)) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (DIR ? -1 : 0) + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < ((DIR ? 0 : -1))) return;
  const int st = DIR ? d.pitch : 1, nz = d.nk;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double subroundoff = 1e-30;
  const double mC = gm(G, d, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[c];
  double Ihattot = 0.0;
  if (HF) {
    double hattot = 0.0;
    for (int k = 0; k < nz; k++) hattot = hattot + A.hf[c + (size_t)k * slab];
    Ihattot = mC / (hattot + A.h_neglect);
  }
#define FRH(k) (HF ? A.hf[c + (size_t)(k) * slab] * Ihattot : A.frhat[c + (size_t)(k) * slab])
  double Iwt_tot = 1.0;
  if (!A.wt_uv_bug) {   // :1032-1059
    double tot = 0.0;
    for (int k = 0; k < nz; k++) {
      double vr = dmin(A.visc_rem[c + k * slab], 1.);
      vr = dmax(vr, 1. - 0.5 * A.Instep / (vr + subroundoff));
      vr = dmax(vr, 0.);
      const double w = FRH(k) * vr;
      tot = (k == 0) ? w : (tot + w);
    }
    Iwt_tot = tot;
    if (fabs(tot) > 0.0) Iwt_tot = mC / tot;
  }
  double ubt_Cor = 0.0, gm_ = 0.0, gp_ = 0.0, uh0sum = 0.0, ubt0 = 0.0, ubt = 0.0, av_rem = 0.0;
  // BT_force starts from the surface (and bottom) stress term :1259-1320
  double BT_force = 0.0;
  if (mC > 0.0) {
    BT_force = A.tau[c] * A.RZ_to_H * A.IDat[c] * A.visc_rem[c];
    if (A.tau_bot) BT_force = BT_force - A.tau_bot[c] * A.RZ_to_H * A.IDat[c];
  }
  for (int k = 0; k < nz; k++) {
    const size_t f = c + k * slab;
    const double vrem = A.visc_rem[f], frh = FRH(k);
    double vr = dmin(vrem, 1.);
    vr = dmax(vr, 1. - 0.5 * A.Instep / (vr + subroundoff));
    vr = dmax(vr, 0.);
    double wt = frh * vr;
    if (!A.wt_uv_bug) wt = wt * Iwt_tot;
    ubt_Cor = ubt_Cor + wt * A.U_Cor[f];
    gm_ = gm_ + A.pbce[f] * wt;
    gp_ = gp_ + A.pbce[f + st] * wt;
    if (A.uh0) {
      uh0sum = uh0sum + A.uh0[f];
      ubt0 = ubt0 + (A.visc_rem_u_uh0 ? wt : frh) * A.u_uh0[f];
    }
    ubt = ubt + wt * A.U_in[f];
    BT_force = BT_force + wt * A.bc_accel[f];
    av_rem = av_rem + frh * vrem;
  }
#undef FRH
  A.ubt_Cor[c] = ubt_Cor;
  A.gtot_m[c] = gm_;
  A.gtot_p[c + st] = gp_;
  if (A.uh0) { A.uh0sum[c] = uh0sum; A.ubt0[c] = ubt0; }
  if (fabs(ubt) < A.vel_underflow) ubt = 0.0;
  A.ubt[c] = ubt;
  A.BT_force[c] = BT_force;
  double bt_rem;
  if (A.strong_drag) {
    bt_rem = mC * ((A.nstep * av_rem) / (1.0 + (A.nstep - 1) * av_rem));
  } else {
    bt_rem = 0.0;
    if (mC * av_rem > 0.0) bt_rem = mC * pow(av_rem, A.Instep);
  }
  A.bt_rem[c] = bt_rem;
}

// ---- set_local_BT_cont_types :4876-5003 -------------------------------------------------------
// step 1: copy the computational-domain values of the 12 BT_cont planes into zeroed temporaries
__global__ void k_btcont_copy(Dm d, mom6x_BT_cont BT, double *tmp) {
  const int i = -1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  if (j >= 0) {
    tmp[0 * slab + c] = BT.uBT_EE[c]; tmp[1 * slab + c] = BT.uBT_WW[c]; tmp[2 * slab + c] = BT.FA_u_EE[c];
    tmp[3 * slab + c] = BT.FA_u_E0[c]; tmp[4 * slab + c] = BT.FA_u_W0[c]; tmp[5 * slab + c] = BT.FA_u_WW[c];
  }
  if (i >= 0) {
    tmp[6 * slab + c] = BT.vBT_NN[c]; tmp[7 * slab + c] = BT.vBT_SS[c]; tmp[8 * slab + c] = BT.FA_v_NN[c];
    tmp[9 * slab + c] = BT.FA_v_N0[c]; tmp[10 * slab + c] = BT.FA_v_S0[c]; tmp[11 * slab + c] = BT.FA_v_SS[c];
  }
}
// USE_BT_CONT_TYPE = False: the transports are Datu * ubt (+ uhbt0) with the face areas of find_face_areas :5146-5237 (its last
// branch: NONLINEAR_BT_CONTINUITY = False, from the bathymetry; halo = 1, :1135).  They go through the SAME fit planes: with all four
// face areas of a fit = Datu, no curvature and break points at -/+ 1e100, find_uhbt(u) is u * (Datu + 0 * (u * u)) = Datu * u in
// every bit (0 * u^2 is 0, Datu + 0 is Datu, the product commutes).  The halo update that follows is the reference's pass_Dat_uv.
// eta != NULL: NONLINEAR_BT_CONTINUITY, the Boussinesq branch :5171-5186 (H = bathyT * Z_to_H + eta).
__global__ void k_face_areas_as_fits(Dm d, const double *__restrict__ G, double Z_ref, double Z_to_H, double *tmp, const double *__restrict__ eta) {
  const int i = -2 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni || j > d.nj) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *bathyT = gm(G, d, MOM6X_G_bathyT);
  const double H1 = eta ? (bathyT[c] * Z_to_H + eta[c]) : ((bathyT[c] + Z_ref) * Z_to_H);
  if (j >= -1) {   // Datu on (is-2 .. ie+1, js-1 .. je+1)
    const double H2 = eta ? (bathyT[c + 1] * Z_to_H + eta[c + 1]) : ((bathyT[c + 1] + Z_ref) * Z_to_H);
    double Dat = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Dat = gm(G, d, MOM6X_G_dy_Cu)[c] * (2.0 * H1 * H2) / (H1 + H2);
    tmp[0 * slab + c] = -1.0e100; tmp[1 * slab + c] = 1.0e100;
    tmp[2 * slab + c] = Dat; tmp[3 * slab + c] = Dat; tmp[4 * slab + c] = Dat; tmp[5 * slab + c] = Dat;
  }
  if (i >= -1) {   // Datv on (is-1 .. ie+1, js-2 .. je+1)
    const double H2 = eta ? (bathyT[c + st] * Z_to_H + eta[c + st]) : ((bathyT[c + st] + Z_ref) * Z_to_H);
    double Dat = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Dat = gm(G, d, MOM6X_G_dx_Cv)[c] * (2.0 * H1 * H2) / (H1 + H2);
    tmp[6 * slab + c] = -1.0e100; tmp[7 * slab + c] = 1.0e100;
    tmp[8 * slab + c] = Dat; tmp[9 * slab + c] = Dat; tmp[10 * slab + c] = Dat; tmp[11 * slab + c] = Dat;
  }
}
// ... and inside the time loop (:2539-2543, evolving_face_areas): the four face areas of every fit rewritten from the current eta on the
// sub-step's ranges (halo = 1 + iev - ie); the other parameters of the degenerate fits do not change.
__global__ void k_face_areas_eta(Dm d, const double *__restrict__ G, double Z_to_H, const double *__restrict__ eta, double *Bu, double *Bv,
                                 int isv, int iev, int jsv, int jev) {
  const int i = isv - 2 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = jsv - 2 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > iev + 1 || j > jev + 1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *bathyT = gm(G, d, MOM6X_G_bathyT);
  const double H1 = bathyT[c] * Z_to_H + eta[c];
  if (j >= jsv - 1) {
    const double H2 = bathyT[c + 1] * Z_to_H + eta[c + 1];
    double Dat = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Dat = gm(G, d, MOM6X_G_dy_Cu)[c] * (2.0 * H1 * H2) / (H1 + H2);
    Bu[B_FA_EE * slab + c] = Dat; Bu[B_FA_E0 * slab + c] = Dat; Bu[B_FA_W0 * slab + c] = Dat; Bu[B_FA_WW * slab + c] = Dat;
  }
  if (i >= isv - 1) {
    const double H2 = bathyT[c + st] * Z_to_H + eta[c + st];
    double Dat = 0.0;
    if ((H1 > 0.0) && (H2 > 0.0)) Dat = gm(G, d, MOM6X_G_dx_Cv)[c] * (2.0 * H1 * H2) / (H1 + H2);
    Bv[B_FA_EE * slab + c] = Dat; Bv[B_FA_E0 * slab + c] = Dat; Bv[B_FA_W0 * slab + c] = Dat; Bv[B_FA_WW * slab + c] = Dat;
  }
}
// step 2 (after the halo update): the cubic-fit parameters as SoA planes
__global__ void k_btcl(Dm d, const double *__restrict__ tmp, double *Bu, double *Bv, int hs) {
  const int i = -hs - 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -hs - 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + hs || j > d.nj - 1 + hs) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double C1_3 = 1.0 / 3.0;
  for (int dir = 0; dir < 2; dir++) {
    if (dir == 0 && j < -hs) continue;
    if (dir == 1 && i < -hs) continue;
    const double *t = tmp + (size_t)(dir * 6) * slab;
    double *B = dir ? Bv : Bu;
    const double FA_EE = t[2 * slab + c], FA_E0 = t[3 * slab + c], FA_W0 = t[4 * slab + c], FA_WW = t[5 * slab + c];
    const double uBT_EE = 1.0 * t[0 * slab + c], uBT_WW = 1.0 * t[1 * slab + c];
    B[B_FA_EE * slab + c] = FA_EE; B[B_FA_E0 * slab + c] = FA_E0; B[B_FA_W0 * slab + c] = FA_W0; B[B_FA_WW * slab + c] = FA_WW;
    B[B_uBT_EE * slab + c] = uBT_EE; B[B_uBT_WW * slab + c] = uBT_WW;
    B[B_uh_EE * slab + c] = uBT_EE * (C1_3 * (2.0 * FA_E0 + FA_EE));
    B[B_uh_WW * slab + c] = uBT_WW * (C1_3 * (2.0 * FA_W0 + FA_WW));
    double crvW = 0.0, crvE = 0.0;
    if (fabs(uBT_WW) > 0.0) crvW = (C1_3 * (FA_WW - FA_W0)) / (uBT_WW * uBT_WW);
    if (fabs(uBT_EE) > 0.0) crvE = (C1_3 * (FA_EE - FA_E0)) / (uBT_EE * uBT_EE);
    B[B_crvW * slab + c] = crvW; B[B_crvE * slab + c] = crvE;
  }
}

// uhbt0 = sum(uh0) - find_uhbt(ubt(u_uh0))  :1203-1209
__global__ void k_uhbt0(Dm d, const double *__restrict__ work_c, double *work) {
  const int i = -1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  if (j >= 0) work[W_uhbt0 * slab + c] = work_c[W_uh0sum * slab + c] - find_uhbt(work_c[W_ubt0 * slab + c], work_c + W_BTCu * slab, c, slab);
  if (i >= 0) work[W_vhbt0 * slab + c] = work_c[W_vh0sum * slab + c] - find_uhbt(work_c[W_vbt0 * slab + c], work_c + W_BTCv * slab, c, slab);
}

// btstep_find_Cor :2836-2894 (OBCmask == 1)
__global__ void k_find_Cor(Dm d, double *work, int Sadourny, int isvf, int ievf, int jsvf, int jevf) {
  const int i = isvf - 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = jsvf - 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > ievf + 1 || j > jevf + 1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *q = work + W_q * slab, *DCor_u = work + W_DCor_u * slab, *DCor_v = work + W_DCor_v * slab;
  double *f4u = work + W_f4u * slab, *f4v = work + W_f4v * slab;
  if (j <= jevf) {   // f_4_v on J=jsvf-1..jevf, i=isvf-1..ievf+1
    if (Sadourny) {
      f4v[0 * slab + c] = 1.0 * DCor_u[c - 1] * q[c - 1];
      f4v[1 * slab + c] = 1.0 * DCor_u[c] * q[c];
      f4v[3 * slab + c] = 1.0 * DCor_u[c + st] * q[c];
      f4v[2 * slab + c] = 1.0 * DCor_u[c - 1 + st] * q[c - 1];
    } else {
      f4v[0 * slab + c] = 1.0 * DCor_u[c - 1] * ((q[c] + q[c - 1 - st]) + q[c - 1]) / 3.0;
      f4v[1 * slab + c] = 1.0 * DCor_u[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
      f4v[3 * slab + c] = 1.0 * DCor_u[c + st] * (q[c] + (q[c - 1] + q[c + st])) / 3.0;
      f4v[2 * slab + c] = 1.0 * DCor_u[c - 1 + st] * ((q[c] + q[c - 1 + st]) + q[c - 1]) / 3.0;
    }
  }
  if (i <= ievf) {   // f_4_u on j=jsvf-1..jevf+1, I=isvf-1..ievf
    if (Sadourny) {
      f4u[3 * slab + c] = 1.0 * DCor_v[c + 1] * q[c];
      f4u[2 * slab + c] = 1.0 * DCor_v[c] * q[c];
      f4u[0 * slab + c] = 1.0 * DCor_v[c - st] * q[c - st];
      f4u[1 * slab + c] = 1.0 * DCor_v[c + 1 - st] * q[c - st];
    } else {
      f4u[3 * slab + c] = 1.0 * DCor_v[c + 1] * (q[c] + (q[c + 1] + q[c - st])) / 3.0;
      f4u[2 * slab + c] = 1.0 * DCor_v[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
      f4u[0 * slab + c] = 1.0 * DCor_v[c - st] * ((q[c] + q[c - 1 - st]) + q[c - st]) / 3.0;
      f4u[1 * slab + c] = 1.0 * DCor_v[c + 1 - st] * ((q[c] + q[c + 1 - st]) + q[c - st]) / 3.0;
    }
  }
}

// Cor_ref :1451-1461 and eta_src :1548-1587
// bound_BT_corr: 0 none; 1 the bounds from the BT_cont fits :1551-1580; 2 eta_cor_bound :1582-1585, formed here as barotropic_init
// forms it (:6164-6173: find_face_areas without eta / add_max, the harmonic means of the resting depths :5221-5236)
__global__ void k_cor_ref_eta_src(Dm d, const double *__restrict__ G, double *work, double *eta_cor, double Instep,
                                  int bound_BT_corr, double maxCFL_Idt, double dt, double Z_to_H, double Z_ref, double maxvel) {
  const int i = -1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *f4u = work + W_f4u * slab, *f4v = work + W_f4v * slab;
  const double *ubt_Cor = work + W_ubt_Cor * slab, *vbt_Cor = work + W_vbt_Cor * slab;
  if (j >= 0)
    work[W_Cor_ref_u * slab + c] = (((f4u[3 * slab + c] * vbt_Cor[c + 1]) + (f4u[0 * slab + c] * vbt_Cor[c - st])) +
                                    ((f4u[2 * slab + c] * vbt_Cor[c]) + (f4u[1 * slab + c] * vbt_Cor[c + 1 - st])));
  if (i >= 0)
    work[W_Cor_ref_v * slab + c] = -1.0 * (((f4v[0 * slab + c] * ubt_Cor[c - 1]) + (f4v[3 * slab + c] * ubt_Cor[c + st])) +
                                           ((f4v[1 * slab + c] * ubt_Cor[c]) + (f4v[2 * slab + c] * ubt_Cor[c - 1 + st])));
  if (i >= 0 && j >= 0) {
    const double mT = gm(G, d, MOM6X_G_mask2dT)[c];
    double ec = eta_cor[c];
    if (bound_BT_corr == 2) {
      const double *bathyT = gm(G, d, MOM6X_G_bathyT), *dy_Cu = gm(G, d, MOM6X_G_dy_Cu), *dx_Cv = gm(G, d, MOM6X_G_dx_Cv);
      auto Dat = [&](size_t a, size_t bb, double len) {
        const double H1 = (bathyT[a] + Z_ref) * Z_to_H, H2 = (bathyT[bb] + Z_ref) * Z_to_H;
        double D = 0.0;
        if ((H1 > 0.0) && (H2 > 0.0)) D = len * (2.0 * H1 * H2) / (H1 + H2);
        return D;
      };
      const double DatuW = Dat(c - 1, c, dy_Cu[c - 1]), DatuE = Dat(c, c + 1, dy_Cu[c]);
      const double DatvN = Dat(c, c + st, dx_Cv[c]), DatvS = Dat(c - st, c, dx_Cv[c - st]);
      const double bound = dt * (gm(G, d, MOM6X_G_IareaT)[c] * 0.1 * maxvel * ((DatuW + DatuE) + (DatvN + DatvS)));
      if (fabs(ec) > bound) ec = copysign(bound, ec);
      eta_cor[c] = ec;
    } else if (bound_BT_corr && mT > 0.0) {
      if (ec > 0.0) {
        const double u_max_cor = gm(G, d, MOM6X_G_dxT)[c] * maxCFL_Idt, v_max_cor = gm(G, d, MOM6X_G_dyT)[c] * maxCFL_Idt;
        const double *Bu = work + W_BTCu * slab, *Bv = work + W_BTCv * slab;
        const double *uhbt0 = work + W_uhbt0 * slab, *vhbt0 = work + W_vhbt0 * slab;
        const double eta_cor_max = dt * (gm(G, d, MOM6X_G_IareaT)[c] *
            (((find_uhbt(u_max_cor, Bu, c, slab) + uhbt0[c]) - (find_uhbt(-u_max_cor, Bu, c - 1, slab) + uhbt0[c - 1])) +
             ((find_uhbt(v_max_cor, Bv, c, slab) + vhbt0[c]) - (find_uhbt(-v_max_cor, Bv, c - st, slab) + vhbt0[c - st]))));
        ec = dmin(ec, dmax(0.0, eta_cor_max));
      } else {
        const double Htot = gm(G, d, MOM6X_G_bathyT)[c] * Z_to_H + work[W_eta * slab + c];
        ec = dmax(ec, -dmax(0.0, Htot));
      }
      eta_cor[c] = ec;
    }
    work[W_eta_src * slab + c] = mT * (Instep * ec);
  }
}

// copy eta_in / eta_PF_in / q_D / D_Cor into the work block :880-893, :996-1001
__global__ void k_bt_copy_in(Dm d, double *work, const double *__restrict__ eta_in, const double *__restrict__ eta_PF_in,
                             const double *__restrict__ q_D, const double *__restrict__ D_u_Cor,
                             const double *__restrict__ D_v_Cor) {
  const int i = -d.halo - 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -d.halo - 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + d.halo || j > d.nj - 1 + d.halo) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  if (i >= -d.halo && j >= -d.halo) { work[W_eta * slab + c] = eta_in[c]; work[W_eta_PF * slab + c] = eta_PF_in[c]; }
  work[W_q * slab + c] = q_D[c]; work[W_DCor_u * slab + c] = D_u_Cor[c]; work[W_DCor_v * slab + c] = D_v_Cor[c];
}

// ---- the sub-cycle ---------------------------------------------------------------------------
struct LoopArgs {
  double dtbt, dgeo_de, vel_underflow, trans_wt1, trans_wt2;
  double wt_accel, wt_trans, wt_vel, wt_eta, wt_accel2;
  int project, bracket_bug, find_etaav;
  int f4_on_the_fly, Sadourny;   // the Coriolis weights recomputed from q, D_u_Cor, D_v_Cor in the velocity kernels
  int store_uhn, have_uhn;       // the velocity kernels leave W_uhn / W_vhn for the next predictor; this predictor finds them
  int pred_next;                 // k_bt_eta also forms the NEXT sub-step's eta predictor (same points, no exchange in between)
  double wt_accel2_next;
  int isv, iev, jsv, jev;   // valid range of this step
};

// btloop_eta_predictor :2956-3018 (use_BT_cont branch) over (isv-1..iev+1, jsv-1..jev+1), plus the
// eta_sum accumulation of btloop_find_PF :3104-3108 over the computational domain.
__global__ void __launch_bounds__(256)
k_bt_pred(Dm d, const double *__restrict__ G, double *work, LoopArgs A, int p_ubt, int p_vbt, int p_pred, int sel) {
  const int i = I_BASE(A.isv - 1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = A.jsv - 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < A.isv - 1 || i > A.iev + 1 || j > A.jev + 1) return;
  // sel 1 / 2: only the tile's own points (whose inputs are all the tile's own) / only the others -- the two halves of a launch
  // around a group pass that is travelling (btstep's loop)
  if (sel) { const bool own = (i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1); if (own != (sel == 1)) return; }
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  double eta_PF_BT;
  if (A.project) {
    eta_PF_BT = work[W_eta * slab + c];
  } else {
    const double *ubt = work + (size_t)p_ubt * slab, *vbt = work + (size_t)p_vbt * slab;
    const double *Bu = work + W_BTCu * slab, *Bv = work + W_BTCv * slab;
    const double *uhbt0 = work + W_uhbt0 * slab, *vhbt0 = work + W_vhbt0 * slab;
    double uW, uE, vS, vN;
    if (A.have_uhn) {   // the same four expressions, evaluated by the last velocity update (k_bt_vel) and passed with the velocities
      const double *uhn = work + W_uhn * slab, *vhn = work + W_vhn * slab;
      uW = uhn[c - 1]; uE = uhn[c]; vS = vhn[c - st]; vN = vhn[c];
    } else {
      uW = find_uhbt(ubt[c - 1], Bu, c - 1, slab) + uhbt0[c - 1];
      uE = find_uhbt(ubt[c], Bu, c, slab) + uhbt0[c];
      vS = find_uhbt(vbt[c - st], Bv, c - st, slab) + vhbt0[c - st];
      vN = find_uhbt(vbt[c], Bv, c, slab) + vhbt0[c];
    }
    eta_PF_BT = (work[W_eta * slab + c] + work[W_eta_src * slab + c]) +
                (A.dtbt * gm(G, d, MOM6X_G_IareaT)[c]) * ((uW - uE) + (vS - vN));
    work[(size_t)p_pred * slab + c] = eta_PF_BT;
  }
  if (A.find_etaav && (fabs(A.wt_accel2) > 0.0) && i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1)
    work[W_eta_sum * slab + c] = work[W_eta_sum * slab + c] + A.wt_accel2 * eta_PF_BT;
}

// The Coriolis weights f_4_u / f_4_v of btstep_find_Cor (:2836-2894) formed where they are used, from q and D_u_Cor / D_v_Cor
// (k_find_Cor's expressions, same bits): three planes read with neighbours instead of eight planes.
__device__ __forceinline__ void f4u_of(const double *__restrict__ q, const double *__restrict__ DCor_v, size_t c, int st, int Sadourny,
                                       double &f0, double &f1, double &f2, double &f3) {
  if (Sadourny) {
    f3 = 1.0 * DCor_v[c + 1] * q[c];
    f2 = 1.0 * DCor_v[c] * q[c];
    f0 = 1.0 * DCor_v[c - st] * q[c - st];
    f1 = 1.0 * DCor_v[c + 1 - st] * q[c - st];
  } else {
    f3 = 1.0 * DCor_v[c + 1] * (q[c] + (q[c + 1] + q[c - st])) / 3.0;
    f2 = 1.0 * DCor_v[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
    f0 = 1.0 * DCor_v[c - st] * ((q[c] + q[c - 1 - st]) + q[c - st]) / 3.0;
    f1 = 1.0 * DCor_v[c + 1 - st] * ((q[c] + q[c + 1 - st]) + q[c - st]) / 3.0;
  }
}
__device__ __forceinline__ void f4v_of(const double *__restrict__ q, const double *__restrict__ DCor_u, size_t c, int st, int Sadourny,
                                       double &f0, double &f1, double &f2, double &f3) {
  if (Sadourny) {
    f0 = 1.0 * DCor_u[c - 1] * q[c - 1];
    f1 = 1.0 * DCor_u[c] * q[c];
    f3 = 1.0 * DCor_u[c + st] * q[c];
    f2 = 1.0 * DCor_u[c - 1 + st] * q[c - 1];
  } else {
    f0 = 1.0 * DCor_u[c - 1] * ((q[c] + q[c - 1 - st]) + q[c - 1]) / 3.0;
    f1 = 1.0 * DCor_u[c] * (q[c] + (q[c - 1] + q[c - st])) / 3.0;
    f3 = 1.0 * DCor_u[c + st] * (q[c] + (q[c - 1] + q[c + st])) / 3.0;
    f2 = 1.0 * DCor_u[c - 1 + st] * ((q[c] + q[c - 1 + st]) + q[c - 1]) / 3.0;
  }
}

// The new velocity of one face (btloop_find_PF :3088-3160 + btloop_update_u :3163 / _v :3273): n0..n3 are the four velocities of
// the OTHER component around the face in the order the Coriolis sums take them (u: vbt at c+1, c-st, c, c+1-st; v: ubt at c-1,
// c, c+st, c-1+st) -- from memory or from a neighbouring thread, the expressions are the same.
template <int DIR>
__device__ __forceinline__ void bt_face_update(const Dm &d, const double *__restrict__ G, const double *work, const LoopArgs &A, size_t c, int st,
                                               size_t slab, const double *etaB, double vel, double n0, double n1, double n2, double n3,
                                               int bracket_bug, double &newv, double &CorPF) {
  const double *eta_PF = work + W_eta_PF * slab;
  double Cor, PF, f0, f1, f2, f3;
  if (DIR == 0) {
    const double *gE = work + W_gtot_E * slab, *gW = work + W_gtot_W * slab;
    PF = (((etaB[c] - eta_PF[c]) * gE[c]) - ((etaB[c + 1] - eta_PF[c + 1]) * gW[c + 1])) * A.dgeo_de * gm(G, d, MOM6X_G_IdxCu)[c];
    f4u_of(work + W_q * slab, work + W_DCor_v * slab, c, st, A.Sadourny, f0, f1, f2, f3);
    Cor = (((f3 * n0) + (f0 * n1)) + ((f2 * n2) + (f1 * n3))) - work[W_Cor_ref_u * slab + c];
    newv = work[W_bt_rem_u * slab + c] * (vel + A.dtbt * ((work[W_BT_force_u * slab + c] + Cor) + PF));
  } else {
    const double *gN = work + W_gtot_N * slab, *gS = work + W_gtot_S * slab;
    PF = (((etaB[c] - eta_PF[c]) * gN[c]) - ((etaB[c + st] - eta_PF[c + st]) * gS[c + st])) * A.dgeo_de * gm(G, d, MOM6X_G_IdyCv)[c];
    f4v_of(work + W_q * slab, work + W_DCor_u * slab, c, st, A.Sadourny, f0, f1, f2, f3);
    if (bracket_bug) Cor = -1.0 * (((f0 * n0) + (f1 * n1)) + ((f3 * n2) + (f2 * n3))) - work[W_Cor_ref_v * slab + c];
    else Cor = -1.0 * (((f0 * n0) + (f3 * n2)) + ((f1 * n1) + (f2 * n3))) - work[W_Cor_ref_v * slab + c];
    newv = work[W_bt_rem_v * slab + c] * (vel + A.dtbt * ((work[W_BT_force_v * slab + c] + Cor) + PF));
  }
  if (fabs(newv) < A.vel_underflow) newv = 0.0;
  CorPF = Cor + PF;
}

// btloop_find_PF + btloop_update_u/v + transports + running sums for ONE velocity component.
// DIR = 0: u (faces I), DIR = 1: v (faces J).  (a0..a1, b0..b1) is the update range.
template <int DIR>
__global__ void __launch_bounds__(256)
k_bt_vel(Dm d, const double *__restrict__ G, double *work, double *btav, double *hbtav, LoopArgs A,
         int a0, int a1, int b0, int b1, int bracket_bug) {
  const int i = I_BASE(a0) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = b0 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < a0 || i > a1 || j > b1) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *etaB = work + (A.project ? W_eta : W_eta_pred) * slab;
  double vel, newv, CorPF;
  if (DIR == 0) {
    const double *vbt = work + W_vbt * slab;
    vel = work[W_ubt * slab + c];
    bt_face_update<0>(d, G, work, A, c, st, slab, etaB, vel, vbt[c + 1], vbt[c - st], vbt[c], vbt[c + 1 - st], 0, newv, CorPF);
    work[W_ubt * slab + c] = newv;
    work[W_u_accel_bt * slab + c] = work[W_u_accel_bt * slab + c] + A.wt_accel * CorPF;
  } else {
    const double *ubt = work + W_ubt * slab;
    vel = work[W_vbt * slab + c];
    bt_face_update<1>(d, G, work, A, c, st, slab, etaB, vel, ubt[c - 1], ubt[c], ubt[c + st], ubt[c - 1 + st], bracket_bug, newv, CorPF);
    work[W_vbt * slab + c] = newv;
    work[W_v_accel_bt * slab + c] = work[W_v_accel_bt * slab + c] + A.wt_accel * CorPF;
  }
  if (A.store_uhn)   // btloop_eta_predictor's transport of this face at the new velocity (:2975-2985), for the next sub-step
    work[(DIR ? W_vhn : W_uhn) * slab + c] = find_uhbt(newv, work + (DIR ? W_BTCv : W_BTCu) * slab, c, slab) + work[(DIR ? W_vhbt0 : W_uhbt0) * slab + c];
  // transports on (isv-1..iev, jsv..jev) | (isv..iev, jsv-1..jev)  :2624-2632
  const bool in_trans = DIR ? (i >= A.isv && i <= A.iev && j >= A.jsv - 1 && j <= A.jev)
                            : (i >= A.isv - 1 && i <= A.iev && j >= A.jsv && j <= A.jev);
  if (in_trans) {
    const double trans = A.trans_wt1 * newv + A.trans_wt2 * vel;
    const double hbt = find_uhbt(trans, work + (DIR ? W_BTCv : W_BTCu) * slab, c, slab) + work[(DIR ? W_vhbt0 : W_uhbt0) * slab + c];
    work[(DIR ? W_vhbt : W_uhbt) * slab + c] = hbt;
    const bool in_c = DIR ? (i >= 0 && i <= d.ni - 1 && j >= -1 && j <= d.nj - 1) : (i >= -1 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1);
    if (in_c) {   // running sums :2690-2700
      btav[c] = btav[c] + A.wt_trans * trans;
      hbtav[c] = hbtav[c] + A.wt_trans * hbt;
      if (A.wt_vel != 0.0) {   // x + 0.0 * newv == x for every x this sum can hold (it starts at +0.0); the filter's weights are
        double *wtd = work + (DIR ? W_vbt_wtd : W_ubt_wtd) * slab;   // zero for the first nstep - nfilter sub-steps (:1738-1795)
        wtd[c] = wtd[c] + A.wt_vel * newv;
      }
    }
  }
}

// eta corrector :2721-2727
__global__ void __launch_bounds__(256)
k_bt_eta(Dm d, const double *__restrict__ G, double *work, LoopArgs A, double Z_to_H, unsigned long long *warn,
         double *warn_info) {
  const int i = I_BASE(A.isv) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = A.jsv + blockIdx.y * blockDim.y + threadIdx.y;
  if (i < A.isv || i > A.iev || j > A.jev) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *uhbt = work + W_uhbt * slab, *vhbt = work + W_vhbt * slab;
  const double e = (work[W_eta * slab + c] + work[W_eta_src * slab + c]) +
                   (A.dtbt * gm(G, d, MOM6X_G_IareaT)[c]) * ((uhbt[c - 1] - uhbt[c]) + (vhbt[c - st] - vhbt[c]));
  work[W_eta * slab + c] = e;
  if (A.wt_eta != 0.0) work[W_eta_wtd * slab + c] = work[W_eta_wtd * slab + c] + e * A.wt_eta;
  if (A.pred_next) {   // btloop_eta_predictor of sub-step n + 1 (k_bt_pred's expressions): its range is this kernel's range, its
    const double *uhn = work + W_uhn * slab, *vhn = work + W_vhn * slab;   // transports are W_uhn / W_vhn, its eta is e
    const double eta_PF_BT = (e + work[W_eta_src * slab + c]) +
                             (A.dtbt * gm(G, d, MOM6X_G_IareaT)[c]) * ((uhn[c - 1] - uhn[c]) + (vhn[c - st] - vhn[c]));
    work[W_eta_pred * slab + c] = eta_PF_BT;
    if (A.find_etaav && (fabs(A.wt_accel2_next) > 0.0) && i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1)
      work[W_eta_sum * slab + c] = work[W_eta_sum * slab + c] + A.wt_accel2_next * eta_PF_BT;
  }
  // :2738-2745 (Boussinesq): unphysical sea surface height over the computational domain -- counted, the first one kept
  if (i >= 0 && i < d.ni && j >= 0 && j < d.nj) {
    const double bT = gm(G, d, MOM6X_G_bathyT)[c];
    if ((e < -Z_to_H * bT) && (gm(G, d, MOM6X_G_mask2dT)[c] > 0.0)) {
      atomicAdd(&warn[0], 1ULL);
      if (atomicCAS(&warn[1], 0ULL, 1ULL) == 0ULL) { warn_info[0] = e; warn_info[1] = -bT; warn_info[2] = (double)i; warn_info[3] = (double)j; }
    }
  }
}

// ONE launch per barotropic sub-step: first velocity component -> second -> eta corrector (+ the next sub-step's eta predictor),
// the same expressions as k_bt_vel<DIR> and k_bt_eta, on a 32 x 16 tile of threads whose results go from stage to stage through LDS.
// A block owns OX x OY cells, their east / north faces (and the west / south faces of the range's first cells, the extra columns /
// rows the first component is advanced on): only owned points are written and summed.  The one-point frame the second component's
// Coriolis term and the corrector's divergence need of the first is recomputed by the block itself -- the same inputs, the same
// expressions, the same bits as the owner's.  What a neighbouring block reads while its owner rewrites it (ubt, vbt, eta_pred) has
// two copies: a sub-step reads set `pin` and writes set `pout`.  (k_bt_vel / k_bt_eta stay for CLIP_BT_VELOCITY and
// BT_PROJECT_VELOCITY, whose predictor is a different one.)  3.25 -> 1.25 launches per sub-step; at the 360 x 540 tile of an 8-GPU
// layout the three launches were 27 us of which half was launch ramp.
struct SubPlanes { int u_in, u_out, v_in, v_out, e_in, e_out; };
template <bool VFIRST, int BSX, int BSY>
__global__ void __launch_bounds__(BSX * BSY)
k_bt_substep(Dm d, const double *__restrict__ G, double *work, double *ubtav, double *uhbtav, double *vbtav, double *vhbtav, LoopArgs A,
             SubPlanes P, int bracket_bug, double Z_to_H, unsigned long long *warn, double *warn_info, int nbx, int sel) {
  constexpr int OX = VFIRST ? BSX - 2 : BSX - 1, OY = VFIRST ? BSY - 1 : BSY - 2;
  __shared__ double s_u[BSY][BSX], s_v[BSY][BSX], s_hu[BSY][BSX], s_hv[BSY][BSX], s_un[BSY][BSX], s_vn[BSY][BSX];
  const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
  const int x0 = A.isv + bx * OX, y0 = A.jsv + by * OY;
  // sel 1 / 2: only the blocks that read nothing but the tile's own points (the threads' points x0-1 .. x0+BSX-2 and one around
  // them) / only the others: the first half runs while the group pass of the state travels, the second when it has arrived.  A
  // point's result does not depend on the launch its block is in.
  if (sel) {
    const bool inner = (x0 - 2 >= 0) && (x0 + BSX - 1 <= d.ni - 1) && (y0 - 2 >= 0) && (y0 + BSY - 1 <= d.nj - 1);
    if (inner != (sel == 1)) return;
  }
  const int x1 = min(x0 + OX - 1, A.iev), y1 = min(y0 + OY - 1, A.jev);
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int i = x0 - 1 + tx, j = y0 - 1 + ty;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *etaB = work + (size_t)P.e_in * slab;
  const double *ubt_in = work + (size_t)P.u_in * slab, *vbt_in = work + (size_t)P.v_in * slab;
  double *ubt_out = work + (size_t)P.u_out * slab, *vbt_out = work + (size_t)P.v_out * slab;
  const bool own_x = (i >= x0 && i <= x1), own_y = (j >= y0 && j <= y1);
  const bool west = (i == A.isv - 1 && x0 == A.isv), south = (j == A.jsv - 1 && y0 == A.jsv);
  s_u[ty][tx] = 0.0; s_v[ty][tx] = 0.0; s_hu[ty][tx] = 0.0; s_hv[ty][tx] = 0.0; s_un[ty][tx] = 0.0; s_vn[ty][tx] = 0.0;

  // one face: the update, the next predictor's transport, this sub-step's transport and the running sums (k_bt_vel's lines)
  auto face = [&](auto dir_tag, bool valid, bool owner, double n0, double n1, double n2, double n3, int bb) {
    constexpr int DIR = decltype(dir_tag)::value;
    if (!valid) return;
    const double vel = (DIR ? vbt_in : ubt_in)[c];
    double newv, CorPF;
    bt_face_update<DIR>(d, G, work, A, c, st, slab, etaB, vel, n0, n1, n2, n3, bb, newv, CorPF);
    (DIR ? s_v : s_u)[ty][tx] = newv;
    const double hn = find_uhbt(newv, work + (DIR ? W_BTCv : W_BTCu) * slab, c, slab) + work[(DIR ? W_vhbt0 : W_uhbt0) * slab + c];
    (DIR ? s_vn : s_un)[ty][tx] = hn;
    if (owner) {
      (DIR ? vbt_out : ubt_out)[c] = newv;
      double *acc = work + (DIR ? W_v_accel_bt : W_u_accel_bt) * slab;
      acc[c] = acc[c] + A.wt_accel * CorPF;
      work[(DIR ? W_vhn : W_uhn) * slab + c] = hn;
    }
    const bool in_trans = DIR ? (i >= A.isv && i <= A.iev && j >= A.jsv - 1 && j <= A.jev)
                              : (i >= A.isv - 1 && i <= A.iev && j >= A.jsv && j <= A.jev);
    if (in_trans) {
      const double trans = A.trans_wt1 * newv + A.trans_wt2 * vel;
      const double hbt = find_uhbt(trans, work + (DIR ? W_BTCv : W_BTCu) * slab, c, slab) + work[(DIR ? W_vhbt0 : W_uhbt0) * slab + c];
      (DIR ? s_hv : s_hu)[ty][tx] = hbt;
      const bool in_c = DIR ? (i >= 0 && i <= d.ni - 1 && j >= -1 && j <= d.nj - 1) : (i >= -1 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1);
      if (owner && in_c) {   // running sums :2690-2700
        double *btav = DIR ? vbtav : ubtav, *hbtav = DIR ? vhbtav : uhbtav;
        btav[c] = btav[c] + A.wt_trans * trans;
        hbtav[c] = hbtav[c] + A.wt_trans * hbt;
        if (A.wt_vel != 0.0) {
          double *wtd = work + (DIR ? W_vbt_wtd : W_ubt_wtd) * slab;
          wtd[c] = wtd[c] + A.wt_vel * newv;
        }
      }
    }
  };
  using U = std::integral_constant<int, 0>;
  using V = std::integral_constant<int, 1>;
  if (VFIRST) {
    // v on (isv-1 .. iev+1, jsv-1 .. jev): every thread of the tile; the old ubt around it from memory
    const bool valid = (i <= A.iev + 1 && j <= A.jev && i <= x1 + 1 && j <= y1);
    const bool owner = (own_x || west || (i == A.iev + 1 && x1 == A.iev)) && (own_y || south);
    double n0 = 0., n1 = 0., n2 = 0., n3 = 0.;
    if (valid) { n0 = ubt_in[c - 1]; n1 = ubt_in[c]; n2 = ubt_in[c + st]; n3 = ubt_in[c - 1 + st]; }
    face(V{}, valid, owner, n0, n1, n2, n3, 0);
    __syncthreads();
    // u on (isv-1 .. iev, jsv .. jev): the new vbt around it from the tile
    const bool valid2 = (tx <= BSX - 2 && ty >= 1 && i <= x1 && j <= y1);
    double m0 = 0., m1 = 0., m2 = 0., m3 = 0.;
    if (valid2) { m0 = s_v[ty][tx + 1]; m1 = s_v[ty - 1][tx]; m2 = s_v[ty][tx]; m3 = s_v[ty - 1][tx + 1]; }
    face(U{}, valid2, (own_x || west) && own_y, m0, m1, m2, m3, 0);
  } else {
    // u on (isv-1 .. iev, jsv-1 .. jev+1)
    const bool valid = (i <= A.iev && j <= A.jev + 1 && i <= x1 && j <= y1 + 1);
    const bool owner = (own_x || west) && (own_y || south || (j == A.jev + 1 && y1 == A.jev));
    double n0 = 0., n1 = 0., n2 = 0., n3 = 0.;
    if (valid) { n0 = vbt_in[c + 1]; n1 = vbt_in[c - st]; n2 = vbt_in[c]; n3 = vbt_in[c + 1 - st]; }
    face(U{}, valid, owner, n0, n1, n2, n3, 0);
    __syncthreads();
    // v on (isv .. iev, jsv-1 .. jev)
    const bool valid2 = (tx >= 1 && ty <= BSY - 2 && i <= x1 && j <= y1);
    double m0 = 0., m1 = 0., m2 = 0., m3 = 0.;
    if (valid2) { m0 = s_u[ty][tx - 1]; m1 = s_u[ty][tx]; m2 = s_u[ty + 1][tx]; m3 = s_u[ty + 1][tx - 1]; }
    face(V{}, valid2, own_x && (own_y || south), m0, m1, m2, m3, bracket_bug);
  }
  __syncthreads();
  // eta corrector :2721-2727 on the block's own cells (k_bt_eta's lines)
  if (own_x && own_y && tx >= 1 && ty >= 1) {
    const double dtA = A.dtbt * gm(G, d, MOM6X_G_IareaT)[c];
    const double e = (work[W_eta * slab + c] + work[W_eta_src * slab + c]) +
                     dtA * ((s_hu[ty][tx - 1] - s_hu[ty][tx]) + (s_hv[ty - 1][tx] - s_hv[ty][tx]));
    work[W_eta * slab + c] = e;
    if (A.wt_eta != 0.0) work[W_eta_wtd * slab + c] = work[W_eta_wtd * slab + c] + e * A.wt_eta;
    if (A.pred_next) {
      const double eta_PF_BT = (e + work[W_eta_src * slab + c]) +
                               dtA * ((s_un[ty][tx - 1] - s_un[ty][tx]) + (s_vn[ty - 1][tx] - s_vn[ty][tx]));
      work[(size_t)P.e_out * slab + c] = eta_PF_BT;
      if (A.find_etaav && (fabs(A.wt_accel2_next) > 0.0) && i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1)
        work[W_eta_sum * slab + c] = work[W_eta_sum * slab + c] + A.wt_accel2_next * eta_PF_BT;
    }
    if (i >= 0 && i < d.ni && j >= 0 && j < d.nj) {
      const double bT = gm(G, d, MOM6X_G_bathyT)[c];
      if ((e < -Z_to_H * bT) && (gm(G, d, MOM6X_G_mask2dT)[c] > 0.0)) {
        atomicAdd(&warn[0], 1ULL);
        if (atomicCAS(&warn[1], 0ULL, 1ULL) == 0ULL) { warn_info[0] = e; warn_info[1] = -bT; warn_info[2] = (double)i; warn_info[3] = (double)j; }
      }
    }
  }
}

// truncate_velocities :2918-2944
__global__ void k_bt_clip(Dm d, const double *__restrict__ G, double *work, double dt, double CFL_trunc,
                          int isv, int iev, int jsv, int jev) {
  const int i = isv - 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = jsv - 1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > iev || j > jev) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const double *IareaT = gm(G, d, MOM6X_G_IareaT), *areaT = gm(G, d, MOM6X_G_areaT);
  if (j >= jsv) {
    const double dy = gm(G, d, MOM6X_G_dy_Cu)[c];
    double u = work[W_ubt * slab + c];
    if ((u * (dt * dy)) * IareaT[c + 1] < -CFL_trunc) u = (-0.95 * CFL_trunc) * (areaT[c + 1] / (dt * dy));
    else if ((u * (dt * dy)) * IareaT[c] > CFL_trunc) u = (0.95 * CFL_trunc) * (areaT[c] / (dt * dy));
    work[W_ubt * slab + c] = u;
  }
  if (i >= isv) {
    const double dx = gm(G, d, MOM6X_G_dx_Cv)[c];
    double v = work[W_vbt * slab + c];
    if ((v * (dt * dx)) * IareaT[c + st] < -CFL_trunc) v = (-0.9 * CFL_trunc) * (areaT[c + st] / (dt * dx));
    else if ((v * (dt * dx)) * IareaT[c] > CFL_trunc) v = (0.9 * CFL_trunc) * (areaT[c] / (dt * dx));
    work[W_vbt * slab + c] = v;
  }
}

// after the loop :1807-1847
__global__ void k_bt_post(Dm d, double *work, const double *__restrict__ eta_in, double *eta_out, double *etaav,
                          double dgeo_de, double I_sum_wt_accel, double I_sum_wt_eta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  if (etaav) etaav[c] = work[W_eta_sum * slab + c] * I_sum_wt_accel;
  work[W_e_anom * slab + c] = dgeo_de * (0.5 * (work[W_eta * slab + c] + eta_in[c]) - work[W_eta_PF * slab + c]);
  eta_out[c] = work[W_eta_wtd * slab + c] * I_sum_wt_eta;
}

// btstep_layer_accel :3432-3504
__global__ void __launch_bounds__(256)
k_layer_accel(Dm d, const double *__restrict__ G, const double *__restrict__ work, const double *__restrict__ pbce,
              double *__restrict__ accel_layer_u, double *__restrict__ accel_layer_v, double accel_underflow) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < (-1)) return;
  const int st = d.pitch;
  const size_t c = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  // the eleven 2-D operands stay in registers while the thread walks KCHUNK layers of pbce
  const double *e_anom = work + W_e_anom * slab;
  const double ea = e_anom[c], ea_E = e_anom[c + 1], ea_N = e_anom[c + st];
  const double g_E = work[W_gtot_E * slab + c], g_W = work[W_gtot_W * slab + c + 1];
  const double g_N = work[W_gtot_N * slab + c], g_S = work[W_gtot_S * slab + c + st];
  const double ua = work[W_u_accel_bt * slab + c], va = work[W_v_accel_bt * slab + c];
  const double IdxCu = gm(G, d, MOM6X_G_IdxCu)[c], IdyCv = gm(G, d, MOM6X_G_IdyCv)[c];
  for (int k = k0; k < k1; k++) {
    const size_t c3 = c + (size_t)k * slab;
    const double pb = pbce[c3];
    if (j >= 0) {
      double a = (ua - (((pbce[c3 + 1] - g_W) * ea_E) - ((pb - g_E) * ea)) * IdxCu);
      if (fabs(a) < accel_underflow) a = 0.0;
      accel_layer_u[c3] = a;
    }
    if (i >= 0) {
      double a = (va - (((pbce[c3 + st] - g_S) * ea_N) - ((pb - g_N) * ea)) * IdyCv);
      if (fabs(a) < accel_underflow) a = 0.0;
      accel_layer_v[c3] = a;
    }
  }
}

inline dim3 blk2() { return dim3(64, 4, 1); }

}  // namespace

void bt_defer_layer_accel(mom6x_ctx *c, bool on) { if (c->bts) c->bts->la_defer = on; }

// btcalc inside the RK2 step: with `on`, mom6x_btcalc(h_u, h_v) only notes its operands; btstep's column pass forms the
// thickness fractions from them (k_bt_col<., true>), and whoever else wants frhatu / frhatv (set_dtbt, mom6x_barotropic_field)
// gets them through bt_frhat_materialize.  MOM6X_BTCALC=eager: btcalc always writes them.
void bt_defer_btcalc(mom6x_ctx *c, bool on) {
  static const bool eager = [] { const char *e = getenv("MOM6X_BTCALC"); return e && !strcmp(e, "eager"); }();
  if (c->bts) c->bts->fr_defer = on && !eager;
}
// btcalc :4360 with h_u, h_v given (BT_THICK_SCHEME = FROM_BT_CONT): the column in registers up to COLS_NK_BOUND layers
static void btcalc_from_faces(mom6x_ctx *c, const double *h_u, const double *h_v) {
  const Dm d = c->d;
  const dim3 b = blk2();
  if (d.nk <= COLS_NK_BOUND) {
#define BTC(NKT) do {                                                                                                                   \
    KLAUNCH(c, "k_btcalc<0>", (k_btcalc_cols<0, NKT>), grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, h_u, c->bts->frhatu, c->GV.H_subroundoff); \
    KLAUNCH(c, "k_btcalc<1>", (k_btcalc_cols<1, NKT>), grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, h_v, c->bts->frhatv, c->GV.H_subroundoff); } while (0)
    COLS_NK_DISPATCH(d.nk, BTC);
#undef BTC
    return;
  }
  KLAUNCH(c, "k_btcalc<0>", k_btcalc<0>, grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, (const double *)nullptr, h_u, c->bts->frhatu,
          c->GV.H_subroundoff, c->GV.Z_to_H, 0);
  KLAUNCH(c, "k_btcalc<1>", k_btcalc<1>, grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, (const double *)nullptr, h_v, c->bts->frhatv,
          c->GV.H_subroundoff, c->GV.Z_to_H, 0);
}
int bt_frhat_materialize(mom6x_ctx *c) {
  BTState *s = c->bts;
  if (!s || !s->fr_pending) return MOM6X_OK;
  HIPCHK(hipSetDevice(c->device));
  btcalc_from_faces(c, s->fr_hu, s->fr_hv);
  s->fr_pending = false;
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

bool bt_layer_accel_src(mom6x_ctx *c, LayerAccelSrc *u, LayerAccelSrc *v) {
  BTState *s = c->bts;
  if (!s || !s->la_pending) return false;
  const size_t slab = (size_t)c->d.slab;
  const double *w = s->work;
  u->pbce = v->pbce = s->la_pbce; u->e_anom = v->e_anom = w + W_e_anom * slab;
  u->g_own = w + W_gtot_E * slab; u->g_nbr = w + W_gtot_W * slab; u->a2d = w + W_u_accel_bt * slab;
  v->g_own = w + W_gtot_N * slab; v->g_nbr = w + W_gtot_S * slab; v->a2d = w + W_v_accel_bt * slab;
  u->underflow = v->underflow = s->la_underflow;
  return true;
}

int bt_layer_accel_materialize(mom6x_ctx *c, double *accel_layer_u, double *accel_layer_v) {
  BTState *s = c->bts;
  // (a btstep from outside the step has replaced the work block the deferred accelerations were to be formed from: the caller
  //  must not hand out arrays nobody wrote)
  REQUIRE(s && s->la_pending, MOM6X_EINVAL, "btstep_layer_accel: the deferred layer accelerations of the step's last btstep are gone "
          "(another btstep has run since)");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b = blk2();
  KLAUNCH(c, "k_layer_accel", k_layer_accel, grid3(nxa(d.ni + 1, -1), d.nj + 1, nchunks(d.nk), b), b, d, c->G, (const double *)s->work,
          s->la_pbce, accel_layer_u, accel_layer_v, s->la_underflow);
  s->la_pending = false;
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

void bt_state_free(mom6x_ctx *c) {
  if (!c->bts) return;
  BTState *s = c->bts;
  double *ptrs[] = { s->frhatu, s->frhatv, s->IDatu, s->IDatv, s->ubtav, s->vbtav, s->eta_cor, s->q_D, s->D_u_Cor,
                     s->D_v_Cor, s->work, s->warn_info };
  for (double *p : ptrs) (void)hipFree(p);
  (void)hipFree(s->warn);
  delete s;
  c->bts = nullptr;
}

extern "C" int mom6x_btstep_warnings(mom6x_ctx *c, int reset, long long *count, double *info) {
  REQUIRE(c && c->bts && count, MOM6X_EINVAL, "mom6x_btstep_warnings: barotropic_init has not been called");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  unsigned long long w[2];
  HIPCHK(hipMemcpy(w, c->bts->warn, sizeof(w), hipMemcpyDeviceToHost));
  *count = (long long)w[0];
  if (info) HIPCHK(hipMemcpy(info, c->bts->warn_info, 4 * sizeof(double), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(c->bts->warn, 0, sizeof(w)));
  return MOM6X_OK;
}

extern "C" int mom6x_barotropic_init(mom6x_ctx *c, const mom6x_barotropic_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_barotropic_init: null argument");
  REQUIRE(p->bt_thick_scheme >= MOM6X_BT_THICK_FROM_BT_CONT && p->bt_thick_scheme <= MOM6X_BT_THICK_ARITHMETIC, MOM6X_EINVAL,
          "barotropic_init: Unrecognized setting of BT_THICK_SCHEME");
  REQUIRE(c->dims.halo >= 2, MOM6X_EINVAL, "barotropic: halo >= 2 required");
  REQUIRE(p->BTHALO <= c->dims.halo, MOM6X_EINVAL,
          "barotropic_init: BTHALO exceeds the halo of the tile context; create the context with halo = max(NIHALO, BTHALO)");
  REQUIRE(p->min_stencil >= 0 && p->min_stencil <= c->dims.halo, MOM6X_EINVAL, "barotropic_init: bad BT_WIDE_HALO_MIN_STENCIL");
  HIPCHK(hipSetDevice(c->device));
  c->bt = *p;
  const Dm d = c->d;
  if (!c->bts) {
    BTState *s = new BTState();
    memset(s, 0, sizeof(*s));
    const size_t n2 = (size_t)d.slab, n3 = n2 * d.nk;
    double **p3[] = { &s->frhatu, &s->frhatv };
    for (double **q : p3) { HIPCHK(hipMalloc(q, n3 * sizeof(double))); HIPCHK(hipMemsetAsync(*q, 0, n3 * sizeof(double), c->stream)); }
    double **p2[] = { &s->IDatu, &s->IDatv, &s->ubtav, &s->vbtav, &s->eta_cor, &s->q_D, &s->D_u_Cor, &s->D_v_Cor };
    for (double **q : p2) { HIPCHK(hipMalloc(q, n2 * sizeof(double))); HIPCHK(hipMemsetAsync(*q, 0, n2 * sizeof(double), c->stream)); }
    HIPCHK(hipMalloc(&s->work, (size_t)W_COUNT * n2 * sizeof(double)));
    HIPCHK(hipMalloc(&s->warn, 2 * sizeof(unsigned long long))); HIPCHK(hipMemsetAsync(s->warn, 0, 2 * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMalloc(&s->warn_info, 4 * sizeof(double))); HIPCHK(hipMemsetAsync(s->warn_info, 0, 4 * sizeof(double), c->stream));
    c->bts = s;
  }
  BTState *s = c->bts;
  s->nstep_last = 0;
  const dim3 b = blk2();
  KLAUNCH(c, "k_bt_init_static", k_bt_init_static, grid3(d.ni + 1, d.nj + 1, 1, b), b, d, c->G, c->GV.Z_to_H, p->Z_ref,
                     p->BT_Coriolis_scale, c->GV.H_subroundoff, s->q_D, s->D_u_Cor, s->D_v_Cor, s->IDatu, s->IDatv);
  double *f[] = { s->q_D, s->D_u_Cor, s->D_v_Cor };
  const int stg[] = { 3, 1, 2 }, nks[] = { 1, 1, 1 };
  halo_wrap(c, f, stg, nks, 3);
  HIPCHK(hipGetLastError());
  c->bt_init = true;
  return MOM6X_OK;
}

extern "C" double *mom6x_barotropic_field(mom6x_ctx *c, int which) {
  if (!c || !c->bts) return nullptr;
  BTState *s = c->bts;
  if ((which == 3 || which == 4) && bt_frhat_materialize(c)) return nullptr;
  switch (which) {
    case 0: return s->ubtav; case 1: return s->vbtav; case 2: return s->eta_cor; case 3: return s->frhatu;
    case 4: return s->frhatv; case 5: return s->IDatu; case 6: return s->IDatv; case 7: return s->q_D;
    case 8: return s->D_u_Cor; case 9: return s->D_v_Cor; case 10: return s->work;
    default: return nullptr;
  }
}

// CS%dtbt, the scalar restart variable "DTBT" (register_barotropic_restarts :6290): read it for save_restart, set it
// after restore_state (barotropic_init :5962-5970 keeps a restart's positive DTBT instead of its own estimate).
extern "C" int mom6x_barotropic_dtbt(mom6x_ctx *c, double *get, const double *set) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "mom6x_barotropic_dtbt: Module MOM_barotropic must be initialized before it is used.");
  if (set) { REQUIRE(*set > 0.0, MOM6X_EINVAL, "mom6x_barotropic_dtbt: DTBT must be positive"); c->bt.dtbt = *set; }
  if (get) *get = c->bt.dtbt;
  return MOM6X_OK;
}

// btcalc without may_use_default (the calls of step_MOM_dyn_split_RK2 :628, :650, :868): :4426-4429
extern "C" int mom6x_btcalc_strict(mom6x_ctx *c, const double *h, const double *h_u, const double *h_v) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "btcalc: Module MOM_barotropic must be initialized before it is used.");
  REQUIRE((h_u && h_v) || c->bt.bt_thick_scheme != MOM6X_BT_THICK_FROM_BT_CONT, MOM6X_EINVAL,
          "btcalc: Inconsistent settings of optional arguments and hvel_scheme.");
  return mom6x_btcalc(c, h, h_u, h_v);
}
extern "C" int mom6x_btcalc(mom6x_ctx *c, const double *h, const double *h_u, const double *h_v) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "btcalc: Module MOM_barotropic must be initialized before it is used.");
  REQUIRE((h_u != nullptr) == (h_v != nullptr), MOM6X_EINVAL, "btcalc: Inconsistent settings of optional arguments");
  REQUIRE(h_u || h, MOM6X_EINVAL, "btcalc: h is required when h_u/h_v are absent");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b = blk2();
  c->bts->fr_pending = false;
  if (h_u && c->bts->fr_defer) { c->bts->fr_pending = true; c->bts->fr_hu = h_u; c->bts->fr_hv = h_v; return MOM6X_OK; }
  if (h_u) {
    btcalc_from_faces(c, h_u, h_v);
    HIPCHK(hipGetLastError());
    return MOM6X_OK;
  }
  const int scheme = (c->bt.bt_thick_scheme == MOM6X_BT_THICK_FROM_BT_CONT) ? MOM6X_BT_THICK_HYBRID : c->bt.bt_thick_scheme;   // (use_default :4421-4424)
  KLAUNCH(c, "k_btcalc<0>", k_btcalc<0>, grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, h, h_u, c->bts->frhatu,
                     c->GV.H_subroundoff, c->GV.Z_to_H, scheme);
  KLAUNCH(c, "k_btcalc<1>", k_btcalc<1>, grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, h, h_v, c->bts->frhatv,
                     c->GV.H_subroundoff, c->GV.Z_to_H, scheme);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

extern "C" int mom6x_bt_mass_source(mom6x_ctx *c, const double *h, const double *eta, int set_cor) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "bt_mass_source: Module MOM_barotropic must be initialized before it is used.");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b = blk2();
  KLAUNCH(c, "k_bt_mass_source", k_bt_mass_source, grid3(d.ni, d.nj, 1, b), b, d, c->G, h, eta, c->bts->eta_cor, set_cor,
                     c->GV.Z_to_H);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

int bt_mass_source_from(mom6x_ctx *c, const double *eta_h, const double *eta, int set_cor) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "bt_mass_source: Module MOM_barotropic must be initialized before it is used.");
  const Dm d = c->d;
  const dim3 b = blk2();
  KLAUNCH(c, "k_bt_mass_source_from", k_bt_mass_source_from, grid3(d.ni, d.nj, 1, b), b, d, eta_h, eta, c->bts->eta_cor, set_cor);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

static int set_dtbt_impl(mom6x_ctx *c, const double *pbce, double gtot_est, int add_max, double SSH_add, double *dtbt_out, const double *eta);
extern "C" int mom6x_set_dtbt(mom6x_ctx *c, const double *pbce, double gtot_est, double SSH_add, double *dtbt_out) {
  return set_dtbt_impl(c, pbce, gtot_est, 1, SSH_add, dtbt_out, nullptr);
}
// set_dtbt(G, GV, US, CS, pbce, eta=eta, SSH_add=) without a BT_cont argument (:3576-3582): the face areas are
// find_face_areas(eta=eta) :5171-5186 when NONLINEAR_BT_CONTINUITY is set (never with a BT_cont_type) and eta is given, and
// find_face_areas(add_max=add_SSH) :5208-5219 otherwise -- the harmonic-mean form of :5221-5236 is never reached from set_dtbt.
extern "C" int mom6x_set_dtbt_pbce_eta(mom6x_ctx *c, const double *pbce, const double *eta, double SSH_add, double *dtbt_out) {
  REQUIRE(pbce, MOM6X_EINVAL, "set_dtbt: Either pbce or gtot_est must be present.");
  const bool nonlin = c && c->bt_init && c->bt.nonlinear_continuity && eta;
  return set_dtbt_impl(c, pbce, 0.0, nonlin ? 0 : 1, nonlin ? 0.0 : SSH_add, dtbt_out, nonlin ? eta : nullptr);
}
// ... as called from step_MOM_dyn_split_RK2 :667 behind a BT_cont_type (eta has no part)
extern "C" int mom6x_set_dtbt_pbce(mom6x_ctx *c, const double *pbce, double *dtbt_out) {
  return mom6x_set_dtbt_pbce_eta(c, pbce, nullptr, 0.0, dtbt_out);
}
int set_dtbt_eta(mom6x_ctx *c, const double *pbce, const double *eta) { return mom6x_set_dtbt_pbce_eta(c, pbce, eta, 0.0, nullptr); }
static int set_dtbt_impl(mom6x_ctx *c, const double *pbce, double gtot_est, int add_max, double SSH_add, double *dtbt_out, const double *eta) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "set_dtbt: Module MOM_barotropic must be initialized before it is used.");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const dim3 b = blk2();
  BTState *s = c->bts;
  { const int rcm = bt_frhat_materialize(c); if (rcm) return rcm; }
  double *tmp = s->work + (size_t)W_eta_pred * d.slab;   // scratch plane
  HIPCHK(hipMemsetAsync(tmp, 0, sizeof(double) * d.slab, c->stream));
  KLAUNCH(c, "k_set_dtbt", k_set_dtbt, grid3(d.ni, d.nj, 1, b), b, d, c->G, pbce, s->frhatu, s->frhatv, gtot_est,
                     c->GV.Z_to_H, c->bt.Z_ref + SSH_add, add_max, c->bt.bebt, c->bt.BT_Coriolis_scale * c->bt.BT_Coriolis_scale, tmp, eta);
  HIPCHK(hipGetLastError());
  // min over the tile in the reference's (j outer, i inner) order is order-independent for min():
  std::vector<double> host((size_t)d.slab);
  HIPCHK(hipMemcpyAsync(host.data(), tmp, sizeof(double) * d.slab, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double min_max_dt2 = 1.0e38;
  for (int j = 0; j < d.nj; j++) for (int i = 0; i < d.ni; i++) {
    const double I2 = host[(size_t)(i + d.ioff) + (size_t)(j + d.joff) * d.pitch];
    if (I2 * min_max_dt2 > 1.0) min_max_dt2 = 1.0 / I2;
  }
  const double dgeo_de = 1.0 + fmax(0.0, c->bt.G_extra);
  double dtbt_max = sqrt(min_max_dt2 / dgeo_de);
  { int rc_ = comm_allreduce_scalar(c, &dtbt_max, 0); if (rc_) return rc_; }   // min_across_PEs(dtbt_max) :3622
  c->bt.dtbt = c->bt.dtbt_fraction * dtbt_max;
  if (dtbt_out) *dtbt_out = c->bt.dtbt;
  return MOM6X_OK;
}

extern "C" int mom6x_btstep(mom6x_ctx *c, const double *U_in, const double *V_in, const double *eta_in, double dt,
                            const double *bc_accel_u, const double *bc_accel_v, const double *taux, const double *tauy,
                            const double *pbce, const double *eta_PF_in, const double *U_Cor, const double *V_Cor,
                            double *accel_layer_u, double *accel_layer_v, double *eta_out, double *uhbtav,
                            double *vhbtav, const double *visc_rem_u, const double *visc_rem_v,
                            const mom6x_BT_cont *BT_cont, const double *taux_bot, const double *tauy_bot,
                            const double *uh0, const double *vh0, const double *u_uh0, const double *v_vh0,
                            double *etaav) {
  REQUIRE(c && c->bt_init, MOM6X_EINVAL, "btstep: Module MOM_barotropic must be initialized before it is used.");
  // BT_cont == NULL: USE_BT_CONT_TYPE = False (k_face_areas_as_fits); BOUND_BT_CORRECTION then bounds eta_cor by eta_cor_bound
  // (:6164-6173, :1582-1585), as it does behind a BT_cont_type with BT_CONT_CORR_BOUNDS = False
  REQUIRE(U_in && V_in && eta_in && bc_accel_u && bc_accel_v && taux && tauy && pbce && eta_PF_in && U_Cor && V_Cor &&
          accel_layer_u && accel_layer_v && eta_out && uhbtav && vhbtav && visc_rem_u && visc_rem_v,
          MOM6X_EINVAL, "btstep: null mandatory array");
  const bool add_uh0 = (uh0 != nullptr);
  REQUIRE(!add_uh0 || (vh0 && u_uh0 && v_vh0), MOM6X_EINVAL,
          "btstep: vh0, u_uh0, and v_vh0 must be associated if uh0 is used.");
  REQUIRE((taux_bot != nullptr) == (tauy_bot != nullptr), MOM6X_EINVAL, "btstep: taux_bot and tauy_bot come together");
  HIPCHK(hipSetDevice(c->device));
  const Dm d = c->d;
  const mom6x_barotropic_params &P = c->bt;
  BTState *s = c->bts;
  double *work = s->work;
  const size_t slab = (size_t)d.slab;
  const dim3 b = blk2();
  hipStream_t st = c->stream;
  const int is = 0, ie = d.ni - 1, js = 0, je = d.nj - 1;

  const double Idt = 1.0 / dt;
  const bool evolving_face_areas = !BT_cont && P.nonlinear_continuity && P.nonlin_cont_update_period > 0;   // :2425
  const int stencil = evolving_face_areas ? std::max(2, P.min_stencil) : ((P.min_stencil > 1) ? P.min_stencil : 1);   // :766-768 (no OBC)
  const int num_cycles = (P.use_wide_halos && d.halo / stencil >= 1) ? d.halo / stencil : 1;   // :790-792; the wide halo = the context's
  const int isvf = is - (num_cycles - 1) * stencil, ievf = ie + (num_cycles - 1) * stencil;
  const int jsvf = js - (num_cycles - 1) * stencil, jevf = je + (num_cycles - 1) * stencil;
  REQUIRE(P.dtbt > 0.0, MOM6X_EINVAL, "btstep: dtbt must be positive (call set_dtbt or set params.dtbt)");
  const int nstep = (int)ceil(dt / P.dtbt - 0.0001);
  s->nstep_last = nstep;
  const double Instep = 1.0 / (double)nstep;
  const double dtbt = dt * Instep;
  const double dgeo_de = 1.0 + P.G_extra;

  HIPCHK(hipMemsetAsync(work, 0, (size_t)W_COUNT * slab * sizeof(double), st));
  KLAUNCH(c, "k_bt_copy_in", k_bt_copy_in, grid3(d.ni + 2 * d.halo + 1, d.nj + 2 * d.halo + 1, 1, b), b, d, work, eta_in,
                     eta_PF_in, s->q_D, s->D_u_Cor, s->D_v_Cor);

  // ---- 3-D -> 2-D column pass
  ColArgs Au;
  memset(&Au, 0, sizeof(Au));
  Au.visc_rem = visc_rem_u; Au.frhat = s->frhatu; Au.U_Cor = U_Cor; Au.pbce = pbce; Au.uh0 = uh0; Au.u_uh0 = u_uh0;
  Au.U_in = U_in; Au.bc_accel = bc_accel_u; Au.tau = taux; Au.tau_bot = taux_bot; Au.IDat = s->IDatu;
  Au.ubt_Cor = work + W_ubt_Cor * slab; Au.gtot_m = work + W_gtot_E * slab; Au.gtot_p = work + W_gtot_W * slab;
  Au.uh0sum = work + W_uh0sum * slab; Au.ubt0 = work + W_ubt0 * slab; Au.ubt = work + W_ubt * slab;
  Au.BT_force = work + W_BT_force_u * slab; Au.bt_rem = work + W_bt_rem_u * slab;
  Au.Instep = Instep; Au.RZ_to_H = c->GV.RZ_to_H; Au.vel_underflow = P.vel_underflow; Au.nstep = nstep;
  Au.wt_uv_bug = P.wt_uv_bug; Au.visc_rem_u_uh0 = P.visc_rem_u_uh0; Au.strong_drag = P.strong_drag;
  ColArgs Av = Au;
  Av.visc_rem = visc_rem_v; Av.frhat = s->frhatv; Av.U_Cor = V_Cor; Av.uh0 = vh0; Av.u_uh0 = v_vh0; Av.U_in = V_in;
  Av.bc_accel = bc_accel_v; Av.tau = tauy; Av.tau_bot = tauy_bot; Av.IDat = s->IDatv;
  Av.ubt_Cor = work + W_vbt_Cor * slab; Av.gtot_m = work + W_gtot_N * slab; Av.gtot_p = work + W_gtot_S * slab;
  Av.uh0sum = work + W_vh0sum * slab; Av.ubt0 = work + W_vbt0 * slab; Av.ubt = work + W_vbt * slab;
  Av.BT_force = work + W_BT_force_v * slab; Av.bt_rem = work + W_bt_rem_v * slab;
  if (s->fr_pending) {   // btcalc's thickness fractions formed in the column pass (bt_defer_btcalc)
    Au.hf = s->fr_hu; Av.hf = s->fr_hv; Au.h_neglect = Av.h_neglect = c->GV.H_subroundoff;
    KLAUNCH(c, "k_bt_col<0>", (k_bt_col<0, true>), grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, Au);
    KLAUNCH(c, "k_bt_col<1>", (k_bt_col<1, true>), grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, Av);
  } else {
    { const int rcm = bt_frhat_materialize(c); if (rcm) return rcm; }
    KLAUNCH(c, "k_bt_col<0>", (k_bt_col<0, false>), grid3(nxa(d.ni + 1, -1), d.nj, 1, b), b, d, c->G, Au);
    KLAUNCH(c, "k_bt_col<1>", (k_bt_col<1, false>), grid3(d.ni, d.nj + 1, 1, b), b, d, c->G, Av);
  }

  // ---- BT_cont fits (set_local_BT_cont_types, halo = 1+ievf-ie)
  double *tmp = work + W_BTtmp * slab;
  if (BT_cont) KLAUNCH(c, "k_btcont_copy", k_btcont_copy, grid3(d.ni + 1, d.nj + 1, 1, b), b, d, *BT_cont, tmp);
  else KLAUNCH(c, "k_face_areas_as_fits", k_face_areas_as_fits, grid3(d.ni + 3, d.nj + 3, 1, b), b, d, c->G, P.Z_ref, c->GV.Z_to_H, tmp,
               P.nonlinear_continuity ? (const double *)(work + W_eta * slab) : (const double *)nullptr);
  {
    // the twelve BT_cont planes (set_local_BT_cont_types :4876, halo = 1+ievf-ie) and, in the same packed message, what the column
    // pass has made and btstep passes next (:1421-1431: gtot_*, ubt_Cor, vbt_Cor -- no kernel between the two reads the other's halos)
    double *f[18]; int stg[18], nks[18];
    for (int m = 0; m < 12; m++) { f[m] = tmp + (size_t)m * slab; stg[m] = (m < 6) ? 1 : 2; nks[m] = 1; }
    double *g[] = { work + W_gtot_E * slab, work + W_gtot_N * slab, work + W_gtot_W * slab, work + W_gtot_S * slab,
                    work + W_ubt_Cor * slab, work + W_vbt_Cor * slab };
    const int gs[] = { 0, 0, 0, 0, 1, 2 };
    for (int m = 0; m < 6; m++) { f[12 + m] = g[m]; stg[12 + m] = gs[m]; nks[12 + m] = 1; }
    halo_wrap(c, f, stg, nks, 18);
  }
  const int hs = 1 + ievf - ie;
  KLAUNCH(c, "k_btcl", k_btcl, grid3(d.ni + 2 * hs + 1, d.nj + 2 * hs + 1, 1, b), b, d, tmp, work + W_BTCu * slab,
                     work + W_BTCv * slab, hs);
  if (add_uh0) KLAUNCH(c, "k_uhbt0", k_uhbt0, grid3(d.ni + 1, d.nj + 1, 1, b), b, d, work, work);

  KLAUNCH(c, "k_find_Cor", k_find_Cor, grid3(ievf - isvf + 3, jevf - jsvf + 3, 1, b), b, d, work, P.Sadourny, isvf, ievf, jsvf, jevf);
  KLAUNCH(c, "k_cor_ref_eta_src", k_cor_ref_eta_src, grid3(d.ni + 1, d.nj + 1, 1, b), b, d, c->G, work, s->eta_cor, Instep,
                     P.bound_BT_corr ? ((BT_cont && P.BT_cont_bounds) ? 1 : 2) : 0, P.maxCFL_BT_cont * Idt, dt, c->GV.Z_to_H, P.Z_ref, P.maxvel);
  {
    std::vector<double *> f = { work + W_eta_PF * slab, work + W_eta_src * slab, work + W_bt_rem_u * slab, work + W_bt_rem_v * slab,
                                work + W_BT_force_u * slab, work + W_BT_force_v * slab };
    std::vector<int> stg = { 0, 0, 1, 2, 1, 2 };
    if (add_uh0) { f.push_back(work + W_uhbt0 * slab); stg.push_back(1); f.push_back(work + W_vhbt0 * slab); stg.push_back(2); }
    f.push_back(work + W_Cor_ref_u * slab); stg.push_back(1); f.push_back(work + W_Cor_ref_v * slab); stg.push_back(2);
    std::vector<int> nks(f.size(), 1);
    halo_wrap(c, f.data(), stg.data(), nks.data(), (int)f.size());
  }

  // ---- filter weights :1726-1795 (host, 1-based)
  double dt_filt;
  if (P.dt_bt_filter >= 0.0) dt_filt = 0.5 * fmax(0.0, fmin(P.dt_bt_filter, 2.0 * dt));
  else dt_filt = 0.5 * fmax(0.0, dt * fmin(-P.dt_bt_filter, 2.0));
  const int nfilter = (int)ceil(dt_filt / dtbt);
  const int nt = nstep + nfilter;
  REQUIRE(nt > 0, MOM6X_EINVAL, "btstep: number of barotropic step (nstep+nfilter) is 0");
  std::vector<double> wt_vel(nt + 2, 0.0), wt_eta(nt + 2, 0.0), wt_trans(nt + 2, 0.0), wt_accel(nt + 2, 0.0), wt_accel2(nt + 2, 0.0);
  double sum_wt_vel = 0.0, sum_wt_eta = 0.0, sum_wt_accel = 0.0, sum_wt_trans = 0.0;
  for (int n = 1; n <= nt; n++) {
    if ((n == nstep) || (dt_filt - abs(n - nstep) * dtbt >= 0.0)) { wt_vel[n] = 1.0; wt_eta[n] = 1.0; }
    else if (dtbt + dt_filt - abs(n - nstep) * dtbt > 0.0) { wt_vel[n] = 1.0 + (dt_filt / dtbt) - abs(n - nstep); wt_eta[n] = wt_vel[n]; }
    else { wt_vel[n] = 0.0; wt_eta[n] = 0.0; }
    sum_wt_vel = sum_wt_vel + wt_vel[n]; sum_wt_eta = sum_wt_eta + wt_eta[n];
  }
  for (int n = nt; n >= 1; n--) {
    wt_trans[n] = wt_trans[n + 1] + wt_eta[n];
    wt_accel[n] = wt_accel[n + 1] + wt_vel[n];
    sum_wt_accel = sum_wt_accel + wt_accel[n]; sum_wt_trans = sum_wt_trans + wt_trans[n];
  }
  const double I_sum_wt_vel = 1.0 / sum_wt_vel, I_sum_wt_accel = 1.0 / sum_wt_accel;
  const double I_sum_wt_eta = 1.0 / sum_wt_eta, I_sum_wt_trans = 1.0 / sum_wt_trans;
  for (int n = 1; n <= nt; n++) {
    wt_vel[n] = wt_vel[n] * I_sum_wt_vel;
    wt_accel2[n] = wt_accel[n] * I_sum_wt_accel;
    wt_trans[n] = wt_trans[n] * I_sum_wt_trans;
    wt_accel[n] = wt_accel[n] * I_sum_wt_accel;
    wt_eta[n] = wt_eta[n] * I_sum_wt_eta;
  }

  // ---- the time loop :2175-2834
  HIPCHK(hipMemsetAsync(s->ubtav, 0, slab * sizeof(double), st));
  HIPCHK(hipMemsetAsync(s->vbtav, 0, slab * sizeof(double), st));
  HIPCHK(hipMemsetAsync(uhbtav, 0, slab * sizeof(double), st));
  HIPCHK(hipMemsetAsync(vhbtav, 0, slab * sizeof(double), st));
  LoopArgs L;
  memset(&L, 0, sizeof(L));
  L.dtbt = dtbt; L.dgeo_de = dgeo_de; L.vel_underflow = P.vel_underflow;
  if (P.BT_project_velocity) { L.trans_wt1 = (1.0 + P.bebt); L.trans_wt2 = -P.bebt; }
  else { L.trans_wt1 = P.bebt; L.trans_wt2 = (1.0 - P.bebt); }
  L.project = P.BT_project_velocity; L.find_etaav = (etaav != nullptr);
  L.f4_on_the_fly = 1; L.Sadourny = P.Sadourny;
  int isv = is, iev = ie, jsv = js, jev = je;
  // The eta predictor of sub-step n + 1 needs find_uhbt at the velocities sub-step n has just made: the velocity stages
  // evaluate it while they hold the velocity and its fit planes and pass two planes on (W_uhn, W_vhn; exchanged with the
  // velocities), instead of the predictor re-reading four velocities and up to sixteen fit planes per cell.  Not with
  // CLIP_BT_VELOCITY (the velocities change in between) or BT_PROJECT_VELOCITY (the predictor does not use transports).
  const bool pass_uhn = !P.clip_velocity && !P.BT_project_velocity && !evolving_face_areas;   // (evolving: the next predictor uses NEW face areas)
  L.store_uhn = pass_uhn ? 1 : 0;
  // ... and when no exchange separates two sub-steps (three times out of four with a halo of 4), the eta corrector of the
  // first forms the predictor of the second on the way: same points, the new eta still in a register.
  // One launch per sub-step (k_bt_substep) wherever that predictor chain holds; MOM6X_BT_SUBSTEP=kernels keeps the three launches.
  // Measured (profiles/README.md, r04): on the 360 x 540 tile of an 8-GPU layout the one-launch form wins (the step 9.14 -> 9.01 ms:
  // its kernels are latency-, not bandwidth-bound there); at 1440 x 1080 the three kernels win (6.6 against 7.4 ms per step: the
  // frame a block recomputes, 20 % of its threads, costs more than the three round trips of ubt, vbt, uhbt save).  So: by size.
  // (Round 5 built TWO sub-steps per launch on the large tile -- a tile with a 2-3 point frame, the state of the first sub-step handed to the
  //  second through LDS, the ~34 time-invariant words per column read once per two sub-steps, the running sums read and written once --
  //  in two forms, coefficients on demand and all of them in registers: bit-identical, 7.3 / 7.4 ms per step against 6.6 for the three
  //  kernels at 1440 x 1080 (profiles/r05_btpair.txt, r05_btpair2.txt; the kernel is in the history: k_bt_substep2).  Removed.)
  static const int substep_env = [] { const char *e = getenv("MOM6X_BT_SUBSTEP"); return !e ? 0 : (!strcmp(e, "kernels") ? 1 : (!strcmp(e, "fused") ? 2 : 0)); }();
  const bool small_tile = ((long)d.ni * d.nj <= 512L * 1024L);
  const bool fused = pass_uhn && (substep_env == 2 || (substep_env == 0 && small_tile));
  // the loop's group pass overlapped with the own-points half of the sub-step that follows it (the one-launch form, tiles with room
  // for such a half: more than two blocks each way)
  const bool overlap_loop = fused && c->bt_overlap && halo_can_overlap(c) && d.ni >= 96 && d.nj >= 32;
  SubPlanes SP = { W_ubt, fused ? W_ubt2 : W_ubt, W_vbt, fused ? W_vbt2 : W_vbt, W_eta_pred, fused ? W_eta_pred2 : W_eta_pred };
  const int loop_stg[] = { 0, 1, 2, 1, 2 }, loop_nk[] = { 1, 1, 1, 1, 1 };
  bool pred_done = false;
  for (int n = 1; n <= nt; n++) {
    if (P.clip_velocity)
      KLAUNCH(c, "k_bt_clip", k_bt_clip, grid3(iev - isv + 2, jev - jsv + 2, 1, b), b, d, c->G, work, dt, P.CFL_trunc, isv, iev, jsv, jev);
    L.have_uhn = (pass_uhn && n > 1) ? 1 : 0;
    bool travelling = false;   // the group pass of the state is on the second stream: this sub-step runs in two halves around it
    if ((iev - stencil < ie) || (jev - stencil < je)) {
      double *loop_f[] = { work + W_eta * slab, work + (size_t)SP.u_in * slab, work + (size_t)SP.v_in * slab, work + W_uhn * slab,
                           work + W_vhn * slab };
      // MOM_barotropic.F90:2505-2512.  With a communicator and the one-launch sub-step the pass overlaps the sub-step's own-points half
      // (pack on this stream -- the half rewrites eta next to the edge --, messages and unpack on the second one)
      if (overlap_loop) travelling = halo_start_packed(c, loop_f, loop_stg, loop_nk, (pass_uhn && n > 1) ? 5 : 3);
      else halo_wrap(c, loop_f, loop_stg, loop_nk, (pass_uhn && n > 1) ? 5 : 3);
      isv = isvf; iev = ievf; jsv = jsvf; jev = jevf;
    } else {
      isv += stencil; iev -= stencil; jsv += stencil; jev -= stencil;
    }
    if (evolving_face_areas && n > 1 && ((n - 1) % P.nonlin_cont_update_period == 0))   // :2539-2543
      KLAUNCH(c, "k_face_areas_eta", k_face_areas_eta, grid3(iev - isv + 4, jev - jsv + 4, 1, b), b, d, c->G, c->GV.Z_to_H,
              (const double *)(work + W_eta * slab), work + W_BTCu * slab, work + W_BTCv * slab, isv, iev, jsv, jev);
    L.isv = isv; L.iev = iev; L.jsv = jsv; L.jev = jev;
    L.wt_accel = wt_accel[n]; L.wt_trans = wt_trans[n]; L.wt_vel = wt_vel[n]; L.wt_eta = wt_eta[n]; L.wt_accel2 = wt_accel2[n];
    const bool do_pred = (!P.BT_project_velocity || L.find_etaav) && !pred_done;
    if (do_pred)
      KLAUNCH(c, "k_bt_pred", k_bt_pred, grid3(nxa(iev - isv + 3, isv - 1), jev - jsv + 3, 1, b), b, d, c->G, work, L, SP.u_in, SP.v_in, SP.e_in,
              travelling ? 1 : 0);
    // the next sub-step: no exchange before it (:2505-2512) and its transports passed on by this one's velocity stages
    pred_done = pass_uhn && n < nt && !((iev - stencil < ie) || (jev - stencil < je));
    L.pred_next = pred_done ? 1 : 0;
    L.wt_accel2_next = pred_done ? wt_accel2[n + 1] : 0.0;
    const bool v_first = (((n + c->first_direction) % 2) == 1);
    if (fused) {
      // (work-groups of 32 x 8; 32 x 16, 16 x 16 and 64 x 4 were measured in round 4 -- 9.01 / 9.03 / 9.20 / 9.14 ms per step on the 8-GPU tile -- and are gone)
#define SUBSTEP(BX, BY, SEL) do {                                                                                                     \
      const int OX = v_first ? BX - 2 : BX - 1, OY = v_first ? BY - 1 : BY - 2;                                                        \
      const int nbx = (iev - isv + OX) / OX, nby = (jev - jsv + OY) / OY;                                                              \
      if (v_first)                                                                                                                    \
        KLAUNCH(c, "k_bt_substep<v>", (k_bt_substep<true, BX, BY>), dim3(nbx * nby), dim3(BX, BY), d, c->G, work, s->ubtav, uhbtav, s->vbtav, \
                vhbtav, L, SP, 0, c->GV.Z_to_H, s->warn, s->warn_info, nbx, SEL);                                                      \
      else                                                                                                                            \
        KLAUNCH(c, "k_bt_substep<u>", (k_bt_substep<false, BX, BY>), dim3(nbx * nby), dim3(BX, BY), d, c->G, work, s->ubtav, uhbtav, s->vbtav, \
                vhbtav, L, SP, P.use_old_coriolis_bracket_bug, c->GV.Z_to_H, s->warn, s->warn_info, nbx, SEL);                         \
      } while (0)
      if (travelling) {
        SUBSTEP(32, 8, 1);          // the blocks that read the tile's own points only, while the messages travel ...
        halo_complete(c);
        if (do_pred)
          KLAUNCH(c, "k_bt_pred", k_bt_pred, grid3(nxa(iev - isv + 3, isv - 1), jev - jsv + 3, 1, b), b, d, c->G, work, L, SP.u_in, SP.v_in, SP.e_in, 2);
        SUBSTEP(32, 8, 2);          // ... and the rest
      } else {
        SUBSTEP(32, 8, 0);
      }
#undef SUBSTEP
      std::swap(SP.u_in, SP.u_out); std::swap(SP.v_in, SP.v_out);
      if (pred_done) std::swap(SP.e_in, SP.e_out);
      continue;
    }
    if (v_first) {
      KLAUNCH(c, "k_bt_vel<1>", k_bt_vel<1>, grid3(nxa(iev - isv + 3, isv - 1), jev - jsv + 2, 1, b), b, d, c->G, work, s->vbtav, vhbtav, L,
                         isv - 1, iev + 1, jsv - 1, jev, 0);
      KLAUNCH(c, "k_bt_vel<0>", k_bt_vel<0>, grid3(nxa(iev - isv + 2, isv - 1), jev - jsv + 1, 1, b), b, d, c->G, work, s->ubtav, uhbtav, L,
                         isv - 1, iev, jsv, jev, 0);
    } else {
      KLAUNCH(c, "k_bt_vel<0>", k_bt_vel<0>, grid3(nxa(iev - isv + 2, isv - 1), jev - jsv + 3, 1, b), b, d, c->G, work, s->ubtav, uhbtav, L,
                         isv - 1, iev, jsv - 1, jev + 1, 0);
      KLAUNCH(c, "k_bt_vel<1>", k_bt_vel<1>, grid3(nxa(iev - isv + 1, isv), jev - jsv + 2, 1, b), b, d, c->G, work, s->vbtav, vhbtav, L,
                         isv, iev, jsv - 1, jev, P.use_old_coriolis_bracket_bug);
    }
    KLAUNCH(c, "k_bt_eta", k_bt_eta, grid3(nxa(iev - isv + 1, isv), jev - jsv + 1, 1, b), b, d, c->G, work, L, c->GV.Z_to_H, s->warn, s->warn_info);
  }

  // ---- after the loop
  KLAUNCH(c, "k_bt_post", k_bt_post, grid3(d.ni, d.nj, 1, b), b, d, work, eta_in, eta_out, etaav, dgeo_de, 1.0, 1.0);
  {
    std::vector<double *> f; std::vector<int> stg;
    if (etaav) { f.push_back(etaav); stg.push_back(0); }
    f.push_back(work + W_e_anom * slab); stg.push_back(0);
    f.push_back(s->ubtav); stg.push_back(1); f.push_back(s->vbtav); stg.push_back(2);
    f.push_back(uhbtav); stg.push_back(1); f.push_back(vhbtav); stg.push_back(2);
    std::vector<int> nks(f.size(), 1);
    halo_wrap(c, f.data(), stg.data(), nks.data(), (int)f.size());
  }
  // btstep_layer_accel :3432-3504.  Inside the RK2 step the consumer of accel_layer_u / _v (the velocity estimate vertvisc_coef
  // forms) evaluates it from pbce and the 2-D results left in the work block: 3 words per cell-layer less, twice per step.
  s->la_pending = false;
  if (s->la_defer) { s->la_pending = true; s->la_pbce = pbce; s->la_underflow = P.vel_underflow * Idt; }
  else
    KLAUNCH(c, "k_layer_accel", k_layer_accel, grid3(nxa(d.ni + 1, -1), d.nj + 1, nchunks(d.nk), b), b, d, c->G, work, pbce, accel_layer_u,
                       accel_layer_v, P.vel_underflow * Idt);
  HIPCHK(hipGetLastError());
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  return MOM6X_OK;
}
