// continuity_wave.hip -- the mass-flux half of continuity_PPM with ONE WAVEFRONT ROW PER FACE COLUMN.
//
// The kernel of mom6x_continuity_params.sum_order == MOM6X_SUM_TREE16 (include/mom6x.h).  Same routines as
// continuity_lds.hip (PPM_reconstruction_x/y :2307/:2442, zonal/meridional_mass_flux :519/:1412 with
// zonal/meridional_flux_adjust :1093/:1992 and set_zonal/merid_BT_cont :1246/:2143, zonal/merid_flux_thickness
// :975/:1866 of MOM_continuity_PPM.F90), different ownership:
//
//   * a wavefront = 4 face columns x 16 layer lanes; lane (f, q) owns the layers k = q, q+16, q+32, ... of face f
//     and keeps EVERYTHING of them in registers: u, visc_rem, the PPM edge values and curvatures of the face's two
//     cells, the last layer transports.  No LDS, no work-group barrier, no second launch.
//   * a column sum is the lane's own partial sum (increasing k) followed by a 4-step butterfly over the 16 lanes of
//     the face's DPP row (quad_perm, quad_perm, row_half_mirror, row_mirror): two v_mov_dpp + one v_add_f64 per step.
//     Addition is commutative bit for bit, so every lane of the row ends with the SAME bits -- the balanced tree over
//     q = 0..15 that oracle/orc_continuity.c::tree16_sum restates -- and the Newton state of flux_adjust is simply
//     replicated over the row instead of being owned by a "face lane".
//   * the Newton loop is wavefront-uniform over 4 faces (the LDS kernel's is work-group-uniform over 16): a face whose
//     do_I is false is frozen exactly as in the reference's row-wide loop.
//   * the CFL limits of du (:646-723) use the cheap-bounds-first scheme of the LDS kernel; the exact k-recurrence, when
//     a Newton step comes within reach of a bound, is walked by the wavefront itself through row broadcasts.
//   * duL / duR of set_*_BT_cont (:1293-1316) are the min / max of the quotients, which is what the reference's
//     recurrence computes in exact arithmetic (sum_order's definition).
// The result of a face depends on that face's column only (never on its tile or wavefront), so runs on different
// tile layouts stay bit-identical.  Against the reference's sequential order the results agree to round-off; against
// the oracle run with the same sum_order they are bit-identical (tests/test_continuity_gpu.py).
#include "continuity_dev.h"
#include "continuity_lds.h"
#include <cstdlib>

namespace {

constexpr int NF = 16;   // faces along i per work-group (4 wavefronts x 4 faces): one 128-byte line per row segment
constexpr int KL = 16;   // layer lanes per face = one DPP row

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Butterflies over the 16 lanes of a DPP row.  Must be called with whole rows active.
#define DPP_XOR1 0xB1    // quad_perm:[1,0,3,2]
#define DPP_XOR2 0x4E    // quad_perm:[2,3,0,1]
#define DPP_HMIR 0x141   // row_half_mirror: after the two quad steps every lane of a quad holds the quad's value
#define DPP_MIR 0x140    // row_mirror
__device__ __forceinline__ double row_sum(double s) {
  s = s + dpp_mov<DPP_XOR1>(s);
  s = s + dpp_mov<DPP_XOR2>(s);
  s = s + dpp_mov<DPP_HMIR>(s);
  s = s + dpp_mov<DPP_MIR>(s);
  return s;
}
__device__ __forceinline__ double row_min(double s) {
  s = dmin(s, dpp_mov<DPP_XOR1>(s));
  s = dmin(s, dpp_mov<DPP_XOR2>(s));
  s = dmin(s, dpp_mov<DPP_HMIR>(s));
  s = dmin(s, dpp_mov<DPP_MIR>(s));
  return s;
}
__device__ __forceinline__ double row_max(double s) {
  s = dmax(s, dpp_mov<DPP_XOR1>(s));
  s = dmax(s, dpp_mov<DPP_XOR2>(s));
  s = dmax(s, dpp_mov<DPP_HMIR>(s));
  s = dmax(s, dpp_mov<DPP_MIR>(s));
  return s;
}
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// What a lane keeps of its MAXL layers of one face column.
template <int MAXL>
struct Col {
  double u[MAXL], v[MAXL];                      // u, visc_rem (0, 0 beyond nk: such a layer transports exactly +0.0)
  double mL[MAXL], mR[MAXL], mC[MAXL];          // minus cell: h_L, h_R, curv_3 = (h_L + h_R) - 2 h
  double pL[MAXL], pR[MAXL], pC[MAXL];          // plus cell
  double dt, IdT_m, IdT_p, Lf;
};

// zonal_flux_layer :896 / merid_flux_layer :1787 from registers.  The two upwind branches of the reference are
// mirror images (a = the edge value on the face side of the upwind cell, b = the far one); operands are selected
// instead of branching so that a wavefront with mixed flow directions does not run both.
template <int MAXL>
__device__ __forceinline__ void flux_reg(const Col<MAXL> &C, int n, double u, double &uh, double &duhdu) {
  const bool pos = (u > 0.0);
  const double a = pos ? C.mR[n] : C.pL[n], b = pos ? C.mL[n] : C.pR[n], curv_3 = pos ? C.mC[n] : C.pC[n];
  const double CFL = fabs(u) * C.dt * (pos ? C.IdT_m : C.IdT_p);
  const double uh_m = C.Lf * u * (a + CFL * (0.5 * (b - a) + curv_3 * (CFL - 1.5)));
  const double hm_m = a + CFL * ((b - a) + 3.0 * curv_3 * (CFL - 1.0));
  const bool moving = (u != 0.0);
  uh = moving ? uh_m : 0.0;
  const double h_marg = moving ? hm_m : 0.5 * (C.pL[n] + C.mR[n]);
  duhdu = C.Lf * h_marg * C.v[n];
}

// zonal_flux_adjust :1093-1242 / meridional_flux_adjust :1992-2140, iterated wavefront-uniformly; the Newton state is
// replicated over the 16 lanes of a face's row.  STORE: keep the last evaluated transports in uh_r (the reference's
// uh_3d argument).  `lazy`: du_max / du_min are a lower / an upper bound of the CFL limits; the first test they do not
// decide ends the solve with need_exact = true (wavefront-uniform) and the caller repeats it with the limits.
template <int MAXL, bool STORE>
__device__ __forceinline__ double wave_flux_adjust(const Col<MAXL> &C, bool active, double IareaMin, double uhbt,
                                                   double uh_tot_0, double duhdu_tot_0, double du_max, double du_min,
                                                   double tol_eta_cs, double tol_vel, int better_iter, bool lazy,
                                                   bool &need_exact, double (&uh_r)[MAXL]) {
  const int max_itts = 20;
  double du = 0.0;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0;
  double uh_err_best = fabs(uh_err);
  bool do_I = active;
  bool max_lazy = lazy, min_lazy = lazy, undecided = false;
  need_exact = false;
  for (int itt = 1; itt <= max_itts; itt++) {
    if (do_I) {
      double tol_eta;
      if (itt <= 1) tol_eta = 1e-6 * tol_eta_cs;
      else if (itt == 2) tol_eta = 1e-4 * tol_eta_cs;
      else if (itt == 3) tol_eta = 1e-2 * tol_eta_cs;
      else tol_eta = tol_eta_cs;

      if (uh_err > 0.0) { du_max = du; max_lazy = false; }
      else if (uh_err < 0.0) { du_min = du; min_lazy = false; }
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
            if (max_lazy && !(du < du_max)) undecided = true;
            if (du >= du_max) {
              du = 0.5 * (du_prev + du_max);
              if (du_max - du_prev < 1.0e-15 * fabs(du)) do_I = false;
            }
          } else {
            if (min_lazy && !(du > du_min)) undecided = true;
            if (du <= du_min) {
              du = 0.5 * (du_prev + du_min);
              if (du_prev - du_min < 1.0e-15 * fabs(du)) do_I = false;
            }
          }
        } else {
          do_I = false;
        }
      }
    }
    if (wave_any(undecided)) { need_exact = true; break; }
    if (!wave_any(do_I)) break;

    if ((itt < max_itts) || STORE) {
      double s_uh = 0.0, s_dd = 0.0;
#pragma unroll
      for (int n = 0; n < MAXL; n++) {
        double uh, dd;
        flux_reg(C, n, C.u[n] + du * C.v[n], uh, dd);
        if (STORE) uh_r[n] = do_I ? uh : uh_r[n];
        s_uh = s_uh + uh; s_dd = s_dd + dd;
      }
      if (itt < max_itts) {
        const double err = row_sum(s_uh) - uhbt, dtot = row_sum(s_dd);
        if (do_I) {
          uh_err = err; duhdu_tot = dtot;
          uh_err_best = dmin(uh_err_best, fabs(uh_err));
        }
      }
    }
  }
  return du;
}

template <int DIR, int MAXL>
__global__ void __launch_bounds__(NF * KL, (MAXL > 5) ? 1 : 2)
k_mass_flux_wave(Dm d, const double *__restrict__ G, FluxArgs A, LdsArgs E) {
  // Tile <-> block id as in continuity_lds.hip: block b runs on XCD b % 8 (observed); every XCD gets a band of
  // E.rows tile rows and walks it along i (zonal) or along j (meridional: neighbours share 5 of their 6 rows of h).
  const int tile = blockIdx.x;
  const int band = tile & 7, slot = tile >> 3;
  const int bx = DIR ? slot / E.rows : slot % E.gx;
  const int by = band * E.rows + (DIR ? slot % E.rows : slot / E.gx);
  if (by >= E.gy) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int fl = (tid >> 6) * 4 + (lane >> 4), kl = lane & 15;
  // tiles start on 128-byte lines of the pitched rows: the four wavefronts of a work-group read and write the four
  // quarters of the same lines at about the same time
  const int i = E.i_base + bx * NF + fl, j = A.b0 + by;
  const bool active = (i >= A.a0 && i <= A.a1);
  if (!wave_any(active)) return;
  const int nk = d.nk;
  const int st = DIR ? d.pitch : 1;
  const size_t slab = (size_t)d.slab;
  const DirMetrics D = dir_metrics<DIR>(G, d);
  const size_t f2 = ix2(d, active ? i : A.a1, j);   // inactive lanes alias the last face (never written)
  const double dt = A.dt;
  const bool use_visc_rem = (A.visc_rem != nullptr);
  const bool need_adjust = (A.uhbt != nullptr) || A.set_BT_cont;

  Col<MAXL> C;
  C.dt = dt; C.IdT_m = D.IdT[f2]; C.IdT_p = D.IdT[f2 + st];
  C.Lf = D.Lface[f2] * 1.0;   // G%dy_Cu * por_face_areaU (== 1)

  // ---- loads: u, visc_rem and the six h values (cells f-2 .. f+3 along the sweep direction) of every layer --------
  {
    double m6[6];
#pragma unroll
    for (int q = 0; q < 6; q++) m6[q] = active ? D.mask2dT[f2 + (size_t)(q - 2) * st] : 0.0;
    double h6[MAXL][6];
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      const bool on = active && (k < nk);
      const size_t f = f2 + (size_t)(on ? k : 0) * slab;
      C.u[n] = on ? A.u[f] : 0.0;
      C.v[n] = on ? (use_visc_rem ? A.visc_rem[f] : 1.0) : 0.0;
#pragma unroll
      for (int q = 0; q < 6; q++) h6[n][q] = on ? A.h_in[f + (size_t)(q - 2) * st] : 0.0;
    }
    // PPM_reconstruction + limiter of the face's two cells (their stencils share four values and two slopes)
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const bool on = active && (kl + KL * n < nk);
      double hl = 0.0, hr = 0.0, c3 = 0.0;
      if (on) edge5(&h6[n][0], &m6[0], E.scheme, E.monotonic, E.h_min, hl, hr, c3);
      C.mL[n] = hl; C.mR[n] = hr; C.mC[n] = c3;
      hl = 0.0; hr = 0.0; c3 = 0.0;
      if (on) edge5(&h6[n][1], &m6[1], E.scheme, E.monotonic, E.h_min, hl, hr, c3);
      C.pL[n] = hl; C.pR[n] = hr; C.pC[n] = c3;
    }
  }
  // every other global load of the kernel is issued here as well (the memory counter is in order)
  const double IareaMin = dmin(D.IareaT[f2], D.IareaT[f2 + st]);
  const double uhbt_f = (A.uhbt != nullptr && active) ? A.uhbt[f2] : 0.0;
  const double dC_f = A.set_BT_cont ? D.dC[f2] : 0.0;
  const double dx_W = D.dT[f2], dx_E = D.dT[f2 + st];
  const double maskC = D.maskC[f2];

  // ---- limits on du that keep the CFL number between -1 and 1 (:646-723) --------------------------------------
  double visc_rem_max = 1.0, du_max_CFL = 0.0, du_min_CFL = 0.0;
  const double CFL_dt = A.CFL_limit_adjust / dt;
  bool lazy = false;
  if (need_adjust) {
    if (use_visc_rem && A.use_visc_rem_max) {
      double pm = 0.0;
#pragma unroll
      for (int n = 0; n < MAXL; n++)
        if (kl + KL * n < nk) pm = dmax(pm, C.v[n]);
      visc_rem_max = row_max(pm);
    }
  }
  double I_vrm = 0.0;
  if (visc_rem_max > 0.0) I_vrm = 1.0 / visc_rem_max;
  // The exact limits with visc_rem: a k-recurrence (two divisions per layer), walked by the wavefront: the lane that
  // owns layer k broadcasts its operands to the row.  Wavefront-uniform; rarely needed.
  auto exact_bounds = [&]() {
    du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
    du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const double q_max = (dx_W * CFL_dt - C.u[n]) / C.v[n];
      const double q_min = -(dx_E * CFL_dt + C.u[n]) / C.v[n];
      for (int q = 0; q < KL; q++) {
        if (q + KL * n >= nk) break;
        const double uk = __shfl(C.u[n], q, KL), vrem = __shfl(C.v[n], q, KL);
        const double qa = __shfl(q_max, q, KL), qi = __shfl(q_min, q, KL);
        if (du_max_CFL * vrem > dx_W * CFL_dt - uk * maskC) du_max_CFL = qa;
        if (du_min_CFL * vrem < -dx_E * CFL_dt - uk * maskC) du_min_CFL = qi;
      }
    }
    du_max_CFL = dmax(du_max_CFL, 0.0);
    du_min_CFL = dmin(du_min_CFL, 0.0);
  };
  if (need_adjust) {
    // min_k (dx_W*CFL_dt - u_k) and min_k (dx_E*CFL_dt + u_k): order-independent, exact
    double nmin = 1.0e300, mmin = 1.0e300;
    bool vr_ok = true;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      if (kl + KL * n < nk) {
        nmin = dmin(nmin, dx_W * CFL_dt - C.u[n]);
        mmin = dmin(mmin, dx_E * CFL_dt + C.u[n]);
        if (!(C.v[n] >= 0.0 && C.v[n] <= 1.0 + 1.0e-10)) vr_ok = false;   // (visc_rem can exceed 1 by round-off)
      }
    }
    if (!vr_ok) { nmin = -1.0; mmin = -1.0; }
    nmin = row_min(nmin); mmin = row_min(mmin);
    const double D0max = 2.0 * (CFL_dt * dx_W) * I_vrm, D0min = -2.0 * (CFL_dt * dx_E) * I_vrm;
    if (!use_visc_rem) {
      // :709-716: plain min / max chains, exact in any order
      du_max_CFL = dmax(dmin(D0max, nmin), 0.0);
      du_min_CFL = dmin(dmax(D0min, -mmin), 0.0);
    } else {
      // With visc_rem the recurrence leaves du_max_CFL equal to D0max or to one of q_k = (dx_W*CFL_dt - u_k)/visc_rem_k,
      // and q_k >= (1 - 1e-10)*(dx_W*CFL_dt - u_k) whenever that is >= 0 and 0 <= visc_rem_k <= 1 + 1e-10: so
      // (1 - 1e-9)*min(D0max, nmin) is a lower bound of du_max_CFL, and likewise for du_min_CFL from above.
      const double shrink = 1.0 - 1.0e-9;
      du_max_CFL = (nmin >= 0.0) ? shrink * dmax(dmin(D0max, nmin), 0.0) : -1.0e300;
      du_min_CFL = (mmin >= 0.0) ? shrink * dmin(dmax(D0min, -mmin), 0.0) : 1.0e300;
      lazy = true;
    }
  }

  // ---- first sweep: layer transports and their column sums (:615-668) -------------------------------------------
  double uh_r[MAXL];
  double uh_tot_0 = 0.0, duhdu_tot_0 = 0.0;
  auto first_sweep = [&]() {
    double s_uh = 0.0, s_dd = 0.0;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double dd;
      flux_reg(C, n, C.u[n], uh_r[n], dd);
      s_uh = s_uh + uh_r[n]; s_dd = s_dd + dd;
    }
    if (need_adjust) { uh_tot_0 = row_sum(s_uh); duhdu_tot_0 = row_sum(s_dd); }
  };
  first_sweep();

  // ---- flux_adjust towards uhbt; uh, u_cor, du_cor ---------------------------------------------------------------
  double du_fin = 0.0;
  const bool corrected = (A.uhbt != nullptr);
  if (corrected) {
    bool redo;
    du_fin = wave_flux_adjust<MAXL, true>(C, active, IareaMin, uhbt_f, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                          A.tol_eta, A.tol_vel, A.better_iter, lazy, redo, uh_r);
    if (redo) {   // wavefront-uniform
      exact_bounds(); lazy = false;
      first_sweep();   // (the abandoned solve has overwritten some of the first transports)
      du_fin = wave_flux_adjust<MAXL, true>(C, active, IareaMin, uhbt_f, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                            A.tol_eta, A.tol_vel, A.better_iter, false, redo, uh_r);
    }
    if (active && kl == 0 && A.du_cor) A.du_cor[f2] = du_fin;
  }
  if (active) {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        const size_t f = f2 + (size_t)k * slab;
        A.uh[f] = uh_r[n];
        if (corrected && A.u_cor) A.u_cor[f] = C.u[n] + du_fin * C.v[n];
      }
    }
  }

  // ---- zonal/merid_flux_thickness (:975 / :1866) at the corrected velocities ------------------------------------
  if (E.h_face && active) {
    const bool use_cor = corrected && (A.u_cor != nullptr);
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const int k = kl + KL * n;
      if (k < nk) {
        const double uf = use_cor ? (C.u[n] + du_fin * C.v[n]) : C.u[n];
        const bool pos = (uf > 0.0);
        const double a = pos ? C.mR[n] : C.pL[n], b = pos ? C.mL[n] : C.pR[n], curv_3 = pos ? C.mC[n] : C.pC[n];
        const double CFL = fabs(uf) * dt * (pos ? C.IdT_m : C.IdT_p);
        double h_avg = a + CFL * (0.5 * (b - a) + curv_3 * (CFL - 1.5));
        double h_marg = a + CFL * ((b - a) + 3.0 * curv_3 * (CFL - 1.0));
        if (uf == 0.0) { h_avg = 0.5 * (C.pL[n] + C.mR[n]); h_marg = h_avg; }
        double hu = E.marginal ? h_marg : h_avg;
        if (use_visc_rem) hu = hu * (C.v[n] * 1.0);
        else hu = hu * 1.0;
        E.h_face[f2 + (size_t)k * slab] = hu;
      }
    }
  }
  if (!A.set_BT_cont) return;

  // ---- set_zonal_BT_cont :1246-1409 / set_merid_BT_cont :2143-2304 -----------------------------------------------
  const double Idt = 1.0 / dt, min_visc_rem = 0.1, CFL_min = 1e-6;
  double du0;
  {
    bool redo;
    double dummy[MAXL];
    du0 = wave_flux_adjust<MAXL, false>(C, active, IareaMin, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                        A.tol_eta, A.tol_vel, A.better_iter, lazy, redo, dummy);
    if (redo) {
      exact_bounds(); lazy = false;
      du0 = wave_flux_adjust<MAXL, false>(C, active, IareaMin, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                          A.tol_eta, A.tol_vel, A.better_iter, false, redo, dummy);
    }
  }
  const double du_CFL = (CFL_min * Idt) * dC_f;
  // duR / duL (:1293-1316): min / max of the quotients (sum_order TREE16's definition)
  double duR = dmin(0.0, du0 - du_CFL), duL = dmax(0.0, du0 + du_CFL);
  {
    const double vrl_floor = min_visc_rem * visc_rem_max;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      if (kl + KL * n < nk) {
        const double uk = C.u[n], vrem = C.v[n];
        const double visc_rem_lim = dmax(vrem, vrl_floor);
        if (visc_rem_lim > 0.0) {
          duR = dmin(duR, -(uk + du_CFL * vrem) / visc_rem_lim);
          duL = dmax(duL, -(uk - du_CFL * vrem) / visc_rem_lim);
        }
      }
    }
    duR = row_min(duR); duL = row_max(duL);
  }
  // three trial velocities (:1330-1349), five column sums
  double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    double uh_0, uh_L, uh_R, d_0, d_L, d_R;
    flux_reg(C, n, C.u[n] + du0 * C.v[n], uh_0, d_0);
    flux_reg(C, n, C.u[n] + duL * C.v[n], uh_L, d_L);
    flux_reg(C, n, C.u[n] + duR * C.v[n], uh_R, d_R);
    FAmt_0 = FAmt_0 + d_0; FAmt_L = FAmt_L + d_L; FAmt_R = FAmt_R + d_R;
    uhtot_L = uhtot_L + uh_L; uhtot_R = uhtot_R + uh_R;
  }
  FAmt_0 = row_sum(FAmt_0); FAmt_L = row_sum(FAmt_L); FAmt_R = row_sum(FAmt_R);
  uhtot_L = row_sum(uhtot_L); uhtot_R = row_sum(uhtot_R);
  if (!(active && kl == 0)) return;

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

template <int DIR, int MAXL>
int launch(mom6x_ctx *c, const FluxArgs &A, const LdsArgs &E0) {
  LdsArgs E = E0;
  E.gx = (A.a1 - E.i_base + NF) / NF; E.gy = A.b1 - A.b0 + 1;
  E.rows = (E.gy + 7) / 8;
  E.retry = nullptr; E.force_walk = 0;
  const dim3 grid(8 * E.gx * E.rows, 1, 1);
  if (c->prof_on) prof_begin(c, DIR ? "k_mass_flux_wave<1>" : "k_mass_flux_wave<0>");
  hipLaunchKernelGGL((k_mass_flux_wave<DIR, MAXL>), grid, dim3(NF * KL, 1, 1), 0, c->stream, c->d, c->G, A, E);
  if (c->prof_on) prof_end(c);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

}  // namespace

bool mass_flux_wave_usable(int nk) { return nk <= 8 * KL; }

int mass_flux_wave(mom6x_ctx *c, int dir, const FluxArgs &A, const LdsArgs &E0) {
  const int nk = c->d.nk;
  LdsArgs E = E0;
  E.i_base = A.a0 - (((A.a0 + c->d.ioff) % NF) + NF) % NF;   // (i_base + ioff) is a multiple of 16 doubles = 128 B
  const int maxl = (nk + KL - 1) / KL;
#define GO(D, M) return launch<D, M>(c, A, E)
  if (dir == 0) { if (maxl <= 2) GO(0, 2); if (maxl <= 5) GO(0, 5); GO(0, 8); }
  if (maxl <= 2) GO(1, 2); if (maxl <= 5) GO(1, 5); GO(1, 8);
#undef GO
}
