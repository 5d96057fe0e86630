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
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

#ifndef MOM6X_MFW_NF   // (A/B: 8 = two wavefronts per work-group)
#define MOM6X_MFW_NF 16
#endif
constexpr int NF = MOM6X_MFW_NF;   // faces along i per work-group (4 wavefronts x 4 faces): one 128-byte line per row segment
constexpr int KL = 16;   // layer lanes per face = one DPP row

// Dev tool (MOM6X_CFLAGS=-DMOM6X_MFL_TIMING python -m mom6_amd.build --force; scripts/prof_continuity.py): shader-clock
// cycles the first wavefront of every work-group spends in each phase of the kernel, summed over the work-groups.
#ifdef MOM6X_MFL_TIMING
// (the first wavefront of a work-group adds its phase times to 16 words of LDS behind the wavefronts' regions -- ds_add_u64, nothing
// returned, nothing waited for -- and to the global sums once, when its march ends)
__device__ unsigned long long g_mfw_t[2][16];
#define TICK_INIT long long t_prev_ = clock64(); unsigned long long *tk_ = (unsigned long long *)(S_all + (NF / 4) * (size_t)WAVE_LDS); \
  if (threadIdx.x < 16) tk_[threadIdx.x] = 0ull
#define TICK(p) do { if (threadIdx.x == 0) { const long long t_ = clock64(); \
  __hip_atomic_fetch_add(&tk_[p], (unsigned long long)(t_ - t_prev_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); t_prev_ = t_; } } while (0)
#define TICK_FLUSH do { if (threadIdx.x < 16 && tk_[threadIdx.x]) atomicAdd(&g_mfw_t[DIR][threadIdx.x], tk_[threadIdx.x]); } while (0)
#define TICK_PARAM , long long &t_prev_, unsigned long long *tk_
#define TICK_ARG , t_prev_, tk_
#define COUNT_ITT(slot, n) do { if (threadIdx.x == 0) { \
  __hip_atomic_fetch_add(&tk_[slot], (unsigned long long)(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
  __hip_atomic_fetch_add(&tk_[14 + (slot) - 8], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } } while (0)
#else
#define TICK_INIT
#define TICK(p)
#define TICK_FLUSH
#define TICK_PARAM
#define TICK_ARG
#define COUNT_ITT(slot, n)
#endif

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  // (every lane of every row is written -- all four controls below are permutations of a row -- so there is no "old" value to
  // keep: with bound_ctrl and full masks the compiler emits the two v_mov_b32_dpp alone, without a copy of x in front of them:
  // 3 instead of 5 VALU instructions per butterfly step)
  int lo = __double2loint(x), hi = __double2hiint(x);
#ifdef MOM6X_MFW_DPP_OLD   // (A/B: the form of rounds 2-4)
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
#else
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
#endif
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
// Keeps the instruction scheduler from interleaving more than two layers of an unrolled layer loop: every layer in
// flight holds about 24 registers of temporaries, and five of them push the kernel into scratch memory -- whose
// reloads then queue behind the next row's LDS-DMA (the memory counter is in order).  Two layers in flight plus the
// second wavefront of the SIMD cover the latency of a dependent FP64 chain.
#ifndef MOM6X_MFW_FENCE
#define MOM6X_MFW_FENCE 2
#endif
#define LAYER_FENCE(n) do { if (MOM6X_MFW_FENCE > 0 && ((n) % MOM6X_MFW_FENCE) == MOM6X_MFW_FENCE - 1) __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// The switches of a launch.  SPEC = 0 reads every one of them from the arguments (uniform branches inside the march, their values
// parked in scalar registers the kernel does not have: 62 of them spilled to lanes of a vector register).  The three launches of a
// step of the split RK2 scheme with the reference's defaults (PPM with PPM_limit_pos, visc_rem, CONT_PPM_BETTER_ITER,
// CONT_PPM_USE_VISC_REM_MAX, CONT_PPM_MARGINAL_FACE_AREAS) are compiled with the switches known:
//   SPEC = 1: visc_rem + BT_cont                   (RK2.F90:613: the predictor's first call)
//   SPEC = 2: visc_rem + uhbt + u_cor + BT_cont    (RK2.F90:727)
//   SPEC = 3: visc_rem + uhbt + u_cor              (RK2.F90:1002: the corrector's call)
// launch() picks SPEC from the arguments; a face's result does not depend on it.
template <int SPEC>
struct Sw {
  bool use_visc_rem, corrected, set_bt, has_ucor, h_face, marginal, better_iter, use_vrm_max;
  int scheme, monotonic;
  __device__ __forceinline__ Sw(const FluxArgs &A, const LdsArgs &E) {
    use_visc_rem = SPEC ? true : (A.visc_rem != nullptr);
    corrected = SPEC ? (SPEC >= 2) : (A.uhbt != nullptr);
    set_bt = SPEC ? (SPEC <= 2) : (A.set_BT_cont != 0);
    has_ucor = SPEC ? (SPEC >= 2) : (A.u_cor != nullptr);
    h_face = SPEC ? (SPEC <= 2) : (E.h_face != nullptr);
    marginal = SPEC ? true : (E.marginal != 0);
    better_iter = SPEC ? true : (A.better_iter != 0);
    use_vrm_max = SPEC ? true : (A.use_visc_rem_max != 0);
    scheme = SPEC ? 0 : E.scheme;
    monotonic = SPEC ? 0 : E.monotonic;
  }
};

// What a lane keeps of its MAXL layers of one face column.
// FMA_ (mom6x_continuity_params.sum_order == MOM6X_SUM_TREE16_FMA): the two Horner chains of *_flux_layer and *_flux_thickness, u + du *
// visc_rem and the PPM edge formulas are evaluated with fused multiply-adds at fixed sites, the same ones as in
// oracle/orc_continuity.c -- results differ from the reference's in the last bits (include/mom6x.h).
template <int MAXL, bool FMA_ = false>
struct Col {
  static constexpr bool FMA = FMA_;
  double u[MAXL], v[MAXL];                      // u, visc_rem (0, 0 beyond nk: such a layer transports exactly +0.0)
  double mL[MAXL], mR[MAXL], mC[MAXL];          // minus cell: h_L, h_R, curv_3 = (h_L + h_R) - 2 h
  double pL[MAXL], pR[MAXL], pC[MAXL];          // plus cell
  double dt, IdT_m, IdT_p, Lf;
  double uh[MAXL];                              // the transports of the face's last evaluation (the result)
};

// zonal_flux_layer :896 / merid_flux_layer :1787 from registers.  The two upwind branches of the reference are
// mirror images (a = the edge value on the face side of the upwind cell, b = the far one); operands are selected
// instead of branching so that a wavefront with mixed flow directions does not run both.
// a + CFL * (p + q * r): the Horner chain of the average (p = (b - a) / 2, q = curv_3, r = CFL - 3/2) and of the marginal
// thickness (p = b - a, q = 3 curv_3, r = CFL - 1)
template <bool FMA>
__device__ __forceinline__ double horner(double a, double CFL, double p, double q, double r) {
  return FMA ? fma(CFL, fma(q, r, p), a) : a + CFL * (p + q * r);
}
template <int MAXL, bool FMA>
__device__ __forceinline__ double vel(const Col<MAXL, FMA> &C, int n, double du) {
  return FMA ? fma(du, C.v[n], C.u[n]) : C.u[n] + du * C.v[n];
}
template <int MAXL, bool FMA>
__device__ __forceinline__ void flux_reg(const Col<MAXL, FMA> &C, int n, double u, double &uh, double &duhdu, double *h_marg_out = nullptr) {
  const bool pos = (u > 0.0);
  const double a = pos ? C.mR[n] : C.pL[n], b = pos ? C.mL[n] : C.pR[n], curv_3 = pos ? C.mC[n] : C.pC[n];
  const double CFL = fabs(u) * C.dt * (pos ? C.IdT_m : C.IdT_p);
  const double uh_m = C.Lf * u * horner<FMA>(a, CFL, 0.5 * (b - a), curv_3, CFL - 1.5);
  const double hm_m = horner<FMA>(a, CFL, b - a, 3.0 * curv_3, CFL - 1.0);
  const bool moving = (u != 0.0);
  uh = moving ? uh_m : 0.0;
  const double h_marg = moving ? hm_m : 0.5 * (C.pL[n] + C.mR[n]);
  duhdu = C.Lf * h_marg * C.v[n];
  if (h_marg_out) *h_marg_out = h_marg;
}

// The same for a wavefront whose layers ALL flow one way (SIDE = +1: from the minus cell, -1: from the plus cell): the upwind cell
// is known, nothing is selected (12 of the ~46 instructions of a layer are v_cndmask, two more the zero test).  A lane at rest
// may take part if its visc_rem is zero as well: its transport is Lf * 0 * (...) and its derivative Lf * h * 0, zeros of either sign
// that leave the column sums as the reference's + 0.0 does (only sums are formed from this function, no transport is stored).
template <int SIDE, int MAXL, bool FMA>
__device__ __forceinline__ void flux_one_way(const Col<MAXL, FMA> &C, int n, double u, double &uh, double &duhdu) {
  const double a = (SIDE > 0) ? C.mR[n] : C.pL[n], b = (SIDE > 0) ? C.mL[n] : C.pR[n], curv_3 = (SIDE > 0) ? C.mC[n] : C.pC[n];
  const double CFL = fabs(u) * C.dt * ((SIDE > 0) ? C.IdT_m : C.IdT_p);
  uh = C.Lf * u * horner<FMA>(a, CFL, 0.5 * (b - a), curv_3, CFL - 1.5);
  const double h_marg = horner<FMA>(a, CFL, b - a, 3.0 * curv_3, CFL - 1.0);
  duhdu = C.Lf * h_marg * C.v[n];
}
template <int SIDE, int MAXL, bool FMA>
__device__ __forceinline__ bool wave_one_way(const Col<MAXL, FMA> &C, double du) {
  bool ok = true;
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    const double u = vel(C, n, du);
    ok = ok && (((SIDE > 0) ? (u > 0.0) : (u < 0.0)) || (u == 0.0 && C.v[n] == 0.0));
  }
  return !wave_any(!ok);
}

// The same test for a sweep whose transports are RESULTS (the first sweep, the sweeps of the solve towards uhbt): a layer at rest
// must not take part (its Lf * u * (...) can be a -0.0 where the reference stores +0.0) unless nobody stores it (`active`: the lane's
// face is in the launch's range; a layer beyond nk never is).
template <int SIDE, int MAXL, bool FMA>
__device__ __forceinline__ bool wave_one_way_strict(const Col<MAXL, FMA> &C, double du, bool active, int kl, int nk) {
  bool ok = true;
#pragma unroll
  for (int n = 0; n < MAXL; n++) {
    const double u = vel(C, n, du);
    ok = ok && (((SIDE > 0) ? (u > 0.0) : (u < 0.0)) || !(active && kl + KL * n < nk));
  }
  return !wave_any(!ok);
}
// MOM6X_MFW_ONEWAY_ALL (experiment of round 5, profiles/r05_mfw.md): EVERY sweep of a row whose first sweep is one-way asks whether it
// is one-way too and takes flux_one_way if so (= 1), or only counts (= 2: g_mfw_ow[0] sweeps, [1] of them one-way).
#ifndef MOM6X_MFW_ONEWAY_ALL
#define MOM6X_MFW_ONEWAY_ALL 0
#endif
#if MOM6X_MFW_ONEWAY_ALL == 2
__device__ unsigned long long g_mfw_ow[2];
#define OW_COUNT(hit) do { if (threadIdx.x == 0) { atomicAdd(&g_mfw_ow[0], 1ull); if (hit) atomicAdd(&g_mfw_ow[1], 1ull); } } while (0)
#else
#define OW_COUNT(hit)
#endif

// The Newton loop evaluates the same unrolled layer loop again and again with a new du.  Left alone, the compiler hoists
// everything of a layer that does not depend on du out of the loop (both upwind variants of b - a, 0.5 (b - a),
// 3 curv_3, ...: ~28 registers per layer) and the kernel goes to scratch memory.  An empty asm that "modifies" the
// layer's values keeps those expressions inside the loop: they cost a few instructions, not registers.
template <int MAXL, bool FMA>
__device__ __forceinline__ void keep_in_loop(Col<MAXL, FMA> &C, int n) {
  asm volatile("" : "+v"(C.mL[n]), "+v"(C.mR[n]), "+v"(C.mC[n]), "+v"(C.pL[n]), "+v"(C.pR[n]), "+v"(C.pC[n]), "+v"(C.u[n]), "+v"(C.v[n]));
}

// zonal_flux_adjust :1093-1242 / meridional_flux_adjust :1992-2140, iterated wavefront-uniformly; the Newton state is
// replicated over the 16 lanes of a face's row.  STORE: the evaluated transports are the result (the reference's
// uh_3d argument): they are kept in C.uh at every evaluation of a face that is still iterating and stored once at the end.  `lazy`: du_max / du_min are a lower / an upper bound of the CFL limits; the first test they do not
// decide ends the solve with need_exact = true (wavefront-uniform) and the caller repeats it with the limits.
template <int MAXL, bool STORE, bool STATS, bool FMA, typename StoreUh>
__device__ __forceinline__ double wave_flux_adjust(Col<MAXL, FMA> &C, bool active, double IareaMin, double uhbt,
                                                   double uh_tot_0, double duhdu_tot_0, double du_max, double du_min,
                                                   double tol_eta_cs, double tol_vel, int better_iter, bool lazy,
                                                   bool &need_exact, StoreUh store_uh, unsigned &evals, int ow_row, int kl, int nk TICK_PARAM,
                                                   double *dd_fin = nullptr, bool *dd_fresh = nullptr) {
  const int max_itts = 20;
  double du = 0.0;
  double uh_err = uh_tot_0 - uhbt, duhdu_tot = duhdu_tot_0;
  bool stale = false;   // du has moved since the sweep duhdu_tot comes from (a solve that ends on a tiny or a bisected step)
  double uh_err_best = fabs(uh_err);
  bool do_I = active;
  bool max_lazy = lazy, min_lazy = lazy, undecided = false;
  need_exact = false;
  int n_eval = 0;
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
          stale = true;
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
      n_eval++;
      double s_uh = 0.0, s_dd = 0.0;
      bool ow = false;
      if (MOM6X_MFW_ONEWAY_ALL && ow_row != 0)
        ow = (ow_row > 0) ? wave_one_way_strict<1>(C, du, active, kl, nk) : wave_one_way_strict<-1>(C, du, active, kl, nk);
      if (MOM6X_MFW_ONEWAY_ALL) OW_COUNT(ow);
      if (MOM6X_MFW_ONEWAY_ALL == 1 && ow && ow_row > 0) {
#pragma unroll
        for (int n = 0; n < MAXL; n++) {
          double uh, dd;
          keep_in_loop(C, n);
          flux_one_way<1>(C, n, vel(C, n, du), uh, dd);
          if (STORE) { if (do_I) C.uh[n] = uh; }
          s_uh = s_uh + uh; s_dd = s_dd + dd;
          LAYER_FENCE(n);
        }
      } else if (MOM6X_MFW_ONEWAY_ALL == 1 && ow) {
#pragma unroll
        for (int n = 0; n < MAXL; n++) {
          double uh, dd;
          keep_in_loop(C, n);
          flux_one_way<-1>(C, n, vel(C, n, du), uh, dd);
          if (STORE) { if (do_I) C.uh[n] = uh; }
          s_uh = s_uh + uh; s_dd = s_dd + dd;
          LAYER_FENCE(n);
        }
      } else {
#pragma unroll
        for (int n = 0; n < MAXL; n++) {
          double uh, dd;
          keep_in_loop(C, n);
          flux_reg(C, n, vel(C, n, du), uh, dd);
          if (STORE) { if (do_I) C.uh[n] = uh; }   // the last evaluation of a face is the one that stays
          s_uh = s_uh + uh; s_dd = s_dd + dd;
          LAYER_FENCE(n);
        }
      }
      if (itt < max_itts) {
        const double err = row_sum(s_uh) - uhbt, dtot = row_sum(s_dd);
        if (do_I) {
          uh_err = err; duhdu_tot = dtot;
          uh_err_best = dmin(uh_err_best, fabs(uh_err));
          stale = false;
        }
      }
    }
  }
  COUNT_ITT(STORE ? 8 : 9, n_eval);   // (slot 8 / 9 of g_mfw_t[0]: flux re-evaluations of the first / second solve; g_mfw_t[1]: solves)
  if (STATS) evals = (unsigned)__builtin_amdgcn_readfirstlane((int)(evals + (unsigned)n_eval));   // (wavefront-uniform: mom6x_continuity_stats)
  if (dd_fin) { *dd_fin = duhdu_tot; *dd_fresh = !stale; }
  return du;
}

// A lane owns one face of 16-layer slots: five 8-byte stores per 3-D result, each wavefront instruction a row of 32-byte pieces.
// What a wavefront pays for a vector-memory instruction does not depend on its width (~170-250 cycles each while the other
// wavefronts of the CU issue theirs: scripts/dev/mb_vmem.hip), so two slots go out as ONE instruction of 16 bytes per lane:
// v_permlane16_swap_b32 (gfx950 only -- as is this library: mom6_amd/build.py) exchanges the odd 16-lane rows of its first operand with the even rows of its second -- with
// A = slot 2m and B = slot 2m+1 of the values, an even face's lane ends with (A of its own face, A of the next face) and an
// odd face's lane with (B of the face before, B of its own): the two neighbouring doubles of layer kl + 32m / kl + 32m + 16.
__device__ __forceinline__ void swap_rows(double &a, double &b) {
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
// base: the array; rowb: the row's byte offset (uniform); lanep: the lane's byte offset of (the even face of its pair, layer
// kl + 16 * (fw & 1)); all lanes of the wavefront call it (`on_pair`: the lane's pair of faces is active).
template <int MAXL>
__device__ __forceinline__ void store_pairs(double *base, size_t rowb, unsigned lanep, size_t slab, const double *X, bool on_pair,
                                            int kl, int fw, int nk) {
#pragma unroll
  for (int m = 0; 2 * m < MAXL; m++) {
    double a = X[2 * m], b = (2 * m + 1 < MAXL) ? X[2 * m + 1] : 0.0;
    swap_rows(a, b);
    const int n = 2 * m + (fw & 1);
    if (on_pair && n < MAXL && kl + KL * n < nk) {
      double2 v2; v2.x = a; v2.y = b;
      *(double2 *)((char *)base + (rowb + (size_t)m * 2 * KL * slab * 8) + lanep) = v2;
    }
  }
}

// ---- staging through LDS ------------------------------------------------------------------------------------------
// A wavefront (4 faces along i = 32 bytes of every pitched row) MARCHES along j.  Everything row jj needs is copied
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, no registers) into the wavefront's OWN region
// while the Newton solves of row jj-1 run from registers, and is picked up with ds_read_b64 once it has landed: no
// barrier, no other wavefront involved (a first version staged whole 128-byte lines per work-group: the two barriers
// per row tied four wavefronts with different Newton iteration counts together, 52 % of a wavefront's life was spent
// waiting).  The four wavefronts of a work-group cover the four quarters of the same lines, so the lines meet in
// that CU's L1 / that XCD's L2.  The region holds segments of 4 doubles:
//   3-D slots [slot][k][4]:  DIR = 1:  h rows jj-1 .. jj+3 (the stencil of cell jj+1), u row jj, visc_rem row jj
//                            DIR = 0:  h cells i0-4..i0-1 | i0..i0+3 | i0+4..i0+7 of row jj, u, visc_rem
//   2-D segments (metrics of the row, see L2D below).
// The 64 lanes (f, q) of a wavefront read 64 consecutive doubles of a slot (k = q, column f): no bank conflict.
// A meridional face (i, jj) lies between the cells jj and jj+1: its minus cell is the plus cell of the previous row of
// the march, so each step reconstructs ONE cell per face (the stand-alone kernel did two) and reads five rows of h of
// which four come from L2.
struct L2D { int plane, dj, dcol; };   // metric plane (MOM6X_G_*; -1: the uhbt argument), row offset, column offset in segments
template <int DIR> struct Stage;
template <> struct Stage<1> {
  static constexpr int NS3 = 7, NL2 = 15, SU = 5, SV = 6;
  __device__ static L2D line(int q) {
    switch (q) {
      case 0: return {MOM6X_G_IdyT, 0, 0};    case 1: return {MOM6X_G_IdyT, 1, 0};   case 2: return {MOM6X_G_dx_Cv, 0, 0};
      case 3: return {MOM6X_G_IareaT, 0, 0};  case 4: return {MOM6X_G_IareaT, 1, 0}; case 5: return {MOM6X_G_dyT, 0, 0};
      case 6: return {MOM6X_G_dyT, 1, 0};     case 7: return {MOM6X_G_dyCv, 0, 0};   case 8: return {MOM6X_G_mask2dCv, 0, 0};
      case 9: return {-1, 0, 0};
      default: return {MOM6X_G_mask2dT, q - 11, 0};   // 10..14: rows jj-1 .. jj+3
    }
  }
};
template <> struct Stage<0> {
  static constexpr int NS3 = 5, NL2 = 13, SU = 3, SV = 4;
  __device__ static L2D line(int q) {
    switch (q) {
      case 0: return {MOM6X_G_IdxT, 0, 0};    case 1: return {MOM6X_G_IdxT, 0, 1};   case 2: return {MOM6X_G_dy_Cu, 0, 0};
      case 3: return {MOM6X_G_IareaT, 0, 0};  case 4: return {MOM6X_G_IareaT, 0, 1}; case 5: return {MOM6X_G_dxT, 0, 0};
      case 6: return {MOM6X_G_dxT, 0, 1};     case 7: return {MOM6X_G_dxCu, 0, 0};   case 8: return {MOM6X_G_mask2dCu, 0, 0};
      case 9: return {-1, 0, 0};
      default: return {MOM6X_G_mask2dT, 0, q - 11};   // 10..12: segments i0-4 | i0 | i0+4
    }
  }
};

__device__ __forceinline__ void glds16(const double *src, double *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The next row's LDS-DMA (issued before face_column is called) is waited for in front of the row's LAST group of stores: see the row
// top of k_mass_flux_wave.
#define ROW_DMA_WAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// Everything of one face column after the reconstruction: first sweep, flux_adjust towards uhbt, stores, flux
// thickness, set_*_BT_cont.  All lanes of the wavefront call it (row reductions inside).
template <int DIR, int MAXL, bool STATS, int SPEC, bool FMA>
__device__ __forceinline__ void face_column(Col<MAXL, FMA> &C, const FluxArgs &A, const LdsArgs &E, size_t rowb, unsigned lane2,
                                            unsigned lane3, size_t slab, bool active, int kl, int nk, double IareaMin,
                                            double uhbt_f, double dC_f, double dx_W_in, double dx_E_in, const double *G,
                                            int pitch, unsigned &evals, unsigned &solves, unsigned &redos, bool pairs, unsigned lanep, int fw TICK_PARAM) {
  // Addresses: (uniform base pointer + uniform byte offset) + a 32-bit per-lane byte offset that never changes
  // (lane2: the face's column in a row; lane3: + the lane's first layer) -- the scalar-base addressing mode, one
  // register per lane instead of a 64-bit address per store that the compiler would keep alive across the march.
  auto st2 = [&](double *base, double v) { *(double *)((char *)base + rowb + lane2) = v; };
  auto st3 = [&](double *base, int n, double v) { *(double *)((char *)base + (rowb + (size_t)n * KL * slab * 8) + lane3) = v; };
  const double dt = A.dt;
  const Sw<SPEC> W(A, E);
  const bool use_visc_rem = W.use_visc_rem;
  const bool need_adjust = W.corrected || W.set_bt;

  // ---- limits on du that keep the CFL number between -1 and 1 (:646-723) --------------------------------------
  double du_max_CFL = 0.0, du_min_CFL = 0.0;
  const double CFL_dt = A.CFL_limit_adjust / dt;
  bool lazy = false;
  auto calc_visc_rem_max = [&]() {   // (recomputed where it is needed: an order-independent max)
    double vrm = 1.0;
    if (need_adjust && use_visc_rem && W.use_vrm_max) {
      double pm = 0.0;
#pragma unroll
      for (int n = 0; n < MAXL; n++)
        if (kl + KL * n < nk) pm = dmax(pm, C.v[n]);
      vrm = row_max(pm);
    }
    return vrm;
  };
  // The exact limits with visc_rem: a k-recurrence (two divisions per layer), walked by the wavefront: the lane that
  // owns layer k broadcasts its operands to the row.  Wavefront-uniform; rarely needed, so the metrics are read again
  // from memory here rather than kept in registers through the solves.
  auto exact_bounds = [&]() {
    const size_t f2 = (rowb + lane2) / 8;
    const double dx_W = gm(G, Dm{0, 0, 0, 0, 0, 0, 0, (int)slab}, DIR ? MOM6X_G_dyT : MOM6X_G_dxT)[f2];
    const double dx_E = gm(G, Dm{0, 0, 0, 0, 0, 0, 0, (int)slab}, DIR ? MOM6X_G_dyT : MOM6X_G_dxT)[f2 + (DIR ? pitch : 1)];
    const double maskC = gm(G, Dm{0, 0, 0, 0, 0, 0, 0, (int)slab}, DIR ? MOM6X_G_mask2dCv : MOM6X_G_mask2dCu)[f2];
    const double vrm = calc_visc_rem_max();
    double I_vrm = 0.0;
    if (vrm > 0.0) I_vrm = 1.0 / vrm;
    du_max_CFL = 2.0 * (CFL_dt * dx_W) * I_vrm;
    du_min_CFL = -2.0 * (CFL_dt * dx_E) * I_vrm;
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      const double q_max = (dx_W * CFL_dt - C.u[n]) / C.v[n];
      const double q_min = -(dx_E * CFL_dt + C.u[n]) / C.v[n];
#pragma unroll 1
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
    const double dx_W = dx_W_in, dx_E = dx_E_in;
    const double vrm = calc_visc_rem_max();
    double I_vrm = 0.0;
    if (vrm > 0.0) I_vrm = 1.0 / vrm;
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
  TICK(1);

  // ---- first sweep: layer transports and their column sums (:615-668) -------------------------------------------
  auto store_uh = [&](int n, double uh, bool on) { if (on && (kl + KL * n < nk)) st3(A.uh, n, uh); };
  double uh_tot_0 = 0.0, duhdu_tot_0 = 0.0;
  // The launch that sets BT_cont without a correction (SPEC 1) wants the marginal thickness at the UNCORRECTED velocity for BT_cont%h_u
  // (zonal_flux_thickness :975): the very h_marg the first sweep forms.  Kept (5 doubles) instead of formed again.
#ifdef MOM6X_MFW_NO_KEEP_HM   // (A/B)
  constexpr bool KEEP_HM = false;
#else
  constexpr bool KEEP_HM = (SPEC == 1) && (MOM6X_MFW_ONEWAY_ALL != 1);
#endif
  double hm0[MAXL];
#pragma unroll
  for (int n = 0; n < MAXL; n++) hm0[n] = 0.0;
  int ow_row = 0;   // (MOM6X_MFW_ONEWAY_ALL) +1 / -1: every layer of the wavefront's four columns flows out of its minus / plus cell at du = 0
  auto first_sweep = [&]() {
    double s_uh = 0.0, s_dd = 0.0;
    if (MOM6X_MFW_ONEWAY_ALL) {
      ow_row = wave_one_way_strict<1>(C, 0.0, active, kl, nk) ? 1 : (wave_one_way_strict<-1>(C, 0.0, active, kl, nk) ? -1 : 0);
      OW_COUNT(ow_row != 0);
    }
    if (MOM6X_MFW_ONEWAY_ALL == 1 && ow_row > 0) {
#pragma unroll
      for (int n = 0; n < MAXL; n++) {
        double uh, dd;
        flux_one_way<1>(C, n, C.u[n], uh, dd);
        C.uh[n] = uh;
        s_uh = s_uh + uh; s_dd = s_dd + dd;
        LAYER_FENCE(n);
      }
    } else if (MOM6X_MFW_ONEWAY_ALL == 1 && ow_row < 0) {
#pragma unroll
      for (int n = 0; n < MAXL; n++) {
        double uh, dd;
        flux_one_way<-1>(C, n, C.u[n], uh, dd);
        C.uh[n] = uh;
        s_uh = s_uh + uh; s_dd = s_dd + dd;
        LAYER_FENCE(n);
      }
    } else {
#pragma unroll
      for (int n = 0; n < MAXL; n++) {
        double uh, dd;
        flux_reg(C, n, C.u[n], uh, dd, KEEP_HM ? &hm0[n] : nullptr);
        C.uh[n] = uh;
        s_uh = s_uh + uh; s_dd = s_dd + dd;
        LAYER_FENCE(n);
      }
    }
    if (need_adjust) { uh_tot_0 = row_sum(s_uh); duhdu_tot_0 = row_sum(s_dd); }
  };
  first_sweep();
  TICK(2);

  // ---- flux_adjust towards uhbt; uh, u_cor, du_cor ---------------------------------------------------------------
  double du_fin = 0.0;
  const bool corrected = W.corrected;
  if (corrected) {
    bool redo;
    if (STATS) solves = (unsigned)__builtin_amdgcn_readfirstlane((int)(solves + 1u));
    du_fin = wave_flux_adjust<MAXL, true, STATS>(C, active, IareaMin, uhbt_f, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                          A.tol_eta, A.tol_vel, W.better_iter, lazy, redo, store_uh, evals, ow_row, kl, nk TICK_ARG);
    if (redo) {   // wavefront-uniform
      if (STATS) redos = (unsigned)__builtin_amdgcn_readfirstlane((int)(redos + 1u));
      exact_bounds(); lazy = false;
      first_sweep();   // (the abandoned solve has overwritten some of the first transports)
      du_fin = wave_flux_adjust<MAXL, true, STATS>(C, active, IareaMin, uhbt_f, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                            A.tol_eta, A.tol_vel, W.better_iter, false, redo, store_uh, evals, ow_row, kl, nk TICK_ARG);
    }
    // The reference's du is never -0.0: it starts at +0.0 and every later value is a sum or a mean with at least one operand that is
    // not -0.0 (round to nearest: x + y = -0 only for x = y = -0).  The wave kernel's select chains can leave a -0.0 on faces whose
    // whole visc_rem column is zero (duhdu_tot = 0: Newton steps of +-inf bisected back to zero); x + 0.0 is the identity but for
    // that one value.  (The file is compiled with signed zeros honoured; tests/helpers.py makes no allowance any more.)
    du_fin = du_fin + 0.0;
    if (!W.set_bt) ROW_DMA_WAIT;
    if (active && kl == 0 && A.du_cor) st2(A.du_cor, du_fin);
  } else if (!W.set_bt) ROW_DMA_WAIT;
  // ONE store per transport (HBM write traffic 3.0 -> 1.9 GB); `pairs` (uniform): every pair of faces of the wavefront is
  // active or inactive as a whole (all but the wavefronts on the rim of the face range)
  if (pairs) store_pairs<MAXL>(A.uh, rowb, lanep, slab, C.uh, active, kl, fw, nk);
  else
#pragma unroll
    for (int n = 0; n < MAXL; n++) store_uh(n, C.uh[n], active);
  TICK(3);

  if (corrected && W.has_ucor) {
    if (pairs) {
      double uc[MAXL];
#pragma unroll
      for (int n = 0; n < MAXL; n++) uc[n] = vel(C, n, du_fin);
      store_pairs<MAXL>(A.u_cor, rowb, lanep, slab, uc, active, kl, fw, nk);
    } else if (active) {
#pragma unroll
      for (int n = 0; n < MAXL; n++)
        if (kl + KL * n < nk) st3(A.u_cor, n, vel(C, n, du_fin));
    }
  }

  // ---- zonal/merid_flux_thickness (:975 / :1866) at the corrected velocities ------------------------------------
  if (W.h_face && (active || pairs)) {
    const bool use_cor = corrected && W.has_ucor;
    double hf[MAXL];
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      if (KEEP_HM) {   // (no correction, marginal thickness, visc_rem: the switches of SPEC 1)
        hf[n] = hm0[n] * (C.v[n] * 1.0);
        continue;
      }
      const double uf = use_cor ? (vel(C, n, du_fin)) : C.u[n];
      const bool pos = (uf > 0.0);
      const double a = pos ? C.mR[n] : C.pL[n], b = pos ? C.mL[n] : C.pR[n], curv_3 = pos ? C.mC[n] : C.pC[n];
      const double CFL = fabs(uf) * dt * (pos ? C.IdT_m : C.IdT_p);
      double h_avg = horner<FMA>(a, CFL, 0.5 * (b - a), curv_3, CFL - 1.5);
      double h_marg = horner<FMA>(a, CFL, b - a, 3.0 * curv_3, CFL - 1.0);
      if (uf == 0.0) { h_avg = 0.5 * (C.pL[n] + C.mR[n]); h_marg = h_avg; }
      double hu = W.marginal ? h_marg : h_avg;
      if (use_visc_rem) hu = hu * (C.v[n] * 1.0);
      else hu = hu * 1.0;
      hf[n] = hu;
    }
    if (pairs) store_pairs<MAXL>(E.h_face, rowb, lanep, slab, hf, active, kl, fw, nk);
    else
#pragma unroll
      for (int n = 0; n < MAXL; n++)
        if (kl + KL * n < nk) st3(E.h_face, n, hf[n]);
  }
  TICK(4);
  if (!W.set_bt) return;

  // ---- set_zonal_BT_cont :1246-1409 / set_merid_BT_cont :2143-2304 -----------------------------------------------
  const double Idt = 1.0 / dt, min_visc_rem = 0.1, CFL_min = 1e-6;
  double du0;
  // FAmt_0 (:1330-1337) is the sum of duhdu at u + du0 visc_rem: the very sum the zero-transport solve's LAST sweep formed (same
  // expression, same order) unless the solve ended on a step it did not evaluate -- then, wavefront-uniformly, the sweep is made.
  double dd0 = 0.0;
  bool dd0_fresh = false;
  {
    bool redo;
    auto no_store = [](int, double, bool) {};
    if (STATS) solves = (unsigned)__builtin_amdgcn_readfirstlane((int)(solves + 1u));
    du0 = wave_flux_adjust<MAXL, false, STATS>(C, active, IareaMin, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                        A.tol_eta, A.tol_vel, W.better_iter, lazy, redo, no_store, evals, ow_row, kl, nk TICK_ARG, &dd0, &dd0_fresh);
    if (redo) {
      if (STATS) redos = (unsigned)__builtin_amdgcn_readfirstlane((int)(redos + 1u));
      exact_bounds(); lazy = false;
      du0 = wave_flux_adjust<MAXL, false, STATS>(C, active, IareaMin, 0.0, uh_tot_0, duhdu_tot_0, du_max_CFL, du_min_CFL,
                                          A.tol_eta, A.tol_vel, W.better_iter, false, redo, no_store, evals, ow_row, kl, nk TICK_ARG, &dd0, &dd0_fresh);
    }
  }
  TICK(5);
  const double du_CFL = (CFL_min * Idt) * dC_f;
  // duR / duL (:1293-1316): min / max of the quotients (sum_order TREE16's definition)
  double duR = dmin(0.0, du0 - du_CFL), duL = dmax(0.0, du0 + du_CFL);
  {
    const double vrl_floor = min_visc_rem * calc_visc_rem_max();
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
  TICK(6);
  // three trial velocities (:1330-1349), five column sums
  double FAmt_L = 0.0, FAmt_R = 0.0, FAmt_0 = 0.0, uhtot_L = 0.0, uhtot_R = 0.0;
  const bool sweep_0 = wave_any(!dd0_fresh);
  if (sweep_0) {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double uh_0, d_0;
      flux_reg(C, n, vel(C, n, du0), uh_0, d_0);
      FAmt_0 = FAmt_0 + d_0;
      LAYER_FENCE(n);
    }
  }
  // duL / duR are made so that every layer flows out of the minus / the plus cell (:1293-1316; a layer whose visc_rem lies under the
  // floor may not): wavefront-uniformly, such a sweep knows its upwind cell
  if (wave_one_way<1>(C, duL)) {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double uh_L, d_L;
      flux_one_way<1>(C, n, vel(C, n, duL), uh_L, d_L);
      FAmt_L = FAmt_L + d_L; uhtot_L = uhtot_L + uh_L;
      LAYER_FENCE(n);
    }
  } else {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double uh_L, d_L;
      flux_reg(C, n, vel(C, n, duL), uh_L, d_L);
      FAmt_L = FAmt_L + d_L; uhtot_L = uhtot_L + uh_L;
      LAYER_FENCE(n);
    }
  }
  if (wave_one_way<-1>(C, duR)) {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double uh_R, d_R;
      flux_one_way<-1>(C, n, vel(C, n, duR), uh_R, d_R);
      FAmt_R = FAmt_R + d_R; uhtot_R = uhtot_R + uh_R;
      LAYER_FENCE(n);
    }
  } else {
#pragma unroll
    for (int n = 0; n < MAXL; n++) {
      double uh_R, d_R;
      flux_reg(C, n, vel(C, n, duR), uh_R, d_R);
      FAmt_R = FAmt_R + d_R; uhtot_R = uhtot_R + uh_R;
      LAYER_FENCE(n);
    }
  }
  FAmt_0 = sweep_0 ? row_sum(FAmt_0) : dd0;
  FAmt_L = row_sum(FAmt_L); FAmt_R = row_sum(FAmt_R);
  uhtot_L = row_sum(uhtot_L); uhtot_R = row_sum(uhtot_R);
  TICK(7);
  // The six planes of BT_cont.  Every lane of a face's row holds the same sums, so lane kl = 0..5 of the row stores plane kl: ONE
  // instruction for the six instead of six with four lanes each.
  double FA_0 = FAmt_0, FA_avg = FAmt_0;
  if ((duL - du0) != 0.0) FA_avg = uhtot_L / (duL - du0);
  if (FA_avg > dmax(FA_0, FAmt_L)) FA_avg = dmax(FA_0, FAmt_L);
  else if (FA_avg < dmin(FA_0, FAmt_L)) FA_0 = FA_avg;
  const double v_m0 = FA_0;
  double v_umm = 0.0;
  if (!(fabs(FA_0 - FAmt_L) <= 1e-12 * FA_0)) v_umm = (1.5 * (duL - du0)) * ((FAmt_L - FA_avg) / (FAmt_L - FA_0));

  FA_0 = FAmt_0; FA_avg = FAmt_0;
  if ((duR - du0) != 0.0) FA_avg = uhtot_R / (duR - du0);
  if (FA_avg > dmax(FA_0, FAmt_R)) FA_avg = dmax(FA_0, FAmt_R);
  else if (FA_avg < dmin(FA_0, FAmt_R)) FA_0 = FA_avg;
  const double v_p0 = FA_0;
  double v_upp = 0.0;
  if (!(fabs(FAmt_R - FA_0) <= 1e-12 * FA_0)) v_upp = (1.5 * (duR - du0)) * ((FAmt_R - FA_avg) / (FAmt_R - FA_0));

  double val = v_m0;
  double *plane = A.FA_m0;
  if (kl == 1) { val = FAmt_L; plane = A.FA_mm; }
  if (kl == 2) { val = v_umm; plane = A.uBT_mm; }
  if (kl == 3) { val = v_p0; plane = A.FA_p0; }
  if (kl == 4) { val = FAmt_R; plane = A.FA_pp; }
  if (kl == 5) { val = v_upp; plane = A.uBT_pp; }
  ROW_DMA_WAIT;
  if (active && kl < 6) *(double *)((char *)plane + rowb + lane2) = val;
}

constexpr int SEG = 4;   // doubles per segment = faces per wavefront

// Wavefronts per SIMD the kernel is compiled for: 8 slots per lane (nk <= 128) leave room for one, 4-5 slots for two; with 3 slots
// (nk <= 48) the state is 56 registers smaller and three fit without scratch (5-16 % faster than two: profiles/r06_mfw.md; the
// 5-slot kernels squeezed to three wavefronts go 84-276 bytes per lane into scratch and lose 8-70 %).
// (of the 3-slot kernels the two lean specialisations fit; the general one and SPEC 2 would spill 12-76 bytes)
constexpr int mfw_occ(int maxl, int spec, bool fma) { return (maxl > 5) ? 1 : ((maxl <= 3 && (spec == 1 || spec == 3)) ? 3 : 2); }
template <int DIR, int MAXL, bool STATS, int SPEC, bool FMA>
__global__ void __launch_bounds__(NF * KL, mfw_occ(MAXL, SPEC, FMA))
k_mass_flux_wave(Dm d, const double *__restrict__ G, FluxArgs A, LdsArgs E) {
  using ST = Stage<DIR>;
  const Sw<SPEC> W(A, E);
  extern __shared__ double S_all[];
  // A work-group = a strip of 16 faces along i (bx) and E.rows consecutive rows (chunk): E.gx strips, E.gy chunks;
  // its wavefronts are independent of each other.
  // (one or two parts: the work-groups of the second part follow those of the first)
  const int n0 = E.pgx[0] * E.pgy[0];
  const int pt = (E.np > 1 && (int)blockIdx.x >= n0) ? 1 : 0;
  const int bid = (int)blockIdx.x - (pt ? n0 : 0);
  const int pa0 = E.pa0[pt], pa1 = E.pa1[pt], pb0 = E.pb0[pt], pb1 = E.pb1[pt];
  const int bx = bid % E.pgx[pt], chunk = bid / E.pgx[pt];
  const int j0 = pb0 + chunk * E.rows, j1 = min(j0 + E.rows - 1, pb1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fw = lane >> 4, kl = lane & 15;
  const int i0 = E.pib[pt] + bx * NF + w * SEG, i = i0 + fw;   // i0: the wavefront's first face
  const bool active = (i >= pa0 && i <= pa1);
  const bool wave_on = wave_any(active);   // (a wavefront without faces still meets the others at the row barrier below)
  const int nk = d.nk;
  const size_t slab = (size_t)d.slab;
  const bool use_visc_rem = W.use_visc_rem;
  const int ns3 = use_visc_rem ? ST::NS3 : ST::NS3 - 1;   // (visc_rem is the last 3-D slot)
  // This wavefront's region: NS3 slots of KP = 16 * MAXL layers (a compile-time stride: every LDS address of a lane is
  // ONE base register + an immediate), then the 2-D segments.
  constexpr int KP = KL * MAXL;
  constexpr int WAVE_LDS = SEG * (ST::NS3 * KP + 16);   // doubles (NL2 <= 16)
  double *S = S_all + (size_t)w * WAVE_LDS;
  double *S2 = S + (size_t)ST::NS3 * KP * SEG;
  TICK_INIT;

  // ---- DMA roles: lane = (segment sg of a round of 32, 16-byte piece pp) ----------------------------------------------
  // Source address = uniform (array + row + 32 layers per round) + ONE per-lane 32-bit byte offset.
  const int sg = lane >> 1, pp = lane & 1;
  const unsigned dma3 = (unsigned)(((size_t)sg * slab + (size_t)(pp * 2)) * 8);
  const L2D L2 = ST::line(sg < ST::NL2 ? sg : 0);
  const unsigned dma2 = (unsigned)(((ptrdiff_t)(L2.plane >= 0 ? L2.plane : 0) * (ptrdiff_t)slab + (ptrdiff_t)(L2.dj + d.joff) * d.pitch +
                                    (ptrdiff_t)L2.dcol * SEG + d.ioff + pp * 2) * 8);   // from G; >= 0: dj + joff >= 0, dcol*SEG + ioff >= 0
  const bool do2 = (sg < ST::NL2) && (L2.plane >= 0);
  const bool do_uhbt = (sg == 9) && W.corrected;
  // DIR = 1: the five h slots are a RING over the rows of the march: row r lives in slot (r + 10) % 5, and a step
  // only fetches the one row that is new to it (jj + 3, into the slot row jj - 2 has just left); `all_rows`: the first
  // step of a chunk fills the ring.
  auto issue_dma = [&](int jj, bool all_rows) {
    const size_t row0 = ((size_t)(i0 + d.ioff) + (size_t)(jj + d.joff) * (size_t)d.pitch) * 8;   // (i0, jj) in a plane, bytes
#pragma unroll
    for (int s = 0; s < ST::NS3; s++) {
      if (s >= ns3) break;
      if (DIR && s < 4 && !all_rows) continue;          // rows jj-1 .. jj+2 are in the ring already
      const double *arr;
      ptrdiff_t shift;
      int slot = s;
      if (DIR) { arr = (s < 5) ? A.h_in : (s == 5 ? A.u : A.visc_rem); shift = (s < 5) ? (ptrdiff_t)(s - 1) * d.pitch * 8 : 0;
                 if (s < 5) slot = (jj + s + 9) % 5; }   // row jj - 1 + s
      else     { arr = (s < 3) ? A.h_in : (s == 3 ? A.u : A.visc_rem); shift = (s < 3) ? (ptrdiff_t)(s - 1) * SEG * 8 : 0; }
      const char *base = (const char *)arr + (ptrdiff_t)row0 + shift;
      // DIR = 0, s = 0: of the segment i0-4 .. i0-1 only the second 16-byte piece (cells i0-2, i0-1) is in anybody's stencil
      const bool piece_on = DIR || s != 0 || pp == 1;
      for (int r = 0; r * 32 < nk; r++) {
        if (piece_on && r * 32 + sg < nk)
          glds16((const double *)(base + (size_t)r * 32 * slab * 8 + dma3), S + (size_t)(slot * KP + r * 32) * SEG);
      }
    }
    const char *g2 = (const char *)G + ((size_t)i0 + (size_t)jj * (size_t)d.pitch) * 8;
    if (do2) glds16((const double *)(g2 + dma2), S2);
    if (do_uhbt) glds16((const double *)((const char *)A.uhbt + row0 + pp * 16), S2);   // (lane-linear: segment 9 again)
  };

  // ---- where a lane finds its values in LDS ------------------------------------------------------------------------
  constexpr int NH = DIR ? 5 : 6;
  const double *Sb = S + kl * SEG + fw;        // layer kl, column fw of slot 0; layer n adds n * 16 segments
  const double *L = S2 + fw;                   // the 2-D segments
  // DIR = 1: h row jj-1+q = slot q.  DIR = 0: cell i-2+q = column c = SEG + fw - 2 + q of the 12 cells i0-4 .. i0+7,
  // slot c >> 2, column c & 3: relative to Sb that is a per-lane offset xo[q] (fw = 0..3, q = 0..5: c = 2..9)
  int xo[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int c = SEG + fw - 2 + q;
    xo[q] = (c >> 2) * KP * SEG + (c & 3) - fw;
  }
  // per-lane byte offsets of the stores (see face_column)
  const unsigned lane2 = (unsigned)((size_t)((active ? i : pa1) + d.ioff) * 8);   // inactive lanes alias the last face (never written)
  const unsigned lane3 = lane2 + (unsigned)((size_t)kl * slab * 8);
  // the paired stores of face_column (store_pairs): the partner of face i0 + fw is i0 + (fw ^ 1) (i0 + ioff is a multiple of 4:
  // the pair is 16-byte aligned)
#ifdef MOM6X_MFW_NO_PAIRS
  const bool pairs = false;
#else
  const int ip_ = i0 + (fw ^ 1);
  const bool pairs = !E.no_pairs && !wave_any(active != (ip_ >= pa0 && ip_ <= pa1));
#endif
  const unsigned lanep = (unsigned)((size_t)((active ? i0 + (fw & ~1) : i0) + d.ioff) * 8) + (unsigned)((size_t)(kl + KL * (fw & 1)) * slab * 8);

  Col<MAXL, FMA> C;
  C.dt = A.dt;
#pragma unroll
  for (int n = 0; n < MAXL; n++) { C.pL[n] = 0.0; C.pR[n] = 0.0; C.pC[n] = 0.0; }
  unsigned st_evals = 0, st_solves = 0, st_redos = 0;   // wavefront-uniform counts over the march (mom6x_continuity_stats)

  const int jstart = DIR ? j0 - 1 : j0;   // meridional: a first step that only reconstructs cell j0
  if (wave_on) issue_dma(jstart, true);
  bool dma_pending = true;   // uniform: the row's DMA has not been waited for yet
  for (int jj = jstart; jj <= j1; jj++) {
    // The four wavefronts of a work-group share nothing but cache lines: a 128-byte line of h, u, visc_rem or of an output
    // holds the 32 bytes of each of them.  Left alone they drift rows apart (their Newton counts differ), every wavefront
    // then fetches the line for itself and the partial lines they store reach memory one by one.  Meeting once per row
    // keeps the four requests within the L2's reach: plain / adjust modes 2.0 -> 1.5 / 2.6 -> 2.25 ms (x).
    // The next row's DMA is waited for inside face_column, in front of the row's LAST group of stores (ROW_DMA_WAIT: the requests are a
    // Newton solve old by then, the wait is free), so the stores stay in flight across the row top instead of being waited for here
    // with the DMA (-5 % zonal / -3 % meridional in the launches without BT_cont's tail behind the stores, profiles/r06_mfw.md); only a
    // row whose predecessor made no face column (the first of a march) waits here.  A raw s_barrier: __syncthreads() puts its own
    // vmcnt(0) in front of the barrier while an LDS-DMA is in flight, and the compiler does not connect an LDS-DMA with the ds_read
    // of its destination -- the waits are this code's business.
    __builtin_amdgcn_s_barrier();
    if (!wave_on) continue;
    if (dma_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // row jj has landed
    dma_pending = true;
    TICK(10);
    const bool face_row = (!DIR) || (jj >= j0);
    // ---- LDS -> registers, layer by layer, with the PPM reconstruction + limiter on the way ---------------------------
    double IareaMin, uhbt_f, dC_f, dx_W, dx_E;
    {
      int ring[5];   // DIR = 1: where the rows jj-1 .. jj+3 are in the ring of h slots (uniform)
#pragma unroll
      for (int q = 0; q < 5; q++) ring[q] = ((jj + q + 9) % 5) * KP * SEG;
      double m6[6];
#pragma unroll
      for (int q = 0; q < NH; q++) {
        const double mv = DIR ? L[(10 + q) * SEG] : L[10 * SEG + SEG - 2 + q];
        m6[q] = active ? mv : 0.0;
      }
#pragma unroll
      for (int n = 0; n < MAXL; n++) {
        const bool on = active && (kl + KL * n < nk);   // (a layer beyond nk reads the segments behind its slot: discarded)
        double hst[6];
#pragma unroll
        for (int q = 0; q < NH; q++) {
          const double hv = DIR ? Sb[ring[q] + n * KL * SEG] : Sb[xo[q] + n * KL * SEG];
          hst[q] = on ? hv : 0.0;
        }
        const double uu = Sb[(ST::SU * KP + n * KL) * SEG];
        const double vv = use_visc_rem ? Sb[(ST::SV * KP + n * KL) * SEG] : 1.0;
        C.u[n] = on ? uu : 0.0;
        C.v[n] = on ? vv : 0.0;
        double hl = 0.0, hr = 0.0, c3 = 0.0;
        if (DIR) {   // the plus cell of the last step is this step's minus cell
          C.mL[n] = C.pL[n]; C.mR[n] = C.pR[n]; C.mC[n] = C.pC[n];
          if (on) edge5<FMA>(&hst[0], &m6[0], W.scheme, W.monotonic, E.h_min, hl, hr, c3);
          C.pL[n] = hl; C.pR[n] = hr; C.pC[n] = c3;
        } else {
          if (on) edge5<FMA>(&hst[0], &m6[0], W.scheme, W.monotonic, E.h_min, hl, hr, c3);
          C.mL[n] = hl; C.mR[n] = hr; C.mC[n] = c3;
          hl = 0.0; hr = 0.0; c3 = 0.0;
          if (on) edge5<FMA>(&hst[1], &m6[1], W.scheme, W.monotonic, E.h_min, hl, hr, c3);
          C.pL[n] = hl; C.pR[n] = hr; C.pC[n] = c3;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int ip = DIR ? SEG : 1;   // the plus neighbour: the next segment (DIR = 1: row jj+1) or the next column
      C.IdT_m = L[0]; C.IdT_p = L[ip];
      C.Lf = L[2 * SEG] * 1.0;   // G%dy_Cu * por_face_areaU (== 1)
      IareaMin = dmin(L[3 * SEG], L[3 * SEG + ip]);
      dx_W = L[5 * SEG]; dx_E = L[5 * SEG + ip];
      dC_f = W.set_bt ? L[7 * SEG] : 0.0;
      uhbt_f = (W.corrected && active) ? L[9 * SEG] : 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // row jj is in registers: its space is free ...
    TICK(11);
    if (jj < j1) issue_dma(jj + 1, false);                     // ... and fills while this row's Newton solves run
    if (!face_row) continue;
    TICK(12);
    const size_t rowb = (size_t)(jj + d.joff) * (size_t)d.pitch * 8;
    face_column<DIR, MAXL, STATS, SPEC>(C, A, E, rowb, lane2, lane3, slab, active, kl, nk, IareaMin, uhbt_f, dC_f, dx_W, dx_E, G, d.pitch, st_evals,
                           st_solves, st_redos, pairs, lanep, fw TICK_ARG);
    dma_pending = false;
    TICK(13);
  }
  TICK_FLUSH;
  if (STATS && E.stats && lane == 0 && wave_on) {   // one update per wavefront and launch: flux sweeps of the Newton solves, solves, redos
    atomicAdd(&E.stats[0], (unsigned long long)st_evals); atomicAdd(&E.stats[1], (unsigned long long)st_solves);
    atomicAdd(&E.stats[2], (unsigned long long)st_redos);
  }
}

inline int d_slab_of(const mom6x_ctx *c) { return c->d.slab; }
template <int DIR, int MAXL>
int launch(mom6x_ctx *c, const FluxArgs &A, const LdsArgs &E0) {
  using ST = Stage<DIR>;
  LdsArgs E = E0;   // (E.np, pa0.., pib set by the caller)
  E.retry = nullptr; E.force_walk = 0;
  // The Newton statistics are a separate instantiation: the counters cost the 253-register kernel its last free registers
  // (139 spills), so they are only compiled into the variant that runs while mom6x_continuity_stats is switched on.
  const bool stats = (c->cont_stats != nullptr) && c->cont_stats_on;
  E.stats = stats ? c->cont_stats : nullptr;
  // store_pairs writes 16 bytes at (array + row + k slab 8 + (i0 + ioff) 8), i0 + ioff a multiple of 4: aligned when the slab is even
  // and the arrays start on 16 bytes (the context's own arrays do; an array a host hands in need not)
  E.no_pairs = ((d_slab_of(c) & 1) || (((uintptr_t)A.uh | (uintptr_t)A.u_cor | (uintptr_t)E.h_face) & 15)) ? 1 : 0;
#ifdef MOM6X_MFL_TIMING
  const size_t lds_bytes = sizeof(double) * (NF / 4) * (size_t)(SEG * (ST::NS3 * KL * MAXL + 16)) + 128;   // + the phase times
#else
  const size_t lds_bytes = sizeof(double) * (NF / 4) * (size_t)(SEG * (ST::NS3 * KL * MAXL + 16));   // 4 x the kernel's WAVE_LDS
#endif
  // The launches of an RK2 step with the reference's defaults run the kernel compiled for their switches (struct Sw); the Newton
  // statistics and everything else the general one.  MOM6X_MFW_SPEC=0 (tests/test_switches_gpu.py): always the general one.
  static const int spec_env = [] { const char *e = getenv("MOM6X_MFW_SPEC"); return e ? atoi(e) : 1; }();
  int spec = 0;
  constexpr bool HAS_SPEC = (MAXL >= 3 && MAXL <= 5);   // (the specialised kernels exist for 33..80 layers)
  if (spec_env && !stats && HAS_SPEC && E.scheme == 0 && !E.monotonic && E.marginal && A.better_iter && A.use_visc_rem_max && A.visc_rem) {
    const bool bt = A.set_BT_cont && E.h_face, cor = A.uhbt && A.u_cor;
    if (bt && !A.uhbt && !A.u_cor) spec = 1;
    else if (bt && cor) spec = 2;
    else if (!A.set_BT_cont && !E.h_face && cor) spec = 3;
  }
  auto kern = stats ? k_mass_flux_wave<DIR, MAXL, true, 0, false> : k_mass_flux_wave<DIR, MAXL, false, 0, false>;
  if (E.fma) kern = stats ? k_mass_flux_wave<DIR, MAXL, true, 0, true> : k_mass_flux_wave<DIR, MAXL, false, 0, true>;
  if (HAS_SPEC) {
    constexpr int S1 = HAS_SPEC ? 1 : 0, S2 = HAS_SPEC ? 2 : 0, S3 = HAS_SPEC ? 3 : 0;   // (keeps the other instantiations of launch() from instantiating them)
    if (spec == 1) kern = E.fma ? k_mass_flux_wave<DIR, MAXL, false, S1, true> : k_mass_flux_wave<DIR, MAXL, false, S1, false>;
    if (spec == 2) kern = E.fma ? k_mass_flux_wave<DIR, MAXL, false, S2, true> : k_mass_flux_wave<DIR, MAXL, false, S2, false>;
    if (spec == 3) kern = E.fma ? k_mass_flux_wave<DIR, MAXL, false, S3, true> : k_mass_flux_wave<DIR, MAXL, false, S3, false>;
  }
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  long strips_rows = 0;
  for (int q = 0; q < E.np; q++) {
    E.pgx[q] = (E.pa1[q] - E.pib[q] + NF) / NF;
    strips_rows += (long)E.pgx[q] * (E.pb1[q] - E.pb0[q] + 1);
  }
  // Rows a work-group marches over: 16 where the grid is many times the chip (1440 x 1080: 6120 work-groups for 512 resident
  // ones).  The work-groups differ in their work (Newton sweeps per row, land) and nothing but the dispatcher balances them, so a
  // launch wants several work-groups per resident slot: on the tile of an 8-GPU layout (360 x 540) 16 rows make 782 work-groups =
  // 1.5 rounds, the second one half empty -- measured there (profiles/r04_mfw_variants.md): 4-10 rows 0.36-0.38 ms per zonal launch,
  // 16 rows 0.41, one round of 25 rows 0.45.  The prologue of a work-group (first DMA; meridional: the ring of h rows and the
  // row that is only reconstructed) is cheap next to that.
  {
    static int slots_cache[2][2][5] = {{{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}, {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}};
    int &slots = slots_cache[DIR][E.fma ? 1 : 0][stats ? 4 : spec];   // (per instantiation of launch(): MAXL is a template parameter)
    if (!slots) {
      int per_cu = 0, ncu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NF * KL, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
      slots = per_cu * ncu;
    }
    const long want = 4L * slots;                                  // work-groups for four rounds
    const long r = strips_rows / want;                             // the march length that gives them
    const int rmin = DIR ? 6 : 4;
    E.rows = (int)std::min<long>(16, std::max<long>(rmin, r));
  }
  int nwg = 0;
  for (int q = 0; q < E.np; q++) { E.pgy[q] = (E.pb1[q] - E.pb0[q] + E.rows) / E.rows; nwg += E.pgx[q] * E.pgy[q]; }
  const dim3 grid(nwg, 1, 1);
  if (c->prof_on) prof_begin(c, DIR ? "k_mass_flux_wave<1>" : "k_mass_flux_wave<0>");
  hipLaunchKernelGGL(kern, grid, dim3(NF * KL, 1, 1), lds_bytes, c->stream, c->d, c->G, A, E);
  if (c->prof_on) prof_end(c);
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

}  // namespace

#ifdef MOM6X_MFL_TIMING
extern "C" int mom6x_debug_mfw_timing(unsigned long long *out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mfw_t), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfw_t), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

#if MOM6X_MFW_ONEWAY_ALL == 2
extern "C" int mom6x_debug_mfw_oneway(unsigned long long *out2, int reset) {
  if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_mfw_ow), sizeof(unsigned long long) * 2) != hipSuccess) return 1;
  if (reset) { unsigned long long z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mfw_ow), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

bool mass_flux_wave_usable(int nk) { return nk <= 8 * KL; }

int mass_flux_wave_pair(mom6x_ctx *c, int dir, const FluxArgs &A, const FluxArgs &A2, const LdsArgs &E0) {
  const int nk = c->d.nk;
  LdsArgs E = E0;
  E.np = 0;
  for (const FluxArgs *P : { &A, &A2 }) {
    if (P->a0 > P->a1 || P->b0 > P->b1) continue;
    const int q = E.np++;
    E.pa0[q] = P->a0; E.pa1[q] = P->a1; E.pb0[q] = P->b0; E.pb1[q] = P->b1;
    E.pib[q] = P->a0 - (((P->a0 + c->d.ioff) % NF) + NF) % NF;   // (pib + ioff) is a multiple of 16 doubles = 128 B
  }
  if (E.np == 0) return MOM6X_OK;
  E.i_base = E.pib[0];
  const int maxl = (nk + KL - 1) / KL;
#define GO(D, M) return launch<D, M>(c, A, E)
  if (dir == 0) { if (maxl <= 2) GO(0, 2); if (maxl <= 3) GO(0, 3); if (maxl <= 4) GO(0, 4); if (maxl <= 5) GO(0, 5); GO(0, 8); }
  if (maxl <= 2) GO(1, 2); if (maxl <= 3) GO(1, 3); if (maxl <= 4) GO(1, 4); if (maxl <= 5) GO(1, 5); GO(1, 8);
#undef GO
}

int mass_flux_wave(mom6x_ctx *c, int dir, const FluxArgs &A, const LdsArgs &E0) {
  FluxArgs none = A;
  none.a0 = 0; none.a1 = -1;
  return mass_flux_wave_pair(c, dir, A, none, E0);
}
