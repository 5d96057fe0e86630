// mom6x_dev.h -- device-side helpers shared by all HIP kernels of the dycore (gfx950 only).
//
// Layout: see include/mom6x.h.  Every kernel gets the tile dims by value (`Dm`), a pointer to
// the metric block and plain FP64 device pointers.  i is the coalesced (lane) index in every
// kernel, exactly as the reference's inner `do i` loops; columns (k) are walked per thread.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/mom6x.h"

struct Dm {            // device copy of the layout part of mom6x_dims
  int ni, nj, nk, halo, ioff, joff, pitch, slab;
};

__host__ __device__ inline Dm make_dm(const mom6x_dims &d) {
  Dm m; m.ni = d.ni; m.nj = d.nj; m.nk = d.nk; m.halo = d.halo; m.ioff = d.ioff; m.joff = d.joff;
  m.pitch = d.pitch; m.slab = d.slab; return m;
}

__device__ __forceinline__ size_t ix2(const Dm &d, int i, int j) {
  return (size_t)(i + d.ioff) + (size_t)(j + d.joff) * (size_t)d.pitch;
}
__device__ __forceinline__ size_t ix3(const Dm &d, int i, int j, int k) {
  return ix2(d, i, j) + (size_t)k * (size_t)d.slab;
}
__device__ __forceinline__ const double *gm(const double *G, const Dm &d, int m) {
  return G + (size_t)m * (size_t)d.slab;
}

__device__ __forceinline__ double dmax(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return (a < b) ? a : b; }
__device__ __forceinline__ double dsign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

// ---------------------------------------------------------------------------------------------
// host-side error plumbing
void mom6x_set_error(const char *fmt, ...);
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      mom6x_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MOM6X_EHIP;                                                                   \
    }                                                                                      \
  } while (0)
#define REQUIRE(cond, code, msg)                  \
  do {                                            \
    if (!(cond)) { mom6x_set_error("%s", msg); return code; } \
  } while (0)

// ---------------------------------------------------------------------------------------------
// The context (opaque to C callers).
#define MOM6X_NSCR 12
enum Scr { SCR_q = 0, SCR_KE, SCR_absv, SCR_e, SCR_c1, SCR_t0, SCR_t1, SCR_t2, SCR_t3 };
struct BTState;   // barotropic.hip
struct RK2State;  // dyn_split_RK2.hip

// u_bc_accel = (CAu + PFu) + diffu of the RK2 predictor (RK2.F90:565-572) formed by the pressure-force kernel that has just made
// PFu (all null: PressureForce on its own)
struct BcFold { const double *CAu, *CAv, *diffu, *diffv; double *u_bc, *v_bc; double *eta_h; };   // eta_h: bt_mass_source's column sum (h summed top-down, less the depth), a by-product
struct mom6x_ctx {
  mom6x_dims dims;
  Dm d;
  int device;
  hipStream_t stream;       // compute stream
  hipStream_t halo_stream;  // halo pack / RCCL send-recv / unpack
  double *G;                // device metric block [MOM6X_G_COUNT][slab]
  mom6x_vgrid GV;
  int first_direction;
  mom6x_continuity_params cont; bool cont_init;
  mom6x_barotropic_params bt; bool bt_init;
  // continuity scratch: edge values of the PPM reconstruction for one direction at a time
  double *hL, *hR;
  int *retry; size_t retry_cap;   // continuity_lds.hip: tiles whose Newton solve needs the exact CFL limits
  BTState *bts;
  RK2State *rk2;
  int *flag;                // device-side error flag (NaN / negative thickness)
  // MOM_CoriolisAdv / MOM_PressureForce / MOM_vert_friction
  mom6x_coriolis_params cor; bool cor_init;
  mom6x_pgf_params pgf; bool pgf_init;
  double *Rlay, *g_prime;   // device copies of GV%Rlay, GV%g_prime (nk)
  const double *tv_T, *tv_S; mom6x_eos_params eos;   // tv%T, tv%S, tv%eqn_of_state (null: layered PressureForce path)
  const double *a_u, *a_v, *h_u, *h_v, *Ray_u, *Ray_v;   // vertvisc coefficients (host-owned device arrays, or vv_* below)
  // vertvisc_init / vertvisc_coef on the device: parameters, vertvisc_type inputs, CS%a_u.. owned by the context
  mom6x_vertvisc_params vv; bool vv_init;
  const double *Kv_bbl_u, *Kv_bbl_v, *bbl_thick_u, *bbl_thick_v, *Kv_shear;
  double *vv_a_u, *vv_a_v, *vv_h_u, *vv_h_v;
  // hor_visc.hip: hor_visc_CS parameters and the 2-D coefficient planes of hor_visc_init
  mom6x_hor_visc_params hv; bool hv_init; double *hv_planes;
  double ds_Hmix; const double *ds_h;   // DIRECT_STRESS: HMIX_STRESS (0 = off) and vertvisc's h argument
  double *regrid_res;       // remap.hip: coordinateResolution of the z* coordinate (nk)
  double *remap_hvel;       // remap.hip: h_u, h_v of both grids for mom6x_ALE_remap_velocities_from_h outside OM4's switch set (allocated on first use)
  double *remap_src;        // remap.hip: the un-remapped velocity of ALE_remap_velocities' KE-conserving correction (allocated on first use)
  double *regrid_vec;       // remap.hip: the host vectors of the density coordinates (resolution | targets | max depths | max thicknesses)
  // lazily allocated 3-D scratch arrays (slot -> nlev levels)
  double *scr[MOM6X_NSCR]; int scr_nlev[MOM6X_NSCR];
  bool prof_on;
  struct Prof *prof;
  void *ta;                 // tracer.hip: tracer-advection state (TAState)
  long long *red; size_t red_cap;   // diag_sums.hip: integer accumulators of the reproducing sums / checksums
  void *diag;               // diag_sums.hip: Sum_output_CS state (depth list, lH) of write_energy
  void *comm;               // halo.hip: tile layout + RCCL communicator (null: single tile, wrap only)
  bool bt_overlap;          // btstep's own group pass of eta, ubt, vbt overlapped with the own-points half of the next sub-step (mom6x_comm_overlap_btstep)
  bool halo_error;          // set by a failed halo exchange inside a stream-ordered sequence
  hipEvent_t ev_ready, ev_done; bool pass_pending;   // halo.hip: the group pass in flight on the halo stream (start_/complete_group_pass)
  // the RK2 step's h_av formed by the convergence kernels of a continuity call (continuity.hip k_convergence): kind 1: h_av =
  // 0.5 * (hin + h) (RK2.F90:808-810); 2: h_av = 0.5 * (h_old + h_new) of an in-place call (:1025-1027 + :1064-1066); 0: off
  int cont_av_kind; double *cont_av; const double *cont_av_src;
  // halo width of the NEXT pass (0: the context's); dyn_pass_width: what the RK2 step sets for its own group passes -- a context
  // whose halo was widened for the barotropic solver (BTHALO) still sends NIHALO rows of the 3-D fields (mom6x_set_dyn_pass_width)
  int pass_w = 0, dyn_pass_width = 0;
  // ... and, field by field, the widths create_group_pass(..., halo=) gave the fields of the NEXT pass (pass_wf_n of them; 0 entries
  // and fields beyond pass_wf_n: pass_w); the RK2 step sends the reference's own 2-3 rows instead of NIHALO = 4 (RK2.F90:484-495)
  int pass_wf[16] = {0}, pass_wf_n = 0;
  long long n_exchanges = 0, n_exchange_bytes = 0;
  BcFold pgf_fold = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  bool pgf_eta_h_written = false;                                     // the last PressureForce call filled pgf_fold.eta_h   // set by the RK2 step around its PressureForce call
  bool cont_stats_on;               // the statistics-collecting (slower) variant of the mass-flux kernel is in use
  unsigned long long *cont_stats;   // device: Newton statistics of the wave-owned mass-flux kernel (mom6x_continuity_stats), or null
  bool cont_h_unused;       // the caller of continuity_PPM does not look at the new thicknesses (RK2.F90:646: hp is overwritten at :781
                            // before anybody reads it): the convergence of the second direction is not computed
};
void comm_free(mom6x_ctx *c);                                         // halo.hip
void halo_start(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);   // start_group_pass
void halo_complete(mom6x_ctx *c);                                     // complete_group_pass
bool halo_start_packed(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);   // ... packed on the compute stream
bool halo_can_overlap(const mom6x_ctx *c);
int comm_allreduce_scalar(mom6x_ctx *c, double *value, int op);       // halo.hip: 0 min, 1 max, 2 sum

// ---------------------------------------------------------------------------------------------
// Per-kernel timing with HIP events on the compute stream (mom6x_prof_* in include/mom6x.h).
// Off by default; when on, every KLAUNCH is bracketed by two events from a pool.
void prof_begin(mom6x_ctx *c, const char *name);
void prof_end(mom6x_ctx *c);
#define KLAUNCH(c, name, kern, grid, blk, ...)                              \
  do {                                                                      \
    if ((c)->prof_on) prof_begin((c), name);                                \
    hipLaunchKernelGGL(kern, grid, blk, 0, (c)->stream, __VA_ARGS__);       \
    if ((c)->prof_on) prof_end((c));                                        \
  } while (0)

#define KLAUNCH_LDS(c, name, kern, grid, blk, lds, ...)                     \
  do {                                                                      \
    if ((c)->prof_on) prof_begin((c), name);                                \
    hipLaunchKernelGGL(kern, grid, blk, lds, (c)->stream, __VA_ARGS__);     \
    if ((c)->prof_on) prof_end((c));                                        \
  } while (0)

int ctx_scratch(mom6x_ctx *c, int slot, int nlev, double **out);
int work_fill_byte();   // 0, or 0xFF with MOM6X_POISON_WORK=1 (ctx.hip): the initial contents of work arrays   // ctx.hip
void hor_visc_free(mom6x_ctx *c);                                  // hor_visc.hip
void diag_sums_free(mom6x_ctx *c);                                 // diag_sums.hip
// dyn_kernels.hip: vertvisc_coef looking at u (mode 0), mask*(u + dtx*u_bc) (1) or mask*(u + dtx*(u_bc + u_abt)) (2)
int CorAdCalc_bc(mom6x_ctx *c, const double *u, const double *v, const double *h, const double *uh, const double *vh, double *CAu,
                 double *CAv, const double *PFu, const double *PFv, const double *diffu, const double *diffv, double *u_bc,
                 double *v_bc, double *uhtr, double *vhtr, double dt_tr);   // dyn_kernels.hip
int vertvisc_coef_upd(mom6x_ctx *c, int mode, const double *u, const double *v, const double *u_bc, const double *v_bc,
                      const double *u_abt, const double *v_abt, double dtx, const double *h, double dt, double *u_out, double *v_out);
// btstep_layer_accel (MOM_barotropic.F90:3432-3504) evaluated by the CONSUMER of accel_layer_u / _v instead of being written to
// HBM and read back: what a face column needs of the barotropic solver's 2-D results (barotropic.hip keeps them in its work block
// until the next btstep).  mode 3 of vertvisc_coef_upd = mode 2 with u_abt formed from these.
struct LayerAccelSrc {
  const double *pbce;      // 3-D
  const double *e_anom;    // 2-D, h points
  const double *g_own;     // gtot_E | gtot_N at the cell of the face's own index
  const double *g_nbr;     // gtot_W | gtot_S at the next cell
  const double *a2d;       // u_accel_bt | v_accel_bt (2-D)
  double underflow;        // accel_underflow = vel_underflow / dt
};
bool vertvisc_coef_solve_usable(mom6x_ctx *c);                           // dyn_kernels.hip: k_vertvisc_coef_cols exists for this configuration
int vertvisc_coef_solve_la(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc,
                           const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx, const double *h, double dt_coef,
                           double *u, double *v, const double *taux, const double *tauy, double dt, double *taux_bot, double *tauy_bot,
                           double *vr_u, double *vr_v, bool keep_coef);
int vertvisc_coef_remnant(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc, double dtx,
                          const double *h, double dt_coef, double *vr_u, double *vr_v, double dt, bool keep_coef);
int bt_mass_source_from(mom6x_ctx *c, const double *eta_h, const double *eta, int set_cor);   // bt_mass_source with the column sum given
int set_dtbt_eta(mom6x_ctx *c, const double *pbce, const double *eta);      // barotropic.hip: set_dtbt(pbce, eta=eta) RK2.F90:667
void bt_defer_btcalc(mom6x_ctx *c, bool on);                            // barotropic.hip: btcalc's fractions formed by btstep's column pass while on
int bt_frhat_materialize(mom6x_ctx *c);                                 // writes frhatu / frhatv if a deferred btcalc is pending
void bt_defer_layer_accel(mom6x_ctx *c, bool on);                       // barotropic.hip: btstep skips k_layer_accel while on
bool bt_layer_accel_src(mom6x_ctx *c, LayerAccelSrc *u, LayerAccelSrc *v);   // false if no deferred result is pending
int bt_layer_accel_materialize(mom6x_ctx *c, double *accel_layer_u, double *accel_layer_v);
int vertvisc_coef_upd_la(mom6x_ctx *c, const double *u, const double *v, const double *u_bc, const double *v_bc,
                         const LayerAccelSrc &LAu, const LayerAccelSrc &LAv, double dtx, const double *h, double dt, double *u_out,
                         double *v_out);
// dyn_kernels.hip: [u = mask*(u_in + dtx*(u_bc + u_abt));] vertvisc(u, v, dt); [vertvisc_remnant(vr_u, vr_v, dt)] in one sweep
int vertvisc_fused(mom6x_ctx *c, const double *u_in, const double *v_in, const double *u_bc, const double *v_bc,
                   const double *u_abt, const double *v_abt, double dtx, double *u, double *v, const double *taux,
                   const double *tauy, double dt, double *taux_bot, double *tauy_bot, double *vr_u, double *vr_v);

// k-chunking for "column-walk" kernels: a thread keeps its 2-D coefficients in registers and walks
// KCHUNK consecutive layers, so the 2-D metric planes are read nk/KCHUNK times instead of nk times.
#define KCHUNK 15
inline int nchunks(int nk) { return (nk + KCHUNK - 1) / KCHUNK; }

// The i-parallel kernels whose first index I0 is negative (u-points start at -1, widened loops at -2, -3) start
// their blocks at i = -IAL, a 128-byte line of the pitched rows (ioff = 16): every wavefront then reads and writes
// whole cache lines instead of sharing its first and last line with the neighbouring block (which usually runs
// on another XCD, i.e. behind another L2).  Lanes below I0 exit.  nxa() is the matching grid extent.
#define IAL 16
#define I_BASE(I0) (((I0) < 0) ? -IAL : (I0))
inline int nxa(int nx, int I0) { return (I0 < 0) ? nx + IAL + I0 : nx; }

inline dim3 grid3(int nx, int ny, int nz, dim3 b) {
  return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y, (nz + b.z - 1) / b.z);
}

// The kernels that keep a whole column on chip (registers + LDS: k_vertvisc_coef_cols, k_vertvisc_cols, k_vertvisc_remnant_cols,
// k_tridiag_cols, k_btcalc_cols, k_regrid_zstar_cols) unroll the column at compile time.  Their template argument NKT is the layer
// count itself (NKT > 0: 75, the headline's) or, negated, a BOUND on it: the register arrays and unrolled loops have -NKT slots and
// a wavefront-uniform test `k < nk` skips the ones beyond the column, so any nk <= -NKT runs the same code.
#define NK_OF(NKT) ((NKT) > 0 ? (NKT) : -(NKT))
#define NK_EXACT(NKT) ((NKT) > 0)
constexpr int COLS_NK_BOUND = 76;   // (k_vertvisc_coef_cols holds 4 x NK registers of column state next to ~190 others: 76 slots fit the 512 of a wavefront alone on its SIMD,
                                    //  78 spill, at 80 the coefficient column stays in scratch; k_regrid_zstar_cols spills at 80 too)
// (three bounds, so that a column fills at least 4/5 of the slots it pays for once it has more than 41 layers.  Not 64: with a bound
//  that is a multiple of the walk's group size, k_vertvisc_coef_cols' switch over the group pairs (cc_file_switch) is affine in the
//  pair's number, the compiler turns it into one store with a computed index, and the coefficient column lives in scratch memory.)
#define COLS_NK_DISPATCH(nk, CALL) do { if ((nk) == 75) { CALL(75); } else if ((nk) <= 52) { CALL(-52); } else if ((nk) <= 66) { CALL(-66); } \
                                        else { CALL(-COLS_NK_BOUND); } } while (0)
