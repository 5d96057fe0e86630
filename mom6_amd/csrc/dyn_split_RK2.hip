// dyn_split_RK2.hip -- the split-explicit RK2 baroclinic step, device-resident and stream-ordered.
//
// Replaces step_MOM_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:294-1205) and the new-run fills of
// initialize_dyn_split_RK2 (:1577-1650).  The whole step is enqueued on the context's compute stream
// with no host synchronisation except inside the optional host callbacks (vertvisc_coef,
// horizontal_viscosity: callees that are not on the ported hot path) and set_dtbt's scalar reduction.
// Every `[halo]` mark of the reference (do_group_pass) is a halo_update() call; on one tile that is
// the periodic wrap kernel, on several tiles the RCCL exchange (halo.hip).
#include <algorithm>
#include <cstring>
#include "mom6x_dev.h"

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n);

struct RK2State {
  mom6x_rk2_params P;
  // MOM_dyn_split_RK2_CS arrays (RK2.F90:85-273)
  double *CAu, *CAv, *CAu_pred, *CAv_pred, *PFu, *PFv, *diffu, *diffv, *visc_rem_u, *visc_rem_v;
  double *u_accel_bt, *v_accel_bt, *u_av, *v_av, *h_av, *pbce;
  double *eta, *eta_PF, *uhbt, *vhbt, *taux_bot, *tauy_bot;
  mom6x_BT_cont BT;
  bool CAu_pred_stored;
  bool accel_bt_deferred;   // u_accel_bt, v_accel_bt have not been written: the corrector's btstep results wait in the work block
  // the routine's stack temporaries :341-357
  double *up, *vp, *hp, *u_bc_accel, *v_bc_accel, *uh_in, *vh_in, *eta_pred, *eta_h;
};

namespace {

inline dim3 blk2() { return dim3(64, 4, 1); }
inline dim3 gridk(int nx, int ny, int nk, dim3 b) { return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y, nchunks(nk)); }

// u_bc_accel = (CAu + PFu) + diffu  (:565-572, :900-907) and optionally up = mask*(u + dt*u_bc_accel) (:591-598)
__global__ void __launch_bounds__(256)
k_bc_accel(Dm d, const double *__restrict__ G, const double *__restrict__ CAu, const double *__restrict__ CAv,
           const double *__restrict__ PFu, const double *__restrict__ PFv, const double *__restrict__ diffu,
           const double *__restrict__ diffv, double *__restrict__ u_bc, double *__restrict__ v_bc,
           const double *__restrict__ u, const double *__restrict__ v, double *__restrict__ up, double *__restrict__ vp,
           double dt) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < (-1)) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_u = (j >= 0), do_v = (i >= 0);
  const double mCu = gm(G, d, MOM6X_G_mask2dCu)[x], mCv = gm(G, d, MOM6X_G_mask2dCv)[x];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    if (do_u) {
      const double a = (CAu[c] + PFu[c]) + diffu[c];
      u_bc[c] = a;
      if (up) up[c] = mCu * (u[c] + dt * a);
    }
    if (do_v) {
      const double a = (CAv[c] + PFv[c]) + diffv[c];
      v_bc[c] = a;
      if (vp) vp[c] = mCv * (v[c] + dt * a);
    }
  }
}

// out = mask * (u + dtx * (bc_accel + accel_bt))   (:681-694, :957-966); out may alias u
__global__ void __launch_bounds__(256)
k_vel_update(Dm d, const double *__restrict__ G, const double *u, const double *v, const double *__restrict__ u_bc,
             const double *__restrict__ v_bc, const double *__restrict__ u_abt, const double *__restrict__ v_abt,
             double *uo, double *vo, double dtx) {
  const int i = I_BASE(-1) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -1 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  if (i < (-1)) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_u = (j >= 0), do_v = (i >= 0);
  const double mCu = gm(G, d, MOM6X_G_mask2dCu)[x], mCv = gm(G, d, MOM6X_G_mask2dCv)[x];
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    if (do_v) vo[c] = mCv * (v[c] + dtx * (v_bc[c] + v_abt[c]));
    if (do_u) uo[c] = mCu * (u[c] + dtx * (u_bc[c] + u_abt[c]));
  }
}

// h_av updates on (is-2..ie+2, js-2..je+2): mode 0: 0.5*(a+b) (:808-810); 1: copy a (:1025-1027);
// 2: 0.5*(h_av + a) (:1064-1066); 3: hp = (1-w)*a + w*hp on (is-1..ie+1) (:828-830)
__global__ void __launch_bounds__(256)
k_h_av(Dm d, double *h_av, const double *__restrict__ a, const double *__restrict__ b, int mode, double w, int ext, int part) {
  const int i = I_BASE(-ext) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -ext + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 + ext || j > d.nj - 1 + ext) return;
  if (i < (-ext)) return;
  // part 1: the tile's own cells (no halo value is read: safe while a group pass of a, b is in flight); part 2: the frame
  // of `ext` halo cells around them, after the pass has completed; 0: both
  const bool own = (i >= 0 && i <= d.ni - 1 && j >= 0 && j <= d.nj - 1);
  if ((part == 1 && !own) || (part == 2 && own)) return;
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    if (mode == 0) h_av[c] = 0.5 * (a[c] + b[c]);
    else if (mode == 1) h_av[c] = a[c];
    else if (mode == 2) h_av[c] = 0.5 * (h_av[c] + a[c]);
    else h_av[c] = (1.0 - w) * a[c] + w * h_av[c];
  }
}

// uhtr += uh*dt, vhtr += vh*dt  :1072-1079
__global__ void __launch_bounds__(256)
k_uhtr(Dm d, double *__restrict__ uhtr, double *__restrict__ vhtr, const double *__restrict__ uh,
       const double *__restrict__ vh, double dt, int ring_only) {
  const int i = I_BASE(-3) + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = -3 + blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni + 1 || j > d.nj + 1) return;
  if (i < (-3)) return;
  if (ring_only && i >= -1 && i <= d.ni - 1 && j >= -1 && j <= d.nj - 1) return;   // the box k_corad_acc has done (CorAdCalc_bc)
  const size_t x = ix2(d, i, j), slab = (size_t)d.slab;
  const int k0 = blockIdx.z * KCHUNK, k1 = min(k0 + KCHUNK, d.nk);
  const bool do_u = (j >= -2), do_v = (i >= -2);
  for (int k = k0; k < k1; k++) {
    const size_t c = x + (size_t)k * slab;
    if (do_u) uhtr[c] = uhtr[c] + uh[c] * dt;
    if (do_v) vhtr[c] = vhtr[c] + vh[c] * dt;
  }
}

// eta = eta_pred on the computational domain :946 ; or eta = sum_k h - Z_to_H*bathyT :1577-1590
__global__ void k_eta(Dm d, const double *__restrict__ G, double *eta, const double *__restrict__ src,
                      const double *__restrict__ h, double Z_to_H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > d.ni - 1 || j > d.nj - 1) return;
  const size_t x = ix2(d, i, j);
  if (src) { eta[x] = src[x]; return; }
  double e = -Z_to_H * gm(G, d, MOM6X_G_bathyT)[x];
  for (int k = 0; k < d.nk; k++) e = e + h[x + (size_t)k * d.slab];
  eta[x] = e;
}

int pass3(mom6x_ctx *c, std::initializer_list<double *> f, std::initializer_list<int> stg, int nk) {
  double *ff[16]; int ss[16], nn[16]; int n = 0;
  auto s = stg.begin();
  for (double *p : f) { ff[n] = p; ss[n] = *s++; nn[n] = nk; n++; }
  c->pass_w = c->dyn_pass_width;
  halo_wrap(c, ff, ss, nn, n);
  c->pass_w = 0;
  return MOM6X_OK;
}

// The widths the reference gives its group passes (create_group_pass(..., halo=), RK2.F90:476-495): cont = continuity_stencil
// (continuity_PPM.F90:2757: 3, SIMPLE_2ND 2, UPWIND_1ST 1), vel = max(2, hor_visc_vel_stencil) (hor_visc.F90:3305: 3 with a Leith
// viscosity).  The device sent NIHALO = 4 rows of everything through round 3; MOM6X_PASS_WIDTHS=full still does.
struct PassWidths { int cont, vel; };
PassWidths pass_widths(const mom6x_ctx *c) {
  static const bool full = [] { const char *e = getenv("MOM6X_PASS_WIDTHS"); return e && !strcmp(e, "full"); }();
  if (full) return PassWidths{ 0, 0 };
  const int cont = c->cont.upwind_1st ? 1 : (c->cont.simple_2nd ? 2 : 3);
  const int vel = (c->hv_init && (c->hv.Leith_Kh || c->hv.Leith_Ah)) ? 3 : 2;
  return PassWidths{ cont, vel };
}
void set_widths(mom6x_ctx *c, std::initializer_list<int> w) {
  c->pass_wf_n = 0;
  for (int x : w) { if (c->pass_wf_n < 16) c->pass_wf[c->pass_wf_n++] = x; }
}

// start_group_pass of 3-D fields on the halo stream (G%nonblocking_updates); halo_complete() is its complete_group_pass
int start3(mom6x_ctx *c, std::initializer_list<double *> f, std::initializer_list<int> stg, int nk, std::initializer_list<int> widths = {}) {
  double *ff[16]; int ss[16], nn[16]; int n = 0;
  auto s = stg.begin();
  for (double *p : f) { ff[n] = p; ss[n] = *s++; nn[n] = nk; n++; }
  c->pass_w = c->dyn_pass_width;
  set_widths(c, widths);
  halo_start(c, ff, ss, nn, n);
  c->pass_w = 0; c->pass_wf_n = 0;
  return MOM6X_OK;
}

int startn(mom6x_ctx *c, std::initializer_list<double *> f, std::initializer_list<int> stg, std::initializer_list<int> nks,
           std::initializer_list<int> widths = {}) {
  double *ff[16]; int ss[16], nn[16]; int n = 0;
  auto s = stg.begin(); auto q = nks.begin();
  for (double *p : f) { ff[n] = p; ss[n] = *s++; nn[n] = *q++; n++; }
  c->pass_w = c->dyn_pass_width;   // (mom6x_set_dyn_pass_width: NIHALO rows of the 3-D fields in a context widened for BTHALO)
  set_widths(c, widths);
  halo_start(c, ff, ss, nn, n);
  c->pass_w = 0; c->pass_wf_n = 0;
  return MOM6X_OK;
}

// One group pass of fields with different numbers of levels: consecutive do_group_pass calls of the reference
// that no computation separates are sent as ONE packed message per neighbour.
int passn(mom6x_ctx *c, std::initializer_list<double *> f, std::initializer_list<int> stg, std::initializer_list<int> nks) {
  double *ff[16]; int ss[16], nn[16]; int n = 0;
  auto s = stg.begin(); auto q = nks.begin();
  for (double *p : f) { ff[n] = p; ss[n] = *s++; nn[n] = *q++; n++; }
  c->pass_w = c->dyn_pass_width;
  halo_wrap(c, ff, ss, nn, n);
  c->pass_w = 0;
  return MOM6X_OK;
}

}  // namespace

void rk2_state_free(mom6x_ctx *c) {
  if (!c->rk2) return;
  RK2State *s = c->rk2;
  (void)bt_frhat_materialize(c);   // (a deferred btcalc points into BT.h_u / BT.h_v, which go away here)
  double *p3[] = { s->CAu, s->CAv, s->CAu_pred, s->CAv_pred, s->PFu, s->PFv, s->diffu, s->diffv, s->visc_rem_u, s->visc_rem_v,
                   s->u_accel_bt, s->v_accel_bt, s->u_av, s->v_av, s->h_av, s->pbce, s->up, s->vp, s->hp, s->u_bc_accel,
                   s->v_bc_accel, s->uh_in, s->vh_in, s->BT.h_u, s->BT.h_v, s->eta, s->eta_PF, s->uhbt, s->vhbt, s->taux_bot,
                   s->tauy_bot, s->eta_pred, s->eta_h, s->BT.FA_u_EE, s->BT.FA_u_E0, s->BT.FA_u_W0, s->BT.FA_u_WW, s->BT.uBT_WW,
                   s->BT.uBT_EE, s->BT.FA_v_NN, s->BT.FA_v_N0, s->BT.FA_v_S0, s->BT.FA_v_SS, s->BT.vBT_SS, s->BT.vBT_NN };
  for (double *p : p3) (void)hipFree(p);
  delete s;
  c->rk2 = nullptr;
}

extern "C" int mom6x_initialize_dyn_split_RK2(mom6x_ctx *c, const mom6x_rk2_params *p) {
  REQUIRE(c && p, MOM6X_EINVAL, "mom6x_initialize_dyn_split_RK2: null argument");
  REQUIRE(c->cont_init && c->bt_init && c->cor_init && c->pgf_init, MOM6X_EINVAL,
          "initialize_dyn_split_RK2: continuity, barotropic, CoriolisAdv and PressureForce must be initialised first");
  REQUIRE(!p->remap_aux || p->store_CAu, MOM6X_EINVAL, "REMAP_AUXILIARY_VARS requires that STORE_CORIOLIS_ACCEL = True.");   // :1474
  REQUIRE(p->BT_use_layer_fluxes && p->store_CAu, MOM6X_EUNSUPPORTED,
          "dyn_split_RK2: only BT_USE_LAYER_FLUXES=True and STORE_CORIOLIS_ACCEL=True are supported");
  HIPCHK(hipSetDevice(c->device));
  if (c->rk2) rk2_state_free(c);
  RK2State *s = new RK2State();
  memset(s, 0, sizeof(*s));
  s->P = *p;
  const size_t n2 = (size_t)c->dims.slab, n3 = n2 * c->dims.nk;
  double **p3[] = { &s->CAu, &s->CAv, &s->CAu_pred, &s->CAv_pred, &s->PFu, &s->PFv, &s->diffu, &s->diffv, &s->visc_rem_u,
                    &s->visc_rem_v, &s->u_accel_bt, &s->v_accel_bt, &s->u_av, &s->v_av, &s->h_av, &s->pbce, &s->up, &s->vp,
                    &s->hp, &s->u_bc_accel, &s->v_bc_accel, &s->uh_in, &s->vh_in, &s->BT.h_u, &s->BT.h_v };
  for (double **q : p3) { HIPCHK(hipMalloc(q, n3 * sizeof(double))); HIPCHK(hipMemsetAsync(*q, 0, n3 * sizeof(double), c->stream)); }
  double **p2[] = { &s->eta, &s->eta_PF, &s->uhbt, &s->vhbt, &s->taux_bot, &s->tauy_bot, &s->eta_pred, &s->eta_h, &s->BT.FA_u_EE,
                    &s->BT.FA_u_E0, &s->BT.FA_u_W0, &s->BT.FA_u_WW, &s->BT.uBT_WW, &s->BT.uBT_EE, &s->BT.FA_v_NN, &s->BT.FA_v_N0,
                    &s->BT.FA_v_S0, &s->BT.FA_v_SS, &s->BT.vBT_SS, &s->BT.vBT_NN };
  for (double **q : p2) { HIPCHK(hipMalloc(q, n2 * sizeof(double))); HIPCHK(hipMemsetAsync(*q, 0, n2 * sizeof(double), c->stream)); }
  s->CAu_pred_stored = false;
  c->rk2 = s;
  return MOM6X_OK;
}

extern "C" double *mom6x_rk2_field(mom6x_ctx *c, int which) {
  if (!c || !c->rk2) return nullptr;
  RK2State *s = c->rk2;
  double *t[] = { s->CAu, s->CAv, s->CAu_pred, s->CAv_pred, s->PFu, s->PFv, s->diffu, s->diffv, s->visc_rem_u, s->visc_rem_v,
                  s->u_accel_bt, s->v_accel_bt, s->u_av, s->v_av, s->h_av, s->pbce, s->eta, s->eta_PF, s->uhbt, s->vhbt,
                  s->taux_bot, s->tauy_bot, s->BT.h_u, s->BT.h_v };
  if (which < 0 || which >= (int)(sizeof(t) / sizeof(t[0]))) return nullptr;
  if ((which == 10 || which == 11) && s->accel_bt_deferred) {   // CS%u_accel_bt / v_accel_bt asked for: form them now
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (bt_layer_accel_materialize(c, s->u_accel_bt, s->v_accel_bt) != MOM6X_OK) return nullptr;   // (also: no pending result any more)
    if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;   // the caller may read it from another stream
    s->accel_bt_deferred = false;
  }
  return t[which];
}

extern "C" int mom6x_rk2_set_CAu_pred_stored(mom6x_ctx *c, int stored) {
  REQUIRE(c && c->rk2, MOM6X_EINVAL, "mom6x_rk2_set_CAu_pred_stored: dyn_split_RK2 not initialised");
  c->rk2->CAu_pred_stored = (stored != 0);
  return MOM6X_OK;
}

#define CHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// remap_dyn_split_RK2_aux_vars :1302-1330
extern "C" int mom6x_remap_dyn_split_RK2_aux_vars(mom6x_ctx *c, const mom6x_remapping_params *p, const double *h_old_u,
                                                  const double *h_old_v, const double *h_new_u, const double *h_new_v) {
  REQUIRE(c && c->rk2, MOM6X_EINVAL, "remap_dyn_split_RK2_aux_vars: dyn_split_RK2 not initialised");
  RK2State *s = c->rk2;
  if (!s->P.remap_aux) return MOM6X_OK;
  if (s->P.store_CAu) {
    CHK(mom6x_ALE_remap_velocities(c, p, h_old_u, h_old_v, h_new_u, h_new_v, s->u_av, s->v_av));
    CHK(pass3(c, {s->u_av, s->v_av}, {1, 2}, c->d.nk));
    CHK(mom6x_ALE_remap_velocities(c, p, h_old_u, h_old_v, h_new_u, h_new_v, s->CAu_pred, s->CAv_pred));
    CHK(pass3(c, {s->CAu_pred, s->CAv_pred}, {1, 2}, c->d.nk));
  }
  CHK(mom6x_ALE_remap_velocities(c, p, h_old_u, h_old_v, h_new_u, h_new_v, s->diffu, s->diffv));
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  return MOM6X_OK;
}

// initialize_dyn_split_RK2 :1577-1668, field by field: what the restart file did not hold is formed as the reference forms it.
// `have` = the MOM6X_RK2_HAVE_* bits of the variables the host has uploaded (query_initialized was true); 0 = a new run.
extern "C" int mom6x_dyn_split_RK2_restart_fills(mom6x_ctx *c, const double *u, const double *v, const double *h, double *uh,
                                                 double *vh, double dt, int have) {
  REQUIRE(c && c->rk2, MOM6X_EINVAL, "dyn_split_RK2_restart_fills: initialize_dyn_split_RK2 must be called first");
  REQUIRE(u && v && h && uh && vh, MOM6X_EINVAL, "dyn_split_RK2_restart_fills: null mandatory array");
  HIPCHK(hipSetDevice(c->device));
  RK2State *s = c->rk2;
  const Dm d = c->d;
  const size_t n3 = (size_t)d.slab * d.nk;
  const dim3 b = blk2();
  if (!(have & MOM6X_RK2_HAVE_ETA))     // :1578-1590
    KLAUNCH(c, "k_eta", k_eta, grid3(d.ni, d.nj, 1, b), b, d, c->G, s->eta, (const double *)nullptr, h, c->GV.Z_to_H);
  if (!(have & MOM6X_RK2_HAVE_DIFFU) && c->hv_init) CHK(mom6x_horizontal_viscosity(c, u, v, h, s->diffu, s->diffv));   // :1599-1606
  if (!(have & MOM6X_RK2_HAVE_U2)) {    // :1608-1614
    HIPCHK(hipMemcpyAsync(s->u_av, u, n3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(s->v_av, v, n3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  if (have & MOM6X_RK2_HAVE_CAU) {      // :1617-1619
    s->CAu_pred_stored = true;
  } else {
    if ((have & MOM6X_RK2_HAVE_UH) && (have & MOM6X_RK2_HAVE_H2)) {   // :1621-1628: an older file's uh, vh, h2
      pass3(c, { s->h_av }, { 0 }, d.nk);
    } else {
      // h_tmp = h ; continuity(u_av, v_av, h, h_tmp, uh, vh, dt) ; h_av = 0.5*(h + h_tmp)  :1629-1636
      double *h_tmp = s->hp;
      HIPCHK(hipMemcpyAsync(h_tmp, h, n3 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      CHK(mom6x_continuity_PPM(c, s->u_av, s->v_av, h, h_tmp, uh, vh, dt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr));
      pass3(c, { h_tmp }, { 0 }, d.nk);
      KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 2 * d.halo, -d.halo), d.nj + 2 * d.halo, d.nk, b), b, d, s->h_av, h, (const double *)h_tmp, 0, 0.0, d.halo, 0);
    }
    pass3(c, { s->u_av, s->v_av, uh, vh }, { 1, 2, 1, 2 }, d.nk);
    CHK(mom6x_CorAdCalc(c, s->u_av, s->v_av, s->h_av, uh, vh, s->CAu_pred, s->CAv_pred));
    s->CAu_pred_stored = true;
  }
  // :1670-1679 (pass_av_h_uvh): the auxiliary velocities and the stored accelerations with their halos
  if (have & (MOM6X_RK2_HAVE_U2 | MOM6X_RK2_HAVE_CAU)) pass3(c, { s->u_av, s->v_av, s->CAu_pred, s->CAv_pred }, { 1, 2, 1, 2 }, d.nk);
  HIPCHK(hipGetLastError());
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  return MOM6X_OK;
}

extern "C" int mom6x_dyn_split_RK2_new_run(mom6x_ctx *c, const double *u, const double *v, const double *h, double *uh,
                                           double *vh, double dt) {
  return mom6x_dyn_split_RK2_restart_fills(c, u, v, h, uh, vh, dt, 0);
}

extern "C" int mom6x_step_dyn_split_RK2(mom6x_ctx *c, double *u_inst, double *v_inst, double *h, double *uh, double *vh,
                                        double *uhtr, double *vhtr, double *eta_av, const double *taux, const double *tauy,
                                        double dt, int calc_dtbt, const mom6x_rk2_hooks *hooks) {
  REQUIRE(c && c->rk2, MOM6X_EINVAL, "step_MOM_dyn_split_RK2: initialize_dyn_split_RK2 must be called first");
  REQUIRE(u_inst && v_inst && h && uh && vh && uhtr && vhtr && eta_av && taux && tauy, MOM6X_EINVAL,
          "step_MOM_dyn_split_RK2: null mandatory array");
  REQUIRE(c->a_u, MOM6X_EINVAL, "step_MOM_dyn_split_RK2: vertical viscosity coefficients have not been set");
  HIPCHK(hipSetDevice(c->device));
  RK2State *s = c->rk2;
  // On EVERY way out of the step (a failed callee returns early) the deferrals it switches on for its own btstep / btcalc calls
  // are switched off again: a btstep called from outside the step writes its accel_layer arrays (the pending result of the
  // step's last btstep stays), and a btcalc called from outside writes frhatu / frhatv.
  struct DeferGuard { mom6x_ctx *c; ~DeferGuard() { bt_defer_layer_accel(c, false); bt_defer_btcalc(c, false); } } defer_guard{c};
  const mom6x_rk2_params &R = s->P;
  const Dm d = c->d;
  const dim3 b = blk2();
  const int nk = d.nk;
  const PassWidths PW = pass_widths(c);
  // USE_BT_CONT_TYPE = False (R.no_BT_cont): CS%BT_cont is not associated -- BT_cont_BT_thick :467-469 is false, btcalc works from h :627,
  // the continuity calls and the two btstep calls run without a BT_cont_type (MOM_barotropic.F90:845: find_face_areas)
  const mom6x_BT_cont *BTp = R.no_BT_cont ? nullptr : &s->BT;
  double *u_av = s->u_av, *v_av = s->v_av, *h_av = s->h_av, *eta = s->eta;
  double *up = s->up, *vp = s->vp, *hp = s->hp, *u_bc = s->u_bc_accel, *v_bc = s->v_bc_accel;
  const double *taux_bot = R.split_bottom_stress ? s->taux_bot : nullptr;
  const double *tauy_bot = R.split_bottom_stress ? s->tauy_bot : nullptr;
  // vertvisc_coef at the three places of the step: the host callback if one is given, else the device routine if
  // vertvisc_init was called, else nothing (the coefficient arrays handed to mom6x_vertvisc_set_coef stay as they are)
  if (c->ds_Hmix > 0.0) c->ds_h = h;   // vertvisc(up, vp, h, ...) :754 and vertvisc(u, v, h, ...) :1013 both carry the step's h
  const bool dev_coef = c->vv_init && !(hooks && hooks->vertvisc_coef);
  auto coef_hook = [&](int stage, const double *uu, const double *vv, double dtt) -> int {
    if (hooks && hooks->vertvisc_coef) {
      HIPCHK(hipStreamSynchronize(c->stream));
      int rc = hooks->vertvisc_coef(hooks->user, stage, uu, vv, h, dtt);
      REQUIRE(rc == 0, MOM6X_EINVAL, "step_MOM_dyn_split_RK2: vertvisc_coef callback failed");
    }
    return MOM6X_OK;
  };

  // up = vp = 0 ; hp = h  :421-425.  Every own face of up, vp and every own cell of hp is overwritten before it is read
  // (:681-694 / the convergence of :781) and the group passes fill the connected halos, so all these assignments leave behind
  // is the halo beyond a closed edge: zero for up, vp -- they are private to this control structure, were zeroed when it was
  // allocated, and nobody writes there -- and h's halo for hp: only that frame is copied (3.7 GB per step less).
  KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 2 * d.halo, -d.halo), d.nj + 2 * d.halo, nk, b), b, d, hp, (const double *)h, (const double *)nullptr, 1, 0.0, d.halo, 2);

  // PFu = d/dx M(h,T,S) ; pbce = dM/deta  :503
  // u_bc_accel = CAu_pred + PFu + diffu ; up = mask*(u + dt*u_bc_accel)  :564-598.  up/vp at this point only feed
  // vertvisc_coef (:602-609) and are recomputed at :681-694, so they are formed only when that callback exists.
  const bool host_coef = (hooks && hooks->vertvisc_coef);   // up/vp are needed on the host before the solve
  // With the stored Coriolis acceleration (the rule: STORE_CORIOLIS_ACCEL) everything u_bc_accel needs but PFu exists already:
  // the pressure-force kernel forms it as it makes PFu (k_bc_accel's 8 words per face-layer -> 4 more in a kernel that runs anyway)
  const bool fold_bc = s->CAu_pred_stored && !host_coef;
  // (and the column sum of h that bt_mass_source :629 is about to form from the same array: one pass over h less)
  static const bool ms_own = [] { const char *e = getenv("MOM6X_BT_MASS_SOURCE"); return e && !strcmp(e, "own"); }();
  c->pgf_fold = BcFold{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ms_own ? nullptr : s->eta_h };
  if (fold_bc) c->pgf_fold = BcFold{ s->CAu_pred, s->CAv_pred, s->diffu, s->diffv, u_bc, v_bc, ms_own ? nullptr : s->eta_h };
  bool have_eta_h = false;
  {
    const int rc_pf = mom6x_PressureForce(c, h, s->PFu, s->PFv, s->pbce, s->eta_PF);
    c->pgf_fold = BcFold{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    have_eta_h = c->pgf_eta_h_written;
    c->pgf_eta_h_written = false;
    if (rc_pf) return rc_pf;
  }
  if (!s->CAu_pred_stored) CHK(mom6x_CorAdCalc(c, u_av, v_av, h_av, uh, vh, s->CAu_pred, s->CAv_pred));   // :552-557
  if (!fold_bc)
    KLAUNCH(c, "k_bc_accel", k_bc_accel, gridk(nxa(d.ni + 1, -1), d.nj + 1, nk, b), b, d, c->G, s->CAu_pred, s->CAv_pred, s->PFu, s->PFv,
            s->diffu, s->diffv, u_bc, v_bc, (const double *)u_inst, (const double *)v_inst, host_coef ? up : (double *)nullptr,
            host_coef ? vp : (double *)nullptr, dt);
  if (dev_coef && vertvisc_coef_solve_usable(c)) {   // :591-610 in one kernel per direction; nobody sees these coefficients (:737 replaces them)
    CHK(vertvisc_coef_remnant(c, u_inst, v_inst, u_bc, v_bc, dt, h, dt, s->visc_rem_u, s->visc_rem_v, dt, false));
  } else {
  if (dev_coef) CHK(vertvisc_coef_upd(c, 1, u_inst, v_inst, u_bc, v_bc, nullptr, nullptr, dt, h, dt, nullptr, nullptr));   // :591-609, up/vp on the fly
  else CHK(coef_hook(0, up, vp, dt));                                   // :602-609
  CHK(mom6x_vertvisc_remnant(c, s->visc_rem_u, s->visc_rem_v, dt));     // :610
  }
  // pass_eta :549/:617 + pass_visc_rem :618/:641 as ONE group started here; bt_mass_source (own cells only) and the own rows of
  // the continuity call below run while it travels; continuity completes it before it touches a halo row
  startn(c, { eta, s->visc_rem_u, s->visc_rem_v }, { 0, 1, 2 }, { 1, nk, nk }, { 0, PW.cont, PW.cont });   // :484-486

  if (!BTp) { bt_defer_btcalc(c, false); CHK(mom6x_btcalc_strict(c, h, nullptr, nullptr)); }   // :627-628 (BT_THICK_SCHEME = HYBRID, HARMONIC or ARITHMETIC)
  if (have_eta_h) CHK(bt_mass_source_from(c, s->eta_h, eta, 1));        // :629, with the sum k_pgf_main left
  else CHK(mom6x_bt_mass_source(c, h, eta, 1));
  // continuity(u, v, h, hp, uh_in, vh_in, dt, visc_rem_u, visc_rem_v, BT_cont)  :646
  // (hp of this call is never read: :781 overwrites it -- the last convergence is skipped; uh_in, vh_in and BT_cont are the results)
  c->cont_h_unused = true;
  {
    const int rc_c = mom6x_continuity_PPM(c, u_inst, v_inst, h, hp, s->uh_in, s->vh_in, dt, nullptr, nullptr, s->visc_rem_u, s->visc_rem_v,
                                          nullptr, nullptr, BTp, nullptr, nullptr);
    c->cont_h_unused = false;
    if (rc_c) return rc_c;
  }
  halo_complete(c);
  if (BTp) {
    bt_defer_btcalc(c, true);          // (the step's own btstep follows: its column pass forms the thickness fractions)
    CHK(mom6x_btcalc(c, h, s->BT.h_u, s->BT.h_v));                      // :649-652
  }
  if (calc_dtbt) CHK(set_dtbt_eta(c, s->pbce, BTp ? nullptr : eta));    // :659-668 (eta matters with NONLINEAR_BT_CONTINUITY only)
  // predictor btstep :673-676.  accel_layer_u / _v are only read by the velocity estimates below: with the device's own
  // vertvisc_coef they are evaluated there (LayerAccelSrc) and never written; mom6x_rk2_field materialises them on request.
  const bool defer_la = dev_coef && !host_coef;
  bt_defer_layer_accel(c, defer_la);
  s->accel_bt_deferred = defer_la;
  LayerAccelSrc LAu, LAv;
  CHK(mom6x_btstep(c, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, s->pbce, s->eta_PF, u_av, v_av, s->u_accel_bt,
                   s->v_accel_bt, s->eta_pred, s->uhbt, s->vhbt, s->visc_rem_u, s->visc_rem_v, BTp, taux_bot, tauy_bot,
                   s->uh_in, s->vh_in, u_inst, v_inst, nullptr));

  const double dt_pred = dt * R.be;                                     // :679
  if (!host_coef) {   // (with the callback, it needs up/vp before the solve: no fusion)
    // :737-738; the coefficient sweep forms the velocity estimate of :681-694 for its upwinding anyway and leaves it in up, vp
    const bool same_dt = (R.visc_rem_dt_bug != 0);
    const bool one_kernel = dev_coef && vertvisc_coef_solve_usable(c);   // coefficients + solve [+ remnant] per direction in ONE kernel
    if (dev_coef) {
      REQUIRE(bt_layer_accel_src(c, &LAu, &LAv), MOM6X_EINVAL, "step_MOM_dyn_split_RK2: no barotropic result to take the layer accelerations from");
      if (one_kernel)
        CHK(vertvisc_coef_solve_la(c, u_inst, v_inst, u_bc, v_bc, LAu, LAv, dt_pred, h, dt_pred, up, vp, taux, tauy, dt_pred, s->taux_bot,
                                   s->tauy_bot, same_dt ? s->visc_rem_u : nullptr, same_dt ? s->visc_rem_v : nullptr,
                                   !same_dt /* (its own vertvisc_remnant follows; else the corrector's coefficients replace these unseen) */));
      else
        CHK(vertvisc_coef_upd_la(c, u_inst, v_inst, u_bc, v_bc, LAu, LAv, dt_pred, h, dt_pred, up, vp));
    }
    // :681-694 + :754 + :763-767 in one column sweep per direction (k_vertvisc_fused / k_vertvisc_cols)
    if (one_kernel) { }
    else if (dev_coef)
      CHK(vertvisc_fused(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, up, vp, taux, tauy, dt_pred, s->taux_bot, s->tauy_bot,
                         same_dt ? s->visc_rem_u : nullptr, same_dt ? s->visc_rem_v : nullptr));
    else
    CHK(vertvisc_fused(c, u_inst, v_inst, u_bc, v_bc, s->u_accel_bt, s->v_accel_bt, dt_pred, up, vp, taux, tauy, dt_pred,
                       s->taux_bot, s->tauy_bot, same_dt ? s->visc_rem_u : nullptr, same_dt ? s->visc_rem_v : nullptr));
    if (!same_dt) CHK(mom6x_vertvisc_remnant(c, s->visc_rem_u, s->visc_rem_v, dt));
  } else {
    KLAUNCH(c, "k_vel_update", k_vel_update, gridk(nxa(d.ni + 1, -1), d.nj + 1, nk, b), b, d, c->G, (const double *)u_inst,
            (const double *)v_inst, u_bc, v_bc, s->u_accel_bt, s->v_accel_bt, up, vp, dt_pred);   // :681-694
    CHK(coef_hook(1, up, vp, dt_pred));                                   // :737-738
    if (R.visc_rem_dt_bug) {   // :754 + :763-767 share dt: one sweep
      CHK(vertvisc_fused(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, up, vp, taux, tauy, dt_pred, s->taux_bot,
                         s->tauy_bot, s->visc_rem_u, s->visc_rem_v));
    } else {
      CHK(mom6x_vertvisc(c, up, vp, taux, tauy, dt_pred, s->taux_bot, s->tauy_bot));               // :754
      CHK(mom6x_vertvisc_remnant(c, s->visc_rem_u, s->visc_rem_v, dt));                            // :763-767
    }
  }
  start3(c, { s->visc_rem_u, s->visc_rem_v, up, vp }, { 1, 2, 1, 2 }, nk, { PW.cont, PW.cont, PW.cont, PW.cont });   // pass_visc_rem :769 + pass_uvp :761/:773 (halo = max(1, cont_stencil) :485-487): completed inside continuity

  // uh = u_av * h ; hp = h + dt * div . uh  :779-781
  // (h_av = 0.5 * (h + hp) of :808-810 on the tile's own cells is written by the call's last convergence kernel, which has
  //  hp in a register and reads nothing twice)
  c->cont_av_kind = 1; c->cont_av = h_av; c->cont_av_src = h;
  {
    const int rc_c = mom6x_continuity_PPM(c, up, vp, h, hp, uh, vh, dt, s->uhbt, s->vhbt, s->visc_rem_u, s->visc_rem_v, u_av, v_av, BTp,
                                          nullptr, nullptr);
    c->cont_av_kind = 0;
    if (rc_c) return rc_c;
  }
  halo_complete(c);
  // hp (pass_hp_uv :785), the averaged velocities and the transports (pass_av_uvh :804 ... :865) travel on the halo stream
  // while the barotropic mass source is formed; the frame of h_av follows the completion
  start3(c, { hp, u_av, v_av, uh, vh }, { 0, 1, 2, 1, 2 }, nk, { PW.vel ? 2 : 0, PW.vel, PW.vel, PW.vel, PW.vel });   // :488-490

  // ---- corrector
  CHK(mom6x_bt_mass_source(c, hp, s->eta_pred, 0));                     // :820
  halo_complete(c);                                                     // :865
  KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 4, -2), d.nj + 4, nk, b), b, d, h_av, (const double *)h, (const double *)hp, 0, 0.0, 2, 2);
  if (R.begw != 0.0) {                                                  // :822-833
    KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 2, -1), d.nj + 2, nk, b), b, d, hp, (const double *)h, (const double *)nullptr, 3, R.begw, 1, 0);
    CHK(mom6x_PressureForce(c, hp, s->PFu, s->PFv, s->pbce, s->eta_PF));
  }
  if (BTp) {
    bt_defer_btcalc(c, true);
    CHK(mom6x_btcalc(c, h, s->BT.h_u, s->BT.h_v));                      // :864-867
  }
  if (hooks && hooks->horizontal_viscosity) {                           // :884-888
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = hooks->horizontal_viscosity(hooks->user, u_av, v_av, h_av, uh, vh, s->diffu, s->diffv);
    REQUIRE(rc == 0, MOM6X_EINVAL, "step_MOM_dyn_split_RK2: horizontal_viscosity callback failed");
  } else if (c->hv_init) {
    CHK(mom6x_horizontal_viscosity(c, u_av, v_av, h_av, s->diffu, s->diffv));
  }
  // :893 + :900-907: u_bc_accel = (CAu + PFu) + diffu is formed by the kernel that makes CAu
  CHK(CorAdCalc_bc(c, u_av, v_av, h_av, uh, vh, s->CAu, s->CAv, s->PFu, s->PFv, s->diffu, s->diffv, u_bc, v_bc, nullptr, nullptr, 0.0));
  // corrector btstep :939-942
  CHK(mom6x_btstep(c, u_inst, v_inst, eta, dt, u_bc, v_bc, taux, tauy, s->pbce, s->eta_PF, u_av, v_av, s->u_accel_bt,
                   s->v_accel_bt, s->eta_pred, s->uhbt, s->vhbt, s->visc_rem_u, s->visc_rem_v, BTp, taux_bot, tauy_bot, uh, vh,
                   u_av, v_av, eta_av));
  KLAUNCH(c, "k_eta", k_eta, grid3(d.ni, d.nj, 1, b), b, d, c->G, eta, (const double *)s->eta_pred, (const double *)nullptr, 0.0);   // :946
  // u = mask*(u + dt*(u_bc_accel + u_accel_bt))  :957-966
  if (!host_coef) {   // :957-966 + :1013 + :1022 in one column sweep per direction
    if (dev_coef) {   // :1002-1003; u = mask*(u + dt*(u_bc_accel + u_accel_bt)) is left in place by the coefficient sweep
      REQUIRE(bt_layer_accel_src(c, &LAu, &LAv), MOM6X_EINVAL, "step_MOM_dyn_split_RK2: no barotropic result to take the layer accelerations from");
      if (vertvisc_coef_solve_usable(c)) {
        CHK(vertvisc_coef_solve_la(c, u_inst, v_inst, u_bc, v_bc, LAu, LAv, dt, h, dt, u_inst, v_inst, taux, tauy, dt, s->taux_bot, s->tauy_bot,
                                   s->visc_rem_u, s->visc_rem_v, true));
      } else {
      CHK(vertvisc_coef_upd_la(c, u_inst, v_inst, u_bc, v_bc, LAu, LAv, dt, h, dt, u_inst, v_inst));
      CHK(vertvisc_fused(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, u_inst, v_inst, taux, tauy, dt, s->taux_bot,
                         s->tauy_bot, s->visc_rem_u, s->visc_rem_v));
      }
    } else
    CHK(vertvisc_fused(c, u_inst, v_inst, u_bc, v_bc, s->u_accel_bt, s->v_accel_bt, dt, u_inst, v_inst, taux, tauy, dt,
                       s->taux_bot, s->tauy_bot, s->visc_rem_u, s->visc_rem_v));
  } else {
    KLAUNCH(c, "k_vel_update", k_vel_update, gridk(nxa(d.ni + 1, -1), d.nj + 1, nk, b), b, d, c->G, (const double *)u_inst,
            (const double *)v_inst, u_bc, v_bc, s->u_accel_bt, s->v_accel_bt, u_inst, v_inst, dt);
    CHK(coef_hook(2, u_inst, v_inst, dt));                                // :1002-1003
    CHK(vertvisc_fused(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, u_inst, v_inst, taux, tauy, dt, s->taux_bot,
                       s->tauy_bot, s->visc_rem_u, s->visc_rem_v));       // :1013 + :1022
  }
  // h_av = h :1025-1027 and h_av = 0.5 * (h_av + h) :1064-1066: on the tile's own cells both ride on the convergence kernels of
  // the in-place continuity call between them (the first keeps the old thickness it is about to overwrite, the second averages);
  // the frame of halo cells is copied here and averaged after the group pass has brought the new thicknesses
  KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 4, -2), d.nj + 4, nk, b), b, d, h_av, (const double *)h, (const double *)nullptr, 1, 0.0, 2, 2);
  const int w_uv = PW.cont ? std::max(2, PW.cont) : 0;   // pass_uv, pass_h: halo = max(2, cont_stencil) :492-493
  start3(c, { s->visc_rem_u, s->visc_rem_v, u_inst, v_inst }, { 1, 2, 1, 2 }, nk, { PW.cont, PW.cont, w_uv, w_uv });   // pass_visc_rem :1030 + pass_uv :1019/:1034: completed inside continuity
  // uh = u_av * h ; h = h + dt * div . uh  :1041-1043
  c->cont_av_kind = 2; c->cont_av = h_av; c->cont_av_src = nullptr;
  {
    const int rc_c = mom6x_continuity_PPM(c, u_inst, v_inst, h, h, uh, vh, dt, s->uhbt, s->vhbt, s->visc_rem_u, s->visc_rem_v, u_av, v_av,
                                          nullptr, nullptr, nullptr);
    c->cont_av_kind = 0;
    if (rc_c) return rc_c;
  }
  halo_complete(c);
  start3(c, { h, u_av, v_av, uh, vh }, { 0, 1, 2, 1, 2 }, nk, { w_uv, PW.vel, PW.vel, PW.vel, PW.vel });   // pass_h :1045 + start_group_pass(CS%pass_av_uvh) :1054 (:493-495)
  halo_complete(c);                                                     // :1072
  KLAUNCH(c, "k_h_av", k_h_av, gridk(nxa(d.ni + 4, -2), d.nj + 4, nk, b), b, d, h_av, (const double *)h, (const double *)nullptr, 2, 0.0, 2, 2);
  // :1072-1079 uhtr += uh*dt: the ring of halo faces here, the box of own faces inside the kernel below, which reads uh, vh anyway
  KLAUNCH(c, "k_uhtr", k_uhtr, gridk(nxa(d.ni + 5, -3), d.nj + 5, nk, b), b, d, uhtr, vhtr, (const double *)uh, (const double *)vh, dt, 1);
  // CAu_pred for the next step :1081-1090
  CHK(CorAdCalc_bc(c, u_av, v_av, h_av, uh, vh, s->CAu_pred, s->CAv_pred, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, uhtr, vhtr, dt));
  s->CAu_pred_stored = true;
  HIPCHK(hipGetLastError());
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  return MOM6X_OK;
}
