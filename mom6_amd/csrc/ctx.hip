// ctx.hip -- context, layout and host<->device transfer entry points of the C ABI.
#include <cstdarg>
#include <vector>
#include "mom6x_dev.h"
#include <cstdlib>

#include <map>
static thread_local char g_err[512] = "";

struct ProfRec { int id; hipEvent_t a, b; };
struct Prof {
  std::vector<std::string> names;
  std::map<std::string, int> ids;
  std::vector<ProfRec> recs;
  std::vector<double> ms;
  std::vector<long> count;
  std::vector<hipEvent_t> pool;
  hipEvent_t cur_a;
  int cur_id;
  std::string filter;   // record only kernels whose name starts with this prefix ("" = all)
};

static hipEvent_t prof_event(Prof *p) {
  if (!p->pool.empty()) { hipEvent_t e = p->pool.back(); p->pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

extern "C" int mom6x_prof_filter(mom6x_ctx *c, const char *prefix) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_prof_filter: null ctx");
  if (!c->prof) c->prof = new Prof();
  c->prof->filter = prefix ? prefix : "";
  return MOM6X_OK;
}

void prof_begin(mom6x_ctx *c, const char *name) {
  Prof *p = c->prof;
  p->cur_id = -1;
  if (!p->filter.empty() && strncmp(name, p->filter.c_str(), p->filter.size()) != 0) return;
  auto it = p->ids.find(name);
  int id;
  if (it == p->ids.end()) {
    id = (int)p->names.size();
    p->names.push_back(name); p->ids[name] = id; p->ms.push_back(0.0); p->count.push_back(0);
  } else id = it->second;
  p->cur_id = id;
  p->cur_a = prof_event(p);
  (void)hipEventRecord(p->cur_a, c->stream);
}

void prof_end(mom6x_ctx *c) {
  Prof *p = c->prof;
  if (p->cur_id < 0) return;
  ProfRec r; r.id = p->cur_id; r.a = p->cur_a; r.b = prof_event(p);
  (void)hipEventRecord(r.b, c->stream);
  p->recs.push_back(r);
}

static void prof_collect(mom6x_ctx *c) {
  Prof *p = c->prof;
  (void)hipStreamSynchronize(c->stream);
  for (auto &r : p->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { p->ms[r.id] += ms; p->count[r.id] += 1; }
    p->pool.push_back(r.a); p->pool.push_back(r.b);
  }
  p->recs.clear();
}

extern "C" int mom6x_prof_enable(mom6x_ctx *c, int on) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_prof_enable: null ctx");
  HIPCHK(hipSetDevice(c->device));
  if (!c->prof) c->prof = new Prof();
  if (!on && c->prof_on) prof_collect(c);
  c->prof_on = (on != 0);
  return MOM6X_OK;
}

extern "C" int mom6x_prof_reset(mom6x_ctx *c) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_prof_reset: null ctx");
  if (!c->prof) return MOM6X_OK;
  prof_collect(c);
  for (auto &m : c->prof->ms) m = 0.0;
  for (auto &n : c->prof->count) n = 0;
  return MOM6X_OK;
}

// Writes "name\tcount\ttotal_ms\n" lines into buf; returns the number of bytes needed.
extern "C" int mom6x_prof_report(mom6x_ctx *c, char *buf, int buflen) {
  if (!c || !c->prof) { if (buf && buflen > 0) buf[0] = 0; return 0; }
  prof_collect(c);
  std::string out;
  char line[256];
  for (size_t i = 0; i < c->prof->names.size(); i++) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", c->prof->names[i].c_str(), c->prof->count[i], c->prof->ms[i]);
    out += line;
  }
  if (buf && buflen > 0) { strncpy(buf, out.c_str(), (size_t)buflen - 1); buf[buflen - 1] = 0; }
  return (int)out.size() + 1;
}

void mom6x_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *mom6x_last_error(void) { return g_err; }
extern "C" int mom6x_abi_version(void) { return MOM6X_ABI_VERSION; }
extern "C" int mom6x_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { mom6x_set_error("mom6x_device_count: hipGetDeviceCount failed"); return -1; }
  return n;
}

// sizeof() of the public structs, so that non-C hosts (ctypes, ISO_C_BINDING) can verify their mirrors.
extern "C" int mom6x_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(mom6x_dims);
    case 1: return (int)sizeof(mom6x_vgrid);
    case 2: return (int)sizeof(mom6x_continuity_params);
    case 3: return (int)sizeof(mom6x_BT_cont);
    case 4: return (int)sizeof(mom6x_barotropic_params);
    case 5: return (int)sizeof(mom6x_coriolis_params);
    case 6: return (int)sizeof(mom6x_pgf_params);
    case 7: return (int)sizeof(mom6x_rk2_params);
    case 8: return (int)sizeof(mom6x_rk2_hooks);
    case 9: return (int)sizeof(mom6x_eos_params);
    case 10: return (int)sizeof(mom6x_vertvisc_params);
    case 11: return (int)sizeof(mom6x_hor_visc_params);
    case 12: return (int)sizeof(mom6x_remapping_params);
    case 13: return (int)sizeof(mom6x_regrid_zstar_params);
    case 14: return (int)sizeof(mom6x_chksum_result);
    case 15: return (int)sizeof(mom6x_sum_output_params);
    case 16: return (int)sizeof(mom6x_energy_sums);
    case 17: return (int)sizeof(mom6x_regrid_rho_params);
    default: return -1;
  }
}

extern "C" int mom6x_dims_init(mom6x_dims *d, int ni, int nj, int nk, int halo) {
  REQUIRE(d && ni > 0 && nj > 0 && nk > 0, MOM6X_EINVAL, "mom6x_dims_init: bad sizes");
  REQUIRE(halo >= 1 && halo + 1 <= 16, MOM6X_EINVAL, "mom6x_dims_init: halo must be in 1..15");
  d->ni = ni; d->nj = nj; d->nk = nk; d->halo = halo;
  d->ioff = 16;               // local i = 0 starts a 128-byte line
  d->joff = halo + 1;
  d->pitch = ((d->ioff + ni + halo + 15) / 16) * 16;
  d->slab = d->pitch * (nj + 2 * halo + 1);
  d->i_glob0 = 0; d->j_glob0 = 0; d->ni_glob = ni; d->nj_glob = nj;
  d->reentrant_x = 0; d->reentrant_y = 0;
  return MOM6X_OK;
}

void bt_state_free(mom6x_ctx *ctx);   // barotropic.hip
void rk2_state_free(mom6x_ctx *ctx);  // dyn_split_RK2.hip
void ta_state_free(mom6x_ctx *ctx);   // tracer.hip

extern "C" int mom6x_ctx_create(mom6x_ctx **out, const mom6x_dims *dims, int device,
                                const double *metrics_host, const mom6x_vgrid *GV,
                                int first_direction) {
  REQUIRE(out && dims && metrics_host && GV, MOM6X_EINVAL, "mom6x_ctx_create: null argument");
  REQUIRE(dims->halo >= 3, MOM6X_EINVAL, "mom6x_ctx_create: halo >= 3 is required by continuity_PPM");
  REQUIRE(dims->ioff >= dims->halo + 1 && dims->joff >= dims->halo + 1 &&
          dims->pitch >= dims->ioff + dims->ni + dims->halo &&
          dims->slab == dims->pitch * (dims->nj + 2 * dims->halo + 1),
          MOM6X_EINVAL, "mom6x_ctx_create: inconsistent layout in dims");
  REQUIRE(GV->Boussinesq == 1, MOM6X_EUNSUPPORTED, "mom6x: only BOUSSINESQ=True is supported");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  REQUIRE(ndev > 0 && device >= 0 && device < ndev, MOM6X_EHIP,
          "mom6x_ctx_create: no such HIP device (the product path has no CPU fallback)");
  HIPCHK(hipSetDevice(device));
  mom6x_ctx *c = new mom6x_ctx();
  c->dims = *dims;
  c->d = make_dm(*dims);
  c->device = device;
  c->GV = *GV;
  c->first_direction = first_direction;
  c->cont_init = false; c->bt_init = false;
  c->prof_on = false; c->prof = nullptr; c->comm = nullptr; c->halo_error = false; c->ta = nullptr;
  c->cor_init = false; c->pgf_init = false; c->Rlay = c->g_prime = nullptr;
  c->a_u = c->a_v = c->h_u = c->h_v = c->Ray_u = c->Ray_v = nullptr;
  for (int m = 0; m < MOM6X_NSCR; m++) { c->scr[m] = nullptr; c->scr_nlev[m] = 0; }
  c->hL = c->hR = nullptr; c->bts = nullptr; c->rk2 = nullptr; c->flag = nullptr; c->G = nullptr;
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
  size_t nG = (size_t)MOM6X_G_COUNT * dims->slab;
  HIPCHK(hipMalloc(&c->G, nG * sizeof(double)));
  HIPCHK(hipMemcpy(c->G, metrics_host, nG * sizeof(double), hipMemcpyHostToDevice));
  size_t n3 = (size_t)dims->slab * dims->nk;
  HIPCHK(hipMalloc(&c->hL, n3 * sizeof(double)));
  HIPCHK(hipMalloc(&c->hR, n3 * sizeof(double)));
  HIPCHK(hipMemset(c->hL, work_fill_byte(), n3 * sizeof(double)));
  HIPCHK(hipMemset(c->hR, work_fill_byte(), n3 * sizeof(double)));
  HIPCHK(hipMalloc(&c->flag, sizeof(int)));
  HIPCHK(hipMemset(c->flag, 0, sizeof(int)));
  *out = c;
  return MOM6X_OK;
}

extern "C" int mom6x_ctx_destroy(mom6x_ctx *c) {
  if (!c) return MOM6X_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->halo_stream);
  bt_state_free(c);
  rk2_state_free(c);
  comm_free(c);
  ta_state_free(c);
  for (int m = 0; m < MOM6X_NSCR; m++) (void)hipFree(c->scr[m]);
  (void)hipFree(c->Rlay); (void)hipFree(c->g_prime); (void)hipFree(c->retry); (void)hipFree(c->cont_stats);
  hor_visc_free(c);
  diag_sums_free(c);
  (void)hipFree(c->regrid_res); (void)hipFree(c->regrid_vec); (void)hipFree(c->remap_src); (void)hipFree(c->remap_hvel);
  (void)hipFree(c->vv_a_u); (void)hipFree(c->vv_a_v); (void)hipFree(c->vv_h_u); (void)hipFree(c->vv_h_v);
  (void)hipFree(c->G); (void)hipFree(c->hL); (void)hipFree(c->hR); (void)hipFree(c->flag);
  if (c->ev_ready) { (void)hipEventDestroy(c->ev_ready); (void)hipEventDestroy(c->ev_done); }
  (void)hipStreamDestroy(c->stream); (void)hipStreamDestroy(c->halo_stream);
  delete c;
  return MOM6X_OK;
}

// The reference's test.nan (.testing/Makefile:399-408: every allocation initialised with signalling NaNs): with MOM6X_POISON_WORK=1 in
// the environment the WORK arrays of the device -- the scratch slots, the PPM edge arrays, the tracer advection's remainders, whatever a
// host asks mom6x_dev_alloc for -- start as NaNs (all bits set) instead of zeros.  A step that reads a word of them it has not
// written differs from the un-poisoned run or trips the numeric flag (tests/test_invariants_gpu.py).  Arrays the reference zeroes
// explicitly after allocating them (the restart fields of the RK2 and barotropic modules) are not work arrays.
int work_fill_byte() {
  const char *e = getenv("MOM6X_POISON_WORK");   // (read at every allocation: a test switches it between two models of one process)
  return (e && atoi(e) != 0) ? 0xFF : 0;
}

int ctx_scratch(mom6x_ctx *c, int slot, int nlev, double **out) {
  REQUIRE(slot >= 0 && slot < MOM6X_NSCR, MOM6X_EINVAL, "ctx_scratch: bad slot");
  if (c->scr[slot] && c->scr_nlev[slot] < nlev) { HIPCHK(hipFree(c->scr[slot])); c->scr[slot] = nullptr; }
  if (!c->scr[slot]) {
    const size_t n = (size_t)c->dims.slab * nlev;
    HIPCHK(hipMalloc(&c->scr[slot], n * sizeof(double)));
    HIPCHK(hipMemsetAsync(c->scr[slot], work_fill_byte(), n * sizeof(double), c->stream));
    c->scr_nlev[slot] = nlev;
  }
  *out = c->scr[slot];
  return MOM6X_OK;
}

extern "C" void *mom6x_ctx_stream(mom6x_ctx *c) { return (void *)c->stream; }
extern "C" const mom6x_dims *mom6x_ctx_dims(const mom6x_ctx *c) { return &c->dims; }
extern "C" const double *mom6x_ctx_metrics_dev(const mom6x_ctx *c) { return c->G; }

extern "C" int mom6x_ctx_sync(mom6x_ctx *c) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->halo_stream));
  int flag = 0;
  HIPCHK(hipMemcpy(&flag, c->flag, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIPCHK(hipMemset(c->flag, 0, sizeof(int)));
    mom6x_set_error("device-side numeric error flag = %d (NaN or negative thickness)", flag);
    return MOM6X_ENUMERIC;
  }
  return MOM6X_OK;
}

extern "C" int mom6x_dev_alloc(mom6x_ctx *c, double **p, size_t n) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(p, n * sizeof(double)));
  HIPCHK(hipMemsetAsync(*p, work_fill_byte(), n * sizeof(double), c->stream));
  return MOM6X_OK;
}
// device -> device on the context's stream (the Fortran shims: a result that lands in a scratch slot although its host array is resident)
extern "C" int mom6x_dev_copy(mom6x_ctx *c, double *dst, const double *src, size_t n) {
  REQUIRE(c && dst && src, MOM6X_EINVAL, "mom6x_dev_copy: null argument");
  HIPCHK(hipSetDevice(c->device));
  if (dst != src) HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return MOM6X_OK;
}
extern "C" int mom6x_dev_free(mom6x_ctx *c, double *p) {
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipFree(p));
  return MOM6X_OK;
}

// Fortran symmetric-memory extents of the four staggerings (MOM_memory_macros.h, dynamic
// symmetric branch): h (isd:ied, jsd:jed), u (isd-1:ied, jsd:jed), v (isd:ied, jsd-1:jed),
// q (isd-1:ied, jsd-1:jed).
static void f_extent(const mom6x_dims &d, int stagger, int &nx, int &ny, int &i0, int &j0) {
  const int xB = (stagger == 1 || stagger == 3), yB = (stagger == 2 || stagger == 3);
  nx = d.ni + 2 * d.halo + xB; ny = d.nj + 2 * d.halo + yB;
  i0 = -d.halo - xB; j0 = -d.halo - yB;
}

static int xfer(mom6x_ctx *c, double *dev, double *host, int stagger, int nk, bool up) {
  REQUIRE(stagger >= 0 && stagger <= 3 && nk >= 1, MOM6X_EINVAL, "mom6x_upload/download: bad stagger or nk");
  HIPCHK(hipSetDevice(c->device));
  const mom6x_dims &d = c->dims;
  int nx, ny, i0, j0;
  f_extent(d, stagger, nx, ny, i0, j0);
  const int nrows = d.nj + 2 * d.halo + 1;
  hipMemcpy3DParms p;
  memset(&p, 0, sizeof(p));
  hipPitchedPtr hp = make_hipPitchedPtr(host, (size_t)nx * sizeof(double), nx, ny);
  hipPitchedPtr dp = make_hipPitchedPtr(dev, (size_t)d.pitch * sizeof(double), d.pitch, nrows);
  hipPos hpos = make_hipPos(0, 0, 0);
  hipPos dpos = make_hipPos((size_t)(i0 + d.ioff) * sizeof(double), (size_t)(j0 + d.joff), 0);
  if (up) { p.srcPtr = hp; p.srcPos = hpos; p.dstPtr = dp; p.dstPos = dpos; p.kind = hipMemcpyHostToDevice; }
  else    { p.srcPtr = dp; p.srcPos = dpos; p.dstPtr = hp; p.dstPos = hpos; p.kind = hipMemcpyDeviceToHost; }
  p.extent = make_hipExtent((size_t)nx * sizeof(double), ny, nk);
  HIPCHK(hipMemcpy3DAsync(&p, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MOM6X_OK;
}

extern "C" int mom6x_upload(mom6x_ctx *c, double *dev, const double *host_f, int stagger, int nk) {
  return xfer(c, dev, const_cast<double *>(host_f), stagger, nk, true);
}
extern "C" int mom6x_download(mom6x_ctx *c, double *host_f, const double *dev, int stagger, int nk) {
  return xfer(c, const_cast<double *>(dev), host_f, stagger, nk, false);
}
