// halo.hip -- halo updates: pass_var / pass_vector / create_group_pass + do_group_pass
// (config_src/infra/FMS2/MOM_domain_infra.F90:171-560, :1141-1190) for MOM6's 2-D tile decomposition,
// one tile per MI355X.
//
// A group pass (any number of 2-D/3-D fields of any staggering) is ONE packed message per neighbour:
//   pack kernel(s) -> ncclGroupStart; ncclSend/ncclRecv to the <= 8 neighbours (edges + corners: To_All
//   without Omit_Corners); ncclGroupEnd -> unpack kernel(s), all stream-ordered on the context's stream.
// xGMI is point-to-point and these messages are latency-dominated, so aggregating a
// whole group pass into one message per peer is the lever, exactly as mpp_create_group_update does.
// Re-entrant directions whose neighbour is the tile itself are device-to-device copies; closed
// boundaries have no neighbour (halo untouched, as mpp_update_domains on land-bounded edges).
//
// Symmetric memory: the u-point computational domain is I = -1..ni-1 (both shared edges owned), so only
// I <= -2 and I >= ni are halo points; likewise for v/q points in j.
//
// RCCL is resolved at run time (dlsym) from the RCCL instance already present in the process (torch's
// bundled librccl under Python; the host's own under Fortran) so that only one RCCL exists per process.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <vector>
#include "mom6x_dev.h"

#define MAXF 24
struct WrapArgs { double *f[MAXF]; int stg[MAXF]; int nk[MAXF]; int n; int rx; int w, w2; int wf[MAXF]; };
// w: rows / columns of halo this pass fills for its 3-D fields (create_group_pass's halo=), <= d.halo; w2: for its 2-D fields
// (always the context's: the barotropic solver reads eta over its wide halo); wf[m] > 0: the width of 3-D field m alone
#define PASS_W(A, m) (((A).nk[m] == 1) ? (A).w2 : ((A).wf[m] > 0 ? (A).wf[m] : (A).w))

// ---- region logic (host + device) ------------------------------------------------------------------
// Directions d = 0..7: W, E, S, N, SW, SE, NW, NE.
__host__ __device__ inline void dir_dxdy(int d, int &dx, int &dy) {
  const int DX[8] = { -1, 1, 0, 0, -1, 1, -1, 1 }, DY[8] = { 0, 0, -1, 1, -1, -1, 1, 1 };
  dx = DX[d]; dy = DY[d];
}
__host__ __device__ inline int dir_opp(int d) { const int O[8] = { 1, 0, 3, 2, 7, 6, 5, 4 }; return O[d]; }

// Inclusive index range along one axis.  n = ni|nj, w = halo, B = 1 for a B-staggered (face/vertex) axis.
// send: the part of MY computational domain the neighbour in direction s (-1,0,+1) needs;
// recv: the part of MY halo that the neighbour in direction s fills.
__host__ __device__ inline void axis_range(int n, int w, int B, int s, int send, int &a0, int &a1) {
  if (s == 0) { a0 = -B; a1 = n - 1; }
  else if (send) { if (s > 0) { a0 = n - w - B; a1 = n - 1 - B; } else { a0 = 0; a1 = w - 1; } }
  else { if (s > 0) { a0 = n; a1 = n + w - 1; } else { a0 = -w - B; a1 = -1 - B; } }
}

extern "C" int mom6x_halo_region(const mom6x_dims *d, int stagger, int dir, int send, int *i0, int *i1, int *j0, int *j1) {
  if (!d || dir < 0 || dir > 7 || stagger < 0 || stagger > 3) return MOM6X_EINVAL;
  int dx, dy; dir_dxdy(dir, dx, dy);
  const int xB = (stagger == 1 || stagger == 3), yB = (stagger == 2 || stagger == 3);
  axis_range(d->ni, d->halo, xB, dx, send, *i0, *i1);
  axis_range(d->nj, d->halo, yB, dy, send, *j0, *j1);
  return MOM6X_OK;
}

// Rank of the neighbour of tile (px,py) in direction dir, or -1 (closed boundary).  Rank = px + npx*py.
extern "C" int mom6x_halo_neighbor(int npx, int npy, int px, int py, int dir, int reentrant_x, int reentrant_y) {
  int dx, dy; dir_dxdy(dir, dx, dy);
  int qx = px + dx, qy = py + dy;
  if (qx < 0 || qx >= npx) { if (!reentrant_x) return -1; qx = (qx + npx) % npx; }
  if (qy < 0 || qy >= npy) { if (!reentrant_y) return -1; qy = (qy + npy) % npy; }
  return qx + npx * qy;
}

// ---- pack / unpack ----------------------------------------------------------------------------------
// ONE launch moves the regions of all (up to 8) directions of every field of the group between the fields
// and the per-direction contiguous buffers (field-major, then k, j, i); blockIdx.y is the direction.
struct Bufs8 { double *p[8]; };

__global__ void __launch_bounds__(256)
k_halo_pack(Dm d, WrapArgs A, Bufs8 B, int send /*1: pack send regions, 0: unpack recv regions*/) {
  const int dir = blockIdx.y;
  double *__restrict__ buf = B.p[dir];
  if (!buf) return;   // no neighbour in this direction
  int dx, dy; dir_dxdy(dir, dx, dy);
  size_t off = 0;
  for (int m = 0; m < A.n; m++) {
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    int i0, i1, j0, j1;
    axis_range(d.ni, PASS_W(A, m), xB, dx, send, i0, i1);
    axis_range(d.nj, PASS_W(A, m), yB, dy, send, j0, j1);
    const unsigned nx = (unsigned)(i1 - i0 + 1), ny = (unsigned)(j1 - j0 + 1);
    const unsigned per_k = nx * ny, tot = per_k * (unsigned)A.nk[m];   // (a region of a tile: far below 2^32 -- 32-bit index arithmetic)
    double *__restrict__ fld = A.f[m];
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += gridDim.x * blockDim.x) {
      const unsigned k = t / per_k;
      const unsigned r = t - k * per_k;
      const unsigned jj = r / nx, ii = r - jj * nx;
      const size_t x = ix3(d, i0 + (int)ii, j0 + (int)jj, (int)k);
      if (send) buf[off + t] = fld[x];
      else fld[x] = buf[off + t];
    }
    off += tot;
  }
}

static size_t region_count(const mom6x_dims &d, const WrapArgs &A, int dir) {
  int dx, dy; dir_dxdy(dir, dx, dy);
  size_t tot = 0;
  for (int m = 0; m < A.n; m++) {
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    int i0, i1, j0, j1;
    axis_range(d.ni, PASS_W(A, m), xB, dx, 1, i0, i1);
    axis_range(d.nj, PASS_W(A, m), yB, dy, 1, j0, j1);
    tot += (size_t)(i1 - i0 + 1) * (j1 - j0 + 1) * A.nk[m];
  }
  return tot;
}

// ---- single-tile wrap (no communicator attached) ---------------------------------------------------
__global__ void k_wrap_x(Dm d, WrapArgs A) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  const int hh = threadIdx.y;
  const int k = blockIdx.z;
  const int ni = d.ni;
  for (int m = 0; m < A.n; m++) {
    if (k >= A.nk[m]) continue;
    const int w = PASS_W(A, m);
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    const int j = -yB + jj;                       // computational rows only: corners come from the y pass
    if (j > d.nj - 1) continue;
    double *p = A.f[m] + (size_t)k * d.slab;
    if (hh < w) {
      const int iw = -w - xB + hh;
      p[ix2(d, iw, j)] = p[ix2(d, iw + ni, j)];
      const int ie = ni + hh;
      p[ix2(d, ie, j)] = p[ix2(d, ie - ni, j)];
    }
  }
}

__global__ void k_wrap_y(Dm d, WrapArgs A) {
  const int ii = blockIdx.x * blockDim.x + threadIdx.x;
  const int hh = threadIdx.y;
  const int k = blockIdx.z;
  const int nj = d.nj;
  for (int m = 0; m < A.n; m++) {
    if (k >= A.nk[m]) continue;
    const int w = PASS_W(A, m);
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    // with a re-entrant x the (already wrapped) x-halo columns are included, which fills the corners
    const int i = (A.rx ? -w - xB : -xB) + ii;
    if (i > d.ni - 1 + (A.rx ? w : 0)) continue;
    double *p = A.f[m] + (size_t)k * d.slab;
    if (hh < w) {
      const int js = -w - yB + hh;
      p[ix2(d, i, js)] = p[ix2(d, i, js + nj)];
      const int jn = nj + hh;
      p[ix2(d, i, jn)] = p[ix2(d, i, jn - nj)];
    }
  }
}

// MOM6X_POISON_HALO=1 (tests/test_layout_gpu.py): after a group pass narrower than the halo, every halo point of the passed fields
// that has a neighbour (or wraps around) but lies BEYOND the passed width becomes a NaN.  The step then only gives the bits of a
// full-width run if no kernel's result depends on the rows the narrow passes leave stale (dyn_split_RK2.hip: pass_widths).
struct Nbr8 { int on[8]; };
__global__ void __launch_bounds__(256)
k_halo_poison(Dm d, WrapArgs A, Nbr8 N) {
  const int nrows = d.slab / d.pitch;
  const unsigned per_k = (unsigned)d.pitch * (unsigned)nrows;
  const double bad = __longlong_as_double(0x7ff8000000000000LL);
  for (int m = 0; m < A.n; m++) {
    const int w = PASS_W(A, m);
    if (w >= d.halo) continue;
    const int xB = (A.stg[m] == 1 || A.stg[m] == 3), yB = (A.stg[m] == 2 || A.stg[m] == 3);
    const unsigned tot = per_k * (unsigned)A.nk[m];
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += gridDim.x * blockDim.x) {
      const unsigned k = t / per_k, r = t - k * per_k;
      const int j = (int)(r / (unsigned)d.pitch) - d.joff, i = (int)(r % (unsigned)d.pitch) - d.ioff;
      if (i < -d.halo - xB || i > d.ni - 1 + d.halo || j < -d.halo - yB || j > d.nj - 1 + d.halo) continue;   // (padding of the pitched rows)
      const int sx = (i < -xB) ? -1 : ((i > d.ni - 1) ? 1 : 0), sy = (j < -yB) ? -1 : ((j > d.nj - 1) ? 1 : 0);
      if (sx == 0 && sy == 0) continue;
      int dir = -1;
      for (int q = 0; q < 8; q++) { int dx, dy; dir_dxdy(q, dx, dy); if (dx == sx && dy == sy) dir = q; }
      if (!N.on[dir]) continue;
      const int dist_x = (sx < 0) ? (-xB - i) : ((sx > 0) ? i - (d.ni - 1) : 0), dist_y = (sy < 0) ? (-yB - j) : ((sy > 0) ? j - (d.nj - 1) : 0);
      if (dist_x > w || dist_y > w) A.f[m][(size_t)k * d.slab + r] = bad;
    }
  }
}
static bool poison_halo() { static const bool on = [] { const char *e = getenv("MOM6X_POISON_HALO"); return e && *e && *e != '0'; }(); return on; }

// ---- the communicator -------------------------------------------------------------------------------
struct NcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
  bool ok;
};

static NcclApi g_rccl = {};      // the RCCL of the process (dlsym)
static NcclApi g_user = {};      // a transport the host supplied (mom6x_comm_set_transport)

// ---- a transport of the host's own (include/mom6x.h: mom6x_transport): GPU-aware MPI under a host that keeps its halo
// traffic in one library, or the in-process transport of the layout tests (tests/transport/).  The host's nine functions
// are called through adapters with RCCL's signatures, so the exchange code below has one shape.
static mom6x_transport g_user_fn = {};
static bool g_user_on = false;
namespace uadapt {
static int dtype_of(ncclDataType_t t) { return (t == ncclInt) ? MOM6X_T_INT32 : ((t == ncclInt64) ? MOM6X_T_INT64 : MOM6X_T_FLOAT64); }
static int op_of(ncclRedOp_t o) { return (o == ncclSum) ? MOM6X_OP_SUM : ((o == ncclMin) ? MOM6X_OP_MIN : MOM6X_OP_MAX); }
static ncclResult_t res(int rc) { return rc == 0 ? ncclSuccess : ncclSystemError; }
static ncclResult_t GetUniqueId(ncclUniqueId *id) { memset(id, 0, sizeof(*id)); return res(g_user_fn.get_unique_id(id->internal)); }
static ncclResult_t CommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  void *h = nullptr;
  const int rc = g_user_fn.comm_init_rank(&h, nranks, id.internal, rank);
  *comm = (ncclComm_t)h;
  return res(rc);
}
static ncclResult_t CommDestroy(ncclComm_t comm) { return res(g_user_fn.comm_destroy((void *)comm)); }
static ncclResult_t Send(const void *buf, size_t n, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  return res(g_user_fn.send(buf, n, dtype_of(t), peer, (void *)comm, (void *)st));
}
static ncclResult_t Recv(void *buf, size_t n, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  return res(g_user_fn.recv(buf, n, dtype_of(t), peer, (void *)comm, (void *)st));
}
static ncclResult_t GroupStart() { return res(g_user_fn.group_start()); }
static ncclResult_t GroupEnd() { return res(g_user_fn.group_end()); }
static ncclResult_t AllReduce(const void *send, void *recv, size_t n, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
  return res(g_user_fn.all_reduce(send, recv, n, dtype_of(t), op_of(op), (void *)comm, (void *)st));
}
static const char *GetErrorString(ncclResult_t) { return g_user_fn.error_string ? g_user_fn.error_string(1) : "the host's transport failed"; }
}  // namespace uadapt

static std::mutex g_transport_mu;
extern "C" int mom6x_comm_set_transport(const mom6x_transport *t) {
  std::lock_guard<std::mutex> lk(g_transport_mu);
  if (!t) { g_user_on = false; return MOM6X_OK; }
  REQUIRE(t->get_unique_id && t->comm_init_rank && t->comm_destroy && t->send && t->recv && t->group_start && t->group_end && t->all_reduce,
          MOM6X_EINVAL, "mom6x_comm_set_transport: every function but error_string is required");
  g_user_fn = *t;
  g_user.GetUniqueId = uadapt::GetUniqueId; g_user.CommInitRank = uadapt::CommInitRank; g_user.CommDestroy = uadapt::CommDestroy;
  g_user.Send = uadapt::Send; g_user.Recv = uadapt::Recv; g_user.GroupStart = uadapt::GroupStart; g_user.GroupEnd = uadapt::GroupEnd;
  g_user.AllReduce = uadapt::AllReduce; g_user.GetErrorString = uadapt::GetErrorString;
  g_user.ok = true;
  g_user_on = true;
  return MOM6X_OK;
}

// The transport a NEW communicator (or unique id) gets -- the host's own if one is set, else the process's RCCL; the
// communicator keeps the one it was made with.
static int nccl_load(NcclApi **out) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  {
    std::lock_guard<std::mutex> lk2(g_transport_mu);
    if (g_user_on) { *out = &g_user; return MOM6X_OK; }
  }
  NcclApi &g_nccl = g_rccl;
  *out = &g_rccl;
  if (g_nccl.ok) return MOM6X_OK;
  void *h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclSend")) {   // no RCCL in the process yet: load the system one
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    REQUIRE(h, MOM6X_EHIP, "mom6x: cannot find RCCL (librccl.so) in the process or on the system");
  }
#define LOADSYM(field, name) *(void **)(&g_nccl.field) = dlsym(h, name); REQUIRE(g_nccl.field, MOM6X_EHIP, "mom6x: RCCL symbol " name " not found")
  LOADSYM(GetUniqueId, "ncclGetUniqueId"); LOADSYM(CommInitRank, "ncclCommInitRank"); LOADSYM(CommDestroy, "ncclCommDestroy");
  LOADSYM(Send, "ncclSend"); LOADSYM(Recv, "ncclRecv"); LOADSYM(GroupStart, "ncclGroupStart"); LOADSYM(GroupEnd, "ncclGroupEnd");
  LOADSYM(AllReduce, "ncclAllReduce"); LOADSYM(GetErrorString, "ncclGetErrorString");
#undef LOADSYM
  g_nccl.ok = true;
  return MOM6X_OK;
}

#define NCCLCHK(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) { mom6x_set_error("%s failed: %s", #expr, api->GetErrorString(r_)); return MOM6X_EHIP; } \
  } while (0)

struct Comm {
  int npx, npy, px, py, nranks, rank;
  int nbr[8];
  NcclApi *api;              // the transport this communicator was made with
  ncclComm_t comm;           // null when every neighbour is the tile itself (single rank)
  bool force_nccl_self;      // test mode: route self-neighbour messages through ncclSend/ncclRecv too
  double *sbuf[8], *rbuf[8];
  size_t cap[8];
  double *red;               // 1-element device buffer for scalar all-reduces
};

extern "C" int mom6x_comm_unique_id(char *id128) {
  REQUIRE(id128, MOM6X_EINVAL, "mom6x_comm_unique_id: null buffer");
  NcclApi *api;
  int rc = nccl_load(&api); if (rc) return rc;
  ncclUniqueId id;
  NCCLCHK(api->GetUniqueId(&id));
  memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
  return MOM6X_OK;
}

void comm_free(mom6x_ctx *c) {
  Comm *m = (Comm *)c->comm;
  if (!m) return;
  for (int d = 0; d < 8; d++) { (void)hipFree(m->sbuf[d]); (void)hipFree(m->rbuf[d]); }
  (void)hipFree(m->red);
  if (m->comm) (void)m->api->CommDestroy(m->comm);
  delete m;
  c->comm = nullptr;
}

// Attach the 2-D tile layout (MOM_domains LAYOUT = npx,npy; this tile = (px,py), rank = px + npx*py) and
// an RCCL communicator created from `id128` (from mom6x_comm_unique_id on rank 0, broadcast by the host
// with MPI_Bcast / torch.distributed).  id128 may be NULL when nranks == 1.
extern "C" int mom6x_comm_init(mom6x_ctx *c, int npx, int npy, int px, int py, const char *id128, int force_nccl_self) {
  REQUIRE(c && npx >= 1 && npy >= 1 && px >= 0 && px < npx && py >= 0 && py < npy, MOM6X_EINVAL, "mom6x_comm_init: bad layout");
  // a tile narrower than its halo (+ the B-point offset) would send from its own halo, i.e. stale data; FMS aborts on such a layout
  REQUIRE(c->dims.ni >= c->dims.halo + 1 && c->dims.nj >= c->dims.halo + 1, MOM6X_EINVAL,
          "mom6x_comm_init: the tile of this layout is narrower than the halo (ni, nj must be at least halo + 1)");
  HIPCHK(hipSetDevice(c->device));
  comm_free(c);
  Comm *m = new Comm();
  memset(m, 0, sizeof(*m));
  m->npx = npx; m->npy = npy; m->px = px; m->py = py; m->nranks = npx * npy; m->rank = px + npx * py;
  m->force_nccl_self = (force_nccl_self != 0);
  bool need_nccl = m->force_nccl_self;
  for (int d = 0; d < 8; d++) {
    m->nbr[d] = mom6x_halo_neighbor(npx, npy, px, py, d, c->dims.reentrant_x, c->dims.reentrant_y);
    if (m->nbr[d] >= 0 && m->nbr[d] != m->rank) need_nccl = true;
  }
  if (m->nranks > 1) need_nccl = true;
  if (need_nccl) {
    REQUIRE(id128, MOM6X_EINVAL, "mom6x_comm_init: a unique id is required for a multi-rank layout");
    NcclApi *api;
    int rc = nccl_load(&api); if (rc) { delete m; return rc; }
    m->api = api;
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    NCCLCHK(api->CommInitRank(&m->comm, m->nranks, id, m->rank));
  }
  HIPCHK(hipMalloc(&m->red, 2 * sizeof(double)));
  c->comm = m;
  return MOM6X_OK;
}

extern "C" int mom6x_comm_rank(const mom6x_ctx *c) { return (c && c->comm) ? ((Comm *)c->comm)->rank : 0; }

// min_across_PEs / max_across_PEs / sum_across_PEs of one scalar (MOM_coms.F90): op 0 min, 1 max, 2 sum.
int comm_allreduce_scalar(mom6x_ctx *c, double *value, int op) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || m->nranks == 1) return MOM6X_OK;
  NcclApi *api = m->api;
  HIPCHK(hipMemcpyAsync(m->red, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
  const ncclRedOp_t rop = (op == 0) ? ncclMin : ((op == 1) ? ncclMax : ncclSum);
  NCCLCHK(api->AllReduce(m->red, m->red + 1, 1, ncclDouble, rop, m->comm, c->stream));
  HIPCHK(hipMemcpyAsync(value, m->red + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MOM6X_OK;
}

// sum_across_PEs / min_across_PEs / max_across_PEs of 64-bit integers, in place on the device and in stream order
// (diag_sums.hip: the limbs of the reproducing sums, bit counts, order keys): op 0 min, 1 max, 2 sum.
int comm_allreduce_i64(mom6x_ctx *c, long long *dev, size_t n, int op) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || m->nranks == 1 || n == 0) return MOM6X_OK;
  NcclApi *api = m->api;
  const ncclRedOp_t rop = (op == 0) ? ncclMin : ((op == 1) ? ncclMax : ncclSum);
  NCCLCHK(api->AllReduce(dev, dev, n, ncclInt64, rop, m->comm, c->stream));
  return MOM6X_OK;
}
int comm_allreduce_f64(mom6x_ctx *c, double *dev, size_t n, int op) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || m->nranks == 1 || n == 0) return MOM6X_OK;
  NcclApi *api = m->api;
  const ncclRedOp_t rop = (op == 0) ? ncclMin : ((op == 1) ? ncclMax : ncclSum);
  NCCLCHK(api->AllReduce(dev, dev, n, ncclDouble, rop, m->comm, c->stream));
  return MOM6X_OK;
}
int comm_nranks(const mom6x_ctx *c) { const Comm *m = (const Comm *)c->comm; return (m && m->comm) ? m->nranks : 1; }

// pack_first: the pack kernel runs on the COMPUTE stream, ahead of whatever the caller launches next there, and the rest (messages,
// unpack) on `st` once it is done -- for a pass whose send regions the overlapped kernels are about to rewrite (halo_start_packed)
static int exchange(mom6x_ctx *c, Comm *m, const WrapArgs &A, hipStream_t st, bool pack_first = false) {
  c->n_exchanges++;   // (mom6x_comm_exchange_count)
  const Dm d = c->d;
  NcclApi *api = m->api;
  size_t cnt[8];
  for (int dir = 0; dir < 8; dir++) {
    cnt[dir] = 0;
    if (m->nbr[dir] < 0) continue;
    cnt[dir] = region_count(c->dims, A, dir);
    if (cnt[dir] > m->cap[dir]) {
      HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipStreamSynchronize(c->halo_stream));
      (void)hipFree(m->sbuf[dir]); (void)hipFree(m->rbuf[dir]);
      m->cap[dir] = cnt[dir] + cnt[dir] / 4;
      HIPCHK(hipMalloc(&m->sbuf[dir], m->cap[dir] * sizeof(double)));
      HIPCHK(hipMalloc(&m->rbuf[dir], m->cap[dir] * sizeof(double)));
    }
  }
  size_t cmax = 0;
  Bufs8 SB, RB;
  for (int dir = 0; dir < 8; dir++) {
    SB.p[dir] = (m->nbr[dir] >= 0) ? m->sbuf[dir] : nullptr;
    RB.p[dir] = (m->nbr[dir] >= 0) ? m->rbuf[dir] : nullptr;
    if (cnt[dir] > cmax) cmax = cnt[dir];
  }
  if (cmax == 0) return MOM6X_OK;
  for (int dir = 0; dir < 8; dir++) c->n_exchange_bytes += (long long)cnt[dir] * 8;   // (mom6x_comm_exchange_bytes: what this tile sends)
  const int blocks = (int)((cmax + 255) / 256 > 512 ? 512 : (cmax + 255) / 256);
  const bool own = (st == c->stream);     // (the per-kernel timing of mom6x_prof_* follows the compute stream only)
  if (own || pack_first) KLAUNCH(c, "k_halo_pack", k_halo_pack, dim3(blocks, 8), dim3(256), d, A, SB, 1);
  else hipLaunchKernelGGL(k_halo_pack, dim3(blocks, 8), dim3(256), 0, st, d, A, SB, 1);
  if (pack_first && !own) {   // the messages leave when the pack is done
    HIPCHK(hipEventRecord(c->ev_ready, c->stream));
    HIPCHK(hipStreamWaitEvent(st, c->ev_ready, 0));
  }
  // sends in direction order; receives in the order of the OPPOSITE directions, so that the j-th send to a
  // peer pairs with the peer's j-th receive from us even when one rank is the neighbour in several directions.
  bool in_group = false;
  for (int dir = 0; dir < 8; dir++) {
    if (m->nbr[dir] < 0) continue;
    const bool self = (m->nbr[dir] == m->rank);
    if (self && !m->force_nccl_self) {
      // my message toward `dir` arrives as my own receive from direction opp(dir)
      HIPCHK(hipMemcpyAsync(m->rbuf[dir_opp(dir)], m->sbuf[dir], cnt[dir] * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
  }
  if (m->comm) {
    NCCLCHK(api->GroupStart());
    in_group = true;
    for (int dir = 0; dir < 8; dir++) {
      if (m->nbr[dir] < 0) continue;
      if (m->nbr[dir] == m->rank && !m->force_nccl_self) continue;
      NCCLCHK(api->Send(m->sbuf[dir], cnt[dir], ncclDouble, m->nbr[dir], m->comm, st));
    }
    for (int dir = 0; dir < 8; dir++) {
      const int r = dir_opp(dir);
      if (m->nbr[r] < 0) continue;
      if (m->nbr[r] == m->rank && !m->force_nccl_self) continue;
      NCCLCHK(api->Recv(m->rbuf[r], cnt[r], ncclDouble, m->nbr[r], m->comm, st));
    }
    NCCLCHK(api->GroupEnd());
  }
  (void)in_group;
  if (own) KLAUNCH(c, "k_halo_unpack", k_halo_pack, dim3(blocks, 8), dim3(256), d, A, RB, 0);
  else hipLaunchKernelGGL(k_halo_pack, dim3(blocks, 8), dim3(256), 0, st, d, A, RB, 0);
  if (poison_halo()) {
    Nbr8 N;
    for (int dir = 0; dir < 8; dir++) N.on[dir] = (m->nbr[dir] >= 0);
    hipLaunchKernelGGL(k_halo_poison, dim3(512), dim3(256), 0, st, d, A, N);
  }
  return MOM6X_OK;
}

void halo_complete(mom6x_ctx *c);

void halo_wrap(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n) {
  const Dm d = c->d;
  Comm *m = (Comm *)c->comm;
  if (c->pass_pending) halo_complete(c);   // the message buffers are shared: one group pass at a time
  if (!m && !c->dims.reentrant_x && !c->dims.reentrant_y) return;
  for (int base = 0; base < n; base += MAXF) {
    WrapArgs A;
    A.n = (n - base < MAXF) ? (n - base) : MAXF;
    A.rx = c->dims.reentrant_x;
    A.w = (c->pass_w > 0 && c->pass_w < d.halo) ? c->pass_w : d.halo; A.w2 = d.halo;
    int nkmax = 1;
    for (int q = 0; q < A.n; q++) {
      A.f[q] = fields[base + q]; A.stg[q] = staggers[base + q]; A.nk[q] = nks[base + q];
      A.wf[q] = (base + q < c->pass_wf_n && c->pass_wf[base + q] > 0 && c->pass_wf[base + q] < A.w) ? c->pass_wf[base + q] : 0;
      if (A.nk[q] > nkmax) nkmax = A.nk[q];
    }
    if (m) {
      if (exchange(c, m, A, c->stream) != MOM6X_OK) c->halo_error = true;
      continue;
    }
    const dim3 b(64, d.halo, 1);
    if (c->dims.reentrant_x) {
      const int rows = d.nj + 2 * d.halo + 1;
      KLAUNCH(c, "k_wrap_x", k_wrap_x, dim3((rows + 63) / 64, 1, nkmax), b, d, A);
    }
    if (c->dims.reentrant_y) {
      const int cols = d.ni + 2 * d.halo + 1;
      KLAUNCH(c, "k_wrap_y", k_wrap_y, dim3((cols + 63) / 64, 1, nkmax), b, d, A);
    }
    if (poison_halo()) {
      Nbr8 N;
      for (int dir = 0; dir < 8; dir++) {
        int dx, dy; dir_dxdy(dir, dx, dy);
        N.on[dir] = (dx == 0 || c->dims.reentrant_x) && (dy == 0 || c->dims.reentrant_y);
      }
      hipLaunchKernelGGL(k_halo_poison, dim3(512), dim3(256), 0, c->stream, d, A, N);
    }
  }
}

// start_group_pass / complete_group_pass (MOM_domain_infra.F90:1155 / :1173; G%nonblocking_updates): the group pass runs on
// the context's SECOND stream -- pack, RCCL send/recv, unpack -- while the compute stream goes on with kernels that neither
// write the passing fields nor read their halos; halo_complete makes the compute stream wait for it.  At most one pass is
// in flight (a blocking pass completes it first).  Without a communicator (one tile) the pass is done at once.
void halo_start(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || n > MAXF) { halo_wrap(c, fields, staggers, nks, n); return; }
  if (c->pass_pending) halo_complete(c);
  if (!c->ev_ready) {
    if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) { c->halo_error = true; return; }
  }
  WrapArgs A;
  A.n = n; A.rx = c->dims.reentrant_x;
  A.w = (c->pass_w > 0 && c->pass_w < c->dims.halo) ? c->pass_w : c->dims.halo; A.w2 = c->dims.halo;
  for (int q = 0; q < n; q++) {
    A.f[q] = fields[q]; A.stg[q] = staggers[q]; A.nk[q] = nks[q];
    A.wf[q] = (q < c->pass_wf_n && c->pass_wf[q] > 0 && c->pass_wf[q] < A.w) ? c->pass_wf[q] : 0;
  }
  // the second stream starts when the compute stream has produced the fields ...
  if (hipEventRecord(c->ev_ready, c->stream) != hipSuccess || hipStreamWaitEvent(c->halo_stream, c->ev_ready, 0) != hipSuccess ||
      exchange(c, m, A, c->halo_stream) != MOM6X_OK || hipEventRecord(c->ev_done, c->halo_stream) != hipSuccess) {
    c->halo_error = true;
    return;
  }
  c->pass_pending = true;
}
// The same for a pass whose SEND regions the compute stream is about to rewrite (the barotropic sub-cycle: the kernels that run while
// the messages travel update eta in place next to the tile's edge): the pack runs on the compute stream, in order with them; messages
// and unpack on the second stream.  false: no communicator that can overlap (the pass has been made, blocking).
bool halo_start_packed(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || n > MAXF) { halo_wrap(c, fields, staggers, nks, n); return false; }
  if (c->pass_pending) halo_complete(c);
  if (!c->ev_ready) {
    if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) { c->halo_error = true; return false; }
  }
  WrapArgs A;
  A.n = n; A.rx = c->dims.reentrant_x;
  A.w = (c->pass_w > 0 && c->pass_w < c->dims.halo) ? c->pass_w : c->dims.halo; A.w2 = c->dims.halo;
  for (int q = 0; q < n; q++) {
    A.f[q] = fields[q]; A.stg[q] = staggers[q]; A.nk[q] = nks[q];
    A.wf[q] = (q < c->pass_wf_n && c->pass_wf[q] > 0 && c->pass_wf[q] < A.w) ? c->pass_wf[q] : 0;
  }
  if (exchange(c, m, A, c->halo_stream, true) != MOM6X_OK || hipEventRecord(c->ev_done, c->halo_stream) != hipSuccess) {
    c->halo_error = true;
    return false;
  }
  c->pass_pending = true;
  return true;
}
bool halo_can_overlap(const mom6x_ctx *c) { const Comm *m = (const Comm *)c->comm; return m && m->comm; }
// ... and the compute stream waits here for the halos
void halo_complete(mom6x_ctx *c) {
  if (!c->pass_pending) return;
  c->pass_pending = false;
  if (hipStreamWaitEvent(c->stream, c->ev_done, 0) != hipSuccess) c->halo_error = true;
}

// btstep's group pass of eta, ubt, vbt (MOM_barotropic.F90:2505-2512) on the second stream, overlapped with the half of the next
// sub-step that reads the tile's own points only (barotropic.hip: halo_start_packed, k_bt_substep / k_bt_pred sel 1, 2).  Off by
// default: on one GPU with RCCL self-sends -- no wire, the only measurement there is -- the two extra launches and the stream
// dependencies cost more than the 45 us of message kernel and unpack they hide (9.26 -> 9.64 ms per step of the 8-GPU tile,
// profiles/r06_tile.md); a host on a fabric whose latency is longer switches it on.  The answers do not depend on it.
extern "C" int mom6x_comm_overlap_btstep(mom6x_ctx *c, int on) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_comm_overlap_btstep: null ctx");
  c->bt_overlap = (on != 0);
  return MOM6X_OK;
}

// pass_var / pass_vector for non-torch hosts and tests: one group pass of n fields.
extern "C" int mom6x_pass_fields(mom6x_ctx *c, double *const *fields, const int *staggers, const int *nks, int n) {
  REQUIRE(c && fields && staggers && nks && n > 0, MOM6X_EINVAL, "mom6x_pass_fields: bad arguments");
  HIPCHK(hipSetDevice(c->device));
  c->halo_error = false;
  halo_wrap(c, fields, staggers, nks, n);
  REQUIRE(!c->halo_error, MOM6X_EHIP, mom6x_last_error());
  HIPCHK(hipGetLastError());
  return MOM6X_OK;
}

// sum_across_PEs of an int array that lives on the device (MOM_coms.F90; advect_tracer :331), in place.
int comm_allreduce_int_sum(mom6x_ctx *c, int *dev, int n) {
  Comm *m = (Comm *)c->comm;
  if (!m || !m->comm || m->nranks == 1) return MOM6X_OK;
  NcclApi *api = m->api;
  NCCLCHK(api->AllReduce(dev, dev, (size_t)n, ncclInt, ncclSum, m->comm, c->stream));
  return MOM6X_OK;
}

// The halo width of the 3-D fields in the RK2 step's own group passes (RK2.F90's create_group_pass calls run on G%Domain, i.e. with
// NIHALO): for a context whose halo was widened for the barotropic solver (BT_USE_WIDE_HALOS with BTHALO > NIHALO,
// MOM_barotropic.F90:5446-5461).  0: the context's halo.  2-D fields (eta) always travel at the context's width.
extern "C" int mom6x_set_dyn_pass_width(mom6x_ctx *c, int width) {
  REQUIRE(c, MOM6X_EINVAL, "mom6x_set_dyn_pass_width: null ctx");
  REQUIRE(width == 0 || (width >= 4 && width <= c->dims.halo), MOM6X_EINVAL,
          "mom6x_set_dyn_pass_width: the width must be 0 (the context's halo) or between 4 and the context's halo");
  c->dyn_pass_width = width;
  return MOM6X_OK;
}

// Bytes this tile has sent in packed group exchanges since the last reset (all neighbours together).
extern "C" long long mom6x_comm_exchange_bytes(mom6x_ctx *c, int reset) {
  if (!c) return -1;
  const long long n = c->n_exchange_bytes;
  if (reset) c->n_exchange_bytes = 0;
  return n;
}

// Packed group exchanges (one message per neighbour each) since the last reset: what a step costs in message latency.
extern "C" long long mom6x_comm_exchange_count(mom6x_ctx *c, int reset) {
  if (!c) return -1;
  const long long n = c->n_exchanges;
  if (reset) c->n_exchanges = 0;
  return n;
}
