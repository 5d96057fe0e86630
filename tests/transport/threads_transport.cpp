// TEST INFRASTRUCTURE -- a halo transport among host THREADS of one process (one tile per thread, all on the same GPU),
// plugged into the library through mom6x_comm_set_transport (include/mom6x.h).
// A single-GPU box cannot host two RCCL ranks, yet the multi-tile logic -- which rows and columns every kernel covers,
// the wide-halo cycles of the barotropic solver, the global reductions -- is independent of the transport.  This one
// lets tests/test_layout_gpu.py and scripts/check_layout_fullsize.py run a 2 x 1, 2 x 2 or 4 x 2 layout on one device and
// compare it with the one-tile run (the reference's test.layout).  Every operation synchronises the calling tile's stream
// and meets the other tiles at host barriers; it is slow and only meant for tests.  Built by __graft_entry__.build() into
// tests/transport/libmom6x_threads_transport.so.
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "mom6x.h"

namespace {
enum { E_ARG = 1, E_SYS = 2, E_USE = 3, E_HIP = 4 };

struct Op { const void *src; void *dst; size_t bytes; int peer; bool send; };
struct Hub {
  std::mutex mu; std::condition_variable cv;
  int nranks = 0, arrived = 0; long gen = 0, joined = 0;
  std::vector<std::vector<Op>> posted;            // sends of the current group, by sending rank
  std::vector<std::vector<char>> red;             // all-reduce contributions, by rank
  bool barrier(std::unique_lock<std::mutex> &lk) {   // all ranks, reusable; false after 120 s (a tile has failed)
    const long g = gen;
    if (++arrived == nranks) { arrived = 0; gen++; cv.notify_all(); return true; }
    return cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g; });
  }
};
struct TC { Hub *hub; int rank; };
static std::mutex g_mu;
static std::map<std::string, Hub *> g_hubs;
static int g_ids = 0;
static thread_local std::vector<Op> t_ops;
static thread_local TC *t_comm = nullptr;
static thread_local hipStream_t t_stream = nullptr;
static thread_local int t_depth = 0;

static int GetUniqueId(char *id128) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id128, 0, 128);
  snprintf(id128, 128, "mom6x-threads-%d", ++g_ids);
  return 0;
}
static int CommInitRank(void **comm, int nranks, const char *id128, int rank) {
  Hub *h;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    Hub *&slot = g_hubs[std::string(id128)];
    if (!slot) { slot = new Hub(); slot->nranks = nranks; slot->posted.resize(nranks); slot->red.resize(nranks); }
    h = slot;
  }
  if (h->nranks != nranks || rank < 0 || rank >= nranks) return E_ARG;
  *comm = (void *) new TC{h, rank};
  std::unique_lock<std::mutex> lk(h->mu);
  return h->barrier(lk) ? 0 : E_SYS;
}
static int CommDestroy(void *comm) { delete (TC *)comm; return 0; }
static size_t type_size(int t) { return (t == MOM6X_T_INT32) ? 4 : 8; }
static int run_group() {
  TC *c = t_comm; Hub *h = c->hub;
  if (hipStreamSynchronize(t_stream) != hipSuccess) return E_HIP;   // my packed messages are complete
  std::unique_lock<std::mutex> lk(h->mu);
  h->posted[c->rank].clear();
  for (const Op &o : t_ops) if (o.send) h->posted[c->rank].push_back(o);
  if (!h->barrier(lk)) return E_SYS;
  std::vector<size_t> used(h->nranks, 0);
  std::vector<Op> copies;
  for (const Op &o : t_ops) {
    if (o.send) continue;
    const std::vector<Op> &ps = h->posted[o.peer];       // the j-th receive from a peer takes its j-th send to me
    size_t &u = used[o.peer];
    while (u < ps.size() && ps[u].peer != c->rank) u++;
    if (u >= ps.size() || ps[u].bytes != o.bytes) return E_USE;
    copies.push_back(Op{ps[u].src, o.dst, o.bytes, o.peer, false});
    u++;
  }
  lk.unlock();
  // (a device-to-device hipMemcpy may return before the data has landed: copy on the tile's stream and wait for it)
  for (const Op &o : copies)
    if (hipMemcpyAsync(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice, t_stream) != hipSuccess) return E_HIP;
  if (hipStreamSynchronize(t_stream) != hipSuccess) return E_HIP;
  lk.lock();
  const bool ok = h->barrier(lk);                         // nobody reuses a send buffer before everybody has copied
  t_ops.clear();
  return ok ? 0 : E_SYS;
}
static int GroupStart() { t_depth++; return 0; }
static int GroupEnd() { if (--t_depth > 0 || t_ops.empty()) return 0; return run_group(); }
static int Send(const void *buf, size_t n, int t, int peer, void *comm, void *st) {
  t_comm = (TC *)comm; t_stream = (hipStream_t)st;
  t_ops.push_back(Op{buf, nullptr, n * type_size(t), peer, true});
  return (t_depth > 0) ? 0 : run_group();
}
static int Recv(void *buf, size_t n, int t, int peer, void *comm, void *st) {
  t_comm = (TC *)comm; t_stream = (hipStream_t)st;
  t_ops.push_back(Op{nullptr, buf, n * type_size(t), peer, false});
  return (t_depth > 0) ? 0 : run_group();
}
template <class T> static void reduce_into(T *acc, const T *x, size_t n, int op) {
  for (size_t i = 0; i < n; i++) acc[i] = (op == MOM6X_OP_SUM) ? acc[i] + x[i] : ((op == MOM6X_OP_MIN) ? (x[i] < acc[i] ? x[i] : acc[i]) : (x[i] > acc[i] ? x[i] : acc[i]));
}
static int AllReduce(const void *send, void *recv, size_t n, int t, int op, void *comm, void *stv) {
  hipStream_t st = (hipStream_t)stv;
  TC *c = (TC *)comm; Hub *h = c->hub;
  const size_t bytes = n * type_size(t);
  std::vector<char> mine(bytes);
  if (hipStreamSynchronize(st) != hipSuccess) return E_HIP;
  if (hipMemcpy(mine.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return E_HIP;
  std::unique_lock<std::mutex> lk(h->mu);
  h->red[c->rank] = mine;
  if (!h->barrier(lk)) return E_SYS;
  std::vector<char> acc = h->red[0];                      // rank order: every tile forms the same result
  for (int r = 1; r < h->nranks; r++) {
    if (h->red[r].size() != bytes) return E_USE;
    if (t == MOM6X_T_FLOAT64) reduce_into((double *)acc.data(), (const double *)h->red[r].data(), n, op);
    else if (t == MOM6X_T_INT64) reduce_into((long long *)acc.data(), (const long long *)h->red[r].data(), n, op);
    else if (t == MOM6X_T_INT32) reduce_into((int *)acc.data(), (const int *)h->red[r].data(), n, op);
    else return E_ARG;
  }
  const bool ok = h->barrier(lk);                         // everybody has read the contributions
  lk.unlock();
  if (!ok) return E_SYS;
  return (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) == hipSuccess) ? 0 : E_HIP;
}
static const char *GetErrorString(int) { return "the in-process threads transport of the tests failed or timed out"; }

const mom6x_transport g_table = { GetUniqueId, CommInitRank, CommDestroy, Send, Recv, GroupStart, GroupEnd, AllReduce, GetErrorString };
}  // namespace

extern "C" const mom6x_transport *mom6x_threads_transport(void) { return &g_table; }
