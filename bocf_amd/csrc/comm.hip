// Multi-GPU side of the C ABI: one process and one context per GPU, an RCCL communicator owned by the context,
// and the ONE collective of the acquisition path -- the global top-k of the sharded candidate batch -- issued on the
// context's own stream from device buffers (no host hop between the local selection and the collective).
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU process never loads it, and a process that already
// holds an RCCL (torch.distributed's) shares that copy.  No CPU path: without RCCL bocf_comm_init fails loudly.
#include "bocf_ctx.h"

#include <dlfcn.h>
#include <cmath>
#include <cstring>
#include <rccl/rccl.h>

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail("bocf_comm: dlopen(librccl.so.1)", dlerror());
  Rccl r;
  r.handle = h;
#define SYM(field, name)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));      \
  if (!r.field) return fail("bocf_comm: missing RCCL symbol", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Broadcast, "ncclBroadcast");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_rccl = r;
  return 0;
}
}  // namespace

#define NCCLCHK(expr)                                                                   \
  do {                                                                                  \
    ncclResult_t r_ = (expr);                                                           \
    if (r_ != ncclSuccess) return fail(#expr, g_rccl.GetErrorString(r_));               \
  } while (0)

extern "C" int bocf_comm_unique_id(char* id_out) {
  if (!id_out) return fail("bocf_comm_unique_id", "null out");
  if (load_rccl()) return -1;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == BOCF_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int bocf_comm_init(bocf_ctx* c, const char* id_bytes, int world, int rank) {
  if (!c || !id_bytes) return fail("bocf_comm_init", "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("bocf_comm_init", "rank / world out of range");
  if (c->comm) return fail("bocf_comm_init", "the context already has a communicator");
  if (load_rccl()) return -1;
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  NCCLCHK(g_rccl.CommInitRank(&comm, world, id, rank));
  c->comm = comm;
  c->world = world;
  c->rank = rank;
  return 0;
}

extern "C" int bocf_comm_destroy(bocf_ctx* c) {
  if (!c) return fail("bocf_comm_destroy", "null ctx");
  if (!c->comm) return 0;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  NCCLCHK(g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm)));
  c->comm = nullptr;
  c->world = 1;
  c->rank = 0;
  return 0;
}

extern "C" int bocf_comm_info(bocf_ctx* c, int* world_out, int* rank_out) {
  if (!c) return fail("bocf_comm_info", "null ctx");
  if (world_out) *world_out = c->comm ? c->world : 1;
  if (rank_out) *rank_out = c->comm ? c->rank : 0;
  return c->comm ? 1 : 0;
}

// local top-k of the last acquisition vector -> c->out_idx / c->out_val (device), packed into `pack` (device, 2*world*k doubles)
static int local_topk_packed(bocf_ctx* c, int k, long long lo, int world, int rank, double* pack) {
  const int nb = topk_num_blocks(c->C);
  if (c->blk_idx.ensure(sizeof(long long) * (size_t)nb * k) || c->blk_val.ensure(sizeof(double) * (size_t)nb * k) ||
      c->out_idx.ensure(sizeof(long long) * k) || c->out_val.ensure(sizeof(double) * k))
    return -1;
  if (c->C > 0)
    launch_topk(c->acq.as<double>(), c->C, k, c->blk_idx.as<long long>(), c->blk_val.as<double>(), c->out_idx.as<long long>(),
                c->out_val.as<double>(), c->stream);
  launch_pack_topk(c->C > 0 ? c->out_idx.as<long long>() : nullptr, c->out_val.as<double>(), k, lo, world, rank, pack, c->stream);
  return 0;
}

extern "C" int bocf_topk_packed(bocf_ctx* c, int k, long long lo, int world, int rank, void* device_buf) {
  if (!c || !device_buf) return fail("bocf_topk_packed", "null argument");
  if (!c->have_acq && c->C > 0) return fail("bocf_topk_packed", "no acquisition vector on the device");
  if (k < 1 || k > 64 || world < 1 || rank < 0 || rank >= world || lo < 0) return fail("bocf_topk_packed", "k (1..64), world, rank or lo out of range");
  HIPCHK(hipSetDevice(c->device));
  if (local_topk_packed(c, k, lo, world, rank, static_cast<double*>(device_buf))) return -1;
  HIPCHK(hipStreamSynchronize(c->stream));   // the caller's collective runs on ITS stream: the buffer is complete on return
  LAUNCHCHK();
  return 0;
}

extern "C" int bocf_merge_packed(bocf_ctx* c, int k, int world, const void* device_buf, long long* idx_out, double* val_out) {
  if (!c || !device_buf || !idx_out) return fail("bocf_merge_packed", "null argument");
  if (k < 1 || k > 64 || world < 1) return fail("bocf_merge_packed", "k (1..64) or world out of range");
  HIPCHK(hipSetDevice(c->device));
  const int n = world * k;
  if (c->gidx.ensure(sizeof(long long) * n) || c->gval.ensure(sizeof(double) * n) || c->out_idx.ensure(sizeof(long long) * k) ||
      c->out_val.ensure(sizeof(double) * k))
    return -1;
  launch_merge_packed(static_cast<const double*>(device_buf), k, world, c->gidx.as<long long>(), c->gval.as<double>(), c->out_idx.as<long long>(),
                      c->out_val.as<double>(), c->stream);
  HIPCHK(hipMemcpyAsync(idx_out, c->out_idx.p, sizeof(long long) * k, hipMemcpyDeviceToHost, c->stream));
  if (val_out) HIPCHK(hipMemcpyAsync(val_out, c->out_val.p, sizeof(double) * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  LAUNCHCHK();
  return 0;
}

// Every rank that has a communicator REACHES the all-reduce whatever happened locally (a missing acquisition vector, a failed launch):
// a rank that returned early would leave its peers blocked in ncclAllReduce for ever.  The local status travels in a spare slot behind
// the 2 * world * k packed doubles (0 = fine, 1 = failed; MAX over the ranks), and all ranks fail together after the exchange.  A rank
// that cannot even allocate the exchange buffer aborts the communicator: its peers' collective then fails instead of waiting.
extern "C" int bocf_global_topk(bocf_ctx* c, int k, long long lo, long long* idx_out, double* val_out) {
  if (!c || !idx_out) return fail("bocf_global_topk", "null argument");
  if (k < 1 || k > 64 || lo < 0) return fail("bocf_global_topk", "k (1..64) or lo out of range");      // (same arguments on every rank: same verdict)
  const int world = c->comm ? c->world : 1, rank = c->comm ? c->rank : 0;
  const int n = world * k;
  int local_rc = 0;
  std::string local_err;
  auto note = [&](int rc) {
    if (rc != 0 && local_rc == 0) {
      local_rc = -1;
      local_err = bocf_last_error();
    }
  };
  if (hipSetDevice(c->device) != hipSuccess) note(fail("bocf_global_topk", "hipSetDevice"));
  if (c->pack.ensure(sizeof(double) * (2 * n + 1)) || c->gidx.ensure(sizeof(long long) * n) || c->gval.ensure(sizeof(double) * n)) {
    if (c->comm) (void)bocf_comm_abort(c);
    return -1;
  }
  double* pack = c->pack.as<double>();
  if (local_rc == 0 && !c->have_acq && c->C > 0) note(fail("bocf_global_topk", "no acquisition vector on the device"));
  if (local_rc == 0) note(local_topk_packed(c, k, lo, world, rank, pack));
  if (local_rc == 0) note(bocf_launch_status());
  if (local_rc != 0) {                                     // nothing usable from this rank: -inf everywhere (the neutral element of MAX)
    std::vector<double> neutral((size_t)2 * n, -INFINITY);
    (void)hipMemcpyAsync(pack, neutral.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream);
    (void)hipStreamSynchronize(c->stream);
  }
  double status = local_rc != 0 ? 1.0 : 0.0;
  (void)hipMemcpyAsync(pack + 2 * n, &status, sizeof(double), hipMemcpyHostToDevice, c->stream);
  int comm_rc = 0;
  PhaseTimer t_comm(c, "allreduce");                       // (option "profile": HIP events around the collective itself, on the context's stream)
  if (c->comm) {   // ONE all-reduce(MAX) over xGMI, in place, on the context's stream (RCCL has no MAXLOC: values | indices, -inf elsewhere | status)
    const ncclResult_t r = g_rccl.AllReduce(c->pack.p, c->pack.p, (size_t)2 * n + 1, ncclDouble, ncclMax, static_cast<ncclComm_t>(c->comm), c->stream);
    if (r != ncclSuccess) comm_rc = fail("ncclAllReduce", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  }
  t_comm.stop();
  double any_failed = status;
  if (comm_rc == 0) {
    launch_merge_packed(pack, k, world, c->gidx.as<long long>(), c->gval.as<double>(), c->out_idx.as<long long>(), c->out_val.as<double>(), c->stream);
    (void)hipMemcpyAsync(&any_failed, pack + 2 * n, sizeof(double), hipMemcpyDeviceToHost, c->stream);
    (void)hipMemcpyAsync(idx_out, c->out_idx.p, sizeof(long long) * k, hipMemcpyDeviceToHost, c->stream);
    if (val_out) (void)hipMemcpyAsync(val_out, c->out_val.p, sizeof(double) * k, hipMemcpyDeviceToHost, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) comm_rc = fail("bocf_global_topk", "stream synchronisation failed");
  }
  if (local_rc != 0) return fail("bocf_global_topk: this rank failed before the exchange", local_err.c_str());
  if (comm_rc != 0) return -1;
  if (any_failed > 0.0) return fail("bocf_global_topk", "another rank failed before the exchange (its own error names the cause)");
  LAUNCHCHK();
  return 0;
}

// Used by the sharded fit (capi.hip): broadcast `count` doubles from rank `root` on the context's stream.
int bocf_comm_broadcast(bocf_ctx* c, double* buf, size_t count, int root) {
  if (!c->comm) return fail("bocf_comm_broadcast", "no communicator");
  NCCLCHK(g_rccl.Broadcast(buf, buf, count, ncclDouble, root, static_cast<ncclComm_t>(c->comm), c->stream));
  return 0;
}
// A rank that cannot take part in an agreed collective tears the communicator down so that its peers fail instead of blocking.
int bocf_comm_abort(bocf_ctx* c) {
  if (!c->comm) return 0;
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort(static_cast<ncclComm_t>(c->comm));
  c->comm = nullptr;
  c->world = 1;
  c->rank = 0;
  return 0;
}
int bocf_comm_allreduce_sum(bocf_ctx* c, double* buf, size_t count) {
  if (!c->comm) return fail("bocf_comm_allreduce_sum", "no communicator");
  NCCLCHK(g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
  return 0;
}
int bocf_comm_group(bool start) {
  if (!g_rccl.handle) return fail("bocf_comm_group", "RCCL not loaded");
  NCCLCHK(start ? g_rccl.GroupStart() : g_rccl.GroupEnd());
  return 0;
}
