// hsad_comm_*: the multi-GPU exchange of a self-play job as C entry points (SURVEY.md section 8e / 8b last row) -- what
// hanabi_sad_amd/dist.py ReplayLink does with torch.distributed, for a host that is not Python: one process per GPU, RCCL over xGMI.
//
//   reference                                                      here
//   BatchRunner::updateModel across devices (batch_runner.h:74-77)  hsad_comm_bcast_params: ONE flat bucket, one ncclBroadcast
//   PrioritizedReplay::sample over all actors' data (208-257)       hsad_comm_gather_batch: 16-byte all-gather of the shard statistics,
//                                                                   device-side stratification + draw (hsad_replay_serve), ONE
//                                                                   fixed-size message per rank in a grouped send / recv
//   PrioritizedReplay::updatePriority                               hsad_comm_scatter_priority: the whole [B] vector is broadcast
//                                                                   (512 B) and every shard keeps what it owns
//
// RCCL is bound at run time (dlopen): libhsad.so has no link-time dependency on it, and a process that already loaded a copy (PyTorch
// ships one) keeps using that copy.  Every call is stream-ordered; nothing here synchronises the host.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <new>

#include "hsad.h"

extern "C" int hsad_internal_set_error(int code, const char* msg);

namespace {

int cfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return hsad_internal_set_error(code, buf);
}

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)   // a copy that is already in the process first
    if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
  for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) return nullptr;
#define BIND(field, sym)                                                   \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym));        \
  if (!r.field) {                                                          \
    r.lib = nullptr;                                                       \
    return nullptr;                                                        \
  }
  BIND(GetUniqueId, "ncclGetUniqueId")
  BIND(CommInitRank, "ncclCommInitRank")
  BIND(CommDestroy, "ncclCommDestroy")
  BIND(Broadcast, "ncclBroadcast")
  BIND(AllGather, "ncclAllGather")
  BIND(Send, "ncclSend")
  BIND(Recv, "ncclRecv")
  BIND(GroupStart, "ncclGroupStart")
  BIND(GroupEnd, "ncclGroupEnd")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  return &r;
}

#define NCCL_TRY(expr)                                                                                          \
  do {                                                                                                          \
    ncclResult_t r_ = (expr);                                                                                   \
    if (r_ != ncclSuccess) return cfail(HSAD_ERR_HIP, "%s failed: %s", #expr, R->GetErrorString(r_));           \
  } while (0)
#define HIP_TRY(expr)                                                                                \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return cfail(HSAD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct hsad_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  double* my_stats = nullptr;   // [2]
  double* all_stats = nullptr;  // [world][2]
  float* prio = nullptr;        // [1024] broadcast landing zone of scatter_priority
  double* next_stats = nullptr; // [world][2] root of a star round: what the replies of the previous round reported
  // pipelined rounds (the `down` communicator owns them): hdr_ready = this round's header is complete on the down stream, recorded by
  // hsad_comm_star_open BEFORE its sends; sync = scratch event of the prime leg.  Persistent: no create / destroy per round, none leaked
  // on an error return
  hipEvent_t hdr_ready = nullptr, sync = nullptr;
};
static int comm_events(hsad_comm* c) {
  if (!c->hdr_ready && hipEventCreateWithFlags(&c->hdr_ready, hipEventDisableTiming) != hipSuccess) return HSAD_ERR_HIP;
  if (!c->sync && hipEventCreateWithFlags(&c->sync, hipEventDisableTiming) != hipSuccess) return HSAD_ERR_HIP;
  return HSAD_OK;
}

extern "C" {

int hsad_comm_unique_id(void* out, int out_bytes) {
  Rccl* R = rccl();
  if (!R) return cfail(HSAD_ERR_STATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  if (!out || out_bytes < (int)sizeof(ncclUniqueId)) return cfail(HSAD_ERR_INVALID, "unique id needs %d bytes", (int)sizeof(ncclUniqueId));
  NCCL_TRY(R->GetUniqueId(static_cast<ncclUniqueId*>(out)));
  return HSAD_OK;
}

int hsad_comm_init(const void* unique_id, int rank, int world, int device, hsad_comm** out) {
  Rccl* R = rccl();
  if (!R) return cfail(HSAD_ERR_STATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  if (!unique_id || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return cfail(HSAD_ERR_INVALID, "comm_init: bad arguments");
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  hsad_comm* c = new (std::nothrow) hsad_comm();
  if (!c) return cfail(HSAD_ERR_NOMEM, "host allocation failed");
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId id = *static_cast<const ncclUniqueId*>(unique_id);
  ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return cfail(HSAD_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString(r));
  }
  if (hipMalloc((void**)&c->my_stats, 16) != hipSuccess || hipMalloc((void**)&c->all_stats, 16 * (size_t)world) != hipSuccess ||
      hipMalloc((void**)&c->prio, 4096) != hipSuccess || hipMalloc((void**)&c->next_stats, 16 * (size_t)world) != hipSuccess ||
      hipMemset(c->next_stats, 0, 16 * (size_t)world) != hipSuccess) {
    hsad_comm_destroy(c);
    return cfail(HSAD_ERR_NOMEM, "hipMalloc failed for the communicator scratch");
  }
  *out = c;
  return HSAD_OK;
}

void hsad_comm_destroy(hsad_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  Rccl* R = rccl();
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  if (c->my_stats) (void)hipFree(c->my_stats);
  if (c->all_stats) (void)hipFree(c->all_stats);
  if (c->prio) (void)hipFree(c->prio);
  if (c->next_stats) (void)hipFree(c->next_stats);
  if (c->hdr_ready) (void)hipEventDestroy(c->hdr_ready);
  if (c->sync) (void)hipEventDestroy(c->sync);
  delete c;
}

int hsad_comm_rank(const hsad_comm* c) { return c ? c->rank : -1; }
int hsad_comm_world(const hsad_comm* c) { return c ? c->world : 0; }

int hsad_comm_bcast_params(hsad_comm* c, float* params, int64_t count, int root, void* stream) {
  Rccl* R = rccl();
  if (!R || !c || !params || count < 1 || root < 0 || root >= c->world) return cfail(HSAD_ERR_INVALID, "comm_bcast_params: bad arguments");
  NCCL_TRY(R->Broadcast(params, params, (size_t)count, ncclFloat32, root, c->comm, (hipStream_t)stream));
  return HSAD_OK;
}

int hsad_comm_gather_batch(hsad_comm* c, hsad_replay* shard, int batch, const float* canon, int root, int32_t* owner_out,
                           uint8_t* wire_mine, uint8_t* wire_all, void* stream) {
  Rccl* R = rccl();
  if (!R || !c || !shard || !canon || !owner_out || !wire_mine || root < 0 || root >= c->world || (c->rank == root && !wire_all))
    return cfail(HSAD_ERR_INVALID, "comm_gather_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int rc = hsad_replay_stats(shard, c->my_stats, stream);
  if (rc) return rc;
  NCCL_TRY(R->AllGather(c->my_stats, c->all_stats, 2, ncclFloat64, c->comm, s));
  rc = hsad_replay_serve(shard, batch, canon, c->all_stats, c->world, c->rank, owner_out, wire_mine, stream);
  if (rc) return rc;
  const size_t bytes = (size_t)batch * hsad_replay_wire_bytes(shard);
  NCCL_TRY(R->GroupStart());
  if (c->rank == root) {
    for (int k = 0; k < c->world; ++k) {
      if (k == root) continue;
      NCCL_TRY(R->Recv(wire_all + (size_t)k * bytes, bytes, ncclUint8, k, c->comm, s));
    }
  } else {
    NCCL_TRY(R->Send(wire_mine, bytes, ncclUint8, root, c->comm, s));
  }
  NCCL_TRY(R->GroupEnd());
  if (c->rank == root) HIP_TRY(hipMemcpyAsync(wire_all + (size_t)root * bytes, wire_mine, bytes, hipMemcpyDeviceToDevice, s));
  return HSAD_OK;
}

int hsad_comm_scatter_priority(hsad_comm* c, hsad_replay* shard, int batch, const float* priority, const int32_t* owner, int root,
                               void* stream) {
  Rccl* R = rccl();
  if (!R || !c || !shard || !owner || batch < 1 || batch > 1024 || root < 0 || root >= c->world || (c->rank == root && !priority))
    return cfail(HSAD_ERR_INVALID, "comm_scatter_priority: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  NCCL_TRY(R->Broadcast(c->rank == root ? priority : c->prio, c->prio, (size_t)batch, ncclFloat32, root, c->comm, s));
  return hsad_replay_update_owned(shard, batch, c->prio, owner, c->rank, stream);
}

// The star-shaped round of dist.ReplayLink (mode "star") for a host that is not Python: point-to-point messages between the root
// (the learner) and each other rank only, so no actor waits for another one.
//   root:   [PRIME: recv every rank's (sum, size)] -> header tail = the statistics the previous replies reported -> its own serve
//           (+ late priorities) -> ONE group: header to every rank, every rank's rows and new statistics back -> [PARAMS: bucket out]
//   others: [PRIME: send (sum, size)] -> recv header -> serve from the header's (older) statistics -> late priorities -> rows and
//           new statistics to root -> [PARAMS: recv bucket]
int hsad_comm_star_round(hsad_comm* c, hsad_replay* shard, int batch, float* hdr, int flags, int root, const int32_t* answer_owner,
                         int32_t* owner_out, uint8_t* wire_mine, uint8_t* wire_all, float* params, int64_t param_count, void* stream) {
  Rccl* R = rccl();
  if (!R || !c || !shard || !hdr || !owner_out || !wire_mine || batch < 1 || batch > 1024 || root < 0 || root >= c->world ||
      (c->rank == root && !wire_all) || ((flags & HSAD_LINK_HAS_PRIO) && !answer_owner) || ((flags & HSAD_LINK_PARAMS) && (!params || param_count < 1)))
    return cfail(HSAD_ERR_INVALID, "comm_star_round: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int W = c->world, me = c->rank;
  const size_t bytes = (size_t)batch * hsad_replay_wire_bytes(shard), hdr_n = 2 * (size_t)batch + 4 * (size_t)W;
  double* hdr_stats = reinterpret_cast<double*>(hdr + 2 * batch);
  int rc;
  if (me != root) {
    if (flags & HSAD_LINK_PRIME) {
      if ((rc = hsad_replay_stats(shard, c->my_stats, stream))) return rc;
      NCCL_TRY(R->Send(c->my_stats, 2, ncclFloat64, root, c->comm, s));
    }
    NCCL_TRY(R->Recv(hdr, hdr_n, ncclFloat32, root, c->comm, s));
    if ((rc = hsad_replay_serve(shard, batch, hdr, hdr_stats, W, me, owner_out, wire_mine, stream))) return rc;
    if ((flags & HSAD_LINK_HAS_PRIO) && (rc = hsad_replay_update_owned(shard, batch, hdr + batch, answer_owner, me, stream))) return rc;
    if ((rc = hsad_replay_stats(shard, c->my_stats, stream))) return rc;
    NCCL_TRY(R->GroupStart());
    NCCL_TRY(R->Send(wire_mine, bytes, ncclUint8, root, c->comm, s));
    NCCL_TRY(R->Send(c->my_stats, 2, ncclFloat64, root, c->comm, s));
    NCCL_TRY(R->GroupEnd());
    if (flags & HSAD_LINK_PARAMS) NCCL_TRY(R->Recv(params, (size_t)param_count, ncclFloat32, root, c->comm, s));
    return HSAD_OK;
  }
  if (flags & HSAD_LINK_PRIME) {
    NCCL_TRY(R->GroupStart());
    for (int k = 0; k < W; ++k)
      if (k != root) NCCL_TRY(R->Recv(c->next_stats + 2 * k, 2, ncclFloat64, k, c->comm, s));
    NCCL_TRY(R->GroupEnd());
    if ((rc = hsad_replay_stats(shard, c->next_stats + 2 * root, stream))) return rc;
  }
  HIP_TRY(hipMemcpyAsync(hdr_stats, c->next_stats, 16 * (size_t)W, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(c->all_stats, c->next_stats, 16 * (size_t)W, hipMemcpyDeviceToDevice, s));   // hsad_comm_all_stats: what THIS draw was cut with
  if ((rc = hsad_replay_serve(shard, batch, hdr, hdr_stats, W, me, owner_out, wire_mine, stream))) return rc;
  if ((flags & HSAD_LINK_HAS_PRIO) && (rc = hsad_replay_update_owned(shard, batch, hdr + batch, answer_owner, me, stream))) return rc;
  if ((rc = hsad_replay_stats(shard, c->next_stats + 2 * root, stream))) return rc;
  HIP_TRY(hipMemcpyAsync(wire_all + (size_t)root * bytes, wire_mine, bytes, hipMemcpyDeviceToDevice, s));
  NCCL_TRY(R->GroupStart());
  for (int k = 0; k < W; ++k) {
    if (k == root) continue;
    NCCL_TRY(R->Send(hdr, hdr_n, ncclFloat32, k, c->comm, s));
    NCCL_TRY(R->Recv(wire_all + (size_t)k * bytes, bytes, ncclUint8, k, c->comm, s));
    NCCL_TRY(R->Recv(c->next_stats + 2 * k, 2, ncclFloat64, k, c->comm, s));
  }
  NCCL_TRY(R->GroupEnd());
  if (flags & HSAD_LINK_PARAMS) {
    NCCL_TRY(R->GroupStart());
    for (int k = 0; k < W; ++k)
      if (k != root) NCCL_TRY(R->Send(params, (size_t)param_count, ncclFloat32, k, c->comm, s));
    NCCL_TRY(R->GroupEnd());
  }
  return HSAD_OK;
}

// ---- the same round in two halves, for rounds that overlap (dist.py ReplayLink(ahead = 3); the reference's prefetch depth) --------
// Operations on one RCCL communicator run in issue order and send / recv rendezvous: on ONE communicator the header of round r + 1
// queues behind the receive of round r's replies.  So the root drives TWO communicators -- `down` (headers, parameters) on one stream,
// `up` (rows, statistics) on another -- and keeps a ring of per-round buffers; an actor rank passes the same two communicators to
// hsad_comm_star_serve and stays on its own stream.  hsad_comm_star_open: root, first half (header of THIS round out).
// hsad_comm_star_collect: root, second half (own shard served, replies received into this round's buffers).
int hsad_comm_star_open(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, const double* stats_known,
                        double* stats_prime, float* params, int64_t param_count, void* stream_down, void* stream_up) {
  Rccl* R = rccl();
  if (!R || !down || !up || !shard || !hdr || batch < 1 || batch > 1024 || down->world != up->world || down->rank != up->rank ||
      ((flags & HSAD_LINK_PRIME) ? !stats_prime : !stats_known) || ((flags & HSAD_LINK_PARAMS) && (!params || param_count < 1)))
    return cfail(HSAD_ERR_INVALID, "comm_star_open: bad arguments");
  hipStream_t sd = (hipStream_t)stream_down, su = (hipStream_t)stream_up;
  const int W = down->world, root = down->rank;
  const size_t hdr_n = 2 * (size_t)batch + 4 * (size_t)W;
  double* hdr_stats = reinterpret_cast<double*>(hdr + 2 * batch);
  int rc;
  if ((rc = comm_events(down))) return cfail(rc, "comm_star_open: hipEventCreate failed");
  if (flags & HSAD_LINK_PRIME) {        // nobody has reported a (sum, size) yet: collect them up front (up), then the header may leave
    NCCL_TRY(R->GroupStart());
    for (int k = 0; k < W; ++k)
      if (k != root) NCCL_TRY(R->Recv(stats_prime + 2 * k, 2, ncclFloat64, k, up->comm, su));
    NCCL_TRY(R->GroupEnd());
    if ((rc = hsad_replay_stats(shard, stats_prime + 2 * root, stream_up))) return rc;
    HIP_TRY(hipEventRecord(down->sync, su));
    HIP_TRY(hipStreamWaitEvent(sd, down->sync, 0));
    stats_known = stats_prime;
  }
  HIP_TRY(hipMemcpyAsync(hdr_stats, stats_known, 16 * (size_t)W, hipMemcpyDeviceToDevice, sd));
  // the header is complete HERE.  hsad_comm_star_collect makes the up stream wait for this point and not for the sends below: the root's
  // receive of an actor's rows must not queue behind the parameter send to that actor -- the actor posts its parameter receive only after
  // its row send has completed, and a 37 MB parameter send does not complete eagerly (a cycle with more than one rank)
  HIP_TRY(hipEventRecord(down->hdr_ready, sd));
  NCCL_TRY(R->GroupStart());
  for (int k = 0; k < W; ++k)
    if (k != root) NCCL_TRY(R->Send(hdr, hdr_n, ncclFloat32, k, down->comm, sd));
  NCCL_TRY(R->GroupEnd());
  if (flags & HSAD_LINK_PARAMS) {
    NCCL_TRY(R->GroupStart());
    for (int k = 0; k < W; ++k)
      if (k != root) NCCL_TRY(R->Send(params, (size_t)param_count, ncclFloat32, k, down->comm, sd));
    NCCL_TRY(R->GroupEnd());
  }
  return HSAD_OK;
}

// root, second half: call it right behind hsad_comm_star_open (the up stream waits for the event that call recorded behind its header
// copy and in front of its sends).  stats_reply [world][2]: where this round's replies put their statistics
// (what a LATER hsad_comm_star_open passes as stats_known once the host has collected this round).
int hsad_comm_star_collect(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, const int32_t* answer_owner,
                           int32_t* owner_out, uint8_t* wire_all, double* stats_reply, void* stream_down, void* stream_up) {
  Rccl* R = rccl();
  if (!R || !down || !up || !shard || !hdr || !owner_out || !wire_all || !stats_reply || batch < 1 || batch > 1024 ||
      ((flags & HSAD_LINK_HAS_PRIO) && !answer_owner))
    return cfail(HSAD_ERR_INVALID, "comm_star_collect: bad arguments");
  hipStream_t sd = (hipStream_t)stream_down, su = (hipStream_t)stream_up;
  const int W = up->world, root = up->rank;
  const size_t bytes = (size_t)batch * hsad_replay_wire_bytes(shard);
  double* hdr_stats = reinterpret_cast<double*>(hdr + 2 * batch);
  int rc;
  (void)sd;
  if (!down->hdr_ready) return cfail(HSAD_ERR_STATE, "comm_star_collect: no hsad_comm_star_open before it on this communicator");
  HIP_TRY(hipStreamWaitEvent(su, down->hdr_ready, 0));   // the header (uniforms, priorities, statistics) as star_open completed it, NOT its sends
  NCCL_TRY(R->GroupStart());             // posted before the own shard is served: the replies travel meanwhile
  for (int k = 0; k < W; ++k) {
    if (k == root) continue;
    NCCL_TRY(R->Recv(wire_all + (size_t)k * bytes, bytes, ncclUint8, k, up->comm, su));
    NCCL_TRY(R->Recv(stats_reply + 2 * k, 2, ncclFloat64, k, up->comm, su));
  }
  NCCL_TRY(R->GroupEnd());
  if ((rc = hsad_replay_serve(shard, batch, hdr, hdr_stats, W, root, owner_out, wire_all + (size_t)root * bytes, stream_up))) return rc;
  if ((flags & HSAD_LINK_HAS_PRIO) && (rc = hsad_replay_update_owned(shard, batch, hdr + batch, answer_owner, root, stream_up))) return rc;
  return hsad_replay_stats(shard, stats_reply + 2 * root, stream_up);
}

// an actor rank's side of a pipelined round (one stream; the two communicators only keep the root's two directions apart)
int hsad_comm_star_serve(hsad_comm* down, hsad_comm* up, hsad_replay* shard, int batch, float* hdr, int flags, int root,
                         const int32_t* answer_owner, int32_t* owner_out, uint8_t* wire_mine, float* params, int64_t param_count, void* stream) {
  Rccl* R = rccl();
  if (!R || !down || !up || !shard || !hdr || !owner_out || !wire_mine || batch < 1 || batch > 1024 || root < 0 || root >= up->world ||
      up->rank == root || ((flags & HSAD_LINK_HAS_PRIO) && !answer_owner) || ((flags & HSAD_LINK_PARAMS) && (!params || param_count < 1)))
    return cfail(HSAD_ERR_INVALID, "comm_star_serve: bad arguments (an actor rank's call)");
  hipStream_t s = (hipStream_t)stream;
  const int W = up->world, me = up->rank;
  const size_t bytes = (size_t)batch * hsad_replay_wire_bytes(shard), hdr_n = 2 * (size_t)batch + 4 * (size_t)W;
  double* hdr_stats = reinterpret_cast<double*>(hdr + 2 * batch);
  int rc;
  if (flags & HSAD_LINK_PRIME) {
    if ((rc = hsad_replay_stats(shard, up->my_stats, stream))) return rc;
    NCCL_TRY(R->Send(up->my_stats, 2, ncclFloat64, root, up->comm, s));
  }
  NCCL_TRY(R->Recv(hdr, hdr_n, ncclFloat32, root, down->comm, s));
  if ((rc = hsad_replay_serve(shard, batch, hdr, hdr_stats, W, me, owner_out, wire_mine, stream))) return rc;
  if ((flags & HSAD_LINK_HAS_PRIO) && (rc = hsad_replay_update_owned(shard, batch, hdr + batch, answer_owner, me, stream))) return rc;
  if ((rc = hsad_replay_stats(shard, up->my_stats, stream))) return rc;
  NCCL_TRY(R->GroupStart());
  NCCL_TRY(R->Send(wire_mine, bytes, ncclUint8, root, up->comm, s));
  NCCL_TRY(R->Send(up->my_stats, 2, ncclFloat64, root, up->comm, s));
  NCCL_TRY(R->GroupEnd());
  if (flags & HSAD_LINK_PARAMS) NCCL_TRY(R->Recv(params, (size_t)param_count, ncclFloat32, root, down->comm, s));
  return HSAD_OK;
}

// ---- one-sided transport between the processes of a node: landing buffers exported by IPC handle, written by the SENDER with a plain
// device-to-device copy (SDMA engines over xGMI between GPUs; no kernel resident on either side), announced through the rendezvous store.
// dist.py ReplayLink(transport = "ipc") builds the learner <-> actor rounds on it: a posted RCCL receive is a kernel that sits on CUs until
// its peer sends, and the learner's whole-chip persistent launches cannot start next to it (measured: tools/resident_probe.py) ----
int hsad_ipc_alloc(int64_t bytes, void** dev_ptr, void* handle_out, int handle_bytes) {
  if (bytes < 1 || !dev_ptr || !handle_out || handle_bytes < (int)sizeof(hipIpcMemHandle_t))
    return cfail(HSAD_ERR_INVALID, "ipc_alloc: bad arguments (the handle needs %d bytes)", (int)sizeof(hipIpcMemHandle_t));
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, (size_t)bytes));
  hipError_t e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return cfail(HSAD_ERR_HIP, "ipc_alloc: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set for dmabuf IPC)", hipGetErrorString(e));
  }
  *dev_ptr = p;
  return HSAD_OK;
}
int hsad_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }
int hsad_ipc_free(void* dev_ptr) {
  if (dev_ptr) HIP_TRY(hipFree(dev_ptr));
  return HSAD_OK;
}
int hsad_ipc_open(const void* handle, void** dev_ptr) {
  if (!handle || !dev_ptr) return cfail(HSAD_ERR_INVALID, "ipc_open: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  HIP_TRY(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return HSAD_OK;
}
int hsad_ipc_close(void* dev_ptr) {
  if (dev_ptr) HIP_TRY(hipIpcCloseMemHandle(dev_ptr));
  return HSAD_OK;
}
/* the put: bytes from this process's device memory into a mapped landing buffer, stream-ordered on the sender */
int hsad_ipc_put(void* dst_mapped, const void* src, int64_t bytes, void* stream) {
  if (!dst_mapped || !src || bytes < 1) return cfail(HSAD_ERR_INVALID, "ipc_put: bad arguments");
  HIP_TRY(hipMemcpyAsync(dst_mapped, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return HSAD_OK;
}

const double* hsad_comm_all_stats(const hsad_comm* c) { return c ? c->all_stats : nullptr; }

}  // extern "C"
