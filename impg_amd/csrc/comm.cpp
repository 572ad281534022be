// Transports of the sharded engine (comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>

namespace impg {

// ---- one rank ------------------------------------------------------------------------
void SelfComm::allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) { memcpy(all, mine, k * 8); }
void SelfComm::alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                         const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) {
  if (send_bytes[0] != recv_bytes[0]) throw Error{IMPG_E_INVALID, "alltoallv: block sizes disagree"};
  if (send_bytes[0])
    IMPG_HIP(hipMemcpyAsync((char *)d_recv + recv_off[0], (const char *)d_send + send_off[0], send_bytes[0], hipMemcpyDeviceToDevice, s));
  IMPG_HIP(hipStreamSynchronize(s));
}

// ---- threads of one process ---------------------------------------------------------------
void LocalFabric::wait_all() {
  std::unique_lock<std::mutex> lk(m);
  if (broken) throw Error{IMPG_E_HIP, "a peer rank failed"};
  const uint64_t gen = generation;
  if (++arrived == world) {
    arrived = 0;
    generation++;
    cv.notify_all();
    return;
  }
  cv.wait(lk, [&] { return generation != gen || broken; });
  if (broken) throw Error{IMPG_E_HIP, "a peer rank failed"};
}
void LocalFabric::poison() {
  std::lock_guard<std::mutex> lk(m);
  broken = true;
  cv.notify_all();
}
void LocalComm::allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) {
  fab->slots[rank].vals = mine;
  fab->wait_all();
  for (int r = 0; r < world; r++) memcpy(all + (size_t)r * k, fab->slots[r].vals, k * 8);
  fab->wait_all();  // nobody overwrites its values while a peer still reads them
}
void LocalComm::alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                          const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) {
  IMPG_HIP(hipStreamSynchronize(s));  // my send blocks are complete before a peer reads them
  LocalFabric::Slot &me = fab->slots[rank];
  me.send = d_send; me.send_off = send_off; me.send_bytes = send_bytes; me.device = device;
  fab->wait_all();
  for (int k = 0; k < world; k++) {
    const int src = (rank + k) % world;  // every rank starts with a different peer: the links are used evenly
    const LocalFabric::Slot &p = fab->slots[src];
    if (p.send_bytes[rank] != recv_bytes[src]) { fab->poison(); throw Error{IMPG_E_INVALID, "alltoallv: block sizes disagree"}; }
    if (!recv_bytes[src]) continue;
    const char *from = (const char *)p.send + p.send_off[rank];
    char *to = (char *)d_recv + recv_off[src];
    if (p.device == device) IMPG_HIP(hipMemcpyAsync(to, from, recv_bytes[src], hipMemcpyDeviceToDevice, s));
    else IMPG_HIP(hipMemcpyPeerAsync(to, device, from, p.device, recv_bytes[src], s));
  }
  IMPG_HIP(hipStreamSynchronize(s));
  fab->wait_all();  // every peer has read my blocks: the send buffer may be reused
}

// ---- RCCL (opened at run time) ----------------------------------------------------------------
namespace {
struct Uid { char b[128]; };  // ncclUniqueId, passed by value
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Uid, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommAbort)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) { r.err = std::string("librccl not found: ") + dlerror(); return; }
    auto sym = [&](const char *n) {
      void *p = dlsym(r.h, n);
      if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  if (!r.err.empty()) throw Error{IMPG_E_HIP, r.err};
  return r;
}
constexpr int NCCL_INT8 = 0, NCCL_UINT64 = 5;  // ncclDataType_t (rccl.h: ncclInt8 = 0 ... ncclUint64 = 5)
void nccl_check(int rc, const char *what) {
  if (rc != 0) throw Error{IMPG_E_HIP, std::string(what) + ": " + rccl().GetErrorString(rc)};
}
}  // namespace

// ---- issue order of a rank's lanes (comm.hpp) ----
void IssueOrder::pass_from(int lane) {
  const int n = (int)active.size();
  for (int k = 1; k <= n; k++) {
    const int l = (lane + k) % n;
    if (active[(size_t)l]) { turn = l; return; }
  }
  in_batch = false;  // nobody left
}
void IssueOrder::begin(int lane, bool takes_part) {
  std::lock_guard<std::mutex> lk(m);
  if (!in_batch) { in_batch = true; turn = -1; }
  active[(size_t)lane] = takes_part ? 1 : 0;
  // the first turn belongs to the lowest lane that takes part (every lane's begin runs before any lane issues)
  turn = -1;
  for (size_t l = 0; l < active.size(); l++) if (active[l]) { turn = (int)l; break; }
  if (turn < 0) in_batch = false;
  cv.notify_all();
}
void IssueOrder::end(int lane) {
  std::lock_guard<std::mutex> lk(m);
  if (!active[(size_t)lane]) return;
  active[(size_t)lane] = 0;
  if (turn == lane) pass_from(lane);
  else {
    bool any = false;
    for (char a : active) any = any || a;
    if (!any) in_batch = false;
  }
  cv.notify_all();
}
void IssueOrder::acquire(int lane) {
  std::unique_lock<std::mutex> lk(m);
  cv.wait(lk, [&] { return !in_batch || !active[(size_t)lane] || turn == lane; });
}
void IssueOrder::release(int lane) {
  std::lock_guard<std::mutex> lk(m);
  if (in_batch && active[(size_t)lane] && turn == lane) pass_from(lane);
  cv.notify_all();
}
namespace {
struct Turn {  // a lane's turn to enqueue, given up when the enqueueing is done (or fails)
  IssueOrder *o;
  int lane;
  bool held = true;
  Turn(IssueOrder *o_, int lane_) : o(o_), lane(lane_) { if (o) o->acquire(lane); }
  void done() { if (o && held) o->release(lane); held = false; }
  ~Turn() { done(); }
};
}  // namespace

void rccl_unique_id(uint8_t *id128) {
  memset(id128, 0, RCCL_UNIQUE_ID_BYTES);
  nccl_check(rccl().GetUniqueId(id128), "ncclGetUniqueId");
}
RcclComm::RcclComm(const uint8_t *id128, int rank_, int world_, int device_) : device(device_) {
  rank = rank_;
  world = world_;
  IMPG_HIP(hipSetDevice(device));
  Uid uid;
  memcpy(uid.b, id128, 128);
  nccl_check(rccl().CommInitRank(&comm, world, uid, rank), "ncclCommInitRank");
  IMPG_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
}
void RcclComm::abort() {
  // (called from any lane's thread: `dead` flips once; `comm` itself stays as it is until the destructor, so that the
  // owner thread, which may be inside a call on it right now, never sees a null handle -- ncclCommAbort makes that call return)
  if (comm && !dead.exchange(true)) {
    if (rccl().CommAbort) (void)rccl().CommAbort(comm);  // frees the communicator; peers' pending operations fail or time out
  }
  if (order) order->end(lane);
}
RcclComm::~RcclComm() {
  if (comm && !dead) (void)rccl().CommDestroy(comm);  // (an aborted communicator is already freed)
  if (h_vals) (void)hipHostFree(h_vals);
  if (cs) (void)hipStreamDestroy(cs);
}
void RcclComm::allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) {
  if (dead) throw Error{IMPG_E_HIP, "the RCCL communicator was aborted after an earlier failure"};
  if (world == 1) { memcpy(all, mine, k * 8); return; }  // (nothing to gather: no collective, no staging)
  IMPG_HIP(hipSetDevice(device));
  const size_t need = (size_t)(world + 1) * k * 8;
  if (need > h_cap) {
    if (h_vals) (void)hipHostFree(h_vals);
    h_cap = std::max<size_t>(need, 4096);
    IMPG_HIP(hipHostMalloc((void **)&h_vals, h_cap, hipHostMallocDefault));
  }
  d_vals.reserve(std::max<size_t>(need, 4096));
  hipStream_t s = cs;
  memcpy(h_vals, mine, k * 8);
  uint64_t *d_mine = d_vals.as<uint64_t>(), *d_all = d_mine + k;
  {
    Turn turn(order.get(), lane);
    IMPG_HIP(hipMemcpyAsync(d_mine, h_vals, k * 8, hipMemcpyHostToDevice, s));
    nccl_check(rccl().AllGather(d_mine, d_all, k, NCCL_UINT64, comm, s), "ncclAllGather");
    IMPG_HIP(hipMemcpyAsync(h_vals + k, d_all, (size_t)world * k * 8, hipMemcpyDeviceToHost, s));
  }
  IMPG_HIP(hipStreamSynchronize(s));
  memcpy(all, h_vals + k, (size_t)world * k * 8);
}
void RcclComm::alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                         const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) {
  if (dead) throw Error{IMPG_E_HIP, "the RCCL communicator was aborted after an earlier failure"};
  IMPG_HIP(hipSetDevice(device));
  // a rank's block for itself never enters RCCL: a device-to-device copy on the same stream (1/world of the volume; a
  // send/receive pair to oneself runs through the proxy's staging kernels at a fraction of the copy's rate)
  if (send_bytes[rank] != recv_bytes[rank]) throw Error{IMPG_E_INVALID, "alltoallv: block sizes disagree"};
  if (send_bytes[rank])
    IMPG_HIP(hipMemcpyAsync((char *)d_recv + recv_off[rank], (const char *)d_send + send_off[rank], send_bytes[rank], hipMemcpyDeviceToDevice, s));
  if (world == 1) { IMPG_HIP(hipStreamSynchronize(s)); return; }
  Turn turn(order.get(), lane);
  // one message per peer and round, <= 256 MiB each: bounded staging inside RCCL, and far below the size at
  // which a single all-to-all message was seen corrupted on this stack in round 1 (> 1 GiB)
  constexpr uint64_t ROUND = 256ull << 20;
  uint64_t most = 0;
  for (int p = 0; p < world; p++) if (p != rank) most = std::max(most, std::max(send_bytes[p], recv_bytes[p]));
  // sends and receives are matched pair by pair: both ends of a pair derive the same number of messages from the
  // same byte count, so ranks may run different numbers of rounds
  for (uint64_t done = 0; done < most; done += ROUND) {
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (int k = 1; k < world; k++) {
      const int p = (rank + k) % world;
      if (send_bytes[p] > done) {
        const uint64_t n = std::min(ROUND, send_bytes[p] - done);
        nccl_check(rccl().Send((const char *)d_send + send_off[p] + done, n, NCCL_INT8, p, comm, s), "ncclSend");
      }
      const int q = (rank - k + world) % world;
      if (recv_bytes[q] > done) {
        const uint64_t n = std::min(ROUND, recv_bytes[q] - done);
        nccl_check(rccl().Recv((char *)d_recv + recv_off[q] + done, n, NCCL_INT8, q, comm, s), "ncclRecv");
      }
    }
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  }
  turn.done();  // enqueued: the next lane may issue while this exchange runs
  IMPG_HIP(hipStreamSynchronize(s));
}
void RcclComm::barrier() {
  uint64_t one = 1;
  std::vector<uint64_t> all(world);
  allgather_u64(&one, 1, all.data());
}

// ---- host-provided transport ----------------------------------------------------------------
HostComm::HostComm(const impg_gpu_host_transport_t &tr, int rank_, int world_) : t(tr) {
  rank = rank_;
  world = world_;
  if (!t.allgather_u64 || !t.alltoallv) throw Error{IMPG_E_INVALID, "host transport needs both callbacks"};
}
HostComm::~HostComm() {
  if (h_send) (void)hipHostFree(h_send);
  if (h_recv) (void)hipHostFree(h_recv);
}
void HostComm::allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) {
  if (t.allgather_u64(t.ctx, mine, k, all) != 0) throw Error{IMPG_E_IO, "host transport: allgather failed"};
}
void HostComm::alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                         const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) {
  // pack the blocks back to back in pinned memory (the callback sees dense buffers and their offsets)
  std::vector<uint64_t> so(world), ro(world);
  uint64_t st = 0, rt = 0;
  for (int p = 0; p < world; p++) { so[p] = st; st += send_bytes[p]; ro[p] = rt; rt += recv_bytes[p]; }
  auto grow = [](char *&p, size_t &cap, size_t need) {
    if (need <= cap && p) return;  // (never null: the callback gets real buffers even for an empty exchange)
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = std::max<size_t>(need + need / 4, 1 << 16);
    IMPG_HIP(hipHostMalloc((void **)&p, cap, hipHostMallocDefault));
  };
  grow(h_send, send_cap, st);
  grow(h_recv, recv_cap, rt);
  for (int p = 0; p < world; p++)
    if (send_bytes[p])
      IMPG_HIP(hipMemcpyAsync(h_send + so[p], (const char *)d_send + send_off[p], send_bytes[p], hipMemcpyDeviceToHost, s));
  IMPG_HIP(hipStreamSynchronize(s));
  if (t.alltoallv(t.ctx, h_send, so.data(), send_bytes, h_recv, ro.data(), recv_bytes) != 0)
    throw Error{IMPG_E_IO, "host transport: alltoallv failed"};
  for (int p = 0; p < world; p++)
    if (recv_bytes[p])
      IMPG_HIP(hipMemcpyAsync((char *)d_recv + recv_off[p], h_recv + ro[p], recv_bytes[p], hipMemcpyHostToDevice, s));
  IMPG_HIP(hipStreamSynchronize(s));
}
void HostComm::barrier() {
  uint64_t one = 1;
  std::vector<uint64_t> all(world);
  allgather_u64(&one, 1, all.data());
}

}  // namespace impg
