// Rank-to-rank transport of the sharded engine (sharded.cpp): what one hop of the
// transitive closure needs from the fabric -- an all-gather of a few host words
// (bucket sizes, liveness) and an all-to-all-v of device buffers (frontier
// records out, hit records back).  Four implementations behind one interface:
//   SelfComm   one rank: copies
//   LocalComm  the ranks are threads of this process, one per GPU (impg_gpu_index_create_multi):
//              every rank PULLS its blocks from its peers' send buffers with hipMemcpyPeerAsync -- direct
//              xGMI reads, no staging, no library in between
//   RcclComm   one process per GPU (the torch.distributed.run layout): ncclSend / ncclRecv groups on the
//              engine's stream; librccl is opened at run time (dlopen), so the library loads without it
//   HostComm   the host brings the transport (MPI, gloo, ...) as two callbacks over host memory; blocks are
//              staged through pinned buffers.  Also what the multi-rank tests on a one-GPU box use.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "impg_internal.hpp"

namespace impg {

struct Comm {
  int rank = 0, world = 1;
  virtual ~Comm() {}
  // all[r * k + i] = value i of rank r
  virtual void allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) = 0;
  // Block d of d_send (send_off[d], send_bytes[d]) goes to rank d; block s of d_recv comes from rank s.
  // Device memory both sides; ordered after the work already queued on `s`; complete on return.
  virtual void alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                         const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) = 0;
  virtual void barrier() = 0;
  // This rank failed INSIDE the transport (a failure outside it is announced through the next all-gather instead,
  // sharded.cpp "failure agreement"): release peers that wait for it, where the transport can.  LocalComm poisons
  // its fabric, RcclComm calls ncclCommAbort (the communicator is dead afterwards); a host transport must bring its
  // own timeout -- the callbacks are the host's.
  virtual void abort() {}
  virtual void reset() {}  // before a new batch, with no rank inside the transport: forget an earlier abort
  // The lanes of one rank issue their collectives in one agreed order (RcclComm: see IssueOrder); a batch tells the
  // transport which lanes take part, and each lane when it has issued its last collective.
  virtual void batch_begin(bool /*takes_part*/) {}
  virtual void batch_end() {}
  virtual const char *kind() const = 0;
};

struct SelfComm : Comm {
  void allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) override;
  void alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                 const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) override;
  void barrier() override {}
  const char *kind() const override { return "self"; }
};

// ---- threads of one process ----------------------------------------------------
struct LocalFabric {  // shared by the `world` LocalComm objects of one lane
  int world;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;  // a rank failed: everybody waiting is released with an error
  struct Slot {
    const uint64_t *vals = nullptr;
    const void *send = nullptr;
    const uint64_t *send_off = nullptr, *send_bytes = nullptr;
    int device = 0;
  };
  std::vector<Slot> slots;
  explicit LocalFabric(int w) : world(w), slots(w) {}
  void wait_all();  // sense-reversing barrier; throws if the fabric broke
  void poison();
  void reset() {
    std::lock_guard<std::mutex> lk(m);
    broken = false;
    arrived = 0;
  }
};
struct LocalComm : Comm {
  std::shared_ptr<LocalFabric> fab;
  int device;
  LocalComm(std::shared_ptr<LocalFabric> f, int rank_, int device_) : fab(std::move(f)), device(device_) {
    rank = rank_;
    world = fab->world;
  }
  void allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) override;
  void alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                 const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) override;
  void barrier() override { fab->wait_all(); }
  void abort() override { fab->poison(); }
  void reset() override { fab->reset(); }
  const char *kind() const override { return "local"; }
};

// ---- RCCL ------------------------------------------------------------------------
constexpr size_t RCCL_UNIQUE_ID_BYTES = 128;  // ncclUniqueId
void rccl_unique_id(uint8_t *id128);          // rank 0 makes it; the host carries it to the other ranks
// Every lane of a rank drives its own RCCL communicator from its own host thread.  Communicators that share a device
// must see their operations ISSUED in the same order on every rank or they can deadlock (each waits for resources
// the other holds on some rank).  The lanes' collective sequences are identical on all ranks (same chunks per lane,
// collective hops), so a round-robin over the lanes by operation index -- lanes dropping out when their batch work is
// done -- is an order every rank derives by itself.  A lane holds its turn only while it enqueues (not while it
// waits for completion), and never while it holds the GPU turn of sharded.cpp.
struct IssueOrder {
  std::mutex m;
  std::condition_variable cv;
  std::vector<char> active;  // lanes taking part in the batch in flight that have collectives left
  int turn = 0;
  bool in_batch = false;
  explicit IssueOrder(int lanes) : active((size_t)lanes, 0) {}
  void begin(int lane, bool takes_part);
  void end(int lane);
  void acquire(int lane);
  void release(int lane);
  void pass_from(int lane);  // (m held) the turn goes to the next active lane after `lane`
};
struct RcclComm : Comm {
  void *comm = nullptr;  // ncclComm_t
  int device;
  std::shared_ptr<IssueOrder> order;  // shared by the lanes of this rank (null: a single lane)
  int lane = 0;
  std::atomic<bool> dead{false};  // aborted after a failure inside the transport (set by whichever lane thread notices; read by the owner)
  DevBuf d_vals;
  uint64_t *h_vals = nullptr;  // pinned
  size_t h_cap = 0;
  hipStream_t cs = nullptr;  // the small all-gathers run on their own stream
  RcclComm(const uint8_t *id128, int rank_, int world_, int device_);
  ~RcclComm() override;
  void allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) override;
  void alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                 const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) override;
  void barrier() override;
  void abort() override;
  void batch_begin(bool takes_part) override { if (order) order->begin(lane, takes_part); }
  void batch_end() override { if (order) order->end(lane); }
  const char *kind() const override { return "rccl"; }
};

// ---- host-provided transport --------------------------------------------------------
struct HostComm : Comm {
  impg_gpu_host_transport_t t;
  char *h_send = nullptr, *h_recv = nullptr;  // pinned staging
  size_t send_cap = 0, recv_cap = 0;
  HostComm(const impg_gpu_host_transport_t &tr, int rank_, int world_);
  ~HostComm() override;
  void allgather_u64(const uint64_t *mine, size_t k, uint64_t *all) override;
  void alltoallv(const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                 const uint64_t *recv_off, const uint64_t *recv_bytes, hipStream_t s) override;
  void barrier() override;
  const char *kind() const override { return "host"; }
};

}  // namespace impg

// the opaque handle of include/impg_gpu.h: one communicator per lane (chunks in flight)
struct impg_gpu_comm {
  int rank = 0, world = 1, device = 0;
  std::vector<std::unique_ptr<impg::Comm>> lanes;
};
