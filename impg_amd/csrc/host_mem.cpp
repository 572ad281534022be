// Pinned host blocks for result arrays, recycled through a process-wide pool.
// A full-results call on BASELINE config 3 returns 2.1e8 rows = 5 GB: the rows are built on the device
// (rows_device.hip) and cross PCIe in one copy, which wants a pinned destination -- and pinning 5 GB of fresh
// memory costs more than the copy.  A block goes back to the pool when its result is freed
// (impg_gpu_results_free) and serves the next call; IMPG_PINNED_POOL_BYTES bounds what the pool keeps
// (default 6 GiB per process -- room for one config-3 result; 0 = keep nothing), impg_gpu_host_pool_trim gives
// blocks back on request (a job of one process per GPU holds the pool once per rank).
#include <mutex>
#include <thread>

#include "impg_internal.hpp"

namespace impg {
namespace {
struct PinnedPool {
  struct Blk { void *p; size_t cap; };
  std::mutex m;
  std::vector<Blk> free_;
  size_t held = 0, max_held;
  PinnedPool() {
    const char *e = getenv("IMPG_PINNED_POOL_BYTES");
    max_held = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)6 << 30;
  }
  // (blocks still on the list at process exit are left to the runtime's own teardown: hipHostFree from a static
  // destructor can run after the HIP runtime has shut down)
};
PinnedPool &pool() {
  static PinnedPool *p = new PinnedPool();
  return *p;
}
}  // namespace

void *pinned_take(size_t bytes, size_t &cap_out) {
  PinnedPool &P = pool();
  {
    std::lock_guard<std::mutex> lk(P.m);
    size_t best = P.free_.size();
    const size_t limit = std::max<size_t>(2 * bytes, 4u << 20);  // not wastefully large for the request
    for (size_t i = 0; i < P.free_.size(); i++)
      if (P.free_[i].cap >= bytes && P.free_[i].cap <= limit && (best == P.free_.size() || P.free_[i].cap < P.free_[best].cap)) best = i;
    if (best != P.free_.size()) {
      const PinnedPool::Blk b = P.free_[best];
      P.free_[best] = P.free_.back();
      P.free_.pop_back();
      P.held -= b.cap;
      cap_out = b.cap;
      return b.p;
    }
  }
  const size_t want = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, want, hipHostMallocPortable);
  if (e != hipSuccess) {  // give the pool's blocks back to the system and try once more
    (void)hipGetLastError();
    std::vector<PinnedPool::Blk> drop;
    {
      std::lock_guard<std::mutex> lk(P.m);
      drop.swap(P.free_);
      P.held = 0;
    }
    for (auto &b : drop) (void)hipHostFree(b.p);
    e = hipHostMalloc(&p, want, hipHostMallocPortable);
    if (e != hipSuccess) throw Error{IMPG_E_OOM, std::string("hipHostMalloc of a result array: ") + hipGetErrorString(e)};
  }
  cap_out = want;
  return p;
}

void pinned_give(void *p, size_t cap) {
  PinnedPool &P = pool();
  std::vector<PinnedPool::Blk> drop;
  {
    std::lock_guard<std::mutex> lk(P.m);
    P.free_.push_back({p, cap});
    P.held += cap;
    while (P.held > P.max_held && !P.free_.empty()) {  // the largest blocks go first
      size_t big = 0;
      for (size_t i = 1; i < P.free_.size(); i++) if (P.free_[i].cap > P.free_[big].cap) big = i;
      drop.push_back(P.free_[big]);
      P.held -= P.free_[big].cap;
      P.free_[big] = P.free_.back();
      P.free_.pop_back();
    }
  }
  for (auto &b : drop) (void)hipHostFree(b.p);
}

size_t pinned_trim(size_t keep_bytes) {
  PinnedPool &P = pool();
  std::vector<PinnedPool::Blk> drop;
  size_t freed = 0;
  {
    std::lock_guard<std::mutex> lk(P.m);
    while (P.held > keep_bytes && !P.free_.empty()) {  // the largest blocks go first
      size_t big = 0;
      for (size_t i = 1; i < P.free_.size(); i++) if (P.free_[i].cap > P.free_[big].cap) big = i;
      drop.push_back(P.free_[big]);
      P.held -= P.free_[big].cap;
      freed += P.free_[big].cap;
      P.free_[big] = P.free_.back();
      P.free_.pop_back();
    }
  }
  for (auto &b : drop) (void)hipHostFree(b.p);
  return freed;
}

void parallel_memcpy(void *dst, const void *src, size_t bytes) {
  const size_t T = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
  if (bytes < ((size_t)64 << 20) || T == 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t step = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
  for (size_t t = 0; t < T; t++) {
    const size_t a = std::min(bytes, t * step), b = std::min(bytes, a + step);
    if (a < b) th.emplace_back([=] { memcpy((char *)dst + a, (const char *)src + a, b - a); });
  }
  for (auto &x : th) x.join();
}

}  // namespace impg
