// The index sharded over GPUs: entries live with the rank that owns their target
// sequence (targets bin-packed by entry count), every query lives with its HOME rank (the rank that
// submitted it: visited sets, DFS stacks, result order), and each hop of the walk sends the frontier
// records to the owners of their targets and brings the hits home:
//
//   home   route: stable partition of the frontier by owner (one short radix sort); the record's home
//          index rides along in place of qidx
//          all-gather of the bucket sizes (+ one liveness word per rank: termination costs no extra round)
//          all-to-all-v of 16-byte frontier records
//   owner  lookup + projection of what arrived, on its shard, in slices under the pair budget
//          (Engine::expand, the same kernels as on one GPU); slots packed into 16- or 32-byte records
//          addressed to the record's home
//          all-gather of the per-home counts, all-to-all-v of the hit records
//   home   blocks arrive per owner, each ascending in the home index; a frontier record has exactly one
//          owner, so a counting pass puts them back in frontier order x visit order; unpacked into the
//          slot arrays a local expansion would have filled -- from there the single-GPU code runs unchanged
//          (visited update, DFS / MultiImpg worklists, masks, subset filter, result assembly)
//
// The transport is a Comm (comm.hpp): threads + peer copies inside one process, RCCL between processes, or
// host callbacks.  Several chunks of a batch are in flight at once, one per LANE (engine + communicator +
// host thread): while one lane waits for an exchange the others keep the GPU busy.
// The north-star text says "allgatherv of the frontier"; all-to-all-v moves 1/world of that volume over the
// point-to-point xGMI links, and is what is built (the all-gather carries the counts).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <exception>
#include <numeric>
#include <thread>

#include "comm.hpp"
#include "engine.hpp"

namespace impg {

// ---- shard map ------------------------------------------------------------------------------
void count_entries_per_target(const impg_gpu_record_t *records, size_t n_records, uint32_t n_seq, bool bidirectional,
                              std::vector<uint64_t> &cnt) {
  cnt.assign(n_seq, 0);
  for (size_t i = 0; i < n_records; i++) {
    const auto &r = records[i];
    if (r.query_id >= n_seq || r.target_id >= n_seq) throw Error{IMPG_E_INVALID, "record sequence id out of range"};
    cnt[r.target_id]++;
    if (bidirectional && r.query_id != r.target_id) cnt[r.query_id]++;  // the reversed entry (impg.rs:1584)
  }
}
void shard_assign(const uint64_t *c, uint32_t n_seq, uint32_t n_shards, uint32_t *owner) {
  if (n_shards == 0) throw Error{IMPG_E_INVALID, "no shards"};
  std::vector<uint32_t> order(n_seq);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c[a] > c[b]; });
  std::vector<uint64_t> load(n_shards, 0);
  for (uint32_t t : order) {
    if (c[t] == 0) { owner[t] = t % n_shards; continue; }  // no entries anywhere: any rank answers "no hits"
    uint32_t best = 0;
    for (uint32_t k = 1; k < n_shards; k++)
      if (load[k] < load[best]) best = k;
    owner[t] = best;
    load[best] += c[t];
  }
}

// ---- one hop ---------------------------------------------------------------------------------
struct ShardedExpander : Expander {
  Comm *comm = nullptr;
  const uint32_t *d_owner = nullptr;
  uint32_t n_seq = 0;
  DevBuf send_fr, recv_fr, hits_out, hits_in, ops_out, ops_in, mslot, iota, route_hist, d_bounds;
  ShardedExpander() {  // (the lane's own buffers: never swapped with an engine's, DevBuf::swap)
    for (DevBuf *b : {&send_fr, &recv_fr, &hits_out, &hits_in, &ops_out, &ops_in, &mslot, &iota, &route_hist, &d_bounds}) b->lane_owned = true;
  }
  LevelBufs owner_L;
  // pinned words every readback / small upload of a hop goes through (a copy from or to pageable memory stages and
  // blocks): [0, W] slot offsets at block boundaries, [W+1, 2W+1] slice-pool offsets there, then the route histogram
  // (W x u64), the reorder error flag, and BOUND_SLOTS x (W+1) block bounds on their way to the device
  uint32_t *h_vals = nullptr;
  size_t h_cap = 0;
  static constexpr uint32_t BOUND_SLOTS = 8;
  uint32_t bound_slot = 0;
  void pinned_words(uint32_t W) {
    const size_t need = ((size_t)(W + 1) * (2 + BOUND_SLOTS) + 2 * (size_t)W + 4) * 4;
    if (need <= h_cap) return;
    if (h_vals) (void)hipHostFree(h_vals);
    h_vals = nullptr;
    h_cap = std::max<size_t>(need, 4096);
    IMPG_HIP(hipHostMalloc((void **)&h_vals, h_cap, hipHostMallocDefault));
  }
  unsigned long long *h_hist(uint32_t W) { return reinterpret_cast<unsigned long long *>(h_vals + 2 * (W + 1)); }
  uint32_t *h_err(uint32_t W) { return h_vals + 2 * (W + 1) + 2 * W; }
  uint32_t *h_bounds(uint32_t W, uint32_t slot) { return h_vals + 2 * (W + 1) + 2 * W + 4 + (size_t)slot * (W + 1); }
  double exchange_s = 0;  // wall time inside the transport, accumulated
  uint64_t bytes_out = 0;
  // Where a hop's wall time goes on this lane, by hop number within its batch (impg_gpu_index_hop_profile; accumulates
  // until read with reset): host wall clocks around the hop's stages, so a stage that ends in a collective includes the
  // wait for the slowest rank.
  static constexpr int PROF_HOPS = 8, PROF_FIELDS = 12;
  enum { PF_HOPS, PF_ROUTE, PF_GATHER1, PF_A2A1, PF_EXPAND, PF_GATHER2, PF_A2A2, PF_HOME, PF_BYTES_FR, PF_BYTES_HITS, PF_RECS_IN, PF_HITS_HOME };
  double prof[PROF_HOPS][PROF_FIELDS] = {};
  // The lanes of a rank take turns on the GPU: two chunks' kernels side by side evict each other's entries and
  // tiles from L2 (the window-order locality every kernel here leans on) and ran 3.5x slower than back to back.
  // A lane gives the GPU up exactly while it sits in the transport, which is the overlap lanes exist for.
  std::mutex *gpu_turn = nullptr;
  // Failure agreement.  A rank that fails OUTSIDE the transport must not leave its peers waiting in the next
  // collective: the all-gathers of a hop carry a status word, a failure on the owner side of a hop is reported
  // in the hop's second all-gather, one anywhere else in the lane's next all-gather (announce_failure, from
  // run_lanes), and every rank leaves the batch with an error at the same point of the protocol (`agreed`:
  // nothing more is to be said on this lane).  A failure INSIDE the transport (`in_transport`) cannot be announced
  // through it: the lane aborts its communicator instead (Comm::abort).
  bool agreed = false, in_transport = false;
  uint32_t hop_no = 0, fail_owner_hop = 0, fail_home_hop = 0;  // failure injection (tests): hop counter of the batch, hops at which to throw
  static constexpr uint64_t ST_ALIVE = 1, ST_FAILED = 2;

  ~ShardedExpander() override {
    if (h_vals) (void)hipHostFree(h_vals);
  }
  // The lane's closing word of a batch (failed = false) or its announcement of a failure outside the transport:
  // an all-gather of the shape every hop starts with, so that peers take it wherever they are -- at their next hop
  // or at their own closing word.
  void closing_word(bool failed) {
    const size_t K = (size_t)comm->world + 1;
    std::vector<uint64_t> mine(K, 0), mat(K * (size_t)comm->world);
    mine[comm->world] = failed ? ST_FAILED : 0;
    comm->allgather_u64(mine.data(), K, mat.data());
    if (failed) return;
    for (int r = 0; r < comm->world; r++)
      if (mat[(size_t)r * K + comm->world] & ST_FAILED) { agreed = true; throw Error{IMPG_E_HIP, "a peer rank failed"}; }
  }

  void route(Engine &E, const FrontierRec *fr, uint32_t n, uint64_t *counts) {
    const uint32_t W = (uint32_t)comm->world;
    hipStream_t s = E.stream;
    const size_t nb = std::max<size_t>((size_t)n * 4, 256);
    E.lo_key.reserve(nb); E.lo_key2.reserve(nb); E.lo_idx.reserve(nb); E.lo_perm.reserve(nb);
    route_hist.reserve(std::max<size_t>((size_t)W * 8, 256));
    IMPG_HIP(hipMemsetAsync(route_hist.p, 0, (size_t)W * 8, s));
    launch_route_keys(fr, n, W, d_owner, n_seq, E.lo_key.as<uint32_t>(), E.lo_idx.as<uint32_t>(),
                      route_hist.as<unsigned long long>(), s);
    if (W > 1) {
      unsigned bits = 1;
      while ((1u << bits) < W) bits++;
      E.sort_tmp.reserve(order_sort_scratch_bytes(n));
      launch_order_sort(E.lo_key.as<uint32_t>(), E.lo_key2.as<uint32_t>(), E.lo_perm.as<uint32_t>(), E.lo_idx.as<uint32_t>(), n, bits,
                        E.sort_tmp.p, s);  // stable: frontier order within an owner
      launch_route_gather(fr, E.lo_perm.as<uint32_t>(), n, send_fr.as<FrontierRec>(), s);
    } else {
      launch_route_gather(fr, E.lo_idx.as<uint32_t>(), n, send_fr.as<FrontierRec>(), s);
    }
    pinned_words(W);
    unsigned long long *h = h_hist(W);
    IMPG_HIP(hipMemcpyAsync(h, route_hist.p, (size_t)W * 8, hipMemcpyDeviceToHost, s));
    IMPG_HIP(hipStreamSynchronize(s));
    for (uint32_t k = 0; k < W; k++) counts[k] = h[k];
  }

  // An allocation between a hop's all-gather and the all-to-all-v it announced: the peers are already on their way into
  // that exchange, so a failure here can be neither agreed on nor announced through the lane -- it counts as a failure
  // inside the transport (the lane's communicator is aborted, run_lanes), like a throw from the exchange itself.
  void reserve_mid_exchange(DevBuf &b, size_t bytes) {
    try {
      b.reserve(bytes);
    } catch (...) {
      in_transport = true;
      throw;
    }
  }

  template <class F> void timed_comm(F f) {
    if (gpu_turn) gpu_turn->unlock();
    const auto t0 = std::chrono::steady_clock::now();
    try {
      f();
    } catch (...) {
      in_transport = true;
      if (gpu_turn) gpu_turn->lock();
      throw;
    }
    exchange_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (gpu_turn) gpu_turn->lock();
  }

  HopResult hop(Engine &E, const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                impg_gpu_stats_t *st, bool need_hits, bool need_rows, bool alive) override {
    const int W = comm->world, me = comm->rank;
    hipStream_t s = E.stream;
    const size_t K = (size_t)W + 1;
    std::vector<uint64_t> mine(K, 0), mat(K * W);
    L.n_pairs = 0;
    hop_no++;
    double *pf = prof[std::min<uint32_t>(hop_no - 1u, PROF_HOPS - 1)];
    auto clk = std::chrono::steady_clock::now();
    auto lap = [&](int field) {
      const auto now = std::chrono::steady_clock::now();
      pf[field] += std::chrono::duration<double>(now - clk).count();
      clk = now;
    };
    pf[PF_HOPS] += 1;
    // ---- home: the frontier bucketed by owner; sizes and liveness to everybody
    send_fr.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
    if (n_fr) route(E, fr, n_fr, mine.data());
    mine[W] = alive ? ST_ALIVE : 0;
    lap(PF_ROUTE);
    timed_comm([&] { comm->allgather_u64(mine.data(), K, mat.data()); });
    lap(PF_GATHER1);
    bool any = false;
    for (int r = 0; r < W; r++) {
      any = any || (mat[(size_t)r * K + W] & ST_ALIVE) != 0;
      if (mat[(size_t)r * K + W] & ST_FAILED) { agreed = true; throw Error{IMPG_E_HIP, "a peer rank failed"}; }
    }
    if (!any) return HopResult{0, true};
    // (limits are checked for every rank from the gathered sizes, so that all ranks fail together)
    for (int d = 0; d < W; d++) {
      uint64_t to_d = 0;
      for (int r = 0; r < W; r++) to_d += mat[(size_t)r * K + d];
      if (to_d >= 0xFFFFFFF0ull) { agreed = true; throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 frontier records for one shard in one hop"}; }
    }
    // ---- frontier records to the owners of their targets
    std::vector<uint64_t> so(W), sb(W), ro(W), rb(W), rstart(W + 1);
    uint64_t acc = 0, n_recv = 0;
    for (int d = 0; d < W; d++) { so[d] = acc * sizeof(FrontierRec); sb[d] = mine[d] * sizeof(FrontierRec); acc += mine[d]; }
    for (int r = 0; r < W; r++) {
      const uint64_t c = mat[(size_t)r * K + me];
      rstart[r] = n_recv; ro[r] = n_recv * sizeof(FrontierRec); rb[r] = c * sizeof(FrontierRec);
      n_recv += c;
    }
    rstart[W] = n_recv;
    reserve_mid_exchange(recv_fr, std::max<size_t>(n_recv * sizeof(FrontierRec), 256));
    timed_comm([&] { comm->alltoallv(send_fr.p, so.data(), sb.data(), recv_fr.p, ro.data(), rb.data(), s); });
    bytes_out += acc * sizeof(FrontierRec);
    pf[PF_BYTES_FR] += (double)(acc * sizeof(FrontierRec));
    pf[PF_RECS_IN] += (double)n_recv;
    lap(PF_A2A1);
    // ---- owner: expand what arrived, in slices under the pair budget
    const uint32_t words = (need_rows || E.multi) ? 8u : 4u;
    // store_cigar: the owner materialises every hit's CIGAR slice (Engine::expand) and the ops follow the hit records
    // home in a second exchange, block by block in the same order; a hit record carries its op count in word 7
    const bool ship_ops = E.store_cigar && need_hits;
    const size_t K2 = 2 * (size_t)W + 1;
    std::vector<uint64_t> back(K2, 0);  // hits per home rank, ops per home rank, this rank's status word
    struct Piece { std::unique_ptr<DevBuf> buf, ops; uint64_t n, n_ops; };
    std::vector<Piece> pieces;
    uint64_t total_pairs = 0, total_hits = 0, total_ops = 0;
    const bool saved_split = E.split_ok;
    const void *out_ptr = nullptr, *ops_ptr = nullptr;
    bool any_by_place = false;
    std::exception_ptr deferred;  // an owner-side failure waits for the all-gather below, where every rank learns of it
    try {
    if (fail_owner_hop && hop_no == fail_owner_hop) throw Error{IMPG_E_INVALID, "injected failure (owner side)"};
    pinned_words((uint32_t)W);
    // A counting hop (nobody at home reads rows) lets the owner lay its slots out in its own lookup order, home rank
    // by home rank: the cheaper lookup and projection of Engine::free_slot_order.  Home puts the runs back in
    // frontier order whatever order they come in (reorder_runs), so nothing else changes.
    const bool owner_order = E.free_slot_order && !need_rows && !E.multi;
    if (owner_order) d_bounds.reserve(std::max<size_t>(((size_t)W + 1) * 4, 256));
    uint64_t a = 0, step = std::max<uint64_t>(n_recv, 1);
    while (a < n_recv) {
      const uint64_t m = std::min(step, n_recv - a);
      E.split_ok = m > 1;
      uint64_t P = 0;
      const FrontierRec *sub = recv_fr.as<FrontierRec>() + a;
      Engine::RecordBlocks blocks{nullptr, (uint32_t)W};
      if (owner_order) {  // the slice's records by home rank: block r = [hb[r], hb[r+1])
        // (through a pinned slot, stream-ordered; a hop rarely has more than one slice: when the slots run out, wait)
        if (bound_slot == BOUND_SLOTS) { IMPG_HIP(hipStreamSynchronize(s)); bound_slot = 0; }
        uint32_t *hbp = h_bounds((uint32_t)W, bound_slot++);
        for (int r = 0; r <= W; r++) hbp[r] = (uint32_t)(std::min(std::max(rstart[r], a), a + m) - a);
        IMPG_HIP(hipMemcpyAsync(d_bounds.p, hbp, ((size_t)W + 1) * 4, hipMemcpyHostToDevice, s));
        blocks.d_bounds = d_bounds.as<uint32_t>();
      }
      try {
        // (a hop nobody at home reads hits of -- the final level of a counting run -- takes its pairs from the count
        // pass's windows: no emit pass, Engine::fuse_final)
        E.fuse_final = E.fuse_allowed && !need_hits;
        E.fuse_need_ranges = false;
        P = E.expand(v, sub, (uint32_t)m, transitive, owner_L, st, /*raw=*/true, owner_order ? &blocks : nullptr);
        E.fuse_final = false;
        any_by_place = any_by_place || E.last_by_place;
      } catch (const SplitBatch &) {
        E.fuse_final = false;
        step = std::max<uint64_t>(1, m / 2);
        continue;
      }
      total_pairs += P;
      if (need_hits && P) {
        // slots per home rank: the records of one home are one contiguous run of `sub`
        std::vector<std::pair<int, std::pair<uint64_t, uint64_t>>> runs;  // (home, [lo, hi) relative to a)
        size_t nread = 0;
        for (int r = 0; r < W; r++) {
          const uint64_t lo = std::max(a, rstart[r]), hi = std::min(a + m, rstart[r + 1]);
          if (lo >= hi) continue;
          runs.push_back({r, {lo - a, hi - a}});
          IMPG_HIP(hipMemcpyAsync(h_vals + nread, E.pair_off.as<uint32_t>() + (lo - a), 4, hipMemcpyDeviceToHost, s));
          nread++;
        }
        Piece pc{std::make_unique<DevBuf>(), std::make_unique<DevBuf>(), P, 0};
        pc.buf->reserve(std::max<size_t>(P * words * 4, 256));
        HitArrays h{owner_L.qid.as<uint32_t>(), owner_L.coords.as<int4>()};
        launch_hits_pack(sub, owner_L.pair_range.as<uint32_t>(), (uint32_t)P, h, E.multi ? E.pair_entry.as<uint32_t>() : nullptr,
                         E.multi ? v.mrank : nullptr, words, pc.buf->p, s, ship_ops ? E.cnt.as<uint32_t>() : nullptr);
        IMPG_HIP(hipStreamSynchronize(s));
        for (size_t k = 0; k < runs.size(); k++) {
          const uint64_t first = h_vals[k], end = k + 1 < runs.size() ? h_vals[k + 1] : P;
          back[runs[k].first] += end - first;
        }
        if (ship_ops) {  // the ops of one home's hits are one stretch of the slice pool (slot order): its ends from slice_pos
          uint32_t *h_pos = h_vals + (W + 1);
          for (size_t k = 0; k < runs.size(); k++) {
            // (a run whose records have no slot at all starts at slot P: past the end of slice_pos, its ops start at the pool's end)
            if (h_vals[k] >= P) h_pos[k] = (uint32_t)owner_L.slice_total;
            else IMPG_HIP(hipMemcpyAsync(h_pos + k, owner_L.slice_pos.as<uint32_t>() + h_vals[k], 4, hipMemcpyDeviceToHost, s));
          }
          IMPG_HIP(hipStreamSynchronize(s));
          for (size_t k = 0; k < runs.size(); k++) {
            const uint64_t first = h_pos[k], end = k + 1 < runs.size() ? h_pos[k + 1] : owner_L.slice_total;
            back[(size_t)W + runs[k].first] += end - first;
          }
          pc.n_ops = owner_L.slice_total;
          pc.ops->swap(owner_L.slice_pool);  // (the next slice materialises into a fresh buffer)
          total_ops += pc.n_ops;
        }
        total_hits += P;
        pieces.push_back(std::move(pc));
      }
      a += m;
    }
    E.split_ok = saved_split;
    if (need_hits) {
      if (total_hits >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 hits leave one shard in one hop"};
      if (pieces.size() == 1) {
        out_ptr = pieces[0].buf->p;
      } else {
        hits_out.reserve(std::max<size_t>(total_hits * words * 4, 256));
        uint64_t pos = 0;
        for (auto &pc : pieces) {
          IMPG_HIP(hipMemcpyAsync((char *)hits_out.p + pos * words * 4, pc.buf->p, pc.n * words * 4, hipMemcpyDeviceToDevice, s));
          pos += pc.n;
        }
        out_ptr = hits_out.p;
      }
      if (ship_ops) {
        if (total_ops >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 CIGAR ops leave one shard in one hop: use smaller chunks (chunk_ranges)"};
        if (pieces.size() == 1) ops_ptr = pieces[0].ops->p;
        else {
          ops_out.reserve(std::max<size_t>(total_ops * 4, 256));
          uint64_t pos = 0;
          for (auto &pc : pieces) {
            if (pc.n_ops) IMPG_HIP(hipMemcpyAsync((char *)ops_out.p + pos * 4, pc.ops->p, pc.n_ops * 4, hipMemcpyDeviceToDevice, s));
            pos += pc.n_ops;
          }
          ops_ptr = ops_out.p;
        }
      }
    }
    } catch (...) {
      E.split_ok = saved_split;
      E.fuse_final = false;
      if (!need_hits) throw;  // no all-gather follows in this hop: run_lanes announces it in the lane's next one
      deferred = std::current_exception();
      std::fill(back.begin(), back.end(), 0);
      back[K2 - 1] = ST_FAILED;
    }
    if (!need_hits) {
      IMPG_HIP(hipStreamSynchronize(s));  // (the profile's clock: the owner's kernels, not just their launches)
      lap(PF_EXPAND);
      return HopResult{total_pairs, false};
    }
    lap(PF_EXPAND);
    // ---- hits go home
    std::vector<uint64_t> mat2(K2 * W);
    timed_comm([&] { comm->allgather_u64(back.data(), K2, mat2.data()); });
    lap(PF_GATHER2);
    for (int o = 0; o < W; o++)
      if (mat2[(size_t)o * K2 + (K2 - 1)] & ST_FAILED) {
        agreed = true;
        if (deferred) std::rethrow_exception(deferred);
        throw Error{IMPG_E_HIP, "a peer rank failed"};
      }
    uint64_t n_home = 0;
    acc = 0;
    const uint64_t rec = (uint64_t)words * 4;
    for (int d = 0; d < W; d++) { so[d] = acc * rec; sb[d] = back[d] * rec; acc += back[d]; }
    for (int o = 0; o < W; o++) {
      const uint64_t c = mat2[(size_t)o * K2 + me];
      ro[o] = n_home * rec; rb[o] = c * rec;
      n_home += c;
    }
    for (int d = 0; d < W; d++) {  // (checked for every home from the gathered counts: all ranks fail together)
      uint64_t to_d = 0;
      for (int o = 0; o < W; o++) to_d += mat2[(size_t)o * K2 + d];
      if (to_d >= 0xFFFFFFF0ull) { agreed = true; throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 hits come home in one hop: use smaller chunks (chunk_ranges)"}; }
    }
    reserve_mid_exchange(hits_in, std::max<size_t>(n_home * rec, 256));
    timed_comm([&] { comm->alltoallv(out_ptr, so.data(), sb.data(), hits_in.p, ro.data(), rb.data(), s); });
    bytes_out += acc * rec;
    uint64_t ops_home = 0;
    if (ship_ops) {  // ... and their CIGAR ops, owner by owner like the hit records
      for (int d = 0; d < W; d++) {
        uint64_t to_d = 0;
        for (int o = 0; o < W; o++) to_d += mat2[(size_t)o * K2 + W + d];
        if (to_d >= 0xFFFFFFF0ull) { agreed = true; throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 CIGAR ops come home in one hop: use smaller chunks (chunk_ranges)"}; }
      }
      acc = 0;
      for (int d = 0; d < W; d++) { so[d] = acc * 4; sb[d] = back[(size_t)W + d] * 4; acc += back[(size_t)W + d]; }
      for (int o = 0; o < W; o++) {
        const uint64_t c = mat2[(size_t)o * K2 + W + me];
        ro[o] = ops_home * 4; rb[o] = c * 4;
        ops_home += c;
      }
      reserve_mid_exchange(ops_in, std::max<size_t>(ops_home * 4, 256));
      timed_comm([&] { comm->alltoallv(ops_ptr, so.data(), sb.data(), ops_in.p, ro.data(), rb.data(), s); });
      bytes_out += acc * 4;
    }
    pieces.clear();
    pf[PF_BYTES_HITS] += (double)(bytes_out_hits(back, W, rec));
    pf[PF_HITS_HOME] += (double)n_home;
    lap(PF_A2A2);
    // ---- home: back into frontier order x visit order, into the slot arrays
    if (fail_home_hop && hop_no == fail_home_hop) throw Error{IMPG_E_INVALID, "injected failure (home side)"};
    L.n_pairs = (uint32_t)n_home;
    const size_t b = std::max<size_t>((size_t)n_home * 4, 256);
    L.pair_range.reserve(b); L.qid.reserve(b); L.coords.reserve(4 * b);
    HitArrays h{L.qid.as<uint32_t>(), L.coords.as<int4>()};
    const bool need_order = W > 1 || E.multi || any_by_place;
    if (E.multi) mslot.reserve(b);
    const uint32_t *slice_at = nullptr;
    uint32_t *slice_pos = nullptr, *slice_n = nullptr;
    if (ship_ops) {
      // where every arrived hit's ops start among the arrived ops: a scan of word 7 in arrival order
      L.sl_a.reserve(b); L.sl_n.reserve(b); L.sl_off.reserve(b); L.sl_rem.reserve(b);
      E.cnt.reserve(b); E.gid.reserve(b);
      L.slice_total = 0;
      if (n_home) {
        launch_hits_slice_n(hits_in.p, (uint32_t)n_home, E.cnt.as<uint32_t>(), s);
        const uint64_t tot = E.scan(E.cnt.as<uint32_t>(), E.gid.as<uint32_t>(), (uint32_t)n_home);
        if (tot != ops_home) throw Error{IMPG_E_INVALID, "CIGAR ops and hit records that came home disagree"};
        L.slice_total = tot;
      }
      slice_at = E.gid.as<uint32_t>();
      slice_pos = L.sl_a.as<uint32_t>();  // (kept in sl_a while the five-key sort may still permute the slots)
      slice_n = L.sl_n.as<uint32_t>();
      L.slice_pool.adopt(ops_in);  // (not swap: ops_in belongs to the lane and must not inherit the engine's buffer pool)
    }
    SliceArrays home_sl{nullptr, nullptr, nullptr, nullptr};
    if (ship_ops) home_sl = SliceArrays{L.sl_a.as<uint32_t>(), L.sl_n.as<uint32_t>(), L.sl_off.as<int32_t>(), L.sl_rem.as<int32_t>()};
    if (n_home) {
      if (need_order) {
        const size_t fb = std::max<size_t>((size_t)n_fr * 4, 256);
        E.lo_key.reserve(fb); E.lo_cnt.reserve(fb); E.lo_off.reserve(fb);
        IMPG_HIP(hipMemsetAsync(E.lo_cnt.p, 0, fb, s));
        IMPG_HIP(hipMemsetAsync(E.counters.as<unsigned long long>() + 4, 0, 8, s));
        uint32_t *err = reinterpret_cast<uint32_t *>(E.counters.as<unsigned long long>() + 4);
        launch_reorder_runs(hits_in.as<uint32_t>(), (uint32_t)n_home, words, n_fr, E.lo_key.as<uint32_t>(), E.lo_cnt.as<uint32_t>(), err, s);
        pinned_words((uint32_t)W);
        IMPG_HIP(hipMemcpyAsync(h_err((uint32_t)W), err, 4, hipMemcpyDeviceToHost, s));  // (read under the scan's own synchronisation)
        const uint64_t total = E.scan(E.lo_cnt.as<uint32_t>(), E.lo_off.as<uint32_t>(), n_fr);
        const uint32_t bad = *h_err((uint32_t)W);
        if (bad || total != n_home) throw Error{IMPG_E_INVALID, "hits came home for a frontier record twice or out of range"};
        launch_hits_unpack(hits_in.p, (uint32_t)n_home, words, n_fr, E.lo_key.as<uint32_t>(), E.lo_off.as<uint32_t>(),
                           L.pair_range.as<uint32_t>(), h, E.multi ? mslot.as<uint32_t>() : nullptr, s, slice_at, slice_pos, slice_n);
      } else {
        launch_hits_unpack(hits_in.p, (uint32_t)n_home, words, n_fr, nullptr, nullptr, L.pair_range.as<uint32_t>(), h, nullptr, s, slice_at,
                           slice_pos, slice_n);
      }
      if (E.multi) {
        iota.reserve(b);
        launch_iota(iota.as<uint32_t>(), (uint32_t)n_home, s);
      }
      E.post_expand(fr, n_fr, L, E.lo_off.as<uint32_t>(), iota.as<uint32_t>(), mslot.as<uint32_t>(), home_sl);
    }
    if (ship_ops) L.slice_pos.swap(L.sl_a);  // what the row builder reads (rows_device.hip): slice_pool[slice_pos[slot] .. + sl_n[slot])
    lap(PF_HOME);
    return HopResult{total_pairs, false};
  }
  static uint64_t bytes_out_hits(const std::vector<uint64_t> &back, int W, uint64_t rec) {
    uint64_t b = 0;
    for (int d = 0; d < W; d++) b += back[(size_t)d] * rec + back[(size_t)W + d] * 4;
    return b;
  }
};

struct ShardCtx {
  std::mutex gpu_turn;            // which lane's kernels are on the GPU
  impg_gpu_comm *comm = nullptr;  // borrowed (the host's, or the cluster's)
  std::vector<uint32_t> owner;
  DevBuf d_owner;
  std::vector<std::unique_ptr<ShardedExpander>> lanes;
};
struct Cluster {
  std::vector<std::unique_ptr<impg_gpu_comm>> comms;   // declared first: destroyed last
  std::vector<std::unique_ptr<impg_gpu_index>> ranks;
};

}  // namespace impg

impg_gpu_index::impg_gpu_index() {}
impg_gpu_index::~impg_gpu_index() {
  delete cluster;
  delete shard;
}

namespace impg {
namespace {

void attach_shard(impg_gpu_index &ix, impg_gpu_comm *comm, const std::vector<uint32_t> &owner) {
  auto S = std::make_unique<ShardCtx>();
  S->comm = comm;
  S->owner = owner;
  IMPG_HIP(hipSetDevice(ix.device));
  S->d_owner.reserve(std::max<size_t>(owner.size() * 4, 256));
  if (!owner.empty()) IMPG_HIP(hipMemcpy(S->d_owner.p, owner.data(), owner.size() * 4, hipMemcpyHostToDevice));
  for (size_t l = 0; l < comm->lanes.size(); l++) {
    auto x = std::make_unique<ShardedExpander>();
    x->comm = comm->lanes[l].get();
    x->d_owner = S->d_owner.as<uint32_t>();
    x->n_seq = (uint32_t)owner.size();
    x->gpu_turn = comm->lanes.size() > 1 ? &S->gpu_turn : nullptr;
    S->lanes.push_back(std::move(x));
  }
  ix.max_engines = std::max<int>(ix.max_engines, (int)comm->lanes.size());
  ix.shard = S.release();
}

struct LaneWork {  // what one lane accumulates
  impg_gpu_stats_t st;
  std::exception_ptr err;
};

// Runs body(lane, engine, chunk_begin, chunk_end) for every chunk of this rank's ranges, chunks dealt to the
// lanes round-robin; every rank runs the same number of chunks (ranks with fewer ranges run empty ones).
// prep(E) runs once per lane, right after the lane has taken its engine: whatever the batch hangs on an engine (mask,
// subset filter) belongs to the LEASE -- the lease's end clears it, and a lane that starts late can be handed the very
// engine an early lane has just given back.  (It used to be applied once per engine pointer: such a late lane then ran
// its chunks unmasked and unfiltered.  Found by scripts/fuzz_parity.py, seed 72686, 4 ranks x 2 lanes.)
// Ranges per chunk of this rank's part of a batch.  Without the option the batch is cut so that every lane has work
// (lanes only overlap exchange and compute across chunks) and no chunk outgrows the 32-bit slot counts of a hop.
// 50 000 ranges is the plain engine's chunk: what an owner expands in one hop is, summed over the homes, about one
// chunk's worth, and every chunk pays its hops' fixed costs (25 000: 74.8 ms per headline step on one rank, 50 000: 69.8).
size_t shard_chunk(const impg_gpu_index &ix, size_t n) {
  if (ix.opt_chunk_ranges) return ix.opt_chunk_ranges;
  const size_t lanes = std::max<size_t>(1, ix.shard->comm->lanes.size());
  return std::min<size_t>(50000, std::max<size_t>(1, (n + lanes - 1) / lanes));
}
template <class P, class F> void run_lanes(impg_gpu_index &ix, size_t n, P prep, F body) {
  ShardCtx &S = *ix.shard;
  const size_t chunk = shard_chunk(ix, n);
  uint64_t my_chunks = std::max<uint64_t>(1, (n + chunk - 1) / chunk);
  std::vector<uint64_t> all(S.comm->world);
  S.comm->lanes[0]->allgather_u64(&my_chunks, 1, all.data());
  const uint64_t n_chunks = *std::max_element(all.begin(), all.end());
  const size_t n_lanes = std::min<uint64_t>(S.comm->lanes.size(), n_chunks);
  std::vector<std::exception_ptr> errs(n_lanes);
  for (size_t l = 0; l < S.comm->lanes.size(); l++) {
    S.comm->lanes[l]->batch_begin(l < n_lanes);
    S.lanes[l]->agreed = S.lanes[l]->in_transport = false;
    S.lanes[l]->hop_no = 0;
    const uint32_t me1 = (uint32_t)S.comm->rank + 1;
    S.lanes[l]->fail_owner_hop = (ix.opt_debug_fail_owner >> 16) == me1 ? (ix.opt_debug_fail_owner & 0xFFFFu) : 0;
    S.lanes[l]->fail_home_hop = (ix.opt_debug_fail_home >> 16) == me1 ? (ix.opt_debug_fail_home & 0xFFFFu) : 0;
  }
  // Forced lane schedules (tests; option "lane_schedule" / IMPG_LANE_SCHEDULE = v > 0): bit l(l-1)/2 + l' of v - 1
  // makes lane l take its engine only after lane l' < l has run all its chunks and handed its engine back -- the
  // late lane is then given that very engine (the pool is last-in first-out).  v = 1 .. 2^(L(L-1)/2) enumerates every
  // hand-over pattern of L lanes instead of leaving them to thread timing (the class of the seed-72686 defect above).
  // Not with several RCCL lanes: their issue order (comm.hpp) needs every lane to start.
  uint64_t sched = ix.opt_lane_schedule;
  if (!sched) if (const char *e = getenv("IMPG_LANE_SCHEDULE")) sched = strtoull(e, nullptr, 10);
  if (n_lanes > 1 && std::string(S.comm->lanes[0]->kind()) == "rccl") sched = 0;
  std::mutex sched_m;
  std::condition_variable sched_cv;
  std::vector<char> lane_done(n_lanes, 0);
  auto lane_main = [&](size_t l) {
    ShardedExpander &X = *S.lanes[l];
    if (sched) {
      std::unique_lock<std::mutex> lk(sched_m);
      for (size_t lp = 0; lp < l; lp++)
        if (((sched - 1) >> (l * (l - 1) / 2 + lp)) & 1ull) sched_cv.wait(lk, [&] { return lane_done[lp] != 0; });
    }
    try {
      IMPG_HIP(hipSetDevice(ix.device));
      EngineLease lease(ix);
      Engine &E = *lease;
      E.remote = &X;
      prep(E);
      for (uint64_t c = l; c < n_chunks; c += n_lanes) {
        const size_t b = std::min<size_t>(n, c * chunk), e = std::min<size_t>(n, b + chunk);
        body(l, E, b, e);
      }
      X.closing_word(false);  // (a peer that failed after its last hop says so here)
    } catch (...) {
      errs[l] = std::current_exception();
      if (X.in_transport) {
        for (auto &c : S.comm->lanes) c->abort();  // the transport itself failed: release whoever waits for this rank, where it can be done
      } else if (!X.agreed) {
        // failed between collectives: the peers of this lane are, or will be, in an all-gather -- tell them there
        try { X.closing_word(true); } catch (...) { for (auto &c : S.comm->lanes) c->abort(); }
      }
    }
    S.comm->lanes[l]->batch_end();
    {
      std::lock_guard<std::mutex> lk(sched_m);
      lane_done[l] = 1;
    }
    sched_cv.notify_all();
  };
  if (n_lanes == 1) lane_main(0);
  else {
    std::vector<std::thread> th;
    for (size_t l = 0; l < n_lanes; l++) th.emplace_back(lane_main, l);
    for (auto &t : th) t.join();
  }
  if (auto e = first_cause(errs)) std::rethrow_exception(e);  // (the lane that failed first, not the lowest-numbered one)
}

void add_stats(impg_gpu_stats_t &tot, const impg_gpu_stats_t &st) {
  tot.projected += st.projected; tot.pairs += st.pairs; tot.frontier_ranges += st.frontier_ranges;
  tot.levels = std::max(tot.levels, st.levels);
  tot.ms_total += st.ms_total; tot.ms_lookup += st.ms_lookup; tot.ms_project += st.ms_project; tot.ms_update += st.ms_update;
  tot.project_launches += st.project_launches;
  tot.ms_exchange += st.ms_exchange;
}

// one rank's part of a collective counting batch
void rank_stats(impg_gpu_index &ix, const impg_gpu_range_t *ranges, bool on_device, size_t n, const impg_gpu_params_t &p,
                uint64_t *per_range_count, uint64_t *per_range_checksum, impg_gpu_stats_t *stats) {
  IMPG_HIP(hipSetDevice(ix.device));
  DevBuf d_ranges, d_cnt, d_ck;
  const impg_gpu_range_t *dr = ranges;
  if (!on_device) {
    check_ranges(ranges, n);
    d_ranges.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
    if (n) IMPG_HIP(hipMemcpy(d_ranges.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice));
    dr = d_ranges.as<impg_gpu_range_t>();
  }
  unsigned long long *dc = nullptr, *dk = nullptr;
  if (per_range_count) {
    d_cnt.reserve(std::max<size_t>(n * 8, 256));
    IMPG_HIP(hipMemset(d_cnt.p, 0, std::max<size_t>(n * 8, 8)));
    dc = d_cnt.as<unsigned long long>();
  }
  if (per_range_checksum) {
    d_ck.reserve(std::max<size_t>(n * 8, 256));
    IMPG_HIP(hipMemset(d_ck.p, 0, std::max<size_t>(n * 8, 8)));
    dk = d_ck.as<unsigned long long>();
  }
  ShardCtx &S = *ix.shard;
  std::vector<impg_gpu_stats_t> per_lane(S.comm->lanes.size());
  for (auto &x : per_lane) memset(&x, 0, sizeof x);
  for (auto &x : S.lanes) x->exchange_s = 0;
  run_lanes(ix, n, [](Engine &) {}, [&](size_t l, Engine &E, size_t b, size_t e) {
    impg_gpu_stats_t st;
    {
      std::unique_lock<std::mutex> turn(S.gpu_turn, std::defer_lock);
      if (S.comm->lanes.size() > 1) turn.lock();
      E.run(ix, dr + b, (uint32_t)(e - b), p, nullptr, dc ? dc + b : nullptr, dk ? dk + b : nullptr, &st, nullptr);
    }
    add_stats(per_lane[l], st);
  });
  impg_gpu_stats_t tot;
  memset(&tot, 0, sizeof tot);
  for (size_t l = 0; l < per_lane.size(); l++) {
    per_lane[l].ms_exchange = (float)(S.lanes[l]->exchange_s * 1e3);
    add_stats(tot, per_lane[l]);
  }
  if (stats) *stats = tot;
  if (per_range_count && n) IMPG_HIP(hipMemcpy(per_range_count, dc, n * 8, hipMemcpyDeviceToHost));
  if (per_range_checksum && n) IMPG_HIP(hipMemcpy(per_range_checksum, dk, n * 8, hipMemcpyDeviceToHost));
}

// one rank's part of a collective full-results batch
void rank_query(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p,
                const impg_gpu_mask_t *mask, const uint8_t *subset_keep, impg_gpu_results &res) {
  IMPG_HIP(hipSetDevice(ix.device));
  check_ranges(ranges, n);
  DevBuf d_ranges;
  d_ranges.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
  if (n) IMPG_HIP(hipMemcpy(d_ranges.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice));
  const size_t chunk = shard_chunk(ix, n);
  std::vector<std::unique_ptr<impg_gpu_results>> parts((n + chunk - 1) / chunk + 1);
  std::atomic<uint64_t> served{0};  // projections done here for other ranks' records during chunks this rank had no ranges for
  run_lanes(ix, n, [&](Engine &E) {
    apply_mask(E, ix, mask, p);  // every lane's lease carries the batch's mask / filter
    apply_subset(E, ix, subset_keep);
  }, [&](size_t, Engine &E, size_t b, size_t e) {
    std::vector<std::unique_ptr<LevelBufs>> levels;
    DevBuf self_dev;
    self_dev.pool = &E.level_pool;
    const auto c0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> turn(ix.shard->gpu_turn, std::defer_lock);
      if (ix.shard->comm->lanes.size() > 1) turn.lock();
      E.run(ix, d_ranges.as<impg_gpu_range_t>() + b, (uint32_t)(e - b), p, &levels, nullptr, nullptr, nullptr, &self_dev);
    }
    const auto c1 = std::chrono::steady_clock::now();
    if (e == b) { served += E.last_projected; return; }  // an empty chunk: this rank only took part in the hops
    auto part = std::make_unique<impg_gpu_results>();
    assemble_results(E, ranges + b, (uint32_t)(e - b), p, levels, self_dev, *part);
    part->run_s = std::chrono::duration<double>(c1 - c0).count();
    part->assemble_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c1).count();
    parts[b / chunk] = std::move(part);
  });
  res.offsets.assign(1, 0);
  for (auto &pt : parts)
    if (pt) append_results(res, *pt);
  res.projected += served.load();
  res.ranges.assign(ranges, ranges + n);
}

// one rank's part of a collective BED batch: the text of its own ranges, chunk by chunk
void rank_bed(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p, const uint8_t *subset_keep,
              int32_t merge_distance, const char *const *range_names, std::vector<std::string> &chunks_out, double *seconds3) {
  IMPG_HIP(hipSetDevice(ix.device));
  check_ranges(ranges, n);
  DevBuf d_ranges;
  d_ranges.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
  if (n) IMPG_HIP(hipMemcpy(d_ranges.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice));
  const size_t chunk = shard_chunk(ix, n);
  chunks_out.assign((n + chunk - 1) / chunk + 1, std::string());
  std::mutex tm;
  double t3[3] = {0, 0, 0};
  run_lanes(ix, n, [&](Engine &E) { apply_subset(E, ix, subset_keep); }, [&](size_t, Engine &E, size_t b, size_t e) {
    std::vector<std::unique_ptr<LevelBufs>> levels;
    DevBuf self_dev, rows;
    self_dev.pool = &E.level_pool;
    std::unique_lock<std::mutex> turn(ix.shard->gpu_turn, std::defer_lock);
    if (ix.shard->comm->lanes.size() > 1) turn.lock();
    const auto c0 = std::chrono::steady_clock::now();
    E.run(ix, d_ranges.as<impg_gpu_range_t>() + b, (uint32_t)(e - b), p, &levels, nullptr, nullptr, nullptr, &self_dev);
    const auto c1 = std::chrono::steady_clock::now();
    if (e == b) return;  // an empty chunk: this rank only took part in the hops
    const uint32_t n_rows = device_bed_rows(E, ix, (uint32_t)(e - b), p, merge_distance, levels, self_dev, rows);
    const auto c2 = std::chrono::steady_clock::now();
    std::string &out = chunks_out[b / chunk];
    device_bed_text(E, ix, rows, n_rows, (uint32_t)(e - b), bed_range_names(ix, ranges, range_names, b, e), p.original_sequence_coordinates != 0,
                    [&](const char *t, size_t k) { out.append(t, k); });
    const auto c3 = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(tm);
    t3[0] += std::chrono::duration<double>(c1 - c0).count();
    t3[1] += std::chrono::duration<double>(c2 - c1).count();
    t3[2] += std::chrono::duration<double>(c3 - c2).count();
  });
  if (seconds3) for (int k = 0; k < 3; k++) seconds3[k] = t3[k];
}

// ranges of a multi-GPU handle are dealt to the ranks in contiguous blocks (results concatenate in order)
void split_blocks(size_t n, size_t W, std::vector<size_t> &cut) {
  cut.resize(W + 1);
  for (size_t r = 0; r <= W; r++) cut[r] = n * r / W;
}

template <class F> void on_every_rank(Cluster &C, F f) {
  const size_t W = C.ranks.size();
  std::vector<std::exception_ptr> errs(W);
  std::vector<std::thread> th;
  for (auto &cm : C.comms)
    for (auto &l : cm->lanes) l->reset();  // (a batch that failed left the fabric poisoned)
  for (size_t r = 0; r < W; r++)
    th.emplace_back([&, r] {
      try {
        f(r);
      } catch (...) {
        errs[r] = std::current_exception();
        for (auto &c : C.comms[r]->lanes) c->abort();
      }
    });
  for (auto &t : th) t.join();
  // report the failure that came first, not the echoes it caused in the other ranks
  if (auto first = first_cause(errs)) std::rethrow_exception(first);
}

}  // namespace

int sharded_query_stats(impg_gpu_index &ix, const impg_gpu_range_t *ranges, bool on_device, size_t n,
                        const impg_gpu_params_t &p, uint64_t *per_range_count, uint64_t *per_range_checksum,
                        impg_gpu_stats_t *stats) {
  if (!ix.cluster) {
    rank_stats(ix, ranges, on_device, n, p, per_range_count, per_range_checksum, stats);
    return IMPG_OK;
  }
  Cluster &C = *ix.cluster;
  check_ranges(ranges, n);
  const size_t W = C.ranks.size();
  std::vector<size_t> cut;
  split_blocks(n, W, cut);
  std::vector<impg_gpu_stats_t> sts(W);
  for (auto &r : C.ranks) { r->opt_chunk_ranges = ix.opt_chunk_ranges; r->opt_pair_budget = ix.opt_pair_budget; r->opt_locality_min = ix.opt_locality_min;
                            r->opt_debug_fail_owner = ix.opt_debug_fail_owner; r->opt_debug_fail_home = ix.opt_debug_fail_home; r->opt_lane_schedule = ix.opt_lane_schedule; }
  on_every_rank(C, [&](size_t r) {
    rank_stats(*C.ranks[r], ranges + cut[r], false, cut[r + 1] - cut[r], p, per_range_count ? per_range_count + cut[r] : nullptr,
               per_range_checksum ? per_range_checksum + cut[r] : nullptr, &sts[r]);
  });
  if (stats) {
    memset(stats, 0, sizeof *stats);
    for (auto &s : sts) {  // work adds up; the ranks ran side by side, so time is the slowest rank's
      stats->projected += s.projected; stats->pairs += s.pairs; stats->frontier_ranges += s.frontier_ranges;
      stats->project_launches += s.project_launches;
      stats->levels = std::max(stats->levels, s.levels);
      stats->ms_total = std::max(stats->ms_total, s.ms_total); stats->ms_lookup = std::max(stats->ms_lookup, s.ms_lookup);
      stats->ms_project = std::max(stats->ms_project, s.ms_project); stats->ms_update = std::max(stats->ms_update, s.ms_update);
      stats->ms_exchange = std::max(stats->ms_exchange, s.ms_exchange);
    }
  }
  return IMPG_OK;
}

void sharded_bed_batch(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p,
                       const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names,
                       const std::function<void(const char *, size_t)> &sink, double *seconds3) {
  if (!ix.cluster) {  // one rank of a process-per-GPU job: the text of ITS ranges
    std::vector<std::string> chunks;
    rank_bed(ix, ranges, n, p, subset_keep, merge_distance, range_names, chunks, seconds3);
    for (auto &c : chunks) if (!c.empty()) sink(c.data(), c.size());
    return;
  }
  Cluster &C = *ix.cluster;
  const size_t W = C.ranks.size();
  std::vector<size_t> cut;
  split_blocks(n, W, cut);
  std::vector<std::vector<std::string>> parts(W);
  std::vector<std::array<double, 3>> secs(W);
  for (auto &r : C.ranks) { r->opt_chunk_ranges = ix.opt_chunk_ranges; r->opt_pair_budget = ix.opt_pair_budget; r->opt_locality_min = ix.opt_locality_min;
                            r->opt_debug_fail_owner = ix.opt_debug_fail_owner; r->opt_debug_fail_home = ix.opt_debug_fail_home; r->opt_lane_schedule = ix.opt_lane_schedule; }
  on_every_rank(C, [&](size_t r) {
    rank_bed(*C.ranks[r], ranges + cut[r], cut[r + 1] - cut[r], p, subset_keep, merge_distance, range_names ? range_names + cut[r] : nullptr, parts[r],
             secs[r].data());
  });
  for (auto &pr : parts)
    for (auto &c : pr) if (!c.empty()) sink(c.data(), c.size());
  if (seconds3) for (int k = 0; k < 3; k++) { seconds3[k] = 0; for (auto &x : secs) seconds3[k] = std::max(seconds3[k], x[k]); }
}

uint64_t shard_agree_max(impg_gpu_index &ix, uint64_t mine) {
  if (!ix.shard) return mine;
  ShardCtx &S = *ix.shard;
  std::vector<uint64_t> all((size_t)S.comm->world);
  S.comm->lanes[0]->allgather_u64(&mine, 1, all.data());
  return *std::max_element(all.begin(), all.end());
}

int sharded_query_batch(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p,
                        const impg_gpu_mask_t *mask, const uint8_t *subset_keep, impg_gpu_results **out) {
  if (mask && !p.transitive) throw Error{IMPG_E_INVALID, "masked_regions belong to the transitive queries"};
  auto res = std::make_unique<impg_gpu_results>();
  if (!ix.cluster) {
    rank_query(ix, ranges, n, p, mask, subset_keep, *res);
    *out = res.release();
    return IMPG_OK;
  }
  Cluster &C = *ix.cluster;
  const size_t W = C.ranks.size();
  std::vector<size_t> cut;
  split_blocks(n, W, cut);
  std::vector<impg_gpu_results> parts(W);
  for (auto &r : C.ranks) { r->opt_chunk_ranges = ix.opt_chunk_ranges; r->opt_pair_budget = ix.opt_pair_budget; r->opt_locality_min = ix.opt_locality_min;
                            r->opt_debug_fail_owner = ix.opt_debug_fail_owner; r->opt_debug_fail_home = ix.opt_debug_fail_home; r->opt_lane_schedule = ix.opt_lane_schedule; }
  on_every_rank(C, [&](size_t r) { rank_query(*C.ranks[r], ranges + cut[r], cut[r + 1] - cut[r], p, mask, subset_keep, parts[r]); });
  res->offsets.assign(1, 0);
  double run_s = 0, asm_s = 0;
  for (auto &pt : parts) {
    run_s = std::max(run_s, pt.run_s); asm_s = std::max(asm_s, pt.assemble_s);
    append_results(*res, pt);
  }
  res->run_s = run_s; res->assemble_s = asm_s;
  res->ranges.assign(ranges, ranges + n);
  *out = res.release();
  return IMPG_OK;
}

}  // namespace impg

using namespace impg;

#define IMPG_TRY try {
#define IMPG_CATCH                                          \
  }                                                         \
  catch (const impg::Error &e) {                            \
    impg::set_error(e.msg);                                 \
    return e.code;                                          \
  }                                                         \
  catch (const std::bad_alloc &) {                          \
    impg::set_error("host out of memory");                  \
    return IMPG_E_OOM;                                      \
  }                                                         \
  catch (const std::exception &e) {                         \
    impg::set_error(std::string("internal: ") + e.what());  \
    return IMPG_E_INVALID;                                  \
  }

namespace {
std::unique_ptr<impg_gpu_index> make_rank_index(const impg_gpu_record_t *records, size_t n_records, const uint32_t *ops, size_t n_ops,
                                                const int64_t *seq_len, uint32_t n_seq, const std::vector<uint64_t> *file_first,
                                                int bidirectional, int order_policy, int device, impg_gpu_comm *comm,
                                                const HostSeqIndex *seq, const std::vector<uint32_t> &owner) {
  auto ix = make_index(records, n_records, ops, n_ops, seq_len, n_seq, bidirectional, order_policy, device, (uint32_t)comm->rank,
                       (uint32_t)comm->world, seq, file_first, owner.data());
  attach_shard(*ix, comm, owner);
  return ix;
}
std::vector<uint32_t> owner_map(const impg_gpu_record_t *records, size_t n_records, uint32_t n_seq, int bidirectional, int world) {
  if ((!records && n_records)) throw Error{IMPG_E_INVALID, "null input array"};
  std::vector<uint64_t> cnt;
  count_entries_per_target(records, n_records, n_seq, bidirectional != 0, cnt);
  std::vector<uint32_t> owner(n_seq);
  shard_assign(cnt.data(), n_seq, (uint32_t)world, owner.data());
  return owner;
}
void check_comm(const impg_gpu_comm *comm) {
  if (!comm || comm->lanes.empty()) throw Error{IMPG_E_INVALID, "null communicator"};
  if (comm->world < 1 || comm->world > (int)ROUTE_WORLD_MAX) throw Error{IMPG_E_INVALID, "world size out of range"};
}

// the ranks' in-process communicators (one LocalComm per lane, all on one fabric per lane) and peer access between their devices
std::unique_ptr<Cluster> make_local_comms(const int *devices, int n_dev, int lanes) {
  auto C = std::make_unique<Cluster>();
  std::vector<std::shared_ptr<LocalFabric>> fabs;
  for (int l = 0; l < lanes; l++) fabs.push_back(std::make_shared<LocalFabric>(n_dev));
  for (int r = 0; r < n_dev; r++) {
    auto cm = std::make_unique<impg_gpu_comm>();
    cm->rank = r; cm->world = n_dev; cm->device = devices[r];
    for (int l = 0; l < lanes; l++) cm->lanes.emplace_back(new LocalComm(fabs[l], r, devices[r]));
    C->comms.push_back(std::move(cm));
  }
  // direct loads over xGMI between the shards' devices
  for (int a = 0; a < n_dev; a++)
    for (int b = 0; b < n_dev; b++) {
      if (devices[a] == devices[b]) continue;
      IMPG_HIP(hipSetDevice(devices[a]));
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
      if (can) {
        hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
  return C;
}

std::unique_ptr<impg_gpu_index> make_cluster(const impg_gpu_record_t *records, size_t n_records, const uint32_t *ops, size_t n_ops,
                                             const int64_t *seq_len, uint32_t n_seq, const std::vector<uint64_t> *file_first,
                                             int bidirectional, int order_policy, const int *devices, int n_dev, int lanes,
                                             const HostSeqIndex *seq, const TpInput *tp = nullptr) {
  if (!devices || n_dev < 1 || n_dev > (int)ROUTE_WORLD_MAX) throw Error{IMPG_E_INVALID, "bad device list"};
  if (lanes < 1 || lanes > 8) throw Error{IMPG_E_INVALID, "lanes must be 1..8"};
  for (int r = 0; r < n_dev; r++) require_device(devices[r]);
  const std::vector<uint32_t> owner = owner_map(records, n_records, n_seq, bidirectional, n_dev);
  auto C = make_local_comms(devices, n_dev, lanes);
  C->ranks.resize(n_dev);
  std::vector<std::exception_ptr> errs(n_dev);
  std::vector<std::thread> th;
  for (int r = 0; r < n_dev; r++)
    th.emplace_back([&, r] {
      try {
        if (tp) {  // a tracepoint index (approximate mode): the host builder takes the rank's share of the alignments
          require_device(devices[r]);
          auto ix = std::make_unique<impg_gpu_index>();
          ix->device = devices[r];
          if (seq) ix->seq = *seq;
          build_index(*ix, records, n_records, nullptr, n_ops, seq_len, n_seq, bidirectional != 0, order_policy, (uint32_t)r, (uint32_t)n_dev,
                      owner.data(), tp);
          { EngineLease warm(*ix); }
          attach_shard(*ix, C->comms[r].get(), owner);
          C->ranks[r] = std::move(ix);
        } else
        C->ranks[r] = make_rank_index(records, n_records, ops, n_ops, seq_len, n_seq, file_first, bidirectional, order_policy,
                                      devices[r], C->comms[r].get(), seq, owner);
      } catch (...) { errs[r] = std::current_exception(); }
    });
  for (auto &t : th) t.join();
  for (auto &e : errs)
    if (e) std::rethrow_exception(e);
  auto front = std::make_unique<impg_gpu_index>();
  front->device = devices[0];
  if (seq) front->seq = *seq;
  else front->seq.lens.assign(seq_len, seq_len + n_seq);
  front->n_records = n_records;
  std::vector<uint64_t> cnt;
  count_entries_per_target(records, n_records, n_seq, bidirectional != 0, cnt);
  front->h_tgt_off.assign(n_seq + 1, 0);
  for (uint32_t t = 0; t < n_seq; t++) {
    front->h_tgt_off[t + 1] = (uint32_t)std::min<uint64_t>(front->h_tgt_off[t] + cnt[t], 0xFFFFFFFFull);
    if (cnt[t]) front->n_targets++;
    front->n_entries += cnt[t];
  }
  for (auto &r : C->ranks) front->device_bytes += r->device_bytes;
  front->view.n_seq = n_seq;
  front->cluster = C.release();
  return front;
}

// ---- a sharded index on disk (impg.rs:1655-1850 is the role: start-up without re-reading the alignments) ----------------
// A rank's shard is one file -- the arrays of that shard plus {world, rank, target -> rank map}; a multi handle is its
// front file (host tables, the map) at `path` and its shards at path.shard<k>of<n>.  Loading wants the same world.
static std::string shard_path(const char *path, int k, int n) { return std::string(path) + ".shard" + std::to_string(k) + "of" + std::to_string(n); }
}  // namespace
namespace impg {
void save_sharded(const impg_gpu_index &ix, const char *path) {
  if (ix.shard) {
    ShardInfo si;
    si.world = (uint32_t)ix.shard->comm->world; si.rank = (uint32_t)ix.shard->comm->rank; si.owner = ix.shard->owner;
    save_index(ix, path, &si, false);
    return;
  }
  const Cluster &C = *ix.cluster;
  const int n = (int)C.ranks.size();
  ShardInfo fi;
  fi.world = (uint32_t)n; fi.rank = 0; fi.owner = C.ranks[0]->shard->owner;
  for (int r = 0; r < n; r++) {
    ShardInfo si;
    si.world = (uint32_t)n; si.rank = (uint32_t)r; si.owner = fi.owner;
    save_index(*C.ranks[r], shard_path(path, r, n).c_str(), &si, false);
  }
  save_index(ix, path, &fi, true);  // written last: a front file means its shards are complete
}
}  // namespace impg

extern "C" {

int impg_gpu_index_load_rank(const char *path, int device, impg_gpu_comm_t *comm, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!path || !out) throw Error{IMPG_E_INVALID, "null argument"};
  check_comm(comm);
  require_device(device);
  auto ix = std::make_unique<impg_gpu_index>();
  ix->device = device;
  ShardInfo si;
  load_index(*ix, path, &si);
  if (si.front) throw Error{IMPG_E_INVALID, std::string(path) + " is the front file of a multi handle: load it with impg_gpu_index_load_multi"};
  if ((int)si.world != comm->world || (int)si.rank != comm->rank)
    throw Error{IMPG_E_INVALID, std::string(path) + " is shard " + std::to_string(si.rank) + " of " + std::to_string(si.world) +
                                ", this communicator is rank " + std::to_string(comm->rank) + " of " + std::to_string(comm->world)};
  { EngineLease warm(*ix); }
  attach_shard(*ix, comm, si.owner);
  *out = ix.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_load_multi(const char *path, const int *devices, int n_dev, int lanes, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!path || !out || !devices || n_dev < 1 || n_dev > (int)ROUTE_WORLD_MAX) throw Error{IMPG_E_INVALID, "bad arguments"};
  if (lanes < 1 || lanes > 8) throw Error{IMPG_E_INVALID, "lanes must be 1..8"};
  for (int r = 0; r < n_dev; r++) require_device(devices[r]);
  auto front = std::make_unique<impg_gpu_index>();
  front->device = devices[0];
  ShardInfo fi;
  load_index(*front, path, &fi);
  if (!fi.front) throw Error{IMPG_E_INVALID, std::string(path) + " is one shard, not the front file of a multi handle"};
  if ((int)fi.world != n_dev) throw Error{IMPG_E_INVALID, std::string(path) + " was saved over " + std::to_string(fi.world) + " GPUs, asked for " + std::to_string(n_dev)};
  auto C = make_local_comms(devices, n_dev, lanes);
  C->ranks.resize(n_dev);
  std::vector<std::exception_ptr> errs(n_dev);
  std::vector<std::thread> th;
  for (int r = 0; r < n_dev; r++)
    th.emplace_back([&, r] {
      try {
        auto ix = std::make_unique<impg_gpu_index>();
        ix->device = devices[r];
        ShardInfo si;
        const std::string sp = shard_path(path, r, n_dev);
        load_index(*ix, sp.c_str(), &si);
        if (si.front || (int)si.world != n_dev || (int)si.rank != r || si.owner != fi.owner)
          throw Error{IMPG_E_INVALID, sp + " does not belong to " + path};
        { EngineLease warm(*ix); }
        attach_shard(*ix, C->comms[r].get(), si.owner);
        C->ranks[r] = std::move(ix);
      } catch (...) { errs[r] = std::current_exception(); }
    });
  for (auto &t : th) t.join();
  for (auto &e : errs)
    if (e) std::rethrow_exception(e);
  front->device_bytes = 0;
  for (auto &r : C->ranks) front->device_bytes += r->device_bytes;
  front->cluster = C.release();
  *out = front.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_shard_assign(const uint64_t *entries_per_target, uint32_t n_seq, uint32_t n_shards, uint32_t *owner_out) {
  IMPG_TRY
  if ((n_seq && (!entries_per_target || !owner_out))) throw Error{IMPG_E_INVALID, "null argument"};
  shard_assign(entries_per_target, n_seq, n_shards, owner_out);
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_comm_unique_id(uint8_t *ids, int lanes) {
  IMPG_TRY
  if (!ids || lanes < 1 || lanes > 8) throw Error{IMPG_E_INVALID, "bad arguments"};
  for (int l = 0; l < lanes; l++) rccl_unique_id(ids + (size_t)l * IMPG_COMM_ID_BYTES);
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_comm_create_rccl(const uint8_t *ids, int lanes, int rank, int world, int device, impg_gpu_comm_t **out) {
  IMPG_TRY
  if (!ids || !out || lanes < 1 || lanes > 8 || world < 1 || rank < 0 || rank >= world) throw Error{IMPG_E_INVALID, "bad arguments"};
  require_device(device);
  auto cm = std::make_unique<impg_gpu_comm>();
  cm->rank = rank; cm->world = world; cm->device = device;
  auto order = lanes > 1 ? std::make_shared<IssueOrder>(lanes) : nullptr;
  for (int l = 0; l < lanes; l++) {
    auto rc = std::make_unique<RcclComm>(ids + (size_t)l * IMPG_COMM_ID_BYTES, rank, world, device);
    rc->order = order;
    rc->lane = l;
    cm->lanes.push_back(std::move(rc));
  }
  *out = cm.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_comm_create_host(const impg_gpu_host_transport_t *transports, int lanes, int rank, int world, int device,
                              impg_gpu_comm_t **out) {
  IMPG_TRY
  if (!transports || !out || lanes < 1 || lanes > 8 || world < 1 || rank < 0 || rank >= world) throw Error{IMPG_E_INVALID, "bad arguments"};
  auto cm = std::make_unique<impg_gpu_comm>();
  cm->rank = rank; cm->world = world; cm->device = device;
  for (int l = 0; l < lanes; l++) cm->lanes.emplace_back(new HostComm(transports[l], rank, world));
  *out = cm.release();
  return IMPG_OK;
  IMPG_CATCH
}

void impg_gpu_comm_destroy(impg_gpu_comm_t *c) { delete c; }

int impg_gpu_comm_info(const impg_gpu_comm_t *c, int *rank, int *world, int *lanes, const char **kind) {
  IMPG_TRY
  check_comm(c);
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (lanes) *lanes = (int)c->lanes.size();
  if (kind) *kind = c->lanes[0]->kind();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_comm_check(impg_gpu_comm_t *c, int lane) {
  IMPG_TRY
  check_comm(c);
  if (lane < 0 || lane >= (int)c->lanes.size()) throw Error{IMPG_E_INVALID, "no such lane"};
  Comm &cm = *c->lanes[lane];
  const int W = cm.world, me = cm.rank;
  // all-gather: rank r contributes (r+1)*1000 + i
  const size_t K = 3;
  std::vector<uint64_t> mine(K), all(K * W);
  for (size_t i = 0; i < K; i++) mine[i] = (uint64_t)(me + 1) * 1000 + i;
  cm.allgather_u64(mine.data(), K, all.data());
  for (int r = 0; r < W; r++)
    for (size_t i = 0; i < K; i++)
      if (all[(size_t)r * K + i] != (uint64_t)(r + 1) * 1000 + i) throw Error{IMPG_E_IO, "transport check: all-gather returned wrong values"};
  HostComm *hc = dynamic_cast<HostComm *>(&cm);
  if (!hc) return IMPG_OK;  // device transports are exercised by the queries themselves
  // all-to-all-v over host memory, ragged: rank s sends (s + 2 d + 1) words to rank d, word j = s<<20 | d<<10 | j
  std::vector<uint64_t> so(W), sb(W), ro(W), rb(W);
  uint64_t st = 0, rt = 0;
  for (int d = 0; d < W; d++) { so[d] = st; sb[d] = (uint64_t)(me + 2 * d + 1) * 4; st += sb[d]; }
  for (int s2 = 0; s2 < W; s2++) { ro[s2] = rt; rb[s2] = (uint64_t)(s2 + 2 * me + 1) * 4; rt += rb[s2]; }
  std::vector<uint32_t> snd(st / 4 + 1), rcv(rt / 4 + 1, 0xFFFFFFFFu);
  for (int d = 0; d < W; d++)
    for (uint64_t j = 0; j < sb[d] / 4; j++) snd[so[d] / 4 + j] = (uint32_t)me << 20 | (uint32_t)d << 10 | (uint32_t)j;
  if (hc->t.alltoallv(hc->t.ctx, snd.data(), so.data(), sb.data(), rcv.data(), ro.data(), rb.data()) != 0)
    throw Error{IMPG_E_IO, "transport check: alltoallv callback failed"};
  for (int s2 = 0; s2 < W; s2++)
    for (uint64_t j = 0; j < rb[s2] / 4; j++)
      if (rcv[ro[s2] / 4 + j] != ((uint32_t)s2 << 20 | (uint32_t)me << 10 | (uint32_t)j))
        throw Error{IMPG_E_IO, "transport check: all-to-all-v delivered wrong bytes"};
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_rank(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                               const int64_t *seq_len, uint32_t n_seq, const uint64_t *file_first_record, uint32_t n_files,
                               int bidirectional, int order_policy, int device, impg_gpu_comm_t *comm, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out) throw Error{IMPG_E_INVALID, "null out"};
  check_comm(comm);
  std::vector<uint64_t> ff;
  if (file_first_record && n_files) { ff.assign(file_first_record, file_first_record + n_files); ff.push_back(n_records); }
  const std::vector<uint32_t> owner = owner_map(records, n_records, n_seq, bidirectional, comm->world);
  *out = make_rank_index(records, n_records, cigar_ops, n_ops, seq_len, n_seq, ff.empty() ? nullptr : &ff, bidirectional, order_policy,
                         device, comm, nullptr, owner).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_from_paf_rank(const char *const *paths, int n_paths, int bidirectional, int order_policy, int device,
                                        impg_gpu_comm_t *comm, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !paths || n_paths <= 0) throw Error{IMPG_E_INVALID, "bad arguments"};
  check_comm(comm);
  require_device(device);
  ParsedPaf pp;
  std::vector<std::string> ps(paths, paths + n_paths);
  parse_paf_files(ps, pp);
  std::vector<int64_t> lens = pp.seq.lens;
  const std::vector<uint32_t> owner = owner_map(pp.records.data(), pp.records.size(), (uint32_t)lens.size(), bidirectional, comm->world);
  *out = make_rank_index(pp.records.data(), pp.records.size(), pp.ops.data(), pp.ops.size(), lens.data(), (uint32_t)lens.size(),
                         &pp.file_first, bidirectional, order_policy, device, comm, &pp.seq, owner).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_multi(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                                const int64_t *seq_len, uint32_t n_seq, const uint64_t *file_first_record, uint32_t n_files,
                                int bidirectional, int order_policy, const int *devices, int n_dev, int lanes,
                                impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || (!seq_len && n_seq)) throw Error{IMPG_E_INVALID, "null argument"};
  std::vector<uint64_t> ff;
  if (file_first_record && n_files) { ff.assign(file_first_record, file_first_record + n_files); ff.push_back(n_records); }
  *out = make_cluster(records, n_records, cigar_ops, n_ops, seq_len, n_seq, ff.empty() ? nullptr : &ff, bidirectional, order_policy,
                      devices, n_dev, lanes, nullptr).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_tracepoints_multi(const impg_gpu_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                            const int32_t *query_deltas, const int32_t *diffs, size_t n_segs_total,
                                            const impg_gpu_tp_mode_t *mode, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                            int order_policy, const int *devices, int n_dev, int lanes, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !mode || (n_records && !records) || (n_segs_total && !tracepoints) || (n_seq && !seq_len))
    throw Error{IMPG_E_INVALID, "null argument"};
  if (mode->fastga) {
    if (n_segs_total && !diffs) throw Error{IMPG_E_INVALID, "FASTGA tracepoints come with per-segment diffs"};
    if (mode->trace_spacing <= 0) throw Error{IMPG_E_INVALID, "trace_spacing must be positive"};
  } else if (n_segs_total && !query_deltas) throw Error{IMPG_E_INVALID, "Standard tracepoints come with per-segment query deltas"};
  std::vector<impg_gpu_record_t> recs(n_records);
  for (size_t i = 0; i < n_records; i++) {
    const impg_gpu_tp_record_t &r = records[i];
    if (r.seg_off + r.n_segs > n_segs_total) throw Error{IMPG_E_INVALID, "record segments outside the pools"};
    recs[i] = impg_gpu_record_t{r.query_id, r.target_id, r.query_start, r.query_end, r.target_start, r.target_end, r.seg_off, r.n_segs, r.strand};
  }
  TpInput tp{records, tracepoints, query_deltas, diffs, n_segs_total, *mode};
  auto ix = make_cluster(recs.data(), n_records, nullptr, n_segs_total, seq_len, n_seq, nullptr, bidirectional, order_policy, devices, n_dev, lanes,
                         nullptr, &tp);
  ix->tp_mode = true;
  *out = ix.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_from_paf_multi(const char *const *paths, int n_paths, int bidirectional, int order_policy,
                                         const int *devices, int n_dev, int lanes, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !paths || n_paths <= 0) throw Error{IMPG_E_INVALID, "bad arguments"};
  ParsedPaf pp;
  std::vector<std::string> ps(paths, paths + n_paths);
  parse_paf_files(ps, pp);
  std::vector<int64_t> lens = pp.seq.lens;
  *out = make_cluster(pp.records.data(), pp.records.size(), pp.ops.data(), pp.ops.size(), lens.data(), (uint32_t)lens.size(),
                      &pp.file_first, bidirectional, order_policy, devices, n_dev, lanes, &pp.seq).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_shard_info(const impg_gpu_index_t *ix, int *rank, int *world, int *lanes, uint32_t *owner_out, size_t cap) {
  IMPG_TRY
  if (!ix) throw Error{IMPG_E_INVALID, "null argument"};
  const ShardCtx *S = ix->shard ? ix->shard : (ix->cluster && !ix->cluster->ranks.empty() ? ix->cluster->ranks[0]->shard : nullptr);
  if (rank) *rank = ix->shard ? S->comm->rank : (ix->cluster ? -1 : 0);
  if (world) *world = S ? S->comm->world : 1;
  if (lanes) *lanes = S ? (int)S->comm->lanes.size() : 1;
  if (owner_out && S)
    for (size_t t = 0; t < S->owner.size() && t < cap; t++) owner_out[t] = S->owner[t];
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_hop_profile(impg_gpu_index_t *ix, double *out, size_t cap, int reset, size_t *n_out) {
  IMPG_TRY
  if (!ix || !n_out) throw Error{IMPG_E_INVALID, "null argument"};
  std::vector<ShardCtx *> ctx;
  if (ix->shard) ctx.push_back(ix->shard);
  else if (ix->cluster) for (auto &r : ix->cluster->ranks) if (r->shard) ctx.push_back(r->shard);
  constexpr size_t per = (size_t)ShardedExpander::PROF_HOPS * ShardedExpander::PROF_FIELDS;
  *n_out = ctx.size() * per;
  if (out && cap < *n_out) throw Error{IMPG_E_INVALID, "hop profile: output too small"};
  for (size_t r = 0; r < ctx.size(); r++) {
    if (out) std::fill(out + r * per, out + (r + 1) * per, 0.0);
    for (auto &x : ctx[r]->lanes) {
      if (out)
        for (int h = 0; h < ShardedExpander::PROF_HOPS; h++)
          for (int f = 0; f < ShardedExpander::PROF_FIELDS; f++) out[r * per + (size_t)h * ShardedExpander::PROF_FIELDS + f] += x->prof[h][f];
      if (reset) memset(x->prof, 0, sizeof x->prof);
    }
  }
  return IMPG_OK;
  IMPG_CATCH
}

}  // extern "C"
