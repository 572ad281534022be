// Host orchestration of a query batch on one GPU: the level-synchronous
// transitive closure of Impg::query_transitive_bfs (src/impg.rs:2311-2597) run
// for ALL ranges of the batch at once (the reference loops over ranges serially,
// src/main.rs:7435), and the one-level Impg::query (src/impg.rs:1852-1928).
//
// Per level:  lookup_count -> scan -> lookup_emit -> project   [-> visited update
// -> next frontier].  The update is skipped on the last level (depth+1 ==
// max_depth): its only outputs, the visited sets and the next frontier, are
// never read again (impg.rs:2376 ends the loop).
#include <cmath>
#include <memory>

#include "engine.hpp"

namespace impg {

namespace {
inline unsigned bits_for(uint64_t n) {  // bits needed to represent values < n
  unsigned b = 0;
  while ((1ull << b) < n) b++;
  return b;
}
}  // namespace

Engine::Engine(int device) {
  IMPG_HIP(hipSetDevice(device));
  IMPG_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  counters.reserve(64);
  acc_slots.reserve(COUNT_BYTES);
  act_slots.reserve(COUNT_BYTES);
  IMPG_HIP(hipHostMalloc((void **)&h_counters, 64, hipHostMallocDefault));
  IMPG_HIP(hipHostMalloc((void **)&h_slots, COUNT_BYTES, hipHostMallocDefault));
}
Engine::~Engine() {
  for (auto e : ev_pool) (void)hipEventDestroy(e);
  if (h_counters) (void)hipHostFree(h_counters);
  if (h_slots) (void)hipHostFree(h_slots);
  if (stream) (void)hipStreamDestroy(stream);
}
hipEvent_t Engine::event() {
  if (ev_next == ev_pool.size()) {
    hipEvent_t e;
    IMPG_HIP(hipEventCreate(&e));
    ev_pool.push_back(e);
  }
  return ev_pool[ev_next++];
}

// counters layout (device, 8 x u64): 0 total pairs, 1 accepted, 2 err flag,
// 3 scan total (groups / caps / pieces), 4 active keys
uint64_t Engine::read_counter(int k) {
  IMPG_HIP(hipMemcpyAsync(h_counters, counters.as<uint64_t>() + k, 8, hipMemcpyDeviceToHost, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  return h_counters[0];
}

uint64_t Engine::read_slots(DevBuf &b) {
  IMPG_HIP(hipMemcpyAsync(h_slots, b.p, COUNT_BYTES, hipMemcpyDeviceToHost, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  uint64_t t = 0;
  for (uint32_t k = 0; k < COUNT_SLOTS; k++) t += h_slots[k * COUNT_STRIDE];
  return t;
}

uint64_t Engine::scan(const uint32_t *in, uint32_t *out, uint32_t n) {
  if (n == 0) return 0;
  scan_tmp.reserve(scan_scratch_bytes(n));
  launch_exclusive_scan(in, out, n, scan_tmp.as<unsigned long long>(), counters.as<unsigned long long>() + 3, stream);
  return read_counter(3);
}

static HitArrays hit_arrays(LevelBufs &L, uint32_t n_pairs) {
  size_t b = std::max<size_t>((size_t)n_pairs * 4, 256);
  L.qid.reserve(b); L.qs.reserve(b); L.qe.reserve(b); L.ts.reserve(b); L.te.reserve(b);
  return HitArrays{L.qid.as<uint32_t>(), L.qs.as<int32_t>(), L.qe.as<int32_t>(), L.ts.as<int32_t>(), L.te.as<int32_t>()};
}

// lookup + projection of one frontier; fills L (pair_range, hit arrays), returns #pairs
uint64_t Engine::expand(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                        impg_gpu_stats_t *st) {
  hipEvent_t e0 = event(), e1 = event(), e2 = event();
  IMPG_HIP(hipEventRecord(e0, stream));
  cnt.reserve((size_t)n_fr * 4);
  win.reserve((size_t)n_fr * 8);
  pair_off.reserve((size_t)n_fr * 4);
  launch_lookup_count(v, fr, n_fr, transitive, cnt.as<uint32_t>(), win.as<uint2>(), stream);
  uint64_t P = scan(cnt.as<uint32_t>(), pair_off.as<uint32_t>(), n_fr);
  if (P > pair_budget || P >= 0xFFFFFFF0ull) {
    if (split_ok) throw SplitBatch{};
    if (P >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 candidate pairs for a single range"};
  }
  L.n_pairs = (uint32_t)P;
  L.pair_range.reserve(std::max<size_t>(P * 4, 256));
  pair_entry.reserve(std::max<size_t>(P * 4, 256));
  launch_lookup_emit(v, fr, n_fr, transitive, pair_off.as<uint32_t>(), win.as<uint2>(), L.pair_range.as<uint32_t>(),
                     pair_entry.as<uint32_t>(), stream);
  IMPG_HIP(hipEventRecord(e1, stream));
  HitArrays h = hit_arrays(L, L.n_pairs);
  launch_project(v, fr, L.pair_range.as<uint32_t>(), pair_entry.as<uint32_t>(), L.n_pairs, transitive, h,
                 acc_slots.as<unsigned long long>(), (uint32_t *)(counters.as<uint64_t>() + 2), min_identity, stream);
  IMPG_HIP(hipEventRecord(e2, stream));
  timed.push_back({e0, e1, 0});
  timed.push_back({e1, e2, 1});
  if (st) {
    st->pairs += P;
    st->frontier_ranges += n_fr;
    if (P) st->project_launches += 1;
  }
  return P;
}

// visited update + next frontier (impg.rs:2471-2584).  Returns the next frontier size.
uint32_t Engine::update(const DeviceIndexView &v, const FrontierRec *fr, LevelBufs &L, uint32_t n_queries,
                        const impg_gpu_params_t &p, DevBuf &next_frontier) {
  hipEvent_t e0 = event(), e1 = event();
  IMPG_HIP(hipEventRecord(e0, stream));
  const uint32_t P = L.n_pairs;
  uint32_t n_next = 0;
  HitArrays h{L.qid.as<uint32_t>(), L.qs.as<int32_t>(), L.qe.as<int32_t>(), L.ts.as<int32_t>(), L.te.as<int32_t>()};
  if (P) {
    keys.reserve((size_t)P * 8); skeys.reserve((size_t)P * 8);
    vals.reserve((size_t)P * 4); svals.reserve((size_t)P * 4);
    IMPG_HIP(hipMemsetAsync(act_slots.p, 0, COUNT_BYTES, stream));
    launch_update_keys(fr, L.pair_range.as<uint32_t>(), P, h, keys.as<unsigned long long>(), vals.as<uint32_t>(),
                       act_slots.as<unsigned long long>(), stream);
    size_t tb = sort_pairs_scratch_bytes(P);
    sort_tmp.reserve(tb);
    launch_sort_pairs(sort_tmp.p, tb, keys.as<unsigned long long>(), skeys.as<unsigned long long>(), vals.as<uint32_t>(),
                      svals.as<uint32_t>(), P, 32 + std::max(1u, bits_for(n_queries)), stream);
    head.reserve((size_t)P * 4); gid.reserve((size_t)P * 4);
    launch_group_heads(skeys.as<unsigned long long>(), P, head.as<uint32_t>(), stream);
    uint32_t n_groups = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), P);
    if (n_groups) {
      auto vt = std::make_unique<VisitedStore>();
      vt->keys.reserve((size_t)n_groups * 8);
      gstart.reserve((size_t)n_groups * 4);
      launch_group_scatter(skeys.as<unsigned long long>(), P, head.as<uint32_t>(), gid.as<uint32_t>(), gstart.as<uint32_t>(),
                           vt->keys.as<unsigned long long>(), stream);
      const uint32_t n_active = (uint32_t)read_slots(act_slots);  // hits that carry a (query, sequence) key
      glen.reserve((size_t)n_groups * 4); old_tab.reserve((size_t)n_groups * 4); old_idx.reserve((size_t)n_groups * 4);
      cap.reserve((size_t)n_groups * 4); pcap.reserve((size_t)n_groups * 4);
      VisitedTables tabs = tables_view();
      launch_group_prepare(tabs, vt->keys.as<unsigned long long>(), gstart.as<uint32_t>(), n_groups, n_active,
                           glen.as<uint32_t>(), old_tab.as<uint32_t>(), old_idx.as<uint32_t>(), cap.as<uint32_t>(),
                           pcap.as<uint32_t>(), stream);
      vt->off.reserve((size_t)n_groups * 4);
      poff.reserve((size_t)n_groups * 4);
      uint64_t cap_total = scan(cap.as<uint32_t>(), vt->off.as<uint32_t>(), n_groups);
      uint64_t pcap_total = scan(pcap.as<uint32_t>(), poff.as<uint32_t>(), n_groups);
      if (cap_total >= 0xFFFFFFF0ull || pcap_total >= 0xFFFFFFF0ull)
        { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "visited sets exceed 2^32 ranges"}; }
      vt->ranges.reserve(std::max<size_t>(cap_total * 8, 256));
      vt->len.reserve((size_t)n_groups * 4);
      pieces.reserve(std::max<size_t>(pcap_total * 8, 256));
      n_pieces.reserve((size_t)n_groups * 4);
      foff.reserve((size_t)n_groups * 4);
      launch_visited_update(tabs, svals.as<uint32_t>(), h, v.seq_len, vt->keys.as<unsigned long long>(),
                            gstart.as<uint32_t>(), glen.as<uint32_t>(), old_tab.as<uint32_t>(), old_idx.as<uint32_t>(),
                            vt->off.as<uint32_t>(), poff.as<uint32_t>(), n_groups, p.min_transitive_len,
                            p.min_distance_between_ranges, vt->ranges.as<int2>(), vt->len.as<uint32_t>(),
                            pieces.as<int2>(), n_pieces.as<uint32_t>(), stream);
      uint64_t nn = scan(n_pieces.as<uint32_t>(), foff.as<uint32_t>(), n_groups);
      if (nn >= 0xFFFFFFF0ull) { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "frontier exceeds 2^32 ranges"}; }
      n_next = (uint32_t)nn;
      next_frontier.reserve(std::max<size_t>((size_t)n_next * sizeof(FrontierRec), 256));
      launch_frontier_emit(vt->keys.as<unsigned long long>(), poff.as<uint32_t>(), n_pieces.as<uint32_t>(),
                           foff.as<uint32_t>(), n_groups, pieces.as<int2>(), next_frontier.as<FrontierRec>(), stream);
      vt->n_groups = n_groups;
      if (tables.size() + 1 >= (size_t)MAX_VISITED_TABLES)
        throw Error{IMPG_E_UNSUPPORTED, "transitive depth exceeds the visited-table limit"};
      tables.push_back(std::move(vt));
    }
  }
  IMPG_HIP(hipEventRecord(e1, stream));
  timed.push_back({e0, e1, 2});
  return n_next;
}

VisitedTables Engine::tables_view() const {
  VisitedTables t;
  memset(&t, 0, sizeof t);
  t.n_tables = (uint32_t)tables.size();
  for (size_t i = 0; i < tables.size(); i++) {
    t.t[i].keys = tables[i]->keys.as<unsigned long long>();
    t.t[i].off = tables[i]->off.as<uint32_t>();
    t.t[i].len = tables[i]->len.as<uint32_t>();
    t.t[i].ranges = tables[i]->ranges.as<int2>();
    t.t[i].n_groups = tables[i]->n_groups;
  }
  return t;
}

uint32_t Engine::begin_transitive(const DeviceIndexView &v, const impg_gpu_range_t *d_ranges, uint32_t n,
                                  const impg_gpu_params_t &p, FrontierRec *d_self, DevBuf &frontier_out) {
  tables.clear();
  auto t = std::make_unique<VisitedStore>();
  t->keys.reserve((size_t)n * 8); t->off.reserve((size_t)n * 4); t->len.reserve((size_t)n * 4);
  t->ranges.reserve((size_t)n * 8);
  t->n_groups = n;
  head.reserve((size_t)n * 4); gid.reserve((size_t)n * 4);
  launch_visited_init(d_ranges, n, v.seq_len, v.n_seq, p.min_transitive_len, t->keys.as<unsigned long long>(),
                      t->off.as<uint32_t>(), t->len.as<uint32_t>(), t->ranges.as<int2>(), d_self, head.as<uint32_t>(), stream);
  uint32_t n_fr = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), n);
  frontier_out.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
  launch_compact_frontier(d_self, head.as<uint32_t>(), gid.as<uint32_t>(), n, frontier_out.as<FrontierRec>(), stream);
  tables.push_back(std::move(t));
  return n_fr;
}

void Engine::check_params(const impg_gpu_params_t &p) {
  if (p.dfs && p.transitive) throw Error{IMPG_E_UNSUPPORTED, "query_transitive_dfs is not built yet"};
  if (p.store_cigar) throw Error{IMPG_E_UNSUPPORTED, "store_cigar (BEDPE/PAF output) is not built yet"};
  if (p.transitive && p.max_depth > 65535) throw Error{IMPG_E_INVALID, "max_depth is a u16 in the reference"};
}

// The batch driver.  d_ranges: device array of n ranges.  If `keep` is non-null
// every level's buffers are appended to it (full-results mode); otherwise the
// level scratch is reused.  count/cksum: optional device arrays [n] of u64.
void Engine::run(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
                 std::vector<std::unique_ptr<LevelBufs>> *keep, unsigned long long *d_count,
                 unsigned long long *d_cksum, impg_gpu_stats_t *st, DevBuf *self_out) {
  check_params(p);
  IMPG_HIP(hipSetDevice(ix.device));
  split_ok = n > 1;
  min_identity = p.min_identity;
  const DeviceIndexView &v = ix.view;
  ev_next = 0;
  timed.clear();
  tables.clear();
  IMPG_HIP(hipMemsetAsync(counters.p, 0, 64, stream));
  IMPG_HIP(hipMemsetAsync(acc_slots.p, 0, COUNT_BYTES, stream));
  hipEvent_t t0 = event(), t1 = event();
  IMPG_HIP(hipEventRecord(t0, stream));
  if (st) memset(st, 0, sizeof *st);
  const bool transitive = p.transitive != 0;

  DevBuf *cur = &frontier_a, *nxt = &frontier_b;
  uint32_t n_fr = 0;
  if (!transitive) {
    cur->reserve(std::max<size_t>((size_t)n * sizeof(FrontierRec), 256));
    launch_ranges_to_frontier(d_ranges, n, cur->as<FrontierRec>(), stream);
    n_fr = n;
  } else {
    DevBuf &self = self_out ? *self_out : self_scratch;
    self.reserve(std::max<size_t>((size_t)n * sizeof(FrontierRec), 256));
    n_fr = begin_transitive(v, d_ranges, n, p, self.as<FrontierRec>(), *cur);
  }

  uint32_t depth = 0;
  while (n_fr > 0 && (!transitive || p.max_depth == 0 || depth < p.max_depth)) {
    std::unique_ptr<LevelBufs> own;
    LevelBufs *L = &level_scratch;
    if (keep) {
      own = std::make_unique<LevelBufs>();
      L = own.get();
    }
    expand(v, cur->as<FrontierRec>(), n_fr, transitive, *L, st);
    L->n_frontier = n_fr;
    if (d_count || d_cksum) {
      HitArrays h{L->qid.as<uint32_t>(), L->qs.as<int32_t>(), L->qe.as<int32_t>(), L->ts.as<int32_t>(), L->te.as<int32_t>()};
      launch_hit_stats(cur->as<FrontierRec>(), L->pair_range.as<uint32_t>(), L->n_pairs, h,
                       transitive ? p.min_output_length : -1, d_count, d_cksum, stream);
    }
    if (st) st->levels += 1;
    const bool last = !transitive || (p.max_depth > 0 && depth + 1 >= p.max_depth);
    uint32_t n_next = 0;
    if (!last) n_next = update(v, cur->as<FrontierRec>(), *L, n, p, *nxt);
    if (keep) {
      // the level keeps its own copy of the frontier (qidx / target per pair)
      L->frontier.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
      IMPG_HIP(hipMemcpyAsync(L->frontier.p, cur->p, (size_t)n_fr * sizeof(FrontierRec), hipMemcpyDeviceToDevice, stream));
      keep->push_back(std::move(own));
    }
    if (last) break;
    std::swap(cur, nxt);
    n_fr = n_next;
    depth += 1;
  }
  IMPG_HIP(hipEventRecord(t1, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  uint64_t hc[3];
  IMPG_HIP(hipMemcpy(hc, counters.p, 24, hipMemcpyDeviceToHost));
  if (hc[2]) throw Error{IMPG_E_INVALID, "an alignment hit by the query has no CIGAR (missing cg:Z tag)"};
  hc[1] = read_slots(acc_slots);
  if (st) {
    st->projected = hc[1];
    float ms = 0;
    IMPG_HIP(hipEventElapsedTime(&ms, t0, t1));
    st->ms_total = ms;
    for (auto &te : timed) {
      IMPG_HIP(hipEventElapsedTime(&ms, te.a, te.b));
      if (te.kind == 0) st->ms_lookup += ms;
      else if (te.kind == 1) st->ms_project += ms;
      else st->ms_update += ms;
    }
  }
  last_projected = hc[1];
}

}  // namespace impg
