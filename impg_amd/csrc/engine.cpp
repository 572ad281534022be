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
#include <cstdio>
#include <cstdlib>
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
  for (DevBuf *b : {&ord_offsets, &ord_rows}) b->pool = &level_pool;  // (handed over to the caller's result)
  win_se.pool = &level_pool;  // (handed over to a kept fused level as its frontier: a pooled block like the level's others)
  counters.reserve(128);  // (words 0..7: the counters of a run; 8..11: the list lengths and work counters of the update)
  acc_slots.reserve(COUNT_BYTES);
  act_slots.reserve(COUNT_BYTES);
  IMPG_HIP(hipHostMalloc((void **)&h_counters, 64, hipHostMallocDefault));
  IMPG_HIP(hipHostMalloc((void **)&h_slots, COUNT_BYTES, hipHostMallocDefault));
}
bool Engine::run_small(const impg_gpu_index &ix, const impg_gpu_range_t *h_ranges, uint32_t n, const impg_gpu_params_t &p,
                       impg_gpu_results &res) {
  // (MultiImpg's plain query sorts its hits by five keys, multi_impg.rs:556-592: the general path does that)
  if (n == 0 || n > SMALL_RANGES || p.transitive || p.store_cigar || p.multi_impg || masked || subset_on || remote) return false;
  if (p.min_identity == p.min_identity && ix.lacks_identity_lines()) const_cast<impg_gpu_index &>(ix).ensure_identity_lines();
  // every candidate pair of a range is an entry of its target: the sum of those segments bounds the pairs, on the host
  if (ix.h_tgt_off.size() != (size_t)ix.view.n_seq + 1) return false;  // (no table to bound the grid with: the general path counts on the device)
  uint64_t bound = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t t = h_ranges[i].target_id;
    if (t + 1 < ix.h_tgt_off.size()) bound += ix.h_tgt_off[t + 1] - ix.h_tgt_off[t];
  }
  if (bound > SMALL_PAIRS) return false;
  const uint32_t B = (uint32_t)std::max<uint64_t>(bound, 1);
  IMPG_HIP(hipSetDevice(ix.device));
  const size_t need_out = SMALL_HEADER_BYTES + (size_t)SMALL_PAIRS * (sizeof(impg_gpu_interval_t) + 4);
  if (!small_in) IMPG_HIP(hipHostMalloc((void **)&small_in, SMALL_RANGES * sizeof(impg_gpu_range_t), hipHostMallocDefault));
  if (!small_out) {
    IMPG_HIP(hipHostMalloc((void **)&small_out, need_out, hipHostMallocMapped));
    IMPG_HIP(hipHostGetDevicePointer(&small_out_dev, small_out, 0));
    small_out_cap = need_out;
  }
  const DeviceIndexView &v = ix.view;
  min_identity = p.min_identity;
  store_cigar = false;
  multi = false;
  memcpy(small_in, h_ranges, (size_t)n * sizeof(impg_gpu_range_t));
  ranges_dev.reserve(std::max<size_t>((size_t)SMALL_RANGES * sizeof(impg_gpu_range_t), 256));
  frontier_a.reserve(std::max<size_t>((size_t)SMALL_RANGES * sizeof(FrontierRec), 256));
  cnt.reserve(SMALL_RANGES * 4); win.reserve(SMALL_RANGES * 16); pair_off.reserve(SMALL_RANGES * 4);
  wide_n.reserve(256); wide_list.reserve(SMALL_RANGES * 8);  // (the wide list + its overflow list)
  LevelBufs &L = level_scratch;
  const size_t pb = std::max<size_t>((size_t)B * 4, 256);
  L.pair_range.reserve(pb); pair_entry.reserve(pb); L.qid.reserve(pb); L.coords.reserve(4 * pb);
  IMPG_HIP(hipMemcpyAsync(ranges_dev.p, small_in, (size_t)n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, stream));
  IMPG_HIP(hipMemsetAsync(counters.p, 0, 64, stream));
  IMPG_HIP(hipMemsetAsync(acc_slots.p, 0, COUNT_BYTES, stream));
  const FrontierRec *fr = frontier_a.as<FrontierRec>();
  launch_ranges_to_frontier(ranges_dev.as<impg_gpu_range_t>(), n, frontier_a.as<FrontierRec>(), stream);
  launch_lookup_count(v, fr, n, false, nullptr, cnt.as<uint32_t>(), win.as<uint4>(), wide_n.as<uint32_t>(), wide_list.as<uint32_t>(), stream);
  uint32_t *d_total = reinterpret_cast<uint32_t *>(counters.as<uint64_t>() + 3);
  launch_small_scan(cnt.as<uint32_t>(), n, pair_off.as<uint32_t>(), d_total, stream);
  launch_lookup_emit(v, fr, n, false, pair_off.as<uint32_t>(), win.as<uint4>(), L.pair_range.as<uint32_t>(), pair_entry.as<uint32_t>(),
                     nullptr, nullptr, ProjList{nullptr, nullptr, nullptr}, wide_n.as<uint32_t>(), wide_list.as<uint32_t>(), stream);
  HitArrays h{L.qid.as<uint32_t>(), L.coords.as<int4>()};
  launch_project(v, fr, L.pair_range.as<uint32_t>(), pair_entry.as<uint32_t>(), B, false, h, acc_slots.as<unsigned long long>(),
                 (uint32_t *)(counters.as<uint64_t>() + 2), min_identity, nullptr, ProjList{nullptr, nullptr, nullptr}, stream, d_total);
  char *od = static_cast<char *>(small_out_dev);
  impg_gpu_interval_t *d_rows = reinterpret_cast<impg_gpu_interval_t *>(od + SMALL_HEADER_BYTES);
  uint32_t *d_rr = reinterpret_cast<uint32_t *>(od + SMALL_HEADER_BYTES + (size_t)SMALL_PAIRS * sizeof(impg_gpu_interval_t));
  launch_small_pack(fr, L.pair_range.as<uint32_t>(), d_total, B, h, (const uint32_t *)(counters.as<uint64_t>() + 2),
                    acc_slots.as<unsigned long long>(), od, d_rows, d_rr, stream);
  IMPG_HIP(hipStreamSynchronize(stream));
  struct Hdr { uint32_t n_pairs, err, p0, p1; unsigned long long accepted; };
  const Hdr *hd = reinterpret_cast<const Hdr *>(small_out);
  if (hd->err & 2) throw Error{IMPG_E_INVALID, "Projection resulted in negative query coordinates"};
  if (hd->err) throw Error{IMPG_E_INVALID, "an alignment hit by the query has no CIGAR (missing cg:Z tag)"};
  const uint32_t P = hd->n_pairs;
  const impg_gpu_interval_t *rows = reinterpret_cast<const impg_gpu_interval_t *>(small_out + SMALL_HEADER_BYTES);
  const uint32_t *rr = reinterpret_cast<const uint32_t *>(small_out + SMALL_HEADER_BYTES + (size_t)SMALL_PAIRS * sizeof(impg_gpu_interval_t));
  // slots are in range order, visit order inside a range: one pass builds the per-range lists, self interval first
  res.ranges.assign(h_ranges, h_ranges + n);
  res.offsets.assign((size_t)n + 1, 0);
  res.intervals.clear();
  res.intervals.reserve((size_t)P + n);
  uint32_t pcur = 0;
  for (uint32_t q = 0; q < n; q++) {
    const impg_gpu_range_t &r = h_ranges[q];
    res.intervals.push_back({r.target_id, r.start, r.end, r.target_id, r.start, r.end});  // impg.rs:1864-1880
    while (pcur < P && rr[pcur] == q) {
      if (rows[pcur].query_id != HIT_NONE) res.intervals.push_back(rows[pcur]);
      pcur++;
    }
    res.offsets[q + 1] = res.intervals.size();
  }
  res.has_cigar = false;
  res.projected = hd->accepted;
  last_projected = hd->accepted;
  return true;
}

bool Engine::walk_applicable(const impg_gpu_index &ix, uint32_t n, const impg_gpu_params_t &p) const {
  if (!walk_allowed || !n || !p.transitive || p.multi_impg || p.store_cigar || remote || ix.tp_mode) return false;
  if (ix.view.n_seq > 65536u) return false;  // (hit keys are sequence << 15 | slot in 32 bits)
  // (a mask with empty ranges gives self pieces that touch: the reference's stack merges them after the first pop, the
  // walk's per-sequence stack lists only where new pieces land -- the batch engine runs those)
  if (masked && mask_has_empty && p.dfs) return false;
  // DFS: always (its steps are single pops; the batch engine pays a launch sequence per pop round).  BFS: a small batch
  // with a depth limit of two or more goes to the grid form (walk_members workgroups per query share its last level
  // out); without one the two engines cost the same per call and the batch engine keeps it -- walk_kernel = 2 sends
  // every small BFS batch here (tests)
  if (p.dfs) return true;
  if (n > SMALL_RANGES) return false;
  if (walk_bfs) return true;
  // (under a mask of long lists the walk's update -- one CU, a thread per (sequence) group on a copy of the map's list in
  // the slab -- loses to the batch engine's wave-per-group kernels as soon as a second level is updated: measured on
  // the headline index with 190 ranges per sequence masked, `-m 2` 0.37 vs 0.70 ms, `-m 3` 1.28 vs 0.98 ms per call)
  if (masked && p.max_depth >= 3 && mask_ranges_total > 16ull * std::max<uint64_t>(1, mask_lists)) return false;
  // (the grid form, or nothing: a batch too big for two workgroups a query -- more than 32 ranges with four engines a handle --
  // costs the one-workgroup walk 2.6 ms where the batch engine takes 2.1)
  return p.max_depth >= 2 && walk_group_size(ix, n, p) >= 2;
}
uint32_t Engine::walk_group_size(const impg_gpu_index &ix, uint32_t n, const impg_gpu_params_t &p) const {
  if (p.dfs || p.max_depth < 2 || n > SMALL_RANGES || walk_members == 1) return 1;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix.device);
  // every workgroup of the launch must be resident (the members wait for each other): one 1024-thread workgroup per CU --
  // and up to max_engines callers of one handle launch side by side on their own streams (rayon workers calling the trait:
  // multi_impg.rs:518-530, partition.rs), whose workgroups the dispatcher may interleave: each launch takes at most its
  // share of the CUs, so that all of them fit together.  (A launch that still cannot become resident -- another process
  // on the device -- gives up after the spin limit and the batch engine answers.)
  // (asked of the runtime once, not assumed: a kernel that grew past one workgroup per CU -- or a device that keeps none
  // resident -- gets the one-workgroup form / the batch engine instead of members spinning for partners that cannot start)
  static const uint32_t per_cu = walk_grid_blocks_per_cu();
  if (!per_cu) return 1;
  const uint32_t share = std::max(1u, (uint32_t)cus * std::min(per_cu, 1u) / (uint32_t)std::max(1, ix.max_engines));
  const uint32_t fit = std::max(1u, share / n);
  return std::min({walk_members ? walk_members : 32u, fit, WALK_MAX_MEMBERS});
}
// the walk's slab shape: a BFS processes whole levels (16 waves per query), a DFS step is one popped range (one wave)
void Engine::walk_caps(bool wide, WalkArgs &a) {
  a.wcap = wide ? 16384u : 4096u;   // a BFS frontier; the pieces of one DFS pop
  a.hcap = wide ? 32767u : 4096u;
  a.vcap = wide ? 131072u : 131072u;  // visited ranges (lists that outgrow their place move and leave it behind)
  a.gcap = wide ? 262144u : 16384u;
  a.scap = wide ? 16u : 65536u;       // DFS stack records (incl. the pieces too deep to be explored, until they are popped)
}
uint64_t Engine::walk_workgroups(const impg_gpu_index &ix, bool wide) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix.device);
  return wide ? SMALL_RANGES : (uint64_t)cus * 4u * WALK_WAVES_PER_SIMD;  // (slabs of ~2.5 MB: a wave per query, latency-bound, wants every wave slot it can get)
}
void Engine::reserve_walk_slabs(const impg_gpu_index &ix, bool dfs_too) {
  size_t want = 0;
  for (int wide = 1; wide >= (dfs_too ? 0 : 1); wide--) {
    WalkArgs a;
    memset(&a, 0, sizeof a);
    walk_caps(wide != 0, a);
    want = std::max(want, walk_slab_bytes(ix.view.n_seq, wide != 0, a.wcap, a.hcap, a.vcap, a.gcap, a.scap) * (size_t)walk_workgroups(ix, wide != 0));
  }
  if (walk_members != 1) {  // the grid form: queries x members workgroups, at most this launch's share of the CUs (walk_group_size)
    WalkArgs a;
    memset(&a, 0, sizeof a);
    walk_caps(true, a);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix.device);
    const size_t share = std::max<size_t>(1, (size_t)cus / (size_t)std::max(1, ix.max_engines));
    want = std::max(want, walk_slab_bytes(ix.view.n_seq, true, a.wcap, a.hcap, a.vcap, a.gcap, a.scap) * std::max<size_t>(share, SMALL_RANGES));
  }
  if (walk_slabs.cap < want) walk_slabs.reserve(want);
  walk_ctr.reserve(256);
  walk_ctl.reserve((size_t)SMALL_RANGES * sizeof(WalkGroupCtl));
}
bool Engine::run_walk(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
                      unsigned long long *d_count, unsigned long long *d_cksum, impg_gpu_stats_t *st, WalkRows *rows, uint32_t *h_n_rows) {
  if (!walk_applicable(ix, n, p)) return false;
  if (p.min_identity == p.min_identity && ix.lacks_identity_lines()) const_cast<impg_gpu_index &>(ix).ensure_identity_lines();
  IMPG_HIP(hipSetDevice(ix.device));
  // a BFS processes whole levels: 16 waves per query; a DFS step is one popped range: one wave, cheap barriers
  const bool wide = !p.dfs;
  WalkArgs a;
  memset(&a, 0, sizeof a);
  walk_caps(wide, a);
  const size_t slab = walk_slab_bytes(ix.view.n_seq, wide, a.wcap, a.hcap, a.vcap, a.gcap, a.scap);
  const uint32_t members = walk_group_size(ix, n, p);
  const uint32_t n_wg = members > 1 ? n * members : (uint32_t)std::min<uint64_t>(n, walk_workgroups(ix, wide));
  if (walk_slabs.cap < slab * n_wg) walk_slabs.reserve(slab * n_wg);  // (never shrunk: option prewarm_walk reserves the largest shape)
  walk_ctr.reserve(256);
  ev_next = 0;
  timed.clear();
  IMPG_HIP(hipMemsetAsync(walk_ctr.p, 0, 16, stream));
  if (members > 1) {
    walk_ctl.reserve((size_t)n * sizeof(WalkGroupCtl));
    IMPG_HIP(hipMemsetAsync(walk_ctl.p, 0, (size_t)n * sizeof(WalkGroupCtl), stream));  // (every polled word, every call)
  }
  IMPG_HIP(hipMemsetAsync(counters.p, 0, 64, stream));
  IMPG_HIP(hipMemsetAsync(acc_slots.p, 0, COUNT_BYTES, stream));
  hipEvent_t t0 = event(), t1 = event();
  IMPG_HIP(hipEventRecord(t0, stream));
  a.v = ix.view;
  a.ranges = d_ranges;
  a.n_queries = n;
  a.dfs = p.dfs ? 1 : 0;
  a.max_depth = p.max_depth;
  a.min_transitive_len = p.min_transitive_len;
  a.mdbr = p.min_distance_between_ranges;
  a.min_output_length = p.min_output_length;
  a.min_identity = p.min_identity;
  bool ident = p.min_identity == p.min_identity;
  if (!ident && !ix.view.pfx) { ident = true; a.min_identity = 0.0; }  // (an index without prefix lines: launch_project does the same)
  a.subset_keep = subset_on ? subset_keep.as<uint8_t>() : nullptr;
  a.count = d_count;
  a.cksum = d_cksum;
  a.accepted = acc_slots.as<unsigned long long>();
  a.err_flag = (uint32_t *)(counters.as<uint64_t>() + 2);
  a.next_query = walk_ctr.as<uint32_t>();
  a.overflow = walk_ctr.as<uint32_t>() + 1;
  a.slabs = walk_slabs.as<char>();
  a.slab_bytes = slab;
  a.members = members;
  a.ctl = members > 1 ? walk_ctl.as<WalkGroupCtl>() : nullptr;
  if (masked) {
    a.mask_off = mask_off.as<uint32_t>();
    a.mask_ranges = mask_ranges.as<int2>();
    a.mask_init_len = mask_init_len.as<int32_t>();
    a.mask_touch_len = mask_touch_len.as<int32_t>();
  }
  if (rows) {
    a.rows = rows->rows.as<impg_gpu_interval_t>();
    a.row_base = rows->base.as<unsigned long long>();
    a.row_cap = rows->cap.as<uint32_t>();
    a.n_rows = rows->n_rows.as<uint32_t>();
  }
  const bool dbg = getenv("IMPG_WALK_DEBUG") != nullptr;
  DevBuf d_dbg;
  if (dbg) {
    d_dbg.reserve(256);
    IMPG_HIP(hipMemsetAsync(d_dbg.p, 0, 256, stream));
    a.dbg = d_dbg.as<unsigned long long>();
  }
  if (st) memset(st, 0, sizeof *st);
  launch_walk(a, n_wg, wide, ident, stream);
  uint32_t flags[2] = {0, 0};
  IMPG_HIP(hipMemcpyAsync(flags, walk_ctr.p, 8, hipMemcpyDeviceToHost, stream));
  if (rows && h_n_rows) IMPG_HIP(hipMemcpyAsync(h_n_rows, rows->n_rows.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));  // (under the same synchronisation)
  finish_run(st, t0, t1);  // (synchronises; raises the projection errors; fills st->projected / ms_total)
  if (dbg) {
    unsigned long long h[32];
    IMPG_HIP(hipMemcpy(h, d_dbg.p, 256, hipMemcpyDeviceToHost));
    fprintf(stderr, "[walk] Mclk: drop %.3f window %.3f count %.3f emit %.3f project %.3f sort %.3f groups %.3f (gap %.3f) pieces %.3f merge %.3f shared last level %.3f | pops %llu dropped %llu max stack %llu max pieces %llu max group hits %llu max group list %llu heads pass %.3f Mclk slowest group %.3f Mclk | fail q=%llu flag=%llu target=%llu n_seq=%llu nw=%llu npc=%llu vused=%llu\n",
            h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6, h[4] / 1e6, h[5] / 1e6, h[6] / 1e6, h[7] / 1e6, h[8] / 1e6, h[9] / 1e6, h[10] / 1e6, h[16], h[17], h[18], h[19], h[20], h[21], h[22] / 1e6, h[23] / 1e6,
            h[24], h[25], h[26], h[27], h[28], h[29], h[30]);
  }
  if (dbg)
    fprintf(stderr, "[walk] n=%u %s wg=%u members=%u slab=%.2f MB dfs=%d depth=%u overflow=0x%x (1 unknown target, 2 pairs per step, 4 visited pool, 8 piece scratch, 16 pieces, 32 stack / frontier, 128 hand-off timed out) %.3f ms\n",
            n, wide ? "wide" : "wave", n_wg, members, slab / 1048576.0, a.dfs, a.max_depth, flags[1], st ? st->ms_total : -1.0f);
  ix.walk_last_members = members;
  if (flags[1]) { ix.walk_fallbacks++; return false; }  // a query outgrew its slab: the batch engine runs the batch
  ix.walk_launches++;
  if (st) st->levels = p.max_depth;
  return true;
}

Engine::~Engine() {
  if (small_in) (void)hipHostFree(small_in);
  if (small_out) (void)hipHostFree(small_out);
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

void Engine::scan2(const uint32_t *in_a, uint32_t *out_a, const uint32_t *in_b, uint32_t *out_b, uint32_t n, uint64_t &total_a, uint64_t &total_b,
                   const uint32_t *d_extra, uint32_t *h_extra, uint32_t n_extra) {
  total_a = total_b = 0;
  if (n) {
    scan_tmp.reserve(scan_scratch_bytes(n));
    scan_tmp2.reserve(scan_scratch_bytes(n));
    launch_exclusive_scan(in_a, out_a, n, scan_tmp.as<unsigned long long>(), counters.as<unsigned long long>() + 12, stream);
    launch_exclusive_scan(in_b, out_b, n, scan_tmp2.as<unsigned long long>(), counters.as<unsigned long long>() + 13, stream);
    IMPG_HIP(hipMemcpyAsync(h_counters, counters.as<uint64_t>() + 12, 16, hipMemcpyDeviceToHost, stream));
  }
  if (n_extra) IMPG_HIP(hipMemcpyAsync(h_counters + 2, d_extra, (size_t)n_extra * 4, hipMemcpyDeviceToHost, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  if (n) { total_a = h_counters[0]; total_b = h_counters[1]; }
  if (n_extra) memcpy(h_extra, h_counters + 2, (size_t)n_extra * 4);
}

static HitArrays hit_arrays(LevelBufs &L, uint32_t n_pairs) {
  size_t b = std::max<size_t>((size_t)n_pairs * 4, 256);
  L.qid.reserve(L.qs_interleaved ? 2 * b : b); L.coords.reserve(4 * b);
  return HitArrays{L.qid.as<uint32_t>(), L.coords.as<int4>()};
}

// Lookup / projection order: the slots stay in the reference's order, but the count,
// emit and projection kernels take the ranges in the order of their (estimated)
// windows in the entry array, so that lanes and workgroups in flight together
// search the same blocks and gather neighbouring entries and CIGAR tiles (L1/L2
// hits instead of HBM lines).  Returns the permutation, or null when not worth it.
const uint32_t *Engine::lookup_order(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, const RecordBlocks *blocks) {
  if (!locality_min || (n_fr < locality_min && !blocks)) { keys_for = nullptr; return nullptr; }
  const size_t nb = (size_t)n_fr * 4;
  lo_key.reserve(nb); lo_key2.reserve(nb); lo_idx.reserve(nb); lo_perm.reserve(nb);
  // keys are positions in the entry array: only the bits below n_entries are sorted (plus the block bits above them)
  unsigned hi_bit = std::max(1u, bits_for((uint32_t)std::min<size_t>(v.n_entries, 0xFFFFFFFFull)));
  unsigned block_shift = 0;
  if (blocks) {
    const unsigned bb = std::max(1u, bits_for(blocks->n_blocks));
    if (hi_bit + bb > 32) return nullptr;  // no room for the block above the window bits: the caller keeps the reference's order
    block_shift = hi_bit;
    hi_bit += bb;
  }
  if (!blocks && keys_for && keys_for == (const void *)fr && keys_n == n_fr) {
    // (frontier_emit wrote them beside this frontier's records)
  } else
    launch_order_keys(v, fr, n_fr, lo_key.as<uint32_t>(), lo_idx.as<uint32_t>(), stream, blocks ? blocks->d_bounds : nullptr,
                      blocks ? blocks->n_blocks : 0u, block_shift);
  keys_for = nullptr;
  // (every bit of the key is sorted.  Round 5, measured: leaving the low 5 bits unsorted -- two radix passes instead of
  // three -- takes 0.35 ms off the lookup and puts 4.4 ms ON the projection (22.1 -> 26.5 ms: the entries kernel's waves
  // find their entry's places in shorter runs); 9 bits: 87 ms.)
  static const bool lib_sort = getenv("IMPG_LIB_SORT") && atoi(getenv("IMPG_LIB_SORT")) != 0;  // (A/B: the library's radix sort)
  if (lib_sort) {
    const size_t tb = sort_u32_scratch_bytes(n_fr);
    sort_tmp.reserve(tb);
    launch_sort_u32(sort_tmp.p, tb, lo_key.as<uint32_t>(), lo_key2.as<uint32_t>(), lo_idx.as<uint32_t>(), lo_perm.as<uint32_t>(), n_fr,
                    stream, 0, hi_bit);
  } else {  // (a key's value is its index: lo_idx is scratch here)
    sort_tmp.reserve(order_sort_scratch_bytes(n_fr));
    launch_order_sort(lo_key.as<uint32_t>(), lo_key2.as<uint32_t>(), lo_perm.as<uint32_t>(), lo_idx.as<uint32_t>(), n_fr, hi_bit, sort_tmp.p, stream);
  }
  return lo_perm.as<uint32_t>();
}
// After the count pass: offp[r] = first place of range r's pairs in that order, and room for slot_of[P].
void Engine::projection_offsets(const uint32_t *d_perm, uint32_t n_fr, const uint32_t *d_cnt, uint64_t P, const uint32_t *&d_offp,
                                ProjList &pl) {
  d_offp = nullptr;
  pl = ProjList{nullptr, nullptr, nullptr};
  if (!d_perm || !P) return;
  const size_t nb = (size_t)n_fr * 4;
  lo_cnt.reserve(nb); lo_off.reserve(nb); lo_offp.reserve(nb);
  const size_t pb = std::max<size_t>(P * 4, 256);
  slot_of.reserve(pb); proj_range.reserve(pb); proj_entry.reserve(pb);
  launch_gather_u32(d_cnt, d_perm, n_fr, lo_cnt.as<uint32_t>(), stream);
  scan(lo_cnt.as<uint32_t>(), lo_off.as<uint32_t>(), n_fr);
  launch_scatter_u32(lo_off.as<uint32_t>(), d_perm, n_fr, lo_offp.as<uint32_t>(), stream);
  pl = ProjList{slot_of.as<uint32_t>(), proj_range.as<uint32_t>(), proj_entry.as<uint32_t>()};
  d_offp = lo_offp.as<uint32_t>();
}

// lookup + projection of one frontier; fills L (pair_range, hit arrays), returns #pairs
uint64_t Engine::expand(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                        impg_gpu_stats_t *st, bool raw, const RecordBlocks *blocks) {
  hipEvent_t e0 = event(), e1 = event(), e2 = event();
  IMPG_HIP(hipEventRecord(e0, stream));
  cnt.reserve((size_t)n_fr * 4);
  win.reserve((size_t)n_fr * 16);
  pair_off.reserve((size_t)n_fr * 4);
  // MultiImpg steps are Impg::query calls (multi_impg.rs:520-530): closed overlap test, unclipped range
  if (multi) transitive = false;
  wide_n.reserve(256);
  wide_list.reserve(std::max<size_t>((size_t)n_fr * 8, 256));  // (the wide list + its overflow list behind it)
  const uint32_t *d_perm = lookup_order(v, fr, n_fr, blocks);
  // Counting runs under the lookup order (see free_slot_order): slot = place in that order.  The count pass leaves
  // its counts and windows at the lanes' places, one scan gives the run offsets and the total, and the emit pass
  // writes the two pair lists as one contiguous piece per wave -- no per-range scatter, no slot list.
  // (the owner side of a sharded hop may do the same when the order runs home rank by home rank: `blocks`)
  const bool by_place = free_slot_order && (!raw || blocks) && !multi && !store_cigar && d_perm;
  last_by_place = by_place;
  last_range_places = false;
  L.qs_interleaved = false;
  L.placed = false;
  expand_n_fr = n_fr;
  bool fused = by_place && fuse_final && emit_by_lanes(v) && !v.tp_mode;
  if (by_place) win_se.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
  const bool ordered = ordered_rows && !raw;
  if (ordered) { L.slot_ref.reserve(std::max<size_t>((size_t)n_fr * 4, 256)); ord_cnt.reserve(std::max<size_t>((size_t)n_fr * 4, 256)); }
  launch_lookup_count(v, fr, n_fr, transitive, d_perm, cnt.as<uint32_t>(), win.as<uint4>(), wide_n.as<uint32_t>(),
                      wide_list.as<uint32_t>(), stream, by_place, by_place ? win_se.as<FrontierRec>() : nullptr,
                      ordered && by_place ? ord_cnt.as<uint32_t>() : nullptr);
  uint64_t P = scan(cnt.as<uint32_t>(), pair_off.as<uint32_t>(), n_fr);
  if (P > pair_budget || P >= 0xFFFFFFF0ull) {
    if (split_ok) throw SplitBatch{};
    if (P >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 candidate pairs for a single range"};
  }
  L.n_pairs = (uint32_t)P;
  L.n_frontier = n_fr;
  bool direct = false;
  if (ordered) {
    ordered_level(fr, n_fr, L, !by_place, P);
    // the fused final level writes its rows itself where the entry-ordered kernel runs it (a dense level); any other level
    // keeps its slots, listed in visit order, and is placed when the walk is over (ordered_finish)
    direct = fused && !store_cigar && !(min_identity == min_identity) && project_entry_major(v, P, min_identity);
    if (!direct) fused = false;
  }
  L.pair_range.reserve(std::max<size_t>(P * 4, 256));  // (a direct level: only the wave-per-range emit of wide windows writes it)
  pair_entry.reserve(std::max<size_t>(P * 4, 256));
  ProjList pl{nullptr, nullptr, nullptr};
  WindowLists wlists{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, 0u, 0u};
  if (fused) {
    // the final level of a counting run: pairs by windows (WindowLists); only the windows too wide for a 64-bit mask are listed
    // (tile_first[]: only project_kernel needs it -- the kernels of a dense level search their own block's offsets)
    const bool staged = project_is_staged(v, P, !store_cigar && !(min_identity == min_identity));
    tile_first.reserve(((size_t)(P + 255) / 256 + 1) * 4);
    if (!staged) launch_tile_first(cnt.as<uint32_t>(), pair_off.as<uint32_t>(), n_fr, tile_first.as<uint32_t>(), stream);
    launch_lookup_emit(v, fr, n_fr, transitive, pair_off.as<uint32_t>(), win.as<uint4>(), L.pair_range.as<uint32_t>(),
                       pair_entry.as<uint32_t>(), d_perm, nullptr, pl, wide_n.as<uint32_t>(), wide_list.as<uint32_t>(), stream, true, true);
    wlists = WindowLists{tile_first.as<uint32_t>(), pair_off.as<uint32_t>(), win.as<uint4>(), win_se.as<FrontierRec>(), d_perm, n_fr,
                         fuse_need_ranges && !direct ? L.pair_range.as<uint32_t>() : nullptr, 1u, fuse_range_places ? 1u : 0u};
    last_range_places = fuse_need_ranges && fuse_range_places && !direct;
    if (project_entry_major(v, P, min_identity) && !store_cigar) {  // the slice list and the range blocks' place counters of its heavy blocks
      const uint32_t cap = project_entry_slice_cap(P), nb = project_entry_blocks(n_fr);
      ent_work.reserve(((size_t)cap + 1) * 4);
      ent_alloc.reserve(std::max<size_t>((size_t)nb * 4, 256));
      IMPG_HIP(hipMemsetAsync(ent_work.p, 0, 4, stream));
      IMPG_HIP(hipMemsetAsync(ent_alloc.p, 0, (size_t)nb * 4, stream));
      wlists.slice_work = ent_work.as<uint32_t>(); wlists.slice_alloc = ent_alloc.as<uint32_t>(); wlists.slice_cap = cap;
    }
    // (a kept fused level: query id and source of a slot as one {qid, place} pair in L.qid -- one store instead of two)
    L.qs_interleaved = last_range_places && project_entry_major(v, P, min_identity) && !(min_identity == min_identity) && !store_cigar && !getenv("IMPG_NO_QS");  // (IMPG_NO_QS: A/B)
    if (L.qs_interleaved) { wlists.masks |= 4u; wlists.range_out = nullptr; }
    if (direct) {
      // every level is counted: the ranges' first rows, then this level's rows straight from the projection kernel --
      // a slot's place in its record's run is the hit's visit position, one byte a pair from the lookup (emit_vpos)
      ordered_offsets();
      ord_dest.reserve(std::max<size_t>((size_t)n_fr * 4, 256));
      ord_vpos.reserve(std::max<size_t>(P + 256, 256));
      const bool by_visit = ordered_rows_by_visit();
      const OrdDestArgs od{win_se.as<FrontierRec>(), d_perm, L.slot_ref.as<uint32_t>(), ord_offsets.as<uint32_t>(), L.lvbase.as<uint32_t>(), ord_dest.as<uint32_t>()};
      launch_emit_vpos(v, n_fr, pair_off.as<uint32_t>(), win.as<uint4>(), ord_vpos.as<uint8_t>(), stream, by_visit, &od);
      wlists.ord = OrderedOut{ord_rows.as<impg_gpu_interval_t>(), ord_dest.as<uint32_t>(), ord_vpos.as<uint8_t>(), ord_min_len, by_visit ? 1u : 0u};
      L.placed = true;
    }
  } else if (by_place) {
    launch_lookup_emit(v, fr, n_fr, transitive, pair_off.as<uint32_t>(), win.as<uint4>(), L.pair_range.as<uint32_t>(),
                       pair_entry.as<uint32_t>(), d_perm, nullptr, pl, wide_n.as<uint32_t>(), wide_list.as<uint32_t>(), stream, true);
    // (which range owns which places, for the staged projection of a dense level)
    wlists = WindowLists{nullptr, pair_off.as<uint32_t>(), win.as<uint4>(), win_se.as<FrontierRec>(), d_perm, n_fr, nullptr, 0u, 0u};
  } else {
    const uint32_t *d_offp = nullptr;
    projection_offsets(d_perm, n_fr, cnt.as<uint32_t>(), P, d_offp, pl);
    // the slot-order entry list is only read by the slice materialisation and the five-key sort
    const bool entry_slots = store_cigar || multi || !pl.slot;
    launch_lookup_emit(v, fr, n_fr, transitive, pair_off.as<uint32_t>(), win.as<uint4>(), L.pair_range.as<uint32_t>(),
                       entry_slots ? pair_entry.as<uint32_t>() : nullptr, pl.slot ? d_perm : nullptr, d_offp, pl,
                       wide_n.as<uint32_t>(), wide_list.as<uint32_t>(), stream);
  }
  IMPG_HIP(hipEventRecord(e1, stream));
  HitArrays h = direct ? HitArrays{nullptr, nullptr} : hit_arrays(L, L.n_pairs);
  SliceArrays sl{nullptr, nullptr, nullptr, nullptr};
  if (store_cigar) {
    size_t b = std::max<size_t>((size_t)L.n_pairs * 4, 256);
    L.sl_a.reserve(b); L.sl_n.reserve(b); L.sl_off.reserve(b); L.sl_rem.reserve(b);
    sl = SliceArrays{L.sl_a.as<uint32_t>(), L.sl_n.as<uint32_t>(), L.sl_off.as<int32_t>(), L.sl_rem.as<int32_t>()};
  }
  launch_project(v, fr, L.pair_range.as<uint32_t>(), pair_entry.as<uint32_t>(), L.n_pairs, transitive, h,
                 acc_slots.as<unsigned long long>(), (uint32_t *)(counters.as<uint64_t>() + 2), min_identity,
                 store_cigar ? &sl : nullptr, pl, stream, nullptr, regroup_pairs, by_place ? &wlists : nullptr);
  if (!raw && !direct) {
    post_expand(fr, n_fr, L, pair_off.as<uint32_t>(), pair_entry.as<uint32_t>(), v.mrank, sl);
    h = HitArrays{L.qid.as<uint32_t>(), L.coords.as<int4>()};
    if (store_cigar) sl = SliceArrays{L.sl_a.as<uint32_t>(), L.sl_n.as<uint32_t>(), L.sl_off.as<int32_t>(), L.sl_rem.as<int32_t>()};
  }
  IMPG_HIP(hipEventRecord(e2, stream));
  if (store_cigar && L.n_pairs) {  // materialise the slices while pair_entry is still this level's
    cnt.reserve((size_t)L.n_pairs * 4);
    L.slice_pos.reserve((size_t)L.n_pairs * 4);
    launch_slice_counts(h, sl, L.n_pairs, cnt.as<uint32_t>(), stream);
    L.slice_total = scan(cnt.as<uint32_t>(), L.slice_pos.as<uint32_t>(), L.n_pairs);
    if (L.slice_total >= 0xFFFFFFF0ull) {
      if (split_ok) throw SplitBatch{};
      throw Error{IMPG_E_UNSUPPORTED, "CIGAR slices of one level exceed 2^32 ops"};
    }
    L.slice_pool.reserve(std::max<size_t>(L.slice_total * 4, 256));
    launch_slice_write(v, pair_entry.as<uint32_t>(), h, sl, L.n_pairs, L.slice_pos.as<uint32_t>(), L.slice_pool.as<uint32_t>(), stream);
  } else {
    L.slice_total = 0;
  }
  timed.push_back({e0, e1, 0});
  timed.push_back({e1, e2, 1});
  if (st) {
    st->pairs += P;
    st->frontier_ranges += n_fr;
    if (P) st->project_launches += 1;
  }
  return P;
}

// What a level's slots go through between the projection and their readers: the subset filter (a hit whose query
// sequence is neither the range's own target nor kept becomes an empty slot) and, under MultiImpg semantics, the
// five-key sort of every frontier record's slot run (multi_impg.rs:582-592).  tie_idx is permuted with the slots
// (it is pair_entry on one GPU: the slice materialisation reads it afterwards).
void Engine::post_expand(const FrontierRec *fr, uint32_t n_fr, LevelBufs &L, const uint32_t *d_pair_off, uint32_t *tie_idx,
                         const uint32_t *tie_rank, SliceArrays sl) {
  HitArrays h{L.qid.as<uint32_t>(), L.coords.as<int4>()};
  if (subset_on)
    launch_subset_filter(fr, L.pair_range.as<uint32_t>(), L.n_pairs, h.qid, subset_keep.as<uint8_t>(), cur_ranges, stream);
  if (multi && L.n_pairs) {
    const size_t b = std::max<size_t>((size_t)L.n_pairs * 4, 256);
    m_dest.reserve(b); m_qid.reserve(b); m_coords.reserve(4 * b); m_pe.reserve(b);
    launch_sort5(fr, n_fr, d_pair_off, L.n_pairs, h, tie_idx, tie_rank, m_dest.as<uint32_t>(), stream);
    HitArrays h2{m_qid.as<uint32_t>(), m_coords.as<int4>()};
    SliceArrays sl2{nullptr, nullptr, nullptr, nullptr};
    if (sl.a) {
      m_sa.reserve(b); m_sn.reserve(b); m_so.reserve(b); m_sr.reserve(b);
      sl2 = SliceArrays{m_sa.as<uint32_t>(), m_sn.as<uint32_t>(), m_so.as<int32_t>(), m_sr.as<int32_t>()};
    }
    launch_permute_slots(m_dest.as<uint32_t>(), L.n_pairs, h, h2, tie_idx, m_pe.as<uint32_t>(), sl, sl2, stream);
    L.qid.swap(m_qid); L.coords.swap(m_coords);
    if (tie_idx == pair_entry.as<uint32_t>()) pair_entry.swap(m_pe);
    if (sl.a) { L.sl_a.swap(m_sa); L.sl_n.swap(m_sn); L.sl_off.swap(m_so); L.sl_rem.swap(m_sr); }
  }
}

// ---- ordered rows placed by slot (kernels.hip "Ordered rows placed slot by slot") ----------------------------------------
// after a level's count pass: the records' slot counts in frontier order (the count pass scattered them there, or `cnt`
// itself when the lookup ran in frontier order) -> their exclusive scan; the queries' level bases; acc += the level's slots
void Engine::ordered_level(const FrontierRec *fr, uint32_t n_fr, LevelBufs &L, bool counts_by_range, uint64_t P) {
  const uint64_t total = scan(counts_by_range ? cnt.as<uint32_t>() : ord_cnt.as<uint32_t>(), L.slot_ref.as<uint32_t>(), n_fr);
  if (total != P) throw Error{IMPG_E_INVALID, "internal: a level's slot counts disagree between its two orders"};
  L.lvbase.reserve(std::max<size_t>((size_t)ord_n * 4, 256));
  launch_ord_level_bases(fr, n_fr, L.slot_ref.as<uint32_t>(), (uint32_t)P, ord_n, ord_acc.as<uint32_t>(), L.lvbase.as<uint32_t>(), stream);
}
void Engine::ordered_offsets() {
  if (ord_offsets_done) return;
  ord_offsets.reserve(((size_t)ord_n + 1) * 4);
  ord_total = scan(ord_acc.as<uint32_t>(), ord_offsets.as<uint32_t>(), ord_n + 1u);  // (acc[n] = 0: offsets[n] = the total)
  if (ord_total >= 0xFFFFFFF0ull) { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 result rows in one chunk: use smaller chunks (chunk_ranges)"}; }
  ord_rows.reserve(std::max<size_t>(ord_total * sizeof(impg_gpu_interval_t), 256));
  ord_offsets_done = true;
}
void Engine::ordered_finish(std::vector<std::unique_ptr<LevelBufs>> &levels) {
  hipEvent_t p0 = event(), p1 = event();
  IMPG_HIP(hipEventRecord(p0, stream));
  ordered_offsets();
  launch_ord_self_rows(ord_self, ord_ranges, ord_n, ord_offsets.as<uint32_t>(), ord_rows.as<impg_gpu_interval_t>(), stream);
  for (auto &Lp : levels) {
    LevelBufs &L = *Lp;
    if (L.placed || !L.n_pairs) continue;
    L.run_start.reserve(std::max<size_t>((size_t)L.n_frontier * 4, 256));
    launch_ord_run_heads(L.pair_range.as<uint32_t>(), L.n_pairs, L.run_start.as<uint32_t>(), stream);
    HitArrays h{L.qid.as<uint32_t>(), L.coords.as<int4>()};
    launch_ord_level_rows(L.frontier.as<FrontierRec>(), L.pair_range.as<uint32_t>(), L.n_pairs, h, L.run_start.as<uint32_t>(), L.slot_ref.as<uint32_t>(),
                          ord_offsets.as<uint32_t>(), L.lvbase.as<uint32_t>(), ord_min_len, ord_rows.as<impg_gpu_interval_t>(), stream);
    L.placed = true;
  }
  IMPG_HIP(hipEventRecord(p1, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  float ms = 0;
  IMPG_HIP(hipEventElapsedTime(&ms, p0, p1));
  ms_place = ms;
}

HopResult Engine::hop(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                      impg_gpu_stats_t *st, bool need_hits, bool need_rows, bool alive) {
  if (remote) return remote->hop(*this, v, fr, n_fr, transitive, L, st, need_hits, need_rows, alive);
  if (!alive) { L.n_pairs = 0; return HopResult{0, true}; }
  if (!n_fr) { L.n_pairs = 0; return HopResult{0, false}; }
  return HopResult{expand(v, fr, n_fr, transitive, L, st), false};
}

// visited update + next frontier (impg.rs:2471-2584).  Returns the next frontier size.
uint32_t Engine::update(const DeviceIndexView &v, const FrontierRec *fr, LevelBufs &L, uint32_t n_queries,
                        const impg_gpu_params_t &p, DevBuf &next_frontier) {
  hipEvent_t e0 = event(), e1 = event();
  IMPG_HIP(hipEventRecord(e0, stream));
  const uint32_t P = L.n_pairs;
  uint32_t n_next = 0;
  HitArrays h{L.qid.as<uint32_t>(), L.coords.as<int4>()};
  if (P) {
    const uint32_t P_slots = P;
    const bool want_flags = filter_covered != 0;
    svals.reserve((size_t)P * 8);
    // The hits that carry a key in the stable order by (query, hit sequence).  By segments (kernels.hip, seg_group_kernel:
    // a query's ranges run by run in frontier order, a counting sort by sequence inside the query) when the batch allows
    // it; with the library's radix sort otherwise.
    uint32_t seg_active = 0, seg_groups = 0;
    uint32_t *sg_run_start = nullptr, *sg_run_end = nullptr, *sg_q = nullptr, *sg_bins = nullptr;
    // (a query with more than a wave's worth of hits is cut into slices of its frontier ranges: seg_group_parts)
    uint32_t seg_parts = seg_group && !want_flags && !multi && seg_group_fits(v.n_seq) && L.n_frontier > 0 ? seg_group_parts(P, n_queries, v.n_seq) : 0u;
    if (seg_parts_force && seg_parts) {  // (forced, for tests: within what the unit index and the counters' buffer take)
      seg_parts = std::max(1u, std::min(seg_parts_force, 4096u));
      while (seg_parts > 1 && ((uint64_t)n_queries * seg_parts >= (1ull << 31) || seg_group_bins_bytes(n_queries, v.n_seq, seg_parts) > (1ull << 30))) seg_parts >>= 1;
    }
    bool by_segments = seg_parts != 0;
    if (by_segments) {
      const uint32_t n_fr = L.n_frontier;
      seg_run.reserve(std::max<size_t>((size_t)n_fr * 8, 256));
      // six words a query -- first / last frontier range, hits with a key, their offset, groups, their offset -- then
      // `unsorted` and the largest query's (slice's) hits
      seg_q.reserve(std::max<size_t>((size_t)n_queries * 24 + 256, 512));
      sg_run_start = seg_run.as<uint32_t>(); sg_run_end = sg_run_start + n_fr;
      sg_q = seg_q.as<uint32_t>();
      uint32_t *qfirst = sg_q, *qlast = qfirst + n_queries, *qact = qlast + n_queries, *qdst = qact + n_queries, *qgrp = qdst + n_queries,
               *gdst = qgrp + n_queries, *unsorted = gdst + n_queries;
      // (the runs from the offsets the lookup left on this device -- by place in its order, or by range -- when they are still
      // there: a hop over other ranks brings slots without them, store_cigar reuses the counts for its slices)
      const bool own_offsets = !remote && !store_cigar && L.n_frontier == expand_n_fr;
      launch_seg_bounds(fr, n_fr, n_queries, L.pair_range.as<uint32_t>(), P, sg_run_start, sg_run_end, qfirst, qlast, unsorted, stream,
                        own_offsets && last_by_place ? lo_perm.as<uint32_t>() : nullptr, own_offsets ? pair_off.as<uint32_t>() : nullptr,
                        own_offsets ? cnt.as<uint32_t>() : nullptr);
      for (int attempt = 0; attempt < 2 && by_segments; attempt++) {
        const size_t bb = seg_group_bins_bytes(n_queries, v.n_seq, seg_parts);
        sg_bins = nullptr;
        if (seg_parts > 1 || bb <= (512ull << 20)) { seg_bins.reserve(std::max<size_t>(bb, 256)); sg_bins = seg_bins.as<uint32_t>(); }  // (kept for the place pass)
        if (seg_parts > 1) seg_tot.reserve(std::max<size_t>(seg_group_bins_bytes(n_queries, v.n_seq, 1), 256));
        launch_seg_group(true, fr, qfirst, qlast, sg_run_start, sg_run_end, h, n_queries, v.n_seq, qact, qdst, qgrp, gdst, nullptr, nullptr, nullptr, sg_bins,
                         stream, seg_parts, seg_tot.as<uint32_t>());
        // both scans and the kernels' two flags behind one synchronisation (the place pass runs below, once the groups' arrays exist)
        uint64_t ta = 0, tg = 0;
        uint32_t bad[2] = {0, 0};
        scan2(qact, qdst, qgrp, gdst, n_queries, ta, tg, unsorted, bad, 2);
        seg_active = (uint32_t)ta;
        if (bad[0]) by_segments = false;  // a frontier that is not sorted by query: the library sort below
        else if (bad[1]) {  // one huge query among small ones (bad[1] = its hits): once more in slices cut for it, or the library sort
          const uint32_t again = seg_parts == 1 && attempt == 0 ? seg_group_parts(P, n_queries, v.n_seq, bad[1]) : 0u;
          if (again > 1) { seg_parts = again; IMPG_HIP(hipMemsetAsync(unsorted, 0, 8, stream)); if (seg_stats) seg_stats[1]++; }
          else by_segments = false;
        } else { seg_groups = (uint32_t)tg; break; }
      }
      if (by_segments && seg_parts > 1 && seg_stats) seg_stats[0]++;
    }
    if (!by_segments) {
      if (seg_stats) seg_stats[2]++;
      keys.reserve((size_t)P * 8); skeys.reserve((size_t)P * 8); vals.reserve((size_t)P * 8);
      IMPG_HIP(hipMemsetAsync(act_slots.p, 0, COUNT_BYTES, stream));
      launch_update_keys(fr, L.pair_range.as<uint32_t>(), P, h, keys.as<unsigned long long>(), vals.as<unsigned long long>(),
                         act_slots.as<unsigned long long>(), stream);
      size_t tb = sort_u64v_scratch_bytes(P);
      sort_tmp.reserve(tb);
      // key = qidx << 32 | sequence id (all ones for a hit without a key): the bits between the sequence id and
      // the query index are zero, so two stable sorts -- the sequence-id bits, then the query bits plus the one
      // above them that only the all-ones keys have -- order the keys in 4 radix passes instead of 6
      const unsigned sbits = std::max(1u, bits_for(v.n_seq)), qbits = std::max(1u, bits_for(n_queries));
      launch_sort_u64v(sort_tmp.p, tb, keys.as<unsigned long long>(), skeys.as<unsigned long long>(), vals.as<unsigned long long>(),
                       svals.as<unsigned long long>(), P, stream, sbits, 0);
      launch_sort_u64v(sort_tmp.p, tb, skeys.as<unsigned long long>(), keys.as<unsigned long long>(), svals.as<unsigned long long>(),
                       vals.as<unsigned long long>(), P, stream, std::min(64u, 32 + qbits + 1), 32);
      keys.swap(skeys);
      vals.swap(svals);
    }
    // (from here on P = the entries of the sorted arrays: all slots after the library sort, whose keyless hits sort last;
    // only the hits that carry a key after the segment form)
    const uint32_t P = by_segments ? seg_active : P_slots;
    // the groups: per-tile head counts + their scan + a fill pass; the per-hit head flags and group ids only exist for
    // the covered-hit filter, which reads them
    uint32_t n_groups;
    if (by_segments) n_groups = seg_groups;
    else if (want_flags) {
      head.reserve((size_t)P * 4); gid.reserve((size_t)P * 4);
      launch_group_heads(skeys.as<unsigned long long>(), P, head.as<uint32_t>(), stream);
      n_groups = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), P);
    } else {
      const uint32_t nt = group_tiles(P);
      head.reserve((size_t)nt * 4); gid.reserve((size_t)nt * 4);
      launch_group_count(skeys.as<unsigned long long>(), P, head.as<uint32_t>(), stream);
      n_groups = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), nt);
    }
    if (n_groups) {
      auto vt = std::make_unique<VisitedStore>(&table_pool);
      vt->keys.reserve((size_t)n_groups * 8);
      gstart.reserve((size_t)n_groups * 4);
      if (by_segments) {  // the hits to their places, and every query's groups (start, key) straight from its sequence counters
        uint32_t *qfirst = sg_q, *qlast = qfirst + n_queries, *qact = qlast + n_queries, *qdst = qact + n_queries, *qgrp = qdst + n_queries,
                 *gdst = qgrp + n_queries;
        launch_seg_group(false, fr, qfirst, qlast, sg_run_start, sg_run_end, h, n_queries, v.n_seq, qact, qdst, qgrp, gdst, gstart.as<uint32_t>(),
                         vt->keys.as<unsigned long long>(), svals.as<unsigned long long>(), sg_bins, stream, seg_parts, seg_tot.as<uint32_t>());
      } else if (want_flags)
        launch_group_scatter(skeys.as<unsigned long long>(), P, head.as<uint32_t>(), gid.as<uint32_t>(), gstart.as<uint32_t>(),
                             vt->keys.as<unsigned long long>(), stream);
      else
        launch_group_fill(skeys.as<unsigned long long>(), P, gid.as<uint32_t>(), gstart.as<uint32_t>(), vt->keys.as<unsigned long long>(), stream);
      const uint32_t n_active = by_segments ? seg_active : (uint32_t)read_slots(act_slots);  // hits that carry a (query, sequence) key
      glen.reserve((size_t)n_groups * 4); old_src.reserve((size_t)n_groups * 8);
      cap.reserve((size_t)n_groups * 4); pcap.reserve((size_t)n_groups * 4);
      VisitedTables tabs = tables_view();
      launch_group_prepare(tabs, vt->keys.as<unsigned long long>(), gstart.as<uint32_t>(), n_groups, n_active,
                           glen.as<uint32_t>(), old_src.as<const int2 *>(), cap.as<uint32_t>(),
                           pcap.as<uint32_t>(), stream);
      if (getenv("IMPG_DEBUG_GROUPS")) {  // histogram of group sizes (tuning aid)
        std::vector<uint32_t> hc(n_groups), hp(n_groups), hl(n_groups);
        IMPG_HIP(hipStreamSynchronize(stream));
        IMPG_HIP(hipMemcpy(hc.data(), cap.p, (size_t)n_groups * 4, hipMemcpyDeviceToHost));
        IMPG_HIP(hipMemcpy(hp.data(), pcap.p, (size_t)n_groups * 4, hipMemcpyDeviceToHost));
        IMPG_HIP(hipMemcpy(hl.data(), glen.p, (size_t)n_groups * 4, hipMemcpyDeviceToHost));
        uint64_t b[8] = {0}, hits[8] = {0};
        const uint32_t edge[8] = {16, 64, 512, 1024, 2048, 4096, 16384, 0xFFFFFFFFu};
        uint32_t mx = 0, mxl = 0;
        for (uint32_t g2 = 0; g2 < n_groups; g2++) {
          int k = 0;
          while (hc[g2] > edge[k]) k++;
          b[k]++; hits[k] += hl[g2];
          mx = std::max(mx, hc[g2]); mxl = std::max(mxl, hl[g2]);
        }
        {  // the short end, which the lane-per-group kernel lives on: groups by cap (old + hits) and by hits, 1..20
          uint64_t bc[21] = {0}, bh[21] = {0};
          for (uint32_t g2 = 0; g2 < n_groups; g2++) { bc[std::min(hc[g2], 20u)]++; bh[std::min(hl[g2], 20u)]++; }
          fprintf(stderr, "[groups] by cap 1..20+:");
          for (int k2 = 1; k2 <= 20; k2++) fprintf(stderr, " %llu", (unsigned long long)bc[k2]);
          fprintf(stderr, "\n[groups] by hits 1..20+:");
          for (int k2 = 1; k2 <= 20; k2++) fprintf(stderr, " %llu", (unsigned long long)bh[k2]);
          fprintf(stderr, "\n");
        }
        fprintf(stderr, "[groups] n=%u P=%u maxcap=%u maxhits=%u | cap<=16:%llu(%llu) 64:%llu(%llu) 512:%llu(%llu) 1k:%llu(%llu) 2k:%llu(%llu) 4k:%llu(%llu) 16k:%llu(%llu) more:%llu(%llu)\n",
                n_groups, P, mx, mxl, (unsigned long long)b[0], (unsigned long long)hits[0], (unsigned long long)b[1], (unsigned long long)hits[1],
                (unsigned long long)b[2], (unsigned long long)hits[2], (unsigned long long)b[3], (unsigned long long)hits[3],
                (unsigned long long)b[4], (unsigned long long)hits[4], (unsigned long long)b[5], (unsigned long long)hits[5],
                (unsigned long long)b[6], (unsigned long long)hits[6], (unsigned long long)b[7], (unsigned long long)hits[7]);
      }
      // Hits that the group's list, as this level found it, already covers are dropped before the replay (they would
      // change nothing): worth its three passes when groups are long and lists exist, i.e. from the second update of
      // a walk on -- the deep levels of a saturating closure are almost all such hits.
      const unsigned long long *replay_vals = svals.as<unsigned long long>();
      const bool filter = filter_covered == 1 || (filter_covered == 2 && tables.size() >= 2 && (uint64_t)n_active >= 8ull * n_groups);
      if (filter && n_active) {
        uint32_t *keep = keys.as<uint32_t>(), *kpos = keep + P;  // (the unsorted key / value buffers are free again)
        launch_covered_flags(svals.as<unsigned long long>(), head.as<uint32_t>(), gid.as<uint32_t>(), vt->keys.as<unsigned long long>(),
                             old_src.as<const int2 *>(), cap.as<uint32_t>(), glen.as<uint32_t>(), masked ? mask_touch_len.as<int32_t>() : v.seq_len,
                             n_active, keep, stream);
        const uint32_t n_kept = (uint32_t)scan(keep, kpos, n_active);
        launch_covered_compact(svals.as<unsigned long long>(), keep, kpos, n_active, vals.as<unsigned long long>(), n_kept, n_groups,
                               gstart.as<uint32_t>(), glen.as<uint32_t>(), cap.as<uint32_t>(), pcap.as<uint32_t>(), stream);
        replay_vals = vals.as<unsigned long long>();
        covered_dropped += n_active - n_kept;
      }
      vt->off.reserve((size_t)n_groups * 4);
      poff.reserve((size_t)n_groups * 4);
      uint64_t cap_total = 0, pcap_total = 0;
      scan2(cap.as<uint32_t>(), vt->off.as<uint32_t>(), pcap.as<uint32_t>(), poff.as<uint32_t>(), n_groups, cap_total, pcap_total);
      if (cap_total >= 0xFFFFFFF0ull || pcap_total >= 0xFFFFFFF0ull)
        { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "visited sets exceed 2^32 ranges"}; }
      vt->ranges.reserve(std::max<size_t>(cap_total * 8, 256));
      vt->len.reserve((size_t)n_groups * 4);
      pieces.reserve(std::max<size_t>(pcap_total * 8, 256));
      n_pieces.reserve((size_t)n_groups * 4);
      foff.reserve((size_t)n_groups * 4);
      big_list.reserve((size_t)n_groups * 12);  // three lists: [small from the front | large from the back], [tiny], [mid] (big_groups_kernel)
      launch_visited_update(replay_vals, masked ? mask_touch_len.as<int32_t>() : v.seq_len, vt->keys.as<unsigned long long>(),
                            gstart.as<uint32_t>(), glen.as<uint32_t>(), old_src.as<const int2 *>(),
                            vt->off.as<uint32_t>(), poff.as<uint32_t>(), n_groups, p.min_transitive_len,
                            p.min_distance_between_ranges, vt->ranges.as<int2>(), vt->len.as<uint32_t>(),
                            pieces.as<int2>(), n_pieces.as<uint32_t>(), cap.as<uint32_t>(), pcap.as<uint32_t>(), big_list.as<uint32_t>(),
                            (uint32_t *)(counters.as<uint64_t>() + 8), stream);
      uint64_t nn = scan(n_pieces.as<uint32_t>(), foff.as<uint32_t>(), n_groups);
      if (nn >= 0xFFFFFFF0ull) { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "frontier exceeds 2^32 ranges"}; }
      n_next = (uint32_t)nn;
      next_frontier.reserve(std::max<size_t>((size_t)n_next * sizeof(FrontierRec), 256));
      // (the next level's lookup-order keys with the records, when that level will be looked up in that order on this device)
      const bool keys_too = !remote && locality_min && n_next >= locality_min;
      if (keys_too) { lo_key.reserve((size_t)n_next * 4); lo_idx.reserve((size_t)n_next * 4); }
      launch_frontier_emit(vt->keys.as<unsigned long long>(), poff.as<uint32_t>(), n_pieces.as<uint32_t>(),
                           foff.as<uint32_t>(), n_groups, pieces.as<int2>(), next_frontier.as<FrontierRec>(), stream,
                           keys_too ? &v : nullptr, lo_key.as<uint32_t>(), lo_idx.as<uint32_t>());
      keys_for = keys_too ? next_frontier.p : nullptr;
      keys_n = n_next;
      vt->n_groups = n_groups;
      index_table(*vt);
      tables.push_back(std::move(vt));
      if (tables.size() + 2 >= (size_t)MAX_VISITED_TABLES) compact_tables();
    }
  }
  IMPG_HIP(hipEventRecord(e1, stream));
  timed.push_back({e0, e1, 2});
  return n_next;
}

void Engine::index_table(VisitedStore &t) {
  t.qoff.reserve(((size_t)table_queries + 1) * 4);
  launch_table_qoff(t.keys.as<unsigned long long>(), t.n_groups, table_queries, t.qoff.as<uint32_t>(), stream);
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
    t.t[i].qoff = tables[i]->qoff.as<uint32_t>();
    t.t[i].n_groups = tables[i]->n_groups;
  }
  if (masked) {
    t.mask_off = mask_off.as<uint32_t>();
    t.mask_ranges = mask_ranges.as<int2>();
  }
  return t;
}

uint32_t Engine::begin_transitive(const DeviceIndexView &v, const impg_gpu_range_t *d_ranges, uint32_t n,
                                  const impg_gpu_params_t &p, FrontierRec *d_self, DevBuf &frontier_out) {
  tables.clear();
  auto t = std::make_unique<VisitedStore>(&table_pool);
  t->keys.reserve((size_t)n * 8); t->off.reserve((size_t)n * 4); t->len.reserve((size_t)n * 4);
  t->ranges.reserve((size_t)n * 8);
  t->n_groups = n;
  head.reserve((size_t)n * 4); gid.reserve((size_t)n * 4);
  launch_visited_init(d_ranges, n, v.seq_len, v.n_seq, p.min_transitive_len, t->keys.as<unsigned long long>(),
                      t->off.as<uint32_t>(), t->len.as<uint32_t>(), t->ranges.as<int2>(), d_self, head.as<uint32_t>(), stream);
  uint32_t n_fr = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), n);
  frontier_out.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
  launch_compact_frontier(d_self, head.as<uint32_t>(), gid.as<uint32_t>(), n, frontier_out.as<FrontierRec>(), stream);
  table_queries = n;
  index_table(*t);
  tables.push_back(std::move(t));
  return n_fr;
}

// level -1 under masked_regions: table 0 holds every query's (query, target) list = its target's mask list with
// the input range inserted; `self` receives the pieces (CSR in self_off / n_self), the long ones open the frontier
uint32_t Engine::begin_transitive_masked(const DeviceIndexView &v, const impg_gpu_range_t *d_ranges, uint32_t n,
                                         const impg_gpu_params_t &p, DevBuf &self, DevBuf &frontier_out) {
  tables.clear();
  auto t = std::make_unique<VisitedStore>(&table_pool);
  t->keys.reserve((size_t)n * 8); t->off.reserve((size_t)n * 4 + 4); t->len.reserve((size_t)n * 4);
  t->n_groups = n;
  cap.reserve((size_t)n * 4);
  launch_mask_caps(d_ranges, n, mask_off.as<uint32_t>(), v.n_seq, cap.as<uint32_t>(), stream);
  const uint64_t total = scan(cap.as<uint32_t>(), t->off.as<uint32_t>(), n);
  if (total >= 0xFFFFFFF0ull) { if (split_ok) throw SplitBatch{}; throw Error{IMPG_E_UNSUPPORTED, "visited sets exceed 2^32 ranges"}; }
  t->ranges.reserve(std::max<size_t>(total * 8, 256));
  pieces.reserve(std::max<size_t>(total * 8, 256));
  n_pieces.reserve((size_t)n * 4); head.reserve((size_t)n * 4); gid.reserve((size_t)n * 4);
  self_off.reserve((size_t)n * 4 + 4);
  launch_visited_init_masked(d_ranges, n, mask_init_len.as<int32_t>(), v.n_seq, mask_off.as<uint32_t>(), mask_ranges.as<int2>(),
                             p.min_transitive_len, t->off.as<uint32_t>(), t->keys.as<unsigned long long>(),
                             t->len.as<uint32_t>(), t->ranges.as<int2>(), pieces.as<int2>(), n_pieces.as<uint32_t>(),
                             head.as<uint32_t>(), stream);
  n_self = scan(n_pieces.as<uint32_t>(), self_off.as<uint32_t>(), n);
  const uint32_t n_fr = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), n);
  self.reserve(std::max<size_t>((size_t)n_self * sizeof(FrontierRec), 256));
  frontier_out.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
  launch_masked_self_emit(d_ranges, n, t->off.as<uint32_t>(), n_pieces.as<uint32_t>(), self_off.as<uint32_t>(),
                          gid.as<uint32_t>(), p.min_transitive_len, pieces.as<int2>(), self.as<FrontierRec>(),
                          frontier_out.as<FrontierRec>(), stream);
  table_queries = n;
  index_table(*t);
  tables.push_back(std::move(t));
  return n_fr;
}

void Engine::check_params(const impg_gpu_params_t &p) {
  if (p.transitive && p.max_depth > 65535) throw Error{IMPG_E_INVALID, "max_depth is a u16 in the reference"};
}

// The batch driver.  d_ranges: device array of n ranges.  If `keep` is non-null
// every level's buffers are appended to it (full-results mode); otherwise the
// level scratch is reused.  count/cksum: optional device arrays [n] of u64.
void Engine::run(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
                 std::vector<std::unique_ptr<LevelBufs>> *keep, unsigned long long *d_count,
                 unsigned long long *d_cksum, impg_gpu_stats_t *st, DevBuf *self_out) {
  check_params(p);
  // (the identity filter on an index built without its identity lines: they are built now, once)
  if (p.min_identity == p.min_identity && ix.lacks_identity_lines()) const_cast<impg_gpu_index &>(ix).ensure_identity_lines();
  IMPG_HIP(hipSetDevice(ix.device));
  split_ok = n > 1 && !remote;  // (ranks of a sharded batch stay in lock step: no re-splitting)
  min_identity = p.min_identity;
  store_cigar = p.store_cigar != 0 && keep != nullptr;  // slices are only materialised for full results
  if (store_cigar && ix.tp_mode)
    throw Error{IMPG_E_UNSUPPORTED, "store_cigar is not offered on a tracepoint index (the approximate mode has no CIGAR to slice)"};
  multi = p.multi_impg != 0;
  // (kept levels too: the device-side row placement, rows_device.hip, only needs a record's slots to be one run)
  free_slot_order = free_slots_allowed;
  const DeviceIndexView &v = ix.view;
  cur_ranges = d_ranges;
  keys_for = nullptr;
  ev_next = 0;
  timed.clear();
  tables.clear();
  IMPG_HIP(hipMemsetAsync(counters.p, 0, 64, stream));
  IMPG_HIP(hipMemsetAsync(acc_slots.p, 0, COUNT_BYTES, stream));
  hipEvent_t t0 = event(), t1 = event();
  IMPG_HIP(hipEventRecord(t0, stream));
  if (st) memset(st, 0, sizeof *st);
  const bool transitive = p.transitive != 0;
  // small transitive batches and DFS batches: the per-query walk, one launch (walk_device.inc); counts only here --
  // callers that want rows ask for them directly (capi.cpp)
  if (!keep && !self_out && walk_applicable(ix, n, p)) {
    ev_next = 0;
    if (run_walk(ix, d_ranges, n, p, d_count, d_cksum, st, nullptr)) return;
    // (not taken after all: start over on the batch path)
    if (d_count) IMPG_HIP(hipMemsetAsync(d_count, 0, (size_t)n * 8, stream));
    if (d_cksum) IMPG_HIP(hipMemsetAsync(d_cksum, 0, (size_t)n * 8, stream));
    ev_next = 0;
    timed.clear();
    IMPG_HIP(hipMemsetAsync(counters.p, 0, 64, stream));
    IMPG_HIP(hipMemsetAsync(acc_slots.p, 0, COUNT_BYTES, stream));
    t0 = event(); t1 = event();
    IMPG_HIP(hipEventRecord(t0, stream));
    if (st) memset(st, 0, sizeof *st);
  }
  if (transitive && (p.dfs || multi)) {  // one worklist pop at a time per query
    ev_next = 0;
    run_dfs(ix, d_ranges, n, p, keep, d_count, d_cksum, st, self_out);
    return;
  }

  DevBuf *cur = &frontier_a, *nxt = &frontier_b;
  uint32_t n_fr = 0;
  cur->reserve(std::max<size_t>((size_t)n * sizeof(FrontierRec), 256));
  if (!n) {  // (a rank of a sharded batch that has no ranges of its own still takes part in every hop)
  } else if (!transitive) {
    launch_ranges_to_frontier(d_ranges, n, cur->as<FrontierRec>(), stream);
    n_fr = n;
  } else {
    DevBuf &self = self_out ? *self_out : self_scratch;
    self.reserve(std::max<size_t>((size_t)n * sizeof(FrontierRec), 256));
    n_fr = masked ? begin_transitive_masked(v, d_ranges, n, p, self, *cur)
                  : begin_transitive(v, d_ranges, n, p, self.as<FrontierRec>(), *cur);
    ord_self = self.as<FrontierRec>();
  }
  if (ordered_rows) {
    if (!keep || masked || multi || store_cigar || remote) throw Error{IMPG_E_INVALID, "internal: ordered rows are placed for plain and BFS batches of one GPU"};
    ord_n = n;
    ord_ranges = d_ranges;
    if (!transitive) ord_self = nullptr;
    ord_min_len = transitive ? p.min_output_length : -1;
    ord_offsets_done = false;
    ord_total = 0;
    ms_place = 0;
    ord_acc.reserve(((size_t)n + 1) * 4);
    IMPG_HIP(hipMemsetAsync(ord_acc.p, 0, ((size_t)n + 1) * 4, stream));
    launch_ord_self_count(ord_self, d_ranges, n, ord_acc.as<uint32_t>(), stream);
  }

  uint32_t depth = 0;
  for (;;) {
    const bool alive = n_fr > 0 && (!transitive || p.max_depth == 0 || depth < p.max_depth);
    const bool last = !transitive || (p.max_depth > 0 && depth + 1 >= p.max_depth);
    std::unique_ptr<LevelBufs> own;
    LevelBufs *L = &level_scratch;
    if (keep) {
      own = std::make_unique<LevelBufs>(&level_pool);
      L = own.get();
    }
    const bool want_stats = d_count || d_cksum;
    // (kept levels too when their reader takes the slots in any order and finds a slot's frontier record through
    // pair_range -- the rows left in HBM, impg_gpu_query_batch_device's attributed layout: keep_any_order)
    fuse_final = fuse_allowed && last && (!keep || keep_any_order || ordered_rows) && !remote;
    // (per-range counts / checksums of a fused level cost two atomics per hit -- its slots are in entry order, a range's
    // are no run -- which a deep closure's final level, 10^4+ ranges a query, does not earn back: config 5 with counts
    // 2.8 s per 4 000 windows fused, 1.4 s not)
    if (want_stats && (uint64_t)n_fr > 8192ull * n) fuse_final = false;
    fuse_need_ranges = want_stats || subset_on || keep != nullptr;
    // (a kept fused level names a slot's range by its place in the lookup order -- no load in the kernel -- and keeps its
    // copy of the frontier in that order; the per-range statistics and the subset filter index the frontier itself)
    fuse_range_places = keep != nullptr && !want_stats && !subset_on;
    if (ordered_rows) fuse_need_ranges = fuse_range_places = false;  // (a fused level of ordered rows writes rows, nothing else)
    const HopResult hr = hop(v, cur->as<FrontierRec>(), alive ? n_fr : 0, transitive, *L, st, keep || want_stats || !last,
                             keep || d_cksum, alive);
    fuse_final = false;
    if (hr.all_dead) break;
    uint32_t n_next = 0;
    if (alive) {
      L->n_frontier = n_fr;
      if (want_stats) {
        HitArrays h{L->qid.as<uint32_t>(), L->coords.as<int4>()};
        rstat.reserve(std::max<size_t>((size_t)n_fr * 16, 256));
        launch_hit_stats(cur->as<FrontierRec>(), n_fr, L->pair_range.as<uint32_t>(), L->n_pairs, h,
                         transitive ? p.min_output_length : -1, false, rstat.as<unsigned long long>(), d_count, d_cksum, stream);
      }
      if (st) st->levels += 1;
      if (!last) n_next = update(v, cur->as<FrontierRec>(), *L, n, p, *nxt);
      if (keep) {
        // the level keeps its own copy of the frontier (qidx / target per pair)
        if (L->placed) {  // (its rows are written: nobody reads its slots)
        } else if (last_range_places && !remote) {
          // (pair_range holds places of the lookup order: the level's frontier in that order is what the count pass left
          // beside the windows -- handed over, not copied; win_se allocates anew at the next by-place level)
          L->frontier.adopt(win_se);
        } else {
          L->frontier.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
          IMPG_HIP(hipMemcpyAsync(L->frontier.p, cur->p, (size_t)n_fr * sizeof(FrontierRec), hipMemcpyDeviceToDevice, stream));
        }
        keep->push_back(std::move(own));
      }
    }
    if (last) break;  // (the depth is the same on every rank)
    std::swap(cur, nxt);
    n_fr = n_next;
    depth += 1;
  }
  if (ordered_rows) ordered_finish(*keep);
  finish_run(st, t0, t1);
}

void Engine::finish_run(impg_gpu_stats_t *st, hipEvent_t t0, hipEvent_t t1) {
  IMPG_HIP(hipEventRecord(t1, stream));
  IMPG_HIP(hipStreamSynchronize(stream));
  uint64_t hc[3];
  IMPG_HIP(hipMemcpy(hc, counters.p, 24, hipMemcpyDeviceToHost));
  if (hc[2] & 2) throw Error{IMPG_E_INVALID, "Projection resulted in negative query coordinates"};  // the reference panics (impg.rs:1509-1514)
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

// Fold every visited table into one: for each key the newest table's list wins
// (it is complete).  Keeps long transitive walks (DFS rounds, unlimited depth)
// within MAX_VISITED_TABLES.
void Engine::compact_tables() {
  if (tables.size() < 2) return;
  uint64_t G = 0;
  for (auto &t : tables) G += t->n_groups;
  if (G >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "visited tables exceed 2^32 groups"};
  const uint32_t g = (uint32_t)G;
  VisitedTables tv = tables_view();
  d_ckey.reserve((size_t)g * 8); d_ckey2.reserve((size_t)g * 8); d_src.reserve((size_t)g * 8); d_src2.reserve((size_t)g * 8);
  uint32_t off = 0;
  for (size_t t = 0; t < tables.size(); t++) {  // oldest first: a stable sort leaves the newest last within a key
    launch_compact_fill(tables[t]->keys.as<unsigned long long>(), tables[t]->n_groups, (uint32_t)t,
                        d_ckey.as<unsigned long long>() + off, d_src.as<unsigned long long>() + off, stream);
    off += tables[t]->n_groups;
  }
  size_t tb = sort_u64v_scratch_bytes(g);
  sort_tmp.reserve(tb);
  launch_sort_u64v(sort_tmp.p, tb, d_ckey.as<unsigned long long>(), d_ckey2.as<unsigned long long>(),
                   d_src.as<unsigned long long>(), d_src2.as<unsigned long long>(), g, stream);
  d_flag2.reserve((size_t)g * 4); d_pos2.reserve((size_t)g * 4);
  launch_compact_last(d_ckey2.as<unsigned long long>(), g, d_flag2.as<uint32_t>(), stream);
  const uint32_t g2 = (uint32_t)scan(d_flag2.as<uint32_t>(), d_pos2.as<uint32_t>(), g);
  auto nt = std::make_unique<VisitedStore>(&table_pool);
  nt->keys.reserve((size_t)g2 * 8); nt->off.reserve((size_t)g2 * 4); nt->len.reserve((size_t)g2 * 4);
  launch_compact_select(tv, d_ckey2.as<unsigned long long>(), d_src2.as<unsigned long long>(), g, d_flag2.as<uint32_t>(),
                        d_pos2.as<uint32_t>(), nt->keys.as<unsigned long long>(), d_src.as<unsigned long long>(),
                        nt->len.as<uint32_t>(), stream);
  const uint64_t total = scan(nt->len.as<uint32_t>(), nt->off.as<uint32_t>(), g2);
  if (total >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "visited sets exceed 2^32 ranges"};
  nt->ranges.reserve(std::max<size_t>(total * 8, 256));
  launch_compact_copy(tv, d_src.as<unsigned long long>(), nt->off.as<uint32_t>(), nt->len.as<uint32_t>(), g2,
                      nt->ranges.as<int2>(), stream);
  nt->n_groups = g2;
  index_table(*nt);
  IMPG_HIP(hipStreamSynchronize(stream));  // the old tables are read by the copy above
  tables.clear();
  tables.push_back(std::move(nt));
}

// query_transitive_dfs (impg.rs:2057-2309) for a whole batch, in rounds: every
// query pops the top of its own stack, the popped ranges are expanded together,
// the visited update runs for all of them, and the stacks are re-sorted and merged
// (impg.rs:2289-2304).  A query's pops happen in the reference's order; queries
// are independent, so interleaving them changes nothing.
void Engine::run_dfs(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
                     std::vector<std::unique_ptr<LevelBufs>> *keep, unsigned long long *d_count, unsigned long long *d_cksum,
                     impg_gpu_stats_t *st, DevBuf *self_out) {
  const DeviceIndexView &v = ix.view;
  hipEvent_t t0 = event(), t1 = event();
  IMPG_HIP(hipEventRecord(t0, stream));
  DevBuf &self = self_out ? *self_out : self_scratch;
  self.reserve(std::max<size_t>((size_t)n * sizeof(FrontierRec), 256));
  uint32_t n_stack = 0;
  if (n) n_stack = masked ? begin_transitive_masked(v, d_ranges, n, p, self, frontier_a)
                          : begin_transitive(v, d_ranges, n, p, self.as<FrontierRec>(), frontier_a);
  auto res4 = [&](DevBuf &k, DevBuf &s, DevBuf &e, DevBuf &d, size_t m) {
    k.reserve(std::max<size_t>(m * 8, 256)); s.reserve(std::max<size_t>(m * 4, 256));
    e.reserve(std::max<size_t>(m * 4, 256)); d.reserve(std::max<size_t>(m * 4, 256));
  };
  for (DevBuf *b : {&dk_a, &ds_a, &de_a, &dd_a, &dk_b, &ds_b, &de_b, &dd_b}) b->pool = &level_pool;  // (swapped with per-round buffers below)
  res4(dk_a, ds_a, de_a, dd_a, n_stack);
  d_popdepth.reserve(std::max<size_t>((size_t)n * 4, 256));
  launch_frontier_to_stack(frontier_a.as<FrontierRec>(), n_stack, nullptr, false, dk_a.as<unsigned long long>(),
                           ds_a.as<int32_t>(), de_a.as<int32_t>(), dd_a.as<uint32_t>(), stream);
  // current stack in the *_a buffers, sorted by (qidx, sequence, start)
  // (a mask with empty ranges gives first pieces that touch: the reference merges them at its first re-sort, which it
  // runs after every explored pop, pushes or not -- impg.rs:2289-2304 -- so the "nothing pushed" shortcut below waits
  // until one round has merged the stacks)
  bool unmerged = masked && mask_has_empty;
  for (;;) {
    const bool alive = n_stack > 0;
    // ---- pop the top of every query's stack -----------------------------------
    uint32_t n_fr = 0, n_keep = 0;
    frontier_b.reserve(256);
    if (alive) {
      head.reserve((size_t)n_stack * 4); gid.reserve((size_t)n_stack * 4);
      d_flag2.reserve((size_t)n_stack * 4); d_pos2.reserve((size_t)n_stack * 4);
      d_popsel.reserve(std::max<size_t>((size_t)n * 4, 256));
      launch_dfs_pop_flags(dk_a.as<unsigned long long>(), dd_a.as<uint32_t>(), n_stack, p.max_depth, multi && !p.dfs, head.as<uint32_t>(),
                           d_flag2.as<uint32_t>(), d_popdepth.as<uint32_t>(), d_popsel.as<uint32_t>(), n, stream);
      n_fr = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), n_stack);
      n_keep = (uint32_t)scan(d_flag2.as<uint32_t>(), d_pos2.as<uint32_t>(), n_stack);
      frontier_b.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
    }
    uint32_t n_pieces = 0;
    // remaining stack goes to the *_b buffers; the new pieces are appended behind it
    std::unique_ptr<LevelBufs> own;
    LevelBufs *L = &level_scratch;
    if (keep) { own = std::make_unique<LevelBufs>(&level_pool); L = own.get(); }
    if (alive) {
      res4(dk_b, ds_b, de_b, dd_b, (size_t)n_keep + 1);
      launch_dfs_pop_scatter(dk_a.as<unsigned long long>(), ds_a.as<int32_t>(), de_a.as<int32_t>(), dd_a.as<uint32_t>(), n_stack,
                             head.as<uint32_t>(), gid.as<uint32_t>(), d_flag2.as<uint32_t>(), d_pos2.as<uint32_t>(),
                             frontier_b.as<FrontierRec>(), dk_b.as<unsigned long long>(), ds_b.as<int32_t>(), de_b.as<int32_t>(),
                             dd_b.as<uint32_t>(), stream);
    }
    const HopResult hr = hop(v, frontier_b.as<FrontierRec>(), n_fr, true, *L, st, true, keep || d_cksum, alive);
    if (hr.all_dead) break;
    if (!alive) continue;
    if (n_fr) {
      L->n_frontier = n_fr;
      if (d_count || d_cksum) {
        HitArrays h{L->qid.as<uint32_t>(), L->coords.as<int4>()};
        rstat.reserve(std::max<size_t>((size_t)n_fr * 16, 256));
        launch_hit_stats(frontier_b.as<FrontierRec>(), n_fr, L->pair_range.as<uint32_t>(), L->n_pairs, h, p.min_output_length,
                         multi, rstat.as<unsigned long long>(), d_count, d_cksum, stream);
      }
      if (st) st->levels += 1;
      n_pieces = update(v, frontier_b.as<FrontierRec>(), *L, n, p, frontier_a);  // pieces, sorted by (qidx, seq, start)
      if (keep) {
        L->frontier.reserve(std::max<size_t>((size_t)n_fr * sizeof(FrontierRec), 256));
        IMPG_HIP(hipMemcpyAsync(L->frontier.p, frontier_b.p, (size_t)n_fr * sizeof(FrontierRec), hipMemcpyDeviceToDevice, stream));
        keep->push_back(std::move(own));
      }
    }
    const uint32_t m = n_keep + n_pieces;
    if (m == 0) { n_stack = 0; continue; }  // (one more hop call tells the other ranks, or ends the walk)
    if (n_pieces == 0 && !unmerged) {  // nothing pushed: the remaining stack is still sorted and merged
      dk_a.swap(dk_b); ds_a.swap(ds_b); de_a.swap(de_b); dd_a.swap(dd_b);
      n_stack = n_keep;
      continue;
    }
    if (m >= 0xFFFFFFF0u) throw Error{IMPG_E_UNSUPPORTED, "DFS stacks exceed 2^32 records"};
    // *_b buffers may have been sized for n_keep only: grow while keeping the kept part
    {
      DevBuf nk, ns, ne, nd;
      nk.pool = ns.pool = ne.pool = nd.pool = &level_pool;  // (new buffers every round: recycled, not hipMalloc'ed)
      res4(nk, ns, ne, nd, m);
      if (n_keep) {
        IMPG_HIP(hipMemcpyAsync(nk.p, dk_b.p, (size_t)n_keep * 8, hipMemcpyDeviceToDevice, stream));
        IMPG_HIP(hipMemcpyAsync(ns.p, ds_b.p, (size_t)n_keep * 4, hipMemcpyDeviceToDevice, stream));
        IMPG_HIP(hipMemcpyAsync(ne.p, de_b.p, (size_t)n_keep * 4, hipMemcpyDeviceToDevice, stream));
        IMPG_HIP(hipMemcpyAsync(nd.p, dd_b.p, (size_t)n_keep * 4, hipMemcpyDeviceToDevice, stream));
      }
      if (n_pieces)
        launch_frontier_to_stack(frontier_a.as<FrontierRec>(), n_pieces, d_popdepth.as<uint32_t>(), true,
                                 nk.as<unsigned long long>() + n_keep, ns.as<int32_t>() + n_keep, ne.as<int32_t>() + n_keep,
                                 nd.as<uint32_t>() + n_keep, stream);
      IMPG_HIP(hipStreamSynchronize(stream));
      dk_b.swap(nk); ds_b.swap(ns); de_b.swap(ne); dd_b.swap(nd);
    }
    // ---- stack.sort_by_key((id, start)) for every query at once: LSD, two stable radix sorts
    d_perm.reserve((size_t)m * 4); d_perm2.reserve((size_t)m * 4); d_k32.reserve((size_t)m * 4); d_k32b.reserve((size_t)m * 4);
    launch_iota(d_perm.as<uint32_t>(), m, stream);
    size_t tb = std::max(sort_u32_scratch_bytes(m), sort_pairs_scratch_bytes(m));
    sort_tmp.reserve(tb);
    launch_sort_u32(sort_tmp.p, tb, reinterpret_cast<const uint32_t *>(ds_b.as<int32_t>()), d_k32b.as<uint32_t>(),
                    d_perm.as<uint32_t>(), d_perm2.as<uint32_t>(), m, stream);
    d_ckey.reserve((size_t)m * 8); d_ckey2.reserve((size_t)m * 8);
    launch_gather_u64(dk_b.as<unsigned long long>(), d_perm2.as<uint32_t>(), m, d_ckey.as<unsigned long long>(), stream);
    launch_sort_pairs(sort_tmp.p, tb, d_ckey.as<unsigned long long>(), d_ckey2.as<unsigned long long>(), d_perm2.as<uint32_t>(),
                      d_perm.as<uint32_t>(), m, 64, stream);
    // ---- merge overlapping / contiguous neighbours of the same (query, sequence)
    head.reserve((size_t)m * 4); gid.reserve((size_t)m * 4);
    launch_run_heads(d_ckey2.as<unsigned long long>(), m, head.as<uint32_t>(), stream);
    const uint32_t n_groups = (uint32_t)scan(head.as<uint32_t>(), gid.as<uint32_t>(), m);
    gstart.reserve((size_t)n_groups * 4); cap.reserve((size_t)n_groups * 4); foff.reserve((size_t)n_groups * 4);
    launch_run_starts(head.as<uint32_t>(), gid.as<uint32_t>(), m, gstart.as<uint32_t>(), stream);
    DevBuf &sm = d_k32, &em = d_k32b, &dm = d_pos2;  // merged records, still at their run's offset
    sm.reserve((size_t)m * 4); em.reserve((size_t)m * 4); dm.reserve((size_t)m * 4);
    launch_dfs_merge(gstart.as<uint32_t>(), n_groups, m, d_perm.as<uint32_t>(), ds_b.as<int32_t>(), de_b.as<int32_t>(),
                     dd_b.as<uint32_t>(), sm.as<int32_t>(), em.as<int32_t>(), dm.as<uint32_t>(), cap.as<uint32_t>(), stream);
    const uint32_t n_new = (uint32_t)scan(cap.as<uint32_t>(), foff.as<uint32_t>(), n_groups);
    res4(dk_a, ds_a, de_a, dd_a, n_new);
    launch_dfs_compact(gstart.as<uint32_t>(), cap.as<uint32_t>(), foff.as<uint32_t>(), n_groups, d_ckey2.as<unsigned long long>(),
                       sm.as<int32_t>(), em.as<int32_t>(), dm.as<uint32_t>(), dk_a.as<unsigned long long>(), ds_a.as<int32_t>(),
                       de_a.as<int32_t>(), dd_a.as<uint32_t>(), stream);
    n_stack = n_new;
    unmerged = false;
  }
  finish_run(st, t0, t1);
}

}  // namespace impg
