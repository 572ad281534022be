// Per-index execution state: stream, events, scratch buffers (engine.cpp).
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace impg {

struct LevelBufs {  // one BFS level: its frontier and its hit slots
  DevBuf frontier, pair_range, qid, coords;  // coords: {q_first, q_last, t_first, t_last} per slot (16 B)
  // store_cigar: slice descriptors per slot, then the materialised slices
  DevBuf sl_a, sl_n, sl_off, sl_rem, slice_pos, slice_pool;
  uint64_t slice_total = 0;
  uint32_t n_frontier = 0, n_pairs = 0;
  // ordered rows placed by slot (Engine::ordered_rows): the level's slots per frontier record -- counts, then their exclusive
  // scan -- in frontier order, and per query where the level's rows start relative to the scan (kernels.hip "Ordered rows")
  DevBuf slot_ref, lvbase, run_start;
  bool qs_interleaved = false;  // qid holds {query id, source} pairs, 8 bytes a slot (a kept fused level); pair_range is not written
  bool placed = false;  // its rows are already in the batch's row array (the fused final level writes them itself)
  LevelBufs() = default;
  // the levels a full-results call keeps are new objects at every level of every call: their blocks are recycled
  // through the engine's pool (hipMalloc / hipFree are 0.1-1 ms each and hipFree synchronises the device)
  explicit LevelBufs(BufPool *pool) {
    for (DevBuf *b : {&frontier, &pair_range, &qid, &coords, &sl_a, &sl_n, &sl_off, &sl_rem, &slice_pos, &slice_pool, &slot_ref, &lvbase, &run_start}) b->pool = pool;
  }
};
struct VisitedStore {  // device storage of one VisitedTable
  DevBuf keys, off, len, ranges, qoff;
  uint32_t n_groups = 0;
  explicit VisitedStore(BufPool *pool) { keys.pool = off.pool = len.pool = ranges.pool = qoff.pool = pool; }
};

struct SplitBatch {};  // thrown when a level exceeds pair_budget: the caller halves the chunk

// Where a frontier is expanded.  On one GPU the engine looks its records up in its own index (Engine::expand).
// With the index sharded over ranks (sharded.cpp) the records travel to the ranks that own their targets and
// the hits come home; the slot arrays of `L` end up exactly as a local expansion would have left them (frontier
// order x visit order), so everything downstream -- visited update, DFS stacks, masks, result assembly -- is
// shared.  A hop is collective: every rank takes part in every hop, with an empty frontier once it has nothing
// left (`alive` = this rank still has work pending); all_dead = no rank has, the walk is over everywhere.
struct HopResult {
  uint64_t pairs;
  bool all_dead;
};
struct Engine;
struct Expander {
  virtual ~Expander() {}
  // need_hits: somebody at home reads the slots (update, counts, rows); need_rows: including the target columns
  virtual HopResult hop(Engine &home, const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive,
                        LevelBufs &L, impg_gpu_stats_t *st, bool need_hits, bool need_rows, bool alive) = 0;
};

struct Engine {
  // The pools come first, so they die last: buffers of every kind end up owning pooled blocks (the slot arrays are
  // swapped with scratch buffers by the MultiImpg sort, the DFS stacks with per-round buffers) and hand them back
  // when they are destroyed.
  BufPool table_pool;  // blocks of the visited tables (new at every level of every chunk)
  BufPool level_pool;  // blocks of the levels kept for a full-results call, and of the DFS driver's per-round buffers
  hipStream_t stream = nullptr;
  DevBuf counters;          // 8 x u64: [2] error flag, [3] scan total
  DevBuf acc_slots, act_slots;  // striped counters: accepted projections, keyed hits
  uint64_t *h_slots = nullptr;
  uint64_t read_slots(DevBuf &b);
  uint64_t *h_counters = nullptr;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;
  struct Timed { hipEvent_t a, b; int kind; };
  std::vector<Timed> timed;
  // scratch, grown on demand and reused across calls
  DevBuf big_list;  // groups of a level whose visited list gets a whole wave
  DevBuf cnt, win, pair_off, pair_entry, scan_tmp, scan_tmp2, keys, skeys, vals, svals, sort_tmp, head, gid, gstart, glen, old_src,
      cap, pcap, poff, pieces, n_pieces, foff, frontier_a, frontier_b, self_scratch, ranges_dev, stat_count,
      stat_cksum, stage_off;
  LevelBufs level_scratch;
  std::vector<std::unique_ptr<VisitedStore>> tables;
  uint32_t table_queries = 0;  // queries of the batch the tables belong to (their qoff arrays have one entry more)
  void index_table(VisitedStore &t);  // qoff[] of a table whose keys are in place
  uint64_t last_projected = 0;
  uint64_t pair_budget = 1ull << 28;  // candidate pairs per level kept in HBM at once
  uint32_t chunk_ranges = 0;          // ranges per chunk (0 = try the whole batch)
  bool split_ok = false;
  double min_identity = __builtin_nan("");  // of the batch / stage call in flight
  bool store_cigar = false;
  bool multi = false;       // MultiImpg semantics for the batch in flight (params.multi_impg)
  // Nobody reads the slots by position: the visited update only needs the hits of one query in frontier order x
  // visit order, and the lookup order preserves exactly that (its key is monotone in (target, start), the order of
  // a query's frontier; ties keep frontier order); the result rows are placed record by record (rows_device.hip),
  // which only needs a frontier record's slots to be one run in visit order.  The slots are therefore laid out in
  // projection order: the emit pass writes two coalesced lists instead of four arrays, and the projection kernel
  // reads and writes them at its own index.  (Not under store_cigar / MultiImpg, whose slice materialisation and
  // five-key sort walk the reference's slot order.)
  bool free_slot_order = false;
  bool free_slots_allowed = true;  // option "free_slot_order" (A/B runs)
  std::function<void()> on_kernels_done;  // the row stream: called by assemble_results once the chunk's kernels have run (the copies follow)
  bool regroup_pairs = true;       // option "regroup_entries": a projection block sorts its 256 pairs by entry first
  // A counting run's final level (max_depth reached, or a plain query): no update follows and no row is kept, so nothing
  // reads its slots in order -- the projection kernel enumerates the pairs from the count pass's windows and the emit
  // pass is skipped (WindowLists, kernels.hpp).  Set by run() around that level's hop.
  bool fuse_allowed = true;        // option "fuse_final_level" (A/B runs)
  bool fuse_final = false, fuse_need_ranges = false, fuse_range_places = false;
  bool last_range_places = false;  // the last expand wrote places of the lookup order into pair_range (a kept fused level)
  // set by the caller around run(): the kept levels' slots may come in any order, as long as pair_range names every
  // slot's frontier record (a kept final level may then be fused like a counting run's)
  bool keep_any_order = false;
  // Ordered rows placed by slot (set by the caller around run(), with `keep`): the batch's rows grouped by range in the
  // reference's emission order, every SLOT at its final place (a None projection stays as a hole row) -- so that where a
  // row goes follows from the lookups' counts alone and the fused final level can write its rows itself.  After run():
  // ord_rows[ord_total] (impg_gpu_interval_t), ord_offsets[n + 1].
  bool ordered_rows = false;
  DevBuf ord_acc, ord_offsets, ord_rows, ord_dest, ord_vpos, ord_cnt;
  uint64_t ord_total = 0;
  bool ord_offsets_done = false;
  uint32_t ord_n = 0;
  int32_t ord_min_len = -1;
  const FrontierRec *ord_self = nullptr;        // transitive: the self intervals (null: the ranges themselves)
  const impg_gpu_range_t *ord_ranges = nullptr;
  void ordered_level(const FrontierRec *fr, uint32_t n_fr, LevelBufs &L, bool counts_by_range, uint64_t P);  // after a level's count pass
  void ordered_offsets();                                                                                   // once every level is counted
  void ordered_finish(std::vector<std::unique_ptr<LevelBufs>> &levels);                                       // self rows + the levels not yet placed
  float ms_place = 0;
  DevBuf ent_work, ent_alloc;      // project_entries_kernel: the slice list of its heavy blocks, a place counter per range block
  DevBuf win_se, tile_first;       // the ranges' (start, end) by place; first range of every projection tile
  int filter_covered = 0;          // option "filter_covered": hits covered by their group's old list dropped before the replay (0 off: it bought nothing on config 5, where hits are covered by the list as it GROWS, not as the level found it; 1 always, 2 long groups)
  uint64_t covered_dropped = 0;    // ... how many that was, over the engine's life (tuning aid)
  DevBuf m_dest, m_qid, m_coords, m_pe, m_sa, m_sn, m_so, m_sr;  // 5-key sort: destination + double buffers
  // projection order (locality): ranges sorted by window position, their slots listed in that order
  DevBuf wide_n, wide_list;  // ranges whose window is wider than the lane-per-range emit pass takes
  DevBuf lo_key, lo_key2, lo_idx, lo_perm, lo_cnt, lo_off, lo_offp, slot_of;
  uint32_t locality_min = 4096;  // frontier ranges below which the reordering is not worth its launches (0 = never reorder)
  const uint32_t *stage_perm = nullptr;  // lookup order of the last stage_count call
  uint32_t stage_n = 0;  // frontier size of the last stage_count call

  explicit Engine(int device);
  ~Engine();
  hipEvent_t event();
  uint64_t read_counter(int k);
  uint64_t scan(const uint32_t *in, uint32_t *out, uint32_t n);
  // two scans over n items each and up to four extra device words (a kernel's flags) behind ONE synchronisation
  void scan2(const uint32_t *in_a, uint32_t *out_a, const uint32_t *in_b, uint32_t *out_b, uint32_t n, uint64_t &total_a, uint64_t &total_b,
             const uint32_t *d_extra = nullptr, uint32_t *h_extra = nullptr, uint32_t n_extra = 0);
  // lookup / projection order (locality): lookup_order() before the count pass gives the permutation (null
  // when not worth it); projection_offsets() after the scan gives every range's first place in slot_of
  // blocks (optional, the owner side of a sharded counting hop): the records are n_blocks contiguous blocks, one per
  // home rank (d_bounds[n_blocks + 1], device); the order then runs block by block, so that the slots -- laid out in
  // that order -- stay grouped by the rank they go back to
  struct RecordBlocks { const uint32_t *d_bounds; uint32_t n_blocks; };
  const uint32_t *lookup_order(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, const RecordBlocks *blocks = nullptr);
  void projection_offsets(const uint32_t *d_perm, uint32_t n_fr, const uint32_t *d_cnt, uint64_t P, const uint32_t *&d_offp,
                          ProjList &pl);
  DevBuf proj_range, proj_entry;  // the pairs' ranges / entries in projection order (next to slot_of)
  // raw: the owner side of a sharded hop -- slots as projected; the subset filter and the MultiImpg sort run at home
  uint64_t expand(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                  impg_gpu_stats_t *st, bool raw = false, const RecordBlocks *blocks = nullptr);
  const void *keys_for = nullptr;  // the frontier whose lookup-order keys frontier_emit has already written (lo_key / lo_idx)
  uint32_t keys_n = 0;
  uint32_t expand_n_fr = 0;    // the frontier the last expand looked up (its cnt / pair_off describe that frontier's runs)
  bool last_by_place = false;  // the last expand laid its slots out in lookup order (pair_off is then by place)
  // subset filter + MultiImpg five-key sort of a level's slots (pair_off: first slot of every frontier record;
  // tie_rank[tie_idx[slot]] = the MultiImpg tie order of the slot's entry)
  void post_expand(const FrontierRec *fr, uint32_t n_fr, LevelBufs &L, const uint32_t *d_pair_off, uint32_t *tie_idx,
                   const uint32_t *tie_rank, SliceArrays sl);
  Expander *remote = nullptr;  // set: frontiers are expanded on the owning shards
  HopResult hop(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n_fr, bool transitive, LevelBufs &L,
                impg_gpu_stats_t *st, bool need_hits, bool need_rows, bool alive);
  uint32_t update(const DeviceIndexView &v, const FrontierRec *fr, LevelBufs &L, uint32_t n_queries,
                  const impg_gpu_params_t &p, DevBuf &next_frontier);
  VisitedTables tables_view() const;
  void compact_tables();   // fold all visited tables into one
  // scratch of the DFS driver / table compaction
  DevBuf dk_a, dk_b, ds_a, ds_b, de_a, de_b, dd_a, dd_b, d_flag2, d_pos2, d_popdepth, d_popsel, d_perm, d_perm2, d_k32, d_k32b,
      d_src, d_src2, d_ckey, d_ckey2;
  void run_dfs(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
               std::vector<std::unique_ptr<LevelBufs>> *keep, unsigned long long *d_count, unsigned long long *d_cksum,
               impg_gpu_stats_t *st, DevBuf *self_out);
  void finish_run(impg_gpu_stats_t *st, hipEvent_t t0, hipEvent_t t1);
  static void check_params(const impg_gpu_params_t &p);
  // level -1 of a transitive batch: visited table 0, self intervals, frontier 0
  uint32_t begin_transitive(const DeviceIndexView &v, const impg_gpu_range_t *d_ranges, uint32_t n,
                            const impg_gpu_params_t &p, FrontierRec *d_self, DevBuf &frontier_out);
  // masked_regions of the batch in flight (capi sets these around Engine::run): CSR over the sequence ids, the
  // sequence length a set starts with at level -1 / on first touch (visited_entry, impg.rs:2041-2055)
  // subset filter of the batch in flight: keep[sequence id] on the device (null = none), and the batch's ranges
  // (a hit on the query's own target always stays)
  DevBuf subset_keep;
  bool subset_on = false;
  const impg_gpu_range_t *cur_ranges = nullptr;
  bool masked = false;
  DevBuf mask_off, mask_ranges, mask_init_len, mask_touch_len;
  bool mask_has_empty = false;  // a mask range with start == end
  uint64_t mask_ranges_total = 0, mask_lists = 0;  // ranges / sequences of the map
  DevBuf self_off;        // masked batches: query q's self intervals are self[self_off[q] .. self_off[q+1])
  uint64_t n_self = 0;
  uint32_t begin_transitive_masked(const DeviceIndexView &v, const impg_gpu_range_t *d_ranges, uint32_t n,
                                   const impg_gpu_params_t &p, DevBuf &self, DevBuf &frontier_out);
  float stage_ms[3] = {0, 0, 0};
  uint64_t stage_launches = 0;
  DevBuf stage_next;       // next frontier produced by the last stage_update
  uint32_t stage_next_n = 0, stage_queries = 0;
  void run(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
           std::vector<std::unique_ptr<LevelBufs>> *keep, unsigned long long *d_count, unsigned long long *d_cksum,
           impg_gpu_stats_t *st, DevBuf *self_out);
  // Impg::query of a small batch (the trait's per-call shape) as one chain of launches and one synchronisation;
  // false = not applicable (too many candidate pairs): the caller takes the general path
  static constexpr uint32_t SMALL_RANGES = 64, SMALL_PAIRS = 1u << 18;
  bool run_small(const impg_gpu_index &ix, const impg_gpu_range_t *h_ranges, uint32_t n, const impg_gpu_params_t &p,
                 impg_gpu_results &res);
  // The per-query walk (walk_device.inc): transitive queries of a small batch, and DFS batches of any size, in ONE
  // launch -- a workgroup per query.  false = not applicable or a query outgrew its slab: the caller takes the batch
  // engine.  rows: null = counts only; else the queries' rows (see WalkRows).
  struct WalkRows {
    DevBuf rows, base, cap, n_rows;           // device: row pool, first row / capacity / row count of every query
    std::vector<uint32_t> h_n_rows;           // row counts, on the host
    std::vector<unsigned long long> h_base;   // where each query's rows start in `rows`
  };
  bool walk_allowed = true, walk_bfs = false;  // option "walk_kernel": 0 never, 1 DFS (default), 2 also small BFS batches
  bool walk_applicable(const impg_gpu_index &ix, uint32_t n, const impg_gpu_params_t &p) const;
  static void walk_caps(bool wide, WalkArgs &a);
  static uint64_t walk_workgroups(const impg_gpu_index &ix, bool wide);
  void reserve_walk_slabs(const impg_gpu_index &ix, bool dfs_too);  // option prewarm_walk
  bool run_walk(const impg_gpu_index &ix, const impg_gpu_range_t *d_ranges, uint32_t n, const impg_gpu_params_t &p,
                unsigned long long *d_count, unsigned long long *d_cksum, impg_gpu_stats_t *st, WalkRows *rows, uint32_t *h_n_rows = nullptr);
  DevBuf walk_slabs, walk_ctr, walk_ctl;
  DevBuf rstat;  // hit_stats: a level's per-range counts / checksums
  DevBuf seg_run, seg_q, seg_bins, seg_tot;  // update by segments: run bounds per frontier range; first / last range, active hits and output offset per query
  std::atomic<uint64_t> *seg_stats = nullptr;  // the handle's counters (impg_gpu_index::seg_stats), or null
  uint32_t seg_parts_force = 0;  // option "segment_parts": every level that groups by segments cuts its queries into this many slices (0: by size)
  bool seg_group = true;  // option "segment_groups": the update's hits grouped query by query instead of by the library's radix sort
  uint32_t walk_members = 0;  // option "walk_members": workgroups per query of the grid form (0: as many as fit, at most 32; 1: no grid form)
  uint32_t walk_group_size(const impg_gpu_index &ix, uint32_t n, const impg_gpu_params_t &p) const;
  char *small_in = nullptr;    // pinned: the ranges on their way in
  char *small_out = nullptr;   // pinned + mapped: header, rows, the rows' ranges (written by the device)
  void *small_out_dev = nullptr;
  size_t small_out_cap = 0;
};

// owner: null = one shard holds everything; else owner[target id] = the shard that holds the target's entries
void build_index(impg_gpu_index &ix, const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops,
                 size_t n_ops, const int64_t *seq_len, uint32_t n_seq, bool bidirectional, int order_policy,
                 uint32_t shard, uint32_t n_shards, const uint32_t *owner, const TpInput *tp = nullptr,
                 const EntryPlan *plan = nullptr);

struct EngineLease {  // an engine of the index for the duration of one call (capi.cpp)
  impg_gpu_index &ix;
  Engine *e = nullptr;
  explicit EngineLease(impg_gpu_index &ix);
  ~EngineLease();
  EngineLease(const EngineLease &) = delete;
  Engine &operator*() { return *e; }
  Engine *operator->() { return e; }
};

// ---- shared by capi.cpp and sharded.cpp --------------------------------------------------------------
void require_device(int device);
std::unique_ptr<impg_gpu_index> make_index(const impg_gpu_record_t *records, size_t n_records, const uint32_t *ops,
                                           size_t n_ops, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                           int order_policy, int device, uint32_t shard, uint32_t n_shards,
                                           const HostSeqIndex *seq, const std::vector<uint64_t> *file_first,
                                           const uint32_t *owner);
void assemble_results(Engine &E, const impg_gpu_range_t *h_ranges, uint32_t n, const impg_gpu_params_t &p,
                      std::vector<std::unique_ptr<LevelBufs>> &levels, DevBuf &self_dev, impg_gpu_results &res,
                      uint64_t max_rows = ~0ull, hipEvent_t kernels_done = nullptr);
// ---- result rows on the device (rows_device.hip) ---------------------------------------------------------------
// Where every emitted slot of a chunk goes among the chunk's result rows (grouped by range, emission order within).
struct RowPlan {
  DevBuf flag, pos;            // per slot over [self records | level 0 | level 1 | ...]
  DevBuf rec_start, rec_dest;  // per record over [self records | level 0's frontier | ...]: pos of its first slot, its first row
  DevBuf offsets;              // [n_ranges + 1] u32: first row of every range
  uint64_t n_slots = 0, n_recs = 0;
  uint32_t n_rows = 0, n_self = 0;
  const FrontierRec *d_self = nullptr;
};
struct RowSinks {  // any of the two forms (null = not wanted)
  impg_gpu_interval_t *rows;      // the trait's rows
  uint32_t *q, *qid, *tid;        // SoA: range, query sequence, target sequence ...
  int4 *c;                        // ... {q_first, q_last, t_first, t_last} (the BED merges work on these)
  uint32_t *clen;                 // store_cigar: number of ops of the row's CIGAR
};
// plain_retain: a plain query's rows also pass perform_query's min_output_length retain (main.rs:11682-11688: the text writers)
void plan_rows(Engine &E, uint32_t n_ranges, const impg_gpu_params_t &p, std::vector<std::unique_ptr<LevelBufs>> &levels,
               DevBuf &self_dev, bool plain_retain, RowPlan &pl);
void scatter_rows(Engine &E, std::vector<std::unique_ptr<LevelBufs>> &levels, const RowPlan &pl, const RowSinks &out);
uint64_t build_row_cigars(Engine &E, std::vector<std::unique_ptr<LevelBufs>> &levels, const RowPlan &pl, const DevBuf &clen, DevBuf &coff,
                          DevBuf &pool);
void append_results(impg_gpu_results &res, impg_gpu_results &part);
// ---- BED on the device (bed_device.hip) -----------------------------------------------------------------------------
uint32_t device_bed_rows(Engine &E, const impg_gpu_index &ix, uint32_t n_ranges, const impg_gpu_params_t &p, int32_t merge_distance,
                         std::vector<std::unique_ptr<LevelBufs>> &levels, DevBuf &self_dev, DevBuf &out);
void device_bed_text(Engine &E, const impg_gpu_index &ix, const DevBuf &rows, uint32_t n_rows, uint32_t n_ranges,
                     const std::vector<std::string> &rnames, bool original_coords,
                     const std::function<void(const char *, size_t)> &sink);
// the names BED rows print for ranges [b, e): the caller's, else "{chrom}:{start}-{end}" (partition.rs:1741, :1762)
std::vector<std::string> bed_range_names(const impg_gpu_index &ix, const impg_gpu_range_t *ranges, const char *const *range_names,
                                         size_t b, size_t e);
// query + both BED merges + text for this rank's / this handle's ranges of a sharded index (sharded.cpp)
void sharded_bed_batch(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p,
                       const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names,
                       const std::function<void(const char *, size_t)> &sink, double *seconds3);  // part's rows behind res's (chunks, ranks)
void check_ranges(const impg_gpu_range_t *ranges, size_t n);
// masked_regions / subset filter of a batch -> an engine's device tables (cleared when the engine's lease ends)
void apply_mask(Engine &E, const impg_gpu_index &ix, const impg_gpu_mask_t *m, const impg_gpu_params_t &p);
void apply_subset(Engine &E, const impg_gpu_index &ix, const uint8_t *subset_keep);
// entry points of sharded.cpp behind the public query calls
// the largest of the ranks' values (a collective on lane 0 of a rank's shard: every rank calls it at the same point)
uint64_t shard_agree_max(impg_gpu_index &ix, uint64_t mine);
int sharded_query_batch(impg_gpu_index &ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t &p,
                        const impg_gpu_mask_t *mask, const uint8_t *subset_keep, impg_gpu_results **out);
int sharded_query_stats(impg_gpu_index &ix, const impg_gpu_range_t *ranges, bool on_device, size_t n,
                        const impg_gpu_params_t &p, uint64_t *per_range_count, uint64_t *per_range_checksum,
                        impg_gpu_stats_t *stats);
// targets bin-packed onto shards by entry count: heaviest first onto the least loaded shard
void shard_assign(const uint64_t *entries_per_target, uint32_t n_seq, uint32_t n_shards, uint32_t *owner);
void count_entries_per_target(const impg_gpu_record_t *records, size_t n_records, uint32_t n_seq, bool bidirectional,
                              std::vector<uint64_t> &cnt);

}  // namespace impg
