// Kernel launch interface (kernels.hip) used by the host engine.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "impg_internal.hpp"

namespace impg {

// A frontier record: query `qidx` of the batch wants [start, end) of target_id looked up.  16 B; also what travels
// between ranks of a sharded index (qidx then carries the record's index in its home frontier).
struct FrontierRec {
  uint32_t target_id;
  int32_t start, end;
  uint32_t qidx;
};

// one level's hits, SoA, indexed by pair slot (slot order = frontier order x visit order)
struct HitArrays {  // one level's hit slots: the query id (0xFFFFFFFF = no hit) and {q_first, q_last, t_first, t_last}
  uint32_t *qid;
  int4 *c;          // one 16-byte store / load per slot instead of four scattered 4-byte ones
};

// store_cigar: which ops of the record form each slot's projected slice
struct SliceArrays {
  uint32_t *a;   // original index of the slice's first op in walking order
  uint32_t *n;   // number of ops
  int32_t *off;  // first_op_offset
  int32_t *rem;  // last_op_remaining
};

// visited sets: one table per BFS level; a key's newest table entry is complete
struct VisitedTable {
  const unsigned long long *keys;  // sorted unique (qidx << 32 | sequence id)
  const uint32_t *off, *len;       // into ranges
  const int2 *ranges;              // sorted disjoint (start,end) per key
  const uint32_t *qoff;            // [n_queries + 1] first key of every query (table_qoff_kernel): a key is searched among its query's only
  uint32_t n_groups;
};
constexpr int MAX_VISITED_TABLES = 48;
// striped device counters: COUNT_SLOTS words, one per 128-byte line
constexpr uint32_t COUNT_SLOTS = 64, COUNT_STRIDE = 16;
constexpr size_t COUNT_BYTES = (size_t)COUNT_SLOTS * COUNT_STRIDE * 8;
struct VisitedTables {
  VisitedTable t[MAX_VISITED_TABLES];
  uint32_t n_tables;
  // masked_regions (impg.rs:2077-2081): the lists a (query, sequence) key starts from while no table holds it,
  // shared by every query of the batch; CSR over the sequence ids.  Null without a mask.
  const uint32_t *mask_off;
  const int2 *mask_ranges;
};

void launch_lookup_count(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, bool transitive, const uint32_t *perm,
                         uint32_t *cnt, uint4 *win, uint32_t *wide_n, uint32_t *wide_list, hipStream_t s, bool by_place = false,
                         FrontierRec *se = nullptr /* by_place: the ranges' records at their places (the projection reads the ends there) */,
                         uint32_t *cnt_ref = nullptr /* by_place: the counts in FRONTIER order too (ordered rows) */);
// visit position of every hit of the windows of <= 64 entries, by place: vpos[pair_off[i] + k] for range i's k-th hit in index order
// (dest: the ranges' first rows, computed by the same kernel: dest[place] = offsets[query] + lvbase[query] + slot_ref[range])
struct OrdDestArgs { const FrontierRec *frp; const uint32_t *perm, *slot_ref, *offsets, *lvbase; uint32_t *dest; };
void launch_emit_vpos(const DeviceIndexView &v, uint32_t n, const uint32_t *pair_off, const uint4 *win, uint8_t *vpos, hipStream_t s, bool by_visit = false,
                      const OrdDestArgs *dest = nullptr);
// ---- ordered rows, placed by slot (Engine::ordered_*): see engine.cpp ------------------------------------------------
void launch_ord_self_count(const FrontierRec *self, const impg_gpu_range_t *ranges, uint32_t n, uint32_t *acc, hipStream_t s);
// qbase[q] = slot_ref of the first record with qidx >= q (total beyond the last); lvbase[q] = acc[q] - qbase[q]; acc[q] += the query's slots
void launch_ord_level_bases(const FrontierRec *fr, uint32_t n_fr, const uint32_t *slot_ref, uint32_t total, uint32_t n_queries, uint32_t *acc,
                            uint32_t *lvbase, hipStream_t s);
void launch_ord_self_rows(const FrontierRec *self, const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *offsets, impg_gpu_interval_t *rows,
                          hipStream_t s);
void launch_ord_run_heads(const uint32_t *pair_range, uint32_t n_pairs, uint32_t *run_start, hipStream_t s);
void launch_ord_level_rows(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h, const uint32_t *run_start,
                           const uint32_t *slot_ref, const uint32_t *offsets, const uint32_t *lvbase, int32_t min_output_length,
                           impg_gpu_interval_t *rows, hipStream_t s);
// Ordered rows written by the final level itself (Engine::ordered_rows): a pair's finished 24-byte row goes to
// rows[dest[the range's place] + the hit's visit position] instead of the level's hit arrays -- the level's slots in the
// reference's emission order (frontier order x visit order, impg.rs:2471-2504) at their final place among the batch's rows.
struct OrderedOut {
  impg_gpu_interval_t *rows;      // null: off
  const uint32_t *dest;        // [n_fr] by place: the row of the range's first slot
  const uint8_t *vpos;         // [n_pairs] by place: visit position of a range's k-th hit in index order (windows of <= 64 entries) -- or, by_visit, the inverse
  int32_t min_output_length;   // rows with |q_last - q_first| below it are holes (impg.rs:2482-2504); -1: none
  uint32_t by_visit;           // 1: the range's place k is its k-th VISITED hit and vpos[place] names that hit's bit in the window's mask (the lane-per-place kernel: rows leave in a row)
};
bool ordered_rows_by_visit();  // which of the two the projection of a direct level wants (kernels.hip: launch_project)
// A counting run's FINAL level listed by windows instead of by pairs (engine.cpp: Engine::expand): nothing reads that
// level's slots by position or in order, so the projection kernel takes its pairs straight from what the count pass
// left per range -- place offset, window, hit mask -- and the emit pass with its two 4-byte-per-pair lists is not run
// (only the windows wider than 64 entries, which the wave-per-range emit still lists in pair_entry).
// (With tile_first null but the rest set -- any level whose slots follow the lookup order -- the struct only tells
// project_staged_kernel which ranges own which places; the pairs' entries then come from the emit pass's list.)
struct WindowLists {
  const uint32_t *tile_first;  // [tiles of PROJ_BLOCK places] the range (by place in the lookup order) that holds the tile's first place; null = the pairs are listed
  const uint32_t *pair_off;    // [n_fr] first place of every range's pairs
  const uint4 *win;            // [n_fr] {lo, ub, hit mask of the first 64 window entries}
  const FrontierRec *se;       // [n_fr] the range's record (its start and end are what the projection reads)
  const uint32_t *perm;        // [n_fr] place -> frontier range
  uint32_t n_fr;
  uint32_t *range_out;         // optional: pair_range[] for the per-range counts / the subset filter
  uint32_t masks;              // bit 0: the pairs are named by the windows' hit masks (tile_first is only needed when project_kernel runs the level);
                               // bit 2: the slots' query ids and sources ({qid, the range's place}) interleaved in HitArrays::qid, one 8-byte store (a kept fused level)
  uint32_t range_places;       // 1: range_out names a pair's range by its PLACE in the lookup order (perm not applied: no load) -- kept levels, whose frontier copy is taken in that order
  uint32_t *slice_work, *slice_alloc;  // project_entries_kernel's heavy blocks (EntSlices): the slice list [1 + slice_cap] (word 0 zeroed), a
  uint32_t slice_cap;                  // counter per range block (zeroed); slice_cap >= n_pairs / 32768 + 1.  Null / 0: blocks are taken whole
  OrderedOut ord;              // rows != null: the kernel writes finished rows (see OrderedOut); range_out and the hit arrays are not used
};
// whether launch_project will run a plain projection of n_pairs by-place pairs on the staged kernels (no tile_first[] needed)
bool project_is_staged(const DeviceIndexView &v, uint64_t n_pairs, bool plain);
bool project_entry_major(const DeviceIndexView &v, uint64_t n_pairs, double min_identity);
uint32_t project_entry_blocks(uint32_t n_fr);          // range blocks of project_entries_kernel over n_fr ranges
uint32_t project_entry_slice_cap(uint64_t n_pairs);    // what its slice list must take  // ... on project_entries_kernel (a level named by hit masks)
void launch_tile_first(const uint32_t *cnt, const uint32_t *pair_off, uint32_t n_fr, uint32_t *tile_first, hipStream_t s);
bool emit_by_lanes(const DeviceIndexView &v);
// The pairs listed in projection order (optional: slot == nullptr means the projection runs in slot order):
// place -> the pair's slot, its frontier range and its entry.
struct ProjList {
  uint32_t *slot, *range, *entry;
};
void launch_lookup_emit(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, bool transitive,
                        const uint32_t *pair_off, const uint4 *win, uint32_t *pair_range, uint32_t *pair_entry,
                        const uint32_t *perm, const uint32_t *offp, ProjList pl, const uint32_t *wide_n,
                        const uint32_t *wide_list, hipStream_t s, bool by_place = false, bool wide_only = false);
constexpr uint32_t ROUTE_WORLD_MAX = 1024;
void launch_route_keys(const FrontierRec *fr, uint32_t n, uint32_t world, const uint32_t *owner, uint32_t n_seq, uint32_t *key,
                       uint32_t *idx, unsigned long long *hist, hipStream_t s);
// hits between ranks: pack at the owner (word 0 = the home's frontier index), reorder + unpack at home
// slice_n (store_cigar, 8-word records): ops of every slot's CIGAR slice, carried in word 7; at home slice_at[i] = where the
// i-th arrived hit's ops start among the arrived ops
void launch_hits_pack(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h, const uint32_t *pair_entry,
                      const uint32_t *mrank, uint32_t words, void *out, hipStream_t s, const uint32_t *slice_n = nullptr);
void launch_hits_slice_n(const void *in, uint32_t n, uint32_t *cnt, hipStream_t s);
void launch_hits_unpack(const void *in, uint32_t n, uint32_t words, uint32_t n_front, const uint32_t *run_start, const uint32_t *off,
                        uint32_t *pair_range, HitArrays h, uint32_t *mslot, hipStream_t s, const uint32_t *slice_at = nullptr,
                        uint32_t *slice_pos = nullptr, uint32_t *slice_n = nullptr);
void launch_route_gather(const FrontierRec *fr, const uint32_t *perm, uint32_t n, FrontierRec *out, hipStream_t s);
void launch_frontier_gather(const FrontierRec *fr, const uint32_t *perm, uint32_t n, FrontierRec *out, hipStream_t s);  // out[i] = fr[perm[i]]
// stable order of hit records (words u32 each, fidx first) by fidx when equal fidx are already contiguous
void launch_reorder_runs(const uint32_t *hits, uint32_t n, uint32_t words, uint32_t n_front, uint32_t *run_start,
                         uint32_t *run_len, uint32_t *err, hipStream_t s);
void launch_order_keys(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, uint32_t *key, uint32_t *idx, hipStream_t s,
                       const uint32_t *bounds = nullptr, uint32_t n_blocks = 0, uint32_t block_shift = 0);
void launch_scatter_u32(const uint32_t *in, const uint32_t *perm, uint32_t n, uint32_t *out, hipStream_t s);
void launch_exclusive_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, unsigned long long *d_bsum,
                           unsigned long long *d_total, hipStream_t s);
size_t scan_scratch_bytes(uint32_t n);
void launch_project(const DeviceIndexView &v, const FrontierRec *fr, const uint32_t *pair_range,
                    const uint32_t *pair_entry, uint32_t n_pairs, bool transitive, HitArrays h,
                    unsigned long long *accepted, uint32_t *err_flag, double min_identity, const SliceArrays *slices,
                    ProjList pl, hipStream_t s, const uint32_t *n_pairs_dev = nullptr, bool regroup = false,
                    const WindowLists *wl = nullptr);
// small batches (engine.cpp: run_small): n <= 1024 counts scanned by one block, total left on the device; results
// packed behind a 64-byte header {n_pairs, err, -, -, accepted} into (host-mapped) memory
constexpr uint32_t SMALL_HEADER_BYTES = 64;
void launch_small_scan(const uint32_t *cnt, uint32_t n, uint32_t *off, uint32_t *total, hipStream_t s);
void launch_small_pack(const FrontierRec *fr, const uint32_t *pair_range, const uint32_t *n_pairs_dev, uint32_t pairs_bound, HitArrays h,
                       const uint32_t *err_flag, const unsigned long long *accepted, void *hdr, impg_gpu_interval_t *rows,
                       uint32_t *row_range, hipStream_t s);
void launch_slice_counts(HitArrays h, SliceArrays sl, uint32_t n_pairs, uint32_t *cnt, hipStream_t s);
void launch_slice_write(const DeviceIndexView &v, const uint32_t *pair_entry, HitArrays h, SliceArrays sl, uint32_t n_pairs,
                        const uint32_t *off, uint32_t *out, hipStream_t s);
// per-range counts / checksums of a level's hits: rstat = 16 bytes of scratch per frontier range (zeroed here)
void launch_hit_stats(const FrontierRec *fr, uint32_t n_fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h,
                      int32_t min_output_length, bool skip_same_target, unsigned long long *rstat, unsigned long long *count,
                      unsigned long long *cksum, hipStream_t s, uint32_t stride = 1 /* words between consecutive slots' pair_range / qid */);
void launch_sort5(const FrontierRec *fr, uint32_t n, const uint32_t *pair_off, uint32_t n_pairs, HitArrays h,
                  const uint32_t *pair_entry, const uint32_t *mrank, uint32_t *dest, hipStream_t s);
void launch_permute_slots(const uint32_t *dest, uint32_t n_pairs, HitArrays in, HitArrays out, const uint32_t *pe_in,
                          uint32_t *pe_out, SliceArrays sin, SliceArrays sout, hipStream_t s);
void launch_update_keys(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h,
                        unsigned long long *keys, unsigned long long *vals, unsigned long long *n_active, hipStream_t s);
size_t sort_pairs_scratch_bytes(uint32_t n);
void launch_sort_pairs(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout,
                       const uint32_t *vin, uint32_t *vout, uint32_t n, unsigned end_bit, hipStream_t s);
// the groups of the sorted keys (runs of equal keys != ~0) in two passes: per-tile head counts, then -- with their exclusive
// scan -- every group's start and key
uint32_t group_tiles(uint32_t n);
void launch_group_count(const unsigned long long *skeys, uint32_t n, uint32_t *tile_heads, hipStream_t s);
void launch_group_fill(const unsigned long long *skeys, uint32_t n, const uint32_t *tile_base, uint32_t *gstart, unsigned long long *gkey, hipStream_t s);
void launch_group_heads(const unsigned long long *skeys, uint32_t n, uint32_t *head, hipStream_t s);
void launch_group_scatter(const unsigned long long *skeys, uint32_t n, const uint32_t *head, const uint32_t *gid,
                          uint32_t *gstart, unsigned long long *gkey, hipStream_t s);
// qoff[q] = first key of table `keys` whose query index is >= q, q = 0 .. n_queries
void launch_table_qoff(const unsigned long long *keys, uint32_t n_groups, uint32_t n_queries, uint32_t *qoff, hipStream_t s);
void launch_group_prepare(const VisitedTables &vt, const unsigned long long *gkey, const uint32_t *gstart,
                          uint32_t n_groups, uint32_t n_active, uint32_t *glen, const int2 **old_src,
                          uint32_t *cap, uint32_t *pcap, hipStream_t s);
void launch_visited_update(const unsigned long long *svals, const int32_t *seq_len,
                           const unsigned long long *gkey, const uint32_t *gstart, const uint32_t *glen,
                           const int2 *const *old_src, const uint32_t *noff, const uint32_t *poff,
                           uint32_t n_groups, int32_t min_transitive_len, int32_t mdbr, int2 *new_ranges,
                           uint32_t *new_len, int2 *pieces, uint32_t *n_pieces, const uint32_t *cap, const uint32_t *pcap,
                           uint32_t *big_list, uint32_t *n_big, hipStream_t s);
// hits covered by their group's old list dropped before the replay (kernels.hip "covered_flags")
void launch_covered_flags(const unsigned long long *svals, const uint32_t *head, const uint32_t *gid,
                          const unsigned long long *gkey, const int2 *const *old_src, const uint32_t *cap, const uint32_t *glen,
                          const int32_t *seq_len, uint32_t n_active, uint32_t *keep, hipStream_t s);
void launch_covered_compact(const unsigned long long *svals, const uint32_t *keep, const uint32_t *kpos, uint32_t n_active,
                            unsigned long long *out, uint32_t n_kept, uint32_t n_groups, uint32_t *gstart, uint32_t *glen, uint32_t *cap,
                            uint32_t *pcap, hipStream_t s);
void launch_frontier_emit(const unsigned long long *gkey, const uint32_t *poff, const uint32_t *n_pieces,
                          const uint32_t *foff, uint32_t n_groups, const int2 *pieces, FrontierRec *out, hipStream_t s,
                          const DeviceIndexView *v = nullptr, uint32_t *key = nullptr, uint32_t *idx = nullptr);  // v + key + idx: the next lookup order's keys too
void launch_subset_filter(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, uint32_t *qid,
                          const uint8_t *keep, const impg_gpu_range_t *ranges, hipStream_t s);
// level -1 under a mask: the input range is inserted into a copy of its target's mask list (impg.rs:2084-2086);
// cap[q] = that list's length + 1 bounds both the new list and the pieces
void launch_mask_caps(const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *mask_off, uint32_t n_seq, uint32_t *cap,
                      hipStream_t s);
void launch_visited_init_masked(const impg_gpu_range_t *ranges, uint32_t n, const int32_t *init_len, uint32_t n_seq,
                                const uint32_t *mask_off, const int2 *mask_ranges, int32_t min_transitive_len,
                                const uint32_t *loff, unsigned long long *keys, uint32_t *len, int2 *rng, int2 *pieces,
                                uint32_t *n_self, uint32_t *n_front, hipStream_t s);
void launch_masked_self_emit(const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *loff, const uint32_t *n_self,
                             const uint32_t *self_off, const uint32_t *front_off, int32_t min_transitive_len,
                             const int2 *pieces, FrontierRec *self_out, FrontierRec *frontier_out, hipStream_t s);
void launch_visited_init(const impg_gpu_range_t *ranges, uint32_t n, const int32_t *seq_len, uint32_t n_seq,
                         int32_t min_transitive_len, unsigned long long *keys, uint32_t *off, uint32_t *len, int2 *rng,
                         FrontierRec *self_iv, uint32_t *in_frontier, hipStream_t s);
void launch_compact_frontier(const FrontierRec *in, const uint32_t *flag, const uint32_t *pos, uint32_t n,
                             FrontierRec *out, hipStream_t s);
void launch_ranges_to_frontier(const impg_gpu_range_t *ranges, uint32_t n, FrontierRec *out, hipStream_t s);

void launch_frontier_to_stack(const FrontierRec *fr, uint32_t n, const uint32_t *pop_depth, bool use_depth,
                              unsigned long long *key, int32_t *st, int32_t *en, uint32_t *depth, hipStream_t s);
void launch_dfs_pop_flags(const unsigned long long *key, const uint32_t *depth, uint32_t n, uint32_t max_depth,
                          bool pop_front, uint32_t *fr_flag, uint32_t *keep_flag, uint32_t *pop_depth, uint32_t *pop_sel /* n_queries words of scratch */,
                          uint32_t n_queries, hipStream_t s);
void launch_dfs_pop_scatter(const unsigned long long *key, const int32_t *st, const int32_t *en, const uint32_t *depth,
                            uint32_t n, const uint32_t *fr_flag, const uint32_t *fr_pos, const uint32_t *keep_flag,
                            const uint32_t *keep_pos, FrontierRec *fr_out, unsigned long long *key_out, int32_t *st_out,
                            int32_t *en_out, uint32_t *depth_out, hipStream_t s);
void launch_iota(uint32_t *v, uint32_t n, hipStream_t s);
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *dst, hipStream_t s);
void launch_gather_u64(const unsigned long long *src, const uint32_t *idx, uint32_t n, unsigned long long *dst, hipStream_t s);
void launch_run_heads(const unsigned long long *skeys, uint32_t n, uint32_t *head, hipStream_t s);
void launch_run_starts(const uint32_t *head, const uint32_t *gid, uint32_t n, uint32_t *gstart, hipStream_t s);
void launch_dfs_merge(const uint32_t *gstart, uint32_t n_groups, uint32_t n, const uint32_t *perm, const int32_t *st,
                      const int32_t *en, const uint32_t *depth, int32_t *st_m, int32_t *en_m, uint32_t *depth_m,
                      uint32_t *cnt, hipStream_t s);
void launch_dfs_compact(const uint32_t *gstart, const uint32_t *cnt, const uint32_t *off, uint32_t n_groups,
                        const unsigned long long *skeys, const int32_t *st_m, const int32_t *en_m, const uint32_t *depth_m,
                        unsigned long long *key_out, int32_t *st_out, int32_t *en_out, uint32_t *depth_out, hipStream_t s);
size_t sort_u32_scratch_bytes(uint32_t n);
void launch_sort_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                     uint32_t n, hipStream_t s, unsigned begin_bit = 0, unsigned end_bit = 32);
// the lookup order's sort (kernels.hip: order_scatter_kernel): perm_out[k] = the index of the k-th smallest key by its bits
// [0, end_bit), equal keys in index order.  keys are left in an unspecified order; key_tmp / perm_tmp: n words each
size_t order_sort_scratch_bytes(uint32_t n);
void launch_order_sort(uint32_t *keys, uint32_t *key_tmp, uint32_t *perm_out, uint32_t *perm_tmp, uint32_t n, unsigned end_bit, void *scratch,
                       hipStream_t s);
#ifdef IMPG_OS_CLOCKS
void order_sort_clocks(unsigned long long *out, bool reset);  // (phase clocks of order_scatter_kernel: a tuning build)
#endif
size_t sort_u64v_scratch_bytes(uint32_t n);
void launch_sort_u64v(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout,
                      const unsigned long long *vin, unsigned long long *vout, uint32_t n, hipStream_t s, unsigned end_bit = 64,
                      unsigned begin_bit = 0);
void launch_compact_fill(const unsigned long long *keys, uint32_t n, uint32_t table, unsigned long long *key_out,
                         unsigned long long *src_out, hipStream_t s);
void launch_compact_last(const unsigned long long *skeys, uint32_t n, uint32_t *flag, hipStream_t s);
void launch_compact_select(const VisitedTables &vt, const unsigned long long *skeys, const unsigned long long *ssrc, uint32_t n,
                           const uint32_t *flag, const uint32_t *pos, unsigned long long *key_out,
                           unsigned long long *src_out, uint32_t *len_out, hipStream_t s);
void launch_compact_copy(const VisitedTables &vt, const unsigned long long *src, const uint32_t *off, const uint32_t *len,
                         uint32_t n, int2 *ranges_out, hipStream_t s);

// the update's hits grouped by (query, sequence) without a global sort (kernels.hip: seg_group_kernel)
bool seg_group_fits(uint32_t n_seq);
void launch_seg_bounds(const FrontierRec *fr, uint32_t n_fr, uint32_t n_queries, const uint32_t *pair_range, uint32_t n_pairs, uint32_t *run_start,
                       uint32_t *run_end, uint32_t *qfirst, uint32_t *qlast, uint32_t *unsorted, hipStream_t s, const uint32_t *off_perm = nullptr,
                       const uint32_t *pair_off = nullptr, const uint32_t *cnt = nullptr);  // pair_off / cnt (+ the order they are listed in): the runs straight from the lookup's offsets
void launch_seg_group(bool count_only, const FrontierRec *fr, const uint32_t *qfirst, const uint32_t *qlast, const uint32_t *run_start,
                      const uint32_t *run_end, HitArrays h, uint32_t n_queries, uint32_t n_seq, uint32_t *qact, const uint32_t *qdst, uint32_t *qgrp,
                      const uint32_t *gdst, uint32_t *gstart, unsigned long long *gkey, unsigned long long *svals, uint32_t *qbins, hipStream_t s,
                      uint32_t parts = 1, uint32_t *qtot = nullptr);  // parts > 1: every query cut into that many slices (qbins per slice, qtot per query: required)
size_t seg_group_bins_bytes(uint32_t n_queries, uint32_t n_seq, uint32_t parts = 1);  // qbins (nullable when parts == 1): the count pass's sequence counters, kept for the place pass
uint32_t seg_group_parts(uint64_t n_hits, uint32_t n_queries, uint32_t n_seq, uint64_t largest_query = 0);  // slices per query for such a level (0: library sort)

// ---- the per-query walk (walk_device.inc): one workgroup takes a query through all its levels / pops ----------------
struct WalkArgs {
  DeviceIndexView v;
  const impg_gpu_range_t *ranges;
  uint32_t n_queries;
  int dfs;
  uint32_t max_depth;
  int32_t min_transitive_len, mdbr, min_output_length;
  double min_identity;  // NaN: no filter
  const uint8_t *subset_keep;
  unsigned long long *count, *cksum;  // per query, hits only (the self interval is not a hit), nullable
  unsigned long long *accepted;       // striped
  uint32_t *err_flag;
  // rows (nullable): query q writes its rows at rows[row_base[q] ..], at most row_cap[q] of them; n_rows[q] counts all
  impg_gpu_interval_t *rows;
  const unsigned long long *row_base;
  const uint32_t *row_cap;
  uint32_t *n_rows;
  uint32_t *next_query;  // work counter
  uint32_t *overflow;    // set: a query did not fit its slab
  char *slabs;
  size_t slab_bytes;
  uint32_t wcap, hcap, vcap, gcap, scap;  // records per frontier / piece list, pairs per step, visited ranges, piece scratch, DFS stack records
  unsigned long long *dbg;           // IMPG_WALK_DEBUG: 32 words of phase clocks and counters (null: off)
  // masked_regions (null: none): the map's lists by sequence id; the length a list clamps to when the range's own
  // target opens it (0 for a sequence the map does not hold, impg.rs:2048-2053) and when a hit opens it
  const uint32_t *mask_off;
  const int2 *mask_ranges;
  const int32_t *mask_init_len, *mask_touch_len;
  // grid form (members > 1; BFS with a depth limit): query q is walked by workgroups [q * members, (q + 1) * members);
  // the first one takes it up to its last level, whose ranges all of them share out (walk_final_level)
  uint32_t members;
  struct WalkGroupCtl *ctl;  // one per query, zeroed before the launch
};
constexpr uint32_t WALK_MAX_MEMBERS = 64;
struct WalkGroupCtl {
  uint32_t state;     // 0: the leader is on the earlier levels; 1: the last level's frontier is published; 2: there is none
  uint32_t nw;        // ranges of the last level's frontier (the leader's wa[])
  uint32_t rows_run;  // rows written before the last level
  uint32_t arrived;   // members whose projections are done
  unsigned long long cksum, count;
  uint32_t emit[WALK_MAX_MEMBERS];  // rows of each member's share
  uint32_t pad[(512 - 32 - 4 * WALK_MAX_MEMBERS) / 4];
};
static_assert(sizeof(WalkGroupCtl) == 512, "one control block per query");
size_t walk_slab_bytes(uint32_t n_seq, bool wide, uint32_t wcap, uint32_t hcap, uint32_t vcap, uint32_t gcap, uint32_t scap);
// wide: 16 waves per workgroup (a handful of queries, latency) instead of one (a batch, throughput); ident_mode: the
// projections take the identity filter's path (a filter is set, or the index has no prefix lines)
#ifndef IMPG_WALK_WAVES
#define IMPG_WALK_WAVES 6  // 100 000-range DFS batch: 4 waves per SIMD (101 VGPRs, what the compiler takes unasked) 4.22 s, 5: 3.80, 6: 3.66, 8: 3.68
#endif
constexpr uint32_t WALK_WAVES_PER_SIMD = IMPG_WALK_WAVES;  // resident waves per SIMD the one-wave-per-query walk is compiled for
uint32_t walk_grid_blocks_per_cu();  // the occupancy query behind the grid form's launch size (Engine::walk_group_size)
void launch_walk(const WalkArgs &a, uint32_t n_workgroups, bool wide, bool ident_mode, hipStream_t s);  // (members > 1: n_workgroups = queries x members)

}  // namespace impg
