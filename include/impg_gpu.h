/*
 * impg_gpu.h -- C ABI of the MI355X-native batch interval-query + CIGAR
 * coordinate-projection engine (libimpg_gpu.so).
 *
 * This is the drop-in boundary for impg's hot path: a Rust
 * `struct GpuImpg; impl ImpgIndex for GpuImpg` (reference
 * src/impg_index.rs:21-121) is ~150 lines of glue over these entry points
 * (binding sketch in INTEGRATION.md).  Plain pointers and sizes only; no C++
 * or torch types; nothing unwinds across the boundary -- every call returns an
 * int status (0 = ok, <0 = error) and impg_gpu_last_error() gives the message
 * for the calling thread.  All coordinates are i32 and all sequence ids u32,
 * exactly the reference's types (src/impg.rs:164-174, :225).
 *
 * The library requires a gfx950 GPU: there is no CPU fallback.  Index creation
 * and queries fail with IMPG_E_HIP when no device is present.
 */
#ifndef IMPG_GPU_H
#define IMPG_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMPG_OK 0
#define IMPG_E_INVALID (-1)     /* bad argument / malformed input (the reference panics: impg.rs:88,:506) */
#define IMPG_E_HIP (-2)         /* HIP runtime error / no device */
#define IMPG_E_OOM (-3)
#define IMPG_E_IO (-4)
#define IMPG_E_UNSUPPORTED (-5) /* valid in the reference, not built yet */

typedef struct impg_gpu_index impg_gpu_index_t;
typedef struct impg_gpu_results impg_gpu_results_t;

/* One alignment record as the reference parses it from PAF
 * (AlignmentRecord, src/alignment_record.rs:12-22; parse_paf_line, src/paf.rs:118-177)
 * with its CIGAR already tokenised into packed ops: op = code<<29 | len,
 * codes '='0 'X'1 'I'2 'D'3 'M'4 -- identical to CigarOp (src/impg.rs:81-93). */
typedef struct {
  uint32_t query_id, target_id;
  int32_t query_start, query_end;
  int32_t target_start, target_end;
  uint64_t cigar_off; /* first op of this record in the op pool */
  uint32_t cigar_len; /* number of ops (0 = record has no cg:Z tag) */
  uint32_t strand;    /* 0 '+', 1 '-' */
} impg_gpu_record_t;

/* A query range: (target_id, range_start, range_end) of ImpgIndex::query
 * (src/impg_index.rs:26-35). */
typedef struct {
  uint32_t target_id;
  int32_t start, end;
} impg_gpu_range_t;

/* AdjustedInterval without its CIGAR (src/impg.rs:225): query interval,
 * target interval.  q_first > q_last encodes the reverse strand. */
typedef struct {
  uint32_t query_id;
  int32_t q_first, q_last;
  uint32_t target_id;
  int32_t t_first, t_last;
} impg_gpu_interval_t;

/* Scalars of query / query_transitive_{bfs,dfs} (src/impg_index.rs:26-94) with
 * the CLI defaults of src/main.rs:4259-4285. */
typedef struct {
  int32_t transitive;                  /* 0: Impg::query          (impg.rs:1852) */
  int32_t dfs;                         /* 1: query_transitive_dfs (impg.rs:2057); 0: _bfs (impg.rs:2311) */
  uint32_t max_depth;                  /* u16 in the reference; 0 = unlimited */
  int32_t min_transitive_len;          /* default 101 */
  int32_t min_distance_between_ranges; /* default 10 */
  int32_t min_output_length;           /* < 0 = None */
  double min_identity;                 /* NaN = None (min_gap_compressed_identity) */
  int32_t store_cigar;                 /* BED output never needs it (main.rs:7447) */
  int32_t multi_impg;                  /* 1: MultiImpg semantics (src/multi_impg.rs:495-595, :796-991): every step's hits
                                          sorted by (query_id, q.first, q.last, t.first, t.last), one worklist pop at a
                                          time (front = BFS, back = DFS), unclipped ranges, same-sequence hits skipped */
  int32_t original_sequence_coordinates; /* text writers only (--original-sequence-coordinates, main.rs:4370): a sequence
                                          named "base:START-END" is printed as "base" with START added to its
                                          coordinates (transform_coordinates_to_original, main.rs:4642-4678); PAF
                                          sequence lengths are then 0, as the reference prints them when it has no
                                          sequence files to ask (get_original_sequence_length, main.rs:4681-4704) */
} impg_gpu_params_t;

/* Order in which overlapping entries of one target are visited; it fixes the
 * emission order of hits (and, through the order-dependent visited-set update
 * of impg.rs:2471-2560, the content of transitive results). */
#define IMPG_ORDER_COITREES 0 /* coitrees 0.4 BasicCOITree::query order (restated) */
#define IMPG_ORDER_SORTED 1   /* ascending target start, ties in input order */

const char *impg_gpu_last_error(void);
int impg_gpu_device_count(void);

/* ---- index: replaces Impg::from_multi_alignment_records (impg.rs:1535-1652),
 *      ForestMap (forest_map.rs:6-32) and the per-target coitrees ------------ */
int impg_gpu_index_create(const impg_gpu_record_t *records, size_t n_records,
                          const uint32_t *cigar_ops, size_t n_ops,
                          const int64_t *seq_len, uint32_t n_seq,
                          int bidirectional, int order_policy, int device,
                          impg_gpu_index_t **out);
/* The same from records of several alignment files: records[file_first_record[f] .. file_first_record[f+1])
 * (the last file ends at n_records) came from file f, as in `records_by_file` of
 * Impg::from_multi_alignment_records (impg.rs:1535).  Only MultiImpg semantics observe the split, and only
 * in the order of hits that agree on all five sort keys (multi_impg.rs:556-592): they stay file by file,
 * each in its own tree's visit order.  impg_gpu_index_create treats all records as one file. */
int impg_gpu_index_create_files(const impg_gpu_record_t *records, size_t n_records,
                                const uint32_t *cigar_ops, size_t n_ops, const int64_t *seq_len,
                                uint32_t n_seq, const uint64_t *file_first_record, uint32_t n_files,
                                int bidirectional, int order_policy, int device, uint32_t shard,
                                uint32_t n_shards, impg_gpu_index_t **out);

/* Same, parsing PAF files on the host the way paf.rs:118-194 does; sequence ids
 * are assigned in first-seen order over the files (query then target per line). */
int impg_gpu_index_create_from_paf(const char *const *paths, int n_paths,
                                   int bidirectional, int order_policy,
                                   int device, impg_gpu_index_t **out);
/* Keep only the entries whose target satisfies target_id % n_shards == shard
 * (multi-GPU sharding by target sequence; applied at create time). */
int impg_gpu_index_create_sharded(const impg_gpu_record_t *records, size_t n_records,
                                  const uint32_t *cigar_ops, size_t n_ops,
                                  const int64_t *seq_len, uint32_t n_seq,
                                  int bidirectional, int order_policy, int device,
                                  uint32_t shard, uint32_t n_shards,
                                  impg_gpu_index_t **out);
int impg_gpu_index_create_from_paf_sharded(const char *const *paths, int n_paths,
                                           int bidirectional, int order_policy, int device,
                                           uint32_t shard, uint32_t n_shards,
                                           impg_gpu_index_t **out);
/* A built index as one file: the role of the reference's `.impg` file (writer impg.rs:1655-1721, reader
 * :1787-1850; `impg index` / the implicit index cache of `impg query`, main.rs:11321-11386): pay for
 * parsing and tokenising once.  The file holds the device arrays as they sit in HBM plus the sequence
 * table (so the alignment files are not needed again, unlike with `.impg`, which stores byte offsets into
 * them); it is this library's own layout ("IMPGHBM1"), tied to the build's tile constants, and NOT the
 * reference's IMPGIDX2 format.  A sharded index saves / loads its shard.  load: IMPG_E_INVALID for a
 * foreign or damaged file, IMPG_E_UNSUPPORTED for a layout version this build does not read. */
int impg_gpu_index_save(const impg_gpu_index_t *, const char *path);
int impg_gpu_index_load(const char *path, int device, impg_gpu_index_t **out);
void impg_gpu_index_destroy(impg_gpu_index_t *);

/* seq_index() (seqidx.rs), target_ids(), num_targets() (impg_index.rs:105-113) */
uint32_t impg_gpu_num_seqs(const impg_gpu_index_t *);
const char *impg_gpu_seq_name(const impg_gpu_index_t *, uint32_t id); /* NULL if created without names */
int64_t impg_gpu_seq_len(const impg_gpu_index_t *, uint32_t id);
int64_t impg_gpu_seq_id(const impg_gpu_index_t *, const char *name);  /* -1 = unknown */
size_t impg_gpu_num_targets(const impg_gpu_index_t *);
size_t impg_gpu_target_ids(const impg_gpu_index_t *, uint32_t *out, size_t cap);
size_t impg_gpu_num_entries(const impg_gpu_index_t *);
size_t impg_gpu_num_records(const impg_gpu_index_t *);
size_t impg_gpu_device_bytes(const impg_gpu_index_t *);

/* Tunables: "pair_budget" (max candidate pairs held in HBM per level; a batch
 * whose level exceeds it is split by ranges, queries being independent) and
 * "chunk_ranges" (initial ranges per chunk, 0 = whole batch), "locality_min"
 * (frontier size from which the projection kernel walks the hit slots in the order
 * of the ranges' windows in the entry array -- a cache-locality order; results
 * are identical either way; 0 = never). */
int impg_gpu_set_option(impg_gpu_index_t *, const char *key, int64_t value);

/* Visit rank of the sorted positions 0..n-1 of an n-entry target under an order
 * policy (what the index stores per entry); host-only, no GPU needed. */
int impg_gpu_visit_rank(uint32_t n, int order_policy, uint32_t *rank_out);

/* ---- queries ------------------------------------------------------------ */
/* Batch form of ImpgIndex::query / query_transitive_bfs / _dfs: one independent
 * query per range (each with its own visited set).  Results are grouped by
 * range, in the reference's emission order, self interval(s) first.
 * Unknown target or a target without alignments => self interval only
 * (impg.rs:1896).  Requires start < end. */
int impg_gpu_query_batch(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n,
                         const impg_gpu_params_t *params, impg_gpu_results_t **out);
/* Single range == batch of one; what a per-call `impl ImpgIndex` binds. */
/* masked_regions: Option<&FxHashMap<u32, SortedRanges>> of query_transitive_bfs / query_transitive_dfs
 * (impg.rs:2062, :2316; multi_impg.rs:692, :727, :801), as partition.rs:250-256, :364, :380 passes it.
 * One map for the whole batch: every range starts from its own clone of it (impg.rs:2077-2081), so a
 * batch under one mask equals the reference's per-range calls with the same map.  The map is given as
 * n_seqs entries in strictly ascending sequence-id order: SortedRanges.sequence_length, and
 * ranges[2*range_off[i] .. 2*range_off[i+1]) as (start, end) pairs, sorted, disjoint and non-touching
 * (the SortedRanges invariant).  Every SortedRanges has min_distance 0, as partition builds them.
 * A sequence absent from the map follows the reference: Impg gives it a set of length 0 (visited_entry,
 * impg.rs:2048-2053: its hits are reported but never expanded), MultiImpg its real length, except the
 * range's own target (entry().or_default(), multi_impg.rs:827-830: no result at all).
 * A range can now have several self intervals, or none; they lead its results in ascending order.
 * mask == NULL is impg_gpu_query_batch.  Only the transitive queries take a mask (IMPG_E_INVALID). */
typedef struct {
  uint32_t n_seqs;
  const uint32_t *seq_id;          /* [n_seqs] */
  const int32_t *sequence_length;  /* [n_seqs] */
  const uint64_t *range_off;       /* [n_seqs + 1] */
  const int32_t *ranges;           /* [2 * range_off[n_seqs]] */
} impg_gpu_mask_t;
int impg_gpu_query_batch_masked(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n,
                                const impg_gpu_params_t *params, const impg_gpu_mask_t *mask,
                                impg_gpu_results_t **out);
/* subset_filter: Option<&SubsetFilter> of the three trait methods (`--subset-sequence-list`).  The name
 * matching (SubsetFilter::matches, subset_filter.rs:23-60: exact / coordinate-stripped / sample / haplotype
 * keys) is host string work and stays with the host: it hands over its verdict per sequence id,
 * subset_keep[num_seqs], non-zero = matches.  A hit is kept iff its query sequence is the range's own
 * target or subset_keep says so -- while exploring for the transitive queries (impg.rs:2176-2185,
 * :2430-2439; multi_impg.rs:888-896: a dropped hit is neither reported nor expanded), after the query
 * otherwise (perform_query, main.rs:11693-11696).  mask and subset_keep may each be NULL. */
int impg_gpu_query_batch_filtered(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n,
                                  const impg_gpu_params_t *params, const impg_gpu_mask_t *mask,
                                  const uint8_t *subset_keep, impg_gpu_results_t **out);
/* The name matching for hosts that do not bring their own SubsetFilter: list_text is the content of a
 * `--subset-sequence-list` file (one name per line, '#' comments; parse_subset_filter, subset_filter.rs:117-141);
 * keep_out[i] = SubsetFilter::matches(names[i]) (:23-60: exact, without ":start-end", or by the
 * sample / sample+haplotype key of "S_hapN…" and "S#N#…" names, :143-176).  *n_entries = SubsetFilter::entry_count
 * (the reference refuses a list with 0 entries, :76-81).  Host-only: needs no device. */
int impg_gpu_subset_keep(const char *list_text, size_t len, const char *const *names, size_t n, uint8_t *keep_out,
                         size_t *n_entries);
int impg_gpu_query(impg_gpu_index_t *, uint32_t target_id, int32_t start, int32_t end,
                   const impg_gpu_params_t *params, impg_gpu_results_t **out);

size_t impg_gpu_results_num_ranges(const impg_gpu_results_t *);
size_t impg_gpu_results_total(const impg_gpu_results_t *);
/* offsets[n_ranges+1] into intervals[]; both owned by the results object */
const uint64_t *impg_gpu_results_offsets(const impg_gpu_results_t *);
const impg_gpu_interval_t *impg_gpu_results_intervals(const impg_gpu_results_t *);
/* store_cigar: the Vec<CigarOp> of every interval (impg.rs:1870-1872, :2878-2886),
 * packed ops at cigar_ops[cigar_offsets[i] .. cigar_offsets[i+1]); NULL when the
 * query ran with store_cigar = 0 */
const uint64_t *impg_gpu_results_cigar_offsets(const impg_gpu_results_t *);
const uint32_t *impg_gpu_results_cigar_ops(const impg_gpu_results_t *);
/* number of Some(..) projections (self intervals excluded) = the work unit of BASELINE.md */
uint64_t impg_gpu_results_projected(const impg_gpu_results_t *);
/* wall seconds the call spent in the engine (lookup / projection / update on the GPU) and in copying the hit
 * slots back and assembling them into per-range result lists on the host */
void impg_gpu_results_timing(const impg_gpu_results_t *, double *engine_s, double *assemble_s);
void impg_gpu_results_free(impg_gpu_results_t *);

/* Throughput form: same computation, results stay in HBM; returns per-range
 * hit counts and an order-independent 64-bit checksum of each range's hits
 * (either may be NULL), the number of projections, and per-stage times. */
typedef struct {
  uint64_t projected;      /* accepted projections */
  uint64_t pairs;          /* (range, entry) candidate pairs examined */
  uint64_t frontier_ranges;/* frontier ranges looked up over all levels */
  uint32_t levels;
  float ms_total;          /* HIP-event time of the whole call on the engine stream */
  float ms_lookup, ms_project, ms_update; /* per-stage kernel time (HIP events) */
  uint64_t project_launches;              /* launches of the projection kernel */
} impg_gpu_stats_t;
int impg_gpu_query_batch_stats(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n,
                               const impg_gpu_params_t *params, uint64_t *per_range_count,
                               uint64_t *per_range_checksum, impg_gpu_stats_t *stats);
/* Same with the ranges already resident in HBM (device pointer). */
int impg_gpu_query_batch_stats_dev(impg_gpu_index_t *, const impg_gpu_range_t *d_ranges, size_t n,
                                   const impg_gpu_params_t *params, uint64_t *per_range_count,
                                   uint64_t *per_range_checksum, impg_gpu_stats_t *stats);

/* ---- BED: merge_adjusted_intervals_gap_2d + merge_query_adjusted_intervals +
 *      output_results_bed (main.rs:12858-13011, :12474-12560, :11849-11892) ---- */
/* In-place merge of one range's results as output_results_bed does for BED
 * (all CIGARs empty).  Returns the new count. */
long impg_gpu_bed_merge(impg_gpu_interval_t *iv, size_t n, int32_t merge_distance,
                        int merge_strands);
/* Render results as BED text.  range_names[i] is BED column 4 of range i.
 * The non-transitive min_output_length retain of perform_query
 * (main.rs:11682-11688) is applied here.  *text is malloc'ed; free() it. */
int impg_gpu_results_bed(const impg_gpu_results_t *, const impg_gpu_index_t *,
                         const char *const *range_names, const impg_gpu_params_t *params,
                         int32_t merge_distance, char **text, size_t *len);

/* ---- PAF / BEDPE: results.remove(0) + merge_adjusted_intervals (CIGAR-faithful
 *      merge: contiguity, identical overlap, gaps <= -d) + output_results_paf /
 *      output_results_bedpe with gi:f / bi:f (main.rs:7472-7496, :11894-12103,
 *      :12563-12845, :13014-13180).  The results must come from a query with
 *      store_cigar = 1.  *text is malloc'ed; free() it. ------------------------ */
#define IMPG_OUT_PAF 0
#define IMPG_OUT_BEDPE 1
int impg_gpu_results_paf(const impg_gpu_results_t *, const impg_gpu_index_t *,
                         const char *const *range_names, const impg_gpu_params_t *params,
                         int32_t merge_distance, int format, char **text, size_t *len);

/* ---- host-side ingest helpers (paf.rs, partition.rs parsers) -------------- */
/* parse_cigar_to_delta (impg.rs:2935-2950): returns #ops or <0 */
long impg_gpu_parse_cigar(const char *cigar, size_t len, uint32_t *ops_out, size_t cap);
/* parse_target_range (partition.rs:1752-1763) */
/* parse_subsequence_coordinates (main.rs:4642-4659): "base:START-END" -> 1, base name and START;
 * 0 when the name carries no parsable coordinates.  Host-only. */
int impg_gpu_parse_subsequence(const char *seq_name, char *base_out, size_t base_cap, int32_t *start_offset);
int impg_gpu_parse_target_range(const char *s, char *name_out, size_t name_cap,
                                int32_t *start, int32_t *end);

/* ---- stage API: one BFS hop split at its exchange points, for a host that
 *      shards the index by target across GPUs.  All
 *      pointers are DEVICE pointers owned by the caller unless noted. -------- */
/* A frontier record: query range `qidx` (index into the caller's batch) wants
 * [start,end) on target_id looked up.  16 B. */
typedef struct {
  uint32_t target_id;
  int32_t start, end;
  uint32_t qidx;
} impg_gpu_frontier_t;
/* A hit: projection of frontier record `fidx` through one alignment. 28 B SoA
 * on device; this AoS form is what crosses ranks. */
typedef struct {
  uint32_t fidx;     /* index of the frontier record in the array passed in */
  uint32_t query_id; /* 0xFFFFFFFF = projection returned None (slot unused) */
  int32_t q_first, q_last, t_first, t_last;
  uint32_t order;    /* visit position within the frontier record */
  uint32_t pad;
} impg_gpu_hit_t;
/* lookup: writes counts[n] (overlapping entries of this shard per record) and
 * returns their sum in *total. */
int impg_gpu_stage_count(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n,
                         int transitive, uint32_t *d_counts, uint64_t *total);
/* project: fills d_hits[total] (slot order = frontier order x visit order); d_hits may be
 * NULL when only *accepted is wanted (the last level of a counting run). */
int impg_gpu_stage_project(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n,
                           int transitive, const impg_gpu_params_t *params,
                           impg_gpu_hit_t *d_hits, uint64_t total, uint64_t *accepted);

/* The same pair of calls on 16-byte hit records {fidx, query_id, q_first, q_last}: all the visited-set
 * update reads.  For runs that only need the closure's frontier and counts (no result rows at home),
 * it halves the bytes the owners send back. */
typedef struct {
  uint32_t fidx;
  uint32_t query_id; /* 0xFFFFFFFF = projection returned None */
  int32_t q_first, q_last;
} impg_gpu_hit16_t;
int impg_gpu_stage_project16(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n,
                             int transitive, const impg_gpu_params_t *params,
                             impg_gpu_hit16_t *d_hits, uint64_t total, uint64_t *accepted);

/* reorder: the home side of a hop.  d_hits[n] (impg_gpu_hit_t: words_per_hit = 8, or impg_gpu_hit16_t: 4;
 * fidx = index into the home frontier of n_frontier records) arrived grouped by owner rank, every
 * owner's block in ascending fidx.  d_out receives them in ascending fidx, order within one fidx kept
 * (what a stable sort by fidx gives): a frontier record lives on exactly one owner, so records of one
 * fidx are already contiguous and a counting pass replaces the sort.  IMPG_E_INVALID if a fidx is out
 * of range or shows up in two separate runs. */
int impg_gpu_stage_reorder(impg_gpu_index_t *, const void *d_hits, size_t n, uint32_t words_per_hit,
                           size_t n_frontier, void *d_out);

/* route: stable partition of a frontier by owner rank (target_id % world).  d_out[n]
 * receives the records grouped by owner, in their original order within a group,
 * with qidx replaced by the record's index in d_frontier (the home index an owner
 * echoes back in impg_gpu_hit_t.fidx); counts[world] (HOST) receives the group sizes. */
int impg_gpu_stage_route(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n, uint32_t world,
                         impg_gpu_frontier_t *d_out, uint64_t *counts);

/* Home-side steps of a sharded transitive batch (visited sets live where the
 * query lives).  stage_begin resets the visited sets to the batch's own ranges
 * (impg.rs:2337-2340), writes the self intervals (qidx = range index) to
 * d_self_out[n] and the level-0 frontier to d_frontier_out (cap n).
 * stage_update replays hits -- sorted by fidx, fidx indexing d_frontier, hits of
 * one record in visit order -- against the visited sets (impg.rs:2471-2560) and
 * builds the next frontier (impg.rs:2566-2584); stage_next_frontier copies it. */
int impg_gpu_stage_begin(impg_gpu_index_t *, const impg_gpu_range_t *d_ranges, size_t n,
                         const impg_gpu_params_t *params, impg_gpu_frontier_t *d_frontier_out,
                         uint64_t *n_frontier, impg_gpu_frontier_t *d_self_out);
int impg_gpu_stage_update(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n_frontier,
                          const impg_gpu_hit_t *d_hits, size_t n_hits, const impg_gpu_params_t *params,
                          uint64_t *n_next);
int impg_gpu_stage_update16(impg_gpu_index_t *, const impg_gpu_frontier_t *d_frontier, size_t n_frontier,
                            const impg_gpu_hit16_t *d_hits, size_t n_hits, const impg_gpu_params_t *params,
                            uint64_t *n_next);
int impg_gpu_stage_next_frontier(impg_gpu_index_t *, impg_gpu_frontier_t *d_out, size_t cap);
/* HIP-event time accumulated by the stage calls since the last reset:
 * ms[0] lookup (count+emit), ms[1] projection kernel, ms[2] visited update;
 * launches = projection launches. */
int impg_gpu_stage_timing(impg_gpu_index_t *, float *ms3, uint64_t *launches, int reset);

/* ---- synthetic workload generators (SplitMix64) ----- */
/* Fills records / ops for `n_records` synthetic alignments (200-op CIGARs by
 * default).  Call with ops == NULL to size: *n_ops_out receives the op count. */
int impg_synth_paf(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len,
                   int32_t target_span, uint32_t n_blocks, impg_gpu_record_t *records,
                   uint32_t *ops, size_t ops_cap, size_t *n_ops_out);
/* Writes the same alignments as PAF text (PanSN names gNNN#H#chr1). */
int impg_synth_paf_text(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len,
                        int32_t target_span, uint32_t n_blocks, const char *path);
int impg_synth_seq_name(uint32_t id, char *out, size_t cap);
int impg_synth_bed(uint64_t seed, size_t n, uint32_t n_seq, int32_t seq_len, int32_t range_len,
                   impg_gpu_range_t *out);

#ifdef __cplusplus
}
#endif
#endif
