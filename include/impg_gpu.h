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
#define IMPG_E_CANCELLED (-6)   /* a callback of the caller's asked the call to stop */

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
  int32_t consider_strandness;         /* BED writer only (--consider-strandness, main.rs:4380): 1 keeps the two strands
                                          apart in the query-axis merge (merge_strands_for_output, main.rs:4395-4409) */
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
                                int bidirectional, int order_policy, int device, impg_gpu_index_t **out);

/* ---- tracepoint alignments: approximate mode (approximate_mode = true of the trait, impg_index.rs:26-94;
 *      scan_overlapping_tracepoints + project_overlapping_interval_fast, impg.rs:646-823, :1317-1533).
 * The .1aln / .tpa readers stay with the host (onealn.rs, tpa crate): it hands every alignment over as its
 * tracepoints -- n_segs target deltas at tracepoints[seg_off ..] -- plus, per TracepointModeData (onealn.rs:772-782),
 *   Standard: query_deltas[seg_off ..] and the file's max_complexity (per-segment diffs are estimated from it),
 *   FASTGA:   diffs[seg_off ..], the file's trace_spacing and the alignment's query_contig_start (the first query
 *             delta is ((query_contig_start / spacing) + 1) * spacing - query_contig_start, impg.rs:726-735).
 * An index built this way answers every query in approximate mode: the query interval of a hit is interpolated
 * inside the first and the last overlapping trace segment (one f64 division and rounding each, bit-for-bit the
 * reference's arithmetic), its target interval is the (clipped) range itself, and the identity filter sees the
 * segment statistics ("N= MX").  Per alignment the index keeps prefix sums of the four per-segment quantities, so
 * a projection is two binary searches instead of a scan.  Tracepoints and query deltas must be non-negative (they
 * are in files FASTGA / the tracepoints crate write; the reference takes abs() of the former in places) and sum
 * below 2^31 per alignment: IMPG_E_UNSUPPORTED otherwise.  store_cigar is not offered (IMPG_E_UNSUPPORTED); the
 * exact mode of these files needs the sequences and a WFA realignment and stays with the host. */
typedef struct {
  uint32_t query_id, target_id;
  int32_t query_start, query_end, target_start, target_end;
  uint64_t seg_off; /* first segment of this alignment in the pools */
  uint32_t n_segs;
  uint32_t strand;  /* 0 '+', 1 '-' */
  int64_t query_contig_start; /* FASTGA only */
} impg_gpu_tp_record_t;
typedef struct {
  int32_t fastga;         /* 0 Standard, 1 FASTGA */
  int32_t trace_spacing;  /* FASTGA */
  int32_t max_complexity; /* Standard */
} impg_gpu_tp_mode_t;
int impg_gpu_index_create_tracepoints(const impg_gpu_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                      const int32_t *query_deltas /* Standard, else NULL */,
                                      const int32_t *diffs /* FASTGA, else NULL */, size_t n_segs_total,
                                      const impg_gpu_tp_mode_t *mode, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                      int order_policy, int device, impg_gpu_index_t **out);

/* The same index sharded by target sequence over the GPUs of this process (see impg_gpu_index_create_multi below): one
 * handle, queries answered in approximate mode by the shards that own the targets. */
int impg_gpu_index_create_tracepoints_multi(const impg_gpu_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                            const int32_t *query_deltas, const int32_t *diffs, size_t n_segs_total,
                                            const impg_gpu_tp_mode_t *mode, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                            int order_policy, const int *devices, int n_dev, int lanes, impg_gpu_index_t **out);

/* Same, parsing PAF files on the host the way paf.rs:118-194 does; sequence ids
 * are assigned in first-seen order over the files (query then target per line). */
int impg_gpu_index_create_from_paf(const char *const *paths, int n_paths,
                                   int bidirectional, int order_policy,
                                   int device, impg_gpu_index_t **out);
/* A built index as one file: the role of the reference's `.impg` file (writer impg.rs:1655-1721, reader
 * :1787-1850; `impg index` / the implicit index cache of `impg query`, main.rs:11321-11386): pay for
 * parsing and tokenising once.  The file holds the device arrays as they sit in HBM plus the sequence
 * table (so the alignment files are not needed again, unlike with `.impg`, which stores byte offsets into
 * them); it is this library's own layout ("IMPGHBM1"), tied to the build's tile constants, and NOT the
 * reference's IMPGIDX2 format.  A rank's shard of a sharded index is saved the same way (one file per rank, the caller
 * names it; it also holds the world size, the rank and the target -> rank map) and comes back with
 * impg_gpu_index_load_rank on a communicator of the same world and rank; a multi handle writes its front file at `path`
 * and its shards at path.shard<k>of<n> and comes back with impg_gpu_index_load_multi -- so the ranks of a job parse the
 * alignments once, not once per rank per start (the role of impg.rs:1655-1850).  load: IMPG_E_INVALID for a
 * foreign or damaged file (or a shard handed to the plain load / the wrong rank), IMPG_E_UNSUPPORTED for a layout
 * version this build does not read. */
int impg_gpu_index_save(const impg_gpu_index_t *, const char *path);
/* The reference's own index file ("IMPGIDX2", or the unidirectional "IMPGIDX1"; writer impg.rs:1655-1721, reader
 * :1787-1850 + :1724-1767; bincode-2 standard encoding): sequence table and every target's intervals come from the
 * file, in the file's order (the order the reference rebuilds its trees from, hence its tie order among equal
 * starts); the CIGARs are read from the alignment files -- the same files, in the same order, the index was built
 * with (impg.rs:1789, :1844; plain-text PAF: offsets into compressed PAF are BGZF virtual offsets,
 * IMPG_E_UNSUPPORTED) -- and tokenised once.  An existing `.impg` cache therefore opens without `impg index`
 * being re-run.  IMPG_E_INVALID for a damaged file or CIGAR text that no longer parses at its offset. */
int impg_gpu_index_load_impg(const char *impg_path, const char *const *alignment_files, int n_files, int order_policy,
                             int device, impg_gpu_index_t **out);
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
/* HBM the index's arrays hold.  Grows once, by 128 bytes per CIGAR tile, when a query first sets min_identity
 * (min_gap_compressed_identity, impg_index.rs:31): the identity lines -- a third of a full index, read by that filter
 * only -- are built on the device at that point (IMPG_E_OOM if they do not fit); IMPG_IDENTITY_LINES=1 in the
 * environment builds them with the index. */
size_t impg_gpu_device_bytes(const impg_gpu_index_t *);
/* 1: the index was built from tracepoints and answers every query in the reference's approximate mode
 * (approximate_mode = true of the trait methods, src/impg_index.rs:34, :93); 0: built from CIGARs, exact mode.
 * The mode is a property of the index here: a host mirroring the trait refuses a call whose approximate_mode
 * disagrees (IMPG_E_UNSUPPORTED) rather than answer in the other mode. */
int impg_gpu_index_approximate(const impg_gpu_index_t *);

/* Tunables: "pair_budget" (max candidate pairs held in HBM per level; a batch
 * whose level exceeds it is split by ranges, queries being independent) and
 * "chunk_ranges" (initial ranges per chunk, 0 = whole batch), "locality_min"
 * (frontier size from which the projection kernel walks the hit slots in the order
 * of the ranges' windows in the entry array -- a cache-locality order; results
 * are identical either way; 0 = never), "free_slot_order" (1, the default: runs
 * that keep no level -- impg_gpu_query_batch_stats -- lay their hit slots out in
 * that order too; 0: always the reference's slot order; counts and checksums are
 * identical either way),
 * "walk_kernel" (the per-query walk: 0 never, 1 -- the default -- DFS batches of any size and those depth-limited
 * (max_depth >= 2) BFS batches of <= 64 ranges that get at least two workgroups a query out of the launch's share of
 * the compute units -- CUs / max_engines, i.e. <= 32 ranges on a 256-CU device with four engines a handle --, 2 also
 * every other BFS batch of <= 64 ranges) and "walk_members" (workgroups per query of the walk's grid form, which
 * shares a depth-limited BFS's last level out: 0 = as many as fit the share, at most 32; N > 1 = at most N, itself at
 * most 64; 1 = none),
 * "segment_groups" (1, the default: the visited update orders a level's hits by (query, hit sequence) query by query --
 * a query's ranges run by run in frontier order, a counting sort by sequence inside the query; 0: with the library's
 * stable radix sort; results are identical either way) and "segment_parts" (0, the default: a query whose level holds
 * more hits than one wave should take is cut into slices of its frontier ranges, as many as the level's size asks for;
 * N = that many slices on every level that groups by segments -- for tests; results are identical),
 * "fuse_final_level" (1, the default: the final level of such a run -- no update follows, no row is kept --
 * takes its (range, entry) pairs straight from the lookup's per-range windows inside the projection kernel; the emit
 * pass and its pair lists are skipped; counts and checksums are identical either way).
 * Actions rather than settings, so that a process's FIRST call costs what its later ones do: "prewarm_result_bytes" = N
 * pins a host block of N bytes into the result pool now (a 5 GB result pins its block inside the first call
 * otherwise, ~0.3-1 s); "prewarm_walk" = 1 / 2 allocates the per-query walk's slabs (1: the per-call / small-batch BFS
 * shape; 2: also the DFS batch's, ~15 GB) on the index's first engine. */
int impg_gpu_set_option(impg_gpu_index_t *, const char *key, int64_t value);
/* Read-only counters of an index handle (no reference counterpart; they tell a test, or an operator, which engine
 * answered): "walk_launches" = per-query walk launches that answered their batch (walk_device.inc: DFS batches, small
 * depth-limited BFS batches incl. masked ones -- the shape of partition.rs:359-391), "walk_fallbacks" = launches whose
 * batch the batch engine had to run again (a query outgrew its slab), "walk_members" = workgroups per query of the
 * last grid-form launch (1: not the grid form); how the visited updates grouped their hits: "segment_sliced_levels" =
 * levels whose queries were cut into slices of their frontier ranges, "segment_retries" = levels counted a second time
 * because one query held more hits than a wave should take, "segment_library_levels" = levels that went through the
 * library's radix sort instead. */
int impg_gpu_get_counter(const impg_gpu_index_t *, const char *key, int64_t *value_out);
/* Large result arrays live in pinned host blocks that are recycled through a process-wide pool (at most
 * IMPG_PINNED_POOL_BYTES, default 6 GiB, are kept when results are freed).  Gives pooled blocks back to the system
 * until at most keep_bytes are held; returns the number of bytes freed. */
uint64_t impg_gpu_host_pool_trim(uint64_t keep_bytes);

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
/* The same rows for a batch too big for one result object (the 100 000-range headline batch returns 2.1 x 10^9 rows,
 * 51 GB): the batch is cut into chunks of chunk_ranges ranges (0 = 8192; a chunk that outgrows the pair budget or
 * max_block_bytes of rows -- 0 = 2.5 GiB, two of which the library's pinned pool keeps between calls -- is halved, and so
 * are the chunks after it), two engines compute them in turn, and `cb` receives every chunk
 * as a results object (the impg_gpu_results_* accessors; its range i is ranges[first_range + i]) valid until the
 * callback returns -- IN RANGE ORDER, one call at a time, from a thread of the library, while the next chunk is
 * computed and copied: what a caller that prints or folds range by range consumes (main.rs:7435-7470).  Host memory:
 * two pinned blocks of at most max_block_bytes (+ the CIGAR pools under store_cigar).  A non-zero return of the
 * callback stops the stream (IMPG_E_CANCELLED).  mask / subset_keep as in impg_gpu_query_batch_filtered (NULL = none).
 * On a sharded index the chunks run one after the other through the collective call (no overlap). */
typedef int (*impg_gpu_stream_cb)(void *ctx, const impg_gpu_results_t *chunk, size_t first_range);
int impg_gpu_query_batch_stream(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                                const impg_gpu_mask_t *mask, const uint8_t *subset_keep, size_t chunk_ranges, size_t max_block_bytes,
                                impg_gpu_stream_cb cb, void *ctx, uint64_t *projected_out);
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
  float ms_exchange;                      /* sharded index: wall time inside the transport (all-gathers + all-to-all-v) */
} impg_gpu_stats_t;
int impg_gpu_query_batch_stats(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n,
                               const impg_gpu_params_t *params, uint64_t *per_range_count,
                               uint64_t *per_range_checksum, impg_gpu_stats_t *stats);
/* Same with the ranges already resident in HBM (device pointer; not for a multi handle, whose ranges span GPUs). */
int impg_gpu_query_batch_stats_dev(impg_gpu_index_t *, const impg_gpu_range_t *d_ranges, size_t n,
                                   const impg_gpu_params_t *params, uint64_t *per_range_count,
                                   uint64_t *per_range_checksum, impg_gpu_stats_t *stats);

/* ---- rows left in HBM ------------------------------------------------------------------------------------
 * The same queries with every result row left ON THE DEVICE, complete and attributable, for a consumer that lives
 * there too (the BED merges of this library are one; a caller's own kernels another): no row crosses PCIe, no
 * per-range statistics are folded.  What the reference's sequential phase pushes per hit -- the AdjustedInterval with
 * the frontier range's target as its target metadata (impg.rs:2491-2503, :1920-1923) -- is here one slot of a PART:
 * the hits of one BFS level (level 0 = the hits of the ranges themselves; a plain query has only that) of one chunk of
 * the batch, as device arrays of n_slots entries
 *     query_id[s*k] the hit's query sequence; 0xFFFFFFFF = the projection returned None (impg.rs:2874-2877), or the
 *                   subset filter dropped the hit: not a row
 *     coords[4k..]  q_first, q_last, t_first, t_last (q_first > q_last: reverse strand)
 *     source[s*k]   index into frontier[]: the frontier record the hit was found with
 *                   (s = slot_stride: 1, or 2 where the part keeps a slot's query_id and source side by side as one
 *                   8-byte pair -- the fused final level, which writes them with one store)
 *     frontier[j]   {target_id, start, end, range_idx}: the row's target sequence (the target interval's metadata) and
 *                   the range of the batch it belongs to, as ranges[first_range + range_idx]
 * i.e. SURVEY-style 24 bytes per slot (query_id, four coordinates, source) with range_idx and target_id one
 * look-up away.  Slot order within a part is unspecified (IMPG_ROWS_ATTRIBUTED): the final level is written entry by
 * entry, which is what makes it fast; a caller that needs the reference's emission order asks impg_gpu_query_batch /
 * _stream for it.  Transitive rows shorter than min_output_length (impg.rs:2482-2504) are NOT removed from the
 * slots: compare |q_last - q_first| as the reference does.  The self interval of a range is the range itself
 * (impg.rs:1864-1880, :2345-2363) and is not stored.
 * Takes Impg::query and query_transitive_bfs on a single-GPU CIGAR or tracepoint index (IMPG_E_UNSUPPORTED: DFS,
 * MultiImpg worklists, store_cigar, sharded handles).  ranges_on_device != 0: `ranges` is a device pointer.
 * The handle holds one of the index's engines (max 4) and its HBM until impg_gpu_device_rows_free. */
typedef struct impg_gpu_device_rows impg_gpu_device_rows_t;
#define IMPG_ROWS_ATTRIBUTED 0
/* IMPG_ROWS_ORDERED: the trait's own rows instead -- one part per chunk: rows[n_slots] (impg_gpu_interval_t, no holes)
 * grouped by range in the reference's emission order, self interval(s) first, offsets[n_ranges + 1] (u32) into them:
 * exactly what impg_gpu_query_batch returns, left in HBM (transitive rows below min_output_length removed). */
#define IMPG_ROWS_ORDERED 1
/* IMPG_ROWS_ORDERED_SLOTS: the same grouping and order with every SLOT at its place -- a (range, alignment) pair whose
 * projection returned None (impg.rs:2874-2877: 7 in 10^5 at the headline), or whose transitive row is shorter than
 * min_output_length, stays as a HOLE row, query_id = 0xFFFFFFFF, instead of moving every row behind it up: skip those
 * (offsets[] counts them).  Where a row goes then follows from the lookups' counts alone, so the final level's kernel
 * writes its rows itself where they belong -- no scans and scatters over 2 x 10^9 rows: the headline batch's rows are in
 * HBM in emission order in ~49 ms where IMPG_ROWS_ORDERED takes 109 (a lane per place, a range's
 * places in visit order, so that a wave's rows are one stretch of the output). */
#define IMPG_ROWS_ORDERED_SLOTS 2
typedef struct {
  uint32_t target_id;
  int32_t start, end;
  uint32_t range_idx; /* relative to the part's first_range */
} impg_gpu_frontier_t;
typedef struct {
  size_t first_range, n_ranges; /* the chunk of the batch the part belongs to (chunks: "chunk_ranges" / "pair_budget") */
  uint32_t level;               /* BFS level of the part's hits */
  uint32_t n_frontier;
  uint32_t slot_stride;         /* 4-byte words between consecutive slots' query_id (and source) */
  uint64_t n_slots;
  const uint32_t *query_id;     /* [slot_stride * n_slots]  device */
  const int32_t *coords;        /* [4 * n_slots]            device, 16-byte aligned */
  const uint32_t *source;       /* [slot_stride * n_slots]  device */
  const impg_gpu_frontier_t *frontier; /* [n_frontier] device */
  const impg_gpu_interval_t *rows;     /* IMPG_ROWS_ORDERED[_SLOTS]: [n_slots] device (the fields above are NULL / 0) */
  const uint32_t *offsets;             /* IMPG_ROWS_ORDERED[_SLOTS]: [n_ranges + 1] device */
} impg_gpu_device_part_t;
int impg_gpu_query_batch_device(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n, int ranges_on_device,
                                const impg_gpu_params_t *params, int layout, impg_gpu_device_rows_t **out);
size_t impg_gpu_device_rows_num_parts(const impg_gpu_device_rows_t *);
int impg_gpu_device_rows_part(const impg_gpu_device_rows_t *, size_t k, impg_gpu_device_part_t *out);
/* projections, candidate pairs and per-stage times of the call (impg_gpu_stats_t as above) */
void impg_gpu_device_rows_stats(const impg_gpu_device_rows_t *, impg_gpu_stats_t *stats);
/* ordered layout: HIP-event milliseconds the row placement took on top of stats->ms_total */
float impg_gpu_device_rows_place_ms(const impg_gpu_device_rows_t *);
/* Verification: the per-range counts and order-independent checksums of impg_gpu_query_batch_stats, recomputed FROM
 * THE ROWS the call left in HBM (every slot attributed through source[] / frontier[]); either may be NULL. */
int impg_gpu_device_rows_check(impg_gpu_device_rows_t *, uint64_t *per_range_count, uint64_t *per_range_checksum);
void impg_gpu_device_rows_free(impg_gpu_device_rows_t *);

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

/* perform_query + output_results_bed for a whole batch in one call, with both merges ON THE DEVICE
 * (bed_device.hip): the hit slots never leave HBM; they are turned into rows, sorted, chained
 * (merge_adjusted_intervals_gap_2d) and swept (merge_query_adjusted_intervals) there, and only the merged rows --
 * 16 bytes each -- are formatted there as well; the text crosses PCIe in pinned pieces while the next piece is being
 * formatted.  The text is byte-identical to impg_gpu_query_batch_filtered +
 * impg_gpu_results_bed (what `impg query -o bed` prints: main.rs:7435-7470, :11849-11892).  range_names as in
 * impg_gpu_results_bed (NULL, or NULL entries, = "{name}:{start}-{end}").  subset_keep may be NULL.  seconds3, if
 * not NULL, receives the wall seconds of {engine, device-side merge + copy back, text}.  On an index sharded over GPUs the
 * merges and the text run on each range's home rank and the pieces are concatenated in the caller's order. */
int impg_gpu_query_batch_bed(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                             const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names, char **text,
                             size_t *len, double *seconds3);
/* The same, the text written to a file descriptor piece by piece instead of collected (what the CLI does: gigabytes
 * of BED never exist as one host buffer). */
int impg_gpu_query_batch_bed_fd(impg_gpu_index_t *, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                                const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names, int fd,
                                uint64_t *bytes_written, double *seconds3);

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

/* ---- multi-GPU: the index sharded by target sequence over the GPUs of one node ---------
 * Entries live with the rank that owns their target (targets bin-packed by entry count, heaviest first onto the
 * least loaded shard: impg_gpu_shard_assign); a record's ops are stored with its forward entry's owner and with its
 * reversed entry's owner.  Every query lives with its HOME rank -- visited sets, worklists and result order, the
 * sequential state of impg.rs:2471-2560 -- and each hop of the walk sends the frontier records to the owners of
 * their targets (all-to-all-v of 16-byte records) and brings the hits home (16- or 32-byte records), where they are
 * put back into frontier order x visit order before the visited-set update.  The whole loop runs inside this
 * library; the reference has no counterpart (its MultiImpg, multi_impg.rs:495-595, queries per-file indices
 * serially on one host).  Two ways to bring the ranks up:
 *
 *  (1) one process, n_dev GPUs: impg_gpu_index_create_multi / _from_paf_multi return ONE handle.  Every query
 *      entry point above works on it unchanged -- a Rust `impl ImpgIndex` gets the whole node for free.  The ranks
 *      are host threads; exchanges are direct peer copies over xGMI (hipMemcpyPeerAsync).  The batch's ranges are
 *      dealt to the ranks in contiguous blocks; results come back in the caller's order.
 *  (2) one process per GPU (torch.distributed.run / mpirun layout): every rank makes a communicator
 *      (impg_gpu_comm_create_rccl: RCCL send/recv groups over xGMI; rank 0 makes the ids with
 *      impg_gpu_comm_unique_id and the launcher carries them to the others -- or impg_gpu_comm_create_host: the
 *      host's own transport as two callbacks), builds its shard (impg_gpu_index_create_rank / _from_paf_rank),
 *      and then the query entry points are COLLECTIVE: every rank calls the same function with the same params
 *      and ITS OWN ranges (possibly none) and gets the results of its own ranges.
 *
 * `lanes` chunks of a batch ("chunk_ranges") are in flight at once, each on its own engine, stream, host thread
 * and communicator, so one lane's exchange overlaps the other lanes' kernels.
 * impg_gpu_index_save works on a shard and on a multi handle (see there); store_cigar, the
 * device-side BED call and tracepoint indexes work there as on one GPU (the ops follow the hits home).  projected / pairs /
 * stage times of a rank (2) count the work done ON THAT RANK's shard; a multi handle (1) reports the sum of
 * the work and the slowest rank's times. */
typedef struct impg_gpu_comm impg_gpu_comm_t;
#define IMPG_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* ids[lanes * IMPG_COMM_ID_BYTES]: one RCCL unique id per lane, made on one rank */
int impg_gpu_comm_unique_id(uint8_t *ids, int lanes);
int impg_gpu_comm_create_rccl(const uint8_t *ids, int lanes, int rank, int world, int device, impg_gpu_comm_t **out);
/* The host's transport.  Both callbacks move HOST memory and return 0 on success; they are called from the lane's
 * own thread (one transport per lane: collectives of different lanes must not share an ordering domain).
 *   allgather_u64: all[r * k + i] = value i of rank r.
 *   alltoallv: bytes [send_off[d], +send_bytes[d]) of `send` go to rank d; bytes from rank s land at recv_off[s]. */
typedef struct {
  void *ctx;
  int (*allgather_u64)(void *ctx, const uint64_t *mine, size_t k, uint64_t *all);
  int (*alltoallv)(void *ctx, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv,
                   const uint64_t *recv_off, const uint64_t *recv_bytes);
} impg_gpu_host_transport_t;
int impg_gpu_comm_create_host(const impg_gpu_host_transport_t *transports /* [lanes] */, int lanes, int rank, int world,
                              int device, impg_gpu_comm_t **out);
void impg_gpu_comm_destroy(impg_gpu_comm_t *); /* after the indexes that use it */
int impg_gpu_comm_info(const impg_gpu_comm_t *, int *rank, int *world, int *lanes, const char **kind /* "rccl", "host", ... */);
/* Collective self-check of one lane's transport before any index is built on it: an all-gather of known words and,
 * for a host transport, a ragged all-to-all-v of a known pattern over host memory (needs no GPU).
 * IMPG_E_IO if anything arrives wrong. */
int impg_gpu_comm_check(impg_gpu_comm_t *, int lane);
/* (2): this rank's shard.  Every rank passes the SAME records (each keeps what its targets need). */
int impg_gpu_index_create_rank(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                               const int64_t *seq_len, uint32_t n_seq, const uint64_t *file_first_record /* may be NULL */,
                               uint32_t n_files, int bidirectional, int order_policy, int device, impg_gpu_comm_t *comm,
                               impg_gpu_index_t **out);
int impg_gpu_index_create_from_paf_rank(const char *const *paths, int n_paths, int bidirectional, int order_policy,
                                        int device, impg_gpu_comm_t *comm, impg_gpu_index_t **out);
/* (1): one handle over n_dev GPUs of this process (a device may be listed more than once: its shards share it). */
int impg_gpu_index_create_multi(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                                const int64_t *seq_len, uint32_t n_seq, const uint64_t *file_first_record /* may be NULL */,
                                uint32_t n_files, int bidirectional, int order_policy, const int *devices, int n_dev,
                                int lanes, impg_gpu_index_t **out);
int impg_gpu_index_create_from_paf_multi(const char *const *paths, int n_paths, int bidirectional, int order_policy,
                                         const int *devices, int n_dev, int lanes, impg_gpu_index_t **out);
/* What impg_gpu_index_save wrote for a rank's shard / for a multi handle, back in HBM without the alignment files.
 * load_rank: collective in the sense that every rank loads its own file before the first query; the file's world and
 * rank must be the communicator's.  load_multi: n_dev must be the number of GPUs the handle was saved over. */
int impg_gpu_index_load_rank(const char *path, int device, impg_gpu_comm_t *comm, impg_gpu_index_t **out);
int impg_gpu_index_load_multi(const char *path, const int *devices, int n_dev, int lanes, impg_gpu_index_t **out);
/* The shard map: owner_out[t] = shard of target t given its entry count (host-only, deterministic). */
int impg_gpu_shard_assign(const uint64_t *entries_per_target, uint32_t n_seq, uint32_t n_shards, uint32_t *owner_out);
/* rank (-1 for a multi handle), world, lanes and the shard map of an index (1 / 0 for a plain one) */
int impg_gpu_index_shard_info(const impg_gpu_index_t *, int *rank, int *world, int *lanes, uint32_t *owner_out, size_t cap);
/* Where the hops of the queries since the last reset spent their wall time, by hop number within a batch (hop 8 and
 * later are counted with hop 8), summed over the lanes: out[(shard * 8 + hop) * 12 + field] with fields
 *   0 hops, then seconds: 1 bucketing the frontier by owner, 2 all-gather of the sizes (incl. the wait for the slowest
 *   rank), 3 frontier records to the owners, 4 the owner's lookup + projection (+ packing the hits), 5 all-gather of
 *   the hit counts, 6 hits (and CIGAR ops) home, 7 putting them back in frontier order; then 8 bytes of frontier
 *   records sent, 9 bytes of hits + ops sent, 10 frontier records received, 11 hits that came home.
 * One shard for a rank's index, n_dev for a multi handle, none for a plain index.  *n_out = doubles written (or
 * needed when out is NULL). */
int impg_gpu_index_hop_profile(impg_gpu_index_t *, double *out, size_t cap, int reset, size_t *n_out);

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
/* The non-uniform workload of `bench.py --workload skewed` as PAF text: log-normal target spans (1 kb ... the sequence,
 * median 8 kb: 20 ... 10^5 ops a CIGAR) and 1 % of the sequences (the first max(1, n_seq / 100)) chosen as target / as query with
 * probability 0.3 each, i.e. holding ~30 % of a bidirectional index's entries.  Same names and record format as
 * impg_synth_paf_text; *n_ops_out = CIGAR ops written. */
int impg_synth_skewed_paf_text(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len, const char *path, uint64_t *n_ops_out);
int impg_synth_bed(uint64_t seed, size_t n, uint32_t n_seq, int32_t seq_len, int32_t range_len,
                   impg_gpu_range_t *out);
/* Diagnostics: the engine's own sort of a level's lookup order (impg_amd/csrc/kernels.hip, order_scatter_kernel -- the place
 * of the reference's per-tree query order, which the engine is free to choose: interval_tree lookups commute) run on `n`
 * pseudo-random keys of `end_bit` bits on `device` and checked on the host: a permutation, keys non-decreasing, equal keys
 * in index order.  IMPG_OK or IMPG_E_INVALID with the first offending place in impg_gpu_last_error(). */
int impg_gpu_selftest_order_sort(int device, uint32_t n, unsigned end_bit, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
