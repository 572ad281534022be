/*
 * impg_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain CPU restatement of the reference
 * algorithm (pangenome/impg 0.5.0, Rust) used as the *checker* for the HIP
 * engine.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product (libimpg_gpu.so) never links or calls it.
 *
 * Parity status: the oracle is pinned by every known-answer test the reference
 * holds for this path (tests/test_oracle_kat.py).  The
 * reference itself cannot be compiled here (no Rust toolchain), and the
 * traversal order of the third-party crate coitrees 0.4.0 (not vendored in the
 * reference tree) is restated from its published algorithm: "visit-order
 * parity unpinned" (DESIGN.md section 3).  The PAF / BEDPE writers
 * (oracle_query_paf: merge_adjusted_intervals and its CIGAR helpers) have no
 * reference test at all: that restatement is pinned by hand-checked rows only
 * (tests/test_oracle_kat.py::test_paf_and_bedpe_rows_by_hand).
 */
#ifndef IMPG_ORACLE_H
#define IMPG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index_t;

/* One AdjustedInterval (impg.rs:225) without its CIGAR. */
typedef struct {
  uint32_t query_id;
  int32_t q_first, q_last;
  uint32_t target_id;
  int32_t t_first, t_last;
} oracle_interval_t;

/* Transitive/query parameters (impg_index.rs:61-94; main.rs:4259-4285). */
typedef struct {
  int32_t transitive;      /* 0 = Impg::query, 1 = query_transitive_* */
  int32_t dfs;             /* 1 = query_transitive_dfs */
  uint32_t max_depth;      /* u16 in the reference, 0 = unlimited */
  int32_t min_transitive_len;
  int32_t min_distance_between_ranges;
  int32_t min_output_length; /* <0 = None */
  double min_identity;       /* NaN = None */
  int32_t store_cigar;
  int32_t multi_impg;        /* 1 = MultiImpg semantics (multi_impg.rs) */
  int32_t original_sequence_coordinates; /* text writers only: --original-sequence-coordinates (main.rs:4370) */
  int32_t consider_strandness;           /* BED writer only: --consider-strandness (main.rs:4380, :4395-4409) */
} oracle_params_t;

/* ---- leaf functions (known-answer tested) ------------------------------ */

/* parse_cigar_to_delta (impg.rs:2935-2950).  Returns number of ops, or -1 on an
 * invalid op letter (the reference panics, impg.rs:88). */
long oracle_parse_cigar(const char *cigar, size_t len, uint32_t *ops_out,
                        size_t cap);

/* invert_cigar_ops_in_place (impg.rs:144-156). */
void oracle_invert_cigar(uint32_t *ops, size_t n, int strand_reverse);

/* project_target_range_through_alignment (impg.rs:2760-2898).
 * out[0..3] = q_start,q_end,t_start,t_end ; slice written to slice_out
 * (adjusted), *slice_len.  Returns 1 = Some, 0 = None. */
int oracle_project(int32_t r0, int32_t r1, int32_t ts, int32_t te, int32_t qs,
                   int32_t qe, int strand_reverse, const uint32_t *ops,
                   size_t n_ops, int32_t *out4, uint32_t *slice_out,
                   size_t *slice_len);

/* calculate_gap_compressed_identity (impg.rs:2952-2973). */
double oracle_gap_compressed_identity(const uint32_t *ops, size_t n);

/* SortedRanges (impg.rs:242-369): opaque handle for unit tests. */
typedef struct oracle_sorted_ranges oracle_sorted_ranges_t;
oracle_sorted_ranges_t *oracle_sr_new(int32_t sequence_length,
                                      int32_t min_distance);
void oracle_sr_free(oracle_sorted_ranges_t *);
/* insert; pieces written as pairs into pieces_out (cap pairs). returns count. */
long oracle_sr_insert(oracle_sorted_ranges_t *, int32_t a, int32_t b,
                      int32_t *pieces_out, size_t cap);
long oracle_sr_get(oracle_sorted_ranges_t *, int32_t *out, size_t cap);

/* ---- index ------------------------------------------------------------- */

/* Build from PAF files (paf.rs:118-194, impg.rs:1535-1652).  CIGARs are read
 * back per hit with pread + ASCII parse (impg.rs:495-551) unless preparse != 0.
 * Sequence ids: first-seen order over files (query then target per line). */
oracle_index_t *oracle_index_from_paf(const char *const *paths, int n_paths,
                                      int bidirectional, int preparse);
/* Same, the "file" is a memory buffer (offsets behave like a file). */
/* One tracepoint alignment (OneAlnAlignment, onealn.rs:786-802, as AlignmentRecord sees it): its n_segs
 * tracepoints (target deltas) start at tracepoints[seg_off]; Standard mode brings query_deltas[] alongside,
 * FASTGA mode diffs[] and the query's start within its contig (first query delta, impg.rs:726-735). */
typedef struct {
  uint32_t query_id, target_id;
  int32_t query_start, query_end, target_start, target_end;
  uint64_t seg_off;
  uint32_t n_segs;
  uint32_t strand;
  int64_t query_contig_start;
} oracle_tp_record_t;
/* The reference's index file "IMPGIDX2" (impg.rs:1655-1721 / :1787-1850; bincode 2 standard encoding restated,
 * PARITY UNPINNED: no .impg file in the reference tree).  write: shuffle_seed != 0 permutes every tree's intervals
 * (a reader rebuilds the tree from whatever order the file has).  from_impg: the trees are rebuilt from the file's
 * order; CIGARs are read from the alignment files given, in the order the index was built with. */
int oracle_index_write_impg(const oracle_index_t *, const char *path, uint64_t shuffle_seed);
oracle_index_t *oracle_index_from_impg(const char *path, const char *const *alignment_files, int n_files, int preparse);
oracle_index_t *oracle_index_from_tracepoints(const oracle_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                              const int32_t *query_deltas, const int32_t *diffs, int fastga,
                                              int32_t trace_spacing, int32_t max_complexity, const int64_t *seq_len,
                                              uint32_t n_seq, int bidirectional);
oracle_index_t *oracle_index_from_paf_text(const char *text, size_t len,
                                           int bidirectional, int preparse);
void oracle_index_free(oracle_index_t *);
const char *oracle_last_error(void);

uint32_t oracle_num_seqs(const oracle_index_t *);
const char *oracle_seq_name(const oracle_index_t *, uint32_t id);
int64_t oracle_seq_len(const oracle_index_t *, uint32_t id);
int64_t oracle_seq_id(const oracle_index_t *, const char *name);
size_t oracle_num_records(const oracle_index_t *);
size_t oracle_num_targets(const oracle_index_t *);
/* number of index entries of one target and their visit-independent listing
 * in tree (sorted) order: first,last,query_id,flags(bit0 strand,bit1 reversed) */
size_t oracle_target_entries(const oracle_index_t *, uint32_t target_id,
                             int32_t *first_last_qid_flags, size_t cap);

/* ---- queries ----------------------------------------------------------- */

/* Impg::query / query_transitive_bfs / _dfs / MultiImpg flavour.  Results in
 * the reference's emission order, self interval(s) first.  Returns the number
 * of results (may exceed cap: call again), <0 on error.  If cigar_off/cigar_ops
 * are non-NULL and store_cigar, op slices are appended (cigar_off has n+1). */
long oracle_query(const oracle_index_t *, uint32_t target_id, int32_t start,
                  int32_t end, const oracle_params_t *p, oracle_interval_t *out,
                  size_t cap);

/* query_transitive_{bfs,dfs} with masked_regions = Some(map) (impg.rs:2062-2081,
 * :2316-2335; multi_impg.rs:801-830): the map as n_mask entries (sequence id,
 * SortedRanges.sequence_length, ranges[mask_off[i]..mask_off[i+1]) as start,end
 * pairs, sorted and disjoint); every SortedRanges has min_distance 0 as
 * partition.rs:250-256 builds them.  Returns -2 for a non-transitive params. */
long oracle_query_masked(const oracle_index_t *, uint32_t target_id, int32_t start, int32_t end,
                         const oracle_params_t *p, uint32_t n_mask, const uint32_t *mask_seq,
                         const int32_t *mask_seq_len, const uint64_t *mask_off,
                         const int32_t *mask_ranges, oracle_interval_t *out, size_t cap);

/* The same with an optional mask (has_mask) and an optional subset filter: subset_keep[id] != 0 where
 * SubsetFilter::matches(name of id) holds (subset_filter.rs:23-60; the string matching stays with the
 * caller).  A hit is kept iff its query id is the query's own target or subset_keep says so: during the
 * exploration for the transitive queries (impg.rs:2176-2185, :2430-2439; multi_impg.rs:888-896), after the
 * query otherwise (main.rs:11693-11696). */
long oracle_query_filtered(const oracle_index_t *, uint32_t target_id, int32_t start, int32_t end,
                           const oracle_params_t *p, int has_mask, uint32_t n_mask, const uint32_t *mask_seq,
                           const int32_t *mask_seq_len, const uint64_t *mask_off, const int32_t *mask_ranges,
                           const uint8_t *subset_keep, oracle_interval_t *out, size_t cap);

/* SubsetFilter: parse_subset_filter(list_text) then matches(names[i]) -> out[i]; returns entry_count
 * (subset_filter.rs:19-60, :117-176). */
long oracle_subset_matches(const char *list_text, const char *const *names, size_t n, uint8_t *out);

/* parse_subsequence_coordinates (main.rs:4642-4659): 1 and (base name, start offset) for "base:START-END",
 * 0 when the name carries no parsable coordinates, -1 if base_out is too small. */
long oracle_parse_subsequence(const char *seq_name, char *base_out, size_t cap, int32_t *offset);

/* Same with store_cigar: cigar_off[cap+1], cigar_ops[ops_cap] receive the
 * Vec<CigarOp> of every result (CSR); *n_ops = total ops (may exceed ops_cap). */
long oracle_query_cigar(const oracle_index_t *, uint32_t target_id, int32_t start,
                        int32_t end, const oracle_params_t *p, oracle_interval_t *out,
                        size_t cap, uint64_t *cigar_off, uint32_t *cigar_ops,
                        size_t ops_cap, uint64_t *n_ops);

/* Process-wide switch, off by default: overlapping entries of a target are visited in ascending start (ties in
 * input order) instead of the coitrees order.  This is NOT reference behaviour; it is the checker of the engine's
 * IMPG_ORDER_SORTED policy and lets tests measure what the visit order can change. */
void oracle_set_sorted_visits(int on);

/* number of Some(..) projections performed by the last oracle_query on this
 * thread (the work unit of BASELINE.md section 3). */
uint64_t oracle_last_projection_count(void);

/* merge_adjusted_intervals_gap_2d (main.rs:12858-13011) followed by
 * merge_query_adjusted_intervals (main.rs:12474-12560), in place, as
 * output_results_bed does (main.rs:11849-11892) with all CIGARs empty.
 * Returns the new count. */
long oracle_bed_merge(oracle_interval_t *iv, size_t n, int32_t merge_distance,
                      int merge_strands);
/* merge_query_adjusted_intervals alone (main.rs:12474-12560) */
long oracle_merge_query(oracle_interval_t *iv, size_t n, int32_t merge_distance,
                        int merge_strands);

/* perform_query + output_results_bed for one target range; appends BED text to
 * a malloc'ed buffer (*buf,*len,*cap grow).  Returns 0 or <0. */
int oracle_query_bed(const oracle_index_t *, const char *target_name,
                     int32_t start, int32_t end, const char *range_name,
                     const oracle_params_t *p, int32_t merge_distance,
                     char **buf, size_t *len, size_t *cap);

/* perform_query (store_cigar = true), results.remove(0), then output_results_paf
 * (format 0) or output_results_bedpe (format 1): merge_adjusted_intervals with
 * its CIGAR concatenation / f32-scaled trims, gi:f / bi:f formatting
 * (main.rs:7472-7496, :11894-12103, :12563-12845, :13014-13180). */
int oracle_query_paf(const oracle_index_t *, const char *target_name, int32_t start,
                     int32_t end, const char *range_name, const oracle_params_t *p,
                     int32_t merge_distance, int format, char **buf, size_t *len,
                     size_t *cap);

/* parse_bed_file / parse_target_range (partition.rs:1719-1789) */
long oracle_parse_bed_text(const char *text, size_t len, char *names_out,
                           size_t names_cap, int32_t *start_end_out,
                           char *rnames_out, size_t rnames_cap, size_t cap);
int oracle_parse_target_range(const char *s, char *name_out, size_t name_cap,
                              int32_t *start, int32_t *end);

/* ---- timed CPU baseline ------------------------- */
/* Runs n ranges.  mode 0 = reference structure: ranges serial, each BFS level's
 * frontier parallel over `threads` (impg.rs:2384-2465; main.rs:7435).
 * mode 1 = ranges parallel over `threads`.  Returns projected-range count and
 * wall seconds of the query phase. */
int oracle_bench(const oracle_index_t *, const uint32_t *target_ids,
                 const int32_t *starts, const int32_t *ends, size_t n,
                 const oracle_params_t *p, int threads, int mode,
                 uint64_t *n_projected, uint64_t *n_results, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
