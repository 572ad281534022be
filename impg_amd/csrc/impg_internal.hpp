// Internal declarations shared by the host side of libimpg_gpu.so.
// Layouts here are the HBM data layout described in DESIGN.md section 4.
#pragma once
#include <atomic>
#include <chrono>
#include <exception>
#include <vector>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <new>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/impg_gpu.h"

namespace impg {

// ---- errors ---------------------------------------------------------------
void set_error(const std::string &msg);
// (when: the moment the error was raised.  A failure in one lane or rank of a sharded batch makes its peers fail too -- with
// "a peer rank failed", or with whatever the transport says when a partner has left a collective -- and the caller is told
// the FIRST one, the cause, not whichever thread's exception happens to be looked at first.)
inline uint64_t error_clock() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Error {
  int code;
  std::string msg;
  uint64_t when = error_clock();
};
// of several threads' exceptions: the earliest Error that is not a transport echo, else the earliest (anything that is
// not an Error -- bad_alloc ... -- comes first)
inline std::exception_ptr first_cause(const std::vector<std::exception_ptr> &errs) {
  std::exception_ptr best;
  uint64_t best_when = ~0ull;
  bool best_echo = true;
  for (const auto &e : errs) {
    if (!e) continue;
    try { std::rethrow_exception(e); }
    catch (const Error &er) {
      const bool echo = er.msg == "a peer rank failed" || er.msg.rfind("alltoallv: block sizes disagree", 0) == 0 ||
                        er.msg.rfind("the RCCL communicator was aborted", 0) == 0 || er.msg.rfind("host transport:", 0) == 0;
      if (!best || (best_echo && !echo) || (best_echo == echo && er.when < best_when)) { best = e; best_when = er.when; best_echo = echo; }
    }
    catch (...) { return e; }
  }
  return best;
}
#define IMPG_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw impg::Error{_e == hipErrorOutOfMemory ? IMPG_E_OOM : IMPG_E_HIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(_e)};                 \
  } while (0)

// ---- packed CIGAR ops (CigarOp, impg.rs:75-140) ----------------------------
constexpr uint32_t OP_LEN_MASK = (1u << 29) - 1;
constexpr uint32_t OP_PAD = 0xE0000000u;  // tile padding, never a valid op: code 7, length 0 (so its deltas are zero without a test)
constexpr uint32_t HIT_NONE = 0xFFFFFFFFu;

// One 128-byte line per tile: a 24-byte header followed by 26 packed ops (words 6..31).
//   w0 T0, w1 Q0          sums of target_delta / |query_delta| of the record's ops before the tile
//   w2 dT1 | dT2 << 16    the same sums, relative to T0, before the tile's ops 6, 14, 22
//   w3 dT3 | dT4 << 16    ... and after its last op (dT4 = the tile's own target sum)
//   w4 dQ1 | dQ2 << 16, w5 dQ3 | dQ4 << 16
// A tile is self-describing (running positions at either end and at three inner
// splits follow from its own line) and splits into four sub-tiles on 16-byte
// boundaries: ops 0..5 (words 6..11), 6..13 (12..19), 14..21 (20..27), 22..25
// (28..31); a scan walks one sub-tile, at most two 16-byte vectors.  A tile whose
// target or query sum does not fit 16 bits ("wide") stores 0xFFFF in all eight
// fields and is walked literally; its end position is the next tile's T0/Q0.
constexpr uint32_t TILE_WORDS = 32;
constexpr uint32_t TILE_OPS = 26;
constexpr uint32_t TILE_SUBS = 4;
constexpr uint32_t TILE_WIDE = 0xFFFFu;
#if defined(__HIPCC__)
#define IMPG_HD __host__ __device__
#else
#define IMPG_HD
#endif
IMPG_HD constexpr uint32_t sub_first_op(uint32_t s) { return s == 0 ? 0u : s == 1 ? 6u : s == 2 ? 14u : s == 3 ? 22u : 26u; }
IMPG_HD constexpr uint32_t subs_with_ops(uint32_t n_ops_in_tile) {  // sub-tiles of a tile that hold at least one op
  return n_ops_in_tile > 22u ? 4u : n_ops_in_tile > 14u ? 3u : n_ops_in_tile > 6u ? 2u : 1u;
}
constexpr uint32_t INLINE_TILES = 8;      // entries of records with <= 8 tiles carry their checkpoints inline

// Prefix line of a tile (array `pfx`, same tile index as the op pool): what the plain projection reads INSTEAD of
// the ops.  The running sums before every op of the tile, 16 bits per axis, relative to the tile's start:
//   w0 T0 | wide << 31    wide: a sum of the tile does not fit 16 bits -- no entries, the tile is walked literally
//   w1 Q0
//   w2, w3                copies of entries 9 and 18: with the header in one 16-byte read they split the tile in three
//   w4 + k, k = 0..27     e_k = dT_k | dQ_k << 16: target_delta / |query_delta| sums of the tile's ops before op k
//                         (k beyond the tile's last op: the tile's own sums)
// An op's deltas are the differences of neighbouring entries and its arm (impg.rs:2806-2868) follows from which of
// them is zero, so the first / last overlapping op is found by comparing entries with the range ends.
constexpr uint32_t PFX_E0 = 4;
constexpr uint32_t PFX_STEP = 9;
constexpr uint32_t PFX_ENTRIES = 28;
// Identity line of a tile (indexes WITH prefix lines; the identity filter's counterpart of the prefix line, round 3):
//   words 0..3   matched bases, mismatched bases, gap ops of the record before the tile; bit u of word 3: op u is 'I' / 'D'
//   word 4 + k   matched | mismatched << 16 bases of the tile's ops before op k (k = 0..27; past the last op: the tile's own sums)
// A tile whose prefix line is not `wide` has target / query sums below 2^16, and matched + mismatched bases are part
// of both, so the 16-bit fields hold.  Gap ops before op k of the tile: popcount of the mask below bit k.
constexpr uint32_t IDL_WORDS = 32;
constexpr uint32_t IDL_E0 = 4;
constexpr uint32_t IDL_ENTRIES = 28;

// ---- device index (HBM layout) ---------------------------------------------
// One 64-byte payload per index entry, stored in per-target start order.
struct alignas(16) Entry {
  int32_t ts, te, qs, qe;  // target_start/end, query_start/end of the ENTRY (already swapped for reversed entries)
  uint32_t query_id;
  uint32_t tile_base;      // first tile of the record's ops in the op pool
  uint32_t nops_flags;     // bits 0..28 n_ops, bit 30 = strand reverse, bit 31 = reversed entry (REVERSED_BIT)
  uint32_t totT;           // sum of target_delta over the record, in the entry's axes
  uint32_t totQ;           // sum of |query_delta|
  // target prefix at the start of effective tile k, k = 1..7 (effective = the
  // order in which THIS entry walks the record); tiles > 8: tcp[0] = offset of
  // the entry's m+1 prefixes in the external checkpoint array
  uint32_t tcp[7];
};
static_assert(sizeof(Entry) == 64, "entry payload is 64 bytes");
constexpr uint32_t EF_STRAND = 1u << 30;
constexpr uint32_t EF_REVERSED = 1u << 31;

// per-target segment + its search levels (64-ary, B+tree style): level k holds
// the last element of every 64-block of level k-1 (level 0 = the entry columns)
constexpr int MAX_LEVELS = 4;
struct SegDesc {
  uint32_t a, n;                 // segment [a, a+n) of the entry arrays
  uint32_t nlev;                 // number of sampled levels above the leaves
  uint32_t off[MAX_LEVELS];      // offset of level k (1-based: off[k-1]) in the level arrays
  uint32_t cnt[MAX_LEVELS];
  uint32_t pad;
};
static_assert(sizeof(SegDesc) == 48 && MAX_LEVELS == 4, "segment descriptor (kernels read it as three 16-byte vectors)");

struct DeviceIndexView {  // passed by value to kernels
  const SegDesc *seg;        // [n_seq] per-target segment table (replaces ForestMap)
  const int32_t *starts;     // [n_entries] t_start, ascending within a segment
  const int32_t *ends;       // [n_entries] t_end
  const int32_t *ends_t;     // [n_entries] t_end, INT_MIN where first >= last (such an entry never overlaps a clipped range)
  const int32_t *pmax;       // [n_entries] running max of t_end within the segment
  const int32_t *starts_lvl; // sampled levels of starts / pmax (same offsets)
  const int32_t *pmax_lvl;
  const uint32_t *rank;      // [n_entries] visit rank within the segment (order policy)
  const uint32_t *mrank;     // [n_entries] MultiImpg tie order: (alignment file, visit rank in THAT file's tree); == rank for one file
  const Entry *entries;      // [n_entries]
  const uint32_t *ops;       // [n_tiles*32] tiles
  const uint32_t *ext_cp;    // effective target prefixes of entries with > 8 tiles
  const uint4 *idp;          // identity filter: with prefix lines [8*n_tiles] one identity line per tile (IDL_*); without them
                             // [4*n_tiles] matched bases, mismatched bases, gap ops of the record before each sub-tile
  const uint32_t *pfx;       // [n_tiles*32] prefix lines (per-op running sums, 16 bits per axis)
  const int32_t *seq_len;    // [n_seq]
  uint32_t n_seq;
  uint32_t n_entries;
  uint32_t sorted_order;     // 1 = rank is the identity (IMPG_ORDER_SORTED)
  uint32_t max_seg;          // entries of the largest segment: visit ranks stay below it
  uint32_t tp_mode;          // 1 = tracepoint index (approximate mode): `ops` holds one uint4 of prefix sums per segment boundary
};

struct HostSeqIndex {  // SequenceIndex (seqidx.rs)
  std::vector<std::string> names;  // may be empty when created from raw records
  std::vector<int64_t> lens;
  std::unordered_map<std::string, uint32_t> name_to_id;
  uint32_t get_or_insert(const std::string &name, int64_t len);
};

// ---- recycled device allocations ------------------------------------------------
// The per-level visited tables are new buffers at every BFS level of every chunk; hipMalloc / hipFree cost
// 0.1-1 ms each and hipFree synchronises the device, so their blocks go back to a free list instead.
struct BufPool {
  struct Blk { void *p; size_t cap; };
  std::vector<Blk> free_;
  size_t held = 0;                                  // bytes on the free list
  static constexpr size_t MAX_HELD = 16ull << 30;   // beyond this the largest blocks are really freed
  size_t max_held = MAX_HELD;                       // (raised by callers whose blocks are larger: the rows a batch leaves in HBM)
  void *take(size_t bytes, size_t &cap_out);
  void give(void *p, size_t cap);
  ~BufPool();
};

// ---- simple growable device buffer ------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  BufPool *pool = nullptr;     // set: blocks come from / go back to the pool
  void reserve(size_t bytes);  // contents are NOT preserved on growth
  void release();
  // Between two buffers of ONE owner only -- an engine's scratch and its levels, say; the blocks take their way of being
  // freed with them.  A lane of a sharded index outlives the engine it leases: its buffers are marked lane_owned, and a
  // swap between one of them and an engine's is refused -- it once left a lane's receive buffer allocating from an
  // engine's pool after the engine had gone to another lane (wrong CIGARs one run in five).  That hand-over is adopt().
  bool lane_owned = false;
  void swap(DevBuf &o) {
    if (lane_owned != o.lane_owned) throw Error{IMPG_E_INVALID, "internal: DevBuf::swap between a lane's buffer and an engine's (use adopt)"};
    void *tp = p; p = o.p; o.p = tp;
    size_t tc = cap; cap = o.cap; o.cap = tc;
    BufPool *tq = pool; pool = o.pool; o.pool = tq;
  }
  // Take over o's block together with o's way of freeing it; o is left empty but keeps allocating as before.  (swap()
  // exchanges the pools too: right between two buffers of one owner, wrong when one of them outlives the other's pool
  // user -- a lane's receive buffer swapped with a level's pooled one kept using an engine's pool after the engine
  // had gone to another lane.)
  void adopt(DevBuf &o) {
    release();
    p = o.p; cap = o.cap; pool = o.pool;
    o.p = nullptr; o.cap = 0;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};

// ---- host arrays a result hands to the caller (host_mem.cpp) ---------------------------------
// Large result arrays are the destination of one device-to-host copy: they live in pinned blocks that are recycled
// through a process-wide pool (pinning 5 GB costs seconds, a recycled block nothing); small ones are plain malloc.
void *pinned_take(size_t bytes, size_t &cap_out);
void pinned_give(void *p, size_t cap);
size_t pinned_trim(size_t keep_bytes);  // frees pooled blocks until at most keep_bytes are held; returns the bytes freed
constexpr size_t PINNED_MIN_BYTES = 1u << 20;
template <class T> struct HostArr {
  T *p = nullptr;
  size_t n = 0, cap = 0;   // elements in use / bytes held
  bool pinned = false;
  HostArr() = default;
  HostArr(const HostArr &) = delete;
  HostArr &operator=(const HostArr &) = delete;
  ~HostArr() { release(); }
  void release() {
    if (p) { if (pinned) pinned_give(p, cap); else free(p); }
    p = nullptr; n = 0; cap = 0; pinned = false;
  }
  // room for `count` elements; contents are kept.  want_pinned: a DMA target (honoured from PINNED_MIN_BYTES on)
  void reserve(size_t count, bool want_pinned = false) {
    const size_t bytes = count * sizeof(T);
    if (bytes <= cap) return;
    T *np;
    size_t ncap;
    const bool pin = want_pinned && bytes >= PINNED_MIN_BYTES;
    if (pin) np = static_cast<T *>(pinned_take(bytes, ncap));
    else {
      ncap = std::max<size_t>(bytes, 64);
      np = static_cast<T *>(malloc(ncap));
      if (!np) throw std::bad_alloc();
    }
    if (n) memcpy(np, p, n * sizeof(T));
    const size_t keep = n;
    release();
    p = np; cap = ncap; pinned = pin; n = keep;
  }
  void resize(size_t count, bool want_pinned = false) { reserve(count, want_pinned); n = count; }  // new elements are NOT initialised
  void push_back(const T &v) {
    if ((n + 1) * sizeof(T) > cap) reserve(std::max<size_t>(2 * n, 16));
    p[n++] = v;
  }
  void assign(const T *b, const T *e) { n = 0; resize((size_t)(e - b)); if (e > b) memcpy(p, b, (size_t)(e - b) * sizeof(T)); }
  void assign(size_t count, const T &v) { n = 0; resize(count); for (size_t i = 0; i < count; i++) p[i] = v; }
  void clear() { n = 0; }
  void swap(HostArr &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(pinned, o.pinned); }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T *data() { return p; }
  const T *data() const { return p; }
  T *begin() { return p; }
  T *end() { return p + n; }
  const T *begin() const { return p; }
  const T *end() const { return p + n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  T &back() { return p[n - 1]; }
};
// dst[0, bytes) = src, on several threads when it is worth it (concatenating the parts of a result)
void parallel_memcpy(void *dst, const void *src, size_t bytes);

// ---- host ingest -------------------------------------------------------------
struct ParsedPaf {
  HostSeqIndex seq;
  std::vector<impg_gpu_record_t> records;
  std::vector<uint32_t> ops;
  std::vector<uint64_t> file_first;  // first record of every input file, then the record count
  // Raw mode (parse_paf_files(..., raw = true)): the CIGAR text is NOT tokenised on the host; records[i].cigar_off /
  // cigar_len are a byte offset / byte count in the concatenation of `texts`, and the device tokenises
  // (tokenize_on_device, index_build_device.hip), after which they are op offsets / counts like everywhere else.
  bool raw = false;
  struct Text { const char *p; size_t n; uint64_t base; };  // base: offset of the span in the concatenation
  std::vector<Text> texts;
  std::vector<std::shared_ptr<void>> holders;  // keep the spans alive (file mappings, decompressed buffers)
};
// parse_paf + parse_cigar_to_delta over whole files (paf.rs:118-194, impg.rs:2935-2950)
void parse_paf_files(const std::vector<std::string> &paths, ParsedPaf &out, bool raw = false);
void parse_paf_text(const char *text, size_t len, ParsedPaf &out, bool raw = false, uint64_t text_base = 0);
// raw mode: CIGAR text -> packed ops on the device (same tokens and the same errors as parse_cigar); fills d_ops and turns
// the records' byte offsets into op offsets.  Returns the number of ops.
uint64_t tokenize_on_device(ParsedPaf &pp, int device, DevBuf &d_ops);
long parse_cigar(const char *s, size_t n, uint32_t *out, size_t cap);

// tracepoint alignments (approximate mode): what build_index reads instead of the op pool
struct TpInput {
  const impg_gpu_tp_record_t *records;
  const int32_t *tracepoints, *query_deltas, *diffs;
  size_t n_segs_total;
  impg_gpu_tp_mode_t mode;
};

// Entries in a given input order instead of the order the records imply (a loaded IMPGIDX2 file lists every target's
// intervals in the order its writer's tree held them; ties among equal starts follow that order when the tree is
// rebuilt, impg.rs:1745-1755): per target, (record << 1 | reversed entry) in input order.
struct EntryPlan {
  std::vector<std::vector<uint64_t>> per_target;
};

// index_build_device.hip: the same index built by kernels from the packed ops; false = not taken (short of device memory)
bool build_index_device(impg_gpu_index &ix, const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                        const int64_t *seq_len, uint32_t n_seq, bool bidirectional, int order_policy, uint32_t shard, uint32_t n_shards,
                        const uint32_t *owner, const uint32_t *d_cigar_ops = nullptr /* the op pool already on the device */);
// visit rank of each sorted position of an n-entry segment (order policy)
void coitrees_visit_rank(uint32_t n, uint32_t *rank_out);

// ---- saved device index (index_io.cpp) ------------------------------------------
struct ShardInfo {  // what a saved part of a sharded index says about itself
  uint32_t world = 0, rank = 0;
  std::vector<uint32_t> owner;  // target sequence -> rank
  bool front = false;           // the file of the handle that fronts the shards of one process (no arrays)
};
void save_index(const impg_gpu_index &ix, const char *path, const ShardInfo *shard = nullptr, bool front = false);
void load_index(impg_gpu_index &ix, const char *path, ShardInfo *shard = nullptr);  // ix.device set; fills everything but the engine
// sharded.cpp: impg_gpu_index_save of a rank's shard (one file) / of a multi handle (path + path.shard<k>of<n>)
void save_sharded(const impg_gpu_index &ix, const char *path);

// ---- subset lists (subset.cpp): keep[i] = the list selects names[i]; returns the number of list entries
size_t subset_select(const char *text, size_t len, const char *const *names, size_t n, uint8_t *keep);

// ---- "base:START-END" sequence names (--original-sequence-coordinates; main.rs:4642-4678) ---------------------
// length of the base name and START if `name` ends in ":<i32>-...", else false
inline bool subsequence_origin(const std::string &name, size_t &base_len, uint32_t &offset) {
  const size_t colon = name.rfind(':');
  if (colon == std::string::npos) return false;
  const size_t dash = name.find('-', colon + 1);
  if (dash == std::string::npos) return false;
  size_t i = colon + 1;
  if (i < dash && name[i] == '+') i++;  // (a '-' cannot lead: the first dash ends the number)
  if (i >= dash) return false;
  uint64_t v = 0;
  for (; i < dash; i++) {
    if (name[i] < '0' || name[i] > '9') return false;
    v = v * 10 + (uint64_t)(name[i] - '0');
    if (v > 2147483647ull) return false;
  }
  base_len = colon;
  offset = (uint32_t)v;
  return true;
}
// appends the printed name and returns the offset to add to its coordinates
inline uint32_t put_original_name(std::string &s, const std::string &name, bool original) {
  size_t base_len = 0;
  uint32_t off = 0;
  if (original && subsequence_origin(name, base_len, off)) {
    s.append(name, 0, base_len);
    return off;
  }
  s += name;
  return 0;
}

// ---- BED (bed.cpp) -----------------------------------------------------------
size_t bed_merge(impg_gpu_interval_t *iv, size_t n, int32_t merge_distance, bool merge_strands);

struct Engine;    // engine.hpp
struct ShardCtx;  // sharded.cpp: this rank's part of an index sharded over GPUs
struct Cluster;   // sharded.cpp: the ranks of an index sharded over the GPUs of this process

}  // namespace impg

// ---- the opaque handles --------------------------------------------------------
struct impg_gpu_index {
  int device = 0;
  hipStream_t stream = nullptr;
  impg::HostSeqIndex seq;
  size_t n_records = 0, n_entries = 0, n_tiles = 0, n_targets = 0;
  std::vector<uint32_t> h_tgt_off;
  std::vector<uint64_t> file_first;  // first record of every alignment file (+ the record count); empty = one file
  impg::DevBuf d_seg, d_starts, d_ends, d_ends_t, d_pmax, d_starts_lvl, d_pmax_lvl, d_rank, d_mrank, d_entries, d_ops, d_ext_cp, d_idp, d_seq_len, d_pfx;
  impg::DeviceIndexView view{};
  // the device arrays in a fixed order with their content sizes (index_io.cpp writes / reads them back)
  static constexpr int N_BLOBS = 15;
  impg::DevBuf *blob(int k);
  size_t blob_bytes[N_BLOBS] = {};
  bool multi_file = false;
  bool tp_mode = false;  // built from tracepoints: every projection is the approximate one
  void bind_view(uint32_t n_seq, uint32_t sorted_order);  // view pointers from the device arrays
  std::atomic<size_t> device_bytes{0};
  // The identity lines (idp[] beside prefix lines: a third of the index, read by the identity filter only) are built on
  // the device the first time a query asks for min_gap_compressed_identity -- from the op lines, byte for byte what the
  // builders write when IMPG_IDENTITY_LINES=1 -- so that an index nobody filters costs 2.2 KB a record, not 3.2.
  std::mutex idl_m;
  void ensure_identity_lines();
  // (the fast path of every filtered query reads this without the lock: an acquire load of a flag that is stored, with
  // release, after view.idp -- a handle serves up to max_engines callers at once)
  std::atomic<bool> has_identity_lines{false};
  bool lacks_identity_lines() const {
    return !tp_mode && n_tiles && blob_bytes[14] && !has_identity_lines.load(std::memory_order_acquire);
  }
  // engines (engine.cpp: stream + scratch + the visited sets of one batch in flight), handed out by EngineLease
  std::vector<std::unique_ptr<impg::Engine>> engines;
  std::vector<impg::Engine *> eng_free;
  std::mutex eng_m;
  std::condition_variable eng_cv;
  int max_engines = 4;
  uint64_t opt_pair_budget = 1ull << 28;  // impg_gpu_set_option values, applied to an engine when it is leased
  uint32_t opt_chunk_ranges = 0, opt_locality_min = 4096;
  // failure injection for the tests of the multi-rank failure agreement (0 = off): (rank + 1) << 16 | hop of the batch
  // (1-based, counted per lane) at which that rank throws on the owner side / on the home side of the hop
  uint32_t opt_debug_fail_owner = 0, opt_debug_fail_home = 0;
  uint64_t opt_lane_schedule = 0;  // IMPG_LANE_SCHEDULE / option "lane_schedule": see run_lanes (sharded.cpp); 0 = off
  uint64_t opt_device_rows_pool = 160ull << 30;  // option "device_rows_pool_bytes"
  bool opt_free_slots = true;
  bool opt_regroup = true;
  bool opt_fuse_final = true;
  int opt_filter_covered = 0;
  int opt_walk = 1;
  uint32_t opt_walk_members = 0;  // option "walk_members" (Engine::walk_members)
  bool opt_seg_group = true;      // option "segment_groups" (Engine::seg_group)
  uint32_t opt_seg_parts = 0;     // option "segment_parts" (Engine::seg_parts_force)
  mutable std::atomic<uint64_t> walk_launches{0}, walk_fallbacks{0}, walk_last_members{1};  // impg_gpu_get_counter
  // ... and how the visited updates of its batches grouped their hits: levels cut into slices, levels counted a second time
  // for one huge query, levels that went to the library sort ([0], [1], [2]; Engine::update)
  mutable std::atomic<uint64_t> seg_stats[3] = {};
  impg::ShardCtx *shard = nullptr;    // set: this index is one rank's shard; queries are collective calls
  impg::Cluster *cluster = nullptr;   // set: this handle fronts n_dev shards in this process (no arrays of its own)
  impg_gpu_index();
  ~impg_gpu_index();
};

struct impg_gpu_results {
  std::vector<uint64_t> offsets;
  impg::HostArr<impg_gpu_interval_t> intervals;
  std::vector<impg_gpu_range_t> ranges;
  bool has_cigar = false;
  impg::HostArr<uint64_t> cigar_off;  // [intervals+1] when has_cigar
  impg::HostArr<uint32_t> cigar_ops;
  uint64_t projected = 0;
  double run_s = 0, assemble_s = 0;  // wall time in the engine (GPU) and in the host-side result assembly
};
