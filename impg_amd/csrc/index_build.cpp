// Device index construction: replaces Impg::from_multi_alignment_records
// (src/impg.rs:1535-1652), the per-target BasicCOITree::new (impg.rs:1630) and
// ForestMap (src/forest_map.rs) with flat arrays in HBM:
//   * entries of one target contiguous and ascending in target start (segment
//     table tgt_off[] indexed by target id),
//   * starts/ends/pmax/rank SoA for the wavefront search,
//   * one op pool in 32-op (128-byte) tiles with a (target,query) prefix
//     checkpoint per tile, shared by the forward and the reversed entry.
#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>

#include "impg_internal.hpp"

namespace impg {

// IMPG_POISON=<hex byte> (debugging aid): every block a DevBuf obtains -- new or recycled -- is filled with that byte
// first, so that a kernel reading a buffer it has not written yet fails the same way every time instead of depending
// on what the block held before.  The parity soak is run under several patterns (scripts/fuzz_parity.py).
static int poison_byte() {
  static const int b = [] {
    const char *e = getenv("IMPG_POISON");
    return e && *e ? (int)(strtoul(e, nullptr, 16) & 0xFFu) : -1;
  }();
  return b;
}
void DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return;
  release();
  size_t want = std::max<size_t>(bytes, 256);
  if (pool) {
    p = pool->take(want, cap);
  } else {
    IMPG_HIP(hipMalloc(&p, want));
    cap = want;
  }
  if (poison_byte() >= 0) {
    IMPG_HIP(hipDeviceSynchronize());  // (a recycled block may still be read by work in flight on another stream)
    IMPG_HIP(hipMemset(p, poison_byte(), cap));
    IMPG_HIP(hipDeviceSynchronize());
  }
}
void DevBuf::release() {
  if (p) {
    if (pool) pool->give(p, cap);
    else (void)hipFree(p);
  }
  p = nullptr;
  cap = 0;
}

void *BufPool::take(size_t bytes, size_t &cap_out) {
  // best fit among the blocks that are not wastefully large for the request
  const size_t limit = std::max<size_t>(4 * bytes, 1u << 20);
  size_t best = free_.size();
  for (size_t i = 0; i < free_.size(); i++)
    if (free_[i].cap >= bytes && free_[i].cap <= limit && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
  if (best != free_.size()) {
    Blk b = free_[best];
    free_[best] = free_.back();
    free_.pop_back();
    held -= b.cap;
    cap_out = b.cap;
    return b.p;
  }
  // a quarter of slack (at most 1 GiB): the same level of the next chunk is about, not exactly, this size
  size_t want = (bytes + std::min<size_t>(bytes / 4, (size_t)1 << 30) + 255) & ~(size_t)255;
  void *p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    for (Blk &b : free_) (void)hipFree(b.p);  // out of memory: drop the free list and ask for the exact size
    free_.clear();
    held = 0;
    want = bytes;
    IMPG_HIP(hipMalloc(&p, want));
  }
  cap_out = want;
  return p;
}
void BufPool::give(void *p, size_t cap) {
  free_.push_back({p, cap});
  held += cap;
  while (held > max_held && !free_.empty()) {
    size_t big = 0;
    for (size_t i = 1; i < free_.size(); i++) if (free_[i].cap > free_[big].cap) big = i;
    (void)hipFree(free_[big].p);
    held -= free_[big].cap;
    free_[big] = free_.back();
    free_.pop_back();
  }
}
BufPool::~BufPool() {
  for (Blk &b : free_) (void)hipFree(b.p);
}

// Visit rank of the sorted positions [0,n) under the restated coitrees 0.4
// BasicCOITree::query order: pre-order over the implicit midpoint BST, except
// that a complete subtree of <= 8 nodes rooted at a "unit root" depth is visited
// in sorted order.  Unit-root depths follow the van Emde Boas split of
// veb_order_recursion: d0 = 0, d' = d + (D - d)/2 + 1 with D = floor(log2 n).
void coitrees_visit_rank(uint32_t n, uint32_t *rank) {
  if (n == 0) return;
  uint32_t D = 31 - __builtin_clz(n);
  struct Fr { uint32_t s, e, d, cd; };
  std::vector<Fr> st;
  st.push_back({0, n, 0, 0});
  uint32_t c = 0;
  while (!st.empty()) {
    Fr f = st.back();
    st.pop_back();
    if (f.s >= f.e) continue;
    uint32_t size = f.e - f.s;
    if (f.d == f.cd) {
      if (size <= 8) {
        for (uint32_t i = f.s; i < f.e; i++) rank[i] = c++;
        continue;
      }
      f.cd = f.d + (D - f.d) / 2 + 1;
    }
    uint32_t r = f.s + size / 2;
    rank[r] = c++;
    st.push_back({r + 1, f.e, f.d + 1, f.cd});
    st.push_back({f.s, r, f.d + 1, f.cd});
  }
}

namespace {

// a large host array whose elements are all written by the parallel builders: not worth a serial value-initialisation
// pass first (1.5 GB of fills for the headline index)
template <class T> struct RawVec {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  explicit RawVec(size_t n_) : p(n_ ? new T[n_] : nullptr), n(n_) {}
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T &operator[](size_t i) { return p[i]; }
};
template <class T> void upload(impg_gpu_index &ix, int k, const RawVec<T> &v, size_t &acc) {
  DevBuf &b = *ix.blob(k);
  b.reserve(std::max<size_t>(v.size() * sizeof(T) + 64, 256));
  if (!v.empty()) IMPG_HIP(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ix.blob_bytes[k] = v.size() * sizeof(T);
  acc += v.size() * sizeof(T);
}
template <class T> void upload(impg_gpu_index &ix, int k, const std::vector<T> &v, size_t &acc) {
  DevBuf &b = *ix.blob(k);
  b.reserve(std::max<size_t>(v.size() * sizeof(T) + 64, 256));  // + slack: kernels read whole 16-byte vectors
  if (!v.empty()) IMPG_HIP(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ix.blob_bytes[k] = v.size() * sizeof(T);
  acc += v.size() * sizeof(T);
}

void parallel_chunks(size_t n, const std::function<void(size_t, size_t)> &f) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n / 1024 + 1));
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++) th.emplace_back([&, t]() { f(n * t / T, n * (t + 1) / T); });
  for (auto &x : th) x.join();
}

}  // namespace

void build_index(impg_gpu_index &ix, const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops,
                 size_t n_ops, const int64_t *seq_len, uint32_t n_seq, bool bidirectional, int order_policy,
                 uint32_t shard, uint32_t n_shards, const uint32_t *owner, const TpInput *tp, const EntryPlan *plan) {
  const bool timing = getenv("IMPG_BUILD_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const double t = tnow();
    fprintf(stderr, "[build] %-28s %.3f s\n", what, t - t_prev);
    t_prev = t;
  };
  if (n_shards == 0 || shard >= n_shards) throw Error{IMPG_E_INVALID, "bad shard"};
  if (order_policy != IMPG_ORDER_COITREES && order_policy != IMPG_ORDER_SORTED)
    throw Error{IMPG_E_INVALID, "bad order policy"};
  if (n_shards > 1 && !owner) throw Error{IMPG_E_INVALID, "a sharded index needs its shard map"};
  // CIGAR indexes in record order are built on the device (index_build_device.hip); this host builder takes the
  // rest (tracepoint indexes, the entry order of a loaded .impg file), is the fallback when the device is short of
  // memory for the build, and the checker of the device build (IMPG_BUILD_HOST=1 forces it).
  if (!tp && !plan && !getenv("IMPG_BUILD_HOST") &&
      build_index_device(ix, records, n_records, cigar_ops, n_ops, seq_len, n_seq, bidirectional, order_policy, shard, n_shards, owner))
    return;
  if (n_records >= (1ull << 31)) throw Error{IMPG_E_UNSUPPORTED, "more than 2^31 records in one index"};
  ix.n_records = n_records;
  if (ix.seq.lens.empty()) ix.seq.lens.assign(seq_len, seq_len + n_seq);

  // ---- validate, decide which records this shard needs -----------------------
  if (n_shards > 1 && !owner) throw Error{IMPG_E_INVALID, "a sharded index needs its shard map"};
  auto owned = [&](uint32_t key) { return n_shards == 1 || owner[key] == shard; };
  std::vector<uint8_t> need(n_records, 0);
  std::vector<uint32_t> seg_count(n_seq + 1, 0);
  for (size_t i = 0; i < n_records; i++) {
    const auto &r = records[i];
    if (r.query_id >= n_seq || r.target_id >= n_seq) throw Error{IMPG_E_INVALID, "record sequence id out of range"};
    if (r.cigar_off + r.cigar_len > n_ops) throw Error{IMPG_E_INVALID, "record CIGAR outside the op pool"};
    if (r.cigar_len > OP_LEN_MASK) throw Error{IMPG_E_UNSUPPORTED, "CIGAR longer than 2^29 ops"};
    if (plan) continue;
    if (owned(r.target_id)) { need[i] = 1; seg_count[r.target_id]++; }
    if (bidirectional && r.query_id != r.target_id && owned(r.query_id)) {  // impg.rs:1584
      need[i] = 1;
      seg_count[r.query_id]++;
    }
  }
  if (plan) {
    if (n_shards != 1 || plan->per_target.size() != n_seq) throw Error{IMPG_E_INVALID, "bad entry plan"};
    for (uint32_t t = 0; t < n_seq; t++)
      for (uint64_t x : plan->per_target[t]) {
        const uint64_t rec = x >> 1;
        if (rec >= n_records) throw Error{IMPG_E_INVALID, "entry plan names an unknown record"};
        const auto &r = records[rec];
        if ((x & 1) ? r.query_id != t : r.target_id != t) throw Error{IMPG_E_INVALID, "entry plan puts an entry on the wrong target"};
        need[rec] = 1;
        seg_count[t]++;
      }
  }

  // ---- op pool: 128-byte tiles {T0,Q0, 8 x u16 inner sums | 26 ops} (impg_internal.hpp) ----
  std::vector<uint32_t> tile_base(n_records, 0);
  std::vector<uint32_t> rec_totT(n_records, 0), rec_totQ(n_records, 0);
  uint64_t n_tiles = 0;
  for (size_t i = 0; i < n_records; i++) {
    if (!need[i]) continue;
    tile_base[i] = (uint32_t)n_tiles;
    n_tiles += (records[i].cigar_len + TILE_OPS - 1) / TILE_OPS;
    if (n_tiles >= (1ull << 32) - 2) throw Error{IMPG_E_UNSUPPORTED, "op pool exceeds 2^32 tiles"};
  }
  if (tp) {  // tracepoint alignments: a record occupies n_segs + 1 prefix-sum boundaries of 16 bytes, no tiles
    n_tiles = 0;
    uint64_t nb = 0;
    for (size_t i = 0; i < n_records; i++) {
      if (!need[i]) continue;
      tile_base[i] = (uint32_t)nb;
      nb += (uint64_t)records[i].cigar_len + 1;
      if (nb >= (1ull << 32) - 2) throw Error{IMPG_E_UNSUPPORTED, "tracepoint pool exceeds 2^32 boundaries"};
    }
    n_tiles = (nb * 4 + TILE_WORDS - 1) / TILE_WORDS;  // (the pool is accounted in 128-byte lines like the op pool)
  }
  RawVec<uint32_t> pool(n_tiles * TILE_WORDS);  // (every line is filled by the builder that owns it)
  const bool with_pfx = !tp && !(getenv("IMPG_PREFIX_LINES") && atoi(getenv("IMPG_PREFIX_LINES")) == 0);
  // identity filter: with prefix lines an *identity line* per tile (per-op matched / mismatched sums, impg_internal.hpp),
  // without them the sums before each sub-tile, where the two short walks start counting
  // (with prefix lines they are built on demand, impg_gpu_index::ensure_identity_lines, unless IMPG_IDENTITY_LINES=1 asks for them now)
  const bool with_idl = with_pfx && getenv("IMPG_IDENTITY_LINES") && atoi(getenv("IMPG_IDENTITY_LINES")) == 1;
  RawVec<uint4> idp(tp ? 0 : (with_pfx ? (with_idl ? IDL_WORDS / 4 : 0) : TILE_SUBS) * n_tiles);
  RawVec<uint32_t> pfx(with_pfx ? n_tiles * TILE_WORDS : 0);  // prefix lines (impg_internal.hpp); optional: see index_build_device.hip
  if (tp && n_tiles) {  // the tail of the last 128-byte line behind the last boundary
    uint64_t nb_words = 0;
    for (size_t i = 0; i < n_records; i++) if (need[i]) nb_words += ((uint64_t)records[i].cigar_len + 1) * 4;
    for (uint64_t w = nb_words; w < n_tiles * TILE_WORDS; w++) pool[w] = OP_PAD;
  }
  std::atomic<bool> bad_op{false};
  std::atomic<bool> bad_tp{false};
  if (tp) parallel_chunks(n_records, [&](size_t lo, size_t hi) {
    // per boundary k of a record: {sum |tracepoint|, sum query delta, sum matches, sum mismatches} over segments < k
    // (scan_overlapping_tracepoints, impg.rs:737-803, turned into prefix sums: the scan becomes two searches)
    for (size_t i = lo; i < hi; i++) {
      if (!need[i]) continue;
      const impg_gpu_tp_record_t &r = tp->records[i];
      uint4 *out = reinterpret_cast<uint4 *>(pool.data()) + tile_base[i];
      uint64_t st = 0, sq = 0, sm = 0, sx = 0;
      int32_t fb = 0;
      if (tp->mode.fastga) {
        const int32_t ts = tp->mode.trace_spacing, qsc = (int32_t)r.query_contig_start;
        fb = ((qsc / ts) + 1) * ts - qsc;  // impg.rs:730
      }
      for (uint32_t k = 0; k <= r.n_segs; k++) {
        out[k] = make_uint4((uint32_t)st, (uint32_t)sq, (uint32_t)sm, (uint32_t)sx);
        if (k == r.n_segs) break;
        const int64_t t = tp->tracepoints[r.seg_off + k];
        const int64_t qd = tp->mode.fastga ? (k == 0 ? fb : tp->mode.trace_spacing) : tp->query_deltas[r.seg_off + k];
        if (t < 0 || qd < 0) { bad_tp = true; break; }
        int64_t nd;  // impg.rs:771-787
        if (tp->mode.fastga) nd = tp->diffs[r.seg_off + k];
        else nd = (qd == 0 || t == 0) ? std::max(qd, t) : (int64_t)tp->mode.max_complexity;
        if (nd < 0) { bad_tp = true; break; }
        st += (uint64_t)t; sq += (uint64_t)qd;
        sm += (uint64_t)std::max<int64_t>(std::min(qd, t) - nd, 0);  // impg.rs:800-802
        sx += (uint64_t)nd;
        if (st >= (1ull << 31) || sq >= (1ull << 31) || sm >= (1ull << 31) || sx >= (1ull << 31)) { bad_tp = true; break; }
      }
      rec_totT[i] = (uint32_t)st;
      rec_totQ[i] = (uint32_t)sq;
    }
  });
  if (bad_tp) throw Error{IMPG_E_UNSUPPORTED, "tracepoints, query deltas and diffs must be non-negative and sum below 2^31 per alignment"};
  if (!tp) parallel_chunks(n_records, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
      if (!need[i]) continue;
      const uint32_t *src = cigar_ops + records[i].cigar_off;
      const uint32_t n = records[i].cigar_len;
      uint32_t st = 0, sq = 0, sm = 0, sx = 0, sg = 0;
      for (uint32_t k0 = 0; k0 < n; k0 += TILE_OPS) {
        const size_t tile = (size_t)tile_base[i] + k0 / TILE_OPS;
        uint32_t *line = pool.data() + tile * TILE_WORDS;
        for (uint32_t w = 0; w < TILE_WORDS; w++) line[w] = OP_PAD;
        const uint32_t t0 = st, q0 = sq;
        uint32_t dt[TILE_SUBS + 1], dq[TILE_SUBS + 1];  // sums before sub-tile s (s = 4: after the tile)
        uint32_t sub = 0;
        const uint32_t cnt = std::min(n - k0, TILE_OPS);
        uint32_t scratch_line[TILE_WORDS];
        uint32_t *pl = with_pfx ? pfx.data() + tile * TILE_WORDS : scratch_line;
        bool pwide = false;
        uint32_t scratch_idl[TILE_WORDS];
        uint32_t *il = with_idl ? reinterpret_cast<uint32_t *>(idp.data()) + tile * IDL_WORDS : scratch_idl;
        const uint32_t m0 = sm, x0 = sx, g0 = sg;
        uint32_t gapmask = 0;
        auto boundary = [&]() {
          dt[sub] = st - t0; dq[sub] = sq - q0;
          if (sub < TILE_SUBS && !with_pfx) idp[TILE_SUBS * tile + sub] = make_uint4(sm, sx, sg, 0);
          sub++;
        };
        for (uint32_t u = 0; u < cnt; u++) {
          if (u == sub_first_op(sub)) boundary();
          if (st - t0 > 0xFFFFu || sq - q0 > 0xFFFFu) pwide = true;
          pl[PFX_E0 + u] = ((st - t0) & 0xFFFFu) | ((sq - q0) << 16);
          if (with_pfx) il[IDL_E0 + u] = ((sm - m0) & 0xFFFFu) | ((sx - x0) << 16);
          const uint32_t v = src[k0 + u], code = v >> 29, len = v & OP_LEN_MASK;
          if (code > 4) bad_op = true;  // CigarOp::new panics (impg.rs:88)
          line[6 + u] = v;
          if (code != 2) st += len;  // target_delta: all but 'I' (impg.rs:115-121)
          if (code != 3) sq += len;  // |query_delta|: all but 'D' (impg.rs:123-135)
          if (code == 0 || code == 4) sm += len;       // 'M' counted as match (impg.rs:2959)
          else if (code == 1) sx += len;
          else { sg += 1; gapmask |= 1u << u; }         // gap-compressed: one per 'I' / 'D' op
        }
        while (sub <= TILE_SUBS) boundary();  // sub-tiles without ops start (and end) at the tile's end
        if (with_pfx) {
          for (uint32_t u = cnt; u < IDL_ENTRIES; u++) il[IDL_E0 + u] = ((sm - m0) & 0xFFFFu) | ((sx - x0) << 16);
          il[0] = m0; il[1] = x0; il[2] = g0; il[3] = gapmask;
        }
        if (st - t0 > 0xFFFFu || sq - q0 > 0xFFFFu) pwide = true;
        for (uint32_t u = cnt; u < PFX_ENTRIES; u++) pl[PFX_E0 + u] = ((st - t0) & 0xFFFFu) | ((sq - q0) << 16);
        // (bit 31 is the flag: a sum that large -- no consistent record has one -- reads as wide, and the literal walk takes over)
        pl[0] = t0 | (pwide ? 1u << 31 : 0u); pl[1] = q0; pl[2] = pl[PFX_E0 + PFX_STEP]; pl[3] = pl[PFX_E0 + 2u * PFX_STEP];
        line[0] = t0; line[1] = q0;
        if (dt[TILE_SUBS] >= TILE_WIDE || dq[TILE_SUBS] >= TILE_WIDE) {
          line[2] = line[3] = line[4] = line[5] = 0xFFFFFFFFu;  // wide tile: no inner splits
        } else {
          line[2] = dt[1] | dt[2] << 16; line[3] = dt[3] | dt[4] << 16;
          line[4] = dq[1] | dq[2] << 16; line[5] = dq[3] | dq[4] << 16;
        }
      }
      rec_totT[i] = st;
      rec_totQ[i] = sq;
    }
  });
  if (bad_op) throw Error{IMPG_E_INVALID, "Invalid CIGAR operation"};

  lap("validate + tiles");
  // ---- entries, grouped by key in input order (impg.rs:1559-1623) --------------
  std::vector<uint32_t> tgt_off(n_seq + 1, 0);
  for (uint32_t s = 0; s < n_seq; s++) tgt_off[s + 1] = tgt_off[s] + seg_count[s];
  ix.h_tgt_off = tgt_off;
  size_t n_entries = tgt_off[n_seq];
  if (n_entries >= (1ull << 32) - 1) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 entries"};
  std::vector<Entry> ent(n_entries);
  std::vector<uint32_t> ent_rec(n_entries);  // record of each entry (for the checkpoints below)
  {
    std::vector<uint32_t> cur(tgt_off.begin(), tgt_off.end() - 1);
    if (plan) {
      for (uint32_t t = 0; t < n_seq; t++)
        for (uint64_t x : plan->per_target[t]) {
          const size_t i = (size_t)(x >> 1);
          const auto &r = records[i];
          const uint32_t fl = (r.cigar_len & OP_LEN_MASK) | (r.strand ? EF_STRAND : 0);
          Entry e{};
          if (!(x & 1)) {
            e.ts = r.target_start; e.te = r.target_end; e.qs = r.query_start; e.qe = r.query_end;
            e.query_id = r.query_id; e.tile_base = tile_base[i]; e.nops_flags = fl;
            e.totT = rec_totT[i]; e.totQ = rec_totQ[i];
          } else {
            e.ts = r.query_start; e.te = r.query_end; e.qs = r.target_start; e.qe = r.target_end;
            e.query_id = r.target_id; e.tile_base = tile_base[i]; e.nops_flags = fl | EF_REVERSED;
            e.totT = rec_totQ[i]; e.totQ = rec_totT[i];
          }
          ent_rec[cur[t]] = (uint32_t)i;
          ent[cur[t]++] = e;
        }
    }
    for (size_t i = 0; i < n_records && !plan; i++) {
      const auto &r = records[i];
      uint32_t fl = (r.cigar_len & OP_LEN_MASK) | (r.strand ? EF_STRAND : 0);
      if (owned(r.target_id)) {
        Entry e{};
        e.ts = r.target_start; e.te = r.target_end; e.qs = r.query_start; e.qe = r.query_end;
        e.query_id = r.query_id; e.tile_base = tile_base[i]; e.nops_flags = fl;
        e.totT = rec_totT[i]; e.totQ = rec_totQ[i];
        ent_rec[cur[r.target_id]] = (uint32_t)i;
        ent[cur[r.target_id]++] = e;
      }
      if (bidirectional && r.query_id != r.target_id && owned(r.query_id)) {
        Entry e{};
        e.ts = r.query_start; e.te = r.query_end; e.qs = r.target_start; e.qe = r.target_end;
        e.query_id = r.target_id; e.tile_base = tile_base[i]; e.nops_flags = fl | EF_REVERSED;
        e.totT = rec_totQ[i]; e.totQ = rec_totT[i];  // axes swapped (impg.rs:1585-1590)
        ent_rec[cur[r.query_id]] = (uint32_t)i;
        ent[cur[r.query_id]++] = e;
      }
    }
  }
  lap("entries");
  // per segment: stable sort by start (coitrees sorts by `first` only)
  {
    std::vector<uint32_t> perm(n_entries);
    for (size_t i = 0; i < n_entries; i++) perm[i] = (uint32_t)i;
    std::atomic<uint32_t> next{0};
    unsigned hw = std::thread::hardware_concurrency();
    size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_seq));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&]() {
        for (;;) {
          uint32_t s = next.fetch_add(1);
          if (s >= n_seq) break;
          std::stable_sort(perm.begin() + tgt_off[s], perm.begin() + tgt_off[s + 1],
                           [&](uint32_t x, uint32_t y) { return ent[x].ts < ent[y].ts; });
        }
      });
    for (auto &x : th) x.join();
    std::vector<Entry> e2(n_entries);
    std::vector<uint32_t> r2(n_entries);
    parallel_chunks(n_entries, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) { e2[i] = ent[perm[i]]; r2[i] = ent_rec[perm[i]]; }
    });
    ent.swap(e2);
    ent_rec.swap(r2);
  }
  lap("per-target sort + permute");
  // effective-order target checkpoints of every entry: prefix at the start of
  // effective tile k.  Forward walk: the tile's own T0 (or Q0 for a reversed
  // entry); back-to-front walk (reversed entry on the reverse strand): total
  // minus the prefix at the END of the mirrored tile.
  std::vector<uint32_t> ext_cp;
  if (!tp) {
    std::vector<uint64_t> ext_off(n_entries + 1, 0);
    for (size_t i = 0; i < n_entries; i++) {
      uint32_t n = ent[i].nops_flags & OP_LEN_MASK, m = (n + TILE_OPS - 1) / TILE_OPS;
      ext_off[i + 1] = ext_off[i] + (m > INLINE_TILES ? m + 1 : 0);
    }
    if (ext_off[n_entries] >= (1ull << 32)) throw Error{IMPG_E_UNSUPPORTED, "external checkpoint array exceeds 2^32"};
    ext_cp.assign(ext_off[n_entries], 0);
    parallel_chunks(n_entries, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        Entry &e = ent[i];
        const uint32_t n = e.nops_flags & OP_LEN_MASK, m = (n + TILE_OPS - 1) / TILE_OPS;
        const bool swp = (e.nops_flags & EF_REVERSED) != 0, flip = swp && (e.nops_flags & EF_STRAND);
        const uint32_t *base = pool.data() + (size_t)e.tile_base * TILE_WORDS;
        auto pre = [&](uint32_t k) -> uint32_t {  // effective target prefix at the start of effective tile k (k <= m)
          if (k == 0) return 0;
          if (k >= m) return e.totT;
          if (!flip) return base[(size_t)k * TILE_WORDS + (swp ? 1 : 0)];
          const uint32_t *line = base + (size_t)(m - k) * TILE_WORDS;  // effective tile k-1 is original tile m-k
          return e.totT - line[swp ? 1 : 0];
        };
        if (m <= INLINE_TILES) {
          // P[1..m-1], then P[m] = the record total, then INT_MAX: the kernel counts "P[i] < x" over all
          // seven slots without asking how many are real (P[8], when m = 8, is the entry's totT field)
          for (uint32_t k = 1; k <= 7; k++) e.tcp[k - 1] = k < m ? pre(k) : k == m ? e.totT : 0x7FFFFFFFu;
        } else {
          e.tcp[0] = (uint32_t)ext_off[i];
          for (uint32_t k = 0; k <= m; k++) ext_cp[ext_off[i] + k] = pre(k);
        }
      }
    });
  }
  lap("checkpoints");
  // SoA columns, running max of end, visit rank, search levels
  std::vector<int32_t> starts(n_entries), ends(n_entries), ends_t(n_entries), pmax(n_entries);
  std::vector<uint32_t> rank(n_entries);
  std::vector<SegDesc> seg(n_seq);
  std::vector<int32_t> starts_lvl, pmax_lvl;
  size_t n_targets = 0;
  for (uint32_t s = 0; s < n_seq; s++) {
    n_targets += seg_count[s] != 0;
    SegDesc d{};
    d.a = tgt_off[s];
    d.n = tgt_off[s + 1] - tgt_off[s];
    uint32_t cnt = d.n, lev = 0;
    uint64_t off = starts_lvl.size();
    while (cnt > 64) {
      if (lev == MAX_LEVELS) throw Error{IMPG_E_UNSUPPORTED, "more than 64^5 entries on one target"};
      cnt = (cnt + 63) / 64;
      d.off[lev] = (uint32_t)off;
      d.cnt[lev] = cnt;
      off += cnt;
      lev++;
    }
    d.nlev = lev;
    if (off >= (1ull << 32)) throw Error{IMPG_E_UNSUPPORTED, "search levels exceed 2^32"};
    starts_lvl.resize(off);
    pmax_lvl.resize(off);
    seg[s] = d;
  }
  {
    std::atomic<uint32_t> next{0};
    unsigned hw = std::thread::hardware_concurrency();
    size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_seq));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&]() {
        for (;;) {
          uint32_t s = next.fetch_add(1);
          if (s >= n_seq) break;
          const SegDesc &d = seg[s];
          uint32_t a = d.a, b = d.a + d.n;
          if (a == b) continue;
          int32_t m = INT32_MIN;
          for (uint32_t i = a; i < b; i++) {
            starts[i] = ent[i].ts;
            ends[i] = ent[i].te;
            ends_t[i] = ent[i].ts < ent[i].te ? ent[i].te : INT32_MIN;
            m = std::max(m, ent[i].te);
            pmax[i] = m;
          }
          if (order_policy == IMPG_ORDER_COITREES) coitrees_visit_rank(b - a, rank.data() + a);
          else for (uint32_t i = a; i < b; i++) rank[i] = i - a;
          // level k = last element of every 64-block of level k-1
          const int32_t *ps = starts.data() + a, *pp = pmax.data() + a;
          uint32_t pn = d.n;
          for (uint32_t k = 0; k < d.nlev; k++) {
            int32_t *ls = starts_lvl.data() + d.off[k], *lp = pmax_lvl.data() + d.off[k];
            for (uint32_t j = 0; j < d.cnt[k]; j++) {
              uint32_t last = std::min(pn, 64 * (j + 1)) - 1;
              ls[j] = ps[last];
              lp[j] = pp[last];
            }
            ps = ls; pp = lp; pn = d.cnt[k];
          }
        }
      });
    for (auto &x : th) x.join();
  }

  // ---- MultiImpg tie order (multi_impg.rs:556-592): the hits of a step are concatenated file by file, each
  // file's in ITS tree's visit order, then sorted stably by five keys -- so hits that agree on all five keep
  // (file, visit rank in that file's tree).  mrank = position of the entry in that order within its segment.
  std::vector<uint32_t> mrank;
  const bool multi_file = ix.file_first.size() > 2;
  if (multi_file) {
    mrank.resize(n_entries);
    auto file_of = [&](uint32_t rec) {
      return (uint32_t)(std::upper_bound(ix.file_first.begin(), ix.file_first.end(), (uint64_t)rec) - ix.file_first.begin() - 1);
    };
    std::atomic<uint32_t> next{0};
    unsigned hw = std::thread::hardware_concurrency();
    size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_seq));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&]() {
        std::vector<std::pair<uint64_t, uint32_t>> key;  // ((file, rank in the file's tree), entry)
        std::vector<uint32_t> members, vr;
        for (;;) {
          uint32_t s = next.fetch_add(1);
          if (s >= n_seq) break;
          const uint32_t a = tgt_off[s], b = tgt_off[s + 1];
          if (a == b) continue;
          // entries a..b are in start order (stable), which restricted to one file is that file's own node order
          std::vector<std::pair<uint32_t, uint32_t>> by_file;  // (file, entry), stable in entry order
          by_file.reserve(b - a);
          for (uint32_t i = a; i < b; i++) by_file.push_back({file_of(ent_rec[i]), i});
          std::stable_sort(by_file.begin(), by_file.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
          key.clear();
          for (size_t u = 0; u < by_file.size();) {
            size_t w = u;
            while (w < by_file.size() && by_file[w].first == by_file[u].first) w++;
            const uint32_t nf = (uint32_t)(w - u);
            vr.resize(nf);
            if (order_policy == IMPG_ORDER_COITREES) coitrees_visit_rank(nf, vr.data());
            else for (uint32_t k = 0; k < nf; k++) vr[k] = k;
            for (uint32_t k = 0; k < nf; k++) key.push_back({((uint64_t)by_file[u].first << 32) | vr[k], by_file[u + k].second});
            u = w;
          }
          std::sort(key.begin(), key.end());
          for (uint32_t k = 0; k < key.size(); k++) mrank[key[k].second] = k;
        }
      });
    for (auto &x : th) x.join();
  }

  lap("columns + ranks + levels");
  // ---- upload -------------------------------------------------------------------
  IMPG_HIP(hipSetDevice(ix.device));
  std::vector<int32_t> sl(n_seq);
  for (uint32_t s = 0; s < n_seq; s++) sl[s] = (int32_t)std::min<int64_t>(std::max<int64_t>(seq_len[s], 0), INT32_MAX);
  size_t acc = 0;
  upload(ix, 0, seg, acc);
  upload(ix, 1, starts, acc);
  upload(ix, 2, ends, acc);
  upload(ix, 3, ends_t, acc);
  upload(ix, 4, pmax, acc);
  upload(ix, 5, starts_lvl, acc);
  upload(ix, 6, pmax_lvl, acc);
  upload(ix, 7, rank, acc);
  if (multi_file) upload(ix, 8, mrank, acc);
  upload(ix, 9, ent, acc);
  upload(ix, 10, pool, acc);
  upload(ix, 11, ext_cp, acc);
  upload(ix, 12, idp, acc);
  upload(ix, 13, sl, acc);
  upload(ix, 14, pfx, acc);
  ix.device_bytes = acc;
  ix.n_entries = n_entries;
  ix.n_tiles = n_tiles;
  ix.n_targets = n_targets;
  lap("upload");
  ix.multi_file = multi_file;
  ix.tp_mode = tp != nullptr;
  ix.bind_view(n_seq, order_policy == IMPG_ORDER_SORTED);
}

}  // namespace impg

impg::DevBuf *impg_gpu_index::blob(int k) {
  impg::DevBuf *const b[N_BLOBS] = {&d_seg, &d_starts, &d_ends, &d_ends_t, &d_pmax, &d_starts_lvl, &d_pmax_lvl,
                                    &d_rank, &d_mrank, &d_entries, &d_ops, &d_ext_cp, &d_idp, &d_seq_len, &d_pfx};
  return b[k];
}
void impg_gpu_index::bind_view(uint32_t n_seq, uint32_t sorted_order) {
  using namespace impg;
  view.seg = d_seg.as<SegDesc>();
  view.starts = d_starts.as<int32_t>();
  view.ends = d_ends.as<int32_t>();
  view.ends_t = d_ends_t.as<int32_t>();
  view.pmax = d_pmax.as<int32_t>();
  view.starts_lvl = d_starts_lvl.as<int32_t>();
  view.pmax_lvl = d_pmax_lvl.as<int32_t>();
  view.rank = d_rank.as<uint32_t>();
  view.mrank = multi_file ? d_mrank.as<uint32_t>() : d_rank.as<uint32_t>();
  view.entries = d_entries.as<Entry>();
  view.ops = d_ops.as<uint32_t>();
  view.ext_cp = d_ext_cp.as<uint32_t>();
  view.idp = d_idp.as<uint4>();
  has_identity_lines.store(blob_bytes[12] != 0, std::memory_order_release);  // (built / loaded with them, or built on demand later)
  view.pfx = blob_bytes[14] ? d_pfx.as<uint32_t>() : nullptr;  // (an index may come without prefix lines)
  view.seq_len = d_seq_len.as<int32_t>();
  view.n_seq = n_seq;
  view.n_entries = (uint32_t)n_entries;
  view.sorted_order = sorted_order;
  view.tp_mode = tp_mode ? 1u : 0u;
  view.max_seg = 0;
  for (size_t t = 0; t + 1 < h_tgt_off.size(); t++) view.max_seg = std::max(view.max_seg, h_tgt_off[t + 1] - h_tgt_off[t]);
  if (h_tgt_off.empty()) view.max_seg = (uint32_t)n_entries;
}
