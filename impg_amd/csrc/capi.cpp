// extern "C" entry points of libimpg_gpu.so (include/impg_gpu.h).  Nothing
// unwinds across this file: every body is wrapped and mapped to a status code.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <functional>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <new>

#include "engine.hpp"

namespace impg {
thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
void render_paf(const impg_gpu_results &res, const impg_gpu_index &ix, const char *const *range_names,
                const impg_gpu_params_t &p, int32_t merge_distance, bool bedpe, std::vector<std::string> &parts);
void render_bed(const impg_gpu_results &res, const impg_gpu_index &ix, const char *const *range_names,
                const impg_gpu_params_t &p, int32_t merge_distance, std::vector<std::string> &parts);
char *join_parts(std::vector<std::string> &parts, size_t *len);  // bed.cpp
}  // namespace impg

using namespace impg;


namespace impg {
// One engine = one stream + one set of scratch buffers + the visited sets of the batch in flight.  Calls on one
// handle from several host threads (the trait is Send + Sync: rayon workers share the index) each take their
// own engine: up to max_engines run side by side, further callers wait for one to come back.
EngineLease::EngineLease(impg_gpu_index &ix_) : ix(ix_) {
  std::unique_lock<std::mutex> lk(ix.eng_m);
  for (;;) {
    if (!ix.eng_free.empty()) { e = ix.eng_free.back(); ix.eng_free.pop_back(); break; }
    if ((int)ix.engines.size() < ix.max_engines) {
      ix.engines.emplace_back(new Engine(ix.device));
      e = ix.engines.back().get();
      break;
    }
    ix.eng_cv.wait(lk);
  }
  e->pair_budget = ix.opt_pair_budget;
  e->chunk_ranges = ix.opt_chunk_ranges;
  e->locality_min = ix.opt_locality_min;
  e->free_slots_allowed = ix.opt_free_slots;
  e->regroup_pairs = ix.opt_regroup;
  e->fuse_allowed = ix.opt_fuse_final;
  e->filter_covered = ix.opt_filter_covered;
  e->walk_allowed = ix.opt_walk != 0 && !getenv("IMPG_NO_WALK");
  e->walk_bfs = ix.opt_walk == 2;  // (the environment switch runs a whole test suite on the batch engine)
  e->walk_members = ix.opt_walk_members;
  e->seg_group = ix.opt_seg_group;
  e->seg_parts_force = ix.opt_seg_parts;
  e->seg_stats = ix.seg_stats;
}
EngineLease::~EngineLease() {
  e->remote = nullptr;
  e->masked = false;
  e->subset_on = false;
  e->on_kernels_done = nullptr;  // (the row stream's hook captures its caller's locals: never past the lease)
  e->keep_any_order = false;
  e->ordered_rows = false;
  std::lock_guard<std::mutex> lk(ix.eng_m);
  ix.eng_free.push_back(e);
  ix.eng_cv.notify_one();
}
}  // namespace impg

#define IMPG_TRY try {
#define IMPG_CATCH                                  \
  }                                                 \
  catch (const impg::Error &e) {                    \
    impg::set_error(e.msg);                         \
    return e.code;                                  \
  }                                                 \
  catch (const std::bad_alloc &) {                  \
    impg::set_error("host out of memory");          \
    return IMPG_E_OOM;                              \
  }                                                 \
  catch (const std::exception &e) {                 \
    impg::set_error(std::string("internal: ") + e.what()); \
    return IMPG_E_INVALID;                          \
  }

namespace impg {

void require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error{IMPG_E_HIP, "no HIP device available (libimpg_gpu has no CPU fallback)"};
  if (device < 0 || device >= n) throw Error{IMPG_E_INVALID, "device ordinal out of range"};
}

std::unique_ptr<impg_gpu_index> make_index(const impg_gpu_record_t *records, size_t n_records, const uint32_t *ops,
                                           size_t n_ops, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                           int order_policy, int device, uint32_t shard, uint32_t n_shards,
                                           const HostSeqIndex *seq, const std::vector<uint64_t> *file_first,
                                           const uint32_t *owner) {
  if ((!records && n_records) || (!ops && n_ops) || (!seq_len && n_seq)) throw Error{IMPG_E_INVALID, "null input array"};
  require_device(device);
  auto ix = std::make_unique<impg_gpu_index>();
  ix->device = device;
  if (seq) ix->seq = *seq;
  if (file_first) {
    if (file_first->size() < 2 || file_first->front() != 0 || file_first->back() != n_records ||
        !std::is_sorted(file_first->begin(), file_first->end()))
      throw Error{IMPG_E_INVALID, "file boundaries must start at 0, end at n_records and be ascending"};
    ix->file_first = *file_first;
  }
  build_index(*ix, records, n_records, ops, n_ops, seq_len, n_seq, bidirectional != 0, order_policy, shard, n_shards, owner);
  { EngineLease warm(*ix); }  // the first engine exists before the first query (stream + counters)
  return ix;
}

}  // namespace impg

namespace impg {
// What the trait returns for one chunk: rows placed by the device (rows_device.hip), one copy across PCIe into a
// pinned block of the result, offsets widened on the host.  `levels` are consumed.
// max_rows: a chunk of several ranges with more rows than this is not assembled -- SplitBatch, the caller halves it
// (the row stream's bound on its host blocks); kernels_done: recorded on the stream once the chunk's last kernel is
// enqueued, ahead of the copies (whoever takes turns on the GPU may let the next one in while the rows cross PCIe)
void assemble_results(Engine &E, const impg_gpu_range_t *h_ranges, uint32_t n, const impg_gpu_params_t &p,
                      std::vector<std::unique_ptr<LevelBufs>> &levels, DevBuf &self_dev, impg_gpu_results &res, uint64_t max_rows,
                      hipEvent_t kernels_done) {
  hipStream_t s = E.stream;
  res.ranges.assign(h_ranges, h_ranges + n);
  res.has_cigar = p.store_cigar != 0;
  RowPlan pl;
  plan_rows(E, n, p, levels, self_dev, false, pl);
  const size_t nr = pl.n_rows;
  if (nr > max_rows && n > 1) throw SplitBatch{};
  DevBuf rows, clen, coff, cpool;
  for (DevBuf *b : {&rows, &clen, &coff, &cpool}) b->pool = &E.level_pool;
  rows.reserve(std::max<size_t>(nr * sizeof(impg_gpu_interval_t), 256));
  if (res.has_cigar) clen.reserve(std::max<size_t>(nr * 4, 256));
  scatter_rows(E, levels, pl, RowSinks{rows.as<impg_gpu_interval_t>(), nullptr, nullptr, nullptr, nullptr,
                                       res.has_cigar ? clen.as<uint32_t>() : nullptr});
  // (the row stream reuses `res` chunk after chunk: a chunk a little larger than every one before it must not pin a new
  // block -- seconds for 5 GB --, so its blocks grow with a quarter to spare, within the stream's bound)
  if (max_rows != ~0ull && nr * sizeof(impg_gpu_interval_t) > res.intervals.cap)
    res.intervals.reserve(std::max<size_t>(nr, (size_t)std::min<uint64_t>(max_rows, nr + nr / 4)), true);
  res.intervals.resize(nr, true);
  std::vector<uint32_t> off32((size_t)n + 1);
  IMPG_HIP(hipMemcpyAsync(off32.data(), pl.offsets.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, s));
  // (without CIGARs the rows' placement was the chunk's last kernel: the GPU turn ends here, the copy below overlaps the
  // other engine's kernels)
  if (kernels_done && !res.has_cigar) IMPG_HIP(hipEventRecord(kernels_done, s));
  if (nr) IMPG_HIP(hipMemcpyAsync(res.intervals.data(), rows.p, nr * sizeof(impg_gpu_interval_t), hipMemcpyDeviceToHost, s));
  if (res.has_cigar) {
    const uint64_t n_ops = build_row_cigars(E, levels, pl, clen, coff, cpool);
    res.cigar_off.resize(nr + 1, true);
    res.cigar_ops.resize(n_ops, true);
    if (kernels_done) IMPG_HIP(hipEventRecord(kernels_done, s));
    IMPG_HIP(hipMemcpyAsync(res.cigar_off.data(), coff.p, (nr + 1) * 8, hipMemcpyDeviceToHost, s));
    if (n_ops) IMPG_HIP(hipMemcpyAsync(res.cigar_ops.data(), cpool.p, n_ops * 4, hipMemcpyDeviceToHost, s));
  }
  if (kernels_done) { IMPG_HIP(hipEventSynchronize(kernels_done)); if (E.on_kernels_done) E.on_kernels_done(); }
  IMPG_HIP(hipStreamSynchronize(s));
  levels.clear();
  res.offsets.resize((size_t)n + 1);
  for (size_t q = 0; q <= n; q++) res.offsets[q] = off32[q];
  res.projected = E.last_projected;
}

}  // namespace impg

namespace {
// Runs fn(begin, end) over [0,n) in chunks; a chunk whose level outgrows the pair
// budget (SplitBatch) is retried at half the size.  Queries are independent, so
// any split by ranges gives identical results.
template <class F> void for_chunks(Engine &E, size_t n, F fn) {
  size_t chunk = E.chunk_ranges ? E.chunk_ranges : n;
  if (chunk == 0) chunk = 1;
  size_t b = 0;
  while (b < n) {
    size_t e = std::min(n, b + chunk);
    try {
      fn(b, e);
      b = e;
    } catch (const SplitBatch &) {
      if (e - b <= 1) throw Error{IMPG_E_UNSUPPORTED, "a single range exceeds the pair budget"};
      chunk = std::max<size_t>(1, (e - b) / 2);
    }
  }
}

}  // namespace

namespace {
// The trait's rows from the per-query walk (Engine::run_walk).  A handful of queries write into generous fixed
// regions of one pool and are done in one launch; a batch (or a query that outgrows its region) is counted first and
// walked again with every query's rows at their final place -- the walk is deterministic, so the second pass writes
// exactly what the first counted.
bool walk_query(impg_gpu_index &ix, Engine &E, const impg_gpu_range_t *h_ranges, uint32_t n, const impg_gpu_params_t &p, impg_gpu_results &res) {
  hipStream_t s = E.stream;
  Engine::WalkRows W;
  for (DevBuf *b : {&W.rows, &W.base, &W.cap, &W.n_rows}) b->pool = &E.level_pool;
  W.base.reserve(std::max<size_t>((size_t)n * 8, 256)); W.cap.reserve(std::max<size_t>((size_t)n * 4, 256));
  W.n_rows.reserve(std::max<size_t>((size_t)n * 4, 256));
  std::vector<unsigned long long> base(n);
  std::vector<uint32_t> cap(n), cnt(n);
  const bool optimistic = n <= Engine::SMALL_RANGES;
  const uint32_t each = optimistic ? (1u << 17) : 0u;
  for (uint32_t q = 0; q < n; q++) { base[q] = (unsigned long long)q * each; cap[q] = each; }
  auto pass = [&](uint64_t total_rows) {
    W.rows.reserve(std::max<size_t>(total_rows * sizeof(impg_gpu_interval_t), 256));
    IMPG_HIP(hipMemcpyAsync(W.base.p, base.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    IMPG_HIP(hipMemcpyAsync(W.cap.p, cap.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    return E.run_walk(ix, E.ranges_dev.as<impg_gpu_range_t>(), n, p, nullptr, nullptr, nullptr, &W, cnt.data());  // (cnt: the rows every query wrote)
  };
  if (!pass((uint64_t)n * each)) return false;
  bool fits = true;
  uint64_t total = 0;
  for (uint32_t q = 0; q < n; q++) { fits = fits && cnt[q] <= cap[q]; total += cnt[q]; }
  bool contiguous = false;
  if (!fits) {
    uint64_t at = 0;
    for (uint32_t q = 0; q < n; q++) { base[q] = at; cap[q] = cnt[q]; at += cnt[q]; }
    const std::vector<uint32_t> want = cnt;
    if (!pass(total)) return false;
    if (cnt != want) throw Error{IMPG_E_INVALID, "internal: the walk's second pass disagrees with its first"};
    contiguous = true;
  }
  res.offsets.assign((size_t)n + 1, 0);
  for (uint32_t q = 0; q < n; q++) res.offsets[q + 1] = res.offsets[q] + cnt[q];
  res.intervals.resize(total, true);
  const impg_gpu_interval_t *d_rows = W.rows.as<impg_gpu_interval_t>();
  if (contiguous) {
    if (total) IMPG_HIP(hipMemcpyAsync(res.intervals.data(), d_rows, total * sizeof(impg_gpu_interval_t), hipMemcpyDeviceToHost, s));
  } else {
    for (uint32_t q = 0; q < n; q++)
      if (cnt[q])
        IMPG_HIP(hipMemcpyAsync(res.intervals.data() + res.offsets[q], d_rows + base[q], (size_t)cnt[q] * sizeof(impg_gpu_interval_t),
                                hipMemcpyDeviceToHost, s));
  }
  IMPG_HIP(hipStreamSynchronize(s));
  res.has_cigar = false;
  res.projected = E.last_projected;
  (void)h_ranges;
  return true;
}
}  // namespace

namespace impg {
std::vector<std::string> bed_range_names(const impg_gpu_index &ix, const impg_gpu_range_t *ranges, const char *const *range_names,
                                         size_t b, size_t e) {
  std::vector<std::string> rn(e - b);
  char buf[64];
  for (size_t i = b; i < e; i++) {
    if (range_names && range_names[i]) rn[i - b] = range_names[i];
    else {
      const impg_gpu_range_t &q = ranges[i];
      rn[i - b] = q.target_id < ix.seq.names.size() ? ix.seq.names[q.target_id] : std::to_string(q.target_id);
      const int k = snprintf(buf, sizeof buf, ":%d-%d", q.start, q.end);
      rn[i - b].append(buf, (size_t)k);
    }
  }
  return rn;
}
void check_ranges(const impg_gpu_range_t *ranges, size_t n) {
  if (n >= (1ull << 31)) throw Error{IMPG_E_UNSUPPORTED, "more than 2^31 ranges in one batch"};
  for (size_t i = 0; i < n; i++)
    if (ranges[i].start >= ranges[i].end) throw Error{IMPG_E_INVALID, "query range must satisfy start < end"};
}
// part's rows behind res's.  The first part hands its arrays over; later ones are copied behind them (the
// destination grows geometrically, pinned like its parts).
template <class T> static void append_arr(HostArr<T> &dst, HostArr<T> &src, size_t drop_front = 0) {
  if (dst.empty() && !drop_front) { dst.swap(src); return; }
  const size_t add = src.size() - drop_front, base = dst.size();
  if ((base + add) * sizeof(T) > dst.cap) dst.reserve(std::max(base + add, base + base / 2), dst.pinned || src.pinned);
  dst.n = base + add;
  parallel_memcpy(dst.data() + base, src.data() + drop_front, add * sizeof(T));
}
void append_results(impg_gpu_results &res, impg_gpu_results &part) {
  if (res.offsets.empty()) res.offsets.assign(1, 0);
  if (part.offsets.empty()) part.offsets.assign(1, 0);
  const uint64_t base = res.intervals.size();
  if (part.has_cigar) {
    res.has_cigar = true;
    if (res.cigar_off.empty()) res.cigar_off.assign((size_t)1, (uint64_t)0);
    const uint64_t cb = res.cigar_ops.size();
    const size_t at = res.cigar_off.size();
    if (part.cigar_off.size() > 1) {
      if (at == 1 && cb == 0) res.cigar_off.swap(part.cigar_off);
      else {
        append_arr(res.cigar_off, part.cigar_off, 1);
        for (size_t i = at; i < res.cigar_off.size(); i++) res.cigar_off[i] += cb;
      }
    }
    append_arr(res.cigar_ops, part.cigar_ops);
  }
  append_arr(res.intervals, part.intervals);
  if (base == 0 && res.offsets.size() == 1) res.offsets.swap(part.offsets);
  else for (size_t i = 1; i < part.offsets.size(); i++) res.offsets.push_back(base + part.offsets[i]);
  res.projected += part.projected;
  res.run_s += part.run_s;
  res.assemble_s += part.assemble_s;
}
}  // namespace impg

extern "C" {

const char *impg_gpu_last_error(void) { return impg::g_error.c_str(); }

int impg_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int impg_gpu_index_create(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                          const int64_t *seq_len, uint32_t n_seq, int bidirectional, int order_policy, int device,
                          impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out) throw Error{IMPG_E_INVALID, "null out"};
  *out = make_index(records, n_records, cigar_ops, n_ops, seq_len, n_seq, bidirectional, order_policy, device, 0, 1, nullptr,
                    nullptr, nullptr).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_files(const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                                const int64_t *seq_len, uint32_t n_seq, const uint64_t *file_first_record, uint32_t n_files,
                                int bidirectional, int order_policy, int device, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !file_first_record || n_files == 0) throw Error{IMPG_E_INVALID, "bad arguments"};
  std::vector<uint64_t> ff(file_first_record, file_first_record + n_files);
  ff.push_back(n_records);
  *out = make_index(records, n_records, cigar_ops, n_ops, seq_len, n_seq, bidirectional, order_policy, device, 0, 1, nullptr,
                    &ff, nullptr).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_tracepoints(const impg_gpu_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                      const int32_t *query_deltas, const int32_t *diffs, size_t n_segs_total,
                                      const impg_gpu_tp_mode_t *mode, const int64_t *seq_len, uint32_t n_seq, int bidirectional,
                                      int order_policy, int device, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !mode || (n_records && !records) || (n_segs_total && !tracepoints) || (n_seq && !seq_len))
    throw Error{IMPG_E_INVALID, "null argument"};
  if (mode->fastga) {
    if (n_segs_total && !diffs) throw Error{IMPG_E_INVALID, "FASTGA tracepoints come with per-segment diffs"};
    if (mode->trace_spacing <= 0) throw Error{IMPG_E_INVALID, "trace_spacing must be positive"};
  } else if (n_segs_total && !query_deltas) throw Error{IMPG_E_INVALID, "Standard tracepoints come with per-segment query deltas"};
  require_device(device);
  // the shared record shape: cigar_off / cigar_len name the alignment's segments
  std::vector<impg_gpu_record_t> recs(n_records);
  for (size_t i = 0; i < n_records; i++) {
    const impg_gpu_tp_record_t &r = records[i];
    if (r.seg_off + r.n_segs > n_segs_total) throw Error{IMPG_E_INVALID, "record segments outside the pools"};
    recs[i] = impg_gpu_record_t{r.query_id, r.target_id, r.query_start, r.query_end, r.target_start, r.target_end, r.seg_off, r.n_segs,
                                r.strand};
  }
  TpInput tp{records, tracepoints, query_deltas, diffs, n_segs_total, *mode};
  auto ix = std::make_unique<impg_gpu_index>();
  ix->device = device;
  build_index(*ix, recs.data(), n_records, nullptr, n_segs_total, seq_len, n_seq, bidirectional != 0, order_policy, 0, 1, nullptr, &tp);
  { EngineLease warm(*ix); }
  *out = ix.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_create_from_paf(const char *const *paths, int n_paths, int bidirectional, int order_policy, int device,
                                   impg_gpu_index_t **out) {
  IMPG_TRY
  if (!out || !paths || n_paths <= 0) throw Error{IMPG_E_INVALID, "bad arguments"};
  require_device(device);
  ParsedPaf pp;
  std::vector<std::string> ps(paths, paths + n_paths);
  const auto t0 = std::chrono::steady_clock::now();
  const bool timing = getenv("IMPG_BUILD_TIMING") != nullptr;
  auto lap = [&](const char *what, std::chrono::steady_clock::time_point from) {
    if (timing) fprintf(stderr, "[build] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - from).count());
  };
  // The CIGAR text is tokenised on the device (parse_cigar_to_delta, impg.rs:2935-2950 -> cigar_tokens_kernel): the
  // host only splits lines and fields, the text crosses PCIe once and the ops never exist on the host.  Inputs whose
  // text and ops would not fit next to the index they become (or IMPG_BUILD_HOST=1) are tokenised by the host parser.
  bool raw = getenv("IMPG_BUILD_HOST") == nullptr;
  if (raw) {
    uint64_t bytes = 0;
    for (auto &p : ps) {
      struct stat st;
      if (stat(p.c_str(), &st) == 0) bytes += (uint64_t)st.st_size * ((p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0) ? 6 : 1);
    }
    size_t free_b = 0, total_b = 0;
    IMPG_HIP(hipSetDevice(device));
    IMPG_HIP(hipMemGetInfo(&free_b, &total_b));
    if (bytes * 8 > free_b) raw = false;  // text + ops (<= 2 bytes of text per op) + the index (~3.4 x the ops)
  }
  parse_paf_files(ps, pp, raw);
  lap(raw ? "parse PAF (lines, fields)" : "parse PAF", t0);
  std::vector<int64_t> lens = pp.seq.lens;
  if (raw) {
    const auto t1 = std::chrono::steady_clock::now();
    DevBuf d_ops;
    const uint64_t n_ops = tokenize_on_device(pp, device, d_ops);
    lap("CIGAR text -> ops (device)", t1);
    auto ix = std::make_unique<impg_gpu_index>();
    ix->device = device;
    ix->seq = pp.seq;
    if (pp.file_first.size() < 2 || pp.file_first.front() != 0 || pp.file_first.back() != pp.records.size())
      throw Error{IMPG_E_INVALID, "internal: bad file boundaries"};
    ix->file_first = pp.file_first;
    if (order_policy != IMPG_ORDER_COITREES && order_policy != IMPG_ORDER_SORTED) throw Error{IMPG_E_INVALID, "bad order policy"};
    if (build_index_device(*ix, pp.records.data(), pp.records.size(), nullptr, n_ops, lens.data(), (uint32_t)lens.size(), bidirectional != 0,
                           order_policy, 0, 1, nullptr, d_ops.as<uint32_t>())) {
      { EngineLease warm(*ix); }
      *out = ix.release();
      return IMPG_OK;
    }
    // (the device was short of memory for the build: the ops come back and the host builder takes over)
    pp.ops.resize(n_ops);
    if (n_ops) IMPG_HIP(hipMemcpy(pp.ops.data(), d_ops.p, n_ops * 4, hipMemcpyDeviceToHost));
  }
  *out = make_index(pp.records.data(), pp.records.size(), pp.ops.data(), pp.ops.size(), lens.data(), (uint32_t)lens.size(),
                    bidirectional, order_policy, device, 0, 1, &pp.seq, &pp.file_first, nullptr).release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_parse_subsequence(const char *seq_name, char *base_out, size_t base_cap, int32_t *start_offset) {
  IMPG_TRY
  if (!seq_name || !base_out || !start_offset) throw Error{IMPG_E_INVALID, "null argument"};
  size_t base_len = 0;
  uint32_t off = 0;
  const std::string name(seq_name);
  if (!subsequence_origin(name, base_len, off)) return 0;
  if (base_len + 1 > base_cap) throw Error{IMPG_E_INVALID, "name buffer too small"};
  memcpy(base_out, name.data(), base_len);
  base_out[base_len] = '\0';
  *start_offset = (int32_t)off;
  return 1;
  IMPG_CATCH
}

int impg_gpu_subset_keep(const char *list_text, size_t len, const char *const *names, size_t n, uint8_t *keep_out,
                         size_t *n_entries) {
  IMPG_TRY
  if ((!list_text && len) || (n && (!names || !keep_out))) throw Error{IMPG_E_INVALID, "null argument"};
  const size_t e = subset_select(list_text, len, names, n, keep_out);
  if (n_entries) *n_entries = e;
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_save(const impg_gpu_index_t *ix, const char *path) {
  IMPG_TRY
  if (!ix || !path) throw Error{IMPG_E_INVALID, "null argument"};
  if (ix->shard || ix->cluster) save_sharded(*ix, path);
  else save_index(*ix, path);
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_index_load(const char *path, int device, impg_gpu_index_t **out) {
  IMPG_TRY
  if (!path || !out) throw Error{IMPG_E_INVALID, "null argument"};
  require_device(device);
  auto ix = std::make_unique<impg_gpu_index>();
  ix->device = device;
  load_index(*ix, path);
  { EngineLease warm(*ix); }
  *out = ix.release();
  return IMPG_OK;
  IMPG_CATCH
}

void impg_gpu_index_destroy(impg_gpu_index_t *ix) { delete ix; }

uint32_t impg_gpu_num_seqs(const impg_gpu_index_t *ix) { return (uint32_t)ix->seq.lens.size(); }
const char *impg_gpu_seq_name(const impg_gpu_index_t *ix, uint32_t id) {
  return id < ix->seq.names.size() ? ix->seq.names[id].c_str() : nullptr;
}
int64_t impg_gpu_seq_len(const impg_gpu_index_t *ix, uint32_t id) { return id < ix->seq.lens.size() ? ix->seq.lens[id] : -1; }
int64_t impg_gpu_seq_id(const impg_gpu_index_t *ix, const char *name) {
  auto it = ix->seq.name_to_id.find(name);
  return it == ix->seq.name_to_id.end() ? -1 : (int64_t)it->second;
}
size_t impg_gpu_num_targets(const impg_gpu_index_t *ix) { return ix->n_targets; }
size_t impg_gpu_target_ids(const impg_gpu_index_t *ix, uint32_t *out, size_t cap) {
  size_t k = 0;
  for (uint32_t s = 0; s + 1 < ix->h_tgt_off.size(); s++)
    if (ix->h_tgt_off[s + 1] != ix->h_tgt_off[s]) {
      if (out && k < cap) out[k] = s;
      k++;
    }
  return k;
}
size_t impg_gpu_num_entries(const impg_gpu_index_t *ix) { return ix->n_entries; }
size_t impg_gpu_num_records(const impg_gpu_index_t *ix) { return ix->n_records; }
size_t impg_gpu_device_bytes(const impg_gpu_index_t *ix) { return ix->device_bytes; }
int impg_gpu_index_approximate(const impg_gpu_index_t *ix) {
  return ix->tp_mode ? 1 : 0;
}

uint64_t impg_gpu_host_pool_trim(uint64_t keep_bytes) { return (uint64_t)impg::pinned_trim((size_t)keep_bytes); }

int impg_gpu_set_option(impg_gpu_index_t *ix, const char *key, int64_t value) {
  IMPG_TRY
  if (!ix || !key) throw Error{IMPG_E_INVALID, "null argument"};
  std::string k(key);
  if (k == "pair_budget") {
    if (value < 1024 || value >= 0xFFFFFFF0ll) throw Error{IMPG_E_INVALID, "pair_budget out of range"};
    ix->opt_pair_budget = (uint64_t)value;
  } else if (k == "chunk_ranges") {
    if (value < 0 || value >= (1ll << 31)) throw Error{IMPG_E_INVALID, "chunk_ranges out of range"};
    ix->opt_chunk_ranges = (uint32_t)value;
  } else if (k == "locality_min") {  // frontier size from which the projection runs in window order (0 = never)
    if (value < 0 || value >= (1ll << 31)) throw Error{IMPG_E_INVALID, "locality_min out of range"};
    ix->opt_locality_min = (uint32_t)value;
  } else if (k == "device_rows_pool_bytes") {  // HBM an engine keeps between impg_gpu_query_batch_device calls (freed slot arrays, reused by the next call)
    if (value < 0) throw Error{IMPG_E_INVALID, "device_rows_pool_bytes must not be negative"};
    ix->opt_device_rows_pool = (uint64_t)value;
  } else if (k == "fuse_final_level") {  // a counting run's final level enumerates its pairs from the count pass's windows: no emit pass (results identical)
    ix->opt_fuse_final = value != 0;
  } else if (k == "regroup_entries") {  // projection blocks regroup their pairs by entry before reading the index (results identical)
    ix->opt_regroup = value != 0;
  } else if (k == "walk_kernel") {  // the per-query walk (walk_device.inc): 0 never, 1 DFS batches of any size and depth-limited BFS batches of <= 64 ranges (default), 2 every BFS batch of <= 64 ranges
    if (value < 0 || value > 2) throw Error{IMPG_E_INVALID, "walk_kernel is 0, 1 or 2"};
    ix->opt_walk = (int)value;
  } else if (k == "segment_groups") {  // the update's hits grouped query by query (1, default) or by the library's radix sort (0); results identical
    ix->opt_seg_group = value != 0;
  } else if (k == "segment_parts") {  // slices a query's hits are grouped in: 0 = from the level's size (default), n = that many on every level that groups by segments (testing; results identical)
    if (value < 0 || value > 4096) throw Error{IMPG_E_INVALID, "segment_parts is 0 .. 4096"};
    ix->opt_seg_parts = (uint32_t)value;
  } else if (k == "walk_members") {  // workgroups per query of the walk's grid form (depth-limited BFS, <= 64 ranges): 0 = as many as fit (<= 32), 1 = no grid form
    if (value < 0 || value > (long long)WALK_MAX_MEMBERS) throw Error{IMPG_E_INVALID, "walk_members is 0 .. 64"};
    ix->opt_walk_members = (uint32_t)value;
  } else if (k == "filter_covered") {  // visited update: hits covered by the old list dropped before the replay (0 off, 1 always, 2 auto; results identical)
    if (value < 0 || value > 2) throw Error{IMPG_E_INVALID, "filter_covered is 0, 1 or 2"};
    ix->opt_filter_covered = (int)value;
  } else if (k == "free_slot_order") {  // counting runs lay their slots out in projection order (1, default) or keep the reference order (0)
    ix->opt_free_slots = value != 0;
  } else if (k == "debug_fail_owner" || k == "debug_fail_home") {  // tests: (rank + 1) << 16 | hop (sharded indexes; 0 = off)
    if (value < 0 || value > 0xFFFFFFFFll) throw Error{IMPG_E_INVALID, "debug_fail_* out of range"};
    (k == "debug_fail_owner" ? ix->opt_debug_fail_owner : ix->opt_debug_fail_home) = (uint32_t)value;
  } else if (k == "lane_schedule") {  // tests: forced lane start / hand-over order of a sharded batch (0 = off)
    if (value < 0) throw Error{IMPG_E_INVALID, "lane_schedule out of range"};
    ix->opt_lane_schedule = (uint64_t)value;
  } else if (k == "prewarm_result_bytes") {
    // A pinned host block of this size goes into the library's pool now (host_mem.cpp), so that a process's FIRST
    // result-returning call copies into a recycled block like every later one (pinning 5 GB costs ~0.3-1 s, inside the
    // call otherwise).  Blocks beyond IMPG_PINNED_POOL_BYTES (6 GiB) are not kept: ask for what the calls return.
    if (value < 0) throw Error{IMPG_E_INVALID, "prewarm_result_bytes is a size"};
    if (value) {
      size_t cap = 0;
      void *b = pinned_take((size_t)value, cap);
      pinned_give(b, cap);
    }
  } else if (k == "prewarm_walk") {
    // The per-query walk's slabs (walk_device.inc) exist before the first call that needs them: 1 = the 64 slabs of the
    // per-call / small-batch BFS shape, 2 = also the DFS batch's slabs (~15 GB: one per resident wave).
    if (value < 0 || value > 2) throw Error{IMPG_E_INVALID, "prewarm_walk is 0, 1 or 2"};
    if (value && !ix->shard && !ix->cluster) {
      IMPG_HIP(hipSetDevice(ix->device));
      EngineLease lease(*ix);
      lease->reserve_walk_slabs(*ix, value >= 2);
    }
  } else throw Error{IMPG_E_INVALID, "unknown option " + k};
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_get_counter(const impg_gpu_index_t *ix, const char *key, int64_t *value_out) {
  IMPG_TRY
  if (!ix || !key || !value_out) throw Error{IMPG_E_INVALID, "null argument"};
  const std::string k = key;
  if (k == "walk_launches") *value_out = (int64_t)ix->walk_launches.load();
  else if (k == "walk_fallbacks") *value_out = (int64_t)ix->walk_fallbacks.load();
  else if (k == "walk_members") *value_out = (int64_t)ix->walk_last_members.load();
  else if (k == "segment_sliced_levels") *value_out = (int64_t)ix->seg_stats[0].load();
  else if (k == "segment_retries") *value_out = (int64_t)ix->seg_stats[1].load();
  else if (k == "segment_library_levels") *value_out = (int64_t)ix->seg_stats[2].load();
  else throw Error{IMPG_E_INVALID, "unknown counter " + k};
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_visit_rank(uint32_t n, int order_policy, uint32_t *rank_out) {
  IMPG_TRY
  if (!rank_out && n) throw Error{IMPG_E_INVALID, "null argument"};
  if (order_policy == IMPG_ORDER_COITREES) coitrees_visit_rank(n, rank_out);
  else if (order_policy == IMPG_ORDER_SORTED) for (uint32_t i = 0; i < n; i++) rank_out[i] = i;
  else throw Error{IMPG_E_INVALID, "bad order policy"};
  return IMPG_OK;
  IMPG_CATCH
}

}  // extern "C"
namespace impg {
// masked_regions -> the engine's device tables (EngineLease clears the flag when the call ends)
void apply_mask(Engine &E, const impg_gpu_index &ix, const impg_gpu_mask_t *m, const impg_gpu_params_t &p) {
  {
    if (!m) return;
    if (!p.transitive) throw Error{IMPG_E_INVALID, "masked_regions belong to the transitive queries"};
    const uint32_t n_seq = ix.view.n_seq;
    if (m->n_seqs && (!m->seq_id || !m->sequence_length || !m->range_off)) throw Error{IMPG_E_INVALID, "null mask array"};
    const uint64_t total = m->n_seqs ? m->range_off[m->n_seqs] : 0;
    if (total >= 0xFFFFFFF0ull) throw Error{IMPG_E_UNSUPPORTED, "mask exceeds 2^32 ranges"};
    if (total && !m->ranges) throw Error{IMPG_E_INVALID, "null mask array"};
    std::vector<uint32_t> off(n_seq + 1, 0);
    // a sequence absent from the map: Impg starts its set with length 0 (visited_entry, impg.rs:2048-2053), MultiImpg
    // with the real length (multi_impg.rs:919-922) except for the query's own target (entry().or_default(), :827-830)
    std::vector<int32_t> init_len(n_seq, 0), touch_len(n_seq, 0);
    bool has_empty = false;
    if (p.multi_impg) for (uint32_t s = 0; s < n_seq; s++) touch_len[s] = (int32_t)std::min<int64_t>(std::max<int64_t>(ix.seq.lens[s], 0), INT32_MAX);
    for (uint32_t i = 0; i < m->n_seqs; i++) {
      const uint32_t s = m->seq_id[i];
      if (s >= n_seq) throw Error{IMPG_E_INVALID, "mask names an unknown sequence id"};
      if (i && s <= m->seq_id[i - 1]) throw Error{IMPG_E_INVALID, "mask sequence ids must be strictly ascending"};
      if (m->range_off[i + 1] < m->range_off[i]) throw Error{IMPG_E_INVALID, "mask offsets must not decrease"};
      off[s + 1] = (uint32_t)(m->range_off[i + 1] - m->range_off[i]);
      init_len[s] = touch_len[s] = m->sequence_length[i];
      for (uint64_t k = m->range_off[i]; k < m->range_off[i + 1]; k++) {
        const int32_t a = m->ranges[2 * k], b = m->ranges[2 * k + 1];
        if (a > b || (k > m->range_off[i] && a <= m->ranges[2 * k - 1]))  // SortedRanges invariant (impg.rs:330-368)
          throw Error{IMPG_E_INVALID, "mask ranges must be sorted, disjoint and non-touching"};
        has_empty = has_empty || a == b;
      }
    }
    for (uint32_t s = 0; s < n_seq; s++) off[s + 1] += off[s];
    // m->ranges is already in sequence-id order
    E.mask_off.reserve((size_t)(n_seq + 1) * 4);
    E.mask_ranges.reserve(std::max<size_t>(total * 8, 256));
    E.mask_init_len.reserve(std::max<size_t>((size_t)n_seq * 4, 256));
    E.mask_touch_len.reserve(std::max<size_t>((size_t)n_seq * 4, 256));
    IMPG_HIP(hipMemcpy(E.mask_off.p, off.data(), (size_t)(n_seq + 1) * 4, hipMemcpyHostToDevice));
    if (total) IMPG_HIP(hipMemcpy(E.mask_ranges.p, m->ranges, total * 8, hipMemcpyHostToDevice));
    if (n_seq) {
      IMPG_HIP(hipMemcpy(E.mask_init_len.p, init_len.data(), (size_t)n_seq * 4, hipMemcpyHostToDevice));
      IMPG_HIP(hipMemcpy(E.mask_touch_len.p, touch_len.data(), (size_t)n_seq * 4, hipMemcpyHostToDevice));
    }
    E.masked = true;
    E.mask_has_empty = has_empty;
    E.mask_ranges_total = total;
    E.mask_lists = m->n_seqs;
  }
}
void apply_subset(Engine &E, const impg_gpu_index &ix, const uint8_t *subset_keep) {
  if (!subset_keep) return;
  const size_t ns = ix.view.n_seq;
  E.subset_keep.reserve(std::max<size_t>(ns, 256));
  if (ns) IMPG_HIP(hipMemcpy(E.subset_keep.p, subset_keep, ns, hipMemcpyHostToDevice));
  E.subset_on = true;
}
}  // namespace impg
extern "C" {

int impg_gpu_query_batch(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                         impg_gpu_results_t **out) {
  return impg_gpu_query_batch_masked(ix, ranges, n, params, nullptr, out);
}

int impg_gpu_query_batch_masked(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n,
                                const impg_gpu_params_t *params, const impg_gpu_mask_t *mask, impg_gpu_results_t **out) {
  return impg_gpu_query_batch_filtered(ix, ranges, n, params, mask, nullptr, out);
}

int impg_gpu_query_batch_filtered(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n,
                                  const impg_gpu_params_t *params, const impg_gpu_mask_t *mask, const uint8_t *subset_keep,
                                  impg_gpu_results_t **out) {
  IMPG_TRY
  if (!ix || !params || !out || (!ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
  check_ranges(ranges, n);
  Engine::check_params(*params);
  if (ix->shard || ix->cluster) return sharded_query_batch(*ix, ranges, n, *params, mask, subset_keep, out);
  IMPG_HIP(hipSetDevice(ix->device));
  EngineLease lease(*ix);
  Engine &E = *lease;
  apply_mask(E, *ix, mask, *params);
  apply_subset(E, *ix, subset_keep);
  auto res = std::make_unique<impg_gpu_results>();
  if (n && n <= Engine::SMALL_RANGES && !params->transitive) {  // the per-call shape: one chain of launches, one sync
    const auto c0 = std::chrono::steady_clock::now();
    if (E.run_small(*ix, ranges, (uint32_t)n, *params, *res)) {
      res->run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
      *out = res.release();
      return IMPG_OK;
    }
  }
  E.ranges_dev.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
  if (n) IMPG_HIP(hipMemcpyAsync(E.ranges_dev.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, E.stream));
  if (n < (1ull << 31) && E.walk_applicable(*ix, (uint32_t)n, *params)) {  // small transitive batches, DFS batches: one launch
    const auto c0 = std::chrono::steady_clock::now();
    if (walk_query(*ix, E, ranges, (uint32_t)n, *params, *res)) {
      res->run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
      res->ranges.assign(ranges, ranges + n);
      *out = res.release();
      return IMPG_OK;
    }
    res = std::make_unique<impg_gpu_results>();  // (a query outgrew its slab: the batch engine takes the batch)
  }
  res->offsets.assign(1, 0);
  for_chunks(E, n, [&](size_t b, size_t e) {
    std::vector<std::unique_ptr<LevelBufs>> levels;
    DevBuf self_dev;
    self_dev.pool = &E.level_pool;
    const auto c0 = std::chrono::steady_clock::now();
    E.run(*ix, E.ranges_dev.as<impg_gpu_range_t>() + b, (uint32_t)(e - b), *params, &levels, nullptr, nullptr, nullptr, &self_dev);
    const auto c1 = std::chrono::steady_clock::now();
    impg_gpu_results part;
    assemble_results(E, ranges + b, (uint32_t)(e - b), *params, levels, self_dev, part);
    part.run_s = std::chrono::duration<double>(c1 - c0).count();
    part.assemble_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c1).count();
    append_results(*res, part);
  });
  res->ranges.assign(ranges, ranges + n);
  *out = res.release();
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_query(impg_gpu_index_t *ix, uint32_t target_id, int32_t start, int32_t end, const impg_gpu_params_t *params,
                   impg_gpu_results_t **out) {
  impg_gpu_range_t r{target_id, start, end};
  return impg_gpu_query_batch(ix, &r, 1, params, out);
}

// The trait's rows for a batch that does not fit one result object (the headline's 100 000 ranges return 2.1 x 10^9 rows,
// 51 GB; the reference prints range by range, main.rs:7435-7470): chunks of the batch are computed on two engines in
// turn, each chunk's rows placed on the device and copied into that engine's pinned block while the other engine
// computes the next chunk, and handed to the callback in range order.
int impg_gpu_query_batch_stream(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                                const impg_gpu_mask_t *mask, const uint8_t *subset_keep, size_t chunk_ranges, size_t max_block_bytes,
                                impg_gpu_stream_cb cb, void *ctx, uint64_t *projected_out) {
  IMPG_TRY
  if (!ix) throw Error{IMPG_E_INVALID, "null argument"};
  // On a rank's shard the call is collective and opens with an agreement on the number of chunks: a rank whose own
  // arguments are refused must still take part in it -- its peers are already waiting there -- and says so with the word it
  // contributes, so that every rank leaves together (the refused one with its own error).
  const bool collective = ix->shard && !ix->cluster;
  std::exception_ptr refused;
  try {
    if (!params || !cb || (!ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
    check_ranges(ranges, n);
    Engine::check_params(*params);
  } catch (...) {
    if (!collective) throw;
    refused = std::current_exception();
  }
  if (!chunk_ranges) chunk_ranges = 8192;
  if (!max_block_bytes) max_block_bytes = 5ull << 29;  // 2.5 GiB: two such blocks go back into the pinned pool (6 GiB) after the call
  const uint64_t max_rows = std::max<uint64_t>(1, max_block_bytes / sizeof(impg_gpu_interval_t));
  uint64_t projected = 0;
  if (ix->shard || ix->cluster) {
    // a sharded index: the chunks one after the other through the collective call.  On a rank's shard (one process per
    // GPU) every chunk is a collective: the ranks agree on the number of calls first -- each brings its own n, so the
    // counts differ -- and a rank that has run out of ranges, or whose consumer has stopped it, keeps taking part with
    // empty chunks until the longest stream is through (its peers' hops need its shard).
    uint64_t my_chunks = std::max<uint64_t>(1, (n + chunk_ranges - 1) / chunk_ranges), n_chunks = my_chunks;
    if (collective) {
      n_chunks = shard_agree_max(*ix, refused ? ~0ull : my_chunks);
      if (refused) std::rethrow_exception(refused);
      if (n_chunks == ~0ull) throw Error{IMPG_E_INVALID, "a peer rank's arguments to the row stream were refused: no rank runs it"};
    }
    bool cancelled = false;
    for (uint64_t c = 0; c < n_chunks; c++) {
      const size_t b = cancelled ? n : std::min<size_t>(n, (size_t)c * chunk_ranges), e = cancelled ? n : std::min(n, b + chunk_ranges);
      impg_gpu_results_t *part = nullptr;
      const int rc = sharded_query_batch(*ix, ranges + b, e - b, *params, mask, subset_keep, &part);
      if (rc != IMPG_OK) return rc;
      std::unique_ptr<impg_gpu_results> own(part);
      projected += part->projected;
      if (e > b && cb(ctx, part, b) != 0) {
        cancelled = true;
        if (ix->cluster) break;  // (one caller, no peers waiting)
      }
    }
    if (cancelled) throw Error{IMPG_E_CANCELLED, "the row stream's consumer stopped it"};
    if (projected_out) *projected_out = projected;
    return IMPG_OK;
  }
  IMPG_HIP(hipSetDevice(ix->device));
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    size_t next_begin = 0, chunk = 0;
    uint64_t next_seq = 0, deliver_seq = 0;
    bool stop = false;
    std::mutex gpu;  // whose kernels are on the GPU (two chunks' kernels side by side evict each other's lines from L2)
    std::exception_ptr err;
    uint64_t projected = 0;
  } sh;
  sh.chunk = chunk_ranges;
  auto worker = [&]() {
    try {
      IMPG_HIP(hipSetDevice(ix->device));
      EngineLease lease(*ix);
      Engine &E = *lease;
      apply_mask(E, *ix, mask, *params);
      apply_subset(E, *ix, subset_keep);
      E.ranges_dev.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
      if (n) IMPG_HIP(hipMemcpyAsync(E.ranges_dev.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, E.stream));
      struct Ev {  // (an event of the call's own: the engine recycles its pool run by run)
        hipEvent_t e = nullptr;
        Ev() { IMPG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
        ~Ev() { if (e) (void)hipEventDestroy(e); }
      } done_ev;
      hipEvent_t done = done_ev.e;
      impg_gpu_results part;  // reused chunk after chunk: its pinned arrays grow to the largest chunk and stay
      for (;;) {
        size_t b, e;
        uint64_t seq;
        {
          std::unique_lock<std::mutex> lk(sh.m);
          if (sh.stop || sh.next_begin >= n) break;
          b = sh.next_begin; e = std::min(n, b + sh.chunk);
          sh.next_begin = e; seq = sh.next_seq++;
        }
        // the chunk, in pieces if it outgrows the pair budget or the block (pieces are delivered in order within the turn)
        std::vector<std::pair<size_t, size_t>> todo{{b, e}};
        while (!todo.empty()) {
          const auto [pb, pe] = todo.back();
          todo.pop_back();
          std::unique_lock<std::mutex> turn(sh.gpu);
          bool released = false;
          E.on_kernels_done = [&]() { if (!released) { released = true; turn.unlock(); } };
          try {
            std::vector<std::unique_ptr<LevelBufs>> levels;
            DevBuf self_dev;
            self_dev.pool = &E.level_pool;
            const auto c0 = std::chrono::steady_clock::now();
            E.run(*ix, E.ranges_dev.as<impg_gpu_range_t>() + pb, (uint32_t)(pe - pb), *params, &levels, nullptr, nullptr, nullptr, &self_dev);
            const auto c1 = std::chrono::steady_clock::now();
            part.offsets.clear(); part.intervals.clear(); part.cigar_off.clear(); part.cigar_ops.clear();
            assemble_results(E, ranges + pb, (uint32_t)(pe - pb), *params, levels, self_dev, part, max_rows, done);
            part.run_s = std::chrono::duration<double>(c1 - c0).count();
            part.assemble_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - c1).count();
          } catch (const SplitBatch &) {
            E.on_kernels_done = nullptr;
            if (pe - pb <= 1) throw Error{IMPG_E_UNSUPPORTED, "a single range exceeds the pair budget"};
            const size_t mid = pb + (pe - pb) / 2;
            {  // (the chunks still to be dealt start at the size that fitted: a split costs the run that found it out)
              std::lock_guard<std::mutex> lk(sh.m);
              sh.chunk = std::min(sh.chunk, std::max<size_t>(1, (pe - pb) / 2));
            }
            todo.push_back({mid, pe});
            todo.push_back({pb, mid});
            continue;
          }
          E.on_kernels_done = nullptr;
          if (!released) turn.unlock();
          // in range order: this chunk's turn comes when every earlier chunk has been handed over
          std::unique_lock<std::mutex> lk(sh.m);
          sh.cv.wait(lk, [&] { return sh.stop || sh.deliver_seq == seq; });
          if (sh.stop) break;
          sh.projected += part.projected;
          lk.unlock();
          if (cb(ctx, &part, pb) != 0) throw Error{IMPG_E_CANCELLED, "the row stream's consumer stopped it"};
        }
        std::lock_guard<std::mutex> lk(sh.m);
        if (sh.stop) break;
        sh.deliver_seq = seq + 1;
        sh.cv.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(sh.m);
      if (!sh.err) sh.err = std::current_exception();
      sh.stop = true;
      sh.cv.notify_all();
    }
  };
  if (n) {
    std::thread second(worker);
    worker();
    second.join();
  }
  if (sh.err) std::rethrow_exception(sh.err);
  if (projected_out) *projected_out = sh.projected;
  return IMPG_OK;
  IMPG_CATCH
}

size_t impg_gpu_results_num_ranges(const impg_gpu_results_t *r) { return r->offsets.empty() ? 0 : r->offsets.size() - 1; }
size_t impg_gpu_results_total(const impg_gpu_results_t *r) { return r->intervals.size(); }
const uint64_t *impg_gpu_results_offsets(const impg_gpu_results_t *r) { return r->offsets.data(); }
const impg_gpu_interval_t *impg_gpu_results_intervals(const impg_gpu_results_t *r) { return r->intervals.data(); }
uint64_t impg_gpu_results_projected(const impg_gpu_results_t *r) { return r->projected; }
const uint64_t *impg_gpu_results_cigar_offsets(const impg_gpu_results_t *r) { return r->has_cigar ? r->cigar_off.data() : nullptr; }
const uint32_t *impg_gpu_results_cigar_ops(const impg_gpu_results_t *r) { return r->has_cigar ? r->cigar_ops.data() : nullptr; }
void impg_gpu_results_timing(const impg_gpu_results_t *r, double *engine_s, double *assemble_s) {
  if (engine_s) *engine_s = r ? r->run_s : 0;
  if (assemble_s) *assemble_s = r ? r->assemble_s : 0;
}
void impg_gpu_results_free(impg_gpu_results_t *r) { delete r; }

static int stats_impl(impg_gpu_index_t *ix, Engine &E, const impg_gpu_range_t *d_ranges, size_t n, const impg_gpu_params_t *params,
                      uint64_t *per_range_count, uint64_t *per_range_checksum, impg_gpu_stats_t *stats) {
  unsigned long long *dc = nullptr, *dk = nullptr;
  if (per_range_count) {
    E.stat_count.reserve(std::max<size_t>(n * 8, 256));
    IMPG_HIP(hipMemsetAsync(E.stat_count.p, 0, n * 8, E.stream));
    dc = E.stat_count.as<unsigned long long>();
  }
  if (per_range_checksum) {
    E.stat_cksum.reserve(std::max<size_t>(n * 8, 256));
    IMPG_HIP(hipMemsetAsync(E.stat_cksum.p, 0, n * 8, E.stream));
    dk = E.stat_cksum.as<unsigned long long>();
  }
  impg_gpu_stats_t tot;
  memset(&tot, 0, sizeof tot);
  for_chunks(E, n, [&](size_t b, size_t e) {
    impg_gpu_stats_t st;
    // a split retry must not double-count the slice
    if (dc) IMPG_HIP(hipMemsetAsync(dc + b, 0, (e - b) * 8, E.stream));
    if (dk) IMPG_HIP(hipMemsetAsync(dk + b, 0, (e - b) * 8, E.stream));
    E.run(*ix, d_ranges + b, (uint32_t)(e - b), *params, nullptr, dc ? dc + b : nullptr, dk ? dk + b : nullptr, &st, nullptr);
    tot.projected += st.projected; tot.pairs += st.pairs; tot.frontier_ranges += st.frontier_ranges;
    tot.levels = std::max(tot.levels, st.levels);
    tot.ms_total += st.ms_total; tot.ms_lookup += st.ms_lookup; tot.ms_project += st.ms_project; tot.ms_update += st.ms_update;
    tot.project_launches += st.project_launches;
  });
  if (stats) *stats = tot;
  if (per_range_count && n) IMPG_HIP(hipMemcpy(per_range_count, dc, n * 8, hipMemcpyDeviceToHost));
  if (per_range_checksum && n) IMPG_HIP(hipMemcpy(per_range_checksum, dk, n * 8, hipMemcpyDeviceToHost));
  return IMPG_OK;
}

int impg_gpu_query_batch_stats(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                               uint64_t *per_range_count, uint64_t *per_range_checksum, impg_gpu_stats_t *stats) {
  IMPG_TRY
  if (!ix || !params || (!ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
  check_ranges(ranges, n);
  Engine::check_params(*params);
  if (ix->shard || ix->cluster) return sharded_query_stats(*ix, ranges, false, n, *params, per_range_count, per_range_checksum, stats);
  IMPG_HIP(hipSetDevice(ix->device));
  EngineLease lease(*ix);
  Engine &E = *lease;
  E.ranges_dev.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
  if (n) IMPG_HIP(hipMemcpyAsync(E.ranges_dev.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, E.stream));
  return stats_impl(ix, E, E.ranges_dev.as<impg_gpu_range_t>(), n, params, per_range_count, per_range_checksum, stats);
  IMPG_CATCH
}

int impg_gpu_query_batch_stats_dev(impg_gpu_index_t *ix, const impg_gpu_range_t *d_ranges, size_t n,
                                   const impg_gpu_params_t *params, uint64_t *per_range_count,
                                   uint64_t *per_range_checksum, impg_gpu_stats_t *stats) {
  IMPG_TRY
  if (!ix || !params || (!d_ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
  if (n >= (1ull << 31)) throw Error{IMPG_E_UNSUPPORTED, "more than 2^31 ranges in one batch"};
  Engine::check_params(*params);
  if (ix->cluster) throw Error{IMPG_E_INVALID, "device-resident ranges belong to one GPU: a multi-GPU handle takes host ranges"};
  if (ix->shard) return sharded_query_stats(*ix, d_ranges, true, n, *params, per_range_count, per_range_checksum, stats);
  IMPG_HIP(hipSetDevice(ix->device));
  EngineLease lease(*ix);
  return stats_impl(ix, *lease, d_ranges, n, params, per_range_count, per_range_checksum, stats);
  IMPG_CATCH
}

// ---- rows left in HBM ------------------------------------------------------------------------------------------------
// impg_gpu_query_batch_device: the batch's result rows stay on the device, complete -- what SURVEY.md 8(d) calls the result
// record: {range_idx, query_id, q_first, q_last, t_first, t_last} -- for a consumer that lives there too (the BED merges of
// bed_device.hip are one; a caller's own kernels another).  The handle keeps the engine it ran on (its buffers are the
// engine's pooled blocks) until it is freed.
struct impg_gpu_device_rows {
  // (declared first, destroyed last: the chunks' buffers go back to the engine's pool while the lease still holds it)
  std::unique_ptr<impg::EngineLease> lease;
  struct Chunk {
    size_t first = 0, n = 0;
    std::vector<std::unique_ptr<impg::LevelBufs>> levels;  // IMPG_ROWS_ATTRIBUTED
    impg::DevBuf rows, offsets;                            // IMPG_ROWS_ORDERED: impg_gpu_interval_t[n_rows], u32 [n + 1]
    uint64_t n_rows = 0;
  };
  std::vector<std::unique_ptr<Chunk>> chunks;  // (a chunk owns device buffers: not movable)
  struct Part { size_t chunk; size_t level; };
  std::vector<Part> parts;
  impg_gpu_index *ix = nullptr;
  impg_gpu_params_t params{};
  size_t n = 0;
  int layout = 0;
  impg_gpu_stats_t stats{};
  float ms_place = 0;  // ordered layout: HIP-event time of the row placement
};

int impg_gpu_query_batch_device(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, int ranges_on_device,
                                const impg_gpu_params_t *params, int layout, impg_gpu_device_rows_t **out) {
  IMPG_TRY
  if (!ix || !params || !out || (!ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
  if (layout != IMPG_ROWS_ATTRIBUTED && layout != IMPG_ROWS_ORDERED && layout != IMPG_ROWS_ORDERED_SLOTS) throw Error{IMPG_E_INVALID, "unknown row layout"};
  if (n >= (1ull << 31)) throw Error{IMPG_E_UNSUPPORTED, "more than 2^31 ranges in one batch"};
  if (!ranges_on_device) check_ranges(ranges, n);
  Engine::check_params(*params);
  if (ix->shard || ix->cluster)
    throw Error{IMPG_E_UNSUPPORTED, "rows left on the device belong to one GPU: a sharded index returns rows through impg_gpu_query_batch"};
  if (params->store_cigar || params->multi_impg || (params->transitive && params->dfs))
    throw Error{IMPG_E_UNSUPPORTED, "impg_gpu_query_batch_device takes Impg::query and query_transitive_bfs without store_cigar"};
  IMPG_HIP(hipSetDevice(ix->device));
  auto h = std::make_unique<impg_gpu_device_rows>();
  h->lease = std::make_unique<EngineLease>(*ix);
  Engine &E = **h->lease;
  // (the slot arrays of a big batch are tens of GB: they go back to the engine's pool when the handle is freed and are
  // the next call's -- the pool's default cap would hipFree / hipMalloc them call after call)
  E.level_pool.max_held = std::max<size_t>(E.level_pool.max_held, (size_t)ix->opt_device_rows_pool);
  h->ix = ix;
  h->params = *params;
  h->n = n;
  h->layout = layout;
  const impg_gpu_range_t *d_ranges = ranges;
  if (!ranges_on_device) {
    E.ranges_dev.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
    if (n) IMPG_HIP(hipMemcpyAsync(E.ranges_dev.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, E.stream));
    d_ranges = E.ranges_dev.as<impg_gpu_range_t>();
  }
  impg_gpu_stats_t tot;
  memset(&tot, 0, sizeof tot);
  for_chunks(E, n, [&](size_t b, size_t e) {
    auto cp = std::make_unique<impg_gpu_device_rows::Chunk>();
    impg_gpu_device_rows::Chunk &c = *cp;
    c.first = b; c.n = e - b;
    impg_gpu_stats_t st;
    DevBuf self_dev;
    self_dev.pool = &E.level_pool;
    // attributed: a slot names its frontier record (pair_range), so the final level may be fused like a counting run's;
    // ordered: every level's slots are runs in visit order, placed below
    E.keep_any_order = layout == IMPG_ROWS_ATTRIBUTED;
    E.ordered_rows = layout == IMPG_ROWS_ORDERED_SLOTS;  // rows placed slot by slot, the fused final level's by its own kernel
    try {
      E.run(*ix, d_ranges + b, (uint32_t)(e - b), *params, &c.levels, nullptr, nullptr, &st, layout != IMPG_ROWS_ATTRIBUTED ? &self_dev : nullptr);
    } catch (...) { E.keep_any_order = E.ordered_rows = false; throw; }
    E.keep_any_order = E.ordered_rows = false;
    if (layout == IMPG_ROWS_ORDERED_SLOTS) {
      c.rows.adopt(E.ord_rows);
      c.offsets.adopt(E.ord_offsets);
      c.n_rows = E.ord_total;
      h->ms_place += E.ms_place;
      c.levels.clear();
    }
    if (layout == IMPG_ROWS_ORDERED) {  // the trait's rows (rows_device.hip), left where they are built
      hipEvent_t r0 = E.event(), r1 = E.event();
      IMPG_HIP(hipEventRecord(r0, E.stream));
      RowPlan pl;
      plan_rows(E, (uint32_t)(e - b), *params, c.levels, self_dev, false, pl);
      c.rows.pool = &E.level_pool;
      c.rows.reserve(std::max<size_t>((size_t)pl.n_rows * sizeof(impg_gpu_interval_t), 256));
      scatter_rows(E, c.levels, pl, RowSinks{c.rows.as<impg_gpu_interval_t>(), nullptr, nullptr, nullptr, nullptr, nullptr});
      IMPG_HIP(hipEventRecord(r1, E.stream));
      IMPG_HIP(hipStreamSynchronize(E.stream));
      float ms = 0;
      IMPG_HIP(hipEventElapsedTime(&ms, r0, r1));
      h->ms_place += ms;
      c.offsets.adopt(pl.offsets);
      c.n_rows = pl.n_rows;
      c.levels.clear();
    }
    tot.projected += st.projected; tot.pairs += st.pairs; tot.frontier_ranges += st.frontier_ranges;
    tot.levels = std::max(tot.levels, st.levels);
    tot.ms_total += st.ms_total; tot.ms_lookup += st.ms_lookup; tot.ms_project += st.ms_project; tot.ms_update += st.ms_update;
    tot.project_launches += st.project_launches;
    h->chunks.push_back(std::move(cp));
  });
  for (size_t c = 0; c < h->chunks.size(); c++) {
    if (layout != IMPG_ROWS_ATTRIBUTED) h->parts.push_back({c, 0});
    else for (size_t l = 0; l < h->chunks[c]->levels.size(); l++) h->parts.push_back({c, l});
  }
  h->stats = tot;
  *out = h.release();
  return IMPG_OK;
  IMPG_CATCH
}

size_t impg_gpu_device_rows_num_parts(const impg_gpu_device_rows_t *h) { return h ? h->parts.size() : 0; }
int impg_gpu_device_rows_part(const impg_gpu_device_rows_t *h, size_t k, impg_gpu_device_part_t *out) {
  IMPG_TRY
  if (!h || !out || k >= h->parts.size()) throw Error{IMPG_E_INVALID, "no such part"};
  const auto &c = *h->chunks[h->parts[k].chunk];
  memset(out, 0, sizeof *out);
  out->first_range = c.first;
  out->n_ranges = c.n;
  if (h->layout != IMPG_ROWS_ATTRIBUTED) {
    out->n_slots = c.n_rows;
    out->rows = c.rows.as<impg_gpu_interval_t>();
    out->offsets = c.offsets.as<uint32_t>();
    return IMPG_OK;
  }
  const LevelBufs &L = *c.levels[h->parts[k].level];
  out->level = (uint32_t)h->parts[k].level;
  out->n_slots = L.n_pairs;
  out->n_frontier = L.n_frontier;
  out->query_id = L.qid.as<uint32_t>();
  out->coords = L.coords.as<int32_t>();
  // (a fused final level stores a slot's query id and source as one pair: one store instruction in the kernel instead of two)
  out->source = L.qs_interleaved ? L.qid.as<uint32_t>() + 1 : L.pair_range.as<uint32_t>();
  out->slot_stride = L.qs_interleaved ? 2u : 1u;
  out->frontier = L.frontier.as<impg_gpu_frontier_t>();
  return IMPG_OK;
  IMPG_CATCH
}
void impg_gpu_device_rows_stats(const impg_gpu_device_rows_t *h, impg_gpu_stats_t *stats) {
  if (h && stats) *stats = h->stats;
}
float impg_gpu_device_rows_place_ms(const impg_gpu_device_rows_t *h) { return h ? h->ms_place : 0.f; }
// the per-range counts and checksums of impg_gpu_query_batch_stats, recomputed from the rows the call left in HBM
int impg_gpu_device_rows_check(impg_gpu_device_rows_t *h, uint64_t *per_range_count, uint64_t *per_range_checksum) {
  IMPG_TRY
  if (!h) throw Error{IMPG_E_INVALID, "null argument"};
  if (h->layout != IMPG_ROWS_ATTRIBUTED) throw Error{IMPG_E_UNSUPPORTED, "impg_gpu_device_rows_check reads the attributed layout (ordered rows: compare them with impg_gpu_query_batch's)"};
  Engine &E = **h->lease;
  IMPG_HIP(hipSetDevice(h->ix->device));
  const size_t n = h->n;
  E.stat_count.reserve(std::max<size_t>(n * 8, 256));
  E.stat_cksum.reserve(std::max<size_t>(n * 8, 256));
  IMPG_HIP(hipMemsetAsync(E.stat_count.p, 0, std::max<size_t>(n * 8, 8), E.stream));
  IMPG_HIP(hipMemsetAsync(E.stat_cksum.p, 0, std::max<size_t>(n * 8, 8), E.stream));
  for (auto &cp : h->chunks)
    for (auto &Lp : cp->levels) {
      auto &c = *cp;
      LevelBufs &L = *Lp;
      if (!L.n_pairs) continue;
      HitArrays ha{L.qid.as<uint32_t>(), L.coords.as<int4>()};
      E.rstat.reserve(std::max<size_t>((size_t)L.n_frontier * 16, 256));
      launch_hit_stats(L.frontier.as<FrontierRec>(), L.n_frontier, L.qs_interleaved ? L.qid.as<uint32_t>() + 1 : L.pair_range.as<uint32_t>(), L.n_pairs, ha,
                       h->params.transitive ? h->params.min_output_length : -1, false, E.rstat.as<unsigned long long>(),
                       E.stat_count.as<unsigned long long>() + c.first, E.stat_cksum.as<unsigned long long>() + c.first, E.stream,
                       L.qs_interleaved ? 2u : 1u);
    }
  IMPG_HIP(hipStreamSynchronize(E.stream));
  if (per_range_count && n) IMPG_HIP(hipMemcpy(per_range_count, E.stat_count.p, n * 8, hipMemcpyDeviceToHost));
  if (per_range_checksum && n) IMPG_HIP(hipMemcpy(per_range_checksum, E.stat_cksum.p, n * 8, hipMemcpyDeviceToHost));
  return IMPG_OK;
  IMPG_CATCH
}
void impg_gpu_device_rows_free(impg_gpu_device_rows_t *h) {
  if (!h) return;
  if (h->ix) (void)hipSetDevice(h->ix->device);
  delete h;
}

long impg_gpu_bed_merge(impg_gpu_interval_t *iv, size_t n, int32_t merge_distance, int merge_strands) {
  try {
    return (long)bed_merge(iv, n, merge_distance, merge_strands != 0);
  } catch (...) {
    impg::set_error("bed_merge failed");
    return IMPG_E_OOM;
  }
}

int impg_gpu_results_bed(const impg_gpu_results_t *res, const impg_gpu_index_t *ix, const char *const *range_names,
                         const impg_gpu_params_t *params, int32_t merge_distance, char **text, size_t *len) {
  IMPG_TRY
  if (!res || !ix || !params || !text || !len) throw Error{IMPG_E_INVALID, "null argument"};
  std::vector<std::string> parts;
  render_bed(*res, *ix, range_names, *params, merge_distance, parts);
  *text = join_parts(parts, len);
  return IMPG_OK;
  IMPG_CATCH
}

// query + both BED merges + the text on the device: only text crosses PCIe, in pinned pieces
namespace {
void bed_batch(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params, const uint8_t *subset_keep,
               int32_t merge_distance, const char *const *range_names, const std::function<void(const char *, size_t)> &sink,
               double *seconds3) {
  if (!ix || !params || (!ranges && n)) throw Error{IMPG_E_INVALID, "null argument"};
  check_ranges(ranges, n);
  Engine::check_params(*params);
  impg_gpu_params_t p = *params;
  p.store_cigar = 0;  // BED never needs CIGARs (main.rs:7447)
  if (ix->shard || ix->cluster) { sharded_bed_batch(*ix, ranges, n, p, subset_keep, merge_distance, range_names, sink, seconds3); return; }
  IMPG_HIP(hipSetDevice(ix->device));
  EngineLease lease(*ix);
  Engine &E = *lease;
  apply_subset(E, *ix, subset_keep);
  E.ranges_dev.reserve(std::max<size_t>(n * sizeof(impg_gpu_range_t), 256));
  if (n) IMPG_HIP(hipMemcpyAsync(E.ranges_dev.p, ranges, n * sizeof(impg_gpu_range_t), hipMemcpyHostToDevice, E.stream));
  double t_engine = 0, t_merge = 0, t_text = 0;
  for_chunks(E, n, [&](size_t b, size_t e) {
    const auto c0 = std::chrono::steady_clock::now();
    std::vector<std::unique_ptr<LevelBufs>> levels;
    DevBuf self_dev, rows;
    self_dev.pool = &E.level_pool;
    E.run(*ix, E.ranges_dev.as<impg_gpu_range_t>() + b, (uint32_t)(e - b), p, &levels, nullptr, nullptr, nullptr, &self_dev);
    const auto c1 = std::chrono::steady_clock::now();
    const uint32_t n_rows = device_bed_rows(E, *ix, (uint32_t)(e - b), p, merge_distance, levels, self_dev, rows);
    const auto c2 = std::chrono::steady_clock::now();
    const std::vector<std::string> rn = bed_range_names(*ix, ranges, range_names, b, e);
    device_bed_text(E, *ix, rows, n_rows, (uint32_t)(e - b), rn, p.original_sequence_coordinates != 0, sink);
    const auto c3 = std::chrono::steady_clock::now();
    t_engine += std::chrono::duration<double>(c1 - c0).count();
    t_merge += std::chrono::duration<double>(c2 - c1).count();
    t_text += std::chrono::duration<double>(c3 - c2).count();
  });
  if (seconds3) { seconds3[0] = t_engine; seconds3[1] = t_merge; seconds3[2] = t_text; }
}
}  // namespace

int impg_gpu_query_batch_bed(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                             const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names, char **text,
                             size_t *len, double *seconds3) {
  IMPG_TRY
  if (!text || !len) throw Error{IMPG_E_INVALID, "null argument"};
  // the text arrives in pieces; it is collected into one malloc'ed buffer that grows geometrically
  char *buf = nullptr;
  size_t used = 0, cap = 0;
  struct Guard { char *&b; bool keep = false; ~Guard() { if (!keep) free(b); } } guard{buf};
  bed_batch(ix, ranges, n, params, subset_keep, merge_distance, range_names, [&](const char *p, size_t k) {
    if (used + k + 1 > cap) {
      const size_t nc = std::max(cap * 2, used + k + 1 + (1u << 20));
      char *nb = (char *)realloc(buf, nc);
      if (!nb) throw Error{IMPG_E_OOM, "host out of memory"};
      buf = nb; cap = nc;
    }
    memcpy(buf + used, p, k);
    used += k;
  }, seconds3);
  if (!buf) { buf = (char *)malloc(1); if (!buf) throw Error{IMPG_E_OOM, "host out of memory"}; }
  buf[used] = 0;
  guard.keep = true;
  *text = buf;
  *len = used;
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_query_batch_bed_fd(impg_gpu_index_t *ix, const impg_gpu_range_t *ranges, size_t n, const impg_gpu_params_t *params,
                                const uint8_t *subset_keep, int32_t merge_distance, const char *const *range_names, int fd,
                                uint64_t *bytes_written, double *seconds3) {
  IMPG_TRY
  uint64_t total = 0;
  bed_batch(ix, ranges, n, params, subset_keep, merge_distance, range_names, [&](const char *p, size_t k) {
    size_t done = 0;
    while (done < k) {
      const ssize_t w = write(fd, p + done, k - done);
      if (w < 0) {
        if (errno == EINTR) continue;
        throw Error{IMPG_E_IO, std::string("write failed: ") + strerror(errno)};
      }
      done += (size_t)w;
    }
    total += k;
  }, seconds3);
  if (bytes_written) *bytes_written = total;
  return IMPG_OK;
  IMPG_CATCH
}

int impg_gpu_results_paf(const impg_gpu_results_t *res, const impg_gpu_index_t *ix, const char *const *range_names,
                         const impg_gpu_params_t *params, int32_t merge_distance, int format, char **text, size_t *len) {
  IMPG_TRY
  if (!res || !ix || !params || !text || !len) throw Error{IMPG_E_INVALID, "null argument"};
  if (format != IMPG_OUT_PAF && format != IMPG_OUT_BEDPE) throw Error{IMPG_E_INVALID, "unknown output format"};
  std::vector<std::string> parts;
  render_paf(*res, *ix, range_names, *params, merge_distance, format == IMPG_OUT_BEDPE, parts);
  *text = join_parts(parts, len);
  return IMPG_OK;
  IMPG_CATCH
}

// The lookup order's sort checked on its own (diagnostics; tests/test_gpu_parity.py): n pseudo-random keys of `end_bit` bits
// (seed; every 7th key repeats its predecessor's, so equal keys are common) through launch_order_sort on `device`; the result
// must be a permutation of 0 .. n-1 that lists the keys in non-decreasing order, equal keys by ascending index.
int impg_gpu_selftest_order_sort(int device, uint32_t n, unsigned end_bit, uint64_t seed) {
  IMPG_TRY
  if (end_bit < 1 || end_bit > 32) throw Error{IMPG_E_INVALID, "end_bit is 1 .. 32"};
  IMPG_HIP(hipSetDevice(device));
  std::vector<uint32_t> keys(n), perm(n);
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
  const uint32_t mask = end_bit >= 32 ? 0xFFFFFFFFu : ((1u << end_bit) - 1u);
  for (uint32_t i = 0; i < n; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    keys[i] = (i % 7u == 3u && i) ? keys[i - 1] : (uint32_t)(x >> 20);  // (bits above end_bit stay set: they must not matter)
  }
  hipStream_t s;
  IMPG_HIP(hipStreamCreate(&s));
  DevBuf ka, kb, pa, pb, sc;
  const size_t nb = std::max<size_t>((size_t)n * 4, 256);
  ka.reserve(nb); kb.reserve(nb); pa.reserve(nb); pb.reserve(nb); sc.reserve(order_sort_scratch_bytes(n));
  IMPG_HIP(hipMemcpyAsync(ka.p, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  IMPG_HIP(hipMemsetAsync(pa.p, 0xA5, nb, s));
  launch_order_sort(ka.as<uint32_t>(), kb.as<uint32_t>(), pa.as<uint32_t>(), pb.as<uint32_t>(), n, end_bit, sc.p, s);
  IMPG_HIP(hipGetLastError());
  IMPG_HIP(hipMemcpyAsync(perm.data(), pa.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  IMPG_HIP(hipStreamSynchronize(s));
  IMPG_HIP(hipStreamDestroy(s));
#ifdef IMPG_OS_CLOCKS
  {
    unsigned long long c[8];
    order_sort_clocks(c, false);
    fprintf(stderr, "[order sort clocks] n=%u bits=%u tiles x passes=%llu: per tile-pass zero+load %.0f rank %.0f bases %.0f lds %.0f out %.0f cycles\n", n, end_bit, c[7],
            (double)c[0] / c[7], (double)c[1] / c[7], (double)c[2] / c[7], (double)c[3] / c[7], (double)c[4] / c[7]);
    order_sort_clocks(nullptr, true);
  }
#endif
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t i = perm[k];
    if (i >= n || seen[i]) throw Error{IMPG_E_INVALID, "order sort: not a permutation at place " + std::to_string(k)};
    seen[i] = 1;
    if (k) {
      const uint32_t a = keys[perm[k - 1]] & mask, b = keys[i] & mask;
      if (a > b || (a == b && perm[k - 1] > i)) throw Error{IMPG_E_INVALID, "order sort: out of order at place " + std::to_string(k)};
    }
  }
  return IMPG_OK;
  IMPG_CATCH
}

}  // extern "C"
