// Reading the reference's own index file, "IMPGIDX2" (writer src/impg.rs:1655-1721, reader :1787-1850 and
// :1724-1767), into the HBM layout.  The file is a 16-byte header (magic, u64 LE offset of the forest map)
// followed by bincode-2 `config::standard()` values: the SequenceIndex (seqidx.rs:5-10), one
// (u32 target_id, Vec<SerializableInterval>) per tree (impg.rs:235-240, :164-174) and the ForestMap
// (forest_map.rs:6-9).  bincode's standard encoding: little-endian varints (< 251 one byte, 0xFB + u16,
// 0xFC + u32, 0xFD + u64), zig-zag for signed, usize as u64, length-prefixed strings / sequences / maps.
// The file holds no alignments, only (alignment_file_index, byte offset, byte count) of every CIGAR: the
// alignment files are named again by the caller, in the order the index was built with (impg.rs:1789, :1844),
// and their CIGARs are tokenised here once, as at any other ingest.  The reference rebuilds each tree from the
// intervals in the order the file lists them (BasicCOITree::new, :1745-1755): that order becomes the tie order
// among equal starts, so the entries are handed to the index builder in exactly that order (EntryPlan).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <thread>

#include "engine.hpp"

namespace impg {
namespace {

struct Dec {
  const unsigned char *p, *e;
  const std::string &what;
  [[noreturn]] void fail() const { throw Error{IMPG_E_INVALID, what + ": damaged or truncated IMPG index"}; }
  uint64_t le(int n) {
    if (e - p < n) fail();
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i);
    p += n;
    return v;
  }
  uint64_t u() {
    if (p >= e) fail();
    const unsigned char t = *p++;
    if (t < 251) return t;
    if (t == 0xFB) return le(2);
    if (t == 0xFC) return le(4);
    if (t == 0xFD) return le(8);
    fail();
  }
  uint32_t u32() {
    const uint64_t v = u();
    if (v > 0xFFFFFFFFull) fail();
    return (uint32_t)v;
  }
  int32_t i32() {
    const uint64_t z = u();
    const int64_t v = (int64_t)(z >> 1) ^ -(int64_t)(z & 1);
    if (v > 2147483647ll || v < -2147483648ll) fail();
    return (int32_t)v;
  }
  std::string str() {
    const uint64_t n = u();
    if ((uint64_t)(e - p) < n) fail();
    std::string x((const char *)p, (size_t)n);
    p += n;
    return x;
  }
};

constexpr uint64_t STRAND_BIT = 0x8000000000000000ull, REVERSED_BIT = 0x4000000000000000ull;  // impg.rs:178-179

struct FileEntry {
  uint32_t target, query_id, file;
  int32_t ts, te, qs, qe;  // the metadata's axes (swapped for a reversed entry)
  uint64_t sado, bytes;
};

}  // namespace

std::unique_ptr<impg_gpu_index> load_impg_file(const char *path, const char *const *alignment_files, int n_files,
                                               int order_policy, int device) {
  const std::string what(path);
  std::string data;
  {
    FILE *f = fopen(path, "rb");
    if (!f) throw Error{IMPG_E_IO, "cannot open " + what};
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
    fclose(f);
  }
  if (data.size() < 16) throw Error{IMPG_E_INVALID, what + " is not an IMPG index"};
  const bool v2 = memcmp(data.data(), "IMPGIDX2", 8) == 0, v1 = memcmp(data.data(), "IMPGIDX1", 8) == 0;
  if (!v1 && !v2) throw Error{IMPG_E_INVALID, what + ": invalid magic bytes - not a valid IMPG index file"};  // impg.rs:1801-1808
  uint64_t fmo = 0;
  for (int i = 0; i < 8; i++) fmo |= (uint64_t)(unsigned char)data[8 + i] << (8 * i);
  if (fmo < 16 || fmo > data.size()) throw Error{IMPG_E_INVALID, what + ": forest map offset outside the file"};
  const unsigned char *base = (const unsigned char *)data.data(), *end = base + data.size();
  // ---- SequenceIndex -------------------------------------------------------------
  HostSeqIndex seq;
  {
    Dec d{base + 16, end, what};
    std::vector<std::pair<std::string, uint32_t>> n2i;
    uint64_t n = d.u();
    for (uint64_t k = 0; k < n; k++) { std::string nm = d.str(); n2i.push_back({std::move(nm), d.u32()}); }
    std::map<uint32_t, std::string> i2n;
    n = d.u();
    for (uint64_t k = 0; k < n; k++) { const uint32_t id = d.u32(); i2n[id] = d.str(); }
    std::map<uint32_t, uint64_t> i2l;
    n = d.u();
    for (uint64_t k = 0; k < n; k++) { const uint32_t id = d.u32(); i2l[id] = d.u(); }
    const uint32_t next_id = d.u32();
    if (next_id > (1u << 30)) throw Error{IMPG_E_INVALID, what + ": unreasonable sequence count"};
    seq.names.assign(next_id, std::string());
    seq.lens.assign(next_id, 0);
    for (auto &kv : i2n) { if (kv.first >= next_id) d.fail(); seq.names[kv.first] = kv.second; }
    for (auto &kv : i2l) { if (kv.first >= next_id) d.fail(); seq.lens[kv.first] = (int64_t)kv.second; }
    for (auto &kv : n2i) { if (kv.second >= next_id) d.fail(); seq.name_to_id.emplace(kv.first, kv.second); }
  }
  const uint32_t n_seq = (uint32_t)seq.lens.size();
  // ---- ForestMap, then every tree it names ----------------------------------------------
  std::vector<std::pair<uint32_t, uint64_t>> forest;
  {
    Dec d{base + fmo, end, what};
    const uint64_t n = d.u();
    for (uint64_t k = 0; k < n; k++) { const uint32_t t = d.u32(); forest.push_back({t, d.u()}); }
  }
  std::vector<std::vector<FileEntry>> per_target(n_seq);
  for (auto &kv : forest) {
    if (kv.first >= n_seq || kv.second < 16 || kv.second >= data.size()) throw Error{IMPG_E_INVALID, what + ": a tree lies outside the file"};
    Dec d{base + kv.second, end, what};
    if (d.u32() != kv.first) throw Error{IMPG_E_INVALID, what + ": tree mismatch"};  // impg.rs:1737-1739
    const uint64_t cnt = d.u();
    if (cnt > (uint64_t)(end - d.p)) d.fail();
    auto &v = per_target[kv.first];
    v.reserve((size_t)cnt);
    for (uint64_t k = 0; k < cnt; k++) {
      FileEntry x;
      x.target = kv.first;
      (void)d.i32(); (void)d.i32();  // first / last: the metadata's target_start / target_end again
      x.query_id = d.u32(); x.ts = d.i32(); x.te = d.i32(); x.qs = d.i32(); x.qe = d.i32();
      x.file = d.u32(); x.sado = d.u(); x.bytes = d.u();
      if (x.query_id >= n_seq || x.file >= (uint32_t)n_files)
        throw Error{IMPG_E_INVALID, what + ": an interval names a sequence or an alignment file that does not exist"};
      v.push_back(x);
    }
  }
  // ---- records: one per forward entry (file, offset); a reversed entry shares its forward partner's -----------
  struct Key { uint32_t file; uint64_t off; bool operator<(const Key &o) const { return file != o.file ? file < o.file : off < o.off; } };
  std::map<Key, uint32_t> rec_of;
  std::vector<impg_gpu_record_t> records;
  std::vector<std::pair<Key, uint64_t>> src;  // where each record's CIGAR text lies
  auto add = [&](const FileEntry &x) { rec_of.emplace(Key{x.file, x.sado & ~(STRAND_BIT | REVERSED_BIT)}, 0u); };
  for (auto &v : per_target) for (auto &x : v) add(x);
  // records in (file, offset) order = the alignment files' own order: the per-file split MultiImpg ties need
  {
    uint32_t id = 0;
    for (auto &kv : rec_of) kv.second = id++;
    records.assign(rec_of.size(), impg_gpu_record_t{});
    src.assign(rec_of.size(), {Key{0, 0}, 0});
  }
  std::vector<uint8_t> have_fwd(records.size(), 0);
  for (auto &v : per_target)
    for (auto &x : v) {
      const Key k{x.file, x.sado & ~(STRAND_BIT | REVERSED_BIT)};
      const uint32_t id = rec_of[k];
      const bool rev = (x.sado & REVERSED_BIT) != 0;
      if (rev && have_fwd[id]) continue;
      impg_gpu_record_t &r = records[id];
      r.query_id = rev ? x.target : x.query_id;
      r.target_id = rev ? x.query_id : x.target;
      r.query_start = rev ? x.ts : x.qs; r.query_end = rev ? x.te : x.qe;
      r.target_start = rev ? x.qs : x.ts; r.target_end = rev ? x.qe : x.te;
      r.strand = (x.sado & STRAND_BIT) ? 1u : 0u;
      src[id] = {k, x.bytes};
      if (!rev) have_fwd[id] = 1;
    }
  std::vector<uint64_t> file_first(1, 0);
  {
    uint32_t id = 0, f = 0;
    for (auto &kv : rec_of) {
      while (f < kv.first.file) { file_first.push_back(id); f++; }
      id++;
    }
    while ((int)file_first.size() < n_files) file_first.push_back(id);
    file_first.push_back(id);
  }
  // ---- CIGARs from the alignment files (pread at the stored offsets; plain-text PAF) -----------------------------
  std::vector<int> fds(n_files, -1);
  struct Closer { std::vector<int> &f; ~Closer() { for (int x : f) if (x >= 0) close(x); } } closer{fds};
  for (int k = 0; k < n_files; k++) {
    const std::string ap(alignment_files[k]);
    if (ap.size() > 3 && (ap.compare(ap.size() - 3, 3, ".gz") == 0 || ap.compare(ap.size() - 4, 4, ".bgz") == 0))
      throw Error{IMPG_E_UNSUPPORTED, ap + ": the offsets of an index over compressed PAF are BGZF virtual offsets; give the plain-text PAF"};
    fds[k] = open(alignment_files[k], O_RDONLY);
    if (fds[k] < 0) throw Error{IMPG_E_IO, "cannot open " + ap};
  }
  const size_t R = records.size();
  std::vector<std::vector<uint32_t>> rec_ops(R);
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, R / 256 + 1));
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++)
    th.emplace_back([&] {
      std::vector<char> buf;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= R) break;
        const uint64_t n = src[i].second;
        if (n == 0) continue;  // a record without cg:Z: an error only if a query reaches it (impg.rs:506-511)
        if (n > (1ull << 32)) { bad = 1; continue; }
        buf.resize((size_t)n);
        size_t got = 0;
        while (got < n) {
          const ssize_t r = pread(fds[src[i].first.file], buf.data() + got, (size_t)n - got, (off_t)(src[i].first.off + got));
          if (r <= 0) break;
          got += (size_t)r;
        }
        if (got != n) { bad = 2; continue; }
        rec_ops[i].resize((size_t)n);
        const long k = parse_cigar(buf.data(), (size_t)n, rec_ops[i].data(), rec_ops[i].size());
        if (k < 0) { bad = 3; continue; }
        rec_ops[i].resize((size_t)k);
      }
    });
  for (auto &x : th) x.join();
  if (bad == 2) throw Error{IMPG_E_IO, what + ": a CIGAR lies outside its alignment file (was the PAF changed after indexing?)"};
  if (bad) throw Error{IMPG_E_INVALID, what + ": invalid CIGAR text at a stored offset (was the PAF changed after indexing?)"};
  std::vector<uint32_t> ops;
  {
    size_t total = 0;
    for (auto &v : rec_ops) total += v.size();
    ops.reserve(total);
    for (size_t i = 0; i < R; i++) {
      records[i].cigar_off = ops.size();
      records[i].cigar_len = (uint32_t)rec_ops[i].size();
      ops.insert(ops.end(), rec_ops[i].begin(), rec_ops[i].end());
      std::vector<uint32_t>().swap(rec_ops[i]);
    }
  }
  // ---- entries in the file's order ------------------------------------------------------------------------------
  EntryPlan plan;
  plan.per_target.resize(n_seq);
  for (uint32_t t = 0; t < n_seq; t++)
    for (auto &x : per_target[t]) {
      const uint32_t id = rec_of[Key{x.file, x.sado & ~(STRAND_BIT | REVERSED_BIT)}];
      plan.per_target[t].push_back(((uint64_t)id << 1) | ((x.sado & REVERSED_BIT) ? 1u : 0u));
    }
  require_device(device);
  auto ix = std::make_unique<impg_gpu_index>();
  ix->device = device;
  ix->seq = seq;
  if (n_files > 1) ix->file_first = file_first;
  std::vector<int64_t> lens = seq.lens;
  build_index(*ix, records.data(), R, ops.data(), ops.size(), lens.data(), n_seq, /*bidirectional (the plan decides)*/ true,
              order_policy, 0, 1, nullptr, nullptr, &plan);
  { EngineLease warm(*ix); }
  return ix;
}

}  // namespace impg

extern "C" int impg_gpu_index_load_impg(const char *impg_path, const char *const *alignment_files, int n_files, int order_policy,
                                        int device, impg_gpu_index_t **out) {
  try {
    if (!impg_path || !out || n_files <= 0 || !alignment_files) throw impg::Error{IMPG_E_INVALID, "bad arguments"};
    *out = impg::load_impg_file(impg_path, alignment_files, n_files, order_policy, device).release();
    return IMPG_OK;
  } catch (const impg::Error &e) {
    impg::set_error(e.msg);
    return e.code;
  } catch (const std::bad_alloc &) {
    impg::set_error("host out of memory");
    return IMPG_E_OOM;
  } catch (const std::exception &e) {
    impg::set_error(std::string("internal: ") + e.what());
    return IMPG_E_INVALID;
  }
}
