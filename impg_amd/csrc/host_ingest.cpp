// Host-side ingest: PAF -> records + packed CIGAR ops, range parsers, and the
// synthetic workload generators.  Mirrors the reference's parsers
// (src/paf.rs:118-194, src/impg.rs:2935-2950, src/commands/partition.rs:1752-1789)
// so that the same text yields the same records and sequence ids.
#include <cmath>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <zlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include "impg_internal.hpp"

namespace impg {

uint32_t HostSeqIndex::get_or_insert(const std::string &name, int64_t len) {
  auto it = name_to_id.find(name);
  if (it != name_to_id.end()) return it->second;  // first length seen wins (seqidx.rs:29-31)
  uint32_t id = (uint32_t)names.size();
  name_to_id.emplace(name, id);
  names.push_back(name);
  lens.push_back(len);
  return id;
}

// "[0-9]+[=XIDM]" tokens; any non-digit byte closes an op (impg.rs:2940-2947).
long parse_cigar(const char *s, size_t n, uint32_t *out, size_t cap) {
  size_t k = 0;
  uint32_t len = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned c = (unsigned char)s[i];
    unsigned d = c - '0';
    if (d <= 9) {
      len = len * 10 + d;
      continue;
    }
    uint32_t code;
    switch (c) {
      case '=': code = 0; break;
      case 'X': code = 1; break;
      case 'I': code = 2; break;
      case 'D': code = 3; break;
      case 'M': code = 4; break;
      default: return -1;
    }
    if (k < cap) out[k] = (code << 29) | (len & OP_LEN_MASK);
    k++;
    len = 0;
  }
  return (long)k;
}

namespace {

struct Field {
  const char *p;
  size_t n;
};

// usize::from_str: optional '+', then digits only
bool to_u64(Field f, uint64_t *v) {
  size_t i = (f.n && f.p[0] == '+') ? 1 : 0;
  if (i >= f.n) return false;
  uint64_t x = 0;
  for (; i < f.n; i++) {
    unsigned d = (unsigned char)f.p[i] - '0';
    if (d > 9) return false;
    x = x * 10 + d;
  }
  *v = x;
  return true;
}

struct ChunkOut {
  const char *raw_text = nullptr;          // raw mode: CIGARs stay text, records carry byte offsets from here ...
  uint64_t raw_base = 0;                   // ... plus this
  std::vector<impg_gpu_record_t> records;  // ids are chunk-local
  std::vector<uint32_t> ops;
  std::vector<std::string> names;          // chunk-local id -> name, first-seen order
  std::vector<int64_t> lens;
  std::unordered_map<std::string, uint32_t> local;
  std::string err;
  size_t err_line = 0;
  uint32_t id_of(Field name, int64_t len) {
    std::string s(name.p, name.n);
    auto it = local.find(s);
    if (it != local.end()) return it->second;
    uint32_t id = (uint32_t)names.size();
    local.emplace(s, id);
    names.push_back(std::move(s));
    lens.push_back(len);
    return id;
  }
};

bool parse_line(const char *line, size_t len, ChunkOut &o) {
  Field f[12];
  size_t nf = 0, st = 0;
  const char *cg = nullptr;
  size_t cgn = 0;
  for (size_t i = 0; i <= len; i++) {
    if (i == len || line[i] == '\t') {
      Field cur{line + st, i - st};
      if (nf < 12) f[nf] = cur;
      nf++;
      // the reference scans every field, from the first, for the tag (paf.rs:155-163)
      if (!cg && cur.n >= 5 && memcmp(cur.p, "cg:Z:", 5) == 0) {
        cg = cur.p + 5;
        cgn = cur.n - 5;
      }
      st = i + 1;
    }
  }
  if (nf < 12) { o.err = "Not enough fields in PAF record"; return false; }
  uint64_t qlen, qs, qe, tlen, ts, te;
  if (!to_u64(f[1], &qlen) || !to_u64(f[2], &qs) || !to_u64(f[3], &qe) || !to_u64(f[6], &tlen) ||
      !to_u64(f[7], &ts) || !to_u64(f[8], &te)) { o.err = "Invalid field"; return false; }
  if (f[4].n == 0) { o.err = "Expected '+' or '-' for strand"; return false; }
  char sc = f[4].p[0];
  if (sc != '+' && sc != '-') { o.err = "Invalid strand"; return false; }
  impg_gpu_record_t r;
  r.query_id = o.id_of(f[0], (int64_t)qlen);   // query first, then target (paf.rs:149-150)
  r.target_id = o.id_of(f[5], (int64_t)tlen);
  r.query_start = (int32_t)qs; r.query_end = (int32_t)qe;
  r.target_start = (int32_t)ts; r.target_end = (int32_t)te;
  r.strand = sc == '-';
  r.cigar_off = o.ops.size();
  r.cigar_len = 0;
  if (o.raw_text) {
    if (cgn >= 0xFFFFFFF0ull) { o.err = "CIGAR longer than 2^32 bytes"; return false; }
    r.cigar_off = cg ? (uint64_t)(cg - o.raw_text) + o.raw_base : 0;
    r.cigar_len = (uint32_t)cgn;
  } else if (cg && cgn) {
    size_t base = o.ops.size();
    o.ops.resize(base + cgn);  // upper bound: one op per byte
    long k = parse_cigar(cg, cgn, o.ops.data() + base, cgn);
    if (k < 0) { o.err = "Invalid CIGAR operation"; return false; }
    o.ops.resize(base + (size_t)k);
    r.cigar_len = (uint32_t)k;
  }
  o.records.push_back(r);
  return true;
}

void parse_chunk(const char *text, size_t begin, size_t end, ChunkOut &o) {
  size_t pos = begin, line_no = 0;
  while (pos < end) {
    const char *nl = (const char *)memchr(text + pos, '\n', end - pos);
    size_t eol = nl ? (size_t)(nl - text) : end;
    size_t l = eol - pos;
    if (l && text[pos + l - 1] == '\r') l--;  // BufRead::lines strips "\r\n"
    if (!parse_line(text + pos, l, o)) { o.err_line = line_no; return; }
    line_no++;
    pos = eol + 1;
  }
}

}  // namespace

void parse_paf_text(const char *text, size_t len, ParsedPaf &out, bool raw, uint64_t text_base) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, len / (4u << 20) + 1));
  std::vector<size_t> cut(T + 1, len);
  cut[0] = 0;
  for (size_t t = 1; t < T; t++) {
    size_t p = len / T * t;
    const char *nl = (const char *)memchr(text + p, '\n', len - p);
    cut[t] = nl ? (size_t)(nl - text) + 1 : len;
  }
  for (size_t t = 1; t <= T; t++) cut[t] = std::max(cut[t], cut[t - 1]);
  std::vector<ChunkOut> chunks(T);
  if (raw) for (auto &c : chunks) { c.raw_text = text; c.raw_base = text_base; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++)
    th.emplace_back([&, t]() { parse_chunk(text, cut[t], cut[t + 1], chunks[t]); });
  for (auto &x : th) x.join();
  for (size_t t = 0; t < T; t++)
    if (!chunks[t].err.empty()) throw Error{IMPG_E_INVALID, "Failed to parse PAF: " + chunks[t].err};
  // Merge: sequence ids in first-seen order over the chunks (serial: a few hundred names), then every chunk's
  // records and ops copied to their final place in parallel (the ops are most of the input's bytes: appended one
  // chunk after the other they cost as much as the parse itself).
  std::vector<std::vector<uint32_t>> remap(T);
  std::vector<size_t> rec_at(T + 1, out.records.size()), ops_at(T + 1, out.ops.size());
  for (size_t t = 0; t < T; t++) {
    remap[t].resize(chunks[t].names.size());
    for (size_t i = 0; i < chunks[t].names.size(); i++) remap[t][i] = out.seq.get_or_insert(chunks[t].names[i], chunks[t].lens[i]);
    rec_at[t + 1] = rec_at[t] + chunks[t].records.size();
    ops_at[t + 1] = ops_at[t] + chunks[t].ops.size();
  }
  out.records.resize(rec_at[T]);
  out.ops.resize(ops_at[T]);
  th.clear();
  for (size_t t = 0; t < T; t++)
    th.emplace_back([&, t]() {
      ChunkOut &c = chunks[t];
      if (!c.ops.empty()) memcpy(out.ops.data() + ops_at[t], c.ops.data(), c.ops.size() * 4);
      impg_gpu_record_t *dst = out.records.data() + rec_at[t];
      for (size_t i = 0; i < c.records.size(); i++) {
        impg_gpu_record_t r = c.records[i];
        r.query_id = remap[t][r.query_id];
        r.target_id = remap[t][r.target_id];
        if (!raw) r.cigar_off += ops_at[t];
        dst[i] = r;
      }
      std::vector<uint32_t>().swap(c.ops);
      std::vector<impg_gpu_record_t>().swap(c.records);
    });
  for (auto &x : th) x.join();
}

void parse_paf_files(const std::vector<std::string> &paths, ParsedPaf &out, bool raw) {
  out.file_first.clear();
  out.raw = raw;
  uint64_t text_base = 0;
  for (const auto &path : paths) {
    out.file_first.push_back(out.records.size());
    if (path.size() > 3 && (path.compare(path.size() - 3, 3, ".gz") == 0 ||
                            (path.size() > 4 && path.compare(path.size() - 4, 4, ".bgz") == 0))) {
      // gzip / BGZF (a BGZF file is a series of gzip members: zlib reads it as one stream).  The reference keeps
      // such files compressed and seeks into them per hit (.gzi index, impg.rs:2903-2933); here the CIGARs
      // are tokenised once at ingest, so a sequential decompression is all that is needed.
      gzFile gz = gzopen(path.c_str(), "rb");
      if (!gz) throw Error{IMPG_E_IO, "Failed to open file '" + path + "'"};
      gzbuffer(gz, 1 << 20);
      std::string text;
      std::vector<char> buf(1 << 22);
      for (;;) {
        int n = gzread(gz, buf.data(), (unsigned)buf.size());
        if (n < 0) {
          int err = 0;
          std::string msg = gzerror(gz, &err);
          gzclose(gz);
          throw Error{IMPG_E_IO, "Failed to decompress '" + path + "': " + msg};
        }
        if (n == 0) break;
        text.append(buf.data(), (size_t)n);
      }
      gzclose(gz);
      if (raw) {
        auto keep = std::make_shared<std::string>(std::move(text));
        out.holders.push_back(keep);
        out.texts.push_back({keep->data(), keep->size(), text_base});
        if (!keep->empty()) parse_paf_text(keep->data(), keep->size(), out, true, text_base);
        text_base += keep->size();
      } else if (!text.empty()) parse_paf_text(text.data(), text.size(), out);
      continue;
    }
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error{IMPG_E_IO, "Failed to open file '" + path + "'"};
    struct stat stt;
    fstat(fd, &stt);
    size_t sz = (size_t)stt.st_size;
    if (sz == 0) { close(fd); continue; }
    void *m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) throw Error{IMPG_E_IO, "Failed to map file '" + path + "'"};
    if (raw) {  // the mapping lives until the device has tokenised the CIGARs
      out.holders.push_back(std::shared_ptr<void>(m, [sz](void *q) { munmap(q, sz); }));
      out.texts.push_back({(const char *)m, sz, text_base});
      parse_paf_text((const char *)m, sz, out, true, text_base);
      text_base += sz;
      continue;
    }
    try {
      parse_paf_text((const char *)m, sz, out);
    } catch (...) {
      munmap(m, sz);
      throw;
    }
    munmap(m, sz);
  }
  out.file_first.push_back(out.records.size());
}

// ---------------------------------------------------------------------------
// synthetic workload
// ---------------------------------------------------------------------------
namespace {
struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint64_t below(uint64_t n) { return next() % n; }
};
SplitMix64 rng_for(uint64_t seed, uint64_t i) {
  SplitMix64 a{seed ^ ((i + 1) * 0xD1342543DE82EF95ull)};
  return SplitMix64{a.next()};
}

struct SynthRecord {
  uint32_t query, target, strand;
  int32_t qs, qe, ts, te;
  uint64_t matches, block;
};
// the CIGAR and the coordinates of a record whose sequences, strand, target span and block count are chosen: appends
// 2*n_blocks ops (n_blocks x {'=' run, one edit: X 60 %, I 1-8 20 %, D 1-8 20 %}) whose target deltas sum to `span`
void synth_fill(SplitMix64 &g, SynthRecord &r, int32_t L, int32_t span, uint32_t n_blocks, std::vector<uint32_t> &ops) {
  std::vector<uint32_t> edit(n_blocks);
  int64_t sumX = 0, sumD = 0, sumI = 0;
  for (uint32_t b = 0; b < n_blocks; b++) {
    uint64_t k = g.below(100);
    if (k < 60) { edit[b] = (1u << 29) | 1; sumX += 1; }
    else if (k < 80) { uint32_t l = 1 + (uint32_t)g.below(8); edit[b] = (2u << 29) | l; sumI += l; }
    else { uint32_t l = 1 + (uint32_t)g.below(8); edit[b] = (3u << 29) | l; sumD += l; }
  }
  int64_t E = (int64_t)span - sumX - sumD;  // bases covered by '=' ops
  int64_t rem = E - n_blocks;
  std::vector<int64_t> cut(n_blocks + 1);
  cut[0] = 0;
  for (uint32_t b = 1; b < n_blocks; b++) cut[b] = (int64_t)g.below((uint64_t)rem + 1);
  cut[n_blocks] = rem;
  std::sort(cut.begin() + 1, cut.begin() + n_blocks);
  for (uint32_t b = 0; b < n_blocks; b++) {
    uint32_t l = (uint32_t)(cut[b + 1] - cut[b] + 1);
    ops.push_back(l);  // code 0 '='
    ops.push_back(edit[b]);
  }
  int64_t qspan = E + sumX + sumI;
  r.ts = (int32_t)g.below((uint64_t)(L - span) + 1);
  r.te = r.ts + span;
  r.qs = (int32_t)g.below((uint64_t)(L - qspan) + 1);
  r.qe = r.qs + (int32_t)qspan;
  r.matches = (uint64_t)E;
  r.block = (uint64_t)(E + sumX + sumI + sumD);
}
// generates record i; appends 2*n_blocks ops
SynthRecord synth_record(uint64_t seed, uint64_t i, uint32_t n_seq, int32_t L, int32_t span,
                         uint32_t n_blocks, std::vector<uint32_t> &ops) {
  SplitMix64 g = rng_for(seed, i);
  SynthRecord r;
  r.target = (uint32_t)g.below(n_seq);
  r.query = (uint32_t)g.below(n_seq - 1);
  if (r.query >= r.target) r.query++;
  r.strand = (uint32_t)g.below(2);
  synth_fill(g, r, L, span, n_blocks, ops);
  return r;
}
// The NON-uniform workload (`bench.py --workload skewed`; SURVEY section 7 "Hard parts": real cohorts have alignments of 10^5-10^6
// ops and repeat-driven hit skew): target spans log-normal (median 8 kb, sigma 1.25 in ln units) clamped to [1 kb, seq_len - 16 kb]
// -- 20 ops to 10^5 ops a CIGAR, one block of {'=' run, edit} per 100 target bases, so records of more than 8 tiles carry
// external checkpoints -- and the first max(1, n_seq / 100) sequences are HOT: a record's target is one of them with
// probability 0.3, and so is its query (1 % of the sequences hold ~30 % of a bidirectional index's entries: windows of
// hundreds to thousands of entries, the wave-per-range emit and the listed wide windows).
SynthRecord synth_record_skewed(uint64_t seed, uint64_t i, uint32_t n_seq, int32_t L, std::vector<uint32_t> &ops) {
  SplitMix64 g = rng_for(seed ^ 0x5CE3ED5EEDull, i);
  SynthRecord r;
  const uint32_t n_hot = std::max<uint32_t>(1, n_seq / 100);
  auto pick = [&]() -> uint32_t { return g.below(10) < 3 ? (uint32_t)g.below(n_hot) : (uint32_t)g.below(n_seq); };
  r.target = pick();
  do { r.query = pick(); } while (r.query == r.target);
  r.strand = (uint32_t)g.below(2);
  // Box-Muller on two 53-bit uniforms
  const double u1 = ((double)(g.next() >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (double)(g.next() >> 11) * (1.0 / 9007199254740992.0);
  const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  double len = 8000.0 * std::exp(1.25 * z);
  const double lo = 1000.0, hi = (double)L - 16384.0;
  if (len < lo) len = lo;
  if (len > hi) len = hi;
  const int32_t span = (int32_t)len;
  const uint32_t n_blocks = std::max<uint32_t>(10u, (uint32_t)(span / 100));
  synth_fill(g, r, L, span, n_blocks, ops);
  return r;
}
}  // namespace
}  // namespace impg

using namespace impg;

extern "C" {

int impg_synth_seq_name(uint32_t id, char *out, size_t cap) {
  return snprintf(out, cap, "g%03u#%u#chr1", id / 4, id % 4 + 1);
}

int impg_synth_paf(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len, int32_t target_span,
                   uint32_t n_blocks, impg_gpu_record_t *records, uint32_t *ops, size_t ops_cap,
                   size_t *n_ops_out) {
  if (n_seq < 2 || target_span <= 0 || seq_len < target_span + 8 * (int32_t)n_blocks ||
      (int64_t)target_span < 10 * (int64_t)n_blocks) {
    set_error("impg_synth_paf: bad shape");
    return IMPG_E_INVALID;
  }
  size_t total = n_records * 2 * (size_t)n_blocks;
  if (n_ops_out) *n_ops_out = total;
  if (!ops || !records) return IMPG_OK;
  if (ops_cap < total) { set_error("impg_synth_paf: ops buffer too small"); return IMPG_E_INVALID; }
  unsigned hw = std::thread::hardware_concurrency();
  size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_records / 4096 + 1));
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++)
    th.emplace_back([=]() {
      std::vector<uint32_t> tmp;
      for (size_t i = t; i < n_records; i += T) {
        tmp.clear();
        SynthRecord s = synth_record(seed, i, n_seq, seq_len, target_span, n_blocks, tmp);
        memcpy(ops + i * 2 * (size_t)n_blocks, tmp.data(), tmp.size() * 4);
        impg_gpu_record_t r;
        r.query_id = s.query; r.target_id = s.target;
        r.query_start = s.qs; r.query_end = s.qe; r.target_start = s.ts; r.target_end = s.te;
        r.cigar_off = i * 2 * (uint64_t)n_blocks; r.cigar_len = 2 * n_blocks; r.strand = s.strand;
        records[i] = r;
      }
    });
  for (auto &x : th) x.join();
  return IMPG_OK;
}

int impg_synth_paf_text(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len, int32_t target_span,
                        uint32_t n_blocks, const char *path) {
  if (n_seq < 2 || target_span <= 0 || seq_len < target_span + 8 * (int32_t)n_blocks ||
      (int64_t)target_span < 10 * (int64_t)n_blocks) {
    set_error("impg_synth_paf_text: bad shape");
    return IMPG_E_INVALID;
  }
  FILE *fp = fopen(path, "wb");
  if (!fp) { set_error(std::string("cannot create ") + path); return IMPG_E_IO; }
  static const char OPC[] = "=XIDM";
  auto put_u = [](std::string &o, unsigned long long v) {
    char b[24];
    int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
  };
  // records are formatted in parallel, block by block, and written in order
  const size_t BLOCK = 4096;
  unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, 32));
  bool ok = true;
  for (size_t base = 0; base < n_records && ok; base += BLOCK * T) {
    std::vector<std::string> bufs(T);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        size_t lo = base + t * BLOCK, hi = std::min(n_records, lo + BLOCK);
        std::string &line = bufs[t];
        std::vector<uint32_t> tmp;
        char qn[32], tn[32];
        for (size_t i = lo; i < hi; i++) {
          tmp.clear();
          SynthRecord s = synth_record(seed, i, n_seq, seq_len, target_span, n_blocks, tmp);
          impg_synth_seq_name(s.query, qn, sizeof qn);
          impg_synth_seq_name(s.target, tn, sizeof tn);
          line += qn; line.push_back('\t'); put_u(line, (unsigned)seq_len); line.push_back('\t');
          put_u(line, (unsigned)s.qs); line.push_back('\t'); put_u(line, (unsigned)s.qe); line.push_back('\t');
          line.push_back(s.strand ? '-' : '+'); line.push_back('\t');
          line += tn; line.push_back('\t'); put_u(line, (unsigned)seq_len); line.push_back('\t');
          put_u(line, (unsigned)s.ts); line.push_back('\t'); put_u(line, (unsigned)s.te); line.push_back('\t');
          put_u(line, s.matches); line.push_back('\t'); put_u(line, s.block); line += "\t255\tcg:Z:";
          for (uint32_t v : tmp) { put_u(line, v & OP_LEN_MASK); line.push_back(OPC[v >> 29]); }
          line.push_back('\n');
        }
      });
    for (auto &x : th) x.join();
    for (size_t t = 0; t < T && ok; t++)
      if (!bufs[t].empty() && fwrite(bufs[t].data(), 1, bufs[t].size(), fp) != bufs[t].size()) ok = false;
  }
  if (fclose(fp) != 0) ok = false;
  if (!ok) { set_error("write failed"); return IMPG_E_IO; }
  return IMPG_OK;
}

// the skewed workload as PAF text (synth_record_skewed): returns the number of ops written in *n_ops_out
int impg_synth_skewed_paf_text(uint64_t seed, size_t n_records, uint32_t n_seq, int32_t seq_len, const char *path, uint64_t *n_ops_out) {
  if (n_seq < 2 || seq_len < 65536) { set_error("impg_synth_skewed_paf_text: bad shape"); return IMPG_E_INVALID; }
  FILE *fp = fopen(path, "wb");
  if (!fp) { set_error(std::string("cannot create ") + path); return IMPG_E_IO; }
  static const char OPC[] = "=XIDM";
  auto put_u = [](std::string &o, unsigned long long v) {
    char b[24];
    int n = 0;
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) o.push_back(b[--n]);
  };
  const size_t BLOCK = 1024;
  unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, 32));
  bool ok = true;
  std::atomic<uint64_t> n_ops{0};
  for (size_t base = 0; base < n_records && ok; base += BLOCK * T) {
    std::vector<std::string> bufs(T);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        size_t lo = base + t * BLOCK, hi = std::min(n_records, lo + BLOCK);
        std::string &line = bufs[t];
        std::vector<uint32_t> tmp;
        char qn[32], tn[32];
        uint64_t mine = 0;
        for (size_t i = lo; i < hi; i++) {
          tmp.clear();
          SynthRecord s = synth_record_skewed(seed, i, n_seq, seq_len, tmp);
          mine += tmp.size();
          impg_synth_seq_name(s.query, qn, sizeof qn);
          impg_synth_seq_name(s.target, tn, sizeof tn);
          line += qn; line.push_back('\t'); put_u(line, (unsigned)seq_len); line.push_back('\t');
          put_u(line, (unsigned)s.qs); line.push_back('\t'); put_u(line, (unsigned)s.qe); line.push_back('\t');
          line.push_back(s.strand ? '-' : '+'); line.push_back('\t');
          line += tn; line.push_back('\t'); put_u(line, (unsigned)seq_len); line.push_back('\t');
          put_u(line, (unsigned)s.ts); line.push_back('\t'); put_u(line, (unsigned)s.te); line.push_back('\t');
          put_u(line, s.matches); line.push_back('\t'); put_u(line, s.block); line += "\t255\tcg:Z:";
          for (uint32_t v : tmp) { put_u(line, v & OP_LEN_MASK); line.push_back(OPC[v >> 29]); }
          line.push_back('\n');
        }
        n_ops += mine;
      });
    for (auto &x : th) x.join();
    for (size_t t = 0; t < T && ok; t++)
      if (!bufs[t].empty() && fwrite(bufs[t].data(), 1, bufs[t].size(), fp) != bufs[t].size()) ok = false;
  }
  if (fclose(fp) != 0) ok = false;
  if (!ok) { set_error("write failed"); return IMPG_E_IO; }
  if (n_ops_out) *n_ops_out = n_ops.load();
  return IMPG_OK;
}

int impg_synth_bed(uint64_t seed, size_t n, uint32_t n_seq, int32_t seq_len, int32_t range_len,
                   impg_gpu_range_t *out) {
  if (range_len <= 0 || range_len > seq_len || n_seq == 0) { set_error("impg_synth_bed: bad shape"); return IMPG_E_INVALID; }
  for (size_t i = 0; i < n; i++) {
    SplitMix64 g = rng_for(seed, i);
    out[i].target_id = (uint32_t)g.below(n_seq);
    out[i].start = (int32_t)g.below((uint64_t)(seq_len - range_len) + 1);
    out[i].end = out[i].start + range_len;
  }
  return IMPG_OK;
}

long impg_gpu_parse_cigar(const char *cigar, size_t len, uint32_t *ops_out, size_t cap) {
  return parse_cigar(cigar, len, ops_out, cap);
}

// parse_target_range: split on the LAST ':', then "start-end" (partition.rs:1752-1789)
int impg_gpu_parse_target_range(const char *s, char *name_out, size_t name_cap, int32_t *start, int32_t *end) {
  const char *colon = strrchr(s, ':');
  if (!colon) { set_error("Target range format should be `seq_name:start-end`"); return IMPG_E_INVALID; }
  const char *r = colon + 1;
  const char *dash = strchr(r, '-');
  if (!dash || strchr(dash + 1, '-')) { set_error("Range format should be `start-end`"); return IMPG_E_INVALID; }
  auto to_i32 = [](const char *p, size_t n, int32_t *v) {
    size_t i = 0;
    bool neg = false;
    if (n && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
    if (i >= n) return false;
    int64_t x = 0;
    for (; i < n; i++) {
      unsigned d = (unsigned char)p[i] - '0';
      if (d > 9) return false;
      x = x * 10 + d;
      if (x > 2147483648ll) return false;
    }
    if (neg) x = -x;
    if (x > 2147483647ll) return false;
    *v = (int32_t)x;
    return true;
  };
  if (!to_i32(r, (size_t)(dash - r), start)) { set_error("Invalid start value"); return IMPG_E_INVALID; }
  if (!to_i32(dash + 1, strlen(dash + 1), end)) { set_error("Invalid end value"); return IMPG_E_INVALID; }
  if (*start >= *end) { set_error("Start value must be less than end value"); return IMPG_E_INVALID; }
  size_t nl = (size_t)(colon - s);
  if (nl + 1 > name_cap) { set_error("name buffer too small"); return IMPG_E_INVALID; }
  memcpy(name_out, s, nl);
  name_out[nl] = 0;
  return IMPG_OK;
}

}  // extern "C"
