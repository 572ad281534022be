// `impg query -o paf | bedpe` text from a batch's results (host side, threaded over
// query ranges).  What the reference does per range after perform_query
// (src/main.rs:7472-7496): drop the first row (the input range), merge neighbouring
// rows of the same (query, target, strand) with their CIGARs
// (merge_adjusted_intervals, main.rs:12563-12845: exact contiguity, identical
// overlap, or a gap of at most -d bases on both axes), then print PAF
// (main.rs:11989-12103) or BEDPE (main.rs:11894-11987) with gi:f / bi:f computed in
// f32 and printed with six decimals, trailing zeros trimmed.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "impg_internal.hpp"

namespace impg {
namespace {

using Ops = std::vector<uint32_t>;

inline uint32_t op_code(uint32_t v) { return v >> 29; }
inline int32_t op_len(uint32_t v) { return (int32_t)(v & OP_LEN_MASK); }
inline uint32_t op_make(int32_t len, uint32_t code) { return (code << 29) | (uint32_t)len; }  // CigarOp::new (impg.rs:90-92)
inline int32_t q_span(uint32_t v) { return op_code(v) == 3u ? 0 : op_len(v); }              // |query_delta| (impg.rs:123-135)
inline int32_t t_span(uint32_t v) { return op_code(v) == 2u ? 0 : op_len(v); }              // target_delta (impg.rs:115-121)
constexpr char OP_CHARS[5] = {'=', 'X', 'I', 'D', 'M'};

// (len as f32 * (num as f32 / den as f32)) as i32 -- the reference scales partial ops in f32
// (main.rs:13075-13077, :13155-13164); every step is rounded to f32 exactly as there.
inline float ratio_f32(int32_t num, int32_t den) {
  volatile float a = (float)num, b = (float)den;
  volatile float r = a / b;
  return r;
}
inline int32_t scaled_len(int32_t len, float ratio) {
  volatile float l = (float)len;
  volatile float x = l * ratio;
  return (int32_t)x;
}

// append `op`, fusing it with the last one if it has the same code (merge_consecutive_cigar_ops, main.rs:13014-13035)
inline void push_fused(Ops &c, uint32_t op) {
  if (!c.empty() && op_code(c.back()) == op_code(op)) c.back() = op_make(op_len(c.back()) + op_len(op), op_code(op));
  else c.push_back(op);
}
void fuse_runs(Ops &c) {
  if (c.size() < 2) return;
  Ops out;
  out.reserve(c.size());
  for (uint32_t op : c) push_fused(out, op);
  c.swap(out);
}

// The ops of `c` that cover its first (from_front) or last `qlen` query bases; a straddling op is
// shortened in proportion (extract_cigar_prefix / extract_cigar_suffix, main.rs:13054-13125).
Ops query_window(const Ops &c, int32_t qlen, bool from_front) {
  Ops out;
  int32_t left = qlen;
  const size_t n = c.size();
  for (size_t k = 0; k < n && left > 0; k++) {
    const uint32_t op = c[from_front ? k : n - 1 - k];
    const int32_t qd = q_span(op);
    if (qd <= left) {
      out.push_back(op);
      left -= qd;
    } else if (qd > 0) {
      out.push_back(op_make(scaled_len(op_len(op), ratio_f32(left, qd)), op_code(op)));
      left = 0;
    }
  }
  if (!from_front) std::reverse(out.begin(), out.end());
  return out;
}

// `c` without the part that covers its first qlen query / tlen target bases (trim_cigar_prefix, main.rs:13127-13180)
Ops drop_front(const Ops &c, int32_t qlen, int32_t tlen) {
  Ops out;
  int32_t qc = 0, tc = 0;
  size_t rest = 0;
  for (size_t k = 0; k < c.size(); k++) {
    const uint32_t op = c[k];
    const int32_t qd = q_span(op), td = t_span(op);
    if (qc + qd > qlen || tc + td > tlen) {
      const int32_t qr = qlen - qc, tr = tlen - tc;
      float skip = 0.0f;
      if (qd > 0 && td > 0) {
        const float a = ratio_f32(qr, qd), b = ratio_f32(tr, td);
        skip = a < b ? a : b;
      } else if (qd > 0) skip = ratio_f32(qr, qd);
      else if (td > 0) skip = ratio_f32(tr, td);
      const int32_t cut = scaled_len(op_len(op), skip);
      if (cut < op_len(op)) out.push_back(op_make(op_len(op) - cut, op_code(op)));
      rest = k + 1;
      break;
    }
    qc += qd;
    tc += td;
    if (qc >= qlen && tc >= tlen) {
      rest = k + 1;
      break;
    }
  }
  out.insert(out.end(), c.begin() + (long)rest, c.end());
  return out;
}

struct Row {
  impg_gpu_interval_t iv;
  Ops cigar;
};
inline bool row_fwd(const Row &r) { return r.iv.q_first <= r.iv.q_last; }

// current <- current (+) next, next lying after current along the walk (before it on the query axis
// for a reverse-strand pair, where the reference prepends; main.rs:12663-12673)
void join(Row &cur, const Row &next, bool forward, const Ops &middle, const Ops &tail) {
  if (forward) {
    cur.iv.q_last = next.iv.q_last;
    cur.iv.t_last = next.iv.t_last;
    cur.cigar.insert(cur.cigar.end(), middle.begin(), middle.end());
    cur.cigar.insert(cur.cigar.end(), tail.begin(), tail.end());
  } else {
    cur.iv.q_first = next.iv.q_first;
    cur.iv.t_first = next.iv.t_first;
    Ops c;
    c.reserve(tail.size() + middle.size() + cur.cigar.size());
    c.insert(c.end(), tail.begin(), tail.end());
    c.insert(c.end(), middle.begin(), middle.end());
    c.insert(c.end(), cur.cigar.begin(), cur.cigar.end());
    cur.cigar.swap(c);
  }
}

void merge_rows(std::vector<Row> &rows, int32_t d) {
  if (rows.size() < 2 || d < 0) return;
  std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) {  // main.rs:12566-12582 (strict '<' for the strand)
    const bool af = a.iv.q_first < a.iv.q_last, bf = b.iv.q_first < b.iv.q_last;
    return std::make_tuple(a.iv.query_id, af, af ? a.iv.q_first : a.iv.q_last, a.iv.target_id, a.iv.t_first) <
           std::make_tuple(b.iv.query_id, bf, bf ? b.iv.q_first : b.iv.q_last, b.iv.target_id, b.iv.t_first);
  });
  std::vector<Row> out;
  out.reserve(rows.size());
  Row cur = std::move(rows[0]);
  const Ops none;
  for (size_t i = 1; i < rows.size(); i++) {
    Row &nx = rows[i];
    const bool f = row_fwd(cur);
    if (cur.iv.t_first > cur.iv.t_last || nx.iv.t_first > nx.iv.t_last)
      throw Error{IMPG_E_INVALID, "Target intervals should always be in forward!"};  // the reference panics (main.rs:12606)
    bool merged = false;
    if (cur.iv.query_id == nx.iv.query_id && cur.iv.target_id == nx.iv.target_id && f == row_fwd(nx)) {
      // the reference's "gaps" (main.rs:12760-12770); its overlap tests (main.rs:12621-12636) in the same terms:
      //   forward: query current.last > next.first  <=> qd < 0      target current.last > next.first  <=> td < 0
      //   reverse: query current.first > next.last  <=> qd > 0 (!)  target current.first < next.last  <=> td < 0
      const int32_t qd = f ? nx.iv.q_first - cur.iv.q_last : cur.iv.q_first - nx.iv.q_last;
      const int32_t td = f ? nx.iv.t_first - cur.iv.t_last : cur.iv.t_first - nx.iv.t_last;
      const bool q_over = f ? qd < 0 : qd > 0;
      const bool t_over = td < 0;
      if (qd == 0 && td == 0) {
        join(cur, nx, f, none, nx.cigar);
        fuse_runs(cur.cigar);
        merged = true;
      } else if (q_over && t_over) {
        // overlap lengths as the reference computes them (main.rs:12680-12692): positive only in its reverse-strand form
        const int32_t qo = f ? nx.iv.q_first - cur.iv.q_last : nx.iv.q_last - cur.iv.q_first;
        const int32_t to = f ? nx.iv.t_first - cur.iv.t_last : cur.iv.t_first - nx.iv.t_last;
        if (qo > 0 && to > 0 && query_window(cur.cigar, qo, false) == query_window(nx.cigar, qo, true)) {
          join(cur, nx, f, none, drop_front(nx.cigar, qo, to));  // (no run fusing on this arm, main.rs:12733-12750)
          merged = true;
        }
      }
      if (!merged && !q_over && !t_over && qd >= 0 && td >= 0 && (qd > 0 || td > 0) && qd <= d && td <= d) {
        Ops gap;
        if (qd > 0) gap.push_back(op_make(qd, 2u));
        if (td > 0) gap.push_back(op_make(td, 3u));
        join(cur, nx, f, gap, nx.cigar);
        fuse_runs(cur.cigar);
        merged = true;
      }
    }
    if (!merged) {
      out.push_back(std::move(cur));
      cur = std::move(nx);
    }
  }
  out.push_back(std::move(cur));
  rows.swap(out);
}

// format!("{x:.6}") then trim_end_matches('0'), trim_end_matches('.')
void put_f32(std::string &s, float x) {
  if (std::isnan(x)) { s += "NaN"; return; }
  if (std::isinf(x)) { s += x < 0 ? "-inf" : "inf"; return; }
  char b[64];
  int n = snprintf(b, sizeof b, "%.6f", (double)x);
  while (n > 0 && b[n - 1] == '0') n--;
  while (n > 0 && b[n - 1] == '.') n--;
  s.append(b, (size_t)n);
}
void put_name(std::string &s, const impg_gpu_index &ix, uint32_t id) {
  if (id < ix.seq.names.size()) s += ix.seq.names[id];
  else s += std::to_string(id);
}
void put_u(std::string &s, unsigned long long v) {
  char b[32];
  int n = snprintf(b, sizeof b, "%llu", v);
  s.append(b, (size_t)n);
}

}  // namespace

void render_paf(const impg_gpu_results &res, const impg_gpu_index &ix, const char *const *range_names,
                const impg_gpu_params_t &p, int32_t merge_distance, bool bedpe, std::vector<std::string> &parts) {
  if (!res.has_cigar) throw Error{IMPG_E_INVALID, "PAF / BEDPE output needs results queried with store_cigar = 1"};
  const size_t nr = res.offsets.size() - 1;
  parts.assign(nr, std::string());
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::string fail_msg;
  unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, nr / 8 + 1));
  auto work = [&]() {
    std::vector<Row> rows;
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= nr || failed.load()) break;
      try {
        rows.clear();
        for (uint64_t k = res.offsets[i]; k < res.offsets[i + 1]; k++) {
          const impg_gpu_interval_t &x = res.intervals[k];
          // perform_query's retain for plain queries (main.rs:11682-11688) comes before the first row is dropped
          if (!p.transitive && p.min_output_length >= 0 && std::abs((int64_t)x.q_last - x.q_first) < p.min_output_length) continue;
          Row r;
          r.iv = x;
          r.cigar.assign(res.cigar_ops.begin() + (long)res.cigar_off[k], res.cigar_ops.begin() + (long)res.cigar_off[k + 1]);
          rows.push_back(std::move(r));
        }
        if (rows.empty()) throw Error{IMPG_E_INVALID, "no result row to drop (the reference panics in Vec::remove(0))"};
        rows.erase(rows.begin());  // the input range itself (main.rs:7474, :7486)
        for (const Row &r : rows)
          if (r.cigar.empty()) throw Error{IMPG_E_UNSUPPORTED, "a result row without CIGAR"};
        merge_rows(rows, merge_distance);
        std::string &s = parts[i];
        std::string rname;
        if (range_names && range_names[i]) rname = range_names[i];
        else {  // "{chrom}:{start}-{end}" (partition.rs:1741, :1762)
          const auto &q = res.ranges[i];
          put_name(rname, ix, q.target_id);
          rname += ':'; rname += std::to_string(q.start); rname += '-'; rname += std::to_string(q.end);
        }
        for (const Row &r : rows) {
          const bool f = row_fwd(r);
          const uint32_t first = (uint32_t)(f ? r.iv.q_first : r.iv.q_last), last = (uint32_t)(f ? r.iv.q_last : r.iv.q_first);
          int32_t m = 0, x = 0, ni = 0, ibp = 0, nd = 0, dbp = 0, bl = 0;
          for (uint32_t op : r.cigar) {
            const int32_t l = op_len(op);
            switch (op_code(op)) {
              case 0: case 4: m += l; bl += l; break;  // 'M' counted as matches (main.rs:12047)
              case 1: x += l; bl += l; break;
              case 2: ni += 1; ibp += l; bl += l; break;
              case 3: nd += 1; dbp += l; bl += l; break;
              default: break;
            }
          }
          volatile float mf = (float)m, g_den = (float)(m + x + ni + nd), b_den = (float)(m + (x + ibp + dbp));
          volatile float gi = mf / g_den, bi = mf / b_den;
          // --original-sequence-coordinates (main.rs:11925-11938, :12014-12046): "base:START-END" names print as
          // "base" with START added; PAF lengths are then 0 (no sequence files to ask)
          const bool orig = p.original_sequence_coordinates != 0;
          auto put_shifted_name = [&](uint32_t id) -> uint32_t {
            if (id < ix.seq.names.size()) return put_original_name(s, ix.seq.names[id], orig);
            put_name(s, ix, id);
            return 0u;
          };
          if (bedpe) {
            const uint32_t qo = put_shifted_name(r.iv.query_id);
            s += '\t'; put_u(s, (uint32_t)first + qo); s += '\t'; put_u(s, (uint32_t)last + qo); s += '\t';
            const uint32_t to = put_shifted_name(r.iv.target_id);
            s += '\t'; put_u(s, (uint32_t)r.iv.t_first + to); s += '\t'; put_u(s, (uint32_t)r.iv.t_last + to);
            s += '\t'; s += rname; s += "\t0\t"; s += f ? '+' : '-'; s += "\t+\tgi:f:"; put_f32(s, gi); s += "\tbi:f:"; put_f32(s, bi);
            s += '\n';
          } else {
            auto seq_len = [&](uint32_t id) -> unsigned long long {
              return id < ix.seq.lens.size() ? (unsigned long long)ix.seq.lens[id] : 0ull;
            };
            const uint32_t qo = put_shifted_name(r.iv.query_id);
            s += '\t'; put_u(s, orig ? 0ull : seq_len(r.iv.query_id)); s += '\t'; put_u(s, (uint32_t)first + qo); s += '\t';
            put_u(s, (uint32_t)last + qo); s += '\t'; s += f ? '+' : '-'; s += '\t';
            const uint32_t to = put_shifted_name(r.iv.target_id);
            s += '\t'; put_u(s, orig ? 0ull : seq_len(r.iv.target_id)); s += '\t'; put_u(s, (uint32_t)r.iv.t_first + to);
            s += '\t'; put_u(s, (uint32_t)r.iv.t_last + to); s += '\t'; s += std::to_string(m); s += '\t'; s += std::to_string(bl);
            s += "\t255\tgi:f:"; put_f32(s, gi); s += "\tbi:f:"; put_f32(s, bi); s += "\tcg:Z:";
            char b[24];
            for (uint32_t op : r.cigar) {
              const uint32_t c = op_code(op);
              int n = snprintf(b, sizeof b, "%d%c", op_len(op), c < 5 ? OP_CHARS[c] : '?');
              s.append(b, (size_t)n);
            }
            s += "\tan:Z:"; s += rname; s += '\n';
          }
        }
      } catch (const Error &e) {
        if (!failed.exchange(e.code)) fail_msg = e.msg;
        break;
      }
    }
  };
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++) th.emplace_back(work);
  for (auto &t : th) t.join();
  if (failed.load()) throw Error{failed.load(), fail_msg};
}

}  // namespace impg
