/*
 * impg_oracle.cpp -- CPU ORACLE: a plain restatement of the reference's
 * interval-query + CIGAR-projection + transitive-closure + BED-merge path.
 *
 * TEST INFRASTRUCTURE ONLY (see impg_oracle.h).  Every function cites the
 * reference lines (pangenome/impg 0.5.0, paths relative to /root/reference)
 * it follows.  The code is deliberately literal and slow: sequential loops,
 * std::vector, no tricks.  Nothing here is shared with the HIP engine.
 *
 * Third-party arithmetic restated from its published algorithm (sources are
 * not in the reference tree):
 *   coitrees 0.4.0 (Cargo.lock:643-646) BasicCOITree  -> struct COITree below.
 *     Only the *visit order* of query() matters to impg; it is restated as
 *     "pre-order over the implicit midpoint BST, except subtrees laid out as
 *     'simple' (<= 8 nodes, reached with childless=true in veb_order_recursion)
 *     which are scanned in sorted order".  VISIT-ORDER PARITY UNPINNED: no
 *     reference test observes it.
 *   rayon par_sort_by_key  -> std::stable_sort (rayon's is documented stable).
 *   rustc-hash FxHashMap iteration order decides impg's sequence ids
 *     (main.rs:11519-11540); ids cross the boundary as data, here they are
 *     assigned in first-seen order.
 */
#include "impg_oracle.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <functional>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace {

std::atomic<int> g_sorted_visits{0}; /* oracle_set_sorted_visits */
thread_local std::string g_err;
thread_local uint64_t g_nproj = 0;

void set_err(const std::string &s) { g_err = s; }

/* ------------------------------------------------------------------------ */
/* CigarOp (impg.rs:75-140)                                                  */
/* ------------------------------------------------------------------------ */
bool cigar_new(int32_t len, char op, uint32_t *out) {
  uint32_t val;
  switch (op) { /* impg.rs:82-89 */
  case '=': val = 0; break;
  case 'X': val = 1; break;
  case 'I': val = 2; break;
  case 'D': val = 3; break;
  case 'M': val = 4; break;
  default: return false; /* panic!("Invalid CIGAR operation") */
  }
  *out = (val << 29) | (uint32_t)len; /* impg.rs:91 */
  return true;
}
inline char cigar_op(uint32_t v) { /* impg.rs:95-105 */
  switch (v >> 29) {
  case 0: return '=';
  case 1: return 'X';
  case 2: return 'I';
  case 3: return 'D';
  case 4: return 'M';
  default: return '?';
  }
}
inline int32_t cigar_len(uint32_t v) { return (int32_t)(v & ((1u << 29) - 1)); } /* :107-109 */
inline int32_t target_delta(uint32_t v) { /* impg.rs:115-121 */
  char o = cigar_op(v);
  return (o == '=' || o == 'X' || o == 'D' || o == 'M') ? cigar_len(v) : 0;
}
inline int32_t query_delta(uint32_t v, bool reverse) { /* impg.rs:123-135 */
  char o = cigar_op(v);
  if (o == '=' || o == 'X' || o == 'I' || o == 'M') return reverse ? -cigar_len(v) : cigar_len(v);
  return 0;
}
inline uint32_t adjust_len(uint32_t v, int32_t delta) { /* impg.rs:137-139 */
  return (v & (7u << 29)) | (uint32_t)(cigar_len(v) + delta);
}

/* parse_cigar_to_delta (impg.rs:2935-2950) */
bool parse_cigar_to_delta(const char *s, size_t n, std::vector<uint32_t> &ops) {
  ops.clear();
  int32_t len = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned char c = (unsigned char)s[i];
    if (c >= '0' && c <= '9') {
      len = len * 10 + (int32_t)(c - '0');
    } else {
      uint32_t v;
      if (!cigar_new(len, (char)c, &v)) return false;
      ops.push_back(v);
      len = 0;
    }
  }
  return true;
}

/* invert_cigar_ops_in_place (impg.rs:144-156) */
void invert_cigar_ops_in_place(std::vector<uint32_t> &ops, bool strand_reverse) {
  for (auto &op : ops) {
    char o = cigar_op(op);
    char n = o == 'I' ? 'D' : (o == 'D' ? 'I' : o);
    uint32_t v = 0;
    cigar_new(cigar_len(op), n, &v);
    op = v;
  }
  if (strand_reverse) std::reverse(ops.begin(), ops.end());
}

/* project_target_range_through_alignment (impg.rs:2760-2898) */
struct Projection {
  int32_t q_start, q_end, t_start, t_end;
  std::vector<uint32_t> cigar;
};
bool project_target_range_through_alignment(int32_t r0, int32_t r1, int32_t target_start,
                                            int32_t target_end, int32_t query_start,
                                            int32_t query_end, bool reverse,
                                            const uint32_t *cigar_ops, size_t n_ops,
                                            Projection &out) {
  const int32_t dir = reverse ? -1 : 1;               /* :2777 */
  int32_t query_pos = reverse ? query_end : query_start; /* :2778-2782 */
  int32_t target_pos = target_start;
  size_t first_op_idx = 0, last_op_idx = 0;
  bool found_overlap = false;
  int32_t projected_query_start = -1, projected_query_end = -1;
  int32_t projected_target_start = -1, projected_target_end = -1;
  int32_t first_op_offset = 0, last_op_remaining = 0;
  const int32_t last_target_pos = std::min(target_end, r1); /* :2798 */

  for (size_t curr_op_idx = 0; curr_op_idx < n_ops; curr_op_idx++) {
    if (target_pos > last_target_pos) break; /* :2802 */
    const uint32_t op = cigar_ops[curr_op_idx];
    const int32_t td = target_delta(op), qd = query_delta(op, reverse);
    if (td == 0) { /* :2807-2821 */
      if (target_pos >= r0) {
        if (!found_overlap) {
          projected_query_start = query_pos;
          projected_target_start = target_pos;
          first_op_idx = curr_op_idx;
          found_overlap = true;
        }
        projected_query_end = query_pos + qd;
        projected_target_end = target_pos;
        last_op_idx = curr_op_idx + 1;
      }
      query_pos += qd;
    } else if (qd == 0) { /* :2822-2841 */
      const int32_t overlap_start = std::max(target_pos, r0);
      const int32_t overlap_end = std::min(target_pos + td, last_target_pos);
      if (overlap_start < overlap_end) {
        if (!found_overlap) {
          projected_query_start = query_pos;
          projected_target_start = overlap_start;
          first_op_idx = curr_op_idx;
          first_op_offset = overlap_start - target_pos;
          found_overlap = true;
        }
        projected_query_end = query_pos;
        projected_target_end = overlap_end;
        last_op_idx = curr_op_idx + 1;
        last_op_remaining = overlap_end - (target_pos + td);
      }
      target_pos += td;
    } else { /* :2842-2867 */
      const int32_t overlap_start = std::max(target_pos, r0);
      const int32_t overlap_end = std::min(target_pos + td, r1);
      if (overlap_start < overlap_end) {
        const int32_t overlap_length = overlap_end - overlap_start;
        const int32_t query_overlap_start = query_pos + (overlap_start - target_pos) * dir;
        const int32_t query_overlap_end = query_overlap_start + overlap_length * dir;
        if (!found_overlap) {
          projected_query_start = query_overlap_start;
          projected_target_start = overlap_start;
          first_op_idx = curr_op_idx;
          first_op_offset = overlap_start - target_pos;
          found_overlap = true;
        }
        projected_query_end = query_overlap_end;
        projected_target_end = overlap_end;
        last_op_idx = curr_op_idx + 1;
        last_op_remaining = overlap_end - (target_pos + td);
      }
      target_pos += td;
      query_pos += qd;
    }
  }
  if (found_overlap && projected_query_start != projected_query_end &&
      projected_target_start != projected_target_end) { /* :2874-2877 */
    out.cigar.assign(cigar_ops + first_op_idx, cigar_ops + last_op_idx);
    if (first_op_offset > 0) out.cigar[0] = adjust_len(out.cigar[0], -first_op_offset);
    if (last_op_remaining < 0) {
      size_t k = last_op_idx - first_op_idx - 1;
      out.cigar[k] = adjust_len(out.cigar[k], last_op_remaining);
    }
    out.q_start = projected_query_start;
    out.q_end = projected_query_end;
    out.t_start = projected_target_start;
    out.t_end = projected_target_end;
    return true;
  }
  return false;
}

/* calculate_gap_compressed_identity (impg.rs:2952-2973) */
double gap_compressed_identity(const uint32_t *ops, size_t n) {
  int32_t m = 0, mm = 0, ins = 0, del = 0;
  for (size_t i = 0; i < n; i++) {
    int32_t len = cigar_len(ops[i]);
    switch (cigar_op(ops[i])) {
    case 'M': case '=': m += len; break;
    case 'X': mm += len; break;
    case 'I': ins += 1; break;
    case 'D': del += 1; break;
    default: break;
    }
  }
  int32_t total = m + mm + ins + del;
  if (total == 0) return 0.0;
  return (double)m / (double)total;
}

/* ------------------------------------------------------------------------ */
/* SortedRanges (impg.rs:242-369)                                            */
/* ------------------------------------------------------------------------ */
struct SortedRanges {
  std::vector<std::pair<int32_t, int32_t>> ranges;
  int32_t sequence_length = 0;
  int32_t min_distance = 0;

  /* Rust binary_search_by_key(&start, |&(s,_)| s): Ok(pos)|Err(pos); starts are
   * strictly increasing in a SortedRanges so both are the lower bound. */
  size_t bsearch(int32_t start) const {
    size_t lo = 0, hi = ranges.size();
    while (lo < hi) {
      size_t mid = lo + (hi - lo) / 2;
      if (ranges[mid].first < start) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
  void merge_forward_from(size_t start_idx) { /* :355-368 */
    size_t write = start_idx, read = start_idx + 1;
    while (read < ranges.size()) {
      if (ranges[write].second >= ranges[read].first) {
        ranges[write].second = std::max(ranges[write].second, ranges[read].second);
      } else {
        write += 1;
        std::swap(ranges[write], ranges[read]);
      }
      read += 1;
    }
    ranges.resize(write + 1);
  }
  std::vector<std::pair<int32_t, int32_t>> insert(std::pair<int32_t, int32_t> new_range) { /* :270-353 */
    int32_t start, end;
    if (new_range.first <= new_range.second) { start = new_range.first; end = new_range.second; }
    else { start = new_range.second; end = new_range.first; }
    size_t i = bsearch(start);
    if (i > 0 && std::abs(start - ranges[i - 1].second) < min_distance) { /* :284 */
      start = ranges[i - 1].second;
      i -= 1;
    } else if (start < min_distance) {
      start = 0;
    }
    if (i < ranges.size() && std::abs(ranges[i].first - end) < min_distance) { /* :292 */
      end = ranges[i].first;
    } else if (end > (sequence_length - min_distance)) {
      end = sequence_length;
    }
    std::vector<std::pair<int32_t, int32_t>> non_overlapping;
    int32_t current = start;
    i = bsearch(start); /* :303 */
    if (i > 0 && ranges[i - 1].second > start) i -= 1;
    while (i < ranges.size() && current < end) { /* :314 */
      int32_t range_start = ranges[i].first, range_end = ranges[i].second;
      if (range_start > end) break;
      if (current < range_start) non_overlapping.push_back({current, range_start});
      current = std::max(current, range_end);
      i += 1;
    }
    if (current < end) non_overlapping.push_back({current, end});
    size_t pos = bsearch(start); /* :330 */
    if (pos > 0 && ranges[pos - 1].second >= start) {
      ranges[pos - 1].second = std::max(ranges[pos - 1].second, end);
      merge_forward_from(pos - 1);
    } else if (pos < ranges.size() && end >= ranges[pos].first) {
      ranges[pos].first = std::min(start, ranges[pos].first);
      ranges[pos].second = std::max(end, ranges[pos].second);
      merge_forward_from(pos);
    } else {
      ranges.insert(ranges.begin() + pos, {start, end});
    }
    return non_overlapping;
  }
};

/* ------------------------------------------------------------------------ */
/* SequenceIndex (seqidx.rs:4-56)                                            */
/* ------------------------------------------------------------------------ */
struct SequenceIndex {
  std::unordered_map<std::string, uint32_t> name_to_id;
  std::vector<std::string> id_to_name;
  std::vector<int64_t> id_to_len; /* -1 = None */
  uint32_t get_or_insert_id(const std::string &name, int64_t length) { /* :22-34 */
    auto it = name_to_id.find(name);
    uint32_t id;
    if (it == name_to_id.end()) {
      id = (uint32_t)id_to_name.size();
      name_to_id.emplace(name, id);
      id_to_name.push_back(name);
      id_to_len.push_back(-1);
    } else id = it->second;
    if (length >= 0 && id_to_len[id] < 0) id_to_len[id] = length; /* first length wins */
    return id;
  }
};

/* AlignmentRecord (alignment_record.rs:12-22) */
struct AlignmentRecord {
  uint32_t query_id;
  uint64_t query_start, query_end;
  uint32_t target_id;
  uint64_t target_start, target_end;
  uint64_t strand_and_data_offset;
  uint64_t data_bytes;
};
const uint64_t STRAND_BIT = 0x8000000000000000ull;   /* impg.rs:178 */
const uint64_t REVERSED_BIT = 0x4000000000000000ull; /* impg.rs:179 */

/* QueryMetadata (impg.rs:164-174) */
struct QueryMetadata {
  uint32_t query_id;
  int32_t target_start, target_end, query_start, query_end;
  uint32_t alignment_file_index;
  uint64_t strand_and_data_offset;
  uint64_t data_bytes;
  bool strand_reverse() const { return (strand_and_data_offset & STRAND_BIT) != 0; }
  bool is_reversed() const { return (strand_and_data_offset & REVERSED_BIT) != 0; }
  uint64_t data_offset() const { return strand_and_data_offset & ~(STRAND_BIT | REVERSED_BIT); }
};
struct IvNode {
  int32_t first, last;
  QueryMetadata metadata;
};

/* ------------------------------------------------------------------------ */
/* coitrees 0.4.0 BasicCOITree, restated (see file header).                  */
/* ------------------------------------------------------------------------ */
const size_t SIMPLE_SUBTREE_CUTOFF = 8;
struct COITree {
  std::vector<IvNode> nodes;         /* sorted by `first`, ties in input order */
  std::vector<int32_t> subtree_last; /* per sorted index: max last over its subtree */
  std::vector<uint8_t> simple_root;  /* per sorted index: root of a 'simple' unit */

  static size_t root_of(size_t s, size_t e) { return s + (e - s) / 2; } /* traverse_recursion */

  int32_t traverse(size_t s, size_t e, uint32_t depth, uint32_t &max_depth) {
    size_t r = root_of(s, e);
    if (depth > max_depth) max_depth = depth;
    int32_t sl = nodes[r].last;
    if (r > s) sl = std::max(sl, traverse(s, r, depth + 1, max_depth));
    if (r + 1 < e) sl = std::max(sl, traverse(r + 1, e, depth + 1, max_depth));
    subtree_last[r] = sl;
    return sl;
  }
  /* mark the complete subtrees rooted `levels` below [s,e)'s root */
  void for_subtrees_at(size_t s, size_t e, uint32_t depth, uint32_t want, uint32_t max_depth) {
    if (s >= e) return;
    if (depth == want) { veb(s, e, want, max_depth); return; }
    size_t r = root_of(s, e);
    for_subtrees_at(s, r, depth + 1, want, max_depth);
    for_subtrees_at(r + 1, e, depth + 1, want, max_depth);
  }
  /* veb_order_recursion restricted to the childless=true chain: only such
   * units can become 'simple' (top parts recurse with childless=false). */
  void veb(size_t s, size_t e, uint32_t min_depth, uint32_t max_depth) {
    size_t n = e - s;
    if (n <= SIMPLE_SUBTREE_CUTOFF) { simple_root[root_of(s, e)] = 1; return; }
    uint32_t pivot_depth = min_depth + (max_depth - min_depth) / 2;
    for_subtrees_at(s, e, min_depth, pivot_depth + 1, max_depth);
  }
  void build(std::vector<IvNode> &&input) {
    nodes = std::move(input);
    bool sorted = true;
    for (size_t i = 1; i < nodes.size(); i++)
      if (nodes[i].first < nodes[i - 1].first) { sorted = false; break; }
    if (!sorted) /* LSD radix sort on `first` == stable sort by first */
      std::stable_sort(nodes.begin(), nodes.end(),
                       [](const IvNode &a, const IvNode &b) { return a.first < b.first; });
    size_t n = nodes.size();
    subtree_last.assign(n, 0);
    simple_root.assign(n, 0);
    if (n == 0) return;
    uint32_t max_depth = 0;
    traverse(0, n, 0, max_depth);
    veb(0, n, 0, max_depth);
  }
  template <class F> void query_recursion(size_t s, size_t e, int32_t first, int32_t last, F &visit) const {
    size_t r = root_of(s, e);
    if (simple_root[r]) {
      for (size_t i = s; i < e; i++) {
        if (last < nodes[i].first) break;
        if (first <= nodes[i].last) visit(nodes[i]);
      }
      return;
    }
    const IvNode &node = nodes[r];
    if (node.first <= last && node.last >= first) visit(node);
    if (r > s) {
      size_t l = root_of(s, r);
      if (subtree_last[l] >= first) query_recursion(s, r, first, last, visit);
    }
    if (r + 1 < e) {
      size_t rr = root_of(r + 1, e);
      if (node.first <= last && subtree_last[rr] >= first) query_recursion(r + 1, e, first, last, visit);
    }
  }
  template <class F> void query(int32_t first, int32_t last, F visit) const {
    if (nodes.empty()) return;
    if (g_sorted_visits) { /* NOT the reference: the engine's IMPG_ORDER_SORTED policy (ascending start, ties in
                              input order), so that both order policies have an exact checker */
      for (const IvNode &nd : nodes) {
        if (last < nd.first) break;
        if (first <= nd.last) visit(nd);
      }
      return;
    }
    query_recursion(0, nodes.size(), first, last, visit);
  }
};

/* ------------------------------------------------------------------------ */
/* alignment files + index                                                   */
/* ------------------------------------------------------------------------ */
/* OneAlnAlignment (onealn.rs:786-802), the fields the approximate mode reads */
struct TpAlignment {
  std::vector<int64_t> tracepoints;
  bool fastga = false;             /* TracepointModeData (onealn.rs:772-782) */
  std::vector<int64_t> diffs;      /* Fastga */
  int64_t trace_spacing = 0;       /* Fastga */
  std::vector<int64_t> query_deltas; /* Standard */
  int64_t max_complexity = 0;      /* Standard */
  int64_t query_contig_start = 0;
};
struct AlnFile {
  std::vector<TpAlignment> tp;  /* tracepoint files (.1aln / .tpa): alignment by record index (impg.rs:568, :612) */
  std::string path;
  int fd = -1;
  const char *mem = nullptr;
  std::string owned;
  size_t mem_len = 0;
  std::unordered_map<uint64_t, std::vector<uint32_t>> preparsed; /* by data_offset */
};

typedef std::unordered_map<uint32_t, COITree> TreeMap;

struct AdjustedInterval { /* impg.rs:225 */
  uint32_t q_id; int32_t q_first, q_last;
  std::vector<uint32_t> cigar;
  uint32_t t_id; int32_t t_first, t_last;
};

} // namespace

struct oracle_index {
  SequenceIndex seq_index;
  std::vector<AlnFile> files;
  TreeMap trees;                    /* Impg (single index over all files) */
  std::vector<TreeMap> file_trees;  /* MultiImpg: one Impg per file */
  bool preparse = false;
  bool approximate = false;  /* built from tracepoints: every projection is project_overlapping_interval_fast */
  size_t n_records = 0;
  ~oracle_index() { for (auto &f : files) if (f.fd >= 0) close(f.fd); }
};

namespace {

/* Rust str::parse::<usize>(): optional '+', at least one digit, digits only. */
bool parse_usize(const char *s, size_t n, uint64_t *out) {
  size_t i = 0;
  if (n > 0 && s[0] == '+') i = 1;
  if (i >= n) return false;
  uint64_t v = 0;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (uint64_t)(s[i] - '0');
  }
  *out = v;
  return true;
}
bool parse_i32(const char *s, size_t n, int32_t *out) { /* str::parse::<i32>() */
  size_t i = 0; bool neg = false;
  if (n > 0 && (s[0] == '+' || s[0] == '-')) { neg = s[0] == '-'; i = 1; }
  if (i >= n) return false;
  int64_t v = 0;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
    if (v > 2147483648LL) return false;
  }
  if (neg) v = -v;
  if (v > 2147483647LL || v < -2147483648LL) return false;
  *out = (int32_t)v;
  return true;
}

/* parse_paf_line (paf.rs:118-177) */
bool parse_paf_line(const char *line, size_t len, uint64_t file_pos, SequenceIndex &seq_index,
                    AlignmentRecord &rec) {
  std::vector<std::pair<const char *, size_t>> fields;
  size_t st = 0;
  for (size_t i = 0; i <= len; i++)
    if (i == len || line[i] == '\t') { fields.push_back({line + st, i - st}); st = i + 1; }
  if (fields.size() < 12) { set_err("Not enough fields in PAF record"); return false; }
  uint64_t qlen, qs, qe, tlen, ts, te;
  if (!parse_usize(fields[1].first, fields[1].second, &qlen) ||
      !parse_usize(fields[2].first, fields[2].second, &qs) ||
      !parse_usize(fields[3].first, fields[3].second, &qe) ||
      !parse_usize(fields[6].first, fields[6].second, &tlen) ||
      !parse_usize(fields[7].first, fields[7].second, &ts) ||
      !parse_usize(fields[8].first, fields[8].second, &te)) { set_err("Invalid field"); return false; }
  if (fields[4].second == 0) { set_err("Expected '+' or '-' for strand"); return false; }
  char sc = fields[4].first[0];
  if (sc != '+' && sc != '-') { set_err("Invalid strand"); return false; }
  std::string qname(fields[0].first, fields[0].second), tname(fields[5].first, fields[5].second);
  uint32_t query_id = seq_index.get_or_insert_id(qname, (int64_t)qlen);
  uint32_t target_id = seq_index.get_or_insert_id(tname, (int64_t)tlen);
  uint64_t cigar_offset = file_pos, cigar_bytes = 0;
  for (auto &f : fields) { /* :155-163 */
    if (f.second >= 5 && memcmp(f.first, "cg:Z:", 5) == 0) {
      cigar_offset += 5;
      cigar_bytes = f.second - 5;
      break;
    } else cigar_offset += f.second + 1;
  }
  rec.query_id = query_id; rec.query_start = qs; rec.query_end = qe;
  rec.target_id = target_id; rec.target_start = ts; rec.target_end = te;
  rec.strand_and_data_offset = cigar_offset; rec.data_bytes = cigar_bytes;
  if (sc == '-') rec.strand_and_data_offset |= STRAND_BIT; else rec.strand_and_data_offset &= ~STRAND_BIT;
  return true;
}

/* parse_paf (paf.rs:179-194): BufRead::lines() strips "\n" and "\r\n";
 * bytes_read += line.len() + 1 */
bool parse_paf(const char *text, size_t len, SequenceIndex &seq_index, std::vector<AlignmentRecord> &records) {
  uint64_t bytes_read = 0;
  size_t pos = 0;
  while (pos < len) {
    size_t eol = pos;
    while (eol < len && text[eol] != '\n') eol++;
    size_t l = eol - pos;
    if (l > 0 && text[pos + l - 1] == '\r') l--;
    AlignmentRecord rec;
    if (!parse_paf_line(text + pos, l, bytes_read, seq_index, rec)) return false;
    records.push_back(rec);
    bytes_read += l + 1;
    pos = eol + 1;
  }
  return true;
}

/* from_multi_alignment_records entry construction (impg.rs:1553-1633) */
void add_entries(const std::vector<AlignmentRecord> &records, uint32_t file_index, bool bidirectional,
                 std::map<uint32_t, std::vector<IvNode>> &intervals) {
  for (const auto &record : records) {
    QueryMetadata fwd;
    fwd.query_id = record.query_id;
    fwd.target_start = (int32_t)record.target_start; fwd.target_end = (int32_t)record.target_end;
    fwd.query_start = (int32_t)record.query_start; fwd.query_end = (int32_t)record.query_end;
    fwd.alignment_file_index = file_index;
    fwd.strand_and_data_offset = record.strand_and_data_offset;
    fwd.data_bytes = record.data_bytes;
    intervals[record.target_id].push_back({(int32_t)record.target_start, (int32_t)record.target_end, fwd});
    if (bidirectional && record.query_id != record.target_id) { /* :1584 */
      QueryMetadata rev;
      rev.query_id = record.target_id;
      rev.target_start = (int32_t)record.query_start; rev.target_end = (int32_t)record.query_end;
      rev.query_start = (int32_t)record.target_start; rev.query_end = (int32_t)record.target_end;
      rev.alignment_file_index = file_index;
      rev.strand_and_data_offset = record.strand_and_data_offset | REVERSED_BIT;
      rev.data_bytes = record.data_bytes;
      intervals[record.query_id].push_back({(int32_t)record.query_start, (int32_t)record.query_end, rev});
    }
  }
}
void build_trees(std::map<uint32_t, std::vector<IvNode>> &intervals, TreeMap &trees) {
  for (auto &kv : intervals) trees[kv.first].build(std::move(kv.second)); /* :1625-1633 */
}

bool read_cigar_bytes(const AlnFile &f, uint64_t offset, size_t n, std::vector<char> &buf) {
  buf.resize(n);
  if (f.mem) { /* memory-backed "file" */
    if (offset + n > f.mem_len) { set_err("read past end"); return false; }
    memcpy(buf.data(), f.mem + offset, n);
    return true;
  }
  size_t got = 0; /* read_exact_at (impg.rs:2923) */
  while (got < n) {
    ssize_t r = pread(f.fd, buf.data() + got, n - got, (off_t)(offset + got));
    if (r <= 0) { set_err("Failed to read CIGAR bytes from '" + f.path + "'"); return false; }
    got += (size_t)r;
  }
  return true;
}

/* get_cigar_ops, PAF branch (impg.rs:495-551) */
bool get_cigar_ops(const oracle_index &ix, const QueryMetadata &md, std::vector<uint32_t> &ops) {
  const AlnFile &f = ix.files[md.alignment_file_index];
  if (md.data_bytes == 0) {
    set_err("The alignment file '" + f.path + "' does not contain CIGAR strings ('cg:Z' tag).");
    return false;
  }
  if (ix.preparse) {
    ops = f.preparsed.at(md.data_offset());
  } else {
    thread_local std::vector<char> buf; /* PAF_CIGAR_BUF (impg.rs:49) */
    if (!read_cigar_bytes(f, md.data_offset(), md.data_bytes, buf)) return false;
    if (!parse_cigar_to_delta(buf.data(), buf.size(), ops)) { set_err("Invalid CIGAR operation"); return false; }
  }
  if (md.is_reversed()) invert_cigar_ops_in_place(ops, md.strand_reverse()); /* :548-550 */
  return true;
}

/* SubsettingResult + scan_overlapping_tracepoints (impg.rs:646-823) */
struct SegInfo { int32_t project_pos, project_delta, seg_start, seg_end, abs_scan_delta, num_diffs; };
struct SubsettingResult {
  size_t first_idx, last_idx;
  int32_t first_query_pos, last_query_pos;
  int num_overlapping_segments;
  double total_matches, total_mismatches;
  SegInfo first_segment_info, last_segment_info;
};
bool scan_overlapping_tracepoints(const TpAlignment &alignment, const QueryMetadata &metadata, int32_t range_start,
                                  int32_t range_end, bool is_reversed_entry, SubsettingResult &out) {
  const bool is_reverse = metadata.strand_reverse();
  const int32_t scan_dir = is_reversed_entry ? 1 : (is_reverse ? -1 : 1);         /* :676-682 */
  const int32_t project_dir = is_reversed_entry ? (is_reverse ? -1 : 1) : 1;       /* :683-691 */
  int32_t scan_pos = is_reversed_entry ? metadata.target_start : (is_reverse ? metadata.target_end : metadata.target_start);
  int32_t project_pos = is_reversed_entry ? (is_reverse ? metadata.query_end : metadata.query_start) : metadata.query_start;
  bool have_first = false;
  size_t first_idx = 0, last_idx = 0;
  int32_t first_project_pos = 0, last_project_pos = 0;
  SegInfo first_info{}, last_info{};
  int num_overlapping = 0;
  double total_matches = 0.0, total_mismatches = 0.0;
  int32_t trace_spacing = 0, first_boundary = 0;
  if (alignment.fastga) { /* :726-735 */
    const int32_t ts = (int32_t)alignment.trace_spacing;
    const int32_t qsc = (int32_t)alignment.query_contig_start;
    trace_spacing = ts;
    first_boundary = ((qsc / ts) + 1) * ts - qsc;
  }
  for (size_t idx = 0; idx < alignment.tracepoints.size(); idx++) {
    const int64_t tracepoint = alignment.tracepoints[idx];
    const int32_t query_delta = alignment.fastga ? (idx == 0 ? first_boundary : trace_spacing) : (int32_t)alignment.query_deltas[idx];
    const int32_t abs_tracepoint = (int32_t)(tracepoint < 0 ? -tracepoint : tracepoint);
    int32_t scan_delta, project_delta, abs_scan_delta;
    if (is_reversed_entry) { scan_delta = query_delta; project_delta = abs_tracepoint * project_dir; abs_scan_delta = query_delta; }
    else { scan_delta = (int32_t)tracepoint * scan_dir; project_delta = query_delta; abs_scan_delta = abs_tracepoint; }
    const int32_t seg_start = std::min(scan_pos, scan_pos + scan_delta);
    const int32_t seg_end = std::max(scan_pos, scan_pos + scan_delta);
    if (seg_start < range_end && seg_end > range_start) { /* :769 */
      num_overlapping += 1;
      int32_t num_diffs;
      if (alignment.fastga) num_diffs = idx < alignment.diffs.size() ? (int32_t)alignment.diffs[idx] : 0;
      else num_diffs = (query_delta == 0 || abs_tracepoint == 0) ? std::max(query_delta, abs_tracepoint) : (int32_t)alignment.max_complexity;
      const SegInfo seg_info{project_pos, project_delta, seg_start, seg_end, abs_scan_delta, num_diffs};
      if (!have_first) { have_first = true; first_idx = idx; first_project_pos = project_pos; first_info = seg_info; }
      last_idx = idx;
      last_project_pos = project_pos + project_delta;
      last_info = seg_info;
      const double aligned_len = std::max((double)std::min(std::abs(project_delta), abs_scan_delta), 0.0);
      total_matches += std::max(aligned_len - (double)num_diffs, 0.0);
      total_mismatches += (double)num_diffs;
    }
    scan_pos += scan_delta;
    project_pos += project_delta;
    if ((scan_dir == -1 && scan_pos <= range_start) || (scan_dir == 1 && scan_pos >= range_end)) break; /* :806-810 */
  }
  if (!have_first) return false;
  out = SubsettingResult{first_idx, last_idx, first_project_pos, last_project_pos, num_overlapping, total_matches, total_mismatches,
                         first_info, last_info};
  return true;
}

/* project_overlapping_interval_fast (impg.rs:1317-1533): returns 1 Some, 0 None, -1 error (the reference panics) */
int project_overlapping_interval_fast(const oracle_index &ix, const QueryMetadata &metadata, uint32_t target_id,
                                      int32_t range_start, int32_t range_end, double min_identity, AdjustedInterval &out) {
  if (metadata.target_start >= range_end || metadata.target_end <= range_start) return 0; /* :1327-1329 */
  const AlnFile &f = ix.files[metadata.alignment_file_index];
  if (metadata.data_offset() >= f.tp.size()) return 0; /* "Cannot fetch tracepoint alignment ... skipping" */
  const TpAlignment &alignment = f.tp[metadata.data_offset()];
  SubsettingResult subset;
  if (!scan_overlapping_tracepoints(alignment, metadata, range_start, range_end, metadata.is_reversed(), subset)) return 0;
  const bool is_reverse = metadata.strand_reverse();
  const int32_t working_query_start = metadata.query_start, working_query_end = metadata.query_end;
  auto refine_boundary = [&](int32_t query_pos, int32_t query_delta, int32_t segment_target_start, int32_t overlap_pos,
                             int32_t abs_target_delta, bool first) -> int32_t {
    const int32_t lo = std::min(working_query_start, working_query_end), hi = std::max(working_query_start, working_query_end);
    if (abs_target_delta == 0) { /* :1381-1398 */
      const int32_t refined_pos = first ? query_pos : query_pos + query_delta;
      return std::min(std::max(refined_pos, lo), hi);
    }
    const double target_fraction = (double)(overlap_pos - segment_target_start) / (double)abs_target_delta;
    const double indel_ratio = (double)query_delta / (double)abs_target_delta;
    const double query_advance = target_fraction * (double)abs_target_delta * indel_ratio;
    const double rounded = std::round(query_advance); /* f64::round: half away from zero */
    int32_t adv; /* `as i32` saturates */
    if (rounded >= 2147483647.0) adv = INT32_MAX; else if (rounded <= -2147483648.0) adv = INT32_MIN; else adv = (int32_t)rounded;
    const int32_t refined_pos = (int32_t)((uint32_t)query_pos + (uint32_t)adv);
    return std::min(std::max(refined_pos, lo), hi);
  };
  const SegInfo &fi = subset.first_segment_info;
  const int32_t overlap_start = std::max(fi.seg_start, range_start);
  const int32_t refined_first = refine_boundary(fi.project_pos, fi.project_delta, fi.seg_start, overlap_start, fi.abs_scan_delta, true);
  const SegInfo &li = subset.last_segment_info;
  const int32_t overlap_end = std::min(li.seg_end, range_end);
  const int32_t refined_last = refine_boundary(li.project_pos, li.project_delta, li.seg_start, overlap_end, li.abs_scan_delta, false);
  std::vector<uint32_t> approx_cigar; /* :1476-1483 */
  if (subset.total_matches > 0.0) { uint32_t v; cigar_new((int32_t)std::round(subset.total_matches), '=', &v); approx_cigar.push_back(v); }
  if (subset.total_mismatches > 0.0) { uint32_t v; cigar_new((int32_t)std::round(subset.total_mismatches), 'X', &v); approx_cigar.push_back(v); }
  if (!std::isnan(min_identity)) {
    if (gap_compressed_identity(approx_cigar.data(), approx_cigar.size()) < min_identity) return 0;
  }
  int32_t query_start = refined_first, query_end = refined_last;
  if (is_reverse && !metadata.is_reversed()) std::swap(query_start, query_end); /* :1497-1501 */
  if (refined_first < 0 || refined_last < 0) { set_err("Projection resulted in negative query coordinates"); return -1; }
  out.q_id = metadata.query_id; out.q_first = query_start; out.q_last = query_end;
  out.t_id = target_id; out.t_first = range_start; out.t_last = range_end;
  out.cigar = std::move(approx_cigar);
  g_nproj++;
  return 1;
}

/* project_overlapping_interval, PAF branch (impg.rs:1260-1312).
 * returns 1 Some, 0 None, -1 error */
int project_overlapping_interval(const oracle_index &ix, const QueryMetadata &md, uint32_t target_id,
                                 int32_t range_start, int32_t range_end, double min_identity,
                                 AdjustedInterval &out) {
  /* an index built from tracepoints answers in approximate mode (approximate_mode = true of impg.rs:1860 ff):
   * the exact mode of .1aln / .tpa needs the sequences and a WFA realignment and is out of scope */
  if (ix.approximate) return project_overlapping_interval_fast(ix, md, target_id, range_start, range_end, min_identity, out);
  thread_local std::vector<uint32_t> cigar_ops;
  if (!get_cigar_ops(ix, md, cigar_ops)) return -1;
  Projection pr;
  if (!project_target_range_through_alignment(range_start, range_end, md.target_start, md.target_end,
                                              md.query_start, md.query_end, md.strand_reverse(),
                                              cigar_ops.data(), cigar_ops.size(), pr))
    return 0;
  if (!std::isnan(min_identity)) { /* :1283-1287 */
    if (gap_compressed_identity(pr.cigar.data(), pr.cigar.size()) < min_identity) return 0;
  }
  out.q_id = md.query_id; out.q_first = pr.q_start; out.q_last = pr.q_end;
  out.t_id = target_id; out.t_first = pr.t_start; out.t_last = pr.t_end;
  out.cigar = std::move(pr.cigar);
  g_nproj++;
  return 1;
}

AdjustedInterval make_self(uint32_t target_id, int32_t s, int32_t e, bool store_cigar) {
  AdjustedInterval a;
  a.q_id = target_id; a.q_first = s; a.q_last = e;
  a.t_id = target_id; a.t_first = s; a.t_last = e;
  if (store_cigar) { uint32_t v = 0; cigar_new(e - s, '=', &v); a.cigar.push_back(v); }
  return a;
}

/* Impg::query (impg.rs:1852-1928) over one TreeMap */
bool impg_query(const oracle_index &ix, const TreeMap &trees, uint32_t target_id, int32_t range_start,
                int32_t range_end, bool store_cigar, double min_identity,
                std::vector<AdjustedInterval> &results) {
  results.clear();
  results.push_back(make_self(target_id, range_start, range_end, store_cigar));
  auto it = trees.find(target_id);
  bool ok = true;
  if (it != trees.end()) {
    it->second.query(range_start, range_end, [&](const IvNode &interval) {
      if (!ok) return;
      AdjustedInterval a;
      int r = project_overlapping_interval(ix, interval.metadata, target_id, range_start, range_end,
                                           min_identity, a);
      if (r < 0) { ok = false; return; }
      if (r == 1) {
        if (!store_cigar) a.cigar.clear();
        results.push_back(std::move(a));
      }
    });
  }
  return ok;
}

struct Hit { /* tuple of impg.rs:2384 */
  uint32_t query_id; int32_t qs, qe; std::vector<uint32_t> cigar; int32_t ts, te; uint32_t cur_target;
};

/* lookup + clipped projection of one frontier range (impg.rs:2392-2460 / 2140-2206) */
bool frontier_hits(const oracle_index &ix, const TreeMap &trees, uint32_t cur_id, int32_t cs, int32_t ce,
                   bool store_cigar, double min_identity, std::vector<Hit> &local,
                   const uint8_t *subset_keep = nullptr, uint32_t root_target = 0) {
  auto it = trees.find(cur_id);
  if (it == trees.end()) return true;
  bool ok = true;
  it->second.query(cs, ce, [&](const IvNode &interval) {
    if (!ok) return;
    int32_t overlap_start = std::max(cs, interval.first);
    int32_t overlap_end = std::min(ce, interval.last);
    if (overlap_start >= overlap_end) return; /* :2401 */
    AdjustedInterval a;
    int r = project_overlapping_interval(ix, interval.metadata, cur_id, overlap_start, overlap_end,
                                         min_identity, a);
    if (r < 0) { ok = false; return; }
    /* subset filter: keep if it is the (original) target or its name matches (impg.rs:2176-2185, :2430-2439);
     * the name matching itself (subset_filter.rs:23-60) stays with the caller: keep[id] is its verdict */
    if (r == 1 && subset_keep && !(a.q_id == root_target || subset_keep[a.q_id])) r = 0;
    if (r == 1) {
      Hit h;
      h.query_id = a.q_id; h.qs = a.q_first; h.qe = a.q_last;
      if (store_cigar) h.cigar = std::move(a.cigar);
      h.ts = a.t_first; h.te = a.t_last; h.cur_target = cur_id;
      local.push_back(std::move(h));
    }
  });
  return ok;
}

/* masked_regions: Option<&FxHashMap<u32, SortedRanges>> (impg.rs:2062, :2316; multi_impg.rs:801) */
typedef std::unordered_map<uint32_t, SortedRanges> MaskMap;

SortedRanges &visited_entry(const oracle_index &ix, std::unordered_map<uint32_t, SortedRanges> &visited,
                            uint32_t id, bool masked_none) { /* impg.rs:2041-2055 */
  auto it = visited.find(id);
  if (it == visited.end()) {
    SortedRanges sr;
    sr.sequence_length = masked_none ? (int32_t)ix.seq_index.id_to_len[id] : 0;
    sr.min_distance = 0;
    it = visited.emplace(id, std::move(sr)).first;
  }
  return it->second;
}

/* the sequential per-hit update shared by BFS (:2482-2558) and DFS (:2218-2281).
 * bfs_short_circuit: BFS checks the next range only if not already rejected (:2538). */
template <class Push>
void update_with_hit(const oracle_index &ix, std::unordered_map<uint32_t, SortedRanges> &visited,
                     Hit &h, int32_t min_output_length, int32_t min_distance_between_ranges,
                     int32_t min_transitive_len, bool bfs_short_circuit, bool masked_none,
                     std::vector<AdjustedInterval> &results, Push push_next) {
  int32_t length = std::abs(h.qe - h.qs);
  bool should_add_to_output = min_output_length >= 0 ? length >= min_output_length : true;
  if (should_add_to_output) {
    AdjustedInterval a;
    a.q_id = h.query_id; a.q_first = h.qs; a.q_last = h.qe; a.cigar = h.cigar;
    a.t_id = h.cur_target; a.t_first = h.ts; a.t_last = h.te;
    results.push_back(std::move(a));
  }
  if (h.query_id != h.cur_target) {
    SortedRanges &ranges = visited_entry(ix, visited, h.query_id, masked_none);
    bool should_add = true;
    if (min_distance_between_ranges > 0) {
      int32_t new_min = std::min(h.qs, h.qe), new_max = std::max(h.qs, h.qe);
      size_t idx = ranges.bsearch(new_min);
      if (idx > 0) {
        int32_t prev_end = ranges.ranges[idx - 1].second;
        if (std::abs(new_min - prev_end) < min_distance_between_ranges) should_add = false;
      }
      if ((!bfs_short_circuit || should_add) && idx < ranges.ranges.size()) {
        int32_t next_start = ranges.ranges[idx].first;
        if (std::abs(next_start - new_max) < min_distance_between_ranges) should_add = false;
      }
    }
    if (should_add) {
      auto new_ranges = ranges.insert({h.qs, h.qe});
      for (auto &nr : new_ranges)
        if (std::abs(nr.second - nr.first) >= min_transitive_len) push_next(h.query_id, nr.first, nr.second);
    }
  }
}

void parallel_for(size_t n, int threads, const std::function<void(size_t)> &f);

/* Impg::query_transitive_bfs (impg.rs:2311-2597) */
bool impg_bfs(const oracle_index &ix, const TreeMap &trees, uint32_t target_id, int32_t range_start,
              int32_t range_end, const oracle_params_t &p, int threads, const MaskMap *mask,
              std::vector<AdjustedInterval> &results, const uint8_t *subset_keep = nullptr) {
  const bool masked_none = mask == nullptr; /* :2330-2335 */
  std::unordered_map<uint32_t, SortedRanges> visited;
  if (mask) visited = *mask;
  auto filtered = visited_entry(ix, visited, target_id, masked_none).insert({range_start, range_end});
  results.clear();
  for (auto &f : filtered) results.push_back(make_self(target_id, f.first, f.second, p.store_cigar));
  struct R { uint32_t id; int32_t s, e; };
  std::vector<R> current;
  for (auto &f : filtered)
    if (std::abs(f.first - f.second) >= p.min_transitive_len) current.push_back({target_id, f.first, f.second});
  uint32_t depth = 0;
  while (!current.empty() && (p.max_depth == 0 || depth < p.max_depth)) {
    std::vector<std::vector<Hit>> query_results(current.size());
    std::atomic<bool> ok{true};
    uint64_t nproj_before = g_nproj;
    std::atomic<uint64_t> nproj_par{0};
    auto body = [&](size_t i) {
      uint64_t b = g_nproj; /* thread-local: worker's or ours */
      if (!frontier_hits(ix, trees, current[i].id, current[i].s, current[i].e, p.store_cigar,
                         p.min_identity, query_results[i], subset_keep, target_id)) ok = false;
      nproj_par += g_nproj - b;
    };
    if (threads > 1 && current.size() > 1) parallel_for(current.size(), threads, body);
    else for (size_t i = 0; i < current.size(); i++) body(i);
    g_nproj = nproj_before + nproj_par;
    if (!ok) return false;
    std::vector<R> next;
    for (auto &qr : query_results)
      for (auto &h : qr)
        update_with_hit(ix, visited, h, p.min_output_length, p.min_distance_between_ranges,
                        p.min_transitive_len, true, masked_none, results,
                        [&](uint32_t id, int32_t s, int32_t e) { next.push_back({id, s, e}); });
    depth += 1;
    if (!next.empty()) { /* :2566-2584 */
      std::stable_sort(next.begin(), next.end(), [](const R &a, const R &b) {
        return a.id != b.id ? a.id < b.id : a.s < b.s;
      });
      size_t write = 0;
      for (size_t read = 1; read < next.size(); read++) {
        if (next[write].id == next[read].id && next[write].e >= next[read].s) {
          next[write].e = std::max(next[write].e, next[read].e);
        } else {
          write += 1;
          std::swap(next[write], next[read]);
        }
      }
      next.resize(write + 1);
    }
    current = std::move(next);
  }
  return true;
}

/* Impg::query_transitive_dfs (impg.rs:2057-2309) */
bool impg_dfs(const oracle_index &ix, const TreeMap &trees, uint32_t target_id, int32_t range_start,
              int32_t range_end, const oracle_params_t &p, const MaskMap *mask,
              std::vector<AdjustedInterval> &results, const uint8_t *subset_keep = nullptr) {
  const bool masked_none = mask == nullptr; /* :2076-2081 */
  std::unordered_map<uint32_t, SortedRanges> visited;
  if (mask) visited = *mask;
  auto filtered = visited_entry(ix, visited, target_id, masked_none).insert({range_start, range_end});
  results.clear();
  struct S { uint32_t id; int32_t s, e; uint32_t depth; };
  std::vector<S> stack;
  for (auto &f : filtered) {
    results.push_back(make_self(target_id, f.first, f.second, p.store_cigar));
    if (std::abs(f.first - f.second) >= p.min_transitive_len) stack.push_back({target_id, f.first, f.second, 0});
  }
  while (!stack.empty()) {
    S cur = stack.back();
    stack.pop_back();
    if (p.max_depth > 0 && cur.depth >= p.max_depth) continue; /* :2125 (skips the re-sort too) */
    std::vector<Hit> hits;
    if (!frontier_hits(ix, trees, cur.id, cur.s, cur.e, p.store_cigar, p.min_identity, hits, subset_keep, target_id)) return false;
    for (auto &h : hits)
      update_with_hit(ix, visited, h, p.min_output_length, p.min_distance_between_ranges,
                      p.min_transitive_len, false, masked_none, results,
                      [&](uint32_t id, int32_t s, int32_t e) { stack.push_back({id, s, e, cur.depth + 1}); });
    std::stable_sort(stack.begin(), stack.end(), [](const S &a, const S &b) { /* :2289 */
      return a.id != b.id ? a.id < b.id : a.s < b.s;
    });
    size_t write = 0;
    for (size_t read = 1; read < stack.size(); read++) {
      if (stack[write].id == stack[read].id && stack[write].e >= stack[read].s) {
        stack[write].e = std::max(stack[write].e, stack[read].e);
      } else {
        write += 1;
        std::swap(stack[write], stack[read]);
      }
    }
    if (!stack.empty()) stack.resize(write + 1); /* truncate(write+1) is a no-op on an empty Vec */
  }
  return true;
}

/* MultiImpg::query_all_indices (multi_impg.rs:495-595); ids are already unified */
bool multi_query_all_indices(const oracle_index &ix, uint32_t target_id, int32_t range_start,
                             int32_t range_end, bool store_cigar, double min_identity,
                             std::vector<AdjustedInterval> &final_results) {
  final_results.clear();
  bool any_location = false;
  bool seen_self = false;
  std::vector<const TreeMap *> subs;  /* one Impg per file; a single file shares the main trees */
  if (ix.file_trees.empty()) subs.push_back(&ix.trees);
  else for (const TreeMap &t : ix.file_trees) subs.push_back(&t);
  for (const TreeMap *tmp : subs) {
    const TreeMap &tm = *tmp;
    if (tm.find(target_id) == tm.end()) continue;
    any_location = true;
    std::vector<AdjustedInterval> local;
    if (!impg_query(ix, tm, target_id, range_start, range_end, store_cigar, min_identity, local)) return false;
    for (auto &r : local) {
      bool is_self = r.q_id == target_id && r.t_id == target_id && r.q_first == range_start && r.q_last == range_end;
      if (is_self) {
        if (!seen_self) { final_results.push_back(std::move(r)); seen_self = true; }
      } else final_results.push_back(std::move(r));
    }
  }
  if (!any_location) { final_results.push_back(make_self(target_id, range_start, range_end, store_cigar)); return true; }
  if (!seen_self) final_results.insert(final_results.begin(), make_self(target_id, range_start, range_end, store_cigar));
  if (final_results.size() > 1) { /* :582-592 */
    AdjustedInterval self_interval = std::move(final_results[0]);
    final_results.erase(final_results.begin());
    std::stable_sort(final_results.begin(), final_results.end(), [](const AdjustedInterval &a, const AdjustedInterval &b) {
      if (a.q_id != b.q_id) return a.q_id < b.q_id;
      if (a.q_first != b.q_first) return a.q_first < b.q_first;
      if (a.q_last != b.q_last) return a.q_last < b.q_last;
      if (a.t_first != b.t_first) return a.t_first < b.t_first;
      return a.t_last < b.t_last;
    });
    final_results.insert(final_results.begin(), std::move(self_interval));
  }
  return true;
}

/* MultiImpg::transitive_query_impl (multi_impg.rs:796-991) */
bool multi_transitive(const oracle_index &ix, uint32_t target_id, int32_t range_start, int32_t range_end,
                      const oracle_params_t &p, bool use_dfs, const MaskMap *mask,
                      std::vector<AdjustedInterval> &results, const uint8_t *subset_keep = nullptr) {
  std::unordered_map<uint32_t, SortedRanges> visited;
  if (mask) visited = *mask; /* :814-815 */
  else for (uint32_t id = 0; id < ix.seq_index.id_to_name.size(); id++) { /* :817-822 */
    SortedRanges sr;
    sr.sequence_length = (int32_t)std::max<int64_t>(ix.seq_index.id_to_len[id], 0);
    visited.emplace(id, std::move(sr));
  }
  auto filtered = visited[target_id].insert({range_start, range_end}); /* :827-830 entry().or_default(): length 0 */
  results.clear();
  struct S { uint32_t id; int32_t s, e; uint32_t depth; };
  std::deque<S> stack;
  for (auto &f : filtered) {
    results.push_back(make_self(target_id, f.first, f.second, p.store_cigar));
    if (std::abs(f.first - f.second) >= p.min_transitive_len) stack.push_back({target_id, f.first, f.second, 0});
  }
  while (!stack.empty()) {
    S cur;
    if (use_dfs) { cur = stack.back(); stack.pop_back(); } else { cur = stack.front(); stack.pop_front(); }
    if (p.max_depth > 0 && cur.depth >= p.max_depth) continue;
    std::vector<AdjustedInterval> step;
    if (!multi_query_all_indices(ix, cur.id, cur.s, cur.e, p.store_cigar, p.min_identity, step)) return false;
    for (auto &result : step) {
      uint32_t query_id = result.q_id;
      if (query_id == cur.id) continue; /* :883-885 */
      if (subset_keep && query_id != target_id && !subset_keep[query_id]) continue; /* :888-896 */
      int32_t aqs = std::min(result.q_first, result.q_last), aqe = std::max(result.q_first, result.q_last);
      int32_t length = std::abs(result.q_last - result.q_first);
      bool out = p.min_output_length >= 0 ? length >= p.min_output_length : true;
      if (out) results.push_back(result);
      auto vit = visited.find(query_id); /* :919-922 or_insert_with(real length) */
      if (vit == visited.end()) {
        SortedRanges sr;
        sr.sequence_length = (int32_t)std::max<int64_t>(ix.seq_index.id_to_len[query_id], 0);
        vit = visited.emplace(query_id, std::move(sr)).first;
      }
      SortedRanges &ranges = vit->second;
      bool should_add = true;
      if (p.min_distance_between_ranges > 0) {
        size_t idx = ranges.bsearch(aqs);
        if (idx > 0 && std::abs(aqs - ranges.ranges[idx - 1].second) < p.min_distance_between_ranges) should_add = false;
        if (idx < ranges.ranges.size() && std::abs(ranges.ranges[idx].first - aqe) < p.min_distance_between_ranges) should_add = false;
      }
      if (should_add) {
        auto nr = ranges.insert({aqs, aqe});
        for (auto &r : nr)
          if (std::abs(r.second - r.first) >= p.min_transitive_len) stack.push_back({query_id, r.first, r.second, cur.depth + 1});
      }
    }
    std::stable_sort(stack.begin(), stack.end(), [](const S &a, const S &b) { /* :969-970 */
      return a.id != b.id ? a.id < b.id : a.s < b.s;
    });
    size_t write = 0;
    for (size_t read = 1; read < stack.size(); read++) {
      if (stack[write].id == stack[read].id && stack[write].e >= stack[read].s) {
        stack[write].e = std::max(stack[write].e, stack[read].e);
      } else {
        write += 1;
        std::swap(stack[write], stack[read]);
      }
    }
    if (!stack.empty()) stack.resize(write + 1);
  }
  return true;
}

/* dispatch as perform_query does (main.rs:11641-11699), without the retain */
bool run_query(const oracle_index &ix, uint32_t target_id, int32_t s, int32_t e, const oracle_params_t &p,
               int threads, std::vector<AdjustedInterval> &results, const MaskMap *mask = nullptr,
               const uint8_t *subset_keep = nullptr) {
  if (p.transitive) {
    if (p.multi_impg) return multi_transitive(ix, target_id, s, e, p, p.dfs != 0, mask, results, subset_keep);
    if (p.dfs) return impg_dfs(ix, ix.trees, target_id, s, e, p, mask, results, subset_keep);
    return impg_bfs(ix, ix.trees, target_id, s, e, p, threads, mask, results, subset_keep);
  }
  bool ok = p.multi_impg ? multi_query_all_indices(ix, target_id, s, e, p.store_cigar, p.min_identity, results)
                         : impg_query(ix, ix.trees, target_id, s, e, p.store_cigar, p.min_identity, results);
  if (ok && subset_keep) { /* non-transitive: filtered after the query (main.rs:11693-11696, subset_filter.rs:84-100) */
    size_t w = 0;
    for (size_t i = 0; i < results.size(); i++)
      if (results[i].q_id == target_id || subset_keep[results[i].q_id]) {
        if (w != i) results[w] = std::move(results[i]);
        w++;
      }
    results.resize(w);
  }
  return ok;
}

/* ------------------------------------------------------------------------ */
/* BED merge (main.rs:12858-13011, 12474-12560)                              */
/* ------------------------------------------------------------------------ */
size_t uf_find(std::vector<size_t> &parent, size_t x) {
  while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
  return x;
}
void merge_adjusted_intervals_gap_2d(std::vector<oracle_interval_t> &results, int32_t merge_distance) {
  if (results.size() <= 1 || merge_distance < 0) return;
  const int64_t d = merge_distance;
  std::map<std::tuple<uint32_t, uint32_t, bool>, std::vector<size_t>> groups;
  for (size_t i = 0; i < results.size(); i++) {
    bool strand_fwd = results[i].q_first <= results[i].q_last;
    groups[{results[i].query_id, results[i].target_id, strand_fwd}].push_back(i);
  }
  size_t n = results.size();
  std::vector<size_t> parent(n);
  for (size_t i = 0; i < n; i++) parent[i] = i;
  for (auto &kv : groups) {
    bool strand_fwd = std::get<2>(kv.first);
    std::vector<size_t> indices = kv.second;
    std::stable_sort(indices.begin(), indices.end(), [&](size_t a, size_t b) {
      int32_t ka = strand_fwd ? results[a].q_first : -results[a].q_first;
      int32_t kb = strand_fwd ? results[b].q_first : -results[b].q_first;
      return ka < kb;
    });
    for (size_t a_pos = 0; a_pos < indices.size(); a_pos++) {
      size_t ia = indices[a_pos];
      const auto &A = results[ia];
      int64_t qa_start = strand_fwd ? A.q_first : A.q_last, qa_end = strand_fwd ? A.q_last : A.q_first;
      int64_t ta_start = A.t_first, ta_end = A.t_last;
      for (size_t b_pos = a_pos + 1; b_pos < indices.size(); b_pos++) {
        size_t ib = indices[b_pos];
        const auto &B = results[ib];
        int64_t qb_start = strand_fwd ? B.q_first : B.q_last;
        if (qb_start < qa_start) continue;
        int64_t q_gap = qb_start - qa_end;
        if (q_gap > d) break;
        int64_t tb_start = B.t_first, tb_end = B.t_last;
        int64_t t_gap; bool t_forward;
        if (strand_fwd) { t_gap = tb_start - ta_end; t_forward = tb_start > ta_start; }
        else { t_gap = ta_start - tb_end; t_forward = tb_end < ta_end; }
        if (!t_forward || t_gap > d) continue;
        size_t ra = uf_find(parent, ia), rb = uf_find(parent, ib);
        if (ra != rb) parent[ra] = rb;
      }
    }
  }
  std::map<size_t, std::vector<size_t>> buckets;
  for (size_t i = 0; i < n; i++) buckets[uf_find(parent, i)].push_back(i);
  std::vector<oracle_interval_t> merged;
  std::vector<bool> taken(n, false);
  for (size_t i = 0; i < n; i++) {
    if (taken[i]) continue;
    size_t r = uf_find(parent, i);
    auto it = buckets.find(r);
    if (it == buckets.end()) continue;
    std::vector<size_t> members = std::move(it->second);
    buckets.erase(it);
    for (size_t m : members) taken[m] = true;
    bool strand_fwd = results[members[0]].q_first <= results[members[0]].q_last;
    std::vector<size_t> ordered = members;
    std::stable_sort(ordered.begin(), ordered.end(), [&](size_t a, size_t b) {
      int32_t ka = strand_fwd ? results[a].q_first : -results[a].q_first;
      int32_t kb = strand_fwd ? results[b].q_first : -results[b].q_first;
      return ka < kb;
    });
    const auto &first = results[ordered[0]];
    int32_t q_lo = first.q_first, q_hi = first.q_last, t_lo = first.t_first, t_hi = first.t_last;
    for (size_t idx : ordered) {
      const auto &x = results[idx];
      if (strand_fwd) { q_lo = std::min(q_lo, x.q_first); q_hi = std::max(q_hi, x.q_last); }
      else { q_lo = std::max(q_lo, x.q_first); q_hi = std::min(q_hi, x.q_last); }
      t_lo = std::min(t_lo, x.t_first); t_hi = std::max(t_hi, x.t_last);
    }
    merged.push_back({first.query_id, q_lo, q_hi, first.target_id, t_lo, t_hi});
  }
  results = std::move(merged);
}

int32_t sat_sub(int32_t a, int32_t b) {
  int64_t r = (int64_t)a - (int64_t)b;
  if (r > 2147483647LL) return 2147483647;
  if (r < -2147483648LL) return (int32_t)-2147483648LL;
  return (int32_t)r;
}
void merge_query_adjusted_intervals(std::vector<oracle_interval_t> &results, int32_t merge_distance, bool merge_strands) {
  if (!(results.size() > 1 && (merge_distance >= 0 || merge_strands))) return;
  std::stable_sort(results.begin(), results.end(), [](const oracle_interval_t &a, const oracle_interval_t &b) {
    bool af = a.q_first <= a.q_last, bf = b.q_first <= b.q_last;
    int32_t as = af ? a.q_first : a.q_last, bs = bf ? b.q_first : b.q_last;
    if (a.query_id != b.query_id) return a.query_id < b.query_id;
    if (as != bs) return as < bs;
    return (int)(!af) < (int)(!bf);
  });
  size_t write_idx = 0;
  for (size_t read_idx = 1; read_idx < results.size(); read_idx++) {
    const auto curr = results[write_idx];
    const auto next = results[read_idx];
    bool curr_is_forward = curr.q_first <= curr.q_last, next_is_forward = next.q_first <= next.q_last;
    int32_t curr_start = curr_is_forward ? curr.q_first : curr.q_last, curr_end = curr_is_forward ? curr.q_last : curr.q_first;
    int32_t next_start = next_is_forward ? next.q_first : next.q_last, next_end = next_is_forward ? next.q_last : next.q_first;
    if (merge_distance < 0 || curr.query_id != next.query_id ||
        (!merge_strands && curr_is_forward != next_is_forward) || next_start > curr_end + merge_distance) {
      write_idx += 1;
      if (write_idx != read_idx) std::swap(results[write_idx], results[read_idx]);
    } else {
      int32_t merged_start = std::min(curr_start, next_start), merged_end = std::max(curr_end, next_end);
      bool merged_is_forward;
      if (merge_strands && curr_is_forward != next_is_forward) {
        int32_t curr_len = sat_sub(curr_end, curr_start), next_len = sat_sub(next_end, next_start);
        merged_is_forward = next_len > curr_len ? next_is_forward : curr_is_forward;
      } else merged_is_forward = curr_is_forward;
      if (merged_is_forward) { results[write_idx].q_first = merged_start; results[write_idx].q_last = merged_end; }
      else { results[write_idx].q_first = merged_end; results[write_idx].q_last = merged_start; }
    }
  }
  results.resize(write_idx + 1);
}

/* rayon keeps one pool of workers for the process (the reference's par_iter calls, impg.rs:2384-2465, reuse
 * it); spawning std::threads per BFS level would time thread creation, not the algorithm.  One job at a time
 * (callers never nest): the caller publishes it, takes part itself, and waits for the workers that joined. */
struct WorkerPool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  const std::function<void(size_t)> *job = nullptr;
  size_t job_n = 0, grain = 1;
  std::atomic<size_t> next{0};
  uint64_t epoch = 0;
  int want = 0, joined = 0, active = 0;
  bool stop = false;
  void drain() {
    for (;;) {
      size_t b = next.fetch_add(grain);
      if (b >= job_n) break;
      size_t e = std::min(job_n, b + grain);
      for (size_t i = b; i < e; i++) (*job)(i);
    }
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || (epoch != seen && joined < want); });
      if (stop) return;
      seen = epoch;
      joined++; active++;
      lk.unlock();
      drain();
      lk.lock();
      if (--active == 0) cv_done.notify_all();
    }
  }
  void run(size_t n, int threads, const std::function<void(size_t)> &f) {
    std::unique_lock<std::mutex> lk(m);
    while ((int)workers.size() < threads - 1) workers.emplace_back([this] { worker(); });
    job = &f; job_n = n; next = 0;
    grain = std::max<size_t>(1, n / ((size_t)threads * 8));
    want = threads - 1; joined = 0; active = 0;
    epoch++;
    lk.unlock();
    cv_work.notify_all();
    drain();
    lk.lock();
    want = joined;  /* late workers stay asleep: the job is exhausted */
    cv_done.wait(lk, [&] { return active == 0; });
    job = nullptr;
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_work.notify_all();
    for (auto &t : workers) t.join();
  }
};
WorkerPool &pool() { static WorkerPool p; return p; }
std::mutex g_pool_user;

void parallel_for(size_t n, int threads, const std::function<void(size_t)> &f) {
  int T = std::max(1, std::min<int>(threads, (int)std::min<size_t>(n, 1u << 20)));
  if (T == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
  std::lock_guard<std::mutex> user(g_pool_user);
  pool().run(n, T, f);
}

void append(char **buf, size_t *len, size_t *cap, const char *s, size_t n) {
  if (*len + n + 1 > *cap) {
    size_t nc = std::max<size_t>(*cap * 2, *len + n + 1024);
    *buf = (char *)realloc(*buf, nc);
    *cap = nc;
  }
  memcpy(*buf + *len, s, n);
  *len += n;
  (*buf)[*len] = 0;
}

oracle_index *finish_index(oracle_index *ix, std::vector<std::vector<AlignmentRecord>> &recs, bool bidirectional, bool preparse) {
  std::map<uint32_t, std::vector<IvNode>> all;
  if (recs.size() > 1) ix->file_trees.resize(recs.size());
  for (size_t f = 0; f < recs.size(); f++) {
    add_entries(recs[f], (uint32_t)f, bidirectional, all);
    if (recs.size() > 1) {
      std::map<uint32_t, std::vector<IvNode>> per;
      add_entries(recs[f], (uint32_t)f, bidirectional, per);
      build_trees(per, ix->file_trees[f]);
    }
    ix->n_records += recs[f].size();
  }
  build_trees(all, ix->trees);
  ix->preparse = preparse;
  if (preparse) {
    for (size_t f = 0; f < recs.size(); f++) {
      AlnFile &af = ix->files[f];
      std::vector<char> buf;
      for (auto &r : recs[f]) {
        if (r.data_bytes == 0) continue;
        uint64_t off = r.strand_and_data_offset & ~(STRAND_BIT | REVERSED_BIT);
        if (!read_cigar_bytes(af, off, r.data_bytes, buf)) { delete ix; return nullptr; }
        std::vector<uint32_t> ops;
        if (!parse_cigar_to_delta(buf.data(), buf.size(), ops)) { set_err("Invalid CIGAR operation"); delete ix; return nullptr; }
        af.preparsed.emplace(off, std::move(ops));
      }
    }
  }
  return ix;
}

} // namespace

/* ======================================================================== */
/* C interface                                                               */
/* ======================================================================== */
extern "C" {

const char *oracle_last_error(void) { return g_err.c_str(); }
void oracle_set_sorted_visits(int on) { g_sorted_visits = on ? 1 : 0; }
uint64_t oracle_last_projection_count(void) { return g_nproj; }

long oracle_parse_cigar(const char *cigar, size_t len, uint32_t *ops_out, size_t cap) {
  std::vector<uint32_t> ops;
  if (!parse_cigar_to_delta(cigar, len, ops)) return -1;
  for (size_t i = 0; i < ops.size() && i < cap; i++) ops_out[i] = ops[i];
  return (long)ops.size();
}
void oracle_invert_cigar(uint32_t *ops, size_t n, int strand_reverse) {
  std::vector<uint32_t> v(ops, ops + n);
  invert_cigar_ops_in_place(v, strand_reverse != 0);
  std::copy(v.begin(), v.end(), ops);
}
int oracle_project(int32_t r0, int32_t r1, int32_t ts, int32_t te, int32_t qs, int32_t qe, int strand_reverse,
                   const uint32_t *ops, size_t n_ops, int32_t *out4, uint32_t *slice_out, size_t *slice_len) {
  Projection pr;
  if (!project_target_range_through_alignment(r0, r1, ts, te, qs, qe, strand_reverse != 0, ops, n_ops, pr)) return 0;
  out4[0] = pr.q_start; out4[1] = pr.q_end; out4[2] = pr.t_start; out4[3] = pr.t_end;
  if (slice_out) std::copy(pr.cigar.begin(), pr.cigar.end(), slice_out);
  if (slice_len) *slice_len = pr.cigar.size();
  return 1;
}
double oracle_gap_compressed_identity(const uint32_t *ops, size_t n) { return gap_compressed_identity(ops, n); }

struct oracle_sorted_ranges { SortedRanges sr; };
oracle_sorted_ranges_t *oracle_sr_new(int32_t sequence_length, int32_t min_distance) {
  auto *p = new oracle_sorted_ranges();
  p->sr.sequence_length = sequence_length; p->sr.min_distance = min_distance;
  return p;
}
void oracle_sr_free(oracle_sorted_ranges_t *p) { delete p; }
long oracle_sr_insert(oracle_sorted_ranges_t *p, int32_t a, int32_t b, int32_t *pieces_out, size_t cap) {
  auto v = p->sr.insert({a, b});
  for (size_t i = 0; i < v.size() && i < cap; i++) { pieces_out[2 * i] = v[i].first; pieces_out[2 * i + 1] = v[i].second; }
  return (long)v.size();
}
long oracle_sr_get(oracle_sorted_ranges_t *p, int32_t *out, size_t cap) {
  for (size_t i = 0; i < p->sr.ranges.size() && i < cap; i++) { out[2 * i] = p->sr.ranges[i].first; out[2 * i + 1] = p->sr.ranges[i].second; }
  return (long)p->sr.ranges.size();
}

oracle_index_t *oracle_index_from_paf(const char *const *paths, int n_paths, int bidirectional, int preparse) {
  auto *ix = new oracle_index();
  std::vector<std::vector<AlignmentRecord>> recs(n_paths);
  for (int f = 0; f < n_paths; f++) {
    AlnFile af;
    af.path = paths[f];
    af.fd = open(paths[f], O_RDONLY);
    if (af.fd < 0) { set_err(std::string("Failed to open file '") + paths[f] + "'"); delete ix; return nullptr; }
    off_t sz = lseek(af.fd, 0, SEEK_END);
    std::string text((size_t)sz, '\0');
    size_t got = 0;
    while (got < (size_t)sz) {
      ssize_t r = pread(af.fd, &text[got], (size_t)sz - got, (off_t)got);
      if (r <= 0) break;
      got += (size_t)r;
    }
    ix->files.push_back(std::move(af));
    if (!parse_paf(text.data(), text.size(), ix->seq_index, recs[f])) { delete ix; return nullptr; }
  }
  return finish_index(ix, recs, bidirectional != 0, preparse != 0);
}
oracle_index_t *oracle_index_from_paf_text(const char *text, size_t len, int bidirectional, int preparse) {
  auto *ix = new oracle_index();
  AlnFile af;
  af.path = "<memory>";
  af.owned.assign(text, len);
  ix->files.push_back(std::move(af));
  ix->files[0].mem = ix->files[0].owned.data();
  ix->files[0].mem_len = len;
  std::vector<std::vector<AlignmentRecord>> recs(1);
  if (!parse_paf(ix->files[0].mem, len, ix->seq_index, recs[0])) { delete ix; return nullptr; }
  return finish_index(ix, recs, bidirectional != 0, preparse != 0);
}
/* An index over tracepoint alignments (what Impg::from_multi_alignment_records builds from .1aln / .tpa records:
 * data_offset = the alignment's index in its file, impg.rs:568, :612).  Queries on it run in approximate mode. */
oracle_index_t *oracle_index_from_tracepoints(const oracle_tp_record_t *records, size_t n_records, const int32_t *tracepoints,
                                              const int32_t *query_deltas, const int32_t *diffs, int fastga, int32_t trace_spacing,
                                              int32_t max_complexity, const int64_t *seq_len, uint32_t n_seq, int bidirectional) {
  auto *ix = new oracle_index();
  for (uint32_t i = 0; i < n_seq; i++) ix->seq_index.get_or_insert_id("seq" + std::to_string(i), seq_len[i]);
  AlnFile af;
  af.path = "<tracepoints>";
  std::vector<std::vector<AlignmentRecord>> recs(1);
  for (size_t i = 0; i < n_records; i++) {
    const oracle_tp_record_t &r = records[i];
    TpAlignment a;
    a.fastga = fastga != 0;
    a.trace_spacing = trace_spacing; a.max_complexity = max_complexity;
    a.query_contig_start = r.query_contig_start;
    for (uint32_t k = 0; k < r.n_segs; k++) {
      a.tracepoints.push_back(tracepoints[r.seg_off + k]);
      if (fastga) a.diffs.push_back(diffs[r.seg_off + k]); else a.query_deltas.push_back(query_deltas[r.seg_off + k]);
    }
    af.tp.push_back(std::move(a));
    AlignmentRecord rec;
    rec.query_id = r.query_id; rec.query_start = (uint64_t)r.query_start; rec.query_end = (uint64_t)r.query_end;
    rec.target_id = r.target_id; rec.target_start = (uint64_t)r.target_start; rec.target_end = (uint64_t)r.target_end;
    rec.strand_and_data_offset = (uint64_t)i | (r.strand ? STRAND_BIT : 0);
    rec.data_bytes = r.n_segs;
    recs[0].push_back(rec);
  }
  ix->files.push_back(std::move(af));
  ix->approximate = true;
  return finish_index(ix, recs, bidirectional != 0, false);
}
/* ---- the reference's own index file, "IMPGIDX2" (writer impg.rs:1655-1721, reader :1787-1850 + :1724-1767) ----
 * 16-byte header (magic, u64 LE offset of the forest map), then bincode 2 `config::standard()` values: the
 * SequenceIndex (seqidx.rs:5-10), one (u32 target_id, Vec<SerializableInterval>) per tree (impg.rs:235-240,
 * :164-174), the ForestMap (forest_map.rs:6-9).  bincode 2.0.1 (Cargo.lock:186-188) is not in the tree; its
 * published standard encoding is restated: little-endian varint integers (< 251 one byte, else 0xFB + u16,
 * 0xFC + u32, 0xFD + u64), signed integers zig-zagged first, usize as u64, strings / sequences / maps a varint
 * length and then the elements (maps: key, value), structs and tuples their fields in order, no framing.
 * PARITY UNPINNED: the reference tree holds no .impg file to check these bytes against. */
namespace {
struct Enc {
  std::string b;
  void u(uint64_t v) {
    if (v < 251) b.push_back((char)v);
    else if (v <= 0xFFFFull) { b.push_back((char)0xFB); for (int i = 0; i < 2; i++) b.push_back((char)(v >> (8 * i))); }
    else if (v <= 0xFFFFFFFFull) { b.push_back((char)0xFC); for (int i = 0; i < 4; i++) b.push_back((char)(v >> (8 * i))); }
    else { b.push_back((char)0xFD); for (int i = 0; i < 8; i++) b.push_back((char)(v >> (8 * i))); }
  }
  void i(int64_t v) { u(((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
  void str(const std::string &x) { u(x.size()); b += x; }
};
struct Dec {
  const unsigned char *p, *e;
  bool ok = true;
  uint64_t le(int n) { uint64_t v = 0; if (e - p < n) { ok = false; return 0; } for (int i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i); p += n; return v; }
  uint64_t u() {
    if (p >= e) { ok = false; return 0; }
    unsigned char t = *p++;
    if (t < 251) return t;
    if (t == 0xFB) return le(2);
    if (t == 0xFC) return le(4);
    if (t == 0xFD) return le(8);
    ok = false; return 0;
  }
  int64_t i() { uint64_t z = u(); return (int64_t)(z >> 1) ^ -(int64_t)(z & 1); }
  std::string str() { uint64_t n = u(); if (!ok || (uint64_t)(e - p) < n) { ok = false; return ""; } std::string x((const char *)p, n); p += n; return x; }
};
} // namespace

/* serialize_with_forest_map (impg.rs:1655-1721).  Trees go out in ascending target id (the reference: FxHashMap
 * iteration order, arbitrary), each tree's intervals in this oracle's node order -- ascending `first`, ties in input
 * order -- unless shuffle_seed != 0, which permutes them: the reference writes them in coitrees' internal layout
 * order (tree.iter(), :1686), and a reader rebuilds the tree from whatever order it finds (:1745-1755), so no order
 * is privileged.  Returns 0 or <0. */
int oracle_index_write_impg(const oracle_index_t *ix, const char *path, uint64_t shuffle_seed) {
  Enc seq;
  const auto &si = ix->seq_index;
  seq.u(si.name_to_id.size());
  for (uint32_t id = 0; id < si.id_to_name.size(); id++) { seq.str(si.id_to_name[id]); seq.u(id); }
  seq.u(si.id_to_name.size());
  for (uint32_t id = 0; id < si.id_to_name.size(); id++) { seq.u(id); seq.str(si.id_to_name[id]); }
  size_t n_len = 0;
  for (auto l : si.id_to_len) n_len += l >= 0;
  seq.u(n_len);
  for (uint32_t id = 0; id < si.id_to_len.size(); id++) if (si.id_to_len[id] >= 0) { seq.u(id); seq.u((uint64_t)si.id_to_len[id]); }
  seq.u(si.id_to_name.size()); /* next_id */
  std::string body = seq.b;
  std::vector<std::pair<uint32_t, uint64_t>> forest;
  std::vector<uint32_t> ids;
  for (auto &kv : ix->trees) ids.push_back(kv.first);
  std::sort(ids.begin(), ids.end());
  uint64_t rng = shuffle_seed;
  for (uint32_t t : ids) {
    forest.push_back({t, 16 + body.size()});
    std::vector<IvNode> nodes = ix->trees.at(t).nodes;
    if (shuffle_seed) for (size_t k = nodes.size(); k > 1; k--) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(nodes[k - 1], nodes[(rng >> 33) % k]); }
    Enc e;
    e.u(t);
    e.u(nodes.size());
    for (auto &nd : nodes) {
      e.i(nd.first); e.i(nd.last);
      const QueryMetadata &m = nd.metadata;
      e.u(m.query_id); e.i(m.target_start); e.i(m.target_end); e.i(m.query_start); e.i(m.query_end);
      e.u(m.alignment_file_index); e.u(m.strand_and_data_offset); e.u(m.data_bytes);
    }
    body += e.b;
  }
  const uint64_t fmo = 16 + body.size();
  Enc fm;
  fm.u(forest.size());
  for (auto &kv : forest) { fm.u(kv.first); fm.u(kv.second); }
  FILE *f = fopen(path, "wb");
  if (!f) { set_err(std::string("cannot create ") + path); return -1; }
  fwrite("IMPGIDX2", 1, 8, f);
  unsigned char off[8];
  for (int i = 0; i < 8; i++) off[i] = (unsigned char)(fmo >> (8 * i));
  fwrite(off, 1, 8, f);
  fwrite(body.data(), 1, body.size(), f);
  fwrite(fm.b.data(), 1, fm.b.size(), f);
  fclose(f);
  return 0;
}

/* load_from_file + load_tree_from_disk for every target (impg.rs:1787-1850, :1724-1767): the trees are rebuilt
 * with BasicCOITree::new from the intervals in FILE order; CIGARs are read from the alignment files given here, in
 * the order the index was built with (:1789, :1844). */
oracle_index_t *oracle_index_from_impg(const char *path, const char *const *alignment_files, int n_files, int preparse) {
  FILE *f = fopen(path, "rb");
  if (!f) { set_err(std::string("cannot open ") + path); return nullptr; }
  std::string data;
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
  fclose(f);
  if (data.size() < 16 || (memcmp(data.data(), "IMPGIDX2", 8) != 0 && memcmp(data.data(), "IMPGIDX1", 8) != 0)) {
    set_err("Invalid magic bytes - not a valid IMPG index file");
    return nullptr;
  }
  uint64_t fmo = 0;
  for (int i = 0; i < 8; i++) fmo |= (uint64_t)(unsigned char)data[8 + i] << (8 * i);
  auto *ix = new oracle_index();
  Dec d{(const unsigned char *)data.data() + 16, (const unsigned char *)data.data() + data.size()};
  std::vector<std::pair<std::string, uint32_t>> n2i;
  uint64_t n = d.u();
  for (uint64_t k = 0; k < n && d.ok; k++) { std::string nm = d.str(); uint32_t id = (uint32_t)d.u(); n2i.push_back({nm, id}); }
  n = d.u();
  std::map<uint32_t, std::string> i2n;
  for (uint64_t k = 0; k < n && d.ok; k++) { uint32_t id = (uint32_t)d.u(); i2n[id] = d.str(); }
  n = d.u();
  std::map<uint32_t, uint64_t> i2l;
  for (uint64_t k = 0; k < n && d.ok; k++) { uint32_t id = (uint32_t)d.u(); i2l[id] = d.u(); }
  const uint32_t next_id = (uint32_t)d.u();
  if (!d.ok) { set_err("Failed to load sequence index"); delete ix; return nullptr; }
  ix->seq_index.id_to_name.assign(next_id, std::string());
  ix->seq_index.id_to_len.assign(next_id, -1);
  for (auto &kv : i2n) if (kv.first < next_id) ix->seq_index.id_to_name[kv.first] = kv.second;
  for (auto &kv : i2l) if (kv.first < next_id) ix->seq_index.id_to_len[kv.first] = (int64_t)kv.second;
  for (auto &kv : n2i) ix->seq_index.name_to_id.emplace(kv.first, kv.second);
  Dec fm{(const unsigned char *)data.data() + fmo, (const unsigned char *)data.data() + data.size()};
  if (fmo > data.size()) { set_err("Failed to load forest map"); delete ix; return nullptr; }
  std::vector<std::pair<uint32_t, uint64_t>> forest;
  n = fm.u();
  for (uint64_t k = 0; k < n && fm.ok; k++) { uint32_t t = (uint32_t)fm.u(); forest.push_back({t, fm.u()}); }
  if (!fm.ok) { set_err("Failed to load forest map"); delete ix; return nullptr; }
  for (int k = 0; k < n_files; k++) {
    AlnFile af;
    af.path = alignment_files[k];
    af.fd = open(alignment_files[k], O_RDONLY);
    if (af.fd < 0) { set_err(std::string("Failed to open file '") + alignment_files[k] + "'"); delete ix; return nullptr; }
    ix->files.push_back(std::move(af));
  }
  std::map<uint32_t, std::vector<IvNode>> all;
  for (auto &kv : forest) {
    if (kv.second > data.size()) { set_err("Failed to deserialize tree"); delete ix; return nullptr; }
    Dec t{(const unsigned char *)data.data() + kv.second, (const unsigned char *)data.data() + data.size()};
    const uint32_t loaded = (uint32_t)t.u();
    if (loaded != kv.first) { set_err("Tree mismatch"); delete ix; return nullptr; } /* :1737-1739 */
    const uint64_t cnt = t.u();
    std::vector<IvNode> &nodes = all[kv.first];
    for (uint64_t k = 0; k < cnt && t.ok; k++) {
      IvNode nd;
      nd.first = (int32_t)t.i(); nd.last = (int32_t)t.i();
      QueryMetadata &m = nd.metadata;
      m.query_id = (uint32_t)t.u(); m.target_start = (int32_t)t.i(); m.target_end = (int32_t)t.i();
      m.query_start = (int32_t)t.i(); m.query_end = (int32_t)t.i(); m.alignment_file_index = (uint32_t)t.u();
      m.strand_and_data_offset = t.u(); m.data_bytes = t.u();
      if (m.alignment_file_index >= (uint32_t)n_files) { set_err("index names an alignment file that was not given"); delete ix; return nullptr; }
      nodes.push_back(nd);
      ix->n_records += !m.is_reversed();
    }
    if (!t.ok) { set_err("Failed to deserialize tree"); delete ix; return nullptr; }
  }
  build_trees(all, ix->trees);
  ix->preparse = false;
  (void)preparse;
  return ix;
}

void oracle_index_free(oracle_index_t *ix) { delete ix; }

uint32_t oracle_num_seqs(const oracle_index_t *ix) { return (uint32_t)ix->seq_index.id_to_name.size(); }
const char *oracle_seq_name(const oracle_index_t *ix, uint32_t id) {
  return id < ix->seq_index.id_to_name.size() ? ix->seq_index.id_to_name[id].c_str() : nullptr;
}
int64_t oracle_seq_len(const oracle_index_t *ix, uint32_t id) {
  return id < ix->seq_index.id_to_len.size() ? ix->seq_index.id_to_len[id] : -1;
}
int64_t oracle_seq_id(const oracle_index_t *ix, const char *name) {
  auto it = ix->seq_index.name_to_id.find(name);
  return it == ix->seq_index.name_to_id.end() ? -1 : (int64_t)it->second;
}
size_t oracle_num_records(const oracle_index_t *ix) { return ix->n_records; }
size_t oracle_num_targets(const oracle_index_t *ix) { return ix->trees.size(); }
size_t oracle_target_entries(const oracle_index_t *ix, uint32_t target_id, int32_t *out, size_t cap) {
  auto it = ix->trees.find(target_id);
  if (it == ix->trees.end()) return 0;
  const auto &nodes = it->second.nodes;
  for (size_t i = 0; i < nodes.size() && i < cap; i++) {
    out[4 * i] = nodes[i].first; out[4 * i + 1] = nodes[i].last;
    out[4 * i + 2] = (int32_t)nodes[i].metadata.query_id;
    out[4 * i + 3] = (nodes[i].metadata.strand_reverse() ? 1 : 0) | (nodes[i].metadata.is_reversed() ? 2 : 0);
  }
  return nodes.size();
}

long oracle_query(const oracle_index_t *ix, uint32_t target_id, int32_t start, int32_t end, const oracle_params_t *p,
                  oracle_interval_t *out, size_t cap) {
  std::vector<AdjustedInterval> results;
  g_nproj = 0;
  if (!run_query(*ix, target_id, start, end, *p, 1, results)) return -1;
  for (size_t i = 0; i < results.size() && i < cap; i++)
    out[i] = {results[i].q_id, results[i].q_first, results[i].q_last, results[i].t_id, results[i].t_first, results[i].t_last};
  return (long)results.size();
}

long oracle_query_masked(const oracle_index_t *ix, uint32_t target_id, int32_t start, int32_t end, const oracle_params_t *p,
                         uint32_t n_mask, const uint32_t *mask_seq, const int32_t *mask_seq_len, const uint64_t *mask_off,
                         const int32_t *mask_ranges, oracle_interval_t *out, size_t cap) {
  return oracle_query_filtered(ix, target_id, start, end, p, 1, n_mask, mask_seq, mask_seq_len, mask_off, mask_ranges, nullptr, out, cap);
}

long oracle_query_filtered(const oracle_index_t *ix, uint32_t target_id, int32_t start, int32_t end, const oracle_params_t *p,
                           int has_mask, uint32_t n_mask, const uint32_t *mask_seq, const int32_t *mask_seq_len,
                           const uint64_t *mask_off, const int32_t *mask_ranges, const uint8_t *subset_keep,
                           oracle_interval_t *out, size_t cap) {
  MaskMap mask;
  if (!has_mask) n_mask = 0;
  for (uint32_t i = 0; i < n_mask; i++) {
    SortedRanges sr;
    sr.sequence_length = mask_seq_len[i];
    sr.min_distance = 0; /* partition.rs:250-256 builds its masks with min_distance 0 */
    for (uint64_t k = mask_off[i]; k < mask_off[i + 1]; k++) sr.ranges.push_back({mask_ranges[2 * k], mask_ranges[2 * k + 1]});
    mask.emplace(mask_seq[i], std::move(sr));
  }
  std::vector<AdjustedInterval> results;
  g_nproj = 0;
  if (has_mask && !p->transitive) return -2; /* only the transitive queries take masked_regions */
  if (!run_query(*ix, target_id, start, end, *p, 1, results, has_mask ? &mask : nullptr, subset_keep)) return -1;
  for (size_t i = 0; i < results.size() && i < cap; i++)
    out[i] = {results[i].q_id, results[i].q_first, results[i].q_last, results[i].t_id, results[i].t_first, results[i].t_last};
  return (long)results.size();
}

/* ------------------------------------------------------------------------ */
/* SubsetFilter (subset_filter.rs:8-60, :117-176), transcribed                */
/* ------------------------------------------------------------------------ */
namespace {
struct SubsetFilter {
  std::set<std::string> exact, normalized, sample_ids;
  std::set<std::pair<std::string, std::string>> sample_haps;
};
std::string rust_trim(const std::string &s) { /* ASCII subset of str::trim */
  size_t a = 0, b = s.size();
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
  while (a < b && ws(s[a])) a++;
  while (b > a && ws(s[b - 1])) b--;
  return s.substr(a, b - a);
}
/* extract_sample_and_hap (:143-176): returns false for None; hap empty = None */
bool extract_sample_and_hap(const std::string &name, std::string &sample, std::string &hap) {
  hap.clear();
  size_t idx = name.find("_hap");
  if (idx != std::string::npos) {
    sample = name.substr(0, idx);
    for (size_t i = idx + 4; i < name.size() && isdigit((unsigned char)name[i]); i++) hap.push_back(name[i]);
    return true;
  }
  size_t h = name.find('#');
  if (h != std::string::npos) { /* split_once('#') */
    sample = name.substr(0, h);
    std::string rest = name.substr(h + 1);
    std::string hap_fragment = rest.substr(0, rest.find('#'));
    for (char c : hap_fragment) { if (!isdigit((unsigned char)c)) break; hap.push_back(c); }
    return true;
  }
  if (name.find(':') == std::string::npos && !rust_trim(name).empty()) { sample = name; return true; }
  return false;
}
SubsetFilter parse_subset_filter(const std::string &contents) { /* :117-141 */
  SubsetFilter f;
  size_t p = 0;
  while (p <= contents.size()) {
    size_t e = contents.find('\n', p);
    if (e == std::string::npos) e = contents.size();
    std::string line = contents.substr(p, e - p);
    p = e + 1;
    std::string trimmed = rust_trim(line);
    if (trimmed.empty() || trimmed[0] == '#') { if (e == contents.size()) break; continue; }
    f.exact.insert(trimmed);
    std::string no_coords = trimmed.substr(0, trimmed.find(':'));
    f.normalized.insert(no_coords);
    std::string sample, hap;
    if (extract_sample_and_hap(no_coords, sample, hap)) {
      if (!hap.empty()) f.sample_haps.insert({sample, hap});
      else f.sample_ids.insert(sample);
    }
    if (e == contents.size()) break;
  }
  return f;
}
bool matches_sample_keys(const SubsetFilter &f, const std::string &seq_name) { /* :45-59 */
  std::string sample, hap;
  if (extract_sample_and_hap(seq_name, sample, hap)) {
    if (!hap.empty() && f.sample_haps.count({sample, hap})) return true;
    if (f.sample_ids.count(sample)) return true;
  }
  return false;
}
bool subset_matches(const SubsetFilter &f, const std::string &seq_name) { /* :23-43 */
  if (f.exact.count(seq_name)) return true;
  std::string no_coords = seq_name.substr(0, seq_name.find(':'));
  if (seq_name != no_coords && f.exact.count(no_coords)) return true;
  if (f.normalized.count(no_coords)) return true;
  if (matches_sample_keys(f, no_coords)) return true;
  return matches_sample_keys(f, seq_name);
}
} // namespace

long oracle_subset_matches(const char *list_text, const char *const *names, size_t n, uint8_t *out) {
  SubsetFilter f = parse_subset_filter(list_text);
  for (size_t i = 0; i < n; i++) out[i] = subset_matches(f, names[i]) ? 1 : 0;
  return (long)f.exact.size(); /* entry_count (:19-21) */
}

long oracle_query_cigar(const oracle_index_t *ix, uint32_t target_id, int32_t start, int32_t end, const oracle_params_t *p,
                        oracle_interval_t *out, size_t cap, uint64_t *cigar_off, uint32_t *cigar_ops, size_t ops_cap,
                        uint64_t *n_ops) {
  std::vector<AdjustedInterval> results;
  g_nproj = 0;
  oracle_params_t q = *p;
  q.store_cigar = 1;
  if (!run_query(*ix, target_id, start, end, q, 1, results)) return -1;
  uint64_t k = 0;
  for (size_t i = 0; i < results.size(); i++) {
    if (i < cap) {
      out[i] = {results[i].q_id, results[i].q_first, results[i].q_last, results[i].t_id, results[i].t_first, results[i].t_last};
      cigar_off[i] = k;
    }
    for (uint32_t v : results[i].cigar) {
      if (i < cap && k < ops_cap) cigar_ops[k] = v;
      k++;
    }
    if (i < cap) cigar_off[i + 1] = k;
  }
  *n_ops = k;
  return (long)results.size();
}

/* ------------------------------------------------------------------------ */
/* PAF / BEDPE output (main.rs:11894-12103, :12563-12845, :13014-13180)       */
/* ------------------------------------------------------------------------ */
namespace {
inline uint32_t cigar_make(int32_t len, char op) { /* CigarOp::new (impg.rs:80-93) on a known-valid op letter */
  uint32_t v = 0;
  cigar_new(len, op, &v);
  return v;
}
/* merge_consecutive_cigar_ops (main.rs:13014-13035) */
void merge_consecutive_cigar_ops(std::vector<uint32_t> &cigar) {
  if (cigar.size() <= 1) return;
  size_t write_idx = 0;
  for (size_t read_idx = 1; read_idx < cigar.size(); read_idx++) {
    if (cigar_op(cigar[write_idx]) == cigar_op(cigar[read_idx])) {
      int32_t combined = cigar_len(cigar[write_idx]) + cigar_len(cigar[read_idx]);
      cigar[write_idx] = cigar_make(combined, cigar_op(cigar[write_idx]));
    } else {
      write_idx += 1;
      if (write_idx != read_idx) cigar[write_idx] = cigar[read_idx];
    }
  }
  cigar.resize(write_idx + 1);
}
/* extract_cigar_suffix (main.rs:13054-13090) */
std::vector<uint32_t> extract_cigar_suffix(const std::vector<uint32_t> &cigar, int32_t query_len, bool forward) {
  std::vector<uint32_t> result;
  int32_t remaining_query = query_len;
  for (size_t k = cigar.size(); k-- > 0;) {
    uint32_t op = cigar[k];
    if (remaining_query <= 0) break;
    int32_t qd = std::abs(query_delta(op, !forward));
    if (qd <= remaining_query) {
      result.push_back(op);
      remaining_query -= qd;
    } else if (qd > 0) {
      volatile float scale = (float)remaining_query / (float)qd;
      volatile float scaled = (float)cigar_len(op) * scale;
      result.push_back(cigar_make((int32_t)scaled, cigar_op(op)));
      remaining_query = 0;
    }
  }
  std::reverse(result.begin(), result.end());
  return result;
}
/* extract_cigar_prefix (main.rs:13092-13125) */
std::vector<uint32_t> extract_cigar_prefix(const std::vector<uint32_t> &cigar, int32_t query_len, bool forward) {
  std::vector<uint32_t> result;
  int32_t remaining_query = query_len;
  for (uint32_t op : cigar) {
    if (remaining_query <= 0) break;
    int32_t qd = std::abs(query_delta(op, !forward));
    if (qd <= remaining_query) {
      result.push_back(op);
      remaining_query -= qd;
    } else if (qd > 0) {
      volatile float scale = (float)remaining_query / (float)qd;
      volatile float scaled = (float)cigar_len(op) * scale;
      result.push_back(cigar_make((int32_t)scaled, cigar_op(op)));
      remaining_query = 0;
    }
  }
  return result;
}
/* check_cigar_overlap_match (main.rs:13037-13052) */
bool check_cigar_overlap_match(const std::vector<uint32_t> &cur, const std::vector<uint32_t> &next, int32_t qlen, bool fwd) {
  return extract_cigar_suffix(cur, qlen, fwd) == extract_cigar_prefix(next, qlen, fwd);
}
/* trim_cigar_prefix (main.rs:13127-13180) */
std::vector<uint32_t> trim_cigar_prefix(const std::vector<uint32_t> &cigar, int32_t query_len, int32_t target_len) {
  std::vector<uint32_t> result;
  int32_t query_consumed = 0, target_consumed = 0;
  size_t start_idx = 0;
  for (size_t idx = 0; idx < cigar.size(); idx++) {
    uint32_t op = cigar[idx];
    int32_t q_delta = std::abs(query_delta(op, false)), t_delta = target_delta(op);
    if (query_consumed + q_delta > query_len || target_consumed + t_delta > target_len) {
      int32_t query_remaining = query_len - query_consumed, target_remaining = target_len - target_consumed;
      volatile float skip_ratio;
      if (q_delta > 0 && t_delta > 0) {
        volatile float a = (float)query_remaining / (float)q_delta, b = (float)target_remaining / (float)t_delta;
        skip_ratio = a < b ? a : b; /* f32::min (no NaNs here) */
      } else if (q_delta > 0) skip_ratio = (float)query_remaining / (float)q_delta;
      else if (t_delta > 0) skip_ratio = (float)target_remaining / (float)t_delta;
      else skip_ratio = 0.0f;
      volatile float sk = (float)cigar_len(op) * skip_ratio;
      int32_t skip_len = (int32_t)sk;
      if (skip_len < cigar_len(op)) result.push_back(cigar_make(cigar_len(op) - skip_len, cigar_op(op)));
      start_idx = idx + 1;
      break;
    }
    query_consumed += q_delta;
    target_consumed += t_delta;
    if (query_consumed >= query_len && target_consumed >= target_len) {
      start_idx = idx + 1;
      break;
    }
  }
  result.insert(result.end(), cigar.begin() + (long)start_idx, cigar.end());
  return result;
}
/* merge_adjusted_intervals (main.rs:12563-12845) */
bool merge_adjusted_intervals(std::vector<AdjustedInterval> &results, int32_t merge_distance) {
  if (!(results.size() > 1 && merge_distance >= 0)) return true;
  std::stable_sort(results.begin(), results.end(), [](const AdjustedInterval &a, const AdjustedInterval &b) {
    bool af = a.q_first < a.q_last, bf = b.q_first < b.q_last; /* strict here (:12567) */
    auto ka = std::make_tuple(a.q_id, af, af ? a.q_first : a.q_last, a.t_id, a.t_first);
    auto kb = std::make_tuple(b.q_id, bf, bf ? b.q_first : b.q_last, b.t_id, b.t_first);
    return ka < kb;
  });
  std::vector<AdjustedInterval> merged;
  AdjustedInterval cur = std::move(results[0]);
  for (size_t i = 1; i < results.size(); i++) {
    AdjustedInterval next = std::move(results[i]);
    bool query_forward = cur.q_first <= cur.q_last, next_query_forward = next.q_first <= next.q_last;
    bool target_forward = cur.t_first <= cur.t_last, next_target_forward = next.t_first <= next.t_last;
    if (!target_forward || !next_target_forward) { set_err("Target intervals should always be in forward!"); return false; }
    if (cur.q_id != next.q_id || cur.t_id != next.t_id || query_forward != next_query_forward) {
      merged.push_back(std::move(cur));
      cur = std::move(next);
      continue;
    }
    bool q_contig, t_contig, q_overlap, t_overlap;
    if (query_forward) {
      q_contig = cur.q_last == next.q_first; t_contig = cur.t_last == next.t_first;
      q_overlap = cur.q_last > next.q_first; t_overlap = cur.t_last > next.t_first;
    } else {
      q_contig = cur.q_first == next.q_last; t_contig = cur.t_first == next.t_last;
      q_overlap = cur.q_first > next.q_last; t_overlap = cur.t_first < next.t_last;
    }
    if (q_contig && t_contig) {
      if (query_forward) {
        cur.q_last = next.q_last; cur.t_last = next.t_last;
        cur.cigar.insert(cur.cigar.end(), next.cigar.begin(), next.cigar.end());
      } else {
        cur.q_first = next.q_first; cur.t_first = next.t_first;
        std::vector<uint32_t> nc(next.cigar);
        nc.insert(nc.end(), cur.cigar.begin(), cur.cigar.end());
        cur.cigar.swap(nc);
      }
      merge_consecutive_cigar_ops(cur.cigar);
      continue;
    }
    if (q_overlap && t_overlap) {
      int32_t qol, tol;
      if (query_forward) { qol = next.q_first - cur.q_last; tol = next.t_first - cur.t_last; }
      else { qol = next.q_last - cur.q_first; tol = cur.t_first - next.t_last; }
      if (qol > 0 && tol > 0) {
        if (check_cigar_overlap_match(cur.cigar, next.cigar, qol, query_forward)) {
          std::vector<uint32_t> trimmed = trim_cigar_prefix(next.cigar, qol, tol);
          if (query_forward) {
            cur.q_last = next.q_last; cur.t_last = next.t_last;
            cur.cigar.insert(cur.cigar.end(), trimmed.begin(), trimmed.end());
          } else {
            cur.q_first = next.q_first; cur.t_first = next.t_first;
            trimmed.insert(trimmed.end(), cur.cigar.begin(), cur.cigar.end());
            cur.cigar.swap(trimmed);
          }
          continue;
        }
      }
    }
    if (!q_overlap && !t_overlap) {
      int32_t query_gap, target_gap;
      if (query_forward) { query_gap = next.q_first - cur.q_last; target_gap = next.t_first - cur.t_last; }
      else { query_gap = cur.q_first - next.q_last; target_gap = cur.t_first - next.t_last; }
      if (query_gap >= 0 && target_gap >= 0 && (query_gap > 0 || target_gap > 0) && query_gap <= merge_distance &&
          target_gap <= merge_distance) {
        std::vector<uint32_t> gap;
        if (query_gap > 0) gap.push_back(cigar_make(query_gap, 'I'));
        if (target_gap > 0) gap.push_back(cigar_make(target_gap, 'D'));
        if (query_forward) {
          cur.q_last = next.q_last; cur.t_last = next.t_last;
          cur.cigar.insert(cur.cigar.end(), gap.begin(), gap.end());
          cur.cigar.insert(cur.cigar.end(), next.cigar.begin(), next.cigar.end());
        } else {
          cur.q_first = next.q_first; cur.t_first = next.t_first;
          std::vector<uint32_t> nc(next.cigar);
          nc.insert(nc.end(), gap.begin(), gap.end());
          nc.insert(nc.end(), cur.cigar.begin(), cur.cigar.end());
          cur.cigar.swap(nc);
        }
        merge_consecutive_cigar_ops(cur.cigar);
        continue;
      }
    }
    merged.push_back(std::move(cur));
    cur = std::move(next);
  }
  merged.push_back(std::move(cur));
  results = std::move(merged);
  return true;
}
/* format!("{x:.6}").trim_end_matches('0').trim_end_matches('.') for an f32 (main.rs:11960-11967) */
std::string fmt_f32_trim(float x) {
  char b[64];
  if (x != x) return "NaN";
  if (x == std::numeric_limits<float>::infinity()) return "inf";
  if (x == -std::numeric_limits<float>::infinity()) return "-inf";
  snprintf(b, sizeof b, "%.6f", (double)x);
  std::string s(b);
  while (!s.empty() && s.back() == '0') s.pop_back();
  while (!s.empty() && s.back() == '.') s.pop_back();
  return s;
}
} // namespace

/* perform_query + results.remove(0) + output_results_paf / output_results_bedpe
 * (main.rs:7472-7496, :11894-12103).  format: 0 = PAF, 1 = BEDPE. */
/* parse_subsequence_coordinates (main.rs:4642-4659) and transform_coordinates_to_original (:4662-4678) */
static bool parse_subsequence_coordinates(const std::string &seq_name, std::string &base_name, int32_t &start_offset) {
  size_t colon_pos = seq_name.rfind(':');
  if (colon_pos == std::string::npos) return false;
  std::string range_part = seq_name.substr(colon_pos + 1);
  size_t dash_pos = range_part.find('-');
  if (dash_pos == std::string::npos) return false;
  std::string start_str = range_part.substr(0, dash_pos);
  /* str::parse::<i32>: optional sign, at least one digit, no overflow */
  size_t i = 0;
  bool neg = false;
  if (i < start_str.size() && (start_str[i] == '+' || start_str[i] == '-')) { neg = start_str[i] == '-'; i++; }
  if (i >= start_str.size()) return false;
  int64_t v = 0;
  for (; i < start_str.size(); i++) {
    if (start_str[i] < '0' || start_str[i] > '9') return false;
    v = v * 10 + (start_str[i] - '0');
    if (v > 2147483648ll) return false;
  }
  if (neg) v = -v;
  if (v > 2147483647ll || v < -2147483648ll) return false;
  base_name = seq_name.substr(0, colon_pos);
  start_offset = (int32_t)v;
  return true;
}
static void transform_coordinates_to_original(const std::string &seq_name, uint32_t start, uint32_t end, bool original_coordinates,
                                              std::string &name_out, uint32_t &start_out, uint32_t &end_out) {
  std::string base; int32_t offset = 0;
  if (original_coordinates && parse_subsequence_coordinates(seq_name, base, offset)) {
    name_out = base; start_out = start + (uint32_t)offset; end_out = end + (uint32_t)offset;
  } else { name_out = seq_name; start_out = start; end_out = end; }
}
long oracle_parse_subsequence(const char *seq_name, char *base_out, size_t cap, int32_t *offset) {
  std::string base; int32_t off = 0;
  if (!parse_subsequence_coordinates(seq_name, base, off)) return 0;
  if (base.size() + 1 > cap) return -1;
  memcpy(base_out, base.c_str(), base.size() + 1);
  *offset = off;
  return 1;
}

int oracle_query_paf(const oracle_index_t *ix, const char *target_name, int32_t start, int32_t end,
                     const char *range_name, const oracle_params_t *p, int32_t merge_distance, int format,
                     char **buf, size_t *len, size_t *cap) {
  auto it = ix->seq_index.name_to_id.find(target_name);
  if (it == ix->seq_index.name_to_id.end()) { set_err(std::string("Sequence '") + target_name + "' not found in index"); return -2; }
  uint32_t target_id = it->second;
  int64_t seq_len = ix->seq_index.id_to_len[target_id];
  if (start < 0 || end < 0 || start >= end || end > (int32_t)seq_len) { set_err("invalid range"); return -3; }
  if (end - start < p->min_transitive_len) { set_err("Range is below minimum length"); return -4; }
  std::vector<AdjustedInterval> results;
  g_nproj = 0;
  oracle_params_t q = *p;
  q.store_cigar = 1; /* main.rs:7447 */
  if (!run_query(*ix, target_id, start, end, q, 1, results)) return -1;
  if (!p->transitive && p->min_output_length >= 0) { /* :11682-11688 retain */
    std::vector<AdjustedInterval> kept;
    for (auto &x : results) if (std::abs(x.q_last - x.q_first) >= p->min_output_length) kept.push_back(std::move(x));
    results.swap(kept);
  }
  if (results.empty()) { set_err("removal index (is 0) should be < len (is 0)"); return -5; } /* Vec::remove(0) panics */
  results.erase(results.begin());
  if (format == 1) { /* output_results_bedpe: a row without CIGAR would switch to the gap-2d merge (:11905-11910) */
    bool any_empty = false;
    for (auto &r : results) any_empty = any_empty || r.cigar.empty();
    if (any_empty) { set_err("empty CIGAR in a BEDPE row (syng output) is outside the restated path"); return -6; }
  }
  if (!merge_adjusted_intervals(results, merge_distance)) return -7;
  for (auto &r : results) {
    const std::string &qn = ix->seq_index.id_to_name[r.q_id], &tn = ix->seq_index.id_to_name[r.t_id];
    int32_t first, last; char strand;
    if (r.q_first <= r.q_last) { first = r.q_first; last = r.q_last; strand = '+'; }
    else { first = r.q_last; last = r.q_first; strand = '-'; }
    int32_t matches = 0, mismatches = 0, insertions = 0, inserted_bp = 0, deletions = 0, deleted_bp = 0, block_len = 0;
    for (uint32_t op : r.cigar) {
      int32_t l = cigar_len(op);
      switch (cigar_op(op)) {
      case 'M': case '=': matches += l; block_len += l; break;
      case 'X': mismatches += l; block_len += l; break;
      case 'I': insertions += 1; inserted_bp += l; block_len += l; break;
      case 'D': deletions += 1; deleted_bp += l; block_len += l; break;
      default: break;
      }
    }
    volatile float gi = (float)matches / (float)(matches + mismatches + insertions + deletions);
    int32_t edit_distance = mismatches + inserted_bp + deleted_bp;
    volatile float bi = (float)matches / (float)(matches + edit_distance);
    std::string gi_s = fmt_f32_trim(gi), bi_s = fmt_f32_trim(bi);
    std::string line;
    char num[64];
    const bool orig = p->original_sequence_coordinates != 0;
    std::string tqn, ttn; uint32_t tf, tl, ttf, ttl; /* :11925-11938, :12014-12027 */
    transform_coordinates_to_original(qn, (uint32_t)first, (uint32_t)last, orig, tqn, tf, tl);
    transform_coordinates_to_original(tn, (uint32_t)r.t_first, (uint32_t)r.t_last, orig, ttn, ttf, ttl);
    /* :12030-12046: with original coordinates the lengths come from the sequence files; none are given here,
     * which the reference answers with a warning and 0 (get_original_sequence_length, :4681-4704) */
    unsigned long long qlen = orig ? 0ull : (unsigned long long)ix->seq_index.id_to_len[r.q_id];
    unsigned long long tlen = orig ? 0ull : (unsigned long long)ix->seq_index.id_to_len[r.t_id];
    if (format == 1) {
      line += tqn; snprintf(num, sizeof num, "\t%u\t%u\t", tf, tl); line += num;
      line += ttn; snprintf(num, sizeof num, "\t%u\t%u\t", ttf, ttl); line += num;
      line += range_name; line += "\t0\t"; line += strand; line += "\t+\tgi:f:"; line += gi_s; line += "\tbi:f:"; line += bi_s;
      line += "\n";
    } else {
      line += tqn; snprintf(num, sizeof num, "\t%llu\t%u\t%u\t%c\t", qlen, tf, tl, strand); line += num;
      line += ttn; snprintf(num, sizeof num, "\t%llu\t%u\t%u\t%d\t%d\t255\tgi:f:", tlen, ttf, ttl, matches, block_len); line += num;
      line += gi_s; line += "\tbi:f:"; line += bi_s; line += "\tcg:Z:";
      for (uint32_t op : r.cigar) { snprintf(num, sizeof num, "%d%c", cigar_len(op), cigar_op(op)); line += num; }
      line += "\tan:Z:"; line += range_name; line += "\n";
    }
    append(buf, len, cap, line.data(), line.size());
  }
  return 0;
}

long oracle_bed_merge(oracle_interval_t *iv, size_t n, int32_t merge_distance, int merge_strands) {
  std::vector<oracle_interval_t> v(iv, iv + n);
  /* output_results_bed (main.rs:11858-11866): any_empty_cigar is true for BED
   * because store_cigar=false (main.rs:7447) */
  merge_adjusted_intervals_gap_2d(v, merge_distance);
  merge_query_adjusted_intervals(v, merge_distance, merge_strands != 0);
  std::copy(v.begin(), v.end(), iv);
  return (long)v.size();
}

/* merge_query_adjusted_intervals alone (main.rs:12474-12560): what the reference's own test
 * test_syng_gfa_intervals_are_merged_before_graph_build (main.rs:13655-13698) drives */
long oracle_merge_query(oracle_interval_t *iv, size_t n, int32_t merge_distance, int merge_strands) {
  std::vector<oracle_interval_t> v(iv, iv + n);
  merge_query_adjusted_intervals(v, merge_distance, merge_strands != 0);
  std::copy(v.begin(), v.end(), iv);
  return (long)v.size();
}

int oracle_query_bed(const oracle_index_t *ix, const char *target_name, int32_t start, int32_t end,
                     const char *range_name, const oracle_params_t *p, int32_t merge_distance,
                     char **buf, size_t *len, size_t *cap) {
  /* validate_sequence_range (main.rs:10458-10520) */
  auto it = ix->seq_index.name_to_id.find(target_name);
  if (it == ix->seq_index.name_to_id.end()) { set_err(std::string("Sequence '") + target_name + "' not found in index"); return -2; }
  uint32_t target_id = it->second;
  int64_t seq_len = ix->seq_index.id_to_len[target_id];
  if (start < 0 || end < 0 || start >= end || end > (int32_t)seq_len) { set_err("invalid range"); return -3; }
  /* validate_range_min_length (main.rs:10387-10403) */
  if (end - start < p->min_transitive_len) { set_err("Range is below minimum length"); return -4; }
  /* perform_query (main.rs:11605-11707) */
  std::vector<AdjustedInterval> results;
  g_nproj = 0;
  oracle_params_t q = *p;
  q.store_cigar = 0; /* BED: main.rs:7447 */
  if (!run_query(*ix, target_id, start, end, q, 1, results)) return -1;
  std::vector<oracle_interval_t> v;
  for (auto &r : results) v.push_back({r.q_id, r.q_first, r.q_last, r.t_id, r.t_first, r.t_last});
  if (!p->transitive && p->min_output_length >= 0) { /* :11682-11688 retain */
    std::vector<oracle_interval_t> kept;
    for (auto &x : v) if (std::abs(x.q_last - x.q_first) >= p->min_output_length) kept.push_back(x);
    v.swap(kept);
  }
  /* output_results_bed (main.rs:11849-11892), merge_strands_for_output("bed") = true */
  merge_adjusted_intervals_gap_2d(v, merge_distance);
  merge_query_adjusted_intervals(v, merge_distance, !p->consider_strandness); /* merge_strands_for_output("bed"), main.rs:4395-4409 */
  for (auto &x : v) {
    const std::string &qn = ix->seq_index.id_to_name[x.query_id];
    int32_t first, last; char strand;
    if (x.q_first <= x.q_last) { first = x.q_first; last = x.q_last; strand = '+'; }
    else { first = x.q_last; last = x.q_first; strand = '-'; }
    char line[64];
    std::string tqn; uint32_t tf, tl; /* :11876-11883 */
    transform_coordinates_to_original(qn, (uint32_t)first, (uint32_t)last, p->original_sequence_coordinates != 0, tqn, tf, tl);
    append(buf, len, cap, tqn.data(), tqn.size());
    int n = snprintf(line, sizeof line, "\t%u\t%u\t", tf, tl);
    append(buf, len, cap, line, (size_t)n);
    append(buf, len, cap, range_name, strlen(range_name));
    n = snprintf(line, sizeof line, "\t.\t%c\n", strand);
    append(buf, len, cap, line, (size_t)n);
  }
  return 0;
}

/* parse_range (partition.rs:1765-1789) */
static int parse_range_parts(const char *a, size_t an, const char *b, size_t bn, int32_t *s, int32_t *e) {
  if (!parse_i32(a, an, s)) return -1;
  if (!parse_i32(b, bn, e)) return -2;
  if (*s >= *e) return -3;
  return 0;
}
/* parse_target_range (partition.rs:1752-1763) */
int oracle_parse_target_range(const char *str, char *name_out, size_t name_cap, int32_t *start, int32_t *end) {
  const char *colon = strrchr(str, ':'); /* rsplitn(2, ':') */
  if (!colon) return -1;
  const char *range = colon + 1;
  /* parts[0].split('-') must give exactly 2 parts */
  size_t rl = strlen(range), ndash = 0, dpos = 0;
  for (size_t i = 0; i < rl; i++) if (range[i] == '-') { ndash++; dpos = i; }
  if (ndash != 1) return -2;
  if (parse_range_parts(range, dpos, range + dpos + 1, rl - dpos - 1, start, end) != 0) return -3;
  size_t nl = (size_t)(colon - str);
  if (nl + 1 > name_cap) return -4;
  memcpy(name_out, str, nl); name_out[nl] = 0;
  return 0;
}
/* parse_bed_file (partition.rs:1719-1750).  names/rnames are '\n'-joined. */
long oracle_parse_bed_text(const char *text, size_t len, char *names_out, size_t names_cap, int32_t *se_out,
                           char *rnames_out, size_t rnames_cap, size_t cap) {
  size_t pos = 0, count = 0, no = 0, ro = 0;
  while (pos < len) {
    size_t eol = pos;
    while (eol < len && text[eol] != '\n') eol++;
    size_t l = eol - pos;
    if (l > 0 && text[pos + l - 1] == '\r') l--;
    std::vector<std::pair<const char *, size_t>> parts;
    size_t st = 0;
    const char *line = text + pos;
    for (size_t i = 0; i <= l; i++) if (i == l || line[i] == '\t') { parts.push_back({line + st, i - st}); st = i + 1; }
    if (parts.size() < 3) return -1;
    int32_t s, e;
    if (parse_range_parts(parts[1].first, parts[1].second, parts[2].first, parts[2].second, &s, &e) != 0) return -2;
    std::string name;
    bool have = false;
    if (parts.size() > 3) {
      std::string t(parts[3].first, parts[3].second);
      size_t a = 0, b = t.size();
      while (a < b && isspace((unsigned char)t[a])) a++;
      while (b > a && isspace((unsigned char)t[b - 1])) b--;
      t = t.substr(a, b - a);
      if (!t.empty() && t != ".") { name = t; have = true; }
    }
    std::string chrom(parts[0].first, parts[0].second);
    if (!have) name = chrom + ":" + std::to_string(s) + "-" + std::to_string(e);
    if (count < cap) {
      if (no + chrom.size() + 1 > names_cap || ro + name.size() + 1 > rnames_cap) return -3;
      memcpy(names_out + no, chrom.data(), chrom.size()); no += chrom.size(); names_out[no++] = '\n';
      memcpy(rnames_out + ro, name.data(), name.size()); ro += name.size(); rnames_out[ro++] = '\n';
      se_out[2 * count] = s; se_out[2 * count + 1] = e;
    }
    count++;
    pos = eol + 1;
  }
  if (no < names_cap) names_out[no] = 0;
  if (ro < rnames_cap) rnames_out[ro] = 0;
  return (long)count;
}

int oracle_bench(const oracle_index_t *ix, const uint32_t *target_ids, const int32_t *starts, const int32_t *ends,
                 size_t n, const oracle_params_t *p, int threads, int mode, uint64_t *n_projected,
                 uint64_t *n_results, double *seconds) {
  std::atomic<uint64_t> proj{0}, nres{0};
  std::atomic<bool> ok{true};
  auto t0 = std::chrono::steady_clock::now();
  if (mode == 0) { /* main.rs:7435 serial over ranges; rayon inside the BFS level */
    for (size_t i = 0; i < n; i++) {
      std::vector<AdjustedInterval> results;
      g_nproj = 0;
      if (!run_query(*ix, target_ids[i], starts[i], ends[i], *p, threads, results)) { ok = false; break; }
      proj += g_nproj; nres += results.size();
    }
  } else {
    parallel_for(n, threads, [&](size_t i) {
      std::vector<AdjustedInterval> results;
      g_nproj = 0;
      if (!run_query(*ix, target_ids[i], starts[i], ends[i], *p, 1, results)) ok = false;
      proj += g_nproj; nres += results.size();
    });
  }
  auto t1 = std::chrono::steady_clock::now();
  *n_projected = proj; *n_results = nres;
  *seconds = std::chrono::duration<double>(t1 - t0).count();
  return ok ? 0 : -1;
}

} // extern "C"
