// BED post-processing of query results: the product-side implementation of
// merge_adjusted_intervals_gap_2d (src/main.rs:12858-13011),
// merge_query_adjusted_intervals (src/main.rs:12474-12560) and
// output_results_bed (src/main.rs:11849-11892).  Independent of the oracle.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <numeric>
#include <thread>

#include "impg_internal.hpp"

namespace impg {

namespace {

struct Dsu {
  std::vector<uint32_t> p;
  explicit Dsu(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
  }
  void unite(uint32_t a, uint32_t b) {
    a = find(a); b = find(b);
    if (a != b) p[a] = b;
  }
};

inline bool fwd(const impg_gpu_interval_t &x) { return x.q_first <= x.q_last; }

// 2-D gap merge: chains of results on the same (query seq, target seq, strand)
// whose query gap and target gap are both <= d and that progress forward on the
// target are collapsed to their bounding box.  Output keeps one interval per
// chain, ordered by the chain's smallest original index.
void gap_2d(std::vector<impg_gpu_interval_t> &r, int32_t merge_distance) {
  const size_t n = r.size();
  if (n <= 1 || merge_distance < 0) return;
  const int64_t d = merge_distance;
  // order indices by group, then by the in-group sort key (q.first ascending on
  // '+', descending on '-'), ties by original index (= stable)
  std::vector<uint32_t> ord(n);
  std::iota(ord.begin(), ord.end(), 0u);
  auto gkey = [&](uint32_t i) { return std::make_tuple(r[i].query_id, r[i].target_id, fwd(r[i])); };
  std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
    auto ka = gkey(a), kb = gkey(b);
    if (ka != kb) return ka < kb;
    int32_t sa = fwd(r[a]) ? r[a].q_first : -r[a].q_first, sb = fwd(r[b]) ? r[b].q_first : -r[b].q_first;
    if (sa != sb) return sa < sb;
    return a < b;
  });
  Dsu dsu(n);
  for (size_t g0 = 0; g0 < n;) {
    size_t g1 = g0 + 1;
    while (g1 < n && gkey(ord[g1]) == gkey(ord[g0])) g1++;
    const bool f = fwd(r[ord[g0]]);
    for (size_t a = g0; a < g1; a++) {
      const auto &A = r[ord[a]];
      const int64_t qa_start = f ? A.q_first : A.q_last, qa_end = f ? A.q_last : A.q_first;
      for (size_t b = a + 1; b < g1; b++) {
        const auto &B = r[ord[b]];
        const int64_t qb_start = f ? B.q_first : B.q_last;
        if (qb_start < qa_start) continue;        // strictly-backward start
        if (qb_start - qa_end > d) break;          // query gap too large: later b are not tried
        const int64_t t_gap = f ? (int64_t)B.t_first - A.t_last : (int64_t)A.t_first - B.t_last;
        const bool t_forward = f ? B.t_first > A.t_first : B.t_last < A.t_last;
        if (t_forward && t_gap <= d) dsu.unite(ord[a], ord[b]);
      }
    }
    g0 = g1;
  }
  // aggregate per component; the representative slot is the smallest member index
  std::vector<uint32_t> first_member(n, UINT32_MAX);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t c = dsu.find(i);
    if (first_member[c] == UINT32_MAX) first_member[c] = i;
  }
  // metadata comes from the member that sorts first in the component (stable by
  // q.first on '+', by -q.first on '-'); the box is order-independent
  std::vector<impg_gpu_interval_t> box(n);
  std::vector<uint8_t> init(n, 0);
  for (size_t k = 0; k < n; k++) {  // ord is already grouped and key-sorted
    uint32_t i = ord[k], c = dsu.find(i);
    const auto &x = r[i];
    if (!init[c]) { box[c] = x; init[c] = 1; continue; }
    auto &m = box[c];
    if (fwd(r[first_member[c]])) { m.q_first = std::min(m.q_first, x.q_first); m.q_last = std::max(m.q_last, x.q_last); }
    else { m.q_first = std::max(m.q_first, x.q_first); m.q_last = std::min(m.q_last, x.q_last); }
    m.t_first = std::min(m.t_first, x.t_first);
    m.t_last = std::max(m.t_last, x.t_last);
  }
  std::vector<impg_gpu_interval_t> out;
  out.reserve(n);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t c = dsu.find(i);
    if (first_member[c] == i) out.push_back(box[c]);
  }
  r.swap(out);
}

// query-axis merge: stable sort by (sequence, start, forward first), then sweep
void merge_query_axis(std::vector<impg_gpu_interval_t> &r, int32_t merge_distance, bool merge_strands) {
  if (!(r.size() > 1 && (merge_distance >= 0 || merge_strands))) return;
  std::stable_sort(r.begin(), r.end(), [](const impg_gpu_interval_t &a, const impg_gpu_interval_t &b) {
    if (a.query_id != b.query_id) return a.query_id < b.query_id;
    int32_t sa = std::min(a.q_first, a.q_last), sb = std::min(b.q_first, b.q_last);
    if (sa != sb) return sa < sb;
    return fwd(a) && !fwd(b);
  });
  size_t w = 0;
  for (size_t k = 1; k < r.size(); k++) {
    const impg_gpu_interval_t cur = r[w], nx = r[k];
    const bool cf = fwd(cur), nf = fwd(nx);
    const int32_t cs = std::min(cur.q_first, cur.q_last), ce = std::max(cur.q_first, cur.q_last);
    const int32_t ns = std::min(nx.q_first, nx.q_last), ne = std::max(nx.q_first, nx.q_last);
    const bool keep_apart = merge_distance < 0 || cur.query_id != nx.query_id || (!merge_strands && cf != nf) ||
                            ns > ce + merge_distance;
    if (keep_apart) {
      w++;
      if (w != k) std::swap(r[w], r[k]);
      continue;
    }
    const int32_t ms = std::min(cs, ns), me = std::max(ce, ne);
    bool mf = cf;  // across strands the longer span decides, ties keep the current one
    if (merge_strands && cf != nf) {
      int64_t cl = (int64_t)ce - cs, nl = (int64_t)ne - ns;
      if (nl > cl) mf = nf;
    }
    r[w].q_first = mf ? ms : me;
    r[w].q_last = mf ? me : ms;
  }
  r.resize(w + 1);
}

}  // namespace

size_t bed_merge(impg_gpu_interval_t *iv, size_t n, int32_t merge_distance, bool merge_strands) {
  std::vector<impg_gpu_interval_t> v(iv, iv + n);
  gap_2d(v, merge_distance);  // BED: every CIGAR is empty (main.rs:11858-11865)
  merge_query_axis(v, merge_distance, merge_strands);
  std::copy(v.begin(), v.end(), iv);
  return v.size();
}

void render_bed(const impg_gpu_results &res, const impg_gpu_index &ix, const char *const *range_names,
                const impg_gpu_params_t &p, int32_t merge_distance, std::vector<std::string> &parts) {
  const size_t nr = res.offsets.size() - 1;
  parts.assign(nr, std::string());
  std::atomic<size_t> next{0};
  unsigned hw = std::thread::hardware_concurrency();
  size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, nr / 16 + 1));
  auto work = [&]() {
    std::vector<impg_gpu_interval_t> v;
    char buf[64];
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= nr) break;
      v.assign(res.intervals.begin() + res.offsets[i], res.intervals.begin() + res.offsets[i + 1]);
      if (!p.transitive && p.min_output_length >= 0) {  // perform_query retain (main.rs:11682-11688)
        v.erase(std::remove_if(v.begin(), v.end(),
                               [&](const impg_gpu_interval_t &x) {
                                 return std::abs((int64_t)x.q_last - x.q_first) < p.min_output_length;
                               }),
                v.end());
      }
      gap_2d(v, merge_distance);
      merge_query_axis(v, merge_distance, !p.consider_strandness);  // merge_strands_for_output("bed") (main.rs:4395-4409)
      std::string &s = parts[i];
      std::string fallback;
      const char *rn = range_names ? range_names[i] : nullptr;
      for (const auto &x : v) {
        uint32_t shift = 0;  // --original-sequence-coordinates (main.rs:11876-11883)
        if (x.query_id < ix.seq.names.size()) shift = put_original_name(s, ix.seq.names[x.query_id], p.original_sequence_coordinates != 0);
        else s += std::to_string(x.query_id);
        int n = snprintf(buf, sizeof buf, "\t%u\t%u\t", (uint32_t)std::min(x.q_first, x.q_last) + shift,
                         (uint32_t)std::max(x.q_first, x.q_last) + shift);
        s.append(buf, (size_t)n);
        if (rn) s += rn;
        else {  // "{chrom}:{start}-{end}" (partition.rs:1741, :1762)
          const auto &q = res.ranges[i];
          if (q.target_id < ix.seq.names.size()) s += ix.seq.names[q.target_id];
          else s += std::to_string(q.target_id);
          n = snprintf(buf, sizeof buf, ":%d-%d", q.start, q.end);
          s.append(buf, (size_t)n);
        }
        s += "\t.\t";
        s += fwd(x) ? '+' : '-';
        s += '\n';
      }
    }
  };
  std::vector<std::thread> th;
  for (size_t t = 0; t < T; t++) th.emplace_back(work);
  for (auto &t : th) t.join();
}

// The ranges' texts -> one malloc'ed, NUL-terminated buffer.  Gigabytes for a transitive batch: the destination is
// written (and its pages first touched) by many threads, each copying a contiguous run of parts.
char *join_parts(std::vector<std::string> &parts, size_t *len) {
  std::vector<size_t> off(parts.size() + 1, 0);
  for (size_t i = 0; i < parts.size(); i++) off[i + 1] = off[i] + parts[i].size();
  const size_t total = off.back();
  char *p = (char *)malloc(total + 1);
  if (!p) throw Error{IMPG_E_OOM, "host out of memory"};
  unsigned hw = std::thread::hardware_concurrency();
  const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, total / (8u << 20) + 1));
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= parts.size()) break;
      if (!parts[i].empty()) memcpy(p + off[i], parts[i].data(), parts[i].size());
      std::string().swap(parts[i]);  // give the part's memory back as we go
    }
  };
  if (T == 1) work();
  else {
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
  }
  p[total] = 0;
  *len = total;
  return p;
}

}  // namespace impg
