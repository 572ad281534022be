// BED post-processing on the device: merge_adjusted_intervals_gap_2d (src/main.rs:12858-13011) followed by
// merge_query_adjusted_intervals (src/main.rs:12474-12560), as output_results_bed (src/main.rs:11849-11892) runs
// them for every query range -- but for all ranges of a chunk at once and on the hit slots where they lie, so that
// only merged rows cross PCIe.  bed.cpp holds the host-side implementation of the same two merges (one range at a
// time, for impg_gpu_results_bed / impg_gpu_bed_merge); the parity tests compare both with the CPU restatement of the reference.
//
//   rows      every emitted result of the chunk (self intervals, then level by level in slot order), numbered in
//             that order: within one range this is the reference's emission order, which all tie rules refer to
//   gap_2d    rows sorted by (range, query seq, target seq, strand | q.first ascending on '+', descending on '-' |
//             emission order): two stable radix sorts.  One thread per row replays the reference's inner loop over
//             the rows behind it in its group (skip strictly-backward starts, stop at the first query gap > d,
//             link when the target also moves forward by a gap <= d) and links with a lock-free union-find whose
//             root is always the smallest member, i.e. the chain's first row in emission order -- the slot the
//             reference writes the merged row to.  Boxes by atomic min / max into the root.
//   query axis rows sorted by (range, query seq | start | forward first | current order); the reference's sweep
//             merges a row into the running one while start <= running end + d.  Sorted by start, "running end" is
//             the prefix maximum of the ends inside the (range, sequence) segment: a run breaks exactly where
//             start > prefix max + d.  The merged strand is the strand of the last row whose own length exceeds the
//             span merged before it (else the run's first row's): main.rs:12535-12547 unrolled.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <functional>
#include <string>

#include "engine.hpp"

namespace impg {

namespace {

__device__ __forceinline__ uint32_t ord_u32(int32_t v) { return (uint32_t)v ^ 0x80000000u; }  // order-preserving

struct Rows {  // SoA rows of a chunk
  uint32_t *q, *qid, *tid;
  int4 *c;  // {q_first, q_last, t_first, t_last}
};

// ---- gap_2d --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_keys_kernel(Rows R, uint32_t n, unsigned seq_bits, uint32_t *__restrict__ k32,
                                                       unsigned long long *__restrict__ k64, uint32_t *__restrict__ idx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int4 c = R.c[i];
  const bool f = c.x <= c.y;
  k32[i] = ord_u32(f ? c.x : -c.x);  // q.first ascending on '+', descending on '-' (main.rs:12893-12901)
  k64[i] = ((((unsigned long long)R.q[i] << seq_bits | R.qid[i]) << seq_bits | R.tid[i]) << 1) | (f ? 1ull : 0ull);
  idx[i] = i;
}
__global__ __launch_bounds__(256) void gather_u64_by_kernel(const unsigned long long *__restrict__ src, const uint32_t *__restrict__ idx,
                                                            uint32_t n, unsigned long long *__restrict__ dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
  uint32_t p = parent[x];
  while (p != x) {
    const uint32_t gp = parent[p];
    if (gp != p) parent[x] = gp;  // path halving (benign race: only ever points further up)
    x = p;
    p = parent[x];
  }
  return x;
}
__device__ __forceinline__ void uf_unite(uint32_t *parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { const uint32_t t = a; a = b; b = t; }  // the smaller index stays the root: the chain's first row in emission order
    if (atomicCAS(&parent[b], b, a) == b) return;
  }
}
// one thread per sorted position a: the reference's loop over b > a inside a's group (main.rs:12925-12960)
__global__ __launch_bounds__(256) void gap_link_kernel(Rows R, const uint32_t *__restrict__ perm, const unsigned long long *__restrict__ skey,
                                                       uint32_t n, int32_t merge_distance, uint32_t *__restrict__ parent) {
  const uint32_t a = blockIdx.x * 256u + threadIdx.x;
  if (a >= n) return;
  const unsigned long long ka = skey[a];
  const bool f = (ka & 1ull) != 0;
  const uint32_t ia = perm[a];
  const int4 A = R.c[ia];
  const long long d = merge_distance;
  const long long qa_start = f ? A.x : A.y, qa_end = f ? A.y : A.x;
  for (uint32_t b = a + 1; b < n && skey[b] == ka; b++) {
    const uint32_t ib = perm[b];
    const int4 B = R.c[ib];
    const long long qb_start = f ? B.x : B.y;
    if (qb_start < qa_start) continue;  // strictly-backward start
    if (qb_start - qa_end > d) break;   // query gap too large: later rows of the group are not tried
    const long long t_gap = f ? (long long)B.z - A.w : (long long)A.z - B.w;
    const bool t_forward = f ? B.z > A.z : B.w < A.w;
    if (t_forward && t_gap <= d) uf_unite(parent, ia, ib);
  }
}
__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t *v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) v[i] = i;
}
// the chain's bounding box, gathered at its root (the box itself is order-independent)
__global__ __launch_bounds__(256) void gap_box_kernel(Rows R, uint32_t n, uint32_t *__restrict__ parent, int4 *__restrict__ box,
                                                      uint32_t *__restrict__ is_root) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = uf_find(parent, i);
  is_root[i] = r == i ? 1u : 0u;
  if (r == i) return;
  const int4 c = R.c[i];
  const bool f = c.x <= c.y;  // (a chain never mixes strands)
  int *b = reinterpret_cast<int *>(box + r);
  if (f) { atomicMin(b + 0, c.x); atomicMax(b + 1, c.y); } else { atomicMax(b + 0, c.x); atomicMin(b + 1, c.y); }
  atomicMin(b + 2, c.z);
  atomicMax(b + 3, c.w);
}
__global__ __launch_bounds__(256) void compact_rows_kernel(Rows in, const int4 *__restrict__ box, uint32_t n, const uint32_t *__restrict__ flag,
                                                           const uint32_t *__restrict__ pos, Rows out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t d = pos[i];
  out.q[d] = in.q[i]; out.qid[d] = in.qid[i]; out.tid[d] = in.tid[i];
  out.c[d] = box[i];
}

// ---- query axis ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void axis_keys_kernel(Rows R, uint32_t n, unsigned seq_bits, unsigned long long *__restrict__ k_lo,
                                                        unsigned long long *__restrict__ k_hi, uint32_t *__restrict__ idx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int4 c = R.c[i];
  const bool f = c.x <= c.y;
  k_lo[i] = ((unsigned long long)ord_u32(min(c.x, c.y)) << 1) | (f ? 0ull : 1ull);  // start, forward first (main.rs:12486-12499)
  k_hi[i] = ((unsigned long long)R.q[i] << seq_bits) | R.qid[i];
  idx[i] = i;
}
// per sorted row: (segment id << 32 | end) for the prefix maximum; a segment = one (range, query sequence) [+ strand
// run when the strands stay apart]
__global__ __launch_bounds__(256) void axis_seg_kernel(Rows R, const uint32_t *__restrict__ perm, const unsigned long long *__restrict__ skey,
                                                       uint32_t n, int merge_strands, uint32_t *__restrict__ seg_head) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  bool head = k == 0 || skey[k] != skey[k - 1];
  if (!head && !merge_strands) {
    const int4 a = R.c[perm[k - 1]], b = R.c[perm[k]];
    head = (a.x <= a.y) != (b.x <= b.y);
  }
  seg_head[k] = head ? 1u : 0u;
}
__global__ __launch_bounds__(256) void axis_val_kernel(Rows R, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ seg_head,
                                                       const uint32_t *__restrict__ seg_id, uint32_t n, unsigned long long *__restrict__ val) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const int4 c = R.c[perm[k]];
  val[k] = ((unsigned long long)(seg_id[k] + seg_head[k]) << 32) | ord_u32(max(c.x, c.y));  // (exclusive scan + own flag = 1-based id)
}
__global__ __launch_bounds__(256) void axis_run_heads_kernel(Rows R, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ seg_head,
                                                             const unsigned long long *__restrict__ pmax, uint32_t n, int32_t merge_distance,
                                                             uint32_t *__restrict__ run_head) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  bool head = seg_head[k] != 0 || merge_distance < 0;  // main.rs:12515: a negative distance keeps everything apart
  if (!head) {
    const int4 c = R.c[perm[k]];
    const long long ns = min(c.x, c.y);
    const long long ce = (long long)(int32_t)((uint32_t)(pmax[k - 1] & 0xFFFFFFFFull) ^ 0x80000000u);
    head = ns > ce + merge_distance;
  }
  run_head[k] = head ? 1u : 0u;
}
struct BedRow {  // a merged row, 16 bytes
  uint32_t q_strand;  // range index | reverse strand << 31
  uint32_t query_id;
  int32_t start, end;  // (printed as u32, like the reference: an inconsistent CIGAR can project below zero)
};
// one thread per sorted row; run r = run_id[k] (exclusive scan of the heads + own flag - 1)
__global__ __launch_bounds__(256) void axis_emit_kernel(Rows R, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ run_head,
                                                        const uint32_t *__restrict__ run_pos, const unsigned long long *__restrict__ pmax,
                                                        const uint32_t *__restrict__ head_of, uint32_t n, int merge_strands,
                                                        BedRow *__restrict__ out, uint32_t *__restrict__ strand_key) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const uint32_t run = run_pos[k] + run_head[k] - 1u;
  const uint32_t i = perm[k];
  const int4 c = R.c[i];
  const bool f = c.x <= c.y;
  const int32_t ns = min(c.x, c.y), ne = max(c.x, c.y);
  const bool tail = k + 1 == n || run_head[k + 1] != 0;
  if (run_head[k]) {
    out[run].q_strand = R.q[i];
    out[run].query_id = R.qid[i];
    out[run].start = ns;
    atomicMax(&strand_key[run], (1u << 1) | (f ? 0u : 1u));  // position 0 of the run: the fallback strand
  } else if (merge_strands) {
    // the merged strand follows the last row whose own length exceeds the span merged before it (main.rs:12535-12547)
    const long long ce_prev = (long long)(int32_t)((uint32_t)(pmax[k - 1] & 0xFFFFFFFFull) ^ 0x80000000u);
    // the span merged so far starts at the run's head (rows of a run are sorted by start): head_of[k] = its position,
    // a prefix maximum over the head positions
    const uint32_t h = head_of[k];
    const int4 hc = R.c[perm[h]];
    const long long cs = min(hc.x, hc.y);
    if ((long long)ne - ns > ce_prev - cs) atomicMax(&strand_key[run], ((k - h + 1u) << 1) | (f ? 0u : 1u));
  }
  if (tail) {
    // (a run that is a single row ends where the row ends: with a negative distance every row is its own run and the
    // prefix maximum, which runs over the whole segment, is not the row's end)
    const int32_t ce = run_head[k] ? ne : (int32_t)((uint32_t)(pmax[k] & 0xFFFFFFFFull) ^ 0x80000000u);
    out[run].end = ce;  // (the strand is filled in by axis_strand_kernel)
  }
}
__global__ __launch_bounds__(256) void head_pos_kernel(const uint32_t *__restrict__ run_head, uint32_t n, uint32_t *__restrict__ v) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k < n) v[k] = run_head[k] ? k : 0u;
}
__global__ __launch_bounds__(256) void axis_strand_kernel(BedRow *__restrict__ out, const uint32_t *__restrict__ strand_key, uint32_t n_runs) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r < n_runs) out[r].q_strand |= (strand_key[r] & 1u) << 31;
}

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
inline unsigned bits_for(uint64_t n) {
  unsigned b = 0;
  while ((1ull << b) < n) b++;
  return std::max(1u, b);
}

}  // namespace

// ---- text ---------------------------------------------------------------------------------------------------------------
// "{name}\t{start}\t{end}\t{range name}\t.\t{strand}\n" per merged row (output_results_bed, main.rs:11868-11890), written
// by the device: a transitive batch prints gigabytes, which the host's cores format slower than PCIe moves them.
struct TextTables {
  const char *names;            // printed sequence names back to back ("base" under --original-sequence-coordinates)
  const uint32_t *name_off;     // [n_seq + 1]
  const uint32_t *shift;        // [n_seq] offset added to the coordinates (main.rs:11876-11883)
  uint32_t n_names;             // 0: sequences print as their ids
  const char *rnames;           // the chunk's range names back to back
  const uint32_t *rname_off;    // [n_ranges + 1]
};
__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
  uint32_t d = 1;
  while (v >= 10u) { v /= 10u; d++; }
  return d;
}
__device__ __forceinline__ char *put_dec(char *p, uint32_t v, uint32_t digits) {
  for (uint32_t k = digits; k > 0; k--) { p[k - 1] = (char)('0' + v % 10u); v /= 10u; }
  return p + digits;
}
__global__ __launch_bounds__(256) void text_len_kernel(const BedRow *__restrict__ rows, uint32_t n, TextTables t, uint32_t *__restrict__ len) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const BedRow x = rows[r];
  const bool named = x.query_id < t.n_names;
  const uint32_t sh = named ? t.shift[x.query_id] : 0u;
  const uint32_t nl = named ? t.name_off[x.query_id + 1] - t.name_off[x.query_id] : dec_digits(x.query_id);
  const uint32_t q = x.q_strand & 0x7FFFFFFFu;
  len[r] = nl + dec_digits((uint32_t)x.start + sh) + dec_digits((uint32_t)x.end + sh) + (t.rname_off[q + 1] - t.rname_off[q]) + 8u;
}
__global__ __launch_bounds__(256) void text_write_kernel(const BedRow *__restrict__ rows, uint32_t n, TextTables t,
                                                         const unsigned long long *__restrict__ off, unsigned long long base, char *__restrict__ out) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const BedRow x = rows[r];
  char *p = out + (off[r] - base);
  const bool named = x.query_id < t.n_names;
  const uint32_t sh = named ? t.shift[x.query_id] : 0u;
  if (named) {
    const char *nm = t.names + t.name_off[x.query_id];
    const uint32_t nl = t.name_off[x.query_id + 1] - t.name_off[x.query_id];
    for (uint32_t k = 0; k < nl; k++) p[k] = nm[k];
    p += nl;
  } else p = put_dec(p, x.query_id, dec_digits(x.query_id));
  *p++ = '\t';
  const uint32_t a = (uint32_t)x.start + sh, b = (uint32_t)x.end + sh;
  const uint32_t q = x.q_strand & 0x7FFFFFFFu;
  p = put_dec(p, a, dec_digits(a));
  *p++ = '\t';
  p = put_dec(p, b, dec_digits(b));
  *p++ = '\t';
  const char *rn = t.rnames + t.rname_off[q];
  const uint32_t rl = t.rname_off[q + 1] - t.rname_off[q];
  for (uint32_t k = 0; k < rl; k++) p[k] = rn[k];
  p += rl;
  *p++ = '\t'; *p++ = '.'; *p++ = '\t';
  *p++ = (x.q_strand >> 31) ? '-' : '+';
  *p++ = '\n';
}

// Rows of one chunk -> merged BED rows, left on the device in `out` (grouped by range, in output order); returns
// their number.  levels / self_dev: what Engine::run kept for the chunk.
uint32_t device_bed_rows(Engine &E, const impg_gpu_index &ix, uint32_t n_ranges, const impg_gpu_params_t &p, int32_t merge_distance,
                         std::vector<std::unique_ptr<LevelBufs>> &levels, DevBuf &self_dev, DevBuf &out) {
  hipStream_t s = E.stream;
  const bool merge_strands = p.consider_strandness == 0;
  // ---- rows: range-major, the reference's emission order within a range (rows_device.hip) ---------------------------
  uint32_t n = 0;
  DevBuf rq, rqid, rtid, rc;
  Rows R{nullptr, nullptr, nullptr, nullptr};
  {
    RowPlan pl;  // (its tables go back to the pool at the end of this block, before the sorts take their memory)
    plan_rows(E, n_ranges, p, levels, self_dev, /*plain_retain=*/true, pl);
    n = pl.n_rows;
    if (!n) return 0;
    rq.reserve((size_t)n * 4); rqid.reserve((size_t)n * 4); rtid.reserve((size_t)n * 4); rc.reserve((size_t)n * 16);
    R = Rows{rq.as<uint32_t>(), rqid.as<uint32_t>(), rtid.as<uint32_t>(), rc.as<int4>()};
    scatter_rows(E, levels, pl, RowSinks{nullptr, R.q, R.qid, R.tid, R.c, nullptr});
    IMPG_HIP(hipStreamSynchronize(s));
  }
  // the levels' slots are not needed any more: give their memory back before the sorts take theirs
  levels.clear();
  const unsigned seq_bits = bits_for(ix.view.n_seq), q_bits = bits_for(n_ranges);
  if (q_bits + 2 * seq_bits + 1 > 64) throw Error{IMPG_E_UNSUPPORTED, "too many sequences for the device-side BED merge (use impg_gpu_results_bed)"};
  DevBuf k32a, k32b, k64a, k64b, ia, ib, tmp;
  const size_t nb4 = (size_t)n * 4, nb8 = (size_t)n * 8;
  k32a.reserve(nb4); k32b.reserve(nb4); k64a.reserve(nb8); k64b.reserve(nb8); ia.reserve(nb4); ib.reserve(nb4);
  tmp.reserve(std::max(sort_u32_scratch_bytes(n), sort_pairs_scratch_bytes(n)));
  DevBuf rq2, rqid2, rtid2, rc2;
  Rows cur = R;
  uint32_t m = n;
  // ---- gap_2d ------------------------------------------------------------------------------------------------------
  if (merge_distance >= 0 && n > 1) {
    gap_keys_kernel<<<cdiv(n, 256), 256, 0, s>>>(R, n, seq_bits, k32a.as<uint32_t>(), k64a.as<unsigned long long>(), ia.as<uint32_t>());
    launch_sort_u32(tmp.p, tmp.cap, k32a.as<uint32_t>(), k32b.as<uint32_t>(), ia.as<uint32_t>(), ib.as<uint32_t>(), n, s, 0, 32);
    gather_u64_by_kernel<<<cdiv(n, 256), 256, 0, s>>>(k64a.as<unsigned long long>(), ib.as<uint32_t>(), n, k64b.as<unsigned long long>());
    launch_sort_pairs(tmp.p, tmp.cap, k64b.as<unsigned long long>(), k64a.as<unsigned long long>(), ib.as<uint32_t>(), ia.as<uint32_t>(), n,
                      q_bits + 2 * seq_bits + 1, s);
    // sorted: keys in k64a, perm in ia
    DevBuf parent, box, root_flag, root_pos;
    parent.reserve(nb4); box.reserve((size_t)n * 16); root_flag.reserve(nb4); root_pos.reserve(nb4);
    iota_u32_kernel<<<cdiv(n, 256), 256, 0, s>>>(parent.as<uint32_t>(), n);
    gap_link_kernel<<<cdiv(n, 256), 256, 0, s>>>(R, ia.as<uint32_t>(), k64a.as<unsigned long long>(), n, merge_distance, parent.as<uint32_t>());
    IMPG_HIP(hipMemcpyAsync(box.p, rc.p, (size_t)n * 16, hipMemcpyDeviceToDevice, s));
    gap_box_kernel<<<cdiv(n, 256), 256, 0, s>>>(R, n, parent.as<uint32_t>(), box.as<int4>(), root_flag.as<uint32_t>());
    m = (uint32_t)E.scan(root_flag.as<uint32_t>(), root_pos.as<uint32_t>(), n);
    rq2.reserve((size_t)m * 4 + 256); rqid2.reserve((size_t)m * 4 + 256); rtid2.reserve((size_t)m * 4 + 256); rc2.reserve((size_t)m * 16 + 256);
    Rows R2{rq2.as<uint32_t>(), rqid2.as<uint32_t>(), rtid2.as<uint32_t>(), rc2.as<int4>()};
    compact_rows_kernel<<<cdiv(n, 256), 256, 0, s>>>(R, box.as<int4>(), n, root_flag.as<uint32_t>(), root_pos.as<uint32_t>(), R2);
    cur = R2;
  }
  // ---- query axis ----------------------------------------------------------------------------------------------------
  DevBuf strand_key, seg_head, seg_id, val, pmax, run_head, run_pos;
  uint32_t n_runs = m;
  out.reserve((size_t)m * sizeof(BedRow) + 256);
  // (main.rs:12479: the sweep runs when a range has more than one row and (d >= 0 or strands merge); a range with one
  // row is its own run either way, so running it for every range changes nothing)
  if (merge_distance >= 0 || merge_strands) {
    axis_keys_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, m, seq_bits, k64a.as<unsigned long long>(), k64b.as<unsigned long long>(), ia.as<uint32_t>());
    DevBuf k64c;
    k64c.reserve(nb8);
    launch_sort_pairs(tmp.p, tmp.cap, k64a.as<unsigned long long>(), k64c.as<unsigned long long>(), ia.as<uint32_t>(), ib.as<uint32_t>(), m, 33, s);
    gather_u64_by_kernel<<<cdiv(m, 256), 256, 0, s>>>(k64b.as<unsigned long long>(), ib.as<uint32_t>(), m, k64a.as<unsigned long long>());
    launch_sort_pairs(tmp.p, tmp.cap, k64a.as<unsigned long long>(), k64c.as<unsigned long long>(), ib.as<uint32_t>(), ia.as<uint32_t>(), m,
                      q_bits + seq_bits, s);
    // sorted: keys in k64c, perm in ia
    seg_head.reserve((size_t)m * 4); seg_id.reserve((size_t)m * 4); val.reserve((size_t)m * 8); pmax.reserve((size_t)m * 8);
    run_head.reserve((size_t)m * 4); run_pos.reserve((size_t)m * 4);
    axis_seg_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ia.as<uint32_t>(), k64c.as<unsigned long long>(), m, merge_strands ? 1 : 0, seg_head.as<uint32_t>());
    (void)E.scan(seg_head.as<uint32_t>(), seg_id.as<uint32_t>(), m);
    axis_val_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ia.as<uint32_t>(), seg_head.as<uint32_t>(), seg_id.as<uint32_t>(), m, val.as<unsigned long long>());
    {
      size_t sb = 0;
      IMPG_HIP(rocprim::inclusive_scan(nullptr, sb, val.as<unsigned long long>(), pmax.as<unsigned long long>(), m,
                                       rocprim::maximum<unsigned long long>(), s));
      DevBuf stmp;
      stmp.reserve(std::max<size_t>(sb, 256));
      IMPG_HIP(rocprim::inclusive_scan(stmp.p, sb, val.as<unsigned long long>(), pmax.as<unsigned long long>(), m,
                                       rocprim::maximum<unsigned long long>(), s));
      IMPG_HIP(hipStreamSynchronize(s));  // (stmp dies here)
    }
    axis_run_heads_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ia.as<uint32_t>(), seg_head.as<uint32_t>(), pmax.as<unsigned long long>(), m, merge_distance,
                                                        run_head.as<uint32_t>());
    n_runs = (uint32_t)E.scan(run_head.as<uint32_t>(), run_pos.as<uint32_t>(), m);
    strand_key.reserve((size_t)n_runs * 4 + 256);
    IMPG_HIP(hipMemsetAsync(strand_key.p, 0, (size_t)n_runs * 4, s));
    DevBuf head_of;
    head_of.reserve((size_t)m * 4);
    if (merge_strands) {
      head_pos_kernel<<<cdiv(m, 256), 256, 0, s>>>(run_head.as<uint32_t>(), m, seg_id.as<uint32_t>());
      size_t sb = 0;
      IMPG_HIP(rocprim::inclusive_scan(nullptr, sb, seg_id.as<uint32_t>(), head_of.as<uint32_t>(), m, rocprim::maximum<uint32_t>(), s));
      DevBuf stmp;
      stmp.reserve(std::max<size_t>(sb, 256));
      IMPG_HIP(rocprim::inclusive_scan(stmp.p, sb, seg_id.as<uint32_t>(), head_of.as<uint32_t>(), m, rocprim::maximum<uint32_t>(), s));
      IMPG_HIP(hipStreamSynchronize(s));
    }
    axis_emit_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ia.as<uint32_t>(), run_head.as<uint32_t>(), run_pos.as<uint32_t>(), pmax.as<unsigned long long>(),
                                                   head_of.as<uint32_t>(), m, merge_strands ? 1 : 0, out.as<BedRow>(), strand_key.as<uint32_t>());
    axis_strand_kernel<<<cdiv(n_runs, 256), 256, 0, s>>>(out.as<BedRow>(), strand_key.as<uint32_t>(), n_runs);
  } else {
    // no merging at all: rows as they are, in emission order (already grouped by range? no: by level) -- sort by range only
    axis_keys_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, m, seq_bits, k64a.as<unsigned long long>(), k64b.as<unsigned long long>(), ia.as<uint32_t>());
    // key = range index alone (stable: emission order within a range)
    DevBuf kq;
    kq.reserve(nb4);
    IMPG_HIP(hipMemcpyAsync(kq.p, cur.q, (size_t)m * 4, hipMemcpyDeviceToDevice, s));
    launch_sort_u32(tmp.p, tmp.cap, kq.as<uint32_t>(), k32b.as<uint32_t>(), ia.as<uint32_t>(), ib.as<uint32_t>(), m, s, 0, q_bits);
    // every row its own run
    run_head.reserve((size_t)m * 4); run_pos.reserve((size_t)m * 4); pmax.reserve((size_t)m * 8); val.reserve((size_t)m * 8);
    seg_head.reserve((size_t)m * 4); seg_id.reserve((size_t)m * 4);
    IMPG_HIP(hipMemsetAsync(seg_head.p, 0, (size_t)m * 4, s));
    IMPG_HIP(hipMemsetAsync(seg_id.p, 0, (size_t)m * 4, s));
    axis_val_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ib.as<uint32_t>(), seg_head.as<uint32_t>(), seg_id.as<uint32_t>(), m, pmax.as<unsigned long long>());
    axis_run_heads_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ib.as<uint32_t>(), seg_head.as<uint32_t>(), pmax.as<unsigned long long>(), m, -1,
                                                        run_head.as<uint32_t>());
    n_runs = (uint32_t)E.scan(run_head.as<uint32_t>(), run_pos.as<uint32_t>(), m);
    strand_key.reserve((size_t)n_runs * 4 + 256);
    IMPG_HIP(hipMemsetAsync(strand_key.p, 0, (size_t)n_runs * 4, s));
    axis_emit_kernel<<<cdiv(m, 256), 256, 0, s>>>(cur, ib.as<uint32_t>(), run_head.as<uint32_t>(), run_pos.as<uint32_t>(), pmax.as<unsigned long long>(),
                                                   nullptr, m, 0, out.as<BedRow>(), strand_key.as<uint32_t>());
    axis_strand_kernel<<<cdiv(n_runs, 256), 256, 0, s>>>(out.as<BedRow>(), strand_key.as<uint32_t>(), n_runs);
  }
  IMPG_HIP(hipStreamSynchronize(s));  // (the scratch buffers of this function die here)
  return n_runs;
}

// The merged rows of a chunk as text, streamed to `sink(ptr, bytes)` in pieces of at most PIECE bytes through two
// pinned buffers: while the host consumes one piece the device formats and copies the next.
void device_bed_text(Engine &E, const impg_gpu_index &ix, const DevBuf &rows, uint32_t n_rows, uint32_t n_ranges,
                     const std::vector<std::string> &rnames, bool original_coords,
                     const std::function<void(const char *, size_t)> &sink) {
  if (!n_rows) return;
  hipStream_t s = E.stream;
  // ---- tables -------------------------------------------------------------------------------------------------------
  std::string nb;
  std::vector<uint32_t> noff(1, 0), nshift;
  for (const std::string &nm : ix.seq.names) {
    nshift.push_back(put_original_name(nb, nm, original_coords));
    noff.push_back((uint32_t)nb.size());
  }
  std::string rb;
  std::vector<uint32_t> roff(1, 0);
  for (uint32_t q = 0; q < n_ranges; q++) { rb += rnames[q]; roff.push_back((uint32_t)rb.size()); }
  DevBuf d_nb, d_noff, d_shift, d_rb, d_roff, len, off, stmp;
  auto up = [&](DevBuf &d, const void *src, size_t bytes) {
    d.reserve(std::max<size_t>(bytes, 256));
    if (bytes) IMPG_HIP(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s));
  };
  up(d_nb, nb.data(), nb.size()); up(d_noff, noff.data(), noff.size() * 4); up(d_shift, nshift.data(), nshift.size() * 4);
  up(d_rb, rb.data(), rb.size()); up(d_roff, roff.data(), roff.size() * 4);
  TextTables t{d_nb.as<char>(), d_noff.as<uint32_t>(), d_shift.as<uint32_t>(), (uint32_t)ix.seq.names.size(), d_rb.as<char>(), d_roff.as<uint32_t>()};
  // ---- lengths -> 64-bit offsets (a transitive batch prints more than 4 GB) ----------------------------------------------
  len.reserve((size_t)n_rows * 4);
  off.reserve((size_t)(n_rows + 1) * 8);
  text_len_kernel<<<cdiv(n_rows, 256), 256, 0, s>>>(rows.as<BedRow>(), n_rows, t, len.as<uint32_t>());
  size_t sb = 0;
  IMPG_HIP(rocprim::exclusive_scan(nullptr, sb, len.as<uint32_t>(), off.as<unsigned long long>(), 0ull, n_rows, rocprim::plus<unsigned long long>(), s));
  stmp.reserve(std::max<size_t>(sb, 256));
  IMPG_HIP(rocprim::exclusive_scan(stmp.p, sb, len.as<uint32_t>(), off.as<unsigned long long>(), 0ull, n_rows, rocprim::plus<unsigned long long>(), s));
  // piece boundaries: whole rows, at most PIECE bytes each; found on the host from a sampled copy of the offsets
  constexpr size_t PIECE = 256ull << 20;
  std::vector<unsigned long long> h_off(n_rows);
  IMPG_HIP(hipMemcpyAsync(h_off.data(), off.p, (size_t)n_rows * 8, hipMemcpyDeviceToHost, s));
  uint32_t last_len = 0;
  IMPG_HIP(hipMemcpyAsync(&last_len, len.as<uint32_t>() + (n_rows - 1), 4, hipMemcpyDeviceToHost, s));
  IMPG_HIP(hipStreamSynchronize(s));
  const unsigned long long total = h_off[n_rows - 1] + last_len;
  DevBuf piece[2];
  char *pin[2] = {nullptr, nullptr};
  struct Unpin { char **p; ~Unpin() { for (int k = 0; k < 2; k++) if (p[k]) (void)hipHostFree(p[k]); } } unpin{pin};
  const size_t cap = (size_t)std::min<unsigned long long>(total, PIECE) + 4096;
  for (int k = 0; k < 2; k++) { piece[k].reserve(cap); IMPG_HIP(hipHostMalloc((void **)&pin[k], cap, hipHostMallocDefault)); }
  hipEvent_t done[2];
  for (int k = 0; k < 2; k++) IMPG_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
  struct EvFree { hipEvent_t *e; ~EvFree() { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); } } evfree{done};
  struct Pending { size_t bytes; bool live; } pend[2] = {{0, false}, {0, false}};
  uint32_t r0 = 0;
  int slot = 0;
  while (r0 < n_rows || pend[0].live || pend[1].live) {
    if (r0 < n_rows) {
      // rows [r0, r1): as many whole rows as fit one piece
      const unsigned long long base = h_off[r0];
      uint32_t r1 = (uint32_t)(std::upper_bound(h_off.begin() + r0 + 1, h_off.end(), base + PIECE) - h_off.begin());
      if (r1 == n_rows) { if (total - base > PIECE && n_rows - 1 > r0) r1 = n_rows - 1; }
      else r1 = std::max(r1 - 1, r0 + 1);
      const unsigned long long end = r1 < n_rows ? h_off[r1] : total;
      if (end - base > cap) throw Error{IMPG_E_UNSUPPORTED, "a single BED row longer than the text buffer"};
      text_write_kernel<<<cdiv(r1 - r0, 256), 256, 0, s>>>(rows.as<BedRow>() + r0, r1 - r0, t, off.as<unsigned long long>() + r0, base, piece[slot].as<char>());
      IMPG_HIP(hipMemcpyAsync(pin[slot], piece[slot].p, (size_t)(end - base), hipMemcpyDeviceToHost, s));
      IMPG_HIP(hipEventRecord(done[slot], s));
      pend[slot] = {(size_t)(end - base), true};
      r0 = r1;
    }
    // consume the OTHER slot's piece while this one is in flight (or drain at the end)
    const int other = slot ^ 1;
    if (pend[other].live) {
      IMPG_HIP(hipEventSynchronize(done[other]));
      sink(pin[other], pend[other].bytes);
      pend[other].live = false;
    } else if (r0 >= n_rows && pend[slot].live) {
      IMPG_HIP(hipEventSynchronize(done[slot]));
      sink(pin[slot], pend[slot].bytes);
      pend[slot].live = false;
    }
    slot ^= 1;
  }
}

}  // namespace impg
