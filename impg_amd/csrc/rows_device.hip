// The result rows of one chunk, built on the device: what ImpgIndex::query / query_transitive_{bfs,dfs} return
// (Vec<AdjustedInterval>, src/impg_index.rs:26-35, :79-94) for every range of the chunk, grouped by range in the
// reference's emission order -- the self interval(s) first (impg.rs:1864-1880, :2345-2363), then level by level /
// pop by pop, each level's hits in frontier order x visit order (impg.rs:1897-1925, :2471-2504).
//
// Until round 3 the host did this: every level's raw slot arrays (24 B per slot, empty slots included) crossed PCIe
// into pageable vectors and were assembled per range on host threads -- 2.4 s for the 2.1e8 rows of BASELINE config 3
// against 11 ms of engine.  Here the rows are placed by the device and cross PCIe once, finished.
//
// The placement does not care how a level's slots are laid out, only that the slots of one frontier record are
// one contiguous run in visit order -- true for the reference's slot order and for the lookup-order layout of
// Engine::free_slot_order alike, so kept levels may use the cheaper layout too:
//
//   flags     per slot over [self records | level 0 | level 1 | ...]: is the slot an emitted row
//   pos       exclusive scan of the flags (32 bits: a chunk has fewer than 2^32 slots)
//   records   one per self record and per frontier record of every level, in that order (= emission order within a
//             range): its range, its first slot's pos and its number of emitted rows (from the heads and tails of
//             the runs of equal pair_range)
//   order     stable radix sort of the records by range: range-major, emission order within a range
//   dest      exclusive scan of the records' row counts in that order = the record's first row; offsets[q] = the
//             first row of range q's first record
//   scatter   row of slot k of record r = dest[r] + pos[k] - pos[first slot of r]
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>

#include "engine.hpp"

namespace impg {

namespace {

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
inline unsigned bits_for(uint64_t n) {
  unsigned b = 0;
  while ((1ull << b) < n) b++;
  return std::max(1u, b);
}

__global__ __launch_bounds__(256) void self_flags_kernel(const FrontierRec *__restrict__ self, uint32_t n_self, int transitive,
                                                         int32_t min_len_plain, uint32_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_self) return;
  const FrontierRec f = self[i];
  bool on = transitive ? f.start < f.end : true;  // impg.rs:2345-2363 / :1864-1880
  if (!transitive && min_len_plain >= 0 && abs(f.end - f.start) < min_len_plain) on = false;  // perform_query retain, main.rs:11682-11688
  flag[i] = on ? 1u : 0u;
}
__global__ __launch_bounds__(256) void slot_flags_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ pair_range,
                                                         uint32_t n_pairs, HitArrays h, int32_t min_len, int skip_same,
                                                         uint32_t *__restrict__ flag) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t qid = h.qid[p];
  bool on = qid != HIT_NONE;
  if (on) {
    const int4 hc = h.c[p];
    if (min_len >= 0 && abs(hc.y - hc.x) < min_len) on = false;            // impg.rs:2482-2504 / main.rs:11682-11688
    if (on && skip_same && qid == fr[pair_range[p]].target_id) on = false;  // multi_impg.rs:883-885
  }
  flag[p] = on ? 1u : 0u;
}
__global__ __launch_bounds__(256) void self_records_kernel(const FrontierRec *__restrict__ self, uint32_t n_self, const uint32_t *__restrict__ flag,
                                                           const uint32_t *__restrict__ pos, uint32_t *__restrict__ rec_q,
                                                           uint32_t *__restrict__ rec_start, uint32_t *__restrict__ rec_cnt) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_self) return;
  rec_q[i] = self[i].qidx;
  rec_start[i] = pos[i];
  rec_cnt[i] = flag[i];
}
// heads and tails of the runs of equal pair_range: pos of the record's first slot, pos past its last emitted row
__global__ __launch_bounds__(256) void record_bounds_kernel(const uint32_t *__restrict__ pair_range, uint32_t n_pairs,
                                                            const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                            uint32_t *__restrict__ rec_start, uint32_t *__restrict__ rec_end) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n_pairs) return;
  const uint32_t r = pair_range[k];
  if (k == 0 || pair_range[k - 1] != r) rec_start[r] = pos[k];
  if (k + 1 == n_pairs || pair_range[k + 1] != r) rec_end[r] = pos[k] + flag[k];
}
__global__ __launch_bounds__(256) void level_records_kernel(const FrontierRec *__restrict__ fr, uint32_t n_fr, uint32_t *__restrict__ rec_q,
                                                            const uint32_t *__restrict__ rec_start, uint32_t *__restrict__ rec_cnt) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n_fr) return;
  rec_q[r] = fr[r].qidx;
  rec_cnt[r] = rec_cnt[r] - rec_start[r];  // (rec_cnt held the end; a record without slots has 0 - 0)
}
__global__ __launch_bounds__(256) void iota_kernel(uint32_t *v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n,
                                                     uint32_t *__restrict__ dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// sorted records -> rec_dest (back at the record's own index) and the ranges' first rows
__global__ __launch_bounds__(256) void record_dest_kernel(const uint32_t *__restrict__ sq, const uint32_t *__restrict__ sidx,
                                                          const uint32_t *__restrict__ sdest, const uint32_t *__restrict__ scnt, uint32_t n_rec,
                                                          uint32_t n_ranges, uint32_t *__restrict__ rec_dest, uint32_t *__restrict__ offsets) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_rec) return;
  const uint32_t q = sq[i], d = sdest[i];
  rec_dest[sidx[i]] = d;
  const long long prev = i ? (long long)sq[i - 1] : -1;
  for (long long x = prev + 1; x <= (long long)q; x++) offsets[x] = d;  // (ranges without records start where the next one does)
  if (i + 1 == n_rec) {
    const uint32_t total = d + scnt[i];
    for (uint32_t x = q + 1; x <= n_ranges; x++) offsets[x] = total;
  }
}

struct Sinks {
  uint2 *rows;  // impg_gpu_interval_t as three 8-byte words
  uint32_t *q, *qid, *tid;
  int4 *c;
  uint32_t *clen;  // store_cigar: ops of the row's CIGAR
};
__device__ __forceinline__ void put_row(const Sinks &S, uint32_t d, uint32_t q, uint32_t qid, uint32_t tid, int4 c) {
  if (S.rows) {
    uint2 *p = S.rows + 3ull * d;
    p[0] = make_uint2(qid, (uint32_t)c.x);
    p[1] = make_uint2((uint32_t)c.y, tid);
    p[2] = make_uint2((uint32_t)c.z, (uint32_t)c.w);
  }
  if (S.c) { S.q[d] = q; S.qid[d] = qid; S.tid[d] = tid; S.c[d] = c; }
}
__global__ __launch_bounds__(256) void self_rows_kernel(const FrontierRec *__restrict__ self, uint32_t n_self, const uint32_t *__restrict__ flag,
                                                        const uint32_t *__restrict__ rec_dest, Sinks S) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_self || !flag[i]) return;
  const FrontierRec f = self[i];
  const uint32_t d = rec_dest[i];
  put_row(S, d, f.qidx, f.target_id, f.target_id, make_int4(f.start, f.end, f.start, f.end));
  if (S.clen) S.clen[d] = 1u;  // vec![CigarOp::new(end - start, '=')] (impg.rs:1870-1872, :2352-2354)
}
__global__ __launch_bounds__(256) void slot_rows_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ pair_range,
                                                        uint32_t n_pairs, HitArrays h, const uint32_t *__restrict__ flag,
                                                        const uint32_t *__restrict__ pos, const uint32_t *__restrict__ rec_start,
                                                        const uint32_t *__restrict__ rec_dest, const uint32_t *__restrict__ sl_n, Sinks S) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n_pairs || !flag[k]) return;
  const uint32_t r = pair_range[k];
  const FrontierRec f = fr[r];
  const uint32_t d = rec_dest[r] + (pos[k] - rec_start[r]);
  put_row(S, d, f.qidx, h.qid[k], f.target_id, h.c[k]);
  if (S.clen) S.clen[d] = sl_n[k];
}

// ---- store_cigar: every row's op list into one pool ----------------------------------------------------------------
__global__ __launch_bounds__(256) void self_cigar_kernel(const FrontierRec *__restrict__ self, uint32_t n_self, const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ rec_dest, const unsigned long long *__restrict__ coff,
                                                         uint32_t *__restrict__ pool) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_self || !flag[i]) return;
  const FrontierRec f = self[i];
  pool[coff[rec_dest[i]]] = (uint32_t)(f.end - f.start);  // code 0 '=' (CigarOp::new, impg.rs:81-93)
}
// one wave per slot: copies the slot's materialised slice (slice_pool[slice_pos[k] .. + sl_n[k])) to its row's place
__global__ __launch_bounds__(256) void slot_cigar_kernel(const uint32_t *__restrict__ pair_range, uint32_t n_pairs, const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ pos, const uint32_t *__restrict__ rec_start,
                                                         const uint32_t *__restrict__ rec_dest, const uint32_t *__restrict__ sl_n,
                                                         const uint32_t *__restrict__ slice_pos, const uint32_t *__restrict__ slice_pool,
                                                         const unsigned long long *__restrict__ coff, uint32_t *__restrict__ pool) {
  const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (k >= n_pairs || !flag[k]) return;
  const uint32_t r = pair_range[k];
  const uint32_t d = rec_dest[r] + (pos[k] - rec_start[r]);
  const uint32_t n = sl_n[k];
  const uint32_t *src = slice_pool + slice_pos[k];
  uint32_t *dst = pool + coff[d];
  for (uint32_t i = threadIdx.x & 63u; i < n; i += 64u) dst[i] = src[i];
}

}  // namespace

void plan_rows(Engine &E, uint32_t n_ranges, const impg_gpu_params_t &p, std::vector<std::unique_ptr<LevelBufs>> &levels,
               DevBuf &self_dev, bool plain_retain, RowPlan &pl) {
  hipStream_t s = E.stream;
  const bool transitive = p.transitive != 0;
  pl.n_self = transitive ? (E.masked ? (uint32_t)E.n_self : n_ranges) : n_ranges;
  pl.d_self = self_dev.as<FrontierRec>();
  // Impg::query: the self interval is the range itself (impg.rs:1864-1880) = frontier 0 of the run
  if (!transitive) pl.d_self = levels.empty() ? nullptr : levels[0]->frontier.as<FrontierRec>();
  if (!pl.d_self) pl.n_self = 0;
  uint64_t S = pl.n_self, R = pl.n_self;
  for (auto &L : levels) { S += L->n_pairs; R += L->n_frontier; }
  if (S >= 0xFFFFFFF0ull || R >= 0xFFFFFFF0ull)
    throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 result slots in one chunk: use smaller chunks (chunk_ranges)"};
  pl.n_slots = S; pl.n_recs = R; pl.n_rows = 0;
  for (DevBuf *b : {&pl.flag, &pl.pos, &pl.rec_start, &pl.rec_dest, &pl.offsets}) b->pool = &E.level_pool;
  pl.offsets.reserve(((size_t)n_ranges + 1) * 4);
  if (!S || !R) {
    IMPG_HIP(hipMemsetAsync(pl.offsets.p, 0, ((size_t)n_ranges + 1) * 4, s));
    return;
  }
  pl.flag.reserve(S * 4); pl.pos.reserve(S * 4);
  uint32_t *fl = pl.flag.as<uint32_t>(), *pos = pl.pos.as<uint32_t>();
  if (pl.n_self)
    self_flags_kernel<<<cdiv(pl.n_self, 256), 256, 0, s>>>(pl.d_self, pl.n_self, transitive ? 1 : 0, plain_retain ? p.min_output_length : -1, fl);
  // transitive: min_output_length is applied while collecting (impg.rs:2482-2504); plain: only by perform_query's
  // retain (main.rs:11682-11688), i.e. for the text writers
  const int32_t min_len = (transitive || plain_retain) ? p.min_output_length : -1;
  uint64_t sb = pl.n_self;
  for (auto &L : levels) {
    if (L->n_pairs) {
      HitArrays h{L->qid.as<uint32_t>(), L->coords.as<int4>()};
      slot_flags_kernel<<<cdiv(L->n_pairs, 256), 256, 0, s>>>(L->frontier.as<FrontierRec>(), L->pair_range.as<uint32_t>(), L->n_pairs, h, min_len,
                                                              (transitive && p.multi_impg) ? 1 : 0, fl + sb);
    }
    sb += L->n_pairs;
  }
  const uint64_t n_rows = E.scan(fl, pos, (uint32_t)S);
  // ---- records ---------------------------------------------------------------------------------------------------
  DevBuf rec_q, rec_cnt, sq, idx, sidx, scnt, sdest, tmp;
  for (DevBuf *b : {&rec_q, &rec_cnt, &sq, &idx, &sidx, &scnt, &sdest, &tmp}) b->pool = &E.level_pool;
  const size_t rb4 = (size_t)R * 4;
  pl.rec_start.reserve(rb4); pl.rec_dest.reserve(rb4);
  rec_q.reserve(rb4); rec_cnt.reserve(rb4); sq.reserve(rb4); idx.reserve(rb4); sidx.reserve(rb4); scnt.reserve(rb4); sdest.reserve(rb4);
  uint32_t *rs = pl.rec_start.as<uint32_t>(), *rc = rec_cnt.as<uint32_t>(), *rq = rec_q.as<uint32_t>();
  if (pl.n_self) self_records_kernel<<<cdiv(pl.n_self, 256), 256, 0, s>>>(pl.d_self, pl.n_self, fl, pos, rq, rs, rc);
  if (R > pl.n_self) {
    IMPG_HIP(hipMemsetAsync(rs + pl.n_self, 0, (size_t)(R - pl.n_self) * 4, s));
    IMPG_HIP(hipMemsetAsync(rc + pl.n_self, 0, (size_t)(R - pl.n_self) * 4, s));
  }
  sb = pl.n_self;
  uint64_t rb = pl.n_self;
  for (auto &L : levels) {
    if (L->n_pairs)
      record_bounds_kernel<<<cdiv(L->n_pairs, 256), 256, 0, s>>>(L->pair_range.as<uint32_t>(), L->n_pairs, fl + sb, pos + sb, rs + rb, rc + rb);
    if (L->n_frontier)
      level_records_kernel<<<cdiv(L->n_frontier, 256), 256, 0, s>>>(L->frontier.as<FrontierRec>(), L->n_frontier, rq + rb, rs + rb, rc + rb);
    sb += L->n_pairs;
    rb += L->n_frontier;
  }
  // ---- range-major order (stable: emission order within a range) -----------------------------------------------------
  iota_kernel<<<cdiv(R, 256), 256, 0, s>>>(idx.as<uint32_t>(), (uint32_t)R);
  const size_t tb = sort_u32_scratch_bytes((uint32_t)R);
  tmp.reserve(tb);
  launch_sort_u32(tmp.p, tb, rq, sq.as<uint32_t>(), idx.as<uint32_t>(), sidx.as<uint32_t>(), (uint32_t)R, s, 0, bits_for(n_ranges));
  gather_kernel<<<cdiv(R, 256), 256, 0, s>>>(rc, sidx.as<uint32_t>(), (uint32_t)R, scnt.as<uint32_t>());
  const uint64_t total = E.scan(scnt.as<uint32_t>(), sdest.as<uint32_t>(), (uint32_t)R);
  if (total != n_rows) throw Error{IMPG_E_INVALID, "internal: result rows and records disagree"};
  record_dest_kernel<<<cdiv(R, 256), 256, 0, s>>>(sq.as<uint32_t>(), sidx.as<uint32_t>(), sdest.as<uint32_t>(), scnt.as<uint32_t>(), (uint32_t)R,
                                                  n_ranges, pl.rec_dest.as<uint32_t>(), pl.offsets.as<uint32_t>());
  pl.n_rows = (uint32_t)n_rows;
  IMPG_HIP(hipStreamSynchronize(s));  // (the scratch buffers above go back to the pool here)
}

void scatter_rows(Engine &E, std::vector<std::unique_ptr<LevelBufs>> &levels, const RowPlan &pl, const RowSinks &out) {
  if (!pl.n_rows) return;
  hipStream_t s = E.stream;
  Sinks S{reinterpret_cast<uint2 *>(out.rows), out.q, out.qid, out.tid, out.c, out.clen};
  const uint32_t *fl = pl.flag.as<uint32_t>(), *pos = pl.pos.as<uint32_t>(), *rs = pl.rec_start.as<uint32_t>(), *rd = pl.rec_dest.as<uint32_t>();
  if (pl.n_self) self_rows_kernel<<<cdiv(pl.n_self, 256), 256, 0, s>>>(pl.d_self, pl.n_self, fl, rd, S);
  uint64_t sb = pl.n_self, rb = pl.n_self;
  for (auto &L : levels) {
    if (L->n_pairs) {
      HitArrays h{L->qid.as<uint32_t>(), L->coords.as<int4>()};
      slot_rows_kernel<<<cdiv(L->n_pairs, 256), 256, 0, s>>>(L->frontier.as<FrontierRec>(), L->pair_range.as<uint32_t>(), L->n_pairs, h, fl + sb, pos + sb,
                                                             rs + rb, rd + rb, out.clen ? L->sl_n.as<uint32_t>() : nullptr, S);
    }
    sb += L->n_pairs;
    rb += L->n_frontier;
  }
}

// store_cigar: clen[row] (written by scatter_rows) -> coff[n_rows + 1] and the pool.  Returns the number of ops.
uint64_t build_row_cigars(Engine &E, std::vector<std::unique_ptr<LevelBufs>> &levels, const RowPlan &pl, const DevBuf &clen, DevBuf &coff,
                          DevBuf &pool) {
  hipStream_t s = E.stream;
  coff.reserve(((size_t)pl.n_rows + 1) * 8);
  if (!pl.n_rows) {
    IMPG_HIP(hipMemsetAsync(coff.p, 0, 8, s));
    return 0;
  }
  size_t sb = 0;
  IMPG_HIP(rocprim::exclusive_scan(nullptr, sb, clen.as<uint32_t>(), coff.as<unsigned long long>(), 0ull, pl.n_rows, rocprim::plus<unsigned long long>(), s));
  DevBuf stmp;
  stmp.pool = &E.level_pool;
  stmp.reserve(std::max<size_t>(sb, 256));
  IMPG_HIP(rocprim::exclusive_scan(stmp.p, sb, clen.as<uint32_t>(), coff.as<unsigned long long>(), 0ull, pl.n_rows, rocprim::plus<unsigned long long>(), s));
  unsigned long long last_off = 0;
  uint32_t last_len = 0;
  IMPG_HIP(hipMemcpyAsync(&last_off, coff.as<unsigned long long>() + (pl.n_rows - 1), 8, hipMemcpyDeviceToHost, s));
  IMPG_HIP(hipMemcpyAsync(&last_len, clen.as<uint32_t>() + (pl.n_rows - 1), 4, hipMemcpyDeviceToHost, s));
  IMPG_HIP(hipStreamSynchronize(s));
  const uint64_t total = last_off + last_len;
  IMPG_HIP(hipMemcpyAsync(coff.as<unsigned long long>() + pl.n_rows, &total, 8, hipMemcpyHostToDevice, s));
  pool.reserve(std::max<size_t>(total * 4, 256));
  const uint32_t *fl = pl.flag.as<uint32_t>(), *pos = pl.pos.as<uint32_t>(), *rs = pl.rec_start.as<uint32_t>(), *rd = pl.rec_dest.as<uint32_t>();
  if (pl.n_self)
    self_cigar_kernel<<<cdiv(pl.n_self, 256), 256, 0, s>>>(pl.d_self, pl.n_self, fl, rd, coff.as<unsigned long long>(), pool.as<uint32_t>());
  uint64_t b = pl.n_self, rb = pl.n_self;
  for (auto &L : levels) {
    if (L->n_pairs)
      slot_cigar_kernel<<<cdiv(L->n_pairs, 4), 256, 0, s>>>(L->pair_range.as<uint32_t>(), L->n_pairs, fl + b, pos + b, rs + rb, rd + rb,
                                                            L->sl_n.as<uint32_t>(), L->slice_pos.as<uint32_t>(), L->slice_pool.as<uint32_t>(),
                                                            coff.as<unsigned long long>(), pool.as<uint32_t>());
    b += L->n_pairs;
    rb += L->n_frontier;
  }
  IMPG_HIP(hipStreamSynchronize(s));  // (`total` on the stack was the source of an async copy; stmp dies here)
  return total;
}

}  // namespace impg
