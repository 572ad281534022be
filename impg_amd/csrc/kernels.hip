// Hand-written HIP kernels for gfx950 (CDNA4, wave64).  Integer/HBM-bound work:
// no MFMA.  See DESIGN.md section 5 for the per-kernel roofline accounting.
//
//   lookup_count / lookup_emit : replace coitrees BasicCOITree::query
//        (call sites src/impg.rs:1897, :2142, :2393).  One wavefront per
//        frontier range: 64-ary coalesced search of the per-target sorted
//        start array and of the running-max-of-end array, then a coalesced
//        scan of the candidate window with ballot compaction; hits are written
//        in the reference's visit order (rank[]).
//   project : replaces get_cigar_ops + project_target_range_through_alignment
//        (src/impg.rs:495-551, :2760-2898).  One lane per (range, entry) pair.
//        The CIGAR walk is cut down to the two 32-op tiles that hold the first
//        and the last overlapping op, located through per-tile prefix
//        checkpoints; reversed entries (REVERSED_BIT) read the same tiles with
//        I<->D swapped and, on the reverse strand, back to front
//        (invert_cigar_ops_in_place, src/impg.rs:144-156) -- no second copy.
//   visited_update / frontier_* : the order-dependent sequential phase of
//        query_transitive_bfs (src/impg.rs:2471-2584) with SortedRanges::insert
//        (src/impg.rs:270-368), parallel over (query, sequence) groups.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "impg_internal.hpp"
#include "kernels.hpp"

namespace impg {

// ---------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Inclusive scan over the wave on the DPP network: four shifts inside each row of 16 lanes, then lane 15 of rows 0 / 2
// into rows 1 / 3 and lane 31 into rows 2 and 3.  A lane without a source (row start, masked row) reads 0.  Six
// v_add_u32_dpp where the __shfl_up form is six ds_bpermute round trips through LDS with a compare + select each.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  x += dpp_or0<0x111, 0xF>(x);  // row_shr:1
  x += dpp_or0<0x112, 0xF>(x);  // row_shr:2
  x += dpp_or0<0x114, 0xF>(x);  // row_shr:4
  x += dpp_or0<0x118, 0xF>(x);  // row_shr:8
  x += dpp_or0<0x142, 0xA>(x);  // row_bcast:15 into rows 1, 3
  x += dpp_or0<0x143, 0xC>(x);  // row_bcast:31 into rows 2, 3
  return x;
}
// max over the wave, in every lane (same network; a lane without a source keeps its own value; lane 63 ends up with it)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_self(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  x = max(x, dpp_or_self<0x111, 0xF>(x));
  x = max(x, dpp_or_self<0x112, 0xF>(x));
  x = max(x, dpp_or_self<0x114, 0xF>(x));
  x = max(x, dpp_or_self<0x118, 0xF>(x));
  x = max(x, dpp_or_self<0x142, 0xA>(x));
  x = max(x, dpp_or_self<0x143, 0xC>(x));
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// Inside a window every entry already starts before the range end (ub), so the
// overlap test needs the end column only.  Transitive levels use ends_t, where an
// entry with first >= last holds INT_MIN: max(cs,first) < min(ce,last)
// (impg.rs:2398-2403) can never hold for it.
template <bool TRANSITIVE>
__device__ __forceinline__ bool window_hit(int32_t te, int32_t qs) {
  return TRANSITIVE ? te > qs : te >= qs;
}
template <bool TRANSITIVE>
__device__ __forceinline__ const int32_t *end_col(const DeviceIndexView &v) { return TRANSITIVE ? v.ends_t : v.ends; }

// ---------------------------------------------------------------------------
// K1a: count overlapping entries per frontier range, one LANE per range.  (The first
// cut, one wave per range descending 64-ary sampled levels with ballots, spent ~200
// VALU instructions per range and was VALU-bound; a lane running its own two searches
// and walking its own window costs ~15, and the gathers it issues instead are what
// the memory pipeline is good at -- rocprofv3)
//   ub = first entry of the segment with start >= range end   (> for the closed test)
//   lo = first entry whose running max of ends > range start  (>= ...)
//   hits = entries of [lo, ub) whose end passes the same test; the mask of the
//   first 64 goes to the emit pass with the window.
// ---------------------------------------------------------------------------
// Two lower-bound searches side by side -- first s in [l1,h1) with S[s] >= qe (> for
// the closed test), first p in [l2,h2) with P[p] > qs (>=) -- 4-ary: three probes per
// search and round are in flight together.
template <bool TRANSITIVE>
__device__ __forceinline__ void dual_search4(const int32_t *__restrict__ S, uint32_t &l1, uint32_t &h1,
                                             const int32_t *__restrict__ P, uint32_t &l2, uint32_t &h2, int32_t qs, int32_t qe) {
  while (l1 < h1 || l2 < h2) {
    const uint32_t w1 = h1 - l1, w2 = h2 - l2;
    const uint32_t a1 = l1 + (w1 >> 2), b1 = l1 + (w1 >> 1), c1 = l1 + (w1 >> 1) + (w1 >> 2);
    const uint32_t a2 = l2 + (w2 >> 2), b2 = l2 + (w2 >> 1), c2 = l2 + (w2 >> 1) + (w2 >> 2);
    const bool g1 = w1 != 0, g2 = w2 != 0;   // (a <= b <= c < h whenever w != 0)
    const int32_t sa = g1 ? S[a1] : 0, sb = g1 ? S[b1] : 0, sc = g1 ? S[c1] : 0;
    const int32_t pa = g2 ? P[a2] : 0, pb = g2 ? P[b2] : 0, pc = g2 ? P[c2] : 0;
    if (g1) {
      const bool ta = TRANSITIVE ? sa >= qe : sa > qe, tb = TRANSITIVE ? sb >= qe : sb > qe, tc = TRANSITIVE ? sc >= qe : sc > qe;
      // the predicate is monotone: pick the quarter that holds the first true
      if (ta) h1 = a1; else if (tb) { l1 = a1 + 1u; h1 = b1; } else if (tc) { l1 = b1 + 1u; h1 = c1; } else l1 = c1 + 1u;
    }
    if (g2) {
      const bool ta = TRANSITIVE ? pa > qs : pa >= qs, tb = TRANSITIVE ? pb > qs : pb >= qs, tc = TRANSITIVE ? pc > qs : pc >= qs;
      if (ta) h2 = a2; else if (tb) { l2 = a2 + 1u; h2 = b2; } else if (tc) { l2 = b2 + 1u; h2 = c2; } else l2 = c2 + 1u;
    }
  }
}

template <bool TRANSITIVE>
__global__ __launch_bounds__(256) void lookup_count_lane_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr,
                                                                uint32_t n, const uint32_t *__restrict__ perm,
                                                                uint32_t *__restrict__ cnt,
                                                                uint4 *__restrict__ win, uint32_t *__restrict__ wide_n,
                                                                uint32_t *__restrict__ wide_list, int by_place, FrontierRec *__restrict__ se,
                                                                uint32_t *__restrict__ cnt_ref) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  // perm (optional): neighbouring lanes take ranges that are neighbours in the entry array, so
  // their searches and windows share cache lines; results still land at the range's own index --
  // or, by_place, at the lane's place in that order (coalesced; the emit pass then reads them there)
  const uint32_t r = perm ? perm[i] : i;
  const uint32_t o = by_place ? i : r;
  const FrontierRec f = fr[r];
  uint32_t a = 0, sn = 0, off0 = 0, cnt0 = 0;
  if (f.target_id < v.n_seq) {
    const uint4 *dp = reinterpret_cast<const uint4 *>(v.seg + f.target_id);
    const uint4 d0 = dp[0], d1 = dp[1];  // {a, n, nlev, off[0]}, {off[1..3], cnt[0]}
    a = d0.x;
    sn = d0.y;
    if (d0.z) { off0 = d0.w; cnt0 = d1.w; }
  }
  const int32_t qs = f.start, qe = f.end;
  // first over the segment's samples (the last element of every 64-entry block: a
  // small array that stays in L2), then inside the one block that holds the answer
  uint32_t l1 = 0, h1 = sn, l2 = 0, h2 = sn;
  if (cnt0) {
    uint32_t b1 = 0, e1 = cnt0, b2 = 0, e2 = cnt0;
    dual_search4<TRANSITIVE>(v.starts_lvl + off0, b1, e1, v.pmax_lvl + off0, b2, e2, qs, qe);
    // b = first block whose last element passes; none: the answer is n
    l1 = b1 < cnt0 ? 64u * b1 : sn; h1 = b1 < cnt0 ? min(l1 + 63u, sn) : sn;  // (that last element itself passes)
    l2 = b2 < cnt0 ? 64u * b2 : sn; h2 = b2 < cnt0 ? min(l2 + 63u, sn) : sn;
  }
  dual_search4<TRANSITIVE>(v.starts + a, l1, h1, v.pmax + a, l2, h2, qs, qe);
  const uint32_t ub = a + l1, lo = a + min(l2, l1);
  // the window, eight entries (two aligned 16-byte vectors) per round
  const int32_t *ecol = end_col<TRANSITIVE>(v);
  uint32_t c = 0;
  unsigned long long mask = 0;
  // (a window wider than the mask is counted by a wave of lookup_count_wide_kernel, which takes it off the wide list: a lane
  // walking thousands of entries of a dense target kept its 63 neighbours waiting -- 3.6 of a skewed step's 22 ms)
  const bool wide = lo < ub && ub - (lo & ~3u) > 64u;
  const bool deferred = wide && wide_list != nullptr;
  for (uint32_t b = lo & ~3u; b < ub && !deferred; b += 8u) {
    const int4 e0 = *reinterpret_cast<const int4 *>(ecol + b);
    int4 e1 = make_int4(0, 0, 0, 0);
    if (b + 4u < ub) e1 = *reinterpret_cast<const int4 *>(ecol + b + 4u);
    const int32_t ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t i = b + k;
      const bool hit = i >= lo && i < ub && window_hit<TRANSITIVE>(ev[k], qs);
      c += hit ? 1u : 0u;
      const uint32_t bit = i - lo;  // (only meaningful for a hit)
      if (hit && bit < 64u) mask |= 1ull << bit;
    }
  }
  cnt[o] = c;
  if (cnt_ref) cnt_ref[r] = c;
  win[o] = make_uint4(lo, ub, (uint32_t)mask, (uint32_t)(mask >> 32));
  if (se) se[o] = f;  // (the whole record at the range's place: the projection reads its ends there, a kept level its range and target)
  // windows too wide for the lane-per-range emit pass (dense targets) are listed for the wave-per-range one
  if (deferred) wide_list[atomicAdd(wide_n, 1u)] = o;
}
// ... and the wide windows' counts: a wave per listed window, 64 entries a round
template <bool TRANSITIVE>
__global__ __launch_bounds__(256) void lookup_count_wide_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ perm,
                                                                const uint32_t *__restrict__ wide_n, const uint32_t *__restrict__ wide_list,
                                                                const uint4 *__restrict__ win, int by_place, uint32_t *__restrict__ cnt,
                                                                uint32_t *__restrict__ cnt_ref) {
  const uint32_t nw = *wide_n, lane = lane_id();
  const int32_t *ecol = end_col<TRANSITIVE>(v);
  for (uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6); k < nw; k += gridDim.x * 4u) {
    const uint32_t o = wide_list[k];
    const uint32_t r = by_place ? perm[o] : o;
    const uint4 w = win[o];
    const int32_t qs = fr[r].start;
    uint32_t c = 0;
    for (uint32_t b = w.x + lane; b < w.y; b += 256u) {  // (four loads in flight)
      int32_t e[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) e[u] = b + 64u * u < w.y ? ecol[b + 64u * u] : 0;
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) c += (b + 64u * u < w.y && window_hit<TRANSITIVE>(e[u], qs)) ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
    if (lane == 0) { cnt[o] = c; if (cnt_ref) cnt_ref[r] = c; }
  }
}

// ---------------------------------------------------------------------------
// K1b: emit (range, entry) pairs in visit order at pair_off[r]
// ---------------------------------------------------------------------------
template <bool TRANSITIVE>
__global__ __launch_bounds__(256) void lookup_emit_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr,
                                                          uint32_t n, const uint32_t *__restrict__ pair_off,
                                                          const uint4 *__restrict__ win,
                                                          uint32_t *__restrict__ pair_range,
                                                          uint32_t *__restrict__ pair_entry,
                                                          const uint32_t *__restrict__ offp,
                                                          ProjList pl,
                                                          const uint32_t *__restrict__ list,
                                                          const uint32_t *__restrict__ list_n,
                                                          const uint32_t *__restrict__ place_perm) {
  // place_perm (optional): items, win[] and pair_off[] are indexed by PLACE in the lookup order; the range is place_perm[place]
  // list (optional): process only these ranges -- the ones whose window is wider than
  // lookup_emit_lane_kernel takes, collected by the count pass
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * 256u) >> 6;
  const unsigned lane = lane_id();
  const int32_t *ecol = end_col<TRANSITIVE>(v);
  uint32_t *const slot_of = pl.slot;
  // pl (optional): the same pairs listed in PROJECTION order -- ranges sorted by
  // their window position in the entry array -- so that neighbouring lanes of the
  // projection kernel gather neighbouring entries and tiles (offp[r] = first
  // position of range r in that order).  Slot order itself never changes.
  // software pipeline: the next range's window record and slot offset are
  // requested before the current range is processed, so a range costs one
  // dependent round trip (its rank column), not three
  const uint32_t n_items = list ? min(*list_n, n) : n;
  uint4 w = make_uint4(0, 0, 0, 0);
  uint32_t off = 0, po = 0, rcur = 0;
  if (wave < n_items) {
    const uint32_t item = list ? list[wave] : wave;
    rcur = place_perm ? place_perm[item] : item;
    w = win[item]; off = pair_off[item]; if (slot_of) po = offp[rcur];
  }
  for (uint32_t it = wave; it < n_items; it += nwaves) {
    const uint32_t r = rcur;
    const uint32_t itn = it + nwaves;
    uint4 wn = make_uint4(0, 0, 0, 0);
    uint32_t offn = 0, pon = 0;
    if (itn < n_items) {
      const uint32_t item = list ? list[itn] : itn;
      rcur = place_perm ? place_perm[item] : item;
      wn = win[item]; offn = pair_off[item]; if (slot_of) pon = offp[rcur];
    }
    const uint32_t lo = w.x, ub = w.y;
    const unsigned long long m0 = ((unsigned long long)w.w << 32) | w.z;  // hits of the first chunk, from the count pass
    const uint32_t off_r = off, po_r = po;
    w = wn;
    off = offn;
    po = pon;
    if (lo >= ub) continue;
    if (ub - lo <= 64u) {
      // the whole window is one chunk (the common case): no column is re-read
      const bool hit = (m0 >> lane) & 1ull;
      if (v.sorted_order) {  // visit order == segment order: plain stream compaction
        if (hit) {
          const uint32_t k = __popcll(m0 & lanemask_lt());
          const uint32_t pos = off_r + k;
          pair_range[pos] = r;
          if (pair_entry) pair_entry[pos] = lo + lane;
          if (slot_of) { slot_of[po_r + k] = pos; pl.range[po_r + k] = r; pl.entry[po_r + k] = lo + lane; }
        }
        continue;
      }
      // visit order = ascending rank[]: a 64-lane bitonic sort of (rank, lane) --
      // 21 compare-exchange stages through the cross-lane network, no scalar loop
      // (the first cut counted smaller ranks with one v_readlane round per hit:
      // 350 VALU per range, half the kernel stalled on the SGPR hand-offs).
      // Non-hits carry rank 0xFFFFFFFF and sink to the end; ranks of hits are distinct.
      uint32_t rk = hit ? v.rank[lo + lane] : 0xFFFFFFFFu;
      uint32_t who = lane;
#pragma unroll
      for (unsigned k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
          const uint32_t prk = (uint32_t)__shfl_xor((int)rk, (int)j);
          const uint32_t pwho = (uint32_t)__shfl_xor((int)who, (int)j);
          const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
          const bool take = keep_min ? prk < rk : prk > rk;
          rk = take ? prk : rk;
          who = take ? pwho : who;
        }
      }
      if (lane < (unsigned)__popcll(m0)) {  // lane i now holds the i-th hit in visit order: coalesced stores
        pair_range[off_r + lane] = r;
        if (pair_entry) pair_entry[off_r + lane] = lo + who;
        if (slot_of) { slot_of[po_r + lane] = off_r + lane; pl.range[po_r + lane] = r; pl.entry[po_r + lane] = lo + who; }
      }
      continue;
    }
    const uint32_t off = off_r;  // (shadows the pipelined register inside the rare path)
    // ---- windows wider than one wave (dense targets): chunked -----------------------
    const int32_t qs = fr[r].start;
    if (v.sorted_order) {
      uint32_t run = 0;
      for (uint32_t base = lo; base < ub; base += 64u) {
        uint32_t i = base + lane;
        bool hit = i < ub && window_hit<TRANSITIVE>(ecol[i], qs);
        unsigned long long m = __ballot(hit);
        if (hit) {
          uint32_t k = run + __popcll(m & lanemask_lt());
          uint32_t pos = off + k;
          pair_range[pos] = r;
          if (pair_entry) pair_entry[pos] = i;
          if (slot_of) { slot_of[po_r + k] = pos; pl.range[po_r + k] = r; pl.entry[po_r + k] = i; }
        }
        run += __popcll(m);
      }
      continue;
    }
    for (uint32_t base = lo; base < ub; base += 64u) {
      uint32_t i = base + lane;
      bool hit = i < ub && window_hit<TRANSITIVE>(ecol[i], qs);
      uint32_t rk = hit ? v.rank[i] : 0xFFFFFFFFu;
      uint32_t pos = 0;
      for (uint32_t b2 = lo; b2 < ub; b2 += 64u) {
        uint32_t i2 = b2 + lane;
        bool hit2;
        uint32_t rk2;
        if (b2 == base) { hit2 = hit; rk2 = rk; }
        else {
          hit2 = i2 < ub && window_hit<TRANSITIVE>(ecol[i2], qs);
          rk2 = hit2 ? v.rank[i2] : 0xFFFFFFFFu;
        }
        unsigned long long m = __ballot(hit2);
        while (m) {
          const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
          m &= m - 1;
          const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rk2, j);
          pos += rj < rk;
        }
      }
      if (hit) {
        pair_range[off + pos] = r;
        if (pair_entry) pair_entry[off + pos] = i;
        if (slot_of) { slot_of[po_r + pos] = off + pos; pl.range[po_r + pos] = r; pl.entry[po_r + pos] = i; }
      }
    }
  }
}

// K1b-wide: the listed wide windows (dense targets, repeat hot spots: `bench.py --workload skewed` has windows of thousands of
// entries), a BLOCK per range: the window's hits are collected as (visit rank, entry) keys in LDS, sorted there (bitonic), and
// written out in visit order -- O(W + H log^2 H) per range.  (The wave-per-range kernel above ranks every hit against every
// other with a readlane loop, O(W x H): 98 % of a skewed step, 3.2 s.)  More than WIDE_CAP hits: the range goes onto
// the overflow list and the kernel above takes it.
constexpr uint32_t WIDE_CAP = 4096, WIDE_BINS = 1024;
template <bool TRANSITIVE>
__global__ __launch_bounds__(256) void lookup_emit_wide_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr, uint32_t n,
                                                               const uint32_t *__restrict__ pair_off, const uint4 *__restrict__ win,
                                                               uint32_t *__restrict__ pair_range, uint32_t *__restrict__ pair_entry,
                                                               const uint32_t *__restrict__ offp, ProjList pl, const uint32_t *__restrict__ list,
                                                               const uint32_t *__restrict__ list_n, const uint32_t *__restrict__ place_perm,
                                                               uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_n) {
  __shared__ unsigned long long keys[WIDE_CAP];
  __shared__ uint32_t hist[WIDE_BINS];
  __shared__ uint16_t gb[WIDE_BINS + 2];  // group g = rank bins [gb[g], gb[g + 1])
  __shared__ uint32_t s_cnt, s_ng;
  const int32_t *ecol = end_col<TRANSITIVE>(v);
  uint32_t *const slot_of = pl.slot;
  const uint32_t n_items = min(*list_n, n), tid = threadIdx.x;
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const uint32_t item = list[it];
    const uint32_t r = place_perm ? place_perm[item] : item;
    const uint4 w = win[item];
    const uint32_t off = pair_off[item], po = slot_of ? offp[r] : 0u;
    const uint32_t lo = w.x, ub = w.y;
    const int32_t qs = fr[r].start;
    // a hit's sort key: its visit rank (its place in the window under the sorted order policy), below `dom`
    const uint32_t dom = v.sorted_order ? ub - lo : max(v.max_seg, 1u);
    uint32_t shift = 0;
    while ((dom >> shift) > WIDE_BINS) shift++;
    if (dom >> shift == WIDE_BINS && (dom & ((1u << shift) - 1u))) shift++;  // (every key >> shift below WIDE_BINS)
    // the window's hits whose rank bin lies in [b0, b1), in visit order, to the range's slots from `at` on; returns their number
    // (or more than WIDE_CAP, nothing written: the caller splits the bins)
    auto emit_bins = [&](uint32_t b0, uint32_t b1, uint32_t at) -> uint32_t {
      if (tid == 0) s_cnt = 0u;
      __syncthreads();
      for (uint32_t base = lo; base < ub; base += 256u) {
        const uint32_t i = base + tid;
        bool hit = i < ub && window_hit<TRANSITIVE>(ecol[i], qs);
        uint32_t rk = 0;
        if (hit) {
          rk = v.sorted_order ? i - lo : v.rank[i];
          const uint32_t bin = rk >> shift;
          hit = bin >= b0 && bin < b1;
        }
        const unsigned long long m = __ballot(hit);
        uint32_t wb = 0;
        if (lane_id() == 0 && m) wb = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
        wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb);
        const uint32_t pos = wb + (uint32_t)__popcll(m & lanemask_lt());
        if (hit && pos < WIDE_CAP) keys[pos] = ((unsigned long long)rk << 32) | i;
      }
      __syncthreads();
      const uint32_t H = s_cnt;
      if (H > WIDE_CAP) return H;
      uint32_t P2 = 64u;
      while (P2 < H) P2 <<= 1;
      for (uint32_t k = H + tid; k < P2; k += 256u) keys[k] = ~0ull;
      __syncthreads();
      for (uint32_t k = 2u; k <= P2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
          for (uint32_t t = tid; t < (P2 >> 1); t += 256u) {
            const uint32_t a = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), b = a | j;  // the t-th pair at distance j
            const bool up = (a & k) == 0u;
            const unsigned long long x = keys[a], y = keys[b];
            if ((x > y) == up) { keys[a] = y; keys[b] = x; }
          }
          __syncthreads();
        }
      }
      for (uint32_t k = tid; k < H; k += 256u) {
        const uint32_t e = (uint32_t)keys[k];
        pair_range[off + at + k] = r;
        if (pair_entry) pair_entry[off + at + k] = e;
        if (slot_of) { slot_of[po + at + k] = off + at + k; pl.range[po + at + k] = r; pl.entry[po + at + k] = e; }
      }
      __syncthreads();
      return H;
    };
    const uint32_t H = emit_bins(0u, WIDE_BINS, 0u);
    if (H <= WIDE_CAP) continue;
    // more hits than the buffer takes (a thousand-fold repeat): the rank bins' histogram, bins gathered into groups of at most
    // WIDE_CAP hits, a collect-sort-write pass per group -- the window is read once more per group
    for (uint32_t b = tid; b < WIDE_BINS; b += 256u) hist[b] = 0u;
    __syncthreads();
    for (uint32_t base = lo; base < ub; base += 256u) {
      const uint32_t i = base + tid;
      if (i < ub && window_hit<TRANSITIVE>(ecol[i], qs)) atomicAdd(&hist[(v.sorted_order ? i - lo : v.rank[i]) >> shift], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t ng = 0, acc = 0;
      bool bad = false;
      gb[0] = 0;
      for (uint32_t b = 0; b < WIDE_BINS; b++) {
        const uint32_t c = hist[b];
        bad = bad || c > WIDE_CAP;
        if (acc + c > WIDE_CAP) { ng++; gb[ng] = (uint16_t)b; acc = 0; }
        acc += c;
      }
      ng++;
      gb[ng] = (uint16_t)WIDE_BINS;
      s_ng = bad ? 0u : ng;
    }
    __syncthreads();
    const uint32_t ng = s_ng;
    if (!ng) {  // one bin alone overflows the buffer: the wave-per-range kernel's turn (correct, quadratic)
      if (tid == 0) over_list[atomicAdd(over_n, 1u)] = item;
      __syncthreads();
      continue;
    }
    uint32_t at = 0;
    for (uint32_t g = 0; g < ng; g++) at += emit_bins(gb[g], gb[g + 1u], at);
  }
}

// ---------------------------------------------------------------------------
// multi-GPU routing: owner rank of a frontier record = owner[target_id] (shard map), else target_id % world
// ---------------------------------------------------------------------------
constexpr uint32_t ROUTE_MAX_WORLD = 1024;
__global__ __launch_bounds__(256) void route_keys_kernel(const FrontierRec *__restrict__ fr, uint32_t n, uint32_t world,
                                                         const uint32_t *__restrict__ owner, uint32_t n_seq,
                                                         uint32_t *__restrict__ key, uint32_t *__restrict__ idx,
                                                         unsigned long long *__restrict__ hist) {
  __shared__ uint32_t h[ROUTE_MAX_WORLD];
  for (uint32_t k = threadIdx.x; k < world; k += 256u) h[k] = 0;
  __syncthreads();
  // (grid-stride: a bounded number of blocks, so that the closing atomics on the `world` global counters -- which
  // serialise at one L2 channel -- are a few thousand, not one per 256 records: 0.32 ms per call before this)
  for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
    const uint32_t i = base + threadIdx.x;
    uint32_t o = 0xFFFFFFFFu;
    if (i < n) {
      // owner table of the shard map (targets bin-packed by entry count); a sequence id the index does not know
      // has no alignments anywhere: any rank may answer "no hits"
      const uint32_t t = fr[i].target_id;
      o = (owner && t < n_seq) ? owner[t] : t % world;
      key[i] = o;
      idx[i] = i;
    }
    // one LDS atomic per owner present in the wave, not one per record
    unsigned long long left = __ballot(o != 0xFFFFFFFFu);
    while (left) {
      const uint32_t o0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)o, (int)(__ffsll((long long)left) - 1)));
      const unsigned long long m = __ballot(o == o0);
      if (lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(&h[o0], (uint32_t)__popcll(m));
      left &= ~m;
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < world; k += 256u)
    if (h[k]) atomicAdd(&hist[k], (unsigned long long)h[k]);
}
__global__ __launch_bounds__(256) void route_gather_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ perm,
                                                           uint32_t n, FrontierRec *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = perm[i];
  FrontierRec f = fr[src];
  f.qidx = src;  // the home index the owner echoes back
  out[i] = f;
}

// a level's frontier in its lookup order: what a kept fused level's pair_range indexes (Engine::run, fuse_range_places)
__global__ __launch_bounds__(256) void frontier_gather_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ perm,
                                                              uint32_t n, FrontierRec *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] = fr[perm[i]];
}

// Home side of a hop: hit records arrive grouped by owner rank, every owner's block in ascending fidx (the index
// of the frontier record at home).  A frontier record lives on exactly one owner, so the runs of equal fidx never
// interleave and the stable order by fidx is a counting pass over the frontier indices, not a sort:
// run bounds -> lengths -> exclusive scan -> every record lands at off[fidx] + its place in its run.
__global__ __launch_bounds__(256) void reorder_runs_kernel(const uint32_t *__restrict__ hits, uint32_t n, uint32_t words,
                                                           uint32_t n_front, uint32_t *__restrict__ run_start,
                                                           uint32_t *__restrict__ run_len, uint32_t *__restrict__ err) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = hits[(size_t)i * words];
  if (f >= n_front) { *err = 1; return; }
  // the run's first record notes where it starts, its last one where it ends (run_len holds the END until
  // run_lengths_kernel turns it into a length): no thread walks a run
  if (i == 0 || hits[(size_t)(i - 1) * words] != f) run_start[f] = i;
  if (i + 1 == n || hits[(size_t)(i + 1) * words] != f) {
    if (atomicExch(&run_len[f], i + 1u) != 0u) *err = 1;  // a second run for the same frontier record
  }
}
__global__ __launch_bounds__(256) void run_lengths_kernel(const uint32_t *__restrict__ run_start, uint32_t *__restrict__ run_len, uint32_t n_front) {
  const uint32_t f = blockIdx.x * 256u + threadIdx.x;
  if (f >= n_front) return;
  const uint32_t e = run_len[f];
  if (e) run_len[f] = e - run_start[f];
}
// Lookup / projection order: ranges sorted by where their window will be in the entry array,
// estimated before any search from the record alone: segment start + start / sequence length x
// segment size (alignments spread evenly enough for a LOCALITY key; exactness is not needed).
// bounds (optional): the records come in n_blocks contiguous blocks (records [bounds[b], bounds[b+1]) -- on a shard,
// what each home rank sent); the block goes above the window bits, so the order is block by block and a block's
// pairs stay together.
// where a range's window will be in the entry array, estimated from the record alone (the lookup order's key)
#ifndef IMPG_ORDER_KEY_F64
#define IMPG_ORDER_KEY_F64 1
#endif
__device__ __forceinline__ uint32_t order_key(const SegDesc *__restrict__ seg, const int32_t *__restrict__ seq_len, uint32_t n_seq,
                                              uint32_t target_id, int32_t start) {
  if (target_id >= n_seq) return 0u;
  const uint2 d = *reinterpret_cast<const uint2 *>(seg + target_id);  // {a, n}
  const int32_t len = seq_len[target_id];
  const uint32_t st = (uint32_t)max(start, 0);
  uint64_t rel = 0;
  if (len > 0) {
    // floor(st * n / len): a 64-bit integer division is a long sequence per record; in double precision the product is
    // exact below 2^53 (a 31-bit start, a segment of < 2^22 entries), the quotient correctly rounded, and a quotient that
    // is no integer lies at least 1 / len from one -- so the truncation is the integer division's result
    if (IMPG_ORDER_KEY_F64 && d.y < (1u << 22)) rel = (uint64_t)((double)st * (double)d.y / (double)(uint32_t)len);
    else rel = (uint64_t)st * d.y / (uint64_t)(uint32_t)len;
  }
  return d.x + (uint32_t)min(rel, (uint64_t)(d.y ? d.y - 1u : 0u));
}
// (Round 5: frontier_emit writes the next level's keys beside the records -- with the 64-bit division the key cost the same
// wherever it ran (lookup 5.65 -> 5.25 ms, update 6.03 -> 6.36) and the fusion was dropped; with the double division it pays.)
__global__ __launch_bounds__(256) void order_keys_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr, uint32_t n,
                                                         uint32_t *__restrict__ key, uint32_t *__restrict__ idx,
                                                         const uint32_t *__restrict__ bounds, uint32_t n_blocks, uint32_t block_shift) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= n) return;
  const FrontierRec f = fr[r];
  uint32_t k = order_key(v.seg, v.seq_len, v.n_seq, f.target_id, f.start);
  if (bounds) {
    uint32_t lo = 0, hi = n_blocks;  // last block b with bounds[b] <= r
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (bounds[mid] <= r) lo = mid; else hi = mid;
    }
    k |= lo << block_shift;
  }
  key[r] = k;
  idx[r] = r;
}
__global__ __launch_bounds__(256) void scatter_u32_kernel(const uint32_t *__restrict__ in, const uint32_t *__restrict__ perm,
                                                          uint32_t n, uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[perm[i]] = in[i];
}

// ---------------------------------------------------------------------------
// K1b': the same for windows of at most 64 entries (counted from the 16-byte
// boundary below lo), one LANE per range.  The wave-per-range kernel above spends
// ~300 instructions per range on a 21-stage cross-lane bitonic sort for ~26 hits;
// here a lane keeps its window's 64 keys (rank << 6 | position, all-ones for a
// non-hit) in registers, sorts them with a fully unrolled network (min/max pairs,
// static register indices, ~20 instructions per range), and the wave stages the
// sorted runs in LDS so that the stores stay coalesced.  Needs ranks < 2^26.
// ---------------------------------------------------------------------------
constexpr uint32_t EMIT_LDS_SLOTS = 64u * 64u;  // a lane emits at most 64 pairs
// N = the network's size (a power of two), W <= N = the places that can hold a key
template <uint32_t N, uint32_t W>
__device__ __forceinline__ void emit_sort(uint32_t (&key)[64]) {
#pragma unroll
  for (uint32_t p = 1; p < N; p <<= 1) {
#pragma unroll
    for (uint32_t k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (uint32_t j = k % p; j + k <= N - 1; j += 2 * k) {
#pragma unroll
        for (uint32_t i = 0; i <= min(k - 1, N - j - k - 1); i++) {
          const uint32_t a = i + j, b = i + j + k;
          if (a / (2 * p) == b / (2 * p) && b < W) {
            // one compare-exchange = v_max_u32 + v_min_u32.  A volatile asm pair so that the exchanges stay in program
            // order: left to itself the scheduler interleaves a whole stage and keeps its inputs and outputs live
            // together (140 VGPRs, 3 waves per SIMD).  The minimum replaces its first input in place, the maximum
            // takes the one new register, and the second input's register is free again.
            uint32_t mn = key[a], mx;
            asm volatile("v_max_u32 %1, %0, %2\n\tv_min_u32 %0, %0, %2" : "+v"(mn), "=&v"(mx) : "v"(key[b]));
            key[a] = mn;
            key[b] = mx;
          }
        }
      }
    }
  }
}
template <uint32_t N, uint32_t W>
__device__ __forceinline__ void emit_keys_sort_stage(const uint32_t *__restrict__ rank, uint32_t b, uint32_t lo, uint32_t ub,
                                                     unsigned long long mask, uint32_t c, uint32_t loff, uint32_t lane, uint16_t *stage) {
  uint32_t key[64];
#pragma unroll
  for (uint32_t q = 0; q < (W + 3u) / 4u; q++) {
    uint4 rk = make_uint4(0, 0, 0, 0);
    if (b + 4u * q < ub) rk = *reinterpret_cast<const uint4 *>(rank + b + 4u * q);
    const uint32_t rv[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
      const uint32_t i = 4u * q + t, pos = b + i;
      const bool hit = pos >= lo && pos < ub && ((mask >> ((pos - lo) & 63u)) & 1ull);
      key[i] = hit ? (rv[t] << 6) | i : 0xFFFFFFFFu;
    }
    if ((q & 3u) == 3u) __builtin_amdgcn_sched_barrier(0);  // four rank vectors in flight at a time, not sixteen
  }
  emit_sort<N, W>(key);
#pragma unroll
  for (uint32_t k = 0; k < W; k++) {
    if (__ballot(k < c) == 0ull) break;
    if (k < c) stage[loff + k] = (uint16_t)((lane << 6) | (key[k] & 63u));
  }
}
template <bool TRANSITIVE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void lookup_emit_lane_kernel(DeviceIndexView v, uint32_t n,
                                                              const uint32_t *__restrict__ pair_off,
                                                              const uint4 *__restrict__ win,
                                                              uint32_t *__restrict__ pair_range,
                                                              uint32_t *__restrict__ pair_entry,
                                                              const uint32_t *__restrict__ perm,
                                                              const uint32_t *__restrict__ offp,
                                                              ProjList pl, int by_place) {
  uint32_t *const slot_of = pl.slot;
  __shared__ uint16_t stage[EMIT_LDS_SLOTS];
  const unsigned lane = threadIdx.x;
  const uint32_t i = blockIdx.x * 64u + lane;
  uint32_t r = 0, lo = 0, ub = 0, off = 0, po = 0;
  unsigned long long mask = 0;
  if (i < n) {
    r = perm ? perm[i] : i;  // (with the lookup order a wave's 64 ranges are neighbours in the entry array
                             //  and in slot_of, where offp[] are then consecutive)
    const uint32_t o = by_place ? i : r;  // (by_place: the count pass left its results at the lane's place)
    const uint4 w = win[o];
    lo = w.x; ub = w.y;
    mask = ((unsigned long long)w.w << 32) | w.z;
    off = pair_off[o];
    if (slot_of) po = offp[r];
  }
  const uint32_t b = lo & ~3u;
  const bool mine = lo < ub && ub - b <= 64u;  // wider windows belong to the wave-per-range kernel
  if (!mine) { ub = b; mask = 0; }
  // The widest window of the wave picks the network: Batcher's odd-even merge sort, whose comparators all put the
  // smaller key at the lower index -- so with +inf beyond that width those places never change, and every comparator
  // that touches one is dropped at compile time: 543 exchanges for 64 places, 384 for 48, 305 for 40, 191 for 32 (the
  // bitonic network this replaces: 672 whatever the width).
  const uint32_t wmax = wave_max_u32(mine ? ub - b : 0u);
  const uint32_t c = (uint32_t)__popcll(mask);
  // LDS offsets of the lanes' runs: exclusive scan of c over the wave
  const uint32_t inc = wave_incl_scan(c);
  const uint32_t loff = inc - c;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  // (each width loads, sorts AND stages inside its own branch: no key register is live where the branches part or
  // meet, which pins all 64 of them to the same registers on every path -- 140 VGPRs instead of 93)
  if (wmax <= 32u) emit_keys_sort_stage<32, 32>(v.rank, b, lo, ub, mask, c, loff, lane, stage);
  else if (wmax <= 40u) emit_keys_sort_stage<64, 40>(v.rank, b, lo, ub, mask, c, loff, lane, stage);
  else if (wmax <= 48u) emit_keys_sort_stage<64, 48>(v.rank, b, lo, ub, mask, c, loff, lane, stage);
  else emit_keys_sort_stage<64, 64>(v.rank, b, lo, ub, mask, c, loff, lane, stage);
  __syncthreads();
  for (uint32_t tb = 0; tb < total; tb += 64u) {  // every lane takes part in the cross-lane reads
    const uint32_t t = tb + lane;
    const bool live = t < total;
    const uint32_t sv = live ? stage[t] : 0u, ln = sv >> 6, rel = sv & 63u;
    const uint32_t lb = (uint32_t)__shfl((int)b, (int)ln), lof = (uint32_t)__shfl((int)off, (int)ln);
    const uint32_t llo = (uint32_t)__shfl((int)loff, (int)ln), lpo = (uint32_t)__shfl((int)po, (int)ln);
    const uint32_t lr = (uint32_t)__shfl((int)r, (int)ln);
    if (live) {
      const uint32_t slot = lof + (t - llo);
      pair_range[slot] = lr;
      if (pair_entry) pair_entry[slot] = lb + rel;
      if (slot_of) {  // (consecutive t are consecutive places when the lanes follow the lookup order: coalesced)
        const uint32_t place = lpo + (t - llo);
        slot_of[place] = slot;
        pl.range[place] = lr;
        pl.entry[place] = lb + rel;
      }
    }
  }
}

// K1b'': the visit order WITHOUT the pair lists (ordered rows written by a fused final level, OrderedOut): the same
// per-lane sorting network, but what leaves the kernel is one byte per hit -- the visit position of a range's k-th hit
// in index order, at the hit's place pair_off[i] + k -- 1 byte a pair instead of the 8 of pair_range + pair_entry.
template <uint32_t N, uint32_t W, bool BY_VISIT>  // BY_VISIT: out[k] = the mask bit of the k-th visited hit (the inverse map)
__device__ __forceinline__ void emit_vpos_sort_stage(const uint32_t *__restrict__ rank, uint32_t b, uint32_t lo, uint32_t ub,
                                                     unsigned long long mask, uint32_t c, uint8_t *out) {
  uint32_t key[64];
#pragma unroll
  for (uint32_t q = 0; q < (W + 3u) / 4u; q++) {
    uint4 rk = make_uint4(0, 0, 0, 0);
    if (b + 4u * q < ub) rk = *reinterpret_cast<const uint4 *>(rank + b + 4u * q);
    const uint32_t rv[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
      const uint32_t i = 4u * q + t, pos = b + i;
      const bool hit = pos >= lo && pos < ub && ((mask >> ((pos - lo) & 63u)) & 1ull);
      key[i] = hit ? (rv[t] << 6) | i : 0xFFFFFFFFu;
    }
    if ((q & 3u) == 3u) __builtin_amdgcn_sched_barrier(0);
  }
  emit_sort<N, W>(key);
  const uint32_t shift = lo - b;  // (window position i = b + i; mask bit = position - lo)
#pragma unroll
  for (uint32_t k = 0; k < W; k++) {
    if (__ballot(k < c) == 0ull) break;
    if (k < c) {
      const uint32_t bit = (key[k] & 63u) - shift;
      if (BY_VISIT) out[k] = (uint8_t)bit;
      else out[__popcll(mask & ((1ull << bit) - 1ull))] = (uint8_t)k;
    }
  }
}
constexpr uint32_t VPOS_LDS = 64u * 64u + 8u;
template <bool BY_VISIT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void emit_vpos_lane_kernel(DeviceIndexView v, uint32_t n,
                                                              const uint32_t *__restrict__ pair_off, const uint4 *__restrict__ win,
                                                              uint8_t *__restrict__ vpos, OrdDestArgs od) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[VPOS_LDS];
  const unsigned lane = threadIdx.x;
  const uint32_t i = blockIdx.x * 64u + lane;
  uint32_t lo = 0, ub = 0, off = 0;
  unsigned long long mask = 0;
  if (i < n) {
    const uint4 w = win[i];
    lo = w.x; ub = w.y;
    mask = ((unsigned long long)w.w << 32) | w.z;
    off = pair_off[i];
    // (the range's first row -- offsets[query] + the level's base + the record's slots -- gathered under the sort: a kernel of its own took 1.5 ms for it)
    if (od.dest) { const uint32_t q = od.frp[i].qidx; od.dest[i] = od.offsets[q] + od.lvbase[q] + od.slot_ref[od.perm[i]]; }
  }
  const uint32_t b = lo & ~3u;
  const bool mine = lo < ub && ub - b <= 64u;  // (wider windows: the wave-per-range emit lists their entries in visit order)
  if (!mine) { ub = b; mask = 0; }
  const uint32_t wmax = wave_max_u32(mine ? ub - b : 0u);
  const uint32_t c = (uint32_t)__popcll(mask);
  // the wave's places are one contiguous piece of vpos[] starting at lane 0's offset; a lane's run is staged in LDS at the
  // same distance from there (+ the piece's misalignment, so that whole words line up), or written straight out when a
  // wide window's run in between pushes it past the buffer
  const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
  const uint32_t mis = off0 & 3u;
  const uint32_t rel = off - off0 + mis;
  const bool in_lds = rel + c <= VPOS_LDS;
  uint8_t *out = in_lds ? stage + rel : vpos + off;
  if (wmax <= 32u) emit_vpos_sort_stage<32, 32, BY_VISIT>(v.rank, b, lo, ub, mask, c, out);
  else if (wmax <= 40u) emit_vpos_sort_stage<64, 40, BY_VISIT>(v.rank, b, lo, ub, mask, c, out);
  else if (wmax <= 48u) emit_vpos_sort_stage<64, 48, BY_VISIT>(v.rank, b, lo, ub, mask, c, out);
  else emit_vpos_sort_stage<64, 64, BY_VISIT>(v.rank, b, lo, ub, mask, c, out);
  __syncthreads();
  // copy out: LDS bytes [mis, covered) are the piece's first covered - mis bytes (a wide window's run in between holds
  // whatever was there: nobody reads those bytes).  Whole aligned words, except where the piece's first / last word is
  // shared with a neighbouring wave's piece.
  const uint32_t covered = wave_max_u32(in_lds ? rel + c : 0u);
  uint8_t *g0 = vpos + (off0 - mis);
  for (uint32_t d = lane; 4u * d < covered; d += 64u) {
    const uint32_t lo_b = 4u * d, hi_b = lo_b + 4u;
    if (lo_b >= mis && hi_b <= covered) *reinterpret_cast<uint32_t *>(g0 + lo_b) = *reinterpret_cast<const uint32_t *>(stage + lo_b);
    else for (uint32_t j = max(lo_b, mis); j < min(hi_b, covered); j++) g0[j] = stage[j];
  }
}

// ---------------------------------------------------------------------------
// exclusive scan of u32 (3 phases; offsets u32, grand total u64)
// ---------------------------------------------------------------------------
constexpr uint32_t SCAN_ITEMS = 8, SCAN_BLOCK = 256, SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

// block-wide exclusive scan of one value per thread (256 threads); returns the
// exclusive prefix and, in *total, the block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t *total) {
  __shared__ uint32_t wsum[4];
  uint32_t inc = wave_incl_scan(x);
  unsigned w = threadIdx.x >> 6;
  if (lane_id() == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (unsigned k = 0; k < w; k++) base += wsum[k];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - x;
}

// the same for a block of NW waves
template <unsigned NW>
__device__ __forceinline__ uint32_t block_excl_scan_n(uint32_t x, uint32_t *wsum /* [NW] shared */) {
  const uint32_t inc = wave_incl_scan(x);
  const unsigned w = threadIdx.x >> 6;
  if (lane_id() == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (unsigned k = 0; k < NW; k++) base += k < w ? wsum[k] : 0u;
  __syncthreads();
  return base + inc - x;
}

__global__ __launch_bounds__(256) void scan_block_sums(const uint32_t *__restrict__ in, uint32_t n,
                                                       unsigned long long *__restrict__ bsum) {
  uint32_t base = blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    uint32_t i = base + k * SCAN_BLOCK + threadIdx.x;
    if (i < n) s += in[i];
  }
  uint32_t tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// single block: exclusive scan of the block sums in place, grand total to *total.  A thread owns one contiguous stretch
// of the sums: it adds them up, the 1 024 stretch totals are scanned once across the block, and it writes its stretch's
// running prefixes back.  (Until round 4: 256 sums per turn through a 16-barrier shared-memory scan -- 150 turns for the
// 4 x 10^4 block sums of a level's pairs, 0.5 ms of a headline step in twelve scans.)
__global__ __launch_bounds__(1024) void scan_of_sums(unsigned long long *bsum, uint32_t nb, unsigned long long *total) {
  __shared__ unsigned long long wsum[16];
  const uint32_t per = (nb + 1023u) / 1024u, a = min(nb, threadIdx.x * per), e = min(nb, a + per);
  unsigned long long mine = 0;
  for (uint32_t i = a; i < e; i++) mine += bsum[i];
  // exclusive scan of `mine` over the block: within the wave by shuffles, then over the 16 waves' totals
  unsigned long long inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(inc, d);
    if ((int)lane_id() >= d) inc += y;
  }
  if (lane_id() == 63u) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned long long base = 0;
  for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) base += wsum[k];
  unsigned long long run = base + inc - mine;
  for (uint32_t i = a; i < e; i++) { const unsigned long long x = bsum[i]; bsum[i] = run; run += x; }
  if (threadIdx.x == 1023u) *total = run;
}
__global__ __launch_bounds__(256) void scan_apply(const uint32_t *__restrict__ in, uint32_t n,
                                                  const unsigned long long *__restrict__ bsum,
                                                  uint32_t *__restrict__ out) {
  // thread t owns items [t*ITEMS, t*ITEMS+ITEMS) of the block tile
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t x[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    uint32_t i = base + k;
    x[k] = i < n ? in[i] : 0;
    s += x[k];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, &tot) + (uint32_t)bsum[blockIdx.x];
#pragma unroll
  for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
    uint32_t i = base + k;
    if (i < n) out[i] = ex;
    ex += x[k];
  }
}

// ---------------------------------------------------------------------------
// The same scan in ONE pass over the data (round 5): a tile's block publishes its sum, then looks back over the tiles
// before it -- a wave reads 64 tile words at a time -- until it meets one whose inclusive prefix is known (decoupled
// look-back).  Tiles are handed out by an atomic ticket, so every tile a block waits for was started before it.
// state[t] = flag << 62 | value (flag 0: nothing yet, 1: the tile's own sum, 2: the sum of everything up to and
// including the tile); state[n_tiles] = the ticket.  A tile's own sum is a 32-bit sum like scan_block_sums' (counts
// per range are small); prefixes are 64-bit, the offsets written are their low 32 bits and *total is exact.
// (n < 2^32 - 4096, like the three-kernel form: a tile's lane offsets are 32-bit; the engine's counts stay far below.)
// The three kernels above read the counts twice and ran the level's 8 x 10^7-range scan at 2.3 TB/s (282 + 89 + 72 us);
// IMPG_SCAN_LOOKBACK = 0 brings them back.
// ---------------------------------------------------------------------------
#ifndef IMPG_SCAN_LOOKBACK
#define IMPG_SCAN_LOOKBACK 1
#endif
constexpr uint32_t LB_ITEMS = 16, LB_BLOCK = 256, LB_TILE = LB_ITEMS * LB_BLOCK;
constexpr unsigned long long LB_FLAG_SUM = 1ull << 62, LB_FLAG_INCL = 2ull << 62, LB_VALUE = (1ull << 62) - 1ull;
__global__ __launch_bounds__(LB_BLOCK) void scan_lookback_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                                                                 unsigned long long *state, uint32_t n_tiles,
                                                                 unsigned long long *__restrict__ total) {
  __shared__ uint32_t s_tile;
  __shared__ unsigned long long s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<uint32_t *>(state + n_tiles), 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t base = tile * LB_TILE + threadIdx.x * LB_ITEMS;
  uint32_t x[LB_ITEMS];
  if (base + LB_ITEMS <= n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
    for (uint32_t k = 0; k < LB_ITEMS / 4u; k++) {
      const uint4 v = p[k];
      x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < LB_ITEMS; k++) x[k] = base + k < n ? in[base + k] : 0u;
  }
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < LB_ITEMS; k++) mine += x[k];
  uint32_t tot;
  uint32_t ex = block_excl_scan(mine, &tot);
  if (tile == 0u) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(state, LB_FLAG_INCL | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = 0ull;
    }
  } else if (threadIdx.x < 64u) {
    if (threadIdx.x == 0)
      __hip_atomic_store(state + tile, LB_FLAG_SUM | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long run = 0ull;
    int64_t j = (int64_t)tile - 1;  // the nearest tile not yet accounted for
    for (;;) {
      const int64_t idx = j - (int64_t)threadIdx.x;
      unsigned long long w = LB_FLAG_INCL;  // (before the first tile: an inclusive prefix of zero)
      unsigned long long incl, empty;
      uint32_t first;
      for (;;) {
        if (idx >= 0) w = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        incl = __ballot((w >> 62) == 2ull);
        empty = __ballot((w >> 62) == 0ull);
        first = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;
        const unsigned long long upto = first >= 63u ? ~0ull : (2ull << first) - 1ull;
        if (!(empty & upto)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      unsigned long long v = threadIdx.x <= first ? (w & LB_VALUE) : 0ull;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      run += v;
      if (first < 64u) break;
      j -= 64;
    }
    if (threadIdx.x == 0) {
      s_prefix = run;
      __hip_atomic_store(state + tile, LB_FLAG_INCL | ((run + tot) & LB_VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const unsigned long long prefix = s_prefix;
  ex += (uint32_t)prefix;
  if (base + LB_ITEMS <= n) {
    uint4 *q = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
    for (uint32_t k = 0; k < LB_ITEMS / 4u; k++) {
      uint4 v;
      v.x = ex; ex += x[4 * k];
      v.y = ex; ex += x[4 * k + 1];
      v.z = ex; ex += x[4 * k + 2];
      v.w = ex; ex += x[4 * k + 3];
      q[k] = v;
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < LB_ITEMS; k++) {
      if (base + k < n) out[base + k] = ex;
      ex += x[k];
    }
  }
  if (tile == n_tiles - 1u && threadIdx.x == 0) *total = prefix + tot;
}

void launch_exclusive_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, unsigned long long *d_bsum,
                           unsigned long long *d_total, hipStream_t s) {
  if (IMPG_SCAN_LOOKBACK && ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u) == 0u) {
    const uint32_t nt = std::max<uint32_t>((n + LB_TILE - 1) / LB_TILE, 1u);
    IMPG_HIP(hipMemsetAsync(d_bsum, 0, ((size_t)nt + 1) * sizeof(unsigned long long), s));
    scan_lookback_kernel<<<nt, LB_BLOCK, 0, s>>>(d_in, d_out, n, d_bsum, nt, d_total);
    return;
  }
  uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) nb = 1;
  // note: block sums are computed with a strided ownership, applied with a
  // blocked ownership; both cover the same tile so the sums agree.
  scan_block_sums<<<nb, SCAN_BLOCK, 0, s>>>(d_in, n, d_bsum);
  scan_of_sums<<<1, 1024, 0, s>>>(d_bsum, nb, d_total);
  scan_apply<<<nb, SCAN_BLOCK, 0, s>>>(d_in, n, d_bsum, d_out);
}
size_t scan_scratch_bytes(uint32_t n) { return ((size_t)(n + SCAN_TILE - 1) / SCAN_TILE + 2) * sizeof(unsigned long long); }  // (block sums, or tile words + the ticket)

// ---------------------------------------------------------------------------
// K2: projection, one lane per (range, entry) pair
// ---------------------------------------------------------------------------
struct TileScan {
  bool found;
  bool any;  // an op passed since the caller last cleared it
  int32_t pqs, pts, pqe, pte;  // query values are direction-normalised offsets from qbase until the end
};
// extra state for min_gap_compressed_identity (impg.rs:2952-2973): running sums of
// matched bases ('=' and 'M'), mismatched bases ('X') and gap OPS ('I'/'D') in
// walking order, snapshots at the first / last overlapping op, and what the
// slice adjustment (impg.rs:2879-2886) takes off those two ops.
struct IdentScan {
  uint32_t rm, rx, rg;        // running (inclusive of the ops walked so far)
  uint32_t fm, fx, fg;        // running sums just BEFORE the first overlapping op
  uint32_t lm, lx, lg;        // running sums just AFTER the last overlapping op
  int32_t first_adj_m, first_adj_x;  // first_op_offset if the first op is a match / mismatch op
  int32_t last_adj_m, last_adj_x;    // last_op_remaining (<= 0) if the last op is a match / mismatch op
  // store_cigar (impg.rs:2878-2886): original indices of the first / last overlapping
  // op and the two length adjustments exactly as the reference carries them
  uint32_t first_oi, last_oi;
  int32_t first_off, last_rem;
};
// MODE_WALK: an index without prefix lines -- the two short walks on the op lines (always together with MODE_IDENT)
constexpr int MODE_IDENT = 1, MODE_CIGAR = 2, MODE_WALK = 4;

struct PairCtx {
  int32_t ts, R0, R1, last_tp;
  bool swp, flip;
  uint32_t zt, zq;       // op code with zero target delta / zero query delta in the entry's view
  uint32_t m;            // number of tiles
  uint32_t totT, totQ;   // record totals in the entry's (effective) axes
  const uint32_t *ops;   // first tile line of the record
};

// Exact per-op step of project_target_range_through_alignment (impg.rs:2800-2869)
// on the effective view of an entry.  T = running target position, Qn = running
// query offset from the entry's query base in units of dir (so it only grows).
// Positions never decrease, so "target_pos > last_target_pos => break"
// (impg.rs:2802) is a per-op test.
//   arm 1 (target_delta == 0):  passes iff T >= R0;                          first (Q, T)       last (Q+qd, T)
//   arm 2 (query_delta == 0):   passes iff max(T,R0) < min(T+td,last_tp);    first (Q, os)      last (Q, oe)
//   arm 3:                      passes iff max(T,R0) < min(T+td,R1);         first (Q+os-T, os) last (Q+oe-T, oe)
// When arm 1 passes, os == T and min(T, lim) == T, so os / oe serve all arms.
// PART: 0 = record both ends; 1 = only look for the FIRST overlapping op (walk A of the plain projection);
// 2 = only keep the LAST one (walk B).  A walk that needs one end skips a fifth of the arithmetic.
constexpr int PART_BOTH = 0, PART_FIRST = 1, PART_LAST = 2;
template <int MODE, int PART = PART_BOTH>
__device__ __forceinline__ void op_step(uint32_t op, const PairCtx &c, int32_t &T, int32_t &Qn, TileScan &s, IdentScan &id,
                                        uint32_t oi = 0) {
  constexpr bool IDENT = (MODE & MODE_IDENT) != 0;
  const bool valid = op != OP_PAD;
  const uint32_t code = op >> 29;
  const int32_t len = (int32_t)(op & OP_LEN_MASK);  // (a padding word carries length 0)
  const int32_t td = code == c.zt ? 0 : len;  // target_delta (impg.rs:115-121), I<->D swapped for reversed entries
  const int32_t qa = code == c.zq ? 0 : len;  // |query_delta| (impg.rs:123-135)
  const bool arm1 = td == 0;
  const bool qzero = qa == 0;
  const int32_t lim = qzero ? c.last_tp : c.R1;  // a deletion clips to last_target_pos, a match to the range end
  const int32_t os = max(T, c.R0);
  const int32_t e = T + td;
  const int32_t oe = min(e, lim);
  // One comparison for the three arms: arm 1 passes iff T >= R0; with T <= last_target_pos <= min(R1, last_tp)
  // its oe is T, so that reads os <= oe, i.e. os < oe + 1; the other arms pass iff os < oe.  (Spelt with the
  // arms' own tests, the compiler materialises each as 0/1 and selects between them: eight VALU per op instead
  // of four; with && / ?: it branches around each comparison.)
  const int32_t slack = 1 - min(td, 1);  // 1 for arm 1, else 0 (td >= 0)
  const bool pass = valid & (T <= c.last_tp) & (os < oe + slack);
  const bool first = PART != PART_LAST && (pass & !s.found);
  if (IDENT) {
    const bool is_m = valid && (code == 0u || code == 4u), is_x = valid && code == 1u;
    const bool is_g = valid && (code == 2u || code == 3u);
    if (first) {
      id.fm = id.rm; id.fx = id.rx; id.fg = id.rg;
      const int32_t off = arm1 ? 0 : os - T;  // first_op_offset (impg.rs:2832, :2856)
      id.first_adj_m = is_m ? off : 0;
      id.first_adj_x = is_x ? off : 0;
    }
    id.rm += is_m ? (uint32_t)len : 0u;
    id.rx += is_x ? (uint32_t)len : 0u;
    id.rg += is_g ? 1u : 0u;
    if (pass) {
      id.lm = id.rm; id.lx = id.rx; id.lg = id.rg;
      const int32_t rem = arm1 ? 0 : oe - e;  // last_op_remaining (impg.rs:2838, :2862); an arm-1 last op is a gap op
      id.last_adj_m = is_m ? rem : 0;
      id.last_adj_x = is_x ? rem : 0;
    }
  }
  if (MODE & MODE_CIGAR) {
    if (first) {
      id.first_oi = oi;
      id.first_off = arm1 ? 0 : os - T;  // only the deletion / match arms set it (impg.rs:2832, :2856)
    }
    if (pass) {
      id.last_oi = oi;
      if (!arm1) id.last_rem = oe - e;   // an insertion leaves the previous value in place (impg.rs:2807-2821)
    }
  }
  if (PART != PART_LAST) {
    const int32_t fq = Qn + (qzero ? 0 : os - T);
    s.pqs = first ? fq : s.pqs;
    s.pts = first ? os : s.pts;
    s.found = s.found | pass;
  }
  if (PART != PART_FIRST) {
    const int32_t lq = Qn + ((arm1 || qzero) ? qa : oe - T);
    s.pqe = pass ? lq : s.pqe;
    s.pte = pass ? oe : s.pte;
    s.any = s.any | pass;
  }
  T = e;
  Qn += qa;
}

struct TileHdr {  // the line header, in the ENTRY's axes: sums before the tile, then (relative to those) before its
  uint32_t t0, q0;                                   // sub-tiles 1..3 and after its last op
  uint32_t bt1, bt2, bt3, bt4, bq1, bq2, bq3, bq4;
  bool wide;                                         // no inner sums (a sum overflows 16 bits): walk literally
};
__device__ __forceinline__ TileHdr tile_header(const PairCtx &c, uint32_t j) {
  const uint32_t *line = c.ops + (size_t)j * TILE_WORDS;
  const uint4 a = *reinterpret_cast<const uint4 *>(line);
  const uint2 b = *reinterpret_cast<const uint2 *>(line + 4);
  TileHdr h;
  h.t0 = c.swp ? a.y : a.x; h.q0 = c.swp ? a.x : a.y;
  const uint32_t t12 = c.swp ? b.x : a.z, t34 = c.swp ? b.y : a.w;
  const uint32_t q12 = c.swp ? a.z : b.x, q34 = c.swp ? a.w : b.y;
  h.bt1 = t12 & 0xFFFFu; h.bt2 = t12 >> 16; h.bt3 = t34 & 0xFFFFu; h.bt4 = t34 >> 16;
  h.bq1 = q12 & 0xFFFFu; h.bq2 = q12 >> 16; h.bq3 = q34 & 0xFFFFu; h.bq4 = q34 >> 16;
  h.wide = (a.w >> 16) == TILE_WIDE;
  return h;
}
// sums before original sub-tile s of the tile (s = 4: after the tile), relative to its start
__device__ __forceinline__ uint32_t sub_bt(const TileHdr &h, uint32_t s) {
  return s == 0 ? 0u : s == 1 ? h.bt1 : s == 2 ? h.bt2 : s == 3 ? h.bt3 : h.bt4;
}
__device__ __forceinline__ uint32_t sub_bq(const TileHdr &h, uint32_t s) {
  return s == 0 ? 0u : s == 1 ? h.bq1 : s == 2 ? h.bq2 : s == 3 ? h.bq3 : h.bq4;
}
// running positions where the walk ENTERS original sub-tile s: its start for a
// forward walk, its end (mirrored) for a back-to-front walk.  Not for wide tiles.
__device__ __forceinline__ void sub_start(const PairCtx &c, const TileHdr &h, uint32_t s, int32_t &T, int32_t &Qn) {
  uint32_t t, q;
  if (!c.flip) { t = h.t0 + sub_bt(h, s); q = h.q0 + sub_bq(h, s); }
  else { t = c.totT - (h.t0 + sub_bt(h, s + 1)); q = c.totQ - (h.q0 + sub_bq(h, s + 1)); }
  T = c.ts + (int32_t)t;
  Qn = (int32_t)q;
}
// running positions where the walk enters TILE j (any tile, wide or not)
__device__ __forceinline__ void tile_start(const PairCtx &c, uint32_t j, int32_t &T, int32_t &Qn) {
  const TileHdr h = tile_header(c, j);
  if (!c.flip) { T = c.ts + (int32_t)h.t0; Qn = (int32_t)h.q0; return; }
  uint32_t et, eq;  // sums after the tile
  if (!h.wide) { et = h.t0 + h.bt4; eq = h.q0 + h.bq4; }
  else if (j + 1 < c.m) { const TileHdr n = tile_header(c, j + 1); et = n.t0; eq = n.q0; }
  else { et = c.totT; eq = c.totQ; }
  T = c.ts + (int32_t)(c.totT - et);
  Qn = (int32_t)(c.totQ - eq);
}

// A position of the walk: effective tile k / effective sub-tile he (both counted in
// the entry's walking order) and the running sums at which the walk enters it.
struct Cursor {
  uint32_t k, he;
  int32_t T, Qn;
};
__device__ __forceinline__ uint32_t orig_tile(const PairCtx &c, uint32_t k) { return c.flip ? c.m - 1u - k : k; }
__device__ __forceinline__ uint32_t orig_sub(const PairCtx &c, uint32_t he) { return c.flip ? 3u - he : he; }
__device__ __forceinline__ uint32_t tile_subs(uint32_t n_ops, uint32_t j) { return subs_with_ops(min(TILE_OPS, n_ops - j * TILE_OPS)); }

// Scan the cursor's sub-tile in walking order: its one or two 16-byte vectors are
// requested together, then eight op slots are replayed (sub-tile 0 shares its
// first vector with two header words and sub-tile 3 has a single vector: those
// slots are padding).  Reverse-strand reversed entries walk back to front.  The
// cursor's sums end up at the sub-tile's far end, which is where the next one starts.
template <int MODE, int PART = PART_BOTH>
__device__ __forceinline__ void scan_cur(const PairCtx &c, Cursor &cur, TileScan &s, IdentScan &id) {
  const uint32_t h = orig_sub(c, cur.he);
  const uint4 *q = reinterpret_cast<const uint4 *>(c.ops + (size_t)orig_tile(c, cur.k) * TILE_WORDS);
  const uint32_t v0 = 1u + 2u * h;
  uint4 a = q[v0];
  uint4 b = make_uint4(OP_PAD, OP_PAD, OP_PAD, OP_PAD);
  if (h != 3u) b = q[v0 + 1u];
  if (h == 0u) a.x = a.y = OP_PAD;  // words 4, 5 are header
  const uint4 f = c.flip ? b : a, g = c.flip ? a : b;
  op_step<MODE, PART>(c.flip ? f.w : f.x, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? f.z : f.y, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? f.y : f.z, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? f.x : f.w, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? g.w : g.x, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? g.z : g.y, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? g.y : g.z, c, cur.T, cur.Qn, s, id);
  op_step<MODE, PART>(c.flip ? g.x : g.w, c, cur.T, cur.Qn, s, id);
}
// next sub-tile that holds ops, in walking order; false at the end of the record
__device__ __forceinline__ bool advance(const PairCtx &c, uint32_t n_ops, Cursor &cur) {
  // forward: the sub-tiles without ops are the last ones of the last tile; backward: they were skipped on entry
  const uint32_t lim = c.flip ? TILE_SUBS : tile_subs(n_ops, cur.k);
  if (cur.he + 1u < lim) { cur.he += 1u; return true; }
  if (cur.k + 1u >= c.m) return false;
  cur.k += 1u;   // every tile but the record's last is full
  cur.he = 0u;
  return true;
}
// Put the cursor on effective sub-tile `he` of effective tile k (header h).
__device__ __forceinline__ void place(const PairCtx &c, uint32_t n_ops, const TileHdr &h, uint32_t k, uint32_t he, Cursor &cur) {
  cur.k = k;
  if (h.wide) {  // no inner sums: enter at the tile's first sub-tile with ops
    cur.he = c.flip ? TILE_SUBS - tile_subs(n_ops, orig_tile(c, k)) : 0u;
    tile_start(c, orig_tile(c, k), cur.T, cur.Qn);
  } else {
    cur.he = he;
    sub_start(c, h, orig_sub(c, he), cur.T, cur.Qn);
  }
}
__device__ __forceinline__ void ident_reset(IdentScan &id) {
  id.rm = id.rx = id.rg = id.fm = id.fx = id.fg = id.lm = id.lx = id.lg = 0;
  id.first_adj_m = id.first_adj_x = id.last_adj_m = id.last_adj_x = 0;
  id.first_oi = id.last_oi = 0;
  id.first_off = id.last_rem = 0;
}

// literal walk over whole tiles A..B in effective order, one op at a time (rare path).
// Inlined on purpose: as a call it forced the whole PairCtx through scratch
// memory for EVERY pair (48 B/lane of extra HBM writes), not only the rare ones.
template <int MODE>
__device__ __forceinline__ TileScan walk_tiles(const PairCtx &c, uint32_t A, uint32_t B, IdentScan &id) {
  TileScan s;
  s.found = s.any = false;
  s.pqs = s.pts = s.pqe = s.pte = -1;
  ident_reset(id);
  int32_t T, Qn;
  tile_start(c, A, T, Qn);
  const int stepj = c.flip ? -1 : 1;
  for (int64_t j = A;; j += stepj) {
    const uint32_t *tp = c.ops + (size_t)j * TILE_WORDS + 6;
    for (int u = 0; u < (int)TILE_OPS && T <= c.last_tp; u++) {
      const int w = c.flip ? (int)TILE_OPS - 1 - u : u;
      op_step<MODE>(tp[w], c, T, Qn, s, id, (uint32_t)j * TILE_OPS + (uint32_t)w);
    }
    if (T > c.last_tp || j == (int64_t)B) break;
  }
  return s;
}

// ---------------------------------------------------------------------------
// Plain projection on the prefix lines (impg_internal.hpp): no op is replayed.
//
// In the entry's walking order positions only grow, so with S_k / E_k the target position where op k starts / ends
//   * the FIRST overlapping op is the first op with E_k >= R0 or the one after it: every op before has
//     E < R0, which fails all three arms (an insertion sits at E = S < R0; the others end before R0);
//   * the LAST overlapping op is the last op with S_k <= last_target_pos or the one before it: every op
//     after starts beyond last_target_pos (impg.rs:2802).
// Both are located by comparing prefix entries with the range ends, and the candidates are then put through
// the reference's own per-op test (pfx_eval = op_step on one op, deltas taken from neighbouring entries): a
// candidate that fails hands over to its neighbour, and if that fails too -- inconsistent CIGARs, empty clipped
// ranges -- the pair takes the literal walk.  So the result never rests on the search being right, only its speed.
//
// In STORAGE order (the order of the line) a reverse-strand reversed entry walks back to front, which turns
// "first op with E >= x" into "last op with start prefix <= totT - x" and the other way round; "last k with
// x_k <= t" is "first k with x_{k+1} >= t + 1", so one search serves both: pfx_locate finds the first k of the
// tile with x_{k+1} >= thr and returns the entries around it, NEXT = the candidate after k (k, k + 1), else the
// one before it (k, k - 1).
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
struct PfxTile {
  int32_t bT, bQ;     // T = bT +- x, Qn = bQ +- y (see pfx_eval)
  uint32_t tsh, qsh;  // bit offsets of the entry's target / query halves (swapped for a reversed entry)
};
// exact test of ONE op given the entries before and after it (op_step, arms as there); first-form = what the first
// overlapping op contributes (query offset, target position), else what the last one does
// adj: what the slice adjustment takes off the op (impg.rs:2879-2886): first_op_offset / last_op_remaining as op_step has them
__device__ __forceinline__ bool pfx_eval(const PairCtx &c, const PfxTile &t, uint32_t ea, uint32_t eb, bool first_form,
                                         int32_t &oq, int32_t &ot, int32_t &adj) {
  const int32_t xa = (int32_t)__builtin_amdgcn_ubfe(ea, t.tsh, 16u), xb = (int32_t)__builtin_amdgcn_ubfe(eb, t.tsh, 16u);
  const int32_t ya = (int32_t)__builtin_amdgcn_ubfe(ea, t.qsh, 16u), yb = (int32_t)__builtin_amdgcn_ubfe(eb, t.qsh, 16u);
  const int32_t td = xb - xa, qa = yb - ya;
  const int32_t T = c.flip ? t.bT - xb : t.bT + xa;    // back to front: the op starts where the NEXT prefix mirrors to
  const int32_t Qn = c.flip ? t.bQ - yb : t.bQ + ya;
  const bool qzero = qa == 0;
  const int32_t lim = qzero ? c.last_tp : c.R1;
  const int32_t os = max(T, c.R0);
  const int32_t oe = min(T + td, lim);
  const int32_t slack = 1 - min(td, 1);
  const bool pass = (T <= c.last_tp) & (os < oe + slack);
  const int32_t fq = Qn + (qzero ? 0 : os - T);
  const int32_t lq = Qn + (((td == 0) | qzero) ? qa : oe - T);
  oq = first_form ? fq : lq;
  ot = first_form ? os : oe;
  adj = td == 0 ? 0 : first_form ? os - T : oe - (T + td);
  return pass;
}
// One end of the projection.  j = storage tile to start in, thr_abs = threshold on the record's storage-order
// target prefix.  Returns true with (oq, ot) set, or false = take the literal walk.
// which: the op that answered (storage tile, op within the tile) and its slice adjustment -- the identity filter's inputs
struct PfxOp { uint32_t j, k; int32_t adj; };
template <bool NEXT, uint32_t LINE_STRIDE = TILE_WORDS>  // LINE_STRIDE: words from one line of the record to the next (LDS copies are padded)
__device__ __forceinline__ bool pfx_end(const PairCtx &c, const uint32_t *__restrict__ pfx_rec, uint32_t n_ops, uint32_t j,
                                        int32_t thr_abs, bool first_form, int32_t &oq, int32_t &ot, PfxOp &which, uint4 h0) {
  for (bool fresh = true;; fresh = false) {
    const uint32_t *line = pfx_rec + (size_t)j * LINE_STRIDE;
    // T0 | wide << 31, Q0, entry 9, entry 18: the first tile's header is handed in (the caller requests both ends' headers together)
    uint4 h = h0;
    if (!fresh) h = *reinterpret_cast<const uint4 *>(line);
    // (one 16-byte read: without the pin the compiler reads word 0 alone, tests the flag, and only then the rest)
    asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w));
    const bool wide = (h.x >> 31) != 0u;
    const uint32_t cnt = min(TILE_OPS, n_ops - j * TILE_OPS);
    PfxTile t;
    t.tsh = c.swp ? 16u : 0u;
    t.qsh = 16u - t.tsh;
    const int32_t X0 = (int32_t)(c.swp ? h.y : h.x), Y0 = (int32_t)(c.swp ? h.x : h.y);
    t.bT = c.flip ? c.ts + (int32_t)c.totT - X0 : c.ts + X0;
    t.bQ = c.flip ? (int32_t)c.totQ - Y0 : Y0;
    const int32_t thr = thr_abs - X0;
    uint32_t s = ((int32_t)__builtin_amdgcn_ubfe(h.z, t.tsh, 16u) < thr ? 1u : 0u) + ((int32_t)__builtin_amdgcn_ubfe(h.w, t.tsh, 16u) < thr ? 1u : 0u);
    s = min(s, (cnt - 1u) / PFX_STEP);
    const uint32_t lb = PFX_STEP * s;
    // twelve entries from e_lb (NEXT) or e_(lb-1): r[i + D] = e_(lb+i).  (4-byte aligned 16-byte loads; the last
    // words of the tile's last third run into the next line and are never selected)
    constexpr int D = NEXT ? 0 : 1;
    const uint32_t *q = line + PFX_E0 + lb - (uint32_t)D;
    u32x4_u r0 = *reinterpret_cast<const u32x4_u *>(q), r1 = *reinterpret_cast<const u32x4_u *>(q + 4),
            r2 = *reinterpret_cast<const u32x4_u *>(q + 8);
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2));
    // m = #{i in 1..8 : x_(lb+i) < thr}: bisection over 1..7, then the eighth; the window r[m..m+2] is selected on the way
    const uint32_t r[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
    const bool b2 = (int32_t)__builtin_amdgcn_ubfe(r[4 + D], t.tsh, 16u) < thr;
    const uint32_t uu[6] = {b2 ? r[4] : r[0], b2 ? r[5] : r[1], b2 ? r[6] : r[2], b2 ? r[7] : r[3], b2 ? r[8] : r[4], b2 ? r[9] : r[5]};
    const bool b1 = (int32_t)__builtin_amdgcn_ubfe(uu[2 + D], t.tsh, 16u) < thr;
    const uint32_t vv[4] = {b1 ? uu[2] : uu[0], b1 ? uu[3] : uu[1], b1 ? uu[4] : uu[2], b1 ? uu[5] : uu[3]};
    const bool b0 = (int32_t)__builtin_amdgcn_ubfe(vv[1 + D], t.tsh, 16u) < thr;
    const bool b3 = (int32_t)__builtin_amdgcn_ubfe(r[8 + D], t.tsh, 16u) < thr;  // (implies the other three)
    uint32_t w0 = b3 ? r[8] : b0 ? vv[1] : vv[0], w1 = b3 ? r[9] : b0 ? vv[2] : vv[1], w2 = b3 ? r[10] : b0 ? vv[3] : vv[2];
    uint32_t k = lb + (b3 ? 8u : (b2 ? 4u : 0u) + (b1 ? 2u : 0u) + (b0 ? 1u : 0u));
    if (wide) return false;  // (tested here so that the window is requested without waiting for the flag)
    if (k >= cnt) {
      // NEXT: the threshold lies beyond the tile (inconsistent record): literal walk.  Else every op of the tile starts
      // at or before the threshold, so the tile's last op is the one looked for
      if (NEXT) return false;
      k = cnt - 1u;
      w0 = line[PFX_E0 + k - 1u]; w1 = line[PFX_E0 + k]; w2 = line[PFX_E0 + k + 1u];
    }
    // window: NEXT: e_k, e_(k+1), e_(k+2); else e_(k-1), e_k, e_(k+1)
    which.j = j; which.k = k;
    if (pfx_eval(c, t, NEXT ? w0 : w1, NEXT ? w1 : w2, first_form, oq, ot, which.adj)) return true;
    if (NEXT) {
      which.k = k + 1u;
      if (k + 1u < cnt) return pfx_eval(c, t, w1, w2, first_form, oq, ot, which.adj);
      if (j + 1u >= c.m) return false;
      j += 1u;
    } else {
      which.k = k - 1u;
      if (k > 0u) return pfx_eval(c, t, w0, w1, first_form, oq, ot, which.adj);
      if (j == 0u) return false;
      j -= 1u;
    }
  }
}

// IDENT: also evaluate calculate_gap_compressed_identity on the projected CIGAR
// slice (impg.rs:1283-1287, :2952-2973) and drop hits below min_identity.
// MODE & MODE_CIGAR: also record which ops form the projected CIGAR slice
// (store_cigar); such launches walk tiles A..B literally -- BEDPE/PAF output is
// not the throughput path.
#ifndef IMPG_PROJ_BLOCK
#define IMPG_PROJ_BLOCK 256
#endif
// -DIMPG_PHASE_CLOCKS (experiments): s_memtime at the projection kernel's dependency boundaries, summed over the waves of
// every 64th block; read (and cleared) by impg_gpu_debug_phase_clocks -- scripts/phase_clocks.py.  Each mark waits for
// everything outstanding first, so the phases are what a wave WAITS for, not what it issues.
#ifdef IMPG_PHASE_CLOCKS
__device__ unsigned long long g_phase_clk[16];
#define PHASE_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); phase_t[i] = __builtin_readcyclecounter(); } while (0)
#define PHASE_MARK_NW(i) do { phase_t[i] = __builtin_readcyclecounter(); } while (0)
#define PHASE_ARG , unsigned long long *phase_t
#define PHASE_PASS , phase_t
#else
#define PHASE_MARK(i) do { } while (0)
#define PHASE_MARK_NW(i) do { } while (0)
#define PHASE_ARG
#define PHASE_PASS
#endif
#ifdef IMPG_PROJECT_WAVES  // (experiments: force the register allocation that gives this many waves per SIMD)
#define PROJECT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(IMPG_PROJECT_WAVES, IMPG_PROJECT_WAVES)))
#else
#define PROJECT_OCCUPANCY
#endif
constexpr uint32_t PROJ_BLOCK = IMPG_PROJ_BLOCK, PROJ_WAVES = PROJ_BLOCK / 64u;
// LDS copies of entries and prefix lines (project_staged_kernel) are padded so that lanes reading the same word of
// different entries / lines do not meet on one bank: the strides are 4 banks (16 bytes) off a multiple of the 32.
constexpr uint32_t STG_ENT_STRIDE = 20u;                                   // words per staged entry (16 + 4)
constexpr uint32_t STG_LINE_STRIDE = TILE_WORDS + 4u;                      // words per staged prefix line
constexpr uint32_t STG_REC_STRIDE = INLINE_TILES * STG_LINE_STRIDE + 4u;   // words per staged record (8 lines)
// project_entries_kernel (IMPG_ENT_CP_LDS, round 5): the entry's words 4 .. 15 -- the record totals and the seven inline
// checkpoints -- also sit behind the wave's LDS record, and a chunk's lanes read them from there (two broadcast reads)
// instead of comparing against scalar registers: held in scalars they were the registers the allocator spilled (the
// kernel runs at its 102-SGPR limit), and every chunk fetched them back with fourteen v_readlane.
#ifndef IMPG_ENT_CP_LDS
#define IMPG_ENT_CP_LDS 1
#endif
constexpr uint32_t ENT_CP_OFF = INLINE_TILES * STG_LINE_STRIDE;           // where the entry's checkpoint words sit in the wave's LDS record
// One (range, entry) pair: project_overlapping_interval's PAF branch (impg.rs:1260-1312) for the range [f_start, f_end)
// against entry eidx.  ok: the projection exists (and passes the identity filter); qid / res: its query sequence and
// {q_first, q_last, t_first, t_last}; slice descriptors go to sl[p] under MODE_CIGAR.  Shared by project_kernel (a
// lane per pair of a level) and the per-query walk kernel (walk_device.inc).
// STAGED (project_staged_kernel): the entry and the record's prefix lines are read from the block's LDS copy
// (st_entry = the entry's four vectors, st_pfx = the record's first prefix line) instead of from the index.
// ORIENT (project_entries_kernel: a wave works on ONE entry, whose words sit in scalar registers): the entry's
// orientation as a compile-time constant -- 0 = a forward entry, 1 = a reversed entry walked front to back, 2 = a
// reversed entry of a reverse-strand record, walked back to front; -1 = read from the entry's flags.  With it every
// `swp ? a : b` / `flip ? a : b` below folds away.
template <bool TRANSITIVE, int MODE, bool STAGED, int ORIENT>
__device__ __forceinline__ void project_core(const DeviceIndexView &v, uint4 e0, uint4 e1, uint4 e2, uint4 e3, int32_t f_start, int32_t f_end,
                                             uint32_t p, double min_identity, uint32_t *__restrict__ err_flag, const SliceArrays &sl,
                                             unsigned long long *__restrict__ accepted, bool &ok, uint32_t &qid, TileScan &res PHASE_ARG,
                                             const uint32_t *st_pfx);
template <bool TRANSITIVE, int MODE, bool STAGED = false>
__device__ __forceinline__ void project_pair(const DeviceIndexView &v, uint32_t eidx, int32_t f_start, int32_t f_end, uint32_t p,
                                             double min_identity, uint32_t *__restrict__ err_flag, const SliceArrays &sl,
                                             unsigned long long *__restrict__ accepted, bool &ok, uint32_t &qid, TileScan &res PHASE_ARG,
                                             const uint4 *st_entry = nullptr, const uint32_t *st_pfx = nullptr) {
  // Two round trips, not four ahead of the tiles: the indices (and the frontier record) above, then the
  // 64-byte entry (coordinates, record totals, inline checkpoints).  The empty asm statement pins the four
  // reads together -- left alone, the compiler sinks part of them below the entry's "has ops" test.
  const uint4 *ep = STAGED ? st_entry : reinterpret_cast<const uint4 *>(v.entries + eidx);
  uint4 e0 = ep[0], e1 = ep[1], e2 = ep[2], e3 = ep[3];
  asm volatile("" : "+v"(e0.x), "+v"(e1.z), "+v"(e2.x), "+v"(e3.x));
  PHASE_MARK(3);
  project_core<TRANSITIVE, MODE, STAGED, -1>(v, e0, e1, e2, e3, f_start, f_end, p, min_identity, err_flag, sl, accepted, ok, qid, res PHASE_PASS, st_pfx);
}
template <bool TRANSITIVE, int MODE, bool STAGED, int ORIENT>
__device__ __forceinline__ void project_core(const DeviceIndexView &v, uint4 e0, uint4 e1, uint4 e2, uint4 e3, int32_t f_start, int32_t f_end,
                                             uint32_t p, double min_identity, uint32_t *__restrict__ err_flag, const SliceArrays &sl,
                                             unsigned long long *__restrict__ accepted, bool &ok, uint32_t &qid, TileScan &res PHASE_ARG,
                                             const uint32_t *st_pfx) {
  constexpr bool IDENT = (MODE & MODE_IDENT) != 0;
  constexpr bool CIGAR = (MODE & MODE_CIGAR) != 0;
  static_assert(!STAGED || MODE == 0 || MODE == MODE_IDENT, "on staged lines: the plain projection, or the one under the identity filter (its identity lines are read from the index)");
  (void)accepted;
  {
    FrontierRec f;
    f.start = f_start; f.end = f_end;
    const int32_t en_ts = (int32_t)e0.x, en_te = (int32_t)e0.y, en_qs = (int32_t)e0.z, en_qe = (int32_t)e0.w;
    const uint32_t nops_flags = e1.z;
    const uint32_t n = nops_flags & OP_LEN_MASK;
    const bool rev = (nops_flags & EF_STRAND) != 0;
    PairCtx c;
    c.swp = ORIENT < 0 ? (nops_flags & EF_REVERSED) != 0 : ORIENT >= 1;
    c.flip = ORIENT < 0 ? c.swp && rev : ORIENT == 2;
    c.zt = c.swp ? 3u : 2u;  // 'I' consumes no target; for a reversed entry 'D' does (impg.rs:146-151)
    c.zq = c.swp ? 2u : 3u;
    c.ts = en_ts;
    const int32_t qbase = rev ? en_qe : en_qs;  // impg.rs:2778-2782
    c.R0 = f.start;
    c.R1 = f.end;
    if (TRANSITIVE) {  // project the clipped overlap (impg.rs:2398-2400)
      c.R0 = max(c.R0, en_ts);
      c.R1 = min(c.R1, en_te);
    }
    c.last_tp = min(en_te, c.R1);  // impg.rs:2798
    c.m = (n + TILE_OPS - 1) / TILE_OPS;
    c.totT = e1.w;
    c.totQ = e2.x;
    if (IMPG_ENT_CP_LDS && STAGED && ORIENT >= 0) {
      // (record totals off the wave's LDS record too: as scalars they were two more of the spilled registers, fetched back
      // ten times a chunk)
      c.totT = st_pfx[ENT_CP_OFF + 3u];
      c.totQ = st_pfx[ENT_CP_OFF + 4u];
    }
    c.ops = v.ops + (size_t)e1.y * TILE_WORDS;
    if (n == 0) {
      atomicOr(err_flag, 1u);  // record without cg:Z (the reference panics, impg.rs:506-511)
    } else {
      // Shortcuts that need no CIGAR bytes (exact):
      //  * the range reaches back to the alignment start: op 0 is the first
      //    overlapping op whatever its type, at (query offset 0, target ts);
      //  * the range reaches the alignment end and the CIGAR is consistent with
      //    the PAF coordinates: the final op is the last overlapping op, ending at
      //    (query offset totQ, target te).
      // (Under the identity filter the slice then starts at the record's first op / ends at its last one with nothing
      // taken off them: the identity lines give those sums without locating anything.  The two-walk and store_cigar
      // forms do not take the shortcuts.)
      constexpr bool ON_LINES = !(MODE & (MODE_WALK | MODE_CIGAR));
      const bool start_cov = ON_LINES && c.R0 <= c.ts && c.last_tp > c.ts;
      const bool end_cov = ON_LINES && c.R1 >= en_te && c.R0 < en_te && (int32_t)c.totT == en_te - c.ts;
      // Effective tile k (k-th tile in this entry's walking order) starts at target
      // prefix P[k]: P[0] = 0, P[m] = totT, P[1..m-1] inline (m <= 8) or external.
      //   A = first k with P[k+1] >= R0 - ts        (holds the first op that can overlap)
      //   B = last  k with P[k]   <= last_tp - ts   (holds the last live op)
      const int32_t xa = c.R0 - c.ts, xb = c.last_tp - c.ts;
      // (the plain projection wants the last tile that STARTS at or before last_target_pos: it counts P[i] <= xb)
      const int32_t xbc = ON_LINES ? xb + 1 : xb;
      uint32_t cA = 0, cB = 0;  // cA = #{i in [1,m] : P[i] < xa}, cB = #{i in [0,m) : P[i] < xbc}
      if (ORIENT >= 0 || c.m <= INLINE_TILES) {  // (a known orientation is only handed in for a record of at most INLINE_TILES tiles)
        // the seven inline slots hold P[1..m-1], P[m] = totT, then INT_MAX (index_build.cpp); P[8] is totT when m = 8
        uint4 p2 = e2, p3 = e3;
        if (IMPG_ENT_CP_LDS && STAGED && ORIENT >= 0) {
          const uint4 *cp = reinterpret_cast<const uint4 *>(__builtin_assume_aligned(st_pfx, 16)) + ENT_CP_OFF / 4u;
          p2 = cp[1];
          p3 = cp[2];
          asm volatile("" : "+v"(p2.x), "+v"(p2.y), "+v"(p2.z), "+v"(p2.w), "+v"(p3.x), "+v"(p3.y), "+v"(p3.z), "+v"(p3.w));  // (two 16-byte reads off the record's address)
        }
        const int32_t P[8] = {(int32_t)p2.y, (int32_t)p2.z, (int32_t)p2.w, (int32_t)p3.x, (int32_t)p3.y, (int32_t)p3.z, (int32_t)p3.w,
                              c.m == INLINE_TILES ? (int32_t)c.totT : 0x7FFFFFFF};
        cB = 0 < xbc ? 1u : 0u;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
          cA += P[i] < xa ? 1u : 0u;
          cB += P[i] < xbc ? 1u : 0u;
        }
        cB = min(cB, c.m);  // (P[m] itself is not a tile start: only an inconsistent CIGAR has totT < xb)
      } else {
        const uint32_t *P = v.ext_cp + e2.y;  // P[0..m]
        // two lower-bound searches side by side, 4-ary (three probes each per round, all in flight together:
        // half the dependent rounds of a binary search): first i in [1,m] with P[i] >= xa, first i in [0,m)
        // with P[i] >= xb
        uint32_t la = 1, ha = c.m + 1, lb = 0, hb = c.m;
        while (la < ha || lb < hb) {
          const uint32_t wa = ha - la, wb = hb - lb;
          const uint32_t a1 = la + (wa >> 2), a2 = la + (wa >> 1), a3 = la + (wa >> 1) + (wa >> 2);
          const uint32_t b1 = lb + (wb >> 2), b2 = lb + (wb >> 1), b3 = lb + (wb >> 1) + (wb >> 2);
          const bool ga = wa != 0, gb = wb != 0;
          const int32_t pa1 = ga ? (int32_t)P[a1] : 0, pa2 = ga ? (int32_t)P[a2] : 0, pa3 = ga ? (int32_t)P[a3] : 0;
          const int32_t pb1 = gb ? (int32_t)P[b1] : 0, pb2 = gb ? (int32_t)P[b2] : 0, pb3 = gb ? (int32_t)P[b3] : 0;
          if (ga) {
            if (pa1 >= xa) ha = a1; else if (pa2 >= xa) { la = a1 + 1; ha = a2; } else if (pa3 >= xa) { la = a2 + 1; ha = a3; } else la = a3 + 1;
          }
          if (gb) {
            if (pb1 >= xbc) hb = b1; else if (pb2 >= xbc) { lb = b1 + 1; hb = b2; } else if (pb3 >= xbc) { lb = b2 + 1; hb = b3; } else lb = b3 + 1;
          }
        }
        cA = la - 1;
        cB = lb;
      }
      if (cA < c.m) {
        // Two short literal walks instead of one long one (exact):
        //  A. from the first sub-tile whose END sum reaches R0 (every op before it ends
        //     before R0 and cannot overlap) until the first overlapping op shows up --
        //     normally in that very sub-tile -- or the ops start past last_target_pos;
        //  B. from the last sub-tile that starts BEFORE last_target_pos until the ops
        //     start past it; it records the last overlapping op.  If that sub-tile is
        //     not beyond the one walk A stopped in, walk A simply carries on.
        // Everything between the two walks lies between the first and the last
        // overlapping op and cannot change the answer.
        const uint32_t kA = cA, kB = max(cB, 1u) - 1u;  // effective tile indices
        TileScan sa;  // one record for both walks: the first overlapping op is set once, the last keeps moving
        sa.found = sa.any = false;
        sa.pqs = sa.pts = sa.pqe = sa.pte = -1;
        IdentScan ia, ib;
        ident_reset(ia);
        ident_reset(ib);
        bool walked = false, joined = false, need_walk = false;
        constexpr int PA = MODE == 0 ? PART_FIRST : PART_BOTH, PB = MODE == 0 ? PART_LAST : PART_BOTH;
        uint32_t ipa = 0, ipb = 0;  // idp[] rows the two walks count from
        int32_t lineM = 0, lineX = 0, lineG = 0;  // the slice's counts off the identity lines (sums of one record: below 2^31)
        if (CIGAR) {
          res = walk_tiles<MODE>(c, orig_tile(c, kA), c.flip ? 0u : c.m - 1u, ia);
          walked = true;
        } else if (!(MODE & MODE_WALK)) {
          // the two ends on the prefix lines (see pfx_end); in storage order a back-to-front entry swaps their roles
          const uint32_t *pfx_rec = STAGED ? st_pfx : v.pfx + (size_t)e1.y * TILE_WORDS;
          bool lit = cB == 0u;  // no tile starts at or before last_target_pos: let the literal walk say so
          const uint32_t kL = cB ? cB - 1u : 0u;  // effective tile of the last op that starts at or before it
          int32_t fq = 0, ft = c.ts, lq = (int32_t)c.totQ, lt = en_te;  // the shortcuts' answers
          const bool do_first = !start_cov, do_last = !end_cov;
          // call 1 looks forward in storage order (k, k + 1): the first end of a front-to-back walk, the last end of a
          // back-to-front one; call 2 looks backward (k, k - 1)
          // (storage order: call 1 finds the lower end, call 2 the upper; an end the shortcuts answer is the record's
          // first / last op)
          PfxOp lo_op{0u, 0u, 0}, hi_op{c.m - 1u, n - (c.m - 1u) * TILE_OPS - 1u, 0};
          const bool lo_edge = c.flip ? !do_last : !do_first;
          // both ends' line headers in one round trip, ahead of the searches that depend on them
          const bool run1 = !lit && (c.flip ? do_last : do_first), run2 = !lit && (c.flip ? do_first : do_last);
          const uint32_t j1 = orig_tile(c, c.flip ? kL : kA), j2 = orig_tile(c, c.flip ? kA : kL);
          uint4 hd1 = make_uint4(0u, 0u, 0u, 0u), hd2 = make_uint4(0u, 0u, 0u, 0u);
          constexpr uint32_t LSTRIDE = STAGED ? STG_LINE_STRIDE : TILE_WORDS;
          if (run1) hd1 = *reinterpret_cast<const uint4 *>(pfx_rec + (size_t)j1 * LSTRIDE);
          if (run2) hd2 = *reinterpret_cast<const uint4 *>(pfx_rec + (size_t)j2 * LSTRIDE);
          asm volatile("" : "+v"(hd1.x), "+v"(hd1.y), "+v"(hd1.z), "+v"(hd1.w), "+v"(hd2.x), "+v"(hd2.y), "+v"(hd2.z), "+v"(hd2.w));
          PHASE_MARK(4);
          if (run1) {
            int32_t oq, ot;
            const bool okc = pfx_end<true, LSTRIDE>(c, pfx_rec, n, j1, c.flip ? (int32_t)c.totT - xb : xa, !c.flip, oq, ot, lo_op, hd1);
            lit = !okc;
            if (c.flip) { lq = oq; lt = ot; } else { fq = oq; ft = ot; }
          }
          PHASE_MARK(5);
          if (run2 && !lit) {
            int32_t oq, ot;
            const bool okc = pfx_end<false, LSTRIDE>(c, pfx_rec, n, j2, (c.flip ? (int32_t)c.totT - xa : xb) + 1, c.flip, oq, ot, hi_op, hd2);
            lit = !okc;
            if (c.flip) { fq = oq; ft = ot; } else { lq = oq; lt = ot; }
          }
          PHASE_MARK(6);
          if (IDENT && !lit && (c.flip ? !do_first : !do_last)) {
            // The upper end came from a shortcut, so no search looked at the record's last tile: if that tile is `wide`
            // its identity line's 16-bit fields do not hold its sums (a 70000= op) -- literal walk, as pfx_end would say.
            lit = (pfx_rec[(size_t)(c.m - 1u) * LSTRIDE] >> 31) != 0u;
          }
          if (IDENT && !lit) {
            // The slice [first op, last op] in storage order is [lo_op, hi_op]: its matched / mismatched bases and gap ops
            // are the identity lines' sums after hi_op minus those before lo_op (IDL_*, impg_internal.hpp); an op is a
            // match / mismatch op iff its own entry difference says so, which is what the two adjustments need to know.
            const uint32_t *idl = reinterpret_cast<const uint32_t *>(v.idp) + (size_t)e1.y * IDL_WORDS;
            const uint32_t *la = idl + (size_t)lo_op.j * IDL_WORDS, *lb = idl + (size_t)hi_op.j * IDL_WORDS;
            uint4 ha = make_uint4(0u, 0u, 0u, 0u), hb = *reinterpret_cast<const uint4 *>(lb);
            u32x2_u ea = {0u, 0u}, eb = *reinterpret_cast<const u32x2_u *>(lb + IDL_E0 + hi_op.k);
            if (!lo_edge) {  // (before the record's first op every sum is zero)
              ha = *reinterpret_cast<const uint4 *>(la);
              ea = *reinterpret_cast<const u32x2_u *>(la + IDL_E0 + lo_op.k);
            }
            asm volatile("" : "+v"(ha.x), "+v"(hb.x), "+v"(ea), "+v"(eb));
            const int32_t am0 = (int32_t)(ea.x & 0xFFFFu), ax0 = (int32_t)(ea.x >> 16), am1 = (int32_t)(ea.y & 0xFFFFu), ax1 = (int32_t)(ea.y >> 16);
            const int32_t bm0 = (int32_t)(eb.x & 0xFFFFu), bx0 = (int32_t)(eb.x >> 16), bm1 = (int32_t)(eb.y & 0xFFFFu), bx1 = (int32_t)(eb.y >> 16);
            lineM = (int32_t)(hb.x - ha.x) + (bm1 - am0);
            lineX = (int32_t)(hb.y - ha.y) + (bx1 - ax0);
            lineG = (int32_t)(hb.z - ha.z) + (__popc(hb.w & ((2u << hi_op.k) - 1u)) - __popc(ha.w & ((1u << lo_op.k) - 1u)));
            // first_op_offset comes off the FIRST op in walking order, last_op_remaining (<= 0) goes onto the LAST
            const int32_t lo_adj = c.flip ? lo_op.adj : -lo_op.adj, hi_adj = c.flip ? -hi_op.adj : hi_op.adj;
            lineM += (am1 != am0 ? lo_adj : 0) + (bm1 != bm0 ? hi_adj : 0);
            lineX += (ax1 != ax0 ? lo_adj : 0) + (bx1 != bx0 ? hi_adj : 0);
          }
          if (lit) {
            res = walk_tiles<MODE>(c, orig_tile(c, kA), c.flip ? 0u : c.m - 1u, ia);
            walked = true;
          } else {
            res.found = true;
            res.pqs = fq; res.pts = ft; res.pqe = lq; res.pte = lt;
          }
        } else {
        Cursor cur;
        cur.k = 0xFFFFFFFFu; cur.he = 0; cur.T = 0; cur.Qn = 0;
        if (start_cov) {
          sa.found = true;
          sa.pqs = 0;
          sa.pts = c.ts;
        } else {
          const TileHdr hd = tile_header(c, orig_tile(c, kA));
          const uint32_t nsub = tile_subs(n, orig_tile(c, kA));
          uint32_t he;
          if (!c.flip) {
            const int32_t y = xa - (int32_t)hd.t0;
            he = ((int32_t)hd.bt1 < y ? 1u : 0u) + ((int32_t)hd.bt2 < y ? 1u : 0u) + ((int32_t)hd.bt3 < y ? 1u : 0u);
            he = min(he, nsub - 1u);
          } else {
            const int32_t z = (int32_t)(c.totT - hd.t0) - xa;
            he = ((int32_t)hd.bt3 > z ? 1u : 0u) + ((int32_t)hd.bt2 > z ? 1u : 0u) + ((int32_t)hd.bt1 > z ? 1u : 0u);
            he = max(he, TILE_SUBS - nsub);  // the sub-tiles without ops come first in a back-to-front walk
          }
          place(c, n, hd, kA, he, cur);
          const uint32_t ipa0 = TILE_SUBS * (e1.y + orig_tile(c, cur.k)) + orig_sub(c, cur.he);
          for (;;) {
            scan_cur<MODE, PA>(c, cur, sa, ia);
            if (sa.found || cur.T > c.last_tp) break;
            if (!advance(c, n, cur)) break;
          }
          ipa = c.flip ? TILE_SUBS * (e1.y + orig_tile(c, cur.k)) + orig_sub(c, cur.he) : ipa0;
        }
        if (sa.found && end_cov) {
          sa.pqe = (int32_t)c.totQ;
          sa.pte = en_te;
        } else if (sa.found) {
          // requested only now: asked for earlier, the line was evicted before use; when walk A
          // read the same tile it is an L1 hit, and not keeping A's header alive saves registers
          const TileHdr hd = tile_header(c, orig_tile(c, kB));
          const uint32_t nsub = tile_subs(n, orig_tile(c, kB));
          uint32_t he;
          if (!c.flip) {
            const int32_t y = xb - (int32_t)hd.t0;
            he = ((int32_t)hd.bt1 < y ? 1u : 0u) + ((int32_t)hd.bt2 < y ? 1u : 0u) + ((int32_t)hd.bt3 < y ? 1u : 0u);
            he = min(he, nsub - 1u);
          } else {
            const int32_t z = (int32_t)(c.totT - hd.t0) - xb;
            he = ((int32_t)hd.bt3 > z ? 1u : 0u) + ((int32_t)hd.bt2 > z ? 1u : 0u) + ((int32_t)hd.bt1 > z ? 1u : 0u);
            he = max(he, TILE_SUBS - nsub);
          }
          if (hd.wide) he = c.flip ? TILE_SUBS - nsub : 0u;
          joined = !start_cov && (kB < cur.k || (kB == cur.k && he <= cur.he));
          bool go;
          if (joined && MODE == 0) {
            // walk A stands in or past that sub-tile but only looked for the first op: walk B starts over
            // at the sub-tile walk A stopped in (whose scan it repeats, now keeping the last op)
            const uint32_t ka = cur.k, ha = cur.he;
            if (ka == kB) place(c, n, hd, ka, ha, cur);
            else place(c, n, tile_header(c, orig_tile(c, ka)), ka, ha, cur);
            go = true;
            sa.any = false;
          } else if (joined) {  // (identity filter) walk A carries on and keeps recording
            ib = ia;
            go = cur.T <= c.last_tp && advance(c, n, cur);
          } else {
            place(c, n, hd, kB, he, cur);
            ipb = TILE_SUBS * (e1.y + orig_tile(c, cur.k)) + orig_sub(c, cur.he);
            go = true;
            sa.any = false;
          }
          while (go) {
            scan_cur<MODE, PB>(c, cur, sa, ib);
            go = cur.T <= c.last_tp && advance(c, n, cur);
          }
          if (c.flip) ipb = TILE_SUBS * (e1.y + orig_tile(c, cur.k)) + orig_sub(c, cur.he);
          need_walk = (MODE == 0 || !joined) && !sa.any;  // walk B saw no overlapping op (cannot happen for a consistent CIGAR; stay exact)
        }
        if (need_walk) {
          res = walk_tiles<MODE>(c, orig_tile(c, kA), c.flip ? 0u : c.m - 1u, ia);
          walked = true;
        } else {
          res = sa;
        }
        }
#ifdef IMPG_DEBUG_WALK
        if (walked) atomicAdd(&accepted[(blockIdx.x % COUNT_SLOTS) * COUNT_STRIDE], 1ull << 40);
#endif
        if (CIGAR) {
          sl.a[p] = ia.first_oi;
          sl.n[p] = (ia.first_oi > ia.last_oi ? ia.first_oi - ia.last_oi : ia.last_oi - ia.first_oi) + 1u;
          sl.off[p] = ia.first_off;
          sl.rem[p] = ia.last_rem;
        }
        if (IDENT && res.found) {
          // op counts of the slice [first op, last op] in ORIGINAL op order:
          //   forward walks : prefix(B's first sub-tile) + lastB  -  (prefix(A's first sub-tile) + firstA)
          //   backward walks: (prefix(A's last sub-tile) + total(A walk) - firstA) - (prefix(B's last sub-tile) + total(B walk) - lastB)
          // where prefix(s) = matched / mismatched bases and gap ops before sub-tile s (idp[])
          // and firstA / lastB are the walking-order running sums snapshotted by op_step.
          int64_t M, X, G;
          if (!walked && !(MODE & MODE_WALK)) {  // both ends on the prefix lines
            M = lineM; X = lineX; G = lineG;
          } else if (walked || joined) {  // one continuous walk: plain difference of its running sums
            const IdentScan &w = walked ? ia : ib;
            M = (int64_t)w.lm - w.fm; X = (int64_t)w.lx - w.fx; G = (int64_t)w.lg - w.fg;
            M += -(int64_t)w.first_adj_m + w.last_adj_m;
            X += -(int64_t)w.first_adj_x + w.last_adj_x;
          } else {
            const uint4 pa = v.idp[ipa], pb = v.idp[ipb];
            if (!c.flip) {
              M = ((int64_t)pb.x + ib.lm) - ((int64_t)pa.x + ia.fm);
              X = ((int64_t)pb.y + ib.lx) - ((int64_t)pa.y + ia.fx);
              G = ((int64_t)pb.z + ib.lg) - ((int64_t)pa.z + ia.fg);
            } else {
              M = ((int64_t)pa.x + ia.rm - ia.fm) - ((int64_t)pb.x + ib.rm - ib.lm);
              X = ((int64_t)pa.y + ia.rx - ia.fx) - ((int64_t)pb.y + ib.rx - ib.lx);
              G = ((int64_t)pa.z + ia.rg - ia.fg) - ((int64_t)pb.z + ib.rg - ib.lg);
            }
            M += -(int64_t)ia.first_adj_m + ib.last_adj_m;
            X += -(int64_t)ia.first_adj_x + ib.last_adj_x;
          }
          // impg.rs:2967, :1284-1286: drop the hit iff fl(M / total) < min_identity (0 for an empty slice).  The rounded
          // quotient is within 2^-53 of M / total and the rounded product below within 2^-53 of min_identity * total, so
          // a comparison with 2^-50 of slack on either side decides all but borderline pairs without the division (a
          // double-precision quotient is ~40 instructions; every lane of every wave paid them).
          const int32_t total = (int32_t)(M + X + G);
          const double dm = (double)(int32_t)M, dt = (double)total, prod = min_identity * dt;
          bool drop;
          if (dm < prod * (1.0 - 0x1p-50)) drop = true;
          else if (dm > prod * (1.0 + 0x1p-50)) drop = false;
          else drop = (total == 0 ? 0.0 : dm / dt) < min_identity;
          if (drop) res.found = false;
        }
      }
      ok = res.found && res.pqs != res.pqe && res.pts != res.pte;  // impg.rs:2874-2877
      if (ok) qid = e1.x;
      // back from direction-normalised offsets to query coordinates
      res.pqs = qbase + (rev ? -res.pqs : res.pqs);
      res.pqe = qbase + (rev ? -res.pqe : res.pqe);
    }
  }
}

// position of the k-th set bit (k from 0) of the 64-bit mask hi:lo; k < popcount
__device__ __forceinline__ uint32_t select_bit64(uint32_t lo, uint32_t hi, uint32_t k) {
  uint32_t c = (uint32_t)__popc(lo), w = lo, base = 0;
  if (k >= c) { k -= c; w = hi; base = 32u; }
  c = (uint32_t)__popc(w & 0xFFFFu); if (k >= c) { k -= c; w >>= 16; base += 16u; }
  c = (uint32_t)__popc(w & 0xFFu);   if (k >= c) { k -= c; w >>= 8; base += 8u; }
  c = (uint32_t)__popc(w & 0xFu);    if (k >= c) { k -= c; w >>= 4; base += 4u; }
  c = (uint32_t)__popc(w & 0x3u);    if (k >= c) { k -= c; w >>= 2; base += 2u; }
  return base + (k >= (w & 1u) ? 1u : 0u);
}
// a finished row at its final place (OrderedOut): the hit, or a hole (query_id = 0xFFFFFFFF) where the projection returned
// None or the row is shorter than min_output_length
__device__ __forceinline__ void put_ordered_row(const OrderedOut &o, uint32_t d, uint32_t qid, uint32_t tid, bool ok, int32_t qs, int32_t qe,
                                                int32_t ts, int32_t te) {
  const bool on = ok && !(o.min_output_length >= 0 && abs(qe - qs) < o.min_output_length);
#ifdef IMPG_ORD_NOSTORE  // (experiment: the kernel without its row stores)
  if (!(qs == -0x7FFFFFF0 && te == 0x7FFFFFF1)) return;
#endif
  // (24 bytes at an 8-byte boundary: a 16-byte and an 8-byte store.  A row is a line of its own among 10^9 and every store
  // instruction of a wave touches 64 lines -- ~12 ms of the headline's final level per such instruction, measured; rows of
  // one aligned 32-byte sector cost the same: it is the scattered store, not a read-modify-write)
  uint32_t *p = reinterpret_cast<uint32_t *>(o.rows) + 6ull * d;
  const uint4 a = make_uint4(on ? qid : HIT_NONE, on ? (uint32_t)qs : 0u, on ? (uint32_t)qe : 0u, tid);
  const uint2 b = make_uint2(on ? (uint32_t)ts : 0u, on ? (uint32_t)te : 0u);
  __builtin_memcpy(p, &a, 16);
  __builtin_memcpy(p + 4, &b, 8);
}
// One place of the projection order -> its pair.  live: the place holds a pair; p: the pair's slot.
struct PairIn {
  uint32_t eidx, p;
  int32_t f_start, f_end;
  uint32_t live;  // (a word, not a bool: copies of the struct then move no padding bytes -- the compiler took those through scratch)
};
// LDS a block needs for pair_of_place (WindowLists form) and regroup_by_entry
constexpr uint32_t PROJ_WL_WORDS = 65u + 3u;                                   // place offsets of 65 ranges (+ padding to 16 bytes)
constexpr uint32_t PROJ_RG_WORDS = PROJ_BLOCK + 4u * PROJ_BLOCK + PROJ_WAVES + (4u - PROJ_WAVES % 4u) % 4u;  // payload, histogram, wave sums
// The pair at place pp of tile `lblock` (a tile = PROJ_BLOCK consecutive places; every thread of the block calls this
// together: the WindowLists form goes through LDS and barriers).
__device__ __forceinline__ PairIn pair_of_place(uint32_t pp, uint32_t lblock, uint32_t n_pairs, const FrontierRec *__restrict__ fr,
                                                const uint32_t *__restrict__ pair_range, const uint32_t *__restrict__ pair_entry,
                                                const ProjList &pl, const WindowLists &wl, uint32_t *wl_off /* LDS, PROJ_WL_WORDS */) {
  PairIn x;
  x.live = pp < n_pairs;
  x.p = pp; x.eidx = 0xFFFFFFFFu; x.f_start = 0; x.f_end = 0;
  uint32_t r = 0;
  // (in projection order the pair's range and entry are listed next to its slot: three coalesced reads)
  if (wl.tile_first && lblock * PROJ_BLOCK >= n_pairs) {
    // (a tile past the list: the grid is rounded up)
  } else if (wl.tile_first) {
    // The pairs of this tile straight from the count pass's per-range records (WindowLists, kernels.hpp): which
    // range holds place pp (a search over the place offsets of the <= 64 ranges from the tile's first one on, in
    // LDS; more of them only where ranges without hits pile up), then the (pp - offset)-th set bit of its hit mask.
    const uint32_t pp_last = min(lblock * PROJ_BLOCK + PROJ_BLOCK, n_pairs) - 1u;
    uint32_t base_i = wl.tile_first[lblock], ri = 0, k = 0;
    bool found = !x.live;
    for (;;) {
      if (threadIdx.x < 65u) wl_off[threadIdx.x] = base_i + threadIdx.x < wl.n_fr ? wl.pair_off[base_i + threadIdx.x] : 0xFFFFFFFFu;
      __syncthreads();
      const uint32_t lim = wl_off[64];
      if (!found && pp < lim) {
        uint32_t j = 0;  // the last of the 64 with offset <= pp (ranges without hits share their successor's offset: the last one holds the place)
#pragma unroll
        for (uint32_t st = 32u; st > 0u; st >>= 1) j += wl_off[j + st] <= pp ? st : 0u;
        ri = base_i + j;
        k = pp - wl_off[j];
        found = true;
      }
      __syncthreads();
      if (lim > pp_last) break;
      base_i += 64u;
    }
    if (x.live) {
      const uint4 w = wl.win[ri];
      const FrontierRec sf = wl.se[ri];
      const int2 se = make_int2(sf.start, sf.end);
      x.f_start = se.x; x.f_end = se.y;
      if (w.y - (w.x & ~3u) > 64u) x.eidx = pair_entry[pp];  // a window wider than the mask: listed by the wave-per-range emit
      else x.eidx = w.x + select_bit64(w.z, w.w, k);
      if (wl.range_out) wl.range_out[pp] = wl.range_places ? ri : wl.perm[ri];
    }
  } else if (x.live) {
    if (pl.slot) { x.p = pl.slot[pp]; r = pl.range[pp]; x.eidx = pl.entry[pp]; }
    else { r = pair_range[pp]; x.eidx = pair_entry[pp]; }
    asm volatile("" : "+v"(r), "+v"(x.eidx));
    const FrontierRec f = fr[r];
    x.f_start = f.start; x.f_end = f.end;
  }
  return x;
}
// The block's 256 pairs, regrouped by entry before anything of the index is read.  In projection order
// neighbouring lanes are the hits of ONE range on different entries: every lane reads its own entry line and
// its own tile lines.  The ranges of a block are neighbours in the lookup order and hit largely the same
// entries, so sorted by entry a wave's lanes share a handful of lines.  A counting sort in LDS, one bin per
// thread; results are stored at the pair's own slot, so which lane projects which pair changes nothing downstream.
// (bin = entry mod the block size: equal entries share a bin and neighbours sit in neighbouring bins, which is all the
// grouping is for -- the block's entries span fewer than 256 places but for a rare wide block, where bins then
// mix two entries.  The round-3 form binned by `entry - the block's smallest entry`: a wave reduction, an LDS round
// and a barrier more.)
template <uint32_t NB>  // threads of the block = pairs regrouped together
__device__ __forceinline__ void regroup_by_entry(PairIn &x, uint32_t *scratch /* LDS, 5 * NB + NB / 64 words, 16-byte aligned */) {
  uint4 *rg_pay = reinterpret_cast<uint4 *>(scratch);
  uint32_t *rg_hist = scratch + 4u * NB, *rg_ws = rg_hist + NB;
  rg_hist[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t bin = x.eidx & (NB - 1u);  // (a place beyond the list carries entry ~0: the last bin)
  const uint32_t pos = atomicAdd(&rg_hist[bin], 1u);
  __syncthreads();
  {  // bin starts: exclusive scan of the counts over the block (one barrier for the waves' sums, one for the starts)
    const uint32_t cnt = rg_hist[threadIdx.x], inc = wave_incl_scan(cnt);
    if (lane_id() == 63u) rg_ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t k = 0; k + 1u < NB / 64u; k++) base += k < (threadIdx.x >> 6) ? rg_ws[k] : 0u;
    rg_hist[threadIdx.x] = base + inc - cnt;
  }
  __syncthreads();
  rg_pay[rg_hist[bin] + pos] = make_uint4(x.eidx, x.p, (uint32_t)x.f_start, (uint32_t)x.f_end);
  __syncthreads();
  const uint4 mine = rg_pay[threadIdx.x];
  x.eidx = mine.x; x.p = mine.y; x.f_start = (int32_t)mine.z; x.f_end = (int32_t)mine.w;
  x.live = x.eidx != 0xFFFFFFFFu;
}
// accepted-projection count: one atomic per BLOCK, spread over COUNT_SLOTS
// cache lines (a single hot word caps the whole chip at ~90 atomics/us)
template <uint32_t NW>  // waves of the block
__device__ __forceinline__ void count_accepted(uint32_t mine, unsigned long long *__restrict__ accepted, uint32_t *wcnt /* LDS, NW */) {
  uint32_t w = mine;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
  if (lane_id() == 0) wcnt[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
#pragma unroll
    for (uint32_t k = 0; k < NW; k++) tot += wcnt[k];
    if (tot) atomicAdd(&accepted[(blockIdx.x % COUNT_SLOTS) * COUNT_STRIDE], (unsigned long long)tot);
  }
}
// (Round 4, measured and dropped: one block working through 8 consecutive tiles as a two-stage pipeline -- the next
// tile's list entries and frontier records in flight under the current tile's projection.  s_memtime at the dependency
// boundaries (-DIMPG_PHASE_CLOCKS, scripts/phase_clocks.py) had shown a wave waiting 3 400 of its 17 900 cycles for those
// two reads; with them hidden the other phases stretched by the same amount -- 17 900 cycles per tile again, at 72
// VGPRs / 7 waves: 36.5 ms against 33.6.  The kernel is bound by a shared pipe (the vector-memory path, see DESIGN
// 5.2), not by the latency of any one read.)
template <bool TRANSITIVE, int MODE>
__global__ __launch_bounds__(PROJ_BLOCK) PROJECT_OCCUPANCY void project_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr,
                                                      const uint32_t *__restrict__ pair_range,
                                                      const uint32_t *__restrict__ pair_entry, uint32_t n_pairs,
                                                      HitArrays h, unsigned long long *__restrict__ accepted,
                                                      uint32_t *__restrict__ err_flag, double min_identity,
                                                      SliceArrays sl, ProjList pl, int xcd_map,
                                                      const uint32_t *__restrict__ n_pairs_dev, int regroup, WindowLists wl) {
  if (n_pairs_dev) n_pairs = *n_pairs_dev;  // small batches: the count stays on the device, the grid covers an upper bound
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give
  // every XCD one contiguous eighth of the (locality-ordered) pair list instead of
  // every eighth block of it.  The grid is a multiple of 8 blocks.
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t lblock = xcd_map ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t pp = lblock * PROJ_BLOCK + threadIdx.x;
#ifdef IMPG_PHASE_CLOCKS
  unsigned long long phase_t[10];
  for (int i = 0; i < 10; i++) phase_t[i] = 0;
  PHASE_MARK(0);
#endif
  bool ok = false;
  TileScan res;
  res.found = res.any = false;
  res.pqs = res.pts = res.pqe = res.pte = -1;
  uint32_t qid = HIT_NONE;
  __shared__ uint32_t wl_off[PROJ_WL_WORDS];
  __shared__ uint4 rg_scratch[PROJ_RG_WORDS / 4u];
  __shared__ uint32_t wcnt[PROJ_WAVES];
  PairIn x = pair_of_place(pp, lblock, n_pairs, fr, pair_range, pair_entry, pl, wl, wl_off);
  PHASE_MARK(1);
  if (regroup) regroup_by_entry<PROJ_BLOCK>(x, reinterpret_cast<uint32_t *>(rg_scratch));
  PHASE_MARK(2);
  if (x.live) {
    project_pair<TRANSITIVE, MODE>(v, x.eidx, x.f_start, x.f_end, x.p, min_identity, err_flag, sl, accepted, ok, qid, res PHASE_PASS);
    h.qid[x.p] = qid;
    if (ok) {
      h.c[x.p] = make_int4(res.pqs, res.pqe, res.pts, res.pte);
    }
  }
  PHASE_MARK(7);
  count_accepted<PROJ_WAVES>(ok ? 1u : 0u, accepted, wcnt);
#ifdef IMPG_PHASE_CLOCKS
  PHASE_MARK(8);
  if ((blockIdx.x & 63u) == 0u && lane_id() == 0u && phase_t[3] && phase_t[4] && phase_t[5] && phase_t[6]) {
    for (int i = 0; i < 8; i++) atomicAdd(&g_phase_clk[i], phase_t[i + 1] - phase_t[i]);
    atomicAdd(&g_phase_clk[15], 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------
// The plain projection with the block's entries and prefix lines STAGED IN LDS (round 4).
//
// Where a level is dense -- the final level of the headline batch lists 2 x 10^9 pairs on 2 x 10^6 entries: every entry
// is hit by a thousand ranges, and in the lookup order those ranges are neighbours -- the per-pair reads of
// project_kernel fetch the same 64-byte entries and 128-byte prefix lines over and over through the vector-memory
// path (192 bytes per pair in un-coalesced 16-byte reads; that path, not HBM and not the ALUs, is what bounds it).
// Here one block takes STG_TILES consecutive tiles of the projection order (2 048 places: ~80 ranges, whose windows
// overlap almost completely), copies the entries they span and those entries' prefix lines into LDS once -- coalesced
// 16-byte reads, ~25 bytes per pair -- and every pair then runs the same search (project_pair<.., STAGED>) on LDS: no
// regrouping, and lanes keep the place order, so the result stores of a wave are one contiguous piece.
// A block whose places span more than STG_ECAP entries (sparse levels, segment borders) runs its tiles the way
// project_kernel does; a pair whose record has more than INLINE_TILES prefix lines reads them from the index.
// launch_project picks this kernel when the level averages >= IMPG_STAGE_DENSITY pairs per index entry (listed levels: >= 128, see stage_density_listed).
// ---------------------------------------------------------------------------
#ifndef IMPG_STG_THREADS
#define IMPG_STG_THREADS 512
#endif
#ifndef IMPG_STG_RANGES
#define IMPG_STG_RANGES 256
#endif
#ifndef IMPG_STG_ECAP
#define IMPG_STG_ECAP 56
#endif
constexpr uint32_t STG_THREADS = IMPG_STG_THREADS, STG_WAVES = STG_THREADS / 64u;
constexpr uint32_t STG_RANGES = IMPG_STG_RANGES;                       // ranges (consecutive in the lookup order) per block
constexpr uint32_t STG_ECAP = IMPG_STG_ECAP;                           // entries staged per block
constexpr uint32_t STG_ENT_V4 = STG_ECAP * STG_ENT_STRIDE / 4u;       // entries: 4 vectors each (+ padding)
constexpr uint32_t STG_LINE_V4 = STG_ECAP * STG_REC_STRIDE / 4u;       // prefix lines: 8 lines of 8 vectors per entry (+ padding)
static_assert((STG_RANGES & (STG_RANGES - 1u)) == 0u && STG_RANGES < STG_THREADS && STG_THREADS % 64u == 0, "a thread per range, whole waves");
static_assert(STG_LINE_V4 * 4u >= 5u * STG_THREADS + STG_WAVES, "the unstaged path's scratch overlays the line buffer");
static_assert(STG_ECAP * 4u <= STG_THREADS, "one pass copies the entries");
// The places [P0, P1) of a block's ranges, a turn of STG_THREADS at a time, a lane per place: the place's range from the
// block's LDS offsets, its entry from the range's hit mask (MASKS) or from the emit pass's list (then the next turn's
// entry is requested a turn ahead), and the projection on the staged copies where the entry is one of the n_e staged
// from emin on, else on the index (regroup_all: nothing is staged and every turn's pairs are regrouped by entry first,
// as project_kernel does; the scratch overlays the line buffer).  Returns the thread's count of accepted projections.
// OUT: what a pair leaves behind -- OUT_SLOTS: the level's hit arrays (+ range_out); OUT_QS: {query id, the range's place} as one
// pair in HitArrays::qid (a kept fused level); OUT_ROWS: a finished row at its final place (OrderedOut)
constexpr int OUT_SLOTS = 0, OUT_QS = 1, OUT_ROWS = 2;
template <bool TRANSITIVE, bool MASKS, bool CAN_STAGE = true, uint32_t NR = STG_RANGES, uint32_t NT = STG_THREADS, int MODE = 0, int OUT = OUT_SLOTS>  // NR: ranges of a block (st_off holds NR + 1 offsets); NT: its threads
__device__ __forceinline__ uint32_t project_places(const DeviceIndexView &v, const uint32_t *__restrict__ pair_entry, const HitArrays &h,
                                                   unsigned long long *__restrict__ accepted, uint32_t *__restrict__ err_flag,
                                                   const WindowLists &wl, uint32_t r0, uint32_t P0, uint32_t P1, uint32_t emin, uint32_t n_e,
                                                   bool regroup_all, const uint32_t *st_off, const uint4 *st_win, const int2 *st_se,
                                                   const uint4 *st_ent, uint4 *st_line PHASE_ARG, double min_identity = 0.0,
                                                   const uint32_t *st_dest = nullptr, const uint32_t *st_tid = nullptr) {
  const SliceArrays no_sl{nullptr, nullptr, nullptr, nullptr};
  uint32_t n_ok = 0;
  constexpr bool ordered = OUT == OUT_ROWS;  // (then nothing is regrouped: a lane keeps its place's range, whose row and target it writes)
  constexpr bool qs = OUT == OUT_QS;         // slots as {query id, the range's place} pairs (see project_entries_kernel)
  if (ordered || qs) regroup_all = false;
  // the block's places, a turn of NT at a time (pair lists: the next turn's entry is requested a turn ahead)
  uint32_t e_next = 0xFFFFFFFFu;
  if (!MASKS && (unsigned long long)P0 + threadIdx.x < P1) e_next = pair_entry[P0 + threadIdx.x];
  // (ordered rows by visit: the place's byte -- which bit of its window's mask -- likewise a turn ahead)
  const bool by_visit = ordered && MASKS && wl.ord.by_visit != 0u;  // (the range's places run in visit order: the rows of a wave's lanes lie in a row)
  uint32_t v_next = 0u;
  if (by_visit && (unsigned long long)P0 + threadIdx.x < P1) v_next = wl.ord.vpos[P0 + threadIdx.x];
#pragma unroll 1
  for (uint32_t base = P0; base < P1; base += NT) {
    if (base + NT < base) break;  // (cannot happen: n_pairs stays 16 below 2^32 and P1 <= n_pairs)
    const uint32_t pp = base + threadIdx.x;
    PairIn y;
    y.live = pp < P1 ? 1u : 0u;
    y.p = pp; y.eidx = 0xFFFFFFFFu; y.f_start = 0; y.f_end = 0;
    // the place's range: the last one with offset <= pp (ranges without hits share their successor's offset)
    uint32_t j = 0;
#pragma unroll
    for (uint32_t st = NR / 2u; st > 0u; st >>= 1) j += st_off[j + st] <= pp ? st : 0u;
    uint32_t o_dest = 0, o_tid = 0;
    if (y.live) {
      const int2 se = st_se[j];
      y.f_start = se.x; y.f_end = se.y;
      if (MASKS) {
        const uint4 w = st_win[j];
        const bool listed = w.y - (w.x & ~3u) > 64u;
        if (listed) y.eidx = pair_entry[pp];  // a window wider than the mask: listed by the wave-per-range emit
        else if (by_visit) y.eidx = w.x + v_next;
        else y.eidx = w.x + select_bit64(w.z, w.w, pp - st_off[j]);
        if (ordered) {  // (a listed window's entries are in visit order already)
          o_dest = st_dest[j] + (listed || by_visit ? pp - st_off[j] : (uint32_t)wl.ord.vpos[pp]);
          o_tid = st_tid ? st_tid[j] : wl.se[r0 + j].target_id;
        } else if (qs) o_tid = r0 + j;
        else if (wl.range_out) wl.range_out[pp] = wl.range_places ? r0 + j : wl.perm[r0 + j];
      } else {
        y.eidx = e_next;
      }
    }
    if (!MASKS) {
      e_next = 0xFFFFFFFFu;
      if ((unsigned long long)pp + NT < P1) e_next = pair_entry[pp + NT];
    }
    if (by_visit) {
      v_next = 0u;
      if ((unsigned long long)pp + NT < P1) v_next = wl.ord.vpos[pp + NT];
    }
    if (regroup_all) regroup_by_entry<NT>(y, reinterpret_cast<uint32_t *>(st_line));
    if (y.live) {
      bool ok = false;
      TileScan res;
      res.found = res.any = false;
      res.pqs = res.pts = res.pqe = res.pte = -1;
      uint32_t qid = HIT_NONE;
      const uint32_t slot = y.eidx - emin;
      const uint4 *se = st_ent + min(slot, STG_ECAP - 1u) * (STG_ENT_STRIDE / 4u);
      if (CAN_STAGE && slot < n_e && (se[1].z & OP_LEN_MASK) <= INLINE_TILES * TILE_OPS)
        project_pair<TRANSITIVE, MODE, true>(v, y.eidx, y.f_start, y.f_end, y.p, min_identity, err_flag, no_sl, accepted, ok, qid, res PHASE_PASS, se,
                                             reinterpret_cast<const uint32_t *>(st_line + slot * (STG_REC_STRIDE / 4u)));
      else  // an entry beyond the staged span, or a record with more prefix lines than a staged record holds
        project_pair<TRANSITIVE, MODE>(v, y.eidx, y.f_start, y.f_end, y.p, min_identity, err_flag, no_sl, accepted, ok, qid, res PHASE_PASS);
      if (ordered) put_ordered_row(wl.ord, o_dest, qid, o_tid, ok, res.pqs, res.pqe, res.pts, res.pte);
      else {
        if (qs) reinterpret_cast<uint2 *>(h.qid)[y.p] = make_uint2(qid, o_tid);
        else h.qid[y.p] = qid;
        if (ok) h.c[y.p] = make_int4(res.pqs, res.pqe, res.pts, res.pte);
      }
      n_ok += ok ? 1u : 0u;
    }
    if (regroup_all) __syncthreads();  // (the next turn's regrouping reuses the scratch)
  }
  return n_ok;
}
#ifdef IMPG_STG_WAVES  // (experiments: force the register allocation that gives this many waves per SIMD)
#define STG_OCCUPANCY __attribute__((amdgpu_waves_per_eu(IMPG_STG_WAVES, IMPG_STG_WAVES)))
#else
#define STG_OCCUPANCY
#endif
// MASKS: the level's pairs are named by the count pass's hit masks (WindowLists with tile_first, a counting run's final
// level); else by the emit pass's list pair_entry[place] (slots by place, kernels.hpp).  Either way a block takes
// STG_RANGES consecutive ranges of the lookup order and all their places -- wl.pair_off[], the windows and the
// ranges' (start, end) go to LDS first, so a place finds its range, its entry and the range's ends without leaving the CU.
template <bool TRANSITIVE, bool MASKS, int OUT = OUT_SLOTS>
__global__ __launch_bounds__(STG_THREADS) STG_OCCUPANCY void project_staged_kernel(DeviceIndexView v, const uint32_t *__restrict__ pair_entry,
                                                      uint32_t n_pairs, HitArrays h, unsigned long long *__restrict__ accepted,
                                                      uint32_t *__restrict__ err_flag, int regroup, WindowLists wl) {
  // (XCD-contiguous mapping as in project_kernel)
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t sblock = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  __shared__ uint4 st_ent[STG_ENT_V4];
  __shared__ uint4 st_line[STG_LINE_V4];
  __shared__ uint4 st_win[STG_RANGES];
  __shared__ int2 st_se[STG_RANGES];
  __shared__ uint32_t st_off[STG_RANGES + 4u];
  __shared__ uint32_t st_dest[OUT == OUT_ROWS ? STG_RANGES : 1u], st_tid[OUT == OUT_ROWS ? STG_RANGES : 1u];  // (ordered rows: the ranges' first rows, their targets)
  __shared__ uint32_t wred[2u * STG_WAVES];
  __shared__ uint32_t wcnt[STG_WAVES];
  const uint32_t r0 = sblock * STG_RANGES;
  if (r0 >= wl.n_fr) return;  // (block-uniform: the grid is rounded up to the 8 XCDs)
  const uint32_t nr = min(STG_RANGES, wl.n_fr - r0);
  if (OUT == OUT_ROWS && threadIdx.x < nr) st_dest[threadIdx.x] = wl.ord.dest[r0 + threadIdx.x];
#ifdef IMPG_PHASE_CLOCKS
  unsigned long long stg_t[8], phase_t[10];
#define STG_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stg_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STG_MARK(i) do { } while (0)
#endif
  STG_MARK(0);
  // the block's ranges: place offsets, windows, ends; and the span of entries their hits lie in
  uint32_t emin = 0xFFFFFFFFu, emax = 0u;
  if (threadIdx.x <= STG_RANGES) {
    uint32_t o = 0xFFFFFFFFu;
    if (threadIdx.x < nr) o = wl.pair_off[r0 + threadIdx.x];
    else if (threadIdx.x == nr) o = r0 + nr < wl.n_fr ? wl.pair_off[r0 + nr] : n_pairs;
    st_off[threadIdx.x] = o;
    if (threadIdx.x < nr) {
      const uint4 w = wl.win[r0 + threadIdx.x];
      st_win[threadIdx.x] = w;
      { const FrontierRec sf = wl.se[r0 + threadIdx.x]; st_se[threadIdx.x] = make_int2(sf.start, sf.end); if (OUT == OUT_ROWS) st_tid[threadIdx.x] = sf.target_id; }
      if (w.y - (w.x & ~3u) > 64u) { emin = w.x; emax = w.y - 1u; }  // (a window wider than the mask: its hits lie somewhere in it)
      else if (w.z | w.w) {
        emin = w.x + (w.z ? (uint32_t)__builtin_ctz(w.z) : 32u + (uint32_t)__builtin_ctz(w.w));
        emax = w.x + (w.w ? 63u - (uint32_t)__builtin_clz(w.w) : 31u - (uint32_t)__builtin_clz(w.z));
      }
    }
  }
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      emin = min(emin, (uint32_t)__shfl_xor((int)emin, o));
      emax = max(emax, (uint32_t)__shfl_xor((int)emax, o));
    }
    if (lane_id() == 0) { wred[threadIdx.x >> 6] = emin; wred[STG_WAVES + (threadIdx.x >> 6)] = emax; }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < STG_WAVES; k++) { emin = min(emin, wred[k]); emax = max(emax, wred[STG_WAVES + k]); }
  }
  const uint32_t P0 = st_off[0], P1 = st_off[nr];
  uint32_t n_ok = 0;
  STG_MARK(1);
  // a sparse stretch of the level (its hits spread over many more entries than can be staged): nothing is staged, the
  // pairs are regrouped by entry and read the index as in project_kernel (the scratch overlays the line buffer)
  const bool sparse = emin <= emax && emax - emin >= 4u * STG_ECAP;
  uint32_t n_e = 0;
  if (emin <= emax && !sparse) {
    // stage 1: the entries emin .. emin + n_e - 1, four 16-byte vectors each
    n_e = min(STG_ECAP, emax - emin + 1u);
    if (threadIdx.x < n_e * 4u) st_ent[(threadIdx.x >> 2) * (STG_ENT_STRIDE / 4u) + (threadIdx.x & 3u)] = reinterpret_cast<const uint4 *>(v.entries + emin)[threadIdx.x];
    __syncthreads();
    // stage 2: their records' prefix lines, a wave per entry and turn, a lane per 16-byte piece of the record's <= 8 lines
    const uint32_t w = threadIdx.x >> 6, l = lane_id();
    constexpr uint32_t TURNS = (STG_ECAP + STG_WAVES - 1u) / STG_WAVES;
    uint4 buf[TURNS];
#pragma unroll
    for (uint32_t k = 0; k < TURNS; k++) {
      const uint32_t i = w + k * STG_WAVES;
      buf[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i < n_e) {
        const uint4 e1 = st_ent[i * (STG_ENT_STRIDE / 4u) + 1u];
        const uint32_t n = e1.z & OP_LEN_MASK, m = (n + TILE_OPS - 1u) / TILE_OPS;
        if (m <= INLINE_TILES && (l >> 3) < m) buf[k] = reinterpret_cast<const uint4 *>(v.pfx + (size_t)e1.y * TILE_WORDS)[l];
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < TURNS; k++) {
      const uint32_t i = w + k * STG_WAVES;
      if (i < n_e) st_line[i * (STG_REC_STRIDE / 4u) + (l >> 3) * (STG_LINE_STRIDE / 4u) + (l & 7u)] = buf[k];
    }
    __syncthreads();
  }
  STG_MARK(2);
  n_ok = project_places<TRANSITIVE, MASKS, true, STG_RANGES, STG_THREADS, 0, OUT>(v, pair_entry, h, accepted, err_flag, wl, r0, P0, P1, emin, n_e,
                                                                                  sparse && regroup != 0, st_off, st_win, st_se, st_ent, st_line PHASE_PASS,
                                                                                  0.0, st_dest, OUT == OUT_ROWS ? st_tid : nullptr);
#ifdef IMPG_PHASE_CLOCKS
  STG_MARK(3);
  if ((blockIdx.x & 15u) == 0u && threadIdx.x == 0u) {
    if (sparse) {
      atomicAdd(&g_phase_clk[13], stg_t[3] - stg_t[0]);
      atomicAdd(&g_phase_clk[14], 1ull);
      atomicAdd(&g_phase_clk[12], (unsigned long long)(emax - emin));
    } else {
      for (int i = 0; i < 3; i++) atomicAdd(&g_phase_clk[i], stg_t[i + 1] - stg_t[i]);
      atomicAdd(&g_phase_clk[9], (unsigned long long)(P1 - P0));
      atomicAdd(&g_phase_clk[10], emin <= emax && emax - emin >= STG_ECAP ? 1ull : 0ull);
      atomicAdd(&g_phase_clk[15], 1ull);
      atomicAdd(&g_phase_clk[11], emin <= emax ? (unsigned long long)(emax - emin + 1u) : 0ull);
    }
  }
#endif
  count_accepted<STG_WAVES>(n_ok, accepted, wcnt);
}
#undef STG_MARK

// ---------------------------------------------------------------------------
// A dense final level, ENTRY by entry (round 4): a wave works on one entry at a time, a lane per range that hits it.
//
// project_staged_kernel keeps project_kernel's shape -- a lane per place, so the 64 lanes of a wave read 64 different
// entries -- and on the headline's final level it is bound by the vector ALUs (0.65 busy) and by LDS bank conflicts
// (60 % of the LDS cycles): every lane decodes its own entry, selects on its own orientation flags, and the lanes'
// records collide on banks.  With the block's ranges and windows in LDS the pairs can just as well be listed the other
// way round: for one staged entry, the block's ranges whose hit mask has its bit -- one ballot per 64 ranges.  The
// wave then holds the entry's 16 words in SCALAR registers (its checkpoints are compared as scalar operands, nothing
// is selected per lane), its orientation is a compile-time constant of the code path taken (project_core's ORIENT),
// and its lanes read the SAME record's lines: equal addresses are one LDS access, different lines of the record sit
// on different banks.  A pair's slot is its range's place offset + the number of mask bits below the entry's: the
// same slots project_kernel's WindowLists form fills, so nothing downstream can tell.
// Ranges whose window is wider than the mask (their pairs are listed by the wave-per-range emit) are projected at the
// end from that list; a block whose pairs are few for the entries they touch takes project_places (nothing staged).
// ---------------------------------------------------------------------------
// IMPG_ENT_GROUP_SKIP = 1: an entry's enumeration only looks at the groups of 64 ranges whose masks can name it (one LDS read and
// a ballot pick one or two of the eight).  Measured, round 5: 21.7 -> 21.8-22.1 ms -- the eight unrolled ballots overlap with
// the record's LDS traffic and cost less than the loop that replaces them.  0 (the default): all eight.
#ifndef IMPG_ENT_GROUP_SKIP
#define IMPG_ENT_GROUP_SKIP 0
#endif
#ifndef IMPG_ENT_RANGES
#define IMPG_ENT_RANGES 512
#endif
#ifndef IMPG_ENT_THREADS
#define IMPG_ENT_THREADS 256
#endif
constexpr uint32_t ENT_RANGES = IMPG_ENT_RANGES;                          // ranges (consecutive in the lookup order) per block
constexpr uint32_t ENT_THREADS = IMPG_ENT_THREADS, ENT_WAVES = ENT_THREADS / 64u;
// IMPG_ENT_E0_LDS: a chunk's lanes also take the entry's coordinates (words 0 .. 3) from the LDS copy, as vector registers:
// an add or a compare with a scalar operand issues at half the rate of one on vector registers (profiles/r3_issue_rate.json),
// and four more scalars are free.  Projection of a headline step 19.1 -> 18.7 ms on the same box.
#ifndef IMPG_ENT_E0_LDS
#define IMPG_ENT_E0_LDS 1
#endif
constexpr uint32_t ENT_REC_STRIDE = ENT_CP_OFF + 16u;                     // words of a wave's LDS record (8 padded lines, then the entry: words 4 .. 15, 0 .. 3)
static_assert((ENT_RANGES & (ENT_RANGES - 1u)) == 0u && ENT_RANGES % ENT_THREADS == 0 && ENT_THREADS % 64u == 0, "whole turns of the block over its ranges");
constexpr uint32_t ENT_REC_V4 = ENT_WAVES * ENT_REC_STRIDE / 4u, ENT_LIST_V4 = ENT_WAVES * ENT_RANGES * 2u / 16u;
static_assert((ENT_REC_V4 + ENT_LIST_V4) * 4u >= 5u * ENT_THREADS + ENT_WAVES, "the unstaged path's scratch overlays the waves' records and lists");
// Waves per SIMD the register allocation is held to.  Round 5: with an entry in flight held as four registers instead of
// sixteen (below) the kernel needs 106 vector registers where it needed 126, and held to 96 / 80 it spills 12 / 44 bytes
// outside the chunk loop: projection of a headline step 20.8 ms at 4 waves (what the allocator picks on its own),
// 19.2 at 5, 18.8 at 6 (the block's 24 KB of LDS allow no more).
#ifndef IMPG_ENT_WAVES
#define IMPG_ENT_WAVES 6
#endif
#if IMPG_ENT_WAVES > 0
#define ENT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(IMPG_ENT_WAVES, IMPG_ENT_WAVES)))
#else
#define ENT_OCCUPANCY
#endif
// (Round 5, measured and dropped: a fast path without the second candidate test -- the neighbouring op chosen up front by two
// comparisons when the located op only touches the threshold, ~2 % of ends; the 0.1 % of pairs that still fail listed in
// LDS and projected 64 at a time by the general form at one place of a flattened (entry, chunk) loop.  Exact -- every test
// green with IMPG_STAGE_DENSITY=0 -- and 8 KB less code, but 374 VALU per 64 pairs where this form runs 338
// (SQ_INSTS_VALU): the flat loop alone, with the old two-candidate search, costs the same 21.9 -> 23.6 ms -- carried
// around it the entry's sixteen scalar words and the chunk counters spill to vector lanes -- and the second test, which
// only 60 % of the chunks execute at all, is worth what the choice costs.)
template <bool TRANSITIVE, int ORIENT, int MODE, int OUT>
__device__ __forceinline__ void project_entry_chunk(const DeviceIndexView &v, uint4 e0, uint4 e1, uint4 e2, uint4 e3, uint32_t eidx, bool live,
                                                    int32_t f_start, int32_t f_end, uint32_t p, const uint32_t *rec, const HitArrays &h,
                                                    unsigned long long *__restrict__ accepted, uint32_t *__restrict__ err_flag,
                                                    uint32_t &n_ok PHASE_ARG, double min_identity, const OrderedOut &ord, uint32_t tid) {
  const SliceArrays no_sl{nullptr, nullptr, nullptr, nullptr};
  if (live) {
    bool ok = false;
    TileScan res;
    res.found = res.any = false;
    res.pqs = res.pts = res.pqe = res.pte = -1;
    uint32_t qid = HIT_NONE;
    if (IMPG_ENT_E0_LDS && IMPG_ENT_CP_LDS && ORIENT >= 0) {
      e0 = reinterpret_cast<const uint4 *>(__builtin_assume_aligned(rec, 16))[ENT_CP_OFF / 4u + 3u];
      asm volatile("" : "+v"(e0.x), "+v"(e0.y), "+v"(e0.z), "+v"(e0.w));
    }
    if (ORIENT >= 0)
      project_core<TRANSITIVE, MODE, true, ORIENT>(v, e0, e1, e2, e3, f_start, f_end, p, min_identity, err_flag, no_sl, accepted, ok, qid, res PHASE_PASS, rec);
    else  // (a record with more prefix lines than a staged record holds: from the index)
      project_pair<TRANSITIVE, MODE>(v, eidx, f_start, f_end, p, min_identity, err_flag, no_sl, accepted, ok, qid, res PHASE_PASS);
    if (OUT == OUT_ROWS) put_ordered_row(ord, p, qid, tid, ok, res.pqs, res.pqe, res.pts, res.pte);  // (p = the row's final place)
    else {
      // (qs: the slot's query id and its source -- the range's place, handed in as `tid` -- as ONE 8-byte store into an
      // interleaved array: a third store instruction per chunk cost the level 2.3 ms, this one nothing measurable)
      if (OUT == OUT_QS) reinterpret_cast<uint2 *>(h.qid)[p] = make_uint2(qid, tid);
      else h.qid[p] = qid;
      if (ok) h.c[p] = make_int4(res.pqs, res.pqe, res.pts, res.pte);
    }
    n_ok += ok ? 1u : 0u;
  }
}
// MODE: 0, or MODE_IDENT -- the identity filter (round 5: `--min-result-identity` used to send the whole final level back to the
// lane-per-pair kernel, 42 ms of projection a headline step against 21.7 plain); the slice's counts come off the identity
// lines in the index (a wave's 64 lanes read the same record's <= 8 lines: L1 hits), everything else as in the plain form.
// Heavy blocks are SLICED (round 6): a block whose 512 ranges list more than ENT_SLICE_PAIRS pairs -- repeat hot spots: a
// thousand hits a range, 7 x 10^5 pairs where the headline's block has 13 000 -- would run alone on its CU long after the rest
// of the grid has drained (the skewed workload's final level: ~80 such blocks, 3 x 10^9 pairs/s).  Its first launch
// (phase 0) takes every n-th entry of the span and leaves the other n - 1 slices on a work list for a second launch
// (phase 1), whose blocks load the same ranges and take the other entries; the slices share the block's places through
// one global counter a range block.
#ifndef IMPG_ENT_SLICE_PAIRS
#define IMPG_ENT_SLICE_PAIRS 32768
#endif
constexpr uint32_t ENT_SLICE_PAIRS = IMPG_ENT_SLICE_PAIRS, ENT_MAX_SLICES = 64u;
#ifndef IMPG_ENT_SPARSE
#define IMPG_ENT_SPARSE 4
#endif
constexpr uint32_t ENT_SPARSE = IMPG_ENT_SPARSE;  // a block with fewer pairs than this per entry of its span runs a lane per place instead
struct EntSlices {
  uint32_t *work;   // [0] items listed, [1 ..] sblock << 12 | slice << 6 | (slices - 1)
  uint32_t *alloc;  // [range blocks] places handed out so far (zeroed before phase 0)
  uint32_t cap;     // items the list takes
  int phase;
};
template <bool TRANSITIVE, int MODE, int OUT>
__global__ __launch_bounds__(ENT_THREADS) ENT_OCCUPANCY void project_entries_kernel(DeviceIndexView v, const uint32_t *__restrict__ pair_entry, uint32_t n_pairs,
                                                      HitArrays h, unsigned long long *__restrict__ accepted,
                                                      uint32_t *__restrict__ err_flag, int regroup, WindowLists wl, double min_identity, EntSlices sl) {
  const uint32_t per_xcd = gridDim.x >> 3;
  uint32_t sblock = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  uint32_t slice = 0u, n_slices = 1u;
  if (sl.phase) {  // one listed slice a block (the launch covers the list's capacity; most of its blocks leave here)
    if (blockIdx.x >= min(sl.work[0], sl.cap)) return;
    const uint32_t item = sl.work[1u + blockIdx.x];
    sblock = item >> 12; slice = (item >> 6) & 63u; n_slices = (item & 63u) + 1u;
  }
  __shared__ uint4 st_work[ENT_REC_V4 + ENT_LIST_V4];
  uint4 (*st_rec)[ENT_REC_STRIDE / 4u] = reinterpret_cast<uint4 (*)[ENT_REC_STRIDE / 4u]>(st_work);  // a wave's current record: its prefix lines (padded)
  uint16_t (*st_list)[ENT_RANGES] = reinterpret_cast<uint16_t (*)[ENT_RANGES]>(st_work + ENT_REC_V4);  // a wave's list of the ranges that hit its entry
  __shared__ uint4 st_win[ENT_RANGES];
  __shared__ int2 st_se[ENT_RANGES];
  __shared__ uint32_t st_off[ENT_RANGES + 4u];
  __shared__ uint16_t st_wide[ENT_RANGES];
  __shared__ uint32_t st_dest[OUT == OUT_ROWS ? ENT_RANGES : 1u];  // ordered rows (OrderedOut): the row of every range's first slot
  constexpr bool ordered = OUT == OUT_ROWS;
  constexpr bool qs = OUT == OUT_QS;  // slots as {query id, the range's place} pairs in h.qid (a kept fused level)
  constexpr bool WIDE_LISTED = ordered;  // windows wider than the hit mask: listed (pair_entry, in visit order) or by the window test
#if IMPG_ENT_GROUP_SKIP
  __shared__ int2 st_grp[ENT_RANGES / 64u];  // per 64 ranges: the first and the last entry their masks name
#endif
  __shared__ uint32_t st_nwide, st_alloc, st_next;
  __shared__ uint32_t wred[2u * ENT_WAVES];
  __shared__ uint32_t wcnt[ENT_WAVES];
  const uint32_t r0 = sblock * ENT_RANGES;
  if (r0 >= wl.n_fr) return;  // (block-uniform: the grid is rounded up to the 8 XCDs)
  const uint32_t nr = min(ENT_RANGES, wl.n_fr - r0);
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = lane_id();
#ifdef IMPG_PHASE_CLOCKS
  unsigned long long stg_t[8], phase_t[10];
#define STG_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stg_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STG_MARK(i) do { } while (0)
#endif
  STG_MARK(0);
  if (threadIdx.x == 0) { st_nwide = 0u; st_alloc = 0u; st_next = 0u; }
  __syncthreads();
  // the block's ranges: place offsets, windows, ends; the span of entries their masks name; the ranges listed instead
  uint32_t emin = 0xFFFFFFFFu, emax = 0u;
#pragma unroll
  for (uint32_t t = threadIdx.x; t < ENT_RANGES; t += ENT_THREADS) {
    // (st_off[nr] = where the block's places end, all ones beyond: the searches never step past the block's ranges)
    if (t < nr) st_off[t] = wl.pair_off[r0 + t];
    else st_off[t + 1u] = 0xFFFFFFFFu;
    if (t == 0) st_off[nr] = r0 + nr < wl.n_fr ? wl.pair_off[r0 + nr] : n_pairs;
    uint32_t glo = 0xFFFFFFFFu, ghi = 0u;
    if (t < nr) {
      const uint4 w = wl.win[r0 + t];
      st_win[t] = w;
      { const FrontierRec sf = wl.se[r0 + t]; st_se[t] = make_int2(sf.start, sf.end); }
      if (ordered) st_dest[t] = wl.ord.dest[r0 + t];
      if (w.y - (w.x & ~3u) > 64u) {
        // a window wider than the hit mask (a dense target, a repeat hot spot).  Where slots may be filled entry by entry its hits
        // join the entry-major loop below, named by the window test itself instead of a mask bit; ordered rows need the hit's
        // visit position, which only the listed form has: those ranges are listed and taken a lane per place at the end
        if (WIDE_LISTED) st_wide[atomicAdd(&st_nwide, 1u)] = (uint16_t)t;
        else if (w.x < w.y) { glo = w.x; ghi = w.y - 1u; emin = min(emin, glo); emax = max(emax, ghi); }
      } else if (w.z | w.w) {
        glo = w.x + (w.z ? (uint32_t)__builtin_ctz(w.z) : 32u + (uint32_t)__builtin_ctz(w.w));
        ghi = w.x + (w.w ? 63u - (uint32_t)__builtin_clz(w.w) : 31u - (uint32_t)__builtin_clz(w.z));
        emin = min(emin, glo);
        emax = max(emax, ghi);
      }
    }
#if IMPG_ENT_GROUP_SKIP
    {  // the 64 ranges this wave has just loaded are one group of the enumeration below
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        glo = min(glo, (uint32_t)__shfl_xor((int)glo, o));
        ghi = max(ghi, (uint32_t)__shfl_xor((int)ghi, o));
      }
      if (l == 0) st_grp[t >> 6] = make_int2((int32_t)glo, (int32_t)ghi);  // (an empty group: lo > hi, nothing lies between)
    }
#endif
  }
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      emin = min(emin, (uint32_t)__shfl_xor((int)emin, o));
      emax = max(emax, (uint32_t)__shfl_xor((int)emax, o));
    }
    if (l == 0) { wred[wv] = emin; wred[ENT_WAVES + wv] = emax; }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ENT_WAVES; k++) { emin = min(emin, wred[k]); emax = max(emax, wred[ENT_WAVES + k]); }
    emin = (uint32_t)__builtin_amdgcn_readfirstlane((int)emin);
    emax = (uint32_t)__builtin_amdgcn_readfirstlane((int)emax);
  }
  const uint32_t P0 = st_off[0], P1 = st_off[nr];
  // Which slot a pair lands in is free here (a counting run's final level: nothing reads its slots by position or in
  // order, only together with range_out), so the block's pairs fill its places [P0, P1) ENTRY BY ENTRY -- an entry's
  // pairs take the next run of places (an LDS counter) and a wave's 64 results are one contiguous store.  By range
  // (place offset + mask bits below the entry's) the lanes of a wave scatter over as many lines as it has lanes, and
  // the store path, not the ALUs, bounded the kernel.  Not when a range of the block is listed instead: its pairs
  // keep their places, so the others do too.
  const bool compact = !ordered;  // (by range -- slot = the range's first place + the mask bits below the entry's -- measured 2.3 x slower: every lane of a store another line)
  uint32_t n_ok = 0;
  STG_MARK(1);
  // few pairs for the entries they touch (a sparse stretch of the level): fetching a record for a pair or two would read
  // more than the pairs do -- by place, regrouped, from the index (the ranges listed instead take the same path there)
  const bool sparse = emin <= emax && (unsigned long long)(P1 - P0) < (unsigned long long)ENT_SPARSE * (emax - emin + 1u);
  if (!sl.phase && sl.work && !sparse && emin <= emax && P1 - P0 > ENT_SLICE_PAIRS && (uint32_t)__builtin_amdgcn_readfirstlane((int)st_nwide) == 0u) {
    // (block-uniform) a heavy block: this launch takes slice 0, the others go onto the list (which cannot overflow: the
    // slices beyond the first number at most P / ENT_SLICE_PAIRS over the whole level, and that is its capacity)
    n_slices = min(ENT_MAX_SLICES, (P1 - P0 + ENT_SLICE_PAIRS - 1u) / ENT_SLICE_PAIRS);
    if (threadIdx.x == 0) {
      const uint32_t at = atomicAdd(&sl.work[0], n_slices - 1u);
      for (uint32_t k = 1; k < n_slices; k++)
        if (at + k - 1u < sl.cap) sl.work[1u + at + k - 1u] = sblock << 12 | k << 6 | (n_slices - 1u);
    }
  }
  if (sparse) {
    n_ok = project_places<TRANSITIVE, true, false, ENT_RANGES, ENT_THREADS, MODE, OUT>(v, pair_entry, h, accepted, err_flag, wl, r0, P0, P1, emin, 0u, regroup != 0, st_off, st_win,
                                                               st_se, nullptr, st_work PHASE_PASS, min_identity, st_dest);
  } else if (emin <= emax) {
    // The waves take the span's entries one by one off an LDS counter (an entry is anything from a handful to 500 pairs:
    // dealt round-robin, a block waited for its unluckiest wave).  Each entry is fetched by its wave alone -- its 64
    // bytes (one address for the whole wave, then scalar registers), then its record's <= 8 prefix lines, a 16-byte
    // piece per lane, into the wave's own LDS record -- two entries ahead / one record ahead of the one being worked
    // on, so that the index's latency hides behind the projections.  No barrier: the block shares only the ranges' data.
    const uint4 *ents = reinterpret_cast<const uint4 *>(v.entries);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t n_span = emax - emin + 1u;
    auto take = [&]() -> uint32_t {  // the next entry of the span (>= n_span: none left); a slice takes every n_slices-th
      uint32_t i = 0;
      if (l == 0) i = atomicAdd(&st_next, 1u);
      i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
      return n_slices > 1u ? min(n_span, slice + i * n_slices) : i;
    };
    uint32_t ic = take(), in = take();
    // (an entry's 64 bytes are held as one 16-byte piece in each of lanes 0 .. 3 -- four registers for an entry in flight
    // instead of sixteen with every lane reading all four pieces: with two entries in flight that is 24 registers of
    // the kernel's 126; its words are picked off the lanes into scalar registers when its turn comes)
    uint4 cq = zero4, nq = zero4, lc = zero4;
    auto fetch_entry = [&](uint32_t i) -> uint4 {
      uint4 q = zero4;
      if (i < n_span && l < 4u) q = ents[(size_t)(emin + i) * 4u + l];
      return q;
    };
#define IMPG_RDL(x, j) (uint32_t)__builtin_amdgcn_readlane((int)(x), j)
    auto fetch_record = [&](const uint4 &q) -> uint4 {  // the record's <= 8 prefix lines, a 16-byte piece per lane
      uint4 r = zero4;
      const uint32_t nz = IMPG_RDL(q.z, 1), ny = IMPG_RDL(q.y, 1);  // the entry's words 6 and 5: op count | flags, first tile
      const uint32_t m = ((nz & OP_LEN_MASK) + TILE_OPS - 1u) / TILE_OPS;
      if (m <= INLINE_TILES && (l >> 3) < m) r = reinterpret_cast<const uint4 *>(v.pfx + (size_t)ny * TILE_WORDS)[l];
      return r;
    };
    cq = fetch_entry(ic);
    nq = fetch_entry(in);
    if (ic < n_span) lc = fetch_record(cq);
#pragma unroll 1
    while (ic < n_span) {
      const uint32_t eidx = emin + ic;
      // this entry's record into the wave's LDS record (the previous entry's reads are done: their results were used)
      __builtin_amdgcn_wave_barrier();
      st_rec[wv][(l >> 3) * (STG_LINE_STRIDE / 4u) + (l & 7u)] = lc;
      if (IMPG_ENT_CP_LDS && l < 4u) st_rec[wv][ENT_CP_OFF / 4u + ((l + 3u) & 3u)] = cq;  // (words 4 .. 15, then 0 .. 3)
      uint4 e0, e1, e2, e3;
      e0.x = IMPG_RDL(cq.x, 0); e0.y = IMPG_RDL(cq.y, 0); e0.z = IMPG_RDL(cq.z, 0); e0.w = IMPG_RDL(cq.w, 0);
      e1.x = IMPG_RDL(cq.x, 1); e1.y = IMPG_RDL(cq.y, 1); e1.z = IMPG_RDL(cq.z, 1); e1.w = IMPG_RDL(cq.w, 1);
      e2.x = IMPG_RDL(cq.x, 2); e2.y = IMPG_RDL(cq.y, 2); e2.z = IMPG_RDL(cq.z, 2); e2.w = IMPG_RDL(cq.w, 2);
      e3.x = IMPG_RDL(cq.x, 3); e3.y = IMPG_RDL(cq.y, 3); e3.z = IMPG_RDL(cq.z, 3); e3.w = IMPG_RDL(cq.w, 3);
      // the next entry of the wave moves up; its record and the entry after it are requested
      ic = in;
      cq = nq;
      lc = zero4;
      if (ic < n_span) {
        lc = fetch_record(cq);
        in = take();
        nq = fetch_entry(in);
      }
      // the block's ranges that hit the entry, 64 at a time: bit (entry - window start) of the range's mask
      uint32_t cnt = 0;
      // (what the count pass tested a wide window's entries with -- ends[] / ends_t[] -- from the entry's own words: no load)
      const int32_t e_end = TRANSITIVE ? ((int32_t)e0.x < (int32_t)e0.y ? (int32_t)e0.y : (-2147483647 - 1)) : (int32_t)e0.y;
#if IMPG_ENT_GROUP_SKIP
      // (in the lookup order a block's 512 ranges climb through its ~50 entries: one or two of the eight groups of 64 can
      // name a given entry at all -- found with one LDS read and a ballot; the other groups' masks are not looked at)
      const int2 gb = st_grp[l & (ENT_RANGES / 64u - 1u)];
      uint32_t groups = (uint32_t)__ballot(l < ENT_RANGES / 64u && eidx >= (uint32_t)gb.x && eidx <= (uint32_t)gb.y);
#pragma unroll 1
      for (; groups; groups &= groups - 1u) {
        const uint32_t q = (uint32_t)__builtin_ctz(groups);
        const uint32_t r = q * 64u + l;
#else
#pragma unroll
      for (uint32_t q = 0; q < ENT_RANGES / 64u; q++) {
        const uint32_t r = q * 64u + l;
#endif
        bool hit = false;
        if (r < nr) {
          const uint4 w = st_win[r];
          const uint32_t d = eidx - w.x;
          if (w.y - (w.x & ~3u) <= 64u) hit = d < 64u && (((d < 32u ? w.z >> d : w.w >> (d - 32u)) & 1u) != 0u);
          else if (!WIDE_LISTED) hit = eidx >= w.x && eidx < w.y && window_hit<TRANSITIVE>(e_end, st_se[r].x);  // the count pass's own test
        }
        const unsigned long long b = __ballot(hit);
        if (hit) st_list[wv][cnt + (uint32_t)__popcll(b & lanemask_lt())] = (uint16_t)r;
        cnt += (uint32_t)__popcll(b);
      }
      uint32_t run = 0;  // compact: the entry's first place
      if (compact) {
        if (l == 0) run = n_slices > 1u ? atomicAdd(&sl.alloc[sblock], cnt) : atomicAdd(&st_alloc, cnt);  // (slices share the block's places)
        run = P0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const uint32_t *rec = reinterpret_cast<const uint32_t *>(&st_rec[wv][0]);
      const int orient = (e1.z & OP_LEN_MASK) > INLINE_TILES * TILE_OPS ? -1 : !(e1.z & EF_REVERSED) ? 0 : (e1.z & EF_STRAND) ? 2 : 1;
      // ordered rows: the target of every row of this entry is the sequence its ranges look up -- the first one's says it
      uint32_t tid = 0;
      if (ordered && cnt) tid = wl.se[r0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)st_list[wv][0])].target_id;
#pragma unroll 1
      for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
        const bool live = k0 + l < cnt;
        const uint32_t r = live ? (uint32_t)st_list[wv][k0 + l] : 0u;
        const int2 se = st_se[r];
        uint32_t p = run + k0 + l;
        if (!compact) {  // slot = the range's first place + the mask bits below the entry's (ordered rows / the by-range experiment: narrow windows only)
          const uint4 w = st_win[r];
          const uint32_t d = eidx - w.x;
          const uint32_t below = d < 32u ? (uint32_t)__popc(w.z & ((1u << d) - 1u))
                                         : (uint32_t)__popc(w.z) + (uint32_t)__popc(w.w & ((1u << (d - 32u)) - 1u));
          p = st_off[r] + below;
        }
        if (ordered) {  // the row's final place: the range's first row + the hit's visit position (the byte the lookup left at its place)
          if (live) p = st_dest[r] + (uint32_t)wl.ord.vpos[p];
        } else if (qs) tid = r0 + r;
        else if (live && wl.range_out) wl.range_out[p] = wl.range_places ? r0 + r : wl.perm[r0 + r];
        if (orient == 0) project_entry_chunk<TRANSITIVE, 0, MODE, OUT>(v, e0, e1, e2, e3, eidx, live, se.x, se.y, p, rec, h, accepted, err_flag, n_ok PHASE_PASS, min_identity, wl.ord, tid);
        else if (orient == 1) project_entry_chunk<TRANSITIVE, 1, MODE, OUT>(v, e0, e1, e2, e3, eidx, live, se.x, se.y, p, rec, h, accepted, err_flag, n_ok PHASE_PASS, min_identity, wl.ord, tid);
        else if (orient == 2) project_entry_chunk<TRANSITIVE, 2, MODE, OUT>(v, e0, e1, e2, e3, eidx, live, se.x, se.y, p, rec, h, accepted, err_flag, n_ok PHASE_PASS, min_identity, wl.ord, tid);
        else project_entry_chunk<TRANSITIVE, -1, MODE, OUT>(v, e0, e1, e2, e3, eidx, live, se.x, se.y, p, rec, h, accepted, err_flag, n_ok PHASE_PASS, min_identity, wl.ord, tid);
      }
    }
#undef IMPG_RDL
#ifdef IMPG_PHASE_CLOCKS
    STG_MARK(2);
    if ((blockIdx.x & 15u) == 0u && threadIdx.x == 0u) {
      atomicAdd(&g_phase_clk[0], stg_t[1] - stg_t[0]);
      atomicAdd(&g_phase_clk[2], stg_t[2] - stg_t[1]);
      atomicAdd(&g_phase_clk[9], (unsigned long long)(P1 - P0));
      atomicAdd(&g_phase_clk[15], 1ull);
      atomicAdd(&g_phase_clk[11], (unsigned long long)(emax - emin + 1u));
    }
#endif
  }
  // the ranges whose pairs are listed (windows wider than the mask), a lane per place, from the index
  if (!sparse) {
    const SliceArrays no_sl{nullptr, nullptr, nullptr, nullptr};
    const uint32_t nw = st_nwide;
#pragma unroll 1
    for (uint32_t k = 0; k < nw; k++) {
      const uint32_t r = st_wide[k];
      const uint32_t a = st_off[r], b = st_off[r + 1u];
      const int2 se = st_se[r];
#pragma unroll 1
      for (uint32_t pp = a + threadIdx.x; pp < b; pp += ENT_THREADS) {
        bool ok = false;
        TileScan res;
        res.found = res.any = false;
        res.pqs = res.pts = res.pqe = res.pte = -1;
        uint32_t qid = HIT_NONE;
        project_pair<TRANSITIVE, MODE>(v, pair_entry[pp], se.x, se.y, pp, min_identity, err_flag, no_sl, accepted, ok, qid, res PHASE_PASS);
        if (ordered)  // (a listed window's entries are in visit order: the slot's position in the run is its visit position)
          put_ordered_row(wl.ord, st_dest[r] + (pp - a), qid, wl.se[r0 + r].target_id, ok, res.pqs, res.pqe, res.pts, res.pte);
        else {
          if (qs) reinterpret_cast<uint2 *>(h.qid)[pp] = make_uint2(qid, r0 + r);
          else h.qid[pp] = qid;
          if (ok) h.c[pp] = make_int4(res.pqs, res.pqe, res.pts, res.pte);
        }
        n_ok += ok ? 1u : 0u;
        if (wl.range_out && !ordered && !qs) wl.range_out[pp] = wl.range_places ? r0 + r : wl.perm[r0 + r];
      }
    }
  }
#ifdef IMPG_PHASE_CLOCKS
  if (sparse && (blockIdx.x & 15u) == 0u && threadIdx.x == 0u) {
    STG_MARK(3);
    atomicAdd(&g_phase_clk[13], stg_t[3] - stg_t[0]);
    atomicAdd(&g_phase_clk[14], 1ull);
    atomicAdd(&g_phase_clk[12], (unsigned long long)(emax - emin));
  }
#endif
  count_accepted<ENT_WAVES>(n_ok, accepted, wcnt);
}
#undef STG_MARK
#ifdef IMPG_PHASE_CLOCKS
extern "C" void impg_gpu_debug_phase_clocks(unsigned long long *out) {
  unsigned long long z[16] = {};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_clk), sizeof(z));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clk), z, sizeof(z));
}
#endif

// ---------------------------------------------------------------------------
// Approximate mode on tracepoint alignments: scan_overlapping_tracepoints + project_overlapping_interval_fast
// (impg.rs:646-823, :1317-1533).  The reference walks an alignment's trace segments from its start and keeps the
// first and the last one that overlap the range plus sums over all of them; segment positions are monotone along
// the scan axis, so with the per-boundary prefix sums {sum |tracepoint|, sum query delta, sum matches, sum
// mismatches} the index stores (index_build.cpp) the first / last overlapping segments are two binary searches
// and the sums two differences.  What remains is the reference's boundary refinement, an f64 interpolation inside
// the first and the last segment, done here with the same operations in the same order (division, two
// multiplications -- nothing a fused multiply-add could contract --, round half away from zero).
// One lane per (range, entry) pair, same pair lists and slot discipline as project_kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t tp_refine(int32_t query_pos, int32_t query_delta, int32_t seg_start, int32_t overlap_pos,
                                             int32_t abs_target_delta, bool first, int32_t lo, int32_t hi) {
  int32_t refined;
  if (abs_target_delta == 0) {  // impg.rs:1381-1398: a pure insertion maps to one target point
    refined = first ? query_pos : query_pos + query_delta;
  } else {
    const double target_fraction = (double)(overlap_pos - seg_start) / (double)abs_target_delta;
    const double indel_ratio = (double)query_delta / (double)abs_target_delta;
    const double query_advance = target_fraction * (double)abs_target_delta * indel_ratio;
    const double r = round(query_advance);
    const int32_t adv = r >= 2147483647.0 ? 2147483647 : r <= -2147483648.0 ? (-2147483647 - 1) : (int32_t)r;  // `as i32` saturates
    refined = (int32_t)((uint32_t)query_pos + (uint32_t)adv);
  }
  return min(max(refined, lo), hi);
}
template <bool TRANSITIVE>
__global__ __launch_bounds__(256) void project_tp_kernel(DeviceIndexView v, const FrontierRec *__restrict__ fr,
                                                         const uint32_t *__restrict__ pair_range,
                                                         const uint32_t *__restrict__ pair_entry, uint32_t n_pairs,
                                                         HitArrays h, unsigned long long *__restrict__ accepted,
                                                         uint32_t *__restrict__ err_flag, double min_identity, int use_ident,
                                                         ProjList pl, const uint32_t *__restrict__ n_pairs_dev) {
  if (n_pairs_dev) n_pairs = *n_pairs_dev;
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t lblock = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);  // one contiguous eighth of the pair list per XCD
  const uint32_t pp = lblock * 256u + threadIdx.x;
  bool ok = false;
  if (pp < n_pairs) {
    const uint32_t p = pl.slot ? pl.slot[pp] : pp;
    const uint32_t r = pl.slot ? pl.range[pp] : pair_range[p];
    const FrontierRec f = fr[r];
    const uint4 *ep = reinterpret_cast<const uint4 *>(v.entries + (pl.slot ? pl.entry[pp] : pair_entry[p]));
    const uint4 e0 = ep[0], e1 = ep[1];
    const int32_t m_ts = (int32_t)e0.x, m_te = (int32_t)e0.y, m_qs = (int32_t)e0.z, m_qe = (int32_t)e0.w;  // the entry's metadata axes
    const uint32_t n = e1.z & OP_LEN_MASK;
    const bool is_reverse = (e1.z & EF_STRAND) != 0, reversed_entry = (e1.z & EF_REVERSED) != 0;
    int32_t rs = f.start, re = f.end;
    if (TRANSITIVE) { rs = max(rs, m_ts); re = min(re, m_te); }  // the clipped overlap (impg.rs:2398-2400)
    uint32_t qid = HIT_NONE;
    int4 out = make_int4(0, 0, 0, 0);
    if (!(m_ts >= re || m_te <= rs) && n) {  // impg.rs:1327-1329
      const uint4 *B = reinterpret_cast<const uint4 *>(v.ops) + e1.y;  // boundaries 0..n of the record
      // scan axis / project axis of this entry (impg.rs:676-713)
      const int sdir = reversed_entry ? 1 : (is_reverse ? -1 : 1);
      const int pdir = reversed_entry ? (is_reverse ? -1 : 1) : 1;
      const int32_t s0 = reversed_entry ? m_ts : (is_reverse ? m_te : m_ts);
      const int32_t p0 = reversed_entry ? (is_reverse ? m_qe : m_qs) : m_qs;
      // S = the scan-axis prefix (x = sum |tracepoint| for a forward entry, y = sum query delta for a reversed one)
      const int64_t x_lo = sdir > 0 ? (int64_t)rs - s0 : (int64_t)s0 - re, x_hi = sdir > 0 ? (int64_t)re - s0 : (int64_t)s0 - rs;
      auto S = [&](uint32_t k) -> int64_t { const uint4 b = B[k]; return (int64_t)(reversed_entry ? b.y : b.x); };
      // first = #{i in [1, n] : S[i] <= x_lo}: the first segment whose far end passes the range's near end
      uint32_t lo = 1, hi = n + 1;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S(mid) <= x_lo) lo = mid + 1; else hi = mid; }
      const uint32_t first = lo - 1;
      // last = #{i in [0, n) : S[i] < x_hi} - 1: the last segment that starts before the range's far end
      lo = 0; hi = n;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S(mid) < x_hi) lo = mid + 1; else hi = mid; }
      if (first < n && lo > 0 && first <= lo - 1) {
        const uint32_t last = lo - 1;
        const uint4 bf0 = B[first], bf1 = B[first + 1], bl0 = B[last], bl1 = B[last + 1];
        auto seg = [&](const uint4 &a, const uint4 &b, int32_t &ppos, int32_t &pdelta, int32_t &sstart, int32_t &send, int32_t &asd) {
          const int32_t sa = (int32_t)(reversed_entry ? a.y : a.x), sb = (int32_t)(reversed_entry ? b.y : b.x);
          const int32_t ja = (int32_t)(reversed_entry ? a.x : a.y), jb = (int32_t)(reversed_entry ? b.x : b.y);
          const int32_t pa = s0 + sdir * sa, pb = s0 + sdir * sb;
          sstart = min(pa, pb); send = max(pa, pb); asd = sb - sa;
          ppos = p0 + pdir * ja; pdelta = pdir * (jb - ja);
        };
        const int32_t qlo = min(m_qs, m_qe), qhi = max(m_qs, m_qe);
        int32_t ppos, pdelta, sstart, send, asd;
        seg(bf0, bf1, ppos, pdelta, sstart, send, asd);
        const int32_t refined_first = tp_refine(ppos, pdelta, sstart, max(sstart, rs), asd, true, qlo, qhi);
        seg(bl0, bl1, ppos, pdelta, sstart, send, asd);
        const int32_t refined_last = tp_refine(ppos, pdelta, sstart, min(send, re), asd, false, qlo, qhi);
        bool keep = true;
        if (use_ident) {  // the approximate CIGAR "M= X X" (impg.rs:1476-1491)
          const int64_t mm = (int64_t)bl1.z - bf0.z, xx = (int64_t)bl1.w - bf0.w;
          const int64_t total = mm + xx;
          const double ident = total == 0 ? 0.0 : (double)mm / (double)total;
          keep = !(ident < min_identity);
        }
        if (keep) {
          if (refined_first < 0 || refined_last < 0) atomicOr(err_flag, 2u);  // the reference panics (impg.rs:1509-1514)
          const bool swap_q = is_reverse && !reversed_entry;  // impg.rs:1497-1501
          out = make_int4(swap_q ? refined_last : refined_first, swap_q ? refined_first : refined_last, rs, re);
          qid = e1.x;
          ok = true;
        }
      }
    }
    h.qid[p] = qid;
    if (ok) h.c[p] = out;
  }
  __shared__ uint32_t wcnt[4];
  unsigned long long m = __ballot(ok);
  if (lane_id() == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (tot) atomicAdd(&accepted[(blockIdx.x % COUNT_SLOTS) * COUNT_STRIDE], (unsigned long long)tot);
  }
}

// ---------------------------------------------------------------------------
// store_cigar: materialise the projected CIGAR slices (impg.rs:2878-2886)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slice_counts_kernel(HitArrays h, SliceArrays sl, uint32_t n_pairs,
                                                           uint32_t *__restrict__ cnt) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  cnt[p] = h.qid[p] == HIT_NONE ? 0u : sl.n[p];
}
__global__ __launch_bounds__(64) void slice_write_kernel(DeviceIndexView v, const uint32_t *__restrict__ pair_entry,
                                                         HitArrays h, SliceArrays sl, uint32_t n_pairs,
                                                         const uint32_t *__restrict__ off, uint32_t *__restrict__ out) {
  const uint32_t p = blockIdx.x * 64u + threadIdx.x;
  if (p >= n_pairs || h.qid[p] == HIT_NONE) return;
  const Entry &en = v.entries[pair_entry[p]];
  const bool swp = (en.nops_flags & EF_REVERSED) != 0, flip = swp && (en.nops_flags & EF_STRAND);
  const uint32_t *rec = v.ops + (size_t)en.tile_base * TILE_WORDS;
  const uint32_t n = sl.n[p], a = sl.a[p];
  uint32_t *o = out + off[p];
  for (uint32_t t = 0; t < n; t++) {
    const uint32_t oi = flip ? a - t : a + t;  // reverse-strand reversed entries list their ops back to front
    uint32_t op = rec[(size_t)(oi / TILE_OPS) * TILE_WORDS + 6 + oi % TILE_OPS];
    if (swp) {  // invert_cigar_ops_in_place: I <-> D (impg.rs:146-151)
      const uint32_t code = op >> 29;
      if (code == 2u) op = (3u << 29) | (op & OP_LEN_MASK);
      else if (code == 3u) op = (2u << 29) | (op & OP_LEN_MASK);
    }
    o[t] = op;
  }
  // adjust_len (impg.rs:137-139), same arithmetic: val = (val & (7 << 29)) | ((len + delta) as u32)
  const int32_t fo = sl.off[p], lr = sl.rem[p];
  if (fo > 0) o[0] = (o[0] & (7u << 29)) | (uint32_t)((int32_t)(o[0] & OP_LEN_MASK) - fo);
  if (lr < 0) o[n - 1] = (o[n - 1] & (7u << 29)) | (uint32_t)((int32_t)(o[n - 1] & OP_LEN_MASK) + lr);
}

// ---------------------------------------------------------------------------
// per-range statistics of one level's hits (verification / counting; not timed)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Per-range counts / checksums in two steps.  (1) A hit adds to its frontier RANGE's pair of words: the slots of a range
// are one run in every layout but the fused final level's (places filled entry by entry), a wave sums each run first, and
// whatever is left lands on as many addresses as the level has ranges.  (2) The ranges' words are summed per query over
// the frontier, which is sorted by query.  (Adding to the QUERY's words directly -- one pair of atomics per run of equal
// query -- was fine while a query's slots were one run; with the slots of a level's final pass in entry order every hit
// of a wave belongs to another range, and a window tiling at depth 5 -- 2 000 queries a chunk, 10^7 hits each -- spent
// 9 of its 10 s on those few thousand addresses.)
__global__ __launch_bounds__(256) void hit_stats_kernel(const FrontierRec *__restrict__ fr,
                                                        const uint32_t *__restrict__ pair_range, uint32_t n_pairs,
                                                        HitArrays h, int32_t min_output_length, int skip_same_target,
                                                        unsigned long long *__restrict__ rstat, int want_ck, uint32_t stride) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  uint32_t r = 0xFFFFFFFFu;
  unsigned long long c = 0, a = 0;
  if (p < n_pairs) {
    // (stride 2: a kept fused level's {query id, source} pairs -- pair_range then points at the second word of the first)
    r = pair_range[(size_t)p * stride];
    const uint32_t qid = h.qid[(size_t)p * stride];
    if (qid != HIT_NONE) {
      const uint32_t tgt = fr[r].target_id;
      const int4 hc = h.c[p];
      const int32_t qs = hc.x, qe = hc.y;
      const bool drop = (min_output_length >= 0 && abs(qe - qs) < min_output_length) ||
                        (skip_same_target && qid == tgt);  // multi_impg.rs:883-885
      if (!drop) {
        if (want_ck) {
          a = mix64(((unsigned long long)qid << 32) | (uint32_t)qs);
          a = mix64(a ^ (((unsigned long long)(uint32_t)qe << 32) | tgt));
          a = mix64(a ^ (((unsigned long long)(uint32_t)hc.z << 32) | (uint32_t)hc.w));
        }
        c = 1;
      }
    }
  }
  // a segmented scan over the wave's runs of equal range, one pair of atomics per run
  const unsigned lane = lane_id();
  const uint32_t prev = (uint32_t)__shfl_up((int)r, 1);
  const unsigned long long heads = __ballot(lane == 0 || prev != r);
  if (__popcll(heads) < 64) {
    const unsigned first = 63u - (unsigned)__clzll((long long)(heads & (lanemask_lt() | (1ull << lane))));
#pragma unroll
    for (unsigned o = 1; o < 64; o <<= 1) {
      const unsigned long long cv = (unsigned long long)__shfl_up((long long)c, o);
      const unsigned long long av = (unsigned long long)__shfl_up((long long)a, o);
      if (lane >= first + o) { c += cv; a += av; }
    }
  }
  const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
  if (tail && c && r != 0xFFFFFFFFu) {
    atomicAdd(&rstat[2u * (size_t)r], c);
    if (want_ck) atomicAdd(&rstat[2u * (size_t)r + 1u], a);
  }
}
__global__ __launch_bounds__(256) void range_stats_reduce_kernel(const FrontierRec *__restrict__ fr, uint32_t n_fr,
                                                                 const unsigned long long *__restrict__ rstat,
                                                                 unsigned long long *__restrict__ count,
                                                                 unsigned long long *__restrict__ cksum) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  uint32_t q = 0xFFFFFFFFu;
  unsigned long long c = 0, a = 0;
  if (i < n_fr) {
    q = fr[i].qidx;
    const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(rstat)[i];
    c = x.x; a = x.y;
  }
  const unsigned lane = lane_id();
  const uint32_t prev = (uint32_t)__shfl_up((int)q, 1);
  const unsigned long long heads = __ballot(lane == 0 || prev != q);
  const unsigned first = 63u - (unsigned)__clzll((long long)(heads & (lanemask_lt() | (1ull << lane))));
#pragma unroll
  for (unsigned o = 1; o < 64; o <<= 1) {
    const unsigned long long cv = (unsigned long long)__shfl_up((long long)c, o);
    const unsigned long long av = (unsigned long long)__shfl_up((long long)a, o);
    if (lane >= first + o) { c += cv; a += av; }
  }
  const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
  if (tail && q != 0xFFFFFFFFu && (c | a)) {
    if (count && c) atomicAdd(&count[q], c);
    if (cksum && a) atomicAdd(&cksum[q], a);
  }
}

// ---------------------------------------------------------------------------
// K3: visited-set update (impg.rs:2471-2560) and next frontier (impg.rs:2566-2584)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_keys_kernel(const FrontierRec *__restrict__ fr,
                                                          const uint32_t *__restrict__ pair_range, uint32_t n_pairs,
                                                          HitArrays h, unsigned long long *__restrict__ keys,
                                                          unsigned long long *__restrict__ vals,
                                                          unsigned long long *__restrict__ n_active) {
  // the sort's payload is the hit's normalised query interval itself (start << 32 | end), all the update
  // reads of a hit: carried as a slot index, every group paid a random 16-byte gather for it (60 % of
  // visited_update).  A stable sort keeps equal keys in slot order = the reference's processing order.
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  unsigned long long k = ~0ull;
  if (p < n_pairs) {
    const uint32_t qid = h.qid[p];
    unsigned long long v = 0;
    if (qid != HIT_NONE) {
      const FrontierRec f = fr[pair_range[p]];
      if (qid != f.target_id) {  // impg.rs:2507
        k = ((unsigned long long)f.qidx << 32) | qid;
        const int4 hc = h.c[p];
        v = ((unsigned long long)(uint32_t)min(hc.x, hc.y) << 32) | (uint32_t)max(hc.x, hc.y);
      }
    }
    keys[p] = k;
    vals[p] = v;
  }
  __shared__ uint32_t wcnt[4];
  unsigned long long m = __ballot(k != ~0ull);
  if (lane_id() == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (tot) atomicAdd(&n_active[(blockIdx.x % COUNT_SLOTS) * COUNT_STRIDE], (unsigned long long)tot);
  }
}

// head flag of each run of equal keys (only keys != ~0 count)
__global__ __launch_bounds__(256) void group_heads_kernel(const unsigned long long *__restrict__ skeys, uint32_t n,
                                                          uint32_t *__restrict__ head) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = skeys[i];
  head[i] = (k != ~0ull) && (i == 0 || skeys[i - 1] != k);
}
__global__ __launch_bounds__(256) void group_scatter_kernel(const unsigned long long *__restrict__ skeys, uint32_t n,
                                                            const uint32_t *__restrict__ head,
                                                            const uint32_t *__restrict__ gid,
                                                            uint32_t *__restrict__ gstart,
                                                            unsigned long long *__restrict__ gkey) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  if (head[i]) {
    gstart[gid[i]] = i;
    gkey[gid[i]] = skeys[i];
  }
}

// The same in two passes over the sorted keys and nothing else (round 4): heads counted per tile, the tiles' counts
// scanned, then every head writes its group's start and key at (tile base + its rank in the tile).  (head flags, a scan
// of them and a scatter were 44 bytes of traffic per hit; this is 16.)  A thread owns GROUP_ITEMS consecutive keys.
constexpr uint32_t GROUP_ITEMS = 8, GROUP_TILE = 256u * GROUP_ITEMS;
__global__ __launch_bounds__(256) void group_count_kernel(const unsigned long long *__restrict__ skeys, uint32_t n, uint32_t *__restrict__ tile_heads) {
  const uint32_t base = blockIdx.x * GROUP_TILE + threadIdx.x * GROUP_ITEMS;
  unsigned long long prev = base > 0 && base <= n ? skeys[base - 1u] : 0ull;
  uint32_t c = 0;
#pragma unroll
  for (uint32_t k = 0; k < GROUP_ITEMS; k++) {
    const uint32_t i = base + k;
    const unsigned long long key = i < n ? skeys[i] : ~0ull;
    c += (key != ~0ull && (i == 0 || key != prev)) ? 1u : 0u;
    prev = key;
  }
  uint32_t tot;
  block_excl_scan(c, &tot);
  if (threadIdx.x == 0) tile_heads[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void group_fill_kernel(const unsigned long long *__restrict__ skeys, uint32_t n, const uint32_t *__restrict__ tile_base,
                                                         uint32_t *__restrict__ gstart, unsigned long long *__restrict__ gkey) {
  const uint32_t base = blockIdx.x * GROUP_TILE + threadIdx.x * GROUP_ITEMS;
  unsigned long long key[GROUP_ITEMS];
  unsigned long long prev = base > 0 && base <= n ? skeys[base - 1u] : 0ull;
  uint32_t c = 0, flags = 0;
#pragma unroll
  for (uint32_t k = 0; k < GROUP_ITEMS; k++) {
    const uint32_t i = base + k;
    key[k] = i < n ? skeys[i] : ~0ull;
    if (key[k] != ~0ull && (i == 0 || key[k] != prev)) { flags |= 1u << k; c += 1u; }
    prev = key[k];
  }
  uint32_t tot;
  uint32_t g = tile_base[blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
  for (uint32_t k = 0; k < GROUP_ITEMS; k++)
    if (flags & (1u << k)) { gstart[g] = base + k; gkey[g] = key[k]; g += 1u; }
}

// ---------------------------------------------------------------------------
// The update's hits grouped by (query, sequence) WITHOUT a global sort (round 4).  What the update needs is the stable
// order by (qidx, hit sequence) of the hits that carry a key.  The slots of one frontier range are one run in every
// layout an updated level can have; restricted to one query, slot order is frontier order x visit order; and the
// frontier is sorted by query.  So "stable by query" is the query's ranges taken in frontier order, run by run, and what
// is left is a stable counting sort by sequence id INSIDE a query -- a wave per query, the sequence counters in LDS,
// ranks within a 64-hit chunk from ballots (slot order kept).  Per hit: 4 bytes read to count (seg_group_kernel<true>),
// 4 + 20 read and 16 written to place (…<false>), against update_keys' 40 + the radix sort's four passes and two
// histograms over 16-byte records (144).  Not for MultiImpg batches, the covered-hit filter, more than SEG_MAX_SEQ
// sequences, or a frontier that is not sorted by query (checked on the device): those keep the library sort.
// ---------------------------------------------------------------------------
constexpr uint32_t SEG_MAX_SEQ = 2048, SEG_WAVES = 4;
// (a query is one wave's work, hit after hit: beyond this many hits in one query -- a saturating closure's deep levels have
// 10^5-10^6 -- the library sort's parallelism wins: config 5's update 0.66 s per 4 000 windows sorted, 2.65 s by segments)
constexpr uint32_t SEG_BIG_QUERY = 32768;
// Past that a query is cut into `parts` slices of its frontier ranges, a wave each (round 6): the count pass leaves every
// slice's sequence counters in qbins, seg_parts_scan_kernel turns them into the slice's offset inside each of the query's
// sequences (an exclusive scan over the slices: slice order = frontier order, so the order stays the stable one) and adds
// the query's totals up; the place pass starts every sequence's counter at (the sequence's offset in the query) + (the
// slice's offset in the sequence).  A slice that still holds more than SEG_BIG_SLICE hits sends the level to the library.
constexpr uint32_t SEG_SLICE_HITS = 8192, SEG_BIG_SLICE = 131072, SEG_MAX_PARTS = 4096;
__global__ __launch_bounds__(256) void run_bounds_kernel(const uint32_t *__restrict__ pair_range, uint32_t n, uint32_t *__restrict__ run_start,
                                                         uint32_t *__restrict__ run_end) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n) return;
  const uint32_t r = pair_range[p];
  if (p == 0 || pair_range[p - 1u] != r) run_start[r] = p;
  if (p + 1u == n || pair_range[p + 1u] != r) run_end[r] = p + 1u;
}
// ... and where the lookup left the ranges' offsets and counts (Engine::expand on this device), straight from those: a pass
// over the frontier instead of one over every slot (0.31 ms of a headline step).  perm: the lookup order the offsets are
// listed in (slot = place in that order), or null for offsets by range.
__global__ __launch_bounds__(256) void run_bounds_by_offsets_kernel(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ pair_off,
                                                                    const uint32_t *__restrict__ cnt, uint32_t n_fr,
                                                                    uint32_t *__restrict__ run_start, uint32_t *__restrict__ run_end) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_fr) return;
  const uint32_t r = perm ? perm[i] : i, a = pair_off[i];
  run_start[r] = a;
  run_end[r] = a + cnt[i];
}
__global__ __launch_bounds__(256) void query_bounds_kernel(const FrontierRec *__restrict__ fr, uint32_t n_fr, uint32_t n_queries, uint32_t *__restrict__ qfirst,
                                                           uint32_t *__restrict__ qlast, uint32_t *__restrict__ unsorted) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_fr) return;
  const uint32_t q = fr[i].qidx;
  if (q >= n_queries) { *unsorted = 1u; return; }
  if (i == 0 || fr[i - 1u].qidx != q) {
    qfirst[q] = i;
    if (i && fr[i - 1u].qidx > q) *unsorted = 1u;
  }
  if (i + 1u == n_fr || fr[i + 1u].qidx != q) qlast[q] = i + 1u;
}
// The slots of the ranges [f0, f1) of one query, 64 at a time in frontier order x slot order (one wave): f(qid, hc, active)
// per chunk, the next chunk's loads already under way (a chunk is one or two memory round trips, and a query's ~20 runs
// of ~40 slots were that many trips in a row).  WITH_HC: the hit's coordinates are loaded beside its sequence id.
template <bool WITH_HC, class F>
__device__ __forceinline__ void seg_for_chunks(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ run_start,
                                               const uint32_t *__restrict__ run_end, const HitArrays &h, uint32_t f0, uint32_t f1, F f) {
  const uint32_t lane = lane_id();
  for (uint32_t base = f0; base < f1; base += 64u) {
    const uint32_t i = base + lane;
    const bool have = i < f1;
    const uint32_t rs = have ? run_start[i] : 0u, re = have ? run_end[i] : 0u, tg = have ? fr[i].target_id : 0u;
    const uint32_t m = min(64u, f1 - base);
    uint32_t j = 0, c0 = 0, e = 0, t = 0;
    auto seek = [&](uint32_t &jj, uint32_t &cc, uint32_t &ee, uint32_t &tt) {  // the first chunk at or after (jj, cc) that holds slots
      while (jj < m && cc >= ee) {
        jj += 1u;
        if (jj < m) {
          cc = (uint32_t)__builtin_amdgcn_readlane((int)rs, (int)jj);
          ee = (uint32_t)__builtin_amdgcn_readlane((int)re, (int)jj);
          tt = (uint32_t)__builtin_amdgcn_readlane((int)tg, (int)jj);
        }
      }
    };
    c0 = (uint32_t)__builtin_amdgcn_readlane((int)rs, 0);
    e = (uint32_t)__builtin_amdgcn_readlane((int)re, 0);
    t = (uint32_t)__builtin_amdgcn_readlane((int)tg, 0);
    seek(j, c0, e, t);
    uint32_t qid = HIT_NONE;
    int4 hc = make_int4(0, 0, 0, 0);
    if (j < m && c0 + lane < e) { qid = h.qid[c0 + lane]; if (WITH_HC) hc = h.c[c0 + lane]; }
    while (j < m) {
      uint32_t nj = j, nc = c0 + 64u, ne = e, nt = t;
      seek(nj, nc, ne, nt);
      uint32_t nqid = HIT_NONE;
      int4 nhc = make_int4(0, 0, 0, 0);
      if (nj < m && nc + lane < ne) { nqid = h.qid[nc + lane]; if (WITH_HC) nhc = h.c[nc + lane]; }
      f(qid, hc, qid != HIT_NONE && qid != t);  // impg.rs:2507
      j = nj; c0 = nc; e = ne; t = nt; qid = nqid; hc = nhc;
    }
  }
}
template <bool COUNT_ONLY>
__global__ __launch_bounds__(64 * SEG_WAVES) void seg_group_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ qfirst,
                                                                   const uint32_t *__restrict__ qlast, const uint32_t *__restrict__ run_start,
                                                                   const uint32_t *__restrict__ run_end, HitArrays h, uint32_t n_queries, uint32_t nb,
                                                                   uint32_t nbits, uint32_t *__restrict__ qact, const uint32_t *__restrict__ qdst,
                                                                   uint32_t *__restrict__ qgrp, const uint32_t *__restrict__ gdst,
                                                                   uint32_t *__restrict__ gstart, unsigned long long *__restrict__ gkey,
                                                                   unsigned long long *__restrict__ svals, uint32_t *__restrict__ qbins,
                                                                   uint32_t parts, uint32_t *__restrict__ qtot) {
  extern __shared__ __attribute__((aligned(16))) uint32_t seg_bins[];  // SEG_WAVES x nb sequence counters
  const uint32_t w = threadIdx.x >> 6, lane = lane_id();
  const uint32_t u = blockIdx.x * SEG_WAVES + w;  // the unit: a query, or one of the `parts` slices of a query's frontier ranges
  if (u >= n_queries * parts) return;  // (no barrier below: the waves of a block share nothing)
  const bool sliced = parts > 1u;
  const uint32_t q = sliced ? u / parts : u, part = u - q * parts;
  uint32_t f0 = qfirst[q], f1 = qlast[q];
  if (sliced) {
    const uint32_t len = f1 > f0 ? f1 - f0 : 0u;
    f1 = f0 + (uint32_t)(((unsigned long long)len * (part + 1u)) / parts);
    f0 = f0 + (uint32_t)(((unsigned long long)len * part) / parts);
  } else if (f0 >= f1) { if (COUNT_ONLY && lane == 0) { qact[q] = 0u; qgrp[q] = 0u; } return; }
  uint32_t *bins = seg_bins + w * nb;
  // pass A: the unit's hits per sequence (the place pass reads the count pass's counters back when they were kept: nb
  // words a unit, instead of walking its scattered runs once more)
  if (COUNT_ONLY || !qbins) {
    for (uint32_t b = lane; b < nb; b += 64u) bins[b] = 0u;
    __builtin_amdgcn_wave_barrier();
    seg_for_chunks<false>(fr, run_start, run_end, h, f0, f1, [&](uint32_t qid, const int4 &, bool active) { if (active) atomicAdd(&bins[qid], 1u); });
  } else if (!sliced) {
    for (uint32_t b = lane; b < nb; b += 64u) bins[b] = qbins[(size_t)q * nb + b];
  }
  __builtin_amdgcn_wave_barrier();
  if (COUNT_ONLY) {  // how many of the query's hits carry a key, and how many sequences they name (its groups)
    uint32_t cnt = 0, grp = 0;
    for (uint32_t b = lane; b < nb; b += 64u) {
      const uint32_t x = bins[b];
      cnt += x; grp += x ? 1u : 0u;
      if (qbins) qbins[(size_t)u * nb + b] = x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += (uint32_t)__shfl_xor((int)cnt, o); grp += (uint32_t)__shfl_xor((int)grp, o); }
    if (lane == 0) {
      if (!sliced) { qact[q] = cnt; qgrp[q] = grp; }  // (sliced: seg_parts_scan_kernel adds the slices up)
      // (the word behind `unsorted`: Engine::update lays the six per-query arrays out)
      if (cnt > (sliced ? SEG_BIG_SLICE : SEG_BIG_QUERY)) atomicMax(qact + 4u * (size_t)n_queries + 1u, cnt);
    }
    return;
  }
  // the sequences' offsets inside the query's stretch of the output; a sequence with hits is a group: its start, its key
  // (sliced: the query's totals from qtot, the slice's own offset inside each sequence from qbins; slice 0 names the groups)
  const uint32_t dst0 = qdst[q];
  const unsigned long long khi = (unsigned long long)q << 32;
  uint32_t carry = 0, g = gdst[q];
  for (uint32_t b = 0; b < nb; b += 64u) {
    const uint32_t x = sliced ? qtot[(size_t)q * nb + b + lane] : bins[b + lane];
    const uint32_t inc = wave_incl_scan(x);
    const uint32_t off = carry + inc - x;
    bins[b + lane] = off + (sliced ? qbins[(size_t)u * nb + b + lane] : 0u);
    carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const unsigned long long gm = __ballot(x != 0u);
    if (x && part == 0u) {
      const uint32_t gi = g + (uint32_t)__popcll(gm & lanemask_lt());
      gstart[gi] = dst0 + off;
      gkey[gi] = khi | (b + lane);
    }
    g += (uint32_t)__popcll(gm);
  }
  __builtin_amdgcn_wave_barrier();
  // pass B: a hit's place = its sequence's counter + its rank among the chunk's earlier hits of that sequence
  seg_for_chunks<true>(fr, run_start, run_end, h, f0, f1, [&](uint32_t qid, const int4 &hc, bool active) {
    const unsigned long long am = __ballot(active);
    if (!am) return;
    unsigned long long mask = am;  // the chunk's hits of this lane's sequence
    for (uint32_t b = 0; b < nbits; b++) {
      const bool bit = (qid >> b) & 1u;
      const unsigned long long bal = __ballot(active && bit);
      mask &= bit ? bal : ~bal;
    }
    if (active) {
      const uint32_t pos = dst0 + bins[qid] + (uint32_t)__popcll(mask & lanemask_lt());
      svals[pos] = ((unsigned long long)(uint32_t)min(hc.x, hc.y) << 32) | (uint32_t)max(hc.x, hc.y);
    }
    __builtin_amdgcn_wave_barrier();
    if (active && lane == 63u - (uint32_t)__clzll((long long)mask)) bins[qid] += (uint32_t)__popcll(mask);
    __builtin_amdgcn_wave_barrier();
  });
}

// (a thread per (query, sequence): the slices' counters to their exclusive prefix, in place; the total to qtot; the query's
// hits and groups added up -- qact / qgrp zeroed by the launcher)
__global__ __launch_bounds__(256) void seg_parts_scan_kernel(uint32_t *__restrict__ qbins, uint32_t nb, uint32_t parts, uint32_t *__restrict__ qtot,
                                                             uint32_t *__restrict__ qact, uint32_t *__restrict__ qgrp) {
  const uint32_t q = blockIdx.x, b = blockIdx.y * 256u + threadIdx.x;
  uint32_t run = 0;
  if (b < nb) {
    uint32_t *p = qbins + (size_t)q * parts * nb + b;
    uint32_t k = 0;
    for (; k + 4u <= parts; k += 4u) {
      const uint32_t x0 = p[(size_t)k * nb], x1 = p[(size_t)(k + 1u) * nb], x2 = p[(size_t)(k + 2u) * nb], x3 = p[(size_t)(k + 3u) * nb];
      p[(size_t)k * nb] = run; run += x0;
      p[(size_t)(k + 1u) * nb] = run; run += x1;
      p[(size_t)(k + 2u) * nb] = run; run += x2;
      p[(size_t)(k + 3u) * nb] = run; run += x3;
    }
    for (; k < parts; k++) { const uint32_t x = p[(size_t)k * nb]; p[(size_t)k * nb] = run; run += x; }
    qtot[(size_t)q * nb + b] = run;
  }
  uint32_t cnt = run, grp = run ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cnt += (uint32_t)__shfl_xor((int)cnt, o); grp += (uint32_t)__shfl_xor((int)grp, o); }
  if (lane_id() == 0 && cnt) { atomicAdd(&qact[q], cnt); atomicAdd(&qgrp[q], grp); }
}

__device__ __forceinline__ uint32_t lower_bound_u64(const unsigned long long *a, uint32_t n, unsigned long long k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void table_qoff_kernel(const unsigned long long *__restrict__ keys, uint32_t n_groups, uint32_t n_queries,
                                                         uint32_t *__restrict__ qoff) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q <= n_queries) qoff[q] = lower_bound_u64(keys, n_groups, (unsigned long long)q << 32);
}
// per group: run length, the group's current visited list (newest table that
// holds the key; its length is cap - glen), capacities for the new list and for the pieces.
// The list is handed on as a POINTER: the update kernels used to resolve (table, index) themselves -- a dynamic index
// into the by-value table array, then the table's offset and length, then the list: three dependent round trips
// before the first range, per group.
__global__ __launch_bounds__(256) void group_prepare_kernel(VisitedTables vt, const unsigned long long *__restrict__ gkey,
                                                            const uint32_t *__restrict__ gstart, uint32_t n_groups,
                                                            uint32_t n_active, uint32_t *__restrict__ glen,
                                                            const int2 **__restrict__ old_src,
                                                            uint32_t *__restrict__ cap, uint32_t *__restrict__ pcap) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t st = gstart[g];
  const uint32_t en = g + 1 < n_groups ? gstart[g + 1] : n_active;
  const uint32_t len = en - st;
  glen[g] = len;
  const unsigned long long k = gkey[g];
  const int2 *src = nullptr;
  uint32_t olen = 0;
  bool found = false;
  const uint32_t q = (uint32_t)(k >> 32);
  for (int t = (int)vt.n_tables - 1; t >= 0; t--) {
    const VisitedTable &T = vt.t[t];
    // (among the query's own keys: a handful of steps instead of log2 of the whole table -- 22 dependent reads on a
    // headline level's 2.8 x 10^6-key table)
    const uint32_t lo = T.qoff[q], hi = T.qoff[q + 1u];
    const uint32_t i = lo + lower_bound_u64(T.keys + lo, hi - lo, k);
    if (i < hi && T.keys[i] == k) {
      src = T.ranges + T.off[i];
      olen = T.len[i];
      found = true;
      break;
    }
  }
  if (!found && vt.mask_off) {  // first touch under a mask: the clone of the mask's list (impg.rs:2077-2078)
    const uint32_t seq = (uint32_t)(k & 0xFFFFFFFFull);
    const uint32_t a = vt.mask_off[seq];
    olen = vt.mask_off[seq + 1] - a;
    src = vt.mask_ranges + a;
  }
  old_src[g] = olen ? src : nullptr;
  cap[g] = olen + len;
  pcap[g] = olen + 2u * len;
}

__device__ __forceinline__ uint32_t lower_bound_start(const int2 *a, uint32_t n, int32_t s) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid].x < s) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// (Round 4, measured and dropped: a block of 256 groups handing them to its lanes sorted by their number of hits, so that
// a wave's replays are equally long -- update 12.7 -> 16.1 ms.  Neighbouring groups own neighbouring hits, lists and
// slices; permuted, a wave's every read and write touches 64 lines instead of ~17: the kernel is bound by the lines
// its memory instructions touch, not by the longest replay of a wave.  Nor does it pay to fetch the wave's stretch of the
// sorted hits into LDS first (coalesced, 4 KB per wave): 12 KB of LDS per wave instead of 8 leave 13 waves per CU instead
// of 20, update 12.6 -> 14.3 ms; with a 3 KB buffer 13.7.)
// one thread per (query, sequence) group: replay the group's hits in emission
// order against its visited list (SortedRanges::insert with min_distance = 0,
// impg.rs:270-368), collect the new pieces.
// The list is searched three times and shifted once per hit, every step a dependent access: lists of at most
// VU_LDS_CAP ranges (nearly all of them) are kept in LDS while they are worked on (lane-interleaved, so the 64
// lists of a wave never share a bank) and written out once at the end; longer ones are worked on in place.
#ifndef IMPG_VU_LDS_CAP
#define IMPG_VU_LDS_CAP 12
#endif
constexpr uint32_t VU_LDS_CAP = IMPG_VU_LDS_CAP;
// IMPG_VW_WINDOWS = 1: every hit of a batch that meets no earlier one takes its turn at once (replay_hits_wave).  Exact -- the
// suite and the config-5 tiling test are green with it -- and measured SLOWER: config 5, 4 000 windows, update 629 -> 1 045 ms.
// Two passes over a ~1 000-range list per batch and 5 KB more LDS a wave (10 waves a CU instead of 17) cost more than the ~25
// hits a batch it frees from the sequential part save.  0 (the default): round 3's two classes.
#ifndef IMPG_VW_WINDOWS
#define IMPG_VW_WINDOWS 0
#endif
#ifndef IMPG_VW_WINDOW
#define IMPG_VW_WINDOW 4   // ranges a hit may start in / swallow and still take its turn on a private copy
#endif
[[maybe_unused]] constexpr uint32_t VW_WINDOW = IMPG_VW_WINDOW;
// groups whose list can outgrow this get a whole wave (visited_update_wave_kernel below)
// two sizes of LDS working set (9 KB: 17 waves per CU; 32 KB: 5): groups with few hits and a short list take the small one
#ifndef IMPG_VW_TINY
#define IMPG_VW_TINY 192  // 1.5 KB: groups whose list cannot outgrow 128 ranges (cap = old + hits <= 128), every wave slot of a CU holds one
#endif
#ifndef IMPG_VW_SMALL
#define IMPG_VW_SMALL 1152  // 9 KB, 17 waves per CU.  With the isolated hits of a batch going in together (replay_hits_wave), config 5
#endif                      // (4 000 windows, update ms): 1 024 entries 997 -- 85 % of the deepest level's groups end at ~1 030 ranges and
                            // finish their replay in global memory at four times the cost per hit --, 1 152: 924, 1 280: 1 012, 1 536: 1 103
// (the tiny tier of round 3 -- short lists of deep groups, 512 / 768 entries -- lost to one 1 152-entry tier on config 5: more
// waves, but lists that outgrew the buffer replayed in global memory.  Round 5's tiny tier is for groups that CANNOT outgrow it.)
// Which kernel takes a group, by cap = old length + hits (what its list cannot outgrow):
//   cap <= VU_TINY_MAX  visited_update_kernel<VU_LDS_CAP, false>: a lane per group, 64 neighbouring groups a wave (98.5 % of a
//                       headline level's groups; list AND pieces in the lane's 12-entry LDS column)
//   cap <= VU_MID_MAX   visited_update_kernel<VU_MID_CAP, true>: a lane per group too, but of a LIST of such groups and with a
//                       64-entry column -- among tiny groups one of them kept 63 lanes waiting for its 30 hits (and, with
//                       17..64 hits, replayed in place on its global slice: two thirds of all wave cycles, round 4)
//   beyond              visited_update_wave_kernel: a wave per group (three LDS sizes)
#ifndef IMPG_VU_TINY_MAX
#define IMPG_VU_TINY_MAX 12
#endif
#ifndef IMPG_VU_MID_MAX
#define IMPG_VU_MID_MAX 48
#endif
#ifndef IMPG_VU_MID_CAP
#define IMPG_VU_MID_CAP 64
#endif
constexpr uint32_t VU_TINY_MAX = IMPG_VU_TINY_MAX, VU_MID_MAX = IMPG_VU_MID_MAX, VU_MID_CAP = IMPG_VU_MID_CAP;
static_assert(VU_TINY_MAX <= VU_LDS_CAP && VU_MID_MAX <= VU_MID_CAP, "a lane's list must fit its LDS column");
constexpr uint32_t VW_CAP_TINY = IMPG_VW_TINY, VW_CAP_SMALL = IMPG_VW_SMALL, VW_CAP_LARGE = 4096,
                   VW_SMALL_HEADROOM = 256;
struct ListInPlace {  // the group's slice of the new table
  int2 *p;
  __device__ __forceinline__ int32_t &x(uint32_t i) const { return p[i].x; }
  __device__ __forceinline__ int32_t &y(uint32_t i) const { return p[i].y; }
};
struct ListInLds {  // element i of this lane's list at [i * 64 + lane]
  int32_t *bx, *by;
  __device__ __forceinline__ int32_t &x(uint32_t i) const { return bx[i * 64u]; }
  __device__ __forceinline__ int32_t &y(uint32_t i) const { return by[i * 64u]; }
};
template <class L> __device__ __forceinline__ uint32_t list_lower_bound(const L &R, uint32_t n, int32_t s) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (R.x(mid) < s) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// one hit's turn against the list R (len entries): the proximity test (impg.rs:2513-2545), then SortedRanges::insert
// with min_distance = 0 (impg.rs:270-368); the not-yet-visited pieces that are long enough go to emit(start, end)
// (grow() is called right before the list gets one range longer -- R[len] is about to be written)
template <class L, class E, class G>
__device__ __forceinline__ void replay_one(const L &R, uint32_t &len, int32_t start, int32_t end, int32_t sequence_length,
                                           int32_t min_transitive_len, int32_t mdbr, E &&emit, G &&grow) {
  bool should_add = true;
  if (mdbr > 0) {  // impg.rs:2513-2545
    const uint32_t idx = list_lower_bound(R, len, start);
    if (idx > 0 && abs(start - R.y(idx - 1)) < mdbr) should_add = false;
    if (should_add && idx < len && abs(R.x(idx) - end) < mdbr) should_add = false;
  }
  if (!should_add) return;
  // ---- SortedRanges::insert, min_distance = 0 --------------------------------
  if (start < 0) start = 0;                       // impg.rs:287-289
  if (end > sequence_length) end = sequence_length;  // impg.rs:294-296
  int32_t current = start;
  const uint32_t pos = list_lower_bound(R, len, start);  // (the same lower bound serves :303 and :330)
  uint32_t i = pos;
  if (i > 0 && R.y(i - 1) > start) i -= 1;
  while (i < len && current < end) {  // impg.rs:314-324
    const int32_t rx = R.x(i), ry = R.y(i);
    if (rx > end) break;
    if (current < rx) {
      if (abs(rx - current) >= min_transitive_len) emit(current, rx);
    }
    current = max(current, ry);
    i += 1;
  }
  if (current < end) {
    if (abs(end - current) >= min_transitive_len) emit(current, end);
  }
  uint32_t mfrom;  // impg.rs:330-343
  if (pos > 0 && R.y(pos - 1) >= start) {
    R.y(pos - 1) = max(R.y(pos - 1), end);
    mfrom = pos - 1;
  } else if (pos < len && end >= R.x(pos)) {
    R.x(pos) = min(start, R.x(pos));
    R.y(pos) = max(end, R.y(pos));
    mfrom = pos;
  } else {
    grow();
    for (uint32_t k = len; k > pos; k--) { R.x(k) = R.x(k - 1); R.y(k) = R.y(k - 1); }
    R.x(pos) = start;
    R.y(pos) = end;
    len += 1;
    return;
  }
  uint32_t write = mfrom, read = mfrom + 1;  // merge_forward_from, impg.rs:355-368
  while (read < len) {
    if (R.y(write) >= R.x(read)) {
      R.y(write) = max(R.y(write), R.y(read));
    } else {
      write += 1;
      const int32_t tx = R.x(write), ty = R.y(write);
      R.x(write) = R.x(read); R.y(write) = R.y(read);
      R.x(read) = tx; R.y(read) = ty;
    }
    read += 1;
  }
  len = write + 1;
}
// replays hits [st, st+n) against the list R (len entries on entry); returns the new length, appends pieces to P
template <class L>
__device__ __forceinline__ uint32_t replay_hits(const L &R, uint32_t len, const unsigned long long *__restrict__ svals,
                                                uint32_t st, uint32_t n, int32_t sequence_length,
                                                int32_t min_transitive_len, int32_t mdbr, int2 *__restrict__ P, uint32_t &np) {
  for (uint32_t t = 0; t < n; t++) {
    const unsigned long long iv = svals[st + t];  // (min, max) of the hit's query interval, packed by update_keys
    replay_one(R, len, (int32_t)(uint32_t)(iv >> 32), (int32_t)(uint32_t)iv, sequence_length, min_transitive_len, mdbr,
               [&](int32_t a, int32_t b) { P[np++] = make_int2(a, b); }, [] {});
  }
  return len;
}
// Round 5: what the kernel waited for (scripts/vu_clocks.py, -DIMPG_VU_CLOCKS: the cycle counter at the phase boundaries).
// (i) Two thirds of all wave cycles went to the 0.5 % of groups with 17..64 hits, which the lane kernel replayed IN PLACE on
// their global slice -- three searches and a shift per hit, every step a memory round trip -- while the 63 other lanes
// of the wave waited: a third of the waves held one.  They now go to a wave of their own (visited_update_wave_kernel's
// 192-entry tier, VW_MIN = the LDS column).  (ii) Every variable-trip loop -- the old list copied into LDS, the hits
// replayed, the pieces fetched back for their sort -- paid one memory round trip PER ITERATION (load, s_waitcnt, use,
// next), and a wave runs as many iterations as its longest lane.  Now: the group's list arrives as a pointer
// (group_prepare), the first four hits and the first four ranges of the old list are requested together right behind
// the group's record, the hits come four at a time with the next four under way while four are replayed, and a
// group's pieces collect at the TOP of the lane's LDS column, growing down towards the list (they move to the group's
// global slice only if the two meet): sorted and merged where they are, stored once -- nothing is read back.
// -DIMPG_VU_CLOCKS (experiments, scripts/vu_clocks.py): the cycle counter at the kernel's phase boundaries (everything
// outstanding waited for first), summed over all waves; read and cleared by impg_gpu_debug_vu_clocks
#ifdef IMPG_VU_CLOCKS
constexpr uint32_t VU_CLK_ROWS = 1024;
__device__ unsigned long long g_vu_clk[VU_CLK_ROWS][16];  // (striped: 14 atomics a wave on ONE row serialised the whole kernel)
#define VU_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); vu_t[i] = __builtin_readcyclecounter(); } while (0)
#else
#define VU_MARK(i) do { } while (0)
#endif
// a visited list through a pointer that was read from memory: eight bytes a range, as a GLOBAL address (through a generic
// pointer the loads are flat_load: the slower path, and a wait on two counters)
typedef const unsigned long long __attribute__((address_space(1))) *global_list_t;
__device__ __forceinline__ int2 list_range(global_list_t p, uint32_t i) {
  const unsigned long long v = p[i];
  return make_int2((int32_t)(uint32_t)v, (int32_t)(uint32_t)(v >> 32));
}
// Waves per SIMD the dense form's register allocation is held to.  Left alone the allocator takes 96 registers -- the 8 KB
// LDS column of a 16-entry cap allowed five waves whatever it did --; with a 12-entry column (6 KB: 26 waves a CU) and
// held to six waves it needs 59 and spills nothing: headline update 6.05 -> 5.80 ms (round 5; 7 waves / 11 entries and
// 8 / 10 send more groups to the listed form than the waves win back: 6.08, 6.40).
#ifndef IMPG_VU_WAVES
#define IMPG_VU_WAVES 6
#endif
#if IMPG_VU_WAVES > 0
#define VU_OCCUPANCY __attribute__((amdgpu_waves_per_eu(IMPG_VU_WAVES, IMPG_VU_WAVES)))
#else
#define VU_OCCUPANCY
#endif
// (the listed form's 32 KB column allows two waves whatever the allocator does: the target is the dense form's)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <uint32_t CAP, bool LISTED>  // CAP: entries of a lane's LDS column; LISTED: the groups of list[0 .. *n_list), grid-strided
__global__ __launch_bounds__(64) VU_OCCUPANCY void visited_update_kernel(const unsigned long long *__restrict__ svals,
                                                            const int32_t *__restrict__ seq_len,
                                                            const unsigned long long *__restrict__ gkey,
                                                            const uint32_t *__restrict__ gstart,
                                                            const uint32_t *__restrict__ glen,
                                                            const unsigned long long *__restrict__ old_src,
                                                            const uint32_t *__restrict__ cap,
                                                            const uint32_t *__restrict__ noff,
                                                            const uint32_t *__restrict__ poff, uint32_t n_groups,
                                                            int32_t min_transitive_len, int32_t mdbr,
                                                            int2 *__restrict__ new_ranges, uint32_t *__restrict__ new_len,
                                                            int2 *__restrict__ pieces, uint32_t *__restrict__ n_pieces,
                                                            const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list) {
  __shared__ int32_t lds_x[CAP * 64], lds_y[CAP * 64];
  const uint32_t n_items = LISTED ? *n_list : n_groups;
  for (uint32_t item0 = blockIdx.x * 64u; item0 < n_items; item0 += gridDim.x * 64u) {  // (one trip unless LISTED)
  const uint32_t item = item0 + threadIdx.x;
#ifdef IMPG_VU_CLOCKS
  unsigned long long vu_t[8];
  VU_MARK(0);
#endif
  // round 1 (coalesced): the group's record.  Lanes without a group of their own (the grid's tail; groups another
  // kernel takes) stay in the wave with nothing to do: the loops below are wave-uniform.
  const bool have = item < n_items;
  const uint32_t g = LISTED ? list[min(item, n_items - 1u)] : item;
  const uint32_t gi = min(g, n_groups - 1u);  // (unconditional loads: one round trip, not one per dependent condition)
  const uint32_t c_ = cap[gi], n_ = glen[gi], st_ = gstart[gi], no = noff[gi], po = poff[gi];
  // (the list pointer is read as an integer and made a GLOBAL pointer: through a generic one its loads were flat_load)
  const global_list_t src_ = (global_list_t)old_src[gi];
  const unsigned long long key = gkey[gi];
  const uint32_t c = have ? c_ : 0u;
  const bool mine = have && (LISTED || c <= VU_TINY_MAX);  // cap = old length + hits bounds the list at every step
  const uint32_t n = mine ? n_ : 0u, st = mine ? st_ : 0u;
  const uint32_t olen = mine ? c - n : 0u;
  int2 *R = new_ranges + no;
  int2 *P = pieces + po;
  VU_MARK(1);
  // round 2: the clamp length, the first hits and the first ranges of the old list, all under way together -- straight-
  // line code: a lane with nothing to fetch reads a harmless address (the compiler cannot count loads across branches
  // and waits for ALL of them at every join)
  const int32_t sequence_length = seq_len[(uint32_t)(key & 0xFFFFFFFFull)];  // visited_entry, impg.rs:2048-2053
  const unsigned long long *hp = svals + st;
  const uint32_t hl = max(n, 1u) - 1u;
  unsigned long long h0 = hp[0], h1 = hp[min(1u, hl)], h2 = hp[min(2u, hl)], h3 = hp[min(3u, hl)];
  const global_list_t src = olen ? src_ : (global_list_t)(unsigned long long)(uintptr_t)svals;
  const uint32_t sl = max(olen, 1u) - 1u;
  const int2 s0 = list_range(src, 0), s1 = list_range(src, min(1u, sl)), s2 = list_range(src, min(2u, sl)), s3 = list_range(src, min(3u, sl));
  VU_MARK(2);
  uint32_t len = olen, np = 0;
  VU_MARK(3);
  const bool fast = mine;
  const ListInLds A{lds_x + threadIdx.x, lds_y + threadIdx.x};
  if (fast) {
    if (olen > 0) { A.x(0) = s0.x; A.y(0) = s0.y; }
    if (olen > 1) { A.x(1) = s1.x; A.y(1) = s1.y; }
    if (olen > 2) { A.x(2) = s2.x; A.y(2) = s2.y; }
    if (olen > 3) { A.x(3) = s3.x; A.y(3) = s3.y; }
    for (uint32_t i = 4; i < olen; i += 4) {  // (four requests a round trip)
      const uint32_t l = olen - 1u;
      const int2 r0 = list_range(src, i), r1 = list_range(src, min(i + 1u, l)), r2 = list_range(src, min(i + 2u, l)), r3 = list_range(src, min(i + 3u, l));
      A.x(i) = r0.x; A.y(i) = r0.y;
      if (i + 1u < olen) { A.x(i + 1u) = r1.x; A.y(i + 1u) = r1.y; }
      if (i + 2u < olen) { A.x(i + 2u) = r2.x; A.y(i + 2u) = r2.y; }
      if (i + 3u < olen) { A.x(i + 3u) = r3.x; A.y(i + 3u) = r3.y; }
    }
  }
  VU_MARK(4);
  // the replay, four hits a batch: the next batch is requested before this one is worked on
  const uint32_t nf = fast ? n : 0u;
  const uint32_t nmax = wave_max_u32(nf);
  // pieces: the k-th at the column's entry CAP - 1 - k while list and pieces do not meet
  bool spilled = false;
  auto spill = [&]() {
    for (uint32_t k = 0; k < np; k++) P[k] = make_int2(A.x(CAP - 1u - k), A.y(CAP - 1u - k));
    spilled = true;
  };
  auto emit = [&](int32_t a, int32_t b) {
    if (!spilled && len + np >= CAP) spill();
    if (!spilled) { A.x(CAP - 1u - np) = a; A.y(CAP - 1u - np) = b; }
    else P[np] = make_int2(a, b);
    np += 1;
  };
  auto grow = [&]() { if (!spilled && len + np >= CAP) spill(); };
  for (uint32_t t0 = 0; t0 < nmax; t0 += 4u) {
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (t0 + 4u < nmax && nf) {  // (a lane past its last hit reads that hit again: never used)
      const unsigned long long *hp = svals + st;
      const uint32_t l = nf - 1u;
      q0 = hp[min(t0 + 4u, l)]; q1 = hp[min(t0 + 5u, l)]; q2 = hp[min(t0 + 6u, l)]; q3 = hp[min(t0 + 7u, l)];
    }
#pragma nounroll
    for (uint32_t k = 0; k < 4u; k++) {
      const unsigned long long iv = h0;
      h0 = h1; h1 = h2; h2 = h3;
      if (t0 + k < nf)
        replay_one(A, len, (int32_t)(uint32_t)(iv >> 32), (int32_t)(uint32_t)iv, sequence_length, min_transitive_len, mdbr, emit, grow);
    }
    h0 = q0; h1 = q1; h2 = q2; h3 = q3;
  }
  VU_MARK(5);
  if (fast) for (uint32_t i = 0; i < len; i++) R[i] = make_int2(A.x(i), A.y(i));
  VU_MARK(6);
  // next-depth ranges of this group: sort by start, merge overlapping/contiguous
  // (impg.rs:2568-2584; ranges of different groups never share (query, id))
  if (fast && !spilled) {
    // the usual case: the pieces are in the column (entry CAP - 1 - k): insertion sort by start, sweep, store
    const uint32_t T = CAP - 1u;
    for (uint32_t i = 1; i < np; i++) {
      const int32_t tx = A.x(T - i), ty = A.y(T - i);
      uint32_t j = i;
      while (j > 0 && A.x(T - (j - 1u)) > tx) { A.x(T - j) = A.x(T - (j - 1u)); A.y(T - j) = A.y(T - (j - 1u)); j--; }
      A.x(T - j) = tx; A.y(T - j) = ty;
    }
    if (np) {
      uint32_t w = 0;
      int32_t cx = A.x(T), cy = A.y(T);
      for (uint32_t r = 1; r < np; r++) {
        const int32_t rx = A.x(T - r), ry = A.y(T - r);
        if (cy >= rx) cy = max(cy, ry);
        else { P[w] = make_int2(cx, cy); w += 1; cx = rx; cy = ry; }
      }
      P[w] = make_int2(cx, cy);
      np = w + 1;
    }
  } else if (mine && np > 1 && np <= CAP) {
    // a handful of pieces: fetched once into the lane's LDS column (the list has been written out), sorted and merged
    // there -- on the global slice every comparison of the insertion sort was a dependent round trip to memory
    const ListInLds S{lds_x + threadIdx.x, lds_y + threadIdx.x};
    for (uint32_t i = 0; i < np; i += 4) {
      const uint32_t l = np - 1u;
      const int2 r0 = P[i], r1 = P[min(i + 1u, l)], r2 = P[min(i + 2u, l)], r3 = P[min(i + 3u, l)];
      S.x(i) = r0.x; S.y(i) = r0.y;
      if (i + 1u < np) { S.x(i + 1u) = r1.x; S.y(i + 1u) = r1.y; }
      if (i + 2u < np) { S.x(i + 2u) = r2.x; S.y(i + 2u) = r2.y; }
      if (i + 3u < np) { S.x(i + 3u) = r3.x; S.y(i + 3u) = r3.y; }
    }
    for (uint32_t i = 1; i < np; i++) {
      const int32_t tx = S.x(i), ty = S.y(i);
      uint32_t j = i;
      while (j > 0 && S.x(j - 1) > tx) { S.x(j) = S.x(j - 1); S.y(j) = S.y(j - 1); j--; }
      S.x(j) = tx; S.y(j) = ty;
    }
    uint32_t w = 0;
    int32_t cx = S.x(0), cy = S.y(0);
    for (uint32_t r = 1; r < np; r++) {
      const int32_t rx = S.x(r), ry = S.y(r);
      if (cy >= rx) cy = max(cy, ry);
      else { P[w] = make_int2(cx, cy); w += 1; cx = rx; cy = ry; }
    }
    P[w] = make_int2(cx, cy);
    np = w + 1;
  } else if (mine) {
    for (uint32_t i = 1; i < np; i++) {
      int2 x = P[i];
      uint32_t j = i;
      while (j > 0 && P[j - 1].x > x.x) { P[j] = P[j - 1]; j--; }
      P[j] = x;
    }
    uint32_t w = 0;
    for (uint32_t r = 1; r < np; r++) {
      if (P[w].y >= P[r].x) P[w].y = max(P[w].y, P[r].y);
      else { w += 1; P[w] = P[r]; }
    }
    if (np) np = w + 1;
  }
  if (mine) {
    new_len[g] = len;
    n_pieces[g] = np;
  }
#ifdef IMPG_VU_CLOCKS
  VU_MARK(7);
  {
    const unsigned long long slow = 0ull, regp = __ballot(fast && spilled);
    const uint32_t nsum = wave_incl_scan(n), smax = 0u;
    const uint32_t ntot = (uint32_t)__builtin_amdgcn_readlane((int)nsum, 63);
    if (threadIdx.x == 0 && (blockIdx.x & 15u) == 0u) {  // (every 16th wave)
      unsigned long long *row = g_vu_clk[(blockIdx.x >> 4) % VU_CLK_ROWS];
      for (int i = 0; i < 7; i++) atomicAdd(&row[i], vu_t[i + 1] - vu_t[i]);
      atomicAdd(&row[8], 1ull);
      atomicAdd(&row[9], slow ? 1ull : 0ull);
      atomicAdd(&row[10], (unsigned long long)nmax);
      atomicAdd(&row[11], (unsigned long long)ntot);
      atomicAdd(&row[12], (unsigned long long)smax);
      atomicAdd(&row[13], regp ? 1ull : 0ull);
      if (slow) atomicAdd(&row[14], vu_t[3] - vu_t[2]);
    }
  }
#endif
  }
}
#pragma clang diagnostic pop
#ifdef IMPG_VU_CLOCKS
extern "C" void impg_gpu_debug_vu_clocks(unsigned long long *out) {
  static unsigned long long rows[VU_CLK_ROWS][16];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_vu_clk), sizeof(rows));
  for (int k = 0; k < 16; k++) { out[k] = 0; for (uint32_t r = 0; r < VU_CLK_ROWS; r++) out[k] += rows[r][k]; }
  memset(rows, 0, sizeof(rows));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_vu_clk), rows, sizeof(rows));
}
#endif

// ---------------------------------------------------------------------------
// The same replay for groups with long lists or many hits (deep closures: at depth 5 of a window tiling a group
// replays thousands of hits against lists of hundreds of ranges): one WAVE per group.  The replay stays
// sequential hit by hit -- the reference's order dependence -- and every lane runs the same scalar logic on the
// same list words (LDS broadcasts); what a single lane did in O(list) per hit is spread over the 64 lanes: the
// lower bounds (64-ary narrowing with ballots), the shift that makes room for an insertion, the shift that
// closes the gap after a merge, and the final sort of the group's pieces (a bitonic network, O(n log^2 n / 64)
// where the lane kernel's insertion sort was O(n^2)).  List and pieces live in LDS while they fit
// (VW_LIST_CAP / VW_PIECE_CAP), else the same code runs on the group's slices in global memory.
// ---------------------------------------------------------------------------
// Lanes of the one wave read words other lanes wrote a few instructions earlier.  For a list in LDS the hardware
// keeps one wave's ds_ instructions in program order, so the only thing to stop is the compiler moving them
// (wave_barrier emits no instruction); a list worked on in global memory gets the real barrier (waitcnt + s_barrier).
template <class L> __device__ __forceinline__ void order_point(const L &) { __syncthreads(); }
struct ListSoA {  // x[i], y[i] in two LDS arrays
  int32_t *px, *py;
  __device__ __forceinline__ int32_t &x(uint32_t i) const { return px[i]; }
  __device__ __forceinline__ int32_t &y(uint32_t i) const { return py[i]; }
};
template <> __device__ __forceinline__ void order_point<ListSoA>(const ListSoA &) { __builtin_amdgcn_wave_barrier(); }
template <class L> __device__ __forceinline__ uint32_t wave_lower_bound(const L &R, uint32_t n, int32_t s) {
  uint32_t lo = 0, hi = n;  // everything before lo is < s, everything from hi on is >= s
  const uint32_t lane = lane_id();
  while (hi - lo > 64u) {
    const uint32_t step = (hi - lo + 63u) >> 6;
    const uint32_t idx = lo + (lane + 1u) * step - 1u;  // last element of this lane's block
    const bool lt = idx < hi && R.x(idx) < s;
    const uint32_t c = (uint32_t)__popcll(__ballot(lt));   // the blocks that lie wholly below s are a prefix
    const uint32_t nlo = lo + c * step;
    hi = min(hi, nlo + step);
    lo = nlo;
  }
  const uint32_t idx = lo + lane;
  return lo + (uint32_t)__popcll(__ballot(idx < hi && R.x(idx) < s));
}
// (Round 5: four chunks of 64 per round trip.  A chunk was read, waited for and written before the next was read -- eight
// dependent LDS round trips to make room in the middle of a 1 000-range list, most of a sequential hit's ~1 200 clocks.
// The chunks of one round neither read what another of them writes -- going down, chunk j writes [b_j + 1, t_j + 1) and
// the chunks below it read below b_j -- so all four are requested, then all four stored; IMPG_VW_SHIFT_CHUNKS = 1 is the old loop.)
#ifndef IMPG_VW_WINDOW_TURN
#define IMPG_VW_WINDOW_TURN 1  // a sequential hit's turn on one 64-range read of the list (replay_hits_wave); 0: round 3's loops
#endif
#ifndef IMPG_VW_SHIFT_CHUNKS
#define IMPG_VW_SHIFT_CHUNKS 1  // (4: measured on config 5, 4 000 windows: update 629-634 -> 644 ms -- the shifts are not what a sequential hit waits for)
#endif
constexpr uint32_t VW_SHIFT_CHUNKS = IMPG_VW_SHIFT_CHUNKS;
template <class L> __device__ __forceinline__ void wave_shift_up(const L &R, uint32_t pos, uint32_t len) {  // [pos, len) -> [pos+1, len+1)
  const uint32_t lane = lane_id();
  for (uint32_t top = len; top > pos;) {
    int32_t vx[VW_SHIFT_CHUNKS], vy[VW_SHIFT_CHUNKS];
    uint32_t at[VW_SHIFT_CHUNKS];
    bool on[VW_SHIFT_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < VW_SHIFT_CHUNKS; c++) {
      const uint32_t base = top > pos + 64u ? top - 64u : pos;
      at[c] = base + lane;
      on[c] = at[c] < top;
      vx[c] = 0; vy[c] = 0;
      if (on[c]) { vx[c] = R.x(at[c]); vy[c] = R.y(at[c]); }
      top = base;  // (an exhausted range leaves top == pos: the remaining chunks of the round are empty)
    }
    order_point(R);
#pragma unroll
    for (uint32_t c = 0; c < VW_SHIFT_CHUNKS; c++)
      if (on[c]) { R.x(at[c] + 1) = vx[c]; R.y(at[c] + 1) = vy[c]; }
    order_point(R);
  }
}
template <class L> __device__ __forceinline__ void wave_shift_down(const L &R, uint32_t from, uint32_t len, uint32_t k) {  // [from, len) -> [from-k, len-k)
  const uint32_t lane = lane_id();
  for (uint32_t base = from; base < len; base += 64u * VW_SHIFT_CHUNKS) {
    int32_t vx[VW_SHIFT_CHUNKS], vy[VW_SHIFT_CHUNKS];
#pragma unroll
    for (uint32_t c = 0; c < VW_SHIFT_CHUNKS; c++) {
      const uint32_t i = base + c * 64u + lane;
      vx[c] = 0; vy[c] = 0;
      if (i < len) { vx[c] = R.x(i); vy[c] = R.y(i); }
    }
    order_point(R);
#pragma unroll
    for (uint32_t c = 0; c < VW_SHIFT_CHUNKS; c++) {
      const uint32_t i = base + c * 64u + lane;
      if (i < len) { R.x(i - k) = vx[c]; R.y(i - k) = vy[c]; }
    }
    order_point(R);
  }
}
// One compare-exchange of the network between a lane's element and another lane's, in registers (the lower place keeps
// the smaller start; equal starts stay where they are; a place past the end holds +infinity and never moves).
__device__ __forceinline__ void sort_cx_lane(int2 &v, uint32_t partner) {
  int2 o;
  o.x = __shfl(v.x, (int)partner);
  o.y = __shfl(v.y, (int)partner);
  const bool take = partner > lane_id() ? v.x > o.x : v.x < o.x;
  if (take) v = o;
}
// The stages of the network whose partners are less than 64 places apart, for every aligned run of 64 places of q[0 .. m):
// a lane takes its place's piece into registers once and runs them all there (round 5: 45 of the 55 stages of a
// 1 024-piece sort; each used to be a pass over LDS with a barrier).  first_k: 0 = the stages j = 32 .. 1 of one merge
// (its wider stages are done), else every merge k = 2 .. first_k <= 64 from its mirror stage down.
__device__ __forceinline__ void sort_close_stages(int2 *q, uint32_t m, uint32_t first_k) {
  const uint32_t lane = lane_id();
  for (uint32_t base = 0; base < m; base += 64u) {
    const uint32_t i = base + lane;
    int2 v = make_int2(0x7FFFFFFF, 0);
    if (i < m) v = q[i];
    if (first_k) {
      for (uint32_t k = 2; k <= first_k; k <<= 1) {
        sort_cx_lane(v, lane ^ (k - 1u));
        for (uint32_t j = k >> 2; j > 0; j >>= 1) sort_cx_lane(v, lane ^ j);
      }
    } else {
#pragma unroll
      for (uint32_t j = 32; j > 0; j >>= 1) sort_cx_lane(v, lane ^ j);
    }
    if (i < m) q[i] = v;
  }
  __syncthreads();
}
// ascending sort of p[0..n) by .x (p in LDS, or a slice of global memory for the in-place lists' few pieces): a bitonic
// network in its all-ascending form (first step of every merge pairs i with its mirror image in the block), so an index
// past n behaves as +infinity without being stored
__device__ __forceinline__ void wave_sort_pieces(int2 *p, uint32_t n) {
  if (n < 2) return;
  const uint32_t lane = lane_id();
  uint32_t n2 = 1;
  while (n2 < n) n2 <<= 1;
  sort_close_stages(p, n, min(n2, 64u));
  for (uint32_t k = 128; k <= n2; k <<= 1) {
    for (uint32_t j = k >> 1; j >= 64u; j >>= 1) {
      for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t l = (j == (k >> 1)) ? (i ^ (k - 1u)) : (i ^ j);
        if (l > i && l < n) {
          const int2 a = p[i], b = p[l];
          if (a.x > b.x) { p[i] = b; p[l] = a; }
        }
      }
      __syncthreads();
    }
    sort_close_stages(p, n, 0u);
  }
}
// The same network for more pieces than the block's LDS holds (round 5: a deep group of config 5 leaves ~1 000 pieces, a
// quarter of the groups more than the 1 152 the buffer takes; sorted on their global slice, all 66 stages of a 2 048-piece
// sort were global-memory round trips, chunk after chunk -- with the piece-by-piece sweep behind it two thirds of the wave
// kernel's cycles, scripts/vw_clocks.py).  Tiles of TS pieces are sorted in LDS; of the merges beyond a tile only the stages
// whose partners lie in another tile run on the global slice, the rest again tile by tile in LDS: 2 048 pieces take one
// global stage and two passes over their tiles.
template <uint32_t TS>
__device__ __forceinline__ void wave_sort_pieces_tiled(int2 *p, uint32_t n, int2 *lds) {
  static_assert((TS & (TS - 1u)) == 0u, "tiles of a power of two");
  if (n < 2) return;
  const uint32_t lane = lane_id();
  uint32_t n2 = 1;
  while (n2 < n) n2 <<= 1;
  // every tile by itself
  for (uint32_t base = 0; base < n; base += TS) {
    const uint32_t m = min(TS, n - base);
    __syncthreads();
    for (uint32_t i = lane; i < m; i += 64u) lds[i] = p[base + i];
    __syncthreads();
    sort_close_stages(lds, m, min(min(TS, n2), 64u));
    for (uint32_t k = 128; k <= min(TS, n2); k <<= 1) {
      for (uint32_t j = k >> 1; j >= 64u; j >>= 1) {
        for (uint32_t i = lane; i < m; i += 64u) {
          const uint32_t l = (j == (k >> 1)) ? (i ^ (k - 1u)) : (i ^ j);
          if (l > i && l < m) {
            const int2 a = lds[i], b = lds[l];
            if (a.x > b.x) { lds[i] = b; lds[l] = a; }
          }
        }
        __syncthreads();
      }
      sort_close_stages(lds, m, 0u);
    }
    for (uint32_t i = lane; i < m; i += 64u) p[base + i] = lds[i];
  }
  __syncthreads();
  for (uint32_t k = 2u * TS; k <= n2; k <<= 1) {
    for (uint32_t j = k >> 1; j >= TS; j >>= 1) {  // partners in another tile: on the global slice
      for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t l = (j == (k >> 1)) ? (i ^ (k - 1u)) : (i ^ j);
        if (l > i && l < n) {
          const int2 a = p[i], b = p[l];
          if (a.x > b.x) { p[i] = b; p[l] = a; }
        }
      }
      __syncthreads();
    }
    for (uint32_t base = 0; base < n; base += TS) {  // partners inside the tile (j < TS <= k / 2: never the mirror form)
      const uint32_t m = min(TS, n - base);
      for (uint32_t i = lane; i < m; i += 64u) lds[i] = p[base + i];
      __syncthreads();
      for (uint32_t j = TS >> 1; j >= 64u; j >>= 1) {
        for (uint32_t i = lane; i < m; i += 64u) {
          const uint32_t l = i ^ j;
          if (l > i && l < m) {
            const int2 a = lds[i], b = lds[l];
            if (a.x > b.x) { lds[i] = b; lds[l] = a; }
          }
        }
        __syncthreads();
      }
      sort_close_stages(lds, m, 0u);
      for (uint32_t i = lane; i < m; i += 64u) p[base + i] = lds[i];
      __syncthreads();
    }
  }
}
// The sweep over the sorted pieces (impg.rs:2568-2584: a piece that starts at or before the running end of the range being
// built extends it, one that starts beyond it closes the range and opens the next), 64 pieces a step: the running end is
// a prefix maximum of the ends -- over ALL earlier pieces, which is the current range's own maximum: an earlier range
// ended before the current one began -- a piece that starts beyond it is a head; a head stores its start as range k's
// and the maximum before it as range k - 1's end.  (Piece by piece this was one dependent read per piece: 10^3 LDS round
// trips a group, global-memory ones when the pieces did not fit the buffer.)  S may be the slice `out` itself: range k
// lies at or below the piece that opens it, and a step's pieces are read before any of its stores.
__device__ __forceinline__ uint32_t wave_merge_sorted_pieces(const int2 *S, uint32_t n, int2 *out) {
  const uint32_t lane = lane_id();
  int32_t carry = (int32_t)0x80000000;
  uint32_t heads = 0;
  int32_t *o = reinterpret_cast<int32_t *>(out);
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t i = base + lane;
    const bool on = i < n;
    int2 q = make_int2(0x7FFFFFFF, (int32_t)0x80000000);
    if (on) q = S[i];
    int32_t m = q.y;  // inclusive prefix maximum of the ends over the step
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t y = __shfl_up(m, d);
      if ((int)lane >= d) m = max(m, y);
    }
    int32_t prev = __shfl_up(m, 1);
    prev = lane == 0 ? carry : max(carry, prev);
    const bool head = on && (i == 0u || q.x > prev);
    const unsigned long long hm = __ballot(head);
    const uint32_t k = heads + (uint32_t)__popcll(hm & lanemask_lt());
    __syncthreads();  // (every piece of the step has been read: its stores may land on them)
    if (head) {
      o[2u * k] = q.x;
      if (k > 0u) o[2u * (k - 1u) + 1u] = prev;
    }
    carry = max(carry, __shfl(m, 63));
    heads += (uint32_t)__popcll(hm);
  }
  if (n && lane == 0) o[2u * (heads - 1u) + 1u] = carry;
  return heads;
}
// The wave's 64 (key, lane) pairs in ascending order of the key, a pair per lane, by a bitonic network on the lanes
// themselves (21 exchanges of two shuffles; ties by lane: a total order, so both partners of an exchange agree).
__device__ __forceinline__ void wave_sort_by_key(int32_t &key, uint32_t &val) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (uint32_t k = 2; k <= 64u; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const int32_t ok = __shfl_xor(key, (int)j);
      const uint32_t ov = (uint32_t)__shfl_xor((int)val, (int)j);
      const bool keep_min = ((lane & j) == 0u) == ((lane & k) == 0u);  // (k = 64: every block ascends)
      const bool other_less = ok < key || (ok == key && ov < val);
      if (keep_min == other_less) { key = ok; val = ov; }
    }
  }
}
// -DIMPG_VW_CLOCKS (experiments, scripts/vw_clocks.py): where a wave of the deep-closure replay spends its cycles
#ifdef IMPG_VW_CLOCKS
constexpr uint32_t VW_CLK_ROWS = 256;
__device__ unsigned long long g_vw_clk[VW_CLK_ROWS][16];
#define VW_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define VW_ADD(slot_, val_) do { if (lane_id() == 0) atomicAdd(&g_vw_clk[blockIdx.x % VW_CLK_ROWS][slot_], (unsigned long long)(val_)); } while (0)
extern "C" void impg_gpu_debug_vw_clocks(unsigned long long *out) {
  static unsigned long long rows[VW_CLK_ROWS][16];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_vw_clk), sizeof(rows));
  for (int k = 0; k < 16; k++) { out[k] = 0; for (uint32_t r = 0; r < VW_CLK_ROWS; r++) out[k] += rows[r][k]; }
  memset(rows, 0, sizeof(rows));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_vw_clk), rows, sizeof(rows));
}
#else
#define VW_T(v) do { } while (0)
#define VW_ADD(slot_, val_) do { } while (0)
#endif
template <class L>
__device__ __forceinline__ uint32_t replay_hits_wave(const L &R, uint32_t len, const unsigned long long *__restrict__ svals,
                                                     uint32_t st, uint32_t n, int32_t sequence_length,
                                                     int32_t min_transitive_len, int32_t mdbr, int2 *P, uint32_t &np,
                                                     uint32_t &t_next, uint32_t list_cap) {
  const bool writer = lane_id() == 0;
  const uint32_t lane = lane_id();
#if !IMPG_VW_WINDOWS
  __shared__ uint32_t iso_point[64];
  const int32_t iso_margin = max(mdbr, 0) + 1;
#endif
  uint32_t t0 = t_next;
  for (; t0 < n; t0 += 64u) {
    if (len + 64u > list_cap) break;  // (a batch adds at most 64 ranges: the caller moves the list somewhere bigger)
    // A hit that one range of the list already covers changes nothing whenever its turn comes: the insert finds no
    // uncovered piece and leaves the list as it is (impg.rs:314-343), and the list only ever grows.  Deep levels of
    // a saturating closure are almost all such hits: every lane tests one hit of the batch against the list as it
    // stands, and only the others take their turn in the sequential replay below.
#if IMPG_VW_WINDOWS
    // ---- round 5: every hit that meets no EARLIER hit of its batch takes its turn at once -------------------------------
    // A hit's turn reads and writes a handful of neighbouring ranges and nothing else: the range before its lower bound
    // (proximity test, impg.rs:2513-2545; the walk's first range, :305-313), the ranges that start inside it (the walk,
    // the insert, the ranges merge_forward swallows, :314-368) and the first range that starts beyond it (where the walk,
    // the second proximity test and the merge stop) -- list positions [lb - 1, ub], lb = first range starting at or after
    // the hit's start, ub = first range starting beyond its end.  Two hits whose position intervals are disjoint cannot
    // tell in which order they were replayed; so every uncovered hit whose interval meets that of no EARLIER uncovered hit
    // of the batch (a later one it meets waits: that one needs this one's result) is replayed NOW, by its own lane, with
    // the reference's own code (replay_one) on a private copy of its <= W + 2 ranges, and the list is rebuilt once for
    // all of them: the copied stretches taken out (one ascending pass), room made for what they became (one descending
    // pass), the private lists written back.  What is left -- hits behind an earlier one they meet, hits on more than W
    // ranges, hits the clamps touch -- takes its turn in the sequential part below, in order.  (Round 3's two classes,
    // hits that meet no range at all and hits that grow exactly one, were the cases ub - lb = 0 and most of = 1: 57 % of
    // the uncovered hits of a deep level; a list of ~1 000 ranges on 5 Mb under 5-10 kb hits has most hits on two or three.)
    constexpr uint32_t W = VW_WINDOW, PRIV = VW_WINDOW + 4u;
    __shared__ int32_t priv_x[PRIV * 64u], priv_y[PRIV * 64u];
    __shared__ uint32_t s_os[64], s_ol[64], s_cr[64], s_cn[64], s_cpos[64];
    unsigned long long todo;
    int32_t rs = 0, re = 0;  // this lane's hit of the batch (the sequential part below reads them lane to lane)
    {
      bool need = false, plain = false;
      uint32_t lb = 0, ub = 0;
      if (t0 + lane < n) {
        const unsigned long long iv = svals[st + t0 + lane];
        rs = (int32_t)(uint32_t)(iv >> 32); re = (int32_t)(uint32_t)iv;
        const int32_t s0 = max(rs, 0), e0 = min(re, sequence_length);
        const uint32_t p0 = list_lower_bound(R, len, s0);
        // (a hit that the clamps leave empty or inverted -- a length-0 set of a masked batch -- takes the literal path)
        const bool covered = s0 < e0 && ((p0 < len && R.x(p0) == s0 && R.y(p0) >= e0) || (p0 > 0 && R.y(p0 - 1) >= e0));
        need = !covered;
        plain = need && rs >= 0 && re <= sequence_length && rs < re && re < 0x7FFFFFFF;
        if (plain) { lb = p0; ub = list_lower_bound(R, len, re + 1); }
      }
      todo = __ballot(need);
      // the hit's interval of list positions, closed, not clamped (-1 and len stand for "before the first" / "behind the
      // last range": two hits in front of the whole list meet there); a hit the clamps touch meets everything
      const int32_t blo = plain ? (int32_t)lb - 1 : (int32_t)0x80000000, bhi = plain ? (int32_t)ub : 0x7FFFFFFF;
      bool earlier = false;
      for (unsigned long long left = todo; left; left &= left - 1ull) {
        const uint32_t j = (uint32_t)__ffsll((long long)left) - 1u;
        const int32_t lo_j = __builtin_amdgcn_readlane(blo, j), hi_j = __builtin_amdgcn_readlane(bhi, j);
        if (j < lane && lo_j <= bhi && blo <= hi_j) earlier = true;
      }
      const bool par = plain && ub - lb <= W && !earlier;
      const unsigned long long pm = __ballot(par);
      if (pm) {
        const uint32_t k = (uint32_t)__popcll(pm);
        // the private copy and the hit's turn on it
        const uint32_t bs = (uint32_t)max(blo, 0), be = min((uint32_t)bhi + 1u, len), ol = par ? be - bs : 0u;
        const ListInLds Q{priv_x + lane, priv_y + lane};
        uint32_t pl = ol, pn = 0;
        int32_t qx0 = 0, qx1 = 0, qx2 = 0, qx3 = 0, qx4 = 0, qx5 = 0, qy0 = 0, qy1 = 0, qy2 = 0, qy3 = 0, qy4 = 0, qy5 = 0;  // its pieces: at most W + 2
        static_assert(VW_WINDOW + 2u <= 6u, "a hit's pieces are held in six register pairs");
        if (par) {
          for (uint32_t i = 0; i < ol; i++) { Q.x(i) = R.x(bs + i); Q.y(i) = R.y(bs + i); }
          replay_one(Q, pl, rs, re, sequence_length, min_transitive_len, mdbr,
                     [&](int32_t a, int32_t b) {
                       if (pn == 0) { qx0 = a; qy0 = b; } else if (pn == 1) { qx1 = a; qy1 = b; } else if (pn == 2) { qx2 = a; qy2 = b; }
                       else if (pn == 3) { qx3 = a; qy3 = b; } else if (pn == 4) { qx4 = a; qy4 = b; } else { qx5 = a; qy5 = b; }
                       pn += 1;
                     }, [] {});
        }
        {  // the pieces, in any order (they are sorted afterwards)
          const uint32_t inc = wave_incl_scan(pn);
          const uint32_t at = np + inc - pn;
          if (pn > 0) P[at] = make_int2(qx0, qy0);
          if (pn > 1) P[at + 1u] = make_int2(qx1, qy1);
          if (pn > 2) P[at + 2u] = make_int2(qx2, qy2);
          if (pn > 3) P[at + 3u] = make_int2(qx3, qy3);
          if (pn > 4) P[at + 4u] = make_int2(qx4, qy4);
          if (pn > 5) P[at + 5u] = make_int2(qx5, qy5);
          np += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        // the copied stretches in list order (their intervals are disjoint: ranked by where they start)
        uint32_t rank = 0;
        for (unsigned long long left = pm; left; left &= left - 1ull) {
          const uint32_t j = (uint32_t)__ffsll((long long)left) - 1u;
          rank += __builtin_amdgcn_readlane(blo, j) < blo ? 1u : 0u;
        }
        __syncthreads();  // (the arrays may still be read by the previous batch)
        if (par) { s_os[rank] = bs; s_ol[rank] = ol; s_cn[rank] = pl; }
        __syncthreads();
        {
          const uint32_t o = lane < k ? s_ol[lane] : 0u, nl = lane < k ? s_cn[lane] : 0u;
          const uint32_t cr = wave_incl_scan(o), cn = wave_incl_scan(nl);
          __syncthreads();
          if (lane < k) { s_cr[lane] = cr; s_cn[lane] = cn; s_cpos[lane] = s_os[lane] - (cr - o); }
          __syncthreads();
        }
        const uint32_t removed = s_cr[k - 1u], added = s_cn[k - 1u];
        // (number of entries of the ascending array a[0 .. k) that are <= v)
        auto count_le = [&](const uint32_t *a, uint32_t v) -> uint32_t {
          uint32_t lo = 0, hi = k;
          while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1u; else hi = mid; }
          return lo;
        };
        // pass A, ascending: the copied stretches out, what stays moves down by what was taken out before it
        for (uint32_t base = s_os[0]; base < len; base += 64u) {
          const uint32_t i = base + lane;
          const bool on = i < len;
          int32_t vx = 0, vy = 0;
          uint32_t dst = 0;
          bool keep = false;
          if (on) {
            vx = R.x(i); vy = R.y(i);
            const uint32_t j = count_le(s_os, i);
            keep = j == 0u || i >= s_os[j - 1u] + s_ol[j - 1u];
            dst = i - (j ? s_cr[j - 1u] : 0u);
          }
          order_point(R);
          if (keep) { R.x(dst) = vx; R.y(dst) = vy; }
          order_point(R);
        }
        const uint32_t len_c = len - removed;
        // pass B, descending: room for what the stretches became
        for (uint32_t top = len_c; top > s_cpos[0];) {
          const uint32_t base = top > s_cpos[0] + 64u ? top - 64u : s_cpos[0];
          const uint32_t c = base + lane;
          const bool on = c < top;
          int32_t vx = 0, vy = 0;
          uint32_t dst = 0;
          if (on) {
            vx = R.x(c); vy = R.y(c);
            const uint32_t j = count_le(s_cpos, c);
            dst = c + (j ? s_cn[j - 1u] : 0u);
          }
          order_point(R);
          if (on) { R.x(dst) = vx; R.y(dst) = vy; }
          order_point(R);
          top = base;
        }
        if (par) {
          const uint32_t at = s_cpos[rank] + (s_cn[rank] - pl);
          for (uint32_t i = 0; i < pl; i++) { R.x(at + i) = Q.x(i); R.y(at + i) = Q.y(i); }
        }
        order_point(R);
        len = len_c + added;
        todo &= ~pm;
      }
    }
#else
    unsigned long long todo;
    int32_t rs = 0, re = 0;  // this lane's hit of the batch (the sequential part below reads them lane to lane)
    VW_T(vw0);
    {
      bool need = false, cand = false;
      uint32_t p0 = 0, q = 0, ext = 0;  // ext: 1 / 2 = the hit grows the one range q (see below)
      int32_t flo = 0, fhi = 0, xq = 0, yq = 0;  // the stretch nothing else may come near: the hit (and its range) plus the margin
      if (t0 + lane < n) {
        const unsigned long long iv = svals[st + t0 + lane];
        rs = (int32_t)(uint32_t)(iv >> 32); re = (int32_t)(uint32_t)iv;
        const int32_t s0 = max(rs, 0), e0 = min(re, sequence_length);
        p0 = list_lower_bound(R, len, s0);
        // (a hit that the clamps leave empty or inverted -- a length-0 set of a masked batch -- takes the literal path)
        const bool covered = s0 < e0 && ((p0 < len && R.x(p0) == s0 && R.y(p0) >= e0) || (p0 > 0 && R.y(p0 - 1) >= e0));
        need = !covered;
        // Isolated hits.  A hit that no clamp touches and that lies farther than the merge distance from every range of
        // the list and from every other hit of the batch still to be replayed interacts with nothing: whenever its turn
        // comes, both distance tests pass (impg.rs:2513-2545), its walk meets no range and emits the hit itself as one
        // piece (:314-328), and the insert neither extends nor merges anything (:330-343) -- and it changes none of that
        // for the others.  Such hits of a batch go into the list together, one pass over the list instead of one
        // shift each: at depth 4-5 of a saturating closure half the hits are such inserts into lists of ~1 000 ranges.
        const bool plain = need && rs >= 0 && re <= sequence_length && rs < re;
        const bool prev_far = p0 == 0 || R.y(p0 - 1) < rs - iso_margin, next_far = p0 == len || R.x(p0) > re + iso_margin;
        cand = plain && prev_far && next_far;
        flo = rs - iso_margin; fhi = re + iso_margin;
        // Hits that grow exactly ONE range.  The hit overlaps or touches the range before its lower bound (A) or the one
        // at it (B), every other range is farther than the merge distance from what the two become together, and no
        // other uncovered hit of the batch comes near that stretch: then its turn in the replay is a few lines of
        // arithmetic on that one range -- the distance tests against it (impg.rs:2513-2545: a hit whose end lies within
        // the merge distance of the range's is dropped), at most two pieces, the uncovered stretch on either side
        // (:314-328), the range's new ends (:330-343), nothing to swallow (:355-368) -- and no range changes its place.
        if (plain && !cand) {
          if (p0 > 0 && R.y(p0 - 1) >= rs && next_far) {  // A: q = p0 - 1 starts before the hit and reaches it
            ext = 1; q = p0 - 1u; xq = R.x(q); yq = R.y(q);
            flo = xq - iso_margin;
          } else if (prev_far && p0 < len && R.x(p0) <= re) {  // B: q = p0 starts inside the hit (or at its end)
            xq = R.x(p0); yq = R.y(p0);
            if (p0 + 1u == len || R.x(p0 + 1u) > max(re, yq) + iso_margin) { ext = 2; q = p0; fhi = max(re, yq) + iso_margin; }
          }
        }
      }
      todo = __ballot(need);
      VW_T(vw1);
      VW_ADD(0, vw1 - vw0); VW_ADD(4, min(64u, n - t0)); VW_ADD(5, __popcll(todo));
      unsigned long long iso = 0ull, grown = 0ull;
      uint32_t iso_rank = 0;  // an isolated hit's place among the isolated hits of the batch, by start
      if (__ballot(cand || ext != 0) != 0ull) {
        // "nothing else that is still to be replayed comes near this hit's stretch": with the hits still to be replayed in
        // the order of their starts, something BEFORE this hit reaches its stretch iff the largest end (+ margin) before
        // it does, something AFTER it iff the next start (- margin) does -- a sort of the wave's 64 starts on the lanes, a
        // prefix maximum and a look at the neighbour, where round 3 had every lane walk over all the others (a
        // v_readlane pair per hit and lane: 14 % of the deep groups' cycles).  The stretch [flo, fhi] contains the hit's
        // own [start - margin, end + margin], which is what makes "before" and "after" one-sided tests.
        const bool in_todo = ((todo >> lane) & 1ull) != 0ull;
        int32_t key = in_todo ? rs - iso_margin : 0x7FFFFFFF;
        uint32_t o = lane;
        wave_sort_by_key(key, o);
        const int32_t hi_o = __shfl(in_todo ? re + iso_margin : (int32_t)0x80000000, (int)o);
        const int32_t flo_o = __shfl(flo, (int)o), fhi_o = __shfl(fhi, (int)o);
        const bool cand_o = __shfl((int)cand, (int)o) != 0;
        int32_t pm = hi_o;  // inclusive prefix maximum of the ends in start order
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int32_t y = __shfl_up(pm, d);
          if ((int)lane >= d) pm = max(pm, y);
        }
        int32_t before = __shfl_up(pm, 1);
        if (lane == 0) before = (int32_t)0x80000000;
        int32_t next = __shfl_down(key, 1);
        if (lane == 63u) next = 0x7FFFFFFF;
        const bool clear_o = before < flo_o && next > fhi_o;
        const unsigned long long iso_o = __ballot(cand_o && clear_o);
        __syncthreads();  // (iso_point may still be read by the previous batch)
        iso_point[o] = (clear_o ? 1u : 0u) | ((uint32_t)__popcll(iso_o & lanemask_lt()) << 1);
        __syncthreads();
        const uint32_t back = iso_point[lane];
        const bool clear = (back & 1u) != 0u;
        iso_rank = back >> 1;
        cand = cand && clear;
        iso = __ballot(cand);
        grown = __ballot(ext != 0 && clear);
        if (ext != 0 && !clear) ext = 0;
      }
      VW_T(vw2);
      VW_ADD(1, vw2 - vw1); VW_ADD(6, __popcll(grown) + (__popcll(iso) >= 2 ? __popcll(iso) : 0));
      if (grown) {
        bool p1 = false, p2 = false;
        int2 pc1 = make_int2(0, 0), pc2 = make_int2(0, 0);
        if (ext == 1) {
          // the first distance test is against this range's end, the second against a range that is far away
          if (!(mdbr > 0 && yq - rs < mdbr)) {
            p1 = re - yq >= min_transitive_len;  // (yq < re: the hit is not covered)
            pc1 = make_int2(yq, re);
            R.y(q) = re;
          }
        } else if (ext == 2) {
          if (!(mdbr > 0 && re - xq < mdbr)) {  // (xq <= re; the first test is against a range that is far away)
            p1 = rs < xq && xq - rs >= min_transitive_len;
            pc1 = make_int2(rs, xq);
            p2 = yq < re && re - yq >= min_transitive_len;
            pc2 = make_int2(yq, re);
            R.x(q) = rs;  // (xq >= rs: the lower bound)
            R.y(q) = max(re, yq);
          }
        }
        const unsigned long long m1 = __ballot(p1), m2 = __ballot(p2);
        if (p1) P[np + (uint32_t)__popcll(m1 & lanemask_lt())] = pc1;
        np += (uint32_t)__popcll(m1);
        if (p2) P[np + (uint32_t)__popcll(m2 & lanemask_lt())] = pc2;
        np += (uint32_t)__popcll(m2);
        order_point(R);
        todo &= ~grown;
      }
      if (__popcll(iso) >= 2) {
        const uint32_t k = (uint32_t)__popcll(iso);
        const uint32_t rank = iso_rank;  // (from the sort above: the isolated hits before this one in start order)
        // their pieces (any order: the pieces are sorted afterwards)
        const bool emit = cand && re - rs >= min_transitive_len;
        const unsigned long long em = __ballot(emit);
        if (emit) P[np + (uint32_t)__popcll(em & lanemask_lt())] = make_int2(rs, re);
        np += (uint32_t)__popcll(em);
        // lane r takes the insertion point of the new range of rank r (ascending with r: the list is sorted too)
        __syncthreads();  // (iso_point may still be read by the previous batch)
        if (cand) iso_point[rank] = p0;
        __syncthreads();
        const uint32_t ps = lane < k ? iso_point[lane] : 0xFFFFFFFFu;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)ps);  // ranges before it stay where they are
        // every range moves up by the number of new ranges that go in at or before it, from the top of the list down:
        // those below the chunk in one ballot, those inside it one by one (a couple per chunk)
        for (uint32_t top = len; top > first;) {
          const uint32_t base = top > first + 64u ? top - 64u : first;
          const uint32_t i = base + lane;
          const bool on = i < top;
          int32_t vx = 0, vy = 0;
          if (on) { vx = R.x(i); vy = R.y(i); }
          uint32_t c = (uint32_t)__popcll(__ballot(ps <= base));
          for (unsigned long long in = __ballot(ps > base && ps < top); in; in &= in - 1ull)
            c += (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)(__ffsll((long long)in) - 1)) <= i ? 1u : 0u;
          order_point(R);
          if (on) { R.x(i + c) = vx; R.y(i + c) = vy; }
          order_point(R);
          top = base;
        }
        if (cand) { R.x(p0 + rank) = rs; R.y(p0 + rank) = re; }
        order_point(R);
        len += k;
        todo &= ~iso;
      }
      VW_T(vw3);
      VW_ADD(2, vw3 - vw2);
    }
#endif
    VW_ADD(8, __popcll(todo));
    while (todo) {
    VW_T(vs0);
    const uint32_t tl = (uint32_t)__ffsll((long long)todo) - 1u;
    todo &= todo - 1ull;
    // (from the lane that loaded it: a second read of svals here was a global-memory round trip per replayed hit)
    int32_t start = __builtin_amdgcn_readlane(rs, tl), end = __builtin_amdgcn_readlane(re, tl);
    uint32_t pos = wave_lower_bound(R, len, start);
#if IMPG_VW_WINDOW_TURN
    // Round 5: a sequential hit's turn on ONE read of the list.  Everything the turn looks at -- the range before the hit's
    // lower bound (proximity test, the walk's first range, the insert's extension), the ranges that start inside the
    // hit (the walk's gaps, the ranges merge_forward swallows) and the first one beyond -- is the 64 ranges from pos - 1:
    // a lane takes one, and the walk (impg.rs:314-328), the insert (:330-343) and the merge (:355-368) become ballots
    // over them instead of loops of dependent LDS reads (4 200 clocks a hit, a third of a deep group's cycles).  The
    // walk's `current` before range k is max(start, end of range k - 1) -- the list's ends ascend --, the ranges it
    // visits are a prefix (start <= end and current < end), and what the grown range swallows are the ranges that start
    // at or below its new end: also a prefix, and the last of them gives the final end.  A hit the clamps touch, or
    // whose ranges run past the 64, takes the loops below.
    {
      const uint32_t w0 = pos ? pos - 1u : 0u, wi = w0 + lane;
      const bool wv = wi < len;
      int32_t wx = 0x7FFFFFFF, wyv = (int32_t)0x80000000;
      if (wv) { wx = R.x(wi); wyv = R.y(wi); }
      const uint32_t tp = pos - w0;  // the lane of range `pos`: 1, or 0 in front of the whole list
      const int32_t y_prev = __builtin_amdgcn_readlane(wyv, 0);                       // (meaningful iff pos > 0)
      const int32_t x_pos = tp ? __builtin_amdgcn_readlane(wx, 1) : __builtin_amdgcn_readlane(wx, 0);  // INT_MAX iff pos == len
      const int32_t y_pos = tp ? __builtin_amdgcn_readlane(wyv, 1) : __builtin_amdgcn_readlane(wyv, 0);
      const bool far = w0 + 64u < len && __builtin_amdgcn_readlane(wx, 63) <= end;
      if (start >= 0 && end <= sequence_length && !far) {
        if (mdbr > 0) {  // impg.rs:2513-2545
          bool should_add = true;
          if (pos > 0 && abs(start - y_prev) < mdbr) should_add = false;
          if (should_add && pos < len && abs(x_pos - end) < mdbr) should_add = false;
          if (!should_add) { VW_T(vsx); VW_ADD(9, vsx - vs0); continue; }
        }
        VW_T(vs1);
        VW_ADD(9, vs1 - vs0);
        // the walk: ranges from lane `off` on
        const uint32_t off = (pos > 0 && y_prev > start) ? 0u : tp;
        const int32_t yp = __shfl_up(wyv, 1);
        const int32_t cur = lane <= off ? start : max(start, yp);
        const bool proc = lane >= off && wv && wx <= end && cur < end;
        const unsigned long long pmask = __ballot(proc) >> off;
        const uint32_t K = ~pmask ? (uint32_t)__builtin_ctzll(~pmask) : 64u - off;  // the ranges the walk visits: lanes [off, off + K)
        const bool gap = lane >= off && lane < off + K && cur < wx && wx - cur >= min_transitive_len;
        const unsigned long long gm = __ballot(gap);
        if (gap) P[np + (uint32_t)__popcll(gm & lanemask_lt())] = make_int2(cur, wx);
        np += (uint32_t)__popcll(gm);
        const int32_t cK = K ? max(start, (int32_t)__shfl(wyv, (int)(off + K - 1u))) : start;
        if (cK < end && end - cK >= min_transitive_len) {
          if (writer) P[np] = make_int2(cK, end);
          np++;
        }
        VW_T(vs2);
        VW_ADD(10, vs2 - vs1);
        // the insert (impg.rs:330-343) and what the grown range swallows (merge_forward_from, :355-368)
        const bool ext_prev = pos > 0 && y_prev >= start, ext_pos = !ext_prev && pos < len && end >= x_pos;
        if (!ext_prev && !ext_pos) {
          wave_shift_up(R, pos, len);
          if (writer) { R.x(pos) = start; R.y(pos) = end; }
          order_point(R);
          len += 1;
          { VW_T(vs3); VW_ADD(11, vs3 - vs2); }
          continue;
        }
        const uint32_t tm = ext_prev ? 0u : tp, mfrom = w0 + tm;
        const int32_t wy0 = ext_prev ? max(y_prev, end) : max(end, y_pos);
        const unsigned long long sw = __ballot(lane > tm && wv && wx <= wy0);
        const uint32_t k = (uint32_t)__popcll(sw);
        const int32_t wyf = k ? max(wy0, (int32_t)__shfl(wyv, (int)(tm + k))) : wy0;
        order_point(R);
        if (writer) {
          R.y(mfrom) = wyf;
          if (ext_pos) R.x(mfrom) = min(start, x_pos);
        }
        order_point(R);
        if (k) {
          wave_shift_down(R, mfrom + 1u + k, len, k);
          len -= k;
        }
        { VW_T(vs4); VW_ADD(12, vs4 - vs2); }
        continue;
      }
    }
#endif
    if (mdbr > 0) {  // impg.rs:2513-2545
      bool should_add = true;
      if (pos > 0 && abs(start - R.y(pos - 1)) < mdbr) should_add = false;
      if (should_add && pos < len && abs(R.x(pos) - end) < mdbr) should_add = false;
      if (!should_add) { VW_T(vsx); VW_ADD(9, vsx - vs0); continue; }
    }
    VW_T(vs1);
    VW_ADD(9, vs1 - vs0);
    if (start < 0) { start = 0; pos = wave_lower_bound(R, len, start); }  // impg.rs:287-289 (never taken on real coordinates)
    if (end > sequence_length) end = sequence_length;                      // impg.rs:294-296
    int32_t current = start;
    uint32_t i = pos;
    if (i > 0 && R.y(i - 1) > start) i -= 1;
    while (i < len && current < end) {  // impg.rs:314-324
      const int32_t rx = R.x(i), ry = R.y(i);
      if (rx > end) break;
      if (current < rx && abs(rx - current) >= min_transitive_len) {
        if (writer) P[np] = make_int2(current, rx);
        np++;
      }
      current = max(current, ry);
      i += 1;
    }
    if (current < end && abs(end - current) >= min_transitive_len) {
      if (writer) P[np] = make_int2(current, end);
      np++;
    }
    VW_T(vs2);
    VW_ADD(10, vs2 - vs1);
    uint32_t mfrom;  // impg.rs:330-343
    if (pos > 0 && R.y(pos - 1) >= start) {
      const int32_t ny = max(R.y(pos - 1), end);
      order_point(R);
      if (writer) R.y(pos - 1) = ny;
      mfrom = pos - 1;
    } else if (pos < len && end >= R.x(pos)) {
      const int32_t nx = min(start, R.x(pos)), ny = max(end, R.y(pos));
      order_point(R);
      if (writer) { R.x(pos) = nx; R.y(pos) = ny; }
      mfrom = pos;
    } else {
      wave_shift_up(R, pos, len);
      if (writer) { R.x(pos) = start; R.y(pos) = end; }
      order_point(R);
      len += 1;
      { VW_T(vs3); VW_ADD(11, vs3 - vs2); }
      continue;
    }
    order_point(R);
    // merge_forward_from (impg.rs:355-368): the list is sorted and its ranges neither overlap nor touch, so the
    // ranges the grown one swallows are one run right behind it; the rest moves down by the run's length
    uint32_t read = mfrom + 1;
    int32_t wy = R.y(mfrom);
    while (read < len && wy >= R.x(read)) { wy = max(wy, R.y(read)); read += 1; }
    const uint32_t k = read - (mfrom + 1);
    if (k) {
      order_point(R);
      if (writer) R.y(mfrom) = wy;
      order_point(R);
      wave_shift_down(R, read, len, k);
      len -= k;
    }
    { VW_T(vs4); VW_ADD(12, vs4 - vs2); }
  }
  }
  t_next = t0;
  return len;
}
template <uint32_t CAP, bool DEAL = false>  // CAP: ranges of the list / pieces of the sort that fit the block's LDS (8 bytes each, one buffer for both)
__global__ __launch_bounds__(64) void visited_update_wave_kernel(const unsigned long long *__restrict__ svals,
                                                                 const int32_t *__restrict__ seq_len,
                                                                 const unsigned long long *__restrict__ gkey,
                                                                 const uint32_t *__restrict__ gstart,
                                                                 const uint32_t *__restrict__ glen,
                                                                 const int2 *const *__restrict__ old_src,
                                                                 const uint32_t *__restrict__ cap,
                                                                 const uint32_t *__restrict__ noff,
                                                                 const uint32_t *__restrict__ poff,
                                                                 const uint32_t *__restrict__ big_list,
                                                                 const uint32_t *__restrict__ n_big, uint32_t *__restrict__ next_group, uint32_t from_back,
                                                                 int32_t min_transitive_len,
                                                                 int32_t mdbr, int2 *__restrict__ new_ranges,
                                                                 uint32_t *__restrict__ new_len, int2 *__restrict__ pieces,
                                                                 uint32_t *__restrict__ n_pieces) {
  // The capacities group_prepare sized (old length + hits, + 2 x hits for the pieces) are worst cases; what a group
  // really needs is usually a fraction (hits pile up on the same regions and merge).  The list therefore starts in
  // LDS whatever its worst case and moves to its global slice only if it really outgrows the buffer; the pieces go
  // straight to their global slice during the replay (plain stores nobody waits for) and come back into the same
  // LDS buffer for the sort once the list has been written out.
  __shared__ int2 lds[CAP];
  int32_t *lx = reinterpret_cast<int32_t *>(lds), *ly = lx + CAP;
  const uint32_t lane = lane_id();
  const uint32_t nb = *n_big;
  // Groups are handed out one at a time (a counter), not dealt round-robin: a group's replay takes anything from
  // microseconds to milliseconds, and with a fixed deal the launch lasted as long as its unluckiest wave -- the waves of a
  // config-5 level were busy a fifth of the kernel's time.
  // (DEAL: groups of much the same small size -- the tiny working set -- are dealt round-robin instead: 10^5 of them
  // fetching their number from ONE counter queued on it for longer than they took to replay)
  __shared__ uint32_t next_b;
  for (uint32_t round = 0;; round++) {
    uint32_t b;
    if (DEAL) {
      b = blockIdx.x + round * gridDim.x;
    } else {
      __syncthreads();  // (everybody has read the previous next_b)
      if (lane == 0) next_b = atomicAdd(next_group, 1u);
      __syncthreads();
      b = next_b;
    }
    if (b >= nb) break;
    const uint32_t g = big_list[from_back ? from_back - 1u - b : b];  // (from_back = n_groups: the list's other end)
    VW_T(vwg0);
    int2 *R = new_ranges + noff[g];
    const uint32_t st = gstart[g], n = glen[g];
    const int2 *src = old_src[g];  // (the group's current list, resolved by group_prepare; cap = its length + the hits)
    uint32_t len = cap[g] - n;
    const int32_t sequence_length = seq_len[(uint32_t)(gkey[g] & 0xFFFFFFFFull)];
    int2 *P = pieces + poff[g];
    uint32_t np = 0, t_next = 0;
    __syncthreads();  // (the previous group's LDS contents are dead)
    if (len + 64u <= CAP) {
      for (uint32_t i = lane; i < len; i += 64u) { const int2 r = src[i]; lx[i] = r.x; ly[i] = r.y; }
      __syncthreads();
      len = replay_hits_wave(ListSoA{lx, ly}, len, svals, st, n, sequence_length, min_transitive_len, mdbr, P, np, t_next, CAP);
      __syncthreads();
      for (uint32_t i = lane; i < len; i += 64u) R[i] = make_int2(lx[i], ly[i]);
    } else {
      for (uint32_t i = lane; i < len; i += 64u) R[i] = src[i];
    }
    __syncthreads();
    if (t_next < n)  // the list outgrew the buffer (or never fitted): the rest of the replay in place
      len = replay_hits_wave(ListInPlace{R}, len, svals, st, n, sequence_length, min_transitive_len, mdbr, P, np, t_next, 0xFFFFFFFFu);
    __syncthreads();
    // next-depth ranges of the group: sorted by start, overlapping / contiguous ones merged (impg.rs:2568-2584)
    VW_T(vwp0);
    int2 *S = P;
    if (np <= CAP) {
      for (uint32_t i = lane; i < np; i += 64u) lds[i] = P[i];
      __syncthreads();
      S = lds;
      wave_sort_pieces(S, np);
    } else {
      constexpr uint32_t TS = CAP >= 4096u ? 4096u : CAP >= 1024u ? 1024u : 128u;  // the largest power of two the buffer holds
      wave_sort_pieces_tiled<TS>(P, np, lds);
    }
    VW_T(vwp1);
    VW_ADD(13, vwp1 - vwp0); VW_ADD(7, np > CAP ? 1 : 0);
    const uint32_t w = wave_merge_sorted_pieces(S, np, pieces + poff[g]);
    __syncthreads();
    { VW_T(vwp2); VW_ADD(14, vwp2 - vwp1); }
    if (lane == 0) { new_len[g] = len; n_pieces[g] = w; }
    { VW_T(vwg1); VW_ADD(15, vwg1 - vwg0); VW_ADD(3, 1); }
  }
}
// ---- hits the OLD list already covers, taken out before the replay --------------------------------------------------
// A hit that one range of its group's list covers changes nothing whenever its turn comes (see replay_hits_wave), and
// the list only grows -- so a hit covered by the list as the level FOUND it can be dropped before the sequential replay,
// by a pass that is parallel over hits instead of over groups.  Deep levels of a saturating closure (BASELINE config 5)
// are almost all such hits, and the replay of their big groups runs at 5 waves per CU.
__global__ __launch_bounds__(256) void covered_flags_kernel(const unsigned long long *__restrict__ svals,
                                                            const uint32_t *__restrict__ head, const uint32_t *__restrict__ gid,
                                                            const unsigned long long *__restrict__ gkey,
                                                            const int2 *const *__restrict__ old_src, const uint32_t *__restrict__ cap,
                                                            const uint32_t *__restrict__ glen,
                                                            const int32_t *__restrict__ seq_len, uint32_t n_active,
                                                            uint32_t *__restrict__ keep) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_active) return;
  const uint32_t g = gid[i] + head[i] - 1u;
  const int2 *src = old_src[g];
  const uint32_t len = cap[g] - glen[g];
  bool covered = false;
  if (len) {
    const int32_t sequence_length = seq_len[(uint32_t)(gkey[g] & 0xFFFFFFFFull)];
    const unsigned long long iv = svals[i];
    const int32_t s0 = max((int32_t)(uint32_t)(iv >> 32), 0), e0 = min((int32_t)(uint32_t)iv, sequence_length);
    const uint32_t p0 = lower_bound_start(src, len, s0);
    covered = s0 < e0 && ((p0 < len && src[p0].x == s0 && src[p0].y >= e0) || (p0 > 0 && src[p0 - 1].y >= e0));
  }
  keep[i] = covered ? 0u : 1u;
}
__global__ __launch_bounds__(256) void covered_compact_kernel(const unsigned long long *__restrict__ svals, const uint32_t *__restrict__ keep,
                                                              const uint32_t *__restrict__ kpos, uint32_t n_active,
                                                              unsigned long long *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_active && keep[i]) out[kpos[i]] = svals[i];
}
// the groups' runs in the compacted hit list, and the capacities that follow from the shorter runs
__global__ __launch_bounds__(256) void covered_regroup_kernel(const uint32_t *__restrict__ kpos, uint32_t n_active, uint32_t n_kept,
                                                              uint32_t n_groups, uint32_t *__restrict__ gstart, uint32_t *__restrict__ glen,
                                                              uint32_t *__restrict__ cap, uint32_t *__restrict__ pcap) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t st = gstart[g], n = glen[g], en = st + n;
  const uint32_t ns = kpos[st], ne = en < n_active ? kpos[en] : n_kept;
  const uint32_t olen = cap[g] - n;
  gstart[g] = ns;
  glen[g] = ne - ns;
  cap[g] = olen + (ne - ns);
  pcap[g] = olen + 2u * (ne - ns);
}

// The groups the dense lane kernel does not take, listed by who does (any order inside a list):
//   big_list[0 ..)                 wave kernel, small LDS working set     (count n_big[0])
//   big_list[n_groups - 1 .. down) wave kernel, large working set         (count n_big[1])
//   big_list[n_groups ..)          wave kernel, tiny working set          (count n_big[2])
//   big_list[2 n_groups ..)        the listed lane kernel (mid groups)    (count n_big[6])
// A block classifies 4 096 groups and claims its stretch of every list with ONE atomic per list: a wave claiming its
// own (a third of a headline level's waves hold a listed group) queued 10^5 atomics on one address -- 1.1 ms.
constexpr uint32_t BG_THREADS = 1024, BG_PER_THREAD = 4;
__global__ __launch_bounds__(BG_THREADS) void big_groups_kernel(const uint32_t *__restrict__ cap, const uint32_t *__restrict__ pcap,
                                                                uint32_t n_groups, uint32_t *__restrict__ big_list,
                                                                uint32_t *__restrict__ n_big) {
  __shared__ uint32_t cnt[4], base[4];
  if (threadIdx.x < 4u) cnt[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t tier[BG_PER_THREAD], off[BG_PER_THREAD];
#pragma unroll
  for (uint32_t k = 0; k < BG_PER_THREAD; k++) {
    const uint32_t g = (blockIdx.x * BG_PER_THREAD + k) * BG_THREADS + threadIdx.x;
    const uint32_t c = g < n_groups ? cap[g] : 0u;  // cap = old length + hits of the level (group_prepare)
    uint32_t t = 0;  // 0: the dense lane kernel's; 1 mid; 2 tiny / 3 small / 4 large working set of the wave kernel
    if (c > VU_TINY_MAX) {
      if (c <= VU_MID_MAX) t = 1u;
      else {
        // cap = old + hits, pcap = old + 2 hits  =>  old = 2 cap - pcap.  The small working set is chosen by the list
        // the group STARTS with, not by its worst case: hits of a deep level pile up on the same few regions, the list
        // grows by a fraction of their number, and a list that does outgrow the buffer moves to its global slice
        // (visited_update_wave_kernel); the tiny one only takes groups that cannot outgrow it.
        const uint32_t old_len = 2u * c - pcap[g];
        t = c + 64u <= VW_CAP_TINY ? 2u : (old_len + VW_SMALL_HEADROOM <= VW_CAP_SMALL - 64u ? 3u : 4u);
      }
    }
    tier[k] = t;
    off[k] = 0u;
    if (__ballot(t != 0u)) {
#pragma unroll
      for (uint32_t tt = 1; tt <= 4u; tt++) {
        const unsigned long long m = __ballot(t == tt);
        if (!m) continue;
        uint32_t o = 0;
        if (lane_id() == 0) o = atomicAdd(&cnt[tt - 1u], (uint32_t)__popcll(m));
        o = (uint32_t)__shfl((int)o, 0);
        if (t == tt) off[k] = o + (uint32_t)__popcll(m & lanemask_lt());
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 4u) {
    const uint32_t slot = threadIdx.x == 0u ? 6u : (threadIdx.x == 1u ? 2u : (threadIdx.x == 2u ? 0u : 1u));
    base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&n_big[slot], cnt[threadIdx.x]) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < BG_PER_THREAD; k++) {
    const uint32_t g = (blockIdx.x * BG_PER_THREAD + k) * BG_THREADS + threadIdx.x;
    const uint32_t t = tier[k];
    if (!t) continue;
    const uint32_t pos = base[t - 1u] + off[k];
    if (t == 1u) big_list[2u * (size_t)n_groups + pos] = g;
    else if (t == 2u) big_list[(size_t)n_groups + pos] = g;
    else if (t == 3u) big_list[pos] = g;
    else big_list[n_groups - 1u - pos] = g;
  }
}

__global__ __launch_bounds__(256) void frontier_emit_kernel(const unsigned long long *__restrict__ gkey,
                                                            const uint32_t *__restrict__ poff,
                                                            const uint32_t *__restrict__ n_pieces,
                                                            const uint32_t *__restrict__ foff, uint32_t n_groups,
                                                            const int2 *__restrict__ pieces,
                                                            FrontierRec *__restrict__ out, const SegDesc *__restrict__ seg,
                                                            const int32_t *__restrict__ seq_len, uint32_t n_seq,
                                                            uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
  // (key / idx: the next level's lookup-order keys beside the records, order_keys_kernel's words, while the record is in
  // registers -- one read of the frontier less)
  // A wave takes 64 consecutive groups, whose records are one contiguous stretch of the frontier: a lane per OUTPUT
  // record (its group found by a search over the lanes' offsets), so that a store instruction writes 1 KB in a row.
  // (A lane per group wrote its ~3.5 records 16 bytes at a time, 64 lines per instruction: 1.0 ms of a headline step.)
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  const bool in = g < n_groups;
  const unsigned long long k = in ? gkey[g] : 0ull;
  const uint32_t np = in ? n_pieces[g] : 0u;
  const uint32_t po = in ? poff[g] : 0u;
  const uint32_t fo = in ? foff[g] : 0xFFFFFFFFu;  // (beyond the list: never at or below an output place)
  const uint32_t first = (uint32_t)__shfl((int)fo, 0);
  // the stretch ends behind the last group of the wave that exists
  const unsigned long long live = __ballot(in);
  if (!live) return;
  const int last = 63 - __builtin_clzll(live);
  const uint32_t end = (uint32_t)__shfl((int)(fo + np), last);
  for (uint32_t o = first + lane_id(); o - lane_id() < end; o += 64u) {  // (wave-uniform trip count: the shuffles need every lane)
    uint32_t j = 0;  // the last group whose offset is <= o (groups without records share their successor's offset)
#pragma unroll
    for (uint32_t st = 32u; st > 0u; st >>= 1) {
      const uint32_t f = (uint32_t)__shfl((int)fo, (int)(j + st));
      j += f <= o ? st : 0u;
    }
    const uint32_t fj = (uint32_t)__shfl((int)fo, (int)j), pj = (uint32_t)__shfl((int)po, (int)j);
    const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)k, (int)j), khi = (uint32_t)__shfl((int)(uint32_t)(k >> 32), (int)j);
    if (o < end) {
      const int2 piece = pieces[pj + (o - fj)];
      FrontierRec f;
      f.target_id = klo;
      f.start = piece.x;
      f.end = piece.y;
      f.qidx = khi;
      out[o] = f;
      if (key) { key[o] = order_key(seg, seq_len, n_seq, klo, piece.x); idx[o] = o; }
    }
  }
}

// subset filter (impg.rs:2176-2185, :2430-2439; multi_impg.rs:888-896; main.rs:11693-11696): a hit survives iff its
// query sequence is the query's own target or the caller's keep[] says its name matches.  A dropped hit becomes a
// slot without a projection, which every later pass already skips.
__global__ __launch_bounds__(256) void subset_filter_kernel(const FrontierRec *__restrict__ fr,
                                                            const uint32_t *__restrict__ pair_range, uint32_t n_pairs,
                                                            uint32_t *__restrict__ qid, const uint8_t *__restrict__ keep,
                                                            const impg_gpu_range_t *__restrict__ ranges) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t q = qid[p];
  if (q == HIT_NONE) return;
  if (keep[q]) return;
  if (q != ranges[fr[pair_range[p]].qidx].target_id) qid[p] = HIT_NONE;
}

// level -1 under masked_regions (impg.rs:2077-2112, :2331-2373): the input range goes through
// SortedRanges::insert (min_distance 0) on a copy of its target's mask list; every piece is a self interval,
// the pieces of at least min_transitive_len open the frontier.
__global__ __launch_bounds__(256) void mask_caps_kernel(const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                        const uint32_t *__restrict__ mask_off, uint32_t n_seq,
                                                        uint32_t *__restrict__ cap) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  const uint32_t t = ranges[q].target_id;
  cap[q] = (t < n_seq ? mask_off[t + 1] - mask_off[t] : 0u) + 1u;
}
__global__ __launch_bounds__(64) void visited_init_masked_kernel(const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                                 const int32_t *__restrict__ init_len, uint32_t n_seq,
                                                                 const uint32_t *__restrict__ mask_off,
                                                                 const int2 *__restrict__ mask_ranges,
                                                                 int32_t min_transitive_len, const uint32_t *__restrict__ loff,
                                                                 unsigned long long *__restrict__ keys,
                                                                 uint32_t *__restrict__ len_out, int2 *__restrict__ rng,
                                                                 int2 *__restrict__ pieces, uint32_t *__restrict__ n_self,
                                                                 uint32_t *__restrict__ n_front) {
  const uint32_t q = blockIdx.x * 64u + threadIdx.x;
  if (q >= n) return;
  const impg_gpu_range_t r = ranges[q];
  int32_t start = min(r.start, r.end), end = max(r.start, r.end);
  const bool known = r.target_id < n_seq;
  const int32_t sequence_length = known ? init_len[r.target_id] : 0;  // absent from the map: length 0 (impg.rs:2048-2053)
  int2 *R = rng + loff[q];
  int2 *P = pieces + loff[q];
  uint32_t len = 0;
  if (known) {
    const uint32_t a = mask_off[r.target_id];
    len = mask_off[r.target_id + 1] - a;
    for (uint32_t i = 0; i < len; i++) R[i] = mask_ranges[a + i];
  }
  if (start < 0) start = 0;                          // impg.rs:287-289
  if (end > sequence_length) end = sequence_length;  // impg.rs:294-296
  uint32_t np = 0, nf = 0;
  int32_t current = start;
  uint32_t i = lower_bound_start(R, len, start);
  if (i > 0 && R[i - 1].y > start) i -= 1;
  while (i < len && current < end) {  // impg.rs:314-324
    const int2 rg = R[i];
    if (rg.x > end) break;
    if (current < rg.x) {
      P[np++] = make_int2(current, rg.x);
      nf += abs(rg.x - current) >= min_transitive_len;
    }
    current = max(current, rg.y);
    i += 1;
  }
  if (current < end) {
    P[np++] = make_int2(current, end);
    nf += abs(end - current) >= min_transitive_len;
  }
  const uint32_t pos = lower_bound_start(R, len, start);  // impg.rs:330-343
  bool merge = true;
  uint32_t mfrom = 0;
  if (pos > 0 && R[pos - 1].y >= start) {
    R[pos - 1].y = max(R[pos - 1].y, end);
    mfrom = pos - 1;
  } else if (pos < len && end >= R[pos].x) {
    R[pos].x = min(start, R[pos].x);
    R[pos].y = max(end, R[pos].y);
    mfrom = pos;
  } else {
    for (uint32_t k = len; k > pos; k--) R[k] = R[k - 1];
    R[pos] = make_int2(start, end);
    len += 1;
    merge = false;
  }
  if (merge) {  // merge_forward_from, impg.rs:355-368
    uint32_t write = mfrom, read = mfrom + 1;
    while (read < len) {
      if (R[write].y >= R[read].x) {
        R[write].y = max(R[write].y, R[read].y);
      } else {
        write += 1;
        int2 tmp = R[write];
        R[write] = R[read];
        R[read] = tmp;
      }
      read += 1;
    }
    len = write + 1;
  }
  keys[q] = ((unsigned long long)q << 32) | r.target_id;
  len_out[q] = len;
  n_self[q] = np;
  n_front[q] = nf;
}
__global__ __launch_bounds__(256) void masked_self_emit_kernel(const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                               const uint32_t *__restrict__ loff,
                                                               const uint32_t *__restrict__ n_self,
                                                               const uint32_t *__restrict__ self_off,
                                                               const uint32_t *__restrict__ front_off,
                                                               int32_t min_transitive_len, const int2 *__restrict__ pieces,
                                                               FrontierRec *__restrict__ self_out,
                                                               FrontierRec *__restrict__ frontier_out) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  const int2 *P = pieces + loff[q];
  const uint32_t np = n_self[q];
  uint32_t so = self_off[q], fo = front_off[q];
  for (uint32_t i = 0; i < np; i++) {
    FrontierRec f;
    f.target_id = ranges[q].target_id;
    f.start = P[i].x;
    f.end = P[i].y;
    f.qidx = q;
    self_out[so++] = f;                                                   // impg.rs:2345-2363
    if (abs(P[i].x - P[i].y) >= min_transitive_len) frontier_out[fo++] = f;  // impg.rs:2369-2373
  }
}

// level -1: every query's own range is visited on its target (impg.rs:2337-2340)
__global__ __launch_bounds__(256) void visited_init_kernel(const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                           const int32_t *__restrict__ seq_len, uint32_t n_seq,
                                                           int32_t min_transitive_len, unsigned long long *__restrict__ keys,
                                                           uint32_t *__restrict__ off, uint32_t *__restrict__ len,
                                                           int2 *__restrict__ rng, FrontierRec *__restrict__ self_iv,
                                                           uint32_t *__restrict__ in_frontier) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  const impg_gpu_range_t r = ranges[q];
  int32_t start = min(r.start, r.end), end = max(r.start, r.end);
  const int32_t sl = r.target_id < n_seq ? seq_len[r.target_id] : 0x7FFFFFFF;
  if (start < 0) start = 0;
  if (end > sl) end = sl;
  keys[q] = ((unsigned long long)q << 32) | r.target_id;
  off[q] = q;
  len[q] = 1;
  rng[q] = make_int2(start, end);
  FrontierRec f;
  f.target_id = r.target_id;
  f.start = start;
  f.end = end;
  f.qidx = q;
  self_iv[q] = f;  // the filtered input range = the self interval (impg.rs:2345-2363)
  in_frontier[q] = (start < end) && (abs(start - end) >= min_transitive_len);  // impg.rs:2369-2373
}
__global__ __launch_bounds__(256) void compact_frontier_kernel(const FrontierRec *__restrict__ in,
                                                               const uint32_t *__restrict__ flag,
                                                               const uint32_t *__restrict__ pos, uint32_t n,
                                                               FrontierRec *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) out[pos[i]] = in[i];
}
__global__ __launch_bounds__(256) void ranges_to_frontier_kernel(const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                                 FrontierRec *__restrict__ out) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  FrontierRec f;
  f.target_id = ranges[q].target_id;
  f.start = ranges[q].start;
  f.end = ranges[q].end;
  f.qidx = q;
  out[q] = f;
}

// ---------------------------------------------------------------------------
// MultiImpg::query_all_indices (multi_impg.rs:556-592): the hits of one step are
// merged over the per-file indices, hits equal to the self interval are dropped,
// and the rest is sorted by (query_id, q.first, q.last, t.first, t.last).  Merging
// then sorting makes the per-file split irrelevant: sort every range's slot run.
// One wave per range; dest = number of slots that sort before (ties by slot).
// ---------------------------------------------------------------------------
// ties on all five keys keep the reference's concatenation order: alignment file by alignment file, each in
// its own tree's visit order (am / bm = the entries' mrank); the slot index only separates empty slots
__device__ __forceinline__ bool key5_less(uint32_t aq, int32_t a1, int32_t a2, int32_t a3, int32_t a4, uint32_t am, uint32_t ai,
                                          uint32_t bq, int32_t b1, int32_t b2, int32_t b3, int32_t b4, uint32_t bm, uint32_t bi) {
  if (aq != bq) return aq < bq;
  if (a1 != b1) return a1 < b1;
  if (a2 != b2) return a2 < b2;
  if (a3 != b3) return a3 < b3;
  if (a4 != b4) return a4 < b4;
  if (am != bm) return am < bm;
  return ai < bi;
}
__global__ __launch_bounds__(256) void sort5_kernel(const FrontierRec *__restrict__ fr, uint32_t n,
                                                    const uint32_t *__restrict__ pair_off, uint32_t n_pairs,
                                                    HitArrays h, const uint32_t *__restrict__ pair_entry,
                                                    const uint32_t *__restrict__ mrank, uint32_t *__restrict__ dest) {
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * 256u) >> 6;
  const unsigned lane = lane_id();
  for (uint32_t r = wave; r < n; r += nwaves) {
    const uint32_t a = pair_off[r], b = r + 1 < n ? pair_off[r + 1] : n_pairs;
    const FrontierRec f = fr[r];
    for (uint32_t base = a; base < b; base += 64u) {
      const uint32_t i = base + lane;
      uint32_t q = HIT_NONE, km = 0;
      int32_t k1 = 0, k2 = 0, k3 = 0, k4 = 0;
      if (i < b) {
        q = h.qid[i];
        if (q != HIT_NONE) {
          const int4 hc = h.c[i];
          k1 = hc.x; k2 = hc.y; k3 = hc.z; k4 = hc.w;
          km = mrank[pair_entry[i]];
          // is_self (multi_impg.rs:558-562): equal to the step's own interval -> dropped
          if (q == f.target_id && k1 == f.start && k2 == f.end) { q = HIT_NONE; h.qid[i] = HIT_NONE; k1 = k2 = k3 = k4 = 0; km = 0; }
        }
      }
      uint32_t pos = 0;
      for (uint32_t b2 = a; b2 < b; b2 += 64u) {
        const uint32_t i2 = b2 + lane;
        uint32_t q2 = HIT_NONE, jm = 0;
        int32_t j1 = 0, j2 = 0, j3 = 0, j4 = 0;
        if (i2 < b) {
          q2 = h.qid[i2];
          if (q2 != HIT_NONE) {
            const int4 hc = h.c[i2];
            j1 = hc.x; j2 = hc.y; j3 = hc.z; j4 = hc.w;
            jm = mrank[pair_entry[i2]];
            if (q2 == f.target_id && j1 == f.start && j2 == f.end) { q2 = HIT_NONE; j1 = j2 = j3 = j4 = 0; jm = 0; }
          }
        }
        const uint32_t cntl = min(64u, b - b2);
        for (uint32_t t = 0; t < cntl; t++) {
          const uint32_t oq = (uint32_t)__shfl((int)q2, (int)t);
          const int32_t o1 = __shfl(j1, (int)t), o2 = __shfl(j2, (int)t), o3 = __shfl(j3, (int)t), o4 = __shfl(j4, (int)t);
          const uint32_t om = (uint32_t)__shfl((int)jm, (int)t);
          pos += key5_less(oq, o1, o2, o3, o4, om, b2 + t, q, k1, k2, k3, k4, km, i) ? 1u : 0u;
        }
      }
      if (i < b) dest[i] = a + pos;  // empty slots (0xFFFFFFFF) sort last
    }
  }
}
// out[dest[p]] = in[p] for every per-slot column
__global__ __launch_bounds__(256) void permute_slots_kernel(const uint32_t *__restrict__ dest, uint32_t n_pairs, HitArrays in,
                                                            HitArrays out, const uint32_t *__restrict__ pe_in,
                                                            uint32_t *__restrict__ pe_out, SliceArrays sin, SliceArrays sout) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t d = dest[p];
  out.qid[d] = in.qid[p]; out.c[d] = in.c[p];
  pe_out[d] = pe_in[p];
  if (sin.a) { sout.a[d] = sin.a[p]; sout.n[d] = sin.n[p]; sout.off[d] = sin.off[p]; sout.rem[d] = sin.rem[p]; }
}

// ---------------------------------------------------------------------------
// DFS (query_transitive_dfs, impg.rs:2057-2309): every query keeps a stack of
// (sequence, start, end, depth) sorted by (sequence, start); one round pops the
// last element of every query's stack.  Stacks of all queries live in one flat
// array sorted by (qidx, sequence, start).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frontier_to_stack_kernel(const FrontierRec *__restrict__ fr, uint32_t n,
                                                                const uint32_t *__restrict__ pop_depth, int use_depth,
                                                                unsigned long long *__restrict__ key,
                                                                int32_t *__restrict__ st, int32_t *__restrict__ en,
                                                                uint32_t *__restrict__ depth) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const FrontierRec f = fr[i];
  key[i] = ((unsigned long long)f.qidx << 32) | f.target_id;
  st[i] = f.start;
  en[i] = f.end;
  depth[i] = use_depth ? pop_depth[f.qidx] + 1u : 0u;  // impg.rs:2277 pushes current_depth + 1
}
// A round's pop (impg.rs:2117-2127; MultiImpg's FIFO worklist pops the first record, multi_impg.rs:849-854).  The record at the
// query's end of the stack is popped; one that is too deep to be explored is dropped without anything else happening (the
// reference's `continue` comes before the re-sort), and so is the next -- so a round takes every too-deep record off that end
// and explores the first one that is not (round 6: one such record a round made a lone MultiImpg -m 3 call 20 000 rounds of
// which 834 explored anything, 1.26 s).  pop_sel[q]: back = 1 + the index of the query's LAST explorable record (0: none),
// front = the index of its FIRST (0xFFFFFFFF: none) -- set by dfs_pop_select_kernel, the launcher clears it.
__global__ __launch_bounds__(256) void dfs_pop_select_kernel(const unsigned long long *__restrict__ key, const uint32_t *__restrict__ depth, uint32_t n,
                                                             uint32_t max_depth, int pop_front, uint32_t *__restrict__ pop_sel) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  if (max_depth > 0 && depth[i] >= max_depth) return;
  const uint32_t q = (uint32_t)(key[i] >> 32);
  if (pop_front) atomicMin(&pop_sel[q], i); else atomicMax(&pop_sel[q], i + 1u);
}
__global__ __launch_bounds__(256) void dfs_pop_flags_kernel(const unsigned long long *__restrict__ key,
                                                            const uint32_t *__restrict__ depth, uint32_t n, int pop_front,
                                                            const uint32_t *__restrict__ pop_sel,
                                                            uint32_t *__restrict__ fr_flag,
                                                            uint32_t *__restrict__ keep_flag,
                                                            uint32_t *__restrict__ pop_depth) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t q = (uint32_t)(key[i] >> 32);
  const uint32_t sel = pop_sel[q];
  // (no explorable record: the whole run is popped and dropped)
  const bool have = pop_front ? sel != 0xFFFFFFFFu : sel != 0u;
  const uint32_t x = pop_front ? sel : sel - 1u;  // the explored record
  const bool explored = have && i == x;
  keep_flag[i] = have && (pop_front ? i > x : i < x) ? 1u : 0u;
  fr_flag[i] = explored ? 1u : 0u;
  if (explored) pop_depth[q] = depth[i];
}
__global__ __launch_bounds__(256) void dfs_pop_scatter_kernel(const unsigned long long *__restrict__ key,
                                                              const int32_t *__restrict__ st, const int32_t *__restrict__ en,
                                                              const uint32_t *__restrict__ depth, uint32_t n,
                                                              const uint32_t *__restrict__ fr_flag,
                                                              const uint32_t *__restrict__ fr_pos,
                                                              const uint32_t *__restrict__ keep_flag,
                                                              const uint32_t *__restrict__ keep_pos,
                                                              FrontierRec *__restrict__ fr_out,
                                                              unsigned long long *__restrict__ key_out,
                                                              int32_t *__restrict__ st_out, int32_t *__restrict__ en_out,
                                                              uint32_t *__restrict__ depth_out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  if (fr_flag[i]) {
    FrontierRec f;
    f.target_id = (uint32_t)(key[i] & 0xFFFFFFFFull);
    f.start = st[i];
    f.end = en[i];
    f.qidx = (uint32_t)(key[i] >> 32);
    fr_out[fr_pos[i]] = f;
  }
  if (keep_flag[i]) {
    const uint32_t p = keep_pos[i];
    key_out[p] = key[i];
    st_out[p] = st[i];
    en_out[p] = en[i];
    depth_out[p] = depth[i];
  }
}
__global__ __launch_bounds__(256) void iota_kernel(uint32_t *__restrict__ v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                                         uint32_t n, uint32_t *__restrict__ dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void gather_u64_kernel(const unsigned long long *__restrict__ src,
                                                         const uint32_t *__restrict__ idx, uint32_t n,
                                                         unsigned long long *__restrict__ dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// head of each run of equal keys (all keys real)
__global__ __launch_bounds__(256) void run_heads_kernel(const unsigned long long *__restrict__ skeys, uint32_t n,
                                                        uint32_t *__restrict__ head) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || skeys[i - 1] != skeys[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void run_starts_kernel(const uint32_t *__restrict__ head, const uint32_t *__restrict__ gid,
                                                         uint32_t n, uint32_t *__restrict__ gstart) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n && head[i]) gstart[gid[i]] = i;
}
// stack merge sweep (impg.rs:2291-2304) inside one (query, sequence) run, records
// gathered through perm[]; merged records are written at the run's own offset
__global__ __launch_bounds__(64) void dfs_merge_kernel(const uint32_t *__restrict__ gstart, uint32_t n_groups, uint32_t n,
                                                       const uint32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ st, const int32_t *__restrict__ en,
                                                       const uint32_t *__restrict__ depth,
                                                       int32_t *__restrict__ st_m, int32_t *__restrict__ en_m,
                                                       uint32_t *__restrict__ depth_m, uint32_t *__restrict__ cnt) {
  const uint32_t g = blockIdx.x * 64u + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t a = gstart[g], b = g + 1 < n_groups ? gstart[g + 1] : n;
  uint32_t w = a;
  uint32_t p0 = perm[a];
  int32_t ws = st[p0], we = en[p0];
  uint32_t wd = depth[p0];
  for (uint32_t r = a + 1; r < b; r++) {
    const uint32_t pr = perm[r];
    const int32_t rs = st[pr], re = en[pr];
    if (we >= rs) {
      we = max(we, re);  // merged element keeps the first one's depth
    } else {
      st_m[w] = ws; en_m[w] = we; depth_m[w] = wd;
      w++;
      ws = rs; we = re; wd = depth[pr];
    }
  }
  st_m[w] = ws; en_m[w] = we; depth_m[w] = wd;
  cnt[g] = w - a + 1;
}
__global__ __launch_bounds__(256) void dfs_compact_kernel(const uint32_t *__restrict__ gstart,
                                                          const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off,
                                                          uint32_t n_groups, const unsigned long long *__restrict__ skeys,
                                                          const int32_t *__restrict__ st_m, const int32_t *__restrict__ en_m,
                                                          const uint32_t *__restrict__ depth_m,
                                                          unsigned long long *__restrict__ key_out,
                                                          int32_t *__restrict__ st_out, int32_t *__restrict__ en_out,
                                                          uint32_t *__restrict__ depth_out) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t a = gstart[g], c = cnt[g], o = off[g];
  const unsigned long long k = skeys[a];
  for (uint32_t i = 0; i < c; i++) {
    key_out[o + i] = k;
    st_out[o + i] = st_m[a + i];
    en_out[o + i] = en_m[a + i];
    depth_out[o + i] = depth_m[a + i];
  }
}

// ---------------------------------------------------------------------------
// visited-table compaction: fold all tables into one (newest entry of a key wins)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_fill_kernel(const unsigned long long *__restrict__ keys, uint32_t n,
                                                           uint32_t table, unsigned long long *__restrict__ key_out,
                                                           unsigned long long *__restrict__ src_out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  key_out[i] = keys[i];
  src_out[i] = ((unsigned long long)table << 32) | i;
}
__global__ __launch_bounds__(256) void compact_last_kernel(const unsigned long long *__restrict__ skeys, uint32_t n,
                                                           uint32_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i + 1 == n || skeys[i + 1] != skeys[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void compact_select_kernel(VisitedTables vt, const unsigned long long *__restrict__ skeys,
                                                             const unsigned long long *__restrict__ ssrc, uint32_t n,
                                                             const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                                             unsigned long long *__restrict__ key_out,
                                                             unsigned long long *__restrict__ src_out,
                                                             uint32_t *__restrict__ len_out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t p = pos[i];
  const unsigned long long s = ssrc[i];
  key_out[p] = skeys[i];
  src_out[p] = s;
  len_out[p] = vt.t[(uint32_t)(s >> 32)].len[(uint32_t)(s & 0xFFFFFFFFull)];
}
__global__ __launch_bounds__(256) void compact_copy_kernel(VisitedTables vt, const unsigned long long *__restrict__ src,
                                                           const uint32_t *__restrict__ off, const uint32_t *__restrict__ len,
                                                           uint32_t n, int2 *__restrict__ ranges_out) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n) return;
  const unsigned long long s = src[g];
  const VisitedTable &T = vt.t[(uint32_t)(s >> 32)];
  const int2 *in = T.ranges + T.off[(uint32_t)(s & 0xFFFFFFFFull)];
  int2 *out = ranges_out + off[g];
  const uint32_t l = len[g];
  for (uint32_t i = 0; i < l; i++) out[i] = in[i];
}

// ---------------------------------------------------------------------------
// Hits between ranks (sharded.cpp).  An owner packs the slots of its expansion into AoS records addressed to the
// record's HOME: word 0 = the home's frontier index (the qidx the routed record carried), then query id and the
// query interval -- all the visited-set update reads (4 words) -- and, for runs that keep result rows or sort
// them (MultiImpg), the target interval and the entry's MultiImpg tie rank (8 words).  Home puts the blocks that
// arrive per owner back into frontier order (a frontier record has one owner: runs never interleave) and
// unpacks them into the slot arrays a local expansion would have filled.
// ---------------------------------------------------------------------------
template <int WORDS>
__global__ __launch_bounds__(256) void hits_pack_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ pair_range,
                                                        uint32_t n_pairs, HitArrays h, const uint32_t *__restrict__ pair_entry,
                                                        const uint32_t *__restrict__ mrank, const uint32_t *__restrict__ slice_n,
                                                        uint4 *__restrict__ out) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t q = h.qid[p];
  const int4 hc = q != HIT_NONE ? h.c[p] : make_int4(0, 0, 0, 0);
  const uint32_t home = fr[pair_range[p]].qidx;
  if (WORDS == 4) {
    out[p] = make_uint4(home, q, (uint32_t)hc.x, (uint32_t)hc.y);
  } else {
    out[2 * (size_t)p] = make_uint4(home, q, (uint32_t)hc.x, (uint32_t)hc.y);
    // word 7, store_cigar: the number of ops of the hit's CIGAR slice (the ops travel in a second exchange, in this order)
    out[2 * (size_t)p + 1] = make_uint4((uint32_t)hc.z, (uint32_t)hc.w, (pair_entry && mrank) ? mrank[pair_entry[p]] : 0u,
                                        (slice_n && q != HIT_NONE) ? slice_n[p] : 0u);
  }
}
// run_start == nullptr: the records are already in frontier order (one owner)
template <int WORDS>
__global__ __launch_bounds__(256) void hits_unpack_kernel(const uint4 *__restrict__ in, uint32_t n, uint32_t n_front,
                                                          const uint32_t *__restrict__ run_start, const uint32_t *__restrict__ off,
                                                          uint32_t *__restrict__ pair_range, HitArrays h,
                                                          uint32_t *__restrict__ mslot, const uint32_t *__restrict__ slice_at,
                                                          uint32_t *__restrict__ slice_pos, uint32_t *__restrict__ slice_n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  constexpr int Q = WORDS / 4;
  const uint4 a = in[(size_t)i * Q];
  const uint32_t f = a.x;
  if (f >= n_front) return;
  const size_t d = run_start ? (size_t)off[f] + (i - run_start[f]) : i;
  pair_range[d] = f;
  h.qid[d] = a.y;
  if (WORDS == 4) {
    h.c[d] = make_int4((int32_t)a.z, (int32_t)a.w, 0, 0);
  } else {
    const uint4 b = in[(size_t)i * Q + 1];
    h.c[d] = make_int4((int32_t)a.z, (int32_t)a.w, (int32_t)b.x, (int32_t)b.y);
    if (mslot) mslot[d] = b.z;
    if (slice_pos) { slice_pos[d] = slice_at[i]; slice_n[d] = b.w; }  // (the ops arrived in the hits' arrival order)
  }
}
// store_cigar over a sharded index: word 7 of every arrived hit record = the number of ops of its slice
__global__ __launch_bounds__(256) void hits_slice_n_kernel(const uint4 *__restrict__ in, uint32_t n, uint32_t *__restrict__ cnt) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) cnt[i] = in[2 * (size_t)i + 1].w;
}

// ---------------------------------------------------------------------------
// Small batches (the trait's per-call shape: one range, a handful of ranges): the whole Impg::query of the batch is
// one chain of launches with no host round trip in between -- the pair count stays on the device (single-block
// scan, projection grid sized for a host-known upper bound) and the results are written straight into
// host-mapped memory, so the call costs one synchronisation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void small_scan_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ off,
                                                          uint32_t *__restrict__ total) {
  __shared__ uint32_t s[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t v = t < n ? cnt[t] : 0u;
  s[t] = v;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t x = t >= d ? s[t - d] : 0u;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  if (t < n) off[t] = s[t] - v;
  if (t == 1023) *total = s[1023];
}
struct SmallHeader {  // first 64 bytes of the mapped result buffer
  uint32_t n_pairs, err, pad0, pad1;
  unsigned long long accepted, pad2;
  uint32_t pad3[8];
};
__global__ __launch_bounds__(256) void small_pack_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ pair_range,
                                                         const uint32_t *__restrict__ n_pairs_dev, HitArrays h,
                                                         const uint32_t *__restrict__ err_flag, const unsigned long long *__restrict__ accepted,
                                                         SmallHeader *__restrict__ hdr, impg_gpu_interval_t *__restrict__ rows,
                                                         uint32_t *__restrict__ row_range) {
  const uint32_t np = *n_pairs_dev;
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p == 0) {
    unsigned long long acc = 0;
    for (uint32_t k = 0; k < COUNT_SLOTS; k++) acc += accepted[k * COUNT_STRIDE];
    hdr->n_pairs = np; hdr->err = *err_flag; hdr->accepted = acc;
  }
  if (p >= np) return;
  const uint32_t r = pair_range[p];
  const FrontierRec f = fr[r];
  const uint32_t qid = h.qid[p];
  impg_gpu_interval_t x;
  x.query_id = qid;
  const int4 c = qid != HIT_NONE ? h.c[p] : make_int4(0, 0, 0, 0);
  x.q_first = c.x; x.q_last = c.y; x.target_id = f.target_id; x.t_first = c.z; x.t_last = c.w;
  rows[p] = x;
  row_range[p] = f.qidx;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
// windows of <= 64 entries are emitted lane-per-range (needs ranks < 2^26 for the packed sort key: a rank is a
// position within its target's segment, so the largest segment decides, not the index)
bool emit_by_lanes(const DeviceIndexView &v) { return v.max_seg < (1u << 26); }
static inline uint32_t wave_grid(uint32_t n_items) {  // one wave per item, 4 waves per block, capped
  uint32_t blocks = cdiv(n_items, 4);
  const uint32_t cap = 256u * 32u;  // 32 blocks per CU worth of grid-stride
  return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

void launch_lookup_count(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, bool transitive, const uint32_t *perm,
                         uint32_t *cnt, uint4 *win, uint32_t *wide_n, uint32_t *wide_list, hipStream_t s, bool by_place, FrontierRec *se,
                         uint32_t *cnt_ref) {
  if (!n) return;
  const bool lanes = emit_by_lanes(v);
  const int bp = by_place && perm ? 1 : 0;
  if (!bp) se = nullptr;
  if (lanes) IMPG_HIP(hipMemsetAsync(wide_n, 0, 8, s));  // (the wide list's length, and its overflow list's: launch_lookup_emit)
  if (transitive) lookup_count_lane_kernel<true><<<cdiv(n, 256), 256, 0, s>>>(v, fr, n, perm, cnt, win, wide_n, lanes ? wide_list : nullptr, bp, se, cnt_ref);
  else lookup_count_lane_kernel<false><<<cdiv(n, 256), 256, 0, s>>>(v, fr, n, perm, cnt, win, wide_n, lanes ? wide_list : nullptr, bp, se, cnt_ref);
  if (lanes) {  // the listed (wide) windows' counts; the list's length stays on the device: a grid of at most 8 192 waves strides over it
    const uint32_t g = std::min(cdiv(n, 4u), 2048u);
    if (transitive) lookup_count_wide_kernel<true><<<g, 256, 0, s>>>(v, fr, perm, wide_n, wide_list, win, bp, cnt, cnt_ref);
    else lookup_count_wide_kernel<false><<<g, 256, 0, s>>>(v, fr, perm, wide_n, wide_list, win, bp, cnt, cnt_ref);
  }
}
// tile_first[t] = the range whose places include place t * PROJ_BLOCK (see WindowLists): one thread per range, which
// names itself at every tile border inside its run of places (a run of <= 64 places crosses at most one)
__global__ __launch_bounds__(256) void tile_first_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off, uint32_t n,
                                                         uint32_t *__restrict__ tile_first) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = cnt[i];
  if (!c) return;
  const uint32_t first = pair_off[i], last = first + c - 1u;
  for (uint32_t t = (first + PROJ_BLOCK - 1u) / PROJ_BLOCK; t <= last / PROJ_BLOCK; t++) tile_first[t] = i;
}
// ---------------------------------------------------------------------------
// Ordered rows placed slot by slot (Engine::ordered_*, engine.cpp): the batch's rows grouped by range in the reference's
// emission order -- self interval, then level by level, each level's slots in frontier order x visit order
// (impg.rs:1864-1925, :2345-2363, :2471-2504) -- with every SLOT at its final place: a slot whose projection returned
// None (or whose row is shorter than min_output_length) stays as a hole row, query_id = 0xFFFFFFFF, instead of
// shifting every row behind it.  Where a row goes is then known from the lookups' counts alone, before anything is
// projected: row(slot k of record r of level l, query q) = offsets[q] + lvbase_l[q] + slot_ref_l[r] + k, with
//   slot_ref_l = exclusive scan of the level's per-record slot counts in frontier order,
//   acc[q]     = rows of q so far (self interval, earlier levels),  lvbase_l[q] = acc[q] - slot_ref_l[q's first record],
//   offsets    = exclusive scan of the final acc.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ord_self_count_kernel(const FrontierRec *__restrict__ self, const impg_gpu_range_t *__restrict__ ranges,
                                                             uint32_t n, uint32_t *__restrict__ acc) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  acc[q] = self ? (self[q].start < self[q].end ? 1u : 0u) : 1u;  // impg.rs:2345-2363 / :1864-1880
  (void)ranges;
}
__global__ __launch_bounds__(256) void ord_level_bases_kernel(const FrontierRec *__restrict__ fr, uint32_t n_fr, const uint32_t *__restrict__ slot_ref,
                                                              uint32_t total, uint32_t n_queries, uint32_t *__restrict__ acc,
                                                              uint32_t *__restrict__ lvbase) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n_queries) return;
  // the query's records: [first record with qidx >= q, first with qidx >= q + 1) of a frontier sorted by query
  uint32_t lo = 0, hi = n_fr;
  while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (fr[m].qidx < q) lo = m + 1u; else hi = m; }
  const uint32_t a = lo;
  hi = n_fr;
  while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (fr[m].qidx <= q) lo = m + 1u; else hi = m; }
  const uint32_t sa = a < n_fr ? slot_ref[a] : total, sb = lo < n_fr ? slot_ref[lo] : total;
  const uint32_t at = acc[q];
  lvbase[q] = at - sa;  // (mod 2^32: row = offsets[q] + lvbase[q] + slot_ref[r] + k)
  acc[q] = at + (sb - sa);
}
__global__ __launch_bounds__(256) void ord_self_rows_kernel(const FrontierRec *__restrict__ self, const impg_gpu_range_t *__restrict__ ranges, uint32_t n,
                                                            const uint32_t *__restrict__ offsets, impg_gpu_interval_t *__restrict__ rows) {
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= n) return;
  uint32_t t; int32_t a, b;
  if (self) { const FrontierRec f = self[q]; if (!(f.start < f.end)) return; t = f.target_id; a = f.start; b = f.end; }
  else { const impg_gpu_range_t r = ranges[q]; t = r.target_id; a = r.start; b = r.end; }
  rows[offsets[q]] = impg_gpu_interval_t{t, a, b, t, a, b};
}
__global__ __launch_bounds__(256) void ord_run_heads_kernel(const uint32_t *__restrict__ pair_range, uint32_t n_pairs, uint32_t *__restrict__ run_start) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t r = pair_range[p];
  if (p == 0 || pair_range[p - 1] != r) run_start[r] = p;  // (a record's slots are one run in every layout of a listed level)
}
__global__ __launch_bounds__(256) void ord_level_rows_kernel(const FrontierRec *__restrict__ fr, const uint32_t *__restrict__ pair_range, uint32_t n_pairs,
                                                             HitArrays h, const uint32_t *__restrict__ run_start, const uint32_t *__restrict__ slot_ref,
                                                             const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ lvbase,
                                                             int32_t min_output_length, impg_gpu_interval_t *__restrict__ rows) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t r = pair_range[p];
  const FrontierRec f = fr[r];
  const uint32_t d = offsets[f.qidx] + lvbase[f.qidx] + slot_ref[r] + (p - run_start[r]);
  const uint32_t qid = h.qid[p];
  const bool ok = qid != HIT_NONE;
  int4 c = make_int4(0, 0, 0, 0);
  if (ok) c = h.c[p];
  OrderedOut o{rows, nullptr, nullptr, min_output_length};
  put_ordered_row(o, d, qid, f.target_id, ok, c.x, c.y, c.z, c.w);
}
void launch_ord_self_count(const FrontierRec *self, const impg_gpu_range_t *ranges, uint32_t n, uint32_t *acc, hipStream_t s) {
  if (n) ord_self_count_kernel<<<cdiv(n, 256), 256, 0, s>>>(self, ranges, n, acc);
}
void launch_ord_level_bases(const FrontierRec *fr, uint32_t n_fr, const uint32_t *slot_ref, uint32_t total, uint32_t n_queries, uint32_t *acc,
                            uint32_t *lvbase, hipStream_t s) {
  if (n_queries) ord_level_bases_kernel<<<cdiv(n_queries, 256), 256, 0, s>>>(fr, n_fr, slot_ref, total, n_queries, acc, lvbase);
}
void launch_ord_self_rows(const FrontierRec *self, const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *offsets, impg_gpu_interval_t *rows,
                          hipStream_t s) {
  if (n) ord_self_rows_kernel<<<cdiv(n, 256), 256, 0, s>>>(self, ranges, n, offsets, rows);
}
void launch_ord_run_heads(const uint32_t *pair_range, uint32_t n_pairs, uint32_t *run_start, hipStream_t s) {
  if (n_pairs) ord_run_heads_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(pair_range, n_pairs, run_start);
}
void launch_ord_level_rows(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h, const uint32_t *run_start,
                           const uint32_t *slot_ref, const uint32_t *offsets, const uint32_t *lvbase, int32_t min_output_length,
                           impg_gpu_interval_t *rows, hipStream_t s) {
  if (n_pairs) ord_level_rows_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, h, run_start, slot_ref, offsets, lvbase, min_output_length, rows);
}
void launch_emit_vpos(const DeviceIndexView &v, uint32_t n, const uint32_t *pair_off, const uint4 *win, uint8_t *vpos, hipStream_t s, bool by_visit,
                      const OrdDestArgs *dest) {
  if (!n) return;
  const OrdDestArgs od = dest ? *dest : OrdDestArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (by_visit) emit_vpos_lane_kernel<true><<<cdiv(n, 64), 64, 0, s>>>(v, n, pair_off, win, vpos, od);
  else emit_vpos_lane_kernel<false><<<cdiv(n, 64), 64, 0, s>>>(v, n, pair_off, win, vpos, od);
}
// (IMPG_ORD_ENTRIES=1: ordered rows from the entry-major kernel, for the comparison -- see launch_project)
bool ordered_rows_by_visit() {
  static const bool by_entries = getenv("IMPG_ORD_ENTRIES") && atoi(getenv("IMPG_ORD_ENTRIES")) != 0;
  return !by_entries;
}
void launch_tile_first(const uint32_t *cnt, const uint32_t *pair_off, uint32_t n_fr, uint32_t *tile_first, hipStream_t s) {
  if (n_fr) tile_first_kernel<<<cdiv(n_fr, 256), 256, 0, s>>>(cnt, pair_off, n_fr, tile_first);
}
void launch_lookup_emit(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, bool transitive,
                        const uint32_t *pair_off, const uint4 *win, uint32_t *pair_range, uint32_t *pair_entry,
                        const uint32_t *perm, const uint32_t *offp, ProjList pl, const uint32_t *wide_n,
                        const uint32_t *wide_list, hipStream_t s, bool by_place, bool wide_only) {
  if (!n) return;
  const int bp = by_place && perm ? 1 : 0;
  const uint32_t *pp = bp ? perm : nullptr;
  // windows of <= 64 entries: lane per range; the rest (dense targets), or everything if a rank could
  // overflow the packed sort key: wave per range
  const bool lanes = emit_by_lanes(v);
  if (lanes && !wide_only) {
    if (transitive) lookup_emit_lane_kernel<true><<<cdiv(n, 64), 64, 0, s>>>(v, n, pair_off, win, pair_range, pair_entry, perm, offp, pl, bp);
    else lookup_emit_lane_kernel<false><<<cdiv(n, 64), 64, 0, s>>>(v, n, pair_off, win, pair_range, pair_entry, perm, offp, pl, bp);
  }
  if (lanes) {
    // the listed wide windows (the list is normally short or empty): a block per range sorts the window's hits by visit rank in
    // LDS; a range with more hits than its buffer takes goes onto the overflow list (behind the wide list's n entries; its
    // count is wide_n[1], zeroed with wide_n[0] by the count pass) for the wave-per-range kernel
    uint32_t *over_list = const_cast<uint32_t *>(wide_list) + n, *over_n = const_cast<uint32_t *>(wide_n) + 1;
    const uint32_t gw = std::min(n, 4096u);
    if (transitive) lookup_emit_wide_kernel<true><<<gw, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, wide_list, wide_n, pp, over_list, over_n);
    else lookup_emit_wide_kernel<false><<<gw, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, wide_list, wide_n, pp, over_list, over_n);
    const uint32_t g = std::min(wave_grid(n), 256u);
    if (transitive) lookup_emit_kernel<true><<<g, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, over_list, over_n, pp);
    else lookup_emit_kernel<false><<<g, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, over_list, over_n, pp);
    return;
  }
  // everything wave per range (ranks too large for the lane kernel's packed keys)
  const uint32_t g = wave_grid(n);
  if (transitive) lookup_emit_kernel<true><<<g, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, nullptr, nullptr, pp);
  else lookup_emit_kernel<false><<<g, 256, 0, s>>>(v, fr, n, pair_off, win, pair_range, pair_entry, offp, pl, nullptr, nullptr, pp);
}
void launch_small_scan(const uint32_t *cnt, uint32_t n, uint32_t *off, uint32_t *total, hipStream_t s) {
  small_scan_kernel<<<1, 1024, 0, s>>>(cnt, n, off, total);
}
void launch_small_pack(const FrontierRec *fr, const uint32_t *pair_range, const uint32_t *n_pairs_dev, uint32_t pairs_bound, HitArrays h,
                       const uint32_t *err_flag, const unsigned long long *accepted, void *hdr, impg_gpu_interval_t *rows,
                       uint32_t *row_range, hipStream_t s) {
  small_pack_kernel<<<std::max(1u, cdiv(pairs_bound, 256)), 256, 0, s>>>(fr, pair_range, n_pairs_dev, h, err_flag, accepted,
                                                                         static_cast<SmallHeader *>(hdr), rows, row_range);
}
void launch_route_keys(const FrontierRec *fr, uint32_t n, uint32_t world, const uint32_t *owner, uint32_t n_seq, uint32_t *key,
                       uint32_t *idx, unsigned long long *hist, hipStream_t s) {
  if (n) route_keys_kernel<<<std::min(cdiv(n, 256), 2048u), 256, 0, s>>>(fr, n, world, owner, n_seq, key, idx, hist);
}
void launch_route_gather(const FrontierRec *fr, const uint32_t *perm, uint32_t n, FrontierRec *out, hipStream_t s) {
  if (n) route_gather_kernel<<<cdiv(n, 256), 256, 0, s>>>(fr, perm, n, out);
}
void launch_frontier_gather(const FrontierRec *fr, const uint32_t *perm, uint32_t n, FrontierRec *out, hipStream_t s) {
  if (n) frontier_gather_kernel<<<cdiv(n, 256), 256, 0, s>>>(fr, perm, n, out);
}
void launch_reorder_runs(const uint32_t *hits, uint32_t n, uint32_t words, uint32_t n_front, uint32_t *run_start,
                         uint32_t *run_len, uint32_t *err, hipStream_t s) {
  if (!n) return;
  reorder_runs_kernel<<<cdiv(n, 256), 256, 0, s>>>(hits, n, words, n_front, run_start, run_len, err);
  run_lengths_kernel<<<cdiv(n_front, 256), 256, 0, s>>>(run_start, run_len, n_front);
}
void launch_order_keys(const DeviceIndexView &v, const FrontierRec *fr, uint32_t n, uint32_t *key, uint32_t *idx, hipStream_t s,
                       const uint32_t *bounds, uint32_t n_blocks, uint32_t block_shift) {
  if (n) order_keys_kernel<<<cdiv(n, 256), 256, 0, s>>>(v, fr, n, key, idx, bounds, n_blocks, block_shift);
}
void launch_scatter_u32(const uint32_t *in, const uint32_t *perm, uint32_t n, uint32_t *out, hipStream_t s) {
  if (n) scatter_u32_kernel<<<cdiv(n, 256), 256, 0, s>>>(in, perm, n, out);
}
static double stage_density() {  // IMPG_STAGE_DENSITY = pairs per index entry from which on a level is "dense" (default 32; 0 = always, < 0 = never)
  static const double d = [] { const char *e = getenv("IMPG_STAGE_DENSITY"); return e ? atof(e) : 32.0; }();
  return d;
}
// ... and for a level whose pairs are LISTED (a dense level that is not a counting run's last: project_staged_kernel): default 128.
// Round 5, measured: the headline's middle level has 39 pairs per entry, and a block's 256 ranges span ~230 entries of which 56
// are staged -- the lane-per-pair kernel on the index takes it 0.37 ms faster (projection 18.5 -> 18.1 ms), and config 5's levels
// between 32 and 128 likewise (4 000 windows: projection 236 -> 230 ms; 256: 233).  IMPG_STAGE_DENSITY, when set, rules both.
static double stage_density_listed() {
  static const double d = [] {
    const char *e = getenv("IMPG_STAGE_DENSITY");
    if (e) return atof(e);
    const char *l = getenv("IMPG_STAGE_DENSITY_LISTED");
    return l ? atof(l) : 128.0;
  }();
  return d;
}
static bool entry_major() {  // (A/B: IMPG_ENTRY_MAJOR=0 keeps a lane per place on the final level too)
  static const bool b = [] { const char *e = getenv("IMPG_ENTRY_MAJOR"); return !e || atoi(e) != 0; }();
  return b;
}
// A plain projection of n_pairs pairs whose slots follow the lookup order runs on the staged kernels (which find a
// place's range in the block's own LDS offsets: no tile_first[]) when the level is dense.
bool project_is_staged(const DeviceIndexView &v, uint64_t n_pairs, bool plain) {
  return plain && v.pfx && !v.tp_mode && stage_density() >= 0.0 && (double)n_pairs >= stage_density() * (double)v.n_entries;
}
// ... and a level named by the lookup's hit masks (a fused final level) entry by entry on project_entries_kernel: the
// plain projection, or the identity filter on an index that holds its identity lines
bool project_entry_major(const DeviceIndexView &v, uint64_t n_pairs, double min_identity) {
  const bool ident = min_identity == min_identity;
  return project_is_staged(v, n_pairs, true) && entry_major() && (!ident || v.idp != nullptr);
}
uint32_t project_entry_blocks(uint32_t n_fr) { return (cdiv(n_fr, ENT_RANGES) + 7u) & ~7u; }
uint32_t project_entry_slice_cap(uint64_t n_pairs) { return (uint32_t)(n_pairs / ENT_SLICE_PAIRS) + 16u; }
void launch_project(const DeviceIndexView &v, const FrontierRec *fr, const uint32_t *pair_range,
                    const uint32_t *pair_entry, uint32_t n_pairs, bool transitive, HitArrays h,
                    unsigned long long *accepted, uint32_t *err_flag, double min_identity, const SliceArrays *slices,
                    ProjList pl, hipStream_t s, const uint32_t *n_pairs_dev, bool regroup, const WindowLists *wlp) {
  if (!n_pairs) return;
  const WindowLists wl = wlp ? *wlp : WindowLists{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, 0u, 0u};
  const int rg = regroup ? 1 : 0;
  bool ident = min_identity == min_identity;  // NaN = no filter
  // An index built without prefix lines (it would not have fitted the device with them: index_build_device.hip) has
  // no plain path: its projections take the identity filter's two short walks on the op lines with a threshold
  // nothing fails (identity >= 0) -- the same answers, about twice the instructions.
  if (!ident && !v.pfx && !v.tp_mode) { ident = true; min_identity = 0.0; }
  const bool two_walks = !v.pfx && !slices;
  if (v.tp_mode) {  // tracepoint index: every projection is the approximate one
    const uint32_t gt = (cdiv(n_pairs, 256) + 7u) & ~7u;
    if (transitive) project_tp_kernel<true><<<gt, 256, 0, s>>>(v, fr, pair_range, pair_entry, n_pairs, h, accepted, err_flag,
                                                               ident ? min_identity : 0.0, ident ? 1 : 0, pl, n_pairs_dev);
    else project_tp_kernel<false><<<gt, 256, 0, s>>>(v, fr, pair_range, pair_entry, n_pairs, h, accepted, err_flag,
                                                     ident ? min_identity : 0.0, ident ? 1 : 0, pl, n_pairs_dev);
    return;
  }
  const uint32_t g = (cdiv(n_pairs, PROJ_BLOCK) + 7u) & ~7u;  // a multiple of the 8 XCDs (see the block mapping in the kernel)
  const int xcd_map = 1;
  const SliceArrays sl = slices ? *slices : SliceArrays{nullptr, nullptr, nullptr, nullptr};
  const int mode = (ident ? MODE_IDENT : 0) | (slices ? MODE_CIGAR : 0) | (two_walks ? MODE_WALK : 0);
  // A dense level -- many pairs per index entry -- runs with the entries and prefix lines staged in LDS
  // (project_staged_kernel / project_entries_kernel)
  const bool dense = !n_pairs_dev && wl.pair_off && wl.se && !pl.slot && project_is_staged(v, n_pairs, true);
  // Ordered rows come from the lane-per-place kernel: a wave's 64 places lie in one or two ranges, so its rows are one or two
  // stretches of the output.  Entry by entry a wave's lanes are 64 RANGES, whose rows lie in 64 queries' stretches of a 51-GB
  // array -- 64 pages per store instruction: the headline's ordered step 74.5 ms that way, 53.5 ms this way (same box;
  // IMPG_ORD_ENTRIES=1 selects the entry-major kernel for the comparison).
  const bool ord_staged = wl.ord.by_visit != 0;
  if (dense && wl.masks != 0 && wl.ord.rows && ord_staged && mode == 0) {
    const uint32_t gs = (cdiv(wl.n_fr, STG_RANGES) + 7u) & ~7u;
    if (transitive) project_staged_kernel<true, true, OUT_ROWS><<<gs, STG_THREADS, 0, s>>>(v, pair_entry, n_pairs, h, accepted, err_flag, rg, wl);
    else project_staged_kernel<false, true, OUT_ROWS><<<gs, STG_THREADS, 0, s>>>(v, pair_entry, n_pairs, h, accepted, err_flag, rg, wl);
    return;
  }
  if (dense && wl.masks != 0 && entry_major() && (mode == 0 || (mode == MODE_IDENT && v.idp))) {
    // the final level of a counting run, entry by entry; also under the identity filter
    const uint32_t ge = (cdiv(wl.n_fr, ENT_RANGES) + 7u) & ~7u;
    const double mi = ident ? min_identity : 0.0;
    // (what a pair leaves behind is a template constant: the plain form carries none of the other two's registers -- the kernel
    // runs at its scalar-register limit.  The engine asks for rows / {qid, place} pairs only under the plain projection.)
    const int out = wl.ord.rows ? OUT_ROWS : (wl.masks & 4u) ? OUT_QS : OUT_SLOTS;
    if (out != OUT_SLOTS && mode != 0) throw Error{IMPG_E_INVALID, "internal: ordered rows / paired slots under the identity filter"};
    // (heavy blocks are sliced: phase 0 over the range blocks, then phase 1 over the slices phase 0 listed -- a launch of the
    // list's capacity whose blocks leave at once where nothing was listed, the usual case)
    EntSlices es{wl.slice_work, wl.slice_alloc, wl.slice_cap, 0};
    if (!es.work || !es.alloc || !es.cap) es = EntSlices{nullptr, nullptr, 0u, 0};
#define IMPG_LAUNCH_ENT(T, M, O) do { project_entries_kernel<T, M, O><<<ge, ENT_THREADS, 0, s>>>(v, pair_entry, n_pairs, h, accepted, err_flag, rg, wl, mi, es); \
      if (es.work) { EntSlices e1 = es; e1.phase = 1; project_entries_kernel<T, M, O><<<es.cap, ENT_THREADS, 0, s>>>(v, pair_entry, n_pairs, h, accepted, err_flag, rg, wl, mi, e1); } } while (0)
    if (mode == 0) {
      if (out == OUT_ROWS) { if (transitive) IMPG_LAUNCH_ENT(true, 0, OUT_ROWS); else IMPG_LAUNCH_ENT(false, 0, OUT_ROWS); }
      else if (out == OUT_QS) { if (transitive) IMPG_LAUNCH_ENT(true, 0, OUT_QS); else IMPG_LAUNCH_ENT(false, 0, OUT_QS); }
      else { if (transitive) IMPG_LAUNCH_ENT(true, 0, OUT_SLOTS); else IMPG_LAUNCH_ENT(false, 0, OUT_SLOTS); }
    } else {
      if (transitive) IMPG_LAUNCH_ENT(true, MODE_IDENT, OUT_SLOTS); else IMPG_LAUNCH_ENT(false, MODE_IDENT, OUT_SLOTS);
    }
#undef IMPG_LAUNCH_ENT
    return;
  }
  // (a level named by masks has no tile_first[] once the engine has seen it dense: it stays on the staged kernels)
  if (mode == 0 && dense && (wl.masks != 0 || (double)n_pairs >= stage_density_listed() * (double)v.n_entries)) {
    const uint32_t gs = (cdiv(wl.n_fr, STG_RANGES) + 7u) & ~7u;
    const bool masks = wl.masks != 0;
#define IMPG_LAUNCH_STG(T, M) project_staged_kernel<T, M><<<gs, STG_THREADS, 0, s>>>(v, pair_entry, n_pairs, h, accepted, err_flag, rg, wl)
    if (transitive) { if (masks) IMPG_LAUNCH_STG(true, true); else IMPG_LAUNCH_STG(true, false); }
    else { if (masks) IMPG_LAUNCH_STG(false, true); else IMPG_LAUNCH_STG(false, false); }
#undef IMPG_LAUNCH_STG
    return;
  }
#define IMPG_LAUNCH(T, M) project_kernel<T, M><<<g, PROJ_BLOCK, 0, s>>>(v, fr, pair_range, pair_entry, n_pairs, h, accepted, err_flag, ident ? min_identity : 0.0, sl, pl, xcd_map, n_pairs_dev, rg, wl)
  if (transitive) {
    switch (mode) { case 0: IMPG_LAUNCH(true, 0); break; case 1: IMPG_LAUNCH(true, 1); break;
                    case 2: IMPG_LAUNCH(true, 2); break; case 3: IMPG_LAUNCH(true, 3); break; default: IMPG_LAUNCH(true, 5); }
  } else {
    switch (mode) { case 0: IMPG_LAUNCH(false, 0); break; case 1: IMPG_LAUNCH(false, 1); break;
                    case 2: IMPG_LAUNCH(false, 2); break; case 3: IMPG_LAUNCH(false, 3); break; default: IMPG_LAUNCH(false, 5); }
  }
#undef IMPG_LAUNCH
}
void launch_slice_counts(HitArrays h, SliceArrays sl, uint32_t n_pairs, uint32_t *cnt, hipStream_t s) {
  if (n_pairs) slice_counts_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(h, sl, n_pairs, cnt);
}
void launch_slice_write(const DeviceIndexView &v, const uint32_t *pair_entry, HitArrays h, SliceArrays sl, uint32_t n_pairs,
                        const uint32_t *off, uint32_t *out, hipStream_t s) {
  if (n_pairs) slice_write_kernel<<<cdiv(n_pairs, 64), 64, 0, s>>>(v, pair_entry, h, sl, n_pairs, off, out);
}
void launch_hit_stats(const FrontierRec *fr, uint32_t n_fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h,
                      int32_t min_output_length, bool skip_same_target, unsigned long long *rstat, unsigned long long *count,
                      unsigned long long *cksum, hipStream_t s, uint32_t stride) {
  if (!n_pairs || !n_fr) return;
  IMPG_HIP(hipMemsetAsync(rstat, 0, (size_t)n_fr * 16, s));
  hit_stats_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, h, min_output_length, skip_same_target ? 1 : 0, rstat,
                                                     cksum ? 1 : 0, stride);
  range_stats_reduce_kernel<<<cdiv(n_fr, 256), 256, 0, s>>>(fr, n_fr, rstat, count, cksum);
}
void launch_update_keys(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h,
                        unsigned long long *keys, unsigned long long *vals, unsigned long long *n_active, hipStream_t s) {
  if (!n_pairs) return;
  update_keys_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, h, keys, vals, n_active);
}
size_t sort_pairs_scratch_bytes(uint32_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, const unsigned long long *, unsigned long long *,
                                  const uint32_t *, uint32_t *>(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 64,
                                                                (hipStream_t)0);
  return bytes;
}
void launch_sort_pairs(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout,
                       const uint32_t *vin, uint32_t *vout, uint32_t n, unsigned end_bit, hipStream_t s) {
  if (!n) return;
  IMPG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, s));
}
bool seg_group_fits(uint32_t n_seq) { return n_seq <= SEG_MAX_SEQ; }
size_t seg_group_bins_bytes(uint32_t n_queries, uint32_t n_seq, uint32_t parts) {
  return (size_t)n_queries * parts * ((std::max(n_seq, 1u) + 63u) & ~63u) * 4;
}
// how many slices a query is cut into for a level of n_hits slots (1: a wave per query; 0: beyond what the slices' counters may
// take -- the library sort)
uint32_t seg_group_parts(uint64_t n_hits, uint32_t n_queries, uint32_t n_seq, uint64_t largest_query) {
  if (!n_queries) return 1;
  const uint64_t per_query = std::max<uint64_t>(n_hits / n_queries, largest_query);
  if (per_query <= (largest_query ? SEG_BIG_QUERY : SEG_BIG_QUERY / 2)) return 1;
  const uint64_t parts = (per_query + SEG_SLICE_HITS - 1) / SEG_SLICE_HITS;
  if (parts > SEG_MAX_PARTS || (uint64_t)n_queries * parts >= (1ull << 31)) return 0;
  if (seg_group_bins_bytes(n_queries, n_seq, (uint32_t)parts) > (1ull << 30)) return 0;
  return (uint32_t)parts;
}
void launch_seg_bounds(const FrontierRec *fr, uint32_t n_fr, uint32_t n_queries, const uint32_t *pair_range, uint32_t n_pairs, uint32_t *run_start,
                       uint32_t *run_end, uint32_t *qfirst, uint32_t *qlast, uint32_t *unsorted, hipStream_t s, const uint32_t *off_perm,
                       const uint32_t *pair_off, const uint32_t *cnt) {
  IMPG_HIP(hipMemsetAsync(qfirst, 0, (size_t)n_queries * 4, s));
  IMPG_HIP(hipMemsetAsync(qlast, 0, (size_t)n_queries * 4, s));
  IMPG_HIP(hipMemsetAsync(unsorted, 0, 8, s));  // (and the largest query's hit count behind it)
  if (pair_off && cnt) {
    if (n_fr) run_bounds_by_offsets_kernel<<<cdiv(n_fr, 256), 256, 0, s>>>(off_perm, pair_off, cnt, n_fr, run_start, run_end);
  } else {
    IMPG_HIP(hipMemsetAsync(run_start, 0, (size_t)n_fr * 4, s));
    IMPG_HIP(hipMemsetAsync(run_end, 0, (size_t)n_fr * 4, s));
    if (n_pairs) run_bounds_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(pair_range, n_pairs, run_start, run_end);
  }
  if (n_fr) query_bounds_kernel<<<cdiv(n_fr, 256), 256, 0, s>>>(fr, n_fr, n_queries, qfirst, qlast, unsorted);
}
void launch_seg_group(bool count_only, const FrontierRec *fr, const uint32_t *qfirst, const uint32_t *qlast, const uint32_t *run_start,
                      const uint32_t *run_end, HitArrays h, uint32_t n_queries, uint32_t n_seq, uint32_t *qact, const uint32_t *qdst, uint32_t *qgrp,
                      const uint32_t *gdst, uint32_t *gstart, unsigned long long *gkey, unsigned long long *svals, uint32_t *qbins, hipStream_t s,
                      uint32_t parts, uint32_t *qtot) {
  if (!n_queries) return;
  if (parts > 1u && (!qbins || !qtot)) throw Error{IMPG_E_INVALID, "internal: sliced segment grouping without its counters"};
  const uint32_t nb = (std::max(n_seq, 1u) + 63u) & ~63u;
  uint32_t nbits = 1;
  while ((1u << nbits) < n_seq) nbits++;
  const uint32_t grid = cdiv(n_queries * parts, SEG_WAVES);
  if (count_only)
    seg_group_kernel<true><<<grid, 64 * SEG_WAVES, SEG_WAVES * nb * 4, s>>>(fr, qfirst, qlast, run_start, run_end, h, n_queries, nb, nbits, qact, qdst, qgrp,
                                                                          gdst, gstart, gkey, svals, qbins, parts, qtot);
  else
    seg_group_kernel<false><<<grid, 64 * SEG_WAVES, SEG_WAVES * nb * 4, s>>>(fr, qfirst, qlast, run_start, run_end, h, n_queries, nb, nbits, qact, qdst, qgrp,
                                                                           gdst, gstart, gkey, svals, qbins, parts, qtot);
  if (count_only && parts > 1u) {
    IMPG_HIP(hipMemsetAsync(qact, 0, (size_t)n_queries * 4, s));
    IMPG_HIP(hipMemsetAsync(qgrp, 0, (size_t)n_queries * 4, s));
    seg_parts_scan_kernel<<<dim3(n_queries, cdiv(nb, 256u)), 256, 0, s>>>(qbins, nb, parts, qtot, qact, qgrp);
  }
}
uint32_t group_tiles(uint32_t n) { return (n + GROUP_TILE - 1u) / GROUP_TILE; }
void launch_group_count(const unsigned long long *skeys, uint32_t n, uint32_t *tile_heads, hipStream_t s) {
  if (n) group_count_kernel<<<group_tiles(n), 256, 0, s>>>(skeys, n, tile_heads);
}
void launch_group_fill(const unsigned long long *skeys, uint32_t n, const uint32_t *tile_base, uint32_t *gstart, unsigned long long *gkey, hipStream_t s) {
  if (n) group_fill_kernel<<<group_tiles(n), 256, 0, s>>>(skeys, n, tile_base, gstart, gkey);
}
void launch_group_heads(const unsigned long long *skeys, uint32_t n, uint32_t *head, hipStream_t s) {
  if (!n) return;
  group_heads_kernel<<<cdiv(n, 256), 256, 0, s>>>(skeys, n, head);
}
void launch_group_scatter(const unsigned long long *skeys, uint32_t n, const uint32_t *head, const uint32_t *gid,
                          uint32_t *gstart, unsigned long long *gkey, hipStream_t s) {
  if (!n) return;
  group_scatter_kernel<<<cdiv(n, 256), 256, 0, s>>>(skeys, n, head, gid, gstart, gkey);
}
void launch_table_qoff(const unsigned long long *keys, uint32_t n_groups, uint32_t n_queries, uint32_t *qoff, hipStream_t s) {
  table_qoff_kernel<<<cdiv((uint64_t)n_queries + 1, 256), 256, 0, s>>>(keys, n_groups, n_queries, qoff);
}
void launch_group_prepare(const VisitedTables &vt, const unsigned long long *gkey, const uint32_t *gstart,
                          uint32_t n_groups, uint32_t n_active, uint32_t *glen, const int2 **old_src,
                          uint32_t *cap, uint32_t *pcap, hipStream_t s) {
  if (!n_groups) return;
  group_prepare_kernel<<<cdiv(n_groups, 256), 256, 0, s>>>(vt, gkey, gstart, n_groups, n_active, glen, old_src, cap, pcap);
}
void launch_visited_update(const unsigned long long *svals, const int32_t *seq_len,
                           const unsigned long long *gkey, const uint32_t *gstart, const uint32_t *glen,
                           const int2 *const *old_src, const uint32_t *noff, const uint32_t *poff,
                           uint32_t n_groups, int32_t min_transitive_len, int32_t mdbr, int2 *new_ranges,
                           uint32_t *new_len, int2 *pieces, uint32_t *n_pieces, const uint32_t *cap, const uint32_t *pcap,
                           uint32_t *big_list, uint32_t *n_big, hipStream_t s) {
  if (!n_groups) return;
  (void)hipMemsetAsync(n_big, 0, 32, s);  // three list lengths, three work counters, the mid list's length
  big_groups_kernel<<<cdiv(n_groups, BG_THREADS * BG_PER_THREAD), BG_THREADS, 0, s>>>(cap, pcap, n_groups, big_list, n_big);
  const unsigned long long *srcs = reinterpret_cast<const unsigned long long *>(old_src);
  visited_update_kernel<VU_LDS_CAP, false><<<cdiv(n_groups, 64), 64, 0, s>>>(svals, seq_len, gkey, gstart, glen, srcs, cap, noff, poff, n_groups,
                                                                             min_transitive_len, mdbr, new_ranges, new_len, pieces, n_pieces,
                                                                             nullptr, nullptr);
  // the listed groups: as many blocks as stay resident (their number is on the device), each striding over the list
  const uint32_t mid_blocks = std::min<uint32_t>(cdiv(n_groups, 64), 256u * std::min(32u, (160u * 1024u) / (VU_MID_CAP * 64u * 8u)));
  visited_update_kernel<VU_MID_CAP, true><<<mid_blocks, 64, 0, s>>>(svals, seq_len, gkey, gstart, glen, srcs, cap, noff, poff, n_groups,
                                                                    min_transitive_len, mdbr, new_ranges, new_len, pieces, n_pieces,
                                                                    big_list + 2u * (size_t)n_groups, n_big + 6);
  // one wave per big group, grid-strided over however many there are (the count stays on the device)
  const uint32_t blocks = std::min<uint32_t>(n_groups, 256u * std::min(32u, (160u * 1024u) / (VW_CAP_SMALL * 8u)));
  visited_update_wave_kernel<VW_CAP_SMALL><<<blocks, 64, 0, s>>>(
      svals, seq_len, gkey, gstart, glen, old_src, cap, noff, poff, big_list, n_big, n_big + 3, 0u, min_transitive_len, mdbr, new_ranges,
      new_len, pieces, n_pieces);
  visited_update_wave_kernel<VW_CAP_TINY, true><<<std::min<uint32_t>(n_groups, 256u * 32u), 64, 0, s>>>(
      svals, seq_len, gkey, gstart, glen, old_src, cap, noff, poff, big_list + n_groups, n_big + 2, n_big + 5, 0u, min_transitive_len, mdbr, new_ranges,
      new_len, pieces, n_pieces);
  visited_update_wave_kernel<VW_CAP_LARGE><<<std::min<uint32_t>(n_groups, 256u * 5u), 64, 0, s>>>(
      svals, seq_len, gkey, gstart, glen, old_src, cap, noff, poff, big_list, n_big + 1, n_big + 4, n_groups, min_transitive_len, mdbr,
      new_ranges, new_len, pieces, n_pieces);
}
void launch_covered_flags(const unsigned long long *svals, const uint32_t *head, const uint32_t *gid,
                          const unsigned long long *gkey, const int2 *const *old_src, const uint32_t *cap, const uint32_t *glen,
                          const int32_t *seq_len, uint32_t n_active, uint32_t *keep, hipStream_t s) {
  if (n_active) covered_flags_kernel<<<cdiv(n_active, 256), 256, 0, s>>>(svals, head, gid, gkey, old_src, cap, glen, seq_len, n_active, keep);
}
void launch_covered_compact(const unsigned long long *svals, const uint32_t *keep, const uint32_t *kpos, uint32_t n_active,
                            unsigned long long *out, uint32_t n_kept, uint32_t n_groups, uint32_t *gstart, uint32_t *glen, uint32_t *cap,
                            uint32_t *pcap, hipStream_t s) {
  if (n_active) covered_compact_kernel<<<cdiv(n_active, 256), 256, 0, s>>>(svals, keep, kpos, n_active, out);
  if (n_groups) covered_regroup_kernel<<<cdiv(n_groups, 256), 256, 0, s>>>(kpos, n_active, n_kept, n_groups, gstart, glen, cap, pcap);
}
void launch_frontier_emit(const unsigned long long *gkey, const uint32_t *poff, const uint32_t *n_pieces,
                          const uint32_t *foff, uint32_t n_groups, const int2 *pieces, FrontierRec *out, hipStream_t s,
                          const DeviceIndexView *v, uint32_t *key, uint32_t *idx) {
  if (!n_groups) return;
  frontier_emit_kernel<<<cdiv(n_groups, 256), 256, 0, s>>>(gkey, poff, n_pieces, foff, n_groups, pieces, out, v ? v->seg : nullptr,
                                                           v ? v->seq_len : nullptr, v ? v->n_seq : 0u, v ? key : nullptr, v ? idx : nullptr);
}
void launch_subset_filter(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, uint32_t *qid,
                          const uint8_t *keep, const impg_gpu_range_t *ranges, hipStream_t s) {
  if (n_pairs) subset_filter_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, qid, keep, ranges);
}
void launch_mask_caps(const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *mask_off, uint32_t n_seq, uint32_t *cap,
                      hipStream_t s) {
  if (n) mask_caps_kernel<<<cdiv(n, 256), 256, 0, s>>>(ranges, n, mask_off, n_seq, cap);
}
void launch_visited_init_masked(const impg_gpu_range_t *ranges, uint32_t n, const int32_t *init_len, uint32_t n_seq,
                                const uint32_t *mask_off, const int2 *mask_ranges, int32_t min_transitive_len,
                                const uint32_t *loff, unsigned long long *keys, uint32_t *len, int2 *rng, int2 *pieces,
                                uint32_t *n_self, uint32_t *n_front, hipStream_t s) {
  if (n) visited_init_masked_kernel<<<cdiv(n, 64), 64, 0, s>>>(ranges, n, init_len, n_seq, mask_off, mask_ranges,
                                                                min_transitive_len, loff, keys, len, rng, pieces, n_self, n_front);
}
void launch_masked_self_emit(const impg_gpu_range_t *ranges, uint32_t n, const uint32_t *loff, const uint32_t *n_self,
                             const uint32_t *self_off, const uint32_t *front_off, int32_t min_transitive_len,
                             const int2 *pieces, FrontierRec *self_out, FrontierRec *frontier_out, hipStream_t s) {
  if (n) masked_self_emit_kernel<<<cdiv(n, 256), 256, 0, s>>>(ranges, n, loff, n_self, self_off, front_off, min_transitive_len,
                                                              pieces, self_out, frontier_out);
}
void launch_visited_init(const impg_gpu_range_t *ranges, uint32_t n, const int32_t *seq_len, uint32_t n_seq,
                         int32_t min_transitive_len, unsigned long long *keys, uint32_t *off, uint32_t *len, int2 *rng,
                         FrontierRec *self_iv, uint32_t *in_frontier, hipStream_t s) {
  if (!n) return;
  visited_init_kernel<<<cdiv(n, 256), 256, 0, s>>>(ranges, n, seq_len, n_seq, min_transitive_len, keys, off, len, rng, self_iv, in_frontier);
}
void launch_compact_frontier(const FrontierRec *in, const uint32_t *flag, const uint32_t *pos, uint32_t n,
                             FrontierRec *out, hipStream_t s) {
  if (!n) return;
  compact_frontier_kernel<<<cdiv(n, 256), 256, 0, s>>>(in, flag, pos, n, out);
}
void launch_ranges_to_frontier(const impg_gpu_range_t *ranges, uint32_t n, FrontierRec *out, hipStream_t s) {
  if (!n) return;
  ranges_to_frontier_kernel<<<cdiv(n, 256), 256, 0, s>>>(ranges, n, out);
}
void launch_frontier_to_stack(const FrontierRec *fr, uint32_t n, const uint32_t *pop_depth, bool use_depth,
                              unsigned long long *key, int32_t *st, int32_t *en, uint32_t *depth, hipStream_t s) {
  if (!n) return;
  frontier_to_stack_kernel<<<cdiv(n, 256), 256, 0, s>>>(fr, n, pop_depth, use_depth ? 1 : 0, key, st, en, depth);
}
void launch_dfs_pop_flags(const unsigned long long *key, const uint32_t *depth, uint32_t n, uint32_t max_depth,
                          bool pop_front, uint32_t *fr_flag, uint32_t *keep_flag, uint32_t *pop_depth, uint32_t *pop_sel, uint32_t n_queries,
                          hipStream_t s) {
  if (!n) return;
  IMPG_HIP(hipMemsetAsync(pop_sel, pop_front ? 0xFF : 0x00, (size_t)n_queries * 4, s));
  dfs_pop_select_kernel<<<cdiv(n, 256), 256, 0, s>>>(key, depth, n, max_depth, pop_front ? 1 : 0, pop_sel);
  dfs_pop_flags_kernel<<<cdiv(n, 256), 256, 0, s>>>(key, depth, n, pop_front ? 1 : 0, pop_sel, fr_flag, keep_flag, pop_depth);
}
void launch_sort5(const FrontierRec *fr, uint32_t n, const uint32_t *pair_off, uint32_t n_pairs, HitArrays h,
                  const uint32_t *pair_entry, const uint32_t *mrank, uint32_t *dest, hipStream_t s) {
  if (n && n_pairs) sort5_kernel<<<wave_grid(n), 256, 0, s>>>(fr, n, pair_off, n_pairs, h, pair_entry, mrank, dest);
}
void launch_permute_slots(const uint32_t *dest, uint32_t n_pairs, HitArrays in, HitArrays out, const uint32_t *pe_in,
                          uint32_t *pe_out, SliceArrays sin, SliceArrays sout, hipStream_t s) {
  if (n_pairs) permute_slots_kernel<<<cdiv(n_pairs, 256), 256, 0, s>>>(dest, n_pairs, in, out, pe_in, pe_out, sin, sout);
}
void launch_dfs_pop_scatter(const unsigned long long *key, const int32_t *st, const int32_t *en, const uint32_t *depth,
                            uint32_t n, const uint32_t *fr_flag, const uint32_t *fr_pos, const uint32_t *keep_flag,
                            const uint32_t *keep_pos, FrontierRec *fr_out, unsigned long long *key_out, int32_t *st_out,
                            int32_t *en_out, uint32_t *depth_out, hipStream_t s) {
  if (!n) return;
  dfs_pop_scatter_kernel<<<cdiv(n, 256), 256, 0, s>>>(key, st, en, depth, n, fr_flag, fr_pos, keep_flag, keep_pos, fr_out,
                                                     key_out, st_out, en_out, depth_out);
}
void launch_iota(uint32_t *v, uint32_t n, hipStream_t s) { if (n) iota_kernel<<<cdiv(n, 256), 256, 0, s>>>(v, n); }
void launch_gather_u32(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *dst, hipStream_t s) {
  if (n) gather_u32_kernel<<<cdiv(n, 256), 256, 0, s>>>(src, idx, n, dst);
}
void launch_gather_u64(const unsigned long long *src, const uint32_t *idx, uint32_t n, unsigned long long *dst, hipStream_t s) {
  if (n) gather_u64_kernel<<<cdiv(n, 256), 256, 0, s>>>(src, idx, n, dst);
}
void launch_run_heads(const unsigned long long *skeys, uint32_t n, uint32_t *head, hipStream_t s) {
  if (n) run_heads_kernel<<<cdiv(n, 256), 256, 0, s>>>(skeys, n, head);
}
void launch_run_starts(const uint32_t *head, const uint32_t *gid, uint32_t n, uint32_t *gstart, hipStream_t s) {
  if (n) run_starts_kernel<<<cdiv(n, 256), 256, 0, s>>>(head, gid, n, gstart);
}
void launch_dfs_merge(const uint32_t *gstart, uint32_t n_groups, uint32_t n, const uint32_t *perm, const int32_t *st,
                      const int32_t *en, const uint32_t *depth, int32_t *st_m, int32_t *en_m, uint32_t *depth_m,
                      uint32_t *cnt, hipStream_t s) {
  if (n_groups) dfs_merge_kernel<<<cdiv(n_groups, 64), 64, 0, s>>>(gstart, n_groups, n, perm, st, en, depth, st_m, en_m, depth_m, cnt);
}
void launch_dfs_compact(const uint32_t *gstart, const uint32_t *cnt, const uint32_t *off, uint32_t n_groups,
                        const unsigned long long *skeys, const int32_t *st_m, const int32_t *en_m, const uint32_t *depth_m,
                        unsigned long long *key_out, int32_t *st_out, int32_t *en_out, uint32_t *depth_out, hipStream_t s) {
  if (n_groups) dfs_compact_kernel<<<cdiv(n_groups, 256), 256, 0, s>>>(gstart, cnt, off, n_groups, skeys, st_m, en_m, depth_m,
                                                                       key_out, st_out, en_out, depth_out);
}
size_t sort_u32_scratch_bytes(uint32_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *>(
      nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 32, (hipStream_t)0);
  return bytes;
}
void launch_sort_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                     uint32_t n, hipStream_t s, unsigned begin_bit, unsigned end_bit) {
  if (!n) return;
  IMPG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s));
}

// ---------------------------------------------------------------------------
// The lookup order's sort, hand-written (round 6): a stable argsort of n 32-bit keys by their low `end_bit` bits.
//
// What is sorted is small -- 4-byte keys whose values are their own indices, 21 key bits at the headline -- and the library's
// onesweep moved three times the algorithmic bytes for it (rocprofv3: 0.5 GB read + 0.5 GB written per pass over 21 x 10^6
// pairs, 555 us a pass: a block's ~4 000 keys over 256 bins leave in runs of 60 bytes).  Here a pass is an LSD counting
// sort by one digit of <= 7 bits (21 bits: three digits), a block per tile of 4 096 keys:
//   order_hist_kernel     the tile's digit histogram -> hist[digit][tile]
//   order_rowscan_kernel  a block per digit: its row of tile counts to exclusive offsets, the digit's total
//   order_scatter_kernel  the tile's keys ranked (a wave owns a quarter of the tile: its keys' ranks among equal digits from
//                         ballots, round by round, on per-wave LDS counters), laid out in LDS in digit order, and written
//                         from there: a digit's keys of a tile leave as ONE run (7-bit digits: 64 keys, 256 bytes).
// The first pass reads no values (a key's value is its index), the last writes no keys.  Stable: tile order, then the
// waves' quarters in order, then rounds, then lanes.
// ---------------------------------------------------------------------------
#ifndef IMPG_OS_TILE
#define IMPG_OS_TILE 4096
#endif
#ifndef IMPG_OS_THREADS
#define IMPG_OS_THREADS 512
#endif
constexpr uint32_t OS_TILE = IMPG_OS_TILE, OS_THREADS = IMPG_OS_THREADS, OS_WAVES = OS_THREADS / 64u, OS_ROUNDS = OS_TILE / OS_THREADS, OS_DIGIT_BITS = 7, OS_MAX_BINS = 1u << OS_DIGIT_BITS;
constexpr uint32_t OS_HIST_THREADS = 256;
static_assert(OS_ROUNDS * OS_THREADS == OS_TILE && OS_THREADS >= OS_MAX_BINS && OS_TILE / OS_WAVES <= 65535u && OS_TILE % (4u * OS_HIST_THREADS) == 0,
              "a thread per bin, whole rounds, a wave's counters fit 16 bits");
__global__ __launch_bounds__(OS_HIST_THREADS) void order_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t bins,
                                                                     uint32_t n_tiles, uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[OS_MAX_BINS];
  // (an XCD takes a contiguous eighth of the tiles, as in order_scatter_kernel: neighbouring tiles' counters are neighbouring words)
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t tile_id = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;
  if (threadIdx.x < OS_MAX_BINS) h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = tile_id * OS_TILE, m = bins - 1u;
  constexpr uint32_t HR = OS_TILE / (4u * OS_HIST_THREADS);
  uint4 k[HR];  // (all of the thread's keys requested before the first one is counted)
#pragma unroll
  for (uint32_t r = 0; r < HR; r++) {  // (tiles start at multiples of the tile: 16-byte aligned)
    const uint32_t j = (r * OS_HIST_THREADS + threadIdx.x) * 4u;
    k[r] = make_uint4(0, 0, 0, 0);
    if (base + j + 4u <= n) k[r] = *reinterpret_cast<const uint4 *>(keys + base + j);
    else {
      if (base + j < n) k[r].x = keys[base + j];
      if (base + j + 1u < n) k[r].y = keys[base + j + 1u];
      if (base + j + 2u < n) k[r].z = keys[base + j + 2u];
    }
  }
#pragma unroll
  for (uint32_t r = 0; r < HR; r++) {
    const uint32_t j = (r * OS_HIST_THREADS + threadIdx.x) * 4u;
    if (base + j < n) atomicAdd(&h[(k[r].x >> shift) & m], 1u);
    if (base + j + 1u < n) atomicAdd(&h[(k[r].y >> shift) & m], 1u);
    if (base + j + 2u < n) atomicAdd(&h[(k[r].z >> shift) & m], 1u);
    if (base + j + 3u < n) atomicAdd(&h[(k[r].w >> shift) & m], 1u);
  }
  __syncthreads();
  if (threadIdx.x < bins) hist[(size_t)threadIdx.x * n_tiles + tile_id] = h[threadIdx.x];
}
constexpr uint32_t OS_ROW_THREADS = 1024;
__global__ __launch_bounds__(OS_ROW_THREADS) void order_rowscan_kernel(uint32_t *__restrict__ hist, uint32_t n_tiles, uint32_t *__restrict__ tot) {
  __shared__ uint32_t wsum[OS_ROW_THREADS / 64u];
  uint32_t *row = hist + (size_t)blockIdx.x * n_tiles;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n_tiles; base += OS_ROW_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < n_tiles ? row[i] : 0u;
    const uint32_t ex = block_excl_scan_n<OS_ROW_THREADS / 64u>(x, wsum);
    if (i < n_tiles) row[i] = carry + ex;
    // (the turn's total: the last thread's inclusive value, through LDS)
    if (threadIdx.x == OS_ROW_THREADS - 1u) wsum[0] = ex + x;
    __syncthreads();
    carry += wsum[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}
#ifdef IMPG_OS_CLOCKS
__device__ unsigned long long g_os_clk[8];
#define OS_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); os_t[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define OS_MARK(i) do { } while (0)
#endif
template <bool FIRST, bool LAST>
#ifndef IMPG_OS_WAVES_PER_EU
#define IMPG_OS_WAVES_PER_EU 6
#endif
__global__ __launch_bounds__(OS_THREADS) __attribute__((amdgpu_waves_per_eu(IMPG_OS_WAVES_PER_EU, IMPG_OS_WAVES_PER_EU))) void order_scatter_kernel(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t n,
                                                                   uint32_t shift, uint32_t bins, uint32_t nbits, uint32_t n_tiles,
                                                                   const uint32_t *__restrict__ hist, const uint32_t *__restrict__ tot,
                                                                   uint32_t *__restrict__ kout, uint32_t *__restrict__ vout) {
  __shared__ uint2 tile[OS_TILE];                  // (key, value) at its place of the tile's digit order
  __shared__ uint16_t wcnt[OS_WAVES][OS_MAX_BINS];  // a wave's digit counters while it ranks; then its offset inside the digit's run
  __shared__ uint32_t dbase[OS_MAX_BINS];           // the digit's first place in the tile
  __shared__ uint32_t gbase[OS_MAX_BINS];           // ... and where that place goes: destination = gbase[digit] + place
  __shared__ uint32_t wsum[OS_WAVES];
  __shared__ unsigned long long wmatch[OS_WAVES][OS_MAX_BINS];  // a round's lanes per digit (see the ranking below)
  const uint32_t w = threadIdx.x >> 6, lane = lane_id(), m = bins - 1u;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: every XCD takes one contiguous eighth of the
  // tiles.  Neighbouring tiles' runs of a digit share their first / last line; dealt to different XCDs each L2 filled and
  // wrote back its own copy of that line (rocprofv3: 1.5 GB of HBM traffic for a pass that moves 0.34).
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t tile_id = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (tile_id >= n_tiles) return;  // (block-uniform: the grid is rounded up to the 8 XCDs)
  const uint32_t base = tile_id * OS_TILE;
  const uint32_t cnt = min(OS_TILE, n - base);
#ifdef IMPG_OS_CLOCKS
  unsigned long long os_t[8];
#endif
  OS_MARK(0);
  for (uint32_t i = threadIdx.x; i < OS_WAVES * OS_MAX_BINS; i += OS_THREADS) { (&wcnt[0][0])[i] = 0; (&wmatch[0][0])[i] = 0ull; }
  // (what the bases step adds up, requested now: the tile's offsets inside the digits' runs, the digits' totals)
  uint32_t my_hist = 0, my_tot = 0;
  if (threadIdx.x < bins) { my_hist = hist[(size_t)threadIdx.x * n_tiles + tile_id]; my_tot = tot[threadIdx.x]; }
  // the wave's stretch of the tile, 64 keys a round: a short chain of rounds per wave, many waves
  uint32_t key[OS_ROUNDS], val[FIRST ? 1 : OS_ROUNDS], loc[OS_ROUNDS];
  const uint32_t j0 = w * (OS_TILE / OS_WAVES) + lane;
#pragma unroll
  for (uint32_t r = 0; r < OS_ROUNDS; r++) {
    const uint32_t j = j0 + r * 64u;
    key[r] = j < cnt ? kin[base + j] : 0xFFFFFFFFu;
    if (!FIRST) val[r] = j < cnt ? vin[base + j] : 0u;
  }
  __syncthreads();
  OS_MARK(1);
#pragma unroll
  for (uint32_t r = 0; r < OS_ROUNDS; r++) {
    const uint32_t j = j0 + r * 64u;
    const bool live = j < cnt;
    const uint32_t d = (key[r] >> shift) & m;
    // the round's lanes of this lane's digit: every lane ORs its bit into the digit's word of the wave (an OR commutes: the
    // word does not depend on the order the LDS serves the lanes in), reads the word back, and the digit's last lane
    // clears it and moves the digit's counter on.  (Ballots over the digit's bits -- eight per round -- kept the vector
    // ALUs busy for 1 600 cycles a round; this is three LDS operations.)
    unsigned long long mask = 1ull << lane;
    uint32_t prior = 0;
    if (live) atomicOr(&wmatch[w][d], 1ull << lane);
    __builtin_amdgcn_wave_barrier();
    if (live) { mask = wmatch[w][d]; prior = wcnt[w][d]; }
    __builtin_amdgcn_wave_barrier();
    if (live && lane == 63u - (uint32_t)__clzll((long long)mask)) { wmatch[w][d] = 0ull; wcnt[w][d] = (uint16_t)(prior + (uint32_t)__popcll(mask)); }
    __builtin_amdgcn_wave_barrier();
    loc[r] = prior + (uint32_t)__popcll(mask & lanemask_lt());
  }
  __syncthreads();
  OS_MARK(2);
  // per digit: the waves' offsets inside its run, its first place in the tile, its run's destination
  {
    const uint32_t t = threadIdx.x;
    uint32_t total = 0;
    if (t < OS_MAX_BINS) {
#pragma unroll
      for (uint32_t k = 0; k < OS_WAVES; k++) { const uint32_t c = wcnt[k][t]; wcnt[k][t] = (uint16_t)total; total += c; }
    }
    const uint32_t ex = block_excl_scan_n<OS_WAVES>(total, wsum);
    const uint32_t gx = block_excl_scan_n<OS_WAVES>(my_tot, wsum);  // the digit's first destination over all tiles
    if (t < OS_MAX_BINS) {
      dbase[t] = ex;
      gbase[t] = t < bins ? gx + my_hist - ex : 0u;
    }
  }
  __syncthreads();
  OS_MARK(3);
#pragma unroll
  for (uint32_t r = 0; r < OS_ROUNDS; r++) {
    const uint32_t j = j0 + r * 64u;
    if (j < cnt) {
      const uint32_t d = (key[r] >> shift) & m;
      tile[dbase[d] + wcnt[w][d] + loc[r]] = make_uint2(key[r], FIRST ? base + j : val[r]);
    }
  }
  __syncthreads();
  OS_MARK(4);
  for (uint32_t p = threadIdx.x; p < cnt; p += OS_THREADS) {
    const uint2 kv = tile[p];
    const uint32_t dest = gbase[(kv.x >> shift) & m] + p;
    if (!LAST) kout[dest] = kv.x;
    vout[dest] = kv.y;
  }
#ifdef IMPG_OS_CLOCKS
  OS_MARK(5);
  if (threadIdx.x == 0) { for (int i = 0; i < 5; i++) atomicAdd(&g_os_clk[i], os_t[i + 1] - os_t[i]); atomicAdd(&g_os_clk[7], 1ull); }
#endif
}
#ifdef IMPG_OS_CLOCKS
void order_sort_clocks(unsigned long long *out, bool reset) {
  if (reset) { unsigned long long z[8] = {0}; IMPG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_os_clk), z, sizeof z)); return; }
  IMPG_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_clk), 64));
}
#endif
static unsigned order_sort_passes(unsigned end_bit) { return std::max(1u, (end_bit + OS_DIGIT_BITS - 1u) / OS_DIGIT_BITS); }
size_t order_sort_scratch_bytes(uint32_t n) {
  const size_t n_tiles = ((size_t)n + OS_TILE - 1) / OS_TILE;
  return (OS_MAX_BINS * std::max<size_t>(n_tiles, 1) + OS_MAX_BINS) * 4;
}
// keys: the n keys (left in an unspecified order); key_tmp, perm_tmp: n words each; scratch: order_sort_scratch_bytes(n).
// perm_out[k] = the index of the k-th smallest key by its bits [0, end_bit), equal keys in index order.
void launch_order_sort(uint32_t *keys, uint32_t *key_tmp, uint32_t *perm_out, uint32_t *perm_tmp, uint32_t n, unsigned end_bit, void *scratch,
                       hipStream_t s) {
  if (!n) return;
  end_bit = std::min(32u, std::max(1u, end_bit));
  const unsigned np = order_sort_passes(end_bit), per = (end_bit + np - 1u) / np;
  const uint32_t n_tiles = cdiv(n, OS_TILE);
  uint32_t *hist = reinterpret_cast<uint32_t *>(scratch), *tot = hist + (size_t)OS_MAX_BINS * n_tiles;
  // pass k (from 1) writes perm_out when np - k is even: the last pass lands there
  uint32_t *kin = keys, *kout = key_tmp;
  const uint32_t *vin = nullptr;
  for (unsigned k = 1, shift = 0; k <= np; k++, shift += per) {
    const unsigned nbits = std::min(per, end_bit - shift);
    const uint32_t bins = 1u << nbits;
    uint32_t *vout = ((np - k) & 1u) ? perm_tmp : perm_out;
    const uint32_t grid8 = (n_tiles + 7u) & ~7u;
    order_hist_kernel<<<grid8, OS_HIST_THREADS, 0, s>>>(kin, n, shift, bins, n_tiles, hist);
    order_rowscan_kernel<<<bins, OS_ROW_THREADS, 0, s>>>(hist, n_tiles, tot);
    const bool first = k == 1, last = k == np;
    if (first && last) order_scatter_kernel<true, true><<<grid8, OS_THREADS, 0, s>>>(kin, vin, n, shift, bins, nbits, n_tiles, hist, tot, kout, vout);
    else if (first) order_scatter_kernel<true, false><<<grid8, OS_THREADS, 0, s>>>(kin, vin, n, shift, bins, nbits, n_tiles, hist, tot, kout, vout);
    else if (last) order_scatter_kernel<false, true><<<grid8, OS_THREADS, 0, s>>>(kin, vin, n, shift, bins, nbits, n_tiles, hist, tot, kout, vout);
    else order_scatter_kernel<false, false><<<grid8, OS_THREADS, 0, s>>>(kin, vin, n, shift, bins, nbits, n_tiles, hist, tot, kout, vout);
    std::swap(kin, kout);
    vin = vout;
  }
}
size_t sort_u64v_scratch_bytes(uint32_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, const unsigned long long *, unsigned long long *,
                                  const unsigned long long *, unsigned long long *>(nullptr, bytes, nullptr, nullptr, nullptr,
                                                                                    nullptr, n, 0, 64, (hipStream_t)0);
  return bytes;
}
void launch_sort_u64v(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout,
                      const unsigned long long *vin, unsigned long long *vout, uint32_t n, hipStream_t s, unsigned end_bit,
                      unsigned begin_bit) {
  if (!n) return;
  IMPG_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s));
}
void launch_compact_fill(const unsigned long long *keys, uint32_t n, uint32_t table, unsigned long long *key_out,
                         unsigned long long *src_out, hipStream_t s) {
  if (n) compact_fill_kernel<<<cdiv(n, 256), 256, 0, s>>>(keys, n, table, key_out, src_out);
}
void launch_compact_last(const unsigned long long *skeys, uint32_t n, uint32_t *flag, hipStream_t s) {
  if (n) compact_last_kernel<<<cdiv(n, 256), 256, 0, s>>>(skeys, n, flag);
}
void launch_compact_select(const VisitedTables &vt, const unsigned long long *skeys, const unsigned long long *ssrc, uint32_t n,
                           const uint32_t *flag, const uint32_t *pos, unsigned long long *key_out,
                           unsigned long long *src_out, uint32_t *len_out, hipStream_t s) {
  if (n) compact_select_kernel<<<cdiv(n, 256), 256, 0, s>>>(vt, skeys, ssrc, n, flag, pos, key_out, src_out, len_out);
}
void launch_compact_copy(const VisitedTables &vt, const unsigned long long *src, const uint32_t *off, const uint32_t *len,
                         uint32_t n, int2 *ranges_out, hipStream_t s) {
  if (n) compact_copy_kernel<<<cdiv(n, 256), 256, 0, s>>>(vt, src, off, len, n, ranges_out);
}
void launch_hits_pack(const FrontierRec *fr, const uint32_t *pair_range, uint32_t n_pairs, HitArrays h, const uint32_t *pair_entry,
                      const uint32_t *mrank, uint32_t words, void *out, hipStream_t s, const uint32_t *slice_n) {
  if (!n_pairs) return;
  if (words == 4) hits_pack_kernel<4><<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, h, pair_entry, mrank, nullptr, (uint4 *)out);
  else hits_pack_kernel<8><<<cdiv(n_pairs, 256), 256, 0, s>>>(fr, pair_range, n_pairs, h, pair_entry, mrank, slice_n, (uint4 *)out);
}
void launch_hits_slice_n(const void *in, uint32_t n, uint32_t *cnt, hipStream_t s) {
  if (n) hits_slice_n_kernel<<<cdiv(n, 256), 256, 0, s>>>((const uint4 *)in, n, cnt);
}
void launch_hits_unpack(const void *in, uint32_t n, uint32_t words, uint32_t n_front, const uint32_t *run_start, const uint32_t *off,
                        uint32_t *pair_range, HitArrays h, uint32_t *mslot, hipStream_t s, const uint32_t *slice_at, uint32_t *slice_pos,
                        uint32_t *slice_n) {
  if (!n) return;
  if (words == 4) hits_unpack_kernel<4><<<cdiv(n, 256), 256, 0, s>>>((const uint4 *)in, n, n_front, run_start, off, pair_range, h, mslot, nullptr, nullptr, nullptr);
  else hits_unpack_kernel<8><<<cdiv(n, 256), 256, 0, s>>>((const uint4 *)in, n, n_front, run_start, off, pair_range, h, mslot, slice_at, slice_pos, slice_n);
}
#include "walk_device.inc"
size_t walk_slab_bytes(uint32_t n_seq, bool wide, uint32_t wcap, uint32_t hcap, uint32_t vcap, uint32_t gcap, uint32_t scap) {
  return walk_slab_layout(nullptr, n_seq, wide ? 1024u : 64u, wcap, hcap, vcap, gcap, scap, nullptr);
}
// Workgroups of the walk's grid form the runtime says one CU keeps resident (its members wait for each other inside one
// launch: a launch must fit the device; 0 = the form cannot run here).  The smallest over the three instantiations.
uint32_t walk_grid_blocks_per_cu() {
  int a = 0, b = 0, c = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, walk_grid_kernel<16, 4096, 0>, 1024, 0) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, walk_grid_kernel<16, 4096, MODE_IDENT>, 1024, 0) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, walk_grid_kernel<16, 4096, MODE_IDENT | MODE_WALK>, 1024, 0) != hipSuccess) return 0;
  return (uint32_t)std::max(0, std::min({a, b, c}));
}
void launch_walk(const WalkArgs &a, uint32_t n_workgroups, bool wide, bool ident_mode, hipStream_t s) {
  if (!n_workgroups) return;
  if (wide && a.members > 1) {
    if (ident_mode && !a.v.pfx) walk_grid_kernel<16, 4096, MODE_IDENT | MODE_WALK><<<n_workgroups, 1024, 0, s>>>(a);
    else if (ident_mode) walk_grid_kernel<16, 4096, MODE_IDENT><<<n_workgroups, 1024, 0, s>>>(a);
    else walk_grid_kernel<16, 4096, 0><<<n_workgroups, 1024, 0, s>>>(a);
  } else if (wide) {
    if (ident_mode && !a.v.pfx) walk_kernel<16, 4096, MODE_IDENT | MODE_WALK><<<n_workgroups, 1024, 0, s>>>(a);
    else if (ident_mode) walk_kernel<16, 4096, MODE_IDENT><<<n_workgroups, 1024, 0, s>>>(a);
    else walk_kernel<16, 4096, 0><<<n_workgroups, 1024, 0, s>>>(a);
  } else {
    if (ident_mode && !a.v.pfx) walk_wave_kernel<256, MODE_IDENT | MODE_WALK><<<n_workgroups, 64, 0, s>>>(a);
    else if (ident_mode) walk_wave_kernel<256, MODE_IDENT><<<n_workgroups, 64, 0, s>>>(a);
    else walk_wave_kernel<256, 0><<<n_workgroups, 64, 0, s>>>(a);
  }
}
}  // namespace impg
