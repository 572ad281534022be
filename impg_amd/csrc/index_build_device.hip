// Index construction on the device ("GPU-side index build"): what index_build.cpp's host
// builder does with threads and 2.7 GB of host arrays per million records -- op lines, prefix lines, identity
// prefixes, per-tile checkpoints, entries grouped by target and sorted by start, the search columns and their
// sampled levels -- done by kernels on the packed ops as the host hands them over.  Only the ops (4 bytes each) and
// the records cross PCIe; the host keeps what depends on counts alone (segment table, visit ranks).
//
//   tiles     one wave per record, two tiles per step: lanes 32 h + 6 + u hold op u of tile h, five wave scans
//             (target / query deltas, matched and mismatched bases, gap ops) give every op its running sums, and
//             lane w of a half stores word w of the tile's op line and of its prefix line (two coalesced 128-byte
//             stores per half); impg_internal.hpp describes both lines, index_build.cpp builds the same words
//   entries   (record, forward | reversed) pairs in record order, keyed by target << 32 | start: one stable radix
//             sort gives the per-target segments in start order with ties in record order -- the order
//             Impg::from_multi_alignment_records + BasicCOITree::new's stable sort produce (impg.rs:1559-1630)
//   columns   starts / ends / running max (a segmented max-scan) / levels, checkpoints from the tiles' own headers
//
// The host builder stays for what this one does not take (tracepoint indexes, a loaded .impg file's entry order) and
// as the checker: tests/test_gpu_parity.py::test_device_build_matches_host_build compares the saved bytes.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>

#include "kernels.hpp"

namespace impg {

namespace {

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
inline unsigned bits_for(uint64_t n) {
  unsigned b = 0;
  while ((1ull << b) < n) b++;
  return std::max(1u, b);
}
__device__ __forceinline__ unsigned lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wscan(uint32_t x) {  // inclusive scan over the wave's 64 lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, d);
    if ((int)lane() >= d) x += y;
  }
  return x;
}
__device__ __forceinline__ uint32_t from_lane(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src); }

// ---- tiles ----------------------------------------------------------------------------------------------------------
// ops: the batch's slice of the op pool (op k of the pool at ops[k - ops_first]); records [rec0, rec0 + n_rec)
__global__ __launch_bounds__(256) void tiles_kernel(const impg_gpu_record_t *__restrict__ records, const uint8_t *__restrict__ need,
                                                    const uint32_t *__restrict__ tile_base, uint32_t rec0, uint32_t n_rec,
                                                    const uint32_t *__restrict__ ops, unsigned long long ops_first,
                                                    uint32_t *__restrict__ pool, uint32_t *__restrict__ pfx, uint4 *__restrict__ idp,
                                                    uint32_t *__restrict__ rec_totT, uint32_t *__restrict__ rec_totQ,
                                                    uint32_t *__restrict__ bad_op) {
  const uint32_t ri = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (ri >= n_rec) return;
  const uint32_t rec = rec0 + ri;
  if (!need[rec]) return;
  const impg_gpu_record_t r = records[rec];
  const uint32_t n = r.cigar_len;
  const uint32_t *src = ops + (r.cigar_off - ops_first);
  const uint32_t l = lane(), h = l >> 5, w = l & 31u, hb = h << 5;
  uint32_t cT = 0, cQ = 0, cM = 0, cX = 0, cG = 0;  // running sums of the record before this step's first op
  const uint32_t m = (n + TILE_OPS - 1) / TILE_OPS;
  for (uint32_t j0 = 0; j0 < m; j0 += 2) {
    const uint32_t j = j0 + h;             // this half's tile
    const uint32_t k0 = j * TILE_OPS;
    const bool tile_on = j < m;
    const uint32_t cnt = tile_on ? min(TILE_OPS, n - k0) : 0u;
    const uint32_t u = w - 6u;             // op of the tile this lane holds (w >= 6)
    const bool on = w >= 6u && u < cnt;
    uint32_t op = OP_PAD, dT = 0, dQ = 0, dM = 0, dX = 0, dG = 0;
    if (on) {
      op = src[k0 + u];
      const uint32_t code = op >> 29, len = op & OP_LEN_MASK;
      if (code > 4u) *bad_op = 1u;  // CigarOp::new panics (impg.rs:88)
      if (code != 2u) dT = len;     // target_delta: all but 'I' (impg.rs:115-121)
      if (code != 3u) dQ = len;     // |query_delta|: all but 'D' (impg.rs:123-135)
      if (code == 0u || code == 4u) dM = len;  // 'M' counted as match (impg.rs:2959)
      else if (code == 1u) dX = len;
      else dG = 1u;                 // gap-compressed: one per 'I' / 'D' op
    }
    const uint32_t iT = cT + wscan(dT), iQ = cQ + wscan(dQ), iM = cM + wscan(dM), iX = cX + wscan(dX), iG = cG + wscan(dG);
    const uint32_t eT = iT - dT, eQ = iQ - dQ, eM = iM - dM, eX = iX - dX, eG = iG - dG;  // before this lane's op
    // the tile's start (before its op 0: lane hb + 6) and end (after its last op: lane hb + 5 + cnt)
    const uint32_t t0 = from_lane(eT, hb + 6u), q0 = from_lane(eQ, hb + 6u);
    const uint32_t last = hb + 5u + max(cnt, 1u);
    const uint32_t endT = from_lane(iT, last), endQ = from_lane(iQ, last), endM = from_lane(iM, last), endX = from_lane(iX, last),
                   endG = from_lane(iG, last);
    // prefix-line entry of this lane's op, and the tile's end entry
    const uint32_t ent = ((eT - t0) & 0xFFFFu) | ((eQ - q0) << 16);
    const uint32_t ent_end = ((endT - t0) & 0xFFFFu) | ((endQ - q0) << 16);
    bool wide = on && (eT - t0 > 0xFFFFu || eQ - q0 > 0xFFFFu);
    const unsigned long long wmask = __ballot(wide);
    const bool pwide = ((wmask >> hb) & 0xFFFFFFFFull) != 0ull || endT - t0 > 0xFFFFu || endQ - q0 > 0xFFFFu;
    // word w of the prefix line: entries start at word PFX_E0, entry k sits with the lane of op k = this lane + 2
    const uint32_t ent_up2 = (uint32_t)__shfl_down((int)ent, 2);
    const uint32_t e9 = 9u < cnt ? from_lane(ent, hb + 15u) : ent_end, e18 = 18u < cnt ? from_lane(ent, hb + 24u) : ent_end;
    uint32_t pw;
    if (w >= PFX_E0) pw = (w - PFX_E0) < cnt ? ent_up2 : ent_end;
    else pw = w == 0u ? (t0 | (pwide ? 1u << 31 : 0u)) : w == 1u ? q0 : w == 2u ? e9 : e18;
    // sums before the sub-tiles (ops 6, 14, 22) and at the end: the op line's header and the identity prefixes
    const uint32_t sT1 = 6u < cnt ? from_lane(eT, hb + 12u) : endT, sT2 = 14u < cnt ? from_lane(eT, hb + 20u) : endT,
                   sT3 = 22u < cnt ? from_lane(eT, hb + 28u) : endT;
    const uint32_t sQ1 = 6u < cnt ? from_lane(eQ, hb + 12u) : endQ, sQ2 = 14u < cnt ? from_lane(eQ, hb + 20u) : endQ,
                   sQ3 = 22u < cnt ? from_lane(eQ, hb + 28u) : endQ;
    const uint32_t dt1 = sT1 - t0, dt2 = sT2 - t0, dt3 = sT3 - t0, dt4 = endT - t0;
    const uint32_t dq1 = sQ1 - q0, dq2 = sQ2 - q0, dq3 = sQ3 - q0, dq4 = endQ - q0;
    uint32_t lw = op;  // words 6..31: the ops, padded
    if (w < 6u) {
      const bool lwide = dt4 >= TILE_WIDE || dq4 >= TILE_WIDE;
      lw = w == 0u ? t0 : w == 1u ? q0 : lwide ? 0xFFFFFFFFu : w == 2u ? (dt1 | dt2 << 16) : w == 3u ? (dt3 | dt4 << 16) : w == 4u ? (dq1 | dq2 << 16)
                                                                                                                                : (dq3 | dq4 << 16);
    }
    // identity prefixes before sub-tile s = w (lanes 0..3 of the half): at the sub-tile's first op, or the tile's end
    const uint32_t sfo = sub_first_op(w & 3u);
    const uint32_t isrc = hb + 6u + sfo;
    const uint32_t pM = from_lane(eM, isrc), pX = from_lane(eX, isrc), pG = from_lane(eG, isrc);
    // word w of the identity line (with prefix lines): header, then entry k with the lane of op k = this lane + 2
    const uint32_t m0 = from_lane(eM, hb + 6u), x0 = from_lane(eX, hb + 6u), g0 = from_lane(eG, hb + 6u);
    const uint32_t ient = ((eM - m0) & 0xFFFFu) | ((eX - x0) << 16), ient_end = ((endM - m0) & 0xFFFFu) | ((endX - x0) << 16);
    const uint32_t ient_up2 = (uint32_t)__shfl_down((int)ient, 2);
    const uint32_t gapmask = (uint32_t)((__ballot(on && dG != 0u) >> (hb + 6u)) & 0x3FFFFFFull);
    uint32_t iw;
    if (w >= IDL_E0) iw = (w - IDL_E0) < cnt ? ient_up2 : ient_end;
    else iw = w == 0u ? m0 : w == 1u ? x0 : w == 2u ? g0 : gapmask;
    if (tile_on) {
      const size_t tile = (size_t)tile_base[rec] + j;
      pool[tile * TILE_WORDS + w] = lw;
      if (pfx) pfx[tile * TILE_WORDS + w] = pw;
      if (pfx) { if (idp) reinterpret_cast<uint32_t *>(idp)[tile * IDL_WORDS + w] = iw; }  // (idp null: the identity lines come later, on demand)
      else if (w < TILE_SUBS) idp[TILE_SUBS * tile + w] = sfo < cnt ? make_uint4(pM, pX, pG, 0u) : make_uint4(endM, endX, endG, 0u);
    }
    // carry: the sums after this step's last op (lane 63 holds them whatever the halves' fill)
    cT = from_lane(iT, 63u); cQ = from_lane(iQ, 63u); cM = from_lane(iM, 63u); cX = from_lane(iX, 63u); cG = from_lane(iG, 63u);
  }
  if (l == 0) { rec_totT[rec] = cT; rec_totQ[rec] = cQ; }
}

// ---- entries --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool owned(const uint32_t *owner, uint32_t shard, uint32_t key) { return !owner || owner[key] == shard; }
__global__ __launch_bounds__(256) void entry_counts_kernel(const impg_gpu_record_t *__restrict__ records, uint32_t n_rec, int bidirectional,
                                                           const uint32_t *__restrict__ owner, uint32_t shard, uint32_t *__restrict__ cnt) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_rec) return;
  const impg_gpu_record_t r = records[i];
  uint32_t c = owned(owner, shard, r.target_id) ? 1u : 0u;
  if (bidirectional && r.query_id != r.target_id && owned(owner, shard, r.query_id)) c += 1u;  // impg.rs:1584
  cnt[i] = c;
}
__global__ __launch_bounds__(256) void entry_keys_kernel(const impg_gpu_record_t *__restrict__ records, uint32_t n_rec, int bidirectional,
                                                         const uint32_t *__restrict__ owner, uint32_t shard, const uint32_t *__restrict__ pos,
                                                         unsigned long long *__restrict__ key, uint32_t *__restrict__ val) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_rec) return;
  const impg_gpu_record_t r = records[i];
  uint32_t p = pos[i];
  if (owned(owner, shard, r.target_id)) {
    key[p] = ((unsigned long long)r.target_id << 32) | ((uint32_t)r.target_start ^ 0x80000000u);
    val[p] = i << 1;
    p++;
  }
  if (bidirectional && r.query_id != r.target_id && owned(owner, shard, r.query_id)) {
    key[p] = ((unsigned long long)r.query_id << 32) | ((uint32_t)r.query_start ^ 0x80000000u);
    val[p] = (i << 1) | 1u;
  }
}
__device__ __forceinline__ uint32_t entry_tiles(uint32_t n_ops) { return (n_ops + TILE_OPS - 1) / TILE_OPS; }
__global__ __launch_bounds__(256) void ext_counts_kernel(const impg_gpu_record_t *__restrict__ records, const uint32_t *__restrict__ sval,
                                                         uint32_t n_ent, uint32_t *__restrict__ cnt) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= n_ent) return;
  const uint32_t m = entry_tiles(records[sval[e] >> 1].cigar_len);
  cnt[e] = m > INLINE_TILES ? m + 1u : 0u;
}
// the entry payloads in their final order, their checkpoints (from the tiles' own headers), the search columns and
// the scan input of the running maximum: segment (= target id) << 32 | end, order-preserving
__global__ __launch_bounds__(256) void entries_kernel(const impg_gpu_record_t *__restrict__ records, const uint32_t *__restrict__ tile_base,
                                                      const uint32_t *__restrict__ rec_totT, const uint32_t *__restrict__ rec_totQ,
                                                      const unsigned long long *__restrict__ skey, const uint32_t *__restrict__ sval,
                                                      const unsigned long long *__restrict__ ext_off, uint32_t n_ent,
                                                      const uint32_t *__restrict__ pool, Entry *__restrict__ ent, uint32_t *__restrict__ ext_cp,
                                                      int32_t *__restrict__ starts, int32_t *__restrict__ ends, int32_t *__restrict__ ends_t,
                                                      unsigned long long *__restrict__ maxin) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_ent) return;
  const uint32_t v = sval[i], rec = v >> 1;
  const bool rev = (v & 1u) != 0;
  const impg_gpu_record_t r = records[rec];
  Entry e;
  const uint32_t fl = (r.cigar_len & OP_LEN_MASK) | (r.strand ? EF_STRAND : 0u);
  e.tile_base = tile_base[rec];
  if (!rev) {
    e.ts = r.target_start; e.te = r.target_end; e.qs = r.query_start; e.qe = r.query_end;
    e.query_id = r.query_id; e.nops_flags = fl; e.totT = rec_totT[rec]; e.totQ = rec_totQ[rec];
  } else {  // axes swapped (impg.rs:1585-1590)
    e.ts = r.query_start; e.te = r.query_end; e.qs = r.target_start; e.qe = r.target_end;
    e.query_id = r.target_id; e.nops_flags = fl | EF_REVERSED; e.totT = rec_totQ[rec]; e.totQ = rec_totT[rec];
  }
  // effective-order target checkpoints: prefix at the start of effective tile k (index_build.cpp has the derivation)
  const uint32_t n = r.cigar_len, m = entry_tiles(n);
  const bool swp = rev, flip = rev && r.strand;
  const uint32_t *base = pool + (size_t)e.tile_base * TILE_WORDS;
  auto pre = [&](uint32_t k) -> uint32_t {
    if (k == 0) return 0u;
    if (k >= m) return e.totT;
    if (!flip) return base[(size_t)k * TILE_WORDS + (swp ? 1 : 0)];
    return e.totT - base[(size_t)(m - k) * TILE_WORDS + (swp ? 1 : 0)];
  };
  if (m <= INLINE_TILES) {
    for (uint32_t k = 1; k <= 7; k++) e.tcp[k - 1] = k < m ? pre(k) : k == m ? e.totT : 0x7FFFFFFFu;
  } else {
    const uint32_t off = (uint32_t)ext_off[i];
    e.tcp[0] = off;
    for (uint32_t k = 1; k < 7; k++) e.tcp[k] = 0u;
    for (uint32_t k = 0; k <= m; k++) ext_cp[off + k] = pre(k);
  }
  ent[i] = e;
  starts[i] = e.ts;
  ends[i] = e.te;
  ends_t[i] = e.ts < e.te ? e.te : INT32_MIN;
  maxin[i] = (skey[i] & 0xFFFFFFFF00000000ull) | ((uint32_t)e.te ^ 0x80000000u);
}
__global__ __launch_bounds__(256) void pmax_kernel(const unsigned long long *__restrict__ scanned, uint32_t n, int32_t *__restrict__ pmax) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) pmax[i] = (int32_t)((uint32_t)(scanned[i] & 0xFFFFFFFFull) ^ 0x80000000u);
}
// level k of every segment = the last element of every 64-block of level k - 1 (level 0: the columns); one block per segment
__global__ __launch_bounds__(256) void levels_kernel(const SegDesc *__restrict__ seg, uint32_t level, const int32_t *__restrict__ starts,
                                                     const int32_t *__restrict__ pmax, int32_t *__restrict__ starts_lvl,
                                                     int32_t *__restrict__ pmax_lvl) {
  const SegDesc d = seg[blockIdx.x];
  if (level >= d.nlev) return;
  const int32_t *ps = level == 0 ? starts + d.a : starts_lvl + d.off[level - 1], *pp = level == 0 ? pmax + d.a : pmax_lvl + d.off[level - 1];
  const uint32_t pn = level == 0 ? d.n : d.cnt[level - 1];
  for (uint32_t j = threadIdx.x; j < d.cnt[level]; j += 256u) {
    const uint32_t last = min(pn, 64u * (j + 1u)) - 1u;
    starts_lvl[d.off[level] + j] = ps[last];
    pmax_lvl[d.off[level] + j] = pp[last];
  }
}

template <class T> void set_blob(impg_gpu_index &ix, int k, const std::vector<T> &v, size_t &acc) {
  DevBuf &b = *ix.blob(k);
  b.reserve(std::max<size_t>(v.size() * sizeof(T) + 64, 256));
  if (!v.empty()) IMPG_HIP(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ix.blob_bytes[k] = v.size() * sizeof(T);
  acc += v.size() * sizeof(T);
}
void size_blob(impg_gpu_index &ix, int k, size_t bytes, size_t &acc) {
  ix.blob(k)->reserve(std::max<size_t>(bytes + 64, 256));  // + slack: kernels read whole 16-byte vectors
  ix.blob_bytes[k] = bytes;
  acc += bytes;
}
// ---- CIGAR text -> packed ops (parse_cigar_to_delta, impg.rs:2935-2950; host twin: parse_cigar, host_ingest.cpp) -----------
// "[0-9]+[=XIDM]" tokens; every non-digit byte closes an op, digits after the last letter are dropped, a byte that
// is neither a digit nor one of the five letters is the reference's panic (CigarOp::new, impg.rs:88).  One wave per
// record, 64 bytes per step.
__device__ __forceinline__ bool is_digit(unsigned c) { return c - (unsigned)'0' <= 9u; }
__global__ __launch_bounds__(256) void cigar_count_kernel(const char *__restrict__ text, const unsigned long long *__restrict__ boff,
                                                          const uint32_t *__restrict__ blen, uint32_t n_rec, uint32_t *__restrict__ cnt,
                                                          uint32_t *__restrict__ bad) {
  const uint32_t rec = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (rec >= n_rec) return;
  const char *s = text + boff[rec];
  const uint32_t n = blen[rec];
  uint32_t c = 0;
  for (uint32_t b = 0; b < n; b += 64u) {
    const uint32_t i = b + lane();
    bool letter = false;
    if (i < n) {
      const unsigned ch = (unsigned char)s[i];
      letter = !is_digit(ch);
      if (letter && ch != '=' && ch != 'X' && ch != 'I' && ch != 'D' && ch != 'M') *bad = 1u;
    }
    c += (uint32_t)__popcll(__ballot(letter));
  }
  if (lane() == 0) cnt[rec] = c;
}
__global__ __launch_bounds__(256) void cigar_tokens_kernel(const char *__restrict__ text, const unsigned long long *__restrict__ boff,
                                                           const uint32_t *__restrict__ blen, uint32_t n_rec,
                                                           const unsigned long long *__restrict__ op_off, uint32_t *__restrict__ ops) {
  const uint32_t rec = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (rec >= n_rec) return;
  const char *s = text + boff[rec];
  const uint32_t n = blen[rec];
  uint32_t *out = ops + op_off[rec];
  uint32_t done = 0;
  for (uint32_t b = 0; b < n; b += 64u) {
    const uint32_t i = b + lane();
    unsigned ch = '0';
    if (i < n) ch = (unsigned char)s[i];
    const bool letter = i < n && !is_digit(ch);
    const unsigned long long m = __ballot(letter);
    if (letter) {
      uint32_t st = i;  // the digits in front of the letter: back to the previous letter or the record's start ...
      while (st > 0 && is_digit((unsigned char)s[st - 1])) st--;
      uint32_t len = 0;  // ... then forward, len * 10 + d in 32-bit arithmetic like the host's tokenizer
      for (uint32_t k = st; k < i; k++) len = len * 10u + ((unsigned char)s[k] - (unsigned)'0');
      const uint32_t code = ch == '=' ? 0u : ch == 'X' ? 1u : ch == 'I' ? 2u : ch == 'D' ? 3u : 4u;
      out[done + (uint32_t)__popcll(m & ((1ull << lane()) - 1ull))] = (code << 29) | (len & OP_LEN_MASK);
    }
    done += (uint32_t)__popcll(m);
  }
}

}  // namespace

// false: not taken (the caller runs the host builder).  Same inputs and the same resulting arrays as build_index.
bool build_index_device(impg_gpu_index &ix, const impg_gpu_record_t *records, size_t n_records, const uint32_t *cigar_ops, size_t n_ops,
                        const int64_t *seq_len, uint32_t n_seq, bool bidirectional, int order_policy, uint32_t shard, uint32_t n_shards,
                        const uint32_t *owner, const uint32_t *d_cigar_ops) {
  const bool timing = getenv("IMPG_BUILD_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const double t = tnow();
    fprintf(stderr, "[build/device] %-28s %.3f s\n", what, t - t_prev);
    t_prev = t;
  };
  if (n_records >= (1ull << 31)) throw Error{IMPG_E_UNSUPPORTED, "more than 2^31 records in one index"};
  auto is_owned = [&](uint32_t key) { return n_shards == 1 || owner[key] == shard; };
  // ---- host: validation, which records this shard needs, tile bases, entries per target ------------------------------
  std::vector<uint8_t> need(n_records, 0);
  std::vector<uint32_t> tile_base(n_records, 0);
  std::vector<uint32_t> seg_count(n_seq + 1, 0);
  uint64_t n_tiles = 0;
  bool monotone = true;
  uint64_t prev_end = 0;
  for (size_t i = 0; i < n_records; i++) {
    const auto &r = records[i];
    if (r.query_id >= n_seq || r.target_id >= n_seq) throw Error{IMPG_E_INVALID, "record sequence id out of range"};
    if (r.cigar_off + r.cigar_len > n_ops) throw Error{IMPG_E_INVALID, "record CIGAR outside the op pool"};
    if (r.cigar_len > OP_LEN_MASK) throw Error{IMPG_E_UNSUPPORTED, "CIGAR longer than 2^29 ops"};
    if (r.cigar_off < prev_end) monotone = false;
    prev_end = r.cigar_off + r.cigar_len;
    bool nd = false;
    if (is_owned(r.target_id)) { nd = true; seg_count[r.target_id]++; }
    if (bidirectional && r.query_id != r.target_id && is_owned(r.query_id)) { nd = true; seg_count[r.query_id]++; }  // impg.rs:1584
    if (!nd) continue;
    need[i] = 1;
    tile_base[i] = (uint32_t)n_tiles;
    n_tiles += (r.cigar_len + TILE_OPS - 1) / TILE_OPS;
    if (n_tiles >= (1ull << 32) - 2) throw Error{IMPG_E_UNSUPPORTED, "op pool exceeds 2^32 tiles"};
  }
  std::vector<uint32_t> tgt_off(n_seq + 1, 0);
  size_t n_targets = 0;
  for (uint32_t s = 0; s < n_seq; s++) {
    const uint64_t nx = (uint64_t)tgt_off[s] + seg_count[s];
    if (nx >= (1ull << 32) - 1) throw Error{IMPG_E_UNSUPPORTED, "more than 2^32 entries"};
    tgt_off[s + 1] = (uint32_t)nx;
    n_targets += seg_count[s] != 0;
  }
  const size_t n_entries = tgt_off[n_seq];
  // The ops reach the device in batches of consecutive records.  That needs the pool in record order (what every
  // producer in this library hands over); a pool addressed at random is uploaded whole when it fits, else left to the
  // host builder.
  IMPG_HIP(hipSetDevice(ix.device));
  size_t free_b = 0, total_b = 0;
  IMPG_HIP(hipMemGetInfo(&free_b, &total_b));
  // The prefix lines are 40 % of the index and only buy speed (the plain projection reads them instead of replaying
  // ops): an index that does not fit the device with them is built without (IMPG_PREFIX_LINES=0 forces that).
  bool with_pfx = !(getenv("IMPG_PREFIX_LINES") && atoi(getenv("IMPG_PREFIX_LINES")) == 0);
  // (the identity lines beside them are built when a query first needs them: impg_gpu_index::ensure_identity_lines)
  const bool with_idl = getenv("IMPG_IDENTITY_LINES") && atoi(getenv("IMPG_IDENTITY_LINES")) == 1;
  auto bytes_for = [&](bool pf) { return n_tiles * ((pf ? 2 : 1) * TILE_WORDS * 4 + (pf ? (with_idl ? IDL_WORDS * 4 : 0) : TILE_SUBS * 16)) + n_entries * (sizeof(Entry) + 40) + n_records * 64; };
  if (with_pfx && bytes_for(true) + (1ull << 30) > free_b) with_pfx = false;
  const size_t out_bytes = bytes_for(with_pfx);
  if (out_bytes + (1ull << 30) > free_b) return false;  // (the host builder reports the shortage in its own words)
  if (!d_cigar_ops && !monotone && n_ops * 4 + out_bytes + (1ull << 30) > free_b) return false;
  ix.n_records = n_records;
  if (ix.seq.lens.empty()) ix.seq.lens.assign(seq_len, seq_len + n_seq);
  ix.h_tgt_off = tgt_off;
  lap("validate (host)");

  // ---- device: records, tiles -------------------------------------------------------------------------------------
  hipStream_t s = nullptr;
  size_t acc = 0;
  DevBuf d_rec, d_need, d_tb, d_totT, d_totQ, d_owner, d_flag;
  d_rec.reserve(std::max<size_t>(n_records * sizeof(impg_gpu_record_t), 256));
  d_need.reserve(std::max<size_t>(n_records, 256));
  d_tb.reserve(std::max<size_t>(n_records * 4, 256));
  d_totT.reserve(std::max<size_t>(n_records * 4, 256));
  d_totQ.reserve(std::max<size_t>(n_records * 4, 256));
  d_flag.reserve(256);
  IMPG_HIP(hipMemset(d_flag.p, 0, 4));
  if (n_records) {
    IMPG_HIP(hipMemcpy(d_rec.p, records, n_records * sizeof(impg_gpu_record_t), hipMemcpyHostToDevice));
    IMPG_HIP(hipMemcpy(d_need.p, need.data(), n_records, hipMemcpyHostToDevice));
    IMPG_HIP(hipMemcpy(d_tb.p, tile_base.data(), n_records * 4, hipMemcpyHostToDevice));
  }
  if (n_shards > 1) {
    d_owner.reserve(std::max<size_t>((size_t)n_seq * 4, 256));
    IMPG_HIP(hipMemcpy(d_owner.p, owner, (size_t)n_seq * 4, hipMemcpyHostToDevice));
  }
  const uint32_t *dev_owner = n_shards > 1 ? d_owner.as<uint32_t>() : nullptr;
  size_blob(ix, 10, n_tiles * TILE_WORDS * 4, acc);
  size_blob(ix, 14, with_pfx ? n_tiles * TILE_WORDS * 4 : 0, acc);
  uint32_t *const d_pfx = with_pfx ? ix.blob(14)->as<uint32_t>() : nullptr;
  size_blob(ix, 12, n_tiles * (with_pfx ? (with_idl ? IDL_WORDS * 4 : 0) : TILE_SUBS * 16), acc);
  uint4 *const d_idp = (with_pfx && !with_idl) ? nullptr : ix.blob(12)->as<uint4>();
  {
    const size_t BATCH_OPS = 64ull << 20;  // 256 MB of ops per upload
    DevBuf d_ops;
    size_t b0 = 0;
    if (d_cigar_ops) {  // the pool is already here (tokenize_on_device): one launch over all records
      if (n_records)
        tiles_kernel<<<cdiv(n_records, 4), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), d_need.as<uint8_t>(), d_tb.as<uint32_t>(), 0u, (uint32_t)n_records,
                                                          d_cigar_ops, 0ull, ix.blob(10)->as<uint32_t>(), d_pfx,
                                                          d_idp, d_totT.as<uint32_t>(), d_totQ.as<uint32_t>(), d_flag.as<uint32_t>());
      IMPG_HIP(hipStreamSynchronize(s));
      b0 = n_records;
    } else if (!monotone && n_ops) {
      d_ops.reserve(n_ops * 4);
      IMPG_HIP(hipMemcpy(d_ops.p, cigar_ops, n_ops * 4, hipMemcpyHostToDevice));
    }
    while (b0 < n_records) {
      size_t b1 = b0;
      uint64_t lo = records[b0].cigar_off, hi = lo;
      if (monotone) {
        while (b1 < n_records && b1 - b0 < (1u << 30) && (b1 == b0 || records[b1].cigar_off + records[b1].cigar_len - lo <= BATCH_OPS)) {
          hi = records[b1].cigar_off + records[b1].cigar_len;
          b1++;
        }
        const size_t nb = (size_t)(hi - lo);
        d_ops.reserve(std::max<size_t>(nb * 4, 256));
        if (nb) IMPG_HIP(hipMemcpy(d_ops.p, cigar_ops + lo, nb * 4, hipMemcpyHostToDevice));
      } else {
        b1 = std::min(n_records, b0 + (1u << 30));
        lo = 0;
      }
      const uint32_t nr = (uint32_t)(b1 - b0);
      tiles_kernel<<<cdiv(nr, 4), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), d_need.as<uint8_t>(), d_tb.as<uint32_t>(), (uint32_t)b0, nr,
                                                d_ops.as<uint32_t>(), lo, ix.blob(10)->as<uint32_t>(), d_pfx,
                                                d_idp, d_totT.as<uint32_t>(), d_totQ.as<uint32_t>(), d_flag.as<uint32_t>());
      IMPG_HIP(hipStreamSynchronize(s));  // (the next batch overwrites d_ops)
      b0 = b1;
    }
    uint32_t bad = 0;
    IMPG_HIP(hipMemcpy(&bad, d_flag.p, 4, hipMemcpyDeviceToHost));
    if (bad) throw Error{IMPG_E_INVALID, "Invalid CIGAR operation"};
  }
  lap("ops upload + tiles");

  // ---- entries: record order -> (target, start) order ----------------------------------------------------------------
  DevBuf cnt, pos, key, key2, val, val2, tmp, ext_cnt, ext_off, maxin, maxout;
  const size_t eb4 = std::max<size_t>(n_entries * 4, 256), eb8 = std::max<size_t>(n_entries * 8, 256);
  cnt.reserve(std::max<size_t>(n_records * 4, 256)); pos.reserve(std::max<size_t>(n_records * 4, 256));
  key.reserve(eb8); key2.reserve(eb8); val.reserve(eb4); val2.reserve(eb4);
  if (n_records) {
    entry_counts_kernel<<<cdiv(n_records, 256), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), (uint32_t)n_records, bidirectional ? 1 : 0, dev_owner, shard,
                                                             cnt.as<uint32_t>());
    size_t sb = 0;
    IMPG_HIP(rocprim::exclusive_scan(nullptr, sb, cnt.as<uint32_t>(), pos.as<uint32_t>(), 0u, n_records, rocprim::plus<uint32_t>(), s));
    tmp.reserve(std::max<size_t>(sb, 256));
    IMPG_HIP(rocprim::exclusive_scan(tmp.p, sb, cnt.as<uint32_t>(), pos.as<uint32_t>(), 0u, n_records, rocprim::plus<uint32_t>(), s));
    entry_keys_kernel<<<cdiv(n_records, 256), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), (uint32_t)n_records, bidirectional ? 1 : 0, dev_owner, shard,
                                                           pos.as<uint32_t>(), key.as<unsigned long long>(), val.as<uint32_t>());
  }
  if (n_entries) {
    const size_t tb = sort_pairs_scratch_bytes((uint32_t)n_entries);
    tmp.reserve(std::max<size_t>(tb, 256));
    launch_sort_pairs(tmp.p, tb, key.as<unsigned long long>(), key2.as<unsigned long long>(), val.as<uint32_t>(), val2.as<uint32_t>(),
                      (uint32_t)n_entries, 32 + bits_for(n_seq), s);
  }
  // external checkpoints of entries with more than 8 tiles
  ext_cnt.reserve(eb4); ext_off.reserve(eb8 + 8);
  uint64_t n_ext = 0;
  if (n_entries) {
    ext_counts_kernel<<<cdiv(n_entries, 256), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), val2.as<uint32_t>(), (uint32_t)n_entries, ext_cnt.as<uint32_t>());
    size_t sb = 0;
    IMPG_HIP(rocprim::exclusive_scan(nullptr, sb, ext_cnt.as<uint32_t>(), ext_off.as<unsigned long long>(), 0ull, n_entries, rocprim::plus<unsigned long long>(), s));
    tmp.reserve(std::max<size_t>(sb, 256));
    IMPG_HIP(rocprim::exclusive_scan(tmp.p, sb, ext_cnt.as<uint32_t>(), ext_off.as<unsigned long long>(), 0ull, n_entries, rocprim::plus<unsigned long long>(), s));
    unsigned long long last_off = 0;
    uint32_t last_cnt = 0;
    IMPG_HIP(hipMemcpyAsync(&last_off, ext_off.as<unsigned long long>() + (n_entries - 1), 8, hipMemcpyDeviceToHost, s));
    IMPG_HIP(hipMemcpyAsync(&last_cnt, ext_cnt.as<uint32_t>() + (n_entries - 1), 4, hipMemcpyDeviceToHost, s));
    IMPG_HIP(hipStreamSynchronize(s));
    n_ext = last_off + last_cnt;
    if (n_ext >= (1ull << 32)) throw Error{IMPG_E_UNSUPPORTED, "external checkpoint array exceeds 2^32"};
  }
  size_blob(ix, 1, n_entries * 4, acc); size_blob(ix, 2, n_entries * 4, acc); size_blob(ix, 3, n_entries * 4, acc); size_blob(ix, 4, n_entries * 4, acc);
  size_blob(ix, 9, n_entries * sizeof(Entry), acc);
  size_blob(ix, 11, n_ext * 4, acc);
  maxin.reserve(eb8); maxout.reserve(eb8);
  if (n_entries) {
    entries_kernel<<<cdiv(n_entries, 256), 256, 0, s>>>(d_rec.as<impg_gpu_record_t>(), d_tb.as<uint32_t>(), d_totT.as<uint32_t>(), d_totQ.as<uint32_t>(),
                                                        key2.as<unsigned long long>(), val2.as<uint32_t>(), ext_off.as<unsigned long long>(),
                                                        (uint32_t)n_entries, ix.blob(10)->as<uint32_t>(), ix.blob(9)->as<Entry>(),
                                                        ix.blob(11)->as<uint32_t>(), ix.blob(1)->as<int32_t>(), ix.blob(2)->as<int32_t>(),
                                                        ix.blob(3)->as<int32_t>(), maxin.as<unsigned long long>());
    size_t sb = 0;
    IMPG_HIP(rocprim::inclusive_scan(nullptr, sb, maxin.as<unsigned long long>(), maxout.as<unsigned long long>(), n_entries,
                                     rocprim::maximum<unsigned long long>(), s));
    tmp.reserve(std::max<size_t>(sb, 256));
    IMPG_HIP(rocprim::inclusive_scan(tmp.p, sb, maxin.as<unsigned long long>(), maxout.as<unsigned long long>(), n_entries,
                                     rocprim::maximum<unsigned long long>(), s));
    pmax_kernel<<<cdiv(n_entries, 256), 256, 0, s>>>(maxout.as<unsigned long long>(), (uint32_t)n_entries, ix.blob(4)->as<int32_t>());
  }
  lap("entries + columns");

  // ---- host: segment table and visit ranks (functions of the counts alone); levels on the device ----------------------
  std::vector<SegDesc> seg(n_seq);
  uint64_t lvl_total = 0;
  for (uint32_t t = 0; t < n_seq; t++) {
    SegDesc d{};
    d.a = tgt_off[t];
    d.n = tgt_off[t + 1] - tgt_off[t];
    uint32_t c = d.n, lev = 0;
    while (c > 64) {
      if (lev == MAX_LEVELS) throw Error{IMPG_E_UNSUPPORTED, "more than 64^5 entries on one target"};
      c = (c + 63) / 64;
      d.off[lev] = (uint32_t)lvl_total;
      d.cnt[lev] = c;
      lvl_total += c;
      lev++;
    }
    d.nlev = lev;
    if (lvl_total >= (1ull << 32)) throw Error{IMPG_E_UNSUPPORTED, "search levels exceed 2^32"};
    seg[t] = d;
  }
  set_blob(ix, 0, seg, acc);
  size_blob(ix, 5, lvl_total * 4, acc);
  size_blob(ix, 6, lvl_total * 4, acc);
  if (n_seq && lvl_total)
    for (uint32_t level = 0; level < (uint32_t)MAX_LEVELS; level++)
      levels_kernel<<<n_seq, 256, 0, s>>>(ix.blob(0)->as<SegDesc>(), level, ix.blob(1)->as<int32_t>(), ix.blob(4)->as<int32_t>(),
                                          ix.blob(5)->as<int32_t>(), ix.blob(6)->as<int32_t>());
  {
    std::vector<uint32_t> rank(n_entries);
    std::atomic<uint32_t> next{0};
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_seq));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&]() {
        for (;;) {
          const uint32_t q = next.fetch_add(1);
          if (q >= n_seq) break;
          const uint32_t a = tgt_off[q], b = tgt_off[q + 1];
          if (order_policy == IMPG_ORDER_COITREES) coitrees_visit_rank(b - a, rank.data() + a);
          else for (uint32_t i = a; i < b; i++) rank[i] = i - a;
        }
      });
    for (auto &x : th) x.join();
    set_blob(ix, 7, rank, acc);
  }
  // MultiImpg tie order (several alignment files): positions by (file, visit rank in that file's own tree) -- host,
  // from the sorted entries' records (index_build.cpp has the derivation)
  const bool multi_file = ix.file_first.size() > 2;
  if (multi_file) {
    std::vector<uint32_t> sval(n_entries), mrank(n_entries);
    if (n_entries) IMPG_HIP(hipMemcpy(sval.data(), val2.p, n_entries * 4, hipMemcpyDeviceToHost));
    auto file_of = [&](uint32_t rec) {
      return (uint32_t)(std::upper_bound(ix.file_first.begin(), ix.file_first.end(), (uint64_t)rec) - ix.file_first.begin() - 1);
    };
    std::atomic<uint32_t> next{0};
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, n_seq));
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
      th.emplace_back([&]() {
        std::vector<std::pair<uint64_t, uint32_t>> keyv;
        std::vector<uint32_t> vr;
        for (;;) {
          const uint32_t q = next.fetch_add(1);
          if (q >= n_seq) break;
          const uint32_t a = tgt_off[q], b = tgt_off[q + 1];
          if (a == b) continue;
          std::vector<std::pair<uint32_t, uint32_t>> by_file;
          by_file.reserve(b - a);
          for (uint32_t i = a; i < b; i++) by_file.push_back({file_of(sval[i] >> 1), i});
          std::stable_sort(by_file.begin(), by_file.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
          keyv.clear();
          for (size_t u = 0; u < by_file.size();) {
            size_t w = u;
            while (w < by_file.size() && by_file[w].first == by_file[u].first) w++;
            const uint32_t nf = (uint32_t)(w - u);
            vr.resize(nf);
            if (order_policy == IMPG_ORDER_COITREES) coitrees_visit_rank(nf, vr.data());
            else for (uint32_t k = 0; k < nf; k++) vr[k] = k;
            for (uint32_t k = 0; k < nf; k++) keyv.push_back({((uint64_t)by_file[u].first << 32) | vr[k], by_file[u + k].second});
            u = w;
          }
          std::sort(keyv.begin(), keyv.end());
          for (uint32_t k = 0; k < keyv.size(); k++) mrank[keyv[k].second] = k;
        }
      });
    for (auto &x : th) x.join();
    set_blob(ix, 8, mrank, acc);
  }
  std::vector<int32_t> sl(n_seq);
  for (uint32_t q = 0; q < n_seq; q++) sl[q] = (int32_t)std::min<int64_t>(std::max<int64_t>(seq_len[q], 0), INT32_MAX);
  set_blob(ix, 13, sl, acc);
  IMPG_HIP(hipStreamSynchronize(s));
  lap("segments + ranks + levels");
  ix.device_bytes = acc;
  ix.n_entries = n_entries;
  ix.n_tiles = n_tiles;
  ix.n_targets = n_targets;
  ix.multi_file = multi_file;
  ix.tp_mode = false;
  ix.bind_view(n_seq, order_policy == IMPG_ORDER_SORTED);
  return true;
}

// The identity lines of an index that was built without them (tiles_kernel's words, from the op LINES this time): a wave
// per entry walks its record's tiles two at a time, a lane per op slot.  A record's two entries write the same bytes.
__global__ __launch_bounds__(256) void identity_lines_kernel(const Entry *__restrict__ entries, uint32_t n_entries,
                                                             const uint32_t *__restrict__ pool, uint32_t *__restrict__ idl) {
  const uint32_t ei = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (ei >= n_entries) return;
  const Entry en = entries[ei];
  const uint32_t n = en.nops_flags & OP_LEN_MASK;
  const uint32_t l = lane(), h = l >> 5, w = l & 31u, hb = h << 5;
  uint32_t cM = 0, cX = 0, cG = 0;  // running sums of the record before this step's first op
  const uint32_t m = (n + TILE_OPS - 1) / TILE_OPS;
  for (uint32_t j0 = 0; j0 < m; j0 += 2) {
    const uint32_t j = j0 + h;
    const bool tile_on = j < m;
    const uint32_t cnt = tile_on ? min(TILE_OPS, n - j * TILE_OPS) : 0u;
    const uint32_t u = w - 6u;
    const bool on = w >= 6u && u < cnt;
    const size_t tile = (size_t)en.tile_base + j;
    uint32_t dM = 0, dX = 0, dG = 0;
    if (on) {
      const uint32_t op = pool[tile * TILE_WORDS + w];
      const uint32_t code = op >> 29, len = op & OP_LEN_MASK;
      if (code == 0u || code == 4u) dM = len;  // 'M' counted as match (impg.rs:2959)
      else if (code == 1u) dX = len;
      else dG = 1u;                 // gap-compressed: one per 'I' / 'D' op
    }
    const uint32_t iM = cM + wscan(dM), iX = cX + wscan(dX), iG = cG + wscan(dG);
    const uint32_t eM = iM - dM, eX = iX - dX, eG = iG - dG;  // before this lane's op
    const uint32_t last = hb + 5u + max(cnt, 1u);
    const uint32_t endM = from_lane(iM, last), endX = from_lane(iX, last);
    const uint32_t m0 = from_lane(eM, hb + 6u), x0 = from_lane(eX, hb + 6u), g0 = from_lane(eG, hb + 6u);
    const uint32_t ient = ((eM - m0) & 0xFFFFu) | ((eX - x0) << 16), ient_end = ((endM - m0) & 0xFFFFu) | ((endX - x0) << 16);
    const uint32_t ient_up2 = (uint32_t)__shfl_down((int)ient, 2);
    const uint32_t gapmask = (uint32_t)((__ballot(on && dG != 0u) >> (hb + 6u)) & 0x3FFFFFFull);
    uint32_t iw;
    if (w >= IDL_E0) iw = (w - IDL_E0) < cnt ? ient_up2 : ient_end;
    else iw = w == 0u ? m0 : w == 1u ? x0 : w == 2u ? g0 : gapmask;
    if (tile_on) idl[tile * IDL_WORDS + w] = iw;
    cM = from_lane(iM, 63u); cX = from_lane(iX, 63u); cG = from_lane(iG, 63u);
  }
}

uint64_t tokenize_on_device(ParsedPaf &pp, int device, DevBuf &d_ops) {
  IMPG_HIP(hipSetDevice(device));
  const size_t n = pp.records.size();
  uint64_t text_bytes = 0;
  for (auto &t : pp.texts) text_bytes += t.n;
  DevBuf d_text, d_boff, d_blen, d_cnt, d_off, d_bad, tmp;
  d_text.reserve(std::max<size_t>(text_bytes + 64, 256));
  for (auto &t : pp.texts)
    if (t.n) IMPG_HIP(hipMemcpy(d_text.as<char>() + t.base, t.p, t.n, hipMemcpyHostToDevice));
  std::vector<unsigned long long> boff(n);
  std::vector<uint32_t> blen(n);
  for (size_t i = 0; i < n; i++) {
    boff[i] = pp.records[i].cigar_off;
    blen[i] = pp.records[i].cigar_len;
    if (boff[i] + blen[i] > text_bytes) throw Error{IMPG_E_INVALID, "internal: CIGAR text outside the input"};
  }
  d_boff.reserve(std::max<size_t>(n * 8, 256)); d_blen.reserve(std::max<size_t>(n * 4, 256));
  d_cnt.reserve(std::max<size_t>(n * 4, 256)); d_off.reserve(std::max<size_t>(n * 8, 256)); d_bad.reserve(256);
  IMPG_HIP(hipMemset(d_bad.p, 0, 4));
  uint64_t n_ops = 0;
  std::vector<uint32_t> cnt(n);
  std::vector<unsigned long long> off(n);
  if (n) {
    IMPG_HIP(hipMemcpy(d_boff.p, boff.data(), n * 8, hipMemcpyHostToDevice));
    IMPG_HIP(hipMemcpy(d_blen.p, blen.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s = nullptr;
    cigar_count_kernel<<<cdiv(n, 4), 256, 0, s>>>(d_text.as<char>(), d_boff.as<unsigned long long>(), d_blen.as<uint32_t>(), (uint32_t)n,
                                                  d_cnt.as<uint32_t>(), d_bad.as<uint32_t>());
    size_t sb = 0;
    IMPG_HIP(rocprim::exclusive_scan(nullptr, sb, d_cnt.as<uint32_t>(), d_off.as<unsigned long long>(), 0ull, n, rocprim::plus<unsigned long long>(), s));
    tmp.reserve(std::max<size_t>(sb, 256));
    IMPG_HIP(rocprim::exclusive_scan(tmp.p, sb, d_cnt.as<uint32_t>(), d_off.as<unsigned long long>(), 0ull, n, rocprim::plus<unsigned long long>(), s));
    IMPG_HIP(hipMemcpyAsync(cnt.data(), d_cnt.p, n * 4, hipMemcpyDeviceToHost, s));
    IMPG_HIP(hipMemcpyAsync(off.data(), d_off.p, n * 8, hipMemcpyDeviceToHost, s));
    uint32_t bad = 0;
    IMPG_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, s));
    IMPG_HIP(hipStreamSynchronize(s));
    if (bad) throw Error{IMPG_E_INVALID, "Failed to parse PAF: Invalid CIGAR operation"};
    n_ops = off[n - 1] + cnt[n - 1];
    d_ops.reserve(std::max<size_t>(n_ops * 4 + 64, 256));
    cigar_tokens_kernel<<<cdiv(n, 4), 256, 0, s>>>(d_text.as<char>(), d_boff.as<unsigned long long>(), d_blen.as<uint32_t>(), (uint32_t)n,
                                                   d_off.as<unsigned long long>(), d_ops.as<uint32_t>());
    IMPG_HIP(hipStreamSynchronize(s));
  } else d_ops.reserve(256);
  for (size_t i = 0; i < n; i++) {
    pp.records[i].cigar_off = off[i];
    pp.records[i].cigar_len = cnt[i];
  }
  pp.raw = false;
  pp.texts.clear();
  pp.holders.clear();
  return n_ops;
}

}  // namespace impg

void impg_gpu_index::ensure_identity_lines() {
  using namespace impg;
  std::lock_guard<std::mutex> lk(idl_m);
  if (!lacks_identity_lines()) return;
  IMPG_HIP(hipSetDevice(device));
  const size_t bytes = n_tiles * (size_t)IDL_WORDS * 4;
  try {
    d_idp.reserve(std::max<size_t>(bytes + 64, 256));
  } catch (const Error &) {
    throw Error{IMPG_E_OOM, "not enough device memory for the identity lines of this index (" + std::to_string(bytes >> 20) +
                                " MiB, built when a query first filters by min_gap_compressed_identity)"};
  }
  // (on a stream of its own, and only that stream is waited for: the handle's other engines keep running.  A launch failure
  // shows in hipGetLastError, not in the synchronisation: the lines count as built only once both have been checked.)
  struct Stream {
    hipStream_t s = nullptr;
    Stream() { IMPG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
  } st;
  if (n_entries) {
    identity_lines_kernel<<<(unsigned)((n_entries + 3) / 4), 256, 0, st.s>>>(d_entries.as<Entry>(), (uint32_t)n_entries, d_ops.as<uint32_t>(),
                                                                            d_idp.as<uint32_t>());
    IMPG_HIP(hipGetLastError());
  }
  IMPG_HIP(hipStreamSynchronize(st.s));
  view.idp = d_idp.as<uint4>();
  blob_bytes[12] = bytes;
  device_bytes += bytes;
  has_identity_lines.store(true, std::memory_order_release);  // (published last: a reader that sees it sees view.idp)
}
