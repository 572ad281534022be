// Internal declarations shared by the host side of libimpg_gpu.so.
// Layouts here are the HBM data layout described in DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/impg_gpu.h"

namespace impg {

// ---- errors ---------------------------------------------------------------
void set_error(const std::string &msg);
struct Error {
  int code;
  std::string msg;
};
#define IMPG_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw impg::Error{_e == hipErrorOutOfMemory ? IMPG_E_OOM : IMPG_E_HIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(_e)};                 \
  } while (0)

// ---- packed CIGAR ops (CigarOp, impg.rs:75-140) ----------------------------
constexpr uint32_t OP_LEN_MASK = (1u << 29) - 1;
constexpr uint32_t OP_PAD = 0xFFFFFFFFu;  // tile padding, never a valid op (code 7)
#ifndef IMPG_TILE_OPS
#define IMPG_TILE_OPS 32
#endif
constexpr uint32_t TILE_OPS = IMPG_TILE_OPS;  // ops per tile (32 = one 128-byte line)
static_assert(TILE_OPS % 4 == 0 && TILE_OPS >= 8 && TILE_OPS <= 64, "tile size");
constexpr uint32_t HIT_NONE = 0xFFFFFFFFu;

// ---- device index (HBM layout) ---------------------------------------------
// One 32-byte payload per index entry, stored in per-target start order.
struct alignas(16) Entry {
  int32_t ts, te, qs, qe;  // target_start/end, query_start/end of the ENTRY (already swapped for reversed entries)
  uint32_t query_id;
  uint32_t tile_base;      // first tile of the record's ops in the op pool
  uint32_t nops_flags;     // bits 0..28 n_ops, bit 30 = strand reverse, bit 31 = reversed entry (REVERSED_BIT)
  uint32_t cp_base;        // first checkpoint of the record (= tile_base + record rank)
};
static_assert(sizeof(Entry) == 32, "entry payload is 32 bytes");
constexpr uint32_t EF_STRAND = 1u << 30;
constexpr uint32_t EF_REVERSED = 1u << 31;

struct DeviceIndexView {  // passed by value to kernels
  const uint32_t *tgt_off;   // [n_seq+1] per-target segment table (replaces ForestMap)
  const int32_t *starts;     // [n_entries] t_start, ascending within a segment
  const int32_t *ends;       // [n_entries] t_end
  const int32_t *pmax;       // [n_entries] running max of t_end within the segment
  const uint32_t *rank;      // [n_entries] visit rank within the segment (order policy)
  const Entry *entries;      // [n_entries]
  const uint32_t *ops;       // [n_tiles*32] packed ops, each record padded to whole tiles
  const uint2 *cp;           // [n_tiles + n_records] per-tile prefix (sum target_delta, sum |query_delta|), +1 total per record
  const int32_t *seq_len;    // [n_seq]
  uint32_t n_seq;
  uint32_t n_entries;
  uint32_t sorted_order;     // 1 = rank is the identity (IMPG_ORDER_SORTED)
};

struct HostSeqIndex {  // SequenceIndex (seqidx.rs)
  std::vector<std::string> names;  // may be empty when created from raw records
  std::vector<int64_t> lens;
  std::unordered_map<std::string, uint32_t> name_to_id;
  uint32_t get_or_insert(const std::string &name, int64_t len);
};

// ---- simple growable device buffer ------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  void reserve(size_t bytes);  // contents are NOT preserved on growth
  void release();
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};

// ---- host ingest -------------------------------------------------------------
struct ParsedPaf {
  HostSeqIndex seq;
  std::vector<impg_gpu_record_t> records;
  std::vector<uint32_t> ops;
};
// parse_paf + parse_cigar_to_delta over whole files (paf.rs:118-194, impg.rs:2935-2950)
void parse_paf_files(const std::vector<std::string> &paths, ParsedPaf &out);
void parse_paf_text(const char *text, size_t len, ParsedPaf &out);
long parse_cigar(const char *s, size_t n, uint32_t *out, size_t cap);

// visit rank of each sorted position of an n-entry segment (order policy)
void coitrees_visit_rank(uint32_t n, uint32_t *rank_out);

// ---- BED (bed.cpp) -----------------------------------------------------------
size_t bed_merge(impg_gpu_interval_t *iv, size_t n, int32_t merge_distance, bool merge_strands);

struct Engine;  // engine.hpp

}  // namespace impg

// ---- the opaque handles --------------------------------------------------------
struct impg_gpu_index {
  int device = 0;
  hipStream_t stream = nullptr;
  impg::HostSeqIndex seq;
  size_t n_records = 0, n_entries = 0, n_tiles = 0, n_targets = 0;
  std::vector<uint32_t> h_tgt_off;
  impg::DevBuf d_tgt_off, d_starts, d_ends, d_pmax, d_rank, d_entries, d_ops, d_cp, d_seq_len;
  impg::DeviceIndexView view{};
  size_t device_bytes = 0;
  impg::Engine *engine = nullptr;  // scratch + streams (engine.cpp)
  ~impg_gpu_index();
};

struct impg_gpu_results {
  std::vector<uint64_t> offsets;
  std::vector<impg_gpu_interval_t> intervals;
  std::vector<impg_gpu_range_t> ranges;
  uint64_t projected = 0;
};
