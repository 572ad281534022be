// A built index written to / read from one file: the device arrays exactly as they sit in HBM
// plus the host-side sequence table.  The reference keeps its index in a `.impg` file for the same reason
// (impg.rs:1655-1721 / :1787-1850): parsing and tokenising the alignments is the expensive part of start-up and is
// done once.  The reference's file stores byte offsets into the PAF and re-reads the CIGAR text on demand; this one
// stores the tokenised, tiled ops themselves, so a load is one read + one host-to-device copy per array and the PAF
// is not needed again.  The two formats are not interchangeable.
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>

#include "impg_internal.hpp"

namespace impg {
namespace {

constexpr char MAGIC[8] = {'I', 'M', 'P', 'G', 'H', 'B', 'M', '1'};
constexpr uint32_t VERSION = 6;  // 2: checksum of the arrays behind the end mark; 3: tile padding words carry length 0, prefix lines;
                                 // 4: the checksum also covers the header and the host tables, and the per-target offsets are mandatory;
                                 // 5: indexes with prefix lines carry identity lines (IDL_*) instead of per-sub-tile identity prefixes
                                 // 6: ... or none (flag 32: they are built on the device when a query first filters by identity; a
                                 //    version-5 reader would reject such a file as a size mismatch instead of by its version)

struct Header {  // fixed-size, little-endian (gfx950 hosts are x86-64)
  char magic[8];
  uint32_t version, tile_words, tile_ops, tile_subs, entry_bytes, n_seq, sorted_order, multi_file;
  uint64_t n_records, n_entries, n_tiles, n_targets, n_names, n_file_first, n_tgt_off;
  uint64_t blob_bytes[impg_gpu_index::N_BLOBS];
};

uint64_t fnv64(uint64_t h, const void *p, size_t n);
constexpr uint64_t FNV_SEED = 0xCBF29CE484222325ull;
struct File {
  FILE *f = nullptr;
  std::string path;
  uint64_t sum = FNV_SEED;  // over everything written / read so far, call by call (save and load make the same calls)
  File(const char *p, const char *mode) : path(p) {
    f = fopen(p, mode);
    if (!f) throw Error{IMPG_E_IO, "cannot open " + path + ": " + strerror(errno)};
  }
  ~File() { if (f) fclose(f); }
  void write(const void *p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) throw Error{IMPG_E_IO, "short write to " + path};
    sum = fnv64(sum, p, n);
  }
  void read(void *p, size_t n) {
    if (n && fread(p, 1, n, f) != n) throw Error{IMPG_E_IO, path + " is truncated"};
    sum = fnv64(sum, p, n);
  }
};

uint64_t fnv64(uint64_t h, const void *p, size_t n) {  // FNV-1a over 8-byte words (+ the tail bytes): damage inside an array is caught
  const unsigned char *b = static_cast<const unsigned char *>(p);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, b + i, 8);
    h = (h ^ w) * 0x100000001B3ull;
  }
  for (; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
  return h;
}

}  // namespace

// shard: this file is one rank's part of an index sharded over GPUs (its world, rank and the target -> rank map follow the
// host tables); front: the handle that fronts the shards of one process -- host tables only, no arrays.
void save_index(const impg_gpu_index &cix, const char *path, const ShardInfo *shard, bool front) {
  impg_gpu_index &ix = const_cast<impg_gpu_index &>(cix);  // (blob() is not const; nothing is modified)
  IMPG_HIP(hipSetDevice(ix.device));
  Header h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, MAGIC, 8);
  h.version = VERSION;
  h.tile_words = TILE_WORDS; h.tile_ops = TILE_OPS; h.tile_subs = TILE_SUBS; h.entry_bytes = sizeof(Entry);
  h.n_seq = ix.view.n_seq; h.sorted_order = ix.view.sorted_order; h.multi_file = (ix.multi_file ? 1 : 0) | (ix.tp_mode ? 2 : 0) | ((!ix.tp_mode && ix.n_tiles && !ix.blob_bytes[14]) ? 4 : 0)  // 4: no prefix lines
                                                                            | (shard ? 8 : 0) | (front ? 16 : 0) | (ix.lacks_identity_lines() ? 32 : 0);  // 32: prefix lines without identity lines
  h.n_records = ix.n_records; h.n_entries = ix.n_entries; h.n_tiles = ix.n_tiles; h.n_targets = ix.n_targets;
  h.n_names = ix.seq.names.size(); h.n_file_first = ix.file_first.size(); h.n_tgt_off = ix.h_tgt_off.size();
  for (int k = 0; k < impg_gpu_index::N_BLOBS; k++) h.blob_bytes[k] = front ? 0 : ix.blob_bytes[k];
  // written next to the target and renamed over it: an interrupted save never leaves a truncated cache behind
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  struct Unlink {
    const std::string &p;
    bool keep = false;
    ~Unlink() { if (!keep) (void)unlink(p.c_str()); }
  } guard{tmp};
  if (ix.h_tgt_off.size() != (size_t)ix.view.n_seq + 1) throw Error{IMPG_E_INVALID, "internal: the index has no per-target offsets"};
  File out(tmp.c_str(), "wb");
  out.write(&h, sizeof h);
  out.write(ix.seq.lens.data(), ix.seq.lens.size() * sizeof(int64_t));
  for (const std::string &nm : ix.seq.names) {
    const uint32_t n = (uint32_t)nm.size();
    out.write(&n, 4);
    out.write(nm.data(), n);
  }
  out.write(ix.file_first.data(), ix.file_first.size() * sizeof(uint64_t));
  out.write(ix.h_tgt_off.data(), ix.h_tgt_off.size() * sizeof(uint32_t));
  if (shard) {
    const uint32_t hd[3] = {shard->world, shard->rank, (uint32_t)shard->owner.size()};
    out.write(hd, sizeof hd);
    out.write(shard->owner.data(), shard->owner.size() * sizeof(uint32_t));
  }
  std::vector<char> buf;
  for (int k = 0; k < impg_gpu_index::N_BLOBS; k++) {
    const size_t n = front ? 0 : ix.blob_bytes[k];
    buf.resize(n);
    if (n) IMPG_HIP(hipMemcpy(buf.data(), ix.blob(k)->p, n, hipMemcpyDeviceToHost));
    out.write(buf.data(), n);
  }
  const uint64_t tail = 0x454E44474D50ull ^ h.n_entries;  // an end mark: a file cut short is caught even if sizes line up
  out.write(&tail, 8);
  const uint64_t sum = out.sum;  // header, sequence table, file and target tables, arrays, end mark
  out.write(&sum, 8);
  if (fflush(out.f) != 0 || fsync(fileno(out.f)) != 0) throw Error{IMPG_E_IO, "cannot flush " + out.path};
  fclose(out.f);
  out.f = nullptr;
  if (rename(tmp.c_str(), path) != 0) throw Error{IMPG_E_IO, std::string("cannot rename onto ") + path + ": " + strerror(errno)};
  guard.keep = true;
}

void load_index(impg_gpu_index &ix, const char *path, ShardInfo *shard) {
  File in(path, "rb");
  Header h;
  in.read(&h, sizeof h);
  if (memcmp(h.magic, MAGIC, 8) != 0) throw Error{IMPG_E_INVALID, in.path + " is not a saved impg-gpu index"};
  if (h.version != VERSION || h.tile_words != TILE_WORDS || h.tile_ops != TILE_OPS || h.tile_subs != TILE_SUBS ||
      h.entry_bytes != sizeof(Entry))
    throw Error{IMPG_E_UNSUPPORTED, in.path + " was written by a build with a different index layout: rebuild it"};
  if (h.n_names != 0 && h.n_names != h.n_seq) throw Error{IMPG_E_INVALID, in.path + ": inconsistent sequence table"};
  const bool has_shard = (h.multi_file & 8) != 0, front = (h.multi_file & 16) != 0;
  if (has_shard && !shard) throw Error{IMPG_E_INVALID, in.path + " is a part of an index sharded over GPUs: load it with impg_gpu_index_load_rank / _load_multi"};
  if (!has_shard && shard) throw Error{IMPG_E_INVALID, in.path + " is not a part of a sharded index"};
  if (front && !has_shard) throw Error{IMPG_E_INVALID, in.path + ": a front file without a shard section (damaged header)"};
  // every array's size follows from the counts in the header: check them all before anything is allocated or uploaded
  {
    if (fseek(in.f, 0, SEEK_END) != 0) throw Error{IMPG_E_IO, "cannot seek in " + in.path};
    const uint64_t file_bytes = (uint64_t)ftell(in.f);
    if (fseek(in.f, (long)sizeof h, SEEK_SET) != 0) throw Error{IMPG_E_IO, "cannot seek in " + in.path};
    const uint64_t E = h.n_entries, T = h.n_tiles, S = h.n_seq;
    if (S > 0x7FFFFFFFull || E >= 0xFFFFFFF0ull || T >= 0xFFFFFFF0ull || S * 8 > file_bytes)
      throw Error{IMPG_E_INVALID, in.path + ": counts out of range"};
    const uint64_t want[impg_gpu_index::N_BLOBS] = {S * sizeof(SegDesc), E * 4, E * 4, E * 4, E * 4, h.blob_bytes[5], h.blob_bytes[5],
                                                    E * 4, (h.multi_file & 1) ? E * 4 : 0, E * sizeof(Entry), T * TILE_WORDS * 4,
                                                    h.blob_bytes[11], (h.multi_file & (2 | 32)) ? 0 : (h.multi_file & 4) ? T * TILE_SUBS * 16 : T * IDL_WORDS * 4, S * 4,
                                                    (h.multi_file & 6) ? 0 : T * TILE_WORDS * 4};
    uint64_t total = 0;
    for (int k = 0; k < impg_gpu_index::N_BLOBS; k++) {
      if (front ? h.blob_bytes[k] != 0 : h.blob_bytes[k] != want[k] || (h.blob_bytes[k] & 3)) throw Error{IMPG_E_INVALID, in.path + ": array sizes do not match the header"};
      total += h.blob_bytes[k];
    }
    if (total > file_bytes) throw Error{IMPG_E_INVALID, in.path + " is truncated"};
    // (the small-batch path sizes its projection grid from these offsets, Engine::run_small: they are not optional)
    if (h.n_tgt_off != S + 1) throw Error{IMPG_E_INVALID, in.path + ": bad table sizes"};
  }
  IMPG_HIP(hipSetDevice(ix.device));
  ix.n_records = h.n_records; ix.n_entries = h.n_entries; ix.n_tiles = h.n_tiles; ix.n_targets = h.n_targets;
  ix.multi_file = (h.multi_file & 1) != 0;
  ix.tp_mode = (h.multi_file & 2) != 0;
  ix.seq.lens.resize(h.n_seq);
  in.read(ix.seq.lens.data(), (size_t)h.n_seq * sizeof(int64_t));
  ix.seq.names.clear();
  ix.seq.name_to_id.clear();
  for (uint64_t i = 0; i < h.n_names; i++) {
    uint32_t n = 0;
    in.read(&n, 4);
    if (n > (1u << 20)) throw Error{IMPG_E_INVALID, in.path + ": unreasonable sequence name length"};
    std::string nm(n, '\0');
    in.read(nm.data(), n);
    ix.seq.name_to_id.emplace(nm, (uint32_t)i);
    ix.seq.names.push_back(std::move(nm));
  }
  if (h.n_file_first > h.n_records + 2 || h.n_tgt_off > (uint64_t)h.n_seq + 1) throw Error{IMPG_E_INVALID, in.path + ": bad table sizes"};
  ix.file_first.resize(h.n_file_first);
  in.read(ix.file_first.data(), h.n_file_first * sizeof(uint64_t));
  ix.h_tgt_off.resize(h.n_tgt_off);
  in.read(ix.h_tgt_off.data(), h.n_tgt_off * sizeof(uint32_t));
  for (size_t i = 0; i < ix.h_tgt_off.size(); i++)
    if ((i && ix.h_tgt_off[i] < ix.h_tgt_off[i - 1]) || ix.h_tgt_off[i] > h.n_entries || (i == 0 && ix.h_tgt_off[0] != 0))
      throw Error{IMPG_E_INVALID, in.path + ": target offsets are not ascending within the entry count"};
  if (!ix.h_tgt_off.empty() && ix.h_tgt_off.back() != h.n_entries) throw Error{IMPG_E_INVALID, in.path + ": target offsets do not end at the entry count"};
  if (has_shard) {
    uint32_t hd[3] = {0, 0, 0};
    in.read(hd, sizeof hd);
    if (hd[0] < 1 || hd[1] >= hd[0] || hd[2] != h.n_seq) throw Error{IMPG_E_INVALID, in.path + ": bad shard section"};
    shard->world = hd[0]; shard->rank = hd[1];
    shard->owner.resize(hd[2]);
    in.read(shard->owner.data(), (size_t)hd[2] * sizeof(uint32_t));
    for (uint32_t o : shard->owner)
      if (o >= hd[0]) throw Error{IMPG_E_INVALID, in.path + ": a target is owned by a rank outside the world"};
    shard->front = front;
  }
  std::vector<char> buf;
  size_t acc = 0;
  for (int k = 0; k < impg_gpu_index::N_BLOBS; k++) {
    const size_t n = h.blob_bytes[k];
    if (n > (1ull << 40)) throw Error{IMPG_E_INVALID, in.path + ": unreasonable array size"};
    buf.resize(n);
    in.read(buf.data(), n);
    if (k == 0 && !front) {  // the segment table addresses every other array: it must stay inside them
      const SegDesc *sg = reinterpret_cast<const SegDesc *>(buf.data());
      const uint64_t lvl_words = h.blob_bytes[5] / 4;
      for (uint32_t t = 0; t < h.n_seq; t++) {
        bool ok = (uint64_t)sg[t].a + sg[t].n <= h.n_entries && sg[t].nlev <= (uint32_t)MAX_LEVELS;
        for (uint32_t l = 0; ok && l < sg[t].nlev; l++) ok = (uint64_t)sg[t].off[l] + sg[t].cnt[l] <= lvl_words;
        if (!ok) throw Error{IMPG_E_INVALID, in.path + ": a segment points outside the arrays"};
      }
    }
    DevBuf &b = *ix.blob(k);
    if (n || k != 8) b.reserve(std::max<size_t>(n + 64, 256));  // (array 8, mrank, only exists for several files)
    if (n) IMPG_HIP(hipMemcpy(b.p, buf.data(), n, hipMemcpyHostToDevice));
    ix.blob_bytes[k] = n;
    acc += n;
  }
  uint64_t tail = 0;
  in.read(&tail, 8);
  if (tail != (0x454E44474D50ull ^ h.n_entries)) throw Error{IMPG_E_INVALID, in.path + " is damaged (end mark)"};
  const uint64_t sum = in.sum;
  uint64_t want_sum = 0;
  in.read(&want_sum, 8);
  if (want_sum != sum) throw Error{IMPG_E_INVALID, in.path + " is damaged (checksum of the header, tables and arrays)"};
  ix.device_bytes = acc;
  if (front) { ix.view.n_seq = h.n_seq; return; }  // (no arrays: the shards hold them)
  ix.bind_view(h.n_seq, h.sorted_order);
}

}  // namespace impg
