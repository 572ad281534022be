// impg-gpu: the `impg query` command line (reference src/main.rs:4259-4381,
// :6513-7520) over libimpg_gpu.so, for the BED output path.  Same flags, same
// defaults, same validation order, same bytes on stdout; every BED row of a
// `-b` file goes to the GPU in ONE batch instead of the reference's serial loop
// (main.rs:7435).
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/impg_gpu.h"

namespace {

[[noreturn]] void die(const std::string &msg, int code = 1) {
  fprintf(stderr, "Error: %s\n", msg.c_str());
  exit(code);
}

// clap's "-d" accepts metric suffixes (50k, 1m): main.rs parse of merge distance
bool parse_metric(const char *s, long long *out) {
  char *end = nullptr;
  double v = strtod(s, &end);
  if (end == s) return false;
  long long mul = 1;
  if (*end == 'k' || *end == 'K') { mul = 1000; end++; }
  else if (*end == 'm' || *end == 'M') { mul = 1000000; end++; }
  else if (*end == 'g' || *end == 'G') { mul = 1000000000; end++; }
  if (*end) return false;
  *out = (long long)llround(v * (double)mul);
  return true;
}

struct Target {
  std::string seq, name;
  int32_t start, end;
};

bool parse_i32(const std::string &s, int32_t *v) {
  if (s.empty()) return false;
  size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
  if (i >= s.size()) return false;
  long long x = 0;
  for (; i < s.size(); i++) {
    if (s[i] < '0' || s[i] > '9') return false;
    x = x * 10 + (s[i] - '0');
    if (x > 2147483648ll) return false;
  }
  if (s[0] == '-') x = -x;
  if (x > 2147483647ll) return false;
  *v = (int32_t)x;
  return true;
}

// parse_bed_file (src/commands/partition.rs:1719-1750)
std::vector<Target> parse_bed_file(const std::string &path) {
  std::ifstream in(path);
  if (!in) die("No such file or directory: " + path);
  std::vector<Target> out;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::vector<std::string> parts;
    size_t st = 0;
    for (size_t i = 0; i <= line.size(); i++)
      if (i == line.size() || line[i] == '\t') { parts.push_back(line.substr(st, i - st)); st = i + 1; }
    if (parts.size() < 3) die("Invalid BED file format");
    Target t;
    if (!parse_i32(parts[1], &t.start)) die("Invalid start value");
    if (!parse_i32(parts[2], &t.end)) die("Invalid end value");
    if (t.start >= t.end) die("Start value must be less than end value");
    t.seq = parts[0];
    std::string nm;
    if (parts.size() > 3) {
      nm = parts[3];
      size_t a = nm.find_first_not_of(" \t\r\n\v\f"), b = nm.find_last_not_of(" \t\r\n\v\f");
      nm = a == std::string::npos ? "" : nm.substr(a, b - a + 1);
    }
    if (nm.empty() || nm == ".") nm = t.seq + ":" + std::to_string(t.start) + "-" + std::to_string(t.end);
    t.name = nm;
    out.push_back(t);
  }
  return out;
}

void usage() {
  fprintf(stderr,
          "impg-gpu index -a <paf>... -i <file> [--unidirectional] [--order coitrees|sorted] [--device N]\n"
          "impg-gpu query (-a <paf>... | -i <file>) (-r seq:start-end | -b <bed>) (-d <bp> | --no-merge) [-x] [-m N]\n"
          "               [--transitive-dfs] [--multi-impg] [--min-transitive-len N] [--min-distance-between-ranges N]\n"
          "               [-l N] [--min-result-identity F] [--subset-sequence-list FILE] [--original-sequence-coordinates]\n"
          "               [--consider-strandness] [-o auto|bed|bedpe|paf] [--unidirectional] [--order coitrees|sorted]\n"
          "               [--device N]\n");
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2 || (strcmp(argv[1], "query") != 0 && strcmp(argv[1], "index") != 0)) {
    usage();
    return 2;
  }
  const bool index_only = strcmp(argv[1], "index") == 0;
  std::string index_file;  // -i: a saved index (impg_gpu_index_save), read if it exists, else written after the build
  std::vector<std::string> pafs;
  std::string range, bed, ofmt = "auto";
  bool have_d = false, no_merge = false, transitive = false, dfs = false, unidirectional = false, multi = false;
  long long merge_d = 0;
  long max_depth = 2, min_tl = -1, mdbr = 10, min_out = -1;
  double min_ident = NAN;
  int verbose = 0;
  bool original_coords = false;  // main.rs:4370
  std::string subset_list;  // --subset-sequence-list: a file of sequence names (main.rs:4357, :11709-11720)
  int device = 0, order = IMPG_ORDER_COITREES;
  // numeric options are parsed whole or refused, as clap does for the reference (a typo must not become 0 = "unlimited")
  auto num = [](const std::string &flag, const char *v, long lo, long hi) -> long {
    char *end = nullptr;
    errno = 0;
    const long x = strtol(v, &end, 10);
    if (errno || end == v || *end != '\0' || x < lo || x > hi) die("invalid value '" + std::string(v) + "' for '" + flag + "'", 2);
    return x;
  };
  auto real = [](const std::string &flag, const char *v) -> double {
    char *end = nullptr;
    errno = 0;
    const double x = strtod(v, &end);
    if (errno || end == v || *end != '\0' || !(x == x)) die("invalid value '" + std::string(v) + "' for '" + flag + "'", 2);
    return x;
  };
  bool consider_strandness = false;  // main.rs:4380
  bool host_merge = false;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    auto need = [&](const char *f) -> const char * {
      if (i + 1 >= argc) die(std::string("a value is required for '") + f + "'", 2);
      return argv[++i];
    };
    if (a == "-a" || a == "--alignment-files") {
      pafs.push_back(need("-a"));
      while (i + 1 < argc && argv[i + 1][0] != '-') pafs.push_back(argv[++i]);
    } else if (a == "-r" || a == "--target-range") range = need("-r");
    else if (a == "-b" || a == "--target-bed") bed = need("-b");
    else if (a == "-d" || a == "--merge-distance") {
      if (!parse_metric(need("-d"), &merge_d) || merge_d < 0 || merge_d > 2147483647ll) die("invalid value for '-d'", 2);
      have_d = true;
    } else if (a == "--no-merge") no_merge = true;
    else if (a == "-x" || a == "--transitive") transitive = true;
    else if (a == "--transitive-dfs") dfs = true;
    else if (a == "--multi-impg") multi = true;  // per-file indices (the reference's MultiImpg, src/multi_impg.rs)
    else if (a == "-m" || a == "--max-depth") max_depth = num(a, need("-m"), 0, 65535);           // u16 (main.rs:4263)
    else if (a == "--min-transitive-len") min_tl = num(a, need(a.c_str()), 0, 2147483647);
    else if (a == "--min-distance-between-ranges") mdbr = num(a, need(a.c_str()), 0, 2147483647);
    else if (a == "-l" || a == "--min-output-length") min_out = num(a, need("-l"), 0, 2147483647);
    else if (a == "--min-result-identity") min_ident = real(a, need(a.c_str()));
    else if (a == "--consider-strandness") consider_strandness = true;
    else if (a == "--host-merge") host_merge = true;  // BED merges on the host (impg_gpu_results_bed) instead of the device
    else if (a == "--subset-sequence-list") subset_list = need(a.c_str());
    else if (a == "--original-sequence-coordinates") original_coords = true;
    else if (a == "-o" || a == "--output-format") ofmt = need("-o");
    else if (a == "--unidirectional") unidirectional = true;
    else if (a == "--device") device = (int)num(a, need(a.c_str()), 0, 1023);
    else if (a == "--order") {
      std::string o = need("--order");
      if (o != "sorted" && o != "coitrees") die("invalid value '" + o + "' for '--order'", 2);
      order = o == "sorted" ? IMPG_ORDER_SORTED : IMPG_ORDER_COITREES;
    }
    else if (a == "-i" || a == "--index") index_file = need("-i");
    else if (a == "-v" || a == "--verbose") verbose = (int)num(a, need(a.c_str()), 0, 9);  // >= 1: phase timings on stderr
    else if (a == "-t" || a == "--threads") need(a.c_str());  // accepted, unused
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else die("unexpected argument '" + a + "'", 2);
  }
  auto file_exists = [](const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; };
  if (index_only) {  // `impg index` (main.rs:11321-11386): build once, keep the result
    if (pafs.empty() || index_file.empty()) die("impg-gpu index needs --alignment-files and --index", 2);
    std::vector<const char *> pp;
    for (auto &p : pafs) pp.push_back(p.c_str());
    impg_gpu_index_t *ix = nullptr;
    if (impg_gpu_index_create_from_paf(pp.data(), (int)pp.size(), unidirectional ? 0 : 1, order, device, &ix) != IMPG_OK)
      die(impg_gpu_last_error());
    if (impg_gpu_index_save(ix, index_file.c_str()) != IMPG_OK) die(impg_gpu_last_error());
    impg_gpu_index_destroy(ix);
    return 0;
  }
  const bool load_saved = !index_file.empty() && file_exists(index_file);
  if (pafs.empty() && !load_saved) die("the following required arguments were not provided: --alignment-files", 2);
  if (range.empty() == bed.empty()) die("exactly one of --target-range and --target-bed is required", 2);
  if (!have_d && !no_merge)  // MERGE_DISTANCE_REQUIRED_TEXT (main.rs:4288-4315)
    die("-d/--merge-distance is required. For `impg query`, pass `-d <bp>`. Use `--no-merge` to explicitly disable merging.");
  const int32_t merge_distance = no_merge ? -1 : (int32_t)merge_d;
  // -o auto: bed for -r, bedpe for -b (main.rs:7365-7373)
  std::string fmt = ofmt == "auto" ? (bed.empty() ? "bed" : "bedpe") : ofmt;
  if (fmt != "bed" && fmt != "bedpe" && fmt != "paf")
    die("output format '" + fmt + "' is not built in impg-gpu (bed, bedpe, paf)");

  std::vector<const char *> pp;
  for (auto &p : pafs) pp.push_back(p.c_str());
  impg_gpu_index_t *ix = nullptr;
  bool loaded = false;
  bool reference_index = false;  // -i names the reference's own IMPGIDX2 / IMPGIDX1 file: read it, never overwrite it
  if (load_saved) {
    char magic[8] = {0};
    if (FILE *mf = fopen(index_file.c_str(), "rb")) { (void)!fread(magic, 1, 8, mf); fclose(mf); }
    reference_index = memcmp(magic, "IMPGIDX", 7) == 0;
  }
  if (load_saved && reference_index) {
    if (pafs.empty()) die("an IMPG index file stores offsets into its alignment files: pass them with -a, in the order it was built with");
    if (impg_gpu_index_load_impg(index_file.c_str(), pp.data(), (int)pp.size(), order, device, &ix) != IMPG_OK) die(impg_gpu_last_error());
    loaded = true;
  } else if (load_saved) {
    if (impg_gpu_index_load(index_file.c_str(), device, &ix) == IMPG_OK) loaded = true;
    else if (pafs.empty()) die(impg_gpu_last_error());
    else fprintf(stderr, "[impg-gpu] %s: %s -- rebuilding it from the alignment files\n", index_file.c_str(), impg_gpu_last_error());
  }
  if (!loaded) {
    if (impg_gpu_index_create_from_paf(pp.data(), (int)pp.size(), unidirectional ? 0 : 1, order, device, &ix) != IMPG_OK)
      die(impg_gpu_last_error());
    if (!index_file.empty() && impg_gpu_index_save(ix, index_file.c_str()) != IMPG_OK) die(impg_gpu_last_error());
  }

  std::vector<Target> targets;
  if (!range.empty()) {
    char name[4096];
    int32_t s, e;
    if (range.find(':') == std::string::npos) {  // no interval: the whole sequence [0, len) (main.rs:7290-7310)
      const long long id = impg_gpu_seq_id(ix, range.c_str());
      if (id < 0) die("Sequence '" + range + "' not found in index");
      const long long len = impg_gpu_seq_len(ix, (uint32_t)id);
      targets.push_back({range, range + ":0-" + std::to_string(len), 0, (int32_t)len});
    } else {
      if (impg_gpu_parse_target_range(range.c_str(), name, sizeof name, &s, &e) != IMPG_OK) die(impg_gpu_last_error());
      targets.push_back({name, std::string(name) + ":" + std::to_string(s) + "-" + std::to_string(e), s, e});
    }
  } else {
    targets = parse_bed_file(bed);
  }
  const int32_t eff_min_tl = min_tl < 0 ? 101 : (int32_t)min_tl;  // effective_min_transitive_len (main.rs:4283)
  std::vector<impg_gpu_range_t> ranges;
  std::vector<const char *> names;
  for (auto &t : targets) {
    // validate_sequence_range (main.rs:10458-10520), validate_range_min_length (:10387-10403), perform_query bound (:11632)
    long long id = impg_gpu_seq_id(ix, t.seq.c_str());
    if (id < 0) die("Sequence '" + t.seq + "' not found in index");
    long long len = impg_gpu_seq_len(ix, (uint32_t)id);
    if (t.start < 0) die("Start position " + std::to_string(t.start) + " cannot be negative");
    if (t.end < 0) die("End position " + std::to_string(t.end) + " cannot be negative");
    if (t.start >= t.end) die("Start position must be less than end position");
    if (t.end > len) die("End position " + std::to_string(t.end) + " exceeds sequence length " + std::to_string(len));
    if (t.end - t.start < eff_min_tl)
      die("Range '" + t.name + "' (" + std::to_string(t.end - t.start) + " bp) is below minimum of " +
          std::to_string(eff_min_tl) + " bp. Lower --min-transitive-len or use a longer range");
    ranges.push_back({(uint32_t)id, t.start, t.end});
    names.push_back(t.name.c_str());
  }
  impg_gpu_params_t p;
  memset(&p, 0, sizeof p);
  p.transitive = transitive || dfs;  // --transitive-dfs implies a transitive query
  p.dfs = dfs;
  p.multi_impg = multi;
  p.max_depth = (uint32_t)max_depth;
  p.min_transitive_len = eff_min_tl;
  p.min_distance_between_ranges = (int32_t)mdbr;
  p.min_output_length = min_out < 0 ? -1 : (int32_t)min_out;
  p.min_identity = min_ident;
  p.store_cigar = fmt != "bed";  // CIGARs for PAF / BEDPE only (main.rs:7447)
  p.original_sequence_coordinates = original_coords;
  p.consider_strandness = consider_strandness;
  impg_gpu_results_t *res = nullptr;
  std::vector<uint8_t> keep;
  if (!subset_list.empty()) {  // load_subset_filter (subset_filter.rs:63-82) + one matches() per sequence
    FILE *f = fopen(subset_list.c_str(), "rb");
    if (!f) die("Failed to read subset sequence list '" + subset_list + "'");
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    const uint32_t ns = impg_gpu_num_seqs(ix);
    std::vector<const char *> nm(ns);
    for (uint32_t i = 0; i < ns; i++) nm[i] = impg_gpu_seq_name(ix, i);
    keep.resize(ns);
    size_t entries = 0;
    if (impg_gpu_subset_keep(text.data(), text.size(), nm.data(), ns, keep.data(), &entries) != IMPG_OK) die(impg_gpu_last_error());
    if (entries == 0) die("Subset sequence list '" + subset_list + "' did not contain any sequence names");
  }
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  if (fmt == "bed" && !host_merge) {  // both merges on the device: only merged rows cross PCIe
    uint64_t len = 0;
    double sec[3] = {0, 0, 0};
    fflush(stdout);
    if (impg_gpu_query_batch_bed_fd(ix, ranges.data(), ranges.size(), &p, keep.empty() ? nullptr : keep.data(), merge_distance, names.data(),
                                    fileno(stdout), &len, sec) != IMPG_OK)
      die(impg_gpu_last_error());
    const double t1 = now();
    if (verbose >= 1)
      fprintf(stderr, "[impg-gpu] query + merge + text + write %.2f s (engine %.2f, device merge %.2f, device text + copy + write %.2f), %llu bytes\n",
              t1 - t0, sec[0], sec[1], sec[2], (unsigned long long)len);
    _exit(0);  // (tearing down gigabytes of host buffers and the device context is not worth a second)
  }
  if (impg_gpu_query_batch_filtered(ix, ranges.data(), ranges.size(), &p, nullptr, keep.empty() ? nullptr : keep.data(), &res) != IMPG_OK)
    die(impg_gpu_last_error());
  const double t1 = now();
  char *text = nullptr;
  size_t len = 0;
  const int rc = fmt == "bed" ? impg_gpu_results_bed(res, ix, names.data(), &p, merge_distance, &text, &len)
                              : impg_gpu_results_paf(res, ix, names.data(), &p, merge_distance,
                                                     fmt == "paf" ? IMPG_OUT_PAF : IMPG_OUT_BEDPE, &text, &len);
  if (rc != IMPG_OK) die(impg_gpu_last_error());
  const double t2 = now();
  fwrite(text, 1, len, stdout);
  fflush(stdout);
  if (verbose >= 1) {
    double eng = 0, asm_s = 0;
    impg_gpu_results_timing(res, &eng, &asm_s);
    fprintf(stderr, "[impg-gpu] query %.2f s (engine %.2f, result assembly %.2f), merge + text %.2f s, write %.2f s, %zu intervals, %zu bytes\n",
            t1 - t0, eng, asm_s, t2 - t1, now() - t2, impg_gpu_results_total(res), len);
  }
  free(text);
  impg_gpu_results_free(res);
  impg_gpu_index_destroy(ix);
  return 0;
}
