// --subset-sequence-list: which sequence names does a subset list select?  Host string work on names (never on the
// device); the verdict per sequence id is what impg_gpu_query_batch_filtered takes.
// Behaviour of parse_subset_filter / SubsetFilter::matches (subset_filter.rs:23-60, :117-176).
#include <cstring>
#include <set>
#include <string>
#include <unordered_set>

#include "impg_internal.hpp"

namespace impg {
namespace {

// A name's (sample, haplotype digits) key: "<sample>_hap<digits>...", else "<sample>#<digits...>#...", else the bare
// name when it carries no coordinates.  has_hap is false when no digits follow.
struct SampleKey {
  bool ok = false, has_hap = false;
  std::string sample, hap;
};
std::string leading_digits(const std::string &s, size_t from, size_t to) {
  size_t e = from;
  while (e < to && s[e] >= '0' && s[e] <= '9') e++;
  return s.substr(from, e - from);
}
bool blank(const std::string &s) {
  for (char c : s)
    if (!(c == ' ' || (c >= '\t' && c <= '\r'))) return false;
  return true;
}
SampleKey sample_key(const std::string &name) {
  SampleKey k;
  const size_t hp = name.find("_hap");
  const size_t hash = name.find('#');
  if (hp != std::string::npos) {
    k.ok = true;
    k.sample = name.substr(0, hp);
    k.hap = leading_digits(name, hp + 4, name.size());
  } else if (hash != std::string::npos) {
    k.ok = true;
    k.sample = name.substr(0, hash);
    size_t stop = name.find('#', hash + 1);
    if (stop == std::string::npos) stop = name.size();
    k.hap = leading_digits(name, hash + 1, stop);
  } else if (name.find(':') == std::string::npos && !blank(name)) {
    k.ok = true;
    k.sample = name;
  }
  k.has_hap = !k.hap.empty();
  return k;
}
std::string before_colon(const std::string &s) { return s.substr(0, s.find(':')); }

struct SubsetList {
  std::unordered_set<std::string> exact, stripped, samples;
  std::set<std::pair<std::string, std::string>> sample_haps;
  bool by_key(const std::string &name) const {
    const SampleKey k = sample_key(name);
    if (!k.ok) return false;
    if (k.has_hap && sample_haps.count({k.sample, k.hap})) return true;
    return samples.count(k.sample) != 0;
  }
  bool selects(const std::string &name) const {
    if (exact.count(name)) return true;
    const std::string bare = before_colon(name);
    if (bare != name && exact.count(bare)) return true;
    return stripped.count(bare) || by_key(bare) || by_key(name);
  }
};

SubsetList read_list(const char *text, size_t len) {
  SubsetList L;
  size_t p = 0;
  while (p < len) {
    size_t e = p;
    while (e < len && text[e] != '\n') e++;
    size_t a = p, b = e;  // trim (ASCII white space; also takes the '\r' of a CRLF line)
    while (a < b && (text[a] == ' ' || (text[a] >= '\t' && text[a] <= '\r'))) a++;
    while (b > a && (text[b - 1] == ' ' || (text[b - 1] >= '\t' && text[b - 1] <= '\r'))) b--;
    p = e + 1;
    if (a == b || text[a] == '#') continue;
    const std::string entry(text + a, b - a);
    L.exact.insert(entry);
    const std::string bare = before_colon(entry);
    L.stripped.insert(bare);
    const SampleKey k = sample_key(bare);
    if (k.ok) {
      if (k.has_hap) L.sample_haps.insert({k.sample, k.hap});
      else L.samples.insert(k.sample);
    }
  }
  return L;
}

}  // namespace

size_t subset_select(const char *text, size_t len, const char *const *names, size_t n, uint8_t *keep) {
  const SubsetList L = read_list(text, len);
  for (size_t i = 0; i < n; i++) keep[i] = names[i] && L.selects(names[i]) ? 1 : 0;
  return L.exact.size();
}

}  // namespace impg
