// str.hh -- string helpers with the interface of aku/str.hh that the tools on the scoring path
// call: str::fmt (phone_probs.cc:121), str2float/str2long, clean, split.  The engine's own recipe
// and configuration parsers (csrc/pipeline.cc, csrc/feat_graph.cc) are pinned against aku/str.cc
// compiled in place (tests/test_recipe_host.py); these inline versions follow the same rules.
#ifndef AKU_AMD_STR_HH
#define AKU_AMD_STR_HH

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace str {

/** printf into a std::string of at most `size` characters (aku/str.cc:11-24) */
inline std::string fmt(size_t size, const char *f, ...) {
  std::vector<char> buf(size + 1);
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf.data(), buf.size(), f, ap);
  va_end(ap);
  return std::string(buf.data());
}

/** removes leading and trailing characters of `chars` (aku/str.cc:124-140) */
inline void clean(std::string &s, const char *chars) {
  const std::string c(chars);
  const size_t a = s.find_first_not_of(c);
  if (a == std::string::npos) {
    s.clear();
    return;
  }
  s = s.substr(a, s.find_last_not_of(c) - a + 1);
}

/** aku/str.cc:142-172: one delimiter (a run of them with `group`) ends a field; no empty last
 * field after a trailing delimiter; with num_fields > 0 the last field takes the rest */
inline std::vector<std::string> split(const std::string &s, const char *delims, bool group,
                                      unsigned int num_fields = 0) {
  std::vector<std::string> fields;
  const std::string d(delims);
  size_t begin = 0;
  while (begin < s.size()) {
    if (num_fields > 0 && fields.size() == num_fields - 1) {
      fields.push_back(s.substr(begin));
      break;
    }
    size_t end = begin;
    while (end < s.size() && d.find(s[end]) == std::string::npos) end++;
    fields.push_back(s.substr(begin, end - begin));
    end++;
    if (group)
      while (end < s.size() && d.find(s[end]) != std::string::npos) end++;
    begin = end;
  }
  return fields;
}

/** strtol / strtod over the whole string; *ok = false otherwise (aku/str.hh:73-93) */
inline long str2long(const char *s, bool *ok) {
  char *end;
  const long v = strtol(s, &end, 10);
  if (*s == 0 || *end != 0) *ok = false;
  return v;
}
inline double str2float(const char *s, bool *ok) {
  char *end;
  const double v = strtod(s, &end);
  if (*s == 0 || *end != 0) *ok = false;
  return v;
}
inline long str2long(const std::string &s, bool *ok) { return str2long(s.c_str(), ok); }
inline double str2float(const std::string &s, bool *ok) { return str2float(s.c_str(), ok); }

}  // namespace str

#endif
