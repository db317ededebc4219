// str.hh -- string helpers with the interface of aku/str.hh (namespace aku::str, the pointer-taking
// signatures of aku/str.hh:13-93) for the callers that compile against these adapters:
// phone_probs.cc / feacat.cc (str::fmt), PhnReader.cc (read_line, chomp, split, str2long), align.cc.
// Written from the documented behaviour of aku/str.cc; the engine's own recipe and configuration
// parsers (csrc/pipeline.cc, csrc/feat_graph.cc) are pinned against aku/str.cc compiled in place
// (tests/test_recipe_host.py) and tests/test_reference_callers.py runs these against the same.
#ifndef AKU_AMD_STR_HH
#define AKU_AMD_STR_HH

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace aku {
namespace str {

/** printf into a std::string of at most size - 1 characters (aku/str.cc:13-24) */
inline std::string fmt(size_t size, const char *f, ...) {
  std::vector<char> buf(size > 0 ? size : 1);
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf.data(), buf.size(), f, ap);
  va_end(ap);
  return std::string(buf.data());
}

/** removes one trailing newline (aku/str.cc:113-121) */
inline void chomp(std::string *s) {
  if (!s->empty() && (*s)[s->size() - 1] == '\n') s->resize(s->size() - 1);
}

/** one line of any length, newline kept unless do_chomp; false at end of file with nothing read or
 * on a read error (aku/str.cc:26-52) */
inline bool read_line(std::string *s, FILE *file = stdin, bool do_chomp = false) {
  char buf[4096];
  s->erase();
  while (fgets(buf, sizeof buf, file)) {
    s->append(buf);
    if ((*s)[s->size() - 1] == '\n') break;
  }
  if (ferror(file) || s->empty()) return false;
  if (do_chomp) chomp(s);
  return true;
}

/** exactly `length` bytes (aku/str.cc:54-80) */
inline bool read_string(std::string *s, size_t length, FILE *file = stdin) {
  s->erase();
  s->resize(length);
  if (length == 0) return true;
  if (fread(&(*s)[0], length, 1, file) != 1) {
    s->erase();
    return false;
  }
  return true;
}

/** the rest of the file, or at most `length` bytes when length > 0 (aku/str.cc:82-111) */
inline bool read_file(std::string *s, FILE *file, size_t length = 0) {
  char buf[4096];
  s->clear();
  for (;;) {
    size_t want = sizeof buf;
    if (length > 0 && s->size() + want > length) want = length - s->size();
    const size_t got = fread(buf, 1, want, file);
    s->append(buf, got);
    if (got < sizeof buf) return !ferror(file);
    if (s->size() == length) return true;
  }
}

/** strips leading and trailing characters of `chars` (aku/str.cc:123-140) */
inline void clean(std::string *s, const char *chars) {
  size_t e = s->size();
  while (e > 0 && strchr(chars, (*s)[e - 1])) e--;
  s->erase(e);
  size_t b = 0;
  while (b < s->size() && strchr(chars, (*s)[b])) b++;
  s->erase(0, b);
}
inline void clean(std::string &s, const char *chars) { clean(&s, chars); }

/** aku/str.cc:142-172: one delimiter (a run of them with `group`) ends a field; no empty last
 * field after a trailing delimiter; with num_fields > 0 the last field takes the rest */
inline void split(const std::string *s, const char *delims, bool group, std::vector<std::string> *fields,
                  int num_fields = 0) {
  fields->clear();
  size_t begin = 0;
  while (begin < s->size()) {
    if (num_fields > 0 && (int)fields->size() == num_fields - 1) {
      fields->push_back(s->substr(begin));
      break;
    }
    size_t end = begin;
    while (end < s->size() && !strchr(delims, (*s)[end])) end++;
    fields->push_back(s->substr(begin, end - begin));
    end++;
    if (group)
      while (end < s->size() && strchr(delims, (*s)[end])) end++;
    begin = end;
  }
}
inline std::vector<std::string> split(const std::string &s, const char *delims, bool group,
                                      unsigned int num_fields = 0) {
  std::vector<std::string> fields;
  split(&s, delims, group, &fields, (int)num_fields);
  return fields;
}

/** strtol / strtod over the whole string; *ok = false otherwise; a value at the type's limit ends
 * the program and str2float rounds through float, as aku/str.cc:243-283 does */
inline long str2long(const char *s, bool *ok) {
  char *end;
  const long v = strtol(s, &end, 10);
  if (v == LONG_MIN || v == LONG_MAX) {
    fprintf(stderr, "str2long(): value out of range\n");
    exit(1);
  }
  if (*s == 0 || *end != 0) *ok = false;
  return v;
}
inline double str2float(const char *s, bool *ok) {
  char *end;
  const float v = (float)strtod(s, &end);
  if (v == HUGE_VALF || v == -HUGE_VALF) {
    fprintf(stderr, "str2float(): value out of range\n");
    exit(1);
  }
  if (*s == 0 || *end != 0) *ok = false;
  return v;
}
inline long str2long(const std::string *s, bool *ok) { return str2long(s->c_str(), ok); }
inline double str2float(const std::string *s, bool *ok) { return str2float(s->c_str(), ok); }
inline long str2long(const std::string &s, bool *ok) { return str2long(s.c_str(), ok); }
inline double str2float(const std::string &s, bool *ok) { return str2float(s.c_str(), ok); }

}  // namespace str
}  // namespace aku

// the engine's own sources call these as ::str
namespace str = aku::str;

#endif
