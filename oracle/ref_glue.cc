// ref_glue.cc -- extern "C" shims over the few reference translation units
// that compile from their own sources in this image (ModuleConfig.cc, str.cc,
// util.cc + util.hh).  TEST INFRASTRUCTURE ONLY: built into
// oracle/_ref/libaku_ref.so by oracle/Makefile when /root/reference is
// present, used to pin the oracle's .cfg parser, str2float and safe_log
// against the real reference code.  Everything else on the hot path
// (FeatureModules.cc, Distributions.cc, HmmSet.cc, AudioReader.cc ...) needs
// LapackPP / libsndfile, which this image lacks -> unbuildable here.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ModuleConfig.hh"
#include "str.hh"
#include "util.hh"

extern "C" {

double ref_safe_log(double x) { return util::safe_log(x); }

// str::str2float (aku/str.cc:260-282): strtod narrowed to float.
double ref_str2float(const char *s, int *ok)
{
  bool b = true;
  double v = aku::str::str2float(s, &b);
  *ok = b ? 1 : 0;
  return v;
}

// str::split / str::clean (aku/str.cc:124-172) as the recipe and speaker-configuration readers
// call them.  Fields come back joined by 0x1f; the return value is the number of fields.
int ref_str_split(const char *s, const char *delims, int group, int num_fields, char *buf, int buflen)
{
  std::string str(s);
  std::vector<std::string> fields;
  aku::str::split(&str, delims, group != 0, &fields, num_fields);
  std::string out;
  for (size_t i = 0; i < fields.size(); i++) {
    if (i) out += "\x1f";
    out += fields[i];
  }
  snprintf(buf, buflen, "%s", out.c_str());
  return (int)fields.size();
}

int ref_str_clean(const char *s, const char *chars, char *buf, int buflen)
{
  std::string str(s);
  aku::str::clean(&str, chars);
  snprintf(buf, buflen, "%s", str.c_str());
  return (int)str.size();
}

// Parse ONE "{ key value ... }" block from a file positioned just after the
// "module" keyword line (ModuleConfig::read, aku/ModuleConfig.cc:166-202) and
// return "key\x1fvalue\x1e..." in buf.  Returns number of lines consumed or
// -1 on a thrown error (message in buf).
int ref_module_config_read(const char *path, long offset, char *buf, int buflen,
                           long *new_offset)
{
  FILE *fp = fopen(path, "r");
  if (!fp) { snprintf(buf, buflen, "cannot open %s", path); return -1; }
  fseek(fp, offset, SEEK_SET);
  aku::ModuleConfig cfg;
  try {
    cfg.read(fp);
  } catch (std::string &e) {
    snprintf(buf, buflen, "%s", e.c_str());
    fclose(fp);
    return -1;
  }
  *new_offset = ftell(fp);
  fclose(fp);
  // ModuleConfig has no iterator; round-trip through its own writer.
  FILE *tmp = tmpfile();
  cfg.write(tmp, 0);
  long n = ftell(tmp);
  rewind(tmp);
  if (n >= buflen) n = buflen - 1;
  size_t got = fread(buf, 1, (size_t)n, tmp);
  buf[got] = 0;
  fclose(tmp);
  return cfg.num_lines_read();
}

// Typed getters through the reference's own conversion code.
int ref_module_config_get_floats(const char *block_text, const char *key,
                                 float *out, int maxn)
{
  FILE *tmp = tmpfile();
  fputs(block_text, tmp);
  rewind(tmp);
  aku::ModuleConfig cfg;
  try {
    cfg.read(tmp);
  } catch (std::string &e) {
    fclose(tmp);
    return -2;
  }
  fclose(tmp);
  std::vector<float> v;
  try {
    if (!cfg.get(key, v)) return -1;
  } catch (std::string &e) {
    return -2;
  }
  int n = (int)v.size();
  for (int i = 0; i < n && i < maxn; i++) out[i] = v[i];
  return n;
}

}  // extern "C"
