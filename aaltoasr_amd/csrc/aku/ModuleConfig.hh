// ModuleConfig.hh -- the option block of a feature / model module with the interface of
// aku::ModuleConfig (aku/ModuleConfig.hh, aku/ModuleConfig.cc): ordered name -> value strings,
// typed set ("%d" / "%g") and get (strtol / strtod narrowed to float, whole value or std::string
// thrown), and the "{ name value ... }" text form of .cfg / .spkc files.  Written for the adapters;
// the engine's own parser (csrc/feat_graph.cc) follows the same rules.
#ifndef AASR_AKU_MODULECONFIG_HH
#define AASR_AKU_MODULECONFIG_HH

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace aku {

class ModuleConfig {
 public:
  ModuleConfig() : m_lines(0) {}

  bool exists(const std::string &name) const { return m_index.count(name) != 0; }

  void set(const std::string &name, int value) { put(name, fmt("%d", value)); }
  void set(const std::string &name, float value) { put(name, fmt("%g", (double)value)); }
  void set(const std::string &name, const std::string &value) { put(name, value); }
  void set(const std::string &name, const char *value) { put(name, value); }
  void set(const std::string &name, const std::vector<int> &vec) {
    std::string v;
    for (size_t i = 0; i < vec.size(); i++) v += (i ? " " : "") + fmt("%d", vec[i]);
    put(name, v);
  }
  void set(const std::string &name, const std::vector<float> &vec) {
    std::string v;
    for (size_t i = 0; i < vec.size(); i++) v += (i ? " " : "") + fmt("%g", (double)vec[i]);
    put(name, v);
  }
  void set(const std::string &name, const std::vector<std::string> &vec) {
    std::string v;
    for (size_t i = 0; i < vec.size(); i++) v += (i ? " " : "") + vec[i];
    put(name, v);
  }

  bool get(const std::string &name, int &value) const {
    const std::string *v = find(name);
    if (!v) return false;
    bool ok = true;
    value = (int)to_long(*v, &ok);
    if (!ok) throw std::string("invalid integer value: ") + *v;
    return true;
  }
  bool get(const std::string &name, float &value) const {
    const std::string *v = find(name);
    if (!v) return false;
    bool ok = true;
    value = to_float(*v, &ok);
    if (!ok) throw std::string("invalid float value: ") + *v;
    return true;
  }
  bool get(const std::string &name, std::string &value) const {
    const std::string *v = find(name);
    if (!v) return false;
    value = *v;
    return true;
  }
  bool get(const std::string &name, std::vector<int> &vec) const {
    const std::string *v = find(name);
    if (!v) return false;
    const std::vector<std::string> f = fields(*v);
    vec.resize(f.size());
    for (size_t i = 0; i < f.size(); i++) {
      bool ok = true;
      vec[i] = (int)to_long(f[i], &ok);
      if (!ok) throw std::string("invalid value '") + f[i] + "'in integer vector: " + *v;
    }
    return true;
  }
  bool get(const std::string &name, std::vector<float> &vec) const {
    const std::string *v = find(name);
    if (!v) return false;
    const std::vector<std::string> f = fields(*v);
    vec.resize(f.size());
    for (size_t i = 0; i < f.size(); i++) {
      bool ok = true;
      vec[i] = to_float(f[i], &ok);
      if (!ok) throw std::string("invalid value '") + f[i] + "' in float vector: " + *v;
    }
    return true;
  }
  bool get(const std::string &name, std::vector<std::string> &vec) const {
    const std::string *v = find(name);
    if (!v) return false;
    vec = fields(*v);
    return true;
  }

  /** reads one "{ ... }" block, line by line (aku/ModuleConfig.cc:166-202) */
  void read(FILE *file, bool allow_empty_values = false) {
    m_lines = 0;
    bool opened = false;
    std::string line;
    while (true) {
      if (!read_line(file, line)) throw std::string("unexpected end of module config file");
      m_lines++;
      trim(line);
      if (line.empty()) continue;
      if (!opened) {
        if (line != "{") throw std::string("'{' expected in module config file: ") + line;
        opened = true;
        continue;
      }
      if (line == "}") break;
      const size_t sp = line.find_first_of(" \t");
      if (sp == std::string::npos) {
        // an empty vector held in memory (QuantEqModule::get_parameters before any estimate,
        // aku/FeatureModules.cc:2097-2102) has no text form the reference's reader accepts;
        // the engine hands parameters over as text, so its own blocks may carry one
        if (!allow_empty_values) throw std::string("value missing for option: ") + line;
        if (exists(line)) throw std::string("value redefined: ") + line;
        put(line, "");
        continue;
      }
      const std::string key = line.substr(0, sp);
      const size_t vb = line.find_first_not_of(" \t", sp);
      if (exists(key)) throw std::string("value redefined: ") + line;
      put(key, line.substr(vb));
    }
  }
  int num_lines_read() const { return m_lines; }
  /** the option names in insertion order (adapter-side helper) */
  void get_names(std::vector<std::string> &names) const { names = m_names; }

  void write(FILE *file, int indent = 0) const { fputs(text(indent).c_str(), file); }
  /** the block as text (what write() prints) */
  std::string text(int indent = 0) const {
    const std::string pad((size_t)indent, ' ');
    std::string t = pad + "{\n";
    for (size_t i = 0; i < m_names.size(); i++) t += pad + "  " + m_names[i] + " " + m_values[i] + "\n";
    t += pad + "}\n";
    return t;
  }
  /** parses a block held in memory */
  void read_text(const std::string &text, bool allow_empty_values = false) {
    FILE *tmp = tmpfile();
    if (!tmp) throw std::string("ModuleConfig: tmpfile() failed");
    fputs(text.c_str(), tmp);
    rewind(tmp);
    try {
      read(tmp, allow_empty_values);
    } catch (...) {
      fclose(tmp);
      throw;
    }
    fclose(tmp);
  }

 private:
  static std::string fmt(const char *f, int v) {
    char b[64];
    snprintf(b, sizeof b, f, v);
    return b;
  }
  static std::string fmt(const char *f, double v) {
    char b[64];
    snprintf(b, sizeof b, f, v);
    return b;
  }
  static long to_long(const std::string &s, bool *ok) {
    char *end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    if (s.empty() || *end) *ok = false;
    return v;
  }
  static float to_float(const std::string &s, bool *ok) {
    char *end = nullptr;
    const float v = (float)strtod(s.c_str(), &end);
    if (s.empty() || *end) *ok = false;
    return v;
  }
  static void trim(std::string &s) {
    const size_t a = s.find_first_not_of(" \t");
    if (a == std::string::npos) {
      s.clear();
      return;
    }
    s = s.substr(a, s.find_last_not_of(" \t") - a + 1);
  }
  static bool read_line(FILE *f, std::string &line) {
    line.clear();
    int c;
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
      any = true;
      if (c == '\n') break;
      line.push_back((char)c);
    }
    return any;
  }
  static std::vector<std::string> fields(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
      size_t e = s.find_first_of(" \t", i);
      if (e == std::string::npos) e = s.size();
      out.push_back(s.substr(i, e - i));
      i = s.find_first_not_of(" \t", e);
      if (i == std::string::npos) break;
    }
    return out;
  }
  const std::string *find(const std::string &name) const {
    const auto it = m_index.find(name);
    return it == m_index.end() ? nullptr : &m_values[it->second];
  }
  void put(const std::string &name, const std::string &value) {
    const auto it = m_index.find(name);
    if (it == m_index.end()) {
      m_index[name] = m_values.size();
      m_names.push_back(name);
      m_values.push_back(value);
    } else {
      m_values[it->second] = value;
    }
  }
  std::vector<std::string> m_names, m_values;
  std::map<std::string, size_t> m_index;
  int m_lines;
};

}  // namespace aku

#endif
