// speaker_config.cc -- per-speaker / per-utterance parameters (aasr_spkc).
//
// Host-side restatement of aku::SpeakerConfig (aku/SpeakerConfig.cc:8-381), the
// ModelTransformer gateway and ConstrainedMllr's parameter handling
// (aku/ModelModules.cc:11-97, 129-232), i.e. what `phone_probs -S FILE` does
// before each recipe line (aku/phone_probs.cc:94-95, 191-196):
//   speaker_conf.set_speaker(info.speaker_id);
//   if (!info.utterance_id.empty()) speaker_conf.set_utterance(info.utterance_id);
//
// Reference behaviours kept on purpose:
//  * set_speaker() first reads the parameters of the CURRENT speaker back from
//    the modules (get_parameters, "%g" formatting) into that speaker's entry, so
//    a speaker that is set again gets the 6-significant-digit round trip of its
//    values -- and a normalization block that used `var` then also holds `scale`
//    and is refused ("Both scale and var ...").
//  * retrieve_utterance_config() calls set_parameters (not get_parameters).
//  * module entries are applied in the order of a std::map over the full key
//    line ("feature <name>" / "model <name>").
//  * CMLLR matrix entries pass through str2float (float precision); transforms
//    are applied in the order of a std::map over their unit-element lists, later
//    ones overriding earlier ones on shared Gaussians; a speaker whose id did not
//    change keeps the loaded model transform even if its parameters did.
// The device sees a parameter only when its text differs from what was applied
// last, and `before_change` (the recipe runner's flush) runs before the first
// such change, so utterances with identical settings stay in one device batch.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <set>
#include <sstream>

#include "feat.h"
#include "gmm.h"
#include "pipeline.h"

namespace aasr {

typedef std::map<std::string, ModuleConfig> ModuleMap;
typedef std::map<std::string, ModuleMap> SpeakerMap;

enum UnitMode { UNIT_PHONE = 0, UNIT_MIX = 1, UNIT_GAUSSIAN = 2, UNIT_NO = 3 };

}  // namespace aasr

struct aasr_spkc {
  aasr_feat *feat = nullptr;
  aasr_gmm *gmm = nullptr;
  aasr::SpeakerMap speakers, utterances;
  aasr::ModuleMap default_speaker, default_utterance;
  bool default_speaker_set = false, default_utterance_set = false;
  std::string cur_speaker, cur_utterance;
  // ModelTransformer + its one module type, ConstrainedMllr
  bool has_cmllr = false;    // module("cmllr") has been requested
  bool trans_is_reset = true;
  bool cmllr_loaded = false;
  int unit_mode = aasr::UNIT_NO;
  std::map<std::vector<std::string>, std::vector<double>> trans;  // W [dim][dim+1]
  bool device_adapted = false;
  // change tracking
  std::map<std::string, std::string> applied;
  std::function<void()> before_change;
  int64_t changes = 0;
};

namespace aasr {

static std::string clean_ws(const std::string &s) { return str_clean(s, " \t"); }

// str::split(&line, " \t", true, &fields, limit) (aku/str.cc:142-172)
static std::vector<std::string> split_fields(const std::string &s, int limit = 0) {
  return str_split(s, " \t", true, limit);
}

static bool next_line(const std::string &text, size_t *pos, std::string *line) {
  if (*pos >= text.size()) return false;
  size_t e = text.find('\n', *pos);
  if (e == std::string::npos) e = text.size();
  *line = text.substr(*pos, e - *pos);
  *pos = e + 1;
  return true;
}

static std::string canonical(const ModuleConfig &c) {
  std::string t;
  for (size_t i = 0; i < c.names.size(); i++) t += c.names[i] + "\x1f" + c.values[i] + "\x1e";
  return t;
}

static void note_change(aasr_spkc *h) {
  if (h->before_change) h->before_change();
  h->changes++;
}

// ModelTransformer::module (aku/ModelModules.cc:20-40)
static void model_module(aasr_spkc *h, const std::string &name) {
  if (name != "cmllr") raise(AASR_ERR_INVALID, "unknown model module requested: %s", name.c_str());
  h->has_cmllr = true;
}

static void feature_module_exists(aasr_spkc *h, const std::string &name) {
  if (!h->feat->by_name.count(name))
    raise(AASR_ERR_INVALID, "unknown module requested: %s", name.c_str());
}

// SpeakerConfig::read_speaker_file (aku/SpeakerConfig.cc:20-153)
void spkc_read_text(aasr_spkc *h, const std::string &text) {
  size_t pos = 0;
  std::string line;
  int lineno = 0;
  while (next_line(text, &pos, &line)) {
    lineno++;
    line = clean_ws(line);
    if (line.empty()) continue;
    std::vector<std::string> fields = split_fields(line);
    if (fields.size() != 2 || (fields[0] != "speaker" && fields[0] != "utterance"))
      raise(AASR_ERR_INVALID, "SpeakerConfig: Syntax error on line %d: %s", lineno, line.c_str());
    const bool fetch_default = fields[1] == "default";
    const bool fetch_speaker = fields[0] == "speaker";
    ModuleMap *target = nullptr;
    if (fetch_speaker) {
      if (fetch_default && h->default_speaker_set)
        raise(AASR_ERR_INVALID,
              "SpeakerConfig: Default speaker configuration already defined, redefinition on line %d: %s",
              lineno, line.c_str());
      if (!fetch_default) target = &h->speakers[fields[1]];
      else {
        h->default_speaker_set = true;
        target = &h->default_speaker;
      }
    } else {
      if (fetch_default && h->default_utterance_set)
        raise(AASR_ERR_INVALID,
              "SpeakerConfig: Default utterance configuration already defined, redefinition on line %d: %s",
              lineno, line.c_str());
      if (!fetch_default) target = &h->utterances[fields[1]];
      else {
        h->default_utterance_set = true;
        target = &h->default_utterance;
      }
    }
    while (next_line(text, &pos, &line)) {
      lineno++;
      line = clean_ws(line);
      if (line.empty()) continue;
      if (line != "{")
        raise(AASR_ERR_INVALID, "'{' expected in speaker config file: %s", line.c_str());
      break;
    }
    while (next_line(text, &pos, &line)) {
      lineno++;
      line = clean_ws(line);
      if (line.empty()) continue;
      if (line == "}") break;
      // module line: "[feature|model] name"; a lone name is a feature module
      std::vector<std::string> parts = split_fields(line, 2);
      if (parts.size() < 2) {
        line = "feature " + line;
        parts = split_fields(line, 2);
      } else if (parts[0] != "model" && parts[0] != "feature") {
        raise(AASR_ERR_INVALID, "SpeakerConfig: Unknown module namespace at line %d", lineno);
      }
      try {
        if (parts[0] == "feature") feature_module_exists(h, parts[1]);
        if (parts[0] == "model") model_module(h, parts[1]);
      } catch (Error &e) {
        raise(e.code, "SpeakerConfig: error on line %d: %s", lineno, e.msg.c_str());
      }
      ModuleConfig config;
      const size_t before = pos;
      try {
        config.read(text, &pos);
      } catch (Error &e) {
        raise(e.code, "SpeakerConfig: Failed reading module parameters around line %d: %s", lineno,
              e.msg.c_str());
      }
      for (size_t i = before; i < pos && i < text.size(); i++)
        if (text[i] == '\n') lineno++;
      target->insert(ModuleMap::value_type(line, config));  // first definition wins
    }
  }
}

// ---- ConstrainedMllr ----------------------------------------------------------

// ConstrainedMllr::set_parameters (aku/ModelModules.cc:62-97)
static void cmllr_set_parameters(aasr_spkc *h, const ModuleConfig &params) {
  if (!h->gmm) raise(AASR_ERR_INVALID, "SpeakerConfig: model parameters without a model");
  h->trans.clear();
  std::string unit_mode;
  params.get("unitmode", unit_mode);
  if (unit_mode == "UNIT_NO") h->unit_mode = UNIT_NO;
  if (unit_mode == "UNIT_GAUSSIAN") h->unit_mode = UNIT_GAUSSIAN;
  if (unit_mode == "UNIT_MIX") h->unit_mode = UNIT_MIX;
  if (unit_mode == "UNIT_PHONE") h->unit_mode = UNIT_PHONE;
  const int dim = h->gmm->dim;
  const size_t matrix_dim = (size_t)dim * (dim + 1);
  const size_t required = h->unit_mode == UNIT_NO ? matrix_dim : matrix_dim + 1;
  for (int i = 1;; i++) {
    char key[32];
    snprintf(key, sizeof key, "w%d", i);
    std::vector<std::string> parts;
    if (!params.get(key, parts)) break;
    if (parts.size() < required)
      raise(AASR_ERR_INVALID, "ERROR: not enough elements for matrix %s", key);
    const size_t unit_elements = parts.size() - matrix_dim;
    std::vector<std::string> elements(parts.begin(), parts.begin() + unit_elements);
    std::vector<double> d(matrix_dim);
    for (size_t k = 0; k < matrix_dim; k++) {
      const std::string &t = parts[unit_elements + k];
      char *end;
      d[k] = (float)strtod(t.c_str(), &end);  // str::str2float: float precision
      if (t.empty() || *end != '\0') raise(AASR_ERR_INVALID, "invalid value: %s", t.c_str());
    }
    h->trans[elements] = d;
  }
  if (h->unit_mode == UNIT_NO && h->trans.size() > 1)
    raise(AASR_ERR_INVALID,
          "ERROR: speaker can only contain one transform when UNIT_NO (global transform) is set");
}

// ConstrainedMllr::get_parameters (aku/ModelModules.cc:129-160)
static void cmllr_get_parameters(const aasr_spkc *h, ModuleConfig &params) {
  int i = 1;
  for (auto &kv : h->trans) {
    std::string line;
    for (auto &e : kv.first) line += (line.empty() ? "" : " ") + e;
    for (double v : kv.second) {
      char buf[64];
      snprintf(buf, sizeof buf, "%g", v);
      line += (line.empty() ? "" : " ") + std::string(buf);
    }
    char key[32];
    snprintf(key, sizeof key, "w%d", i++);
    params.insert(key, line);
  }
  static const char *names[] = {"UNIT_PHONE", "UNIT_MIX", "UNIT_GAUSSIAN", "UNIT_NO"};
  params.insert("unitmode", names[h->unit_mode]);
}

// Hmm::get_center_phone (aku/HmmSet.cc:21-40)
static std::string center_phone(const std::string &label) {
  const size_t p1 = label.find_last_of('-'), p2 = label.find_first_of('+');
  const bool h1 = p1 != std::string::npos, h2 = p2 != std::string::npos;
  std::string t;
  if (h1 && h2) {
    if (p2 > p1 + 1) t = label.substr(p1 + 1, p2 - p1 - 1);
  } else if (h1) {
    t = label.substr(p1 + 1);
  } else if (h2) {
    t = label.substr(0, p2);
  } else {
    t = label;
  }
  if (t.empty()) raise(AASR_ERR_INVALID, "Invalid phone label %s", label.c_str());
  return t;
}

// RegClassTree::Unit*::get_gaussians (aku/RegClassTree.cc:301-322, 367-385, 443-454, 473-479)
static void unit_gaussians(const aasr_spkc *h, const std::vector<std::string> &elems,
                           std::set<int32_t> &out) {
  const HostModel &m = h->gmm->host;
  out.clear();
  auto add_mixture = [&](long mix) {
    if (mix < 0 || mix >= m.S)
      raise(AASR_ERR_INVALID, "CMLLR unit refers to mixture %ld outside the model", mix);
    for (int32_t k = m.mix_off[mix]; k < m.mix_off[mix + 1]; k++) out.insert(m.mix_idx[k]);
  };
  switch (h->unit_mode) {
    case UNIT_NO:
      for (int64_t g = 0; g < m.G; g++) out.insert((int32_t)g);
      break;
    case UNIT_GAUSSIAN:
      for (auto &e : elems) {
        long g = strtol(e.c_str(), nullptr, 10);
        if (g < 0 || g >= m.G)
          raise(AASR_ERR_INVALID, "CMLLR unit refers to Gaussian %ld outside the pool", g);
        out.insert((int32_t)g);
      }
      break;
    case UNIT_MIX:
      for (auto &e : elems) {
        char *end;
        long mix = strtol(e.c_str(), &end, 10);
        if (e.empty() || *end != '\0') continue;  // str2long failure: skipped like the reference
        add_mixture(mix);
      }
      break;
    case UNIT_PHONE:
      if (m.hmm_label.empty())
        raise(AASR_ERR_UNSUPPORTED,
              "UNIT_PHONE transforms need the HMM inventory of a .ph file (model was built in memory)");
      for (size_t hi = 0; hi < m.hmm_label.size(); hi++) {
        const std::string c = center_phone(m.hmm_label[hi]);
        bool hit = false;
        for (auto &e : elems) hit = hit || e == c;
        if (!hit) continue;
        for (int32_t st : m.hmm_states[hi]) add_mixture(st);
      }
      break;
  }
}

// ModelTransformer::load_transforms -> ConstrainedMllr::load_transform
// (aku/ModelModules.cc:42-50, 164-232)
static void load_transforms(aasr_spkc *h) {
  if (h->has_cmllr && !h->cmllr_loaded) {
    aasr_gmm *g = h->gmm;
    if (!g) raise(AASR_ERR_INVALID, "SpeakerConfig: model transform without a model");
    if (h->trans.empty()) {
      if (h->device_adapted) {
        note_change(h);
        gmm_set_transforms(g, 0, nullptr, nullptr);
        h->device_adapted = false;
      }
    } else {
      const int dim = g->dim;
      std::vector<int32_t> g2t((size_t)g->G, -1);
      std::vector<double> W;
      int t = 0;
      std::set<int32_t> gs;
      for (auto &kv : h->trans) {  // map order; later transforms override
        unit_gaussians(h, kv.first, gs);
        for (int32_t gi : gs) g2t[(size_t)gi] = t;
        W.insert(W.end(), kv.second.begin(), kv.second.end());
        t++;
      }
      note_change(h);
      (void)dim;
      gmm_set_transforms(g, t, g2t.data(), W.data());
      h->device_adapted = true;
    }
    h->cmllr_loaded = true;
  }
  h->trans_is_reset = false;
}

// ModelTransformer::reset_transforms (:52-60).  The device model is only touched
// by the load that always follows in set_speaker().
static void reset_transforms(aasr_spkc *h) {
  if (!h->trans_is_reset && h->has_cmllr) {
    h->cmllr_loaded = false;
    h->trans_is_reset = true;
  }
}

// ---- SpeakerConfig ------------------------------------------------------------

static void apply_feature(aasr_spkc *h, const std::string &module, const ModuleConfig &c) {
  const std::string text = canonical(c);
  auto it = h->applied.find(module);
  if (it != h->applied.end() && it->second == text) return;
  note_change(h);
  // the module may refuse the block (dimension checks); nothing is recorded then
  h->applied.erase(module);
  feat_set_parameters(h->feat, module, c);
  h->applied[module] = text;
}

// SpeakerConfig::set_modules (aku/SpeakerConfig.cc:365-378)
static void set_modules(aasr_spkc *h, const ModuleMap &modules) {
  for (auto &kv : modules) {
    std::vector<std::string> parts = split_fields(kv.first, 2);
    if (parts[0] == "feature") apply_feature(h, parts[1], kv.second);
    if (parts[0] == "model") cmllr_set_parameters(h, kv.second);
  }
}

// SpeakerConfig::retrieve_speaker_config (:322-340): modules -> stored entry
static void retrieve_speaker_config(aasr_spkc *h, const std::string &speaker_id) {
  auto it = h->speakers.find(speaker_id);
  if (it == h->speakers.end())
    raise(AASR_ERR_INVALID, "SpeakerConfig: Unknown speaker %s", speaker_id.c_str());
  for (auto &kv : it->second) {
    std::vector<std::string> parts = split_fields(kv.first, 2);
    if (parts[0] == "feature") feat_get_parameters(h->feat, parts[1], kv.second);
    if (parts[0] == "model") cmllr_get_parameters(h, kv.second);
  }
}

// SpeakerConfig::retrieve_utterance_config (:343-362) -- it SETS the parameters
static void retrieve_utterance_config(aasr_spkc *h, const std::string &utterance_id) {
  auto it = h->utterances.find(utterance_id);
  if (it == h->utterances.end())
    raise(AASR_ERR_INVALID, "SpeakerConfig: Unknown utterance %s", utterance_id.c_str());
  set_modules(h, it->second);
}

// SpeakerConfig::set_utterance (:288-318)
void spkc_set_utterance(aasr_spkc *h, const std::string &utterance_id) {
  if (!h->cur_utterance.empty()) retrieve_utterance_config(h, h->cur_utterance);
  if (utterance_id.empty()) {
    if (!h->default_utterance_set)
      raise(AASR_ERR_INVALID, "SpeakerConfig: Default utterance is required.");
    set_modules(h, h->default_utterance);
  } else {
    auto it = h->utterances.find(utterance_id);
    if (it == h->utterances.end()) {
      if (!h->default_utterance_set)
        raise(AASR_ERR_INVALID,
              "SpeakerConfig: Unknown utterance %s, and default utterance settings are missing.",
              utterance_id.c_str());
      it = h->utterances.insert(SpeakerMap::value_type(utterance_id, h->default_utterance)).first;
    }
    set_modules(h, it->second);
  }
  h->cur_utterance = utterance_id;
}

// SpeakerConfig::set_speaker (:239-286)
void spkc_set_speaker(aasr_spkc *h, const std::string &speaker_id) {
  if (!h->cur_speaker.empty()) retrieve_speaker_config(h, h->cur_speaker);
  if (!h->cur_utterance.empty()) spkc_set_utterance(h, "");
  if (speaker_id != h->cur_speaker) reset_transforms(h);
  const bool load_new = h->trans_is_reset;
  if (speaker_id.empty()) {
    if (!h->default_speaker_set)
      raise(AASR_ERR_INVALID, "SpeakerConfig: No speaker defined, needs a default speaker.");
    set_modules(h, h->default_speaker);
  } else {
    auto it = h->speakers.find(speaker_id);
    if (it == h->speakers.end()) {
      if (!h->default_speaker_set)
        raise(AASR_ERR_INVALID,
              "SpeakerConfig: Unknown speaker %s, and default speaker settings are missing.",
              speaker_id.c_str());
      it = h->speakers.insert(SpeakerMap::value_type(speaker_id, h->default_speaker)).first;
    }
    set_modules(h, it->second);
  }
  h->cur_speaker = speaker_id;
  if (load_new) load_transforms(h);
}

void spkc_set_before_change(aasr_spkc *h, std::function<void()> fn) { h->before_change = std::move(fn); }

}  // namespace aasr

using namespace aasr;

extern "C" {

aasr_status aasr_spkc_create(aasr_feat *feat, aasr_gmm *gmm, aasr_spkc **out) {
  return guarded([&] {
    if (!feat || !out) raise(AASR_ERR_INVALID, "aasr_spkc_create: null argument");
    aasr_spkc *h = new aasr_spkc();
    h->feat = feat;
    h->gmm = gmm;
    *out = h;
  });
}

void aasr_spkc_destroy(aasr_spkc *h) { delete h; }

aasr_status aasr_spkc_set_model(aasr_spkc *h, aasr_gmm *gmm) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    h->gmm = gmm;
  });
}

aasr_status aasr_spkc_read_text(aasr_spkc *h, const char *text) {
  return guarded([&] {
    if (!h || !text) raise(AASR_ERR_INVALID, "aasr_spkc_read_text: null argument");
    spkc_read_text(h, text);
  });
}

aasr_status aasr_spkc_read_file(aasr_spkc *h, const char *path) {
  return guarded([&] {
    if (!h || !path) raise(AASR_ERR_INVALID, "aasr_spkc_read_file: null argument");
    std::ifstream in(path);
    if (!in) raise(AASR_ERR_IO, "could not open %s", path);
    std::stringstream ss;
    ss << in.rdbuf();
    spkc_read_text(h, ss.str());
  });
}

aasr_status aasr_spkc_set_speaker(aasr_spkc *h, const char *speaker_id) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    spkc_set_speaker(h, speaker_id ? speaker_id : "");
  });
}

aasr_status aasr_spkc_set_utterance(aasr_spkc *h, const char *utterance_id) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    spkc_set_utterance(h, utterance_id ? utterance_id : "");
  });
}

int64_t aasr_spkc_num_changes(const aasr_spkc *h) { return h ? h->changes : 0; }

aasr_status aasr_spkc_write_text(aasr_spkc *h, const char *const *speakers, int32_t n_speakers,
                                 const char *const *utterances, int32_t n_utterances, char **text_out,
                                 int64_t *text_len) {
  return guarded([&] {
    if (!h || !text_out || !text_len) raise(AASR_ERR_INVALID, "aasr_spkc_write_text: null argument");
    // SpeakerConfig::write_speaker_file (aku/SpeakerConfig.cc:156-236): the current speaker's and
    // utterance's module parameters are fetched first, then default speaker, speakers, default
    // utterance, utterances -- each filtered by the given set (n < 0: no filter)
    if (!h->cur_speaker.empty()) retrieve_speaker_config(h, h->cur_speaker);
    if (!h->cur_utterance.empty()) retrieve_utterance_config(h, h->cur_utterance);
    auto wanted = [](const char *const *set, int32_t n, const std::string &id) {
      if (n < 0) return true;
      for (int32_t i = 0; i < n; i++)
        if (set && set[i] && id == set[i]) return true;
      return false;
    };
    std::string t;
    auto block = [&](const char *kind, const std::string &id, const aasr::ModuleMap &mods) {
      t += std::string(kind) + " " + id + "\n{\n";
      for (const auto &kv : mods) {
        t += "  " + kv.first + "\n  {\n";  // ModuleConfig::write(file, 2) (aku/ModuleConfig.cc:204-222)
        for (size_t i = 0; i < kv.second.names.size(); i++) t += "    " + kv.second.names[i] + " " + kv.second.values[i] + "\n";
        t += "  }\n\n";
      }
      t += "}\n\n";
    };
    if (h->default_speaker_set && wanted(speakers, n_speakers, "default")) block("speaker", "default", h->default_speaker);
    for (const auto &sp : h->speakers)
      if (wanted(speakers, n_speakers, sp.first)) block("speaker", sp.first, sp.second);
    if (h->default_utterance_set && wanted(utterances, n_utterances, "default"))
      block("utterance", "default", h->default_utterance);
    for (const auto &ut : h->utterances)
      if (wanted(utterances, n_utterances, ut.first)) block("utterance", ut.first, ut.second);
    char *out = (char *)malloc(t.size() + 1);
    if (!out) raise(AASR_ERR_INVALID, "aasr_spkc_write_text: out of memory");
    memcpy(out, t.c_str(), t.size() + 1);
    *text_out = out;
    *text_len = (int64_t)t.size();
  });
}

}  // extern "C"
