// conf.hh -- command-line options with the grammar of the reference's
// aku::conf::Config (aku/conf.hh:12-40, aku/conf.cc:120-330), written for the
// engine's tools so that a phone_probs / feacat command line behaves the same:
//   * grouped short options ("-aN", "-hi 10"); arguments of options are taken
//     from the FOLLOWING words, first pending option first, whatever those
//     words look like ("-i -3" gives info = -3; "-i10" is three options);
//   * "--name value" and "--name=value"; "--" ends option processing; a lone
//     "-" is an ordinary argument;
//   * no abbreviation of long names, unknown names are fatal;
//   * messages, exit codes and the help layout of the reference.
// Config::read (option files) is not provided; no tool on the path uses it.
#ifndef AASR_AKU_CONF_HH
#define AASR_AKU_CONF_HH

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <string>
#include <vector>

namespace aku {
namespace conf {

struct Option {
  unsigned char short_name = 0;
  std::string long_name;  // as declared, e.g. "config=FILE"
  std::string value;
  bool required = false, needs_argument = false, specified = false;
  std::string help;
  std::string name;  // "-c --config"

  // strtol / strtod over the whole value or exit(1) (aku/conf.cc:12-50)
  int get_int() const {
    char *end = nullptr;
    const long v = strtol(value.c_str(), &end, 10);
    if (end == value.c_str() || *end) bad_value();
    return (int)v;
  }
  double get_double() const {
    char *end = nullptr;
    const double v = strtod(value.c_str(), &end);
    if (end == value.c_str() || *end) bad_value();
    return v;
  }
  float get_float() const { return (float)get_double(); }
  const std::string &get_str() const { return value; }
  const char *get_c_str() const { return value.c_str(); }

  // " -c --config=FILE", the form used in diagnostics
  std::string label() const {
    std::string s;
    if (short_name) s += std::string(" -") + (char)short_name;
    if (!long_name.empty()) s += " --" + long_name;
    return s;
  }

 private:
  [[noreturn]] void bad_value() const {
    fprintf(stderr, "invalid value for option %s: %s\n", name.c_str(), value.c_str());
    exit(1);
  }
};

class Config {
 public:
  std::string usage_line;
  std::vector<Option> options;
  std::vector<std::string> arguments;  // the words that are not options

  Config &operator()(const std::string &usage) {
    usage_line = usage;
    return *this;
  }

  // type: blank-separated "arg" (takes an argument) and/or "must" (required)
  Config &operator()(unsigned char short_name, const std::string &long_name, const std::string &type = "",
                     const std::string &default_value = "", const std::string &help = "") {
    Option o;
    o.short_name = short_name;
    o.long_name = long_name;
    o.value = default_value;
    o.help = help;
    size_t at = 0;
    while (at < type.size()) {
      const size_t b = type.find_first_not_of(" \t", at);
      if (b == std::string::npos) break;
      const size_t e = type.find_first_of(" \t", b);
      const std::string word = type.substr(b, e == std::string::npos ? std::string::npos : e - b);
      if (word == "arg") o.needs_argument = true;
      else if (word == "must") o.required = true;
      else {
        fprintf(stderr, "invalid option type %s for option -%c --%s\n", word.c_str(), short_name, long_name.c_str());
        abort();
      }
      at = e == std::string::npos ? type.size() : e;
    }
    const std::string key = long_name.substr(0, long_name.find('='));
    if (short_name) {
      if (by_short_.count(short_name)) {
        fprintf(stderr, "trying to add option -%c twice\n", short_name);
        exit(1);
      }
      by_short_[short_name] = options.size();
      o.name = std::string("-") + (char)short_name;
    }
    if (!key.empty()) {
      if (by_long_.count(key)) {
        fprintf(stderr, "trying to add option --%s twice\n", key.c_str());
        exit(1);
      }
      by_long_[key] = options.size();
      if (long_name.size() > widest_) widest_ = long_name.size();
      o.name += (o.name.empty() ? "--" : " --") + key;
    }
    options.push_back(o);
    return *this;
  }

  void parse(int argc, char *argv[], bool override = true) {
    std::deque<std::string> words;
    for (int i = 1; i < argc; i++) words.push_back(argv[i]);
    bool options_open = true;   // no "--" seen yet
    std::deque<size_t> waiting;  // options whose argument has not arrived yet
    auto note = [&](size_t idx) {
      if (options[idx].needs_argument) waiting.push_back(idx);
      else options[idx].specified = true;
    };
    while (!words.empty()) {
      const std::string w = words.front();
      words.pop_front();
      const bool optionlike = options_open && waiting.empty() && w.size() > 1 && w[0] == '-';
      if (!optionlike) {
        if (waiting.empty()) {
          arguments.push_back(w);
          continue;
        }
        Option &o = options[waiting.front()];
        waiting.pop_front();
        if (override || !o.specified) o.value = w;
        o.specified = true;
        continue;
      }
      if (w == "--") {
        options_open = false;
        continue;
      }
      if (w[1] != '-') {
        for (size_t i = 1; i < w.size(); i++) {
          const auto it = by_short_.find((unsigned char)w[i]);
          if (it == by_short_.end()) {
            fprintf(stderr, "invalid option -%c\n", (unsigned char)w[i]);
            exit(1);
          }
          note(it->second);
        }
        continue;
      }
      std::string key = w.substr(2);
      const size_t eq = key.find('=');
      if (eq != std::string::npos) {
        words.push_front(key.substr(eq + 1));  // becomes the next word, whatever the option is
        key.erase(eq);
      }
      const auto it = by_long_.find(key);
      if (it == by_long_.end()) {
        fprintf(stderr, "invalid option --%s\n", key.c_str());
        exit(1);
      }
      note(it->second);
    }
    if (!waiting.empty()) {
      fprintf(stderr, "option%s lacks an argument\n", options[waiting.front()].label().c_str());
      exit(1);
    }
  }

  // parse + "help" + required options (aku/conf.cc:273-284)
  void default_parse(int argc, char *argv[]) {
    parse(argc, argv);
    if ((*this)["help"].specified) {
      fputs(help_string().c_str(), stdout);
      exit(0);
    }
    check_required();
  }

  void check_required() const {
    for (const Option &o : options)
      if (o.required && !o.specified) {
        fprintf(stderr, "option%s required\n", o.label().c_str());
        print_help(stderr, 1);
      }
  }

  [[noreturn]] void print_help(FILE *file = stdout, int exit_value = 0) const {
    fputs(help_string().c_str(), file);
    exit(exit_value);
  }

  std::string help_string() const {
    std::string h = usage_line;
    for (const Option &o : options) {
      h += "  ";
      h += o.short_name ? std::string("-") + (char)o.short_name : std::string("  ");
      if (!o.long_name.empty()) {
        h += o.short_name ? ", " : "  ";
        h += "--" + o.long_name + std::string(widest_ - o.long_name.size(), ' ');
      } else if (widest_ > 0) {
        h += std::string(widest_ + 4, ' ');
      }
      h += "  " + o.help + "\n";
    }
    return h;
  }

  const Option &operator[](unsigned char short_name) const {
    const auto it = by_short_.find(short_name);
    if (it == by_short_.end()) {
      fprintf(stderr, "Config::get(): unknown option %c\n", short_name);
      abort();
    }
    return options[it->second];
  }
  const Option &operator[](const std::string &long_name) const {
    const auto it = by_long_.find(long_name);
    if (it == by_long_.end()) {
      fprintf(stderr, "Config::get(): unknown option %s\n", long_name.c_str());
      abort();
    }
    return options[it->second];
  }
  const Option &operator[](const char *long_name) const { return (*this)[std::string(long_name)]; }

 private:
  std::map<unsigned char, size_t> by_short_;
  std::map<std::string, size_t> by_long_;
  size_t widest_ = 0;
};

}  // namespace conf
}  // namespace aku

#endif
