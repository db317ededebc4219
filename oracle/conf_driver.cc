// conf_driver.cc -- drives an aku::conf::Config from a table.  TEST
// INFRASTRUCTURE ONLY.  The same file is compiled twice: against the
// reference's own aku/conf.cc (-> oracle/_ref/conf_ref, oracle/Makefile) and
// against the engine's aaltoasr_amd/csrc/aku/conf.hh (-> oracle/conf_engine);
// tests/test_conf_cli.py runs both on the same command lines and compares
// stdout, stderr and exit status.
//
//   conf_driver SPEC WORD...
// SPEC: first line = usage text; then one option per line,
//   short TAB long TAB type TAB default TAB help TAB getter(i|f|d|s)
// ("0" as short = no short name).  WORD... is the command line to parse.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "conf.hh"

int main(int argc, char *argv[]) {
  if (argc < 2) return 64;
  std::ifstream in(argv[1]);
  std::string line;
  std::getline(in, line);
  aku::conf::Config config;
  config(line + "\n");
  std::vector<std::string> keys, getters;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::vector<std::string> f;
    size_t at = 0;
    while (true) {
      const size_t t = line.find('\t', at);
      f.push_back(line.substr(at, t == std::string::npos ? std::string::npos : t - at));
      if (t == std::string::npos) break;
      at = t + 1;
    }
    if (f.size() != 6) return 65;
    config(f[0] == "0" ? (unsigned char)0 : (unsigned char)f[0][0], f[1], f[2], f[3], f[4]);
    keys.push_back(f[1].substr(0, f[1].find('=')));
    getters.push_back(f[5]);
  }
  config.default_parse(argc - 1, argv + 1);
  for (size_t i = 0; i < keys.size(); i++) {
    if (keys[i].empty()) continue;
    const aku::conf::Option &o = config[keys[i]];
    printf("%s|%s|%d|%s|", keys[i].c_str(), o.name.c_str(), o.specified ? 1 : 0, o.get_str().c_str());
    // the reference's tools only call typed getters on options they know hold a value
    if (o.specified || !o.get_str().empty()) {
      if (getters[i] == "i") printf("%d", o.get_int());
      else if (getters[i] == "f") printf("%.9g", (double)o.get_float());
      else if (getters[i] == "d") printf("%.17g", o.get_double());
    }
    printf("\n");
  }
  for (size_t i = 0; i < config.arguments.size(); i++) printf("arg|%s\n", config.arguments[i].c_str());
  return 0;
}
