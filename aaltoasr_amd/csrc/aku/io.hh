// io.hh -- io::Stream with the interface of aku/io.hh: a FILE* that opens plain files, "-"
// (stdin / stdout), gzip files (through a gzip pipe) and process pipes ("cmd|" for reading,
// "|cmd" for writing), converts to FILE* implicitly and closes itself.  Tools written against
// aku pass it straight to FeatureGenerator::load_configuration / open, Recipe::read and
// SpeakerConfig::read_speaker_file (aku/phone_probs.cc:84-136, aku/feacat.cc:72-87).
#ifndef AKU_AMD_IO_HH
#define AKU_AMD_IO_HH

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace io {

struct Stream {
  Stream() : file(NULL), is_pipe(false), close_allowed(true) {}
  Stream(std::string file_name, std::string mode = "r", bool allow_close = true)
      : file(NULL), is_pipe(false), close_allowed(true) {
    open(file_name, mode, allow_close);
  }
  ~Stream() { close(); }

  void open(std::string file_name, std::string mode = "r", bool allow_close = true) {
    close();
    close_allowed = allow_close;
    if (file_name.empty()) {
      fprintf(stderr, "io::Stream::open(): empty file name\n");
      exit(1);
    }
    is_pipe = false;
    const bool rd = mode.find('r') != std::string::npos;
    if (file_name == "-") {
      file = rd ? stdin : stdout;
      close_allowed = false;
    } else if (file_name[file_name.size() - 1] == '|') {
      if (!rd) {
        fprintf(stderr, "io::Stream::open(): pipe %s must be read\n", file_name.c_str());
        exit(1);
      }
      file = popen(file_name.substr(0, file_name.size() - 1).c_str(), "r");
      is_pipe = true;
    } else if (file_name[0] == '|') {
      file = popen(file_name.substr(1).c_str(), "w");
      is_pipe = true;
    } else if (file_name.size() > 3 && file_name.compare(file_name.size() - 3, 3, ".gz") == 0) {
      const std::string cmd = rd ? "gzip -dc '" + file_name + "'" : "gzip > '" + file_name + "'";
      file = popen(cmd.c_str(), rd ? "r" : "w");
      is_pipe = true;
    } else {
      file = fopen(file_name.c_str(), mode.c_str());
    }
    if (file == NULL) {
      fprintf(stderr, "io::Stream::open(): could not open %s: %s\n", file_name.c_str(), strerror(errno));
      exit(1);
    }
  }

  void close() {
    if (file == NULL || !close_allowed) {
      file = NULL;
      return;
    }
    if (is_pipe) pclose(file);
    else fclose(file);
    file = NULL;
    is_pipe = false;
  }

  operator FILE *() { return file; }

  FILE *file;
  bool is_pipe;
  bool close_allowed;

private:
  Stream(const Stream &);
  Stream &operator=(const Stream &);
};

}  // namespace io

#endif
