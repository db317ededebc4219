// PhoneProbsToolbox.cc -- see PhoneProbsToolbox.hh.
#include "PhoneProbsToolbox.hh"

#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../pipeline.h"

namespace aku {

void PPToolbox::read_configuration(const std::string &cfgname) {
  FILE *fp = fopen(cfgname.c_str(), "r");
  if (!fp) throw std::string("could not open ") + cfgname;
  try {
    m_gen.load_configuration(fp);
  } catch (...) {
    fclose(fp);
    throw;
  }
  fclose(fp);
}

void PPToolbox::read_models(const std::string &base) { m_model.read_all(base); }

// aku/PhoneProbsToolbox.cc:50-53
void PPToolbox::set_clustering(const std::string &clfile_name, double eval_minc, double eval_ming) {
  m_model.read_clustering(clfile_name);
  m_model.set_clustering_min_evals(eval_minc, eval_ming);
}

static void write_all(int fd, const uint8_t *p, size_t n) {
  while (n > 0) {
    ssize_t w = write(fd, p, n);
    if (w <= 0) throw std::string("Write error");
    p += w;
    n -= (size_t)w;
  }
}

static void run(FeatureGenerator &gen, HmmSet &model, const std::vector<int16_t> &pcm, int out_fd) {
  if (model.dim() != gen.dim()) {
    char buf[256];
    snprintf(buf, sizeof buf, "Gaussian dimension is %d but feature dimension is %d.", model.dim(),
             gen.dim());
    throw std::string(buf);
  }
  uint8_t *lna = nullptr;
  int64_t len = 0, frames = 0;
  if (aasr_run_utterance(gen.handle(), model.handle(), pcm.data(), (int64_t)pcm.size(), 0, 0, 1, 2,
                         &lna, &len, &frames) != AASR_OK)
    throw std::string(aasr_last_error());
  try {
    write_all(out_fd, lna, (size_t)len);
  } catch (...) {
    aasr_free(lna);
    throw;
  }
  aasr_free(lna);
}

void PPToolbox::generate_from_file_to_fd(const std::string &input, int out_fd, bool) {
  // the reference never looks at raw_flag (aku/PhoneProbsToolbox.cc:135-151 calls gen.open(name));
  // headerless input is the audiofile module's `raw` option or the open fallback
  std::vector<int16_t> pcm;
  try {
    pcm = aasr::read_input_file(m_gen.handle(), input, false);
  } catch (aasr::Error &e) {
    throw std::string(e.msg);
  }
  run(m_gen, m_model, pcm, out_fd);
}

void PPToolbox::generate_to_fd(int in_fd, int out_fd, bool) {
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  ssize_t n;
  while ((n = read(in_fd, buf, sizeof buf)) > 0) data.insert(data.end(), buf, buf + n);
  // FeatureGenerator::open_fd -> open(FILE*) (aku/FeatureGenerator.cc:55-66): raw_audio is ignored there too
  std::vector<int16_t> pcm;
  try {
    pcm = aasr::decode_input_data(m_gen.handle(), std::vector<char>(data.begin(), data.end()), "(fd)");
  } catch (aasr::Error &e) {
    throw std::string(e.msg);
  }
  run(m_gen, m_model, pcm, out_fd);
}

void PPToolbox::generate(const std::string &input, const std::string &output, bool raw) {
  FILE *fp = fopen(output.c_str(), "wb");
  if (!fp) throw std::string("could not open ") + output;
  try {
    fflush(fp);
    generate_from_file_to_fd(input, fileno(fp), raw);
  } catch (...) {
    fclose(fp);
    throw;
  }
  fclose(fp);
}

}  // namespace aku
