// pipeline.cc -- recipe handling and the whole-path driver
// (audio -> features -> state likelihoods -> LNA files) behind
// aasr_run_recipe / aasr_run_utterance.
//
// Replaces the body of phone_probs' main loop (aku/phone_probs.cc:145-267) and
// PPToolbox::generate_from_file_to_fd (aku/PhoneProbsToolbox.cc:135-208).  The
// reference walks one frame at a time; here the utterances of a recipe slice
// are packed into blocks of frames, each block runs the feature graph, the
// scoring kernel and the LNA packer back to back on the device, and the packed
// bytes come back in one copy per block.
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <sched.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "feat.h"
#include "gmm.h"
#include "pipeline.h"

namespace aasr {

void lna_encode_f64_launch(const double *d_lik, int64_t F, int S, int normalize, int lnabytes, float *d_lp,
                           uint8_t *d_bytes, hipStream_t stream);

// ------------------------------------------------------------------ recipe --

// Recipe::read's batch arithmetic (aku/Recipe.cc:63-112) for
// cluster_speakers=false: walks the lines exactly like the reference loop and
// reports which lines land in batch_index.
void recipe_batch_range(int total, int num_batches, int batch_index, int *first, int *count) {
  if (num_batches > 1 && (batch_index < 1 || batch_index > num_batches))
    raise(AASR_ERR_INVALID, "Invalid batch index");
  int target_lines, batch_remainder = 0;
  if (num_batches <= 1) {
    target_lines = total;
  } else {
    target_lines = total / num_batches;
    batch_remainder = total % num_batches;
  }
  int extra_line = 1;
  if (target_lines < 1) {
    target_lines = 1;
    extra_line = 0;
  }
  if (batch_remainder == 0) extra_line = 0;
  int cur_index = 1, cur_line = 0;
  *first = -1;
  *count = 0;
  for (int i = 0; i < total; i++) {
    if (num_batches > 1 && cur_index < num_batches) {
      if (cur_line >= target_lines + extra_line) {
        cur_index++;
        if (cur_index > batch_index) break;
        cur_line -= target_lines + extra_line;
        if (cur_index > batch_remainder) extra_line = 0;
      }
    }
    if (num_batches <= 1 || cur_index == batch_index) {
      if (*first < 0) *first = i;
      (*count)++;
    }
    cur_line++;
  }
  if (*first < 0) *first = total;
}

// str::clean (aku/str.cc:124-140): drop leading and trailing characters of `chars`
std::string str_clean(const std::string &s, const char *chars) {
  const size_t a = s.find_first_not_of(chars);
  if (a == std::string::npos) return "";
  const size_t b = s.find_last_not_of(chars);
  return s.substr(a, b - a + 1);
}

// str::split (aku/str.cc:142-172).  Kept as the reference has it: a field ends at the next
// delimiter, ONE delimiter is eaten (all adjacent ones with `group`), and the loop stops when the
// text is used up -- so a trailing delimiter does not open an empty last field ("a=" is one field,
// "a=b=" two, "=b" two with an empty first).  With num_fields > 0 the last field takes the rest.
std::vector<std::string> str_split(const std::string &s, const char *delims, bool group, int num_fields) {
  std::vector<std::string> fields;
  size_t begin = 0;
  while (begin < s.size()) {
    if (num_fields > 0 && (int)fields.size() == num_fields - 1) {
      fields.push_back(s.substr(begin));
      break;
    }
    size_t end = s.find_first_of(delims, begin);
    if (end == std::string::npos) end = s.size();
    fields.push_back(s.substr(begin, end - begin));
    end++;
    if (group)
      while (end < s.size() && strchr(delims, s[end]) && s[end]) end++;
    begin = end;
  }
  return fields;
}

// Recipe::read (aku/Recipe.cc:23-149).  Quirk kept: the key=value map is not
// cleared between lines, so a key missing on a line inherits the value of the
// previous line that set it -- including lines that belong to other batches.
// The walk is the reference's: a line is parsed, then the batch counter may advance (with
// cluster_speakers only where the speaker changes), then the line is kept if the counter
// equals batch_index; the loop stops at the first line of a later batch, so that line is still
// checked for syntax (aku/Recipe.cc:79-103).
std::vector<RecipeInfo> recipe_read(const std::string &text, int num_batches, int batch_index,
                                    bool cluster_speakers) {
  if (num_batches > 1 && (batch_index < 1 || batch_index > num_batches))
    raise(AASR_ERR_INVALID, "Invalid batch index");
  std::vector<std::string> lines;
  {
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
      line = str_clean(line, "\n\t ");  // not '\r': the reference keeps it (aku/Recipe.cc:57)
      if (line.empty() || line[0] == '#') continue;
      lines.push_back(line);
    }
  }
  int target_lines, batch_remainder = 0;
  if (num_batches <= 1) {
    target_lines = (int)lines.size();
  } else {
    target_lines = (int)lines.size() / num_batches;
    batch_remainder = (int)lines.size() % num_batches;
  }
  int extra_line = 1;
  if (target_lines < 1) {
    target_lines = 1;
    extra_line = 0;
  }
  if (batch_remainder == 0) extra_line = 0;
  std::vector<RecipeInfo> infos;
  std::map<std::string, std::string> kv;
  int cur_index = 1, cur_line = 0;
  std::string cur_speaker;
  for (const std::string &line : lines) {
    for (const std::string &field : str_split(line, " \t", true)) {
      const std::vector<std::string> key_value = str_split(field, "=", false);
      if (key_value.size() != 2) raise(AASR_ERR_INVALID, "Invalid recipe line: %s", line.c_str());
      kv[key_value[0]] = key_value[1];
    }
    if (num_batches > 1 && cur_index < num_batches) {
      auto sp = kv.find("speaker");
      const std::string new_speaker = sp == kv.end() ? "" : sp->second;
      if (cur_line >= target_lines + extra_line &&
          (!cluster_speakers || cur_speaker.empty() || cur_speaker != new_speaker)) {
        cur_index++;
        if (cur_index > batch_index) break;
        cur_line -= target_lines + extra_line;
        if (cur_index > batch_remainder) extra_line = 0;
      }
      cur_speaker = new_speaker;
    }
    if (num_batches <= 1 || cur_index == batch_index) {
      RecipeInfo info;
      auto get = [&](const char *k, std::string &dst) {
        auto it = kv.find(k);
        if (it != kv.end()) dst = it->second;
      };
      get("audio", info.audio_path);
      get("alt-audio", info.alt_audio_path);
      get("transcript", info.transcript_path);
      get("alignment", info.alignment_path);
      get("hmmnet", info.hmmnet_path);
      get("den-hmmnet", info.den_hmmnet_path);
      get("lna", info.lna_path);
      get("speaker", info.speaker_id);
      get("utterance", info.utterance_id);
      auto it = kv.find("start-time");
      if (it != kv.end()) info.start_time = (float)atof(it->second.c_str());
      it = kv.find("end-time");
      if (it != kv.end()) info.end_time = (float)atof(it->second.c_str());
      it = kv.find("start-line");
      if (it != kv.end()) info.start_line = atoi(it->second.c_str());
      it = kv.find("end-line");
      if (it != kv.end()) info.end_line = atoi(it->second.c_str());
      infos.push_back(info);
    }
    cur_line++;
  }
  return infos;
}

// ------------------------------------------------------------------- audio --

// read_audio_file / decode_audio: audio_reader.cc

// PreModule::set_file (aku/FeatureModules.cc:603-631): a feature file is the
// dimension (native int32, or one byte for legacy files) followed by float32
// frames.  Returned in int16 units (two per float), the layout every feature
// entry point shares with audio.
std::vector<int16_t> parse_feature_data(const std::vector<char> &data, int dim, bool legacy) {
  size_t off;
  int file_dim;
  if (legacy) {
    if (data.size() < 1) raise(AASR_ERR_IO, "PreModule: Could not read the file.");
    file_dim = (signed char)data[0];
    off = 1;
  } else {
    if (data.size() < 4) raise(AASR_ERR_IO, "PreModule: Could not read the file.");
    int32_t d;
    memcpy(&d, data.data(), 4);
    file_dim = d;
    off = 4;
  }
  if (file_dim != dim) raise(AASR_ERR_INVALID, "PreModule: The file has invalid dimension");
  const size_t frames = (data.size() - off) / ((size_t)dim * 4);
  std::vector<int16_t> out(frames * dim * 2);
  if (!out.empty()) memcpy(out.data(), data.data() + off, out.size() * 2);
  return out;
}

std::vector<int16_t> read_feature_file(const std::string &path, int dim, bool legacy) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) raise(AASR_ERR_IO, "could not open file %s", path.c_str());
  std::vector<char> data;
  char buf[65536];
  size_t got;
  if (fseek(fp, 0, SEEK_END) == 0) {
    const long n = ftell(fp);
    rewind(fp);
    if (n > 0) {
      data.resize((size_t)n);
      data.resize(fread(data.data(), 1, (size_t)n, fp));
    }
  }
  // not seekable, or a stream whose size reads as 0 (procfs, character devices): chunked read
  if (data.empty()) {
    clearerr(fp);
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + got);
  }
  fclose(fp);
  return parse_feature_data(data, dim, legacy);
}

// FeatureGenerator::open for whichever base module the graph has
std::vector<int16_t> read_input_file(const aasr_feat *feat, const std::string &path, bool force_raw) {
  const FeatModule &b = feat->mods[0];
  if (b.type == MOD_PRE) return read_feature_file(path, b.dim, b.legacy_file != 0);
  const bool raw = force_raw || b.raw_audio != 0;
  return read_audio_file(path, raw, b.sample_rate, b.endian == 2);
}

std::vector<int16_t> decode_input_data(const aasr_feat *feat, const std::vector<char> &data,
                                       const std::string &name) {
  const FeatModule &b = feat->mods[0];
  if (b.type == MOD_PRE) return parse_feature_data(data, b.dim, b.legacy_file != 0);
  return decode_audio(data, name, b.raw_audio != 0, b.endian == 2, b.sample_rate);
}

// ------------------------------------------------------------------ driver --

// How many engine processes share this host's cores (one per GPU of the node).  The recipe driver's
// helper threads are sized from usable_cores / share: eight ranks that each start the thread count
// tuned for a rank that owns the host oversubscribe the CPU quota eight times over.
static std::atomic<int> g_host_share{0};
static std::atomic<int64_t> g_upload_ring_limit{0};

int host_usable_cores() {
  // affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 logical CPUs and
  // schedule 16 of them)
  int n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
  auto quota = [&](const char *path_max, const char *path_q, const char *path_p) {
    if (FILE *fp = fopen(path_max, "r")) {  // cgroup v2: "<quota|max> <period>"
      char q[64];
      long long period = 0;
      if (fscanf(fp, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
        n = std::min(n, (int)std::max<long long>(1, atoll(q) / period));
      fclose(fp);
      return;
    }
    long long q = -1, per = 0;
    if (FILE *fp = fopen(path_q, "r")) {
      if (fscanf(fp, "%lld", &q) != 1) q = -1;
      fclose(fp);
    }
    if (FILE *fp = fopen(path_p, "r")) {
      if (fscanf(fp, "%lld", &per) != 1) per = 0;
      fclose(fp);
    }
    if (q > 0 && per > 0) n = std::min(n, (int)std::max<long long>(1, q / per));
  };
  quota("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  return std::max(1, n);
}

int host_share() {
  int n = g_host_share.load();
  if (n <= 0)
    for (const char *name : {"AASR_LOCAL_RANKS", "LOCAL_WORLD_SIZE"})
      if (const char *e = getenv(name))
        if (atoi(e) > 0) {
          n = atoi(e);
          break;
        }
  return std::max(1, n);
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Job {
  size_t info_index;
  std::vector<int16_t> pcm;
  // recipe driver: the samples in the pinned upload ring instead (pcm empty): [pin, pin + pin_n), ring space up to the
  // virtual offset pin_end is the job's and everything before it
  const int16_t *pin = nullptr;
  size_t pin_n = 0;
  uint64_t pin_end = 0;
  int32_t start, count;  // frames start .. start+count-1
  std::string out_file;
  const int16_t *samples() const { return pin ? pin : pcm.data(); }
  size_t n_samples() const { return pin ? pin_n : pcm.size(); }
};

// Rehearsal of N ranks on a box with ONE GPU (tools/rehearse_ranks.py): with AASR_RECIPE_STUB=1 an ABLATION build
// (AASR_BUILD_ABLATION=1, aaltoasr_amd/lib_ablation) skips the block's kernels and its device -> host copy, so
// that what is timed is the host side of a rank alone -- file reads, uploads, the writer pool -- without N ranks
// queueing on one device and one PCIe link.  The files written hold whatever the pinned slots held: the product
// library has no such switch.
static bool stub_device() {
#if defined(AASR_ABLATION) && AASR_ABLATION
  static const bool on = AASR_EXPERIMENT_ENV("AASR_RECIPE_STUB") && atoi(AASR_EXPERIMENT_ENV("AASR_RECIPE_STUB")) == 1;
  return on;
#else
  return false;
#endif
}

// Result slots of the recipe driver: a block's packed rows live in a device buffer and a pinned host buffer of its slot
// from the launch until its last file is closed.  More than two: block k's kernels are enqueued while block k-1 crosses PCIe and
// block k-2 is being written -- with two, the launch of block k waited for the writers of block k-2 (a copy takes 4.5 ms,
// writing its files ~5), its kernels started when they were done, and the link idled for a block's device time in
// every cycle (wall 1.21-1.29 s against 0.98 s of copies on the 10 000-utterance recipe; three slots 1.04-1.17 s with
// 0.04-0.11 s of waits for a slot left, hence four).
constexpr int kSlots = 4;

struct BlockRunner {
  aasr_feat *feat = nullptr;
  aasr_gmm *gmm = nullptr;
  int lnabytes = 2, normalize = 1;
  BlockRunner() = default;
  BlockRunner(aasr_feat *f, aasr_gmm *g, int lb, int nz) : feat(f), gmm(g), lnabytes(lb), normalize(nz) {}
  DevBuf<int16_t> d_pcm;
  DevBuf<float> d_fea, d_ll;
  DevBuf<double> d_fea64, d_lik64;  // AASR_PREC_F64: double features, linear state likelihoods
  DevBuf<uint8_t> d_bytes, d_slot_bytes[kSlots];
  std::vector<uint8_t> h_bytes;
  double device_seconds = 0;
  // recipe driver: kernels on one stream, the copy of the packed rows to the host on another, so
  // block n+1 is computed while block n crosses PCIe (6.25 kB per frame: the copy is the longer leg)
  hipStream_t s_compute = nullptr, s_copy = nullptr;
  hipEvent_t ev_start[kSlots] = {}, ev_kernels[kSlots] = {}, ev_copy0[kSlots] = {}, ev_copied[kSlots] = {};
  double copy_seconds = 0;
  ~BlockRunner() {
    for (int i = 0; i < kSlots; i++) {
      if (ev_start[i]) (void)hipEventDestroy(ev_start[i]);
      if (ev_kernels[i]) (void)hipEventDestroy(ev_kernels[i]);
      if (ev_copy0[i]) (void)hipEventDestroy(ev_copy0[i]);
      if (ev_copied[i]) (void)hipEventDestroy(ev_copied[i]);
    }
    if (s_compute) (void)hipStreamDestroy(s_compute);
    if (s_copy) (void)hipStreamDestroy(s_copy);
  }
  void init_streams() {
    if (s_compute) return;
    AASR_HIP(hipStreamCreateWithFlags(&s_compute, hipStreamNonBlocking));
    AASR_HIP(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    for (int i = 0; i < kSlots; i++) {
      AASR_HIP(hipEventCreate(&ev_start[i]));
      AASR_HIP(hipEventCreate(&ev_kernels[i]));
      AASR_HIP(hipEventCreate(&ev_copy0[i]));
      AASR_HIP(hipEventCreate(&ev_copied[i]));
    }
  }

  // Asynchronous form: enqueue block `jobs` (kernels on s_compute writing the rows of slot `slot`,
  // their copy to `dst` on s_copy); finish(slot) waits for the copy and books the times.
  void launch(const std::vector<Job *> &jobs, int slot, uint8_t *dst, size_t dst_cap) {
    init_streams();
    UttBatch ub;
    ub.n_utts = (int32_t)jobs.size();
    ub.frame_off.assign(1, 0);
    ub.pcm_off.assign(1, 0);
    for (Job *j : jobs) {
      ub.first.push_back(j->start);
      ub.frame_off.push_back(ub.frame_off.back() + j->count);
      ub.pcm_off.push_back(ub.pcm_off.back() + (int64_t)j->n_samples());
    }
    const int64_t F = ub.frame_off.back();
    const int dim = feat->mods.back().dim;
    const int64_t S = gmm->S;
    const size_t nb = (size_t)F * S * lnabytes;
    if (nb > dst_cap) raise(AASR_ERR_INVALID, "internal: result buffer too small");
    DevBuf<uint8_t> &bytes = d_slot_bytes[slot];
    AASR_HIP(hipEventRecord(ev_start[slot], s_compute));
    if (F > 0) {
      // growing a buffer frees the old one: nothing may still be reading it
      if ((size_t)ub.pcm_off.back() > d_pcm.n || (size_t)F * dim > d_fea.n || nb > bytes.n)
        AASR_HIP(hipDeviceSynchronize());
      d_pcm.ensure((size_t)ub.pcm_off.back());
      // one copy per run of utterances whose samples lie back to back on the host (the reader's upload ring hands out
      // consecutive space, so a block is normally one or two copies instead of ~46 -- a copy costs ~10 us of stream time)
      for (size_t k = 0; k < jobs.size();) {
        size_t e = k + 1;
        while (e < jobs.size() && jobs[e]->samples() == jobs[e - 1]->samples() + jobs[e - 1]->n_samples()) e++;
        const size_t n = (size_t)(ub.pcm_off[e] - ub.pcm_off[k]);
        if (n) AASR_HIP(hipMemcpyAsync(d_pcm.p + ub.pcm_off[k], jobs[k]->samples(), n * sizeof(int16_t), hipMemcpyHostToDevice, s_compute));
        k = e;
      }
      d_fea.ensure((size_t)F * dim);
      const int64_t pitch = gmm_engine_pitch(gmm);
      if ((size_t)F * pitch > d_ll.n) AASR_HIP(hipDeviceSynchronize());
      d_ll.ensure((size_t)F * pitch);
      bytes.ensure(nb);
      if (gmm->precision == AASR_PREC_F64) {
        // the reference's arithmetic end to end: double features, double scoring, the LNA tail as written
        if ((size_t)F * dim > d_fea64.n || (size_t)F * S > d_lik64.n) AASR_HIP(hipDeviceSynchronize());
        d_fea64.ensure((size_t)F * dim);
        d_lik64.ensure((size_t)F * S);
        feat_run_batch(feat, d_pcm.p, ub, (int)feat->mods.size() - 1, nullptr, d_fea64.p, s_compute);
        gmm_score_f64_launch(gmm, d_fea64.p, F, d_lik64.p, 1, s_compute);
        lna_encode_f64_launch(d_lik64.p, F, (int)S, normalize, lnabytes, nullptr, bytes.p, s_compute);
      } else if (!stub_device()) {
        feat_run_batch(feat, d_pcm.p, ub, (int)feat->mods.size() - 1, d_fea.p, nullptr, s_compute);
        gmm_score_launch_engine(gmm, d_fea.p, F, d_ll.p, pitch, s_compute);
        lna_encode_launch(d_ll.p, F, (int)S, normalize, lnabytes, nullptr, bytes.p, s_compute, pitch, gmm_engine_colmap(gmm));
      }
    }
    AASR_HIP(hipEventRecord(ev_kernels[slot], s_compute));
    AASR_HIP(hipStreamWaitEvent(s_copy, ev_kernels[slot], 0));
    AASR_HIP(hipEventRecord(ev_copy0[slot], s_copy));   // (behind the previous block's copy: the copy's own time is booked)
    if (nb && !stub_device()) AASR_HIP(hipMemcpyAsync(dst, bytes.p, nb, hipMemcpyDeviceToHost, s_copy));
    AASR_HIP(hipEventRecord(ev_copied[slot], s_copy));
  }
  void finish(int slot) {
    AASR_HIP(hipEventSynchronize(ev_copied[slot]));
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev_start[slot], ev_kernels[slot]) == hipSuccess) device_seconds += ms * 1e-3;
    if (hipEventElapsedTime(&ms, ev_copy0[slot], ev_copied[slot]) == hipSuccess) copy_seconds += ms * 1e-3;
  }

  // features + scoring + LNA for a block of jobs; the packed rows [sum count][S*lnabytes]
  // go to `dst` (a pinned buffer of the recipe runner) or, when null, to h_bytes
  void run(const std::vector<Job *> &jobs, uint8_t *dst = nullptr, size_t dst_cap = 0) {
    UttBatch ub;
    ub.n_utts = (int32_t)jobs.size();
    ub.frame_off.assign(1, 0);
    ub.pcm_off.assign(1, 0);
    for (Job *j : jobs) {
      ub.first.push_back(j->start);
      ub.frame_off.push_back(ub.frame_off.back() + j->count);
      ub.pcm_off.push_back(ub.pcm_off.back() + (int64_t)j->n_samples());
    }
    const int64_t F = ub.frame_off.back();
    const int dim = feat->mods.back().dim;
    const int64_t S = gmm->S;
    if (F <= 0) {
      h_bytes.clear();
      return;
    }
    double t0 = now_s();
    d_pcm.ensure((size_t)ub.pcm_off.back());
    for (size_t k = 0; k < jobs.size(); k++)
      AASR_HIP(hipMemcpyAsync(d_pcm.p + ub.pcm_off[k], jobs[k]->samples(),
                              jobs[k]->n_samples() * sizeof(int16_t), hipMemcpyHostToDevice, nullptr));
    d_fea.ensure((size_t)F * dim);
    // the score matrix never leaves the device: rows padded to whole 64-byte lines where the
    // scoring kernel can write them that way (every output group then is one full line)
    const int64_t pitch = gmm_engine_pitch(gmm);
    d_ll.ensure((size_t)F * pitch);
    d_bytes.ensure((size_t)F * S * lnabytes);
    if (gmm->precision == AASR_PREC_F64) {
      d_fea64.ensure((size_t)F * dim);
      d_lik64.ensure((size_t)F * S);
      feat_run_batch(feat, d_pcm.p, ub, (int)feat->mods.size() - 1, nullptr, d_fea64.p, nullptr);
      gmm_score_f64_launch(gmm, d_fea64.p, F, d_lik64.p, 1, nullptr);
      lna_encode_f64_launch(d_lik64.p, F, (int)S, normalize, lnabytes, nullptr, d_bytes.p, nullptr);
    } else {
      feat_run_batch(feat, d_pcm.p, ub, (int)feat->mods.size() - 1, d_fea.p, nullptr, nullptr);
      gmm_score_launch_engine(gmm, d_fea.p, F, d_ll.p, pitch, nullptr);
      lna_encode_launch(d_ll.p, F, (int)S, normalize, lnabytes, nullptr, d_bytes.p, nullptr, pitch, gmm_engine_colmap(gmm));
    }
    const size_t nb = (size_t)F * S * lnabytes;
    if (dst) {
      if (nb > dst_cap) raise(AASR_ERR_INVALID, "internal: result buffer too small");
      AASR_HIP(hipMemcpy(dst, d_bytes.p, nb, hipMemcpyDeviceToHost));
    } else {
      h_bytes.resize(nb);
      AASR_HIP(hipMemcpy(h_bytes.data(), d_bytes.p, nb, hipMemcpyDeviceToHost));
    }
    device_seconds += now_s() - t0;
  }
};

// The recipe driver's state that outlives a call (kept on the model handle).
struct RecipeScratch {
  BlockRunner br;
  uint8_t *pinned[kSlots] = {};
  size_t pinned_cap = 0;
  int16_t *pcm_ring = nullptr;   // pinned upload ring of the reader thread (samples)
  size_t ring_cap = 0;
  std::array<double, 10> timing{};  // aasr_recipe_last_timing
  ~RecipeScratch() {
    for (int i = 0; i < kSlots; i++)
      if (pinned[i]) (void)hipHostFree(pinned[i]);
    if (pcm_ring) (void)hipHostFree(pcm_ring);
  }
};

static void frame_limits(float start_time, float end_time, float fr, int *start_frame, int *end_frame) {
  // (int)(time * frame_rate) (aku/phone_probs.cc:199-206): both operands are float there
  // (Recipe::Info::start_time, FeatureGenerator::frame_rate()), so the product is rounded to
  // float before the truncation -- 2.008 s * 125 is frame 250, not 251
  volatile float start_prod = start_time * fr, end_prod = end_time * fr;
  *start_frame = (int)start_prod;
  *end_frame = (int)end_prod;
  if (*end_frame == 0) *end_frame = INT_MAX;
}

static void frame_range(aasr_feat *feat, int64_t n_samples, float start_time, float end_time,
                        int32_t *start, int32_t *count) {
  // the loop stops at the first frame for which AudioFileModule::eof() holds
  int start_frame, end_frame;
  frame_limits(start_time, end_time, feat->mods[0].frame_rate, &start_frame, &end_frame);
  if (feat->mods[0].type != MOD_PRE && n_samples < feat->mods[0].width + 1)
    raise(AASR_ERR_SHORT_AUDIO, "audio shorter than frame");
  int eof_frame = feat_eof_frame(feat, n_samples);
  int stop = std::min(end_frame, eof_frame);
  *start = start_frame;
  *count = stop > start_frame ? stop - start_frame : 0;
}

static void write_lna_file(const std::string &path, int32_t S, int lnabytes, const uint8_t *body,
                           size_t nbytes) {
  // write to a temporary name and rename: a killed run never leaves a
  // truncated .lna that --no-overwrite would later trust
  std::string tmp = path + ".tmp";
  FILE *fp = fopen(tmp.c_str(), "wb");
  if (!fp) raise(AASR_ERR_IO, "could not open %s for writing", tmp.c_str());
  uint8_t hdr[5];
  aasr_lna_header(S, lnabytes, hdr);
  bool ok = fwrite(hdr, 1, 5, fp) == 5 && (nbytes == 0 || fwrite(body, 1, nbytes, fp) == nbytes);
  ok = (fclose(fp) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
    remove(tmp.c_str());
    raise(AASR_ERR_IO, "Write error");
  }
}

void run_recipe(aasr_feat *feat, aasr_gmm *gmm, const std::string &recipe_path,
                const aasr_run_options &opt, aasr_run_stats *stats) {
  if (opt.lnabytes != 2 && opt.lnabytes != 4) raise(AASR_ERR_INVALID, "Invalid number of LNA bytes");
  if (gmm->dim != feat->mods.back().dim)
    raise(AASR_ERR_INVALID, "Gaussian dimension is %d but feature dimension is %d.", gmm->dim,
          feat->mods.back().dim);
  std::ifstream rin(recipe_path);
  if (!rin) raise(AASR_ERR_IO, "could not open recipe %s", recipe_path.c_str());
  std::stringstream ss;
  ss << rin.rdbuf();
  std::vector<RecipeInfo> infos = recipe_read(ss.str(), opt.num_batches, opt.batch_index);
  if (opt.sort_recipe)  // Recipe::sort_infos after the batch slice (aku/phone_probs.cc:140-141)
    std::stable_sort(infos.begin(), infos.end(), [](const RecipeInfo &a, const RecipeInfo &b) {
      return a.speaker_id < b.speaker_id;
    });
  std::string out_dir = opt.out_dir ? opt.out_dir : "";
  if (!out_dir.empty() && out_dir.back() != '/') out_dir += "/";

  double t_start = now_s();
  // Device buffers, the two streams and the pinned result slots live on the model handle: a second
  // recipe through the same handle (the bench's passes, a server) pays for none of them again --
  // pinning 3 x 0.25 GB alone is ~0.2 s.
  std::shared_ptr<RecipeScratch> scratch = std::static_pointer_cast<RecipeScratch>(gmm->recipe_scratch);
  if (!scratch) {
    scratch = std::make_shared<RecipeScratch>();
    gmm->recipe_scratch = scratch;
  }
  BlockRunner &br = scratch->br;
  br.feat = feat;
  br.gmm = gmm;
  br.lnabytes = opt.lnabytes;
  br.normalize = opt.normalize;
  br.device_seconds = br.copy_seconds = 0;
  const int64_t S = gmm->S;
  // frames per device block: ~0.75 GB of state scores + codes (40 000 frames at S = 3125: five
  // rounds of the scoring grid; the pinned result slots stay at 0.25 GB each)
  const int64_t block_frames = std::max<int64_t>(4096, (int64_t)(0.75e9 / (double)(S * (4 + opt.lnabytes))));

  // Three stages: a reader thread (recipe order: skip checks, audio files, frame ranges), this
  // thread (speaker settings, device blocks) and a pool of writer threads (LNA files), so file IO
  // overlaps the device.  Results travel through kSlots pinned buffers.
  struct Item {
    Job job;
    std::exception_ptr error;  // reading this utterance failed: rethrown in order
    bool end = false;
  };
  struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Item> q;
    int64_t frames = 0;
    bool abort = false;
  } inq;
  // results: kSlots pinned buffers ("slots"); the utterances of a finished block are handed to a
  // pool of writer threads one file each (one thread saturates at ~2.7 GB/s on tmpfs -- page
  // allocation -- which was 90 % of the wall time of a recipe run), the slot is free again when
  // its last file is closed
  struct WriteTask {
    int slot = -1;
    size_t job = 0;     // index into blocks[slot]
    size_t offset = 0;  // byte offset of the utterance's rows in the slot
    bool end = false;
  };
  struct OutQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<WriteTask> q;
    std::vector<Job> blocks[kSlots];
    size_t pending[kSlots] = {};
    bool slot_busy[kSlots] = {};
    bool others_idle(int slot) const {
      for (int i = 0; i < kSlots; i++)
        if (i != slot && slot_busy[i]) return false;
      return true;
    }
    std::exception_ptr error;
  } outq;
  uint8_t *(&pinned)[kSlots] = scratch->pinned;
  size_t &pinned_cap = scratch->pinned_cap;
  // Upload ring: the reader thread copies every file's samples into pinned memory, so that the calling thread's
  // uploads are asynchronous copies from pinned memory.  From the readers' pageable vectors hipMemcpyAsync stages and
  // BLOCKS: 0.14-0.67 s of the calling thread's 1.0-1.2 s on the 10 000-utterance recipe, varying with what the writer
  // pool left of the cores, and the device queue ran dry behind it.  Space is handed out in recipe order and returned in
  // recipe order when a block retires; a file that finds no room (or is larger than a quarter of the ring) keeps its vector.
  if (!scratch->pcm_ring) {
    const size_t cap = (size_t)64 << 20;   // samples: 128 MB
    if (hipHostMalloc((void **)&scratch->pcm_ring, cap * sizeof(int16_t), hipHostMallocDefault) == hipSuccess) scratch->ring_cap = cap;
    else { scratch->pcm_ring = nullptr; (void)hipGetLastError(); }
  }
  struct Ring {
    std::mutex m;
    uint64_t head = 0, tail = 0;   // virtual sample offsets: allocated up to head, free again up to tail
  } ring;
  int16_t *const ring_base = scratch->pcm_ring;
  const int64_t ring_limit = g_upload_ring_limit.load();   // (aasr_debug_set_upload_ring_samples: tests of the fallback)
  const size_t ring_cap = ring_limit > 0 ? std::min<size_t>((size_t)ring_limit, scratch->ring_cap) : scratch->ring_cap;

  // The reader only looks at the base module (mods[0]: sample rate, byte order, window, frame
  // rate); set_parameters is a no-op for audiofile / pre (FeatureModule::set_parameters,
  // aku/FeatureModule.hh:107), so the speaker settings applied by this thread never touch what
  // the read-ahead uses.
  std::thread reader([&] {
    for (size_t ri = 0; ri < infos.size(); ri++) {
      Item it;
      try {
        const RecipeInfo &info = infos[ri];
        if (opt.info > 0) {
          printf("Processing file %d/%d\n", (int)ri + 1, (int)infos.size());
          printf("Input: %s\n", info.audio_path.c_str());
        }
        std::string out_file = out_dir + info.lna_path;
        if (opt.afname) {
          std::string file = info.audio_path;
          size_t pos = file.rfind('/');
          if (pos != std::string::npos && pos + 1 < file.size()) file = file.substr(pos + 1);
          pos = file.rfind('.');
          if (pos != std::string::npos && pos > 0) file.erase(pos);
          out_file = out_dir + file + ".lna";
        }
        if (opt.info > 0) printf("Output: %s\n", out_file.c_str());
        if (opt.no_overwrite) {
          struct stat sb;
          if (stat(out_file.c_str(), &sb) == 0) {
            fprintf(stderr, "WARNING: skipping existing lna file %s\n", out_file.c_str());
            continue;
          }
        }
        it.job.info_index = ri;
        it.job.out_file = out_file;
        it.job.pcm = read_input_file(feat, info.audio_path, opt.raw_audio != 0);
        frame_range(feat, (int64_t)it.job.pcm.size(), info.start_time, info.end_time, &it.job.start,
                    &it.job.count);
        const size_t ns = it.job.pcm.size();
        if (ring_base && ns > 0 && ns <= ring_cap / 4) {
          // space is taken only when it is free NOW: the reader never waits for the ring (with few states a block is
          // more audio than the ring holds, and the calling thread would be waiting for this thread to fill it)
          uint64_t at;
          bool got;
          {
            std::lock_guard<std::mutex> lk(ring.m);
            at = ring.head;
            if (at % ring_cap + ns > ring_cap) at += ring_cap - at % ring_cap;   // no wrap inside a file
            got = at + ns - ring.tail <= ring_cap;
            if (got) ring.head = at + ns;
          }
          if (got) {
            memcpy(ring_base + at % ring_cap, it.job.pcm.data(), ns * sizeof(int16_t));
            it.job.pin = ring_base + at % ring_cap;
            it.job.pin_n = ns;
            it.job.pin_end = at + ns;
            std::vector<int16_t>().swap(it.job.pcm);
          }
        }
        if (opt.info > 0 && (it.job.start != 0 || info.end_time != 0))
          printf("Generating frames %d - %d\n", it.job.start, it.job.start + it.job.count);
      } catch (...) {
        it.error = std::current_exception();
      }
      const bool failed = (bool)it.error;
      {
        std::unique_lock<std::mutex> lk(inq.m);
        inq.cv.wait(lk, [&] { return inq.abort || inq.frames < 2 * block_frames; });
        if (inq.abort) return;
        inq.frames += it.job.count;
        inq.q.push_back(std::move(it));
      }
      inq.cv.notify_all();
      if (failed) return;  // the reference stops at the first error
    }
    {
      std::lock_guard<std::mutex> lk(inq.m);
      Item e;
      e.end = true;
      inq.q.push_back(std::move(e));
    }
    inq.cv.notify_all();
  });

  // One writer per usable core of this rank's share of the host (measured on a 16-CPU quota with one rank: 8 -> 5.2,
  // 12 -> 5.9, 16 -> 6.5, 24 -> 6.9 M frames/s, fresh files on tmpfs: a writer is a core's worth of page allocation +
  // copy, 2.7 GB/s).  N ranks on one host (aasr_set_host_share / LOCAL_WORLD_SIZE) divide the cores between them;
  // a constant 16 per rank put 128 writers on a 16-CPU quota at eight ranks.
  // (half as many writers again as cores: a writer that waits for a page does not hold its core -- 16 -> 24 threads on the
  // 16-CPU quota measured 6.5 -> 6.9 M frames/s)
  int n_writers = std::max(2, std::min(48, host_usable_cores() * 3 / 2 / host_share()));
  if (const char *e = getenv("AASR_WRITER_THREADS")) n_writers = std::max(1, std::min(64, atoi(e)));
  auto writer_main = [&] {
    // The writers are the throughput threads (a core's worth of page allocation + copy each) and outnumber the cores;
    // the reader and the calling thread are the two that keep the device fed.  Under the scheduler's fair share 25 runnable
    // threads on a 16-CPU quota left the reader 64 % of a core and the calling thread waited for it (0.06-0.65 s of
    // a 1.0-1.4 s run): the writers take a lower priority (a thread's own nice value; lowering needs no privilege).
    (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);
    for (;;) {
      WriteTask t;
      {
        std::unique_lock<std::mutex> lk(outq.m);
        outq.cv.wait(lk, [&] { return !outq.q.empty(); });
        t = outq.q.front();
        if (t.end) return;  // stays queued: every writer sees it
        outq.q.pop_front();
      }
      try {
        bool skip;
        {
          std::lock_guard<std::mutex> lk(outq.m);
          skip = (bool)outq.error;
        }
        if (!skip) {
          const Job &j = outq.blocks[t.slot][t.job];
          write_lna_file(j.out_file, (int32_t)S, opt.lnabytes, pinned[t.slot] + t.offset,
                         (size_t)j.count * S * opt.lnabytes);
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(outq.m);
        if (!outq.error) outq.error = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lk(outq.m);
        if (--outq.pending[t.slot] == 0) outq.slot_busy[t.slot] = false;
      }
      outq.cv.notify_all();
    }
  };
  std::vector<std::thread> writers;
  for (int i = 0; i < n_writers; i++) writers.emplace_back(writer_main);

  std::vector<Job> pending;
  int64_t pending_frames = 0, total_frames = 0, total_utts = 0;
  int next_slot = 0;
  std::exception_ptr failure;

  // AASR_RECIPE_TIMING=1: where the calling thread's time goes (stderr, one line per run)
  double t_wait_reader = 0, t_wait_slot = 0, t_launch = 0, t_finish = 0;
  int inflight_slot = -1;
  std::vector<Job> inflight_jobs;
  auto retire = [&]() {
    if (inflight_slot < 0) return;
    const int slot = inflight_slot;
    inflight_slot = -1;
    try {
      const double t0 = now_s();
      br.finish(slot);
      t_finish += now_s() - t0;
    } catch (...) {
      std::lock_guard<std::mutex> lk(outq.m);
      outq.slot_busy[slot] = false;
      throw;
    }
    uint64_t freed = 0;
    for (Job &j : inflight_jobs) {
      std::vector<int16_t>().swap(j.pcm);
      freed = std::max(freed, j.pin_end);
      j.pin = nullptr;
    }
    if (freed) {
      std::lock_guard<std::mutex> lk(ring.m);
      ring.tail = std::max(ring.tail, freed);
    }
    {
      std::lock_guard<std::mutex> lk(outq.m);
      outq.blocks[slot] = std::move(inflight_jobs);
      outq.pending[slot] = outq.blocks[slot].size();
      size_t off = 0;
      for (size_t k = 0; k < outq.blocks[slot].size(); k++) {
        WriteTask t;
        t.slot = slot;
        t.job = k;
        t.offset = off;
        outq.q.push_back(t);
        off += (size_t)outq.blocks[slot][k].count * S * opt.lnabytes;
      }
      if (outq.blocks[slot].empty()) outq.slot_busy[slot] = false;
    }
    inflight_jobs.clear();
    outq.cv.notify_all();
  };
  auto flush = [&]() {
    if (pending.empty()) return;
    const size_t need = (size_t)pending_frames * S * opt.lnabytes;
    const int slot = next_slot;
    next_slot = (next_slot + 1) % kSlots;
    if (need > pinned_cap) retire();  // growing the result buffers needs all of them idle
    const double t_slot0 = now_s();
    {
      std::unique_lock<std::mutex> lk(outq.m);
      outq.cv.wait(lk, [&] { return !outq.slot_busy[slot] && (need <= pinned_cap || outq.others_idle(slot)); });
      if (outq.error) std::rethrow_exception(outq.error);
      outq.slot_busy[slot] = true;
    }
    if (need > pinned_cap) {  // every slot idle here: grow them together
      const size_t cap = std::max(need, (size_t)block_frames * S * opt.lnabytes);
      for (int i = 0; i < kSlots; i++) {
        if (pinned[i]) (void)hipHostFree(pinned[i]);
        pinned[i] = nullptr;
        AASR_HIP(hipHostMalloc((void **)&pinned[i], cap, hipHostMallocDefault));
      }
      pinned_cap = cap;
    }
    std::vector<Job *> jobs;
    for (Job &j : pending) jobs.push_back(&j);
    t_wait_slot += now_s() - t_slot0;
    try {
      const double t0 = now_s();
      br.launch(jobs, slot, pinned[slot], pinned_cap);
      t_launch += now_s() - t0;
    } catch (...) {
      std::lock_guard<std::mutex> lk(outq.m);
      outq.slot_busy[slot] = false;
      throw;
    }
    // the block launched before this one has had the device to itself until now: collect it
    // (its copy overlaps the kernels just enqueued) and hand its utterances to the writers
    retire();
    inflight_slot = slot;
    inflight_jobs = std::move(pending);
    pending.clear();
    pending_frames = 0;
  };

  // -S: parameters change between utterances; everything queued so far must be
  // computed with the old ones
  struct Unhook {
    aasr_spkc *s;
    ~Unhook() { if (s) spkc_set_before_change(s, nullptr); }
  } unhook{opt.speakers};
  // a parameter change must find the device idle: the block in flight still runs with the old ones
  if (opt.speakers) spkc_set_before_change(opt.speakers, [&]() {
    flush();
    retire();
  });

  try {
    for (;;) {
      Item it;
      {
        const double t0 = now_s();
        std::unique_lock<std::mutex> lk(inq.m);
        inq.cv.wait(lk, [&] { return !inq.q.empty(); });
        t_wait_reader += now_s() - t0;
        it = std::move(inq.q.front());
        inq.q.pop_front();
        inq.frames -= it.job.count;
      }
      inq.cv.notify_all();
      if (it.end) break;
      if (it.error) {
        // the reference has written every LNA before the failing recipe line when it aborts:
        // finish what is queued, then report the original error
        try {
          flush();
          retire();
        } catch (...) {
        }
        std::rethrow_exception(it.error);
      }
      const RecipeInfo &info = infos[it.job.info_index];
      if (opt.speakers) {  // aku/phone_probs.cc:191-196
        spkc_set_speaker(opt.speakers, info.speaker_id);
        if (!info.utterance_id.empty()) spkc_set_utterance(opt.speakers, info.utterance_id);
      }
      if (pending_frames + it.job.count > block_frames) flush();
      pending_frames += it.job.count;
      total_frames += it.job.count;
      total_utts++;
      pending.push_back(std::move(it.job));
    }
    flush();
    retire();
  } catch (...) {
    failure = std::current_exception();
    // blocks may still be in flight: their uploads read the jobs' pageable PCM buffers and their kernels write
    // the scratch that stays on the handle -- nothing may be queued when those go out of scope or a later
    // recipe starts on the same handle
    if (br.s_compute) (void)hipStreamSynchronize(br.s_compute);
    if (br.s_copy) (void)hipStreamSynchronize(br.s_copy);
  }
  {  // stop the helpers (the writer first finishes what was queued)
    std::lock_guard<std::mutex> lk(inq.m);
    inq.abort = true;
  }
  inq.cv.notify_all();

  {
    std::lock_guard<std::mutex> lk(outq.m);
    WriteTask e;
    e.end = true;
    outq.q.push_back(e);
  }
  outq.cv.notify_all();
  reader.join();
  for (std::thread &t : writers) t.join();
  if (failure) std::rethrow_exception(failure);
  if (outq.error) std::rethrow_exception(outq.error);
  scratch->timing = {now_s() - t_start, t_wait_reader, t_wait_slot, t_launch, t_finish, br.device_seconds,
                     br.copy_seconds, (double)n_writers, (double)host_usable_cores(), (double)host_share()};
  if (getenv("AASR_RECIPE_TIMING"))
    fprintf(stderr, "recipe timing: total %.3f s; calling thread waited %.3f s for the reader, %.3f s for a result "
            "slot (pinned allocation, writers), %.3f s enqueueing blocks (incl. pageable uploads), %.3f s for copies\n",
            now_s() - t_start, t_wait_reader, t_wait_slot, t_launch, t_finish);
  if (stats) {
    stats->utterances = total_utts;
    stats->frames = total_frames;
    stats->seconds_total = now_s() - t_start;
    stats->seconds_device = br.device_seconds;
    stats->seconds_copy_out = br.copy_seconds;
  }
}

// One utterance through the whole path.  The packed rows are copied from the device straight into
// the malloc'ed result (header + rows); the device buffers live on the model handle between calls
// (a decoder front-end calls this once per utterance: no allocation after the first call of a size).
void run_utterance(aasr_feat *feat, aasr_gmm *gmm, const int16_t *pcm, int64_t n_samples,
                   int32_t start_frame, int32_t end_frame, int normalize, int lnabytes,
                   uint8_t **lna_out, int64_t *lna_len, int64_t *frames_out) {
  if (lnabytes != 2 && lnabytes != 4) raise(AASR_ERR_INVALID, "Invalid number of LNA bytes");
  if (gmm->dim != feat->mods.back().dim)
    raise(AASR_ERR_INVALID, "Gaussian dimension is %d but feature dimension is %d.", gmm->dim,
          feat->mods.back().dim);
  if (feat->mods[0].type != MOD_PRE && n_samples < feat->mods[0].width + 1)
    raise(AASR_ERR_SHORT_AUDIO, "audio shorter than frame");
  Job j;
  j.pcm.assign(pcm, pcm + n_samples);
  int eof_frame = feat_eof_frame(feat, n_samples);
  if (end_frame <= 0) end_frame = INT_MAX;
  int stop = std::min(end_frame, eof_frame);
  j.start = start_frame;
  j.count = stop > start_frame ? stop - start_frame : 0;
  std::shared_ptr<BlockRunner> br = std::static_pointer_cast<BlockRunner>(gmm->utt_scratch);
  if (!br) {
    br = std::make_shared<BlockRunner>();
    gmm->utt_scratch = br;
  }
  br->feat = feat;
  br->gmm = gmm;
  br->lnabytes = lnabytes;
  br->normalize = normalize;
  const size_t nb = (size_t)j.count * (size_t)gmm->S * (size_t)lnabytes;
  uint8_t *out = (uint8_t *)malloc(5 + nb);
  if (!out) raise(AASR_ERR_INVALID, "out of memory");
  try {
    aasr_lna_header((int32_t)gmm->S, lnabytes, out);
    std::vector<Job *> jobs{&j};
    br->run(jobs, out + 5, nb);
  } catch (...) {
    free(out);
    throw;
  }
  *lna_out = out;
  *lna_len = (int64_t)(5 + nb);
  *frames_out = j.count;
}

}  // namespace aasr

using namespace aasr;

extern "C" {

aasr_status aasr_recipe_batch_range(int32_t num_lines_total, int32_t num_batches,
                                    int32_t batch_index, int32_t *first_line, int32_t *num_lines) {
  return guarded([&] {
    if (!first_line || !num_lines || num_lines_total < 0)
      raise(AASR_ERR_INVALID, "aasr_recipe_batch_range: bad argument");
    int f = 0, c = 0;
    recipe_batch_range(num_lines_total, num_batches, batch_index, &f, &c);
    *first_line = f;
    *num_lines = c;
  });
}

void aasr_recipe_frame_limits(float start_time, float end_time, float frame_rate, int32_t *start_frame,
                              int32_t *end_frame) {
  int a, b;
  aasr::frame_limits(start_time, end_time, frame_rate, &a, &b);
  if (start_frame) *start_frame = a;
  if (end_frame) *end_frame = b;
}

aasr_status aasr_lna_read_file(const char *path, int32_t *num_states, int32_t *lnabytes, int64_t *frames,
                               float **log_probs) {
  return guarded([&] {
    if (!path || !num_states || !lnabytes || !frames || !log_probs)
      raise(AASR_ERR_INVALID, "aasr_lna_read_file: null argument");
    FILE *f = fopen(path, "rb");
    if (!f) raise(AASR_ERR_IO, "aasr_lna_read_file: could not open %s", path);
    unsigned char head[5];
    if (fread(head, 1, 5, f) != 5) {
      fclose(f);
      raise(AASR_ERR_IO, "aasr_lna_read_file: no LNA header in %s", path);
    }
    const int64_t S = ((int64_t)head[0] << 24) | (head[1] << 16) | (head[2] << 8) | head[3];
    const int nb = head[4];
    if (S <= 0 || S > 0x7fffffff || (nb != 1 && nb != 2 && nb != 4)) {
      fclose(f);
      raise(AASR_ERR_INVALID, "aasr_lna_read_file: invalid header in %s", path);
    }
    std::vector<unsigned char> raw;
    std::vector<unsigned char> chunk(1 << 22);
    size_t got;
    while ((got = fread(chunk.data(), 1, chunk.size(), f)) > 0) raw.insert(raw.end(), chunk.begin(), chunk.begin() + got);
    fclose(f);
    const int64_t F = (int64_t)(raw.size() / (size_t)(S * nb));
    float *out = (float *)malloc(sizeof(float) * (size_t)std::max<int64_t>(F * S, 1));
    if (!out) raise(AASR_ERR_INVALID, "aasr_lna_read_file: out of memory");
    const unsigned char *p = raw.data();
    const int64_t n = F * S;
    if (nb == 4) {
      memcpy(out, p, (size_t)n * 4);  // little-endian floats on a little-endian host
    } else if (nb == 2) {
      for (int64_t i = 0; i < n; i++) out[i] = (float)((p[2 * i] * 256 + p[2 * i + 1]) / -1820.0);
    } else {
      for (int64_t i = 0; i < n; i++) out[i] = (float)(p[i] / -24.0);
    }
    *num_states = (int32_t)S;
    *lnabytes = nb;
    *frames = F;
    *log_probs = out;
  });
}

aasr_status aasr_recipe_read_all(const char *recipe_text, int32_t num_batches, int32_t batch_index,
                                 int32_t cluster_speakers, char **table_out, int64_t *table_len) {
  return guarded([&] {
    if (!recipe_text || !table_out || !table_len) raise(AASR_ERR_INVALID, "aasr_recipe_read_all: null argument");
    std::string t;
    char num[96];
    for (const RecipeInfo &i : recipe_read(recipe_text, num_batches, batch_index, cluster_speakers != 0)) {
      for (const std::string *f : {&i.audio_path, &i.alt_audio_path, &i.transcript_path, &i.alignment_path,
                                   &i.hmmnet_path, &i.den_hmmnet_path, &i.lna_path})
        t += *f + "\x1f";
      snprintf(num, sizeof num, "%.9g\x1f%.9g\x1f%d\x1f%d\x1f", (double)i.start_time, (double)i.end_time,
               i.start_line, i.end_line);
      t += num;
      t += i.speaker_id + "\x1f" + i.utterance_id + "\n";
    }
    char *out = (char *)malloc(t.size() + 1);
    if (!out) raise(AASR_ERR_INVALID, "aasr_recipe_read_all: out of memory");
    memcpy(out, t.c_str(), t.size() + 1);
    *table_out = out;
    *table_len = (int64_t)t.size();
  });
}

aasr_status aasr_recipe_read(const char *recipe_text, int32_t num_batches, int32_t batch_index,
                             char **table_out, int64_t *table_len) {
  return guarded([&] {
    if (!recipe_text || !table_out || !table_len) raise(AASR_ERR_INVALID, "aasr_recipe_read: null argument");
    std::string t;
    char num[64];
    for (const RecipeInfo &i : recipe_read(recipe_text, num_batches, batch_index)) {
      t += i.audio_path + "\x1f" + i.lna_path + "\x1f" + i.speaker_id + "\x1f" + i.utterance_id + "\x1f";
      snprintf(num, sizeof num, "%.9g\x1f%.9g\n", (double)i.start_time, (double)i.end_time);
      t += num;
    }
    char *out = (char *)malloc(t.size() + 1);
    if (!out) raise(AASR_ERR_INVALID, "aasr_recipe_read: out of memory");
    memcpy(out, t.c_str(), t.size() + 1);
    *table_out = out;
    *table_len = (int64_t)t.size();
  });
}

aasr_status aasr_run_recipe(aasr_feat *feat, aasr_gmm *gmm, const char *recipe_path,
                            const aasr_run_options *opt, aasr_run_stats *stats) {
  return guarded([&] {
    if (!feat || !gmm || !recipe_path || !opt) raise(AASR_ERR_INVALID, "aasr_run_recipe: null argument");
    run_recipe(feat, gmm, recipe_path, *opt, stats);
  });
}

aasr_status aasr_set_host_share(int32_t processes) {
  return guarded([&] {
    if (processes < 0) raise(AASR_ERR_INVALID, "aasr_set_host_share: negative process count");
    g_host_share.store(processes);
  });
}

int32_t aasr_host_usable_cores(void) { return (int32_t)host_usable_cores(); }

// Diagnostic (not part of the public ABI): the recipe driver's pinned upload ring uses its first `samples` samples only
// (0: all of it), so that a test can make files miss the ring and keep their pageable buffers.
void aasr_debug_set_upload_ring_samples(int64_t samples) { g_upload_ring_limit.store(samples); }

aasr_status aasr_recipe_last_timing(const aasr_gmm *gmm, aasr_recipe_timing *out) {
  return guarded([&] {
    if (!gmm || !out) raise(AASR_ERR_INVALID, "aasr_recipe_last_timing: null argument");
    std::shared_ptr<RecipeScratch> sc = std::static_pointer_cast<RecipeScratch>(gmm->recipe_scratch);
    if (!sc) raise(AASR_ERR_INVALID, "aasr_recipe_last_timing: no recipe has run on this handle");
    const std::array<double, 10> &t = sc->timing;
    out->seconds_total = t[0];
    out->wait_reader = t[1];
    out->wait_result_slot = t[2];
    out->enqueue = t[3];
    out->wait_copies = t[4];
    out->device = t[5];
    out->copy_out = t[6];
    out->writer_threads = (int32_t)t[7];
    out->usable_cores = (int32_t)t[8];
    out->host_share = (int32_t)t[9];
  });
}

aasr_status aasr_run_utterance(aasr_feat *feat, aasr_gmm *gmm, const int16_t *pcm,
                               int64_t n_samples, int32_t start_frame, int32_t end_frame,
                               int normalize, int lnabytes, uint8_t **lna_out, int64_t *lna_len,
                               int64_t *frames_out) {
  return guarded([&] {
    if (!feat || !gmm || !pcm || !lna_out || !lna_len)
      raise(AASR_ERR_INVALID, "aasr_run_utterance: null argument");
    int64_t frames = 0;
    run_utterance(feat, gmm, pcm, n_samples, start_frame, end_frame, normalize, lnabytes, lna_out, lna_len,
                  &frames);
    if (frames_out) *frames_out = frames;
  });
}

}  // extern "C"
