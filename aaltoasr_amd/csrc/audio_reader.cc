// audio_reader.cc -- host-side audio input of the audiofile module.
//
// Replaces the libsndfile calls of AudioReader::open / check_audio_parameters /
// read_from_file (aku/AudioReader.cc:86-110, 145-156, 170-213).  libsndfile is a
// system dependency of the reference (no version pinned in its CMake files) and
// is not in this image, so the container parsing and the sf_read_short()
// conversions are restated from libsndfile 1.0.x's published behaviour:
//   * containers: RIFF/WAVE (PCM, EXTENSIBLE, A-law, mu-law), Sun/NeXT AU,
//     AIFF / AIFF-C (NONE, sowt, ulaw, alaw), NIST SPHERE (pcm, ulaw, alaw);
//   * sf_read_short from integer PCM keeps the 16 most significant bits
//     (8-bit: value << 8, WAV 8-bit is unsigned; 24/32-bit: arithmetic shift),
//     G.711 codes expand through the standard tables;
//   * anything sf_open() would not recognise falls back to headerless PCM16 in
//     the configured byte order, exactly as AudioReader::open does (:94-108).
// Deliberate differences (loud instead of silent garbage): float / double /
// ADPCM / shorten-compressed payloads raise AASR_ERR_UNSUPPORTED -- the
// reference would either print a warning and read values in [-1, 1] as shorts
// or decode the compressed bytes as raw PCM.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "common.h"
#include "pipeline.h"

namespace aasr {

namespace {

enum Coding { PCM_S8, PCM_U8, PCM_16, PCM_24, PCM_32, ULAW, ALAW };

struct Payload {
  size_t offset = 0, nbytes = 0;
  Coding coding = PCM_16;
  bool big_endian = false;
  int channels = 1, rate = 0;
};

struct Bytes {
  const uint8_t *p;
  size_t n;
  uint32_t le32(size_t o) const { return p[o] | (p[o + 1] << 8) | (p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); }
  uint32_t be32(size_t o) const { return ((uint32_t)p[o] << 24) | (p[o + 1] << 16) | (p[o + 2] << 8) | p[o + 3]; }
  uint16_t le16(size_t o) const { return (uint16_t)(p[o] | (p[o + 1] << 8)); }
  uint16_t be16(size_t o) const { return (uint16_t)((p[o] << 8) | p[o + 1]); }
  bool tag(size_t o, const char *t) const { return o + 4 <= n && !memcmp(p + o, t, 4); }
};

int16_t ulaw_to_short(uint8_t code) {
  const int u = ~code & 0xFF;
  const int t = (((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7);
  return (int16_t)((u & 0x80) ? 0x84 - t : t - 0x84);
}

int16_t alaw_to_short(uint8_t code) {
  const int a = code ^ 0x55;
  const int e = (a >> 4) & 7, m = a & 0x0F;
  const int t = e == 0 ? (m << 4) + 8 : ((m << 4) + 0x108) << (e - 1);
  return (int16_t)((a & 0x80) ? t : -t);
}

Coding pcm_bits(int bits, bool unsigned8, const std::string &name) {
  switch (bits) {
    case 8: return unsigned8 ? PCM_U8 : PCM_S8;
    case 16: return PCM_16;
    case 24: return PCM_24;
    case 32: return PCM_32;
  }
  raise(AASR_ERR_UNSUPPORTED, "audio file %s: %d-bit PCM is not built", name.c_str(), bits);
}

// ---- RIFF / WAVE ------------------------------------------------------------
bool parse_wav(const Bytes &b, const std::string &name, Payload &out) {
  if (!(b.n >= 12 && b.tag(0, "RIFF") && b.tag(8, "WAVE"))) return false;
  size_t pos = 12;
  bool have_fmt = false;
  int fmt = 0, bits = 0;
  while (pos + 8 <= b.n) {
    const uint32_t len = b.le32(pos + 4);
    if (b.tag(pos, "fmt ") && pos + 8 + 16 <= b.n) {
      fmt = b.le16(pos + 8);
      out.channels = b.le16(pos + 10);
      out.rate = (int)b.le32(pos + 12);
      bits = b.le16(pos + 22);
      if (fmt == 0xFFFE && len >= 40 && pos + 8 + 26 <= b.n) fmt = b.le16(pos + 8 + 24);  // sub-format GUID
      have_fmt = true;
    } else if (b.tag(pos, "data")) {
      if (!have_fmt) raise(AASR_ERR_IO, "malformed WAV file: %s", name.c_str());
      out.offset = pos + 8;
      out.nbytes = std::min<size_t>(len, b.n - out.offset);
      if (len == 0 || len == 0xFFFFFFFFu) out.nbytes = b.n - out.offset;  // streamed writers
      out.big_endian = false;
      if (fmt == 1) out.coding = pcm_bits(bits, true, name);
      else if (fmt == 6) out.coding = ALAW;
      else if (fmt == 7) out.coding = ULAW;
      else
        raise(AASR_ERR_UNSUPPORTED, "audio sample format is not integer PCM or G.711 (WAVE format %d, %d bits): %s",
              fmt, bits, name.c_str());
      return true;
    }
    pos += 8 + (size_t)len + (len & 1);
  }
  raise(AASR_ERR_IO, "malformed WAV file: %s", name.c_str());
}

// ---- Sun / NeXT AU ------------------------------------------------------------
bool parse_au(const Bytes &b, const std::string &name, Payload &out) {
  if (b.n < 24) return false;
  const bool be = b.tag(0, ".snd"), le = b.tag(0, "dns.");
  if (!be && !le) return false;
  auto u32 = [&](size_t o) { return be ? b.be32(o) : b.le32(o); };
  out.offset = u32(4);
  const uint32_t size = u32(8), enc = u32(12);
  out.rate = (int)u32(16);
  out.channels = (int)u32(20);
  if (out.offset < 24 || out.offset > b.n) raise(AASR_ERR_IO, "malformed AU file: %s", name.c_str());
  out.nbytes = std::min<size_t>(size == 0xFFFFFFFFu ? b.n : size, b.n - out.offset);
  out.big_endian = be;
  switch (enc) {
    case 1: out.coding = ULAW; break;
    case 2: out.coding = PCM_S8; break;
    case 3: out.coding = PCM_16; break;
    case 4: out.coding = PCM_24; break;
    case 5: out.coding = PCM_32; break;
    case 27: out.coding = ALAW; break;
    default:
      raise(AASR_ERR_UNSUPPORTED, "audio sample format is not integer PCM or G.711 (AU encoding %u): %s", enc,
            name.c_str());
  }
  return true;
}

// ---- AIFF / AIFF-C ------------------------------------------------------------
bool parse_aiff(const Bytes &b, const std::string &name, Payload &out) {
  if (!(b.n >= 12 && b.tag(0, "FORM") && (b.tag(8, "AIFF") || b.tag(8, "AIFC")))) return false;
  const bool aifc = b.tag(8, "AIFC");
  size_t pos = 12;
  bool have_comm = false, have_data = false;
  int bits = 0;
  uint32_t frames = 0;
  out.big_endian = true;
  while (pos + 8 <= b.n) {
    const uint32_t len = b.be32(pos + 4);
    if (b.tag(pos, "COMM") && pos + 8 + 18 <= b.n) {
      out.channels = b.be16(pos + 8);
      frames = b.be32(pos + 10);
      bits = b.be16(pos + 14);
      // 80-bit extended sample rate: sign/exponent (15-bit bias 16383) + 64-bit mantissa
      const int e = (b.be16(pos + 16) & 0x7FFF) - 16383;
      const uint32_t hi = b.be32(pos + 18);
      out.rate = (e >= 0 && e < 32) ? (int)(hi >> (31 - e)) : 0;
      out.coding = pcm_bits(bits <= 8 ? 8 : bits <= 16 ? 16 : bits <= 24 ? 24 : 32, false, name);
      if (aifc && len >= 22 && pos + 8 + 22 <= b.n) {
        if (b.tag(pos + 26, "NONE") || b.tag(pos + 26, "twos")) {
        } else if (b.tag(pos + 26, "sowt")) {
          out.big_endian = false;
        } else if (b.tag(pos + 26, "ulaw") || b.tag(pos + 26, "ULAW")) {
          out.coding = ULAW;
        } else if (b.tag(pos + 26, "alaw") || b.tag(pos + 26, "ALAW")) {
          out.coding = ALAW;
        } else {
          raise(AASR_ERR_UNSUPPORTED, "audio sample format is not integer PCM or G.711 (AIFF-C '%.4s'): %s",
                (const char *)b.p + pos + 26, name.c_str());
        }
      }
      have_comm = true;
    } else if (b.tag(pos, "SSND") && pos + 16 <= b.n) {
      out.offset = pos + 16 + b.be32(pos + 8);
      have_data = true;
    }
    pos += 8 + (size_t)len + (len & 1);
  }
  if (!have_comm || !have_data || out.offset > b.n) raise(AASR_ERR_IO, "malformed AIFF file: %s", name.c_str());
  const size_t width = out.coding == PCM_16 ? 2 : out.coding == PCM_24 ? 3 : out.coding == PCM_32 ? 4 : 1;
  out.nbytes = std::min<size_t>((size_t)frames * width * (size_t)std::max(out.channels, 1), b.n - out.offset);
  return true;
}

// ---- NIST SPHERE ------------------------------------------------------------
bool parse_nist(const Bytes &b, const std::string &name, Payload &out) {
  if (!(b.n >= 16 && !memcmp(b.p, "NIST_1A\n", 8))) return false;
  const size_t hsize = (size_t)strtol((const char *)b.p + 8, nullptr, 10);
  if (hsize < 16 || hsize > b.n) raise(AASR_ERR_IO, "malformed NIST SPHERE file: %s", name.c_str());
  const std::string head((const char *)b.p, hsize);
  auto field = [&](const char *key, std::string &val) {
    size_t at = 0;
    const std::string k = std::string("\n") + key + " -";
    if ((at = head.find(k)) == std::string::npos) return false;
    at = head.find(' ', at + k.size());  // skip the type token (i, r, sN)
    if (at == std::string::npos) return false;
    const size_t end = head.find('\n', at);
    val = head.substr(at + 1, end == std::string::npos ? std::string::npos : end - at - 1);
    while (!val.empty() && (val.back() == ' ' || val.back() == '\r')) val.pop_back();
    return true;
  };
  std::string v;
  int nbytes = 2;
  long count = -1;
  out.channels = 1;
  if (field("channel_count", v)) out.channels = atoi(v.c_str());
  if (field("sample_rate", v)) out.rate = atoi(v.c_str());
  if (field("sample_n_bytes", v)) nbytes = atoi(v.c_str());
  if (field("sample_count", v)) count = atol(v.c_str());
  out.big_endian = false;
  if (field("sample_byte_format", v)) out.big_endian = v.compare(0, 2, "10") == 0;
  std::string coding = "pcm";
  field("sample_coding", coding);
  if (coding == "pcm") out.coding = pcm_bits(8 * nbytes, false, name);
  else if (coding == "ulaw" || coding == "mu-law") out.coding = ULAW;
  else if (coding == "alaw") out.coding = ALAW;
  else
    raise(AASR_ERR_UNSUPPORTED, "NIST SPHERE sample_coding '%s' is not built (decompress the file first): %s",
          coding.c_str(), name.c_str());
  out.offset = hsize;
  out.nbytes = b.n - hsize;
  if (count >= 0) {
    const size_t width = (out.coding == ULAW || out.coding == ALAW) ? 1 : (size_t)nbytes;
    out.nbytes = std::min<size_t>(out.nbytes, (size_t)count * width * (size_t)std::max(out.channels, 1));
  }
  return true;
}

std::vector<int16_t> expand(const Bytes &b, const Payload &pl) {
  const uint8_t *s = b.p + pl.offset;
  std::vector<int16_t> pcm;
  switch (pl.coding) {
    case PCM_S8:
      pcm.resize(pl.nbytes);
      for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (int16_t)((int8_t)s[i] * 256);
      break;
    case PCM_U8:
      pcm.resize(pl.nbytes);
      for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (int16_t)(((int)s[i] - 128) * 256);
      break;
    case ULAW:
      pcm.resize(pl.nbytes);
      for (size_t i = 0; i < pcm.size(); i++) pcm[i] = ulaw_to_short(s[i]);
      break;
    case ALAW:
      pcm.resize(pl.nbytes);
      for (size_t i = 0; i < pcm.size(); i++) pcm[i] = alaw_to_short(s[i]);
      break;
    case PCM_16:
    case PCM_24:
    case PCM_32: {
      // the 16 most significant bits of each sample
      const size_t w = pl.coding == PCM_16 ? 2 : pl.coding == PCM_24 ? 3 : 4;
      pcm.resize(pl.nbytes / w);
      const size_t hi = pl.big_endian ? 0 : w - 1, lo = pl.big_endian ? 1 : w - 2;
      for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (int16_t)((s[i * w + hi] << 8) | s[i * w + lo]);
      break;
    }
  }
  return pcm;
}

}  // namespace

std::vector<int16_t> decode_audio(const std::vector<char> &data, const std::string &name, bool force_raw,
                                  bool big_endian, int expect_rate, int *rate_out) {
  const Bytes b{(const uint8_t *)data.data(), data.size()};
  Payload pl;
  bool container = false;
  if (!force_raw)
    container = parse_wav(b, name, pl) || parse_au(b, name, pl) || parse_aiff(b, name, pl) || parse_nist(b, name, pl);
  if (!container) {
    // AudioReader::open's RAW mode (aku/AudioReader.cc:94-104): PCM16, one channel, the configured rate
    pl = Payload();
    pl.nbytes = b.n;
    pl.big_endian = big_endian;
    pl.rate = expect_rate;
  }
  if (pl.channels != 1)
    raise(AASR_ERR_INVALID, "AudioReader: sorry, audio files with multiple channels not supported");
  if (container && expect_rate > 0 && pl.rate != expect_rate)
    raise(AASR_ERR_INVALID, "Audio file sample rate (%d Hz) and model configuration (%d Hz) don't agree.", pl.rate,
          expect_rate);
  if (rate_out) *rate_out = pl.rate;
  return expand(b, pl);
}

std::vector<int16_t> read_audio_file(const std::string &path, bool force_raw, int expect_rate, bool big_endian,
                                     int *rate_out) {
  // one bulk read (a character iterator over the stream moved ~0.4 GB/s: the reader thread of the
  // recipe driver was the slowest stage once the LNA writers were parallel)
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) raise(AASR_ERR_IO, "AudioReader::open(): could not open file:%s", path.c_str());
  std::vector<char> data;
  if (fseek(fp, 0, SEEK_END) == 0) {
    const long n = ftell(fp);
    rewind(fp);
    if (n > 0) {
      data.resize((size_t)n);
      const size_t got = fread(data.data(), 1, (size_t)n, fp);
      data.resize(got);
    }
  }
  if (data.empty()) {  // not seekable (a pipe): read in pieces
    char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + got);
  }
  fclose(fp);
  return decode_audio(data, path, force_raw, big_endian, expect_rate, rate_out);
}

}  // namespace aasr
