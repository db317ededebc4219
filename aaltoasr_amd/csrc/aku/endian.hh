// endian.hh -- byte order helpers with the interface of aku/endian.hh (endian::big,
// endian::convert), used by phone_probs.cc:241-242 for the 4-byte LNA floats.
#ifndef AKU_AMD_ENDIAN_HH
#define AKU_AMD_ENDIAN_HH

#include <cstddef>

namespace endian {

inline bool detect_big() {
  const unsigned short probe = 1;
  return *reinterpret_cast<const unsigned char *>(&probe) == 0;
}
static const bool big = detect_big();

/** reverses `len` bytes in place */
inline void convert(void *buf, size_t len) {
  unsigned char *p = static_cast<unsigned char *>(buf);
  for (size_t i = 0; i < len / 2; i++) {
    const unsigned char t = p[i];
    p[i] = p[len - 1 - i];
    p[len - 1 - i] = t;
  }
}

}  // namespace endian

#endif
