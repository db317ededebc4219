// util.hh -- the few helpers of aku/util.hh callers of the scoring path use.
#ifndef AKU_AMD_UTIL_HH
#define AKU_AMD_UTIL_HH

#include <cmath>

namespace util {

/** aku/util.hh:14,132-139: log(x) floored at log(1e-50) */
static const double tiny_for_log = 1e-50;
inline double safe_log(double x) {
  if (x < tiny_for_log) return std::log(tiny_for_log);
  return std::log(x);
}

/** aku/util.hh: mathematical modulo (non-negative result) */
inline int modulo(int a, int b) {
  int r = a % b;
  if (r < 0) r += b;
  return r;
}

template <class T>
inline T sqr(T a) { return a * a; }

}  // namespace util

#endif
