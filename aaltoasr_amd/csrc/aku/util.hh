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

/** aku/util.hh:62-130: maximum, and log(e^a + e^b) in natural and decimal logarithms */
template <typename T>
T max(const T &a, const T &b) {
  if (a < b) return b;
  return a;
}
inline float logaddf(float a, float b) {
  float delta = a - b;
  if (delta > 0) {
    b = a;
    delta = -delta;
  }
  return b + log1pf(expf(delta));
}
inline double logadd(double a, double b) {
  double delta = a - b;
  if (delta > 0) {
    b = a;
    delta = -delta;
  }
  return b + log1p(exp(delta));
}

}  // namespace util

#endif
