// common.h -- shared host-side helpers for libaasr (error state, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/aasr.h"

// Environment variables.  The product library reads the few that INTEGRATION.md lists (AASR_PREC, AASR_PCGMM_AS_WRITTEN,
// AASR_WRITER_THREADS, AASR_RECIPE_TIMING and the test hooks AASR_F16_PROBE_TOL / AASR_PG_PIVOT_COST) with plain getenv.
// Every other switch is an EXPERIMENT switch -- it selects a kernel variant, a layout or an arithmetic for an A/B
// measurement -- and exists only in a build made with AASR_BUILD_ABLATION=1 (-DAASR_ABLATION=1, into lib_ablation/): in
// the product library the expression below is a null pointer and the name is not even in the object's strings
// (tests/test_env_switches.py checks exactly that).
#ifndef AASR_ABLATION
#define AASR_ABLATION 0
#endif
#if AASR_ABLATION
#define AASR_EXPERIMENT_ENV(name) getenv(name)
#else
#define AASR_EXPERIMENT_ENV(name) ((const char *)nullptr)
#endif

namespace aasr {

// thread-local message behind aasr_last_error()
std::string &last_error();
aasr_status fail(aasr_status code, const char *fmt, ...);

// Thrown inside the library, converted to a status at the C boundary.
struct Error {
  aasr_status code;
  std::string msg;
};

[[noreturn]] void raise(aasr_status code, const char *fmt, ...);

#define AASR_HIP(expr)                                                        \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess)                                                     \
      ::aasr::raise(AASR_ERR_NO_DEVICE, "%s failed: %s (%s:%d)", #expr,       \
                    hipGetErrorString(_e), __FILE__, __LINE__);               \
  } while (0)

// Runs body, maps exceptions to status codes.
template <class F>
aasr_status guarded(F &&body) {
  try {
    body();
    return AASR_OK;
  } catch (const Error &e) {
    last_error() = e.msg;
    return e.code;
  } catch (const std::exception &e) {
    last_error() = e.what();
    return AASR_ERR_INVALID;
  } catch (const std::string &s) {
    last_error() = s;
    return AASR_ERR_INVALID;
  }
}

// device buffer with RAII
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) {
    o.p = nullptr;
    o.n = 0;
  }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      n = o.n;
      o.p = nullptr;
      o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) return;
    AASR_HIP(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void ensure(size_t count) {
    if (count > n) alloc(count);
  }
  void upload(const T *src, size_t count) {
    alloc(count);
    if (count) AASR_HIP(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  }
};

void require_device();

}  // namespace aasr
