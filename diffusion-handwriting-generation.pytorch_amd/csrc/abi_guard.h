// abi_guard.h — the no-throw barrier of the C-ABI.
//
// include/dhw.h promises "nothing throws across the ABI": the callers are ctypes / cgo-style bindings, for which a C++
// exception that leaves an extern "C" function is std::terminate — the host process aborts (round 4: a std::out_of_range
// from a string-keyed workspace lookup escaped dhw_forward and killed pytest three times, once leaving a process stuck under
// rocprofv3).  Every extern "C" entry point of dhw_api.cpp, dhw_style_api.cpp and dhw_train_api.cpp runs its body inside
// abi_guard(): any exception becomes DHW_ERR_INTERNAL (-5) plus a message readable through the library's *_last_error.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

// A fixed error-message buffer: setting it cannot throw (a std::string assignment can, and the failure being reported may be
// std::bad_alloc itself).
struct ErrBuf {
  char s[512];
  ErrBuf() noexcept { s[0] = 0; }
  void set(const char* m) noexcept {
    std::strncpy(s, m ? m : "", sizeof s - 1);
    s[sizeof s - 1] = 0;
  }
  void vsetf(const char* fmt, va_list ap) noexcept { vsnprintf(s, sizeof s, fmt, ap); }
  const char* c_str() const noexcept { return s; }
  bool empty() const noexcept { return s[0] == 0; }
};

// R = the entry point's return type (int, int64_t); on_fail(fn, what) records the message and returns the negative status.
template <typename R, typename Fail, typename Body>
inline R abi_guard(const char* fn, Fail&& on_fail, Body&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return (R)on_fail(fn, "out of host memory (std::bad_alloc)");
  } catch (const std::exception& e) {
    return (R)on_fail(fn, e.what());
  } catch (...) {
    return (R)on_fail(fn, "unknown C++ exception");
  }
}

// what a debug entry raises on request (dhw_debug_raise: tests of the barrier itself)
enum { DHW_RAISE_OUT_OF_RANGE = 1, DHW_RAISE_BAD_ALLOC = 2, DHW_RAISE_UNKNOWN = 3 };
