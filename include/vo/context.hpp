// vo/context.hpp -- process-wide GPU context used by the facade classes, and
// the error convention: the reference signals failure with bool/cout; a failed
// libvo_hip call (no device, HIP error, bad index) throws vo::Error instead of
// silently computing something else.
#pragma once

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../vo_hip.h"

namespace vo {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc, const char* where) {
  if (rc != VO_OK) throw Error(rc, std::string(where) + ": " + vo_last_error());
}

class Context {
 public:
  explicit Context(int device = 0, void* stream = nullptr) { check(vo_ctx_create(device, stream, &h_), "vo_ctx_create"); }
  ~Context() { vo_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  vo_ctx* handle() const { return h_; }
  void synchronize() const { check(vo_ctx_synchronize(h_), "vo_ctx_synchronize"); }

 private:
  vo_ctx* h_ = nullptr;
};

// One context per process, on device $VO_DEVICE (default 0), created on first use.
inline Context& default_context() {
  static Context ctx([] { const char* e = std::getenv("VO_DEVICE"); return e ? std::atoi(e) : 0; }());
  return ctx;
}

}  // namespace vo
