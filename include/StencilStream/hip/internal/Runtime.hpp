// C++ convenience layer over the C ABI of libststhip.so (include/ststhip.h): turns status codes
// into exceptions and owns nothing.  Used by stencil::hip::Grid and stencil::hip::StencilUpdate.
#pragma once
#include <ststhip.h>

#include <stdexcept>
#include <string>

namespace stencil {
namespace hip {
namespace internal {

class runtime_error : public std::runtime_error {
  public:
    runtime_error(int status, const char *what_failed)
        : std::runtime_error(std::string("ststhip: ") + what_failed + ": " + ststhip_last_error()),
          status(status) {}
    int status;
};

inline void check(int status, const char *what_failed) {
    if (status != STSTHIP_OK)
        throw runtime_error(status, what_failed);
}

// Make sure the runtime is up on the requested device (-1 = current device).
inline void ensure_runtime(int device_index) { check(ststhip_init(device_index), "ststhip_init"); }

inline ststhip_stream default_stream() {
    ststhip_stream s = nullptr;
    check(ststhip_default_stream(&s), "ststhip_default_stream");
    return s;
}

inline void *device_alloc(std::size_t bytes) {
    void *p = nullptr;
    check(ststhip_malloc(&p, bytes ? bytes : 1), "ststhip_malloc");
    return p;
}

// for work on `stream`: a pooled block released on another stream is waited for on `stream`, not by the host
inline void *device_alloc_on(std::size_t bytes, ststhip_stream stream) {
    void *p = nullptr;
    check(ststhip_malloc_async(&p, bytes ? bytes : 1, stream), "ststhip_malloc_async");
    return p;
}

inline void *pinned_alloc(std::size_t bytes) {
    void *p = nullptr;
    check(ststhip_host_malloc(&p, bytes ? bytes : 1), "ststhip_host_malloc");
    return p;
}

} // namespace internal
} // namespace hip
} // namespace stencil
