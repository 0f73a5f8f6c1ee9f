// Host-side SYCL *vocabulary* for StencilStream user code on the MI355X backend.
//
// This is NOT a SYCL runtime.  StencilStream's public API leaks a handful of
// SYCL value types into application code (sycl::id/range/buffer/host_accessor,
// access modes, sycl::device, sycl::exception_list, sycl::cos/exp/isinf; list in
// SURVEY.md section 8b).  They are provided here as ordinary C++ types so that
// application sources written against the reference compile unchanged; all
// device work goes through the HIP backend (StencilStream/hip) and the C-ABI
// runtime (include/ststhip.h).
#pragma once
#include "id.hpp"
#include "range.hpp"

#include <bit>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

namespace sycl {

namespace access {
enum class mode { read = 1024, write, read_write, discard_write, discard_read_write, atomic };
enum class target { device, host_task, host_buffer };
} // namespace access
using access_mode = access::mode;

template <access::mode M> struct mode_tag_t {
    explicit constexpr mode_tag_t() = default;
};
inline constexpr mode_tag_t<access::mode::read> read_only{};
inline constexpr mode_tag_t<access::mode::write> write_only{};
inline constexpr mode_tag_t<access::mode::read_write> read_write{};

// Names the accelerator a StencilUpdate runs on.  Default = the process' current HIP device
// (index -1 leaves the choice to the runtime, see ststhip_init).
class device {
  public:
    constexpr device() = default;
    constexpr explicit device(int hip_device_index) : index(hip_device_index) {}
    constexpr int hip_index() const { return index; }

  private:
    int index = -1;
};

class exception : public std::runtime_error {
  public:
    using std::runtime_error::runtime_error;
    exception() : std::runtime_error("sycl::exception") {}
};

using exception_list = std::vector<std::exception_ptr>;

// Shared, reference-counted H x W (x D) array of T on the host.
template <typename T, int N = 1> class buffer {
  public:
    using value_type = T;

    buffer(range<N> r) : extent(r), cells(new T[r.size()](), std::default_delete<T[]>()) {}
    buffer(T *host_data, range<N> r) : buffer(r) {
        for (std::size_t i = 0; i < r.size(); i++)
            cells.get()[i] = host_data[i];
    }

    range<N> get_range() const { return extent; }
    std::size_t size() const { return extent.size(); }
    std::size_t byte_size() const { return extent.size() * sizeof(T); }
    T *data() const { return cells.get(); }

    friend bool operator==(buffer const &a, buffer const &b) { return a.cells == b.cells; }

  private:
    range<N> extent;
    std::shared_ptr<T> cells;
};

namespace detail {
template <typename Ref, int Rem> class RowProxy;
template <typename Ref> class RowProxy<Ref, 1> {
  public:
    RowProxy(std::remove_reference_t<Ref> *base, std::size_t const *strides)
        : base(base), strides(strides) {}
    Ref operator[](std::size_t i) const { return base[i * strides[0]]; }

  private:
    std::remove_reference_t<Ref> *base;
    std::size_t const *strides;
};
template <typename Ref, int Rem> class RowProxy {
  public:
    RowProxy(std::remove_reference_t<Ref> *base, std::size_t const *strides)
        : base(base), strides(strides) {}
    RowProxy<Ref, Rem - 1> operator[](std::size_t i) const {
        return RowProxy<Ref, Rem - 1>(base + i * strides[0], strides + 1);
    }

  private:
    std::remove_reference_t<Ref> *base;
    std::size_t const *strides;
};
} // namespace detail

// Direct view of a buffer's host memory.
template <typename T, int N = 1, access::mode M = access::mode::read_write> class host_accessor {
    static constexpr bool is_read_only = (M == access::mode::read);

  public:
    using value_type = std::conditional_t<is_read_only, const T, T>;
    using reference = value_type &;

    host_accessor(buffer<T, N> &b) : extent(b.get_range()), base(b.data()) { init(); }
    host_accessor(buffer<T, N> &b, mode_tag_t<M>) : host_accessor(b) {}

    range<N> get_range() const { return extent; }
    std::size_t size() const { return extent.size(); }
    std::size_t byte_size() const { return extent.size() * sizeof(T); }
    value_type *get_pointer() const { return base; }

    reference operator[](id<N> i) const {
        std::size_t off = 0;
        for (int d = 0; d < N; d++)
            off += i[d] * strides[d];
        return base[off];
    }
    decltype(auto) operator[](std::size_t i) const
        requires(N > 1)
    {
        return detail::RowProxy<reference, N - 1>(base + i * strides[0], strides + 1);
    }
    reference operator[](std::size_t i) const
        requires(N == 1)
    {
        return base[i];
    }

  protected:
    host_accessor(range<N> extent, T *base) : extent(extent), base(base) { init(); }

  private:
    void init() {
        std::size_t s = 1;
        for (int d = N - 1; d >= 0; d--) {
            strides[d] = s;
            s *= extent[d];
        }
    }
    range<N> extent;
    value_type *base;
    std::size_t strides[N];
};

template <typename T, int N> host_accessor(buffer<T, N> &) -> host_accessor<T, N>;
template <typename T, int N, access::mode M>
host_accessor(buffer<T, N> &, mode_tag_t<M>) -> host_accessor<T, N, M>;

// Math names the examples use through the sycl namespace (examples/fdtd/src/Kernel.hpp:83,
// material/Material.hpp:41,51).  Host libm, float overloads included.
using std::cos;
using std::exp;
using std::isinf;
using std::sin;
using std::sqrt;

} // namespace sycl
