// sycl::range<N> -- plain value type (not a SYCL runtime).  Part of the host
// vocabulary that StencilStream user code names; see INTEGRATION.md.
#pragma once
#include "detail_hd.hpp"
#include <cstddef>

namespace sycl {

namespace detail {
// N extents/indices with elementwise comparison; shared by id<N> and range<N>.
template <int N> struct IndexArray {
    static_assert(N >= 1 && N <= 3);
    std::size_t v[N];

    STST_HD constexpr IndexArray() : v{} {}
    STST_HD constexpr IndexArray(std::size_t a)
        requires(N == 1)
        : v{a} {}
    STST_HD constexpr IndexArray(std::size_t a, std::size_t b)
        requires(N == 2)
        : v{a, b} {}
    STST_HD constexpr IndexArray(std::size_t a, std::size_t b, std::size_t c)
        requires(N == 3)
        : v{a, b, c} {}

    STST_HD constexpr std::size_t &operator[](int i) { return v[i]; }
    STST_HD constexpr std::size_t const &operator[](int i) const { return v[i]; }
    STST_HD constexpr std::size_t get(int i) const { return v[i]; }

    STST_HD constexpr bool same_as(IndexArray const &o) const {
        bool eq = true;
        for (int i = 0; i < N; i++)
            eq = eq && (v[i] == o.v[i]);
        return eq;
    }
};
} // namespace detail

template <int N = 1> class range : public detail::IndexArray<N> {
  public:
    using detail::IndexArray<N>::IndexArray;
    static constexpr int dimensions = N;

    STST_HD constexpr std::size_t size() const {
        std::size_t s = 1;
        for (int i = 0; i < N; i++)
            s *= this->v[i];
        return s;
    }
    STST_HD friend constexpr bool operator==(range const &a, range const &b) {
        return a.same_as(b);
    }
    STST_HD friend constexpr bool operator!=(range const &a, range const &b) {
        return !a.same_as(b);
    }
};

range(std::size_t) -> range<1>;
range(std::size_t, std::size_t) -> range<2>;
range(std::size_t, std::size_t, std::size_t) -> range<3>;

} // namespace sycl
