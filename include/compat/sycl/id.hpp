// sycl::id<N> -- plain value type (not a SYCL runtime).
#pragma once
#include "range.hpp"

namespace sycl {

template <int N = 1> class id : public detail::IndexArray<N> {
  public:
    using detail::IndexArray<N>::IndexArray;
    static constexpr int dimensions = N;

    STST_HD constexpr id() = default;
    STST_HD constexpr id(range<N> const &r) {
        for (int i = 0; i < N; i++)
            this->v[i] = r[i];
    }
    // a 1-D id converts to its index, as in SYCL
    STST_HD constexpr operator std::size_t() const
        requires(N == 1)
    {
        return this->v[0];
    }
    STST_HD friend constexpr bool operator==(id const &a, id const &b) { return a.same_as(b); }
    STST_HD friend constexpr bool operator!=(id const &a, id const &b) { return !a.same_as(b); }
};

id(std::size_t) -> id<1>;
id(std::size_t, std::size_t) -> id<2>;
id(std::size_t, std::size_t, std::size_t) -> id<3>;

} // namespace sycl
