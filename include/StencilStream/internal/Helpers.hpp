// Small helpers shared by backends and applications.
// Interface parity: StencilStream/internal/Helpers.hpp:23-29,46-48 of the reference
// (int_ceil_div and the kernel-naming macros; the FPGA pipe plumbing of that file has no
// counterpart on MI355X).
#pragma once
#include <sycl/detail_hd.hpp>

#include <bit>
#include <cstddef>
#include <cstdint>
#include <type_traits>

// The reference names its SYCL kernels through these macros.  HIP kernels are named by their
// template instantiation, so the macros only keep source compatibility.
#define STENCILSTREAM_NAMED_SINGLE_TASK(Name, argument) single_task(argument)
#define STENCILSTREAM_NAMED_PARALLEL_FOR(Name, range, kernel) parallel_for(range, kernel)

namespace stencil {
namespace internal {

// ceil(a / b) for unsigned or positive operands
template <typename T> STST_HD inline constexpr T int_ceil_div(T a, T b) {
    return (a % b == 0) ? a / b : a / b + 1;
}

template <typename T> STST_HD inline constexpr T round_up(T a, T b) {
    return int_ceil_div(a, b) * b;
}

// compile-time loop: body(std::integral_constant<int, i>) for i in [0, N)
template <int Begin, int End, typename Body> STST_HD inline constexpr void static_for(Body &&body) {
    if constexpr (Begin < End) {
        body(std::integral_constant<int, Begin>{});
        static_for<Begin + 1, End>(body);
    }
}

} // namespace internal
} // namespace stencil
