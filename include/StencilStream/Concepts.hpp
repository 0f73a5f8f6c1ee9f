// The compile-time contract between applications and backends.
//
// Interface parity with StencilStream/Concepts.hpp:61-172 of the reference: the four concepts
// TransitionFunction, GridAccessor, Grid and StencilUpdate accept and reject the same types; they
// are written here as compositions of smaller named requirements.
#pragma once
#include "Stencil.hpp"

#include <concepts>
#include <sycl.hpp>
#include <type_traits>

namespace stencil {
namespace concepts {

namespace detail {
template <typename T>
concept HasStencilGeometry = std::same_as<decltype(T::stencil_radius), const std::size_t> &&
                             std::same_as<decltype(T::n_subiterations), const std::size_t> &&
                             (T::stencil_radius >= 1) && (T::n_subiterations >= 1);

template <typename T>
using StencilOf = Stencil<typename T::Cell, T::stencil_radius, typename T::TimeDependentValue>;

template <typename P, typename TF>
concept UpdateParams = std::is_class_v<P> && requires(P p) {
    { p.transition_function } -> std::same_as<TF &>;
    { p.halo_value } -> std::same_as<typename TF::Cell &>;
    { p.iteration_offset } -> std::same_as<std::size_t &>;
    { p.n_iterations } -> std::same_as<std::size_t &>;
};
} // namespace detail

// A callable cell -> cell rule with a radius, a sub-iteration count and a per-iteration value.
template <typename T>
concept TransitionFunction =
    std::semiregular<typename T::Cell> && std::copyable<typename T::TimeDependentValue> &&
    detail::HasStencilGeometry<T> &&
    requires(T const &f, detail::StencilOf<T> const &neighbourhood, std::size_t iteration) {
        { f(neighbourhood) } -> std::same_as<typename T::Cell>;
        { f.get_time_dependent_value(iteration) } -> std::same_as<typename T::TimeDependentValue>;
    };

// Host-side view of a grid: ac[id<2>] and ac[r][c] both give a cell reference.
template <typename Accessor, typename Cell>
concept GridAccessor = requires(Accessor ac, std::size_t r, std::size_t c) {
    { ac[sycl::id<2>(r, c)] } -> std::same_as<Cell &>;
    { ac[r][c] } -> std::same_as<Cell &>;
};

// A two-dimensional container of cells a backend can update.
template <typename G, typename Cell>
concept Grid = requires(G &grid, sycl::buffer<Cell, 2> buffer, std::size_t r, std::size_t c) {
    { G(r, c) } -> std::same_as<G>;
    { G(sycl::range<2>(r, c)) } -> std::same_as<G>;
    { G(buffer) } -> std::same_as<G>;
    { grid.copy_from_buffer(buffer) } -> std::same_as<void>;
    { grid.copy_to_buffer(buffer) } -> std::same_as<void>;
    { grid.get_grid_height() } -> std::convertible_to<std::size_t>;
    { grid.get_grid_width() } -> std::convertible_to<std::size_t>;
    { grid.get_grid_range() } -> std::convertible_to<sycl::range<2>>;
    { grid.make_similar() } -> std::same_as<G>;
    { typename G::template GridAccessor<sycl::access::mode::read_write>(grid) } -> GridAccessor<Cell>;
};

// The object that advances a grid by Params::n_iterations generations.
template <typename SU, typename TF, typename G>
concept StencilUpdate = TransitionFunction<TF> && Grid<G, typename TF::Cell> &&
                        detail::UpdateParams<typename SU::Params, TF> &&
                        requires(SU update, G &grid, typename SU::Params params) {
                            { SU(params) } -> std::same_as<SU>;
                            { update.get_params() } -> std::same_as<typename SU::Params &>;
                            { update(grid) } -> std::same_as<G>;
                        };

} // namespace concepts
} // namespace stencil
