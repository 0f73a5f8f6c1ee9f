// Stencil<Cell, radius, TDV>: the (2r+1)x(2r+1) neighbourhood a transition function receives.
//
// Interface parity with StencilStream/Stencil.hpp:45-181 of the reference: same template
// parameters, `diameter`, both constructors, signed two-level indexing st[dr][dc] with the origin
// at the centre (dr = -1 is north, dc = -1 is west), unsigned indexing st[sycl::id<2>] with the
// origin at the north-west corner, and the public const members id / iteration / subiteration /
// grid_range / time_dependent_value.  Every member is callable from HIP device code because the
// sweep kernels of the MI355X backend build one Stencil per cell update in registers.
#pragma once
#include "internal/Helpers.hpp"

#include <concepts>
#include <limits>
#include <sycl/id.hpp>
#include <sycl/range.hpp>
#include <type_traits>
#include <variant>

namespace stencil {

template <typename Cell, std::size_t stencil_radius, typename TimeDependentValue = std::monostate>
    requires std::semiregular<Cell> && (stencil_radius >= 1)
class Stencil {
  public:
    static constexpr std::size_t diameter = 2 * stencil_radius + 1;
    static_assert(diameter <= std::size_t(std::numeric_limits<int>::max()));

    // Neighbourhood left uninitialised; the backend fills it cell by cell.
    STST_HD Stencil(sycl::id<2> id, sycl::range<2> grid_range, std::size_t iteration,
                    std::size_t subiteration, TimeDependentValue tdv)
        : id(id), iteration(iteration), subiteration(subiteration), grid_range(grid_range),
          time_dependent_value(tdv), cells() {}

    // Neighbourhood copied from a diameter x diameter array (row-major, NW origin).
    STST_HD Stencil(sycl::id<2> id, sycl::range<2> grid_range, std::size_t iteration,
                    std::size_t subiteration, TimeDependentValue tdv,
                    Cell raw[diameter][diameter])
        : id(id), iteration(iteration), subiteration(subiteration), grid_range(grid_range),
          time_dependent_value(tdv), cells() {
        for (std::size_t i = 0; i < diameter * diameter; i++)
            cells[i / diameter][i % diameter] = raw[i / diameter][i % diameter];
    }

    // Result of st[dr]; a second [dc] yields the cell.
    template <std::signed_integral index_t>
        requires(stencil_radius <= std::size_t(std::numeric_limits<index_t>::max()))
    class StencilSubscript {
      public:
        STST_HD StencilSubscript(Stencil const &owner, index_t row_offset)
            : owner(owner), row_offset(row_offset) {}

        STST_HD Cell const &operator[](index_t column_offset) const {
            return owner.cells[row_offset + index_t(stencil_radius)]
                              [column_offset + index_t(stencil_radius)];
        }

      private:
        Stencil const &owner;
        index_t row_offset;
    };

    template <std::signed_integral index_t>
    STST_HD StencilSubscript<index_t> operator[](index_t row_offset) const
        requires(stencil_radius <= std::size_t(std::numeric_limits<index_t>::max()))
    {
        return StencilSubscript<index_t>(*this, row_offset);
    }

    STST_HD Cell const &operator[](sycl::id<2> nw_index) const {
        return cells[nw_index[0]][nw_index[1]];
    }
    STST_HD Cell &operator[](sycl::id<2> nw_index) { return cells[nw_index[0]][nw_index[1]]; }

    const sycl::id<2> id;                          // position of the centre cell in the grid
    const std::size_t iteration;                   // generation the neighbourhood belongs to
    const std::size_t subiteration;                // sub-iteration within that generation
    const sycl::range<2> grid_range;               // (height, width) of the grid
    const TimeDependentValue time_dependent_value; // per-iteration value computed on the host

  private:
    Cell cells[diameter][diameter];
};

} // namespace stencil
