// Device-side reductions over a stencil::hip::Grid -- an EXTENSION of the reference's API (it has none): an
// application that checks convergence between StencilUpdate calls (examples/convection/convection.cpp:412-438 scans
// the whole grid through a host accessor every `nerr` iterations) asks the device for the few numbers it needs
// instead of downloading every cell.
//
//     auto m = stencil::hip::max_abs(grid, {stencil::hip::over(&Cell::ErrV, nx, ny + 1),
//                                           stencil::hip::over(&Cell::Vx, nx + 1, ny)});
//     // m[i] = max |field i| over rows < row_limit, columns < col_limit; -infinity if that range is empty
//
// One pass over the AoS cells in HBM (ststhip_reduce_max_abs: wave DPP reduce + one atomic per wave and field).
#pragma once
#include "Grid.hpp"

#include <cstddef>
#include <initializer_list>
#include <type_traits>
#include <vector>

namespace stencil {
namespace hip {

// One field of the cell and the sub-rectangle [0, row_limit) x [0, col_limit) it is reduced over.
struct ReduceField {
    ststhip_reduce_field raw;
};

template <typename Cell, typename Real>
ReduceField over(Real Cell::*member, std::size_t row_limit, std::size_t col_limit) {
    static_assert(std::is_same_v<Real, float> || std::is_same_v<Real, double>,
                  "max_abs reduces float and double fields");
    Cell probe{};
    ReduceField f;
    f.raw.offset = std::uint32_t(reinterpret_cast<const char *>(&(probe.*member)) -
                                 reinterpret_cast<const char *>(&probe));
    f.raw.type = std::is_same_v<Real, double> ? STSTHIP_F64 : STSTHIP_F32;
    f.raw.row_limit = row_limit;
    f.raw.col_limit = col_limit;
    return f;
}

// Maximum of |field| per entry of `fields` (at most 8), computed on the device; blocks until it is known.
template <typename Cell> std::vector<double> max_abs(Grid<Cell> &grid, std::initializer_list<ReduceField> fields) {
    std::vector<ststhip_reduce_field> raw;
    for (ReduceField const &f : fields)
        raw.push_back(f.raw);
    std::vector<double> result(raw.size());
    if (raw.empty())
        return result;
    internal::ensure_runtime(-1);
    internal::check(ststhip_reduce_max_abs(grid.device_cells(), sizeof(Cell), grid.get_grid_height(),
                                           grid.get_grid_width(), grid.get_grid_width(), int(raw.size()),
                                           raw.data(), result.data(), internal::default_stream()),
                    "ststhip_reduce_max_abs");
    return result;
}

} // namespace hip
} // namespace stencil
