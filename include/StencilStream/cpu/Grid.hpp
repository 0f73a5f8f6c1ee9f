// stencil::cpu::Grid -- host-resident AoS grid for the explicitly selected CPU backend.
//
// Interface parity: StencilStream/cpu/Grid.hpp:50-188 (constructors from (rows, cols), range and
// buffer; copy = shared handle; copy_from/to_buffer throwing std::range_error on a size mismatch;
// GridAccessor<mode>; getters; make_similar; get_buffer).
#pragma once
#include <sycl.hpp>

#include <cstring>
#include <stdexcept>

namespace stencil {
namespace cpu {

template <typename Cell> class Grid {
  public:
    static constexpr std::size_t dimensions = 2;

    Grid(std::size_t n_rows, std::size_t n_columns) : storage(sycl::range<2>(n_rows, n_columns)) {}
    Grid(sycl::range<2> extent) : storage(extent) {}
    Grid(sycl::buffer<Cell, 2> source) : storage(source.get_range()) { copy_from_buffer(source); }
    // Copies are handles onto the same cells.
    Grid(Grid const &) = default;
    Grid &operator=(Grid const &) = default;

    void copy_from_buffer(sycl::buffer<Cell, 2> source) {
        require_same_extent(source);
        std::memcpy(static_cast<void *>(storage.data()), source.data(), storage.byte_size());
    }

    void copy_to_buffer(sycl::buffer<Cell, 2> target) {
        require_same_extent(target);
        std::memcpy(static_cast<void *>(target.data()), storage.data(), storage.byte_size());
    }

    template <sycl::access::mode access_mode = sycl::access::mode::read_write>
    class GridAccessor : public sycl::host_accessor<Cell, 2, access_mode> {
      public:
        GridAccessor(Grid &grid) : sycl::host_accessor<Cell, 2, access_mode>(grid.storage) {}
    };

    std::size_t get_grid_height() const { return storage.get_range()[0]; }
    std::size_t get_grid_width() const { return storage.get_range()[1]; }
    sycl::range<2> get_grid_range() const { return storage.get_range(); }
    Grid make_similar() const { return Grid(storage.get_range()); }
    sycl::buffer<Cell, 2> &get_buffer() { return storage; }

  private:
    void require_same_extent(sycl::buffer<Cell, 2> const &other) const {
        if (other.get_range() != storage.get_range())
            throw std::range_error("The target buffer has not the same size as the grid");
    }
    sycl::buffer<Cell, 2> storage;
};

} // namespace cpu
} // namespace stencil
