// Per-field plumbing for cells that opt into split storage with
//     static constexpr auto fields = std::make_tuple(&Cell::a, &Cell::b, ...);
// Interface parity: StencilStream/cuda/internal/Helpers.hpp:37-67 of the reference
// (alloc_field_buffers / FieldBuffers / for_each_in_two_tuples).  The reference builds a tuple of
// sycl::buffer<Field, 1>; here the planes are typed device pointers from the runtime's pool, owned by
// the returned object.
#pragma once
#include "Runtime.hpp"
#include "Sweep.hpp"

#include <tuple>
#include <utility>

namespace stencil {
namespace hip {
namespace internal {

// f(get<I>(a), get<I>(b)) for every I
template <typename TupleA, typename TupleB, typename Fn>
void for_each_in_two_tuples(TupleA &&a, TupleB &&b, Fn &&fn) {
    constexpr std::size_t n = std::tuple_size_v<std::decay_t<TupleA>>;
    static_assert(n == std::tuple_size_v<std::decay_t<TupleB>>, "Tuples must have same size");
    [&]<std::size_t... Is>(std::index_sequence<Is...>) {
        (fn(std::get<Is>(std::forward<TupleA>(a)), std::get<Is>(std::forward<TupleB>(b))), ...);
    }(std::make_index_sequence<n>{});
}

// One dense device plane per field of CellT, `n_cells` elements each.
template <typename CellT> class FieldBuffers {
    static_assert(SplittableCell<CellT>, "CellT::fields is missing");
    static constexpr int n = field_count<CellT>();

    template <std::size_t... Is> static auto pointer_tuple(std::index_sequence<Is...>) {
        return std::tuple<FieldType<CellT, int(Is)> *...>{};
    }

  public:
    using Pointers = decltype(pointer_tuple(std::make_index_sequence<std::size_t(n)>{}));

    explicit FieldBuffers(std::size_t n_cells) : n_cells(n_cells) {
        static_for<0, n>([&](auto f) {
            using E = FieldType<CellT, f>;
            std::get<f>(planes) = static_cast<E *>(device_alloc(n_cells * sizeof(E)));
        });
    }
    FieldBuffers(FieldBuffers const &) = delete;
    FieldBuffers &operator=(FieldBuffers const &) = delete;
    FieldBuffers(FieldBuffers &&other) noexcept : n_cells(other.n_cells), planes(other.planes) {
        other.planes = Pointers{};
    }
    ~FieldBuffers() {
        static_for<0, n>([&](auto f) { ststhip_free(std::get<f>(planes)); });
    }

    Pointers const &pointers() const { return planes; }
    std::size_t size() const { return n_cells; }

    // the same planes in the form the sweep kernels take
    PlaneSet<CellT, true> plane_set() const {
        PlaneSet<CellT, true> set;
        static_for<0, n>([&](auto f) { set.plane[f] = std::get<f>(planes); });
        return set;
    }

  private:
    std::size_t n_cells;
    Pointers planes{};
};

template <typename CellT> FieldBuffers<CellT> alloc_field_buffers(std::size_t n_cells) {
    return FieldBuffers<CellT>(n_cells);
}

} // namespace internal
} // namespace hip
} // namespace stencil
