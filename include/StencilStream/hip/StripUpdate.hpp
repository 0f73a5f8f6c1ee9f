// stencil::hip::StripUpdate -- StencilUpdate for ONE ROW STRIP of a grid that is cut over several GPUs, one process per
// GPU: the template-level face of the native strip driver (ststhip_strip_*, include/ststhip.h).  An EXTENSION of the
// reference's API: its GPU backends know one device; its multi-device mode (monotile, FPGA) sits behind the same
// StencilUpdate interface (StencilStream/monotile/StencilUpdate.hpp:166-227), which is the role model here --
// same Params, same transition function, the kernel instantiated in the user's translation unit.
//
//     stencil::hip::StripUpdate<Kernel> strip({.transition_function = k, .halo_value = h, .n_iterations = 1000},
//                                             total_rows, width, rank, n_ranks, comm);   // comm: ststhip_comm_create
//     strip.upload(my_rows);                       // (end_row() - first_row()) x width cells, row-major
//     strip();                                     // n_iterations generations of the WHOLE grid; ghost rows over RCCL
//     strip.get_params().iteration_offset += 1000; // as with StencilUpdate, the caller moves the offset
//     strip.download(my_rows);
//
// Every rank must make the same calls in the same order (the ghost-row exchange is a grouped send / receive with both
// neighbours once per launch).  Ranks that RCCL cannot join (tests: several strips on one GPU) pass comm = nullptr and
// an exchange callback with the contract of ststhip_comm_exchange_rows.
#pragma once
#include "StencilUpdate.hpp"

#include <memory>
#include <vector>

namespace stencil {
namespace hip {

template <concepts::TransitionFunction F, bool split_cell_structure = false,
          typename TDVStrategy = tdv::single_pass::PrecomputeOnHostStrategy>
class StripUpdate {
    using Update = StencilUpdate<F, split_cell_structure, TDVStrategy>;

  public:
    using Cell = typename F::Cell;
    using Params = typename Update::Params;

    StripUpdate(Params params, std::size_t total_rows, std::size_t width, int rank, int n_ranks, ststhip_comm comm,
                ststhip_exchange_fn exchange = nullptr, void *exchange_ctx = nullptr)
        : update(std::make_unique<Update>(params)), width(width) {
        internal::ensure_runtime(params.device.hip_index());
        const ststhip_sweep_desc desc = Update::sweep_description_with_host_values();
        internal::check(ststhip_strip_create_custom(Update::launch_entry(), update.get(), &desc, total_rows, width, rank,
                                                    n_ranks, comm, exchange, exchange_ctx, &strip),
                        "ststhip_strip_create_custom");
        std::uint64_t a = 0, b = 0;
        internal::check(ststhip_strip_rows(strip, &a, &b), "ststhip_strip_rows");
        row_begin = a;
        row_end = b;
    }
    StripUpdate(StripUpdate const &) = delete;
    StripUpdate &operator=(StripUpdate const &) = delete;
    ~StripUpdate() {
        if (strip)
            ststhip_strip_destroy(strip);
    }

    Params &get_params() { return update->get_params(); }
    std::size_t first_row() const { return row_begin; } // global rows [first_row, end_row) are this strip's
    std::size_t end_row() const { return row_end; }
    std::size_t n_cells() const { return (row_end - row_begin) * width; }

    // the owned rows, row-major AoS cells, from / to host memory
    void upload(Cell const *owned_rows) { transfer(const_cast<Cell *>(owned_rows), true); }
    void download(Cell *owned_rows) { transfer(owned_rows, false); }

    // RCCL creates its point-to-point channels on first use: once, outside of anything that is timed
    void warm_up() { internal::check(ststhip_strip_warm_up(strip), "ststhip_strip_warm_up"); }

    // n_iterations generations of the whole distributed grid, starting at iteration_offset
    void operator()() {
        Params const &p = update->get_params();
        internal::check(ststhip_strip_advance(strip, p.iteration_offset, p.n_iterations, p.blocking ? 1 : 0),
                        "ststhip_strip_advance");
    }
    void synchronize() { internal::check(ststhip_strip_synchronize(strip), "ststhip_strip_synchronize"); }

  private:
    void transfer(Cell *host_rows, bool to_device) {
        ststhip_stream stream = nullptr;
        internal::check(ststhip_strip_stream(strip, &stream), "ststhip_strip_stream");
        const std::size_t n = n_cells();
        if (n == 0)
            return;
        if constexpr (!Update::sweeps_on_planes) {
            void *rows = nullptr;
            internal::check(ststhip_strip_plane(strip, 0, &rows, nullptr), "ststhip_strip_plane");
            internal::check(to_device ? ststhip_memcpy_h2d(rows, host_rows, n * sizeof(Cell), stream)
                                      : ststhip_memcpy_d2h(host_rows, rows, n * sizeof(Cell), stream),
                            "strip transfer");
        } else {
            // per-field planes: through an AoS staging buffer in HBM and the LDS-staged scatter / gather kernels
            constexpr int n_planes = Update::n_planes;
            void *plane[n_planes];
            std::size_t offsets[n_planes], sizes[n_planes];
            for (int f = 0; f < n_planes; f++) {
                internal::check(ststhip_strip_plane(strip, unsigned(f), &plane[f], nullptr), "ststhip_strip_plane");
                offsets[f] = Update::plane_elem_offset(f);
                sizes[f] = Update::plane_elem_size(f);
            }
            void *staging = internal::device_alloc_on(n * sizeof(Cell), stream);
            if (to_device) {
                internal::check(ststhip_memcpy_h2d(staging, host_rows, n * sizeof(Cell), stream), "strip upload");
                internal::check(ststhip_scatter_fields(staging, sizeof(Cell), n, n_planes, offsets, sizes, plane, stream),
                                "scatter");
            } else {
                internal::check(ststhip_gather_fields(staging, sizeof(Cell), n, n_planes, offsets, sizes,
                                                      const_cast<const void *const *>(plane), stream),
                                "gather");
                internal::check(ststhip_memcpy_d2h(host_rows, staging, n * sizeof(Cell), stream), "strip download");
            }
            ststhip_free_async(staging, stream);
        }
        internal::check(ststhip_stream_synchronize(stream), "strip transfer");
    }

    std::unique_ptr<Update> update; // the launch callback's context: must not move
    ststhip_strip strip = nullptr;
    std::size_t width, row_begin = 0, row_end = 0;
};

} // namespace hip
} // namespace stencil
