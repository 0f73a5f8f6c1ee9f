// stencil::hip::BlockUpdate -- StencilUpdate for ONE BLOCK of a grid that is cut over a MESH of GPUs (mesh_rows x
// mesh_cols blocks, rank = mesh_row * mesh_cols + mesh_col), one process per GPU: the template-level face of the native
// block driver (ststhip_block_*, include/ststhip.h), as hip::StripUpdate is of the strip driver.  An EXTENSION of the
// reference's API; the role model is its tiled multi-pass update (StencilStream/tiling/StencilUpdate.hpp:216-247) over the
// tile-with-halo geometry of tiling/Grid.hpp:305-450 -- same Params, same transition function, the kernel instantiated in
// the user's translation unit.
//
//     stencil::hip::BlockUpdate<Kernel> block({.transition_function = k, .halo_value = h, .n_iterations = 1000},
//                                             total_rows, total_cols, rank, mesh_rows, mesh_cols, comm);
//     block.upload(my_cells);      // (end_row() - first_row()) x (end_col() - first_col()) cells, row-major, dense
//     block();                     // n_iterations generations of the WHOLE grid; ghost columns, then ghost rows (corners)
//     block.download(my_cells);
//
// `comm`: a communicator of the mesh's ranks wired for the mesh (ststhip_comm_create + ststhip_comm_set_mesh).  Ranks
// that RCCL cannot join (tests: several blocks on one GPU) pass comm = nullptr and two exchange callbacks with the
// contract of ststhip_comm_exchange_rows (ststhip.h, ststhip_block_create).
#pragma once
#include "StencilUpdate.hpp"

#include <cstring>
#include <memory>
#include <vector>

namespace stencil {
namespace hip {

template <concepts::TransitionFunction F, bool split_cell_structure = false,
          typename TDVStrategy = tdv::single_pass::PrecomputeOnHostStrategy>
class BlockUpdate {
    using Update = StencilUpdate<F, split_cell_structure, TDVStrategy>;

  public:
    using Cell = typename F::Cell;
    using Params = typename Update::Params;

    BlockUpdate(Params params, std::size_t total_rows, std::size_t total_cols, int rank, int mesh_rows, int mesh_cols,
                ststhip_comm comm, ststhip_exchange_fn exchange_rows = nullptr, void *exchange_rows_ctx = nullptr,
                ststhip_exchange_fn exchange_cols = nullptr, void *exchange_cols_ctx = nullptr)
        : update(std::make_unique<Update>(params)) {
        internal::ensure_runtime(params.device.hip_index());
        const ststhip_sweep_desc desc = Update::sweep_description_with_host_values();
        internal::check(ststhip_block_create_custom(Update::launch_entry(), update.get(), &desc, total_rows, total_cols, rank,
                                                    mesh_rows, mesh_cols, comm, exchange_rows, exchange_rows_ctx,
                                                    exchange_cols, exchange_cols_ctx, &block),
                        "ststhip_block_create_custom");
        std::uint64_t r0 = 0, r1 = 0, c0 = 0, c1 = 0;
        internal::check(ststhip_block_geometry(block, &r0, &r1, &c0, &c1), "ststhip_block_geometry");
        row_begin = r0, row_end = r1, col_begin = c0, col_end = c1;
    }
    BlockUpdate(BlockUpdate const &) = delete;
    BlockUpdate &operator=(BlockUpdate const &) = delete;
    ~BlockUpdate() {
        if (block)
            ststhip_strip_destroy(block);
    }

    Params &get_params() { return update->get_params(); }
    // global rows [first_row, end_row) x columns [first_col, end_col) are this block's
    std::size_t first_row() const { return row_begin; }
    std::size_t end_row() const { return row_end; }
    std::size_t first_col() const { return col_begin; }
    std::size_t end_col() const { return col_end; }
    std::size_t n_cells() const { return (row_end - row_begin) * (col_end - col_begin); }

    // the owned cells, row-major and dense, from / to host memory
    void upload(Cell const *owned_cells) { transfer(const_cast<Cell *>(owned_cells), true); }
    void download(Cell *owned_cells) { transfer(owned_cells, false); }

    // RCCL creates its point-to-point channels on first use: once, outside of anything that is timed
    void warm_up() { internal::check(ststhip_strip_warm_up(block), "ststhip_strip_warm_up"); }

    // n_iterations generations of the whole distributed grid, starting at iteration_offset
    void operator()() {
        Params const &p = update->get_params();
        internal::check(ststhip_strip_advance(block, p.iteration_offset, p.n_iterations, p.blocking ? 1 : 0),
                        "ststhip_strip_advance");
    }
    void synchronize() { internal::check(ststhip_strip_synchronize(block), "ststhip_strip_synchronize"); }

  private:
    void transfer(Cell *host_cells, bool to_device) {
        const std::size_t n = n_cells(), cols = col_end - col_begin;
        if (n == 0)
            return;
        if constexpr (!Update::sweeps_on_planes) {
            internal::check(to_device ? ststhip_block_upload(block, 0, host_cells, cols * sizeof(Cell))
                                      : ststhip_block_download(block, 0, host_cells, cols * sizeof(Cell)),
                            "block transfer");
        } else {
            // per-field planes: the fields are taken apart / put together on the host (a block's edge is small against
            // the sweeps between two transfers; the single-GPU path has the LDS-staged scatter / gather kernels)
            constexpr int n_planes = Update::n_planes;
            std::vector<unsigned char> plane;
            for (int f = 0; f < n_planes; f++) {
                const std::size_t size = Update::plane_elem_size(f), offset = Update::plane_elem_offset(f);
                plane.resize(n * size);
                unsigned char *cells = reinterpret_cast<unsigned char *>(host_cells);
                if (to_device) {
                    for (std::size_t i = 0; i < n; i++)
                        std::memcpy(plane.data() + i * size, cells + i * sizeof(Cell) + offset, size);
                    internal::check(ststhip_block_upload(block, unsigned(f), plane.data(), cols * size), "block upload");
                } else {
                    internal::check(ststhip_block_download(block, unsigned(f), plane.data(), cols * size), "block download");
                    for (std::size_t i = 0; i < n; i++)
                        std::memcpy(cells + i * sizeof(Cell) + offset, plane.data() + i * size, size);
                }
            }
        }
    }

    std::unique_ptr<Update> update; // the launch callback's context: must not move
    ststhip_strip block = nullptr;
    std::size_t row_begin = 0, row_end = 0, col_begin = 0, col_end = 0;
};

} // namespace hip
} // namespace stencil
