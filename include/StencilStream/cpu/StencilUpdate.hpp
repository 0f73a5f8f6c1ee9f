// stencil::cpu::StencilUpdate -- the explicitly selected host backend (OpenMP over rows).
//
// Behaviour parity with StencilStream/cpu/StencilUpdate.hpp:40-229: Params in the same aggregate
// order (:51-92); operator() double-buffers (the input grid is only read, pass 0 writes the first
// scratch grid, later passes ping-pong; :109-142), evaluates the time-dependent value once per
// iteration on the host (:197) and replaces every out-of-grid neighbour by Params::halo_value in
// every sweep (:202-216); counters accumulate over calls and sub-iterations are not counted as
// cell updates (:135-139).  This backend is never used by stencil::hip as a fallback.
#pragma once
#include "../Concepts.hpp"
#include "../Stencil.hpp"
#include "Grid.hpp"

#include <chrono>
#include <utility>

namespace stencil {
namespace cpu {

template <concepts::TransitionFunction F> class StencilUpdate {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    static constexpr std::size_t radius = F::stencil_radius;

  public:
    using GridImpl = Grid<Cell>;

    struct Params {
        F transition_function;
        Cell halo_value = Cell();
        std::size_t iteration_offset = 0;
        std::size_t n_iterations = 1;
        sycl::device device = sycl::device();
        bool blocking = false;
    };

    StencilUpdate(Params params) : params(params), n_processed_cells(0), walltime(0.0) {}

    Params &get_params() { return params; }
    std::size_t get_n_processed_cells() const { return n_processed_cells; }
    double get_walltime() const { return walltime; }

    GridImpl operator()(GridImpl &source_grid) {
        GridImpl scratch[2] = {source_grid.make_similar(), source_grid.make_similar()};
        GridImpl *read_from = &source_grid;
        std::size_t write_slot = 0;

        auto started = std::chrono::high_resolution_clock::now();
        for (std::size_t i = 0; i < params.n_iterations; i++) {
            const std::size_t iteration = params.iteration_offset + i;
            const TDV tdv = params.transition_function.get_time_dependent_value(iteration);
            for (std::size_t sub = 0; sub < F::n_subiterations; sub++) {
                sweep(*read_from, scratch[write_slot], iteration, sub, tdv);
                read_from = &scratch[write_slot];
                write_slot ^= 1;
            }
        }
        std::chrono::duration<double> elapsed =
            std::chrono::high_resolution_clock::now() - started;
        walltime += elapsed.count();
        n_processed_cells +=
            params.n_iterations * source_grid.get_grid_height() * source_grid.get_grid_width();
        return *read_from;
    }

  private:
    void sweep(GridImpl &source, GridImpl &target, std::size_t iteration, std::size_t subiteration,
               TDV const &tdv) {
        const std::size_t height = source.get_grid_height(), width = source.get_grid_width();
        const sycl::range<2> extent(height, width);
        Cell const *src = source.get_buffer().data();
        Cell *dst = target.get_buffer().data();
        F const &f = params.transition_function;
        Cell const halo = params.halo_value;

#pragma omp parallel for schedule(static)
        for (long long row = 0; row < (long long)height; row++) {
            for (std::size_t col = 0; col < width; col++) {
                Stencil<Cell, radius, TDV> st(sycl::id<2>(row, col), extent, iteration,
                                              subiteration, tdv);
                for (std::size_t dr = 0; dr <= 2 * radius; dr++) {
                    const std::size_t nr = std::size_t(row) + dr; // neighbour row + radius
                    const bool row_inside = nr >= radius && nr < height + radius;
                    for (std::size_t dc = 0; dc <= 2 * radius; dc++) {
                        const std::size_t nc = col + dc;
                        const bool inside = row_inside && nc >= radius && nc < width + radius;
                        st[sycl::id<2>(dr, dc)] =
                            inside ? src[(nr - radius) * width + (nc - radius)] : halo;
                    }
                }
                dst[std::size_t(row) * width + col] = f(st);
            }
        }
    }

    Params params;
    std::size_t n_processed_cells;
    double walltime;
};

} // namespace cpu
} // namespace stencil
