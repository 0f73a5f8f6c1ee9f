// stencil::hip::StencilUpdate -- advances a grid on one MI355X.
//
// Interface parity with StencilStream/cuda/StencilUpdate.hpp:41-198: template parameters
// <TransitionFunction F, bool split_cell_structure = false>; `GridImpl`; `Params` with the fields
// transition_function, halo_value, iteration_offset, n_iterations, device, blocking, profiling in
// this order (:54-105); StencilUpdate(Params); GridImpl operator()(GridImpl&); get_params();
// get_n_processed_cells(); get_walltime(); get_kernel_runtime().
//
// Behaviour parity: the source grid is only read, the result is a new grid
// handle (n_iterations == 0 returns a handle onto the source, cuda/StencilUpdate.hpp:206,275);
// the walltime covers scratch allocation, scatter, sweeps, gather and -- if `blocking` -- the final
// synchronisation (:129-139); n_processed_cells does not count sub-iterations (:140-141);
// split_cell_structure = true scatters the AoS cells into per-field planes, sweeps on those and
// gathers back (:294-438).
//
// What differs by design: instead of one kernel per (iteration, sub-iteration), one kernel
// advances up to SweepTuning<F>::max_generations generations (hip/internal/Sweep.hpp).
#pragma once
#include "../Concepts.hpp"
#include "Grid.hpp"
#include "internal/Sweep.hpp"

#include <chrono>
#include <utility>
#include <vector>

namespace stencil {
namespace hip {

template <concepts::TransitionFunction F, bool split_cell_structure = false> class StencilUpdate {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    using Planes = internal::PlaneSet<Cell, split_cell_structure>;

  public:
    using GridImpl = Grid<Cell>;

    struct Params {
        F transition_function;
        Cell halo_value = Cell();
        std::size_t iteration_offset = 0;
        std::size_t n_iterations = 1;
        sycl::device device = sycl::device();
        bool blocking = false;
        bool profiling = false;
    };

    StencilUpdate(Params params)
        : params(params), n_processed_cells(0), walltime(0.0), kernel_runtime(0.0), timed() {}

    StencilUpdate(StencilUpdate const &other)
        : params(other.params), n_processed_cells(other.n_processed_cells),
          walltime(other.walltime), kernel_runtime(other.get_kernel_runtime()), timed() {}

    ~StencilUpdate() { drop_events(); }

    Params &get_params() { return params; }
    std::size_t get_n_processed_cells() const { return n_processed_cells; }
    double get_walltime() const { return walltime; }

    // Sum of the sweep kernels' device time in seconds; needs Params::profiling.
    double get_kernel_runtime() const {
        double seconds = kernel_runtime;
        for (auto const &pair : timed) {
            float ms = 0.0f;
            ststhip_event_synchronize(pair.second);
            if (ststhip_event_elapsed_ms(pair.first, pair.second, &ms) == STSTHIP_OK)
                seconds += double(ms) * 1e-3;
        }
        return seconds;
    }

    GridImpl operator()(GridImpl &source_grid) {
        internal::ensure_runtime(params.device.hip_index());
        ststhip_stream stream = internal::default_stream();
        fold_events();

        auto started = std::chrono::high_resolution_clock::now();
        GridImpl result = simulate(source_grid, stream);
        if (params.blocking)
            internal::check(ststhip_stream_synchronize(stream), "stream synchronize");
        std::chrono::duration<double> elapsed =
            std::chrono::high_resolution_clock::now() - started;

        walltime += elapsed.count();
        n_processed_cells +=
            params.n_iterations * source_grid.get_grid_height() * source_grid.get_grid_width();
        return result;
    }

  private:
    // All passes of one call: `remaining` generations in chunks of the compiled blocking depths.
    template <typename NextTarget>
    void run_passes(ststhip_domain const &dom, Planes first_source, NextTarget &&next_target,
                    ststhip_stream stream) {
        Planes source = first_source;
        std::uint64_t iteration = params.iteration_offset;
        std::uint64_t remaining = params.n_iterations;
        std::vector<TDV> tdv;
        while (remaining > 0) {
            const int depth = internal::next_pass_depth<F, split_cell_structure>(remaining);
            tdv.clear();
            for (int t = 0; t < depth; t++) // host side, once per generation
                tdv.push_back(params.transition_function.get_time_dependent_value(iteration + t));
            Planes target = next_target();

            ststhip_event ev_start = nullptr, ev_stop = nullptr;
            if (params.profiling) {
                internal::check(ststhip_event_create(&ev_start), "event create");
                internal::check(ststhip_event_create(&ev_stop), "event create");
                ststhip_event_record(ev_start, stream);
            }
            internal::dispatch_sweep<F, split_cell_structure>(
                depth, params.transition_function, params.halo_value, tdv.data(), dom, source,
                target, 0, dom.global_height, iteration, stream);
            if (params.profiling) {
                ststhip_event_record(ev_stop, stream);
                timed.emplace_back(ev_start, ev_stop);
            }
            source = target;
            iteration += depth;
            remaining -= depth;
        }
    }

    GridImpl simulate(GridImpl &source_grid, ststhip_stream stream)
        requires(!split_cell_structure)
    {
        if (params.n_iterations == 0)
            return source_grid;
        ststhip_domain dom = domain_of(source_grid);
        GridImpl scratch[2] = {source_grid.make_similar(), source_grid.make_similar()};
        Planes first;
        first.plane[0] = const_cast<Cell *>(source_grid.device_cells());
        int slot = 1, last_written = -1;
        run_passes(
            dom, first,
            [&]() {
                slot ^= 1;
                last_written = slot;
                Planes target;
                target.plane[0] = scratch[slot].device_cells_for_overwrite();
                return target;
            },
            stream);
        return scratch[last_written];
    }

    GridImpl simulate(GridImpl &source_grid, ststhip_stream stream)
        requires(split_cell_structure)
    {
        ststhip_domain dom = domain_of(source_grid);
        const std::size_t n_cells = source_grid.get_grid_height() * source_grid.get_grid_width();
        constexpr int n = Planes::n_planes;
        std::size_t offsets[n], sizes[n];
        for (int f = 0; f < n; f++) {
            offsets[f] = Planes::elem_offset(f);
            sizes[f] = Planes::elem_size(f);
        }
        Planes sets[2];
        for (auto &set : sets)
            for (int f = 0; f < n; f++)
                set.plane[f] = internal::device_alloc(n_cells * sizes[f]);

        internal::check(ststhip_scatter_fields(source_grid.device_cells(), sizeof(Cell), n_cells, n,
                                               offsets, sizes, sets[0].plane, stream),
                        "scatter");
        int slot = 0;
        run_passes(
            dom, sets[0],
            [&]() {
                slot ^= 1;
                return sets[slot];
            },
            stream);
        GridImpl result = source_grid.make_similar();
        internal::check(ststhip_gather_fields(result.device_cells_for_overwrite(), sizeof(Cell),
                                              n_cells, n, offsets, sizes,
                                              const_cast<const void *const *>(sets[slot].plane),
                                              stream),
                        "gather");
        // pool blocks are recycled in stream order, so they can be returned while work is queued
        for (auto &set : sets)
            for (int f = 0; f < n; f++)
                ststhip_free(set.plane[f]);
        return result;
    }

    static ststhip_domain domain_of(GridImpl const &grid) {
        ststhip_domain dom;
        dom.global_height = grid.get_grid_height();
        dom.global_width = grid.get_grid_width();
        dom.row_origin = 0;
        dom.local_rows = grid.get_grid_height();
        dom.pitch = grid.get_grid_width();
        return dom;
    }

    void fold_events() {
        kernel_runtime = get_kernel_runtime();
        drop_events();
    }
    void drop_events() {
        for (auto &pair : timed) {
            ststhip_event_destroy(pair.first);
            ststhip_event_destroy(pair.second);
        }
        timed.clear();
    }

    Params params;
    std::size_t n_processed_cells;
    double walltime;
    double kernel_runtime;
    std::vector<std::pair<ststhip_event, ststhip_event>> timed;
};

} // namespace hip
} // namespace stencil
