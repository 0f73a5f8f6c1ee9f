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
// gathers back (:294-438) -- except for cells of 16 or 32 bytes with four or more fields, which are swept as AoS (SplitCellPolicy below).
//
// What differs by design: instead of one kernel per (iteration, sub-iteration), one kernel
// advances up to SweepTuning<F>::max_generations generations (hip/internal/Sweep.hpp).
#pragma once
#include "../Concepts.hpp"
#include "../tdv/SinglePassStrategies.hpp"
#include "Grid.hpp"
#include "internal/Sweep.hpp"

#include <chrono>
#include <utility>
#include <vector>

namespace stencil {
namespace hip {

// Which layout the sweeps of StencilUpdate<F, true> run on.  The reference's GPU backend needs per-field
// planes for coalescing because one work-item loads one cell (cuda/StencilUpdate.hpp:346-396).  Here a lane
// loads a whole cell, and for cells of 16 or 32 bytes made of four or more fields that is one or two full-width
// vector accesses instead of many narrow ones: the AoS
// sweep is then the faster one (FDTD, 32 bytes: 342 vs 270 Gcell/s at 4608^2, profiles/r01_tune_shapes_apps_2.txt;
// the unchanged example 14.3 -> 11.2 s together with the depth rule) and needs no scatter / gather passes.
// Thin cells keep their planes (HotSpot, 8 bytes: 1632 vs 1583; in fp64, two 8-byte fields: 1216 vs 1110), and so
// do cells that do not fill whole vector
// accesses or are fatter (convection, 88 bytes: 0.34 s on planes against 0.39-0.42 s as AoS cells,
// profiles/r01_ab_examples.txt).  Results are identical either way, so split_cell_structure = true is honoured as
// a request for planes except where AoS is known to win.  Specialise to decide differently for a function.
template <typename F> struct SplitCellPolicy {
    static constexpr bool sweep_on_planes = [] {
        if constexpr (internal::SplittableCell<typename F::Cell>)
            return !((sizeof(typename F::Cell) == 16 || sizeof(typename F::Cell) == 32) &&
                     internal::field_count<typename F::Cell>() >= 4);
        else
            return true; // PlaneSet reports the missing Cell::fields
    }();
};

namespace internal {
// PrecomputeOnDeviceStrategy: the values of generations [offset, offset + n) computed by the device
template <typename F>
__global__ void fill_tdv_kernel(const F f, std::size_t offset, std::size_t n, typename F::TimeDependentValue *values) {
    const std::size_t i = blockIdx.x * std::size_t(blockDim.x) + threadIdx.x;
    if (i < n)
        values[i] = f.get_time_dependent_value(offset + i);
}
} // namespace internal

// TDVStrategy (an extension: the reference's cuda::StencilUpdate has two template parameters and always evaluates
// on the host): where the time-dependent values come from, see tdv/SinglePassStrategies.hpp.  The default gives
// the kernels the host's values, bit for bit what the reference's cpu / cuda backends feed theirs.
template <concepts::TransitionFunction F, bool split_cell_structure = false,
          typename TDVStrategy = tdv::single_pass::PrecomputeOnHostStrategy>
class StencilUpdate {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    static constexpr bool has_tdv = !std::is_empty_v<TDV>;
    static constexpr bool inline_tdv = has_tdv && TDVStrategy::kind == tdv::single_pass::Kind::Inline;
    static constexpr bool on_planes = [] {
        if constexpr (split_cell_structure)
            return SplitCellPolicy<F>::sweep_on_planes;
        else
            return false;
    }();
    using Planes = internal::PlaneSet<Cell, on_planes>;

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
        : params(params), n_processed_cells(0), walltime(0.0), kernel_runtime(0.0) {
        // Building the update object loads its kernels: the code object of this instantiation is read into the device
        // at the first question asked of one of its kernels (8 ms for the Jacobi example's) -- here, once, instead of
        // inside the first call, where it used to hide behind the grid's upload and would now sit in front of the
        // passes that follow the upload block by block (simulate()).  A host without a usable GPU learns so from
        // operator(), as before.
        try {
            internal::ensure_runtime(params.device.hip_index());
            (void)spill_free_depth();
        } catch (...) {
        }
    }

    Params &get_params() { return params; }
    std::size_t get_n_processed_cells() const { return n_processed_cells; }
    double get_walltime() const { return walltime; }

    // For drivers other than operator() (hip/StripUpdate.hpp: one strip of a grid cut over several GPUs): the launch
    // callback of this instantiation -- its context is a pointer to this object, which must stay where it is -- and
    // the description ststhip_run_passes / ststhip_strip_create_custom plan their passes with.
    static ststhip_sweep_fn launch_entry() { return &sweep_trampoline; }
    static ststhip_sweep_desc sweep_description() {
        ststhip_sweep_desc desc = {};
        desc.n_planes = Planes::n_planes;
        desc.max_generations = std::uint32_t(spill_free_depth());
        desc.halo_depth_per_generation = std::uint32_t(F::stencil_radius * F::n_subiterations);
        desc.strip_width = std::uint32_t(internal::SweepOf<F, on_planes>::OW_PER_WAVE);
        for (int f = 0; f < Planes::n_planes; f++)
            desc.plane_elem_size[f] = Planes::elem_size(f);
        // a family compiled deeper than its rule trusts (SweepTuning::default_generations): unmeasured launches run the
        // trusted depth, the pass driver times both on the first long call for a grid shape
        constexpr int trusted = internal::default_generations_for<F, on_planes>();
        if (trusted < SweepTuning<F, on_planes>::max_generations && std::uint32_t(trusted) < desc.max_generations) {
            desc.alt_generations = std::uint32_t(trusted);
            desc.tune_key = reinterpret_cast<std::uintptr_t>(&sweep_trampoline);
        }
        return desc;
    }
    // the same with the host-side source of the time-dependent values: a driver builds one device table per call
    // from it (the strip driver; operator() decides per strategy in run_passes)
    static ststhip_sweep_desc sweep_description_with_host_values() {
        ststhip_sweep_desc desc = sweep_description();
        if constexpr (has_tdv && !inline_tdv) {
            desc.tdv_size = sizeof(TDV);
            desc.fill_tdv = &fill_values;
        }
        return desc;
    }
    static constexpr bool sweeps_on_planes = on_planes;
    static constexpr int n_planes = Planes::n_planes;
    static std::size_t plane_elem_size(int f) { return Planes::elem_size(f); }
    static std::size_t plane_elem_offset(int f) { return Planes::elem_offset(f); }

    // Sum of the sweep kernels' device time in seconds (HIP events around every launch); needs
    // Params::profiling, which also makes the call synchronous.
    double get_kernel_runtime() const { return kernel_runtime; }

    GridImpl operator()(GridImpl &source_grid) {
        internal::ensure_runtime(params.device.hip_index());
        ststhip_stream stream = internal::default_stream();

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
    static void fill_values(void *ctx, std::uint64_t offset, std::uint64_t n, void *values) {
        StencilUpdate const *self = static_cast<StencilUpdate const *>(ctx);
        for (std::uint64_t i = 0; i < n; i++)
            static_cast<TDV *>(values)[i] = self->params.transition_function.get_time_dependent_value(offset + i);
    }

    // One launch, called back by the runtime's pass driver (ststhip_run_passes): evaluates the
    // time-dependent values of the launch's generations on the host and starts the sweep kernel that
    // was instantiated for F in this translation unit.
    static int sweep_trampoline(void *ctx, const ststhip_domain *dom, const void *const *src,
                                void *const *dst, std::uint64_t out_begin, std::uint64_t out_end,
                                std::uint64_t iteration, std::uint32_t depth, ststhip_stream stream) {
        StencilUpdate const *self = static_cast<StencilUpdate const *>(ctx);
        try {
            // the launch's values come from the call's device table (ststhip_current_tdv_table) or from the kernel
            // itself; evaluated here only if there is neither (a driver without a table)
            std::vector<TDV> tdv;
            const void *table = nullptr;
            ststhip_current_tdv_table(&table, nullptr, nullptr, nullptr);
            if (has_tdv && !inline_tdv && !table) {
                tdv.reserve(depth);
                for (std::uint32_t t = 0; t < depth; t++)
                    tdv.push_back(self->params.transition_function.get_time_dependent_value(iteration + t));
            }
            Planes from, to;
            for (int f = 0; f < Planes::n_planes; f++) {
                from.plane[f] = const_cast<void *>(src[f]);
                to.plane[f] = dst[f];
            }
            internal::dispatch_sweep<F, on_planes, SweepTuning<F, on_planes>::max_generations, inline_tdv>(
                int(depth), self->params.transition_function, self->params.halo_value,
                tdv.empty() ? nullptr : tdv.data(), *dom, from, to, out_begin, out_end, iteration, stream);
            return STSTHIP_OK;
        } catch (internal::runtime_error const &e) {
            return e.status;
        } catch (std::exception const &e) {
            ststhip_set_last_error(e.what());
            return STSTHIP_ERR_INVALID;
        }
    }

    // SweepTuning's generic rule sizes the pipeline by the cell alone.  A transition function with a lot of state of
    // its own (FDTD's RenderResolver: sixteen ring bounds and material sets compared per cell) can need far more
    // registers; the compiler then builds the deep kernels with spills to scratch, and they run several times slower
    // than a shallower one (render example, 4608^2: T = 8: 54, T = 4: 69, T = 2: 176 Gcell-updates/s).  So: the
    // deepest compiled depth whose kernel has no scratch, asked of the code object once per instantiation.
    static int spill_free_depth() {
        static const int depth = [] {
            if (internal::options().allow_spilling_depths)
                return int(SweepTuning<F, on_planes>::max_generations);
            return probe_depth<SweepTuning<F, on_planes>::max_generations>();
        }();
        return depth;
    }
    template <int T> static int probe_depth() {
        using Tuning = SweepTuning<F, on_planes>;
        const void *kernel = reinterpret_cast<const void *>(
            &internal::sweep_kernel<internal::SweepOf<F, on_planes, T, inline_tdv>, Tuning::min_waves_per_simd>);
        std::size_t scratch = 0;
        if constexpr (T > 1)
            if (ststhip_kernel_scratch_bytes(kernel, &scratch) == STSTHIP_OK && scratch > 0)
                return probe_depth<T / 2>();
        return T;
    }

    // All passes of one call, from `source` planes into `target` planes.
    void run_passes(ststhip_domain const &dom, Planes const &source, Planes const &target,
                    ststhip_stream stream, std::vector<ststhip_source_block> const &arriving = {}) {
        ststhip_sweep_desc desc = sweep_description();
        // one device table of time-dependent values per call: filled by the host, or by the device
        void *device_values = nullptr;
        if constexpr (has_tdv && !inline_tdv) {
            desc.tdv_size = sizeof(TDV);
            if constexpr (TDVStrategy::kind == tdv::single_pass::Kind::PrecomputeOnDevice) {
                if (params.n_iterations > 0) {
                    device_values = internal::device_alloc_on(params.n_iterations * sizeof(TDV), stream);
                    const F f = params.transition_function;
                    std::size_t offset = params.iteration_offset, n = params.n_iterations;
                    TDV *out = static_cast<TDV *>(device_values);
                    void *kernel_args[] = {const_cast<F *>(&f), &offset, &n, &out};
                    internal::check(ststhip_launch(reinterpret_cast<const void *>(&internal::fill_tdv_kernel<F>),
                                                   unsigned((n + 255) / 256), 1, 1, 256, 1, 1, kernel_args, 0, stream),
                                    "time-dependent values");
                    desc.tdv_device_table = device_values;
                }
            } else {
                desc.fill_tdv = &fill_values;
            }
        }
        ststhip_run_info info = {};
        // (the source's row blocks, for this call: ststhip.h, ststhip_set_source_arrival)
        internal::check(ststhip_set_source_arrival(arriving.data(), std::uint32_t(arriving.size())), "ststhip_set_source_arrival");
        internal::check(ststhip_run_passes(&sweep_trampoline, this, &desc, &dom,
                                           const_cast<const void *const *>(source.plane), target.plane,
                                           params.iteration_offset, params.n_iterations, 0,
                                           params.profiling ? 1 : 0, stream, &info),
                        "ststhip_run_passes");
        kernel_runtime += info.kernel_time_s;
        if (device_values)
            ststhip_free_async(device_values, stream);
    }

    // The pass driver takes its swap planes from the runtime's pool when it starts.  With the upload in row blocks the
    // chip is busy by then, and an allocation the pool has to pass on to the device waits for it (hipMalloc beside
    // running kernels: 15 ms for 512 MiB, measured in front of the HotSpot example's first pass).  So: the planes go
    // through the pool once BEFORE the upload is queued -- allocated and released on `stream`, where the driver's
    // request finds them.
    void reserve_swap_planes(ststhip_domain const &dom, ststhip_stream stream) {
        if (params.n_iterations < 2)
            return;
        const std::size_t n_cells = std::size_t(dom.local_rows) * dom.pitch;
        void *held[Planes::n_planes];
        for (int f = 0; f < Planes::n_planes; f++)
            held[f] = internal::device_alloc_on(n_cells * Planes::elem_size(f), stream);
        for (int f = 0; f < Planes::n_planes; f++)
            ststhip_free_async(held[f], stream);
    }

    GridImpl simulate(GridImpl &source_grid, ststhip_stream stream)
        requires(!on_planes)
    {
        // the reference's split path always returns a fresh grid (cuda/StencilUpdate.hpp:285,440), its AoS
        // path a handle onto the source when there is nothing to do (:206,275)
        if (params.n_iterations == 0 && !split_cell_structure)
            return source_grid;
        ststhip_domain dom = domain_of(source_grid);
        GridImpl result = source_grid.make_similar();
        Planes from, to;
        // (the upload is queued in front of the sweeps and not waited for: the result grid and the pass driver's swap
        // buffer are allocated while it runs)
        // ... in row blocks when it is worth it, and the pass driver starts on the rows that have arrived)
        std::vector<ststhip_source_block> arriving;
        to.plane[0] = result.device_cells_for_overwrite();
        reserve_swap_planes(dom, stream);
        from.plane[0] = const_cast<Cell *>(
            source_grid.device_cells_arriving(stream, arriving, false, [](ststhip_stream, std::size_t, std::size_t) {}));
        run_passes(dom, from, to, stream, arriving);
        return result;
    }

    GridImpl simulate(GridImpl &source_grid, ststhip_stream stream)
        requires(on_planes)
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
                set.plane[f] = internal::device_alloc_on(n_cells * sizes[f], stream);
        // (the result's cells too, while the chip is idle: see reserve_swap_planes)
        GridImpl result = source_grid.make_similar();
        Cell *result_cells = result.device_cells_for_overwrite();
        // The cells come up in row blocks when the grid is not in HBM yet and large enough, each block scattered into
        // the planes behind its copy, and the pass driver follows the blocks; else one scatter on `stream`.
        const std::size_t width = source_grid.get_grid_width();
        Cell const *cells_base = nullptr; // (known only once the grid has its device buffer)
        std::vector<ststhip_source_block> arriving;
        auto scatter_rows = [&](ststhip_stream on, std::size_t first_row, std::size_t end_row) {
            void *rows_of[n];
            for (int f = 0; f < n; f++)
                rows_of[f] = static_cast<char *>(sets[0].plane[f]) + first_row * width * sizes[f];
            internal::check(ststhip_scatter_fields(cells_base + first_row * width, sizeof(Cell),
                                                   (end_row - first_row) * width, n, offsets, sizes, rows_of, on),
                            "scatter");
        };
        cells_base = source_grid.device_cells_base();
        reserve_swap_planes(dom, stream);
        source_grid.device_cells_arriving(stream, arriving, true, scatter_rows);
        if (arriving.empty())
            scatter_rows(stream, 0, source_grid.get_grid_height());
        run_passes(dom, sets[0], sets[1], stream, arriving); // n_iterations == 0 copies the planes
        internal::check(ststhip_gather_fields(result_cells, sizeof(Cell),
                                              n_cells, n, offsets, sizes,
                                              const_cast<const void *const *>(sets[1].plane), stream),
                        "gather");
        // releases are ordered after the work queued on `stream` (ststhip_free_async), so the planes can be
        // returned while the sweeps and the gather are still in flight
        for (auto &set : sets)
            for (int f = 0; f < n; f++)
                ststhip_free_async(set.plane[f], stream);
        return result;
    }

    static ststhip_domain domain_of(GridImpl const &grid) {
        ststhip_domain dom = {};
        dom.global_height = grid.get_grid_height();
        dom.global_width = grid.get_grid_width();
        dom.row_origin = 0;
        dom.local_rows = grid.get_grid_height();
        dom.pitch = grid.get_grid_width();
        return dom;
    }

    Params params;
    std::size_t n_processed_cells;
    double walltime;
    double kernel_runtime;
};

} // namespace hip
} // namespace stencil
