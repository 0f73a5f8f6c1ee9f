// Glue between the precompiled transition functions (apps/*.hpp) and layer 1 of the C ABI
// (include/ststhip.h: ststhip_app_*).  Each application translation unit instantiates the same
// sweep templates a C++ user would (StencilStream/hip/internal/Sweep.hpp) and registers a
// type-erased entry here.
#pragma once
#include <StencilStream/hip/internal/Sweep.hpp>
#include <ststhip.h>

#include <cstring>
#include <exception>
#include <vector>

namespace ststhip_detail {

struct AppEntry {
    ststhip_app_info info;
    int (*sweep)(const void *tf_params, const void *halo_cell, const ststhip_domain *dom,
                 const void *const *src, void *const *dst, std::uint64_t out_begin,
                 std::uint64_t out_end, std::uint64_t iteration, std::uint32_t n_generations,
                 ststhip_stream stream);
    // values of generations [iteration_offset, iteration_offset + n) into `values` (n * info.tdv_size bytes),
    // evaluated on the host; nullptr for functions without a time-dependent value
    void (*fill_tdv)(const void *tf_params, std::uint64_t iteration_offset, std::uint64_t n, void *values);
    // largest scratch (private memory) per work-item over the kernels a launch of `n_generations` may start: the
    // default shape, its narrow form, the variant that leaves out constant planes
    int (*scratch_bytes)(std::uint32_t n_generations, std::size_t *bytes);
};

void register_app(AppEntry const &entry);
const AppEntry *find_app(const char *name);
void set_error(const char *message);
int fail(int status, const char *message);

template <typename F, bool SOA> struct AppAdapter {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    using Planes = stencil::hip::internal::PlaneSet<Cell, SOA>;
    using Tuning = stencil::hip::SweepTuning<F, SOA>;

    static int sweep(const void *tf_params, const void *halo_cell, const ststhip_domain *dom,
                     const void *const *src, void *const *dst, std::uint64_t out_begin,
                     std::uint64_t out_end, std::uint64_t iteration, std::uint32_t n_generations,
                     ststhip_stream stream) {
        try {
            typename F::Block block;
            std::memcpy(&block, tf_params, sizeof block);
            const F f = F::from_params(block);
            Cell halo;
            std::memcpy(static_cast<void *>(&halo), halo_cell, sizeof(Cell));
            // a single launch through ststhip_app_sweep: the launch's values, evaluated here; inside
            // ststhip_app_run the pass driver has put the whole call's values into a device table
            TDV tdv[Tuning::max_generations];
            const void *table = nullptr;
            ststhip_current_tdv_table(&table, nullptr, nullptr, nullptr);
            if (!table)
                for (std::uint32_t t = 0; t < n_generations && t < std::uint32_t(Tuning::max_generations); t++)
                    tdv[t] = f.get_time_dependent_value(iteration + t);
            Planes s, d;
            for (int i = 0; i < Planes::n_planes; i++) {
                s.plane[i] = const_cast<void *>(src[i]);
                d.plane[i] = dst[i];
            }
            stencil::hip::internal::dispatch_sweep<F, SOA>(int(n_generations), f, halo, table ? nullptr : tdv, *dom,
                                                           s, d, out_begin, out_end, iteration,
                                                           stream);
            return STSTHIP_OK;
        } catch (stencil::hip::internal::runtime_error const &e) {
            return e.status; // message already recorded by the failing runtime call
        } catch (std::exception const &e) {
            return fail(STSTHIP_ERR_INVALID, e.what());
        }
    }

    static void fill_tdv(const void *tf_params, std::uint64_t iteration_offset, std::uint64_t n, void *values) {
        typename F::Block block;
        std::memcpy(&block, tf_params, sizeof block);
        const F f = F::from_params(block);
        for (std::uint64_t i = 0; i < n; i++) {
            const TDV v = f.get_time_dependent_value(iteration_offset + i);
            std::memcpy(static_cast<unsigned char *>(values) + i * sizeof(TDV), &v, sizeof(TDV));
        }
    }

    // scratch of every kernel dispatch_sweep<F, SOA> can reach at depth T (and of the narrow form's)
    template <typename G, int T> static int scratch_of_depth(std::uint32_t n_generations, std::size_t &worst) {
        namespace in = stencil::hip::internal;
        using GT = stencil::hip::SweepTuning<G, SOA>;
        if (n_generations == std::uint32_t(T)) {
            auto ask = [&](const void *kernel) {
                std::size_t b = 0;
                const int rc = ststhip_kernel_scratch_bytes(kernel, &b);
                worst = std::max(worst, b);
                return rc;
            };
            if (int rc = ask(reinterpret_cast<const void *>(&in::sweep_kernel<in::SweepOf<G, SOA, T>, GT::min_waves_per_simd>)))
                return rc;
            if constexpr (SOA && in::constant_plane_mask<G>() != 0)
                if (int rc = ask(reinterpret_cast<const void *>(&in::sweep_kernel<in::SweepOf<G, SOA, T>, GT::min_waves_per_simd, true>)))
                    return rc;
            return STSTHIP_OK;
        }
        if constexpr (T > 1)
            return scratch_of_depth<G, T / 2>(n_generations, worst);
        else
            return fail(STSTHIP_ERR_INVALID, "n_generations is not a compiled temporal-blocking depth");
    }
    static int scratch_bytes(std::uint32_t n_generations, std::size_t *bytes) {
        namespace in = stencil::hip::internal;
        std::size_t worst = 0;
        if (int rc = scratch_of_depth<F, Tuning::max_generations>(n_generations, worst))
            return rc;
        if constexpr (in::has_narrow_form<F, SOA>())
            if (int rc = scratch_of_depth<in::NarrowForm<F>, Tuning::max_generations>(n_generations, worst))
                return rc;
        *bytes = worst;
        return STSTHIP_OK;
    }

    static AppEntry make(const char *name) {
        AppEntry e;
        std::memset(&e.info, 0, sizeof e.info);
        e.info.name = name;
        e.info.cell_size = sizeof(Cell);
        e.info.params_size = sizeof(typename F::Block);
        e.info.stencil_radius = std::uint32_t(F::stencil_radius);
        e.info.n_subiterations = std::uint32_t(F::n_subiterations);
        e.info.n_planes = Planes::n_planes;
        for (int i = 0; i < Planes::n_planes; i++) {
            e.info.plane_elem_size[i] = std::uint32_t(Planes::elem_size(i));
            e.info.field_offset[i] = std::uint32_t(Planes::elem_offset(i));
        }
        e.info.max_generations = Tuning::max_generations;
        e.info.tdv_size = std::is_same_v<TDV, std::monostate> ? 0 : std::uint32_t(sizeof(TDV));
        e.info.halo_depth_per_generation = std::uint32_t(F::stencil_radius * F::n_subiterations);
        e.info.strip_width = std::uint32_t(stencil::hip::internal::SweepOf<F, SOA>::OW_PER_WAVE);
        e.info.cells_per_lane = std::uint32_t(Tuning::cells_per_lane);
        e.info.prefetch_rows = std::uint32_t(Tuning::prefetch_rows);
        e.info.stages = std::uint32_t(stencil::hip::internal::SweepOf<F, SOA>::W);
        e.info.default_generations = std::uint32_t(stencil::hip::internal::default_generations_for<F, SOA>());
        e.sweep = &sweep;
        e.fill_tdv = e.info.tdv_size ? &fill_tdv : nullptr;
        e.scratch_bytes = &scratch_bytes;
        return e;
    }
};

// A transition function with an explicit pipeline shape (used to register tuning experiments and
// hand-picked shapes next to the heuristic default).  STAGES: waves of a workgroup that share one column strip as
// a pipeline over the levels (Sweep.hpp).  An explicit shape is launched as it is: no narrow form.
template <typename F, int K, int T, int P, int MINW = 1, bool INTERIOR = true, int STAGES = 1, bool PINNED = false>
struct Shaped : public F {
    using Block = typename F::Block;
    Shaped() = default;
    Shaped(F const &f) : F(f) {}
    static Shaped from_params(Block const &b) { return Shaped(F::from_params(b)); }
};

struct AppRegistrar {
    AppRegistrar(AppEntry const &entry) { register_app(entry); }
};

} // namespace ststhip_detail

namespace stencil {
namespace hip {
template <typename F, int K, int T, int P, int MINW, bool INTERIOR, int STAGES, bool PINNED, bool SOA>
struct SweepTuning<::ststhip_detail::Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>, SOA> {
    static constexpr int cells_per_lane = K;
    static constexpr int max_generations = T;
    static constexpr int prefetch_rows = P;
    static constexpr bool interior_variant = INTERIOR;
    static constexpr int min_waves_per_simd = MINW;
    static constexpr int stages = STAGES;
    static constexpr bool pinned_loads = PINNED;
    static constexpr bool narrow_form = false;
    // hints the wrapped function's own tuning carries
    static constexpr bool trapezoid_fill = internal::trapezoid_fill_for<F, SOA>();
    static constexpr bool streaming_stores = internal::streaming_stores_for<F, SOA>();
};
} // namespace hip
} // namespace stencil

#define STSTHIP_CONCAT2(a, b) a##b
#define STSTHIP_CONCAT(a, b) STSTHIP_CONCAT2(a, b)
#define STSTHIP_REGISTER_APP(name, F, SOA)                                                         \
    static ::ststhip_detail::AppRegistrar STSTHIP_CONCAT(ststhip_app_registrar_, __COUNTER__)(     \
        ::ststhip_detail::AppAdapter<F, SOA>::make(name))
