// 2-D TE-mode FDTD update with per-cell material coefficients, precompiled into libststhip.so.
// Arithmetic parity: examples/fdtd/src/Kernel.hpp:52-141 with material/CoefResolver.hpp:24-68 of the
// reference: sub-iteration 0 updates ex, ey from hz; sub-iteration 1 updates hz from ex, ey, adds the
// source term (time-dependent value = source amplitude) inside the source radius until the cut-off
// iteration and accumulates hz^2 after the detection iteration.
#pragma once
#include <StencilStream/Stencil.hpp>
#include <cmath>
#include <ststhip.h>
#include <tuple>

namespace stencil {
namespace apps {

struct FdtdCell {
    float ex, ey, hz, hz_sum;
    float ca, cb, da, db;
    static constexpr auto fields =
        std::make_tuple(&FdtdCell::ex, &FdtdCell::ey, &FdtdCell::hz, &FdtdCell::hz_sum,
                        &FdtdCell::ca, &FdtdCell::cb, &FdtdCell::da, &FdtdCell::db);
};

struct Fdtd {
    using Cell = FdtdCell;
    using TimeDependentValue = float;
    using Block = ststhip_fdtd_params;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 2;
    // sub-iteration 0 reads hz of the cell, its WEST and its NORTH neighbour (Kernel.hpp:96-101), sub-iteration 1 reads
    // ex of the EAST and ey of the SOUTH neighbour (:103-105): over a generation the dependency cone grows by one
    // column per side, not by radius x sub-iterations = two (hip/internal/Sweep.hpp: GC)
    static constexpr std::size_t halo_columns_per_generation = 1;

    Block p;

    // the material coefficients are copied through unchanged (Kernel.hpp:94-140 never assigns them)
    static constexpr auto constant_fields =
        std::make_tuple(&FdtdCell::ca, &FdtdCell::cb, &FdtdCell::da, &FdtdCell::db);

    static Fdtd from_params(Block const &block) { return Fdtd{block}; }

    // host side, once per iteration (Kernel.hpp:80-84): float arithmetic, libm cosf/expf
    float get_time_dependent_value(std::size_t i_iteration) const {
        float current_time = i_iteration * p.dt;
        float wave_progress = (current_time - p.t_0) / p.tau;
        return std::cos(p.omega * current_time) * std::exp(-1 * wave_progress * wave_progress);
    }

    STST_HD Cell operator()(Stencil<Cell, 1, float> const &s) const {
        Cell cell = s[0][0];
        const float r = s.id[0];
        const float c = s.id[1];
        const float source_distance_score = r * (r - 2 * p.source_r) + c * (c - 2 * p.source_c);

        if (s.subiteration == 0) {
            cell.ex *= cell.ca;
            cell.ex += cell.cb * (s[0][0].hz - s[0][-1].hz);
            cell.ey *= cell.ca;
            cell.ey += cell.cb * (s[-1][0].hz - s[0][0].hz);
        } else {
            cell.hz *= cell.da;
            cell.hz += cell.db * (s[0][1].ex - s[0][0].ex + s[0][0].ey - s[1][0].ey);

            if (source_distance_score <= p.source_distance_bound &&
                s.iteration <= p.cutoff_iteration) {
                float interp_factor;
                if (p.source_radius_squared != 0) {
                    float cell_distance_squared = source_distance_score +
                                                  p.source_c * p.source_c + p.source_r * p.source_r;
                    // the literal 1.0 is a double: subtraction in fp64, as in the reference
                    interp_factor = 1.0 - float(cell_distance_squared) / p.source_radius_squared;
                } else {
                    interp_factor = 1.0;
                }
                cell.hz += interp_factor * s.time_dependent_value;
            }
            if (s.iteration > p.detect_iteration)
                cell.hz_sum += cell.hz * cell.hz;
        }
        return cell;
    }
};

// The same cell as two 16-byte halves -- what the update changes, and the material coefficients it only copies --
// and the same update on it.  Swept on planes this gives TWO planes of 16-byte elements instead of eight of 4:
// every access is one full-width vector per lane, and the coefficient plane is read but never written again
// (constant_fields), which the 32-byte AoS cell cannot offer (a half-written cell costs the same transaction) and
// the eight thin planes pay for with misaligned 160-byte strips.  The bytes of a cell are those of FdtdCell /
// the reference's CoefCell (material/CoefResolver.hpp:24-31), so AoS grids interchange.
struct FdtdFields {
    float ex, ey, hz, hz_sum;
};
struct FdtdMaterial {
    float ca, cb, da, db;
};
struct FdtdGroupedCell {
    FdtdFields f;
    FdtdMaterial m;
    static constexpr auto fields = std::make_tuple(&FdtdGroupedCell::f, &FdtdGroupedCell::m);
};
static_assert(sizeof(FdtdGroupedCell) == sizeof(FdtdCell));

struct FdtdGrouped {
    using Cell = FdtdGroupedCell;
    using TimeDependentValue = float;
    using Block = ststhip_fdtd_params;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 2;
    static constexpr std::size_t halo_columns_per_generation = 1; // as Fdtd above: west / north, then east / south

    Block p;

    static constexpr auto constant_fields = std::make_tuple(&FdtdGroupedCell::m);

    static FdtdGrouped from_params(Block const &block) { return FdtdGrouped{block}; }
    float get_time_dependent_value(std::size_t i_iteration) const { return Fdtd{p}.get_time_dependent_value(i_iteration); }

    // Kernel.hpp:86-128, expression for expression as in Fdtd::operator() above
    STST_HD Cell operator()(Stencil<Cell, 1, float> const &s) const {
        Cell cell = s[0][0];
        const float r = s.id[0];
        const float c = s.id[1];
        const float source_distance_score = r * (r - 2 * p.source_r) + c * (c - 2 * p.source_c);

        if (s.subiteration == 0) {
            cell.f.ex *= cell.m.ca;
            cell.f.ex += cell.m.cb * (s[0][0].f.hz - s[0][-1].f.hz);
            cell.f.ey *= cell.m.ca;
            cell.f.ey += cell.m.cb * (s[-1][0].f.hz - s[0][0].f.hz);
        } else {
            cell.f.hz *= cell.m.da;
            cell.f.hz += cell.m.db * (s[0][1].f.ex - s[0][0].f.ex + s[0][0].f.ey - s[1][0].f.ey);

            if (source_distance_score <= p.source_distance_bound && s.iteration <= p.cutoff_iteration) {
                float interp_factor;
                if (p.source_radius_squared != 0) {
                    float cell_distance_squared =
                        source_distance_score + p.source_c * p.source_c + p.source_r * p.source_r;
                    interp_factor = 1.0 - float(cell_distance_squared) / p.source_radius_squared;
                } else {
                    interp_factor = 1.0;
                }
                cell.f.hz += interp_factor * s.time_dependent_value;
            }
            if (s.iteration > p.detect_iteration)
                cell.f.hz_sum += cell.f.hz * cell.f.hz;
        }
        return cell;
    }
};

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// Independent waves (rounds 1 and 2; profiles/r01_tune_shapes_apps.txt, 4608^2): K = 1 with T = 4, P = 4: 270 / 231
// (AoS / planes), T = 6, P = 2: 347 / 284, T = 7: 234 (register cliff) Gcell/s.  Round 3, four stages per column strip
// (a wave keeps the windows of S/4 levels only, so the launch can be deeper): AoS T = 6: 378 -> 409, T = 8: 464;
// the two-plane layout T = 6: 449 -> 509, T = 8: 509 (profiles/r03_tune_staged.txt).  Launch depths 8, 4, 2, 1.
template <bool SOA> struct SweepTuning<apps::FdtdGrouped, SOA> {
    static constexpr int cells_per_lane = 1;
    static constexpr int max_generations = 8;
    static constexpr int prefetch_rows = 2;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
};
template <bool SOA> struct SweepTuning<apps::Fdtd, SOA> {
    static constexpr int cells_per_lane = 1;
    static constexpr int max_generations = 8;
    static constexpr int prefetch_rows = 2;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
};
} // namespace hip
} // namespace stencil
