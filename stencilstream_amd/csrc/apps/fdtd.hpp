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

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// Measured (profiles/r01_tune_shapes_apps.txt, 4608^2): K=1 with T=4,P=4: 270 / 231 (AoS / planes),
// T=5,P=2: 314 / 244, T=6,P=2: 347 / 284, T=7,P=2: 234 (register cliff) Gcell/s.  Launch depths 6, 3, 1.
template <bool SOA> struct SweepTuning<apps::Fdtd, SOA> {
    static constexpr int cells_per_lane = 1;
    static constexpr int max_generations = 6;
    static constexpr int prefetch_rows = 2;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
};
} // namespace hip
} // namespace stencil
