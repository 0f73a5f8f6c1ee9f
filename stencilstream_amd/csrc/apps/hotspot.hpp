// Rodinia HotSpot transition function, precompiled into libststhip.so.
// Arithmetic parity: examples/hotspot/hotspot.cpp:57-97 of the reference (two-field cell with the SoA
// opt-in tuple, reflecting edges by substituting the centre temperature, power carried through unchanged,
// ambient temperature 80).  The reference computes in fp32 (`typedef float FLOAT`, hotspot.cpp:38);
// Real = double is the same formula in fp64 (BASELINE.json names an fp64 HotSpot; it is an extra here).
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>
#include <ststhip.h>
#include <tuple>
#include <type_traits>

namespace stencil {
namespace apps {

template <typename Real> struct HotspotCellT {
    Real temp;
    Real power;
    static constexpr auto fields = std::make_tuple(&HotspotCellT::temp, &HotspotCellT::power);
};

template <typename Real> struct HotspotT : public BaseTransitionFunction {
    using Cell = HotspotCellT<Real>;
    using Block = std::conditional_t<std::is_same_v<Real, float>, ststhip_hotspot_params,
                                     ststhip_hotspot_params_f64>;

    Real Rx_1, Ry_1, Rz_1, Cap_1;

    // `power` is copied through unchanged (hotspot.cpp:96): the planes layout need not store it again
    static constexpr auto constant_fields = std::make_tuple(&Cell::power);

    static HotspotT from_params(Block const &p) {
        HotspotT h;
        h.Rx_1 = p.Rx_1;
        h.Ry_1 = p.Ry_1;
        h.Rz_1 = p.Rz_1;
        h.Cap_1 = p.Cap_1;
        return h;
    }

    STST_HD Cell operator()(Stencil<Cell, 1> const &s) const {
        const Real amb_temp = Real(80.0);
        const Real power = s[0][0].power;
        const Real old = s[0][0].temp;
        Real top = s[-1][0].temp, bottom = s[1][0].temp;
        Real left = s[0][-1].temp, right = s[0][1].temp;

        if (s.id[0] == 0)
            top = old;
        else if (s.id[0] == s.grid_range[0] - 1)
            bottom = old;
        if (s.id[1] == 0)
            left = old;
        else if (s.id[1] == s.grid_range[1] - 1)
            right = old;

        const Real next = old + Cap_1 * (power + (bottom + top - Real(2) * old) * Ry_1 +
                                         (right + left - Real(2) * old) * Rx_1 +
                                         (amb_temp - old) * Rz_1);
        return Cell{next, power};
    }

    // The same update for a cell that is not on the rim of the grid (no edge reflection to decide);
    // the sweep uses it in waves that lie completely inside the grid.  Same expression, same bits.
    STST_HD Cell interior(Stencil<Cell, 1> const &s) const {
        const Real amb_temp = Real(80.0);
        const Real power = s[0][0].power;
        const Real old = s[0][0].temp;
        const Real top = s[-1][0].temp, bottom = s[1][0].temp;
        const Real left = s[0][-1].temp, right = s[0][1].temp;
        const Real next = old + Cap_1 * (power + (bottom + top - Real(2) * old) * Ry_1 +
                                         (right + left - Real(2) * old) * Rx_1 +
                                         (amb_temp - old) * Rz_1);
        return Cell{next, power};
    }
};

using HotspotCell = HotspotCellT<float>;
using Hotspot = HotspotT<float>;

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// fp32 HotSpot, 8192^2 (profiles/r03_tune_staged.txt): the independent-wave shapes K = 1, T = 8 on planes 1777 and
// K = 2, T = 8 as AoS 1821 Gcell/s; four stages per column strip: planes K = 2, T = 12: 2090 (K = 1, T = 8: 1750,
// K = 2, T = 8: 1820, K = 2, T = 16: 1940, K = 4, T = 12: 1950), AoS K = 2, T = 16: 1970 (T = 8: 1810).
template <> struct SweepTuning<apps::HotspotT<float>, true> {
    static constexpr int cells_per_lane = 2;
    static constexpr int max_generations = 12;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
};
template <> struct SweepTuning<apps::HotspotT<float>, false> {
    static constexpr int cells_per_lane = 2;
    static constexpr int max_generations = 16;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
};
} // namespace hip
} // namespace stencil
