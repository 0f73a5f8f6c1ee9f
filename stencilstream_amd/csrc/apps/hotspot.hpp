// Rodinia HotSpot transition function, precompiled into libststhip.so.
// Arithmetic parity: examples/hotspot/hotspot.cpp:57-97 of the reference (fp32, two-field cell with
// the SoA opt-in tuple, reflecting edges by substituting the centre temperature, power carried
// through unchanged, ambient temperature 80).
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>
#include <ststhip.h>
#include <tuple>

namespace stencil {
namespace apps {

struct HotspotCell {
    float temp;
    float power;
    static constexpr auto fields = std::make_tuple(&HotspotCell::temp, &HotspotCell::power);
};

struct Hotspot : public BaseTransitionFunction {
    using Cell = HotspotCell;
    using Block = ststhip_hotspot_params;

    float Rx_1, Ry_1, Rz_1, Cap_1;

    static Hotspot from_params(Block const &p) {
        Hotspot h;
        h.Rx_1 = p.Rx_1;
        h.Ry_1 = p.Ry_1;
        h.Rz_1 = p.Rz_1;
        h.Cap_1 = p.Cap_1;
        return h;
    }

    STST_HD Cell operator()(Stencil<HotspotCell, 1> const &s) const {
        const float amb_temp = 80.0f;
        const float power = s[0][0].power;
        const float old = s[0][0].temp;
        float top = s[-1][0].temp, bottom = s[1][0].temp;
        float left = s[0][-1].temp, right = s[0][1].temp;

        if (s.id[0] == 0)
            top = old;
        else if (s.id[0] == s.grid_range[0] - 1)
            bottom = old;
        if (s.id[1] == 0)
            left = old;
        else if (s.id[1] == s.grid_range[1] - 1)
            right = old;

        const float next = old + Cap_1 * (power + (bottom + top - 2.f * old) * Ry_1 +
                                          (right + left - 2.f * old) * Rx_1 +
                                          (amb_temp - old) * Rz_1);
        return HotspotCell{next, power};
    }

    // The same update for a cell that is not on the rim of the grid (no edge reflection to decide);
    // the sweep uses it in waves that lie completely inside the grid.  Same expression, same bits.
    STST_HD Cell interior(Stencil<HotspotCell, 1> const &s) const {
        const float amb_temp = 80.0f;
        const float power = s[0][0].power;
        const float old = s[0][0].temp;
        const float top = s[-1][0].temp, bottom = s[1][0].temp;
        const float left = s[0][-1].temp, right = s[0][1].temp;
        const float next = old + Cap_1 * (power + (bottom + top - 2.f * old) * Ry_1 +
                                          (right + left - 2.f * old) * Rx_1 +
                                          (amb_temp - old) * Rz_1);
        return HotspotCell{next, power};
    }
};

} // namespace apps
} // namespace stencil
