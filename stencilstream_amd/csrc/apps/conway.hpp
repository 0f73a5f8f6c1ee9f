// Conway's Game of Life (B3/S23), precompiled into libststhip.so.
// Parity: examples/conway/conway.cpp:35-56 of the reference (Cell = bool, halo = false).
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>

namespace stencil {
namespace apps {

struct Conway : public BaseTransitionFunction {
    using Cell = bool;
    struct Block {
        int unused;
    };
    static Conway from_params(Block const &) { return Conway(); }

    STST_HD bool operator()(Stencil<bool, 1> const &s) const {
        int alive = 0;
#pragma unroll
        for (int r = -1; r <= 1; r++)
#pragma unroll
            for (int c = -1; c <= 1; c++)
                alive += (s[r][c] && (r != 0 || c != 0)) ? 1 : 0;
        return s[0][0] ? (alive == 2 || alive == 3) : (alive == 3);
    }
};

} // namespace apps
} // namespace stencil
