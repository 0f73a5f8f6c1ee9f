// Conway's Game of Life (B3/S23), precompiled into libststhip.so.
// Parity: examples/conway/conway.cpp:35-56 of the reference (Cell = bool, halo = false).
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>

#include <cstdint>

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

// The same rule on four cells at once: a 32-bit word holds four adjacent one-byte cells (byte 0 = lowest
// column), and the grid is swept as a grid of words.  Every byte-wise sum below stays under 16, so no carry
// crosses a byte and one integer instruction updates four cells:
//     v  = N + C + S                       column sums of the word itself (0..3 per byte)
//     vw, ve                               the same for the words to the west / east
//     t  = (v << 8 | vw >> 24) + v + (v >> 8 | ve << 24)      3x3 block sums, the cell included (0..9)
//     alive next  <=>  ((t - c) | c) == 3                     B3/S23: neighbours == 3, or == 2 and alive
// Bit-identical to Conway above (conway.cpp:35-56) for cells that are 0 or 1, the only values a bool holds.
// Needs halo_value = false and a width (and pitch) that is a multiple of four cells; ststhip_app_run
// ("conway") switches to it when that holds and runs the byte-per-lane kernel otherwise.
struct ConwayPacked : public BaseTransitionFunction {
    using Cell = std::uint32_t;
    struct Block {
        int unused;
    };
    static ConwayPacked from_params(Block const &) { return ConwayPacked(); }
    static constexpr int cells_per_word = 4;

    STST_HD std::uint32_t operator()(Stencil<std::uint32_t, 1> const &s) const {
        const std::uint32_t c = s[0][0];
        const std::uint32_t v = s[-1][0] + c + s[1][0];
        const std::uint32_t vw = s[-1][-1] + s[0][-1] + s[1][-1];
        const std::uint32_t ve = s[-1][1] + s[0][1] + s[1][1];
        const std::uint32_t t = ((v << 8) | (vw >> 24)) + v + ((v >> 8) | (ve << 24));
        const std::uint32_t q = ((t - c) | c) ^ 0x03030303u; // 0 where the cell lives on; below 16 everywhere
        return (~(q + 0x0f0f0f0fu) >> 4) & 0x01010101u;      // bit 4 of q + 15 is set unless q == 0
    }
};

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// K = 4 words, T = 8 is the best of eight shapes (profiles/r01_tune_conway_packed.txt); plain stores: the
// non-temporal ones lose 3 % here (profiles/r01_ab_nt_stores.txt)
template <> struct SweepTuning<apps::ConwayPacked, false> {
    static constexpr int cells_per_lane = 4;
    static constexpr int max_generations = 8;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr bool streaming_stores = false;
    static constexpr int stages = 4; // 16384^2: 9620 -> 12370 Gcell/s (profiles/r03_tune_staged.txt)
};
} // namespace hip
} // namespace stencil
