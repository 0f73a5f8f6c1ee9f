// Self-checking transition function: every cell records where and when it is, and every update
// verifies that all neighbours carry the right coordinates, iteration and sub-iteration, that
// out-of-grid neighbours are the halo cell, and that the time-dependent value equals the
// iteration.  Parity: tests/TransFuncs.hpp:33-104 of the reference (FPGATransFunc<radius>).
#pragma once
#include <StencilStream/Stencil.hpp>
#include <tuple>

namespace stencil {
namespace apps {

enum class SelfCheckStatus : int { Normal = 0, Invalid = 1, Halo = 2 };

struct SelfCheckCell {
    int r, c, i_iteration, i_subiteration;
    SelfCheckStatus status;

    STST_HD static SelfCheckCell halo() {
        return SelfCheckCell{0, 0, 0, 0, SelfCheckStatus::Halo};
    }
    static constexpr auto fields =
        std::make_tuple(&SelfCheckCell::r, &SelfCheckCell::c, &SelfCheckCell::i_iteration,
                        &SelfCheckCell::i_subiteration, &SelfCheckCell::status);
};

template <std::size_t radius> struct SelfCheck {
    using Cell = SelfCheckCell;
    using TimeDependentValue = std::size_t;
    struct Block {
        int unused;
    };
    static constexpr std::size_t stencil_radius = radius;
    static constexpr std::size_t n_subiterations = 2;

    static SelfCheck from_params(Block const &) { return SelfCheck(); }

    STST_HD std::size_t get_time_dependent_value(std::size_t i_iteration) const {
        return i_iteration;
    }

    STST_HD Cell operator()(Stencil<Cell, radius, std::size_t> const &s) const {
        Cell next = s[0][0];
        bool ok = true;
#pragma unroll
        for (int dr = -int(radius); dr <= int(radius); dr++) {
#pragma unroll
            for (int dc = -int(radius); dc <= int(radius); dc++) {
                const Cell seen = s[dr][dc];
                const int at_r = int(s.id[0]) + dr, at_c = int(s.id[1]) + dc;
                const bool inside = at_r >= 0 && at_c >= 0 && std::size_t(at_r) < s.grid_range[0] &&
                                    std::size_t(at_c) < s.grid_range[1];
                const Cell expect =
                    inside ? Cell{at_r, at_c, int(s.iteration), int(s.subiteration),
                                  SelfCheckStatus::Normal}
                           : Cell::halo();
                ok = ok && seen.r == expect.r && seen.c == expect.c &&
                     seen.i_iteration == expect.i_iteration &&
                     seen.i_subiteration == expect.i_subiteration && seen.status == expect.status;
            }
        }
        ok = ok && s.time_dependent_value == s.iteration;

        next.status = ok ? SelfCheckStatus::Normal : SelfCheckStatus::Invalid;
        if (next.i_subiteration == int(n_subiterations) - 1) {
            next.i_iteration += 1;
            next.i_subiteration = 0;
        } else {
            next.i_subiteration += 1;
        }
        return next;
    }
};

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// 25 comparisons per neighbour: the kernel is register-bound long before the heuristic's window is
// (T = 4: 248 VGPRs).  Correctness only, so the shallow pipeline stays.
template <bool SOA> struct SweepTuning<apps::SelfCheck<1>, SOA> {
    static constexpr int cells_per_lane = 1;
    static constexpr int max_generations = 4;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
};
} // namespace hip
} // namespace stencil
