// Jacobi transition functions, precompiled into libststhip.so.
// Arithmetic parity: examples/jacobi/kernels.hpp:34-319 of the reference -- same operand order
// per variant (N, W, S, E, C), Cell = float, radius 1, one sub-iteration, no time-dependent value.
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>
#include <cstdint>
#include <ststhip.h>

namespace stencil {
namespace apps {

enum class JacobiVariant {
    General1,  // c0*C                                    kernels.hpp:63-66
    Constant2, // (N+S)*0.5                               kernels.hpp:95-98
    Constant3, // (N+C+S)*0.33333334                      kernels.hpp:127-130
    Constant4, // (N+W+S+E)*0.25                          kernels.hpp:159-162
    Constant5, // (N+W+S+E+C)*0.2                         kernels.hpp:191-195
    General4,  // c0*N+c1*W+c2*S+c3*E                     kernels.hpp:229-233
    General5,  // c0*N+c1*W+c2*S+c3*E+c4*C                kernels.hpp:267-271
    General9   // sum over rows then columns of c[r][c]*x kernels.hpp:307-318
};

template <JacobiVariant V> struct Jacobi : public BaseTransitionFunction {
    using Cell = float;
    using Block = ststhip_jacobi_params;
    static constexpr JacobiVariant variant = V;

    float coef[9];

    static Jacobi from_params(Block const &p) {
        Jacobi j;
        for (int i = 0; i < 9; i++)
            j.coef[i] = p.coef[i];
        return j;
    }

    STST_HD float operator()(Stencil<float, 1> const &s) const {
        const float n = s[-1][0], w = s[0][-1], c = s[0][0], e = s[0][1], so = s[1][0];
        if constexpr (V == JacobiVariant::General1) {
            return coef[0] * c;
        } else if constexpr (V == JacobiVariant::Constant2) {
            return (n + so) * 0.5f;
        } else if constexpr (V == JacobiVariant::Constant3) {
            return (n + c + so) * 0.33333334f;
        } else if constexpr (V == JacobiVariant::Constant4) {
            return (n + w + so + e) * 0.25f;
        } else if constexpr (V == JacobiVariant::Constant5) {
            return (n + w + so + e + c) * 0.2f;
        } else if constexpr (V == JacobiVariant::General4) {
            return coef[0] * n + coef[1] * w + coef[2] * so + coef[3] * e;
        } else if constexpr (V == JacobiVariant::General5) {
            return coef[0] * n + coef[1] * w + coef[2] * so + coef[3] * e + coef[4] * c;
        } else {
            float sum = 0.0f;
#pragma unroll
            for (int r = -1; r <= 1; r++)
#pragma unroll
                for (int k = -1; k <= 1; k++)
                    sum += coef[(r + 1) * 3 + (k + 1)] * s[r][k];
            return sum;
        }
    }
};

// A dense 5 x 5 Jacobi of radius 2: Jacobi9General's loop (kernels.hpp:307-318) over a radius-2 stencil.  Not an
// application of the reference (SURVEY 8(f)4 asks for a tuned radius > 1 kernel); it exercises the radius-2
// stencil indexing (Stencil.hpp:120-146) and two-cell halos in the sweep.
struct Jacobi25 {
    using Cell = float;
    using TimeDependentValue = std::monostate;
    using Block = ststhip_jacobi25_params;
    static constexpr std::size_t stencil_radius = 2;
    static constexpr std::size_t n_subiterations = 1;

    float coef[25];

    static Jacobi25 from_params(Block const &p) {
        Jacobi25 j;
        for (int i = 0; i < 25; i++)
            j.coef[i] = p.coef[i];
        return j;
    }
    STST_HD std::monostate get_time_dependent_value(std::size_t) const { return {}; }

    STST_HD float operator()(Stencil<float, 2> const &s) const {
        float sum = 0.0f;
#pragma unroll
        for (int r = -2; r <= 2; r++)
#pragma unroll
            for (int c = -2; c <= 2; c++)
                sum += coef[(r + 2) * 5 + (c + 2)] * s[r][c];
        return sum;
    }
};

} // namespace apps
} // namespace stencil

namespace stencil {
namespace apps {

// Jacobi5General for the case that all five coefficients are the same number c (the reference's own
// benchmark setting, examples/jacobi/scripts/benchmark.jl:44-45).  Then every product c*x is the same float
// no matter which neighbour uses it, so a cell can carry p = fl(c*x) through the generations instead of x:
//     out = ((((p_N + p_W) + p_S) + p_E) + p_C)          same additions, same order, same values
//     p_out = fl(c * out)                                  one multiplication per cell instead of five
// Results are bit-identical to Jacobi5General (kernels.hpp:267-271); the work per cell-update drops from
// 9 to 5 floating-point operations.  Grids enter and leave as ordinary values: the first pipeline level of
// the first launch of a run multiplies its (raw) inputs itself, the last level of the last launch leaves
// its sum un-multiplied.  Which launch a kernel is for is a compile-time property (FirstLaunch /
// LastLaunch), the level inside the launch comes from the sweep (at_level), so no level carries a
// run-time mode.  Needs halo_value = +0 and c > 0 (then c*halo = halo bit for bit); the runtime falls
// back to Jacobi5General otherwise.
template <bool FirstLaunch, bool LastLaunch> struct Jacobi5Uniform : public BaseTransitionFunction {
    using Cell = float;
    struct Block {
        float c;
    };

    float c;

    static Jacobi5Uniform from_params(Block const &b) {
        Jacobi5Uniform j;
        j.c = b.c;
        return j;
    }

    // level = 0 .. levels-1 inside one launch
    template <int level, int levels> STST_HD float at_level(Stencil<float, 1> const &s) const {
        constexpr bool raw_inputs = FirstLaunch && level == 0;
        constexpr bool raw_output = LastLaunch && level == levels - 1;
        float n = s[-1][0], w = s[0][-1], so = s[1][0], e = s[0][1], x = s[0][0];
        if constexpr (raw_inputs) {
            n = c * n;
            w = c * w;
            so = c * so;
            e = c * e;
            x = c * x;
        }
        const float sum = n + w + so + e + x;
        if constexpr (raw_output)
            return sum;
        else
            return c * sum;
    }

    // one generation on its own is the original expression (first and last level at once)
    STST_HD float operator()(Stencil<float, 1> const &s) const {
        return c * s[-1][0] + c * s[0][-1] + c * s[1][0] + c * s[0][1] + c * s[0][0];
    }
};

} // namespace apps

namespace hip {
template <typename F, bool SOA> struct SweepTuning;
// The dense 3 x 3 (17 flops and six lane shifts per cell) is bound by its instructions at any depth; four
// generations per launch leave the fewest warm-up rows and halo columns that HBM still hides
// (profiles/r02_tune_radius.txt, 16384^2: K=4 T=8: 1815, T=4: 1947, K=3 T=8: 1733, K=2 T=8: 1630, K=3 T=12: 1623).
template <> struct SweepTuning<apps::Jacobi<apps::JacobiVariant::General9>, false> {
    static constexpr int cells_per_lane = 4;
    static constexpr int max_generations = 4;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
};
// With 5 flops per cell the kernel is HBM bound at 8 generations per launch; 12 generations per launch on 3 cells per
// lane was the optimum of the independent-wave sweep (profiles/r01_tune_jacobi_uniform.txt).  Round 3: four stages
// per column strip leave room for 4 cells per lane at five to six waves per SIMD -- 16384^2: 5690 -> 6000 Gcell/s as
// single launches, 2048 x 16384 (the strip of an 8-GPU run): 3190 -> 4090, 1000 x 1500: 570 -> 710 at T = 12 -- and,
// with stage 0's loads pinned, for 16 generations per launch (four levels per stage, launch depths 16, 8, 4, 2, 1; a
// quarter fewer HBM bytes per generation: the timed path ran at 0.69 of the HBM peak at T = 12): two strips 5730 ->
// 5920, 2048-row strip 3950 -> 4130 (profiles/r03_tune_staged.txt).
template <bool FirstLaunch, bool LastLaunch>
struct SweepTuning<apps::Jacobi5Uniform<FirstLaunch, LastLaunch>, false> {
    static constexpr int cells_per_lane = 4;
    static constexpr int max_generations = 16;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
    // two row strips side by side: 16 chunks of ~500 rows per strip, no tapered end (6375 -> 6470)
    static constexpr int tail_permille_beside = 200;
    static constexpr bool taper_beside = false;
};
} // namespace hip
} // namespace stencil
