// Jacobi transition functions, precompiled into libststhip.so.
// Arithmetic parity: examples/jacobi/kernels.hpp:34-319 of the reference -- same operand order
// per variant (N, W, S, E, C), Cell = float, radius 1, one sub-iteration, no time-dependent value.
#pragma once
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/Stencil.hpp>
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

} // namespace apps
} // namespace stencil
