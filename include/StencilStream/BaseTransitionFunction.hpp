// Defaults a transition function can inherit: no time-dependent value, radius 1, one sweep per
// generation.  Interface parity: StencilStream/BaseTransitionFunction.hpp:40-81.
#pragma once
#include "Concepts.hpp"
#include <variant>

namespace stencil {

class BaseTransitionFunction {
  public:
    using TimeDependentValue = std::monostate;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 1;

    STST_HD std::monostate get_time_dependent_value(std::size_t /*i_iteration*/) const {
        return {};
    }
};

} // namespace stencil
