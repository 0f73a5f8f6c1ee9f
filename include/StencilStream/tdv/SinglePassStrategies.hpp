// Time-dependent-value strategies named by applications (examples/fdtd/src/fdtd.cpp:37-45 selects one
// with TDVS_TYPE even for cpu/cuda builds).  In the reference these types parameterise the FPGA
// backends only (StencilStream/tdv/SinglePassStrategies.hpp:114-264); its cpu and cuda backends call
// F::get_time_dependent_value on the host once per iteration, and so does the MI355X backend, which
// ships the values of one launch (up to max_generations of them) as kernel arguments.  The three
// names therefore select the same behaviour here and exist for source compatibility.
#pragma once
#include "../Concepts.hpp"

namespace stencil {
namespace tdv {
namespace single_pass {

// Host-side table of the values of iterations [offset, offset + n): what one pass of a backend needs.
template <concepts::TransitionFunction F> class HostValues {
  public:
    using TDV = typename F::TimeDependentValue;
    HostValues(F const &f, std::size_t iteration_offset, std::size_t n_iterations)
        : offset(iteration_offset) {
        values.reserve(n_iterations);
        for (std::size_t i = 0; i < n_iterations; i++)
            values.push_back(f.get_time_dependent_value(iteration_offset + i));
    }
    TDV get_time_dependent_value(std::size_t i_local) const { return values[i_local]; }
    std::size_t get_iteration_offset() const { return offset; }
    std::size_t size() const { return values.size(); }

  private:
    std::size_t offset;
    std::vector<TDV> values;
};

struct InlineStrategy {
    template <concepts::TransitionFunction F, std::size_t> using GlobalState = HostValues<F>;
};
struct PrecomputeOnDeviceStrategy {
    template <concepts::TransitionFunction F, std::size_t> using GlobalState = HostValues<F>;
};
struct PrecomputeOnHostStrategy {
    template <concepts::TransitionFunction F, std::size_t> using GlobalState = HostValues<F>;
};

} // namespace single_pass
} // namespace tdv
} // namespace stencil
