// Strategies for providing the time-dependent value of a transition function to a pass of the update.
//
// Interface parity: StencilStream/tdv/SinglePassStrategies.hpp:114-264 of the reference.  A strategy names a
// GlobalState<F, max_n_iterations> -- constructed once per update call from (F, iteration_offset,
// n_iterations) --, from which a KernelArgument is made per pass for (iteration offset of the pass,
// iterations of the pass), from which the executing side makes a LocalState; LocalState::
// get_time_dependent_value(i) is the value of the pass's i-th iteration.  In the reference only the FPGA backends
// take a strategy (its cpu and cuda backends evaluate on the host once per iteration); applications name one
// regardless (examples/fdtd/src/fdtd.cpp:37-45).
//
// The MI355X backend takes the strategy as the third template argument of stencil::hip::StencilUpdate and maps
// the three behaviours onto the sweep kernel's three value sources (hip/internal/Sweep.hpp, Args::tdv_table):
//   InlineStrategy              the kernel calls F::get_time_dependent_value itself, once per pipeline level and row
//                               (:114-158: "computes the value every time it is requested")
//   PrecomputeOnDeviceStrategy  a device kernel fills one table per update call, the sweeps index it
//                               (:160-207: values computed on the device before the pass consumes them)
//   PrecomputeOnHostStrategy    the host fills the table once per update call and uploads it (:209-262); the
//                               default, and the only one whose values are the host libm's bit for bit, i.e. equal
//                               to what the reference's cpu / cuda backends feed their kernels (cuda/StencilUpdate.hpp:224)
// The protocol types below are complete and host-usable (tests/cpp/api_tests.hpp walks them); sycl::handler is
// an opaque tag here since there is no command group to bind to.
#pragma once
#include "../Concepts.hpp"

#include <algorithm>
#include <cassert>
#include <memory>
#include <vector>

namespace sycl {
class handler; // only ever passed through by reference
} // namespace sycl

namespace stencil {
namespace tdv {
namespace single_pass {

template <typename T, typename F>
concept LocalState = concepts::TransitionFunction<F> && requires(T const &state, std::size_t i) {
    { state.get_time_dependent_value(i) } -> std::same_as<typename F::TimeDependentValue>;
};

enum class Kind { Inline, PrecomputeOnDevice, PrecomputeOnHost };

// Evaluates on demand: nothing is stored, every request calls the transition function.
struct InlineStrategy {
    static constexpr Kind kind = Kind::Inline;

    template <concepts::TransitionFunction F, std::size_t max_n_iterations> class GlobalState {
      public:
        using TDV = typename F::TimeDependentValue;
        GlobalState(F function, std::size_t, std::size_t) : function(function) {}

        class KernelArgument {
          public:
            using LocalState = KernelArgument;
            KernelArgument(GlobalState &global, sycl::handler &, std::size_t pass_offset, std::size_t)
                : function(global.function), pass_offset(pass_offset) {}
            KernelArgument(GlobalState &global, std::size_t pass_offset)
                : function(global.function), pass_offset(pass_offset) {}
            STST_HD TDV get_time_dependent_value(std::size_t i) const {
                return function.get_time_dependent_value(pass_offset + i);
            }

          private:
            F function;
            std::size_t pass_offset;
        };

      private:
        F function;
    };
};

// The executing side evaluates the pass's max_n_iterations values once, when it builds its LocalState.
struct PrecomputeOnDeviceStrategy {
    static constexpr Kind kind = Kind::PrecomputeOnDevice;

    template <concepts::TransitionFunction F, std::size_t max_n_iterations> class GlobalState {
      public:
        using TDV = typename F::TimeDependentValue;
        GlobalState(F function, std::size_t, std::size_t) : function(function) {}

        class KernelArgument {
          public:
            KernelArgument(GlobalState &global, sycl::handler &, std::size_t pass_offset, std::size_t)
                : function(global.function), pass_offset(pass_offset) {}
            KernelArgument(GlobalState &global, std::size_t pass_offset)
                : function(global.function), pass_offset(pass_offset) {}

            class LocalState {
              public:
                STST_HD LocalState(KernelArgument const &argument) {
                    for (std::size_t i = 0; i < max_n_iterations; i++)
                        values[i] = argument.function.get_time_dependent_value(argument.pass_offset + i);
                }
                STST_HD TDV get_time_dependent_value(std::size_t i) const { return values[i]; }

              private:
                TDV values[max_n_iterations];
            };

          private:
            F function;
            std::size_t pass_offset;
        };

      private:
        F function;
    };
};

// The host evaluates all n_iterations values of the call up front; a pass sees its window of that table.
struct PrecomputeOnHostStrategy {
    static constexpr Kind kind = Kind::PrecomputeOnHost;

    template <concepts::TransitionFunction F, std::size_t max_n_iterations> class GlobalState {
      public:
        using TDV = typename F::TimeDependentValue;
        GlobalState(F function, std::size_t iteration_offset, std::size_t n_iterations)
            : iteration_offset(iteration_offset), table(std::make_shared<std::vector<TDV>>()) {
            table->reserve(n_iterations);
            for (std::size_t i = 0; i < n_iterations; i++)
                table->push_back(function.get_time_dependent_value(iteration_offset + i));
        }
        // copies share the table (the reference's copies share the sycl::buffer, :228-230)

        std::vector<TDV> const &values() const { return *table; }
        std::size_t get_iteration_offset() const { return iteration_offset; }

        class KernelArgument {
          public:
            KernelArgument(GlobalState &global, sycl::handler &, std::size_t pass_iteration, std::size_t n_iterations)
                : KernelArgument(global, pass_iteration, n_iterations) {}
            KernelArgument(GlobalState &global, std::size_t pass_iteration, std::size_t n_iterations)
                : keep_alive(global.table) {
                assert(n_iterations <= max_n_iterations);
                const std::size_t start =
                    pass_iteration >= global.iteration_offset ? pass_iteration - global.iteration_offset : 0;
                const std::size_t end = std::min(global.table->size(), start + n_iterations);
                first = global.table->data() + std::min(start, end);
                count = end > start ? end - start : 0;
            }

            class LocalState {
              public:
                LocalState(KernelArgument const &argument) : values() {
                    for (std::size_t i = 0; i < std::min(max_n_iterations, argument.count); i++)
                        values[i] = argument.first[i];
                }
                TDV get_time_dependent_value(std::size_t i) const { return values[i]; }

              private:
                TDV values[max_n_iterations];
            };

          private:
            std::shared_ptr<std::vector<TDV>> keep_alive;
            TDV const *first;
            std::size_t count;
        };

      private:
        std::size_t iteration_offset;
        std::shared_ptr<std::vector<TDV>> table;
    };
};

template <typename T, typename F, std::size_t max_n_iterations>
concept Strategy = concepts::TransitionFunction<F> &&
                   std::constructible_from<typename T::template GlobalState<F, max_n_iterations>, F, std::size_t,
                                           std::size_t> &&
                   LocalState<typename T::template GlobalState<F, max_n_iterations>::KernelArgument::LocalState, F>;

} // namespace single_pass
} // namespace tdv
} // namespace stencil
