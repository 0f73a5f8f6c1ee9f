// Round 3: staged sweeps of the general-coefficient Jacobi kernel.  Shaped<F, K, T, P, MINW, INTERIOR, STAGES>.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
using G1 = Shaped<J5, 4, 8, 4, 1, true, 4>;
using G2 = Shaped<J5, 4, 8, 4, 1, true, 2>;
using G3 = Shaped<J5, 4, 12, 4, 1, true, 4>;
using G4 = Shaped<J5, 4, 16, 4, 1, true, 4>;
STSTHIP_REGISTER_APP("x_j5_k4t8s4", G1, false);
STSTHIP_REGISTER_APP("x_j5_k4t8s2", G2, false);
STSTHIP_REGISTER_APP("x_j5_k4t12s4", G3, false);
STSTHIP_REGISTER_APP("x_j5_k4t16s4", G4, false);
