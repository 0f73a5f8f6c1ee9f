// Pipeline-shape experiments for the headline kernels: same transition function, different
// (cells per lane, generations per launch, prefetch depth, occupancy floor).  Registered under their own
// names so one process can time them side by side (tools/tune_shapes.py).  Shaped<F, K, T, P, MINW>.
// Results of round 1: profiles/r01_tune_jacobi_uniform.txt, profiles/r01_tune_shapes_*.txt.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
using G1 = Shaped<J5, 3, 8, 4>;
using G2 = Shaped<J5, 4, 10, 4>;
using G3 = Shaped<J5, 4, 12, 4>;
using G4 = Shaped<J5, 3, 10, 4>;
using G5 = Shaped<J5, 3, 12, 4>;
using G6 = Shaped<J5, 4, 6, 4>;
STSTHIP_REGISTER_APP("x_j5_k3t8p4", G1, false);
STSTHIP_REGISTER_APP("x_j5_k4t10p4", G2, false);
STSTHIP_REGISTER_APP("x_j5_k4t12p4", G3, false);
STSTHIP_REGISTER_APP("x_j5_k3t10p4", G4, false);
STSTHIP_REGISTER_APP("x_j5_k3t12p4", G5, false);
STSTHIP_REGISTER_APP("x_j5_k4t6p4", G6, false);
// the product-carrying form (middle-launch variant; timing only)
using JU = Jacobi5Uniform<false, false>;
using U1 = Shaped<JU, 3, 12, 6>;
using U2 = Shaped<JU, 3, 12, 4>;
STSTHIP_REGISTER_APP("x_ju_k3t12p6", U1, false);
STSTHIP_REGISTER_APP("x_ju_k3t12p4", U2, false);
