// Occupancy-floor experiments for the product-carrying Jacobi kernel: Shaped<F, K, T, P, MINW> with MINW = 4
// waves per SIMD (__launch_bounds__ second argument).  Results: profiles/r01_tune_jacobi_uniform_occupancy.txt
#include "app_registry.hpp"
#include "apps/jacobi.hpp"
using namespace stencil::apps;
using ststhip_detail::Shaped;
using JU = Jacobi5Uniform<false, false>;
using A1 = Shaped<JU, 3, 12, 4, 4>;
using A2 = Shaped<JU, 3, 10, 4, 4>;
using A3 = Shaped<JU, 3, 8, 4, 4>;
using A4 = Shaped<JU, 3, 10, 4, 1>;
using A5 = Shaped<JU, 2, 12, 4, 4>;
using A6 = Shaped<JU, 2, 16, 4, 4>;
STSTHIP_REGISTER_APP("x_ju_k3t12p4w4", A1, false);
STSTHIP_REGISTER_APP("x_ju_k3t10p4w4", A2, false);
STSTHIP_REGISTER_APP("x_ju_k3t8p4w4", A3, false);
STSTHIP_REGISTER_APP("x_ju_k3t10p4w1", A4, false);
STSTHIP_REGISTER_APP("x_ju_k2t12p4w4", A5, false);
STSTHIP_REGISTER_APP("x_ju_k2t16p4w4", A6, false);
// wider lanes with an occupancy floor of three waves (after the chunk taper made the end of a launch cheap)
using B1 = Shaped<JU, 4, 12, 4, 3>;
using B2 = Shaped<JU, 4, 12, 4, 1>;
using B3 = Shaped<JU, 4, 10, 4, 3>;
using B4 = Shaped<JU, 5, 8, 4, 3>;
using B5 = Shaped<JU, 4, 16, 4, 2>;
using B6 = Shaped<JU, 3, 12, 4, 1>;
STSTHIP_REGISTER_APP("x_ju_k4t12p4w3", B1, false);
STSTHIP_REGISTER_APP("x_ju_k4t12p4w1", B2, false);
STSTHIP_REGISTER_APP("x_ju_k4t10p4w3", B3, false);
STSTHIP_REGISTER_APP("x_ju_k5t8p4w3", B4, false);
STSTHIP_REGISTER_APP("x_ju_k4t16p4w2", B5, false);
STSTHIP_REGISTER_APP("x_ju_k3t12p4w1", B6, false);
