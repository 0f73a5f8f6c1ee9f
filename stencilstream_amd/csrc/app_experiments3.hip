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
