// Pipeline-shape experiments for the headline kernel: same transition function, different
// (cells per lane, generations per launch, prefetch depth, occupancy floor).  Registered under their own
// names so one process can time them side by side (tools/tune_shapes.py).  Shaped<F, K, T, P, MINW>.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
using X2 = Shaped<J5, 4, 8, 2>;
STSTHIP_REGISTER_APP("x_j5_k4t8p2", X2, false);
using X6 = Shaped<J5, 2, 8, 4>;
STSTHIP_REGISTER_APP("x_j5_k2t8p4", X6, false);
// the product-carrying form (middle-launch variant; timing only)
using JU = Jacobi5Uniform<false, false>;
using U1 = Shaped<JU, 4, 8, 4>;
using U2 = Shaped<JU, 2, 16, 4>;
using U3 = Shaped<JU, 2, 16, 2>;
using U4 = Shaped<JU, 4, 16, 2>;
using U5 = Shaped<JU, 4, 8, 6>;
using U6 = Shaped<JU, 4, 8, 8>;
STSTHIP_REGISTER_APP("x_ju_k4t8p4", U1, false);
STSTHIP_REGISTER_APP("x_ju_k2t16p4", U2, false);
STSTHIP_REGISTER_APP("x_ju_k2t16p2", U3, false);
STSTHIP_REGISTER_APP("x_ju_k4t16p2", U4, false);
STSTHIP_REGISTER_APP("x_ju_k4t8p6", U5, false);
STSTHIP_REGISTER_APP("x_ju_k4t8p8", U6, false);
using V1 = Shaped<JU, 4, 12, 4>;
using V2 = Shaped<JU, 4, 12, 2>;
using V3 = Shaped<JU, 3, 12, 4>;
using V4 = Shaped<JU, 2, 12, 4>;
using V5 = Shaped<JU, 2, 24, 4>;
STSTHIP_REGISTER_APP("x_ju_k4t12p4", V1, false);
STSTHIP_REGISTER_APP("x_ju_k4t12p2", V2, false);
STSTHIP_REGISTER_APP("x_ju_k3t12p4", V3, false);
STSTHIP_REGISTER_APP("x_ju_k2t12p4", V4, false);
STSTHIP_REGISTER_APP("x_ju_k2t24p4", V5, false);
using W1 = Shaped<JU, 3, 10, 4>;
using W2 = Shaped<JU, 3, 14, 4>;
using W3 = Shaped<JU, 4, 10, 4>;
using W4 = Shaped<JU, 3, 12, 6>;
using W5 = Shaped<JU, 3, 12, 2>;
using W6 = Shaped<JU, 4, 14, 4>;
STSTHIP_REGISTER_APP("x_ju_k3t10p4", W1, false);
STSTHIP_REGISTER_APP("x_ju_k3t14p4", W2, false);
STSTHIP_REGISTER_APP("x_ju_k4t10p4", W3, false);
STSTHIP_REGISTER_APP("x_ju_k3t12p6", W4, false);
STSTHIP_REGISTER_APP("x_ju_k3t12p2", W5, false);
STSTHIP_REGISTER_APP("x_ju_k4t14p4", W6, false);
using U7 = Shaped<JU, 3, 16, 4>;
using U8 = Shaped<JU, 2, 16, 6>;
using U9 = Shaped<JU, 4, 16, 4>;
using U10 = Shaped<JU, 3, 16, 2>;
STSTHIP_REGISTER_APP("x_ju_k3t16p4", U7, false);
STSTHIP_REGISTER_APP("x_ju_k2t16p6", U8, false);
STSTHIP_REGISTER_APP("x_ju_k4t16p4", U9, false);
STSTHIP_REGISTER_APP("x_ju_k3t16p2", U10, false);
