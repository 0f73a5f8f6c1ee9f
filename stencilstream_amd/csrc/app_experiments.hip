// Pipeline-shape experiments for the headline kernel (Jacobi5General): same transition function,
// different (cells per lane, generations per launch, prefetch depth, occupancy floor).  Registered
// under their own names so one process can time them side by side (tools/tune_shapes.py).
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
using X2 = Shaped<J5, 4, 8, 2>;
STSTHIP_REGISTER_APP("x_j5_k4t8p2", X2, false);
using X3 = Shaped<J5, 4, 8, 4, 3>;
STSTHIP_REGISTER_APP("x_j5_k4t8p4w3", X3, false);
using X6 = Shaped<J5, 2, 8, 4>;
STSTHIP_REGISTER_APP("x_j5_k2t8p4", X6, false);
