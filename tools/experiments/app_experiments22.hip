// Round 4, after the scheduling hooks became defaults: shapes that were close calls before.
//   HotSpot on planes with three / four cells per lane (the shipped shape has two: 128 / 104 columns of halo overhead)
//   general-coefficient Jacobi at 12 generations per launch (both roofs are close at 8)
// Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>; tools/bench_apps.py x_hs_*; tools/ab_staged.py general with AB_ONLY.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H2 = Shaped<Hotspot, 2, 12, 4, 1, true, 4, true>;
using H3 = Shaped<Hotspot, 3, 12, 4, 1, true, 4, true>;
using H4 = Shaped<Hotspot, 4, 12, 4, 1, true, 4, true>;
using H4b = Shaped<Hotspot, 4, 8, 4, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_hs_k2t12", H2, true);
STSTHIP_REGISTER_APP("x_hs_k3t12", H3, true);
STSTHIP_REGISTER_APP("x_hs_k4t12", H4, true);
STSTHIP_REGISTER_APP("x_hs_k4t8", H4b, true);
using J5 = Jacobi<JacobiVariant::General5>;
using G8 = Shaped<J5, 4, 8, 4, 1, true, 4, true>;
using G12 = Shaped<J5, 4, 12, 4, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_j5_t8", G8, false);
STSTHIP_REGISTER_APP("x_j5_t12", G12, false);
