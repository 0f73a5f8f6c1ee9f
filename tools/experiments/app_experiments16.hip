// Round 3: staged sweeps of the packed Game of Life (four cells per 32-bit word).
#include "app_registry.hpp"
#include "apps/conway.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using A1 = Shaped<ConwayPacked, 4, 8, 4, 1, true, 4>;
using A2 = Shaped<ConwayPacked, 4, 12, 4, 1, true, 4>;
using A3 = Shaped<ConwayPacked, 4, 16, 4, 1, true, 4>;
using A4 = Shaped<ConwayPacked, 4, 8, 4, 1, true, 2>;
STSTHIP_REGISTER_APP("x_cw_k4t8s4", A1, false);
STSTHIP_REGISTER_APP("x_cw_k4t12s4", A2, false);
STSTHIP_REGISTER_APP("x_cw_k4t16s4", A3, false);
STSTHIP_REGISTER_APP("x_cw_k4t8s2", A4, false);
