// Pipeline-shape experiments for the packed Game of Life kernel (names x_cw_*; cells are 32-bit words of
// four cells, so `tools/bench_apps.py` sweeps a 16384 x 4096 grid of words for them).
#include "app_registry.hpp"
#include "apps/conway.hpp"
using namespace stencil::apps;
using ststhip_detail::Shaped;
using C1 = Shaped<ConwayPacked, 4, 8, 4>;
using C2 = Shaped<ConwayPacked, 4, 12, 4>;
using C3 = Shaped<ConwayPacked, 4, 16, 4>;
using C4 = Shaped<ConwayPacked, 2, 8, 4>;
using C5 = Shaped<ConwayPacked, 2, 16, 4>;
using C6 = Shaped<ConwayPacked, 3, 12, 4>;
using C7 = Shaped<ConwayPacked, 2, 24, 4>;
using C8 = Shaped<ConwayPacked, 1, 16, 4>;
STSTHIP_REGISTER_APP("x_cw_k4t8", C1, false);
STSTHIP_REGISTER_APP("x_cw_k4t12", C2, false);
STSTHIP_REGISTER_APP("x_cw_k4t16", C3, false);
STSTHIP_REGISTER_APP("x_cw_k2t8", C4, false);
STSTHIP_REGISTER_APP("x_cw_k2t16", C5, false);
STSTHIP_REGISTER_APP("x_cw_k3t12", C6, false);
STSTHIP_REGISTER_APP("x_cw_k2t24", C7, false);
STSTHIP_REGISTER_APP("x_cw_k1t16", C8, false);
