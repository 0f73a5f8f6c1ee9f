// Round 3: the default (independent-wave) shapes of HotSpot, FDTD and the packed Game of Life under experiment names,
// so that tools/bench_apps.py experiments runs them on the same inputs and generation counts as the staged shapes
// (bit-equality by checksum).  Shaped<F, K, T, P, MINW, INTERIOR, STAGES>.
#include "app_registry.hpp"
#include "apps/conway.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H1 = Shaped<Hotspot, 1, 8, 4, 1, true, 1>;
using H2 = Shaped<Hotspot, 2, 8, 4, 1, true, 1>;
STSTHIP_REGISTER_APP("x_hs_soa_k1t8s1", H1, true);
STSTHIP_REGISTER_APP("x_hs_aos_k2t8s1", H2, false);
using F1 = Shaped<Fdtd, 1, 6, 2, 1, true, 1>;
using F2 = Shaped<FdtdGrouped, 1, 6, 2, 1, true, 1>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6s1", F1, false);
STSTHIP_REGISTER_APP("x_fd_grp_k1t6s1", F2, true);
using C1 = Shaped<ConwayPacked, 4, 8, 4, 1, true, 1>;
STSTHIP_REGISTER_APP("x_cw_k4t8s1", C1, false);
