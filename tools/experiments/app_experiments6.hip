// Round 2: cooperative strips (four waves of a workgroup share their edge columns through LDS) against the
// independent-wave kernels, and the deeper launches the narrower halo makes affordable.
// Shaped<F, K, T, P, MINW, INTERIOR, COOP>.  Results: profiles/r02_ab_cooperative.txt.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H64 = HotspotT<double>;

using X1 = Shaped<Fdtd, 1, 6, 2, 1, true, true>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6_coop", X1, false);
STSTHIP_REGISTER_APP("x_fd_soa_k1t6_coop", X1, true);
using X5 = Shaped<Hotspot, 1, 8, 4, 1, true, true>;
STSTHIP_REGISTER_APP("x_hs_soa_k1t8_coop", X5, true);
using X10 = Shaped<H64, 1, 8, 4, 1, true, true>;
STSTHIP_REGISTER_APP("x_h64_soa_k1t8_coop", X10, true);
