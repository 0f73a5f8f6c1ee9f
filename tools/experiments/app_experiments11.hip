// Round 2: deeper launches for HotSpot (fewer HBM bytes per generation against more halo columns per wave).
#include "app_registry.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H64 = HotspotT<double>;
using A1 = Shaped<H64, 1, 12, 4>;
using A2 = Shaped<H64, 1, 12, 2>;
using A3 = Shaped<H64, 1, 16, 2>;
using A4 = Shaped<H64, 1, 6, 4>;
STSTHIP_REGISTER_APP("x_h64_soa_k1t12p4", A1, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t12p2", A2, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t16p2", A3, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t6p4", A4, true);
using B1 = Shaped<Hotspot, 1, 12, 4>;
using B2 = Shaped<Hotspot, 1, 16, 4>;
STSTHIP_REGISTER_APP("x_hs_soa_k1t12p4", B1, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t16p4", B2, true);
