// Round 2: HotSpot shapes -- deeper launches (fewer HBM bytes per generation against more halo columns per wave) and,
// for fp64, two cells per lane (16-byte accesses, half the halo columns, twice the registers).
#include "app_registry.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H64 = HotspotT<double>;
using A1 = Shaped<H64, 1, 12, 4>;
using A4 = Shaped<H64, 1, 6, 4>;
using A5 = Shaped<H64, 2, 8, 2>;
using A6 = Shaped<H64, 2, 8, 4>;
using A7 = Shaped<H64, 2, 6, 2>;
using A8 = Shaped<H64, 2, 4, 4>;
STSTHIP_REGISTER_APP("x_h64_soa_k1t12p4", A1, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t6p4", A4, true);
STSTHIP_REGISTER_APP("x_h64_soa_k2t8p2", A5, true);
STSTHIP_REGISTER_APP("x_h64_soa_k2t8p4", A6, true);
STSTHIP_REGISTER_APP("x_h64_soa_k2t6p2", A7, true);
STSTHIP_REGISTER_APP("x_h64_soa_k2t4p4", A8, true);
using B1 = Shaped<Hotspot, 1, 12, 4>;
STSTHIP_REGISTER_APP("x_hs_soa_k1t12p4", B1, true);
