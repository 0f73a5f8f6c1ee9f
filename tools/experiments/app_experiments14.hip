// Round 3: staged sweeps of HotSpot (two-field cell; planes and AoS).  Shaped<F, K, T, P, MINW, INTERIOR, STAGES>.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using A1 = Shaped<Hotspot, 1, 8, 4, 1, true, 4>;
using A2 = Shaped<Hotspot, 1, 12, 4, 1, true, 4>;
using A3 = Shaped<Hotspot, 1, 16, 4, 1, true, 4>;
using A4 = Shaped<Hotspot, 2, 8, 4, 1, true, 4>;
using A5 = Shaped<Hotspot, 2, 12, 4, 1, true, 4>;
using A6 = Shaped<Hotspot, 2, 16, 4, 1, true, 4>;
using A7 = Shaped<Hotspot, 4, 12, 4, 1, true, 4>;
STSTHIP_REGISTER_APP("x_hs_soa_k1t8s4", A1, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t12s4", A2, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t16s4", A3, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t8s4", A4, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t12s4", A5, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t16s4", A6, true);
STSTHIP_REGISTER_APP("x_hs_soa_k4t12s4", A7, true);
STSTHIP_REGISTER_APP("x_hs_aos_k2t8s4", A4, false);
STSTHIP_REGISTER_APP("x_hs_aos_k2t16s4", A6, false);
