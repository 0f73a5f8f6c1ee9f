// FDTD with two cells per lane (names x_fd_*k2*): fewer halo columns per wave (K = 1, T = 6 produces 40 of
// the 64 columns a wave loads; K = 2, T = 4 produces 112 of 128) against twice the registers per level.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
using namespace stencil::apps;
using ststhip_detail::Shaped;
using G1 = Shaped<Fdtd, 2, 4, 2>;
using G2 = Shaped<Fdtd, 2, 3, 2>;
using G3 = Shaped<Fdtd, 2, 6, 2>;
STSTHIP_REGISTER_APP("x_fd_aos_k2t4p2", G1, false);
STSTHIP_REGISTER_APP("x_fd_soa_k2t4p2", G1, true);
STSTHIP_REGISTER_APP("x_fd_aos_k2t3p2", G2, false);
STSTHIP_REGISTER_APP("x_fd_aos_k2t6p2", G3, false);

// HotSpot shapes again, after the interior() form and the chunk taper changed the balance
#include "apps/hotspot.hpp"
using H1 = Shaped<Hotspot, 2, 8, 4>;
using H2 = Shaped<Hotspot, 2, 8, 2>;
using H3 = Shaped<Hotspot, 2, 12, 4>;
using H4 = Shaped<Hotspot, 1, 12, 4>;
using H5 = Shaped<Hotspot, 3, 8, 4>;
using H6 = Shaped<Hotspot, 2, 6, 4>;
using H7 = Shaped<Hotspot, 1, 8, 4>;
STSTHIP_REGISTER_APP("x_hs_soa_k2t8p4", H1, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t8p2", H2, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t12p4", H3, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t12p4", H4, true);
STSTHIP_REGISTER_APP("x_hs_soa_k3t8p4", H5, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t6p4", H6, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t8p4", H7, true);
STSTHIP_REGISTER_APP("x_hs_aos_k2t12p4", H3, false);
STSTHIP_REGISTER_APP("x_hs_aos_k3t8p4", H5, false);
STSTHIP_REGISTER_APP("x_hs_aos_k1t12p4", H4, false);

// HotSpot in fp64 is HBM bound at T = 8 (5.1 TB/s, profiles/r01_apps_summary.json): deeper launches move fewer
// bytes per generation but produce fewer of the 64 columns a wave loads
using Hotspot64 = HotspotT<double>;
using D1 = Shaped<Hotspot64, 1, 12, 4>;
using D2 = Shaped<Hotspot64, 1, 10, 4>;
using D3 = Shaped<Hotspot64, 1, 16, 4>;
using D4 = Shaped<Hotspot64, 1, 12, 2>;
STSTHIP_REGISTER_APP("x_h64_soa_k1t12p4", D1, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t10p4", D2, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t16p4", D3, true);
STSTHIP_REGISTER_APP("x_h64_soa_k1t12p2", D4, true);
STSTHIP_REGISTER_APP("x_h64_aos_k1t12p4", D1, false);
