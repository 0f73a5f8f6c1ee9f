// Pipeline-shape experiments for HotSpot, FDTD and Conway (names x_hs_*, x_fd_*, x_cw_*), timed by
// `tools/bench_apps.py experiments`.  Shaped<F, K, T, P, MINW, INTERIOR>.
// (Conway with K = 8 crashes this compiler -- an <8 x i1> vector issue -- and is left out.)
// Round-1 results are in profiles/r01_tune_shapes_apps.txt.
#include "app_registry.hpp"
#include "apps/conway.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using H1 = Shaped<Hotspot, 2, 8, 4>;
using H2 = Shaped<Hotspot, 2, 8, 2>;
using H3 = Shaped<Hotspot, 4, 8, 4>;
using H4 = Shaped<Hotspot, 1, 8, 4>;
using H5 = Shaped<Hotspot, 2, 8, 6>;
STSTHIP_REGISTER_APP("x_hs_soa_k2t8p4", H1, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t8p2", H2, true);
STSTHIP_REGISTER_APP("x_hs_soa_k4t8p4", H3, true);
STSTHIP_REGISTER_APP("x_hs_soa_k1t8p4", H4, true);
STSTHIP_REGISTER_APP("x_hs_aos_k2t8p4", H1, false);
STSTHIP_REGISTER_APP("x_hs_aos_k2t8p2", H2, false);
STSTHIP_REGISTER_APP("x_hs_aos_k4t8p4", H3, false);
STSTHIP_REGISTER_APP("x_hs_aos_k1t8p4", H4, false);
STSTHIP_REGISTER_APP("x_hs_aos_k2t8p6", H5, false);
using F2 = Shaped<Fdtd, 1, 4, 2>;
using F3 = Shaped<Fdtd, 1, 4, 4>;
STSTHIP_REGISTER_APP("x_fd_soa_k1t4p2", F2, true);
STSTHIP_REGISTER_APP("x_fd_soa_k1t4p4", F3, true);
STSTHIP_REGISTER_APP("x_fd_aos_k1t4p2", F2, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t4p4", F3, false);
