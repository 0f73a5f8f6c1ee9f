// Pipeline-shape experiments for HotSpot and FDTD (names x_hs_*, x_fd_*), timed by
// `tools/bench_apps.py experiments`.  Shaped<F, K, T, P, MINW, INTERIOR>.
// (Conway with K = 8 crashes this compiler -- an <8 x i1> vector issue -- and is left out.)
// Round-1 results are in profiles/r01_tune_shapes_apps.txt.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using F1 = Shaped<Fdtd, 1, 6, 2>;
using F2 = Shaped<Fdtd, 1, 7, 2>;
using F3 = Shaped<Fdtd, 1, 8, 2>;
using F4 = Shaped<Fdtd, 1, 5, 2>;
using F5 = Shaped<Fdtd, 1, 4, 2>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6p2", F1, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t7p2", F2, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t8p2", F3, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t5p2", F4, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t4p2", F5, false);
STSTHIP_REGISTER_APP("x_fd_soa_k1t6p2", F1, true);
STSTHIP_REGISTER_APP("x_fd_soa_k1t5p2", F4, true);
STSTHIP_REGISTER_APP("x_fd_soa_k1t8p2", F3, true);
