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
