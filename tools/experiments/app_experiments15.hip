// Round 3: staged sweeps of FDTD (8-word cell, two sub-iterations): AoS and the two-plane grouped layout.
// Shaped<F, K, T, P, MINW, INTERIOR, STAGES>.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using A1 = Shaped<Fdtd, 1, 6, 2, 1, true, 4>;
using A2 = Shaped<Fdtd, 1, 8, 2, 1, true, 4>;
using A3 = Shaped<Fdtd, 1, 12, 2, 1, true, 6>;
using A4 = Shaped<Fdtd, 1, 6, 2, 1, true, 6>;
using A5 = Shaped<Fdtd, 1, 12, 2, 1, true, 4>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6s4", A1, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t8s4", A2, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t12s6", A3, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t6s6", A4, false);
using B1 = Shaped<FdtdGrouped, 1, 6, 2, 1, true, 4>;
using B2 = Shaped<FdtdGrouped, 1, 8, 2, 1, true, 4>;
using B3 = Shaped<FdtdGrouped, 1, 12, 2, 1, true, 6>;
using B4 = Shaped<FdtdGrouped, 1, 12, 2, 1, true, 4>;
STSTHIP_REGISTER_APP("x_fd_grp_k1t6s4", B1, true);
STSTHIP_REGISTER_APP("x_fd_grp_k1t8s4", B2, true);
STSTHIP_REGISTER_APP("x_fd_grp_k1t12s6", B3, true);
STSTHIP_REGISTER_APP("x_fd_grp_k1t12s4", B4, true);
