// Round 3: stage 0's loads pinned where the source has them (PINNED = last parameter) against loads the scheduler
// may hoist, for the fat-cell kernels.  Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using B0 = Shaped<FdtdGrouped, 1, 8, 2, 1, true, 4, false>;
using B1 = Shaped<FdtdGrouped, 1, 8, 2, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_fd_grp_t8s4_free", B0, true);
STSTHIP_REGISTER_APP("x_fd_grp_t8s4_pinned", B1, true);
using A0 = Shaped<Fdtd, 1, 8, 2, 1, true, 4, false>;
using A1 = Shaped<Fdtd, 1, 8, 2, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_fd_aos_t8s4_free", A0, false);
STSTHIP_REGISTER_APP("x_fd_aos_t8s4_pinned", A1, false);
using H0 = Shaped<Hotspot, 2, 12, 4, 1, true, 4, false>;
using H1 = Shaped<Hotspot, 2, 12, 4, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_hs_soa_k2t12s4_free", H0, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t12s4_pinned", H1, true);
