// Round 3: stage 0 of the staged sweep with its HBM loads pinned where the source has them (sched_barrier) instead of
// hoisted to the top of the batch.  Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using JU = Jacobi5Uniform<false, false>;
using U0 = Shaped<JU, 4, 12, 4, 1, true, 4, false>;
using U1 = Shaped<JU, 4, 12, 4, 1, true, 4, true>;
using U2 = Shaped<JU, 4, 16, 4, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_ju_k4t12s4f1", U0, false);
STSTHIP_REGISTER_APP("x_ju_k4t12s4pin", U1, false);
STSTHIP_REGISTER_APP("x_ju_k4t16s4pin", U2, false);
using H0 = Shaped<Hotspot, 2, 12, 4, 1, true, 4, false>;
using H1 = Shaped<Hotspot, 2, 12, 4, 1, true, 4, true>;
STSTHIP_REGISTER_APP("x_hs_soa_k2t12s4", H0, true);
STSTHIP_REGISTER_APP("x_hs_soa_k2t12s4pin", H1, true);
