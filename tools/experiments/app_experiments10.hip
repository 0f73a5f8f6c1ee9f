// Round 2: persistent waves (one residency round of waves claims fine row chunks and continues into the chunk below
// without re-warming the pipeline) against the chunked launches.  Shaped<F, K, T, P, MINW, INTERIOR, COOP, DBG, PERSIST>.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using U = Jacobi5Uniform<false, false>;
using J5 = Jacobi<JacobiVariant::General5>;
using P1 = Shaped<U, 3, 12, 4, 1, true, false, 0, true>;
using P2 = Shaped<J5, 4, 8, 4, 1, true, false, 0, true>;
using P3 = Shaped<Hotspot, 1, 8, 4, 1, true, false, 0, true>;
using P4 = Shaped<FdtdGrouped, 1, 6, 2, 1, true, false, 0, true>;
STSTHIP_REGISTER_APP("x_ju_k3t12_persist", P1, false);
STSTHIP_REGISTER_APP("x_j5_k4t8_persist", P2, false);
STSTHIP_REGISTER_APP("x_hs_soa_k1t8_persist", P3, true);
STSTHIP_REGISTER_APP("x_fdg_k1t6_persist", P4, true);
using Q1 = Shaped<U, 3, 12, 4, 1, true, false, 0, false>;
STSTHIP_REGISTER_APP("x_ju_k3t12", Q1, false);
