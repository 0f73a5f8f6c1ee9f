// Alternative launch schemes of precompiled sweeps that were built, measured and not adopted.
// Cooperative-strip forms of two precompiled sweeps (Sweep<..., COOP = true>: the four waves of a workgroup take
// adjacent strips and exchange their edge columns through LDS).  Bit-identical to "jacobi5general" / "hotspot";
// measured slower or equal on MI355X (profiles/r02_ab_cooperative.txt), so nothing switches to them on its own --
// they are registered so that the parity tests keep the cooperative path honest.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using JacobiCoop = Shaped<Jacobi<JacobiVariant::General5>, 4, 8, 4, 1, true, true>;
using HotspotCoop = Shaped<Hotspot, 1, 8, 4, 1, true, true>;
STSTHIP_REGISTER_APP("jacobi5general_coop", JacobiCoop, false);
// persistent waves (a wave continues into the row chunk below without re-warming its pipeline): bit-identical,
// measured 2-11 % slower than the chunked launches (profiles/r02_ab_persistent.txt); kept under the parity tests
using JacobiPersistent = Shaped<Jacobi<JacobiVariant::General5>, 4, 8, 4, 1, true, false, 0, true>;
STSTHIP_REGISTER_APP("jacobi5general_persistent", JacobiPersistent, false);
STSTHIP_REGISTER_APP("hotspot_coop", HotspotCoop, true);
