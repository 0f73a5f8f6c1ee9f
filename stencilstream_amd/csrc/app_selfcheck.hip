// Precompiled self-checking sweeps (the reference's known-answer test of the update machinery),
// radius 1 and 2, AoS and per-field planes.
#include "app_registry.hpp"
#include "apps/selfcheck.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("selfcheck1", SelfCheck<1>, false);
STSTHIP_REGISTER_APP("selfcheck1_soa", SelfCheck<1>, true);
STSTHIP_REGISTER_APP("selfcheck2", SelfCheck<2>, false);
