// Precompiled HotSpot sweeps: per-field planes (split cell structure) and AoS cells.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("hotspot", Hotspot, true);
STSTHIP_REGISTER_APP("hotspot_aos", Hotspot, false);
// the same formula in fp64 (the reference is fp32; BASELINE.json names an fp64 configuration)
using Hotspot64 = HotspotT<double>;
STSTHIP_REGISTER_APP("hotspot_f64", Hotspot64, true);
STSTHIP_REGISTER_APP("hotspot_f64_aos", Hotspot64, false);
