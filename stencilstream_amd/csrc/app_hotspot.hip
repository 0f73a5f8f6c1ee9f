// Precompiled HotSpot sweeps: per-field planes (split cell structure) and AoS cells.
#include "app_registry.hpp"
#include "apps/hotspot.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("hotspot", Hotspot, true);
STSTHIP_REGISTER_APP("hotspot_aos", Hotspot, false);
