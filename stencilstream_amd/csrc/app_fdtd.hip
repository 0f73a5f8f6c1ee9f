// Precompiled FDTD sweeps (coefficient resolver): per-field planes and AoS cells.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("fdtd_coef", Fdtd, true);
STSTHIP_REGISTER_APP("fdtd_coef_aos", Fdtd, false);
