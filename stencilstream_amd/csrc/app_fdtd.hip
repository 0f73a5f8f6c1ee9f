// Precompiled FDTD sweeps (coefficient resolver): per-field planes and AoS cells.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("fdtd_coef", Fdtd, true);
STSTHIP_REGISTER_APP("fdtd_coef_aos", Fdtd, false);
// two planes of 16-byte elements: the fields the update changes / the material coefficients it only copies
STSTHIP_REGISTER_APP("fdtd_coef_grouped", FdtdGrouped, true);
