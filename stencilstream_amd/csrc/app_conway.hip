// Precompiled Game of Life sweeps: one byte per cell per register ("conway"), and four cells per 32-bit
// word ("conway_packed", what ststhip_app_run("conway") uses when the width is a multiple of four).
#include "app_registry.hpp"
#include "apps/conway.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("conway", Conway, false);
STSTHIP_REGISTER_APP("conway_packed", ConwayPacked, false);
