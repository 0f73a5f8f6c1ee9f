// Precompiled Game of Life sweep (one byte per cell).
#include "app_registry.hpp"
#include "apps/conway.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("conway", Conway, false);
