// Precompiled dense 5 x 5 Jacobi of radius 2 (apps/jacobi.hpp, Jacobi25): an extra beyond the reference's
// applications, the tuned radius > 1 kernel of SURVEY 8(f)4.  Shape from profiles/r02_tune_radius.txt.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("jacobi25general", Jacobi25, false);
