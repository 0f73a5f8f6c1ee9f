// The uniform-coefficient (product-carrying) form of Jacobi5General, precompiled: four kernels per depth (middle /
// first / last / only launch of a run).  Its own translation unit so that it compiles beside app_jacobi.hip.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

// the uniform-coefficient form of Jacobi5General (apps/jacobi.hpp); ststhip_app_run switches to it on its
// own and picks the variant per launch: middle launches, the first, the last, the only launch of a run
using U00 = stencil::apps::Jacobi5Uniform<false, false>;
using U10 = stencil::apps::Jacobi5Uniform<true, false>;
using U01 = stencil::apps::Jacobi5Uniform<false, true>;
using U11 = stencil::apps::Jacobi5Uniform<true, true>;
STSTHIP_REGISTER_APP("jacobi5uniform", U00, false);
STSTHIP_REGISTER_APP("jacobi5uniform_first", U10, false);
STSTHIP_REGISTER_APP("jacobi5uniform_last", U01, false);
STSTHIP_REGISTER_APP("jacobi5uniform_only", U11, false);
