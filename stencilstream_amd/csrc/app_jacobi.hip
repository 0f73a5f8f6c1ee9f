// Precompiled Jacobi sweeps (AoS == single fp32 plane).
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
STSTHIP_REGISTER_APP("jacobi1general", Jacobi<JacobiVariant::General1>, false);
STSTHIP_REGISTER_APP("jacobi2constant", Jacobi<JacobiVariant::Constant2>, false);
STSTHIP_REGISTER_APP("jacobi3constant", Jacobi<JacobiVariant::Constant3>, false);
STSTHIP_REGISTER_APP("jacobi4constant", Jacobi<JacobiVariant::Constant4>, false);
STSTHIP_REGISTER_APP("jacobi5constant", Jacobi<JacobiVariant::Constant5>, false);
STSTHIP_REGISTER_APP("jacobi4general", Jacobi<JacobiVariant::General4>, false);
STSTHIP_REGISTER_APP("jacobi5general", Jacobi<JacobiVariant::General5>, false);
STSTHIP_REGISTER_APP("jacobi9general", Jacobi<JacobiVariant::General9>, false);
