// Round 2: shapes for the dense stencils -- Jacobi9General (3 x 3, radius 1) and Jacobi25 (5 x 5, radius 2).
// Results: profiles/r02_tune_radius.txt.  Shaped<F, K, T, P>.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J9 = Jacobi<JacobiVariant::General9>;
using A1 = Shaped<J9, 4, 8, 4>;
using A2 = Shaped<J9, 3, 8, 4>;
using A3 = Shaped<J9, 2, 8, 4>;
using A4 = Shaped<J9, 4, 4, 4>;
using A5 = Shaped<J9, 3, 12, 4>;
STSTHIP_REGISTER_APP("x_j9_k4t8", A1, false);
STSTHIP_REGISTER_APP("x_j9_k3t8", A2, false);
STSTHIP_REGISTER_APP("x_j9_k2t8", A3, false);
STSTHIP_REGISTER_APP("x_j9_k4t4", A4, false);
STSTHIP_REGISTER_APP("x_j9_k3t12", A5, false);
using B1 = Shaped<Jacobi25, 4, 8, 4>;
using B2 = Shaped<Jacobi25, 4, 4, 4>;
using B3 = Shaped<Jacobi25, 2, 8, 4>;
using B4 = Shaped<Jacobi25, 2, 4, 4>;
using B5 = Shaped<Jacobi25, 4, 2, 4>;
using B6 = Shaped<Jacobi25, 3, 4, 4>;
STSTHIP_REGISTER_APP("x_j25_k4t8", B1, false);
STSTHIP_REGISTER_APP("x_j25_k4t4", B2, false);
STSTHIP_REGISTER_APP("x_j25_k2t8", B3, false);
STSTHIP_REGISTER_APP("x_j25_k2t4", B4, false);
STSTHIP_REGISTER_APP("x_j25_k4t2", B5, false);
STSTHIP_REGISTER_APP("x_j25_k3t4", B6, false);
