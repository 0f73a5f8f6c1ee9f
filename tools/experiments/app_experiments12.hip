// Round 3: staged sweeps (the waves of a workgroup as a software pipeline over the levels of one column strip,
// rows handed from stage to stage through LDS) against the independent-wave shapes.
// Shaped<F, K, T, P, MINW, INTERIOR, STAGES>.  Results: profiles/r03_tune_staged.txt.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using JU = Jacobi5Uniform<false, false>;
using J5 = Jacobi<JacobiVariant::General5>;
using U0 = Shaped<JU, 3, 12, 4, 1, true, 1>;
using U1 = Shaped<JU, 3, 12, 4, 1, true, 4>;
using U2 = Shaped<JU, 4, 12, 4, 1, true, 4>;
using U3 = Shaped<JU, 4, 12, 4, 1, true, 2>;
using U4 = Shaped<JU, 4, 12, 4, 1, true, 6>;
using U5 = Shaped<JU, 4, 12, 2, 1, true, 4>;
using U6 = Shaped<JU, 4, 16, 4, 1, true, 4>;
using U7 = Shaped<JU, 2, 12, 4, 1, true, 4>;
using U8 = Shaped<JU, 4, 12, 4, 1, true, 3>;
using U9 = Shaped<JU, 8, 12, 4, 1, true, 4>;
STSTHIP_REGISTER_APP("x_ju_k3t12s1", U0, false);
STSTHIP_REGISTER_APP("x_ju_k3t12s4", U1, false);
STSTHIP_REGISTER_APP("x_ju_k4t12s4", U2, false);
STSTHIP_REGISTER_APP("x_ju_k4t12s2", U3, false);
STSTHIP_REGISTER_APP("x_ju_k4t12s6", U4, false);
STSTHIP_REGISTER_APP("x_ju_k4t12p2s4", U5, false);
STSTHIP_REGISTER_APP("x_ju_k4t16s4", U6, false);
STSTHIP_REGISTER_APP("x_ju_k2t12s4", U7, false);
STSTHIP_REGISTER_APP("x_ju_k4t12s3", U8, false);
STSTHIP_REGISTER_APP("x_ju_k8t12s4", U9, false);
