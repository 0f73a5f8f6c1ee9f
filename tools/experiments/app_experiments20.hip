// Round 4: shapes of the uniform-coefficient Jacobi kernel around the shipped one (K = 4, T = 16, P = 4, four stages,
// loads pinned) that round 3's sweep (profiles/r03_tune_staged.txt, T = 12 mostly) left out: fewer / more stages at 16
// generations, deeper launches on more stages, six cells per lane.  Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>.
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using JU = Jacobi5Uniform<false, false>;
using X1 = Shaped<JU, 4, 16, 4, 1, true, 4, true>;   // the shipped shape
STSTHIP_REGISTER_APP("x_ju4_k4t16s4", X1, false);
using X2 = Shaped<JU, 4, 16, 4, 1, true, 2, true>;   // two stages of eight levels
STSTHIP_REGISTER_APP("x_ju4_k4t16s2", X2, false);
using X3 = Shaped<JU, 4, 16, 4, 1, true, 8, true>;   // eight stages of two levels
STSTHIP_REGISTER_APP("x_ju4_k4t16s8", X3, false);
using X4 = Shaped<JU, 4, 16, 2, 1, true, 4, true>; // batches of two rows
STSTHIP_REGISTER_APP("x_ju4_k4t16p2s4", X4, false);
using X5 = Shaped<JU, 4, 20, 4, 1, true, 5, true>;   // five stages of four levels
STSTHIP_REGISTER_APP("x_ju4_k4t20s5", X5, false);
using X6 = Shaped<JU, 4, 24, 4, 1, true, 4, true>;   // four stages of six levels
STSTHIP_REGISTER_APP("x_ju4_k4t24s4", X6, false);
using X7 = Shaped<JU, 6, 16, 4, 1, true, 4, true>;   // six cells per lane
STSTHIP_REGISTER_APP("x_ju4_k6t16s4", X7, false);
