// Round 4 (late): FDTD on two planes after the column hint (one halo column per generation and side instead of two): the
// shipped shape holds one cell per lane -- 48 of 64 columns of a wave are useful at T = 8; two cells per lane would make it
// 112 of 128 (-14 % bytes) at twice the register window; T = 12 at one cell per lane 40 of 64 columns but fewer passes.
// Shaped<F, K, T, P, MINW, INTERIOR, STAGES, PINNED>; tools/bench_apps.py x_fd_*.  -> profiles/r04_shapes.txt
#include "app_registry.hpp"
#include "apps/fdtd.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using F1 = Shaped<FdtdGrouped, 1, 8, 2, 1, true, 4, false>;
using F2 = Shaped<FdtdGrouped, 2, 8, 2, 1, true, 4, false>;
using F1t12 = Shaped<FdtdGrouped, 1, 12, 2, 1, true, 4, false>;
STSTHIP_REGISTER_APP("x_fd_k1t8", F1, true);
STSTHIP_REGISTER_APP("x_fd_k2t8", F2, true);
STSTHIP_REGISTER_APP("x_fd_k1t12", F1t12, true);
