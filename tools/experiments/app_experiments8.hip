// Round 2: where the cooperative kernel loses its time -- the same kernel without the barrier (1), without the
// LDS traffic (2), without both (3).  Results of these are WRONG by construction; only their durations count.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using D1 = Shaped<Fdtd, 1, 6, 2, 1, true, true, 1>;
using D2 = Shaped<Fdtd, 1, 6, 2, 1, true, true, 2>;
using D3 = Shaped<Fdtd, 1, 6, 2, 1, true, true, 3>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6_coop_nobarrier", D1, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t6_coop_nolds", D2, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t6_coop_neither", D3, false);
// deeper register prefetch for the independent-wave kernel (is FDTD waiting for its rows?)
using P4 = Shaped<Fdtd, 1, 6, 4, 1, true, false>;
using P6 = Shaped<Fdtd, 1, 6, 6, 1, true, false>;
using P8 = Shaped<Fdtd, 1, 6, 8, 1, true, false>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t6p4", P4, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t6p6", P6, false);
STSTHIP_REGISTER_APP("x_fd_aos_k1t6p8", P8, false);
using Q4 = Shaped<Fdtd, 1, 4, 8, 1, true, false>;
STSTHIP_REGISTER_APP("x_fd_aos_k1t4p8", Q4, false);
// small grids: one cell per lane, shallow launches (latency of a wave's warm-up rows instead of throughput)
#include "apps/jacobi.hpp"
using J5s = Jacobi<JacobiVariant::General5>;
using S1 = Shaped<J5s, 1, 4, 4>;
using S2 = Shaped<J5s, 1, 2, 4>;
using S3 = Shaped<J5s, 2, 4, 4>;
using S4 = Shaped<J5s, 1, 8, 4>;
using S5 = Shaped<J5s, 2, 8, 4>;
STSTHIP_REGISTER_APP("x_j5_k1t4", S1, false);
STSTHIP_REGISTER_APP("x_j5_k1t2", S2, false);
STSTHIP_REGISTER_APP("x_j5_k2t4", S3, false);
STSTHIP_REGISTER_APP("x_j5_k1t8", S4, false);
STSTHIP_REGISTER_APP("x_j5_k2t8", S5, false);
