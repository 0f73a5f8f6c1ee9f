// Round 4: the two scheduling hooks of the staged sweep, off and on, per kernel (profiles/r04_micro_variants.txt):
//   scalar_function  the transition function in scalar registers for the whole kernel (SweepTuning<F>::scalar_function)
//   pinned_stores    a finished row's store -- LDS ring or HBM -- right behind the row (SweepTuning<F>::pinned_stores)
// Names: x_ju5_<s><p> uniform Jacobi (tools/ab_staged.py uniform, AB_ONLY), x_j5_<s><p> general Jacobi (ab_staged.py general),
// x_hs_<s><p> HotSpot planes, x_fd_<s><p> FDTD two planes, x_cw_<s><p> packed Game of Life (tools/bench_apps.py <names>).
#include "app_registry.hpp"
#include "apps/conway.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
template <typename B, bool SCALAR, bool PINNED> struct Hooked : public B {
    using Block = typename B::Block;
    Hooked() = default;
    Hooked(B const &f) : B(f) {}
    static Hooked from_params(Block const &b) { return Hooked(B::from_params(b)); }
};
using JU = Jacobi5Uniform<false, false>;
using J5 = Jacobi<JacobiVariant::General5>;
namespace stencil {
namespace hip {
template <typename B, bool S, bool P, bool SOA> struct SweepTuning<Hooked<B, S, P>, SOA> : SweepTuning<B, SOA> {
    static constexpr bool narrow_form = false;
    static constexpr bool scalar_function = S;
    static constexpr bool pinned_stores = P;
};
// (the general Jacobi kernel at its trusted depth: the hooks are about the kernel, not about the depth measurement)
template <bool S, bool P, bool SOA> struct SweepTuning<Hooked<J5, S, P>, SOA> {
    static constexpr int cells_per_lane = 4;
    static constexpr int max_generations = 8;
    static constexpr int prefetch_rows = 4;
    static constexpr bool interior_variant = true;
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = 4;
    static constexpr bool narrow_form = false;
    static constexpr bool scalar_function = S;
    static constexpr bool pinned_stores = P;
};
} // namespace hip
} // namespace stencil
#define BOTH(prefix, B, SOA)                                                                                          \
    using prefix##00 = Hooked<B, false, false>;                                                                      \
    using prefix##10 = Hooked<B, true, false>;                                                                       \
    using prefix##01 = Hooked<B, false, true>;                                                                       \
    using prefix##11 = Hooked<B, true, true>;
BOTH(U, JU, false)
STSTHIP_REGISTER_APP("x_ju5_00", U00, false);
STSTHIP_REGISTER_APP("x_ju5_10", U10, false);
STSTHIP_REGISTER_APP("x_ju5_01", U01, false);
STSTHIP_REGISTER_APP("x_ju5_11", U11, false);
BOTH(G, J5, false)
STSTHIP_REGISTER_APP("x_j5_00", G00, false);
STSTHIP_REGISTER_APP("x_j5_11", G11, false);
BOTH(H, Hotspot, true)
STSTHIP_REGISTER_APP("x_hs_00", H00, true);
STSTHIP_REGISTER_APP("x_hs_11", H11, true);
BOTH(F, FdtdGrouped, true)
STSTHIP_REGISTER_APP("x_fd_00", F00, true);
STSTHIP_REGISTER_APP("x_fd_10", F10, true);
STSTHIP_REGISTER_APP("x_fd_01", F01, true);
STSTHIP_REGISTER_APP("x_fd_11", F11, true);
BOTH(C, ConwayPacked, false)
STSTHIP_REGISTER_APP("x_cw_00", C00, false);
STSTHIP_REGISTER_APP("x_cw_01", C01, false);
