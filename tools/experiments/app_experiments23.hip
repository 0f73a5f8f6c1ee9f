// Round 4: what sets the 84 VGPRs (5 waves per SIMD) of the headline kernel -- the trapezoid fill's code or the steady
// loops?  x_ju5_nofill: the uniform Jacobi kernel without the trapezoid fill (tools/kernel_resources.sh on this file;
// tools/ab_staged.py uniform with AB_ONLY).  -> profiles/r04_micro_variants.txt
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
template <typename B, int V> struct Variant : public B {
    using Block = typename B::Block;
    Variant() = default;
    Variant(B const &f) : B(f) {}
    static Variant from_params(Block const &b) { return Variant(B::from_params(b)); }
};
using JU = Jacobi5Uniform<false, false>;
namespace stencil {
namespace hip {
template <typename B, bool SOA> struct SweepTuning<Variant<B, 0>, SOA> : SweepTuning<B, SOA> {
    static constexpr bool narrow_form = false;
};
template <typename B, bool SOA> struct SweepTuning<Variant<B, 1>, SOA> : SweepTuning<B, SOA> {
    static constexpr bool narrow_form = false;
    static constexpr bool trapezoid_fill = false;
};
template <typename B, bool SOA> struct SweepTuning<Variant<B, 2>, SOA> : SweepTuning<B, SOA> {
    static constexpr bool narrow_form = false;
    static constexpr int min_waves_per_simd = 6;
};
} // namespace hip
} // namespace stencil
using V0 = Variant<JU, 0>;
using V1 = Variant<JU, 1>;
using V2 = Variant<JU, 2>;
STSTHIP_REGISTER_APP("x_ju5_asis", V0, false);
STSTHIP_REGISTER_APP("x_ju5_nofill", V1, false);
STSTHIP_REGISTER_APP("x_ju5_six", V2, false);
