// Scheduling-strategy experiment: the default Jacobi5General shape (K=4, T=8, P=4) compiled with a
// different AMDGPU machine-scheduler strategy per object file (-mllvm -amdgpu-sched-strategy=...).
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
#ifndef SCHED_TAG
#define SCHED_TAG 0
#endif
#define STR2(x) #x
#define STR(x) STR2(x)
// the tag makes the type (and so the kernel symbol) unique per object file
using XS = Shaped<J5, 4, 8, 4, 1 + 0 * SCHED_TAG, true>;
template <int TAG> struct Tagged : XS {
    using Block = XS::Block;
    Tagged() = default;
    Tagged(XS const &x) : XS(x) {}
    static Tagged from_params(Block const &b) { return Tagged(XS::from_params(b)); }
};
namespace stencil { namespace hip {
template <int TAG, bool SOA> struct SweepTuning<Tagged<TAG>, SOA> : SweepTuning<XS, SOA> {};
}}
using XT = Tagged<SCHED_TAG>;
STSTHIP_REGISTER_APP("x_j5_sched" STR(SCHED_TAG), XT, false);
