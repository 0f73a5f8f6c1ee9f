// Precompiled Jacobi sweeps (AoS == single fp32 plane).
// Built twice: with -ffp-contract=off (names below, bit-identical to the reference's cpu backend) and with
// -ffp-contract=fast -DSTSTHIP_FMA_FLAVOUR (names + "_fma": multiply-adds fused, as the reference's own
// GPU compiler does by default; results within a few ulp per generation).
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

#ifdef STSTHIP_FMA_FLAVOUR
#define NAME(n) n "_fma"
namespace {
// distinct types so that both flavours can live in one library
template <stencil::apps::JacobiVariant V> struct Fused : stencil::apps::Jacobi<V> {
    using Block = typename stencil::apps::Jacobi<V>::Block;
    Fused() = default;
    Fused(stencil::apps::Jacobi<V> const &j) : stencil::apps::Jacobi<V>(j) {}
    static Fused from_params(Block const &b) { return Fused(stencil::apps::Jacobi<V>::from_params(b)); }
};
} // namespace
#define KERNEL(V) Fused<stencil::apps::JacobiVariant::V>
#else
#define NAME(n) n
#define KERNEL(V) stencil::apps::Jacobi<stencil::apps::JacobiVariant::V>
#endif

using K1 = KERNEL(General1);
using K2 = KERNEL(Constant2);
using K3 = KERNEL(Constant3);
using K4 = KERNEL(Constant4);
using K5 = KERNEL(Constant5);
using K6 = KERNEL(General4);
using K7 = KERNEL(General5);
using K8 = KERNEL(General9);
// JACOBI_SET = 0 .. 3: the variants are compiled in four translation units side by side (each one-word variant is a
// family of five depths in two lane widths: one unit took a quarter of an hour)
#ifndef JACOBI_SET
#define JACOBI_SET 0
#endif
#ifndef STSTHIP_FMA_FLAVOUR
#if JACOBI_SET == 1
STSTHIP_REGISTER_APP(NAME("jacobi1general"), K1, false);
STSTHIP_REGISTER_APP(NAME("jacobi2constant"), K2, false);
#elif JACOBI_SET == 2
STSTHIP_REGISTER_APP(NAME("jacobi3constant"), K3, false);
STSTHIP_REGISTER_APP(NAME("jacobi4constant"), K4, false);
#elif JACOBI_SET == 3
STSTHIP_REGISTER_APP(NAME("jacobi5constant"), K5, false);
STSTHIP_REGISTER_APP(NAME("jacobi4general"), K6, false);
#else
STSTHIP_REGISTER_APP(NAME("jacobi9general"), K8, false);
#endif
#endif
#if JACOBI_SET == 0
STSTHIP_REGISTER_APP(NAME("jacobi5general"), K7, false);
#endif
