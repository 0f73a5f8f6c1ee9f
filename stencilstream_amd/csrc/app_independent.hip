// The independent-wave shapes (stages = 1: every wave runs all levels of its own column strip, no LDS, no barriers)
// of three precompiled sweeps, as they were before round 3 made the staged sweep the default.  Registered for the
// race screen of the staged kernels (tests/test_parity_gpu.py: repeated full-size runs must equal these bit for bit)
// and as the A/B baseline of profiles/r03_tune_staged.txt; nothing switches to them on its own.
#include "app_registry.hpp"
#include "apps/fdtd.hpp"
#include "apps/hotspot.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using JacobiIndependent = Shaped<Jacobi<JacobiVariant::General5>, 4, 8, 4, 1, true, 1>;
using HotspotIndependent = Shaped<Hotspot, 1, 8, 4, 1, true, 1>;
using FdtdIndependent = Shaped<Fdtd, 1, 6, 2, 1, true, 1>;
STSTHIP_REGISTER_APP("jacobi5general_independent", JacobiIndependent, false);
STSTHIP_REGISTER_APP("hotspot_independent", HotspotIndependent, true);
STSTHIP_REGISTER_APP("fdtd_coef_aos_independent", FdtdIndependent, false);
