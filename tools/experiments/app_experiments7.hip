// Round 2: cooperative strips for the thin-cell Jacobi kernels (VALU bound: the LDS value has to be merged into
// the DPP shift, which un-fuses v_add_f32_dpp -- measured here rather than argued).
#include "app_registry.hpp"
#include "apps/jacobi.hpp"

using namespace stencil::apps;
using ststhip_detail::Shaped;
using J5 = Jacobi<JacobiVariant::General5>;
using U = Jacobi5Uniform<false, false>;

using X1 = Shaped<J5, 4, 8, 4, 1, true, true>;
STSTHIP_REGISTER_APP("x_j5_k4t8_coop", X1, false);
using X4 = Shaped<U, 3, 12, 4, 1, true, true>;
STSTHIP_REGISTER_APP("x_ju_k3t12_coop", X4, false);
using X6 = Shaped<U, 3, 12, 4, 1, true, false>;
STSTHIP_REGISTER_APP("x_ju_k3t12", X6, false);
