// Source compatibility: applications written against the reference select the GPU backend with
// STENCILSTREAM_BACKEND_CUDA and name stencil::cuda::Grid (e.g. examples/hotspot/hotspot.cpp:30-31,
// 135-138).  On this build the GPU backend is stencil::hip.
#pragma once
#include "../hip/Grid.hpp"
namespace stencil {
namespace cuda = hip;
}
