// Source compatibility: stencil::cuda::StencilUpdate is stencil::hip::StencilUpdate (see Grid.hpp).
#pragma once
#include "../hip/StencilUpdate.hpp"
#include "Grid.hpp"
