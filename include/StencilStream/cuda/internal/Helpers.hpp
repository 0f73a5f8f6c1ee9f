// Source compatibility: stencil::cuda::internal is stencil::hip::internal (see ../Grid.hpp).
#pragma once
#include "../../hip/internal/Helpers.hpp"
#include "../Grid.hpp"
