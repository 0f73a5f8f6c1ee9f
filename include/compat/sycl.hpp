#pragma once
#include "sycl/sycl.hpp"
