"""MI355X-native StencilUpdate backend for the StencilStream programming model.

The product is the C++ template API under include/StencilStream (drop-in for the reference's
cuda backend) on top of the C-ABI HIP library libststhip.so.  This package is the Python host
binding of that library for the precompiled transition functions: `capi` (raw ctypes), `update`
(Grid / StencilUpdate mirror of the C++ interface) and `dist` (row-strip domain decomposition
over torch.distributed).  There is no CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
