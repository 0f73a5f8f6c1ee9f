// Host/device annotation used by the value types that travel into HIP kernels.
#pragma once
#if defined(__HIP__)
    #define STST_HD __attribute__((host)) __attribute__((device))
    #define STST_DEVICE __attribute__((device))
#else
    #define STST_HD
    #define STST_DEVICE
#endif
