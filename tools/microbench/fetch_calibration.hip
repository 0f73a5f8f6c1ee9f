// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths the sweeps use (MI355X_MICROARCH.md, HBM:
// "on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B per lane) ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Three copies of
// the same 1 GiB, streamed row-wise by waves the way a sweep's stage 0 loads and its last stage stores: 4, 8 and 16
// bytes per lane and access.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; the true byte count per
// launch is printed, tools/summarize_legs_profile.py turns the two into factors per width.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calibration.hip -o build/fetch_calibration
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

typedef float f1;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// (plain names, not template instantiations: the summariser reads the width from the kernel's name)
#define COPY_KERNELS(W, V)                                                                                             \
    __global__ void __launch_bounds__(256) calibration_copy_b##W##_nontemporal(const V *__restrict__ in, V *__restrict__ out, size_t n) { \
        for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)                  \
            __builtin_nontemporal_store(in[i], &out[i]);                                                               \
    }                                                                                                                  \
    __global__ void __launch_bounds__(256) calibration_copy_b##W##_plain(const V *__restrict__ in, V *__restrict__ out, size_t n) { \
        for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)                  \
            out[i] = in[i];                                                                                            \
    }
COPY_KERNELS(4, f1)
COPY_KERNELS(8, f2)
COPY_KERNELS(16, f4)

int main() {
    const size_t bytes = size_t(1) << 30;
    void *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calibration_copy_b4_nontemporal, dim3(16384), dim3(256), 0, 0, (const f1 *)in, (f1 *)out, bytes / 4);
        hipLaunchKernelGGL(calibration_copy_b8_nontemporal, dim3(16384), dim3(256), 0, 0, (const f2 *)in, (f2 *)out, bytes / 8);
        hipLaunchKernelGGL(calibration_copy_b16_nontemporal, dim3(16384), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, bytes / 16);
        hipLaunchKernelGGL(calibration_copy_b4_plain, dim3(16384), dim3(256), 0, 0, (const f1 *)in, (f1 *)out, bytes / 4);
        hipLaunchKernelGGL(calibration_copy_b8_plain, dim3(16384), dim3(256), 0, 0, (const f2 *)in, (f2 *)out, bytes / 8);
        hipLaunchKernelGGL(calibration_copy_b16_plain, dim3(16384), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, bytes / 16);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("{\"calibration_bytes_read_per_launch\": %zu, \"calibration_bytes_written_per_launch\": %zu}\n", bytes, bytes);
    return 0;
}
