// What HBM delivers for the sweep's access pattern, without any arithmetic: a copy in which every wave streams down a
// column strip -- one row of 64 x 16 B = 1 KiB per step, rows one grid pitch apart, P rows in flight -- against a flat
// copy of the same bytes.  Answers whether "4.2-4.9 TB/s" of the HBM-leaning sweeps is the memory
// system's rate for this pattern or something the kernels leave on the table.
// Prints one JSON line per variant.  Build: hipcc --offload-arch=gfx950 -O3.
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

typedef float float4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) flat_copy(const float4_t *__restrict__ in, float4_t *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
        out[i] = in[i];
}

// wave w: strip = w % n_strips (64 float4 = 256 floats wide), chunk = w / n_strips (chunk_rows rows)
template <int P>
__global__ void __launch_bounds__(256) strip_copy(const float4_t *__restrict__ in, float4_t *__restrict__ out, int rows,
                                                  int pitch4, int n_strips, int chunk_rows) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int strip = wave % n_strips, chunk = wave / n_strips;
    const int r0 = chunk * chunk_rows;
    if (r0 >= rows)
        return;
    const int r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    const size_t col = size_t(strip) * 64 + lane;
    float4_t pre[P];
#pragma unroll
    for (int u = 0; u < P; u++) {
        const int r = r0 + u < rows ? r0 + u : rows - 1;
        pre[u] = in[size_t(r) * pitch4 + col];
    }
    for (int r = r0; r < r1; r += P) {
#pragma unroll
        for (int u = 0; u < P; u++) {
            const float4_t v = pre[u];
            const int next = r + u + P < rows ? r + u + P : rows - 1;
            pre[u] = in[size_t(next) * pitch4 + col];
            if (r + u < r1)
                out[size_t(r + u) * pitch4 + col] = v;
        }
    }
}

template <typename Launch> static float time_ms(Launch launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(a, 0));
        launch();
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 16384; // N x N floats
    const size_t bytes = size_t(N) * N * 4;
    float4_t *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    const size_t n4 = bytes / 16;
    const double gb = 2.0 * bytes / 1e9; // read + write
    for (int blocks : {2048, 8192, 65536}) {
        const float ms = time_ms([&] { hipLaunchKernelGGL(flat_copy, dim3(blocks), dim3(256), 0, 0, in, out, n4); }, 10);
        std::printf("{\"kernel\": \"flat_copy\", \"grid\": %d, \"blocks\": %d, \"ms\": %.4f, \"TB_per_s\": %.2f}\n", N, blocks, ms,
                    gb / ms);
    }
    const int pitch4 = N / 4, n_strips = N / 256;
    for (int chunk_rows : {64, 133, 256, 512}) {
        const int chunks = (N + chunk_rows - 1) / chunk_rows;
        const unsigned blocks = unsigned((size_t(n_strips) * chunks + 3) / 4);
        const float m2 = time_ms([&] { hipLaunchKernelGGL(strip_copy<2>, dim3(blocks), dim3(256), 0, 0, in, out, N, pitch4, n_strips, chunk_rows); }, 10);
        const float m4 = time_ms([&] { hipLaunchKernelGGL(strip_copy<4>, dim3(blocks), dim3(256), 0, 0, in, out, N, pitch4, n_strips, chunk_rows); }, 10);
        const float m8 = time_ms([&] { hipLaunchKernelGGL(strip_copy<8>, dim3(blocks), dim3(256), 0, 0, in, out, N, pitch4, n_strips, chunk_rows); }, 10);
        std::printf("{\"kernel\": \"strip_copy\", \"grid\": %d, \"chunk_rows\": %d, \"waves\": %u, \"TB_per_s_P2\": %.2f, "
                    "\"TB_per_s_P4\": %.2f, \"TB_per_s_P8\": %.2f}\n",
                    N, chunk_rows, blocks * 4, gb / m2, gb / m4, gb / m8);
    }
    CHECK(hipFree(in));
    CHECK(hipFree(out));
    return 0;
}
