// MFMA A/B for dense (2R+1)^2 convolutions (Jacobi9General, a radius-2 5x5 Jacobi): one generation of
//     out[r][c] = sum_{dr,dc} coef[dr][dc] * in[r+dr][c+dc]          (halo 0)
// as banded-Toeplitz matrix products on the matrix cores,
//     Out(16 x 16) = sum_dr  In[rows + dr, cols - R .. cols + 15 + R] (16 x (16+2R))  x  Band_dr ((16+2R) x 16),
//     Band_dr[j][c] = coef[dr][j - c] for 0 <= j - c <= 2R, else 0,
// with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact products, fused accumulation in k order -- NOT the
// operation order of the reference's expression, so parity is by tolerance, like the "_fma" flavour).
// A wave owns a 16 x 16 output tile per step; its (16+2R) x (16+2R) input patch is staged in LDS once and read as
// the A operands; the Band matrices sit in registers.
//
// What it answers (north_star: "MFMA only where the transition function reduces to a dense 3x3/5x5 convolution,
// the choice evidenced by" measurement): the time of this kernel and its MFMA-issue floor against the VALU sweep of
// libststhip.so on the same grid.  Prints one JSON line per radius.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

typedef float float4_t __attribute__((ext_vector_type(4)));

template <int R> struct Shape {
    static constexpr int D = 2 * R + 1;
    static constexpr int KDIM = 16 + 2 * R;          // columns of the input patch a tile needs
    static constexpr int STEPS = (KDIM + 3) / 4;     // k-steps of 4
    static constexpr int PATCH_ROWS = 16 + 2 * R;
    static constexpr int PATCH_PITCH = 4 * STEPS + 1; // +1: rows land in different LDS banks
};

template <int R>
__global__ void __launch_bounds__(256) conv_mfma(const float *__restrict__ in, float *__restrict__ out, int H, int W,
                                                 const float *__restrict__ coef, int tiles_per_wave) {
    using S = Shape<R>;
    __shared__ float patches[4][S::PATCH_ROWS * S::PATCH_PITCH];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float *patch = patches[wib];
    const int tile_cols = (W + 15) / 16;
    const long n_tiles = long((H + 15) / 16) * tile_cols;

    // B operands: lane l holds Band[k = 4 s + l / 16][n = l % 16] for every (dr, s)
    float band[S::D][S::STEPS];
    const int n = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int dr = 0; dr < S::D; dr++)
#pragma unroll
        for (int s = 0; s < S::STEPS; s++) {
            const int j = 4 * s + kk, tap = j - n;
            band[dr][s] = (tap >= 0 && tap < S::D && j < S::KDIM) ? coef[dr * S::D + tap] : 0.0f;
        }

    const long first = (long(blockIdx.x) * 4 + wib) * tiles_per_wave;
    for (long t = first; t < first + tiles_per_wave && t < n_tiles; t++) {
        const int r0 = int(t / tile_cols) * 16, c0 = int(t % tile_cols) * 16;
        // stage the patch: rows r0-R .. r0+15+R, columns c0-R .. c0-R+4*STEPS-1 (zero outside the grid)
        for (int e = lane; e < S::PATCH_ROWS * 4 * S::STEPS; e += 64) {
            const int pr = e / (4 * S::STEPS), pc = e % (4 * S::STEPS);
            const int r = r0 - R + pr, c = c0 - R + pc;
            patch[pr * S::PATCH_PITCH + pc] = (r >= 0 && r < H && c >= 0 && c < W) ? in[long(r) * W + c] : 0.0f;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
        // A operand: lane l holds In[row = l % 16][k = l / 16] of the k-step
#pragma unroll
        for (int dr = 0; dr < S::D; dr++)
#pragma unroll
            for (int s = 0; s < S::STEPS; s++) {
                const float a = patch[(n + dr) * S::PATCH_PITCH + 4 * s + kk];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, band[dr][s], acc, 0, 0, 0);
            }
        // D: lane l, register i -> out[row = 4 (l / 16) + i][col = l % 16]
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + 4 * kk + i, c = c0 + n;
            if (r < H && c < W)
                out[long(r) * W + c] = acc[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int R> static void run(int N, int reps) {
    using S = Shape<R>;
    std::vector<float> coef(S::D * S::D);
    for (int i = 0; i < S::D * S::D; i++)
        coef[i] = (1.0f + 0.01f * float(i % 5)) / float(S::D * S::D);
    float *d_in, *d_out, *d_coef;
    CHECK(hipMalloc(&d_in, size_t(N) * N * 4));
    CHECK(hipMalloc(&d_out, size_t(N) * N * 4));
    CHECK(hipMalloc(&d_coef, coef.size() * 4));
    CHECK(hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> host(size_t(N) * N);
    unsigned long long state = 0x5EED;
    for (auto &v : host) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        v = float((state >> 40) & 0xFFFF) / 65536.0f;
    }
    CHECK(hipMemcpy(d_in, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const long n_tiles = long((N + 15) / 16) * ((N + 15) / 16);
    const int tiles_per_wave = 8;
    const unsigned blocks = unsigned((n_tiles + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave));
    hipLaunchKernelGGL(conv_mfma<R>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, N, N, d_coef, tiles_per_wave);
    CHECK(hipDeviceSynchronize());
    // check a few thousand cells against the expression evaluated in double
    std::vector<float> got(size_t(N) * N);
    CHECK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int probe = 0; probe < 4000; probe++) {
        const int r = (probe * 7919) % N, c = (probe * 104729) % N;
        double want = 0;
        for (int dr = -R; dr <= R; dr++)
            for (int dc = -R; dc <= R; dc++) {
                const int rr = r + dr, cc = c + dc;
                if (rr >= 0 && rr < N && cc >= 0 && cc < N)
                    want += double(coef[(dr + R) * S::D + dc + R]) * double(host[size_t(rr) * N + cc]);
            }
        worst = std::fmax(worst, std::fabs(want - double(got[size_t(r) * N + c])));
    }
    // corners and edges too
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < reps; rep++) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(conv_mfma<R>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, N, N, d_coef, tiles_per_wave);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = std::fmin(best, ms);
    }
    // v_mfma_f32_16x16x4_f32 issues every 32 cycles per SIMD (MI355X_MICROARCH.md); 1024 SIMDs at 2.4 GHz
    const double mfma_per_tile = double(S::D) * S::STEPS;
    const double floor_ms = double(n_tiles) * mfma_per_tile * 32.0 / (1024.0 * 2.4e9) * 1e3;
    std::printf("{\"kernel\": \"mfma_conv_%dx%d\", \"grid\": %d, \"generations_per_launch\": 1, \"ms\": %.4f, "
                "\"Gcell_updates_per_s\": %.1f, \"mfma_per_256_cells\": %.0f, \"mfma_issue_floor_ms\": %.4f, "
                "\"mfma_issue_floor_Gcell_per_s\": %.1f, \"max_abs_error_vs_fp64\": %.3g, "
                "\"useful_fraction_of_mfma_flops\": %.3f}\n",
                S::D, S::D, N, best, double(N) * N / (best * 1e-3) / 1e9, mfma_per_tile, floor_ms,
                double(N) * N / (floor_ms * 1e-3) / 1e9, worst,
                double(S::D * S::D * 256) / (mfma_per_tile * 16 * 16 * 4));
    CHECK(hipFree(d_in));
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_coef));
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 16384;
    run<1>(N, 10);
    run<2>(N, 10);
    return 0;
}
