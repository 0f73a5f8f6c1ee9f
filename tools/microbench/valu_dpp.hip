// Issue rate of the DPP forms the sweep uses for its west/east neighbours, against a plain v_add_f32:
//   0: v_add_f32 (no DPP)
//   1: v_add_f32_dpp wave_shr:1   (the sweep's form: a full 64-lane shift)
//   2: v_add_f32_dpp row_shr:1    (shift inside rows of 16 lanes)
//   3: v_mov_b32_dpp wave_shr:1 followed by a plain v_add_f32 (two instructions per step)
//   4: v_add_f32_dpp quad_perm:[0,0,1,2]
// Eight independent accumulators per lane, so dependencies do not limit the issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>

#define PLAIN(acc, x) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc) : "v"(x));
#define WAVE(acc, x) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));
#define ROW(acc, x) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));
#define MOVADD(acc, x) asm volatile("v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %0, %1, %0" : "+v"(acc), "=&v"(t) : "v"(x));
#define QUAD(acc, x) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x));

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, int iters) {
    float x[8], t = 0.f;
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-9f + i * 1e-9f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (MODE == 0) PLAIN(x[i], x[(i + 3) & 7])
                else if constexpr (MODE == 1) WAVE(x[i], x[(i + 3) & 7])
                else if constexpr (MODE == 2) ROW(x[i], x[(i + 3) & 7])
                else if constexpr (MODE == 3) MOVADD(x[i], x[(i + 3) & 7])
                else QUAD(x[i], x[(i + 3) & 7])
            }
        }
    }
    float s = t;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    float *out;
    int blocks = 256 * waves_per_simd;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int steps_per_iter = 32 * 8;
    int iters = 2000000 / steps_per_iter;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 2);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double steps_per_simd = double(iters) * steps_per_iter * waves_per_simd;
    printf("%-40s waves/SIMD=%d: %.3f ns per step per SIMD\n", name, waves_per_simd, ms * 1e6 / steps_per_simd);
    hipFree(out);
}

int main() {
    for (int w : {1, 3, 4, 8}) {
        run<0>("v_add_f32", w);
        run<1>("v_add_f32_dpp wave_shr:1", w);
        run<2>("v_add_f32_dpp row_shr:1", w);
        run<3>("v_mov_b32_dpp wave_shr:1 + v_add_f32", w);
        run<4>("v_add_f32_dpp quad_perm", w);
    }
    return 0;
}
