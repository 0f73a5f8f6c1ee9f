// Does the Jacobi sweep's instruction mix issue slower than plain VALU ops?  Variants:
//   0: v_mul(sgpr, vgpr) -> v_add chains like the sweep, short loop body (128 instructions)
//   1: the same, long straight-line body (4096 instructions per iteration: instruction-fetch bound?)
//   2: long body, all-VGPR operands
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEP_S(acc, x) asm volatile("v_mul_f32 %0, %2, %3\n v_add_f32 %1, %0, %1" : "=&v"(t), "+v"(acc) : "s"(c), "v"(x));
#define STEP_V(acc, x) asm volatile("v_mul_f32 %0, %2, %3\n v_add_f32 %1, %0, %1" : "=&v"(t), "+v"(acc) : "v"(cv), "v"(x));

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, float c, int iters) {
    float x[8], t;
    float cv = c + threadIdx.x * 0.f;
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) STEP_S(x[i], x[(i + 3) & 7])
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 256; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) STEP_S(x[i], x[(i + 3) & 7])
            }
        } else {
#pragma unroll
            for (int r = 0; r < 256; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) STEP_V(x[i], x[(i + 3) & 7])
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    float *out;
    int blocks = 256 * waves_per_simd;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int per_iter = (MODE == 0 ? 8 : 256) * 8 * 2;
    int iters = 4000000 / per_iter;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 2);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = double(iters) * per_iter * waves_per_simd;
    printf("%-34s waves/SIMD=%d: %.3f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms * 1e6 / instr_per_simd);
    hipFree(out);
}

int main() {
    for (int w : {2, 4, 8}) {
        run<0>("mul(sgpr)+add chains, short body", w);
        run<1>("mul(sgpr)+add chains, 4096-instr body", w);
        run<2>("mul(vgpr)+add chains, 4096-instr body", w);
    }
    return 0;
}
