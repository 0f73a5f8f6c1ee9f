// Dependent-chain VALU microbenchmark: N independent accumulators per wave (N = 1, 2, 4, 8),
// 1-4 waves per SIMD.  Shows how much ILP a wave needs to reach the 2-cycle issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NACC> __global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                // mul -> add pair on the same accumulator, like the stencil sum chain
                float t;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(x[(i + 1) % 8]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(t));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> void run(int waves_per_simd) {
    float *out;
    int blocks = 256 * waves_per_simd;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 1.0001f, 0.5f, 10);
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = double(iters) * 32 * NACC * 2 * waves_per_simd;
    printf("chains=%d waves/SIMD=%d: %.3f ns per wave-instruction per SIMD\n", NACC, waves_per_simd, ms * 1e6 / instr_per_simd);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 4, 6}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
