// VALU issue-rate microbenchmark for gfx950: plain vs packed fp32 ops, DPP moves.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters) {
    float x[8];
    float2v p[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; p[i] = float2v{x[i], x[i] + 0.5f}; }
    float2v pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (MODE == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
                if constexpr (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if constexpr (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pa));
                if constexpr (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                if constexpr (MODE == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
                if constexpr (MODE == 5) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(x[i]));
                if constexpr (MODE == 6) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pa));
                if constexpr (MODE == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    float *out;
    int blocks = 256 * waves_per_simd;  // 256 CUs, 4 waves per block = 1 per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 0.5f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = double(iters) * 64 * waves_per_simd;  // wave-instructions issued per SIMD
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name,
           waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32", w);
        run<4>("v_add_f32", w);
        run<1>("v_fmac_f32", w);
        run<2>("v_pk_mul_f32", w);
        run<6>("v_pk_add_f32", w);
        run<3>("v_pk_fma_f32", w);
        run<5>("v_mov_dpp+nop", w);
        run<7>("v_mov_b32", w);
    }
    return 0;
}
