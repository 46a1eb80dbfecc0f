// Calibration: sustained v_mfma_f32_32x32x2_f32 rate with operands in registers (no memory traffic).
// usage: mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 1e-3f + r;
    float a = a0 + threadIdx.x * 1e-4f, b = b0 - threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    int nblocks = 256 * blocks_per_cu, iters = 4000;
    float *out;
    hipMalloc(&out, nblocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(nblocks), dim3(256), 0, 0, out, iters, 0.001f, 0.002f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)nblocks * 4 /*waves*/ * iters * 16 * 4 * 4096.0;
        printf("blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, flop / ms / 1e9);
    }
    return 0;
}
