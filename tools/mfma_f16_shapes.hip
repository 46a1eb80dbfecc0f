// Calibration for round 2: the conv K loop's MFMA / LDS pattern on the two f16 MFMA shapes, random operands in LDS.
//   One wave per SIMD (256 threads, 1 block per CU), wave tile = 128 output channels x 64 pixels (128 accumulator
//   registers), f16x3 products (3 MFMAs per operand pair), every operand fragment re-read from LDS with ds_read_b128.
//   shape A: v_mfma_f32_32x32x16_f16 (what the kernels use), shape B: v_mfma_f32_16x16x32_f16.
// Prints the MFMA FLOP/s of both (same FLOPs, same LDS bytes per FLOP).  usage: mfma_f16_shapes [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int LDS_BYTES = 64 * 1024;

__global__ void __launch_bounds__(256, 1) shape_a(const f16x8 *src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) ((f16x8 *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[2][4];
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;
    const char *base = smem + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const char *wb = base + (it & 3) * 4096;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {  // one "tap": K = 16 channels
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bh[p] = *(const f16x8 *)(wb + 36864 + (kx * 4 + p * 2) * 1024);
                bl[p] = *(const f16x8 *)(wb + 36864 + (kx * 4 + p * 2 + 1) * 1024);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 ah = *(const f16x8 *)(wb + ((kx * 4 + c) * 2) * 1024 % 24576);
                const f16x8 al = *(const f16x8 *)(wb + ((kx * 4 + c) * 2 + 1) * 1024 % 24576);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[p], acc[p][c], 0, 0, 0);
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[p], acc[p][c], 0, 0, 0);
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[p], acc[p][c], 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[p][c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256, 1) shape_a_dma(const f16x8 *src, float *out, int iters, const char *stream_src, size_t stream_mask, int pieces) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) ((f16x8 *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[2][4];
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;
    const char *base = smem + lane * 16;
    // + `pieces` LDS-DMA instructions (1 KiB each) per wave and iteration from `stream_src`, never waited for inside the
    // loop (the landing zone is a spare 16 KiB of LDS): prices the staging traffic of the conv K loop (11 per wave
    // and stage there) without its barriers and waits
    const int wave = threadIdx.x >> 6;
    size_t soff = ((size_t)blockIdx.x * 4 + wave) * 65536 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const char *wb = base + (it & 3) * 4096;
        for (int j = 0; j < pieces; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(stream_src + (soff & stream_mask)),
                                             (__attribute__((address_space(3))) void *)(smem + 49152 + ((wave * 4 + (j & 3)) * 1024)), 16, 0, 0);
            soff += 1024 * 1031;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {  // one "tap": K = 16 channels
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bh[p] = *(const f16x8 *)(wb + 36864 + (kx * 4 + p * 2) * 1024);
                bl[p] = *(const f16x8 *)(wb + 36864 + (kx * 4 + p * 2 + 1) * 1024);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 ah = *(const f16x8 *)(wb + ((kx * 4 + c) * 2) * 1024 % 24576);
                const f16x8 al = *(const f16x8 *)(wb + ((kx * 4 + c) * 2 + 1) * 1024 % 24576);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[p], acc[p][c], 0, 0, 0);
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[p], acc[p][c], 0, 0, 0);
                    acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[p], acc[p][c], 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int p = 0; p < 2; ++p)
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[p][c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


__global__ void __launch_bounds__(256, 1) shape_b(const f16x8 *src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) ((f16x8 *)smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[4][8];  // [pixel tile of 16][channel tile of 16]
    for (int p = 0; p < 4; ++p)
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) acc[p][c][r] = 0.f;
    const char *base = smem + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const char *wb = base + (it & 3) * 4096;
        // the same FLOPs as three K=16 taps of shape A = 1.5 K=32 steps; run 3 half-weight steps: 3 x (8 A pairs
        // + 4 B pairs feeding 96 MFMAs of K=32) would be 2x the FLOPs, so each step uses 4 of the 8 channel tiles.
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            f16x8 bh[4], bl[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                bh[p] = *(const f16x8 *)(wb + 36864 + (kx * 8 + p * 2) * 1024 % 12288);
                bl[p] = *(const f16x8 *)(wb + 36864 + (kx * 8 + p * 2 + 1) * 1024 % 12288);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = (kx & 1) * 4 + c;
                const f16x8 ah = *(const f16x8 *)(wb + ((kx * 4 + c) * 2) * 1024 % 24576);
                const f16x8 al = *(const f16x8 *)(wb + ((kx * 4 + c) * 2 + 1) * 1024 % 24576);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    acc[p][cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[p], acc[p][cc], 0, 0, 0);
                    acc[p][cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[p], acc[p][cc], 0, 0, 0);
                    acc[p][cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[p], acc[p][cc], 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for (int p = 0; p < 4; ++p)
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) s += acc[p][c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int dev = 0, ncu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    std::vector<_Float16> h(LDS_BYTES / 2);
    unsigned s = 12345;
    for (auto &v : h) {
        s = s * 1664525u + 1013904223u;
        v = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f);  // random in [-1, 1]
    }
    f16x8 *src;
    float *out;
    hipMalloc(&src, LDS_BYTES);
    hipMalloc(&out, ncu * 256 * 4);
    hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)shape_a, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void *)shape_b, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // per iteration and wave: shape A 3 taps x 24 MFMAs x (2*32*32*16) FLOP; shape B 3 x 48 MFMAs x (2*16*16*32)
    const double flop_a = (double)ncu * 4 * iters * 3 * 24 * 2.0 * 32 * 32 * 16;
    const double flop_b = (double)ncu * 4 * iters * 3 * 48 * 2.0 * 16 * 16 * 32;
    for (int rep = 0; rep < 4; ++rep) {
        float ms_a, ms_b;
        hipEventRecord(e0);
        hipLaunchKernelGGL(shape_a, dim3(ncu), dim3(256), LDS_BYTES, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_a, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(shape_b, dim3(ncu), dim3(256), LDS_BYTES, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_b, e0, e1);
        printf("32x32x16: %8.3f ms %7.1f TFLOP/s | 16x16x32: %8.3f ms %7.1f TFLOP/s | ratio %.3f\n", ms_a,
               flop_a / ms_a / 1e9, ms_b, flop_b / ms_b / 1e9, (flop_b / ms_b) / (flop_a / ms_a));
    }
    // staging traffic beside the MFMAs: L2-resident source (8 MiB) and HBM-streaming source (2 GiB)
    hipFuncSetAttribute((const void *)shape_a_dma, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    char *big;
    const size_t big_bytes = (size_t)2 << 30;
    hipMalloc(&big, big_bytes);
    hipMemset(big, 1, big_bytes);
    for (int pieces : {0, 4, 11, 22}) {
        for (int hbm = 0; hbm < 2; ++hbm) {
            const size_t mask = (hbm ? big_bytes : ((size_t)8 << 20)) - 1;
            float ms;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(shape_a_dma, dim3(ncu), dim3(256), LDS_BYTES, 0, src, out, iters, big, mask & ~(size_t)15, pieces);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)ncu * 4 * iters * pieces * 1024.0;
            printf("32x32x16 + %2d KiB LDS-DMA per wave and 72 MFMAs from %s: %8.3f ms %7.1f TFLOP/s, %.2f TB/s staged\n", pieces,
                   hbm ? "HBM (2 GiB)" : "L2  (8 MiB)", ms, flop_a / ms / 1e9, bytes / ms / 1e9);
        }
    }
    return 0;
}
