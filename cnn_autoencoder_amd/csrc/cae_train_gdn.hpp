// Fused GDN / IGDN kernels of the training path, exact fp32 (v_mfma_f32_32x32x2_f32).  Reference: compressai's GDN
// under autograd, reached from DownsamplingUnit / UpsamplingUnit (models/tasks/_autoencoders.py:29-30) in
// `loss.backward()` (train_cae_ms.py:214).
//
//   forward   n = beta + Gamma z^2 ;  f = n^(-1/2) (IGDN: n^(1/2)) ;  y = z f            -> y (bf16), f (fp32, saved)
//   backward  g_n = -(1/2) g_y z f^3 (IGDN: +(1/2) g_y z / f) ;  t = Gamma^T g_n ;  g_z = g_y f + 2 z t   -> g_z (bf16)
//             g_Gamma = sum_pix g_n (x) z^2 ;  g_beta = sum_pix g_n
//
// One block = CT waves on one pixel tile, wave w owns the 32-channel tile w of every OUTPUT; all contractions are written
// with the channel on the accumulator ROWS and the pixel on the LANES (D[c][pix]): a lane then holds, for its pixel, four
// runs of four consecutive channels -- exactly the 16-byte pieces of the pixel's row -- so
//   * the z / g_y tiles are staged ONCE per pixel tile by LDS-DMA (whole 128-byte row segments, XOR-swizzled 16-byte
//     slots) and serve as MFMA operand (all channels of a pixel) and as element-wise operand (the wave's own channels),
//   * g_n goes back into the g_y tile in place and is the B operand of Gamma^T g_n without any transpose,
//   * the saved factor f is dumped in REGISTER order ([32-pixel tile][wave][run][lane][4]): both kernels use the same
//     tiling, so it is written and read as whole 1-KiB lines and never staged,
//   * outputs leave through a padded LDS image as whole rows.
// The parameter gradients are the third contraction of the same resident tiles (over pixels: channel on the lane, two
// pixels per k-step), accumulated in registers over the block's whole pixel range and flushed once with atomics.
// Next tile's inputs are prefetched (double buffer) under the current tile's MFMAs.
//
// The three-kernel form of round 2 (gdn_gemm_a<.,1>, <.,2>, gdn_gemm_b: 42 bytes of HBM traffic per element, element-wise
// work as dependent global loads after the MFMAs of a 1-wave-per-SIMD block) ran at 0.09-0.2 of the fp32 MFMA peak and was
// 64 % of a training step's kernel time at batch 128 (profiles/r03_experiments.md).
#pragma once
#include "cae_train_kernels.hpp"

namespace cae {
namespace tr {

struct GdnFusedArgs {
    const float *z;       // fp32 [pixels][C]
    const float *gamma;   // fp32 [C][C] (effective)
    const float *beta;    // forward: [C] (effective)
    float *f;             // saved factor, register order (see above): forward writes, backward reads
    void *y16;            // forward: y bf16 [pixels][C]
    FoldSrc gy;           // backward: gradient w.r.t. y, fp32, extended domain with padding gy.P (0: plain)
    int img_h, img_w;     // backward: pixel index -> (n, y, x)
    void *gz16;           // backward: g_z bf16 [pixels][C]
    float *ggamma, *gbeta;  // backward: [C][C], [C], zeroed by the caller
    long pixels;
    int inverse;
};

// byte offset of 16-byte piece `s` of tile pixel `p` in a staged fp32 tile of C channels
template <int C>
__device__ __forceinline__ int gdn_slot(int p, int s) {
    return p * (C * 4) + ((s & ~7) | ((s & 7) ^ ((p >> 1) & 7))) * 16;
}

// LDS-DMA of `PT` pixel rows (fp32, C channels) starting at tile pixel 0 = global pixel pix0; src_of(pixel) -> row address.
// Instruction j (of PT * C / 256, spread over the NW waves) fills 1 KiB = 256 / C pixel rows.
template <int C, int PT, int NW, class SrcOf>
__device__ __forceinline__ void gdn_stage(char *buf, int wave, int lane, long pix0, long pixels, SrcOf src_of) {
    constexpr int INSTR = PT * C * 4 / 1024;
#pragma unroll
    for (int i = 0; i < (INSTR + NW - 1) / NW; ++i) {
        const int j = wave + NW * i;
        if (INSTR % NW == 0 || j < INSTR) {
            const int byte = j * 1024 + lane * 16;
            const int p = byte / (C * 4), slot = (byte % (C * 4)) >> 4;
            const int s = (slot & ~7) | ((slot & 7) ^ ((p >> 1) & 7));  // the piece whose swizzled slot this lane fills
            long gp = pix0 + p;
            gp = gp < pixels ? gp : pixels - 1;  // clamped rows are computed and never stored / masked
            glds16(src_of(gp) + s * 16, buf + j * 1024);
        }
    }
}

template <int C>
__device__ __forceinline__ void gdn_load_gamma(float *mlds, const float *gamma, int tid, int nthreads) {
    constexpr int LD = C + 4;
    for (int i = tid; i < C * C / 4; i += nthreads) {
        const int r = i / (C / 4), c4 = i - r * (C / 4);
        *(f32x4 *)(mlds + r * LD + 4 * c4) = *(const f32x4 *)(gamma + (size_t)r * C + 4 * c4);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: 64-pixel tiles (two 32-pixel MFMA column tiles per wave)
// ---------------------------------------------------------------------------------------------------------------
template <int CT>
__global__ void __launch_bounds__(CT * 64, 1) gdn_fwd_fused_kernel(const GdnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = CT * 32, LD = C + 4, PT = 64;
    constexpr int M_BYTES = C * LD * 4, Z_BYTES = PT * C * 4, YS = C * 2 + 16;  // padded bf16 output rows
    float *mlds = (float *)smem;
    char *zbuf = smem + M_BYTES;            // 2 x Z_BYTES
    char *ybuf = zbuf + 2 * Z_BYTES;        // PT x YS
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    gdn_load_gamma<C>(mlds, p.gamma, threadIdx.x, CT * 64);
    const long tiles = (p.pixels + PT - 1) / PT;
    auto zrow = [&](long gp) { return (const char *)(p.z + gp * C); };
    long tile = blockIdx.x;
    if (tile < tiles) gdn_stage<C, PT, CT>(zbuf, wave, lane, tile * PT, p.pixels, zrow);
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
        char *cur = zbuf + (it & 1) * Z_BYTES;
        wait_vm0();
        __syncthreads();  // this tile landed; the other buffer and ybuf are free
        if (tile + gridDim.x < tiles)
            gdn_stage<C, PT, CT>(zbuf + ((it + 1) & 1) * Z_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, zrow);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = p.beta[32 * wave + acc_row(r) + 4 * h];
            acc[0][r] = b;
            acc[1][r] = b;
        }
        // n[c][pix] = beta[c] + sum_j Gamma[c][j] z[pix][j]^2 :  A = Gamma rows of this wave's tile, B = z^2 (lane = pixel)
#pragma unroll 4
        for (int q = 0; q < C / 8; ++q) {
            const f32x4 mf = *(const f32x4 *)(mlds + (32 * wave + m) * LD + 8 * q + 4 * h);
            f32x4 v0 = *(const f32x4 *)(cur + gdn_slot<C>(m, 2 * q + h));
            f32x4 v1 = *(const f32x4 *)(cur + gdn_slot<C>(32 + m, 2 * q + h));
            v0 *= v0;
            v1 *= v1;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[s], v0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[s], v1[s], acc[1], 0, 0, 0);
            }
        }
        // element-wise on the wave's own channels: run g = channels 32 wave + 8 g + 4 h .. + 3 of pixel 32 pt + m
        static_for<2>([&](auto pt_tag) {
            constexpr int pt = decltype(pt_tag)::value;
            const long tile32 = tile * 2 + pt;
            static_for<4>([&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                const f32x4 zz = *(const f32x4 *)(cur + gdn_slot<C>(32 * pt + m, 8 * wave + 2 * g + h));
                f32x4 f;
                bf16x4 y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = acc[pt][4 * g + k];
                    f[k] = p.inverse ? __builtin_amdgcn_sqrtf(d) : __builtin_amdgcn_rsqf(d);
                    y[k] = (__bf16)(zz[k] * f[k]);
                }
                *(f32x4 *)(p.f + (((tile32 * CT + wave) * 4 + g) * 64 + lane) * 4) = f;
                *(bf16x4 *)(ybuf + (32 * pt + m) * YS + (32 * wave + 8 * g + 4 * h) * 2) = y;
            });
        });
        __syncthreads();  // the output image is complete
        for (int i = threadIdx.x; i < PT * (C / 8); i += CT * 64) {
            const int px = i / (C / 8), part = i - px * (C / 8);
            const long gp = tile * PT + px;
            if (gp < p.pixels)
                *(f32x4 *)((char *)p.y16 + (gp * C) * 2 + part * 16) = *(const f32x4 *)(ybuf + px * YS + part * 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: 32-pixel tiles
// ---------------------------------------------------------------------------------------------------------------
template <int CT>
__global__ void __launch_bounds__(CT * 64, 1) gdn_bwd_fused_kernel(const GdnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = CT * 32, LD = C + 4, PT = 32;
    constexpr int M_BYTES = C * LD * 4, T_BYTES = PT * C * 4, OS = C * 2 + 16;
    float *mlds = (float *)smem;
    char *zbuf = smem + M_BYTES;        // 2 x T_BYTES
    char *gbuf = zbuf + 2 * T_BYTES;    // 2 x T_BYTES: g_y, then g_n in place
    char *obuf = gbuf + 2 * T_BYTES;    // PT x OS
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    gdn_load_gamma<C>(mlds, p.gamma, threadIdx.x, CT * 64);
    const int P = p.gy.P, HP = p.img_h + 2 * P, WP = p.img_w + 2 * P;
    const unsigned hw = (unsigned)(p.img_h * p.img_w);
    auto zrow = [&](long gp) { return (const char *)(p.z + gp * C); };
    auto split = [&](long gp, int &n, int &y, int &x) {
        const unsigned u = (unsigned)gp;
        n = (int)(u / hw);
        const unsigned rem = u - (unsigned)n * hw;
        y = (int)(rem / (unsigned)p.img_w);
        x = (int)(rem - (unsigned)y * (unsigned)p.img_w);
    };
    auto grow = [&](long gp) {  // the pixel's own row in the extended-domain gradient
        int n, y, x;
        split(gp, n, y, x);
        return (const char *)(p.gy.g + (((size_t)n * HP + y + P) * WP + x + P) * C);
    };
    // reflect fold (P > 0): pixels next to the border also collect the gradient of their mirror images in the padding
    // ring; the lane that staged a piece of such a pixel rewrites it with the folded sum before the tile is published
    auto fold_fix = [&](char *buf, long pix0) {
        constexpr int INSTR = PT * C * 4 / 1024;
#pragma unroll
        for (int i = 0; i < (INSTR + CT - 1) / CT; ++i) {
            const int j = wave + CT * i;
            if (INSTR % CT == 0 || j < INSTR) {
                const int byte = j * 1024 + lane * 16;
                const int px = byte / (C * 4), slot = (byte % (C * 4)) >> 4;
                const int s = (slot & ~7) | ((slot & 7) ^ ((px >> 1) & 7));
                long gp = pix0 + px;
                gp = gp < p.pixels ? gp : p.pixels - 1;
                int n, y, x;
                split(gp, n, y, x);
                const bool interior = y > P && y < p.img_h - 1 - P && x > P && x < p.img_w - 1 - P;
                if (!interior) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fold_read(p.gy, n, y, x, C, 4 * s + k);
                    *(f32x4 *)(buf + byte) = v;
                }
            }
        }
    };

    f32x16 accg[CT];  // g_Gamma[32 wave + row][32 jt + lane]
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[jt][r] = 0.0f;
    float bsum = 0.0f;

    const long tiles = (p.pixels + PT - 1) / PT;
    long tile = blockIdx.x;
    if (tile < tiles) {
        gdn_stage<C, PT, CT>(zbuf, wave, lane, tile * PT, p.pixels, zrow);
        gdn_stage<C, PT, CT>(gbuf, wave, lane, tile * PT, p.pixels, grow);
    }
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it) {
        char *zc = zbuf + (it & 1) * T_BYTES, *gc = gbuf + (it & 1) * T_BYTES;
        wait_vm0();
        if (P > 0) fold_fix(gc, tile * PT);
        __syncthreads();  // both tiles landed (and folded); the other buffers and obuf are free
        if (tile + gridDim.x < tiles) {
            gdn_stage<C, PT, CT>(zbuf + ((it + 1) & 1) * T_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, zrow);
            gdn_stage<C, PT, CT>(gbuf + ((it + 1) & 1) * T_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, grow);
        }
        // element-wise 1 on the wave's own channels of pixel m: g_n (back into the g_y tile), direct term, z kept
        const bool valid = tile * PT + m < p.pixels;  // pixels past the end contribute nothing to the parameter gradients
        f32x4 zz[4], gzd[4];
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            const int off = gdn_slot<C>(m, 8 * wave + 2 * g + h);
            zz[g] = *(const f32x4 *)(zc + off);
            const f32x4 gy = *(const f32x4 *)(gc + off);
            const f32x4 f = *(const f32x4 *)(p.f + (((tile * CT + wave) * 4 + g) * 64 + lane) * 4);
            f32x4 gn;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gzd[g][k] = gy[k] * f[k];
                const float v = p.inverse ? 0.5f * gy[k] * zz[g][k] * __builtin_amdgcn_rcpf(f[k])
                                          : -0.5f * gy[k] * zz[g][k] * f[k] * f[k] * f[k];
                gn[k] = valid ? v : 0.0f;
            }
            *(f32x4 *)(gc + off) = gn;
        });
        __syncthreads();  // g_n of every channel is in place
        // t[j][pix] = sum_c Gamma[c][j] g_n[c][pix] :  A = Gamma^T rows j of this wave's tile (column reads), B = g_n
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.0f;
#pragma unroll 4
        for (int q = 0; q < C / 8; ++q) {
            const f32x4 vb = *(const f32x4 *)(gc + gdn_slot<C>(m, 2 * q + h));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a = mlds[(8 * q + 4 * h + s) * LD + 32 * wave + m];
                t = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb[s], t, 0, 0, 0);
            }
        }
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__bf16)(gzd[g][k] + 2.0f * zz[g][k] * t[4 * g + k]);
            *(bf16x4 *)(obuf + m * OS + (32 * wave + 8 * g + 4 * h) * 2) = o;
        });
        // g_Gamma[c][j] += sum_pix g_n[c][pix] z[pix][j]^2 : channel on the lane, pixels 2 s + h on the k index
#pragma unroll 4
        for (int s = 0; s < PT / 2; ++s) {
            const int px = 2 * s + h;
            const float a = *(const float *)(gc + gdn_slot<C>(px, 8 * wave + (m >> 2)) + (m & 3) * 4);
            bsum += a;
#pragma unroll
            for (int jt = 0; jt < CT; ++jt) {
                float b = *(const float *)(zc + gdn_slot<C>(px, 8 * jt + (m >> 2)) + (m & 3) * 4);
                b *= b;
                accg[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accg[jt], 0, 0, 0);
            }
        }
        __syncthreads();  // the output image is complete
        for (int i = threadIdx.x; i < PT * (C / 8); i += CT * 64) {
            const int px = i / (C / 8), part = i - px * (C / 8);
            const long gp = tile * PT + px;
            if (gp < p.pixels)
                *(f32x4 *)((char *)p.gz16 + (gp * C) * 2 + part * 16) = *(const f32x4 *)(obuf + px * OS + part * 16);
        }
    }
    // D: register r = channel 32 wave + acc_row(r) + 4h, lane = j
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
        static_for<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            atomicAdd(p.ggamma + (size_t)(32 * wave + acc_row(r) + 4 * h) * C + 32 * jt + m, accg[jt][r]);
        });
    atomicAdd(p.gbeta + 32 * wave + m, bsum);  // both lane halves add their pixels' share
}

}  // namespace tr
}  // namespace cae
