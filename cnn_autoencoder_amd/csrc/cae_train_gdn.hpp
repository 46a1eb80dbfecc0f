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

#ifndef GDN_ABL
#define GDN_ABL 0  // timing experiments (profiles/r03_experiments.md): bit 0 no Gamma^T g_n loop, 1 no g_Gamma loop,
#endif             // 2 no staging after the first tile, 3 no output stores, 4 no element-wise 1 -- results are wrong

namespace cae {
namespace tr {

struct GdnFusedArgs {
    const float *z;       // fp32 [pixels][C]
    const float *gamma;   // fp32 [C][C] (effective)
    const float *beta;    // forward: [C] (effective)
    float *f;             // saved factor, register order (see above): forward writes, backward reads
    void *y16;            // forward: y bf16 [pixels][C]
    FoldSrc gy;           // backward: gradient w.r.t. y, fp32, extended domain with padding gy.P, ALREADY folded in place
    int img_h, img_w;     // backward: pixel index -> (n, y, x)
    void *gz16;           // backward: g_z bf16 [pixels][C]
    float *ggamma, *gbeta;  // backward: [C][C], [C], zeroed by the caller
    long pixels;
    int inverse;
};

// Block barrier that publishes this wave's LDS writes WITHOUT waiting for vector memory: __syncthreads() compiles to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, i.e. a barrier after the prefetch has been issued waits for the whole
// prefetch -- the first form of these kernels ran at exactly MFMA time + memory time (profiles/r03_experiments.md).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Reflect fold of an extended-domain gradient IN PLACE: every interior pixel next to the border collects the values its
// mirror images in the padding ring hold (fold_read's sum; the sources lie in the ring and are never written).  One block
// per (sample, row); lanes over channels.  Afterwards the interior of g is the gradient with respect to the unpadded
// tensor and consumers read it with a plain offset.
static __global__ void fold_inplace_kernel(float *g, int H, int W, int P, int C) {
    // jobs per sample: one block per fold ROW (rows 1..P and H-1-P..H-2: every column), plus one block for the fold
    // COLUMNS of all other rows -- a block per (sample, row) spent its time being launched (16 384 blocks for 2 % of the pixels)
    const int jobs = 2 * P + 1;
    const int n = blockIdx.x / jobs, job = blockIdx.x - n * jobs;
    const int HP = H + 2 * P, WP = W + 2 * P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const FoldSrc f{g, H, W, P};
    auto row_folds = [&](int y) { return (y >= 1 && y <= P) || (y <= H - 2 && y >= H - 1 - P); };
    auto col_folds = [&](int x) { return (x >= 1 && x <= P) || (x <= W - 2 && x >= W - 1 - P); };
    auto fix = [&](int y, int x) {
        float *dst = g + (((size_t)n * HP + y + P) * WP + x + P) * C;
        for (int c = lane; c < C; c += 64) dst[c] = fold_read(f, n, y, x, C, c);
    };
    if (job < 2 * P) {
        const int y = job < P ? 1 + job : H - 1 - P + (job - P);
        // (tiny images: the two bands overlap -- a row of the second band that the first band owns is skipped; row 0 is NOT in
        //  the first band: `y <= P` here dropped it for H <= P + 1, found by tests/fuzz/fuzz_train.py on 3 x 5 inputs with k = 5)
        if (y < 0 || y >= H || !row_folds(y) || (job >= P && y >= 1 && y <= P)) return;
        for (int x = wave; x < W; x += nw) fix(y, x);
    } else {
        for (int y = wave; y < H; y += nw) {
            if (row_folds(y)) continue;  // done by its row block
            for (int k = 0; k < 2 * P; ++k) {
                const int x = k < P ? 1 + k : W - 1 - P + (k - P);
                if (x < 0 || x >= W || !col_folds(x) || (k >= P && x >= 1 && x <= P)) continue;
                fix(y, x);
            }
        }
    }
}

// byte offset of 16-byte piece `s` of tile pixel `p` in a staged fp32 tile of C channels
// (LDS serves a ds_read_b128 sixteen lanes at a time: sixteen consecutive pixels reading the same piece must hit sixteen
// different 16-byte bank groups.  Rows of a multiple of 256 bytes all start in the same group -> XOR the low 4 bits of the
// pixel into the slot; 128-byte rows alternate between the two halves of the banks by themselves -> 3 bits of p >> 1.)
template <int C>
__device__ __forceinline__ int gdn_swz(int p, int s) {
    return (C % 64 == 0) ? (s ^ (p & 15)) : ((s & ~7) | ((s & 7) ^ ((p >> 1) & 7)));
}
template <int C>
__device__ __forceinline__ int gdn_slot(int p, int s) {
    return p * (C * 4) + gdn_swz<C>(p, s) * 16;
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
            const int s = gdn_swz<C>(p, slot);  // the piece whose swizzled slot this lane fills (the XOR is an involution)
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
// forward: 32-pixel tiles
// ---------------------------------------------------------------------------------------------------------------
// Gamma lives in REGISTERS here: the A operand of a wave is its 32 rows of Gamma, i.e. C / 8 x 16 bytes per lane = 64 VGPRs
// for 128 channels, loaded once per block.  Without the 66-KiB LDS image two blocks share a CU (32-pixel tiles: 41 KiB
// each), so one block's element-wise / staging phases run under the other's MFMAs (one block per CU with Gamma in the
// LDS and 64-pixel tiles: 0.47-0.55 of the fp32 MFMA peak; profiles/r03_experiments.md).
template <int CT>
__global__ void __launch_bounds__(CT * 64, 2) gdn_fwd_fused_kernel(const GdnFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = CT * 32, PT = 32;
    constexpr int Z_BYTES = PT * C * 4, YS = C * 2 + 16;  // padded bf16 output rows
    char *zbuf = smem;                      // 2 x Z_BYTES
    char *ybuf = zbuf + 2 * Z_BYTES;        // PT x YS
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    f32x4 gam[C / 8];  // Gamma[32 wave + m][8 q + 4 h .. + 3]
#pragma unroll
    for (int q = 0; q < C / 8; ++q) gam[q] = *(const f32x4 *)(p.gamma + (size_t)(32 * wave + m) * C + 8 * q + 4 * h);
    const long tiles = (p.pixels + PT - 1) / PT;
    auto zrow = [&](long gp) { return (const char *)(p.z + gp * C); };
    // The previous tile's outputs leave at the TOP of an iteration, before the next prefetch is issued: vmcnt counts loads
    // and stores alike, and a store issued after the prefetch would make the wait for that prefetch a wait for the store's
    // round trip to L2 as well (measured: the kernel ran at 0.45 of the fp32 MFMA peak with the stores at the bottom).
    auto copy_out = [&](long t) {
        for (int i = threadIdx.x; i < PT * (C / 8); i += CT * 64) {
            const int px = i / (C / 8), part = i - px * (C / 8);
            const long gp = t * PT + px;
            if (gp < p.pixels)
                *(f32x4 *)((char *)p.y16 + (gp * C) * 2 + part * 16) = *(const f32x4 *)(ybuf + px * YS + part * 16);
        }
    };
    f32x4 fkeep[4];  // the factors of the previous tile, stored one iteration later for the same reason
    auto store_f = [&](long t) {
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            *(f32x4 *)(p.f + (((t * CT + wave) * 4 + g) * 64 + lane) * 4) = fkeep[g];
        });
    };
    float bet[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bet[r] = p.beta[32 * wave + acc_row(r) + 4 * h];
    long tile = blockIdx.x, prev = -1;
    if (tile < tiles) gdn_stage<C, PT, CT>(zbuf, wave, lane, tile * PT, p.pixels, zrow);
    for (int it = 0; tile < tiles; prev = tile, tile += gridDim.x, ++it) {
        char *cur = zbuf + (it & 1) * Z_BYTES;
        wait_vm0();
        __syncthreads();  // this tile landed; the previous tile's output image is complete; the other z buffer is free
        if (prev >= 0) {
            copy_out(prev);
            store_f(prev);
        }
        if (tile + gridDim.x < tiles)
            gdn_stage<C, PT, CT>(zbuf + ((it + 1) & 1) * Z_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, zrow);
        f32x16 acc, acc1;  // two chains (even / odd k-steps)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = bet[r];
            acc1[r] = 0.0f;
        }
        // n[c][pix] = beta[c] + sum_j Gamma[c][j] z[pix][j]^2 :  A = Gamma rows of this wave's tile (registers), B = z^2
        // (lane = pixel); the z pieces of step q + 1 are read while the MFMAs of step q run
        {
            f32x4 v[2];
            v[0] = *(const f32x4 *)(cur + gdn_slot<C>(m, h));
            static_for<C / 8>([&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value, k = q & 1;
                if (q + 1 < C / 8) v[k ^ 1] = *(const f32x4 *)(cur + gdn_slot<C>(m, 2 * (q + 1) + h));
                v[k] *= v[k];
#pragma unroll
                for (int s = 0; s < 4; s += 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gam[q][s], v[k][s], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gam[q][s + 1], v[k][s + 1], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        }
        lds_barrier();  // the copy-out of the previous tile has read the output image (no wait for the prefetch)
        // element-wise on the wave's own channels: run g = channels 32 wave + 8 g + 4 h .. + 3 of pixel m
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            const f32x4 zz = *(const f32x4 *)(cur + gdn_slot<C>(m, 8 * wave + 2 * g + h));
            bf16x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = acc[4 * g + k];
                fkeep[g][k] = p.inverse ? __builtin_amdgcn_sqrtf(d) : __builtin_amdgcn_rsqf(d);
                y[k] = (__bf16)(zz[k] * fkeep[g][k]);
            }
            *(bf16x4 *)(ybuf + m * YS + (32 * wave + 8 * g + 4 * h) * 2) = y;
        });
    }
    __syncthreads();
    if (prev >= 0) {
        copy_out(prev);
        store_f(prev);
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
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.img_w;
    auto zrow = [&](long gp) { return (const char *)(p.z + gp * C); };
    // pixel index -> (n, y, x) by float reciprocal + one correction step (quotients stay far below 2^23; an integer
    // division costs ~40 vector instructions and a thread needs sixteen per tile)
    auto fdiv = [](unsigned u, unsigned d, float inv, unsigned &q, unsigned &r) {
        q = (unsigned)((float)u * inv);
        int rr = (int)(u - q * d);
        if (rr < 0) {
            --q;
            rr += (int)d;
        } else if (rr >= (int)d) {
            ++q;
            rr -= (int)d;
        }
        r = (unsigned)rr;
    };
    auto split = [&](long gp, int &n, int &y, int &x) {
        unsigned q, r, yy, xx;
        fdiv((unsigned)gp, hw, inv_hw, q, r);
        fdiv(r, (unsigned)p.img_w, inv_w, yy, xx);
        n = (int)q;
        y = (int)yy;
        x = (int)xx;
    };
    auto grow = [&](long gp) {  // the pixel's own row in the extended-domain gradient
        if (P == 0) return (const char *)(p.gy.g + gp * C);
        int n, y, x;
        split(gp, n, y, x);
        return (const char *)(p.gy.g + (((size_t)n * HP + y + P) * WP + x + P) * C);
    };
    // (as in the forward kernel: the previous tile's output leaves at the top of an iteration, before the next prefetch)
    auto copy_out = [&](long t) {
        for (int i = threadIdx.x; i < PT * (C / 8); i += CT * 64) {
            const int px = i / (C / 8), part = i - px * (C / 8);
            const long gp = t * PT + px;
            if (gp < p.pixels)
                *(f32x4 *)((char *)p.gz16 + (gp * C) * 2 + part * 16) = *(const f32x4 *)(obuf + px * OS + part * 16);
        }
    };
    f32x4 fnext[4];  // the saved factors of the NEXT tile, loaded a tile ahead beside the LDS-DMA prefetch
    auto load_f = [&](long t) {
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            fnext[g] = *(const f32x4 *)(p.f + (((t * CT + wave) * 4 + g) * 64 + lane) * 4);
        });
    };

    f32x16 accg[CT];  // g_Gamma[32 wave + row][32 jt + lane]
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[jt][r] = 0.0f;
    float bsum = 0.0f;

    const long tiles = (p.pixels + PT - 1) / PT;
    long tile = blockIdx.x, prev = -1;
    if (tile < tiles) {
        gdn_stage<C, PT, CT>(zbuf, wave, lane, tile * PT, p.pixels, zrow);
        gdn_stage<C, PT, CT>(gbuf, wave, lane, tile * PT, p.pixels, grow);
        load_f(tile);
    }
    for (int it = 0; tile < tiles; prev = tile, tile += gridDim.x, ++it) {
        char *zc = zbuf + (it & 1) * T_BYTES, *gc = gbuf + (it & 1) * T_BYTES;
        wait_vm0();
        f32x4 f[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) f[g] = fnext[g];
        __syncthreads();  // both tiles landed; the previous output image is complete; the other buffers are free
        if (prev >= 0 && !(GDN_ABL & 8)) copy_out(prev);
        if (tile + gridDim.x < tiles && !(GDN_ABL & 4)) {
            gdn_stage<C, PT, CT>(zbuf + ((it + 1) & 1) * T_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, zrow);
            gdn_stage<C, PT, CT>(gbuf + ((it + 1) & 1) * T_BYTES, wave, lane, (tile + gridDim.x) * PT, p.pixels, grow);
            load_f(tile + gridDim.x);
        }
        // element-wise 1 on the wave's own channels of pixel m: g_n (back into the g_y tile), direct term, z kept
        const bool valid = tile * PT + m < p.pixels;  // pixels past the end contribute nothing to the parameter gradients
        f32x4 zz[4], gzd[4];
        if (GDN_ABL & 16)
            for (int g = 0; g < 4; ++g) zz[g] = gzd[g] = f[g];
        if (!(GDN_ABL & 16)) static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            const int off = gdn_slot<C>(m, 8 * wave + 2 * g + h);
            zz[g] = *(const f32x4 *)(zc + off);
            const f32x4 gy = *(const f32x4 *)(gc + off);
            f32x4 gn;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gzd[g][k] = gy[k] * f[g][k];
                const float v = p.inverse ? 0.5f * gy[k] * zz[g][k] * __builtin_amdgcn_rcpf(f[g][k])
                                          : -0.5f * gy[k] * zz[g][k] * f[g][k] * f[g][k] * f[g][k];
                gn[k] = valid ? v : 0.0f;
            }
            *(f32x4 *)(gc + off) = gn;
        });
        lds_barrier();  // g_n of every channel is in place; the copy-out has read the output image
        // t[j][pix] = sum_c Gamma[c][j] g_n[c][pix] :  A = Gamma^T rows j of this wave's tile (column reads), B = g_n.
        // Operands of step q + 1 are read while the MFMAs of step q run (registers: the compiler's own schedule waited for
        // every LDS read right before its MFMA pair); two accumulators break the dependent chain.
        f32x16 t, t1;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = t1[r] = 0.0f;
        {
            constexpr int NQ = (GDN_ABL & 1) ? 0 : C / 8;
            f32x4 vb[2];
            float a[2][4];
            auto fetch = [&](int q, int k) {
                vb[k] = *(const f32x4 *)(gc + gdn_slot<C>(m, 2 * q + h));
#pragma unroll
                for (int s = 0; s < 4; ++s) a[k][s] = mlds[(8 * q + 4 * h + s) * LD + 32 * wave + m];
            };
            if (NQ) fetch(0, 0);
            static_for<NQ>([&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value;
                if (q + 1 < NQ) fetch(q + 1, (q + 1) & 1);
#pragma unroll
                for (int s = 0; s < 4; s += 2) {
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][s], vb[q & 1][s], t, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][s + 1], vb[q & 1][s + 1], t1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] += t1[r];
        }
        static_for<4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__bf16)(gzd[g][k] + 2.0f * zz[g][k] * t[4 * g + k]);
            *(bf16x4 *)(obuf + m * OS + (32 * wave + 8 * g + 4 * h) * 2) = o;
        });
        // g_Gamma[c][j] += sum_pix g_n[c][pix] z[pix][j]^2 : channel on the lane, two pixels per k-step.  The pair of a
        // step is (px, px + 8): with the tile's swizzle the two lane halves then read different banks.
        {
            constexpr int NS = (GDN_ABL & 2) ? 0 : PT / 2;
            float a[2], b[2][CT];
            auto fetch = [&](int s, int k) {
                const int px = (s & 7) + 16 * (s >> 3) + 8 * h;
                a[k] = *(const float *)(gc + gdn_slot<C>(px, 8 * wave + (m >> 2)) + (m & 3) * 4);
#pragma unroll
                for (int jt = 0; jt < CT; ++jt)
                    b[k][jt] = *(const float *)(zc + gdn_slot<C>(px, 8 * jt + (m >> 2)) + (m & 3) * 4);
            };
            if (NS) fetch(0, 0);
            static_for<NS>([&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                if (s + 1 < NS) fetch(s + 1, (s + 1) & 1);
                bsum += a[s & 1];
#pragma unroll
                for (int jt = 0; jt < CT; ++jt)
                    accg[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1], b[s & 1][jt] * b[s & 1][jt], accg[jt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    __syncthreads();
    if (prev >= 0) copy_out(prev);
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
