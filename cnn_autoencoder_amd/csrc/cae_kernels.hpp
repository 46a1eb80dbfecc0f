// CDNA4 (gfx950) kernels of the convolutional-autoencoder hot path.
//
// Data layout in HBM ("C8"): activations between layers are fp32 [N][P][H][W][8], P = planes of
// 8 channels.  One plane of one image is an H x W picture of 32-byte pixels, so
//   * the input halo of an (8-channel chunk, kernel-row) stage is a set of contiguous row
//     segments that LDS-DMA (global_load_lds_dwordx4) copies straight into LDS, reflect padding
//     resolved in the per-lane SOURCE address (no im2col, no padded copy in HBM);
//   * a 32x32 MFMA accumulator tile (rows = output channels, columns = pixels on the lanes)
//     stores as 16-byte vectors that land in 512-byte contiguous runs.
//
// Contraction: v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain numerics).  A = packed weights
// (32 output channels x 2 k), B = activations (2 k x 32 pixels).  Each ds_read_b128 of either
// operand feeds four consecutive MFMA k-steps (lane half h carries channels 4h..4h+3 of the
// chunk), so LDS traffic is ~20 B/clk/CU: the kernels are MFMA-issue bound by construction.
//
// GDN / IGDN are fused as an epilogue: the conv accumulators (channel rows x pixel lanes) are
// squared in registers and fed back as the B operand of a second MFMA chain against the packed
// gamma matrix ("accumulator tile as the next MFMA's operand"), so y never leaves registers
// between the convolution and the normalisation.
//
// Reference semantics implemented (file:line under /root/reference/src/models/tasks):
//   conv_s2_kernel    nn.Conv2d(k, stride 2, padding k//2, padding_mode='reflect')
//                     _autoencoders.py:78-85 (+ GDN :29-30)
//   deconv_s2_kernel  nn.ConvTranspose2d(k, stride 2, padding k//2, output_padding 1)
//                     _autoencoders.py:204-211 (+ IGDN :29-30), u8 epilogue :576-580
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace cae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum OutFmt { OUT_C8 = 0, OUT_NCHW = 1, OUT_U8HWC = 2, OUT_SYM = 3, OUT_PMAP = 4 };  // OUT_SYM: NCHW int32 round(v - median_c)
// OUT_PMAP (f16x3 synthesis, second-to-last layer): instead of its activations the layer stores, per pixel, their products
// with every (tap, output channel) column of the LAST layer's weights: fp32 [N][OH][OW][32] (cae_kernels_f16.hpp, pmap)

struct LayerArgs {
    const float *in;    // C8 [N][in_planes][H][W][8]
    void *out;          // C8 [N][out_planes][OH][OW][8] | NCHW float | HWC uint8
    const float *wp;    // packed weights  [chunk][ky][kx][ct][64 lanes][4]
    const float *bias;  // [CT*32] (zero padded) or nullptr
    const float *gp;    // packed gamma    [jt][co][q][64 lanes][4]
    const float *beta;  // [CT*32] (padding = 1)
    const float *zero;  // >= 64 B of zeros in HBM (deconv: out-of-range halo source)
    int N, H, W, OH, OW;
    int in_planes;      // plane stride of the input buffer
    int cci;            // number of 8-channel input chunks actually contracted
    int out_planes;     // plane stride of a C8 output buffer
    int cout;           // real number of output channels (NCHW / U8 epilogues)
    int tiles_x, tiles_y;
    int outfmt;
    int act;            // epilogue activation: 0 none, 1 LeakyReLU(0.01), 2 ReLU (nn.LeakyReLU / nn.ReLU defaults)
    const float *res;   // C8 tensor [N][res_planes][OH][OW][8] added after `act` (residual units), or nullptr
    int res_planes;     // planes of `res` (the unit input may carry fewer padded planes than the output)
    int post_act;       // activation after the residual sum
    const float *medians;  // >= 192 floats: per-channel medians (OUT_SYM, fused quantiser) or zeros; never null
    const void *pm;     // OUT_PMAP: packed weights of the last layer [jt][s][hl][64 lanes][8 f16]
    int *flag;          // f16x3 range guard: set to 1 when a value leaves the f16 range (never null on the f16x3 path)
};

// NaN-safe float -> integer epilogues (a NaN / out-of-range cast is undefined behaviour): v_max / v_min return the
// non-NaN operand, so a NaN becomes 255 (uint8) or the lower clamp (symbols; the coder then rejects the value)
__device__ __forceinline__ uint8_t clip_u8(float q) {
    return (uint8_t)__builtin_fmaxf(__builtin_fminf(q, 255.0f), 0.0f);
}
__device__ __forceinline__ int round_sym(float d) {
    return (int)rintf(__builtin_fminf(__builtin_fmaxf(d, -1073741824.0f), 1073741824.0f));
}

// compile-time unrolled loop: f(std::integral_constant<int, I>{}) for I = 0..N-1
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    i = i < 0 ? 0 : i;
    return i >= n ? n - 1 : i;
}

// 16-byte LDS-DMA: lane l copies 16 B from its own global address to lds_wave_base + 16*l.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// row of a 32x32 accumulator register: row = (r&3) + 8*(r>>2) + 4*h
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

template <int CT>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[CT], const float *vec, int h, float fill) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = vec ? vec[32 * ct + acc_row(r) + 4 * h] : fill;
}

// ---- fused GDN / IGDN on the accumulators ---------------------------------------------------
// y[ct] : CT accumulator tiles (rows = channels 32ct.., cols = this wave's 32 pixels).
// Runs CT staged pieces of the packed gamma through the double buffer.  On entry piece 0 has
// been issued into buffer (sc & 1); on exit `tail(nxt)` has been called once to prefetch whatever
// follows.  y is replaced by y * rsqrt(norm) (GDN) or y * sqrt(norm) (IGDN).
template <int CT, int NW, bool INVERSE, int STAGE_BYTES, class Tail>
__device__ __forceinline__ void gdn_stages(f32x16 (&y)[CT], const LayerArgs &p, char *smem, int &sc,
                                           int wave, int lane, Tail tail) {
    constexpr int G_BYTES = CT * 4096;
    const int h = lane >> 5;
    f32x16 nrm[CT];
    init_acc<CT>(nrm, p.beta, h, 1.0f);
#pragma unroll
    for (int jt = 0; jt < CT; ++jt) {
        wait_vm0();
        __syncthreads();
        char *cur = smem + (sc & 1) * STAGE_BYTES;
        char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
        const char *gb = cur + lane * 16;
        f32x4 g_cur[CT];
#pragma unroll
        for (int co = 0; co < CT; ++co) g_cur[co] = *(const f32x4 *)(gb + (co * 4) * 1024);
        if (jt + 1 < CT) {
            const char *src = (const char *)p.gp + (size_t)(jt + 1) * G_BYTES;
#pragma unroll
            for (int i = 0; i < (CT * 4 + NW - 1) / NW; ++i) {
                const int j = wave + i * NW;
                if (j < CT * 4) glds16(src + j * 1024 + lane * 16, nxt + j * 1024);
            }
        } else {
            tail(nxt);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 g_nxt[CT];
            if (q + 1 < 4) {
#pragma unroll
                for (int co = 0; co < CT; ++co) g_nxt[co] = *(const f32x4 *)(gb + (co * 4 + q + 1) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float v = y[jt][4 * q + jj];
                const float sq = v * v;
#pragma unroll
                for (int co = 0; co < CT; ++co)
                    nrm[co] = __builtin_amdgcn_mfma_f32_32x32x2f32(g_cur[co][jj], sq, nrm[co], 0, 0, 0);
            }
            if (q + 1 < 4) {
#pragma unroll
                for (int co = 0; co < CT; ++co) g_cur[co] = g_nxt[co];
            }
        }
        ++sc;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float nv = nrm[ct][r];
            y[ct][r] *= INVERSE ? __builtin_amdgcn_sqrtf(nv) : __builtin_amdgcn_rsqf(nv);
        }
}

// issue packed-gamma piece 0 into `buf`
template <int CT, int NW>
__device__ __forceinline__ void issue_gamma0(const LayerArgs &p, char *buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (CT * 4 + NW - 1) / NW; ++i) {
        const int j = wave + i * NW;
        if (j < CT * 4) glds16((const char *)p.gp + j * 1024 + lane * 16, buf + j * 1024);
    }
}

__device__ __forceinline__ float apply_act(float v, int act) {
    // _define_act_layer (_autoencoders.py:19-34): nn.LeakyReLU() has negative_slope 0.01
    return act == 0 ? v : (act == 1 ? (v > 0.0f ? v : 0.01f * v) : (v > 0.0f ? v : 0.0f));
}

// ---- epilogue store of CT accumulator tiles ---------------------------------------------------
// (oy, ox): this lane's output pixel; valid: inside the image.
// ACT = false compiles the activation out: with the run-time selector inside the element loops the last analysis
// layer (NCHW epilogue, no activation) ran 3x slower (0.65 vs 0.21 ms), so callers branch once per tile instead.
template <int CT, bool ACT>
__device__ __forceinline__ void store_tiles_impl(const f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox,
                                                 int h, bool valid) {
    if (!valid) return;
    if (p.outfmt == OUT_C8) {
        float *out = (float *)p.out;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int plane = 4 * ct + g;
                if (plane < p.out_planes) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = ACT ? apply_act(acc[ct][4 * g + k], p.act) : acc[ct][4 * g + k];
                    const size_t off = ((((size_t)n * p.out_planes + plane) * p.OH + oy) * p.OW + ox) * 8 + 4 * h;
                    if (ACT) {  // residual units: + unit input, then the strided layer's pre-activation
                        if (p.res && plane < p.res_planes)
                            v += *(const f32x4 *)(p.res + ((((size_t)n * p.res_planes + plane) * p.OH + oy) * p.OW + ox) * 8 + 4 * h);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], p.post_act);
                    }
                    *(f32x4 *)(out + off) = v;
                }
            }
    } else {
        // NCHW float | NCHW int32 symbols (fused quantiser) | HWC uint8.  static_for, not `#pragma unroll`: with
        // three 16*CT-element loops in one function the unroller gave up on some of them, the accumulators were then
        // indexed dynamically and lived in scratch memory (3x slower last analysis layer).
        const int fmt = p.outfmt;
        static_for<CT>([&](auto ct_tag) __attribute__((always_inline)) {
            constexpr int ct = decltype(ct_tag)::value;
            static_for<16>([&](auto r_tag) __attribute__((always_inline)) {
                constexpr int r = decltype(r_tag)::value;
                const int c = 32 * ct + acc_row(r) + 4 * h;
                if (c < p.cout) {
                    const float v = ACT ? apply_act(acc[ct][r], p.act) : acc[ct][r];
                    if (fmt == OUT_U8HWC) {  // x*255 -> clip(0,255) -> truncating cast  (_autoencoders.py:576-580)
                        ((uint8_t *)p.out)[(((size_t)n * p.OH + oy) * p.OW + ox) * p.cout + c] = clip_u8(v * 255.0f);
                    } else {
                        // OUT_SYM: symbols = int(round_half_even(y - median_c)), stored through the same float store
                        // (bit pattern); p.medians points at zeros for OUT_NCHW (never null), so both formats share one code path.
                        // (A separate int32 store path made the compiler keep the accumulators in scratch memory.)
                        const size_t off = (((size_t)n * p.cout + c) * p.OH + oy) * p.OW + ox;
                        const float d = v - p.medians[c];
                        ((float *)p.out)[off] = fmt == OUT_SYM ? __int_as_float(round_sym(d)) : d;
                    }
                }
            });
        });
    }
}

template <int CT, bool MAY_ACT = true>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox,
                                            int h, bool valid) {
    if (MAY_ACT && (p.act != 0 || p.res != nullptr || p.post_act != 0))
        store_tiles_impl<CT, true>(acc, p, n, oy, ox, h, valid);
    else
        store_tiles_impl<CT, false>(acc, p, n, oy, ox, h, valid);
}

// =================================================================================================
// Strided reflect convolution (+ bias) (+ GDN)
//   block  = NW waves; wave w owns output rows 2w, 2w+1 of the tile, 16 columns each -> 32 pixels
//   tile   = (2 NW) x 16 output pixels, all CT*32 output channels
//   stage  = (8-channel chunk c, kernel row ky): KS taps x CT KiB of weights + TY halo rows
// =================================================================================================
#ifndef CAE_CONV_WAVES
#define CAE_CONV_WAVES 2
#endif
// XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, each with a private L2, so
// neighbouring tiles -- which share halo rows / columns and re-read them per kernel-row stage -- land on different L2s.
// Give every XCD a contiguous range of tiles instead (whole images at bench sizes).  Bijective for any grid size; a
// placement guess that only affects speed (the block -> XCD map is not a contract).
#ifndef CAE_XCD_SWIZZLE
#define CAE_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int xcd_tile_order(int bid, int nwg) {
#if CAE_XCD_SWIZZLE
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
#else
    return bid;
#endif
}

// S = stride (2: DownsamplingUnit's strided conv; 1: the pre-activation conv of the LeakyReLU/ReLU units,
// _autoencoders.py:62-70, and -- with ZEROPAD and flipped weights -- ConvTranspose2d(stride 1), :187-196)
template <int KS, int CT, int NW, bool GDN, int S = 2, bool ZEROPAD = false, bool INV = false>
__global__ void __launch_bounds__(NW * 64, (GDN && CT >= 6) ? 1 : (CT <= 4 ? CAE_CONV_WAVES : 2)) conv_s2_kernel(const LayerArgs p) {
    constexpr int PAD = KS / 2;
    constexpr int TX = 16, TY = 2 * NW;
    constexpr int WH = S * (TX - 1) + KS;  // halo columns
    constexpr int HALO_PIECES = TY * WH * 2;
    constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;
    constexpr int W_INSTR = KS * CT;
    constexpr int W_BYTES = W_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int CONV_STAGE = W_BYTES + HALO_INSTR * 1024;
    constexpr int STAGE_BYTES = CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES;
    constexpr int MAXP = (HALO_INSTR + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;

    int bid = xcd_tile_order(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX;

    // per-lane halo pieces: piece = (row r, column x, half) -> LDS offset 16*piece.  The byte offset
    // of every piece inside one 8-channel plane is fixed per kernel row ky (reflect padding resolved
    // here, once), so a stage's DMA is: scalar plane base + 32-bit lane offset.
    unsigned hoff[MAXP][KS];  // 0xFFFFFFFF: outside the image (zero padding)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        const int r = pc / (2 * WH);
        const int rem = pc - r * (2 * WH);
        const int x = rem >> 1;
        const int ixr = S * ox0 - PAD + x;
        const bool xok = !ZEROPAD || (ixr >= 0 && ixr < p.W);
        const unsigned xo = (unsigned)reflect_idx(ixr, p.W) * 32u + (unsigned)(rem & 1) * 16u;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iyr = S * (oy0 + r) - PAD + ky;
            const bool ok = xok && (!ZEROPAD || (iyr >= 0 && iyr < p.H));
            hoff[i][ky] = ok ? (unsigned)reflect_idx(iyr, p.H) * (unsigned)p.W * 32u + xo : 0xFFFFFFFFu;
        }
    }
    const size_t plane_bytes = (size_t)p.H * p.W * 32;
    const char *in_n = (const char *)p.in + (size_t)n * p.in_planes * plane_bytes;
    const unsigned woff = (unsigned)lane * 16u;

    // stage (c, ky): weights [c][ky] (KS*CT KiB, contiguous) + the TY halo rows of plane c
    auto issue_stage = [&](int c, auto ky_tag, char *buf) {
        constexpr int ky = decltype(ky_tag)::value;
        const char *wsrc = (const char *)p.wp + (size_t)(c * KS + ky) * W_BYTES;
#pragma unroll
        for (int i = 0; i < (W_INSTR + NW - 1) / NW; ++i) {
            const int j = wave + i * NW;
            if (j < W_INSTR) glds16(wsrc + j * 1024 + woff, buf + j * 1024);
        }
        const char *plane = in_n + (size_t)c * plane_bytes;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * NW;
            if (j < HALO_INSTR) {
                if (ZEROPAD)
                    glds16(hoff[i][ky] != 0xFFFFFFFFu ? (const void *)(plane + hoff[i][ky]) : (const void *)p.zero,
                           buf + W_BYTES + j * 1024);
                else
                    glds16(plane + hoff[i][ky], buf + W_BYTES + j * 1024);
            }
        }
    };

    f32x16 acc[CT];
    init_acc<CT>(acc, p.bias, h, 0.0f);

    const int wrow = 2 * wave + (m >> 4);
    const int b_off = W_BYTES + ((wrow * WH + S * (m & 15)) * 8 + 4 * h) * 4;
    int sc = 0;

    issue_stage(0, std::integral_constant<int, 0>{}, smem);
    for (int c = 0; c < p.cci; ++c) {
        static_for<KS>([&](auto ky_tag) {
            constexpr int ky = decltype(ky_tag)::value;
#if defined(CAE_EXP_NOWAIT)
            __builtin_amdgcn_s_barrier();
#elif !defined(CAE_EXP_NOBARRIER)
            wait_vm0();
            __syncthreads();
#endif
            char *cur = smem + (sc & 1) * STAGE_BYTES;
            char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
            const char *wb = cur + lane * 16;
            const char *hb = cur + b_off;
            // software pipeline inside the wave: operands of tap kx+1 are read while tap kx multiplies,
            // and the next stage's LDS-DMA is issued under the latency of the first reads
            f32x4 b_cur = *(const f32x4 *)(hb);
            f32x4 a_cur[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) a_cur[ct] = *(const f32x4 *)(wb + ct * 1024);
#ifndef CAE_EXP_NODMA
            if constexpr (ky + 1 < KS) {
                issue_stage(c, std::integral_constant<int, ky + 1>{}, nxt);
            } else {
                if (c + 1 < p.cci) {
                    issue_stage(c + 1, std::integral_constant<int, 0>{}, nxt);
                } else if (GDN) {
                    issue_gamma0<CT, NW>(p, nxt, wave, lane);
                }
            }
#endif
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f32x4 b_nxt = b_cur;
                f32x4 a_nxt[CT];
                if (kx + 1 < KS) {
                    b_nxt = *(const f32x4 *)(hb + (kx + 1) * 32);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        a_nxt[ct] = *(const f32x4 *)(wb + ((kx + 1) * CT + ct) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this tap's MFMAs
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct][j], b_cur[j], acc[ct], 0, 0, 0);
                if (kx + 1 < KS) {
                    b_cur = b_nxt;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) a_cur[ct] = a_nxt[ct];
                }
            }
            ++sc;
        });
    }

    if constexpr (GDN) {
        gdn_stages<CT, NW, INV, STAGE_BYTES>(acc, p, smem, sc, wave, lane, [](char *) {});  // INV: IGDN (synthesis res_model)
    }

    const int oy = oy0 + wrow, ox = ox0 + (m & 15);
    store_tiles<CT>(acc, p, n, oy, ox, h, oy < p.OH && ox < p.OW);
}

// =================================================================================================
// Stride-2 transposed convolution (+ bias) (+ IGDN), sub-pixel (phase) decomposition:
//   out[2i+py][2j+px] = sum_{d,dx} in[i-d][j-dx] . W[:, :, 2d+py+P, 2dx+px+P]      (P = KS/2)
//   block  = NW waves; wave w owns INPUT row ty0+w, 32 input columns -> output rows 2i, 2i+1
//   per py: two accumulator sets (px = 0, 1); stage = (chunk c, kernel row ky(py, d))
// =================================================================================================
template <int KS, int CT, int NW, bool IGDN>
struct DeconvGeom {
    static constexpr int P = KS / 2;
    static constexpr int DLO = -((P + 1) / 2), DHI = (KS - 1 - P) / 2;  // range of dx over both px
    static constexpr int WH = 32 + DHI - DLO;
    static constexpr int HALO_PIECES = NW * WH * 2;
    static constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;
    static constexpr int W_INSTR = KS * CT;
    static constexpr int W_BYTES = W_INSTR * 1024;
    static constexpr int G_BYTES = IGDN ? CT * 4096 : 0;
    static constexpr int CONV_STAGE = W_BYTES + HALO_INSTR * 1024;
    static constexpr int STAGE_BYTES = CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES;
    static constexpr int MAXP = (HALO_INSTR + NW - 1) / NW;
    static constexpr int dmin(int py) { return -((py + P) / 2); }
    static constexpr int nky(int py) { return (KS - 1 - py - P) / 2 - dmin(py) + 1; }
};

template <int KS, int CT, int NW, bool IGDN, int PY>
__device__ __forceinline__ void deconv_issue(const LayerArgs &p, const float *in_n, size_t plane_sz, int s, char *buf,
                                             const int *hrow, const int *hxoff, int wave, int lane) {
    using G = DeconvGeom<KS, CT, NW, IGDN>;
    constexpr int NKY = G::nky(PY);
    const int c = s / NKY, d = G::dmin(PY) + (s - c * NKY);
    const int ky = 2 * d + PY + G::P;
    const char *wsrc = (const char *)p.wp + (size_t)(c * KS + ky) * G::W_BYTES;
#pragma unroll
    for (int i = 0; i < (G::W_INSTR + NW - 1) / NW; ++i) {
        const int j = wave + i * NW;
        if (j < G::W_INSTR) glds16(wsrc + j * 1024 + lane * 16, buf + j * 1024);
    }
    const float *plane = in_n + (size_t)c * plane_sz;
#pragma unroll
    for (int i = 0; i < G::MAXP; ++i) {
        const int j = wave + i * NW;
        if (j < G::HALO_INSTR) {
            const int iy = hrow[i] - d;
            const bool ok = iy >= 0 && iy < p.H && hxoff[i] >= 0;
            const float *src = ok ? plane + (size_t)iy * p.W * 8 + hxoff[i] : p.zero;
            glds16(src, buf + G::W_BYTES + j * 1024);
        }
    }
}

template <int KS, int CT, int NW, bool IGDN, int PY>
__device__ __forceinline__ void deconv_phase(const LayerArgs &p, const float *in_n, size_t plane_sz, char *smem,
                                             int &sc, const int *hrow, const int *hxoff, int wave, int lane,
                                             int b_off, int n, int iy, int ix, bool valid) {
    using G = DeconvGeom<KS, CT, NW, IGDN>;
    constexpr int P = G::P;
    constexpr int STAGE_BYTES = G::STAGE_BYTES;
    const int h = lane >> 5;
    const int NS = p.cci * G::nky(PY);
    f32x16 acc0[CT], acc1[CT];
    init_acc<CT>(acc0, p.bias, h, 0.0f);
    init_acc<CT>(acc1, p.bias, h, 0.0f);

    for (int s = 0; s < NS; ++s) {
        wait_vm0();
        __syncthreads();
        char *cur = smem + (sc & 1) * STAGE_BYTES;
        char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
        const char *wb = cur + lane * 16;
        const char *hb = cur + b_off;
        // kx = 2 dx + px + P  ->  px = parity(kx - P), dx = (kx - P - px) / 2;  B(kx) at hb - dx*32
        constexpr int DX0 = (0 - P - ((0 + P) & 1)) / 2;
        f32x4 b_cur = *(const f32x4 *)(hb - DX0 * 32);
        f32x4 a_cur[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) a_cur[ct] = *(const f32x4 *)(wb + ct * 1024);
        if (s + 1 < NS) {
            deconv_issue<KS, CT, NW, IGDN, PY>(p, in_n, plane_sz, s + 1, nxt, hrow, hxoff, wave, lane);
        } else if (IGDN) {
            issue_gamma0<CT, NW>(p, nxt, wave, lane);
        } else if (PY == 0) {
            deconv_issue<KS, CT, NW, IGDN, 1>(p, in_n, plane_sz, 0, nxt, hrow, hxoff, wave, lane);
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int px = (kx + P) & 1;
            f32x4 b_nxt = b_cur;
            f32x4 a_nxt[CT];
            if (kx + 1 < KS) {
                const int pxn = (kx + 1 + P) & 1;
                const int dxn = (kx + 1 - P - pxn) / 2;
                b_nxt = *(const f32x4 *)(hb - dxn * 32);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) a_nxt[ct] = *(const f32x4 *)(wb + ((kx + 1) * CT + ct) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this tap's MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (px == 0)
                        acc0[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct][j], b_cur[j], acc0[ct], 0, 0, 0);
                    else
                        acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct][j], b_cur[j], acc1[ct], 0, 0, 0);
                }
            if (kx + 1 < KS) {
                b_cur = b_nxt;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) a_cur[ct] = a_nxt[ct];
            }
        }
        ++sc;
    }

    if constexpr (IGDN) {
        gdn_stages<CT, NW, true, STAGE_BYTES>(acc0, p, smem, sc, wave, lane,
                                              [&](char *nxt) { issue_gamma0<CT, NW>(p, nxt, wave, lane); });
        store_tiles<CT>(acc0, p, n, 2 * iy + PY, 2 * ix, h, valid);
        gdn_stages<CT, NW, true, STAGE_BYTES>(acc1, p, smem, sc, wave, lane, [&](char *nxt) {
            if (PY == 0) deconv_issue<KS, CT, NW, IGDN, 1>(p, in_n, plane_sz, 0, nxt, hrow, hxoff, wave, lane);
        });
        store_tiles<CT>(acc1, p, n, 2 * iy + PY, 2 * ix + 1, h, valid);
    } else {
        store_tiles<CT>(acc0, p, n, 2 * iy + PY, 2 * ix, h, valid);
        store_tiles<CT>(acc1, p, n, 2 * iy + PY, 2 * ix + 1, h, valid);
    }
}

template <int KS, int CT, int NW, bool IGDN>
__global__ void __launch_bounds__(NW * 64, CT >= 6 ? 1 : 2) deconv_s2_kernel(const LayerArgs p) {
    using G = DeconvGeom<KS, CT, NW, IGDN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;

    int bid = xcd_tile_order(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int iy0 = ty * NW, ix0 = tx * 32;

    int hrow[G::MAXP], hxoff[G::MAXP];  // hxoff < 0: column outside the image
#pragma unroll
    for (int i = 0; i < G::MAXP; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < G::HALO_PIECES ? pc : G::HALO_PIECES - 1;
        const int r = pc / (2 * G::WH);
        const int rem = pc - r * (2 * G::WH);
        const int ix = ix0 + (rem >> 1) - G::DHI;
        hrow[i] = iy0 + r;
        hxoff[i] = (ix >= 0 && ix < p.W) ? ix * 8 + (rem & 1) * 4 : -1;
    }
    const size_t plane_sz = (size_t)p.H * p.W * 8;
    const float *in_n = p.in + (size_t)n * p.in_planes * plane_sz;

    const int b_off = G::W_BYTES + ((wave * G::WH + m + G::DHI) * 8 + 4 * h) * 4;
    const int iy = iy0 + wave, ix = ix0 + m;
    const bool valid = iy < p.H && ix < p.W;
    int sc = 0;

    deconv_issue<KS, CT, NW, IGDN, 0>(p, in_n, plane_sz, 0, smem, hrow, hxoff, wave, lane);
    deconv_phase<KS, CT, NW, IGDN, 0>(p, in_n, plane_sz, smem, sc, hrow, hxoff, wave, lane, b_off, n, iy, ix, valid);
    deconv_phase<KS, CT, NW, IGDN, 1>(p, in_n, plane_sz, smem, sc, hrow, hxoff, wave, lane, b_off, n, iy, ix, valid);
}

// =================================================================================================
// First analysis layer: few input channels (Cin <= 4, e.g. RGB) read straight from the caller's
// tensor -- uint8 HWC (fusing the x/255 of codec.encode, _autoencoders.py:542-545) or float NCHW.
//   K is packed as (tap, 4 channels): 2*KS*KS MFMA k-steps instead of the 4*KS*KS a padded
//   8-channel chunk would cost.  The whole halo ((2TY+KS-2) x (2TX+KS-2) pixels x float4) and all
//   packed weights stay in LDS for the block; gamma pieces stream through a double buffer.
//   packed weights: [tap][ct][lane][2]: W(cout = 32ct + (lane&31), ch = 2j + (lane>>5), tap), j = 0,1
// =================================================================================================
struct FirstArgs {
    const void *in;  // (N,H,W,C) uint8  |  (N,C,H,W) float
    int in_is_u8;
    int cin;
};

template <int KS, int CT, int NW, bool GDN>
__global__ void __launch_bounds__(NW * 64, 2) conv_first_kernel(const LayerArgs p, const FirstArgs f) {
    constexpr int PAD = KS / 2;
    constexpr int TX = 16, TY = 2 * NW;
    constexpr int WH = 2 * TX + KS - 2, HH = 2 * TY + KS - 2;
    constexpr int W_BYTES = KS * KS * CT * 512;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *gbuf = smem;                       // 2 * G_BYTES (double-buffered gamma pieces)
    char *wbuf = smem + 2 * G_BYTES;         // W_BYTES
    char *hbuf = wbuf + W_BYTES;             // halo: HH*WH float4, then a 256-entry table of x/255.0f
    float *lut = (float *)(hbuf + ((HH * WH * 16 + 1023) / 1024) * 1024);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX;

    int sc = 0;
    if (GDN) issue_gamma0<CT, NW>(p, gbuf, wave, lane);
    // weights: plain copy of W_BYTES
    for (int i = threadIdx.x; i < W_BYTES / 16; i += NW * 64)
        *(f32x4 *)(wbuf + i * 16) = *(const f32x4 *)((const char *)p.wp + i * 16);
    // uint8 -> float by table: the 256 possible results of the true division torch performs
    if (f.in_is_u8) {
        for (int i = threadIdx.x; i < 256; i += NW * 64) lut[i] = (float)i / 255.0f;
        __syncthreads();
    }
    // halo: one pixel (float4, channels >= cin are zero) per thread-iteration
    for (int i = threadIdx.x; i < HH * WH; i += NW * 64) {
        const int r = i / WH, x = i - r * WH;
        const int iy = reflect_idx(2 * oy0 - PAD + r, p.H), ix = reflect_idx(2 * ox0 - PAD + x, p.W);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f.in_is_u8) {
            const uint8_t *src = (const uint8_t *)f.in + (((size_t)n * p.H + iy) * p.W + ix) * f.cin;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < f.cin) v[c] = lut[src[c]];
        } else {
            const float *src = (const float *)f.in + (size_t)n * f.cin * p.H * p.W + (size_t)iy * p.W + ix;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < f.cin) v[c] = src[(size_t)c * p.H * p.W];
        }
        *(f32x4 *)(hbuf + i * 16) = v;
    }
    __syncthreads();

    f32x16 acc[CT];
    init_acc<CT>(acc, p.bias, h, 0.0f);
    const int wrow = 2 * wave + (m >> 4);
    const char *hb = hbuf + ((2 * wrow) * WH + 2 * (m & 15)) * 16;
    const char *wb = wbuf + lane * 8;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const f32x4 v = *(const f32x4 *)(hb + (ky * WH + kx) * 16);
            const float b0 = h ? v[1] : v[0];
            const float b1 = h ? v[3] : v[2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 a = *(const f32x2 *)(wb + ((ky * KS + kx) * CT + ct) * 512);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1, acc[ct], 0, 0, 0);
            }
        }

    if constexpr (GDN) {
        gdn_stages<CT, NW, false, G_BYTES>(acc, p, gbuf, sc, wave, lane, [](char *) {});
    }
    const int oy = oy0 + wrow, ox = ox0 + (m & 15);
    store_tiles<CT>(acc, p, n, oy, ox, h, oy < p.OH && ox < p.OW);
}

// =================================================================================================
// Last synthesis layer: few output channels (Cout <= 4, e.g. RGB), no IGDN.
//   Gather form on v_mfma_f32_16x16x4_f32: D rows = (cout c, phase p) = 4c + p (p = 2 py + px),
//   D cols = 16 input pixels of one row, K = (neighbour (d, dx), cin):
//     out[2i+py][2j+px][c] = sum_{d,dx,cin} in[i-d][j-dx][cin] * W[cin][c][2d+py+P][2dx+px+P]
//   (taps outside the kernel are zero rows of the packed matrix).  All packed weights live in LDS;
//   the activations stream from HBM/L2 straight into registers (each lane 16 B = 4 channels).
//   packed weights: [nbr][q][lane][4]: A(row = lane&15, k = cin 16q + 4(lane>>4) + s), s = 0..3
// =================================================================================================
template <int KS, int NW>
__global__ void __launch_bounds__(NW * 64, 2) deconv_last_kernel(const LayerArgs p) {
    constexpr int P = KS / 2;
    constexpr int DLO = -((P + 1) / 2), DHI = (KS - 1 - P) / 2;
    constexpr int NB = DHI - DLO + 1;        // neighbours per axis
    constexpr int TXC = 64;                  // input columns per block (4 MFMA column tiles per wave)
    constexpr int WH = TXC + NB - 1, HR = NW + NB - 1;
    constexpr int PLANE_PIECES = HR * WH;    // 16-byte pieces per 4-channel group
    constexpr int HALO_PIECES = 4 * PLANE_PIECES;
    constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;
    constexpr int STAGE_BYTES = HALO_INSTR * 1024;
    constexpr int MAXP = (HALO_INSTR + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][weights]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int nq = p.cci;  // number of 16-channel groups
    char *wbuf = smem + 2 * STAGE_BYTES;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int iy0 = ty * NW, ix0 = tx * TXC;
    const size_t plane_sz = (size_t)p.H * p.W * 8;
    const float *in_n = p.in + (size_t)n * p.in_planes * plane_sz;

    // halo pieces of this lane: LDS image [g'][row][x] of 16-byte pieces (4 channels each)
    long hsrc[MAXP];  // float offset inside the 16-channel group, or -1 (outside the image)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        const int gg = pc / PLANE_PIECES;
        const int rem = pc - gg * PLANE_PIECES;
        const int r = rem / WH, x = rem - r * WH;
        const int sy = iy0 - DHI + r, sx = ix0 - DHI + x;
        const bool ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
        hsrc[i] = ok ? (long)(gg >> 1) * (long)plane_sz + ((long)sy * p.W + sx) * 8 + (gg & 1) * 4 : -1;
    }
    auto issue = [&](int q, char *buf) {
        const float *base = in_n + (size_t)(2 * q) * plane_sz;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * NW;
            if (j < HALO_INSTR) glds16(hsrc[i] >= 0 ? (const void *)(base + hsrc[i]) : (const void *)p.zero, buf + j * 1024);
        }
    };

    issue(0, smem);
    const int w_bytes = NB * NB * nq * 1024;
    for (int i = threadIdx.x; i < w_bytes / 16; i += NW * 64)
        *(f32x4 *)(wbuf + i * 16) = *(const f32x4 *)((const char *)p.wp + i * 16);

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = (p.bias && g < p.cout) ? p.bias[g] : 0.0f;

    for (int q = 0; q < nq; ++q) {
        wait_vm0();
        __syncthreads();
        char *cur = smem + (q & 1) * STAGE_BYTES;
        if (q + 1 < nq) issue(q + 1, smem + ((q + 1) & 1) * STAGE_BYTES);
#pragma unroll
        for (int nd = 0; nd < NB; ++nd)
#pragma unroll
            for (int ndx = 0; ndx < NB; ++ndx) {
                // in[i - d][j - dx], d = DLO + nd: halo row wave - d + DHI, halo column col - dx + DHI
                const int hr = wave - (DLO + nd) + DHI;
                const int hx = col - (DLO + ndx) + DHI;
                const f32x4 a = *(const f32x4 *)(wbuf + ((nd * NB + ndx) * nq + q) * 1024 + lane * 16);
                const char *hb = cur + ((g * HR + hr) * WH + hx) * 16;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4 b = *(const f32x4 *)(hb + t * 256);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2], b[s2], acc[t], 0, 0, 0);
                }
            }
    }
    // lane (col, g = cout): acc[t][r] = output phase r = 2 py + px of input pixel (iy, ix0 + 16 t + col)
    const int iy = iy0 + wave;
    if (iy < p.H && g < p.cout) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ix = ix0 + 16 * t + col;
            if (ix < p.W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oy = 2 * iy + (r >> 1), ox = 2 * ix + (r & 1);
                    if (p.outfmt == OUT_U8HWC) {
                        ((uint8_t *)p.out)[(((size_t)n * p.OH + oy) * p.OW + ox) * p.cout + g] = clip_u8(acc[t][r] * 255.0f);
                    } else {
                        ((float *)p.out)[(((size_t)n * p.cout + g) * p.OH + oy) * p.OW + ox] = acc[t][r];
                    }
                }
            }
        }
    }
}

// =================================================================================================
// Stand-alone GDN / IGDN on a C8 tensor (nn.Module surface for compressai.layers.GDN)
//   block = NW waves, each wave 32 consecutive pixels of the flattened H*W plane.
// =================================================================================================
template <int CT, int NW, bool INVERSE>
__global__ void __launch_bounds__(NW * 64, CT >= 6 ? 1 : 2) gdn_c8_kernel(const LayerArgs p) {
    constexpr int STAGE_BYTES = CT * 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int hw = p.H * p.W;
    const int blocks_per_img = (hw + NW * 32 - 1) / (NW * 32);
    const int n = blockIdx.x / blocks_per_img;
    const int pix = (blockIdx.x - n * blocks_per_img) * (NW * 32) + wave * 32 + m;
    const bool valid = pix < hw;
    int sc = 0;
    issue_gamma0<CT, NW>(p, smem, wave, lane);
    f32x16 y[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int plane = 4 * ct + g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (valid && plane < p.in_planes)
                v = *(const f32x4 *)(p.in + (((size_t)n * p.in_planes + plane) * hw + pix) * 8 + 4 * h);
#pragma unroll
            for (int k = 0; k < 4; ++k) y[ct][4 * g + k] = v[k];
        }
    gdn_stages<CT, NW, INVERSE, STAGE_BYTES>(y, p, smem, sc, wave, lane, [](char *) {});
    const int oy = pix / p.W, ox = pix - oy * p.W;
    store_tiles<CT>(y, p, n, oy, ox, h, valid);
}

// ---- layout conversion / quantiser kernels (HBM-bound, one element group per thread) ------------

// (n,h,w,c) uint8 -> C8 float / 255   (_autoencoders.py:542-545; true division, as torch does)
static __global__ void u8hwc_to_c8_kernel(const uint8_t *in, float *out, int N, int H, int W, int C, int planes) {
    const size_t total = (size_t)N * planes * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % ((size_t)H * W);
        const size_t np = i / ((size_t)H * W);
        const int plane = (int)(np % planes);
        const size_t n = np / planes;
        const uint8_t *src = in + (n * H * W + pix) * C;
        f32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c0 = plane * 8 + k, c1 = c0 + 4;
            lo[k] = c0 < C ? (float)src[c0] / 255.0f : 0.0f;
            hi[k] = c1 < C ? (float)src[c1] / 255.0f : 0.0f;
        }
        float *dst = out + i * 8;
        *(f32x4 *)dst = lo;
        *(f32x4 *)(dst + 4) = hi;
    }
}

// (sym != nullptr: fused dequantiser, value = float(sym) + median_c instead of in[])
static __global__ void nchw_to_c8_kernel(const float *in, float *out, int N, int C, int HW, int planes,
                                         const int32_t *sym = nullptr, const float *medians = nullptr) {
    const size_t total = (size_t)N * planes * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % HW;
        const size_t np = i / HW;
        const int plane = (int)(np % planes);
        const size_t n = np / planes;
        f32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c0 = plane * 8 + k, c1 = c0 + 4;
            if (sym) {
                lo[k] = c0 < C ? (float)sym[(n * C + c0) * HW + pix] + medians[c0] : 0.0f;
                hi[k] = c1 < C ? (float)sym[(n * C + c1) * HW + pix] + medians[c1] : 0.0f;
            } else {
                lo[k] = c0 < C ? in[(n * C + c0) * HW + pix] : 0.0f;
                hi[k] = c1 < C ? in[(n * C + c1) * HW + pix] : 0.0f;
            }
        }
        float *dst = out + i * 8;
        *(f32x4 *)dst = lo;
        *(f32x4 *)(dst + 4) = hi;
    }
}

static __global__ void c8_to_nchw_kernel(const float *in, float *out, int N, int C, int HW, int planes) {
    const size_t total = (size_t)N * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % HW;
        const size_t nc = i / HW;
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        out[i] = in[((n * planes + (c >> 3)) * HW + pix) * 8 + (c & 7)];
    }
}

// symbols = int(round_half_even(y - median_c))     (EntropyBottleneck.compress, Appendix A.3)
static __global__ void quantize_kernel(const float *y, const float *medians, int32_t *sym, int C, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        sym[i] = round_sym(y[i] - medians[c]);
    }
}

// The same quantiser for the PCIe export (cae_quantize_export): 16 elements per thread and iteration (four 16-byte
// loads in flight, then four 16-byte stores), so that a handful of workgroups keeps the link busy.  HW % 4 == 0.
static __global__ void __launch_bounds__(1024)
quantize_export4_kernel(const float *y, const float *medians, int32_t *sym, int C, int HW4, size_t total4) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = i0 + k * stride;
            if (i < total4) v[k] = ((const f32x4 *)y)[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = i0 + k * stride;
            if (i < total4) {
                const float med = medians[(i / HW4) % C];
                i32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = round_sym(v[k][e] - med);
                ((i32x4 *)sym)[i] = q;
            }
        }
    }
}

// y_hat = float(symbols) + median_c                (EntropyModel.dequantize)
static __global__ void dequantize_kernel(const int32_t *sym, const float *medians, float *y, int C, int HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        y[i] = (float)sym[i] + medians[c];
    }
}

// ---- factorized density (EntropyBottleneck.__call__, eval mode) -------------------------
// Per-channel density network of uniform hidden width R with K hidden layers, effective parameters
// (softplus(matrix), bias, tanh(factor)) packed per channel as
//   [M0 R][b0 R][t0 R]  { [Mi RxR][bi R][ti R] } i=1..K-1   [MK R][bK 1]
template <int R>
__device__ __forceinline__ float logits_cumulative(const float *__restrict__ p, int K, float x) {
    float v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        v[j] = p[j] * x + p[R + j];
        v[j] += p[2 * R + j] * tanhf(v[j]);
    }
    p += 3 * R;
    for (int i = 1; i < K; ++i) {
        float u[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < R; ++k) acc += p[j * R + k] * v[k];
            acc += p[R * R + j];
            u[j] = acc + p[R * R + R + j] * tanhf(acc);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = u[j];
        p += R * R + 2 * R;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) acc += p[k] * v[k];
    return acc + p[R];
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// One block per (channel, tile) plane: y_hat = round(y - m_c) + m_c, p = max(|sig(s*u) - sig(s*l)|, bound) with
// l,u = logits(y_hat -/+ 0.5), s = -sign(l + u); bits_part[n][c] = -sum log2(p) over the plane (fixed order).
template <int R>
__global__ void __launch_bounds__(256)
likelihood_kernel(const float *__restrict__ y, const float *__restrict__ medians, const float *__restrict__ params,
                  int per_channel, int K, float bound, int plain, int C, int HW, float *__restrict__ yhat,
                  float *__restrict__ lik, double *__restrict__ bits_part) {
    const int c = blockIdx.x, n = blockIdx.y;
    const float *pc = params + (size_t)c * per_channel;
    const float med = medians[c];
    const size_t base = ((size_t)n * C + c) * HW;
    double bits = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float q = rintf(y[base + i] - med) + med;
        const float lo = logits_cumulative<R>(pc, K, q - 0.5f);
        const float up = logits_cumulative<R>(pc, K, q + 0.5f);
        const float t = lo + up;
        const float sgn = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
        // plain: sigmoid(u) - sigmoid(l) (compressai >= 1.2.x); else the sign trick |sigmoid(s u) - sigmoid(s l)|
        float pr = plain ? sigmoidf(up) - sigmoidf(lo) : fabsf(sigmoidf(sgn * up) - sigmoidf(sgn * lo));
        pr = fmaxf(pr, bound);
        if (yhat) yhat[base + i] = q;
        if (lik) lik[base + i] = pr;
        bits -= (double)log2f(pr);
    }
    if (!bits_part) return;
    __shared__ double red[256];
    red[threadIdx.x] = bits;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bits_part[(size_t)n * C + c] = red[0];
}

// bits[n] = sum over channels of bits_part[n][c], fixed tree order (deterministic)
static __global__ void bits_reduce_kernel(const double *part, int C, double *bits) {
    __shared__ double red[256];
    const int n = blockIdx.x;
    double a = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) a += part[(size_t)n * C + c];
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bits[n] = red[0];
}

// per-tile sum of squared byte differences; one block row per tile, exact integer partial sums
static __global__ void tile_sse_kernel(const uint8_t *a, const uint8_t *b, size_t elems, unsigned long long *out) {
    const int tile = blockIdx.y;
    const uint8_t *pa = a + (size_t)tile * elems, *pb = b + (size_t)tile * elems;
    unsigned long long acc = 0;
    // 16-byte vector loads when both tiles are 16-byte aligned, else a plain byte loop (ragged tile sizes)
    const bool vec = ((((uintptr_t)pa) | ((uintptr_t)pb)) & 15) == 0;
    const size_t nvec = vec ? elems / 16 : 0;
    // sum (a - b)^2 = a.a + b.b - 2 a.b per dword with v_dot4_u32_u8 (exact in 32 bits: 8 x 255^2 per dword); two
    // vectors per thread and trip so that four loads are in flight
    auto sq = [](const uint4 &va, const uint4 &vb) {
        const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        unsigned same = 0, cross = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            same = __builtin_amdgcn_udot4(wa[k], wa[k], same, false);
            same = __builtin_amdgcn_udot4(wb[k], wb[k], same, false);
            cross = __builtin_amdgcn_udot4(wa[k], wb[k], cross, false);
        }
        return same - 2u * cross;
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < nvec; i += 2 * stride) {
        const uint4 va0 = ((const uint4 *)pa)[i], vb0 = ((const uint4 *)pb)[i];
        const uint4 va1 = ((const uint4 *)pa)[i + stride], vb1 = ((const uint4 *)pb)[i + stride];
        acc += sq(va0, vb0);
        acc += sq(va1, vb1);
    }
    if (i < nvec) acc += sq(((const uint4 *)pa)[i], ((const uint4 *)pb)[i]);
    if (vec) {
        if (blockIdx.x == 0)
            for (size_t i = nvec * 16 + threadIdx.x; i < elems; i += blockDim.x) {
                const int d = (int)pa[i] - (int)pb[i];
                acc += (unsigned)(d * d);
            }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
            const int d = (int)pa[i] - (int)pb[i];
            acc += (unsigned)(d * d);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + tile, acc);
}

// ---- SSIM of two uint8 HWC tile batches (skimage.metrics.structural_similarity as the reference's harness calls it,
// test_cae.py:55-57: 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, data range 255, 3-pixel border cropped,
// mean over channels).  The five window sums are exact integers; the per-pixel index is evaluated in float64.
// Block = 32x32 output pixels of one tile: rows of horizontal 7-sums in LDS, then vertical 7-sums per thread.
static __global__ void __launch_bounds__(256)
tile_ssim_kernel(const uint8_t *a, const uint8_t *b, int H, int W, int C, int bx_per_row, int blocks_per_tile,
                 double *part) {
    constexpr int T = 32, WIN = 7, IN = T + WIN - 1;
    __shared__ uint8_t sa[IN][IN + 2], sb[IN][IN + 2];
    __shared__ int hs[5][IN][T];
    __shared__ double red[256];
    const int tile = blockIdx.y, blk = blockIdx.x;
    const int by = blk / bx_per_row, bx = blk - by * bx_per_row;
    const int oy0 = by * T, ox0 = bx * T;       // output (= top-left input) coordinates of this block
    const int OH = H - WIN + 1, OW = W - WIN + 1;  // valid outputs
    const size_t base = (size_t)tile * H * W * C;
    const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
    const double cov_norm = 49.0 / 48.0;
    double acc = 0.0;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
        for (int i = threadIdx.x; i < IN * IN; i += 256) {
            const int r = i / IN, x = i - r * IN;
            const int iy = oy0 + r, ix = ox0 + x;
            const bool ok = iy < H && ix < W;
            const size_t off = base + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * C + c;
            sa[r][x] = ok ? a[off] : 0;
            sb[r][x] = ok ? b[off] : 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < IN * T; i += 256) {
            const int r = i / T, x = i - r * T;
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const int u = sa[r][x + k], v = sb[r][x + k];
                s0 += u;
                s1 += v;
                s2 += u * u;
                s3 += v * v;
                s4 += u * v;
            }
            hs[0][r][x] = s0;
            hs[1][r][x] = s1;
            hs[2][r][x] = s2;
            hs[3][r][x] = s3;
            hs[4][r][x] = s4;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < T * T; i += 256) {
            const int y = i / T, x = i - y * T;
            if (oy0 + y < OH && ox0 + x < OW) {
                int s[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < WIN; ++k)
#pragma unroll
                    for (int q = 0; q < 5; ++q) s[q] += hs[q][y + k][x];
                const double ux = s[0] / 49.0, uy = s[1] / 49.0;
                const double vx = cov_norm * (s[2] / 49.0 - ux * ux), vy = cov_norm * (s[3] / 49.0 - uy * uy);
                const double vxy = cov_norm * (s[4] / 49.0 - ux * uy);
                const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2;
                const double B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
                acc += (A1 * A2) / (B1 * B2);
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)tile * blocks_per_tile + blk] = red[0];
}

// ssim[tile] = sum of the block partials (fixed order) / number of (pixel, channel) samples
static __global__ void ssim_reduce_kernel(const double *part, int blocks_per_tile, double samples, double *ssim) {
    __shared__ double red[256];
    const int tile = blockIdx.x;
    double v = 0.0;
    for (int i = threadIdx.x; i < blocks_per_tile; i += 256) v += part[(size_t)tile * blocks_per_tile + i];
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) ssim[tile] = red[0] / samples;
}

// ---- mean CIE76 colour difference of two uint8 RGB tile batches (skimage.color rgb2lab + deltaE_cie76 as the
// reference's harness calls them, test_cae.py:21-45): sRGB -> linear (table of the 256 levels) -> XYZ (sRGB / D65
// matrix) -> L*a*b* (2-degree D65 white) -> Euclidean distance, all float64; per-block partial sums, fixed order.
__device__ __forceinline__ void rgb_to_lab(const double *lin, const uint8_t *px, double &L, double &A, double &B) {
    const double r = lin[px[0]], g = lin[px[1]], b = lin[px[2]];
    double x = (0.412453 * r + 0.357580 * g + 0.180423 * b) / 0.95047;
    double y = (0.212671 * r + 0.715160 * g + 0.072169 * b) / 1.0;
    double z = (0.019334 * r + 0.119193 * g + 0.950227 * b) / 1.08883;
    x = x > 0.008856 ? cbrt(x) : 7.787 * x + 16.0 / 116.0;
    y = y > 0.008856 ? cbrt(y) : 7.787 * y + 16.0 / 116.0;
    z = z > 0.008856 ? cbrt(z) : 7.787 * z + 16.0 / 116.0;
    L = 116.0 * y - 16.0;
    A = 500.0 * (x - y);
    B = 200.0 * (y - z);
}

static __global__ void __launch_bounds__(256)
tile_delta_e_kernel(const uint8_t *a, const uint8_t *b, size_t pixels, int blocks_per_tile, double *part) {
    __shared__ double lin[256];
    __shared__ double red[256];
    {
        const double v = threadIdx.x / 255.0;  // img_as_float, then the sRGB companding inverse
        lin[threadIdx.x] = v > 0.04045 ? pow((v + 0.055) / 1.055, 2.4) : v / 12.92;
    }
    __syncthreads();
    const int tile = blockIdx.y;
    const uint8_t *pa = a + (size_t)tile * pixels * 3, *pb = b + (size_t)tile * pixels * 3;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (size_t)blocks_per_tile * 256) {
        double l0, a0, b0, l1, a1, b1;
        rgb_to_lab(lin, pa + 3 * i, l0, a0, b0);
        rgb_to_lab(lin, pb + 3 * i, l1, a1, b1);
        acc += sqrt((l0 - l1) * (l0 - l1) + (a0 - a1) * (a0 - a1) + (b0 - b1) * (b0 - b1));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)tile * blocks_per_tile + blockIdx.x] = red[0];
}

// ---- MS-SSIM building blocks (pytorch_msssim.ms_ssim as the reference's harness calls it, test_cae.py:47-52):
// planar float32 images, 11-tap Gaussian window (sigma 1.5) applied separably without padding, K = (0.01, 0.03),
// data range 255; per level the means of the ssim map and of the contrast-structure map; 2x2 average pooling
// (zero padding of odd sizes, padded samples counted) between levels.
static __global__ void u8hwc_to_planes_kernel(const uint8_t *in, float *out, int N, int H, int W, int C) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % ((size_t)H * W);
        const size_t nc = i / ((size_t)H * W);
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        out[i] = (float)in[(n * H * W + pix) * C + c];
    }
}

static __global__ void avgpool2_kernel(const float *in, float *out, int planes, int H, int W, int OH, int OW) {
    const int ph = H & 1, pw = W & 1;
    const size_t total = (size_t)planes * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const size_t pl = i / ((size_t)OW * OH);
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * oy - ph + dy, x = 2 * ox - pw + dx;
                if (y >= 0 && y < H && x >= 0 && x < W) s += in[(pl * H + y) * W + x];
            }
        out[i] = s * 0.25f;
    }
}

// one block = 32x32 outputs of one plane; part[(plane * bpp + blk) * 2 + {0: ssim, 1: cs}] = partial sums (float64)
static __global__ void __launch_bounds__(256)
msssim_level_kernel(const float *X, const float *Y, int H, int W, int bx_per_row, int bpp, const float *win, double *part) {
    constexpr int T = 32, WIN = 11, IN = T + WIN - 1;
    __shared__ float sx[IN][IN + 1], sy[IN][IN + 1];
    __shared__ float vs[5][T][IN + 1];  // after the vertical pass: [quantity][out row][in col]
    __shared__ double red[2][256];
    __shared__ float g[WIN];
    if (threadIdx.x < WIN) g[threadIdx.x] = win[threadIdx.x];
    const int plane = blockIdx.y, blk = blockIdx.x;
    const int by = blk / bx_per_row, bx = blk - by * bx_per_row;
    const int oy0 = by * T, ox0 = bx * T;
    const int OH = H - WIN + 1, OW = W - WIN + 1;
    const float *px = X + (size_t)plane * H * W, *py = Y + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < IN * IN; i += 256) {
        const int r = i / IN, x = i - r * IN;
        const int iy = oy0 + r, ix = ox0 + x;
        const bool ok = iy < H && ix < W;
        sx[r][x] = ok ? px[(size_t)iy * W + ix] : 0.f;
        sy[r][x] = ok ? py[(size_t)iy * W + ix] : 0.f;
    }
    __syncthreads();
    // dimension 2 (rows) first, as pytorch_msssim's gaussian_filter does
    for (int i = threadIdx.x; i < T * IN; i += 256) {
        const int y = i / IN, x = i - y * IN;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float u = sx[y + k][x], v = sy[y + k][x], w = g[k];
            a0 += w * u;
            a1 += w * v;
            a2 += w * (u * u);
            a3 += w * (v * v);
            a4 += w * (u * v);
        }
        vs[0][y][x] = a0;
        vs[1][y][x] = a1;
        vs[2][y][x] = a2;
        vs[3][y][x] = a3;
        vs[4][y][x] = a4;
    }
    __syncthreads();
    const float C1 = (0.01f * 255.f) * (0.01f * 255.f), C2 = (0.03f * 255.f) * (0.03f * 255.f);
    double ssum = 0.0, csum = 0.0;
    for (int i = threadIdx.x; i < T * T; i += 256) {
        const int y = i / T, x = i - y * T;
        if (oy0 + y < OH && ox0 + x < OW) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < WIN; ++k)
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += g[k] * vs[q][y][x + k];
            const float mu1 = m[0], mu2 = m[1];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
            const float cs = (2.f * s12 + C2) / (s1 + s2 + C2);
            const float ss = ((2.f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
            ssum += (double)ss;
            csum += (double)cs;
        }
    }
    red[0][threadIdx.x] = ssum;
    red[1][threadIdx.x] = csum;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[((size_t)plane * bpp + blk) * 2 + 0] = red[0][0];
        part[((size_t)plane * bpp + blk) * 2 + 1] = red[1][0];
    }
}

// out[plane * 2 + j] = sum of the block partials (fixed order) / samples
static __global__ void msssim_reduce_kernel(const double *part, int bpp, double samples, double *out) {
    __shared__ double red[2][256];
    const int plane = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < bpp; i += 256) {
        a += part[((size_t)plane * bpp + i) * 2 + 0];
        b += part[((size_t)plane * bpp + i) * 2 + 1];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[(size_t)plane * 2 + 0] = red[0][0] / samples;
        out[(size_t)plane * 2 + 1] = red[1][0] / samples;
    }
}

static __global__ void u64_to_f64_kernel(const unsigned long long *in, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

}  // namespace cae
