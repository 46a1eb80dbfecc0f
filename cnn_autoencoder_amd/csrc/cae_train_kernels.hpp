// CDNA4 (gfx950) kernels of the TRAINING path (BASELINE config 5: bf16 convolutions, fp32 GDN).
//
// Reference semantics (file:line under /root/reference/src): the forward of DownsamplingUnit / UpsamplingUnit
// (models/tasks/_autoencoders.py:78-85, :204-211) and compressai's GDN (:29-30) under autograd, i.e. what
// `loss.backward()` of train_cae_ms.py:214 differentiates: convolution data / weight gradients and the GDN gradient.
//
// Layout "T": activations and gradients are channels-last, [N][H][W][Cp] with Cp = channels padded to 32 (zeros):
// bf16 for everything an MFMA consumes (activations into a convolution, gradients into dgrad / wgrad), fp32 for the
// convolution outputs the GDN reads and for gradients on their way into the GDN backward.  A pixel's channels are
// contiguous, so (i) a 16-channel MFMA k-step of a pixel is one 16-byte LDS read, (ii) an accumulator tile with the
// output channels on the lanes stores whole 128-byte rows, (iii) the weight gradient -- a contraction over PIXELS --
// reads its operands with the transposing LDS read ds_read_b64_tr_b16.
//
// Three MFMA kernels cover the six convolution products of a stride-2 layer and its transpose:
//   gather_gemm   out[pos][n] = sum_{tap, k} in[S pos + d_tap][k] W[tap][k][n]       v_mfma_f32_32x32x16_bf16
//                 S = 2: strided correlation      -> conv forward (reflect padding), deconv data gradient (zero padding)
//                 S = 1, one launch per output parity: its transpose
//                                                 -> deconv forward (cropped), conv data gradient (extended domain,
//                                                    the reflect fold is done by the consumer: fold_read)
//   wgrad         gW[tap][a][b] = sum_pos X[2 pos + d_tap][a] Y[pos][b]              (both operands transposed reads)
//   gdn_gemm_a / gdn_gemm_b   the C x C contractions of GDN / IGDN forward and backward, exact fp32
//                 (v_mfma_f32_32x32x2_f32), with the element-wise parts fused as prologue / epilogue.
#pragma once
#include "cae_kernels.hpp"

#ifndef GG8_ABL
#define GG8_ABL 0  // timing experiments on gg8_kernel (results wrong): bit 0 no MFMA, 1 no LDS-DMA after the first slice, 2 no stores
#endif
#ifndef GG_ABL
#define GG_ABL 0  // timing experiments on gather_gemm (results wrong): bit 0 no MFMA, 1 no LDS-DMA after the first slice,
#endif            // 2 operand reads from one fixed LDS address, 3 no output stores

namespace cae {
namespace tr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_TAPS = 25;

// ---------------------------------------------------------------------------------------------------------------
// weights -> MFMA B fragments, on the device (the weights change every optimiser step):
//   packed[q][t][nt][s][lane][e] = W(k = 32q + 16s + 8(lane>>5) + e, n = 32nt + (lane&31), tap t)  as bf16,
//   W(k, n, t) = w[k * sk + n * sn + t]  (a (cout,cin,k,k) or (cin,cout,k,k) tensor read with either dim contracted)
// ---------------------------------------------------------------------------------------------------------------
static __global__ void pack_weights_kernel(const float *w, __bf16 *out, int Kc, int Nc, int kk, long sk, long sn,
                                           int kchunks, int NT) {
    const size_t total = (size_t)kchunks * kk * NT * 2 * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), s = (int)((i >> 9) & 1);
        size_t r = i >> 10;
        const int nt = (int)(r % NT);
        r /= NT;
        const int t = (int)(r % kk);
        const int q = (int)(r / kk);
        const int n = 32 * nt + (lane & 31), k = 32 * q + 16 * s + 8 * (lane >> 5) + e;
        out[i] = (__bf16)((k < Kc && n < Nc) ? w[(size_t)k * sk + (size_t)n * sn + t] : 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gather_gemm
//   block = 4 waves, 16 x 16 logical positions; wave w: rows 4w .. 4w+3 = two m-tiles of 2 rows x 16 columns.
//   per 32-channel chunk of the contraction: halo [quarter][row][col] x 16 B in LDS (LDS-DMA, padding resolved in
//   the source address), weights of `taps_per_stage` taps at a time.
// ---------------------------------------------------------------------------------------------------------------
struct GGArgs {
    const void *in;     // bf16 [N][IH][IW][Ck]
    float *out32;       // fp32 [N][OH][OW][Cn] or null
    void *out16;        // bf16 [N][OH][OW][Cn] or null
    const void *wp;     // packed weights
    const float *bias;  // [Cn] or null
    const void *zero;   // >= 16 B of zeros
    int N, IH, IW, Ck, Cn, OH, OW;
    int LH, LW;         // logical position grid
    int S;              // input pixel of position (i, j) and tap t: (S i + dy[t], S j + dx[t])
    int SO, oy0, ox0;   // output pixel of position (i, j): (SO i + oy0, SO j + ox0)
    int reflect;        // 1: reflect padding of the input, 0: zeros outside
    int act;            // activation on the way out: 0 none, 1 LeakyReLU(0.01), 2 ReLU (_autoencoders.py:19-34)
    int ntaps, ktaps;   // taps of this launch, taps of the packed weights (KS * KS)
    int dymin, dxmin, HR, HC;
    int taps_per_stage;
    unsigned m_plane, m_hc;  // ceil(2^32 / (HR * HC)), ceil(2^32 / HC): exact quotients of piece indices by __umulhi
    int nq;             // pipelined form: 8-channel quarters of the contraction per staged slice (4 = a chunk, 2 = half), else 0
    int npb;            // gg8_kernel: consecutive samples a block walks (same tile of each)
    int nt0, nt_all;    // gg8_kernel: first n-tile of this launch and the n-tiles of the packed weights / output rows (0: NT):
                        // a 192-channel output goes as two launches of three n-tiles
    int tiles_x, tiles_y;
    short dy[MAX_TAPS], dx[MAX_TAPS], wt[MAX_TAPS];
};

template <int NT, bool PIPE>
__global__ void __launch_bounds__(256, 1) gather_gemm_kernel(const GGArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int i0 = ty * 16, j0 = tx * 16;

    const int plane = p.HR * p.HC;  // pieces per quarter
    const int pieces = 4 * plane;
    const int halo_instr = (pieces + 63) / 64;
    char *halo = smem;
    char *wbuf = smem + (size_t)halo_instr * 1024;
    const char *in_n = (const char *)p.in + (size_t)n * p.IH * p.IW * p.Ck * 2;

    constexpr int MAXP = 20;  // ceil(4 * 35 * 35 / 64 / 4)
    // this thread's halo pieces (same for every chunk / slice): byte offset inside the sample, or ~0 = zeros
    const int npieces = PIPE ? p.nq * plane : pieces;
    unsigned hoff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (wave + i * 4) * 64 + lane;
        pc = pc < npieces ? pc : npieces - 1;
        // (a block lives for one tile: twenty pairs of integer divisions by run-time values here were a quarter of its time)
        const int quarter = (int)__umulhi((unsigned)pc, p.m_plane);
        const int rem = pc - quarter * plane;
        const int r = (int)__umulhi((unsigned)rem, p.m_hc), c = rem - r * p.HC;
        int iy = p.S * i0 + p.dymin + r, ix = p.S * j0 + p.dxmin + c;
        bool ok = true;
        if (p.reflect) {
            iy = reflect_idx(iy, p.IH);
            ix = reflect_idx(ix, p.IW);
        } else {
            ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        }
        hoff[i] = ok ? (unsigned)((iy * p.IW + ix) * p.Ck * 2 + quarter * 16) : 0xFFFFFFFFu;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float b = p.bias ? p.bias[32 * nt + m] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][nt][r] = b;
            acc[1][nt][r] = b;
        }
    }

    // A operand of m-tile pt: position (4w + 2pt + (m>>4), m&15)
    const int arow = p.S * (4 * wave + (m >> 4)), acol = p.S * (m & 15);
    if constexpr (PIPE) {
        // PIPELINED: the contraction is walked in slices of nq quarters (32 or 16 channels); a slice = its halo + the
        // weights of ALL taps, double-buffered: slice sl + 1 is on its way by LDS-DMA while the MFMAs of slice sl run,
        // one barrier per slice.  (The first form loaded a chunk, waited, consumed it: MFMA pipes busy 0.09 of the time.)
        const int nq = p.nq, ksteps = nq / 2, per_chunk = 4 / nq;
        const int qplane = plane;
        const int hpieces = nq * qplane, h_instr = (hpieces + 63) / 64;
        const int w_instr = p.ntaps * NT * ksteps;
        const size_t buf_bytes = (size_t)(h_instr + w_instr) * 1024;
        const int slices = p.Ck / (8 * nq);
        // this thread's weight pieces of a slice, relative to the slice's first k-step: [tap][nt][k-step of the slice]
        constexpr int MAXW = 10;  // ceil(9 taps * 4 tiles * 2 k-steps / 4 waves / 2) .. more taps: the loop below
        unsigned woff[MAXW];
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            int f = wave + 4 * i;
            f = f < w_instr ? f : w_instr - 1;
            const int tl = ksteps == 2 ? f / (NT * 2) : f / NT, rest = f - tl * (NT * ksteps);  // (compile-time divisors)
            const int nt = ksteps == 2 ? rest >> 1 : rest, s = rest - nt * ksteps;
            woff[i] = (unsigned)(((p.wt[tl] * (NT * 2) + nt * 2 + s) * 1024) + lane * 16);
        }
        const unsigned chunk_bytes = (unsigned)p.ktaps * (NT * 2) * 1024;
        auto issue = [&](int sl, char *buf) {
            const int q = sl / per_chunk, s0 = (sl - q * per_chunk) * ksteps;
            const char *in_s = in_n + sl * (16 * nq);
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int j = wave + i * 4;
                if (j < h_instr) glds16(hoff[i] != 0xFFFFFFFFu ? (const void *)(in_s + hoff[i]) : p.zero, buf + j * 1024);
            }
            char *wb = buf + (size_t)h_instr * 1024;
            const char *wsrc = (const char *)p.wp + (size_t)q * chunk_bytes + s0 * 1024;
#pragma unroll
            for (int i = 0; i < MAXW; ++i) {
                const int f = wave + 4 * i;
                if (f < w_instr) glds16(wsrc + woff[i], wb + f * 1024);
            }
            for (int f = wave + 4 * MAXW; f < w_instr; f += 4) {  // (k = 5: more than 40 pieces)
                const int tl = f / (NT * ksteps), rest = f - tl * (NT * ksteps);
                const int nt = rest / ksteps, s = rest - nt * ksteps;
                glds16(wsrc + ((p.wt[tl] * (NT * 2) + nt * 2 + s) * 1024) + lane * 16, wb + f * 1024);
            }
        };
        issue(0, smem);
        for (int sl = 0; sl < slices; ++sl) {
            char *cur = smem + (size_t)(sl & 1) * buf_bytes;
            wait_vm0();
            __syncthreads();  // slice sl landed; the other buffer's readers (slice sl - 1) are done
            if (sl + 1 < slices && !(GG_ABL & 2)) issue(sl + 1, smem + (size_t)((sl + 1) & 1) * buf_bytes);
            const char *wcur = cur + (size_t)h_instr * 1024;
            for (int t = 0; t < p.ntaps; ++t) {
                const int hr = arow + p.dy[t] - p.dymin, hc = acol + p.dx[t] - p.dxmin;
                for (int s = 0; s < ksteps; ++s) {
                    const char *ab = (GG_ABL & 4) ? cur + lane * 16 : cur + ((((2 * s + h) * p.HR + hr) * p.HC + hc) * 16);
                    const bf16x8 a0 = *(const bf16x8 *)ab;
                    const bf16x8 a1 = *(const bf16x8 *)(ab + 2 * p.S * p.HC * 16);  // m-tile 1: two rows down
                    const char *wb = wcur + ((t * NT) * ksteps + s) * 1024 + lane * 16;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 b = *(const bf16x8 *)(wb + nt * ksteps * 1024);
                        if (GG_ABL & 1) {
                            acc[0][nt][0] += (float)a0[0] * (float)b[0];
                            acc[1][nt][0] += (float)a1[0] * (float)b[1];
                        } else {
                            acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][nt], 0, 0, 0);
                            acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][nt], 0, 0, 0);
                        }
                    }
                }
            }
        }
    } else {
    const int chunks = p.Ck / 32;
    for (int q = 0; q < chunks; ++q) {
        __syncthreads();  // the previous chunk's reads of the halo are done
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * 4;
            if (j < halo_instr) {
                const void *src = hoff[i] != 0xFFFFFFFFu ? (const void *)(in_n + hoff[i] + q * 64) : p.zero;
                glds16(src, halo + j * 1024);
            }
        }
        for (int t0 = 0; t0 < p.ntaps; t0 += p.taps_per_stage) {
            const int nst = min(p.taps_per_stage, p.ntaps - t0);
            if (t0 > 0) __syncthreads();  // the previous stage's reads of the weights are done
            const int w_instr = nst * NT * 2;
            for (int f = wave; f < w_instr; f += 4) {
                const int tl = f / (NT * 2), rest = f - tl * (NT * 2);
                const char *src = (const char *)p.wp +
                                  (((size_t)q * p.ktaps + p.wt[t0 + tl]) * (NT * 2) + rest) * 1024 + lane * 16;
                glds16(src, wbuf + f * 1024);
            }
            wait_vm0();
            __syncthreads();
            for (int tl = 0; tl < nst; ++tl) {
                const int t = t0 + tl;
                const int hr = arow + p.dy[t] - p.dymin, hc = acol + p.dx[t] - p.dxmin;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char *ab = halo + ((((2 * s + h) * p.HR + hr) * p.HC + hc) * 16);
                    const bf16x8 a0 = *(const bf16x8 *)ab;
                    const bf16x8 a1 = *(const bf16x8 *)(ab + 2 * p.S * p.HC * 16);  // m-tile 1: two rows down
                    const char *wb = wbuf + ((tl * NT) * 2 + s) * 1024 + lane * 16;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 b = *(const bf16x8 *)(wb + nt * 2048);
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    }

    // D: lane = output channel 32nt + m, register r = position acc_row(r) + 4h of the m-tile
    static_for<2>([&](auto pt_tag) {
        constexpr int pt = decltype(pt_tag)::value;
        static_for<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            const int mp = acc_row(r) + 4 * h;
            const int li = i0 + 4 * wave + 2 * pt + (mp >> 4), lj = j0 + (mp & 15);
            const int oy = p.SO * li + p.oy0, ox = p.SO * lj + p.ox0;
            if (li < p.LH && lj < p.LW && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW && (!(GG_ABL & 8) || p.N < 0)) {
                const size_t base = (((size_t)n * p.OH + oy) * p.OW + ox) * p.Cn + m;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float v = acc[pt][nt][r];
                    if (p.act) v = v > 0.0f ? v : (p.act == 1 ? 0.01f * v : 0.0f);
                    if (p.out32) p.out32[base + 32 * nt] = v;
                    if (p.out16) ((__bf16 *)p.out16)[base + 32 * nt] = (__bf16)v;
                }
            }
        });
    });
}

// ---------------------------------------------------------------------------------------------------------------
// gg8: the gather-GEMM with TWO waves per SIMD and blocks that live for several samples (round 3, late).
//   gather_gemm_kernel above ran at 0.11 of the bf16 MFMA peak on the canonical 128 -> 128 layers: one wave per SIMD walks
//   DMA issue, barrier, operand reads and MFMAs in order, the tap loop has a run-time trip count (scalar loads of the tap
//   table inside it, waits that also drain the LDS queue), and a block lives for ONE 16 x 16 tile, so its prologue (the
//   source offset of every halo piece), the pipeline fill and the drain are paid per tile (26 us of a 70-us block with MFMAs,
//   DMA, operand addressing and stores all ablated, profiles/r03_experiments.md 4).
//   Here: 8 waves, wave w = rows 2w, 2w+1 of the tile = ONE m-tile x all NT n-tiles (1 + NT operand reads per NT MFMAs; the
//   LDS feeds that at 0.8 of the MFMA peak); tap count NTAPS and slice width NQ are template parameters, the slice body is fully
//   unrolled with the tap offsets in scalar registers; a block walks the same tile of p.npb consecutive samples -- the halo
//   offsets depend on (ty, tx) only -- as one continuous slice pipeline: sample n + 1's first slice is in flight while
//   sample n's last slice multiplies and its tile is stored.
// ---------------------------------------------------------------------------------------------------------------
// LDS slot of halo piece p (16 bytes): the A-operand reads of a wave step by 2 or 4 pieces from lane to lane (64 bytes: a pixel
// stride of S * NQ pieces), i.e. they touch 4 of the 16 bank groups -- the PMC pass showed 2.7 - 4.1 conflict cycles per 7 - 8
// LDS-active cycles in gg8 / gg8t.  XOR-ing bits 4..5 of the piece index into bits 0..1 spreads each run of 16 lanes over all 16
// groups; the LDS-DMA writes lane-linear slots, so the same involution picks the SOURCE piece of every slot when staging.
__device__ __forceinline__ int gg8_swz(int piece) { return piece ^ ((piece >> 4) & 3); }

template <int NT, int NQ, int NTAPS>
__global__ void __launch_bounds__(512, 1) gg8_kernel(const GGArgs p) {
    constexpr int NW = 8, KSTEPS = NQ / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n0 = (bid / p.tiles_y) * p.npb;
    const int nsamp = min(p.npb, p.N - n0);
    const int i0 = ty * 16, j0 = tx * 16;

    const int plane = p.HR * p.HC;  // pieces per 8-channel quarter
    const int hpieces = NQ * plane, h_instr = (hpieces + 63) / 64;
    constexpr int W_INSTR = NTAPS * NT * KSTEPS;
    const size_t buf_bytes = (size_t)(h_instr + W_INSTR) * 1024;
    const size_t sample_bytes = (size_t)p.IH * p.IW * p.Ck * 2;

    // this thread's halo pieces of a slice: byte offset inside the sample, or ~0 = zeros
    constexpr int MAXP = 10;  // 80 LDS-DMA instructions (80 KiB of halo) over 8 waves
    unsigned hoff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = gg8_swz((wave + i * NW) * 64 + lane);  // (the piece that lives in this lane's slot)
        pc = pc < hpieces ? pc : hpieces - 1;
        // pixel-major: the NQ 16-byte quarters of a pixel's slice sit on consecutive lanes = one 32- or 64-byte run of its
        // channel vector (quarter-major, one lane per cache line and every line fetched NQ times, held the kernel at the
        // rate of its LDS-DMA: 0.38 ms with, 0.17 ms without the copies on the 128 -> 128 layer)
        const int quarter = pc % NQ, rem = pc / NQ;
        const int r = (int)__umulhi((unsigned)rem, p.m_hc), c = rem - r * p.HC;
        int iy = p.S * i0 + p.dymin + r, ix = p.S * j0 + p.dxmin + c;
        bool ok = true;
        if (p.reflect) {
            iy = reflect_idx(iy, p.IH);
            ix = reflect_idx(ix, p.IW);
        } else {
            ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        }
        hoff[i] = ok ? (unsigned)((iy * p.IW + ix) * p.Ck * 2 + quarter * 16) : 0xFFFFFFFFu;
    }
    // this thread's weight pieces of a slice, relative to the slice's first k-step: [tap][nt][k-step of the slice]
    constexpr int MAXW = (W_INSTR + NW - 1) / NW;
    const int nta = p.nt_all ? p.nt_all : NT;  // n-tiles of the packed weights (this launch covers NT of them from p.nt0)
    unsigned woff[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        int f = wave + NW * i;
        f = f < W_INSTR ? f : W_INSTR - 1;
        const int tl = f / (NT * KSTEPS), rest = f - tl * (NT * KSTEPS);
        const int nt = rest / KSTEPS, ks = rest - nt * KSTEPS;
        woff[i] = (unsigned)(((p.wt[tl] * (nta * 2) + (p.nt0 + nt) * 2 + ks) * 1024) + lane * 16);
    }
    const unsigned chunk_bytes = (unsigned)p.ktaps * (nta * 2) * 1024;
    const int slices = p.Ck / (8 * NQ);
    constexpr int PER_CHUNK = 4 / NQ;

    auto issue = [&](int ns, int sl, char *buf) {  // slice sl of the block's sample ns
        const int q = sl / PER_CHUNK, s0 = (sl - q * PER_CHUNK) * KSTEPS;
        const char *in_s = (const char *)p.in + (size_t)(n0 + ns) * sample_bytes + sl * (16 * NQ);
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * NW;
            if (j < h_instr) glds16(hoff[i] != 0xFFFFFFFFu ? (const void *)(in_s + hoff[i]) : p.zero, buf + j * 1024);
        }
        char *wb = buf + (size_t)h_instr * 1024;
        const char *wsrc = (const char *)p.wp + (size_t)q * chunk_bytes + s0 * 1024;
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            const int f = wave + NW * i;
            if (f < W_INSTR) glds16(wsrc + woff[i], wb + f * 1024);
        }
    };

    f32x16 acc[NT];
    auto init_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float b = p.bias ? p.bias[32 * (p.nt0 + nt) + m] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = b;
        }
    };
    init_acc();

    // A operand: halo [row][col][quarter]; position (2 wave + (m >> 4), m & 15), quarter 2 ks + h, tap offset per tap (uniform)
    const int a_base = ((p.S * (2 * wave + (m >> 4))) * p.HC + p.S * (m & 15)) * NQ + h;  // (in pieces)
    constexpr int a_kstep = 2;

    const int total = nsamp * slices;
    issue(0, 0, smem);
    int ns = 0, sl = 0;
    for (int it = 0; it < total; ++it) {
        char *cur = smem + (size_t)(it & 1) * buf_bytes;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slice `it` landed; the other buffer's readers are done
        if (it + 1 < total && !(GG8_ABL & 2)) {
            const bool wrap = sl + 1 == slices;
            issue(wrap ? ns + 1 : ns, wrap ? 0 : sl + 1, smem + (size_t)((it + 1) & 1) * buf_bytes);
        }
        const char *wb = cur + (size_t)h_instr * 1024 + lane * 16;
        static_for<NTAPS>([&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            const int toff = ((p.dy[t] - p.dymin) * p.HC + (p.dx[t] - p.dxmin)) * NQ;  // (uniform; kernel-argument loads, hoisted)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 a = *(const bf16x8 *)(cur + gg8_swz(a_base + toff + ks * a_kstep) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 b = *(const bf16x8 *)(wb + ((t * NT + nt) * KSTEPS + ks) * 1024);
                    if (GG8_ABL & 1)
                        acc[nt][0] += (float)a[0] * (float)b[0];
                    else
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
                }
            }
        });
        if (++sl == slices) {
            // D: lane = output channel 32 nt + m, register r = position acc_row(r) + 4h of the wave's m-tile
            const int n = n0 + ns;
            static_for<16>([&](auto r_tag) __attribute__((always_inline)) {
                constexpr int r = decltype(r_tag)::value;
                const int mp = acc_row(r) + 4 * h;
                const int li = i0 + 2 * wave + (mp >> 4), lj = j0 + (mp & 15);
                const int oy = p.SO * li + p.oy0, ox = p.SO * lj + p.ox0;
                if (li < p.LH && lj < p.LW && oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW && (!(GG8_ABL & 4) || p.N < 0)) {
                    const size_t base = (((size_t)n * p.OH + oy) * p.OW + ox) * p.Cn + 32 * p.nt0 + m;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[nt][r];
                        if (p.act) v = v > 0.0f ? v : (p.act == 1 ? 0.01f * v : 0.0f);
                        if (p.out32) p.out32[base + 32 * nt] = v;
                        if (p.out16) ((__bf16 *)p.out16)[base + 32 * nt] = (__bf16)v;
                    }
                }
            });
            init_acc();
            sl = 0;
            ++ns;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gg8t: the TRANSPOSE of the strided correlation (k = 3) with its four output parities in ONE launch.
//   As four gg8 launches (1 + 2 + 2 + 4 taps) the transposed convolutions ran at half the rate of the strided ones for the same
//   FLOP: every launch stages the same input halo again and its blocks live for one to four taps only.  Here a block = 16 x 16
//   logical positions (i, j) x 64 output channels (two n-tiles; blockIdx.y picks them) and produces all four output pixels
//   (2i + py, 2j + px) of every position: 9 taps per 32-channel slice, 4 x 2 accumulator tiles per wave (128 registers), one
//   17 x 17 halo.  The taps come parity-major from the host (counts {1, 2, 2, 4} for the cropped form = ConvTranspose2d forward,
//   {4, 2, 2, 1} for the extended form = data gradient of the strided convolution), so tap -> parity is static.
// ---------------------------------------------------------------------------------------------------------------
template <bool EXT>
__host__ __device__ constexpr int gg8t_parity(int t) {
    return EXT ? (t < 4 ? 0 : (t < 6 ? 1 : (t < 8 ? 2 : 3))) : (t < 1 ? 0 : (t < 3 ? 1 : (t < 5 ? 2 : 3)));
}

template <bool EXT>
__global__ void __launch_bounds__(512, 1) gg8t_kernel(const GGArgs p) {
    constexpr int NW = 8, NT = 2, NQ = 4, KSTEPS = 2, NTAPS = 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n0 = (bid / p.tiles_y) * p.npb;
    const int nsamp = min(p.npb, p.N - n0);
    const int i0 = ty * 16, j0 = tx * 16;
    const int nt0 = NT * blockIdx.y, nta = p.Cn / 32;

    const int plane = p.HR * p.HC;
    const int hpieces = NQ * plane, h_instr = (hpieces + 63) / 64;
    constexpr int W_INSTR = NTAPS * NT * KSTEPS;
    const size_t buf_bytes = (size_t)(h_instr + W_INSTR) * 1024;
    const size_t sample_bytes = (size_t)p.IH * p.IW * p.Ck * 2;

    constexpr int MAXP = 4;  // 4 x 18 x 18 pieces / 64 / 8 waves
    unsigned hoff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = gg8_swz((wave + i * NW) * 64 + lane);
        pc = pc < hpieces ? pc : hpieces - 1;
        const int quarter = pc % NQ, rem = pc / NQ;  // pixel-major, swizzled slots (see gg8_kernel)
        const int r = (int)__umulhi((unsigned)rem, p.m_hc), c = rem - r * p.HC;
        const int iy = i0 + p.dymin + r, ix = j0 + p.dxmin + c;
        const bool ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        hoff[i] = ok ? (unsigned)((iy * p.IW + ix) * p.Ck * 2 + quarter * 16) : 0xFFFFFFFFu;
    }
    constexpr int MAXW = (W_INSTR + NW - 1) / NW;
    unsigned woff[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        int f = wave + NW * i;
        f = f < W_INSTR ? f : W_INSTR - 1;
        const int tl = f / (NT * KSTEPS), rest = f - tl * (NT * KSTEPS);
        const int nt = rest / KSTEPS, ks = rest - nt * KSTEPS;
        woff[i] = (unsigned)(((p.wt[tl] * (nta * 2) + (nt0 + nt) * 2 + ks) * 1024) + lane * 16);
    }
    const unsigned chunk_bytes = (unsigned)p.ktaps * (nta * 2) * 1024;
    const int slices = p.Ck / 32;

    auto issue = [&](int ns, int sl, char *buf) {
        const char *in_s = (const char *)p.in + (size_t)(n0 + ns) * sample_bytes + sl * 64;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * NW;
            if (j < h_instr) glds16(hoff[i] != 0xFFFFFFFFu ? (const void *)(in_s + hoff[i]) : p.zero, buf + j * 1024);
        }
        char *wb = buf + (size_t)h_instr * 1024;
        const char *wsrc = (const char *)p.wp + (size_t)sl * chunk_bytes;
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            const int f = wave + NW * i;
            if (f < W_INSTR) glds16(wsrc + woff[i], wb + f * 1024);
        }
    };

    f32x16 acc[4][NT];
    auto init_acc = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float b = p.bias ? p.bias[32 * (nt0 + nt) + m] : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][nt][r] = b;
        }
    };
    init_acc();

    const int a_base = ((2 * wave + (m >> 4)) * p.HC + (m & 15)) * NQ + h;  // (in pieces)
    const int total = nsamp * slices;
    issue(0, 0, smem);
    int ns = 0, sl = 0;
    for (int it = 0; it < total; ++it) {
        char *cur = smem + (size_t)(it & 1) * buf_bytes;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (it + 1 < total) {
            const bool wrap = sl + 1 == slices;
            issue(wrap ? ns + 1 : ns, wrap ? 0 : sl + 1, smem + (size_t)((it + 1) & 1) * buf_bytes);
        }
        const char *wb = cur + (size_t)h_instr * 1024 + lane * 16;
        static_for<NTAPS>([&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            constexpr int par = gg8t_parity<EXT>(t);
            const int toff = ((p.dy[t] - p.dymin) * p.HC + (p.dx[t] - p.dxmin)) * NQ;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 a = *(const bf16x8 *)(cur + gg8_swz(a_base + toff + ks * 2) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bf16x8 b = *(const bf16x8 *)(wb + ((t * NT + nt) * KSTEPS + ks) * 1024);
                    acc[par][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[par][nt], 0, 0, 0);
                }
            }
        });
        if (++sl == slices) {
            const int n = n0 + ns;
            static_for<4>([&](auto q_tag) __attribute__((always_inline)) {
                constexpr int par = decltype(q_tag)::value, py = par >> 1, px = par & 1;
                static_for<16>([&](auto r_tag) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_tag)::value;
                    const int mp = acc_row(r) + 4 * h;
                    const int li = i0 + 2 * wave + (mp >> 4), lj = j0 + (mp & 15);
                    const int oy = 2 * li + py, ox = 2 * lj + px;
                    if (oy < p.OH && ox < p.OW) {
                        const size_t base = (((size_t)n * p.OH + oy) * p.OW + ox) * p.Cn + 32 * nt0 + m;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            float v = acc[par][nt][r];
                            if (p.act) v = v > 0.0f ? v : (p.act == 1 ? 0.01f * v : 0.0f);
                            if (p.out32) p.out32[base + 32 * nt] = v;
                            if (p.out16) ((__bf16 *)p.out16)[base + 32 * nt] = (__bf16)v;
                        }
                    }
                });
            });
            init_acc();
            sl = 0;
            ++ns;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad:  gW[tap][a][b] += sum over positions (n, i, j) of X[n][2i + dy][2j + dx][a] * Y[n][i][j][b]
//   block = 4 waves; one 32-channel a-tile, <= 9 taps, every b-tile (wave w: b-tiles w, w + 4, ...);
//   walks position tiles of 8 x 16 (K = 128), X halo and Y tile pixel-major in LDS, operands by ds_read_b64_tr_b16;
//   partial sums stay in registers over the block's whole position range, one atomic flush at the end.
// ---------------------------------------------------------------------------------------------------------------
struct WGArgs {
    const void *x;     // bf16 [N][H][W][Ca]   (the stride-2-sampled tensor)
    const void *y;     // bf16 [N][OH][OW][Cb]
    float *gw;         // fp32 [kk][Ca][Cb], zeroed by the caller
    const void *zero;
    int N, H, W, Ca, OH, OW, Cb;
    int reflect;       // padding of X: 1 reflect, 0 zeros
    unsigned m_hc, m_ypp;  // ceil(2^32 / HC), ceil(2^32 / (Cb / 8))
    int S;             // X is sampled at S * position + tap offset: 2 (the strided layers) or 1 (the stride-1 pre-convolutions)
    int kk;            // taps of the layer
    int dymin, dxmin, HR, HC;
    int tiles_x, tiles_y, total_tiles;
    int cb0, Cbs;      // wgrad8_kernel: first b channel of this launch and the channel count of Y / gW rows (Cb = this launch's share)
    short dy[MAX_TAPS], dx[MAX_TAPS];
};

template <int NB>  // b-tiles per wave
__global__ void __launch_bounds__(256, 1) wgrad_kernel(const WGArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int at = blockIdx.y;
    const int tap0 = blockIdx.z * 9;
    const int ntaps = min(9, p.kk - tap0);
    const int nbt = p.Cb / 32;

    const int x_pieces = p.HR * p.HC * 4;  // pixel-major: 4 x 16 B per pixel (32 channels of the a-tile)
    const int x_instr = (x_pieces + 63) / 64;
    const int ypp = p.Cb / 8;              // 16-byte pieces per position
    const int y_pieces = 128 * ypp;
    const int y_instr = (y_pieces + 63) / 64;
    char *xbuf = smem;
    char *ybuf = smem + (size_t)x_instr * 1024;

    f32x16 acc[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.0f;

    // transposed reads: 16-lane group g: channel half g&1, k-group g>>1; lane 4q+pp of the group addresses row q
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int col_off = (16 * (g & 1) + 4 * pp) * 2;  // bytes inside a 32-channel record

    for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
        int rem = tile;
        const int tx = rem % p.tiles_x;
        rem /= p.tiles_x;
        const int ty = rem % p.tiles_y;
        const int n = rem / p.tiles_y;
        const int i0 = ty * 8, j0 = tx * 16;
        __syncthreads();  // the previous tile's reads are done
        const char *x_n = (const char *)p.x + (size_t)n * p.H * p.W * p.Ca * 2;
        for (int j = wave; j < x_instr; j += 4) {
            int pc = j * 64 + lane;
            pc = pc < x_pieces ? pc : x_pieces - 1;
            const int pix = pc >> 2, quarter = pc & 3;
            const int r = (int)__umulhi((unsigned)pix, p.m_hc), c = pix - r * p.HC;
            int iy = p.S * i0 + p.dymin + r, ix = p.S * j0 + p.dxmin + c;
            bool ok = true;
            if (p.reflect) {
                iy = reflect_idx(iy, p.H);
                ix = reflect_idx(ix, p.W);
            } else {
                ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            }
            const void *src = ok ? (const void *)(x_n + (((size_t)iy * p.W + ix) * p.Ca + 32 * at) * 2 + quarter * 16)
                                 : p.zero;
            glds16(src, xbuf + j * 1024);
        }
        const char *y_n = (const char *)p.y + (size_t)n * p.OH * p.OW * p.Cb * 2;
        for (int j = wave; j < y_instr; j += 4) {
            int pc = j * 64 + lane;
            pc = pc < y_pieces ? pc : y_pieces - 1;
            const int pos = (int)__umulhi((unsigned)pc, p.m_ypp), part = pc - pos * ypp;
            const int i = i0 + (pos >> 4), jj = j0 + (pos & 15);
            const bool ok = i < p.OH && jj < p.OW;  // positions outside contribute zero
            const void *src = ok ? (const void *)(y_n + (((size_t)i * p.OW + jj) * p.Cb) * 2 + part * 16) : p.zero;
            glds16(src, ybuf + j * 1024);
        }
        wait_vm0();
        __syncthreads();
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {  // 16 positions: tile row ks, columns 0..15
            // this lane's rows of the two 4-row blocks: columns 8(g>>1) + 4rr + q4
            const int lj0 = 8 * (g >> 1) + q4, lj1 = lj0 + 4;
            bf16x8 bfr[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int bt = wave + 4 * b;
                const int btc = bt < nbt ? bt : 0;
                const char *yb = ybuf + (size_t)(16 * ks) * p.Cb * 2 + 64 * btc + col_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(yb + (size_t)lj0 * p.Cb * 2));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)(yb + (size_t)lj1 * p.Cb * 2));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[b] = __builtin_bit_cast(bf16x8, v);
            }
            static_for<9>([&](auto t_tag) {
                constexpr int t = decltype(t_tag)::value;
                if (t < ntaps) {
                    const int hr = p.S * ks + p.dy[tap0 + t] - p.dymin;
                    const int hc = p.dx[tap0 + t] - p.dxmin;
                    const char *xb = xbuf + (size_t)(hr * p.HC + hc) * 64 + col_off;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(xb + p.S * lj0 * 64));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(xb + p.S * lj1 * 64));
                    const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 a = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[b], acc[t][b], 0, 0, 0);
                }
            });
        }
    }

    // D: register r = a-channel acc_row(r) + 4h of the a-tile, lane = b-channel
    const int h = lane >> 5, m = lane & 31;
    static_for<9>([&](auto t_tag) {
        constexpr int t = decltype(t_tag)::value;
        if (t < ntaps) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int bt = wave + 4 * b;
                if (bt < nbt) {
                    static_for<16>([&](auto r_tag) {
                        constexpr int r = decltype(r_tag)::value;
                        const int a = 32 * at + acc_row(r) + 4 * h;
                        atomicAdd(p.gw + ((size_t)(tap0 + t) * p.Ca + a) * p.Cb + 32 * bt + m, acc[t][b][r]);
                    });
                }
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad8: wgrad_kernel with two waves per SIMD and a double-buffered tile pipeline (round 3, late).
//   wgrad_kernel stages a tile, waits, multiplies, and starts over (one wave per SIMD, source offsets recomputed per tile):
//   0.07 of the bf16 MFMA peak on the 128 -> 128 layers.  Here a block owns ONE tile position (ty, tx) and walks samples
//   n = first, first + step, ...: the source offsets of its staging pieces are computed once; sample k + 1 is on its way by
//   LDS-DMA while sample k multiplies; 8 waves = 2 position groups (tile rows 4 kg .. 4 kg + 3) x 4 b-tiles, every wave
//   with its own partial sums (9 taps x 16 registers), flushed by atomics at the end as before.  Cb <= 128.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 1) wgrad8_kernel(const WGArgs p, int tiles_per_image, int sample_step) {
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wb = wave & 3, kg = wave >> 2;
    const int at = blockIdx.y;
    const int tap0 = blockIdx.z * 9;
    const int ntaps = min(9, p.kk - tap0);
    const int nbt = p.Cb / 32;

    const int x_pieces = p.HR * p.HC * 4;  // pixel-major: 4 x 16 B per pixel (32 channels of the a-tile)
    const int x_instr = (x_pieces + 63) / 64;
    const int ypp = p.Cb / 8;              // 16-byte pieces per position
    const int y_pieces = 128 * ypp;
    const int y_instr = (y_pieces + 63) / 64;
    const size_t buf_bytes = (size_t)(x_instr + y_instr) * 1024;

    const int tp = blockIdx.x % tiles_per_image, n_first = blockIdx.x / tiles_per_image;
    const int tx = tp % p.tiles_x, ty = tp / p.tiles_x;
    const int i0 = ty * 8, j0 = tx * 16;

    // source offsets of this thread's staging pieces inside a sample (~0 = zeros), once per block
    constexpr int MAXX = 8, MAXY = 4;  // <= 64 KiB of X halo, 32 KiB of Y (Cb <= 128)
    unsigned xoff[MAXX], yoff[MAXY];
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < x_pieces ? pc : x_pieces - 1;
        const int pix = pc >> 2, quarter = pc & 3;
        const int r = (int)__umulhi((unsigned)pix, p.m_hc), c = pix - r * p.HC;
        int iy = p.S * i0 + p.dymin + r, ix = p.S * j0 + p.dxmin + c;
        bool ok = true;
        if (p.reflect) {
            iy = reflect_idx(iy, p.H);
            ix = reflect_idx(ix, p.W);
        } else {
            ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        }
        xoff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.Ca + 32 * at) * 2 + quarter * 16) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < MAXY; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < y_pieces ? pc : y_pieces - 1;
        const int pos = (int)__umulhi((unsigned)pc, p.m_ypp), part = pc - pos * ypp;
        const int i_ = i0 + (pos >> 4), jj = j0 + (pos & 15);
        const bool ok = i_ < p.OH && jj < p.OW;  // positions outside contribute zero
        yoff[i] = ok ? (unsigned)(((i_ * p.OW + jj) * p.Cbs + p.cb0) * 2 + part * 16) : 0xFFFFFFFFu;
    }
    const size_t x_sample = (size_t)p.H * p.W * p.Ca * 2, y_sample = (size_t)p.OH * p.OW * p.Cbs * 2;
    auto issue = [&](int n, char *buf) {
        const char *x_n = (const char *)p.x + (size_t)n * x_sample;
#pragma unroll
        for (int i = 0; i < MAXX; ++i) {
            const int j = wave + i * NW;
            if (j < x_instr) glds16(xoff[i] != 0xFFFFFFFFu ? (const void *)(x_n + xoff[i]) : p.zero, buf + j * 1024);
        }
        const char *y_n = (const char *)p.y + (size_t)n * y_sample;
        char *yb = buf + (size_t)x_instr * 1024;
#pragma unroll
        for (int i = 0; i < MAXY; ++i) {
            const int j = wave + i * NW;
            if (j < y_instr) glds16(yoff[i] != 0xFFFFFFFFu ? (const void *)(y_n + yoff[i]) : p.zero, yb + j * 1024);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // transposed reads: 16-lane group g: channel half g&1, k-group g>>1; lane 4q+pp of the group addresses row q
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
    const int col_off = (16 * (g & 1) + 4 * pp) * 2;  // bytes inside a 32-channel record
    const int lj0 = 8 * (g >> 1) + q4, lj1 = lj0 + 4;  // this lane's columns of the two 4-row blocks of a 16-position k-step
    const int btc = wb < nbt ? wb : 0;

    if (n_first < p.N) issue(n_first, smem);
    int it = 0;
    for (int n = n_first; n < p.N; n += sample_step, ++it) {
        char *cur = smem + (size_t)(it & 1) * buf_bytes;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // sample n landed; the other buffer's readers are done
        if (n + sample_step < p.N) issue(n + sample_step, smem + (size_t)((it + 1) & 1) * buf_bytes);
        const char *xbuf = cur, *ybuf = cur + (size_t)x_instr * 1024;
#pragma unroll 2
        for (int kq = 0; kq < 4; ++kq) {  // 16 positions: tile row ks, columns 0..15
            const int ks = 4 * kg + kq;
            const char *yb = ybuf + (size_t)(16 * ks) * p.Cb * 2 + 64 * btc + col_off;
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(yb + (size_t)lj0 * p.Cb * 2));
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4 *)(yb + (size_t)lj1 * p.Cb * 2));
            const short bv[8] = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, bv);
            static_for<9>([&](auto t_tag) __attribute__((always_inline)) {
                constexpr int t = decltype(t_tag)::value;
                if (t < ntaps) {
                    const int hr = p.S * ks + p.dy[tap0 + t] - p.dymin;
                    const int hc = p.dx[tap0 + t] - p.dxmin;
                    const char *xb = xbuf + (size_t)(hr * p.HC + hc) * 64 + col_off;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(xb + p.S * lj0 * 64));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(xb + p.S * lj1 * 64));
                    const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), bfr, acc[t], 0, 0, 0);
                }
            });
        }
    }

    // The two position groups add their partial sums in LDS first (the staging buffers are dead now), so that a block flushes
    // each of its 9 x 32 x Cb sums with ONE atomic instead of two: group 1 parks five, then four taps (20 KiB per wave), group 0
    // adds them to its own.
    const int h = lane >> 5, m = lane & 31;
    {
        float *park = (float *)smem + (size_t)wb * (5 * 16 * 64) + lane;
        static_for<2>([&](auto half_tag) __attribute__((always_inline)) {
            constexpr int half = decltype(half_tag)::value, t0 = half * 5, t1 = half ? 9 : 5;
            __syncthreads();  // every wave is done with the staging buffers / group 0 has read the previous half
            if (kg == 1) {
                static_for<t1 - t0>([&](auto i_tag) __attribute__((always_inline)) {
                    constexpr int t = t0 + decltype(i_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) park[((t - t0) * 16 + r) * 64] = acc[t][r];
                });
            }
            __syncthreads();
            if (kg == 0) {
                static_for<t1 - t0>([&](auto i_tag) __attribute__((always_inline)) {
                    constexpr int t = t0 + decltype(i_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += park[((t - t0) * 16 + r) * 64];
                });
            }
        });
    }
    // D: register r = a-channel acc_row(r) + 4h of the a-tile, lane = b-channel
    if (kg == 0 && wb < nbt) {
        static_for<9>([&](auto t_tag) __attribute__((always_inline)) {
            constexpr int t = decltype(t_tag)::value;
            if (t < ntaps) {
                static_for<16>([&](auto r_tag) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_tag)::value;
                    const int a = 32 * at + acc_row(r) + 4 * h;
                    atomicAdd(p.gw + ((size_t)(tap0 + t) * p.Ca + a) * p.Cbs + p.cb0 + 32 * wb + m, acc[t][r]);
                });
            }
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Reflect fold of a data gradient computed on the extended domain [-P, H + P) x [-P, W + P): the gradient of
// `reflect pad, then valid convolution` (nn.Conv2d(padding_mode='reflect'), _autoencoders.py:78-85) with respect to
// the unpadded input adds the border of the padded gradient back onto its mirror pixels.  P = 0: plain read.
// ---------------------------------------------------------------------------------------------------------------
struct FoldSrc {
    const float *g;  // fp32 [N][H + 2P][W + 2P][C]
    int H, W, P;
};

__device__ __forceinline__ float fold_read(const FoldSrc &f, int n, int y, int x, int C, int c) {
    const int HP = f.H + 2 * f.P, WP = f.W + 2 * f.P;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y;
    if (y >= 1 && y <= f.P) ys[ny++] = -y;
    if (y <= f.H - 2 && y >= f.H - 1 - f.P) ys[ny++] = 2 * (f.H - 1) - y;
    xs[nx++] = x;
    if (x >= 1 && x <= f.P) xs[nx++] = -x;
    if (x <= f.W - 2 && x >= f.W - 1 - f.P) xs[nx++] = 2 * (f.W - 1) - x;
    float s = 0.0f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b)
            s += f.g[(((size_t)n * HP + ys[a] + f.P) * WP + xs[b] + f.P) * C + c];
    return s;
}

// Activation backward: out = g * (y > 0 ? 1 : slope), y = the activation's OUTPUT (LeakyReLU keeps the sign, ReLU's
// output is positive exactly where its input was).  g: bf16 [N][H][W][C], or fp32 on the extended domain f (already
// folded in place: fold_inplace_kernel), of which the interior is read.
static __global__ void act_bwd_kernel(const __bf16 *g16, FoldSrc f, const __bf16 *y, float slope, __bf16 *out, int N, int H,
                                      int W, int C) {
    const size_t total = (size_t)N * H * W * C;
    const int HP = H + 2 * f.P, WP = W + 2 * f.P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float g;
        if (g16) {
            g = (float)g16[i];
        } else {
            const int c = (int)(i % C);
            size_t r = i / C;
            const int x = (int)(r % W);
            r /= W;
            const int yy = (int)(r % H);
            const int n = (int)(r / H);
            g = f.g[(((size_t)n * HP + yy + f.P) * WP + x + f.P) * C + c];
        }
        out[i] = (__bf16)((float)y[i] > 0.0f ? g : slope * g);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm2d in TRAINING mode (batch statistics; the units' optional nn.BatchNorm2d, _autoencoders.py:72-73, :87-88) on
// NCHW fp32 tensors.  Both directions are one pair of per-channel moments and one per-channel affine map:
//   forward   S1 = sum x, S2 = sum x x            y  = x A + C             A = w rstd, C = b - mean A
//   backward  S1 = sum dy, S2 = sum dy x          dx = dy A + x B + C      (coefficients from S1, S2, mean, rstd: train.py)
// bn_moments_kernel: grid (C, splits); a block walks its share of the N x HW elements of channel c, sums in double and
// adds its two partial sums with one double atomic each.
// ---------------------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) bn_moments_kernel(const float *a, const float *b, int N, int C, long HW,
                                                                 double *s1, double *s2) {
    const int c = blockIdx.x;
    const long per_n = HW, total = (long)N * HW;
    double t1 = 0.0, t2 = 0.0;
    for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.y * blockDim.x) {
        const long n = i / per_n, r = i - n * per_n;
        const size_t off = ((size_t)n * C + c) * HW + r;
        const float va = a[off], vb = b[off];
        t1 += (double)va;
        t2 += (double)va * (double)vb;
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = t1;
    red[1][threadIdx.x] = t2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(s1 + c, red[0][0]);
        atomicAdd(s2 + c, red[1][0]);
    }
}

// out = a A[c] + (b ? b B[c] : 0) + Cc[c]
static __global__ void bn_affine_kernel(const float *a, const float *b, int C, long HW, size_t total, const float *A,
                                        const float *B, const float *Cc, float *out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / (size_t)HW) % (size_t)C);
        float v = a[i] * A[c] + Cc[c];
        if (b) v += b[i] * B[c];
        out[i] = v;
    }
}

// folded gradient -> bf16 T layout (layers without GDN between two convolutions)
static __global__ void fold_to_bf16_kernel(FoldSrc f, __bf16 *out, int N, int C) {
    const size_t total = (size_t)N * f.H * f.W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % f.W);
        r /= f.W;
        const int y = (int)(r % f.H);
        const int n = (int)(r / f.H);
        out[i] = (__bf16)fold_read(f, n, y, x, C, c);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GDN / IGDN, exact fp32.   n[pix][c] = beta[c] + sum_j gamma[c][j] z[pix][j]^2 ;  y = z * n^(-1/2)  (IGDN: n^(1/2))
//
// gdn_gemm_a:  D[pix][c] = (beta[c]) + sum_j f(A[pix][j]) M[c][j]   with M (C x C, row-major) resident in LDS
//   block = 4 waves x 64 pixels; A fragments straight from HBM (16 B per lane = 4 MFMA k-steps: lane half h carries
//   j = 8q + 4h .. + 3); D has the channel on the lane and pixels in the registers -> every epilogue access is a
//   128-byte row.
//   MODE 0  forward:     f = square; y = z * rsqrt(D) (sqrt) -> y16 (and y32)
//   MODE 1  backward 1:  f = square; from g_y (fold_read): g_n = -(1/2) g_y z D^(-3/2)  (IGDN: +(1/2) g_y z D^(-1/2))
//                        -> gn32;  direct term g_y D^(-1/2) (IGDN: g_y D^(1/2)) -> gzd32
//   MODE 2  backward 2:  f = identity on A = g_n, M = gamma^T: t[pix][j] = sum_c g_n[c] gamma[c][j];
//                        g_z = gzd + 2 z t -> gz16 (and gz32)
// ---------------------------------------------------------------------------------------------------------------
struct GdnArgs {
    const float *a;      // A operand, fp32 [pixels][C]
    const float *mat;    // M, fp32 [C][C]
    const float *beta;   // [C] or null (MODE 2)
    const float *z;      // fp32 [pixels][C]
    FoldSrc gy;          // MODE 1
    int img_h, img_w;    // MODE 1: pixel index -> (n, y, x)
    float *o32a;         // MODE 0: y32 | MODE 1: gn32 | MODE 2: gz32   (may be null)
    float *o32b;         // MODE 1: gzd32 | MODE 2 (input): gzd32
    void *o16;           // MODE 0: y16 | MODE 2: gz16  (may be null)
    long pixels;
    int C, inverse;
};

// A operand staging (CT <= 4: LDS has room beside M): per 32-channel chunk the block's 256 x 32 fp32 tile is copied by
// LDS-DMA as whole 128-byte row segments (8 pixels per instruction) into a dense, XOR-swizzled image -- slot of piece s
// of pixel p = 8 p + (s ^ ((p >> 1) & 7)) -- which the MFMA lanes (one pixel each, stride 128 B) then read without bank
// conflicts; double-buffered under the 128 MFMAs a wave issues per chunk.  Reading A straight from HBM (16 bytes per
// lane and pixel row, 32 lines per instruction, the tile thrashing the 32-KiB L1) made these kernels 12x slower than
// their MFMA time (profiles/r02_experiments.md 8); that form remains for CT > 4, where M alone fills the LDS.
template <int CT, int MODE>
__global__ void __launch_bounds__(256, 1) gdn_gemm_a_kernel(const GdnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = CT * 32, LD = C + 4;
    constexpr bool STAGED = CT <= 4;
    constexpr int M_BYTES = C * LD * 4;
    float *mlds = (float *)smem;
    char *abuf = smem + M_BYTES;  // STAGED: 2 x 32 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    for (int i = threadIdx.x; i < C * C / 4; i += 256) {
        const int r = i / (C / 4), c4 = i - r * (C / 4);
        *(f32x4 *)(mlds + r * LD + 4 * c4) = *(const f32x4 *)(p.mat + (size_t)r * C + 4 * c4);
    }
    __syncthreads();

    const long tiles = (p.pixels + 255) / 256;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long p0 = tile * 256 + wave * 64;
        f32x16 acc[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float b = (MODE != 2 && p.beta) ? p.beta[32 * ct + m] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][ct][r] = b;
                acc[1][ct][r] = b;
            }
        }
        if constexpr (STAGED) {
            // this thread's 8 pieces of a chunk: instruction j = wave + 4 i covers pixels 8j .. 8j+7 of the tile
            const char *src[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = wave + 4 * i;
                const int px = 8 * j + (lane >> 3);
                long gp = tile * 256 + px;
                gp = gp < p.pixels ? gp : p.pixels - 1;  // clamped rows are computed and never stored
                const int s = (lane & 7) ^ ((px >> 1) & 7);
                src[i] = (const char *)(p.a + gp * C) + s * 16;
            }
            auto issue = [&](int chunk, char *buf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) glds16(src[i] + chunk * 128, buf + (wave + 4 * i) * 1024);
            };
            __syncthreads();  // the previous tile's reads of both buffers are done
            issue(0, abuf);
            const int pl0 = 64 * wave + m, pl1 = pl0 + 32;
            for (int chunk = 0; chunk < CT; ++chunk) {
                wait_vm0();
                __syncthreads();
                char *cur = abuf + (chunk & 1) * 32768;
                if (chunk + 1 < CT) issue(chunk + 1, abuf + ((chunk + 1) & 1) * 32768);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v0 = *(const f32x4 *)(cur + (pl0 * 8 + ((2 * q + h) ^ ((pl0 >> 1) & 7))) * 16);
                    f32x4 v1 = *(const f32x4 *)(cur + (pl1 * 8 + ((2 * q + h) ^ ((pl1 >> 1) & 7))) * 16);
                    if (MODE != 2) {
                        v0 *= v0;
                        v1 *= v1;
                    }
                    f32x4 mf[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        mf[ct] = *(const f32x4 *)(mlds + (32 * ct + m) * LD + 32 * chunk + 8 * q + 4 * h);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], mf[ct][s], acc[0][ct], 0, 0, 0);
                            acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], mf[ct][s], acc[1][ct], 0, 0, 0);
                        }
                }
            }
        } else {
            long pa0 = p0 + m, pa1 = p0 + 32 + m;
            pa0 = pa0 < p.pixels ? pa0 : p.pixels - 1;  // clamped rows are computed and never stored
            pa1 = pa1 < p.pixels ? pa1 : p.pixels - 1;
            const float *a0 = p.a + pa0 * C + 4 * h, *a1 = p.a + pa1 * C + 4 * h;
#pragma unroll 2
            for (int q = 0; q < C / 8; ++q) {
                f32x4 v0 = *(const f32x4 *)(a0 + 8 * q), v1 = *(const f32x4 *)(a1 + 8 * q);
                if (MODE != 2) {
                    v0 *= v0;
                    v1 *= v1;
                }
                f32x4 mf[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) mf[ct] = *(const f32x4 *)(mlds + (32 * ct + m) * LD + 8 * q + 4 * h);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[s], mf[ct][s], acc[0][ct], 0, 0, 0);
                        acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s], mf[ct][s], acc[1][ct], 0, 0, 0);
                    }
            }
        }
        static_for<2>([&](auto pt_tag) {
            constexpr int pt = decltype(pt_tag)::value;
            static_for<16>([&](auto r_tag) {
                constexpr int r = decltype(r_tag)::value;
                const long pix = p0 + 32 * pt + acc_row(r) + 4 * h;
                if (pix < p.pixels) {
                    // (n, y, x) of the pixel once per accumulator row, in 32-bit arithmetic: the 64-bit divisions of the
                    // first version, repeated per channel tile, were most of the backward kernel's time
                    int nimg = 0, py = 0, px = 0;
                    bool interior = true;
                    if (MODE == 1) {
                        const unsigned hw = (unsigned)(p.img_h * p.img_w), upix = (unsigned)pix;
                        nimg = (int)(upix / hw);
                        const unsigned rem = upix - (unsigned)nimg * hw;
                        py = (int)(rem / (unsigned)p.img_w);
                        px = (int)(rem - (unsigned)py * (unsigned)p.img_w);
                        const int P = p.gy.P;
                        interior = P == 0 || (py > P && py < p.img_h - 1 - P && px > P && px < p.img_w - 1 - P);
                    }
                    const size_t gbase = MODE == 1 ? (((size_t)nimg * (p.img_h + 2 * p.gy.P) + py + p.gy.P) *
                                                          (p.img_w + 2 * p.gy.P) + px + p.gy.P) * C
                                                   : 0;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int c = 32 * ct + m;
                        const size_t off = (size_t)pix * C + c;
                        const float d = acc[pt][ct][r];
                        if (MODE == 0) {
                            const float y = p.z[off] * (p.inverse ? __builtin_amdgcn_sqrtf(d) : __builtin_amdgcn_rsqf(d));
                            if (p.o32a) p.o32a[off] = y;
                            if (p.o16) ((__bf16 *)p.o16)[off] = (__bf16)y;
                        } else if (MODE == 1) {
                            const float gy = interior ? p.gy.g[gbase + c] : fold_read(p.gy, nimg, py, px, C, c);
                            const float zz = p.z[off];
                            float gn, gd;
                            if (p.inverse) {
                                const float sq = __builtin_amdgcn_sqrtf(d);
                                gd = gy * sq;
                                gn = 0.5f * gy * zz * __builtin_amdgcn_rsqf(d);
                            } else {
                                const float rs = __builtin_amdgcn_rsqf(d);
                                gd = gy * rs;
                                gn = -0.5f * gy * zz * rs * rs * rs;
                            }
                            p.o32a[off] = gn;
                            p.o32b[off] = gd;
                        } else {
                            const float gz = p.o32b[off] + 2.0f * p.z[off] * d;
                            if (p.o32a) p.o32a[off] = gz;
                            if (p.o16) ((__bf16 *)p.o16)[off] = (__bf16)gz;
                        }
                    }
                }
            });
        });
    }
}

// gdn_gemm_b: parameter gradients, a contraction over pixels.
//   g_gamma[c][j] += sum_pix g_n[pix][c] z[pix][j]^2 ;  g_beta[c] += sum_pix g_n[pix][c]
//   block = CT waves (wave w: channel tile c = 32w ..), each wave all j-tiles; 2 pixels per MFMA k-step, operands are
//   coalesced 128-byte row reads straight from HBM; one atomic flush per block.
template <int CT>
__global__ void __launch_bounds__(CT * 64, 1) gdn_gemm_b_kernel(const float *gn, const float *z, long pixels, float *ggamma,
                                                                float *gbeta) {
    constexpr int C = CT * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    f32x16 acc[CT];
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
    float bsum = 0.0f;
    const long per = (pixels + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < pixels ? lo + per : pixels;
    for (long p0 = lo; p0 < hi; p0 += 8) {
        float a[4], b[4][CT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long pix = p0 + 2 * u + h;
            const bool ok = pix < hi;
            const size_t off = (size_t)(ok ? pix : lo) * C;
            a[u] = ok ? gn[off + 32 * wave + m] : 0.0f;
#pragma unroll
            for (int jt = 0; jt < CT; ++jt) {
                const float v = z[off + 32 * jt + m];
                b[u][jt] = v * v;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum += a[u];
#pragma unroll
            for (int jt = 0; jt < CT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][jt], acc[jt], 0, 0, 0);
        }
    }
    // D: register r = channel c = 32 wave + acc_row(r) + 4h, lane = j
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
        static_for<16>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            atomicAdd(ggamma + (size_t)(32 * wave + acc_row(r) + 4 * h) * C + 32 * jt + m, acc[jt][r]);
        });
    atomicAdd(gbeta + 32 * wave + m, bsum);  // both lane halves add their pixels' share
}

// ---------------------------------------------------------------------------------------------------------------
// Edge layers (3 image channels): the first analysis layer and the last synthesis layer as POINTWISE GEMMs over
// K = (tap, channel) <= 32 instead of 32-channel padding of a 3-channel tensor (ten times the useful work, and a
// 32-channel copy of the image): im2col of the module-boundary NCHW tensor straight into [N][OH][OW][32] bf16
// (j = tap * C + c; reflect padding: the first conv's input; zeros: the last layer's output gradient), a 1 x 1
// gather-GEMM / weight gradient on it, and for the last layer's forward the transposed form: per INPUT position the 27
// products u[pos][(tap, co)], then col2im sums the <= 4 terms of every output pixel into NCHW fp32.
// ---------------------------------------------------------------------------------------------------------------
// One thread per (position, 16-byte quarter of its record): eight values gathered, one 16-byte store -- consecutive threads
// write consecutive pieces.  (The element-per-thread form spent its time on 64-bit index arithmetic and 2-byte stores:
// 0.20 ms per 128 x 3 x 256^2 image batch, now a quarter of that.)
static __global__ void im2col_s2_kernel(const float *x, __bf16 *out, int N, int C, int H, int W, int OH, int OW, int ks,
                                        int reflect) {
    const size_t total = (size_t)N * OH * OW * 4;
    const int P = ks / 2, K = ks * ks * C;
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        size_t r = i >> 2;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        const float *xn = x + (size_t)n * C * plane;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * q + e;
            float f = 0.0f;
            if (j < K) {
                const int tap = j / C, c = j - tap * C;
                const int ky = tap / ks, kx = tap - ky * ks;
                int iy = 2 * oy + ky - P, ix = 2 * ox + kx - P;
                bool ok = true;
                if (reflect) {
                    iy = reflect_idx(iy, H);
                    ix = reflect_idx(ix, W);
                } else {
                    ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                }
                if (ok) f = xn[(size_t)c * plane + (size_t)iy * W + ix];
            }
            v[e] = (__bf16)f;
        }
        *(bf16x8 *)(out + i * 8) = v;
    }
}

// out[n][c][Y][X] = bias[c] + sum over taps (ky, kx) with (Y + P - ky), (X + P - kx) even and inside of
// u[n][(Y + P - ky) / 2][(X + P - kx) / 2][(ky * ks + kx) * C + c]      (ConvTranspose2d(k, stride 2, padding k//2, output_padding 1))
// One thread per output PIXEL, all its channels (C <= 3 for k = 3): the <= 4 (k = 3) or <= 9 (k = 5) terms of a pixel share their
// index arithmetic and read C adjacent floats of each record.
static __global__ void col2im_s2_kernel(const float *u, const float *bias, float *out, int N, int C, int H, int W, int ks) {
    const int OH = 2 * H, OW = 2 * W, P = ks / 2;
    const size_t total = (size_t)N * OH * OW, oplane = (size_t)OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW);
        size_t r = i / OW;
        const int Y = (int)(r % OH);
        const int n = (int)(r / OH);
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // (C <= 3 when ks * ks * C <= 32 and ks >= 3)
        for (int ky = (Y + P) & 1; ky < ks; ky += 2) {
            const int iy = (Y + P - ky) >> 1;
            if (iy < 0 || iy >= H || Y + P - ky < 0) continue;
            for (int kx = (X + P) & 1; kx < ks; kx += 2) {
                const int ix = (X + P - kx) >> 1;
                if (ix < 0 || ix >= W || X + P - kx < 0) continue;
                const float *rec = u + (((size_t)n * H + iy) * W + ix) * 32 + (ky * ks + kx) * C;
                for (int c = 0; c < C; ++c) s[c] += rec[c];
            }
        }
        for (int c = 0; c < C; ++c) out[((size_t)n * C + c) * oplane + (size_t)Y * OW + X] = s[c] + (bias ? bias[c] : 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// layout conversions between the module boundary (NCHW fp32) and the T layout
// ---------------------------------------------------------------------------------------------------------------
static __global__ void nchw_to_t_kernel(const float *in, __bf16 *o16, float *o32, int N, int C, int H, int W, int Cp) {
    const size_t total = (size_t)N * H * W * Cp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const size_t pix = i / Cp;
        const size_t hw = (size_t)H * W;
        const size_t n = pix / hw, r = pix - n * hw;
        const float v = c < C ? in[(n * C + c) * hw + r] : 0.0f;
        if (o16) o16[i] = (__bf16)v;
        if (o32) o32[i] = v;
    }
}

static __global__ void t_to_nchw_kernel(const float *in, float *out, int N, int C, int H, int W, int Cp) {
    const size_t hw = (size_t)H * W, total = (size_t)N * C * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i % hw;
        const size_t nc = i / hw;
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        out[i] = in[(n * hw + r) * Cp + c];
    }
}

// column sums of a bf16 T tensor (bias gradient): out[c] += sum_pix g[pix][c]
static __global__ void colsum_bf16_kernel(const __bf16 *g, long pixels, int C, float *out) {
    const int c = threadIdx.x % C;
    const int rows_per_block = blockDim.x / C;
    float s = 0.0f;
    for (long pix = (long)blockIdx.x * rows_per_block + threadIdx.x / C; pix < pixels; pix += (long)gridDim.x * rows_per_block)
        s += (float)g[(size_t)pix * C + c];
    if (threadIdx.x < rows_per_block * C) atomicAdd(out + c, s);
}

}  // namespace tr
}  // namespace cae
