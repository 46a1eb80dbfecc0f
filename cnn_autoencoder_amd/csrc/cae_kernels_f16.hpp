// Split-precision ("f16x3") variants of the conv / deconv / GDN kernels for CDNA4.
//
// Every fp32 value v is carried as two halves  v = hi + lo,  hi = f16(v), lo = f16(v - hi)
// (22 significant bits).  A product is three f16 MFMAs accumulated in fp32:
//     a*b ~= ah*bh + ah*bl + al*bh            (the dropped al*bl term is < 2^-22 |a b|)
// On v_mfma_f32_32x32x16_f16 this costs 3 x 32 cycles per 32x32x16 block against 8 x 64 cycles on
// v_mfma_f32_32x32x2_f32: 5.3x less matrix-pipe time at fp32-class accuracy (measured through the
// whole analysis stack: max |err| 1.5e-6 vs 1.1e-6 for plain fp32, both against fp64).
//
// HBM layout "C8S": activations are [N][P][H][row], one 16-byte hi piece and one 16-byte lo piece
// (8 x f16 each) per pixel for the 8 channels of plane P; a row is a sequence of 1-KiB groups
// [32 hi pieces][32 lo pieces] (see "row layouts" below; C8SP is the synthesis track's variant).
// Same bytes as fp32 C8, produced once in the epilogue of the previous layer, so consumers never
// convert.  A 16-channel MFMA k-step = two planes; lane half h of the wave supplies the 8 channels
// of plane 2q+h.
//
// Block = 4 waves, ONE block per CU (each wave may use the whole 512-entry register file):
//   wave tile = all CT*32 output channels x 64 pixels (two 32-pixel MFMA column tiles), so every
//   A (weight) fragment read from LDS feeds 6 MFMAs and LDS traffic stays ~60 B/clk/CU.
#pragma once
#include "cae_kernels.hpp"

namespace cae {

#ifndef CAE_FIRST_PIPE
#define CAE_FIRST_PIPE 1  // conv_first_f16: hand-placed GDN schedule (gdn_resident_pipe_f16); 0: the compiler's
#endif
#ifndef CAE_F16_PIPE
#define CAE_F16_PIPE 0  // 1: software-pipelined operand reads in deconv_s2_f16's K loop (measured null: the stack is power-limited)
#endif
#ifndef CAE_F16_ISSUERS
#define CAE_F16_ISSUERS 2  // LDS-DMA issuer waves of conv_s2_f16 / deconv_s2_f16: NI = NW / CAE_F16_ISSUERS
#endif
// issuer index of wave w (0 .. NI-1), or -1: the first NI waves, or (CAE_F16_ISSUE_HI) the last NI
#ifdef CAE_F16_ISSUE_HI
#define ISSUER(w, NW, NI) ((w) >= (NW) - (NI) ? (w) - ((NW) - (NI)) : -1)
#else
#define ISSUER(w, NW, NI) ((w) < (NI) ? (w) : -1)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// The same split for two values in four instructions: v_cvt_pk_f16_f32 (hi pair), two v_fma_mix_f32 (hi x -1 + v with the
// f16 source converted on the fly: the exact residual, as v - (float)hi) and v_cvt_pk_f16_f32 (lo pair).  The portable form
// above compiles to 6-8 (cvt, cvt back, sub per value, packing); the f16x3 kernels split 150-200 values per 32-pixel tile
// and their vector issue is what the matrix pipes wait for (conv_first_f16) or what costs energy (the others).
// Bit-identical results.  -> {packed hi pair, packed lo pair} (low half = a)
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2v split2_f16(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 ab = {a, b};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(b));
    const f32x2 l = {l0, l1};
    return u32x2v{hi, __builtin_bit_cast(unsigned, __builtin_convertvector(l, f16x2))};
}

// max(mx, |a|, |b|) as ONE v_max3_f32 with source modifiers (the portable fmaxf / fabsf form canonicalises each operand
// first: 3.5 instructions per pair measured in conv_first_f16; every tile takes two such maxima over 16 CT values)
__device__ __forceinline__ float absmax3(float mx, float a, float b) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(mx));
    return r;
}

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// ---- range guard ---------------------------------------------------------------------------------------
// f16 carries 5 exponent bits: a value above F16_MAX cannot be stored in the split format (hi = inf).  Every
// kernel that writes split rows raises *flag when that happens; the host side then repeats the call on the exact-fp32
// kernels (cae_range_check, include/cae_hip.h).  Values are finite by induction (finite weights checked on upload,
// finite inputs checked here), so a maximum is enough to see an overflow.
constexpr float F16_MAX = 65504.0f;

__device__ __forceinline__ void raise_if_over(float mx, int *flag) {
    if (!(mx <= F16_MAX)) *flag = 1;  // (negated compare: a NaN raises too)
}

// Per-pixel power-of-two scale for the GDN squares: the squares of a pixel's channels are formed from y * sc with
// sc = 2^-e chosen so that the pixel's largest |y * sc| lies in [64, 128): (y sc)^2 <= 2^14 fits f16 whatever the
// magnitude of y (unscaled squares overflowed at |y| > 255.9), and squares of small activations no longer fall
// into the f16 subnormals.  beta enters the norm accumulator as beta * sc^2 and the result is multiplied by
// 1 / sc^2: every factor is a power of two, so apart from the range the arithmetic is unchanged.  A pixel is one MFMA
// column (lanes l and l + 32), the contraction runs over rows: a per-column scale commutes with it.
// -> sc (this lane's pixel); *isc = 1 / sc.  The exponent is clamped to +-60 so sc^2 and 1 / sc^2 stay normal fp32.
__device__ __forceinline__ float pixel_scale(float mx, float *isc) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = __builtin_fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // both lanes of the pixel
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127 - 6;          // mx * 2^-e in [64, 128)
    e = mx > 0.0f ? (e < -60 ? -60 : (e > 60 ? 60 : e)) : 0;
    *isc = __uint_as_float((unsigned)(127 + e) << 23);
    return __uint_as_float((unsigned)(127 - e) << 23);
}

__device__ __forceinline__ f32x16 mfma3(const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl,
                                        f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    return acc;
}

// ---- fused GDN / IGDN on PT x CT accumulator tiles, f16x3 ---------------------------------------
// packed gamma: [jt][co][s(2)][hl(2)][64 lanes][8 f16]:
//   G(c = 32co + (lane&31), j = 32jt + row(8s + e) + 4(lane>>5)),  row(r) = (r&3) + 8(r>>2)
// (the k order inside a 16-deep step is the accumulator's own row order, so y*y needs no shuffle).
template <int CT, int PTA, int NW, bool INVERSE, int STAGE_BYTES, int PT0 = 0, int PT = PTA, class Tail>
__device__ __forceinline__ void gdn_stages_f16(f32x16 (&yy)[PTA][CT], const LayerArgs &p, char *smem, int &sc,
                                               int wave, int lane, Tail tail) {
    // normalises the row tiles PT0 .. PT0+PT-1 of yy (the others are left untouched)
    constexpr int G_BYTES = CT * 4096;  // per jt: CT co x 2 s x 2 hl x 1 KiB
    const int h = lane >> 5;
    f32x16(&y)[PTA][CT] = yy;
    f32x16 nrm[PT][CT];
    float psc[PT], pisc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        float mx = 0.0f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                mx = absmax3(mx, y[PT0 + pt][ct][r], y[PT0 + pt][ct][r + 1]);
        psc[pt] = pixel_scale(mx, &pisc[pt]);
        init_acc<CT>(nrm[pt], p.beta, h, 1.0f);
        const float sc2 = psc[pt] * psc[pt];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[pt][ct][r] *= sc2;
    }
#pragma unroll
    for (int jt = 0; jt < CT; ++jt) {
        wait_vm0();
        __syncthreads();
        char *cur = smem + (sc & 1) * STAGE_BYTES;
        char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
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
        const char *gb = cur + lane * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 sh[PT], sl[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                u32x4v hw, lw;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = y[PT0 + pt][jt][8 * s + 2 * e] * psc[pt], v1 = y[PT0 + pt][jt][8 * s + 2 * e + 1] * psc[pt];
                    { const u32x2v t_ = split2_f16(v0 * v0, v1 * v1); hw[e] = t_[0]; lw[e] = t_[1]; }
                    if constexpr (!INVERSE) {  // keep y sc: y rsqrt(n) = (y sc) rsqrt(n sc^2), one multiply less at the end
                        y[PT0 + pt][jt][8 * s + 2 * e] = v0;
                        y[PT0 + pt][jt][8 * s + 2 * e + 1] = v1;
                    }
                }
                sh[pt] = __builtin_bit_cast(f16x8, hw);
                sl[pt] = __builtin_bit_cast(f16x8, lw);
            }
#pragma unroll
            for (int co = 0; co < CT; ++co) {
                const f16x8 gh = *(const f16x8 *)(gb + ((co * 2 + s) * 2 + 0) * 1024);
                const f16x8 gl = *(const f16x8 *)(gb + ((co * 2 + s) * 2 + 1) * 1024);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) nrm[pt][co] = mfma3(gh, gl, sh[pt], sl[pt], nrm[pt][co]);
            }
        }
        ++sc;
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        // nrm = n sc^2:  sqrt(n) = sqrt(nrm) / sc;  y rsqrt(n) = (y sc) rsqrt(nrm), and y already holds y sc
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float nv = nrm[pt][ct][r];
                if constexpr (INVERSE)
                    y[PT0 + pt][ct][r] *= __builtin_amdgcn_sqrtf(nv) * pisc[pt];
                else
                    y[PT0 + pt][ct][r] *= __builtin_amdgcn_rsqf(nv);
            }
    }
}

// ---- row layouts of the split format ------------------------------------------------------------------
// A row of one 8-channel plane is a sequence of 1-KiB groups [32 hi pieces (16 B each)][32 lo pieces]:
//   C8S  (analysis track):  group g holds pixels 32g .. 32g+31;
//   C8SP (synthesis track): groups 2b and 2b+1 hold the even and the odd pixels of the 64-pixel block b.
// Why: (1) an LDS-DMA instruction fetches the hi (or lo) pieces of consecutive pixels, which are now
// contiguous 16-byte pieces (whole cache lines) instead of every second 16 bytes of interleaved records;
// (2) a sub-pixel phase of the transposed convolution produces every second pixel of an output row; into
// interleaved rows that was a 32-byte record every 64 bytes, i.e. half-written cache lines, measured at 1.3 of
// the 3.7 ms of the largest layer (profiles/r01_experiments.md).  With parity-split rows a phase writes one
// contiguous 1-KiB group per wave instruction.  The row pitch is a whole number of groups (blocks).
template <bool SP>
__host__ __device__ __forceinline__ size_t c8s_row_bytes(int w) {
    return SP ? (size_t)((w + 63) >> 6) * 2048 : (size_t)((w + 31) >> 5) * 1024;
}
// byte offset of pixel x's hi piece inside its row; the lo piece is 512 bytes further
template <bool SP>
__host__ __device__ __forceinline__ unsigned c8s_piece(int x) {
    return SP ? (unsigned)(x >> 6) * 2048u + (unsigned)(x & 1) * 1024u + (unsigned)((x & 63) >> 1) * 16u
              : (unsigned)(x >> 5) * 1024u + (unsigned)(x & 31) * 16u;
}

// store CT accumulator tiles of one pixel column-tile: C8S (split halves; SP: C8SP rows) | NCHW fp32 | HWC uint8
template <int CT, bool SP = false>
__device__ __forceinline__ void store_split_f16(const f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox, int h,
                                                int ct0 = 0) {  // acc[ct] = channel tile ct0 + ct
    {
        // lane (px, h) holds channels 4h..4h+3 of every plane as hi (H) and lo (L) halves.  One
        // v_permlane32_swap per dword leaves the lower lane with [H(ch 0-3) | H(ch 4-7)] and the upper
        // lane with [L(ch 0-3) | L(ch 4-7)]: one 16-byte store per lane and plane instead of two 8-byte.
        char *out = (char *)p.out;
        float mx = 0.0f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                mx = absmax3(mx, acc[ct][r], acc[ct][r + 1]);
        raise_if_over(mx, p.flag);
        const size_t row_bytes = c8s_row_bytes<SP>(p.OW), plane_bytes = (size_t)p.OH * row_bytes;
        char *dst0 = out + (((size_t)n * p.out_planes + 4 * ct0) * p.OH + oy) * row_bytes + c8s_piece<SP>(ox) + 512 * h;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x2 yh, xl;
                { const u32x2v t_ = split2_f16(acc[ct][4 * g + 0], acc[ct][4 * g + 1]); yh[0] = t_[0]; xl[0] = t_[1]; }
                { const u32x2v t_ = split2_f16(acc[ct][4 * g + 2], acc[ct][4 * g + 3]); yh[1] = t_[0]; xl[1] = t_[1]; }
                const auto r0 = __builtin_amdgcn_permlane32_swap(yh[0], xl[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(yh[1], xl[1], false, false);
                const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                char *dst = dst0;
                dst0 += plane_bytes;  // next plane: one 64-bit add of a uniform (no per-plane 64-bit multiplies)
#ifdef CAE_EXP_F16_NOSTORE  // timing-only ablation: activations are computed but not written (wrong results)
                if (p.N < 0)
#endif
                *(u32x4 *)dst = v;
            }
    }
}

// Residual units (_autoencoders.py:104-174, :230-304): acc += the unit's input, read from its split rows.  Lane (px, h)
// owns channels 4h .. 4h+3 of every plane: their hi halves are 8 bytes of the pixel's hi piece, their lo halves the same
// 8 bytes of the lo piece 512 bytes on, and hi + lo in fp32 is the stored value.  (Two 8-byte loads per plane, not the
// mirror image of store_split_f16's 16-byte load + v_permlane32_swap: hipcc 7.2 folded the two results of the swap
// into one -- it emitted 2 hi -- when both were consumed arithmetically in the same lane.)  Then the activation in front
// of the strided layer (p.post_act).  The unit input may have fewer planes than the stage writes (first analysis unit).
template <int CT, bool SP>
__device__ __forceinline__ void add_residual_f16(f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox, int h) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const size_t row_bytes = c8s_row_bytes<SP>(p.OW), plane_bytes = (size_t)p.OH * row_bytes;
    const char *src = (const char *)p.res + ((size_t)n * p.res_planes * p.OH + oy) * row_bytes + c8s_piece<SP>(ox) + 8 * h;
    const float slope = p.post_act == 0 ? 1.0f : (p.post_act == 1 ? 0.01f : 0.0f);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * ct + g < p.res_planes) {  // (uniform)
                const f16x4 hi = *(const f16x4 *)src, lo = *(const f16x4 *)(src + 512);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[ct][4 * g + k] += (float)hi[k] + (float)lo[k];
            }
            src += plane_bytes;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = acc[ct][4 * g + k];
                acc[ct][4 * g + k] = v > 0.0f ? v : slope * v;
            }
        }
}

// EPI: bit 0 = activation (p.act), bit 1 = residual sum + p.post_act (stride-1 stages of residual units)
template <int CT, bool SP = false, int EPI = 0>
__device__ __forceinline__ void store_tiles_f16(f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox,
                                                int h, bool valid) {
    // (the half-wave exchange of the split store needs both lanes of a pixel: lanes l and l+32 share (oy, ox),
    //  hence the same `valid`; invalid pairs skip the whole store)
    if (!valid) return;
    if (p.outfmt == OUT_C8) {
        if constexpr (EPI & 1) {
            // LeakyReLU / ReLU units (_autoencoders.py:62-76, :187-202): the activation before the split store, in place and
            // branch-free (slope 1 = no activation; a branch on p.act around the tile made the 192-channel kernels spill).
            // Only the kernels without a fused GDN carry it: GDN units have no other activation.
            const float slope = p.act == 0 ? 1.0f : (p.act == 1 ? 0.01f : 0.0f);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = acc[ct][r] > 0.0f ? acc[ct][r] : slope * acc[ct][r];
        }
        if constexpr (EPI & 2) {
            if (p.res) add_residual_f16<CT, SP>(acc, p, n, oy, ox, h);
        }
        store_split_f16<CT, SP>(acc, p, n, oy, ox, h);
    } else {
        store_tiles<CT, false>(acc, p, n, oy, ox, h, valid);  // (last layers / colour layers: no activation)
    }
}

// =================================================================================================
// conv_s2_f16_kernel: strided reflect conv (+bias) (+GDN), f16x3, C8S in.
//   block tile = 16 x 16 output pixels, NW waves x PT = 8 / NW column tiles of 2 rows x 16 pixels
//   (wave w: rows 2 PT w .. 2 PT w + 2 PT - 1; column tile pt: rows 2 PT w + 2 pt, +1)
//   stage = (16-channel chunk q, kernel row ky):
//     weights [kx][ct][hl][64][8 f16]  (KS*CT*2 KiB)   +   halo [pl][hl][16 rows][WH][16 B]
// =================================================================================================
// waves per block: 8 waves x 1 column tile (two waves per SIMD cover each other's operand waits) measured 4-8 % (conv2),
// 2-4 % (conv3), 17 % (conv4) faster than 4 waves x 2 column tiles at the same staging traffic (r01_experiments.md)
#ifndef CAE_CONV_F16_NW
#define CAE_CONV_F16_NW 8
#endif
// S = 1 (the stride-1 convolution in front of the strided one in LeakyReLU / ReLU units, _autoencoders.py:62-76): same
// staging, WH = 16 + KS - 1 halo columns.  SP: rows in and out are C8SP (synthesis track); ZP: zero padding instead of
// reflection -- the synthesis units' ConvTranspose2d(stride 1, padding k//2) (:187-202) is the zero-padded correlation
// with the flipped kernel (flipped when packed).
// RES: stage of a residual unit -- (I)GDN or activation, + the unit's input, + the strided layer's pre-activation
// (add_residual_f16); the synthesis units (ZP) normalise with the inverse GDN.
template <int KS, int CT, bool GDN, int S = 2, bool SP = false, bool ZP = false, bool RES = false>
__global__ void __launch_bounds__(CAE_CONV_F16_NW * 64, 1) conv_s2_f16_kernel(const LayerArgs p) {
    constexpr int NW = CAE_CONV_F16_NW, PT = 8 / NW;  // 8 waves x 1 column tile (shipped) | 4 waves x 2
    constexpr int PAD = KS / 2;
    constexpr int TX = 16, TY = 16;
    constexpr int WH = S * TX + KS - S;
    // (Round 2 tried an "aligned" halo image -- the 32 columns that coincide with one 32-pixel group of the C8S rows as
    //  whole 512-byte runs, the KS - 2 stray columns apart: no gain (profiles/r02_experiments.md 4), and its 512-byte row
    //  pitch made the stride-2 operand reads 2-way bank conflicted (1.4 conflict cycles per LDS-active cycle); the
    //  33-piece pitch of this row-major image is conflict-free.  deconv_s2_f16, whose reads are contiguous, keeps it.)
    constexpr int PLANE_PIECES = TY * WH;          // 16-byte pieces per (plane, half)
    constexpr int HALO_PIECES = 4 * PLANE_PIECES;  // [pl][hl][row][x]
    constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;
    constexpr int W_INSTR = KS * CT * 2;
    constexpr int W_BYTES = W_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int CONV_STAGE = W_BYTES + HALO_INSTR * 1024;
    constexpr int STAGE_BYTES = CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES;
    // LDS-DMA issuers: NI waves stage the next stage, the others go straight to their MFMAs.  Waves w and w + NW/2 share
    // a SIMD: with NI = NW/2 one wave per SIMD spends the first part of a stage issuing the copies (tens of cycles per
    // instruction) while its partner has the matrix pipe to itself, then computes while the partner waits at the
    // barrier -- the two waves of a SIMD run half a stage apart instead of in lockstep (profiles/r02_experiments.md 9).
#ifndef CAE_F16_ISSUERS
#define CAE_F16_ISSUERS 2  // divisor: NI = NW / CAE_F16_ISSUERS
#endif
    constexpr int NI = NW / CAE_F16_ISSUERS > 0 ? NW / CAE_F16_ISSUERS : 1;
    constexpr int MAXP = (HALO_INSTR + NI - 1) / NI;
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

    const size_t plane_bytes = (size_t)p.H * c8s_row_bytes<SP>(p.W);
    const char *in_n = (const char *)p.in + (size_t)n * p.in_planes * plane_bytes;
    unsigned hoff[MAXP][KS];  // ZP: 0xFFFFFFFF = outside the image (the zero page)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (ISSUER(wave, NW, NI) + i * NI) * 64 + lane;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        const int plhl = pc / PLANE_PIECES;
        const int rem = pc - plhl * PLANE_PIECES;
        const int r = rem / WH, x = rem - r * WH;
        const int xi = S * ox0 - PAD + x;
        const bool xin = xi >= 0 && xi < p.W;
        const unsigned base = (unsigned)(plhl >> 1) * (unsigned)plane_bytes + (unsigned)(plhl & 1) * 512u +
                              c8s_piece<SP>(ZP ? (xin ? xi : 0) : reflect_idx(xi, p.W));
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int yi = S * (oy0 + r) - PAD + ky;
            if (ZP)
                hoff[i][ky] = (xin && yi >= 0 && yi < p.H) ? base + (unsigned)yi * (unsigned)c8s_row_bytes<SP>(p.W)
                                                            : 0xFFFFFFFFu;
            else
                hoff[i][ky] = base + (unsigned)reflect_idx(yi, p.H) * (unsigned)c8s_row_bytes<SP>(p.W);
        }
    }
    const unsigned woff = (unsigned)lane * 16u;

    auto issue_stage = [&](int q, auto ky_tag, char *buf) {
        constexpr int ky = decltype(ky_tag)::value;
        if (ISSUER(wave, NW, NI) < 0) return;
        const int iw = ISSUER(wave, NW, NI);
        const char *wsrc = (const char *)p.wp + (size_t)(q * KS + ky) * W_BYTES;
#ifdef CAE_EXP_F16C_NOWDMA  // timing-only ablation: weights staged for stage 0 only (wrong results)
        if (q == 0 && ky == 0)
#endif
#pragma unroll
        for (int i = 0; i < (W_INSTR + NI - 1) / NI; ++i) {
            const int j = iw + i * NI;
            if (j < W_INSTR) glds16(wsrc + j * 1024 + woff, buf + j * 1024);
        }
        const char *planes = in_n + (size_t)(2 * q) * plane_bytes;
#ifdef CAE_EXP_F16C_NOHDMA  // timing-only ablation: halo staged for stage 0 only (wrong results)
        if (q == 0 && ky == 0)
#endif
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = iw + i * NI;
            if (j < HALO_INSTR) {
                const void *src = (ZP && hoff[i][ky] == 0xFFFFFFFFu) ? (const void *)p.zero : (const void *)(planes + hoff[i][ky]);
                glds16(src, buf + W_BYTES + j * 1024);
            }
        }
    };

    f32x16 acc[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) init_acc<CT>(acc[pt], p.bias, h, 0.0f);

    // B operand of (column tile pt, tap kx): halo [pl = h][hl][row 4w + 2pt + (m>>4)][S (m&15) + kx]
    const int b_off = W_BYTES + (((2 * h) * TY + 2 * PT * wave + (m >> 4)) * WH + S * (m & 15)) * 16;
    constexpr int B_HL = PLANE_PIECES * 16;  // hi -> lo
    constexpr int B_PT = 2 * WH * 16;        // column tile 0 -> 1 (two rows down)
    int sc = 0;

#ifdef CAE_EXP_SETPRIO  // static priority for the younger half of the block (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    issue_stage(0, std::integral_constant<int, 0>{}, smem);
    for (int q = 0; q < p.cci; ++q) {
        static_for<KS>([&](auto ky_tag) {
            constexpr int ky = decltype(ky_tag)::value;
            wait_vm0();
#ifndef CAE_EXP_F16C_NOBAR  // timing-only ablations of this loop (wrong results): profiles/r01_experiments.md
            __syncthreads();
#endif
            char *cur = smem + (sc & 1) * STAGE_BYTES;
            char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
            // the whole next stage is issued here, at the top: spreading the LDS-DMA instructions between the MFMA
            // groups was measured 8-11 % slower (the data simply lands later), profiles/r01_experiments.md
#ifndef CAE_EXP_F16C_NODMA
            if constexpr (ky + 1 < KS) {
                issue_stage(q, std::integral_constant<int, ky + 1>{}, nxt);
            } else {
                if (q + 1 < p.cci) {
                    issue_stage(q + 1, std::integral_constant<int, 0>{}, nxt);
                } else if (GDN) {
                    issue_gamma0<CT, NW>(p, nxt, wave, lane);
                }
            }
#endif
            const char *wb = cur + lane * 16;
            const char *hb = cur + b_off;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#ifdef CAE_EXP_F16C_ONETAP
                constexpr int kxr = 0;  // every tap re-uses tap 0's operands: one third of the LDS reads
#else
                const int kxr = kx;
#endif
                f16x8 bh[PT], bl[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    bh[pt] = *(const f16x8 *)(hb + pt * B_PT + kxr * 16);
                    bl[pt] = *(const f16x8 *)(hb + pt * B_PT + kxr * 16 + B_HL);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f16x8 ah = *(const f16x8 *)(wb + ((kxr * CT + ct) * 2 + 0) * 1024);
                    const f16x8 al = *(const f16x8 *)(wb + ((kxr * CT + ct) * 2 + 1) * 1024);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt][ct] = mfma3(ah, al, bh[pt], bl[pt], acc[pt][ct]);
                }
            }
            ++sc;
        });
    }

    if constexpr (GDN) {
        gdn_stages_f16<CT, PT, NW, ZP, STAGE_BYTES>(acc, p, smem, sc, wave, lane, [](char *) {});
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + 2 * PT * wave + 2 * pt + (m >> 4), ox = ox0 + (m & 15);
        store_tiles_f16<CT, SP, (GDN ? 0 : 1) | (RES ? 2 : 0)>(acc[pt], p, n, oy, ox, h, oy < p.OH && ox < p.OW);
    }
}

// GDN / IGDN with the whole packed gamma resident in LDS (no staging, no barriers): persistent kernels.
template <int CT, bool INVERSE>
__device__ __forceinline__ void gdn_resident_f16(f32x16 (&y)[CT], const char *gbuf, const float *beta_lds, int lane) {
    const int h = lane >> 5;
    f32x16 nrm[CT];
    float mx = 0.0f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            mx = absmax3(mx, y[ct][r], y[ct][r + 1]);
    float isc;
    const float sc = pixel_scale(mx, &isc), sc2 = sc * sc;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) nrm[ct][r] = beta_lds[32 * ct + acc_row(r) + 4 * h] * sc2;
#pragma unroll
    for (int jt = 0; jt < CT; ++jt) {
        const char *gb = gbuf + jt * (CT * 4096) + lane * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_sched_barrier(0);  // keep the y*y splits of later steps from being hoisted (registers)
            u32x4v hw, lw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = y[jt][8 * s + 2 * e] * sc, v1 = y[jt][8 * s + 2 * e + 1] * sc;
                { const u32x2v t_ = split2_f16(v0 * v0, v1 * v1); hw[e] = t_[0]; lw[e] = t_[1]; }
                if constexpr (!INVERSE) {  // keep y sc (see gdn_stages_f16)
                    y[jt][8 * s + 2 * e] = v0;
                    y[jt][8 * s + 2 * e + 1] = v1;
                }
            }
            const f16x8 sh = __builtin_bit_cast(f16x8, hw), sl = __builtin_bit_cast(f16x8, lw);
#pragma unroll
            for (int co = 0; co < CT; ++co) {
                // (at most two fragment pairs in flight: with all CT hoisted the 256-register kernels spilled)
                if (co & 1) __builtin_amdgcn_sched_barrier(0);
                const f16x8 gh = *(const f16x8 *)(gb + ((co * 2 + s) * 2 + 0) * 1024);
                const f16x8 gl = *(const f16x8 *)(gb + ((co * 2 + s) * 2 + 1) * 1024);
                nrm[co] = mfma3(gh, gl, sh, sl, nrm[co]);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float nv = nrm[ct][r];
            if constexpr (INVERSE)
                y[ct][r] *= __builtin_amdgcn_sqrtf(nv) * isc;
            else
                y[ct][r] *= __builtin_amdgcn_rsqf(nv);  // y holds y sc
        }
}

// gdn_resident_f16 with a hand-placed schedule, for kernels that are NOT at the power cap (conv_first_f16: MFMA pipes
// busy 0.29 of the cycles at 1.9 GHz).  The compiler's form alternates per 16-deep step [split 8 squares: ~30 VALU]
// [per channel tile: 2 ds_read -> lgkmcnt(0) -> 3 MFMAs], i.e. the VALU work and 4 LDS latencies per step are serial
// with the MFMAs of the same wave.  Here every MFMA is followed, inside its own 32-cycle shadow, by a slice of the NEXT
// step's split (12 slots: 4 value pairs x {scale + square, split2_f16, -}) and the gamma fragments of the
// next channel tile are read one tile ahead into a second register set.  `sched_barrier(0)` fences pin the order.
// Same operations on the same values in the same accumulation order: results identical to gdn_resident_f16.
template <int CT, bool INVERSE>
__device__ __forceinline__ void gdn_resident_pipe_f16(f32x16 (&y)[CT], const char *gbuf, const float *beta_lds, int lane) {
    const int h = lane >> 5;
    f32x16 nrm[CT];
    float mx = 0.0f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            mx = absmax3(mx, y[ct][r], y[ct][r + 1]);
    float isc;
    const float sc = pixel_scale(mx, &isc), sc2 = sc * sc;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) nrm[ct][r] = beta_lds[32 * ct + acc_row(r) + 4 * h] * sc2;

    constexpr int NSTEP = 2 * CT, CHUNKS = 3 * CT;          // 16-deep steps; MFMAs (= slots for split slices) per step
    constexpr int PPC = (12 + CHUNKS - 1) / CHUNKS;          // split slices per MFMA slot
    u32x4v shw[2], slw[2];                                   // squares of step (set): packed hi / lo pairs
    f16x8 g[2][2];                                           // gamma fragments [set][hi, lo]
    float q[2];
    auto slice = [&](int step, int part) {  // slice `part` (0..11) of the split of step `step` into set step & 1
        // (the empty asm statements give each slice's results a use where the slice stands: without them the compiler sinks
        //  all three slices of a pair to the last one; contraction off: round(v v) - hi as in split_f16, not fma(v, v, -hi))
#pragma clang fp contract(off)
        const int jt = step >> 1, s2 = step & 1, pr = part / 3, set = step & 1;
        if (part % 3 == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float v = y[jt][8 * s2 + 2 * pr + k] * sc;
                q[k] = v * v;
                if constexpr (!INVERSE) y[jt][8 * s2 + 2 * pr + k] = v;  // keep y sc (see gdn_stages_f16)
            }
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
        } else if (part % 3 == 1) {
            { const u32x2v t_ = split2_f16(q[0], q[1]); shw[set][pr] = t_[0]; slw[set][pr] = t_[1]; }
            asm volatile("" : "+v"(shw[set][pr]), "+v"(slw[set][pr]));
        }
    };
    auto load_g = [&](int step, int co, int set) {
        const char *gb = gbuf + (step >> 1) * (CT * 4096) + lane * 16;
        g[set][0] = *(const f16x8 *)(gb + ((co * 2 + (step & 1)) * 2 + 0) * 1024);
        g[set][1] = *(const f16x8 *)(gb + ((co * 2 + (step & 1)) * 2 + 1) * 1024);
    };
#pragma unroll
    for (int part = 0; part < 12; ++part) slice(0, part);
    load_g(0, 0, 0);
    static_for<NSTEP * CT>([&](auto it) {
        constexpr int i = decltype(it)::value, step = i / CT, co = i % CT;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + 1 < NSTEP * CT) load_g((i + 1) / CT, (i + 1) % CT, (i + 1) & 1);
        const f16x8 gh = g[i & 1][0], gl = g[i & 1][1];
        const f16x8 bh = __builtin_bit_cast(f16x8, shw[step & 1]), bl = __builtin_bit_cast(f16x8, slw[step & 1]);
        static_for<3>([&](auto mt) {
            constexpr int m = decltype(mt)::value, chunk = co * 3 + m;
            nrm[co] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m == 2 ? gl : gh, m == 1 ? bl : bh, nrm[co], 0, 0, 0);
            if constexpr (step + 1 < NSTEP) {
#pragma unroll
                for (int part = chunk * PPC; part < (chunk + 1) * PPC && part < 12; ++part) slice(step + 1, part);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    });
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float nv = nrm[ct][r];
            if constexpr (INVERSE)
                y[ct][r] *= __builtin_amdgcn_sqrtf(nv) * isc;
            else
                y[ct][r] *= __builtin_amdgcn_rsqf(nv);  // y holds y sc
        }
}

// =================================================================================================
// gdn_f16_kernel: GDN / IGDN as a kernel of its own, in place on split rows (C8S, SP: C8SP).
//   For layers wider than 128 channels: the fused epilogues keep accumulators + norms of one pixel tile in registers
//   (2 x 16 CT), which fits the 256-register budget of two waves per SIMD up to CT = 4.  Wider layers run their
//   convolution without the normalisation and this kernel after it: one more round trip of the activations, still the
//   f16x3 arithmetic (the pre-GDN values pass through the split format: 2 x 11 bits of mantissa, |y| <= 65504 or the
//   range guard repeats the call on fp32).  The op is pointwise, so a wave simply takes the 32 pixels of one 1-KiB
//   group of a row (their order inside a row does not matter), loads them in the accumulator layout of
//   v_mfma_f32_32x32x16_f16 -- lane (pixel, h) holds channels 4h..4h+3 of every plane: 8 bytes of the hi and of the
//   lo piece -- and reuses gdn_resident_f16 / store_tiles_f16.  PERSISTENT: one 4-wave block per CU, the whole packed gamma
//   (CT^2 x 4 KiB: 144 KiB for 192 channels) resident in LDS.
// =================================================================================================
template <int CT, bool INVERSE, bool SP>
__global__ void __launch_bounds__(256, 1) gdn_f16_kernel(const LayerArgs p) {
    constexpr int NW = 4;  // one wave per SIMD: 2 x 16 CT accumulators + norms need the 512-register budget
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *gbuf = smem;
    float *beta_lds = (float *)(smem + CT * CT * 4096);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    for (int j = wave; j < CT * CT * 4; j += NW) glds16((const char *)p.gp + (size_t)j * 1024 + lane * 16, gbuf + j * 1024);
    for (int i = threadIdx.x; i < CT * 32; i += NW * 64) beta_lds[i] = p.beta[i];
    wait_vm0();
    __syncthreads();

    const int groups = (int)(c8s_row_bytes<SP>(p.OW) / 1024);  // 32-pixel groups per row
    const long total = (long)p.N * p.OH * groups;
    const size_t row_bytes = c8s_row_bytes<SP>(p.OW);
    char *base = (char *)p.out;
    for (long t = (long)blockIdx.x * NW + wave; t < total; t += (long)gridDim.x * NW) {
        const int g = (int)(t % groups);
        const long rest = t / groups;
        const int oy = (int)(rest % p.OH), n = (int)(rest / p.OH);
        const int ox = SP ? (g >> 1) * 64 + 2 * m + (g & 1) : g * 32 + m;
        const bool valid = ox < p.OW;
        f32x16 y[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const size_t row = ((size_t)n * p.out_planes + 4 * ct + pl) * p.OH + oy;
                const char *src = base + row * row_bytes + (size_t)g * 1024 + m * 16 + 8 * h;
                const f16x4 vh = *(const f16x4 *)src, vl = *(const f16x4 *)(src + 512);
#pragma unroll
                for (int k = 0; k < 4; ++k) y[ct][4 * pl + k] = valid ? (float)vh[k] + (float)vl[k] : 0.0f;
            }
        // the norms of HALF the output channel tiles at a time (y: 16 CT registers, norms 8 CT): y*y is split twice,
        // but everything stays in registers (all CT norm tiles at once spilled hundreds of registers even at 512)
        float mx = 0.0f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                mx = absmax3(mx, y[ct][r], y[ct][r + 1]);
        float isc;
        const float sc = pixel_scale(mx, &isc), sc2 = sc * sc;
        constexpr int CH = CT / 2;
        static_assert(CT % 2 == 0, "channel tiles are halved");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 nrm[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[c][r] = beta_lds[32 * (half * CH + c) + acc_row(r) + 4 * h] * sc2;
#pragma unroll
            for (int jt = 0; jt < CT; ++jt) {
                const char *gb = gbuf + jt * (CT * 4096) + lane * 16;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    __builtin_amdgcn_sched_barrier(0);  // keep the y*y splits of later steps from being hoisted
                    u32x4v hw, lw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = y[jt][8 * s2 + 2 * e] * sc, v1 = y[jt][8 * s2 + 2 * e + 1] * sc;
                        { const u32x2v t_ = split2_f16(v0 * v0, v1 * v1); hw[e] = t_[0]; lw[e] = t_[1]; }
                    }
                    const f16x8 sh = __builtin_bit_cast(f16x8, hw), sl = __builtin_bit_cast(f16x8, lw);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int co = half * CH + c;
                        const f16x8 gh = *(const f16x8 *)(gb + ((co * 2 + s2) * 2 + 0) * 1024);
                        const f16x8 gl = *(const f16x8 *)(gb + ((co * 2 + s2) * 2 + 1) * 1024);
                        nrm[c] = mfma3(gh, gl, sh, sl, nrm[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float nv = nrm[c][r];
                    nrm[c][r] = y[half * CH + c][r] * (INVERSE ? __builtin_amdgcn_sqrtf(nv) : __builtin_amdgcn_rsqf(nv)) *
                                (INVERSE ? isc : sc);
                }
            // in place: every input of the tile is in registers already (lanes l and l + 32 share the pixel, hence `valid`)
            if (valid) store_split_f16<CH, SP>(nrm, p, n, oy, ox, h, half * CH);
        }
    }
}

// =================================================================================================
// conv_first_f16_kernel: first analysis layer (Cin <= 4) on f16x3, input uint8 HWC or float NCHW.
//   K = (tap, 4 channels): k-step s, lane half h covers taps 4s+2h, 4s+2h+1 (two float4 halo reads).
//   packed weights: [s][ct][hl][lane][8]: W(cout, tap = 4s + 2(lane>>5) + (j>>2), ch = j&3)
//   PERSISTENT: one block of 8 waves per CU walks over 16x16-pixel output tiles; the packed weights, the
//   whole packed gamma, bias/beta and the /255 table stay in LDS, each wave keeps the input pixels behind its
//   own two output rows in a private LDS buffer and fetches the next tile's into registers under the current
//   tile's MFMAs; after the prologue there is no block barrier.  (The one-tile-per-block form re-read 120 KB
//   of weights and gamma per 128 output pixels and exposed its whole prologue: 1.9 ms per 32 tiles.)
//   The output write (fp32-sized split rows, 134 MB per 1024x1024 tile) bounds this layer.
// =================================================================================================
template <int KS, int CT, bool GDN>
struct FirstGeomF16 {
    static constexpr int NW = 8;
    static constexpr int PAD = KS / 2;
    static constexpr int TX = 16, TY = 2 * NW;
    static constexpr int WH = 2 * TX + KS - 2, HH = 2 * TY + KS - 2;
    static constexpr int NS = (KS * KS + 3) / 4;  // k-steps of 4 taps
    static constexpr int W_BYTES = NS * CT * 2048;
    static constexpr int G_BYTES = GDN ? CT * CT * 4096 : 0;
    // wave-private halo: the KS+2 input rows x WH columns behind a wave's two output rows.  Private buffers need no
    // block barrier per tile (the waves drift apart, so one wave's split / store work overlaps another's MFMAs) and
    // no double buffer (the wave that reads a buffer is the one that refills it).
    static constexpr int RH = KS + 2;
    static constexpr int WAVE_HALO_BYTES = ((RH * WH * 16 + 1023) / 1024) * 1024;
    static constexpr int VEC_BYTES = 1024 + ((2 * CT * 32 * 4 + 1023) / 1024) * 1024;  // /255 table | bias | beta
    static constexpr int FIXED_BYTES = G_BYTES + W_BYTES + VEC_BYTES;
    static constexpr int LDS_BYTES = FIXED_BYTES + NW * WAVE_HALO_BYTES;
    static constexpr int NPOS = (RH * WH + 63) / 64;  // halo pixels per lane
};

template <int KS, int CT, bool GDN, bool U8>
__global__ void __launch_bounds__(512, 1) conv_first_f16_kernel(const LayerArgs p, const FirstArgs f) {
    using G = FirstGeomF16<KS, CT, GDN>;
    constexpr int NW = G::NW, PAD = G::PAD, TX = G::TX, TY = G::TY, WH = G::WH, HH = G::HH, NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *gbuf = smem;
    char *wbuf = smem + G::G_BYTES;
    float *lut = (float *)(wbuf + G::W_BYTES);
    float *bias_lds = lut + 256;
    float *beta_lds = bias_lds + CT * 32;
    char *hbuf = smem + G::FIXED_BYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int total = p.N * tiles_img;

    // resident operands: gamma and weights by LDS-DMA (gbuf and wbuf are adjacent), small vectors by hand
    for (int j = wave; j < (G::G_BYTES + G::W_BYTES) / 1024; j += NW) {
        const char *src = j < G::G_BYTES / 1024 ? (const char *)p.gp + (size_t)j * 1024
                                                 : (const char *)p.wp + (size_t)(j - G::G_BYTES / 1024) * 1024;
        glds16(src + lane * 16, smem + j * 1024);
    }
    for (int i = threadIdx.x; i < 256; i += NW * 64) lut[i] = (float)i / 255.0f;
    for (int i = threadIdx.x; i < CT * 32; i += NW * 64) {
        bias_lds[i] = p.bias ? p.bias[i] : 0.0f;
        beta_lds[i] = (GDN && p.beta) ? p.beta[i] : 1.0f;
    }

    // next tile's input pixels of THIS WAVE's halo, raw (4 bytes or 4 floats per pixel).  Branch-free on purpose:
    // every load is issued unconditionally (clamped pixel / channel index) so that all of them stay in flight under
    // the current tile's MFMAs; absent channels are zeroed when the values are committed to LDS.
    // (the input format is a template parameter so that the loaded registers have a single definition: with a
    //  run-time format the compiler joined the two load paths and waited for the data right after issuing)
    typedef typename std::conditional<U8, uint8_t, float>::type raw_t;
    raw_t raw[G::NPOS][4];
    const int c_last = f.cin - 1;
    auto fetch = [&](int t) {
        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int k = 0; k < G::NPOS; ++k) {
            int i = lane + k * 64;
            i = i < G::RH * WH ? i : G::RH * WH - 1;
            const int r = i / WH, x = i - r * WH;
            const int iy = reflect_idx(2 * (ty * TY + 2 * wave) - PAD + r, p.H);
            const int ix = reflect_idx(2 * tx * TX - PAD + x, p.W);
            if constexpr (U8) {
                const uint8_t *src = (const uint8_t *)f.in + (((size_t)n * p.H + iy) * p.W + ix) * f.cin;
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[k][c] = src[c < c_last ? c : c_last];
            } else {
                const float *src = (const float *)f.in + (size_t)n * f.cin * p.H * p.W + (size_t)iy * p.W + ix;
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[k][c] = src[(size_t)(c < c_last ? c : c_last) * p.H * p.W];
            }
        }
    };
    char *whalo = hbuf + wave * G::WAVE_HALO_BYTES;
    auto commit = [&]() {  // exact x/255 through the table
#pragma unroll
        for (int k = 0; k < G::NPOS; ++k) {
            const int i = lane + k * 64;
            if (i < G::RH * WH) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float val;
                    if constexpr (U8) val = lut[raw[k][c]]; else val = raw[k][c];
                    v[c] = c <= c_last ? val : 0.0f;
                }
                *(f32x4 *)(whalo + i * 16) = v;
            }
        }
    };

    int t = blockIdx.x;
    if (t < total) fetch(t);
    wait_vm0();
    __syncthreads();  // resident operands and the table visible to every wave; the only block barrier
    if (t < total) commit();
    __builtin_amdgcn_wave_barrier();

    const char *hb = whalo + ((2 * (m >> 4)) * WH + 2 * (m & 15)) * 16;
    const char *wb = wbuf + lane * 16;
    // (starting the two waves of a SIMD half a tile period apart, so that one's VALU phase meets the other's MFMA
    //  phase, was measured: no effect -- profiles/r01_experiments.md: free-running waves drift back.)
    // Holding the two waves of a SIMD in opposite phases with block barriers (upper half one phase behind, GDN of one
    // wave against convolution / split / store of the other) was measured too: 1.39 vs 1.26 ms (r02_experiments.md 11).
    for (; t < total; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool has_next = tn < total;
        if (has_next) fetch(tn);

        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = bias_lds[32 * ct + acc_row(r) + 4 * h];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            u32x4v bhw, blw;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // tap index of this lane half: 4s + 2h + half  (compile-time per h)
                const int t0 = 4 * s + half, t1 = 4 * s + 2 + half;
                const int o0 = t0 < KS * KS ? ((t0 / KS) * WH + (t0 % KS)) * 16 : -1;
                const int o1 = t1 < KS * KS ? ((t1 / KS) * WH + (t1 % KS)) * 16 : -1;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
                if (o0 >= 0) v0 = *(const f32x4 *)(hb + o0);
                if (o1 >= 0) v1 = *(const f32x4 *)(hb + o1);
                const f32x4 v = h ? v1 : v0;
                { const u32x2v t_ = split2_f16(v[0], v[1]); bhw[2 * half] = t_[0]; blw[2 * half] = t_[1]; }
                { const u32x2v t_ = split2_f16(v[2], v[3]); bhw[2 * half + 1] = t_[0]; blw[2 * half + 1] = t_[1]; }
            }
            const f16x8 bh = __builtin_bit_cast(f16x8, bhw), bl = __builtin_bit_cast(f16x8, blw);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f16x8 ah = *(const f16x8 *)(wb + ((s * CT + ct) * 2 + 0) * 1024);
                const f16x8 al = *(const f16x8 *)(wb + ((s * CT + ct) * 2 + 1) * 1024);
                acc[ct] = mfma3(ah, al, bh, bl, acc[ct]);
            }
        }
        // refill this wave's halo as soon as the convolution has read it: LDS operations of one wave execute in order,
        // the wave barriers only keep the compiler from moving the LDS stores above the reads of the tile just finished
        // (or the next tile's reads above them).  BEFORE this tile's output stores on purpose: vmcnt counts loads and
        // stores together in issue order, and with the commit after the stores the wait for the prefetched pixels came
        // out as vmcnt(11..0) on the merged control flow -- a drain of all 16 output stores (an HBM write round trip) per
        // tile.  Here the only older operations are the previous tile's stores, a whole tile period old.
        __builtin_amdgcn_wave_barrier();
        if (has_next) commit();
        __builtin_amdgcn_wave_barrier();
#if CAE_FIRST_PIPE
        if constexpr (GDN) gdn_resident_pipe_f16<CT, false>(acc, gbuf, beta_lds, lane);
#else
        if constexpr (GDN) gdn_resident_f16<CT, false>(acc, gbuf, beta_lds, lane);
#endif
        const int oy = ty * TY + 2 * wave + (m >> 4), ox = tx * TX + (m & 15);
        store_tiles_f16<CT, false, !GDN>(acc, p, n, oy, ox, h, oy < p.OH && ox < p.OW);
    }
}

// ---- last-layer product map ("pmap") ------------------------------------------------------------------------
// The last synthesis layer (ConvTranspose2d 128 -> 3, _autoencoders.py:204-211) reads its 128-channel input once per
// tile: 134 MB for 0.9 GMAC, the most memory-bound kernel of the step.  Its channel contraction commutes with the
// spatial gather: out[2a+py][2b+px][c] = bias[c] + sum_{d,dx} P[a-d][b-dx][tap(d,dx)][c] with
// P[pixel][tap][c] = sum_j x[pixel][j] W[j][c][tap] a POINTWISE linear map of the input pixel.  So the layer that produces
// x applies that map to its accumulator tile (one more MFMA chain, the accumulators as B operand in their own row
// order exactly as in the fused IGDN) and stores 27 (padded to 32) fp32 values per pixel instead of 128 split
// channels: 128 B instead of 512 B written, and the last layer becomes a 4-term gather (pmap_gather_kernel) that
// reads 128 B per pixel instead of 512 B.  Same fp32-class arithmetic (f16x3 products, fp32 accumulation); only the
// summation order of the last layer changes.
//   packed map: [jt][s(2)][hl(2)][64 lanes][8 f16]: A(row = lane&31 = record slot (tap, c) in the order pmap_gather_kernel
//   reads them, k = 32jt + row(8s + e) + 4(lane>>5))
//   map in HBM: fp32 [N][OH][2][OW/2][32]: rows split by pixel parity (as C8SP), because a sub-pixel phase produces every
//   second pixel: the 32 pixels of a wave are then 32 consecutive 128-byte records.
__host__ __device__ __forceinline__ size_t pmap_record(int n, int OH, int OW, int oy, int ox) {
    return ((((size_t)n * OH + oy) * 2 + (ox & 1)) * (OW >> 1) + (ox >> 1)) * 32;
}

// j0: index (in its parity row) of the wave's first pixel; tbuf: 4 KiB of LDS private to the wave.  The accumulator
// layout gives every lane four scattered 16-byte pieces of its pixel's record; written like that (16 bytes per lane at a
// 128-byte stride) the store cost 0.25 ms of deconv3's 3.07 (CAE_EXP_PMAP_NOSTORE).  So the tile is transposed through
// LDS (XOR-swizzled pieces: conflict-free both ways) and leaves as four instructions of 1 KiB contiguous each.
template <int CT>
__device__ __forceinline__ void store_pmap_f16(const f32x16 (&y)[CT], const LayerArgs &p, char *tbuf, int n, int oy, int par,
                                               int j0, int lane, bool row_valid) {
    const int h = lane >> 5, m = lane & 31;
    f32x16 pm;
#pragma unroll
    for (int r = 0; r < 16; ++r) pm[r] = 0.0f;
    float mx = 0.0f;
    const char *ab = (const char *)p.pm + lane * 16;
#pragma unroll
    for (int jt = 0; jt < CT; ++jt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4v hw, lw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = y[jt][8 * s + 2 * e], v1 = y[jt][8 * s + 2 * e + 1];
                mx = absmax3(mx, v0, v1);
                const u32x2v t_ = split2_f16(v0, v1);
                hw[e] = t_[0];
                lw[e] = t_[1];
            }
            const f16x8 sh = __builtin_bit_cast(f16x8, hw), sl = __builtin_bit_cast(f16x8, lw);
            const f16x8 ah = *(const f16x8 *)(ab + ((jt * 2 + s) * 2 + 0) * 1024);
            const f16x8 al = *(const f16x8 *)(ab + ((jt * 2 + s) * 2 + 1) * 1024);
            pm = mfma3(ah, al, sh, sl, pm);
        }
    raise_if_over(mx, p.flag);
#ifdef CAE_EXP_PMAP_NOSTORE  // timing-only ablation: the map is computed but not written (wrong results)
    if (p.N > 0) return;
#endif
    // D: register r = map row acc_row(r) + 4h: lane (m, h) holds pieces q = 2g + h (g = 0..3) of pixel m's record
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = {pm[4 * g], pm[4 * g + 1], pm[4 * g + 2], pm[4 * g + 3]};
        *(f32x4 *)(tbuf + m * 128 + (((2 * g + h) ^ (m & 7)) * 16)) = v;
    }
    __builtin_amdgcn_wave_barrier();
    const int half_w = p.OW >> 1;
    float *row = (float *)p.out + ((((size_t)n * p.OH + oy) * 2 + par) * half_w) * 32;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int px = (lane >> 3) + 8 * it, q = lane & 7;
        const f32x4 v = *(const f32x4 *)(tbuf + px * 128 + ((q ^ (px & 7)) * 16));
        if (row_valid && j0 + px < half_w) *(f32x4 *)(row + (size_t)(j0 + px) * 32 + 4 * q) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// out[n][2a+py][2b+px][c] = bias[c] + sum over (d, dx) of P[a-d][b-dx][(2d+py+1)*3 + (2dx+px+1)][c]   (k = 3)
// block = 16 x 16 input pixels (a, b) -> 32 x 32 output pixels.  The 17 x 17 records (one halo row / column) are staged
// in LDS with coalesced 16-byte loads (record pitch 36 floats), every thread sums the <= 4 terms of its
// 2 x 2 output pixels, and uint8 output rows leave through LDS as whole dwords (the per-thread byte stores of the first
// version made this kernel as slow as the layer it replaces).  uint8 HWC (x255, clip, truncate) or fp32 NCHW.
static __global__ void __launch_bounds__(256) pmap_gather_kernel(const float *pm, const float *bias, void *out, int N, int H,
                                                               int W, int cout, int fmt, int tiles_x, int tiles_y) {
    // record pitch 36 floats: 16-byte aligned pieces (vector LDS stores / loads) and conflict-free 16-byte reads of 16
    // consecutive records (36 i mod 64 steps through all 16 four-bank groups)
    constexpr int T = 16, R = T + 1, PITCH = 36;
    __shared__ __attribute__((aligned(16))) float rec[R * R * PITCH];
    __shared__ __attribute__((aligned(4))) uint8_t orow[2 * T][2 * T * 3 + 4];
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int a0 = ty * T, b0 = tx * T;
    // all of a thread's loads first, then its LDS stores: written as one loop the compiler kept ONE load in flight per
    // thread (load, vmcnt(0), LDS store, next): 12 KB in flight per CU, the kernel ran at 3 TB/s of its 1.2 GB
    constexpr int NIT = (R * R * 8 + 255) / 256;
    f32x4 stage[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int q = i & 7, r = i >> 3;
        const int ra = r / R, rb = r - ra * R;
        const int a = a0 + ra, b = b0 + rb;
        stage[it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // records outside the image contribute nothing
        if (i < R * R * 8 && a < H && b < W) stage[it] = *(const f32x4 *)(pm + pmap_record(n, H, W, a, b) + 4 * q);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < R * R * 8) *(f32x4 *)(rec + (i >> 3) * PITCH + 4 * (i & 7)) = stage[it];
    }
    __syncthreads();
    const int la = threadIdx.x >> 4, lb = threadIdx.x & 15;
    const int a = a0 + la, b = b0 + lb;
    const float *p00 = rec + (la * R + lb) * PITCH, *p01 = p00 + PITCH, *p10 = p00 + R * PITCH, *p11 = p10 + PITCH;
    const int OH = 2 * H, OW = 2 * W;
    const bool u8 = fmt == OUT_U8HWC;
    // The record's 32 slots are ordered by WHO reads them (pack_pmap_f16, cae_api.hip), 3 channels per tap t = 3 ky + kx:
    //   [0..11] taps 4, 5, 7, 8: this pixel's own terms of its four outputs      (py = 0: ky = 1 at row a; py = 1: ky = 2
    //   [12..17] taps 3, 6: terms for the outputs of pixel (a, b - 1): read from p01 = record (a, b + 1); at row a and
    //   [20..25] taps 1, 2: terms for pixel (a - 1, b): read from p10 = record (a + 1, b);           ky = 0 at row a + 1;
    //   [28..30] tap 0:     term for pixel (a - 1, b - 1): read from p11 = record (a + 1, b + 1)     same for columns)
    // so a thread needs eight 16- / 8-byte LDS reads instead of 27 scalar ones.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x4 s0 = *(const f32x4 *)(p00), s1 = *(const f32x4 *)(p00 + 4), s2 = *(const f32x4 *)(p00 + 8);
    const f32x4 l0 = *(const f32x4 *)(p01 + 12), u0 = *(const f32x4 *)(p10 + 20), d0 = *(const f32x4 *)(p11 + 28);
    const f32x2 l1 = *(const f32x2 *)(p01 + 16), u1 = *(const f32x2 *)(p10 + 24);
    const float self[12] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
    const float left[6] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1]};  // taps 3, 6 of record (a, b + 1)
    const float up[6] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1]};    // taps 1, 2 of (a + 1, b)
    const float diag[3] = {d0[0], d0[1], d0[2]};                       // tap 0 of (a + 1, b + 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c >= cout) break;
        const float bs = bias ? bias[c] : 0.0f;
        const float o00 = bs + self[c];
        const float o01 = bs + self[3 + c] + left[c];
        const float o10 = bs + self[6 + c] + up[c];
        const float o11 = bs + self[9 + c] + left[3 + c] + up[3 + c] + diag[c];
        if (u8) {
            orow[2 * la][(2 * lb) * cout + c] = clip_u8(o00 * 255.0f);
            orow[2 * la][(2 * lb + 1) * cout + c] = clip_u8(o01 * 255.0f);
            orow[2 * la + 1][(2 * lb) * cout + c] = clip_u8(o10 * 255.0f);
            orow[2 * la + 1][(2 * lb + 1) * cout + c] = clip_u8(o11 * 255.0f);
        } else if (a < H && b < W) {
            float *o = (float *)out + (((size_t)n * cout + c) * OH + 2 * a) * OW + 2 * b;
            o[0] = o00;
            o[1] = o01;
            o[OW] = o10;
            o[OW + 1] = o11;
        }
    }
    if (!u8) return;
    __syncthreads();
    // rows of 2T pixels x cout bytes; whole dwords when the row segment is dword-aligned and inside the image
    const int row_bytes = 2 * T * cout;
    const int valid_rows = min(2 * T, OH - 2 * a0), valid_bytes = min(2 * T, OW - 2 * b0) * cout;
    uint8_t *obase = (uint8_t *)out + (((size_t)n * OH + 2 * a0) * OW + 2 * b0) * cout;
    const bool dwords = ((((size_t)OW * cout) & 3) == 0) && (((uintptr_t)obase & 3) == 0) && (row_bytes & 3) == 0 &&
                        valid_bytes == row_bytes;
    if (dwords) {
        const int per_row = row_bytes / 4;
        for (int i = threadIdx.x; i < valid_rows * per_row; i += 256) {
            const int r = i / per_row, q = i - r * per_row;
            *(uint32_t *)(obase + (size_t)r * OW * cout + 4 * q) = *(const uint32_t *)(&orow[r][4 * q]);
        }
    } else {
        for (int i = threadIdx.x; i < valid_rows * valid_bytes; i += 256) {
            const int r = i / valid_bytes, q = i - r * valid_bytes;
            obase[(size_t)r * OW * cout + q] = orow[r][q];
        }
    }
}

// =================================================================================================
// deconv_s2_f16_kernel: stride-2 transposed conv (+bias) (+IGDN), f16x3, C8S in.
//   block = NW waves, wave w owns INPUT row ty0+w, 32 input columns; per output-row parity py two
//   accumulator sets (px = 0, 1).  stage = (py, 16-channel chunk q, kernel row ky(py, d)):
//     weights [kx][ct][hl][64][8 f16]  +  halo [pl][hl][NW rows][WH][16 B]
// =================================================================================================
template <int KS, int CT, int NW, int PT, bool IGDN>
struct DeconvGeomF16 {
    static constexpr int P = KS / 2;
    static constexpr int DLO = -((P + 1) / 2), DHI = (KS - 1 - P) / 2;
    static constexpr int WH = 32 + DHI - DLO;
    static constexpr int ROWS = NW * PT;  // input rows per block
    // halo image (as in conv_s2_f16_kernel): the 32 columns ix0 .. ix0 + 31 are whole 256-byte runs of the C8SP rows
    // and form an aligned image [pl][hl][row][32]; the DHI columns to the left and -DLO to the right are strays,
    // [stray][pl][hl][row], packed into one more LDS-DMA instruction
    static constexpr int NSTRAY = WH - 32;
    static constexpr int AL_INSTR = 4 * ROWS * 32 / 64;
    static constexpr int AL_BYTES = AL_INSTR * 1024;
    static constexpr int STRAY_INSTR = (NSTRAY * 4 * ROWS + 63) / 64;
    static constexpr int HALO_INSTR = AL_INSTR + STRAY_INSTR;
    static_assert((4 * ROWS * 32) % 64 == 0 && 64 % (4 * ROWS) == 0, "halo image assumes 4, 8 or 16 input rows per block");
    static constexpr int W_INSTR = KS * CT * 2;
    static constexpr int W_BYTES = W_INSTR * 1024;
    static constexpr int G_BYTES = IGDN ? CT * 4096 : 0;
    static constexpr int CONV_STAGE = W_BYTES + HALO_INSTR * 1024;
    static constexpr int STAGE_BYTES = CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES;
    static constexpr int NI = NW / CAE_F16_ISSUERS > 0 ? NW / CAE_F16_ISSUERS : 1;  // LDS-DMA issuer waves (conv_s2_f16)
    static constexpr int MAXP = (HALO_INSTR + NI - 1) / NI;
    // aligned pieces per issuer and their (plane, half) step: piece i of issuer iw is instruction iw + i NI, i.e. halo
    // piece (iw 64 + lane) + i NI 64 -- the same (row, column), AL_STEP (plane, half) images further
    static constexpr int AL_PER = AL_INSTR / NI, AL_STEP = NI * 64 / (ROWS * 32);
    static_assert(AL_INSTR % NI == 0 && (NI * 64) % (ROWS * 32) == 0 && AL_STEP <= 2 && STRAY_INSTR <= NI,
                  "issuer waves must each cover whole (plane, half) images of the aligned halo");
    // IGDN with the WHOLE packed gamma resident in LDS (loaded once per block) when it fits beside the two stage buffers:
    // the streamed form re-fetched gamma for every (px, row) tile -- 8 passes per block, each of CT stages with a barrier,
    // a vmcnt wait and only 24 MFMAs per wave to hide them behind.
    static constexpr int G_ALL = IGDN ? CT * CT * 4096 : 0;
    static constexpr bool RESIDENT = IGDN && 2 * STAGE_BYTES + G_ALL <= 160 * 1024;
    static constexpr int PMAP_LDS = NW * 4096;  // OUT_PMAP: one 4-KiB transpose buffer per wave
    // ... which live behind everything else, or -- when that exceeds the LDS -- in the stage buffer the K loop has
    // just finished with (one block barrier before the epilogue)
    static constexpr bool TBUF_IN_STAGE = RESIDENT && 2 * STAGE_BYTES + G_ALL + PMAP_LDS > 160 * 1024;
    static_assert(!TBUF_IN_STAGE || STAGE_BYTES >= PMAP_LDS, "transpose buffers must fit a stage buffer");
    static constexpr int lds_bytes(bool pmap) {
        return 2 * STAGE_BYTES + (RESIDENT ? G_ALL : 0) + ((pmap && !TBUF_IN_STAGE) ? PMAP_LDS : 0);
    }
    static constexpr int dmin(int py) { return -((py + P) / 2); }
    static constexpr int nky(int py) { return (KS - 1 - py - P) / 2 - dmin(py) + 1; }
};

template <int KS, int CT, int NW, int PT, bool IGDN, int PY>
__device__ __forceinline__ void deconv_issue_f16(const LayerArgs &p, const char *in_n, size_t plane_bytes, int s,
                                                 char *buf, const int *hrow, const int *hbase, int wave, int lane) {
    using G = DeconvGeomF16<KS, CT, NW, PT, IGDN>;
    constexpr int NKY = G::nky(PY);
    const int q = s / NKY, d = G::dmin(PY) + (s - q * NKY);
    const int ky = 2 * d + PY + G::P;
    if (ISSUER(wave, NW, G::NI) < 0) return;
    const int iw = ISSUER(wave, NW, G::NI);
    const char *wsrc = (const char *)p.wp + (size_t)(q * KS + ky) * G::W_BYTES;
#pragma unroll
    for (int i = 0; i < (G::W_INSTR + G::NI - 1) / G::NI; ++i) {
        const int j = iw + i * G::NI;
        if (j < G::W_INSTR) glds16(wsrc + j * 1024 + lane * 16, buf + j * 1024);
    }
    const char *planes = in_n + (size_t)(2 * q) * plane_bytes;
    const size_t row_bytes = c8s_row_bytes<true>(p.W);
    {
        // aligned image: the issuer's pieces i = 0 .. AL_PER - 1 are the same (row, column) of consecutive
        // (plane, half) images, so ONE lane offset + uniform strides address them all (hbase[0], hrow[0])
        const int iy = hrow[0] - d;
        const bool ok = iy >= 0 && iy < p.H && hbase[0] >= 0;
        const char *src = planes + hbase[0] + (size_t)iy * row_bytes;
#pragma unroll
        for (int i = 0; i < G::AL_PER; ++i) {
            constexpr int K = G::AL_STEP;
            const char *g = ok ? src + (size_t)((i * K) >> 1) * plane_bytes + ((i * K) & 1) * 512 : (const char *)p.zero;
            glds16(g, buf + G::W_BYTES + (iw + i * G::NI) * 1024);
        }
    }
    if (iw < G::STRAY_INSTR) {  // stray columns: one instruction per issuer (hbase[1], hrow[1])
        const int iy = hrow[1] - d;
        const bool ok = iy >= 0 && iy < p.H && hbase[1] >= 0;
        const char *g = ok ? planes + hbase[1] + (size_t)iy * row_bytes : (const char *)p.zero;
        glds16(g, buf + G::W_BYTES + (G::AL_INSTR + iw) * 1024);
    }
}

template <int KS, int CT, int NW, int PT, bool IGDN, int PY>
__device__ __forceinline__ void deconv_phase_f16(const LayerArgs &p, const char *in_n, size_t plane_bytes, char *smem,
                                                 int &sc, const int *hrow, const int *hbase, int wave, int lane,
                                                 const int (&b_off)[KS], int stray_mask, int n, int iy, int ix) {
    using G = DeconvGeomF16<KS, CT, NW, PT, IGDN>;
    constexpr int P = G::P;
    constexpr int STAGE_BYTES = G::STAGE_BYTES;
    const int h = lane >> 5;
    const int NS = p.cci * G::nky(PY);
    f32x16 acc[2][PT][CT];  // [px][row tile][ct]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) init_acc<CT>(acc[x][pt], p.bias, h, 0.0f);

    for (int s = 0; s < NS; ++s) {
        wait_vm0();
        __syncthreads();
        char *cur = smem + (sc & 1) * STAGE_BYTES;
        char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
#ifndef CAE_EXP_F16_NODMA  // timing-only ablation: no LDS-DMA after the first stage (wrong results)
        if (s + 1 < NS) {
            deconv_issue_f16<KS, CT, NW, PT, IGDN, PY>(p, in_n, plane_bytes, s + 1, nxt, hrow, hbase, wave, lane);
        } else if (IGDN && !G::RESIDENT) {
            issue_gamma0<CT, NW>(p, nxt, wave, lane);
        } else if (PY == 0) {
            deconv_issue_f16<KS, CT, NW, PT, IGDN, 1>(p, in_n, plane_bytes, 0, nxt, hrow, hbase, wave, lane);
        }
#endif
        const char *wb = cur + lane * 16;
#if CAE_F16_PIPE
        if constexpr (PT == 1) {
            // software-pipelined stage: the operand fragments of group pair g + 1 are read from LDS before the six
            // MFMAs of pair g are issued (one register set ahead), and the two groups of a pair (two channel tiles of a
            // tap) alternate so that consecutive MFMAs never depend on each other.  Without this the compiler
            // re-used ONE fragment register set: read -> lgkmcnt(0) -> 1-2 MFMAs, the LDS latency exposed 24 times
            // per stage and wave.
            static_assert(CT % 2 == 0 || CT == 1, "channel tiles are paired");
            constexpr int GP = CT == 1 ? 1 : 2;          // groups (channel tiles) per step
            constexpr int NSTEP = KS * CT / GP;
            f16x8 a[2][GP][2], b[2][2];                  // [set][group][hi, lo], [set][hi, lo]
            auto load_b = [&](int kx, int set) {
                const bool stray = (stray_mask >> kx) & 1;
                const int b_hl = stray ? G::ROWS * 16 : G::ROWS * 32 * 16;
                b[set][0] = *(const f16x8 *)(cur + b_off[kx]);
                b[set][1] = *(const f16x8 *)(cur + b_off[kx] + b_hl);
            };
            auto load_a = [&](int step, int set) {
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    a[set][g][0] = *(const f16x8 *)(wb + ((step * GP + g) * 2 + 0) * 1024);
                    a[set][g][1] = *(const f16x8 *)(wb + ((step * GP + g) * 2 + 1) * 1024);
                }
            };
            load_b(0, 0);
            load_a(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * GP, 0);
            static_for<NSTEP>([&](auto it) {
                constexpr int st = decltype(it)::value;
                constexpr int kx = st * GP / CT, ct0 = st * GP % CT;
                constexpr int px = (kx + P) & 1;
                constexpr bool more = st + 1 < NSTEP, next_tap = more && (st + 1) * GP % CT == 0;
                constexpr int nread = more ? 2 * GP + (next_tap ? 2 : 0) : 0;
                if constexpr (more) load_a(st + 1, (st + 1) & 1);
                if constexpr (next_tap) load_b(kx + 1, (kx + 1) & 1);
                const f16x8 bh = b[kx & 1][0], bl = b[kx & 1][1];
                f32x16 *dst = px == 0 ? acc[0][0] : acc[1][0];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int g = 0; g < GP; ++g)
                        dst[ct0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st & 1][g][m == 2], m == 1 ? bl : bh,
                                                                              dst[ct0 + g], 0, 0, 0);
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);  // next step's reads,
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * GP, 0);                         // then this step's MFMAs
            });
        } else
#endif
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            if (PT > 1) __builtin_amdgcn_sched_barrier(0);  // bound operand live ranges to one tap (register budget)
            const int px = (kx + P) & 1;
            f16x8 bh[PT], bl[PT];
            // (hi -> lo and row -> row strides of the aligned image or of a stray column's, per lane and tap)
            const bool stray = (stray_mask >> kx) & 1;
            const int b_hl = stray ? G::ROWS * 16 : G::ROWS * 32 * 16, b_pt = stray ? 16 : 32 * 16;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                bh[pt] = *(const f16x8 *)(cur + b_off[kx] + pt * b_pt);
                bl[pt] = *(const f16x8 *)(cur + b_off[kx] + pt * b_pt + b_hl);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#ifdef CAE_EXP_F16_FEWREADS  // timing-only ablation: one weight fragment pair per tap instead of CT (wrong results)
                constexpr int ctr = 0;
#else
                const int ctr = ct;
#endif
                const f16x8 ah = *(const f16x8 *)(wb + ((kx * CT + ctr) * 2 + 0) * 1024);
                const f16x8 al = *(const f16x8 *)(wb + ((kx * CT + ctr) * 2 + 1) * 1024);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    if (px == 0)
                        acc[0][pt][ct] = mfma3(ah, al, bh[pt], bl[pt], acc[0][pt][ct]);
                    else
                        acc[1][pt][ct] = mfma3(ah, al, bh[pt], bl[pt], acc[1][pt][ct]);
                }
            }
        }
        ++sc;
    }

    // transpose buffer of this wave for the product-map store
    char *tbuf = smem + 2 * STAGE_BYTES + (G::RESIDENT ? G::G_ALL : 0) + wave * 4096;
    if constexpr (G::TBUF_IN_STAGE) {
        __syncthreads();  // every wave is done with the last stage's buffer (sc was incremented past it)
        tbuf = smem + ((sc + 1) & 1) * STAGE_BYTES + wave * 4096;
    }
    auto store = [&](auto x_tag, auto pt_tag) {
        constexpr int x = decltype(x_tag)::value, pt = decltype(pt_tag)::value;
        if (p.outfmt == OUT_PMAP)
            store_pmap_f16<CT>(acc[x][pt], p, tbuf, n, 2 * (iy + pt) + PY, x, ix - (lane & 31), lane, (iy + pt) < p.H);
        else
            store_tiles_f16<CT, true, !IGDN && CT <= 4>(acc[x][pt], p, n, 2 * (iy + pt) + PY, 2 * ix + x, h, (iy + pt) < p.H && ix < p.W);
    };
    if constexpr (IGDN && G::RESIDENT) {
        // gamma resident: no staging, no barriers; one (px, row tile) at a time (64 norm accumulators live)
        const char *gbuf = smem + 2 * STAGE_BYTES;
        static_for<2 * PT>([&](auto it) {
            constexpr int idx = decltype(it)::value;
            constexpr int x = idx / PT, pt = idx % PT;
            gdn_resident_f16<CT, true>(acc[x][pt], gbuf, p.beta, lane);
            store(std::integral_constant<int, x>{}, std::integral_constant<int, pt>{});
        });
    } else if constexpr (IGDN) {
        // one (px, row tile) at a time: 64 norm accumulators live instead of 128 (register budget);
        // gamma is re-streamed per tile (L2-resident, 64 KiB)
        static_for<2 * PT>([&](auto it) {
            constexpr int idx = decltype(it)::value;
            constexpr int x = idx / PT, pt = idx % PT;
            gdn_stages_f16<CT, PT, NW, true, STAGE_BYTES, pt, 1>(acc[x], p, smem, sc, wave, lane, [&](char *nxt) {
                if constexpr (idx + 1 < 2 * PT) {
                    issue_gamma0<CT, NW>(p, nxt, wave, lane);
                } else if (PY == 0) {
                    deconv_issue_f16<KS, CT, NW, PT, IGDN, 1>(p, in_n, plane_bytes, 0, nxt, hrow, hbase, wave, lane);
                }
            });
            store(std::integral_constant<int, x>{}, std::integral_constant<int, pt>{});
        });
    } else {
        static_for<2 * PT>([&](auto it) {
            constexpr int idx = decltype(it)::value;
            store(std::integral_constant<int, idx / PT>{}, std::integral_constant<int, idx % PT>{});
        });
    }
}

// PT input rows per wave: PT = 2 halves the LDS bytes read per MFMA (4 fat waves, one block per CU)
template <int KS, int CT, int NW, int PT, bool IGDN>
__global__ void __launch_bounds__(NW * 64, NW * PT <= 4 ? 2 : 1) deconv_s2_f16_kernel(const LayerArgs p) {
    using G = DeconvGeomF16<KS, CT, NW, PT, IGDN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    int bid = xcd_tile_order(blockIdx.x, gridDim.x);
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int iy0 = ty * G::ROWS, ix0 = tx * 32;

    // LDS-DMA sources of this lane: [0] its piece of the aligned image (first of AL_PER), [1] its stray piece.
    // hbase = byte offset of (plane-in-chunk, half, column) inside the chunk (< 2^31: two planes), < 0: outside;
    // hrow = input row of the piece at d = 0
    int hrow[2], hbase[2];
    {
        const int iw = ISSUER(wave, NW, G::NI) < 0 ? 0 : ISSUER(wave, NW, G::NI);
        const int plane_i = (int)((size_t)p.H * c8s_row_bytes<true>(p.W));
        const int pc = iw * 64 + lane;
        int plhl = pc / (G::ROWS * 32);
        int ixp = ix0 + (pc & 31);
        hrow[0] = iy0 + (pc / 32) % G::ROWS;
        hbase[0] = ixp < p.W ? (plhl >> 1) * plane_i + (int)c8s_piece<true>(ixp) + (plhl & 1) * 512 : -1;
        int js = iw * (64 / (4 * G::ROWS)) + lane / (4 * G::ROWS);
        js = js < G::NSTRAY ? js : G::NSTRAY - 1;
        plhl = (lane / G::ROWS) & 3;
        ixp = ix0 + (js < G::DHI ? js : 32 + js) - G::DHI;
        hrow[1] = iy0 + lane % G::ROWS;
        hbase[1] = (ixp >= 0 && ixp < p.W) ? (plhl >> 1) * plane_i + (int)c8s_piece<true>(ixp) + (plhl & 1) * 512 : -1;
    }
    const size_t plane_bytes = (size_t)p.H * c8s_row_bytes<true>(p.W);  // input rows are C8SP
    const char *in_n = (const char *)p.in + (size_t)n * p.in_planes * plane_bytes;
    // B operand of tap kx: plane h, halo row PT*wave + pt, halo column m + DHI - dx(kx): aligned image or a stray's
    int b_off[KS], stray_mask = 0;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        const int pxk = (kx + G::P) & 1, dx = (kx - G::P - pxk) / 2;
        const int c = m + G::DHI - dx, row = PT * wave;
        if (c >= G::DHI && c < G::DHI + 32) {
            b_off[kx] = G::W_BYTES + (((2 * h) * G::ROWS + row) * 32 + c - G::DHI) * 16;
        } else {
            const int js = c < G::DHI ? c : c - 32;
            b_off[kx] = G::W_BYTES + G::AL_BYTES + ((js * 4 + 2 * h) * G::ROWS + row) * 16;
            stray_mask |= 1 << kx;
        }
    }
    const int iy = iy0 + PT * wave, ix = ix0 + m;
    int sc = 0;
    if constexpr (G::RESIDENT) {  // the whole packed gamma, once (landed long before the first epilogue: every stage waits vmcnt 0)
        for (int j = wave; j < G::G_ALL / 1024; j += NW)
            glds16((const char *)p.gp + (size_t)j * 1024 + lane * 16, smem + 2 * G::STAGE_BYTES + j * 1024);
    }
#ifdef CAE_EXP_SETPRIO
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    deconv_issue_f16<KS, CT, NW, PT, IGDN, 0>(p, in_n, plane_bytes, 0, smem, hrow, hbase, wave, lane);
    deconv_phase_f16<KS, CT, NW, PT, IGDN, 0>(p, in_n, plane_bytes, smem, sc, hrow, hbase, wave, lane, b_off, stray_mask, n,
                                              iy, ix);
    deconv_phase_f16<KS, CT, NW, PT, IGDN, 1>(p, in_n, plane_bytes, smem, sc, hrow, hbase, wave, lane, b_off, stray_mask, n,
                                              iy, ix);
}

// =================================================================================================
// deconv_last_f16_kernel: last synthesis layer (Cout <= 4, no IGDN) on v_mfma_f32_16x16x32_f16, C8SP in.
//   D rows = 4c + p (p = 2 py + px), cols = 16 input pixels; K = (neighbour, cin); one MFMA k-step
//   = 32 channels = 4 planes (lane group g = lane>>4 supplies plane 4q+g).
//   packed weights: [nd][ndx][q][hl][lane][8]: A(row = lane&15, cin = 32q + 8(lane>>4) + j)
//   PERSISTENT: the layer is a 134 MB-per-tile read with ~1 FLOP per byte, i.e. bound by how many bytes are
//   in flight.  One block per CU walks over (NW rows x 32 columns) input tiles with all weights resident and
//   a DEPTH-deep ring of halo stages [g][hl][row][x] (16-byte pieces) filled by LDS-DMA, so that DEPTH-1
//   stages (~40 KiB each) are always outstanding per CU; `s_waitcnt vmcnt(MAXP*(DEPTH-2))` retires exactly
//   the oldest stage (every wave issues MAXP pieces per stage; stores issued in between only make the wait
//   conservative).
// =================================================================================================
template <int KS, int NW_, int DEPTH_>
struct LastGeomF16 {
    static constexpr int NW = NW_, DEPTH = DEPTH_;
    static constexpr int P = KS / 2;
    static constexpr int DLO = -((P + 1) / 2), DHI = (KS - 1 - P) / 2;
    static constexpr int NB = DHI - DLO + 1;
    static constexpr int TXC = 32, NT = TXC / 16;
    static constexpr int WH = TXC + NB - 1, HR = NW + NB - 1;
    static constexpr int PLANE_PIECES = HR * WH;
    static constexpr int HALO_PIECES = 8 * PLANE_PIECES;  // [g(4)][hl(2)]
    static constexpr int MAXP = ((HALO_PIECES + 63) / 64 + NW - 1) / NW;  // LDS-DMA instructions per wave and stage
    static constexpr int STAGE_BYTES = MAXP * NW * 1024;
    static constexpr int lds_bytes(int nq) { return DEPTH * STAGE_BYTES + NB * NB * nq * 2048; }
};

template <int KS, int NW, int DEPTH>
__global__ void __launch_bounds__(NW * 64, 1) deconv_last_f16_kernel(const LayerArgs p) {
    using G = LastGeomF16<KS, NW, DEPTH>;
    constexpr int DLO = G::DLO, DHI = G::DHI, NB = G::NB, NT = G::NT, WH = G::WH, HR = G::HR;
    constexpr int PLANE_PIECES = G::PLANE_PIECES, MAXP = G::MAXP, STAGE_BYTES = G::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int nq = p.cci;  // 32-channel groups
    char *wbuf = smem + DEPTH * STAGE_BYTES;

    const int tiles_img = p.tiles_x * p.tiles_y;
    const int total = p.N * tiles_img;
    const size_t row_bytes = c8s_row_bytes<true>(p.W);  // input rows are C8SP
    const size_t plane_bytes = (size_t)p.H * row_bytes;

    // this thread's pieces of a stage: (plane-in-group/half, halo row, halo column); clamped duplicates past the end
    int pc_r[MAXP], pc_x[MAXP], pc_ghl[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < G::HALO_PIECES ? pc : G::HALO_PIECES - 1;
        pc_ghl[i] = pc / PLANE_PIECES;
        const int rem = pc - pc_ghl[i] * PLANE_PIECES;
        pc_r[i] = rem / WH;
        pc_x[i] = rem - pc_r[i] * WH;
    }
    // stage `step` of this block: tile = blockIdx.x + (step / nq) * gridDim.x, channel group q = step % nq
    auto issue = [&](int step) {
        const int it = step / nq, q = step - it * nq;
        const long t = (long)blockIdx.x + (long)it * gridDim.x;
        char *buf = smem + (step % DEPTH) * STAGE_BYTES;
        const bool live = t < total;
        const int tt = live ? (int)t : 0;
        const int n = tt / tiles_img, rem = tt - n * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const char *base = (const char *)p.in + ((size_t)n * p.in_planes + 4 * q) * plane_bytes;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int sy = ty * NW - DHI + pc_r[i], sx = tx * G::TXC - DHI + pc_x[i];
            const bool ok = live && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
            const char *src = base + (size_t)(pc_ghl[i] >> 1) * plane_bytes + (size_t)sy * row_bytes +
                              c8s_piece<true>(sx) + (pc_ghl[i] & 1) * 512;
            glds16(ok ? (const void *)src : (const void *)p.zero, buf + (wave + i * NW) * 1024);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d);
    const int w_bytes = NB * NB * nq * 2048;
    for (int i = threadIdx.x; i < w_bytes / 16; i += NW * 64)
        *(f32x4 *)(wbuf + i * 16) = *(const f32x4 *)((const char *)p.wp + i * 16);
    const float bias = (p.bias && g < p.cout) ? p.bias[g] : 0.0f;

    f32x4 acc[NT];
    int step = 0;
    for (long t = blockIdx.x; t < total; t += gridDim.x) {
        const int n = (int)t / tiles_img, rem = (int)t - n * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = bias;
        for (int q = 0; q < nq; ++q, ++step) {
            // oldest outstanding stage (this one) has landed once at most DEPTH-2 younger stages remain in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXP * (DEPTH - 2)) : "memory");
            __syncthreads();  // ... for every wave's pieces; and slot (step-1) % DEPTH is free again
            issue(step + DEPTH - 1);
            const char *cur = smem + (step % DEPTH) * STAGE_BYTES;
#pragma unroll
            for (int nd = 0; nd < NB; ++nd)
#pragma unroll
                for (int ndx = 0; ndx < NB; ++ndx) {
                    const int hr = wave - (DLO + nd) + DHI;
                    const int hx = col - (DLO + ndx) + DHI;
                    const char *wq = wbuf + (((nd * NB + ndx) * nq + q) * 2) * 1024 + lane * 16;
                    const f16x8 ah = *(const f16x8 *)(wq);
                    const f16x8 al = *(const f16x8 *)(wq + 1024);
                    const char *hb = cur + (((2 * g) * HR + hr) * WH + hx) * 16;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const f16x8 bh = *(const f16x8 *)(hb + tt * 256);
                        const f16x8 bl = *(const f16x8 *)(hb + tt * 256 + PLANE_PIECES * 16);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[tt], 0, 0, 0);
                    }
                }
        }
        const int iy = ty * NW + wave;
        if (iy < p.H && g < p.cout) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int ix = tx * G::TXC + 16 * tt + col;
                if (ix < p.W) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oy = 2 * iy + (r >> 1), ox = 2 * ix + (r & 1);
                        if (p.outfmt == OUT_U8HWC) {
                            ((uint8_t *)p.out)[(((size_t)n * p.OH + oy) * p.OW + ox) * p.cout + g] = clip_u8(acc[tt][r] * 255.0f);
                        } else {
                            ((float *)p.out)[(((size_t)n * p.cout + g) * p.OH + oy) * p.OW + ox] = acc[tt][r];
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the ring's look-ahead before the LDS is released
}

// fp32 NCHW -> split rows (module boundary / latents into the synthesis track); SP: C8SP rows, else C8S
template <bool SP>
__global__ void nchw_to_c8s_kernel(const float *in, char *out, int N, int C, int H, int W, int planes, int *flag,
                                   const int32_t *sym = nullptr, const float *medians = nullptr) {
    const size_t HW = (size_t)H * W, total = (size_t)N * planes * HW;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % HW;
        const size_t np = i / HW;
        const int plane = (int)(np % planes);
        const size_t n = np / planes;
        f16x8 vh, vl;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = plane * 8 + k;
            // sym != nullptr: fused dequantiser (float(sym) + median_c)
            const float v = c < C ? (sym ? (float)sym[(n * C + c) * HW + pix] + medians[c] : in[(n * C + c) * HW + pix]) : 0.0f;
            bad |= !(__builtin_fabsf(v) <= F16_MAX);
            _Float16 a, b;
            split_f16(v, a, b);
            vh[k] = a;
            vl[k] = b;
        }
        char *dst = out + (np * H + pix / W) * c8s_row_bytes<SP>(W) + c8s_piece<SP>((int)(pix % W));
        *(f16x8 *)dst = vh;
        *(f16x8 *)(dst + 512) = vl;
    }
    if (bad) *flag = 1;
}

// split rows -> fp32 NCHW (bridges)
template <bool SP>
__global__ void c8s_to_nchw_kernel(const char *in, float *out, int N, int C, int H, int W, int planes) {
    const size_t HW = (size_t)H * W, total = (size_t)N * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % HW;
        const size_t nc = i / HW;
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        const char *src = in + ((n * planes + (c >> 3)) * H + pix / W) * c8s_row_bytes<SP>(W) +
                          c8s_piece<SP>((int)(pix % W));
        out[i] = (float)((const _Float16 *)src)[c & 7] + (float)((const _Float16 *)(src + 512))[c & 7];
    }
}

// fp32 C8 [rows][W][8] -> C8S rows (more than 4 input channels given as uint8: rare); SP: C8SP rows (synthesis track)
template <bool SP = false>
static __global__ void c8_to_c8s_kernel(const float *in, char *out, size_t rows, int W, int *flag) {
    const size_t npix = rows * W;
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(in + i * 8), b = *(const f32x4 *)(in + i * 8 + 4);
        f16x8 vh, vl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mx = absmax3(mx, a[k], b[k]);
            _Float16 x, y;
            split_f16(a[k], x, y);
            vh[k] = x;
            vl[k] = y;
            split_f16(b[k], x, y);
            vh[4 + k] = x;
            vl[4 + k] = y;
        }
        char *dst = out + (i / W) * c8s_row_bytes<SP>(W) + c8s_piece<SP>((int)(i % W));
        *(f16x8 *)dst = vh;
        *(f16x8 *)(dst + 512) = vl;
    }
    raise_if_over(mx, flag);
}

// C8S rows -> fp32 C8 [rows][W][8] (a layer the f16x3 kernels do not cover runs on the fp32 kernel)
template <bool SP = false>
static __global__ void c8s_to_c8_kernel(const char *in, float *out, size_t rows, int W) {
    const size_t npix = rows * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const char *src = in + (i / W) * c8s_row_bytes<SP>(W) + c8s_piece<SP>((int)(i % W));
        const f16x8 vh = *(const f16x8 *)src, vl = *(const f16x8 *)(src + 512);
        f32x4 a, b;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = (float)vh[k] + (float)vl[k];
            b[k] = (float)vh[4 + k] + (float)vl[4 + k];
        }
        *(f32x4 *)(out + i * 8) = a;
        *(f32x4 *)(out + i * 8 + 4) = b;
    }
}

}  // namespace cae
