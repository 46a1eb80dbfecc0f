// Split-precision ("f16x3") variants of the conv / deconv / GDN kernels for CDNA4.
//
// Every fp32 value v is carried as two halves  v = hi + lo,  hi = f16(v), lo = f16(v - hi)
// (22 significant bits).  A product is three f16 MFMAs accumulated in fp32:
//     a*b ~= ah*bh + ah*bl + al*bh            (the dropped al*bl term is < 2^-22 |a b|)
// On v_mfma_f32_32x32x16_f16 this costs 3 x 32 cycles per 32x32x16 block against 8 x 64 cycles on
// v_mfma_f32_32x32x2_f32: 5.3x less matrix-pipe time at fp32-class accuracy (measured through the
// whole analysis stack: max |err| 1.5e-6 vs 1.1e-6 for plain fp32, both against fp64).
//
// HBM layout "C8S": activations are [N][P][H][W] records of 32 B = [8 x f16 hi][8 x f16 lo] for
// the 8 channels of plane P.  Same bytes as fp32 C8, produced once in the epilogue of the
// previous layer, so consumers never convert.  A 16-channel MFMA k-step = two planes; lane half h
// of the wave supplies the 8 channels of plane 2q+h.
//
// Block = 4 waves, ONE block per CU (each wave may use the whole 512-entry register file):
//   wave tile = all CT*32 output channels x 64 pixels (two 32-pixel MFMA column tiles), so every
//   A (weight) fragment read from LDS feeds 6 MFMAs and LDS traffic stays ~60 B/clk/CU.
#pragma once
#include "cae_kernels.hpp"

namespace cae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
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
template <int CT, int PT, int NW, bool INVERSE, int STAGE_BYTES, class Tail>
__device__ __forceinline__ void gdn_stages_f16(f32x16 (&y)[PT][CT], const LayerArgs &p, char *smem, int &sc,
                                               int wave, int lane, Tail tail) {
    constexpr int G_BYTES = CT * 4096;  // per jt: CT co x 2 s x 2 hl x 1 KiB
    const int h = lane >> 5;
    f32x16 nrm[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) init_acc<CT>(nrm[pt], p.beta, h, 1.0f);
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
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = y[pt][jt][8 * s + e];
                    _Float16 a, b;
                    split_f16(v * v, a, b);
                    sh[pt][e] = a;
                    sl[pt][e] = b;
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
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float nv = nrm[pt][ct][r];
                y[pt][ct][r] *= INVERSE ? __builtin_amdgcn_sqrtf(nv) : __builtin_amdgcn_rsqf(nv);
            }
}

// store CT accumulator tiles of one pixel column-tile: C8S (split halves) | NCHW fp32 | HWC uint8
template <int CT>
__device__ __forceinline__ void store_tiles_f16(const f32x16 (&acc)[CT], const LayerArgs &p, int n, int oy, int ox,
                                                int h, bool valid) {
    if (!valid) return;
    if (p.outfmt == OUT_C8) {
        char *out = (char *)p.out;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int plane = 4 * ct + g;
                f16x4 vh, vl;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    _Float16 a, b;
                    split_f16(acc[ct][4 * g + k], a, b);
                    vh[k] = a;
                    vl[k] = b;
                }
                char *dst = out + ((((size_t)n * p.out_planes + plane) * p.OH + oy) * p.OW + ox) * 32 + 8 * h;
                *(f16x4 *)dst = vh;
                *(f16x4 *)(dst + 16) = vl;
            }
    } else {
        store_tiles<CT>(acc, p, n, oy, ox, h, valid);
    }
}

// =================================================================================================
// conv_s2_f16_kernel: strided reflect conv (+bias) (+GDN), f16x3, C8S in.
//   block tile = 16 x 16 output pixels (wave w: rows 4w..4w+3; column tile pt: rows 4w+2pt, +1)
//   stage = (16-channel chunk q, kernel row ky):
//     weights [kx][ct][hl][64][8 f16]  (KS*CT*2 KiB)   +   halo [pl][hl][16 rows][WH][16 B]
// =================================================================================================
template <int KS, int CT, bool GDN>
__global__ void __launch_bounds__(256, 1) conv_s2_f16_kernel(const LayerArgs p) {
    constexpr int NW = 4, PT = 2;
    constexpr int PAD = KS / 2;
    constexpr int TX = 16, TY = 16;
    constexpr int WH = 2 * TX + KS - 2;
    constexpr int PLANE_PIECES = TY * WH;          // 16-byte pieces per (plane, half)
    constexpr int HALO_PIECES = 4 * PLANE_PIECES;  // [pl][hl][row][x]
    constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;
    constexpr int W_INSTR = KS * CT * 2;
    constexpr int W_BYTES = W_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int CONV_STAGE = W_BYTES + HALO_INSTR * 1024;
    constexpr int STAGE_BYTES = CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES;
    constexpr int MAXP = (HALO_INSTR + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX;

    const size_t plane_bytes = (size_t)p.H * p.W * 32;
    const char *in_n = (const char *)p.in + (size_t)n * p.in_planes * plane_bytes;
    unsigned hoff[MAXP][KS];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int pc = (wave + i * NW) * 64 + lane;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        const int plhl = pc / PLANE_PIECES;
        const int rem = pc - plhl * PLANE_PIECES;
        const int r = rem / WH, x = rem - r * WH;
        const unsigned base = (unsigned)(plhl >> 1) * (unsigned)plane_bytes + (unsigned)(plhl & 1) * 16u +
                              (unsigned)reflect_idx(2 * ox0 - PAD + x, p.W) * 32u;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
            hoff[i][ky] = base + (unsigned)reflect_idx(2 * (oy0 + r) - PAD + ky, p.H) * (unsigned)p.W * 32u;
    }
    const unsigned woff = (unsigned)lane * 16u;

    auto issue_stage = [&](int q, auto ky_tag, char *buf) {
        constexpr int ky = decltype(ky_tag)::value;
        const char *wsrc = (const char *)p.wp + (size_t)(q * KS + ky) * W_BYTES;
#pragma unroll
        for (int i = 0; i < (W_INSTR + NW - 1) / NW; ++i) {
            const int j = wave + i * NW;
            if (j < W_INSTR) glds16(wsrc + j * 1024 + woff, buf + j * 1024);
        }
        const char *planes = in_n + (size_t)(2 * q) * plane_bytes;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + i * NW;
            if (j < HALO_INSTR) glds16(planes + hoff[i][ky], buf + W_BYTES + j * 1024);
        }
    };

    f32x16 acc[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) init_acc<CT>(acc[pt], p.bias, h, 0.0f);

    // B operand of (column tile pt, tap kx): halo [pl = h][hl][row 4w + 2pt + (m>>4)][2(m&15) + kx]
    const int b_off = W_BYTES + (((2 * h) * TY + 4 * wave + (m >> 4)) * WH + 2 * (m & 15)) * 16;
    constexpr int B_HL = PLANE_PIECES * 16;  // hi -> lo
    constexpr int B_PT = 2 * WH * 16;        // column tile 0 -> 1 (two rows down)
    int sc = 0;

    issue_stage(0, std::integral_constant<int, 0>{}, smem);
    for (int q = 0; q < p.cci; ++q) {
        static_for<KS>([&](auto ky_tag) {
            constexpr int ky = decltype(ky_tag)::value;
            wait_vm0();
            __syncthreads();
            char *cur = smem + (sc & 1) * STAGE_BYTES;
            char *nxt = smem + ((sc + 1) & 1) * STAGE_BYTES;
            if constexpr (ky + 1 < KS) {
                issue_stage(q, std::integral_constant<int, ky + 1>{}, nxt);
            } else {
                if (q + 1 < p.cci) {
                    issue_stage(q + 1, std::integral_constant<int, 0>{}, nxt);
                } else if (GDN) {
                    issue_gamma0<CT, NW>(p, nxt, wave, lane);
                }
            }
            const char *wb = cur + lane * 16;
            const char *hb = cur + b_off;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f16x8 bh[PT], bl[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    bh[pt] = *(const f16x8 *)(hb + pt * B_PT + kx * 16);
                    bl[pt] = *(const f16x8 *)(hb + pt * B_PT + kx * 16 + B_HL);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f16x8 ah = *(const f16x8 *)(wb + ((kx * CT + ct) * 2 + 0) * 1024);
                    const f16x8 al = *(const f16x8 *)(wb + ((kx * CT + ct) * 2 + 1) * 1024);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[pt][ct] = mfma3(ah, al, bh[pt], bl[pt], acc[pt][ct]);
                }
            }
            ++sc;
        });
    }

    if constexpr (GDN) {
        gdn_stages_f16<CT, PT, NW, false, STAGE_BYTES>(acc, p, smem, sc, wave, lane, [](char *) {});
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + 4 * wave + 2 * pt + (m >> 4), ox = ox0 + (m & 15);
        store_tiles_f16<CT>(acc[pt], p, n, oy, ox, h, oy < p.OH && ox < p.OW);
    }
}

// ---- layout conversions for the split format ------------------------------------------------------
// fp32 C8 [..][8] -> C8S record [8 hi][8 lo]
__global__ void c8_to_c8s_kernel(const float *in, char *out, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(in + i * 8), b = *(const f32x4 *)(in + i * 8 + 4);
        f16x8 vh, vl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            _Float16 x, y;
            split_f16(a[k], x, y);
            vh[k] = x;
            vl[k] = y;
            split_f16(b[k], x, y);
            vh[4 + k] = x;
            vl[4 + k] = y;
        }
        *(f16x8 *)(out + i * 32) = vh;
        *(f16x8 *)(out + i * 32 + 16) = vl;
    }
}

__global__ void c8s_to_c8_kernel(const char *in, float *out, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const f16x8 vh = *(const f16x8 *)(in + i * 32), vl = *(const f16x8 *)(in + i * 32 + 16);
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
