// C ABI of the training path (include/cae_hip.h, "training"): stateless launches on caller-owned device buffers.
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_train_kernels.hpp"
#include "cae_train_gdn.hpp"

#include <algorithm>
#include <cstdlib>

using namespace cae;
using namespace cae::tr;

namespace {

const void *zero_page() {
    static void *z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 1024) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 1024);
    }
    return z;
}

unsigned ew_grid(size_t total) {
    const size_t b = (total + 255) / 256;
    return (unsigned)std::min<size_t>(std::max<size_t>(b, 1), 256 * 8 * 4);
}

bool bad_channels(int c) { return c < 32 || c % 32 != 0 || c > 192; }

template <int NT, bool PIPE>
int launch_gg_tp(const GGArgs &a, size_t lds, hipStream_t st) {
    auto kern = gather_gemm_kernel<NT, PIPE>;
    static size_t attr = 0;
    if (lds > attr) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

// gg8_kernel (two waves per SIMD, compile-time taps, blocks walking several samples): the shapes of the canonical model
template <int NT, int NQ, int NTAPS>
int launch_gg8(GGArgs &a, hipStream_t st) {
    auto kern = gg8_kernel<NT, NQ, NTAPS>;
    const size_t h_instr = (size_t)(NQ * a.HR * a.HC + 63) / 64;
    const size_t lds = 2 * (h_instr + (size_t)NTAPS * NT * (NQ / 2)) * 1024;
    static size_t attr = 0;
    if (lds > attr) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    // samples per block: enough blocks for two rounds over the 256 CUs, the rest of the batch amortises a block's prologue
    const size_t tiles = (size_t)a.tiles_x * a.tiles_y;
    const char *enpb = std::getenv("CAE_GG8_NPB");  // (read per call: tests and A/B runs switch it)
    const int forced = enpb ? std::atoi(enpb) : 0;
    a.npb = forced > 0 ? std::min(forced, a.N) : (int)std::min<size_t>(std::max<size_t>((size_t)a.N * tiles / 512, 1), (size_t)a.N);
    const unsigned grid = (unsigned)(tiles * (size_t)((a.N + a.npb - 1) / a.npb));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

// -> true when a gg8 instantiation covers the launch (rc holds its result)
bool try_gg8(GGArgs &a, int NT, hipStream_t st, int &rc) {
    static const char *e8 = std::getenv("CAE_GG8");
    static const bool off = std::getenv("CAE_GG_LEGACY") != nullptr || (e8 && e8[0] == '0');
    if (off) return false;
    const char *ewide = std::getenv("CAE_GG8_WIDE");  // "0": 192-channel outputs stay on gather_gemm_kernel (A/B)
    if (NT == 6 && !(ewide && ewide[0] == '0') && (a.ntaps == 9 || a.ntaps == 4 || a.ntaps == 2 || a.ntaps == 1)) {
        // 192 output channels: two launches of three n-tiles (six do not leave room for two slice buffers in the LDS)
        int nq = 0;
        for (int q : {4, 2}) {
            const size_t slice = (size_t)((q * a.HR * a.HC + 63) / 64) * 1024 + (size_t)a.ntaps * 3 * (q / 2) * 1024;
            if (2 * slice <= 160 * 1024 && (q * a.HR * a.HC + 63) / 64 <= 80) {
                nq = q;
                break;
            }
        }
        if ((a.ntaps == 9 && nq != 2) || (a.ntaps != 9 && nq != 4)) return false;
        for (int part = 0; part < 2; ++part) {
            GGArgs b = a;
            b.nq = nq;
            b.nt0 = 3 * part;
            b.nt_all = 6;
            rc = a.ntaps == 9 ? launch_gg8<3, 2, 9>(b, st)
                 : a.ntaps == 4 ? launch_gg8<3, 4, 4>(b, st)
                 : a.ntaps == 2 ? launch_gg8<3, 4, 2>(b, st) : launch_gg8<3, 4, 1>(b, st);
            if (rc) return true;
        }
        return true;
    }
    if (a.nq == 0 || (a.nq * a.HR * a.HC + 63) / 64 > 80) return false;
    if (NT == 4 && a.nq == 2 && a.ntaps == 9) { rc = launch_gg8<4, 2, 9>(a, st); return true; }
    if (NT == 4 && a.nq == 4 && a.ntaps == 4) { rc = launch_gg8<4, 4, 4>(a, st); return true; }
    if (NT == 4 && a.nq == 4 && a.ntaps == 2) { rc = launch_gg8<4, 4, 2>(a, st); return true; }
    if (NT == 4 && a.nq == 4 && a.ntaps == 1) { rc = launch_gg8<4, 4, 1>(a, st); return true; }
    if (NT == 1 && a.nq == 4 && a.ntaps == 1) { rc = launch_gg8<1, 4, 1>(a, st); return true; }
    return false;
}

template <int NT>
int launch_gg_t(const GGArgs &a, size_t lds, hipStream_t st) {
    return a.nq ? launch_gg_tp<NT, true>(a, lds, st) : launch_gg_tp<NT, false>(a, lds, st);
}

// fills the halo / staging geometry of `a` from its tap list and launches
int launch_gg(GGArgs &a, hipStream_t st) {
    if (bad_channels(a.Ck) || bad_channels(a.Cn)) return fail(CAE_ERR_ARG, "channel counts must be multiples of 32, at most 192");
    if (a.ntaps < 1 || a.ntaps > MAX_TAPS) return fail(CAE_ERR_ARG, "bad tap count");
    int dymin = a.dy[0], dymax = a.dy[0], dxmin = a.dx[0], dxmax = a.dx[0];
    for (int t = 1; t < a.ntaps; ++t) {
        dymin = std::min<int>(dymin, a.dy[t]);
        dymax = std::max<int>(dymax, a.dy[t]);
        dxmin = std::min<int>(dxmin, a.dx[t]);
        dxmax = std::max<int>(dxmax, a.dx[t]);
    }
    a.dymin = dymin;
    a.dxmin = dxmin;
    a.HR = a.S * 15 + (dymax - dymin) + 1;
    a.HC = a.S * 15 + (dxmax - dxmin) + 1;
    const int NT = a.Cn / 32;
    const int pieces = 4 * a.HR * a.HC;
    if ((pieces + 63) / 64 > 80) return fail(CAE_ERR_UNSUPPORTED, "halo too large");
    const size_t halo = (size_t)((pieces + 63) / 64) * 1024;
    const size_t budget = 160 * 1024 - halo;
    a.taps_per_stage = std::min<int>(a.ntaps, (int)(budget / ((size_t)NT * 2048)));
    if (a.taps_per_stage < 1) return fail(CAE_ERR_UNSUPPORTED, "weights of one tap do not fit the LDS");
    size_t lds = halo + (size_t)a.taps_per_stage * NT * 2048;
    // pipelined form: a slice (32 or 16 channels: halo + the weights of all taps) twice in the LDS
    a.nq = 0;
    for (int nq : {4, 2}) {
        const size_t slice = (size_t)((nq * a.HR * a.HC + 63) / 64) * 1024 + (size_t)a.ntaps * NT * (nq / 2) * 1024;
        if (2 * slice <= 160 * 1024 && !std::getenv("CAE_GG_LEGACY")) {
            a.nq = nq;
            lds = 2 * slice;
            break;
        }
    }
    a.m_plane = (unsigned)(((1ull << 32) + (unsigned)(a.HR * a.HC) - 1) / (unsigned)(a.HR * a.HC));
    a.m_hc = (unsigned)(((1ull << 32) + (unsigned)a.HC - 1) / (unsigned)a.HC);
    a.tiles_x = (a.LW + 15) / 16;
    a.tiles_y = (a.LH + 15) / 16;
    a.zero = zero_page();
    if (!a.zero) return fail(CAE_ERR_NOMEM, "zero page");
    int rc8 = CAE_OK;
    if (try_gg8(a, NT, st, rc8)) return rc8;
    switch (NT) {
        case 1: return launch_gg_t<1>(a, lds, st);
        case 2: return launch_gg_t<2>(a, lds, st);
        case 3: return launch_gg_t<3>(a, lds, st);
        case 4: return launch_gg_t<4>(a, lds, st);
        case 5: return launch_gg_t<5>(a, lds, st);
        default: return launch_gg_t<6>(a, lds, st);
    }
}

// strided correlation: position (i, j) <- input (2i + ky - P, 2j + kx - P), every tap
int strided_corr(const void *in16, int n, int ih, int iw, int ck, const void *packed, int ks, int reflect, float *out32,
                 void *out16, int cn, int oh, int ow, const float *bias, hipStream_t st, int act = 0) {
    if (ks != 3 && ks != 5) return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
    GGArgs a{};
    a.in = in16;
    a.out32 = out32;
    a.out16 = out16;
    a.wp = packed;
    a.bias = bias;
    a.N = n;
    a.IH = ih;
    a.IW = iw;
    a.Ck = ck;
    a.Cn = cn;
    a.OH = oh;
    a.OW = ow;
    a.LH = oh;
    a.LW = ow;
    a.S = 2;
    a.SO = 1;
    a.reflect = reflect;
    a.act = act;
    a.ktaps = ks * ks;
    a.ntaps = ks * ks;
    const int P = ks / 2;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int t = ky * ks + kx;
            a.dy[t] = (short)(ky - P);
            a.dx[t] = (short)(kx - P);
            a.wt[t] = (short)t;
        }
    return launch_gg(a, st);
}

// transpose of the strided correlation, one launch per output parity.
//   shift = P : cropped domain   out[Y] = sum in[(Y + P - ky) / 2]  (ConvTranspose2d(k, 2, k//2, output_padding 1))
//   shift = 0 : extended domain  out[Y] = sum in[(Y - ky) / 2],  Y in [0, 2 ih + k - 2]   (data gradient of a valid
//               convolution on the reflect-padded input; Y = y + P)
// all four output parities of the k = 3 transpose in one launch (gg8t_kernel); -> false when the shape is not covered
template <bool EXT>
int launch_gg8t(GGArgs &a, hipStream_t st) {
    auto kern = gg8t_kernel<EXT>;
    const size_t h_instr = (size_t)(4 * a.HR * a.HC + 63) / 64;
    const size_t lds = 2 * (h_instr + 9 * 2 * 2) * 1024;
    static size_t attr = 0;
    if (lds > attr) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    const size_t tiles = (size_t)a.tiles_x * a.tiles_y, halves = (size_t)a.Cn / 64;
    a.npb = (int)std::min<size_t>(std::max<size_t>((size_t)a.N * tiles * halves / 512, 1), (size_t)a.N);
    const unsigned grid = (unsigned)(tiles * (size_t)((a.N + a.npb - 1) / a.npb));
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)halves), dim3(512), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

bool try_gg8t(const void *in16, int n, int ih, int iw, int ck, const void *packed, int shift, float *out32, void *out16, int cn,
              int oh, int ow, const float *bias, hipStream_t st, int act, int &rc) {
    const char *e = std::getenv("CAE_GG8T");
    if ((e && e[0] == '0') || std::getenv("CAE_GG_LEGACY") || cn % 64 || bad_channels(ck) || bad_channels(cn)) return false;
    GGArgs a{};
    a.in = in16;
    a.out32 = out32;
    a.out16 = out16;
    a.wp = packed;
    a.bias = bias;
    a.N = n;
    a.IH = ih;
    a.IW = iw;
    a.Ck = ck;
    a.Cn = cn;
    a.OH = oh;
    a.OW = ow;
    a.LH = (oh + 1) / 2;
    a.LW = (ow + 1) / 2;
    a.S = 1;
    a.SO = 2;
    a.act = act;
    a.ktaps = 9;
    a.ntaps = 9;
    int nt = 0, dymin = 1 << 20, dymax = -(1 << 20), dxmin = 1 << 20, dxmax = -(1 << 20);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)  // parity-major: (0,0), (0,1), (1,0), (1,1) = gg8t_parity's order
            for (int ky = 0; ky < 3; ++ky) {
                if (((py + shift - ky) & 1) != 0) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    if (((px + shift - kx) & 1) != 0) continue;
                    a.dy[nt] = (short)((py + shift - ky) / 2);
                    a.dx[nt] = (short)((px + shift - kx) / 2);
                    a.wt[nt] = (short)(ky * 3 + kx);
                    dymin = std::min<int>(dymin, a.dy[nt]);
                    dymax = std::max<int>(dymax, a.dy[nt]);
                    dxmin = std::min<int>(dxmin, a.dx[nt]);
                    dxmax = std::max<int>(dxmax, a.dx[nt]);
                    ++nt;
                }
            }
    if (nt != 9) return false;
    a.dymin = dymin;
    a.dxmin = dxmin;
    a.HR = 15 + (dymax - dymin) + 1;
    a.HC = 15 + (dxmax - dxmin) + 1;
    if ((4 * a.HR * a.HC + 63) / 64 > 32) return false;
    a.m_hc = (unsigned)(((1ull << 32) + (unsigned)a.HC - 1) / (unsigned)a.HC);
    a.tiles_x = (a.LW + 15) / 16;
    a.tiles_y = (a.LH + 15) / 16;
    a.zero = zero_page();
    if (!a.zero) return false;
    rc = shift == 0 ? launch_gg8t<true>(a, st) : launch_gg8t<false>(a, st);
    return true;
}

int strided_corr_t(const void *in16, int n, int ih, int iw, int ck, const void *packed, int ks, int shift, float *out32,
                   void *out16, int cn, int oh, int ow, const float *bias, hipStream_t st, int act = 0) {
    if (ks != 3 && ks != 5) return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
    int rc8t = CAE_OK;
    if (ks == 3 && (shift == 0 || shift == 1) &&
        try_gg8t(in16, n, ih, iw, ck, packed, shift, out32, out16, cn, oh, ow, bias, st, act, rc8t))
        return rc8t;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            GGArgs a{};
            a.in = in16;
            a.out32 = out32;
            a.out16 = out16;
            a.wp = packed;
            a.bias = bias;
            a.N = n;
            a.IH = ih;
            a.IW = iw;
            a.Ck = ck;
            a.Cn = cn;
            a.OH = oh;
            a.OW = ow;
            a.LH = (oh - py + 1) / 2;
            a.LW = (ow - px + 1) / 2;
            if (a.LH < 1 || a.LW < 1) continue;
            a.S = 1;
            a.SO = 2;
            a.oy0 = py;
            a.ox0 = px;
            a.reflect = 0;
            a.act = act;
            a.ktaps = ks * ks;
            int nt = 0;
            for (int ky = 0; ky < ks; ++ky) {
                if (((py + shift - ky) & 1) != 0) continue;
                for (int kx = 0; kx < ks; ++kx) {
                    if (((px + shift - kx) & 1) != 0) continue;
                    a.dy[nt] = (short)((py + shift - ky) / 2);  // exact: the numerator is even
                    a.dx[nt] = (short)((px + shift - kx) / 2);
                    a.wt[nt] = (short)(ky * ks + kx);
                    ++nt;
                }
            }
            a.ntaps = nt;
            if (nt == 0) continue;
            int rc = launch_gg(a, st);
            if (rc) return rc;
        }
    return CAE_OK;
}

// stride-1 correlations of the LeakyReLU / ReLU units' pre-convolutions: position (i, j) <- input (i + d_ky, j + d_kx)
//   mode 0  analysis pre-convolution forward      d = k - P, reflect padding                 (Conv2d(cin, cin, k, 1, k//2, reflect))
//   mode 1  its data gradient, EXTENDED domain    position Y = y + P <- g[Y - k], zeros      (folded by the consumer)
//   mode 2  synthesis pre-convolution forward     d = P - k, zeros                           (ConvTranspose2d(cin, cin, k, 1, k//2))
//   mode 3  its data gradient                     d = k - P, zeros
int stride1_corr(const void *in16, int n, int ih, int iw, int ck, const void *packed, int ks, int mode, float *out32,
                 void *out16, int cn, const float *bias, int act, hipStream_t st) {
    if (ks != 3 && ks != 5) return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
    if (mode < 0 || mode > 3) return fail(CAE_ERR_ARG, "bad mode %d", mode);
    const int P = ks / 2;
    GGArgs a{};
    a.in = in16;
    a.out32 = out32;
    a.out16 = out16;
    a.wp = packed;
    a.bias = bias;
    a.N = n;
    a.IH = ih;
    a.IW = iw;
    a.Ck = ck;
    a.Cn = cn;
    a.OH = mode == 1 ? ih + 2 * P : ih;
    a.OW = mode == 1 ? iw + 2 * P : iw;
    a.LH = a.OH;
    a.LW = a.OW;
    a.S = 1;
    a.SO = 1;
    a.reflect = mode == 0;
    a.act = act;
    a.ktaps = ks * ks;
    a.ntaps = ks * ks;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int t = ky * ks + kx;
            a.dy[t] = (short)(mode == 1 ? -ky : (mode == 2 ? P - ky : ky - P));
            a.dx[t] = (short)(mode == 1 ? -kx : (mode == 2 ? P - kx : kx - P));
            a.wt[t] = (short)t;
        }
    return launch_gg(a, st);
}

template <int NB>
int launch_wg_t(const WGArgs &a, size_t lds, int ksplit, int at, int tg, hipStream_t st) {
    auto kern = wgrad_kernel<NB>;
    static size_t attr = 0;
    if (lds > attr) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(kern, dim3(ksplit, at, tg), dim3(256), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

template <int CT, int MODE>
int launch_gdn_a_t(const GdnArgs &a, hipStream_t st) {
    auto kern = gdn_gemm_a_kernel<CT, MODE>;
    constexpr int LDS = CT * 32 * (CT * 32 + 4) * 4 + (CT <= 4 ? 2 * 32768 : 0);  // M (+ the A double buffer)
    static bool done = false;
    if (!done) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        done = true;
    }
    const long tiles = (a.pixels + 255) / 256;
    const unsigned grid = (unsigned)std::min<long>(tiles, 512);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

template <int MODE>
int launch_gdn_a(const GdnArgs &a, hipStream_t st) {
    switch (a.C / 32) {
        case 1: return launch_gdn_a_t<1, MODE>(a, st);
        case 2: return launch_gdn_a_t<2, MODE>(a, st);
        case 3: return launch_gdn_a_t<3, MODE>(a, st);
        case 4: return launch_gdn_a_t<4, MODE>(a, st);
        case 5: return launch_gdn_a_t<5, MODE>(a, st);
        default: return launch_gdn_a_t<6, MODE>(a, st);
    }
}

template <int CT>
int launch_gdn_b_t(const float *gn, const float *z, long pixels, float *gg, float *gb, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<long>(std::max<long>(pixels / 256, 1), 256);
    hipLaunchKernelGGL(gdn_gemm_b_kernel<CT>, dim3(grid), dim3(CT * 64), 0, st, gn, z, pixels, gg, gb);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

template <int CT>
int launch_gdn_fused_t(const GdnFusedArgs &a, bool backward, hipStream_t st) {
    constexpr int C = CT * 32;
    constexpr int LDS_F = 2 * 32 * C * 4 + 32 * (C * 2 + 16);  // (Gamma in registers)
    constexpr int LDS_B = C * (C + 4) * 4 + 4 * 32 * C * 4 + 32 * (C * 2 + 16);
    auto kf = gdn_fwd_fused_kernel<CT>;
    auto kb = gdn_bwd_fused_kernel<CT>;
    static bool done = false;
    if (!done) {
        HIP_TRY(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F));
        HIP_TRY(hipFuncSetAttribute((const void *)kb, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        done = true;
    }
    const long tiles = (a.pixels + 31) / 32;
    // persistent blocks walk the tiles: one per CU (backward: Gamma + two tile sets fill the LDS), two per CU (forward)
    const unsigned grid = (unsigned)std::min<long>(tiles, backward ? 256 : 512);
    if (backward)
        hipLaunchKernelGGL(kb, dim3(grid), dim3(CT * 64), LDS_B, st, a);
    else
        hipLaunchKernelGGL(kf, dim3(grid), dim3(CT * 64), LDS_F, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int launch_gdn_fused(const GdnFusedArgs &a, int cp, bool backward, hipStream_t st) {
    switch (cp / 32) {
        case 1: return launch_gdn_fused_t<1>(a, backward, st);
        case 2: return launch_gdn_fused_t<2>(a, backward, st);
        case 3: return launch_gdn_fused_t<3>(a, backward, st);
        case 4: return launch_gdn_fused_t<4>(a, backward, st);
        default: return fail(CAE_ERR_UNSUPPORTED, "fused GDN kernels are built for at most 128 channels, got %d", cp);
    }
}

}  // namespace

static int wgrad_impl(const void *xbig16, int n, int h, int w, int ca, const void *ysmall16, int oh, int ow, int cb, int ks,
                      int reflect, int S, float *gw32, void *stream);

extern "C" {

size_t cae_t_packed_bytes(int contract_channels, int out_channels, int kernel_size) {
    const size_t q = (contract_channels + 31) / 32, nt = (out_channels + 31) / 32;
    return q * kernel_size * kernel_size * nt * 2 * 512 * 2;
}

int cae_t_pack_weights(const float *w, int dim0, int dim1, int ks, int contract_dim, void *packed, void *stream) {
    if (!w || !packed) return fail(CAE_ERR_ARG, "NULL argument");
    if (dim0 < 1 || dim1 < 1 || (contract_dim != 0 && contract_dim != 1)) return fail(CAE_ERR_ARG, "bad weight shape");
    const int kk = ks * ks;
    const int Kc = contract_dim == 0 ? dim0 : dim1, Nc = contract_dim == 0 ? dim1 : dim0;
    const long s0 = (long)dim1 * kk, s1 = kk;  // element strides of dim0 / dim1
    const long sk = contract_dim == 0 ? s0 : s1, sn = contract_dim == 0 ? s1 : s0;
    const int kchunks = (Kc + 31) / 32, NT = (Nc + 31) / 32;
    const size_t total = (size_t)kchunks * kk * NT * 1024;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, (__bf16 *)packed, Kc,
                       Nc, kk, sk, sn, kchunks, NT);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_from_nchw(const float *x, int n, int c, int h, int w, int cp, void *out16, float *out32, void *stream) {
    if (!x || (!out16 && !out32)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || h < 1 || w < 1 || cp < c || cp % 32) return fail(CAE_ERR_ARG, "bad shape");
    hipLaunchKernelGGL(nchw_to_t_kernel, dim3(ew_grid((size_t)n * h * w * cp)), dim3(256), 0, (hipStream_t)stream, x,
                       (__bf16 *)out16, out32, n, c, h, w, cp);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_to_nchw(const float *t32, int n, int c, int h, int w, int cp, float *out, void *stream) {
    if (!t32 || !out) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || h < 1 || w < 1 || cp < c || cp % 32) return fail(CAE_ERR_ARG, "bad shape");
    hipLaunchKernelGGL(t_to_nchw_kernel, dim3(ew_grid((size_t)n * c * h * w)), dim3(256), 0, (hipStream_t)stream, t32, out, n,
                       c, h, w, cp);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_conv_forward(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                       int cout_p, const float *bias, void *stream) {
    if (!x16 || !packed || (!z32 && !z16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 2 || w < 2) return fail(CAE_ERR_ARG, "bad shape");
    return strided_corr(x16, n, h, w, cin_p, packed, ks, 1, z32, z16, cout_p, (h + 1) / 2, (w + 1) / 2, bias,
                        (hipStream_t)stream);
}

int cae_t_conv_forward_act(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                           int cout_p, const float *bias, int act, void *stream) {
    if (!x16 || !packed || (!z32 && !z16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 2 || w < 2 || act < 0 || act > 2) return fail(CAE_ERR_ARG, "bad shape or activation");
    return strided_corr(x16, n, h, w, cin_p, packed, ks, 1, z32, z16, cout_p, (h + 1) / 2, (w + 1) / 2, bias,
                        (hipStream_t)stream, act);
}

int cae_t_deconv_forward_act(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                             int cout_p, const float *bias, int act, void *stream) {
    if (!x16 || !packed || (!z32 && !z16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || act < 0 || act > 2) return fail(CAE_ERR_ARG, "bad shape or activation");
    return strided_corr_t(x16, n, h, w, cin_p, packed, ks, ks / 2, z32, z16, cout_p, 2 * h, 2 * w, bias, (hipStream_t)stream,
                          act);
}

int cae_t_corr_s1(const void *x16, int n, int h, int w, int ck, const void *packed, int ks, int mode, float *out32, void *out16,
                  int cn, const float *bias, int act, void *stream) {
    if (!x16 || !packed || (!out32 && !out16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || act < 0 || act > 2) return fail(CAE_ERR_ARG, "bad shape or activation");
    return stride1_corr(x16, n, h, w, ck, packed, ks, mode, out32, out16, cn, bias, act, (hipStream_t)stream);
}

int cae_t_pointwise(const void *x16, int n, int h, int w, int ck, const void *packed, float *out32, void *out16, int cn,
                    const float *bias, int act, void *stream) {
    if (!x16 || !packed || (!out32 && !out16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || act < 0 || act > 2) return fail(CAE_ERR_ARG, "bad shape or activation");
    GGArgs a{};
    a.in = x16;
    a.out32 = out32;
    a.out16 = out16;
    a.wp = packed;
    a.bias = bias;
    a.N = n;
    a.IH = h;
    a.IW = w;
    a.Ck = ck;
    a.Cn = cn;
    a.OH = h;
    a.OW = w;
    a.LH = h;
    a.LW = w;
    a.S = 1;
    a.SO = 1;
    a.reflect = 0;
    a.act = act;
    a.ktaps = 1;
    a.ntaps = 1;
    a.dy[0] = a.dx[0] = a.wt[0] = 0;
    return launch_gg(a, (hipStream_t)stream);
}

int cae_t_wgrad_pointwise(const void *x16, const void *y16, int n, int h, int w, int ca, int cb, float *gw32, void *stream) {
    if (n < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad shape");
    return wgrad_impl(x16, n, h, w, ca, y16, h, w, cb, 1, 0, 1, gw32, stream);
}

int cae_t_im2col_s2(const float *x_nchw, int n, int c, int h, int w, int oh, int ow, int ks, int reflect, void *out16,
                    void *stream) {
    if (!x_nchw || !out16) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || (ks != 3 && ks != 5) || ks * ks * c > 32)
        return fail(CAE_ERR_ARG, "bad shape (kernel_size^2 * channels must fit 32)");
    hipLaunchKernelGGL(im2col_s2_kernel, dim3(ew_grid((size_t)n * oh * ow * 4)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                       (__bf16 *)out16, n, c, h, w, oh, ow, ks, reflect);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_col2im_s2(const float *u32, const float *bias, int n, int c, int h, int w, int ks, float *out_nchw, void *stream) {
    if (!u32 || !out_nchw) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || h < 1 || w < 1 || (ks != 3 && ks != 5) || ks * ks * c > 32)
        return fail(CAE_ERR_ARG, "bad shape (kernel_size^2 * channels must fit 32)");
    hipLaunchKernelGGL(col2im_s2_kernel, dim3(ew_grid((size_t)n * 4 * h * w)), dim3(256), 0, (hipStream_t)stream, u32, bias,
                       out_nchw, n, c, h, w, ks);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_act_backward(const void *g16, float *gext32, int pad, const void *y16, int n, int h, int w, int cp, int act,
                       void *out16, void *stream) {
    if ((!g16 && !gext32) || !y16 || !out16) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || pad < 0 || cp % 32 || act < 1 || act > 2) return fail(CAE_ERR_ARG, "bad shape or activation");
    hipStream_t st = (hipStream_t)stream;
    if (!g16 && pad > 0) {  // reflect fold of the extended-domain gradient, in place
        hipLaunchKernelGGL(fold_inplace_kernel, dim3((unsigned)(n * (2 * pad + 1))), dim3(256), 0, st, gext32, h, w, pad, cp);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid((size_t)n * h * w * cp)), dim3(256), 0, st, (const __bf16 *)g16,
                       FoldSrc{gext32, h, w, g16 ? 0 : pad}, (const __bf16 *)y16, act == 1 ? 0.01f : 0.0f, (__bf16 *)out16, n, h,
                       w, cp);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_conv_dgrad_ext(const void *gz16, int n, int oh, int ow, int cout_p, const void *packed, int ks, int h, int w,
                         float *gext32, int cin_p, void *stream) {
    if (!gz16 || !packed || !gext32) return fail(CAE_ERR_ARG, "NULL argument");
    if (oh != (h + 1) / 2 || ow != (w + 1) / 2) return fail(CAE_ERR_ARG, "gradient shape does not match the input shape");
    const int P = ks / 2, eh = h + 2 * P, ew = w + 2 * P;
    // (every position of the extended domain belongs to exactly one of the four parity launches below and each of them has
    //  at least one tap for k >= 2, so all of gext32 is written: positions beyond 2 oh + k - 2 (odd input sizes) read
    //  nothing but the zero page and come out as 0 -- no memset of the ~1 GB tensor)
    return strided_corr_t(gz16, n, oh, ow, cout_p, packed, ks, 0, gext32, nullptr, cin_p, eh, ew, nullptr, (hipStream_t)stream);
}

int cae_t_deconv_forward(const void *x16, int n, int h, int w, int cin_p, const void *packed, int ks, float *z32, void *z16,
                         int cout_p, const float *bias, void *stream) {
    if (!x16 || !packed || (!z32 && !z16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad shape");
    return strided_corr_t(x16, n, h, w, cin_p, packed, ks, ks / 2, z32, z16, cout_p, 2 * h, 2 * w, bias, (hipStream_t)stream);
}

int cae_t_deconv_dgrad(const void *gz16, int n, int h, int w, int cout_p, const void *packed, int ks, float *gx32, void *gx16,
                       int cin_p, void *stream) {
    if (!gz16 || !packed || (!gx32 && !gx16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad shape");
    return strided_corr(gz16, n, 2 * h, 2 * w, cout_p, packed, ks, 0, gx32, gx16, cin_p, h, w, nullptr, (hipStream_t)stream);
}

int cae_t_wgrad(const void *xbig16, int n, int h, int w, int ca, const void *ysmall16, int oh, int ow, int cb, int ks,
                int reflect, float *gw32, void *stream) {
    if (n < 1 || oh < 1 || ow < 1 || 2 * oh < h || 2 * ow < w) return fail(CAE_ERR_ARG, "bad shape");
    return wgrad_impl(xbig16, n, h, w, ca, ysmall16, oh, ow, cb, ks, reflect, 2, gw32, stream);
}

int cae_t_wgrad_s1(const void *x16, int n, int h, int w, int ca, const void *y16, int cb, int ks, int reflect, float *gw32,
                   void *stream) {
    if (n < 1 || h < 1 || w < 1) return fail(CAE_ERR_ARG, "bad shape");
    return wgrad_impl(x16, n, h, w, ca, y16, h, w, cb, ks, reflect, 1, gw32, stream);
}

static int wgrad_impl(const void *xbig16, int n, int h, int w, int ca, const void *ysmall16, int oh, int ow, int cb, int ks,
                      int reflect, int S, float *gw32, void *stream) {
    if (!xbig16 || !ysmall16 || !gw32) return fail(CAE_ERR_ARG, "NULL argument");
    if (ks != 1 && ks != 3 && ks != 5) return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (1, 3 or 5)", ks);
    if (bad_channels(ca) || bad_channels(cb)) return fail(CAE_ERR_ARG, "channel counts must be multiples of 32, at most 192");
    hipStream_t st = (hipStream_t)stream;
    WGArgs a{};
    a.S = S;
    a.x = xbig16;
    a.y = ysmall16;
    a.gw = gw32;
    a.zero = zero_page();
    if (!a.zero) return fail(CAE_ERR_NOMEM, "zero page");
    a.N = n;
    a.H = h;
    a.W = w;
    a.Ca = ca;
    a.OH = oh;
    a.OW = ow;
    a.Cb = cb;
    a.reflect = reflect;
    a.kk = ks * ks;
    const int P = ks / 2;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            a.dy[ky * ks + kx] = (short)(ky - P);
            a.dx[ky * ks + kx] = (short)(kx - P);
        }
    a.dymin = -P;
    a.dxmin = -P;
    a.HR = S * 7 + ks;
    a.HC = S * 15 + ks;
    a.m_hc = (unsigned)(((1ull << 32) + (unsigned)a.HC - 1) / (unsigned)a.HC);
    a.m_ypp = (unsigned)(((1ull << 32) + (unsigned)(cb / 8) - 1) / (unsigned)(cb / 8));
    a.tiles_x = (ow + 15) / 16;
    a.tiles_y = (oh + 7) / 8;
    a.total_tiles = n * a.tiles_x * a.tiles_y;
    HIP_TRY(hipMemsetAsync(gw32, 0, (size_t)a.kk * ca * cb * sizeof(float), st));
    const size_t lds = (size_t)((a.HR * a.HC * 4 + 63) / 64) * 1024 + (size_t)((128 * (cb / 8) + 63) / 64) * 1024;
    const int a_tiles = ca / 32, tap_groups = (a.kk + 8) / 9;
    {
        // wgrad8_kernel: 8 waves, double-buffered samples; needs two staging buffers in the LDS and Cb <= 128
        static const char *e8 = std::getenv("CAE_WG8");
        // up to 128 b channels per launch; a wider Y (192) goes in two halves of 96 (three of the four b-tile waves busy)
        // (measured: the two-launch form is SLOWER than wgrad_kernel<2> on the canonical 192-channel layers -- 13.97 vs 13.59 ms
        //  per 128 x 256^2 step, X staged twice and a quarter of the waves idle -- so it is opt-in: CAE_WG8_WIDE=1)
        static const char *ew = std::getenv("CAE_WG8_WIDE");
        const bool wide_off = cb > 128 && !(ew && ew[0] == '1');
        const int parts = cb <= 128 ? 1 : 2, cbl = cb / parts;
        const int x_instr = (a.HR * a.HC * 4 + 63) / 64, y_instr = (128 * (cbl / 8) + 63) / 64;
        // (two staging buffers; at least the 80 KiB in which the position groups merge their partial sums at the end)
        const size_t lds8 = std::max<size_t>(2 * (size_t)(x_instr + y_instr) * 1024, 4 * 5 * 16 * 64 * sizeof(float));
        if (!(e8 && e8[0] == '0') && !wide_off && cbl % 32 == 0 && cbl <= 128 && x_instr <= 64 && lds8 <= 160 * 1024) {
            static size_t attr = 0;
            if (lds8 > attr) {
                HIP_TRY(hipFuncSetAttribute((const void *)wgrad8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
                attr = lds8;
            }
            const int tpi = a.tiles_x * a.tiles_y;
            // sample lanes: about two rounds of blocks over the 256 CUs, each block walking n, n + step, ...
            // sample lanes: about ONE block per CU, and at least four samples per block -- every block ends with an atomic flush
            // of its 9 x 32 x Cb partial sums, and that flush, not the contraction, set the time with more blocks
            // (128 -> 128, 128^2 / 64^2 inputs; batch 128: 512 blocks 0.50 / 0.18 ms, 256 blocks 0.46 / 0.14 ms;
            //  batch 16: 512 blocks 0.15 / 0.13 ms, 256 / 128 blocks 0.105 / 0.053 ms).  CAE_WG8_BLOCKS overrides the target.
            const char *estep = std::getenv("CAE_WG8_BLOCKS");
            const int target = estep ? std::max(1, std::atoi(estep)) : 256;
            const int base = std::max(1, tpi * a_tiles * tap_groups);
            const int step = std::max(1, std::min(std::max(1, estep ? n : n / 4), target / base));
            for (int part = 0; part < parts; ++part) {
                WGArgs b = a;
                b.Cbs = cb;
                b.cb0 = part * cbl;
                b.Cb = cbl;
                b.m_ypp = (unsigned)(((1ull << 32) + (unsigned)(cbl / 8) - 1) / (unsigned)(cbl / 8));
                hipLaunchKernelGGL(wgrad8_kernel, dim3(tpi * step, a_tiles, tap_groups), dim3(512), lds8, st, b, tpi, step);
                HIP_TRY(hipGetLastError());
            }
            return CAE_OK;
        }
    }
    // (about one block per CU here too: 256 blocks 13.06 - 13.11 ms per 128 x 256^2 step, 512: 13.18, 128: 13.15 - 13.18, 64: 13.45 - 13.5)
    const char *eblk = std::getenv("CAE_WG_BLOCKS");
    const int ksplit = std::max(1, std::min(a.total_tiles, (eblk ? std::max(1, std::atoi(eblk)) : 256) / (a_tiles * tap_groups)));
    if (cb / 32 <= 4) return launch_wg_t<1>(a, lds, ksplit, a_tiles, tap_groups, st);
    return launch_wg_t<2>(a, lds, ksplit, a_tiles, tap_groups, st);
}

int cae_t_gdn_forward(const float *z32, long pixels, int cp, const float *beta, const float *gamma, int inverse, float *y32,
                      void *y16, void *stream) {
    if (!z32 || !beta || !gamma || (!y32 && !y16)) return fail(CAE_ERR_ARG, "NULL argument");
    if (pixels < 1 || bad_channels(cp)) return fail(CAE_ERR_ARG, "bad shape");
    GdnArgs a{};
    a.a = z32;
    a.mat = gamma;
    a.beta = beta;
    a.z = z32;
    a.o32a = y32;
    a.o16 = y16;
    a.pixels = pixels;
    a.C = cp;
    a.inverse = inverse;
    return launch_gdn_a<0>(a, (hipStream_t)stream);
}

int cae_t_gdn_backward(const float *z32, const float *gext32, int n, int h, int w, int pad, int cp, const float *beta,
                       const float *gamma, const float *gamma_t, int inverse, float *gn_ws32, float *gzd_ws32, float *gz32,
                       void *gz16, float *ggamma, float *gbeta, void *stream) {
    if (!z32 || !gext32 || !beta || !gamma || !gamma_t || !gn_ws32 || !gzd_ws32 || (!gz32 && !gz16) || !ggamma || !gbeta)
        return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || pad < 0 || bad_channels(cp)) return fail(CAE_ERR_ARG, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long pixels = (long)n * h * w;
    if (pixels >= (1l << 31)) return fail(CAE_ERR_ARG, "more than 2^31 pixels per call");
    GdnArgs a{};
    a.a = z32;
    a.mat = gamma;
    a.beta = beta;
    a.z = z32;
    a.gy = FoldSrc{gext32, h, w, pad};
    a.img_h = h;
    a.img_w = w;
    a.o32a = gn_ws32;
    a.o32b = gzd_ws32;
    a.pixels = pixels;
    a.C = cp;
    a.inverse = inverse;
    int rc = launch_gdn_a<1>(a, st);
    if (rc) return rc;
    GdnArgs b{};
    b.a = gn_ws32;
    b.mat = gamma_t;
    b.z = z32;
    b.o32a = gz32;
    b.o32b = gzd_ws32;
    b.o16 = gz16;
    b.pixels = pixels;
    b.C = cp;
    b.inverse = inverse;
    if ((rc = launch_gdn_a<2>(b, st))) return rc;
    HIP_TRY(hipMemsetAsync(ggamma, 0, (size_t)cp * cp * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(gbeta, 0, (size_t)cp * sizeof(float), st));
    switch (cp / 32) {
        case 1: return launch_gdn_b_t<1>(gn_ws32, z32, pixels, ggamma, gbeta, st);
        case 2: return launch_gdn_b_t<2>(gn_ws32, z32, pixels, ggamma, gbeta, st);
        case 3: return launch_gdn_b_t<3>(gn_ws32, z32, pixels, ggamma, gbeta, st);
        case 4: return launch_gdn_b_t<4>(gn_ws32, z32, pixels, ggamma, gbeta, st);
        case 5: return launch_gdn_b_t<5>(gn_ws32, z32, pixels, ggamma, gbeta, st);
        default: return launch_gdn_b_t<6>(gn_ws32, z32, pixels, ggamma, gbeta, st);
    }
}

size_t cae_t_gdn_saved_elems(long pixels, int cp) {
    if (pixels < 1 || cp < 32 || cp > 128 || cp % 32) return 0;
    return (size_t)((pixels + 63) / 64) * 64 * (size_t)cp;
}

int cae_t_gdn_forward_save(const float *z32, long pixels, int cp, const float *beta, const float *gamma, int inverse,
                           void *y16, float *f_saved, void *stream) {
    if (!z32 || !beta || !gamma || !y16 || !f_saved) return fail(CAE_ERR_ARG, "NULL argument");
    if (pixels < 1 || pixels >= (1l << 31) || cp % 32) return fail(CAE_ERR_ARG, "bad shape");
    GdnFusedArgs a{};
    a.z = z32;
    a.gamma = gamma;
    a.beta = beta;
    a.f = f_saved;
    a.y16 = y16;
    a.pixels = pixels;
    a.inverse = inverse;
    return launch_gdn_fused(a, cp, false, (hipStream_t)stream);
}

int cae_t_gdn_backward_fused(const float *z32, const float *f_saved, float *gext32, int n, int h, int w, int pad, int cp,
                             const float *gamma, int inverse, void *gz16, float *ggamma, float *gbeta, void *stream) {
    if (!z32 || !f_saved || !gext32 || !gamma || !gz16 || !ggamma || !gbeta) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || pad < 0 || cp % 32) return fail(CAE_ERR_ARG, "bad shape");
    const long pixels = (long)n * h * w;
    if (pixels >= (1l << 31)) return fail(CAE_ERR_ARG, "more than 2^31 pixels per call");
    if (cp > 128) return fail(CAE_ERR_UNSUPPORTED, "fused GDN kernels are built for at most 128 channels, got %d", cp);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(ggamma, 0, (size_t)cp * cp * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(gbeta, 0, (size_t)cp * sizeof(float), st));
    if (pad > 0) {  // reflect fold of the extended-domain gradient, in place (touches the border pixels only)
        hipLaunchKernelGGL(fold_inplace_kernel, dim3((unsigned)(n * (2 * pad + 1))), dim3(256), 0, st, gext32, h, w, pad, cp);
        HIP_TRY(hipGetLastError());
    }
    GdnFusedArgs a{};
    a.z = z32;
    a.gamma = gamma;
    a.f = const_cast<float *>(f_saved);
    a.gy = FoldSrc{gext32, h, w, pad};
    a.img_h = h;
    a.img_w = w;
    a.gz16 = gz16;
    a.ggamma = ggamma;
    a.gbeta = gbeta;
    a.pixels = pixels;
    a.inverse = inverse;
    return launch_gdn_fused(a, cp, true, st);
}

int cae_t_fold_to_bf16(const float *gext32, int n, int h, int w, int pad, int cp, void *out16, void *stream) {
    if (!gext32 || !out16) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || h < 1 || w < 1 || pad < 0 || cp % 32) return fail(CAE_ERR_ARG, "bad shape");
    hipLaunchKernelGGL(fold_to_bf16_kernel, dim3(ew_grid((size_t)n * h * w * cp)), dim3(256), 0, (hipStream_t)stream,
                       FoldSrc{gext32, h, w, pad}, (__bf16 *)out16, n, cp);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_bn_moments(const float *a, const float *b, int n, int c, long hw, double *s1, double *s2, void *stream) {
    if (!a || !b || !s1 || !s2) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || c > 65535 || hw < 1) return fail(CAE_ERR_ARG, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(s1, 0, (size_t)c * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(s2, 0, (size_t)c * sizeof(double), st));
    const long total = (long)n * hw;
    const unsigned splits = (unsigned)std::min<long>(std::max<long>(total / 4096, 1), std::max<long>(2048 / c, 1));
    hipLaunchKernelGGL(bn_moments_kernel, dim3((unsigned)c, splits), dim3(256), 0, st, a, b, n, c, hw, s1, s2);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_bn_affine(const float *a, const float *b, int n, int c, long hw, const float *A, const float *B, const float *C,
                    float *out, void *stream) {
    if (!a || !A || !C || !out || (b && !B)) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || c < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad shape");
    const size_t total = (size_t)n * c * hw;
    hipLaunchKernelGGL(bn_affine_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, b, c, hw, total, A, B, C,
                       out);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_colsum(const void *g16, long pixels, int cp, float *out, void *stream) {
    if (!g16 || !out) return fail(CAE_ERR_ARG, "NULL argument");
    if (pixels < 1 || cp % 32 || cp > 256) return fail(CAE_ERR_ARG, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)cp * sizeof(float), st));
    const int threads = (256 / cp) * cp;
    const long rows = 256 / cp;
    const unsigned grid = (unsigned)std::min<long>(std::max<long>((pixels + rows - 1) / rows, 1), 1024);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(grid), dim3(threads), 0, st, (const __bf16 *)g16, pixels, cp, out);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // extern "C"
