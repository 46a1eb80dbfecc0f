// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels_f16.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_conv_f16_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_CONV_F16_NW;
    constexpr int WH = 2 * 16 + KS - 2;
    constexpr int HALO_INSTR = (4 * 16 * WH + 63) / 64;
    constexpr int CONV_STAGE = KS * CT * 2 * 1024 + HALO_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int LDS = 2 * (CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES);
    if constexpr (LDS > 160 * 1024) {
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: this kernel_size/channel combination exceeds the LDS; use fp32");
    } else {
        auto kern = conv_s2_f16_kernel<KS, CT, GDN>;
        static bool attr_done = false;
        if (!attr_done) {
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_done = true;
        }
        const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, st, a);
        HIP_TRY(hipGetLastError());
        return CAE_OK;
    }
}

#define DISPATCH_F16(FN, KS_)                                                               \
    switch (ct) {                                                                          \
        case 1: return gdn ? FN<KS_, 1, true>(a, st) : FN<KS_, 1, false>(a, st);           \
        case 2: return gdn ? FN<KS_, 2, true>(a, st) : FN<KS_, 2, false>(a, st);           \
        case 4: return gdn ? FN<KS_, 4, true>(a, st) : FN<KS_, 4, false>(a, st);           \
        case 6: { /* wider than 128 channels: the normalisation runs as a kernel of its own */ \
            const int rc6 = FN<KS_, 6, false>(a, st);                                      \
            return rc6 || !gdn ? rc6 : launch_gdn_f16(6, false, a, st);                    \
        }                                                                                  \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);      \
    }

// stride-1 stage of a LeakyReLU / ReLU unit or of a residual unit.  Epilogue: GDN (analysis) / IGDN (synthesis) or the
// activation a.act, then -- up to 128 channels -- + a.res and a.post_act.  SYN: C8SP rows; ZP: zero padding (the synthesis
// units' transposed convolutions) instead of reflection (analysis units, and the colour layers of the synthesis track).
template <int KS, int CT, bool GDN, bool SYN, bool ZP>
static int launch_conv_s1_f16_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_CONV_F16_NW;
    constexpr int WH = 16 + KS - 1;
    constexpr int HALO_INSTR = (4 * 16 * WH + 63) / 64;
    constexpr int CONV_STAGE = KS * CT * 2 * 1024 + HALO_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int LDS = 2 * (CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES);
    if constexpr (LDS > 160 * 1024) {
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: this kernel_size/channel combination exceeds the LDS; use fp32");
    } else {
        auto kern = conv_s2_f16_kernel<KS, CT, GDN, 1, SYN, ZP, (CT <= 4)>;
        static bool attr_done = false;
        if (!attr_done) {
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_done = true;
        }
        const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, st, a);
        HIP_TRY(hipGetLastError());
        return CAE_OK;
    }
}

int launch_conv_s1_f16(int ks, int ct, bool synthesis, bool gdn, const LayerArgs &a, hipStream_t st) {
    if ((gdn || a.res || a.post_act) && ct > 4)
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: GDN / residual stages wider than 128 channels run on the fp32 path");
#define S1_SIDE(KS_, CT_, GDN_) \
    return synthesis ? launch_conv_s1_f16_t<KS_, CT_, GDN_, true, true>(a, st) : launch_conv_s1_f16_t<KS_, CT_, GDN_, false, false>(a, st)
#define S1_CASE(KS_)                                                                   \
    switch (ct) {                                                                      \
        case 1: if (gdn) { S1_SIDE(KS_, 1, true); } else { S1_SIDE(KS_, 1, false); }   \
        case 2: if (gdn) { S1_SIDE(KS_, 2, true); } else { S1_SIDE(KS_, 2, false); }   \
        case 4: if (gdn) { S1_SIDE(KS_, 4, true); } else { S1_SIDE(KS_, 4, false); }   \
        case 6: S1_SIDE(KS_, 6, false);                                                \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);  \
    }
    if (ks == 3) { S1_CASE(3) }
    if (ks == 5) { S1_CASE(5) }
#undef S1_CASE
#undef S1_SIDE
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

// multiscale colour layer (_autoencoders.py:417-436): reflect convolution of a synthesis level (C8SP rows) to the image
// channels, NCHW fp32 out
int launch_color_f16(int ks, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) return launch_conv_s1_f16_t<3, 1, false, true, false>(a, st);
    if (ks == 5) return launch_conv_s1_f16_t<5, 1, false, true, false>(a, st);
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

int launch_conv_f16(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) { DISPATCH_F16(launch_conv_f16_t, 3) }
    if (ks == 5) { DISPATCH_F16(launch_conv_f16_t, 5) }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
