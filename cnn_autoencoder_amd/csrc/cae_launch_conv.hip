// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_conv_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_CONV_NW;
    constexpr int WH = 2 * 16 + KS - 2;
    constexpr int HALO_INSTR = (2 * NW * WH * 2 + 63) / 64;
    constexpr int CONV_STAGE = KS * CT * 1024 + HALO_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
#ifdef CAE_EXP_1BLOCK
    constexpr int LDS = 96 * 1024;
#else
    constexpr int LDS = 2 * (CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES);
#endif
    auto kern = conv_s2_kernel<KS, CT, NW, GDN>;
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

#define DISPATCH_CT(FN, KS_, GDN_)                                               \
    switch (ct) {                                                                \
        case 1: return FN<KS_, 1, GDN_>(a, st);                                  \
        case 2: return FN<KS_, 2, GDN_>(a, st);                                  \
        case 4: return FN<KS_, 4, GDN_>(a, st);                                  \
        case 6: return FN<KS_, 6, GDN_>(a, st);                                  \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct); \
    }

// stride-1 convolutions of the LeakyReLU / ReLU units: reflect padded (analysis) or zero padded with
// flipped weights (ConvTranspose2d stride 1, synthesis)
template <int KS, int CT, bool ZP, bool GDN>
static int launch_conv_s1_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_CONV_NW;
    constexpr int WH = 15 + KS;
    constexpr int HALO_INSTR = (2 * NW * WH * 2 + 63) / 64;
    constexpr int CONV_STAGE = KS * CT * 1024 + HALO_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int LDS = 2 * (CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES);
    // GDN on the analysis track (reflect padded), IGDN on the synthesis track (zero padded, flipped weights)
    auto kern = conv_s2_kernel<KS, CT, NW, GDN, 1, ZP, GDN && ZP>;
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

int launch_conv_s1(int ks, int ct, bool zeropad, bool gdn, const LayerArgs &a, hipStream_t st) {
#define S1_CT(KS_, ZP_, G_)                                                      \
    switch (ct) {                                                                \
        case 1: return launch_conv_s1_t<KS_, 1, ZP_, G_>(a, st);                 \
        case 2: return launch_conv_s1_t<KS_, 2, ZP_, G_>(a, st);                 \
        case 4: return launch_conv_s1_t<KS_, 4, ZP_, G_>(a, st);                 \
        case 6: return launch_conv_s1_t<KS_, 6, ZP_, G_>(a, st);                 \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct); \
    }
#define S1_ZP(KS_, G_)                                   \
    if (zeropad) { S1_CT(KS_, true, G_) } else { S1_CT(KS_, false, G_) }
    if (ks == 3) {
        if (gdn) { S1_ZP(3, true) } else { S1_ZP(3, false) }
    } else if (ks == 5) {
        if (gdn) { S1_ZP(5, true) } else { S1_ZP(5, false) }
    }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

int launch_conv(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) {
        if (gdn) { DISPATCH_CT(launch_conv_t, 3, true) } else { DISPATCH_CT(launch_conv_t, 3, false) }
    } else if (ks == 5) {
        if (gdn) { DISPATCH_CT(launch_conv_t, 5, true) } else { DISPATCH_CT(launch_conv_t, 5, false) }
    }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
