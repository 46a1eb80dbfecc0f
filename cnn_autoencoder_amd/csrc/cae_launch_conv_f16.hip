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

// stride-1 stage of a LeakyReLU / ReLU unit (activation in the epilogue: a.act); synthesis: C8SP rows, zero padding
template <int KS, int CT, bool SYN>
static int launch_conv_s1_f16_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_CONV_F16_NW;
    constexpr int WH = 16 + KS - 1;
    constexpr int HALO_INSTR = (4 * 16 * WH + 63) / 64;
    constexpr int LDS = 2 * (KS * CT * 2 * 1024 + HALO_INSTR * 1024);
    if constexpr (LDS > 160 * 1024) {
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: this kernel_size/channel combination exceeds the LDS; use fp32");
    } else {
        auto kern = conv_s2_f16_kernel<KS, CT, false, 1, SYN, SYN>;
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

int launch_conv_s1_f16(int ks, int ct, bool synthesis, const LayerArgs &a, hipStream_t st) {
#define S1_CASE(KS_)                                                                                               \
    switch (ct) {                                                                                                  \
        case 1: return synthesis ? launch_conv_s1_f16_t<KS_, 1, true>(a, st) : launch_conv_s1_f16_t<KS_, 1, false>(a, st); \
        case 2: return synthesis ? launch_conv_s1_f16_t<KS_, 2, true>(a, st) : launch_conv_s1_f16_t<KS_, 2, false>(a, st); \
        case 4: return synthesis ? launch_conv_s1_f16_t<KS_, 4, true>(a, st) : launch_conv_s1_f16_t<KS_, 4, false>(a, st); \
        case 6: return synthesis ? launch_conv_s1_f16_t<KS_, 6, true>(a, st) : launch_conv_s1_f16_t<KS_, 6, false>(a, st); \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);                              \
    }
    if (ks == 3) { S1_CASE(3) }
    if (ks == 5) { S1_CASE(5) }
#undef S1_CASE
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

int launch_conv_f16(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) { DISPATCH_F16(launch_conv_f16_t, 3) }
    if (ks == 5) { DISPATCH_F16(launch_conv_f16_t, 5) }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
