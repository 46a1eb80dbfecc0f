// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels_f16.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_first_f16_t(const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int WH = 2 * 16 + KS - 2, HH = 4 * NW + KS - 2;
    constexpr int NS = (KS * KS + 3) / 4;
    constexpr int LDS = 2 * (GDN ? CT * 4096 : 0) + NS * CT * 2048 + ((HH * WH * 16 + 1023) / 1024) * 1024 + 1024;
    auto kern = conv_first_f16_kernel<KS, CT, GDN>;
    static bool attr_done = false;
    if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_done = true;
    }
    const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, st, a, f);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int launch_first_f16(int ks, int ct, bool gdn, const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
#define FIRST_F16(KS_)                                                                                       \
    switch (ct) {                                                                                            \
        case 1: return gdn ? launch_first_f16_t<KS_, 1, true>(a, f, st) : launch_first_f16_t<KS_, 1, false>(a, f, st); \
        case 2: return gdn ? launch_first_f16_t<KS_, 2, true>(a, f, st) : launch_first_f16_t<KS_, 2, false>(a, f, st); \
        case 4: return gdn ? launch_first_f16_t<KS_, 4, true>(a, f, st) : launch_first_f16_t<KS_, 4, false>(a, f, st); \
        case 6:                                                                                              \
            if (!gdn) return launch_first_f16_t<KS_, 6, false>(a, f, st);                                    \
            return fail(CAE_ERR_UNSUPPORTED, "f16x3: GDN with more than 128 channels is not built; use fp32"); \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);                       \
    }
    if (ks == 3) { FIRST_F16(3) }
    if (ks == 5) { FIRST_F16(5) }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

template <int KS>
static int launch_last_f16_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int P = KS / 2;
    constexpr int NB = (KS - 1 - P) / 2 + (P + 1) / 2 + 1;
    constexpr int HALO_INSTR = (8 * (NW + NB - 1) * (32 + NB - 1) + 63) / 64;
    const int lds = 2 * HALO_INSTR * 1024 + NB * NB * a.cci * 2048;
    if (lds > 160 * 1024) return fail(CAE_ERR_UNSUPPORTED, "last-layer weights do not fit the LDS");
    auto kern = deconv_last_f16_kernel<KS, NW>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int launch_last_f16(int ks, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) return launch_last_f16_t<3>(a, st);
    if (ks == 5) return launch_last_f16_t<5>(a, st);
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
