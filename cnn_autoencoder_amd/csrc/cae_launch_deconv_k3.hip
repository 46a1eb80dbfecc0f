// fp32 transposed-convolution launchers, kernel size 3 (one translation unit per kernel size: this family is the
// slowest to compile, and hipcc compiles translation units in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_deconv_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = CAE_DECONV_NW;
    constexpr int P = KS / 2;
    constexpr int WH = 32 + (KS - 1 - P) / 2 + (P + 1) / 2;
    constexpr int HALO_INSTR = (NW * WH * 2 + 63) / 64;
    constexpr int CONV_STAGE = KS * CT * 1024 + HALO_INSTR * 1024;
    constexpr int G_BYTES = GDN ? CT * 4096 : 0;
    constexpr int LDS = 2 * (CONV_STAGE > G_BYTES ? CONV_STAGE : G_BYTES);
    auto kern = deconv_s2_kernel<KS, CT, NW, GDN>;
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

int launch_deconv_k3(int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (gdn) { DISPATCH_CT(launch_deconv_t, 3, true) } else { DISPATCH_CT(launch_deconv_t, 3, false) }
}

}  // namespace cae
