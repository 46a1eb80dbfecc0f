// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels_f16.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_deconv_f16_t(const LayerArgs &a, hipStream_t st) {
    // 8 waves x 1 input row measured faster than 4 waves x 2 rows (register spills at 512 VGPRs):
    // profiles/r01_experiments.md
#ifndef CAE_DF16_NW
#define CAE_DF16_NW 8
#endif
    constexpr int NW = CAE_DF16_NW, PT = 1;
    using G = DeconvGeomF16<KS, CT, NW, PT, GDN>;
    constexpr int LDS = G::lds_bytes(false);
    if constexpr (2 * G::STAGE_BYTES > 160 * 1024) {
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: this kernel_size/channel combination exceeds the LDS; use fp32");
    } else {
        auto kern = deconv_s2_f16_kernel<KS, CT, NW, PT, GDN>;
        const bool pmap = a.outfmt == OUT_PMAP;
        const int lds = pmap ? G::lds_bytes(true) : LDS;
        if (lds > 160 * 1024) return fail(CAE_ERR_UNSUPPORTED, "product map: LDS exceeded");
        static int attr_bytes = 0;
        if (lds > attr_bytes) {
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr_bytes = lds;
        }
        LayerArgs b = a;
        b.tiles_y = (a.H + G::ROWS - 1) / G::ROWS;  // input rows per block follow the kernel's wave count
        const unsigned grid = (unsigned)((size_t)b.N * b.tiles_x * b.tiles_y);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, b);
        HIP_TRY(hipGetLastError());
        return CAE_OK;
    }
}

#define DISPATCH_F16(FN, KS_, INV)                                                               \
    switch (ct) {                                                                          \
        case 1: return gdn ? FN<KS_, 1, true>(a, st) : FN<KS_, 1, false>(a, st);           \
        case 2: return gdn ? FN<KS_, 2, true>(a, st) : FN<KS_, 2, false>(a, st);           \
        case 4: return gdn ? FN<KS_, 4, true>(a, st) : FN<KS_, 4, false>(a, st);           \
        case 6: { /* wider than 128 channels: the normalisation runs as a kernel of its own */ \
            const int rc6 = FN<KS_, 6, false>(a, st);                                      \
            return rc6 || !gdn ? rc6 : launch_gdn_f16(6, INV, a, st);                      \
        }                                                                                  \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);      \
    }

int launch_deconv_f16(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) { DISPATCH_F16(launch_deconv_f16_t, 3, true) }
    if (ks == 5) { DISPATCH_F16(launch_deconv_f16_t, 5, true) }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
