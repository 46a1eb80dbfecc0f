// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include <algorithm>

#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels_f16.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_first_f16_t(const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
    using G = FirstGeomF16<KS, CT, GDN>;
    constexpr int LDS = G::LDS_BYTES;
    if constexpr (LDS > 160 * 1024) {
        return fail(CAE_ERR_UNSUPPORTED, "f16x3: first-layer operands exceed the LDS for this shape; use fp32");
    } else {
        auto kern_u8 = conv_first_f16_kernel<KS, CT, GDN, true>;
        auto kern_f32 = conv_first_f16_kernel<KS, CT, GDN, false>;
        static bool attr_done = false;
        static int n_cu = 0;
        if (!attr_done) {
            HIP_TRY(hipFuncSetAttribute((const void *)kern_u8, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            HIP_TRY(hipFuncSetAttribute((const void *)kern_f32, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
            attr_done = true;
        }
        LayerArgs b = a;
        b.tiles_x = (a.OW + G::TX - 1) / G::TX;
        b.tiles_y = (a.OH + G::TY - 1) / G::TY;
        const size_t total = (size_t)b.N * b.tiles_x * b.tiles_y;
        if (total > 0x7fffffff) return fail(CAE_ERR_ARG, "batch too large");
        // persistent: one block per CU walks over the tiles
        const unsigned grid = (unsigned)std::min<size_t>(total, (size_t)std::max(n_cu, 1));
        hipLaunchKernelGGL(f.in_is_u8 ? kern_u8 : kern_f32, dim3(grid), dim3(G::NW * 64), LDS, st, b, f);
        HIP_TRY(hipGetLastError());
        return CAE_OK;
    }
}

int launch_first_f16(int ks, int ct, bool gdn, const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
#define FIRST_F16(KS_)                                                                                       \
    switch (ct) {                                                                                            \
        case 1: return gdn ? launch_first_f16_t<KS_, 1, true>(a, f, st) : launch_first_f16_t<KS_, 1, false>(a, f, st); \
        case 2: return gdn ? launch_first_f16_t<KS_, 2, true>(a, f, st) : launch_first_f16_t<KS_, 2, false>(a, f, st); \
        case 4: return gdn ? launch_first_f16_t<KS_, 4, true>(a, f, st) : launch_first_f16_t<KS_, 4, false>(a, f, st); \
        case 6: {                                                                                            \
            const int rc6 = launch_first_f16_t<KS_, 6, false>(a, f, st);                                     \
            return rc6 || !gdn ? rc6 : launch_gdn_f16(6, false, a, st);                                      \
        }                                                                                                    \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);                       \
    }
    if (ks == 3) { FIRST_F16(3) }
    if (ks == 5) { FIRST_F16(5) }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

template <int CT, bool INVERSE>
static int launch_gdn_f16_t(const LayerArgs &a, hipStream_t st) {
    constexpr int LDS = CT * CT * 4096 + CT * 32 * 4;
    static_assert(LDS <= 160 * 1024, "packed gamma must fit the LDS");
    auto kern = gdn_f16_kernel<CT, INVERSE, INVERSE>;  // analysis rows are C8S, synthesis rows C8SP
    static bool attr_done = false;
    static int n_cu = 0;
    if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    const size_t groups = c8s_row_bytes<INVERSE>(a.OW) / 1024;
    const size_t total = (size_t)a.N * a.OH * groups;  // wave tiles of 32 pixels
    if (total > 0x7fffffff) return fail(CAE_ERR_ARG, "batch too large");
    const unsigned grid = (unsigned)std::min<size_t>((total + 3) / 4, (size_t)std::max(n_cu, 1));  // persistent
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int launch_gdn_f16(int ct, bool inverse, const LayerArgs &a, hipStream_t st) {
    if (a.outfmt != OUT_C8) return fail(CAE_ERR_UNSUPPORTED, "f16x3: stand-alone GDN needs split rows as output");
    if (ct == 6) return inverse ? launch_gdn_f16_t<6, true>(a, st) : launch_gdn_f16_t<6, false>(a, st);
    return fail(CAE_ERR_UNSUPPORTED, "stand-alone GDN is built for 192-channel layers (channel tiles %d)", ct);
}

template <int KS, int NW, int DEPTH>
static int launch_last_f16_t(const LayerArgs &a, hipStream_t st) {
    using G = LastGeomF16<KS, NW, DEPTH>;
    const int lds = G::lds_bytes(a.cci);
    if (lds > 160 * 1024) return fail(CAE_ERR_UNSUPPORTED, "last-layer weights do not fit the LDS");
    auto kern = deconv_last_f16_kernel<KS, NW, DEPTH>;
    static bool attr_done = false;
    static int n_cu = 0;
    if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        attr_done = true;
    }
    LayerArgs b = a;
    b.tiles_x = (a.W + G::TXC - 1) / G::TXC;
    b.tiles_y = (a.H + NW - 1) / NW;
    const size_t total = (size_t)b.N * b.tiles_x * b.tiles_y;
    if (total > 0x7fffffff) return fail(CAE_ERR_ARG, "batch too large");
    const unsigned grid = (unsigned)std::min<size_t>(total, (size_t)std::max(n_cu, 1));  // persistent
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, b);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int launch_last_f16(int ks, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) return launch_last_f16_t<3, 8, 3>(a, st);  // 3 x 40 KiB ring + 32 KiB weights (128 channels)
    if (ks == 5) return launch_last_f16_t<5, 4, 2>(a, st);  // 3x3 neighbours: 72 KiB of weights, shallower ring
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
