// (generated split of the launcher code: one translation unit per kernel family so hipcc
//  compiles them in parallel; see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
#include "cae_kernels.hpp"
namespace cae {
template <int KS, int CT, bool GDN>
static int launch_first_t(const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int WH = 2 * 16 + KS - 2, HH = 4 * NW + KS - 2;
    constexpr int LDS = 2 * (GDN ? CT * 4096 : 0) + KS * KS * CT * 512 + ((HH * WH * 16 + 1023) / 1024) * 1024 + 1024;
    auto kern = conv_first_kernel<KS, CT, NW, GDN>;
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

template <int KS>
static int launch_last_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int P = KS / 2;
    constexpr int NB = (KS - 1 - P) / 2 + (P + 1) / 2 + 1;
    constexpr int HALO_INSTR = (4 * (NW + NB - 1) * (64 + NB - 1) + 63) / 64;
    const int lds = 2 * HALO_INSTR * 1024 + NB * NB * a.cci * 1024;
    auto kern = deconv_last_kernel<KS, NW>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned grid = (unsigned)((size_t)a.N * a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

template <int CT, bool INV>
static int launch_gdn_t(const LayerArgs &a, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int LDS = 2 * CT * 4096;
    auto kern = gdn_c8_kernel<CT, NW, INV>;
    const int hw = a.H * a.W;
    const unsigned grid = (unsigned)((size_t)a.N * ((hw + NW * 32 - 1) / (NW * 32)));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, st, a);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

#define DISPATCH_CT_F(KS_, GDN_)                                                 \
    switch (ct) {                                                                \
        case 1: return launch_first_t<KS_, 1, GDN_>(a, f, st);                   \
        case 2: return launch_first_t<KS_, 2, GDN_>(a, f, st);                   \
        case 4: return launch_first_t<KS_, 4, GDN_>(a, f, st);                   \
        case 6: return launch_first_t<KS_, 6, GDN_>(a, f, st);                   \
        default: return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct); \
    }

int launch_first(int ks, int ct, bool gdn, const LayerArgs &a, const FirstArgs &f, hipStream_t st) {
    if (ks == 3) {
        if (gdn) { DISPATCH_CT_F(3, true) } else { DISPATCH_CT_F(3, false) }
    } else if (ks == 5) {
        if (gdn) { DISPATCH_CT_F(5, true) } else { DISPATCH_CT_F(5, false) }
    }
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

int launch_last(int ks, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) return launch_last_t<3>(a, st);
    if (ks == 5) return launch_last_t<5>(a, st);
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

int launch_gdn(int ct, bool inverse, const LayerArgs &a, hipStream_t st) {
    switch (ct) {
        case 1: return inverse ? launch_gdn_t<1, true>(a, st) : launch_gdn_t<1, false>(a, st);
        case 2: return inverse ? launch_gdn_t<2, true>(a, st) : launch_gdn_t<2, false>(a, st);
        case 4: return inverse ? launch_gdn_t<4, true>(a, st) : launch_gdn_t<4, false>(a, st);
        case 6: return inverse ? launch_gdn_t<6, true>(a, st) : launch_gdn_t<6, false>(a, st);
    }
    return fail(CAE_ERR_UNSUPPORTED, "unsupported channel tiles %d", ct);
}

}  // namespace cae
