// Launchers of the kernel families; each is defined in its own translation unit (cae_launch_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "cae_internal.hpp"

#ifndef CAE_CONV_NW
#define CAE_CONV_NW 4
#endif
#ifndef CAE_DECONV_NW
#define CAE_DECONV_NW 4
#endif

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ::cae::fail(CAE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

namespace cae {
struct LayerArgs;
struct FirstArgs;
int launch_conv(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_conv_s1(int ks, int ct, bool zeropad, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_deconv(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_gdn(int ct, bool inverse, const LayerArgs &a, hipStream_t st);
int launch_first(int ks, int ct, bool gdn, const LayerArgs &a, const FirstArgs &f, hipStream_t st);
int launch_last(int ks, const LayerArgs &a, hipStream_t st);
int launch_conv_f16(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_deconv_f16(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_conv_s1_f16(int ks, int ct, bool synthesis, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_color_f16(int ks, const LayerArgs &a, hipStream_t st);
int launch_first_f16(int ks, int ct, bool gdn, const LayerArgs &a, const FirstArgs &f, hipStream_t st);
int launch_last_f16(int ks, const LayerArgs &a, hipStream_t st);
// GDN / IGDN in place on the split rows a.out (layers wider than 128 channels; a.outfmt must be OUT_C8)
int launch_gdn_f16(int ct, bool inverse, const LayerArgs &a, hipStream_t st);
}  // namespace cae
