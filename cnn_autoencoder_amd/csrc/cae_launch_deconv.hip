// (launcher code is split into one translation unit per kernel family so hipcc compiles them in parallel;
//  see cae_launch.hpp)
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"
namespace cae {
int launch_deconv_k3(int ct, bool gdn, const LayerArgs &a, hipStream_t st);
int launch_deconv_k5(int ct, bool gdn, const LayerArgs &a, hipStream_t st);

int launch_deconv(int ks, int ct, bool gdn, const LayerArgs &a, hipStream_t st) {
    if (ks == 3) return launch_deconv_k3(ct, gdn, a, st);
    if (ks == 5) return launch_deconv_k5(ct, gdn, a, st);
    return fail(CAE_ERR_UNSUPPORTED, "kernel_size %d not supported (3 or 5)", ks);
}

}  // namespace cae
