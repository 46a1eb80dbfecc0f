// Device -> pinned-host transfer on the SDMA engines (cae_copy_to_host).
//
// Why not hipMemcpyAsync: inside a PyTorch-ROCm 7.0 process the HIP runtime executes D2H copies with a blit
// KERNEL (__amd_rocclr_copyBuffer).  For the 100 MB of symbols of a 32-tile batch that kernel sits on every CU's
// wave slots for the 1.8 ms the data needs to cross PCIe, and the main stream's next kernel waited that long
// (profiles/r01_experiments.md, "D2H").  hsa_amd_memory_async_copy of the same runtime uses the DMA engines and
// leaves the CUs alone.
#include "cae_hip.h"
#include "cae_internal.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <mutex>

using namespace cae;

extern "C" int cae_copy_to_host(void *dst_host, const void *src_dev, size_t bytes) {
    if (!dst_host || !src_dev) return fail(CAE_ERR_ARG, "NULL argument");
    if (bytes == 0) return CAE_OK;
    static std::once_flag once;
    static hsa_status_t init_status = HSA_STATUS_SUCCESS;
    std::call_once(once, [] { init_status = hsa_init(); });  // reference-counted; HIP initialised it already
    if (init_status != HSA_STATUS_SUCCESS) return fail(CAE_ERR_HIP, "hsa_init failed (%d)", (int)init_status);

    hsa_amd_pointer_info_t si{}, di{};
    si.size = sizeof(si);
    di.size = sizeof(di);
    hsa_status_t st = hsa_amd_pointer_info(const_cast<void *>(src_dev), &si, nullptr, nullptr, nullptr);
    if (st != HSA_STATUS_SUCCESS || si.type != HSA_EXT_POINTER_TYPE_HSA)
        return fail(CAE_ERR_ARG, "source is not a device allocation of this process");
    st = hsa_amd_pointer_info(dst_host, &di, nullptr, nullptr, nullptr);
    if (st != HSA_STATUS_SUCCESS || (di.type != HSA_EXT_POINTER_TYPE_HSA && di.type != HSA_EXT_POINTER_TYPE_LOCKED))
        return fail(CAE_ERR_ARG, "destination must be pinned host memory (hipHostMalloc / a pinned tensor)");
    const char *s0 = (const char *)si.agentBaseAddress, *d0 = (const char *)di.agentBaseAddress;
    if ((const char *)src_dev + bytes > s0 + si.sizeInBytes || (const char *)dst_host + bytes > d0 + di.sizeInBytes)
        return fail(CAE_ERR_ARG, "copy of %zu bytes leaves its allocation", bytes);

    hsa_signal_t sig;
    st = hsa_signal_create(1, 0, nullptr, &sig);
    if (st != HSA_STATUS_SUCCESS) return fail(CAE_ERR_HIP, "hsa_signal_create failed (%d)", (int)st);
    st = hsa_amd_memory_async_copy(dst_host, di.agentOwner, src_dev, si.agentOwner, bytes, 0, nullptr, sig);
    if (st != HSA_STATUS_SUCCESS) {
        hsa_signal_destroy(sig);
        return fail(CAE_ERR_HIP, "hsa_amd_memory_async_copy failed (%d)", (int)st);
    }
    while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {
    }
    hsa_signal_destroy(sig);
    return CAE_OK;
}
