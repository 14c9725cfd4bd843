// Host-side staging helpers shared by the host pipelines (gs_espnet_segment_host, gs_espnet_segment_crops_host,
// gs_detector_detect_host): the page-locked test; the threaded copies are host_jobs.h.
#pragma once
#include <hip/hip_runtime.h>

#include "host_jobs.h"

namespace gs {

// is `p` page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory)?  Such buffers are DMA'd in place.
static inline bool host_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// do the bytes [p, p + bytes) lie inside ONE page-locked allocation?  Two separately pinned buffers can be neighbours in virtual
// memory; a single DMA across both registrations is something this stack does not promise, so block copies are only
// issued for ranges this says yes to.
static inline bool host_block_is_pinned(const void *p, size_t bytes)
{
    if (!host_is_pinned(p))
        return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const char *b = static_cast<const char *>(base), *q = static_cast<const char *>(p);
    return q >= b && bytes <= size && (size_t)(q - b) <= size - bytes;
}

}  // namespace gs
