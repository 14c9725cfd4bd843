// The decoder tail's translation unit: dec_tail_kernel's instantiations and launch_dec_tail (see dec_tail.h, dec_tail_args.h).
// reference: module/espnet/test/Model.py:375-377, module/espnet/test/VisualizeResults_iou.py:128,151-155
#include "dec_tail.h"

namespace gs {
gs_status dec_tail_fault_flags(int *flags)
{
    int v = 0;
    GS_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_dec_tail_fault), sizeof(int)));
    if (v) {
        const int zero = 0;
        GS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dec_tail_fault), &zero, sizeof(int)));
    }
    *flags = v;
    return GS_OK;
}
}  // namespace gs
