// The decoder tail's translation unit: dec_tail_kernel's instantiations and launch_dec_tail (see dec_tail.h, dec_tail_args.h).
// reference: module/espnet/test/Model.py:375-377, module/espnet/test/VisualizeResults_iou.py:128,151-155
#include "dec_tail.h"
