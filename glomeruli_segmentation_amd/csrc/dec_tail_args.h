// Decoder tail: the interface between the forward schedule (espnet.hip) and the kernel's own translation unit (dec_tail.hip;
// the kernel template and its launcher live in dec_tail.h).  A separate object file: the dozen instantiations of that kernel
// take as long to compile as the rest of the ESPNet forward, and the two now build side by side.
#pragma once
#include "gs_internal.h"

namespace gs {

struct DecTailArgs {
    // concat buffer: planes in torch.cat order, zero halo of one row / one column (Model.py:375)
    const float *in;
    long long in_sn;
    int in_sc, in_pitch, in_off;
    unsigned in_img_bytes;
    const float *wpack;   // [3 ty][6 plane groups][64 lanes] A operands | 16: BN scale, shift, alpha [3][5] | classifier.weight [5][5][2][2]
    float *logits;        // [N][CLS][2*H1][2*W1] or null
    unsigned char *mask;  // [N][2*H1][2*W1] or null
    unsigned long long *hist;   // [N][CLS] or null (zeroed by the caller)
    float *ff;            // optional: the CBR output (stage "conv") as a gs::Act
    long long ff_sn;
    int ff_sc, ff_pitch, ff_off;
    // ensemble (BASELINE cfg 5, definition in DESIGN.md): prob [N][CLS][2*H1][2*W1] accumulates ens_w * softmax(logits) over the
    // member models.  ens_mode 1: first member (store), 2: a middle member (add), 3: the last member (add, then argmax of the
    // sum -> mask + counts; nothing is written back), 4: a single member (softmax -> argmax, prob untouched).  0: no ensemble.
    float *prob;
    int ens_mode;
    float ens_w;
    int N, H1, W1;
    int xbase, nstrips;   // this launch covers strips of 16P-2 output columns starting at column xbase
    int bands, R, k3;     // bands of R = 3*k3 + 2 output rows
    int total_tasks;
    int wu;               // waves (exchanging form: teams) of a workgroup that take tasks (all of them unless there are fewer tasks than slots)
    int team;             // exchanging form (dec_tail.h, EXCH): waves per team = strips per row rounded up to a power of two
};

constexpr int DT_A_FLOATS = 18 * 64;
constexpr int DT_PACK_FLOATS = DT_A_FLOATS + 16 + 100;

// conv CBR(19+classes, classes, 3) + classifier deconvolution + argmax + counts in one or two launches (dec_tail.h)
gs_status launch_dec_tail(DecTailArgs a, int num_cus, hipStream_t stream);
// The exchanging form's mailbox waits are bounded; a wave that gives one up sets a device-side fault word instead of going on
// silently with values it did not receive.  Reads AND clears that word (the caller has synchronised the work it asks about);
// *flags != 0: results of dec_tail launches since the last call are not to be trusted.
gs_status dec_tail_fault_flags(int *flags);

}  // namespace gs
