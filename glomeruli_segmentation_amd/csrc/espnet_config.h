// The shipped configuration of the ESPNet forward, in one place: the template arguments of every conv_mfma_kernel
// instantiation (CFG_*), the build-time choices made by measurement (fusion, half-row tasks, lazy b2, ...) and the cache
// policy per launch class (POL_*).  Every macro is `#ifndef`-guarded so that an experiment build can override it from the
// command line (`python -m glomeruli_segmentation_amd.build --out variants_so/x.so -- -DCFG_...`); the defaults below are what
// ships, and the comment at each says what was measured (details: profiles/README.md).  Included by espnet.hip only.
#pragma once
#include "conv_mfma.h"

namespace gs {

// ------------------------------------------------------------------------------------------
// kernel configurations (template arguments of conv_mfma_kernel); see DESIGN.md "kernels"
//                       MT WAVES CINP TAPS STRIDE NDIL NOUT1 NOUT  P  G   (ring depth = G * taps-per-row)
#ifndef L2P
#define L2P 8
#endif
#define CFG_L2_C1S       16, 8,   20,  9,   2,     1,   12,   12,   L2P, 3
#define CFG_L2_C1        16, 8,   64,  1,   1,     1,   12,   12,   L2P, 8
#define CFG_L2_BR        16, 8,   12,  9,   1,     5,   16,   12,   L2P, 3
#ifndef L2SP
#define L2SP 4
#endif
#ifndef L2SG
#define L2SG 9
#endif
#ifndef L2W
#define L2W 8     // waves per workgroup of the level-2 branch kernels (4: two four-wave workgroups per CU, finer launch tails)
#endif
#define CFG_L2_BR_P4     16, L2W, 12,  9,   1,     5,   16,   12,   L2SP, L2SG   // shipped (4, 9): a whole dilation's operands in flight
// the level-2 down-sampler's branches with a chunk of ONE tap row (nine steps, a nine-step ring) and F_SKIP_PAD: 0.1868 -> 0.1764 ms
// (profiles/r06_ab_l2g3.txt); the ESP blocks lose with that shape (0.1948 -> 0.2024) and keep the whole-dilation chunk
#define CFG_L2_BR_P4S    16, L2W, 12,  9,   1,     5,   16,   12,   L2SP, 3
#ifndef CFG_L2_DOWN_SKIP
#define CFG_L2_DOWN_SKIP 1
#endif
#define CFG_L3_C1S       32, 8,   132, 9,   2,     1,   25,   25,   4, 6
#ifndef L3C1S_BNL_G
#define L3C1S_BNL_G 6     // F_BNLOAD form; a deeper ring measured no better (9: 0.192 ms vs 0.189) or spilled (11: 0.287)
#endif
#define CFG_L3_C1S_BNL   32, 8,   132, 9,   2,     1,   25,   25,   4, L3C1S_BNL_G
#define CFG_L3_C1        32, 8,   128, 1,   1,     1,   25,   25,   4, 16
#define CFG_L3_BR        32, 8,   26,  9,   1,     5,   28,   25,   4, 3
#define CFG_L3_BR_P2     32, 8,   26,  9,   1,     5,   28,   25,   2, 13
#define CFG_L3_BR_P2F    32, 8,   26,  9,   1,     5,   28,   25,   2, 3    // with the fused 1x1: 32 more accumulators
#ifndef L3WB
#define L3WB 8    // waves per workgroup of the shipped level-3 branch form (4 needs CFG_AGL_L3: two workgroups' weight images do not fit the LDS)
#endif
#define CFG_L3_BR_P2R    32, L3WB, 26, 9,   1,     5,   28,   25,   2, 13   // shipped fused ESP form: a third of a dilation in flight
// small batches (launches with fewer tasks than SIMDs): 32-pixel strips -- the same accumulation chain per pixel, four /
// two times the tasks (forward_impl picks the shape per launch from the task count; tools/latency.py)
#ifndef CFG_SMALL_AGL
#define CFG_SMALL_AGL 1   // the small-batch level-3 forms take their weights from L2 through the operand ring (F_A_GLOBAL: no LDS staging
                          // phase in front of a lone wave's task; one tile 0.492 -> 0.481 ms, 0.0358 -> 0.0347 ms per launch; at full batches
                          // the same flag LOSES 3-8 %, CFG_AGL_L3)
#endif
#ifndef CFG_SMALL2_WAVES
#define CFG_SMALL2_WAVES 4   // the same at level 2 (32-pixel strips, two pixels per lane; 0 = off): one tile 0.482 -> 0.471 ms; with 8 two
                           // tiles took 0.511 instead of 0.505, so only while the tasks are at most one per SIMD
#endif
#define CFG_L2_BR_P2S     16, 8,   12,  9,   1,     5,   16,   12,   2, 9
#ifndef CFG_SMALL3_WAVES
#define CFG_SMALL3_WAVES 8   // 32-pixel level-3 tasks while there are at most this many of them per CU (up to 8 tiles; with 4 -- one per
                           // SIMD, up to 4 tiles -- eight tiles took 0.904 ms instead of 0.873, six 0.856 instead of 0.817)
#endif
#define CFG_L3_BR_P1R    32, 8,   26,  9,   1,     5,   28,   25,   1, 13
// experiment (round 6): sixteen waves per workgroup = four per SIMD (<= 128 registers), 32-pixel tasks
#ifndef CFG_L3_W16
#define CFG_L3_W16 0      // bit 0: the fused ESP launches, bit 1: the last ESP launch, bit 2: the down-sampler
#endif
#ifndef L3W
#define L3W 16
#endif
#define CFG_L3_BR_W16    32, L3W, 26,  9,   1,     5,   28,   25,   1, 13
#define CFG_L3_C1S_BNL_P1 32, 8,  132, 9,   2,     1,   25,   25,   1, L3C1S_BNL_G
#define CFG_DEC_CONV     16, 8,   24,  9,   1,     1,   5,    5,    8, 3
#define CFG_DEC_CONV_XM  16, 8,   24,  3,   1,     1,   5,    5,    8, 6


// Build-time choices, each made by measurement (profiles/README.md); the defaults are what ships.
// F_FUSE1X1 (the next block's 1x1 reduce computed in a block's epilogue), measured at batch 32 (profiles/README.md):
//   level 2: down-sampler 0.223 -> 0.242 ms, ESP block 0.189 -> 0.21 ms (three waves per SIMD instead of four), against
//            0.063 ms per separate 1x1 launch: -0.084 ms per step.  On.
//   level 3: down-sampler (no residual: the second accumulator set fits beside four pixels per lane) 0.159 -> 0.175 ms
//            against 0.032 ms for the 1x1 launch: on.  ESP blocks: beside the residual registers the second accumulator
//            set only fits at two pixels per lane, and that form takes 0.1995 ms = exactly branch kernel + 1x1 kernel
//            (0.167 + 0.032); with the residual through a half-slot register ring (F_RES_RING) it fits at four pixels
//            per lane (24 registers spilled) and takes 0.190-0.197 ms.  Shipped since: two pixels per lane WITH that
//            register ring and a 39-step operand ring (CFG_L3_BR_P2R, 255 registers, no spill): 0.183 ms, because half-row
//            tasks halve the images an XCD has in flight and the reduced maps stay in its L2.  CFG_L3_FUSE_P4=1 selects
//            the four-pixel form.
#ifndef CFG_FUSE_L3
#define CFG_FUSE_L3 2   // 0 off, 1 down-sampler only, 2 every block
#endif
#ifndef CFG_FUSE_L2
#define CFG_FUSE_L2 1
#endif
#ifndef CFG_L3_FUSE_P4
#define CFG_L3_FUSE_P4 0
#endif
// The last (unfused) level-3 block in the half-row task shape of the fused ones (CFG_L3_BR_P2R + F_SKIP_PAD) instead of the
// whole-row four-pixel form: beyond-L2 fetch of that launch 528 -> 235 MB, step 2.853 -> 2.835 ms (profiles/README.md, round 3).
#ifndef CFG_L3_LAST_P2
#define CFG_L3_LAST_P2 1
#endif
// ... and the level-3 down-sampler's branches likewise: kernel time unchanged (0.1749 -> 0.1748 ms) but its beyond-L2 fetch
// 371 -> 106 MB, which the other lane's kernels feel: step 2.903 -> 2.878 ms with two batches in flight.
#ifndef CFG_L3_DOWN_P2
#define CFG_L3_DOWN_P2 1
#endif
// F_A_GLOBAL (weights from L2 through the operand ring, no LDS image): level-3 ESP block 0.167 -> 0.1715 ms, level-2
// blocks +6-10 %, stride-2 reduces +1-8 %: the 9 us staging phase it removes is cheaper than the slower loop.  Off.
#ifndef CFG_AGL_L3
#define CFG_AGL_L3 0
#endif
#ifndef CFG_AGL_L2
#define CFG_AGL_L2 0
#endif
#ifndef CFG_AGL_S2
#define CFG_AGL_S2 0    // ... and the stride-2 reduces
#endif
constexpr int AGL_L3 = CFG_AGL_L3 ? F_A_GLOBAL : 0, AGL_L2 = CFG_AGL_L2 ? F_A_GLOBAL : 0, AGL_S2 = CFG_AGL_S2 ? F_A_GLOBAL : 0;
constexpr int FUSE_L3 = CFG_FUSE_L3 ? F_FUSE1X1 : 0, FUSE_L2 = CFG_FUSE_L2 ? F_FUSE1X1 : 0;
// F_S2_FLIP (odd output rows of the stride-2 reduces walk their tap rows bottom-up, so neighbouring waves fetch the input
// row they share together): level 3 0.160 -> 0.153 ms, beyond-L2 fetch 910 -> 693 MB; level 2 0.0957 -> 0.0909 ms,
// 524 -> 430 MB.  On for both.
// The level-2 stride-2 reduce takes its images last to first: it re-reads the 319 MB the stem has just written, more than the
// 256 MB Infinity Cache holds, and the images the stem wrote LAST are the ones that may still be there.  Measured in
// profiles/r06_ab_l2_reduce.txt.
#ifndef CFG_L2_C1S_REV
#define CFG_L2_C1S_REV 0
#endif
#ifndef CFG_S2_FLIP
#define CFG_S2_FLIP 3   // bit 0: level-2 stride-2 reduce, bit 1: level-3
#endif
constexpr int S2FLIP_L2 = (CFG_S2_FLIP & 1) ? F_S2_FLIP : 0, S2FLIP_L3 = (CFG_S2_FLIP & 2) ? F_S2_FLIP : 0;
// F_SKIP_PAD (tap rows of a dilated branch that lie wholly in the zero halo are not multiplied): the fused level-3 ESP form,
// whose chunk is exactly one tap row.  Measured in profiles/README.md (round 3).
#ifndef CFG_SKIP_PAD
#define CFG_SKIP_PAD 1
#endif
constexpr int SKIP_L3 = CFG_SKIP_PAD ? F_SKIP_PAD : 0;
// ... and at level 2 (needs a chunk of one tap row there too: L2SG = 3; 3.2 % of the level-2 branch k-steps)
#ifndef CFG_SKIP_PAD_L2
#define CFG_SKIP_PAD_L2 0
#endif
constexpr int SKIP_L2 = CFG_SKIP_PAD_L2 ? F_SKIP_PAD : 0;
// Lazy b2: b2 = BR(131) over cat([output1, output1_0, inp2]) (Model.py:359) used to be fused into its producers, the
// down-sampler writing output1_0 TWICE (raw for the level-2 ESP blocks, b2-normalised into planes 64..127 of output1_cat).
// The normalised copy is now never written: its two consumers -- the level-3 stride-2 reduce (F_IN2: BN + PReLU on the B
// operands of those 64 channels) and dec2 -- read the raw output and apply b2 on load.  The store it saves is 268 MB per
// 32-tile step; dropping it outright (a timing-only build) moved the down-sampler 0.231 -> 0.186 ms and the two-lane step
// 2.896 -> 2.771 ms.  Built: down-sampler 0.228 -> 0.183 ms, the stride-2 reduce 0.156 -> 0.190 ms (its on-load BN + PReLU is
// interleaved with the matrix instructions but not free: that kernel's matrix pipe is 81 % busy), dec2 unchanged; one lane
// about even, two batches in flight 2.87 -> 2.81-2.82 ms per step (profiles/README.md, round 3).
#ifndef CFG_LAZY_B2
#define CFG_LAZY_B2 1
#endif


// F_EPI_PIPE (round 5): the level-2 branch kernels' epilogue of concat slot d between the k-steps of dilation d + 1 (conv_mfma.h).
// bit 0: the ESP blocks, bit 1: the down-sampler.  Measured in profiles/r05_ab_l2_epilogue.txt.
#ifndef CFG_L2_EPI_PIPE
#define CFG_L2_EPI_PIPE 0
#endif
constexpr int EPIPE_L2_ESP = (CFG_L2_EPI_PIPE & 1) ? F_EPI_PIPE : 0, EPIPE_L2_DOWN = (CFG_L2_EPI_PIPE & 2) ? F_EPI_PIPE : 0;

// cache-policy flags per launch class (F_RES_NT / F_ST_NT / F_ST2_NT, conv_mfma.h)
#ifndef POL_L2_DOWN
#define POL_L2_DOWN (F_ST_NT | F_ST2_NT)
#endif
#ifndef CFG_L2_XFLAGS
#define CFG_L2_XFLAGS 0   // -DGS_DIAG ablation builds: F_X_NOEPI / F_X_NOLOAD / F_X_NOLDS ORed into the level-2 ESP launches
#endif
#ifndef POL_L2_ESP
#define POL_L2_ESP (F_RES_NT | F_ST_NT | CFG_L2_XFLAGS)
#endif
#ifndef POL_L2_LAST
#define POL_L2_LAST (F_RES_NT | F_ST2_NT)
#endif
#ifndef POL_L3_DOWN
#define POL_L3_DOWN 0
#endif
#ifndef CFG_L3_XFLAGS
#define CFG_L3_XFLAGS 0   // -DGS_DIAG ablation builds: F_X_NOEPI / F_X_NOLOAD / F_X_NOLDS / F_X_NOEPIMEM ORed into the level-3 ESP launches
#endif
#ifndef POL_L3_ESP
#define POL_L3_ESP (F_RES_NT | CFG_L3_XFLAGS)
#endif
#ifndef POL_L2_C1
#define POL_L2_C1 0
#endif
#ifndef POL_L3_C1
#define POL_L3_C1 0
#endif
#ifndef CFG_L2C1S_XFLAGS
#define CFG_L2C1S_XFLAGS 0   // -DGS_DIAG: F_X_NOLOAD on the level-2 stride-2 reduce = that launch without its 319 MB re-read of output0_cat
#endif
#ifndef POL_L2_C1S
#define POL_L2_C1S CFG_L2C1S_XFLAGS
#endif
#ifndef POL_L3_C1S
#define POL_L3_C1S F_IN_NT   // 0.179 -> 0.165 ms; the same flag on the other inputs lost (decoder conv 0.19 -> 0.41: it lives on L2 row reuse)
#endif
#ifndef POL_DEC_CONV
#define POL_DEC_CONV 0
#endif


}  // namespace gs
