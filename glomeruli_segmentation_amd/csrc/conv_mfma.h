// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).
//
// One kernel template covers every dense contraction of the ESPNet trunk:
//   * C / CDilated 1x1 and strided 3x3 "reduce" convolutions        (reference Model.py:98-104,135,178)
//   * the five dilated 3x3 branches of DownSamplerB / the ESP block, their hierarchical
//     feature fusion, channel concat, residual add, BatchNorm and PReLU (Model.py:144-160,187-214)
//
// Mapping.  GEMM-N (the MFMA column / the lane) is a run of MT consecutive output pixels of one
// row, GEMM-M (the MFMA row) is the output channel, GEMM-K walks (tap, input channel).  A wave owns
// P such pixel runs (P*MT consecutive pixels) and all output channels of them, so:
//   * the B operand of a k-step is MT consecutive floats of an input row: one coalesced
//     buffer_load_dword per pixel run straight from L1/L2 -- an fp32 MFMA needs only 4 B per lane
//     per operand per 32-64 cycles, so no LDS staging of activations is needed; the zero halo of
//     the activation buffers (gs::Act) makes every tap unconditional;
//   * the A operand (weights, pre-packed [dilation][tap][cin][NROW] on the host) comes from LDS,
//     one conflict-free ds_read_b32 per k-step shared by the P MFMAs of that step;
//   * the fusion add2 = add1 + d4, add3 = add2 + d8, ... (Model.py:152-155) costs nothing: the
//     d4/d8/d16 branches simply keep accumulating into the d2 accumulator, which is written out
//     after each branch;
//   * the lane<->pixel, register<->channel accumulator layout stores 128-byte (MT=32) or 64-byte
//     (MT=16) row segments per channel plane.
#pragma once
#include <map>
#include <mutex>
#include <type_traits>

#include "gs_internal.h"

namespace gs {

// Timing / ablation variants (results wrong by construction), per-wave stamps and the launch-time environment knobs
// exist only in builds made with -DGS_DIAG; the product library contains none of them and reads no environment.
#ifdef GS_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in that order, every index a compile-time constant
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT>
struct Mfma;
template <>
struct Mfma<32> {
    static constexpr int KL = 2, NACC = 16;
    using acc_t = f32x16;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // output row held by accumulator register r of a lane in k-group kq (= lane / 32)
    static __device__ __forceinline__ int row(int r, int kq) { return (r & 3) + 8 * (r >> 2) + 4 * kq; }
};
template <>
struct Mfma<16> {
    static constexpr int KL = 4, NACC = 4;
    using acc_t = f32x4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int kq) { return kq * 4 + r; }
};

struct ConvArgs {
    // input activation (zero halo wide enough for every tap)
    const float *in;
    long long in_sn;
    int in_sc, in_pitch, in_off;
    unsigned in_img_bytes;
    // packed weights (+ folded BN scale/shift/alpha appended), staged to LDS once per workgroup
    const float *wpack;
    int wfloats;
    // output, optional residual
    float *out;
    long long out_sn;
    int out_sc, out_pitch, out_off;
    unsigned out_img_bytes, res_img_bytes;
    const float *res;
    long long res_sn;
    int res_sc, res_pitch, res_off;
    // optional second output (F_DUAL): channel c of this kernel lands in plane out2_coff + c
    float *out2;
    long long out2_sn;
    int out2_sc, out2_pitch, out2_off, out2_coff;
    unsigned out2_img_bytes;
    // F_FUSE1X1: the NEXT block's 1x1 reduce (Model.py:193 `output1 = self.c1(input)`) of this kernel's output
    float *out3;
    long long out3_sn;
    int out3_sc, out3_pitch, out3_off, nout3;
    unsigned out3_img_bytes;
    // F_BNLOAD (stride-2 reduce): a per-input-channel BN + PReLU is applied to the B operands as they are consumed -- the
    // consumer side of a "BR over a torch.cat" (Model.py:359) whose producer then stores its RAW output into the concat
    // buffer and nothing else.  The table [scale | shift | alpha][CINP + KL] follows the weight image in the blob: identity
    // for channels that arrive normalised, and a last all-zero slot.  k-groups [bnl_s0, bnl_s1) hold the raw channels: BN of
    // their zero halo would not be zero, so a tap row above the image takes the zero slot and column -1 is re-zeroed.
    int bnl_s0, bnl_s1;
    int lds_tile_off;   // F_XMERGE: float offset of the per-wave LDS tiles (after the weight image)
    // GS_DIAG builds only:
    int stagger;   // units of 1024 cycles by which waves WAVES/2.. start late (0 = off)
    int prio_mode; // wave priority of the two halves of a workgroup: 0 alternates per dilation, 1 per task (shipped), 2 off, 3 fixed
    unsigned long long *stamp;   // F_X_STAMP: [wave][8] 100 MHz timestamps
    int N, H, W;   // OUTPUT size
    int strips;    // pixel strips per output row
    int total_tasks;
    int wu;        // waves of a workgroup that take tasks (launch_conv_mfma: WAVES unless the launch has fewer tasks than wave slots)
    int rev_n;     // 1: images are taken last to first (the launch reads what its producer wrote most recently first)
};

typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// N consecutive floats through one (N = 2, 4) or two (N = 8) buffer instructions.  Elements are copied out
// before the bit cast: __builtin_bit_cast applied directly to v[j] reads element 0 with this hipcc.
template <int N, int AUX = 0>
__device__ __forceinline__ void buf_load_vec(const __amdgpu_buffer_rsrc_t &rs, int voff, int soff, float *dst)
{
    if constexpr (N == 1) {   // (only instantiated, never run: one-pixel-per-lane forms do not use the vector mapping)
        dst[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX));
    } else if constexpr (N == 2) {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
        const unsigned e0 = v[0], e1 = v[1];
        dst[0] = __builtin_bit_cast(float, e0);
        dst[1] = __builtin_bit_cast(float, e1);
    } else {
        static_assert(N == 4 || N == 8, "vector width");
#pragma unroll
        for (int h = 0; h < N / 4; ++h) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * h, soff, AUX);
            const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
            dst[4 * h + 0] = __builtin_bit_cast(float, e0);
            dst[4 * h + 1] = __builtin_bit_cast(float, e1);
            dst[4 * h + 2] = __builtin_bit_cast(float, e2);
            dst[4 * h + 3] = __builtin_bit_cast(float, e3);
        }
    }
}
// The uniform part of the address is added into the per-lane offset and the instruction's soffset is the
// constant 0: with an SGPR soffset hipcc assumes that a >64-bit buffer store has no write-data hazard and lets
// the next VALU instruction overwrite the data registers, and on gfx950 the store then sometimes writes the new
// register contents (seen as lane offsets appearing in the output).
template <int N, int AUX = 0>
__device__ __forceinline__ void buf_store_vec(const __amdgpu_buffer_rsrc_t &rs, int voff, const float *src)
{
    if constexpr (N == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, src[0]), rs, voff, 0, AUX);
    } else if constexpr (N == 2) {
        u32x2_t v;
        v[0] = __builtin_bit_cast(unsigned, src[0]);
        v[1] = __builtin_bit_cast(unsigned, src[1]);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, 0, AUX);
    } else {
#pragma unroll
        for (int h = 0; h < N / 4; ++h) {
            u32x4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = __builtin_bit_cast(unsigned, src[4 * h + j]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + 16 * h, 0, AUX);
        }
    }
}

// PReLU in two VALU instructions: for slope a <= 1 (negative slopes included) PReLU(v) = max(v, a*v), for a > 1 it is
// min(v, a*v); both are the median of (v, a*v, pin) with pin = +inf resp. -inf, a per-channel constant.  Bit-identical
// to the select form (the same a*v, the same v).  Next to fp32 MFMAs every VALU instruction of the epilogue costs
// the issuing wave about one MFMA slot, so the compare / select pair it replaces was worth removing.
__device__ __forceinline__ float prelu_pin(float alpha)
{
    return alpha <= 1.0f ? __builtin_inff() : -__builtin_inff();
}
__device__ __forceinline__ float prelu_med3(float v, float alpha, float pin)
{
    return __builtin_amdgcn_fmed3f(v, alpha * v, pin);
}

// epilogue flags
constexpr int F_BNACT = 1;   // folded BatchNorm + PReLU on the way out
constexpr int F_RES = 2;     // add the residual input before BN (ESP block, Model.py:211-213)
constexpr int F_NOSTORE = 4; // skip the primary store (the block output is only consumed through out2)
constexpr int F_DUAL = 8;     // second store into a concat buffer through a second BN+PReLU: the b2 / b3
                             // "BR over a torch.cat" stages (Model.py:359) fused into the producers
constexpr int F_XMERGE = 256;   // TAPS == 3 only: the three horizontal taps are folded into the MFMA rows (see kernel)
constexpr int F_S2PAIR = 512;   // STRIDE == 2, TAPS == 9: the three horizontal taps of an output pixel (inputs 2x-1..2x+1)
                                // come from one 12-byte load instead of three stride-2 dword loads, which cost the
                                // texture-address unit 16 cycles each (4 for a unit-stride one)
constexpr int F_VEC = 1024;  // lane j owns P CONSECUTIVE pixels (instead of pixel j of P runs): operands, residual and
                             // results move as 8/16-byte accesses.  Needs STRIDE == 1, W % P == 0, P in {2, 4, 8}.
// Cache policy of the once-only streams (aux bit 1 = nt, "non-temporal"): the residual is read once and the results are
// read by a later kernel, so neither should displace the reduced map the taps re-read, and results written nt are not
// left dirty in L2 for the kernel boundary to flush.  Measured per kernel (profiles/README.md); not used where it lost.
constexpr int F_RES_NT = 8192;    // residual loads
constexpr int F_ST_NT = 16384;    // result stores (first output)
constexpr int F_ST2_NT = 32768;   // F_DUAL second output
constexpr int F_IN_NT = 65536;    // activation (B operand) loads: inputs that are read once (1x1 reduces)
// Weights (A operand) straight from L2 through the operand ring instead of an LDS image: no staging phase (the 131 KB
// image of a level-3 branch kernel took ~9 us of a ~165 us launch to fill, with every matrix pipe idle), no barrier, and
// only the BN / PReLU parameters (+ the F_FUSE1X1 table) remain in LDS.  Every wave streams the same image in the same
// order, so all but the first touch of a line hit the XCD's L2.
constexpr int F_A_GLOBAL = 131072;
// The NEXT block's 1x1 reduce computed in this kernel's epilogue: a finished output register (channel row r of the two
// k-groups, one value per pixel column) IS the B operand of a k-step of `c1(out)`, so one extra MFMA per register
// accumulates W_c1[:, ch] * out[ch] into a second accumulator set that is stored to the next block's reduced map when the
// task ends.  Removes the separate 1x1 kernel and its re-read of the whole block output.
constexpr int F_FUSE1X1 = 262144;
constexpr int F_RES_RING = 524288;   // residual values through a half-slot register ring (see the kernel)
// Stride-2 3x3 reduces: output rows y and y+1 share input row 2y+1.  Odd output rows walk their three tap rows bottom-up,
// so that neighbouring tasks (neighbouring waves of one workgroup) ask for the shared row at the same moment -- the end of
// the even row's k-loop, and of the odd row's -- instead of a whole task apart, when it has long left the caches.
constexpr int F_S2_FLIP = 1048576;
// Dilated 3x3 branches: the tap row ty of dilation d reads input row y + (ty-1)*d, which for rows within d of the top /
// bottom edge lies wholly in the zero halo -- a third of that dilation's matrix work multiplying zeros (over a 64-row
// level-3 map: d16 half of the rows, d8 a quarter, ...: 6.5 % of the branch k-steps).  When a chunk is exactly one tap row
// (G == NSTEP) such a chunk is skipped (wave-uniform), and the operand ring is refilled with the next LIVE chunk instead.
constexpr int F_SKIP_PAD = 4194304;
constexpr int F_BNLOAD = 8388608;  // see ConvArgs::bnl_s0
// Branch kernels whose chunk is a whole dilation (level 2): the epilogue of concat slot d -- residual add, BN, PReLU, stores, the
// fused 1x1's matrix instructions -- is not run in one piece between two dilations (a stretch in which this wave feeds the matrix
// pipe nothing) but register by register BETWEEN the k-steps of dilation d + 1, from a snapshot of the accumulator (HFF keeps
// adding into the accumulator itself).  The last slot of a task still runs in one piece.
constexpr int F_EPI_PIPE = 16777216;
constexpr int F_X_NOLOAD = 16;  // GS_DIAG timing experiments only (results are garbage): no activation loads in the loop
constexpr int F_X_NOLDS = 32;   // GS_DIAG: no LDS weight reads in the loop
constexpr int F_X_NOEPI = 64;   // GS_DIAG: no epilogue at all
constexpr int F_X_STAMP2 = 4096;   // GS_DIAG: [wave][STAMP2_SLOTS] stamps of EVERY task of the wave: 0 kernel start, 1 weights staged;
                                   // task ti from slot 2 + ti * 2 * NCHUNK: + 2c after the MFMA steps of chunk c, + 2c + 1 after the
                                   // epilogue that follows chunk c (tools/stamps3.py)
constexpr int STAMP2_SLOTS = 128;
constexpr int F_X_STAMP = 128;  // GS_DIAG: per-wave s_memrealtime stamps into a.stamp (start, staged, per-dilation, end)
constexpr int F_X_NOEPIMEM = 2097152;   // GS_DIAG: the epilogue's arithmetic (and fused MFMAs) but none of its residual loads / output stores
constexpr int F_X_RESL2 = 33554432;     // GS_DIAG: the residual loads fall into a wave-private 8 KiB window of the image (cache hits): what their LATENCY costs
constexpr int F_X_STL2 = 67108864;      // GS_DIAG: likewise the result stores
constexpr int F_X_ALL = F_X_NOLOAD | F_X_NOLDS | F_X_NOEPI | F_X_STAMP | F_X_STAMP2 | F_X_NOEPIMEM | F_X_RESL2 | F_X_STL2;

// Float layout of a configuration's packed image in the weight blob: [weights NDIL*TAPS*CINP*NROW | BN scale, shift,
// alpha (3*COUT, twice with F_DUAL) | F_FUSE1X1 table NDIL*NACC*64], rounded up to whole float4s.
struct ConvImage {
    int nrow, cout, w, bn, tab, total;
};
constexpr ConvImage conv_image(int CINP, int TAPS, int NDIL, int NOUT1, int NOUT, bool bn, bool dual = false, bool xmerge = false,
                               int fuse_nacc = 0)
{
    ConvImage im{};
    im.nrow = xmerge ? 3 * NOUT1 : (NOUT1 > NOUT ? NOUT1 : NOUT);
    im.cout = NOUT1 + (NDIL - 1) * NOUT;
    im.w = NDIL * TAPS * CINP * im.nrow;
    im.bn = (bn ? 3 * im.cout : 0) + (dual ? 3 * im.cout : 0);
    im.tab = NDIL * fuse_nacc * 64;
    im.total = (im.w + im.bn + im.tab + 3) / 4 * 4;
    return im;
}

// Waves per SIMD the register allocator must leave room for in the level-2 branch kernels (16x16x4, four pixels per
// lane).  Round 1 shipped a 9-step operand ring at four waves per SIMD (<= 128 VGPRs); with the fused 1x1 those forms
// need ~135.  Measured at batch 32 (profiles/README.md): the 27-step ring (G = 9: a whole dilation in flight, ~200-240
// VGPRs, two waves per SIMD) runs the down-sampler in 0.239 ms against 0.246 and the fused ESP block in 0.1996 against
// 0.204; forcing four waves onto the fused forms spills and loses 5 %.  So: no floor (1) and the deep ring.
#ifndef CFG_L2_MINW
#define CFG_L2_MINW 1
#endif
// F_RES_RING: the register ring holds 1 / CFG_RES_RING_DIV of a slot's residual values (2: 24 registers spilled in the
// fused level-3 ESP kernel, 0.190 ms; 4: no spill but the residual latency shows, 0.192 ms)
#ifndef CFG_RES_RING_DIV
#define CFG_RES_RING_DIV 1   // round 6: with the 13-step operand ring (CFG_L3_RING) a whole slot's residual fits (196 registers): requested at
                             // the top of its dilation (CFG_RES_TOP), no refill inside the epilogue
#endif
// ---- round-6 switches of the branch kernels' epilogue and k-step; each measured interleaved on one box with the same bits
// (profiles/README.md, "Round 6", section 3); the defaults are what ships
#ifndef CFG_RES_TOP
#define CFG_RES_TOP 1       // a slot's residual requested at the top of its own dilation: level-2 ESP launch 0.1975 -> 0.194 ms (r06_ab_combo.txt)
#endif
#ifndef CFG_EPI_PRIO
#define CFG_EPI_PRIO 0      // wave priority inside the epilogue (2 or 3 = above both task priorities): level-3 ESP -2 % alone, +0.5 % beside the
                            // 13-step ring (r06_ab_epiprio.txt, r06_ab_combo.txt): off
#endif
#ifndef CFG_EPI_SPLIT
#define CFG_EPI_SPLIT 0     // the fused 1x1's matrix instructions in one piece behind the slot's arithmetic: 0.1850 -> 0.1878 ms (r06_ab_sp2.txt): off
#endif
#ifndef CFG_EPI_PRELOAD
#define CFG_EPI_PRELOAD 0   // 16x16x4 forms: a slot's BN constants all read up front: ESP unchanged, down-sampler 0.182 -> 0.196 (r06_ab_epi_preload.txt): off
#endif
#ifndef CFG_REFILL_MID
#define CFG_REFILL_MID 0    // ring refill between the matrix instructions of the next step: level-3 ESP 0.1805 -> 0.1788, step +-0.3 % (r06_ab_refill_mid.txt): off
#endif
#ifndef CFG_X_EPI
#define CFG_X_EPI 0   // GS_DIAG timing experiments on the epilogue's memory instructions (results wrong): 1 no residual loads, 2 no result
                      // stores, 4 residual loads of the even registers only, 8 result stores of the even registers only
#endif
#ifndef CFG_STAGE_ROT
#define CFG_STAGE_ROT 17   // 0 = every workgroup stages the weight image in the same order
#endif
// Depth of the operand ring in k-steps.  A chunk (the unrolled unit, D = G * taps-per-row steps) and the ring used to be the
// same thing; the ring may be any divisor of the chunk: step u of a chunk lives in slot u % R and is refilled with step u + R
// (of this chunk, or of the next one).  A shallower ring gives registers back (level 3, two pixels per lane: 3 per step).
#ifndef CFG_L3_RING
#define CFG_L3_RING 13   // round 6 (profiles/r06_ab_ring.txt): 3, 13 and 39 steps measure within 1 % of each other once the waits are the
                         // compiler's exact ones (see the staging barrier below); 13 leaves room for the whole-slot residual prefetch
#endif
#ifndef CFG_L2_RING
#define CFG_L2_RING 27
#endif
constexpr int conv_ring_depth(int MT, int TAPS, int NDIL, int P, int G, int FLAGS)
{
    return (MT == 32 && TAPS == 9 && NDIL == 5 && (P == 2 || (P == 1 && CFG_L3_RING != 39)) && G == 13) ? CFG_L3_RING
           : (MT == 16 && TAPS == 9 && NDIL == 5 && P == 4 && G == 9) ? CFG_L2_RING
                                                                    : G * (TAPS == 9 ? 3 : 1);
}
constexpr int conv_min_waves(int MT, int TAPS, int NDIL, int P, int FLAGS)
{
    return (MT == 16 && TAPS == 9 && NDIL == 5 && P == 4) ? CFG_L2_MINW : 1;
}

#define M_KL_OF(MT_) (Mfma<MT_>::KL)
template <int MT, int WAVES, int CINP, int TAPS, int STRIDE, int NDIL, int NOUT1, int NOUT, int P, int G, int FLAGS>
__global__ void __launch_bounds__(WAVES * 64, WAVES >= 8 ? conv_min_waves(MT, TAPS, NDIL, P, FLAGS) * WAVES / 8 : 2) conv_mfma_kernel(const ConvArgs a)
{
    constexpr bool BNACT = FLAGS & F_BNACT, RES = FLAGS & F_RES, STORE1 = !(FLAGS & F_NOSTORE), DUAL = FLAGS & F_DUAL;
    constexpr bool S2P = FLAGS & F_S2PAIR;
    constexpr bool XMERGE_ = FLAGS & F_XMERGE;
    constexpr bool VEC = FLAGS & F_VEC;
    constexpr bool AGL = FLAGS & F_A_GLOBAL, FUSE = FLAGS & F_FUSE1X1;
    constexpr bool S2FLIP = FLAGS & F_S2_FLIP;
    constexpr bool SKIP = FLAGS & F_SKIP_PAD;
    constexpr bool BNL = FLAGS & F_BNLOAD;
    constexpr bool EPI_PIPED = FLAGS & F_EPI_PIPE;
    static_assert(!BNL || ((FLAGS & F_S2PAIR) && !(FLAGS & (F_BNACT | F_A_GLOBAL | F_VEC))), "F_BNLOAD is for the plain stride-2 reduce");
    static_assert(!SKIP || (TAPS == 9 && STRIDE == 1 && !(FLAGS & (F_S2PAIR | F_XMERGE)) && G * M_KL_OF(MT) == CINP),
                  "F_SKIP_PAD: unit-stride 3x3 with one tap row per chunk");
    static_assert(!S2FLIP || (STRIDE == 2 && TAPS == 9 && NDIL == 1), "F_S2_FLIP is for the stride-2 3x3 reduce");
    static_assert(kDiag || !(FLAGS & F_X_ALL), "timing / stamp variants exist in -DGS_DIAG builds only");
    constexpr int IAUX = (FLAGS & F_IN_NT) ? 2 : 0;
    constexpr int RAUX = (FLAGS & F_RES_NT) ? 2 : 0, SAUX = (FLAGS & F_ST_NT) ? 2 : 0, SAUX2 = (FLAGS & F_ST2_NT) ? 2 : 0;
    static_assert(!VEC || (STRIDE == 1 && (P == 2 || P == 4 || P == 8)), "F_VEC needs unit stride and P in {2,4,8}");
    static_assert(!S2P || (STRIDE == 2 && TAPS == 9 && NDIL == 1), "F_S2PAIR is for the stride-2 3x3 reduce");
    using M = Mfma<MT>;
    constexpr int KL = M::KL;
    constexpr int NSTEP = CINP / KL;
    constexpr int NROW = (FLAGS & F_XMERGE) ? 3 * NOUT1 : (NOUT1 > NOUT ? NOUT1 : NOUT);   // MFMA rows in use
    constexpr int COUT = NOUT1 + (NDIL - 1) * NOUT;
    // TAPS == 3 (with F_XMERGE) is a 3x3 convolution whose few output channels o and three horizontal taps tx
    // are BOTH put on the MFMA rows (row = tx*NOUT1 + o): the k-loop then walks only (tap row, channel) -- a third
    // of the k-steps, and 3*NOUT1 of MT rows busy instead of NOUT1 -- and the epilogue adds the three row groups
    // of a pixel's neighbours through a per-wave LDS tile.  Strips overlap by two columns.
    constexpr bool XMERGE = FLAGS & F_XMERGE;
    constexpr int TYN = TAPS == 1 ? 1 : 3;
    constexpr int TXN = TAPS == 9 ? 3 : 1;
    constexpr int XSTEP = XMERGE ? P * MT - 2 : P * MT;   // output pixels a strip produces
    // k-steps of one dilation in the order (row group rg = ty*NSTEP + cin-group, tx): tx fastest.  A
    // chunk is G consecutive row groups x all TXN horizontal taps = D steps, so inside a chunk the tap
    // and slot of every step are compile-time constants and only G (ty, cin-group) pairs are decoded
    // on the scalar unit per chunk (decoding every step cost ~40 SALU instructions per 4 MFMAs).
    constexpr int RGN = TYN * NSTEP;          // row groups per dilation
    constexpr int D = G * TXN;                // steps per chunk
    constexpr int R = conv_ring_depth(MT, TAPS, NDIL, P, G, FLAGS);   // ring depth
    // REFILL_MID: the refill of a ring slot is issued BETWEEN the matrix instructions of the NEXT step (behind the first half of
    // them) instead of behind all of its own step's: a wave issues in order, so `mfma, mfma, load, ds_read, scalar work` leaves
    // everything but the matrix instructions to the one window behind the second of them; with the refill in the middle both windows
    // are used.  The slot refilled is the previous step's (its own step's B registers are still to be read), so the ring holds R - 1
    // steps ahead.
    constexpr bool MID = CFG_REFILL_MID && TAPS == 9 && NDIL == 5 && P >= 2 && !(FLAGS & (F_S2PAIR | F_BNLOAD | F_EPI_PIPE));
    constexpr int RB = MID ? 1 : 0;   // steps by which the refill lags
    constexpr int RA = R;   // (ring depth of the A operands)
    static_assert(D % R == 0 && (R == D || !(FLAGS & (F_S2PAIR | F_BNLOAD | F_EPI_PIPE))), "the ring divides the chunk");
    constexpr int CPD = RGN / G;              // chunks per dilation
    constexpr int NCHUNK = NDIL * CPD;
    static_assert(CINP % KL == 0, "k-steps must tile");
    static_assert(TAPS == 1 || TAPS == 9 || (TAPS == 3 && XMERGE && NDIL == 1 && STRIDE == 1 && 3 * NOUT1 <= MT), "1x1, 3x3 or row-merged 3x3");
    static_assert(RGN % G == 0, "chunk must divide the row groups of one dilation");
    static_assert(!EPI_PIPED || (CPD == 1 && NDIL > 1 && TAPS == 9 && (FLAGS & F_BNACT) && !(FLAGS & (F_XMERGE | F_SKIP_PAD | F_RES_RING)) && D >= 4 * Mfma<MT>::NACC),
                  "F_EPI_PIPE: branch kernels whose chunk is one dilation, with a k-step to spare per accumulator register");
    static_assert(NROW <= MT, "one MFMA row block");
    constexpr int KSTR = 4;   // accumulator rows of k-group kq sit KSTR*kq above those of group 0 (both shapes)

    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    unsigned long long tstamp0 = 0;
    if (FLAGS & (F_X_STAMP | F_X_STAMP2))
        tstamp0 = __builtin_amdgcn_s_memrealtime();
    // LDS image = the blob image (conv_image) from float LDS_SRC0 on: everything, or with F_A_GLOBAL only what follows
    // the weights
    constexpr int WFL = NDIL * TAPS * CINP * NROW;
    constexpr int LDS_SRC0 = AGL ? WFL : 0;
    static_assert(!AGL || WFL % 4 == 0, "LDS-DMA source must stay 16-byte aligned");
    static_assert(!FUSE || (TAPS == 9 && !XMERGE_ && BNACT), "the fused 1x1 follows a branch kernel's epilogue");
    const float *bnp = lds + (WFL - LDS_SRC0);   // [scale | shift | alpha][COUT] (x2 with F_DUAL)
    constexpr int BNFL = (BNACT ? 3 * COUT : 0) + (DUAL ? 3 * COUT : 0);
    const float *tab = bnp + BNFL;               // F_FUSE1X1: [NDIL][NACC][64] A operands of the next block's 1x1
    // F_BNLOAD: [scale | shift | alpha][BNL_C] right after the (rounded) image
    constexpr int BNL_C = CINP + M::KL;
    const float *bnl = lds + (conv_image(CINP, TAPS, NDIL, NOUT1, NOUT, BNACT, DUAL, XMERGE_, FUSE ? M::NACC : 0).total - LDS_SRC0);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpack), 0, AGL ? WFL * 4 : 0, 0x00020000);

    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane % MT, kq = lane / MT;
    // pixel owned by (lane, p): x0 + p*MT + px (run form) or x0 + px*P + p (F_VEC: P consecutive pixels per lane)
    const int xl = VEC ? px * P : px;
    const int voff = (kq * a.in_sc + xl * STRIDE) * 4;
    const int lbase = kq * NROW + (px < NROW ? px : NROW - 1);
    const int vout = (kq * KSTR * a.out_sc + xl) * 4;
    const int vres = RES ? (kq * KSTR * a.res_sc + xl) * 4 : 0;
    const int vout2 = DUAL ? (kq * KSTR * a.out2_sc + xl) * 4 : 0;
    const int vout3 = FUSE ? (kq * KSTR * a.out3_sc + xl) * 4 : 0;
    const int prio_mode = kDiag ? a.prio_mode : 1;
    // the two waves that share a SIMD (their priorities alternate): waves wid and wid + 4 of an eight-wave workgroup; with
    // four-wave workgroups (two per CU) the SIMD's other wave belongs to the workgroup dispatched half a grid later
    const bool second_of_simd = WAVES >= 8 ? wid >= WAVES / 2 : blockIdx.x >= gridDim.x / 2;

    // Task order is (image, row, strip).  Workgroups that share an XCD (equal blockIdx % 8 under
    // round-robin dispatch: a speed assumption only) own one contiguous eighth of the tasks, and inside
    // it the XCD's waves sweep together: wave j takes tasks j, j + waves_per_xcd, ...  At any moment
    // the waves of one XCD therefore work on neighbouring rows of the same image(s), so the taps
    // they share stay in that XCD's 4 MiB L2 (with one contiguous range per wave every image of the
    // eighth was live at once and half of the B-operand L2 requests went on to the Infinity Cache).
    // A launch with fewer tasks than wave slots (small batches) is SPREAD: only the first a.wu waves of a workgroup take tasks --
    // waves 0..3 sit on four different SIMDs -- and the grid covers as many CUs as there are tasks for, so that a task has a
    // matrix pipe to itself instead of sharing it while three quarters of the chip idle.  The spare waves only help stage.
    const int NB = gridDim.x;
    const int wu = a.wu;
    int t0, t1, tstride;
    if ((NB & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = NB >> 3;
        const int c0 = (int)((long long)a.total_tasks * xcd / 8), c1 = (int)((long long)a.total_tasks * (xcd + 1) / 8);
        t0 = c0 + (blockIdx.x >> 3) * wu + wid;
        t1 = c1;
        tstride = per * wu;
    } else {
        t0 = blockIdx.x * wu + wid;
        t1 = a.total_tasks;
        tstride = NB * wu;
    }
    if (wid >= wu)
        t0 = t1;
    const long long wg = (long long)blockIdx.x * WAVES + wid;   // global wave id (diagnostic stamps)
    const int tasks_per_img = a.H * a.strips;
    bool staged = false;

    // operand ring (see below); it lives across tasks: the last chunk of a task refills it with the first
    // chunk of the wave's NEXT task, so only a wave's very first loads are exposed
    float aq[RA], bq[S2P ? 1 : R][P];
    float bl[S2P ? G : 1][P][3];   // F_S2PAIR: per row group, the inputs 2x-1, 2x, 2x+1 of every pixel

    for (int task = t0; task < t1 || !staged; task += tstride) {
        const bool idle = task >= t1;   // a wave without work still has to help stage the weights
        const int tk = idle ? (a.total_tasks - 1) : task;
        const int n0_ = tk / tasks_per_img;
        const int n = a.rev_n ? a.N - 1 - n0_ : n0_;
        const int rem = tk - n0_ * tasks_per_img;
        // F_SKIP_PAD: rows near the top / bottom edge skip tap rows and finish early, and a wave's tasks are the SAME row of
        // images `img_stride` apart -- so every other group of images has its rows rotated by half the height (a bijection
        // per image): a wave then owns one edge row and one middle row, and the saving is spread over all waves
        const int img_stride = tstride / tasks_per_img > 0 ? tstride / tasks_per_img : 1;
        const int yrot = SKIP ? a.H / 2 : 0;
        const int y0r = rem / a.strips;
        const int y = SKIP && ((n / img_stride) & 1) ? (y0r + yrot >= a.H ? y0r + yrot - a.H : y0r + yrot) : y0r;
        const int x0 = (rem - y0r * a.strips) * XSTEP;

        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.in + (long long)n * a.in_sn), 0, a.in_img_bytes, 0x00020000);
        const int sbase = (a.in_off + y * STRIDE * a.in_pitch + x0 * STRIDE - (XMERGE ? 1 : 0)) * 4;
        // the wave's next task (itself when this is the last one: a harmless redundant prefetch)
        const int tn = task + tstride < t1 ? task + tstride : tk;
        const int n0n_ = tn / tasks_per_img;
        const int n_n = a.rev_n ? a.N - 1 - n0n_ : n0n_;
        const int rem_n = tn - n0n_ * tasks_per_img;
        const int y0r_n = rem_n / a.strips;
        const int y_n = SKIP && ((n_n / img_stride) & 1) ? (y0r_n + yrot >= a.H ? y0r_n + yrot - a.H : y0r_n + yrot) : y0r_n;
        const __amdgpu_buffer_rsrc_t rsrc_n = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.in + (long long)n_n * a.in_sn), 0, a.in_img_bytes, 0x00020000);
        const int sbase_n = (a.in_off + y_n * STRIDE * a.in_pitch + (rem_n - y0r_n * a.strips) * XSTEP * STRIDE - (XMERGE ? 1 : 0)) * 4;
        const bool flip = S2FLIP && (y & 1), flip_n = S2FLIP && (y_n & 1);   // (wave-uniform)
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc(a.out + (long long)n * a.out_sn, 0, a.out_img_bytes, 0x00020000);
        const int sout = (a.out_off + y * a.out_pitch + x0) * 4;
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(RES ? a.res + (long long)n * a.res_sn : a.in), 0, RES ? a.res_img_bytes : 0u, 0x00020000);
        const int sres = RES ? (a.res_off + y * a.res_pitch + x0) * 4 : 0;
        const __amdgpu_buffer_rsrc_t rout2 = __builtin_amdgcn_make_buffer_rsrc(
            DUAL ? a.out2 + (long long)n * a.out2_sn : a.out, 0, DUAL ? a.out2_img_bytes : 0u, 0x00020000);
        const int sout2 = DUAL ? (a.out2_off + y * a.out2_pitch + x0) * 4 : 0;
        const __amdgpu_buffer_rsrc_t rout3 = __builtin_amdgcn_make_buffer_rsrc(
            FUSE ? a.out3 + (long long)n * a.out3_sn : a.out, 0, FUSE ? a.out3_img_bytes : 0u, 0x00020000);
        const int sout3 = FUSE ? (a.out3_off + y * a.out3_pitch + x0) * 4 : 0;

        typename M::acc_t acc[P];
        typename M::acc_t acc2[FUSE ? P : 1];   // F_FUSE1X1: the next block's reduced map of this strip
        // F_SKIP_PAD: chunk c = (dilation c / 3, tap row c % 3) reads input row yy + (ty - 1) << di; the middle row always exists
        auto chunk_live = [&](int yy, int c) {
            const int di = c / CPD, ty = c - di * CPD;
            return ty == 1 || (ty == 0 ? yy >= (1 << di) : yy + (1 << di) < a.H);
        };
        const int c_first = SKIP && !chunk_live(y, 0) ? 1 : 0;                 // (wave-uniform)
        const int c_first_n = SKIP && !chunk_live(y_n, 0) ? 1 : 0;

        // per-lane epilogue offsets: lanes beyond the row end get an offset past num_records, which the
        // buffer range check turns into a dropped store / zero load (no exec-mask branches)
        constexpr int OOB = 0x7ffffff0;
        int vo[P], vr[P], vo2[P], vo3[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const bool xok = x0 + (VEC ? xl : p * MT + px) < a.W;   // F_VEC: W % P == 0, a lane's pixels are all in or all out
            vo[p] = xok ? vout + (VEC ? 0 : p * MT * 4) : OOB;
            vr[p] = xok ? vres + (VEC ? 0 : p * MT * 4) : OOB;
            vo2[p] = xok ? vout2 + (VEC ? 0 : p * MT * 4) : OOB;
            vo3[p] = xok ? vout3 + (VEC ? 0 : p * MT * 4) : OOB;
        }
        // The residual (block input) values of a whole concat slot live in dedicated registers and are
        // requested a full dilation ahead of the epilogue that adds them: fetched next to the store,
        // they cost four exposed HBM round trips per slot (measured 33 us of a 183 us launch).
        // F_RES_RING: only RR registers' worth of residual values is in flight -- register r's slot is refilled with register
        // r + RR's values (of this concat slot, or of the next one) as soon as the epilogue has consumed it.  With the
        // fused 1x1 the kernel has no room for a whole slot (16 x P registers) beside its two accumulator sets.
        constexpr int RR = !RES ? 1 : (FLAGS & F_RES_RING) ? M::NACC / CFG_RES_RING_DIV : M::NACC;
        float resv[RR][P];
        constexpr bool RES_TOP = RES && RR == M::NACC && CFG_RES_TOP;
        auto load_res = [&](int di, int r) {   // residual values of accumulator register r of concat slot di
            if (!RES)
                return;
            if ((FLAGS & F_X_NOEPIMEM) || (kDiag && ((CFG_X_EPI & 1) || ((CFG_X_EPI & 4) && (r & 1))))) {
#pragma unroll
                for (int p = 0; p < P; ++p)
                    resv[r % RR][p] = __builtin_bit_cast(float, di + r + p);
                return;
            }
            const int nout = di == 0 ? NOUT1 : NOUT;
            const int cb = di == 0 ? 0 : NOUT1 + (di - 1) * NOUT;
            const int ch0 = M::row(r, 0);
            const bool live = ch0 + kq * KSTR < nout;
            const int sr = (cb + ch0) * a.res_sc * 4 + sres;
            if (VEC) {
                if (FLAGS & F_X_RESL2)
                    buf_load_vec<P, RAUX>(rres, live ? ((int)(wg & 255) * 8192 + ((vr[0] + sr) & 0x1ff0)) : OOB, 0, resv[r % RR]);
                else
                    buf_load_vec<P, RAUX>(rres, live ? vr[0] : OOB, sr, resv[r % RR]);
                return;
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
                resv[r % RR][p] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rres, live ? vr[p] : OOB, sr, RAUX));
        };
        auto prefetch_res = [&](int di) {   // everything the ring holds of slot di
#pragma unroll
            for (int r = 0; r < RR; ++r)
                load_res(di, r);
        };

        // The k-loop is one flat sequence of k-steps run through a ring of D operand slots: right
        // after the MFMAs of a step have consumed their slot, the slot is refilled with the step D
        // later.  Every load therefore has D steps of MFMA work (thousands of cycles) to land, at
        // the register cost of a single operand set.  (The first version waited on loads it had just
        // issued: SQ_WAIT_ANY 57 % of wave cycles, MFMA pipe 42 % busy.)  The three horizontal taps
        // of a row are consecutive steps, so two of three B loads hit lines the wave has just pulled
        // into L1.
        // (tap row, input-channel group) of row group g of chunk cw WITHIN its dilation.  g is a compile-time constant after
        // unrolling; written as `(cw * G + g) / NSTEP` the compiler divided by NSTEP (mul_hi + shifts + fix-up: ~16 scalar
        // instructions) for every row group of every chunk although the quotient is known per chunk: the fused level-3 ESP
        // launch issued 394 scalar instructions per 78 matrix instructions, now 140 (worth 0.5-1 %: scalar issue overlaps the
        // matrix pipe well -- profiles/r04_ab_bnload_isa.txt).  So the three shapes that occur are spelled out.
        auto decode_rg = [&](int cw, int g, int &ty0, int &sidx) {
            if constexpr (G % NSTEP == 0) {          // a chunk is whole tap rows (level 2: a whole dilation)
                ty0 = cw * (G / NSTEP) + g / NSTEP;
                sidx = g % NSTEP;
            } else if constexpr (NSTEP % G == 0) {   // a tap row is whole chunks (stride-2 reduces, 1x1s)
                constexpr int CPR = NSTEP / G;
                ty0 = cw / CPR;
                sidx = (cw - ty0 * CPR) * G + g;
            } else {
                const int rg = cw * G + g;
                ty0 = rg / NSTEP;
                sidx = rg - ty0 * NSTEP;
            }
        };
        // operands of chunk c (dilation c / CPD, row groups (c % CPD)*G ..) into ring slots 0..D-1
        auto fetch_b = [&](const __amdgpu_buffer_rsrc_t &rs, int sb, int c, int g, int tx, bool fl) {
            const int di = c / CPD;
            int ty0, sidx;
            decode_rg(c - di * CPD, g, ty0, sidx);
            const int ty = S2FLIP && fl ? TYN - 1 - ty0 : ty0;
            const int toff = TAPS == 9 ? ((ty - 1) * a.in_pitch + (tx - 1)) << di : TAPS == 3 ? (ty - 1) * a.in_pitch : 0;
            const int soff = sb + (toff + sidx * KL * a.in_sc) * 4;
            if ((FLAGS & F_X_NOLOAD) && staged)   // (timing only: the ring keeps the values of the task's first chunk -- no instruction at all)
                return;
            if (VEC) {
                buf_load_vec<P, IAUX>(rs, voff, soff, bq[S2P ? 0 : (g * TXN + tx) % R]);
                return;
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
                bq[(g * TXN + tx) % R][p] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + p * MT * STRIDE * 4, soff, IAUX));
        };
        auto fetch_pair = [&](const __amdgpu_buffer_rsrc_t &rs, int sb, int c, int g, bool fl) {
            if ((FLAGS & F_X_NOLOAD) && staged)   // (timing only, as in fetch_b)
                return;
            int ty0, sidx;
            decode_rg(c, g, ty0, sidx);
            const int ty = S2FLIP && fl ? TYN - 1 - ty0 : ty0;
            const int soff = sb + ((ty - 1) * a.in_pitch - 1 + sidx * KL * a.in_sc) * 4;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff + p * MT * STRIDE * 4, soff, IAUX);
                // (copy the elements out first: __builtin_bit_cast applied directly to v[1] reads element 0 with this hipcc)
                const unsigned e0 = v[0], e1 = v[1], e2 = v[2];
                bl[g][p][0] = __builtin_bit_cast(float, e0);
                bl[g][p][1] = __builtin_bit_cast(float, e1);
                bl[g][p][2] = __builtin_bit_cast(float, e2);
            }
        };
        auto fetch_a = [&](int c, int g, int tx, bool fl) {
            const int di = c / CPD;
            int ty0, sidx;
            decode_rg(c - di * CPD, g, ty0, sidx);
            const int ty = S2FLIP && fl ? TYN - 1 - ty0 : ty0;
            const int tap = TAPS == 9 ? ty * 3 + tx : TAPS == 3 ? ty : 0;
            if ((FLAGS & F_X_NOLDS) && staged)   // (timing only: the ring keeps the values of the task's first chunk -- no instruction at all)
                return;
            if (AGL)
                aq[(g * TXN + tx) % R] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_w, lbase * 4, (((di * TAPS + tap) * CINP + sidx * KL) * NROW) * 4, 0));
            else
                aq[(g * TXN + tx) % R] = lds[((di * TAPS + tap) * CINP + sidx * KL) * NROW + lbase];
        };

        // F_BNLOAD: the three parameters of this lane's channel in row group g of chunk c (task row yy); one group ahead
        auto bnl_fetch = [&](int cc, int gg, int yy, bool ff, float &sc2, float &sh2, float &al2, bool &raw) {
            int ty0c, sidxc;
            decode_rg(cc, gg, ty0c, sidxc);
            const int tyc = S2FLIP && ff ? TYN - 1 - ty0c : ty0c;
            raw = sidxc >= a.bnl_s0 && sidxc < a.bnl_s1;                // (wave-uniform) the other groups have identity parameters
            const bool padrow = yy * STRIDE + tyc - 1 < 0 && raw;
            const int ch = (padrow ? NSTEP : sidxc) * KL + kq;
            sc2 = bnl[ch];
            sh2 = bnl[BNL_C + ch];
            al2 = bnl[2 * BNL_C + ch];
        };
        // The transform of row group k+1 is interleaved with the matrix instructions of group k (one value after each MFMA:
        // three VALU instructions that run while the MFMA occupies the pipe); done in one piece in front of a group's first
        // MFMA it stalled the wave's matrix stream for ~150 cycles per group (0.156 -> 0.196 ms per launch).
        float tsc = 1.0f, tsh = 0.0f, tal = 1.0f, tpin = 0.0f;   // parameters of the group being transformed (the next one)
        float nsc = 1.0f, nsh = 0.0f, nal = 1.0f;                // ... and of the one after it, in flight from LDS
        bool traw = false, nraw = false;                         // (whether those groups hold raw channels at all; an identity-skipping
                                                                 // variant behind a per-MFMA uniform branch measured slower)
        // (the transformed operands go to a two-group buffer of their own: rewritten in place, every value cost a register copy --
        // 72 v_mov per chunk -- because the ring's registers are load destinations)
        float bt[BNL ? 2 : 1][P][3];
        static_assert(!BNL || G % 2 == 0, "F_BNLOAD: the two-group operand buffer alternates by row-group parity");
        auto bnl_apply = [&](float &dst, float v, float sc2, float sh2, float al2, float pin2, bool zero) {
#if defined(GS_DIAG) && defined(CFG_BNL_ABLATE)
            // timing-only ablations (results wrong): 1 = only wait for the operand (no arithmetic), 2 = the arithmetic on a
            // value that does not come from the operand (no early wait)
            if (CFG_BNL_ABLATE == 1) {
                asm volatile("" : "+v"(v));
                dst = v;
                return;
            }
            float w = sc2;
            w = w * sc2 + sh2;
            w = prelu_med3(w, al2, pin2);
            asm volatile("" ::"v"(w));
            dst = v;
            return;
#endif
            v = v * sc2 + sh2;
            v = prelu_med3(v, al2, pin2);
            dst = zero ? 0.0f : v;
        };

        // prologue: the first chunk's activations are requested before the weights are staged, so
        // their latency overlaps the LDS fill
        if (task == t0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (S2P) {
                    fetch_pair(rsrc, sbase, 0, g, flip);
                    continue;
                }
#pragma unroll
                for (int tx = 0; tx < TXN; ++tx)
                    if (g * TXN + tx < R - RB)
                        fetch_b(rsrc, sbase, c_first, g, tx, flip);
            }
        }
        if (!staged) {
            // weights -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR
            // round trip), every piece in flight at once; a register-staged copy loop took 9 us of a
            // 183 us launch here
            const int pieces = (a.wfloats - LDS_SRC0 + 255) / 256;
            if ((FLAGS & F_X_STAMP2) && lane == 0)   // (diagnostic: the ramp in three pieces -- start -> here -> DMA issued -> staged)
                a.stamp[wg * STAMP2_SLOTS + STAMP2_SLOTS - 2] = __builtin_amdgcn_s_memrealtime();
            // every workgroup starts at a different piece, so that the 256 CUs do not ask one L2 channel for the same
            // lines at the same moment (level-3 ESP launch 0.1888 -> 0.1866 ms, measured three times)
            const int rot = (int)((blockIdx.x * (unsigned)CFG_STAGE_ROT) % (unsigned)(pieces > 0 ? pieces : 1));
            for (int j0 = wid; j0 < pieces; j0 += WAVES) {
                const int j = j0 + rot < pieces ? j0 + rot : j0 + rot - pieces;
#if defined(CFG_X_STAGE_PLAIN)
                *reinterpret_cast<f32x4 *>(lds + j * 256 + lane * 4) = *reinterpret_cast<const f32x4 *>(a.wpack + LDS_SRC0 + j * 256 + lane * 4);
#else
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(a.wpack + LDS_SRC0 + j * 256 + lane * 4),
                    (__attribute__((address_space(3))) void *)(lds + j * 256), 16, 0, 0);
#endif
            }
            // The barrier (with the `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of it) is UNCONDITIONAL on this path: an LDS-DMA
            // instruction is a FLAT operation to the compiler's wait-count pass, and while one may be pending on ANY path into a
            // block every wait in that block is forced to vmcnt(0) / lgkmcnt(0).  Behind `if (pieces > 0)` the path without the
            // wait kept "a FLAT operation may be pending" alive around the whole task loop: every chunk began with a full drain
            // of the operand ring, and the epilogue's first residual use with another (round 6; tools/explore: the assembly of
            // both builds).  Same registers, same bits; measured the same speed at a 39-step ring (profiles/r06_ab_waitfix.txt), but
            // the ring and residual experiments of rounds 1-5 were all measured UNDER those forced drains.
            if ((FLAGS & F_X_STAMP2) && lane == 0)
                a.stamp[wg * STAMP2_SLOTS + STAMP2_SLOTS - 1] = __builtin_amdgcn_s_memrealtime();
#if defined(CFG_X_BARRIER_COND)
            if (pieces > 0)
#endif
                __syncthreads();
            staged = true;
        }
        if (idle)
            break;
        if ((FLAGS & F_X_STAMP) && lane == 0 && task == t0) {
            a.stamp[wg * 8 + 0] = tstamp0;
            a.stamp[wg * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        if ((FLAGS & F_X_STAMP2) && lane == 0 && task == t0) {
            a.stamp[wg * STAMP2_SLOTS + 0] = tstamp0;
            a.stamp[wg * STAMP2_SLOTS + 1] = __builtin_amdgcn_s_memrealtime();
        }
        // F_X_STAMP2: first slot of this task (tasks beyond the slots are not stamped)
        const int ti_ = (FLAGS & F_X_STAMP2) ? (task - t0) / tstride : 0;
        const bool stamp2 = (FLAGS & F_X_STAMP2) && lane == 0 && 2 + (ti_ + 1) * 2 * NCHUNK <= STAMP2_SLOTS;
        const long long sbase2 = wg * STAMP2_SLOTS + 2 + ti_ * 2 * NCHUNK;
        // optional stagger: the two waves that share a SIMD run the same program on equal-sized tasks
        // and would otherwise reach their epilogues (no MFMA issue) together
        if (kDiag && a.stagger > 0 && task == t0) {
            const int ph = second_of_simd ? 1 : 0;   // the second wave of every SIMD starts late
            for (int z = 0; z < a.stagger * ph; ++z)
                __builtin_amdgcn_s_sleep(16);
        }
        if (task == t0) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int tx = 0; tx < TXN; ++tx)
                    if (g * TXN + tx < R - RB)
                        fetch_a(c_first, g, tx, flip);
        }

        if (BNL) {   // the task's first row group is transformed here, in one piece (once per ~150 k cycles)
            bnl_fetch(c_first, 0, y, flip, tsc, tsh, tal, traw);
            bnl_fetch(c_first, 1, y, flip, nsc, nsh, nal, nraw);
            tpin = prelu_pin(tal);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    bnl_apply(bt[0][p][t], bl[0][p][t], tsc, tsh, tal, tpin, t == 0 && p == 0 && x0 == 0 && px == 0);
        }
        if (!RES_TOP)
            prefetch_res(0);
        if (FUSE) {
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc2[p] = (typename M::acc_t)(0.0f);
        }
        if (prio_mode == 1) {
            if ((((task - t0) / tstride) + (second_of_simd ? 1 : 0)) & 1)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        } else if (prio_mode == 3 && task == t0) {
            if (second_of_simd)
                __builtin_amdgcn_s_setprio(1);
        }

        // One accumulator register (channel row r of both k-groups) of concat slot di on its way out: + residual, BN, PReLU,
        // store(s), the fused 1x1's matrix instructions.  `src` is the accumulator itself or (F_EPI_PIPE) its snapshot.
        // SPLIT: a slot's outputs are kept (osave) and the fused 1x1's matrix instructions run in one piece behind the slot's
        // arithmetic and memory instructions.  Interleaved register by register (the round-2 form), every one of them waited for
        // a free slot of the matrix pipe -- which the SIMD's other wave keeps busy -- in the middle of a serial stretch of LDS reads
        // and VALU work: an epilogue took 5-7 us, during which that other wave alone could not keep the pipe busy.
        constexpr bool SPLIT = FUSE && CFG_EPI_SPLIT && !EPI_PIPED;
        float osave[SPLIT ? M::NACC : 1][P];
        // the per-register constants of an epilogue step: BN scale / shift / PReLU slope of the lane's channel (twice with F_DUAL)
        // and the fused 1x1's A operand
        struct EpiParams {
            float scale = 1.0f, shift = 0.0f, alpha = 1.0f, scale2 = 1.0f, shift2 = 0.0f, alpha2 = 1.0f, a2 = 0.0f;
        };
        auto epi_params = [&](int di, auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const int nout = di == 0 ? NOUT1 : NOUT;
            const int cb = di == 0 ? 0 : NOUT1 + (di - 1) * NOUT;
            const int ch0 = M::row(r, 0);
            const bool live = ch0 + kq * KSTR < nout;
            EpiParams q;
            // (read at the top of the register's step, not inside the uniform branch around its MFMAs: the LDS
            // latency then runs under the BN / PReLU arithmetic instead of in front of the matrix instructions)
            q.a2 = (FUSE && !SPLIT) ? tab[(di * M::NACC + r) * 64 + lane] : 0.0f;
            if (BNACT) {
                const float *bp = bnp + (live ? cb + ch0 + kq * KSTR : 0);
                q.scale = bp[0];
                q.shift = bp[COUT];
                q.alpha = bp[2 * COUT];
                if (DUAL) {
                    q.scale2 = bp[3 * COUT];
                    q.shift2 = bp[4 * COUT];
                    q.alpha2 = bp[5 * COUT];
                }
            }
            return q;
        };
        // EPI_PRELOAD (few accumulator registers: the 16x16x4 forms): a slot's constants are all requested up front -- read register
        // by register, each of the four steps began with three or four LDS reads and a wait for them
        constexpr bool EPI_PRELOAD = CFG_EPI_PRELOAD && M::NACC <= 4 && !EPI_PIPED;
        auto epi_reg = [&](int di, auto r_, const typename M::acc_t *src, bool refill_next, const EpiParams *pre = nullptr) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const int nout = di == 0 ? NOUT1 : NOUT;
            const int cb = di == 0 ? 0 : NOUT1 + (di - 1) * NOUT;
            const int ch0 = M::row(r, 0);   // channel held by k-group 0; group kq holds ch0 + kq*KSTR
            const bool live = ch0 + kq * KSTR < nout;
            const EpiParams q = pre ? pre[r] : epi_params(di, r_);
            const float a2 = q.a2;
            const int so = (cb + ch0) * a.out_sc * 4 + sout;
            const int so2 = DUAL ? (a.out2_coff + cb + ch0) * a.out2_sc * 4 + sout2 : 0;
            const float scale = q.scale, shift = q.shift, alpha = q.alpha, scale2 = q.scale2, shift2 = q.shift2, alpha2 = q.alpha2;
            float o1[P], o2[P];
            const float pin = prelu_pin(alpha), pin2 = prelu_pin(alpha2);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float v = src[p][r];
                if (RES)
                    v += resv[r % RR][p];
                if (BNACT) {
                    v = v * scale + shift;
                    v = prelu_med3(v, alpha, pin);
                }
                o1[p] = v;
                if (!VEC && STORE1)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, live ? vo[p] : OOB, so, SAUX);
                if (DUAL) {
                    float v2 = v * scale2 + shift2;
                    v2 = prelu_med3(v2, alpha2, pin2);
                    o2[p] = v2;
                    if (!VEC)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v2), rout2,
                                                              live ? vo2[p] : OOB, so2, SAUX2);
                }
            }
            if (RES && RR < M::NACC) {   // the slot just consumed is refilled at once (uniform control flow)
                if (r + RR < M::NACC)
                    load_res(di, r + RR);
                else if (di + 1 < NDIL)
                    load_res(di + 1, r + RR - M::NACC);
            }
            if (RES && refill_next && di + 1 < NDIL)   // F_EPI_PIPE: register r's residual of the NEXT slot, a dilation ahead of its use
                load_res(di + 1, r);
            if (VEC && STORE1 && !(FLAGS & F_X_NOEPIMEM) && !(kDiag && ((CFG_X_EPI & 2) || ((CFG_X_EPI & 8) && (r & 1)))))
                buf_store_vec<P, SAUX>(rout, live ? ((FLAGS & F_X_STL2) ? ((int)(wg & 255) * 8192 + ((vo[0] + so) & 0x1ff0)) : vo[0] + so) : OOB, o1);
            if (VEC && DUAL)
                buf_store_vec<P, SAUX2>(rout2, live ? vo2[0] + so2 : OOB, o2);
            if (FUSE && ch0 < nout) {   // (uniform) registers whose two channels are both beyond the slot hold nothing
                // k = lane's k-group <-> channel cb + ch0 + kq*KSTR; the table row is zero for channels beyond the slot
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (SPLIT)
                        osave[SPLIT ? r : 0][p] = o1[p];
                    else
                        acc2[p] = M::run(a2, o1[p], acc2[p]);
                }
            }
        };
        // SPLIT: the fused 1x1's matrix instructions of a whole slot, after the slot's arithmetic, loads and stores (same
        // accumulation order per accumulator: registers 0, 1, 2, ...)
        auto epi_fused = [&](int di) __attribute__((always_inline)) {
            static_for<M::NACC>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                const int nout = di == 0 ? NOUT1 : NOUT;
                if (M::row(r, 0) < nout) {
                    const float a2 = tab[(di * M::NACC + r) * 64 + lane];
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        acc2[p] = M::run(a2, osave[SPLIT ? r : 0][p], acc2[p]);
                }
            });
        };
        typename M::acc_t snap[EPI_PIPED ? P : 1];   // F_EPI_PIPE: the accumulator as the previous dilation left it
#if defined(GS_DIAG) && defined(CFG_X_S2_EXTRA)
        typename M::acc_t xacc[BNL ? P : 1];
        if (BNL) {
#pragma unroll
            for (int p = 0; p < P; ++p)
                xacc[p] = (typename M::acc_t)(0.0f);
        }
#endif

        for (int c = 0; c < NCHUNK; ++c) {
            if (NDIL > 1 && c % CPD == 0 && prio_mode == 0) {
                // The two waves of a SIMD run the same program; arbitration prefers the older one, which
                // then finishes its task ~20 % earlier and leaves its partner alone on the pipe.
                // Alternating static priority per dilation keeps the pair level.
                if (((c / CPD) + (second_of_simd ? 1 : 0)) & 1)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
            // RES_TOP: a slot's residual is requested at the top of its own dilation (a dilation of k-steps ahead of the epilogue that
            // adds it) instead of at the end of the previous slot's epilogue: the same distance, but the values are no longer carried
            // around the loop -- the compiler rotated them through a second register set with a copy and an `s_waitcnt vmcnt(0)` at
            // every dilation boundary
            if (RES_TOP && c % CPD == 0)
                prefetch_res(c / CPD);
            if (c % CPD == 0 && c < 2 * CPD) {   // d1 and d2 start fresh; d4, d8, d16 keep adding (HFF)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p] = (typename M::acc_t)(0.0f);
            }
            // the ring is refilled with the next (live) chunk of this task, or with the first one of the next task
            int nxl = c + 1;
            if (SKIP)
                while (nxl < NCHUNK && !chunk_live(y, nxl))
                    ++nxl;
            const bool last = nxl >= NCHUNK;
            const int nx = last ? c_first_n : nxl;
            const __amdgpu_buffer_rsrc_t rs = last ? rsrc_n : rsrc;
            const int sb = last ? sbase_n : sbase;
            const bool fl = last ? flip_n : flip;
            // refill of the ring slot that step u (MID: step u - 1) of this chunk has just left: the step R (R - 1) later
            auto refill = [&](int u) __attribute__((always_inline)) {
                const int v = u + R - RB;
                if (v < D && (R < D || MID)) {
                    fetch_b(rsrc, sbase, c, v / TXN, v % TXN, flip);
                    fetch_a(c, v / TXN, v % TXN, flip);
                } else {
                    const int gn = (v - D) / TXN, txn = (v - D) % TXN;
                    if (!S2P)
                        fetch_b(rs, sb, nx, gn, txn, fl);
                    else if (txn == 2)
                        fetch_pair(rs, sb, nx, gn, fl);
                    fetch_a(nx, gn, txn, fl);
                }
            };
            if (!SKIP || chunk_live(y, c)) {
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int tx = 0; tx < TXN; ++tx) {
                    const int u = g * TXN + tx;
                    // F_BNLOAD: while group g multiplies, group g+1 (of this chunk, or the first of the next) is transformed
                    const bool ahead = BNL && (g + 1 < G || !last);
                    if (BNL && tx == 0) {
                        tsc = nsc;
                        tsh = nsh;
                        tal = nal;
                        traw = nraw;
                        tpin = prelu_pin(tal);
                        if (g + 2 < G)
                            bnl_fetch(c, g + 2, y, flip, nsc, nsh, nal, nraw);
                        else if (!last)
                            bnl_fetch(nx, g + 2 - G, y, flip, nsc, nsh, nal, nraw);
                    }
#if defined(GS_DIAG) && defined(CFG_X_MFMA_KEEP)
                    // ceiling experiment (results are garbage): only every CFG_X_MFMA_KEEP-th k-step's matrix instructions
                    // are issued; the operands of the others are still loaded and waited for
                    if (u % CFG_X_MFMA_KEEP != 0) {
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const float bv = S2P ? bl[S2P ? g : 0][p][S2P ? tx : 0] : bq[S2P ? 0 : u % R][p];
                            asm volatile("" ::"v"(bv), "v"(aq[u % RA]));
                        }
                    } else
#endif
#if defined(GS_DIAG) && defined(CFG_X_S2_EXTRA)
                    // cost experiment for "level3_C as 20 extra rows of the stride-2 reduce" (profiles/r05_ab_level3c_fusion.txt; results
                    // unchanged, time only): the four taps (ty, tx) in {1,2} x {1,2} -- the four level-2 pixels under a level-3 pixel --
                    // each issue a second matrix instruction into a second accumulator set, as a second 32-row block would
                    if (BNL) {
                        int ty0x, sidxx;
                        decode_rg(c, g, ty0x, sidxx);
                        if (ty0x >= 1 && tx >= 1) {
#pragma unroll
                            for (int p = 0; p < P; ++p)
                                xacc[p] = M::run(aq[u % RA], bt[BNL ? g & 1 : 0][p][S2P ? tx : 0], xacc[p]);
                        }
                    }
#endif
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        acc[p] = M::run(aq[u % RA], BNL ? bt[BNL ? g & 1 : 0][p][S2P ? tx : 0]
                                                       : S2P ? bl[S2P ? g : 0][p][S2P ? tx : 0] : bq[S2P ? 0 : u % R][p], acc[p]);
                        if (MID && p == P / 2 - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            refill(u);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (ahead) {   // element e = tx*P + p of the next group: (pixel e / 3, tap e % 3)
                            const int e = tx * P + p;
                            if (e < 3 * P)
                                bnl_apply(bt[BNL ? (g + 1) & 1 : 0][e / 3][e % 3], bl[S2P ? (g + 1) % G : 0][e / 3][e % 3], tsc, tsh, tal, tpin,
                                          e == 0 && x0 == 0 && px == 0);
                        }
                    }
                    if (EPI_PIPED && c > 0) {
                        // the previous slot's epilogue, one accumulator register after every EPI_EVERY-th k-step (early in the
                        // dilation: the residual reloads it issues are for this dilation's own slot)
                        constexpr int EPI_EVERY = D / (2 * M::NACC) > 0 ? D / (2 * M::NACC) : 1;
                        static_for<M::NACC>([&](auto r_) {   // (u is a constant once the step loops are unrolled: one call survives)
                            if (u == decltype(r_)::value * EPI_EVERY + EPI_EVERY - 1)
                                epi_reg(c - 1, r_, snap, true);
                        });
                    }
                    // the slot just consumed is refilled with the step R later: of this chunk, or of the next one
                    if (!MID)
                        refill(u);
                    // pin the ring order: left alone, hipcc sinks the refill loads to the end of the
                    // chunk, which shrinks the prefetch distance from D steps to a few
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (stamp2)
                a.stamp[sbase2 + 2 * c] = __builtin_amdgcn_s_memrealtime();
            if ((c + 1) % CPD != 0)
                continue;
            if (FLAGS & F_X_NOEPI) {
                float keep = 0.0f;   // keep every accumulator register live
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r)
                        keep += acc[p][r];
                if (keep == 123.456f)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, keep), rout, vout, sout, 0);
                continue;
            }
            // last chunk of a dilation: write this branch's concat slot (the accumulator keeps
            // running for the fusion adds)
            const int di = c / CPD;
            if (EPI_PIPED && di + 1 < NDIL) {   // ... between the k-steps of the next dilation, from a snapshot
#pragma unroll
                for (int p = 0; p < P; ++p)
                    snap[EPI_PIPED ? p : 0] = acc[p];
                continue;
            }
            if (XMERGE) {
                // per-wave LDS tile [MT rows][P*MT columns]; column i is input x' = x0 - 1 + i
                // row pitch P*MT + 4 floats: the four k-groups of a store write rows 4 apart, which a pitch of 128
                // floats put on the same 16 banks (4-way conflict on every tile write)
                constexpr int TS = P * MT + 4;
                float *tile = lds + a.lds_tile_off + wid * (MT * TS);
#pragma unroll
                for (int r = 0; r < M::NACC; ++r)
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        tile[M::row(r, kq) * TS + (VEC ? xl + p : p * MT + px)] = acc[p][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // out[o][x0 + m] = Z[0*NOUT1+o][m] + Z[1*NOUT1+o][m+1] + Z[2*NOUT1+o][m+2], m < XSTEP
#pragma unroll
                for (int t = 0; t < (XSTEP + 63) / 64; ++t) {
                    const int mcol = t * 64 + lane;
                    const bool ok = mcol < XSTEP && x0 + mcol < a.W;
                    const int mc = mcol < XSTEP ? mcol : 0;
#pragma unroll
                    for (int o = 0; o < NOUT1; ++o) {
                        float v = tile[o * TS + mc] + tile[(NOUT1 + o) * TS + mc + 1] + tile[(2 * NOUT1 + o) * TS + mc + 2];
                        if (BNACT) {
                            v = v * bnp[o] + bnp[COUT + o];
                            const float al = bnp[2 * COUT + o];
                            v = prelu_med3(v, al, prelu_pin(al));
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, ok ? mcol * 4 : OOB,
                                                              o * a.out_sc * 4 + sout, SAUX);
                    }
                }
                __builtin_amdgcn_wave_barrier();   // the tile is rewritten by this wave's next task
                continue;
            }
            // Branch-free epilogue: addresses are (uniform per accumulator register, in an SGPR) + (one
            // per-lane offset), so the whole slot is straight-line VALU + buffer stores.
#if CFG_EPI_PRIO
            __builtin_amdgcn_s_setprio(CFG_EPI_PRIO);
#endif
            if constexpr (EPI_PRELOAD) {
                EpiParams pre[M::NACC];
                static_for<M::NACC>([&](auto r_) { pre[decltype(r_)::value] = epi_params(di, r_); });
                static_for<M::NACC>([&](auto r_) { epi_reg(di, r_, acc, false, pre); });
            } else {
                static_for<M::NACC>([&](auto r_) { epi_reg(di, r_, acc, false); });
            }
#if defined(GS_DIAG) && defined(CFG_X_STAMP_A)
            if (stamp2)   // (diagnostic: the epilogue's end stamp BEFORE the fused matrix instructions)
                a.stamp[sbase2 + 2 * c + 1] = __builtin_amdgcn_s_memrealtime();
#endif
            if (SPLIT) {
                __builtin_amdgcn_sched_barrier(0);   // (or the scheduler interleaves the two phases again)
                epi_fused(di);
            }
#if CFG_EPI_PRIO
            if (prio_mode == 1 && ((((task - t0) / tstride) + (second_of_simd ? 1 : 0)) & 1))
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
#endif
#if defined(GS_DIAG) && defined(CFG_X_S2_EXTRA)
            if (BNL) {   // keep the second set live; it would be stored as 20 more planes (5 classes x 4 level-2 pixels)
                float keep = 0.0f;
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r)
                        keep += xacc[p][r];
                if (keep == 123.456f)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, keep), rout, vout, sout, 0);
            }
#endif
            if (FUSE && di + 1 == NDIL) {
                // the block's output is complete for this strip: its 1x1 reduce leaves for the next block's reduced map
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) {
                    const int ch0 = M::row(r, 0);
                    if (ch0 >= a.nout3)
                        continue;
                    const bool live3 = ch0 + kq * KSTR < a.nout3;
                    const int so3 = ch0 * a.out3_sc * 4 + sout3;
                    float o3[P];
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        o3[p] = acc2[p][r];
                        if (!VEC)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o3[p]), rout3, live3 ? vo3[p] : OOB, so3, 0);
                    }
                    if (VEC)
                        buf_store_vec<P, 0>(rout3, live3 ? vo3[0] + so3 : OOB, o3);
                }
            }
            if (!RES_TOP && RR == M::NACC && di + 1 < NDIL)
                prefetch_res(di + 1);
            if ((FLAGS & F_X_STAMP) && lane == 0 && task == t0)
                a.stamp[wg * 8 + 2 + di] = __builtin_amdgcn_s_memrealtime();
#if !(defined(GS_DIAG) && defined(CFG_X_STAMP_A))
            if (stamp2)
                a.stamp[sbase2 + 2 * c + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        }
    }
}

// number of floats of a configuration's image in the weight blob (see conv_image)
constexpr int conv_wfloats(int CINP, int TAPS, int NDIL, int NOUT1, int NOUT, bool bn, bool dual = false, bool xmerge = false,
                           int fuse_nacc = 0)
{
    return conv_image(CINP, TAPS, NDIL, NOUT1, NOUT, bn, dual, xmerge, fuse_nacc).total;
}

// Per (kernel instantiation, device): the dynamic-LDS attribute and the occupancy figure.  One process normally drives one
// GPU, but a handle may be created on any device and from any thread, so the cache is keyed by device and locked.
struct LaunchInfo {
    bool attr_done = false;
    int per_cu = 0;
};

template <int MT, int WAVES, int CINP, int TAPS, int STRIDE, int NDIL, int NOUT1, int NOUT, int P, int G, int FLAGS>
gs_status launch_conv_mfma(ConvArgs a, int num_cus, hipStream_t stream)
{
    auto kern = conv_mfma_kernel<MT, WAVES, CINP, TAPS, STRIDE, NDIL, NOUT1, NOUT, P, G, FLAGS>;
    constexpr ConvImage im = conv_image(CINP, TAPS, NDIL, NOUT1, NOUT, FLAGS & F_BNACT, FLAGS & F_DUAL, FLAGS & F_XMERGE,
                                        (FLAGS & F_FUSE1X1) ? Mfma<MT>::NACC : 0);
    a.strips = cdiv(a.W, (FLAGS & F_XMERGE) ? P * MT - 2 : P * MT);
    a.total_tasks = a.N * a.H * a.strips;
    a.prio_mode = 1;   // measured on the level-3 branch kernel: per dilation 0.171 ms, per task 0.167, off 0.168, fixed 0.166
#ifdef GS_DIAG
    if (const char *e = std::getenv("GS_PRIO"))
        a.prio_mode = std::atoi(e);
    if (const char *e = std::getenv("GS_STAGGER"))
        a.stagger = std::atoi(e);
#endif
    constexpr int EXTRA = (FLAGS & F_BNLOAD) ? 3 * (CINP + Mfma<MT>::KL) : 0;   // the on-load BN / PReLU table follows the image
    a.wfloats = im.total + EXTRA;
    const int lds_floats = im.total + EXTRA - ((FLAGS & F_A_GLOBAL) ? im.w : 0);
    a.lds_tile_off = (lds_floats + 255) / 256 * 256;
    const size_t lds_bytes = (size_t)(a.lds_tile_off + ((FLAGS & F_XMERGE) ? WAVES * MT * (P * MT + 4) : 0)) * sizeof(float);   // whole 1-KiB DMA pieces (+ tiles)
    static std::mutex mu;
    static std::map<int, LaunchInfo> by_device;
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    int per_cu;
    {
        std::lock_guard<std::mutex> lock(mu);
        LaunchInfo &li = by_device[dev];
        if (!li.attr_done) {
            GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
            li.attr_done = true;
        }
        // resident workgroups per CU from the occupancy query (registers, LDS, wave slots)
        if (li.per_cu == 0) {
            int nb = 0;
            GS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), WAVES * 64, lds_bytes));
            li.per_cu = nb < 1 ? 1 : nb;
        }
        per_cu = li.per_cu;
    }
    int grid = num_cus * per_cu;
    if (grid >= 8) grid = grid / 8 * 8;
    a.wu = WAVES;
    if ((long long)a.total_tasks < (long long)grid * WAVES) {
        // fewer tasks than wave slots: as few waves per workgroup as cover the tasks on all CUs, then as many workgroups
        // as that needs (a whole multiple of 8, rounded UP: a wave with two tasks would double the launch)
        a.wu = cdiv(a.total_tasks, grid);
        if (a.wu < 1) a.wu = 1;
        int g = cdiv(a.total_tasks, a.wu);
        if (g >= 8) g = (g + 7) / 8 * 8;
        grid = g < grid ? g : grid;
    }
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
