// Decoder tail in ONE kernel: conv = CBR(19+classes, classes, 3) over cat([comb_l2_l3, output0_cat]) ->
// classifier ConvTranspose2d(classes, classes, 2, stride 2) -> logits -> first-max argmax -> uint8 mask ->
// per-class pixel counts.  reference: Model.py:375-377, VisualizeResults_iou.py:128 (argmax), :151-155 (counts).
//
// The 3x3 convolution runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) in the row-merged form: the MFMA rows
// carry (horizontal tap tx, output channel o) = 15 of 16 rows, the MFMA columns are 16 consecutive input columns, K
// walks the 24 input planes four at a time.  What is new against a per-row kernel:
//   * a wave owns a BAND of output rows of its column strip and sweeps down it.  An input row segment is loaded ONCE
//     and feeds three MFMAs -- the vertical taps ty = 0,1,2 of the three output rows it belongs to, held in three
//     rolling accumulator sets -- so operand traffic is (R+2)/R x the input instead of 3x (the per-row kernel spent
//     40 % of its time waiting on those loads);
//   * the 18 A operands (3 vertical taps x 6 plane groups) of a lane never change: they live in registers, the k-loop
//     reads no weights at all;
//   * the epilogue goes on to the classifier deconvolution, the argmax and the counts, so the half-resolution
//     5-channel map is neither written nor read back (2 x 84 MB per 32 tiles and one launch less).
// The arithmetic (order of every accumulation chain) is the one of the kernels this replaces: masks are bit-identical.
#pragma once
#include <type_traits>

#include "conv_mfma.h"
#include "dec_tail_args.h"

namespace gs {


// set by a wave whose bounded mailbox wait ran out (EXCH below); read and cleared by dec_tail_fault_flags()
__device__ int g_dec_tail_fault = 0;

template <int N_>
using IC = std::integral_constant<int, N_>;
template <bool B_>
using BC = std::integral_constant<bool, B_>;

// A wave always drives P = 8 MFMA column blocks (128 columns).  NRUN of them make one strip of an image (16*NRUN input
// columns -> 16*NRUN - 2 output pixels); with NRUN < 8 the wave carries the same narrow strip of 8 / NRUN consecutive
// images side by side, so the narrow rest of a row (512 = 4 x 126 + 8) costs 1/8 of a task per image, epilogue included.
template <int CLS, int NRUN>
struct DecTailGeom {
    static constexpr int P = 8;
    static constexpr int IMGS = P / NRUN;       // images per task
    static constexpr int NG = 6;                // plane groups of four (24 planes)
    static constexpr int TS = 16 * P + 4;       // LDS tile row pitch (floats); + 4: the four k-groups of a tile write hit different banks
    static constexpr int CW = 16 * NRUN;        // tile columns of one image
    static constexpr int XS = CW - 2;           // output pixels per strip
    static constexpr int LPI = 8 * NRUN;        // lanes per image in the epilogue (two pixels per lane)
    static_assert(CLS == 5, "row layout tx*5+o is written for five classes");
    static_assert(NRUN == 1 || NRUN == 2 || NRUN == 4 || NRUN == 8, "column blocks per image strip");
};

// (everything is local arrays + generic lambdas with compile-time indices: kept in one function body so that the
// accumulators, the operand ring and the weights stay in registers)
// MODE 1 (DBG): also write the logits and the half-resolution CBR output (tests).  MODE 2 (ENS): the logits go through a softmax
// into the ensemble's probability accumulator instead of straight to the argmax.  Separate instantiations, so that the
// per-lane addresses of those outputs do not sit in registers of the mask-only kernel (MODE 0).
// timing-only ablations (results wrong by construction), -DGS_DIAG builds only
#if defined(GS_DIAG) && defined(DT_X_NOLOAD)
constexpr bool kDtNoLoad = true;
#else
constexpr bool kDtNoLoad = false;
#endif
#if defined(GS_DIAG) && defined(DT_X_NOEPI)
constexpr bool kDtNoEpi = true;
#else
constexpr bool kDtNoEpi = false;
#endif
#ifndef DT_PRIO
#define DT_PRIO 1      // second half of the workgroup's waves at s_setprio 1 (see below)
#endif
#ifndef DT_MAIN_WAVES
#define DT_MAIN_WAVES 8
#endif
// EXCH (round 5): the strips of a row are cut at multiples of 128 columns WITHOUT overlap and the waves that hold horizontally
// adjacent strips of one band -- a "team" of 1, 2, 4 or 8 waves of one workgroup -- hand each other the two values per class a
// finished row needs from across the cut (the left neighbour's tx = 0 partial sums of its last column, the right neighbour's
// tx = 2 sums of its first) through a two-slot LDS mailbox.  Against the overlapping 126-column strips this (a) covers a
// 512-column row with four strips and no narrow rest strip (one launch instead of two), (b) makes every operand load a whole
// 128-byte line (the 126-column strips start one float before a line: five lines fetched for four), and (c) puts the waves
// that touch neighbouring lines in one workgroup.  Every output is the same (z0 + z1) + z2 of the same partial sums as before:
// same bits.  A team's outer edges are the image's: the sums of the zero halo column are exact zeros and are written as such.
template <int CLS, int NRUN, int MODE, int WAVES, bool EXCH = false>
__global__ void __launch_bounds__(WAVES * 64) dec_tail_kernel(const DecTailArgs a)
{
    constexpr bool DBG = MODE == 1, ENS = MODE == 2;
    using DT = DecTailGeom<CLS, NRUN>;
    constexpr int P = DT::P, NG = DT::NG, TS = DT::TS, IMGS = DT::IMGS, CW = DT::CW, LPI = DT::LPI;
    constexpr int XS = EXCH ? CW : DT::XS;   // output pixels per strip
    static_assert(!EXCH || NRUN == 8, "the exchanging form is for full-width (128-column) strips");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (tid < 128)
        lds[tid] = tid < 116 ? a.wpack[DT_A_FLOATS + tid] : 0.0f;
    // EXCH: behind the tiles, mailbox [wave][slot = row parity][16] and one row counter per wave
    float *mbox = lds + 128 + WAVES * (16 * TS);
    int *rows_done = reinterpret_cast<int *>(mbox + WAVES * 32);
    if (EXCH && tid < WAVES)
        rows_done[tid] = 0;
    __syncthreads();
    const int lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *tile = lds + 128 + wid * (16 * TS);
    static_assert(WAVES % 4 == 0, "whole waves per SIMD");
    int epoch = 0;   // EXCH: rows this wave has finished (every wave of a team finishes the same rows in the same order)

    float A[3][NG];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            A[ty][g] = a.wpack[(ty * NG + g) * 64 + lane];
    f32x4 acc[3][P];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int p = 0; p < P; ++p)
            acc[sl][p] = (f32x4)(0.0f);
    float bq[NG][P];
    const int voff = (kq * a.in_sc + j) * 4;
    const int H = 2 * a.H1, W = 2 * a.W1;

    // tasks: (image, strip, band), the bands of one column adjacent
    // The waves that share a SIMD run the same program on equal tasks: left alone they reach their MFMA phases and their
    // (VALU / LDS) epilogues together and the matrix pipe idles through every epilogue.  With the later half at a higher
    // static priority the pair falls into anti-phase by itself: the preferred wave takes the pipe, and while it is in an
    // epilogue the other one computes.
    if (DT_PRIO && wid >= WAVES / 2)
        __builtin_amdgcn_s_setprio(1);
    // (a launch with fewer tasks than wave slots is spread over more CUs, a.wu waves of each taking tasks: launch_dec_tail_p)
    // EXCH: tasks are (image, band) pairs taken by TEAMS (a.team waves: wave t of a team owns strip t); a.wu counts the teams
    // of a workgroup that take tasks
    const int team_sz = EXCH ? a.team : 1;
    const int team = EXCH ? wid / team_sz : wid;
    const int sidx = EXCH ? wid - team * team_sz : 0;
    const bool has_left = EXCH && sidx > 0, has_right = EXCH && sidx + 1 < a.nstrips;
    const int nwaves = gridDim.x * a.wu;
    for (int task = (team < a.wu && (!EXCH || sidx < a.nstrips)) ? blockIdx.x * a.wu + team : a.total_tasks; task < a.total_tasks; task += nwaves) {
        const int col = task / a.bands;
        const int b = task - col * a.bands;
        const int ig = EXCH ? col : col / a.nstrips;
        const int s = EXCH ? sidx : col - ig * a.nstrips;
        const int n0 = ig * IMGS;                 // first image of the task
        const int nimg = min(IMGS, a.N - n0);     // (the last group may be short: its spare column blocks recompute the last image)
        // every band has exactly R = 3k+2 rows; the last one is shifted up to end at the image bottom and re-computes
        // (bit-identically re-stores) the rows it shares with its neighbour, which it must not count twice
        const int rows = a.R;
        const int yb = b * a.R;
        const int y0 = yb + rows <= a.H1 ? yb : a.H1 - rows;
        const int x0 = a.xbase + XS * s;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.in + (long long)n0 * a.in_sn), 0, a.in_img_bytes * (unsigned)nimg, 0x00020000);
        const int sbase = (a.in_off + x0 - (EXCH ? 0 : 1)) * 4;   // column x0-1 (EXCH: x0, a whole line) of row 0; row -1 is the zero halo row
        // (uniform) offset of column block p: block p % NRUN of image n0 + p / NRUN -- added to the scalar offset of a load
        int roffp[NRUN == 8 ? 1 : P];
        if (NRUN < 8) {
#pragma unroll
            for (int p = 0; p < P; ++p)
                roffp[p] = min(p / NRUN, nimg - 1) * (int)a.in_img_bytes + (p % NRUN) * 64;
        }
        const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
            a.mask ? a.mask : reinterpret_cast<unsigned char *>(const_cast<float *>(a.in)), 0,
            a.mask ? (unsigned)((long long)a.N * H * W) : 0u, 0x00020000);
        unsigned long long counts = 0;               // per-lane packed per-class counts, 12 bits each

        auto fetch = [&](auto g_, int i) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            const int soff = sbase + (i * a.in_pitch + 4 * g * a.in_sc) * 4;
#pragma unroll
            for (int p = 0; p < P; ++p)
                bq[g][p] = kDtNoLoad ? __builtin_bit_cast(float, soff + p)
                                     : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                     rs, NRUN == 8 ? voff + p * 64 : voff,
                                                                     NRUN == 8 ? soff : soff + roffp[NRUN == 8 ? 0 : p], 0));
        };
        // one input row: every plane group feeds the vertical taps whose output row lies inside the band
        auto row_step = [&](auto ph_, auto v0_, auto v1_, auto v2_, auto more_, int inext) __attribute__((always_inline)) {
            constexpr int PH = decltype(ph_)::value;
            constexpr bool V0 = decltype(v0_)::value, V1 = decltype(v1_)::value, V2 = decltype(v2_)::value;
            constexpr bool more = decltype(more_)::value;
            auto group = [&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (V0)
                        acc[PH][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][g], bq[g][p], acc[PH][p], 0, 0, 0);
                    if (V1)
                        acc[(PH + 2) % 3][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1][g], bq[g][p], acc[(PH + 2) % 3][p], 0, 0, 0);
                    if (V2)
                        acc[(PH + 1) % 3][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2][g], bq[g][p], acc[(PH + 1) % 3][p], 0, 0, 0);
                }
                // (pinned behind the MFMAs that read the slot: hoisted above them the refill needs a second register set and
                // a copy -- with a wait for the load -- at the loop head)
                __builtin_amdgcn_sched_barrier(0);
                if (more)   // the slot is refilled with the same plane group of the next input row: a whole row of MFMAs to land
                    fetch(g_, inext);
                __builtin_amdgcn_sched_barrier(0);
            };
            group(IC<0>{});
            group(IC<1>{});
            group(IC<2>{});
            group(IC<3>{});
            group(IC<4>{});
            group(IC<5>{});
        };
        // finished output row yo (half resolution) of accumulator set SL: tx merge -> BN -> PReLU -> deconv -> argmax -> stores
        auto finish_row = [&](auto sl_, int yo) __attribute__((always_inline)) {
            constexpr int SL = decltype(sl_)::value;
            if (kDtNoEpi) {
                float keep = 0.0f;
#pragma unroll
                for (int p = 0; p < P; ++p)
                    keep += acc[SL][p][0] + acc[SL][p][1] + acc[SL][p][2] + acc[SL][p][3];
                if (keep == 123.456f)
                    counts += 1;
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[SL][p] = (f32x4)(0.0f);
                return;
            }
            // the 115 epilogue constants are read from LDS every time: behind an offset the compiler cannot see through,
            // or it keeps all of them in registers across the whole kernel (115 VGPRs the accumulators need)
            int coff = 0;
            asm volatile("" : "+v"(coff));
            const float *bnl = lds + coff, *wl = lds + 16 + coff;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    tile[(4 * kq + r) * TS + 16 * p + j + (EXCH ? 1 : 0)] = acc[SL][p][r];   // (EXCH: tile column t = strip column + 1)
            // a zero the compiler cannot fold into the next MFMA's C operand: `mfma d, a, b, 0` writes a fresh register
            // range and the three accumulator sets then rotate through copies at the loop head
            float zero = 0.0f;
            asm volatile("" : "+v"(zero));
#pragma unroll
            for (int p = 0; p < P; ++p)
                acc[SL][p] = (f32x4)(zero);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (EXCH) {
                // across the cuts: tile column 0 <- the left neighbour's tx = 0 sums of ITS last column (its tile column 128),
                // tile column 129 <- the right neighbour's tx = 2 sums of its first (its column 1).  Published through a
                // mailbox slot per row parity: a wave writes row e + 2's values only after its neighbours have published row
                // e + 1, i.e. after they have read row e's.  Waits are bounded: a team's waves are all resident (one
                // workgroup) and nobody waits before publishing, so the bound is never met unless the kernel is wrong.
                const int par = epoch & 1;
                float *mine = mbox + (wid * 2 + par) * 16;
                // (the lane index behind the opaque zero: hoisted out of the band loop, the per-lane addresses of this block
                // would each hold a register for the whole kernel, and the mask-only form has none to spare)
                const int ln = lane + coff;
                if (ln < 10)
                    mine[ln] = ln < 5 ? tile[ln * TS + 128] : tile[(ln + 5) * TS + 1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0)
                    __hip_atomic_store(&rows_done[wid], epoch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                // (bounded, so that a broken protocol cannot hang the GPU -- but a wait that runs out is REPORTED: the values copied
                // below would be stale, so the wave raises the device-side fault word, which the host entries that synchronise
                // turn into GS_ERR_DEVICE_FAULT (gs_device_fault_check) instead of returning masks nobody should trust)
                bool gave_up = false;
                if (has_left) {
                    int spin = 0;
                    while (__hip_atomic_load(&rows_done[wid - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= epoch && spin < (1 << 22)) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spin;
                    }
                    gave_up = gave_up || spin >= (1 << 22);
                }
                if (has_right) {
                    int spin = 0;
                    while (__hip_atomic_load(&rows_done[wid + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= epoch && spin < (1 << 22)) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spin;
                    }
                    gave_up = gave_up || spin >= (1 << 22);
                }
                if (gave_up && lane == 0)
                    atomicOr(&g_dec_tail_fault, 1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (ln < 5)
                    tile[ln * TS] = has_left ? mbox[((wid - 1) * 2 + par) * 16 + ln] : 0.0f;
                else if (ln < 10)
                    tile[(ln + 5) * TS + 129] = has_right ? mbox[((wid + 1) * 2 + par) * 16 + ln] : 0.0f;
                ++epoch;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // lane l owns the output pixels m = 2q, 2q+1 (q = l % LPI) of the strip of image n0 + l / LPI:
            // out[o][m] = Z[o][m] + Z[5+o][m+1] + Z[10+o][m+2]
            const int gi = lane / LPI, m0 = 2 * (lane % LPI);
            const int n = n0 + gi;
            const int x = x0 + m0;
            const bool ok = m0 < XS && x < a.W1 && gi < nimg;   // W1 and x are even: the lane's two pixels are both inside or both outside
            const int mc = gi * CW + (m0 < XS ? m0 : 0);
            float f[2][CLS];
#pragma unroll
            for (int o = 0; o < CLS; ++o) {
                const float2 z0 = *reinterpret_cast<const float2 *>(&tile[o * TS + mc]);
                const float2 z1a = *reinterpret_cast<const float2 *>(&tile[(CLS + o) * TS + mc]);
                const float2 z1b = *reinterpret_cast<const float2 *>(&tile[(CLS + o) * TS + mc + 2]);
                const float2 z2 = *reinterpret_cast<const float2 *>(&tile[(2 * CLS + o) * TS + mc + 2]);
                float v0 = z0.x + z1a.y + z2.x;
                float v1 = z0.y + z1b.x + z2.y;
                v0 = v0 * bnl[o] + bnl[CLS + o];
                v1 = v1 * bnl[o] + bnl[CLS + o];
                const float al = bnl[2 * CLS + o];
                f[0][o] = prelu_med3(v0, al, prelu_pin(al));
                f[1][o] = prelu_med3(v1, al, prelu_pin(al));
            }
            __builtin_amdgcn_wave_barrier();   // the tile is rewritten by the next finished row
            if (DBG && a.ff && ok) {
#pragma unroll
                for (int o = 0; o < CLS; ++o)
                    *reinterpret_cast<float2 *>(a.ff + (long long)n * a.ff_sn + (long long)o * a.ff_sc + a.ff_off + yo * a.ff_pitch + x) =
                        make_float2(f[0][o], f[1][o]);
            }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                float lg[CLS][4];   // [class][pixel q, dx] = output columns 2x .. 2x+3
                unsigned mbytes = 0;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    // one (dy, dx) phase of the deconvolution at a time: 25 weights in flight, not all 100 -- left to
                    // itself the scheduler hoists every LDS read of the epilogue to its top, 100 registers this kernel
                    // does not have beside its three accumulator sets
                    __builtin_amdgcn_sched_barrier(0);
                    float best[2] = {0.0f, 0.0f};
                    int bi[2] = {0, 0};
#pragma unroll
                    for (int o = 0; o < CLS; ++o) {
                        float w5[CLS];
#pragma unroll
                        for (int i = 0; i < CLS; ++i)
                            w5[i] = wl[((i * CLS + o) * 2 + dy) * 2 + dx];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float t = 0.0f;
#pragma unroll
                            for (int i = 0; i < CLS; ++i)
                                t = fmaf(f[q][i], w5[i], t);
                            if (DBG || ENS)
                                lg[o][2 * q + dx] = t;
                            if (!ENS && (o == 0 || t > best[q])) {   // strict '>': the first maximum wins (torch.max semantics)
                                best[q] = t;
                                bi[q] = o;
                            }
                        }
                    }
                    if (!ENS) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            mbytes |= (unsigned)bi[q] << (8 * (2 * q + dx));
                            if (ok && yo >= yb)
                                counts += 1ull << (12 * bi[q]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ENS) {
                    // prob (+)= ens_w * softmax(logits), in the arithmetic of the two-kernel form this replaces (max-shifted
                    // expf, one division per class); the last member goes on to the first-max argmax of the sum
                    float pr[CLS][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float mx = -3.4e38f;
#pragma unroll
                        for (int o = 0; o < CLS; ++o)
                            mx = fmaxf(mx, lg[o][c]);
                        float sum = 0.0f;
#pragma unroll
                        for (int o = 0; o < CLS; ++o) {
                            pr[o][c] = expf(lg[o][c] - mx);
                            sum += pr[o][c];
                        }
#pragma unroll
                        for (int o = 0; o < CLS; ++o)
                            pr[o][c] = pr[o][c] / sum * a.ens_w;
                    }
                    // (yo >= yb: the rows a shifted last band shares with its neighbour belong to the neighbour -- a mask store may
                    // be repeated, an accumulation may not)
                    if (ok && yo >= yb) {
                        if (a.ens_mode == 2 || a.ens_mode == 3) {
#pragma unroll
                            for (int o = 0; o < CLS; ++o) {
                                const float4 old = *reinterpret_cast<const float4 *>(
                                    a.prob + (((long long)n * CLS + o) * H + 2 * yo + dy) * W + 2 * x);
                                pr[o][0] = old.x + pr[o][0];
                                pr[o][1] = old.y + pr[o][1];
                                pr[o][2] = old.z + pr[o][2];
                                pr[o][3] = old.w + pr[o][3];
                            }
                        }
                        if (a.ens_mode == 1 || a.ens_mode == 2) {
#pragma unroll
                            for (int o = 0; o < CLS; ++o)
                                *reinterpret_cast<float4 *>(a.prob + (((long long)n * CLS + o) * H + 2 * yo + dy) * W + 2 * x) =
                                    make_float4(pr[o][0], pr[o][1], pr[o][2], pr[o][3]);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float best = pr[0][c];
                                int bi = 0;
#pragma unroll
                                for (int o = 1; o < CLS; ++o)
                                    if (pr[o][c] > best) {
                                        best = pr[o][c];
                                        bi = o;
                                    }
                                mbytes |= (unsigned)bi << (8 * c);
                                counts += 1ull << (12 * bi);
                            }
                            if (a.mask)
                                __builtin_amdgcn_raw_buffer_store_b32(mbytes, rmask, n * (H * W) + 2 * x, (2 * yo + dy) * W, 2 /* nt */);
                        }
                    }
                } else if (ok) {
                    if (a.mask)   // (uniform) one 32-bit offset per lane, the row in the scalar offset
                        __builtin_amdgcn_raw_buffer_store_b32(mbytes, rmask, n * (H * W) + 2 * x, (2 * yo + dy) * W, 2 /* nt */);
                    if (DBG && a.logits) {
#pragma unroll
                        for (int o = 0; o < CLS; ++o)
                            *reinterpret_cast<float4 *>(a.logits + (((long long)n * CLS + o) * H + 2 * yo + dy) * W + 2 * x) =
                                make_float4(lg[o][0], lg[o][1], lg[o][2], lg[o][3]);
                    }
                }
            }
        };
        // Step t handles input row y0-1+t: tap ty=0 goes to output row y0+t, ty=1 to y0+t-1, ty=2 to y0+t-2; after it,
        // output row y0+t-2 is complete.  The accumulator set of output row y0+u is u % 3 and step t runs in "phase"
        // t % 3, so every register index below is a compile-time constant.  The band is a straight line: two opening
        // steps (no finished row yet), the steady loop three steps at a time, its remainder, two closing steps (no new
        // row started) -- no joins inside the loop, which the register allocator needs to keep 96 accumulators in place.
        auto do_step = [&](auto ph_, auto v0_, auto v1_, auto v2_, auto more_, int t) __attribute__((always_inline)) {
            constexpr int PH = decltype(ph_)::value;
            row_step(ph_, v0_, v1_, v2_, more_, y0 + t);
            if (decltype(v2_)::value)
                finish_row(IC<(PH + 1) % 3>{}, y0 + t - 2);
        };
        auto closing = [&](auto ph_, int t) __attribute__((always_inline)) {   // t == rows
            constexpr int PH = decltype(ph_)::value;
            do_step(ph_, BC<false>{}, BC<true>{}, BC<true>{}, BC<true>{}, t);
            do_step(IC<(PH + 1) % 3>{}, BC<false>{}, BC<false>{}, BC<true>{}, BC<false>{}, t + 1);
        };
        constexpr BC<true> T{};
        constexpr BC<false> F{};
        fetch(IC<0>{}, y0 - 1);
        fetch(IC<1>{}, y0 - 1);
        fetch(IC<2>{}, y0 - 1);
        fetch(IC<3>{}, y0 - 1);
        fetch(IC<4>{}, y0 - 1);
        fetch(IC<5>{}, y0 - 1);
        do_step(IC<0>{}, T, F, F, T, 0);
        do_step(IC<1>{}, T, T, F, T, 1);
        int t = 2;
        if (a.k3 > 0) {   // rows = 3*k3 + 2: the steady part is whole periods, no remainder
            int it = a.k3;
            do {
                do_step(IC<2>{}, T, T, T, T, t);
                do_step(IC<0>{}, T, T, T, T, t + 1);
                do_step(IC<1>{}, T, T, T, T, t + 2);
                t += 3;
            } while (--it > 0);
        }
        closing(IC<2>{}, t);
        if (a.hist) {
            // per-class totals of every image of the task: unpack, butterfly-add over the image's lanes, one atomic per class
            const int gi = lane / LPI;
#pragma unroll
            for (int k = 0; k < CLS; ++k) {
                int c = (int)((counts >> (12 * k)) & 0xfffull);
#pragma unroll
                for (int sh = LPI / 2; sh >= 1; sh >>= 1)
                    c += __shfl_xor(c, sh, 64);
                if (lane % LPI == 0 && gi < nimg && c)
                    atomicAdd(&a.hist[(long long)(n0 + gi) * CLS + k], (unsigned long long)c);
            }
        }
    }
}

template <int NRUN, int WAVES>
static gs_status launch_dec_tail_p(DecTailArgs a, int num_cus, hipStream_t stream)
{
    using DT = DecTailGeom<5, NRUN>;
    // band height R = 3k+2 (the kernel's loop is whole periods of its three rolling accumulator sets): about one round of
    // tasks over the resident waves when the batch allows it -- a task is a serial sweep, so few long tasks would leave
    // most of the chip idle; operand re-reads are (R+2)/R
    const int slots = num_cus * WAVES;
    const int cols = cdiv(a.N, DT::IMGS) * a.nstrips;
    int bands = slots / cols;
    if (bands < 1) bands = 1;
    int k3 = (cdiv(a.H1, bands) - 2 + 2) / 3;   // smallest k with 3k+2 >= H1/bands
    if (k3 < 0) k3 = 0;
    while (k3 > 0 && 3 * k3 + 2 > a.H1) --k3;   // a band never exceeds the image (H1 >= 4; k3 == 0 gives two-row bands)
    // a lane counts 8 pixels per band row into 12-bit fields: R <= 509 rows keeps an all-one-class band (4072) below 4096
    if (k3 > 169) k3 = 169;
    a.k3 = k3;
    a.R = 3 * k3 + 2;
    a.bands = cdiv(a.H1, a.R);
    a.total_tasks = cols * a.bands;
    if ((long long)a.N * 4 * a.H1 * a.W1 >= (1ll << 32) || (long long)DT::IMGS * a.in_img_bytes >= (1ll << 32)) {
        set_error("dec_tail: batch of %d tiles of %dx%d exceeds the 32-bit offsets of the mask / input descriptors", a.N, 2 * a.H1, 2 * a.W1);
        return GS_ERR_UNSUPPORTED;
    }
    const size_t lds_bytes = (size_t)(128 + WAVES * 16 * DT::TS) * sizeof(float);
    const int mode = a.ens_mode ? 2 : a.logits ? 1 : 0;
    auto kern = mode == 2 ? dec_tail_kernel<5, NRUN, 2, WAVES> : mode == 1 ? dec_tail_kernel<5, NRUN, 1, WAVES> : dec_tail_kernel<5, NRUN, 0, WAVES>;
    static std::mutex mu;
    static std::map<int, bool> attr_done;
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    dev = dev * 3 + mode;   // (device, instantiation)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_done[dev]) {
            GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            attr_done[dev] = true;
        }
    }
    int grid = num_cus;
    a.wu = WAVES;
    if (a.total_tasks < grid * WAVES) {   // small batches and the rest strips: one wave per SIMD on as many CUs as there are tasks for
        a.wu = cdiv(a.total_tasks, grid);
        grid = cdiv(a.total_tasks, a.wu);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// The exchanging form (EXCH): rows of 65 .. 1024 half-resolution pixels as ceil(W1 / 128) aligned strips taken by teams of waves,
// ONE launch, no rest strip.
#ifndef CFG_DEC_TAIL_EXCH
#define CFG_DEC_TAIL_EXCH 1
#endif
static gs_status launch_dec_tail_exch(DecTailArgs a, int num_cus, hipStream_t stream)
{
    constexpr int WAVES = 8;
    using DT = DecTailGeom<5, 8>;
    a.xbase = 0;
    a.nstrips = cdiv(a.W1, DT::CW);
    a.team = a.nstrips <= 1 ? 1 : a.nstrips <= 2 ? 2 : a.nstrips <= 4 ? 4 : 8;
    const int tpw = WAVES / a.team;             // teams per workgroup
    const int slots = num_cus * tpw;
    const int cols = a.N;                       // one image per task
    int bands = slots / cols;
    if (bands < 1) bands = 1;
    int k3 = (cdiv(a.H1, bands) - 2 + 2) / 3;   // as launch_dec_tail_p: bands of R = 3k+2 rows, about one round of tasks
    if (k3 < 0) k3 = 0;
    while (k3 > 0 && 3 * k3 + 2 > a.H1) --k3;
    if (k3 > 169) k3 = 169;                     // 12-bit count fields: R <= 509
    a.k3 = k3;
    a.R = 3 * k3 + 2;
    a.bands = cdiv(a.H1, a.R);
    a.total_tasks = cols * a.bands;
    if ((long long)a.N * 4 * a.H1 * a.W1 >= (1ll << 32)) {
        set_error("dec_tail: batch of %d tiles of %dx%d exceeds the 32-bit offsets of the mask descriptor", a.N, 2 * a.H1, 2 * a.W1);
        return GS_ERR_UNSUPPORTED;
    }
    const size_t lds_bytes = (size_t)(128 + WAVES * 16 * DT::TS + WAVES * 32 + WAVES) * sizeof(float);
    const int mode = a.ens_mode ? 2 : a.logits ? 1 : 0;
    auto kern = mode == 2 ? dec_tail_kernel<5, 8, 2, WAVES, true> : mode == 1 ? dec_tail_kernel<5, 8, 1, WAVES, true> : dec_tail_kernel<5, 8, 0, WAVES, true>;
    static std::mutex mu;
    static std::map<int, bool> attr_done;
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    dev = dev * 3 + mode;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_done[dev]) {
            GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            attr_done[dev] = true;
        }
    }
    int grid = num_cus;
    a.wu = tpw;
    if (a.total_tasks < grid * tpw) {   // small batches: as few teams per workgroup as cover the tasks on all CUs (team 0 first:
        a.wu = cdiv(a.total_tasks, grid);   // its waves sit on different SIMDs)
        grid = cdiv(a.total_tasks, a.wu);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Full-width strips (126 output columns, eight MFMA column blocks) in one launch; the narrow rest of the row, if any,
// in a second one that packs the rest strips of 8 / NRUN images into every task.  (Round 4 built the rest strips into the main
// launch twice -- as a second task list of the same waves, and as extra workgroups behind the main grid on a block-uniform
// branch: 0.162-0.165 and 0.1618 ms against 0.1572 for the two launches (profiles/r04_ab_dec_tail_merge.txt).  Either way
// the mask-only kernel, which holds 96 accumulators in 256 registers without a spill, started to spill (24-60 bytes), and its
// 2048 main tasks end together, so the rest tasks find no idle tail to fill.  Not kept.)
gs_status launch_dec_tail(DecTailArgs a, int num_cus, hipStream_t stream)
{
    // rows of 65 .. 1024 pixels: aligned 128-column strips, neighbours exchange across the cuts (one launch).  Narrower rows
    // pack several images into a wave, wider ones have more strips than a workgroup has waves: the overlapping strips below.
    if (CFG_DEC_TAIL_EXCH && a.W1 > 64 && a.W1 <= 1024)
        return launch_dec_tail_exch(a, num_cus, stream);
    constexpr int XS = DecTailGeom<5, 8>::XS;
    const int full_strips = a.W1 / XS, rest = a.W1 - full_strips * XS;
    if (full_strips > 0) {
        a.xbase = 0;
        a.nstrips = full_strips;
        gs_status st = launch_dec_tail_p<8, DT_MAIN_WAVES>(a, num_cus, stream);
        if (st != GS_OK) return st;
    }
    if (rest > 0) {
        a.xbase = full_strips * XS;
        a.nstrips = 1;
        const int nrun = cdiv(rest + 2, 16);
        if (nrun <= 1) return launch_dec_tail_p<1, 8>(a, num_cus, stream);
        if (nrun <= 2) return launch_dec_tail_p<2, 8>(a, num_cus, stream);
        if (nrun <= 4) return launch_dec_tail_p<4, 8>(a, num_cus, stream);
        return launch_dec_tail_p<8, 8>(a, num_cus, stream);
    }
    return GS_OK;
}

}  // namespace gs
