// Bandwidth-bound (non-GEMM) stages of the ESPNet forward as fused VALU kernels.  Each kernel
// names the reference lines it fuses.  Weights are read with wave-uniform indices, so hipcc emits
// scalar (SMEM) loads for them; activations are read coalesced along x.
#pragma once
#include <type_traits>
#include "gs_internal.h"

namespace gs {

struct ActV {   // device-side view of gs::Act
    float *base;
    long long sn;
    int sc, pitch, off;
    int H, W;
};
static inline ActV view(const Act &a) { return ActV{a.base, a.sn, a.sc, a.pitch, a.off, a.H, a.W}; }

__device__ __forceinline__ float *at(const ActV &t, int n, int c, int y, int x)
{
    return t.base + (long long)n * t.sn + (long long)c * t.sc + t.off + y * t.pitch + x;
}
// zero-padding read (conv / pool padding semantics).  Branch-free on purpose: the coordinate is
// clamped, the load is unconditional and the result is selected afterwards, so a thread's taps are
// all in flight together (a branch per tap made hipcc wait vmcnt(0) after every load).
__device__ __forceinline__ float ldz(const ActV &t, int n, int c, int y, int x)
{
    const bool ok = (unsigned)y < (unsigned)t.H && (unsigned)x < (unsigned)t.W;
    const int yc = min(max(y, 0), t.H - 1), xc = min(max(x, 0), t.W - 1);
    const float v = *at(t, n, c, yc, xc);
    return ok ? v : 0.0f;
}
// Once-only streams are marked non-temporal (see F_RES_NT in conv_mfma.h); per-kernel switches for measurement:
// dec2 0.117 -> 0.100 ms, dec1 0.070 -> 0.067, the stride-2 reduce after the stem 0.107 -> 0.095, dec4 unchanged.
#ifndef NT_STEM_ST
#define NT_STEM_ST 1
#endif
#ifndef NT_DEC1_LD
#define NT_DEC1_LD 1
#endif
#ifndef NT_DEC2_LD
#define NT_DEC2_LD 1
#endif
#ifndef NT_DEC4_ST
#define NT_DEC4_ST 1
#endif
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float ld_stream(const float *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void st_stream(T *p, T v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// PReLU as max/min arithmetic (bit-identical to the select: one of the two terms is always zero).  Written as
// `v > 0 ? v : alpha * v` with alpha behind a pointer, hipcc branches around a scalar load of alpha and waits
// for it inside the branch: one serialised round trip per activation.
__device__ __forceinline__ float prelu(float v, float alpha)
{
    // median form (conv_mfma.h, prelu_med3): alpha is uniform here, so the +-inf pin is a scalar select
    return __builtin_amdgcn_fmed3f(v, alpha * v, alpha <= 1.0f ? __builtin_inff() : -__builtin_inff());
}
__device__ __forceinline__ float bn_prelu(float v, const float *bnp, int C, int c)
{
    return prelu(v * bnp[c] + bnp[C + c], bnp[2 * C + c]);
}
// select form for the stem, where it measured faster (0.122 vs 0.131 ms): that kernel is short of scalar-load
// bandwidth (432 weights per wave) and the branch skips the alpha loads of all-positive waves
__device__ __forceinline__ float bn_prelu_sel(float v, const float *bnp, int C, int c)
{
    v = v * bnp[c] + bnp[C + c];
    return v > 0.0f ? v : bnp[2 * C + c] * v;
}

// ---------------------------------------------------------------------------------------------
// Stem: normalise -> level1 CBR(3,16,3,2) -> sample1 avg-pool -> cat -> b1 BR(19)
// reference: VisualizeResults_iou.py:107-117, Model.py:346-350 (278-282), CBR :24-32, BR :47-54,
// InputProjectionA :232-239.
struct StemArgs {
    const void *in;          // uint8 NHWC BGR or fp32 NCHW
    float mean[3], std[3];
    // The 537 parameters travel BY VALUE in the kernel-argument segment: read-only and alias-free by construction, so hipcc
    // fetches them with wide scalar loads well ahead of their use.  Behind pointers they arrived as 158 single-dword scalar
    // loads per wave, each waited for where it was used, and the kernel was bound by that latency chain.
    float w1[432];           // level1.conv.weight [16][3][3][3]
    float bn1[48];           // level1 bn+act folded [3][16]
    float b1[57];            // b1 folded [3][19]
    ActV a0;                 // out: output0_cat, 19 channels, (H/2 x W/2)
    ActV inp1;               // out: raw pooled input, 3 channels (feeds sample2's second pool)
    unsigned long long *hist_zero;   // optional: per-class counters the decoder tail adds into, zeroed here (first kernel of the
    int hist_count;                  // forward) instead of by a separate fill launch
    int N, H, W;             // INPUT size
};

// uint8 -> normalised fp32 goes through a 3x256 table built once per workgroup with exactly the
// reference's three fp32 roundings (numpy: x - mean, / std, / 255; VisualizeResults_iou.py:109,111,116),
// so it is bit-identical to computing them per pixel but costs one LDS read instead of two IEEE
// divisions per value.
template <bool U8>
__device__ __forceinline__ float stem_fetch(const StemArgs &a, const float (*lut)[256], int n, int c, int y, int x)
{
    // branch-free (see ldz): clamp, load, select; padding acts on the normalised tensor
    const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
    float v;
    if (U8)
        v = lut[c][static_cast<const unsigned char *>(a.in)[(((long long)n * a.H + yc) * a.W + xc) * 3 + c]];
    else
        v = static_cast<const float *>(a.in)[(((long long)n * 3 + c) * a.H + yc) * a.W + xc];
    return ok ? v : 0.0f;
}

// Two horizontally adjacent output pixels per thread: the 3x5 input window is fetched once (15 values per channel
// instead of 18), every weight that arrives in an SGPR feeds two FMAs, and the 19 + 3 results leave as 8-byte stores.
constexpr int STEM_PX = 2;
// uint8 input: a thread's window row is 15 consecutive bytes (5 pixels x BGR) that start one byte past a dword boundary
// (x is even, so byte 6x - 3), i.e. bytes 1..15 of four consecutive dwords: 12 dword loads + 45 bit-field extracts per thread
// instead of 45 single-byte loads with a 64-bit address, a clamp and a select each -- the kernel was VALU-bound on exactly
// that bookkeeping (1 935 VALU instructions per wave for 864 FMAs): 0.1227 -> 0.0850 ms per launch at batch 32
// (profiles/README.md, round 3).  STEM_LUT_COPIES lane-swizzled copies of the table (copy = lane % COPIES; a ds_read_b32
// banks by (a/4) % 32 within each 32-lane half) were measured on top: 1 copy 0.0850, 8 copies 0.0876, 16 copies 0.1090 ms --
// the bank conflicts of the data-dependent lookups are not what bounds the kernel, the extra address arithmetic and table
// fill cost more than they save.  One copy ships.
#ifndef STEM_DWORD
#define STEM_DWORD 1
#endif
#ifndef STEM_LUT_COPIES
#define STEM_LUT_COPIES 1
#endif
template <bool U8>
__global__ void __launch_bounds__(256) stem_kernel(const StemArgs a)
{
    constexpr int NC = (U8 && STEM_DWORD) ? STEM_LUT_COPIES : 1;
    __shared__ float lut[3][256 * NC];
    if (U8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)threadIdx.x;
            v = v - a.mean[c];
            v = v / a.std[c];
            v = v / 255.0f;
#pragma unroll
            for (int k = 0; k < NC; ++k)
                lut[c][threadIdx.x * NC + k] = v;
        }
        __syncthreads();
    }
    if (a.hist_zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.hist_count; i += 256)
            a.hist_zero[i] = 0ull;
    const int H1 = a.H / 2, W1 = a.W / 2;
    const int W1p = (W1 + STEM_PX - 1) / STEM_PX;   // pixel pairs per row
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.N * H1 * W1p)
        return;
    const int x = (int)(idx % W1p) * STEM_PX;
    const int y = (int)((idx / W1p) % H1);
    const int n = (int)(idx / ((long long)W1p * H1));
    const bool two = x + 1 < W1;

    float v[3][3][2 * STEM_PX + 1];
    if (U8 && STEM_DWORD) {
        // rows 2y-1 .. 2y+1, bytes [6x-4, 6x+12) of each (W is a multiple of 8: rows start dword-aligned and the last
        // thread of a row ends exactly at the row's end)
        const unsigned char *img = static_cast<const unsigned char *>(a.in) + (long long)n * a.H * a.W * 3;
        const int cp = (threadIdx.x & (NC - 1));
        unsigned d[3][4];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = min(max(2 * y - 1 + ky, 0), a.H - 1);
            const unsigned *row = reinterpret_cast<const unsigned *>(img + (long long)yy * a.W * 3) + (x > 0 ? (6 * x - 4) / 4 : 0);
            // (unaligned-to-16 but dword-aligned: four dword loads the compiler may merge)
            d[ky][0] = row[0];
            d[ky][1] = row[1];
            d[ky][2] = row[2];
            d[ky][3] = row[3];
            if (x == 0) {   // no column -1: the row was fetched from its first byte, one dword further right
                d[ky][3] = d[ky][2];
                d[ky][2] = d[ky][1];
                d[ky][1] = d[ky][0];
                d[ky][0] = 0u;
            }
        }
        const bool top = y == 0, left = x == 0;   // the only padding a window can meet (H, W even; x even)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2 * STEM_PX + 1; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int pos = 1 + kx * 3 + c;                       // byte 1..15 of the 16
                    const unsigned b = (d[ky][pos >> 2] >> (8 * (pos & 3))) & 0xffu;
                    float t = lut[c][b * NC + cp];
                    if ((ky == 0 && top) || (kx == 0 && left))
                        t = 0.0f;                                          // padding acts on the normalised tensor
                    v[c][ky][kx] = t;
                }
    } else if (!U8 && STEM_DWORD) {
        // fp32 NCHW input (the tensor the crop pipeline resamples into): five consecutive floats per window row and channel,
        // the same bookkeeping-free form -- only the top row and the left column can be padding
        const float *img = static_cast<const float *>(a.in) + (long long)n * 3 * a.H * a.W;
        const bool top = y == 0, left = x == 0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = max(2 * y - 1 + ky, 0);
                const float *row = img + ((long long)c * a.H + yy) * a.W + (left ? 0 : 2 * x - 1);
                float r[5];
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
                    r[kx] = row[kx];          // (x == 0: columns 0..4, one further right; W >= 8)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    float t = left ? (kx > 0 ? r[kx - 1] : 0.0f) : r[kx];
                    if (ky == 0 && top)
                        t = 0.0f;
                    v[c][ky][kx] = t;
                }
            }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2 * STEM_PX + 1; ++kx)
                    v[c][ky][kx] = stem_fetch<U8>(a, reinterpret_cast<const float(*)[256]>(&lut[0][0]), n, c, 2 * y - 1 + ky, 2 * x - 1 + kx);
    }

    // all arithmetic first, all stores last: a store between two weight reads would force hipcc to
    // re-read the (possibly aliasing) weights from memory with vector loads and a full wait each time
    float outv[19][STEM_PX], poolv[3][STEM_PX];
    // (two output channels per v_pk_fma_f32 with the weights as SGPR pairs -- half the VALU instructions -- measured slower,
    // 0.130 vs 0.122 ms: the kernel is bound by its load -> table -> arithmetic -> store latency chain, not by VALU issue)
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float s[STEM_PX];
#pragma unroll
        for (int q = 0; q < STEM_PX; ++q)
            s[q] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wgt = a.w1[((o * 3 + c) * 3 + ky) * 3 + kx];
#pragma unroll
                    for (int q = 0; q < STEM_PX; ++q)
                        s[q] = fmaf(wgt, v[c][ky][2 * q + kx], s[q]);
                }
#pragma unroll
        for (int q = 0; q < STEM_PX; ++q)
            outv[o][q] = bn_prelu_sel(bn_prelu_sel(s[q], a.bn1, 16, o), a.b1, 19, o);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < STEM_PX; ++q) {
            float s = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    s += v[c][ky][2 * q + kx];
            s = s / 9.0f;   // count_include_pad=True
            poolv[c][q] = s;
            outv[16 + c][q] = bn_prelu_sel(s, a.b1, 19, 16 + c);
        }
    // interior rows start on a 128-byte line and x is even: the pair is one aligned 8-byte store
#pragma unroll
    for (int o = 0; o < 19; ++o) {
        if (two)
            st_stream<NT_STEM_ST>(reinterpret_cast<f32x2_t *>(at(a.a0, n, o, y, x)), f32x2_t{outv[o][0], outv[o][1]});
        else
            st_stream<NT_STEM_ST>(at(a.a0, n, o, y, x), outv[o][0]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (two)
            st_stream<NT_STEM_ST>(reinterpret_cast<f32x2_t *>(at(a.inp1, n, c, y, x)), f32x2_t{poolv[c][0], poolv[c][1]});
        else
            st_stream<NT_STEM_ST>(at(a.inp1, n, c, y, x), poolv[c][0]);
    }
}

// second AvgPool2d(3,2,1) of sample2.  reference: Model.py:232-239,348
__global__ void __launch_bounds__(256)
pool_kernel(const ActV in, const ActV out, int N, int C, const float *bnp2, const ActV out2, int coff2, int C2)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * C * out.H * out.W)
        return;
    const int x = (int)(idx % out.W);
    const int y = (int)((idx / out.W) % out.H);
    const int c = (int)((idx / ((long long)out.W * out.H)) % C);
    const int n = (int)(idx / ((long long)out.W * out.H * C));
    float s = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            s += ldz(in, n, c, 2 * y - 1 + ky, 2 * x - 1 + kx);
    s = s / 9.0f;
    *at(out, n, c, y, x) = s;
    if (bnp2)   // the inp2 slice of b2 = BR(131) over cat([output1, output1_0, inp2]) (Model.py:359)
        *at(out2, n, coff2 + c, y, x) = bn_prelu(s, bnp2, C2, coff2 + c);
}

// b2: cat([output1, output1_0, inp2]) -> BR(131).  reference: Model.py:359 (291)
__global__ void __launch_bounds__(256)
cat_b2_kernel(const ActV o1, const ActV o10, const ActV inp2, const float *bnp, const ActV out, int N)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * 131 * out.H * out.W)
        return;
    const int x = (int)(idx % out.W);
    const int y = (int)((idx / out.W) % out.H);
    const int c = (int)((idx / ((long long)out.W * out.H)) % 131);
    const int n = (int)(idx / ((long long)out.W * out.H * 131));
    const float v = c < 64 ? *at(o1, n, c, y, x) : c < 128 ? *at(o10, n, c - 64, y, x) : *at(inp2, n, c - 128, y, x);
    *at(out, n, c, y, x) = bn_prelu(v, bnp, 131, c);
}

// b3 -> encoder classifier -> (br BN -> up_l3 deconv).  One thread per 1/8-scale pixel.
// reference: Model.py:368-370 (b3: cat([output2_0, output2]) -> BR(256); classifier C(256,classes,1);
// br = BatchNorm2d(classes); up_l3 = ConvTranspose2d(classes,classes,2,stride=2)).
// CLASS COUNTS.  The decoder kernels are instantiated for a PADDED class count CLS -- 5 (the shipped networks: the fast path, whose
// code and bits are those of rounds 1-4) or a multiple of four up to 20 -- and take the model's real count (Model.py:311: any
// `classes`, 20 by default) at run time: weights, BN parameters and activation planes beyond it are zero (packed so by
// gs_espnet_create), stores and the argmax stop at it.  A zero term leaves an fmaf chain unchanged, so a real class's value is
// the one an exact-width kernel would compute.
template <int CLS>
__device__ __forceinline__ int real_classes(int classes)
{
    return CLS == 5 ? 5 : classes;   // (folds away on the fast path)
}
// The padded lanes of a class vector are zero only as long as every activation that met their zero weights was finite:
// 0 * inf = NaN would sit in a padding lane and -- through the deconvolutions below, which mix ALL input lanes into every
// output class -- reach the real classes, where an exact-width kernel (and the reference) would confine a non-finite value to
// the classes that actually use it.  So the lanes beyond the model's count are re-zeroed in front of every such mixing chain.
template <int CLS>
__device__ __forceinline__ void zero_padding_lanes(float *s, int ncls)
{
    if constexpr (CLS != 5) {
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            s[k] = k < ncls ? s[k] : 0.0f;
    }
}
// floats per channel record [scale, shift, alpha, w[0..CLS)] of dec1, and per weight row of dec2: whole float4s
template <int CLS>
constexpr int dec1_record() { return (3 + CLS + 3) / 4 * 4; }
template <int CLS>
constexpr int dec2_record() { return (CLS + 3) / 4 * 4; }

struct Dec1Args {
    ActV c0, clast;      // output2_0, output2 (128 channels each)
    const float *b3w;    // [256][dec1_record]: folded b3 {scale, shift, alpha} + encoder.classifier.conv.weight[0..CLS)[c]
    const float *br;     // folded br scale/shift [2][CLS]
    const float *wup;    // up_l3.0.weight [CLS][CLS][2][2]
    ActV out;            // output2_c: CLS channels at 1/4 scale
    float *enc_logits;   // encoder-only mode: [N][classes][H3][W3], else null
    int N;
    int classes;         // the model's class count (<= CLS)
};

// A workgroup = 64 pixels x 4 waves: wave w sums channels 64w .. 64w+63 of its lane's pixel, the four partial sums meet in
// LDS and are added as (q0 + q1) + (q2 + q3) -- at every batch size, so a tile's bits do not depend on the batch.  (Round 3
// summed the 256 channels in one chain per thread: at one tile that chain is the kernel -- 0.036 ms, 140 ns per channel
// whatever the constants came from (scalar loads, LDS, v_readlane: all measured) -- every load of a wave touches another
// channel plane, 33 KB from the last.)  Batches of CB channels: all their loads are issued before the first use, and the
// PReLU is written max/min so that it stays straight-line (as `v > 0 ? v : alpha * v` hipcc branched around the scalar load
// of alpha, which put one full memory round trip per channel on the critical path).
template <int CLS, int CB>
__global__ void __launch_bounds__(256) dec1_kernel(const Dec1Args a)
{
    const int H3 = a.c0.H, W3 = a.c0.W;
    __shared__ float part[3][CLS][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long total = (long long)a.N * H3 * W3;
    long long idx = (long long)blockIdx.x * 64 + lane;
    const bool active = idx < total;
    if (!active)
        idx = total - 1;    // (every thread reaches the barrier; it computes a pixel again and stores nothing)
    const int x = (int)(idx % W3);
    const int y = (int)((idx / W3) % H3);
    const int n = (int)(idx / ((long long)W3 * H3));
    float s[CLS];
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = 0.0f;
    // output2_0 (channels 0..127) and output2 (128..255) share one layout (both are level-3 ping-pong buffers)
    const long long pix = (long long)n * a.c0.sn + a.c0.off + y * a.c0.pitch + x;
    const float *plane0 = (w < 2 ? a.c0.base : a.clast.base) + pix + (long long)((w & 1) * 64) * a.c0.sc;
    auto fetch = [&](int c0, float *dst) {
#pragma unroll
        for (int j = 0; j < CB; ++j)
            dst[j] = ld_stream<NT_DEC1_LD>(plane0 + (long long)(c0 + j) * a.c0.sc);
    };
    float cur[CB], nxt[CB];
    fetch(0, cur);
    for (int c0 = 0; c0 < 64; c0 += CB) {
        if (c0 + CB < 64)
            fetch(c0 + CB, nxt);   // the next batch is in flight while this one is consumed
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            // per-channel constants packed [c][8] = {scale, shift, alpha, w[0..CLS)}: one scalar load per channel
            const float *pc = a.b3w + (w * 64 + c0 + j) * dec1_record<CLS>();
            const float t = cur[j] * pc[0] + pc[1];
            const float v = prelu(t, pc[2]);
#pragma unroll
            for (int k = 0; k < CLS; ++k)
                s[k] = fmaf(pc[3 + k], v, s[k]);
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
            cur[j] = nxt[j];
    }
    if (w > 0) {
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            part[w - 1][k][lane] = s[k];
    }
    __syncthreads();
    if (w > 0 || !active)
        return;
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = (s[k] + part[0][k][lane]) + (part[1][k][lane] + part[2][k][lane]);
    const int ncls = real_classes<CLS>(a.classes);
    if (a.enc_logits) {
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            if (k < ncls)
                a.enc_logits[(((long long)n * ncls + k) * H3 + y) * W3 + x] = s[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = s[k] * a.br[k] + a.br[CLS + k];
    zero_padding_lanes<CLS>(s, ncls);
    if constexpr (CLS != 5) {
        // one output class at a time (a rolled loop): unrolled, the CLS * CLS * 4 weights were all fetched up front -- 1 600
        // registers' worth at twenty classes, 4 KB of scratch per lane
        for (int o = 0; o < ncls; ++o) {
            float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < CLS; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    t[k] = fmaf(s[i], a.wup[(i * CLS + o) * 4 + k], t[k]);
            *reinterpret_cast<float2 *>(at(a.out, n, o, 2 * y, 2 * x)) = make_float2(t[0], t[1]);
            *reinterpret_cast<float2 *>(at(a.out, n, o, 2 * y + 1, 2 * x)) = make_float2(t[2], t[3]);
        }
        return;
    }
    float up[CLS][2][2];   // arithmetic first, stores last (see stem_kernel)
#pragma unroll
    for (int o = 0; o < CLS; ++o)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float t = 0.0f;
#pragma unroll
                for (int i = 0; i < CLS; ++i)
                    t = fmaf(s[i], a.wup[((i * CLS + o) * 2 + dy) * 2 + dx], t);
                up[o][dy][dx] = t;
            }
#pragma unroll
    for (int o = 0; o < CLS; ++o)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
            *reinterpret_cast<float2 *>(at(a.out, n, o, 2 * y + dy, 2 * x)) = make_float2(up[o][dy][0], up[o][dy][1]);
}

// level3_C 1x1 on output1_cat, cat with output2_c, BR(2*classes).  One thread per 1/4-scale pixel.
// reference: Model.py:372-373 (level3_C = C(131,classes,1); combine_l2_l3[0] = BR(2*classes))
struct Dec2Args {
    ActV a1;             // output1_cat (131 channels)
    ActV raw;            // lazy b2: planes raw_c0 .. raw_c0 + raw_cn - 1 of output1_cat are not materialised; they are
    const float *b2;     // BN + PReLU (b2 folded [3][131]) of this raw block output, applied here on load
    int raw_c0, raw_cn;
    ActV o2c;            // output2_c (CLS)
    const float *w3c;    // level3_C.conv.weight packed [131][dec2_record] (first CLS of each row used)
    const float *br;     // combine_l2_l3.0 folded [3][2*CLS]: padded channel k <-> cat channel k (level3_C's), CLS + k <-> classes + k
    ActV t;              // out: 2*CLS planes in that padded order
    int N;
    int classes;
};

// A workgroup = 64 pixels x 4 waves, as dec1_kernel: wave w sums channels 32w .. 32w+31 (wave 3 goes on through 128..130), the
// partial sums are added as (q0 + q1) + (q2 + q3) at every batch size.  A wave's 32 channels are all raw or all normalised
// (raw_c0 = 64, raw_cn = 0 or 64: checked by the caller), so the BN + PReLU of the lazy b2 is a wave-uniform choice.
template <int CLS>
__global__ void __launch_bounds__(256) dec2_kernel(const Dec2Args a)
{
    const int H2 = a.a1.H, W2 = a.a1.W;
    __shared__ float part[3][CLS][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long total = (long long)a.N * H2 * W2;
    long long idx = (long long)blockIdx.x * 64 + lane;
    const bool active = idx < total;
    if (!active)
        idx = total - 1;    // (every thread reaches the barrier)
    const int x = (int)(idx % W2);
    const int y = (int)((idx / W2) % H2);
    const int n = (int)(idx / ((long long)W2 * H2));
    float s[CLS];
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = 0.0f;
    const int cb = 32 * w;
#if defined(GS_DIAG) && defined(DEC2_X_NOCAT)
    // gain ceiling for "level3_C computed elsewhere" (profiles/r05_ab_level3c_fusion.txt; results wrong by construction): dec2
    // without its read of the 131 planes of output1_cat -- five planes of a precomputed level3_C output stand in
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            s[k] = ld_stream<NT_DEC2_LD>(at(a.a1, n, k, y, x));
    }
    if (false)
#endif
    if (cb >= a.raw_c0 && cb < a.raw_c0 + a.raw_cn) {   // (uniform) lazy b2: these planes are raw; BN + PReLU of the cat's BR here
#pragma unroll 16
        for (int c = cb; c < cb + 32; ++c) {
            const float v = bn_prelu(ld_stream<NT_DEC2_LD>(at(a.raw, n, c - a.raw_c0, y, x)), a.b2, 131, c);
            const float *pc = a.w3c + c * dec2_record<CLS>();   // level3_C weights packed [c][record]: one scalar load per channel
#pragma unroll
            for (int k = 0; k < CLS; ++k)
                s[k] = fmaf(pc[k], v, s[k]);
        }
    } else {
#pragma unroll 16
        for (int c = cb; c < cb + 32; ++c) {
            const float v = ld_stream<NT_DEC2_LD>(at(a.a1, n, c, y, x));
            const float *pc = a.w3c + c * dec2_record<CLS>();
#pragma unroll
            for (int k = 0; k < CLS; ++k)
                s[k] = fmaf(pc[k], v, s[k]);
        }
    }
#if defined(GS_DIAG) && defined(DEC2_X_NOCAT)
    if (false)
#endif
    if (w == 3) {
#pragma unroll
        for (int c = 128; c < 131; ++c) {
            const float v = ld_stream<NT_DEC2_LD>(at(a.a1, n, c, y, x));
            const float *pc = a.w3c + c * dec2_record<CLS>();
#pragma unroll
            for (int k = 0; k < CLS; ++k)
                s[k] = fmaf(pc[k], v, s[k]);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            part[w - 1][k][lane] = s[k];
    }
    __syncthreads();
    if (w > 0 || !active)
        return;
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = (s[k] + part[0][k][lane]) + (part[1][k][lane] + part[2][k][lane]);
    float tv[2 * CLS];
#pragma unroll
    for (int k = 0; k < CLS; ++k) {
        tv[k] = bn_prelu(s[k], a.br, 2 * CLS, k);
        tv[CLS + k] = bn_prelu(*at(a.o2c, n, k, y, x), a.br, 2 * CLS, CLS + k);
    }
    const int ncls = real_classes<CLS>(a.classes);
#pragma unroll
    for (int k = 0; k < 2 * CLS; ++k)
        if (k % CLS < ncls)
            *at(a.t, n, k, y, x) = tv[k];
}

// combine_l2_l3[1] CBR(2*classes,classes,3) -> up_l2 deconv -> BR(classes).  One thread per
// 1/4-scale pixel.  reference: Model.py:373 (335,337)
struct Dec3Args {
    ActV t;              // 2*CLS channels
    const float *wc;     // combine_l2_l3.1.conv.weight [CLS][2*CLS][3][3]; for CLS != 5 repacked [2*CLS][3][3][CLS]: the CLS weights of a
                         // (plane, tap) are one run of scalar loads instead of CLS loads 18*CLS floats apart
    const float *bnc;    // combine_l2_l3.1 bn+act folded [3][CLS]
    const float *wup;    // up_l2.0.weight [CLS][CLS][2][2]
    const float *bnu;    // up_l2.1 folded [3][CLS]
    ActV e;              // out: comb_l2_l3 at 1/2 scale, CLS channels
    int N;
    int classes;
};

template <int CLS>
__global__ void __launch_bounds__(256) dec3_kernel(const Dec3Args a)
{
    const int H2 = a.t.H, W2 = a.t.W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.N * H2 * W2)
        return;
    const int x = (int)(idx % W2);
    const int y = (int)((idx / W2) % H2);
    const int n = (int)(idx / ((long long)W2 * H2));
    float s[CLS];
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = 0.0f;
    const int ncls = real_classes<CLS>(a.classes);
    for (int c = 0; c < 2 * CLS; ++c) {
        if (CLS != 5 && c % CLS >= ncls)
            continue;   // (uniform) a padding plane: zeros times zero weights
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = ldz(a.t, n, c, y - 1 + ky, x - 1 + kx);
#pragma unroll
                for (int k = 0; k < CLS; ++k)
                    s[k] = fmaf(CLS == 5 ? a.wc[((k * 2 * CLS + c) * 3 + ky) * 3 + kx] : a.wc[((c * 3 + ky) * 3 + kx) * CLS + k], v, s[k]);
            }
    }
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = bn_prelu(s[k], a.bnc, CLS, k);
    zero_padding_lanes<CLS>(s, ncls);
    float up[CLS][2][2];
#pragma unroll
    for (int o = 0; o < CLS; ++o)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float t = 0.0f;
#pragma unroll
                for (int i = 0; i < CLS; ++i)
                    t = fmaf(s[i], a.wup[((i * CLS + o) * 2 + dy) * 2 + dx], t);
                up[o][dy][dx] = bn_prelu(t, a.bnu, CLS, o);
            }
#pragma unroll
    for (int o = 0; o < CLS; ++o)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
            if (o < ncls)
                *reinterpret_cast<float2 *>(at(a.e, n, o, 2 * y + dy, 2 * x)) = make_float2(up[o][dy][0], up[o][dy][1]);
}

// up_l2 alone: ConvTranspose2d(classes,classes,2,2) -> BR(classes) on the output of combine_l2_l3.1 when that convolution ran on the
// matrix cores (twelve classes and more; a.t = its output, CLS planes).  reference: Model.py:373 (337)
template <int CLS>
__global__ void __launch_bounds__(256) dec3b_kernel(const Dec3Args a)
{
    const int H2 = a.t.H, W2 = a.t.W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.N * H2 * W2)
        return;
    const int x = (int)(idx % W2);
    const int y = (int)((idx / W2) % H2);
    const int n = (int)(idx / ((long long)W2 * H2));
    const int ncls = real_classes<CLS>(a.classes);
    float s[CLS];
#pragma unroll
    for (int k = 0; k < CLS; ++k)
        s[k] = *at(a.t, n, k, y, x);
    zero_padding_lanes<CLS>(s, ncls);
    for (int o = 0; o < ncls; ++o) {   // (a rolled loop, as dec1's: CLS * CLS * 4 weights fetched up front would not fit the registers)
        float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < CLS; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t[k] = fmaf(s[i], a.wup[(i * CLS + o) * 4 + k], t[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            t[k] = bn_prelu(t[k], a.bnu, CLS, o);
        *reinterpret_cast<float2 *>(at(a.e, n, o, 2 * y, 2 * x)) = make_float2(t[0], t[1]);
        *reinterpret_cast<float2 *>(at(a.e, n, o, 2 * y + 1, 2 * x)) = make_float2(t[2], t[3]);
    }
}

// classifier ConvTranspose2d(classes,classes,2,2) -> logits -> first-max argmax -> uint8 mask ->
// per-class pixel counts, on the output of conv CBR(19+classes,classes,3) (which runs on the matrix
// cores, a conv_mfma_kernel instantiation).  One thread per 1/2-scale pixel (= a 2x2 block of output pixels).
// reference: Model.py:377; VisualizeResults_iou.py:128 (argmax), :151-155 (counts)
// This two-kernel tail is what every class count other than 5 runs (the five-class networks take dec_tail.h's fused kernel):
// round 1's tail, generalised.  ENS: the ensemble of BASELINE cfg 5 (definition in DESIGN.md) -- prob [N][classes][H][W]
// accumulates ens_w * softmax(logits) over the member models exactly as dec_tail_kernel does for five classes (mode 1: first
// member stores, 2: a middle member adds, 3: the last adds and goes on to the first-max argmax of the sum, 4: a single member).
struct Dec4Args {
    ActV f;              // concat_features: conv CBR output, CLS channels at 1/2 scale
    const float *wcl;    // classifier.weight [CLS][CLS][2][2]
    float *logits;       // [N][classes][H][W] or null
    unsigned char *mask; // [N][H][W] or null
    unsigned long long *hist;   // [N][classes] or null
    float *prob;         // ENS: the accumulator
    int ens_mode;
    float ens_w;
    int N;
    int classes;
};

// half-scale pixels per thread: four for few classes (see below); with many classes one pixel's deconvolution is already CLS^2 * 4
// FMAs and the unrolled body of four would be too large to unroll at all (the per-pixel arrays would then live in scratch)
template <int CLS>
constexpr int dec4_px() { return CLS <= 8 ? 4 : CLS <= 12 ? 2 : 1; }
template <int CLS, bool ENS>
__global__ void __launch_bounds__(256) dec4_kernel(const Dec4Args a)
{
    __shared__ unsigned int lhist[CLS];
    const int H1 = a.f.H, W1 = a.f.W;
    const int H = 2 * H1, W = 2 * W1;
    const int n = blockIdx.y;   // one block never straddles two images
    const int ncls = real_classes<CLS>(a.classes);
    constexpr int DEC4_PX = dec4_px<CLS>();
    // classifier weights through LDS: read from global memory they were re-fetched with vector loads after every
    // mask / logits store (possible aliasing), 395 loads per thread
    __shared__ float wl[CLS * CLS * 4];
    for (int i = threadIdx.x; i < CLS * CLS * 4; i += 256)
        wl[i] = a.wcl[i];
    if (threadIdx.x < CLS)
        lhist[threadIdx.x] = 0;
    __syncthreads();
    // DEC4_PX half-scale pixels per thread, 256 apart, all their inputs requested before the first use: the kernel
    // is a chain of dependent round trips (inputs -> arithmetic -> stores -> ballots -> atomics) and one pixel per
    // thread left the memory system idle most of the time (0.085 ms for 0.1 GB)
    int cls_of[DEC4_PX][4];
    float s[DEC4_PX][CLS];
    bool live[DEC4_PX];
    int xs[DEC4_PX], ys[DEC4_PX];
#pragma unroll
    for (int q = 0; q < DEC4_PX; ++q) {
        const int idx = (blockIdx.x * DEC4_PX + q) * 256 + threadIdx.x;
        live[q] = idx < H1 * W1;
        const int ic = live[q] ? idx : 0;
        xs[q] = ic % W1;
        ys[q] = ic / W1;
#pragma unroll
        for (int k = 0; k < CLS; ++k)
            s[q][k] = *at(a.f, n, k, ys[q], xs[q]);
        zero_padding_lanes<CLS>(s[q], ncls);
    }
#pragma unroll
    for (int q = 0; q < DEC4_PX; ++q) {
        const int x = xs[q], y = ys[q];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            unsigned char m[2];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float best = 0.0f;
                int bi = 0;
                float lg[CLS];
#pragma unroll
                for (int o = 0; o < CLS; ++o) {
                    float t = 0.0f;
#pragma unroll
                    for (int i = 0; i < CLS; ++i)
                        t = fmaf(s[q][i], wl[((i * CLS + o) * 2 + dy) * 2 + dx], t);
                    lg[o] = t;
                    if (!ENS && o < ncls && (o == 0 || t > best)) {   // strict '>' : first maximum wins
                        best = t;
                        bi = o;
                    }
                }
                const long long pix = ((long long)(2 * y + dy)) * W + 2 * x + dx;
                if (ENS) {
                    // prob (+)= ens_w * softmax(logits): max-shifted expf, one division per class (dec_tail.h's arithmetic)
                    float mx = -3.4e38f;
#pragma unroll
                    for (int o = 0; o < CLS; ++o)
                        if (o < ncls)
                            mx = fmaxf(mx, lg[o]);
                    float sum = 0.0f;
#pragma unroll
                    for (int o = 0; o < CLS; ++o)
                        if (o < ncls) {
                            lg[o] = expf(lg[o] - mx);
                            sum += lg[o];
                        }
                    bool first = true;
#pragma unroll
                    for (int o = 0; o < CLS; ++o)
                        if (o < ncls && live[q]) {
                            float pr = lg[o] / sum * a.ens_w;
                            float *pp = a.prob + ((long long)n * ncls + o) * H * W + pix;
                            if (a.ens_mode == 2 || a.ens_mode == 3)
                                pr = *pp + pr;
                            if (a.ens_mode == 1 || a.ens_mode == 2)
                                *pp = pr;
                            else if (first || pr > best) {
                                best = pr;
                                bi = o;
                            }
                            first = false;
                        }
                } else if (a.logits && live[q]) {
#pragma unroll
                    for (int o = 0; o < CLS; ++o)
                        if (o < ncls)
                            a.logits[((long long)n * ncls + o) * H * W + pix] = lg[o];
                }
                m[dx] = (unsigned char)bi;
                cls_of[q][dy * 2 + dx] = live[q] ? bi : -1;   // -1: no pixel here
            }
            if (a.mask && live[q] && (!ENS || a.ens_mode >= 3))
                st_stream<NT_DEC4_ST>(reinterpret_cast<unsigned short *>(a.mask + ((long long)n * H + 2 * y + dy) * W + 2 * x),
                                      (unsigned short)(m[0] | (m[1] << 8)));
        }
    }
    if (a.hist && (!ENS || a.ens_mode >= 3)) {
        // per-class counts by wave ballot + popcount: one LDS atomic per class per wave (a per-pixel
        // LDS atomic on five hot words serialised the whole workgroup)
#pragma unroll
        for (int k = 0; k < CLS; ++k) {
            unsigned cnt = 0;   // (a padding class is never an argmax: its count is zero)
#pragma unroll
            for (int q = 0; q < DEC4_PX; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cnt += (unsigned)__popcll(__ballot(cls_of[q][j] == k));
            if ((threadIdx.x & 63) == 0 && cnt)
                atomicAdd(&lhist[k], cnt);
        }
        __syncthreads();
        if (threadIdx.x < ncls && lhist[threadIdx.x])
            atomicAdd(&a.hist[(long long)n * ncls + threadIdx.x], (unsigned long long)lhist[threadIdx.x]);
    }
}

// lazy b2 (debug / tests): planes c0 .. c0+cn-1 of a dense CHW copy of output1_cat are raw block outputs; apply b2 to them
__global__ void __launch_bounds__(256) b2_apply_kernel(float *chw, const float *b2, int hw, int c0, int cn)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)cn * hw)
        return;
    const int c = c0 + (int)(idx / hw);
    float *p = chw + (long long)c0 * hw + idx;
    *p = bn_prelu(*p, b2, 131, c);
}

// copy one image of a padded activation to a dense CHW buffer (debug / tests)
__global__ void __launch_bounds__(256) unpad_kernel(const ActV t, int n, int C, float *dst)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)C * t.H * t.W)
        return;
    const int x = (int)(idx % t.W);
    const int y = (int)((idx / t.W) % t.H);
    const int c = (int)(idx / ((long long)t.W * t.H));
    dst[idx] = *at(t, n, c, y, x);
}

// inverse of unpad_kernel: dense CHW -> interior of one image of a padded activation (test hook gs_espnet_block_forward)
__global__ void __launch_bounds__(256) pad_kernel(const ActV t, int n, int C, const float *src)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)C * t.H * t.W)
        return;
    const int x = (int)(idx % t.W);
    const int y = (int)((idx / t.W) % t.H);
    const int c = (int)(idx / ((long long)t.W * t.H));
    *at(t, n, c, y, x) = src[idx];
}

}  // namespace gs
