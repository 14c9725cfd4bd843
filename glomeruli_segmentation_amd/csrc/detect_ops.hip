// Detector-side device primitives: NHWC convolution on the fp32 matrix cores, crop_and_resize
// (the ROI pooling of TF object-detection Faster R-CNN graphs) and greedy IoU NMS.
//
// The reference's detector is an external TensorFlow-1.12 frozen graph fed through
// sess.run (module/faster-rcnn/detect_glomus_test.py:350-352); neither its architecture nor its
// weights are in the reference, so these kernels implement the published semantics of the TF ops
// such graphs contain and are tested for self-consistency only (parity unpinned).
#include <algorithm>
#include <vector>

#include "crop_sample.h"
#include "gs_internal.h"

namespace gs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// conv2d NHWC: GEMM-M = output pixels (32 per wave), GEMM-N = output channels (32 per wave, on the
// lanes, so NHWC stores are 128-byte rows), GEMM-K = (ky,kx,cin) two at a time.
// (ConvNhwcArgs: gs_internal.h)

__global__ void __launch_bounds__(256) conv2d_nhwc_kernel(const ConvNhwcArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int r = lane & 31, kq = lane >> 5;
    const long long npix = (long long)a.n * a.ho * a.wo;
    const long long pix0 = ((long long)blockIdx.x * 4 + wid) * 32;
    const int co0 = blockIdx.y * 32;
    if (pix0 >= npix)
        return;
    // A row r = pixel pix0 + r
    const long long pix = pix0 + r;
    const bool pv = pix < npix;
    const int ox = (int)(pix % a.wo), oy = (int)((pix / a.wo) % a.ho), img = (int)(pix / ((long long)a.wo * a.ho));
    const int co = co0 + r;   // B column r = output channel
    f32x16 acc = (f32x16)(0.0f);
    for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        for (int kx = 0; kx < a.kw; ++kx) {
            const int ix = ox * a.stride - a.pad + kx;
            const bool ok = pv && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
            const float *ip = a.in + (((long long)img * a.h + iy) * a.w_ + ix) * a.cin;
            const float *wp = a.w + ((long long)(ky * a.kw + kx) * a.cin) * a.cout + co;
            for (int c = 0; c < a.cin; c += 2) {
                const int ci = c + kq;
                const float av = (ok && ci < a.cin) ? ip[ci] : 0.0f;
                const float bv = (co < a.cout && ci < a.cin) ? wp[(long long)ci * a.cout] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    }
    // D: column = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * kq (pixel)
    if (co < a.cout) {
        const float b = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long long p = pix0 + (reg & 3) + 8 * (reg >> 2) + 4 * kq;
            if (p < npix) {
                float v = acc[reg] + b;
                if (a.relu)
                    v = fmaxf(v, 0.0f);
                a.out[p * a.cout + co] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv2d NHWC, tiled form for cin % 8 == 0 (every backbone layer but the first).
// A wave owns 64 output pixels x 64 output channels (2 x 2 MFMA tiles, 64 accumulator registers); a workgroup is
// four such waves along the pixel axis.  K walks (tap, 8-channel chunk): per chunk a lane fetches FOUR consecutive
// input channels of each of its two pixels with one 16-byte load (k-group kq takes channels c+4kq..c+4kq+3, so
// element s of the load is the lane's operand of k-step s) and the four weight rows it needs for both channel
// tiles as coalesced dword loads (the weight tensor is a few hundred KB and stays in L2).  16 MFMAs per chunk per
// wave run on the operands of the previous fetch while the next chunk's 10 loads are in flight.
typedef unsigned nhwc_u4 __attribute__((ext_vector_type(4)));

struct NhwcChunk {
    float a[2][4];   // [pixel tile][k-step]
    float b[2][4];   // [channel tile][k-step]
};

// WP4: the weights are pre-packed [k / 4][cout][4] (k = (tap, input channel) flattened; conv2d_nhwc_pack4), so that the
// four k-steps of a lane's channel come with ONE 16-byte load instead of four dword loads from four weight rows
// (10 -> 4 vector-memory instructions per 16 MFMAs).  The public gs_conv2d_nhwc takes TensorFlow's [kh,kw,cin,cout] as is;
// handles that own their weights (gs_detector) pack them once.
template <bool WP4>
__global__ void __launch_bounds__(256) conv2d_nhwc_tiled_kernel(const ConvNhwcArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int r = lane & 31, kq = lane >> 5;
    const long long npix = (long long)a.n * a.ho * a.wo;
    const long long pix0 = ((long long)blockIdx.x * 4 + wid) * 64;
    const int co0 = blockIdx.y * 64;
    if (pix0 >= npix)
        return;
    constexpr int OOB = 0x7ffffff0;
    // whole tensors behind buffer descriptors: a lane with nothing to read uses an out-of-range offset and gets 0
    const unsigned in_bytes = (unsigned)min((long long)a.n * a.h * a.w_ * a.cin * 4, (long long)0x7fffffff);
    const unsigned w_bytes = (unsigned)((long long)a.kh * a.kw * a.cin * a.cout * 4);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, w_bytes, 0x00020000);
    int oy[2], ox[2];
    long long ibase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long pix = pix0 + t * 32 + r;
        pv[t] = pix < npix;
        const long long pc = pv[t] ? pix : 0;
        ox[t] = (int)(pc % a.wo);
        oy[t] = (int)((pc / a.wo) % a.ho);
        ibase[t] = (pc / ((long long)a.wo * a.ho)) * a.h;
    }
    const bool cv[2] = {co0 + r < a.cout, co0 + 32 + r < a.cout};
    const int nchunk = a.cin / 8;
    const int total = a.kh * a.kw * nchunk;

    // Fetch iterator over (tap, 8-channel chunk), chunk fastest.  Everything that depends on the tap -- the two input
    // positions, their validity, the byte offsets of the lane's first activation and weight -- is computed once per tap;
    // inside a tap a fetch only adds the chunk's constant strides (no divisions, no 64-bit address arithmetic per chunk:
    // with those in every fetch the VALU work beside 16 MFMAs held the kernel at 46 % of the matrix peak).
    int f_cch = 0, f_ky = 0, f_kx = 0, f_tap = 0;
    int f_aoff[2], f_woff;
    auto set_tap = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iy = oy[t] * a.stride - a.pad + f_ky, ix = ox[t] * a.stride - a.pad + f_kx;
            const bool ok = pv[t] && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
            f_aoff[t] = ok ? (int)((((ibase[t] + iy) * a.w_ + ix) * a.cin + 4 * kq) * 4) : OOB;
        }
        f_woff = WP4 ? (((f_tap * a.cin) / 4 + kq) * a.cout + co0 + r) * 16 : ((f_tap * a.cin + 4 * kq) * a.cout + co0 + r) * 4;
    };
    set_tap();
    const int wstep = WP4 ? 2 * a.cout * 16 : 8 * a.cout * 4;   // bytes between the weights of consecutive chunks
    auto fetch = [&](NhwcChunk &q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // (an invalid position keeps its out-of-range offset: OOB + chunk stride is still beyond the descriptor)
#if defined(GS_DIAG) && defined(DET_X_SAMECHUNK)
            const nhwc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, f_aoff[t] == OOB ? OOB : f_aoff[t], 0, 0);   // timing only
#elif defined(GS_DIAG) && defined(DET_X_NOALOAD)
            const nhwc_u4 v = {(unsigned)f_aoff[t], (unsigned)f_cch, 1u, 2u};   // timing only
#else
            const nhwc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, f_aoff[t] == OOB ? OOB : f_aoff[t] + f_cch * 32, 0, 0);
#endif
            const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];   // (bit_cast straight from v[i] reads element 0)
            q.a[t][0] = __builtin_bit_cast(float, e0);
            q.a[t][1] = __builtin_bit_cast(float, e1);
            q.a[t][2] = __builtin_bit_cast(float, e2);
            q.a[t][3] = __builtin_bit_cast(float, e3);
        }
        const int wrow = f_woff + f_cch * wstep;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (WP4) {
                const nhwc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, cv[u] ? wrow + 32 * u * 16 : OOB, 0, 0);
                const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                q.b[u][0] = __builtin_bit_cast(float, e0);
                q.b[u][1] = __builtin_bit_cast(float, e1);
                q.b[u][2] = __builtin_bit_cast(float, e2);
                q.b[u][3] = __builtin_bit_cast(float, e3);
                continue;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                q.b[u][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          rw, cv[u] ? wrow + (s * a.cout + 32 * u) * 4 : OOB, 0, 0));
        }
        if (++f_cch == nchunk) {   // (uniform) next tap
            f_cch = 0;
            ++f_tap;
            if (++f_kx == a.kw) {
                f_kx = 0;
                ++f_ky;
            }
            set_tap();
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[t][u] = (f32x16)(0.0f);
    NhwcChunk cur, nxt;
    fetch(cur);
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total)
            fetch(nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t][s], cur.b[u][s], acc[t][u], 0, 0, 0);
        cur = nxt;
    }
    // D: column = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * kq (pixel)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int co = co0 + 32 * u + r;
        if (co >= a.cout)
            continue;
        const float b = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long p = pix0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kq;
                if (p < npix) {
                    float v = acc[t][u][reg] + b;
                    if (a.relu)
                        v = fmaxf(v, 0.0f);
                    a.out[p * a.cout + co] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// conv2d NHWC with packed weights for cin % 32 == 0 (the detector's 64 / 128 / 256-channel layers): the tiled kernel with
// its activation fetch widened to a whole 128-byte line per pixel.  In NHWC a lane's pixel is cin * 4 bytes from its
// neighbour's, so the 16-byte operand load of an 8-channel chunk uses an eighth of every line it pulls into L1, and by
// the time the same wave wants the next 16 bytes of that line -- a whole 16-MFMA chunk later, with every other wave of
// the CU doing the same -- the line has left the 32 KB L1: measured 86 TFLOP/s against 115 with the activation loads
// removed.  Here a lane fetches the 64 bytes it will need from the line (its k-group's four channels of FOUR consecutive
// chunks) with four back-to-back 16-byte loads, the partner k-group's lanes take the other 64, and the 64 MFMAs of the
// block run on registers.  Activations double-buffered per 32-channel block, weights per 8-channel chunk.
template <int NJ>
struct NhwcWideA {
    float a[2][4 * NJ];   // [pixel tile][chunk j * 4 + k-step]
};
struct NhwcWideW {
    float b[2][4];    // [channel tile][k-step]
};
// NJ = 8-channel chunks per block (4: a whole 128-byte line per pixel); WP4: weights packed [K/4][cout][4] (one 16-byte load
// per channel tile and chunk) or TensorFlow's [kh,kw,cin,cout] as given to the public gs_conv2d_nhwc (four dword loads)
template <int NJ, bool WP4>
__global__ void __launch_bounds__(256) conv2d_nhwc_wide_kernel(const ConvNhwcArgs a)
{
    static_assert(NJ % 2 == 0, "the weight registers ping-pong by chunk parity across blocks");
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int r = lane & 31, kq = lane >> 5;
    const long long npix = (long long)a.n * a.ho * a.wo;
    const long long pix0 = ((long long)blockIdx.x * 4 + wid) * 64;
    const int co0 = blockIdx.y * 64;
    if (pix0 >= npix)
        return;
    constexpr int OOB = 0x7ffffff0;
    const unsigned in_bytes = (unsigned)min((long long)a.n * a.h * a.w_ * a.cin * 4, (long long)0x7fffffff);
    const unsigned w_bytes = (unsigned)((long long)a.kh * a.kw * a.cin * a.cout * 4);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, w_bytes, 0x00020000);
    int oy[2], ox[2];
    long long ibase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long pix = pix0 + t * 32 + r;
        pv[t] = pix < npix;
        const long long pc = pv[t] ? pix : 0;
        ox[t] = (int)(pc % a.wo);
        oy[t] = (int)((pc / a.wo) % a.ho);
        ibase[t] = (pc / ((long long)a.wo * a.ho)) * a.h;
    }
    const bool cv[2] = {co0 + r < a.cout, co0 + 32 + r < a.cout};
    const int nblk = a.cin / (8 * NJ);
    const int total = a.kh * a.kw * nblk;   // blocks of the flattened (tap, channel) axis

    // activation iterator over (tap, block), block fastest; the tap-dependent part is computed once per tap
    int f_blk = 0, f_ky = 0, f_kx = 0;
    int f_aoff[2];
    auto set_tap = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iy = oy[t] * a.stride - a.pad + f_ky, ix = ox[t] * a.stride - a.pad + f_kx;
            const bool ok = pv[t] && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
            f_aoff[t] = ok ? (int)((((ibase[t] + iy) * a.w_ + ix) * a.cin + 4 * kq) * 4) : OOB;
        }
    };
    set_tap();
    auto fetch_a = [&](NhwcWideA<NJ> &q) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const nhwc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, f_aoff[t] == OOB ? OOB : f_aoff[t] + f_blk * (NJ * 32) + j * 32, 0, 0);
                const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];   // (bit_cast straight from v[i] reads element 0)
                q.a[t][4 * j + 0] = __builtin_bit_cast(float, e0);
                q.a[t][4 * j + 1] = __builtin_bit_cast(float, e1);
                q.a[t][4 * j + 2] = __builtin_bit_cast(float, e2);
                q.a[t][4 * j + 3] = __builtin_bit_cast(float, e3);
            }
        if (++f_blk == nblk) {   // (uniform) next tap
            f_blk = 0;
            if (++f_kx == a.kw) {
                f_kx = 0;
                ++f_ky;
            }
            set_tap();
        }
    };
    // weights: chunk g of the flattened (tap, channel) axis starts at k = 8g (+ 4 for the second k-group), so the offset is
    // linear in g in both layouts
    int w_off = WP4 ? (kq * a.cout + co0 + r) * 16 : (4 * kq * a.cout + co0 + r) * 4;
    const int wstep = WP4 ? 2 * a.cout * 16 : 8 * a.cout * 4;
    auto fetch_w = [&](NhwcWideW &q) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (WP4) {
                const nhwc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, cv[u] ? w_off + 32 * u * 16 : OOB, 0, 0);
                const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                q.b[u][0] = __builtin_bit_cast(float, e0);
                q.b[u][1] = __builtin_bit_cast(float, e1);
                q.b[u][2] = __builtin_bit_cast(float, e2);
                q.b[u][3] = __builtin_bit_cast(float, e3);
                continue;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                q.b[u][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, cv[u] ? w_off + (s * a.cout + 32 * u) * 4 : OOB, 0, 0));
        }
        w_off += wstep;   // (beyond the last chunk the offset leaves the descriptor: the load returns zeros)
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[t][u] = (f32x16)(0.0f);
    NhwcWideA<NJ> a0, a1;
    NhwcWideW w0, w1;
    fetch_a(a0);
    fetch_w(w0);
    // one block: prefetch the next block's activations, then its NJ chunks on ping-pong weight registers
    auto block = [&](const NhwcWideA<NJ> &use, NhwcWideA<NJ> &pre, bool more) __attribute__((always_inline)) {
        if (more)
            fetch_a(pre);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            NhwcWideW &wu = (j & 1) ? w1 : w0;
            NhwcWideW &wp = (j & 1) ? w0 : w1;
            fetch_w(wp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(use.a[t][4 * j + s], wu.b[u][s], acc[t][u], 0, 0, 0);
        }
    };
    for (int it = 0; it < total; it += 2) {
        block(a0, a1, it + 1 < total);
        if (it + 1 < total)
            block(a1, a0, it + 2 < total);
    }
    // D: column = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * kq (pixel)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int co = co0 + 32 * u + r;
        if (co >= a.cout)
            continue;
        const float b = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long p = pix0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kq;
                if (p < npix) {
                    float v = acc[t][u][reg] + b;
                    if (a.relu)
                        v = fmaxf(v, 0.0f);
                    a.out[p * a.cout + co] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// conv2d NHWC for few input channels (the 3-channel first layer of every detector backbone): K = kh*kw*cin is walked
// FLATTENED, two k per MFMA step, so a 3x3x3 layer is 14 k-steps (27 -> 28) instead of nine taps of two half-empty
// channel steps.  Same 64-pixel x 64-channel wave tile and register double-buffering as the tiled kernel; a k-step's
// (ky, kx, c) comes from a small table in LDS, its two activation operands are scalar gathers (the whole input is a
// few MB and lives in L2).  On 16 windows of 1000x1000x3 -> 64 channels, stride 2: 22 -> see profiles/ TFLOP/s.
__global__ void __launch_bounds__(256) conv2d_nhwc_smallcin_kernel(const ConvNhwcArgs a)
{
    __shared__ int lut[512];   // k -> ky << 20 | kx << 10 | c   (K <= 512)
    const int K = a.kh * a.kw * a.cin;
    for (int k = threadIdx.x; k < 512; k += 256) {
        const int tap = k / a.cin, c = k - tap * a.cin;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        lut[k] = k < K ? (ky << 20 | kx << 10 | c) : -1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int r = lane & 31, kq = lane >> 5;
    const long long npix = (long long)a.n * a.ho * a.wo;
    const long long pix0 = ((long long)blockIdx.x * 4 + wid) * 64;
    const int co0 = blockIdx.y * 64;
    if (pix0 >= npix)
        return;
    constexpr int OOB = 0x7ffffff0;
    const unsigned in_bytes = (unsigned)min((long long)a.n * a.h * a.w_ * a.cin * 4, (long long)0x7fffffff);
    const unsigned w_bytes = (unsigned)((long long)K * a.cout * 4);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, w_bytes, 0x00020000);
    int iy0[2], ix0[2];
    long long ibase[2];
    bool pv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long pix = pix0 + t * 32 + r;
        pv[t] = pix < npix;
        const long long pc = pv[t] ? pix : 0;
        ix0[t] = (int)(pc % a.wo) * a.stride - a.pad;
        iy0[t] = (int)((pc / a.wo) % a.ho) * a.stride - a.pad;
        ibase[t] = (pc / ((long long)a.wo * a.ho)) * a.h;
    }
    const bool cv[2] = {co0 + r < a.cout, co0 + 32 + r < a.cout};
    const int nstep = (K + 1) / 2;
    struct Ops {
        float a[2], b[2];
    };
    auto fetch = [&](int s, Ops &q) {
        const int k = 2 * s + kq;
        const int e = lut[k];
        const int ky = e >> 20, kx = (e >> 10) & 1023, c = e & 1023;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iy = iy0[t] + ky, ix = ix0[t] + kx;
            const bool ok = e >= 0 && pv[t] && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
            const long long off = (((ibase[t] + iy) * a.w_ + ix) * a.cin + c) * 4;
            q.a[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, ok ? (int)off : OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            q.b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rw, (e >= 0 && cv[u]) ? (k * a.cout + co0 + 32 * u + r) * 4 : OOB, 0, 0));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            acc[t][u] = (f32x16)(0.0f);
    Ops cur, nxt;
    fetch(0, cur);
    for (int s = 0; s < nstep; ++s) {
        if (s + 1 < nstep)
            fetch(s + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t], cur.b[u], acc[t][u], 0, 0, 0);
        cur = nxt;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int co = co0 + 32 * u + r;
        if (co >= a.cout)
            continue;
        const float b = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long p = pix0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kq;
                if (p < npix) {
                    float v = acc[t][u][reg] + b;
                    if (a.relu)
                        v = fmaxf(v, 0.0f);
                    a.out[p * a.cout + co] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// tf.image.crop_and_resize, method="bilinear", extrapolation_value=0.  One thread per
// (box, y, x, channel); channels are innermost so loads/stores coalesce in NHWC.
__global__ void __launch_bounds__(256)
roialign_kernel(const float *feat, int n, int h, int w, int c, const float *boxes, const int *box_image, int n_boxes,
                int crop, float *out)
{
    // no fused multiply-add here: a sample that lands exactly on the border of the feature map is inside or extrapolated
    // by the last bit of in_y / in_x, and TensorFlow rounds every operation (see det_crop_pool_kernel)
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)n_boxes * crop * crop * c;
    if (idx >= total)
        return;
    const int ch = (int)(idx % c);
    const int x = (int)((idx / c) % crop);
    const int y = (int)((idx / ((long long)c * crop)) % crop);
    const int b = (int)(idx / ((long long)c * crop * crop));
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1], y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    const int img = box_image[b];
    float v = 0.0f;
    if (img >= 0 && img < n) {
        const float hs = crop > 1 ? (y2 - y1) * (float)(h - 1) / (float)(crop - 1) : 0.0f;
        const float ws = crop > 1 ? (x2 - x1) * (float)(w - 1) / (float)(crop - 1) : 0.0f;
        const float in_y = crop > 1 ? y1 * (float)(h - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(h - 1);
        const float in_x = crop > 1 ? x1 * (float)(w - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(w - 1);
        if (in_y >= 0.0f && in_y <= (float)(h - 1) && in_x >= 0.0f && in_x <= (float)(w - 1)) {
            const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
            const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
            const float fy = in_y - (float)ty, fx = in_x - (float)lx;
            const float *base = feat + (long long)img * h * w * c + ch;
            const float tl = base[((long long)ty * w + lx) * c], tr = base[((long long)ty * w + rx) * c];
            const float bl = base[((long long)by * w + lx) * c], br = base[((long long)by * w + rx) * c];
            const float top = tl + (tr - tl) * fx, bot = bl + (br - bl) * fx;
            v = top + (bot - top) * fy;
        }
    }
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// NMS.  (1) rank boxes by descending score (ties: lower index first) with an O(k^2) count, which
// also yields the sorted order without a sort network; (2) 64-bit suppression masks with one wave
// ballot per 64x64 block; (3) one wave walks the sorted list.
__device__ __forceinline__ float box_iou(const float *a, const float *b)
{
    const float ay1 = fminf(a[0], a[2]), ax1 = fminf(a[1], a[3]), ay2 = fmaxf(a[0], a[2]), ax2 = fmaxf(a[1], a[3]);
    const float by1 = fminf(b[0], b[2]), bx1 = fminf(b[1], b[3]), by2 = fmaxf(b[0], b[2]), bx2 = fmaxf(b[1], b[3]);
    const float aa = (ay2 - ay1) * (ax2 - ax1), ab = (by2 - by1) * (bx2 - bx1);
    if (aa <= 0.0f || ab <= 0.0f)
        return 0.0f;
    const float ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f);
    const float inter = ih * iw;
    return inter / (aa + ab - inter);
}

__global__ void __launch_bounds__(256) nms_rank_kernel(const float *scores, int k, float score_thr, int *order, int *n_valid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k)
        return;
    const float s = scores[i];
    if (!(s > score_thr))
        return;
    int rank = 0;
    for (int j = 0; j < k; ++j) {
        const float t = scores[j];
        rank += (t > score_thr) && (t > s || (t == s && j < i));
    }
    order[rank] = i;
    atomicAdd(n_valid, 1);
}

__global__ void __launch_bounds__(64)
nms_mask_kernel(const float *boxes, const int *order, const int *n_valid, float thr, int words, unsigned long long *mask)
{
    const int nv = *n_valid;
    const int row = blockIdx.y * 64 + threadIdx.x;   // sorted position of the candidate suppressor
    const int colb = blockIdx.x;                     // 64-wide block of sorted positions it may suppress
    if (blockIdx.y * 64 >= nv || colb * 64 >= nv)
        return;
    __shared__ float cb[64][4];
    const int cpos = colb * 64 + threadIdx.x;
    if (cpos < nv) {
        const float *p = boxes + (long long)order[cpos] * 4;
        cb[threadIdx.x][0] = p[0];
        cb[threadIdx.x][1] = p[1];
        cb[threadIdx.x][2] = p[2];
        cb[threadIdx.x][3] = p[3];
    }
    __syncthreads();
    if (row >= nv)
        return;
    float rb[4];
    const float *p = boxes + (long long)order[row] * 4;
    rb[0] = p[0];
    rb[1] = p[1];
    rb[2] = p[2];
    rb[3] = p[3];
    unsigned long long bits = 0;
    for (int j = 0; j < 64; ++j) {
        const int c = colb * 64 + j;
        if (c < nv && c > row && box_iou(rb, cb[j]) > thr)
            bits |= 1ull << j;
    }
    mask[(long long)row * words + colb] = bits;
}

__global__ void __launch_bounds__(64)
nms_scan_kernel(const unsigned long long *mask, const int *order, const int *n_valid, int words, int max_out, int *keep,
                int *n_keep)
{
    extern __shared__ unsigned long long removed[];   // [words]
    const int nv = *n_valid;
    for (int w = threadIdx.x; w < words; w += 64)
        removed[w] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < nv && kept < max_out; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;   // wave-uniform
        if (!dead) {
            if (threadIdx.x == 0)
                keep[kept] = order[i];
            ++kept;
            for (int w = (i >> 6) + threadIdx.x; w < words; w += 64)
                removed[w] |= mask[(long long)i * words + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *n_keep = kept;
}

// ---------------------------------------------------------------------------------------------
// Crop stage, one crop per launch; the sampling rules are in crop_sample.h (shared with crops.hip)
struct CropArgs {
    const unsigned char *src;
    int h, w, oh, ow;
    float mean[3], std[3];
    float *out;
};

__global__ void __launch_bounds__(256) crop_preprocess_kernel(const CropArgs a)
{
    __shared__ float lut[3][256];
    crop_norm_table(lut, a.mean, a.std);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.oh * a.ow)
        return;
    const int ox = idx % a.ow, oy = idx / a.ow;
    int x0, x1, y0, y1;
    float wx, wy;
    linear_tap(ox, cv_inv_scale(a.ow, a.w), a.w, x0, x1, wx);
    linear_tap(oy, cv_inv_scale(a.oh, a.h), a.h, y0, y1, wy);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        a.out[((long long)c * a.oh + oy) * a.ow + ox] = crop_sample(a.src, a.w, c, x0, x1, y0, y1, wx, wy, lut[c]);
}

__global__ void __launch_bounds__(256)
mask_nearest_kernel(const unsigned char *src, int h, int w, int oh, int ow, unsigned char *out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= oh * ow)
        return;
    const int ox = idx % ow, oy = idx / ow;
    const int sx = nearest_src(ox, cv_inv_scale(ow, w), w), sy = nearest_src(oy, cv_inv_scale(oh, h), h);
    out[idx] = src[(long long)sy * w + sx];
}

// ---------------------------------------------------------------------------------------------
// WSI compositor
__global__ void __launch_bounds__(256)
wsi_paste_max_kernel(unsigned char *map, int map_h, int map_w, int ds, const unsigned char *crop, int h, int w, int x1,
                     int y1, int X0, int Y0, int nx, int ny)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * ny)
        return;
    const int X = X0 + idx % nx, Y = Y0 + idx / nx;
    const int cx = X * ds - x1, cy = Y * ds - y1;
    if (X < 0 || Y < 0 || X >= map_w || Y >= map_h || cx < 0 || cy < 0 || cx >= w || cy >= h)
        return;
    const unsigned char v = crop[(long long)cy * w + cx];
    unsigned char *dst = map + (long long)Y * map_w + X;
    if (v > *dst)
        *dst = v;
}

// The same paste with the level-0 sample position of every map column / row given by a table (-1 = the reference
// never writes that column / row): reproduces the reference's 2400-px window walk exactly, including its partial edge
// windows, whose INTER_NEAREST step is not 8 (eval_wsi_segmentation.py:229), and the windows it skips (:386).
__global__ void __launch_bounds__(256)
wsi_paste_max_lut_kernel(unsigned char *map, int map_h, int map_w, const int *sx, const int *sy, const unsigned char *crop, int h,
                         int w, int x1, int y1, int X0, int Y0, int nx, int ny)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nx * ny)
        return;
    const int X = X0 + idx % nx, Y = Y0 + idx / nx;
    if (X < 0 || Y < 0 || X >= map_w || Y >= map_h)
        return;
    const int px = sx[X], py = sy[Y];
    if (px < 0 || py < 0)
        return;
    const int cx = px - x1, cy = py - y1;
    if (cx < 0 || cy < 0 || cx >= w || cy >= h)
        return;
    const unsigned char v = crop[(long long)cy * w + cx];
    unsigned char *dst = map + (long long)Y * map_w + X;
    if (v > *dst)
        *dst = v;
}

__global__ void __launch_bounds__(256)
overlay_kernel(const unsigned char *region, const unsigned char *cls, long long npix, const unsigned char *pal, int ncol, float wa,
               float wb, unsigned char *out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix)
        return;
    const int c = cls[idx];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // palette rows are RGB, the image is BGR (classMap_numpy_color[...] = [b, g, r])
        const float col = c < ncol ? (float)pal[c * 3 + (2 - ch)] : 0.0f;
        const float v = __fadd_rn(__fmul_rn((float)region[idx * 3 + ch], wa), __fmul_rn(col, wb));   // (no fused multiply-add: numpy's roundings)
        out[idx * 3 + ch] = (unsigned char)fminf(fmaxf(rintf(v), 0.0f), 255.0f);   // saturate_cast<uchar>(cvRound)
    }
}

__global__ void __launch_bounds__(256)
confusion_kernel(const unsigned char *pred, const unsigned char *gt, long long n, int classes, unsigned long long *hist)
{
    extern __shared__ unsigned int lh[];
    const int cells = classes * classes;
    for (int i = threadIdx.x; i < cells; i += 256)
        lh[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int g = gt[i], p = pred[i];
        if (g < classes && p < classes)
            atomicAdd(&lh[classes * g + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += 256)
        if (lh[i])
            atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

}  // namespace gs

using namespace gs;

namespace gs {

bool conv2d_nhwc_can_pack4(int n, int h, int w, int cin, int kh, int kw, int cout)
{
    return cin % 8 == 0 && (long long)n * h * w * cin * 4 < 0x7fffffffLL && (long long)kh * kw * cin * cout * 4 < 0x7fffffffLL;
}

// [kh,kw,cin,cout] -> [(kh*kw*cin) / 4][cout][4]  (host)
void conv2d_nhwc_pack4(const float *w, int kh, int kw, int cin, int cout, float *dst)
{
    const int K = kh * kw * cin;
    for (int k = 0; k < K; ++k)
        for (int co = 0; co < cout; ++co)
            dst[((size_t)(k / 4) * cout + co) * 4 + (k & 3)] = w[(size_t)k * cout + co];
}

gs_status conv2d_nhwc_packed4(ConvNhwcArgs a, hipStream_t stream)
{
    a.ho = (a.h + 2 * a.pad - a.kh) / a.stride + 1;
    a.wo = (a.w_ + 2 * a.pad - a.kw) / a.stride + 1;
    const long long npix = (long long)a.n * a.ho * a.wo;
    if (a.ho <= 0 || a.wo <= 0 || !conv2d_nhwc_can_pack4(a.n, a.h, a.w_, a.cin, a.kh, a.kw, a.cout)) {
        set_error("conv2d_nhwc_packed4: shape not supported by the packed-weight kernel");
        return GS_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)((npix + 255) / 256), (unsigned)((a.cout + 63) / 64));
    // whole-line activation fetches where a pixel has at least a line of channels and the map is not a handful of pixels.
    // Measured on the detector (16 windows of 1000 x 1000): 64..256-channel backbone layers 86-90 -> 98-119 TFLOP/s; the
    // 16-channel first layer at two chunks per block 876 -> 1026 us and the box head's 7x7 -> 4x4 layer 264 -> 312 us,
    // so those stay on the chunk-at-a-time kernel.
    if (a.cin % 32 == 0 && a.ho * a.wo >= 64)
        hipLaunchKernelGGL((conv2d_nhwc_wide_kernel<4, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(conv2d_nhwc_tiled_kernel<true>, grid, dim3(256), 0, stream, a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs

extern "C" {

gs_status gs_crop_preprocess(const uint8_t *crop_bgr, int h, int w, const float mean[3], const float std[3], int out_h,
                             int out_w, float *out_chw, void *hip_stream)
{
    GS_REQUIRE(crop_bgr && mean && std && out_chw, "gs_crop_preprocess: null pointer");
    GS_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0, "gs_crop_preprocess: bad size");
    CropArgs a{crop_bgr, h, w, out_h, out_w, {mean[0], mean[1], mean[2]}, {std[0], std[1], std[2]}, out_chw};
    for (int i = 0; i < 3; ++i)
        GS_REQUIRE(std[i] != 0.0f, "std[%d] is zero", i);
    hipLaunchKernelGGL(crop_preprocess_kernel, dim3((out_h * out_w + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_mask_resize_nearest(const uint8_t *mask, int h, int w, int out_h, int out_w, uint8_t *out, void *hip_stream)
{
    GS_REQUIRE(mask && out, "gs_mask_resize_nearest: null pointer");
    GS_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0, "gs_mask_resize_nearest: bad size");
    hipLaunchKernelGGL(mask_nearest_kernel, dim3((out_h * out_w + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), mask, h, w, out_h, out_w, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_wsi_paste_max(uint8_t *slide_map, int map_h, int map_w, int ds, const uint8_t *crop_mask, int h, int w, int x1,
                           int y1, void *hip_stream)
{
    GS_REQUIRE(slide_map && crop_mask, "gs_wsi_paste_max: null pointer");
    GS_REQUIRE(map_h > 0 && map_w > 0 && ds > 0 && h > 0 && w > 0, "gs_wsi_paste_max: bad size");
    // footprint of the crop on the map grid: X with x1 <= ds*X < x1 + w
    auto ceil_div = [](long long a, long long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); };
    const int X0 = (int)ceil_div(x1, ds), Y0 = (int)ceil_div(y1, ds);
    const int X1 = (int)ceil_div((long long)x1 + w, ds), Y1 = (int)ceil_div((long long)y1 + h, ds);
    const int nx = X1 - X0, ny = Y1 - Y0;
    if (nx <= 0 || ny <= 0)
        return GS_OK;
    hipLaunchKernelGGL(wsi_paste_max_kernel, dim3((nx * ny + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       slide_map, map_h, map_w, ds, crop_mask, h, w, x1, y1, X0, Y0, nx, ny);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_wsi_paste_max_lut(uint8_t *slide_map, int map_h, int map_w, int ds, const int *sx_lut, const int *sy_lut,
                               const uint8_t *crop_mask, int h, int w, int x1, int y1, void *hip_stream)
{
    GS_REQUIRE(slide_map && crop_mask && sx_lut && sy_lut, "gs_wsi_paste_max_lut: null pointer");
    GS_REQUIRE(map_h > 0 && map_w > 0 && ds > 0 && h > 0 && w > 0, "gs_wsi_paste_max_lut: bad size");
    // candidate map cells: the table is monotone with steps of >= ds, so the crop's cells lie within one cell of its
    // footprint on the regular grid; the kernel tests each candidate against the table
    const int X0 = x1 / ds - 1, Y0 = y1 / ds - 1;
    const int nx = (x1 + w) / ds + 2 - X0, ny = (y1 + h) / ds + 2 - Y0;
    hipLaunchKernelGGL(wsi_paste_max_lut_kernel, dim3((nx * ny + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       slide_map, map_h, map_w, sx_lut, sy_lut, crop_mask, h, w, x1, y1, X0, Y0, nx, ny);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_overlay_classmap(const uint8_t *region_bgr, const uint8_t *class_map, int h, int w, const uint8_t *palette_rgb,
                              int n_colours, float wa, float wb, uint8_t *out_bgr, void *hip_stream)
{
    GS_REQUIRE(region_bgr && class_map && palette_rgb && out_bgr, "gs_overlay_classmap: null pointer");
    GS_REQUIRE(h > 0 && w > 0 && n_colours > 0, "gs_overlay_classmap: bad size");
    const long long npix = (long long)h * w;
    hipLaunchKernelGGL(overlay_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       region_bgr, class_map, npix, palette_rgb, n_colours, wa, wb, out_bgr);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_confusion_u8(const uint8_t *pred, const uint8_t *gt, long long n, int classes, unsigned long long *hist,
                          void *hip_stream)
{
    GS_REQUIRE(pred && gt && hist, "gs_confusion_u8: null pointer");
    GS_REQUIRE(n >= 0 && classes > 0 && classes <= 64, "gs_confusion_u8: bad size");
    if (n == 0)
        return GS_OK;
    long long blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), (size_t)classes * classes * sizeof(unsigned),
                       static_cast<hipStream_t>(hip_stream), pred, gt, n, classes, hist);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_conv2d_nhwc(const float *in, int n, int h, int w, int cin, const float *weight, int kh, int kw, int cout,
                         const float *bias_or_null, int stride, int pad, int relu, float *out, void *hip_stream)
{
    GS_REQUIRE(in && weight && out, "gs_conv2d_nhwc: null pointer");
    GS_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
               "gs_conv2d_nhwc: bad dimensions");
    ConvNhwcArgs a{in, weight, bias_or_null, out, n, h, w, cin, kh, kw, cout, stride, pad, relu, 0, 0};
    a.ho = (h + 2 * pad - kh) / stride + 1;
    a.wo = (w + 2 * pad - kw) / stride + 1;
    GS_REQUIRE(a.ho > 0 && a.wo > 0, "gs_conv2d_nhwc: empty output");
    const long long npix = (long long)n * a.ho * a.wo;
    // tiled kernel: 8-channel chunks, 32-bit byte offsets into the input and the weights
    if (cin % 8 == 0 && (long long)n * h * w * cin * 4 < 0x7fffffffLL && (long long)kh * kw * cin * cout * 4 < 0x7fffffffLL) {
        dim3 grid((unsigned)((npix + 255) / 256), (unsigned)((cout + 63) / 64));
        if (cin % 32 == 0 && a.ho * a.wo >= 64)   // whole-line activation fetches (see conv2d_nhwc_packed4)
            hipLaunchKernelGGL((conv2d_nhwc_wide_kernel<4, false>), grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
        else
            hipLaunchKernelGGL(conv2d_nhwc_tiled_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
        GS_HIP(hipGetLastError());
        return GS_OK;
    }
    // few input channels: flattened-K kernel (K = kh*kw*cin up to 512, channel / tap indices below 1024)
    if (cin < 8 && (long long)kh * kw * cin <= 512 && kh < 1024 && kw < 1024 && (long long)n * h * w * cin * 4 < 0x7fffffffLL) {
        dim3 grid((unsigned)((npix + 255) / 256), (unsigned)((cout + 63) / 64));
        hipLaunchKernelGGL(conv2d_nhwc_smallcin_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
        GS_HIP(hipGetLastError());
        return GS_OK;
    }
    dim3 grid((unsigned)((npix + 127) / 128), (unsigned)((cout + 31) / 32));
    hipLaunchKernelGGL(conv2d_nhwc_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_roialign(const float *feat, int n, int h, int w, int c, const float *boxes, const int *box_image, int n_boxes,
                      int crop, float *out, void *hip_stream)
{
    GS_REQUIRE(feat && boxes && box_image && out, "gs_roialign: null pointer");
    GS_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && crop > 0 && n_boxes >= 0, "gs_roialign: bad dimensions");
    if (n_boxes == 0)
        return GS_OK;
    const long long total = (long long)n_boxes * crop * crop * c;
    hipLaunchKernelGGL(roialign_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), feat, n, h, w, c, boxes, box_image, n_boxes, crop, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_nms(const float *boxes, const float *scores, int k, float iou_threshold, float score_threshold, int max_out,
                 int *keep, int *n_keep, void *hip_stream)
{
    GS_REQUIRE(boxes && scores && keep && n_keep, "gs_nms: null pointer");
    GS_REQUIRE(k >= 0 && max_out >= 0, "gs_nms: negative count");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (k == 0 || max_out == 0) {
        GS_HIP(hipMemsetAsync(n_keep, 0, sizeof(int), s));
        return GS_OK;
    }
    const int words = (k + 63) / 64;
    GS_REQUIRE((size_t)words * 8 <= 64 * 1024, "gs_nms: more than %d boxes are not supported", 8192 * 64);
    // scratch: order[k] | n_valid | mask[k*words]
    const size_t scratch = round_up((size_t)(k + 4) * sizeof(int), 16) + (size_t)k * words * 8;
    // stream-ordered scratch: no host synchronisation, the call stays asynchronous like the other primitives
    char *d = nullptr;
    GS_HIP(hipMallocAsync(reinterpret_cast<void **>(&d), scratch, s));
    int *order = reinterpret_cast<int *>(d);
    int *n_valid = order + k;
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(d + round_up((size_t)(k + 4) * sizeof(int), 16));
    hipError_t e = hipMemsetAsync(d, 0, scratch, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(nms_rank_kernel, dim3((k + 255) / 256), dim3(256), 0, s, scores, k, score_threshold, order, n_valid);
        hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, s, boxes, order, n_valid, iou_threshold, words, mask);
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), (size_t)words * 8, s, mask, order, n_valid, words, max_out, keep, n_keep);
        e = hipGetLastError();
    }
    hipError_t e2 = hipFreeAsync(d, s);
    GS_HIP(e);
    GS_HIP(e2);
    return GS_OK;
}

}  // extern "C"
