// gs_detector_*: an assembled two-stage (Faster R-CNN) glomerulus detector forward on the GPU, behind the tensor
// contract of the reference's `detect_box` (module/faster-rcnn/detect_glomus_test.py:349-368, tensors :443-450):
//   uint8 RGB [N,H,W,3]  ->  detection_boxes [N,D,4] (normalised ymin,xmin,ymax,xmax), detection_scores [N,D]
//   (descending), detection_classes [N,D], num_detections [N].
//
// The reference's network is an external TensorFlow-1.12 frozen graph that is NOT part of the reference (:419-427;
// download in example/README.md:22), so neither its architecture nor its weights can be reproduced: parity for the
// detector is unpinned.  What this file provides is the SHAPE of such a graph -- backbone conv stack ->
// RPN (objectness + box deltas over grid anchors) -> box decode + clip -> top-k -> NMS -> crop_and_resize ->
// max-pool -> box head -> softmax + box decode -> per-class NMS -> padded, score-sorted outputs -- assembled from
// this library's own kernels (gs_conv2d_nhwc, gs_roialign and the glue kernels below) with caller-supplied
// weights, so that BASELINE config 3 (1000x1000 windows, batch 16) and the detection leg of the slide pipeline run
// on the GPU.  oracle/detector_oracle.py restates the same graph over torch CPU ops (not-reference-parity).
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gs_internal.h"
#include "host_copy.h"

namespace gs {

constexpr int DET_A = 12;            // anchors per cell: scales {0.25,0.5,1,2} x aspect ratios {0.5,1,2}
constexpr int DET_STRIDE = 16;       // feature stride
constexpr float DET_BASE = 256.0f;   // anchor base size (TF object-detection grid_anchor_generator)
constexpr int DET_PRE_NMS = 1024;    // RPN candidates kept by score before NMS
constexpr int DET_PROPOSALS = 300;   // first_stage_max_proposals
constexpr int DET_CROP = 14;         // initial_crop_size; followed by a 2x2 max-pool
constexpr int DET_MAX_DET = 100;     // max_total_detections
constexpr int DET_CF = 256;          // feature channels
constexpr int DET_CH = 128;          // box-head channels

// ---------------------------------------------------------------------------------------------
// uint8 RGB -> [-1,1] floats (2/255 x - 1, the TF-OD Faster R-CNN preprocessor), 2x2 space-to-depth, channels padded to
// 16: [N,H,W,3] u8 -> [N,ceil(H/2),ceil(W/2),16] fp32, channel (dy*2+dx)*3 + c; pixels beyond an odd edge read as 0.
__global__ void __launch_bounds__(256) det_preprocess_kernel(const unsigned char *in, int n, int h, int w, float *out)
{
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per (pixel of the half-size map, quad position)
    if (idx >= (long long)n * h2 * w2 * 4)
        return;
    const int q = (int)(idx & 3);
    const long long p = idx >> 2;
    const int x = (int)(p % w2), y = (int)((p / w2) % h2), img = (int)(p / ((long long)w2 * h2));
    const int sy = 2 * y + (q >> 1), sx = 2 * x + (q & 1);
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (sy < h && sx < w) {
        const unsigned char *s = in + (((long long)img * h + sy) * w + sx) * 3;
        r = (float)s[0] * (2.0f / 255.0f) - 1.0f;
        g = (float)s[1] * (2.0f / 255.0f) - 1.0f;
        b = (float)s[2] * (2.0f / 255.0f) - 1.0f;
    }
    float *o = out + p * 16 + q * 3;
    o[0] = r;
    o[1] = g;
    o[2] = b;
    if (q == 3)   // the thread of the last quad position also writes the four zero padding channels 12..15
        *reinterpret_cast<float4 *>(out + p * 16 + 12) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// max-pool, NHWC, window k, stride s, padding pad (padding never wins: -inf); 4 channels per thread
__global__ void __launch_bounds__(256) det_maxpool_kernel(const float *in, long long n, int h, int w, int c, int k, int s, int pad,
                                                          int ho, int wo, float *out)
{
    const int c4 = c / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * ho * wo * c4)
        return;
    const int cq = (int)(idx % c4);
    const long long p = idx / c4;
    const int x = (int)(p % wo), y = (int)((p / wo) % ho);
    const long long img = p / ((long long)wo * ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < k; ++ky) {
        const int iy = y * s - pad + ky;
        if (iy < 0 || iy >= h)
            continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = x * s - pad + kx;
            if (ix < 0 || ix >= w)
                continue;
            const float4 v = *reinterpret_cast<const float4 *>(in + ((img * h + iy) * w + ix) * c + cq * 4);
            m.x = fmaxf(m.x, v.x);
            m.y = fmaxf(m.y, v.y);
            m.z = fmaxf(m.z, v.z);
            m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4 *>(out + p * c + cq * 4) = m;
}

// tf.image.crop_and_resize (bilinear, extrapolation 0; the arithmetic of gs_roialign) of a crop x crop grid followed by the
// 2x2 / stride-2 max-pool, in one pass: the 14x14 crops (0.96 GB for 4800 boxes of 256 channels) are neither written nor
// read back.  One thread per (box, pooled y, pooled x, four channels).
__global__ void __launch_bounds__(256)
det_crop_pool_kernel(const float *feat, int n, int h, int w, int c, const float *boxes, const int *box_image, int n_boxes, int crop,
                     float *out)
{
    // every product and sum rounded on its own in this kernel (no fused multiply-add): whether the last sample of a box
    // that ends exactly on the feature map's border is inside (<= h-1) or extrapolated (0) hangs on the last bit of in_y /
    // in_x, and tf.image.crop_and_resize (and the oracle) round each operation
#pragma clang fp contract(off)
    const int half = crop / 2, c4 = c / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_boxes * half * half * c4)
        return;
    const int cq = (int)(idx % c4);
    const int px = (int)((idx / c4) % half), py = (int)((idx / ((long long)c4 * half)) % half);
    const int b = (int)(idx / ((long long)c4 * half * half));
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1], y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    const int img = box_image[b];
    const float hs = (y2 - y1) * (float)(h - 1) / (float)(crop - 1);
    const float ws = (x2 - x1) * (float)(w - 1) / (float)(crop - 1);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int y = 2 * py + dy, x = 2 * px + dx;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float in_y = y1 * (float)(h - 1) + (float)y * hs;
            const float in_x = x1 * (float)(w - 1) + (float)x * ws;
            if (img >= 0 && img < n && in_y >= 0.0f && in_y <= (float)(h - 1) && in_x >= 0.0f && in_x <= (float)(w - 1)) {
                const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
                const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
                const float fy = in_y - (float)ty, fx = in_x - (float)lx;
                const float *base = feat + (long long)img * h * w * c + cq * 4;
                const float4 tl = *reinterpret_cast<const float4 *>(base + ((long long)ty * w + lx) * c);
                const float4 tr = *reinterpret_cast<const float4 *>(base + ((long long)ty * w + rx) * c);
                const float4 bl = *reinterpret_cast<const float4 *>(base + ((long long)by * w + lx) * c);
                const float4 br = *reinterpret_cast<const float4 *>(base + ((long long)by * w + rx) * c);
                auto lerp2 = [&](float a, float bb, float cc, float d) {
                    const float top = a + (bb - a) * fx, bot = cc + (d - cc) * fx;
                    return top + (bot - top) * fy;
                };
                v = make_float4(lerp2(tl.x, tr.x, bl.x, br.x), lerp2(tl.y, tr.y, bl.y, br.y), lerp2(tl.z, tr.z, bl.z, br.z),
                                lerp2(tl.w, tr.w, bl.w, br.w));
            }
            m.x = fmaxf(m.x, v.x);
            m.y = fmaxf(m.y, v.y);
            m.z = fmaxf(m.z, v.z);
            m.w = fmaxf(m.w, v.w);
        }
    *reinterpret_cast<float4 *>(out + (((long long)b * half + py) * half + px) * c + cq * 4) = m;
}

// spatial mean of [n, hw, c] -> [n, c] (sum in index order, then / hw)
__global__ void __launch_bounds__(256) det_avgpool_kernel(const float *in, long long n, int hw, int c, float *out)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * c)
        return;
    const int ch = (int)(idx % c);
    const long long b = idx / c;
    float s = 0.0f;
    for (int i = 0; i < hw; ++i)
        s += in[(b * hw + i) * c + ch];
    out[idx] = s / (float)hw;
}

// RPN objectness of every anchor: softmax over (background, foreground) logits = sigmoid(fg - bg).
// rpn: [N, cells, 6A] with [0,2A) = class logits (anchor-major, bg then fg) and [2A,6A) = box deltas (ty,tx,th,tw)
__global__ void __launch_bounds__(256) det_objectness_kernel(const float *rpn, long long total /* N*cells*A */, float *score)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total)
        return;
    const int a = (int)(idx % DET_A);
    const long long cell = idx / DET_A;
    const float *p = rpn + cell * (6 * DET_A) + 2 * a;
    score[idx] = 1.0f / (1.0f + expf(p[0] - p[1]));
}

// order-preserving float <-> unsigned key (any sign)
__device__ __forceinline__ unsigned f2k(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float k2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Top-K by score, descending, ties to the lower index: one workgroup of 1024 threads per image.  An 8-bit radix select
// over the order-preserving score keys finds the K-th value, the candidates above it plus the lowest-indexed ties are
// collected and a bitonic network in LDS sorts their 64-bit keys (score bits, ~index).  Fewer than K inputs: the tail is
// index -1 / score -1.
template <int K>
__global__ void __launch_bounds__(1024) det_topk_kernel(const float *score, int n_per, int *out_idx, float *out_score)
{
    static_assert((K & (K - 1)) == 0 && K <= 1024, "K is a power of two, one element per thread");
    __shared__ unsigned hist[16][256];          // one histogram per wave: the digits of sorted-looking scores cluster, and
                                                // 1024 threads on one set of counters serialise on the LDS atomics
    __shared__ unsigned tot[256], suf[256];
    __shared__ unsigned long long keys[K];
    __shared__ unsigned sh_prefix, sh_remaining, sh_count, scan[1024];
    const int img = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const float *s = score + (long long)img * n_per;
    const int want = n_per < K ? n_per : K;
    // radix select on the order-preserving keys
    if (tid == 0) {
        sh_prefix = 0;
        sh_remaining = (unsigned)want;
    }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int e = tid; e < 16 * 256; e += 1024)
            (&hist[0][0])[e] = 0;
        __syncthreads();
        const unsigned prefix = sh_prefix, rem = sh_remaining;   // (read by everyone before the barriers that precede their update)
        const unsigned pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n_per; i += 1024) {
            const unsigned b = f2k(s[i]);
            if ((b & pmask) == prefix)
                atomicAdd(&hist[wave][(b >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            unsigned t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w)
                t += hist[w][tid];
            tot[tid] = t;
            suf[tid] = t;
        }
        __syncthreads();
        // inclusive suffix sums over the 256 digits (suf[d] = elements with digit >= d), eight doubling steps
        for (int step = 1; step < 256; step <<= 1) {
            unsigned add = 0;
            if (tid < 256 && tid + step < 256)
                add = suf[tid + step];
            __syncthreads();
            if (tid < 256)
                suf[tid] += add;
            __syncthreads();
        }
        // the digit that holds the rem-th largest: suf[d + 1] < rem <= suf[d]  (d = 0 if fewer than rem elements remain)
        if (tid < 256) {
            const unsigned above = tid == 255 ? 0u : suf[tid + 1];
            if ((above < rem && rem <= suf[tid]) || (tid == 0 && suf[0] < rem)) {
                sh_prefix = prefix | ((unsigned)tid << shift);
                sh_remaining = rem - above;
            }
        }
        __syncthreads();
    }
    const unsigned kth = sh_prefix;        // bits of the K-th largest score
    const unsigned ties = sh_remaining;    // how many elements equal to it are taken (lowest indices first)
    if (tid == 0)
        sh_count = 0;
    for (int i = tid; i < K; i += 1024)
        keys[i] = 0ull;                    // key 0 = padding (sorts last)
    __syncthreads();
    // strictly greater: any order (the sort below fixes it); equal: blocked index ranges + scan keep the lowest indices
    const int per = (n_per + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, n_per);
    unsigned my_ties = 0;
    for (int i = lo; i < hi; ++i) {
        const unsigned b = f2k(s[i]);
        if (b > kth) {
            const unsigned slot = atomicAdd(&sh_count, 1u);
            keys[slot] = ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        } else if (b == kth)
            ++my_ties;
    }
    // exclusive scan of the 1024 tie counts (ten doubling steps)
    scan[tid] = my_ties;
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1) {
        const unsigned add = tid >= step ? scan[tid - step] : 0u;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    {
        unsigned pos = scan[tid] - my_ties;
        const unsigned base = sh_count;   // all strictly-greater keys are in (count is final after the barriers above)
        for (int i = lo; i < hi && pos < ties; ++i) {
            const unsigned b = f2k(s[i]);
            if (b == kth) {
                keys[base + pos] = ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
                ++pos;
            }
        }
    }
    __syncthreads();
    // bitonic sort, descending.  Partners less than 64 apart are in the same wave: those stages need no workgroup barrier.
    for (int size = 2; size <= K; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < K) {
                const int partner = tid ^ stride;
                if (partner > tid) {
                    const bool desc = (tid & size) == 0;
                    const unsigned long long a = keys[tid], b = keys[partner];
                    if (desc ? a < b : a > b) {
                        keys[tid] = b;
                        keys[partner] = a;
                    }
                }
            }
            if (stride >= 64 || (stride == 1 && size >= 64))   // this stage or the next one pairs elements of different waves
                __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    __syncthreads();
    for (int i = tid; i < K; i += 1024) {
        const unsigned long long k = keys[i];
        const bool valid = i < want;
        out_idx[(long long)img * K + i] = valid ? (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) : -1;
        out_score[(long long)img * K + i] = valid ? k2f((unsigned)(k >> 32)) : -1.0f;
    }
}

// faster_rcnn_box_coder decode (scale factors 10, 10, 5, 5) + clip to the window; anchors/boxes are [ymin,xmin,ymax,xmax] px
__device__ __forceinline__ void decode_clip(float ay1, float ax1, float ay2, float ax2, float ty, float tx, float th, float tw,
                                            float H, float W, float *o)
{
    const float ha = ay2 - ay1, wa = ax2 - ax1;
    const float yca = ay1 + 0.5f * ha, xca = ax1 + 0.5f * wa;
    // (size deltas are bounded like every practical decoder bounds them: exp() of an untrained head must not overflow)
    const float LIM = 4.135166556742356f;   // log(1000 / 16)
    th = fminf(th * 0.2f, LIM);
    tw = fminf(tw * 0.2f, LIM);
    const float hh = expf(th) * ha, ww = expf(tw) * wa;
    const float yc = ty * 0.1f * ha + yca, xc = tx * 0.1f * wa + xca;
    o[0] = fminf(fmaxf(yc - 0.5f * hh, 0.0f), H);
    o[1] = fminf(fmaxf(xc - 0.5f * ww, 0.0f), W);
    o[2] = fminf(fmaxf(yc + 0.5f * hh, 0.0f), H);
    o[3] = fminf(fmaxf(xc + 0.5f * ww, 0.0f), W);
}

// grid anchor a of cell (cy, cx): scale index a / 3, ratio index a % 3; centre (cy*stride, cx*stride)
__device__ __forceinline__ void anchor_box(int a, int cy, int cx, float *o)
{
    const float scales[4] = {0.25f, 0.5f, 1.0f, 2.0f};
    const float ratio_sqrt[3] = {0.70710678118654752f, 1.0f, 1.41421356237309505f};
    const float sc = scales[a / 3], rs = ratio_sqrt[a % 3];
    const float ah = sc / rs * DET_BASE, aw = sc * rs * DET_BASE;
    const float yc = (float)(cy * DET_STRIDE), xc = (float)(cx * DET_STRIDE);
    o[0] = yc - 0.5f * ah;
    o[1] = xc - 0.5f * aw;
    o[2] = yc + 0.5f * ah;
    o[3] = xc + 0.5f * aw;
}

// boxes of the K selected anchors of every image (already in descending score order)
__global__ void __launch_bounds__(256)
det_rpn_decode_kernel(const float *rpn, const int *sel, int n, int hf, int wf, int K, float H, float W, float *boxes)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * K)
        return;
    const int img = idx / K;
    const int ai = sel[idx];
    float *o = boxes + (long long)idx * 4;
    if (ai < 0) {
        o[0] = o[1] = o[2] = o[3] = 0.0f;
        return;
    }
    const int a = ai % DET_A, cell = ai / DET_A;
    const int cx = cell % wf, cy = cell / wf;
    float ab[4];
    anchor_box(a, cy, cx, ab);
    const float *d = rpn + ((long long)img * hf * wf + cell) * (6 * DET_A) + 2 * DET_A + 4 * a;
    decode_clip(ab[0], ab[1], ab[2], ab[3], d[0], d[1], d[2], d[3], H, W, o);
}

__device__ __forceinline__ float iou_yxyx(const float *a, const float *b)
{
    const float aa = (a[2] - a[0]) * (a[3] - a[1]), ab = (b[2] - b[0]) * (b[3] - b[1]);
    if (aa <= 0.0f || ab <= 0.0f)
        return 0.0f;
    const float ih = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.0f);
    const float iw = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.0f);
    const float inter = ih * iw;
    return inter / (aa + ab - inter);
}

// Batched greedy NMS over candidates that are ALREADY sorted by descending score (blockIdx.z = image):
// 64-bit suppression masks, one wave per (64 suppressors x 64 candidates) block ...
__global__ void __launch_bounds__(64)
det_nms_mask_kernel(const float *boxes, const float *scores, int K, float thr, float score_thr, unsigned long long *mask)
{
    const int img = blockIdx.z, words = K / 64;
    const int row = blockIdx.y * 64 + threadIdx.x, colb = blockIdx.x;
    if (colb < blockIdx.y)   // a box only suppresses later (lower-scored) ones
        return;
    __shared__ float cb[64][4];
    const float *bimg = boxes + (long long)img * K * 4;
    const float *simg = scores + (long long)img * K;
    {
        const float *p = bimg + (long long)(colb * 64 + threadIdx.x) * 4;
        cb[threadIdx.x][0] = p[0];
        cb[threadIdx.x][1] = p[1];
        cb[threadIdx.x][2] = p[2];
        cb[threadIdx.x][3] = p[3];
    }
    __syncthreads();
    float rb[4];
    const float *p = bimg + (long long)row * 4;
    rb[0] = p[0];
    rb[1] = p[1];
    rb[2] = p[2];
    rb[3] = p[3];
    unsigned long long bits = 0;
    if (simg[row] > score_thr)
        for (int j = 0; j < 64; ++j) {
            const int c = colb * 64 + j;
            if (c > row && iou_yxyx(rb, cb[j]) > thr)
                bits |= 1ull << j;
        }
    mask[((long long)img * K + row) * words + colb] = bits;
}
// ... and one workgroup per image walks the sorted list: all 256 threads first pull the image's whole suppression matrix
// (K x K/64 words: 128 KB at K = 1024) into LDS in one coalesced sweep, then the first wave scans it serially with the
// removed-set in registers (lane t holds word t) -- a kept candidate costs an LDS read instead of a dependent round trip to
// L2 (148 -> see profiles/ us for 16 images of 1024 candidates).  keep[img][0..max_out) = kept positions (-1 beyond the count).
__global__ void __launch_bounds__(256)
det_nms_scan_kernel(const unsigned long long *mask, const float *scores, int K, float score_thr, int max_out, int *keep, int *n_keep)
{
    extern __shared__ unsigned long long rows[];   // [K][words]
    __shared__ int n_valid_sh;
    const int img = blockIdx.x, words = K / 64, tid = threadIdx.x;
    if (tid == 0)
        n_valid_sh = 0;
    __syncthreads();
    const unsigned long long *mimg = mask + (long long)img * K * words;
    for (int e = tid; e < K * words; e += 256)
        rows[e] = mimg[e];
    const float *simg = scores + (long long)img * K;
    int cnt = 0;
    for (int e = tid; e < K; e += 256)
        cnt += simg[e] > score_thr ? 1 : 0;   // sorted descending: the valid candidates are a prefix
    if (cnt)
        atomicAdd(&n_valid_sh, cnt);
    __syncthreads();
    if (tid >= 64)
        return;
    const int n_valid = n_valid_sh;
    unsigned long long removed = 0;   // lane t: word t of the removed set
    int kept = 0;
    // word by word; inside a word jump straight to the next candidate that is still alive (most are not: the loop runs once
    // per KEPT box, not once per candidate)
    for (int wi = 0; wi < words && wi * 64 < n_valid && kept < max_out; ++wi) {
        const int left = n_valid - wi * 64;
        const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        unsigned long long done = 0;   // candidates of this word already looked at (uniform)
        while (kept < max_out) {
            const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(removed & 0xffffffffull), wi);
            const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(removed >> 32), wi);
            const unsigned long long alive = ~(((unsigned long long)hi << 32) | lo) & valid & ~done;
            if (alive == 0)
                break;
            const int b = __builtin_ctzll(alive);
            const int i = wi * 64 + b;
            done |= b == 63 ? ~0ull : ((2ull << b) - 1ull);
            if (tid == 0)
                keep[(long long)img * max_out + kept] = i;
            ++kept;
            if (tid >= wi && tid < words)
                removed |= rows[i * words + tid];
        }
    }
    for (int j = kept + tid; j < max_out; j += 64)
        keep[(long long)img * max_out + j] = -1;
    if (tid == 0)
        n_keep[img] = kept;
}

// proposals of an image = its kept RPN boxes, zero-padded to DET_PROPOSALS (as TF-OD pads); also the normalised form
// crop_and_resize takes and the image index of every box
__global__ void __launch_bounds__(256)
det_gather_proposals_kernel(const float *boxes, const int *keep, int n, int K, float H, float W, float *prop, float *prop_norm,
                            int *box_image)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * DET_PROPOSALS)
        return;
    const int img = idx / DET_PROPOSALS;
    const int k = keep[idx];
    float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k >= 0) {
        const float *p = boxes + ((long long)img * K + k) * 4;
        b[0] = p[0];
        b[1] = p[1];
        b[2] = p[2];
        b[3] = p[3];
    }
    float *o = prop + (long long)idx * 4, *on = prop_norm + (long long)idx * 4;
    o[0] = b[0];
    o[1] = b[1];
    o[2] = b[2];
    o[3] = b[3];
    on[0] = b[0] / H;
    on[1] = b[1] / W;
    on[2] = b[2] / H;
    on[3] = b[3] / W;
    box_image[idx] = img;
}

// second stage: softmax over (background, glomerulus), refined box = decode(head deltas) relative to the proposal, clipped.
// head: [n*P, 6] = 2 class logits + 4 deltas.  Padded proposals get score -1 (never detected).
__global__ void __launch_bounds__(256)
det_head_decode_kernel(const float *head, const float *prop, const int *keep, int total, float H, float W, float *score, float *boxes)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total)
        return;
    const float *h = head + (long long)idx * 6;
    const float *p = prop + (long long)idx * 4;
    score[idx] = keep[idx] >= 0 ? 1.0f / (1.0f + expf(h[0] - h[1])) : -1.0f;
    decode_clip(p[0], p[1], p[2], p[3], h[2], h[3], h[4], h[5], H, W, boxes + (long long)idx * 4);
}

// gather boxes / scores of one image through an index list (K entries per image, -1 = none)
__global__ void __launch_bounds__(256)
det_gather_kernel(const float *boxes, const int *sel, int n, int n_src, int K, float *out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * K)
        return;
    const int img = idx / K, k = sel[idx];
    float *o = out + (long long)idx * 4;
    if (k < 0) {
        o[0] = o[1] = o[2] = o[3] = 0.0f;
        return;
    }
    const float *p = boxes + ((long long)img * n_src + k) * 4;
    o[0] = p[0];
    o[1] = p[1];
    o[2] = p[2];
    o[3] = p[3];
}

// final tensors of the detect_box contract: boxes normalised by the window, scores descending, class 1.0, count
__global__ void __launch_bounds__(256)
det_output_kernel(const float *boxes, const float *scores, const int *keep, const int *n_keep, int n, int K, float H, float W,
                  float *out_boxes, float *out_scores, float *out_classes, float *out_num)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * DET_MAX_DET)
        return;
    const int img = idx / DET_MAX_DET, j = idx - img * DET_MAX_DET;
    const int k = keep[idx];
    float b[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s = 0.0f, c = 0.0f;
    if (k >= 0) {
        const float *p = boxes + ((long long)img * K + k) * 4;
        b[0] = p[0] / H;
        b[1] = p[1] / W;
        b[2] = p[2] / H;
        b[3] = p[3] / W;
        s = scores[(long long)img * K + k];
        c = 1.0f;
    }
    float *o = out_boxes + (long long)idx * 4;
    o[0] = b[0];
    o[1] = b[1];
    o[2] = b[2];
    o[3] = b[3];
    out_scores[idx] = s;
    out_classes[idx] = c;
    if (j == 0)
        out_num[img] = (float)n_keep[img];
}

// ---------------------------------------------------------------------------------------------
struct DetLayer {
    long long w = -1, b = -1;   // float offsets into the device blob
    int k = 0, cin = 0, cout = 0, stride = 1, pad = 0;
};

struct Detector {
    int device = 0;
    float *dblob = nullptr;
    DetLayer c1, c2, c3, c4, c5, c6, rpn, rpn_head, h1, h2, fc;
    float rpn_iou = 0.7f, det_iou = 0.6f, det_score = 0.0f;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // host pipeline (gs_detector_detect_host): three slots of pinned + device staging, kept across calls
    struct Slot {
        unsigned char *hin = nullptr, *din = nullptr;
        float *hres = nullptr, *dres = nullptr;   // [boxes n*D*4 | scores n*D | classes n*D | num n]
        hipEvent_t up = nullptr, down = nullptr;
        int first = -1, count = 0;
    } sl[3];
    size_t pipe_in_bytes = 0;
    int pipe_batch = 0;
    hipStream_t pipe_h2d = nullptr, pipe_compute = nullptr;
};

static void free_det_pipeline(Detector &d)
{
    for (auto &s : d.sl) {
        if (s.hin) hipHostFree(s.hin);
        if (s.hres) hipHostFree(s.hres);
        if (s.din) hipFree(s.din);
        if (s.dres) hipFree(s.dres);
        if (s.up) hipEventDestroy(s.up);
        if (s.down) hipEventDestroy(s.down);
        s = Detector::Slot();
    }
    d.pipe_in_bytes = 0;
    d.pipe_batch = 0;
}

static inline unsigned nblk(long long items) { return (unsigned)((items + 255) / 256); }

static gs_status conv(const Detector &d, const DetLayer &l, const float *in, int n, int h, int w, int relu, float *out, hipStream_t s)
{
    ConvNhwcArgs a{in, d.dblob + l.w, d.dblob + l.b, out, n, h, w, l.cin, l.k, l.k, l.cout, l.stride, l.pad, relu, 0, 0};
    return conv2d_nhwc_packed4(a, s);
}
static inline int conv_out(int x, const DetLayer &l) { return (x + 2 * l.pad - l.k) / l.stride + 1; }

}  // namespace gs

using namespace gs;

extern "C" {

struct gs_detector {
    Detector d;
};

int gs_detector_max_detections(void) { return DET_MAX_DET; }
int gs_detector_num_proposals(void) { return DET_PROPOSALS; }

gs_status gs_detector_create(const float *blob, const gs_layer_desc *table, int n_layers, gs_detector **out)
{
    GS_REQUIRE(blob && table && out && n_layers > 0, "gs_detector_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("gs_detector_create: no HIP device visible");
        return GS_ERR_NODEVICE;
    }
    std::unique_ptr<gs_detector> h(new gs_detector());
    Detector &d = h->d;
    GS_HIP(hipGetDevice(&d.device));
    // the NMS walk keeps an image's whole suppression matrix in LDS (128 KB for the 1024 pre-NMS candidates)
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(det_nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               DET_PRE_NMS * (DET_PRE_NMS / 64) * 8));
    std::map<std::string, const gs_layer_desc *> by_name;
    for (int i = 0; i < n_layers; ++i)
        by_name[std::string(table[i].name)] = &table[i];
    std::vector<float> host;
    auto take = [&](const std::string &name, DetLayer &l, int k, int cin, int cout, int stride, int pad) -> bool {
        auto wi = by_name.find(name + ".weight"), bi = by_name.find(name + ".bias");
        if (wi == by_name.end() || bi == by_name.end()) {
            set_error("detector tensor '%s' missing from the table", name.c_str());
            return false;
        }
        const gs_layer_desc *wd = wi->second, *bd = bi->second;
        if (wd->ndim != 4 || wd->shape[0] != k || wd->shape[1] != k || wd->shape[2] != cin || wd->shape[3] != cout || bd->ndim != 1 ||
            bd->shape[0] != cout) {
            set_error("detector tensor '%s' has the wrong shape (want [%d,%d,%d,%d] + [%d])", name.c_str(), k, k, cin, cout, cout);
            return false;
        }
        l.k = k;
        l.cin = cin;
        l.cout = cout;
        l.stride = stride;
        l.pad = pad;
        l.w = (long long)host.size();
        host.resize(host.size() + (size_t)k * k * cin * cout);
        // every layer of this graph has cin % 8 == 0: weights go to the device in the packed [K/4][cout][4] form
        conv2d_nhwc_pack4(blob + wd->offset, k, k, cin, cout, host.data() + l.w);
        l.b = (long long)host.size();
        host.insert(host.end(), blob + bd->offset, blob + bd->offset + cout);
        while (host.size() % 4)
            host.push_back(0.0f);
        return true;
    };
    if (!take("backbone.c1", d.c1, 3, 16, 64, 1, 1) || !take("backbone.c2", d.c2, 3, 64, 64, 1, 1) ||
        !take("backbone.c3", d.c3, 3, 64, 128, 2, 1) || !take("backbone.c4", d.c4, 3, 128, 128, 1, 1) ||
        !take("backbone.c5", d.c5, 3, 128, DET_CF, 2, 1) || !take("backbone.c6", d.c6, 3, DET_CF, DET_CF, 1, 1) ||
        !take("rpn.conv", d.rpn, 3, DET_CF, DET_CF, 1, 1) || !take("rpn.head", d.rpn_head, 1, DET_CF, 6 * DET_A, 1, 0) ||
        !take("head.h1", d.h1, 1, DET_CF, DET_CH, 1, 0) || !take("head.h2", d.h2, 3, DET_CH, DET_CH, 2, 1) ||
        !take("head.fc", d.fc, 1, DET_CH, 6, 1, 0))
        return GS_ERR_INVALID;
    GS_HIP(hipMalloc(reinterpret_cast<void **>(&d.dblob), host.size() * sizeof(float)));
    GS_HIP(hipMemcpy(d.dblob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = h.release();
    return GS_OK;
}

void gs_detector_destroy(gs_detector *h)
{
    if (!h)
        return;
    hipDeviceSynchronize();
    free_det_pipeline(h->d);
    if (h->d.pipe_h2d) hipStreamDestroy(h->d.pipe_h2d);
    if (h->d.pipe_compute) hipStreamDestroy(h->d.pipe_compute);
    if (h->d.ws) hipFree(h->d.ws);
    if (h->d.dblob) hipFree(h->d.dblob);
    delete h;
}

gs_status gs_detector_set_thresholds(gs_detector *h, float rpn_nms_iou, float det_nms_iou, float det_score_threshold)
{
    GS_REQUIRE(h, "null handle");
    GS_REQUIRE(rpn_nms_iou > 0.0f && det_nms_iou > 0.0f && det_score_threshold >= 0.0f, "thresholds must be positive (score >= 0)");
    h->d.rpn_iou = rpn_nms_iou;
    h->d.det_iou = det_nms_iou;
    h->d.det_score = det_score_threshold;
    return GS_OK;
}

// Debug taps (any may be NULL): features [N,hf,wf,256], rpn [N,hf,wf,72], proposals [N,300,4] (pixels), head [N*300,6]
gs_status gs_detector_forward(gs_detector *h, const uint8_t *images_rgb, int n, int height, int width, float *boxes, float *scores,
                              float *classes, float *num, float *dbg_features, float *dbg_rpn, float *dbg_proposals,
                              float *dbg_head, void *hip_stream)
{
    GS_REQUIRE(h && images_rgb && boxes && scores && classes && num, "gs_detector_forward: null argument");
    GS_REQUIRE(n > 0 && height >= 32 && width >= 32, "gs_detector_forward: windows must be at least 32x32 (got %dx%d, n=%d)", height,
               width, n);
    Detector &d = h->d;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int h2 = (height + 1) / 2, w2 = (width + 1) / 2;          // c1 (stride-1) size
    const int h4 = (h2 + 2 - 3) / 2 + 1, w4 = (w2 + 2 - 3) / 2 + 1;   // max-pool 3x3 s2 p1
    const int h8 = conv_out(h4, d.c3), w8 = conv_out(w4, d.c3);
    const int hf = conv_out(h8, d.c5), wf = conv_out(w8, d.c5);
    const long long cells = (long long)hf * wf, anchors = cells * DET_A;
    // (the tiled convolution addresses a whole input tensor with 32-bit byte offsets)
    GS_REQUIRE(anchors < (1 << 24) && (long long)n * h2 * w2 * 16 * 4 < 0x7fffffffLL && (long long)n * DET_PROPOSALS * 49 * DET_CF * 4 < 0x7fffffffLL,
               "gs_detector_forward: batch too large (n=%d windows of %dx%d): split it", n, height, width);
    const int P = DET_PROPOSALS, K1 = DET_PRE_NMS, K2 = 512;
    // workspace carve-up (floats unless noted)
    struct Piece { size_t off, bytes; };
    size_t total = 0;
    auto piece = [&](size_t bytes) {
        Piece p{total, bytes};
        total += (bytes + 255) / 256 * 256;
        return p;
    };
    const Piece pA = piece((size_t)n * h2 * w2 * 64 * 4);    // ping (largest: c1 output)
    const Piece pB = piece((size_t)n * h2 * w2 * 16 * 4 > (size_t)n * h4 * w4 * 64 * 4 ? (size_t)n * h2 * w2 * 16 * 4
                                                                                      : (size_t)n * h4 * w4 * 64 * 4);   // pong
    const Piece pF = piece((size_t)n * cells * DET_CF * 4);
    const Piece pR = piece((size_t)n * cells * 6 * DET_A * 4);
    const Piece pS = piece((size_t)n * anchors * 4);
    const Piece pI1 = piece((size_t)n * K1 * 4), pS1 = piece((size_t)n * K1 * 4), pB1 = piece((size_t)n * K1 * 16);
    const Piece pM = piece((size_t)n * K1 * (K1 / 64) * 8);
    const Piece pK1 = piece((size_t)n * P * 4), pN1 = piece((size_t)n * 4);
    const Piece pP = piece((size_t)n * P * 16), pPn = piece((size_t)n * P * 16), pBi = piece((size_t)n * P * 4);
    const Piece pC = piece((size_t)n * P * 49 * DET_CH * 4);      // box-head intermediates (the 14x14 crops are never materialised)
    const Piece pC2 = piece((size_t)n * P * 49 * DET_CF * 4);
    const Piece pH = piece((size_t)n * P * 6 * 4), pS2 = piece((size_t)n * P * 4), pB2 = piece((size_t)n * P * 16);
    const Piece pI2 = piece((size_t)n * K2 * 4), pS2s = piece((size_t)n * K2 * 4), pB2s = piece((size_t)n * K2 * 16);
    const Piece pK2 = piece((size_t)n * DET_MAX_DET * 4), pN2 = piece((size_t)n * 4);
    if (total > d.ws_bytes) {
        if (d.ws) {
            GS_HIP(hipDeviceSynchronize());
            GS_HIP(hipFree(d.ws));
            d.ws = nullptr;
            d.ws_bytes = 0;
        }
        if (hipMalloc(&d.ws, total) != hipSuccess) {
            set_error("detector workspace allocation of %zu bytes failed (n=%d, %dx%d)", total, n, height, width);
            return GS_ERR_NOMEM;
        }
        d.ws_bytes = total;
    }
    char *base = static_cast<char *>(d.ws);
    auto F = [&](const Piece &p) { return reinterpret_cast<float *>(base + p.off); };
    auto I = [&](const Piece &p) { return reinterpret_cast<int *>(base + p.off); };
    const float Hf = (float)height, Wf = (float)width;
    gs_status st;
#define DET_TRY(x) do { st = (x); if (st != GS_OK) return st; } while (0)

    // ---- backbone
    hipLaunchKernelGGL(det_preprocess_kernel, dim3(nblk((long long)n * h2 * w2 * 4)), dim3(256), 0, s, images_rgb, n, height, width, F(pB));
    DET_TRY(conv(d, d.c1, F(pB), n, h2, w2, 1, F(pA), s));
    hipLaunchKernelGGL(det_maxpool_kernel, dim3(nblk((long long)n * h4 * w4 * 16)), dim3(256), 0, s, F(pA), (long long)n, h2, w2, 64, 3, 2, 1,
                       h4, w4, F(pB));
    DET_TRY(conv(d, d.c2, F(pB), n, h4, w4, 1, F(pA), s));
    DET_TRY(conv(d, d.c3, F(pA), n, h4, w4, 1, F(pB), s));
    DET_TRY(conv(d, d.c4, F(pB), n, h8, w8, 1, F(pA), s));
    DET_TRY(conv(d, d.c5, F(pA), n, h8, w8, 1, F(pB), s));
    DET_TRY(conv(d, d.c6, F(pB), n, hf, wf, 1, F(pF), s));
    // ---- region proposal network
    DET_TRY(conv(d, d.rpn, F(pF), n, hf, wf, 1, F(pA), s));
    DET_TRY(conv(d, d.rpn_head, F(pA), n, hf, wf, 0, F(pR), s));
    hipLaunchKernelGGL(det_objectness_kernel, dim3(nblk((long long)n * anchors)), dim3(256), 0, s, F(pR), (long long)n * anchors, F(pS));
    hipLaunchKernelGGL(det_topk_kernel<DET_PRE_NMS>, dim3(n), dim3(1024), 0, s, F(pS), (int)anchors, I(pI1), F(pS1));
    hipLaunchKernelGGL(det_rpn_decode_kernel, dim3(nblk((long long)n * K1)), dim3(256), 0, s, F(pR), I(pI1), n, hf, wf, K1, Hf, Wf, F(pB1));
    hipLaunchKernelGGL(det_nms_mask_kernel, dim3(K1 / 64, K1 / 64, n), dim3(64), 0, s, F(pB1), F(pS1), K1, d.rpn_iou, 0.0f,
                       reinterpret_cast<unsigned long long *>(base + pM.off));
    hipLaunchKernelGGL(det_nms_scan_kernel, dim3(n), dim3(256), (size_t)K1 * (K1 / 64) * 8, s, reinterpret_cast<unsigned long long *>(base + pM.off), F(pS1), K1, 0.0f,
                       P, I(pK1), I(pN1));
    hipLaunchKernelGGL(det_gather_proposals_kernel, dim3(nblk((long long)n * P)), dim3(256), 0, s, F(pB1), I(pK1), n, K1, Hf, Wf, F(pP),
                       F(pPn), I(pBi));
    // ---- box head on crop_and_resize'd features
    hipLaunchKernelGGL(det_crop_pool_kernel, dim3(nblk((long long)n * P * 49 * (DET_CF / 4))), dim3(256), 0, s, F(pF), n, hf, wf, DET_CF, F(pPn),
                       I(pBi), n * P, DET_CROP, F(pC2));
    DET_TRY(conv(d, d.h1, F(pC2), n * P, 7, 7, 1, F(pC), s));             // [nP,7,7,128]
    DET_TRY(conv(d, d.h2, F(pC), n * P, 7, 7, 1, F(pC2), s));             // [nP,4,4,128]
    hipLaunchKernelGGL(det_avgpool_kernel, dim3(nblk((long long)n * P * DET_CH)), dim3(256), 0, s, F(pC2), (long long)n * P, 16, DET_CH, F(pC));
    DET_TRY(conv(d, d.fc, F(pC), n * P, 1, 1, 0, F(pH), s));              // [nP,6]
    hipLaunchKernelGGL(det_head_decode_kernel, dim3(nblk((long long)n * P)), dim3(256), 0, s, F(pH), F(pP), I(pK1), n * P, Hf, Wf, F(pS2),
                       F(pB2));
    // ---- per-class (one foreground class) NMS, score-sorted, padded outputs
    hipLaunchKernelGGL(det_topk_kernel<512>, dim3(n), dim3(1024), 0, s, F(pS2), P, I(pI2), F(pS2s));
    hipLaunchKernelGGL(det_gather_kernel, dim3(nblk((long long)n * K2)), dim3(256), 0, s, F(pB2), I(pI2), n, P, K2, F(pB2s));
    hipLaunchKernelGGL(det_nms_mask_kernel, dim3(K2 / 64, K2 / 64, n), dim3(64), 0, s, F(pB2s), F(pS2s), K2, d.det_iou, d.det_score,
                       reinterpret_cast<unsigned long long *>(base + pM.off));
    hipLaunchKernelGGL(det_nms_scan_kernel, dim3(n), dim3(256), (size_t)K2 * (K2 / 64) * 8, s, reinterpret_cast<unsigned long long *>(base + pM.off), F(pS2s), K2,
                       d.det_score, DET_MAX_DET, I(pK2), I(pN2));
    hipLaunchKernelGGL(det_output_kernel, dim3(nblk((long long)n * DET_MAX_DET)), dim3(256), 0, s, F(pB2s), F(pS2s), I(pK2), I(pN2), n, K2, Hf,
                       Wf, boxes, scores, classes, num);
    GS_HIP(hipGetLastError());
    if (dbg_features) GS_HIP(hipMemcpyAsync(dbg_features, F(pF), (size_t)n * cells * DET_CF * 4, hipMemcpyDeviceToDevice, s));
    if (dbg_rpn) GS_HIP(hipMemcpyAsync(dbg_rpn, F(pR), (size_t)n * cells * 6 * DET_A * 4, hipMemcpyDeviceToDevice, s));
    if (dbg_proposals) GS_HIP(hipMemcpyAsync(dbg_proposals, F(pP), (size_t)n * P * 16, hipMemcpyDeviceToDevice, s));
    if (dbg_head) GS_HIP(hipMemcpyAsync(dbg_head, F(pH), (size_t)n * P * 6 * 4, hipMemcpyDeviceToDevice, s));
#undef DET_TRY
    return GS_OK;
}

gs_status gs_detector_detect_host(gs_detector *h, const uint8_t *const *windows, int n, int height, int width, int batch, float *boxes,
                                  float *scores, float *classes, float *num)
{
    GS_REQUIRE(h && windows && boxes && scores && classes && num, "gs_detector_detect_host: null argument");
    GS_REQUIRE(n > 0 && batch > 0, "n and batch must be positive");
    for (int i = 0; i < n; ++i)
        GS_REQUIRE(windows[i], "window %d is a null pointer", i);
    if (batch > n) batch = n;
    Detector &d = h->d;
    constexpr int NSLOT = 3, D = DET_MAX_DET;
    const size_t in_b = (size_t)height * width * 3;
    const size_t res_f = (size_t)batch * (D * 6 + 1);   // floats per slot
    gs_status rc = GS_OK;
    auto fail = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == GS_OK) {
            set_error("%s failed: %s", what, hipGetErrorString(e));
            rc = GS_ERR_HIP;
        }
        return e != hipSuccess;
    };
    if (!d.pipe_h2d || !d.pipe_compute) {   // both or neither: a partial pair would run later calls on the NULL stream
        int lo = 0, hi = 0;
        fail(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange");
        if (rc == GS_OK && !d.pipe_h2d)
            fail(hipStreamCreateWithPriority(&d.pipe_h2d, hipStreamNonBlocking, hi), "hipStreamCreate");
        if (rc == GS_OK && !d.pipe_compute)
            fail(hipStreamCreateWithPriority(&d.pipe_compute, hipStreamNonBlocking, lo), "hipStreamCreate");
        if (rc != GS_OK) {
            if (d.pipe_h2d) hipStreamDestroy(d.pipe_h2d);
            if (d.pipe_compute) hipStreamDestroy(d.pipe_compute);
            d.pipe_h2d = d.pipe_compute = nullptr;
            return rc;
        }
    }
    if (d.pipe_in_bytes < in_b * batch || d.pipe_batch < batch) {
        fail(hipDeviceSynchronize(), "hipDeviceSynchronize");
        free_det_pipeline(d);
        for (auto &s : d.sl) {
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hin), in_b * batch, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hres), res_f * sizeof(float), hipHostMallocDefault), "hipHostMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.din), in_b * batch), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dres), res_f * sizeof(float)), "hipMalloc");
            fail(hipEventCreateWithFlags(&s.up, hipEventDisableTiming), "hipEventCreate");
            fail(hipEventCreateWithFlags(&s.down, hipEventDisableTiming), "hipEventCreate");
        }
        if (rc != GS_OK) {
            free_det_pipeline(d);
            return rc;
        }
        d.pipe_in_bytes = in_b * batch;
        d.pipe_batch = batch;
    }
    for (auto &s : d.sl)
        s.first = -1;
    const int pb = d.pipe_batch;   // the slot layout follows the batch the buffers were made for
    auto drain = [&](Detector::Slot &s) {
        if (s.first < 0 || rc != GS_OK)
            return;
        if (fail(hipEventSynchronize(s.down), "hipEventSynchronize")) return;
        const float *r = s.hres;
        std::memcpy(boxes + (size_t)s.first * D * 4, r, sizeof(float) * s.count * D * 4);
        std::memcpy(scores + (size_t)s.first * D, r + (size_t)pb * D * 4, sizeof(float) * s.count * D);
        std::memcpy(classes + (size_t)s.first * D, r + (size_t)pb * D * 5, sizeof(float) * s.count * D);
        std::memcpy(num + s.first, r + (size_t)pb * D * 6, sizeof(float) * s.count);
        s.first = -1;
    };
    int slot = 0;
    for (int first = 0; first < n && rc == GS_OK; first += batch, slot = (slot + 1) % NSLOT) {
        Detector::Slot &s = d.sl[slot];
        drain(s);
        if (rc != GS_OK) break;
        const int cnt = n - first < batch ? n - first : batch;
        bool direct = true;
        for (int j = 0; j < cnt; ++j)
            direct = direct && host_is_pinned(windows[first + j]);
        if (direct) {
            for (int j = 0; j < cnt && rc == GS_OK; ++j)
                fail(hipMemcpyAsync(s.din + (size_t)j * in_b, windows[first + j], in_b, hipMemcpyHostToDevice, d.pipe_h2d), "H2D copy");
        } else {
            parallel_jobs(cnt, 4, [&](int j) { std::memcpy(s.hin + (size_t)j * in_b, windows[first + j], in_b); });
            fail(hipMemcpyAsync(s.din, s.hin, in_b * cnt, hipMemcpyHostToDevice, d.pipe_h2d), "H2D copy");
        }
        if (rc != GS_OK) break;
        fail(hipEventRecord(s.up, d.pipe_h2d), "hipEventRecord");
        fail(hipStreamWaitEvent(d.pipe_compute, s.up, 0), "hipStreamWaitEvent");
        float *r = s.dres;
        gs_status st2 = gs_detector_forward(h, s.din, cnt, height, width, r, r + (size_t)pb * D * 4, r + (size_t)pb * D * 5,
                                            r + (size_t)pb * D * 6, nullptr, nullptr, nullptr, nullptr, d.pipe_compute);
        if (st2 != GS_OK) { rc = st2; break; }
        const size_t rb = (size_t)pb * (D * 6 + 1) * sizeof(float);
        fail(hipMemcpy2DAsync(s.hres, rb, s.dres, rb, rb, 1, hipMemcpyDeviceToHost, d.pipe_compute), "D2H copy");
        fail(hipEventRecord(s.down, d.pipe_compute), "hipEventRecord");
        s.first = first;
        s.count = cnt;
    }
    for (int k = 0; k < NSLOT; ++k)
        drain(d.sl[(slot + k) % NSLOT]);
    if (rc != GS_OK)
        hipDeviceSynchronize();
    return rc;
}

}  // extern "C"
