// Sampling rules of the crop stage, shared by the per-crop kernels (detect_ops.hip: gs_crop_preprocess,
// gs_mask_resize_nearest) and the batched descriptor-table kernels (crops.hip), so that both produce the same bits.
// reference: module/espnet/test/VisualizeResults_iou.py:107-116 (normalise -> cv2.resize INTER_LINEAR -> /255) and :129
// (cv2.resize INTER_NEAREST back to the crop size).
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

// cv2.resize INTER_LINEAR on a float image: fx = (float)((dx+0.5)*scale-0.5), sx = floor(fx), fx -= sx, clamped at both
// borders; horizontal pass first, then vertical.
__device__ __forceinline__ void linear_tap(int d, double scale, int n, int &i0, int &i1, float &w1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
        s = 0;
        f = 0.0f;
    }
    if (s >= n - 1) {
        s = n - 1;
        f = 0.0f;
    }
    i0 = s;
    i1 = s + 1 < n ? s + 1 : n - 1;
    w1 = f;
}
// OpenCV's own expression for the scale: 1. / inv_scale with inv_scale = (double)dst / src
__device__ __forceinline__ double cv_inv_scale(int dst, int src) { return 1.0 / ((double)dst / (double)src); }

// One output value of channel c: the four normalised taps ((x - mean) / std, two fp32 roundings as numpy does them),
// the horizontal blend of each row, the vertical blend, /255.  Every product and sum is rounded on its own (no fused
// multiply-add), as the two-pass CPU implementations round them.
__device__ __forceinline__ float crop_sample(const unsigned char *src, int w, int c, int x0, int x1, int y0, int y1, float wx, float wy,
                                             float mean, float stdv)
{
#pragma clang fp contract(off)
    const unsigned char *r0 = src + ((long long)y0 * w) * 3 + c, *r1 = src + ((long long)y1 * w) * 3 + c;
    const float a00 = ((float)r0[x0 * 3] - mean) / stdv, a01 = ((float)r0[x1 * 3] - mean) / stdv;
    const float a10 = ((float)r1[x0 * 3] - mean) / stdv, a11 = ((float)r1[x1 * 3] - mean) / stdv;
    const float ux = 1.0f - wx, uy = 1.0f - wy;
    const float top = a00 * ux + a01 * wx;
    const float bot = a10 * ux + a11 * wx;
    const float v = top * uy + bot * wy;
    return v / 255.0f;
}

// OpenCV resizeNN: ifx = 1. / fx with fx = (double)dst / src; sx = min(cvFloor(x * ifx), src - 1)
__device__ __forceinline__ int nearest_src(int d, double ifx, int src)
{
    const int s = (int)floor((double)d * ifx);
    return s < src - 1 ? s : src - 1;
}

}  // namespace gs
