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

// (x - mean) / std of one uint8 value, two fp32 roundings as numpy does them (VisualizeResults_iou.py:109,111).  The kernels
// tabulate it once per workgroup (3 x 256 entries in LDS): the same bits as computing it per tap, without the IEEE
// divisions -- 48 per output pixel group -- that made the resampling kernels VALU-bound.
__device__ __forceinline__ float crop_norm(int b, float mean, float stdv)
{
#pragma clang fp contract(off)
    return ((float)b - mean) / stdv;
}
__device__ __forceinline__ void crop_norm_table(float (*lut)[256], const float mean[3], const float stdv[3])
{
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x)
        lut[i >> 8][i & 255] = crop_norm(i & 255, mean[i >> 8], stdv[i >> 8]);
    __syncthreads();
}

// One output value of channel c from the table: the horizontal blend of each row, the vertical blend, /255.  Every product
// and sum is rounded on its own (no fused multiply-add), as the two-pass CPU implementations round them.
__device__ __forceinline__ float crop_sample(const unsigned char *src, int w, int c, int x0, int x1, int y0, int y1, float wx, float wy,
                                             const float *lut_c)
{
#pragma clang fp contract(off)
    const unsigned char *r0 = src + ((long long)y0 * w) * 3 + c, *r1 = src + ((long long)y1 * w) * 3 + c;
    const float a00 = lut_c[r0[x0 * 3]], a01 = lut_c[r0[x1 * 3]];
    const float a10 = lut_c[r1[x0 * 3]], a11 = lut_c[r1[x1 * 3]];
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
