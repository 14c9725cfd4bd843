/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the ESPNet per-patch forward pass.
 *
 * Plain-C restatement of the arithmetic the reference delegates to PyTorch's
 * CPU kernels in module/espnet/test/Model.py (nn.Conv2d / BatchNorm2d / PReLU /
 * AvgPool2d / ConvTranspose2d).  The graph itself (which layer feeds which) is
 * composed in oracle/espnet_oracle.py, following Model.py:341-378 line by line.
 *
 * Parity pinning: checked against golden vectors produced by importing the
 * reference itself (tests/golden/make_golden.py -> tests/golden/*.npz), see
 * tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this.  Nothing in glomeruli_segmentation_amd/ links or imports it.
 *
 * All tensors are fp32, single image, CHW contiguous.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* nn.Conv2d(cin, cout, (k,k), stride, padding=((k-1)/2)*d, dilation=d, bias=False)
 * reference: Model.py:96 (C), Model.py:119-120 (CDilated), Model.py:20 (CBR.conv) */
void gso_conv2d(const float *x, int cin, int h, int w, const float *wt, int cout, int k, int stride,
                int dil, float *y)
{
    const int pad = ((k - 1) / 2) * dil;
    const int ho = (h + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    const int wo = (w + 2 * pad - dil * (k - 1) - 1) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < cout; ++co) {
        for (int oy = 0; oy < ho; ++oy) {
            float *yr = y + ((size_t)co * ho + oy) * wo;
            memset(yr, 0, sizeof(float) * wo);
            for (int ci = 0; ci < cin; ++ci) {
                for (int ky = 0; ky < k; ++ky) {
                    const int iy = oy * stride - pad + ky * dil;
                    if (iy < 0 || iy >= h)
                        continue;
                    const float *xr = x + ((size_t)ci * h + iy) * w;
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wt[(((size_t)co * cin + ci) * k + ky) * k + kx];
                        const int off = kx * dil - pad; /* ix = ox*stride + off */
                        int lo = 0, hi = wo;
                        if (off < 0)
                            lo = (-off + stride - 1) / stride;
                        if (w - 1 - off < 0)
                            hi = 0;
                        else if ((hi - 1) * stride + off >= w)
                            hi = (w - 1 - off) / stride + 1;
                        if (stride == 1) {
                            const float *xs = xr + off;
                            for (int ox = lo; ox < hi; ++ox)
                                yr[ox] += wv * xs[ox];
                        } else {
                            for (int ox = lo; ox < hi; ++ox)
                                yr[ox] += wv * xr[ox * stride + off];
                        }
                    }
                }
            }
        }
    }
}

/* nn.BatchNorm2d(c, eps) in eval mode: (x-mean)/sqrt(var+eps)*gamma+beta. reference: Model.py:21,44,70,141,331 */
void gso_bn_eval(float *x, int c, int hw, const float *gamma, const float *beta, const float *mean,
                 const float *var, float eps)
{
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch) {
        const float inv = 1.0f / sqrtf(var[ch] + eps);
        float *p = x + (size_t)ch * hw;
        for (int i = 0; i < hw; ++i)
            p[i] = (p[i] - mean[ch]) * inv * gamma[ch] + beta[ch];
    }
}

/* nn.PReLU(c): per-channel slope on the negative side. reference: Model.py:22,45,142 */
void gso_prelu(float *x, int c, int hw, const float *alpha)
{
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch) {
        float *p = x + (size_t)ch * hw;
        const float a = alpha[ch];
        for (int i = 0; i < hw; ++i)
            p[i] = p[i] > 0.0f ? p[i] : a * p[i];
    }
}

/* nn.AvgPool2d(3, stride=2, padding=1); count_include_pad defaults to True, so the divisor is
 * always 9. reference: Model.py:230 */
void gso_avgpool3s2(const float *x, int c, int h, int w, float *y)
{
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch)
        for (int oy = 0; oy < ho; ++oy)
            for (int ox = 0; ox < wo; ++ox) {
                float s = 0.0f;
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = oy * 2 - 1 + ky;
                    if (iy < 0 || iy >= h)
                        continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = ox * 2 - 1 + kx;
                        if (ix < 0 || ix >= w)
                            continue;
                        s += x[((size_t)ch * h + iy) * w + ix];
                    }
                }
                y[((size_t)ch * ho + oy) * wo + ox] = s / 9.0f;
            }
}

/* nn.ConvTranspose2d(cin, cout, 2, stride=2, padding=0, bias=False): windows do not overlap,
 * out[o,2y+a,2x+b] = sum_i in[i,y,x]*W[i,o,a,b]. reference: Model.py:334,337,339 */
void gso_deconv2x2s2(const float *x, int cin, int h, int w, const float *wt, int cout, float *y)
{
    const int ho = 2 * h, wo = 2 * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < cout; ++co)
        for (int iy = 0; iy < h; ++iy)
            for (int ix = 0; ix < w; ++ix)
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        float s = 0.0f;
                        for (int ci = 0; ci < cin; ++ci)
                            s += x[((size_t)ci * h + iy) * w + ix] * wt[(((size_t)ci * cout + co) * 2 + a) * 2 + b];
                        y[((size_t)co * ho + 2 * iy + a) * wo + 2 * ix + b] = s;
                    }
}

/* img_out[0].max(0)[1].byte(): index of the first maximum over channels.
 * reference: VisualizeResults_iou.py:128 */
void gso_argmax(const float *logits, int c, int hw, uint8_t *mask)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < hw; ++i) {
        int best = 0;
        float bv = logits[i];
        for (int ch = 1; ch < c; ++ch) {
            const float v = logits[(size_t)ch * hw + i];
            if (v > bv) {
                bv = v;
                best = ch;
            }
        }
        mask[i] = (uint8_t)best;
    }
}

/* (x - mean_c)/std_c, then /255, HWC uint8 BGR -> CHW fp32, in fp32 like numpy does.
 * reference: VisualizeResults_iou.py:107-117 (resize at :114 is the identity at equal size) */
void gso_normalize_u8(const uint8_t *hwc, int h, int w, const float *mean, const float *std, float *chw)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = (float)hwc[((size_t)y * w + x) * 3 + c];
                v = v - mean[c];
                v = v / std[c];
                v = v / 255.0f;
                chw[((size_t)c * h + y) * w + x] = v;
            }
}
