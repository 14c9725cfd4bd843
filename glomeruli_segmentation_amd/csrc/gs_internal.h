// Shared internals of libglomseg.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/glomseg.h"

namespace gs {

void set_error(const char *fmt, ...);

#define GS_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            gs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GS_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define GS_REQUIRE(cond, ...)                                                                     \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            gs::set_error(__VA_ARGS__);                                                           \
            return GS_ERR_INVALID;                                                                \
        }                                                                                         \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long round_up(long long a, long long b) { return (a + b - 1) / b * b; }

// A [N][C][H][W] fp32 activation in HBM with a zero halo.  `base` is the start of the allocation of
// image 0; (n,c,y,x) lives at base + n*sn + c*sc + off + y*pitch + x (all in floats).  Kernels only
// ever write the interior, so the halo (zeroed when the workspace is laid out) stays zero and the
// convolution taps that fall outside the image read exact zeros without any predication.
struct Act {
    float *base = nullptr;
    long long sn = 0;   // floats per image
    int sc = 0;         // floats per channel plane
    int pitch = 0;      // floats per row
    int off = 0;        // pad_top*pitch + pad_left
    int C = 0, Cp = 0;  // real / allocated channel planes (extra planes stay zero)
    int H = 0, W = 0;
    size_t bytes(int n) const { return (size_t)n * sn * sizeof(float); }
};

// NHWC convolution arguments (csrc/detect_ops.hip); ho / wo are filled in by the launchers
struct ConvNhwcArgs {
    const float *in, *w, *bias;
    float *out;
    int n, h, w_, cin, kh, kw, cout, stride, pad, relu, ho, wo;
};
// packed-weight form of the tiled NHWC convolution for handles that own their weights (detect_ops.hip)
bool conv2d_nhwc_can_pack4(int n, int h, int w, int cin, int kh, int kw, int cout);
void conv2d_nhwc_pack4(const float *w, int kh, int kw, int cin, int cout, float *dst);
gs_status conv2d_nhwc_packed4(ConvNhwcArgs a, hipStream_t stream);


// ---- espnet.hip internals that the crop pipeline (crops.hip) builds on
struct CropPipe;                                   // staging state of gs_espnet_segment_crops*, owned by the handle
CropPipe *&espnet_crop_pipe(gs_espnet *h);
void crop_pipe_destroy(CropPipe *p);               // crops.hip; called by gs_espnet_destroy
int espnet_device(gs_espnet *h);
int espnet_is_full_net(gs_espnet *h);
int espnet_lanes(gs_espnet *h);
int espnet_classes(gs_espnet *h);
gs_status ensemble_scratch(gs_espnet *h, int n, int height, int width, float **prob);
gs_status espnet_forward_ex(gs_espnet *h, int lane, const void *in, int in_format, int n, int height, int width, const float *mean,
                            const float *stdv, float *logits, uint8_t *mask, unsigned long long *hist, float *prob, int ens_mode,
                            float ens_w, hipStream_t s);

}  // namespace gs
