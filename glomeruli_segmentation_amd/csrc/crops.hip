// gs_espnet_segment_crops*: the per-patch loop of module/espnet/test/VisualizeResults_iou.py:100-156 for the crops a slide
// really produces -- every merged box at its own size (module/faster-rcnn/make_seg_data.py:357-361) -- a whole batch per
// launch.  The stages either side of the forward are descriptor-table kernels: the table (up to 64 crops) travels as a
// kernel argument, so a call allocates nothing, copies nothing and is stream-ordered like any other launch.
//
//   crops_prep_kernel    :107-116  (x - mean) / std at crop resolution -> cv2.resize INTER_LINEAR -> / 255 -> fp32 NCHW
//   (forward)            :123,128  espnet_forward_ex: the mask comes out of the decoder tail, no logits
//   crops_back_kernel    :129,151-155  cv2.resize INTER_NEAREST back to every crop's size + per-class counts of THAT map
//   crops_paste_kernel   eval_wsi_segmentation.py:311-312  np.max into the 1/ds slide map
//
// All four are bandwidth-bound byte / fp32 streams (bound: HBM); per 1024x512 network tile and ~0.6 Mpx crop they move
// 6.3 MB (fp32 tensor out) + ~1.8 MB (crop in), 0.5 MB + 0.6 MB, and a few KB.
#include <algorithm>
#include <memory>
#include <vector>

#include "crop_plan.h"
#include "crop_sample.h"
#include "gs_internal.h"
#include "host_copy.h"

namespace gs {

constexpr int MAXC = GS_MAX_CROPS_PER_CALL;
struct CropTable {
    gs_crop_desc d[MAXC];
};

struct PrepArgs {
    const unsigned char *in;   // packed crops
    float *out;                // [n][3][net_h][net_w]
    int net_h, net_w;
    float mean[3], std[3];
    unsigned long long *hist_zero;   // optional: counters crops_back_kernel adds into, zeroed here
    int hist_count;
};

// one thread = four consecutive output columns of one row, all three channels: 16-byte stores
__global__ void __launch_bounds__(256) crops_prep_kernel(const CropTable t, const PrepArgs a)
{
    __shared__ float lut[3][256];
    crop_norm_table(lut, a.mean, a.std);
    const int i = blockIdx.y;
    if (a.hist_zero && blockIdx.x == 0 && i == 0)
        for (int k = threadIdx.x; k < a.hist_count; k += 256)
            a.hist_zero[k] = 0ull;
    const int w4 = a.net_w / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.net_h * w4)
        return;
    const int oy = idx / w4, ox = (idx - oy * w4) * 4;
    const int h = t.d[i].h, w = t.d[i].w;
    const unsigned char *src = a.in + t.d[i].in_off;
    const double sx = cv_inv_scale(a.net_w, w), sy = cv_inv_scale(a.net_h, h);
    int y0, y1, x0[4], x1[4];
    float wy, wx[4];
    linear_tap(oy, sy, h, y0, y1, wy);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        linear_tap(ox + k, sx, w, x0[k], x1[k], wx[k]);
    float *dst = a.out + (((long long)i * 3) * a.net_h + oy) * a.net_w + ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float4 v;
        v.x = crop_sample(src, w, c, x0[0], x1[0], y0, y1, wx[0], wy, lut[c]);
        v.y = crop_sample(src, w, c, x0[1], x1[1], y0, y1, wx[1], wy, lut[c]);
        v.z = crop_sample(src, w, c, x0[2], x1[2], y0, y1, wx[2], wy, lut[c]);
        v.w = crop_sample(src, w, c, x0[3], x1[3], y0, y1, wx[3], wy, lut[c]);
        *reinterpret_cast<float4 *>(dst + (long long)c * a.net_h * a.net_w) = v;
    }
}

// Nearest resize of every network-resolution mask back to its crop's size, four consecutive bytes of the flat crop map per
// thread (one 32-bit store), and the per-class counts of the crop-size map: packed per-lane counters (12 bits per class, at
// most 4 added per iteration and the grid is sized for <= 512 iterations), a butterfly add over the wave, one LDS atomic per
// class per wave, one global atomic per class per workgroup.  NW = 64-bit counter words per lane, five classes each (1 for the
// five-class networks, up to 4 for GS_MAX_CLASSES = 20); hist is [n][classes].
template <int NW>
__global__ void __launch_bounds__(256)
crops_back_kernel(const CropTable t, const unsigned char *net, int net_h, int net_w, unsigned char *out, unsigned long long *hist, int classes)
{
    __shared__ unsigned lh[5 * NW];
    const int i = blockIdx.y;
    if (threadIdx.x < 5 * NW)
        lh[threadIdx.x] = 0;
    __syncthreads();
    const int h = t.d[i].h, w = t.d[i].w;
    const long long hw = (long long)h * w;
    const unsigned char *src = net + (long long)i * net_h * net_w;
    unsigned char *dst = out ? out + t.d[i].out_off : nullptr;
    const double ifx = cv_inv_scale(w, net_w), ify = cv_inv_scale(h, net_h);
    unsigned long long counts[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q)
        counts[q] = 0;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; base < hw; base += (long long)gridDim.x * 1024) {
        int oy = (int)(base / w), ox = (int)(base - (long long)oy * w);
        unsigned packed = 0;
        int sy = nearest_src(oy, ify, net_h);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (base + k < hw) {
                const unsigned v = src[(long long)sy * net_w + nearest_src(ox, ifx, net_w)];
                packed |= v << (8 * k);
                const unsigned vc = v < (unsigned)classes ? v : 0;   // (a value no class map holds counts as background)
                if (NW == 1) {
                    counts[0] += 1ull << (12 * vc);
                } else {
#pragma unroll
                    for (int q = 0; q < NW; ++q)
                        if (vc / 5 == (unsigned)q)
                            counts[q] += 1ull << (12 * (vc % 5));
                }
            }
            if (++ox == w) {   // the four bytes may run over a row end
                ox = 0;
                ++oy;
                sy = nearest_src(oy, ify, net_h);
            }
        }
        if (dst) {
            if (base + 4 <= hw)
                *reinterpret_cast<unsigned *>(dst + base) = packed;   // out_off and base are multiples of 4
            else
                for (int k = 0; base + k < hw; ++k)
                    dst[base + k] = (unsigned char)(packed >> (8 * k));
        }
    }
    if (hist) {
#pragma unroll
        for (int k = 0; k < 5 * NW; ++k) {
            int c = (int)((counts[k / 5] >> (12 * (k % 5))) & 0xfffull);
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1)
                c += __shfl_xor(c, sh, 64);
            if ((threadIdx.x & 63) == 0 && c)
                atomicAdd(&lh[k], (unsigned)c);
        }
        __syncthreads();
        if ((int)threadIdx.x < classes && lh[threadIdx.x])
            atomicAdd(&hist[(long long)i * classes + threadIdx.x], (unsigned long long)lh[threadIdx.x]);
    }
}

// Palette colouring + cv2.addWeighted of every crop of the batch (VisualizeResults_iou.py:139-146): four pixels per thread -- one
// dword of the crop-size class map, three dwords of BGR in, three out.  The two products and their sum are rounded separately
// (__fmul_rn / __fadd_rn: no fused multiply-add), so the bytes are those of numpy's float32 arithmetic.
struct OverlayArgs {
    const unsigned char *crops;   // packed BGR crops (gs_crop_desc::in_off)
    const unsigned char *maps;    // packed crop-size class maps (out_off)
    unsigned char *out;           // packed overlays, at in_off
    float wa, wb;
    int n_colours;
    unsigned char pal[GS_MAX_PALETTE * 3];   // RGB rows
};

__global__ void __launch_bounds__(256) crops_overlay_kernel(const CropTable t, const OverlayArgs a)
{
    const int i = blockIdx.y;
    const long long hw = (long long)t.d[i].h * t.d[i].w;
    const unsigned char *src = a.crops + t.d[i].in_off, *cls = a.maps + t.d[i].out_off;
    unsigned char *dst = a.out + t.d[i].in_off;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; base < hw; base += (long long)gridDim.x * 1024) {
        const int np = hw - base < 4 ? (int)(hw - base) : 4;
        unsigned cw = 0, pw[3] = {0, 0, 0}, ow[3] = {0, 0, 0};   // bytes little-endian in dwords
        if (np == 4) {   // (in_off / out_off are multiples of 256 and base of 4: aligned dwords)
            cw = *reinterpret_cast<const unsigned *>(cls + base);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                pw[k] = reinterpret_cast<const unsigned *>(src + base * 3)[k];
        } else {
            for (int k = 0; k < np; ++k)
                cw |= (unsigned)cls[base + k] << (8 * k);
            for (int k = 0; k < np * 3; ++k)
                pw[k >> 2] |= (unsigned)src[base * 3 + k] << (8 * (k & 3));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (int)((cw >> (8 * k)) & 0xffu);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int bi = k * 3 + ch;
                const float pix = (float)((pw[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
                // palette rows are RGB, the image is BGR (classMap_numpy_color[...] = [b, g, r], :143)
                const float col = c < a.n_colours ? (float)a.pal[c * 3 + (2 - ch)] : 0.0f;
                const float v = __fadd_rn(__fmul_rn(pix, a.wa), __fmul_rn(col, a.wb));
                ow[bi >> 2] |= (unsigned)fminf(fmaxf(rintf(v), 0.0f), 255.0f) << (8 * (bi & 3));   // saturate_cast<uchar>(cvRound)
            }
        }
        if (np == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                reinterpret_cast<unsigned *>(dst + base * 3)[k] = ow[k];
        } else {
            for (int k = 0; k < np * 3; ++k)
                dst[base * 3 + k] = (unsigned char)(ow[k >> 2] >> (8 * (k & 3)));
        }
    }
}

// per-byte maximum into a map that other crops of the same launch (or of the other compute stream) may be writing
__device__ __forceinline__ void byte_max(unsigned char *p, unsigned v)
{
    if (v == 0)
        return;   // the map starts at zero: a background pixel never changes it
    unsigned *wp = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned long long>(p) & ~3ull);
    const int sh = (int)(reinterpret_cast<unsigned long long>(p) & 3ull) * 8;
    unsigned old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (((old >> sh) & 0xffu) < v) {
        const unsigned nv = (old & ~(0xffu << sh)) | (v << sh);
        const unsigned prev = atomicCAS(wp, old, nv);
        if (prev == old)
            break;
        old = prev;
    }
}

// Max-composite of every crop's class map into the 1/ds slide map.  Map cell (X, Y) shows level-0 pixel (ds*X, ds*Y) --
// or (sx[X], sy[Y]) under the reference's window walk (gs_wsi_paste_max_lut) -- i.e. crop pixel (px - x1, py - y1), whose
// class is the network-resolution mask at that pixel's INTER_NEAREST source: the crop-size map need not exist.
__global__ void __launch_bounds__(256)
crops_paste_kernel(const CropTable t, const unsigned char *net, int net_h, int net_w, const gs_paste_target p)
{
    const int i = blockIdx.y;
    const int h = t.d[i].h, w = t.d[i].w, x1 = t.d[i].x1, y1 = t.d[i].y1;
    const unsigned char *src = net + (long long)i * net_h * net_w;
    const int ds = p.ds;
    // candidate cells: within one cell of the crop's footprint on the regular grid (the tables are monotone with steps >= ds)
    const int X0 = x1 / ds - 1, Y0 = y1 / ds - 1;
    const int nx = (x1 + w) / ds + 2 - X0, ny = (y1 + h) / ds + 2 - Y0;
    const double ifx = cv_inv_scale(w, net_w), ify = cv_inv_scale(h, net_h);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < nx * ny; idx += gridDim.x * 256) {
        const int X = X0 + idx % nx, Y = Y0 + idx / nx;
        if (X < 0 || Y < 0 || X >= p.map_w || Y >= p.map_h)
            continue;
        const int px = p.sx_lut ? p.sx_lut[X] : X * ds, py = p.sy_lut ? p.sy_lut[Y] : Y * ds;
        if (px < 0 || py < 0)
            continue;
        const int cx = px - x1, cy = py - y1;
        if (cx < 0 || cy < 0 || cx >= w || cy >= h)
            continue;
        const unsigned v = src[(long long)nearest_src(cy, ify, net_h) * net_w + nearest_src(cx, ifx, net_w)];
        byte_max(p.slide_map + (long long)Y * p.map_w + X, v);
    }
}

// ---------------------------------------------------------------------------------------------
// staging state, owned by the (first) model handle
struct CropPipe {
    int device = 0;
    struct LaneScratch {
        float *f32 = nullptr;           // the network input [n,3,net_h,net_w]
        size_t f32_bytes = 0;
        unsigned char *net = nullptr;   // network-resolution masks when the caller keeps none
        size_t net_bytes = 0;
        float *prob = nullptr;          // ensemble accumulator [n,classes,net_h,net_w]
        size_t prob_bytes = 0;
    } lane[4];
    // host pipeline
    struct Slot {
        unsigned char *hin = nullptr, *din = nullptr, *hout = nullptr, *dout = nullptr, *hnet = nullptr, *dnet = nullptr;
        unsigned char *hov = nullptr, *dov = nullptr;   // overlays (allocated on first use, sized like the packed input)
        bool ov_direct = false;
        unsigned long long *hh = nullptr, *dh = nullptr;
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        int first = -1, count = 0;
        bool out_direct = false;
        std::vector<gs_crop_desc> descs;
    } sl[4];
    size_t cap_in = 0, cap_out = 0, cap_net = 0, cap_ov = 0;
    int cap_batch = 0;
    hipStream_t h2d = nullptr, compute[2] = {nullptr, nullptr};
};

static void free_slots(CropPipe &p)
{
    for (auto &s : p.sl) {
        if (s.hin) hipHostFree(s.hin);
        if (s.hout) hipHostFree(s.hout);
        if (s.hnet) hipHostFree(s.hnet);
        if (s.hh) hipHostFree(s.hh);
        if (s.din) hipFree(s.din);
        if (s.dout) hipFree(s.dout);
        if (s.dnet) hipFree(s.dnet);
        if (s.dh) hipFree(s.dh);
        if (s.hov) hipHostFree(s.hov);
        if (s.dov) hipFree(s.dov);
        if (s.up) hipEventDestroy(s.up);
        if (s.done) hipEventDestroy(s.done);
        if (s.down) hipEventDestroy(s.down);
        s = CropPipe::Slot();
    }
    p.cap_in = p.cap_out = p.cap_net = p.cap_ov = 0;
    p.cap_batch = 0;
}

void crop_pipe_destroy(CropPipe *p)
{
    if (!p)
        return;
    free_slots(*p);
    for (auto &l : p->lane) {
        if (l.f32) hipFree(l.f32);
        if (l.net) hipFree(l.net);
        if (l.prob) hipFree(l.prob);
    }
    if (p->h2d) hipStreamDestroy(p->h2d);
    for (auto &c : p->compute)
        if (c) hipStreamDestroy(c);
    delete p;
}

static CropPipe *pipe_of(gs_espnet *h)
{
    CropPipe *&p = espnet_crop_pipe(h);
    if (!p) {
        p = new (std::nothrow) CropPipe();
        if (p)
            p->device = espnet_device(h);
    }
    return p;
}

template <typename T>
static gs_status grow(T *&buf, size_t &have, size_t need, const char *what)
{
    if (have >= need)
        return GS_OK;
    if (buf) {
        GS_HIP(hipDeviceSynchronize());   // work in flight may still use the old buffer
        GS_HIP(hipFree(buf));
        buf = nullptr;
        have = 0;
    }
    if (hipMalloc(reinterpret_cast<void **>(&buf), need) != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s allocation of %zu bytes failed", what, need);
        return GS_ERR_NOMEM;
    }
    have = need;
    return GS_OK;
}

static gs_status check_common(gs_espnet *const *models, int n_models, const float *means, const float *stds, int net_h, int net_w)
{
    GS_REQUIRE(models && n_models > 0 && means && stds, "segment_crops: null argument");
    GS_REQUIRE(net_h >= 8 && net_w >= 8 && net_h % 8 == 0 && net_w % 8 == 0,
               "network size must be a positive multiple of 8 in both dimensions (got %dx%d)", net_h, net_w);
    for (int k = 0; k < n_models; ++k) {
        GS_REQUIRE(models[k] && espnet_is_full_net(models[k]), "model %d is not a full ESPNet handle (the crop entries need the decoder)", k);
        GS_REQUIRE(espnet_classes(models[k]) == espnet_classes(models[0]), "model %d has %d classes, model 0 has %d", k,
                   espnet_classes(models[k]), espnet_classes(models[0]));
        for (int i = 0; i < 3; ++i)
            GS_REQUIRE(stds[3 * k + i] != 0.0f, "model %d: std[%d] is zero", k, i);
    }
    return GS_OK;
}

// One batch, everything on stream s: the table is complete (offsets within packed_in / packed_out).
static gs_status run_batch(gs_espnet *const *models, int n_models, int lane, const unsigned char *packed_in, const gs_crop_desc *descs,
                           int n, const float *means, const float *stds, int net_h, int net_w, unsigned char *net_masks,
                           unsigned char *packed_out, unsigned long long *hist, const gs_paste_target *paste, hipStream_t s,
                           const gs_crop_overlay *overlay = nullptr, unsigned char *overlay_out = nullptr)
{
    CropPipe *pipe = pipe_of(models[0]);
    GS_REQUIRE(pipe, "out of host memory");
    GS_REQUIRE(n >= 1 && n <= MAXC, "internal: a batch of %d crops does not fit the %d-entry descriptor table", n, MAXC);
    GS_REQUIRE(lane >= 0 && lane < 4, "lane %d out of range", lane);
    for (int k = 0; k < n_models; ++k)
        GS_REQUIRE(lane < espnet_lanes(models[k]), "model %d has no lane %d (gs_espnet_set_lanes)", k, lane);
    CropPipe::LaneScratch &ls = pipe->lane[lane];
    const size_t npx = (size_t)net_h * net_w;
    const int classes = espnet_classes(models[0]);
    gs_status st = grow(ls.f32, ls.f32_bytes, (size_t)n * 3 * npx * sizeof(float), "crop tensor");
    if (st != GS_OK) return st;
    if (!net_masks) {
        st = grow(ls.net, ls.net_bytes, (size_t)n * npx, "network-resolution mask");
        if (st != GS_OK) return st;
        net_masks = ls.net;
    }
    if (n_models > 1) {
        st = grow(ls.prob, ls.prob_bytes, (size_t)n * classes * npx * sizeof(float), "ensemble accumulator");
        if (st != GS_OK) return st;
    }
    CropTable tab;
    std::memset(&tab, 0, sizeof tab);
    long long max_hw = 1, max_cells = 1;
    for (int i = 0; i < n; ++i) {
        tab.d[i] = descs[i];
        max_hw = std::max(max_hw, (long long)descs[i].h * descs[i].w);
        if (paste)
            max_cells = std::max(max_cells, (long long)(descs[i].w / paste->ds + 3) * (descs[i].h / paste->ds + 3));
    }
    const dim3 prep_grid((unsigned)((net_h * (net_w / 4) + 255) / 256), (unsigned)n);
    for (int k = 0; k < n_models; ++k) {
        PrepArgs a{};
        a.in = packed_in;
        a.out = ls.f32;
        a.net_h = net_h;
        a.net_w = net_w;
        for (int c = 0; c < 3; ++c) {
            a.mean[c] = means[3 * k + c];
            a.std[c] = stds[3 * k + c];
        }
        if (hist && k == 0) {
            a.hist_zero = hist;
            a.hist_count = n * classes;
        }
        hipLaunchKernelGGL(crops_prep_kernel, prep_grid, dim3(256), 0, s, tab, a);
        GS_HIP(hipGetLastError());
        const int mode = n_models == 1 ? 0 : k == 0 ? 1 : k == n_models - 1 ? 3 : 2;
        st = espnet_forward_ex(models[k], lane, ls.f32, GS_IN_F32_NCHW, n, net_h, net_w, nullptr, nullptr, nullptr, net_masks, nullptr,
                               n_models > 1 ? ls.prob : nullptr, mode, 1.0f / (float)n_models, s);
        if (st != GS_OK) return st;
    }
    if (packed_out || hist) {
        long long gx = (max_hw + 8 * 1024 - 1) / (8 * 1024);   // about eight iterations per workgroup, never more than 512
        gx = std::max(gx, (max_hw + 512 * 1024 - 1) / (512 * 1024));
        gx = std::min(std::max(gx, 1ll), 65535ll);
        const dim3 grid((unsigned)gx, (unsigned)n);
        switch ((classes + 4) / 5) {
        case 1: hipLaunchKernelGGL(crops_back_kernel<1>, grid, dim3(256), 0, s, tab, net_masks, net_h, net_w, packed_out, hist, classes); break;
        case 2: hipLaunchKernelGGL(crops_back_kernel<2>, grid, dim3(256), 0, s, tab, net_masks, net_h, net_w, packed_out, hist, classes); break;
        case 3: hipLaunchKernelGGL(crops_back_kernel<3>, grid, dim3(256), 0, s, tab, net_masks, net_h, net_w, packed_out, hist, classes); break;
        default: hipLaunchKernelGGL(crops_back_kernel<4>, grid, dim3(256), 0, s, tab, net_masks, net_h, net_w, packed_out, hist, classes); break;
        }
        GS_HIP(hipGetLastError());
    }
    if (overlay && overlay_out) {   // needs the crop-size maps: the caller passes packed_out with it
        OverlayArgs oa{};
        oa.crops = packed_in;
        oa.maps = packed_out;
        oa.out = overlay_out;
        oa.wa = overlay->wa;
        oa.wb = overlay->wb;
        oa.n_colours = overlay->n_colours;
        std::memcpy(oa.pal, overlay->palette_rgb, (size_t)overlay->n_colours * 3);
        const long long gx = std::min(std::max((max_hw + 8 * 1024 - 1) / (8 * 1024), 1ll), 65535ll);
        hipLaunchKernelGGL(crops_overlay_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, tab, oa);
        GS_HIP(hipGetLastError());
    }
    if (paste) {
        const unsigned gx = (unsigned)std::min<long long>((max_cells + 255) / 256, 4096);
        hipLaunchKernelGGL(crops_paste_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, tab, net_masks, net_h, net_w, *paste);
        GS_HIP(hipGetLastError());
    }
    return GS_OK;
}

static gs_status check_descs(const gs_crop_desc *descs, int n, bool need_out)
{
    GS_REQUIRE(descs && n > 0 && n <= MAXC, "1 to %d crops per call (got %d)", MAXC, n);
    for (int i = 0; i < n; ++i) {
        GS_REQUIRE(descs[i].h > 0 && descs[i].w > 0 && (long long)descs[i].h * descs[i].w < (1ll << 31), "crop %d has a bad size %dx%d", i,
                   descs[i].h, descs[i].w);
        GS_REQUIRE(descs[i].in_off >= 0 && (!need_out || (descs[i].out_off >= 0 && descs[i].out_off % 4 == 0)),
                   "crop %d: offsets must be non-negative and out_off a multiple of 4", i);
    }
    return GS_OK;
}

static gs_status check_paste(const gs_paste_target *p)
{
    if (!p)
        return GS_OK;
    GS_REQUIRE(p->slide_map && p->map_h > 0 && p->map_w > 0 && p->ds > 0, "paste target: null map or bad size");
    GS_REQUIRE((p->sx_lut == nullptr) == (p->sy_lut == nullptr), "paste target: give both tables or neither");
    // crops of one launch (or of the other stream) that overlap meet through a 32-bit compare-and-swap on the aligned word that
    // holds the byte: the map must start on a 4-byte boundary and its allocation must cover whole words
    GS_REQUIRE((reinterpret_cast<uintptr_t>(p->slide_map) & 3u) == 0,
               "paste target: slide_map must be 4-byte aligned (and its allocation padded to a multiple of 4 bytes)");
    return GS_OK;
}

}  // namespace gs

using namespace gs;

extern "C" {

gs_status gs_espnet_ensemble_segment_crops(gs_espnet *const *models, int n_models, const uint8_t *packed_in, const gs_crop_desc *descs,
                                           int n, const float *means, const float *stds, int net_h, int net_w, uint8_t *net_masks,
                                           uint8_t *packed_out, unsigned long long *hist, const gs_paste_target *paste, void *hip_stream)
{
    gs_status st = check_common(models, n_models, means, stds, net_h, net_w);
    if (st != GS_OK) return st;
    GS_REQUIRE(packed_in, "segment_crops: null input");
    GS_REQUIRE(net_masks || packed_out || hist || paste, "nothing to compute: every output is NULL");
    st = check_descs(descs, n, packed_out != nullptr);
    if (st != GS_OK) return st;
    st = check_paste(paste);
    if (st != GS_OK) return st;
    return run_batch(models, n_models, 0, packed_in, descs, n, means, stds, net_h, net_w, net_masks, packed_out, hist, paste,
                     static_cast<hipStream_t>(hip_stream));
}

gs_status gs_espnet_segment_crops(gs_espnet *h, int lane, const uint8_t *packed_in, const gs_crop_desc *descs, int n, const float mean[3],
                                  const float std[3], int net_h, int net_w, uint8_t *net_masks, uint8_t *packed_out,
                                  unsigned long long *hist, const gs_paste_target *paste, void *hip_stream)
{
    gs_espnet *models[1] = {h};
    gs_status st = check_common(models, h ? 1 : 0, mean, std, net_h, net_w);
    if (st != GS_OK) return st;
    GS_REQUIRE(packed_in, "segment_crops: null input");
    GS_REQUIRE(net_masks || packed_out || hist || paste, "nothing to compute: every output is NULL");
    st = check_descs(descs, n, packed_out != nullptr);
    if (st != GS_OK) return st;
    st = check_paste(paste);
    if (st != GS_OK) return st;
    return run_batch(models, 1, lane, packed_in, descs, n, mean, std, net_h, net_w, net_masks, packed_out, hist, paste,
                     static_cast<hipStream_t>(hip_stream));
}

gs_status gs_plan_crop_batches(const int *heights, const int *widths, int n_crops, int batch, int *starts, int cap, int *n_batches)
{
    GS_REQUIRE(heights && widths && n_batches, "gs_plan_crop_batches: null argument");
    GS_REQUIRE(n_crops > 0 && batch > 0, "gs_plan_crop_batches: n_crops and batch must be positive");
    for (int i = 0; i < n_crops; ++i)
        GS_REQUIRE(heights[i] > 0 && widths[i] > 0 && (long long)heights[i] * widths[i] < (1ll << 29), "crop %d has a bad size %dx%d", i,
                   heights[i], widths[i]);
    const CropBatchPlan plan = plan_crop_batches(heights, widths, n_crops, batch, MAXC);
    *n_batches = (int)plan.starts.size() - 1;
    if (!starts)
        return GS_OK;
    GS_REQUIRE(cap >= (int)plan.starts.size(), "gs_plan_crop_batches: %d entries needed, room for %d", (int)plan.starts.size(), cap);
    std::copy(plan.starts.begin(), plan.starts.end(), starts);
    return GS_OK;
}

int gs_host_block_is_pinned(const void *p, size_t bytes) { return p && host_block_is_pinned(p, bytes) ? 1 : 0; }

gs_status gs_espnet_segment_crops_host(gs_espnet *const *models, int n_models, const uint8_t *const *crops, const int *heights,
                                       const int *widths, int n_crops, const float *means, const float *stds, int net_h, int net_w,
                                       int batch, uint8_t *const *masks, uint8_t *net_masks, unsigned long long *hist,
                                       const gs_paste_target *paste, const int *x1, const int *y1, const gs_crop_overlay *overlay)
{
    gs_status st = check_common(models, n_models, means, stds, net_h, net_w);
    if (st != GS_OK) return st;
    GS_REQUIRE(crops && heights && widths && n_crops > 0, "segment_crops_host: null crop list");
    GS_REQUIRE(batch > 0, "batch must be positive");
    GS_REQUIRE(masks || net_masks || hist || paste || overlay, "nothing to compute: every output is NULL");
    GS_REQUIRE(!paste || (x1 && y1), "a paste target needs the crops' level-0 origins");
    if (overlay) {
        GS_REQUIRE(overlay->palette_rgb && overlay->out_bgr && overlay->n_colours >= 1 && overlay->n_colours <= GS_MAX_PALETTE,
                   "overlay: null palette / outputs or a table of %d colours (1 .. %d)", overlay->n_colours, GS_MAX_PALETTE);
        for (int i = 0; i < n_crops; ++i)
            GS_REQUIRE(overlay->out_bgr[i], "overlay: crop %d has no output buffer", i);
    }
    st = check_paste(paste);
    if (st != GS_OK) return st;
    for (int i = 0; i < n_crops; ++i) {
        GS_REQUIRE(crops[i] && (!masks || masks[i]), "crop %d: null pointer", i);
        GS_REQUIRE(heights[i] > 0 && widths[i] > 0 && (long long)heights[i] * widths[i] < (1ll << 29), "crop %d has a bad size %dx%d", i,
                   heights[i], widths[i]);
    }
    CropPipe *pp = pipe_of(models[0]);
    GS_REQUIRE(pp, "out of host memory");
    CropPipe &p = *pp;
    int nl = 2;   // batches alternate between two lanes when every member has them
    for (int k = 0; k < n_models; ++k)
        if (espnet_lanes(models[k]) < 2) nl = 1;
    const size_t npx = (size_t)net_h * net_w;
    const size_t ncl = (size_t)espnet_classes(models[0]);   // hist is [n_crops][classes]
    // which crops go into which batch, and the staging a batch needs (csrc/crop_plan.h: host-only, sanitised on its own)
    const CropBatchPlan plan = plan_crop_batches(heights, widths, n_crops, batch, MAXC);
    const std::vector<int> &starts = plan.starts;
    const size_t need_in = plan.need_in, need_out = plan.need_out;
    batch = plan.max_count;
    GS_REQUIRE(batch >= 1 && batch <= MAXC, "internal: planned a batch of %d crops", batch);
    constexpr int NSLOT = 4;
    gs_status rc = GS_OK;
    auto fail = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == GS_OK) {
            set_error("%s failed: %s", what, hipGetErrorString(e));
            rc = GS_ERR_HIP;
        }
        return e != hipSuccess;
    };
    // three streams at three priorities, as in gs_espnet_segment_host (espnet.hip): HIP keeps a pool of hardware queues per
    // priority, so the upload stream and the two compute streams never share a queue whatever else the process has made
    if (!p.h2d || !p.compute[0] || !p.compute[1]) {   // all three or none: a partial set would run later calls on the NULL stream
        int lo = 0, hi = 0;
        fail(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange");
        hipStream_t *want[3] = {&p.h2d, &p.compute[0], &p.compute[1]};
        const int prio[3] = {hi, (lo + hi) / 2, lo};
        for (int k = 0; k < 3 && rc == GS_OK; ++k)
            if (!*want[k])
                fail(hipStreamCreateWithPriority(want[k], hipStreamNonBlocking, prio[k]), "hipStreamCreate");
        if (rc != GS_OK) {
            for (hipStream_t *w : want) {
                if (*w) hipStreamDestroy(*w);
                *w = nullptr;
            }
            return rc;
        }
    }
    if (p.cap_in < need_in || p.cap_out < need_out || p.cap_net < npx * batch || p.cap_batch < batch) {
        fail(hipDeviceSynchronize(), "hipDeviceSynchronize");
        const size_t ci = std::max(p.cap_in, need_in), co = std::max(p.cap_out, need_out), cn = std::max(p.cap_net, npx * batch);
        const int cb = std::max(p.cap_batch, batch);
        free_slots(p);
        for (int i = 0; i < NSLOT && rc == GS_OK; ++i) {
            CropPipe::Slot &s = p.sl[i];
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hin), ci, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hout), co, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hnet), cn, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hh), sizeof(unsigned long long) * GS_MAX_CLASSES * cb, hipHostMallocDefault), "hipHostMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.din), ci), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dout), co), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dnet), cn), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dh), sizeof(unsigned long long) * GS_MAX_CLASSES * cb), "hipMalloc");
            fail(hipEventCreateWithFlags(&s.up, hipEventDisableTiming), "hipEventCreate");
            fail(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
            fail(hipEventCreateWithFlags(&s.down, hipEventDisableTiming), "hipEventCreate");
        }
        if (rc != GS_OK) {
            free_slots(p);
            return rc;
        }
        p.cap_in = ci;
        p.cap_out = co;
        p.cap_net = cn;
        p.cap_batch = cb;
    }
    if (overlay && p.cap_ov < p.cap_in) {   // overlay staging: as large as the packed input, made when first asked for
        fail(hipDeviceSynchronize(), "hipDeviceSynchronize");
        for (int i = 0; i < NSLOT && rc == GS_OK; ++i) {
            CropPipe::Slot &s = p.sl[i];
            if (s.hov) hipHostFree(s.hov);
            if (s.dov) hipFree(s.dov);
            s.hov = s.dov = nullptr;
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hov), p.cap_in, hipHostMallocDefault), "hipHostMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dov), p.cap_in), "hipMalloc");
        }
        if (rc != GS_OK) {
            free_slots(p);
            return rc;
        }
        p.cap_ov = p.cap_in;
    }
    for (auto &s : p.sl)
        s.first = -1;
    const bool net_pinned = net_masks && host_is_pinned(net_masks), hist_pinned = hist && host_is_pinned(hist);
    auto drain = [&](CropPipe::Slot &s) {
        if (s.first < 0 || rc != GS_OK)
            return;
        if (fail(hipEventSynchronize(s.down), "hipEventSynchronize")) return;
        if (masks && !s.out_direct)
            parallel_jobs(s.count, 4, [&](int j) {
                std::memcpy(masks[s.first + j], s.hout + s.descs[j].out_off, (size_t)s.descs[j].h * s.descs[j].w);
            });
        if (overlay && !s.ov_direct)
            parallel_jobs(s.count, 4, [&](int j) {
                std::memcpy(overlay->out_bgr[s.first + j], s.hov + s.descs[j].in_off, (size_t)s.descs[j].h * s.descs[j].w * 3);
            });
        if (net_masks && !net_pinned)
            parallel_memcpy(net_masks + (size_t)s.first * npx, s.hnet, npx * s.count);
        if (hist && !hist_pinned)
            std::memcpy(hist + (size_t)s.first * ncl, s.hh, sizeof(unsigned long long) * ncl * s.count);
        s.first = -1;
    };
    int slot = 0, bi = 0;
    for (; bi + 1 < (int)starts.size() && rc == GS_OK; slot = (slot + 1) % NSLOT, ++bi) {
        const int first = starts[bi];
        CropPipe::Slot &s = p.sl[slot];
        hipStream_t compute = p.compute[bi & 1];
        drain(s);   // the slot's previous batch must have left its buffers
        if (rc != GS_OK) break;
        const int cnt = starts[bi + 1] - first;
        s.descs.assign(cnt, gs_crop_desc{});
        size_t oi = 0, oo = 0;
        fill_crop_descs(heights, widths, x1, y1, first, cnt, s.descs.data(), &oi, &oo);
        bool in_direct = true;
        s.out_direct = masks != nullptr;
        s.ov_direct = overlay != nullptr;
        for (int j = 0; j < cnt; ++j) {
            in_direct = in_direct && host_is_pinned(crops[first + j]);
            if (masks)
                s.out_direct = s.out_direct && host_is_pinned(masks[first + j]);
            if (overlay)
                s.ov_direct = s.ov_direct && host_is_pinned(overlay->out_bgr[first + j]);
        }
        // uploads: page-locked crops are DMA'd in place, pageable ones are packed into the slot's pinned buffer by a few
        // threads (one core copies ~10 GB/s) and leave as one copy
        if (in_direct) {
            for (int j = 0; j < cnt && rc == GS_OK; ++j)
                fail(hipMemcpyAsync(s.din + s.descs[j].in_off, crops[first + j], (size_t)s.descs[j].h * s.descs[j].w * 3, hipMemcpyHostToDevice,
                                    p.h2d), "H2D copy");
        } else {
            parallel_jobs(cnt, bi == 0 ? 8 : 4, [&](int j) {   // (the first batch's staging is exposed: more threads)
                std::memcpy(s.hin + s.descs[j].in_off, crops[first + j], (size_t)s.descs[j].h * s.descs[j].w * 3);
            });
            fail(hipMemcpyAsync(s.din, s.hin, oi, hipMemcpyHostToDevice, p.h2d), "H2D copy");
        }
        if (rc != GS_OK) break;
        fail(hipEventRecord(s.up, p.h2d), "hipEventRecord");
        fail(hipStreamWaitEvent(compute, s.up, 0), "hipStreamWaitEvent");
        if (nl == 1 && bi > 0)   // one workspace: this batch after the previous one (on the other stream)
            fail(hipStreamWaitEvent(compute, p.sl[(slot + NSLOT - 1) % NSLOT].done, 0), "hipStreamWaitEvent");
        gs_status st2 = run_batch(models, n_models, bi % nl, s.din, s.descs.data(), cnt, means, stds, net_h, net_w, s.dnet,
                                  (masks || overlay) ? s.dout : nullptr, hist ? s.dh : nullptr, paste, compute, overlay, overlay ? s.dov : nullptr);
        if (st2 != GS_OK) { rc = st2; break; }
        fail(hipEventRecord(s.done, compute), "hipEventRecord");
        // downloads through hipMemcpy2DAsync: the SDMA engine, not a blit kernel that would take CUs from the next forward
        // (gs_espnet_segment_host)
        if (masks) {
            if (s.out_direct) {
                // page-locked destinations laid out like the packed device buffer (every map at its 256-byte-aligned offset
                // behind the batch's first one: what engine.segment_crops_host allocates) leave as ONE copy; 32 separate DMA
                // commands per batch sat in the compute stream between two forwards
                bool packed = true;
                for (int j = 0; j < cnt; ++j)
                    packed = packed && masks[first + j] == masks[first] + s.descs[j].out_off;
                const size_t b = (size_t)s.descs[cnt - 1].out_off + (size_t)s.descs[cnt - 1].h * s.descs[cnt - 1].w;
                // ... and only when the whole range is ONE page-locked allocation: separately pinned buffers that happen to be
                // neighbours in virtual memory are written map by map
                packed = packed && host_block_is_pinned(masks[first], b);
                if (packed) {
                    fail(hipMemcpy2DAsync(masks[first], b, s.dout, b, b, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
                } else {
                    for (int j = 0; j < cnt && rc == GS_OK; ++j) {
                        const size_t b = (size_t)s.descs[j].h * s.descs[j].w;
                        fail(hipMemcpy2DAsync(masks[first + j], b, s.dout + s.descs[j].out_off, b, b, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
                    }
                }
            } else {
                fail(hipMemcpy2DAsync(s.hout, oo, s.dout, oo, oo, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
            }
        }
        if (overlay) {
            if (s.ov_direct) {   // (as the maps: one DMA for a batch laid out like the packed buffer inside ONE pinned allocation)
                bool packed = true;
                for (int j = 0; j < cnt; ++j)
                    packed = packed && overlay->out_bgr[first + j] == overlay->out_bgr[first] + s.descs[j].in_off;
                const size_t b = (size_t)s.descs[cnt - 1].in_off + (size_t)s.descs[cnt - 1].h * s.descs[cnt - 1].w * 3;
                packed = packed && host_block_is_pinned(overlay->out_bgr[first], b);
                if (packed) {
                    fail(hipMemcpy2DAsync(overlay->out_bgr[first], b, s.dov, b, b, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
                } else {
                    for (int j = 0; j < cnt && rc == GS_OK; ++j) {
                        const size_t bj = (size_t)s.descs[j].h * s.descs[j].w * 3;
                        fail(hipMemcpy2DAsync(overlay->out_bgr[first + j], bj, s.dov + s.descs[j].in_off, bj, bj, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
                    }
                }
            } else {
                fail(hipMemcpy2DAsync(s.hov, oi, s.dov, oi, oi, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
            }
        }
        if (net_masks)
            fail(hipMemcpy2DAsync(net_pinned ? net_masks + (size_t)first * npx : s.hnet, npx, s.dnet, npx, npx, cnt, hipMemcpyDeviceToHost, compute),
                 "D2H copy");
        if (hist) {
            const size_t b = sizeof(unsigned long long) * ncl * cnt;
            fail(hipMemcpy2DAsync(hist_pinned ? hist + (size_t)first * ncl : s.hh, b, s.dh, b, b, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
        }
        fail(hipEventRecord(s.down, compute), "hipEventRecord");
        s.first = first;
        s.count = cnt;
    }
    for (int k = 0; k < NSLOT; ++k)
        drain(p.sl[(slot + k) % NSLOT]);   // oldest first
    if (rc != GS_OK) {
        hipDeviceSynchronize();
        return rc;
    }
    return gs_device_fault_check();
}

}  // extern "C"
