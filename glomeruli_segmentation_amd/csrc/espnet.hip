// gs_espnet_*: model handle, weight packing, HBM workspace and the forward schedule.
// Reference path replaced: module/espnet/test/Model.py ESPNet.forward (:341-378) /
// ESPNet_Encoder.forward (:273-304) called from module/espnet/test/VisualizeResults_iou.py:123.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "conv_mfma.h"
#include "dec_tail_args.h"
#include "espnet_config.h"
#include "espnet_kernels.h"
#include "host_copy.h"

namespace gs {

static thread_local std::string g_err;
void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

#ifdef GS_DIAG
static bool getenv_flag(const char *name)
{
    const char *e = std::getenv(name);
    return e && std::atoi(e) != 0;
}
static bool no_vec() { return getenv_flag("GS_NO_VEC"); }
#else
static constexpr bool no_vec() { return false; }
#endif

// Every unit-stride conv launch exists in two pixel mappings; the vector one (F_VEC) needs the output width to be a
// multiple of P (the 9th configuration parameter).
template <int FLAGS, int... C>
static gs_status launch_vec(const ConvArgs &ca, int num_cus, hipStream_t s)
{
    constexpr int cfg[] = {C...};
    if (ca.W % cfg[8] == 0 && !no_vec())
        return launch_conv_mfma<C..., FLAGS | F_VEC>(ca, num_cus, s);
    return launch_conv_mfma<C..., FLAGS>(ca, num_cus, s);
}
enum KernelId {
    K_STEM, K_POOL, K_L2_C1S, K_L2_DOWN, K_L2_C1, K_L2_ESP, K_CAT_B2, K_L3_C1S, K_L3_DOWN, K_L3_C1, K_L3_ESP,
    K_DEC1, K_DEC2, K_DEC3, K_DEC_CONV, K_DEC4, K_DEC_TAIL, K_COUNT
};
static const char *kKernelNames[K_COUNT] = {
    "stem_kernel", "pool_kernel", "conv_l2_reduce_s2", "conv_l2_down_branches", "conv_l2_reduce_1x1",
    "conv_l2_esp_branches", "cat_b2_kernel", "conv_l3_reduce_s2", "conv_l3_down_branches", "conv_l3_reduce_1x1",
    "conv_l3_esp_branches", "dec1_kernel", "dec2_kernel", "dec3_kernel", "conv_dec_cbr", "dec4_kernel", "dec_tail_kernel"};

struct PackedConv {   // float offsets into the device weight blob
    long long c1 = -1, br = -1;
    bool fused_next = false;   // br carries the F_FUSE1X1 table of the following block's c1
};

struct Model {
    int classes = 0, p = 0, q = 0;
    int cp = 0;   // the padded class count the decoder kernels are instantiated for: 5 for the five-class networks (the fast path),
                  // else `classes` rounded up to a multiple of four (espnet_kernels.h, "CLASS COUNTS")
    bool encoder_only = false;
    int device = 0, num_cus = 256;
#ifdef GS_DIAG
    int variant = 0;   // GS_VARIANT env (diagnostic builds only): kernel A/B experiments (0 = shipped configuration)
#else
    static constexpr int variant = 0;
#endif
    float *dblob = nullptr;
    // offsets (floats) into dblob
    float stem_params[537] = {0};   // host copy of level1 weights + folded bn1 + folded b1: they travel as kernel arguments
    long long w1, bn1, b1, b2, b3, wcls, br, wup3, w3c, cbr0, wcc, bncc, wup2, bnu2, wconv, wconv_xm, wclassifier, wtail;
    long long wcc_mfma = -1;   // twelve classes and more: combine_l2_l3.1 as a conv_mfma image (see forward_impl)
    PackedConv l2_0;
    std::vector<PackedConv> l2, l3;
    PackedConv l3_0;

    // workspace
    void *ws = nullptr;
    size_t ws_bytes = 0;
    int ws_n = 0, ws_h = 0, ws_w = 0;
    Act a0c, a0, inp1, inp2, r2[2], bb[3], a1, r3[2], cc[3], o2c, tt, t3, ee, ff;
    float *prob = nullptr;   // ensemble scratch
    size_t prob_bytes = 0;
    std::map<std::string, std::pair<Act, int>> stages;   // name -> (activation, channels) of the last forward
    int last_n = 0;
    bool b2_lazy = false;    // planes 64..127 of the stage "b2" are not materialised by the last forward (read_stage fills them)

    // host pipeline (gs_espnet_segment_host): two slots of pinned + device staging, kept across calls
    struct Slot {
        uint8_t *hin = nullptr, *hout = nullptr, *din = nullptr, *dout = nullptr;
        unsigned long long *hh = nullptr, *dh = nullptr;
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        int first = -1, count = 0;
    } sl[4];   // four slots, two per stream: the host runs up to four batches ahead of the GPU
    hipStream_t pipe_compute = nullptr, pipe_h2d = nullptr;   // the first of the pipeline's two compute streams (even batches) and its upload stream; see gs_espnet_segment_host
    size_t pipe_in_bytes = 0, pipe_out_bytes = 0;
    int pipe_batch = 0;

    // profiling
    bool profile = false;
    struct Ev { hipEvent_t a, b; int k; };
    std::vector<Ev> events;
    double prof_ms[K_COUNT] = {0};
    long long prof_launches[K_COUNT] = {0};
    double prof_flops[K_COUNT] = {0};
};

// ------------------------------------------------------------------------------------------
struct WeightTable {
    const float *blob;
    std::map<std::string, const gs_layer_desc *> by_name;
    std::string prefix;   // "encoder." for the full net, "" for ESPNet_Encoder tables
    bool ok = true;
    const float *get(const std::string &name, std::initializer_list<int> shape)
    {
        auto it = by_name.find(name);
        if (it == by_name.end()) {
            set_error("weight tensor '%s' missing from the table", name.c_str());
            ok = false;
            return nullptr;
        }
        const gs_layer_desc *d = it->second;
        int i = 0;
        bool match = d->ndim == (int)shape.size();
        for (int s : shape)
            match = match && d->shape[i++] == s;
        if (!match) {
            set_error("weight tensor '%s' has the wrong shape", name.c_str());
            ok = false;
            return nullptr;
        }
        return blob + d->offset;
    }
};

struct BlobBuilder {
    std::vector<float> data;
    long long reserve(size_t n)
    {
        const size_t at = (data.size() + 3) / 4 * 4;   // 16-byte aligned pieces (float4 LDS staging)
        data.resize(at + (n + 3) / 4 * 4, 0.0f);
        return (long long)at;
    }
    long long push(const float *src, size_t n)
    {
        const long long at = reserve(n);
        std::memcpy(data.data() + at, src, n * sizeof(float));
        return at;
    }
};

// BatchNorm2d(eps=1e-3).eval() folded to y = x*scale + shift, plus the PReLU slope (1 when absent):
// layout [scale | shift | alpha][C].  reference: Model.py:21-22,44-45,141-142
static bool fold_bn(WeightTable &t, const std::string &bn, const std::string &act, int C, float *dst, bool with_alpha = true)
{
    const float *g = t.get(bn + ".weight", {C}), *b = t.get(bn + ".bias", {C});
    const float *m = t.get(bn + ".running_mean", {C}), *v = t.get(bn + ".running_var", {C});
    const float *al = act.empty() ? nullptr : t.get(act + ".weight", {C});
    if (!t.ok)
        return false;
    for (int c = 0; c < C; ++c) {
        const double inv = 1.0 / std::sqrt((double)v[c] + 1e-3);
        dst[c] = (float)((double)g[c] * inv);
        dst[C + c] = (float)((double)b[c] - (double)m[c] * (double)g[c] * inv);
        if (with_alpha)
            dst[2 * C + c] = al ? al[c] : 1.0f;
    }
    return true;
}

// conv weight [cout][cin][k][k] -> LDS image rows [tap][cin_padded][nrow] of dilation slot `slot`
static void pack_conv(const float *w, int cout, int cin, int k, float *dst, int slot, int taps, int cinp, int nrow)
{
    for (int tap = 0; tap < taps; ++tap)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co)
                dst[(((size_t)slot * taps + tap) * cinp + ci) * nrow + co] = w[((size_t)co * cin + ci) * k * k + tap];
}

// `next` names the block whose c1 (1x1 reduce of THIS block's output, Model.py:193) is computed in this block's epilogue
// (F_FUSE1X1); empty = no fusion.
static bool pack_block(WeightTable &t, BlobBuilder &bb, const std::string &pre, bool down, int level, PackedConv &pc,
                       const float *dual = nullptr, int dual_coff = 0, int dual_c = 0, const std::string &next = "",
                       const float *in2_bn = nullptr, int in2_c0 = 0, int in2_cn = 0, int in2_c = 0)
{
    // level 2: cin 19 (down) / 64, n = 12, n1 = 16;  level 3: cin 131 (down) / 128, n = 25, n1 = 28
    const int n = level == 2 ? 12 : 25, n1 = level == 2 ? 16 : 28, nOut = n1 + 4 * n;
    const int cin = level == 2 ? (down ? 19 : 64) : (down ? 131 : 128);
    const int kl = level == 2 ? 4 : 2;
    const int cinp = (cin + kl - 1) / kl * kl;
    const int taps = down ? 9 : 1;
    const float *wc1 = t.get(pre + ".c1.conv.weight", {n, cin, down ? 3 : 1, down ? 3 : 1});
    if (!t.ok)
        return false;
    const int c1_floats = conv_wfloats(cinp, taps, 1, n, n, false);
    const int bnl_c = cinp + kl;   // F_BNLOAD table: one entry per (padded) input channel + an all-zero slot of one k-group
    pc.c1 = bb.reserve(c1_floats + (in2_bn ? 3 * bnl_c : 0));
    pack_conv(wc1, n, cin, down ? 3 : 1, bb.data.data() + pc.c1, 0, taps, cinp, n);
    if (in2_bn) {   // [scale | shift | alpha][bnl_c]: identity, except the cat's BR for the channels that are stored raw
        float *x = bb.data.data() + pc.c1 + c1_floats;
        for (int c = 0; c < bnl_c; ++c) {
            const bool raw = c >= in2_c0 && c < in2_c0 + in2_cn, zero = c >= cinp;
            x[c] = zero ? 0.0f : raw ? in2_bn[c] : 1.0f;
            x[bnl_c + c] = zero ? 0.0f : raw ? in2_bn[in2_c + c] : 0.0f;
            x[2 * bnl_c + c] = raw ? in2_bn[2 * in2_c + c] : 1.0f;
        }
    }

    const int rcinp = (n + kl - 1) / kl * kl;
    const int mt = level == 2 ? 16 : 32, nacc = level == 2 ? 4 : 16;
    pc.fused_next = !next.empty();
    const ConvImage im = conv_image(rcinp, 9, 5, n1, n, true, dual != nullptr, false, pc.fused_next ? nacc : 0);
    pc.br = bb.reserve(im.total);
    static const char *dn[5] = {".d1", ".d2", ".d4", ".d8", ".d16"};
    for (int di = 0; di < 5; ++di) {
        const int co = di == 0 ? n1 : n;
        const float *w = t.get(pre + dn[di] + ".conv.weight", {co, n, 3, 3});
        if (!t.ok)
            return false;
        pack_conv(w, co, n, 3, bb.data.data() + pc.br, di, 9, rcinp, n1);
    }
    if (pc.fused_next) {
        // table[di][r][lane]: the A operand of the k-step "accumulator register r of slot di": lane = (k-group, c1 output
        // row i); k-group kq of register r holds this block's channel cb + row(r, kq)
        const float *w2 = t.get(next + ".c1.conv.weight", {n, nOut, 1, 1});
        if (!t.ok)
            return false;
        float *tab = bb.data.data() + pc.br + im.w + im.bn;
        for (int di = 0; di < 5; ++di) {
            const int nout = di == 0 ? n1 : n, cb = di == 0 ? 0 : n1 + (di - 1) * n;
            for (int r = 0; r < nacc; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane % mt, kq = lane / mt;
                    const int row = mt == 32 ? (r & 3) + 8 * (r >> 2) + 4 * kq : kq * 4 + r;
                    tab[(di * nacc + r) * 64 + lane] = (row < nout && i < n) ? w2[(size_t)i * nOut + cb + row] : 0.0f;
                }
        }
    }
    float *bnp = bb.data.data() + pc.br + im.w;
    // DownSamplerB: self.bn / self.act (Model.py:141-142); ESP block: self.bn = BR(nOut) (Model.py:184)
    if (dual)   // slice of the following concat's BR parameters, same [scale | shift | alpha][nOut] layout
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < nOut; ++c)
                bnp[(3 + j) * nOut + c] = dual[j * dual_c + dual_coff + c];
    return down ? fold_bn(t, pre + ".bn", pre + ".act", nOut, bnp) : fold_bn(t, pre + ".bn.bn", pre + ".bn.act", nOut, bnp);
}

// ------------------------------------------------------------------------------------------
static Act make_act(int C, int Cp, int H, int W, int pad_t, int pad_b, int pad_l, int pad_r)
{
    Act a;
    a.C = C;
    a.Cp = Cp;
    a.H = H;
    a.W = W;
    a.pitch = (int)round_up(pad_l + W + pad_r, 32);   // 128-byte rows: interior stores stay line-aligned
    // + 96 floats: a plane stride that is a power of two (8192 floats at 1/8 scale) lands every channel of a
    // pixel on the same HBM channel / L2 slice when a kernel walks the channels (dec1, the 1x1 reduces)
    a.sc = (pad_t + H + pad_b) * a.pitch + 96;
    a.off = pad_t * a.pitch + pad_l;
    a.sn = (long long)Cp * a.sc;
    return a;
}

static gs_status layout_workspace(Model *m, int n, int H, int W)
{
    if (m->ws && n <= m->ws_n && H == m->ws_h && W == m->ws_w)
        return GS_OK;
    if (m->ws) {
        GS_HIP(hipDeviceSynchronize());
        GS_HIP(hipFree(m->ws));
        m->ws = nullptr;
    }
    const int cls = m->cp;   // padded class planes (the planes beyond m->classes stay zero)
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8;
    // comb_l2_l3 (planes 0..cls-1, written by dec3) and output0_cat (planes cls..cls+18, written by the stem) share one
    // buffer in the order of the decoder's torch.cat (Model.py:375), so conv CBR(19+classes, classes, 3) reads its input as
    // ONE 24-plane activation with zero pad on all four sides.  Plane cls+19 is never written: the level-2 strided reduce
    // reads output0_cat padded to 20 channels (a multiple of the k-step) and its padding plane must be zeros, not a
    // plane some other stage writes (a non-finite value there would survive the zero weight).
    m->a0c = make_act(cls + 19, cls + 20, H1, W1, 1, 1, 32, 1);
    m->inp1 = make_act(3, 3, H1, W1, 0, 0, 0, 0);
    m->inp2 = make_act(3, 3, H2, W2, 0, 0, 0, 0);
    for (int i = 0; i < 2; ++i)   // two reduced maps: a block reads one while its epilogue writes the next block's
        m->r2[i] = make_act(12, 12, H2, W2, 16, 16, 32, 16);  // dilation up to 16
    for (int i = 0; i < 3; ++i)
        m->bb[i] = make_act(64, 64, H2, W2, 0, 0, 0, 0);
    m->a1 = make_act(131, 132, H2, W2, 1, 0, 32, 1);
    for (int i = 0; i < 2; ++i)
        m->r3[i] = make_act(25, 26, H3, W3, 16, 16, 32, 16);
    for (int i = 0; i < 3; ++i)
        m->cc[i] = make_act(128, 128, H3, W3, 0, 0, 0, 0);
    m->o2c = make_act(cls, cls, H2, W2, 0, 0, 0, 0);
    // (twelve classes and more: combine_l2_l3.1's 3x3 runs on the matrix cores and reads its input with a zero halo; t3 is its output)
    const bool dec3_mfma = cls >= 12;
    m->tt = dec3_mfma ? make_act(2 * cls, 2 * cls, H2, W2, 1, 1, 32, 1) : make_act(2 * cls, 2 * cls, H2, W2, 0, 0, 0, 0);
    m->t3 = dec3_mfma ? make_act(cls, cls, H2, W2, 0, 0, 0, 0) : make_act(1, 1, 8, 8, 0, 0, 0, 0);
    m->ff = make_act(cls, cls, H1, W1, 0, 0, 0, 0);
    // Lazy b2 (p > 0): output1_0 is stored RAW, once, straight into planes 64..127 of output1_cat -- bb[0] becomes a view of
    // them -- and the consumers of output1_cat apply b2 to those planes on load (CFG_LAZY_B2 above).
    const bool lazy_b2 = m->p > 0 && CFG_LAZY_B2;
    if (lazy_b2)
        m->bb[0] = Act();   // no storage of its own
    Act *all[] = {&m->a0c, &m->inp1, &m->inp2, &m->r2[0], &m->r2[1], &m->bb[0], &m->bb[1], &m->bb[2], &m->a1, &m->r3[0], &m->r3[1],
                  &m->cc[0], &m->cc[1], &m->cc[2], &m->o2c, &m->tt, &m->t3, &m->ff};
    for (Act *a : all) {   // kernels address one image with 32-bit byte offsets (buffer soffset / voffset)
        if ((unsigned long long)a->sn * sizeof(float) >= (1ull << 31)) {
            set_error("tile %dx%d is too large: an activation of one image exceeds 2 GiB", H, W);
            return GS_ERR_UNSUPPORTED;
        }
    }
    const size_t slack = 64 * 1024;   // strips may over-read past a buffer's last row (masked lanes only)
    size_t total = 0;
    for (Act *a : all)
        total += round_up(a->bytes(n) + slack, 256);
    void *ws = nullptr;
    if (hipMalloc(&ws, total) != hipSuccess) {
        set_error("workspace allocation of %zu bytes failed (n=%d, %dx%d)", total, n, H, W);
        return GS_ERR_NOMEM;
    }
    GS_HIP(hipMemset(ws, 0, total));   // halos and padded channel planes are zero from here on
    // (the fill runs on the NULL stream, which non-blocking streams -- the host pipelines' own, torch's side streams -- do not
    // wait for: a forward launched on one of them right after this call must not meet the fill still in flight)
    GS_HIP(hipDeviceSynchronize());
    size_t at = 0;
    for (Act *a : all) {
        a->base = reinterpret_cast<float *>(static_cast<char *>(ws) + at);
        at += round_up(a->bytes(n) + slack, 256);
    }
    if (lazy_b2) {
        m->bb[0] = m->a1;
        m->bb[0].base = m->a1.base + (long long)64 * m->a1.sc;
        m->bb[0].C = 64;
        m->bb[0].Cp = 64;
    }
    m->ee = m->a0c;   // comb_l2_l3 = the first planes of the concat buffer
    m->ee.C = cls;
    m->a0 = m->a0c;   // output0_cat = the planes after it (+ the zero plane)
    m->a0.base = m->a0c.base + (long long)cls * m->a0c.sc;
    m->a0.C = 19;
    m->ws = ws;
    m->ws_bytes = total;
    m->ws_n = n;
    m->ws_h = H;
    m->ws_w = W;
    return GS_OK;
}

// ------------------------------------------------------------------------------------------
struct Launcher {
    Model *m;
    hipStream_t s;
    gs_status st = GS_OK;
    int n;
    template <typename F>
    void run(int kid, double flops_per_tile, F &&f)
    {
        if (st != GS_OK)
            return;
        Model::Ev ev{nullptr, nullptr, kid};
        if (m->profile) {
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess ||
                hipEventRecord(ev.a, s) != hipSuccess) {
                set_error("profiling event setup failed");
                st = GS_ERR_HIP;
                return;
            }
        }
        st = f();
        if (st == GS_OK && hipGetLastError() != hipSuccess) {
            set_error("launch of %s failed", kKernelNames[kid]);
            st = GS_ERR_HIP;
        }
        if (m->profile && st == GS_OK) {
            hipEventRecord(ev.b, s);
            m->events.push_back(ev);
            m->prof_flops[kid] += flops_per_tile;   // summed like the times: launches of one kernel name may differ (fused / plain)
        }
    }
};

static ConvArgs conv_args(const Act &in, const float *wpack, const Act &out, const Act *res, int n)
{
    ConvArgs a{};
    a.in = in.base;
    a.in_sn = in.sn;
    a.in_sc = in.sc;
    a.in_pitch = in.pitch;
    a.in_off = in.off;
    a.in_img_bytes = (unsigned)(in.sn * sizeof(float));
    a.wpack = wpack;
    a.out = out.base;
    a.out_sn = out.sn;
    a.out_sc = out.sc;
    a.out_pitch = out.pitch;
    a.out_off = out.off;
    a.out_img_bytes = (unsigned)(out.sn * sizeof(float));
    if (res) {
        a.res = res->base;
        a.res_sn = res->sn;
        a.res_sc = res->sc;
        a.res_pitch = res->pitch;
        a.res_off = res->off;
        a.res_img_bytes = (unsigned)(res->sn * sizeof(float));
    }
    a.N = n;
    a.H = out.H;
    a.W = out.W;
    return a;
}

static inline unsigned blocks_for(long long items) { return (unsigned)((items + 255) / 256); }

// Timing / stamp variants of the forward (GS_VARIANT: ablations, per-wave stamps, the kernels a round replaced) live in
// espnet_diag.inc and exist in -DGS_DIAG builds only.  forward_impl offers them its launch sites through GS_DIAG_TRY; in the
// product build that macro expands to nothing (its arguments are not even evaluated), so what follows is the shipped
// schedule and nothing else.
#ifdef GS_DIAG
#include "espnet_diag.inc"
#define GS_DIAG_TRY(call)     \
    do {                      \
        gs_status dst_;       \
        if (call) return dst_; \
    } while (0)
#else
#define GS_DIAG_TRY(call) \
    do {                  \
    } while (0)
#define GS_DIAG_STAMPED(var, path, ca, ...)
#endif

template <int CLS>
static gs_status forward_impl(Model *m, const void *in, int in_format, int n, int H, int W, const float *mean,
                              const float *stdv, float *logits, uint8_t *mask, unsigned long long *hist, hipStream_t s,
                              float *prob = nullptr, int ens_mode = 0, float ens_w = 1.0f)
{
    const float *wb = m->dblob;
    Launcher L{m, s, GS_OK, n};
    const int H1 = H / 2, W1 = W / 2, H2 = H / 4, W2 = W / 4, H3 = H / 8, W3 = W / 8;
    const double px1 = (double)H1 * W1, px2 = (double)H2 * W2, px3 = (double)H3 * W3;
    m->stages.clear();
    m->last_n = n;
    // a ping-pong buffer that is written again no longer holds the stage recorded for it earlier
    auto set_stage = [&](const std::string &name, const Act &act, int C) {
        for (auto it = m->stages.begin(); it != m->stages.end();)
            it = it->second.first.base == act.base ? m->stages.erase(it) : std::next(it);
        m->stages[name] = {act, C};
    };

    // ---- stem (Model.py:346-350)
    L.run(K_STEM, px1 * (27 * 16 * 2), [&] {
        StemArgs a{};
        a.in = in;
        for (int i = 0; i < 3; ++i) {
            a.mean[i] = mean ? mean[i] : 0.0f;
            a.std[i] = stdv ? stdv[i] : 1.0f;
        }
        std::memcpy(a.w1, m->stem_params, sizeof(float) * 432);
        std::memcpy(a.bn1, m->stem_params + 432, sizeof(float) * 48);
        std::memcpy(a.b1, m->stem_params + 480, sizeof(float) * 57);
        a.a0 = view(m->a0);
        a.inp1 = view(m->inp1);
        a.N = n;
        a.H = H;
        a.W = W;
        if (hist && !m->encoder_only && (ens_mode == 0 || ens_mode >= 3)) {   // zeroed by the first kernel of the forward; the last one adds into it
            a.hist_zero = hist;
            a.hist_count = n * m->classes;
        }
        if (in_format == GS_IN_U8_BGR_NHWC)
            hipLaunchKernelGGL(stem_kernel<true>, dim3(blocks_for((long long)n * H1 * ((W1 + STEM_PX - 1) / STEM_PX))), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(stem_kernel<false>, dim3(blocks_for((long long)n * H1 * ((W1 + STEM_PX - 1) / STEM_PX))), dim3(256), 0, s, a);
        return GS_OK;
    });
    L.run(K_POOL, 0, [&] {
        hipLaunchKernelGGL(pool_kernel, dim3(blocks_for((long long)n * 3 * H2 * W2)), dim3(256), 0, s, view(m->inp1),
                           view(m->inp2), n, 3, m->p > 0 ? wb + m->b2 : nullptr, view(m->a1), 128, 131);
        return GS_OK;
    });
    set_stage("b1", m->a0, 19);
    set_stage("sample2", m->inp2, 3);

    // ---- level 2 (Model.py:351-357): DownSamplerB(19,64) then p ESP blocks
    L.run(K_L2_C1S, px2 * (19 * 9 * 12 * 2), [&] {
        GS_DIAG_TRY(diag_reduce_s2(m, 2, conv_args(m->a0, wb + m->l2_0.c1, m->r2[0], nullptr, n), s, dst_));
        ConvArgs ca = conv_args(m->a0, wb + m->l2_0.c1, m->r2[0], nullptr, n);
        ca.rev_n = CFG_L2_C1S_REV;
        return launch_conv_mfma<CFG_L2_C1S, F_S2PAIR | POL_L2_C1S | AGL_S2 | S2FLIP_L2>(ca, m->num_cus, s);
    });
    // b2 = BR(131) over cat([output1, output1_0, inp2]) (Model.py:359) never runs as a kernel: the last ESP block stores
    // only its b2-normalised form (planes 0..63 of output1_cat), the pool kernel writes planes 128..130 normalised, and
    // output1_0 is stored RAW into planes 64..127 (lazy b2, CFG_LAZY_B2 above: the consumers normalise on load; the eager
    // form stored it twice, raw for the ESP blocks and normalised for the cat).  With p == 0 the unfused cat kernel runs.
    auto with_dual = [&](ConvArgs a, int coff) {
        a.out2 = m->a1.base;
        a.out2_sn = m->a1.sn;
        a.out2_sc = m->a1.sc;
        a.out2_pitch = m->a1.pitch;
        a.out2_off = m->a1.off;
        a.out2_coff = coff;
        a.out2_img_bytes = (unsigned)(m->a1.sn * sizeof(float));
        return a;
    };
    // F_FUSE1X1: the next block's reduced map is written by this block's epilogue (the two maps alternate)
    auto with_fused = [&](ConvArgs a, const Act &r, int nout3) {
        a.out3 = r.base;
        a.out3_sn = r.sn;
        a.out3_sc = r.sc;
        a.out3_pitch = r.pitch;
        a.out3_off = r.off;
        a.out3_img_bytes = (unsigned)(r.sn * sizeof(float));
        a.nout3 = nout3;
        return a;
    };
    // small batches: 32-pixel level-2 tasks while there are at most CFG_SMALL2_WAVES of them per CU (see small3 below)
    const bool small2 = CFG_SMALL2_WAVES > 0 && (long long)n * H2 * cdiv(W2, 64) * 2 <= (long long)m->num_cus * CFG_SMALL2_WAVES && W2 % 2 == 0 && !no_vec();
    const bool fuse_b2 = m->p > 0;
    const bool lazy_b2 = fuse_b2 && CFG_LAZY_B2;
    m->b2_lazy = lazy_b2;
    int rd2 = 0;   // index of the reduced map the next level-2 branch kernel reads
    L.run(K_L2_DOWN, px2 * (12 * 9 * 64 * 2) + (m->l2_0.fused_next ? px2 * (64 * 12 * 2) : 0), [&] {
        ConvArgs ca = conv_args(m->r2[rd2], wb + m->l2_0.br, m->bb[0], nullptr, n);
        if (lazy_b2) {   // raw output only: b2 is applied by the consumers (level-3 stride-2 reduce, dec2)
            if (m->l2_0.fused_next)
            {
                if (small2)
                    return launch_vec<F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | FUSE_L2, CFG_L2_BR_P2S>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
                GS_DIAG_STAMPED(162, "gpurun_out/stamps_l2down.txt", with_fused(ca, m->r2[rd2 ^ 1], 12), F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | FUSE_L2 | F_VEC | SKIP_L2, CFG_L2_BR_P4)
#if CFG_L2_DOWN_SKIP
                if (ca.W % L2SP == 0 && !no_vec())   // tap-row chunks, tap rows in the zero halo skipped (espnet_config.h)
                    return launch_conv_mfma<CFG_L2_BR_P4S, F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | FUSE_L2 | F_VEC | F_SKIP_PAD>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
#endif
                return launch_vec<F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | FUSE_L2 | EPIPE_L2_DOWN | SKIP_L2, CFG_L2_BR_P4>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
            }
            return launch_vec<F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(ca, m->num_cus, s);
        }
        if (fuse_b2) {
            ca = with_dual(ca, 64);
            GS_DIAG_TRY(diag_l2_down(m, ca, with_fused(conv_args(m->r2[rd2], wb + m->l2_0.br, m->bb[0], nullptr, n), m->r2[rd2 ^ 1], 12), s, dst_));
            if (m->l2_0.fused_next)
                return launch_vec<F_BNACT | F_DUAL | POL_L2_DOWN | AGL_L2 | FUSE_L2 | SKIP_L2, CFG_L2_BR_P4>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
            return launch_vec<F_BNACT | F_DUAL | POL_L2_DOWN | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(ca, m->num_cus, s);
        }
        return launch_vec<F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(ca, m->num_cus, s);
    });
    bool have_r2 = m->l2_0.fused_next;   // the reduced map of the next block already exists
    rd2 ^= have_r2 ? 1 : 0;
    set_stage("level2_0", m->bb[0], 64);
    int cur2 = 0;
    for (int i = 0; i < m->p; ++i) {
        const int nxt = cur2 == 1 ? 2 : 1;
        const bool last = i == m->p - 1;
        const bool fuse_next = m->l2[i].fused_next;
        if (!have_r2)
            L.run(K_L2_C1, px2 * (64 * 12 * 2), [&] {
                ConvArgs ca = conv_args(m->bb[cur2], wb + m->l2[i].c1, m->r2[rd2], nullptr, n);
                return launch_conv_mfma<CFG_L2_C1, POL_L2_C1>(ca, m->num_cus, s);   // 1x1: the run mapping measured no slower
            });
        L.run(K_L2_ESP, px2 * (12 * 9 * 64 * 2) + (fuse_next ? px2 * (64 * 12 * 2) : 0), [&] {
            ConvArgs ca = conv_args(m->r2[rd2], wb + m->l2[i].br, m->bb[nxt], &m->bb[cur2], n);
            GS_DIAG_TRY(diag_l2_esp(m, ca, with_dual(ca, 0), last, s, dst_));
            if (last) {
                if (small2)
                    return launch_vec<F_BNACT | F_RES | F_NOSTORE | F_DUAL | POL_L2_LAST | AGL_L2, CFG_L2_BR_P2S>(with_dual(ca, 0), m->num_cus, s);
                GS_DIAG_STAMPED(163, "gpurun_out/stamps_l2last.txt", with_dual(ca, 0), F_BNACT | F_RES | F_NOSTORE | F_DUAL | POL_L2_LAST | AGL_L2 | F_VEC | SKIP_L2, CFG_L2_BR_P4)
                return launch_vec<F_BNACT | F_RES | F_NOSTORE | F_DUAL | POL_L2_LAST | AGL_L2 | EPIPE_L2_ESP | SKIP_L2, CFG_L2_BR_P4>(with_dual(ca, 0), m->num_cus, s);
            }
            if (fuse_next) {
                if (small2)
                    return launch_vec<F_BNACT | F_RES | POL_L2_ESP | AGL_L2 | FUSE_L2, CFG_L2_BR_P2S>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
                GS_DIAG_STAMPED(161, "gpurun_out/stamps_l2esp.txt", with_fused(ca, m->r2[rd2 ^ 1], 12), F_BNACT | F_RES | POL_L2_ESP | AGL_L2 | FUSE_L2 | F_VEC | SKIP_L2, CFG_L2_BR_P4)
                return launch_vec<F_BNACT | F_RES | POL_L2_ESP | AGL_L2 | FUSE_L2 | EPIPE_L2_ESP | SKIP_L2, CFG_L2_BR_P4>(with_fused(ca, m->r2[rd2 ^ 1], 12), m->num_cus, s);
            }
            return launch_vec<F_BNACT | F_RES | POL_L2_ESP | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(ca, m->num_cus, s);
        });
        have_r2 = fuse_next;
        rd2 ^= have_r2 ? 1 : 0;
        cur2 = nxt;
        if (!last)
            set_stage("level2." + std::to_string(i), m->bb[cur2], 64);
    }
    if (!fuse_b2) {
        L.run(K_CAT_B2, 0, [&] {
            hipLaunchKernelGGL(cat_b2_kernel, dim3(blocks_for((long long)n * 131 * H2 * W2)), dim3(256), 0, s, view(m->bb[cur2]),
                               view(m->bb[0]), view(m->inp2), wb + m->b2, view(m->a1), n);
            return GS_OK;
        });
    }
    set_stage("b2", m->a1, 131);

    // ---- level 3 (Model.py:361-366)
    // Small batches: a half-row task per SIMD does not fill the chip below 8 tiles (64 rows x 2 strips x n tasks for 1024
    // SIMDs); with 32-pixel strips (one pixel per lane, the same accumulation chain per pixel: same bits) there are twice as
    // many, each half as long.  Used up to one such task per wave slot (CFG_SMALL3_WAVES per CU: 8 tiles).
    const bool small3 = (long long)n * H3 * cdiv(W3, 64) * 2 <= (long long)m->num_cus * CFG_SMALL3_WAVES && !no_vec();
    int rd3 = 0;
    L.run(K_L3_C1S, px3 * (131 * 9 * 25 * 2), [&] {
        GS_DIAG_TRY(diag_reduce_s2(m, 3, conv_args(m->a1, wb + m->l3_0.c1, m->r3[0], nullptr, n), s, dst_));
        if (lazy_b2) {   // planes 64..127 of output1_cat hold output1_0 RAW: b2 is applied to the B operands on load
            ConvArgs ca = conv_args(m->a1, wb + m->l3_0.c1, m->r3[0], nullptr, n);
            ca.bnl_s0 = 64 / 2;      // k-groups of two channels
            ca.bnl_s1 = 128 / 2;
            // whole-row tasks (128 pixels) are one per wave at batch 32; below a quarter of that the row is cut into 32-pixel
            // tasks (batch 1: 64 -> 256 tasks, 0.180 -> see profiles/r04_latency.json)
            if ((long long)n * H3 * cdiv(W3, 128) * 4 <= (long long)m->num_cus * 8)
                return launch_conv_mfma<CFG_L3_C1S_BNL_P1, F_S2PAIR | POL_L3_C1S | S2FLIP_L3 | F_BNLOAD>(ca, m->num_cus, s);
            GS_DIAG_STAMPED(165, "gpurun_out/stamps_l3c1s.txt", ca, F_S2PAIR | POL_L3_C1S | S2FLIP_L3 | F_BNLOAD, CFG_L3_C1S_BNL)
            return launch_conv_mfma<CFG_L3_C1S_BNL, F_S2PAIR | POL_L3_C1S | S2FLIP_L3 | F_BNLOAD>(ca, m->num_cus, s);
        }
        return launch_conv_mfma<CFG_L3_C1S, F_S2PAIR | POL_L3_C1S | AGL_S2 | S2FLIP_L3>(conv_args(m->a1, wb + m->l3_0.c1, m->r3[0], nullptr, n), m->num_cus, s);
    });
    L.run(K_L3_DOWN, px3 * (25 * 9 * 128 * 2) + (m->l3_0.fused_next ? px3 * (128 * 25 * 2) : 0), [&] {
        ConvArgs ca = conv_args(m->r3[rd3], wb + m->l3_0.br, m->cc[0], nullptr, n);
        if (m->l3_0.fused_next) {   // no residual here: the four-pixel vector mapping still fits with the second accumulator set
            if (small3)
                return launch_conv_mfma<CFG_L3_BR_P1R, F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3 | SKIP_L3 | (CFG_SMALL_AGL ? F_A_GLOBAL : 0)>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
#if CFG_L3_W16 & 4
            return launch_conv_mfma<CFG_L3_BR_W16, F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3 | SKIP_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
#endif
#if CFG_L3_DOWN_P2
            if (ca.W % 2 == 0 && !no_vec()) {
                GS_DIAG_STAMPED(164, "gpurun_out/stamps_l3down.txt", with_fused(ca, m->r3[rd3 ^ 1], 25), F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3 | F_VEC | SKIP_L3, CFG_L3_BR_P2R)
                return launch_conv_mfma<CFG_L3_BR_P2R, F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3 | F_VEC | SKIP_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
            }
#endif
            if (ca.W % 4 == 0 && !no_vec())
                return launch_conv_mfma<CFG_L3_BR, F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3 | F_VEC>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
            return launch_conv_mfma<CFG_L3_BR_P2F, F_BNACT | POL_L3_DOWN | AGL_L3 | FUSE_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
        }
        return launch_vec<F_BNACT | POL_L3_DOWN | AGL_L3, CFG_L3_BR>(ca, m->num_cus, s);
    });
    bool have_r3 = m->l3_0.fused_next;
    set_stage("level3_reduce", m->r3[rd3], 25);      // (debug: valid until the second ESP block overwrites the map)
    rd3 ^= have_r3 ? 1 : 0;
    set_stage("level3_0", m->cc[0], 128);
    int cur3 = 0;
    for (int i = 0; i < m->q; ++i) {
        const int nxt = cur3 == 1 ? 2 : 1;
        const bool fuse_next = m->l3[i].fused_next;
        if (!have_r3)
            L.run(K_L3_C1, px3 * (128 * 25 * 2), [&] {
                ConvArgs ca = conv_args(m->cc[cur3], wb + m->l3[i].c1, m->r3[rd3], nullptr, n);
                return launch_conv_mfma<CFG_L3_C1, POL_L3_C1>(ca, m->num_cus, s);
            });
        L.run(K_L3_ESP, px3 * (25 * 9 * 128 * 2) + (fuse_next ? px3 * (128 * 25 * 2) : 0), [&] {
            ConvArgs ca = conv_args(m->r3[rd3], wb + m->l3[i].br, m->cc[nxt], &m->cc[cur3], n);
            GS_DIAG_TRY(diag_l3_esp(m, ca, i, s, dst_));
            if (fuse_next) {
                if (small3)   // (one pixel per lane: a whole slot's residual fits in registers, requested a dilation ahead)
                    return launch_conv_mfma<CFG_L3_BR_P1R, F_BNACT | F_RES | POL_L3_ESP | AGL_L3 | FUSE_L3 | SKIP_L3 | (CFG_SMALL_AGL ? F_A_GLOBAL : 0)>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
#if CFG_L3_W16 & 1
                return launch_conv_mfma<CFG_L3_BR_W16, F_BNACT | F_RES | POL_L3_ESP | AGL_L3 | FUSE_L3 | SKIP_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
#endif
#if CFG_L3_FUSE_P4
                // four pixels per lane with the residual through a half-slot register ring (round 2's first fused form)
                if (ca.W % 4 == 0 && !no_vec())
                    return launch_conv_mfma<CFG_L3_BR, F_BNACT | F_RES | F_RES_RING | F_VEC | POL_L3_ESP | AGL_L3 | FUSE_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
#endif
                // Two consecutive pixels per lane, a 39-step operand ring, the residual through the half-slot register
                // ring.  A task is then half a row, so the 256 waves of an XCD have TWO images in flight instead of four
                // and the reduced maps the taps re-read stay in that XCD's 4 MiB L2: beyond-L2 fetch of a launch
                // 542 -> 296 MB, 0.1898 -> 0.1834 ms (profiles/README.md).
                if (ca.W % 2 == 0 && !no_vec()) {
                    if (i == 1) {
                        GS_DIAG_STAMPED(160, "gpurun_out/stamps_l3esp.txt", with_fused(ca, m->r3[rd3 ^ 1], 25), F_BNACT | F_RES | F_RES_RING | F_VEC | POL_L3_ESP | AGL_L3 | FUSE_L3 | SKIP_L3, CFG_L3_BR_P2R)
                    }
                    return launch_conv_mfma<CFG_L3_BR_P2R, F_BNACT | F_RES | F_RES_RING | F_VEC | POL_L3_ESP | AGL_L3 | FUSE_L3 | SKIP_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
                }
                return launch_conv_mfma<CFG_L3_BR_P2F, F_BNACT | F_RES | AGL_L3 | FUSE_L3>(with_fused(ca, m->r3[rd3 ^ 1], 25), m->num_cus, s);
            }
            if (small3)
                return launch_conv_mfma<CFG_L3_BR_P1R, F_BNACT | F_RES | POL_L3_ESP | AGL_L3 | SKIP_L3 | (CFG_SMALL_AGL ? F_A_GLOBAL : 0)>(ca, m->num_cus, s);
#if CFG_L3_W16 & 2
            return launch_conv_mfma<CFG_L3_BR_W16, F_BNACT | F_RES | POL_L3_ESP | AGL_L3 | SKIP_L3>(ca, m->num_cus, s);
#endif
#if CFG_L3_LAST_P2
            // the last (unfused) block in the half-row task shape of the fused ones, tap rows in the halo skipped
            if (ca.W % 2 == 0 && !no_vec())
                return launch_conv_mfma<CFG_L3_BR_P2R, F_BNACT | F_RES | F_RES_RING | F_VEC | POL_L3_ESP | AGL_L3 | SKIP_L3>(ca, m->num_cus, s);
#endif
            // four consecutive pixels per lane and 16-byte accesses when the width allows it (0.170 ms per
            // launch at batch 32), else the two-run mapping with its deeper ring (0.175 ms)
            if (ca.W % 4 == 0 && !no_vec())
                return launch_conv_mfma<CFG_L3_BR, F_BNACT | F_RES | F_VEC | POL_L3_ESP | AGL_L3>(ca, m->num_cus, s);
            return launch_conv_mfma<CFG_L3_BR_P2, F_BNACT | F_RES | AGL_L3>(ca, m->num_cus, s);
        });
        have_r3 = fuse_next;
        rd3 ^= have_r3 ? 1 : 0;
        cur3 = nxt;
        set_stage("level3." + std::to_string(i), m->cc[cur3], 128);
    }

    // ---- b3 + classifier (+ br + up_l3)  (Model.py:368-370)
    const int ncls = m->classes;   // real class count (CLS is the padded one): algorithmic FLOPs, output widths
    L.run(K_DEC1, px3 * (256 * ncls * 2) + px3 * (ncls * ncls * 4 * 2), [&] {
        Dec1Args a{};
        a.c0 = view(m->cc[0]);
        a.clast = view(m->cc[cur3]);
        a.b3w = wb + m->b3;
        a.br = m->encoder_only ? nullptr : wb + m->br;
        a.wup = m->encoder_only ? nullptr : wb + m->wup3;
        a.out = view(m->o2c);
        a.enc_logits = m->encoder_only ? logits : nullptr;
        a.N = n;
        a.classes = ncls;
        // (channel batch: 16 channels x (3 + CLS) scalar constants in flight fit the scalar registers for five classes only)
        constexpr int CB = CLS <= 5 ? 16 : CLS <= 8 ? 8 : 4;
        hipLaunchKernelGGL((dec1_kernel<CLS, CB>), dim3((unsigned)(((long long)n * H3 * W3 + 63) / 64)), dim3(256), 0, s, a);
        return GS_OK;
    });
    if (m->encoder_only)
        return L.st;
    set_stage("up_l3", m->o2c, ncls);

    // ---- level3_C + cat + BR (Model.py:372-373)
    L.run(K_DEC2, px2 * (131 * ncls * 2), [&] {
        Dec2Args a{};
        a.a1 = view(m->a1);
        a.raw = view(m->bb[0]);
        a.b2 = wb + m->b2;
        a.raw_c0 = 64;
        a.raw_cn = lazy_b2 ? 64 : 0;
        a.o2c = view(m->o2c);
        a.w3c = wb + m->w3c;
        a.br = wb + m->cbr0;
        a.t = view(m->tt);
        a.N = n;
        a.classes = ncls;
        hipLaunchKernelGGL(dec2_kernel<CLS>, dim3((unsigned)(((long long)n * H2 * W2 + 63) / 64)), dim3(256), 0, s, a);   // 64 pixels x 4 channel quarters
        return GS_OK;
    });
    if (ncls == CLS)   // (with padding planes in between the two halves are not one contiguous stage)
        set_stage("combine_t", m->tt, 2 * CLS);
    // ---- CBR(2c,c,3) + up_l2 (Model.py:373)
    if constexpr (CLS >= 12) {
        // many classes: the 3x3 over 2 * CLS planes is 7 200 FMAs per pixel at twenty classes -- on the matrix cores (a plain
        // conv_mfma launch, BN + PReLU in its epilogue), the deconvolution + BR as a second, small kernel
        L.run(K_DEC3, px2 * (2 * ncls * 9 * ncls * 2), [&] {
            const ConvArgs ca = conv_args(m->tt, wb + m->wcc_mfma, m->t3, nullptr, n);
            if constexpr (CLS <= 16)
                return launch_vec<F_BNACT, 16, 8, 2 * CLS, 9, 1, 1, CLS, CLS, 8, 3>(ca, m->num_cus, s);
            else
                return launch_vec<F_BNACT, 32, 8, 2 * CLS, 9, 1, 1, CLS, CLS, 4, 3>(ca, m->num_cus, s);
        });
        L.run(K_DEC3, px2 * (ncls * ncls * 4 * 2), [&] {
            Dec3Args a{};
            a.t = view(m->t3);
            a.wup = wb + m->wup2;
            a.bnu = wb + m->bnu2;
            a.e = view(m->ee);
            a.N = n;
            a.classes = ncls;
            hipLaunchKernelGGL(dec3b_kernel<CLS>, dim3(blocks_for((long long)n * H2 * W2)), dim3(256), 0, s, a);
            return GS_OK;
        });
    } else {
        L.run(K_DEC3, px2 * (2 * ncls * 9 * ncls * 2) + px2 * (ncls * ncls * 4 * 2), [&] {
            Dec3Args a{};
            a.t = view(m->tt);
            a.wc = wb + m->wcc;
            a.bnc = wb + m->bncc;
            a.wup = wb + m->wup2;
            a.bnu = wb + m->bnu2;
            a.e = view(m->ee);
            a.N = n;
            a.classes = ncls;
            hipLaunchKernelGGL(dec3_kernel<CLS>, dim3(blocks_for((long long)n * H2 * W2)), dim3(256), 0, s, a);
            return GS_OK;
        });
    }
    set_stage("up_l2", m->ee, ncls);
    // ---- conv CBR(19+c,c,3) + classifier deconv + argmax + counts (Model.py:375-377, VisualizeResults_iou.py:128,151-155)
    if constexpr (CLS == 5) {
        GS_DIAG_TRY((diag_two_kernel_tail<CLS>(m, L, n, H1, W1, logits, mask, hist, s, set_stage, dst_)));
        L.run(K_DEC_TAIL, px1 * ((19 + CLS) * 9 * CLS * 2) + px1 * (CLS * CLS * 4 * 2), [&] {
            DecTailArgs a{};
            a.in = m->a0c.base;
            a.in_sn = m->a0c.sn;
            a.in_sc = m->a0c.sc;
            a.in_pitch = m->a0c.pitch;
            a.in_off = m->a0c.off;
            a.in_img_bytes = (unsigned)(m->a0c.sn * sizeof(float));
            a.wpack = wb + m->wtail;
            a.logits = logits;
            a.mask = mask;
            a.hist = (ens_mode == 1 || ens_mode == 2) ? nullptr : hist;   // only the last member of an ensemble counts
            a.prob = prob;
            a.ens_mode = ens_mode;
            a.ens_w = ens_w;
            if (logits) {   // debug / test path: the half-resolution CBR output is kept as stage "conv"
                a.ff = m->ff.base;
                a.ff_sn = m->ff.sn;
                a.ff_sc = m->ff.sc;
                a.ff_pitch = m->ff.pitch;
                a.ff_off = m->ff.off;
            }
            a.N = n;
            a.H1 = H1;
            a.W1 = W1;
            return launch_dec_tail(a, m->num_cus, s);
        });
        if (logits)
            set_stage("conv", m->ff, CLS);
    } else {
        // Any other class count (Model.py:311: `classes` is free, 20 by default): the 3x3 over the 19 + classes planes of the concat
        // buffer as a plain conv_mfma launch (MFMA rows = the padded output channels, BN + PReLU in its epilogue), then the
        // classifier deconvolution + argmax + counts (+ the ensemble's softmax accumulation) as a second kernel.
        L.run(K_DEC_CONV, px1 * ((19 + ncls) * 9 * ncls * 2), [&] {
            const ConvArgs ca = conv_args(m->a0c, wb + m->wconv, m->ff, nullptr, n);
            constexpr int CINP = (19 + CLS + 3) / 4 * 4;
            // (deeper operand rings and four pixels per lane were measured on these launches: no gain -- at twenty classes the launch is
            // at 90 % of what its padded matrix work allows, 20 of 32 rows)
            if constexpr (CLS <= 16)
                return launch_vec<F_BNACT | POL_DEC_CONV, 16, 8, CINP, 9, 1, 1, CLS, CLS, 8, 3>(ca, m->num_cus, s);
            else
                return launch_vec<F_BNACT | POL_DEC_CONV, 32, 8, CINP, 9, 1, 1, CLS, CLS, 4, 3>(ca, m->num_cus, s);
        });
        set_stage("conv", m->ff, ncls);
        L.run(K_DEC4, px1 * (ncls * ncls * 4 * 2), [&] {
            Dec4Args a{};
            a.f = view(m->ff);
            a.wcl = wb + m->wclassifier;
            a.logits = logits;
            a.mask = mask;
            a.hist = hist;
            a.prob = prob;
            a.ens_mode = ens_mode;
            a.ens_w = ens_w;
            a.N = n;
            a.classes = ncls;
            const dim3 grid(blocks_for(((long long)H1 * W1 + dec4_px<CLS>() - 1) / dec4_px<CLS>()), n);
            if (ens_mode)
                hipLaunchKernelGGL((dec4_kernel<CLS, true>), grid, dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL((dec4_kernel<CLS, false>), grid, dim3(256), 0, s, a);
            return GS_OK;
        });
    }
    return L.st;
}

// the decoder kernels exist for the padded class counts 4, 5, 8, 12, 16, 20 (Model::cp)
static gs_status forward_any(Model *m, const void *in, int in_format, int n, int H, int W, const float *mean, const float *stdv,
                             float *logits, uint8_t *mask, unsigned long long *hist, hipStream_t s, float *prob = nullptr,
                             int ens_mode = 0, float ens_w = 1.0f)
{
    switch (m->cp) {
    case 5: return forward_impl<5>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    case 4: return forward_impl<4>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    case 8: return forward_impl<8>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    case 12: return forward_impl<12>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    case 16: return forward_impl<16>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    case 20: return forward_impl<20>(m, in, in_format, n, H, W, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
    }
    set_error("internal: no decoder instantiation for %d padded classes", m->cp);
    return GS_ERR_UNSUPPORTED;
}

static void free_pipeline(Model &m)
{
    for (auto &s : m.sl) {
        if (s.hin) hipHostFree(s.hin);
        if (s.hout) hipHostFree(s.hout);
        if (s.hh) hipHostFree(s.hh);
        if (s.din) hipFree(s.din);
        if (s.dout) hipFree(s.dout);
        if (s.dh) hipFree(s.dh);
        if (s.up) hipEventDestroy(s.up);
        if (s.done) hipEventDestroy(s.done);
        if (s.down) hipEventDestroy(s.down);
        s = Model::Slot();
    }
    m.pipe_in_bytes = m.pipe_out_bytes = 0;
    m.pipe_batch = 0;
}

}  // namespace gs

using namespace gs;

// ==========================================================================================
extern "C" {

const char *gs_last_error(void) { return g_err.c_str(); }
int gs_abi_version(void) { return 5; }   // 2: lanes, block hook, detector, compositor LUT; 3: batched crop entries, detector host entry, build flags; 4: any class count 2..20 (hist is [n,classes]), batch planner, pinned-block query, overlays from the crop pipeline; 5: gs_device_fault_check, GS_ERR_DEVICE_FAULT
gs_status gs_device_fault_check(void)
{
    GS_HIP(hipDeviceSynchronize());
    int flags = 0;
    gs_status st = dec_tail_fault_flags(&flags);
    if (st != GS_OK) return st;
    if (flags) {
        set_error("device fault word %d: a decoder-tail wave gave up its strip-boundary exchange; the results of the calls since the last check are invalid", flags);
        return GS_ERR_DEVICE_FAULT;
    }
    return GS_OK;
}
int gs_build_flags(void)
{
#ifdef GS_DIAG
    return GS_BUILD_DIAG;
#else
    return 0;
#endif
}

// A handle = the model (weights + lane-0 workspace) plus optional extra LANES: shallow copies of the model that share
// the device weight blob and own a workspace of their own, so that two batches can be in flight on two HIP streams
// (gs_espnet_forward_lane).  Consecutive kernels of one forward cannot overlap (each waits for its predecessor and the
// big ones fill every CU), but the tail of one batch's kernel and the head of another batch's can.
struct gs_espnet {
    Model m;
    std::vector<std::unique_ptr<Model>> lanes;   // lane k >= 1 is lanes[k - 1]
    hipStream_t pipe_compute2 = nullptr;         // gs_espnet_segment_host: the second of its two streams (odd batches)
    gs::CropPipe *crop_pipe = nullptr;           // gs_espnet_segment_crops*: staging state (csrc/crops.hip)
    Model &lane(int k) { return k == 0 ? m : *lanes[k - 1]; }
};

}  // extern "C"

namespace gs {
CropPipe *&espnet_crop_pipe(gs_espnet *h) { return h->crop_pipe; }
int espnet_device(gs_espnet *h) { return h->m.device; }
int espnet_is_full_net(gs_espnet *h) { return h->m.encoder_only ? 0 : 1; }
int espnet_lanes(gs_espnet *h) { return 1 + (int)h->lanes.size(); }
int espnet_classes(gs_espnet *h) { return h->m.classes; }
// the ensemble's fp32 probability accumulator [n][classes][height][width], owned by the first member's handle
gs_status ensemble_scratch(gs_espnet *h, int n, int height, int width, float **prob)
{
    Model &m0 = h->m;
    const size_t need = (size_t)n * m0.classes * height * width * sizeof(float);
    if (m0.prob_bytes < need) {
        if (m0.prob) {
            GS_HIP(hipDeviceSynchronize());
            GS_HIP(hipFree(m0.prob));
        }
        m0.prob = nullptr;
        m0.prob_bytes = 0;
        if (hipMalloc(reinterpret_cast<void **>(&m0.prob), need) != hipSuccess) {
            set_error("ensemble scratch allocation of %zu bytes failed", need);
            return GS_ERR_NOMEM;
        }
        m0.prob_bytes = need;
    }
    *prob = m0.prob;
    return GS_OK;
}
// the forward with every option (ensemble accumulator included); arguments are the caller's responsibility beyond the
// checks of gs_espnet_forward_lane
gs_status espnet_forward_ex(gs_espnet *h, int lane, const void *in, int in_format, int n, int height, int width, const float *mean,
                            const float *stdv, float *logits, uint8_t *mask, unsigned long long *hist, float *prob, int ens_mode,
                            float ens_w, hipStream_t s)
{
    GS_REQUIRE(h && in, "forward: null handle or input");
    GS_REQUIRE(lane >= 0 && lane <= (int)h->lanes.size(), "lane %d does not exist (gs_espnet_set_lanes)", lane);
    Model &m = h->lane(lane);
    GS_REQUIRE(!m.encoder_only || ens_mode == 0, "an ESPNet-C handle cannot be an ensemble member");
    gs_status st = layout_workspace(&m, n, height, width);
    if (st != GS_OK) return st;
    return forward_any(&m, in, in_format, n, height, width, mean, stdv, logits, mask, hist, s, prob, ens_mode, ens_w);
}
}  // namespace gs

extern "C" {

gs_status gs_espnet_create(const float *blob, const gs_layer_desc *table, int n_layers, int classes, int p, int q,
                           int encoder_only, gs_espnet **out)
{
    GS_REQUIRE(blob && table && out && n_layers > 0, "gs_espnet_create: null argument");
    GS_REQUIRE(p >= 0 && q >= 0, "gs_espnet_create: p and q must be non-negative");
    // Model.py:311,246: `classes` is free (20 by default).  The decoder kernels exist for 2..20 (padded to 4, 5, 8, 12, 16, 20
    // planes: Model::cp); class maps are uint8 and a lane's per-class counters are sized for at most 20.
    if (classes < 2 || classes > 20) {
        set_error("gs_espnet_create: classes must be 2..20 (got %d)", classes);
        return GS_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("gs_espnet_create: no HIP device visible");
        return GS_ERR_NODEVICE;
    }
    std::unique_ptr<gs_espnet> h(new gs_espnet());
    Model &m = h->m;
    m.classes = classes;
    m.cp = classes == 5 ? 5 : (classes + 3) / 4 * 4;
    m.p = p;
    m.q = q;
    m.encoder_only = encoder_only != 0;
    GS_HIP(hipGetDevice(&m.device));
    hipDeviceProp_t prop;
    GS_HIP(hipGetDeviceProperties(&prop, m.device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("gs_espnet_create: device %d is %s; this library is built for gfx950 only", m.device, prop.gcnArchName);
        return GS_ERR_NODEVICE;
    }
    m.num_cus = prop.multiProcessorCount;
#ifdef GS_DIAG
    if (const char *v = std::getenv("GS_VARIANT"))
        m.variant = std::atoi(v);
#endif

    WeightTable t;
    t.blob = blob;
    for (int i = 0; i < n_layers; ++i)
        t.by_name[std::string(table[i].name)] = &table[i];
    const std::string e = m.encoder_only ? "" : "encoder.";
    const int c = classes;
    BlobBuilder bb;
    std::vector<float> tmp(3 * 256);

    const float *w;
    if (!(w = t.get(e + "level1.conv.weight", {16, 3, 3, 3}))) return GS_ERR_INVALID;
    m.w1 = bb.push(w, 432);
    if (!fold_bn(t, e + "level1.bn", e + "level1.act", 16, tmp.data())) return GS_ERR_INVALID;
    m.bn1 = bb.push(tmp.data(), 48);
    if (!fold_bn(t, e + "b1.bn", e + "b1.act", 19, tmp.data())) return GS_ERR_INVALID;
    m.b1 = bb.push(tmp.data(), 57);
    std::memcpy(m.stem_params, bb.data.data() + m.w1, sizeof(float) * 432);
    std::memcpy(m.stem_params + 432, bb.data.data() + m.bn1, sizeof(float) * 48);
    std::memcpy(m.stem_params + 480, bb.data.data() + m.b1, sizeof(float) * 57);
    std::vector<float> b2f(3 * 131);
    if (!fold_bn(t, e + "b2.bn", e + "b2.act", 131, b2f.data())) return GS_ERR_INVALID;
    m.b2 = bb.push(b2f.data(), 393);
    auto next2 = [&](int i) { return (CFG_FUSE_L2 && i < p) ? e + "level2." + std::to_string(i) : std::string(); };
    auto next3 = [&](int i) { return ((CFG_FUSE_L3 == 2 || (CFG_FUSE_L3 == 1 && i == 0)) && i < q) ? e + "level3." + std::to_string(i) : std::string(); };
    // (lazy b2: the down-sampler has no second store, so its image carries no second BN section)
    if (!pack_block(t, bb, e + "level2_0", true, 2, m.l2_0, (p > 0 && !CFG_LAZY_B2) ? b2f.data() : nullptr, 64, 131, next2(0))) return GS_ERR_INVALID;
    m.l2.resize(p);
    for (int i = 0; i < p; ++i)
        if (!pack_block(t, bb, e + "level2." + std::to_string(i), false, 2, m.l2[i], i == p - 1 ? b2f.data() : nullptr, 0, 131, next2(i + 1)))
            return GS_ERR_INVALID;
    if (!pack_block(t, bb, e + "level3_0", true, 3, m.l3_0, nullptr, 0, 0, next3(0), b2f.data(), 64, 64, 131)) return GS_ERR_INVALID;
    m.l3.resize(q);
    for (int i = 0; i < q; ++i)
        if (!pack_block(t, bb, e + "level3." + std::to_string(i), false, 3, m.l3[i], nullptr, 0, 0, next3(i + 1))) return GS_ERR_INVALID;
    // ---- decoder: every piece is packed for cp class planes, zero beyond the model's c (espnet_kernels.h, "CLASS COUNTS")
    const int cp = m.cp;
    // [rows][cols][2][2] deconvolution weights -> [cp][cp][2][2]
    auto pad_deconv = [&](const float *src) {
        std::vector<float> o((size_t)cp * cp * 4, 0.0f);
        for (int i = 0; i < c; ++i)
            for (int o2 = 0; o2 < c; ++o2)
                for (int k = 0; k < 4; ++k)
                    o[((size_t)i * cp + o2) * 4 + k] = src[((size_t)i * c + o2) * 4 + k];
        return o;
    };
    // folded BN (+ PReLU slope) [rows][C] -> [rows][cpn] through `map` (padded index -> source channel or -1): padding planes
    // get scale 0, shift 0, slope 1, so they stay exact zeros
    auto pad_bn = [&](const float *src, int C, int rows, int cpn, const std::function<int(int)> &map) {
        std::vector<float> o((size_t)rows * cpn, 0.0f);
        for (int k = 0; k < cpn; ++k) {
            const int sc = map(k);
            for (int r = 0; r < rows; ++r)
                o[(size_t)r * cpn + k] = sc >= 0 ? src[(size_t)r * C + sc] : (r == 2 ? 1.0f : 0.0f);
        }
        return o;
    };
    auto ident = [&](int k) { return k < c ? k : -1; };
    if (!fold_bn(t, e + "b3.bn", e + "b3.act", 256, tmp.data())) return GS_ERR_INVALID;
    if (!(w = t.get(e + "classifier.conv.weight", {c, 256, 1, 1}))) return GS_ERR_INVALID;
    {
        const int rec = (3 + cp + 3) / 4 * 4;   // dec1_record<cp>
        std::vector<float> pk((size_t)256 * rec, 0.0f);   // [channel][scale, shift, alpha, w0..]
        for (int ch = 0; ch < 256; ++ch) {
            for (int j = 0; j < 3; ++j) pk[(size_t)ch * rec + j] = tmp[j * 256 + ch];
            for (int k = 0; k < c; ++k) pk[(size_t)ch * rec + 3 + k] = w[k * 256 + ch];
        }
        m.b3 = bb.push(pk.data(), pk.size());
        m.wcls = m.b3;
    }
    if (!m.encoder_only) {
        if (!fold_bn(t, "br", "", c, tmp.data(), false)) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_bn(tmp.data(), c, 2, cp, ident);
            m.br = bb.push(v.data(), v.size());
        }
        if (!(w = t.get("up_l3.0.weight", {c, c, 2, 2}))) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_deconv(w);
            m.wup3 = bb.push(v.data(), v.size());
        }
        if (!(w = t.get("level3_C.conv.weight", {c, 131, 1, 1}))) return GS_ERR_INVALID;
        {
            const int rec = (cp + 3) / 4 * 4;   // dec2_record<cp>
            std::vector<float> pk((size_t)131 * rec, 0.0f);
            for (int ch = 0; ch < 131; ++ch)
                for (int k = 0; k < c; ++k) pk[(size_t)ch * rec + k] = w[k * 131 + ch];
            m.w3c = bb.push(pk.data(), pk.size());
        }
        // combine_l2_l3: cat([level3_C out, up_l3 out]) (Model.py:373) lives in 2 * cp planes, each half padded on its own
        auto cat2 = [&](int k) { return k < cp ? (k < c ? k : -1) : (k - cp < c ? c + k - cp : -1); };
        if (!fold_bn(t, "combine_l2_l3.0.bn", "combine_l2_l3.0.act", 2 * c, tmp.data())) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_bn(tmp.data(), 2 * c, 3, 2 * cp, cat2);
            m.cbr0 = bb.push(v.data(), v.size());
        }
        if (!(w = t.get("combine_l2_l3.1.conv.weight", {c, 2 * c, 3, 3}))) return GS_ERR_INVALID;
        {
            std::vector<float> v((size_t)cp * 2 * cp * 9, 0.0f);
            for (int k = 0; k < c; ++k)
                for (int ch = 0; ch < 2 * cp; ++ch) {
                    const int sc = cat2(ch);
                    if (sc < 0) continue;
                    for (int tap = 0; tap < 9; ++tap)   // (five classes: the reference's own order; else [plane][tap][class], see Dec3Args)
                        v[cp == 5 ? ((size_t)k * 2 * cp + ch) * 9 + tap : ((size_t)ch * 9 + tap) * cp + k] = w[((size_t)k * 2 * c + sc) * 9 + tap];
                }
            m.wcc = bb.push(v.data(), v.size());
        }
        if (cp >= 12) {   // the same convolution as a conv_mfma image: [tap][2 * cp planes][cp rows], then its folded BN + PReLU
            m.wcc_mfma = bb.reserve(conv_wfloats(2 * cp, 9, 1, cp, cp, true));
            float *dst = bb.data.data() + m.wcc_mfma;
            for (int tap = 0; tap < 9; ++tap)
                for (int ch = 0; ch < 2 * cp; ++ch) {
                    const int sc = cat2(ch);
                    if (sc < 0) continue;
                    for (int k = 0; k < c; ++k)
                        dst[((size_t)tap * 2 * cp + ch) * cp + k] = w[((size_t)k * 2 * c + sc) * 9 + tap];
                }
        }
        if (!fold_bn(t, "combine_l2_l3.1.bn", "combine_l2_l3.1.act", c, tmp.data())) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_bn(tmp.data(), c, 3, cp, ident);
            m.bncc = bb.push(v.data(), v.size());
            if (m.wcc_mfma >= 0)
                std::memcpy(bb.data.data() + m.wcc_mfma + (size_t)9 * 2 * cp * cp, v.data(), sizeof(float) * 3 * cp);
        }
        if (!(w = t.get("up_l2.0.weight", {c, c, 2, 2}))) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_deconv(w);
            m.wup2 = bb.push(v.data(), v.size());
        }
        if (!fold_bn(t, "up_l2.1.bn", "up_l2.1.act", c, tmp.data())) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_bn(tmp.data(), c, 3, cp, ident);
            m.bnu2 = bb.push(v.data(), v.size());
        }
        if (!(w = t.get("conv.conv.weight", {c, 19 + c, 3, 3}))) return GS_ERR_INVALID;
        // plane `pl` of the concat buffer [comb_l2_l3 (cp planes) | output0_cat (19) | zero planes] <-> channel of the reference's
        // torch.cat([comb_l2_l3, output0_cat]) (Model.py:375)
        auto cat_ch = [&](int pl) { return pl < cp ? (pl < c ? pl : -1) : (pl - cp < 19 ? c + pl - cp : -1); };
        if (!fold_bn(t, "conv.bn", "conv.act", c, tmp.data())) return GS_ERR_INVALID;
        const std::vector<float> bn_conv = pad_bn(tmp.data(), c, 3, cp, ident);
        if (cp == 5) {
            // LDS image [tap][24 planes][c]: plane = cat channel
            const int npl = 19 + c;
            m.wconv = bb.reserve(conv_wfloats(npl, 9, 1, c, c, true));
            float *dst = bb.data.data() + m.wconv;
            for (int tap = 0; tap < 9; ++tap)
                for (int pl = 0; pl < npl; ++pl) {
                    const int wch = pl;
                    for (int co = 0; co < c; ++co)
                        dst[((size_t)tap * npl + pl) * c + co] = w[((size_t)co * (19 + c) + wch) * 9 + tap];
                }
            std::memcpy(dst + (size_t)9 * npl * c, bn_conv.data(), sizeof(float) * 3 * c);
            // row-merged form (F_XMERGE): LDS image [ty][plane][row = tx*c + o]
            m.wconv_xm = bb.reserve(conv_wfloats(npl, 3, 1, c, c, true, false, true));
            float *dx = bb.data.data() + m.wconv_xm;
            for (int ty = 0; ty < 3; ++ty)
                for (int pl = 0; pl < npl; ++pl) {
                    const int wch = pl;
                    for (int tx = 0; tx < 3; ++tx)
                        for (int co = 0; co < c; ++co)
                            dx[((size_t)ty * npl + pl) * (3 * c) + tx * c + co] = w[((size_t)co * (19 + c) + wch) * 9 + ty * 3 + tx];
                }
            std::memcpy(dx + (size_t)3 * npl * 3 * c, bn_conv.data(), sizeof(float) * 3 * c);
        } else {
            // the generic tail's conv_mfma image [tap][CINP planes][cp rows] + BN: CINP = 19 + cp rounded up to the k-step
            const int cinp = (19 + cp + 3) / 4 * 4;
            m.wconv = bb.reserve(conv_wfloats(cinp, 9, 1, cp, cp, true));
            float *dst = bb.data.data() + m.wconv;
            for (int tap = 0; tap < 9; ++tap)
                for (int pl = 0; pl < cinp; ++pl) {
                    const int wch = cat_ch(pl);
                    if (wch < 0) continue;
                    for (int co = 0; co < c; ++co)
                        dst[((size_t)tap * cinp + pl) * cp + co] = w[((size_t)co * (19 + c) + wch) * 9 + tap];
                }
            std::memcpy(dst + (size_t)9 * cinp * cp, bn_conv.data(), sizeof(float) * 3 * cp);
            m.wconv_xm = m.wconv;
        }
        if (!(w = t.get("classifier.weight", {c, c, 2, 2}))) return GS_ERR_INVALID;
        {
            const std::vector<float> v = pad_deconv(w);
            m.wclassifier = bb.push(v.data(), v.size());
        }
        if (cp == 5) {
            // dec_tail image: A operands [ty][plane group][lane] (lane = k-group * 16 + MFMA row, row = tx*c + o),
            // then BN scale / shift / alpha of conv, then classifier.weight
            const float *wc = t.get("conv.conv.weight", {c, 19 + c, 3, 3});
            if (!wc) return GS_ERR_INVALID;
            m.wtail = bb.reserve(DT_PACK_FLOATS);
            float *dt = bb.data.data() + m.wtail;
            for (int ty = 0; ty < 3; ++ty)
                for (int g = 0; g < 6; ++g)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, ch = 4 * g + (lane >> 4);
                        dt[(ty * 6 + g) * 64 + lane] =
                            rho < 3 * c ? wc[(((size_t)(rho % c) * (19 + c) + ch) * 3 + ty) * 3 + rho / c] : 0.0f;
                    }
            std::memcpy(dt + DT_A_FLOATS, bn_conv.data(), sizeof(float) * 3 * c);
            std::memcpy(dt + DT_A_FLOATS + 16, w, sizeof(float) * c * c * 4);
        }
    }
    bb.reserve(512);   // tail guard: LDS-DMA staging reads whole 1-KiB pieces
    GS_HIP(hipMalloc(reinterpret_cast<void **>(&m.dblob), bb.data.size() * sizeof(float)));
    GS_HIP(hipMemcpy(m.dblob, bb.data.data(), bb.data.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = h.release();
    return GS_OK;
}

void gs_espnet_destroy(gs_espnet *h)
{
    if (!h)
        return;
    hipDeviceSynchronize();
    for (auto &ev : h->m.events) {
        hipEventDestroy(ev.a);
        hipEventDestroy(ev.b);
    }
    free_pipeline(h->m);
    crop_pipe_destroy(h->crop_pipe);
    if (h->m.pipe_compute) hipStreamDestroy(h->m.pipe_compute);
    if (h->m.pipe_h2d) hipStreamDestroy(h->m.pipe_h2d);
    if (h->pipe_compute2) hipStreamDestroy(h->pipe_compute2);
    for (auto &l : h->lanes) {
        for (auto &ev : l->events) {
            hipEventDestroy(ev.a);
            hipEventDestroy(ev.b);
        }
        if (l->ws) hipFree(l->ws);
    }
    if (h->m.ws) hipFree(h->m.ws);
    if (h->m.prob) hipFree(h->m.prob);
    if (h->m.dblob) hipFree(h->m.dblob);
    delete h;
}

static gs_status check_shape(int n, int height, int width)
{
    GS_REQUIRE(n > 0, "batch size must be positive (got %d)", n);
    GS_REQUIRE(height >= 8 && width >= 8 && height % 8 == 0 && width % 8 == 0,
               "tile size must be a positive multiple of 8 in both dimensions (got %dx%d)", height, width);
    return GS_OK;
}

gs_status gs_espnet_reserve(gs_espnet *h, int n, int height, int width)
{
    GS_REQUIRE(h, "null handle");
    gs_status st = check_shape(n, height, width);
    if (st != GS_OK) return st;
    for (int k = 0; k <= (int)h->lanes.size(); ++k) {
        st = layout_workspace(&h->lane(k), n, height, width);
        if (st != GS_OK) return st;
    }
    return GS_OK;
}

gs_status gs_espnet_set_lanes(gs_espnet *h, int n_lanes)
{
    GS_REQUIRE(h, "null handle");
    GS_REQUIRE(n_lanes >= 1 && n_lanes <= 4, "gs_espnet_set_lanes: 1 to 4 lanes (got %d)", n_lanes);
    GS_HIP(hipDeviceSynchronize());
    while ((int)h->lanes.size() > n_lanes - 1) {
        if (h->lanes.back()->ws) hipFree(h->lanes.back()->ws);
        h->lanes.pop_back();
    }
    while ((int)h->lanes.size() < n_lanes - 1) {
        std::unique_ptr<Model> l(new Model(h->m));   // same weights (shared device blob), same configuration ...
        l->ws = nullptr;                             // ... and nothing else of the original: own workspace, no pipeline, no profile
        l->ws_bytes = 0;
        l->ws_n = l->ws_h = l->ws_w = 0;
        l->prob = nullptr;
        l->prob_bytes = 0;
        l->stages.clear();
        l->events.clear();
        l->profile = false;
        for (auto &s : l->sl)
            s = Model::Slot();
        l->pipe_compute = l->pipe_h2d = nullptr;
        l->pipe_in_bytes = l->pipe_out_bytes = 0;
        l->pipe_batch = 0;
        h->lanes.push_back(std::move(l));
    }
    return GS_OK;
}

int gs_espnet_lanes(gs_espnet *h) { return h ? 1 + (int)h->lanes.size() : 0; }

gs_status gs_espnet_forward(gs_espnet *h, const void *in, int in_format, int n, int height, int width,
                            const float mean[3], const float std[3], float *logits, uint8_t *mask,
                            unsigned long long *hist, void *hip_stream)
{
    return gs_espnet_forward_lane(h, 0, in, in_format, n, height, width, mean, std, logits, mask, hist, hip_stream);
}

gs_status gs_espnet_forward_lane(gs_espnet *h, int lane, const void *in, int in_format, int n, int height, int width,
                                 const float mean[3], const float std[3], float *logits, uint8_t *mask,
                                 unsigned long long *hist, void *hip_stream)
{
    GS_REQUIRE(h && in, "gs_espnet_forward: null handle or input");
    GS_REQUIRE(lane >= 0 && lane <= (int)h->lanes.size(), "lane %d does not exist (gs_espnet_set_lanes)", lane);
    gs_status st = check_shape(n, height, width);
    if (st != GS_OK) return st;
    GS_REQUIRE(in_format == GS_IN_U8_BGR_NHWC || in_format == GS_IN_F32_NCHW, "unknown input format %d", in_format);
    GS_REQUIRE(in_format != GS_IN_U8_BGR_NHWC || (mean && std), "uint8 input needs mean and std");
    Model &m = h->lane(lane);
    if (m.encoder_only) {
        GS_REQUIRE(logits && !mask && !hist, "ESPNet-C handle: only the 1/8-scale logits output exists");
    } else {
        GS_REQUIRE(logits || mask, "nothing to compute: logits and mask are both NULL");
        GS_REQUIRE(!hist || mask, "hist requires the mask output");
    }
    if (in_format == GS_IN_U8_BGR_NHWC)
        for (int i = 0; i < 3; ++i)
            GS_REQUIRE(std[i] != 0.0f, "std[%d] is zero", i);
    st = layout_workspace(&m, n, height, width);
    if (st != GS_OK) return st;
    return forward_any(&m, in, in_format, n, height, width, mean, std, logits, mask, hist, static_cast<hipStream_t>(hip_stream));
}

gs_status gs_espnet_read_stage(gs_espnet *h, const char *stage, int image, float *dst, size_t cap, int dims[3])
{
    GS_REQUIRE(h && stage && dims, "gs_espnet_read_stage: null argument");
    Model &m = h->m;
    auto it = m.stages.find(stage);
    GS_REQUIRE(it != m.stages.end(), "stage '%s' was not produced by the last forward", stage);
    GS_REQUIRE(image >= 0 && image < m.last_n, "image index %d out of range", image);
    const Act &a = it->second.first;
    const int C = it->second.second;
    dims[0] = C;
    dims[1] = a.H;
    dims[2] = a.W;
    const size_t count = (size_t)C * a.H * a.W;
    if (!dst)
        return GS_OK;
    GS_REQUIRE(cap >= count, "destination too small for stage '%s'", stage);
    float *tmp = nullptr;
    GS_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), count * sizeof(float)));
    hipLaunchKernelGGL(unpad_kernel, dim3(blocks_for((long long)count)), dim3(256), 0, 0, view(a), image, C, tmp);
    if (m.b2_lazy && std::string(stage) == "b2")   // planes 64..127 are kept raw in the workspace: normalise the copy
        hipLaunchKernelGGL(b2_apply_kernel, dim3(blocks_for((long long)64 * a.H * a.W)), dim3(256), 0, 0, tmp, m.dblob + m.b2, a.H * a.W, 64, 64);
    hipError_t e = hipMemcpy(dst, tmp, count * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(tmp);
    GS_HIP(e);
    return GS_OK;
}

gs_status gs_espnet_block_forward(gs_espnet *h, int kind, int level, int index, const float *in, int height, int width,
                                  float *out)
{
    GS_REQUIRE(h && in && out, "gs_espnet_block_forward: null argument");
    GS_REQUIRE((kind == 0 || kind == 1) && (level == 2 || level == 3), "kind must be 0/1 and level 2/3");
    GS_REQUIRE(height >= 1 && width >= 1, "empty input");
    Model &m = h->m;
    GS_REQUIRE(kind == 1 || (index >= 0 && index < (level == 2 ? m.p : m.q)), "no ESP block %d at level %d", index, level);
    GS_REQUIRE(kind == 0 || (height % 2 == 0 && width % 2 == 0), "the down-sampler needs an even input size");
    // the tile size whose pyramid has this block at the given size
    const int up = kind == 0 ? (level == 2 ? 4 : 8) : (level == 2 ? 2 : 4);
    gs_status st = layout_workspace(&m, 1, height * up, width * up);
    if (st != GS_OK) return st;
    const float *wb = m.dblob;
    const Act &src = kind == 0 ? (level == 2 ? m.bb[0] : m.cc[0]) : (level == 2 ? m.a0 : m.a1);
    const Act &dst = level == 2 ? (kind == 0 ? m.bb[1] : m.bb[0]) : (kind == 0 ? m.cc[1] : m.cc[0]);
    const Act &red = level == 2 ? m.r2[0] : m.r3[0];
    const int cin = kind == 0 ? (level == 2 ? 64 : 128) : (level == 2 ? 19 : 131);
    const int cout = level == 2 ? 64 : 128;
    GS_REQUIRE(src.H == height && src.W == width, "internal: workspace level size mismatch");
    const size_t nin = (size_t)cin * height * width, nout = (size_t)cout * dst.H * dst.W;
    float *tmp = nullptr;
    GS_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), (nin > nout ? nin : nout) * sizeof(float)));
    gs_status rc = GS_OK;
    hipStream_t s = nullptr;
    auto body = [&]() -> gs_status {
        GS_HIP(hipMemcpy(tmp, in, nin * sizeof(float), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pad_kernel, dim3(blocks_for((long long)nin)), dim3(256), 0, s, view(src), 0, cin, tmp);
        const PackedConv &pc = kind == 1 ? (level == 2 ? m.l2_0 : m.l3_0) : (level == 2 ? m.l2[index] : m.l3[index]);
        gs_status r;
        if (level == 2) {
            r = kind == 1 ? launch_conv_mfma<CFG_L2_C1S, F_S2PAIR | POL_L2_C1S | AGL_S2 | S2FLIP_L2>(conv_args(src, wb + pc.c1, red, nullptr, 1), m.num_cus, s)
                          : launch_conv_mfma<CFG_L2_C1, POL_L2_C1>(conv_args(src, wb + pc.c1, red, nullptr, 1), m.num_cus, s);
            if (r != GS_OK) return r;
            r = kind == 1 ? launch_vec<F_BNACT | (POL_L2_DOWN & F_ST_NT) | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(conv_args(red, wb + pc.br, dst, nullptr, 1), m.num_cus, s)
                          : launch_vec<F_BNACT | F_RES | POL_L2_ESP | AGL_L2 | SKIP_L2, CFG_L2_BR_P4>(conv_args(red, wb + pc.br, dst, &src, 1), m.num_cus, s);
        } else {
            r = kind == 1 ? launch_conv_mfma<CFG_L3_C1S, F_S2PAIR | POL_L3_C1S | AGL_S2 | S2FLIP_L3>(conv_args(src, wb + pc.c1, red, nullptr, 1), m.num_cus, s)
                          : launch_conv_mfma<CFG_L3_C1, POL_L3_C1>(conv_args(src, wb + pc.c1, red, nullptr, 1), m.num_cus, s);
            if (r != GS_OK) return r;
            if (kind == 1)
                r = launch_vec<F_BNACT | POL_L3_DOWN | AGL_L3, CFG_L3_BR>(conv_args(red, wb + pc.br, dst, nullptr, 1), m.num_cus, s);
            else if (dst.W % 4 == 0)
                r = launch_conv_mfma<CFG_L3_BR, F_BNACT | F_RES | F_VEC | POL_L3_ESP | AGL_L3>(conv_args(red, wb + pc.br, dst, &src, 1), m.num_cus, s);
            else
                r = launch_conv_mfma<CFG_L3_BR_P2, F_BNACT | F_RES | AGL_L3>(conv_args(red, wb + pc.br, dst, &src, 1), m.num_cus, s);
        }
        if (r != GS_OK) return r;
        hipLaunchKernelGGL(unpad_kernel, dim3(blocks_for((long long)nout)), dim3(256), 0, s, view(dst), 0, cout, tmp);
        GS_HIP(hipGetLastError());
        GS_HIP(hipMemcpy(out, tmp, nout * sizeof(float), hipMemcpyDeviceToHost));
        return GS_OK;
    };
    rc = body();
    hipFree(tmp);
    m.stages.clear();   // the workspace no longer holds a forward's stages
    return rc;
}

gs_status gs_espnet_profile_enable(gs_espnet *h, int on)
{
    GS_REQUIRE(h, "null handle");
    h->m.profile = on != 0;
    return GS_OK;
}

gs_status gs_espnet_profile_read(gs_espnet *h, gs_kernel_time *out, int cap, int *n_out)
{
    GS_REQUIRE(h && n_out, "null argument");
    Model &m = h->m;
    for (auto &ev : m.events) {
        GS_HIP(hipEventSynchronize(ev.b));
        float ms = 0.0f;
        GS_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        m.prof_ms[ev.k] += ms;
        m.prof_launches[ev.k] += 1;
        hipEventDestroy(ev.a);
        hipEventDestroy(ev.b);
    }
    m.events.clear();
    int k = 0;
    for (int i = 0; i < K_COUNT && out && k < cap; ++i) {
        if (!m.prof_launches[i])
            continue;
        std::snprintf(out[k].name, sizeof out[k].name, "%s", kKernelNames[i]);
        out[k].total_ms = m.prof_ms[i];
        out[k].launches = m.prof_launches[i];
        out[k].flops_per_tile = m.prof_flops[i] / (double)m.prof_launches[i];   // mean over the launches timed
        ++k;
    }
    *n_out = k;
    for (int i = 0; i < K_COUNT; ++i) {
        m.prof_ms[i] = 0;
        m.prof_launches[i] = 0;
        m.prof_flops[i] = 0;
    }
    return GS_OK;
}

gs_status gs_espnet_ensemble_forward(gs_espnet *const *models, int n_models, const void *in_u8, int n, int height,
                                     int width, const float *means, const float *stds, uint8_t *mask,
                                     unsigned long long *hist, void *hip_stream)
{
    GS_REQUIRE(models && n_models > 0 && in_u8 && means && stds && mask, "gs_espnet_ensemble_forward: null argument");
    gs_status st = check_shape(n, height, width);
    if (st != GS_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    for (int k = 0; k < n_models; ++k) {
        GS_REQUIRE(models[k] && !models[k]->m.encoder_only, "ensemble member %d is not a full ESPNet", k);
        GS_REQUIRE(models[k]->m.classes == models[0]->m.classes, "ensemble member %d has %d classes, member 0 has %d", k,
                   models[k]->m.classes, models[0]->m.classes);
        for (int i = 0; i < 3; ++i)
            GS_REQUIRE(stds[3 * k + i] != 0.0f, "ensemble member %d: std[%d] is zero", k, i);
    }
    float *prob = nullptr;
    st = ensemble_scratch(models[0], n, height, width, &prob);
    if (st != GS_OK) return st;
    // Every member's decoder tail turns its five logits into probabilities in registers and adds 1/K of them into ONE fp32
    // accumulator (first member stores, middle members add, the last adds and goes on to the argmax and the counts): the
    // logits are never written, and the accumulator is read K-1 and written K-1 times.
    for (int k = 0; k < n_models; ++k) {
        const int mode = n_models == 1 ? 4 : k == 0 ? 1 : k == n_models - 1 ? 3 : 2;
        st = espnet_forward_ex(models[k], 0, in_u8, GS_IN_U8_BGR_NHWC, n, height, width, means + 3 * k, stds + 3 * k, nullptr, mask,
                               hist, prob, mode, 1.0f / (float)n_models, s);
        if (st != GS_OK) return st;
    }
    return GS_OK;
}

gs_status gs_espnet_segment_host(gs_espnet *h, const uint8_t *tiles, int n_tiles, int height, int width,
                                 const float mean[3], const float std[3], int batch, uint8_t *masks,
                                 unsigned long long *hist)
{
    GS_REQUIRE(h && tiles && masks && mean && std, "gs_espnet_segment_host: null argument");
    GS_REQUIRE(n_tiles > 0 && batch > 0, "n_tiles and batch must be positive");
    gs_status st = check_shape(batch, height, width);
    if (st != GS_OK) return st;
    if (batch > n_tiles) batch = n_tiles;
    const size_t in_b = (size_t)height * width * 3, out_b = (size_t)height * width;
    const size_t ncl = (size_t)h->m.classes;   // hist is [n_tiles][classes]
    constexpr int NSLOT = 4;
    const int nl = h->lanes.empty() ? 1 : 2;   // batches alternate between (at most) two lanes, each on its own compute stream
    // caller buffers that are already page-locked (hipHostMalloc / hipHostRegister) are DMA'd in place;
    // pageable ones are staged through the pinned slot buffers with a host memcpy
    const bool in_pinned = host_is_pinned(tiles), out_pinned = host_is_pinned(masks) && (!hist || host_is_pinned(hist));
    Model &m = h->m;
    using Slot = Model::Slot;
    Slot *sl = m.sl;
    gs_status rc = GS_OK;
    auto fail = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == GS_OK) {
            set_error("%s failed: %s", what, hipGetErrorString(e));
            rc = GS_ERR_HIP;
        }
        return e != hipSuccess;
    };
    // Three streams.  Uploads run ahead on their own stream, which never waits for anything on the GPU.  Batch b's forward
    // AND its download (SDMA, see below) are on compute stream b % 2, in order, behind one wait for the batch's
    // upload; with two lanes the two compute streams have no dependency on each other, with one lane (a single workspace)
    // each forward also waits for the previous batch's.
    // (Round 2 first had a download stream as well, chained to the forward by an event.  HIP multiplexes streams onto a few
    // hardware queues and a queue runs its packets in order whatever stream they came from: the download's wait-for-the-
    // forward packet sat in front of later uploads, and in the rocprofv3 timeline every batch's first kernel -- and the
    // upload two batches ahead -- started only when the previous batch's download had ended: 0.4 ms lost per batch.  With
    // upload, forward and download of a batch all on one stream the two streams fell into step and copied at the same
    // time.  No packet that waits for a kernel may sit on a stream that others might queue behind.)
    // The three streams get three different priorities, because HIP keeps a separate pool of hardware queues per priority:
    // whatever other streams the process has made (torch's, the engine's lane streams), these three never share a queue
    // with each other.  (With equal priorities and two torch streams made first, the same pipeline ran at 9.2 k instead of
    // 10.4 k patches/s.)  The upload stream is the high one; the two compute streams differ only nominally.
    int prio_lo = 0, prio_hi = 0;
    fail(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), "hipDeviceGetStreamPriorityRange");   // (least, greatest): numerically lo >= hi
    const int prio_mid = (prio_lo + prio_hi) / 2;
    if (!m.pipe_compute)
        fail(hipStreamCreateWithPriority(&m.pipe_compute, hipStreamNonBlocking, prio_mid), "hipStreamCreate");
    if (!h->pipe_compute2)
        fail(hipStreamCreateWithPriority(&h->pipe_compute2, hipStreamNonBlocking, prio_lo), "hipStreamCreate");
    if (!m.pipe_h2d)
        fail(hipStreamCreateWithPriority(&m.pipe_h2d, hipStreamNonBlocking, prio_hi), "hipStreamCreate");
    if (m.pipe_in_bytes < in_b * batch || m.pipe_out_bytes < out_b * batch || m.pipe_batch < batch) {
        free_pipeline(m);
        for (int i = 0; i < NSLOT; ++i) {
            Slot &s = sl[i];
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hin), in_b * batch, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hout), out_b * batch, hipHostMallocDefault), "hipHostMalloc");
            fail(hipHostMalloc(reinterpret_cast<void **>(&s.hh), sizeof(unsigned long long) * GS_MAX_CLASSES * batch, hipHostMallocDefault), "hipHostMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.din), in_b * batch), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dout), out_b * batch), "hipMalloc");
            fail(hipMalloc(reinterpret_cast<void **>(&s.dh), sizeof(unsigned long long) * GS_MAX_CLASSES * batch), "hipMalloc");
            fail(hipEventCreateWithFlags(&s.up, hipEventDisableTiming), "hipEventCreate");
            fail(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate");
            fail(hipEventCreateWithFlags(&s.down, hipEventDisableTiming), "hipEventCreate");
        }
        m.pipe_in_bytes = in_b * batch;
        m.pipe_out_bytes = out_b * batch;
        m.pipe_batch = batch;
    }
    for (int i = 0; i < NSLOT; ++i)
        sl[i].first = -1;
    auto drain = [&](Slot &s) {   // wait for the slot's masks and hand them to the caller
        if (s.first < 0 || rc != GS_OK)
            return;
        if (fail(hipEventSynchronize(s.down), "hipEventSynchronize")) return;
        if (!out_pinned) {
            parallel_memcpy(masks + (size_t)s.first * out_b, s.hout, out_b * s.count);
            if (hist) std::memcpy(hist + (size_t)s.first * ncl, s.hh, sizeof(unsigned long long) * ncl * s.count);
        }
        s.first = -1;
    };
    int slot = 0, bi = 0;
#ifdef GS_DIAG
    // host-side timeline of the loop (GS_PIPE_TRACE=1): where does the enqueueing thread wait?
    const bool ptrace = std::getenv("GS_PIPE_TRACE") != nullptr;
    const auto pt0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (ptrace && bi < 12)
            std::fprintf(stderr, "pipe %2d %-10s %9.1f us\n", bi, what,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pt0).count());
    };
    const int pskip = std::getenv("GS_PIPE_SKIP") ? std::atoi(std::getenv("GS_PIPE_SKIP")) : 0;   // 1: no uploads, 2: no downloads (timing only)
#else
    auto stamp = [](const char *) {};
    constexpr int pskip = 0;
#endif
    for (int first = 0; first < n_tiles && rc == GS_OK; first += batch, slot = (slot + 1) % NSLOT, ++bi) {
        Slot &s = sl[slot];
        const int lane = bi % nl;
        hipStream_t compute = (bi & 1) == 0 ? m.pipe_compute : h->pipe_compute2;
        stamp("top");
        drain(s);   // the slot's previous batch must have left its pinned buffers
        stamp("drained");
        if (rc != GS_OK) break;
        const int cnt = n_tiles - first < batch ? n_tiles - first : batch;
        const uint8_t *src = tiles + (size_t)first * in_b;
        if (!in_pinned) {
            parallel_memcpy(s.hin, src, in_b * cnt);
            src = s.hin;
        }
        // (the slot's device buffers: its previous batch was drained above, i.e. computed and downloaded)
        if (!(pskip & 1) || bi < NSLOT)
            if (fail(hipMemcpyAsync(s.din, src, in_b * cnt, hipMemcpyHostToDevice, m.pipe_h2d), "H2D copy")) break;
        stamp("h2d");
        fail(hipEventRecord(s.up, m.pipe_h2d), "hipEventRecord");
        fail(hipStreamWaitEvent(compute, s.up, 0), "hipStreamWaitEvent");
        if (nl == 1 && bi > 0)   // one workspace: this forward after the previous batch's (recorded on the other stream)
            fail(hipStreamWaitEvent(compute, sl[(slot + NSLOT - 1) % NSLOT].done, 0), "hipStreamWaitEvent");
        stamp("waitev");
        gs_status st2 = gs_espnet_forward_lane(h, lane, s.din, GS_IN_U8_BGR_NHWC, cnt, height, width, mean, std, nullptr,
                                               s.dout, s.dh, compute);
        if (st2 != GS_OK) { rc = st2; break; }
        stamp("forward");
        fail(hipEventRecord(s.done, compute), "hipEventRecord");
        stamp("waitdone");
        // Results -> pinned host memory through hipMemcpy2DAsync: on this stack a plain hipMemcpyAsync(DeviceToHost) runs as a
        // blit KERNEL (__amd_rocclr_copyBuffer) and the rectangular copy goes to the SDMA engine (rocprofv3: a memory-copy
        // record instead of a kernel, the same 53 GB/s).  A copy kernel of any size costs the pipeline its whole 0.33 ms:
        // the level-2 / level-3 launches need every CU's full register file (one workgroup per CU), so every CU that holds a
        // copy wave sends a launch into a second round -- measured 3.29 ms per batch with a 48-workgroup copy kernel, 2.96
        // without the download, 2.93 without any copy.
        uint8_t *hdst = out_pinned ? masks + (size_t)first * out_b : s.hout;
        unsigned long long *hhdst = (out_pinned && hist) ? hist + (size_t)first * ncl : s.hh;
        if (!(pskip & 2))
            fail(hipMemcpy2DAsync(hdst, out_b, s.dout, out_b, out_b, cnt, hipMemcpyDeviceToHost, compute), "D2H copy");
        if (hist || !out_pinned)
            fail(hipMemcpy2DAsync(hhdst, sizeof(unsigned long long) * ncl * cnt, s.dh, sizeof(unsigned long long) * ncl * cnt,
                                  sizeof(unsigned long long) * ncl * cnt, 1, hipMemcpyDeviceToHost, compute), "D2H copy");
        fail(hipEventRecord(s.down, compute), "hipEventRecord");
        stamp("d2h");
        s.first = first;
        s.count = cnt;
    }
    for (int k = 0; k < NSLOT; ++k)
        drain(sl[(slot + k) % NSLOT]);   // oldest first
    if (rc != GS_OK) {
        hipDeviceSynchronize();
        return rc;
    }
    return gs_device_fault_check();   // (every batch has been drained: a few microseconds)
}

}  // extern "C"
