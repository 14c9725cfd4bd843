/*
 * glomseg.h -- C ABI of libglomseg.so: the MI355X (gfx950) replacement for the per-patch
 * inference hot path of jinseikenai/glomeruli_segmentation.
 *
 * The reference has no native code; its "operator API" for this path is
 * torch.nn.Module.__call__ on module/espnet/test/Model.py's ESPNet / ESPNet_Encoder
 * (called at module/espnet/test/VisualizeResults_iou.py:123) and tf.Session.run on a frozen
 * detector graph (module/faster-rcnn/detect_glomus_test.py:350-352).  These entry points are
 * what a ctypes / cffi / pybind stub on the reference side binds instead.
 *
 * Conventions: plain pointers and sizes, no torch types.  Every function returns gs_status and
 * never throws; gs_last_error() gives the message of the calling thread's last failure.
 * Device pointers must belong to the HIP device that is current at the call.  A handle is bound
 * to that device and is not thread-safe; work is stream-ordered on the hipStream_t passed in
 * (NULL = the legacy default stream).  A handle has one activation workspace per LANE (one lane unless
 * gs_espnet_set_lanes asked for more): work submitted for the same lane on different streams must be ordered by the
 * caller.  The host pipelines (gs_espnet_segment_host, gs_espnet_segment_crops_host, gs_detector_detect_host) run on
 * streams of their own, use lanes 0 and 1, and expect every earlier call on the handle to have completed.
 */
#ifndef GLOMSEG_H
#define GLOMSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gs_status {
    GS_OK = 0,
    GS_ERR_INVALID = 1,     /* bad argument / missing or mis-shaped weight tensor */
    GS_ERR_HIP = 2,         /* a HIP runtime call failed */
    GS_ERR_NOMEM = 3,
    GS_ERR_UNSUPPORTED = 4, /* configuration the kernels are not built for */
    GS_ERR_NODEVICE = 5,    /* no gfx950 device visible */
    GS_ERR_DEVICE_FAULT = 6 /* a kernel reported, through the device-side fault word, that it went on without data it was waiting for */
} gs_status;

const char *gs_last_error(void);
/* ABI version, bumped on any signature change. */
int gs_abi_version(void);
/* How the library was compiled: GS_BUILD_DIAG = a -DGS_DIAG experiment build (timing variants that return wrong results by
 * construction can be switched on through the environment); the product library returns 0 and reads no environment. */
#define GS_BUILD_DIAG 1
int gs_build_flags(void);
/* Synchronises the current device, then reads and clears the device-side fault word: GS_ERR_DEVICE_FAULT when a kernel of an
 * earlier call gave up a bounded wait (the decoder tail's strip-boundary exchange) and went on with stale values -- the masks,
 * counts and logits of calls since the previous check are then not to be used.  The host pipelines (gs_espnet_segment_host,
 * gs_espnet_segment_crops_host) make this check themselves before they return; callers of the stream-ordered entries
 * (gs_espnet_forward*, gs_espnet_segment_crops, gs_espnet_ensemble_*) call it where they synchronise.  No reference
 * counterpart (cuDNN has no such exchange); ABI 5. */
gs_status gs_device_fault_check(void);

/* ------------------------------------------------------------------ weights
 * One entry per state_dict tensor, named exactly as in models/espnet_fold*.pth
 * (e.g. "encoder.level3.7.d16.conv.weight").  `offset` counts floats into `blob`.
 * Replaces: torch.load + load_state_dict, VisualizeResults_iou.py:272,279. */
typedef struct gs_layer_desc {
    char name[96];
    int64_t offset;
    int32_t ndim;
    int32_t shape[4];
} gs_layer_desc;

typedef struct gs_espnet gs_espnet;

/* Class counts: ESPNet(classes, p, q) takes any `classes` (Model.py:311; the constructor's default is 20, the shipped networks
 * have 5, VisualizeResults_iou.py:315 passes --classes through).  This library builds 2 <= classes <= GS_MAX_CLASSES; anything
 * else is refused by gs_espnet_create with GS_ERR_UNSUPPORTED (class maps are uint8; the per-class counters and the padded
 * decoder instantiations stop at 20).  Five classes run the fused decoder tail; other counts a two-kernel tail that is
 * slower per tile.  Every `hist` below is [n][classes] and every ensemble accumulator [n][classes][h][w]. */
#define GS_MAX_CLASSES 20

/* Build a model handle on the current HIP device from a host fp32 weight blob.
 * classes/p/q as ESPNet(classes, p, q) (Model.py:311); encoder_only != 0 builds
 * ESPNet_Encoder (ESPNet-C, Model.py:246) whose table uses keys without the "encoder." prefix. */
gs_status gs_espnet_create(const float *blob, const gs_layer_desc *table, int n_layers, int classes,
                           int p, int q, int encoder_only, gs_espnet **out);
void gs_espnet_destroy(gs_espnet *h);

/* Pre-size the activation workspace for batches up to n tiles of h x w (both multiples of 8).
 * Optional: forward grows the workspace on demand (which allocates, so call this first when the
 * forward is to be captured in a hipGraph). */
gs_status gs_espnet_reserve(gs_espnet *h, int n, int height, int width);

typedef enum gs_input_format {
    GS_IN_U8_BGR_NHWC = 0, /* what cv2.imread yields (VisualizeResults_iou.py:103); normalised on the fly */
    GS_IN_F32_NCHW = 1     /* already-normalised tensor, the nn.Module.forward argument (Model.py:341) */
} gs_input_format;

/* One pass of the hot path over a batch of n tiles resident in device memory.
 * Replaces VisualizeResults_iou.py:107-128 + :151-155 (normalise -> model(x) -> argmax -> counts).
 *   in        device, format per in_format; mean/std (host, 3 floats, BGR) used only for U8 input
 *   logits    device fp32 [n,classes,h,w] or NULL   (ESPNet-C: [n,classes,h/8,w/8])
 *   mask      device uint8 [n,h,w] or NULL          (first maximum wins, torch semantics)
 *   hist      device uint64 [n,classes] per-class pixel counts or NULL (requires the mask pass)
 * At least one of logits/mask must be non-NULL. */
gs_status gs_espnet_forward(gs_espnet *h, const void *in, int in_format, int n, int height, int width,
                            const float mean[3], const float std[3], float *logits, uint8_t *mask,
                            unsigned long long *hist, void *hip_stream);

/* LANES: a handle can hold up to four activation workspaces ("lanes"; weights are shared) so that several batches are in
 * flight at once, each on its own HIP stream: the kernels of ONE forward run strictly one after the other and the big ones
 * fill every CU, but the tail of one batch's kernel overlaps the head of another batch's (measured on MI355X, batch 32:
 * +5.5 % throughput with two lanes, nothing more with three).  gs_espnet_forward is lane 0.  Work on one lane is ordered by
 * the caller exactly as for a handle without lanes; different lanes are independent.  gs_espnet_set_lanes synchronises the
 * device and may be called again to shrink or grow; gs_espnet_segment_host alternates its batches between two lanes when
 * the handle has them. */
gs_status gs_espnet_set_lanes(gs_espnet *h, int n_lanes);
int gs_espnet_lanes(gs_espnet *h);
gs_status gs_espnet_forward_lane(gs_espnet *h, int lane, const void *in, int in_format, int n, int height, int width,
                                 const float mean[3], const float std[3], float *logits, uint8_t *mask,
                                 unsigned long long *hist, void *hip_stream);

/* Host-to-host batch pipeline: n_tiles uint8 BGR tiles in (pageable or pinned) host memory are
 * uploaded on a stream of their own (page-locked buffers in place, pageable ones through pinned staging slots), `batch`
 * tiles per step; each batch's forward and its download (SDMA) run in order on one of two compute streams; up to four
 * batches are queued ahead.  Masks (and optional per-tile histograms) come back to host memory; the call returns when
 * all of them have.  Replaces the whole loop
 * VisualizeResults_iou.py:100-156 for a list of equal-size tiles. */
gs_status gs_espnet_segment_host(gs_espnet *h, const uint8_t *tiles, int n_tiles, int height, int width,
                                 const float mean[3], const float std[3], int batch, uint8_t *masks,
                                 unsigned long long *hist);

/* Crop stage either side of the forward for crops that are not already network-sized
 * (VisualizeResults_iou.py:107-116 and :129).  gs_crop_preprocess: uint8 BGR crop [h,w,3] (device) ->
 * normalise at crop resolution ((x-mean)/std), cv2.resize INTER_LINEAR sampling (half-pixel centres, no
 * antialias, horizontal pass first) to out_h x out_w, then /255, written as fp32 CHW -- the tensor
 * gs_espnet_forward(GS_IN_F32_NCHW) takes.  gs_mask_resize_nearest: cv2.resize INTER_NEAREST of a uint8
 * class map back to the crop size (src = min(floor(dst * src_size / dst_size), src_size - 1)). */
gs_status gs_crop_preprocess(const uint8_t *crop_bgr, int h, int w, const float mean[3], const float std[3],
                             int out_h, int out_w, float *out_chw, void *hip_stream);
gs_status gs_mask_resize_nearest(const uint8_t *mask, int h, int w, int out_h, int out_w, uint8_t *out,
                                 void *hip_stream);

/* ------------------------------------------------------------------ batched variable-size crops
 * The loop body of VisualizeResults_iou.py:100-156 for the crops a slide really produces (make_seg_data.py:357-361: every
 * merged box at its own size): normalise at crop resolution -> cv2.resize INTER_LINEAR to the network size -> /255 ->
 * forward -> argmax -> cv2.resize INTER_NEAREST back to the crop size -> per-class counts of THAT map (:151-155), for a
 * whole batch per launch: ONE descriptor-table kernel resamples every crop of the batch into the network's input, the mask
 * comes straight from the decoder tail (no logits), ONE kernel resizes all masks back and counts, ONE kernel pastes them
 * into the 1/ds slide map (eval_wsi_segmentation.py:311-312, np.max: overlapping crops of one launch meet through a
 * compare-and-swap).  The arithmetic per pixel is that of gs_crop_preprocess / gs_mask_resize_nearest / gs_wsi_paste_max. */
typedef struct gs_crop_desc {
    int64_t in_off;  /* byte offset of the crop's uint8 BGR [h,w,3] pixels in the packed input buffer */
    int64_t out_off; /* byte offset of its uint8 class map [h,w] in the packed output buffer (multiple of 4) */
    int32_t h, w;
    int32_t x1, y1;  /* level-0 origin of the crop on the slide (used by the paste only) */
} gs_crop_desc;

typedef struct gs_paste_target {
    uint8_t *slide_map;   /* device uint8 [map_h,map_w], accumulated into.  4-byte aligned, allocation padded to a multiple of 4
                           * bytes: overlapping crops meet through 32-bit compare-and-swap on the word around a byte */
    int32_t map_h, map_w, ds;
    const int *sx_lut;    /* device tables of gs_wsi_paste_max_lut, or both NULL for the regular grid */
    const int *sy_lut;
} gs_paste_target;

#define GS_MAX_CROPS_PER_CALL 64

/* Device-resident form, stream-ordered on lane `lane`: n <= GS_MAX_CROPS_PER_CALL crops packed in `packed_in` (device), their
 * descriptors in HOST memory (they travel as kernel arguments).  Any of net_masks (device uint8 [n,net_h,net_w], the map the
 * reference scores at network resolution, :202), packed_out (device, crop-size maps at out_off), hist (device uint64 [n,classes],
 * counts of the crop-size maps) and paste may be NULL, but not all of them. */
gs_status gs_espnet_segment_crops(gs_espnet *h, int lane, const uint8_t *packed_in, const gs_crop_desc *descs, int n,
                                  const float mean[3], const float std[3], int net_h, int net_w, uint8_t *net_masks,
                                  uint8_t *packed_out, unsigned long long *hist, const gs_paste_target *paste, void *hip_stream);

/* The same for an ensemble (cfg 5 definition: mean over members of softmax, each member with its own mean/std): every member
 * resamples the crops with its own normalisation (the reference normalises BEFORE it resizes) and adds its probabilities in
 * the decoder tail; the last member's tail writes the masks. */
gs_status gs_espnet_ensemble_segment_crops(gs_espnet *const *models, int n_models, const uint8_t *packed_in, const gs_crop_desc *descs,
                                           int n, const float *means, const float *stds, int net_h, int net_w, uint8_t *net_masks,
                                           uint8_t *packed_out, unsigned long long *hist, const gs_paste_target *paste,
                                           void *hip_stream);

/* Optional overlay output of the host pipeline below (VisualizeResults_iou.py:139-146): every crop's class map coloured with the
 * palette (rows RGB as in the reference's table, written [b, g, r]) and blended over the crop as
 * cv2.addWeighted(crop, wa, colour, wb, 0) = saturate_cast<uchar>(round(crop * wa + colour * wb)), each product and the sum
 * rounded to fp32 on its own (no fused multiply-add), round-half-to-even: bit for bit what numpy computes for that expression
 * (imageops.add_weighted, the checker of the GPU tests).  NOT pinned against cv2 itself -- OpenCV is not installable in the build
 * image, and an FMA3 build of its 8u addWeighted may fuse the second product into the sum, which differs from two separately
 * rounded products on exact ties only (at weights 0.4 / 0.6 a tie needs crop * 0.4f + colour * 0.6f to land on x.5 in fp32).
 * out_bgr[i] is host
 * uint8 [heights[i], widths[i], 3]; page-locked buffers laid out like the packed input (every crop at the 256-byte-aligned
 * offset behind its batch's first one, in ONE allocation) are written a batch per DMA, anything else crop by crop. */
#define GS_MAX_PALETTE 64
typedef struct gs_crop_overlay {
    const uint8_t *palette_rgb; /* host, n_colours * 3 bytes; a class beyond the table is black */
    int32_t n_colours;          /* 1 .. GS_MAX_PALETTE */
    float wa, wb;
    uint8_t *const *out_bgr;
} gs_crop_overlay;

/* Host-to-host pipeline over a list of crops of any sizes (the whole loop :100-156): crops[i] is uint8 BGR [heights[i],
 * widths[i],3] in host memory (page-locked buffers are DMA'd in place, pageable ones staged through pinned slots by a few
 * threads); up to `batch` (<= GS_MAX_CROPS_PER_CALL) crops per step; uploads on a stream of their own, batches alternate
 * between two compute streams (and two lanes when the handle has them), results come back by SDMA.  Outputs, each optional:
 * masks[i] (host uint8 [heights[i],widths[i]]), net_masks (host uint8 [n_crops,net_h,net_w]), hist (host uint64 [n_crops,classes],
 * counts of the crop-size maps), paste + x1/y1 (level-0 origins), overlay (above).  n_models == 1 is the plain model; > 1 the ensemble.
 * A list shorter than four full batches is cut into a small first batch (a seventh of the list, at least eight crops: its
 * upload is the pipeline's fill) and three equal ones.
 * Page-locked masks[] that lie in ONE block, every map in a 256-byte-aligned slot right behind the previous one, are
 * written a batch per DMA -- the up to 255 padding bytes behind a map are written too (unspecified values); any other
 * layout is written map by map, exactly heights[i] * widths[i] bytes each.
 * Returns when everything has arrived. */
gs_status gs_espnet_segment_crops_host(gs_espnet *const *models, int n_models, const uint8_t *const *crops, const int *heights,
                                       const int *widths, int n_crops, const float *means, const float *stds, int net_h,
                                       int net_w, int batch, uint8_t *const *masks, uint8_t *net_masks,
                                       unsigned long long *hist, const gs_paste_target *paste, const int *x1, const int *y1,
                                       const gs_crop_overlay *overlay /* or NULL */);

/* The batch plan gs_espnet_segment_crops_host follows, as a host-only function (no device work; csrc/crop_plan.h): batch b
 * holds crops [starts[b], starts[b+1]); *n_batches batches, starts gets *n_batches + 1 entries (cap counts ints; with
 * starts == NULL only the count is reported).  No batch holds more than min(batch, GS_MAX_CROPS_PER_CALL) crops or -- beyond its
 * first crop -- more than 256 MiB of crop pixels. */
gs_status gs_plan_crop_batches(const int *heights, const int *widths, int n_crops, int batch, int *starts, int cap, int *n_batches);
/* 1 when the bytes [p, p + bytes) lie inside ONE page-locked host allocation (hipHostMalloc / hipHostRegister / torch
 * pin_memory): such buffers are DMA'd in place by the host pipelines, and a batch's maps leave in one copy.  0 otherwise. */
int gs_host_block_is_pinned(const void *p, size_t bytes);

/* WSI compositor (the consumer of the gathered masks; eval_wsi_segmentation.py:243-316,215-241,359-394).
 * The slide-level class map lives at 1/ds of level 0 (ds = 8 in the reference): pixel (X,Y) holds the
 * class at level-0 pixel (ds*X, ds*Y), i.e. what INTER_NEAREST of a full 2400-px window yields (:229).
 * gs_wsi_paste_max: max-composite one crop's class map (uint8 [h,w], level-0 resolution, top-left at
 * level-0 (x1,y1)) into the slide map (uint8 [map_h,map_w]) -- np.max of window and crop (:311-312).
 * Calls on one stream are ordered, so overlapping crops need no atomics.
 * gs_overlay_classmap: palette colouring + cv2.addWeighted(region, wa, colour, wb) on a BGR uint8 image
 * (:236-240; palette rows are RGB as in the reference table, 25 entries).
 * gs_confusion_u8: iouEval.fast_hist (IOUEval.py:19-21): hist[classes*gt + pred] += 1 for gt < classes. */
gs_status gs_wsi_paste_max(uint8_t *slide_map, int map_h, int map_w, int ds, const uint8_t *crop_mask, int h, int w,
                           int x1, int y1, void *hip_stream);
/* The reference's own window walk (eval_wsi_segmentation.py:372-393): sx_lut[X] / sy_lut[Y] (device, map_w / map_h ints)
 * give the level-0 column / row that map column X / row Y shows, -1 where the reference writes nothing.  Inside full
 * 2400-px windows that is ds*X; in the partial windows at the right / bottom edge the INTER_NEAREST step of :229 is
 * (window size) / int(window size / ds); windows the reference skips (`ymax > slide_width`, :386) stay empty.
 * composite.reference_window_luts builds the tables. */
gs_status gs_wsi_paste_max_lut(uint8_t *slide_map, int map_h, int map_w, int ds, const int *sx_lut, const int *sy_lut,
                               const uint8_t *crop_mask, int h, int w, int x1, int y1, void *hip_stream);
gs_status gs_overlay_classmap(const uint8_t *region_bgr, const uint8_t *class_map, int h, int w,
                              const uint8_t *palette_rgb /*[n_colours*3] device*/, int n_colours, float wa, float wb,
                              uint8_t *out_bgr, void *hip_stream);
gs_status gs_confusion_u8(const uint8_t *pred, const uint8_t *gt, long long n, int classes,
                          unsigned long long *hist /*[classes*classes] device, accumulated into*/, void *hip_stream);

/* Host-side polygon extraction for the per-crop labelme JSON (boundary_extractor.py:33-47): borders of a binary
 * uint8 image (Suzuki-Abe border following, every outer and hole border = RETR_LIST; simple != 0 keeps only the
 * points where the direction changes = CHAIN_APPROX_SIMPLE), perimeter and Ramer-Douglas-Peucker simplification of
 * a closed curve.  Pure host code (no device work).  gs_find_contours with points == NULL only reports the sizes. */
gs_status gs_find_contours(const uint8_t *img, int h, int w, int simple, int *points /*xy pairs*/, int cap_points,
                           int *offsets /*n_contours+1*/, int cap_contours, int *n_contours, int *n_points);
double gs_arc_length_closed(const int *xy, int n);
int gs_approx_poly_closed(const int *xy, int n, double epsilon, int *out_xy);

/* 5-fold style ensemble (BASELINE cfg 5; definition in DESIGN.md): probability = mean over
 * models of softmax(logits_k), each model with its own mean/std; writes argmax mask. */
gs_status gs_espnet_ensemble_forward(gs_espnet *const *models, int n_models, const void *in_u8, int n,
                                     int height, int width, const float *means /*[n_models*3]*/,
                                     const float *stds /*[n_models*3]*/, uint8_t *mask,
                                     unsigned long long *hist, void *hip_stream);

/* Debug/test hook: copy one named intermediate activation of the LAST forward (image index
 * `image`) to host memory as contiguous CHW fp32.  Names follow tests/golden stage names
 * ("b1","level2_0","level2.0",...,"b2","level3_0","level3.7","up_l3","combine_t","up_l2",...).
 * dims receives {C,H,W}.  cap counts floats. */
gs_status gs_espnet_read_stage(gs_espnet *h, const char *stage, int image, float *dst, size_t cap,
                               int dims[3]);

/* Debug/test hook: run ONE block of the trunk on a caller-supplied input (host, contiguous CHW fp32) with the
 * handle's weights and return its output (host, CHW fp32): the reference's single-module known-answer tests
 * (DilatedParllelResidualBlockB / DownSamplerB called alone, Model.py:187-214,144-160).
 *   kind 0: ESP block `index` of level `level` (2 or 3): in [64|128, h, w] -> out [64|128, h, w]
 *   kind 1: the level's DownSamplerB (level2_0 / level3_0): in [19|131, h, w] -> out [64|128, h/2, w/2]
 * h, w: input size (any size >= 1 for kind 0; even for kind 1).  Runs the plain (unfused) kernels of the block. */
gs_status gs_espnet_block_forward(gs_espnet *h, int kind, int level, int index, const float *in, int height, int width,
                                  float *out);

/* Per-kernel timing with HIP events recorded on the launch stream.  While enabled every kernel of
 * gs_espnet_forward is bracketed by an event pair; gs_espnet_profile_read synchronises and
 * accumulates.  Used by bench.py for the roofline line. */
typedef struct gs_kernel_time {
    char name[64];
    double total_ms;
    int64_t launches;
    double flops_per_tile; /* algorithmic (unpadded) FLOPs this kernel performs per launch per tile */
} gs_kernel_time;
gs_status gs_espnet_profile_enable(gs_espnet *h, int on);
gs_status gs_espnet_profile_read(gs_espnet *h, gs_kernel_time *out, int cap, int *n_out);

/* ------------------------------------------------------------------ detector-side primitives
 * The detector network is an external TF1 frozen graph that is not in the reference
 * (detect_glomus_test.py:419-427); these are the device ops such a graph is made of (gs_detector_* below assembles
 * them).  Parity for them is unpinned. */

/* conv2d, NHWC fp32, weights [kh,kw,cin,cout] (TF layout), SAME-style explicit padding, + bias, optional ReLU. */
gs_status gs_conv2d_nhwc(const float *in, int n, int h, int w, int cin, const float *weight, int kh, int kw,
                         int cout, const float *bias_or_null, int stride, int pad, int relu, float *out,
                         void *hip_stream);
/* tf.image.crop_and_resize (the ROI pooling of TF-OD Faster R-CNN): boxes normalised [y1,x1,y2,x2],
 * bilinear, extrapolation value 0.  feat NHWC fp32 -> out [n_boxes, crop, crop, c]. */
gs_status gs_roialign(const float *feat, int n, int h, int w, int c, const float *boxes, const int *box_image,
                      int n_boxes, int crop, float *out, void *hip_stream);
/* Greedy IoU non-maximum suppression (tf.image.non_max_suppression): boxes [k,4] yxyx, scores [k];
 * keep (device int32 [max_out]) receives indices in descending score order, *n_keep the count. */
gs_status gs_nms(const float *boxes, const float *scores, int k, float iou_threshold, float score_threshold,
                 int max_out, int *keep, int *n_keep, void *hip_stream);

/* ------------------------------------------------------------------ assembled detector
 * A two-stage (Faster R-CNN shaped) detector forward behind the tensor contract of the reference's detect_box
 * (detect_glomus_test.py:349-352 sess.run, tensors :443-450): uint8 RGB windows in, detection_boxes /
 * detection_scores / detection_classes / num_detections out.  The reference's own network is an external frozen graph
 * (not in the reference), so the weights here are the caller's: `table` names backbone.c1..c6, rpn.conv, rpn.head,
 * head.h1, head.h2, head.fc (+ ".weight" [kh,kw,cin,cout] / ".bias" [cout]); glomeruli_segmentation_amd/detector.py lists
 * the shapes and makes seeded synthetic ones.  Parity with the reference's graph is unpinned.
 * Graph: 2/255 x - 1 -> space-to-depth(2) -> 6 conv layers (stride 16) -> RPN 3x3 + 1x1 heads over 12 grid anchors per
 * cell -> box decode (10,10,5,5) + clip -> top-1024 -> NMS 0.7 -> 300 proposals -> crop_and_resize 14x14 -> max-pool 2 ->
 * box head -> softmax + decode -> NMS 0.6 -> up to 100 detections, scores descending, zero padded. */
typedef struct gs_detector gs_detector;
gs_status gs_detector_create(const float *blob, const gs_layer_desc *table, int n_layers, gs_detector **out);
void gs_detector_destroy(gs_detector *h);
int gs_detector_max_detections(void); /* D = 100 */
int gs_detector_num_proposals(void);  /* 300 */
gs_status gs_detector_set_thresholds(gs_detector *h, float rpn_nms_iou, float det_nms_iou, float det_score_threshold);
/* images_rgb: device uint8 [n,height,width,3].  boxes [n,D,4] normalised [ymin,xmin,ymax,xmax], scores [n,D] descending,
 * classes [n,D] (1.0 = glomerulus, 0 = padding), num [n]: device fp32.  dbg_*: optional device taps (NULL to skip):
 * features [n,hf,wf,256], rpn [n,hf,wf,72] (24 class logits + 48 box deltas per cell), proposals [n,300,4] in pixels,
 * head [n*300,6] (2 class logits + 4 box deltas). */
gs_status gs_detector_forward(gs_detector *h, const uint8_t *images_rgb, int n, int height, int width, float *boxes, float *scores,
                              float *classes, float *num, float *dbg_features, float *dbg_rpn, float *dbg_proposals,
                              float *dbg_head, void *hip_stream);

/* Host-to-host scan of one slide's windows (the loop of detect_glomus_test.py:270-284 around sess.run): n equal-size uint8
 * RGB windows in host memory (pageable ones staged through pinned slots by a few threads, page-locked ones DMA'd in place),
 * `batch` windows per forward, uploads on a stream of their own one batch ahead of the forward.  Outputs in host memory:
 * boxes [n,D,4], scores [n,D], classes [n,D], num [n] as gs_detector_forward writes them. */
gs_status gs_detector_detect_host(gs_detector *h, const uint8_t *const *windows, int n, int height, int width, int batch,
                                  float *boxes, float *scores, float *classes, float *num);

#ifdef __cplusplus
}
#endif
#endif /* GLOMSEG_H */
