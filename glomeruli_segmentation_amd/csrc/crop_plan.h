// Batch planner of gs_espnet_segment_crops_host (csrc/crops.hip): which crops of a list (any sizes; one slide's merged boxes,
// module/faster-rcnn/make_seg_data.py:357-361) go into which batch of the loop that replaces
// module/espnet/test/VisualizeResults_iou.py:100-156, and where each crop and each crop-size map sits in a batch's packed
// staging buffers.  Host-only and HIP-free on purpose: tests/test_sanitizers.py compiles it on its own under
// -fsanitize=address,undefined / -fsanitize=thread, and gs_plan_crop_batches (include/glomseg.h) exposes it to CPU tests.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/glomseg.h"

// lists shorter than four full batches -- 0: four equal batches; 1: a small first batch (a seventh of the list, at least 8 crops: its
// upload is the pipeline's fill), the rest in three (56 crops: 7.6 -> 6.9 ms, profiles/README.md round 4)
#ifndef CFG_SHORT_LIST_SPLIT
#define CFG_SHORT_LIST_SPLIT 1
#endif

namespace gs {

constexpr size_t kCropSlotAlign = 256;          // every crop / map starts on a 256-byte boundary of its packed buffer
constexpr size_t kCropBatchBytes = 256u << 20;  // crop pixels per batch (one oversize crop still forms a batch of its own), so
                                                // that a list of very large crops does not ask for gigabytes of pinned staging per slot
static inline size_t crop_slot(size_t bytes) { return (bytes + kCropSlotAlign - 1) / kCropSlotAlign * kCropSlotAlign; }

struct CropBatchPlan {
    std::vector<int> starts;   // batch b holds crops [starts[b], starts[b+1]); starts.back() == n_crops
    size_t need_in = 0;        // largest packed input of a batch (bytes)
    size_t need_out = 0;       // largest packed crop-size-map output of a batch
    int max_count = 0;         // crops in the largest batch: never more than min(batch, max_per_call)
    int first_batch = 0, batch = 0;   // the two batch sizes the list was cut with
};

// `batch` is the caller's (> 0); no batch ever exceeds min(batch, max_per_call, n_crops).
// A short list -- one slide's crops (56 on the example slide, 7 per rank on eight GPUs) -- is cut into four batches rather than
// one or two full ones: the first batch's upload and the last one's download are exposed, and the forward keeps ~90 % of its
// full-batch rate down to 14-16 tiles (profiles/r04_latency.json).  The first batch is the smallest (nothing overlaps its
// upload): 56 crops as 8 + 16 + 16 + 16 take 6.8-7.0 ms, as 4 x 14: 7.6, as 32 + 24: 10.1.  When the three equal batches
// behind a seventh-of-the-list first one would not fit the caller's batch size (lists of 3.5 to 4 batches), the list is cut
// into four equal ones instead (round 4 let them grow past the descriptor table: ADVICE r4).
static inline CropBatchPlan plan_crop_batches(const int *heights, const int *widths, int n_crops, int batch, int max_per_call)
{
    CropBatchPlan p;
    if (n_crops <= 0 || batch <= 0 || max_per_call <= 0)
        return p;
    const int batch0 = std::min(std::min(batch, max_per_call), n_crops);
    int first_batch = batch0, rest = batch0;
    if (n_crops < 4 * batch0) {
        const int floor8 = std::min(batch0, 8);
        const int equal4 = std::max(floor8, (n_crops + 3) / 4);   // <= batch0: n_crops < 4 * batch0
        first_batch = rest = equal4;
        if (CFG_SHORT_LIST_SPLIT && n_crops >= 32) {
            const int f = std::max(floor8, (n_crops + 6) / 7);
            const int r = std::max(floor8, (n_crops - f + 2) / 3);
            if (f <= batch0 && r <= batch0) {
                first_batch = f;
                rest = r;
            }
        }
    }
    p.first_batch = first_batch;
    p.batch = rest;
    for (int first = 0; first < n_crops;) {
        size_t bi = 0, bo = 0;
        int i = first;
        const int cap = first == 0 ? first_batch : rest;
        while (i < n_crops && i - first < cap) {
            const size_t px = (size_t)heights[i] * (size_t)widths[i];
            const size_t ci = crop_slot(px * 3);
            if (i > first && bi + ci > kCropBatchBytes)
                break;
            bi += ci;
            bo += crop_slot(px);
            ++i;
        }
        p.starts.push_back(first);
        p.need_in = std::max(p.need_in, bi);
        p.need_out = std::max(p.need_out, bo);
        p.max_count = std::max(p.max_count, i - first);
        first = i;
    }
    p.starts.push_back(n_crops);
    return p;
}

// descriptors of one batch: crop j at the packed offsets its predecessors leave; returns the packed sizes
static inline void fill_crop_descs(const int *heights, const int *widths, const int *x1, const int *y1, int first, int count,
                                   gs_crop_desc *descs, size_t *in_bytes, size_t *out_bytes)
{
    size_t oi = 0, oo = 0;
    for (int j = 0; j < count; ++j) {
        gs_crop_desc &d = descs[j];
        d.h = heights[first + j];
        d.w = widths[first + j];
        d.x1 = x1 ? x1[first + j] : 0;
        d.y1 = y1 ? y1[first + j] : 0;
        d.in_off = (int64_t)oi;
        d.out_off = (int64_t)oo;
        oi += crop_slot((size_t)d.h * (size_t)d.w * 3);
        oo += crop_slot((size_t)d.h * (size_t)d.w);
    }
    *in_bytes = oi;
    *out_bytes = oo;
}

}  // namespace gs
