// Host-side polygon extraction for the per-crop JSON export: what the reference gets from
// cv2.findContours(thresh, RETR_LIST, CHAIN_APPROX_SIMPLE) + cv2.arcLength + cv2.approxPolyDP in
// module/common/boundary_extractor.py:33-47.  OpenCV is not installed, so both are restated from the published
// algorithms: Suzuki & Abe (1985) border following with 8-connectivity, and Ramer-Douglas-Peucker on a closed curve.
// Checked point for point (order, start point, direction, vertex lists) against oracle/contour_oracle.py, a second restatement
// of OpenCV 4.3's functions written independently of this file (tests/test_contour_oracle.py); cv2 itself cannot run here.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "gs_internal.h"

namespace {

// 8-neighbourhood in OpenCV's chain-code order: 0 = east, counter-clockwise on screen
// (y grows downwards): E, NE, N, NW, W, SW, S, SE
const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Pt {
    int x, y;
};

// follow one border starting at (x,y) whose "previous" background neighbour is in direction `from_dir`
// (4 = west for outer borders, 0 = east for hole borders); marks visited pixels with +-nbd like the paper.
void follow(std::vector<int> &f, int W, int x0, int y0, int from_dir, int nbd, std::vector<Pt> &out, bool simple)
{
    auto at = [&](int x, int y) -> int & { return f[(size_t)y * W + x]; };
    // step 3.1: clockwise search around (x0,y0) starting from the background neighbour
    int s = from_dir, s_end = from_dir;
    do {
        s = (s - 1) & 7;   // clockwise
        if (at(x0 + DX[s], y0 + DY[s]) != 0)
            break;
    } while (s != s_end);
    if (s == s_end && at(x0 + DX[s], y0 + DY[s]) == 0) {   // isolated pixel
        at(x0, y0) = -nbd;
        out.push_back({x0, y0});
        return;
    }
    int x2 = x0 + DX[s], y2 = y0 + DY[s];   // (i1,j1) of the paper; becomes the "previous" point
    int x3 = x0, y3 = y0;                   // current point
    const int xl = x2, yl = y2;             // to recognise the end of the border
    // OpenCV starts with "the direction we arrived by" = opposite of the first clockwise hit (icvFetchContour: prev_s = s ^ 4),
    // so a start pixel in the middle of a straight run (a hole whose left wall is a SW-NE diagonal through it) is not a
    // CHAIN_APPROX_SIMPLE vertex; found by oracle/contour_oracle.py (round 4), -1 here wrote one point too many
    int prev_dir = s ^ 4;
    for (;;) {
        // step 3.3: counter-clockwise search around the current point starting after the previous one
        int sd = 0;
        for (int k = 0; k < 8; ++k)
            if (x3 + DX[k] == x2 && y3 + DY[k] == y2)
                sd = k;
        bool east_bg = false;   // was the east neighbour examined and found background?
        int nd = sd;
        for (int k = 1; k <= 8; ++k) {
            nd = (sd + k) & 7;
            if (at(x3 + DX[nd], y3 + DY[nd]) != 0)
                break;
            if (nd == 0)
                east_bg = true;
        }
        // step 3.4: label the current pixel
        if (east_bg)
            at(x3, y3) = -nbd;
        else if (at(x3, y3) == 1)
            at(x3, y3) = nbd;
        // emit the point (CHAIN_APPROX_SIMPLE keeps only direction changes)
        if (!simple || nd != prev_dir)
            out.push_back({x3, y3});
        prev_dir = nd;
        const int x4 = x3 + DX[nd], y4 = y3 + DY[nd];
        // step 3.5: back at the start with the same successor -> border closed
        if (x4 == x0 && y4 == y0 && x3 == xl && y3 == yl)
            break;
        x2 = x3;
        y2 = y3;
        x3 = x4;
        y3 = y4;
    }
}

}  // namespace

extern "C" {

// img: uint8 [h,w], non-zero = foreground.  Writes every border (outer and hole, RETR_LIST) as a run of (x,y)
// pairs into `points`, with offsets[k]..offsets[k+1] delimiting contour k (offsets has n_contours+1 entries).
// Returns GS_ERR_NOMEM (and the needed sizes in n_points / n_contours) when a capacity is too small.
gs_status gs_find_contours(const uint8_t *img, int h, int w, int simple, int *points, int cap_points, int *offsets,
                           int cap_contours, int *n_contours, int *n_points)
{
    GS_REQUIRE(img && n_contours && n_points, "gs_find_contours: null pointer");
    GS_REQUIRE(h > 0 && w > 0, "gs_find_contours: bad size");
    const int W = w + 2, H = h + 2;   // one pixel of background all round, as OpenCV does
    std::vector<int> f((size_t)W * H, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            f[(size_t)(y + 1) * W + x + 1] = img[(size_t)y * w + x] ? 1 : 0;
    std::vector<std::vector<Pt>> found;
    int nbd = 1;
    for (int y = 1; y <= h; ++y) {
        for (int x = 1; x <= w; ++x) {
            const int v = f[(size_t)y * W + x];
            if (v == 0)
                continue;
            int from = -1;
            if (v == 1 && f[(size_t)y * W + x - 1] == 0)
                from = 4;   // outer border: background to the west
            else if (v >= 1 && f[(size_t)y * W + x + 1] == 0)
                from = 0;   // hole border: background to the east
            if (from < 0)
                continue;
            ++nbd;
            found.emplace_back();
            follow(f, W, x, y, from, nbd, found.back(), simple != 0);
        }
    }
    long long total = 0;
    for (auto &c : found)
        total += (long long)c.size();
    *n_contours = (int)found.size();
    *n_points = (int)total;
    if (!points || !offsets || total > cap_points || (int)found.size() + 1 > cap_contours) {
        if (points || offsets) {
            gs::set_error("gs_find_contours: need room for %lld points and %zu contours", total, found.size());
            return GS_ERR_NOMEM;
        }
        return GS_OK;   // size query
    }
    int at = 0, k = 0;
    // OpenCV hands contours back last-found first
    for (auto it = found.rbegin(); it != found.rend(); ++it, ++k) {
        offsets[k] = at;
        for (const Pt &p : *it) {
            points[2 * at] = p.x - 1;
            points[2 * at + 1] = p.y - 1;
            ++at;
        }
    }
    offsets[k] = at;
    return GS_OK;
}

// cv2.arcLength(curve, closed=True): perimeter of a closed polygon given as n (x,y) pairs.  As cv::arcLength computes it
// (shapedescr.cpp): every segment in float32 (points converted to Point2f, `float dx, dy`, float sqrt), summed in a double,
// the closing segment last -> first taken FIRST.  The low bits matter: epsilon = 0.003 * this decides approxPolyDP's ties.
double gs_arc_length_closed(const int *xy, int n)
{
    if (n <= 1)
        return 0.0;
    double s = 0.0;
    float px = (float)xy[2 * (n - 1)], py = (float)xy[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        const float qx = (float)xy[2 * i], qy = (float)xy[2 * i + 1];
        const float dx = qx - px, dy = qy - py;
        const float a = dx * dx, b = dy * dy;      // separate roundings: no fused multiply-add across them
        s += (double)std::sqrt(a + b);
        px = qx;
        py = qy;
    }
    return s;
}

// cv2.approxPolyDP(curve, epsilon, closed=True).  cv2 is absent from this stack, so this restates the published algorithm of
// the version the reference pins (opencv-python==4.3.0.38, docker/requirements.txt:5; modules/imgproc/src/approx.cpp,
// approxPolyDP_<int>) step by step, because its start-point rule and its clean-up pass decide WHICH vertices a polygon in the
// labelme JSON has (boundary_extractor.py:43-47):
//   1. three hops "to the point farthest from the current one", starting at point 0: the last hop's two ends are the initial cut;
//   2. Ramer-Douglas-Peucker on an explicit stack of index ranges that may wrap (left range first, so vertices come out in
//      contour order starting at the cut), distance test |cross| ^ 2 <= eps^2 * |chord|^2, ties to the first farthest point;
//   3. a clean-up pass over the result that drops a vertex lying within sqrt(0.5) * eps of the chord of its neighbours when
//      that chord is neither horizontal nor vertical and the vertex does not fold back.
// Equality with the independent restatement in oracle/contour_oracle.py is tested in tests/test_contour_oracle.py.
int gs_approx_poly_closed(const int *xy, int n, double epsilon, int *out)
{
    if (n <= 0)
        return 0;
    const int count0 = n;
    int count = n;
    auto srcx = [&](int i) { return xy[2 * i]; };
    auto srcy = [&](int i) { return xy[2 * i + 1]; };
    struct Range { int start, end; };
    std::vector<Range> stack;
    std::vector<Pt> dst((size_t)n + 1);
    int new_count = 0;
    double eps = epsilon * epsilon;
    Range slice{0, 0}, right{0, 0};
    int pos = 0;
    bool le_eps = false;
    Pt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    auto read_pt = [&](Pt &p, int &at) {
        p = Pt{srcx(at), srcy(at)};
        if (++at >= count) at = 0;
    };
    // 1. approximately the two farthest points of the contour
    right.start = 0;
    for (int it = 0; it < 3; ++it) {
        double max_dist = 0;
        pos = (pos + right.start) % count;
        read_pt(start_pt, pos);
        for (int jj = 1; jj < count; ++jj) {
            read_pt(pt, pos);
            const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) {
                max_dist = dist;
                right.start = jj;
            }
        }
        le_eps = max_dist <= eps;
    }
    // 2. the stack
    if (!le_eps) {
        right.end = slice.start = pos % count;
        slice.end = right.start = (right.start + slice.start) % count;
        stack.push_back(right);
        stack.push_back(slice);
    } else {
        dst[new_count++] = start_pt;
    }
    // 3. the recursion, unrolled
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = Pt{srcx(slice.end), srcy(slice.end)};
        pos = slice.start;
        read_pt(start_pt, pos);
        if (pos != slice.end) {
            double max_dist = 0;
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            const bool same = dx == 0 && dy == 0;   // (OpenCV asserts here; a border that passes one pixel twice can get here)
            while (pos != slice.end) {
                read_pt(pt, pos);
                const double dist = same ? std::sqrt((double)(pt.x - start_pt.x) * (pt.x - start_pt.x) + (double)(pt.y - start_pt.y) * (pt.y - start_pt.y))
                                         : std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right.start = (pos + count - 1) % count;
                }
            }
            le_eps = same ? max_dist * max_dist <= eps : max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = Pt{srcx(slice.start), srcy(slice.start)};
        }
        if (le_eps) {
            dst[new_count++] = start_pt;
        } else {
            right.end = slice.end;
            slice.end = right.start;
            stack.push_back(right);
            stack.push_back(slice);
        }
    }
    // last stage: remove extra points on the [almost] straight lines
    count = new_count;
    auto read_dst = [&](Pt &p, int &at) {
        p = dst[at];
        if (++at >= count) at = 0;
    };
    if (count > 0) {
        pos = count - 1;
        read_dst(start_pt, pos);
        int wpos = pos;
        read_dst(pt, pos);
        for (int i = 0; i < count && new_count > 2; ++i) {
            read_dst(end_pt, pos);
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            const double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
            const double inner = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
            if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
                --new_count;
                dst[wpos] = start_pt = end_pt;
                if (++wpos >= count) wpos = 0;
                read_dst(pt, pos);
                ++i;
                continue;
            }
            dst[wpos] = start_pt = pt;
            if (++wpos >= count) wpos = 0;
            pt = end_pt;
        }
    }
    (void)count0;
    for (int i = 0; i < new_count; ++i) {
        out[2 * i] = dst[i].x;
        out[2 * i + 1] = dst[i].y;
    }
    return new_count;
}

}  // extern "C"
