// Host-side polygon extraction for the per-crop JSON export: what the reference gets from
// cv2.findContours(thresh, RETR_LIST, CHAIN_APPROX_SIMPLE) + cv2.arcLength + cv2.approxPolyDP in
// module/common/boundary_extractor.py:33-47.  OpenCV is not installed, so both are restated from the published
// algorithms: Suzuki & Abe (1985) border following with 8-connectivity, and Ramer-Douglas-Peucker on a closed curve.
// Parity with cv2 itself is unpinned; the tests pin geometric invariants instead.
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
    int prev_dir = -1;
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

double perp_dist2(const Pt &p, const Pt &a, const Pt &b)
{
    const double dx = b.x - a.x, dy = b.y - a.y;
    const double len2 = dx * dx + dy * dy;
    if (len2 == 0.0) {
        const double ex = p.x - a.x, ey = p.y - a.y;
        return ex * ex + ey * ey;
    }
    const double cr = dx * (p.y - a.y) - dy * (p.x - a.x);
    return cr * cr / len2;
}

void rdp(const std::vector<Pt> &pts, int i0, int i1, double eps2, std::vector<char> &keep)
{
    // iterative Ramer-Douglas-Peucker on the open chain pts[i0..i1]
    std::vector<std::pair<int, int>> st{{i0, i1}};
    while (!st.empty()) {
        auto [a, b] = st.back();
        st.pop_back();
        double best = -1.0;
        int bi = -1;
        for (int i = a + 1; i < b; ++i) {
            const double d = perp_dist2(pts[i], pts[a], pts[b]);
            if (d > best) {
                best = d;
                bi = i;
            }
        }
        if (bi >= 0 && best > eps2) {
            keep[bi] = 1;
            st.push_back({a, bi});
            st.push_back({bi, b});
        }
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

// cv2.arcLength(curve, closed=True): perimeter of a closed polygon given as n (x,y) pairs.
double gs_arc_length_closed(const int *xy, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        s += std::sqrt((double)(xy[2 * j] - xy[2 * i]) * (xy[2 * j] - xy[2 * i]) +
                       (double)(xy[2 * j + 1] - xy[2 * i + 1]) * (xy[2 * j + 1] - xy[2 * i + 1]));
    }
    return s;
}

// cv2.approxPolyDP(curve, epsilon, closed=True): Ramer-Douglas-Peucker.  The closed curve is cut at point 0 and at the
// point farthest from it; keep[] order follows the input order.  Returns the number of kept points (written to out).
int gs_approx_poly_closed(const int *xy, int n, double epsilon, int *out)
{
    if (n <= 2) {
        for (int i = 0; i < 2 * n; ++i)
            out[i] = xy[i];
        return n;
    }
    std::vector<Pt> pts(n + 1);
    for (int i = 0; i < n; ++i)
        pts[i] = {xy[2 * i], xy[2 * i + 1]};
    pts[n] = pts[0];
    int far = 0;
    double best = -1.0;
    for (int i = 1; i < n; ++i) {
        const double dx = pts[i].x - pts[0].x, dy = pts[i].y - pts[0].y;
        if (dx * dx + dy * dy > best) {
            best = dx * dx + dy * dy;
            far = i;
        }
    }
    std::vector<char> keep(n + 1, 0);
    keep[0] = keep[far] = keep[n] = 1;
    rdp(pts, 0, far, epsilon * epsilon, keep);
    rdp(pts, far, n, epsilon * epsilon, keep);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (keep[i]) {
            out[2 * m] = pts[i].x;
            out[2 * m + 1] = pts[i].y;
            ++m;
        }
    return m;
}

}  // extern "C"
