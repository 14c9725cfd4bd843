// Sanitizer driver (tests/test_sanitizers.py) for the host-only native code behind gs_espnet_segment_crops_host: the batch
// planner and packed-slot layout (csrc/crop_plan.h) and the threaded staging copies (csrc/host_jobs.h).  Replays what the
// pipeline does with them on random crop lists -- plan, lay out every batch, copy every crop into the packed staging buffer on a few
// threads, copy every map back out -- with plain heap buffers in place of the pinned ones, so that AddressSanitizer sees every
// byte the product would touch and ThreadSanitizer every thread it would start.  Built with g++ only: no HIP in here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "crop_plan.h"
#include "host_jobs.h"

using namespace gs;

static int fail(const char *what, int a, int b)
{
    std::fprintf(stderr, "FAILED: %s (%d, %d)\n", what, a, b);
    return 1;
}

static int run_list(const std::vector<int> &hs, const std::vector<int> &ws, int batch, bool copy)
{
    const int n = (int)hs.size();
    const CropBatchPlan plan = plan_crop_batches(hs.data(), ws.data(), n, batch, GS_MAX_CROPS_PER_CALL);
    if (n <= 0 || batch <= 0)
        return plan.starts.empty() ? 0 : fail("a plan for an empty list", n, batch);
    if (plan.starts.empty() || plan.starts.front() != 0 || plan.starts.back() != n)
        return fail("plan does not cover the list", n, batch);
    if (plan.max_count > batch || plan.max_count > GS_MAX_CROPS_PER_CALL)
        return fail("batch larger than asked for", plan.max_count, batch);
    std::vector<unsigned char> hin(copy ? plan.need_in : 0), hout(copy ? plan.need_out : 0);
    for (size_t b = 0; b + 1 < plan.starts.size(); ++b) {
        const int first = plan.starts[b], cnt = plan.starts[b + 1] - first;
        if (cnt < 1 || cnt > plan.max_count)
            return fail("batch size", cnt, plan.max_count);
        gs_crop_desc tab[GS_MAX_CROPS_PER_CALL];   // the by-value table of the kernels: cnt must fit
        size_t oi = 0, oo = 0;
        fill_crop_descs(hs.data(), ws.data(), nullptr, nullptr, first, cnt, tab, &oi, &oo);
        if (oi > plan.need_in || oo > plan.need_out)
            return fail("batch needs more staging than the plan reserved", (int)b, cnt);
        for (int j = 0; j < cnt; ++j)
            if (tab[j].in_off % 256 || tab[j].out_off % 256 || (size_t)tab[j].in_off + (size_t)tab[j].h * tab[j].w * 3 > oi ||
                (size_t)tab[j].out_off + (size_t)tab[j].h * tab[j].w > oo)
                return fail("slot layout", (int)b, j);
        if (!copy)
            continue;
        // uploads: every crop into its packed slot on a few threads (crops.hip: parallel_jobs(cnt, 8 / 4, memcpy))
        std::vector<std::vector<unsigned char>> crops(cnt), maps(cnt);
        for (int j = 0; j < cnt; ++j) {
            crops[j].assign((size_t)tab[j].h * tab[j].w * 3, (unsigned char)(first + j));
            maps[j].resize((size_t)tab[j].h * tab[j].w);
        }
        parallel_jobs(cnt, b == 0 ? 8 : 4, [&](int j) { std::memcpy(hin.data() + tab[j].in_off, crops[j].data(), crops[j].size()); });
        for (int j = 0; j < cnt; ++j)
            if (hin[tab[j].in_off] != (unsigned char)(first + j) || hin[tab[j].in_off + crops[j].size() - 1] != (unsigned char)(first + j))
                return fail("staged crop", (int)b, j);
        // downloads: every map out of its packed slot
        std::memset(hout.data(), 7, oo);
        parallel_jobs(cnt, 4, [&](int j) { std::memcpy(maps[j].data(), hout.data() + tab[j].out_off, maps[j].size()); });
        for (int j = 0; j < cnt; ++j)
            if (maps[j].front() != 7 || maps[j].back() != 7)
                return fail("map copy", (int)b, j);
    }
    return 0;
}

int main(int argc, char **argv)
{
    const int lists = argc > 1 ? std::atoi(argv[1]) : 1000;
    std::mt19937 rng(12345);
    int rc = 0;
    // the list lengths the review named, at several batch sizes
    const int named[] = {0, 1, 7, 8, 63, 64, 65, 114, 127, 226, 230, 255, 256, 257};
    for (int n : named)
        for (int batch : {1, 7, 8, 32, 57, 64, 65, 100}) {
            std::vector<int> hs(n), ws(n);
            for (int i = 0; i < n; ++i) {
                hs[i] = 1 + (int)(rng() % 40);
                ws[i] = 1 + (int)(rng() % 40);
            }
            rc |= run_list(hs, ws, batch, true);
        }
    for (int t = 0; t < lists && !rc; ++t) {
        const int n = (int)(rng() % 300), batch = 1 + (int)(rng() % 70);
        std::vector<int> hs(n), ws(n);
        for (int i = 0; i < n; ++i) {
            hs[i] = 1 + (int)(rng() % 90);
            ws[i] = 1 + (int)(rng() % 120);
        }
        rc |= run_list(hs, ws, batch, t % 4 == 0);
    }
    // a 4000 x 7000 crop (84 MB of pixels) among small ones: the byte cap cuts the batch, the copies still fit
    {
        std::vector<int> hs = {30, 4000, 20, 4000, 4000, 4000, 10}, ws = {40, 7000, 20, 7000, 7000, 7000, 10};
        rc |= run_list(hs, ws, 32, true);
    }
    // the threaded memcpy around its thread-count thresholds (4 MiB per thread, at most four threads)
    for (size_t bytes : {(size_t)0, (size_t)1, (size_t)(4u << 20) - 1, (size_t)(8u << 20), (size_t)(8u << 20) + 63, (size_t)(13u << 20) + 5, (size_t)(40u << 20) + 1}) {
        std::vector<unsigned char> a(bytes + 1, 3), b(bytes + 1, 0);
        a[bytes] = 9;
        parallel_memcpy(b.data(), a.data(), bytes);
        for (size_t i = 0; i < bytes; ++i)
            if (b[i] != 3) {
                rc |= fail("parallel_memcpy", (int)(bytes >> 20), (int)i);
                break;
            }
        if (b[bytes] != 0)
            rc |= fail("parallel_memcpy wrote past the end", (int)(bytes >> 20), 0);
    }
    // parallel_jobs: every job exactly once, whatever the thread count
    for (int nj : {0, 1, 3, 8, 65})
        for (unsigned nt : {0u, 1u, 2u, 8u, 64u}) {
            std::vector<int> hit(nj, 0);
            parallel_jobs(nj, nt, [&](int j) { hit[j] += 1; });
            for (int j = 0; j < nj; ++j)
                if (hit[j] != 1)
                    rc |= fail("parallel_jobs", nj, (int)nt);
        }
    std::printf("host side ok: %d\n", rc == 0);
    return rc;
}
