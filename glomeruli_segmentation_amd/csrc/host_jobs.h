// Host-only staging helpers (no HIP in here: csrc/host_copy.h adds the one HIP query): threaded jobs and a threaded memcpy
// between caller memory and pinned staging buffers.  Compiled on their own under -fsanitize=address,undefined and
// -fsanitize=thread by tests/test_sanitizers.py.
#pragma once
#include <cstddef>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace gs {

// Run job(0..n_jobs-1) on up to `max_threads` threads (the calling thread included).  A std::thread that cannot be created
// (cgroup thread limit: std::system_error) never crosses the extern "C" boundary: the jobs it would have taken run on the
// calling thread instead.
static inline void parallel_jobs(int n_jobs, unsigned max_threads, const std::function<void(int)> &job)
{
    unsigned nt = max_threads;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if ((int)nt > n_jobs) nt = (unsigned)n_jobs;
    if (nt <= 1) {
        for (int i = 0; i < n_jobs; ++i) job(i);
        return;
    }
    std::vector<std::thread> th;
    unsigned started = 1;   // thread 0 is the caller
    for (unsigned t = 1; t < nt; ++t) {
        try {
            th.emplace_back([=, &job] {
                for (int i = (int)t; i < n_jobs; i += (int)nt) job(i);
            });
            ++started;
        } catch (...) {
            break;
        }
    }
    for (int i = 0; i < n_jobs; i += (int)nt) job(i);
    // strides of the threads that could not be created
    for (unsigned t = started; t < nt; ++t)
        for (int i = (int)t; i < n_jobs; i += (int)nt) job(i);
    for (auto &t : th)
        t.join();
}

// memcpy of a staging buffer on a few threads: one core copies ~10 GB/s, and a 50 MB batch of pageable tiles copied by the
// enqueueing thread alone (4-5 ms) is slower than the GPU's 2.9 ms per batch
static inline void parallel_memcpy(void *dst, const void *src, size_t bytes)
{
    constexpr size_t kMinPerThread = 4u << 20;
    unsigned nt = (unsigned)(bytes / kMinPerThread);
    if (nt > 4) nt = 4;
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t chunk = (bytes / nt + 63) / 64 * 64;
    parallel_jobs((int)nt, nt, [&](int i) {
        const size_t lo = (size_t)i * chunk, hi = (unsigned)i + 1 == nt ? bytes : ((size_t)i + 1) * chunk;
        if (lo < hi && lo < bytes)
            std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, (hi < bytes ? hi : bytes) - lo);
    });
}

}  // namespace gs
