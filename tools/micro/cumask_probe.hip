// Which compute units does a CU-masked stream (hipExtStreamCreateWithCUMask) run on, per XCD?  Prints, for a few mask patterns, the
// number of workgroups that landed on every XCD (HW_REG_XCC_ID) and the count of distinct (XCD, SE, CU) places used.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
        unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
    // stay resident a little so that blocks spread over every permitted CU
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}

static void run(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        std::printf("%s: stream creation failed\n", name);
        return;
    }
    const int nb = 2048;
    unsigned *d;
    hipMalloc(&d, nb * 2 * sizeof(unsigned));
    hipMemsetAsync(d, 0xff, nb * 2 * sizeof(unsigned), s);
    hipLaunchKernelGGL(where, dim3(nb), dim3(512), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc;
    std::set<unsigned long long> places;
    for (int b = 0; b < nb; ++b) {
        const unsigned xcc = h[b * 2] & 0xf, hw = h[b * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc]++;
        places.insert(((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu);
    }
    std::printf("%-28s places %3zu  blocks per XCD:", name, places.size());
    for (auto &kv : per_xcc) std::printf(" %u:%d", kv.first, kv.second);
    std::printf("\n");
    hipFree(d);
    hipStreamDestroy(s);
}

__global__ void spin(long long cycles, unsigned *sink)
{
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1000) sink[0] = 1;
}

// do kernels on two streams with DISJOINT CU masks run at the same time?  (each: one 512-thread workgroup per permitted CU, ~1 ms)
static void concurrency(int words)
{
    std::vector<uint32_t> low(words, 0), high(words, 0);
    for (int i = 0; i < words / 2; ++i) low[i] = 0xffffffffu;
    for (int i = words / 2; i < words; ++i) high[i] = 0xffffffffu;
    hipStream_t a, b, pa, pb;
    hipExtStreamCreateWithCUMask(&a, (uint32_t)words, low.data());
    hipExtStreamCreateWithCUMask(&b, (uint32_t)words, high.data());
    hipStreamCreateWithFlags(&pa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&pb, hipStreamNonBlocking);
    auto timeit = [&](const char *name, hipStream_t s0, hipStream_t s1, int blocks, size_t lds) {
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), lds, s0, 2400000ll, nullptr);
            if (s1) hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), lds, s1, 2400000ll, nullptr);
            hipStreamSynchronize(s0);
            if (s1) hipStreamSynchronize(s1);
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%-58s %.2f ms\n", name, ms);
    };
    timeit("one masked stream, 128 blocks", a, nullptr, 128, 0);
    timeit("two masked streams (disjoint halves), 128 blocks each", a, b, 128, 0);
    timeit("two plain streams, 128 blocks each", pa, pb, 128, 0);
    timeit("two masked streams, 128 blocks each, 128 KB LDS per block", a, b, 128, 128 * 1024);
    timeit("two plain streams, 256 blocks each, 128 KB LDS per block", pa, pb, 256, 128 * 1024);
}

// CHAINS of kernels (as a forward is) on two CU-masked streams: does the second stream's chain run beside the first's, or after it?
static void chains(int words)
{
    std::vector<uint32_t> low(words, 0), high(words, 0);
    for (int i = 0; i < words / 2; ++i) low[i] = 0xffffffffu;
    for (int i = words / 2; i < words; ++i) high[i] = 0xffffffffu;
    const int NS = 8;
    hipStream_t sa[NS], sb[NS], plain[2];
    for (int k = 0; k < NS; ++k) {
        (void)hipExtStreamCreateWithCUMask(&sa[k], (uint32_t)words, low.data());
        (void)hipExtStreamCreateWithCUMask(&sb[k], (uint32_t)words, high.data());
    }
    (void)hipStreamCreateWithFlags(&plain[0], hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&plain[1], hipStreamNonBlocking);
    auto run2 = [&](const char *name, hipStream_t s0, hipStream_t s1, int blocks) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s0);
            for (int k = 0; k < 10; ++k) {
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, s0, 480000ll, nullptr);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, s1, 480000ll, nullptr);
            }
            (void)hipStreamSynchronize(s1);
            (void)hipEventRecord(e1, s0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        std::printf("%-64s %.2f ms  (10 x 0.2 ms per stream: 2.0 side by side, 4.0 one after the other)\n", name, best);
    };
    run2("chains on two plain streams, 128 blocks", plain[0], plain[1], 128);
    for (int k = 0; k < NS; ++k) {
        char name[96];
        std::snprintf(name, sizeof name, "chains on masked streams low[0] + high[%d], 128 blocks", k);
        run2(name, sa[0], sb[k], 128);
    }
    run2("chains on masked streams low[3] + high[5], 128 blocks", sa[3], sb[5], 128);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> all(words, 0xffffffffu), low(words, 0), high(words, 0), even(words, 0x55555555u), odd(words, 0xaaaaaaaau);
    std::vector<uint32_t> evenpairs(words, 0x33333333u), nib(words, 0x0f0f0f0fu), bytes(words, 0x00ff00ffu), halves(words, 0x0000ffffu);
    for (int i = 0; i < words / 2; ++i) low[i] = 0xffffffffu;
    for (int i = words / 2; i < words; ++i) high[i] = 0xffffffffu;
    run("all", all);
    run("low half of the bits", low);
    run("high half of the bits", high);
    run("even bits", even);
    run("odd bits", odd);
    run("pairs 0x33333333", evenpairs);
    run("nibbles 0x0f0f0f0f", nib);
    run("bytes 0x00ff00ff", bytes);
    run("halfwords 0x0000ffff", halves);
    std::vector<uint32_t> one(words, 0);
    one[0] = 0xff;
    run("bits 0-7 only", one);
    one[0] = 0xff00;
    run("bits 8-15 only", one);
    hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    concurrency(words);
    chains(words);
    return 0;
}
