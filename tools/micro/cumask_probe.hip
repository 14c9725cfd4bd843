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
    return 0;
}
