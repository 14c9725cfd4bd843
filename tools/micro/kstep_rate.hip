// Microbenchmark (round 6): the matrix-pipe share a k-loop of conv_mfma.h's SHAPE sustains -- P independent 32x32x2 (or 16x16x4)
// accumulators per wave, per k-step ONE LDS read (the A operand, shared by the P matrix instructions) and ONE buffer load of
// P floats per lane (the B operands) through a ring of R steps -- with one or two waves per SIMD, against what the branch kernels
// measure (paired k-loops 0.76 of the pipe, a lone wave 0.65 at level 3; profiles/README.md round 6).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/kstep_rate.hip -o /tmp/kstep_rate && /tmp/kstep_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <bool BIG>
struct Acc {
    using t = f16v;
    static __device__ __forceinline__ t run(float a, float b, t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};
template <>
struct Acc<false> {
    using t = f4;
    static __device__ __forceinline__ t run(float a, float b, t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// MODE bit 0: LDS read of the A operand per step, bit 1: buffer load of the B operands per step, bit 2: SALU filler (8 scalar
// instructions per step, the address arithmetic of the real loop)
template <bool BIG, int P, int R, int MODE>
__global__ void __launch_bounds__(512) k(const float *in, float *out, int iters, int bytes, int sstep)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x)
        lds[i] = 0.001f * i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, bytes, 0x00020000);
    typename Acc<BIG>::t acc[P];
    for (int p = 0; p < P; ++p) acc[p] = (typename Acc<BIG>::t)(0.0f);
    float a[R], b[R][P];
    for (int g = 0; g < R; ++g) {
        a[g] = lds[(lane + g * 64) & 8191];
        for (int p = 0; p < P; ++p) b[g][p] = 1.0f + p + g;
    }
    const int voff = ((blockIdx.x * 64 + lane) * 4 * P) % (bytes - 65536);
    int soff = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < R; ++g) {
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = Acc<BIG>::run(a[g], b[g][p], acc[p]);
            if (MODE & 2) {
                if (P == 2) {
                    u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
                    unsigned e0 = v[0], e1 = v[1];
                    b[g][0] = __builtin_bit_cast(float, e0);
                    b[g][1 % P] = __builtin_bit_cast(float, e1);
                } else if (P == 4) {
                    u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
                    unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                    b[g][0] = __builtin_bit_cast(float, e0);
                    b[g][1 % P] = __builtin_bit_cast(float, e1);
                    b[g][2 % P] = __builtin_bit_cast(float, e2);
                    b[g][3 % P] = __builtin_bit_cast(float, e3);
                } else {
                    b[g][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
                }
            }
            if (MODE & 1)
                a[g] = lds[(lane + it * 64 + g * 16) & 8191];
            if ((MODE & 8) && g % 4 == 3 && R % 4 == 0) {   // one 16-byte LDS read per FOUR steps (the A operands of steps g+1 .. g+4)
                const f4 v = *reinterpret_cast<const f4 *>(&lds[((lane + it * 64 + g * 16) * 4) & 8188]);
                a[(g + 1) % R] = v[0]; a[(g + 2) % R] = v[1]; a[(g + 3) % R] = v[2]; a[(g + 4) % R] = v[3];
            }
            if ((MODE & 16) && g % 2 == 1 && R % 2 == 0) {  // one 8-byte LDS read per TWO steps
                const float2 v = *reinterpret_cast<const float2 *>(&lds[((lane + it * 64 + g * 16) * 2) & 8190]);
                a[(g + 1) % R] = v.x; a[(g + 2) % R] = v.y;
            }
            if ((MODE & 32)) {   // the A operand by a buffer load (weights from L2 through the ring, F_A_GLOBAL)
                a[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, (soff >> 2) & 0x3fffc, 0));
            }
            soff = (soff + sstep) & 0x3ffff;
            if (MODE & 4) {
                int z = soff;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    z = (z << 1) ^ (z + q);
                soff ^= (z & 1) << 20 >> 20 & 0;   // (keeps the scalar chain alive)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int p = 0; p < P; ++p)
        for (int r = 0; r < (BIG ? 16 : 4); ++r) s += acc[p][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <bool BIG, int P, int R, int MODE>
void run(const char *name, int waves_per_cu, const float *in, float *out, int bytes)
{
    const int iters = 40000 / R;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = waves_per_cu >= 8 ? 512 : waves_per_cu * 64;
    const int blocks = 256 * (waves_per_cu >= 8 ? waves_per_cu / 8 : 1);
    k<BIG, P, R, MODE><<<blocks, threads>>>(in, out, 10, bytes, 2048);
    hipEventRecord(e0);
    k<BIG, P, R, MODE><<<blocks, threads>>>(in, out, iters, bytes, 2048);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * R * P * (BIG ? 4096.0 : 2048.0);
    printf("%-44s waves/CU %2d  %7.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)\n", name, waves_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main()
{
    const int bytes = 1 << 20;
    float *in, *out;
    hipMalloc(&in, bytes); hipMalloc(&out, 4 << 20);
    hipMemset(in, 0, bytes);
    for (int w : {4, 8}) {
        run<true, 2, 13, 0>("32x32x2 P=2 pure", w, in, out, bytes);
        run<true, 2, 13, 1>("32x32x2 P=2 + ds_read", w, in, out, bytes);
        run<true, 2, 13, 2>("32x32x2 P=2 + x2 load, ring 13", w, in, out, bytes);
        run<true, 2, 13, 3>("32x32x2 P=2 + ds_read + x2 load, ring 13", w, in, out, bytes);
        run<true, 2, 13, 7>("32x32x2 P=2 + ds_read + x2 load + salu, r 13", w, in, out, bytes);
        run<true, 2, 3, 3>("32x32x2 P=2 + ds_read + x2 load, ring 3", w, in, out, bytes);
        run<true, 2, 39, 3>("32x32x2 P=2 + ds_read + x2 load, ring 39", w, in, out, bytes);
        run<true, 2, 12, 10>("32x32x2 P=2 + ds_read_b128/4 steps + x2 load", w, in, out, bytes);
        run<true, 2, 12, 18>("32x32x2 P=2 + ds_read_b64/2 steps + x2 load", w, in, out, bytes);
        run<true, 2, 13, 34>("32x32x2 P=2 + A by buffer load + x2 load", w, in, out, bytes);
        run<false, 4, 12, 10>("16x16x4 P=4 + ds_read_b128/4 steps + x4 load", w, in, out, bytes);
        run<false, 4, 27, 34>("16x16x4 P=4 + A by buffer load + x4 load", w, in, out, bytes);
        run<true, 4, 9, 3>("32x32x2 P=4 + ds_read + x4 load, ring 9", w, in, out, bytes);
        run<true, 1, 13, 3>("32x32x2 P=1 + ds_read + x1 load, ring 13", w, in, out, bytes);
        run<false, 4, 27, 0>("16x16x4 P=4 pure", w, in, out, bytes);
        run<false, 4, 27, 3>("16x16x4 P=4 + ds_read + x4 load, ring 27", w, in, out, bytes);
        run<false, 4, 9, 3>("16x16x4 P=4 + ds_read + x4 load, ring 9", w, in, out, bytes);
    }
    return 0;
}
