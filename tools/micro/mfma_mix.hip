// Microbenchmark: what an 8-accumulator v_mfma_f32_16x16x4_f32 / 4-accumulator 32x32x2 loop sustains per SIMD when
// LDS reads and L2-resident buffer loads are issued beside it (the instruction mix of conv_mfma.h's k-step).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE, bool BIG>   // MODE bit0: ds_read per step, bit1: two x4 buffer loads per step (ring of 3), bit2: x1 loads (8 per step)
__global__ void __launch_bounds__(512) k(const float *in, float *out, int iters, int bytes)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x)
        lds[i] = 0.001f * i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, bytes, 0x00020000);
    f4 acc[8];
    f16v accb[4];
    for (int p = 0; p < 8; ++p) acc[p] = (f4)(0.0f);
    for (int p = 0; p < 4; ++p) accb[p] = (f16v)(0.0f);
    float a = lds[lane], b[3][8];
    int voff = ((blockIdx.x * 64 + lane) * 32) % (bytes - 4096);
    for (int g = 0; g < 3; ++g)
        for (int p = 0; p < 8; ++p) b[g][p] = 1.0f + p + g;
    int soff = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (BIG) {
#pragma unroll
                for (int p = 0; p < 4; ++p) accb[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[g][p], accb[p], 0, 0, 0);
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[g][p], acc[p], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int h = 0; h < (BIG ? 1 : 2); ++h) {
                    u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * h, soff, 0);
                    unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
                    b[g][4 * h + 0] = __builtin_bit_cast(float, e0);
                    b[g][4 * h + 1] = __builtin_bit_cast(float, e1);
                    b[g][4 * h + 2] = __builtin_bit_cast(float, e2);
                    b[g][4 * h + 3] = __builtin_bit_cast(float, e3);
                }
            }
            if (MODE & 4) {
#pragma unroll
                for (int p = 0; p < (BIG ? 4 : 8); ++p)
                    b[g][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (voff >> 3) + p * 64, soff, 0));
            }
            if (MODE & 1)
                a = lds[(lane + it * 64 + g * 16) & 8191];
            soff = (soff + 2048) & 0x3ffff;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    for (int p = 0; p < 4; ++p) for (int r = 0; r < 16; ++r) s += accb[p][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, bool BIG>
void run(const char *name, int waves_per_cu, const float *in, float *out, int bytes)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = waves_per_cu >= 8 ? 512 : waves_per_cu * 64;
    const int blocks = 256 * (waves_per_cu >= 8 ? waves_per_cu / 8 : 1);
    k<MODE, BIG><<<blocks, threads>>>(in, out, 10, bytes);
    hipEventRecord(e0);
    k<MODE, BIG><<<blocks, threads>>>(in, out, iters, bytes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 3 * (BIG ? 4 * 4096.0 : 8 * 2048.0);
    printf("%-34s waves/CU %2d  %7.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)\n", name, waves_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main()
{
    const int bytes = 1 << 20;
    float *in, *out;
    hipMalloc(&in, bytes); hipMalloc(&out, 4 << 20);
    hipMemset(in, 0, bytes);
    for (int w : {4, 8, 16}) {
        run<0, false>("16x16x4 pure", w, in, out, bytes);
        run<1, false>("16x16x4 + ds_read", w, in, out, bytes);
        run<2, false>("16x16x4 + 2 x4 loads", w, in, out, bytes);
        run<3, false>("16x16x4 + ds_read + 2 x4 loads", w, in, out, bytes);
        run<5, false>("16x16x4 + ds_read + 8 x1 loads", w, in, out, bytes);
        run<0, true>("32x32x2 pure", w, in, out, bytes);
        run<3, true>("32x32x2 + ds_read + 1 x4 load", w, in, out, bytes);
        run<5, true>("32x32x2 + ds_read + 4 x1 loads", w, in, out, bytes);
    }
    return 0;
}
