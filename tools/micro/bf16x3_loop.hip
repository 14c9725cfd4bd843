// Microbenchmark for the 3-term bf16 split: what does the inner loop of a branch kernel sustain per SIMD
// when the fp32 activations are split in registers (hi = bf16(x), lo = bf16(x - hi)) and multiplied on the bf16 matrix
// cores as ah*bh + ah*bl + al*bh with fp32 accumulation -- against the same contraction on the fp32 matrix cores?
// Bare loops (operands come from a small L2-resident buffer, as in conv_mfma.h's operand ring), two or four waves per
// SIMD, plus an accuracy check of one 32x32x16 tile against a double-precision product.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/bf16x3_loop.hip -o /tmp/bf16x3_loop && /tmp/bf16x3_loop
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float *v, bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}

// hi / lo by truncation, full-rate instructions only: hi = x & 0xffff0000 (exact in bf16), lo = x - hi (exact in fp32), its
// upper half kept; pairs packed with v_perm_b32.  (v_cvt_pk_bf16_f32 rounds to nearest but issues at a quarter of the rate.)
__device__ __forceinline__ void split8_trunc(const float *v, bf16x8 &hi, bf16x8 &lo)
{
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned u0 = __builtin_bit_cast(unsigned, v[2 * j]), u1 = __builtin_bit_cast(unsigned, v[2 * j + 1]);
        const float l0 = v[2 * j] - __builtin_bit_cast(float, u0 & 0xffff0000u), l1 = v[2 * j + 1] - __builtin_bit_cast(float, u1 & 0xffff0000u);
        h[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);   // {u0.hi16, u1.hi16}
        l[j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, l1), __builtin_bit_cast(unsigned, l0), 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// MODE 0: fp32 MFMA (32x32x2), 8 k-steps per 16 channels, P pixel sets
// MODE 1: bf16x3, activations split in the loop with v_cvt_pk_bf16_f32 (weights pre-split), 3 MFMAs (32x32x16) per 16 channels
// MODE 2: bf16x3 with pre-split activations too (the layout a producer kernel would write): loads + MFMAs only
// MODE 3: bf16x3, activations split in the loop by truncation (and / sub / perm)
template <int MODE, int P>
__global__ void __launch_bounds__(1024) loop_kernel(const float *in, float *out, int iters, int bytes)
{
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, bytes, 0x00020000);
    f32x16 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
        acc[p] = (f32x16)(0.0f);
    int voff = ((blockIdx.x * 64 + lane) * 8) % (bytes - 8192);
    // weights (A operand): constant registers here (in the real kernel they come from LDS, pre-split on the host)
    float aw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        aw[j] = 0.01f * (lane + j);
    bf16x8 ah, al;
    split8(aw, ah, al);
    for (int it = 0; it < iters; ++it) {
        // 16 channels of P pixel sets: 8 values per lane per pixel set (the lane's k-group), as P-wide vector loads per channel
        float b[P][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (P == 2) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (it & 63) * 64 + j * 512, 0);
                const unsigned e0 = v[0], e1 = v[1];
                b[0][j] = __builtin_bit_cast(float, e0);
                b[1][j] = __builtin_bit_cast(float, e1);
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p)
                    b[p][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * p, (it & 63) * 64 + j * 512, 0));
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int p = 0; p < P; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j], b[p][j], acc[p], 0, 0, 0);
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                bf16x8 bh, bl;
                if (MODE == 1) {
                    split8(b[p], bh, bl);
                } else if (MODE == 3) {
                    split8_trunc(b[p], bh, bl);
                } else {   // pre-split: the 8 loaded dwords ARE the 16 bf16 values (hi in the first four registers, lo in the last four)
                    u32x4 uh, ul;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uh[j] = __builtin_bit_cast(unsigned, b[p][j]);
                        ul[j] = __builtin_bit_cast(unsigned, b[p][4 + j]);
                    }
                    bh = __builtin_bit_cast(bf16x8, uh);
                    bl = __builtin_bit_cast(bf16x8, ul);
                }
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[p], 0, 0, 0);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[p], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            s += acc[p][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one 32x32x16 product: A [32][16], B [16][32] fp32 in, D [32][32] out, through the three bf16 MFMAs
__global__ void __launch_bounds__(64) tile_kernel(const float *A, const float *B, float *D, int terms, int trunc)
{
    const int lane = threadIdx.x, i = lane & 31, kq = lane >> 5;
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = A[i * 16 + 8 * kq + j];
        bv[j] = B[(8 * kq + j) * 32 + i];
    }
    bf16x8 ah, al, bh, bl;
    if (trunc) {
        split8_trunc(av, ah, al);
        split8_trunc(bv, bh, bl);
    } else {
        split8(av, ah, al);
        split8(bv, bh, bl);
    }
    f32x16 acc = (f32x16)(0.0f);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    if (terms >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
    if (terms >= 4)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        D[((r & 3) + 8 * (r >> 2) + 4 * kq) * 32 + i] = acc[r];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int P>
static int run(const char *name, const float *din, float *dout, int bytes, int waves_per_simd)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 4096;
    const int threads = waves_per_simd * 4 * 64;   // one workgroup per CU
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((loop_kernel<MODE, P>), dim3(cus), dim3(threads), 0, 0, din, dout, 16, bytes);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((loop_kernel<MODE, P>), dim3(cus), dim3(threads), 0, 0, din, dout, iters, bytes);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    // per iteration and wave: 32 rows x 32 columns x 16 k x P pixel sets, 2 FLOP each
    const double flop = (double)cus * waves_per_simd * 4 * iters * (32.0 * 32 * 16 * P * 2);
    std::printf("%-34s P=%d %d waves/SIMD: %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)  %6.1f cycles per 16-channel step per wave at 2.4 GHz\n", name, P,
                waves_per_simd, ms, flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / iters);
    return 0;
}

int main()
{
    const int bytes = 1 << 20;
    float *din, *dout;
    CK(hipMalloc(&din, bytes));
    CK(hipMalloc(&dout, 1 << 22));
    std::vector<float> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = std::sin(0.37 * i) * (1.0f + (i % 7));
    CK(hipMemcpy(din, h.data(), bytes, hipMemcpyHostToDevice));
    for (int w : {2, 4}) {
        if (run<0, 2>("fp32 MFMA 32x32x2", din, dout, bytes, w)) return 1;
        if (run<1, 2>("bf16x3, split by cvt_pk in the loop", din, dout, bytes, w)) return 1;
        if (run<3, 2>("bf16x3, split by truncation", din, dout, bytes, w)) return 1;
        if (run<2, 2>("bf16x3, operands pre-split", din, dout, bytes, w)) return 1;
    }
    if (run<0, 4>("fp32 MFMA 32x32x2", din, dout, bytes, 2)) return 1;
    if (run<1, 4>("bf16x3, split by cvt_pk in the loop", din, dout, bytes, 2)) return 1;
    if (run<3, 4>("bf16x3, split by truncation", din, dout, bytes, 2)) return 1;
    if (run<2, 4>("bf16x3, operands pre-split", din, dout, bytes, 2)) return 1;

    // accuracy of one tile
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
    for (int i = 0; i < 32 * 16; ++i) {
        A[i] = std::sin(1.3 * i) * 0.7f;
        B[i] = std::cos(0.9 * i) * 2.1f;
    }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 4; ++variant) {
        const int terms = variant == 0 ? 1 : variant == 3 ? 4 : 3, trunc = variant == 2;
        hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, terms, trunc);
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                for (int k = 0; k < 16; ++k) {
                    ref += (double)A[i * 16 + k] * B[k * 32 + j];
                    mag += std::fabs((double)A[i * 16 + k] * B[k * 32 + j]);
                }
                worst = std::fmax(worst, std::fabs(D[i * 32 + j] - ref));
                scale = std::fmax(scale, mag);
            }
        std::printf("tile 32x32x16, %d bf16 term(s)%s: max abs error %.3e  (relative to the largest sum of |products| %.3e)\n", terms, trunc ? " (truncating split)" : "", worst, worst / scale);
    }
    return 0;
}
