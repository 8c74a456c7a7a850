// Diagnostic: do f32 MFMAs and vector-ALU FMAs of two waves on one SIMD overlap?
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_ov(int mode, int iters, float *sink, unsigned long long *out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    __syncthreads();
    const unsigned long long t0 = clock64();
    const bool do_mfma = (mode & 1) && wave < 4;
    const bool do_valu = (mode & 2) && wave >= 4;
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_fmaf(v0, y, x); v1 = __builtin_fmaf(v1, y, x); v2 = __builtin_fmaf(v2, y, x); v3 = __builtin_fmaf(v3, y, x);
                v4 = __builtin_fmaf(v4, y, x); v5 = __builtin_fmaf(v5, y, x); v6 = __builtin_fmaf(v6, y, x); v7 = __builtin_fmaf(v7, y, x);
            }
        }
    }
    const unsigned long long t1 = clock64();
    sink[blockIdx.x * 512 + threadIdx.x] = a0.x + a1.y + a2.z + a3.w + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
extern "C" int ov_run(int mode, int iters, float *sink, unsigned long long *out) {
    hipLaunchKernelGGL(k_ov, dim3(256), dim3(512), 0, 0, mode, iters, sink, out);
    return (int)hipDeviceSynchronize();
}
