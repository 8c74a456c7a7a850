// Issue rate of the fp32 matrix instructions from W waves per SIMD with independent
// accumulators: cycles per instruction as seen by each wave (shader clock).
#include <hip/hip_runtime.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void __launch_bounds__(1024) k_rate16(int iters, float *sink, long long *out) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int NACC>
__global__ void __launch_bounds__(1024) k_rate32(int iters, float *sink, long long *out) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}

extern "C" int rate_run(int kind, int waves, int iters, float *sink, long long *out) {
    dim3 grid(256), block(64 * waves);
    if (kind == 16) hipLaunchKernelGGL(k_rate16<16>, grid, block, 0, 0, iters, sink, out);
    else if (kind == 164) hipLaunchKernelGGL(k_rate16<4>, grid, block, 0, 0, iters, sink, out);
    else if (kind == 32) hipLaunchKernelGGL(k_rate32<4>, grid, block, 0, 0, iters, sink, out);
    else hipLaunchKernelGGL(k_rate32<2>, grid, block, 0, 0, iters, sink, out);
    return (int)hipDeviceSynchronize();
}
