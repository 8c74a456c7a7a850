// Diagnostic microbenchmark of the dense routines of mz_mlp_device.h in isolation: one
// 512-thread workgroup per CU repeats one layer shape `iters` times; thread 0 of workgroup 0
// reports the average shader-clock cycles per call (including the trailing barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "mz_common.h"
#include "mz_mlp_device.h"
using namespace mzd;

template <int VARIANT>
__global__ void __launch_bounds__(kThreads) k_mb(const float *__restrict__ blob, int iters, unsigned long long *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int ld_d = lds_stride(168), ld_h = lds_stride(256), ld_r = lds_stride(128);
    lds_f *Z = to_lds(smem);
    lds_f *Tab = Z + 16 * ld_d;
    lds_f *H1 = Tab + 8 * ld_d;
    lds_f *H2 = H1 + 16 * ld_h;
    lds_f *R1 = H2 + 16 * ld_h;
    for (int i = threadIdx.x; i < 16 * ld_d + 8 * ld_d + 32 * ld_h; i += kThreads) Z[i] = 0.001f * (i % 97);
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#ifdef MZ_EXP_TIMELINE
        if ((threadIdx.x & 63) == 0) mz_tl[threadIdx.x >> 6][0] = clock64();
#endif
        if (VARIANT == 0) {  // reward layer 1 over 112 child rows: K=176, N=128, MT=7
            
#ifdef MB_LDSROWS
            dense_group1<7>(LdsRows{R1, ld_r}, blob,
#else
            dense_group1<7>(ChildRows{Z, ld_d, Tab, ld_d}, blob,
#endif
 176, DensePart{0, 22528, 128, 128, MZ_ACT_RELU, R1, ld_r, nullptr, 0, nullptr, -1, nullptr});
        } else if (VARIANT == 1) {  // policy layer 1 | merge: K=176, N=256+168
            dense_group2<1>(LdsRows{Z, ld_d}, blob, 176, DensePart{0, 90000, 256, 256, MZ_ACT_RELU, H1, ld_h, nullptr, 0, nullptr, -1, nullptr},
                            DensePart{45056, 90300, 168, 176, MZ_ACT_NONE, R1, ld_d, nullptr, 0, nullptr, -1, nullptr});
        } else if (VARIANT == 2) {  // policy layer 2: K=256, N=256
            dense_group1<1>(LdsRows{H1, ld_h}, blob, 256, DensePart{0, 90000, 256, 256, MZ_ACT_RELU, H2, ld_h, nullptr, 0, nullptr, -1, nullptr});
        } else if (VARIANT == 3) {  // value layer 2: K=256, N=128
            dense_group1<1>(LdsRows{H1, ld_h}, blob, 256, DensePart{0, 90000, 128, 128, MZ_ACT_RELU, H2, ld_h, nullptr, 0, nullptr, -1, nullptr});
        } else if (VARIANT == 13) {  // reward layer 1 over 112 child rows on a team of 4 waves
            if (uni(threadIdx.x >> 6) < 4) {
                Team te{0, 4, nullptr, 0};
                dense_group1<7>(ChildRows{Z, ld_d, Tab, ld_d}, blob, 176, DensePart{0, 22528, 128, 128, MZ_ACT_RELU, R1, ld_r, nullptr, 0, nullptr, -1, nullptr}, te);
            }
        } else if (VARIANT >= 10) {
            const int half = uni(threadIdx.x >> 6) >= 4;
            Team te{0, 4, nullptr, 0}, ts{4, 4, nullptr, 0};
            if (!half && (VARIANT == 10 || VARIANT == 12)) {
                for (int pass = 0; pass < 2; ++pass)
                    dense_group1<4>(ChildRows{Z, ld_d, Tab, ld_d}, blob, 176, DensePart{0, 22528, 128, 128, MZ_ACT_RELU, R1, ld_r, nullptr, 0, nullptr, -1, nullptr}, te);
            }
            if (half && (VARIANT == 11 || VARIANT == 12)) {
                dense_group2<1>(LdsRows{Z + 16 * ld_d, ld_d}, blob, 176, DensePart{0, 90000, 256, 256, MZ_ACT_RELU, H1, ld_h, nullptr, 0, nullptr, -1, nullptr},
                                DensePart{45056, 90300, 168, 176, MZ_ACT_NONE, H2, ld_d, nullptr, 0, nullptr, -1, nullptr}, ts);
                dense_group1<1>(LdsRows{H1, ld_h}, blob, 256, DensePart{0, 90000, 256, 256, MZ_ACT_RELU, H2, ld_h, nullptr, 0, nullptr, -1, nullptr}, ts);
                dense_group1<1>(LdsRows{H1, ld_h}, blob, 256, DensePart{0, 90000, 256, 256, MZ_ACT_RELU, H2, ld_h, nullptr, 0, nullptr, -1, nullptr}, ts);
            }
        } else if (VARIANT == 4) {  // reward layer 1 on 16 rows: K=176, N=128
            dense_group1<1>(LdsRows{Z, ld_d}, blob, 176, DensePart{0, 90000, 128, 128, MZ_ACT_RELU, H2, ld_h, nullptr, 0, nullptr, -1, nullptr});
        }
#ifdef MZ_EXP_TIMELINE
        if ((threadIdx.x & 63) == 0) mz_tl[threadIdx.x >> 6][7] = clock64();
#endif
        __syncthreads();
#ifdef MZ_EXP_TIMELINE
        if ((threadIdx.x & 63) == 0) mz_tl[threadIdx.x >> 6][8] = clock64();
#endif
    }
    unsigned long long t1 = clock64();
#ifdef MZ_EXP_TIMELINE
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 128) out[8 + threadIdx.x] = mz_tl[threadIdx.x >> 4][threadIdx.x & 15];
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (t1 - t0) / (unsigned long long)iters;
}

extern "C" int mb_run(int variant, int iters, const float *blob, unsigned long long *out, int blocks) {
    const size_t lds = sizeof(float) * (24 * lds_stride(168) + 32 * lds_stride(256) + 112 * lds_stride(128));
    auto go = [&](auto k) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(kThreads), lds, 0, blob, iters, out);
        return (int)hipDeviceSynchronize();
    };
    switch (variant) {
        case 0: return go(k_mb<0>);
        case 1: return go(k_mb<1>);
        case 2: return go(k_mb<2>);
        case 3: return go(k_mb<3>);
        case 10: return go(k_mb<10>);
        case 11: return go(k_mb<11>);
        case 12: return go(k_mb<12>);
        case 13: return go(k_mb<13>);
        default: return go(k_mb<4>);
    }
}
