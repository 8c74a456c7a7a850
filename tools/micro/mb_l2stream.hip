// How fast a compute unit streams an L2-resident weight blob into registers: 256 workgroups x 8 waves, every
// workgroup reads the same `bytes` region `reps` times with global_load_dwordx4, `depth` KiB in flight per wave.
// mode 0: the 8 waves read disjoint eighths of the region (every byte enters the CU once per repetition);
// mode 1: waves w and w+4 read the same quarter at the same time (the second reader can hit in the vector L1);
// mode 2: as 0, but only waves 0..3 run (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int DEPTH>  // 1 KiB loads in flight per wave
__global__ void __launch_bounds__(512) k_stream(const f32x4 *__restrict__ w, int kib_per_wave, int stride_kib, int mode, int reps,
                                                float *sink, long long *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (mode == 2 && wave >= 4) return;
    const int part = mode == 1 ? (wave & 3) : wave;
    const f32x4 *base = w + (size_t)part * stride_kib * 64 + lane;
    f32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < kib_per_wave; k += DEPTH) {
            f32x4 v[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) v[j] = base[(size_t)(k + j) * 64];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc += v[j];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

int main(int argc, char **argv) {
    const int total_kib = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps = 20;
    f32x4 *w;
    float *sink;
    long long *out;
    hipMalloc(&w, (size_t)total_kib * 1024 + (64 << 10));
    hipMemset(w, 0, (size_t)total_kib * 1024 + (64 << 10));
    hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&out, 8 * 8);
    for (int mode = 0; mode < 3; ++mode)
        for (int depth : {4, 8, 16, 32}) {
            const int parts = mode == 1 ? 4 : (mode == 2 ? 4 : 8);
            const int kib = total_kib / parts;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            auto launch = [&]() {
                if (depth == 4) hipLaunchKernelGGL(k_stream<4>, dim3(256), dim3(512), 0, 0, w, kib, kib, mode, reps, sink, out);
                else if (depth == 8) hipLaunchKernelGGL(k_stream<8>, dim3(256), dim3(512), 0, 0, w, kib, kib, mode, reps, sink, out);
                else if (depth == 16) hipLaunchKernelGGL(k_stream<16>, dim3(256), dim3(512), 0, 0, w, kib, kib, mode, reps, sink, out);
                else hipLaunchKernelGGL(k_stream<32>, dim3(256), dim3(512), 0, 0, w, kib, kib, mode, reps, sink, out);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            long long h[8];
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            const double bytes_into_regs = (double)total_kib * 1024 * reps * (mode == 1 ? 2 : 1);
            const double unique = (double)total_kib * 1024 * reps;
            printf("mode %d depth %2d KiB/wave in flight: %.3f ms, wave0 %lld cycles -> %.1f B/clk/CU into registers, %.1f B/clk/CU unique, %.2f TB/s chip (unique)\n",
                   mode, depth, ms, h[0], bytes_into_regs / h[0], unique / h[0], unique * 256 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
