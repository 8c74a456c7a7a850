// Data-side kernels of the self-play / learner path (gfx950): observation history ring +
// history tensor, the MLP-config representation network (1x1 convolution), epsilon-greedy
// action pick, and unroll value targets.  All are one-pass, HBM-bound element kernels.
#include <cmath>

#include "mz_common.h"

namespace {

// ring f32[G][K][F] (F = C*H*W floats per frame), act_ring i32[G][K]; out f32[G][C*K+K][H*W].
// Step t = 0 of an episode replicates the frame into every slot (FrameRingBuffer.fill); step
// t > 0 overwrites slot (t-1) % K (FrameRingBuffer.add).  The history is read in SLOT order.
__global__ void k_history(float *__restrict__ ring, int32_t *__restrict__ act_ring, const float *__restrict__ frame,
                          const int32_t *__restrict__ action, const int32_t *__restrict__ step, int G, int K, int F,
                          int P, float *__restrict__ out) {
    const long long per_game = (long long)K * F + (long long)K * P;
    const long long total = (long long)G * per_game;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i / per_game);
        const long long e = i % per_game;
        const int t = step[g];
        const int wr = (t <= 0) ? -1 : (t - 1) % K;  // -1: every slot
        if (e < (long long)K * F) {
            const int k = (int)(e / F), f = (int)(e % F);
            float v;
            if (wr < 0 || k == wr) {
                v = frame[(size_t)g * F + f];
                ring[((size_t)g * K + k) * F + f] = v;
            } else {
                v = ring[((size_t)g * K + k) * F + f];
            }
            out[(size_t)g * per_game + e] = v;
        } else {
            const long long r = e - (long long)K * F;
            const int k = (int)(r / P), p = (int)(r % P);
            int a;
            if (wr < 0 || k == wr) {
                a = action ? action[g] : 0;
                if (p == 0) act_ring[(size_t)g * K + k] = a;
            } else {
                a = act_ring[(size_t)g * K + k];
            }
            out[(size_t)g * per_game + e] = (float)a;
        }
    }
}

// out[g][o][p] = bias[o] + sum_c W[o][c] * x[g][c][p]
__global__ void k_pointwise_conv(const float *__restrict__ x, const float *__restrict__ W,
                                 const float *__restrict__ bias, int G, int Cin, int Cout, int P,
                                 float *__restrict__ out) {
    const long long total = (long long)G * Cout * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const int o = (int)((i / P) % Cout);
        const int g = (int)(i / ((long long)P * Cout));
        float s = bias ? bias[o] : 0.0f;
        for (int c = 0; c < Cin; ++c) s = fmaf(W[o * Cin + c], x[((size_t)g * Cin + c) * P + p], s);
        out[i] = s;
    }
}

// action[g] = explore[g] ? random_action[g] : first argmax of policy[g][:]
__global__ void k_pick_action(const float *__restrict__ policy, const int8_t *__restrict__ explore,
                              const int32_t *__restrict__ random_action, int G, int A, int32_t *__restrict__ action) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (explore && explore[g]) {
        action[g] = random_action[g];
        return;
    }
    const float *p = policy + (size_t)g * A;
    int best = 0;
    for (int a = 1; a < A; ++a)
        if (p[a] > p[best]) best = a;
    action[g] = best;
}

// z[b][k] = sum_{i=0..w-k} pw[i] * R[b][k+i] + pw[w-k+1] * V[b][w],  pw[j] = gamma^(j-1),
// accumulated left to right in float64 from 0.0, one rounding per operation.
__global__ void k_z_targets(const double *__restrict__ R, const double *__restrict__ V,
                            const double *__restrict__ pw, int B, int w, double *__restrict__ z) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (w + 1)) return;
    const int b = i / (w + 1), k = i % (w + 1);
    double acc = 0.0;
    for (int j = 0; k + j <= w; ++j) {
        double term = pw[j] * R[(size_t)b * (w + 1) + k + j];
        acc = acc + term;
    }
    double boot = pw[w - k + 1] * V[(size_t)b * (w + 1) + w];
    z[i] = acc + boot;
}

inline int grid_for(long long total, int block) { return (int)std::min<long long>((total + block - 1) / block, 65535); }

}  // namespace

extern "C" {

int mz_history_push_assemble(float *ring, int32_t *act_ring, const float *frame, const int32_t *action,
                             const int32_t *step, int32_t G, int32_t K, int32_t C, int32_t H, int32_t W, float *out,
                             void *stream) {
    if (!ring || !act_ring || !frame || !step || !out || G <= 0 || K <= 0 || C <= 0 || H <= 0 || W <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_history_push_assemble: NULL argument or non-positive dimension");
    const int P = H * W, F = C * P;
    long long total = (long long)G * ((long long)K * F + (long long)K * P);
    hipLaunchKernelGGL(k_history, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ring, act_ring, frame,
                       action, step, G, K, F, P, out);
    return mz::check_launch("k_history");
}

int mz_repr_pointwise(const float *x, const float *weight, const float *bias, int32_t G, int32_t Cin, int32_t Cout,
                      int32_t P, float *out, void *stream) {
    if (!x || !weight || !out || G <= 0 || Cin <= 0 || Cout <= 0 || P <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_repr_pointwise: NULL argument or non-positive dimension");
    long long total = (long long)G * Cout * P;
    hipLaunchKernelGGL(k_pointwise_conv, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, weight, bias,
                       G, Cin, Cout, P, out);
    return mz::check_launch("k_pointwise_conv");
}

int mz_pick_action(const float *policy, const int8_t *explore, const int32_t *random_action, int32_t G, int32_t A,
                   int32_t *action, void *stream) {
    if (!policy || !action || G <= 0 || A <= 0 || (explore && !random_action))
        return mz::fail(MZ_ERR_ARG, "mz_pick_action: NULL argument or non-positive dimension");
    hipLaunchKernelGGL(k_pick_action, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, policy, explore,
                       random_action, G, A, action);
    return mz::check_launch("k_pick_action");
}

int mz_gamma_powers(double gamma, int32_t w, double *h_out) {
    if (!h_out || w < 0) return mz::fail(MZ_ERR_ARG, "mz_gamma_powers: bad arguments");
    for (int j = 0; j <= w + 1; ++j) h_out[j] = std::pow(gamma, (double)(j - 1));
    return MZ_OK;
}

int mz_z_targets(const double *R, const double *V, const double *pow_tab, int32_t B, int32_t w, double *z,
                 void *stream) {
    if (!R || !V || !pow_tab || !z || B <= 0 || w < 0) return mz::fail(MZ_ERR_ARG, "mz_z_targets: bad arguments");
    int total = B * (w + 1);
    hipLaunchKernelGGL(k_z_targets, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, R, V, pow_tab, B, w, z);
    return mz::check_launch("k_z_targets");
}

}  // extern "C"
