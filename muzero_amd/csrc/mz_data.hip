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


// ---- the random inputs of one lock-step move, drawn on the device in one launch --------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): a counter-based generator, so every
// game reads its own stream (key = seed, counter = move number, game, block) with no state but the move counter.
struct Philox {
    uint32_t k0, k1, c0, c1, c2;
    __device__ void block(uint32_t index, uint32_t (&out)[4]) const {
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = index, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t h0 = __umulhi(0xD2511F53u, x0), l0 = 0xD2511F53u * x0;
            const uint32_t h1 = __umulhi(0xCD9E8D57u, x2), l1 = 0xCD9E8D57u * x2;
            x0 = h1 ^ x1 ^ a, x1 = l1, x2 = h0 ^ x3 ^ b, x3 = l0;
            a += 0x9E3779B9u, b += 0xBB67AE85u;
        }
        out[0] = x0, out[1] = x1, out[2] = x2, out[3] = x3;
    }
};
// sequential 32-bit words of one game's stream
struct WordStream {
    const Philox &g;
    uint32_t index, buf[4];
    int have;
    __device__ WordStream(const Philox &gen, uint32_t first_block) : g(gen), index(first_block), have(0) {}
    __device__ uint32_t next() {
        if (have == 0) g.block(index++, buf), have = 4;
        return buf[4 - have--];
    }
    __device__ float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }            // [0, 1)
    __device__ float uniform_open() { return ((float)(next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0, 1)
    __device__ float normal() {  // Box-Muller
        const float u1 = uniform_open(), u2 = uniform();
        return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    // Gamma(alpha, 1), alpha > 0: Marsaglia & Tsang (2000); alpha < 1 through Gamma(alpha + 1) * U^(1 / alpha)
    __device__ float gamma(float alpha) {
        const float boost = alpha < 1.0f ? powf(uniform_open(), 1.0f / alpha) : 1.0f;
        const float a = alpha < 1.0f ? alpha + 1.0f : alpha;
        const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
        for (int tries = 0; tries < 64; ++tries) {
            const float x = normal(), t = 1.0f + c * x;
            if (t <= 0.0f) continue;
            const float v = t * t * t, u = uniform_open();
            if (u < 1.0f - 0.0331f * x * x * x * x || logf(u) < 0.5f * x * x + d * (1.0f - v + logf(v))) return d * v * boost;
        }
        return d * boost;  // (never reached in practice: the acceptance rate is above 0.95)
    }
};

// state i64[2]: the move counter (read by every workgroup, advanced by the last one to finish) and the arrival
// ticket.  One lane per game.
__global__ void __launch_bounds__(256)
k_move_draws(unsigned long long seed, long long *__restrict__ state, const float *__restrict__ eps, const float *__restrict__ p_random,
             float alpha, int G, int A, int S1, int32_t *__restrict__ rand_child, float *__restrict__ eta,
             int8_t *__restrict__ explore, int32_t *__restrict__ random_action) {
    const unsigned long long move = (unsigned long long)*reinterpret_cast<volatile long long *>(state);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) {
        const Philox gen{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)move, (uint32_t)(move >> 32), (uint32_t)g};
        {   // the child each simulation starts from (src/search/mcts.py:64-66 with expansion.py:35): uniform over the actions
            WordStream w(gen, 0);
            for (int i = 0; i < S1; ++i) rand_child[(size_t)g * S1 + i] = (int32_t)__umulhi(w.next(), (uint32_t)A);
        }
        {   // exploration coin and action of _select_action (src/training_data_generator.py:148-163)
            WordStream w(gen, 0x40000000u);
            explore[g] = w.uniform() < eps[0] ? 1 : 0;
            int a = A - 1;
            if (p_random) {
                const float u = w.uniform();
                float cum = 0.0f;
                for (int i = 0; i < A; ++i) {
                    cum += p_random[i];
                    if (u < cum) {
                        a = i;
                        break;
                    }
                }
            } else {
                a = (int)__umulhi(w.next(), (uint32_t)A);
            }
            random_action[g] = a;
        }
        {   // Dirichlet(alpha) noise of the root (src/search/mcts.py:85-94): normalised Gamma(alpha, 1) draws
            WordStream w(gen, 0x80000000u);
            float sum = 0.0f;
            for (int i = 0; i < A; ++i) {
                const float v = fmaxf(w.gamma(alpha), 1e-30f);
                eta[(size_t)g * A + i] = v;
                sum += v;
            }
            for (int i = 0; i < A; ++i) eta[(size_t)g * A + i] /= sum;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long *>(state + 1), 1ULL);
        if (ticket == (unsigned long long)gridDim.x - 1) {  // every workgroup has read the counter: the next launch sees the next move
            state[1] = 0;
            state[0] = (long long)(move + 1);
        }
    }
}


// end of a lock-step move: the step counters advance; a game ends when its environment says so or at the step limit
// (src/training_data_generator.py:107-142: `for step in range(max_steps_per_episode)` / `if done: break`), and the
// counter of an ended game returns to 0 for its next episode
__global__ void k_episode_end(const int8_t *__restrict__ done, int32_t *__restrict__ step, int max_steps, int G,
                              int8_t *__restrict__ ended) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int t = step[g] + 1;
    const bool e = done[g] != 0 || t >= max_steps;
    ended[g] = e ? 1 : 0;
    step[g] = e ? 0 : t;
}

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

int mz_move_draws(uint64_t seed, int64_t *state, const float *epsilon, const float *p_random, float alpha, int32_t G, int32_t A,
                  int32_t draws_per_game, int32_t *rand_child, float *eta, int8_t *explore, int32_t *random_action, void *stream) {
    if (!state || !epsilon || !rand_child || !eta || !explore || !random_action || G <= 0 || A <= 0 || draws_per_game <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_move_draws: NULL argument or non-positive dimension");
    if (!(alpha > 0.0f)) return mz::fail(MZ_ERR_ARG, "mz_move_draws: the Dirichlet parameter must be positive");
    hipLaunchKernelGGL(k_move_draws, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed,
                       reinterpret_cast<long long *>(state), epsilon, p_random, alpha, G, A, draws_per_game, rand_child, eta, explore,
                       random_action);
    return mz::check_launch("k_move_draws");
}

int mz_episode_end(const int8_t *done, int32_t *step_in_episode, int32_t max_steps, int32_t G, int8_t *ended, void *stream) {
    if (!done || !step_in_episode || !ended || G <= 0) return mz::fail(MZ_ERR_ARG, "mz_episode_end: NULL argument or non-positive dimension");
    hipLaunchKernelGGL(k_episode_end, dim3((G + 255) / 256), dim3(256), 0, (hipStream_t)stream, done, step_in_episode, max_steps, G, ended);
    return mz::check_launch("k_episode_end");
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
