// Replay side of the learner on the device (gfx950): episode-level prioritised sampling
// (ReplayBuffer.sample_batch / update_priorities, src/replay_buffer.py:72-130) over the priorities of
// a fixed-shape episode store, and the cut of training windows out of that store
// (src/training.py:131-160 with the history assembly of bptt, :252-262).
//
// Randomness is an input, as everywhere in this library: the caller passes the uniform numbers
// np.random.choice would draw (RandomState.random_sample), and the kernel repeats numpy's
// inverse-CDF search on them -- cumulative sum left to right in float64, normalised by its last
// entry, searchsorted(side="right") -- so the same stream selects the same episodes.
// Compiled with -ffp-contract=off (the sums are chains of individually rounded additions).
#include <cmath>

#include "mz_common.h"

namespace {

constexpr int kRT = 256;
constexpr int kChunk = 2048;  // doubles staged per pass of the sequential sums

// One workgroup.  priority f64[n]; probs / cdf f64[n] are outputs that double as scratch.
// u f64[B] -> draw i32[B], weight f64[B] = (n P(draw))^-beta / max over the draw.
// Optional window choice (done_len != NULL): a draw whose episode is shorter than `need` chunks is
// dropped -- it stays in the batch as a window of the longest stored episode with weight 0 and
// slot = cap -- and the window start is lookback + floor(u_start * span) over the span admissible
// starts (src/training.py:139-142, 210-221).
__global__ void __launch_bounds__(kRT)
k_per_sample(const double *__restrict__ priority, int n, double alpha, double beta, const double *__restrict__ u, int B,
             double *__restrict__ probs, double *__restrict__ cdf, int32_t *__restrict__ draw, double *__restrict__ weight,
             const long long *__restrict__ done_len, int need, int lookback, int roll_ahead, int cap,
             const double *__restrict__ u_start, float *__restrict__ weight32, int32_t *__restrict__ slot,
             int32_t *__restrict__ cut_from, int32_t *__restrict__ start) {
    __shared__ double s_red[kRT];
    __shared__ int s_idx[kRT];
    __shared__ double s_total;
    __shared__ int s_longest;
    // The two left-to-right sums (Python's sum() and numpy's cumsum) are chains of dependent additions:
    // one lane runs them, over chunks that all lanes stage in LDS first, so that the chain waits for
    // the adder only and never for global memory.
    __shared__ double s_chunk[kChunk];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += kRT) probs[i] = pow(priority[i], alpha);
    __syncthreads();
    {
        double total = 0.0;
        for (int c0 = 0; c0 < n; c0 += kChunk) {
            const int len = min(kChunk, n - c0);
            for (int i = tid; i < len; i += kRT) s_chunk[i] = probs[c0 + i];
            __syncthreads();
            if (tid == 0) {
#pragma unroll 8
                for (int i = 0; i < len; ++i) total = total + s_chunk[i];
            }
            __syncthreads();
        }
        if (tid == 0) s_total = total;
    }
    __syncthreads();
    const double total = s_total;
    for (int i = tid; i < n; i += kRT) probs[i] = total < 1e-8 ? 1.0 / (double)n : probs[i] / total;
    __syncthreads();
    {
        double run = 0.0;
        for (int c0 = 0; c0 < n; c0 += kChunk) {
            const int len = min(kChunk, n - c0);
            for (int i = tid; i < len; i += kRT) s_chunk[i] = probs[c0 + i];
            __syncthreads();
            if (tid == 0) {
#pragma unroll 8
                for (int i = 0; i < len; ++i) {
                    run = run + s_chunk[i];
                    s_chunk[i] = run;
                }
            }
            __syncthreads();
            for (int i = tid; i < len; i += kRT) cdf[c0 + i] = s_chunk[i];
            __syncthreads();
        }
        if (tid == 0) s_total = run;
    }
    __syncthreads();
    const double last = s_total;
    for (int i = tid; i < n; i += kRT) cdf[i] = cdf[i] / last;
    // the longest stored episode (first maximum) stands in for dropped draws
    if (done_len) {
        long long best = -1;
        int bi = 0;
        for (int i = tid; i < n; i += kRT)
            if (done_len[i] > best) {
                best = done_len[i];
                bi = i;
            }
        s_red[tid] = (double)best;
        s_idx[tid] = bi;
    }
    __syncthreads();
    if (done_len && tid == 0) {
        double best = s_red[0];
        int bi = s_idx[0];
        for (int t = 1; t < kRT; ++t)
            if (s_red[t] > best || (s_red[t] == best && s_idx[t] < bi)) {
                best = s_red[t];
                bi = s_idx[t];
            }
        s_longest = bi;
    }
    __syncthreads();
    double wmax = 0.0;
    for (int b = tid; b < B; b += kRT) {
        const double x = u[b];
        int lo = 0, hi = n;  // first index with cdf[index] > x
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= x) lo = mid + 1;
            else hi = mid;
        }
        const int d = lo < n ? lo : n - 1;
        draw[b] = d;
        const double w = pow((double)n * probs[d], -beta);
        weight[b] = w;
        wmax = fmax(wmax, w);
    }
    s_red[tid] = wmax;
    __syncthreads();
    for (int s = kRT / 2; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] = fmax(s_red[tid], s_red[tid + s]);
        __syncthreads();
    }
    wmax = s_red[0];
    for (int b = tid; b < B; b += kRT) {
        const double w = weight[b] / wmax;
        weight[b] = w;
        if (done_len) {
            const int d = draw[b];
            const bool keep = done_len[d] >= need;
            const int from = keep ? d : s_longest;
            const long long span = done_len[from] - roll_ahead - lookback;
            long long off = (long long)(u_start[b] * (double)span);
            if (off > span - 1) off = span - 1;
            if (off < 0) off = 0;
            weight32[b] = keep ? (float)w : 0.0f;
            slot[b] = keep ? d : cap;
            cut_from[b] = from;
            start[b] = lookback + (int)off;
        }
    }
}

// priority[slot] = |error| + 1e-6 in draw order (the last occurrence of a slot wins, as in the
// reference's loop), the running maximum follows; slot >= cap entries are ignored.
__global__ void k_per_update(double *__restrict__ priority, double *__restrict__ max_priority, const int32_t *__restrict__ slot,
                             const float *__restrict__ error, int B, int cap) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double mx = *max_priority;
    for (int b = 0; b < B; ++b) {
        const int s = slot[b];
        if (s < 0 || s >= cap) continue;
        const double pri = (double)fabsf(error[b]) + 1e-6;
        mx = fmax(mx, pri);
        priority[s] = pri;
    }
    *max_priority = mx;
}

// One workgroup per window b = (episode row cut_from[b], start k[b]).  History: the K-slot ring of
// bptt filled with the first past frame and the later ones added (slot j ends up holding past frame
// start + 1 + a, a the largest add index = j mod K, or the first past frame), frames in SLOT order
// on the channel axis followed by K planes holding each slot's action id.  Targets: chunks k .. k+w.
__global__ void __launch_bounds__(kRT)
k_cut_windows(const float *__restrict__ obs, const long long *__restrict__ action, const float *__restrict__ policy,
              const double *__restrict__ value, const float *__restrict__ reward, int T, int F, int P, int A,
              const int32_t *__restrict__ rows, const int32_t *__restrict__ ks, int lookback, int w, int K,
              const double *__restrict__ pw, float *__restrict__ history, long long *__restrict__ actions,
              float *__restrict__ policies, float *__restrict__ z, float *__restrict__ rewards) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long row = rows[b];
    const int k = ks[b];
    const int start = max(0, k - lookback);
    const int n_add = k - start - 1;
    const long long per_win = (long long)K * F + (long long)K * P;
    float *hist = history + (long long)b * per_win;
    for (int j = 0; j < K; ++j) {
        int frame = start;
        if (n_add > j) {
            // largest a < n_add with a = j (mod K)
            const int a = j + K * ((n_add - 1 - j) / K);
            frame = start + 1 + a;
        }
        const float *src = obs + (row * T + frame) * (long long)F;
        for (int i = tid; i < F; i += kRT) hist[(long long)j * F + i] = src[i];
        const float av = (float)action[row * T + frame];
        for (int i = tid; i < P; i += kRT) hist[(long long)K * F + (long long)j * P + i] = av;
    }
    for (int i = tid; i <= w; i += kRT) {
        actions[(long long)b * (w + 1) + i] = action[row * T + k + i];
        rewards[(long long)b * (w + 1) + i] = reward[row * T + k + i];
        // _compute_z_targets (src/training.py:223-242): the arithmetic of k_z_targets on this window
        double acc = 0.0;
        for (int j = 0; i + j <= w; ++j) {
            const double term = pw[j] * (double)reward[row * T + k + i + j];
            acc = acc + term;
        }
        const double boot = pw[w - i + 1] * value[row * T + k + w];
        z[(long long)b * (w + 1) + i] = (float)(acc + boot);
    }
    for (int i = tid; i < (w + 1) * A; i += kRT)
        policies[(long long)b * (w + 1) * A + i] = policy[(row * T + k) * (long long)A + i];
}

}  // namespace

extern "C" {

int mz_per_sample(const double *priority, int32_t n, double alpha, double beta, const double *u, int32_t B, double *probs,
                  double *cdf, int32_t *draw, double *weight, const int64_t *done_len, int32_t need, int32_t lookback,
                  int32_t roll_ahead, int32_t cap, const double *u_start, float *weight32, int32_t *slot,
                  int32_t *cut_from, int32_t *start, void *stream) {
    if (!priority || !u || !probs || !cdf || !draw || !weight || n <= 0 || B <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_per_sample: NULL argument, empty store or empty batch");
    if (done_len && (!u_start || !weight32 || !slot || !cut_from || !start || need <= 0 || lookback < 0 || roll_ahead < 0))
        return mz::fail(MZ_ERR_ARG, "mz_per_sample: the window choice needs u_start and all four outputs");
    hipLaunchKernelGGL(k_per_sample, dim3(1), dim3(kRT), 0, (hipStream_t)stream, priority, n, alpha, beta, u, B, probs, cdf,
                       draw, weight, reinterpret_cast<const long long *>(done_len), need, lookback, roll_ahead, cap, u_start,
                       weight32, slot, cut_from, start);
    return mz::check_launch("k_per_sample");
}

int mz_per_update(double *priority, double *max_priority, const int32_t *slot, const float *error, int32_t B, int32_t cap,
                  void *stream) {
    if (!priority || !max_priority || !slot || !error || B <= 0 || cap <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_per_update: NULL or non-positive argument");
    hipLaunchKernelGGL(k_per_update, dim3(1), dim3(64), 0, (hipStream_t)stream, priority, max_priority, slot, error, B, cap);
    return mz::check_launch("k_per_update");
}

int mz_replay_cut_windows(const float *obs, const int64_t *action, const float *policy, const double *value,
                          const float *reward, int32_t T, int32_t C, int32_t H, int32_t W, int32_t A,
                          const int32_t *rows, const int32_t *k, int32_t B, int32_t lookback, int32_t roll_ahead,
                          int32_t K, const double *pow_tab, float *history, int64_t *actions, float *policies, float *z,
                          float *rewards, void *stream) {
    if (!obs || !action || !policy || !value || !reward || !rows || !k || !pow_tab || !history || !actions || !policies ||
        !z || !rewards)
        return mz::fail(MZ_ERR_ARG, "mz_replay_cut_windows: NULL argument");
    if (T <= 0 || C <= 0 || H <= 0 || W <= 0 || A <= 0 || B <= 0 || lookback < 0 || roll_ahead < 0 || K <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_replay_cut_windows: non-positive dimension");
    hipLaunchKernelGGL(k_cut_windows, dim3(B), dim3(kRT), 0, (hipStream_t)stream, obs, reinterpret_cast<const long long *>(action),
                       policy, value, reward, T, C * H * W, H * W, A, rows, k, lookback, roll_ahead, K, pow_tab, history,
                       reinterpret_cast<long long *>(actions), policies, z, rewards);
    return mz::check_launch("k_cut_windows");
}

}  // extern "C"
