// Unroll loss of the learner (gfx950): the three losses of src/training.py:307-320 summed over the
// unroll steps of src/training.py:275-303 for B training windows in ONE launch, together with their
// gradients with respect to the network outputs (the backward pass of the loss is a scaling of those).
//   policy  F.cross_entropy(pred_policy, target_policy) with the already-softmaxed policy used as
//           logits: -sum_a pi_a log_softmax(p)_a                      (all w + 1 steps)
//   value   (v - z)^2                                                  (all w + 1 steps)
//   reward  (r_pred - r)^2                                             (the w unrolled steps)
// loss_b = cp * sum policy + cv * sum value + cr * sum reward; weighted = sum_b is_weight_b * loss_b.
#include <cmath>

#include "mz_common.h"

namespace {

constexpr int kLT = 256;

// One lane per window.  pred_p f32[w+1][B][A], pred_v f32[w+1][B], pred_r f32[w][B] (step-major, as
// the unroll produces them); target_p f32[B][w+1][A], z / r f32[B][w+1]; gradients in the layouts of
// the predictions.
__global__ void __launch_bounds__(kLT)
k_unroll_loss(const float *__restrict__ pred_p, const float *__restrict__ pred_v, const float *__restrict__ pred_r,
              const float *__restrict__ target_p, const float *__restrict__ z, const float *__restrict__ r,
              const float *__restrict__ isw, int B, int w, int A, float cp, float cv, float cr,
              float *__restrict__ per_window, float *__restrict__ parts, float *__restrict__ g_p, float *__restrict__ g_v,
              float *__restrict__ g_r) {
    const int b = blockIdx.x * kLT + threadIdx.x;
    if (b >= B) return;
    const float wb = isw[b];
    float lp = 0.0f, lv = 0.0f, lr = 0.0f;
    for (int i = 0; i <= w; ++i) {
        const float *p = pred_p + ((long long)i * B + b) * A;
        const float *t = target_p + ((long long)b * (w + 1) + i) * A;
        float mx = p[0];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, p[a]);
        float se = 0.0f, tsum = 0.0f;
        for (int a = 0; a < A; ++a) {
            se += expf(p[a] - mx);
            tsum += t[a];
        }
        const float lse = logf(se) + mx;
        float ce = 0.0f;
        for (int a = 0; a < A; ++a) {
            ce -= t[a] * (p[a] - lse);
            // d ce / d p_a = softmax(p)_a * sum(t) - t_a
            g_p[((long long)i * B + b) * A + a] = wb * cp * (expf(p[a] - lse) * tsum - t[a]);
        }
        lp += ce;
        const float dv = pred_v[(long long)i * B + b] - z[(long long)b * (w + 1) + i];
        lv += dv * dv;
        g_v[(long long)i * B + b] = wb * cv * 2.0f * dv;
        if (i < w) {
            const float dr = pred_r[(long long)i * B + b] - r[(long long)b * (w + 1) + i];
            lr += dr * dr;
            g_r[(long long)i * B + b] = wb * cr * 2.0f * dr;
        }
    }
    lp *= cp;
    lv *= cv;
    lr *= cr;
    per_window[b] = lp + lv + lr;
    parts[b] = lp;
    parts[B + b] = lv;
    parts[2 * B + b] = lr;
}

// sums[0] = sum_b isw_b * per_window_b, sums[1..3] = sum_b of the policy / value / reward parts: one
// workgroup, pairwise tree in a fixed order (the result does not depend on the launch).
__global__ void __launch_bounds__(kLT)
k_loss_sums(const float *__restrict__ per_window, const float *__restrict__ parts, const float *__restrict__ isw, int B,
            float *__restrict__ sums) {
    __shared__ float s[4][kLT];
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int b = threadIdx.x; b < B; b += kLT) {
        a0 += isw[b] * per_window[b];
        a1 += parts[b];
        a2 += parts[B + b];
        a3 += parts[2 * B + b];
    }
    s[0][threadIdx.x] = a0, s[1][threadIdx.x] = a1, s[2][threadIdx.x] = a2, s[3][threadIdx.x] = a3;
    __syncthreads();
    for (int k = kLT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k)
            for (int j = 0; j < 4; ++j) s[j][threadIdx.x] += s[j][threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = s[threadIdx.x][0];
}

}  // namespace

extern "C" {

int mz_unroll_loss(const float *pred_p, const float *pred_v, const float *pred_r, const float *target_p, const float *z,
                   const float *r, const float *is_weights, int32_t B, int32_t w, int32_t A, float cp, float cv, float cr,
                   float *per_window, float *parts, float *sums, float *g_p, float *g_v, float *g_r, void *stream) {
    if (!pred_p || !pred_v || (w > 0 && !pred_r) || !target_p || !z || !r || !is_weights || !per_window || !parts || !sums ||
        !g_p || !g_v || (w > 0 && !g_r))
        return mz::fail(MZ_ERR_ARG, "mz_unroll_loss: NULL argument");
    if (B <= 0 || w < 0 || A <= 0) return mz::fail(MZ_ERR_ARG, "mz_unroll_loss: non-positive dimension");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_unroll_loss, dim3((B + kLT - 1) / kLT), dim3(kLT), 0, st, pred_p, pred_v, pred_r, target_p, z, r,
                       is_weights, B, w, A, cp, cv, cr, per_window, parts, g_p, g_v, g_r);
    MZ_TRY(mz::check_launch("k_unroll_loss"));
    hipLaunchKernelGGL(k_loss_sums, dim3(1), dim3(kLT), 0, st, per_window, parts, is_weights, B, sums);
    return mz::check_launch("k_loss_sums");
}

}  // extern "C"
