// Batched MFMA forward passes of the MLP-only h/g/f networks (gfx950) and the weight
// re-tiling they read (fp32 weights as two fp16 planes for the matrix layers, fp32 tiles
// for narrow layers and for the wide-layer kernels).  Device-side building blocks: mz_mlp_device.h.
#include <algorithm>
#include <cstring>

#include "mz_common.h"
#include "mz_mlp_device.h"
#include "mz_ring.h"

using namespace mzd;

namespace {

// ------------------------------------------------------------------ weight packing
__global__ void k_pack_layer(const float *__restrict__ W, int ldw, const float *__restrict__ bias, int K, int N,
                             int Kp, int Np, float *__restrict__ w_out, float *__restrict__ b_out) {
    const long long total = (long long)Np * Kp;
    const int KT = Kp >> 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const long long tile = i >> 8;
        const int kt = (int)(tile % KT), nt = (int)(tile / KT);
        const int n = nt * 16 + (lane & 15);
        const int k = kt * 16 + 4 * (lane >> 4) + j;
        w_out[i] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Np; n += gridDim.x * blockDim.x)
        b_out[n] = (bias != nullptr && n < N) ? bias[n] : 0.0f;
}

// Matrix layers: tile (nt, c) = two planes of 64 x 16 bytes (value, scaled residual; see mz_mlp_device.h),
// lane l = 16 kq + i holding W[16 nt + i][32 c + 8 kq .. +7].  One lane operand per thread.
__global__ void k_pack_layer_split(const float *__restrict__ W, int ldw, const float *__restrict__ bias, int K, int N,
                                   int KC, int Np, float *__restrict__ w_out, float *__restrict__ b_out) {
    const long long total = (long long)(Np >> 4) * KC * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long tile = i >> 6;
        const int c = (int)(tile % KC), nt = (int)(tile / KC);
        const int n = nt * 16 + (lane & 15);
        const int k0 = 32 * c + 8 * (lane >> 4);
        float v[8];
        for (int e = 0; e < 8; ++e) v[e] = (n < N && k0 + e < K) ? W[(size_t)n * ldw + k0 + e] : 0.0f;
        f32x4 hi, lo;
        split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, hi, lo);
        f32x4 *out = reinterpret_cast<f32x4 *>(w_out) + tile * 128 + lane;
        out[0] = hi;
        out[64] = lo;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Np; n += gridDim.x * blockDim.x)
        b_out[n] = (bias != nullptr && n < N) ? bias[n] : 0.0f;
}

// table[a][n] = merge_bias[n] + sum_k merge_weight[n][D + k] * embedding[a][k]
__global__ void k_action_table(const float *__restrict__ emb, const float *__restrict__ Wm,
                               const float *__restrict__ bm, int D, int Dp, int A, float *__restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * Dp) return;
    const int a = i / Dp, n = i % Dp;
    float s = 0.0f;
    if (n < D) {
        s = bm ? bm[n] : 0.0f;
        for (int k = 0; k < D; ++k) s = fmaf(Wm[(size_t)n * 2 * D + D + k], emb[(size_t)a * D + k], s);
    }
    table[i] = s;
}

// ------------------------------------------------------------------ forward kernels
template <int MT>
__global__ void __launch_bounds__(kThreads)
k_mlp_forward(mz_packed_mlp M, const float *__restrict__ blob, const float *__restrict__ x, long long ldx,
              const int32_t *__restrict__ x_rows, float *__restrict__ y, long long ldy, int B, int softmax_times,
              int32_t *__restrict__ argmax) {
    extern __shared__ __align__(16) float smem[];
    constexpr int TM = 16 * MT;
    const int ld_in = lds_stride(M.in_features), ld_w = lds_stride(M.max_width);
    lds_f *X = to_lds(smem) + kLdsReserve, *Xs = X + TM * ld_in, *bufA = Xs + TM * ld_in, *bufB = bufA + TM * ld_w;
    const int row0 = blockIdx.x * TM;
    const int rows = min(TM, B - row0);
    const bool ring = uni((int)mlp_ring_ok(M));
    MlpRing<MT> R;
    if (ring) R.begin(M, blob);  // the first layer's weights travel while the rows are loaded
    load_rows_split(X, Xs, ld_in, MT, rows, M.in_features, [&](int r) -> const float * {
        long long src = x_rows ? (long long)x_rows[row0 + r] : (long long)(row0 + r);
        return src < 0 ? nullptr : x + src * ldx;
    });
    int ld_out;
    Team all = whole_workgroup();
    lds_f *out = ring ? R.run(M, blob, Xs, ld_in, bufA, bufB, &ld_out)
                      : mlp_tile<MT>(all, M, 0, blob, Xs, X, ld_in, bufA, bufB, &ld_out);
    const int N = M.out_features;
    // policy heads: the first maximum of each row, then `softmax_times` more softmax passes
    // (src/search/simulation.py:35, src/search/expansion.py:30) before the store
    if (argmax && (int)threadIdx.x < rows) {
        const lds_f *p = out + threadIdx.x * ld_out;
        int best = 0;
        for (int i = 1; i < N; ++i)
            if (p[i] > p[best]) best = i;
        argmax[row0 + threadIdx.x] = best;
    }
    if (y == nullptr) return;
    if (softmax_times > 0) {
        __syncthreads();
        softmax_rows(out, ld_out, TM, N, softmax_times);
    }
    for (int i = threadIdx.x; i < rows * N; i += kThreads) {
        const int r = i / N, c = i - r * N;
        y[(long long)(row0 + r) * ldy + c] = out[r * ld_out + c];
    }
}

// Two independent MLPs in one launch (workgroups [0, blocks0) run the first, the rest the second): heads that
// have their inputs at the same moment of a latency-bound chain share a launch instead of queueing.
struct MlpJob {
    const float *x;
    long long ldx;
    const int32_t *x_rows;
    float *y;
    long long ldy;
    int B;
};
template <int MT>
__device__ __forceinline__ void mlp_rows_forward(const mz_packed_mlp &M, const float *__restrict__ blob, const MlpJob &J, int row0,
                                                 float *smem) {
    constexpr int TM = 16 * MT;
    const int ld_in = lds_stride(M.in_features), ld_w = lds_stride(M.max_width);
    lds_f *X = to_lds(smem) + kLdsReserve, *Xs = X + TM * ld_in, *bufA = Xs + TM * ld_in, *bufB = bufA + TM * ld_w;
    const int rows = min(TM, J.B - row0);
    const bool ring = uni((int)mlp_ring_ok(M));
    MlpRing<MT> R;
    if (ring) R.begin(M, blob);
    const float *x = J.x;
    const int32_t *x_rows = J.x_rows;
    const long long ldx = J.ldx;
    load_rows_split(X, Xs, ld_in, MT, rows, M.in_features, [&](int r) -> const float * {
        long long src = x_rows ? (long long)x_rows[row0 + r] : (long long)(row0 + r);
        return src < 0 ? nullptr : x + src * ldx;
    });
    int ld_out;
    Team all = whole_workgroup();
    lds_f *out = ring ? R.run(M, blob, Xs, ld_in, bufA, bufB, &ld_out)
                      : mlp_tile<MT>(all, M, 0, blob, Xs, X, ld_in, bufA, bufB, &ld_out);
    const int N = M.out_features;
    float *y = J.y;
    const long long ldy = J.ldy;
    for (int i = threadIdx.x; i < rows * N; i += kThreads) {
        const int r = i / N, c = i - r * N;
        y[(long long)(row0 + r) * ldy + c] = out[r * ld_out + c];
    }
}
// (two branches rather than a selected plan: a plan chosen at run time would be copied to per-lane stack memory;
// each job with the row tile its own batch size asks for)
template <int MT0, int MT1>
__global__ void __launch_bounds__(kThreads)
k_mlp_forward_pair(mz_packed_mlp M0, mz_packed_mlp M1, const float *__restrict__ blob, MlpJob j0, MlpJob j1, int blocks0) {
    extern __shared__ __align__(16) float smem[];
    if ((int)blockIdx.x < blocks0)
        mlp_rows_forward<MT0>(M0, blob, j0, (int)blockIdx.x * 16 * MT0, smem);
    else
        mlp_rows_forward<MT1>(M1, blob, j1, ((int)blockIdx.x - blocks0) * 16 * MT1, smem);
}

template <int MT>
__global__ void __launch_bounds__(kThreads)
k_dynamics(mz_packed_nets P, const float *__restrict__ blob, const float *__restrict__ latent, long long ld_latent,
           const int32_t *__restrict__ latent_rows, const int32_t *__restrict__ action,
           float *__restrict__ next_latent, long long ld_next, const int32_t *__restrict__ next_rows,
           float *__restrict__ reward, int B) {
    extern __shared__ __align__(16) float smem[];
    constexpr int TM = 16 * MT;
    const int ld_d = lds_stride(P.D), ld_w = lds_stride(P.reward.max_width);
    // X: the latents as fp32 rows, later the next latents (fp32 rows); Xs / Cs: their split images
    lds_f *X = to_lds(smem) + kLdsReserve, *Xs = X + TM * ld_d, *Cs = Xs + TM * ld_d, *bufA = Cs + TM * ld_d, *bufB = bufA + TM * ld_w;
    lds_f *table = bufB + TM * ld_w;  // [A][Dp] per-action additive table of the merge layer
    lds_i *sel = (lds_i *)(table + P.A * P.Dp);
    const int row0 = blockIdx.x * TM;
    const int rows = min(TM, B - row0);
    for (int i = threadIdx.x; i < P.A * P.Dp; i += kThreads) table[i] = blob[P.action_table_off + i];
    if ((int)threadIdx.x < TM) {
        const int r = row0 + threadIdx.x;
        int a = 0;
        if ((int)threadIdx.x < rows) a = action ? action[r] : (r % P.A);
        sel[threadIdx.x] = min(max(a, 0), P.A - 1);
    }
    load_rows_split(X, Xs, ld_d, MT, rows, P.D, [&](int r) -> const float * {
        long long src = latent_rows ? (long long)latent_rows[row0 + r] : (long long)(row0 + r);
        return src < 0 ? nullptr : latent + src * ld_latent;
    });
    // next = relu(W[:, :D] x + (W[:, D:] e_a + b)): as fp32 rows over X (for the store and a narrow reward
    // layer) and as the image the reward head's first matrix layer reads
    Team all = whole_workgroup();
    {
        const mz_packed_layer Lm = P.merge.layer[0];
        DensePart part = make_part(Lm, X, ld_d, Lm.act, false, table, P.Dp, sel);
        part.Ss = Cs;
        dense_group1<MT>(Xs, ld_d, blob, Lm.Kp, part, all);
        __syncthreads();
    }
    if (next_latent) {
        for (int i = threadIdx.x; i < rows * P.D; i += kThreads) {
            const int r = i / P.D, c = i - r * P.D;
            long long dst = next_rows ? (long long)next_rows[row0 + r] : (long long)(row0 + r);
            if (dst >= 0) next_latent[dst * ld_next + c] = X[r * ld_d + c];
        }
    }
    if (reward) {
        int ld_out;
        const lds_f *out = mlp_tile<MT>(all, P.reward, 0, blob, Cs, X, ld_d, bufA, bufB, &ld_out);
        if ((int)threadIdx.x < rows) reward[row0 + threadIdx.x] = out[threadIdx.x * ld_out];
    }
}

template <int MT>
__global__ void __launch_bounds__(kThreads)
k_prediction(mz_packed_nets P, const float *__restrict__ blob, const float *__restrict__ latent, long long ld_latent,
             const int32_t *__restrict__ latent_rows, int softmax_times, float *__restrict__ policy,
             float *__restrict__ value, int32_t *__restrict__ argmax, int B) {
    extern __shared__ __align__(16) float smem[];
    constexpr int TM = 16 * MT;
    const int ld_d = lds_stride(P.D);
    const int ld_w = lds_stride(max(P.value.max_width, P.policy.max_width));
    lds_f *X = to_lds(smem) + kLdsReserve, *Xs = X + TM * ld_d, *bufA = Xs + TM * ld_d, *bufB = bufA + TM * ld_w;
    const int row0 = blockIdx.x * TM;
    const int rows = min(TM, B - row0);
    load_rows_split(X, Xs, ld_d, MT, rows, P.D, [&](int r) -> const float * {
        long long src = latent_rows ? (long long)latent_rows[row0 + r] : (long long)(row0 + r);
        return src < 0 ? nullptr : latent + src * ld_latent;
    });
    int ld_out;
    Team all = whole_workgroup();
    if (policy || argmax) {
        lds_f *out = mlp_tile<MT>(all, P.policy, 0, blob, Xs, X, ld_d, bufA, bufB, &ld_out);
        if (argmax && (int)threadIdx.x < rows) {
            const lds_f *p = out + threadIdx.x * ld_out;
            int best = 0;
            for (int i = 1; i < P.A; ++i)
                if (p[i] > p[best]) best = i;
            argmax[row0 + threadIdx.x] = best;
        }
        if (policy) {
            if (softmax_times > 0) {
                __syncthreads();
                softmax_rows(out, ld_out, TM, P.A, softmax_times);
            }
            for (int i = threadIdx.x; i < rows * P.A; i += kThreads) {
                const int r = i / P.A, c = i - r * P.A;
                policy[(long long)(row0 + r) * P.A + c] = out[r * ld_out + c];
            }
        }
        __syncthreads();
    }
    if (value) {
        const lds_f *out = mlp_tile<MT>(all, P.value, 0, blob, Xs, X, ld_d, bufA, bufB, &ld_out);
        if ((int)threadIdx.x < rows) value[row0 + threadIdx.x] = out[threadIdx.x * ld_out];
    }
}

// ------------------------------------------------------------------ host helpers
int validate_mlp(const mz_mlp *m, const char *what) {
    if (!m || m->n_layers <= 0 || m->n_layers > MZ_MAX_LAYERS)
        return mz::fail(MZ_ERR_ARG, "%s: n_layers must be 1..%d", what, MZ_MAX_LAYERS);
    for (int i = 0; i < m->n_layers; ++i) {
        const mz_dense &d = m->layer[i];
        if (d.in_features <= 0 || d.out_features <= 0 || !d.weight)
            return mz::fail(MZ_ERR_ARG, "%s: layer %d has bad dims or NULL weight", what, i);
        if (i > 0 && d.in_features != m->layer[i - 1].out_features)
            return mz::fail(MZ_ERR_ARG, "%s: layer %d in_features %d != previous out_features %d", what, i,
                            d.in_features, m->layer[i - 1].out_features);
        if (d.activation < MZ_ACT_NONE || d.activation > MZ_ACT_SOFTMAX)
            return mz::fail(MZ_ERR_ARG, "%s: layer %d has unknown activation %d", what, i, d.activation);
        if (d.out_features > 4096 || d.in_features > 4096)
            return mz::fail(MZ_ERR_ARG, "%s: layer %d wider than 4096 does not fit the LDS-resident tile", what, i);
    }
    return MZ_OK;
}

int pack_layer(const mz_dense &d, const mz_packed_layer &L, float *blob, hipStream_t st) {
    const long long total = (long long)L.Np * L.Kp;
    if (L.fmt != 0) {
        const int nb = (int)std::min<long long>((total / 8 + 255) / 256 + 1, 4096);
        hipLaunchKernelGGL(k_pack_layer_split, dim3(nb), dim3(256), 0, st, d.weight, d.ldw > 0 ? d.ldw : d.in_features, d.bias,
                           L.K, L.N, L.Kp >> 5, L.Np, blob + L.w_off, blob + L.b_off);
        return mz::check_launch("k_pack_layer_split");
    }
    const int nb = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pack_layer, dim3(nb), dim3(256), 0, st, d.weight, d.ldw > 0 ? d.ldw : d.in_features, d.bias,
                       L.K, L.N, L.Kp, L.Np, blob + L.w_off, blob + L.b_off);
    return mz::check_launch("k_pack_layer");
}

// Packed layout of an MLP; `all_matrix`: every layer in the matrix form whatever its width (the merge layer,
// which the kernels always evaluate on the matrix cores).
int64_t plan_mlp(const mz_mlp *h_mlp, int64_t base_off, mz_packed_mlp *h_out, bool all_matrix);

constexpr size_t kMaxLds = 160 * 1024;

// Dynamic LDS above 64 KiB has to be enabled per kernel.
template <typename K>
int allow_lds(K kernel, size_t bytes, const char *what) {
    if (bytes <= 64 * 1024) return MZ_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "%s: cannot enable %zu bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    return MZ_OK;
}

// rows from which the 32-row tile is used (MZ_MLP_MT2_MIN_ROWS: a tuning aid)
int mt2_min_rows() {
    static const int v = [] { const char *e = getenv("MZ_MLP_MT2_MIN_ROWS"); return e && atoi(e) > 0 ? atoi(e) : 16 * 512; }();
    return v;
}

template <typename Launch1, typename Launch2>
int launch_by_rows(int B, size_t lds_per_row_floats, size_t extra_bytes, Launch1 &&l1, Launch2 &&l2) {
    // 32-row tiles halve the weight traffic per row; 16-row tiles keep all 256 CUs busy when
    // the batch is small.
    extra_bytes += kLdsReserve * sizeof(float);
    int mt = (B >= mt2_min_rows()) ? 2 : 1;
    if ((size_t)mt * 16 * lds_per_row_floats * sizeof(float) + extra_bytes > kMaxLds) mt = 1;
    size_t lds = (size_t)mt * 16 * lds_per_row_floats * sizeof(float) + extra_bytes;
    if (lds > kMaxLds) return mz::fail(MZ_ERR_ARG, "network too wide: needs %zu bytes of LDS per tile", lds);
    return mt == 2 ? l2(lds) : l1(lds);
}

}  // namespace

extern "C" {

constexpr int64_t kPackedSlackFloats = 1024;

int64_t mz_mlp_plan(const mz_mlp *h_mlp, int64_t base_off, mz_packed_mlp *h_out) { return plan_mlp(h_mlp, base_off, h_out, false); }

}  // extern "C"

namespace {
int64_t plan_mlp(const mz_mlp *h_mlp, int64_t base_off, mz_packed_mlp *h_out, bool all_matrix) {
    if (validate_mlp(h_mlp, "mz_mlp_plan") != MZ_OK) return MZ_ERR_ARG;
    if (!h_out || base_off < 0 || (base_off & 3))
        return mz::fail(MZ_ERR_ARG, "mz_mlp_plan: h_out is NULL or base_off is not a non-negative multiple of 4");
    std::memset(h_out, 0, sizeof(*h_out));
    int64_t off = base_off;
    h_out->n_layers = h_mlp->n_layers;
    h_out->in_features = h_mlp->layer[0].in_features;
    h_out->out_features = h_mlp->layer[h_mlp->n_layers - 1].out_features;
    for (int i = 0; i < h_mlp->n_layers; ++i) {
        const mz_dense &d = h_mlp->layer[i];
        mz_packed_layer &L = h_out->layer[i];
        L.K = d.in_features;
        L.N = d.out_features;
        // every layer runs on the matrix cores, except a final one-output layer behind a wide layer: that pair is
        // folded (tail_foldable), the epilogue of the wide layer reads the output layer's weights as fp32
        const bool folded_tail = !all_matrix && i == h_mlp->n_layers - 1 && i >= 1 && d.out_features == 1 &&
                                 h_mlp->layer[i - 1].out_features > 8 && d.in_features == h_mlp->layer[i - 1].out_features;
        L.fmt = folded_tail ? 0 : 1;
        L.Kp = L.fmt ? ceil32(L.K) : ceil16(L.K);
        L.Np = ceil16(L.N);
        L.act = d.activation;
        L.w_off = off;
        off += (int64_t)L.Kp * L.Np;
        L.b_off = off;
        off += L.Np;
        h_out->max_width = std::max(h_out->max_width, L.N);
    }
    // read slack: the kernels fetch whole batches of k-steps and may run up to 3 KiB past a
    // layer's last weight tile (values that feed no arithmetic)
    off += kPackedSlackFloats;
    return off - base_off;
}
}  // namespace

extern "C" {

int mz_mlp_pack(const mz_mlp *h_mlp, const mz_packed_mlp *h_plan, float *blob, void *stream) {
    MZ_TRY(validate_mlp(h_mlp, "mz_mlp_pack"));
    if (!h_plan || !blob || h_plan->n_layers != h_mlp->n_layers)
        return mz::fail(MZ_ERR_ARG, "mz_mlp_pack: plan does not match the network");
    for (int i = 0; i < h_mlp->n_layers; ++i) MZ_TRY(pack_layer(h_mlp->layer[i], h_plan->layer[i], blob, (hipStream_t)stream));
    return MZ_OK;
}

int mz_mlp_forward_post(const mz_packed_mlp *h_plan, const float *blob, const float *x, int64_t ldx,
                        const int32_t *x_rows, float *y, int64_t ldy, int32_t B, int32_t softmax_times,
                        int32_t *argmax, void *stream) {
    if (!h_plan || !blob || !x || (!y && !argmax) || B < 0 || ldx < h_plan->in_features ||
        (y && ldy < h_plan->out_features) || softmax_times < 0)
        return mz::fail(MZ_ERR_ARG, "mz_mlp_forward: NULL argument, negative B or softmax_times, or row stride below "
                        "the row width");
    if (B == 0) return MZ_OK;
    hipStream_t st = (hipStream_t)stream;
    size_t per_row = 2 * (size_t)lds_stride(h_plan->in_features) + 2 * (size_t)lds_stride(h_plan->max_width);
    return launch_by_rows(
        B, per_row, 0,
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_mlp_forward<1>, lds, "k_mlp_forward"));
            hipLaunchKernelGGL(k_mlp_forward<1>, dim3((B + 15) / 16), dim3(kThreads), lds, st, *h_plan, blob, x,
                               (long long)ldx, x_rows, y, (long long)ldy, B, softmax_times, argmax);
            return mz::check_launch("k_mlp_forward<1>");
        },
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_mlp_forward<2>, lds, "k_mlp_forward"));
            hipLaunchKernelGGL(k_mlp_forward<2>, dim3((B + 31) / 32), dim3(kThreads), lds, st, *h_plan, blob, x,
                               (long long)ldx, x_rows, y, (long long)ldy, B, softmax_times, argmax);
            return mz::check_launch("k_mlp_forward<2>");
        });
}

int mz_mlp_forward_pair(const mz_packed_mlp *h_plan0, const float *x0, int64_t ldx0, const int32_t *x_rows0, float *y0,
                        int64_t ldy0, int32_t B0, const mz_packed_mlp *h_plan1, const float *x1, int64_t ldx1,
                        const int32_t *x_rows1, float *y1, int64_t ldy1, int32_t B1, const float *blob, void *stream) {
    if (!h_plan0 || !h_plan1 || !blob || !x0 || !x1 || !y0 || !y1 || B0 < 0 || B1 < 0 || ldx0 < h_plan0->in_features ||
        ldx1 < h_plan1->in_features || ldy0 < h_plan0->out_features || ldy1 < h_plan1->out_features)
        return mz::fail(MZ_ERR_ARG, "mz_mlp_forward_pair: NULL argument, negative B, or row stride below the row width");
    if (B0 == 0 || B1 == 0) {
        MZ_TRY(mz_mlp_forward(h_plan0, blob, x0, ldx0, x_rows0, y0, ldy0, B0, stream));
        return mz_mlp_forward(h_plan1, blob, x1, ldx1, x_rows1, y1, ldy1, B1, stream);
    }
    auto per_row = [](const mz_packed_mlp *p) { return 2 * (size_t)lds_stride(p->in_features) + 2 * (size_t)lds_stride(p->max_width); };
    // row tile per job as mz_mlp_forward picks it: 32 rows from 8192 rows on (half the weight traffic per row)
    auto tile = [&](const mz_packed_mlp *p, int B) {
        return (B >= mt2_min_rows() && (32 * per_row(p) + kLdsReserve) * sizeof(float) <= kMaxLds) ? 2 : 1;
    };
    const int mt0 = tile(h_plan0, B0), mt1 = tile(h_plan1, B1);
    const size_t lds = (16 * std::max(mt0 * per_row(h_plan0), mt1 * per_row(h_plan1)) + kLdsReserve) * sizeof(float);
    if (lds > kMaxLds) return mz::fail(MZ_ERR_ARG, "network too wide: needs %zu bytes of LDS per tile", lds);
    const int blocks0 = (B0 + 16 * mt0 - 1) / (16 * mt0), blocks1 = (B1 + 16 * mt1 - 1) / (16 * mt1);
    auto launch = [&](auto kernel) -> int {
        MZ_TRY(allow_lds(kernel, lds, "k_mlp_forward_pair"));
        hipLaunchKernelGGL(kernel, dim3(blocks0 + blocks1), dim3(kThreads), lds, (hipStream_t)stream, *h_plan0, *h_plan1, blob,
                           MlpJob{x0, (long long)ldx0, x_rows0, y0, (long long)ldy0, B0},
                           MlpJob{x1, (long long)ldx1, x_rows1, y1, (long long)ldy1, B1}, blocks0);
        return MZ_OK;
    };
    if (mt0 == 1 && mt1 == 1) MZ_TRY(launch(k_mlp_forward_pair<1, 1>));
    else if (mt0 == 2 && mt1 == 1) MZ_TRY(launch(k_mlp_forward_pair<2, 1>));
    else if (mt0 == 1 && mt1 == 2) MZ_TRY(launch(k_mlp_forward_pair<1, 2>));
    else MZ_TRY(launch(k_mlp_forward_pair<2, 2>));
    return mz::check_launch("k_mlp_forward_pair");
}

int mz_mlp_forward(const mz_packed_mlp *h_plan, const float *blob, const float *x, int64_t ldx,
                   const int32_t *x_rows, float *y, int64_t ldy, int32_t B, void *stream) {
    if (!y) return mz::fail(MZ_ERR_ARG, "mz_mlp_forward: NULL output");
    return mz_mlp_forward_post(h_plan, blob, x, ldx, x_rows, y, ldy, B, 0, nullptr, stream);
}

// ---- one layer of any width (read by the chunked kernels of mz_tower.hip)
static int validate_wide(const mz_dense *d, const char *what) {
    if (!d || d->in_features <= 0 || d->out_features <= 0 || !d->weight)
        return mz::fail(MZ_ERR_ARG, "%s: NULL layer, bad dims or NULL weight", what);
    if (d->activation < MZ_ACT_NONE || d->activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "%s: activation %d is not element-wise", what, d->activation);
    if (d->ldw != 0 && d->ldw < d->in_features) return mz::fail(MZ_ERR_ARG, "%s: ldw below in_features", what);
    return MZ_OK;
}

int64_t mz_wide_plan(const mz_dense *h_layer, int64_t base_off, mz_packed_layer *h_out) {
    if (validate_wide(h_layer, "mz_wide_plan") != MZ_OK) return MZ_ERR_ARG;
    if (!h_out || base_off < 0 || (base_off & 3))
        return mz::fail(MZ_ERR_ARG, "mz_wide_plan: h_out is NULL or base_off is not a non-negative multiple of 4");
    std::memset(h_out, 0, sizeof(*h_out));
    h_out->K = h_layer->in_features;
    h_out->N = h_layer->out_features;
    h_out->Kp = ceil16(h_out->K);
    h_out->Np = ceil16(h_out->N);
    h_out->act = h_layer->activation;
    h_out->w_off = base_off;
    h_out->b_off = base_off + (int64_t)h_out->Kp * h_out->Np;
    return (int64_t)h_out->Kp * h_out->Np + h_out->Np + kPackedSlackFloats;
}

int mz_wide_pack(const mz_dense *h_layer, const mz_packed_layer *h_plan, float *blob, void *stream) {
    MZ_TRY(validate_wide(h_layer, "mz_wide_pack"));
    if (!h_plan || !blob || h_plan->K != h_layer->in_features || h_plan->N != h_layer->out_features)
        return mz::fail(MZ_ERR_ARG, "mz_wide_pack: plan does not match the layer");
    return pack_layer(*h_layer, *h_plan, blob, (hipStream_t)stream);
}

int mz_action_table(const float *embedding, const float *merge_weight, const float *merge_bias, int32_t D,
                    int32_t A, float *table, void *stream) {
    if (!embedding || !merge_weight || !table || D <= 0 || A <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_action_table: NULL argument or non-positive D, A");
    hipLaunchKernelGGL(k_action_table, dim3((unsigned)(((long long)A * D + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, embedding, merge_weight, merge_bias, D, D, A, table);
    return mz::check_launch("k_action_table");
}

int mz_nets_plan(const mz_dynamics_net *h_dyn, const mz_prediction_net *h_pred, mz_packed_nets *h_out) {
    if (!h_dyn || !h_pred || !h_out) return mz::fail(MZ_ERR_ARG, "mz_nets_plan: NULL argument");
    if (h_dyn->D <= 0 || h_dyn->A <= 0 || h_dyn->A > 64 || h_pred->D != h_dyn->D || h_pred->A != h_dyn->A)
        return mz::fail(MZ_ERR_ARG, "mz_nets_plan: dynamics (D=%d, A=%d) and prediction (D=%d, A=%d) dims disagree",
                        h_dyn->D, h_dyn->A, h_pred->D, h_pred->A);
    MZ_TRY(validate_mlp(&h_dyn->reward, "reward_mlp"));
    MZ_TRY(validate_mlp(&h_pred->value, "value_mlp"));
    MZ_TRY(validate_mlp(&h_pred->policy, "policy_mlp"));
    const int D = h_dyn->D, A = h_dyn->A;
    if (h_dyn->reward.layer[0].in_features != D || h_pred->value.layer[0].in_features != D ||
        h_pred->policy.layer[0].in_features != D)
        return mz::fail(MZ_ERR_ARG, "mz_nets_plan: a head's input width differs from D=%d", D);
    if (h_dyn->reward.layer[h_dyn->reward.n_layers - 1].out_features != 1 ||
        h_pred->value.layer[h_pred->value.n_layers - 1].out_features != 1 ||
        h_pred->policy.layer[h_pred->policy.n_layers - 1].out_features != A)
        return mz::fail(MZ_ERR_ARG, "mz_nets_plan: heads must end in 1 (reward, value) and A (policy) outputs");
    if (!h_dyn->embedding || !h_dyn->merge_weight) return mz::fail(MZ_ERR_ARG, "mz_nets_plan: NULL dynamics weight");
    std::memset(h_out, 0, sizeof(*h_out));
    h_out->D = D;
    h_out->A = A;
    h_out->Dp = ceil16(D);
    mz_mlp merge;
    std::memset(&merge, 0, sizeof(merge));
    merge.n_layers = 1;
    merge.layer[0] = mz_dense{h_dyn->merge_weight, nullptr, D, D, MZ_ACT_RELU, 2 * D};
    int64_t off = 0, n;
    if ((n = plan_mlp(&merge, off, &h_out->merge, true)) < 0) return (int)n;
    off += n;
    h_out->action_table_off = off;
    off += (int64_t)A * h_out->Dp;
    if ((n = mz_mlp_plan(&h_dyn->reward, off, &h_out->reward)) < 0) return (int)n;
    off += n;
    if ((n = mz_mlp_plan(&h_pred->value, off, &h_out->value)) < 0) return (int)n;
    off += n;
    if ((n = mz_mlp_plan(&h_pred->policy, off, &h_out->policy)) < 0) return (int)n;
    off += n;
    h_out->total_floats = off;
    return MZ_OK;
}

int mz_nets_pack(const mz_dynamics_net *h_dyn, const mz_prediction_net *h_pred, const mz_packed_nets *h_plan,
                 float *blob, void *stream) {
    if (!h_dyn || !h_pred || !h_plan || !blob) return mz::fail(MZ_ERR_ARG, "mz_nets_pack: NULL argument");
    if (h_plan->D != h_dyn->D || h_plan->A != h_dyn->A) return mz::fail(MZ_ERR_ARG, "mz_nets_pack: plan/network mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int D = h_dyn->D, A = h_dyn->A;
    mz_dense merge{h_dyn->merge_weight, nullptr, D, D, MZ_ACT_RELU, 2 * D};
    MZ_TRY(pack_layer(merge, h_plan->merge.layer[0], blob, st));
    hipLaunchKernelGGL(k_action_table, dim3((A * h_plan->Dp + 255) / 256), dim3(256), 0, st, h_dyn->embedding,
                       h_dyn->merge_weight, h_dyn->merge_bias, D, h_plan->Dp, A, blob + h_plan->action_table_off);
    MZ_TRY(mz::check_launch("k_action_table"));
    MZ_TRY(mz_mlp_pack(&h_dyn->reward, &h_plan->reward, blob, stream));
    MZ_TRY(mz_mlp_pack(&h_pred->value, &h_plan->value, blob, stream));
    MZ_TRY(mz_mlp_pack(&h_pred->policy, &h_plan->policy, blob, stream));
    return MZ_OK;
}

int mz_dynamics_forward(const mz_packed_nets *h_plan, const float *blob, const float *latent, int64_t ld_latent,
                        const int32_t *latent_rows, const int32_t *action, float *next_latent, int64_t ld_next,
                        const int32_t *next_rows, float *reward, int32_t B, void *stream) {
    if (!h_plan || !blob || !latent || B < 0 || ld_latent < h_plan->D || (next_latent && ld_next < h_plan->D))
        return mz::fail(MZ_ERR_ARG, "mz_dynamics_forward: NULL argument, negative B or row stride below D");
    if (B == 0) return MZ_OK;
    hipStream_t st = (hipStream_t)stream;
    size_t per_row = 3 * (size_t)lds_stride(h_plan->D) + 2 * (size_t)lds_stride(h_plan->reward.max_width);
    return launch_by_rows(
        B, per_row, 32 * sizeof(int) + sizeof(float) * (size_t)h_plan->A * h_plan->Dp,
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_dynamics<1>, lds, "k_dynamics"));
            hipLaunchKernelGGL(k_dynamics<1>, dim3((B + 15) / 16), dim3(kThreads), lds, st, *h_plan, blob, latent,
                               (long long)ld_latent, latent_rows, action, next_latent, (long long)ld_next, next_rows,
                               reward, B);
            return mz::check_launch("k_dynamics<1>");
        },
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_dynamics<2>, lds, "k_dynamics"));
            hipLaunchKernelGGL(k_dynamics<2>, dim3((B + 31) / 32), dim3(kThreads), lds, st, *h_plan, blob, latent,
                               (long long)ld_latent, latent_rows, action, next_latent, (long long)ld_next, next_rows,
                               reward, B);
            return mz::check_launch("k_dynamics<2>");
        });
}

int mz_prediction_forward(const mz_packed_nets *h_plan, const float *blob, const float *latent, int64_t ld_latent,
                          const int32_t *latent_rows, int32_t softmax_times, float *policy, float *value,
                          int32_t *argmax, int32_t B, void *stream) {
    if (!h_plan || !blob || !latent || B < 0 || ld_latent < h_plan->D || softmax_times < 0)
        return mz::fail(MZ_ERR_ARG, "mz_prediction_forward: NULL argument, negative B or row stride below D");
    if (B == 0 || (!policy && !value && !argmax)) return MZ_OK;
    hipStream_t st = (hipStream_t)stream;
    size_t per_row = 2 * (size_t)lds_stride(h_plan->D) +
                     2 * (size_t)lds_stride(std::max(h_plan->value.max_width, h_plan->policy.max_width));
    return launch_by_rows(
        B, per_row, 0,
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_prediction<1>, lds, "k_prediction"));
            hipLaunchKernelGGL(k_prediction<1>, dim3((B + 15) / 16), dim3(kThreads), lds, st, *h_plan, blob, latent,
                               (long long)ld_latent, latent_rows, softmax_times, policy, value, argmax, B);
            return mz::check_launch("k_prediction<1>");
        },
        [&](size_t lds) {
            MZ_TRY(allow_lds(k_prediction<2>, lds, "k_prediction"));
            hipLaunchKernelGGL(k_prediction<2>, dim3((B + 31) / 32), dim3(kThreads), lds, st, *h_plan, blob, latent,
                               (long long)ld_latent, latent_rows, softmax_times, policy, value, argmax, B);
            return mz::check_launch("k_prediction<2>");
        });
}

}  // extern "C"
