// Device-side building blocks of the fused fp32 MLP kernels (gfx950).
//
// A workgroup of kWaves wavefronts owns a tile of TM = 16*MT rows whose activations live in
// LDS; one dense layer is   Y[TM][N] = act(X[TM][K] * W^T + b)   on v_mfma_f32_16x16x4_f32
// (exact fp32 multiply-add chains, so results differ from another fp32 GEMM only by
// summation order).
//
// Weights are re-tiled once per weight update ("packed"): for a layer with K inputs and N
// outputs, Kp = ceil16(K), Np = ceil16(N), tile (nt, kt) is 64 float4:
//     lane l, element j  =  W[nt*16 + (l & 15)][kt*16 + 4*(l >> 4) + j]      (0 outside W)
// so a wavefront fetches the B operands of four consecutive MFMA k-steps with one fully
// coalesced 1 KiB global_load_dwordx4, and the matching A operands with one ds_read_b128 per
// 16-row tile.  Lane l of MFMA k-step j then multiplies X[row l&15][kt*16 + 4*(l>>4) + j]
// by W[col l&15][same k]: both operands agree on k, which is all the instruction needs.
// Bias is stored padded to Np behind the tiles.
#pragma once
#include <hip/hip_runtime.h>

#include "muzero_hip.h"

namespace mzd {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

using f32x4 = __attribute__((ext_vector_type(4))) float;

__host__ __device__ inline int ceil16(int x) { return (x + 15) & ~15; }
// LDS row stride (floats) for a buffer holding rows of `width` valid floats: padded to a
// multiple of 16 (so a K-loop may read whole 16-float chunks) plus 4 floats of skew.
__host__ __device__ inline int lds_stride(int width) { return ceil16(width) + 4; }

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case MZ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case MZ_ACT_TANH: return tanhf(v);
        case MZ_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

// One dense layer over the workgroup's row tile.  Xs: LDS [16*MT][ldx], columns [K, Kp)
// must be zero.  Ys: LDS [16*MT][ldy]; columns [N, Np) are written as act(0 + 0) of the
// zero-padded weights, i.e. 0 for none/relu/tanh -- callers that chain a sigmoid layer
// must not rely on the padding (sigmoid(0) = 0.5), so the padding is re-zeroed here.
// `row_add` (optional, global memory): per-row additive table row_add[row_sel[r]][n] added
// before the activation (the action-embedding contribution of the dynamics merge layer).
// Ends with __syncthreads(): Ys is complete and Xs may be overwritten afterwards.
template <int MT>
__device__ __forceinline__ void dense_tile(const float *Xs, int ldx, const float *__restrict__ packed,
                                           const mz_packed_layer &L, float *Ys, int ldy,
                                           const float *__restrict__ row_add, int row_add_ld, const int *row_sel) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int KT = L.Kp >> 4, NT = L.Np >> 4;
    const float4 *__restrict__ Wp = reinterpret_cast<const float4 *>(packed + L.w_off);
    const float *__restrict__ bias = packed + L.b_off;
    for (int nt = wave; nt < NT; nt += kWaves) {
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 *wp = Wp + (size_t)nt * KT * 64 + lane;
        const float *xp = Xs + m * ldx + 4 * kq;
#pragma unroll 2
        for (int kt = 0; kt < KT; ++kt) {
            const float4 w = wp[(size_t)kt * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 x = *reinterpret_cast<const float4 *>(xp + mt * 16 * ldx + kt * 16);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, w.x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, w.y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, w.z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, w.w, acc[mt], 0, 0, 0);
            }
        }
        // C layout: column = lane & 15, row = 4*(lane >> 4) + reg
        const int col = nt * 16 + m;
        const float b = bias[col];
        const bool real_col = col < L.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + kq * 4 + r;
                float v = acc[mt][r] + b;
                if (row_add != nullptr && real_col) v += row_add[(size_t)row_sel[row] * row_add_ld + col];
                Ys[row * ldy + col] = real_col ? activate(v, L.act) : 0.0f;
            }
        }
    }
    __syncthreads();
}

// In-place softmax over the first N columns of each of `rows` LDS rows, applied `times`
// times (torch.softmax semantics: subtract the row maximum, exp, normalise).  One lane per
// row; N is the action count (small).  Ends with __syncthreads().
__device__ __forceinline__ void softmax_rows(float *Ys, int ldy, int rows, int N, int times) {
    if ((int)threadIdx.x < rows) {
        float *y = Ys + threadIdx.x * ldy;
        for (int t = 0; t < times; ++t) {
            float mx = y[0];
            for (int i = 1; i < N; ++i) mx = fmaxf(mx, y[i]);
            float s = 0.0f;
            for (int i = 0; i < N; ++i) {
                float e = expf(y[i] - mx);
                y[i] = e;
                s += e;
            }
            for (int i = 0; i < N; ++i) y[i] = y[i] / s;
        }
    }
    __syncthreads();
}

// Runs every layer of `M` over the row tile, ping-ponging between two LDS buffers.
// Returns the buffer that holds the output.  A final MZ_ACT_SOFTMAX layer is computed as a
// linear layer followed by softmax_rows.
template <int MT>
__device__ __forceinline__ float *mlp_tile(const mz_packed_mlp &M, const float *__restrict__ packed, float *in_buf,
                                           int in_ld, float *buf_a, float *buf_b, int *out_ld) {
    float *src = in_buf;
    int ld = in_ld;
    float *dst = buf_a;
    for (int i = 0; i < M.n_layers; ++i) {
        const mz_packed_layer &L = M.layer[i];
        const int ldo = lds_stride(L.N);
        dense_tile<MT>(src, ld, packed, L, dst, ldo, nullptr, 0, nullptr);
        if (L.act == MZ_ACT_SOFTMAX) softmax_rows(dst, ldo, 16 * MT, L.N, 1);
        src = dst;
        ld = ldo;
        dst = (dst == buf_a) ? buf_b : buf_a;
    }
    *out_ld = ld;
    return src;
}

// Cooperative load of up to `rows` global rows (row r from src + row_index(r)*src_ld) into
// an LDS tile [TM][ld], zero-filling columns [K, ld) and rows >= rows.
template <typename RowPtr>
__device__ __forceinline__ void load_rows(float *Xs, int ld, int TM, int rows, int K, RowPtr row_ptr) {
    for (int i = threadIdx.x; i < TM * ld; i += kThreads) {
        const int r = i / ld, c = i - r * ld;
        float v = 0.0f;
        if (r < rows && c < K) {
            const float *p = row_ptr(r);
            if (p) v = p[c];
        }
        Xs[i] = v;
    }
    __syncthreads();
}

}  // namespace mzd
