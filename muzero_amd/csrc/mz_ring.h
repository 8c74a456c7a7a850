// A whole MLP for 16 * MT rows by one workgroup with the weights flowing through a ring of two register slots
// per wave, a whole column tile (all its k-chunks, both fp16 planes: up to 16 KiB per wave) per slot.
//
// mlp_tile (mz_mlp_device.h) streams a layer's weights a few k-chunks at a time and starts every layer with a
// round trip to L2: for the small heads of the lock-step searches (a few hundred rows, three layers) the call is
// that latency chain and nothing else.  Here a wave owns at most two column tiles per layer, requests both in full
// before it needs them, and requests the NEXT layer's tiles as soon as a slot has been multiplied -- in flight
// through the epilogue and the barrier between the layers.  Biases and the folded one-output layer's weights are
// requested in front of the tile they belong to (loads return in order).
//
// Per accumulator the instruction sequence is stream_tiles' and the epilogue arithmetic store_tile_core's: the
// results equal mlp_tile's bit for bit.  Covers MLPs whose matrix layers have at most 16 column tiles (two per
// wave) and 8 k-chunks, relu or no activation after the hidden layers; mlp_ring_ok() says whether a plan qualifies.
#pragma once
#include "mz_mlp_device.h"

namespace mzd {

constexpr int kRingChunks = 8;  // k-chunks a slot holds (Kp <= 256)

struct RingSlot {
    f32x4 w[kRingChunks][2];
};

__host__ __device__ inline bool mlp_ring_ok(const mz_packed_mlp &M) {
    const int n = M.n_layers;
    if (n < 1 || n > MZ_MAX_LAYERS) return false;
    const bool fold = tail_foldable(M);
    const int matrix_layers = fold ? n - 1 : n;
    for (int i = 0; i < matrix_layers; ++i) {
        const mz_packed_layer &L = M.layer[i];
        if (!is_matrix_layer(L) || (L.Np >> 4) > 2 * kWaves || (L.Kp >> 5) > kRingChunks) return false;
        const bool last = i == n - 1;
        if (!last && L.act != MZ_ACT_RELU && L.act != MZ_ACT_NONE) return false;
        if (last && L.act != MZ_ACT_NONE && L.act != MZ_ACT_SOFTMAX && L.act != MZ_ACT_RELU) return false;
    }
    return true;
}

// kc chunks of a column tile -> slot (the branches are wave-uniform).
__device__ __forceinline__ void ring_fetch(RingSlot &s, const float *tile, int kc, int lane) {
    glb_f4 *b = (glb_f4 *)tile + lane;
#pragma unroll
    for (int c = 0; c < kRingChunks; ++c)
        if (c < kc) {
            s.w[c][0] = b[(2 * c) * 64];
            s.w[c][1] = b[(2 * c + 1) * 64];
        }
}

// acc = tile x rows (the rows' split image at img: this lane's 16 bytes of chunk 0 of row tile 0; row tiles x_tile
// floats apart; the lo planes kc chunks behind the hi planes).
template <int MT>
__device__ __forceinline__ void ring_mma(const RingSlot &s, const lds_f *img, int x_tile, int kc, f32x4 (&hi)[MT], f32x4 (&lo)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hi[mt] = lo[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < kRingChunks; ++c)
        if (c < kc) {
            const h8 wh = __builtin_bit_cast(h8, s.w[c][0]), wl = __builtin_bit_cast(h8, s.w[c][1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const lds_f *p = img + mt * x_tile + c * kChunkFloats;
                const h8 ah = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p));
                const h8 al = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p + kc * kChunkFloats));
                hi[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, hi[mt], 0, 0, 0);
                lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, lo[mt], 0, 0, 0);
                lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, lo[mt], 0, 0, 0);
            }
        }
}

__device__ __forceinline__ f32x4 ring_value(const f32x4 &hi, const f32x4 &lo, const f32x4 &b, bool relu) {
    f32x4 v;
    v.x = (hi.x + __fmul_rn(lo.x, kLoUnscale)) + b.x;
    v.y = (hi.y + __fmul_rn(lo.y, kLoUnscale)) + b.y;
    v.z = (hi.z + __fmul_rn(lo.z, kLoUnscale)) + b.z;
    v.w = (hi.w + __fmul_rn(lo.w, kLoUnscale)) + b.w;
    if (relu) {
        v.x = v.x > 0.0f ? v.x : 0.0f;
        v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f;
        v.w = v.w > 0.0f ? v.w : 0.0f;
    }
    return v;
}

// One column tile's epilogue.  kind 0: split image of the next layer's operand; 1: fp32 rows; 2: folded dot.
template <int MT>
__device__ __forceinline__ void ring_epilogue(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, int nt_count, const f32x4 &b,
                                              const f32x4 &w2, bool relu, int kind, lds_f *dst, int ldo, int s_kc, int lane) {
    const int j = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 v = ring_value(hi[mt], lo[mt], b, relu);
        if (kind == 0) {
            f32x2 h, l;
            split4(v, h, l);
            lds_f *img = dst + mt * 16 * ldo + (nt >> 1) * kChunkFloats + (2 * (nt & 1) + (rq >> 1)) * 64 + j * 4 + 2 * (rq & 1);
            *reinterpret_cast<lds_f2 *>(img) = h;
            *reinterpret_cast<lds_f2 *>(img + s_kc * kChunkFloats) = l;
            if ((nt & 1) == 0 && nt + 1 == nt_count) {  // an odd number of column tiles: the chunk's upper half is zero
                const f32x2 z = {0.f, 0.f};
                *reinterpret_cast<lds_f2 *>(img + 128) = z;
                *reinterpret_cast<lds_f2 *>(img + 128 + s_kc * kChunkFloats) = z;
            }
        } else if (kind == 1) {
            *reinterpret_cast<lds_f4 *>(dst + (mt * 16 + j) * ldo + nt * 16 + 4 * rq) = v;
        } else {
            float s = __fmul_rn(v.x, w2.x);
            s += __fmul_rn(v.y, w2.y);
            s += __fmul_rn(v.z, w2.z);
            s += __fmul_rn(v.w, w2.w);
            dst[(nt * 4 + rq) * (16 * MT) + mt * 16 + j] = s;
        }
    }
}

// mlp_tile for a plan mlp_ring_ok() accepts, in two steps so that the first layer's weights travel while the caller
// loads the input rows: begin() requests them (call it first, before anything that waits on memory), run() does the
// rest.  run(): `src` is the split image of the input rows (row stride ld); returns the buffer that holds the output
// (fp32 rows of stride *out_ld; a folded tail leaves row r's value at out[r]).  Every wave of the workgroup must
// call both.
template <int MT>
struct MlpRing {
    RingSlot s0, s1;
    f32x4 bias0, bias1, w20, w21;
    int wave, lane;

    // this wave's column tiles of a layer with `total` tiles, dealt in pairs (as dense_group_body deals them)
    __device__ __forceinline__ void share(int total, int &first, int &count) const {
        const int units = (total + 1) >> 1;
        const int sh = units >> 3, extra = units - (sh << 3);
        first = (wave * sh + min(wave, extra)) * 2;
        const int groups = sh + (wave < extra ? 1 : 0);
        count = max(0, min(groups * 2, total - first));
    }
    __device__ __forceinline__ void request(const mz_packed_mlp &M, const float *__restrict__ packed, int i) {
        const int n = uni(M.n_layers);
        const bool fold = uni((int)tail_foldable(M));
        const mz_packed_layer L = load_layer(M.layer[i]);
        const int kc = L.Kp >> 5, rq = lane >> 4;
        int first, count;
        share(L.Np >> 4, first, count);
        const bool dot = fold && i == n - 2;
        const int w2_off = dot ? uni((int)M.layer[n - 1].w_off) : 0;
        if (count > 0) {
            bias0 = *(const glb_f4 *)(packed + L.b_off + first * 16 + 4 * rq);
            if (dot) w20 = *(const glb_f4 *)(packed + w2_off + (first * 64 + 16 * rq) * 4);
            ring_fetch(s0, packed + L.w_off + (size_t)first * kc * 512, kc, lane);
        }
        if (count > 1) {
            bias1 = *(const glb_f4 *)(packed + L.b_off + (first + 1) * 16 + 4 * rq);
            if (dot) w21 = *(const glb_f4 *)(packed + w2_off + ((first + 1) * 64 + 16 * rq) * 4);
            ring_fetch(s1, packed + L.w_off + (size_t)(first + 1) * kc * 512, kc, lane);
        }
    }
    __device__ __forceinline__ void begin(const mz_packed_mlp &M, const float *__restrict__ packed) {
        wave = uni((int)threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        bias0 = bias1 = w20 = w21 = f32x4{0.f, 0.f, 0.f, 0.f};
        request(M, packed, 0);
    }
    __device__ __forceinline__ lds_f *run(const mz_packed_mlp &M, const float *__restrict__ packed, lds_f *src, int ld, lds_f *buf_a,
                                          lds_f *buf_b, int *out_ld) {
        const int n = uni(M.n_layers);
        const bool fold = uni((int)tail_foldable(M));
        const int nm = fold ? n - 1 : n;  // matrix layers
        for (int i = 0; i < nm; ++i) {
            const mz_packed_layer L = load_layer(M.layer[i]);
            const int kc = L.Kp >> 5, nt_count = L.Np >> 4;
            int first, count;
            share(nt_count, first, count);
            lds_f *dst = (src == buf_a) ? buf_b : buf_a;
            const bool dot = fold && i == nm - 1;
            const bool last = i == n - 1;
            const int kind = dot ? 2 : (last ? 1 : 0);
            const int ldo = lds_stride(L.N);
            const int s_kc = ceil32(L.N) >> 5;
            const bool relu = L.act == MZ_ACT_RELU;
            f32x4 hi[MT], lo[MT];
            if (count > 0) {
                ring_mma<MT>(s0, src + lane * 4, 16 * ld, kc, hi, lo);
                ring_epilogue<MT>(hi, lo, first, nt_count, bias0, w20, relu, kind, dst, ldo, s_kc, lane);
            }
            if (count > 1) {
                ring_mma<MT>(s1, src + lane * 4, 16 * ld, kc, hi, lo);
                ring_epilogue<MT>(hi, lo, first + 1, nt_count, bias1, w21, relu, kind, dst, ldo, s_kc, lane);
            }
            if (i + 1 < nm) request(M, packed, i + 1);  // in flight through the barrier
            __syncthreads();
            if (dot) {
                Team all = whole_workgroup();
                const mz_packed_layer L1 = load_layer(M.layer[n - 1]);
                finish_dot(dst, nt_count, 16 * MT, packed, L1, dst, 1, all);  // in place: row r ends in dst[r]
                __syncthreads();
                src = dst;
                ld = 1;
            } else {
                if (last && L.act == MZ_ACT_SOFTMAX) {
                    Team all = whole_workgroup();
                    softmax_rows(dst, ldo, 16 * MT, L.N, 1, all);
                }
                src = dst;
                ld = ldo;
            }
        }
        *out_ld = ld;
        return src;
    }
};

}  // namespace mzd
