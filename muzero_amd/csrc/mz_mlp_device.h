// Device-side building blocks of the fused MLP kernels (gfx950).
//
// A workgroup of kWaves wavefronts owns a tile of 16*MT rows whose activations live in LDS;
// one dense layer is   Y[rows][N] = act(X[rows][K] * W^T + b).
//
// Arithmetic.  gfx950's fp32 matrix instruction runs at 1/16 of the fp16 rate, so the products
// run on v_mfma_f32_16x16x32_f16 with every fp32 operand v held as TWO fp16 values
//     hi = fp16(v),   lo = fp16((v - hi) * 2^11)            (v = hi + lo * 2^-11 to 2^-22 |v|)
// and a product x*w formed as   hi_x*hi_w  +  2^-11 * (hi_x*lo_w + lo_x*hi_w):  three matrix
// instructions per 16x16x32 block instead of eight fp32 ones, the same 4 bytes per weight, fp32
// accumulation in two accumulators (the cross terms keep their 2^11 scale until the epilogue, so
// neither small activations nor small weights fall into fp16's subnormal range).  The dropped
// lo*lo term is 2^-22 of the product: outputs agree with an fp32 evaluation to a few 1e-7 of
// their scale.  Operands must stay below 65504 in magnitude (fp16's range): beyond it the
// conversion gives infinity and the results are NaN -- loud, not silently wrong.
//
// The product is formed transposed -- the weight tile is the A operand (16 output columns x 32 k),
// the activations the B operand (32 k x 16 rows) -- so a lane's four accumulator values are four
// CONSECUTIVE output columns of one row: the epilogue hands them to the next layer (whose k index
// they are) as one 8-byte store per plane.
//
// Activations feeding a matrix layer are kept in LDS as its operand image ("split image"): per
// 16-row tile and per 32-wide k-chunk c two 1 KiB planes (hi, then all lo planes behind all hi
// planes), lane l = 16*kq + m holding row m, k = 32c + 8kq .. +7 as 16 bytes at l*16 -- a wave
// reads an operand with one conflict-free ds_read_b128.  Activations feeding a narrow layer
// (N <= 8: scalar heads, policy logits; evaluated on the vector ALU in plain fp32) and the values
// other code reads (merge pre-activations, logits) stay fp32 rows [16*MT][ld].  Both forms of a
// buffer of width W fit 16*MT*lds_stride(W) floats.
//
// Weights are re-tiled once per weight update ("packed", mz_mlp.hip).  Matrix layers (fmt 1):
// Kp = ceil32(K), Np = ceil16(N); tile (nt, c) is two 1 KiB planes, lane l = 16*kq + i holding
// W[16nt + i][32c + 8kq .. +7], so a wave fetches an A operand with one coalesced
// global_load_dwordx4 per plane.  Narrow layers (fmt 0) keep fp32 tiles (see PackedNarrowWeights).
// Bias is stored padded to Np behind the tiles.
//
// The row tile is small (16..112 rows) and every weight is used once per workgroup, so this is
// the "GEMV-like" regime: weights go straight from L2 to registers (no LDS staging), a batch of
// k-chunks ahead of the matrix instructions that consume them, and the weight stream of a wave
// runs on across its consecutive tiles.  With the fp16 forms the matrix time is a fraction of
// the time the weight stream needs from L2; the routines are organised around keeping that
// stream busy.
//
// The layer functions are NOT inlined into their callers by default: each gets its own register
// allocation (the fused search kernel calls them ~15 times per simulation round).  LDS operands
// are passed as address-space-3 pointers so that the out-of-line code still uses ds_read/ds_write
// rather than flat accesses.
#pragma once
#include <hip/hip_runtime.h>

#include "muzero_hip.h"

#ifdef MZ_DENSE_TIMELINE
// diagnostic build: shader-clock stamps of the most recent matrix-layer call of each wave
__shared__ unsigned long long mz_tl[8][8];
#define MZ_TL(slot) if ((threadIdx.x & 63) == 0) mz_tl[threadIdx.x >> 6][slot] = clock64();
#else
#define MZ_TL(slot)
#endif

namespace mzd {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
#ifndef MZ_PREFETCH
#define MZ_PREFETCH 2
#endif
constexpr int kPrefetch = MZ_PREFETCH;  // k-chunks of weights per batch held in registers ahead of the MFMAs
#ifndef MZ_PREFETCH_TALL
#define MZ_PREFETCH_TALL 1
#endif
constexpr int kPrefetchTall = MZ_PREFETCH_TALL;  // the same for row tiles of 64 rows and more (one column tile at a time)
constexpr int kTallTiles = 3;  // row tiles of 16 * kTallTiles rows and more take the "tall" form of the matrix routine
#ifndef MZ_WEIGHT_SETS_TALL
#define MZ_WEIGHT_SETS_TALL 4
#endif
constexpr int kWeightSetsTall = MZ_WEIGHT_SETS_TALL;
#ifndef MZ_WEIGHT_SETS
#define MZ_WEIGHT_SETS 2
#endif
constexpr int kWeightSets = MZ_WEIGHT_SETS;  // register sets of such batches a wave rotates through
#ifndef MZ_DENSE_CALL
#define MZ_DENSE_CALL __noinline__
#endif

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(3))) f32x2 lds_f2;
typedef __attribute__((address_space(1))) const float glb_f;    // read-only global memory (weights)
typedef __attribute__((address_space(1))) const f32x4 glb_f4;

__device__ __forceinline__ lds_f *to_lds(void *generic) { return (lds_f *)generic; }
__device__ __forceinline__ lds_i *to_lds_i(void *generic) { return (lds_i *)generic; }

__host__ __device__ inline int ceil16(int x) { return (x + 15) & ~15; }
__host__ __device__ inline int ceil32(int x) { return (x + 31) & ~31; }
// LDS row stride (floats) of a buffer holding rows of `width` valid floats: padded to a multiple of
// 32 (whole k-chunks) plus 4 floats of skew.  16 rows of it also hold the split image of the tile.
__host__ __device__ inline int lds_stride(int width) { return ceil32(width) + 4; }
// LDS offset 0 compares equal to the null pointer of the LDS address space, and optional buffers are
// passed as NULL: the kernels leave the first kLdsReserve floats of their dynamic LDS unused.
constexpr int kLdsReserve = 4;
// Floats of one plane of one k-chunk of a 16-row split image (64 lanes x 16 bytes).
constexpr int kChunkFloats = 256;

constexpr float kLoScale = 2048.0f;          // 2^11: scale of the lo plane
constexpr float kLoUnscale = 1.0f / 2048.0f;

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case MZ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case MZ_ACT_TANH: return tanhf(v);
        case MZ_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

// v -> (hi, lo) as above.  Exact rounding steps (no contraction): every kernel splits alike.
__device__ __forceinline__ void split1(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)__fmul_rn(__fsub_rn(v, (float)hi), kLoScale);
}
// Four consecutive k of one row -> 8 bytes of the hi plane and 8 bytes of the lo plane.
__device__ __forceinline__ void split4(const f32x4 &v, f32x2 &hi, f32x2 &lo) {
    _Float16 a[4], b[4];
    split1(v.x, a[0], b[0]);
    split1(v.y, a[1], b[1]);
    split1(v.z, a[2], b[2]);
    split1(v.w, a[3], b[3]);
    hi = __builtin_bit_cast(f32x2, h4{a[0], a[1], a[2], a[3]});
    lo = __builtin_bit_cast(f32x2, h4{b[0], b[1], b[2], b[3]});
}
// Eight consecutive k of one row (one lane's operand) -> 16 bytes per plane.
__device__ __forceinline__ void split8(const f32x4 &v0, const f32x4 &v1, f32x4 &hi, f32x4 &lo) {
    f32x2 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    hi = f32x4{h0.x, h0.y, h1.x, h1.y};
    lo = f32x4{l0.x, l0.y, l1.x, l1.y};
}

// Wave-uniform values arrive in vector registers when a function is not inlined (arguments
// travel in VGPRs); these helpers move them to scalar registers so that loop bounds, tile
// bases and LDS bases are scalar again and control flow stays branch-based, not exec-masked.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *uni_ptr(T *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ lds_f *uni_lds(const lds_f *p) {
    return (lds_f *)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p);
}
__device__ __forceinline__ lds_i *uni_lds(const lds_i *p) {
    return (lds_i *)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p);
}

// A team is a contiguous range of a workgroup's waves that runs one routine together; two
// teams of one workgroup may run different routines at the same time.  nwaves must be a power
// of two.  The whole workgroup synchronises with s_barrier; a partial team with an arrival
// counter in LDS (one increment per wave and sync, never reset while the team runs) that its
// waves poll, so the teams never wait for each other.  The poll is bounded: a team whose wave
// never arrives sets *err instead of hanging the GPU.
struct Team {
    int wave0, nwaves;
    lds_i *ctr;   // partial teams: arrival counter, err = ctr + 1
    int arrived;  // arrivals expected after this wave's latest sync
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x - 64 * wave0; }
    __device__ __forceinline__ int nthreads() const { return 64 * nwaves; }
    __device__ __forceinline__ void sync() {
        if (ctr == nullptr) {  // the whole workgroup
            __syncthreads();
            return;
        }
        arrived += nwaves;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int polls = 0;
        while (uni(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < arrived) {
            if (++polls > (1 << 19)) {
                ctr[1] = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};
__device__ __forceinline__ Team whole_workgroup() { return Team{0, kWaves, nullptr, 0}; }

// One output of a group of matrix layers that read the same input rows (passed by value).
// The output goes to Ys as fp32 rows (row stride ldy) and / or to Ss as a split image (tile
// stride s_tile floats, s_kc = ceil32(N) / 32 chunks per plane); either may be NULL.
struct DensePart {
    int w_off, b_off;  // float offsets of the packed tiles / padded bias inside the blob
    int N, Np, act;
    lds_f *Ys;
    int ldy;
    lds_f *Ss;
    int s_tile, s_kc;
    const lds_f *row_add;  // optional additive table in LDS, indexed [row_sel[row]][col]
    int row_add_ld;
    const lds_i *row_sel;
    // Optional folded scalar layer: instead of storing Y, the epilogue multiplies act(Y) by the
    // weights of a following one-output layer (fp32 tiles at dot_w_off) and leaves the sums over
    // the four quarters of this tile's 16 columns in dot_out[4 tile + quarter][row] (row stride
    // 16*MT); finish_dot() adds them.  dot_w_off < 0: plain layer.
    int dot_w_off;
    lds_f *dot_out;
    __device__ __forceinline__ DensePart uniform() const {
        return DensePart{uni(w_off), uni(b_off), uni(N), uni(Np), uni(act), uni_lds(Ys), uni(ldy), uni_lds(Ss), uni(s_tile),
                         uni(s_kc), uni_lds(row_add), uni(row_add_ld), uni_lds(row_sel), uni(dot_w_off), uni_lds(dot_out)};
    }
};
// Output as fp32 rows in Ys (split = false) or as a split image in the same memory (split = true).
__device__ __forceinline__ DensePart make_part(const mz_packed_layer &L, lds_f *Ys, int ldy, int act, bool split = false,
                                               const lds_f *row_add = nullptr, int row_add_ld = 0,
                                               const lds_i *row_sel = nullptr) {
    return DensePart{(int)L.w_off, (int)L.b_off, L.N, L.Np, act, split ? nullptr : Ys, ldy, split ? Ys : nullptr, 16 * ldy,
                     ceil32(L.N) >> 5, row_add, row_add_ld, row_sel, -1, nullptr};
}
// Layer L followed by the one-output layer L1, folded (see DensePart::dot_out).
__device__ __forceinline__ DensePart make_dot_part(const mz_packed_layer &L, const mz_packed_layer &L1, lds_f *dot_out) {
    return DensePart{(int)L.w_off, (int)L.b_off, L.N, L.Np, L.act, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, (int)L1.w_off, dot_out};
}

// Epilogue of one 16-column tile.  C layout of the transposed product: row = lane & 15, columns
// 16 nt + 4 (lane >> 4) + r for the lane's four values r.
// w2: for a folded layer (P.dot_w_off >= 0) the one-output layer's weights of this lane's four columns.
template <int MT>
__device__ __forceinline__ void store_tile_core(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, const f32x4 &b, const DensePart &P,
                                                const f32x4 &w2) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, rq = lane >> 4;
    const int col0 = nt * 16 + 4 * rq;
    const bool plain = P.act == MZ_ACT_RELU || P.act == MZ_ACT_NONE;  // act(0) = 0: padded columns need no select
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + j;
        f32x4 v;
        // the cross terms leave their 2^11 scale here (exact), then bias
        v.x = (hi[mt].x + __fmul_rn(lo[mt].x, kLoUnscale)) + b.x;
        v.y = (hi[mt].y + __fmul_rn(lo[mt].y, kLoUnscale)) + b.y;
        v.z = (hi[mt].z + __fmul_rn(lo[mt].z, kLoUnscale)) + b.z;
        v.w = (hi[mt].w + __fmul_rn(lo[mt].w, kLoUnscale)) + b.w;
        if (P.row_add != nullptr) {
            const f32x4 t = *reinterpret_cast<const lds_f4 *>(P.row_add + P.row_sel[row] * P.row_add_ld + col0);
            v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
        }
        if (plain) {
            if (P.act == MZ_ACT_RELU) {  // as activate(): NaN -> 0
                v.x = v.x > 0.0f ? v.x : 0.0f;
                v.y = v.y > 0.0f ? v.y : 0.0f;
                v.z = v.z > 0.0f ? v.z : 0.0f;
                v.w = v.w > 0.0f ? v.w : 0.0f;
            }
            if (P.row_add != nullptr) {  // the table may be non-zero in padded columns
                v.x = col0 + 0 < P.N ? v.x : 0.0f;
                v.y = col0 + 1 < P.N ? v.y : 0.0f;
                v.z = col0 + 2 < P.N ? v.z : 0.0f;
                v.w = col0 + 3 < P.N ? v.w : 0.0f;
            }
        } else {
            v.x = col0 + 0 < P.N ? activate(v.x, P.act) : 0.0f;
            v.y = col0 + 1 < P.N ? activate(v.y, P.act) : 0.0f;
            v.z = col0 + 2 < P.N ? activate(v.z, P.act) : 0.0f;
            v.w = col0 + 3 < P.N ? activate(v.w, P.act) : 0.0f;
        }
        if (P.dot_w_off >= 0) {
            // __fmul_rn: never contracted into the following add, whatever the file's
            // -ffp-contract setting, so every kernel forms the same sums
            float s = __fmul_rn(v.x, w2.x);
            s += __fmul_rn(v.y, w2.y);
            s += __fmul_rn(v.z, w2.z);
            s += __fmul_rn(v.w, w2.w);
            // one partial sum per (tile, quarter of its columns, row): no lane exchange; finish_dot adds them
            P.dot_out[(nt * 4 + rq) * (16 * MT) + row] = s;
            continue;
        }
        if (P.Ys != nullptr) *reinterpret_cast<lds_f4 *>(P.Ys + row * P.ldy + col0) = v;
        if (P.Ss != nullptr) {
            // columns col0 .. col0+3 are k = 32 c + 8 kq + e .. e+3 of the next layer's operand
            f32x2 h, l;
            split4(v, h, l);
            lds_f *img = P.Ss + mt * P.s_tile + (nt >> 1) * kChunkFloats + (2 * (nt & 1) + (rq >> 1)) * 64 + j * 4 + 2 * (rq & 1);
            *reinterpret_cast<lds_f2 *>(img) = h;
            *reinterpret_cast<lds_f2 *>(img + P.s_kc * kChunkFloats) = l;
            if ((nt & 1) == 0 && nt + 1 == (P.Np >> 4)) {  // an odd number of column tiles: the chunk's upper half is zero
                const f32x2 z = {0.f, 0.f};
                *reinterpret_cast<lds_f2 *>(img + 128) = z;
                *reinterpret_cast<lds_f2 *>(img + 128 + P.s_kc * kChunkFloats) = z;
            }
        }
    }
}

template <int MT>
__device__ __forceinline__ void store_tile(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, const f32x4 &b, const DensePart &P,
                                           glb_f *packed_f) {
    f32x4 w2 = {0.f, 0.f, 0.f, 0.f};
    if (P.dot_w_off >= 0)  // weights of columns 16 nt + 4 rq .. +3 in the fp32 tiles of the one-output layer (zero beyond N)
        w2 = *(glb_f4 *)(packed_f + P.dot_w_off + (nt * 64 + 16 * ((threadIdx.x & 63) >> 4)) * 4);
    store_tile_core<MT>(hi, lo, nt, b, P, w2);
}

// parts[part] without indexing the array by a run-time value (which would force the array
// into per-lane stack memory): scalar selects over the compile-time positions instead.
template <int NPARTS>
__device__ __forceinline__ DensePart pick_part(const DensePart (&parts)[NPARTS], int part) {
    DensePart r = parts[0];
#pragma unroll
    for (int i = 1; i < NPARTS; ++i)
        if (part == i) r = parts[i];
    return r;
}

// Several matrix layers over the SAME input rows in one pass (e.g. the policy head's first
// layer and the dynamics merge layer both read the latent): their column tiles form one list
// that is split evenly over the waves, so narrow layers do not leave waves idle and the
// workgroup pays one barrier for all of them.  All parts must share Kp.
//
// xs: split image of the input rows, MT tiles x_tile floats apart, KC chunks per plane.
// A wave walks its tiles NTW at a time.  The weights of those tiles are consumed as a stream
// of batches of KP k-chunks; NS register sets rotate so that batch i+1 is in flight
// (two global_load_dwordx4 of 1 KiB per tile and chunk) while the MFMAs of batch i issue, and the
// stream runs on from one tile group into the next without draining.  Every loop bound below
// is wave-uniform (the wave index goes through readfirstlane), so control flow is scalar.
// Per accumulator the instruction sequence is fixed -- chunks ascending; per chunk hi_w*hi_x,
// then hi_w*lo_x and lo_w*hi_x into the cross-term accumulator -- whatever MT, NTW or the caller:
// every kernel built on this routine produces the same bits.
// The wave's tiles are first_tile .. first_tile + ntiles - 1, walked in ngroups = ceil(ntiles / NTW) groups; a last
// group that is not full repeats the wave's last tile in its unused places (fetched and multiplied, not stored).
template <int MT, int NTW, int NPARTS>
__device__ __forceinline__ void stream_tiles(const lds_f *xs, int x_tile, int KC, const float *__restrict__ packed,
                                             const DensePart (&parts)[NPARTS], int first_tile, int ngroups, int ntiles) {
    constexpr int KP = MT >= kTallTiles ? kPrefetchTall : kPrefetch;
    const int lane = threadIdx.x & 63;
    const int nb = (KC + KP - 1) / KP;  // batches per tile group
    const int total = ngroups * nb;
    auto locate = [&](int tl, int &part, int &nt) {
        part = 0;
        nt = tl;
#pragma unroll
        for (int i = 0; i < NPARTS - 1; ++i) {
            const int n = parts[i].Np >> 4;
            if (part == i && nt >= n) {
                nt -= n;
                part = i + 1;
            }
        }
    };
    // fetch cursor: (group fg, batch fb) of the next weight batch to request
    int fg = 0, fb = 0;
    glb_f4 *fbase[NTW];
    auto fetch_bases = [&]() {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            int part, nt;
            locate(first_tile + min(fg * NTW + t, ntiles - 1), part, nt);
            fbase[t] = (glb_f4 *)(packed + pick_part(parts, part).w_off) + (size_t)nt * KC * 128;
        }
    };
    fetch_bases();
    auto fetch = [&](f32x4 (&w)[NTW][KP][2]) {
        const int kc0 = fb * KP;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            // one scalar base per tile and batch, the planes at constant offsets (1 KiB apart): a
            // group's last batch may read up to KP - 1 chunks past its tile, i.e. into the next
            // tile or the slack every packed MLP ends with (mz_mlp_plan); those values feed no MFMA
            glb_f4 *base = fbase[t] + (size_t)kc0 * 128;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                w[t][k][0] = base[(2 * k) * 64 + lane];
                w[t][k][1] = base[(2 * k + 1) * 64 + lane];
            }
        }
        // The cursor stops at the last batch: fetches past the end of the stream (issued unconditionally, so
        // that every wait below counts an exact number of loads in flight) request that batch once more.
        if (fb + 1 < nb) {
            ++fb;
        } else if (fg + 1 < ngroups) {
            fb = 0;
            ++fg;
            fetch_bases();
        }
    };
    // compute cursor
    int cg = 0, cb = 0;
    f32x4 acc_hi[NTW][MT], acc_lo[NTW][MT];
    f32x4 bias[NTW];
    int c_part[NTW], c_nt[NTW];
    f32x4 xh[MT], xl[MT];
    const lds_f *xlane = xs + lane * 4;
    const int lo_off = KC * kChunkFloats;
    auto read_x = [&](int kc, f32x4 (&h)[MT], f32x4 (&l)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const lds_f *p = xlane + mt * x_tile + kc * kChunkFloats;
            h[mt] = *reinterpret_cast<const lds_f4 *>(p);
            l[mt] = *reinterpret_cast<const lds_f4 *>(p + lo_off);
        }
    };
    auto compute = [&](const f32x4 (&w)[NTW][KP][2]) {
        const int kc0 = cb * KP;
        if (cb == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                locate(first_tile + min(cg * NTW + t, ntiles - 1), c_part[t], c_nt[t]);
                // requested now, needed only by the epilogue after the last batch
                bias[t] = *(glb_f4 *)(packed + pick_part(parts, c_part[t]).b_off + c_nt[t] * 16 + 4 * (lane >> 4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc_hi[t][mt] = acc_lo[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // The B operands run one chunk ahead of the MFMAs that consume them (xh/xl: current,
        // nh/nl: next; the sequence wraps to chunk 0 at the end of a tile group), so their LDS
        // latency hides behind the MFMAs.
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            // chunks past the end of K (last batch of a group) skip their MFMAs only; the
            // operand pipeline advances unconditionally so that it is pure register renaming
            const bool valid = kc0 + k < KC;
            const int kn = (kc0 + k + 1 < KC) ? kc0 + k + 1 : 0;
            f32x4 nh[MT], nl[MT];
            if constexpr (MT >= kTallTiles) {  // tall row tiles: no room for two sets of operands; the other wave of the SIMD covers the read
                if (valid) read_x(kc0 + k, xh, xl);
            } else {
                read_x(kn, nh, nl);
            }
            // every operand read of the step is issued before its matrix instructions (left to itself the
            // scheduler re-uses one register set and waits for each read in turn)
            if constexpr (MT <= 4) __builtin_amdgcn_sched_barrier(0);
            if (valid) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        const h8 wh = __builtin_bit_cast(h8, w[t][k][0]), wl = __builtin_bit_cast(h8, w[t][k][1]);
                        const h8 ah = __builtin_bit_cast(h8, xh[mt]), al = __builtin_bit_cast(h8, xl[mt]);
                        acc_hi[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, acc_hi[t][mt], 0, 0, 0);
                        acc_lo[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, acc_lo[t][mt], 0, 0, 0);
                        acc_lo[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, acc_lo[t][mt], 0, 0, 0);
                    }
                }
            }
            if constexpr (MT < 4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xh[mt] = nh[mt], xl[mt] = nl[mt];
            }
        }
        if (++cb == nb) {
            cb = 0;
            ++cg;
            if (cg == ngroups) { MZ_TL(4) }
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                if ((cg - 1) * NTW + t < ntiles)
                    store_tile<MT>(acc_hi[t], acc_lo[t], c_nt[t], bias[t], pick_part(parts, c_part[t]), (glb_f *)packed);
        }
    };
    // NS register sets rotate: NS - 1 batches are in flight while one is consumed.
    constexpr int NS = MT >= kTallTiles ? kWeightSetsTall : kWeightSets;
    f32x4 w[NS][NTW][KP][2];
    MZ_TL(1)
    // the weights have the longest way: requested before the operand pipeline is primed
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) fetch(w[u]);
    if constexpr (MT < kTallTiles) read_x(0, xh, xl);
    MZ_TL(2)
    for (int i = 0; i < total; i += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            fetch(w[(u + NS - 1) % NS]);
            if (i + u < total) compute(w[u]);
            if (i == 0 && u == 0) { MZ_TL(3) }
        }
    }
    MZ_TL(5)
}

// Entry points (not inlined: own register allocation).  Kp is shared by all parts.  No
// trailing barrier.
template <int MT, int NPARTS>
__device__ __forceinline__ void dense_group_body(const lds_f *xs, int x_tile, const float *packed, int Kp,
                                                 const DensePart (&parts)[NPARTS], int wave0, int nwaves) {
    const int wave = uni(threadIdx.x >> 6) - wave0;
    const int KC = Kp >> 5;
    int total = 0;
#pragma unroll
    for (int i = 0; i < NPARTS; ++i) total += parts[i].Np >> 4;
    // Tiles are dealt as evenly as possible (waves w and w+4 usually share a SIMD, so the larger shares go to
    // the first waves and every SIMD ends up with nearly the same count) -- in units of what a wave walks at a
    // time, so that every wave makes ONE pass over its share: a second pass would pay the start of a weight
    // stream (a round trip to L2) again.
    const int lg = 31 - __builtin_clz(nwaves);
    constexpr int NTW = MT >= kTallTiles ? 1 : 2;  // tall row tiles: the accumulators and operands of one column tile fill the registers
    const int units = (total + NTW - 1) / NTW;
    const int share = units >> lg, extra = units - (share << lg);
    const int first = (wave * share + min(wave, extra)) * NTW;
    const int groups = share + (wave < extra ? 1 : 0);
    if (groups <= 0) return;
    stream_tiles<MT, NTW, NPARTS>(xs, x_tile, KC, packed, parts, first, groups, min(groups * NTW, total - first));
}
// Out-of-line entry points.  Every argument is a scalar: aggregates would be passed through
// per-lane stack memory (written 64 times per wave and eventually to HBM), scalars travel in
// registers.
template <int MT>
__device__ MZ_DENSE_CALL void dense_group1_impl(const lds_f *xs, int x_tile, const float *packed, int Kp, int w_off, int b_off,
                                                int N, int Np, int act, lds_f *Ys, int ldy, lds_f *Ss, int s_tile, int s_kc,
                                                const lds_f *row_add, int row_add_ld, const lds_i *row_sel, int dot_w,
                                                lds_f *dot_out, int wave0, int nwaves) {
    MZ_TL(0)
    const DensePart parts[1] = {
        DensePart{w_off, b_off, N, Np, act, Ys, ldy, Ss, s_tile, s_kc, row_add, row_add_ld, row_sel, dot_w, dot_out}.uniform()};
    dense_group_body<MT, 1>(uni_lds(xs), uni(x_tile), uni_ptr(packed), uni(Kp), parts, uni(wave0), uni(nwaves));
}
template <int MT>
__device__ MZ_DENSE_CALL void dense_group2_impl(const lds_f *xs, int x_tile, const float *packed, int Kp, int w0, int b0, int N0,
                                                int Np0, int act0, lds_f *Y0, int ldy0, lds_f *S0, int st0, int sk0, int dw0,
                                                lds_f *do0, int w1, int b1, int N1, int Np1, int act1, lds_f *Y1, int ldy1,
                                                lds_f *S1, int st1, int sk1, int dw1, lds_f *do1, int wave0, int nwaves) {
    MZ_TL(0)
    const DensePart parts[2] = {DensePart{w0, b0, N0, Np0, act0, Y0, ldy0, S0, st0, sk0, nullptr, 0, nullptr, dw0, do0}.uniform(),
                                DensePart{w1, b1, N1, Np1, act1, Y1, ldy1, S1, st1, sk1, nullptr, 0, nullptr, dw1, do1}.uniform()};
    dense_group_body<MT, 2>(uni_lds(xs), uni(x_tile), uni_ptr(packed), uni(Kp), parts, uni(wave0), uni(nwaves));
}
// xs / ldx: the split image of the input rows in a buffer of row stride ldx (tiles 16 * ldx floats apart).
template <int MT>
__device__ __forceinline__ void dense_group1(const lds_f *xs, int ldx, const float *packed, int Kp, const DensePart &p,
                                             const Team &tm = whole_workgroup()) {
    dense_group1_impl<MT>(xs, 16 * ldx, packed, Kp, p.w_off, p.b_off, p.N, p.Np, p.act, p.Ys, p.ldy, p.Ss, p.s_tile, p.s_kc,
                          p.row_add, p.row_add_ld, p.row_sel, p.dot_w_off, p.dot_out, tm.wave0, tm.nwaves);
}
template <int MT>
__device__ __forceinline__ void dense_group2(const lds_f *xs, int ldx, const float *packed, int Kp, const DensePart &p,
                                             const DensePart &q, const Team &tm = whole_workgroup()) {
    dense_group2_impl<MT>(xs, 16 * ldx, packed, Kp, p.w_off, p.b_off, p.N, p.Np, p.act, p.Ys, p.ldy, p.Ss, p.s_tile, p.s_kc,
                          p.dot_w_off, p.dot_out, q.w_off, q.b_off, q.N, q.Np, q.act, q.Ys, q.ldy, q.Ss, q.s_tile, q.s_kc,
                          q.dot_w_off, q.dot_out, tm.wave0, tm.nwaves);
}

// Floats of the partial-sum buffer of a folded layer with `tiles` column tiles over `rows` rows.
__host__ __device__ inline int dot_floats(int tiles, int rows) { return 4 * tiles * rows; }
// Second half of a folded scalar layer: out[row * out_ld] = act(bias + sum over the tiles' quarter sums
// dot[4 tile + quarter][row] in ascending order).  `rows` = 16*MT of the producing call.  No sync.
__device__ __forceinline__ void finish_dot(const lds_f *dot, int tiles, int rows, const float *__restrict__ packed,
                                           const mz_packed_layer &L1, lds_f *out, int out_ld, const Team &tm) {
    const float b = packed[L1.b_off];
    for (int r = tm.tid(); r < rows; r += tm.nthreads()) {
        float s = 0.0f;
        for (int t = 0; t < 4 * tiles; ++t) s += dot[t * rows + r];
        out[r * out_ld] = activate(s + b, L1.act);
    }
}

// fp32 rows [16*tiles][ld] -> their split image in `dst` (tile stride 16*ld_dst floats); columns >= valid
// (a multiple of 4) are taken as zero.  One operand (8 consecutive k of one row) per lane and pass.  No sync.
__device__ __forceinline__ void rows_to_split(const lds_f *src, int ld, lds_f *dst, int ld_dst, int tiles, int KC, int valid,
                                              const Team &tm) {
    const int n = tiles * KC * 64;
    for (int i = tm.tid(); i < n; i += tm.nthreads()) {
        const int l = i & 63, c = (i >> 6) % KC, mt = (i >> 6) / KC;
        const int k0 = 32 * c + 8 * (l >> 4);
        const lds_f *p = src + (mt * 16 + (l & 15)) * ld + k0;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 hi, lo;
        split8(k0 < valid ? *reinterpret_cast<const lds_f4 *>(p) : zero, k0 + 4 < valid ? *reinterpret_cast<const lds_f4 *>(p + 4) : zero,
               hi, lo);
        lds_f *q = dst + mt * 16 * ld_dst + c * kChunkFloats + l * 4;
        *reinterpret_cast<lds_f4 *>(q) = hi;
        *reinterpret_cast<lds_f4 *>(q + KC * kChunkFloats) = lo;
    }
}

// Weight sources of dense_narrow: chunk q (4 consecutive k) of output column `col`, from the fp32
// tiles of a narrow layer: tile kt is 64 float4, lane l, element j = W[l & 15][16 kt + 4 (l >> 4) + j].
struct PackedNarrowWeights {  // straight from the packed blob in global memory
    const float *wp, *bias;
    static __device__ __forceinline__ PackedNarrowWeights from(const float *w, const float *b, const lds_f *, int, int) {
        return PackedNarrowWeights{uni_ptr(w), uni_ptr(b)};
    }
    __device__ __forceinline__ f32x4 operator()(int q, int col) const { return ((glb_f4 *)wp)[(size_t)(q >> 2) * 64 + (q & 3) * 16 + col]; }
    __device__ __forceinline__ float b(int col) const { return ((glb_f *)bias)[col]; }
};
struct LdsNarrowWeights {  // compact copy [Kp/4][N] of float4 + 8 bias floats staged in LDS
    const lds_f *wl;
    int N, quads;
    static __device__ __forceinline__ LdsNarrowWeights from(const float *, const float *, const lds_f *w, int n, int q) {
        return LdsNarrowWeights{uni_lds(w), n, q};
    }
    __device__ __forceinline__ f32x4 operator()(int q, int col) const {
        return *reinterpret_cast<const lds_f4 *>(wl + (q * N + col) * 4);
    }
    __device__ __forceinline__ float b(int col) const { return wl[quads * N * 4 + col]; }
};

// Narrow layers (N <= 8: value/reward scalars, policy logits): a 16-wide MFMA tile would idle
// all but one wave, so these run on the vector ALU in plain fp32.  Four adjacent lanes share one
// output (row, col), each summing a quarter of K as float4 dot products, then a lane
// exchange adds the quarters.  The summation order is fixed by this function alone, so the
// result does not depend on where the weights are fetched from.  No trailing barrier.
// Scalar arguments only (see dense_group1_impl); gw/gb or lw is unused depending on Weights.
template <typename Weights>
__device__ __noinline__ void dense_narrow(const lds_f *Xs_, int ldx_, int rows_, const float *gw, const float *gb,
                                          const lds_f *lw, int Kp_, int N_, int act_, lds_f *Ys_, int ldy_, int wave0_,
                                          int nwaves_) {
    const int tid = (int)threadIdx.x - 64 * uni(wave0_), nthreads = 64 * uni(nwaves_);
    const lds_f *Xs = uni_lds(Xs_);
    lds_f *Ys = uni_lds(Ys_);
    const int ldx = uni(ldx_), rows = uni(rows_), Kp = uni(Kp_), N = uni(N_), act = uni(act_), ldy = uni(ldy_);
    const Weights wf = Weights::from(gw, gb, lw, N, Kp >> 2);
    const int quads = Kp >> 2;  // float4 chunks along K
    const int outputs = rows * N;
    for (int o0 = 0; o0 < outputs; o0 += nthreads / 4) {
        const int o = o0 + (tid >> 2);
        const int part = tid & 3;
        float s = 0.0f;
        int row = 0, col = 0;
        if (o < outputs) {
            row = o / N;
            col = o - row * N;
            const lds_f *x = Xs + row * ldx;
#pragma unroll 4
            for (int q = part; q < quads; q += 4) {
                const f32x4 w = wf(q, col);
                const f32x4 v = *reinterpret_cast<const lds_f4 *>(x + 4 * q);
                s = fmaf(v.x, w.x, s);
                s = fmaf(v.y, w.y, s);
                s = fmaf(v.z, w.z, s);
                s = fmaf(v.w, w.w, s);
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (o < outputs && part == 0) Ys[row * ldy + col] = activate(s + wf.b(col), act);
    }
    // zero the padding columns so a following layer may read whole 16-float chunks
    const int pad = ceil16(N) - N;
    for (int i = tid; i < rows * pad; i += nthreads) {
        const int r = i / pad, c = N + i % pad;
        Ys[r * ldy + c] = 0.0f;
    }
}

// Copies a narrow layer's weights and bias into LDS in the layout LdsNarrowWeights reads:
// [Kp/4][N] float4 followed by 8 bias floats.
__host__ __device__ inline int narrow_lds_floats(const mz_packed_layer &L) { return (L.Kp >> 2) * L.N * 4 + 8; }
__device__ __forceinline__ void stage_narrow(const float *__restrict__ packed, const mz_packed_layer &L, lds_f *dst) {
    const int quads = L.Kp >> 2;
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(packed + L.w_off);
    for (int i = threadIdx.x; i < quads * L.N; i += blockDim.x) {
        const int q = i / L.N, col = i - q * L.N;
        *reinterpret_cast<lds_f4 *>(dst + i * 4) = wp[(size_t)(q >> 2) * 64 + (q & 3) * 16 + col];
    }
    for (int i = threadIdx.x; i < 8; i += blockDim.x) dst[quads * L.N * 4 + i] = (i < L.N) ? packed[L.b_off + i] : 0.0f;
}

// A layer runs on the matrix cores (reads a split image) or on the vector ALU (reads fp32 rows):
// decided at pack time (mz_packed_layer::fmt).
template <typename LayerRef>
__host__ __device__ inline bool is_matrix_layer(const LayerRef &L) { return L.fmt != 0; }

// One dense layer over the workgroup's row tile.  Xs: LDS buffer of row stride ldx holding the
// input in the form the layer reads (split image for a matrix layer, fp32 rows with columns
// [K, Kp) zero for a narrow one).  Ys: LDS buffer of row stride ldy that receives the output as a
// split image (`split_out`, matrix layers only) or as fp32 rows with columns [N, Np) zero.
// `row_add` (optional, LDS): per-row additive table row_add[row_sel[r]][n] added before the
// activation (the action-embedding contribution of the dynamics merge layer).
// `narrow_lds` (optional): LDS copy of a narrow layer's weights made by stage_narrow.
// Ends with a team sync: Ys is complete and Xs may be overwritten afterwards.
template <int MT>
__device__ __forceinline__ void dense_tile(const lds_f *Xs, int ldx, const float *__restrict__ packed,
                                           const mz_packed_layer &L, lds_f *Ys, int ldy, bool split_out, const lds_f *row_add,
                                           int row_add_ld, const lds_i *row_sel, const lds_f *narrow_lds, Team &tm) {
    if (!is_matrix_layer(L)) {
        if (narrow_lds)
            dense_narrow<LdsNarrowWeights>(Xs, ldx, 16 * MT, nullptr, nullptr, narrow_lds, L.Kp, L.N, L.act, Ys, ldy,
                                           tm.wave0, tm.nwaves);
        else
            dense_narrow<PackedNarrowWeights>(Xs, ldx, 16 * MT, packed + L.w_off, packed + L.b_off, nullptr, L.Kp, L.N,
                                              L.act, Ys, ldy, tm.wave0, tm.nwaves);
        tm.sync();
        return;
    }
    dense_group1<MT>(Xs, ldx, packed, L.Kp, make_part(L, Ys, ldy, L.act, split_out, row_add, row_add_ld, row_sel), tm);
    tm.sync();
}

// In-place softmax over the first N columns of each of `rows` LDS rows, applied `times`
// times (torch.softmax semantics: subtract the row maximum, exp, normalise; the sum runs over
// the columns in ascending order).  N <= 8 (action logits): eight lanes per row, one column
// each, the row-wide maximum and sum formed by lane exchanges in that same order.  Wider rows:
// one lane per row.  Ends with a team sync.
__device__ __forceinline__ void softmax_rows(lds_f *Ys, int ldy, int rows, int N, int times, Team &tm) {
    const int tid = tm.tid(), nthreads = tm.nthreads();
    if (N <= 8) {
        for (int r0 = 0; r0 < rows; r0 += nthreads >> 3) {
            const int row = r0 + (tid >> 3), j = tid & 7;
            const bool live = row < rows && j < N;
            float v = live ? Ys[row * ldy + j] : 0.0f;
            for (int t = 0; t < times; ++t) {
                float mx = __shfl(v, 0, 8);
                for (int i = 1; i < N; ++i) mx = fmaxf(mx, __shfl(v, i, 8));
                const float e = expf(v - mx);
                float s = 0.0f;
                for (int i = 0; i < N; ++i) s += __shfl(e, i, 8);
                v = e / s;
            }
            if (live) Ys[row * ldy + j] = v;
        }
    } else {
        for (int row = tid; row < rows; row += nthreads) {
            lds_f *y = Ys + row * ldy;
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
    }
    tm.sync();
}
__device__ __forceinline__ void softmax_rows(lds_f *Ys, int ldy, int rows, int N, int times) {
    Team tm = whole_workgroup();
    softmax_rows(Ys, ldy, rows, N, times, tm);
}

// Layer descriptors may live in kernel arguments or in LDS (a kernel that holds many of them
// copies them to LDS to keep them out of scalar registers between uses): these helpers accept
// either, and return wave-uniform values.
template <typename LayerRef>
__device__ __forceinline__ mz_packed_layer load_layer(const LayerRef &s) {
    mz_packed_layer L;
    L.K = uni(s.K);
    L.N = uni(s.N);
    L.Kp = uni(s.Kp);
    L.Np = uni(s.Np);
    L.act = uni(s.act);
    L.fmt = uni(s.fmt);
    L.w_off = uni((int)s.w_off);
    L.b_off = uni((int)s.b_off);
    return L;
}

// A head that ends in [matrix layer, one-output layer] has that pair folded: the matrix layer's
// epilogue forms the dot products with the output layer's weights (DensePart::dot_out).
template <typename MLP>
__host__ __device__ inline bool tail_foldable(const MLP &M) {
    const int n = M.n_layers;
    return n >= 2 && M.layer[n - 1].N == 1 && M.layer[n - 2].N > 8 && M.layer[n - 1].K == M.layer[n - 2].N;
}

// Runs layers [first, n) of `M` over the team's row tile, ping-ponging between two LDS buffers.
// The input of layer `first` is expected in `src` (row stride ld) in the form that layer reads --
// see dense_tile; a caller whose input exists in both forms passes the other one as `src_alt`
// (`src` = split image, `src_alt` = fp32 rows) and the layer picks.  Returns the buffer that holds
// the output as fp32 rows, row stride *out_ld.  A MZ_ACT_SOFTMAX layer is a linear layer followed
// by softmax_rows.  `narrow` (optional): byte offsets into `smem` of LDS copies of narrow layers'
// weights, or -1.  Every layer ends with a team sync.
template <int MT, typename MLP, typename NarrowPtr = const int *>
__device__ __forceinline__ lds_f *mlp_tile(Team &tm, const MLP &M, int first, const float *__restrict__ packed,
                                           lds_f *src, lds_f *src_alt, int ld, lds_f *buf_a, lds_f *buf_b, int *out_ld,
                                           unsigned char *smem = nullptr, NarrowPtr narrow = nullptr) {
    const int n = uni(M.n_layers);
    const bool fold = uni((int)tail_foldable(M)) && first <= n - 2;
    const int plain_end = fold ? n - 2 : n;
    if (src_alt != nullptr && first < n && !uni((int)is_matrix_layer(M.layer[first]))) src = src_alt;
    for (int i = first; i < plain_end; ++i) {
        const mz_packed_layer L = load_layer(M.layer[i]);
        lds_f *dst = (src == buf_a) ? buf_b : buf_a;
        const int ldo = lds_stride(L.N);
        const int nwo = narrow ? uni(narrow[i]) : -1;
        const lds_f *nw = nwo >= 0 ? to_lds(smem + nwo) : nullptr;
        // the output takes the form its reader wants: the next layer if it is a matrix layer, fp32 rows otherwise
        const bool split_out = i + 1 < n && uni((int)is_matrix_layer(M.layer[i + 1]));
        dense_tile<MT>(src, ld, packed, L, dst, ldo, split_out, nullptr, 0, nullptr, nw, tm);
        if (L.act == MZ_ACT_SOFTMAX) softmax_rows(dst, ldo, 16 * MT, L.N, 1, tm);
        if (!is_matrix_layer(L) && i + 1 < n && uni((int)is_matrix_layer(M.layer[i + 1]))) {
            // fp32 rows of a narrow layer feed a matrix layer: their image goes to the other buffer
            lds_f *img = (dst == buf_a) ? buf_b : buf_a;
            rows_to_split(dst, ldo, img, ldo, MT, ceil32(L.N) >> 5, ceil16(L.N), tm);
            tm.sync();
            dst = img;
        }
        src = dst;
        ld = ldo;
    }
    if (fold) {
        const mz_packed_layer L = load_layer(M.layer[n - 2]), L1 = load_layer(M.layer[n - 1]);
        lds_f *dst = (src == buf_a) ? buf_b : buf_a;
        dense_group1<MT>(src, ld, packed, L.Kp, make_dot_part(L, L1, dst), tm);
        tm.sync();
        finish_dot(dst, L.Np >> 4, 16 * MT, packed, L1, dst, 1, tm);  // in place: row r ends in dst[r]
        tm.sync();
        src = dst;
        ld = 1;
    }
    *out_ld = ld;
    return src;
}

// Cooperative load of up to `rows` global rows (row r from row_ptr(r), NULL: zeros) into an LDS tile
// [TM][ld] of fp32 rows, zero-filling columns [K, ld) and rows >= rows.  Two round trips to memory, not
// one per element: a wave takes whole rows, asks for the row's address once (wave-uniform, so it lives in
// scalar registers) and requests all of the row's values before the first is stored.  Ends with a barrier.
template <typename RowPtr>
__device__ __forceinline__ void load_rows(lds_f *Xs, int ld, int TM, int rows, int K, RowPtr row_ptr) {
    const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
    constexpr int kMaxPieces = 4;  // 64 * 4 values of a row in flight per pass
    for (int r0 = wave; r0 < TM; r0 += 2 * kWaves) {
        // two rows per pass: their address loads and their value loads overlap
        const float *p[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + h * kWaves;
            p[h] = (r < rows) ? row_ptr(r) : nullptr;
        }
        for (int c0 = 0; c0 < ld; c0 += 64 * kMaxPieces) {
            float v[2][kMaxPieces];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < kMaxPieces; ++j) {
                    const int c = c0 + 64 * j + lane;
                    v[h][j] = (p[h] != nullptr && c < K) ? p[h][c] : 0.0f;
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < kMaxPieces; ++j) {
                    const int r = r0 + h * kWaves, c = c0 + 64 * j + lane;
                    if (r < TM && c < ld) Xs[r * ld + c] = v[h][j];
                }
        }
    }
    __syncthreads();
}
// load_rows followed by the rows' split image in Ss (same row stride); ends with a barrier.
template <typename RowPtr>
__device__ __forceinline__ void load_rows_split(lds_f *Xs, lds_f *Ss, int ld, int MT, int rows, int K, RowPtr row_ptr) {
    load_rows(Xs, ld, 16 * MT, rows, K, row_ptr);
    rows_to_split(Xs, ld, Ss, ld, MT, ceil32(K) >> 5, ld, whole_workgroup());
    __syncthreads();
}

}  // namespace mzd
