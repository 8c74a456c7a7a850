// Device-side building blocks of the fused fp32 MLP kernels (gfx950).
//
// A workgroup of kWaves wavefronts owns a tile of 16*MT rows whose activations live in LDS;
// one dense layer is   Y[rows][N] = act(X[rows][K] * W^T + b)   on v_mfma_f32_16x16x4_f32
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
//
// The row tile is small (16..112 rows) and every weight is used once per workgroup, so this
// is the "GEMV-like" regime: weights go straight from L2 to registers (no LDS staging),
// kPrefetch k-steps ahead, a wave keeps two column tiles in flight so that the MFMA pipe
// always has an independent accumulator to issue into, and the weight stream of a wave runs
// on across its consecutive tiles.  On gfx950 an f32 MFMA and vector-ALU instructions of the
// same SIMD do not execute concurrently (measured: a wave of independent v_fma makes no
// progress while another wave issues v_mfma_f32_16x16x4_f32 back to back), so every vector
// instruction in these routines is paid for in MFMA time: operands are formed with as few of
// them as possible, addresses stay scalar, and the epilogues are branch-free.
//
// The layer functions are deliberately NOT inlined into their callers: each gets its own
// register allocation (the fused search kernel calls them ~15 times per simulation round and
// would otherwise spill).  LDS operands are passed as address-space-3 pointers so that the
// out-of-line code still uses ds_read/ds_write rather than flat accesses.
#pragma once
#include <hip/hip_runtime.h>

#include "muzero_hip.h"

#ifdef MZ_EXP_TIMELINE
// diagnostic: per-wave time stamps of the most recent dense call, slot-indexed
__shared__ unsigned long long mz_tl[8][16];
#define MZ_TL(slot) if ((threadIdx.x & 63) == 0) mz_tl[threadIdx.x >> 6][slot] = clock64();
#else
#define MZ_TL(slot)
#endif

namespace mzd {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
#ifndef MZ_PREFETCH
#define MZ_PREFETCH 4
#endif
constexpr int kPrefetch = MZ_PREFETCH;  // k-steps of weights held in registers ahead of the MFMAs
#ifndef MZ_WEIGHT_SETS
#define MZ_WEIGHT_SETS 2
#endif
constexpr int kWeightSets = MZ_WEIGHT_SETS;  // register sets of such batches a wave rotates through (short row tiles)

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(1))) const float glb_f;    // read-only global memory (weights)
typedef __attribute__((address_space(1))) const f32x4 glb_f4;

__device__ __forceinline__ lds_f *to_lds(void *generic) { return (lds_f *)generic; }
__device__ __forceinline__ lds_i *to_lds_i(void *generic) { return (lds_i *)generic; }

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

// One output of a group of layers that read the same input rows (passed by value).
struct DensePart {
    int w_off, b_off;  // float offsets of the packed tiles / padded bias inside the blob
    int N, Np, act;
    lds_f *Ys;
    int ldy;
    const lds_f *row_add;  // optional additive table in LDS, indexed [row_sel[row]][col]
    int row_add_ld;
    const lds_i *row_sel;
    // Optional folded scalar layer: instead of storing Y, the epilogue multiplies act(Y) by the
    // weights of a following one-output layer (packed tiles at dot_w_off) and leaves the sum
    // over this tile's 16 columns in dot_out[tile][row] (row stride 16*MT); finish_dot() adds
    // the tiles.  dot_w_off < 0: plain layer.
    int dot_w_off;
    lds_f *dot_out;
    __device__ __forceinline__ DensePart uniform() const {
        return DensePart{uni(w_off), uni(b_off), uni(N), uni(Np), uni(act), uni_lds(Ys), uni(ldy),
                         uni_lds(row_add), uni(row_add_ld), uni_lds(row_sel), uni(dot_w_off), uni_lds(dot_out)};
    }
};
__device__ __forceinline__ DensePart make_part(const mz_packed_layer &L, lds_f *Ys, int ldy, int act,
                                               const lds_f *row_add = nullptr, int row_add_ld = 0,
                                               const lds_i *row_sel = nullptr) {
    return DensePart{(int)L.w_off, (int)L.b_off, L.N, L.Np, act, Ys, ldy, row_add, row_add_ld, row_sel, -1, nullptr};
}
// Layer L followed by the one-output layer L1, folded (see DensePart::dot_out).
__device__ __forceinline__ DensePart make_dot_part(const mz_packed_layer &L, const mz_packed_layer &L1, lds_f *dot_out) {
    return DensePart{(int)L.w_off, (int)L.b_off, L.N, L.Np, L.act, nullptr, 0, nullptr, 0, nullptr, (int)L1.w_off, dot_out};
}
// Sum over the 16 lanes of a DPP row (every lane receives it; the association is fixed).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror
    v += dpp_f32<0x140>(v);  // row_mirror
    return v;
}

// Epilogue of one 16-column tile: bias, optional per-row additive table, activation, store.
// C layout of the MFMA: column = lane & 15, row = 4*(lane >> 4) + reg.
template <int MT>
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[MT], int nt, float b, const DensePart &P, glb_f *packed_f) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int col = nt * 16 + m;
    const bool real_col = col < P.N;
    if (P.dot_w_off >= 0) {
        // weight of column `col` in the packed one-output layer: tile kt = nt, row 0 of the
        // tile lives in lanes 16*q, element j, with col = 16*nt + 4*q + j
        const float w2 = real_col ? packed_f[P.dot_w_off + (nt * 64 + 16 * (m >> 2)) * 4 + (m & 3)] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = real_col ? activate(acc[mt][r] + b, P.act) : 0.0f;
                // __fmul_rn: never contracted into the following add, whatever the file's
                // -ffp-contract setting, so every kernel forms the same sums
                const float sum = row16_sum(__fmul_rn(h, w2));
                if (m == 0) P.dot_out[nt * (16 * MT) + mt * 16 + kq * 4 + r] = sum;
            }
        }
        return;
    }
    // the common cases (plain relu / identity layers) take a branch-free path; the branches
    // here are wave-uniform
    if (P.row_add == nullptr && (P.act == MZ_ACT_RELU || P.act == MZ_ACT_NONE)) {
        // padded columns need no select here: their packed weights and bias are zero, so the
        // accumulator holds 0 and relu / identity keep it.  Address arithmetic is scalar: one
        // base per lane, the row step added per store.
        lds_f *y = P.Ys + (kq * 4) * P.ldy + col;
        const int step = P.ldy;
        if (P.act == MZ_ACT_RELU) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][r] + b;
                    y[(mt * 16 + r) * step] = v > 0.0f ? v : 0.0f;  // activate(): NaN -> 0
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[(mt * 16 + r) * step] = acc[mt][r] + b;
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + kq * 4 + r;
            float v = acc[mt][r] + b;
            if (P.row_add != nullptr && real_col) v += P.row_add[P.row_sel[row] * P.row_add_ld + col];
            P.Ys[row * P.ldy + col] = real_col ? activate(v, P.act) : 0.0f;
        }
    }
}

// A-operand sources.  X[row 16*mt + m][koff .. koff+3] = form(shared(m, koff), raw(mt, m, koff)):
// `shared` and `raw` are the LDS reads (shared: common to all row tiles of one k-step), `form`
// the vector work on them, so a caller can issue every read of a k-step first and form the
// operands later, between MFMAs.
struct LdsRows {  // activations resident in LDS
    const lds_f *x;
    int ld;
    static __device__ __forceinline__ LdsRows from(const lds_f *a, int b, const lds_f *, int, const lds_i *) { return LdsRows{a, b}; }
    __device__ __forceinline__ const lds_i *s4() const { return nullptr; }
    __device__ __forceinline__ const lds_f *s0() const { return x; }
    __device__ __forceinline__ int s1() const { return ld; }
    __device__ __forceinline__ const lds_f *s2() const { return nullptr; }
    __device__ __forceinline__ int s3() const { return 0; }
    __device__ __forceinline__ f32x4 shared(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 raw(int mt, int m, int koff) const {
        return *reinterpret_cast<const lds_f4 *>(x + (mt * 16 + m) * ld + koff);
    }
    static __device__ __forceinline__ f32x4 form(const f32x4 &, const f32x4 &r) { return r; }
};
__device__ __forceinline__ f32x4 relu_sum(const f32x4 &a, const f32x4 &b) {
    f32x4 r;
    r.x = fmaxf(a.x + b.x, 0.f);
    r.y = fmaxf(a.y + b.y, 0.f);
    r.z = fmaxf(a.z + b.z, 0.f);
    r.w = fmaxf(a.w + b.w, 0.f);
    return r;
}
// Rows of the expansion: row (mt = action a, m = tree t) = relu(z[t][:] + table[a][:]), formed
// on the fly from the merge pre-activation z (LDS, 16 rows) and the per-action table (LDS), so
// the A*16 child latents never need an LDS image of their own.
struct ChildRows {
    const lds_f *z;
    int ldz;
    const lds_f *table;
    int ldt;
    static __device__ __forceinline__ ChildRows from(const lds_f *a, int b, const lds_f *c, int d, const lds_i *) { return ChildRows{a, b, c, d}; }
    __device__ __forceinline__ const lds_i *s4() const { return nullptr; }
    __device__ __forceinline__ const lds_f *s0() const { return z; }
    __device__ __forceinline__ int s1() const { return ldz; }
    __device__ __forceinline__ const lds_f *s2() const { return table; }
    __device__ __forceinline__ int s3() const { return ldt; }
    __device__ __forceinline__ f32x4 shared(int m, int koff) const { return *reinterpret_cast<const lds_f4 *>(z + m * ldz + koff); }
    __device__ __forceinline__ f32x4 raw(int mt, int, int koff) const { return *reinterpret_cast<const lds_f4 *>(table + mt * ldt + koff); }
    static __device__ __forceinline__ f32x4 form(const f32x4 &sh, const f32x4 &r) { return relu_sum(sh, r); }
};

// Sixteen rows, row m = relu(z[m][:] + table[sel[m]][:]): the child of row m's tree reached by
// action sel[m], formed on the fly (one row tile only).
struct PickedRows {
    const lds_f *z;
    int ldz;
    const lds_f *table;
    int ldt;
    const lds_i *sel;
    const lds_f *trow;  // per lane: table row of this lane's A-operand row (lane & 15)
    static __device__ __forceinline__ PickedRows from(const lds_f *a, int b, const lds_f *c, int d, const lds_i *e) {
        return PickedRows{a, b, c, d, e, c + e[threadIdx.x & 15] * d};
    }
    __device__ __forceinline__ const lds_f *s0() const { return z; }
    __device__ __forceinline__ int s1() const { return ldz; }
    __device__ __forceinline__ const lds_f *s2() const { return table; }
    __device__ __forceinline__ int s3() const { return ldt; }
    __device__ __forceinline__ const lds_i *s4() const { return sel; }
    __device__ __forceinline__ f32x4 shared(int m, int koff) const { return *reinterpret_cast<const lds_f4 *>(z + m * ldz + koff); }
    __device__ __forceinline__ f32x4 raw(int, int, int koff) const { return *reinterpret_cast<const lds_f4 *>(trow + koff); }
    static __device__ __forceinline__ f32x4 form(const f32x4 &sh, const f32x4 &r) { return relu_sum(sh, r); }
};

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

// Several dense layers over the SAME input rows in one pass (e.g. the policy head's first
// layer and the dynamics merge layer both read the latent): their column tiles form one list
// that is split evenly over the waves, so narrow layers do not leave waves idle and the
// workgroup pays one barrier for all of them.  All parts must share Kp.
//
// A wave walks its tiles NTW at a time.  The weights of those tiles are consumed as a stream
// of batches of kPrefetch k-steps; two register sets alternate so that batch i+1 is in flight
// (global_load_dwordx4, 1 KiB per tile and k-step) while the MFMAs of batch i issue, and the
// stream runs on from one tile group into the next without draining.  Every loop bound below
// is wave-uniform (the wave index goes through readfirstlane), so control flow is scalar.
template <int MT, int NTW, int NPARTS, typename Rows>
__device__ __forceinline__ void stream_tiles(const Rows &rows, const float *__restrict__ packed,
                                             const DensePart (&parts)[NPARTS], int KT, int first_tile, int ngroups) {
    // tall row tiles keep many accumulators and operands: their weight batches are shorter (a
    // batch still covers >= 100 MFMAs), which keeps the routine out of the callee-saved registers
    constexpr int KP = MT >= 4 ? 2 : kPrefetch;
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    const int nb = (KT + KP - 1) / KP;  // batches per tile group
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
            locate(first_tile + fg * NTW + t, part, nt);
            fbase[t] = (glb_f4 *)(packed + pick_part(parts, part).w_off) + (size_t)nt * KT * 64;
        }
    };
    fetch_bases();
    auto fetch = [&](f32x4 (&w)[NTW][KP]) {
        const int kt0 = fb * KP;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            // one scalar base per tile and batch, the k-steps at constant offsets (1 KiB apart):
            // a group's last batch may read up to KP - 1 k-steps past its tile, i.e. into the next
            // tile or the slack every packed MLP ends with (mz_mlp_plan); those values feed no MFMA
            glb_f4 *base = fbase[t] + (size_t)kt0 * 64;
#pragma unroll
            for (int j = 0; j < KP; ++j) w[t][j] = base[j * 64 + lane];
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
    f32x4 acc[NTW][MT];
    float bias[NTW];
    int c_part[NTW], c_nt[NTW];
    f32x4 xc[MT];
    auto compute = [&](const f32x4 (&w)[NTW][KP]) {
        const int kt0 = cb * KP;
        if (cb == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                locate(first_tile + cg * NTW + t, c_part[t], c_nt[t]);
                // requested now, needed only by the epilogue after the last batch
                bias[t] = ((glb_f *)packed)[pick_part(parts, c_part[t]).b_off + c_nt[t] * 16 + m];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // The A operands run one k-step ahead of the MFMAs that consume them (xc: current,
        // xn: next; the sequence wraps to k = 0 at the end of a tile group), so their LDS
        // latency and, for ChildRows, the vector work forming them hide behind the MFMAs.
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            // steps past the end of K (last batch of a group) skip their MFMAs only; the
            // operand pipeline advances unconditionally so that it is pure register renaming
            const bool valid = kt0 + j < KT;
            const int kn = (kt0 + j + 1 < KT) ? kt0 + j + 1 : 0;
            // every LDS read of the next k-step is issued before this k-step's MFMAs ...
            f32x4 xn[MT];
            const f32x4 sh = rows.shared(m, kn * 16 + 4 * kq);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[mt] = rows.raw(mt, m, kn * 16 + 4 * kq);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (valid) {
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[mt].x, w[t][j].x, acc[t][mt], 0, 0, 0);
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[mt].y, w[t][j].y, acc[t][mt], 0, 0, 0);
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[mt].z, w[t][j].z, acc[t][mt], 0, 0, 0);
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[mt].w, w[t][j].w, acc[t][mt], 0, 0, 0);
                    }
                }
                // ... and formed into operands behind the MFMAs of one row tile each
                xn[mt] = Rows::form(sh, xn[mt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xc[mt] = xn[mt];
        }
        if (++cb == nb) {
            cb = 0;
            ++cg;
            if (cg == ngroups) { MZ_TL(6) }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                store_tile<MT>(acc[t], c_nt[t], bias[t], pick_part(parts, c_part[t]), (glb_f *)packed);
            }
        }
    };
    // NS register sets rotate: NS - 1 batches are in flight while one is consumed.  Two is the measured
    // optimum: a third set for the short row tiles (-DMZ_WEIGHT_SETS=3) left the search 1.4 % slower --
    // the weight stream is bound by what a compute unit gets from L2, not by the latency one wave sees.
    constexpr int NS = MT >= 4 ? 2 : kWeightSets;
    f32x4 w[NS][NTW][KP];
    MZ_TL(1)
    // the weights have the longest way: requested before the operand pipeline is primed
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) fetch(w[u]);
    MZ_TL(2)
    {
        const f32x4 sh = rows.shared(m, 4 * kq);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xc[mt] = Rows::form(sh, rows.raw(mt, m, 4 * kq));
    }
    for (int i = 0; i < total; i += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            fetch(w[(u + NS - 1) % NS]);
            if (i + u < total) compute(w[u]);
        }
        if (i == 0) { MZ_TL(3) }
    }
    MZ_TL(5)
}

// Entry points (not inlined: own register allocation).  Kp is shared by all parts.  No
// trailing barrier.
template <int MT, int NPARTS, typename Rows>
__device__ __forceinline__ void dense_group_body(const Rows &rows, const float *packed, int Kp,
                                                 const DensePart (&parts)[NPARTS], int wave0, int nwaves) {
    const int wave = uni(threadIdx.x >> 6) - wave0;
    const int KT = Kp >> 4;
    int total = 0;
#pragma unroll
    for (int i = 0; i < NPARTS; ++i) total += parts[i].Np >> 4;
    // tiles are dealt as evenly as possible: waves w and w+4 usually share a SIMD, so the
    // larger shares go to the first waves and every SIMD ends up with nearly the same count
    const int lg = 31 - __builtin_clz(nwaves);
    const int share = total >> lg, extra = total - (share << lg);
    const int first = wave * share + min(wave, extra);
    const int count = share + (wave < extra ? 1 : 0);
    if (count <= 0) return;
    const int pairs = count >> 1;
    if (pairs > 0) stream_tiles<MT, 2, NPARTS>(rows, packed, parts, KT, first, pairs);
    if (count & 1) stream_tiles<MT, 1, NPARTS>(rows, packed, parts, KT, first + 2 * pairs, 1);
}
// Out-of-line entry points.  Every argument is a scalar: aggregates would be passed through
// per-lane stack memory (written 64 times per wave and eventually to HBM), scalars travel in
// registers.  `Rows` is rebuilt from four scalars (see LdsRows / ChildRows ::from).
template <int MT, typename Rows>
__device__ __noinline__ void dense_group1_impl(const lds_f *r0, int r1, const lds_f *r2, int r3, const lds_i *r4,
                                               const float *packed, int Kp, int w_off, int b_off, int N, int Np, int act,
                                               lds_f *Ys, int ldy, const lds_f *row_add, int row_add_ld,
                                               const lds_i *row_sel, int dot_w, lds_f *dot_out, int wave0, int nwaves) {
    const DensePart parts[1] = {
        DensePart{w_off, b_off, N, Np, act, Ys, ldy, row_add, row_add_ld, row_sel, dot_w, dot_out}.uniform()};
    dense_group_body<MT, 1>(Rows::from(uni_lds(r0), uni(r1), uni_lds(r2), uni(r3), uni_lds(r4)), uni_ptr(packed), uni(Kp),
                            parts, uni(wave0), uni(nwaves));
}
template <int MT, typename Rows>
__device__ __noinline__ void dense_group2_impl(const lds_f *r0, int r1, const lds_f *r2, int r3, const lds_i *r4,
                                               const float *packed, int Kp, int w0, int b0, int N0, int Np0, int act0,
                                               lds_f *Y0, int ldy0, int dw0, lds_f *do0, int w1, int b1, int N1, int Np1,
                                               int act1, lds_f *Y1, int ldy1, int dw1, lds_f *do1, int wave0, int nwaves) {
    const DensePart parts[2] = {DensePart{w0, b0, N0, Np0, act0, Y0, ldy0, nullptr, 0, nullptr, dw0, do0}.uniform(),
                                DensePart{w1, b1, N1, Np1, act1, Y1, ldy1, nullptr, 0, nullptr, dw1, do1}.uniform()};
    dense_group_body<MT, 2>(Rows::from(uni_lds(r0), uni(r1), uni_lds(r2), uni(r3), uni_lds(r4)), uni_ptr(packed), uni(Kp),
                            parts, uni(wave0), uni(nwaves));
}
template <int MT, typename Rows>
__device__ __forceinline__ void dense_group1(const Rows &rows, const float *packed, int Kp, const DensePart &p,
                                             const Team &tm = whole_workgroup()) {
    dense_group1_impl<MT, Rows>(rows.s0(), rows.s1(), rows.s2(), rows.s3(), rows.s4(), packed, Kp, p.w_off, p.b_off, p.N,
                                p.Np, p.act, p.Ys, p.ldy, p.row_add, p.row_add_ld, p.row_sel, p.dot_w_off, p.dot_out,
                                tm.wave0, tm.nwaves);
}
template <int MT, typename Rows>
__device__ __forceinline__ void dense_group2(const Rows &rows, const float *packed, int Kp, const DensePart &p,
                                             const DensePart &q, const Team &tm = whole_workgroup()) {
    dense_group2_impl<MT, Rows>(rows.s0(), rows.s1(), rows.s2(), rows.s3(), rows.s4(), packed, Kp, p.w_off, p.b_off, p.N,
                                p.Np, p.act, p.Ys, p.ldy, p.dot_w_off, p.dot_out, q.w_off, q.b_off, q.N, q.Np, q.act, q.Ys,
                                q.ldy, q.dot_w_off, q.dot_out, tm.wave0, tm.nwaves);
}

// Second half of a folded scalar layer: out[row * out_ld] = act(bias + sum over the tiles of
// dot[tile][row]), tiles in ascending order.  `rows` = 16*MT of the producing call.  No sync.
__device__ __forceinline__ void finish_dot(const lds_f *dot, int tiles, int rows, const float *__restrict__ packed,
                                           const mz_packed_layer &L1, lds_f *out, int out_ld, const Team &tm) {
    const float b = packed[L1.b_off];
    for (int r = tm.tid(); r < rows; r += tm.nthreads()) {
        float s = 0.0f;
        for (int t = 0; t < tiles; ++t) s += dot[t * rows + r];
        out[r * out_ld] = activate(s + b, L1.act);
    }
}

// Weight sources of dense_narrow: chunk q (4 consecutive k) of output column `col`.
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
// all but one wave, so these run on the vector ALU instead.  Four adjacent lanes share one
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

// One dense layer over the workgroup's row tile.  Xs: LDS [16*MT][ldx], columns [K, Kp)
// must be zero.  Ys: LDS [16*MT][ldy]; columns [N, Np) are written as zero.
// `row_add` (optional, LDS): per-row additive table row_add[row_sel[r]][n] added before the
// activation (the action-embedding contribution of the dynamics merge layer).
// `narrow_lds` (optional): LDS copy of a narrow layer's weights made by stage_narrow.
// Ends with a team sync: Ys is complete and Xs may be overwritten afterwards.
template <int MT>
__device__ __forceinline__ void dense_tile(const lds_f *Xs, int ldx, const float *__restrict__ packed,
                                           const mz_packed_layer &L, lds_f *Ys, int ldy, const lds_f *row_add,
                                           int row_add_ld, const lds_i *row_sel, const lds_f *narrow_lds, Team &tm) {
    if (L.N <= 8 && row_add == nullptr) {
        if (narrow_lds)
            dense_narrow<LdsNarrowWeights>(Xs, ldx, 16 * MT, nullptr, nullptr, narrow_lds, L.Kp, L.N, L.act, Ys, ldy,
                                           tm.wave0, tm.nwaves);
        else
            dense_narrow<PackedNarrowWeights>(Xs, ldx, 16 * MT, packed + L.w_off, packed + L.b_off, nullptr, L.Kp, L.N,
                                              L.act, Ys, ldy, tm.wave0, tm.nwaves);
        tm.sync();
        return;
    }
    dense_group1<MT>(LdsRows{Xs, ldx}, packed, L.Kp, make_part(L, Ys, ldy, L.act, row_add, row_add_ld, row_sel), tm);
    tm.sync();
}
template <int MT>
__device__ __forceinline__ void dense_tile(const lds_f *Xs, int ldx, const float *__restrict__ packed,
                                           const mz_packed_layer &L, lds_f *Ys, int ldy, const lds_f *row_add = nullptr,
                                           int row_add_ld = 0, const lds_i *row_sel = nullptr,
                                           const lds_f *narrow_lds = nullptr) {
    Team tm = whole_workgroup();
    dense_tile<MT>(Xs, ldx, packed, L, Ys, ldy, row_add, row_add_ld, row_sel, narrow_lds, tm);
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
    L._pad = 0;
    L.w_off = uni((int)s.w_off);
    L.b_off = uni((int)s.b_off);
    return L;
}

// A head that ends in [wide layer, one-output layer] has that pair folded: the wide layer's
// epilogue forms the dot products with the output layer's weights (DensePart::dot_out).
template <typename MLP>
__host__ __device__ inline bool tail_foldable(const MLP &M) {
    const int n = M.n_layers;
    return n >= 2 && M.layer[n - 1].N == 1 && M.layer[n - 2].N > 8 && M.layer[n - 1].K == M.layer[n - 2].N;
}

// Runs layers [first, n) of `M` over the team's row tile (input `src`, row stride `ld`),
// ping-ponging between two LDS buffers.  Returns the buffer that holds the output, row stride
// *out_ld.  A MZ_ACT_SOFTMAX layer is a linear layer followed by softmax_rows.  `narrow`
// (optional): byte offsets into `smem` of LDS copies of narrow layers' weights, or -1.
// Every layer ends with a team sync.
template <int MT, typename MLP, typename NarrowPtr = const int *>
__device__ __forceinline__ lds_f *mlp_tile(Team &tm, const MLP &M, int first, const float *__restrict__ packed,
                                           lds_f *src, int ld, lds_f *buf_a, lds_f *buf_b, int *out_ld,
                                           unsigned char *smem = nullptr, NarrowPtr narrow = nullptr) {
    const int n = uni(M.n_layers);
    const bool fold = uni((int)tail_foldable(M)) && first <= n - 2;
    const int plain_end = fold ? n - 2 : n;
    for (int i = first; i < plain_end; ++i) {
        const mz_packed_layer L = load_layer(M.layer[i]);
        lds_f *dst = (src == buf_a) ? buf_b : buf_a;
        const int ldo = lds_stride(L.N);
        const int nwo = narrow ? uni(narrow[i]) : -1;
        const lds_f *nw = nwo >= 0 ? to_lds(smem + nwo) : nullptr;
        dense_tile<MT>(src, ld, packed, L, dst, ldo, nullptr, 0, nullptr, nw, tm);
        if (L.act == MZ_ACT_SOFTMAX) softmax_rows(dst, ldo, 16 * MT, L.N, 1, tm);
        src = dst;
        ld = ldo;
    }
    if (fold) {
        const mz_packed_layer L = load_layer(M.layer[n - 2]), L1 = load_layer(M.layer[n - 1]);
        lds_f *dst = (src == buf_a) ? buf_b : buf_a;
        dense_group1<MT>(LdsRows{src, ld}, packed, L.Kp, make_dot_part(L, L1, dst), tm);
        tm.sync();
        finish_dot(dst, L.Np >> 4, 16 * MT, packed, L1, dst, 1, tm);  // in place: row r ends in dst[r]
        tm.sync();
        src = dst;
        ld = 1;
    }
    *out_ld = ld;
    return src;
}
template <int MT>
__device__ __forceinline__ lds_f *mlp_tile(const mz_packed_mlp &M, const float *__restrict__ packed, lds_f *in_buf,
                                           int in_ld, lds_f *buf_a, lds_f *buf_b, int *out_ld) {
    Team tm = whole_workgroup();
    return mlp_tile<MT>(tm, M, 0, packed, in_buf, in_ld, buf_a, buf_b, out_ld);
}

// Cooperative load of up to `rows` global rows (row r from row_ptr(r), NULL: zeros) into an LDS tile
// [TM][ld], zero-filling columns [K, ld) and rows >= rows.  Two round trips to memory, not one per
// element: a wave takes whole rows, asks for the row's address once (wave-uniform, so it lives in scalar
// registers) and requests all of the row's values before the first is stored.
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

}  // namespace mzd
