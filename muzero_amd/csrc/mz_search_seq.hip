// Second form of the single-launch search (src/search/mcts.py:44-74 for G trees): the same
// arithmetic as k_search (mz_search.hip) for the network shape the benchmark configuration and
// the reference's config.yaml heads have, organised around the resource that bounds that kernel.
//
// A 16-tree workgroup reads ~1 MB of packed weights per simulation round for 16 rows of work.
// They come from L2 (the whole model is 1 MB) into registers at no more than ~60 bytes per clock
// and compute unit; a wave that issues a tile's loads is held for as long as the port needs to
// take them, every matrix step would start with a round trip to L2, and between the matrix steps
// lie latency chains (selection, softmax, backup) during which the port would idle.  Here all eight
// waves run ONE sequential program per round, separated by workgroup barriers:
//
//   select, leaf latent          (implicit child latents: formed from the parent's merge row)
//   merge(leaf)                                         (the drawn child's latent needs only this)
//   policy 1 for leaf AND child as one 32-row pass | merge(drawn child)
//   policy 2 for leaf AND child as one 32-row pass      (their weights are read once, not twice)
//   policy 3 for the 32 rows (one column tile: one wave), softmax; priors of the leaf, greedy action of the child
//   value 1 (rollout latent) | reward 1 of the rollout latent
//   value 2
//   rollout sums, backup                                (the leaf's child records are written beside the rollout latent)
//
// Where every wave has a full share of every layer's column tiles (the benchmark's widths) the round runs a
// schedule written out for those shares, with a third register slot (template parameter FULL below).
//
// The rewards of the A children of every expansion (src/search/expansion.py:29-33) are outputs only --
// no selection, rollout or backup reads a node's reward -- so they are not computed inside the rounds:
// after the last round the expansions' merge rows are read back from z_pool and reward 1 runs over all
// (S + 1) * A * 16 child rows in one dense, barrier-free phase (the rounds' serial chain is 20 % shorter for it).
//
// Weights flow through a ring of two register slots per wave, a whole column tile (all its
// k-chunks, 12-16 KiB per wave) per slot: while one tile is multiplied the next is in flight, and
// the ring runs on ACROSS steps -- the first tile of the next step is requested before the current
// step's epilogue, barrier and the vector work between the steps, so no step starts with an
// exposed round trip; where a latency chain follows that only some waves run, the OTHER waves issue
// the requests right after the barrier, beside it.  Biases and the folded one-output layers' weights are staged in LDS once:
// inside a round the only global loads are the ring's (a small load issued behind a tile fetch
// would wait for the whole tile).  Steps are separated by workgroup barriers (no wave teams).
//
// Per accumulator the instruction sequence is the one stream_tiles (mz_mlp_device.h) issues and
// the epilogues are store_tile_core: node records, priors, rewards and rollouts equal k_search's
// and the per-stage kernels' bit for bit (tests/test_gpu_search.py).
//
// Covers: implicit child latents (pool->latent == NULL), model_look_ahead = 1, policy head
// [matrix, matrix, softmax layer of at most 8 actions, its hidden widths in as many k-chunks], value head [matrix, matrix, 1] and reward head [matrix, 1]
// with the last two layers folded, relu after every matrix layer, at most 8 column tiles in value 2 and reward 1 and 16 in
// the other layers, k-chunk counts matching an instantiation below.  mz_search_run falls back to k_search for everything else.
//
// Compiled with -ffp-contract=off (every rounding is part of the contract).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mz_common.h"
#include "mz_mlp_device.h"
#include "mz_search_lds.h"
#include "mz_tree_device.h"

using namespace mzd;
using namespace mzs;

namespace {


// LDS layout.  Every offset is a compile-time constant of the instantiation (k-chunk counts): buffers whose size
// depends on run-time values -- action count, head widths -- are laid out for the largest values the kernel accepts
// (8 actions, 16 column tiles per layer, 8 in value 2 / reward 1), so the kernel addresses LDS with literals instead
// of two dozen run-time bases held in (spilled) scalar registers; only the square-root / logarithm tables, whose
// length is a launch parameter, follow at the end.
//   xi: split image of the current input latents (leaf, drawn child, rollout latent in turn); z / zs: merge
//   pre-activations of the leaf / of the drawn child (fp32 rows, one buffer: their lives do not overlap); h1: policy
//   layer 1 of leaf (tile 0) and child (tile 1), split image -- and, once policy 2 has read it, hv: value layer 1
//   (split image) and lg: logits / policies of the 32 rows; h2: policy layer 2 of leaf and child; res: resident merge tiles; dote / dotv: partial sums of the folded layers; b_*: padded biases; dw_*: the
//   folded one-output layers' weights by k.  After the last round everything from `xi` up to `dote` is one arena for
//   the children's rewards: ew holds every column tile of reward 1 (as packed: two fp16 planes per k-chunk).
constexpr int kSeqMaxA = 8, kSeqMaxTiles = 2 * kWaves, kSeqMaxFoldTiles = kWaves;
template <int KCD, int KCP, int KCV>
struct SeqLayout {
    static constexpr int ld_d = KCD * 32 + 4, ld_p = KCP * 32 + 4, ld_q = KCP * 32 + 4, ld_v = KCV * 32 + 4, ld_l = 32 + 4;
    static constexpr size_t a16(size_t v) { return (v + 15) & ~(size_t)15; }
    static constexpr size_t tile_d = sizeof(float) * kTrees * ld_d;
    static constexpr size_t tab = sizeof(float) * kLdsReserve;
    static constexpr size_t xi = a16(tab + sizeof(float) * kSeqMaxA * ld_d);  // row tile 0: leaf (later the rollout latent), row tile 1: drawn child
    static constexpr size_t ew = xi;
    static constexpr size_t z = a16(xi + 2 * tile_d);
    static constexpr size_t zs = z;    // (the leaf's rows are stored and the child formed before the child's merge writes here)
    static constexpr size_t h1 = a16(z + tile_d);
    static constexpr size_t h1_bytes = sizeof(float) * 2 * kTrees * ld_p;
    static constexpr size_t h2 = a16(h1 + h1_bytes);
    static constexpr size_t hv = h1;   // (policy 1's rows are dead once policy 2 has read them)
    static constexpr size_t lg = a16(hv + sizeof(float) * kTrees * ld_v);
    static_assert(lg + sizeof(float) * 2 * kTrees * ld_l <= h1 + h1_bytes, "value 1's rows and the logits share policy 1's buffer");
    // merge column tiles beyond one per wave (as packed), resident for the whole search: a wave's second merge tile is
    // read from here, twice a round, instead of from L2
    static constexpr int res_tiles = 3;
    static constexpr size_t res = a16(h2 + sizeof(float) * 2 * kTrees * ld_q);
    static constexpr size_t dote = a16(res + sizeof(float) * (size_t)res_tiles * KCD * 512);
    static constexpr size_t ew_bytes = sizeof(float) * (size_t)(16 * kSeqMaxFoldTiles) * (KCD * 32);  // reward 1 at its widest
    static_assert(ew + ew_bytes <= dote, "the arena of the phase after the rounds fits in front of the small buffers");
    static constexpr size_t dotv = a16(dote + sizeof(float) * 4 * kSeqMaxFoldTiles * kTrees);
    static constexpr size_t b_p0 = a16(dotv + sizeof(float) * 4 * kSeqMaxFoldTiles * kTrees);
    static constexpr size_t b_m = b_p0 + sizeof(float) * 16 * kSeqMaxTiles;
    static constexpr size_t b_p1 = b_m + sizeof(float) * 16 * kSeqMaxTiles;
    static constexpr size_t b_v0 = b_p1 + sizeof(float) * 16 * kSeqMaxTiles;
    static constexpr size_t b_v1 = b_v0 + sizeof(float) * 16 * kSeqMaxTiles;           // + the one-output layer's bias
    static constexpr size_t b_r0 = b_v1 + sizeof(float) * (16 * kSeqMaxFoldTiles + 4);
    static constexpr size_t dw_v = b_r0 + sizeof(float) * (16 * kSeqMaxFoldTiles + 4);
    static constexpr size_t dw_r = dw_v + sizeof(float) * 16 * kSeqMaxFoldTiles;
    static constexpr size_t b_p2 = dw_r + sizeof(float) * 16 * kSeqMaxFoldTiles;
    static constexpr size_t prior = b_p2 + sizeof(float) * 16;
    static constexpr size_t roll = prior + sizeof(float) * kTrees * kSeqMaxA;
    static constexpr size_t ints = roll + sizeof(float) * kTrees * 2;
    static constexpr size_t path = ints + sizeof(int) * kTrees * 8;
    static constexpr size_t tabs = a16(path + sizeof(int) * kTrees * kPathCap);
    static constexpr size_t total(int tab_len) { return tabs + sizeof(double) * 2 * tab_len; }
};

// One ring slot: a whole column tile of a matrix layer (up to KMAX k-chunks x two planes of 16 bytes per lane).
template <int KMAX>
struct Slot {
    f32x4 w[KMAX][2];
};

template <int KC, int KMAX>
__device__ __forceinline__ void fetch_tile(Slot<KMAX> &s, const float *tile, int lane) {
    glb_f4 *b = (glb_f4 *)tile + lane;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        s.w[c][0] = b[(2 * c) * 64];
        s.w[c][1] = b[(2 * c + 1) * 64];
    }
}

// Ends the life of a slot's contents for the register allocator (no instruction): a slot filled under one condition
// and used under another would otherwise count as live from its last use all the way round the loop to the next fill.
template <int KMAX>
__device__ __forceinline__ void forget(Slot<KMAX> &s) {
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        asm volatile("" : "=v"(s.w[c][0]));
        asm volatile("" : "=v"(s.w[c][1]));
    }
}

// acc = tile x rows, the rows' split image at img (this lane's 16 bytes of chunk 0 of row tile 0), row tiles
// x_tile floats apart.  Same instruction sequence per accumulator as stream_tiles.
template <int KC, int MT, int AHEAD = (MT == 1 ? 2 : 1), int KMAX>
__device__ __forceinline__ void mma_tile(const Slot<KMAX> &s, const lds_f *img, int x_tile, f32x4 (&hi)[MT], f32x4 (&lo)[MT]) {
    // The rows' operands are read from LDS kAhead chunks ahead of the matrix instructions that use them, with
    // scheduling fences between the groups: left to itself the compiler issues each chunk's reads right in front of
    // its products and every chunk pays an LDS round trip.
    constexpr int kAhead = AHEAD, kBuf = kAhead + 1;
    h8 ah[kBuf][MT], al[kBuf][MT];
    auto read = [&](int c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const lds_f *p = img + mt * x_tile + c * kChunkFloats;
            ah[c % kBuf][mt] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p));
            al[c % kBuf][mt] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p + KC * kChunkFloats));
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hi[mt] = lo[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < kAhead && c < KC; ++c) read(c);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (c + kAhead < KC) read(c + kAhead);
        __builtin_amdgcn_sched_barrier(0);
        const h8 wh = __builtin_bit_cast(h8, s.w[c][0]), wl = __builtin_bit_cast(h8, s.w[c][1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            hi[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[c % kBuf][mt], hi[mt], 0, 0, 0);
            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[c % kBuf][mt], lo[mt], 0, 0, 0);
            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[c % kBuf][mt], lo[mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same with the tile's weights read from LDS (a tile as packed, this lane's 16 bytes of chunk 0 at wt).
template <int KC, int MT>
__device__ __forceinline__ void mma_tile_lds(const lds_f *wt, const lds_f *img, int x_tile, f32x4 (&hi)[MT], f32x4 (&lo)[MT]) {
    constexpr int kAhead = 2, kBuf = kAhead + 1;
    h8 ah[kBuf][MT], al[kBuf][MT], wh[kBuf], wl[kBuf];
    auto read = [&](int c) {
        wh[c % kBuf] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(wt + (2 * c) * 256));
        wl[c % kBuf] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(wt + (2 * c + 1) * 256));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const lds_f *p = img + mt * x_tile + c * kChunkFloats;
            ah[c % kBuf][mt] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p));
            al[c % kBuf][mt] = __builtin_bit_cast(h8, *reinterpret_cast<const lds_f4 *>(p + KC * kChunkFloats));
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hi[mt] = lo[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < kAhead && c < KC; ++c) read(c);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        if (c + kAhead < KC) read(c + kAhead);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            hi[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c % kBuf], ah[c % kBuf][mt], hi[mt], 0, 0, 0);
            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c % kBuf], al[c % kBuf][mt], lo[mt], 0, 0, 0);
            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c % kBuf], ah[c % kBuf][mt], lo[mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// v of the lane a DPP control selects (all rows and banks enabled; a lane without a source keeps its own value).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false));
}

// A float4 of an LDS array as a register value.
__device__ __forceinline__ f32x4 lds4(const lds_f *p) { return *reinterpret_cast<const lds_f4 *>(p); }

// Epilogues of one 16-column tile for layers whose activation is relu (RELU) or none: the arithmetic of
// store_tile_core, step for step, without its run-time choices.  C layout: row = lane & 15, columns
// 16 nt + 4 (lane >> 4) + r.
template <bool RELU>
__device__ __forceinline__ f32x4 epi_value(const f32x4 &hi, const f32x4 &lo, const f32x4 &b) {
    f32x4 v;
    v.x = (hi.x + __fmul_rn(lo.x, kLoUnscale)) + b.x;
    v.y = (hi.y + __fmul_rn(lo.y, kLoUnscale)) + b.y;
    v.z = (hi.z + __fmul_rn(lo.z, kLoUnscale)) + b.z;
    v.w = (hi.w + __fmul_rn(lo.w, kLoUnscale)) + b.w;
    if (RELU) {
        v.x = v.x > 0.0f ? v.x : 0.0f;
        v.y = v.y > 0.0f ? v.y : 0.0f;
        v.z = v.z > 0.0f ? v.z : 0.0f;
        v.w = v.w > 0.0f ? v.w : 0.0f;
    }
    return v;
}
// fp32 rows: Ys[row][16 nt + ..]
template <int MT, bool RELU>
__device__ __forceinline__ void epi_rows(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, const f32x4 &b, lds_f *Ys, int ldy, int lane) {
    const int j = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        *reinterpret_cast<lds_f4 *>(Ys + (mt * 16 + j) * ldy + nt * 16 + 4 * rq) = epi_value<RELU>(hi[mt], lo[mt], b);
}
// split image of the next layer's operand (row tiles s_tile floats apart, s_kc chunks per plane, nt_count column tiles)
template <int MT, bool RELU>
__device__ __forceinline__ void epi_split(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, const f32x4 &b, lds_f *Ss, int s_tile,
                                          int s_kc, int nt_count, int lane) {
    const int j = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x2 h, l;
        split4(epi_value<RELU>(hi[mt], lo[mt], b), h, l);
        lds_f *img = Ss + mt * s_tile + (nt >> 1) * kChunkFloats + (2 * (nt & 1) + (rq >> 1)) * 64 + j * 4 + 2 * (rq & 1);
        *reinterpret_cast<lds_f2 *>(img) = h;
        *reinterpret_cast<lds_f2 *>(img + s_kc * kChunkFloats) = l;
        if ((nt & 1) == 0 && nt + 1 == nt_count) {
            const f32x2 z = {0.f, 0.f};
            *reinterpret_cast<lds_f2 *>(img + 128) = z;
            *reinterpret_cast<lds_f2 *>(img + 128 + s_kc * kChunkFloats) = z;
        }
    }
}
// folded one-output layer: partial sums dot[(4 nt + quarter) * 16 MT + row]
template <int MT, bool RELU>
__device__ __forceinline__ void epi_dot(const f32x4 (&hi)[MT], const f32x4 (&lo)[MT], int nt, const f32x4 &b, const f32x4 &w2, lds_f *dot,
                                        int lane) {
    const int j = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 v = epi_value<RELU>(hi[mt], lo[mt], b);
        float s = __fmul_rn(v.x, w2.x);
        s += __fmul_rn(v.y, w2.y);
        s += __fmul_rn(v.z, w2.z);
        s += __fmul_rn(v.w, w2.w);
        dot[(nt * 4 + rq) * (16 * MT) + mt * 16 + j] = s;
    }
}

// Sum of n (<= 32) partial sums p[0], p[stride], ... in ascending order (the order of finish_dot), with all the
// loads issued before the first addition: one LDS round trip instead of one per few additions.
__device__ __forceinline__ float ordered_sum(const lds_f *p, int stride, int n) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = p[min(i, n - 1) * stride];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < n) s += v[i];
    return s;
}

// The same for a length known at compile time (constant offsets, no index arithmetic).
template <int N>
__device__ __forceinline__ float ordered_sum_n(const lds_f *p, int stride) {
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i * stride];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) s += v[i];
    return s;
}

// This wave's share of `total` column tiles, dealt in pairs (as dense_group_body): first tile and count.
__device__ __forceinline__ void deal_pairs(int total, int wave, int &first, int &count) {
    const int units = (total + 1) >> 1;
    const int share = units >> 3, extra = units - (share << 3);
    first = (wave * share + min(wave, extra)) * 2;
    const int groups = share + (wave < extra ? 1 : 0);
    count = max(0, min(groups * 2, total - first));
}

// STORE: the pool keeps every node's latent (written from the root rows and in the phase after the rounds).
// FULL: every wave has two column tiles in policy 1, policy 2 and value 1, one in value 2 and reward 1 and one or
// two in the merge (16 / 16 / 16 / 8 / 8 / 8-16 column tiles): the ring gets a third slot of the merge's width, and
// the schedule of a round is written out for exactly these shares --
//   * a wave's first merge tile stays in its slot from the leaf's merge to the drawn child's (the same weights:
//     requested once per round instead of twice), the merge tiles beyond eight are resident in LDS,
//   * policy 1's tiles are both on their way during the selection, policy 2's first tile is requested behind
//     policy 1's first product.
template <int KCD, int KCP, int KCV, bool STORE, bool FULL>
__global__ void __launch_bounds__(kThreads)
k_search_seq(mz_pool p, mz_packed_nets Pk, const float *__restrict__ blob, mz_search_params q,
             double *__restrict__ policy, float *__restrict__ policy32, double *__restrict__ value,
             int32_t *__restrict__ visits) {
    constexpr int KMAX = KCD > KCP ? (KCD > KCV ? KCD : KCV) : (KCP > KCV ? KCP : KCV);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    using L = SeqLayout<KCD, KCP, KCV>;
    lds_f *XI = to_lds(smem_raw + L::xi);
    lds_f *Z = to_lds(smem_raw + L::z);
    lds_f *Zs = to_lds(smem_raw + L::zs);
    lds_f *Tab = to_lds(smem_raw + L::tab);
    lds_f *H1 = to_lds(smem_raw + L::h1);
    lds_f *H2 = to_lds(smem_raw + L::h2);
    lds_f *Hv = to_lds(smem_raw + L::hv);
    lds_f *Lg = to_lds(smem_raw + L::lg);
    lds_f *DotE = to_lds(smem_raw + L::dote);
    lds_f *DotV = to_lds(smem_raw + L::dotv);
    lds_f *Bp0 = to_lds(smem_raw + L::b_p0), *Bm = to_lds(smem_raw + L::b_m), *Bp1 = to_lds(smem_raw + L::b_p1);
    lds_f *Bv0 = to_lds(smem_raw + L::b_v0), *Bv1 = to_lds(smem_raw + L::b_v1), *Br0 = to_lds(smem_raw + L::b_r0);
    lds_f *DwV = to_lds(smem_raw + L::dw_v), *DwR = to_lds(smem_raw + L::dw_r);
    lds_f *Bp2 = to_lds(smem_raw + L::b_p2);
    lds_f *s_prior = to_lds(smem_raw + L::prior);
    lds_f *s_roll = to_lds(smem_raw + L::roll);
    lds_i *s_leaf = to_lds_i(smem_raw + L::ints);
    lds_i *s_child = s_leaf + kTrees;
    lds_i *s_act = s_child + kTrees;    // action of the drawn child
    lds_i *s_act2 = s_act + kTrees;     // greedy action of the rollout step
    lds_i *s_leaf_tp = s_act2 + kTrees;
    lds_i *s_depth = s_leaf_tp + kTrees;
    lds_i *s_root = s_depth + kTrees;
    int *s_path = reinterpret_cast<int *>(smem_raw + L::path);
    double *s_tab = reinterpret_cast<double *>(smem_raw + L::tabs);

    __shared__ unsigned long long s_stamp[32];
    unsigned long long t_prev = 0;
    const bool stamping = q.stamps != nullptr && blockIdx.x == 0;
    if (stamping && threadIdx.x < 32) s_stamp[threadIdx.x] = 0;
    if (stamping && threadIdx.x == 0) t_prev = clock64();
#define MZ_STAMP(slot)                                       \
    if (stamping && threadIdx.x == 0) {                      \
        const unsigned long long t_now = clock64();          \
        s_stamp[slot] += t_now - t_prev;                     \
        t_prev = t_now;                                      \
    }

    const int A = Pk.A, D = Pk.D, S = q.S;
    constexpr int ld_d = L::ld_d, ld_p = L::ld_p, ld_q = L::ld_q, ld_v = L::ld_v, ld_l = L::ld_l;
    const int wave = uni(threadIdx.x >> 6);
    int tid = threadIdx.x, lane = tid & 63;  // (re-derived every round: see the round loop)
    const int g0 = blockIdx.x * kTrees;
    const int trees = min(kTrees, p.G - g0);
    const int lpt = q.lanes_per_tree & 0xff;
    const int Dp16 = Pk.merge.layer[0].Np;
    // layers, as wave-uniform values
    const mz_packed_layer Lp0 = Pk.policy.layer[0], Lp1 = Pk.policy.layer[1], Lp2 = Pk.policy.layer[2], Lm0 = Pk.merge.layer[0];
    const mz_packed_layer Lv0 = Pk.value.layer[0], Lv1 = Pk.value.layer[1], Lv2 = Pk.value.layer[2];
    const mz_packed_layer Lr0 = Pk.reward.layer[0], Lr1 = Pk.reward.layer[1];
    const int np0 = Lp0.Np >> 4, nm0 = Lm0.Np >> 4, np1 = Lp1.Np >> 4, nv0 = Lv0.Np >> 4, nv1 = Lv1.Np >> 4, nr0 = Lr0.Np >> 4;
    auto tile_of = [&](const mz_packed_layer &Ly, int KC, int nt) { return blob + Ly.w_off + (size_t)nt * KC * 512; };

    // ---- one-time staging ----------------------------------------------------------------------
    for (int i = tid; i < q.tab_len; i += kThreads) {
        s_tab[i] = q.sqrt_tab[i];
        s_tab[q.tab_len + i] = q.log_tab[i];
    }
    for (int i = tid; i < A * ld_d; i += kThreads) {
        const int a = i / ld_d, c = i - a * ld_d;
        Tab[i] = (a < A && c < Pk.Dp) ? blob[Pk.action_table_off + (size_t)a * Pk.Dp + c] : 0.0f;
    }
    auto stage_bias = [&](lds_f *dst, const mz_packed_layer &Ly) {
        for (int i = tid; i < Ly.Np; i += kThreads) dst[i] = blob[Ly.b_off + i];
    };
    stage_bias(Bp0, Lp0), stage_bias(Bm, Lm0), stage_bias(Bp1, Lp1), stage_bias(Bv0, Lv0), stage_bias(Bv1, Lv1), stage_bias(Br0, Lr0), stage_bias(Bp2, Lp2);
    if (tid == 0) {
        Bv1[Lv1.Np] = blob[Lv2.b_off];
        Br0[Lr0.Np] = blob[Lr1.b_off];
    }
    // the one-output layers' weights by k: fp32 tile kt holds W[0][16 kt + 4 q + j] in lane 16 q, element j
    for (int k = tid; k < Lv1.Np; k += kThreads) DwV[k] = blob[Lv2.w_off + ((k >> 4) * 64 + 16 * ((k >> 2) & 3)) * 4 + (k & 3)];
    for (int k = tid; k < Lr0.Np; k += kThreads) DwR[k] = blob[Lr1.w_off + ((k >> 4) * 64 + 16 * ((k >> 2) & 3)) * 4 + (k & 3)];
    for (int i = tid; i < kTrees * ld_d; i += kThreads) {  // root latents as fp32 rows (Z is free until the first merge)
        const int t = i / ld_d, c = i - t * ld_d;
        const float v = (t < trees && c < D) ? q.root_latent[(size_t)(g0 + t) * D + c] : 0.0f;
        Z[i] = v;
        if (STORE && t < trees && c < D) p.latent[(size_t)(g0 + t) * p.cap * D + c] = v;  // node 0
    }
    if (tid < kTrees) {
        int tp = 0;
        if (tid < trees) {
            const size_t o = (size_t)(g0 + tid) * p.cap;
            tp = q.root_to_play[g0 + tid];
            mzt::write_node(p, o, -1, tp, 0.0, 0.0f);
        }
        s_leaf[tid] = 0;
        s_leaf_tp[tid] = tp;
        s_root[tid] = tp;
        s_depth[tid] = 0;
        s_act[tid] = 0;
        s_act2[tid] = 0;
    }
    lds_f *RES = to_lds(smem_raw + L::res);
    if constexpr (FULL) {
        const glb_f4 *src = (const glb_f4 *)tile_of(Lm0, KCD, kWaves);
        for (int i = tid; i < (nm0 - kWaves) * KCD * 128; i += kThreads) *reinterpret_cast<lds_f4 *>(RES + i * 4) = src[i];
    }
    __syncthreads();
    rows_to_split(Z, ld_d, XI, ld_d, 1, KCD, ld_d, whole_workgroup());
    for (int i = tid; i < kTrees * ld_d; i += kThreads) XI[kTrees * ld_d + i] = 0.0f;  // (no drawn child in the root expansion)
    __syncthreads();
    MZ_STAMP(0)  // staging

    // ---- this wave's tiles in every step (static over the search) -----------------------------------
    int p0_first, p0_n, p1_first, p1_n, v0_first, v0_n;
    deal_pairs(np0, wave, p0_first, p0_n);
    deal_pairs(np1, wave, p1_first, p1_n);
    deal_pairs(nv0, wave, v0_first, v0_n);
    // merge: column tiles wave, wave + 8 (at most two: checked by the host)
    const int m_n = (wave < nm0 ? 1 : 0) + (wave + kWaves < nm0 ? 1 : 0);
    auto m_tile = [&](int j) { return tile_of(Lm0, KCD, min(wave + kWaves * min(j, max(m_n - 1, 0)), nm0 - 1)); };
    const float *m_tile0 = m_tile(0), *m_tile1 = m_tile(1);
    const float *p0_tile0 = tile_of(Lp0, KCD, min(p0_first, np0 - 1)), *p0_tile1 = tile_of(Lp0, KCD, min(p0_first + 1, np0 - 1));
    const float *p1_tile0 = tile_of(Lp1, KCP, min(p1_first, np1 - 1)), *p1_tile1 = tile_of(Lp1, KCP, min(p1_first + 1, np1 - 1));
    const float *v0_tile0 = tile_of(Lv0, KCD, min(v0_first, nv0 - 1)), *v0_tile1 = tile_of(Lv0, KCD, min(v0_first + 1, nv0 - 1));
    const float *v1_tile = tile_of(Lv1, KCV, min(wave, nv1 - 1));   // value 2 and reward 1: tile `wave` each
    const float *r0_tile = tile_of(Lr0, KCD, min(wave, nr0 - 1));

    Slot<KMAX> s0, s1;
    Slot<KCD> sx;  // FULL only

    // Runs a step: `n` tiles of KC chunks for this wave through the ring.  On entry slot 0 and slot 1 hold the
    // step's first two tiles (requested ahead by the caller, a whole inter-step phase earlier: prefetch2); the
    // step requests tile j + 2 as soon as tile j has been multiplied.  ptr(j): address of the wave's j-th tile
    // (j clamped by the callee); work(slot, j): multiply and store it.  Inside the loop the fetches are issued
    // unconditionally -- but for the one past an odd share -- so that the waits in front of the matrix instructions
    // count the loads in flight behind the tile they need instead of draining the queue.  MULTI = false: at most
    // two tiles per wave.
    auto run_step = [&](auto kc_tag, auto multi_tag, int n, auto &&ptr, auto &&work, auto &&after_first, auto &&odd_share) {
        constexpr int KC = decltype(kc_tag)::value;
        constexpr bool MULTI = decltype(multi_tag)::value;
        int j = 0;
        if constexpr (MULTI) {
            for (; j + 2 < n; j += 2) {
                work(s0, j);
                fetch_tile<KC>(s0, ptr(j + 2), lane);
                work(s1, j + 1);
                // (an odd share has no fourth tile: fetching its last one again, to keep the loads in flight countable,
                // cost more on the compute unit's L2 port than the drained queue costs the waves with an even share)
                if (j + 3 < n) fetch_tile<KC>(s1, ptr(j + 3), lane);
                else if (j + 3 == n) odd_share();  // slot 1 is free for good: the caller's request for the next step
            }
        }
        if (j < n) work(s0, j);
        after_first();  // slot 0 is free from here on
        if (j + 1 < n) work(s1, j + 1);
    };
    auto nothing = [] {};
    // Requests the first two tiles (kcn chunks each) of the step that follows the phase about to begin.
    auto prefetch2 = [&](const float *tile0, const float *tile1, int kcn) {
        if (kcn == KCD) fetch_tile<KCD>(s0, tile0, lane), fetch_tile<KCD>(s1, tile1, lane);
        else if (kcn == KCP) fetch_tile<KCP>(s0, tile0, lane), fetch_tile<KCP>(s1, tile1, lane);
        else fetch_tile<KCV>(s0, tile0, lane), fetch_tile<KCV>(s1, tile1, lane);
    };
    // the merge layer's tiles of this wave: one or two (a wave with one does not request it twice: its free slot
    // takes policy 1's second tile already, a whole latency phase before that step)
    auto prefetch_merge = [&] {
        fetch_tile<KCD>(s0, m_tile0, lane);
        if constexpr (FULL) {
            fetch_tile<KCD>(s1, p0_tile1, lane);
            fetch_tile<KCD>(sx, p0_tile0, lane);
        } else {
            fetch_tile<KCD>(s1, m_n > 1 ? m_tile1 : p0_tile1, lane);
        }
    };
    using MultiPair = std::true_type;
    using OnePair = std::false_type;

    lds_f *XC = XI + kTrees * ld_d;  // the drawn child's image
    // one column tile t of: the merge of sixteen rows (image at img) -> fp32 rows at dst; policy 1 of leaf AND drawn
    // child (two row tiles) -> H1 (split image); value 1 of the rollout latent -> Hv (split image); value 2 and
    // reward 1 with their folded one-output layers -> partial sums
    constexpr int kAhead1 = 2;
    auto work_merge = [&](auto &s, const lds_f *img, int t, lds_f *dst) {
        f32x4 hi[1], lo[1];
        mma_tile<KCD, 1, kAhead1>(s, img + lane * 4, 0, hi, lo);
        epi_rows<1, false>(hi, lo, t, lds4(Bm + t * 16 + 4 * (lane >> 4)), dst, ld_d, lane);
    };
    auto work_merge_resident = [&](const lds_f *img, lds_f *dst) {  // this wave's second merge tile
        const int t = wave + kWaves;
        f32x4 hi[1], lo[1];
        mma_tile_lds<KCD, 1>(RES + wave * (KCD * 512) + lane * 4, img + lane * 4, 0, hi, lo);
        epi_rows<1, false>(hi, lo, t, lds4(Bm + t * 16 + 4 * (lane >> 4)), dst, ld_d, lane);
    };
    auto work_policy1 = [&](auto &s, int t) {
        f32x4 hi[2], lo[2];
        mma_tile<KCD, 2>(s, XI + lane * 4, 16 * ld_d, hi, lo);
        epi_split<2, true>(hi, lo, t, lds4(Bp0 + t * 16 + 4 * (lane >> 4)), H1, 16 * ld_p, KCP, np0, lane);
    };
    auto work_value1 = [&](auto &s, int t) {
        f32x4 hi[1], lo[1];
        mma_tile<KCD, 1, kAhead1>(s, XI + lane * 4, 0, hi, lo);
        epi_split<1, true>(hi, lo, t, lds4(Bv0 + t * 16 + 4 * (lane >> 4)), Hv, 16 * ld_v, KCV, nv0, lane);
    };
    auto work_value2 = [&](auto &s) {
        f32x4 hi[1], lo[1];
        mma_tile<KCV, 1, kAhead1>(s, Hv + lane * 4, 0, hi, lo);
        epi_dot<1, true>(hi, lo, wave, lds4(Bv1 + wave * 16 + 4 * (lane >> 4)), lds4(DwV + wave * 16 + 4 * (lane >> 4)), DotV, lane);
    };
    auto work_reward1 = [&](auto &s) {
        f32x4 hi[1], lo[1];
        mma_tile<KCD, 1, kAhead1>(s, XI + lane * 4, 0, hi, lo);
        epi_dot<1, true>(hi, lo, wave, lds4(Br0 + wave * 16 + 4 * (lane >> 4)), lds4(DwR + wave * 16 + 4 * (lane >> 4)), DotE, lane);
    };
    // merge of the leaf (image in XI) -> Z (fp32 rows): at most two tiles per wave, both prefetched
    auto merge_leaf = [&] {
        run_step(std::integral_constant<int, KCD>{}, OnePair{}, m_n, m_tile,
                 [&](Slot<KMAX> &s, int j) { work_merge(s, XI, wave + kWaves * j, Z); },
                 [&] { fetch_tile<KCD>(s0, p0_tile0, lane); },  // policy 1's first tile: beside a second merge tile
                 nothing);
    };
    // policy 1 of leaf AND drawn child as one pass over two row tiles (its weights are read once per round) -> H1
    // (split image, two row tiles); with `child`, the merge of the drawn child (row tile 1) -> Zs behind it
    auto policy1_and_child_merge = [&](bool child) {
        const int n = p0_n + (child ? m_n : 0);
        run_step(std::integral_constant<int, KCD>{}, MultiPair{}, n,
                 [&](int j) {
                     j = min(j, n - 1);
                     return j < p0_n ? tile_of(Lp0, KCD, min(p0_first + j, np0 - 1)) : m_tile(j - p0_n);
                 },
                 [&](Slot<KMAX> &s, int j) {
                     if (j < p0_n) work_policy1(s, p0_first + j);
                     else work_merge(s, XC, wave + kWaves * (j - p0_n), Zs);
                 },
                 [&] { fetch_tile<KCP>(s0, p1_tile0, lane); },   // policy 2's first tile: beside this wave's last tile
                 [&] { fetch_tile<KCP>(s1, p1_tile1, lane); });  // its second: as early as an odd share frees slot 1
        if (!(n >= 3 && (n & 1))) fetch_tile<KCP>(s1, p1_tile1, lane);
    };
    // split image of sixteen rows relu(zsrc[t] + table[action of tree t]) -> dst (a 16-row tile); act_fixed >= 0:
    // the same action for every tree (the children of one action), else the trees' own actions `sel`
    auto form_item = [&](int i, const lds_f *zsrc, int act_fixed, const lds_i *sel, lds_f *dst) {
        const int l = i & 63, c = i >> 6, t = l & 15;
        const int k0 = 32 * c + 8 * (l >> 4);
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (k0 < Dp16) {
            const int a = act_fixed >= 0 ? act_fixed : sel[t];
            const lds_f *zr = zsrc + t * ld_d + k0, *tr = Tab + a * ld_d + k0;
            const f32x4 z0 = *reinterpret_cast<const lds_f4 *>(zr), z1 = *reinterpret_cast<const lds_f4 *>(zr + 4);
            const f32x4 b0 = *reinterpret_cast<const lds_f4 *>(tr), b1 = *reinterpret_cast<const lds_f4 *>(tr + 4);
            v0 = f32x4{fmaxf(z0.x + b0.x, 0.f), fmaxf(z0.y + b0.y, 0.f), fmaxf(z0.z + b0.z, 0.f), fmaxf(z0.w + b0.w, 0.f)};
            v1 = f32x4{fmaxf(z1.x + b1.x, 0.f), fmaxf(z1.y + b1.y, 0.f), fmaxf(z1.z + b1.z, 0.f), fmaxf(z1.w + b1.w, 0.f)};
        }
        f32x4 hi, lo;
        split8(v0, v1, hi, lo);
        lds_f *qd = dst + c * kChunkFloats + l * 4;
        *reinterpret_cast<lds_f4 *>(qd) = hi;
        *reinterpret_cast<lds_f4 *>(qd + KCD * kChunkFloats) = lo;
    };
    // expansion e = 0 is the root expansion (no simulation / backup follows it);
    // expansion e >= 1 belongs to simulation s = e - 1
    for (int e = 0; e <= S; ++e) {
        const int base = 1 + e * A;
        const bool sim = e > 0;
        mzt::NodeStat path_stat = {0.0, 0, 0};  // backup lanes: statistics of this lane's node on the selected path
        // Thread-dependent addresses are re-derived every round from an opaque copy of the thread index: hoisted out
        // of the loop they stay live through the whole kernel, spill, and every reload from scratch waits for ALL
        // outstanding loads -- the ring's tiles in flight included.
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;

        if (sim) {
            // ---- select, draw, leaf latent -----------------------------------------------------
            if (tid >= kThreads - kTrees) {  // the child each simulation starts from (src/search/mcts.py:64-66)
                const int t = tid - (kThreads - kTrees);
                int rc = 0;
                if (t < trees) rc = q.rand_child[(size_t)(g0 + t) * (S + 1) + e];
                rc = min(max(rc, 0), A - 1);
                s_act[t] = rc;
                s_child[t] = base + rc;
                if (q.child_trace && t < trees) q.child_trace[(size_t)(g0 + t) * S + (e - 1)] = base + rc;
            }
            // Issuing a tile's loads occupies a wave for as long as the compute unit's L2 port needs to take them, and a
            // barrier makes everybody wait for the issuing waves: the six waves that do not select request the first
            // tiles of the leaf's merge HERE, beside the selection (a latency chain of dependent record reads); the
            // two selecting waves have requested theirs at the end of the round before
            if (wave >= 2) prefetch_merge();
            if (tid < kTrees * lpt) {
                const int t = tid / lpt, a = tid % lpt;
                int leaf = 0, dep = 0;
                if (t < trees) {
                    const size_t o = (size_t)(g0 + t) * p.cap;
                    switch (lpt) {
                        case 2: leaf = mzt::select_leaf<2>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 4: leaf = mzt::select_leaf<4>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 8: leaf = mzt::select_leaf<8>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 16: leaf = mzt::select_leaf<16>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        default: leaf = mzt::select_leaf<32>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                    }
                    if (a == 0) {
                        leaf = max(leaf, 0);
                        s_leaf[t] = leaf;
                        s_depth[t] = dep;
                        s_leaf_tp[t] = (dep & 1) ? -s_root[t] : s_root[t];
                        if (q.leaf_trace) q.leaf_trace[(size_t)(g0 + t) * S + (e - 1)] = leaf;
                    }
                }
            }
            __syncthreads();
            MZ_STAMP(1)  // select
            // node n >= 1 is child a = (n - 1) % A of expansion (n - 1) / A: its latent is relu(z_pool[expansion] +
            // table[a]), formed straight into the split image
            {   // (KCD * 64 <= 512 operands: at most one per thread)
                const int i = tid;
                const int l = i & 63, c = i >> 6, t = l & 15;
                const int k0 = 32 * c + 8 * (l >> 4);
                const int n = max(s_leaf[t] - 1, 0), ex = n / A, a = n - ex * A;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                const bool mine = i < KCD * 64;
                const bool live = mine && t < trees && k0 < Dp16;
                float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (live) {
                    const float *zr = q.z_pool + ((size_t)(g0 + t) * (S + 1) + ex) * D + k0;
                    if ((D & 3) == 0) {
                        const float4 z0 = k0 < D ? *reinterpret_cast<const float4 *>(zr) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 z1 = k0 + 4 < D ? *reinterpret_cast<const float4 *>(zr + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        z[0] = z0.x, z[1] = z0.y, z[2] = z0.z, z[3] = z0.w, z[4] = z1.x, z[5] = z1.y, z[6] = z1.z, z[7] = z1.w;
                    } else {
                        for (int j = 0; j < 8; ++j) z[j] = k0 + j < D ? zr[j] : 0.0f;
                    }
                }
                // The backup of this round will add to the statistics of the nodes on the path just selected: nothing
                // writes them before that, so the backup lanes read them now and the backup itself waits for no load
                if (tid >= kThreads / 2 && tid < kThreads / 2 + kTrees * kPathCap) {
                    const int bt = (tid - kThreads / 2) / kPathCap, bi = (tid - kThreads / 2) % kPathCap;
                    const int dl = s_depth[bt];
                    if (bt < trees && dl + 1 <= kPathCap - 1 && bi >= 1 && bi <= dl + 1)
                        path_stat = mzt::load_stat(p, (size_t)(g0 + bt) * p.cap + s_path[bt * kPathCap + (dl - (bi - 1))]);
                }
                if (live) {
                    const lds_f *tr = Tab + a * ld_d + k0;
                    v0 = f32x4{fmaxf(z[0] + tr[0], 0.f), fmaxf(z[1] + tr[1], 0.f), fmaxf(z[2] + tr[2], 0.f), fmaxf(z[3] + tr[3], 0.f)};
                    v1 = f32x4{fmaxf(z[4] + tr[4], 0.f), fmaxf(z[5] + tr[5], 0.f), fmaxf(z[6] + tr[6], 0.f), fmaxf(z[7] + tr[7], 0.f)};
                }
                if (mine) {
                    f32x4 hi, lo;
                    split8(v0, v1, hi, lo);
                    lds_f *qd = XI + c * kChunkFloats + l * 4;
                    *reinterpret_cast<lds_f4 *>(qd) = hi;
                    *reinterpret_cast<lds_f4 *>(qd + KCD * kChunkFloats) = lo;
                }
            }
            __syncthreads();
            MZ_STAMP(2)  // leaf latent
        }

        // ---- merge of the leaf; the drawn child's latent; policy 1 of both | merge of the child -----------------
        if (!sim) prefetch_merge();  // (root expansion: nothing to hide it behind)
        if constexpr (FULL) {
            work_merge(s0, XI, wave, Z);  // (slot 0 keeps the tile for the drawn child's merge)
            if (m_n > 1) work_merge_resident(XI, Z);
        } else {
            merge_leaf();
            if (m_n > 1) fetch_tile<KCD>(s1, p0_tile1, lane);  // in flight through the barrier, the row store and the forming of the child
        }
        __syncthreads();
        MZ_STAMP(3)  // merge (leaf)
        // the leaf's merge row is all that is kept of its children's latents
        if ((D & 3) == 0) {
            const int d4 = D >> 2;
            for (int i = tid; i < trees * d4; i += kThreads) {
                const int t = i / d4, c = (i - t * d4) * 4;
                *reinterpret_cast<float4 *>(q.z_pool + ((size_t)(g0 + t) * (S + 1) + e) * D + c) =
                    __builtin_bit_cast(float4, lds4(Z + t * ld_d + c));
            }
        } else {
            for (int i = tid; i < trees * D; i += kThreads) {
                const int t = i / D, c = i - t * D;
                q.z_pool[((size_t)(g0 + t) * (S + 1) + e) * D + c] = Z[t * ld_d + c];
            }
        }
        if (sim) {
            for (int i = tid; i < KCD * 64; i += kThreads) form_item(i, Z, -1, s_act, XC);
            __syncthreads();
            MZ_STAMP(16)  // child latent
        }
        if constexpr (FULL) {
            // slot x: policy 1's first tile; slot 1: its second; slot 0: the merge tile the leaf used.  Policy 2's
            // tiles are requested as slots 1 and 0 fall free
            work_policy1(sx, p0_first);
            forget(sx);
            if (sim) work_merge(s0, XC, wave, Zs);
            fetch_tile<KCP>(s0, p1_tile0, lane);
            work_policy1(s1, p0_first + 1);
            fetch_tile<KCP>(s1, p1_tile1, lane);
            if (sim && m_n > 1) work_merge_resident(XC, Zs);
        } else {
            policy1_and_child_merge(sim);
        }
        __syncthreads();
        MZ_STAMP(18)  // policy 1 (leaf + child) | merge (child)

        // ---- policy 2 of leaf and child: one pass over two row tiles ---------------------------------
        {
            auto work = [&](Slot<KMAX> &s, int j) {
                const int t = p1_first + j;
                f32x4 hi[2], lo[2];
                mma_tile<KCP, 2>(s, H1 + lane * 4, 16 * ld_p, hi, lo);
                epi_split<2, true>(hi, lo, t, lds4(Bp1 + t * 16 + 4 * (lane >> 4)), H2, 16 * ld_q, KCP, np1, lane);
            };
            auto ptr = [&](int j) { return tile_of(Lp1, KCP, min(p1_first + j, np1 - 1)); };
            // policy 3 is one column tile: the first wave multiplies it, and requests it as soon as its slot 0 is free
            if constexpr (FULL) {
                // (in the order the tiles were requested)
                work(s0, 0);
                if (wave == 0) fetch_tile<KCP>(s0, tile_of(Lp2, KCP, 0), lane);
                work(s1, 1);
            } else {
                run_step(std::integral_constant<int, KCP>{}, OnePair{}, p1_n, ptr, work, [&] {
                    if (wave == 0) fetch_tile<KCP>(s0, tile_of(Lp2, KCP, 0), lane);
                }, nothing);
            }
            // value 1's first tiles: in flight through policy 3, the softmax phase and the forming of the rollout latent
        }
        __syncthreads();
        MZ_STAMP(4)  // policy 2 (leaf + child)

        // ---- policy 3 (narrow), softmax; priors of the leaf, greedy action of the child ----------------------
        {
            // value 1's first tiles: waves 1-3 request them beside policy 3 (they idle while the first wave multiplies;
            // more waves would hold up the barrier behind it), waves 4-7 beside the softmax phase, the first wave after it
            if (sim && wave >= 1 && wave < 4) prefetch2(v0_tile0, v0_tile1, KCD);
            if (wave == 0) {  // policy 3 of the 32 rows -> logits (fp32 rows; the layer's own softmax follows)
                f32x4 hi[2], lo[2];
                mma_tile<KCP, 2>(s0, H2 + lane * 4, 16 * ld_q, hi, lo);
                epi_rows<2, false>(hi, lo, 0, lds4(Bp2 + 4 * (lane >> 4)), Lg, ld_l, lane);
            }
            __syncthreads();
            MZ_STAMP(19)  // policy 3
            if (sim && wave >= 4) prefetch2(v0_tile0, v0_tile1, KCD);
            // softmax_rows' arithmetic for A <= 8 (row maximum; sum over the columns in ascending order) with eight lanes
            // per row, all 32 rows at once, the lanes of a row exchanging values through DPP moves instead of LDS round
            // trips: the network's own softmax; then the first maximum of the child's policy
            // (src/search/simulation.py:36) and expand_node's extra softmax for the leaf (src/search/expansion.py:27)
            if (tid < 2 * kTrees * 8) {
                const int row = tid >> 3, jx = tid & 7;
                const bool live = jx < A;
                auto softmax8 = [&](float v) {  // v: this lane's column (dead lanes: any value)
                    float mx = live ? v : -INFINITY;
                    mx = fmaxf(mx, dpp_f32<0xB1>(mx));    // quad_perm [1,0,3,2]
                    mx = fmaxf(mx, dpp_f32<0x4E>(mx));    // quad_perm [2,3,0,1]
                    mx = fmaxf(mx, dpp_f32<0x141>(mx));   // row_half_mirror: all eight lanes hold the row maximum
                    const float ev = live ? expf(v - mx) : 0.0f;
                    float sum = ev;                       // lane k ends with e0 + e1 + ... + ek, added in that order
#pragma unroll
                    for (int k = 1; k < 8; ++k) {
                        const float below = dpp_f32<0x111>(sum);  // row_shr:1
                        if (jx == k) sum = below + ev;
                    }
                    const float top = dpp_f32<0xFF>(sum);         // quad_perm [3,3,3,3]: lanes 4-7 hold lane 7's total
                    const float mir = dpp_f32<0x141>(top);        // lanes 0-3 take it from lanes 7-4
                    return ev / (jx < 4 ? mir : top);
                };
                float v = live ? Lg[row * ld_l + jx] : 0.0f;
                v = softmax8(v);
                if (row >= kTrees) {
                    float mx = live ? v : -INFINITY;
                    mx = fmaxf(mx, dpp_f32<0xB1>(mx));
                    mx = fmaxf(mx, dpp_f32<0x4E>(mx));
                    mx = fmaxf(mx, dpp_f32<0x141>(mx));
                    const unsigned long long hit = __ballot(live && v == mx);
                    const unsigned bits = (unsigned)(hit >> (lane & ~7)) & 0xffu;
                    if (jx == 0) s_act2[row - kTrees] = bits ? __ffs(bits) - 1 : 0;  // the first maximum
                } else {
                    v = softmax8(v);
                    if (live) {
                        const int t = row;
                        s_prior[t * A + jx] = v;
                        if (q.prior_trace && t < trees) q.prior_trace[((size_t)(g0 + t) * (S + 1) + e) * A + jx] = v;
                    }
                }
            }
            __syncthreads();
        }
        MZ_STAMP(5)  // policy 3, softmax, priors, greedy action
        // ---- child records ---------------------------------------------------------------------------
        // (by the last two waves, which have no share in the forming of the rollout latent that follows: nothing reads
        // these records before the next round's selection, and the backup only writes to them)
        for (int i = tid - (kThreads - 128); i >= 0 && i < trees * A; i += 128) {
            const int t = i / A, a = i - t * A;
            const size_t o = (size_t)(g0 + t) * p.cap;
            float pr = s_prior[t * A + a];
            if (e == 0) pr = mzt::noisy_prior((double)pr, q.eta[(size_t)(g0 + t) * A + a], q.noise_frac);
            // (the reward follows after the rounds; the statistics of the drawn child come from the backup, which
            // knows what a new record holds, unless the path is longer than the backup lanes cover)
            if (sim && a == s_act[t] && s_depth[t] + 1 <= kPathCap - 1)
                mzt::store_edge(p, o + base + a, mzt::NodeEdge{(double)pr, 0.0f, -1});
            else
                mzt::write_child(p, o, base, a, s_leaf[t], pr, 0.0f, (int8_t)s_leaf_tp[t]);
            if (a == 0) p.node[o + s_leaf[t]].first_child = base;
        }
        if (sim) {
            if (wave == 0) prefetch2(v0_tile0, v0_tile1, KCD);
            // ---- rollout latent ---------------------------------------------------------------------------
            for (int i = tid; i < KCD * 64; i += kThreads) form_item(i, Zs, -1, s_act2, XI);
            __syncthreads();
            MZ_STAMP(6)  // rollout latent
            // ---- value 1 of the rollout latent -> Hv (split image) | reward 1 of the rollout latent ---------------
            if constexpr (FULL) {
                // the third slot takes reward 1's tile at once, slot 0 value 2's tile behind the first product
                forget(sx);
                fetch_tile<KCD>(sx, r0_tile, lane);
                work_value1(s0, v0_first);
                fetch_tile<KCV>(s0, v1_tile, lane);
                work_value1(s1, v0_first + 1);
                work_reward1(sx);
                forget(sx);
            } else {
                run_step(std::integral_constant<int, KCD>{}, OnePair{}, v0_n, [&](int j) { return tile_of(Lv0, KCD, min(v0_first + j, nv0 - 1)); },
                         [&](Slot<KMAX> &s, int j) { work_value1(s, v0_first + j); },
                         [&] { fetch_tile<KCD>(s0, r0_tile, lane); },  // reward 1's tile: beside this wave's second tile
                         nothing);
                fetch_tile<KCV>(s1, v1_tile, lane);  // value 2's tile: in flight through reward 1 and the barrier
                // reward 1 of the rollout latent (its input is the one value 1 has just read)
                if (wave < nr0) work_reward1(s0);
            }
            __syncthreads();
            MZ_STAMP(7)  // value 1 | reward 1
            // ---- value 2 ------------------------------------------------------------------------------------
            if constexpr (FULL) work_value2(s0);
            else if (wave < nv1) work_value2(s1);
            __syncthreads();
            MZ_STAMP(8)  // value 2
            if (tid < kTrees) {  // value of the rollout latent (as finish_dot)
                const float sum = FULL ? ordered_sum_n<4 * kWaves>(DotV + tid, kTrees) : ordered_sum(DotV + tid, kTrees, 4 * nv1);
                s_roll[tid * 2 + 1] = activate(sum + Bv1[Lv1.Np], Lv2.act);
            } else if (tid >= 64 && tid < 64 + kTrees) {
                const int t = tid - 64;
                const float sum = FULL ? ordered_sum_n<4 * kWaves>(DotE + t, kTrees) : ordered_sum(DotE + t, kTrees, 4 * nr0);
                s_roll[t * 2] = activate(sum + Br0[Lr0.Np], Lr1.act);
            }
        }

        __syncthreads();  // s_roll is complete (and the records of a root expansion exist before the first selection)
        // The two selecting waves request the next round's first tiles HERE, where they idle beside the backup: issued
        // in the leaf-latent phase, their 72 KB stood in the port's queue in front of the other waves' loads of the
        // leaf's rows.  (Their first record reads of the selection return behind these tiles: the backup covers most of it.)
        forget(s0), forget(s1);  // (every slot is free here)
        if constexpr (FULL) forget(sx);
        if (wave < 2 && e < S) prefetch_merge();
        if (!sim) continue;
        // ---- backup: the recorded path is updated by independent lanes (see k_search) --------------------------
        if (tid >= kThreads / 2 && tid < kThreads / 2 + kTrees * kPathCap) {
            const int t = (tid - kThreads / 2) / kPathCap, i = (tid - kThreads / 2) % kPathCap;
            if (t < trees) {
                const size_t o = (size_t)(g0 + t) * p.cap;
                float roll[2] = {s_roll[t * 2], s_roll[t * 2 + 1]};
                const int dl = s_depth[t];
                const double ret = mzt::discounted_return(roll, 2, q.discount);
                if (i == 0 && q.rollout_trace) {
                    q.rollout_trace[((size_t)(g0 + t) * S + (e - 1)) * 2] = roll[0];
                    q.rollout_trace[((size_t)(g0 + t) * S + (e - 1)) * 2 + 1] = roll[1];
                }
                if (dl + 1 <= kPathCap - 1) {
                    if (i <= dl + 1) {
                        const int node = (i == 0) ? s_child[t] : s_path[t * kPathCap + (dl - (i - 1))];
                        const bool same = s_root[t] == 0 || (i >= 1 && ((i - 1) & 1) == 0);
                        const size_t c = o + node;
                        mzt::NodeStat st = path_stat;
                        if (i == 0) st = mzt::NodeStat{0.0, 0, MZ_LINK(s_leaf[t], (int8_t)(-s_leaf_tp[t]))};  // a new record (write_child)
                        st.visit += 1;
                        const double add = same ? ret : -ret;
                        st.value_sum = st.value_sum + add;
                        mzt::store_stat(p, c, st);
                    }
                } else if (i == 0) {
                    mzt::backup_path(p, o, s_child[t], (int8_t)s_leaf_tp[t], ret);
                }
            }
        }
        __syncthreads();
        MZ_STAMP(11)  // child records, backup
    }

    // ---- rewards of every expansion's children (src/search/expansion.py:29-33) --------------------------------
    // Row-stationary and barrier-free: reward 1's weights (all column tiles) are copied into LDS once; a wave takes
    // pairs of children of one expansion, reads its lanes' share of the expansion's merge row from z_pool (requested
    // while the previous pair is multiplied), forms the operand of the pair IN REGISTERS (a lane's eight k of row
    // lane & 15 are exactly its piece of the B operand: no LDS round trip for the activations), multiplies it with
    // every column tile, and folds
    // the one-output layer as it goes: the four quarter sums of a tile sit in the lanes 16 apart and are added in
    // ascending order through lane exchanges -- finish_dot's order.  Same per-accumulator instruction sequence and
    // epilogue arithmetic as the rounds.
    {
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        lds_f *EW = to_lds(smem_raw + L::ew);
        const float b2 = Br0[Lr0.Np];
        __syncthreads();  // the rounds' buffers are dead from here on
        {
            const glb_f4 *src = (const glb_f4 *)(blob + Lr0.w_off);
            for (int i = tid; i < nr0 * KCD * 128; i += kThreads) *reinterpret_cast<lds_f4 *>(EW + i * 4) = src[i];
        }
        __syncthreads();
        // Work items: up to four children of one expansion, multiplied as pairs (2 p, 2 p + 1) -- a last single child
        // on its own, at half the cost of a pair -- UP = ceil(A / 4) items per expansion.  Items are dealt in blocks of
        // eight, a wave's position in the block rotating from block to block, so that every wave gets its share of the
        // cheaper items (with seven actions: four children, then three).
        const int UP = (A + 3) >> 2, items = (S + 1) * UP;
        const int j = lane & 15, rq = lane >> 4;
        // this lane's part of an expansion's merge row: k = 32 c + 8 rq .. + 7 of tree j
        f32x4 z0[KCD], z1[KCD];
        auto load_z = [&](int ex) {
#pragma unroll
            for (int c = 0; c < KCD; ++c) {
                const int k0 = 32 * c + 8 * rq;
                z0[c] = z1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j < trees) {
                    const float *zr = q.z_pool + ((size_t)(g0 + j) * (S + 1) + ex) * D + k0;
                    if ((D & 3) == 0) {
                        if (k0 < D) z0[c] = *(const glb_f4 *)(zr);
                        if (k0 + 4 < D) z1[c] = *(const glb_f4 *)(zr + 4);
                    } else {
                        float t8[8];
                        for (int i = 0; i < 8; ++i) t8[i] = k0 + i < D ? zr[i] : 0.0f;
                        z0[c] = f32x4{t8[0], t8[1], t8[2], t8[3]};
                        z1[c] = f32x4{t8[4], t8[5], t8[6], t8[7]};
                    }
                }
            }
        };
        auto item_of = [&](int blk) { return blk * kWaves + ((wave - blk) & (kWaves - 1)); };
        if (item_of(0) < items) load_z(item_of(0) / UP);
        for (int blk = 0;; ++blk) {
            const int v = item_of(blk);
            if (v >= items) break;
            const int ex = v / UP, a0 = (v - ex * UP) * 4;
            auto pass = [&](auto mt_tag, int a_first, bool last) {
                constexpr int MT = decltype(mt_tag)::value;
                h8 xh[MT][KCD], xl[MT][KCD];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int a = a_first + mt;
#pragma unroll
                    for (int c = 0; c < KCD; ++c) {
                        const int k0 = 32 * c + 8 * rq;
                        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                        if (k0 < Dp16) {
                            const lds_f *tr = Tab + a * ld_d + k0;
                            const f32x4 b0 = lds4(tr), b1 = lds4(tr + 4);
                            v0 = f32x4{fmaxf(z0[c].x + b0.x, 0.f), fmaxf(z0[c].y + b0.y, 0.f), fmaxf(z0[c].z + b0.z, 0.f), fmaxf(z0[c].w + b0.w, 0.f)};
                            v1 = f32x4{fmaxf(z1[c].x + b1.x, 0.f), fmaxf(z1[c].y + b1.y, 0.f), fmaxf(z1[c].z + b1.z, 0.f), fmaxf(z1[c].w + b1.w, 0.f)};
                        }
                        if (STORE && j < trees && k0 < D) {
                            // every node's latent stored (Node.latent_state, src/search/expansion.py:30-33): this lane's eight values
                            float *dst = p.latent + ((size_t)(g0 + j) * p.cap + 1 + ex * A + a) * D + k0;
                            if ((D & 3) == 0) {
                                *(f32x4 *)dst = v0;
                                if (k0 + 4 < D) *(f32x4 *)(dst + 4) = v1;
                            } else {
                                const float t8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                                for (int i = 0; i < 8; ++i)
                                    if (k0 + i < D) dst[i] = t8[i];
                            }
                        }
                        f32x4 hi, lo;
                        split8(v0, v1, hi, lo);
                        xh[mt][c] = __builtin_bit_cast(h8, hi);
                        xl[mt][c] = __builtin_bit_cast(h8, lo);
                    }
                }
                // the item's last operands are formed: the next item's merge row can be on its way during the products
                const int vn = item_of(blk + 1);
                if (last && vn < items) load_z(vn / UP);
                float sum[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sum[mt] = 0.0f;
                for (int nt = 0; nt < nr0; ++nt) {
                    f32x4 hi[MT], lo[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) hi[mt] = lo[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const lds_f *wt = EW + (size_t)nt * KCD * 512 + lane * 4;
#pragma unroll
                    for (int c = 0; c < KCD; ++c) {
                        const h8 wh = __builtin_bit_cast(h8, lds4(wt + (2 * c) * 256)), wl = __builtin_bit_cast(h8, lds4(wt + (2 * c + 1) * 256));
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            hi[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[mt][c], hi[mt], 0, 0, 0);
                            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[mt][c], lo[mt], 0, 0, 0);
                            lo[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[mt][c], lo[mt], 0, 0, 0);
                        }
                    }
                    const f32x4 bb = lds4(Br0 + nt * 16 + 4 * rq), w2 = lds4(DwR + nt * 16 + 4 * rq);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 v = epi_value<true>(hi[mt], lo[mt], bb);
                        float part = __fmul_rn(v.x, w2.x);
                        part += __fmul_rn(v.y, w2.y);
                        part += __fmul_rn(v.z, w2.z);
                        part += __fmul_rn(v.w, w2.w);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) sum[mt] += __shfl(part, j + 16 * qq, 64);  // quarters in ascending order
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (rq == 0 && j < trees)
                        p.node[(size_t)(g0 + j) * p.cap + 1 + ex * A + a_first + mt].reward = activate(sum[mt] + b2, Lr1.act);
            };
            for (int a_first = a0; a_first < min(a0 + 4, A); a_first += 2) {
                const bool last = a_first + 2 >= min(a0 + 4, A);
                if (a_first + 1 < A) pass(std::integral_constant<int, 2>{}, a_first, last);
                else pass(std::integral_constant<int, 1>{}, a_first, last);
            }
        }
        MZ_STAMP(9)  // the children's rewards
    }

    // ---- result -------------------------------------------------------------------------------
    if (tid < trees) {
        p.n_nodes[g0 + tid] = 1 + (S + 1) * A;
        mzt::tree_result(p, g0 + tid, policy, policy32, value, visits);
    }
    if (stamping) {
        __syncthreads();
        if (threadIdx.x < 32) q.stamps[threadIdx.x] = s_stamp[threadIdx.x];
    }
#undef MZ_STAMP
}

template <int KCD, int KCP, int KCV, bool STORE, bool FULL>
int launch_seq(const mz_pool *pool, const mz_packed_nets *plan, const float *blob, const mz_search_params &q, size_t lds_total,
               double *policy, float *policy32, double *value, int32_t *visits, hipStream_t st) {
    auto kernel = k_search_seq<KCD, KCP, KCV, STORE, FULL>;
    if (lds_total > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total);
        if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "k_search_seq: cannot enable %zu bytes of LDS: %s", lds_total, hipGetErrorString(e));
    }
    const int blocks = (pool->G + kTrees - 1) / kTrees;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kThreads), lds_total, st, *pool, *plan, blob, q, policy, policy32, value, visits);
    return mz::check_launch("k_search_seq");
}

// The written-out schedule with a third ring slot (FULL) is taken wherever the layer widths allow it; the knob
// (environment variable at load time, mz_search_seq_three_slots at run time) lets the general schedule run on the
// same searches.
int g_three_slots = [] {
    const char *e = getenv("MZ_SEARCH_SEQ_THREE_SLOTS");
    return e ? atoi(e) : 1;
}();

}  // namespace

extern "C" int mz_search_seq_three_slots(int32_t new_value) {
    const int old = g_three_slots;
    if (new_value >= 0) g_three_slots = new_value;
    return old;
}

// The sequential-program form of mz_search_run for the searches it covers: returns 1 after launching it,
// 0 when the search needs k_search, a negative code on a failed launch.  Arguments are validated by the caller.
int mz_search_seq_try(const mz_pool *pool, const mz_packed_nets *plan, const float *blob, const mz_search_params *prm,
                      double *policy, float *policy32, double *value, int32_t *visits, void *stream) {
    static const bool off = [] { const char *e = getenv("MZ_SEARCH_SEQ"); return e && atoi(e) == 0; }();
    if (off) return 0;
    const mz_packed_nets &P = *plan;
    if (prm->z_pool == nullptr || prm->depth != 1 || P.A > 32) return 0;  // (z_pool: the expansions' merge rows, also when the latents are stored)
    if (P.policy.n_layers != 3 || P.value.n_layers != 3 || P.reward.n_layers != 2) return 0;
    if (!is_matrix_layer(P.policy.layer[0]) || !is_matrix_layer(P.policy.layer[1]) || !is_matrix_layer(P.policy.layer[2]) ||
        P.policy.layer[2].act != MZ_ACT_SOFTMAX || P.policy.layer[2].N != P.A || P.A > 8 || P.policy.layer[2].Kp != P.policy.layer[1].Kp)
        return 0;
    if (!is_matrix_layer(P.value.layer[0]) || !tail_foldable(P.value) || !is_matrix_layer(P.reward.layer[0]) || !tail_foldable(P.reward))
        return 0;
    if (P.policy.layer[0].act != MZ_ACT_RELU || P.policy.layer[1].act != MZ_ACT_RELU || P.value.layer[0].act != MZ_ACT_RELU ||
        P.value.layer[1].act != MZ_ACT_RELU || P.reward.layer[0].act != MZ_ACT_RELU)
        return 0;
    if ((P.value.layer[1].Np >> 4) > kWaves || (P.reward.layer[0].Np >> 4) > kWaves) return 0;
    if ((P.policy.layer[1].Np >> 4) > 2 * kWaves || (P.value.layer[0].Np >> 4) > 2 * kWaves) return 0;
    if ((P.policy.layer[0].Np >> 4) > 2 * kWaves || (P.merge.layer[0].Np >> 4) > 2 * kWaves) return 0;
    const int kcd = P.merge.layer[0].Kp >> 5, kcp = P.policy.layer[1].Kp >> 5, kcv = P.value.layer[1].Kp >> 5;
    if (P.policy.layer[0].Kp != P.merge.layer[0].Kp || P.value.layer[0].Kp != P.merge.layer[0].Kp ||
        P.reward.layer[0].Kp != P.merge.layer[0].Kp)
        return 0;
    // the strides of the compile-time layout are those of the packed layers (a layer's padded input width is the
    // padded output width of the one before it)
    if (lds_stride(P.D) != kcd * 32 + 4 || lds_stride(P.policy.layer[0].N) != kcp * 32 + 4 || lds_stride(P.policy.layer[1].N) != kcp * 32 + 4 ||
        lds_stride(P.value.layer[0].N) != kcv * 32 + 4 || P.A > kSeqMaxA || prm->tab_len < 0)
        return 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool store = pool->latent != nullptr;
    // every wave with two tiles in policy 1, policy 2 and value 1, one in value 2 and reward 1, one or two in the merge
    const bool full = g_three_slots && (P.policy.layer[0].Np >> 4) == 2 * kWaves && (P.policy.layer[1].Np >> 4) == 2 * kWaves &&
                      (P.value.layer[0].Np >> 4) == 2 * kWaves && (P.value.layer[1].Np >> 4) == kWaves &&
                      (P.reward.layer[0].Np >> 4) == kWaves && (P.merge.layer[0].Np >> 4) >= kWaves &&
                      (P.merge.layer[0].Np >> 4) <= kWaves + 3;  // (SeqLayout::res_tiles resident merge tiles)
#define MZ_SEQ_LAUNCH(A_, B_, C_, F_)                                                                                          \
    {                                                                                                                          \
        const size_t lds_total = SeqLayout<A_, B_, C_>::total(prm->tab_len);                                                   \
        if (lds_total > 160 * 1024) return 0;                                                                                  \
        rc = store ? launch_seq<A_, B_, C_, true, F_>(pool, plan, blob, *prm, lds_total, policy, policy32, value, visits, st)  \
                   : launch_seq<A_, B_, C_, false, F_>(pool, plan, blob, *prm, lds_total, policy, policy32, value, visits, st);\
    }
    if (kcd == 6 && kcp == 8 && kcv == 8 && full) MZ_SEQ_LAUNCH(6, 8, 8, true)
    else if (kcd == 6 && kcp == 8 && kcv == 8) MZ_SEQ_LAUNCH(6, 8, 8, false)
    else if (kcd == 2 && kcp == 4 && kcv == 4) MZ_SEQ_LAUNCH(2, 4, 4, false)
    else return 0;
#undef MZ_SEQ_LAUNCH
    return rc == MZ_OK ? 1 : rc;
}
