// Convolutions of the representation network at the fp16 matrix rate with float32 accuracy (gfx950).
//
// The fp32 matrix instruction of gfx950 runs at 1/16 of the fp16 rate.  A float32 value is held as two fp16 terms,
//     hi = fp16(v),   lo = fp16((v - hi) * 2^11)          (v = hi + lo * 2^-11 to 2^-22 |v|; the scale keeps lo out of
//                                                          fp16's subnormal range),
// and a product of two such values is, up to the lo * lo term of relative size 2^-22,
//     x w = x_hi w_hi + 2^-11 (x_hi w_lo + x_lo w_hi).
// So that ONE float32 accumulator takes all three terms, the main term uses w_hs = w_hi * 2^11 (an exact exponent
// shift, formed from the w_hi operand in registers: four packed multiplications per operand):
//     2^11 x w = x_hi w_hs + x_hi w_lo + x_lo w_hi,
// three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block into one accumulator, scaled back by 2^-11 (exact) in the
// epilogue -- float32 accuracy at 16 / 3 = 5.3 times the fp32 matrix rate, with the register budget of a plain
// product.  Operands must stay below 65504 in magnitude (weights below 32): beyond that the conversion gives
// infinity and the results are NaN -- loud, not silently wrong.
// Split operands in memory are two planes of 16-bit values [plane][rows][C] (channels last): plane 0 = hi, plane 1 = lo.
//
//   k_conv3        implicit-GEMM convolution: 256 output pixels x 128 output channels per workgroup
//                  (8 waves, 64 x 64 per wave), reduction over (tap, input channel) in steps of 32,
//                  A rows gathered through the pixel index table of mz_conv_index, both operands
//                  staged through a double-buffered, XOR-swizzled LDS image (conflict-free b128
//                  reads: see lds_off), one barrier per step; channel chunks outermost so that the taps of a
//                  chunk find their pixels in cache
//   k_split3       float32 rows -> the planes (weights at pack time, network input)
//   k_bn_apply3    BatchNorm scale/shift + residual + activation of a ResBlock, reading the raw
//                  float32 convolution output, writing the float32 and / or split activation of the next layer
//   k_unsplit3     planes -> float32 (for the few consumers outside this path; 22 significant bits)
#include <algorithm>
#include <type_traits>
#include <utility>

#include "mz_common.h"

// Diagnostic builds (-DMZ_CONV3_SKIP=mask, tools/diag_conv3.sh) leave parts of k_conv3_same out to
// see what bounds it: 1 matrix instructions, 8 weight staging, 16 restaging of the pixel range after the
// first chunk, 32 clock stamps.  Results are wrong in those builds.
#ifndef MZ_CONV3_SKIP
#define MZ_CONV3_SKIP 0
#endif

namespace {

using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kT3 = 512;          // threads: 8 waves
constexpr int kTM = 256, kTN = 128, kTK = 32;
constexpr int kARow = 64;         // bytes of one LDS row: 32 fp16
constexpr int kPA = 2, kPW = 2;   // planes of an activation operand and of a weight operand (hi, lo)
constexpr int kAStage = kPA * kTM * kARow, kBStage = kPW * kTN * kARow;
constexpr int kStage = kAStage + kBStage;   // 57,344 bytes; two stages
constexpr float kLoScale = 2048.0f, kLoUnscale = 1.0f / 2048.0f;   // 2^11: scale of the lo and hs planes

__device__ __forceinline__ float act3(float v, int act) {
    switch (act) {
        case MZ_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case MZ_ACT_TANH: return tanhf(v);
        case MZ_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

// f(integral_constant<int, K>) for every K of the sequence, in order
template <int... K, class F>
__device__ __forceinline__ void for_each_index(std::integer_sequence<int, K...>, F &&f) {
    (f(std::integral_constant<int, K>{}), ...);
}

__device__ __forceinline__ unsigned short f16_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float f16_value(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// x -> (hi, lo): x = hi + lo * 2^-11 to 2^-22 |x|.  Conversions round to nearest even; the subtraction is exact.
__device__ __forceinline__ void split2(float x, unsigned short &hi, unsigned short &lo) {
    const _Float16 h = (_Float16)x;
    hi = f16_bits(h);
    lo = f16_bits((_Float16)((x - (float)h) * kLoScale));
}
// w_hi * 2^11 of a weight operand (exact unless it overflows: |w| >= 32)
__device__ __forceinline__ h8 scaled_hi(const h8 &hi) { return hi * (_Float16)kLoScale; }
// the value two planes hold
__device__ __forceinline__ float unsplit2(unsigned short hi, unsigned short lo) { return f16_value(hi) + f16_value(lo) * kLoUnscale; }

// byte offset of the 16-byte piece `pc` (8 reduction indices) of row `r` inside one plane image.
// A ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
// the same plus 32 -- each of which must touch 16 different 16-byte columns of the 256-byte bank row.
// With lane = 16 * piece + row, a group reads piece p of rows 0-3 and 12-15 and piece p ^ 1 of rows
// 4-11; a row is 64 bytes, so rows r, r+4, r+8, r+12 share their columns and XOR-ing bit 1 of the
// piece with bit 2 of the row separates them -- for any sixteen consecutive rows, wherever they start.
__device__ __forceinline__ int lds_off(int r, int pc) { return r * kARow + ((pc ^ ((r >> 1) & 2)) << 4); }

// Output tile of a workgroup (256 rows x 128 columns from row m0, column n0; wave (wm, wn) owns the
// 64 x 64 block at (64 wm, 64 wn) and its lane (m, kg) holds rows 16 mt + 4 kg + j of columns
// 16 nt + m): bias, activation, then float32 and / or split stores.  The accumulator layout gives a
// lane one column of four rows, which would store 64-byte pieces; the tile goes through LDS (free once
// the reduction is over; `tile` needs kOutLds bytes) and leaves as whole 512-byte rows.
// `sums` (optional, f64[rows / 64][N][2]): the sum and the sum of squares of the stored values of
// every column over each wave's 64 rows, added in a fixed order in float64 -- the per-game BatchNorm
// statistics are sums of whole blocks when a game's pixel count is a multiple of 64.
// Every wave of the workgroup must call it, after a barrier that ends the last use of LDS.
constexpr int kOutLd = kTN + 4;                    // floats per staged row: 16-byte aligned, rows 4 banks apart
constexpr int kOutLds = kTM * kOutLd * 4;          // 135,168 bytes

__device__ __forceinline__ void conv3_epilogue(const f32x4 (&acc)[4][4], unsigned char *lds, int m0, int n0, int wm, int wn, int m,
                                               int kg, const float *__restrict__ bias, int act, float *__restrict__ y,
                                               unsigned short *__restrict__ y3, long long y_plane, int rows, int N,
                                               double *__restrict__ sums) {
    float *tile = reinterpret_cast<float *>(lds);
    const int row0 = m0 + wm * 64;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int lc = wn * 64 + nt * 16 + m;
        const float bv = bias ? bias[n0 + lc] : 0.0f;
        double cs = 0.0, cq = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr = wm * 64 + mt * 16 + kg * 4 + j;
                const float v = act3(acc[mt][nt][j] * kLoUnscale + bv, act);
                tile[lr * kOutLd + lc] = v;
                if (m0 + lr < rows) {
                    cs += (double)v;
                    cq += (double)v * (double)v;
                }
            }
        if (sums) {  // wave-uniform
            cs += __shfl_xor(cs, 16, 64);
            cq += __shfl_xor(cq, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            cq += __shfl_xor(cq, 32, 64);
            if (kg == 0 && row0 < rows) {
                double *dst = sums + ((long long)(row0 >> 6) * N + n0 + lc) * 2;
                dst[0] = cs;
                dst[1] = cq;
            }
        }
    }
    __syncthreads();
    // 32 lanes x 4 columns = one row; a wave stores two whole rows per instruction
    const int c4 = (threadIdx.x & 31) * 4;
    for (int lr = threadIdx.x >> 5; lr < kTM; lr += kT3 / 32) {
        const int row = m0 + lr;
        if (row >= rows) break;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + lr * kOutLd + c4);
        const long long o = (long long)row * N + n0 + c4;
        if (y) *reinterpret_cast<f32x4 *>(y + o) = v;
        if (y3) {
            unsigned short p[kPA][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2(v[e], p[0][e], p[1][e]);
#pragma unroll
            for (int q = 0; q < kPA; ++q)
                *reinterpret_cast<uint2 *>(y3 + q * y_plane + o) =
                    uint2{(unsigned)p[q][0] | ((unsigned)p[q][1] << 16), (unsigned)p[q][2] | ((unsigned)p[q][3] << 16)};
        }
    }
}

// RAW: the input is float32 rows xf[pixel][C] (a row's 32-channel chunk is one whole 128-byte line instead of two
// 64-byte pieces); it is split while it is staged.
template <bool RAW>
__global__ void __launch_bounds__(kT3)
k_conv3(const unsigned short *__restrict__ x3, long long x_plane, const float *__restrict__ xf, int C, const int32_t *__restrict__ idx, int taps,
        const unsigned short *__restrict__ w3, long long w_plane, int K, const float *__restrict__ bias, int act,
        float *__restrict__ y, unsigned short *__restrict__ y3, long long y_plane, int rows, int N,
        double *__restrict__ sums) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
    const int chunks = C / kTK, steps = taps * chunks;
    // Reduction order: channel chunk outermost, taps innermost -- the taps of one chunk re-read nearly
    // the same pixels (shifted by a column or a row), so their gathers hit in L1 / L2 instead of
    // returning to each pixel once per tap with the whole channel sweep in between.
    // The pixel indices of this workgroup's 256 rows for every tap sit in LDS behind the two stages.
    int *s_idx = reinterpret_cast<int *>(lds + 2 * kStage);  // [taps][kTM]
    for (int i = tid; i < taps * kTM; i += kT3) {
        const int tap = i / kTM, r = m0 + (i - tap * kTM);
        s_idx[i] = r < rows ? idx[(long long)tap * rows + r] : -1;
    }
    __syncthreads();

    // staging roles: A rows ra and ra + 128 (three planes each), B column rb, piece pc
    const int ra = tid >> 2, pc = tid & 3, rb = tid >> 2;
    int srow[2];
    u32x4 ga[4], gb[kPW];
    auto fetch = [&](int s) {
        const int chunk = s / taps, tap = s - chunk * taps, c0 = chunk * kTK + pc * 8;
        srow[0] = s_idx[tap * kTM + ra];
        srow[1] = s_idx[tap * kTM + ra + 128];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (RAW) {
                const float *src = xf + (long long)max(srow[h], 0) * C + c0;
                ga[h * 2] = *reinterpret_cast<const u32x4 *>(src);
                ga[h * 2 + 1] = *reinterpret_cast<const u32x4 *>(src + 4);
            } else {
#pragma unroll
                for (int p = 0; p < kPA; ++p) {
                    const unsigned short *src = x3 + p * x_plane + (long long)max(srow[h], 0) * C + c0;
                    ga[h * kPA + p] = *reinterpret_cast<const u32x4 *>(src);
                }
            }
        }
        const int kk = tap * C + c0;
#pragma unroll
        for (int p = 0; p < kPW; ++p) gb[p] = *reinterpret_cast<const u32x4 *>(w3 + p * w_plane + (long long)(n0 + rb) * K + kk);
    };
    auto stash = [&](int stage) {
        unsigned char *base = lds + stage * kStage;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (RAW) {
                unsigned int pl[kPA][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 v4 = __builtin_bit_cast(f32x4, ga[h * 2 + q]);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        unsigned short ev[kPA], od[kPA];   // the planes of an even / odd channel
                        split2(v4[e], ev[0], ev[1]);
                        split2(v4[e + 1], od[0], od[1]);
#pragma unroll
                        for (int p = 0; p < kPA; ++p) pl[p][q * 2 + (e >> 1)] = srow[h] < 0 ? 0u : ((unsigned int)ev[p] | ((unsigned int)od[p] << 16));
                    }
                }
#pragma unroll
                for (int p = 0; p < kPA; ++p)
                    *reinterpret_cast<u32x4 *>(base + p * (kTM * kARow) + lds_off(ra + 128 * h, pc)) = u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
            } else {
#pragma unroll
                for (int p = 0; p < kPA; ++p) {
                    u32x4 v = ga[h * kPA + p];
                    if (srow[h] < 0) v = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(base + p * (kTM * kARow) + lds_off(ra + 128 * h, pc)) = v;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < kPW; ++p) *reinterpret_cast<u32x4 *>(base + kAStage + p * (kTN * kARow) + lds_off(rb, pc)) = gb[p];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const int stage = s & 1;
        const bool more = s + 1 < steps;
        if (more) fetch(s + 1);
        const unsigned char *sa = lds + stage * kStage, *sb = sa + kAStage;
        h8 a[kPA][4], b[kPW][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int p = 0; p < kPA; ++p) a[p][t] = *reinterpret_cast<const h8 *>(sa + p * (kTM * kARow) + lds_off(wm * 64 + t * 16 + m, kg));
#pragma unroll
            for (int p = 0; p < kPW; ++p) b[p][t] = *reinterpret_cast<const h8 *>(sb + p * (kTN * kARow) + lds_off(wn * 64 + t * 16 + m, kg));
        }
        // x_lo * w_hi, x_hi * w_hs, x_hi * w_lo; consecutive instructions never share an accumulator
#define MZ_TERM(P, Q)                                                                                        \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)       \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[P][mt], b[Q][nt], acc[mt][nt], 0, 0, 0);
        MZ_TERM(1, 0)
#pragma unroll
        for (int t = 0; t < 4; ++t) b[0][t] = scaled_hi(b[0][t]);
        MZ_TERM(0, 0)
        MZ_TERM(0, 1)
#undef MZ_TERM
        if (more) stash(stage ^ 1);
        __syncthreads();
    }

    conv3_epilogue(acc, lds, m0, n0, wm, wn, m, kg, bias, act, y, y3, y_plane, rows, N, sums);
}

// Stride-1 "same" convolutions (odd kernel, padding (k-1)/2): the input pixels the taps of a tile
// of 256 consecutive output pixels read form one contiguous range of the flat pixel index,
// [m0 - reach, m0 + 255 + reach] with reach = pad * (W + 1).  That range (the halo) of one 32-channel
// chunk is staged in LDS ONCE and the k*k taps of the chunk read their A operands from it through a
// per-(tap, row) slot table -- a padded tap names a slot of zeros -- instead of gathering every tap's
// rows from global memory again: the activation traffic into the compute unit drops by ~k*k / 1.4,
// which is what bounds k_conv3 on these layers.  Same reduction order and instruction sequence as
// k_conv3, so the two produce identical bits.
//
// The range is staged by groups of 128 slots (one per thread quartet): NJ = 3 groups hold images up to 63
// pixels wide; wider images (up to 144 pixels: 256 + 2 * 145 slots) take 4 or 5 groups, whose slots live in
// registers between their fetch and the end of the chunk; they are offered in the RAW form only (the form the
// representation network uses for every layer).
constexpr int kHaloSlots = 384;                       // NJ = 3: 256 rows + 2 * 64 pixels of reach
constexpr int kSlotBytes = 9 * kTM * 2;               // the slot table u16[9][256]
constexpr int kMaxLds = 160 * 1024;
__host__ __device__ constexpr int halo_plane(int cap) { return (cap + 4) * kARow; }  // + the slot of zeros (index cap)

// LDS of one launch: the planes of `cap` slots, three weight stages, the slot table, the scale / shift rows
__host__ __device__ constexpr size_t same_lds(int cap, int bn_games, int C) {
    return (size_t)kPA * halo_plane(cap) + 3 * kBStage + kSlotBytes + (size_t)2 * bn_games * C * sizeof(float);
}

//
// RAW: the input is the float32 output xf of the previous convolution, and the operand is
// in_act(xf * in_scale[game][c] + in_shift[game][c]) -- that layer's BatchNorm and activation --
// formed and split while the range is staged, so the normalised activation is never written to
// memory.  The scale / shift rows of the few games a range can touch sit in LDS behind the slot table.
// RES (with RAW): the operand is the OUTPUT OF A WHOLE ResBlock, in_act(xf * in_scale + in_shift + r) with r the block's
// skip path -- its input activation `res` (identity skip) or res * res_scale + res_shift (the raw 1x1 skip
// convolution with its BatchNorm), network_builder.py:138-145 -- formed with the roundings of k_bn_apply3; the
// workgroups of the first column tile also write it to act_out (float32) for their own 256 pixels: the next block's
// skip path and nothing else reads it, so the block output costs one write and no pass of its own.
template <bool RAW, int NJ, bool RES = false>
__global__ void __launch_bounds__(kT3)
k_conv3_same(const unsigned short *__restrict__ x3, long long x_plane, const float *__restrict__ xf,
             const float *__restrict__ in_scale, const float *__restrict__ in_shift, const float *__restrict__ res,
             const float *__restrict__ res_scale, const float *__restrict__ res_shift, float *__restrict__ act_out, int in_act,
             int bn_games, int cap, int C, int H,
             int W, int ksize, const unsigned short *__restrict__ w3, long long w_plane, int K,
             const float *__restrict__ bias, int act, float *__restrict__ y, unsigned short *__restrict__ y3,
             long long y_plane, int rows, int N, double *__restrict__ sums) {
    extern __shared__ __align__(16) unsigned char lds[];
    const long long t_start = (MZ_CONV3_SKIP & 32) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
    const int taps = ksize * ksize, pad = (ksize - 1) >> 1, reach = pad * (W + 1);
    const int chunks = C / kTK, steps = taps * chunks;
    const int kHaloPlane = halo_plane(cap);
    unsigned char *s_halo = lds, *s_b = lds + kPA * kHaloPlane;
    unsigned short *s_slot = reinterpret_cast<unsigned short *>(s_b + 3 * kBStage);  // [tap][wm][m][t]
    if (tid < kTM) {  // one row per thread: its image coordinates once, then the taps
        const int r = tid, row = m0 + r;
        const int w = row % W, h = (row / W) % H;
        unsigned short *dst = s_slot + ((r >> 6) * 16 + (r & 15)) * 4 + ((r >> 4) & 3);
        int tap = 0;
        for (int dy = -pad; dy <= pad; ++dy)
            for (int dx = -pad; dx <= pad; ++dx, ++tap) {
                const bool in = row < rows && h + dy >= 0 && h + dy < H && w + dx >= 0 && w + dx < W;
                dst[tap * (4 * 16 * 4)] = (unsigned short)lds_off(in ? r + reach + dy * W + dx : cap, 0);  // (below 64 Ki: cap <= 640)
            }
    }
    if (tid < 4 * kPA) *reinterpret_cast<u32x4 *>(s_halo + (tid >> 2) * kHaloPlane + cap * kARow + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};

    // staging roles: halo slots hs + 128 j, j < NJ (two planes each), B column rb, piece pc
    const int hs = tid >> 2, pc = tid & 3, rb = tid >> 2;
    const int nslots = kTM + 2 * reach;
    const int P = H * W;
    // RAW: scale rows then shift rows [bn_games][C] of the games from the one holding the first staged pixel on
    float *s_bn = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(s_slot) + kSlotBytes);
    const int g_first = max(m0 - reach, 0) / P;
    int gl[NJ] = {};
    if constexpr (RAW) {
        const int G = rows / P;
        for (int i = tid; in_scale && i < bn_games * C; i += kT3) {
            const int g = min(g_first + i / C, G - 1), c = i % C;
            s_bn[i] = in_scale[(long long)g * C + c];
            s_bn[bn_games * C + i] = in_shift[(long long)g * C + c];
            if (RES && res_scale) {
                s_bn[2 * bn_games * C + i] = res_scale[(long long)g * C + c];
                s_bn[3 * bn_games * C + i] = res_shift[(long long)g * C + c];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) gl[j] = min(max(m0 - reach + hs + 128 * j, 0) / P - g_first, bn_games - 1);
    }
    constexpr int kGH = 2 * NJ;   // RAW: two float32 pieces per slot, else one piece of each plane
    u32x4 gh[kGH], gb[kPW];
    u32x4 gr[RES ? 2 * NJ : 1];
    auto fetch_halo = [&](int chunk) {
        const int c0 = chunk * kTK + pc * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int slot = hs + 128 * j;
            const long long pix = (long long)m0 - reach + slot;
            const bool ok = slot < nslots && pix >= 0 && pix < rows;
            if constexpr (RAW) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    gh[j * 2 + q] = ok ? *reinterpret_cast<const u32x4 *>(xf + pix * C + c0 + 4 * q) : u32x4{0u, 0u, 0u, 0u};
                    if constexpr (RES) gr[j * 2 + q] = ok ? *reinterpret_cast<const u32x4 *>(res + pix * C + c0 + 4 * q) : u32x4{0u, 0u, 0u, 0u};
                }
            } else {
#pragma unroll
                for (int p = 0; p < kPA; ++p)
                    gh[j * kPA + p] = ok ? *reinterpret_cast<const u32x4 *>(x3 + p * x_plane + pix * C + c0) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    // RAW: a staged slot goes from its fetched float32 form to the operand planes in place -- the piece (j, q), four
    // channels of slot hs + 128 j, turns gh[2 j + q] into {two words of the hi plane, two words of the lo plane}.
    // (The input activation is a run-time value: a piece is instantiated for relu, the usual case, and for the
    // general switch, so that the common path is straight-line code.  Spreading the pieces between the matrix
    // instructions of a chunk's last two steps was measured: slower, 589 -> 708 us per 48 x 48 x 128 layer.)
    auto convert_piece_as = [&](int chunk, auto k_tag, auto relu_tag) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value, j = k >> 1, q = k & 1;
        constexpr bool kRelu = decltype(relu_tag)::value;
        auto in_activation = [&](float v) { return kRelu ? (v > 0.0f ? v : 0.0f) : act3(v, in_act); };
        const int c0 = chunk * kTK + pc * 8;
        const int slot = hs + 128 * j;
        const long long pix = (long long)m0 - reach + slot;
        const bool ok = slot < nslots && pix >= 0 && pix < rows;
        const float *sc = s_bn + gl[j] * C + c0, *sh = sc + bn_games * C;
        const f32x4 x4 = __builtin_bit_cast(f32x4, gh[k]);
        f32x4 v4 = x4;
        if constexpr (RES) {
            const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(sc + 4 * q), sh4 = *reinterpret_cast<const f32x4 *>(sh + 4 * q);
            const f32x4 r4 = __builtin_bit_cast(f32x4, gr[k]);
            f32x4 add = r4;
            if (res_scale) {  // uniform
                const float *rs = sc + 2 * bn_games * C, *rh = sc + 3 * bn_games * C;
                const f32x4 rs4 = *reinterpret_cast<const f32x4 *>(rs + 4 * q), rh4 = *reinterpret_cast<const f32x4 *>(rh + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) add[e] = fmaf(r4[e], rs4[e], rh4[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaf(x4[e], sc4[e], sh4[e]);  // the roundings of k_bn_apply3
                v += add[e];
                v4[e] = in_activation(v);
            }
            if (act_out && blockIdx.x == 0 && ok && slot >= reach && slot < reach + kTM)
                *reinterpret_cast<f32x4 *>(act_out + pix * C + c0 + 4 * q) = v4;
        } else if (in_scale) {  // uniform
            const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(sc + 4 * q), sh4 = *reinterpret_cast<const f32x4 *>(sh + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = in_activation(fmaf(x4[e], sc4[e], sh4[e]));
        }
        unsigned int pl[kPA][2];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            unsigned short ev[kPA], od[kPA];   // the planes of an even / odd channel
            split2(v4[e], ev[0], ev[1]);
            split2(v4[e + 1], od[0], od[1]);
#pragma unroll
            for (int p = 0; p < kPA; ++p) pl[p][e >> 1] = ok ? ((unsigned int)ev[p] | ((unsigned int)od[p] << 16)) : 0u;
        }
        gh[k] = u32x4{pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
    };
    auto convert_piece = [&](int chunk, auto k_tag) __attribute__((always_inline)) {
        if constexpr (RAW && decltype(k_tag)::value < 2 * NJ) {
            if (in_act == MZ_ACT_RELU) convert_piece_as(chunk, k_tag, std::true_type{});
            else convert_piece_as(chunk, k_tag, std::false_type{});
        }
    };
    // the converted slots (RAW) or the fetched planes -> the LDS image
    auto write_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int slot = hs + 128 * j;
            if (NJ > 3 && slot >= cap) continue;  // the image ends at `cap` slots (+ the slot of zeros)
            const u32x4 u = gh[j * 2], v = gh[j * 2 + 1];
            const u32x4 hi = RAW ? u32x4{u.x, u.y, v.x, v.y} : u, lo = RAW ? u32x4{u.z, u.w, v.z, v.w} : v;
            *reinterpret_cast<u32x4 *>(s_halo + lds_off(slot, pc)) = hi;
            *reinterpret_cast<u32x4 *>(s_halo + kHaloPlane + lds_off(slot, pc)) = lo;
        }
    };
    // the weights of consecutive steps, one call per step (tap innermost); the last step's are requested again
    // past the end (harmless: a step is one straight-line block)
    const unsigned short *wrow = w3 + (long long)(n0 + rb) * K + pc * 8;
    int f_tap = 0, f_chunk = 0;
    auto fetch_b_next = [&]() {
        const int kk = f_tap * C + f_chunk * kTK;  // (wave-uniform)
#pragma unroll
        for (int p = 0; p < kPW; ++p) gb[p] = *reinterpret_cast<const u32x4 *>(wrow + p * w_plane + kk);
        if (f_tap + 1 < taps) ++f_tap;
        else if (f_chunk + 1 < chunks) f_tap = 0, ++f_chunk;
    };
    const int b_off = lds_off(rb, pc);
    auto stash_b = [&](int stage) {
#pragma unroll
        for (int p = 0; p < kPW; ++p) *reinterpret_cast<u32x4 *>(s_b + stage * kBStage + p * (kTN * kARow) + b_off) = gb[p];
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch_halo(0);
    fetch_b_next();
    if constexpr (RAW) __syncthreads();  // the scale / shift rows are in place
    for_each_index(std::make_integer_sequence<int, 2 * NJ>{}, [&](auto k_tag) __attribute__((always_inline)) { convert_piece(0, k_tag); });
    write_halo();
    stash_b(0);
    fetch_b_next();
    stash_b(1);
    fetch_b_next();  // (the weights of step 2) stored by step 0
    __syncthreads();
    // The step loop is software-pipelined through the operand registers: every LDS read of step s + 1 is issued
    // during the matrix instructions of step s, into the register whose operand has just been multiplied for the
    // last time (x_lo[mt] after its four products of the first term, w_hi[nt] after the second term's, w_lo[nt] and
    // x_hi[mt] during the third), one read behind every four products: the reads, the few vector instructions and
    // the matrix instructions of a wave interleave instead of alternating in phases (both waves of a SIMD leave the
    // barrier in the same phase, so what one wave does not overlap itself is not overlapped at all), and a step's
    // products start right behind the barrier.  The weights therefore rotate through THREE stages: step s
    // multiplies stage s % 3 (read during step s - 1), reads stage (s + 1) % 3 ahead, and at its very start stores
    // the weights of step s + 2 -- requested from memory at the start of step s - 1, a whole step earlier, so the
    // store does not wait for them -- then requests those of step s + 3.  The step's barrier sits right behind that
    // store, where the wave must wait for its prefetched operands anyway.  The pixel range is constant over the
    // taps of a chunk, only a chunk's first step waits for its reads.  Reads and fetches past the last step / tap
    // are made anyway, from clamped addresses (a step is one straight-line block).
    // Terms: x_lo * w_hi, x_hi * w_hs, x_hi * w_lo (k_conv3 adds them in the same order).
    const long long t_loop = (MZ_CONV3_SKIP & 32) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    int bo[4], ao[4], an[4];  // LDS offsets of the weight operands, of the pixel operands read in this step, in the next
#pragma unroll
    for (int t = 0; t < 4; ++t) bo[t] = lds_off(wn * 64 + t * 16 + m, kg);
    // the table holds the byte offset of a slot's piece 0; piece kg of the same row is that XOR 16 kg (lds_off)
    auto a_offsets = [&](int tap, int (&dst)[4]) {
        const uint2 sl = *reinterpret_cast<const uint2 *>(s_slot + ((tap * 4 + wm) * 16 + m) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[t] = ((int)((t < 2 ? sl.x : sl.y) >> (16 * (t & 1))) & 0xffff) ^ (kg << 4);
    };
    auto read_a = [&](int plane, int t) { return *reinterpret_cast<const h8 *>(s_halo + plane * kHaloPlane + ao[t]); };
    auto read_b = [&](int plane, int stage, int t) {
        return *reinterpret_cast<const h8 *>(s_b + stage * kBStage + plane * (kTN * kARow) + bo[t]);
    };
#define MZ_MMA(MT, NT, AA, BB)                                                                                         \
    if (!(MZ_CONV3_SKIP & 1)) acc[MT][NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AA, BB, acc[MT][NT], 0, 0, 0);     \
    else asm volatile("" ::"v"(AA), "v"(BB));
    h8 xl[4], xh[4], wh[4], wl[4];
    a_offsets(0, ao);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xl[t] = read_a(1, t);
        wh[t] = read_b(0, 0, t);
        xh[t] = read_a(0, t);
        wl[t] = read_b(1, 0, t);
    }
    a_offsets(min(1, taps - 1), an);
    int s = 0, stage = 0;  // stage = s % 3
    long long seg[5] = {0, 0, 0, 0, 0};  // (stamped diagnostic build) cycles of a step's parts, summed over the steps
#define MZ_SEG(i)                                                          \
    if (MZ_CONV3_SKIP & 32) {                                              \
        const long long t_now = (long long)__builtin_amdgcn_s_memtime();   \
        seg[i] += t_now - t_seg;                                           \
        t_seg = t_now;                                                     \
    }
    for (int chunk = 0; chunk < chunks; ++chunk) {
        const bool next_chunk = chunk + 1 < chunks;
        for (int tap = 0; tap < taps; ++tap, ++s) {
            const int st1 = stage == 2 ? 0 : stage + 1, st2 = st1 == 2 ? 0 : st1 + 1;
            long long t_seg = (MZ_CONV3_SKIP & 32) ? (long long)__builtin_amdgcn_s_memtime() : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) ao[t] = an[t];
            MZ_SEG(0)
            __syncthreads();
            MZ_SEG(1)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) { MZ_MMA(mt, nt, xl[mt], wh[nt]) }
                xl[mt] = read_a(1, mt);
                if (mt == 0) {
                    // behind the first products, not in front of the barrier: the store of the weights of step
                    // s + 2, the request for those of step s + 3
                    if (!(MZ_CONV3_SKIP & 8)) {
                        stash_b(st2);
                        fetch_b_next();  // (step s + 3)
                    }
                    // the next chunk's pixel range: in flight behind the taps of this chunk.  Requested BEHIND the
                    // weights: loads return in order, and these come from memory -- in front they would hold the weights back
                    if (tap == 0 && next_chunk && !(MZ_CONV3_SKIP & 16)) fetch_halo(chunk + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MZ_SEG(2)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const h8 ws = scaled_hi(wh[nt]);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { MZ_MMA(mt, nt, xh[mt], ws) }
                wh[nt] = read_b(0, st1, nt);
                if (nt == 0) a_offsets(min(tap + 2, taps - 1), an);  // (the pixel operands of the next step's reads)
                __builtin_amdgcn_sched_barrier(0);
            }
            MZ_SEG(3)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { MZ_MMA(mt, nt, xh[mt], wl[nt]) }
                wl[nt] = read_b(1, st1, nt);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                MZ_MMA(mt, 3, xh[mt], wl[3])
                xh[mt] = read_a(0, mt);
            }
            wl[3] = read_b(1, st1, 3);
            MZ_SEG(4)
            stage = st1;
        }
        if (next_chunk && !(MZ_CONV3_SKIP & 16)) {
            __syncthreads();  // every wave has left the old range
            for_each_index(std::make_integer_sequence<int, 2 * NJ>{}, [&](auto k_tag) __attribute__((always_inline)) { convert_piece(chunk + 1, k_tag); });
            write_halo();
            __syncthreads();
            a_offsets(0, ao);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xl[t] = read_a(1, t);
                xh[t] = read_a(0, t);
            }
            a_offsets(min(1, taps - 1), an);
        }
    }
#undef MZ_MMA
#undef MZ_SEG
    __syncthreads();  // (the epilogue's tile overlays the operand images)

    const long long t_epi = (MZ_CONV3_SKIP & 32) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    conv3_epilogue(acc, lds, m0, n0, wm, wn, m, kg, bias, act, y, y3, y_plane, rows, N, sums);
    if ((MZ_CONV3_SKIP & 32) && y) {  // diagnostic: cycles of the three parts, per wave, over the first outputs of the tile
        __builtin_amdgcn_s_waitcnt(0);
        const long long t_end = (long long)__builtin_amdgcn_s_memtime();
        if (lane == 0) {
            float *dst = y + (long long)m0 * N + n0 + wave * 8;
            dst[0] = (float)(t_loop - t_start);
            dst[1] = (float)(t_epi - t_loop);
            dst[2] = (float)(t_end - t_epi);
            for (int i = 0; i < 5; ++i) dst[3 + i] = (float)seg[i];
        }
    }
}

__global__ void __launch_bounds__(256)
k_split3(const float *__restrict__ x, long long total, unsigned short *__restrict__ out, long long plane) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        unsigned short hi, lo;
        split2(x[i], hi, lo);
        out[i] = hi;
        out[plane + i] = lo;
    }
}

// The network input: [G][C][P] float32 (channels first) -> the two activation planes [G * P][C] (channels last),
// transposed through a 64-channel x 32-pixel LDS tile so that reads run along the pixels and writes
// along the channels.
__global__ void __launch_bounds__(256)
k_nchw_split3(const float *__restrict__ x, int C, int P, unsigned short *__restrict__ out, long long plane) {
    __shared__ float tile[64][33];
    const long long img = blockIdx.z;
    const float *src = x + img * (long long)C * P;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 64;
    {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 pixels x 8 channels per pass
        for (int j = ty; j < 64; j += 8) {
            const int c = c0 + j, p = p0 + tx;
            tile[j][tx] = (c < C && p < P) ? src[(long long)c * P + p] : 0.0f;
        }
    }
    __syncthreads();
    const int tc = threadIdx.x & 63, tp = threadIdx.x >> 6;  // 64 channels x 4 pixels per pass
    for (int j = tp; j < 32; j += 4) {
        const int c = c0 + tc, p = p0 + j;
        if (c >= C || p >= P) continue;
        unsigned short hi, lo;
        split2(tile[tc][j], hi, lo);
        const long long o = (img * P + p) * C + c;
        out[o] = hi;
        out[plane + o] = lo;
    }
}

__global__ void __launch_bounds__(256)
k_unsplit3(const unsigned short *__restrict__ x3, long long plane, long long total, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        out[i] = unsplit2(x3[i], x3[plane + i]);
}

// y = act(x * scale[g][c] + shift[g][c] + r): r = 0, the split activation res3 (identity skip) or
// res * rscale + rshift (raw output of the 1x1 skip convolution with its BatchNorm).
__global__ void __launch_bounds__(256)
k_bn_apply3(const float *__restrict__ x, const float *__restrict__ scale, const float *__restrict__ shift,
            const float *__restrict__ res, const float *__restrict__ rscale, const float *__restrict__ rshift,
            const unsigned short *__restrict__ res3, long long res_plane, int act, float *__restrict__ y,
            unsigned short *__restrict__ y3, long long y_plane, long long total, int PC, int C) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i / PC), c = (int)(i % C);
        const long long o = (long long)g * C + c;
        float v = fmaf(x[i], scale[o], shift[o]);  // the same single rounding as the staged form of k_conv3_same
        if (res) v += rscale ? fmaf(res[i], rscale[o], rshift[o]) : res[i];
        if (res3) v += unsplit2(res3[i], res3[res_plane + i]);
        v = act3(v, act);
        if (y) y[i] = v;
        if (y3) {
            unsigned short hi, lo;
            split2(v, hi, lo);
            y3[i] = hi;
            y3[y_plane + i] = lo;
        }
    }
}

// BatchNorm scale / shift of every (game, channel) from the block sums of conv3_epilogue:
// mean and biased variance over the P pixels of the game (P / 64 consecutive blocks, added in order),
// folded with the affine parameters exactly as k_bn_stats does.
__global__ void __launch_bounds__(64)
k_bn_from_sums(const double *__restrict__ sums, int P, int C, const float *__restrict__ weight, const float *__restrict__ bias,
               float eps, float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean_out,
               float *__restrict__ var_out) {
    const int g = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const int nb = P >> 6;
    const double *src = sums + ((long long)g * nb * C + c) * 2;
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < nb; ++i) {
        S += src[(long long)i * C * 2];
        Q += src[(long long)i * C * 2 + 1];
    }
    const double mean = S / P;
    double var = Q / P - mean * mean;
    if (var < 0.0) var = 0.0;
    const double sc = (double)weight[c] / sqrt(var + (double)eps);
    const long long o = (long long)g * C + c;
    scale[o] = (float)sc;
    shift[o] = (float)((double)bias[c] - mean * sc);
    if (mean_out) mean_out[o] = (float)mean;
    if (var_out) var_out[o] = (float)var;
}

inline int blocks3(long long total) { return (int)std::min<long long>((total + 255) / 256, 1 << 20); }

}  // namespace

extern "C" {

int mz_split3(const float *x, int64_t total, uint16_t *out, int64_t plane_stride, void *stream) {
    if (!x || !out || total <= 0 || plane_stride < total) return mz::fail(MZ_ERR_ARG, "mz_split3: NULL argument or planes that overlap");
    hipLaunchKernelGGL(k_split3, dim3(blocks3(total)), dim3(256), 0, (hipStream_t)stream, x, (long long)total, out,
                       (long long)plane_stride);
    return mz::check_launch("k_split3");
}

int mz_nchw_split3(const float *x, int32_t G, int32_t C, int32_t P, uint16_t *out, int64_t plane_stride, void *stream) {
    if (!x || !out || G <= 0 || C <= 0 || P <= 0) return mz::fail(MZ_ERR_ARG, "mz_nchw_split3: NULL or non-positive argument");
    if (plane_stride < (long long)G * C * P) return mz::fail(MZ_ERR_ARG, "mz_nchw_split3: planes overlap");
    if (G > 65535 || (C + 63) / 64 > 65535) return mz::fail(MZ_ERR_ARG, "mz_nchw_split3: more than 65535 images or channel tiles");
    hipLaunchKernelGGL(k_nchw_split3, dim3((P + 31) / 32, (C + 63) / 64, G), dim3(256), 0, (hipStream_t)stream, x, C, P, out,
                       (long long)plane_stride);
    return mz::check_launch("k_nchw_split3");
}

int mz_unsplit3(const uint16_t *x3, int64_t plane_stride, int64_t total, float *out, void *stream) {
    if (!x3 || !out || total <= 0 || plane_stride < total) return mz::fail(MZ_ERR_ARG, "mz_unsplit3: NULL argument or planes that overlap");
    hipLaunchKernelGGL(k_unsplit3, dim3(blocks3(total)), dim3(256), 0, (hipStream_t)stream, x3, (long long)plane_stride,
                       (long long)total, out);
    return mz::check_launch("k_unsplit3");
}

static int conv3_launch(const char *who, const uint16_t *x3, int64_t x_plane, const float *xf, int32_t C, const int32_t *idx, int32_t taps,
                        const uint16_t *w3, int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3,
                        int64_t y_plane, int32_t rows, int32_t N, double *block_sums, void *stream) {
    if ((!x3 && !xf) || !idx || !w3 || (!y && !y3) || rows <= 0 || taps <= 0)
        return mz::fail(MZ_ERR_ARG, "%s: NULL argument or non-positive size", who);
    if (C <= 0 || C % 64 != 0 || N <= 0 || N % kTN != 0)
        return mz::fail(MZ_ERR_ARG, "%s: input channels (%d) must be a multiple of 64 and output channels (%d) "
                        "a multiple of %d (other layers use mz_wide_forward_gather / mz_conv_direct_nhwc)", who, C, N, kTN);
    if (activation < MZ_ACT_NONE || activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "%s: activation %d is not element-wise", who, activation);
    const long long K = (long long)taps * C;
    if (w_plane < K * N || (!xf && x_plane <= 0) || (y3 && y_plane < (long long)rows * N) || K > 0x7fffffffLL)
        return mz::fail(MZ_ERR_ARG, "%s: plane strides too small", who);
    if ((reinterpret_cast<uintptr_t>(x3) | reinterpret_cast<uintptr_t>(xf) | reinterpret_cast<uintptr_t>(w3) |
         reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y3)) & 15 || (x_plane & 7) || (w_plane & 7) || (y_plane & 7))
        return mz::fail(MZ_ERR_ARG, "%s: operands must be 16-byte aligned", who);
    if (taps > 16) return mz::fail(MZ_ERR_ARG, "%s: at most 16 taps (kernel size up to 4)", who);
    if (block_sums && rows % 64 != 0) return mz::fail(MZ_ERR_ARG, "%s: block sums need a multiple of 64 rows", who);
    // two operand stages + the index table; the epilogue reuses the space as its output tile
    const size_t lds = std::max<size_t>(2 * (size_t)kStage + (size_t)taps * kTM * sizeof(int), kOutLds);
    auto launch = [&](auto kernel) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)std::max<size_t>(2 * (size_t)kStage + 16 * kTM * sizeof(int), kOutLds));
        if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "k_conv3: cannot enable %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        const dim3 grid(N / kTN, (rows + kTM - 1) / kTM);
        hipLaunchKernelGGL(kernel, grid, dim3(kT3), lds, (hipStream_t)stream, x3, (long long)x_plane, xf, C, idx, taps, w3,
                           (long long)w_plane, (int)K, bias, activation, y, y3, (long long)y_plane, rows, N, block_sums);
        return mz::check_launch("k_conv3");
    };
    return xf ? launch(k_conv3<true>) : launch(k_conv3<false>);
}

int mz_conv3_forward(const uint16_t *x3, int64_t x_plane, int32_t C, const int32_t *idx, int32_t taps, const uint16_t *w3,
                     int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3, int64_t y_plane,
                     int32_t rows, int32_t N, double *block_sums, void *stream) {
    if (!x3) return mz::fail(MZ_ERR_ARG, "mz_conv3_forward: x3 is NULL");
    return conv3_launch("mz_conv3_forward", x3, x_plane, nullptr, C, idx, taps, w3, w_plane, bias, activation, y, y3, y_plane, rows, N,
                        block_sums, stream);
}

int mz_conv3_forward_f32(const float *x, int32_t C, const int32_t *idx, int32_t taps, const uint16_t *w3, int64_t w_plane,
                         const float *bias, int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t rows, int32_t N,
                         double *block_sums, void *stream) {
    if (!x) return mz::fail(MZ_ERR_ARG, "mz_conv3_forward_f32: x is NULL");
    return conv3_launch("mz_conv3_forward_f32", nullptr, 0, x, C, idx, taps, w3, w_plane, bias, activation, y, y3, y_plane, rows, N,
                        block_sums, stream);
}

// Shape of a staged-range launch: groups of 128 slots, slots of the LDS image, games whose scale / shift rows a
// range can touch, LDS bytes.  Returns NULL when the kernel takes the shape, else why not.
struct SameShape { int nj, cap, bn_games; size_t lds; };
static const char *same_shape(int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, int32_t N, bool raw, bool with_bn,
                              SameShape &s, bool with_res = false, bool with_res_bn = false) {
    if (G <= 0 || H <= 0 || W <= 0) return "non-positive size";
    if (ksize < 1 || (ksize & 1) == 0 || ksize > 3) return "kernel size must be 1 or 3 (others use mz_conv3_forward)";
    if (C <= 0 || C % 64 != 0 || N <= 0 || N % kTN != 0)
        return "input channels must be a multiple of 64 and output channels a multiple of 128";
    const int reach = ((ksize - 1) / 2) * (W + 1), nslots = kTM + 2 * reach;
    s.nj = std::max(3, (nslots + 127) / 128);
    if (s.nj > 5) return "the image width exceeds the staged range (256 + 2 * (W + 1) slots, at most 640)";
    if (s.nj > 3 && !raw) return "an image width above 63 is staged from the float32 form only (mz_conv3_same_bn_forward)";
    s.cap = s.nj == 3 ? kHaloSlots : (nslots + 3) & ~3;
    // games a staged range can touch: it holds 256 + 2 * reach consecutive pixels
    s.bn_games = with_bn ? (int)std::min<long long>(G, (nslots - 2) / ((long long)H * W) + 2) : 0;
    if (with_res && (!with_bn || s.nj != 3)) return "the residual form needs the BatchNorm rows and an image width of at most 63";
    const size_t need = same_lds(s.cap, s.bn_games * (with_res_bn ? 2 : 1), C);
    s.lds = std::max<size_t>(need, kOutLds);
    if (need > (size_t)kMaxLds) return "the staged range and the scale / shift rows do not fit in LDS (apply the BatchNorm with mz_bn_apply3 first)";
    return nullptr;
}

static int conv3_same_launch(const char *who, const uint16_t *x3, int64_t x_plane, const float *xf, const float *in_scale,
                             const float *in_shift, int32_t in_act, int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize,
                             const uint16_t *w3, int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3,
                             int64_t y_plane, int32_t N, double *block_sums, void *stream, const float *res = nullptr,
                             const float *res_scale = nullptr, const float *res_shift = nullptr, float *act_out = nullptr) {
    const bool raw = xf != nullptr;
    if ((!x3 && !xf) || !w3 || (!y && !y3))
        return mz::fail(MZ_ERR_ARG, "%s: NULL argument", who);
    if (raw && in_scale && !in_shift) return mz::fail(MZ_ERR_ARG, "%s: the input shift is NULL", who);
    SameShape shp{};
    if (res_scale && !res_shift) return mz::fail(MZ_ERR_ARG, "%s: the residual's shift is NULL", who);
    if (const char *why = same_shape(C, G, H, W, ksize, N, raw, raw && in_scale, shp, res != nullptr, res_scale != nullptr))
        return mz::fail(MZ_ERR_ARG, "%s: %s (C %d, N %d, G %d, H %d, W %d, kernel %d)", who, why, C, N, G, H, W, ksize);
    if (activation < MZ_ACT_NONE || activation > MZ_ACT_SIGMOID || (raw && (in_act < MZ_ACT_NONE || in_act > MZ_ACT_SIGMOID)))
        return mz::fail(MZ_ERR_ARG, "%s: activation is not element-wise", who);
    const long long rows = (long long)G * H * W, K = (long long)ksize * ksize * C;
    if (rows > 0x7fffffffLL - kTM) return mz::fail(MZ_ERR_ARG, "%s: more than 2^31 pixels", who);
    if (block_sums && rows % 64 != 0) return mz::fail(MZ_ERR_ARG, "%s: block sums need a multiple of 64 rows", who);
    if (w_plane < K * N || (!raw && x_plane < rows * C) || (y3 && y_plane < rows * N))
        return mz::fail(MZ_ERR_ARG, "%s: plane strides too small", who);
    if ((reinterpret_cast<uintptr_t>(x3) | reinterpret_cast<uintptr_t>(xf) | reinterpret_cast<uintptr_t>(w3) |
         reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y3) | reinterpret_cast<uintptr_t>(res) |
         reinterpret_cast<uintptr_t>(act_out)) & 15 || (x_plane & 7) || (w_plane & 7) || (y_plane & 7))
        return mz::fail(MZ_ERR_ARG, "%s: operands must be 16-byte aligned", who);
    auto launch = [&](auto kernel) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "%s: cannot enable %zu bytes of LDS: %s", who, shp.lds, hipGetErrorString(e));
        const dim3 grid(N / kTN, (unsigned)((rows + kTM - 1) / kTM));
        hipLaunchKernelGGL(kernel, grid, dim3(kT3), shp.lds, (hipStream_t)stream, x3, (long long)x_plane, xf, in_scale, in_shift, res,
                           res_scale, res_shift, act_out, in_act, shp.bn_games, shp.cap, C, H, W, ksize, w3, (long long)w_plane, (int)K, bias, activation, y, y3,
                           (long long)y_plane, (int)rows, N, block_sums);
        return mz::check_launch("k_conv3_same");
    };
    if (!raw) return launch(k_conv3_same<false, 3>);
    if (res) return launch(k_conv3_same<true, 3, true>);
    switch (shp.nj) {
        case 3: return launch(k_conv3_same<true, 3>);
        case 4: return launch(k_conv3_same<true, 4>);
        default: return launch(k_conv3_same<true, 5>);
    }
}

int mz_conv3_same_fits(int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, int32_t N, int32_t raw, int32_t with_bn) {
    SameShape shp{};
    return same_shape(C, G, H, W, ksize, N, raw != 0, raw != 0 && with_bn != 0, shp, with_bn >= 2, with_bn >= 3) == nullptr ? 1 : 0;
}

int mz_conv3_same_block_forward(const float *x, const float *in_scale, const float *in_shift, const float *residual,
                                const float *res_scale, const float *res_shift, int32_t in_activation, float *block_out,
                                int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, const uint16_t *w3, int64_t w_plane,
                                const float *bias, int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t N,
                                double *block_sums, void *stream) {
    if (!x || !in_scale || !in_shift || !residual)
        return mz::fail(MZ_ERR_ARG, "mz_conv3_same_block_forward: x, its scale / shift and the residual are required");
    return conv3_same_launch("mz_conv3_same_block_forward", nullptr, 0, x, in_scale, in_shift, in_activation, C, G, H, W, ksize, w3,
                             w_plane, bias, activation, y, y3, y_plane, N, block_sums, stream, residual, res_scale, res_shift, block_out);
}

int mz_conv3_same_forward(const uint16_t *x3, int64_t x_plane, int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize,
                          const uint16_t *w3, int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3,
                          int64_t y_plane, int32_t N, double *block_sums, void *stream) {
    if (!x3) return mz::fail(MZ_ERR_ARG, "mz_conv3_same_forward: x3 is NULL");
    return conv3_same_launch("mz_conv3_same_forward", x3, x_plane, nullptr, nullptr, nullptr, 0, C, G, H, W, ksize, w3, w_plane, bias,
                             activation, y, y3, y_plane, N, block_sums, stream);
}

int mz_conv3_same_bn_forward(const float *x, const float *in_scale, const float *in_shift, int32_t in_activation, int32_t C,
                             int32_t G, int32_t H, int32_t W, int32_t ksize, const uint16_t *w3, int64_t w_plane,
                             const float *bias, int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t N,
                             double *block_sums, void *stream) {
    if (!x) return mz::fail(MZ_ERR_ARG, "mz_conv3_same_bn_forward: x is NULL");
    return conv3_same_launch("mz_conv3_same_bn_forward", nullptr, 0, x, in_scale, in_shift, in_activation, C, G, H, W, ksize, w3,
                             w_plane, bias, activation, y, y3, y_plane, N, block_sums, stream);
}

int mz_bn_stats_from_sums(const double *block_sums, int32_t G, int32_t P, int32_t C, const float *weight, const float *bias,
                          float eps, float *scale, float *shift, float *mean, float *var, void *stream) {
    if (!block_sums || !weight || !bias || !scale || !shift || G <= 0 || P <= 0 || C <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_bn_stats_from_sums: NULL or non-positive argument");
    if (P % 64 != 0) return mz::fail(MZ_ERR_ARG, "mz_bn_stats_from_sums: %d pixels per game are not whole 64-row blocks (use mz_bn_stats)", P);
    hipLaunchKernelGGL(k_bn_from_sums, dim3((C + 63) / 64, G), dim3(64), 0, (hipStream_t)stream, block_sums, P, C, weight, bias,
                       eps, scale, shift, mean, var);
    return mz::check_launch("k_bn_from_sums");
}

int mz_bn_apply3(const float *x, const float *scale, const float *shift, const float *residual, const float *res_scale,
                 const float *res_shift, const uint16_t *residual3, int64_t res_plane, int32_t activation, float *y,
                 uint16_t *y3, int64_t y_plane, int32_t G, int32_t P, int32_t C, void *stream) {
    if (!x || !scale || !shift || (!y && !y3) || G <= 0 || P <= 0 || C <= 0 || (res_scale && (!residual || !res_shift)) ||
        (residual && residual3))
        return mz::fail(MZ_ERR_ARG, "mz_bn_apply3: NULL or inconsistent argument");
    if (activation < MZ_ACT_NONE || activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "mz_bn_apply3: activation %d is not element-wise", activation);
    const long long total = (long long)G * P * C;
    if ((y3 && y_plane < total) || (residual3 && res_plane < total)) return mz::fail(MZ_ERR_ARG, "mz_bn_apply3: plane stride too small");
    hipLaunchKernelGGL(k_bn_apply3, dim3(blocks3(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, residual,
                       res_scale, res_shift, residual3, (long long)res_plane, activation, y, y3, (long long)y_plane, total,
                       P * C, C);
    return mz::check_launch("k_bn_apply3");
}

}  // extern "C"
