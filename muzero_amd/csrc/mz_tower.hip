// Residual-tower pieces of g and f (gfx950):
//   * a dense layer of any width, its reduction dimension streamed through LDS in chunks and
//     optionally split across workgroups (fc_merge and the head inputs at D = 9216), on
//     v_mfma_f32_16x16x4_f32 with the packed weight tiles of mz_mlp_pack;
//   * the ResBlock tower with the batch statistics of one network call per group of rows;
//   * policy post-processing.
// The convolutions of the shipped g/f towers are skinny (256 -> 3 -> 3 -> 256 channels on a
// 6 x 6 latent).  The first one (and the 1 x 1 skip convolution beside it) is one matrix product
// over (output channel, tap) columns -- 27 + 3 of 32 columns of two 16 x 16 x 4 tiles -- whose
// shifted rows are then summed; the 3 -> 3 convolution after it is a thread per output value.
// Blocks with more output channels, or whose (channel, tap) columns exceed 32, run on the
// 4 x 4 x 1 matrix instruction (16 independent 4-position x 4-channel blocks per wave), where a
// 16-wide tile would idle on most of its columns.
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include <cstring>

#include "mz_common.h"
#include "mz_mlp_device.h"

using namespace mzd;

namespace {

// Partial sums of mz_wide_forward_static's split reduction: a fixed device array instead of a
// caller-provided workspace, for callers that cannot allocate.  One copy per GPU, held until the
// process ends; its contents are whatever the last call on that GPU left (every call writes each
// partial sum before k_wide_finish reads it, so nothing needs clearing).  Host code reaches it only
// through the per-device address hipGetSymbolAddress returns -- the array's name is not a device
// pointer on the host.
constexpr long long kWideScratchFloats = 4ll << 20;  // 16 MiB
constexpr int kMaxDevices = 64;
static __device__ __align__(256) float g_wide_scratch[kWideScratchFloats];
static float *g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

constexpr int kWT = 256;       // threads per workgroup (the tower kernel has its own count)
constexpr int kKC = 64;        // floats of the reduction dimension per LDS chunk (4 k-tiles)
constexpr int kLdX = kKC + 4;  // LDS row stride of a chunk: 16 rows x b128 reads hit 64 distinct banks
constexpr size_t kMaxLds = 160 * 1024;

// ------------------------------------------------------------------ wide dense layer
// Workgroup = 2 x 2 waves, wave tile = (16 MTW) x (16 NTW), workgroup tile TM x TN = 32 MTW x 32 NTW.
// Activations: global -> registers -> LDS chunk (double buffered, one barrier per chunk);
// weights: packed tiles straight from global/L2 into registers, one k-tile ahead.
//
// SEG (convolutions as gathered products): the reduction dimension is `n_seg` segments of
// seg_chunks * 64 floats, and segment s of output row r is read from input row x_rows[s * B + r]
// (negative: zeros) -- one segment per kernel tap, the input rows being the pixels of a
// channels-last image whose channel count is the segment length.
template <int MTW, int NTW, bool VEC, bool SEG = false>
__global__ void __launch_bounds__(kWT) __attribute__((amdgpu_waves_per_eu(2)))
k_wide(mz_packed_layer L, const float *__restrict__ blob, const float *__restrict__ x, long long ldx,
       const int32_t *__restrict__ x_rows, float *__restrict__ y, long long ldy, int B, int splitk,
       int chunks_per_split, float *__restrict__ ws, int seg_chunks = 0) {
    constexpr int TM = 32 * MTW, TN = 32 * NTW, XR = TM / 16;
    extern __shared__ __align__(16) float xs[];  // [2][TM][kLdX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m = lane & 15, kq = lane >> 4;
    const int KT = L.Kp >> 4, NT = L.Np >> 4;
    const int n0 = blockIdx.x * TN, m0 = blockIdx.y * TM, z = blockIdx.z;
    const int nchunks = (KT + 3) >> 2;
    const int c_lo = z * chunks_per_split, c_hi = min(nchunks, c_lo + chunks_per_split);
    const int c_full = min(c_hi, KT >> 2);  // chunks [c_lo, c_full) hold four k-tiles

    // activation rows of this thread: source row index (-1: zeros) kept as 32 bits
    const int sc = (tid & 15) * 4, sr = tid >> 4;
    int srow[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int r = m0 + sr + 16 * i;
        int src = -1;
        if (!SEG && r < B) src = x_rows ? x_rows[r] : r;
        srow[i] = src;
    }
    constexpr int XH = XR / 2;  // the chunk is fetched in two halves to halve the registers it holds
    // VEC: branch-free -- an absent row or a column past K reads a valid address here and is
    // zeroed when it is written to LDS (so nothing waits for the load where it is issued)
    auto fetch = [&](int c, int half, f32x4 (&v)[XH]) {
        int k = c * kKC + sc;
        if (SEG) {
            const int seg = c / seg_chunks;
            k -= seg * seg_chunks * kKC;  // offset inside the segment = channel of the pixel
            if (half == 0 && (c == c_lo || c - seg * seg_chunks == 0)) {
#pragma unroll
                for (int i = 0; i < XR; ++i) {
                    const int r = m0 + sr + 16 * i;
                    srow[i] = r < B ? x_rows[(long long)seg * B + r] : -1;
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < XH; ++ii) {
            const int i = half * XH + ii;
            if (VEC) {
                const float *p = x + (long long)max(srow[i], 0) * ldx + (SEG ? k : min(k, L.K - 4));
                v[ii] = *reinterpret_cast<const f32x4 *>(p);
            } else {
                f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
                if (srow[i] >= 0) {
                    const float *p = x + (long long)srow[i] * ldx;
                    if (k < L.K) t.x = p[k];
                    if (k + 1 < L.K) t.y = p[k + 1];
                    if (k + 2 < L.K) t.z = p[k + 2];
                    if (k + 3 < L.K) t.w = p[k + 3];
                }
                v[ii] = t;
            }
        }
    };
    auto stash = [&](int buf, int c, int half, const f32x4 (&v)[XH]) {
        const bool kin = c * kKC + sc < L.K;
#pragma unroll
        for (int ii = 0; ii < XH; ++ii) {
            const int i = half * XH + ii;
            f32x4 t = v[ii];
            if (VEC && !(kin && srow[i] >= 0)) t = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4 *>(&xs[(buf * TM + sr + 16 * i) * kLdX + sc]) = t;
        }
    };
    // Packed weight tiles of this wave's column tiles.  The ring below requests up to 3 tiles
    // past the last one of a column tile: those addresses are the next column tile's weights,
    // or the bias and the read slack behind the layer -- valid memory that feeds no arithmetic.
    const f32x4 *wq[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
        wq[t] = reinterpret_cast<const f32x4 *>(blob + L.w_off) + lane +
                ((long long)min((n0 >> 4) + wn * NTW + t, NT - 1) * KT + c_lo * 4) * 64;
    // tile (first tile of the current chunk) + j
    auto wload = [&](int j, f32x4 (&w)[NTW]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) w[t] = wq[t][j * 64];
    };

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    auto aload = [&](int buf, int t, f32x4 (&a)[MTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
            a[mt] = *reinterpret_cast<const f32x4 *>(
                &xs[(buf * TM + (wm * MTW + mt) * 16 + m) * kLdX + t * 16 + kq * 4]);
    };
    // 4 k-steps of one k-tile; the k-step is the outer loop so that consecutive MFMAs never
    // depend on each other
    auto mm = [&](const f32x4 (&a)[MTW], const f32x4 (&w)[NTW]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], w[nt][j], acc[mt][nt], 0, 0, 0);
    };

    if (c_lo < c_hi) {
        // Weights stream from HBM: three k-tiles are in flight per wave (ring w0..w3) so that a
        // request is about 200 MFMAs old when it is consumed; the scheduling barriers keep the
        // requests ahead of the MFMAs they were written before.
        f32x4 xv[XH], w0[NTW], w1[NTW], w2[NTW], w3[NTW], a0[MTW], a1[MTW];
        fetch(c_lo, 0, xv);
        wload(0, w0);
        wload(1, w1);
        wload(2, w2);
        stash(0, c_lo, 0, xv);
        fetch(c_lo, 1, xv);
        stash(0, c_lo, 1, xv);
        __syncthreads();
        int buf = 0;
        for (int c = c_lo; c < c_full; ++c) {
            const bool more = c + 1 < c_hi;  // a full chunk may still be the last of this split
            if (more) fetch(c + 1, 0, xv);
            aload(buf, 0, a0);
            wload(3, w3);
            aload(buf, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, w0);
            wload(4, w0);
            aload(buf, 2, a0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, w1);
            if (more) {
                stash(buf ^ 1, c + 1, 0, xv);
                fetch(c + 1, 1, xv);
            }
            wload(5, w1);
            aload(buf, 3, a1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, w2);
            wload(6, w2);
#pragma unroll
            for (int t = 0; t < NTW; ++t) wq[t] += 4 * 64;
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, w3);
            if (more) stash(buf ^ 1, c + 1, 1, xv);
            __syncthreads();
            buf ^= 1;
        }
        if (c_full < c_hi) {
            // last, partial chunk of the layer's reduction: its 1..3 tiles are already in w0..w2
            const int nk = KT - c_full * 4;
            aload(buf, 0, a0);
            mm(a0, w0);
            if (nk > 1) {
                aload(buf, 1, a1);
                mm(a1, w1);
            }
            if (nk > 2) {
                aload(buf, 2, a0);
                mm(a0, w2);
            }
        }
    }

    const float *bias = blob + L.b_off;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int col = n0 + (wn * NTW + nt) * 16 + m;
        if (col >= L.Np) continue;
        const float b = bias[col];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = m0 + (wm * MTW + mt) * 16 + kq * 4 + j;
                if (row >= B) continue;
                if (splitk == 1) {
                    if (col < L.N) y[(long long)row * ldy + col] = activate(acc[mt][nt][j] + b, L.act);
                } else {
                    ws[((long long)z * B + row) * L.Np + col] = acc[mt][nt][j];
                }
            }
    }
}

// y = act(sum over splits (fixed order) + bias)
__global__ void __launch_bounds__(kWT)
k_wide_finish(mz_packed_layer L, const float *__restrict__ blob, const float *__restrict__ ws, int splitk, int B,
              float *__restrict__ y, long long ldy) {
    const long long total = (long long)B * L.N;
    const float *bias = blob + L.b_off;
    for (long long i = (long long)blockIdx.x * kWT + threadIdx.x; i < total; i += (long long)gridDim.x * kWT) {
        const int r = (int)(i / L.N), c = (int)(i - (long long)r * L.N);
        float s = 0.0f;
        for (int z = 0; z < splitk; ++z) s += ws[((long long)z * B + r) * L.Np + c];
        y[(long long)r * ldy + c] = activate(s + bias[c], L.act);
    }
}

struct WideShape {
    bool big;
    int tm, tn, tiles_m, tiles_n, splitk, chunks_per_split;
};

WideShape wide_shape(const mz_packed_layer &L, int B) {
    WideShape s{};
    s.big = B > 64 && L.Np > 64;
    // a layer with a handful of 128 x 128 tiles (the heads: 9216 -> 256 or 128 on a few hundred rows)
    // would run as few, deeply split workgroups of three chunks each; 64 x 64 tiles give four times
    // the workgroups at a quarter of the partial-sum traffic (measured: 37 -> 23 us at 256 x 9216 x 256)
    if (s.big && (long long)((B + 127) / 128) * ((L.Np + 127) / 128) < 12) s.big = false;
    // a short reduction (the bottleneck form of fc_merge: K = 108) is all prologue and epilogue per
    // workgroup: more, smaller workgroups hide that latency better
    static const int short_k = [] {
        const char *e = getenv("MZ_WIDE_SHORT_K");  // tuning aid: chunks at or below which 64 x 64 tiles are used
        return e ? atoi(e) : 4;
    }();
    if (s.big && (((L.Kp >> 4) + 3) >> 2) <= short_k) s.big = false;
    s.tm = s.tn = s.big ? 128 : 64;
    s.tiles_m = (B + s.tm - 1) / s.tm;
    s.tiles_n = (L.Np + s.tn - 1) / s.tn;
    const int nchunks = ((L.Kp >> 4) + 3) >> 2;
    // Split of the reduction dimension: a CU holds two 4-wave workgroups and works through
    // ceil(workgroups / 256) of them; one workgroup costs its chunks plus a fixed part (first
    // loads, partial-sum store) of about `kFixed` chunks.  Take the split with the least
    // estimated time; ties go to the smaller split (less partial-sum traffic).
    constexpr int kFixed = 4, kMaxSplit = 64;
    const long long tiles = (long long)s.tiles_m * s.tiles_n;
    long long best = -1;
    int best_k = 1;
    static const int forced = [] {
        const char *e = getenv("MZ_WIDE_SPLITK");  // tuning aid
        return e ? atoi(e) : 0;
    }();
    for (int k = 1; k <= std::min(kMaxSplit, nchunks); ++k) {
        const int cps = (nchunks + k - 1) / k;
        const int real_k = (nchunks + cps - 1) / cps;
        if (real_k != k) continue;
        const long long rounds = (tiles * k + 255) / 256;
        const long long cost = rounds * (cps + kFixed);
        if (best < 0 || cost < best) {
            best = cost;
            best_k = k;
        }
    }
    if (forced > 0) best_k = std::min(forced, nchunks);
    s.chunks_per_split = (nchunks + best_k - 1) / best_k;
    s.splitk = (nchunks + s.chunks_per_split - 1) / s.chunks_per_split;
    return s;
}

template <typename K>
int allow_lds(K kernel, size_t bytes, const char *what) {
    if (bytes <= 64 * 1024) return MZ_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
        return mz::fail(MZ_ERR_HIP, "%s: cannot enable %zu bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    return MZ_OK;
}

// ------------------------------------------------------------------ residual tower
constexpr int kTT = 512;       // threads of the tower kernel
constexpr int kTW = kTT / 64;  // its waves

struct TowerGeom {
    int C, H, W, HW, P;  // P = positions held by a workgroup (rows per workgroup * HW)
    int group0, seg;     // rows per reference call, seg = group0 * HW positions normalised together
    int cmax;            // widest block output
    int stage_c;         // input channels of the first block staged per pass
    int pw, ks;          // waves across positions / slices of the input channels (pw * ks = kTW)
    int vec_ok;          // rows may be read as 16-byte vectors
    int gemm1;           // the first block's convolutions run as one matrix product (see gemm_conv1)
    int y_ld;            // its result's row stride in LDS
    int part_ld;         // row stride of its per-wave partial sums
    int off_a, off_b, off_c, off_red, off_par, off_stage, off_w;  // LDS offsets in floats
};

constexpr int COB = 4;  // output channels accumulated together per pass
constexpr int kPar = 512;  // floats of LDS for a block's small parameters (one per thread)

// Diagnostics build (-DMZ_TOWER_TIMELINE): workgroup 0 records the shader clock at the marked
// points of its last launch; mz_debug_tower_stamps copies them out.
#ifdef MZ_TOWER_TIMELINE
__device__ unsigned long long g_tower_stamps[16];
#define MZ_TT(i)                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_tower_stamps[i] = __builtin_readcyclecounter();
#else
#define MZ_TT(i)
#endif

// Weights are read-only for the whole launch and indexed wave-uniformly: reading them through
// the constant address space makes those reads scalar loads (one fetch per wave, no vector
// memory instruction, no vector register).
typedef __attribute__((address_space(4))) const float cst_f;
__device__ __forceinline__ cst_f *as_constant(const float *p) { return (cst_f *)p; }

// Sum of a double over the wave, the same value in every lane: an inclusive scan inside each row
// of 16 lanes on the DPP path (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0),
// then the four row totals added in order.  No LDS traffic, unlike a shuffle tree.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_total(double v) {
    v = dpp_add<0x111>(v);
    v = dpp_add<0x112>(v);
    v = dpp_add<0x114>(v);
    v = dpp_add<0x118>(v);
    auto row_total = [&](int lane) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                                __builtin_amdgcn_readlane(__double2loint(v), lane));
    };
    return ((row_total(15) + row_total(31)) + row_total(47)) + row_total(63);
}

// In-place BatchNorm2d with the statistics of the P values of each channel (mean and biased
// variance, accumulated in double as torch's CPU kernels do), then the activation.  One wave per
// channel; a second buffer (the skip path), if given, is normalised by other waves at the same
// time.  gamma/beta come from LDS (lg/lb, fetched earlier) when given, else from global memory.
// `stat` (optional): the layer's batch statistics go to global memory, mean f32[calls][C] then
// biased variance f32[calls][C], call = call0 + segment -- what a running-statistics update needs.
struct BnJob {
    lds_f *buf;
    const float *gamma, *beta;
    const lds_f *lg, *lb;
    float eps;
    int act;
    float *stat;
};
__device__ void bn_act(const BnJob &j0, const BnJob &j1, int C, int P, int seg, int call0 = 0, int calls = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = P / seg, per_buf = C * nseg;  // a workgroup may hold several calls: one segment each
    const int jobs = j1.buf ? 2 * per_buf : per_buf;
    for (int job = wave; job < jobs; job += kTW) {
        const bool second = job >= per_buf;
        const BnJob &J = second ? j1 : j0;
        const int jj = second ? job - per_buf : job;
        const int c = jj / nseg, sg = jj - c * nseg;
        lds_f *v = J.buf + c * P + sg * seg;
        // one sweep: sums of (x - pivot) and (x - pivot)^2, the pivot (the channel's first value)
        // keeping the squares small
        const double pivot = (double)v[0];
        double s = 0.0, q = 0.0;
        for (int p = lane; p < seg; p += 64) {
            const double d = (double)v[p] - pivot;
            s += d;
            q += d * d;
        }
        s = wave_total(s);
        q = wave_total(q);
        const double ds = s / seg, mean = pivot + ds;
        const double var = fmax(q / seg - ds * ds, 0.0);
        const double invstd = 1.0 / sqrt(var + (double)J.eps);
        const double ga = J.lg ? (double)J.lg[c] : (double)as_constant(J.gamma)[c];
        const double be = J.lb ? (double)J.lb[c] : (double)as_constant(J.beta)[c];
        const float scale = (float)(ga * invstd);
        const float shift = (float)(be - mean * ga * invstd);
        if (J.stat != nullptr && lane == 0) {
            J.stat[(long long)(call0 + sg) * C + c] = (float)mean;
            J.stat[(long long)(calls + call0 + sg) * C + c] = (float)var;
        }
        for (int p = lane; p < seg; p += 64) v[p] = activate(v[p] * scale + shift, J.act);
    }
}

// Adds the k x k convolution (and, if DOWN, the 1x1 convolution) over the input channels
// [c_lo, c_hi), held in LDS as src[(ci - c_base) * P + position], for 64 positions x COB output
// channels per wave on v_mfma_f32_4x4x1_16b_f32: the instruction performs 16 independent 4 x 4
// rank-1 updates, block b = lanes 4b..4b+3.  Lane 4b+i supplies the input value of position
// (first position of the wave) + 4b+i, lane 4b+j the weight of output channel j (the same for
// every block), and element i of lane 4b+j's accumulator is the sum for position 4b+i, channel j.
// So a (channel, tap) step costs each lane one 4-byte read of its input value, one 4-byte read
// of a weight (4 distinct addresses per wave) and one matrix instruction -- no 16-wide tile to
// pad the 3 output channels of the shipped towers into.
// `p` = this lane's position (valid or not: `live`).  The staged weights wk[(ci - c_base)][tap][COB]
// and wd[(ci - c_base)][COB] come from load_weights.  KF = k when known at compile time.
template <bool DOWN, int KF>
__device__ __forceinline__ void conv_accumulate(const lds_f *src, int c_base, int c_lo, int c_hi, const lds_f *wk,
                                                const lds_f *wd, int k_rt, int H, int W, int P, int p, bool live,
                                                f32x4 &acc, f32x4 &accd) {
    static_assert(COB == 4, "one 4 x 4 block column per output channel");
    const int k = KF ? KF : k_rt;
    const int HW = H * W, pad = (k - 1) >> 1, kk = k * k;
    const int row = p / HW, s = p - row * HW, py = s / W, px = s - py * W;
    const int j = threadIdx.x & 3;
    if (KF) {
        // tap offsets and validity once per call; every read below is unconditional (a tap outside
        // the image reads the centre and is zeroed by a select), so the loop body is straight-line
        // code the scheduler can overlap
        constexpr int KK = KF * KF;
        int off[KK ? KK : 1];
        bool ok[KK ? KK : 1];
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int yy = py + t / KF - pad, xx = px + t % KF - pad;
            ok[t] = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
            off[t] = ok[t] ? yy * W + xx : s;
        }
#pragma unroll 2
        for (int ci = c_lo; ci < c_hi; ++ci) {
            const lds_f *plane = src + (ci - c_base) * P + row * HW;
            const lds_f *wrow = wk + (ci - c_base) * KK * COB + j;
            float v[KK ? KK : 1], w[KK ? KK : 1];
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                v[t] = plane[off[t]];
                w[t] = wrow[t * COB];
            }
            float vc = 0.0f, wc = 0.0f;
            if (DOWN) {
                vc = plane[s];
                wc = wd[(ci - c_base) * COB + j];
            }
#pragma unroll
            for (int t = 0; t < KK; ++t)
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(ok[t] ? v[t] : 0.0f, w[t], acc, 0, 0, 0);
            if (DOWN) accd = __builtin_amdgcn_mfma_f32_4x4x1f32(live ? vc : 0.0f, wc, accd, 0, 0, 0);
        }
        return;
    }
    for (int ci = c_lo; ci < c_hi; ++ci) {
        const lds_f *plane = src + (ci - c_base) * P + row * HW;
        const lds_f *wrow = wk + (ci - c_base) * kk * COB + j;
        for (int ky = 0; ky < k; ++ky) {
            const int yy = py + ky - pad;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = px + kx - pad;
                const bool in = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float v = in ? plane[yy * W + xx] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(v, wrow[(ky * k + kx) * COB], acc, 0, 0, 0);
            }
        }
        if (DOWN) {
            const float v = live ? plane[s] : 0.0f;
            accd = __builtin_amdgcn_mfma_f32_4x4x1f32(v, wd[(ci - c_base) * COB + j], accd, 0, 0, 0);
        }
    }
}

template <bool DOWN>
__device__ __forceinline__ void conv_dispatch(const lds_f *src, int c_base, int c_lo, int c_hi, const lds_f *wk,
                                              const lds_f *wd, int k, int H, int W, int P, int p, bool live,
                                              f32x4 &acc, f32x4 &accd) {
    if (k == 3) conv_accumulate<DOWN, 3>(src, c_base, c_lo, c_hi, wk, wd, k, H, W, P, p, live, acc, accd);
    else if (k == 1) conv_accumulate<DOWN, 1>(src, c_base, c_lo, c_hi, wk, wd, k, H, W, P, p, live, acc, accd);
    else conv_accumulate<DOWN, 0>(src, c_base, c_lo, c_hi, wk, wd, k, H, W, P, p, live, acc, accd);
}

// Sum over the taps of a k x k window for one output value, inputs and weights in LDS:
//   sum_t  wsrc[t * w_stride] * src[(tap t's position) ... ]  with src advanced by `src_tap` per tap
// (src_tap = 0: one plane, taps select positions only; otherwise tap t reads plane t).  KF = k
// when known at compile time: the reads are then unconditional (a tap outside the image reads the
// centre and is dropped by a select) and issued together.
template <int KF>
__device__ __forceinline__ float window_sum(const lds_f *src, int src_tap, const lds_f *wsrc, int k_rt, int H, int W,
                                            int py, int px) {
    const int k = KF ? KF : k_rt, pad = (k - 1) >> 1;
    float v = 0.0f;
    if (KF) {
        constexpr int KK = KF ? KF * KF : 1;
        float x[KK], w[KK];
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int yy = py + t / KF - pad, xx = px + t % KF - pad;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            x[t] = src[t * src_tap + (ok ? yy * W + xx : py * W + px)];
            w[t] = ok ? (wsrc ? wsrc[t] : 1.0f) : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < KK; ++t) v += w[t] * x[t];
        return v;
    }
    for (int ky = 0; ky < k; ++ky) {
        const int yy = py + ky - pad;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int xx = px + kx - pad, t = ky * k + kx;
            if (xx >= 0 && xx < W) v += (wsrc ? wsrc[t] : 1.0f) * src[t * src_tap + yy * W + xx];
        }
    }
    return v;
}
__device__ __forceinline__ float window_sum_k(const lds_f *src, int src_tap, const lds_f *wsrc, int k, int H, int W,
                                              int py, int px) {
    if (k == 3) return window_sum<3>(src, src_tap, wsrc, k, H, W, py, px);
    if (k == 1) return window_sum<1>(src, src_tap, wsrc, k, H, W, py, px);
    return window_sum<0>(src, src_tap, wsrc, k, H, W, py, px);
}

constexpr int kG1Cols = 32;  // columns of the first block's matrix product (two 16-wide tiles)

// One workgroup per group of rows (one network call of the reference).  LDS holds three
// activation buffers of cmax x P floats (a: block input / result, b: after conv1, c: skip path
// or conv2 output), the staging area for the first block's input and the slice partials.
template <int DEPTH>
__device__ __forceinline__ void
tower_body(const mz_tower &T, const TowerGeom &G, const float *__restrict__ x, long long ldx,
           const int32_t *__restrict__ x_rows, const float *__restrict__ row_add, long long ld_add,
           const int32_t *__restrict__ row_sel, float *__restrict__ y, long long ldy,
           const int32_t *__restrict__ y_rows, float *__restrict__ t_out, long long ld_t, int group,
           float *__restrict__ stats, int calls) {
    extern __shared__ __align__(16) float smem[];
    lds_f *S = to_lds(smem);
    lds_f *cur = S + G.off_a, *mid = S + G.off_b, *aux = S + G.off_c, *red = S + G.off_red, *stage = S + G.off_stage;
    lds_f *wst = S + G.off_w;  // staged convolution weights
    lds_f *par = S + G.off_par;  // the current block's small parameters
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = G.HW, P = G.P;
    const int r0 = blockIdx.x * group;
    const int pwi = wave % G.pw, ksi = wave / G.pw;  // position slot / channel slice of this wave
    const int pstride = G.pw * 64;

    auto src_row = [&](int r) -> const float * {
        const long long s = x_rows ? (long long)x_rows[r0 + r] : (long long)(r0 + r);
        return s < 0 ? nullptr : x + s * ldx;
    };
    auto add_row = [&](int r) -> const float * {
        if (row_add == nullptr) return nullptr;
        const int sel = row_sel ? row_sel[r0 + r] : r % G.group0;
        return row_add + (long long)sel * ld_add;
    };
    // tower input (merge split applied) for channels [cb, cb+cn) -> dst[c * P + p].  A row's
    // channels are contiguous in memory (cn * HW floats), so the copy is coalesced.
    auto load_input = [&](lds_f *dst, int cb, int cn) {
        const int n = cn * HW;
        if (G.vec_ok) {
            // all rows of the call as one index space of 16-byte items, four items per thread in
            // flight: the loads of a batch are issued before any of them is used
            const int per_row = n >> 2, total = group * per_row;
            constexpr int UN = 4;
            for (int i0 = tid; i0 < total; i0 += kTT * UN) {
                f32x4 v[UN], a[UN];
                int item[UN];
                bool has_add[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    item[u] = i0 + u * kTT;
                    v[u] = a[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    has_add[u] = false;
                    if (item[u] < total) {
                        const int r = item[u] / per_row, j = (item[u] - r * per_row) << 2;
                        const float *xr = src_row(r), *ar = add_row(r);
                        if (xr) v[u] = *reinterpret_cast<const f32x4 *>(xr + cb * HW + j);
                        if (ar) {
                            a[u] = *reinterpret_cast<const f32x4 *>(ar + cb * HW + j);
                            has_add[u] = true;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (item[u] >= total) continue;
                    const int r = item[u] / per_row, j = (item[u] - r * per_row) << 2;
                    f32x4 w = v[u];
                    if (has_add[u]) {
                        w.x = fmaxf(w.x + a[u].x, 0.0f);
                        w.y = fmaxf(w.y + a[u].y, 0.0f);
                        w.z = fmaxf(w.z + a[u].z, 0.0f);
                        w.w = fmaxf(w.w + a[u].w, 0.0f);
                    }
                    const int c = j / HW, s = j - c * HW;  // HW % 4 == 0: the four share a channel
                    *reinterpret_cast<lds_f4 *>(dst + r * HW + c * P + s) = w;
                }
            }
            return;
        }
        for (int r = 0; r < group; ++r) {
            const float *xr = src_row(r), *ar = add_row(r);
            if (xr) xr += cb * HW;
            if (ar) ar += cb * HW;
            lds_f *d = dst + r * HW;
            for (int j = tid; j < n; j += kTT) {
                float v = xr ? xr[j] : 0.0f;
                if (ar) v = fmaxf(v + ar[j], 0.0f);
                const int c = j / HW, s = j - c * HW;
                d[c * P + s] = v;
            }
        }
    };
    // convolution weights of output channels co0 .. co0+COB-1 (surplus channels repeat the last
    // one and are dropped later) and input channels [cb, cb+cn): wst[(c * kk + tap) * COB + o],
    // followed by the 1x1 skip weights [c * COB + o].  Returns where the latter start.
    auto load_weights = [&](const float *w_g, const float *wd_g, int co0, int Cout, int Cin, int cb, int cn,
                            int kk) -> lds_f * {
        lds_f *wdst = wst + cn * kk * COB;
        for (int o = 0; o < COB; ++o) {
            const int co = min(co0 + o, Cout - 1);
            const float *src = w_g + ((long long)co * Cin + cb) * kk;
            for (int i = tid; i < cn * kk; i += kTT) wst[i * COB + o] = src[i];
            if (wd_g) {
                const float *sd = wd_g + (long long)co * Cin + cb;
                for (int i = tid; i < cn; i += kTT) wdst[i * COB + o] = sd[i];
            }
        }
        return wdst;
    };
    // every wave leaves its slice's partial sums in `red`
    auto put_partials = [&](const f32x4 &acc, int which, int slot) {
        for (int o = 0; o < COB; ++o) red[((ksi * 2 + which) * COB + o) * pstride + slot] = acc[o];
    };
    // all threads: sum the `ks` slices in a fixed order and store, same element -> (channel,
    // position) map as store_acc
    auto reduce_store = [&](lds_f *dst, int which, int ks, int co0, int Cout, int pb, const float *bias_map) {
        for (int e = tid; e < COB * pstride; e += kTT) {
            const int slot = e % pstride, o = e / pstride;
            float s = 0.0f;
            for (int q = 0; q < ks; ++q) s += red[((q * 2 + which) * COB + o) * pstride + slot];
            const int ln = slot & 63, co = co0 + (ln & 3), pos = pb + (slot & ~63) + (ln & ~3) + o;
            if (co < Cout && pos < P) dst[co * P + pos] = s + (bias_map ? bias_map[co * HW + pos % HW] : 0.0f);
        }
    };
    // convolutions over a handful of input channels are not worth slicing
    auto slices_for = [&](int cin) { return cin <= 8 ? 1 : G.ks; };
    // accumulator of lane 4b+j, element i = position (wave's first position) + 4b+i, channel co0+j;
    // bias_map (optional) adds a per-(channel, position-in-image) constant
    auto store_acc = [&](lds_f *dst, const f32x4 &acc, int co0, int Cout, int first_pos, const float *bias_map) {
        const int co = co0 + (lane & 3);
        if (co >= Cout) return;
        for (int i = 0; i < 4; ++i) {
            const int pos = first_pos + (lane & ~3) + i;
            if (pos < P) dst[co * P + pos] = acc[i] + (bias_map ? bias_map[co * HW + pos % HW] : 0.0f);
        }
    };

    MZ_TT(0)
    long long stat_off = 0;
    for (int b = 0; b < T.n_blocks; ++b) {
        const mz_resblock &R = T.block[b];
        const int Cin = R.in_channels, Cout = R.out_channels, k = R.kernel_size;
        const bool down = R.down_w != nullptr;
        const bool from_global = b == 0;
        // A block with few channels: its second convolution's weights and the BatchNorm
        // parameters are fetched now, one value per thread, and parked in LDS after conv1, so
        // that the later stages do not wait on global memory.
        const int n2 = Cout * Cout * k * k, npar = n2 + 6 * Cout;
        const bool small = npar <= kPar && Cout * k * k <= 128;
        float pv = 0.0f;
        if (small && tid < npar) {
            if (tid < n2) pv = R.conv2_w[tid];
            else {
                const int j = (tid - n2) / Cout, c = (tid - n2) - j * Cout;
                const float *src = j == 0 ? R.bn1_w : j == 1 ? R.bn1_b : j == 2 ? R.bn2_w : j == 3 ? R.bn2_b
                                   : j == 4 ? R.dbn_w : R.dbn_b;
                if (src) pv = src[c];
            }
        }
        const lds_f *pg = small ? par + n2 : nullptr;  // [bn1_w, bn1_b, bn2_w, bn2_b, dbn_w, dbn_b][Cout]
        // ---- conv1 -> mid, and the 1x1 skip convolution of the same input -> aux
        if (from_global && G.gemm1) {
            // The first block reads C channels and writes a few: as a matrix product
            //   Yt[column][position] = sum_ci  W[ci][column] * input[ci][position],
            // columns = (output channel, tap) pairs followed by the skip convolution's channels,
            // the k x k convolution is the sum over taps of Yt shifted by the tap's offset.  The
            // input goes from global memory straight into the matrix instruction's operand (no
            // staging): lane (m, q) of v_mfma_f32_16x16x4 supplies positions 4m..4m+3 of channel
            // 4*step + q, one 16-byte load, used by the four tiles t = 0..3 whose row m is position
            // 4m + t.  The waves split the channels; their partial tiles are summed in a fixed order.
            const int kk = k * k, ncol = Cout * kk + (down ? Cout : 0);
            lds_f *part = stage, *Yt = stage + (kTW / 2) * kG1Cols * G.part_ld;
            const int m = lane & 15, q = lane >> 4;
            const int nsteps = (Cin + 3) >> 2;
            const float *wb[2];
            int wstr[2];
            for (int nt = 0; nt < 2; ++nt) {
                const int col = nt * 16 + m;
                wb[nt] = nullptr;
                wstr[nt] = 0;
                if (col < Cout * kk) {
                    const int ch = col / kk;
                    wb[nt] = R.conv1_w + (long long)ch * Cin * kk + (col - ch * kk);
                    wstr[nt] = kk;
                } else if (col < ncol) {
                    wb[nt] = R.down_w + (long long)(col - Cout * kk) * Cin;
                    wstr[nt] = 1;
                }
            }
            // Passes of 64 positions.  A call with several passes (the expansion's A children, or
            // several calls per workgroup) gives them to CS wave sets working side by side, each
            // set splitting the channels KS = kTW / CS ways: one round of barriers for CS passes.
            const int npass = (P + 63) >> 6;
            const int CS = npass >= 3 ? 4 : npass, KS = kTW / CS;
            const int cs = wave % CS, ks = wave / CS;
            for (int pb0 = 0; pb0 < P; pb0 += 64 * CS) {
                MZ_TT(1)
                const int pb = pb0 + 64 * cs;
                const int p4 = pb + 4 * m;
                const bool live = p4 < P;  // P % 4 == 0: the four positions are live together
                const int r = live ? p4 / HW : 0, sp = live ? p4 - r * HW : 0;
                const float *xr = live ? src_row(r) : nullptr, *ar = live ? add_row(r) : nullptr;
                if (xr) xr += sp;
                if (ar) ar += sp;
                f32x4 acc[4][2];
                for (int t = 0; t < 4; ++t)
                    for (int nt = 0; nt < 2; ++nt) acc[t][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                constexpr int UN = DEPTH;  // k-steps whose loads are in flight together
                if (pb < P)
                for (int k0 = ks; k0 < nsteps; k0 += KS * UN) {
                    f32x4 xv[UN], av[UN];
                    float wv[UN][2];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int ci = 4 * (k0 + u * KS) + q;
                        const bool ok = ci < Cin;
                        xv[u] = av[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        if (ok && xr) xv[u] = *reinterpret_cast<const f32x4 *>(xr + (long long)ci * HW);
                        if (ok && ar) av[u] = *reinterpret_cast<const f32x4 *>(ar + (long long)ci * HW);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            wv[u][nt] = (ok && wb[nt]) ? wb[nt][(long long)ci * wstr[nt]] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        f32x4 z = xv[u];
                        if (ar) {
                            z.x = fmaxf(z.x + av[u].x, 0.0f);
                            z.y = fmaxf(z.y + av[u].y, 0.0f);
                            z.z = fmaxf(z.z + av[u].z, 0.0f);
                            z.w = fmaxf(z.w + av[u].w, 0.0f);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[t], wv[u][nt], acc[t][nt], 0, 0, 0);
                    }
                }
                MZ_TT(2)
                // element i of tile (t, nt) in lane (n, mq) is position 16 mq + 4 i + t, column 16 nt + n.
                // The upper half of a set's channel slices parks its tiles, the lower half adds its own
                // (slice j + slice j + KS/2), then all threads sum the KS/2 rows of each pass in order.
                const int lim = pb < P ? min(64, (P - pb + 15) & ~15) : 0;  // positions of this pass that are stored
                const int pl = G.part_ld;
                lds_f *mine = part + ((((ks % (KS / 2)) * CS + cs) * kG1Cols + m) * pl + 16 * q);
                __syncthreads();  // the previous round's partials have been summed
                if (ks >= KS / 2 && 16 * q < lim) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            *reinterpret_cast<lds_f4 *>(mine + nt * 16 * pl + 4 * i) =
                                f32x4{acc[0][nt][i], acc[1][nt][i], acc[2][nt][i], acc[3][nt][i]};
                }
                __syncthreads();
                if (ks < KS / 2 && 16 * q < lim) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 o = *reinterpret_cast<const lds_f4 *>(mine + nt * 16 * pl + 4 * i);
                            *reinterpret_cast<lds_f4 *>(mine + nt * 16 * pl + 4 * i) =
                                f32x4{acc[0][nt][i] + o.x, acc[1][nt][i] + o.y, acc[2][nt][i] + o.z, acc[3][nt][i] + o.w};
                        }
                }
                __syncthreads();
                for (int v = tid; v < CS * kG1Cols * 16; v += kTT) {  // per pass: 32 columns x 16 vectors
                    const int c = v / (kG1Cols * 16), col = (v >> 4) & (kG1Cols - 1), c4 = (v & 15) << 2;
                    const int pbc = pb0 + 64 * c;
                    if (pbc < P && c4 < min(64, (P - pbc + 15) & ~15)) {
                        f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
                        for (int j = 0; j < KS / 2; ++j) {
                            const f32x4 w4 = *reinterpret_cast<const lds_f4 *>(part + ((j * CS + c) * kG1Cols + col) * pl + c4);
                            sum.x += w4.x;
                            sum.y += w4.y;
                            sum.z += w4.z;
                            sum.w += w4.w;
                        }
                        *reinterpret_cast<lds_f4 *>(Yt + col * G.y_ld + pbc + c4) = sum;
                    }
                }
            }
            __syncthreads();
            for (int e = tid; e < Cout * P; e += kTT) {
                const int ch = e / P, p = e - ch * P;
                const int r = p / HW, sp = p - r * HW, py = sp / G.W, px = sp - py * G.W;
                // tap t of channel ch is row ch * kk + t of the product
                float v = window_sum_k(Yt + ch * kk * G.y_ld + r * HW, G.y_ld, nullptr, k, G.H, G.W, py, px);
                if (R.conv1_bias) v += R.conv1_bias[ch * HW + sp];
                mid[e] = v;
                if (down) aux[e] = Yt[(Cout * kk + ch) * G.y_ld + p];
            }
            MZ_TT(3)
        } else
        for (int co0 = 0; co0 < Cout; co0 += COB) {
            for (int pb = 0; pb < P; pb += pstride) {
                const int slot = pwi * 64 + lane, p = pb + slot;
                const bool live = p < P;
                const int pc = live ? p : 0;
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accd = {0.0f, 0.0f, 0.0f, 0.0f};
                const int ks1 = slices_for(from_global ? min(G.stage_c, Cin) : Cin);
                if (from_global) {
                    for (int cb = 0; cb < Cin; cb += G.stage_c) {
                        const int cn = min(G.stage_c, Cin - cb);
                        __syncthreads();  // the previous pass has been consumed
                        MZ_TT(1)
                        load_input(stage, cb, cn);
                        const lds_f *wdn = load_weights(R.conv1_w, R.down_w, co0, Cout, Cin, cb, cn, k * k);
                        __syncthreads();
                        MZ_TT(2)
                        const int per = (cn + ks1 - 1) / ks1;
                        const int lo = cb + min(cn, ksi * per), hi = cb + min(cn, (ksi + 1) * per);
                        if (ksi >= ks1) continue;
                        if (down) conv_dispatch<true>(stage, cb, lo, hi, wst, wdn, k, G.H, G.W, P, pc, live, acc, accd);
                        else conv_dispatch<false>(stage, cb, lo, hi, wst, wdn, k, G.H, G.W, P, pc, live, acc, accd);
                    }
                } else {
                    __syncthreads();
                    const lds_f *wdn = load_weights(R.conv1_w, R.down_w, co0, Cout, Cin, 0, Cin, k * k);
                    __syncthreads();
                    const int per = (Cin + ks1 - 1) / ks1;
                    const int lo = min(Cin, ksi * per), hi = min(Cin, (ksi + 1) * per);
                    if (ksi < ks1) {
                        if (down) conv_dispatch<true>(cur, 0, lo, hi, wst, wdn, k, G.H, G.W, P, pc, live, acc, accd);
                        else conv_dispatch<false>(cur, 0, lo, hi, wst, wdn, k, G.H, G.W, P, pc, live, acc, accd);
                    }
                }
                MZ_TT(3)
                if (ks1 > 1) {
                    __syncthreads();  // earlier partials have been read
                    put_partials(acc, 0, slot);
                    if (down) put_partials(accd, 1, slot);
                    __syncthreads();
                    reduce_store(mid, 0, ks1, co0, Cout, pb, R.conv1_bias);
                    if (down) reduce_store(aux, 1, ks1, co0, Cout, pb, nullptr);
                } else if (ksi == 0) {
                    store_acc(mid, acc, co0, Cout, pb + pwi * 64, R.conv1_bias);
                    if (down) store_acc(aux, accd, co0, Cout, pb + pwi * 64, nullptr);
                }
            }
        }
        if (small && tid < npar) par[tid] = pv;
        __syncthreads();
        MZ_TT(4)
        // statistics of block b: bn1, bn2, the skip path's BatchNorm -- 2 * calls * Cout floats each
        float *st1 = stats ? stats + stat_off : nullptr;
        float *st2 = stats ? st1 + 2LL * calls * Cout : nullptr, *std_ = stats ? st1 + 4LL * calls * Cout : nullptr;
        const int call0 = blockIdx.x * (P / G.seg);
        stat_off += 6LL * calls * Cout;
        bn_act(BnJob{mid, R.bn1_w, R.bn1_b, pg, pg ? pg + Cout : nullptr, R.eps1, R.activation, st1},
               BnJob{down ? aux : nullptr, R.dbn_w, R.dbn_b, pg ? pg + 4 * Cout : nullptr,
                     pg ? pg + 5 * Cout : nullptr, R.epsd, MZ_ACT_NONE, std_}, Cout, P, G.seg, call0, calls);
        __syncthreads();
        MZ_TT(5)
        // ---- conv2: mid -> dst.  `cur` is free unless it still holds the identity skip.
        lds_f *dst = (!from_global && !down) ? aux : cur;
        if (small) {
            // few channels: a thread per output value, weights from LDS
            const int kk = k * k;
            for (int e = tid; e < Cout * P; e += kTT) {
                const int co = e / P, p = e - co * P;
                const int r = p / HW, sp = p - r * HW, py = sp / G.W, px = sp - py * G.W;
                float v = 0.0f;
                for (int ci = 0; ci < Cout; ++ci)
                    v += window_sum_k(mid + ci * P + r * HW, 0, par + (co * Cout + ci) * kk, k, G.H, G.W, py, px);
                dst[e] = v;
            }
        } else
        for (int co0 = 0; co0 < Cout; co0 += COB) {
            for (int pb = 0; pb < P; pb += pstride) {
                const int slot = pwi * 64 + lane, p = pb + slot;
                const bool live = p < P;
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accd = {0.0f, 0.0f, 0.0f, 0.0f};
                const int ks2 = slices_for(Cout);
                const int per = (Cout + ks2 - 1) / ks2;
                const int lo = min(Cout, ksi * per), hi = min(Cout, (ksi + 1) * per);
                __syncthreads();
                load_weights(R.conv2_w, nullptr, co0, Cout, Cout, 0, Cout, k * k);
                __syncthreads();
                if (ksi < ks2)
                    conv_dispatch<false>(mid, 0, lo, hi, wst, wst, k, G.H, G.W, P, live ? p : 0, live, acc, accd);
                if (ks2 > 1) {
                    __syncthreads();
                    put_partials(acc, 0, slot);
                    __syncthreads();
                    reduce_store(dst, 0, ks2, co0, Cout, pb, nullptr);
                } else if (ksi == 0) {
                    store_acc(dst, acc, co0, Cout, pb + pwi * 64, nullptr);
                }
            }
        }
        __syncthreads();
        MZ_TT(6)
        bn_act(BnJob{dst, R.bn2_w, R.bn2_b, pg ? pg + 2 * Cout : nullptr, pg ? pg + 3 * Cout : nullptr, R.eps2,
                     MZ_ACT_NONE, st2},
               BnJob{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, MZ_ACT_NONE, nullptr}, Cout, P, G.seg, call0, calls);
        __syncthreads();
        MZ_TT(7)
        // ---- skip connection + activation -> cur (input of the next block)
        if (from_global && !down) {
            // identity skip from the tower input: stage it (Cin == Cout <= cmax fits `mid`, now free)
            load_input(mid, 0, Cout);
            __syncthreads();
        }
        const lds_f *skip = down ? aux : (from_global ? mid : cur);
        for (int i = tid; i < Cout * P; i += kTT) cur[i] = activate(dst[i] + skip[i], R.activation);
        __syncthreads();
    }

    // ---- res_final (1x1 convolution with bias back to C channels) and the store: a wave per
    // output channel (wave-uniform weights), lanes along the positions
    MZ_TT(8)
    const int Cl = T.n_blocks ? T.block[T.n_blocks - 1].out_channels : G.C;
    const int wv = tid >> 6;
    // the last block's output (the input of res_final), same row scatter as y
    if (t_out != nullptr && T.n_blocks) {
        for (int i = tid; i < Cl * P; i += kTT) {
            const int ci = i / P, p = i - ci * P;
            const int r = p / HW, s = p - r * HW;
            const long long dr = y_rows ? (long long)y_rows[r0 + r] : (long long)(r0 + r);
            if (dr >= 0) t_out[dr * ld_t + ci * HW + s] = cur[i];
        }
    }
    if (y == nullptr) return;
    // res_final's weights and bias, one row of `fs` floats per output channel (bias last)
    const int fs = Cl <= 3 ? 4 : Cl + 1;
    lds_f *wfin = stage;
    if (T.final_w != nullptr) {
        for (int i = tid; i < G.C * fs; i += kTT) {
            const int c = i / fs, j = i - c * fs;
            float v = 0.0f;
            if (j < Cl) v = T.final_w[c * Cl + j];
            else if (j == fs - 1 && T.final_b) v = T.final_b[c];
            wfin[i] = v;
        }
        __syncthreads();
    }
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        const bool live = p < P;
        const int pc = live ? p : 0;
        const int r = pc / HW, s = pc - r * HW;
        const long long dr = y_rows ? (long long)y_rows[r0 + r] : (long long)(r0 + r);
        if (!live || dr < 0) continue;
        float *yr = y + dr * ldy + s;
        if (T.final_w != nullptr) {
            if (Cl <= 3) {
                // the shipped towers end in 3 channels: the position's values stay in registers
                const float i0 = cur[p], i1 = Cl > 1 ? cur[P + p] : 0.0f, i2 = Cl > 2 ? cur[2 * P + p] : 0.0f;
#pragma unroll 4
                for (int c = wv; c < G.C; c += kTW) {
                    const f32x4 q = *reinterpret_cast<const lds_f4 *>(wfin + c * 4);
                    yr[c * HW] = fmaf(q.z, i2, fmaf(q.y, i1, fmaf(q.x, i0, q.w)));
                }
            } else {
                for (int c = wv; c < G.C; c += kTW) {
                    float v = wfin[c * fs + fs - 1];
                    for (int ci = 0; ci < Cl; ++ci) v = fmaf(wfin[c * fs + ci], cur[ci * P + p], v);
                    yr[c * HW] = v;
                }
            }
        } else if (T.n_blocks) {
            for (int c = wv; c < G.C; c += kTW) yr[c * HW] = cur[c * P + p];
        } else {
            const float *xr = src_row(r), *ar = add_row(r);
            for (int c = wv; c < G.C; c += kTW) {
                float v = xr ? xr[c * HW + s] : 0.0f;
                if (ar) v = fmaxf(v + ar[c * HW + s], 0.0f);
                yr[c * HW] = v;
            }
        }
    }
    MZ_TT(9)
}

// Two builds of the tower.  Up to one workgroup per compute unit each call is a chain of memory
// round trips: 4 k-steps of loads in flight per wave (8 were no faster, 2 slower).  With more
// workgroups than compute units the second build is held to 128 vector registers so that two
// workgroups share a unit and hide each other's round trips; there 2 k-steps in flight are best.
__global__ void __launch_bounds__(kTT)
k_tower(mz_tower T, TowerGeom G, const float *__restrict__ x, long long ldx, const int32_t *__restrict__ x_rows,
        const float *__restrict__ row_add, long long ld_add, const int32_t *__restrict__ row_sel,
        float *__restrict__ y, long long ldy, const int32_t *__restrict__ y_rows, float *__restrict__ t_out,
        long long ld_t, int group, float *__restrict__ stats, int calls) {
    tower_body<4>(T, G, x, ldx, x_rows, row_add, ld_add, row_sel, y, ldy, y_rows, t_out, ld_t, group, stats, calls);
}
__global__ void __launch_bounds__(kTT) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_tower_packed(mz_tower T, TowerGeom G, const float *__restrict__ x, long long ldx,
               const int32_t *__restrict__ x_rows, const float *__restrict__ row_add, long long ld_add,
               const int32_t *__restrict__ row_sel, float *__restrict__ y, long long ldy,
               const int32_t *__restrict__ y_rows, float *__restrict__ t_out, long long ld_t, int group,
               float *__restrict__ stats, int calls) {
    tower_body<2>(T, G, x, ldx, x_rows, row_add, ld_add, row_sel, y, ldy, y_rows, t_out, ld_t, group, stats, calls);
}

// ------------------------------------------------------------------ thin towers
// One residual block on at most 4 channels (f evaluated on the bottleneck values of g's last block:
// 3 -> 3 channels with pre-multiplied weights; the tail of g once its 256 -> 3 convolution has been
// taken as a matrix product): a few thousand multiply-adds per call.  The general kernel spends a
// workgroup and a dozen barriers on that; here ONE WAVE owns a call -- its positions sit in the lanes
// (up to 4 per lane), every reduction is a wave reduction, and there is no workgroup barrier at all.
// The kernel is bound by the instructions a wave issues, so it is written to issue few:
//   * activations go through a private slice of LDS position-major, the (at most 4) channels of a
//     position in one 16-byte cell, followed by one cell of zeros: a tap that falls outside the image
//     names the zero cell, so the nine reads of a 3 x 3 window are unconditional 16-byte reads that
//     serve every channel (no per-tap exec masking, 9 reads instead of 27-36 per window);
//   * weights and BatchNorm parameters are wave-uniform: read through the scalar cache straight into
//     scalar operands of the multiply-adds (nothing staged in LDS);
//   * the float64 tail of a BatchNorm (two divisions, a square root, a reciprocal) is evaluated once
//     for all channels of a layer pair, channel c in lane c, instead of once per channel in every
//     lane; the results return to scalar registers by lane reads.
// Four calls per 256-thread workgroup.
//
// FROM_Y: the input rows are not the block's input but the products
//   Y[position][column] = sum_ci W[ci][column] x[ci][position],  column = (output channel, tap) pairs
// of conv1 followed by the skip convolution's channels (32 floats per position): conv1's output is
// the sum over taps of Y shifted by the tap's offset.
constexpr int kThinMaxC = 4, kThinMaxPos = 4;   // channels; positions per lane (P <= 256)
constexpr int kThinT = 256;
constexpr int kYCols = 32;
// LDS floats of one wave (= one call): one activation buffer of P + 1 four-channel cells for conv2's input, a
// second one for the block's input unless the rows are products (FROM_Y), which take (P + 1) x 32 floats instead.
inline size_t thin_wave_floats(bool from_y, int P) {
    const size_t cells = (size_t)(P + 1) * kThinMaxC;
    return from_y ? cells + (size_t)(P + 1) * kYCols : 2 * cells;
}
// Workgroups of four waves while four calls' LDS lets four workgroups share a compute unit, else one wave
// per workgroup (no workgroup barrier anywhere, so the size is free): the expansion's 180 positions x 32
// products are 26 KB per call.
inline int thin_threads(size_t wave_floats) { return 4 * wave_floats * sizeof(float) <= 40 * 1024 ? kThinT : 64; }

// What the thin kernel needs of a block: a handful of pointers and numbers (passing the whole mz_tower
// would make every field a separate load from the argument segment, each with its own wait).
struct ThinBlock {
    const float *conv1_w, *conv2_w, *down_w, *conv1_bias;
    const float *bn[6];  // bn1 weight, bias, bn2 weight, bias, skip BatchNorm weight, bias
    float eps1, eps2, epsd;
    int cin, cout, act, H, W;
};

// BatchNorm2d statistics of up to 8 channels of a call at once.  The values sit in LDS as 4-channel cells
// (position-major); lane l adds up channel l >> 3 over the positions (l & 7), (l & 7) + 8, ... (deviations
// from the channel's value at position 0 and their squares, float64), an exchange over each group of 8 lanes
// (two quad permutations and a half-row mirror on the DPP path) leaves every lane of a group with the
// channel's totals, and all lanes evaluate the float64 tail of their channel together; scale / shift return as
// wave-uniform values by lane reads.  A few dozen instructions per layer, against a wave-wide reduction
// per channel and sum.
// src: this lane's channel in cell 0 (any valid address for lanes without a channel: `on` false);
// mean_dst / var_dst: per-lane addresses for the channel's mean and biased variance (NULL: no export).
template <int N, int NP>
__device__ __forceinline__ void thin_bn_lanes(const lds_f *src, bool on, int P, float ga, float be, float ep,
                                              float *mean_dst, float *var_dst, float (&scale)[N], float (&shift)[N]) {
    static_assert(N <= 8, "eight groups of eight lanes");
    const int j = threadIdx.x & 7;
    const float pv = src[0];
    double S = 0.0, Q = 0.0;
#pragma unroll
    for (int i = 0; i < 8 * NP; ++i) {
        if (8 * i >= P) break;  // wave-uniform
        const int pp = j + 8 * i;
        const float x = src[4 * min(pp, P - 1)];
        const double d = pp < P ? (double)x - (double)pv : 0.0;
        S += d;
        Q += d * d;
    }
    S = dpp_add<0xB1>(S);   // quad_perm [1, 0, 3, 2]
    Q = dpp_add<0xB1>(Q);
    S = dpp_add<0x4E>(S);   // quad_perm [2, 3, 0, 1]
    Q = dpp_add<0x4E>(Q);
    S = dpp_add<0x141>(S);  // row_half_mirror: the other quad of the group of 8
    Q = dpp_add<0x141>(Q);
    const double ds = S / P, mean = (double)pv + ds;
    const double var = fmax(Q / P - ds * ds, 0.0);
    const double invstd = 1.0 / sqrt(var + (double)ep);
    const float sc = (float)((double)ga * invstd), sh = (float)((double)be - mean * (double)ga * invstd);
    if (on && mean_dst != nullptr) {
        *mean_dst = (float)mean;
        *var_dst = (float)var;
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        scale[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), 8 * c));
        shift[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sh), 8 * c));
    }
}

// CI / CO: compile-time bounds of the channel counts (the shipped towers: exactly 3 -> 3, EXACT; other
// shapes run with CI = CO = 4 and guards), so that the per-channel values live in registers and every
// loop unrolls.  NP: position slots per lane the build carries (P <= 64 NP); calls of at most 64 positions
// -- every call but the expansion's -- run the NP = 1 build, a third of the instructions and registers.
template <bool FROM_Y, int K, int CI, int CO, bool EXACT, int NP = kThinMaxPos, bool SPLIT = false>
__global__ void __launch_bounds__(kThinT)
k_tower_thin(ThinBlock R, const float *__restrict__ x, long long ldx, const int32_t *__restrict__ x_rows,
             float *__restrict__ t_out, long long ld_t, const int32_t *__restrict__ y_rows, int group, int calls,
             float *__restrict__ stats) {
    extern __shared__ __align__(16) float smem[];
    constexpr int KK = K * K, PAD = (K - 1) / 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = R.H, W = R.W, HW = H * W, P = group * HW;
    const int Cin = EXACT ? CI : R.cin, Cout = EXACT ? CO : R.cout;
    const bool down = R.down_w != nullptr;
    // SPLIT (NP = 1): the waves of a workgroup share ONE call, wave w takes positions 64 w .. 64 w + 63 -- the
    // expansion's calls of 180 positions as three short waves instead of one long one (their LDS, 26 KB per call,
    // lets six calls share a compute unit either way); the waves meet at workgroup barriers
    static_assert(!SPLIT || NP == 1, "one position slot per wave");
    const int call = SPLIT ? (int)blockIdx.x : blockIdx.x * ((int)blockDim.x >> 6) + wave;
    if (call >= calls) return;  // workgroup-uniform with SPLIT; no workgroup barrier anywhere without
    const int cells = (P + 1) * kThinMaxC;
    lds_f *a_mid = to_lds(smem) + (SPLIT ? 0 : wave) * (FROM_Y ? cells + (P + 1) * kYCols : 2 * cells), *a_in = a_mid + cells;  // [P + 1][4] each
    const int r0 = call * group;
    const int npos = SPLIT ? 1 : (P + 63) >> 6;  // position slots in use (wave-uniform)
    const int lane_p = lane + (SPLIT ? 64 * wave : 0);  // this lane's first position
    auto sync = [] {  // between writing LDS and reading what other lanes wrote
        if constexpr (SPLIT) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    cst_f *w1 = as_constant(R.conv1_w), *w2 = as_constant(R.conv2_w), *wd = as_constant(R.down_w);

    // this lane's positions: row of the call, position in the image, the cell each tap reads (P: the zero cell)
    int prow[NP], psp[NP], cell[NP][KK];
    bool live[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        live[u] = u < npos && lane_p + 64 * u < P;
        const int p = min(lane_p + 64 * u, P - 1);
        prow[u] = p / HW;
        psp[u] = p - prow[u] * HW;
        const int py = psp[u] / W, px = psp[u] - py * W;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int yy = py + tap / K - PAD, xx = px + tap % K - PAD;
            cell[u][tap] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p + (tap / K - PAD) * W + (tap % K - PAD) : P;
        }
    }
    // K x K windows of one position slot for every output channel: src holds 4-channel cells, weights
    // w[(co * n_ci + ci) * KK + tap]; acc[co] continues the sum (input channels outermost, taps innermost)
    auto windows = [&](const lds_f *src, cst_f *w, int n_ci, int u, float (&acc)[CO]) {
        f32x4 a[KK];
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) a[tap] = *reinterpret_cast<const lds_f4 *>(src + 4 * cell[u][tap]);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            if (!EXACT && co >= Cout) break;
            // one output channel's weights at a time in scalar registers (81 at once would spill them)
            __builtin_amdgcn_sched_barrier(0);
            float v = acc[co];
#pragma unroll
            for (int ci = 0; ci < kThinMaxC; ++ci) {
                if (ci >= n_ci) break;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) v = fmaf(w[(co * n_ci + ci) * KK + tap], a[tap][ci], v);
            }
            acc[co] = v;
        }
    };

    float c1[CO][NP], sk[CO][NP];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int u = 0; u < NP; ++u) c1[c][u] = sk[c][u] = 0.0f;
    // destination rows of the result, requested now
    long long drow[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u)
        drow[u] = live[u] ? (y_rows ? (long long)y_rows[r0 + prow[u]] : (long long)(r0 + prow[u])) : -1;
    if (lane < (FROM_Y ? 1 : 2) * kThinMaxC) a_mid[(lane < kThinMaxC ? 0 : cells - kThinMaxC) + P * kThinMaxC + lane] = 0.0f;  // the zero cells

    if constexpr (FROM_Y) {
        // conv1[co][p] = sum_tap Y[p + offset(tap)][co * KK + tap]; skip[co][p] = Y[p][Cout * KK + co].
        // The call's rows of Y are contiguous (P x 32 floats): copied to LDS as whole 16-byte pieces and
        // followed by a row of zeros, then every lane picks its windows from there.
        lds_f *ys = a_mid + cells;
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(x + (long long)r0 * ldx);
            const int n4 = P * (kYCols / 4);
            const int nt = SPLIT ? (int)blockDim.x : 64;  // threads that copy this call's rows
            for (int i0 = SPLIT ? (int)threadIdx.x : lane; i0 < n4; i0 += nt * 8) {  // eight pieces per lane in flight
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + nt * j < n4) v[j] = src[i0 + nt * j];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + nt * j < n4) *reinterpret_cast<lds_f4 *>(ys + 4 * (i0 + nt * j)) = v[j];
            }
            if (lane < kYCols) ys[P * kYCols + lane] = 0.0f;
        }
        sync();
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (u >= npos) break;
            const lds_f *yc = ys + (prow[u] * HW + psp[u]) * kYCols;
            const lds_f *yt[KK];
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) yt[tap] = ys + cell[u][tap] * kYCols;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                if (!EXACT && co >= Cout) break;
                float v = 0.0f;
#pragma unroll
                for (int tap = 0; tap < KK; ++tap) v += yt[tap][co * KK + tap];
                c1[co][u] = v;
                sk[co][u] = yc[Cout * KK + co];
            }
        }
    } else {
        // the block's input -> LDS cells (every value requested before the first is used), then conv1 and the skip path
        float xin[kThinMaxC][NP];
        const float *xr[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            xr[u] = nullptr;
            if (live[u]) {
                const long long s = x_rows ? (long long)x_rows[r0 + prow[u]] : (long long)(r0 + prow[u]);
                if (s >= 0) xr[u] = x + s * ldx + psp[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int ci = 0; ci < kThinMaxC; ++ci)
                xin[ci][u] = (ci < CI && xr[u] && (EXACT || ci < Cin)) ? xr[u][ci * HW] : 0.0f;
        if (R.conv1_bias)
#pragma unroll
            for (int co = 0; co < CO; ++co)
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    if ((EXACT || co < Cout) && live[u]) c1[co][u] = R.conv1_bias[co * HW + psp[u]];
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (live[u])
                *reinterpret_cast<lds_f4 *>(a_in + 4 * (lane_p + 64 * u)) = f32x4{xin[0][u], xin[1][u], xin[2][u], xin[3][u]};
        sync();
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            if (u >= npos) break;
            float acc[CO];
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = c1[co][u];
            windows(a_in, w1, Cin, u, acc);
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                if (!EXACT && co >= Cout) break;
                c1[co][u] = acc[co];
                float sv = 0.0f;
                if (down) {
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci)
                        if (EXACT || ci < Cin) sv = fmaf(wd[co * Cin + ci], xin[ci][u], sv);
                } else {
                    sv = xin[co < kThinMaxC ? co : 0][u];  // identity skip (Cin == Cout)
                }
                sk[co][u] = sv;
            }
        }
    }

    // BatchNorm2d over the call's P values of a channel (mean / biased variance in float64 around a pivot, as
    // the general kernel).  bn1 and the skip path's BatchNorm share one pass (2 * CO channels): the values go to
    // LDS cells first (conv1's output where conv2's input will be, the skip path's over the block's input, which
    // is in registers by now), see thin_bn_lanes.
    auto cell4 = [&](const float (&v)[CO][NP], int u) {
        return f32x4{v[0][u], CO > 1 ? v[CO > 1 ? 1 : 0][u] : 0.0f, CO > 2 ? v[CO > 2 ? 2 : 0][u] : 0.0f, CO > 3 ? v[CO > 3 ? 3 : 0][u] : 0.0f};
    };
    lds_f *a_skip = FROM_Y ? a_mid + cells : a_in;  // P cells: the products / the block's input are not read again
    float *st1 = stats, *st2 = stats ? stats + 2LL * calls * Cout : nullptr, *st3 = stats ? stats + 4LL * calls * Cout : nullptr;
    const int sch = lane >> 3;  // the channel this lane adds up
    {
        constexpr int N = 2 * CO;
        const bool is_skip = sch >= CO;
        const int comp = is_skip ? sch - CO : sch;
        const bool on = sch < N && comp < Cout && (!is_skip || down);
        float ga = 0.0f, be = 0.0f;
        if (on) {
            ga = R.bn[is_skip ? 4 : 0][comp];
            be = R.bn[is_skip ? 5 : 1][comp];
        }
        sync();  // every lane has taken its windows
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (live[u]) {
                *reinterpret_cast<lds_f4 *>(a_mid + 4 * (lane_p + 64 * u)) = cell4(c1, u);
                if (down) *reinterpret_cast<lds_f4 *>(a_skip + 4 * (lane_p + 64 * u)) = cell4(sk, u);
            }
        sync();
        float *mean_dst = nullptr, *var_dst = nullptr;
        if (stats != nullptr && on && (lane & 7) == 0 && (!SPLIT || wave == 0)) {
            float *st = is_skip ? st3 : st1;
            mean_dst = st + (long long)call * Cout + comp;
            var_dst = st + (long long)(calls + call) * Cout + comp;
        }
        float scale[N], shift[N];
        thin_bn_lanes<N, SPLIT ? kThinMaxPos : NP>((is_skip ? a_skip : a_mid) + (on ? comp : 0), on, P, ga, be, is_skip ? R.epsd : R.eps1, mean_dst,
                             var_dst, scale, shift);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            if (!EXACT && co >= Cout) break;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                c1[co][u] = activate(c1[co][u] * scale[co] + shift[co], R.act);
                if (down) sk[co][u] = sk[co][u] * scale[CO + co] + shift[CO + co];
            }
        }
    }
    if constexpr (SPLIT) __syncthreads();  // the other waves have read conv1's raw output
#pragma unroll
    for (int u = 0; u < NP; ++u)
        if (live[u])
            *reinterpret_cast<lds_f4 *>(a_mid + 4 * (lane_p + 64 * u)) =
                f32x4{c1[0][u], CO > 1 ? c1[CO > 1 ? 1 : 0][u] : 0.0f, CO > 2 ? c1[CO > 2 ? 2 : 0][u] : 0.0f, CO > 3 ? c1[CO > 3 ? 3 : 0][u] : 0.0f};
    sync();
    // conv2 (Cout -> Cout), its BatchNorm, the skip sum and the activation
    float c2[CO][NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.0f;
        if (u < npos) windows(a_mid, w2, Cout, u, acc);
#pragma unroll
        for (int co = 0; co < CO; ++co) c2[co][u] = acc[co];
    }
    {
        const bool on = sch < CO && sch < Cout;
        float ga = 0.0f, be = 0.0f;
        if (on) {
            ga = R.bn[2][sch];
            be = R.bn[3][sch];
        }
        sync();  // every lane has read its windows of conv2's input: its cells take conv2's output
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (live[u]) *reinterpret_cast<lds_f4 *>(a_mid + 4 * (lane_p + 64 * u)) = cell4(c2, u);
        sync();
        float *mean_dst = nullptr, *var_dst = nullptr;
        if (stats != nullptr && on && (lane & 7) == 0 && (!SPLIT || wave == 0)) {
            mean_dst = st2 + (long long)call * Cout + sch;
            var_dst = st2 + (long long)(calls + call) * Cout + sch;
        }
        float scale[CO], shift[CO];
        thin_bn_lanes<CO, SPLIT ? kThinMaxPos : NP>(a_mid + (on ? sch : 0), on, P, ga, be, R.eps2, mean_dst, var_dst, scale, shift);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            if (!EXACT && co >= Cout) break;
#pragma unroll
            for (int u = 0; u < NP; ++u)
                if (live[u] && drow[u] >= 0)
                    t_out[drow[u] * ld_t + co * HW + psp[u]] = activate((c2[co][u] * scale[co] + shift[co]) + sk[co][u], R.act);
        }
    }
}

inline ThinBlock thin_block(const mz_tower *T) {
    const mz_resblock &R = T->block[0];
    ThinBlock b;
    b.conv1_w = R.conv1_w;
    b.conv2_w = R.conv2_w;
    b.down_w = R.down_w;
    b.conv1_bias = R.conv1_bias;
    b.bn[0] = R.bn1_w;
    b.bn[1] = R.bn1_b;
    b.bn[2] = R.bn2_w;
    b.bn[3] = R.bn2_b;
    b.bn[4] = R.dbn_w;
    b.bn[5] = R.dbn_b;
    b.eps1 = R.eps1;
    b.eps2 = R.eps2;
    b.epsd = R.epsd;
    b.cin = R.in_channels;
    b.cout = R.out_channels;
    b.act = R.activation;
    b.H = T->H;
    b.W = T->W;
    return b;
}

// ------------------------------------------------------------------ fc_merge + first convolution of g in one pass
// g on the bottleneck values t of its input latent (towers.py): z = Wm t is the fc_merge product (C*H*W
// values per row), x = relu(z + table[action]) the tower's input, and the tower's first block starts with
//   Y[position][column] = sum_c Wg[c][column] x[c][position]   (conv1's (channel, tap) pairs + the skip
// convolution: 32 columns).  Writing z to memory and reading it back is most of the traffic of that
// path; here a workgroup owns 64 rows and ONE position: it forms z for that position's C = 256 channels
// (rows of Wm permuted position-major on the host), turns it into x and multiplies by Wg at once -- only
// Y, 32 floats per position, leaves the chip.  With `action` NULL every row is expanded with all A actions
// (the expansion: z is formed once, x and Y per action).
// Both products on v_mfma_f32_16x16x4_f32 with the packed tiles of mz_wide_pack.  The merge product is
// formed TRANSPOSED (weights as the A operand, the rows of t as B): lane l then holds
// z[row l & 15][channel 4 (l >> 4) + i], which is exactly the A-operand layout of the convolution product
// over those channels -- x never leaves the registers.  A wave owns 32 channels: its convolution product is
// a partial sum over them for all 64 rows x 32 columns; the eight partials meet in LDS and are added in
// wave order.
constexpr int kMcT = 512, kMcC = 256;  // rows per workgroup: 16 * MTR (64, or 32 when there are few rows)
constexpr int kMcLdP = kYCols + 4;  // row stride of a partial product in LDS: four rows apart land 16 banks apart
constexpr int kMcMaxKT = 8;         // merge product: at most 128 bottleneck values per row
constexpr size_t mc_lds(int rows) { return ((size_t)(kMcT / 64) * rows * kMcLdP + rows) * sizeof(float); }

// Two workgroups share a compute unit (128 registers, 74 KB of LDS each): while one waits at a barrier or
// for its rows, the other's waves fill the matrix pipe.
template <int MTR>
__global__ void __launch_bounds__(kMcT) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_merge_conv(const float *__restrict__ blob, long long wm_off, int KTm, int Kt, long long wg_off,
             const float *__restrict__ t, long long ld_t, const int32_t *__restrict__ t_rows, int B,
             const float *__restrict__ table, const int32_t *__restrict__ action, int A, int nA, int HW,
             float *__restrict__ Y, int skip) {
    extern __shared__ __align__(16) float smem[];
    constexpr int kMcRows = 16 * MTR, TPR = 8;  // rows per workgroup; threads that stage one row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int pos = blockIdx.y, r0 = blockIdx.x * kMcRows;
    const int ldT = KTm * 16 + 4;
    // the partial products take the place of the bottleneck rows once z is formed
    lds_f *Ts = to_lds(smem), *Ps = Ts;
    lds_i *s_act = to_lds_i(smem + (size_t)(kMcT / 64) * kMcRows * kMcLdP);
    const f32x4 *wm = reinterpret_cast<const f32x4 *>(blob + wm_off) + ((long long)(pos * (kMcC / 16) + 2 * wave) * KTm) * 64 + lane;
    const f32x4 *wg = reinterpret_cast<const f32x4 *>(blob + wg_off) + (long long)(2 * wave) * 64 + lane;
    // the first two k-tiles of this wave's merge weights are on their way while the rows are staged
    f32x4 b0[2] = {wm[0], wm[(long long)KTm * 64]}, b1[2] = {b0[0], b0[1]};
    if (KTm > 1) {
        b1[0] = wm[64];
        b1[1] = wm[(long long)(KTm + 1) * 64];
    }
    {
        // TPR threads per row: the row index is fetched once, then the row's values are requested together
        const int r = tid / TPR, k0 = tid % TPR;
        long long row = -1;
        if (r < kMcRows && r0 + r < B) row = t_rows ? (long long)t_rows[r0 + r] : (long long)(r0 + r);
        constexpr int NV = 16 * kMcMaxKT / TPR;  // values per thread
        float v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = k0 + TPR * j;
            v[j] = (row >= 0 && k < Kt && !(skip & 8)) ? t[row * ld_t + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = k0 + TPR * j;
            if (k < KTm * 16 && r < kMcRows) Ts[r * ldT + k] = v[j];
        }
    }
    if (tid < kMcRows) {
        int a = 0;
        if (action && r0 + tid < B) a = min(max(action[r0 + tid], 0), A - 1);
        s_act[tid] = a;
    }
    // this wave's convolution weights: channels 32 w .. 32 w + 31 (k-tiles 2 w, 2 w + 1) x both column tiles
    f32x4 g[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) g[n][c2] = wg[(long long)(c2 * (kMcC / 16) + n) * 64];
    __syncthreads();

    // ---- z for this position, transposed: wave w owns channels 32 w .. 32 w + 31 (two tiles), all four row tiles;
    //      acc[mt][n][i] = z[row 16 mt + m][channel 32 w + 16 n + 4 kq + i]
    f32x4 acc[MTR][2];
#pragma unroll
    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[mt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < KTm; ++kt) {
        const f32x4 bc[2] = {b0[0], b0[1]};
        b0[0] = b1[0];
        b0[1] = b1[1];
        if (kt + 2 < KTm) {
            b1[0] = wm[(long long)(kt + 2) * 64];
            b1[1] = wm[(long long)(KTm + kt + 2) * 64];
        }
        f32x4 a4[MTR];
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) a4[mt] = *reinterpret_cast<const lds_f4 *>(Ts + (mt * 16 + m) * ldT + kt * 16 + 4 * kq);
        if (!(skip & 1))
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[n][j], a4[mt][j], acc[mt][n], 0, 0, 0);
    }
    int act_m[MTR] = {};
    if (action)
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) act_m[mt] = s_act[mt * 16 + m];
    __syncthreads();  // every wave has read its rows of t
    // ---- per action: x = relu(z + table[action]) in registers, this wave's part of Y = x Wg, partials added through LDS
    const int rrow = tid >> 3, rc4 = tid & 7;  // the output piece this thread adds up: row, four columns
    for (int ai = 0; ai < nA; ++ai) {
        f32x4 y[MTR][2];
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) y[mt][0] = y[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 *trow = reinterpret_cast<const f32x4 *>(table + ((long long)ai * HW + pos) * kMcC + 32 * wave + 4 * kq);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            // the table values of this lane's four channels: per row tile where rows carry their own action
            f32x4 tv[MTR];
            if (action) {
#pragma unroll
                for (int mt = 0; mt < MTR; ++mt)
                    tv[mt] = *reinterpret_cast<const f32x4 *>(table + ((long long)act_m[mt] * HW + pos) * kMcC + 32 * wave + 16 * n + 4 * kq);
            } else {
                tv[0] = trow[4 * n];
#pragma unroll
                for (int mt = 1; mt < MTR; ++mt) tv[mt] = tv[0];
            }
            f32x4 x[MTR];
#pragma unroll
            for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) x[mt][i] = fmaxf(acc[mt][n][i] + tv[mt][i], 0.0f);
            if (!(skip & 2))
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2)
                            y[mt][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][j], g[n][c2][j], y[mt][c2], 0, 0, 0);
        }
        // y[mt][c2][i] = partial Y[row 16 mt + 4 kq + i][column 16 c2 + m]
        lds_f *pw = Ps + (size_t)wave * kMcRows * kMcLdP;
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[(mt * 16 + 4 * kq + i) * kMcLdP + 16 * c2 + m] = y[mt][c2][i];
        __syncthreads();
        if (tid < kMcRows * 8) {
            f32x4 sum = *reinterpret_cast<const lds_f4 *>(Ps + rrow * kMcLdP + 4 * rc4);
#pragma unroll
            for (int w = 1; w < kMcT / 64; ++w) {
                const f32x4 v = *reinterpret_cast<const lds_f4 *>(Ps + ((size_t)w * kMcRows + rrow) * kMcLdP + 4 * rc4);
                sum.x += v.x;
                sum.y += v.y;
                sum.z += v.z;
                sum.w += v.w;
            }
            const int rg = r0 + rrow;
            if (rg < B) *reinterpret_cast<f32x4 *>(Y + (((long long)rg * nA + ai) * HW + pos) * kYCols + 4 * rc4) = sum;
        }
        if (ai + 1 < nA) __syncthreads();
    }
}

// The same pass at the fp16 matrix rate (v_mfma_f32_16x16x32_f16: 16 times the rate of the fp32 instruction) with
// float32 accuracy.  A float32 value is held as two fp16 terms, hi = fp16(v) and lo = fp16((v - hi) * 2^11)
// (mz_mlp_device.h: v = hi + lo * 2^-11 to 2^-22 |v|), and a product is, up to the lo * lo term of relative size 2^-22,
//     2^11 x w = x_hi w_hs + x_hi w_lo + x_lo w_hi      with w_hs = w_hi * 2^11 (an exact exponent shift, made from the
// w_hi operand in registers):  three instructions into ONE accumulator cover 32 reduction indices where the fp32
// form needs eight -- 5.3 times less matrix time; the accumulator is scaled back by 2^-11 (exact) when it is read.
// The packed float32 weights are split once per weight update (mz_packed_split3), the rows of t once while they are
// staged (two planes in LDS, a lane's eight values of a pair of k-tiles in one 16-byte piece per plane),
// x = relu(z + table) in registers: its eight channels per lane are exactly one instruction's operand.
// Operands must stay below 65504 in magnitude (weights below 32): beyond that the results are NaN.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
constexpr int kMcPlanes = 2;
// four values -> their two planes, two words (four fp16) each
__device__ __forceinline__ void split2w(const f32x4 &v, u32x2 &hi, u32x2 &lo) {
    f32x2 h, l;
    split4(v, h, l);
    hi = __builtin_bit_cast(u32x2, h);
    lo = __builtin_bit_cast(u32x2, l);
}
__device__ __forceinline__ h8 as_h8(const u32x2 &a, const u32x2 &b) { return __builtin_bit_cast(h8, u32x4{a.x, a.y, b.x, b.y}); }
__device__ __forceinline__ h8 scaled_hi(const h8 &hi) { return hi * (_Float16)kLoScale; }
// A packed float32 layer (tiles [column tile][k-tile][lane] of float4, mz_wide_pack) -> its two fp16 planes as
// matrix-instruction operands: out[((column tile * pairs + pair) * 2 + plane) * 64 + lane] = the lane's four values of
// k-tile 2 pair followed by those of k-tile 2 pair + 1 (zeros past the last tile).
__global__ void k_packed_split3(const float *__restrict__ w, int KT, int NT, u32x4 *__restrict__ out) {
    const int npairs = (KT + 1) >> 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)NT * npairs * 64) return;
    const int lane = (int)(i & 63), pr = (int)((i >> 6) % npairs), nt = (int)((i >> 6) / npairs);
    const f32x4 *src = reinterpret_cast<const f32x4 *>(w) + ((long long)nt * KT + 2 * pr) * 64 + lane;
    const f32x4 a = src[0], b = 2 * pr + 1 < KT ? src[64] : f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 pa[kMcPlanes], pb[kMcPlanes];
    split2w(a, pa[0], pa[1]);
    split2w(b, pb[0], pb[1]);
#pragma unroll
    for (int pl = 0; pl < kMcPlanes; ++pl)
        out[(((long long)nt * npairs + pr) * kMcPlanes + pl) * 64 + lane] = u32x4{pa[pl].x, pa[pl].y, pb[pl].x, pb[pl].y};
}

// Products of operands given as two planes each (w: the weights, with ws = scaled_hi(w[0]); x: the activations): the
// three terms, cross terms first.  Two accumulators at once (alternating, so that no instruction waits for the one
// before it): acc0 += w0 x, acc1 += w1 x  (weights as the first operand) or acc0 += x w0, acc1 += x w1 (as the second).
__device__ __forceinline__ void mfma3_weights_first(const h8 (&w0)[2], const h8 &w0s, const h8 (&w1)[2], const h8 &w1s,
                                                    const h8 (&x)[2], f32x4 &acc0, f32x4 &acc1) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[0], x[1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[0], x[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[1], x[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[1], x[0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0s, x[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1s, x[0], acc1, 0, 0, 0);
}
__device__ __forceinline__ void mfma3_weights_second(const h8 (&x)[2], const h8 (&w0)[2], const h8 &w0s, const h8 (&w1)[2],
                                                     const h8 &w1s, f32x4 &acc0, f32x4 &acc1) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[1], w0[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[1], w1[0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[0], w0[1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[0], w1[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[0], w0s, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[0], w1s, acc1, 0, 0, 0);
}

constexpr int kMcPairs = kMcMaxKT / 2;          // pairs of k-tiles of the merge product
constexpr int kMcLdT3 = kMcPairs * 16 + 4;      // words per row and plane of the split rows of t (skewed)
static_assert(kMcPlanes * kMcLdT3 <= (kMcT / 64) * kMcLdP, "the split rows fit the area of the partial products");

template <int MTR>
__global__ void __launch_bounds__(kMcT) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_merge_conv_bf(const u32x4 *__restrict__ merge3, int KTm, int Kt, const u32x4 *__restrict__ conv3,
                const float *__restrict__ t, long long ld_t, const int32_t *__restrict__ t_rows, int B,
                const float *__restrict__ table, const int32_t *__restrict__ action, int A, int nA, int HW,
                float *__restrict__ Y) {
    extern __shared__ __align__(16) float smem[];
    constexpr int kMcRows = 16 * MTR, TPR = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int pos = blockIdx.y, r0 = blockIdx.x * kMcRows;
    lds_f *Ps = to_lds(smem);
    lds_i *T3 = to_lds_i(smem);  // [plane][row][kMcLdT3]: piece (pair, kq) of a row = its eight values of that pair
    lds_i *s_act = to_lds_i(smem + (size_t)(kMcT / 64) * kMcRows * kMcLdP);
    // weights pre-split by mz_packed_split3: [column tile][pair of k-tiles][plane][lane] 16-byte operands
    const int npairs = (KTm + 1) >> 1;
    const u32x4 *wm = merge3 + ((long long)(pos * (kMcC / 16) + 2 * wave) * npairs) * kMcPlanes * 64 + lane;
    const u32x4 *wg = conv3 + (long long)wave * kMcPlanes * 64 + lane;  // pair w of the 256 channels' 8 pairs
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the first pair of k-tiles of this wave's merge weights (both column tiles) is on its way while the rows are
    // staged (PF = 2, two pairs ahead, spills registers at every tile size and is slower: measured)
    constexpr int PF = 1;
    u32x4 wn[PF][2][kMcPlanes];  // [stage][column tile][plane] of the pairs to come
#pragma unroll
    for (int st = 0; st < PF; ++st)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int pl = 0; pl < kMcPlanes; ++pl)
                wn[st][n][pl] = st < npairs ? wm[(((long long)n * npairs + st) * kMcPlanes + pl) * 64] : u32x4{0u, 0u, 0u, 0u};
    {
        // TPR threads per row; a thread takes the pieces (four consecutive values) k0, k0 + TPR, ... of its row,
        // splits them and stores each plane's two words into the 16-byte piece of its (pair, lane group)
        const int r = tid / TPR, k0 = tid % TPR;
        long long row = -1;
        if (r < kMcRows && r0 + r < B) row = t_rows ? (long long)t_rows[r0 + r] : (long long)(r0 + r);
        constexpr int NQ = 4 * kMcMaxKT / TPR;  // pieces per thread: every piece of every pair is written (zeros past K)
        f32x4 v[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int k = 4 * (k0 + TPR * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[j][i] = (row >= 0 && k + i < Kt) ? t[row * ld_t + k + i] : 0.0f;
        }
        if (r < kMcRows)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = k0 + TPR * j, kt = q >> 2, g4 = q & 3;  // piece -> k-tile, lane group
                u32x2 p[kMcPlanes];
                split2w(v[j], p[0], p[1]);
#pragma unroll
                for (int pl = 0; pl < kMcPlanes; ++pl)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(
                        T3 + ((size_t)pl * kMcRows + r) * kMcLdT3 + ((kt >> 1) * 4 + g4) * 4 + (kt & 1) * 2) = p[pl];
            }
    }
    if (tid < kMcRows) {
        int a = 0;
        if (action && r0 + tid < B) a = min(max(action[r0 + tid], 0), A - 1);
        s_act[tid] = a;
    }
    // this wave's convolution weights: channels 32 w .. 32 w + 31 (k-tiles 2 w, 2 w + 1) x both column tiles
    h8 g3[2][kMcPlanes], gs[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
        for (int pl = 0; pl < kMcPlanes; ++pl) g3[c2][pl] = __builtin_bit_cast(h8, wg[((long long)c2 * (kMcC / 32) * kMcPlanes + pl) * 64]);
        gs[c2] = scaled_hi(g3[c2][0]);
    }
    __syncthreads();

    // ---- z for this position, transposed (weights as the A operand): acc[mt][n][i] = z[row 16 mt + m][channel 32 w + 16 n + 4 kq + i]
    f32x4 acc[MTR][2];
#pragma unroll
    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[mt][n] = zero4;
    for (int kt = 0, pr = 0; kt < KTm; kt += 2, ++pr) {
        h8 w3[2][kMcPlanes], ws[2];  // this pair's weights as operands: [column tile][plane], and the scaled hi planes
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int pl = 0; pl < kMcPlanes; ++pl) {
                w3[n][pl] = __builtin_bit_cast(h8, wn[0][n][pl]);
                if (PF > 1) wn[0][n][pl] = wn[PF - 1][n][pl];
                if (pr + PF < npairs) wn[PF - 1][n][pl] = wm[(((long long)n * npairs + pr + PF) * kMcPlanes + pl) * 64];
            }
            ws[n] = scaled_hi(w3[n][0]);
        }
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) {
            h8 t3[kMcPlanes];
#pragma unroll
            for (int pl = 0; pl < kMcPlanes; ++pl)
                t3[pl] = __builtin_bit_cast(h8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(
                                                    T3 + ((size_t)pl * kMcRows + mt * 16 + m) * kMcLdT3 + (pr * 4 + kq) * 4));
            mfma3_weights_first(w3[0], ws[0], w3[1], ws[1], t3, acc[mt][0], acc[mt][1]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[mt][n] *= kLoUnscale;  // (exact: a power of two)
    // the table values of the first action are requested before the barrier (expansion); rows with their own action
    // read theirs where they are used
    const float *tab = table + (long long)pos * kMcC + 32 * wave + 4 * kq;  // + action * HW * C + 16 n
    f32x4 tvn[2] = {zero4, zero4};
    int act_m[MTR] = {};
    if (action) {
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) act_m[mt] = s_act[mt * 16 + m];
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n) tvn[n] = *reinterpret_cast<const f32x4 *>(tab + 16 * n);
    }
    __syncthreads();  // every wave has read its rows of t
    // ---- per action: x = relu(z + table[action]) in registers, this wave's part of Y = x Wg, partials added through LDS
    const int rrow = tid >> 3, rc4 = tid & 7;
    for (int ai = 0; ai < nA; ++ai) {
        f32x4 y[MTR][2];
        const f32x4 tv0[2] = {tvn[0], tvn[1]};
        if (!action && ai + 1 < nA)  // the next action's values are on their way during this one
#pragma unroll
            for (int n = 0; n < 2; ++n) tvn[n] = *reinterpret_cast<const f32x4 *>(tab + (long long)(ai + 1) * HW * kMcC + 16 * n);
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt) {
            f32x4 x[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 tv = action ? *reinterpret_cast<const f32x4 *>(tab + (long long)act_m[mt] * HW * kMcC + 16 * n) : tv0[n];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[n][i] = fmaxf(acc[mt][n][i] + tv[i], 0.0f);
            }
            u32x2 pa[kMcPlanes], pb[kMcPlanes];
            split2w(x[0], pa[0], pa[1]);
            split2w(x[1], pb[0], pb[1]);
            h8 x3[kMcPlanes];
#pragma unroll
            for (int pl = 0; pl < kMcPlanes; ++pl) x3[pl] = as_h8(pa[pl], pb[pl]);
            y[mt][0] = y[mt][1] = zero4;
            mfma3_weights_second(x3, g3[0], gs[0], g3[1], gs[1], y[mt][0], y[mt][1]);
        }
        // y[mt][c2][i] = partial Y[row 16 mt + 4 kq + i][column 16 c2 + m]
        lds_f *pw = Ps + (size_t)wave * kMcRows * kMcLdP;
#pragma unroll
        for (int mt = 0; mt < MTR; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[(mt * 16 + 4 * kq + i) * kMcLdP + 16 * c2 + m] = y[mt][c2][i] * kLoUnscale;
        __syncthreads();
        if (tid < kMcRows * 8) {
            f32x4 sum = *reinterpret_cast<const lds_f4 *>(Ps + rrow * kMcLdP + 4 * rc4);
#pragma unroll
            for (int w = 1; w < kMcT / 64; ++w) {
                const f32x4 v = *reinterpret_cast<const lds_f4 *>(Ps + ((size_t)w * kMcRows + rrow) * kMcLdP + 4 * rc4);
                sum.x += v.x;
                sum.y += v.y;
                sum.z += v.z;
                sum.w += v.w;
            }
            const int rg = r0 + rrow;
            if (rg < B) *reinterpret_cast<f32x4 *>(Y + (((long long)rg * nA + ai) * HW + pos) * kYCols + 4 * rc4) = sum;
        }
        if (ai + 1 < nA) __syncthreads();
    }
}

int tower_geometry(const mz_tower *T, int group, int ng, bool vec_ok, TowerGeom *g, size_t *lds_bytes) {
    if (!T || T->n_blocks < 0 || T->n_blocks > MZ_MAX_BLOCKS || T->C <= 0 || T->H <= 0 || T->W <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_tower_forward: bad tower (n_blocks 0..%d, positive C, H, W)", MZ_MAX_BLOCKS);
    if (group <= 0) return mz::fail(MZ_ERR_ARG, "mz_tower_forward: group must be positive");
    std::memset(g, 0, sizeof(*g));
    g->C = T->C;
    g->H = T->H;
    g->W = T->W;
    g->HW = T->H * T->W;
    g->group0 = group;
    g->seg = group * g->HW;
    g->P = ng * g->seg;  // ng reference calls per workgroup
    if (T->n_blocks && g->seg < 2)
        // torch.nn.BatchNorm2d: "Expected more than 1 value per channel when training"
        return mz::fail(MZ_ERR_ARG, "mz_tower_forward: batch statistics need more than 1 value per channel");
    int cin = T->C;
    for (int b = 0; b < T->n_blocks; ++b) {
        const mz_resblock &R = T->block[b];
        if (R.in_channels != cin || R.out_channels <= 0 || R.kernel_size <= 0 || !(R.kernel_size & 1))
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block %d: in_channels %d (expected %d), out_channels %d, "
                            "kernel_size %d (must be odd)", b, R.in_channels, cin, R.out_channels, R.kernel_size);
        if (!R.conv1_w || !R.bn1_w || !R.bn1_b || !R.conv2_w || !R.bn2_w || !R.bn2_b)
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block %d has a NULL weight", b);
        if ((R.in_channels != R.out_channels && !R.down_w) || (R.down_w && (!R.dbn_w || !R.dbn_b)))
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block %d: a 1x1 skip convolution with its BatchNorm is "
                            "needed when the channel count changes", b);
        if (R.activation < MZ_ACT_NONE || R.activation > MZ_ACT_SIGMOID)
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block %d has unsupported activation %d", b, R.activation);
        g->cmax = std::max(g->cmax, R.out_channels);
        cin = R.out_channels;
    }
    if (T->final_w == nullptr && cin != T->C)
        return mz::fail(MZ_ERR_ARG, "mz_tower_forward: the last block leaves %d channels, res_final is needed for C=%d",
                        cin, T->C);
    const int slots = (g->P + 63) / 64;
    g->pw = 1;  // waves across the positions: a power of two
    while (g->pw < slots && g->pw < kTW) g->pw <<= 1;
    g->ks = kTW / g->pw;
    auto up4 = [](long long v) { return (v + 3) & ~3LL; };
    int kkmax = 1;
    for (int b = 0; b < T->n_blocks; ++b) kkmax = std::max(kkmax, T->block[b].kernel_size * T->block[b].kernel_size);
    const long long wpc = (long long)(kkmax + 1) * COB;  // staged weight floats per input channel
    const long long buf = up4((long long)std::max(1, g->cmax) * g->P);
    const long long lds_floats = (long long)(kMaxLds / sizeof(float));
    const int cl = T->n_blocks ? T->block[T->n_blocks - 1].out_channels : T->C;
    const long long fin = T->final_w ? (long long)T->C * (cl <= 3 ? 4 : cl + 1) : 0;
    // the slice partials are needed by every convolution that runs on the sliced path with more than
    // 8 input channels: any first convolution but a matrix-product one, second ones of blocks whose
    // parameters do not fit the small-parameter area
    auto small_block = [](const mz_resblock &R) {
        const int kk = R.kernel_size * R.kernel_size;
        return R.out_channels * R.out_channels * kk + 6 * R.out_channels <= kPar && R.out_channels * kk <= 128;
    };
    auto red_floats = [&](bool gemm_first) -> long long {
        bool need = false;
        for (int b = 0; b < T->n_blocks; ++b) {
            const mz_resblock &R = T->block[b];
            if (!(b == 0 && gemm_first) && R.in_channels > 8) need = true;
            if (!small_block(R) && R.out_channels > 8) need = true;
        }
        return need && g->ks > 1 ? (long long)g->ks * 2 * COB * g->pw * 64 : 0;
    };
    // the first block as a matrix product: its columns fit two 16-wide tiles, rows are read as
    // 16-byte vectors, and the per-wave partials plus the product fit next to the buffers
    static const bool gemm_off = [] {
        const char *e = getenv("MZ_TOWER_GEMM1");  // tuning aid: 0 keeps the staged form
        return e && atoi(e) == 0;
    }();
    if (T->n_blocks && vec_ok && !gemm_off) {
        const mz_resblock &R = T->block[0];
        const int ncol = R.out_channels * R.kernel_size * R.kernel_size + (R.down_w ? R.out_channels : 0);
        const long long p16 = (g->P + 15) & ~15LL;
        const long long part_ld = std::min<long long>(64, p16) + 4, y_ld = p16 + 4;
        const long long need = (long long)(kTW / 2) * kG1Cols * part_ld + kG1Cols * y_ld;
        const long long later = (long long)g->cmax * wpc;  // staged weights of the later convolutions
        const long long red = red_floats(true), fixed = 3 * buf + red + kPar;
        if (ncol <= kG1Cols && fixed + std::max(std::max(need, later), fin) <= lds_floats) {
            g->gemm1 = 1;
            g->y_ld = (int)y_ld;
            g->part_ld = (int)part_ld;
            g->off_a = 0;
            g->off_b = (int)buf;
            g->off_c = (int)(2 * buf);
            g->off_red = (int)(3 * buf);
            g->off_par = (int)(3 * buf + red);
            g->off_stage = g->off_w = (int)fixed;
            g->stage_c = T->C;
            *lds_bytes = (size_t)(fixed + std::max(std::max(need, later), fin)) * sizeof(float);
            return MZ_OK;
        }
    }
    const long long red = red_floats(false), fixed = 3 * buf + red + kPar;
    // the first block's input is staged `stage_c` channels at a time next to their weights
    int sc = 0;
    for (int c = T->C; c >= 1; --c)
        if (fixed + up4((long long)c * g->P) + (long long)std::max(c, g->cmax) * wpc <= lds_floats) {
            sc = c;
            break;
        }
    if (!T->n_blocks) sc = 1;
    if (sc == 0 || fixed + fin > lds_floats)
        return mz::fail(MZ_ERR_ARG, "mz_tower_forward: %d channels x %d positions per call exceed the LDS-resident "
                        "tower", g->cmax, g->P);
    g->off_a = 0;
    g->off_b = (int)buf;
    g->off_c = (int)(2 * buf);
    g->off_red = (int)(3 * buf);
    g->off_par = (int)(3 * buf + red);
    g->off_stage = (int)fixed;
    g->stage_c = sc;
    g->off_w = (int)(fixed + up4((long long)sc * g->P));
    const long long stage_floats = T->n_blocks ? up4((long long)sc * g->P) + (long long)std::max(sc, g->cmax) * wpc : 0;
    *lds_bytes = (size_t)(fixed + std::max(stage_floats, fin)) * sizeof(float);
    return MZ_OK;
}

// ------------------------------------------------------------------ policy post-processing
__global__ void __launch_bounds__(kWT)
k_policy_post(const float *__restrict__ in, int B, int A, int times, float *__restrict__ out,
              int32_t *__restrict__ argmax) {
    const int r = blockIdx.x * kWT + threadIdx.x;
    if (r >= B) return;
    const float *p = in + (long long)r * A;
    if (argmax) {
        int best = 0;
        for (int i = 1; i < A; ++i)
            if (p[i] > p[best]) best = i;
        argmax[r] = best;
    }
    if (out == nullptr) return;
    float *q = out + (long long)r * A;
    if (q != p)
        for (int i = 0; i < A; ++i) q[i] = p[i];
    for (int t = 0; t < times; ++t) {
        float mx = q[0];
        for (int i = 1; i < A; ++i) mx = fmaxf(mx, q[i]);
        float s = 0.0f;
        for (int i = 0; i < A; ++i) {
            const float e = expf(q[i] - mx);
            q[i] = e;
            s += e;
        }
        for (int i = 0; i < A; ++i) q[i] = q[i] / s;
    }
}

}  // namespace

extern "C" {

int64_t mz_wide_workspace(const mz_packed_layer *h_plan, int32_t B) {
    if (!h_plan || B < 0) return mz::fail(MZ_ERR_ARG, "mz_wide_workspace: NULL plan or negative B");
    if (h_plan->Kp <= 0 || h_plan->Np <= 0) return mz::fail(MZ_ERR_ARG, "mz_wide_workspace: the plan is empty (mz_wide_plan failed?)");
    if (B == 0) return 0;
    const WideShape s = wide_shape(*h_plan, B);
    return s.splitk > 1 ? (int64_t)s.splitk * B * h_plan->Np : 0;
}

static int wide_forward_impl(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                             const int32_t *x_rows, float *y, int64_t ldy, int32_t B, float *workspace,
                             int64_t workspace_floats, void *stream, int n_seg = 0) {
    if (!h_plan || !blob || !x || !y || B < 0 || (n_seg == 0 && ldx < h_plan->K) || ldy < h_plan->N || h_plan->Kp <= 0 || h_plan->Np <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_wide_forward: NULL argument, empty plan, negative B or row stride below the row width");
    if (h_plan->act < MZ_ACT_NONE || h_plan->act > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "mz_wide_forward: activation %d is not element-wise", h_plan->act);
    if (B == 0) return MZ_OK;
    const mz_packed_layer L = *h_plan;
    const WideShape s = wide_shape(L, B);
    const int64_t need = s.splitk > 1 ? (int64_t)s.splitk * B * L.Np : 0;
    if (need > 0 && (!workspace || workspace_floats < need))
        return mz::fail(MZ_ERR_ARG, "mz_wide_forward: workspace of %lld floats needed, %lld given", (long long)need,
                        (long long)(workspace ? workspace_floats : 0));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (ldx % 4 == 0) && (L.K % 4 == 0) && L.K >= 4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const dim3 grid(s.tiles_n, s.tiles_m, s.splitk);
    const size_t lds = (size_t)2 * s.tm * kLdX * sizeof(float);
    int seg_chunks = 0;
    if (n_seg > 0) {
        const int seg_k = L.K / n_seg;
        if (!x_rows || (long long)seg_k * n_seg != L.K || seg_k % kKC != 0 || ldx < seg_k || !vec)
            return mz::fail(MZ_ERR_ARG, "mz_wide_forward_gather: K=%d must be n_seg=%d segments of a multiple of %d floats, "
                            "rows 16-byte aligned with a stride of at least one segment, and x_rows given", L.K, n_seg, kKC);
        seg_chunks = seg_k / kKC;
    }
    auto launch = [&](auto kernel) -> int {
        MZ_TRY(allow_lds(kernel, lds, "k_wide"));
        hipLaunchKernelGGL(kernel, grid, dim3(kWT), lds, st, L, blob, x, (long long)ldx, x_rows, y, (long long)ldy, B,
                           s.splitk, s.chunks_per_split, workspace, seg_chunks);
        return MZ_OK;
    };
    if (n_seg > 0) MZ_TRY(s.big ? launch(k_wide<4, 4, true, true>) : launch(k_wide<2, 2, true, true>));
    else if (s.big) MZ_TRY(vec ? launch(k_wide<4, 4, true>) : launch(k_wide<4, 4, false>));
    else MZ_TRY(vec ? launch(k_wide<2, 2, true>) : launch(k_wide<2, 2, false>));
    MZ_TRY(mz::check_launch("k_wide"));
    if (s.splitk > 1) {
        const long long total = (long long)B * L.N;
        const int nb = (int)std::min<long long>((total + kWT - 1) / kWT, 8192);
        hipLaunchKernelGGL(k_wide_finish, dim3(nb), dim3(kWT), 0, st, L, blob, workspace, s.splitk, B, y,
                           (long long)ldy);
        MZ_TRY(mz::check_launch("k_wide_finish"));
    }
    return MZ_OK;
}

int mz_wide_forward(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                    const int32_t *x_rows, float *y, int64_t ldy, int32_t B, float *workspace,
                    int64_t workspace_floats, void *stream) {
    return wide_forward_impl(h_plan, blob, x, ldx, x_rows, y, ldy, B, workspace, workspace_floats, stream);
}

int mz_wide_forward_gather(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                           const int32_t *x_rows, int32_t n_seg, float *y, int64_t ldy, int32_t B, float *workspace,
                           int64_t workspace_floats, void *stream) {
    if (n_seg <= 0) return mz::fail(MZ_ERR_ARG, "mz_wide_forward_gather: n_seg must be positive");
    return wide_forward_impl(h_plan, blob, x, ldx, x_rows, y, ldy, B, workspace, workspace_floats, stream, n_seg);
}

int64_t mz_wide_static_workspace_floats(void) { return kWideScratchFloats; }

// The built-in array's address on the current device, looked up once per device (each GPU holds its own copy).
static int wide_static_scratch(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "mz_wide_static: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices)
        return mz::fail(MZ_ERR_ARG, "mz_wide_static: device index %d is outside the %d devices the built-in scratch is tracked for",
                        dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_wide_scratch));
        if (e != hipSuccess || !ptr)
            return mz::fail(MZ_ERR_HIP, "mz_wide_static: hipGetSymbolAddress: %s (the first call on a device must not be made "
                            "while a stream of it is being captured: call mz_wide_static_init() beforehand)", hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return MZ_OK;
}

int mz_wide_static_init(void) {
    float *p = nullptr;
    return wide_static_scratch(&p);
}

int mz_wide_forward_static(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                           const int32_t *x_rows, float *y, int64_t ldy, int32_t B, void *stream) {
    if (!h_plan || B < 0 || h_plan->Kp <= 0 || h_plan->Np <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_wide_forward_static: NULL or empty plan, or negative B");
    int64_t need = 0;
    if (B > 0) {
        const WideShape s = wide_shape(*h_plan, B);
        need = s.splitk > 1 ? (int64_t)s.splitk * B * h_plan->Np : 0;
    }
    if (need > kWideScratchFloats)
        return mz::fail(MZ_ERR_ARG, "mz_wide_forward_static: this call needs %lld floats of partial sums, the built-in "
                        "scratch holds %lld (use mz_wide_forward with a caller-provided workspace)", (long long)need,
                        (long long)kWideScratchFloats);
    float *scratch = nullptr;
    if (need > 0) {
        const int rc = wide_static_scratch(&scratch);
        if (rc != MZ_OK) return rc;
    }
    return wide_forward_impl(h_plan, blob, x, ldx, x_rows, y, ldy, B, scratch, need, stream);
}

static int merge_conv_impl(const mz_packed_layer *h_merge, const mz_packed_layer *h_conv, const float *blob, const float *t,
                           int64_t ld_t, const int32_t *t_rows, const float *table, const int32_t *action, int32_t A, int32_t B,
                           int32_t C, int32_t HW, float *Y, void *stream, const uint32_t *merge3 = nullptr,
                           const uint32_t *conv3 = nullptr) {
    const bool split = merge3 != nullptr;
    if (split && !blob) blob = reinterpret_cast<const float *>(merge3);  // the float32 tiles are not read in this form
    if (!h_merge || !h_conv || !blob || !t || !table || !Y || A <= 0 || B <= 0 || HW <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward: NULL argument or non-positive size");
    if (C != kMcC) return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward: %d channels (the kernel is built for %d)", C, kMcC);
    if (h_merge->N != HW * C || h_merge->K <= 0 || h_merge->Kp > 16 * kMcMaxKT || ld_t < h_merge->K)
        return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward: the merge layer must map at most 128 values to H*W*C = %d (position-major rows)",
                        HW * C);
    if (h_conv->K != C || h_conv->N > kYCols || h_conv->Np != kYCols)
        return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward: the convolution product must map %d channels to at most %d columns", C, kYCols);
    if (HW > 65535) return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward: more than 65535 positions");
    const int KTm = h_merge->Kp >> 4;
    // Rows per workgroup.  64 with many workgroups per compute unit (the larger tile's weight reuse wins); 32 when
    // 64-row workgroups number more than the units but fewer than a few per unit (512 .. 2048 trees: two half-size
    // workgroups share a unit and hide each other's barriers and round trips, 1024 trees 3.90 -> 4.08 M sims/s);
    // with fewer 64-row workgroups than units (a few hundred trees: the launch is one latency chain per workgroup)
    // the smallest of 64 / 48 / 32 that still gives every workgroup a unit of its own.
    static const int small_lo = [] { const char *e = getenv("MZ_MC_SMALL_LO"); return e ? atoi(e) : 256; }();   // tuning aids
    static const int small_hi = [] { const char *e = getenv("MZ_MC_SMALL_HI"); return e ? atoi(e) : 1200; }();
    auto wgs = [&](int r) { return (long long)((B + r - 1) / r) * HW; };
    int rows = 64;
    if (wgs(64) > small_lo && wgs(64) < small_hi) rows = 32;
    else if (wgs(64) <= small_lo) rows = wgs(32) <= small_lo ? 32 : wgs(48) <= small_lo ? 48 : 64;
    static const int force_rows = [] { const char *e = getenv("MZ_MC_ROWS"); return e ? atoi(e) : 0; }();
    if (force_rows == 32 || force_rows == 48 || force_rows == 64) rows = force_rows;
    const size_t lds = mc_lds(rows);  // the rows of t (at most 132 floats each) fit the area of the partial products
    MZ_TRY(allow_lds(k_merge_conv<4>, mc_lds(64), "k_merge_conv"));
    MZ_TRY(allow_lds(k_merge_conv<2>, mc_lds(32), "k_merge_conv"));
    MZ_TRY(allow_lds(k_merge_conv<3>, mc_lds(48), "k_merge_conv"));
    static const int skip = [] {
        const char *e = getenv("MZ_MC_SKIP");  // timing aid: leaves out parts of the kernel (results are then meaningless)
        return e ? atoi(e) : 0;
    }();
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((B + rows - 1) / rows, HW), dim3(kMcT), lds, (hipStream_t)stream, blob,
                           (long long)h_merge->w_off, KTm, h_merge->K, (long long)h_conv->w_off, t, (long long)ld_t, t_rows, B,
                           table, action, A, action ? 1 : A, HW, Y, skip);
    };
    if (split) {
        MZ_TRY(allow_lds(k_merge_conv_bf<4>, mc_lds(64), "k_merge_conv_bf"));
        MZ_TRY(allow_lds(k_merge_conv_bf<2>, mc_lds(32), "k_merge_conv_bf"));
        MZ_TRY(allow_lds(k_merge_conv_bf<3>, mc_lds(48), "k_merge_conv_bf"));
        auto launch_bf = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((B + rows - 1) / rows, HW), dim3(kMcT), lds, (hipStream_t)stream,
                               reinterpret_cast<const u32x4 *>(merge3), KTm, h_merge->K, reinterpret_cast<const u32x4 *>(conv3), t,
                               (long long)ld_t, t_rows, B, table, action, A, action ? 1 : A, HW, Y);
        };
        if (rows == 32) launch_bf(k_merge_conv_bf<2>);
        else if (rows == 48) launch_bf(k_merge_conv_bf<3>);
        else launch_bf(k_merge_conv_bf<4>);
        return mz::check_launch("k_merge_conv_bf");
    }
    if (rows == 32) launch(k_merge_conv<2>);
    else if (rows == 48) launch(k_merge_conv<3>);
    else launch(k_merge_conv<4>);
    return mz::check_launch("k_merge_conv");
}

int mz_merge_conv_forward(const mz_packed_layer *h_merge, const mz_packed_layer *h_conv, const float *blob, const float *t,
                          int64_t ld_t, const int32_t *t_rows, const float *table, const int32_t *action, int32_t A, int32_t B,
                          int32_t C, int32_t HW, float *Y, void *stream) {
    return merge_conv_impl(h_merge, h_conv, blob, t, ld_t, t_rows, table, action, A, B, C, HW, Y, stream);
}

int mz_merge_conv_forward_split(const mz_packed_layer *h_merge, const uint32_t *merge3, const mz_packed_layer *h_conv,
                                const uint32_t *conv3, const float *t, int64_t ld_t, const int32_t *t_rows, const float *table,
                                const int32_t *action, int32_t A, int32_t B, int32_t C, int32_t HW, float *Y, void *stream) {
    if (!merge3 || !conv3) return mz::fail(MZ_ERR_ARG, "mz_merge_conv_forward_split: NULL split weights");
    return merge_conv_impl(h_merge, h_conv, nullptr, t, ld_t, t_rows, table, action, A, B, C, HW, Y, stream, merge3, conv3);
}

int64_t mz_packed_split3_words(const mz_packed_layer *h_plan) {
    if (!h_plan || h_plan->Kp <= 0 || h_plan->Np <= 0) return mz::fail(MZ_ERR_ARG, "mz_packed_split3_words: NULL or empty layer");
    return (int64_t)(h_plan->Np / 16) * (((h_plan->Kp / 16) + 1) / 2) * kMcPlanes * 64 * 4;
}

int mz_packed_split3(const mz_packed_layer *h_plan, const float *blob, uint32_t *out, void *stream) {
    if (!h_plan || !blob || !out || h_plan->Kp <= 0 || h_plan->Np <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_packed_split3: NULL argument or an empty layer");
    const int KT = h_plan->Kp / 16, NT = h_plan->Np / 16;
    const long long n = (long long)NT * ((KT + 1) / 2) * 64;
    hipLaunchKernelGGL(k_packed_split3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, blob + h_plan->w_off,
                       KT, NT, reinterpret_cast<u32x4 *>(out));
    return mz::check_launch("k_packed_split3");
}

int mz_tower_tail_forward(const mz_tower *h_tower, const float *Y, int32_t B, int32_t group, float *t_out, int64_t ld_t,
                          const int32_t *y_rows, float *bn_stats, void *stream) {
    if (!h_tower || !Y || !t_out || B <= 0 || group <= 0 || B % group)
        return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: NULL argument, or B is not a positive multiple of group");
    if (h_tower->n_blocks != 1) return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: one block expected");
    const mz_resblock &R = h_tower->block[0];
    const int hw = h_tower->H * h_tower->W, kk = R.kernel_size * R.kernel_size;
    if (R.kernel_size != 3 || !R.down_w || R.out_channels > kThinMaxC || R.out_channels * kk + R.out_channels > kYCols)
        return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: a 3x3 block with a skip convolution and at most 3 output channels expected");
    if (!R.bn1_w || !R.bn1_b || !R.conv2_w || !R.bn2_w || !R.bn2_b || !R.dbn_w || !R.dbn_b)
        return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: block 0 has a NULL weight");
    if (R.activation < MZ_ACT_NONE || R.activation > MZ_ACT_SIGMOID)
        return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: unsupported activation %d", R.activation);
    if ((long long)group * hw > 64 * kThinMaxPos || group * hw < 2)
        return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: %d positions per call (2 .. %d)", group * hw, 64 * kThinMaxPos);
    if (ld_t < (long long)R.out_channels * hw) return mz::fail(MZ_ERR_ARG, "mz_tower_tail_forward: ld_t below the block's values per row");
    const size_t wave_floats = thin_wave_floats(true, group * hw);
    const int calls = B / group, threads = thin_threads(wave_floats), per_wg = threads / 64;
    const size_t lds = per_wg * wave_floats * sizeof(float);
    const ThinBlock tb = thin_block(h_tower);
    auto launch = [&](auto kernel) {
        MZ_TRY(allow_lds(kernel, lds, "k_tower_thin"));
        hipLaunchKernelGGL(kernel, dim3((calls + per_wg - 1) / per_wg), dim3(threads), lds, (hipStream_t)stream, tb, Y,
                           (long long)hw * kYCols, nullptr, t_out, (long long)ld_t, y_rows, group, calls, bn_stats);
        return mz::check_launch("k_tower_thin");
    };
    if (R.out_channels == 3) {
        const int P = group * hw;
        if (P <= 64) return launch(k_tower_thin<true, 3, 3, 3, true, 1>);
        static const bool split = [] { const char *e = getenv("MZ_THIN_SPLIT"); return !e || atoi(e) != 0; }();  // cross-checking aid
        if (split) {  // one workgroup per call, one wave per 64 positions
            auto kernel = k_tower_thin<true, 3, 3, 3, true, 1, true>;
            const size_t lds1 = wave_floats * sizeof(float);
            MZ_TRY(allow_lds(kernel, lds1, "k_tower_thin"));
            hipLaunchKernelGGL(kernel, dim3(calls), dim3(64 * ((P + 63) / 64)), lds1, (hipStream_t)stream, tb, Y,
                               (long long)hw * kYCols, nullptr, t_out, (long long)ld_t, y_rows, group, calls, bn_stats);
            return mz::check_launch("k_tower_thin");
        }
        return P <= 192 ? launch(k_tower_thin<true, 3, 3, 3, true, 3>) : launch(k_tower_thin<true, 3, 3, 3, true>);
    }
    return launch(k_tower_thin<true, 3, 4, 4, false>);
}

// Calls per launch from which the one-wave-per-call kernel takes over thin towers.  It is as fast as the
// general kernel's workgroup per call at a few hundred calls (11-16 us against 12-14 us at 256) and several
// times faster at thousands (25 us against 38-147 us at 4096), so the default is "always"; the threshold
// remains as a knob (environment variable at load time, mz_tower_thin_min_calls at run time) so that the
// general kernel can be run on the same rows.
static int g_thin_min_calls = [] {
    const char *e = getenv("MZ_TOWER_THIN_MIN_CALLS");
    return e ? atoi(e) : 1;
}();
static int thin_min_calls() { return g_thin_min_calls; }

int mz_tower_thin_min_calls(int32_t new_value) {
    const int old = g_thin_min_calls;
    if (new_value >= 0) g_thin_min_calls = new_value;
    return old;
}

int64_t mz_tower_stats_floats(const mz_tower *h_tower, int32_t calls) {
    if (!h_tower || calls <= 0 || h_tower->n_blocks < 0 || h_tower->n_blocks > MZ_MAX_BLOCKS) return 0;
    int64_t n = 0;
    for (int b = 0; b < h_tower->n_blocks; ++b) n += 6LL * calls * h_tower->block[b].out_channels;
    return n;
}

int mz_tower_forward(const mz_tower *h_tower, const float *x, int64_t ldx, const int32_t *x_rows,
                     const float *row_add, int64_t ld_add, const int32_t *row_sel, float *y,
                     int64_t ldy, const int32_t *y_rows, float *t_out, int64_t ld_t, int32_t B, int32_t group,
                     void *stream) {
    return mz_tower_forward_stats(h_tower, x, ldx, x_rows, row_add, ld_add, row_sel, y, ldy, y_rows, t_out, ld_t, B, group,
                                  nullptr, stream);
}

int mz_tower_forward_stats(const mz_tower *h_tower, const float *x, int64_t ldx, const int32_t *x_rows,
                           const float *row_add, int64_t ld_add, const int32_t *row_sel, float *y,
                           int64_t ldy, const int32_t *y_rows, float *t_out, int64_t ld_t, int32_t B, int32_t group,
                           float *bn_stats, void *stream) {
    // A single block on a handful of channels whose caller wants the block output only: one wave per call
    if (h_tower && h_tower->n_blocks == 1 && h_tower->C <= kThinMaxC && h_tower->block[0].in_channels == h_tower->C &&
        h_tower->block[0].out_channels <= kThinMaxC && (h_tower->block[0].kernel_size == 1 || h_tower->block[0].kernel_size == 3) &&
        (h_tower->block[0].down_w || h_tower->block[0].in_channels == h_tower->block[0].out_channels) && !y && t_out && !row_add &&
        x && group > 0 && B > 0 && B % group == 0 && h_tower->H > 0 && h_tower->W > 0 &&
        (long long)group * h_tower->H * h_tower->W <= 64 * kThinMaxPos && B / group >= thin_min_calls()) {
        const mz_resblock &R = h_tower->block[0];
        const int hw = h_tower->H * h_tower->W;
        if (ldx < (long long)h_tower->C * hw || ld_t < (long long)R.out_channels * hw)
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: a row stride is below the row's %d values", h_tower->C * hw);
        if (!R.conv1_w || !R.bn1_w || !R.bn1_b || !R.conv2_w || !R.bn2_w || !R.bn2_b || (R.down_w && (!R.dbn_w || !R.dbn_b)))
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block 0 has a NULL weight");
        if (R.activation < MZ_ACT_NONE || R.activation > MZ_ACT_SIGMOID)
            return mz::fail(MZ_ERR_ARG, "mz_tower_forward: block 0 has unsupported activation %d", R.activation);
        if (group * hw < 2) return mz::fail(MZ_ERR_ARG, "mz_tower_forward: batch statistics need more than 1 value per channel");
        const size_t wave_floats = thin_wave_floats(false, group * hw);
        const int calls = B / group, threads = thin_threads(wave_floats), per_wg = threads / 64;
        const size_t lds_thin = per_wg * wave_floats * sizeof(float);
        const ThinBlock tb = thin_block(h_tower);
        auto launch_thin = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((calls + per_wg - 1) / per_wg), dim3(threads), lds_thin, (hipStream_t)stream, tb, x,
                               (long long)ldx, x_rows, t_out, (long long)ld_t, y_rows, group, calls, bn_stats);
            return mz::check_launch("k_tower_thin");
        };
        if (R.kernel_size == 3 && R.in_channels == 3 && R.out_channels == 3)
            return group * hw <= 64 ? launch_thin(k_tower_thin<false, 3, 3, 3, true, 1>) : launch_thin(k_tower_thin<false, 3, 3, 3, true>);
        return R.kernel_size == 3 ? launch_thin(k_tower_thin<false, 3, 4, 4, false>) : launch_thin(k_tower_thin<false, 1, 4, 4, false>);
    }
    TowerGeom g;
    size_t lds = 0;
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec_ok = h_tower && h_tower->H > 0 && h_tower->W > 0 && (h_tower->H * h_tower->W) % 4 == 0 &&
                        ldx % 4 == 0 && aligned(x) && (!row_add || (ld_add % 4 == 0 && aligned(row_add)));
    // With several groups per compute unit a workgroup takes up to four reference calls at once
    // (each keeps its own batch statistics): the chain of round trips and barriers is paid once for
    // them.  Kept to 192 positions per workgroup and to at least two workgroups per compute unit.
    static const int ng_cap = [] {
        const char *e = getenv("MZ_TOWER_CALLS_PER_WG");  // tuning aid
        return e ? atoi(e) : 4;
    }();
    int ng = 1;
    if (h_tower && group > 0 && B > 0 && B % group == 0 && h_tower->H > 0 && h_tower->W > 0) {
        const int groups = B / group, seg = group * h_tower->H * h_tower->W;
        for (int c = std::min(ng_cap, 4); c >= 2; --c)
            if (groups % c == 0 && groups / c >= 512 && c * seg <= 192) {
                ng = c;
                break;
            }
    }
    if (ng > 1 && tower_geometry(h_tower, group, ng, vec_ok, &g, &lds) != MZ_OK) ng = 1;
    if (ng == 1) MZ_TRY(tower_geometry(h_tower, group, 1, vec_ok, &g, &lds));
    const int D = g.C * g.HW;
    if (!x || (!y && !t_out) || B < 0 || ldx < D || (y && ldy < D) || (row_add && ld_add < D))
        return mz::fail(MZ_ERR_ARG, "mz_tower_forward: NULL argument, negative B or a row stride below C*H*W=%d", D);
    if (t_out) {
        if (!h_tower->n_blocks) return mz::fail(MZ_ERR_ARG, "mz_tower_forward: t_out needs at least one block");
        const int dt = h_tower->block[h_tower->n_blocks - 1].out_channels * g.HW;
        if (ld_t < dt) return mz::fail(MZ_ERR_ARG, "mz_tower_forward: ld_t below the last block's %d values per row", dt);
    }
    if (B % group) return mz::fail(MZ_ERR_ARG, "mz_tower_forward: B=%d is not a multiple of group=%d", B, group);
    if (B == 0) return MZ_OK;
    g.vec_ok = vec_ok;
    // more workgroups than compute units, and two of them fit a unit's LDS: the build that shares it
    static const int packed_mode = [] {
        const char *e = getenv("MZ_TOWER_PACKED");  // tuning aid: 0 never, 1 always
        return e ? atoi(e) : -1;
    }();
    const int rows_per_wg = ng * group, wgs = B / rows_per_wg;
    const bool packed = packed_mode >= 0 ? packed_mode == 1 : (wgs > 256 && 2 * (lds + 1024) <= kMaxLds);
    auto launch = [&](auto kernel, const char *name) -> int {
        MZ_TRY(allow_lds(kernel, lds, name));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(kTT), lds, (hipStream_t)stream, *h_tower, g, x,
                           (long long)ldx, x_rows, row_add, (long long)ld_add, row_sel, y, (long long)ldy, y_rows,
                           t_out, (long long)ld_t, rows_per_wg, bn_stats, B / group);
        return mz::check_launch(name);
    };
    return packed ? launch(k_tower_packed, "k_tower_packed") : launch(k_tower, "k_tower");
}

#ifdef MZ_TOWER_TIMELINE
int mz_debug_tower_stamps(unsigned long long *h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_tower_stamps), sizeof(unsigned long long) * 16) == hipSuccess
               ? MZ_OK : MZ_ERR_HIP;
}
#endif

int mz_policy_post(const float *policy_in, int32_t B, int32_t A, int32_t softmax_times, float *policy_out,
                   int32_t *argmax, void *stream) {
    if (!policy_in || B < 0 || A <= 0 || softmax_times < 0)
        return mz::fail(MZ_ERR_ARG, "mz_policy_post: NULL input, negative B or softmax_times, or A <= 0");
    if (B == 0 || (!policy_out && !argmax)) return MZ_OK;
    hipLaunchKernelGGL(k_policy_post, dim3((B + kWT - 1) / kWT), dim3(kWT), 0, (hipStream_t)stream, policy_in, B, A,
                       softmax_times, policy_out, argmax);
    return mz::check_launch("k_policy_post");
}

}  // extern "C"
