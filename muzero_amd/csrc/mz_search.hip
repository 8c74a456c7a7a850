// Fused search kernel for gfx950: one launch runs a whole MCTS.run (src/search/mcts.py:44-74)
// for G trees.  A 512-thread workgroup owns 16 trees for all S simulations -- trees never
// interact, so there is no inter-workgroup communication and no grid-wide barrier.
//
// Per round and workgroup (16 trees = one MFMA row tile):
//   select        16 lane groups walk their trees (DPP/ballot argmax, tables in LDS); the leaf
//                 latents are gathered into LDS and the child each simulation starts from is drawn
//   first layers  {policy layer 1 | merge layer W[:, :D]} of the leaf in one MFMA pass
//   then two teams of four waves run concurrently (see k_search):
//     expansion   remaining policy layers, softmax twice -> priors; child latents
//                 relu(z + table[a]) streamed to the node pool; reward head over all 16*A
//                 children with the child rows formed on the fly from z and the per-action table
//                 and its scalar output layer folded into the hidden layer's epilogue; child records
//     simulation  depth x {policy -> first-max action -> merge -> reward}, then value, the
//                 rollout latents read on the fly as relu(z + table[action])
//   backup        the recorded path is updated by independent lanes
// Weights are read from L2 in the packed MFMA layout (mz_mlp_device.h: matrix layers as two fp16
// planes per fp32 weight, three v_mfma_f32_16x16x32_f16 per product block); activations (fp32 rows or
// the split operand images the matrix layers read), priors, rewards and rollouts stay in LDS; the node pool in HBM receives exactly the records the
// stand-alone kernels would write, so both paths are interchangeable and comparable bit for bit.
//
// Compiled with -ffp-contract=off (every rounding is part of the contract).
#include <algorithm>
#include <cstdlib>

#include "mz_common.h"
#include "mz_mlp_device.h"
#include "mz_search_lds.h"
#include "mz_tree_device.h"

using namespace mzd;
using namespace mzs;

namespace {

// Diagnostic build only (-DMZ_SEARCH_DIAG, never the shipped library): stages named by the bits of the
// environment variable MZ_SEARCH_SKIP are left out, so that instruction counters and durations of two
// runs can be subtracted.  Results are meaningless with any bit set.
#ifdef MZ_SEARCH_DIAG
#define MZ_SKIP(bit) ((q.lanes_per_tree >> (8 + (bit))) & 1)
#else
#define MZ_SKIP(bit) 0
#endif

// Layers [first, n) of an MLP over the team's 16 rows, ping-ponging h1/h2 (src: the input in the form layer
// `first` reads; src_alt: its fp32 rows where src is the split image and both exist).
template <typename MLP>
__device__ __forceinline__ lds_f *mlp16(Team &tm, const MLP &M, int first, const float *__restrict__ blob, lds_f *src,
                                        lds_f *src_alt, int ld, lds_f *h1, lds_f *h2, int *out_ld, unsigned char *smem,
                                        const lds_i *narrow) {
    return mlp_tile<1>(tm, M, first, blob, src, src_alt, ld, h1, h2, out_ld, smem, narrow);
}

// After the leaf's first layers, a simulation round splits the workgroup into two teams of
// four waves (one wave per SIMD each) that run concurrently and never wait for each other:
//   expansion team   rest of the leaf's policy head -> priors; child latents to the pool; reward
//                    head over the 16*A children; child records
//   simulation team  the greedy rollout from the drawn child (which needs only the leaf's merge
//                    pre-activation and the drawn action) and the value head
// The two chains share no buffer that either writes, so a wave stalled on LDS, memory, a short
// serial step or its team's sync leaves its SIMD to the other team's wave instead of idling
// it: f32 MFMAs and vector-ALU work of one SIMD do not overlap, but waiting and issuing do.
// Inside a team the waves synchronise through an LDS arrival counter (Team::sync); the teams
// meet again at the workgroup barrier before the backup.
template <int MTC>
__global__ void __launch_bounds__(kThreads)
k_search(mz_pool p, mz_packed_nets Pk, const float *__restrict__ blob, mz_search_params q, Lds L,
         double *__restrict__ policy, float *__restrict__ policy32, double *__restrict__ value,
         int32_t *__restrict__ visits) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    lds_f *X32 = to_lds(smem_raw + L.x32);  // leaf latents, fp32 rows (as gathered)
    lds_f *X = to_lds(smem_raw + L.x);      // their split image
    lds_f *XS = X32;                        // the simulation's current latent, split image (X32 is dead by then)
    lds_f *XS32 = L.need_x32 ? to_lds(smem_raw + L.xs32) : nullptr;  // and as fp32 rows where a narrow layer reads it
    lds_f *Z = to_lds(smem_raw + L.z);
    lds_f *Zs = to_lds(smem_raw + L.zs);
    lds_f *Tab = to_lds(smem_raw + L.tab);
    lds_f *He1 = to_lds(smem_raw + L.he1);
    lds_f *He2 = to_lds(smem_raw + L.he2);
    lds_f *Hs1 = to_lds(smem_raw + L.hs1);
    lds_f *Hs2 = to_lds(smem_raw + L.hs2);
    lds_f *R = to_lds(smem_raw + L.r);      // the children's latents, split image (over X | He1 | He2)
    lds_f *R32 = to_lds(smem_raw + L.r32);
    lds_f *Ra = to_lds(smem_raw + L.ra);
    lds_f *Rb = to_lds(smem_raw + L.rb);
    lds_f *DotE = to_lds(smem_raw + L.dot);
    lds_f *DotS = to_lds(smem_raw + L.dots);
    lds_f *DotV = to_lds(smem_raw + L.dotv);
    lds_f *s_prior = to_lds(smem_raw + L.prior);
    lds_f *s_rew = to_lds(smem_raw + L.rew);
    lds_f *s_roll = to_lds(smem_raw + L.roll);
    lds_i *s_leaf = to_lds_i(smem_raw + L.ints);
    lds_i *s_child = s_leaf + kTrees;
    lds_i *s_act = s_child + kTrees;
    lds_i *s_leaf_tp = s_act + kTrees;
    lds_i *s_depth = s_leaf_tp + kTrees;
    lds_i *s_root = s_depth + kTrees;   // root to_play per tree
    lds_i *s_team = s_root + kTrees;    // [4] per team: arrival counter, error flag
    // The network plan (over a kilobyte of layer descriptors) and the narrow-layer offsets are
    // copied to LDS once and read from there where a layer is run: as kernel arguments they
    // would sit in scalar registers for the whole kernel and spill.
    typedef __attribute__((address_space(3))) const mz_packed_nets lds_nets;
    const lds_nets &P = *reinterpret_cast<lds_nets *>((lds_i *)to_lds_i(smem_raw + L.nets));
    const lds_i *s_narrow = to_lds_i(smem_raw + L.nets + sizeof(mz_packed_nets));  // [3][MZ_MAX_LAYERS]
    {
        const int words = (int)(sizeof(mz_packed_nets) / sizeof(int));
        lds_i *dst = to_lds_i(smem_raw + L.nets);
        const int *src = reinterpret_cast<const int *>(&Pk);
        if (threadIdx.x == 0) {
#pragma unroll 8
            for (int i = 0; i < words; ++i) dst[i] = src[i];
            for (int m = 0; m < 3; ++m)
                for (int i = 0; i < MZ_MAX_LAYERS; ++i) dst[words + m * MZ_MAX_LAYERS + i] = L.narrow[m][i];
        }
    }
    __syncthreads();
    int *s_path = reinterpret_cast<int *>(smem_raw + L.path);
    double *s_tab = reinterpret_cast<double *>(smem_raw + L.tabs);

    // optional cycle stamps per stage (diagnostics: q.stamps == NULL in production); slots
    // 0..15 are stamped by thread 0 (expansion team), 16..31 by the simulation team's first thread
    __shared__ unsigned long long s_stamp[32];
    unsigned long long t_prev = 0;
    const bool stamping = q.stamps != nullptr && blockIdx.x == 0;
    if (stamping && threadIdx.x < 32) s_stamp[threadIdx.x] = 0;
    if (stamping && (threadIdx.x == 0 || threadIdx.x == kThreads / 2)) t_prev = clock64();
#define MZ_STAMP(slot)                                             \
    if (stamping && (int)threadIdx.x == ((slot) < 16 ? 0 : kThreads / 2)) { \
        const unsigned long long t_now = clock64();                \
        s_stamp[slot] += t_now - t_prev;                           \
        t_prev = t_now;                                            \
    }

    const int A = Pk.A, D = Pk.D, Dp = Pk.Dp, S = q.S, depth = q.depth;
    // small facts about the heads, kept in registers; everything else about a layer is read from
    // the LDS copy where the layer runs
    const int pol_layers = Pk.policy.n_layers, val_layers = Pk.value.n_layers, rew_layers = Pk.reward.n_layers;
    const bool pol0_softmax_only = Pk.policy.layer[0].act == MZ_ACT_SOFTMAX && pol_layers == 1;
    const bool rew0_matrix = is_matrix_layer(Pk.reward.layer[0]);
    const bool grouped_first = is_matrix_layer(Pk.policy.layer[0]) && Pk.policy.layer[0].Kp == Pk.merge.layer[0].Kp;
    // the output of the policy head's first layer takes the form its reader wants (mlp_tile's rule)
    const bool pol0_split = pol_layers >= 2 && is_matrix_layer(Pk.policy.layer[1]);
    const bool val0_split = val_layers >= 2 && is_matrix_layer(Pk.value.layer[1]);
    const int Dp16 = Pk.merge.layer[0].Np;  // columns of a merge pre-activation row that were written
    const int KCd = Pk.merge.layer[0].Kp >> 5;  // k-chunks of a latent's split image
    const bool fold_value2 = val_layers == 2 && tail_foldable(Pk.value);
    const bool grouped_last = L.fold_reward && val_layers >= 2 && is_matrix_layer(Pk.value.layer[0]) &&
                              Pk.value.layer[0].Kp == Pk.reward.layer[0].Kp;
    (void)val_layers;
    const int ld_d = L.ld_d;
    const bool implicit = p.latent == nullptr;  // children kept as their parent's merge row (q.z_pool)
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * kTrees;
    const int trees = min(kTrees, p.G - g0);
    const int lpt = q.lanes_per_tree & 0xff;
    const int ld_p1 = lds_stride(Pk.policy.layer[0].N);
    const int half = uni(tid >> 6) >= kWaves / 2 ? 1 : 0;
    int team_arrived = 0;  // arrivals this wave's team counter shows after the wave's latest team sync

    // policy layer 1 -> h_out and merge pre-activation W[:, :D] x -> z_out (fp32 rows), from the same
    // input rows (xs: their split image, x32: their fp32 rows where a narrow first layer reads them);
    // ends with a team sync.  Returns the buffer that holds policy layer 1's output in the form layer 2
    // reads (h_out, or h_alt after a conversion).  (The merge layer's stored epilogue is a relu for the
    // stand-alone kernel; here its pre-activation is kept and the per-action table is added where a child
    // latent is formed.)
    auto first_layers = [&](Team &tm, const lds_f *xs, const lds_f *x32, lds_f *h_out, lds_f *h_alt, lds_f *z_out) -> lds_f * {
        const mz_packed_layer Lp0 = load_layer(P.policy.layer[0]), Lm0 = load_layer(P.merge.layer[0]);
        if (grouped_first) {
            dense_group2<1>(xs, ld_d, blob, Lm0.Kp, make_part(Lp0, h_out, ld_p1, Lp0.act, pol0_split),
                            make_part(Lm0, z_out, ld_d, MZ_ACT_NONE), tm);
            tm.sync();
            return h_out;
        }
        // a narrow first policy layer reads the fp32 rows
        const int nwo = uni(s_narrow[0]);
        const lds_f *nw = nwo >= 0 ? to_lds(smem_raw + nwo) : nullptr;
        dense_tile<1>(is_matrix_layer(Lp0) ? xs : x32, ld_d, blob, Lp0, h_out, ld_p1, pol0_split && is_matrix_layer(Lp0), nullptr, 0,
                      nullptr, nw, tm);
        dense_group1<1>(xs, ld_d, blob, Lm0.Kp, make_part(Lm0, z_out, ld_d, MZ_ACT_NONE), tm);
        lds_f *res = h_out;
        if (!is_matrix_layer(Lp0) && pol0_split) {  // fp32 rows of a narrow layer feed a matrix layer
            rows_to_split(h_out, ld_p1, h_alt, ld_p1, 1, ceil32(Lp0.N) >> 5, ceil16(Lp0.N), tm);
            res = h_alt;
        }
        tm.sync();
        return res;
    };
    // Sixteen rows relu(zsrc[t] + table[sel[t]]) -- the latents the dynamics network gives for the actions
    // sel -- as a split image in `dst` (and as fp32 rows in dst32 when not NULL); no sync.
    auto form_rows = [&](const Team &tm, const lds_f *zsrc, const lds_i *sel, lds_f *dst, lds_f *dst32) {
        for (int i = tm.tid(); i < KCd * 64; i += tm.nthreads()) {
            const int l = i & 63, c = i >> 6, t = l & 15;
            const int k0 = 32 * c + 8 * (l >> 4);
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (k0 < Dp16) {
                const lds_f *zr = zsrc + t * ld_d + k0, *tr = Tab + sel[t] * ld_d + k0;
                const f32x4 z0 = *reinterpret_cast<const lds_f4 *>(zr), z1 = *reinterpret_cast<const lds_f4 *>(zr + 4);
                const f32x4 b0 = *reinterpret_cast<const lds_f4 *>(tr), b1 = *reinterpret_cast<const lds_f4 *>(tr + 4);
                v0 = f32x4{fmaxf(z0.x + b0.x, 0.f), fmaxf(z0.y + b0.y, 0.f), fmaxf(z0.z + b0.z, 0.f), fmaxf(z0.w + b0.w, 0.f)};
                v1 = f32x4{fmaxf(z1.x + b1.x, 0.f), fmaxf(z1.y + b1.y, 0.f), fmaxf(z1.z + b1.z, 0.f), fmaxf(z1.w + b1.w, 0.f)};
            }
            f32x4 hi, lo;
            split8(v0, v1, hi, lo);
            lds_f *q = dst + c * kChunkFloats + l * 4;
            *reinterpret_cast<lds_f4 *>(q) = hi;
            *reinterpret_cast<lds_f4 *>(q + KCd * kChunkFloats) = lo;
            if (dst32 != nullptr) {
                *reinterpret_cast<lds_f4 *>(dst32 + t * ld_d + k0) = v0;
                *reinterpret_cast<lds_f4 *>(dst32 + t * ld_d + k0 + 4) = v1;
            }
        }
    };

    // ---- the expansion after the leaf's first layers (He1, Z valid) ---------------------------
    auto expand_tail = [&](Team &tm, int e, int base, lds_f *pol_in) {
        const int ttid = tm.tid(), tn = tm.nthreads();
        int ld_p = ld_p1;
        lds_f *pol = pol_in;
        if (!MZ_SKIP(6)) pol = mlp16(tm, P.policy, 1, blob, pol_in, nullptr, ld_p1, He1, He2, &ld_p, smem_raw, s_narrow);
        MZ_STAMP(4)  // expand: remaining policy layers
        if (pol0_softmax_only) softmax_rows(pol, ld_p, kTrees, A, 1, tm);
        softmax_rows(pol, ld_p, kTrees, A, 1, tm);  // expand_node's extra softmax (expansion.py:27)
        for (int i = ttid; i < kTrees * A; i += tn) {
            const int t = i / A, a = i - t * A;
            const float pr = pol[t * ld_p + a];
            s_prior[i] = pr;
            if (q.prior_trace && t < trees) q.prior_trace[((size_t)(g0 + t) * (S + 1) + e) * A + a] = pr;
        }
        tm.sync();  // the children's image below overwrites the policy head's buffers
        MZ_STAMP(5)  // expand: softmax + priors
        // child latents -> pool (coalesced along the latent).  Written once, read back at most a
        // few times much later: streaming (non-temporal) stores keep them from displacing the
        // weights and tree statistics in L2
        if (MZ_SKIP(7)) {
        } else if (implicit) {
            // one row per tree instead of A: the children's latents are relu(z + table[a]) of it
            for (int i = ttid; i < trees * D; i += tn) {
                const int t = i / D, c = i - t * D;
                q.z_pool[((size_t)(g0 + t) * (S + 1) + e) * D + c] = Z[t * ld_d + c];
            }
        } else if ((D & 3) == 0) {
            // one (tree, action) row per wave and pass, lanes along the latent: the row index is
            // wave-uniform, so the index arithmetic stays on the scalar unit
            const int q4 = D >> 2;
            const int wv = uni(ttid >> 6), lane = ttid & 63;
            const int nrows = trees * A;
            const int inv_a = (65536 + A - 1) / A;  // r / A == (r * inv_a) >> 16 for r * A < 65536
            for (int c4 = lane; c4 < q4; c4 += 64) {
                // four rows at a time: their LDS reads are in flight together
                for (int r0 = 4 * wv; r0 < nrows; r0 += 4 * tm.nwaves) {
                    f32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = min(r0 + u, nrows - 1);
                        const int t = (r * inv_a) >> 16, a = r - t * A;
                        const f32x4 z = *reinterpret_cast<const lds_f4 *>(Z + t * ld_d + 4 * c4);
                        const f32x4 b = *reinterpret_cast<const lds_f4 *>(Tab + a * ld_d + 4 * c4);
                        v[u] = f32x4{fmaxf(z.x + b.x, 0.f), fmaxf(z.y + b.y, 0.f), fmaxf(z.z + b.z, 0.f), fmaxf(z.w + b.w, 0.f)};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = r0 + u;
                        if (r < nrows) {
                            const int t = (r * inv_a) >> 16, a = r - t * A;
                            f32x4 *dst = reinterpret_cast<f32x4 *>(p.latent + ((size_t)(g0 + t) * p.cap + base + a) * D) + c4;
                            __builtin_nontemporal_store(v[u], dst);  // one 16-byte streaming store per lane
                        }
                    }
                }
            }
        } else {
            for (int i = ttid; i < trees * A * D; i += tn) {
                const int c = i % D, r = i / D;
                const int a = r % A, t = r / A;
                p.latent[((size_t)(g0 + t) * p.cap + base + a) * D + c] = fmaxf(Z[t * ld_d + c] + Tab[a * ld_d + c], 0.0f);
            }
        }
        MZ_STAMP(6)  // expand: child latents to the pool
        // reward MLP over the children, MTC actions (= MTC row tiles of 16 trees) at a time: row tile mt of a
        // pass holds the children of action a0 + mt, relu(z[t] + table[a0 + mt]), as a split image
        for (int a0 = 0; a0 < A && !MZ_SKIP(8); a0 += MTC) {
            for (int i = ttid; i < MTC * KCd * 64; i += tn) {
                const int l = i & 63, c = (i >> 6) % KCd, mt = (i >> 6) / KCd, t = l & 15;
                const int k0 = 32 * c + 8 * (l >> 4);
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (k0 < Dp16) {
                    const lds_f *zr = Z + t * ld_d + k0, *tr = Tab + (a0 + mt) * ld_d + k0;
                    const f32x4 z0 = *reinterpret_cast<const lds_f4 *>(zr), z1 = *reinterpret_cast<const lds_f4 *>(zr + 4);
                    const f32x4 b0 = *reinterpret_cast<const lds_f4 *>(tr), b1 = *reinterpret_cast<const lds_f4 *>(tr + 4);
                    v0 = f32x4{fmaxf(z0.x + b0.x, 0.f), fmaxf(z0.y + b0.y, 0.f), fmaxf(z0.z + b0.z, 0.f), fmaxf(z0.w + b0.w, 0.f)};
                    v1 = f32x4{fmaxf(z1.x + b1.x, 0.f), fmaxf(z1.y + b1.y, 0.f), fmaxf(z1.z + b1.z, 0.f), fmaxf(z1.w + b1.w, 0.f)};
                }
                f32x4 hi, lo;
                split8(v0, v1, hi, lo);
                lds_f *qd = R + mt * 16 * ld_d + c * kChunkFloats + l * 4;
                *reinterpret_cast<lds_f4 *>(qd) = hi;
                *reinterpret_cast<lds_f4 *>(qd + KCd * kChunkFloats) = lo;
                if (!rew0_matrix) {
                    *reinterpret_cast<lds_f4 *>(R32 + (mt * 16 + t) * ld_d + k0) = v0;
                    *reinterpret_cast<lds_f4 *>(R32 + (mt * 16 + t) * ld_d + k0 + 4) = v1;
                }
            }
            tm.sync();
            if (L.fold_reward) {
                // hidden layer over the children with the output layer folded into its epilogue
                const mz_packed_layer L0 = load_layer(P.reward.layer[0]), L1 = load_layer(P.reward.layer[1]);
                dense_group1<MTC>(R, ld_d, blob, L0.Kp, make_dot_part(L0, L1, DotE), tm);
                tm.sync();
                MZ_TL(6)
                MZ_STAMP(7)  // expand: reward layer 1 over 16*A children
                const float b2 = blob[L1.b_off];
                const int tiles = L0.Np >> 4;
                for (int r = ttid; r < kTrees * MTC; r += tn) {  // as finish_dot
                    float sum = 0.0f;
                    for (int k = 0; k < 4 * tiles; ++k) sum += DotE[k * (kTrees * MTC) + r];
                    s_rew[(r % kTrees) * L.tab_rows + a0 + r / kTrees] = activate(sum + b2, L1.act);
                }
                tm.sync();
                continue;
            }
            int ld;
            const lds_f *src = mlp_tile<MTC>(tm, P.reward, 0, blob, R, rew0_matrix ? nullptr : R32, ld_d, Ra, Rb, &ld, smem_raw,
                                             s_narrow + 2 * MZ_MAX_LAYERS);
            for (int i = ttid; i < kTrees * MTC; i += tn) {
                const int al = i / kTrees, t = i % kTrees;
                s_rew[t * L.tab_rows + a0 + al] = src[i * ld];
            }
            tm.sync();
        }
        MZ_STAMP(8)  // expand: remaining reward layers
        // ---- commit child records ------------------------------------------------------------
        for (int i = ttid; i < trees * A; i += tn) {
            const int t = i / A, a = i - t * A;
            const size_t o = (size_t)(g0 + t) * p.cap;
            float pr = s_prior[t * A + a];
            if (e == 0) pr = mzt::noisy_prior((double)pr, q.eta[(size_t)(g0 + t) * A + a], q.noise_frac);
            mzt::write_child(p, o, base, a, s_leaf[t], pr, s_rew[t * L.tab_rows + a], (int8_t)s_leaf_tp[t]);
            if (a == 0) p.node[o + s_leaf[t]].first_child = base;
        }
        MZ_STAMP(9)  // commit
    };

    // ---- the simulation from the drawn child (Z, s_act valid; writes XS, Zs, Hs1, Hs2) --------
    // A latent along the rollout is relu(z + table[action]) with z the merge pre-activation of
    // the previous latent (Z for the drawn child, Zs afterwards), formed as the split image the
    // matrix layers read (form_rows).
    auto simulate = [&](Team &tm, int e, int base) {
        const int ttid = tm.tid();
        for (int d = 0; d < depth; ++d) {
            form_rows(tm, d == 0 ? Z : Zs, s_act, XS, XS32);
            tm.sync();
            MZ_STAMP(16)  // simulate: next latent
            if (d > 0) {  // reward of the step just taken (src/search/simulation.py:37-38)
                int ld_o;
                const lds_f *r = mlp16(tm, P.reward, 0, blob, XS, XS32, ld_d, Hs1, Hs2, &ld_o, smem_raw, s_narrow + 2 * MZ_MAX_LAYERS);
                if (ttid < kTrees) s_roll[ttid * (depth + 1) + (d - 1)] = r[ttid * ld_o];
                tm.sync();
                MZ_STAMP(17)  // simulate: reward head
            }
            lds_f *pol_in = first_layers(tm, XS, XS32, Hs1, Hs2, Zs);
            MZ_STAMP(18)  // simulate: policy layer 1 | merge
            int ld_q;
            lds_f *pq = mlp16(tm, P.policy, 1, blob, pol_in, nullptr, ld_p1, Hs1, Hs2, &ld_q, smem_raw, s_narrow);
            MZ_STAMP(19)  // simulate: remaining policy layers
            if (pol0_softmax_only) softmax_rows(pq, ld_q, kTrees, A, 1, tm);
            if (ttid < kTrees) {  // first maximum of the network's policy (simulation.py:36)
                const lds_f *row = pq + ttid * ld_q;
                int best = 0;
                for (int a = 1; a < A; ++a)
                    if (row[a] > row[best]) best = a;
                s_act[ttid] = best;
            }
            tm.sync();
            MZ_STAMP(20)  // simulate: argmax
        }
        // the last latent: reward of the last step (depth >= 1) and value
        form_rows(tm, depth == 0 ? Z : Zs, s_act, XS, XS32);
        tm.sync();
        MZ_STAMP(16)  // simulate: next latent
        if (depth >= 1 && grouped_last) {
            const mz_packed_layer R0 = load_layer(P.reward.layer[0]), R1l = load_layer(P.reward.layer[1]),
                                  V0 = load_layer(P.value.layer[0]);
            const int ld_v = lds_stride(V0.N);
            // a two-layer value head is folded like the reward head (as every kernel does)
            const bool fold_v = fold_value2;
            const mz_packed_layer V1l = load_layer(P.value.layer[1]);
            dense_group2<1>(XS, ld_d, blob, V0.Kp,
                            fold_v ? make_dot_part(V0, V1l, DotV) : make_part(V0, Hs1, ld_v, V0.act, val0_split),
                            make_dot_part(R0, R1l, DotS), tm);
            tm.sync();
            finish_dot(DotS, R0.Np >> 4, kTrees, blob, R1l, s_roll + (depth - 1), depth + 1, tm);
            MZ_TL(6)
            MZ_STAMP(17)  // simulate: reward head | value layer 1
            if (fold_v) {
                finish_dot(DotV, V0.Np >> 4, kTrees, blob, V1l, s_roll + depth, depth + 1, tm);
            } else {
                int ld_o;
                const lds_f *v = mlp16(tm, P.value, 1, blob, Hs1, nullptr, ld_v, Hs1, Hs2, &ld_o, smem_raw, s_narrow + MZ_MAX_LAYERS);
                if (ttid < kTrees) s_roll[ttid * (depth + 1) + depth] = v[ttid * ld_o];
            }
            MZ_STAMP(21)  // simulate: value head
        } else {
            if (depth >= 1) {
                int ld_o;
                const lds_f *r = mlp16(tm, P.reward, 0, blob, XS, XS32, ld_d, Hs1, Hs2, &ld_o, smem_raw, s_narrow + 2 * MZ_MAX_LAYERS);
                if (ttid < kTrees) s_roll[ttid * (depth + 1) + (depth - 1)] = r[ttid * ld_o];
                tm.sync();
                MZ_STAMP(17)  // simulate: reward head
            }
            int ld_o;
            const lds_f *v = mlp16(tm, P.value, 0, blob, XS, XS32, ld_d, Hs1, Hs2, &ld_o, smem_raw, s_narrow + MZ_MAX_LAYERS);
            if (ttid < kTrees) s_roll[ttid * (depth + 1) + depth] = v[ttid * ld_o];
            MZ_STAMP(21)  // simulate: value head
        }
    };

    // ---- one-time staging: PUCT tables, per-action table, root records ------------------
    for (int i = tid; i < q.tab_len; i += kThreads) {
        s_tab[i] = q.sqrt_tab[i];
        s_tab[q.tab_len + i] = q.log_tab[i];
    }
    for (int i = tid; i < L.tab_rows * ld_d; i += kThreads) {
        const int a = i / ld_d, c = i - a * ld_d;
        Tab[i] = (a < A && c < Dp) ? blob[Pk.action_table_off + (size_t)a * Dp + c] : 0.0f;
    }
    {  // narrow layers (scalar heads, policy logits) keep their weights in LDS for the whole search
        const mz_packed_mlp *mlps[3] = {&Pk.policy, &Pk.value, &Pk.reward};
        for (int m = 0; m < 3; ++m)
            for (int i = 0; i < mlps[m]->n_layers; ++i)
                if (L.narrow[m][i] >= 0)
                    stage_narrow(blob, mlps[m]->layer[i], to_lds(smem_raw + L.narrow[m][i]));
    }
    for (int i = tid; i < kTrees * ld_d; i += kThreads) {
        const int t = i / ld_d, c = i - t * ld_d;
        float v = 0.0f;
        if (t < trees && c < D) {
            v = q.root_latent[(size_t)(g0 + t) * D + c];
            if (!implicit) p.latent[((size_t)(g0 + t) * p.cap) * D + c] = v;
        }
        X32[i] = v;
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
    }
    if (tid < 4) s_team[tid] = 0;
    __syncthreads();
    MZ_STAMP(0)  // staging

    // expansion e = 0 is the root expansion (no simulation / backup follows it);
    // expansion e >= 1 belongs to simulation s = e - 1
    for (int e = 0; e <= S; ++e) {
        const int base = 1 + e * A;  // pool index of this expansion's first child (lock step)

        // ---- select (simulations only) + leaf latents --------------------------------------
        if (e > 0) {
            if (!MZ_SKIP(0) && tid < kTrees * lpt) {
                const int t = tid / lpt, a = tid % lpt;
                int leaf = 0, dep = 0;
                if (t < trees) {
                    const size_t o = (size_t)(g0 + t) * p.cap;
                    // the root's children start at node 1 and the root has e - 1 completed backups
                    switch (lpt) {
                        case 2: leaf = mzt::select_leaf<2>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 4: leaf = mzt::select_leaf<4>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 8: leaf = mzt::select_leaf<8>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        case 16: leaf = mzt::select_leaf<16>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                        default: leaf = mzt::select_leaf<32>(p, o, a, s_tab, q.tab_len, true, &dep, 1, e - 1, s_path + t * kPathCap, kPathCap); break;
                    }
                    if (a == 0) {
                        // a visit count outside the tables cannot happen for tab_len >= S + 2
                        leaf = max(leaf, 0);
                        s_leaf[t] = leaf;
                        s_depth[t] = dep;
                        // to_play alternates with depth below the root (src/search/nodes.py:28)
                        s_leaf_tp[t] = (dep & 1) ? -s_root[t] : s_root[t];
                        if (q.leaf_trace) q.leaf_trace[(size_t)(g0 + t) * S + (e - 1)] = leaf;
                    }
                }
            }
            __syncthreads();
            MZ_STAMP(1)  // select
            if (tid >= kThreads - kTrees) {  // the child each simulation starts from (src/search/mcts.py:64-66)
                const int t = tid - (kThreads - kTrees);
                int rc = 0;
                if (t < trees) rc = q.rand_child[(size_t)(g0 + t) * (S + 1) + e];
                rc = min(max(rc, 0), A - 1);
                s_act[t] = rc;
                s_child[t] = base + rc;
                if (q.child_trace && t < trees) q.child_trace[(size_t)(g0 + t) * S + (e - 1)] = base + rc;
            }
            if (MZ_SKIP(5)) {
            } else if (implicit) {
                // node n >= 1 is child a = (n - 1) % A of expansion (n - 1) / A (lock step: every expansion adds A
                // nodes): its latent is relu(z_pool[expansion] + table[a]), formed straight into the split image
                for (int i = tid; i < KCd * 64; i += kThreads) {
                    const int l = i & 63, c = i >> 6, t = l & 15;
                    const int k0 = 32 * c + 8 * (l >> 4);
                    const int n = max(s_leaf[t] - 1, 0), ex = n / A, a = n - ex * A;
                    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                    if (t < trees && k0 < Dp16) {
                        const float *zr = q.z_pool + ((size_t)(g0 + t) * (S + 1) + ex) * D + k0;
                        const lds_f *tr = Tab + a * ld_d + k0;
                        float z[8];
                        if ((D & 3) == 0) {
                            const float4 z0 = k0 < D ? *reinterpret_cast<const float4 *>(zr) : make_float4(0.f, 0.f, 0.f, 0.f);
                            const float4 z1 = k0 + 4 < D ? *reinterpret_cast<const float4 *>(zr + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                            z[0] = z0.x, z[1] = z0.y, z[2] = z0.z, z[3] = z0.w, z[4] = z1.x, z[5] = z1.y, z[6] = z1.z, z[7] = z1.w;
                        } else {
                            for (int j = 0; j < 8; ++j) z[j] = k0 + j < D ? zr[j] : 0.0f;
                        }
                        // columns D .. Dp16 of a merge pre-activation are zero, as is the table there
                        v0 = f32x4{fmaxf(z[0] + tr[0], 0.f), fmaxf(z[1] + tr[1], 0.f), fmaxf(z[2] + tr[2], 0.f), fmaxf(z[3] + tr[3], 0.f)};
                        v1 = f32x4{fmaxf(z[4] + tr[4], 0.f), fmaxf(z[5] + tr[5], 0.f), fmaxf(z[6] + tr[6], 0.f), fmaxf(z[7] + tr[7], 0.f)};
                    }
                    f32x4 hi, lo;
                    split8(v0, v1, hi, lo);
                    lds_f *qd = X + c * kChunkFloats + l * 4;
                    *reinterpret_cast<lds_f4 *>(qd) = hi;
                    *reinterpret_cast<lds_f4 *>(qd + KCd * kChunkFloats) = lo;
                    if (L.need_x32) {
                        *reinterpret_cast<lds_f4 *>(X32 + t * ld_d + k0) = v0;
                        *reinterpret_cast<lds_f4 *>(X32 + t * ld_d + k0 + 4) = v1;
                    }
                }
            } else if ((D & 3) == 0) {  // 16-byte rows: one tree per wave and pass, one float4 per lane
                const int q4 = ld_d >> 2, d4 = D >> 2;
                const int wv = uni(tid >> 6), lane = tid & 63;
                for (int t = wv; t < kTrees; t += kWaves) {
                    const float4 *src = reinterpret_cast<const float4 *>(p.latent + ((size_t)(g0 + t) * p.cap + s_leaf[t]) * D);
                    for (int c4 = lane; c4 < q4; c4 += 64) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (t < trees && c4 < d4) v = src[c4];
                        *reinterpret_cast<lds_f4 *>(X32 + t * ld_d + 4 * c4) = f32x4{v.x, v.y, v.z, v.w};
                    }
                }
            } else {
                for (int i = tid; i < kTrees * ld_d; i += kThreads) {
                    const int t = i / ld_d, c = i - t * ld_d;
                    X32[i] = (t < trees && c < D) ? p.latent[((size_t)(g0 + t) * p.cap + s_leaf[t]) * D + c] : 0.0f;
                }
            }
            __syncthreads();
            MZ_STAMP(2)  // leaf latent gather
        }

        // ---- the leaf's policy layer 1 and merge layer share the input: whole workgroup --------
        Team all = whole_workgroup();
        if (e == 0 || !implicit) {  // (the implicit gather wrote the image itself)
            rows_to_split(X32, ld_d, X, ld_d, 1, KCd, ld_d, all);
            __syncthreads();
        }
        lds_f *pol_in = He1;
        if (!MZ_SKIP(3)) pol_in = first_layers(all, X, X32, He1, He2, Z);
        MZ_STAMP(3)  // expand: policy layer 1 | merge

        // the root expansion has no simulation: the whole workgroup expands
        const bool sim = e > 0;
        Team tm = all;
        if (sim) {
            // the arrival counters are monotonic over the whole search: a team has synced
            // team_syncs times so far (the same number in every wave of the team)
            tm = half ? Team{kWaves / 2, kWaves / 2, s_team + 2, team_arrived} : Team{0, kWaves / 2, s_team, team_arrived};
        }
        if (stamping && tid == kThreads / 2) t_prev = clock64();
        if (!sim || half == 0) {
            if (!MZ_SKIP(1)) expand_tail(tm, e, base, pol_in);
        } else {
            // the simulation is the longer chain: its waves issue first whenever both teams'
            // waves of a SIMD are ready
            __builtin_amdgcn_s_setprio(3);
            if (!MZ_SKIP(2)) simulate(tm, e, base);
            __builtin_amdgcn_s_setprio(0);
        }
        if (sim) team_arrived = tm.arrived;
        __syncthreads();  // the teams meet
        if (!sim) continue;
        MZ_STAMP(10)  // waiting for the other team
        MZ_STAMP(22)

        // ---- backup --------------------------------------------------------------------------
        // The path root .. leaf was recorded by select, so its nodes are updated by independent
        // lanes (lane i of a tree: i = 0 the drawn child, i >= 1 the node i - 1 levels above the
        // leaf) instead of one lane chasing parent pointers.  Signs follow to_play, which
        // alternates with depth; a root to_play of 0 makes every node equal to the leaf's.
        if (!MZ_SKIP(4) && tid < kTrees * kPathCap) {
            const int t = tid / kPathCap, i = tid % kPathCap;
            if (t < trees) {
                const size_t o = (size_t)(g0 + t) * p.cap;
                float roll[8];
                for (int k = 0; k <= depth; ++k) roll[k] = s_roll[t * (depth + 1) + k];
                const int dl = s_depth[t];  // depth of the leaf
                const double ret = mzt::discounted_return(roll, depth + 1, q.discount);
                if (i == 0 && q.rollout_trace)
                    for (int k = 0; k <= depth; ++k) q.rollout_trace[((size_t)(g0 + t) * S + (e - 1)) * (depth + 1) + k] = roll[k];
                if (dl + 1 <= kPathCap - 1) {
                    if (i <= dl + 1) {
                        const int node = (i == 0) ? s_child[t] : s_path[t * kPathCap + (dl - (i - 1))];
                        const bool same = s_root[t] == 0 || (i >= 1 && ((i - 1) & 1) == 0);
                        const size_t c = o + node;
                        mzt::NodeStat st = mzt::load_stat(p, c);
                        st.visit += 1;
                        const double add = same ? ret : -ret;
                        st.value_sum = st.value_sum + add;
                        mzt::store_stat(p, c, st);
                    }
                } else if (i == 0) {  // deeper than the recorded path: walk the parent pointers
                    mzt::backup_path(p, o, s_child[t], (int8_t)s_leaf_tp[t], ret);
                }
            }
        }
        __syncthreads();
        MZ_STAMP(11)  // backup
    }

    // ---- result -------------------------------------------------------------------------------
    if (tid < trees) {
        p.n_nodes[g0 + tid] = 1 + (S + 1) * A;
        mzt::tree_result(p, g0 + tid, policy, policy32, value, visits);
        // a team sync that gave up waiting (never expected) must not pass for a result
        if ((s_team[1] | s_team[3]) != 0 && value) value[g0 + tid] = __builtin_nan("");
    }
    MZ_STAMP(12)  // result
    if (stamping) {
        __syncthreads();
        if (threadIdx.x < 32) q.stamps[threadIdx.x] = s_stamp[threadIdx.x];
#ifdef MZ_DENSE_TIMELINE
        if (threadIdx.x < 64) q.stamps[32 + threadIdx.x] = mz_tl[threadIdx.x >> 3][threadIdx.x & 7];
#endif
    }
#undef MZ_STAMP
}

}  // namespace

int mz_search_seq_try(const mz_pool *pool, const mz_packed_nets *plan, const float *blob, const mz_search_params *prm,
                      double *policy, float *policy32, double *value, int32_t *visits, void *stream);  // mz_search_seq.hip

extern "C" {

int mz_search_run(const mz_pool *pool, const mz_packed_nets *plan, const float *blob, const mz_search_params *prm,
                  double *policy, float *policy32, double *value, int32_t *visits, void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    if (!plan || !blob || !prm) return mz::fail(MZ_ERR_ARG, "mz_search_run: NULL argument");
    if (pool->D <= 0 || (!pool->latent && !prm->z_pool))
        return mz::fail(MZ_ERR_ARG, "mz_search_run: the pool has no latent storage (D=%d) and no z_pool is given", pool->D);
    if (plan->D != pool->D || plan->A != pool->A)
        return mz::fail(MZ_ERR_ARG, "mz_search_run: pool (D=%d, A=%d) and networks (D=%d, A=%d) disagree", pool->D,
                        pool->A, plan->D, plan->A);
    if (pool->A > 32) return mz::fail(MZ_ERR_ARG, "mz_search_run: at most 32 actions (use the per-stage entry points)");
    if (prm->depth > 7) return mz::fail(MZ_ERR_ARG, "mz_search_run: model_look_ahead above 7 is not supported by the fused kernel");
    if (prm->S < 0 || prm->depth < 0 || pool->cap < 1 + (prm->S + 1) * pool->A)
        return mz::fail(MZ_ERR_ARG, "mz_search_run: pool capacity %d below 1 + (S+1)*A = %d", pool->cap,
                        1 + (prm->S + 1) * pool->A);
    if (!prm->root_latent || !prm->root_to_play || !prm->eta || !prm->rand_child || !prm->sqrt_tab || !prm->log_tab)
        return mz::fail(MZ_ERR_ARG, "mz_search_run: NULL input stream or table");
    if (prm->tab_len < prm->S + 2 || prm->tab_len > 4096)
        return mz::fail(MZ_ERR_ARG, "mz_search_run: tab_len must be in [S+2, 4096]");
    mz_search_params q = *prm;
    q.lanes_per_tree = mz::lanes_per_tree(pool->A);
#ifdef MZ_SEARCH_DIAG
    if (const char *e = getenv("MZ_SEARCH_SKIP")) q.lanes_per_tree |= atoi(e) << 8;
#endif
    hipStream_t st = (hipStream_t)stream;
    // the sequential-program form where it applies (same results), else the two-team kernel below
    if (const int handled = mz_search_seq_try(pool, plan, blob, &q, policy, policy32, value, visits, stream))
        return handled < 0 ? handled : MZ_OK;
    const int blocks = (pool->G + kTrees - 1) / kTrees;
    const int candidates[5] = {7, 4, 3, 2, 1};
    for (int mtc : candidates) {
        if (mtc > pool->A && mtc != 1) continue;
        const Lds L = lds_layout(*plan, mtc, q.depth, q.tab_len);
        if (L.total > 160 * 1024) continue;
        auto launch = [&](auto kernel) {
            if (L.total > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
                if (e != hipSuccess) return mz::fail(MZ_ERR_HIP, "k_search: cannot enable %zu bytes of LDS: %s", L.total, hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kThreads), L.total, st, *pool, *plan, blob, q, L, policy, policy32,
                               value, visits);
            return mz::check_launch("k_search");
        };
        switch (mtc) {
            case 7: return launch(k_search<7>);
            case 4: return launch(k_search<4>);
            case 3: return launch(k_search<3>);
            case 2: return launch(k_search<2>);
            default: return launch(k_search<1>);
        }
    }
    return mz::fail(MZ_ERR_ARG, "mz_search_run: networks too wide for the LDS-resident search kernel");
}

}  // extern "C"
