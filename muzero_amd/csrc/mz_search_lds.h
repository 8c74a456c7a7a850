// LDS layout of the fused search kernels (csrc/mz_search.hip): one
// description shared by the host (size, launch) and the device (buffer offsets).
#pragma once
#include "mz_mlp_device.h"

namespace mzs {

using namespace mzd;

constexpr int kTrees = 16;  // trees per workgroup = one MFMA row tile

constexpr int kPathCap = 16;  // select records the first kPathCap levels of its path in LDS

struct Lds {
    int ld_d, ld_h, ld_r, rows_r, tab_rows;
    int fold_reward;  // reward head = [matrix layer, one-output layer]: folded over the children, no ra/rb
    int need_x32;     // a head's first layer is narrow: latents are also kept as fp32 rows (r32, xs32)
    // byte offsets.  x32: leaf latents as fp32 rows (as gathered), later the simulation's current latent
    // as a split image; x: the leaf latents' split image; he1/he2: hidden activations of the expansion;
    // r: split image of the children's latents, mtc row tiles, over x | he1 | he2 (dead by then);
    // z: merge pre-activation of the leaf; zs: merge pre-activation along the simulation; hs1/hs2: hidden
    // activations of the simulation; xs32: the simulation's current latent as fp32 rows; r32: the children's
    // latents as fp32 rows; ra/rb: hidden activations of the reward head over the children
    size_t x32, x, he1, he2, r, z, zs, tab, hs1, hs2, xs32, r32, ra, rb, dot, dots, dotv, nets, prior, rew, roll, ints, path, tabs, total;
    int narrow[3][MZ_MAX_LAYERS];  // byte offset of a narrow layer's LDS weight copy (policy, value, reward), or -1
};

constexpr int kIntsPerTree = 6;  // s_leaf, s_child, s_act, s_leaf_tp, s_depth, s_root

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// n_teams: partial teams of the kernel (two ints each: arrival counter, error flag)
__host__ __device__ inline Lds lds_layout(const mz_packed_nets &P, int mtc, int depth, int tab_len, int n_teams = 2) {
    Lds L;
    const int A = P.A;
    L.ld_d = lds_stride(P.D);
    int wh = P.policy.max_width > P.value.max_width ? P.policy.max_width : P.value.max_width;
    if (P.reward.max_width > wh) wh = P.reward.max_width;
    L.ld_h = lds_stride(wh);
    L.ld_r = lds_stride(P.reward.max_width);
    L.rows_r = kTrees * mtc;
    L.tab_rows = ((A + mtc - 1) / mtc) * mtc;
    L.need_x32 = (!is_matrix_layer(P.policy.layer[0]) || !is_matrix_layer(P.value.layer[0]) || !is_matrix_layer(P.reward.layer[0])) ? 1 : 0;
    size_t off = sizeof(float) * kLdsReserve;  // offset 0 is never handed out (it compares equal to NULL)
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    const size_t tile_d = sizeof(float) * kTrees * L.ld_d;
    L.x32 = take(tile_d);
    const int ld_he = lds_stride(P.policy.max_width);
    L.x = take(tile_d);
    L.he1 = take(sizeof(float) * kTrees * ld_he);
    L.he2 = take(P.policy.n_layers >= 2 ? sizeof(float) * kTrees * ld_he : 0);
    L.r = L.x;
    if (off < L.r + tile_d * mtc) off = align16(L.r + tile_d * mtc);
    L.z = take(tile_d);
    L.zs = take(tile_d);
    L.tab = take(sizeof(float) * L.tab_rows * L.ld_d);
    L.hs1 = take(sizeof(float) * kTrees * L.ld_h);
    L.hs2 = take(sizeof(float) * kTrees * L.ld_h);
    L.xs32 = take(L.need_x32 ? tile_d : 0);
    L.r32 = take(!is_matrix_layer(P.reward.layer[0]) ? tile_d * mtc : 0);
    L.fold_reward = (P.reward.n_layers == 2 && tail_foldable(P.reward)) ? 1 : 0;
    L.ra = take(L.fold_reward ? 0 : sizeof(float) * L.rows_r * L.ld_r);
    L.rb = take(L.fold_reward ? 0 : sizeof(float) * L.rows_r * L.ld_r);
    L.dot = take(L.fold_reward ? sizeof(float) * dot_floats(P.reward.layer[0].Np >> 4, L.rows_r) : 0);
    L.dots = take(L.fold_reward ? sizeof(float) * dot_floats(P.reward.layer[0].Np >> 4, kTrees) : 0);
    L.dotv = take(P.value.n_layers == 2 && tail_foldable(P.value) ? sizeof(float) * dot_floats(P.value.layer[0].Np >> 4, kTrees) : 0);
    L.nets = take(sizeof(mz_packed_nets) + sizeof(int) * 3 * MZ_MAX_LAYERS);  // plan + narrow-layer offsets
    L.prior = take(sizeof(float) * kTrees * A);
    L.rew = take(sizeof(float) * kTrees * L.tab_rows);
    L.roll = take(sizeof(float) * kTrees * (depth + 1));
    L.ints = take(sizeof(int) * (kTrees * kIntsPerTree + 2 * n_teams));
    L.path = take(sizeof(int) * kTrees * kPathCap);
    L.tabs = take(sizeof(double) * 2 * tab_len);
    const mz_packed_mlp *mlps[3] = {&P.policy, &P.value, &P.reward};
    for (int m = 0; m < 3; ++m)
        for (int i = 0; i < MZ_MAX_LAYERS; ++i) {
            L.narrow[m][i] = -1;
            if (i < mlps[m]->n_layers && !is_matrix_layer(mlps[m]->layer[i]))
                L.narrow[m][i] = (int)take(sizeof(float) * narrow_lds_floats(mlps[m]->layer[i]));
        }
    L.total = off;
    return L;
}

}  // namespace mzs
