// Tree kernels for gfx950: G independent MCTS trees held as a flat pool of 32-byte node records in HBM.
//
// Work decomposition: each tree is served by a group of LPT = next_pow2(A) adjacent lanes
// of one wavefront (LPT = 64 -> one wavefront per tree); lane a of the group owns child a
// of the node being scored.  Group-wide maxima use DPP lane exchanges (quad_perm,
// row_half_mirror, row_mirror) and the first-maximum tie-break is a wave ballot + ffs.
// The PUCT sqrt/log tables are staged in LDS (per-lane random index, so LDS not L1).
//
// This file is compiled with -ffp-contract=off: every score and return is a chain of
// individually rounded IEEE operations in the reference's order and precision, never an
// FMA, so that argmax decisions and float64 value sums are bit-identical to CPython's.
#include <cstdint>
#include <algorithm>
#include <type_traits>

#include "mz_common.h"
#include "mz_tree_device.h"

namespace {

constexpr int kBlock = 64;  // one wavefront per block: spreads G*LPT/64 blocks over all CUs

// ---------------------------------------------------------------- kernels
__global__ void k_init(mz_pool p, const int8_t *__restrict__ root_to_play) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G) return;
    size_t o = (size_t)g * p.cap;
    mzt::write_node(p, o, -1, root_to_play[g], 0.0, 0.0f);
    p.n_nodes[g] = 1;
}

// dst row (g, base_g + a) of the latent pool <- src row (g, a); base_g = n_nodes[g] when
// `use_n_nodes`, else `const_base`.
__global__ void k_scatter_latent(mz_pool p, const float *__restrict__ src, int rows_per_tree, int use_n_nodes,
                                 int const_base) {
    const int D = p.D;
    long long total = (long long)p.G * rows_per_tree * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int d = (int)(i % D);
        long long r = i / D;
        int a = (int)(r % rows_per_tree);
        int g = (int)(r / rows_per_tree);
        int base = use_n_nodes ? p.n_nodes[g] : const_base;
        if (base + rows_per_tree > p.cap) continue;
        p.latent[((size_t)g * p.cap + base + a) * D + d] = src[i];
    }
}

__global__ void k_gather_latent(mz_pool p, const int32_t *__restrict__ idx, float *__restrict__ out) {
    const int D = p.D;
    long long total = (long long)p.G * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int d = (int)(i % D);
        int g = (int)(i / D);
        int n = idx[g];
        out[i] = (n >= 0 && n < p.cap) ? p.latent[((size_t)g * p.cap + n) * D + d] : 0.0f;
    }
}

// src/search/selection.py:43-59.  flags bit 0: children of the root carry float32 priors
// and are scored in float32 (the state MCTS.run leaves them in, src/search/mcts.py:94).
template <int LPT>
__global__ void k_select(mz_pool p, const double *__restrict__ sqrt_tab, const double *__restrict__ log_tab,
                         int tab_len, int flags, int32_t *__restrict__ out_leaf, int32_t *__restrict__ out_depth,
                         int32_t *__restrict__ out_row, int32_t *__restrict__ out_child_rows,
                         const int32_t *__restrict__ rand_child, int32_t *__restrict__ out_drawn_row) {
    extern __shared__ double s_tab[];  // [tab_len] sqrt, then [tab_len] log term
    for (int i = threadIdx.x; i < tab_len; i += blockDim.x) {
        s_tab[i] = sqrt_tab[i];
        s_tab[tab_len + i] = log_tab[i];
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / LPT;
    const int a = t % LPT;
    if (g >= p.G) return;
    int depth = 0;
    const int leaf = mzt::select_leaf<LPT>(p, (size_t)g * p.cap, a, s_tab, tab_len, flags & 1, &depth);
    if (a == 0) {
        out_leaf[g] = leaf;
        if (out_depth) out_depth[g] = depth;
        if (out_row) out_row[g] = leaf >= 0 ? g * p.cap + leaf : -1;  // a failed descent names no row (read as zeros)
        if (out_drawn_row) {  // the child the simulation will start from, as k_expand_commit will name it
            const int rc = rand_child[g];
            const bool ok = leaf >= 0 && p.n_nodes[g] + p.A <= p.cap && rc >= 0 && rc < p.A;
            out_drawn_row[g] = ok ? g * p.cap + p.n_nodes[g] + rc : -1;
        }
    }
    // pool rows an expansion of this leaf would give its children
    if (out_child_rows && a < p.A)
        out_child_rows[g * p.A + a] = (leaf >= 0 && p.n_nodes[g] + p.A <= p.cap) ? g * p.cap + p.n_nodes[g] + a : -1;
}

// src/search/expansion.py:30-35 + src/search/nodes.py:24-30.
template <int LPT>
__global__ void k_expand_commit(mz_pool p, const int32_t *__restrict__ leaf, const float *__restrict__ prior,
                                const float *__restrict__ reward, const int32_t *__restrict__ rand_child,
                                int32_t *__restrict__ out_child, int32_t *__restrict__ out_child_row) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / LPT;
    const int a = t % LPT;
    if (g >= p.G) return;
    const size_t o = (size_t)g * p.cap;
    const int base = p.n_nodes[g];
    const int lf = leaf[g];
    const bool ok = lf >= 0 && lf < base && base + p.A <= p.cap && p.node[o + lf].first_child < 0;
    if (ok && a < p.A)
        mzt::write_child(p, o, base, a, lf, prior[(size_t)g * p.A + a], reward[(size_t)g * p.A + a], (int8_t)mzt::node_to_play(p, o + lf));
    // every lane of the group has read n_nodes / first_child before lane 0 rewrites them
    __builtin_amdgcn_wave_barrier();
    if (a == 0) {
        if (ok) {
            p.node[o + lf].first_child = base;
            p.n_nodes[g] = base + p.A;
            int rc = rand_child[g];
            out_child[g] = (rc >= 0 && rc < p.A) ? base + rc : -1;
        } else {
            out_child[g] = -1;
        }
        if (out_child_row) out_child_row[g] = out_child[g] >= 0 ? g * p.cap + out_child[g] : -1;
    }
}

// src/search/mcts.py:91-94, all in float32 as the reference's tensor arithmetic does.
__global__ void k_root_noise(mz_pool p, const float *__restrict__ eta, double noise_frac) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = t / p.A, a = t % p.A;
    if (g >= p.G) return;
    size_t o = (size_t)g * p.cap;
    int fc = p.node[o].first_child;
    if (fc < 0) return;
    p.node[o + fc + a].prior = (double)mzt::noisy_prior(p.node[o + fc + a].prior, eta[(size_t)g * p.A + a], noise_frac);
}

// src/search/backpropagation.py:9-18.  One lane per tree: the path is a pointer chase of
// depth+2 nodes; trees are independent so the wave's 64 chases overlap.
__global__ void k_backup(mz_pool p, const int32_t *__restrict__ child, const int32_t *__restrict__ leaf,
                         const float *__restrict__ rollout, int n_rollout, double discount) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G) return;
    int node = child[g];
    int lf = leaf[g];
    if (node < 0 || lf < 0) return;
    size_t o = (size_t)g * p.cap;
    const double ret = mzt::discounted_return(rollout + (size_t)g * n_rollout, n_rollout, discount);
    mzt::backup_path(p, o, node, (int8_t)mzt::node_to_play(p, o + lf), ret);
}

// The tree side of one lock-step simulation step as ONE launch instead of three: the child records of this
// step's expansion (k_expand_commit), the backup of this step's rollout (k_backup) and, unless this was the
// last step, the descent of the NEXT step (k_select) -- all per tree and in this order, LPT lanes per tree.
// The lanes of a tree sit in one wave; between the phases the wave's stores are made visible to its own
// later loads (workgroup-scope release / acquire around a wave barrier).  `leaf` holds this step's leaves on
// entry and the next step's on exit.
template <int LPT>
__global__ void k_commit_backup_select(mz_pool p, int32_t *__restrict__ leaf, const float *__restrict__ prior,
                                       const float *__restrict__ reward, const int32_t *__restrict__ rand_child,
                                       int32_t *__restrict__ out_child, const float *__restrict__ rollout,
                                       const float *__restrict__ reward0, int n_rollout,
                                       double discount, const double *__restrict__ sqrt_tab,
                                       const double *__restrict__ log_tab, int tab_len, int flags, int do_select,
                                       int32_t *__restrict__ out_row, int32_t *__restrict__ out_child_rows,
                                       const int32_t *__restrict__ rand_next, int32_t *__restrict__ out_drawn_row) {
    extern __shared__ double s_tab[];  // [tab_len] sqrt, then [tab_len] log term
    if (do_select)
        for (int i = threadIdx.x; i < tab_len; i += blockDim.x) {
            s_tab[i] = sqrt_tab[i];
            s_tab[tab_len + i] = log_tab[i];
        }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t / LPT;
    const int a = t % LPT;
    if (g >= p.G) return;
    const size_t o = (size_t)g * p.cap;
    auto phase_fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // ---- child records (k_expand_commit)
    const int base = p.n_nodes[g];
    const int lf = leaf[g];
    const bool ok = lf >= 0 && lf < base && base + p.A <= p.cap && p.node[o + lf].first_child < 0;
    if (ok && a < p.A)
        mzt::write_child(p, o, base, a, lf, prior[(size_t)g * p.A + a], reward[(size_t)g * p.A + a], (int8_t)mzt::node_to_play(p, o + lf));
    __builtin_amdgcn_wave_barrier();  // every lane of the group has read n_nodes / first_child before lane 0 rewrites them
    int child = -1;
    if (a == 0) {
        if (ok) {
            p.node[o + lf].first_child = base;
            p.n_nodes[g] = base + p.A;
            const int rc = rand_child[g];
            child = (rc >= 0 && rc < p.A) ? base + rc : -1;
        }
        out_child[g] = child;
    }
    phase_fence();
    // ---- backup (k_backup)
    if (a == 0 && child >= 0 && lf >= 0) {
        double ret;
        if (reward0 != nullptr) {  // the first step's reward arrives in an array of its own (n_rollout <= 8 then)
            float roll[8];
            for (int k = 0; k < n_rollout; ++k) roll[k] = rollout[(size_t)g * n_rollout + k];
            roll[0] = reward0[g];
            ret = mzt::discounted_return(roll, n_rollout, discount);
        } else {
            ret = mzt::discounted_return(rollout + (size_t)g * n_rollout, n_rollout, discount);
        }
        mzt::backup_path(p, o, child, (int8_t)mzt::node_to_play(p, o + lf), ret);
    }
    if (!do_select) return;
    phase_fence();
    // ---- the next step's descent (k_select)
    int depth = 0;
    const int next = mzt::select_leaf<LPT>(p, o, a, s_tab, tab_len, flags & 1, &depth);
    const int nn = p.n_nodes[g];
    if (a == 0) {
        leaf[g] = next;
        if (out_row) out_row[g] = next >= 0 ? g * p.cap + next : -1;
        if (out_drawn_row) {
            const int rc = rand_next[g];
            out_drawn_row[g] = (next >= 0 && nn + p.A <= p.cap && rc >= 0 && rc < p.A) ? g * p.cap + nn + rc : -1;
        }
    }
    if (out_child_rows && a < p.A) out_child_rows[g * p.A + a] = (next >= 0 && nn + p.A <= p.cap) ? g * p.cap + nn + a : -1;
}

// src/search/mcts.py:71-72,97-103.
__global__ void k_result(mz_pool p, double *__restrict__ policy, float *__restrict__ policy32,
                         double *__restrict__ value, int32_t *__restrict__ visits) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G) return;
    mzt::tree_result(p, g, policy, policy32, value, visits);
}

template <typename F>
int dispatch_lpt(int A, F &&f) {
    switch (mz::lanes_per_tree(A)) {
        case 2: return f(std::integral_constant<int, 2>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 8: return f(std::integral_constant<int, 8>{});
        case 16: return f(std::integral_constant<int, 16>{});
        case 32: return f(std::integral_constant<int, 32>{});
        default: return f(std::integral_constant<int, 64>{});
    }
}

inline int blocks_for(long long threads, int block) { return (int)((threads + block - 1) / block); }

}  // namespace

extern "C" {

int mz_puct_tables(double *h_sqrt_tab, double *h_log_tab, int32_t len) {
    if (!h_sqrt_tab || !h_log_tab || len <= 0) return mz::fail(MZ_ERR_ARG, "mz_puct_tables: bad arguments");
    for (int n = 0; n < len; ++n) {
        h_sqrt_tab[n] = sqrt((double)n);
        h_log_tab[n] = log(((double)n + 19652.0 + 1.0) / 19652.0) + 1.25;
    }
    return MZ_OK;
}

int mz_tree_init(const mz_pool *pool, const int8_t *root_to_play, const float *root_latent, void *stream) {
    MZ_TRY(mz::check_pool(pool, root_latent != nullptr));
    if (!root_to_play) return mz::fail(MZ_ERR_ARG, "mz_tree_init: root_to_play is NULL");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_init, dim3(blocks_for(pool->G, kBlock)), dim3(kBlock), 0, st, *pool, root_to_play);
    MZ_TRY(mz::check_launch("k_init"));
    if (root_latent) {
        long long total = (long long)pool->G * pool->D;
        int nb = (int)std::min<long long>(blocks_for(total, 256), 65535);
        hipLaunchKernelGGL(k_scatter_latent, dim3(nb), dim3(256), 0, st, *pool, root_latent, 1, 0, 0);
        MZ_TRY(mz::check_launch("k_scatter_latent(root)"));
    }
    return MZ_OK;
}

int mz_tree_select_rows(const mz_pool *pool, const double *sqrt_tab, const double *log_tab, int32_t tab_len,
                        int32_t flags, int32_t *out_leaf, int32_t *out_depth, int32_t *out_row,
                        int32_t *out_child_rows, void *stream) {
    return mz_tree_select_rows_drawn(pool, sqrt_tab, log_tab, tab_len, flags, out_leaf, out_depth, out_row, out_child_rows,
                                     nullptr, nullptr, stream);
}

int mz_tree_select_rows_drawn(const mz_pool *pool, const double *sqrt_tab, const double *log_tab, int32_t tab_len,
                              int32_t flags, int32_t *out_leaf, int32_t *out_depth, int32_t *out_row,
                              int32_t *out_child_rows, const int32_t *rand_child, int32_t *out_drawn_row, void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    if (!sqrt_tab || !log_tab || tab_len <= 0 || !out_leaf || (out_drawn_row && !rand_child))
        return mz::fail(MZ_ERR_ARG, "mz_tree_select: NULL table/output or tab_len <= 0");
    if ((long long)pool->G * pool->cap > INT32_MAX)
        return mz::fail(MZ_ERR_ARG, "mz_tree_select: G * cap exceeds the 32-bit row index");
    size_t lds = sizeof(double) * 2 * (size_t)tab_len;
    if (lds > 64 * 1024) return mz::fail(MZ_ERR_ARG, "mz_tree_select: tab_len %d exceeds the 4096-entry LDS table", tab_len);
    hipStream_t st = (hipStream_t)stream;
    return dispatch_lpt(pool->A, [&](auto lpt) {
        constexpr int LPT = decltype(lpt)::value;
        hipLaunchKernelGGL(k_select<LPT>, dim3(blocks_for((long long)pool->G * LPT, kBlock)), dim3(kBlock), lds, st,
                           *pool, sqrt_tab, log_tab, tab_len, flags, out_leaf, out_depth, out_row, out_child_rows, rand_child,
                           out_drawn_row);
        return mz::check_launch("k_select");
    });
}

int mz_tree_select(const mz_pool *pool, const double *sqrt_tab, const double *log_tab, int32_t tab_len,
                   int32_t flags, int32_t *out_leaf, int32_t *out_depth, void *stream) {
    return mz_tree_select_rows(pool, sqrt_tab, log_tab, tab_len, flags, out_leaf, out_depth, nullptr, nullptr, stream);
}

int mz_tree_expand_commit_rows(const mz_pool *pool, const int32_t *leaf, const float *prior, const float *reward,
                               const float *next_latent, const int32_t *rand_child, int32_t *out_child,
                               int32_t *out_child_row, void *stream) {
    MZ_TRY(mz::check_pool(pool, next_latent != nullptr));
    if (!leaf || !prior || !reward || !rand_child || !out_child)
        return mz::fail(MZ_ERR_ARG, "mz_tree_expand_commit: NULL argument");
    if ((long long)pool->G * pool->cap > INT32_MAX)
        return mz::fail(MZ_ERR_ARG, "mz_tree_expand_commit: G * cap exceeds the 32-bit row index");
    hipStream_t st = (hipStream_t)stream;
    if (next_latent) {
        long long total = (long long)pool->G * pool->A * pool->D;
        int nb = (int)std::min<long long>(blocks_for(total, 256), 65535);
        hipLaunchKernelGGL(k_scatter_latent, dim3(nb), dim3(256), 0, st, *pool, next_latent, pool->A, 1, 0);
        MZ_TRY(mz::check_launch("k_scatter_latent(children)"));
    }
    return dispatch_lpt(pool->A, [&](auto lpt) {
        constexpr int LPT = decltype(lpt)::value;
        hipLaunchKernelGGL(k_expand_commit<LPT>, dim3(blocks_for((long long)pool->G * LPT, kBlock)), dim3(kBlock), 0,
                           st, *pool, leaf, prior, reward, rand_child, out_child, out_child_row);
        return mz::check_launch("k_expand_commit");
    });
}

int mz_tree_expand_commit(const mz_pool *pool, const int32_t *leaf, const float *prior, const float *reward,
                          const float *next_latent, const int32_t *rand_child, int32_t *out_child, void *stream) {
    return mz_tree_expand_commit_rows(pool, leaf, prior, reward, next_latent, rand_child, out_child, nullptr, stream);
}

int mz_tree_root_noise(const mz_pool *pool, const float *eta, double noise_frac, void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    if (!eta) return mz::fail(MZ_ERR_ARG, "mz_tree_root_noise: eta is NULL");
    hipLaunchKernelGGL(k_root_noise, dim3(blocks_for((long long)pool->G * pool->A, 256)), dim3(256), 0,
                       (hipStream_t)stream, *pool, eta, noise_frac);
    return mz::check_launch("k_root_noise");
}

int mz_tree_backup(const mz_pool *pool, const int32_t *child, const int32_t *leaf, const float *rollout,
                   int32_t n_rollout, double discount, void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    if (!child || !leaf || !rollout || n_rollout <= 0)
        return mz::fail(MZ_ERR_ARG, "mz_tree_backup: NULL argument or n_rollout <= 0");
    hipLaunchKernelGGL(k_backup, dim3(blocks_for(pool->G, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *pool, child,
                       leaf, rollout, n_rollout, discount);
    return mz::check_launch("k_backup");
}

int mz_tree_commit_backup_select(const mz_pool *pool, int32_t *leaf, const float *prior, const float *reward,
                                 const int32_t *rand_child, int32_t *out_child, const float *rollout, const float *reward0,
                                 int32_t n_rollout, double discount, const double *sqrt_tab, const double *log_tab, int32_t tab_len,
                                 int32_t flags, int32_t do_select, int32_t *out_row, int32_t *out_child_rows,
                                 const int32_t *rand_next, int32_t *out_drawn_row, void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    if (!leaf || !prior || !reward || !rand_child || !out_child || !rollout || n_rollout <= 0 || (reward0 && n_rollout > 8))
        return mz::fail(MZ_ERR_ARG, "mz_tree_commit_backup_select: NULL argument, n_rollout <= 0, or reward0 with n_rollout > 8");
    if (do_select && (!sqrt_tab || !log_tab || tab_len <= 0 || (out_drawn_row && !rand_next)))
        return mz::fail(MZ_ERR_ARG, "mz_tree_commit_backup_select: NULL table or tab_len <= 0");
    if ((long long)pool->G * pool->cap > INT32_MAX)
        return mz::fail(MZ_ERR_ARG, "mz_tree_commit_backup_select: G * cap exceeds the 32-bit row index");
    const size_t lds = do_select ? sizeof(double) * 2 * (size_t)tab_len : 0;
    if (lds > 64 * 1024) return mz::fail(MZ_ERR_ARG, "mz_tree_commit_backup_select: tab_len %d exceeds the 4096-entry LDS table", tab_len);
    hipStream_t st = (hipStream_t)stream;
    return dispatch_lpt(pool->A, [&](auto lpt) {
        constexpr int LPT = decltype(lpt)::value;
        hipLaunchKernelGGL(k_commit_backup_select<LPT>, dim3(blocks_for((long long)pool->G * LPT, kBlock)), dim3(kBlock), lds,
                           st, *pool, leaf, prior, reward, rand_child, out_child, rollout, reward0, n_rollout, discount, sqrt_tab,
                           log_tab, tab_len, flags, do_select, out_row, out_child_rows, rand_next, out_drawn_row);
        return mz::check_launch("k_commit_backup_select");
    });
}

int mz_tree_result(const mz_pool *pool, double *policy, float *policy32, double *value, int32_t *visits,
                   void *stream) {
    MZ_TRY(mz::check_pool(pool, false));
    hipLaunchKernelGGL(k_result, dim3(blocks_for(pool->G, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, *pool,
                       policy, policy32, value, visits);
    return mz::check_launch("k_result");
}

int mz_tree_gather_latent(const mz_pool *pool, const int32_t *idx, float *out, void *stream) {
    MZ_TRY(mz::check_pool(pool, true));
    if (!idx || !out) return mz::fail(MZ_ERR_ARG, "mz_tree_gather_latent: NULL argument");
    long long total = (long long)pool->G * pool->D;
    int nb = (int)std::min<long long>(blocks_for(total, 256), 65535);
    hipLaunchKernelGGL(k_gather_latent, dim3(nb), dim3(256), 0, (hipStream_t)stream, *pool, idx, out);
    return mz::check_launch("k_gather_latent");
}

}  // extern "C"
