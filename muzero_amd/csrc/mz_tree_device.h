// Device-side tree operations shared by the stand-alone tree kernels and the fused search
// kernel.  Translation units that include this header must be compiled with
// -ffp-contract=off: every score and return below is a chain of individually rounded IEEE
// operations in the reference's order and precision, never an FMA, so that argmax decisions
// and float64 value sums are bit-identical to CPython's.
#pragma once
#include <hip/hip_runtime.h>

#include "muzero_hip.h"

namespace mzt {

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double max_nan_last(double a, double b) { return b > a ? b : a; }

// The two 16-byte halves of a node record, each moved by one dwordx4 access.
struct alignas(16) NodeStat {
    double value_sum;
    int visit;
    int link;
};
struct alignas(16) NodeEdge {
    double prior;
    float reward;
    int first_child;
};
static_assert(sizeof(mz_node) == 32 && sizeof(NodeStat) == 16 && sizeof(NodeEdge) == 16, "node record layout");

__device__ __forceinline__ NodeStat load_stat(const mz_pool &p, size_t c) {
    return *reinterpret_cast<const NodeStat *>(p.node + c);
}
__device__ __forceinline__ void store_stat(const mz_pool &p, size_t c, const NodeStat &s) {
    *reinterpret_cast<NodeStat *>(p.node + c) = s;
}
__device__ __forceinline__ NodeEdge load_edge(const mz_pool &p, size_t c) {
    return *reinterpret_cast<const NodeEdge *>(reinterpret_cast<const char *>(p.node + c) + 16);
}
__device__ __forceinline__ void store_edge(const mz_pool &p, size_t c, const NodeEdge &e) {
    *reinterpret_cast<NodeEdge *>(reinterpret_cast<char *>(p.node + c) + 16) = e;
}
__device__ __forceinline__ int node_first_child(const mz_pool &p, size_t c) { return p.node[c].first_child; }
__device__ __forceinline__ int node_to_play(const mz_pool &p, size_t c) { return MZ_LINK_TO_PLAY(p.node[c].link); }

// A node without statistics or children (src/search/nodes.py:7-22).
__device__ __forceinline__ void write_node(const mz_pool &p, size_t c, int parent, int to_play, double prior, float reward) {
    store_stat(p, c, NodeStat{0.0, 0, MZ_LINK(parent, to_play)});
    store_edge(p, c, NodeEdge{prior, reward, -1});
}

// Maximum over the LPT adjacent lanes of a group (LPT a power of two, groups aligned to LPT);
// every lane of the group receives it.
template <int LPT>
__device__ __forceinline__ double group_max(double v) {
    v = max_nan_last(v, dpp_f64<0xB1>(v));                          // quad_perm [1,0,3,2]
    if constexpr (LPT > 2) v = max_nan_last(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    if constexpr (LPT > 4) v = max_nan_last(v, dpp_f64<0x141>(v));  // row_half_mirror
    if constexpr (LPT > 8) v = max_nan_last(v, dpp_f64<0x140>(v));  // row_mirror
    if constexpr (LPT > 16) v = max_nan_last(v, __shfl_xor(v, 16, 64));
    if constexpr (LPT > 32) v = max_nan_last(v, __shfl_xor(v, 32, 64));
    return v;
}

template <int LPT>
__device__ __forceinline__ unsigned long long group_bits(unsigned long long ballot, int group_lane0) {
    if constexpr (LPT == 64) return ballot;
    return (ballot >> group_lane0) & ((1ull << LPT) - 1);
}

// PUCT descent of one tree by the LPT lanes of a group (src/search/selection.py:43-59).
// `o` = g*cap, `a` = lane within the group, s_tab = [tab_len] sqrt table then [tab_len] log
// table (LDS).  Children of the root are scored in float32 when root_f32 (their priors are
// float32 tensors after src/search/mcts.py:94), everything else in float64; the first
// maximum wins.  Returns the leaf (or -1 when a visit count is outside the tables); every
// lane of the group returns the same value.  Must be called by all lanes of the group.
//
// Optional: `known_root_fc` / `known_root_visit` >= 0 skip the two root loads when the caller
// already knows them; `path` (path_cap entries, e.g. in LDS) receives the visited nodes,
// path[0] = root .. path[depth] = leaf, for depths below path_cap.
template <int LPT>
__device__ __forceinline__ int select_leaf(const mz_pool &p, size_t o, int a, const double *s_tab, int tab_len,
                                           bool root_f32, int *out_depth, int known_root_fc = -1,
                                           int known_root_visit = -1, int *path = nullptr, int path_cap = 0) {
    const int lane = threadIdx.x & 63;
    const int group_lane0 = lane - a;
    const bool child_lane = a < p.A;
    int node = 0, depth = 0;
    int fc = known_root_fc >= 0 ? known_root_fc : p.node[o].first_child;
    int n_parent = known_root_visit >= 0 ? known_root_visit : p.node[o].visit;
    bool failed = false;
    if (path && a == 0 && path_cap > 0) path[0] = 0;
    while (fc >= 0 && depth < p.cap) {
        int n = 0, cfc = -1;
        double w = 0.0, pr = 0.0;
        if (child_lane) {
            const size_t c = o + fc + a;
            const NodeStat st = load_stat(p, c);
            const NodeEdge ed = load_edge(p, c);
            n = st.visit;
            w = st.value_sum;
            pr = ed.prior;
            cfc = ed.first_child;
        }
        const bool bad = n_parent >= tab_len || n_parent < 0 || n >= tab_len || n < 0;
        if (group_bits<LPT>(__ballot(bad), group_lane0)) {
            failed = true;
            break;
        }
        double score = -__builtin_inf();
        if (child_lane) {
            // (both quotients formed side by side, in one block: two dependent float64 divisions in a row are the
            // longest stretch of a level; the mean value is only used where the child has been visited)
            const double q = w / (double)(n > 0 ? n : 1);
            double pb = s_tab[n_parent] / (double)(n + 1);
            pb = pb * s_tab[tab_len + n];
            if (node == 0 && root_f32) {
                float ps = (float)pb * (float)pr;
                float sc = ps;
                if (n > 0) sc = (float)q + ps;
                score = (double)sc;
            } else {
                double ps = pb * pr;
                score = (n > 0) ? (q + ps) : (0.0 + ps);
            }
        }
        const double best = group_max<LPT>(score);
        const unsigned long long hit = group_bits<LPT>(__ballot(child_lane && score == best), group_lane0);
        // all-NaN scores leave no hit: fall back to child 0 like max() keeping its first item
        const int best_a = hit ? (__ffsll((long long)hit) - 1) : 0;
        node = fc + best_a;
        n_parent = __shfl(n, group_lane0 + best_a, 64);
        fc = __shfl(cfc, group_lane0 + best_a, 64);
        ++depth;
        if (path && a == 0 && depth < path_cap) path[depth] = node;
    }
    if (out_depth) *out_depth = depth;
    return failed ? -1 : node;
}

// Child record a of an expansion (src/search/expansion.py:30-33, src/search/nodes.py:24-30).
__device__ __forceinline__ void write_child(const mz_pool &p, size_t o, int base, int a, int leaf, float prior,
                                            float reward, int8_t leaf_to_play) {
    write_node(p, o + base + a, leaf, (int8_t)(-leaf_to_play), (double)prior, reward);
}

// Root prior after the perturbation of src/search/mcts.py:94, all in float32:
// P + ((1 - eps) * P  +  eps * eta), the first product taken in float64 and rounded.
__device__ __forceinline__ float noisy_prior(double prior, float eta, double noise_frac) {
    float keep = (float)((1.0 - noise_frac) * prior);
    float mix = (float)noise_frac * eta;
    float add = keep + mix;
    return (float)prior + add;
}

// Discounted return of a rollout [r_1 .. r_depth, v] (src/search/backpropagation.py:11-13).
__device__ __forceinline__ double discounted_return(const float *rollout, int n, double discount) {
    double ret = 0.0;
    for (int i = n - 1; i >= 0; --i) {
        double m = discount * ret;
        ret = (double)rollout[i] + m;
    }
    return ret;
}

// Path update child -> root (src/search/backpropagation.py:15-18).
__device__ __forceinline__ void backup_path(const mz_pool &p, size_t o, int node, int8_t leaf_to_play, double ret) {
    for (int steps = 0; node >= 0 && steps < p.cap; ++steps) {
        const size_t c = o + node;
        NodeStat st = load_stat(p, c);
        st.visit += 1;
        const double add = ((int8_t)MZ_LINK_TO_PLAY(st.link) == leaf_to_play) ? ret : -ret;
        st.value_sum = st.value_sum + add;
        store_stat(p, c, st);
        node = MZ_LINK_PARENT(st.link);
    }
}

// Search result of one tree (src/search/mcts.py:71-72,97-103).
__device__ __forceinline__ void tree_result(const mz_pool &p, int g, double *policy, float *policy32, double *value,
                                            int32_t *visits) {
    const size_t o = (size_t)g * p.cap;
    const int fc = p.node[o].first_child;
    if (value) value[g] = p.node[o].value_sum / (double)p.node[o].visit;
    if (fc < 0) return;
    double s = 0.0;
    for (int a = 0; a < p.A; ++a) s = s + exp(p.node[o + fc + a].value_sum);
    for (int a = 0; a < p.A; ++a) {
        const double e = exp(p.node[o + fc + a].value_sum) / s;
        if (policy) policy[(size_t)g * p.A + a] = e;
        if (policy32) policy32[(size_t)g * p.A + a] = (float)e;
        if (visits) visits[(size_t)g * p.A + a] = p.node[o + fc + a].visit;
    }
}

}  // namespace mzt
