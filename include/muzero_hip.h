/*
 * libmuzero_hip.so -- C ABI of the MI355X (gfx950) MuZero search / self-play hot path.
 *
 * The reference (SverreNystad/MuZero) is pure Python with no FFI, so the boundary this
 * library replaces is the set of Python call sites named beside each entry point; the
 * host-side mirror in muzero_amd/ binds these symbols with ctypes and keeps the
 * reference's Python signatures on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name starts with `h_`;
 *   - the caller owns all memory; the library never allocates device memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on that stream and keep no host state between calls;
 *   - return value 0 = success, negative = failure (argument validation or a HIP launch
 *     error); mz_last_error() returns a thread-local description of the last failure;
 *   - arrays are row-major and densely packed; `G` = number of independent trees/games,
 *     `A` = actions per expansion (1..64), `cap` = node capacity per tree, `D` = floats
 *     per latent state.
 */
#ifndef MUZERO_HIP_H
#define MUZERO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_OK 0
#define MZ_ERR_ARG (-1)
#define MZ_ERR_HIP (-2)

#define MZ_SELECT_ROOT_F32 1

#define MZ_ACT_NONE 0
#define MZ_ACT_RELU 1
#define MZ_ACT_TANH 2
#define MZ_ACT_SIGMOID 3
#define MZ_ACT_SOFTMAX 4

const char *mz_last_error(void);
int mz_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Node pool: G trees x cap nodes, one 32-byte record per node.  Node 0 of every tree is its
 * root; an expansion appends A children in action order, so the children of node i are
 * first_child .. first_child+A-1 and their records are adjacent in memory.   Replaces the
 * per-node Python object src/search/nodes.py:7-30 (fields visit_count, value_sum,
 * policy_priority, reward, parent, children, to_play, latent_state).
 *
 * The first 16 bytes of a record are all a path update touches (one read-modify-write per
 * node, src/search/backpropagation.py:15-18); a descent reads both halves of the A adjacent
 * child records (src/search/selection.py:43-59).
 * ------------------------------------------------------------------------------------ */
typedef struct mz_node {
    double value_sum;    /* Node.value_sum                                                  */
    int32_t visit;       /* Node.visit_count                                                */
    int32_t link;        /* bits 0..23: parent index, all ones for the root (no parent);    */
                         /* bits 24..31: Node.to_play as a signed byte                      */
    double prior;        /* Node.policy_priority                                            */
    float reward;        /* Node.reward                                                     */
    int32_t first_child; /* -1 while the node has no children                               */
} mz_node;

#define MZ_LINK(parent, to_play) ((int32_t)(((uint32_t)(uint8_t)(int8_t)(to_play) << 24) | ((uint32_t)(parent) & 0xFFFFFFu)))
#define MZ_LINK_PARENT(link) ((int32_t)((uint32_t)(link) << 8) >> 8)
#define MZ_LINK_TO_PLAY(link) ((int32_t)(link) >> 24)

typedef struct mz_pool {
    int32_t G, A, cap, D; /* cap <= 2^23 nodes per tree                                      */
    mz_node *node;        /* [G][cap]      32-byte aligned                                   */
    int32_t *n_nodes;     /* [G]           nodes in use                                      */
    float *latent;        /* [G][cap][D]   Node.latent_state (may be NULL: statistics only)  */
} mz_pool;

/* Fills the two PUCT lookup tables on the HOST with the C library's sqrt/log so the
 * device never evaluates a transcendental on the argmax path:
 *   h_sqrt_tab[n] = sqrt(n),  h_log_tab[n] = log((n + 19652 + 1) / 19652) + 1.25,  n < len
 * (src/search/selection.py:46-47; c1 = 1.25, c2 = 19652 are the reference's fixed
 * defaults, src/search/factory.py:37).  The caller copies them to the device. */
int mz_puct_tables(double *h_sqrt_tab, double *h_log_tab, int32_t len);

/* Root creation: Node(latent_state, to_play) -- src/training_data_generator.py:119.
 * root_to_play i8[G]; root_latent f32[G][D] or NULL. */
int mz_tree_init(const mz_pool *pool, const int8_t *root_to_play, const float *root_latent,
                 void *stream);

/* PUCT descent root -> childless node: src/search/selection.py:43-59.  Children of the
 * root are scored in float32 when `flags & MZ_SELECT_ROOT_F32` (the state MCTS.run leaves
 * them in: src/search/mcts.py:94 turns their priors into float32 tensors), every other
 * level in float64; the first maximum wins.  out_leaf i32[G], out_depth i32[G] (may be NULL).
 * sqrt_tab/log_tab: device copies of mz_puct_tables, tab_len entries; a visit count
 * >= tab_len fails the call's tree with leaf = -1. */
int mz_tree_select(const mz_pool *pool, const double *sqrt_tab, const double *log_tab,
                   int32_t tab_len, int32_t flags, int32_t *out_leaf, int32_t *out_depth,
                   void *stream);

/* Child creation for one expansion per tree: src/search/expansion.py:30-35 with
 * src/search/nodes.py:24-30.  leaf i32[G]; prior f32[G][A] (the value the reference stores
 * with .item() after its extra softmax); reward f32[G][A]; next_latent f32[G][A][D] or NULL
 * (NULL: latents were already written in place by mz_dynamics_expand); rand_child i32[G]
 * in [0,A) = the index random.choice drew; out_child i32[G] = pool index of that child.
 * A tree whose pool is full is left unchanged and gets out_child = -1. */
int mz_tree_expand_commit(const mz_pool *pool, const int32_t *leaf, const float *prior,
                          const float *reward, const float *next_latent,
                          const int32_t *rand_child, int32_t *out_child, void *stream);

/* mz_tree_select / mz_tree_expand_commit that also return pool ROW indices (tree * cap + node),
 * the form the network entry points gather and scatter latents by: out_row i32[G] = row of the
 * selected leaf, out_child_rows i32[G][A] = rows its A children would get (n_nodes .. n_nodes+A-1),
 * out_child_row i32[G] = row of the drawn child.  A tree whose descent or expansion failed (leaf or
 * out_child = -1, a full pool) gets -1 in place of every row, which the network entry points read as a
 * zero row and skip when scattering.  Any of the three may be NULL.  Same reference lines as the two calls above; the lock-step search uses
 * these to save the index arithmetic launches between selection and the network calls. */
int mz_tree_select_rows(const mz_pool *pool, const double *sqrt_tab, const double *log_tab,
                        int32_t tab_len, int32_t flags, int32_t *out_leaf, int32_t *out_depth,
                        int32_t *out_row, int32_t *out_child_rows, void *stream);
/* mz_tree_select_rows that also names the row of the child the simulation will start from
 * (src/search/mcts.py:64-66 draws it right after the expansion): out_drawn_row i32[G] = row of child
 * rand_child[g] among the rows of out_child_rows, -1 where mz_tree_expand_commit would give out_child = -1.
 * Known before the expansion is committed, so the rollout's network calls need not wait for the child
 * records (priors, rewards) to be written. */
int mz_tree_select_rows_drawn(const mz_pool *pool, const double *sqrt_tab, const double *log_tab,
                              int32_t tab_len, int32_t flags, int32_t *out_leaf, int32_t *out_depth,
                              int32_t *out_row, int32_t *out_child_rows, const int32_t *rand_child,
                              int32_t *out_drawn_row, void *stream);
/* The tree side of one lock-step step in one launch: mz_tree_expand_commit (child records of this step's
 * expansion; `leaf` holds this step's leaves), mz_tree_backup of this step's rollout from out_child, and --
 * with do_select != 0 -- mz_tree_select_rows_drawn for the NEXT step (its leaves overwrite `leaf`, rand_next
 * names its drawn children), per tree and in this order.  reward0 f32[G] (optional, n_rollout <= 8): read in
 * place of rollout[g][0] (a caller that evaluates the first rollout reward together with the children's
 * rewards leaves it in an array of its own).  Same records and the same descents as the three calls
 * (src/search/expansion.py:30-35, backpropagation.py:9-18, selection.py:43-59). */
int mz_tree_commit_backup_select(const mz_pool *pool, int32_t *leaf, const float *prior, const float *reward,
                                 const int32_t *rand_child, int32_t *out_child, const float *rollout,
                                 const float *reward0, int32_t n_rollout, double discount, const double *sqrt_tab,
                                 const double *log_tab, int32_t tab_len, int32_t flags, int32_t do_select,
                                 int32_t *out_row, int32_t *out_child_rows, const int32_t *rand_next,
                                 int32_t *out_drawn_row, void *stream);
int mz_tree_expand_commit_rows(const mz_pool *pool, const int32_t *leaf, const float *prior,
                               const float *reward, const float *next_latent,
                               const int32_t *rand_child, int32_t *out_child,
                               int32_t *out_child_row, void *stream);

/* Root prior perturbation: src/search/mcts.py:91-94 (in float32:
 * P <- P + ((1-eps)*P + eps*eta), i.e. (2-eps)*P + eps*eta).  eta f32[G][A]. */
int mz_tree_root_noise(const mz_pool *pool, const float *eta, double noise_frac, void *stream);

/* Discounted return + path update: src/search/backpropagation.py:9-18.
 * child i32[G] (start node), leaf i32[G] (its to_play decides the sign),
 * rollout f32[G][n_rollout] = [r_1 .. r_depth, v] from src/search/simulation.py:44.
 * Trees with child < 0 are skipped. */
int mz_tree_backup(const mz_pool *pool, const int32_t *child, const int32_t *leaf,
                   const float *rollout, int32_t n_rollout, double discount, void *stream);

/* Search result: src/search/mcts.py:71-72,97-103.  policy f64[G][A] = softmax over the
 * root children's value sums, policy32 f32[G][A] = its float32 cast
 * (src/training_data_generator.py:122), value f64[G] = root value_sum / visit_count,
 * visits i32[G][A] = root children's visit counts.  Any output may be NULL. */
int mz_tree_result(const mz_pool *pool, double *policy, float *policy32, double *value,
                   int32_t *visits, void *stream);

/* out[g][:] = pool.latent[g][idx[g]][:]   (idx < 0 -> zeros). */
int mz_tree_gather_latent(const mz_pool *pool, const int32_t *idx, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * MLP networks (`res_net: []` configurations): fused fp32 MFMA kernels.
 * Weights are passed in torch.nn.Linear layout W[out][in] (row stride ldw), bias[out].
 * ------------------------------------------------------------------------------------ */
#define MZ_MAX_LAYERS 8

typedef struct mz_dense {
    const float *weight; /* [out][in], row stride ldw floats (0 = in_features) */
    const float *bias;   /* [out] or NULL */
    int32_t in_features, out_features, activation, ldw;
} mz_dense;

/* Sequence of Linear+activation layers: build_mlp, src/neural_networks/network_builder.py:183-205. */
typedef struct mz_mlp {
    int32_t n_layers;
    mz_dense layer[MZ_MAX_LAYERS];
} mz_mlp;

/* Layout of one MLP inside a packed weight blob (offsets in floats from the blob start).
 * fmt 1: a layer evaluated on the matrix cores -- Kp = K rounded up to 32, weights stored as two fp16
 * planes (value and scaled residual, 4 bytes per weight); fmt 0: fp32 tiles, Kp = K rounded up to 16
 * (a final one-output layer that is folded into the epilogue of the wide layer before it; every
 * layer of mz_wide_plan). */
typedef struct mz_packed_layer {
    int32_t K, N, Kp, Np, act, fmt;
    int64_t w_off, b_off;
} mz_packed_layer;

typedef struct mz_packed_mlp {
    int32_t n_layers, in_features, out_features, max_width;
    mz_packed_layer layer[MZ_MAX_LAYERS];
} mz_packed_mlp;

/* Host-only: computes the packed layout of `h_mlp` starting at float offset `base_off`;
 * returns the number of floats it occupies, 4 KiB of read slack at the end included
 * (negative on error). */
int64_t mz_mlp_plan(const mz_mlp *h_mlp, int64_t base_off, mz_packed_mlp *h_out);
/* Re-tiles the weights named by `h_mlp` (device pointers) into blob[plan offsets]. */
int mz_mlp_pack(const mz_mlp *h_mlp, const mz_packed_mlp *h_plan, float *blob, void *stream);
/* y = MLP(x) for B rows.  x f32 rows of in_features floats, row stride ldx, optional row
 * gather x_rows i32[B] (row r is read from x + x_rows[r]*ldx; negative -> zeros);
 * y f32[B][out_features] with row stride ldy.  A MZ_ACT_SOFTMAX layer normalises over its
 * out_features. */
int mz_mlp_forward(const mz_packed_mlp *h_plan, const float *blob, const float *x, int64_t ldx,
                   const int32_t *x_rows, float *y, int64_t ldy, int32_t B, void *stream);

/* mz_mlp_forward for a policy head, with the post-processing of the search in the same launch:
 * argmax i32[B] (may be NULL) = first maximum of each output row (the action MuZeroSimulation
 * takes, src/search/simulation.py:35), then softmax_times further softmax passes over the row
 * (expand_node's extra softmax, src/search/expansion.py:30) before y is stored.  y may be NULL when
 * only the argmax is wanted.  Same arithmetic as mz_mlp_forward followed by mz_policy_post. */
int mz_mlp_forward_post(const mz_packed_mlp *plan, const float *blob, const float *x, int64_t ldx,
                        const int32_t *x_rows, float *y, int64_t ldy, int32_t B,
                        int32_t softmax_times, int32_t *argmax, void *stream);
/* Two mz_mlp_forward calls in one launch (both packed in the same blob; no softmax / argmax): heads whose inputs
 * are ready at the same moment, e.g. reward_mlp over new children and value_mlp at the end of a rollout
 * (neural_network.py:186-188, :256-262).  Same results as the two calls; batches of 8192 rows or more are
 * run as two launches. */
int mz_mlp_forward_pair(const mz_packed_mlp *plan0, const float *x0, int64_t ldx0, const int32_t *x_rows0,
                        float *y0, int64_t ldy0, int32_t B0, const mz_packed_mlp *plan1, const float *x1,
                        int64_t ldx1, const int32_t *x_rows1, float *y1, int64_t ldy1, int32_t B1,
                        const float *blob, void *stream);

/* g: DynamicsNetwork.forward for MLP-only configs, src/neural_networks/neural_network.py:152-188
 *    next = relu(fc_merge([latent ; action_embedding[action]])), reward = reward_mlp(next).
 * f: PredictionNetwork.forward, :234-262: value = value_mlp(latent), policy = policy_mlp(latent)
 *    (policy_mlp ends in a softmax layer). */
typedef struct mz_dynamics_net {
    const float *embedding;    /* action_embedding.weight [A][D]   */
    const float *merge_weight; /* fc_merge.weight [D][2D]          */
    const float *merge_bias;   /* fc_merge.bias [D]                */
    mz_mlp reward;             /* reward_mlp, D -> ... -> 1        */
    int32_t D, A;
} mz_dynamics_net;

typedef struct mz_prediction_net {
    mz_mlp value;  /* value_mlp, D -> ... -> 1                      */
    mz_mlp policy; /* policy_mlp, D -> ... -> A (last act softmax)  */
    int32_t D, A;
} mz_prediction_net;

typedef struct mz_packed_nets {
    int32_t D, A, Dp, _pad;
    int64_t total_floats;
    int64_t action_table_off; /* [A][Dp]: fc_merge.weight[:, D:] @ embedding[a] + fc_merge.bias */
    mz_packed_mlp merge;      /* one linear layer: fc_merge.weight[:, :D], no bias              */
    mz_packed_mlp reward, value, policy;
} mz_packed_nets;

/* Host-only layout computation; fills *h_out including total_floats (blob size). */
int mz_nets_plan(const mz_dynamics_net *h_dyn, const mz_prediction_net *h_pred, mz_packed_nets *h_out);
/* Fills blob (total_floats floats of device memory) from the networks' device weights. */
int mz_nets_pack(const mz_dynamics_net *h_dyn, const mz_prediction_net *h_pred,
                 const mz_packed_nets *h_plan, float *blob, void *stream);

/* Batched g for B rows: latent rows gathered like mz_mlp_forward's x; action i32[B] or NULL
 * (NULL: action of row r is r % A, the expansion pattern where consecutive groups of A rows
 * share one latent -- pass x_rows accordingly).  next_latent f32[B][D] row stride ld_next
 * with optional row scatter next_rows i32[B]; reward f32[B]. */
int mz_dynamics_forward(const mz_packed_nets *h_plan, const float *blob, const float *latent,
                        int64_t ld_latent, const int32_t *latent_rows, const int32_t *action,
                        float *next_latent, int64_t ld_next, const int32_t *next_rows,
                        float *reward, int32_t B, void *stream);

/* Batched f for B rows.  policy f32[B][A] (softmax_times extra softmax passes applied on
 * top of the network's own: 0 = network output, 1 = the prior expand_node stores,
 * src/search/expansion.py:26-27), value f32[B], argmax i32[B] (first maximum of the network
 * policy, src/search/simulation.py:36).  NULL outputs are skipped, including their head. */
int mz_prediction_forward(const mz_packed_nets *h_plan, const float *blob, const float *latent,
                          int64_t ld_latent, const int32_t *latent_rows, int32_t softmax_times,
                          float *policy, float *value, int32_t *argmax, int32_t B, void *stream);

/* ------------------------------------------------------------------------------------
 * Connect Four on the GPU: the environment behind src/environments/connect_four.py:18-72
 * (PettingZoo connect_four_v3 seen through the reference's adapter).
 *   pieces  u64[G][2]  bitboards of player_0 / player_1 (bit = column*7 + height)
 *   to_move i8[G]      0 = player_0 (to_play +1), 1 = player_1 (to_play -1)
 *   done    i8[G]
 * ------------------------------------------------------------------------------------ */
typedef struct mz_c4_state {
    int32_t G, _pad;
    unsigned long long *pieces;
    int8_t *to_move;
    int8_t *done;
} mz_c4_state;

/* ConnectFour.reset, connect_four.py:59-66.  mask i8[G] or NULL: reset only games with
 * mask != 0. */
int mz_c4_reset(const mz_c4_state *state, const int8_t *mask, void *stream);
/* ConnectFour.step, connect_four.py:43-51: action i32[G]; reward f32[G] and done i8[G] are
 * those of the player to move AFTER the step (a winning move reports -1).  A full column or
 * out-of-range action ends the game (mover -1, opponent 0, player_0's view reported).
 * Finished games are left unchanged and report reward 0, done 1. */
int mz_c4_step(const mz_c4_state *state, const int32_t *action, float *reward, int8_t *done,
               void *stream);
/* ConnectFour.get_state + get_to_play, connect_four.py:26-30,53-57: obs f32[G][2][6][7]
 * (plane 0 = player to move, row 0 = top), to_play i8[G] in {+1,-1} (may be NULL). */
int mz_c4_observe(const mz_c4_state *state, float *obs, int8_t *to_play, void *stream);
/* The end of a lock-step self-play move in one launch (src/training_data_generator.py:107-142 for every game):
 * mz_pick_action on policy f32[G][A] / explore / random_action, mz_c4_step with that action, mz_episode_end on
 * step_in_episode with max_steps, and mz_c4_reset of the games that ended -- the results of those four calls in a
 * row.  action i32[G] and action_copy (optional, i32[G]: a second copy for the caller's record of the move) receive
 * the action, reward f32[G] and ended i8[G] as mz_c4_step / mz_episode_end leave them. */
int mz_c4_move_end(const mz_c4_state *state, const float *policy, int32_t A, const int8_t *explore,
                   const int32_t *random_action, int32_t *step_in_episode, int32_t max_steps, int32_t *action,
                   int32_t *action_copy, float *reward, int8_t *ended, void *stream);

/* ------------------------------------------------------------------------------------
 * Data side of self-play and of the learner's targets.
 * ------------------------------------------------------------------------------------ */

/* FrameRingBuffer.fill/add + make_history_tensor, src/ring_buffer.py:11-83, for G games.
 * ring f32[G][K][C*H*W] and act_ring i32[G][K] persist between calls (caller-owned).
 * frame f32[G][C][H][W] is this step's observation, action i32[G] the action id stored with
 * it (NULL = 0, as self-play does: src/training_data_generator.py:110).  step i32[G] = moves
 * already played in the game's episode: 0 replicates the frame into every slot (fill),
 * t > 0 overwrites slot (t-1) % K (add).  out f32[G][C*K+K][H][W] = frames in SLOT order on
 * the channel axis, then K planes filled with each slot's action id. */
int mz_history_push_assemble(float *ring, int32_t *act_ring, const float *frame,
                             const int32_t *action, const int32_t *step, int32_t G, int32_t K,
                             int32_t C, int32_t H, int32_t W, float *out, void *stream);

/* RepresentationNetwork.forward for `downsample: [], res_net: []` configs whose latent H, W
 * equal the observation's: a 1x1 convolution (final_conv) followed by an identity pool,
 * src/neural_networks/neural_network.py:83-102.  x f32[G][Cin][P], weight f32[Cout][Cin],
 * bias f32[Cout] or NULL, out f32[G][Cout][P] (P = H*W). */
int mz_repr_pointwise(const float *x, const float *weight, const float *bias, int32_t G,
                      int32_t Cin, int32_t Cout, int32_t P, float *out, void *stream);

/* Greedy / exploratory action pick of _select_action, src/training_data_generator.py:148-163:
 * action[g] = explore[g] ? random_action[g] : first argmax of policy[g][0..A).  The caller
 * draws explore (random() < epsilon) and random_action on the host.  explore may be NULL. */
int mz_pick_action(const float *policy, const int8_t *explore, const int32_t *random_action,
                   int32_t G, int32_t A, int32_t *action, void *stream);

/* The random inputs of one lock-step move of G games, drawn on the device in one launch (the reference draws them game
 * after game from Python's and torch's global generators): rand_child i32[G][draws_per_game], the child every
 * simulation starts from (src/search/mcts.py:64-66 with expansion.py:35: uniform over the A children; the root
 * expansion's slot is drawn too); eta f32[G][A], the root's Dirichlet(alpha) noise (mcts.py:85-94); explore i8[G] =
 * uniform < *epsilon and random_action i32[G] ~ p_random (f32[A], NULL: uniform), the coin and the action of
 * _select_action (src/training_data_generator.py:148-163).  Counter-based generator (Philox4x32-10): the streams are
 * a function of (seed, move number, game) only.  state: device i64[2], zero before the first call -- the move
 * number, advanced by the launch itself (so a recorded launch draws fresh numbers at every replay), and a work
 * counter; epsilon: device f32[1].  Same distributions as the reference's draws, not the same streams. */
int mz_move_draws(uint64_t seed, int64_t *state, const float *epsilon, const float *p_random, float alpha, int32_t G,
                  int32_t A, int32_t draws_per_game, int32_t *rand_child, float *eta, int8_t *explore,
                  int32_t *random_action, void *stream);

/* End of a lock-step move for G games: step_in_episode[g] += 1; ended[g] = done[g] or the counter reached max_steps
 * (the episode loop of src/training_data_generator.py:107-142: `for step in range(max_steps_per_episode)` with
 * `if done: break`); the counter of an ended game returns to 0. */
int mz_episode_end(const int8_t *done, int32_t *step_in_episode, int32_t max_steps, int32_t G, int8_t *ended, void *stream);

/* Host-only: h_out[j] = gamma^(j-1) for j = 0..w+1 with the C library's pow (the factors
 * _compute_z_targets uses, src/training.py:236-240). */
int mz_gamma_powers(double gamma, int32_t w, double *h_out);
/* _compute_z_targets, src/training.py:223-242, for B unroll windows: R, V f64[B][w+1]
 * (chunks k..k+w), pow_tab = device copy of mz_gamma_powers, z f64[B][w+1]. */
int mz_z_targets(const double *R, const double *V, const double *pow_tab, int32_t B, int32_t w,
                 double *z, void *stream);

/* ------------------------------------------------------------------------------------
 * Convolutional representation network h: RepresentationNetwork.forward
 * (src/neural_networks/neural_network.py:83-102) built from ConvLayer, PoolLayer and ResBlock
 * (src/neural_networks/network_builder.py:32-151), for G games per call.  Activations between
 * these entry points are channels-last, f32[G][H][W][C]; the host side composes them
 * (muzero_amd/conv_repr.py).
 * ------------------------------------------------------------------------------------ */

/* Layout changes at the two ends: the history tensor of mz_history_push_assemble is [G][C][P]
 * (P = H*W), the latent the search reads is [G][c][h*w]. */
int mz_nchw_to_nhwc(const float *x, int32_t G, int32_t C, int32_t P, float *y, void *stream);
int mz_nhwc_to_nchw(const float *x, int32_t G, int32_t C, int32_t P, float *y, void *stream);

/* nn.Conv2d as ONE matrix product (network_builder.py:39-45,87-105,110-118): the pixel index table
 * idx i32[KS*KS][G*Ho*Wo] names, for every tap (ky, kx) and output pixel (g, oy, ox), the input pixel
 * (g*H + iy)*W + ix it multiplies, or -1 in the zero padding; mz_wide_forward_gather then evaluates
 *   y[row][co] = act(sum over taps and ci of x[idx[tap][row]][ci] * Wk[co][tap][ci] + bias[co])
 * with Wk the weight in [Cout][KS][KS][Cin] order (mz_conv_weight_kmajor) packed by mz_wide_plan /
 * mz_wide_pack as a Linear layer of KS*KS*Cin inputs.  n_seg = KS*KS segments of Cin floats; Cin must
 * be a multiple of 64 and x 16-byte aligned with row stride ldx >= Cin (a multiple of 4).
 * Workspace as for mz_wide_forward. */
int mz_conv_index(int32_t *idx, int32_t G, int32_t H, int32_t W, int32_t KS, int32_t stride, int32_t pad,
                  int32_t Ho, int32_t Wo, void *stream);
int mz_conv_weight_kmajor(const float *weight, int32_t Cout, int32_t Cin, int32_t KS, float *out, void *stream);
int mz_wide_forward_gather(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                           const int32_t *x_rows, int32_t n_seg, float *y, int64_t ldy, int32_t B,
                           float *workspace, int64_t workspace_floats, void *stream);
/* The same convolution evaluated directly (one lane per output value) for the layers the matrix
 * product does not cover: input channels no multiple of 64, or very few output channels.
 * weight_kmajor f32[Cout][KS][KS][Cin], bias f32[Cout] or NULL, y f32[G][Ho][Wo][Cout]. */
int mz_conv_direct_nhwc(const float *x, const float *weight_kmajor, const float *bias, int32_t G, int32_t H,
                        int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t pad,
                        int32_t activation, float *y, void *stream);

/* The same convolutions at the fp16 matrix rate with float32 accuracy.  A float32 value v is held as two
 * fp16 terms, hi = fp16(v) and lo = fp16((v - hi) * 2^11) (v = hi + lo * 2^-11 to 2^-22 |v|); a product x*w is,
 * up to the lo*lo term of relative size 2^-22, 2^-11 (x_hi (w_hi 2^11) + x_hi w_lo + x_lo w_hi): three
 * v_mfma_f32_16x16x32_f16 into one float32 accumulator -- 16/3 of the fp32 matrix rate.  Operands must stay
 * below 65504 in magnitude (weights below 32): beyond that the results are NaN, not silently wrong.  Split operands
 * in memory are TWO planes of fp16 bit patterns (u16), plane p at element offset p * plane_stride (the "3" in the
 * entry names is historical: it stood for three bf16 planes).
 *   mz_split3 / mz_unsplit3   float32 <-> planes, element-wise (the planes hold 22 significant bits);
 *   mz_nchw_split3            mz_nchw_to_nhwc and mz_split3 in one pass: the channels-first network input
 *       f32[G][C][P] (the argument of RepresentationNetwork.forward, neural_network.py:83) -> planes [G * P][C];
 *   mz_conv3_forward          nn.Conv2d as in mz_wide_forward_gather: x3 planes [input pixels][C], idx the
 *       table of mz_conv_index (i32[taps][rows]), w3 the split of the k-major weight f32[N][taps*C]
 *       (mz_conv_weight_kmajor), bias f32[N] or NULL; output rows x N as float32 (y) and / or split (y3).
 *       C must be a multiple of 64 and N of 128;
 *   mz_conv3_same_forward     the same product for stride-1 convolutions with kernel size 1 or 3 and padding
 *       (k-1)/2 on G images of H x W pixels (ResBlock.conv1 / conv2, network_builder.py:84-120): the
 *       pixels a tile's taps read are one contiguous range of the flat pixel index, staged in LDS once
 *       per channel chunk instead of once per tap; needs no index table.  Bit-identical to
 *       mz_conv3_forward; W is limited to 63 for k = 3 (wider images, up to 144 pixels, are staged from
 *       the float32 form only: mz_conv3_same_bn_forward);
 *   mz_conv3_same_block_forward  mz_conv3_same_bn_forward whose operand is the output of a whole ResBlock
 *       (network_builder.py:138-145): in_activation(x * in_scale + in_shift + r), r = residual (identity skip: the
 *       block's float32 input) or residual * res_scale + res_shift (the raw 1x1 skip convolution and its BatchNorm),
 *       with mz_bn_apply3's roundings; block_out (optional, f32[G*H*W][C]) receives that output once -- the next
 *       block's skip path needs it -- so the block's bn2 / skip sum / activation costs no pass of its own.  W <= 63;
 *   mz_conv3_same_fits        1 when mz_conv3_same_forward (raw = 0) or mz_conv3_same_bn_forward (raw = 1;
 *       with_bn = 1: in_scale / in_shift given; 2: mz_conv3_same_block_forward with an identity skip; 3: with the
 *       skip path's scale / shift) takes the shape -- image width and the LDS the staged range and
 *       the scale / shift rows need -- else 0; sets no error.  Callers choose between the staged kernels,
 *       mz_conv3_forward and a separate mz_bn_apply3 with it instead of repeating the kernel's limits;
 *   mz_conv3_same_bn_forward  mz_conv3_same_forward whose input is the raw float32 output x of the previous
 *       convolution with that layer's per-game BatchNorm scale / shift f32[G][C] and activation: the operand
 *       in_activation(x * in_scale + in_shift) is formed and split while it is staged (conv2(relu(bn1(.))) of
 *       a ResBlock, network_builder.py:131-136), bit-identical to mz_bn_apply3 followed by
 *       mz_conv3_same_forward.  in_scale = in_shift = NULL: the operand is x itself, split while staged (a
 *       float32 activation needs no split copy in memory);
 *   block_sums (all three convolutions, optional): f64[rows / 64][N][2] receives the sum and the sum of squares
 *       of every output column over each block of 64 consecutive rows (rows must be a multiple of 64);
 *       mz_bn_stats_from_sums turns them into the per-game BatchNorm scale / shift (and mean / variance) of
 *       mz_bn_stats when a game has a multiple of 64 pixels -- the statistics pass over the convolution
 *       output (nn.BatchNorm2d in train mode, network_builder.py:94,104,118) then costs no read of it;
 *   mz_bn_apply3              mz_bn_apply on a raw float32 convolution output, with the residual either
 *       float32 (`residual`, optionally with the skip path's scale / shift) or split (`residual3`, the
 *       block's own input), writing float32 and / or split. */
int mz_split3(const float *x, int64_t total, uint16_t *out, int64_t plane_stride, void *stream);
int mz_nchw_split3(const float *x, int32_t G, int32_t C, int32_t P, uint16_t *out, int64_t plane_stride, void *stream);
int mz_unsplit3(const uint16_t *x3, int64_t plane_stride, int64_t total, float *out, void *stream);
int mz_conv3_forward(const uint16_t *x3, int64_t x_plane, int32_t C, const int32_t *idx, int32_t taps,
                     const uint16_t *w3, int64_t w_plane, const float *bias, int32_t activation, float *y,
                     uint16_t *y3, int64_t y_plane, int32_t rows, int32_t N, double *block_sums, void *stream);
/* mz_conv3_forward on float32 input rows x[input pixels][C], split while they are staged: whole 128-byte lines per
 * gathered chunk instead of two 64-byte pieces, and no split copy in memory (the strided ConvLayers,
 * network_builder.py:33-60, whose input nothing else reads in split form).  Identical bits to mz_split3 +
 * mz_conv3_forward. */
int mz_conv3_forward_f32(const float *x, int32_t C, const int32_t *idx, int32_t taps, const uint16_t *w3, int64_t w_plane,
                         const float *bias, int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t rows,
                         int32_t N, double *block_sums, void *stream);
int mz_conv3_same_forward(const uint16_t *x3, int64_t x_plane, int32_t C, int32_t G, int32_t H, int32_t W,
                          int32_t ksize, const uint16_t *w3, int64_t w_plane, const float *bias,
                          int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t N, double *block_sums,
                          void *stream);
int mz_conv3_same_bn_forward(const float *x, const float *in_scale, const float *in_shift, int32_t in_activation,
                             int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, const uint16_t *w3,
                             int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3,
                             int64_t y_plane, int32_t N, double *block_sums, void *stream);
int mz_conv3_same_fits(int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, int32_t N, int32_t raw, int32_t with_bn);
int mz_conv3_same_block_forward(const float *x, const float *in_scale, const float *in_shift, const float *residual,
                                const float *res_scale, const float *res_shift, int32_t in_activation, float *block_out,
                                int32_t C, int32_t G, int32_t H, int32_t W, int32_t ksize, const uint16_t *w3,
                                int64_t w_plane, const float *bias, int32_t activation, float *y, uint16_t *y3,
                                int64_t y_plane, int32_t N, double *block_sums, void *stream);
int mz_bn_stats_from_sums(const double *block_sums, int32_t G, int32_t P, int32_t C, const float *weight,
                          const float *bias, float eps, float *scale, float *shift, float *mean, float *var,
                          void *stream);
int mz_bn_apply3(const float *x, const float *scale, const float *shift, const float *residual,
                 const float *res_scale, const float *res_shift, const uint16_t *residual3, int64_t res_plane,
                 int32_t activation, float *y, uint16_t *y3, int64_t y_plane, int32_t G, int32_t P, int32_t C,
                 void *stream);

/* nn.BatchNorm2d in train mode with the batch statistics of ONE game per call, as the reference
 * evaluates h (src/training_data_generator.py:117 calls it with batch size 1 and no network ever
 * enters eval mode): per (game, channel) the mean and biased variance over the P positions of
 * x f32[G][P][C], folded with the affine parameters into scale = weight / sqrt(var + eps) and
 * shift = bias - mean * scale, both f32[G][C].  mean / var f32[G][C] are optional outputs. */
int mz_bn_stats(const float *x, int32_t G, int32_t P, int32_t C, const float *weight, const float *bias,
                float eps, float *scale, float *shift, float *mean, float *var, void *stream);
/* y = act(x * scale[g][c] + shift[g][c] + r) with r = 0 (residual NULL), the residual itself
 * (identity skip of a ResBlock, network_builder.py:139-142) or residual * res_scale + res_shift (the
 * BatchNorm of the 1x1 skip convolution, :108-120).  y may alias x. */
int mz_bn_apply(const float *x, const float *scale, const float *shift, const float *residual,
                const float *res_scale, const float *res_shift, int32_t activation, float *y, int32_t G,
                int32_t P, int32_t C, void *stream);
/* The running statistics a BatchNorm2d holds after those G calls in game order (momentum m,
 * unbiased variance of the P values of each call): running <- (1 - m) running + m stat, G times.
 * mean / var as written by mz_bn_stats; running_mean / running_var f32[C] are updated in place. */
int mz_bn_running_update(const float *mean, const float *var, int32_t G, int32_t C, int32_t P, float momentum,
                         float *running_mean, float *running_var, void *stream);

/* nn.MaxPool2d / nn.AvgPool2d with a square window, no padding (PoolLayer, network_builder.py:53-70;
 * the optional pooling of a ResBlock, :123-129).  y f32[G][Ho][Wo][C], Ho = (H - kernel)/stride + 1. */
int mz_pool2d_nhwc(const float *x, int32_t G, int32_t H, int32_t W, int32_t C, int32_t kernel, int32_t stride,
                   int32_t is_max, float *y, void *stream);
/* F.adaptive_avg_pool2d to the latent's h x w (neural_network.py:100-101), written as the
 * f32[G][C][h][w] latent the search reads. */
int mz_adaptive_avg_to_nchw(const float *x, int32_t G, int32_t H, int32_t W, int32_t C, int32_t h, int32_t w,
                            float *y, void *stream);

/* ------------------------------------------------------------------------------------
 * Replay side of the learner on the device: the prioritised episode sampling of
 * ReplayBuffer.sample_batch / update_priorities (src/replay_buffer.py:72-130) over the priorities of a
 * fixed-shape episode store, and the cut of training windows (src/training.py:131-160, history as
 * assembled by bptt, :252-262).
 * ------------------------------------------------------------------------------------ */

/* sample_batch for n stored episodes: P(i) = priority[i]^alpha / sum (uniform when the sum is below
 * 1e-8), B draws with replacement by numpy's inverse-CDF search (RandomState.choice: cumulative sum
 * normalised by its last entry, searchsorted(side="right")) on the caller's uniform numbers u f64[B]
 * (what RandomState.random_sample(B) returns), weight f64[B] = (n P(draw))^-beta / max over the draw.
 * probs / cdf f64[n] are outputs (and the kernel's scratch).
 * Optional window choice (done_len i64[n] != NULL; episode lengths): a draw whose episode has fewer
 * than `need` = lookback + roll_ahead + 1 chunks is dropped (src/training.py:210-221) -- it stays in
 * the batch as a window of the longest stored episode with weight32 = 0 and slot = cap -- and
 * start = lookback + floor(u_start * span) over the span admissible window starts (:139-142);
 * cut_from i32[B] is the episode row to cut the window from, slot i32[B] what mz_per_update takes. */
int mz_per_sample(const double *priority, int32_t n, double alpha, double beta, const double *u, int32_t B,
                  double *probs, double *cdf, int32_t *draw, double *weight, const int64_t *done_len,
                  int32_t need, int32_t lookback, int32_t roll_ahead, int32_t cap, const double *u_start,
                  float *weight32, int32_t *slot, int32_t *cut_from, int32_t *start, void *stream);
/* update_priorities: priority[slot[b]] = |error[b]| + 1e-6 in draw order (the last occurrence of a
 * slot wins), *max_priority follows; entries with slot outside [0, cap) are ignored. */
int mz_per_update(double *priority, double *max_priority, const int32_t *slot, const float *error, int32_t B,
                  int32_t cap, void *stream);
/* B training windows (rows[b], k[b]) out of the store obs f32[rows][T][C][H][W], action i64[rows][T],
 * policy f32[rows][T][A], value f64[rows][T], reward f32[rows][T]: history f32[B][C*K+K][H][W] (the
 * K-slot ring filled with the first of the `lookback` past frames, the later ones added; frames in
 * slot order, then K planes with each slot's action id), actions i64[B][w+1], policies f32[B][w+1][A],
 * rewards f32[B][w+1] of chunks k .. k+w, and the value targets z f32[B][w+1] of _compute_z_targets
 * (src/training.py:223-242; pow_tab = device copy of mz_gamma_powers). */
int mz_replay_cut_windows(const float *obs, const int64_t *action, const float *policy, const double *value,
                          const float *reward, int32_t T, int32_t C, int32_t H, int32_t W, int32_t A,
                          const int32_t *rows, const int32_t *k, int32_t B, int32_t lookback,
                          int32_t roll_ahead, int32_t K, const double *pow_tab, float *history,
                          int64_t *actions, float *policies, float *z, float *rewards, void *stream);

/* The learner's unroll loss for B windows in one launch, with its gradients with respect to the
 * network outputs: policy_loss / value_loss / reward_loss of src/training.py:307-320 summed over the
 * steps of bptt (:275-303; the final step adds value and policy, :297-303).
 *   pred_p f32[w+1][B][A] (the networks' softmaxed policies, used as logits by F.cross_entropy),
 *   pred_v f32[w+1][B], pred_r f32[w][B]: step-major, as the batched unroll produces them;
 *   target_p f32[B][w+1][A], z f32[B][w+1], r f32[B][w+1] (its first w columns are used), is_weights f32[B].
 * per_window f32[B] = cp * sum CE + cv * sum (v - z)^2 + cr * sum (r_pred - r)^2 (the |loss| that
 * becomes the episode's priority, :171), parts f32[3][B] its three terms, sums f32[4] =
 * [sum_b is_weights_b * per_window_b, sum policy, sum value, sum reward] (fixed summation order),
 * g_p / g_v / g_r = d sums[0] / d pred_p / pred_v / pred_r in the predictions' layouts. */
int mz_unroll_loss(const float *pred_p, const float *pred_v, const float *pred_r, const float *target_p,
                   const float *z, const float *r, const float *is_weights, int32_t B, int32_t w, int32_t A,
                   float cp, float cv, float cr, float *per_window, float *parts, float *sums, float *g_p,
                   float *g_v, float *g_r, void *stream);

/* ------------------------------------------------------------------------------------
 * Whole search in one launch: MCTS.run (src/search/mcts.py:44-74) for all G trees of the pool
 * -- root creation, root expansion, prior perturbation, S rounds of select / expand / simulate
 * / backup, and the result -- with every network evaluation fused in (MLP-only g and f).
 * Produces the same pool contents as the per-stage entry points driven in lock step.
 * ------------------------------------------------------------------------------------ */
typedef struct mz_search_params {
    int32_t S;              /* simulations per tree (MCTSConfig.max_iterations)                */
    int32_t depth;          /* rollout steps (MCTSConfig.model_look_ahead)                     */
    int32_t tab_len;        /* entries of sqrt_tab / log_tab, >= S + 2                         */
    int32_t lanes_per_tree; /* filled in by the library                                        */
    double discount;        /* MCTSConfig.discount_factor                                      */
    double noise_frac;      /* MCTSConfig.noise_frac                                           */
    const double *sqrt_tab; /* device copies of mz_puct_tables                                 */
    const double *log_tab;
    const float *root_latent;   /* f32[G][D]   h(observation history)                          */
    const int8_t *root_to_play; /* i8[G]                                                       */
    const float *eta;           /* f32[G][A]   Dirichlet sample of src/search/mcts.py:91       */
    const int32_t *rand_child;  /* i32[G][S+1] random.choice draws, column 0 = root expansion  */
    int32_t *leaf_trace;        /* optional i32[G][S]: node selected by each simulation        */
    int32_t *child_trace;       /* optional i32[G][S]: child the simulation started from       */
    float *rollout_trace;       /* optional f32[G][S][depth+1]: [r_1..r_depth, v] per simulation */
    float *prior_trace;         /* optional f32[G][S+1][A]: priors written by each expansion   */
    unsigned long long *stamps; /* optional u64[32]: shader-clock cycles workgroup 0 spent per
                                   stage (diagnostics; leave NULL in production)               */
    float *z_pool;              /* f32[G][S+1][D], required when pool->latent is NULL: per tree
                                   and expansion the ONE row z = fc_merge.weight[:, :D] latent
                                   from which the A children's latents are relu(z + table[a]).
                                   With pool->latent given it is an optional workspace: searches
                                   of the shape the sequential-program kernel covers then run in
                                   that form and write every node's latent as well (same bits);
                                   NULL selects the general kernel                              */
} mz_search_params;

/* Requires pool->cap >= 1 + (S+1)*A, A <= 32, and either pool->latent (every node's latent is
 * stored, as Node.latent_state holds it: src/search/expansion.py:25-33) or prm->z_pool with
 * pool->latent == NULL (implicit child latents: an expansion stores one row instead of A, and a
 * selected leaf's latent is formed from its parent's row when it is expanded -- the same float32
 * values, so both forms leave bit-identical trees and results; only the drop-in MCTS.run, whose
 * callers read node latents, needs the first).  policy f64[G][A], policy32 f32[G][A], value
 * f64[G], visits i32[G][A] as in mz_tree_result (any may be NULL). */
int mz_search_run(const mz_pool *pool, const mz_packed_nets *plan, const float *blob,
                  const mz_search_params *prm, double *policy, float *policy32, double *value,
                  int32_t *visits, void *stream);

/* Schedule choice of the sequential-program kernel behind mz_search_run: where every wavefront of a
 * workgroup has a full share of every layer's column tiles (policy and value hidden widths 241-256,
 * value 2 / reward 1 widths 113-128, latents of 113-176 values -- the benchmark configuration), a
 * schedule written out for those shares runs, with a third register slot for weight tiles and the merge
 * tiles beyond one per wavefront resident in LDS; 0 makes such
 * searches take the general schedule.  Default 1.  Sets the choice unless new_value < 0; returns the
 * previous one.  Both leave bit-identical pools and results; no counterpart in the reference, a
 * cross-checking knob only. */
int mz_search_seq_three_slots(int32_t new_value);

/* ------------------------------------------------------------------------------------
 * Residual-tower networks (`res_net` non-empty; latent [C][H][W] with D = C*H*W up to the
 * shipped 256*6*6 = 9216): the pieces of DynamicsNetwork.forward / PredictionNetwork.forward
 * (src/neural_networks/neural_network.py:152-188, :234-262) that the LDS-resident MLP kernels
 * above do not cover.  The host side composes them (muzero_amd/towers.py).
 * ------------------------------------------------------------------------------------ */

/* One Linear layer of any width (fc_merge and the first layer of each head at D = 9216):
 * same packed tile layout as mz_mlp_pack, no 4096 limit.  Returns the floats it occupies in
 * the blob, read slack included (negative on error). */
int64_t mz_wide_plan(const mz_dense *h_layer, int64_t base_off, mz_packed_layer *h_out);
int mz_wide_pack(const mz_dense *h_layer, const mz_packed_layer *h_plan, float *blob, void *stream);
/* Floats of workspace mz_wide_forward needs for B rows (partial sums of the split
 * reduction dimension; 0 when the layer is not split). */
int64_t mz_wide_workspace(const mz_packed_layer *h_plan, int32_t B);
/* y[r][:] = act(x[x_rows[r]][:] W^T + bias) for B rows; x rows of K floats with stride ldx and
 * optional gather x_rows (negative -> zeros); y f32[B][N] with row stride ldy.  `workspace`
 * holds at least mz_wide_workspace() floats (may be NULL when that is 0). */
int mz_wide_forward(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                    const int32_t *x_rows, float *y, int64_t ldy, int32_t B, float *workspace,
                    int64_t workspace_floats, void *stream);

/* mz_wide_forward for callers that cannot allocate or pass device memory of their own (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in
 * the reference.  The partial sums of the split reduction go to an array built into the library
 * (mz_wide_static_workspace_floats() floats = 16 MiB, one copy per GPU, held until the process ends)
 * instead of a caller-provided workspace; the call allocates nothing.  A call whose
 * mz_wide_workspace() exceeds that size fails with MZ_ERR_ARG.  The device the stream belongs to
 * must be the current device.  All calls on one GPU share the array: calls that overlap in time on
 * the same GPU (different streams or threads) race; calls on one stream are ordered and safe.
 * The array's device address is looked up once per device, on the first call that needs it or in
 * mz_wide_static_init() (current device; no counterpart in the reference either): that look-up is a
 * runtime query which must not fall inside a stream capture, so a caller that captures its first use
 * of a device calls mz_wide_static_init() on that device beforehand. */
int64_t mz_wide_static_workspace_floats(void);
int mz_wide_static_init(void);
int mz_wide_forward_static(const mz_packed_layer *h_plan, const float *blob, const float *x, int64_t ldx,
                           const int32_t *x_rows, float *y, int64_t ldy, int32_t B, void *stream);

/* ResBlock (src/neural_networks/network_builder.py:73-151) with stride 1, padding
 * (kernel_size-1)/2 and no pooling -- the only form that keeps the latent's H x W, which g and f
 * require.  Weights are device pointers in torch layout: conv [out][in][k][k] without bias,
 * BatchNorm2d weight/bias [out]; down_w (1x1 conv [out][in]) and its BatchNorm are required when
 * in_channels != out_channels and NULL for an identity skip.  conv1_bias (optional, [out][H][W]) is
 * added to conv1's output before bn1: a block whose input is itself an affine image of fewer
 * channels (the previous network's res_final) can be given pre-multiplied weights on those
 * channels, and the folded bias then depends on the position because of the zero padding. */
typedef struct mz_resblock {
    const float *conv1_w, *bn1_w, *bn1_b;
    const float *conv2_w, *bn2_w, *bn2_b;
    const float *down_w, *dbn_w, *dbn_b;
    const float *conv1_bias;
    int32_t in_channels, out_channels, kernel_size, activation;
    float eps1, eps2, epsd, _pad;
} mz_resblock;

#define MZ_MAX_BLOCKS 4

/* res_blocks followed by res_final (1x1 conv with bias back to C channels, or the identity when
 * final_w is NULL): neural_network.py:130-144, :209-221. */
typedef struct mz_tower {
    int32_t n_blocks, C, H, W;
    mz_resblock block[MZ_MAX_BLOCKS];
    const float *final_w; /* [C][last out_channels] or NULL */
    const float *final_b; /* [C] or NULL                    */
} mz_tower;

/* y[y_rows[r]] = res_final(res_blocks(x_r)) for B rows of C*H*W floats.
 * Batch statistics: the reference never leaves train mode, so every BatchNorm2d normalises
 * with the mean and biased variance of ONE network call (src/search/expansion.py:25-28 calls g
 * with the A children of a node as one batch; every other call has batch size 1).  Here
 * consecutive groups of `group` rows form one such call (B must be a multiple of `group`);
 * running statistics are not updated (nothing on the search path reads them; mz_tower_forward_stats exports
 * what an update needs).
 * Input: x_r = x[x_rows[r]] (gather optional); with row_add != NULL the fc_merge split is
 * applied while loading, x_r = relu(x[x_rows[r]] + row_add[row_sel[r]]) (row_sel NULL: r % group).
 * y_rows NULL: row r; negative entries are skipped.
 * t_out (optional, row stride ld_t, same row scatter as y): the last block's output, i.e. the
 * input of res_final, [last out_channels][H][W] per row.  res_final is linear, so a consumer
 * whose first operation on the latent is linear as well (fc_merge, the first Linear of a head)
 * can be evaluated on these few values with pre-multiplied weights instead of on the C*H*W
 * latent.  y may be NULL when only t_out is wanted. */
int mz_tower_forward(const mz_tower *h_tower, const float *x, int64_t ldx, const int32_t *x_rows,
                     const float *row_add, int64_t ld_add, const int32_t *row_sel, float *y,
                     int64_t ldy, const int32_t *y_rows, float *t_out, int64_t ld_t, int32_t B,
                     int32_t group, void *stream);

/* mz_tower_forward that also exports every BatchNorm2d's batch statistics, so that a caller can apply
 * the side effect the reference's train-mode calls have on running_mean / running_var
 * (network_builder.py:94,104,118): bn_stats receives, for block b in order and for each of its three
 * BatchNorm layers (bn1, bn2, the skip path's; the last left untouched for an identity skip),
 * mean f32[calls][out_channels] followed by the biased variance f32[calls][out_channels], calls = B / group
 * in row order -- mz_tower_stats_floats(tower, calls) floats in all.  mz_bn_running_update on a layer's two
 * arrays (G = calls, P = group * H * W) advances its buffers as `calls` consecutive reference calls would.
 * bn_stats NULL: mz_tower_forward. */
int64_t mz_tower_stats_floats(const mz_tower *h_tower, int32_t calls);
int mz_tower_forward_stats(const mz_tower *h_tower, const float *x, int64_t ldx, const int32_t *x_rows,
                           const float *row_add, int64_t ld_add, const int32_t *row_sel, float *y,
                           int64_t ldy, const int32_t *y_rows, float *t_out, int64_t ld_t, int32_t B,
                           int32_t group, float *bn_stats, void *stream);

/* Kernel choice of mz_tower_forward(_stats) for a tower of one block on at most 4 channels whose caller
 * wants the block output only (f on the bottleneck values, neural_network.py:234-262 through the folded
 * res_final): from `calls` reference calls per launch on, one wavefront evaluates a call instead of one
 * workgroup.  Default 1 (always).  Sets the threshold unless new_value < 0; returns the previous one.  Both
 * kernels compute the same block (results agree to 1e-6 of scale: the window sums run in another order);
 * no counterpart in the reference, a tuning and cross-checking knob only. */
int mz_tower_thin_min_calls(int32_t new_value);

/* g on the bottleneck values t of its input latent, up to the first matrix product of its tower, without
 * the C*H*W-wide intermediate in memory (neural_network.py:168-182: fc_merge, reshape, the first
 * ResBlock's conv1 and skip convolution).  h_merge: the latent half of fc_merge composed with the previous
 * res_final (K = values per row of t, N = H*W*C) with its output rows permuted position-major
 * (row p*C + c = latent index c*H*W + p), packed by mz_wide_pack; table f32[A][H*W][C]: the per-action
 * half plus bias in the same order; h_conv: the [C] -> 32 column product (columns co*9 + tap of conv1, then
 * the skip convolution's channels), packed likewise.  action i32[B] (one call per row) or NULL (every row
 * with all A actions, rows of Y then in (row, action) order).  Y f32[rows][H*W][32].
 * mz_tower_tail_forward finishes the block from Y (the tap sum, bn1, conv2, bn2, the skip path's
 * BatchNorm, sum, activation; batch statistics per `group` rows, as mz_tower_forward) into the block
 * output t_out (row scatter y_rows); bn_stats as in mz_tower_forward_stats. */
int mz_merge_conv_forward(const mz_packed_layer *h_merge, const mz_packed_layer *h_conv, const float *blob,
                          const float *t, int64_t ld_t, const int32_t *t_rows, const float *table,
                          const int32_t *action, int32_t A, int32_t B, int32_t C, int32_t HW, float *Y,
                          void *stream);
/* The same pass at the fp16 matrix rate with float32 accuracy: every float32 operand is held as two fp16 terms
 * (hi, lo * 2^11, as the convolutions above) and a product evaluated as three fp16 products in one float32 accumulator
 * (results equal mz_merge_conv_forward's to within 1e-6 of their scale, tests/test_gpu_towers.py; no
 * counterpart in the reference, which evaluates these layers in float32).  merge3 / conv3: the two layers' weights
 * as split operands, written once per weight update by mz_packed_split3 from the packed float32 layers
 * (mz_packed_split3_words 32-bit words each); the other arguments as mz_merge_conv_forward. */
int64_t mz_packed_split3_words(const mz_packed_layer *h_plan);
int mz_packed_split3(const mz_packed_layer *h_plan, const float *blob, uint32_t *out, void *stream);
int mz_merge_conv_forward_split(const mz_packed_layer *h_merge, const uint32_t *merge3, const mz_packed_layer *h_conv,
                                const uint32_t *conv3, const float *t, int64_t ld_t, const int32_t *t_rows,
                                const float *table, const int32_t *action, int32_t A, int32_t B, int32_t C, int32_t HW,
                                float *Y, void *stream);
int mz_tower_tail_forward(const mz_tower *h_tower, const float *Y, int32_t B, int32_t group, float *t_out,
                          int64_t ld_t, const int32_t *y_rows, float *bn_stats, void *stream);

/* table[a][:] = fc_merge.weight[:, D:] @ action_embedding[a] + fc_merge.bias for a < A
 * (f32[A][D]): the per-action half of fc_merge, neural_network.py:168-172. */
int mz_action_table(const float *embedding, const float *merge_weight, const float *merge_bias,
                    int32_t D, int32_t A, float *table, void *stream);

/* Policy post-processing on f32[B][A] rows: `softmax_times` softmax passes (expand_node's
 * softmax of the network's already-normalised policy, src/search/expansion.py:26-27) into
 * policy_out (may alias policy_in, or NULL) and the first maximum of the INPUT row into
 * argmax i32[B] (src/search/simulation.py:36), or NULL. */
int mz_policy_post(const float *policy_in, int32_t B, int32_t A, int32_t softmax_times,
                   float *policy_out, int32_t *argmax, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_HIP_H */
