"""Batched search engine: G independent MCTS trees advanced in lock step on one MI355X.

Host-side orchestration only -- every tree operation and every network evaluation is a
kernel of libmuzero_hip.so reached through muzero_amd._hip (no CPU fallback).  The data
model is the flat node pool of include/muzero_hip.h; PyTorch supplies device memory and
the stream.

The per-move algorithm is the reference's MCTS.run (src/search/mcts.py:44-74) with its
randomness turned into inputs: `eta` (the Dirichlet sample of mcts.py:91) and `rand_child`
(the S+1 random.choice draws of src/search/expansion.py:35, index 0 being the discarded
draw of the root expansion).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from muzero_amd import _hip
from muzero_amd._hip import ptr


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class TreePool:
    """Device storage of G trees x cap nodes, one 32-byte record per node (include/muzero_hip.h,
    mz_node), and the PUCT lookup tables.  `visit`, `value_sum`, `prior`, `reward` and `first_child`
    are strided views of the records (readable and writable in place); `parent` and `to_play` share
    one word of the record, so they are read as decoded copies and written with set_link."""

    def __init__(self, G: int, A: int, num_simulations: int, D: int, device="cuda", with_latent=True,
                 cap: int | None = None):
        self.G, self.A, self.D = G, A, D
        self.cap = cap or (1 + A + num_simulations * A)
        self.device = torch.device(device)
        dev = self.device
        self.node = torch.zeros((G, self.cap, 8), dtype=torch.int32, device=dev)
        assert self.node.data_ptr() % 32 == 0
        as_f64 = self.node.view(torch.float64)   # [G, cap, 4]
        as_f32 = self.node.view(torch.float32)   # [G, cap, 8]
        self.value_sum = as_f64[..., 0]
        self.visit = self.node[..., 2]
        self.link = self.node[..., 3]
        self.prior = as_f64[..., 2]
        self.reward = as_f32[..., 6]
        self.first_child = self.node[..., 7]
        self.link.fill_(0xFFFFFF)      # no parent, to_play 0
        self.first_child.fill_(-1)
        self.n_nodes = torch.zeros(G, dtype=torch.int32, device=dev)
        self.latent = torch.zeros((G, self.cap, D), dtype=torch.float32, device=dev) if with_latent else None
        self.c = _hip.Pool(G, A, self.cap, D, ptr(self.node), ptr(self.n_nodes), ptr(self.latent))
        self.set_table_len(max(num_simulations + 2, 64))

    def set_table_len(self, n: int) -> None:
        sq = np.zeros(n, np.float64)
        lg = np.zeros(n, np.float64)
        _hip.check(_hip.lib().mz_puct_tables(ptr(sq), ptr(lg), n), "mz_puct_tables")
        self.tab_len = n
        self.sqrt_tab = torch.from_numpy(sq).to(self.device)
        self.log_tab = torch.from_numpy(lg).to(self.device)

    @property
    def parent(self) -> torch.Tensor:
        """i32 [G, cap]: parent index of every node, -1 for a root (a decoded copy)."""
        return ((self.link & 0xFFFFFF) ^ 0x800000) - 0x800000

    @property
    def to_play(self) -> torch.Tensor:
        """i8 [G, cap]: Node.to_play of every node (a decoded copy)."""
        return (self.link >> 24).to(torch.int8)

    def set_link(self, where, parent=None, to_play=None) -> None:
        """Writes the parent index and / or to_play of the nodes `where` selects (an index into [G, cap])."""
        word = self.link[where]
        if parent is not None:
            parent = torch.as_tensor(parent, dtype=torch.int32, device=self.device)
            word = (word & ~0xFFFFFF) | (parent & 0xFFFFFF)
        if to_play is not None:
            to_play = torch.as_tensor(to_play, dtype=torch.int32, device=self.device)
            word = (word & 0xFFFFFF) | (to_play << 24)
        self.link[where] = word

    # ---- thin wrappers over the C ABI -------------------------------------------------
    def init(self, root_to_play: torch.Tensor, root_latent: torch.Tensor | None = None) -> None:
        assert root_to_play.dtype == torch.int8 and root_to_play.numel() == self.G
        if root_latent is not None:
            assert root_latent.dtype == torch.float32 and root_latent.numel() == self.G * self.D
            assert root_latent.is_contiguous()
        _hip.check(_hip.lib().mz_tree_init(C.byref(self.c), ptr(root_to_play), ptr(root_latent), _stream()),
                   "mz_tree_init")

    def select(self, out_leaf: torch.Tensor, out_depth: torch.Tensor | None = None, root_f32: bool = True,
               out_row: torch.Tensor | None = None, out_child_rows: torch.Tensor | None = None,
               rand_child: torch.Tensor | None = None, out_drawn_row: torch.Tensor | None = None) -> None:
        """out_row i32[G] / out_child_rows i32[G, A] (optional): pool rows (tree * cap + node) of the
        leaf and of the children an expansion would give it; out_drawn_row i32[G] (with rand_child i32[G]):
        the row of the child expand_commit will draw with the same rand_child."""
        assert out_leaf.dtype == torch.int32 and out_leaf.numel() == self.G
        for t, n in ((out_row, self.G), (out_child_rows, self.G * self.A), (rand_child, self.G), (out_drawn_row, self.G)):
            assert t is None or (t.dtype == torch.int32 and t.numel() == n and t.is_contiguous())
        assert out_drawn_row is None or rand_child is not None
        _hip.check(_hip.lib().mz_tree_select_rows_drawn(C.byref(self.c), ptr(self.sqrt_tab), ptr(self.log_tab), self.tab_len,
                                                        _hip.SELECT_ROOT_F32 if root_f32 else 0, ptr(out_leaf),
                                                        ptr(out_depth), ptr(out_row), ptr(out_child_rows), ptr(rand_child),
                                                        ptr(out_drawn_row), _stream()),
                   "mz_tree_select")

    def expand_commit(self, leaf, prior, reward, rand_child, out_child, next_latent=None, out_child_row=None) -> None:
        assert prior.dtype == torch.float32 and prior.numel() == self.G * self.A and prior.is_contiguous()
        assert reward.dtype == torch.float32 and reward.numel() == self.G * self.A and reward.is_contiguous()
        assert rand_child.dtype == torch.int32 and rand_child.numel() == self.G and rand_child.is_contiguous()
        assert leaf.dtype == torch.int32 and out_child.dtype == torch.int32
        assert out_child_row is None or (out_child_row.dtype == torch.int32 and out_child_row.numel() == self.G)
        if next_latent is not None:
            assert next_latent.dtype == torch.float32 and next_latent.numel() == self.G * self.A * self.D
        _hip.check(_hip.lib().mz_tree_expand_commit_rows(C.byref(self.c), ptr(leaf), ptr(prior), ptr(reward),
                                                         ptr(next_latent), ptr(rand_child), ptr(out_child),
                                                         ptr(out_child_row), _stream()),
                   "mz_tree_expand_commit")

    def root_noise(self, eta: torch.Tensor, noise_frac: float) -> None:
        assert eta.dtype == torch.float32 and eta.numel() == self.G * self.A and eta.is_contiguous()
        _hip.check(_hip.lib().mz_tree_root_noise(C.byref(self.c), ptr(eta), float(noise_frac), _stream()),
                   "mz_tree_root_noise")

    def backup(self, child, leaf, rollout: torch.Tensor, discount: float) -> None:
        assert rollout.dtype == torch.float32 and rollout.is_contiguous() and rollout.shape[0] == self.G
        _hip.check(_hip.lib().mz_tree_backup(C.byref(self.c), ptr(child), ptr(leaf), ptr(rollout),
                                             rollout.shape[1], float(discount), _stream()), "mz_tree_backup")

    def commit_backup_select(self, leaf, prior, reward, rand_child, out_child, rollout, discount, *, select_next: bool,
                             root_f32: bool = True, out_row=None, out_child_rows=None, rand_next=None,
                             out_drawn_row=None, reward0=None) -> None:
        """expand_commit + backup of one step and (select_next) the selection of the next one in one launch; `leaf`
        holds this step's leaves on entry and the next step's on exit."""
        assert rollout.dtype == torch.float32 and rollout.is_contiguous() and rollout.shape[0] == self.G
        assert prior.is_contiguous() and reward.is_contiguous() and prior.numel() == reward.numel() == self.G * self.A
        for t, n in ((leaf, self.G), (rand_child, self.G), (out_child, self.G), (out_row, self.G),
                     (out_child_rows, self.G * self.A), (rand_next, self.G), (out_drawn_row, self.G)):
            assert t is None or (t.dtype == torch.int32 and t.numel() == n and t.is_contiguous())
        assert reward0 is None or (reward0.dtype == torch.float32 and reward0.numel() == self.G and reward0.is_contiguous())
        _hip.check(_hip.lib().mz_tree_commit_backup_select(
            C.byref(self.c), ptr(leaf), ptr(prior), ptr(reward), ptr(rand_child), ptr(out_child), ptr(rollout),
            ptr(reward0), rollout.shape[1], float(discount), ptr(self.sqrt_tab), ptr(self.log_tab), self.tab_len,
            _hip.SELECT_ROOT_F32 if root_f32 else 0, 1 if select_next else 0, ptr(out_row), ptr(out_child_rows),
            ptr(rand_next), ptr(out_drawn_row), _stream()), "mz_tree_commit_backup_select")

    def result(self):
        dev = self.device
        policy = torch.empty((self.G, self.A), dtype=torch.float64, device=dev)
        policy32 = torch.empty((self.G, self.A), dtype=torch.float32, device=dev)
        value = torch.empty(self.G, dtype=torch.float64, device=dev)
        visits = torch.empty((self.G, self.A), dtype=torch.int32, device=dev)
        _hip.check(_hip.lib().mz_tree_result(C.byref(self.c), ptr(policy), ptr(policy32), ptr(value), ptr(visits),
                                             _stream()), "mz_tree_result")
        return policy, policy32, value, visits


class PackedNetworks:
    """Device copy of the g/f weights in the MFMA tile layout of the fused kernels.

    Accepts DynamicsNetwork / PredictionNetwork modules whose `res_net` is empty (the MLP
    configurations); call `refresh()` after the weights change."""

    def __init__(self, dynamics: torch.nn.Module, prediction: torch.nn.Module, device="cuda"):
        if len(dynamics.res_blocks) or len(prediction.res_blocks):
            raise NotImplementedError("the fused HIP networks cover res_net: [] configurations; "
                                      "use make_networks() / muzero_amd.towers.TowerNetworks")
        if not isinstance(dynamics.res_final, torch.nn.Identity) or not isinstance(prediction.res_final, torch.nn.Identity):
            raise NotImplementedError("res_final must be the identity for MLP-only networks")
        self.device = torch.device(device)
        self.dynamics_net, self.prediction_net = dynamics, prediction
        c, h, w = dynamics.latent_shape
        self.D, self.A = c * h * w, dynamics.num_actions
        self._keep: list[torch.Tensor] = []
        self.plan = _hip.PackedNets()
        self.blob = None
        self.refresh()

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _mlp(self, seq: torch.nn.Sequential) -> _hip.Mlp:
        m = _hip.Mlp()
        mods = list(seq)
        i = n = 0
        while i < len(mods):
            lin = mods[i]
            assert isinstance(lin, torch.nn.Linear), f"unexpected module {type(lin)} in MLP"
            act = mods[i + 1] if i + 1 < len(mods) else torch.nn.Identity()
            name = {torch.nn.Identity: "none", torch.nn.ReLU: "relu", torch.nn.Tanh: "tanh",
                    torch.nn.Sigmoid: "sigmoid", torch.nn.Softmax: "softmax"}[type(act)]
            w = self._dev(lin.weight)
            b = self._dev(lin.bias) if lin.bias is not None else None
            m.layer[n] = _hip.Dense(ptr(w), ptr(b), lin.in_features, lin.out_features, _hip.ACT[name], 0)
            n += 1
            i += 2
        m.n_layers = n
        return m

    def refresh(self) -> None:
        self._keep = []
        g, f = self.dynamics_net, self.prediction_net
        dyn = _hip.DynamicsNet(ptr(self._dev(g.action_embedding.weight)), ptr(self._dev(g.fc_merge.weight)),
                               ptr(self._dev(g.fc_merge.bias)), self._mlp(g.reward_mlp), self.D, self.A)
        pred = _hip.PredictionNet(self._mlp(f.value_mlp), self._mlp(f.policy_mlp), self.D, self.A)
        lib = _hip.lib()
        _hip.check(lib.mz_nets_plan(C.byref(dyn), C.byref(pred), C.byref(self.plan)), "mz_nets_plan")
        if self.blob is None or self.blob.numel() != self.plan.total_floats:
            self.blob = torch.empty(self.plan.total_floats, dtype=torch.float32, device=self.device)
        _hip.check(lib.mz_nets_pack(C.byref(dyn), C.byref(pred), C.byref(self.plan), ptr(self.blob), _stream()),
                   "mz_nets_pack")

    # ---- batched g / f ------------------------------------------------------------------
    def dynamics(self, latent, B, *, ld=None, rows=None, action=None, next_latent=None, ld_next=None, next_rows=None,
                 reward=None) -> None:
        _hip.check(_hip.lib().mz_dynamics_forward(
            C.byref(self.plan), ptr(self.blob), ptr(latent), ld or self.D, ptr(rows), ptr(action), ptr(next_latent),
            ld_next or self.D, ptr(next_rows), ptr(reward), B, _stream()), "mz_dynamics_forward")

    def prediction(self, latent, B, *, ld=None, rows=None, softmax_times=0, policy=None, value=None, argmax=None) -> None:
        _hip.check(_hip.lib().mz_prediction_forward(
            C.byref(self.plan), ptr(self.blob), ptr(latent), ld or self.D, ptr(rows), softmax_times, ptr(policy),
            ptr(value), ptr(argmax), B, _stream()), "mz_prediction_forward")


def make_networks(dynamics: torch.nn.Module, prediction: torch.nn.Module, device="cuda", track_running_stats: bool = False,
                  root_final: torch.nn.Module | None = None):
    """Device evaluator for a (g, f) pair: `PackedNetworks` (fused kernels, single-launch search)
    for `res_net: []` configurations, `muzero_amd.towers.TowerNetworks` when either network has
    residual blocks or a res_final convolution (`track_running_stats`: its BatchNorm layers' running
    statistics advance with every call as in the reference; the MLP networks have none)."""
    plain = (len(dynamics.res_blocks) == 0 and len(prediction.res_blocks) == 0 and
             isinstance(dynamics.res_final, torch.nn.Identity) and isinstance(prediction.res_final, torch.nn.Identity))
    if plain:
        return PackedNetworks(dynamics, prediction, device=device)
    from muzero_amd.towers import TowerNetworks
    return TowerNetworks(dynamics, prediction, device=device, track_running_stats=track_running_stats, root_final=root_final)


@dataclass
class SearchResult:
    policy: torch.Tensor    # f64 [G, A]  softmax of root children's value sums
    policy32: torch.Tensor  # f32 [G, A]  the tensor the reference stores in a Chunk
    value: torch.Tensor     # f64 [G]     root mean value
    visits: torch.Tensor    # i32 [G, A]


class BatchedMCTS:
    """MCTS.run for G trees at once (PUCT selection, `max_iterations` mode)."""

    def __init__(self, nets: PackedNetworks, G: int, num_simulations: int, model_look_ahead: int,
                 discount: float, noise_frac: float, device="cuda", keep_latents: bool = True,
                 merge_row_workspace: bool = True):
        """keep_latents=False: with networks in bottleneck form (muzero_amd.towers) the non-root nodes'
        latents are kept only as their bottleneck values (`t_pool`); `pool.latent` then holds the
        roots alone.  With the MLP networks (`PackedNetworks`) the single-launch search keeps, per
        expansion, the one merge pre-activation row its A children's latents are formed from
        (`z_pool`) and the pool holds no latents at all (`implicit`); such an engine runs fused searches
        only.  The search result does not depend on it: the trees are bit-identical.
        merge_row_workspace (MLP networks, keep_latents=True): also allocate the merge-row buffer
        (G x (S+1) x D floats) so that the single-launch search can take its sequential-program form
        and write the node latents from it; False keeps the general kernel (same bits, slower)."""
        self.nets = nets
        self.G, self.S, self.depth = G, num_simulations, model_look_ahead
        self.discount, self.noise_frac = discount, noise_frac
        self.A, self.D = nets.A, nets.D
        self.implicit = (not keep_latents) and isinstance(nets, PackedNetworks) and self.A <= 32
        self.pool = TreePool(G, self.A, num_simulations, self.D, device=device, with_latent=not self.implicit)
        want_rows = self.implicit or (merge_row_workspace and isinstance(nets, PackedNetworks) and self.A <= 32)
        self.z_pool = (torch.empty((G, num_simulations + 1, self.D), dtype=torch.float32, device=self.pool.device)
                       if want_rows else None)
        dev = self.pool.device
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        G, A, D = self.G, self.A, self.D
        self.leaf = torch.zeros(G, **i32)
        self.child = torch.zeros(G, **i32)
        # the leaf's priors are the first G rows of a [2G, A] buffer, the greedy actions of a rollout's first step
        # the last G entries of a [2G] buffer: one f call can serve the leaf and the drawn child (_step_chain)
        self.prior2 = torch.zeros((2 * G, A), **f32)
        self.prior = self.prior2[:G]
        # the children's rewards, then (one-chain schedule) the reward of the rollout's first step: one head call
        self.reward3 = torch.zeros(G * A + G, **f32)
        self.reward = self.reward3[:G * A].view(G, A)
        self.rollout_t = torch.zeros((self.depth + 1, G), **f32)
        self.act2 = torch.zeros(2 * G, **i32)
        self.act = self.act2[G:]
        self.tmp = [torch.zeros((G, D), **f32) for _ in range(2)]
        self.stamps = None  # set to an int64[32] device tensor to collect per-stage cycle counts
        self.tree_base = (torch.arange(G, **i32) * self.pool.cap)
        self.arange_a = torch.arange(A, **i32)
        # networks evaluated by several kernels per call (residual towers) run f next to the part of g
        # that does not need f's result, on a second stream
        self.side = torch.cuda.Stream(device=dev) if getattr(nets, "concurrent", False) and dev.type == "cuda" else None
        # networks whose latents are an affine image of a few values (see muzero_amd.towers) keep those
        # values for every node next to the latent
        dt = getattr(nets, "bottleneck_g", 0)
        self.rollout = torch.zeros((G, self.depth + 1), **f32)   # [r_1 .. r_depth, v] per tree
        self.rows2 = torch.zeros(2 * G, **i32)   # rows of the leaf, then rows of the drawn child
        self.rows_buf = [self.rows2[:G], self.rows2[G:]]
        self.rows3 = torch.zeros(G * A + G, **i32)   # rows of the children-to-be, then the rows the first rollout step writes
        self.next_rows_buf = self.rows3[:G * A].view(G, A)
        self.child_slots = (self.tree_base[:, None] + self.arange_a[None, :]).contiguous()
        self.keep_latents = keep_latents or not dt or getattr(nets, "tower_f_t", None) is None
        self.use_graph = True
        self.fork_rollout = os.environ.get("MZ_FORK_ROLLOUT", "1") != "0"  # see _step_forked / _step_chain
        self.step_schedule = os.environ.get("MZ_STEP_SCHEDULE", "chain")   # "chain" or "fork"
        self.fuse_tree_kernels = os.environ.get("MZ_FUSE_TREE_KERNELS", "1") != "0"  # see _run_steps
        self._graph, self._graph_version = None, -1
        self._whole, self._whole_version = None, -1   # root creation + all steps as one graph (_replay_whole)
        self.rand_static = torch.zeros((0, G), **i32)  # the recorded search's child draws [n, G]
        self.replays = 0
        self._warm_version = None
        # bottleneck values of every node, followed by G rows for the latents of the rollout's first step
        self.t_pool = torch.zeros((G * self.pool.cap + G, dt), **f32) if dt else None
        self.tmp_t = [self.t_pool[G * self.pool.cap:], torch.zeros((G, dt), **f32)] if dt else None
        if dt:
            self.rows3[G * A:] = torch.arange(G * self.pool.cap, G * self.pool.cap + G, **i32)

    def _expand(self, node: torch.Tensor, rand_child: torch.Tensor, out_child: torch.Tensor, root: bool = False,
                root_form: bool = False) -> None:
        """expand_node (src/search/expansion.py:9-35) for one node per tree (`root`: the nodes are
        the roots, whose latents come from h and have no stored bottleneck values)."""
        pool, nets, G, A = self.pool, self.nets, self.G, self.A
        if root:  # no selection came before: the index vectors are formed here
            torch.add(self.tree_base, node, out=self.rows_buf[0])
            torch.add(self.child_slots, pool.n_nodes[:, None], out=self.next_rows_buf)
        rows, next_rows = self.rows_buf[0], self.next_rows_buf.view(-1)  # else: written by pool.select
        rows_rep = rows if getattr(nets, "strided_outputs", False) else rows.repeat_interleave(A)  # per group is enough there
        main = torch.cuda.current_stream()
        f_extra = {"t_in": (self.t_pool, rows)} if self.t_pool is not None and (not root or root_form) else {}
        if root_form:
            f_extra["root_form"] = True
        # the priors and the children both depend on the leaf only: side by side on two queues, except in the root
        # form, where both are short bottleneck-form calls that only slow each other down
        side = self.side if not root_form else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                nets.prediction(pool.latent, G, rows=rows, softmax_times=1, policy=self.prior, **f_extra)
        else:
            nets.prediction(pool.latent, G, rows=rows, softmax_times=1, policy=self.prior, **f_extra)
        extra = {}
        if self.t_pool is not None:
            extra = dict(t_in=None if (root and not root_form) else (self.t_pool, rows), t_out=self.t_pool,
                         want_latent=self.keep_latents)
            if root_form:
                extra["root_form"] = True
        nets.dynamics(pool.latent, G * A, rows=rows_rep, action=None, next_latent=pool.latent, next_rows=next_rows,
                      reward=self.reward, **extra)
        if side is not None:
            main.wait_stream(side)
        pool.expand_commit(node, self.prior, self.reward, rand_child, out_child, out_child_row=self.rows_buf[1])

    def _simulate(self, child: torch.Tensor, first_action_known: bool = False, reward_with_children: bool = False) -> torch.Tensor:
        """MuZeroSimulation (src/search/simulation.py:23-45) from one node per tree;
        returns rollout f32[G, depth+1].  first_action_known: self.act already holds the greedy action of the
        first step (f was evaluated for the drawn child together with the leaf)."""
        pool, nets, G = self.pool, self.nets, self.G
        cur, rows = pool.latent, self.rows_buf[1]  # rows of `child`, written by pool.expand_commit
        strided = getattr(nets, "strided_outputs", False)  # heads write into the columns of self.rollout directly
        out_col = (lambda d: self.rollout[:, d]) if strided else (lambda d: self.rollout_t[d])
        main = torch.cuda.current_stream()
        t_in = (self.t_pool, rows) if self.t_pool is not None else None
        for d in range(self.depth):
            nxt = self.tmp[d & 1]
            extra = dict(t_in=t_in, t_out=self.tmp_t[d & 1], want_latent=self.keep_latents) \
                if self.t_pool is not None else {}
            f_extra = {"t_in": t_in} if t_in is not None else {}
            # (with fc_merge fused into the tower's first product there is nothing to run beside f)
            fused_g = (getattr(nets, "fused_g", False) and t_in is not None and not self.keep_latents
                       and G >= getattr(nets, "fuse_min_rows", 0))
            if self.side is not None and not fused_g:  # fc_merge's latent half needs no action: it runs next to f
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    nets.prediction(cur, G, rows=rows, argmax=self.act, **f_extra)
                base = nets.merge_rows(cur, G, rows=rows, **({"t_in": t_in} if t_in is not None else {}))
                main.wait_stream(self.side)
                nets.dynamics(cur, G, rows=rows, action=self.act, next_latent=nxt, reward=out_col(d), base=base, **extra)
            else:
                if not (first_action_known and d == 0):
                    nets.prediction(cur, G, rows=rows, argmax=self.act, **f_extra)
                if reward_with_children and d == 0:
                    # the first step's reward and the A children's rewards in one head call: the step's bottleneck
                    # values were written behind the pool's (tmp_t[0]), rows3 names both sets of rows; the step's
                    # reward stays in reward3[G * A:] (the tree kernel reads it from there)
                    nets.dynamics(cur, G, rows=rows, action=self.act, next_latent=nxt, reward=None, **extra)
                    if self.depth == 1 and hasattr(nets, "value_with_rewards") and nets.value_with_rewards(
                            (self.tmp_t[0], None), G, out_col(1), self.t_pool, self.rows3, G * self.A + G, self.reward3):
                        return self.rollout if strided else self.rollout_t.t().contiguous()  # f's block, then both heads at once
                    nets.reward_on_t(self.t_pool, self.rows3, G * self.A + G, self.reward3)
                else:
                    nets.dynamics(cur, G, rows=rows, action=self.act, next_latent=nxt, reward=out_col(d), **extra)
            cur, rows = nxt, None
            if self.t_pool is not None:
                t_in = (self.tmp_t[d & 1], None)
        nets.prediction(cur, G, rows=rows, value=out_col(self.depth), **({"t_in": t_in} if t_in is not None else {}))
        return self.rollout if strided else self.rollout_t.t().contiguous()

    def begin(self, root_latent: torch.Tensor | None, root_to_play: torch.Tensor, eta: torch.Tensor,
              rand_root: torch.Tensor, root_t: torch.Tensor | None = None) -> None:
        """Root creation, root expansion (its child draw `rand_root` i32[G] is discarded, as in
        src/search/mcts.py:89) and the prior perturbation of mcts.py:91-94."""
        pool = self.pool
        root_form = self.root_form_ok(root_t)
        assert root_form or root_latent is not None, "root latents are needed unless their bottleneck values can be used"
        if root_form:
            root_latent = None  # nothing reads a root's latent in this form: no copy into the pool
        pool.init(root_to_play, root_latent.reshape(self.G, self.D).contiguous() if root_latent is not None else None)
        self.leaf.zero_()
        if root_form:
            # the roots' latents are final_conv(root_t) of the representation network: kept, like every other node,
            # as those few values, and expanded in the bottleneck form with weights composed with that convolution
            # (row tree * cap of the pool: one strided copy)
            G, cap = self.G, pool.cap
            self.t_pool[:G * cap].view(G, cap, -1)[:, 0].copy_(root_t.reshape(G, -1))
        self._expand(self.leaf, rand_root.reshape(self.G).contiguous(), self.child, root=True, root_form=root_form)
        pool.root_noise(eta.contiguous(), self.noise_frac)

    def root_form_ok(self, root_t) -> bool:
        """Whether a search can start from the roots' bottleneck values alone (see begin)."""
        return (root_t is not None and self.t_pool is not None and not self.keep_latents
                and bool(getattr(self.nets, "has_root_form", False)) and root_t.numel() == self.G * self.t_pool.shape[1])

    def _step(self, rand_column: torch.Tensor) -> torch.Tensor:
        """One `_step` (src/search/mcts.py:76-83) for every tree; returns the rollouts."""
        if self._forked():
            return self._step_chain(rand_column) if self.step_schedule == "chain" else self._step_forked(rand_column)
        self.pool.select(self.leaf, out_row=self.rows_buf[0], out_child_rows=self.next_rows_buf)
        self._expand(self.leaf, rand_column, self.child)
        rollout = self._simulate(self.child)
        self.pool.backup(self.child, self.leaf, rollout, self.discount)
        return rollout

    def _step_chain(self, rand_column: torch.Tensor) -> torch.Tensor:
        """The same step as ONE chain of launches on one stream (no hand-over between queues, each of which
        costs 5-10 us inside a replayed graph): g's tower for the A children first, then f for the leaf AND the
        drawn child in one call (2G rows: priors for the first G, greedy actions for the last G), the rollout,
        and last the children's rewards and the child records, which nothing before the backup reads.
        Same kernels on the same data as `_expand` + `_simulate`: identical results."""
        pool, nets, G, A = self.pool, self.nets, self.G, self.A
        pool.select(self.leaf, out_row=self.rows_buf[0], out_child_rows=self.next_rows_buf, rand_child=rand_column,
                    out_drawn_row=self.rows_buf[1])
        rows, next_rows = self.rows_buf[0], self.next_rows_buf.view(-1)
        nets.dynamics(pool.latent, G * A, rows=rows, action=None, next_latent=pool.latent, next_rows=next_rows, reward=None,
                      t_in=(self.t_pool, rows), t_out=self.t_pool, want_latent=False)
        nets.prediction(pool.latent, 2 * G, rows=self.rows2, softmax_times=1, policy=self.prior2, argmax=self.act2,
                        t_in=(self.t_pool, self.rows2))
        rollout = self._simulate(self.child, first_action_known=True)
        nets.reward_on_t(self.t_pool, next_rows, G * A, self.reward)
        pool.expand_commit(self.leaf, self.prior, self.reward, rand_column, self.child)
        pool.backup(self.child, self.leaf, rollout, self.discount)
        return rollout

    def _run_steps(self, rand: torch.Tensor) -> None:
        """All steps of a search (rand i32[n, G]: one row of child draws per step).  In the chain schedule the tree
        side of a step is one launch: the child records and the backup of step s and the selection of step s + 1
        (mz_tree_commit_backup_select) -- 11 launches per step instead of 13."""
        n = rand.shape[0]
        if not (self._forked() and self.step_schedule == "chain" and self.fuse_tree_kernels and self.strided_rollout()):
            for s in range(n):
                self._step(rand[s])
            return
        pool, nets, G, A = self.pool, self.nets, self.G, self.A
        rows, next_rows = self.rows_buf[0], self.next_rows_buf.view(-1)
        pool.select(self.leaf, out_row=rows, out_child_rows=self.next_rows_buf, rand_child=rand[0], out_drawn_row=self.rows_buf[1])
        for s in range(n):
            nets.dynamics(pool.latent, G * A, rows=rows, action=None, next_latent=pool.latent, next_rows=next_rows, reward=None,
                          t_in=(self.t_pool, rows), t_out=self.t_pool, want_latent=False)
            nets.prediction(pool.latent, 2 * G, rows=self.rows2, softmax_times=1, policy=self.prior2, argmax=self.act2,
                            t_in=(self.t_pool, self.rows2))
            merged = self.depth >= 1 and self.depth + 1 <= 8
            rollout = self._simulate(self.child, first_action_known=True, reward_with_children=merged)
            if not merged:
                nets.reward_on_t(self.t_pool, next_rows, G * A, self.reward)
            last = s + 1 == n
            pool.commit_backup_select(self.leaf, self.prior, self.reward, rand[s], self.child, rollout, self.discount,
                                      select_next=not last, out_row=rows, out_child_rows=self.next_rows_buf,
                                      rand_next=rand[s if last else s + 1], out_drawn_row=self.rows_buf[1],
                                      reward0=self.reward3[G * A:] if merged else None)

    def strided_rollout(self) -> bool:
        return bool(getattr(self.nets, "strided_outputs", False))

    def _forked(self) -> bool:
        """Whether a step can run the rollout beside the commit of the expansion: the rollout reads only the
        children's bottleneck values (no tree record), the row of the drawn child is known at selection time,
        and the networks can evaluate the reward head on its own (`reward_on_t`)."""
        nets = self.nets
        return (self.fork_rollout and self.side is not None and self.t_pool is not None and not self.keep_latents
                and getattr(nets, "fused_g", False) and hasattr(nets, "reward_on_t")
                and not getattr(nets, "track_running_stats", False)  # f would advance its buffers on both streams at once
                and self.G >= getattr(nets, "fuse_min_rows", 0))

    def _step_forked(self, rand_column: torch.Tensor) -> torch.Tensor:
        """The same step with a shorter chain of dependent launches: as soon as the children's bottleneck values
        exist (g's tower for the A children), the main stream starts the rollout from the drawn child, while the
        side stream evaluates the leaf's priors, the children's rewards and writes the child records
        (src/search/expansion.py:25-35).  Both meet before the backup.  Same kernels on the same data as
        `_expand` + `_simulate`: identical results."""
        pool, nets, G, A = self.pool, self.nets, self.G, self.A
        pool.select(self.leaf, out_row=self.rows_buf[0], out_child_rows=self.next_rows_buf, rand_child=rand_column,
                    out_drawn_row=self.rows_buf[1])
        rows, next_rows = self.rows_buf[0], self.next_rows_buf.view(-1)
        main, side = torch.cuda.current_stream(), self.side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            nets.prediction(pool.latent, G, rows=rows, softmax_times=1, policy=self.prior, t_in=(self.t_pool, rows))
        nets.dynamics(pool.latent, G * A, rows=rows, action=None, next_latent=pool.latent, next_rows=next_rows, reward=None,
                      t_in=(self.t_pool, rows), t_out=self.t_pool, want_latent=False)
        side.wait_stream(main)  # the children's bottleneck values are in the pool
        with torch.cuda.stream(side):
            nets.reward_on_t(self.t_pool, next_rows, G * A, self.reward)
            pool.expand_commit(self.leaf, self.prior, self.reward, rand_column, self.child)
        rollout = self._simulate(self.child)  # reads rows_buf[1] (written by select) and bottleneck values only
        main.wait_stream(side)
        pool.backup(self.child, self.leaf, rollout, self.discount)
        return rollout

    def advance(self, rand_child: torch.Tensor, trace: dict | None = None) -> None:
        """One `_step` per column of rand_child i32[G, n].

        With networks evaluated by many small kernels (`nets.concurrent`) a step is some 30 launches
        and, at a few hundred trees, bound by the launch rate.  The first search after construction
        (or after the weights changed) runs launch by launch; the next one records all n steps, both
        streams included, as one HIP graph, and from then on a search copies its child draws into the
        graph's buffer and replays it: one launch per search.  Tracing runs the launches one by one."""
        rand_t = rand_child.t().contiguous()
        n = rand_t.shape[0]
        use_graph = self.use_graph and trace is None and self.side is not None and n > 2
        version = getattr(self.nets, "version", 0)
        if use_graph and self._warm_version == version:
            if self._graph is None or self._graph_version != version or self.rand_static.shape[0] != n:
                # recording launches nothing; every scratch buffer exists since the earlier search
                self._graph = None
                self.rand_static = torch.zeros_like(rand_t)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._run_steps(self.rand_static)
                self._graph, self._graph_version = graph, version
            self.rand_static.copy_(rand_t)
            self._graph.replay()
            self.replays += 1
            return
        for s in range(n):
            rollout = self._step(rand_t[s])
            if trace is not None:
                trace.setdefault("leaf", []).append(self.leaf.clone())
                trace.setdefault("child", []).append(self.child.clone())
                trace.setdefault("rollout", []).append(rollout.clone())
                trace.setdefault("prior", []).append(self.prior.clone())
        self._warm_version = version

    def finish(self) -> SearchResult:
        return SearchResult(*self.pool.result())

    @staticmethod
    def check(result: SearchResult) -> None:
        """Raises if a search reported a failure: the single-launch kernel poisons the root value of a
        tree (NaN) when a team of waves gave up waiting for its partner, and a lock-step search leaves
        simulations undone when a descent or an expansion failed.  Waits for the GPU (diagnostic)."""
        if bool(torch.isnan(result.value).any()):
            raise _hip.HipError("the search kernel reported a stalled wave team (NaN root value)")

    def run(self, root_latent: torch.Tensor | None, root_to_play: torch.Tensor, eta: torch.Tensor,
            rand_child: torch.Tensor, trace: dict | None = None, fused: bool | None = None,
            root_t: torch.Tensor | None = None) -> SearchResult:
        """root_latent f32[G, D]; root_to_play i8[G]; eta f32[G, A]; rand_child i32[G, S+1].
        root_t f32[G, bottleneck] (optional): the few values the roots' latents are the representation network's
        final 1x1 convolution of; used instead of root_latent (which may then be None) where `root_form_ok`.

        `fused` (default: whenever A <= 32) runs the whole search as ONE kernel launch
        (mz_search_run); otherwise the per-stage kernels are driven in lock step.  Both leave
        bit-identical pools."""
        assert rand_child.shape == (self.G, self.S + 1) and rand_child.dtype == torch.int32
        fusable = getattr(self.nets, "fusable", True)
        if fused is None:
            fused = self.A <= 32 and fusable
        if fused and not fusable:
            raise NotImplementedError("the single-launch search covers the MLP-only networks; "
                                      "these networks run in lock step (fused=False)")
        if self.implicit and not fused:
            raise NotImplementedError("an engine built with keep_latents=False stores no node latents: the lock-step "
                                      "kernels need them (fused=True, or keep_latents=True)")
        if fused:
            return self.run_fused(root_latent, root_to_play, eta, rand_child, trace)
        if trace is None and self._replay_whole(root_to_play, eta, rand_child, root_t):
            return self.finish()
        self.begin(root_latent, root_to_play, eta, rand_child[:, 0], root_t=root_t)
        self.advance(rand_child[:, 1:], trace)
        return self.finish()

    def _replay_whole(self, root_to_play, eta, rand_child, root_t) -> bool:
        """A search that starts from the roots' bottleneck values has only small inputs (a few hundred bytes per
        tree): root creation and root expansion are then recorded in front of the steps, the inputs are copied into
        the graph's buffers, and the whole search is one replay -- at a few hundred trees the dozen launches of
        `begin`, issued one by one, took 0.3 ms of a 6.3 ms search.  False (nothing done) until a launch-by-launch
        search has run with the current weights, or when the search does not qualify."""
        version = getattr(self.nets, "version", 0)
        if not (self.use_graph and self.side is not None and self.S > 2 and self._warm_version == version
                and self.root_form_ok(root_t)):
            return False
        G, A, S = self.G, self.A, self.S
        if self._whole is None or self._whole_version != version:
            dev = self.pool.device
            self._whole = None
            self.in_t = torch.zeros((G, self.t_pool.shape[1]), dtype=torch.float32, device=dev)
            self.in_to_play = torch.zeros(G, dtype=torch.int8, device=dev)
            self.in_eta = torch.zeros((G, A), dtype=torch.float32, device=dev)
            self.in_rand = torch.zeros((S + 1, G), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.begin(None, self.in_to_play, self.in_eta, self.in_rand[0], root_t=self.in_t)
                self._run_steps(self.in_rand[1:])
            self._whole, self._whole_version = graph, version
        self.in_t.copy_(root_t.reshape(G, -1))
        self.in_to_play.copy_(root_to_play.reshape(G))
        self.in_eta.copy_(eta.reshape(G, A))
        self.in_rand.copy_(rand_child.t())
        self._whole.replay()
        self.replays += 1
        return True

    def run_fused(self, root_latent, root_to_play, eta, rand_child, trace: dict | None = None) -> SearchResult:
        pool, G, A, S, D = self.pool, self.G, self.A, self.S, self.D
        dev = pool.device
        root_latent = root_latent.reshape(G, D).contiguous()
        rand_child = rand_child.contiguous()
        eta = eta.contiguous()
        assert root_latent.dtype == torch.float32 and eta.dtype == torch.float32 and root_to_play.dtype == torch.int8
        assert eta.numel() == G * A and root_to_play.numel() == G
        t_leaf = t_child = t_roll = t_prior = None
        if trace is not None:
            t_leaf = torch.zeros((G, S), dtype=torch.int32, device=dev)
            t_child = torch.zeros((G, S), dtype=torch.int32, device=dev)
            t_roll = torch.zeros((G, S, self.depth + 1), dtype=torch.float32, device=dev)
            t_prior = torch.zeros((G, S + 1, A), dtype=torch.float32, device=dev)
        prm = _hip.SearchParams(S, self.depth, pool.tab_len, 0, float(self.discount), float(self.noise_frac),
                                ptr(pool.sqrt_tab), ptr(pool.log_tab), ptr(root_latent), ptr(root_to_play), ptr(eta),
                                ptr(rand_child), ptr(t_leaf), ptr(t_child), ptr(t_roll), ptr(t_prior),
                                ptr(self.stamps), ptr(self.z_pool))
        policy = torch.empty((G, A), dtype=torch.float64, device=dev)
        policy32 = torch.empty((G, A), dtype=torch.float32, device=dev)
        value = torch.empty(G, dtype=torch.float64, device=dev)
        visits = torch.empty((G, A), dtype=torch.int32, device=dev)
        _hip.check(_hip.lib().mz_search_run(C.byref(pool.c), C.byref(self.nets.plan), ptr(self.nets.blob), C.byref(prm),
                                            ptr(policy), ptr(policy32), ptr(value), ptr(visits), _stream()),
                   "mz_search_run")
        if trace is not None:
            trace["leaf"] = [t_leaf[:, s].contiguous() for s in range(S)]
            trace["child"] = [t_child[:, s].contiguous() for s in range(S)]
            trace["rollout"] = [t_roll[:, s].contiguous() for s in range(S)]
            trace["prior"] = [t_prior[:, s + 1].contiguous() for s in range(S)]
            trace["root_prior"] = t_prior[:, 0].contiguous()
        return SearchResult(policy, policy32, value, visits)


def run_recorded(pool: TreePool, root_to_play, prior_stream, reward_stream, rand_stream, sim_stream, eta,
                 noise_frac: float, discount: float, fused_steps: bool = False):
    """Tree kernels only, driven by recorded network outputs (the parity harness of the
    bit-exact contract).  Streams are device tensors shaped [S+1, G, A], [S+1, G, A],
    [S+1, G], [S, G, depth+1], eta [G, A].  Returns (leaf[S, G], child[S, G]).
    fused_steps: one launch per step (mz_tree_commit_backup_select) instead of select / expand_commit / backup;
    also returns the rows it reported, (row[S, G], child_rows[S, G, A], drawn_row[S, G])."""
    S = sim_stream.shape[0]
    G = pool.G
    dev = pool.device
    leaf = torch.zeros(G, dtype=torch.int32, device=dev)
    child = torch.zeros(G, dtype=torch.int32, device=dev)
    leaves = torch.zeros((S, G), dtype=torch.int32, device=dev)
    children = torch.zeros((S, G), dtype=torch.int32, device=dev)
    pool.init(root_to_play)
    pool.expand_commit(leaf, prior_stream[0], reward_stream[0], rand_stream[0], child)
    pool.root_noise(eta, noise_frac)
    if fused_steps:
        A = pool.A
        row = torch.zeros(G, dtype=torch.int32, device=dev)
        crows = torch.zeros((G, A), dtype=torch.int32, device=dev)
        drawn = torch.zeros(G, dtype=torch.int32, device=dev)
        rows_s = torch.zeros((S, G), dtype=torch.int32, device=dev)
        crows_s = torch.zeros((S, G, A), dtype=torch.int32, device=dev)
        drawn_s = torch.zeros((S, G), dtype=torch.int32, device=dev)
        if S > 0:
            pool.select(leaf, out_row=row, out_child_rows=crows, rand_child=rand_stream[1].contiguous(), out_drawn_row=drawn)
        for s in range(S):
            leaves[s].copy_(leaf)
            rows_s[s].copy_(row)
            crows_s[s].copy_(crows)
            drawn_s[s].copy_(drawn)
            last = s + 1 == S
            pool.commit_backup_select(leaf, prior_stream[s + 1], reward_stream[s + 1], rand_stream[s + 1].contiguous(), child,
                                      sim_stream[s], discount, select_next=not last, out_row=row, out_child_rows=crows,
                                      rand_next=rand_stream[s + 1 if last else s + 2].contiguous(), out_drawn_row=drawn)
            children[s].copy_(child)
        return leaves, children, (rows_s, crows_s, drawn_s)
    for s in range(S):
        pool.select(leaf)
        pool.expand_commit(leaf, prior_stream[s + 1], reward_stream[s + 1], rand_stream[s + 1], child)
        pool.backup(child, leaf, sim_stream[s], discount)
        leaves[s].copy_(leaf)
        children[s].copy_(child)
    return leaves, children
