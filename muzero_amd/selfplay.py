"""Batched self-play: G Connect-Four games advance in lock step entirely on one GPU.

Per move (the pipeline of src/training_data_generator.py:107-142, for all games at once):
observe (k_c4_observe) -> history ring + tensor (k_history) -> h (k_pointwise_conv, or the torch
module on the GPU for convolutional configs) -> MCTS.run (muzero_amd.engine.BatchedMCTS) ->
epsilon-greedy pick (k_pick_action) -> env step (k_c4_step) -> append to the trajectory store.
Finished games restart in place, so the batch never drains.

Randomness: the reference plays games one after another from three global generators, an
interleaving a batched engine cannot reproduce; here every stream is drawn per move from one
seeded numpy Generator (child choices, exploration coin and action) and torch's Dirichlet.
For the stream-exact single-game path use muzero_amd.training_data_generator.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from muzero_amd import _hip
from muzero_amd._hip import ptr
from muzero_amd.engine import BatchedMCTS, _stream, make_networks
from muzero_amd.environments.connect_four import BatchedConnectFour
from muzero_amd.neural_networks.per_call import per_call_statistics  # noqa: F401  (re-exported: batched torch evaluation with per-game statistics)


class DeviceHistory:
    """K-slot observation ring per game + history tensor assembly (src/ring_buffer.py)."""

    def __init__(self, G: int, K: int, obs_shape: tuple[int, int, int], device="cuda"):
        self.G, self.K = G, K
        self.C, self.H, self.W = obs_shape
        dev = torch.device(device)
        self.ring = torch.zeros((G, K, self.C * self.H * self.W), dtype=torch.float32, device=dev)
        self.act_ring = torch.zeros((G, K), dtype=torch.int32, device=dev)
        self.out = torch.zeros((G, self.C * K + K, self.H, self.W), dtype=torch.float32, device=dev)

    def push(self, frame: torch.Tensor, step: torch.Tensor, action: torch.Tensor | None = None) -> torch.Tensor:
        assert frame.dtype == torch.float32 and frame.is_contiguous() and frame.numel() == self.G * self.C * self.H * self.W
        assert step.dtype == torch.int32 and step.numel() == self.G
        _hip.check(_hip.lib().mz_history_push_assemble(ptr(self.ring), ptr(self.act_ring), ptr(frame), ptr(action),
                                                       ptr(step), self.G, self.K, self.C, self.H, self.W, ptr(self.out),
                                                       _stream()), "mz_history_push_assemble")
        return self.out


class RepresentationRunner:
    """h on the GPU.  MLP-style configs (`downsample: [], res_net: []`, latent H, W equal to the
    observation's) are one 1x1 convolution and run the point-wise HIP kernel; every other
    configuration runs layer by layer on the convolution kernels (muzero_amd.conv_repr:
    gathered fp32 matrix products, BatchNorm with the per-game batch statistics of the reference's
    one-game-at-a-time calls).

    The caller's module is never moved or modified: the runner keeps its own device copy of the
    weights and re-reads them whenever a parameter of the module has changed (in-place updates
    bump a tensor's version counter; a replaced tensor has a new address), so a generator that is
    reused after training searches with the trained weights.  `track_running_stats=True`
    additionally advances the BatchNorm running statistics as the reference's calls would;
    `sync_running_stats()` writes them back into the caller's module."""

    def __init__(self, repr_net: torch.nn.Module, device="cuda", track_running_stats: bool = False):
        self.src = repr_net
        self.device = torch.device(device)
        self.pointwise = (len(repr_net.downsample_layers) == 0 and len(repr_net.res_blocks) == 0
                          and isinstance(repr_net.final_conv, torch.nn.Conv2d))
        self.latent_hw = (repr_net.latent_height, repr_net.latent_width)
        self.track_running_stats = track_running_stats
        self._seen = None
        self.conv = None
        self.refresh()

    def _versions(self):
        return tuple((t._version, t.data_ptr()) for t in self.src.parameters())

    def refresh(self, force: bool = False) -> None:
        """Re-reads the module's weights if they changed since the last call (always with `force`)."""
        seen = self._versions()
        if not force and seen == self._seen:
            return
        if self.pointwise:
            conv = self.src.final_conv
            self.weight = conv.weight.detach().to(self.device, torch.float32).reshape(conv.out_channels, -1).contiguous().clone()
            self.bias = conv.bias.detach().to(self.device, torch.float32).contiguous().clone() if conv.bias is not None else None
        if self.conv is not None:
            self.conv.sync_running_stats()
            self.conv.refresh()
        elif not self.pointwise:
            from muzero_amd.conv_repr import ConvRepresentation
            self.conv = ConvRepresentation(self.src, device=self.device, track_running_stats=self.track_running_stats)
        self._seen = seen

    def sync_running_stats(self) -> None:
        if self.conv is not None:
            self.conv.sync_running_stats()

    def root_final(self):
        """The network's final 1x1 convolution when root latents are its image of a pooled tensor (then
        `__call__(..., want_root_t=True)` leaves that tensor in `self.root_t`), else None."""
        if self.pointwise or self.conv is None or not self.conv.final_is_pointwise():
            return None
        return self.src.final_conv

    def __call__(self, history: torch.Tensor, want_root_t: bool = False) -> torch.Tensor:
        self.refresh()
        self.root_t = None
        G, cin, H, W = history.shape
        if self.pointwise and (H, W) == self.latent_hw:
            x = history.to(self.device, torch.float32).contiguous()
            cout = self.weight.shape[0]
            out = torch.empty((G, cout, H, W), dtype=torch.float32, device=self.device)
            _hip.check(_hip.lib().mz_repr_pointwise(ptr(x), ptr(self.weight), ptr(self.bias), G, cin, cout, H * W,
                                                    ptr(out), _stream()), "mz_repr_pointwise")
            return out
        if self.conv is None:  # a point-wise module asked for another latent size: 1x1 convolution + adaptive pool
            from muzero_amd.conv_repr import ConvRepresentation
            self.conv = ConvRepresentation(self.src, device=self.device, track_running_stats=self.track_running_stats)
        out = self.conv(history, want_root_t=want_root_t)
        self.root_t = self.conv.root_t
        return out


def pick_action(policy32: torch.Tensor, explore: torch.Tensor | None, random_action: torch.Tensor | None,
                out: torch.Tensor) -> torch.Tensor:
    G, A = policy32.shape
    _hip.check(_hip.lib().mz_pick_action(ptr(policy32), ptr(explore), ptr(random_action), G, A, ptr(out), _stream()),
               "mz_pick_action")
    return out


class MoveStreams:
    """The random inputs of one lock-step move of G games: child draws i32[G, S+1], Dirichlet noise f32[G, A],
    exploration coin i8[G] and the uniform action i32[G] (src/training_data_generator.py:107-142 draws them
    from three global generators, game after game).  One numpy Generator on the host (reproducible against
    a CPU replay), or one kernel on the device (`device_rng`: `mz_move_draws`, a counter-based generator keyed by
    (seed, move number, game) -- the whole move stays on the GPU and costs one launch for all four streams)."""

    def __init__(self, G: int, A: int, S: int, config, device, seed: int = 0, device_rng: bool = False):
        self.G, self.A, self.S, self.cfg, self.device = G, A, S, config, torch.device(device)
        self.alpha = config.mcts.dirichlet_alpha
        self.rng = np.random.default_rng(seed)
        self.device_rng = device_rng
        if device_rng:
            dev = self.device
            self.seed = int(seed) & (2 ** 64 - 1)
            self._state = torch.zeros(2, dtype=torch.int64, device=dev)   # move number, work counter (advanced by the kernel)
            p = config.random_action_probability
            self._p_random = None if p is None else torch.as_tensor(np.asarray(p, np.float32)).to(dev)
            self._eps = torch.zeros(1, dtype=torch.float32, device=dev)
            self._rand = torch.empty((G, S + 1), dtype=torch.int32, device=dev)
            self._eta = torch.empty((G, A), dtype=torch.float32, device=dev)
            self._explore = torch.empty(G, dtype=torch.int8, device=dev)
            self._rnd = torch.empty(G, dtype=torch.int32, device=dev)

    def draw(self, eps: float):
        G, A, S, dev = self.G, self.A, self.S, self.device
        if self.device_rng:
            # (the four outputs are this object's own buffers, overwritten by the next draw)
            if torch.is_tensor(eps):
                eps_t = eps if eps.dtype == torch.float32 and eps.numel() == 1 else eps.to(torch.float32).reshape(1)
            else:
                self._eps.fill_(float(eps))
                eps_t = self._eps
            rand, eta, explore, rnd = self._rand, self._eta, self._explore, self._rnd
            _hip.check(_hip.lib().mz_move_draws(self.seed, ptr(self._state), ptr(eps_t), ptr(self._p_random), float(self.alpha), G, A,
                                                S + 1, ptr(rand), ptr(eta), ptr(explore), ptr(rnd),
                                                torch.cuda.current_stream().cuda_stream), "mz_move_draws")
        else:
            rand = torch.from_numpy(self.rng.integers(0, A, size=(G, S + 1), dtype=np.int32)).to(dev)
            eta = torch.from_numpy(self.rng.dirichlet(np.full(A, self.alpha), size=G).astype(np.float32)).to(dev)
            explore = torch.from_numpy((self.rng.random(G) < eps).astype(np.int8)).to(dev)
            p = self.cfg.random_action_probability
            rnd = torch.from_numpy(self.rng.choice(A, size=G, p=p).astype(np.int32)).to(dev)
        return rand, eta, explore, rnd


def episodes_from_records(rec, G: int):
    """Finished episodes of `rec`, a list of per-move tuples (obs, policy32, value, action, reward, ended) over
    G games, as lists of Chunk (states on the CPU) in the reference's Episode/Chunk format
    (src/training_data_generator.py:20-40), game by game."""
    from muzero_amd.training_data_generator import Chunk, Episode
    obs, pol, val, act, rew, end = (torch.stack([r[i] for r in rec]).cpu() for i in range(6))
    episodes = []
    for g in range(G):
        cur = []
        for t in range(len(rec)):
            cur.append(Chunk(state=obs[t, g].unsqueeze(0), policy=pol[t, g], reward=float(rew[t, g]),
                             value=float(val[t, g]), best_action=int(act[t, g])))
            if end[t, g]:
                episodes.append(Episode(chunks=cur))
                cur = []
    return episodes


class SelfPlayEngine:
    def __init__(self, repr_net, dyn_net, pred_net, G: int, config, device="cuda", seed: int = 0,
                 device_rng: bool = False, env=None, track_running_stats: bool = False, root_form: bool = True,
                 share: "SelfPlayEngine | None" = None, graph_moves: bool = False):
        """`config`: TrainingDataGeneratorConfig (mcts settings, epsilon schedule, max steps).
        `env`: a batched environment (`observe`, `step`, `reset(mask)`, `num_actions`,
        `observation_space`); default: G Connect-Four games on the GPU.  Host-side simulators go
        through `muzero_amd.environments.host_pool.HostEnvPool`.

        The search consumes no random numbers itself: the Dirichlet noise, the child draws and the
        exploration decisions of a move are inputs.  By default they come from a numpy generator on
        the host (reproducible against a CPU replay); `device_rng=True` draws them on the GPU with
        torch's generator instead, which keeps the whole move on the device.

        `track_running_stats`: the BatchNorm running statistics of h, g and f advance with every network
        call as the reference's train-mode calls make them (no output depends on them);
        `sync_running_stats()` writes them back into the three modules (`generate` does so at its end).

        `root_form`: with a representation network that ends in a 1x1 convolution of a few channels, a
        residual-tower search starts from those channels' pooled values: h's final convolution is composed
        into the first layers of g and f, and the root expansion runs on the split (two fp16 planes) merge kernel.  Its
        network outputs agree with the reference-ordered float32 evaluation to 1e-5 of their scale, a bound
        and not bit parity: a selection or a greedy action that two candidates tie for to that precision
        may fall the other way.  False keeps the roots as full latents through the plain kernels.

        `graph_moves` (device environment, device_rng, single-launch search only): after two ordinary moves the whole
        move -- random draws, observation, history, h, search, action pick, environment step, restarts, some twenty-five
        launches of which all but the search take a few microseconds -- is recorded as one HIP graph and replayed;
        `move` then returns the graph's own output tensors, valid until the next move (`generate` copies them).

        `share`: another engine over the same three modules whose device copies of the weights (h runner,
        packed g / f) this one uses instead of packing its own; the two must issue their device work on the
        same stream."""
        self.G, self.cfg = G, config
        self.device = torch.device(device)
        m = config.mcts
        if m.max_iterations <= 0:
            raise ValueError("batched self-play needs max_iterations > 0 (the time-bounded mode is single-game only)")
        self.env = env if env is not None else BatchedConnectFour(G, device=device)
        assert getattr(self.env, "G", G) == G, "the environment batch must hold G games"
        self.A = self.env.num_actions
        if share is not None:
            self.h, self._root_final, self.nets = share.h, share._root_final, share.nets
        else:
            self.h = RepresentationRunner(repr_net, device=device, track_running_stats=track_running_stats)
            self._root_final = self.h.root_final() if root_form else None
            self.nets = make_networks(dyn_net, pred_net, device=device, track_running_stats=track_running_stats,
                                      root_final=self._root_final)
        self.mcts = BatchedMCTS(self.nets, G, m.max_iterations, m.model_look_ahead, m.discount_factor, m.noise_frac,
                                device=device, keep_latents=False)  # nothing here reads the children's latents
        self.history = DeviceHistory(G, repr_net.history_length, self.env.observation_space, device=device)
        self.streams = MoveStreams(G, self.A, m.max_iterations, config, device, seed, device_rng)
        self.device_rng = device_rng
        dev = self.device
        self.step_in_episode = torch.zeros(G, dtype=torch.int32, device=dev)
        self.action = torch.zeros(G, dtype=torch.int32, device=dev)
        self.env.reset()
        self.graph_moves = bool(graph_moves)
        if self.graph_moves and not (device_rng and env is None and getattr(self.nets, "fusable", True)):
            raise ValueError("graph_moves needs the device environment, device_rng=True and networks the single-launch search covers")
        self._move_graph = self._move_out = self._move_seen = None
        self._moves_played = 0
        self._eps = torch.zeros((), dtype=torch.float32, device=dev)

    @property
    def rng(self):
        return self.streams.rng

    def sync_running_stats(self) -> None:
        """With track_running_stats: the advanced BatchNorm buffers of h, g and f go back into the modules."""
        self.h.sync_running_stats()
        if hasattr(self.nets, "sync_running_stats"):
            self.nets.sync_running_stats()

    def refresh(self) -> None:
        """Re-reads the three modules' weights (call once after a weight update; `move` also notices a changed
        representation network or final convolution by itself)."""
        self.h.refresh()
        self.nets.refresh()

    def _root_versions(self):
        if self._root_final is None:
            return None
        return tuple((t._version, t.data_ptr()) for t in self._root_final.parameters())

    def epsilon(self, training_steps: int) -> float:
        c = self.cfg
        return max(c.epsilon * (c.epsilon_decay ** training_steps), c.min_epsilon)

    def launch(self, draws, pick: bool = True):
        """The device side of one move, up to the action pick (left in `self.action`; `pick=False`: left to the
        caller): nothing here waits for the GPU.  `draws`: this engine's games' (rand, eta, explore, rnd) of
        MoveStreams.draw.  Returns (obs, SearchResult)."""
        G = self.G
        rand, eta, explore, rnd = draws
        obs, to_play = self.env.observe()
        hist = self.history.push(obs, self.step_in_episode, None)  # self-play stores action id 0 with every frame
        root_form = self._root_final is not None and bool(getattr(self.nets, "has_root_form", False))
        latent = self.h(hist, want_root_t=root_form).reshape(G, -1)
        root_t = self.h.root_t.reshape(G, -1) if root_form and self.h.root_t is not None else None
        if root_t is not None and not self.mcts.root_form_ok(root_t):
            root_t = None
        if root_t is not None:
            if self._root_versions() != getattr(self.nets, "root_seen", None):
                self.nets.refresh()  # the composed root weights hold a copy of the final convolution's (it records what it read)
        res = self.mcts.run(latent if root_t is None else None, to_play, eta, rand, root_t=root_t)
        if pick:
            pick_action(res.policy32, explore, rnd, self.action)
        return obs, res

    def move(self, training_steps: int = 0):
        """One move of every game.  Returns device tensors (obs, policy32, value, action, reward,
        done) for this move; games that ended (or hit max_steps_per_episode) restart."""
        if self.graph_moves:
            return self._move_replayed(self.epsilon(training_steps))
        return self._move(self.epsilon(training_steps))

    def _move_replayed(self, eps: float):
        self.h.refresh()
        seen = (getattr(self.nets, "version", 0), self.h._seen)
        if self._move_graph is not None and seen == self._move_seen:
            self._eps.fill_(eps)
            self._move_graph.replay()
            return self._move_out
        self._move_graph = None
        self._eps.fill_(eps)
        out = self._move(self._eps)  # an ordinary move (lazy set-up of every component happens in the first ones)
        self._moves_played += 1
        if self._moves_played >= 2:
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):  # (the draws' move counter lives on the device: every replay draws fresh numbers)
                self._move_out = self._move(self._eps)  # recorded, not executed: the games stay where they are
            self._move_graph, self._move_seen = graph, seen
        return out

    def _move(self, eps):
        if hasattr(self.env, "move_end") and self.step_in_episode.dtype == torch.int32:
            # the device environment ends the move in one launch: pick, step, episode bookkeeping, restarts
            draws = self.streams.draw(eps)
            obs, res = self.launch(draws, pick=False)
            played, reward, ended = self.env.move_end(res.policy32, draws[2], draws[3], self.step_in_episode,
                                                      int(self.cfg.max_steps_per_episode), self.action)
            return obs, res.policy32, res.value, played, reward, ended
        obs, res = self.launch(self.streams.draw(eps))
        reward, done = self.env.step(self.action)
        if done.is_cuda and done.dtype == torch.int8 and self.step_in_episode.dtype == torch.int32:
            ended = torch.empty(self.G, dtype=torch.int8, device=done.device)
            _hip.check(_hip.lib().mz_episode_end(ptr(done), ptr(self.step_in_episode), int(self.cfg.max_steps_per_episode), self.G,
                                                 ptr(ended), _stream()), "mz_episode_end")
            out = (obs, res.policy32, res.value, self.action.clone(), reward, ended)
            self.env.reset(ended)
            return out
        self.step_in_episode += 1
        ended = (done.bool() | (self.step_in_episode >= self.cfg.max_steps_per_episode)).to(torch.int8)
        out = (obs, res.policy32, res.value, self.action.clone(), reward, ended)
        self.env.reset(ended)
        self.step_in_episode.masked_fill_(ended.bool(), 0)
        return out

    def generate(self, num_moves: int, training_steps: int = 0):
        """Plays `num_moves` lock-step moves and returns the finished episodes as lists of Chunk
        (states on the CPU), in the reference's Episode/Chunk format."""
        rec = []
        for _ in range(num_moves):
            out = self.move(training_steps)
            rec.append(tuple(t.clone() for t in out) if self.graph_moves else out)
        if getattr(self.h, "track_running_stats", False):
            self.sync_running_stats()  # training advances the same buffers next (src/training.py runs the modules in train mode)
        return episodes_from_records(rec, self.G)


class PipelinedSelfPlay:
    """Self-play with host-side simulators in two half-batches that alternate between the host and the
    device: while the thread pool steps the simulators of one half (src/training_data_generator.py:131-137
    for each of its games), the GPU runs observation upload, history, h, the search and the action pick of
    the other half, launched beforehand on the one stream both halves use.  The host never waits for device
    work other than the half it is about to step (one event per half), so a move costs about
    max(host time, device time) instead of their sum.

    The two halves share one device copy of the networks; each has its own tree pool, launch graph, history
    ring and environment pool.  The random inputs of a move are drawn for all G games at once, exactly as
    SelfPlayEngine draws them, so that with the same seed both engines play the same games wherever the
    search's results do not depend on the batch size (they do not for the kernels used here: every tree
    and every game's h call is computed independently of its neighbours).

    `envs`: G single-game simulators with the reference's Environment methods (src/environment.py:6-94).
    After every `move` the device is already working on the next one: one look-ahead search per half is
    discarded when playing stops, and the exploration rate passed to `move` applies to the move launched
    inside that call."""

    def __init__(self, repr_net, dyn_net, pred_net, envs, config, device="cuda", seed: int = 0,
                 device_rng: bool = False, workers: int = 0, track_running_stats: bool = False, root_form: bool = True):
        from muzero_amd.environments.host_pool import HostEnvPool
        envs = list(envs)
        self.G, self.cfg, self.device = len(envs), config, torch.device(device)
        assert self.G >= 2, "two half-batches need at least two games"
        half = (self.G + 1) // 2
        self.bounds = [(0, half), (half, self.G)]
        lead = SelfPlayEngine(repr_net, dyn_net, pred_net, half, config, device=device, seed=seed, device_rng=device_rng,
                              env=HostEnvPool(envs[:half], device=device, workers=workers),
                              track_running_stats=track_running_stats, root_form=root_form)
        rest = SelfPlayEngine(repr_net, dyn_net, pred_net, self.G - half, config, device=device, seed=seed,
                              device_rng=device_rng, env=HostEnvPool(envs[half:], device=device, workers=workers),
                              share=lead)
        self.halves = [lead, rest]
        self.A = lead.A
        self.streams = MoveStreams(self.G, self.A, config.mcts.max_iterations, config, device, seed, device_rng)
        pin = self.device.type == "cuda"
        self._action_host = [torch.zeros(e.G, dtype=torch.int32, pin_memory=pin) for e in self.halves]
        self._steps_host = [torch.zeros(e.G, dtype=torch.int32, pin_memory=pin) for e in self.halves]
        self._event = [torch.cuda.Event() if pin else None for _ in self.halves]
        self._pending = [None, None]
        self._draws = None

    @property
    def nets(self):
        return self.halves[0].nets

    def epsilon(self, training_steps: int) -> float:
        return self.halves[0].epsilon(training_steps)

    def refresh(self) -> None:
        self.halves[0].refresh()

    def sync_running_stats(self) -> None:
        self.halves[0].sync_running_stats()

    def _launch(self, i: int) -> None:
        e = self.halves[i]
        lo, hi = self.bounds[i]
        obs, res = e.launch(tuple(t[lo:hi] for t in self._draws))
        self._action_host[i].copy_(e.action, non_blocking=True)
        if self._event[i] is not None:
            self._event[i].record()
        self._pending[i] = (obs, res.policy32, res.value, e.action.clone())

    def _complete(self, i: int):
        """Waits for half i's device work, steps its simulators on the host, restarts finished games."""
        e = self.halves[i]
        if self._event[i] is not None:
            self._event[i].synchronize()
        obs, policy32, value, action = self._pending[i]
        reward, done = e.env.step_host(self._action_host[i].tolist())
        steps = self._steps_host[i]
        steps += 1
        ended = (done.bool() | (steps >= self.cfg.max_steps_per_episode)).to(torch.int8)
        e.env.reset_host(torch.nonzero(ended).reshape(-1).tolist())
        steps.masked_fill_(ended.bool(), 0)
        e.step_in_episode.copy_(steps, non_blocking=True)  # queued behind the other half's search; read by this half's next push
        return obs, policy32, value, action, reward, ended

    def move(self, training_steps: int = 0):
        """One move of every game: (obs, policy32, value, action) on the device, (reward, ended) on the host."""
        eps = self.epsilon(training_steps)
        if self._draws is None:
            self._draws = self.streams.draw(eps)
            for i in range(2):
                self._launch(i)
        outs = []
        for i in range(2):
            outs.append(self._complete(i))   # the other half's launched work keeps the GPU busy meanwhile
            if i == 0:
                self._draws = self.streams.draw(eps)
            self._launch(i)
        return tuple(torch.cat([o[k] for o in outs]) for k in range(6))

    def generate(self, num_moves: int, training_steps: int = 0):
        rec = [self.move(training_steps) for _ in range(num_moves)]
        if getattr(self.halves[0].h, "track_running_stats", False):
            self.sync_running_stats()
        return episodes_from_records(rec, self.G)

    def close(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for e in self.halves:
            e.env.close()
