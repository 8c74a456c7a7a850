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


class SelfPlayEngine:
    def __init__(self, repr_net, dyn_net, pred_net, G: int, config, device="cuda", seed: int = 0,
                 device_rng: bool = False, env=None, track_running_stats: bool = False):
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
        `sync_running_stats()` writes them back into the three modules."""
        self.G, self.cfg = G, config
        self.device = torch.device(device)
        m = config.mcts
        if m.max_iterations <= 0:
            raise ValueError("batched self-play needs max_iterations > 0 (the time-bounded mode is single-game only)")
        self.env = env if env is not None else BatchedConnectFour(G, device=device)
        assert getattr(self.env, "G", G) == G, "the environment batch must hold G games"
        self.A = self.env.num_actions
        self.h = RepresentationRunner(repr_net, device=device, track_running_stats=track_running_stats)
        # with a representation network that ends in a 1x1 convolution of a few channels, a residual-tower search starts
        # from those channels' pooled values (the bottleneck form of its expansions) instead of the 256-channel latent
        self._root_final = self.h.root_final()
        self.nets = make_networks(dyn_net, pred_net, device=device, track_running_stats=track_running_stats,
                                  root_final=self._root_final)
        self.mcts = BatchedMCTS(self.nets, G, m.max_iterations, m.model_look_ahead, m.discount_factor, m.noise_frac,
                                device=device, keep_latents=False)  # nothing here reads the children's latents
        self.history = DeviceHistory(G, repr_net.history_length, self.env.observation_space, device=device)
        self.rng = np.random.default_rng(seed)
        self.device_rng = device_rng
        if device_rng:
            self.gen = torch.Generator(device=self.device)
            self.gen.manual_seed(seed)
            self._alpha = torch.full((G, self.A), float(m.dirichlet_alpha), device=self.device)
            p = config.random_action_probability
            p = np.full(self.A, 1.0 / self.A) if p is None else np.asarray(p)
            self._p_random = torch.as_tensor(p.astype(np.float32)).to(self.device)
        self.alpha = m.dirichlet_alpha
        dev = self.device
        self.step_in_episode = torch.zeros(G, dtype=torch.int32, device=dev)
        self.action = torch.zeros(G, dtype=torch.int32, device=dev)
        self.env.reset()

    def sync_running_stats(self) -> None:
        """With track_running_stats: the advanced BatchNorm buffers of h, g and f go back into the modules."""
        self.h.sync_running_stats()
        if hasattr(self.nets, "sync_running_stats"):
            self.nets.sync_running_stats()

    def epsilon(self, training_steps: int) -> float:
        c = self.cfg
        return max(c.epsilon * (c.epsilon_decay ** training_steps), c.min_epsilon)

    def move(self, training_steps: int = 0):
        """One move of every game.  Returns device tensors (obs, policy32, value, action, reward,
        done) for this move; games that ended (or hit max_steps_per_episode) restart."""
        G, A, S, dev = self.G, self.A, self.mcts.S, self.device
        obs, to_play = self.env.observe()
        hist = self.history.push(obs, self.step_in_episode, None)  # self-play stores action id 0 with every frame
        root_form = bool(getattr(self.nets, "has_root_form", False))
        latent = self.h(hist, want_root_t=root_form).reshape(G, -1)
        root_t = self.h.root_t.reshape(G, -1) if root_form and self.h.root_t is not None else None
        if root_t is not None and not self.mcts.root_form_ok(root_t):
            root_t = None
        if root_t is not None:
            seen = tuple((t._version, t.data_ptr()) for t in self._root_final.parameters())
            if seen != getattr(self, "_root_seen", seen):
                self.nets.refresh()  # the composed root weights hold a copy of the final convolution's
            self._root_seen = seen
        eps = self.epsilon(training_steps)
        if self.device_rng:
            rand = torch.randint(0, A, (G, S + 1), device=dev, dtype=torch.int32, generator=self.gen)
            # Dirichlet(alpha) = normalised Gamma(alpha, 1) draws
            gam = torch._standard_gamma(self._alpha, generator=self.gen).clamp_min_(1e-30)
            eta = (gam / gam.sum(dim=1, keepdim=True)).contiguous()
            explore = (torch.rand(G, device=dev, generator=self.gen) < eps).to(torch.int8)
            rnd = torch.multinomial(self._p_random, G, replacement=True, generator=self.gen).to(torch.int32)
        else:
            rand = torch.from_numpy(self.rng.integers(0, A, size=(G, S + 1), dtype=np.int32)).to(dev)
            eta = torch.from_numpy(self.rng.dirichlet(np.full(A, self.alpha), size=G).astype(np.float32)).to(dev)
            explore = torch.from_numpy((self.rng.random(G) < eps).astype(np.int8)).to(dev)
            p = self.cfg.random_action_probability
            rnd = torch.from_numpy(self.rng.choice(A, size=G, p=p).astype(np.int32)).to(dev)
        res = self.mcts.run(latent if root_t is None else None, to_play, eta, rand, root_t=root_t)
        pick_action(res.policy32, explore, rnd, self.action)
        reward, done = self.env.step(self.action)
        self.step_in_episode += 1
        ended = (done.bool() | (self.step_in_episode >= self.cfg.max_steps_per_episode)).to(torch.int8)
        out = (obs, res.policy32, res.value, self.action.clone(), reward, ended)
        self.env.reset(ended)
        self.step_in_episode.masked_fill_(ended.bool(), 0)
        return out

    def generate(self, num_moves: int, training_steps: int = 0):
        """Plays `num_moves` lock-step moves and returns the finished episodes as lists of Chunk
        (states on the CPU), in the reference's Episode/Chunk format."""
        from muzero_amd.training_data_generator import Chunk, Episode
        rec = [self.move(training_steps) for _ in range(num_moves)]
        obs = torch.stack([r[0] for r in rec]).cpu()
        pol = torch.stack([r[1] for r in rec]).cpu()
        val = torch.stack([r[2] for r in rec]).cpu()
        act = torch.stack([r[3] for r in rec]).cpu()
        rew = torch.stack([r[4] for r in rec]).cpu()
        end = torch.stack([r[5] for r in rec]).cpu()
        episodes = []
        for g in range(self.G):
            cur = []
            for t in range(num_moves):
                cur.append(Chunk(state=obs[t, g].unsqueeze(0), policy=pol[t, g], reward=float(rew[t, g]),
                                 value=float(val[t, g]), best_action=int(act[t, g])))
                if end[t, g]:
                    episodes.append(Episode(chunks=cur))
                    cur = []
        return episodes
