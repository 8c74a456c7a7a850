"""G single-game simulators on the host cores behind the batched-environment interface of the
self-play engine (the one `BatchedConnectFour` offers on the GPU): `observe()`, `step(actions)`,
`reset(mask)`.

For environments whose rules live in host-side third-party simulators (the reference's
CarRacing / FlappyBird / LunarLander adapters, src/environments/*.py): any object with the
reference's `Environment` methods (src/environment.py:6-94) works.  Observations are written
into one pinned host buffer and reach the device as ONE asynchronous copy per move; actions
come back as one small copy.  Environment steps run on a thread pool (the simulators release
the GIL in their native step; pure-Python ones simply run in turn).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import torch


class HostEnvPool:
    def __init__(self, envs, device="cuda", workers: int = 0):
        assert len(envs) > 0
        self.envs = list(envs)
        self.G = len(self.envs)
        self.device = torch.device(device)
        self.num_actions = int(self.envs[0].get_action_space()[0])
        self.observation_space = tuple(int(v) for v in self.envs[0].get_observation_space())
        pin = self.device.type == "cuda"
        self._obs = torch.zeros((self.G,) + self.observation_space, dtype=torch.float32, pin_memory=pin)
        self._to_play = torch.zeros(self.G, dtype=torch.int8, pin_memory=pin)
        self._reward = torch.zeros(self.G, dtype=torch.float32, pin_memory=pin)
        self._done = torch.zeros(self.G, dtype=torch.int8, pin_memory=pin)
        # numpy views of the same (pinned) memory: scalar writes through them cost a tenth of a tensor item assignment
        self._to_play_np, self._reward_np, self._done_np = self._to_play.numpy(), self._reward.numpy(), self._done.numpy()
        self._pool = ThreadPoolExecutor(max_workers=workers) if workers > 1 else None
        for i, e in enumerate(self.envs):
            self._store(i, e.get_state())

    def _store(self, i: int, state: torch.Tensor) -> None:
        # adapters return [1, C, H, W] (src/environments/connect_four.py:48-50 and siblings)
        self._obs[i].copy_(state.reshape(self.observation_space))
        self._to_play_np[i] = int(self.envs[i].get_to_play())

    def _map(self, fn, items) -> None:
        if self._pool is None:
            for it in items:
                fn(it)
        else:
            list(self._pool.map(fn, items))

    def observe(self):
        """obs f32[G, C, H, W] and to_play i8[G] on the device (state before the move)."""
        return (self._obs.to(self.device, non_blocking=True).clone(),
                self._to_play.to(self.device, non_blocking=True).clone())

    def step_host(self, acts):
        """acts: G ints on the host.  Steps every simulator; returns reward f32[G], done i8[G] as host tensors
        (copies).  Touches no device: a caller that overlaps this with GPU work never waits for the stream."""
        def one(i: int) -> None:
            state, reward, done = self.envs[i].step(int(acts[i]))
            self._reward_np[i] = float(reward)
            self._done_np[i] = 1 if done else 0
            self._store(i, state)

        self._map(one, range(self.G))
        return self._reward.clone(), self._done.clone()

    def step(self, action: torch.Tensor):
        """action i32[G] (device or host).  Returns reward f32[G], done i8[G] on the device."""
        self.step_host(action.detach().to("cpu").tolist())
        return (self._reward.to(self.device, non_blocking=True).clone(),
                self._done.to(self.device, non_blocking=True).clone())

    def reset_host(self, which) -> None:
        """Restarts the games with the given indices (host ints)."""
        def one(i: int) -> None:
            self._store(i, self.envs[i].reset())

        self._map(one, which)

    def reset(self, mask: torch.Tensor | None = None) -> None:
        """Restarts the games whose mask entry is non-zero (all games when mask is None)."""
        self.reset_host(range(self.G) if mask is None else torch.nonzero(mask.detach().to("cpu")).reshape(-1).tolist())

    def close(self) -> None:
        if self._pool is not None:
            self._pool.shutdown()
        for e in self.envs:
            if hasattr(e, "close"):
                e.close()
