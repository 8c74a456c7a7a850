"""Connect Four on the GPU: G concurrent games as bitboards (kernels in csrc/mz_env.hip).

`BatchedConnectFour` is the engine-facing form.  `ConnectFour` keeps the reference adapter's
single-game interface (src/environments/connect_four.py:18-72) on top of a one-game batch.
The rules restate PettingZoo's connect_four_v3 as seen through that adapter (step reports the
NEXT player's reward/termination); they are not pinned by any reference fixture.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import torch

from muzero_amd import _hip
from muzero_amd._hip import ptr
from muzero_amd.config import ConnectFourConfig


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class BatchedConnectFour:
    num_actions = 7
    observation_space = (2, 6, 7)

    def __init__(self, G: int, device="cuda"):
        self.G = G
        self.device = torch.device(device)
        self.pieces = torch.zeros((G, 2), dtype=torch.int64, device=self.device)  # u64 bit patterns
        self.to_move = torch.zeros(G, dtype=torch.int8, device=self.device)
        self.done = torch.zeros(G, dtype=torch.int8, device=self.device)
        self.c = _hip.C4State(G, 0, ptr(self.pieces), ptr(self.to_move), ptr(self.done))

    def reset(self, mask: torch.Tensor | None = None) -> None:
        if mask is not None:
            assert mask.dtype == torch.int8 and mask.numel() == self.G
        _hip.check(_hip.lib().mz_c4_reset(C.byref(self.c), ptr(mask), _stream()), "mz_c4_reset")

    def step(self, action: torch.Tensor, reward: torch.Tensor | None = None, done: torch.Tensor | None = None):
        assert action.dtype == torch.int32 and action.numel() == self.G and action.is_contiguous()
        reward = reward if reward is not None else torch.empty(self.G, dtype=torch.float32, device=self.device)
        done = done if done is not None else torch.empty(self.G, dtype=torch.int8, device=self.device)
        _hip.check(_hip.lib().mz_c4_step(C.byref(self.c), ptr(action), ptr(reward), ptr(done), _stream()), "mz_c4_step")
        return reward, done

    def move_end(self, policy32: torch.Tensor, explore, random_action, step_in_episode: torch.Tensor, max_steps: int,
                 action: torch.Tensor):
        """Action pick, step, episode bookkeeping and the restart of ended games for all G games in one launch
        (mz_c4_move_end: what pick_action, step, mz_episode_end and reset do in a row).  `action` i32[G] receives the
        actions; returns (a fresh copy of the actions, reward f32[G], ended i8[G])."""
        G, A = policy32.shape
        assert G == self.G and policy32.dtype == torch.float32 and policy32.is_contiguous()
        assert step_in_episode.dtype == torch.int32 and action.dtype == torch.int32
        played = torch.empty(G, dtype=torch.int32, device=self.device)
        reward = torch.empty(G, dtype=torch.float32, device=self.device)
        ended = torch.empty(G, dtype=torch.int8, device=self.device)
        _hip.check(_hip.lib().mz_c4_move_end(C.byref(self.c), ptr(policy32), A, ptr(explore), ptr(random_action), ptr(step_in_episode),
                                             int(max_steps), ptr(action), ptr(played), ptr(reward), ptr(ended), _stream()),
                   "mz_c4_move_end")
        return played, reward, ended

    def observe(self, obs: torch.Tensor | None = None, to_play: torch.Tensor | None = None):
        obs = obs if obs is not None else torch.empty((self.G, 2, 6, 7), dtype=torch.float32, device=self.device)
        to_play = to_play if to_play is not None else torch.empty(self.G, dtype=torch.int8, device=self.device)
        _hip.check(_hip.lib().mz_c4_observe(C.byref(self.c), ptr(obs), ptr(to_play), _stream()), "mz_c4_observe")
        return obs, to_play


class ConnectFour:
    """Single-game adapter with the reference's method set."""

    def __init__(self, config: ConnectFourConfig | None = None, device="cuda"):
        self.config = config or ConnectFourConfig()
        self.device = device
        self.games = BatchedConnectFour(1, device="cuda")
        self.games.reset()

    def get_to_play(self) -> int:
        return int(self.games.observe()[1].item())

    def get_action_space(self) -> tuple[int, ...]:
        return tuple(range(7))

    def get_observation_space(self) -> tuple[int, ...]:
        return (2, 6, 7)

    def get_state(self) -> torch.Tensor:
        return self.games.observe()[0].to(self.device)

    def reset(self) -> torch.Tensor:
        self.games.reset()
        return self.get_state()

    def step(self, action: int) -> tuple[torch.Tensor, float, bool]:
        a = torch.tensor([int(action)], dtype=torch.int32, device="cuda")
        reward, done = self.games.step(a)
        return self.get_state(), float(reward.item()), bool(done.item())

    def render(self) -> Any:
        p = self.games.pieces[0].tolist()
        rows = []
        for r in range(6):
            rows.append("".join("XO."[0 if (p[0] >> (c * 7 + 5 - r)) & 1 else 1 if (p[1] >> (c * 7 + 5 - r)) & 1 else 2]
                                for c in range(7)))
        return "\n".join(rows)

    def close(self) -> None:
        pass
