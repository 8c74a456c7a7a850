"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/c4_oracle.c plus a duck-typed
single-game environment with the reference adapter's interface
(src/environments/connect_four.py:18-72).  The reward convention of PettingZoo is UNPINNED, legality
and the win test are pinned by published position counts (see c4_oracle.c)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import lib


class _State(C.Structure):
    _fields_ = [("board", C.c_int8 * 42), ("to_move", C.c_int8), ("done", C.c_int8)]


class C4:
    def __init__(self):
        self.s = _State()
        self.reset()

    def reset(self):
        lib().mzo_c4_reset(C.byref(self.s))

    def step(self, action: int):
        d = C.c_int(0)
        f = lib().mzo_c4_step
        f.restype = C.c_float
        r = f(C.byref(self.s), C.c_int(int(action)), C.byref(d))
        return float(r), bool(d.value)

    def observe(self) -> np.ndarray:
        obs = np.zeros((2, 6, 7), np.float32)
        lib().mzo_c4_observe(C.byref(self.s), obs.ctypes.data_as(C.POINTER(C.c_float)))
        return obs

    @property
    def to_play(self) -> int:
        return int(lib().mzo_c4_to_play(C.byref(self.s)))

    @property
    def board(self) -> np.ndarray:
        return np.array(self.s.board, dtype=np.int8).reshape(6, 7)


class C4Env:
    """Environment-protocol wrapper (src/environment.py:6-94) over the oracle rules, used to
    drive the reference's / the port's single-game self-play loop in tests."""

    def __init__(self):
        import torch
        self._torch = torch
        self.g = C4()

    def get_to_play(self):
        return self.g.to_play

    def get_action_space(self):
        return tuple(range(7))

    def get_observation_space(self):
        return (2, 6, 7)

    def get_state(self):
        return self._torch.from_numpy(self.g.observe()).unsqueeze(0)

    def reset(self):
        self.g.reset()
        return self.get_state()

    def step(self, action):
        r, d = self.g.step(action)
        return self.get_state(), r, d

    def render(self):
        return self.g.board

    def close(self):
        pass
