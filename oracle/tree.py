"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/tree_oracle.c (stream-driven
MCTS.run for G trees on the CPU).  See that file for the reference citations."""
from __future__ import annotations

import ctypes as C

import numpy as np

from oracle import lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def run_batch(A, S, depth, root_to_play, prior_stream, reward_stream, rand_stream, sim_stream, eta,
              noise_frac, discount, want_trace=True):
    """All arrays are [G]-major numpy arrays:
    prior_stream/reward_stream f32[G,S+1,A], rand_stream i32[G,S+1], sim_stream f32[G,S,depth+1],
    eta f32[G,A], root_to_play i8[G].  Returns a dict of pool arrays [G,cap] plus leaf/child
    traces [G,S], policy f64[G,A] and utility f64[G]."""
    G = int(root_to_play.shape[0])
    cap = 1 + A + S * A
    prior_stream = np.ascontiguousarray(prior_stream, dtype=np.float32).reshape(G, S + 1, A)
    reward_stream = np.ascontiguousarray(reward_stream, dtype=np.float32).reshape(G, S + 1, A)
    rand_stream = np.ascontiguousarray(rand_stream, dtype=np.int32).reshape(G, S + 1)
    sim_stream = np.ascontiguousarray(sim_stream, dtype=np.float32).reshape(G, S, depth + 1)
    eta = np.ascontiguousarray(eta, dtype=np.float32).reshape(G, A)
    root_to_play = np.ascontiguousarray(root_to_play, dtype=np.int8)
    out = {
        "visit": np.zeros((G, cap), np.int32),
        "value_sum": np.zeros((G, cap), np.float64),
        "prior": np.zeros((G, cap), np.float64),
        "reward": np.zeros((G, cap), np.float32),
        "parent": np.full((G, cap), -1, np.int32),
        "first_child": np.full((G, cap), -1, np.int32),
        "to_play": np.zeros((G, cap), np.int8),
        "leaf": np.zeros((G, S), np.int32) if want_trace else None,
        "child": np.zeros((G, S), np.int32) if want_trace else None,
        "policy": np.zeros((G, A), np.float64),
        "utility": np.zeros((G,), np.float64),
    }
    f = lib().mzo_run_batch
    f.restype = None
    f(C.c_int32(G), C.c_int32(A), C.c_int32(S), C.c_int32(depth), C.c_int32(cap),
      _p(out["visit"], C.c_int32), _p(out["value_sum"], C.c_double), _p(out["prior"], C.c_double),
      _p(out["reward"], C.c_float), _p(out["parent"], C.c_int32), _p(out["first_child"], C.c_int32),
      _p(out["to_play"], C.c_int8), _p(root_to_play, C.c_int8), _p(prior_stream, C.c_float),
      _p(reward_stream, C.c_float), _p(rand_stream, C.c_int32), _p(sim_stream, C.c_float),
      _p(eta, C.c_float), C.c_double(noise_frac), C.c_double(discount),
      _p(out["leaf"], C.c_int32), _p(out["child"], C.c_int32), _p(out["policy"], C.c_double),
      _p(out["utility"], C.c_double))
    return out


class Tree:
    """Step-wise access to one oracle tree (select / expand / backup / noise / result) for
    unit-level checks such as the reference's selection and backup known-answer tests."""

    class _S(C.Structure):
        _fields_ = [("A", C.c_int32), ("cap", C.c_int32), ("n_nodes", C.c_int32),
                    ("visit", C.POINTER(C.c_int32)), ("value_sum", C.POINTER(C.c_double)),
                    ("prior", C.POINTER(C.c_double)), ("reward", C.POINTER(C.c_float)),
                    ("parent", C.POINTER(C.c_int32)), ("first_child", C.POINTER(C.c_int32)),
                    ("to_play", C.POINTER(C.c_int8))]

    def __init__(self, A: int, cap: int, root_to_play: int = 0):
        self.A, self.cap = A, cap
        self.visit = np.zeros(cap, np.int32)
        self.value_sum = np.zeros(cap, np.float64)
        self.prior = np.zeros(cap, np.float64)
        self.reward = np.zeros(cap, np.float32)
        self.parent = np.full(cap, -1, np.int32)
        self.first_child = np.full(cap, -1, np.int32)
        self.to_play = np.zeros(cap, np.int8)
        self.s = self._S(A, cap, 0, _p(self.visit, C.c_int32), _p(self.value_sum, C.c_double),
                         _p(self.prior, C.c_double), _p(self.reward, C.c_float),
                         _p(self.parent, C.c_int32), _p(self.first_child, C.c_int32),
                         _p(self.to_play, C.c_int8))
        lib().mzo_init(C.byref(self.s), C.c_int32(root_to_play))

    @property
    def n_nodes(self):
        return self.s.n_nodes

    def expand(self, leaf, prior32, reward32, rand_child):
        p = np.ascontiguousarray(prior32, np.float32)
        r = np.ascontiguousarray(reward32, np.float32)
        return lib().mzo_expand(C.byref(self.s), C.c_int32(leaf), _p(p, C.c_float), _p(r, C.c_float),
                                C.c_int32(rand_child))

    def root_noise(self, eta, noise_frac):
        e = np.ascontiguousarray(eta, np.float32)
        lib().mzo_root_noise(C.byref(self.s), _p(e, C.c_float), C.c_double(noise_frac))

    def select(self, root_f32: bool = True):
        d = C.c_int32(0)
        leaf = lib().mzo_select(C.byref(self.s), C.c_int32(int(root_f32)), C.byref(d))
        return leaf, d.value

    def backup(self, child, rewards, leaf_to_play, discount):
        r = np.ascontiguousarray(rewards, np.float32)
        f = lib().mzo_backup
        f.restype = C.c_double
        return f(C.byref(self.s), C.c_int32(child), _p(r, C.c_float), C.c_int32(len(r)),
                 C.c_int32(leaf_to_play), C.c_double(discount))

    def result(self):
        pol = np.zeros(self.A, np.float64)
        u = C.c_double(0)
        lib().mzo_result(C.byref(self.s), _p(pol, C.c_double), C.byref(u))
        return pol, u.value
