"""Fixed-shape episode store on the device: the actor -> learner hand-off without Python objects.

The reference keeps self-play output as lists of `Chunk` dataclasses and pickles them
(src/training_data_generator.py:28-51,188-235); the learner then slices windows out of those
lists one sample at a time (src/training.py:131-160).  With thousands of concurrent games that
costs more than the search, so the batched engine records every move of every game straight
into tensors `[slot, step, ...]` and windows are cut out of them with gathers:

    store.record(obs, policy, value, action, reward, ended)   # once per lock-step move
    batch = store.sample_windows(B, lookback, roll_ahead, K)  # -> muzero_amd.learner.WindowBatch

`to_episodes()` converts back to the reference's Episode/Chunk lists, and
`tests/test_episode_store.py` checks that windows cut from the tensors equal the windows the
list-based path (muzero_amd.learner.windows_to_batch) cuts from those lists.

A store on a CUDA device samples, cuts and re-prioritises with the kernels of libmuzero_hip.so
(`mz_per_sample`, `mz_replay_cut_windows`, `mz_per_update`: one launch each); a store on the CPU
(host-side tests of the window semantics) evaluates the same definitions with torch operations.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from muzero_amd.learner import WindowBatch


class DeviceEpisodeStore:
    def __init__(self, games: int, max_steps: int, obs_shape, num_actions: int, capacity: int, device="cuda"):
        """`capacity`: finished episodes kept (a ring: the oldest is overwritten)."""
        if capacity < games:
            # every game may finish in the same move: the finished ring must hold them all, or two games
            # would be copied into one slot in one record() call
            raise ValueError(f"capacity ({capacity}) must be at least the number of games ({games})")
        self.G, self.T, self.A, self.cap = games, max_steps, num_actions, capacity
        self.device = torch.device(device)
        C, H, W = obs_shape
        dev = self.device

        def buf(n, *shape, dtype=torch.float32):
            return torch.zeros((n, max_steps) + shape, dtype=dtype, device=dev)

        # episodes in progress, one row per game
        self.cur = dict(obs=buf(games, C, H, W), policy=buf(games, num_actions), value=buf(games, dtype=torch.float64),
                        reward=buf(games), action=buf(games, dtype=torch.int64))
        self.cur_len = torch.zeros(games, dtype=torch.int64, device=dev)
        # finished episodes (+ one scratch row that absorbs the rows of games still in progress,
        # so that a move is recorded with fixed-shape copies and no host synchronisation)
        self.done = dict(obs=buf(capacity + 1, C, H, W), policy=buf(capacity + 1, num_actions),
                         value=buf(capacity + 1, dtype=torch.float64), reward=buf(capacity + 1),
                         action=buf(capacity + 1, dtype=torch.int64))
        self.done_len = torch.zeros(capacity + 1, dtype=torch.int64, device=dev)
        self._n_finished = torch.zeros((), dtype=torch.int64, device=dev)  # episodes ever finished
        # episode priorities of the prioritised replay (src/replay_buffer.py:39-44,121-130)
        self.priority = torch.zeros(capacity + 1, dtype=torch.float64, device=dev)
        self.max_priority = torch.ones((), dtype=torch.float64, device=dev)

    def record(self, obs, policy, value, action, reward, ended) -> None:
        """One lock-step move of all games (the tuple SelfPlayEngine.move returns).  Games whose
        `ended` flag is set (or that reach max_steps) are moved to the finished ring, slot =
        running count modulo capacity.  Entirely on the device: nothing here waits for the GPU."""
        g = torch.arange(self.G, device=self.device)
        t = self.cur_len
        self.cur["obs"][g, t] = obs.reshape(self.G, *self.cur["obs"].shape[2:]).to(torch.float32)
        self.cur["policy"][g, t] = policy.to(torch.float32)
        self.cur["value"][g, t] = value.to(torch.float64)
        self.cur["reward"][g, t] = reward.to(torch.float32)
        self.cur["action"][g, t] = action.to(torch.int64)
        new_len = t + 1
        fin = ended.to(torch.bool) | (new_len >= self.T)
        order = torch.cumsum(fin.to(torch.int64), 0) - 1
        slot = torch.where(fin, (self._n_finished + order) % self.cap, torch.full_like(order, self.cap))
        for k in self.cur:
            self.done[k].index_copy_(0, slot, self.cur[k])
        self.done_len.index_copy_(0, slot, new_len)
        self.done_len[self.cap] = 0
        self.priority.index_copy_(0, slot, self.max_priority.expand(self.G))  # new episodes: the largest priority so far
        self.cur_len = torch.where(fin, torch.zeros_like(new_len), new_len)
        self._n_finished = self._n_finished + fin.sum()

    @property
    def n_finished(self) -> int:
        return int(self._n_finished.item())

    @property
    def size(self) -> int:
        return min(self.n_finished, self.cap)

    def eligible(self, lookback: int, roll_ahead: int) -> torch.Tensor:
        """Slots whose episode is long enough for a window (src/training.py:216-221)."""
        return torch.nonzero(self.done_len[: self.size] >= lookback + roll_ahead + 1).flatten()

    def cut_windows(self, slots: torch.Tensor, k: torch.Tensor, lookback: int, roll_ahead: int, history_length: int,
                    discount: float, z_targets, is_weights=None) -> WindowBatch:
        """Windows (slot[i], k[i]).  History: the ring buffer of bptt (src/training.py:252-262) is
        filled with the first past frame and the later ones are added, so slot j of the K-frame
        ring ends up holding past frame start + 1 + a, a the largest add index = j (mod K), or the
        first past frame if there was none.  `z_targets(R, V, discount) -> Z` (f64 [B, w+1])."""
        dev, K, w = self.device, history_length, roll_ahead
        B = int(slots.numel())
        if dev.type == "cuda":
            return self._cut_windows_kernel(slots, k, lookback, roll_ahead, history_length, discount, is_weights)
        start = torch.clamp(k - lookback, min=0)
        n_add = k - start - 1
        j = torch.arange(K, device=dev)[None, :]
        last = j + K * torch.div(n_add[:, None] - 1 - j, K, rounding_mode="floor")
        frame = torch.where(n_add[:, None] > j, start[:, None] + 1 + last, start[:, None])  # [B, K]
        obs = self.done["obs"][slots[:, None], frame]                                       # [B, K, C, H, W]
        act = self.done["action"][slots[:, None], frame].to(torch.float32)                  # [B, K]
        C, H, W = obs.shape[2:]
        planes = act[:, :, None, None].expand(B, K, H, W)
        history = torch.cat([obs.reshape(B, K * C, H, W), planes], dim=1)
        steps = k[:, None] + torch.arange(w + 1, device=dev)[None, :]
        sl = slots[:, None]
        R = self.done["reward"][sl, steps].to(torch.float64)
        V = self.done["value"][sl, steps]
        Z = z_targets(R, V, discount)
        isw = torch.ones(B, device=dev) if is_weights is None else torch.as_tensor(is_weights, dtype=torch.float32, device=dev)
        return WindowBatch(history, self.done["action"][sl, steps], self.done["policy"][sl, steps],
                           torch.as_tensor(Z, device=dev).to(torch.float32), R.to(torch.float32), isw)

    def _pow_tab(self, discount: float, w: int) -> torch.Tensor:
        key = (float(discount), int(w))
        tab = getattr(self, "_pow_tabs", {}).get(key)
        if tab is None:
            from muzero_amd import _hip
            host = np.zeros(w + 2, np.float64)
            _hip.check(_hip.lib().mz_gamma_powers(float(discount), w, _hip.ptr(host)), "mz_gamma_powers")
            tab = torch.from_numpy(host).to(self.device)
            self.__dict__.setdefault("_pow_tabs", {})[key] = tab
        return tab

    def _cut_windows_kernel(self, rows, k, lookback, roll_ahead, history_length, discount, is_weights) -> WindowBatch:
        """cut_windows in one launch (mz_replay_cut_windows), value targets included."""
        from muzero_amd import _hip
        from muzero_amd._hip import ptr
        dev, K, w = self.device, history_length, roll_ahead
        B = int(rows.numel())
        obs = self.done["obs"]
        C_, H, W = obs.shape[2:]
        rows32 = rows.to(torch.int32).contiguous()
        k32 = k.to(torch.int32).contiguous()
        history = torch.empty((B, C_ * K + K, H, W), dtype=torch.float32, device=dev)
        actions = torch.empty((B, w + 1), dtype=torch.int64, device=dev)
        policies = torch.empty((B, w + 1, self.A), dtype=torch.float32, device=dev)
        z = torch.empty((B, w + 1), dtype=torch.float32, device=dev)
        rewards = torch.empty((B, w + 1), dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().mz_replay_cut_windows(
            ptr(obs), ptr(self.done["action"]), ptr(self.done["policy"]), ptr(self.done["value"]), ptr(self.done["reward"]),
            self.T, C_, H, W, self.A, ptr(rows32), ptr(k32), B, lookback, w, K, ptr(self._pow_tab(discount, w)),
            ptr(history), ptr(actions), ptr(policies), ptr(z), ptr(rewards), torch.cuda.current_stream().cuda_stream),
            "mz_replay_cut_windows")
        isw = torch.ones(B, device=dev) if is_weights is None else torch.as_tensor(is_weights, dtype=torch.float32, device=dev)
        return WindowBatch(history, actions, policies, z, rewards, isw)

    def sample_windows(self, batch_size: int, lookback: int, roll_ahead: int, history_length: int, discount: float,
                       z_targets, generator=None):
        """Uniform over eligible episodes, window start uniform in [lookback, len - roll_ahead - 1]
        as random.randrange does (src/training.py:139-142).  Returns (WindowBatch, slots, k)."""
        el = self.eligible(lookback, roll_ahead)
        if el.numel() == 0:
            raise ValueError("no stored episode is long enough for a training window")
        pick = torch.randint(0, int(el.numel()), (batch_size,), device=self.device, generator=generator)
        slots = el[pick]
        span = self.done_len[slots] - roll_ahead - lookback  # number of admissible starts
        u = torch.rand(batch_size, device=self.device, generator=generator)
        k = lookback + torch.minimum((u * span).long(), span - 1)
        return self.cut_windows(slots, k, lookback, roll_ahead, history_length, discount, z_targets), slots, k

    def sampling_probabilities(self, alpha: float) -> torch.Tensor:
        """P(i) = p_i^alpha / sum_j p_j^alpha over the stored episodes, uniform if the sum
        vanishes (src/replay_buffer.py:88-96).  f64[size]."""
        n = self.size
        scaled = self.priority[:n] ** alpha
        total = scaled.sum()
        uniform = torch.full_like(scaled, 1.0 / max(n, 1))
        return torch.where(total < 1e-8, uniform, scaled / total)

    def sample_prioritized(self, batch_size: int, lookback: int, roll_ahead: int, history_length: int,
                           discount: float, z_targets, alpha: float = 0.6, beta: float = 1.0, generator=None,
                           uniforms=None):
        """ReplayBuffer.sample_batch + the learner's window cut, on the device
        (src/replay_buffer.py:72-119, src/training.py:115-142): min(batch_size, n) episodes drawn
        with replacement with P(i) ~ priority^alpha, importance weights (n P(i))^-beta divided
        by their maximum over the draw, episodes too short for a window dropped afterwards
        (src/training.py:210-221), window start uniform over the admissible ones.
        The batch keeps its shape: a dropped draw stays in it as a window of the longest stored
        episode with importance weight 0 (no contribution to the weighted loss, its gradient or
        the weight total) and slot = capacity (the scratch row, ignored by update_priorities), so
        the learner sees `min(batch_size, n)` windows every step instead of a new shape per step.

        `uniforms` (CUDA stores): (u f64[B], u_start f64[B]) to use instead of torch's generator --
        with u = np.random.random_sample(B) the draw equals np.random.choice's on the same stream, i.e.
        the episodes ReplayBuffer.sample_batch would pick.

        Returns (WindowBatch, slots, k); `slots` are what update_priorities takes.  Each kept
        window carries the weight and the slot of ITS episode (the reference indexes both by the
        position in the filtered list, src/training.py:133,164, which pairs a window with the
        weight and index of another draw once an episode has been dropped)."""
        n = self.size
        if n == 0:
            raise ValueError("the store holds no finished episode")
        need = lookback + roll_ahead + 1
        longest = torch.argmax(self.done_len[:n])
        if int(self.done_len[longest]) < need:
            raise ValueError("no stored episode is long enough for a training window")
        if self.device.type == "cuda":
            return self._sample_prioritized_kernel(min(batch_size, n), n, lookback, roll_ahead, history_length, discount,
                                                   alpha, beta, generator, uniforms)
        probs = self.sampling_probabilities(alpha)
        draw = torch.multinomial(probs, min(batch_size, n), replacement=True, generator=generator)
        weights = (n * probs[draw]) ** (-beta)
        weights = weights / weights.max()
        keep = self.done_len[draw] >= need
        cut_from = torch.where(keep, draw, longest)
        span = self.done_len[cut_from] - roll_ahead - lookback
        u = torch.rand(cut_from.numel(), device=self.device, generator=generator)
        k = lookback + torch.minimum((u * span).long(), span - 1)
        batch = self.cut_windows(cut_from, k, lookback, roll_ahead, history_length, discount, z_targets,
                                 is_weights=torch.where(keep, weights, torch.zeros_like(weights)).to(torch.float32))
        return batch, torch.where(keep, draw, torch.full_like(draw, self.cap)), k

    def _sample_prioritized_kernel(self, B, n, lookback, roll_ahead, history_length, discount, alpha, beta, generator,
                                   uniforms):
        from muzero_amd import _hip
        from muzero_amd._hip import ptr
        dev = self.device
        if uniforms is None:
            u = torch.rand(B, dtype=torch.float64, device=dev, generator=generator)
            u_start = torch.rand(B, dtype=torch.float64, device=dev, generator=generator)
        else:
            u, u_start = (torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float64).to(dev).contiguous()
                          for x in uniforms)
            assert u.numel() == B and u_start.numel() == B
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.last_probs, cdf = torch.empty(n, **f64), torch.empty(n, **f64)
        draw, self.last_weights = torch.empty(B, **i32), torch.empty(B, **f64)
        weight32 = torch.empty(B, dtype=torch.float32, device=dev)
        slot, cut_from, start = torch.empty(B, **i32), torch.empty(B, **i32), torch.empty(B, **i32)
        _hip.check(_hip.lib().mz_per_sample(
            ptr(self.priority), n, float(alpha), float(beta), ptr(u), B, ptr(self.last_probs), ptr(cdf), ptr(draw),
            ptr(self.last_weights), ptr(self.done_len), lookback + roll_ahead + 1, lookback, roll_ahead, self.cap, ptr(u_start),
            ptr(weight32), ptr(slot), ptr(cut_from), ptr(start), torch.cuda.current_stream().cuda_stream), "mz_per_sample")
        self.last_draw = draw
        batch = self._cut_windows_kernel(cut_from, start, lookback, roll_ahead, history_length, discount, weight32)
        return batch, slot.to(torch.int64), start.to(torch.int64)

    def update_priorities(self, slots: torch.Tensor, errors: torch.Tensor) -> None:
        """priority[slot] = |error| + 1e-6, the running maximum follows
        (src/replay_buffer.py:121-130).  When a slot occurs more than once the last occurrence
        wins, as in the reference's loop; entries with slot = capacity are ignored."""
        if self.device.type == "cuda":
            from muzero_amd import _hip
            from muzero_amd._hip import ptr
            s32 = slots.to(device=self.device, dtype=torch.int32).contiguous()
            err = errors.to(device=self.device, dtype=torch.float32).contiguous()
            _hip.check(_hip.lib().mz_per_update(ptr(self.priority), ptr(self.max_priority), ptr(s32), ptr(err), int(s32.numel()),
                                                self.cap, torch.cuda.current_stream().cuda_stream), "mz_per_update")
            self.priority[self.cap] = 0.0  # the scratch row never carries a priority
            return
        slots = slots.to(self.device)
        pri = errors.to(device=self.device, dtype=torch.float64).abs() + 1e-6
        pos = torch.arange(slots.numel(), device=self.device)
        last = torch.full((self.cap + 1,), -1, dtype=torch.int64, device=self.device)
        last.scatter_reduce_(0, slots, pos, reduce="amax")
        winner = last[slots] == pos
        self.priority.index_copy_(0, slots[winner], pri[winner])
        self.priority[self.cap] = 0.0
        real = slots < self.cap
        self.max_priority = torch.maximum(self.max_priority, torch.where(real, pri, torch.zeros_like(pri)).max())

    def to_episodes(self):
        """The stored episodes as the reference's Episode/Chunk lists (states on the CPU)."""
        from muzero_amd.training_data_generator import Chunk, Episode
        out = []
        host = {k: v[: self.size].cpu() for k, v in self.done.items()}
        lens = self.done_len[: self.size].cpu()
        for s in range(self.size):
            out.append(Episode(chunks=[
                Chunk(state=host["obs"][s, t].unsqueeze(0), policy=host["policy"][s, t], reward=float(host["reward"][s, t]),
                      value=float(host["value"][s, t]), best_action=int(host["action"][s, t])) for t in range(int(lens[s]))]))
        return out
