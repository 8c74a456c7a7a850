"""Episode-level prioritised replay (src/replay_buffer.py:10-133): host-side bookkeeping over
at most `buffer_size` (tens of) episodes, sampled with numpy's global RNG like the reference.
Metrics that the reference sends to Weights & Biases go to an optional `logger` callable."""
from __future__ import annotations

import numpy as np

np.random.seed(0)  # the reference seeds numpy's global RNG when this module is imported (replay_buffer.py:7)


class ReplayBuffer:
    def __init__(self, buffer_size: int, batch_size: int, alpha=0.6, beta: float = 1.0, logger=None):
        if buffer_size <= 0:
            raise ValueError("Buffer size must be greater than 0.")
        if batch_size <= 0:
            raise ValueError("Batch size must be greater than 0.")
        if buffer_size < batch_size:
            raise ValueError("Buffer size must be greater than batch size.")
        self.buffer_capacity = buffer_size
        self.batch_size = batch_size
        self.alpha = alpha
        self.beta = beta
        self.episode_buffer: list[tuple] = []  # (Episode, priority)
        self.max_priority = 1.0
        self.logger = logger

    def _log(self, payload: dict) -> None:
        if self.logger is not None:
            self.logger(payload)

    def add_episodes(self, episodes: list) -> None:
        self.episode_buffer.extend((ep, self.max_priority) for ep in episodes)
        extra = len(self.episode_buffer) - self.buffer_capacity
        if extra > 0:
            del self.episode_buffer[:extra]  # FIFO: keep the newest
        if self.logger is not None and self.episode_buffer:
            totals = [sum(c.reward for c in ep.chunks) for ep, _ in self.episode_buffer]
            self._log({
                "replay/average_states_per_episode": float(np.mean([len(ep.chunks) for ep, _ in self.episode_buffer])),
                "replay/total_episodes_in_buffer": len(self.episode_buffer),
                "replay/max_reward": float(np.max(totals)),
                "replay/median_reward": float(np.median(totals)),
            })

    def sample_batch(self) -> tuple[list, list[int], list[float]]:
        n = len(self.episode_buffer)
        if n == 0:
            return [], [], []
        scaled = [p ** self.alpha for _, p in self.episode_buffer]
        total = sum(scaled)
        probs = [1.0 / n] * n if total < 1e-8 else [s / total for s in scaled]
        indices = np.random.choice(n, size=min(self.batch_size, n), replace=True, p=probs)
        weights = [(n * probs[i]) ** (-self.beta) for i in indices]
        top = max(weights)
        weights = [w / top for w in weights]
        if self.logger is not None:
            entropy = -np.dot(probs, np.log(np.array(probs, dtype=np.float32) + 1e-8))
            self._log({"replay/sampling_entropy": float(entropy)})
        return [self.episode_buffer[i][0] for i in indices], list(indices), weights

    def update_priorities(self, indices: list[int], new_errors: list[float]) -> None:
        for idx, err in zip(indices, new_errors):
            pri = abs(err) + 1e-6
            self.max_priority = max(self.max_priority, pri)
            self.episode_buffer[idx] = (self.episode_buffer[idx][0], pri)

    def __len__(self):
        return len(self.episode_buffer)
