"""Frame ring buffer and history tensor (src/ring_buffer.py:6-83).

The single-game classes below keep the reference's interface on host/torch tensors.  The
batched engine keeps the same ring per game on the device and assembles the history with
the `mz_history_push_assemble` kernel (muzero_amd.selfplay.DeviceHistory)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor


@dataclass
class Frame:
    state: Tensor
    action: int


class FrameRingBuffer:
    """K slots.  `fill` replicates one frame into every empty slot without moving the write
    cursor; `add` appends while not full, else overwrites slot `index`; the cursor always
    advances.  `get_all` returns SLOT order, which is what the history tensor uses."""

    def __init__(self, size: int) -> None:
        self.size = size
        self.buffer: list[Frame] = []
        self.index = 0

    def add(self, item: Frame) -> None:
        if len(self.buffer) < self.size:
            self.buffer.append(item)
        else:
            self.buffer[self.index] = item
        self.index = (self.index + 1) % self.size

    def fill(self, item: Frame) -> None:
        self.buffer.extend([item] * (self.size - len(self.buffer)))

    def get_all(self) -> list[Frame]:
        return self.buffer

    def full(self) -> bool:
        return len(self.buffer) >= self.size


def make_history_tensor(buffer: FrameRingBuffer) -> Tensor:
    """[1, C*K + K, H, W]: the K stored frames concatenated on the channel axis (slot order)
    followed by K planes, each filled with the raw action id stored in its slot."""
    assert buffer.full(), "Ring buffer is not yet full"
    frames = buffer.get_all()
    img = torch.cat([f.state for f in frames], dim=1)
    acts = torch.tensor([float(f.action) for f in frames], dtype=img.dtype, device=img.device)
    planes = acts.view(1, -1, 1, 1).expand(1, len(frames), img.shape[2], img.shape[3])
    return torch.cat([img, planes], dim=1)
