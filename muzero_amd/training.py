"""Learner with the reference's interface (src/training.py:31-440): prioritised episode
sampling, unroll-window slicing, value targets, BPTT through h/g/f and checkpoint I/O.

What runs where: window slicing is host logic; the z-targets of a whole mini-batch are one
`mz_z_targets` kernel launch; the history tensors use muzero_amd.ring_buffer; the unrolled
forward/backward and the optimizer are stock PyTorch on the selected device (a batched,
collective-overlapped learner is the next widening step and is not claimed here).
"""
from __future__ import annotations

import datetime
import os
import random
import re

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.optim import lr_scheduler

from muzero_amd import _hip
from muzero_amd._hip import ptr
from muzero_amd.config import TrainingConfig
from muzero_amd.replay_buffer import ReplayBuffer
from muzero_amd.ring_buffer import Frame, FrameRingBuffer, make_history_tensor

FOLDER_REGEX = re.compile(r"^(\d+)_(\d{8}_\d{6})$")
BASE_PATH = "training_runs"

random.seed(0)  # the reference seeds Python's global RNG when this module is imported (:28)


def z_targets_device(R: Tensor, V: Tensor, gamma: float) -> Tensor:
    """z-targets for B unroll windows that already live on the GPU (f64 [B, w+1] in and out): one
    `mz_z_targets` launch, no host round trip (src/training.py:223-242)."""
    B, n = R.shape
    w = n - 1
    pw = np.zeros(w + 2, np.float64)
    _hip.check(_hip.lib().mz_gamma_powers(float(gamma), w, ptr(pw)), "mz_gamma_powers")
    d_r, d_v = R.to(torch.float64).contiguous(), V.to(torch.float64).contiguous()
    d_p = torch.from_numpy(pw).to(R.device)
    z = torch.empty((B, n), dtype=torch.float64, device=R.device)
    _hip.check(_hip.lib().mz_z_targets(ptr(d_r), ptr(d_v), ptr(d_p), B, w, ptr(z),
                                       torch.cuda.current_stream().cuda_stream), "mz_z_targets")
    return z


def compute_z_targets_batch(R: np.ndarray, V: np.ndarray, gamma: float) -> np.ndarray:
    """z-targets for B unroll windows on the GPU.  R, V: float64 [B, w+1]."""
    B, n = R.shape
    w = n - 1
    pw = np.zeros(w + 2, np.float64)
    _hip.check(_hip.lib().mz_gamma_powers(float(gamma), w, ptr(pw)), "mz_gamma_powers")
    d_r = torch.from_numpy(np.ascontiguousarray(R, np.float64)).cuda()
    d_v = torch.from_numpy(np.ascontiguousarray(V, np.float64)).cuda()
    d_p = torch.from_numpy(pw).cuda()
    z = torch.empty((B, n), dtype=torch.float64, device="cuda")
    _hip.check(_hip.lib().mz_z_targets(ptr(d_r), ptr(d_v), ptr(d_p), B, w, ptr(z),
                                       torch.cuda.current_stream().cuda_stream), "mz_z_targets")
    return z.cpu().numpy()


class NeuralNetworkManager:
    def __init__(self, config: TrainingConfig, repr_net, dyn_net, pred_net, device="cpu", logger=None):
        self.config = config
        self.lookback = config.look_back
        self.roll_ahead = config.roll_ahead
        self.mbs = config.mini_batch_size
        self.repr_net, self.dyn_net, self.pred_net = repr_net, dyn_net, pred_net
        self.loss_history: list[float] = []
        self.device = device
        self.logger = logger
        params = list(repr_net.parameters()) + list(dyn_net.parameters()) + list(pred_net.parameters())
        kind = config.optimizer.lower()
        if kind == "sgd":
            self.optimizer = torch.optim.SGD(params, lr=config.learning_rate, weight_decay=config.weight_decay,
                                             momentum=config.momentum)
        elif kind == "adam":
            self.optimizer = torch.optim.Adam(params, lr=config.learning_rate, betas=config.betas,
                                              weight_decay=config.weight_decay)
        elif kind == "adamw":
            self.optimizer = torch.optim.AdamW(params, lr=config.learning_rate, betas=config.betas,
                                               weight_decay=config.weight_decay)
        elif kind == "rmsprop":
            self.optimizer = torch.optim.RMSprop(params, lr=config.learning_rate, weight_decay=config.weight_decay)
        else:
            raise ValueError(f"Unsupported optimizer type: {config.optimizer}")
        # Only the schedules whose settings exist in TrainingConfig can be built; the reference's
        # "step", "multi_step" and "reduce_lr_on_plateau" branches read fields the config does
        # not define (src/training.py:91,95,105) and any other name (e.g. "cosine") means none.
        self.scheduler = None
        sched = config.lr_schedule.lower()
        if sched == "exponential":
            self.scheduler = lr_scheduler.ExponentialLR(self.optimizer, gamma=config.scheduler_gamma)
        elif sched == "cosine_annealing":
            self.scheduler = lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=config.scheduler_T_max,
                                                            eta_min=config.scheduler_eta_min)
        elif sched in ("step", "multi_step", "reduce_lr_on_plateau"):
            raise AttributeError(f"lr_schedule '{sched}' needs settings TrainingConfig does not define")

    # ------------------------------------------------------------------ training
    def train(self, replay_buffer: ReplayBuffer) -> float:
        last = 0.0
        for _ in range(self.mbs):
            batch, indices, is_weights = replay_buffer.sample_batch()
            batch = self._filter_for_valid_episodes(batch)
            self.optimizer.zero_grad()
            windows = []
            for local_idx, episode in enumerate(batch):
                # as in the reference, indices / is_weights are addressed by position in the
                # FILTERED batch (src/training.py:134,162)
                max_k = len(episode.chunks) - (self.roll_ahead + 1)
                if max_k < 0:
                    continue
                k = random.randrange(self.lookback, max_k + 1)
                windows.append((local_idx, episode, k))
            if not windows:
                continue
            w = self.roll_ahead
            R = np.array([[float(ep.chunks[i].reward) for i in range(k, k + w + 1)] for _, ep, k in windows], np.float64)
            V = np.array([[float(ep.chunks[i].value) for i in range(k, k + w + 1)] for _, ep, k in windows], np.float64)
            Z = self._compute_z_targets_batch(R, V)
            total = torch.zeros(1, dtype=torch.float32, device=self.device)
            sums = [torch.zeros(1, dtype=torch.float32, device=self.device) for _ in range(3)]
            errors, used_idx, used_w = [], [], []
            for row, (local_idx, ep, k) in enumerate(windows):
                start = max(0, k - self.lookback)
                past_states = [ep.chunks[i].state for i in range(start, k)]
                past_actions = [ep.chunks[i].best_action for i in range(start, k)]
                actions = [ep.chunks[i].best_action for i in range(k, k + w + 1)]
                policies = [ep.chunks[i].policy for i in range(k, k + w + 1)]
                p_loss, v_loss, r_loss = self.bptt(past_states, past_actions, actions,
                                                   (policies, [float(z) for z in Z[row]], [float(r) for r in R[row]]))
                step_loss = p_loss + v_loss + r_loss
                wgt = is_weights[local_idx]
                total += wgt * step_loss
                used_w.append(wgt)
                for acc, part in zip(sums, (p_loss, v_loss, r_loss)):
                    acc += part
                errors.append(step_loss.detach().abs().item())
                used_idx.append(indices[local_idx])
            norm = sum(used_w)
            total /= norm
            for acc in sums:
                acc /= norm
            total.backward()
            self.optimizer.step()
            if self.scheduler:
                self.scheduler.step()
            if self.logger is not None:
                self.logger({"loss/batch": total.item(), "loss/policy": sums[0].item(), "loss/value": sums[1].item(),
                             "loss/reward": sums[2].item(), "learning_rate": self.optimizer.param_groups[0]["lr"]})
            last = total.item()
            self.loss_history.append(last)
            replay_buffer.update_priorities(used_idx, errors)
        return last

    def _filter_for_valid_episodes(self, episodes: list) -> list:
        need = self.lookback + self.roll_ahead + 1
        keep = []
        for ep in episodes:
            if len(ep.chunks) >= need:
                keep.append(ep)
            else:
                print(f"Skipping episode with {len(ep.chunks)} chunks; requires at least {need}.")
        return keep

    def _compute_z_targets_batch(self, R: np.ndarray, V: np.ndarray) -> np.ndarray:
        return compute_z_targets_batch(R, V, self.config.discount_factor)

    def _compute_z_targets(self, Rb_k, Vb_k, gamma):
        """Single-window form with the reference's signature (src/training.py:223-242)."""
        R = np.array([[float(x) for x in Rb_k]], np.float64)
        V = np.array([[float(x) for x in Vb_k]], np.float64)
        return [float(z) for z in compute_z_targets_batch(R, V, gamma)[0]]

    def bptt(self, past_observations, past_actions, Ab_k, PVR):
        Pb_k, Zb_k, Rb_k = PVR
        dev = self.device
        ring = FrameRingBuffer(size=self.repr_net.history_length)
        ring.fill(Frame(state=past_observations[0], action=past_actions[0]))
        for obs, act in zip(past_observations[1:], past_actions[1:]):
            ring.add(Frame(state=obs, action=act))
        latent = self.repr_net(make_history_tensor(ring).to(dev))
        total_p = torch.zeros(1, dtype=torch.float32, device=dev)
        total_v = torch.zeros(1, dtype=torch.float32, device=dev)
        total_r = torch.zeros(1, dtype=torch.float32, device=dev)
        for i in range(self.roll_ahead):
            pred_p, pred_v = self.pred_net(latent)
            t_p = torch.as_tensor(Pb_k[i], dtype=torch.float32).to(dev)
            t_v = torch.tensor([Zb_k[i]], dtype=torch.float32, device=dev)
            t_r = torch.tensor([Rb_k[i]], dtype=torch.float32, device=dev)
            action = Ab_k[i]
            if not isinstance(action, torch.Tensor):
                action = torch.tensor([action], dtype=torch.long, device=dev)
            latent, pred_r = self.dyn_net(latent, action)
            total_p += self.policy_loss(t_p, pred_p[0])
            total_v += self.value_loss(t_v, pred_v[0])
            total_r += self.reward_loss(t_r, pred_r[0])
        final_p, final_v = self.pred_net(latent)
        total_v += self.value_loss(torch.tensor([Zb_k[self.roll_ahead]], dtype=torch.float32, device=dev), final_v[0])
        total_p += self.policy_loss(torch.as_tensor(Pb_k[self.roll_ahead], dtype=torch.float32).to(dev), final_p[0])
        return total_p, total_v, total_r

    def reward_loss(self, target: Tensor, pred: Tensor) -> Tensor:
        return F.mse_loss(pred, target) * self.config.reward_coefficient

    def value_loss(self, target: Tensor, pred: Tensor) -> Tensor:
        return F.mse_loss(pred, target) * self.config.value_coefficient

    def policy_loss(self, target: Tensor, pred: Tensor) -> Tensor:
        # cross entropy with the (already softmaxed) prediction used as logits, as the
        # reference does (src/training.py:320)
        return F.cross_entropy(pred, target) * self.config.policy_coefficient

    # ------------------------------------------------------------------ checkpoints
    def save_models(self, final_loss_val: float = 0.0, env_config=None, show_plot: bool = False):
        counter = self._get_next_model_counter(BASE_PATH)
        stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
        folder = f"{counter}_{stamp}"
        model_path = os.path.join(BASE_PATH, "models", folder)
        os.makedirs(model_path, exist_ok=True)
        torch.save(self.repr_net.state_dict(), os.path.join(model_path, "repr.pth"))
        torch.save(self.dyn_net.state_dict(), os.path.join(model_path, "dyn.pth"))
        torch.save(self.pred_net.state_dict(), os.path.join(model_path, "pred.pth"))
        info_path = os.path.join(BASE_PATH, "info", folder)
        os.makedirs(info_path, exist_ok=True)
        with open(os.path.join(info_path, "training_info.txt"), "w") as fh:
            fh.write("MuZero Training Info\n")
            fh.write(f"Environment: {getattr(env_config, 'type', None)}\nModel saved at: {stamp}\n\nHyperparameters:\n")
            for name in ("look_back", "roll_ahead", "mini_batch_size", "learning_rate", "betas"):
                fh.write(f"{name}: {getattr(self.config, name)}\n")
            fh.write(f"\nFinal Loss: {final_loss_val}\n\nNetwork configuration:\n\n")
            fh.write(f"{self.repr_net}\n\n{self.dyn_net}\n\n{self.pred_net}\n\n")
        print(f"Saved model to: {model_path}/")
        return self.repr_net, self.dyn_net, self.pred_net

    def load_model(self, counter: int) -> None:
        base = os.path.join(BASE_PATH, "models")
        found = sorted((m.group(2), name) for name in os.listdir(base)
                       if (m := FOLDER_REGEX.match(name)) and int(m.group(1)) == counter)
        if not found:
            raise FileNotFoundError(f"No folder found in '{base}' with counter = {counter}")
        full = os.path.join(base, found[-1][1])
        for net, name in ((self.repr_net, "repr"), (self.dyn_net, "dyn"), (self.pred_net, "pred")):
            net.load_state_dict(torch.load(os.path.join(full, f"{name}.pth"), weights_only=True))
        print(f"Loaded model from: {full}/")

    def _get_next_model_counter(self, base_path: str) -> int:
        top = -1
        for sub in ("models", "info"):
            path = os.path.join(base_path, sub)
            os.makedirs(path, exist_ok=True)
            for name in os.listdir(path):
                m = FOLDER_REGEX.match(name)
                if m:
                    top = max(top, int(m.group(1)))
        return top + 1
