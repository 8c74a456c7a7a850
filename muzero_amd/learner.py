"""Batched and data-parallel learner step.

The reference trains one window at a time at batch size 1 (src/training.py:131-172, bptt
:244-305) and sums the per-window losses before one optimizer step.  The same loss and the same gradient come
from ONE batched unroll over all windows: no layer of the MLP configurations mixes rows, and the
BatchNorm layers of residual configurations are given per-window statistics
(muzero_amd.neural_networks.per_call) -- which is what `batched_unroll_loss` computes;
`DataParallelLearner` then splits the windows over ranks (one process per GPU), all-reduces
the flat gradient once per optimizer step (RCCL over xGMI with the `nccl` backend, gloo on
the CPU) and keeps the actors' packed weights in step with `broadcast_weights`.

Loss definitions are the reference's (src/training.py:307-320): MSE for value and reward,
cross entropy with the already-softmaxed policy used as logits.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from muzero_amd.neural_networks.per_call import per_call_statistics
from muzero_amd.ring_buffer import Frame, FrameRingBuffer, make_history_tensor


@dataclass
class WindowBatch:
    """B training windows in tensor form (what R3's slicing produces, src/training.py:131-160)."""
    history: torch.Tensor     # f32[B, C*K+K, H, W]  input of h (past frames + action planes)
    actions: torch.Tensor     # i64[B, w+1]
    policies: torch.Tensor    # f32[B, w+1, A]
    z: torch.Tensor           # f32[B, w+1]   value targets
    rewards: torch.Tensor     # f32[B, w+1]
    is_weights: torch.Tensor  # f32[B]

    def to(self, device):
        return WindowBatch(*(t.to(device) for t in (self.history, self.actions, self.policies, self.z, self.rewards,
                                                    self.is_weights)))

    def rows(self, idx):
        return WindowBatch(self.history[idx], self.actions[idx], self.policies[idx], self.z[idx], self.rewards[idx],
                           self.is_weights[idx])

    def __len__(self):
        return int(self.history.shape[0])


def windows_to_batch(windows, z_targets: np.ndarray, rewards: np.ndarray, is_weights, history_length: int,
                     lookback: int, roll_ahead: int) -> WindowBatch:
    """`windows`: list of (episode, k).  History assembly follows bptt (src/training.py:252-262):
    the ring buffer is filled with the first past frame and the following ones are added."""
    hist, acts, pols = [], [], []
    w = roll_ahead
    for ep, k in windows:
        start = max(0, k - lookback)
        ring = FrameRingBuffer(size=history_length)
        ring.fill(Frame(state=ep.chunks[start].state, action=ep.chunks[start].best_action))
        for i in range(start + 1, k):
            ring.add(Frame(state=ep.chunks[i].state, action=ep.chunks[i].best_action))
        hist.append(make_history_tensor(ring))
        acts.append([int(ep.chunks[i].best_action) for i in range(k, k + w + 1)])
        pols.append(torch.stack([torch.as_tensor(ep.chunks[i].policy, dtype=torch.float32) for i in range(k, k + w + 1)]))
    return WindowBatch(torch.cat(hist, 0), torch.tensor(acts, dtype=torch.long), torch.stack(pols),
                       torch.as_tensor(np.asarray(z_targets), dtype=torch.float32),
                       torch.as_tensor(np.asarray(rewards), dtype=torch.float32),
                       torch.as_tensor(np.asarray(is_weights, dtype=np.float64), dtype=torch.float32))


def batched_unroll_loss(repr_net, dyn_net, pred_net, batch: WindowBatch, roll_ahead: int, coeff=(1.0, 1.0, 1.0), latent=None):
    """Returns (weighted_sum, per_window_loss[B], (policy_sum, value_sum, reward_sum)).

    weighted_sum = sum_b is_weights[b] * loss_b, NOT yet divided by sum(is_weights): the caller
    normalises (a data-parallel caller after summing over ranks).  Networks with BatchNorm layers
    are evaluated with per-window statistics: the reference unrolls one window per call.
    `latent`: the root latents when the caller has evaluated repr_net itself (it is not called then)."""
    cp, cv, cr = coeff
    with per_call_statistics(repr_net), per_call_statistics(dyn_net), per_call_statistics(pred_net):
        return _unroll(repr_net, dyn_net, pred_net, batch, roll_ahead, cp, cv, cr, latent)


def representation_latents(repr_net, batch: WindowBatch):
    """h on the windows' first histories, one window per call's statistics (src/training.py:252-256)."""
    with per_call_statistics(repr_net):
        return repr_net(batch.history)


class _FusedUnrollLoss(torch.autograd.Function):
    """The three losses over all unroll steps and windows in one launch (mz_unroll_loss); the kernel
    also leaves d(weighted sum)/d(prediction), so backward is a scaling."""

    @staticmethod
    def forward(ctx, pred_p, pred_v, pred_r, target_p, z, r, is_weights, cp, cv, cr):
        from muzero_amd import _hip
        from muzero_amd._hip import ptr
        W1, B, A = pred_p.shape
        w = W1 - 1
        dev = pred_p.device
        pred_p, pred_v, pred_r = pred_p.contiguous(), pred_v.contiguous(), pred_r.contiguous()
        target_p, z, r, is_weights = (t.to(torch.float32).contiguous() for t in (target_p, z, r, is_weights))
        per_window = torch.empty(B, dtype=torch.float32, device=dev)
        parts = torch.empty((3, B), dtype=torch.float32, device=dev)
        sums = torch.empty(4, dtype=torch.float32, device=dev)
        g_p, g_v, g_r = torch.empty_like(pred_p), torch.empty_like(pred_v), torch.empty_like(pred_r)
        _hip.check(_hip.lib().mz_unroll_loss(ptr(pred_p), ptr(pred_v), ptr(pred_r) if w > 0 else None, ptr(target_p), ptr(z),
                                             ptr(r), ptr(is_weights), B, w, A, float(cp), float(cv), float(cr),
                                             ptr(per_window), ptr(parts), ptr(sums), ptr(g_p), ptr(g_v),
                                             ptr(g_r) if w > 0 else None, torch.cuda.current_stream().cuda_stream),
                   "mz_unroll_loss")
        ctx.save_for_backward(g_p, g_v, g_r)
        ctx.mark_non_differentiable(per_window, sums)
        return sums[0], per_window, sums

    @staticmethod
    def backward(ctx, g_weighted, _g_per_window, _g_sums):
        g_p, g_v, g_r = ctx.saved_tensors
        return g_p * g_weighted, g_v * g_weighted, g_r * g_weighted, None, None, None, None, None, None, None


def _unroll(repr_net, dyn_net, pred_net, batch: WindowBatch, roll_ahead: int, cp: float, cv: float, cr: float, latent=None):
    if latent is None:
        latent = repr_net(batch.history)
    B = len(batch)
    if latent.is_cuda:
        # the networks step by step (torch / rocBLAS, autograd), the losses of all steps in one kernel
        pp, pv, pr = [], [], []
        for i in range(roll_ahead):
            pred_p, pred_v = pred_net(latent)
            latent, pred_r = dyn_net(latent, batch.actions[:, i])
            pp.append(pred_p), pv.append(pred_v[:, 0]), pr.append(pred_r[:, 0])
        final_p, final_v = pred_net(latent)
        pp.append(final_p), pv.append(final_v[:, 0])
        pr_t = torch.stack(pr) if pr else torch.zeros((0, B), dtype=torch.float32, device=latent.device)
        weighted, per_window, sums = _FusedUnrollLoss.apply(torch.stack(pp), torch.stack(pv), pr_t, batch.policies, batch.z,
                                                            batch.rewards, batch.is_weights, cp, cv, cr)
        return weighted, per_window, (sums[1], sums[2], sums[3])
    lp = torch.zeros(B, dtype=torch.float32, device=latent.device)
    lv = torch.zeros_like(lp)
    lr = torch.zeros_like(lp)
    for i in range(roll_ahead):
        pred_p, pred_v = pred_net(latent)
        latent, pred_r = dyn_net(latent, batch.actions[:, i])
        lp = lp + F.cross_entropy(pred_p, batch.policies[:, i], reduction="none") * cp
        lv = lv + (pred_v[:, 0] - batch.z[:, i]) ** 2 * cv
        lr = lr + (pred_r[:, 0] - batch.rewards[:, i]) ** 2 * cr
    final_p, final_v = pred_net(latent)
    lv = lv + (final_v[:, 0] - batch.z[:, roll_ahead]) ** 2 * cv
    lp = lp + F.cross_entropy(final_p, batch.policies[:, roll_ahead], reduction="none") * cp
    per_window = lp + lv + lr
    return (batch.is_weights * per_window).sum(), per_window, (lp.sum(), lv.sum(), lr.sum())


def tune_small_gemms(results_file: str | None = None, max_ms: int = 30, max_iterations: int = 20) -> None:
    """Lets torch time the rocBLAS / hipBLASLt solutions for every matrix product it meets and keep
    the fastest (torch.cuda.tunable).  The libraries' default choice for the learner's products on
    the small networks (a few hundred rows, 128-336 wide) is one 256 x 256 tile, i.e. ONE workgroup
    per product, 30-70 us each; the tuned choices take 10-16 us.  The first step with a new shape
    pays for the timing (seconds); `results_file` keeps the choices between runs (they are tied to
    the library versions named in the file and ignored otherwise)."""
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.set_max_tuning_duration(max_ms)
    tunable.set_max_tuning_iterations(max_iterations)
    if results_file is not None:
        tunable.set_filename(results_file)


class DataParallelLearner:
    """One optimizer step over a batch of windows split across ranks.

    Every rank holds identical weights and an identical optimizer; rank r takes the windows
    r, r + world, ...; gradients of the weighted loss SUM and the weight total are all-reduced
    in one flat buffer (a single collective of n_params + 1 floats, ~1 MB for c4_mlp: one
    message per step suits point-to-point xGMI links better than per-tensor calls)."""

    def __init__(self, repr_net, dyn_net, pred_net, optimizer, roll_ahead: int, coeff=(1.0, 1.0, 1.0), dist=None):
        self.nets = (repr_net, dyn_net, pred_net)
        self.params = [p for n in self.nets for p in n.parameters()]
        self.optimizer, self.roll_ahead, self.coeff, self.dist = optimizer, roll_ahead, coeff, dist
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    def step(self, batch: WindowBatch):
        """`batch`: the FULL batch (every rank passes the same one; each uses its share).
        Returns (normalised loss, per-window |loss| of this rank's windows, their positions)."""
        mine = torch.arange(self.rank, len(batch), self.world)
        loss, per_window = self.step_local(batch.rows(mine) if len(mine) > 0 else None)
        return loss, per_window, mine

    def _buckets(self):
        """One flat gradient buffer per network (the order backward finishes them in: f and g, whose
        first use is the first unroll step, then h), every parameter's .grad a view into it: no copy
        into a flat buffer before the collective, none back after it."""
        if getattr(self, "_bucket_list", None) is None:
            self._bucket_list = []
            for net in (self.nets[2], self.nets[1], self.nets[0]):
                ps = [p for p in net.parameters() if p.requires_grad]
                if not ps:
                    continue
                flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
                off = 0
                for p in ps:
                    p.grad = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
                self._bucket_list.append((flat, ps))
        return self._bucket_list

    def step_local(self, local: WindowBatch | None):
        """One optimizer step where every rank brings its OWN windows (e.g. cut from its own
        episode store): the gradient of the weighted loss sum and the weight total are summed
        over ranks, so the result is the step a single rank would take on the union of the
        windows.  Returns (normalised loss over all ranks, per-window |loss| of the local ones).

        Communication overlaps the backward pass: the [weight total, weighted loss] pair is reduced
        while backward runs, and each network's flat gradient bucket starts its all-reduce from the
        hook of the last parameter whose gradient arrives (src/training.py:174-191 does backward and
        step on one device; here the collective of a bucket runs beside the rest of backward)."""
        dev = self.params[0].device
        buckets = self._buckets()
        for flat, ps in buckets:
            flat.zero_()
            off = 0
            for p in ps:  # an optimizer or a caller may have replaced .grad: point it at the bucket again
                if p.grad is None or p.grad.data_ptr() != flat[off:off + 1].data_ptr():
                    p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        flat_extra = torch.zeros(2, dtype=torch.float32, device=dev)  # [sum of weights, weighted loss sum]
        per_window = torch.zeros(0)
        works, handles = [], []
        multi = self.world > 1
        if local is not None and len(local) > 0:
            local = local.to(dev)
            wsum, per_window, _ = batched_unroll_loss(*self.nets, local, self.roll_ahead, self.coeff)
            flat_extra[0] = local.is_weights.sum()
            flat_extra[1] = wsum.detach()
            ready = [False] * len(buckets)
            nxt = [0]  # buckets are reduced in list order on every rank, whatever order backward finishes them in

            def launch_ready():
                while nxt[0] < len(buckets) and ready[nxt[0]]:
                    works.append(self.dist.all_reduce(buckets[nxt[0]][0], op=self.dist.ReduceOp.SUM, async_op=True))
                    nxt[0] += 1
            if multi:
                works.append(self.dist.all_reduce(flat_extra, op=self.dist.ReduceOp.SUM, async_op=True))
                for bi, (flat, ps) in enumerate(buckets):
                    left = [len(ps)]

                    def arrived(_p, bi=bi, left=left):
                        left[0] -= 1
                        if left[0] == 0:
                            ready[bi] = True
                            launch_ready()
                    handles += [p.register_post_accumulate_grad_hook(arrived) for p in ps]
            wsum.backward()
            for hnd in handles:
                hnd.remove()
            if multi:  # a parameter no window reached never fires its hook: reduce what is left, in order
                ready = [True] * len(buckets)
                launch_ready()
        elif multi:  # a rank without windows still takes part in every collective
            works.append(self.dist.all_reduce(flat_extra, op=self.dist.ReduceOp.SUM, async_op=True))
            for flat, _ in buckets:
                works.append(self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, async_op=True))
        for wk in works:
            wk.wait()
        norm = flat_extra[0]
        if float(norm) <= 0.0:
            # no usable window on any rank (every draw was an episode too short for a window): the
            # reference skips the optimizer step then (`if ep_indices_used:`, src/training.py:174);
            # dividing by the zero weight total would write NaN into every weight
            for flat, _ in buckets:
                flat.zero_()
            return 0.0, per_window.detach().abs().cpu()
        for flat, _ in buckets:
            flat.div_(norm)
        self.optimizer.step()
        return float(flat_extra[1] / norm), per_window.detach().abs().cpu()

    # ------------------------------------------------------------------ recorded step
    def step_local_recorded(self, local: WindowBatch):
        """step_local replayed from HIP graphs, in three segments: (1) h, the unroll and the backward pass
        through the unroll -- the gradients of f and g are complete there; (2) the backward pass through h;
        (3) the division by the weight total and the optimizer step.  With several ranks the all-reduce of
        the f / g gradients (with the weight total and the loss sum) is issued between (1) and (2) without
        waiting for it, so it runs beside h's backward pass -- the bulk of the step for a convolutional h --
        as the bucket hooks of step_local arrange it in the eager path (src/training.py:174-191 does backward
        and step on one device); the all-reduce of h's gradients follows (2), and (3) waits for both on the
        stream only.  On the small networks a step is a few hundred launches of microsecond kernels, so
        replaying it costs a fraction of issuing it.

        Needs a fixed batch shape (DeviceEpisodeStore's samplers give one) and an optimizer whose
        step can be recorded (torch.optim.Adam/AdamW(..., capturable=True), plain SGD; a learning
        rate that changes must be a tensor).  The first call records (after one trial step whose
        effect on the weights and the optimizer state is undone).  Returns DEVICE tensors
        (normalised loss over all ranks, per-window |loss| of the local windows): nothing here
        waits for the GPU (with the RCCL backend; gloo's collectives wait on the host).

        A batch whose weight total is zero on every rank (only zero-weight windows) cannot skip the
        recorded optimizer step; its gradient is forced to exactly zero instead of 0/0, and the
        weights are put back to their values from before the step, so momentum-free and
        momentum-carrying optimizers alike leave the weights unchanged (the optimizer's own step
        counter still advances).  DeviceEpisodeStore.sample_prioritized never produces such a
        batch unless no stored episode is long enough, which it reports as an error."""
        dev = self.params[0].device
        fields = ("history", "actions", "policies", "z", "rewards", "is_weights")
        shapes = tuple(tuple(getattr(local, f).shape) for f in fields)
        if getattr(self, "_rec_shapes", None) != shapes:
            self._record(local.to(dev), shapes)
        for f in fields:
            getattr(self._rec_batch, f).copy_(getattr(local, f), non_blocking=True)
        self._rec_backward.replay()
        waits = []
        if self.world > 1:  # f / g gradients + [weight total, loss sum]: on its way while h's backward pass runs
            waits.append(self._all_reduce_async(0, self._rec_split))
        self._rec_backward_h.replay()
        if self.world > 1:
            if self._rec_split < self._rec_flat.numel():
                waits.append(self._all_reduce_async(self._rec_split, self._rec_flat.numel()))
            for wait in waits:
                wait()
        self._rec_update.replay()
        return self._rec_loss, self._rec_per_window

    def _all_reduce_async(self, lo: int, hi: int):
        """Starts the sum over ranks of _rec_flat[lo:hi]; returns the call that makes the current stream wait for it.
        RCCL reduces the device buffer in place on its own stream.  gloo (several ranks rehearsing on one card; its
        device-tensor collectives do not complete on this platform) gets a pinned host copy: the host waits for the
        segment that produced the slice, the reduction then runs on gloo's thread beside the next segment."""
        part = self._rec_flat[lo:hi]
        if self.dist.get_backend() != "gloo":
            return self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM, async_op=True).wait
        if getattr(self, "_rec_host", None) is None or self._rec_host.numel() != self._rec_flat.numel():
            self._rec_host = torch.empty(self._rec_flat.numel(), dtype=self._rec_flat.dtype, pin_memory=True)
        host = self._rec_host[lo:hi]
        host.copy_(part, non_blocking=True)
        torch.cuda.current_stream(part.device).synchronize()
        work = self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, async_op=True)

        def wait():
            work.wait()
            part.copy_(host, non_blocking=True)
        return wait

    def _point_grads_at_flat(self):
        """Every parameter's .grad is a view into the flat buffer: [f, g parameters | weight total, loss sum | h
        parameters] -- no copy into a flat buffer before a collective, none back after it."""
        off = 0
        for group in (self._rec_fg, None, self._rec_h):
            if group is None:
                off += 2
                continue
            for p in group:
                n = p.numel()
                if p.grad is None or p.grad.data_ptr() != self._rec_flat[off:off + 1].data_ptr():
                    p.grad = self._rec_flat[off:off + n].view_as(p)
                off += n

    def _backward_unroll(self):
        self._rec_flat.zero_()
        lat0 = representation_latents(self.nets[0], self._rec_batch)
        self._rec_lat0 = lat0
        lat = lat0.detach().requires_grad_(lat0.requires_grad)
        wsum, per_window, _ = batched_unroll_loss(*self.nets, self._rec_batch, self.roll_ahead, self.coeff, latent=lat)
        wsum.backward()
        self._rec_lat_grad = lat.grad
        n = self._rec_split - 2
        self._rec_flat[n] = self._rec_batch.is_weights.sum()
        self._rec_flat[n + 1] = wsum.detach()
        self._rec_per_window.copy_(per_window.detach().abs())

    def _backward_h(self):
        if self._rec_lat_grad is not None:
            self._rec_lat0.backward(self._rec_lat_grad)
        self._rec_lat0 = self._rec_lat_grad = None

    def _update_from_flat(self):
        flat = self._rec_flat
        n = self._rec_split - 2
        norm = flat[n]
        ok = norm > 0
        safe, live = torch.where(ok, norm, torch.ones_like(norm)), ok.to(flat.dtype)  # x / safe * live: exact when ok
        self._rec_loss.copy_(flat[n + 1] / safe * live)
        flat[:n].copy_(flat[:n] / safe * live)
        flat[n + 2:].copy_(flat[n + 2:] / safe * live)
        torch._foreach_copy_(self._rec_before, [p.data for p in self.params])
        self.optimizer.step()
        # weight total zero: the step above ran on a zero gradient; undo whatever it still moved
        # (momentum, weight decay) -- the reference takes no step at all for such a batch
        with torch.no_grad():
            for p, b in zip(self.params, self._rec_before):
                p.copy_(torch.where(ok, p, b))

    def _record(self, local: WindowBatch, shapes):
        import copy
        dev = self.params[0].device
        self._rec_batch = WindowBatch(*(t.clone() for t in (local.history, local.actions, local.policies, local.z,
                                                            local.rewards, local.is_weights)))
        self._rec_fg = [p for net in (self.nets[2], self.nets[1]) for p in net.parameters()]
        self._rec_h = list(self.nets[0].parameters())
        n = sum(p.numel() for p in self.params)
        self._rec_split = sum(p.numel() for p in self._rec_fg) + 2
        self._rec_flat = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self._rec_per_window = torch.zeros(len(local), dtype=torch.float32, device=dev)
        self._rec_loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._rec_before = [torch.empty_like(p.data) for p in self.params]
        self._bucket_list = None  # the eager path's buckets no longer own the .grad views
        for p in self.params:
            p.grad = None
        self._point_grads_at_flat()
        # one trial step on a side stream (creates the optimizer state and every workspace), then
        # the weights and the optimizer state go back to what they were
        weights = [p.detach().clone() for p in self.params]
        had_state = len(self.optimizer.state) > 0
        saved = copy.deepcopy(self.optimizer.state_dict()) if had_state else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._backward_unroll()
            self._backward_h()
            self._update_from_flat()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, w in zip(self.params, weights):
                p.copy_(w)
            if had_state:
                fresh = self.optimizer.state_dict()["state"]
                for key, st in saved["state"].items():
                    for name, v in st.items():
                        if torch.is_tensor(v):
                            fresh[key][name].copy_(v)
            else:
                for st in self.optimizer.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        self._point_grads_at_flat()
        torch.cuda.synchronize(dev)
        self._rec_backward, self._rec_backward_h, self._rec_update = (torch.cuda.CUDAGraph() for _ in range(3))
        with torch.cuda.graph(self._rec_backward):
            self._backward_unroll()
        # the second segment runs backward through what the first one's forward pass saved: same memory pool,
        # replayed in the order recorded
        with torch.cuda.graph(self._rec_backward_h, pool=self._rec_backward.pool()):
            self._backward_h()
        with torch.cuda.graph(self._rec_update, pool=self._rec_backward.pool()):
            self._update_from_flat()
        self._rec_shapes = shapes

    def broadcast_weights(self, src: int = 0):
        """Learner -> actors: one flat broadcast; actors re-pack their MFMA weight blob afterwards
        (PackedNetworks.refresh)."""
        if self.world == 1:
            return
        flat = torch.cat([p.data.reshape(-1) for p in self.params])
        self.dist.broadcast(flat, src=src)
        off = 0
        for p in self.params:
            n = p.numel()
            p.data.copy_(flat[off:off + n].view_as(p))
            off += n
