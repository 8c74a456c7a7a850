"""Batched g / f for residual-tower configurations (`res_net` non-empty) on the HIP kernels.

`TowerNetworks` offers the same `dynamics(...)` / `prediction(...)` calls as
`muzero_amd.engine.PackedNetworks`, so `BatchedMCTS` drives it in lock step (one launch per
stage per simulation for all G trees).  What the kernels compute follows
src/neural_networks/neural_network.py:152-188 (g) and :234-262 (f):

* fc_merge is split as for the MLP networks: `W[:, :D] @ latent` once per distinct latent
  (`mz_wide_forward`, fp32 MFMA, reduction dimension streamed through LDS) plus a per-action
  table `W[:, D:] @ E[a] + b` (`mz_action_table`) added while the tower loads its input;
* the ResBlocks and res_final run in `mz_tower_forward` with the batch statistics of ONE
  reference call per group of rows: the reference never leaves train mode, so expansion
  normalises over the A children of a node together (src/search/expansion.py:25-28) and every
  other call over a single latent;
* heads: the first Linear of a head reads D inputs (`mz_wide_forward` when D exceeds the
  LDS-resident MLP kernels' width, else the whole head goes through `mz_mlp_forward`).

Bottleneck form.  In every shipped configuration the towers end in a narrow block (3 channels)
followed by `res_final`, a 1x1 convolution back to 256 channels: a latent produced by g is an
affine image `F t + b` of the block output t (3 x 6 x 6 = 108 values).  Whatever consumes a latent
linearly first -- fc_merge's latent half, the first Linear of a head -- is therefore evaluated on
t with weights multiplied by F once per weight update (`W F`, `W b` folded into the bias): the
9216 x 9216 product per row becomes 9216 x 108.  Exact up to float32 rounding.  The search keeps t
for every non-root node next to its latent (`bottleneck_g` values per node); callers that have
no t (the root, the drop-in `expand_node`) use the plain form.

There is no CPU fallback: every call goes through libmuzero_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from muzero_amd import _hip
from muzero_amd._hip import ptr
from muzero_amd.neural_networks.network_builder import ResBlock

MAX_BLOCKS = _hip.MAX_BLOCKS
CTower = _hip.Tower

_ACT_OF = {torch.nn.Identity: "none", torch.nn.ReLU: "relu", torch.nn.Tanh: "tanh", torch.nn.Sigmoid: "sigmoid",
           torch.nn.Softmax: "softmax"}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Head:
    """One MLP head: optional wide first layer + the rest through the LDS-resident MLP kernel."""

    def __init__(self):
        self.wide: _hip.PackedLayer | None = None
        self.rest: _hip.PackedMlp | None = None
        self.extra_softmax = 0   # the network's own softmax when no kernel layer applies it
        self.out_features = 0
        self.mid_features = 0


class TowerNetworks:
    """Device copy of g / f with residual towers, evaluated by the HIP kernels.

    `wide_above`: heads whose input is wider than this use the chunked wide-layer kernel for
    their first layer (the LDS-resident MLP kernels hold rows of at most 4096 floats)."""

    fusable = False     # no single-launch search for these networks: BatchedMCTS runs in lock step
    concurrent = True   # f and the fc_merge product of g may run on two streams (scratch is per stream)
    strided_outputs = True  # reward / value may be 1-D views with a stride (a column of a [G, n] buffer)

    def __init__(self, dynamics: torch.nn.Module, prediction: torch.nn.Module, device="cuda", wide_above: int = 4096,
                 track_running_stats: bool = False, fuse_merge_conv: bool = True, root_final: torch.nn.Module | None = None):
        """track_running_stats: every tower call also advances the BatchNorm running statistics of its
        blocks as the reference's train-mode calls do (network_builder.py:94,104,118), one update per
        reference call, the calls of one launch in row order (with one tree that is the reference's
        order; with G trees in lock step the trees' calls interleave).  `sync_running_stats()` writes
        the buffers and num_batches_tracked back into the modules.  No output depends on them."""
        self.device = torch.device(device)
        # root_final: the representation network's final 1x1 convolution (neural_network.py:51,60-61) when it widens a
        # few channels to the latent's C: a root latent is then the same kind of affine image, final_conv(pool(x)), of
        # a few values as g's outputs are, and the root expansion can run in the bottleneck form too (`root_form`)
        self.root_final = root_final
        self.fuse_merge_conv = fuse_merge_conv  # False keeps fc_merge and the tower of g as separate kernels
        # the fused pass on operands split into two fp16 terms each: float32 accuracy at the fp16 matrix rate
        # (MZ_MC_SPLIT=0: float32 matrix instructions)
        self.split_merge_conv = os.environ.get("MZ_MC_SPLIT", "1") != "0"
        # source rows from which the fused form is used (default: always; 256 trees 1.33 -> 1.41 M sims/s, the
        # expansion's separate kernels take 57 us there, the fused pair 43 us)
        self.fuse_min_rows = int(os.environ.get("MZ_FUSE_MIN_ROWS", "1"))
        self.track_running_stats = track_running_stats
        self._buffers_seen: dict = {}  # id(bn module) -> its running buffers as last read or written here
        self._running: dict = {}   # id(tower struct) -> [(bn module, running_mean, running_var)] per block x 3
        self._calls: dict = {}     # id(tower struct) -> device counter of reference calls since the last sync
        self._mean_shift: dict = {}  # id(tower struct) -> {layer: constant missing from the exported batch mean}
        self.dynamics_net, self.prediction_net = dynamics, prediction
        c, h, w = dynamics.latent_shape
        self.shape = (c, h, w)
        self.D, self.A = c * h * w, dynamics.num_actions
        self.wide_above = wide_above
        self._keep: list[torch.Tensor] = []
        self._bufs: dict = {}
        self.blob = None
        self.refresh()

    # ---- packing ----------------------------------------------------------------------------
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _tower(self, blocks, res_final) -> CTower:
        c, h, w = self.shape
        t = CTower()
        t.n_blocks, t.C, t.H, t.W = len(blocks), c, h, w
        if len(blocks) > MAX_BLOCKS:
            raise NotImplementedError(f"at most {MAX_BLOCKS} residual blocks per tower")
        for i, blk in enumerate(blocks):
            assert isinstance(blk, ResBlock)
            k = blk.conv1.kernel_size[0]
            if (blk.conv1.kernel_size != (k, k) or blk.conv1.stride != (1, 1) or
                    blk.conv1.padding != ((k - 1) // 2, (k - 1) // 2) or k % 2 == 0 or
                    not isinstance(blk.pool, torch.nn.Identity)):
                # anything else changes H x W, which neural_network.py:185,254 cannot flatten into the heads
                raise NotImplementedError("g/f residual blocks must keep the latent's H x W "
                                          "(stride 1, padding (k-1)/2, no pooling)")
            b = t.block[i]
            b.conv1_w, b.bn1_w, b.bn1_b = (ptr(self._dev(x)) for x in (blk.conv1.weight, blk.bn1.weight, blk.bn1.bias))
            b.conv2_w, b.bn2_w, b.bn2_b = (ptr(self._dev(x)) for x in (blk.conv2.weight, blk.bn2.weight, blk.bn2.bias))
            if blk.downsample is not None:
                conv, bn = blk.downsample[0], blk.downsample[1]
                b.down_w, b.dbn_w, b.dbn_b = (ptr(self._dev(x)) for x in (conv.weight, bn.weight, bn.bias))
                b.epsd = bn.eps
            b.in_channels, b.out_channels, b.kernel_size = blk.conv1.in_channels, blk.conv1.out_channels, k
            b.activation = _hip.ACT[_ACT_OF[type(blk.activation)]]
            b.eps1, b.eps2 = blk.bn1.eps, blk.bn2.eps
        if not isinstance(res_final, torch.nn.Identity):
            t.final_w = ptr(self._dev(res_final.weight.reshape(res_final.out_channels, res_final.in_channels)))
            t.final_b = ptr(self._dev(res_final.bias)) if res_final.bias is not None else None
        return t

    def _layers(self, seq: torch.nn.Sequential):
        mods = list(seq)
        out = []
        for i in range(0, len(mods), 2):
            lin = mods[i]
            assert isinstance(lin, torch.nn.Linear), f"unexpected module {type(lin)} in MLP"
            act = mods[i + 1] if i + 1 < len(mods) else torch.nn.Identity()
            out.append((lin, _ACT_OF[type(act)]))
        return out

    def _dense(self, lin: torch.nn.Linear, act: str, ldw: int = 0, cols=None) -> _hip.Dense:
        w = self._dev(lin.weight)
        b = self._dev(lin.bias) if lin.bias is not None else None
        return _hip.Dense(ptr(w), ptr(b), cols or lin.in_features, lin.out_features, _hip.ACT[act], ldw)

    def refresh(self) -> None:
        """Re-reads the modules' weights (call after a weight update)."""
        if self.track_running_stats and self._running:
            # the modules' buffers are the one copy self-play and training share: hand ours over, re-read theirs
            self.sync_running_stats()
            self._running, self._calls, self._mean_shift = {}, {}, {}
        self._keep = []
        self.version = getattr(self, "version", 0) + 1  # captured launch graphs hold the old buffers' addresses
        # what this read of the final convolution saw: a caller checks it to learn whether the composed root weights are stale
        self.root_seen = None if self.root_final is None else \
            tuple((t._version, t.data_ptr()) for t in self.root_final.parameters())
        lib = _hip.lib()
        g, f = self.dynamics_net, self.prediction_net
        D, A = self.D, self.A
        self.tower_g = self._tower(list(g.res_blocks), g.res_final)
        self.tower_f = self._tower(list(f.res_blocks), f.res_final)
        self._track(self.tower_g, list(g.res_blocks))
        self._track(self.tower_f, list(f.res_blocks))
        # plan every packed piece into one blob
        plans = []  # (kind, dense-or-mlp, plan)
        off = 0

        def plan_wide(dense):
            nonlocal off
            pl = _hip.PackedLayer()
            n = lib.mz_wide_plan(C.byref(dense), off, C.byref(pl))
            if n < 0:
                _hip.check(int(n), "mz_wide_plan")
            off += n
            plans.append(("wide", dense, pl))
            return pl

        def plan_mlp(layers):
            nonlocal off
            m = _hip.Mlp()
            m.n_layers = len(layers)
            for i, entry in enumerate(layers):
                m.layer[i] = entry if isinstance(entry, _hip.Dense) else self._dense(*entry)
            pm = _hip.PackedMlp()
            n = lib.mz_mlp_plan(C.byref(m), off, C.byref(pm))
            if n < 0:
                _hip.check(int(n), "mz_mlp_plan")
            off += n
            plans.append(("mlp", m, pm))
            return pm

        def plan_head(seq):
            layers = self._layers(seq)
            hd = _Head()
            hd.out_features = layers[-1][0].out_features
            if D > self.wide_above:
                lin, act = layers[0]
                if act == "softmax":   # a one-layer policy head: normalise afterwards
                    act, hd.extra_softmax = "none", 1
                hd.wide = plan_wide(self._dense(lin, act))
                hd.mid_features = lin.out_features
                layers = layers[1:]
            if layers:
                hd.rest = plan_mlp(layers)
            return hd

        merge = _hip.Dense(ptr(self._dev(g.fc_merge.weight)), None, D, D, _hip.ACT["none"], 2 * D)
        self.merge = plan_wide(merge)
        self.reward_head = plan_head(g.reward_mlp)
        self.value_head = plan_head(f.value_mlp)
        self.policy_head = plan_head(f.policy_mlp)
        # ---- bottleneck form: consumers of a latent that is res_final(t), evaluated on t
        def bottleneck(net):
            """(Cl, Wf [C, Cl] f64, bf [C] f64) when the tower ends in a narrowing res_final, else None."""
            fin = net.res_final
            if isinstance(fin, torch.nn.Identity) or len(net.res_blocks) == 0:
                return None
            Cl = fin.in_channels
            dt = Cl * self.shape[1] * self.shape[2]
            if dt >= D or dt > 4096:
                return None
            wf = fin.weight.detach().to(self.device, torch.float64).reshape(fin.out_channels, Cl)
            bf = (fin.bias.detach().to(self.device, torch.float64) if fin.bias is not None
                  else torch.zeros(fin.out_channels, dtype=torch.float64, device=self.device))
            return Cl, wf, bf

        def through(w: torch.Tensor, bias, bn):
            """Weights [N, D] and bias of a Linear on res_final(t) -> weights [N, Cl*HW] and bias on t."""
            Cl, wf, bf = bn
            hw = self.shape[1] * self.shape[2]
            w3 = w.detach().to(self.device, torch.float64).reshape(w.shape[0], self.shape[0], hw)
            wt = torch.einsum("ncp,cj->njp", w3, wf).reshape(w.shape[0], Cl * hw)
            bt = torch.einsum("ncp,c->n", w3, bf)
            if bias is not None:
                bt = bt + bias.detach().to(self.device, torch.float64)
            return self._dev(wt.to(torch.float32)), self._dev(bt.to(torch.float32))

        def head_on_t(seq, bn):
            layers = self._layers(seq)
            lin, act = layers[0]
            wt, bt = through(lin.weight, lin.bias, bn)
            first = _hip.Dense(ptr(wt), ptr(bt), wt.shape[1], wt.shape[0], _hip.ACT[act], 0)
            return plan_mlp([first] + layers[1:])

        bn_g, bn_f = bottleneck(g), bottleneck(f)
        self.bottleneck_g = bn_g[0] * self.shape[1] * self.shape[2] if bn_g else 0
        self.bottleneck_f = bn_f[0] * self.shape[1] * self.shape[2] if bn_f else 0
        self.merge_t = self.reward_t = self.value_t = self.policy_t = None
        merge_const = None
        self.fused_g = False     # fc_merge + the tower's first convolution in one pass (mz_merge_conv_forward)
        self.merge_t_perm = self.conv_g = self.table_t_perm = None
        if bn_g:
            wt, merge_const = through(g.fc_merge.weight[:, :D], None, bn_g)
            self.merge_t = plan_wide(_hip.Dense(ptr(wt), None, wt.shape[1], D, _hip.ACT["none"], 0))
            self.reward_t = head_on_t(g.reward_mlp, bn_g)
            blk = g.res_blocks[0] if len(g.res_blocks) == 1 else None
            c, hgt, wid = self.shape
            if (self.fuse_merge_conv and blk is not None and c == 256 and blk.downsample is not None and blk.conv1.kernel_size == (3, 3)
                    and blk.conv1.out_channels <= 3 and wt.shape[1] <= 128):
                co = blk.conv1.out_channels
                hw = hgt * wid
                # rows of the merge product position-major: row p * C + ch <- latent index ch * HW + p
                wperm = self._dev(wt.view(c, hw, wt.shape[1]).permute(1, 0, 2).reshape(D, wt.shape[1]))
                self.merge_t_perm = plan_wide(_hip.Dense(ptr(wperm), None, wt.shape[1], D, _hip.ACT["none"], 0))
                # columns of the first block's product: (output channel, tap) pairs of conv1, then the skip convolution
                wg = torch.zeros((32, c), dtype=torch.float32, device=self.device)
                wg[:co * 9] = blk.conv1.weight.detach().to(self.device, torch.float32).permute(0, 2, 3, 1).reshape(co * 9, c)
                wg[co * 9:co * 10] = blk.downsample[0].weight.detach().to(self.device, torch.float32).reshape(co, c)
                self.conv_g = plan_wide(_hip.Dense(ptr(self._dev(wg)), None, c, 32, _hip.ACT["none"], 0))
                self.fused_g = True
        if bn_f:
            self.value_t = head_on_t(f.value_mlp, bn_f)
            self.policy_t = head_on_t(f.policy_mlp, bn_f)
        # f's first block read through g's res_final: conv1 and the 1x1 skip convolution of a latent
        # F t + b are convolutions of t with weights pre-multiplied by F.  The folded bias of conv1
        # depends on the position (taps outside the image contribute nothing) -> conv1_bias map; the
        # constant the skip convolution picks up cancels in its BatchNorm.
        self.tower_f_t = self.tower_f_root = None
        self.merge_root_perm = self.table_root_perm = None
        blk0 = f.res_blocks[0] if len(f.res_blocks) else None

        def f_tower_on(bn):
            Cl, wf, bf = bn
            c, hgt, wid = self.shape
            wc = blk0.conv1.weight.detach().to(self.device, torch.float64)          # [o, C, k, k]
            w1 = torch.einsum("ocyx,cj->ojyx", wc, wf).contiguous()                 # [o, Cl, k, k]
            pad = blk0.conv1.padding[0]
            bias_map = torch.nn.functional.conv2d(bf.view(1, c, 1, 1).expand(1, c, hgt, wid), wc, padding=pad)[0]
            wd = blk0.downsample[0].weight.detach().to(self.device, torch.float64).reshape(-1, c) @ wf  # [o, Cl]
            t = _hip.Tower.from_buffer_copy(self.tower_f)
            t.C = Cl
            b0 = t.block[0]
            b0.in_channels = Cl
            b0.conv1_w = ptr(self._dev(w1.to(torch.float32)))
            b0.conv1_bias = ptr(self._dev(bias_map.to(torch.float32)))
            b0.down_w = ptr(self._dev(wd.to(torch.float32)))
            self._running[id(t)] = self._running.get(id(self.tower_f))  # the same blocks, pre-multiplied weights
            self._calls[id(t)] = self._calls.get(id(self.tower_f))
            # ... but it is part of that layer's batch mean (layer 2 of block 0: the skip path's BatchNorm)
            wdf = blk0.downsample[0].weight.detach().to(self.device, torch.float64).reshape(-1, c)
            self._mean_shift[id(t)] = {2: (wdf @ bf).to(torch.float32).contiguous()}
            return t

        if bn_g and bn_f and blk0 is not None and blk0.downsample is not None:
            self.tower_f_t = f_tower_on(bn_g)
        # the same for a root latent, which is h's final 1x1 convolution of a few pooled channels
        bn_h = None
        fin = self.root_final
        if (isinstance(fin, torch.nn.Conv2d) and fin.kernel_size == (1, 1) and fin.out_channels == self.shape[0]
                and fin.in_channels * self.shape[1] * self.shape[2] == self.bottleneck_g):
            bn_h = (fin.in_channels, fin.weight.detach().to(self.device, torch.float64).reshape(fin.out_channels, fin.in_channels),
                    fin.bias.detach().to(self.device, torch.float64) if fin.bias is not None
                    else torch.zeros(fin.out_channels, dtype=torch.float64, device=self.device))
        merge_const_root = None
        if bn_h is not None and self.fused_g and self.tower_f_t is not None:
            c, hgt, wid = self.shape
            wr, merge_const_root = through(g.fc_merge.weight[:, :D], None, bn_h)
            wrp = self._dev(wr.view(c, hgt * wid, wr.shape[1]).permute(1, 0, 2).reshape(D, wr.shape[1]))
            self.merge_root_perm = plan_wide(_hip.Dense(ptr(wrp), None, wr.shape[1], D, _hip.ACT["none"], 0))
            self.tower_f_root = f_tower_on(bn_h)
        if self.blob is None or self.blob.numel() != off:
            self.blob = torch.empty(off, dtype=torch.float32, device=self.device)
        for kind, src, pl in plans:
            if kind == "wide":
                _hip.check(lib.mz_wide_pack(C.byref(src), C.byref(pl), ptr(self.blob), _stream()), "mz_wide_pack")
            else:
                _hip.check(lib.mz_mlp_pack(C.byref(src), C.byref(pl), ptr(self.blob), _stream()), "mz_mlp_pack")
        # the fused fc_merge + first-convolution pass at the fp16 matrix rate: its layers as two fp16 planes
        self.split3 = {}
        if self.fused_g and self.split_merge_conv:
            for name in ("merge_t_perm", "conv_g", "merge_root_perm"):
                pl = getattr(self, name)
                if pl is not None:
                    out = torch.empty(int(lib.mz_packed_split3_words(C.byref(pl))), dtype=torch.int32, device=self.device)
                    _hip.check(lib.mz_packed_split3(C.byref(pl), ptr(self.blob), ptr(out), _stream()), "mz_packed_split3")
                    self.split3[name] = out
        self.table = torch.empty((A, D), dtype=torch.float32, device=self.device)
        _hip.check(lib.mz_action_table(ptr(self._dev(g.action_embedding.weight)), ptr(self._dev(g.fc_merge.weight)),
                                       ptr(self._dev(g.fc_merge.bias)) if g.fc_merge.bias is not None else None,
                                       D, A, ptr(self.table), _stream()), "mz_action_table")
        # per-action table of the bottleneck form: the constant part W b of the folded res_final included
        self.table_t = (self.table + merge_const[None, :]).contiguous() if merge_const is not None else None
        if self.fused_g:
            c, hgt, wid = self.shape
            self.table_t_perm = self.table_t.view(A, c, hgt * wid).permute(0, 2, 1).contiguous()  # [A][HW][C]
            if merge_const_root is not None:
                self.table_root_perm = (self.table + merge_const_root[None, :]).view(A, c, hgt * wid).permute(0, 2, 1).contiguous()
        self.has_root_form = self.merge_root_perm is not None and self.tower_f_root is not None and self.table_root_perm is not None

    # ---- scratch ----------------------------------------------------------------------------
    def _buf(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        """Scratch tensor, one per (name, current stream): calls on different streams never share it."""
        name = (name, _stream())
        t = self._bufs.get(name)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:n].view(*shape)

    def _group_index(self, B: int, A: int) -> torch.Tensor:
        key = ("gidx", B, A)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.arange(B, dtype=torch.int32, device=self.device) // A).contiguous()
            self._bufs[key] = t
        return t

    # ---- kernels ----------------------------------------------------------------------------
    def _wide(self, plan, x, ldx, x_rows, y, B) -> None:
        lib = _hip.lib()
        need = lib.mz_wide_workspace(C.byref(plan), B)
        ws = self._buf("wide_ws", (need,)) if need > 0 else None
        _hip.check(lib.mz_wide_forward(C.byref(plan), ptr(self.blob), ptr(x), ldx, ptr(x_rows), ptr(y), y.shape[-1], B,
                                       ptr(ws), need, _stream()), "mz_wide_forward")

    def _head(self, hd: _Head, name: str, x, ldx, x_rows, B) -> torch.Tensor:
        """Returns f32[B][out_features] (network output of the head, its own softmax included)."""
        lib = _hip.lib()
        out = self._buf(name + "_out", (B, hd.out_features))
        if hd.wide is not None:
            first = out if hd.rest is None else self._buf(name + "_mid", (B, hd.mid_features))
            self._wide(hd.wide, x, ldx, x_rows, first, B)
            x, ldx, x_rows = first, hd.mid_features, None
        if hd.rest is not None:
            _hip.check(lib.mz_mlp_forward(C.byref(hd.rest), ptr(self.blob), ptr(x), ldx, ptr(x_rows), ptr(out),
                                          hd.out_features, B, _stream()), "mz_mlp_forward")
        if hd.extra_softmax:
            _hip.check(lib.mz_policy_post(ptr(out), B, hd.out_features, hd.extra_softmax, ptr(out), None, _stream()),
                       "mz_policy_post")
        return out

    def _track(self, tower, blocks) -> None:
        """Device copies of the running statistics of a tower's BatchNorm layers (bn1, bn2, skip)."""
        if not self.track_running_stats:
            return
        layers = []
        for blk in blocks:
            for bn in (blk.bn1, blk.bn2, blk.downsample[1] if blk.downsample is not None else None):
                if bn is None or not bn.track_running_stats or bn.running_mean is None:
                    layers.append(None)
                    continue
                layers.append((bn, bn.running_mean.detach().to(self.device, torch.float32).clone(),
                               bn.running_var.detach().to(self.device, torch.float32).clone()))
        for e in layers:
            if e is not None:  # the buffers' state as read here: a later change of theirs is someone else's update
                self._buffers_seen[id(e[0])] = (e[0].running_mean.detach().clone(), e[0].running_var.detach().clone())
        self._running[id(tower)] = layers
        # counted on the device, so that launches replayed from a recorded graph count as well
        self._calls[id(tower)] = torch.zeros(1, dtype=torch.int64, device=self.device)

    def sync_running_stats(self) -> None:
        """Writes the advanced running statistics and call counts back into the torch modules.  A module whose
        buffers changed since they were read here (a training step ran the module in train mode before this
        evaluator's updates were handed over) keeps its own values: overwriting them would undo that step's
        update; the updates held here are dropped with a warning (only the call count is added)."""
        seen, counters = set(), {}
        for key, layers in self._running.items():
            cnt = self._calls.get(key)
            for e in layers or ():
                if e is None:
                    continue
                counters.setdefault(id(e[0]), {})[id(cnt)] = cnt  # towers sharing blocks share the counter
                if id(e[0]) in seen:
                    continue
                seen.add(id(e[0]))
                bn, rm, rv = e
                was = self._buffers_seen.get(id(bn))
                if was is None or not (torch.equal(bn.running_mean, was[0]) and torch.equal(bn.running_var, was[1])):
                    import warnings
                    warnings.warn("BatchNorm running statistics changed in the module since self-play read them: "
                                  "self-play's updates are dropped (call sync_running_stats() before training)",
                                  RuntimeWarning, stacklevel=2)
                    with torch.no_grad():  # continue from the module's values
                        rm.copy_(bn.running_mean.detach().to(rm.device, torch.float32))
                        rv.copy_(bn.running_var.detach().to(rv.device, torch.float32))
                else:
                    with torch.no_grad():
                        bn.running_mean.copy_(rm.to(bn.running_mean.device))
                        bn.running_var.copy_(rv.to(bn.running_var.device))
                self._buffers_seen[id(bn)] = (bn.running_mean.detach().clone(), bn.running_var.detach().clone())
        for key, layers in self._running.items():
            for e in layers or ():
                if e is not None and id(e[0]) in counters:
                    n = sum(int(c.item()) for c in counters.pop(id(e[0])).values())
                    if e[0].num_batches_tracked is not None:
                        with torch.no_grad():
                            e[0].num_batches_tracked += n
        for c in {id(c): c for c in self._calls.values() if c is not None}.values():
            c.zero_()

    def _tower_call(self, tower, x, ldx, x_rows, row_add, row_sel, y, ldy, y_rows, B, group, t_out=None) -> None:
        lib = _hip.lib()
        layers = self._running.get(id(tower)) if self.track_running_stats else None
        stats = None
        if layers and B > 0:
            calls = B // group
            stats = self._buf(("bn_stats", id(tower), torch.cuda.current_stream().cuda_stream),
                              (lib.mz_tower_stats_floats(C.byref(tower), calls),))
        _hip.check(lib.mz_tower_forward_stats(C.byref(tower), ptr(x), ldx, ptr(x_rows), ptr(row_add), self.D,
                                              ptr(row_sel), ptr(y), ldy, ptr(y_rows), ptr(t_out),
                                              t_out.shape[-1] if t_out is not None else 0, B, group, ptr(stats), _stream()),
                   "mz_tower_forward")
        if stats is not None:
            self._advance_running(tower, layers, stats, calls, group)

    def _advance_running(self, tower, layers, stats, calls, group) -> None:
        """The momentum chain of every tracked BatchNorm layer over the statistics one launch exported."""
        lib = _hip.lib()
        off, hw = 0, self.shape[1] * self.shape[2]
        for b in range(tower.n_blocks):
            c = tower.block[b].out_channels
            for j in range(3):
                e = layers[3 * b + j]
                if e is not None:
                    bn, rm, rv = e
                    mean, var = stats[off:off + calls * c], stats[off + calls * c:off + 2 * calls * c]
                    shift = self._mean_shift.get(id(tower), {}).get(3 * b + j)
                    if shift is not None:
                        mean.view(calls, c).add_(shift)
                    _hip.check(lib.mz_bn_running_update(ptr(mean), ptr(var), calls, c, group * hw, float(bn.momentum),
                                                        ptr(rm), ptr(rv), _stream()), "mz_bn_running_update")
                off += 2 * calls * c
        self._calls[id(tower)].add_(calls)

    def _mlp_on(self, plan, name: str, x, ldx, x_rows, B, out=None):
        """Head on bottleneck values.  `out`: one-output heads may write straight into the caller's
        tensor (flat contiguous, or a 1-D view with a stride); returns None then."""
        ldy = plan.out_features
        direct = out is not None and plan.out_features == 1 and out.dtype == torch.float32
        if direct:
            ldy = 1 if out.is_contiguous() else (out.stride(0) if out.dim() == 1 else 0)
            direct = ldy > 0
        if not direct:
            res = self._buf(name + "_out", (B, plan.out_features))
            ldy = plan.out_features
        _hip.check(_hip.lib().mz_mlp_forward(C.byref(plan), ptr(self.blob), ptr(x), ldx, ptr(x_rows),
                                             ptr(out if direct else res), ldy, B, _stream()), "mz_mlp_forward")
        if direct:
            return None
        if out is not None:
            out.view(-1)[:B].copy_(res.view(-1))
            return None
        return res

    def reward_on_t(self, t, rows, B, out) -> None:
        """reward_mlp on the bottleneck values of B latents (rows of t), into out f32[B] (bottleneck form only)."""
        self._mlp_on(self.reward_t, "reward_t", t, t.shape[-1], rows, B, out=out)

    def value_with_rewards(self, t_in, B, value, t, rows, n, reward) -> bool:
        """f's value for B nodes given by their bottleneck values (t_in = (t, rows)) and reward_mlp for n other rows
        of `t`, the two heads in one launch (mz_mlp_forward_pair) after f's block.  value: f32[B], contiguous or a
        1-D view with a stride; reward f32[n] contiguous.  False: not available in this form (nothing was run)."""
        if not (self.bottleneck_f and self.tower_f_t is not None and self.value_t is not None and self.reward_t is not None):
            return False
        if value.dtype != torch.float32 or reward.dtype != torch.float32 or not reward.is_contiguous():
            return False
        ldv = 1 if value.is_contiguous() else (value.stride(0) if value.dim() == 1 else 0)
        if ldv <= 0:
            return False
        src, src_rows = t_in
        ft = self._buf("f_t", (B, self.bottleneck_f))
        self._tower_call(self.tower_f_t, src, src.shape[-1], src_rows, None, None, None, 0, None, B, 1, t_out=ft)
        _hip.check(_hip.lib().mz_mlp_forward_pair(C.byref(self.reward_t), ptr(t), t.shape[-1], ptr(rows), ptr(reward), 1, n,
                                                  C.byref(self.value_t), ptr(ft), ft.shape[-1], None, ptr(value), ldv, B,
                                                  ptr(self.blob), _stream()), "mz_mlp_forward_pair")
        return True

    def merge_conv(self, t, t_rows, action, n_src, root_form: bool = False) -> torch.Tensor:
        """fc_merge on bottleneck values + the per-action term + the first product of g's tower in one launch
        (only when `fused_g`): Y f32[rows][H*W][32] for rows = n_src (action i32[n_src]) or n_src * A (action None:
        every source row with all A actions)."""
        c, hgt, wid = self.shape
        B = n_src if action is not None else n_src * self.A
        Y = self._buf("g_Y", (B, hgt * wid, 32))
        plan, table = (self.merge_root_perm, self.table_root_perm) if root_form else (self.merge_t_perm, self.table_t_perm)
        if self.split_merge_conv:
            m3 = self.split3["merge_root_perm" if root_form else "merge_t_perm"]
            _hip.check(_hip.lib().mz_merge_conv_forward_split(C.byref(plan), ptr(m3), C.byref(self.conv_g), ptr(self.split3["conv_g"]),
                                                              ptr(t), t.shape[-1], ptr(t_rows), ptr(table), ptr(action), self.A,
                                                              n_src, c, hgt * wid, ptr(Y), _stream()), "mz_merge_conv_forward_split")
        else:
            _hip.check(_hip.lib().mz_merge_conv_forward(C.byref(plan), C.byref(self.conv_g), ptr(self.blob), ptr(t),
                                                        t.shape[-1], ptr(t_rows), ptr(table), ptr(action), self.A, n_src,
                                                        c, hgt * wid, ptr(Y), _stream()), "mz_merge_conv_forward")
        return Y

    # ---- batched g / f (same calls as PackedNetworks) -----------------------------------------
    def merge_rows(self, latent, B, *, ld=None, rows=None, t_in=None) -> torch.Tensor:
        """fc_merge's latent half for B rows (needs no action): pass the result to
        `dynamics(..., action=..., base=...)` to overlap it with the f call that picks the action.
        t_in = (t, rows): evaluate it on the bottleneck values of those latents instead."""
        base = self._buf("merge_base", (B, self.D))
        if t_in is not None:
            t, t_rows = t_in
            self._wide(self.merge_t, t, t.shape[-1], t_rows, base, B)
        else:
            self._wide(self.merge, latent, ld or self.D, rows, base, B)
        return base

    def dynamics(self, latent, B, *, ld=None, rows=None, action=None, next_latent=None, ld_next=None, next_rows=None,
                 reward=None, base=None, t_in=None, t_out=None, want_latent=True, root_form: bool = False) -> None:
        """action None: the expansion pattern -- consecutive groups of A rows share one latent and
        take actions 0..A-1; each group is one call of the reference (shared batch statistics).
        action i32[B]: B independent calls of batch size 1.
        Bottleneck form (only when `bottleneck_g` > 0): t_in = (t f32[*, bottleneck_g], rows) holds the
        bottleneck values of the source latents (row index like `rows`; for the expansion pattern one
        entry per group is read, `rows[::A]`); t_out f32[*, bottleneck_g] receives those of the results,
        scattered by next_rows like next_latent.  want_latent=False (bottleneck form only) skips
        res_final and the store of the C*H*W latents: t_out is the result."""
        D, A = self.D, self.A
        ld = ld or D
        use_t = t_in is not None and self.merge_t is not None
        if (use_t and self.fused_g and base is None and not want_latent and t_out is not None and self.reward_t is not None
                and (action is not None or B % A == 0) and (B if action is not None else B // A) >= self.fuse_min_rows):
            # fc_merge, the action term and the first block's big product in one kernel, the rest of the block in a second
            lib = _hip.lib()
            t, t_rows = t_in
            c, hgt, wid = self.shape
            if action is None:
                U = B // A
                if t_rows is not None and t_rows.numel() != U:
                    t_rows = t_rows[::A].contiguous()
                n_src, group = U, A
            else:
                n_src, group = B, 1
            Y = self.merge_conv(t, t_rows, action, n_src, root_form=root_form)
            layers = self._running.get(id(self.tower_g)) if self.track_running_stats else None
            stats, calls = None, B // group
            if layers:
                stats = self._buf(("bn_stats", id(self.tower_g)), (lib.mz_tower_stats_floats(C.byref(self.tower_g), calls),))
            _hip.check(lib.mz_tower_tail_forward(C.byref(self.tower_g), ptr(Y), B, group, ptr(t_out), t_out.shape[-1], ptr(next_rows),
                                                 ptr(stats), _stream()), "mz_tower_tail_forward")
            if stats is not None:
                self._advance_running(self.tower_g, layers, stats, calls, group)
            if reward is not None:
                self._mlp_on(self.reward_t, "reward_t", t_out, t_out.shape[-1], next_rows, B, out=reward)
            return
        assert not root_form, "the root form needs the fused fc_merge + convolution path"
        if action is None:
            assert B % A == 0, "expansion rows come in groups of A"
            U = B // A

            def per_group(r):  # one source row per group: given as such (U entries) or repeated A times (B entries)
                if r is None:
                    return torch.arange(U, dtype=torch.int32, device=self.device) * A
                return r if r.numel() == U else r[::A].contiguous()

            if use_t:
                t, t_rows = t_in
                base = self.merge_rows(None, U, t_in=(t, per_group(t_rows)))
            else:
                base = self.merge_rows(latent, U, ld=ld, rows=per_group(rows))
            x_rows, row_sel, group = self._group_index(B, A), None, A
        else:
            if base is None:
                base = self.merge_rows(latent, B, ld=ld, rows=rows, t_in=t_in if use_t else None)
            else:
                assert t_in is None or use_t
            x_rows, row_sel, group = None, action, 1
        if t_out is not None and not self.bottleneck_g:
            t_out = None
        if not want_latent and t_out is not None and self.reward_t is not None:
            next_latent = None
        elif next_latent is None:
            next_latent, ld_next, next_rows = self._buf("g_next", (B, D)), D, None
        ld_next = ld_next or D
        self._tower_call(self.tower_g, base, D, x_rows, self.table_t if use_t else self.table, row_sel, next_latent,
                         ld_next if next_latent is not None else 0, next_rows, B, group, t_out=t_out)
        if reward is not None:
            if t_out is not None and self.reward_t is not None:
                self._mlp_on(self.reward_t, "reward_t", t_out, t_out.shape[-1], next_rows, B, out=reward)
            else:
                out = self._head(self.reward_head, "reward", next_latent, ld_next, next_rows, B)
                (reward if reward.dim() == 1 else reward.view(-1)[:B]).copy_(out.view(-1))

    def prediction(self, latent, B, *, ld=None, rows=None, softmax_times=0, policy=None, value=None, argmax=None,
                   t_in=None, root_form: bool = False) -> None:
        """t_in = (t f32[*, bottleneck_g], rows): the bottleneck values of the latents (non-root nodes of
        the search); f's first block then reads them instead of the latents."""
        if policy is None and value is None and argmax is None:
            return
        D = self.D
        if self.bottleneck_f:
            # f's own tower ends in res_final too: its heads read the block output directly
            dt = self.bottleneck_f
            t = self._buf("f_t", (B, dt))
            if t_in is not None and self.tower_f_t is not None:
                src, src_rows = t_in
                self._tower_call(self.tower_f_root if root_form else self.tower_f_t, src, src.shape[-1], src_rows, None, None, None,
                                 0, None, B, 1, t_out=t)
            else:
                self._tower_call(self.tower_f, latent, ld or D, rows, None, None, None, 0, None, B, 1, t_out=t)
            head = lambda plan, name: self._mlp_on(plan, name, t, dt, None, B)  # noqa: E731
            p_plan, v_plan = self.policy_t, self.value_t
        else:
            t = self._buf("f_tower", (B, D))
            self._tower_call(self.tower_f, latent, ld or D, rows, None, None, t, D, None, B, 1)
            head = lambda hd, name: self._head(hd, name, t, D, None, B)  # noqa: E731
            p_plan, v_plan = self.policy_head, self.value_head
        if (policy is not None or argmax is not None) and self.bottleneck_f and (policy is None or (
                policy.dtype == torch.float32 and policy.is_contiguous() and policy.numel() == B * self.A)):
            # the whole head, the extra softmax passes and the argmax in one launch
            _hip.check(_hip.lib().mz_mlp_forward_post(C.byref(p_plan), ptr(self.blob), ptr(t), dt, None, ptr(policy),
                                                      self.A, B, softmax_times, ptr(argmax), _stream()),
                       "mz_mlp_forward_post")
        elif policy is not None or argmax is not None:
            p = head(p_plan, "policy")
            _hip.check(_hip.lib().mz_policy_post(ptr(p), B, self.A, softmax_times, ptr(policy), ptr(argmax), _stream()),
                       "mz_policy_post")
        if value is not None:
            if self.bottleneck_f:
                self._mlp_on(v_plan, "value", t, dt, None, B, out=value)
            else:
                v = head(v_plan, "value")
                (value if value.dim() == 1 else value.view(-1)[:B]).copy_(v.view(-1))
