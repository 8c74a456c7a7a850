"""Convolutional representation network h on the HIP kernels, for G games per call.

`ConvRepresentation(repr_net)` evaluates RepresentationNetwork.forward
(src/neural_networks/neural_network.py:83-102) layer by layer -- ConvLayer, PoolLayer, ResBlock
(src/neural_networks/network_builder.py:32-151), the optional 1x1 final convolution and the
adaptive average pool -- on channels-last activations:

* a convolution is ONE matrix product on the matrix cores (rows = output pixels of all games,
  reduction = (tap, input channel), A operand gathered through a pixel index table, bias and
  activation in the epilogue).  Layers with input channels a multiple of 64 and output channels a
  multiple of 128 -- all the heavy ones -- run at the fp16 matrix rate with float32 accuracy
  (`mz_conv3_forward`: every float32 operand is held as two fp16 terms, three fp16 products per float32
  product into one float32 accumulator; the kernels split float32 activations while they stage them);
  other layers with 64-multiple input channels use the fp32 matrix instruction
  (`mz_wide_forward_gather`), the rest (few input channels) the direct kernel.
  Stride-1 "same" convolutions among them (the ResBlock convolutions) stage the contiguous pixel
  range a tile's taps read in LDS once per channel chunk (`mz_conv3_same_forward`; `staged_range=False`
  keeps them on the gathering kernel, whose bits they reproduce).
  `fast_convolutions=False` keeps everything on the fp32 instruction;
* every BatchNorm2d normalises with the statistics of ONE game, because the reference calls h with
  batch size 1 and never leaves train mode (src/training_data_generator.py:117, main.py:39 only
  disables grad): `mz_bn_stats` per (game, channel) -- or, after a split-product convolution whose games
  are whole 64-pixel blocks, `mz_bn_stats_from_sums` on the block sums the convolution's epilogue left,
  without another pass over its output (`fused_statistics=False` keeps the separate pass) --
  and `mz_bn_apply` with the residual sum and the activation of the block fused in; inside a block
  the first BatchNorm and activation are applied by the second convolution while it stages its input
  (`mz_conv3_same_bn_forward`; `fused_input_bn=False` writes the normalised activation out instead);
* `track_running_stats=True` also advances the BatchNorm running statistics exactly as those G
  batch-size-1 calls in game order would (`mz_bn_running_update`), and `sync_running_stats()` writes
  them (and num_batches_tracked) back into the caller's module, so that a state_dict saved after
  self-play equals the reference's.

There is no CPU fallback: every step is a kernel of libmuzero_hip.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from muzero_amd import _hip
from muzero_amd._hip import ptr
from muzero_amd.neural_networks.network_builder import ConvLayer, PoolLayer, ResBlock

_ACT_OF = {torch.nn.Identity: "none", torch.nn.ReLU: "relu", torch.nn.Tanh: "tanh", torch.nn.Sigmoid: "sigmoid"}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _act(module) -> int:
    kind = type(module)
    if kind not in _ACT_OF:
        raise NotImplementedError(f"activation {kind.__name__} inside the representation network is not element-wise")
    return _hip.ACT[_ACT_OF[kind]]


def _one(v) -> int:
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError("only square kernels / equal strides and paddings are supported")
        return int(v[0])
    return int(v)


class _Conv:
    """One nn.Conv2d: packed for the gathered matrix product, or k-major weights for the direct kernel."""

    def __init__(self, conv: torch.nn.Conv2d, act: int):
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.k, self.stride, self.pad = _one(conv.kernel_size), _one(conv.stride), _one(conv.padding)
        if _one(conv.dilation) != 1 or conv.groups != 1 or conv.padding_mode != "zeros":
            raise NotImplementedError("dilated, grouped or non-zero-padded convolutions are not supported")
        self.act = act
        self.module = conv
        # the matrix product pads the output channels to 16: still far cheaper than one lane per output value
        # once the reduction is long
        self.matrix = self.cin % 64 == 0
        self.fast = self.cin % 64 == 0 and self.cout % 128 == 0 and self.k <= 4   # split-fp16 product (mz_conv3_forward)
        self.w3 = None        # i16[2][cout * k * k * cin]: the split k-major weight (fp16 hi and lo planes)
        self.plan = None      # _hip.PackedLayer (matrix form)
        self.wk = None        # f32[cout][k][k][cin]
        self.bias = None

    def out_hw(self, H: int, W: int) -> tuple[int, int]:
        return (H + 2 * self.pad - self.k) // self.stride + 1, (W + 2 * self.pad - self.k) // self.stride + 1


class _Bn:
    def __init__(self, bn: torch.nn.BatchNorm2d):
        self.module = bn
        self.C = bn.num_features
        self.eps = float(bn.eps)
        self.momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not supported")
        self.weight = self.bias = None
        self.running_mean = self.running_var = None
        self.seen = None  # the module's running buffers as last read or written by the evaluator (the native BatchNorm call does not bump their version counters)
        self.calls = 0   # batch-size-1 calls since the running statistics were last written back


class ConvRepresentation:
    def __init__(self, repr_net: torch.nn.Module, device="cuda", track_running_stats: bool = False,
                 fast_convolutions: bool = True, staged_range: bool = True, fused_statistics: bool = True,
                 fused_input_bn: bool = True, fused_block_output: bool = True, f32_gather: bool = True):
        self.src = repr_net
        # the strided convolutions gather float32 rows and split them while staging (mz_conv3_forward_f32): their
        # producers then write the float32 value only, no split planes; identical bits either way
        self.f32_gather = f32_gather
        self.fused_input_bn = fused_input_bn
        # a ResBlock's bn2 + skip sum + activation applied while the NEXT block's conv1 stages its input (which also
        # writes the block output once, for that block's own skip path) instead of a pass of its own
        self.fused_block_output = fused_block_output
        self.staged_range = staged_range
        self.fused_statistics = fused_statistics
        self.device = torch.device(device)
        self.track_running_stats = track_running_stats
        self.fast_convolutions = fast_convolutions
        self.latent_hw = (repr_net.latent_height, repr_net.latent_width)
        self._bufs: dict = {}
        self._index: dict = {}
        self.blob = None
        self._keep: list = []
        self.refresh()

    # ---- packing ------------------------------------------------------------------------------
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(self.device, torch.float32).contiguous().clone()
        self._keep.append(t)
        return t

    def refresh(self) -> None:
        """Re-reads the module's weights (call after a weight update)."""
        self._keep = []
        self.steps = []          # ("conv", _Conv) | ("pool", k, s, is_max) | ("block", conv1, bn1, conv2, bn2, down, dbn, act, pool)
        self.convs: list[_Conv] = []
        self.bns: list[_Bn] = []

        def conv(c, act):
            cv = _Conv(c, act)
            self.convs.append(cv)
            return cv

        def bn(b):
            bb = _Bn(b)
            self.bns.append(bb)
            return bb

        def block(blk: ResBlock):
            act = _act(blk.activation)
            down = dbn = None
            if blk.downsample is not None:
                down, dbn = conv(blk.downsample[0], _hip.ACT["none"]), bn(blk.downsample[1])
            pool = None
            if not isinstance(blk.pool, torch.nn.Identity):
                pool = (_one(blk.pool.kernel_size), _one(blk.pool.stride))
                if _one(blk.pool.padding) != 0:
                    raise NotImplementedError("padded pooling is not supported")
            self.steps.append(("block", conv(blk.conv1, _hip.ACT["none"]), bn(blk.bn1), conv(blk.conv2, _hip.ACT["none"]),
                               bn(blk.bn2), down, dbn, act, pool))

        for layer in self.src.downsample_layers:
            if isinstance(layer, ConvLayer):
                self.steps.append(("conv", conv(layer.conv, _act(layer.activation))))
            elif isinstance(layer, PoolLayer):
                p = layer.pool
                if _one(p.padding) != 0 or getattr(p, "ceil_mode", False):
                    raise NotImplementedError("padded / ceil-mode pooling is not supported")
                self.steps.append(("pool", _one(p.kernel_size), _one(p.stride), isinstance(p, torch.nn.MaxPool2d)))
            elif isinstance(layer, ResBlock):
                block(layer)
            else:
                raise NotImplementedError(f"unexpected layer {type(layer).__name__} in the representation network")
        for blk in self.src.res_blocks:
            block(blk)
        if isinstance(self.src.final_conv, torch.nn.Conv2d):
            self.steps.append(("conv", conv(self.src.final_conv, _hip.ACT["none"])))

        lib = _hip.lib()
        off = 0
        packs = []
        for cv in self.convs:
            w = self._dev(cv.module.weight)
            cv.wk = torch.empty(cv.cout * cv.k * cv.k * cv.cin, dtype=torch.float32, device=self.device)
            _hip.check(lib.mz_conv_weight_kmajor(ptr(w), cv.cout, cv.cin, cv.k, ptr(cv.wk), _stream()), "mz_conv_weight_kmajor")
            cv.bias = self._dev(cv.module.bias) if cv.module.bias is not None else None
            cv.fast = cv.fast and self.fast_convolutions
            if cv.fast:
                n = cv.wk.numel()
                cv.w3 = torch.empty((2, n), dtype=torch.int16, device=self.device)
                _hip.check(lib.mz_split3(ptr(cv.wk), n, ptr(cv.w3), n, _stream()), "mz_split3")
            elif cv.matrix:
                dense = _hip.Dense(ptr(cv.wk), ptr(cv.bias), cv.k * cv.k * cv.cin, cv.cout, cv.act, 0)
                cv.plan = _hip.PackedLayer()
                n = lib.mz_wide_plan(C.byref(dense), off, C.byref(cv.plan))
                if n < 0:
                    _hip.check(int(n), "mz_wide_plan")
                off += n
                packs.append((dense, cv.plan))
        if self.blob is None or self.blob.numel() != max(off, 1):
            self.blob = torch.empty(max(off, 1), dtype=torch.float32, device=self.device)
        for dense, plan in packs:
            _hip.check(lib.mz_wide_pack(C.byref(dense), C.byref(plan), ptr(self.blob), _stream()), "mz_wide_pack")
        for b in self.bns:
            m = b.module
            ones = torch.ones(b.C) if m.weight is None else m.weight
            zeros = torch.zeros(b.C) if m.bias is None else m.bias
            b.weight, b.bias = self._dev(ones), self._dev(zeros)
            if m.track_running_stats and m.running_mean is not None:
                b.running_mean, b.running_var = self._dev(m.running_mean), self._dev(m.running_var)
                b.seen = (m.running_mean.detach().clone(), m.running_var.detach().clone())  # a later change is someone else's update
            b.calls = 0

    def sync_running_stats(self) -> None:
        """Writes the running statistics (and call counts) accumulated by this evaluator into the
        caller's module, as the reference's own calls would have left them."""
        with torch.no_grad():
            for b in self.bns:
                m = b.module
                if b.calls == 0 or b.running_mean is None:
                    continue
                if not (torch.equal(m.running_mean, b.seen[0]) and torch.equal(m.running_var, b.seen[1])):
                    # a training step advanced the module's buffers before these updates were handed over: writing
                    # them back would undo that step; keep the module's values and go on from them
                    import warnings
                    warnings.warn("BatchNorm running statistics changed in the module since self-play read them: "
                                  "self-play's updates are dropped (call sync_running_stats() before training)",
                                  RuntimeWarning, stacklevel=2)
                    b.running_mean.copy_(m.running_mean.detach().to(self.device, torch.float32))
                    b.running_var.copy_(m.running_var.detach().to(self.device, torch.float32))
                else:
                    m.running_mean.copy_(b.running_mean.to(m.running_mean.device))
                    m.running_var.copy_(b.running_var.to(m.running_var.device))
                b.seen = (m.running_mean.detach().clone(), m.running_var.detach().clone())
                if m.num_batches_tracked is not None:
                    m.num_batches_tracked += b.calls
                b.calls = 0

    # ---- scratch ------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:n].view(*shape)

    def _index_table(self, G, H, W, cv: _Conv, Ho, Wo) -> torch.Tensor:
        key = (G, H, W, cv.k, cv.stride, cv.pad)
        t = self._index.get(key)
        if t is None:
            t = torch.empty((cv.k * cv.k, G * Ho * Wo), dtype=torch.int32, device=self.device)
            _hip.check(_hip.lib().mz_conv_index(ptr(t), G, H, W, cv.k, cv.stride, cv.pad, Ho, Wo, _stream()), "mz_conv_index")
            self._index[key] = t
        return t

    # ---- layers -------------------------------------------------------------------------------
    # An activation is held as float32 [G, H, W, C] (`f`), as two fp16 planes i16[2, G*H*W*C] (`s`), or
    # both; whichever form a consumer lacks is made on demand.
    class _Act:
        __slots__ = ("f", "s", "n", "sums")

        def __init__(self, f=None, s=None, n=0):
            self.f, self.s, self.n, self.sums = f, s, n, None

    def _f32(self, a, name):
        if a.f is None:
            a.f = self._buf(name + "_f", (a.n,))
            _hip.check(_hip.lib().mz_unsplit3(ptr(a.s), a.n, a.n, ptr(a.f), _stream()), "mz_unsplit3")
        return a.f

    def _s3(self, a, name):
        if a.s is None:
            a.s = self._buf(name + "_s", (2, a.n), torch.int16)
            _hip.check(_hip.lib().mz_split3(ptr(a.f), a.n, ptr(a.s), a.n, _stream()), "mz_split3")
        return a.s

    def _staged(self, cv: _Conv, G: int, H: int, W: int, raw: bool = True, with_bn=False) -> bool:
        """Whether `cv` runs on the kernel that stages a tile's contiguous pixel range (stride-1 "same") for G images
        of H x W pixels; `raw`: from a float32 input (split while staged), `with_bn`: with the previous layer's
        BatchNorm applied while staging.  The kernel's own limits decide (mz_conv3_same_fits)."""
        if not (cv.fast and self.staged_range and cv.stride == 1 and cv.k in (1, 3) and cv.pad == (cv.k - 1) // 2):
            return False
        return bool(_hip.lib().mz_conv3_same_fits(cv.cin, G, H, W, cv.k, cv.cout, 1 if raw else 0, int(with_bn)))

    def _conv(self, a, G, H, W, cv: _Conv, name: str, split_out: bool = False, sums: bool = False, pre=None, block=None):
        """a: input activation [G, H, W, cin] -> (output activation, Ho, Wo).  The output is float32
        unless `split_out` and the layer runs on the split-fp16 product (then split only).
        `sums`: where the kernel can, also leave the column sums of every 64-row block in `out.sums`
        (the BatchNorm statistics that follow then need no pass over the output).
        `pre` = (scale, shift, activation): `a` is the raw float32 output of the previous convolution and
        the layer reads activation(a * scale + shift) (only for layers `_staged` accepts).
        `block` = (scale, shift, activation, residual, res_scale, res_shift, block_out): `a` is the raw output of the
        previous ResBlock's conv2 and the layer reads that block's output activation(a * scale + shift + skip), which it
        also writes to `block_out` (mz_conv3_same_block_forward)."""
        lib = _hip.lib()
        Ho, Wo = cv.out_hw(H, W)
        if Ho <= 0 or Wo <= 0:
            raise ValueError(f"convolution kernel {cv.k} does not fit a {H}x{W} input")
        rows = G * Ho * Wo
        n_out = rows * cv.cout
        out = self._Act(n=n_out)
        if cv.fast:
            if block is None and pre is None and a.f is not None and self._staged(cv, G, H, W) and \
                    (a.s is None or not self._staged(cv, G, H, W, raw=False)):
                pre = (None, None, 0)   # a float32 activation is split while it is staged
            x3 = None
            if split_out:
                out.s = self._buf(name + "_s", (2, n_out), torch.int16)
            else:
                out.f = self._buf(name, (n_out,))
            if sums and self.fused_statistics and (Ho * Wo) % 64 == 0:
                out.sums = self._buf(name + "_sums", (rows // 64, cv.cout, 2), torch.float64)
            if block is not None:
                sc, sh, act_in, res, rsc, rsh, block_out = block
                _hip.check(lib.mz_conv3_same_block_forward(ptr(a.f), ptr(sc), ptr(sh), ptr(res), ptr(rsc), ptr(rsh), act_in, ptr(block_out),
                                                           cv.cin, G, H, W, cv.k, ptr(cv.w3), cv.w3.shape[1], ptr(cv.bias), cv.act,
                                                           ptr(out.f), ptr(out.s), n_out, cv.cout, ptr(out.sums), _stream()),
                           "mz_conv3_same_block_forward")
                return out, Ho, Wo
            if pre is not None:
                _hip.check(lib.mz_conv3_same_bn_forward(ptr(a.f), ptr(pre[0]), ptr(pre[1]), pre[2], cv.cin, G, H, W, cv.k, ptr(cv.w3),
                                                        cv.w3.shape[1], ptr(cv.bias), cv.act, ptr(out.f), ptr(out.s), n_out, cv.cout,
                                                        ptr(out.sums), _stream()), "mz_conv3_same_bn_forward")
                return out, Ho, Wo
            if not (self.f32_gather and a.f is not None and a.s is None):
                x3 = self._s3(a, name + "_in")
            if x3 is not None and self._staged(cv, G, H, W, raw=False):
                # the taps of a stride-1 "same" convolution read one contiguous pixel range per tile
                _hip.check(lib.mz_conv3_same_forward(ptr(x3), a.n, cv.cin, G, H, W, cv.k, ptr(cv.w3), cv.w3.shape[1], ptr(cv.bias),
                                                     cv.act, ptr(out.f), ptr(out.s), n_out, cv.cout, ptr(out.sums), _stream()),
                           "mz_conv3_same_forward")
                return out, Ho, Wo
            idx = self._index_table(G, H, W, cv, Ho, Wo)
            if x3 is None:
                _hip.check(lib.mz_conv3_forward_f32(ptr(a.f), cv.cin, ptr(idx), cv.k * cv.k, ptr(cv.w3), cv.w3.shape[1], ptr(cv.bias),
                                                    cv.act, ptr(out.f), ptr(out.s), n_out, rows, cv.cout, ptr(out.sums), _stream()),
                           "mz_conv3_forward_f32")
                return out, Ho, Wo
            _hip.check(lib.mz_conv3_forward(ptr(x3), a.n, cv.cin, ptr(idx), cv.k * cv.k, ptr(cv.w3), cv.w3.shape[1], ptr(cv.bias),
                                            cv.act, ptr(out.f), ptr(out.s), n_out, rows, cv.cout, ptr(out.sums), _stream()),
                       "mz_conv3_forward")
            return out, Ho, Wo
        x = self._f32(a, name + "_in")
        out.f = self._buf(name, (n_out,))
        if cv.matrix:
            need = lib.mz_wide_workspace(C.byref(cv.plan), rows)
            ws = self._buf("ws", (need,)) if need > 0 else None
            if cv.k == 1 and cv.stride == 1 and cv.pad == 0:
                _hip.check(lib.mz_wide_forward(C.byref(cv.plan), ptr(self.blob), ptr(x), cv.cin, None, ptr(out.f), cv.cout, rows,
                                               ptr(ws), need, _stream()), "mz_wide_forward")
            else:
                idx = self._index_table(G, H, W, cv, Ho, Wo)
                _hip.check(lib.mz_wide_forward_gather(C.byref(cv.plan), ptr(self.blob), ptr(x), cv.cin, ptr(idx), cv.k * cv.k,
                                                      ptr(out.f), cv.cout, rows, ptr(ws), need, _stream()), "mz_wide_forward_gather")
        else:
            _hip.check(lib.mz_conv_direct_nhwc(ptr(x), ptr(cv.wk), ptr(cv.bias), G, H, W, cv.cin, cv.cout, cv.k, cv.stride,
                                               cv.pad, cv.act, ptr(out.f), _stream()), "mz_conv_direct_nhwc")
        return out, Ho, Wo

    def _stats(self, raw, G, P, b: _Bn, name: str):
        """Per-game BatchNorm scale / shift of the raw convolution output `raw` (an _Act with .f)."""
        lib = _hip.lib()
        scale, shift = self._buf(name + "_scale", (G, b.C)), self._buf(name + "_shift", (G, b.C))
        track = self.track_running_stats and b.running_mean is not None
        mean = self._buf(name + "_mean", (G, b.C)) if track else None
        var = self._buf(name + "_var", (G, b.C)) if track else None
        if raw.sums is not None:
            _hip.check(lib.mz_bn_stats_from_sums(ptr(raw.sums), G, P, b.C, ptr(b.weight), ptr(b.bias), b.eps, ptr(scale),
                                                 ptr(shift), ptr(mean), ptr(var), _stream()), "mz_bn_stats_from_sums")
        else:
            _hip.check(lib.mz_bn_stats(ptr(raw.f), G, P, b.C, ptr(b.weight), ptr(b.bias), b.eps, ptr(scale), ptr(shift),
                                       ptr(mean), ptr(var), _stream()), "mz_bn_stats")
        if track:
            _hip.check(lib.mz_bn_running_update(ptr(mean), ptr(var), G, b.C, P, b.momentum, ptr(b.running_mean),
                                                ptr(b.running_var), _stream()), "mz_bn_running_update")
            b.calls += G
        return scale, shift

    def _apply(self, raw, scale, shift, act, G, P, Cc, name, split_out, res=None, res_scale=None, res_shift=None):
        """BatchNorm scale / shift (+ residual) + activation on the raw convolution output `raw`;
        `res`: None, an activation (identity skip) or a raw float32 tensor with its own scale / shift."""
        lib = _hip.lib()
        n = G * P * Cc
        out = self._Act(n=n)
        if split_out:
            out.s = self._buf(name + "_s", (2, n), torch.int16)
        else:
            out.f = self._buf(name, (n,))
        res_f = res3 = None
        if res is not None and res_scale is not None:
            res_f = res
        elif res is not None:
            if res.f is not None:
                res_f = res.f
            else:
                res3 = res.s
        _hip.check(lib.mz_bn_apply3(ptr(raw), ptr(scale), ptr(shift), ptr(res_f), ptr(res_scale), ptr(res_shift), ptr(res3), n,
                                    act, ptr(out.f), ptr(out.s), n, G, P, Cc, _stream()), "mz_bn_apply3")
        return out

    def _wants_split(self, si: int) -> bool:
        """Whether the consumer of step si's output reads the split form."""
        if si + 1 >= len(self.steps):
            return False
        nxt = self.steps[si + 1]
        if self.f32_gather:
            return False   # every split-product kernel reads float32 rows and splits them while staging
        first = nxt[1] if nxt[0] in ("conv", "block") else None
        # the staged-range kernel splits a float32 activation on the fly; the gathering one reads planes
        return first is not None and first.fast and not (self.staged_range and first.stride == 1 and first.k in (1, 3)
                                                         and first.pad == (first.k - 1) // 2)

    def final_is_pointwise(self) -> bool:
        """Whether the network ends in a 1x1, stride-1 convolution (neural_network.py:51,60-61): the latent is then
        that convolution of the pooled tensor before it, final_conv(pool(x)) = pool(final_conv(x))."""
        last = self.steps[-1] if self.steps else None
        return (last is not None and last[0] == "conv" and last[1].module is self.src.final_conv
                and last[1].k == 1 and last[1].stride == 1 and last[1].pad == 0)

    def __call__(self, history: torch.Tensor, want_root_t: bool = False) -> torch.Tensor:
        """history f32[G, C*K+K, H, W] (the tensor of mz_history_push_assemble) -> latent f32[G, c, h, w].
        `want_root_t`: when the network ends in a 1x1 convolution, `self.root_t` f32[G, c_in, h, w] also receives the
        pooled input of that convolution (the values a residual-tower search can start from instead of the latent);
        None otherwise."""
        lib = _hip.lib()
        self.root_t = None
        pre_final = want_root_t and self.final_is_pointwise()
        G, Cc, H, W = history.shape
        x_in = history.to(self.device, torch.float32).contiguous()
        n_in = G * H * W * Cc
        if self._wants_split(-1) and self.steps[0][0] == "conv":  # the first layer reads the split form only
            x = self._Act(s=self._buf("in_nhwc_s", (2, n_in), torch.int16), n=n_in)
            _hip.check(lib.mz_nchw_split3(ptr(x_in), G, Cc, H * W, ptr(x.s), n_in, _stream()), "mz_nchw_split3")
        else:
            x = self._Act(f=self._buf("in_nhwc", (n_in,)), n=n_in)
            _hip.check(lib.mz_nchw_to_nhwc(ptr(x_in), G, Cc, H * W, ptr(x.f), _stream()), "mz_nchw_to_nhwc")
        # activations ping-pong between named buffers; a block's input must survive until its skip sum
        pending = None  # a block output not formed yet: the next block's conv1 forms it (fused_block_output)
        for si, step in enumerate(self.steps):
            tag = f"s{si % 2}"
            split_next = self._wants_split(si)
            if pre_final and si == len(self.steps) - 1:
                h, w = self.latent_hw
                self.root_t = torch.empty((G, Cc, h, w), dtype=torch.float32, device=self.device)
                _hip.check(lib.mz_adaptive_avg_to_nchw(ptr(self._f32(x, "pre_final")), G, H, W, Cc, h, w, ptr(self.root_t),
                                                       _stream()), "mz_adaptive_avg_to_nchw")
            if step[0] == "conv":
                cv = step[1]
                if cv.cin != Cc:
                    raise ValueError(f"layer {si} expects {cv.cin} channels, got {Cc}")
                x, H, W = self._conv(x, G, H, W, cv, tag + "_y", split_out=split_next)
                Cc = cv.cout
            elif step[0] == "pool":
                _, k, s_, is_max = step
                Ho, Wo = (H - k) // s_ + 1, (W - k) // s_ + 1
                y = self._Act(f=self._buf(tag + "_y", (G * Ho * Wo * Cc,)), n=G * Ho * Wo * Cc)
                _hip.check(lib.mz_pool2d_nhwc(ptr(self._f32(x, tag + "_pin")), G, H, W, Cc, k, s_, 1 if is_max else 0, ptr(y.f),
                                              _stream()), "mz_pool2d_nhwc")
                x, H, W = y, Ho, Wo
            else:
                _, c1, b1, c2, b2, down, dbn, act, pool = step
                if c1.cin != Cc:
                    raise ValueError(f"block {si} expects {c1.cin} channels, got {Cc}")
                if pending is not None:
                    # the previous block's bn2 + skip sum + activation, applied while conv1 stages its input; `x` becomes
                    # that block's output, written by the same launch
                    raw_p, sc_p, sh_p, act_p, res_p, rsc_p, rsh_p, n_p, name_p = pending
                    y_prev = self._Act(f=self._buf(name_p, (n_p,)), n=n_p)
                    raw1, H1, W1 = self._conv(raw_p, G, H, W, c1, tag + "_c1", sums=True,
                                              block=(sc_p, sh_p, act_p, res_p, rsc_p, rsh_p, y_prev.f))
                    x, pending = y_prev, None
                else:
                    raw1, H1, W1 = self._conv(x, G, H, W, c1, tag + "_c1", sums=True)
                sc1, sh1 = self._stats(raw1, G, H1 * W1, b1, tag + "_bn1")
                if self.fused_input_bn and self._staged(c2, G, H1, W1, with_bn=True):
                    # bn1 and the activation are applied while conv2 stages its input
                    raw2, H2, W2 = self._conv(raw1, G, H1, W1, c2, tag + "_c2", sums=True, pre=(sc1, sh1, act))
                else:
                    a1 = self._apply(raw1.f, sc1, sh1, act, G, H1 * W1, c1.cout, tag + "_a1", split_out=c2.fast)
                    raw2, H2, W2 = self._conv(a1, G, H1, W1, c2, tag + "_c2", sums=True)
                sc2, sh2 = self._stats(raw2, G, H2 * W2, b2, tag + "_bn2")
                out_split = split_next and pool is None
                nxt = self.steps[si + 1] if si + 1 < len(self.steps) else None
                defer = (self.fused_block_output and pool is None and nxt is not None and nxt[0] == "block" and raw2.f is not None
                         and not (pre_final and si + 1 == len(self.steps) - 1)
                         and nxt[1].cin == c2.cout and (down is not None or x.f is not None)
                         and self._staged(nxt[1], G, H2, W2, with_bn=3 if down is not None else 2))
                if down is not None:
                    d, Hd, Wd = self._conv(x, G, H, W, down, tag + "_d", sums=True)
                    if (Hd, Wd) != (H2, W2):
                        raise ValueError("the skip path and the main path of a residual block disagree in size")
                    scd, shd = self._stats(d, G, Hd * Wd, dbn, tag + "_bnd")
                    if defer:
                        pending = (raw2, sc2, sh2, act, d.f, scd, shd, G * H2 * W2 * c2.cout, tag + "_y")
                        y = None
                    else:
                        y = self._apply(raw2.f, sc2, sh2, act, G, H2 * W2, c2.cout, tag + "_y", out_split, res=d.f, res_scale=scd,
                                        res_shift=shd)
                else:
                    if (H, W, Cc) != (H2, W2, c2.cout):
                        raise ValueError("an identity skip needs equal input and output shapes")
                    if defer:
                        pending = (raw2, sc2, sh2, act, x.f, None, None, G * H2 * W2 * c2.cout, tag + "_y")
                        y = None
                    else:
                        y = self._apply(raw2.f, sc2, sh2, act, G, H2 * W2, c2.cout, tag + "_y", out_split, res=x)
                x, H, W, Cc = y, H2, W2, c2.cout
                if pool is not None:
                    k, s_ = pool
                    Ho, Wo = (H - k) // s_ + 1, (W - k) // s_ + 1
                    y = self._Act(f=self._buf(tag + "_p", (G * Ho * Wo * Cc,)), n=G * Ho * Wo * Cc)
                    _hip.check(lib.mz_pool2d_nhwc(ptr(x.f), G, H, W, Cc, k, s_, 1, ptr(y.f), _stream()), "mz_pool2d_nhwc")
                    x, H, W = y, Ho, Wo
        h, w = self.latent_hw
        out = torch.empty((G, Cc, h, w), dtype=torch.float32, device=self.device)
        _hip.check(lib.mz_adaptive_avg_to_nchw(ptr(self._f32(x, "last")), G, H, W, Cc, h, w, ptr(out), _stream()),
                   "mz_adaptive_avg_to_nchw")
        return out
