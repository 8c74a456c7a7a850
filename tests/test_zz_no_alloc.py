"""The allocation-free wide-layer entry point (mz_wide_forward_static: split-reduction partial sums
in a device array built into the library) against mz_wide_forward with a caller-provided
workspace: identical bits, on two calls in a row, the second with a smaller input; oversize
requests are refused."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(K, N, act="relu", seed=0):
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    torch.manual_seed(seed)
    w = (torch.randn(N, K) / K ** 0.5).cuda()
    b = torch.randn(N).cuda()
    d = _hip.Dense(ptr(w), ptr(b), K, N, _hip.ACT[act], 0)
    pl = _hip.PackedLayer()
    n = lib.mz_wide_plan(C.byref(d), 0, C.byref(pl))
    assert n > 0
    blob = torch.zeros(n, device="cuda")
    _hip.check(lib.mz_wide_pack(C.byref(d), C.byref(pl), ptr(blob), torch.cuda.current_stream().cuda_stream), "pack")
    torch.cuda.synchronize()
    return pl, blob, (w, b)


def _both(pl, blob, x):
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, N = x.shape[0], pl.N
    need = lib.mz_wide_workspace(C.byref(pl), B)
    ws = torch.empty(max(need, 1), device="cuda")
    y_ref = torch.full((B, N), 3.0, device="cuda")
    _hip.check(lib.mz_wide_forward(C.byref(pl), ptr(blob), ptr(x), x.shape[1], None, ptr(y_ref), N, B, ptr(ws), need, st),
               "mz_wide_forward")
    y = torch.full((B, N), 5.0, device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    _hip.check(lib.mz_wide_forward_static(C.byref(pl), ptr(blob), ptr(x), x.shape[1], None, ptr(y), N, B, st),
               "mz_wide_forward_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return y, y_ref, need


def test_static_scratch_entry_point_equals_workspace_entry_point():
    pl, blob, (w, b) = _layer(1024, 512)
    x1 = torch.randn(256, 1024, device="cuda")
    y1, ref1, need1 = _both(pl, blob, x1)
    assert need1 > 0, "this shape must split the reduction, or the scratch array is not exercised"
    assert torch.equal(y1, ref1)
    x2 = torch.randn(40, 1024, device="cuda")      # smaller: the array still holds the first call's sums
    y2, ref2, need2 = _both(pl, blob, x2)
    assert 0 < need2 < need1
    assert torch.equal(y2, ref2)
    want = torch.relu(x2.double() @ w.double().t() + b.double()).float()
    torch.testing.assert_close(y2, want, rtol=1e-5, atol=1e-5)
    # a deep split on a tiny batch (the shipped head shape) and a layer that needs no scratch at all
    pl3, blob3, _ = _layer(9216, 24, act="none", seed=1)
    y3, ref3, need3 = _both(pl3, blob3, torch.randn(2, 9216, device="cuda"))
    assert need3 > 0 and torch.equal(y3, ref3)
    # partial sums that fill the built-in array to the last float
    from muzero_amd import _hip
    pl5, blob5, _ = _layer(4096, 2048, seed=4)
    y5, ref5, need5 = _both(pl5, blob5, torch.randn(512, 4096, device="cuda"))
    assert need5 == _hip.lib().mz_wide_static_workspace_floats() and torch.equal(y5, ref5)
    pl4, blob4, _ = _layer(64, 64, seed=2)
    y4, ref4, need4 = _both(pl4, blob4, torch.randn(64, 64, device="cuda"))
    assert need4 == 0 and torch.equal(y4, ref4)


def test_static_scratch_entry_point_refuses_what_does_not_fit():
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    cap = lib.mz_wide_static_workspace_floats()
    assert cap == 4 << 20
    # a small batch over a long reduction splits deepest: pick the first candidate whose partial sums
    # exceed the built-in array (the layout query needs no weights)
    K = N = B = None
    for K_, N_, B_ in ((4096, 2048, 512), (8192, 2048, 512), (8192, 4096, 384), (16384, 4096, 256)):
        probe = _hip.PackedLayer()
        d = _hip.Dense(16, None, K_, N_, _hip.ACT["relu"], 0)   # the layout query never follows the weight pointer
        assert lib.mz_wide_plan(C.byref(d), 0, C.byref(probe)) > 0
        if lib.mz_wide_workspace(C.byref(probe), B_) > cap:
            K, N, B = K_, N_, B_
            break
    assert K is not None, "no candidate shape exceeds the built-in scratch"
    pl, blob, _ = _layer(K, N, seed=3)
    x = torch.randn(B, K, device="cuda")
    y = torch.zeros(B, N, device="cuda")
    rc = lib.mz_wide_forward_static(C.byref(pl), ptr(blob), ptr(x), K, None, ptr(y), N, B,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc < 0 and b"scratch holds" in lib.mz_last_error()
    torch.cuda.synchronize()
    assert (y == 0).all()


def test_static_scratch_entry_point_under_graph_capture():
    """The callers the entry point is for: after mz_wide_static_init() on the device, a call captured into a graph
    (no allocation, no runtime query inside the capture) replays to the bits of the workspace entry point, also
    after the input buffer's contents change."""
    from muzero_amd import _hip
    from muzero_amd._hip import ptr
    lib = _hip.lib()
    _hip.check(lib.mz_wide_static_init(), "mz_wide_static_init")
    pl, blob, _ = _layer(2048, 256, seed=5)
    B, N = 96, pl.N
    x = torch.randn(B, 2048, device="cuda")
    y = torch.zeros(B, N, device="cuda")
    need = lib.mz_wide_workspace(C.byref(pl), B)
    assert 0 < need <= lib.mz_wide_static_workspace_floats()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _hip.check(lib.mz_wide_forward_static(C.byref(pl), ptr(blob), ptr(x), x.shape[1], None, ptr(y), N, B,
                                              torch.cuda.current_stream().cuda_stream), "mz_wide_forward_static (captured)")
    ws = torch.empty(need, device="cuda")
    for seed in (1, 2):
        x.copy_(torch.randn(B, 2048, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)))
        graph.replay()
        want = torch.empty(B, N, device="cuda")
        _hip.check(lib.mz_wide_forward(C.byref(pl), ptr(blob), ptr(x), x.shape[1], None, ptr(want), N, B, ptr(ws), need,
                                       torch.cuda.current_stream().cuda_stream), "mz_wide_forward")
        torch.cuda.synchronize()
        assert torch.equal(y, want)
