"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every
symbol include/muzero_hip.h declares (no kernel is launched here)."""
import ctypes
import os
import re

from muzero_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "muzero_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/muzero_hip.h but not exported"


def test_binding_covers_header():
    assert set(_declared()) <= set(_hip.exported_symbols())
    lib = _hip.lib()
    assert lib.mz_abi_version() == 3  # 2: one 32-byte record per node; 3: matrix layers packed as two fp16 planes (mz_packed_layer.fmt)


def test_host_only_entry_points():
    """mz_puct_tables and the *_plan calls run on the host: check them without a GPU."""
    import math

    import numpy as np
    sq, lg = np.zeros(64), np.zeros(64)
    assert _hip.lib().mz_puct_tables(_hip.ptr(sq), _hip.ptr(lg), 64) == 0
    for n in (0, 1, 7, 50, 63):
        assert sq[n] == math.sqrt(n)
        assert lg[n] == math.log((n + 19652 + 1) / 19652) + 1.25  # src/search/selection.py:47
    m = _hip.Mlp()
    m.n_layers = 2
    m.layer[0] = _hip.Dense(1, None, 168, 256, 1, 0)
    m.layer[1] = _hip.Dense(1, None, 256, 7, 4, 0)
    plan = _hip.PackedMlp()
    n = _hip.lib().mz_mlp_plan(ctypes.byref(m), 0, ctypes.byref(plan))
    # tiles + padded bias per layer, 4 KiB read slack: both layers run on the matrix cores (K padded to 32, two fp16
    # planes = 4 bytes per weight)
    assert n == 192 * 256 + 256 + 256 * 16 + 16 + 1024
    assert (plan.layer[0].fmt, plan.layer[0].Kp, plan.layer[1].fmt, plan.layer[1].Kp) == (1, 192, 1, 256)
    # a one-output layer behind a wide layer keeps fp32 tiles: it is folded into that layer's epilogue
    m2 = _hip.Mlp()
    m2.n_layers = 2
    m2.layer[0] = _hip.Dense(1, None, 168, 128, 1, 0)
    m2.layer[1] = _hip.Dense(1, None, 128, 1, 0, 0)
    plan2 = _hip.PackedMlp()
    assert _hip.lib().mz_mlp_plan(ctypes.byref(m2), 0, ctypes.byref(plan2)) > 0
    assert (plan2.layer[0].fmt, plan2.layer[1].fmt, plan2.layer[1].Kp) == (1, 0, 128)
    assert (plan.in_features, plan.out_features, plan.max_width) == (168, 7, 256)
    m.layer[1].in_features = 255  # chain mismatch must be rejected with a message
    assert _hip.lib().mz_mlp_plan(ctypes.byref(m), 0, ctypes.byref(plan)) < 0
    assert b"in_features" in _hip.lib().mz_last_error()


def _search_args(A=7, D=8, G=4, S=5, cap=None, depth=1, tab_len=None):
    """Argument structs for mz_search_run with non-NULL dummy pointers: the checks under test run
    on the host before anything is launched, so the pointers are never dereferenced."""
    import ctypes as C
    dummy = 0x1000
    pool = _hip.Pool()
    pool.G, pool.A, pool.cap, pool.D = G, A, (1 + (S + 1) * A) if cap is None else cap, D
    for n in ("node", "n_nodes", "latent"):
        setattr(pool, n, dummy)
    plan = _hip.PackedNets()
    plan.D, plan.A = D, A
    prm = _hip.SearchParams()
    prm.S, prm.depth, prm.tab_len = S, depth, (S + 2) if tab_len is None else tab_len
    for n in ("sqrt_tab", "log_tab", "root_latent", "root_to_play", "eta", "rand_child"):
        setattr(prm, n, dummy)
    return pool, plan, prm, C.c_void_p(dummy)


def _run_search(pool, plan, prm, blob):
    import ctypes as C
    lib = _hip.lib()
    rc = lib.mz_search_run(C.byref(pool), C.byref(plan), blob, C.byref(prm), None, None, None, None, None)
    return rc, lib.mz_last_error().decode()


def test_search_entry_point_rejects_bad_arguments_before_launching():
    """mz_search_run validates on the host and reports through mz_last_error(); nothing here
    reaches a GPU."""
    lib = _hip.lib()
    assert lib.mz_search_run(None, None, None, None, None, None, None, None, None) != 0
    rc, msg = _run_search(*_search_args(A=40))
    assert rc != 0 and "32 actions" in msg
    rc, msg = _run_search(*_search_args(depth=8))
    assert rc != 0 and "model_look_ahead" in msg
    rc, msg = _run_search(*_search_args(cap=10))
    assert rc != 0 and "capacity" in msg
    rc, msg = _run_search(*_search_args(tab_len=3))
    assert rc != 0 and "tab_len" in msg
    pool, plan, prm, blob = _search_args()
    plan.A = 5  # networks built for another action count than the pool
    rc, msg = _run_search(pool, plan, prm, blob)
    assert rc != 0 and "disagree" in msg
    pool, plan, prm, blob = _search_args()
    prm.eta = None
    rc, msg = _run_search(pool, plan, prm, blob)
    assert rc != 0 and "NULL" in msg
