"""The arithmetic the matrix kernels use for float32 products, restated in numpy: an operand v is held as
hi = fp16(v), lo = fp16((v - hi) * 2^11), and x * w is evaluated as
    2^-11 * (x_hi * (w_hi * 2^11) + x_hi * w_lo + x_lo * w_hi)
in float32 accumulation (csrc/mz_mlp_device.h, csrc/mz_conv3.hip, k_merge_conv_bf in csrc/mz_tower.hip).
Checked here, without a GPU: the two planes carry 22 significant bits over the magnitudes the networks see, the
three-term product is within 2^-20 of the exact one, a long dot product stays far inside the 1e-5 the network
outputs are held to, and values beyond fp16's range turn into infinities (loud), not into wrong finite numbers."""
import numpy as np

SCALE = np.float32(2048.0)


def split2(v):
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * SCALE).astype(np.float16)
    return hi, lo


def product3(x, w):
    """The three fp16 products of one (x, w) pair as the matrix instruction forms them (exact fp16 x fp16
    products, float32 sum), scaled back."""
    xh, xl = split2(x)
    wh, wl = split2(w)
    f = np.float32
    ws = (wh.astype(f) * SCALE).astype(np.float16)  # exact exponent shift
    assert np.array_equal(ws.astype(f), wh.astype(f) * SCALE)
    acc = xl.astype(f) * wh.astype(f)
    acc = acc + xh.astype(f) * wl.astype(f)
    acc = acc + xh.astype(f) * ws.astype(f)
    return acc * f(1.0 / 2048.0)


def test_two_planes_hold_22_bits():
    rng = np.random.default_rng(0)
    for mag in (1e-3, 0.03, 1.0, 50.0, 255.0, 3.0e4):
        v = (rng.standard_normal(100000) * mag).astype(np.float32)
        v = v[np.abs(v) < 65000]
        hi, lo = split2(v)
        back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
        v64 = v.astype(np.float64)
        # 11 + 11 bits, each rounded to nearest; values in fp16's subnormal range keep an absolute accuracy instead
        assert (np.abs(back - v64) <= 2.0 ** -21 * np.abs(v64) + 2.0 ** -36).all(), mag


def test_small_values_keep_absolute_accuracy():
    # below fp16's normal range hi is a subnormal; the scaled lo plane still resolves the rest
    v = np.float32([1e-6, 3e-7, 5e-5, 6.1e-5, 1e-9])
    hi, lo = split2(v)
    back = hi.astype(np.float64) + lo.astype(np.float64) / 2048.0
    assert np.abs(back - v.astype(np.float64)).max() <= 2.0 ** -36


def test_three_term_product():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(200000) * 3.0).astype(np.float32)
    w = (rng.standard_normal(200000) * 0.05).astype(np.float32)
    got = product3(x, w).astype(np.float64)
    want = x.astype(np.float64) * w.astype(np.float64)
    # two operand roundings of 2^-22 each, the dropped lo * lo term, float32 sums; an absolute floor for tiny operands
    assert (np.abs(got - want) <= 2.0 ** -20 * np.abs(want) + 1e-10).all()


def test_long_dot_product_against_float64():
    rng = np.random.default_rng(2)
    K = 1152  # a 3 x 3 convolution over 128 channels
    x = np.maximum(rng.standard_normal((64, K)), 0).astype(np.float32)
    w = (rng.standard_normal((K, 32)) * 0.03).astype(np.float32)
    acc = np.zeros((64, 32), np.float32)
    for k in range(K):  # float32 accumulation, one reduction index at a time
        acc += product3(x[:, k:k + 1], w[k:k + 1, :])
    want = x.astype(np.float64) @ w.astype(np.float64)
    plain = np.zeros((64, 32), np.float32)
    for k in range(K):
        plain += x[:, k:k + 1] * w[k:k + 1, :]
    scale = np.abs(want).max()
    ours = np.abs(acc - want).max() / scale
    theirs = np.abs(plain - want).max() / scale
    assert ours <= 1e-6 and ours <= 4 * theirs + 2e-7, (ours, theirs)


def test_out_of_range_operands_are_loud():
    hi, lo = split2(np.float32([1.0e5, -7.0e4]))
    assert np.isinf(hi.astype(np.float32)).all()
    with np.errstate(invalid="ignore", over="ignore"):
        p = product3(np.float32([1.0e5]), np.float32([0.5]))
    assert not np.isfinite(p).any()
