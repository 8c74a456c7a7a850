"""Times the shipped representation network h (src/config/config_car_racing.yaml:57-131: 32-frame
history of 3x96x96 -> latent [256, 6, 6]) for G games per call: the HIP path (muzero_amd.conv_repr)
and, beside it, the torch module on the GPU with per-game statistics (rocBLAS / MIOpen).

    python tools/bench_conv_repr.py [--games 256] [--iters 5] [--no-torch]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def shipped_h():
    from muzero_amd.config import ConvLayerConfig, PoolLayerConfig, RepresentationNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import RepresentationNetwork
    cv = lambda c: ConvLayerConfig(out_channels=c, kernel_size=2, stride=2, padding=0, activation="relu")  # noqa: E731
    rb = lambda c: ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu", pool_kernel_size=1, pool_stride=1)  # noqa: E731
    torch.manual_seed(0)
    return RepresentationNetwork((3, 96, 96), (256, 6, 6), RepresentationNetworkConfig(
        history_length=32, downsample=[cv(128), rb(128), rb(128), cv(256), rb(256), rb(256), rb(256),
                                       PoolLayerConfig(kernel_size=2, stride=2, pool_type="max")], res_net=[rb(3)]))


def flops_per_game():
    f = 48 * 48 * 128 * 128 * 4 * 2            # conv 2x2/2 128 -> 128
    f += 4 * 48 * 48 * 128 * 128 * 9 * 2       # two blocks of two 3x3 convs at 48x48
    f += 24 * 24 * 256 * 128 * 4 * 2           # conv 2x2/2 128 -> 256
    f += 6 * 24 * 24 * 256 * 256 * 9 * 2       # three blocks at 24x24
    f += 12 * 12 * (3 * 256 * 9 + 3 * 3 * 9 + 3 * 256) * 2 + 12 * 12 * 256 * 3 * 2
    return f


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--fp32", action="store_true", help="keep every convolution on the fp32 matrix instruction")
    args = ap.parse_args()
    from muzero_amd.conv_repr import ConvRepresentation
    from muzero_amd.neural_networks.per_call import per_call_statistics
    G = args.games
    h = shipped_h()
    x = torch.rand(G, 128, 96, 96, device="cuda") * 255.0
    run = ConvRepresentation(h, fast_convolutions=not args.fp32)
    t_hip = timed(lambda: run(x), args.iters)
    out = {"games": G, "convolutions": "fp32 MFMA" if args.fp32 else "split-bf16 MFMA (6 products)", "hip_ms": t_hip * 1e3, "flop_per_game": flops_per_game(),
           "hip_tflops": flops_per_game() * G / t_hip / 1e12, "fp32_mfma_peak_tflops": 157.3}
    if not args.no_torch:
        import copy
        m = copy.deepcopy(h).cuda()

        def torch_call():
            with torch.no_grad(), per_call_statistics(m):
                return m(x)
        t_torch = timed(torch_call, args.iters)
        out.update(torch_ms=t_torch * 1e3, torch_tflops=flops_per_game() * G / t_torch / 1e12)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
