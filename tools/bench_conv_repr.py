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


def flappy_h():
    """src/config/config_flappy_bird.yaml:58-165: 32-frame history of 3x512x288 -> latent [256, 6, 6]."""
    from muzero_amd.config import ConvLayerConfig, PoolLayerConfig, RepresentationNetworkConfig, ResBlockConfig
    from muzero_amd.neural_networks import RepresentationNetwork
    cv = lambda c: ConvLayerConfig(out_channels=c, kernel_size=2, stride=2, padding=0, activation="relu")  # noqa: E731
    rb = lambda c: ResBlockConfig(out_channels=c, kernel_size=3, stride=1, padding=1, activation="relu", pool_kernel_size=1, pool_stride=1)  # noqa: E731
    avg = lambda: PoolLayerConfig(kernel_size=2, stride=2, pool_type="avg")  # noqa: E731
    torch.manual_seed(0)
    return RepresentationNetwork((3, 512, 288), (256, 6, 6), RepresentationNetworkConfig(
        history_length=32, downsample=[cv(128), rb(128), rb(128), cv(256), rb(256), rb(256), rb(256), avg(),
                                       rb(256), rb(256), rb(256), avg()], res_net=[rb(3)]))


def flappy_flops_per_game():
    f = 256 * 144 * 128 * 128 * 4 * 2           # conv 2x2/2 128 -> 128
    f += 4 * 256 * 144 * 128 * 128 * 9 * 2      # two blocks of two 3x3 convs at 256x144
    f += 128 * 72 * 256 * 128 * 4 * 2           # conv 2x2/2 128 -> 256
    f += 6 * 128 * 72 * 256 * 256 * 9 * 2       # three blocks at 128x72
    f += 6 * 64 * 36 * 256 * 256 * 9 * 2        # three blocks at 64x36
    f += 32 * 18 * (3 * 256 * 9 + 3 * 3 * 9 + 3 * 256) * 2 + 32 * 18 * 256 * 3 * 2
    return f


def flappy_time(G: int, iters: int = 2):
    """Milliseconds per call of FlappyBird's h for G games on the HIP convolution path, with its matrix rates."""
    from muzero_amd.conv_repr import ConvRepresentation
    run = ConvRepresentation(flappy_h())
    x = torch.rand(G, 128, 512, 288, device="cuda") * 255.0
    s = timed(lambda: run(x), iters)
    del run, x
    torch.cuda.empty_cache()
    tf = flappy_flops_per_game() * G / s / 1e12
    return {"games": G, "ms_per_move": s * 1e3, "flop_per_game": flappy_flops_per_game(), "tflops": tf,
            "fp32_mfma_peak_tflops": 157.3, "f16_executed_tflops": 3 * tf, "f16_mfma_peak_tflops": 2500.0,
            "f16_executed_frac": 3 * tf / 2500.0,
            "kernel": "as the CarRacing h; the 144- and 72-pixel-wide layers on the staged-range kernel with 5 and 4 slot groups"}


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
    ap.add_argument("--flappy", action="store_true", help="FlappyBird's h (3x512x288 frames) instead; --games defaults to 16")
    args = ap.parse_args()
    if args.flappy:
        print(json.dumps(flappy_time(16 if args.games == 256 else args.games, args.iters)))
        return
    from muzero_amd.conv_repr import ConvRepresentation
    from muzero_amd.neural_networks.per_call import per_call_statistics
    G = args.games
    h = shipped_h()
    x = torch.rand(G, 128, 96, 96, device="cuda") * 255.0
    run = ConvRepresentation(h, fast_convolutions=not args.fp32)
    t_hip = timed(lambda: run(x), args.iters)
    out = {"games": G, "convolutions": "fp32 MFMA" if args.fp32 else "two-plane fp16 MFMA (3 products)", "hip_ms": t_hip * 1e3, "flop_per_game": flops_per_game(),
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
