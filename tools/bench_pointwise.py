"""Full-resolution pointwise kernels of a 1080p P-frame, stand-alone (warp of the 64-channel feature, 2x2 pooling, PixelShuffle(4) of the
inter-layer feature): time and bytes moved.   python tools/bench_pointwise.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402

dev = "cuda:0"
H, W = 1088, 1920


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


torch.manual_seed(0)
x = ops.to_act(torch.randn(1, 64, H, W, device=dev), torch.float16)
flow = ops.to_act(torch.randn(1, 2, H, W, device=dev) * 3, torch.float32)
out = ops.new_act(1, H, W, 64, torch.float16, dev)
t = timeit(lambda: ops.warp(x, flow, out))
print(f"warp 64 ch f16 @{H}x{W}: {t:6.1f} us ({(2 * 64 * 2 + 8) * H * W / t / 1e6:4.2f} TB/s of in + out + flow)")
x3 = ops.to_act(torch.randn(1, 3, H, W, device=dev), torch.float32)
o3 = ops.new_act(1, H, W, 3, torch.float32, dev)
t = timeit(lambda: ops.warp(x3, flow, o3))
print(f"warp 3 ch fp32 @{H}x{W}: {t:6.1f} us")
po = ops.new_act(1, H // 2, W // 2, 64, torch.float16, dev)
t = timeit(lambda: ops.pool2x2(x, "avg", po))
print(f"avg-pool 2x2 64 ch f16 @{H}x{W}: {t:6.1f} us ({1.25 * 64 * 2 * H * W / t / 1e6:4.2f} TB/s)")
xn = torch.randn(1, 64 * 16, H // 4, W // 4, device=dev)
so = ops.new_act(1, H, W, 64, torch.float16, dev)
t = timeit(lambda: ops.pixel_shuffle_nchw(xn, 4, so))
print(f"PixelShuffle(4) nchw fp32 -> nhwc f16 64 ch @{H}x{W}: {t:6.1f} us ({64 * 6 * H * W / t / 1e6:4.2f} TB/s)")
