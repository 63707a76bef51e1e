"""8-channel-input convs of a 1080p P-frame: the x-tap-packing kernel (csrc/conv_cin8.hip) against the kernels that took these launches before
("cin8" = 0: general kernel / weight-stationary 3x3 kernel).   python tools/bench_cin8.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"


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


for cout, k, H, W, act in [(32, 7, 1088, 1920, 1), (32, 7, 544, 960, 1), (32, 7, 272, 480, 1), (32, 7, 136, 240, 1), (64, 3, 1088, 1920, 0), (64, 3, 544, 960, 0)]:
    torch.manual_seed(0)
    m = Conv2d(8, cout, k, 1, padding=k // 2).to(dev)
    x = ops.to_act(torch.randn(1, 8, H, W, device=dev), torch.float16)
    out = ops.new_act(1, H, W, cout, torch.float16, dev)
    n0 = ops.get_counter("conv_cin8")
    t_new = timeit(lambda: m.run([x], out, act=act))
    took = ops.get_counter("conv_cin8") > n0
    ops.set_option("cin8", 0)
    t_old = timeit(lambda: m.run([x], out, act=act))
    ops.set_option("cin8", 1)
    nbytes = (8 + cout) * 2 * H * W
    print(f"{k}x{k} 8->{cout} @{H}x{W}: cin8 kernel {'' if took else '(NOT TAKEN) '}{t_new:7.1f} us ({nbytes / t_new / 1e6:5.2f} TB/s in + out, {2 * H * W * k * k * 8 * cout / t_new / 1e6:6.1f} TFLOP/s)   before {t_old:7.1f} us", flush=True)
