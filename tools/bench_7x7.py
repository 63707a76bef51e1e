"""ME_Spynet's inner 7x7 convs of a 1080p P-frame: the row-reuse kernel (csrc/conv7x7_rr.hip) against the general kernel ("rr7" = 0).
   python tools/bench_7x7.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for cin, cout in [(64, 32), (32, 64), (32, 16)]:
    for H, W in [(1088, 1920), (544, 960), (272, 480)]:
        torch.manual_seed(0)
        m = Conv2d(cin, cout, 7, 1, padding=3).to(dev)
        x = ops.to_act(torch.randn(1, cin, H, W, device=dev), torch.float16)
        out = ops.new_act(1, H, W, cout, torch.float16, dev)
        n0 = ops.get_counter("conv7x7_rr")
        t_new = timeit(lambda: m.run([x], out, act=1))
        took = ops.get_counter("conv7x7_rr") > n0
        ops.set_option("rr7", 0)
        t_old = timeit(lambda: m.run([x], out, act=1))
        ops.set_option("rr7", 1)
        fl = 2 * H * W * 49 * cin * cout
        print(f"7x7 {cin}->{cout} @{H}x{W}: row-reuse kernel {'' if took else '(NOT TAKEN) '}{t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TFLOP/s = {fl / t_new / 1e6 / 2500:.3f} of peak)   general kernel {t_old:7.1f} us ({fl / t_old / 1e6 / 2500:.3f})", flush=True)
