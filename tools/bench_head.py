"""Narrow output heads (Cout <= 4) of a 1080p P-frame: the dot-product kernel (csrc/conv_head.hip) against the MFMA kernels that took these
launches before ("head" = 0: conv3x3_pp HEAD variant / generic kernel).   python tools/bench_head.py"""
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


for segs, cout, k, H, W, res in [((64,), 3, 3, 1088, 1920, False), ((32,), 3, 3, 1088, 1920, False), ((64,), 2, 3, 1088, 1920, True), ((64, 64), 3, 3, 1088, 1920, False),
                                 ((16,), 2, 7, 1088, 1920, True), ((16,), 2, 7, 544, 960, True), ((16,), 2, 7, 272, 480, True)]:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, k, 1, padding=k // 2).to(dev)
    xs = [ops.to_act(torch.randn(1, c, H, W, device=dev), torch.float16) for c in segs]
    out = ops.new_act(1, H, W, cout, torch.float32, dev)
    kw = dict(res1=ops.to_act(torch.randn(1, cout, H, W, device=dev), torch.float32)) if res else {}
    t_tap = timeit(lambda: m.run(xs, out, **kw))         # round 5: the tap-GEMM kernel (csrc/conv_headtap.hip) takes the 3x3 shapes first
    tap_took = ops.get_counter("conv_headtap") > 0
    ops.set_option("headtap", 0)
    ops.set_option("head", 2)                            # 2: every shape the dot-product kernel covers
    t_head = timeit(lambda: m.run(xs, out, **kw))
    ops.set_option("head", 0)
    t_mfma = timeit(lambda: m.run(xs, out, **kw))
    ops.set_option("head", 1)
    ops.set_option("headtap", 1)
    nbytes = sum(segs) * 2 * H * W
    tap = f"tap-GEMM kernel {t_tap:7.1f} us ({nbytes / t_tap / 1e6:5.2f} TB/s of input)   " if k == 3 and tap_took else ""
    print(f"{k}x{k} {'+'.join(map(str, segs))}->{cout} @{H}x{W}{' + res' if res else ''}: {tap}dot-product kernel {t_head:7.1f} us ({nbytes / t_head / 1e6:5.2f} TB/s of input)   MFMA kernels {t_mfma:7.1f} us", flush=True)
