"""Variants of the dominant launch (3x3 s1 64->64 @1088x1920, f16) on an idle GPU: which epilogue / staging option costs what.
    python tools/bench_pp.py [--iters 20]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); args = ap.parse_args()
ops.set_precision("f16")
dev = "cuda:0"
H, W = 1088, 1920
h = torch.float16
x = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
r1 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
r2 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
out = ops.new_act(1, H, W, 64, h, dev)
conv = Conv2d(64, 64, 3, 1, 1).to(dev)
c1 = Conv2d(64, 64, 1, 1, 0).to(dev)
post = c1._packed([64], dev)
CASES = [("plain", {}), ("act lrelu", dict(act=N.ACT_LRELU)), ("pre relu + act relu", dict(pre_act=N.ACT_RELU, act=N.ACT_RELU, slope=0.0)),
         ("pre lrelu + act lrelu", dict(pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01)), ("pre lrelu", dict(pre_act=N.ACT_LRELU, slope=0.01)),
         ("res1", dict(res1=r1)), ("res1 + res2", dict(res1=r1, res2=r2)), ("act + res1 + post(1x1)", dict(act=N.ACT_LRELU, res1=r1, post=post))]
for name, kw in CASES:
    for _ in range(3):
        conv.run(x, out, out_dtype=h, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(args.iters):
        conv.run(x, out, out_dtype=h, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    nb = (2 + ("res1" in kw) + ("res2" in kw)) * H * W * 64 * 2
    print(f"{name:28s} {us:7.1f} us  {154.0e9 / us / 1e6:6.3f} PFLOP/s  {nb / us / 1e6:5.2f} TB/s algorithmic", flush=True)
