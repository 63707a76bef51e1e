"""Micro-benchmark of the conv kernel on the shapes that dominate a 1080p P-frame.
    python tools/bench_conv.py [--precision f16|fp32] [--iters 20]
Prints per-shape average launch time (HIP events on the launch stream) and achieved TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

SHAPES = [  # (name, segs, cout, k, stride, H, W, kwargs)
    ("3x3 64->64 @1/1", (64,), 64, 3, 1, 1088, 1920, {}),
    ("3x3 64->64 @1/1 +res +lrelu pre", (64,), 64, 3, 1, 1088, 1920, {"res": True, "pre": True}),
    ("3x3 128->128 @1/4", (128,), 128, 3, 1, 272, 480, {}),
    ("3x3 64+64->64 @1/1", (64, 64), 64, 3, 1, 1088, 1920, {}),
    ("3x3 96->96 @1/2", (96,), 96, 3, 1, 544, 960, {}),
    ("1x1 64->64 @1/1", (64,), 64, 1, 1, 1088, 1920, {}),
    ("3x3 64->32 @1/1", (64,), 32, 3, 1, 1088, 1920, {}),
    ("3x3 64->3 @1/1", (64,), 3, 3, 1, 1088, 1920, {}),
    ("3x3 3->64 @1/1 (fp32 in)", (3,), 64, 3, 1, 1088, 1920, {"f32in": True}),
    ("7x7 32->64 @1/1", (32,), 64, 7, 1, 1088, 1920, {}),
    ("5x5s2 70->96 @1/1", (70,), 96, 5, 2, 1088, 1920, {}),
    ("3x3 64->64 @1/16", (64,), 64, 3, 1, 68, 120, {}),
    ("3x3 96->12 @1/16", (96,), 12, 3, 1, 68, 120, {}),
    ("5x5s2 8+64+8->96 @1/1", (8, 64, 8), 96, 5, 2, 1088, 1920, {}),
    ("5x5s2 96->96 @1/2", (96,), 96, 5, 2, 544, 960, {}),
    ("3x3 64+64->64 @1/1", (64, 64), 64, 3, 1, 1088, 1920, {}),
    ("3x3 128->128 @1/4 +res", (128,), 128, 3, 1, 272, 480, {"res": True}),
    ("7x7 32->64 @1/1", (32,), 64, 7, 1, 1088, 1920, {}),
    ("1x1 96->96 @1/2", (96,), 96, 1, 1, 544, 960, {}),
    ("1x1 96->96 @1/2 +res", (96,), 96, 1, 1, 544, 960, {"res": True}),
    ("1x1 128->128 @1/2", (128,), 128, 1, 1, 544, 960, {}),
    ("1x1 64->64 @1/2", (64,), 64, 1, 1, 544, 960, {}),
    ("1x1 32->32 @1/1", (32,), 32, 1, 1, 1088, 1920, {}),
    ("1x1 32+32->64 @1/1 gate+res", (32, 32), 64, 1, 1, 1088, 1920, {"res": True, "gate": True}),
    ("1x1 64->32 @1/1 gate+res", (64,), 32, 1, 1, 1088, 1920, {"res": True, "gate": True}),
    ("1x1 64+64->64 @1/2 gate+res", (64, 64), 64, 1, 1, 544, 960, {"res": True, "gate": True}),
]
SHAPES += [
    ("3x3 32+32->64 @1/1 lrelu", (32, 32), 64, 3, 1, 1088, 1920, {"act": True}),
    ("3x3 64+32->64 @1/1", (64, 32), 64, 3, 1, 1088, 1920, {}),
    ("3x3 64+8->64 @1/1", (64, 8), 64, 3, 1, 1088, 1920, {}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1, help="run only shape #i")
    args = ap.parse_args()
    ops.set_precision(args.precision)
    dev = "cuda:0"
    st = ops.RT.storage
    for idx, (name, segs, cout, k, s, H, W, kw) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        torch.manual_seed(0)
        m = Conv2d(sum(segs), cout, k, s, padding=k // 2).to(dev)
        ins = [ops.Act(torch.randn((1, H, W, c), device=dev).to(torch.float32 if kw.get("f32in") else st)) for c in segs]
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = ops.new_act(1, Ho, Wo, cout, st, dev)
        res = ops.Act(torch.randn((1, Ho, Wo, cout), device=dev).to(st)) if kw.get("res") else None
        ep = dict(res1=res, pre_act=N.ACT_LRELU if kw.get("pre") else N.ACT_NONE, act=N.ACT_LRELU if (kw.get("pre") or kw.get("act")) else N.ACT_NONE)
        if kw.get("gate"):
            ep["gate"] = torch.rand((1, cout), device=dev)
        for _ in range(3):
            m.run(ins, out, **ep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            m.run(ins, out, **ep)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        flop = 2.0 * Ho * Wo * cout * sum(segs) * k * k
        es = 2 if st == torch.float16 else 4
        gbytes = (H * W * sum(segs) * (4 if kw.get("f32in") else es) + Ho * Wo * cout * es * (2 if kw.get("res") else 1)) / 1e9
        print(f"{name:36s} {ms*1e3:9.1f} us  {flop/ms/1e9:8.1f} TFLOP/s  {gbytes/ms*1e3:7.0f} GB/s (algorithmic)", flush=True)


if __name__ == "__main__":
    main()
