"""The streamed-weights two-group conv (csrc/conv3x3_pps.hip) against the general kernel on the frame's multi-chunk 3x3 shapes, same process,
alternating, HIP events around 20 back-to-back launches on random data.   python tools/bench_pps.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d

dev = "cuda:0"
ops.set_precision("f16")
SHAPES = [((64, 64), 64, 1088, 1920, 0), ((64, 64), 64, 544, 960, 2), ((64, 32), 64, 1088, 1920, 0), ((64, 8), 64, 1088, 1920, 0),
          ((128,), 128, 272, 480, 1), ((96,), 96, 544, 960, 2), ((96,), 96, 272, 480, 2), ((64,), 128, 272, 480, 2)]
for segs, cout, H, W, act in SHAPES:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, 3, 1, 1).to(dev)
    ins = [ops.Act(torch.randn((1, H, W, c), device=dev).to(torch.float16)) for c in segs]
    out = ops.new_act(1, H, W, cout, torch.float16, dev)
    flop = 2.0 * H * W * cout * sum(segs) * 9
    res = {}
    for rnd in range(3):
        for opt in (1, 0):
            ops.set_option("pps", opt)
            for _ in range(3):
                m.run(ins, out, act=act, slope=0.01)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                m.run(ins, out, act=act, slope=0.01)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(opt, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    ops.set_option("pps", 1)
    a, b = min(res[1]), min(res[0])
    print(f"3x3 {'+'.join(map(str, segs))} -> {cout} @{H}x{W} act {act}: streamed-weights {a:7.1f} us ({flop / a / 1e6:6.0f} TFLOP/s = {flop / a / 1e6 / 2500:.3f} of peak)   "
          f"general {b:7.1f} us ({flop / b / 1e6 / 2500:.3f})   rounds {['%.0f/%.0f' % (x, y) for x, y in zip(res[1], res[0])]}", flush=True)

# ---- fp32 (parity) mode: the dominant 3x3 64 -> 64 shape and its epilogue variants ----
ops.set_precision("fp32")
for name, segs, pre, act, nres in (("plain", (64,), 0, 0, 0), ("pre+act", (64,), 1, 1, 0), ("+res", (64,), 0, 0, 1), ("+2 res", (64,), 0, 0, 2), ("64+64", (64, 64), 0, 0, 0), ("32 (two chunks)", (32,), 0, 2, 0), ("32+32", (32, 32), 0, 2, 0)):
    H, W = 1088, 1920
    torch.manual_seed(0)
    m = Conv2d(sum(segs), 64, 3, 1, 1).to(dev)
    ins = [ops.Act(torch.randn((1, H, W, c), device=dev)) for c in segs]
    out = ops.new_act(1, H, W, 64, torch.float32, dev)
    rs = [ops.Act(torch.randn((1, H, W, 64), device=dev)) for _ in range(nres)]
    kw = dict(pre_act=pre, act=act, slope=0.01)
    if nres >= 1: kw["res1"] = rs[0]
    if nres == 2: kw["res2"] = rs[1]
    flop = 2.0 * H * W * 64 * sum(segs) * 9
    res = {}
    for rnd in range(3):
        for opt in (1, 0):
            ops.set_option("pps_f32", 2 if (opt and nres == 2) else opt)      # (2: historical value for the two-residual variant, now the same as 1)
            for _ in range(2):
                m.run(ins, out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m.run(ins, out, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(opt, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    ops.set_option("pps_f32", 1)
    a, b = min(res[1]), min(res[0])
    print(f"fp32 3x3 {'+'.join(map(str, segs))} -> 64 @{H}x{W} {name}: streamed-weights {a:7.1f} us ({flop / a / 1e6:6.1f} TFLOP/s = {flop / a / 1e6 / 157.3:.3f} of the fp32 peak)   "
          f"general {b:7.1f} us ({flop / b / 1e6 / 157.3:.3f})   rounds {['%.0f/%.0f' % (x, y) for x, y in zip(res[1], res[0])]}", flush=True)
