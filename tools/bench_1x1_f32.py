"""fp32 mode's pointwise convs: streaming kernel (csrc/conv1x1_f32.hip) against the general kernel, isolated, per shape of a 1080p frame.
python tools/bench_1x1_f32.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("fp32")
dev = "cuda:0"
SHAPES = [(64, 64, 1088, 1920, 0, True), (64, 64, 1088, 1920, 2, False), (96, 96, 544, 960, 0, True), (96, 96, 544, 960, 2, False),
          (64, 64, 544, 960, 0, True), (96, 96, 272, 480, 0, True), (96, 96, 272, 480, 2, False), (64, 64, 272, 480, 0, True),
          (96, 96, 136, 240, 0, True), (64, 64, 136, 240, 2, False)]
for cin, cout, H, W, act, res in SHAPES:
    torch.manual_seed(0)
    m = Conv2d(cin, cout, 1, 1, 0).to(dev)
    x = ops.Act(torch.randn((1, H, W, cin), device=dev))
    out = ops.new_act(1, H, W, cout, torch.float32, dev)
    kw = dict(act=act, slope=0.01, out_dtype=torch.float32)
    if res:
        kw["res1"] = ops.Act(torch.randn((1, H, W, cout), device=dev))
    nbytes = 4.0 * H * W * (cin + cout * (2 if res else 1))
    t = {}
    for rnd in range(3):
        for on in (1, 0):
            ops.set_option("conv1x1_f32", on)
            for _ in range(3):
                m.run([x], out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                m.run([x], out, **kw)
            e1.record(); torch.cuda.synchronize()
            t.setdefault(on, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    a, b = min(t[1]), min(t[0])
    print(f"1x1 {cin} -> {cout} @{H}x{W} act {act} res {int(res)}: streaming {a:7.1f} us ({nbytes / a / 1e6:5.2f} TB/s)   general {b:7.1f} us ({nbytes / b / 1e6:5.2f} TB/s)", flush=True)
ops.set_option("conv1x1_f32", 1)
