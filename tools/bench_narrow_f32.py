"""fp32 launches with 2- / 3-channel segments: general kernel with / without the all-zero MFMAs skipped ("skip_zero_k").
python tools/bench_narrow_f32.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("fp32")
dev = "cuda:0"
H, W = 1088, 1920
for segs, cout, k, stride in [((3,), 64, 3, 1), ((2, 3), 64, 3, 1), ((3, 2, 3), 64, 5, 2), ((3, 3, 2), 32, 7, 1), ((64, 3), 64, 3, 1), ((3, 64, 3), 96, 5, 2)]:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, k, stride, k // 2).to(dev)
    xs = [ops.Act(torch.randn((1, H, W, c), device=dev)) for c in segs]
    t = {}
    for rnd in range(3):
        for on in (1, 0):
            ops.set_option("skip_zero_k", on)
            for _ in range(2):
                m.run(xs, out_dtype=torch.float32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                m.run(xs, out_dtype=torch.float32)
            e1.record(); torch.cuda.synchronize()
            t.setdefault(on, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{k}x{k} s{stride} {'+'.join(map(str, segs))} -> {cout}: zero k slots skipped {min(t[1]):7.1f} us   all MFMAs {min(t[0]):7.1f} us", flush=True)
ops.set_option("skip_zero_k", 1)
