"""Pointwise convs of a 1080p P-frame: row-granular global access through an LDS transposition ("conv1x1_rows" = 1) against lanes
reading / writing 32-byte pieces of 32 different rows per instruction (0).   python tools/bench_1x1.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
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


for segs, cout, H, W, res, gate in [((64,), 64, 1088, 1920, False, False), ((64,), 64, 1088, 1920, True, False), ((96,), 96, 544, 960, False, False), ((96,), 96, 544, 960, True, False),
                                    ((32, 32), 64, 1088, 1920, True, True), ((64,), 32, 1088, 1920, True, True), ((64,), 64, 544, 960, False, False), ((64,), 64, 272, 480, True, False)]:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, 1, 1, padding=0).to(dev)
    xs = [ops.to_act(torch.randn(1, c, H, W, device=dev), torch.float16) for c in segs]
    out = ops.new_act(1, H, W, cout, torch.float16, dev)
    kw = dict(act=N.ACT_LRELU, slope=0.1)
    if res:
        kw["res1"] = ops.to_act(torch.randn(1, cout, H, W, device=dev), torch.float16)
    if gate:
        kw["gate"] = torch.rand(1, cout, device=dev)
    res_t = []
    for rnd in range(2):
        for opt in (1, 0):
            ops.set_option("conv1x1_rows", opt)
            res_t.append(timeit(lambda: m.run(xs, out, **kw)))
    ops.set_option("conv1x1_rows", 1)
    nbytes = (sum(segs) + cout * (2 if res else 1)) * 2 * H * W
    print(f"1x1 {'+'.join(map(str, segs))}->{cout} @{H}x{W}{' + res' if res else ''}{' gated' if gate else ''}: rows {res_t[0]:6.1f} / {res_t[2]:6.1f} us ({nbytes / min(res_t[0], res_t[2]) / 1e6:4.2f} TB/s)   "
          f"pieces {res_t[1]:6.1f} / {res_t[3]:6.1f} us", flush=True)
