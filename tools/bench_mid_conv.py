"""Per-launch time (HIP-graph replay, no Python/launch overhead) of mid-resolution convs of the analysis/synthesis transforms.
    python tools/bench_mid_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d

SHAPES = [  # (name, cin, cout, k, H, W, act, res)
    ("1x1 96->96 @1/8 lrelu", 96, 96, 1, 136, 240, N.ACT_LRELU, False),
    ("1x1 96->96 @1/8 +res", 96, 96, 1, 136, 240, N.ACT_NONE, True),
    ("1x1 96->96 @1/4 lrelu", 96, 96, 1, 272, 480, N.ACT_LRELU, False),
    ("1x1 96->96 @1/4 +res", 96, 96, 1, 272, 480, N.ACT_NONE, True),
    ("1x1 64->64 @1/8 lrelu", 64, 64, 1, 136, 240, N.ACT_LRELU, False),
    ("1x1 64->64 @1/4 +res", 64, 64, 1, 272, 480, N.ACT_NONE, True),
    ("3x3 96->96 @1/8 lrelu", 96, 96, 3, 136, 240, N.ACT_LRELU, False),
    ("3x3 96->96 @1/4 lrelu", 96, 96, 3, 272, 480, N.ACT_LRELU, False),
    ("3x3 64->64 @1/4 lrelu", 64, 64, 3, 272, 480, N.ACT_LRELU, False),
    ("3x3 64->64 @1/8 lrelu", 64, 64, 3, 136, 240, N.ACT_LRELU, False),
]
dev = "cuda:0"
ops.set_precision("f16")
NL = 20
for name, cin, cout, k, H, W, act, res in SHAPES:
    torch.manual_seed(0)
    m = Conv2d(cin, cout, k, 1, k // 2).to(dev)
    x = ops.Act(torch.randn((1, H, W, cin), device=dev).to(torch.float16))
    out = ops.new_act(1, H, W, cout, torch.float16, dev)
    r = ops.Act(torch.randn((1, H, W, cout), device=dev).to(torch.float16)) if res else None
    for _ in range(3):
        m.run(x, out, act=act, res1=r)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(NL):
            m.run(x, out, act=act, res1=r)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * NL) * 1e3
    mb = (H * W * (cin + cout * (2 if res else 1)) * 2) / 1e6
    print(f"{name:26s} {us:7.1f} us   {mb:6.1f} MB  {mb/us*1e-3:6.2f} TB/s  {2.0*H*W*cin*cout*k*k/us/1e6:7.1f} TF/s", flush=True)
