import os, sys, torch
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d
ops.set_precision("f16")
dev, h = "cuda:0", torch.float16
H, W = 1088, 1920
torch.manual_seed(0)
x = ops.Act((torch.randn((1, H, W, 64), device=dev)).to(h))
r1 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
r2 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
conv = Conv2d(64, 64, 3, 1, 1).to(dev)
c32 = Conv2d(64, 32, 3, 1, 1).to(dev)
c1 = Conv2d(32, 64, 3, 1, 1).to(dev)
x32 = ops.Act(torch.randn((1, H, W, 32), device=dev).to(h))
post = Conv2d(64, 64, 1, 1, 0).to(dev)._packed([64], dev)
CASES = [("plain", conv, x, {}), ("act lrelu", conv, x, dict(act=N.ACT_LRELU)), ("pre relu + act relu", conv, x, dict(pre_act=N.ACT_RELU, act=N.ACT_RELU, slope=0.0)),
         ("pre lrelu + act lrelu", conv, x, dict(pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01)),
         ("res1", conv, x, dict(res1=r1)), ("res1 + res2", conv, x, dict(res1=r1, res2=r2)), ("act + res1 + post(1x1)", conv, x, dict(act=N.ACT_LRELU, res1=r1, post=post)),
         ("64->32", c32, x, {}), ("32->64 (1 chunk)", c1, x32, dict(act=N.ACT_LRELU))]
outs = {}
for rnd in range(3):
    for name, m, inp, kw in CASES:
        for opt in (0, 1):
            ops.set_option("pp16", opt)
            out = ops.new_act(1, H, W, m.out_channels, h, dev)
            for _ in range(2):
                m.run(inp, out, out_dtype=h, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            n = 15
            for _ in range(n):
                m.run(inp, out, out_dtype=h, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            outs.setdefault(name, {})[opt] = out.buf.float().clone() if rnd == 0 else outs[name][opt]
            print(f"round {rnd} {name:28s} pp16={opt} {us:7.1f} us", flush=True)
        if rnd == 0:
            a, b = outs[name][0], outs[name][1]
            d = (a - b).abs()
            print(f"   max |32x32 - 16x16| = {float(d.max()):.3e} (scale {float(a.abs().max()):.2f}), differing elements {float((d > 0).float().mean()):.2e}, finite {bool(torch.isfinite(b).all())}", flush=True)
ops.set_option("pp16", 1)
