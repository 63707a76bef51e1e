"""fp32 mode's narrow heads: vector-ALU fma chain (csrc/conv_head_f32.hip) against the MFMA kernel, isolated, at 1088x1920.
python tools/bench_head_f32.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("fp32")
dev = "cuda:0"
H, W = 1088, 1920
for segs, cout, k, act, res in [((64,), 3, 3, 4, False), ((64,), 2, 3, 0, True), ((64, 64), 3, 3, 0, False), ((32,), 3, 3, 0, False), ((16,), 2, 7, 0, True)]:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, k, 1, k // 2).to(dev)
    xs = [ops.Act(torch.randn((1, H, W, c), device=dev)) for c in segs]
    out = ops.new_act(1, H, W, cout, torch.float32, dev)
    kw = dict(act=act, slope=0.01, out_dtype=torch.float32)
    if res:
        kw["res1"] = ops.Act(torch.randn((1, H, W, cout), device=dev))
    t = {}
    for rnd in range(3):
        for on in (1, 0):
            ops.set_option("head_f32", on)
            for _ in range(2):
                m.run(xs, out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                m.run(xs, out, **kw)
            e1.record(); torch.cuda.synchronize()
            t.setdefault(on, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    nfma = H * W * sum(segs) * k * k * cout
    print(f"{k}x{k} {'+'.join(map(str, segs))} -> {cout} act {act} res {int(res)}: fma chain {min(t[1]):7.1f} us ({2e-6 * nfma / min(t[1]):5.1f} TFLOP/s on the vector ALU)   MFMA kernel {min(t[0]):7.1f} us", flush=True)
ops.set_option("head_f32", 1)
