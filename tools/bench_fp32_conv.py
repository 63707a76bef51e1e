"""fp32 mode's dominant conv (3x3 s1 64->64 @1088x1920, generic kernel) with and without the input-patch prefetch, plus the other
fp32 shapes of a frame.   python tools/bench_fp32_conv.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

OPT = sys.argv[1] if len(sys.argv) > 1 else "patch_prefetch"          # option to A/B (patch_prefetch | wg_skew)
MODE = sys.argv[2] if len(sys.argv) > 2 else "fp32"
ops.set_precision(MODE)
DT = torch.float32 if MODE == "fp32" else torch.float16
dev = "cuda:0"
H, W = 1088, 1920
torch.manual_seed(0)


def run(name, cin, cout, k, Hh, Ww, kw=None, n=10):
    xs = [ops.Act(torch.randn((1, Hh, Ww, c), device=dev).to(DT)) for c in cin]
    conv = Conv2d(sum(cin), cout, k, 1, k // 2).to(dev)
    out = ops.new_act(1, Hh, Ww, cout, DT, dev)
    kw = dict(kw or {})
    if kw.pop("res", False):
        kw["res1"] = ops.Act(torch.randn((1, Hh, Ww, cout), device=dev).to(DT))
    res = {}
    for pf in (0, 1, 0, 1, 1, 0):
        ops.set_option(OPT, pf)
        for _ in range(2):
            conv.run(xs, out, out_dtype=DT, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            conv.run(xs, out, out_dtype=DT, **kw)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(pf, []).append(e0.elapsed_time(e1) / n * 1e3)
        if pf == 0:
            ref = out.buf.clone()
        else:
            assert torch.equal(ref, out.buf), f"{name}: patch prefetch changed the fp32 result"
    flop = 2.0 * Hh * Ww * sum(cin) * cout * k * k
    best = {p: min(v) for p, v in res.items()}
    print("      raw us:", {k: [round(x, 1) for x in v] for k, v in res.items()})
    peak = 157.3 if MODE == "fp32" else 2500.0
    print(f"{name:36s} {OPT}=0 {best[0]:8.1f} us ({flop / best[0] / 1e6:6.1f} TFLOP/s)   {OPT}=1 {best[1]:8.1f} us ({flop / best[1] / 1e6:6.1f} TFLOP/s = "
          f"{flop / best[1] / 1e6 / peak:.2f} of the {MODE} MFMA peak)", flush=True)


run("3x3 64->64 @1088x1920", [64], 64, 3, H, W)
run("3x3 64->64 + act + res", [64], 64, 3, H, W, dict(act=N.ACT_LRELU, pre_act=N.ACT_LRELU, slope=0.01, res=True))
run("3x3 64+64->64 @1088x1920", [64, 64], 64, 3, H, W)
run("3x3 64->32 @1088x1920", [64], 32, 3, H, W)
run("3x3 96->96 @544x960", [96], 96, 3, H // 2, W // 2)
run("3x3 128->128 @272x480", [128], 128, 3, H // 4, W // 4)
run("7x7 64->32 @1088x1920", [64], 32, 7, H, W)
ops.set_option(OPT, 1)
ops.set_precision("fp32")
