"""A/B of the streamed-weights kernel's tile-invariant addressing (option pps_oneseg, csrc/conv3x3_pps.hip): fp32 3x3 64 -> 64 @1088x1920 (parity
mode's dominant launches: plain, pre + act, one / two residuals) and the f16 128 -> 128 split @272x480, option off / on interleaved; outputs compared
bit for bit.    python tools/bench_pps_oneseg.py [--rounds 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=3); args = ap.parse_args()
dev = "cuda:0"
best, ref = {}, {}
for mode, dt, H, W, cin, cout, n in (("fp32", torch.float32, 1088, 1920, 64, 64, 8), ("f16", torch.float16, 272, 480, 128, 128, 30)):
    ops.set_precision(mode)
    torch.manual_seed(0)
    x = ops.Act(torch.randn((1, H, W, cin), device=dev).to(dt)); r1 = ops.Act(torch.randn((1, H, W, cout), device=dev).to(dt)); r2 = ops.Act(torch.randn((1, H, W, cout), device=dev).to(dt))
    conv = Conv2d(cin, cout, 3, 1, 1).to(dev)
    cases = [("plain", {}), ("pre + act", dict(pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01)), ("res1", dict(res1=r1))] + ([("res1 + res2", dict(res1=r1, res2=r2))] if mode == "fp32" else [])
    for rnd in range(args.rounds):
        for name, kw in cases:
            for opt in (0, 1):
                ops.set_option("pps_oneseg", opt)
                out = ops.new_act(1, H, W, cout, dt, dev)
                c0 = ops.get_counter("conv3x3_pps")
                for _ in range(2): conv.run(x, out, out_dtype=dt, **kw)
                took = ops.get_counter("conv3x3_pps") - c0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(n): conv.run(x, out, out_dtype=dt, **kw)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                key = (mode, name, opt); best[key] = min(best.get(key, 1e9), us)
                if rnd == 0:
                    if opt == 0: ref[(mode, name)] = out.buf.clone()
                    else: print(f"   {mode} {name}: streamed-weights launches {took // 2} per run, bit-identical {torch.equal(ref[(mode, name)].view(torch.int32 if dt == torch.float32 else torch.int16), out.buf.view(torch.int32 if dt == torch.float32 else torch.int16))}", flush=True)
    for name, kw in cases:
        print(f"{mode} {name:14s} pps_oneseg 0 -> 1: {best[(mode, name, 0)]:8.1f} -> {best[(mode, name, 1)]:8.1f} us", flush=True)
ops.set_option("pps_oneseg", 1); ops.set_precision("fp32")
