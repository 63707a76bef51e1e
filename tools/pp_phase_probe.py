"""Phase lengths of the ping-pong conv kernel (csrc/conv3x3_pp.hip) per epilogue variant: for wave 0 of each group, shader clocks
of work and of waiting at the barrier in each of the four phases (M0 | S1 | M1 | E+S0; group 1 runs one phase behind group 0).
Needs the instrumented build:
    DSVC_BUILD_DEFINES=-DDSVC_PP_PROBE python -m deepsvc_amd.build --force && cp deepsvc_amd/lib/libdsvc.so deepsvc_amd/lib/libdsvc_probe.so
    python -m deepsvc_amd.build --force        # back to the product build
    DSVC_LIB=$PWD/deepsvc_amd/lib/libdsvc_probe.so python tools/pp_phase_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("f16")
dev, h = "cuda:0", torch.float16
H, W = 1088, 1920
L = ctypes.CDLL(N.LIB_PATH)
x = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
r1 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
r2 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
out = ops.new_act(1, H, W, 64, h, dev)
conv = Conv2d(64, 64, 3, 1, 1).to(dev)
post = Conv2d(64, 64, 1, 1, 0).to(dev)._packed([64], dev)
CASES = [("plain", {}), ("pre lrelu + act lrelu", dict(pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01)), ("res1", dict(res1=r1)),
         ("res1 + res2", dict(res1=r1, res2=r2)), ("act + res1 + post(1x1)", dict(act=N.ACT_LRELU, res1=r1, post=post))]
names = ["M0", "S1", "M1", "E+S0"]
buf = (ctypes.c_ulonglong * 17)()
for name, kw in CASES:
    for _ in range(3):
        conv.run(x, out, out_dtype=h, **kw)
    torch.cuda.synchronize()
    L.dsvc_dbg_pp_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv.run(x, out, out_dtype=h, **kw)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_pp_phase_read(buf, 1)
    v = list(buf)
    it = max(v[16], 1)
    print(f"{name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (instrumented); clocks per iteration (= one tile per group), work + barrier wait:")
    for g in (0, 1):
        row = "  ".join(f"{names[i]:5s} {v[g * 8 + 2 * i] / it:6.0f} +{v[g * 8 + 2 * i + 1] / it:6.0f}" for i in range(4))
        tot = sum(v[g * 8:g * 8 + 8]) / it
        print(f"    group {g}: {row}   | total {tot:7.0f}")
