"""The shader clock the chip holds INSIDE the dominant conv kernel, measured the guide's way (MI355X_MICROARCH.md, DVFS give-back
item 6): delta(s_memtime) / delta(s_memrealtime) x 100 MHz around the tile loop, after >= 2 s of back-to-back launches on random data.
Needs the diagnostic build:
    DSVC_BUILD_DEFINES=-DDSVC_PP_CLOCK python -m deepsvc_amd.build --force && cp deepsvc_amd/lib/libdsvc.so deepsvc_amd/lib/libdsvc_clock.so
    python -m deepsvc_amd.build --force        # back to the product build
    DSVC_LIB=$PWD/deepsvc_amd/lib/libdsvc_clock.so python tools/pp_clock.py"""
import ctypes, os, sys, time
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
buf = (ctypes.c_ulonglong * 3)()
for name, kw in CASES:
    t0 = time.time()
    while time.time() - t0 < 2.2:                      # >= 2 s of back-to-back launches: the clock has settled under this load
        for _ in range(50):
            conv.run(x, out, out_dtype=h, **kw)
        torch.cuda.synchronize()
    L.dsvc_dbg_pp_clock_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 200
    for _ in range(n):
        conv.run(x, out, out_dtype=h, **kw)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_pp_clock_read(buf, 1)
    dt, dr, cnt = list(buf)
    us = e0.elapsed_time(e1) / n * 1e3
    ghz = dt / max(dr, 1) * 0.1
    loop_us = dr / max(cnt, 1) / 100.0
    print(f"{name:28s} {us:7.1f} us per launch, tile loop {loop_us:6.1f} us, in-kernel clock {ghz:.3f} GHz "
          f"-> {154.0e9 / us / 1e6:.1f} TFLOP/s = {154.0e9 / us / 1e6 / (2500.0 * ghz / 2.4):.2f} of the MFMA rate available at that clock "
          f"({2500.0 * ghz / 2.4:.0f} TFLOP/s)", flush=True)

# ---- the general kernel (csrc/conv_mfma.hip): fp32 mode's dominant conv and the f16 shapes that do not fit the ping-pong kernel ----
def generic(name, mode, cin, cout, k, Hh, Ww, flop):
    ops.set_precision(mode)
    dt = torch.float32 if mode == "fp32" else torch.float16
    xs = [ops.Act(torch.randn((1, Hh, Ww, c), device=dev).to(dt)) for c in cin]
    conv = Conv2d(sum(cin), cout, k, 1, k // 2).to(dev)
    out = ops.new_act(1, Hh, Ww, cout, dt, dev)
    t0 = time.time()
    while time.time() - t0 < 2.2:
        for _ in range(10):
            conv.run(xs, out, out_dtype=dt)
        torch.cuda.synchronize()
    L.dsvc_dbg_mfma_clock_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 40
    for _ in range(n):
        conv.run(xs, out, out_dtype=dt)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_mfma_clock_read(buf, 1)
    dtk, dr, cnt = list(buf)
    us = e0.elapsed_time(e1) / n * 1e3
    ghz = dtk / max(dr, 1) * 0.1
    peak = (157.3 if mode == "fp32" else 2500.0) * ghz / 2.4
    print(f"{name:34s} {us:8.1f} us per launch, in-kernel clock {ghz:.3f} GHz -> {flop / us / 1e6:7.1f} TFLOP/s = {flop / us / 1e6 / peak:.2f} of the "
          f"{peak:.0f} TFLOP/s available at that clock ({'fp32' if mode == 'fp32' else 'f16'} MFMA)", flush=True)


F = lambda cin, cout, k, Hh, Ww: 2.0 * Hh * Ww * cin * cout * k * k
generic("fp32 3x3 64->64 @1088x1920", "fp32", [64], 64, 3, H, W, F(64, 64, 3, H, W))
generic("f16 3x3 64+64->64 @1088x1920", "f16", [64, 64], 64, 3, H, W, F(128, 64, 3, H, W))
generic("f16 3x3 96->96 @544x960", "f16", [96], 96, 3, H // 2, W // 2, F(96, 96, 3, H // 2, W // 2))
generic("f16 3x3 128->128 @272x480", "f16", [128], 128, 3, H // 4, W // 4, F(128, 128, 3, H // 4, W // 4))
generic("f16 7x7 64->32 @1088x1920", "f16", [64], 32, 7, H, W, F(64, 32, 7, H, W))
ops.set_precision("fp32")
