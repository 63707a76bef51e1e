"""Phase lengths of the streamed-weights two-group kernel (csrc/conv3x3_pps.hip), fp32 and f16: for wave 0 of each group, shader clocks per
stream step (one 16- / 32-channel chunk of one tile) in the M phase, at the barrier after it, in the S (+ epilogue) slot and at its barrier.
    python -m deepsvc_amd.build --variant ppsprobe -DDSVC_PPS_PROBE
    DSVC_LIB=$PWD/deepsvc_amd/lib/libdsvc_ppsprobe.so python tools/pps_phase_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

dev = "cuda:0"
H, W = 1088, 1920
L = ctypes.CDLL(N.LIB_PATH)
buf = (ctypes.c_ulonglong * 17)()
for mode, dt, segs in (("fp32", torch.float32, (64,)), ("fp32", torch.float32, (64, 64)), ("f16", torch.float16, (64, 64))):
    ops.set_precision(mode)
    xs = [ops.Act(torch.randn((1, H, W, c), device=dev).to(dt)) for c in segs]
    r1 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(dt))
    out = ops.new_act(1, H, W, 64, dt, dev)
    conv = Conv2d(sum(segs), 64, 3, 1, 1).to(dev)
    for name, kw in (("plain", {}), ("+ res1", dict(res1=r1))):
        if mode == "f16" and kw:
            continue
        for _ in range(3):
            conv.run(xs, out, out_dtype=dt, **kw)
        torch.cuda.synchronize()
        L.dsvc_dbg_pps_phase_read(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv.run(xs, out, out_dtype=dt, **kw)
        e1.record(); torch.cuda.synchronize()
        L.dsvc_dbg_pps_phase_read(buf, 1)
        v = list(buf)
        steps = max(v[16], 1)
        print(f"{mode} {'+'.join(map(str, segs))} -> 64 {name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (instrumented); clocks per stream step:")
        for g in (0, 1):
            b = v[g * 8:g * 8 + 8]
            ghz = b[6] / max(b[5], 1) * 0.1
            print(f"    group {g}: M {b[0] / steps:7.0f} + wait {b[1] / steps:6.0f} | S(+E) {b[2] / steps:7.0f} (epilogue share {b[4] / steps:6.0f}) + wait {b[3] / steps:6.0f}"
                  f"   | sum {sum(b[:4]) / steps:7.0f}   clock {ghz:.2f} GHz")
