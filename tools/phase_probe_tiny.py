"""Per-phase shader clocks of the general conv kernel on the SMALL launches of a 1080p frame (entropy-scan and 1/8 - 1/16 resolution
convs: ~250 launches and ~7 ms of a 38 ms frame, 15-50 us each for a few hundred MFLOP).  Instrumented build as tools/phase_probe.py:
    python -m deepsvc_amd.build --variant probe -DDSVC_PHASE_PROBE
    DSVC_LIB=deepsvc_amd/lib/libdsvc_probe.so python tools/phase_probe_tiny.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"
L = ctypes.CDLL(N.LIB_PATH)
names = ["barrier(pre-stage)", "patch staging", "weight staging", "barrier(pre-mfma)", "mfma phase", "epilogue"]
SHAPES = [((64,), 32, 3, 68, 120, 1), ((64,), 32, 3, 68, 120, 4), ((96,), 48, 3, 68, 120, 1), ((96, 48), 96, 3, 68, 120, 1), ((64,), 64, 3, 136, 240, 1),
          ((96,), 96, 3, 136, 240, 1), ((64,), 64, 1, 136, 240, 1), ((128,), 128, 3, 272, 480, 1),
          # the large shapes of DESIGN.md section 9 item 1b
          ((96,), 96, 3, 544, 960, 1), ((64, 64), 64, 3, 1088, 1920, 1), ((64,), 32, 7, 1088, 1920, 1), ((32,), 64, 7, 1088, 1920, 1)]
for segs, cout, k, H, W, B in SHAPES:
    m = Conv2d(sum(segs), cout, k, 1, padding=k // 2).to(dev)
    ins = [ops.Act(torch.randn((B, H, W, c), device=dev).to(torch.float16)) for c in segs]
    out = m.run(ins, act=N.ACT_LRELU)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.dsvc_dbg_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.run(ins, out, act=N.ACT_LRELU)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_phase_read(buf, 1)
    v = list(buf)
    tot, nwg = v[6], v[7]
    print(f"{k}x{k} {'+'.join(map(str, segs))}->{cout} @{H}x{W} x{B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch back to back (instrumented), {nwg // 20} workgroups, "
          f"{tot / max(nwg, 1):.0f} clocks per workgroup")
    print("    " + "   ".join(f"{n} {v[i] / max(nwg, 1):.0f}" for i, n in enumerate(names)))
