"""Where the general conv kernel's time goes: per-phase shader-clock totals of wave 0 of every workgroup.
Needs the instrumented build:
    DSVC_BUILD_DEFINES=-DDSVC_PHASE_PROBE  (objects compiled with deepsvc_amd.build.FLAGS, linked to deepsvc_amd/lib/libdsvc_probe.so)
    DSVC_LIB=deepsvc_amd/lib/libdsvc_probe.so python tools/phase_probe.py [shape index of tools/bench_conv.py ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv import SHAPES  # noqa: E402

ops.set_precision("f16")
ops.set_option("specialised", 0)
dev = "cuda:0"
L = ctypes.CDLL(N.LIB_PATH)
names = ["barrier(pre-stage)", "patch staging", "weight staging", "barrier(pre-mfma)", "mfma phase", "epilogue"]
for idx in [int(a) for a in sys.argv[1:]] or [4, 2, 3, 9]:
    name, segs, cout, k, s, H, W, kw = SHAPES[idx]
    m = Conv2d(sum(segs), cout, k, s, padding=k // 2).to(dev)
    ins = [ops.Act(torch.randn((1, H, W, c), device=dev).to(torch.float16)) for c in segs]
    out = m.run(ins)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.dsvc_dbg_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.run(ins, out)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_phase_read(buf, 1)
    v = list(buf)
    tot, nwg = v[6], v[7]
    print(f"{name}: {e0.elapsed_time(e1)/5*1e3:.1f} us per launch (instrumented), {nwg//5} workgroups, {tot/max(nwg,1):.0f} clocks per workgroup")
    for i, n in enumerate(names):
        print(f"    {n:20s} {100.0*v[i]/max(tot,1):5.1f} %   {v[i]/max(nwg,1):9.0f} clocks per workgroup")
