"""Phase shares of the general conv kernel on fp32 mode's dominant shape (3x3 s1 64->64 @1088x1920), wave 0 of every workgroup.
Needs the instrumented build:  python -m deepsvc_amd.build --variant probe -DDSVC_PHASE_PROBE
    DSVC_LIB=$PWD/deepsvc_amd/lib/libdsvc_probe.so python tools/phase_probe_fp32.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

dev = "cuda:0"
L = ctypes.CDLL(N.LIB_PATH)
names = ["barrier(pre-stage)", "patch staging", "weight staging", "barrier(pre-mfma)", "mfma phase", "epilogue"]
for mode, dt, cin, cout, k, H, W in (("fp32", torch.float32, [64], 64, 3, 1088, 1920), ("fp32", torch.float32, [96], 96, 3, 544, 960),
                                     ("f16", torch.float16, [64, 64], 64, 3, 1088, 1920), ("f16", torch.float16, [96], 96, 3, 544, 960)):
    ops.set_precision(mode)
    ops.set_option("specialised", 0)
    m = Conv2d(sum(cin), cout, k, 1, padding=k // 2).to(dev)
    ins = [ops.Act(torch.randn((1, H, W, c), device=dev).to(dt)) for c in cin]
    out = m.run(ins, out_dtype=dt)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.dsvc_dbg_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.run(ins, out, out_dtype=dt)
    e1.record(); torch.cuda.synchronize()
    L.dsvc_dbg_phase_read(buf, 1)
    v = list(buf)
    tot, nwg = v[6], v[7]
    print(f"{mode} {k}x{k} {'+'.join(map(str, cin))}->{cout} @{H}x{W}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch (instrumented), {nwg // 5} workgroups, "
          f"{tot / max(nwg, 1):.0f} clocks per workgroup (wave 0: from its first to its last instruction)")
    for i, n in enumerate(names):
        print(f"    {n:20s} {100.0 * v[i] / max(tot, 1):5.1f} %   {v[i] / max(nwg, 1):9.0f} clocks per workgroup")
ops.set_precision("fp32"); ops.set_option("specialised", 1)
