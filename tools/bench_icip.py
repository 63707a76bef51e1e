"""I-frame codec (ICIP2020ResB) encode / decode time at 1080p (padded 1088x1920), f16 and fp32.   python tools/bench_icip.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth
from deepsvc_amd.modules import pad
dev = "cuda:0"
for prec in ("f16", "fp32"):
    ops.set_precision(prec)
    m = synth.make_icodec("W0"); m.update(force=True); m = m.to(dev)
    x = pad(synth.make_frames(1080, 1920, 1)[0].to(dev), 64)
    for _ in range(3):
        enc = m.compress(x); dec = m.decompress(enc["strings"], enc["shape"])
    torch.cuda.synchronize()
    n = 5 if prec == "f16" else 2
    te = td = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); enc = m.compress(x); torch.cuda.synchronize(); t1 = time.perf_counter()
        dec = m.decompress(enc["strings"], enc["shape"]); torch.cuda.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; td += t2 - t1
    nb = sum(len(s[0]) for s in enc["strings"])
    print(f"{prec}: I-frame 1080p encode {1e3*te/n:.1f} ms  decode {1e3*td/n:.1f} ms  {nb} B = {nb*8/(1080*1920):.3f} bpp", flush=True)
