"""What the GPU-decodable multi-stream range coder (coder = "rans_ms", a LABELLED non-reference string format) buys and costs at 1088x1920:
the intra codec's encode + decode time and string size in both modes, and one P-frame's
compress + decompress with one GOP in flight.   python tools/bench_ms.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth

dev = "cuda:0"
ops.set_precision("f16")
H, W = 1088, 1920
x = synth.make_frames(H, W, 1)[0].to(dev)
for coder in ("rans", "rans_ms"):
    m = synth.make_icodec("W0"); m.update(force=True); m = m.to(dev); m.coder = coder
    for _ in range(3):
        enc = m.compress(x); m.decompress(enc["strings"], enc["shape"])
    torch.cuda.synchronize()
    te = td = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); enc = m.compress(x); _ = enc["strings"]; torch.cuda.synchronize(); t1 = time.perf_counter()
        m.decompress(enc["strings"], enc["shape"]); torch.cuda.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; td += t2 - t1
    ny, nz = len(enc["strings"][0][0]), len(enc["strings"][1][0])
    print(f"ICIP2020ResB 1088x1920 f16 {coder:8s}: encode {te / 5 * 1e3:6.1f} ms  decode {td / 5 * 1e3:6.1f} ms  y string {ny} B  z string {nz} B  ({8.0 * (ny + nz) / (1080 * 1920):.4f} bpp)", flush=True)
fr = [f.to(dev) for f in synth.make_frames(H, W, 7)]
sms = [None] + [synth.make_sm_fea(H, W, t).to(dev) for t in range(1, 7)]
for coder in ("rans", "rans_ms"):
    m = synth.make_model("W0"); m.update(force=True); m = m.to(dev); m.set_coder(coder)
    def gop():
        ref, feat, nb = fr[0], None, 0
        for t in range(1, 7):
            mv, res = m.compress(ref, fr[t], sms[t], feat)
            feat, ref, _, _ = m.decompress(ref, mv, res, sms[t], feat)
            nb += sum(len(s[0]) for e in (mv, res) for s in e["strings"])
        torch.cuda.synchronize()
        return nb
    gop(); gop()
    t0 = time.perf_counter(); nb = gop(); nb = gop(); dt = (time.perf_counter() - t0) / 12
    print(f"P-frames 1088x1920 f16 one GOP in flight {coder:8s}: {dt * 1e3:6.2f} ms per frame = {1 / dt:5.2f} fps, {nb / 6:.0f} B per frame ({8.0 * nb / 6 / (1080 * 1920):.4f} bpp)", flush=True)
