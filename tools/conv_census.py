"""Which conv shapes does a 1080p P-frame spend its GPU time in?  One steady-state compress()+decompress() issued launch by launch
(no recorded programs, no graphs), every ops.conv2d call bracketed by events on the issuing stream, grouped by shape.

    DSVC_NO_FRAME_PROGRAMS=1 DSVC_NO_SCAN_STEP=1 DSVC_NO_GRAPHS=1 python tools/conv_census.py [f16|fp32]

Prints per shape: launches per frame, mean duration, share of the conv time, achieved TFLOP/s and the fraction of the dense MFMA peak
(2.5 PFLOP/s f16, 157 TFLOP/s fp32) -- the list a kernel rebuild is prioritised by."""
import os
import sys
import collections
import torch
for k in ("DSVC_NO_FRAME_PROGRAMS", "DSVC_NO_SCAN_STEP", "DSVC_NO_GRAPHS"):
    os.environ.setdefault(k, "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
ops.set_precision(mode)
peak = 2.5e15 if mode == "f16" else 157.3e12
dev = "cuda:0"
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
fr = [pad(f.to(dev), 64) for f in synth.make_frames(1080, 1920, 4)]
sms = [None] + [synth.make_sm_fea(1088, 1920, t).to(dev) for t in range(1, 4)]

records = None
real_conv2d = ops.conv2d


def conv2d(pc, inputs, out=None, **kw):
    if records is None:
        return real_conv2d(pc, inputs, out, **kw)
    x0 = inputs[0]
    items = kw.get("items")
    G = pc.items if items is not None else 1
    flags = "".join(s for s, on in (("p", kw.get("pre_act", 0)), ("a", kw.get("act", 0)), ("1", kw.get("res1") is not None),
                                    ("2", kw.get("res2") is not None), ("g", kw.get("gate") is not None), ("P", kw.get("post") is not None),
                                    ("s", kw.get("shuffle", 1) > 1), ("D", getattr(pc, "s2d", False))) if on)
    key = (pc.kh, pc.stride, "+".join(str(a.C) for a in inputs), pc.Cout, x0.H, x0.W, G * x0.B, flags)
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    r = real_conv2d(pc, inputs, out, **kw)
    b.record(st)
    records.append((key, a, b))
    return r


ops.conv2d = conv2d
import deepsvc_amd.modules as MOD, deepsvc_amd.image_model as IM, deepsvc_amd.video_model as VM  # noqa: E402
for mod in (MOD, IM, VM):
    if getattr(mod, "conv2d", None) is real_conv2d:
        mod.conv2d = conv2d


def frame(ref, feat, t):
    mv, res = m.compress(ref, fr[t], sms[t], feat)
    return m.decompress(ref, mv, res, sms[t], feat)


feat, ref, _, _ = frame(fr[0], None, 1)
feat, ref, _, _ = frame(ref, feat, 2)
torch.cuda.synchronize()
acc = collections.defaultdict(list)
REPS = 3
for _ in range(REPS):
    records = []
    frame(ref, feat, 3)
    torch.cuda.synchronize()
    for key, a, b in records:
        acc[key].append(a.elapsed_time(b) * 1e3)
    records = None
rows = []
for key, us in acc.items():
    kh, stride, cin, cout, H, W, B, flags = key
    C = sum(int(c) for c in cin.split("+")) * (1 if "D" not in flags else 1)
    Ho, Wo = (H // stride, W // stride)
    if "D" in flags:
        Ho, Wo = H // 2, W // 2
    flops = 2.0 * B * Ho * Wo * C * cout * kh * kh
    n = len(us) / REPS
    mean = sum(us) / len(us)
    rows.append((n * mean, n, mean, flops, key))
total = sum(r[0] for r in rows)
rows.sort(reverse=True)
print(f"{mode}: {sum(r[1] for r in rows):.0f} conv launches per P-frame (compress + decompress), {total / 1e3:.2f} ms between their events")
print(f"{'k':>2} {'s':>1} {'Cin':>10} {'Cout':>5} {'H':>5} {'W':>5} {'B':>3} {'flags':>6} {'n/frame':>8} {'mean us':>9} {'share':>6} {'TFLOP/s':>8} {'of peak':>7}")
for tot, n, mean, flops, key in rows:
    kh, stride, cin, cout, H, W, B, flags = key
    tf = flops / (mean * 1e-6) / 1e12
    print(f"{kh:>2} {stride:>1} {cin:>10} {cout:>5} {H:>5} {W:>5} {B:>3} {flags:>6} {n:>8.1f} {mean:>9.1f} {tot / total * 100:>5.1f}% {tf:>8.1f} {tf * 1e12 / peak:>7.3f}")
