"""Per-configuration time of every conv launch in one I-frame (ICIP2020ResB compress+decompress) at 1080p, each launch timed
in isolation with events (graphs off; ~30 us of event/idle overhead per launch included).  python tools/icip_time_census.py [min_us]"""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"
m = synth.make_icodec("W0"); m.update(force=True); m = m.to(dev)
x = pad(synth.make_frames(1080, 1920, 1)[0].to(dev), 64)
for _ in range(2):
    enc = m.compress(x); dec = m.decompress(enc["strings"], enc["shape"])
torch.cuda.synchronize()
t0 = time.perf_counter(); enc = m.compress(x); torch.cuda.synchronize(); t1 = time.perf_counter()
dec = m.decompress(enc["strings"], enc["shape"]); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graphs on: encode {1e3*(t1-t0):.1f} ms decode {1e3*(t2-t1):.1f} ms")
ops.USE_GRAPHS[0] = False
cnt, tim = collections.Counter(), collections.Counter()
orig = ops.conv2d
ACT = ["none", "relu", "lrelu", "gelu", "sigm", "htanh"]
def dt(a): return "-" if a is None else str(a.dtype)[6:].replace("float32", "f32").replace("float16", "f16")
def wrapped(pc, inputs, out=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    o = orig(pc, inputs, out, **kw)
    e1.record(); torch.cuda.synchronize()
    x0 = inputs[0]
    k = (pc.kh, pc.stride, pc.segC, tuple(dt(a) for a in inputs), pc.Cout, x0.H, x0.W, "s2d" if getattr(pc, "s2d", False) else "",
         "items%d" % pc.items if kw.get("items") else "", "act=" + ACT[kw.get("act", 0)], "r1=" + dt(kw.get("res1")),
         "sh%d" % kw.get("shuffle", 1), "post" if kw.get("post") is not None else "", "out=" + dt(o))
    cnt[k] += 1; tim[k] += e0.elapsed_time(e1) * 1e3
    return o
ops.conv2d = wrapped
enc = m.compress(x); dict(enc)
m.decompress(enc["strings"], enc["shape"])
torch.cuda.synchronize()
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
print(f"total conv time (isolated launches) {sum(tim.values())/1e3:.2f} ms over {sum(cnt.values())} launches")
for k, t in sorted(tim.items(), key=lambda kv: -kv[1]):
    if t < min_us: continue
    n = cnt[k]
    items = int(k[8][5:]) if k[8] else 1
    flop = 2.0 * k[5] * k[6] * sum(k[2]) * k[4] * k[0] ** 2 / k[1] ** 2 * items / (4 if k[7] else 1) * (1 if not k[7] else 1)
    print(f"{n:4d} x {t/n:7.1f} us = {t/1e3:6.2f} ms  {flop/(t/n)/1e6:7.1f} TF/s ", *k)
