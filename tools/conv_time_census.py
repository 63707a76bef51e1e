"""Per-configuration time of every conv launch in one steady-state P-frame (compress+decompress), each launch timed in
isolation with events (graphs off).  python tools/conv_time_census.py [min_us]"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth, _native as N  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402

ops.set_precision(os.environ.get("CENSUS_MODE", "f16"))
dev = "cuda:0"
model = synth.make_model("W0"); model.update(force=True); model = model.to(dev)
frames = [pad(f.to(dev), 64) for f in synth.make_frames(1080, 1920, 3)]
sms = [synth.make_sm_fea(1088, 1920, t).to(dev) for t in range(3)]
mv, res = model.compress(frames[0], frames[1], sms[1], None)
feat, ref, _, _ = model.decompress(frames[0], mv, res, sms[1], None)
torch.cuda.synchronize()
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
    x = inputs[0]
    k = (pc.kh, pc.stride, pc.segC, tuple(dt(a) for a in inputs), pc.Cout, x.H, x.W, "pre=" + ACT[kw.get("pre_act", 0)],
         "act=" + ACT[kw.get("act", 0)], "r1=" + dt(kw.get("res1")), "gate" if kw.get("gate") is not None else "",
         "r2=" + dt(kw.get("res2")), "sh%d" % kw.get("shuffle", 1), "post" if kw.get("post") is not None else "", "out=" + dt(o))
    cnt[k] += 1; tim[k] += e0.elapsed_time(e1) * 1e3
    return o
ops.conv2d = wrapped
import deepsvc_amd.modules as M
mv, res = model.compress(ref, frames[2], sms[2], feat)
dict(mv); dict(res)
model.decompress(ref, mv, res, sms[2], feat)
torch.cuda.synchronize()
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
tot = sum(tim.values())
print(f"total conv time (isolated launches) {tot/1e3:.2f} ms/frame over {sum(cnt.values())} launches")
for k, t in sorted(tim.items(), key=lambda kv: -kv[1]):
    if t < min_us: continue
    n = cnt[k]
    flop = 2.0 * k[5] * k[6] * sum(k[2]) * k[4] * k[0] ** 2 / k[1] ** 2
    print(f"{n:4d} x {t/n:7.1f} us = {t/1e3:6.2f} ms  {flop/(t/n)/1e6:7.1f} TF/s ", *k)
