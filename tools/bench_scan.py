"""What the entropy scan costs on an otherwise idle GPU: every capturable scan segment of both P-frame codecs at the 1080p latent size
(68 x 120) replayed as its HIP graph and timed with events -- the encoder scan (one graph) and the six decoder segments (dec0 = hyper synthesis +
level-0 nets, dec1..5 = dequantise + LRP of a level + nets of the next) -- plus the launch count of each segment.  These are the ~250 small
launches of a frame: the figure a fused per-level kernel has to beat.   python tools/bench_scan.py [f16|fp32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda:0")
ops.set_precision(mode)
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
H, W = 1088, 1920
fr = [f.to(dev) for f in synth.make_frames(H, W, 2)]
sm = synth.make_sm_fea(H, W, 1).to(dev)
for _ in range(3):                                   # warm: weights packed, scan programs recorded, graphs captured
    mv, res = m.compress(fr[0], fr[1], sm, None)
    m.decompress(fr[0], mv, res, sm, None)
torch.cuda.synchronize()
tot = {}
for name, codec in (("mv", m.mv_codec), ("res", m.res_codec)):
    ws = codec._plan(1, H // 16, W // 16, dev)
    codec.capture_scan_graphs(1, H // 16, W // 16, dev)
    for key in codec._segments(ws):
        g, prog = ws.graphs.get(key), ws.programs.get(key)
        if g is None:
            print(f"{name} {key}: no graph"); continue
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        n = prog.n if prog else -1
        tot[name] = tot.get(name, 0.0) + (us if key != "enc" else 0.0)
        print(f"{mode} {name:3s} {key:5s}: {n:3d} launches, {us:8.1f} us per replay ({us / max(n, 1):5.1f} us per launch)", flush=True)
    print(f"{mode} {name}: decoder segments together {tot[name]:.0f} us", flush=True)
