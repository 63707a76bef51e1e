"""Per-shape conv time inside real P-frames: every conv launch of `steps` compress+decompress calls is bracketed by
HIP events (ops.PROFILE hook) and aggregated by shape.  python tools/profile_frame.py [--precision f16] [--steps 2]"""
import argparse
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    ops.set_precision(args.precision)
    dev = "cuda:0"
    model = synth.make_model("W0")
    model.update(force=True)
    model = model.to(dev)
    h, w = 1080, 1920
    frames = [pad(f.to(dev), 64) for f in synth.make_frames(h, w, args.steps + 2)]
    sms = [synth.make_sm_fea(1088, 1920, t).to(dev) for t in range(args.steps + 2)]
    ref, feat = frames[0], None
    mv, res = model.compress(ref, frames[1], sms[1], feat)          # warm-up (packs weights)
    feat, ref, _, _ = model.decompress(ref, mv, res, sms[1], feat)
    keys = []
    ops.PROFILE = {"match": lambda pc, x0: keys.append((pc.kh, pc.stride, pc.segC, pc.Cout, x0.H, x0.W, str(x0.dtype)[6:])) or True, "pairs": []}
    torch.cuda.synchronize()
    t_c = t_d = 0.0
    for t in range(2, args.steps + 2):
        t0 = time.perf_counter()
        mv, res = model.compress(ref, frames[t], sms[t], feat)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        feat, ref, _, _ = model.decompress(ref, mv, res, sms[t], feat)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t_c += t1 - t0; t_d += t2 - t1
    prof, ops.PROFILE = ops.PROFILE, None
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k, (a, b) in zip(keys, prof["pairs"]):
        agg[k][0] += 1
        agg[k][1] += a.elapsed_time(b)
    tot = sum(v[1] for v in agg.values())
    print(f"steps {args.steps}: compress {1e3*t_c/args.steps:.1f} ms  decompress {1e3*t_d/args.steps:.1f} ms per frame (with event overhead); "
          f"conv kernels {tot/args.steps:.1f} ms/frame in {sum(v[0] for v in agg.values())//args.steps} launches")
    print(f"{'k':>2} {'s':>1} {'segC':>14} {'Cout':>4} {'HxW':>10} {'in':>8} {'n/frame':>7} {'ms/frame':>8} {'us/launch':>9} {'TFLOP/s':>8}")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        kh, st, segC, cout, H, W, dt = k
        flop = 2.0 * (H // st) * (W // st) * cout * sum(segC) * kh * kh
        print(f"{kh:>2} {st:>1} {str(segC):>14} {cout:>4} {H:>4}x{W:<5} {dt:>8} {n/args.steps:>7.1f} {ms/args.steps:>8.2f} {1e3*ms/n:>9.1f} {flop*n/ms/1e9:>8.1f}")


if __name__ == "__main__":
    main()
