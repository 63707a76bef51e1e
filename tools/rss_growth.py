"""Does the host side leak?  Resident set size of the process after N and after 10 N chained 1080p P-frames on one lane.
    python tools/rss_growth.py [N=20]"""
import os, resource, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ops.set_precision("f16")
dev = "cuda:0"
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
fr = [pad(f.to(dev), 64) for f in synth.make_frames(1080, 1920, 7)]
sms = [None] + [synth.make_sm_fea(1088, 1920, t).to(dev) for t in range(1, 7)]


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def run(n):
    done = 0
    while done < n:
        ref, feat = fr[0], None
        for t in range(1, 7):
            mv, res = m.compress(ref, fr[t], sms[t], feat)
            feat, ref, _, _ = m.decompress(ref, mv, res, sms[t], feat)
            done += 1
    torch.cuda.synchronize()


run(12)
a = rss_mb(); ga = torch.cuda.memory_allocated() / 2**20
run(N)
b = rss_mb(); gb = torch.cuda.memory_allocated() / 2**20
run(10 * N)
c = rss_mb(); gc_ = torch.cuda.memory_allocated() / 2**20
print(f"host RSS after warm-up {a:.0f} MB, after {N} more frames {b:.0f} MB, after {10 * N} more {c:.0f} MB  ->  {(c - b) / (10 * N) * 1024:.1f} KB per frame")
print(f"HBM held: {ga:.0f} / {gb:.0f} / {gc_:.0f} MB")
