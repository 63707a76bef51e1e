"""Where does the HOST spend a 1080p frame (one GOP lane)?  cProfile over 6 chained P-frames: blocking calls (stream syncs, the C
segment runs, lock waits) show which side is the critical path.   python tools/host_profile_1080p.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402
from deepsvc_amd.modules import pad  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
fr = [pad(f.to(dev), 64) for f in synth.make_frames(1080, 1920, 7)]
sms = [None] + [synth.make_sm_fea(1088, 1920, t).to(dev) for t in range(1, 7)]
stamps = []
def gop(log=False):
    ref, feat = fr[0], None
    for t in range(1, 7):
        t0 = time.perf_counter()
        mv, res = m.compress(ref, fr[t], sms[t], feat)
        t1 = time.perf_counter()
        feat, ref, _, _ = m.decompress(ref, mv, res, sms[t], feat)
        t2 = time.perf_counter()
        if log:
            stamps.append((t1 - t0, t2 - t1))
    torch.cuda.synchronize()
gop(); gop()
t0 = time.perf_counter(); gop(True); dt = (time.perf_counter() - t0) / 6
print(f"1080p: {dt * 1e3:.2f} ms per frame; host time inside compress() / decompress() per frame: " + ", ".join(f"{a * 1e3:.1f}/{b * 1e3:.1f}" for a, b in stamps))
from deepsvc_amd import video_model as VM  # noqa: E402
VM.HOST_TRACE = []
gop()
tr, VM.HOST_TRACE = VM.HOST_TRACE, None
t_first = tr[0][1]
print("host stamps (ms since the first compress() of the GOP):")
print("  " + "  ".join(f"{lab}@{(t - t_first) * 1e3:.1f}" for lab, t in tr))
pr = cProfile.Profile(); pr.enable(); gop(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
