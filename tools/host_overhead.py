"""Host-side cost of one P-frame: at 64x64 the GPU work is negligible, so the wall time of compress + decompress is the Python /
ctypes / launch / sync overhead that every lane thread pays per frame while holding the GIL part of the time.
    python tools/host_overhead.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
for size in (64, 256):
    fr = [f.to(dev) for f in synth.make_frames(size, size, 7)]
    sms = [None] + [synth.make_sm_fea(size, size, t).to(dev) for t in range(1, 7)]
    def gop(n=1):
        for _ in range(n):
            ref, feat = fr[0], None
            for t in range(1, 7):
                mv, res = m.compress(ref, fr[t], sms[t], feat)
                feat, ref, _, _ = m.decompress(ref, mv, res, sms[t], feat)
        torch.cuda.synchronize()
    gop(2)
    t0 = time.perf_counter(); gop(3); dt = (time.perf_counter() - t0) / 18
    print(f"{size}x{size}: {dt * 1e3:.2f} ms per frame (host-bound)")
    if size == 64:
        pr = cProfile.Profile(); pr.enable(); gop(2); pr.disable()
        st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
