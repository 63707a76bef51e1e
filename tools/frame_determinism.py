"""Is a 1080p f16 P-frame bit-stable?  compress twice, decompress three times through the Python layer (frame programs, side stream, worker threads:
the memory pipeline is busy), compare strings and decoded tensors.  This is the check that exposed the wide-store hazard of round 5 (csrc/conv3x3_pp.hip,
PP_EPI_ST_SOFF): every isolated launch was bit-exact, the frame was not.
    python tools/frame_determinism.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops, synth
from deepsvc_amd.modules import pad
ops.set_precision("f16")
dev = "cuda:0"
m = synth.make_model("W0"); m.update(force=True); m = m.to(dev)
fr = [pad(f.to(dev), 64) for f in synth.make_frames(1080, 1920, 3)]
sms = [None] + [synth.make_sm_fea(1088, 1920, t).to(dev) for t in (1, 2)]
ref, feat = fr[0], None
for t in (1, 2):
    outs = []
    strs = []
    for rep in range(2):
        mv, res = m.compress(ref, fr[t], sms[t], feat)
        strs.append([mv["strings"][0][0], mv["strings"][1][0], res["strings"][0][0], res["strings"][1][0]])
    for rep in range(3):
        f2, r2, w2, p2 = m.decompress(ref, mv, res, sms[t], feat)
        torch.cuda.synchronize()
        outs.append((f2.float().clone(), r2.clone(), w2.clone(), p2.clone()))
    print(f"frame {t}: strings stable {strs[0] == strs[1]}; decode 2 vs 1: " + ", ".join(f"{nm} differing {int((a != b).sum())}" for nm, a, b in zip(("feature", "recon", "warped", "predict"), outs[0], outs[1]))
          + "; 3 vs 2: " + ", ".join(f"{int((a != b).sum())}" for a, b in zip(outs[1], outs[2])), flush=True)
    feat, ref = outs[2][0].to(torch.float16), outs[2][1]
