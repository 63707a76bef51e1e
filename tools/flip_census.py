"""Where does the fp32 HIP path differ from the reference at 1088x1920?  (GPU box; fixture tests/golden/deepsvc_ref_1080p.npz)

Prints, per frame and codec: differing symbols / indexes, the ROOT decisions (the reference's own decision was within ~1e-5 of
flipping: tests/parity_util.explain_plane_mismatches), unexplained mismatches, byte counts; then what the GPU decoder makes of the
REFERENCE's strings and of its own strings, against the reference's decoded tensors (strided samples, block means, and the share of
16x16 blocks whose mean is off by more than 1e-4).

    python tools/flip_census.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as PU  # noqa: E402
from deepsvc_amd import ops, synth  # noqa: E402

DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "deepsvc_ref_1080p.npz"))


def enc_of(pre, nm):
    return {"strings": [[g[f"{pre}/enc/{nm}_y"].tobytes()], [g[f"{pre}/enc/{nm}_z"].tobytes()]], "shape": tuple(int(v) for v in g[f"{pre}/enc/{nm}_shape"])}


def block_report(key, t):
    a = t.detach().float().cpu().numpy()
    blk = int(g[key + "/blk"][0])
    *lead, H, W = a.shape
    bm = a.astype(np.float64).reshape(*lead, H // blk, blk, W // blk, blk).mean(axis=(-3, -1))
    d = np.abs(bm - g[key + "/blockmean"])
    e_s, e_b, _ = PU.golden_full_errors(g, key, t)
    return f"samples {e_s:.2e}, block means {e_b:.2e}, blocks off by > 1e-4: {int((d > 1e-4).sum())}/{d.size}, max block diff {d.max():.2e}"


def report(tag, r):
    print(f"  {tag}: differing symbols {r['symbols']} / indexes {r['indexes']} of {r['total']} (lines {r['lines']}); roots {len(r['roots'])}, unexplained {len(r['unexplained'])}")
    for key, kind, m in r["roots"][:20]:
        print(f"    root {kind} at (c, y, x) = {key}: reference margin {m:.2e}")
    for u in r["unexplained"][:20]:
        print(f"    UNEXPLAINED {u}")


ops.set_precision("fp32")
m = synth.make_model("W0"); m.update(force=True); m = m.to(DEV)
fr = [f.to(DEV) for f in synth.make_frames(1088, 1920, 3)]
ref, feat = fr[0], None
for t in (1, 2):
    pre, sm = f"s1080/f{t}", synth.make_sm_fea(1088, 1920, t).to(DEV)
    mv, res = m.compress(ref, fr[t], sm, feat)
    mv, res = dict(mv.items()), dict(res.items())
    print(f"frame {t}")
    for nm, enc, codec in (("mv", mv, m.mv_codec), ("res", res, m.res_codec)):
        _, ysym, yidx = PU.product_symbol_planes(codec, 68, 120, DEV)
        r = PU.explain_plane_mismatches(g, f"{pre}/enc/{nm}", ysym, yidx, codec.slice_depth, codec.max_support_slices)
        want = enc_of(pre, nm)
        print(f"  {nm}: bytes {len(enc['strings'][0][0])}+{len(enc['strings'][1][0])} vs reference {len(want['strings'][0][0])}+{len(want['strings'][1][0])}, "
              f"identical {enc['strings'][0][0] == want['strings'][0][0]} / {enc['strings'][1][0] == want['strings'][1][0]}")
        report(nm, r)
    for name, (e_mv, e_res) in (("reference's strings", (enc_of(pre, "mv"), enc_of(pre, "res"))), ("own strings", (mv, res))):
        try:
            out = m.decompress(ref, e_mv, e_res, sm, feat)
            torch.cuda.synchronize()
            for j, nm in ((1, "recon"), (2, "warped"), (3, "predict"), (0, "feature")):
                print(f"  decode of {name}: {nm}: " + block_report(f"{pre}/dec/{nm}", out[j]))
            if name.startswith("reference"):
                out_ref = out
        except Exception as e:                     # a desynchronised range decoder may also run out of stream
            print(f"  decode of {name}: FAILED {type(e).__name__}: {e}")
            out_ref = out if name != "reference's strings" else None
    # chain on the decode of the reference's strings when it worked, else on the own decode
    use = out_ref if out_ref is not None else out
    feat, ref = use[0], use[1]

im = synth.make_icodec("W0"); im.update(force=True); im = im.to(DEV)
enc = im.compress(fr[0])
_, ysym, yidx = PU.product_symbol_planes(im, 68, 120, DEV)
r = PU.explain_plane_mismatches(g, "icip1080/enc/y", ysym, yidx, im.slice_depth, im.max_support_slices)
print(f"icip: bytes {len(enc['strings'][0][0])}+{len(enc['strings'][1][0])} vs reference {g['icip1080/enc/y'].size}+{g['icip1080/enc/z'].size}")
report("icip", r)
fs = g["icip1080/enc/y_sym_fragile"]
print("  fragile symbol decisions listed:", len(fs), " with relative margin <= 2e-5:", int((fs[:, 3] / (1 + fs[:, 4]) <= 2e-5).sum()),
      " median |y|+|mu| there:", float(np.median(fs[:, 4])) if len(fs) else None)
for name, strings in (("reference's strings", [[g["icip1080/enc/y"].tobytes()], [g["icip1080/enc/z"].tobytes()]]), ("own strings", enc["strings"])):
    try:
        x = im.decompress(strings, enc["shape"])["x_hat"]
        print(f"  decode of {name}: x_hat: " + block_report("icip1080/dec/x_hat", x))
    except Exception as e:
        print(f"  decode of {name}: FAILED {type(e).__name__}: {e}")
