"""Make deepsvc_amd/data/fcodec_w0_calibration.json: per-tensor multipliers that turn the seeded default init of the semantic
layer's feature codec (`StudentLayer1().layer1` = cFeatureCompress(N=72)) into a weight set that exercises its whole path
(same recipe as tools/calibrate_w0.py: std(y) ~ 1, std(z) ~ 1, per slice std(mu) ~ 0.5 and sigma ~ 0.6 +- 0.35, decoded
feature x_hat ~ 0.3 +- 0.8 so that the ReLU behind RefineNet1 passes about two thirds of it).

Runs on the CPU through the oracle (test infrastructure); the product only consumes the JSON.
    python tools/calibrate_fcodec.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepsvc_amd import synth  # noqa: E402
from oracle import entropy as E  # noqa: E402
from oracle import fcodec as C  # noqa: E402
from oracle.nets import P  # noqa: E402
from tools.calibrate_w0 import Calib  # noqa: E402


@torch.no_grad()
def calibrate(seed_set="W0", size=128):
    model = synth.make_student_layer1(seed_set, calibrated=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cal = Calib(sd)
    q = P(sd).sub("layer1")
    pre = "layer1."
    x, supp = synth.make_semantic_input(size, size, 1)
    fea = supp[-1:]
    xin = C.analysis_input(q, x, fea)
    y = C.g_a(q, xin)
    cal.scale_conv(pre + "g_a.9.", 1.0 / float(y.std()))
    y = C.g_a(q, xin)
    z = C.h_a(q, y)
    cal.scale_conv(pre + "h_a.6.", 1.0 / float(z.std()))
    z = C.h_a(q, y)
    med = E.eb_medians(sd, pre + "entropy_bottleneck.").reshape(1, -1, 1, 1)
    z_hat = torch.round(z - med) + med
    scales, means = C.h_s(q, "h_scale_s", z_hat), C.h_s(q, "h_mean_s", z_hat)
    sm = C.fea_convert(q, fea)
    cal.scale_conv(pre + "fea_convert.2.", 1.0 / float(sm.std()))          # side feature at the latents' scale
    sm = C.fea_convert(q, fea)
    y_hat_slices = []
    for i, y_slice in enumerate(y.chunk(C.NUM_SLICES, 1)):
        support = y_hat_slices[:C.MAX_SUPPORT]
        ms = torch.cat([means, sm] + support, 1)
        ss = torch.cat([scales, sm] + support, 1)
        mu = C.stack(q.sub(f"cc_mean_transforms.{i}"), ms, 4)
        cal.scale_conv(f"{pre}cc_mean_transforms.{i}.6.", 0.5 / float(mu.std()))
        mu = C.stack(q.sub(f"cc_mean_transforms.{i}"), ms, 4)
        sc = C.stack(q.sub(f"cc_scale_transforms.{i}"), ss, 4)
        cal.scale_conv(f"{pre}cc_scale_transforms.{i}.6.", 0.35 / float(sc.std()))
        sc = C.stack(q.sub(f"cc_scale_transforms.{i}"), ss, 4)
        cal.scale(f"{pre}cc_scale_transforms.{i}.6.bias", 1.0, 0.6 - float(sc.mean()))
        y_hat = torch.round(y_slice - mu) + mu
        lrp_in = torch.cat([ms, y_hat], 1)
        l = C.stack(q.sub(f"lrp_transforms.{i}"), lrp_in, 3)
        cal.scale_conv(f"{pre}lrp_transforms.{i}.4.", 0.5 / float(l.std()))
        y_hat_slices.append(C._lrp(q, i, ms, y_hat))
    y_hat = torch.cat(y_hat_slices, 1)
    x1 = C.g_s(q, y_hat)
    cal.scale_conv(pre + "g_s.6.", 1.0 / float(x1.std()))
    x1 = C.g_s(q, y_hat)
    xh = C.final(q, x1, fea)
    g = 0.8 / float(xh.std())
    # the last ResBottleneck adds onto its input: scale the conv that feeds the two blocks AND their output branches alike
    cal.scale_conv(pre + "g_s1.0.", g)
    xh = C.final(q, x1, fea)
    cal.scale(pre + "g_s1.3.conv3.bias", 1.0, 0.3 - float(xh.mean()))
    xh = C.final(q, x1, fea)
    print(f"std(y)={float(y.std()):.2f} std(z)={float(z.std()):.2f} x_hat mean {float(xh.mean()):.3f} std {float(xh.std()):.3f}, "
          f"ReLU passes {float((xh > 0).float().mean()):.2f}")
    # RefineNet1: correlations spread enough for a non-uniform softmax
    r = P(sd).sub("refine")
    rx = C.N.conv(r.sub("refine.2"), C.N.conv(r.sub("refine.0"), xh))
    print(f"refine feature std {float(rx.std()):.3f}")
    return cal.spec


if __name__ == "__main__":
    spec = calibrate("W0")
    out = {"recipe": "tools/calibrate_fcodec.py: seed-0 default init of StudentLayer1(N=72) x these multipliers (tensor <- tensor*mul + add)",
           "tensors": {k: v for k, v in sorted(spec.items())}}
    with open(synth.FCODEC_CALIBRATION, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {synth.FCODEC_CALIBRATION} ({len(out['tensors'])} tensors)")
