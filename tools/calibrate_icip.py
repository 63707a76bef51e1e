"""Make deepsvc_amd/data/icip_w0_calibration.json: per-tensor multipliers that turn the seeded default init of
`ICIP2020ResB()` into a weight set that exercises the whole I-frame path (same recipe as tools/calibrate_w0.py:
std(y) ~ 1, std(z) ~ 1, per slice std(mu) ~ 0.5 and sigma ~ 0.6 +- 0.35, x_hat ~ 0.5 +- 0.2).

Runs on the CPU through the oracle (test infrastructure); the product only consumes the JSON.
    python tools/calibrate_icip.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepsvc_amd import synth  # noqa: E402
from oracle import entropy as E  # noqa: E402
from oracle import icodec as C  # noqa: E402
from oracle.nets import P  # noqa: E402
from tools.calibrate_w0 import Calib  # noqa: E402


@torch.no_grad()
def calibrate(seed_set="W0", size=128):
    model = synth.make_icodec(seed_set, calibrated=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cal = Calib(sd)
    q = P(sd)
    x = synth.make_frames(size, size, 1)[0]
    y = C.g_a(q, x)
    cal.scale_conv("g_a.12.", 1.0 / float(y.std()))
    y = C.g_a(q, x)
    z = C.h_a(q, y)
    cal.scale_conv("h_a.4.", 1.0 / float(z.std()))
    z = C.h_a(q, y)
    med = E.eb_medians(sd, "entropy_bottleneck.").reshape(1, -1, 1, 1)
    z_hat = torch.round(z - med) + med
    scales, means = C.h_s(q, "h_scale_s", z_hat), C.h_s(q, "h_mean_s", z_hat)
    y_hat_slices = []
    for i, y_slice in enumerate(y.chunk(C.NUM_SLICES, 1)):
        support = y_hat_slices[:C.MAX_SUPPORT]
        ms = torch.cat([means] + support, 1)
        ss = torch.cat([scales] + support, 1)
        mu = C.stack3(q.sub(f"cc_mean_transforms.{i}"), ms)
        cal.scale_conv(f"cc_mean_transforms.{i}.4.", 0.5 / float(mu.std()))
        mu = C.stack3(q.sub(f"cc_mean_transforms.{i}"), ms)
        sc = C.stack3(q.sub(f"cc_scale_transforms.{i}"), ss)
        cal.scale_conv(f"cc_scale_transforms.{i}.4.", 0.35 / float(sc.std()))
        sc = C.stack3(q.sub(f"cc_scale_transforms.{i}"), ss)
        cal.scale(f"cc_scale_transforms.{i}.4.bias", 1.0, 0.6 - float(sc.mean()))
        y_hat_slices.append(C._lrp(q, i, ms, torch.round(y_slice - mu) + mu))
    x_hat = C.g_s(q, torch.cat(y_hat_slices, 1))
    cal.scale_conv("g_s.12.", 0.2 / float(x_hat.std()))
    x_hat = C.g_s(q, torch.cat(y_hat_slices, 1))
    cal.scale("g_s.12.bias", 1.0, 0.5 - float(x_hat.mean()))
    x_hat = C.g_s(q, torch.cat(y_hat_slices, 1))
    print(f"std(y)={float(y.std()):.2f} std(z)={float(z.std()):.2f} x_hat mean {float(x_hat.mean()):.3f} std {float(x_hat.std()):.3f}")
    return cal.spec


if __name__ == "__main__":
    spec = calibrate("W0")
    out = {"recipe": "tools/calibrate_icip.py: seed-0 default init of ICIP2020ResB() x these multipliers (tensor <- tensor*mul + add)",
           "tensors": {k: {"mul": float(v["mul"]), "add": float(v["add"])} for k, v in sorted(spec.items())}}
    with open(synth.ICIP_CALIBRATION, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {synth.ICIP_CALIBRATION} ({len(out['tensors'])} tensors)")
