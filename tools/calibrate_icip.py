"""Make deepsvc_amd/data/icip_w0_calibration.json: the recipe that turns the seeded default init of `ICIP2020ResB()` into a synthetic
I-frame weight set whose key frame is a USABLE picture (round 4; VERDICT round 3 item 7 -- the round-1..3 set decoded at 11-13 dB and made
sigma a small difference of large terms: 0.4 % of the symbols differed between the CPU and the GPU at 1088x1920).

A random analysis / synthesis pair is not an auto-encoder, and no read-out of its last layer makes it one (tried: ridge-fitted last
transposed conv, near-identity residual blocks, fine quantisation -- 14 dB, the PSNR of the mean picture).  So the recipe GENERATES the eight
strided convs instead of scaling them (deepsvc_amd/synth.py::structured_tensor; nothing is stored but a kind name and a gain):
  * g_a.0 / .4 / .8: exact space-to-depth (the 1/8-resolution tensor holds every pixel of the 8x8 block in its 192 channels); g_a.12: the
    picture decimated by two, times G = 16, centred, in latent channels 0..191 (std ~3: symbols mostly in -8..8); channels 192..319 keep the
    seeded init (live, un-structured, std ~1);
  * g_s.0 / .4 / .8: the inverse depth-to-space steps; g_s.12: x2 bilinear interpolation back to full resolution;
  * the 18 residual blocks keep their seeded init with the branch's last 1x1 conv scaled by 0.05 (every kernel runs on real data; the branch is
    a ~1 % perturbation, so the chain stays well conditioned: CPU and GPU agree to ~1e-6 instead of ~3e-4);
  * hyper path, parameter stacks and LRP stacks: seeded init, last conv of each scaled / shifted as for the P-frame codecs -- std(z) ~ 1,
    per slice std(mu) ~ 0.3 std(y), sigma ~ 1.1 std(y - mu) with a spread of +-30 % (about ten table levels per slice, ~25 over the
    slices), |0.5 tanh(lrp)| ~ 0.1.
Result through the oracle (printed below): key frame ~29 dB at 256x256 (border-limited: the interpolation zero-pads), ~33 dB at 1088x1920.

Runs on the CPU through the oracle (test infrastructure); the product only consumes the JSON.
    python tools/calibrate_icip.py
"""
import json
import math
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

GAIN, EPS, RB_BRANCH, TAIL = 16.0, 0.02, 0.05, 3.0


def structured_spec():
    spec = {}

    def S(key, kind, **kw):
        spec[key] = {"structured": dict(kind=kind, **kw), "eps": EPS}
    S("g_a.0.weight", "s2d", n_in=3)
    S("g_a.4.weight", "s2d", n_in=12)
    S("g_a.8.weight", "s2d", n_in=48)
    S("g_a.12.weight", "s2d_dec", gain=GAIN)
    spec["g_a.12.weight"]["eps_tail"] = [192, TAIL]
    S("g_s.0.weight", "d2s", n_in=48, gain=1.0 / GAIN)
    S("g_s.4.weight", "d2s", n_in=12)
    S("g_s.8.weight", "d2s", n_in=3)
    S("g_s.12.weight", "bilinear", n_in=3)
    for net in ("g_a", "g_s"):
        for i in (0, 4, 8, 12):
            spec[f"{net}.{i}.bias"] = {"mul": EPS, "add": 0.0}
        for i in range(13):
            if i % 4:
                for leaf in ("weight", "bias"):
                    spec[f"{net}.{i}.conv3.{leaf}"] = {"mul": RB_BRANCH, "add": 0.0}
    spec["g_a.12.bias"] = {"mul": EPS, "add": [-0.5 * GAIN] * 192 + [0.0] * 128}      # centre the picture channels
    spec["g_s.0.bias"] = {"mul": EPS, "add": [0.5] * 48 + [0.0] * 144}
    return spec


@torch.no_grad()
def calibrate(seed_set="W0", size=256):
    model = synth.make_icodec(seed_set, calibrated=False)
    base = {k: v.detach().clone() for k, v in model.state_dict().items()}
    spec = structured_spec()
    sd = synth.apply_calibration(base, {"tensors": spec})
    cal = Calib(sd)
    cal.spec = spec
    q = P(sd)
    x = synth.make_frames(size, size, 1)[0]
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
        sy = float(y_slice.std())
        mu = C.stack3(q.sub(f"cc_mean_transforms.{i}"), ms)
        cal.scale_conv(f"cc_mean_transforms.{i}.4.", 0.3 * sy / float(mu.std()))
        mu = C.stack3(q.sub(f"cc_mean_transforms.{i}"), ms)
        target = 1.1 * float((y_slice - mu).std())
        sc = C.stack3(q.sub(f"cc_scale_transforms.{i}"), ss)
        cal.scale_conv(f"cc_scale_transforms.{i}.4.", 0.3 * target / float(sc.std()))
        sc = C.stack3(q.sub(f"cc_scale_transforms.{i}"), ss)
        cal.scale(f"cc_scale_transforms.{i}.4.bias", 1.0, target - float(sc.mean()))
        y_q = torch.round(y_slice - mu) + mu
        lrp_support = torch.cat([ms, y_q], 1) if i < C.MAX_SUPPORT else torch.cat([ms, y_q], 1)
        lrp = C.stack3(q.sub(f"lrp_transforms.{i}"), lrp_support)
        cal.scale_conv(f"lrp_transforms.{i}.4.", 0.2 / float(lrp.std()))
        y_hat_slices.append(C._lrp(q, i, ms, y_q))
    x_hat = C.g_s(q, torch.cat(y_hat_slices, 1))
    psnr = -10 * math.log10(float(((x_hat.clamp(0, 1) - x) ** 2).mean()))
    print(f"std(y[:192])={float(y[:, :192].std()):.2f} std(y[192:])={float(y[:, 192:].std()):.2f} std(z)={float(z.std()):.2f} key-frame PSNR at {size}x{size}: {psnr:.2f} dB")
    return cal.spec


if __name__ == "__main__":
    spec = calibrate("W0")
    out = {"recipe": "tools/calibrate_icip.py: seed-0 default init of ICIP2020ResB() x these entries (structured: generated by synth.structured_tensor + eps * init; "
                     "others tensor <- tensor*mul + add)",
           "tensors": {k: v for k, v in sorted(spec.items())}}
    with open(synth.ICIP_CALIBRATION, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {synth.ICIP_CALIBRATION} ({len(out['tensors'])} tensors)")
    # ---- the whole codec through the oracle with the file just written ----
    from oracle.icodec import ICIPOracle
    o = ICIPOracle(synth.make_icodec("W0").state_dict())
    o.update()
    for size in (128, 256):
        x = synth.make_frames(size, size, 1)[0]
        with torch.no_grad():
            enc = o.compress(x)
            dec = o.decompress(enc["strings"], enc["shape"])["x_hat"]
        nb = sum(len(s[0]) for s in enc["strings"])
        print(f"oracle {size}x{size}: {nb} B = {8.0 * nb / size / size:.3f} bpp, PSNR {-10 * math.log10(float(((dec - x) ** 2).mean())):.2f} dB")
