"""What does v_mfma_f32_32x32x2_f32 compute, bit for bit?  A one-chunk fp32 1x1 conv (16 -> 32 channels) through the general kernel
against exact-arithmetic emulations of the candidate orders:
  H1  acc = fma(a1, b1, fma(a0, b0, acc))        (k slot 0 first, one rounding per product-add)
  H2  acc = fma(a0, b0, fma(a1, b1, acc))        (k slot 1 first)
  H3  acc = round(acc + a0*b0 + a1*b1)           (one rounding per instruction)
  H4  acc = round(round(acc + round(a0*b0)) + round(a1*b1))   (unfused)
python tools/mfma_f32_order_probe.py"""
import os, sys
from fractions import Fraction
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402


def rnd(fr):
    """Fraction -> nearest float32 (ties to even), exactly."""
    if fr == 0:
        return np.float32(0.0)
    c = np.float32(float(fr))                       # a candidate within an ulp or so
    best, bd = None, None
    for cand in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
        d = abs(Fraction(float(cand)) - fr)
        if bd is None or d < bd or (d == bd and (int(np.float32(cand).view(np.uint32)) & 1) == 0):
            best, bd = np.float32(cand), d
    return best


F = lambda x: Fraction(float(x))
ops.set_precision("fp32")
torch.manual_seed(3)
dev = "cuda:0"
m = Conv2d(16, 32, 1, 1, 0).to(dev)
x = torch.randn(1, 16, 8, 32)
got = m.run([ops.Act(x.permute(0, 2, 3, 1).contiguous().to(dev))], out_dtype=torch.float32).nchw().float().cpu().numpy()
w, b, xs = m.weight.detach().cpu().numpy()[:, :, 0, 0], m.bias.detach().cpu().numpy(), x.numpy()
pairs = [(4 * ks + j, 8 + 4 * ks + j) for ks in range(2) for j in range(4)]
hits = {"H1": 0, "H2": 0, "H3": 0, "H4": 0}
n = 0
for co in range(0, 32, 3):
    for px in range(0, 32, 2):
        for py in (0, 5):
            a = xs[0, :, py, px]
            acc = {k: np.float32(0.0) for k in hits}
            for c0, c1 in pairs:
                p0, p1 = F(a[c0]) * F(w[co, c0]), F(a[c1]) * F(w[co, c1])
                acc["H1"] = rnd(F(rnd(F(acc["H1"]) + p0)) + p1)
                acc["H2"] = rnd(F(rnd(F(acc["H2"]) + p1)) + p0)
                acc["H3"] = rnd(F(acc["H3"]) + p0 + p1)
                acc["H4"] = rnd(F(rnd(F(acc["H4"]) + F(rnd(p0)))) + F(rnd(p1)))
            n += 1
            for k in hits:
                hits[k] += int(rnd(F(acc[k]) + F(b[co])).view(np.uint32) == got[0, co, py, px].view(np.uint32))
print(f"{n} outputs; bit-equal to the kernel's result:", hits)
