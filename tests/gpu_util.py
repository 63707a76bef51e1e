"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch


def report_close(name, got, want, rtol, atol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} != {tuple(want.shape)}"
    diff = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = diff > bound
    if bool(bad.any()) or not bool(torch.isfinite(got).all()):
        i = int(torch.argmax(diff - bound))
        idx = np.unravel_index(i, got.shape)
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{got.numel()} elements out of tolerance (rtol={rtol}, atol={atol}); "
            f"max|d|={float(diff.max()):.3e} at {idx}: got {float(got[idx]):.6f} want {float(want[idx]):.6f}; "
            f"ref rms={float(want.pow(2).mean().sqrt()):.3e} finite={bool(torch.isfinite(got).all())}")


def psnr(a, b):
    mse = float(((a.float() - b.float()) ** 2).mean())
    return 99.0 if mse == 0 else -10 * np.log10(mse)


HALF_ULP_F16 = 2.0 ** -11      # binary16: 11-bit significand -> a correctly rounded store is within 2^-11 relative of the fp32 value


def close_f16(name, got, want, out_f16=True, extra=0.0, both_f16=False):
    """An f16-MFMA conv result against an fp32 reference evaluated on the SAME f16-rounded operands (torch on the CPU, or the
    fp32-output launch of another kernel): the only legitimate differences are the fp32 summation order (<= 3e-5 of the tensor
    scale for the K <= 1.2k sums here) and, for f16 outputs, ONE correct rounding of the stored value (half an ulp, 2^-11
    relative).  A single wrong / missing filter tap is ~50x above this.  `extra` (fraction of the tensor scale) covers fused
    launches whose f16-rounded intermediate can round the other way than the emulation's for isolated elements."""
    scale = float(want.abs().max())
    rt = (2.0 if both_f16 else 1.0) * HALF_ULP_F16 * 1.02 if out_f16 else 1e-6    # two independently rounded f16 results: one ulp apart at most
    report_close(name, got, want, rt, (3e-5 + extra) * scale)
