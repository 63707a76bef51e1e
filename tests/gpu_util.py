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
