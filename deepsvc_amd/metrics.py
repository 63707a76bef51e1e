"""Harness metrics on the GPU: MS-SSIM with pytorch-msssim 0.2.1 semantics, the function the
reference imports as `from pytorch_msssim import ms_ssim` (video_model.py:10, test_video.py:12) and calls as
`ms_ssim(x, y, data_range=1.0)` (video_model.py:107,114,132; test_video.py:311,359-363)."""
import torch

from . import ops
from .ops import to_act

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
WIN_SIZE = 11


def ms_ssim(X, Y, data_range=255, size_average=True, weights=None):
    """X, Y: [B,C,H,W] device tensors (or NHWC `Act`s).  -> 0-dim tensor (size_average) or [B].
    Five scales: per scale the Gaussian-window means of the contrast-structure term (ssim itself on the last one),
    relu'd, raised to the scale's weight and multiplied; 2x2 average pooling (padding = size % 2) between scales."""
    x, y = to_act(X), to_act(Y)
    if (x.B, x.H, x.W, x.C) != (y.B, y.H, y.W, y.C):
        raise ValueError("Input images should have the same dimensions")
    if min(x.H, x.W) <= (WIN_SIZE - 1) * 2 ** 4:
        raise AssertionError("Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((WIN_SIZE - 1) * 2 ** 4))
    w = torch.tensor(MS_WEIGHTS if weights is None else weights, dtype=torch.float32, device=x.device)
    levels = w.numel()
    terms = []
    for i in range(levels):
        ssim_c, cs = ops.ssim_level(x, y, data_range)
        if i < levels - 1:
            terms.append(torch.relu(cs))
            x, y = ops.avgpool2_pad(x), ops.avgpool2_pad(y)
    terms.append(torch.relu(ssim_c))
    stack = torch.stack(terms, 0).reshape(levels, -1)                     # (level, batch*channel): 15 numbers for one RGB frame
    val = torch.prod(stack ** w.reshape(-1, 1), 0).reshape(y.B, -1)
    return val.mean() if size_average else val.mean(1)
