"""Oracle restatement of pytorch-msssim 0.2.1's `ms_ssim` (env.txt:58; call sites video_model.py:107,114,132 and
test_video.py:311,359-363).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the package is absent from /root/reference and from this
image and the reference holds no vectors for it; this is its published algorithm (Gaussian window 11 / sigma 1.5 applied
separably with no padding, K = (0.01, 0.03), five scales with weights .0448/.2856/.3001/.2363/.1333, 2x2 average pooling
with padding = size % 2 between scales, relu on the per-scale terms), checked by properties in tests/.
"""
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_1d(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, win):
    C = x.shape[1]
    k = win.reshape(1, 1, -1).repeat(C, 1, 1)
    out = F.conv2d(x, k.unsqueeze(-1), groups=C)          # along H
    return F.conv2d(out, k.unsqueeze(-2), groups=C)       # along W


def _ssim(X, Y, data_range, win, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(X, win), _filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter(X * X, win) - mu1_sq
    s2 = _filter(Y * Y, win) - mu2_sq
    s12 = _filter(X * Y, win) - mu1_mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X, Y, data_range=255, size_average=True):
    assert X.shape == Y.shape and min(X.shape[-2:]) > (11 - 1) * 2 ** 4
    win = _gauss_1d()
    w = torch.tensor(WEIGHTS, dtype=torch.float32)
    mcs = []
    for i in range(5):
        ssim_c, cs = _ssim(X, Y, data_range, win)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X, Y = F.avg_pool2d(X, kernel_size=2, padding=pad), F.avg_pool2d(Y, kernel_size=2, padding=pad)
    stack = torch.stack(mcs + [torch.relu(ssim_c)], 0)
    val = torch.prod(stack ** w.view(-1, 1, 1), 0)
    return val.mean() if size_average else val.mean(1)
