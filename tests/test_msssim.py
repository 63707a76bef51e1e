"""MS-SSIM (pytorch-msssim 0.2.1 semantics, PARITY UNPINNED -- the package is absent): properties of
the oracle restatement on the CPU, and the HIP implementation (through the C ABI) against it on the GPU."""
import pytest
import torch

from oracle import msssim as O


def _pair(B, C, H, W, noise, seed=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand((B, C, H // 8 + 2, W // 8 + 2), generator=g), size=(H, W), mode="bilinear",
                                           align_corners=False)
    x = base.clamp(0, 1)
    y = (base + noise * torch.randn((B, C, H, W), generator=g)).clamp(0, 1)
    return x.contiguous(), y.contiguous()


def test_oracle_properties():
    x, _ = _pair(1, 3, 176, 208, 0.0)
    assert abs(float(O.ms_ssim(x, x, 1.0)) - 1.0) < 1e-6                      # identity
    vals = []
    for noise in (0.01, 0.05, 0.2):
        _, y = _pair(1, 3, 176, 208, noise)
        v = float(O.ms_ssim(x, y, 1.0))
        assert 0.0 < v < 1.0
        assert abs(v - float(O.ms_ssim(y, x, 1.0))) < 1e-6                    # symmetric
        vals.append(v)
    assert vals[0] > vals[1] > vals[2]                                        # more distortion, lower score
    xo, yo = _pair(2, 3, 181, 333, 0.05)                                      # odd sizes: padded pooling
    per_image = O.ms_ssim(xo, yo, 1.0, size_average=False)
    assert per_image.shape == (2,) and abs(float(per_image.mean()) - float(O.ms_ssim(xo, yo, 1.0))) < 1e-6
    # data_range only enters through C1, C2: scaling both images and the range leaves the score unchanged
    assert abs(float(O.ms_ssim(x * 255, _pair(1, 3, 176, 208, 0.05)[1] * 255, 255.0)) - vals[1]) < 1e-4
    with pytest.raises(AssertionError):
        O.ms_ssim(torch.rand(1, 3, 160, 200), torch.rand(1, 3, 160, 200), 1.0)   # needs > 160 on the smaller side


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,noise", [(1, 3, 176, 208, 0.05), (2, 3, 181, 333, 0.1), (1, 1, 200, 161, 0.02), (1, 3, 1088, 1920, 0.05)])
def test_gpu_ms_ssim_matches_oracle(B, C, H, W, noise):
    from deepsvc_amd.metrics import ms_ssim
    x, y = _pair(B, C, H, W, noise, seed=H)
    want = O.ms_ssim(x, y, 1.0, size_average=False)
    got = ms_ssim(x.cuda(), y.cuda(), 1.0, size_average=False).cpu()
    print(f"ms_ssim {B}x{C}x{H}x{W}: gpu {got.tolist()} oracle {want.tolist()}")
    assert torch.allclose(got, want, rtol=0, atol=2e-5)
    assert abs(float(ms_ssim(x.cuda(), x.cuda(), 1.0)) - 1.0) < 1e-5
    with pytest.raises(AssertionError):
        ms_ssim(torch.rand(1, 3, 160, 200).cuda(), torch.rand(1, 3, 160, 200).cuda(), 1.0)


@pytest.mark.gpu
def test_forward_msssim_contract():
    """video_model.py:96-135: same graph as forward(), MS-SSIM in place of the three MSEs."""
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    m = synth.make_model("W0"); m.update(force=True); m = m.cuda()
    size = 192
    fr = [f.cuda() for f in synth.make_frames(size, size, 2)]
    sm = synth.make_sm_fea(size, size, 1).cuda()
    a = m.forward(fr[0], fr[1], sm, None)
    b = m.forward_msssim(fr[0], fr[1], sm, None)
    assert len(b) == 8 and torch.equal(a[0], b[0]) and torch.equal(a[1].float(), b[1].float())
    for i in (5, 6, 7):
        assert float(a[i]) == float(b[i])
    want = O.ms_ssim(b[0].float().cpu(), fr[1].cpu(), 1.0)
    assert abs(float(b[2]) - float(want)) < 2e-5
    assert all(0.0 <= float(v) <= 1.0 for v in b[2:5])
