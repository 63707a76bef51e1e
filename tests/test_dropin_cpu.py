"""The drop-in boundary on the CPU (no compute): `dropin/` ahead on sys.path resolves the import lines of the reference's
test_video.py:8-11 to the MI355X-native modules; host helpers (read_image / pad / crop / cal_psnr); the checkpoint contract
of test_video.py:230-236 (`load_state_dict(torch.load(...)["state_dict"])` of a model that has run `update()`), legacy key
spellings (utils.py:105-125) and the aux losses (video_model.py:169-177) against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_through_dropin(tmp_path):
    """Executes, in a fresh interpreter with dropin/ first on sys.path, the reference harness's own import lines for the path
    (test_video.py:8,10,11; line 9 imports the out-of-scope semantic layer, line 7 the absent pytorch_msssim) and the host
    helpers on a real PNG."""
    code = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'dropin')!r}); sys.path.insert(1, {ROOT!r})
from image_model import ICIP2020ResB
from video_model import DeepSVC
from modules import read_image, cal_psnr, crop, pad
import numpy as np, torch
from PIL import Image
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(70, 100, 3), dtype=np.uint8)
Image.fromarray(img).save({str(tmp_path / 'f.png')!r})
x = read_image({str(tmp_path / 'f.png')!r})
assert x.dtype == torch.float32 and tuple(x.shape) == (3, 70, 100)
assert torch.equal(x, torch.from_numpy(img).permute(2, 0, 1).float() / 255.0)
x = x.unsqueeze(0)
p = pad(x, 64)
assert tuple(p.shape) == (1, 3, 128, 128) and torch.equal(crop(p, (70, 100)), x)
assert float(p[0, :, :29].abs().max()) == 0.0 and float(p[0, :, :, :14].abs().max()) == 0.0      # centred zero pad (modules.py:76-89)
y = (x + 0.01).clamp(0, 1)
assert abs(cal_psnr(x, y) - (-10 * np.log10(float(((x - y) ** 2).mean())))) < 1e-9
import deepsvc_amd.video_model, deepsvc_amd.image_model
assert DeepSVC is deepsvc_amd.video_model.DeepSVC and ICIP2020ResB is deepsvc_amd.image_model.ICIP2020ResB
try:
    read_image({str(tmp_path / 'missing.png')!r})
except AssertionError:
    pass
else:
    raise SystemExit("read_image of a missing file must fail like the reference's assert (modules.py:66)")
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.fixture(scope="module")
def updated_model():
    from deepsvc_amd import synth
    m = synth.make_model("W0")
    m.update(force=True)
    return m


def test_checkpoint_round_trip_with_sized_entropy_buffers(updated_model, tmp_path):
    """test_video.py:230-236: a checkpoint saved from a model that has run update() carries sized `_quantized_cdf / _offset /
    _cdf_length / scale_table` buffers; a fresh DeepSVC() (whose buffers are empty) must load it strictly
    (video_model.py:184-191, image_model.py:304-317) and code with the LOADED tables without another update()."""
    from deepsvc_amd.video_model import DeepSVC
    path = str(tmp_path / "ckpt.pth")
    torch.save({"state_dict": updated_model.state_dict(), "epoch": 3}, path)
    fresh = DeepSVC()
    assert fresh.mv_codec.entropy_bottleneck._quantized_cdf.numel() == 0
    ckpt = torch.load(path, map_location="cpu")
    result = fresh.load_state_dict(ckpt["state_dict"])
    assert not result.missing_keys and not result.unexpected_keys
    fresh.eval()
    a, b = updated_model.state_dict(), fresh.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for codec in ("mv_codec", "res_codec"):
        for em in ("entropy_bottleneck", "gaussian_conditional"):
            ta, tb = getattr(getattr(updated_model, codec), em).host_tables(), getattr(getattr(fresh, codec), em).host_tables()
            assert np.array_equal(ta.cdf, tb.cdf) and np.array_equal(ta.cdf_length, tb.cdf_length) and np.array_equal(ta.offset, tb.offset)
    assert fresh.update(force=False) is False          # tables present: nothing to rebuild
    assert fresh.update(force=True) is True


def test_legacy_key_spellings_load(updated_model):
    """utils.py:105-125: `module.` prefixes and `entropy_bottleneck._matrices.N / _biases.N / _factors.N`."""
    from deepsvc_amd.video_model import DeepSVC
    legacy = {}
    for k, v in updated_model.state_dict().items():
        for new, old in (("_matrix", "_matrices."), ("_bias", "_biases."), ("_factor", "_factors.")):
            tag = "entropy_bottleneck." + new
            if tag in k:
                k = k.replace(tag, "entropy_bottleneck." + old)
        legacy["module." + k] = v
    assert any("_matrices.3" in k for k in legacy)
    fresh = DeepSVC()
    fresh.load_state_dict(legacy)
    for k, v in updated_model.state_dict().items():
        assert torch.equal(v, fresh.state_dict()[k]), k


def test_aux_losses_match_oracle(updated_model, oracle_w0):
    """video_model.py:169-177 vs the oracle's restatement of EntropyBottleneck.loss() (parity unpinned: compressai)."""
    from oracle import entropy as E
    mv = float(E.eb_loss(oracle_w0.sd, "mv_codec.entropy_bottleneck."))
    res = float(E.eb_loss(oracle_w0.sd, "res_codec.entropy_bottleneck."))
    assert abs(float(updated_model.mv_aux_loss()) - mv) <= 1e-4 * abs(mv)
    assert abs(float(updated_model.res_aux_loss()) - res) <= 1e-4 * abs(res)
    assert abs(float(updated_model.aux_loss()) - (mv + res)) <= 1e-4 * abs(mv + res)
    assert mv > 0 and res > 0
    # the loss is differentiable w.r.t. the `.quantiles` parameters only in the way the trainers use it (Learner.py:1681-1682)
    names = [n for n, _ in updated_model.named_parameters() if n.endswith(".quantiles")]
    assert sorted(names) == ["mv_codec.entropy_bottleneck.quantiles", "res_codec.entropy_bottleneck.quantiles"]
