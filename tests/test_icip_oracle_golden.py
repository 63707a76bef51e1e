"""I-frame codec (image_model.py:ICIP2020ResB): the oracle restatement (oracle/icodec.py) against
golden vectors produced by the REFERENCE class itself (tests/golden/make_golden_icip.py), plus the product's
state_dict contract.  CPU only."""
import os

import numpy as np
import pytest
import torch

from deepsvc_amd import synth
from oracle.icodec import ICIPOracle
from test_oracle_golden import _check

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def icip_oracle():
    o = ICIPOracle(synth.make_icodec("W0").state_dict())
    o.update(force=True)
    return o


@pytest.mark.parametrize("size", [64, 128])
def test_icip_oracle_matches_reference(icip_oracle, size):
    g = np.load(os.path.join(GOLD, f"icip_ref_{size}.npz"))
    tag = f"s{size}"
    x = synth.make_frames(size, size, 1)[0]
    taps = {}
    out = icip_oracle.forward(x, taps=taps)
    pre = f"{tag}/fwd/"
    _check(g, pre + "g_a@0", taps["y"])
    _check(g, pre + "h_a@0", taps["z"])
    _check(g, pre + "h_mean_s@0", taps["latent_means"])
    _check(g, pre + "h_scale_s@0", taps["latent_scales"])
    for i in (0, 4, 5, 9):
        _check(g, pre + f"cc_mean_transforms.{i}@0", taps["mu"][i])
        _check(g, pre + f"cc_scale_transforms.{i}@0", taps["scale"][i])
    _check(g, pre + "g_s@0", out["x_hat"])
    _check(g, pre + "x_hat", out["x_hat"])
    logs = [float(torch.log(out["likelihoods"]["y"]).sum()), float(torch.log(out["likelihoods"]["z"]).sum())]
    np.testing.assert_allclose(logs, g[pre + "scalars"], rtol=1e-4, atol=1e-3)
    # real bitstream path: byte-identical strings, identical reconstruction
    enc = icip_oracle.compress(x)
    assert enc["strings"][0][0] == g[f"{tag}/enc/y"].tobytes()
    assert enc["strings"][1][0] == g[f"{tag}/enc/z"].tobytes()
    assert tuple(enc["shape"]) == tuple(g[f"{tag}/enc/shape"])
    dec = icip_oracle.decompress(enc["strings"], enc["shape"])
    _check(g, f"{tag}/dec/x_hat", dec["x_hat"])
    assert float(dec["x_hat"].min()) >= 0.0 and float(dec["x_hat"].max()) <= 1.0
    nbytes = len(enc["strings"][0][0]) + len(enc["strings"][1][0])
    assert nbytes > 100, "calibrated weights must exercise the entropy path"


def test_icip_state_dict_contract():
    """Same non-entropy keys and shapes as the reference class (recorded when the goldens were made); 57.9 M parameters."""
    want = {}
    with open(os.path.join(GOLD, "icip_state_dict_keys.txt")) as f:
        for line in f:
            k, shp = line.split(" ", 1)
            want[k] = tuple(int(v) for v in shp.strip()[1:-1].split(",") if v.strip())
    sd = synth.make_icodec("W0").state_dict()
    got = {k: tuple(v.shape) for k, v in sd.items() if "entropy_bottleneck" not in k and "gaussian_conditional" not in k}
    assert got == want
    g = np.load(os.path.join(GOLD, "icip_ref_64.npz"))
    assert int(g["meta/n_keys_non_entropy"][0]) == len(want)
