"""Property / known-answer tests of the oracle's entropy arithmetic ("parity unpinned" part)."""
import numpy as np
import pytest
import torch

from oracle import entropy as E
from oracle import build as obuild


def test_scale_table_endpoints():
    t = E.get_scale_table()
    assert len(t) == 64 and abs(float(t[0]) - 0.11) < 1e-6 and abs(float(t[-1]) - 256) < 1e-3
    assert bool((t[1:] > t[:-1]).all())


@pytest.fixture(scope="module")
def gc():
    return E.gc_tables()


def test_gaussian_tables_are_valid_cdfs(gc):
    cdf, ln, off = gc["cdf"], gc["cdf_length"], gc["offset"]
    assert cdf.shape == (64, 3133)
    assert ln.min() == 5 and ln.max() == 3133 and off.min() == -1565 and off.max() == -1
    for i in range(64):
        row = cdf[i, : ln[i]]
        assert row[0] == 0 and row[-1] == 65536
        assert np.all(np.diff(row) > 0), f"row {i} has an empty bin"
        assert np.all(cdf[i, ln[i]:] == 0)


def test_build_indexes_monotone(gc):
    table = gc["scale_table"]
    s = torch.tensor([-1.0, 0.0, 0.11, 0.1100001, 0.5, 1.0, 200.0, 256.0, 1e6])
    idx = E.gc_build_indexes(s, table)
    assert idx[0] == 0 and idx[2] == 0 and idx[-1] == 63 and idx[-2] == 63
    assert bool((idx[1:] >= idx[:-1]).all())
    assert int(E.gc_build_indexes(table[62:63] * 1.0001, table)[0]) == 63


def test_pmf_to_quantized_cdf_python_equals_c():
    lib = obuild.load()
    assert lib is not None, "gcc build of oracle/rans_ref.c failed"
    rng = np.random.default_rng(1)
    for n in (2, 3, 17, 400):
        p = rng.random(n).astype(np.float32) ** 8     # many tiny probabilities -> exercises the steal loop
        p /= p.sum()
        a, b = E.pmf_to_quantized_cdf(p), lib.pmf_to_quantized_cdf(p)
        assert np.array_equal(a, b)
        assert a[0] == 0 and a[-1] == 65536 and np.all(np.diff(a) > 0)
    with pytest.raises(ValueError):
        E.pmf_to_quantized_cdf(np.array([0.5, -0.1], dtype=np.float32))
    with pytest.raises(ValueError):
        E.pmf_to_quantized_cdf(np.zeros(4, dtype=np.float32))


def _random_stream(gc, n, seed, escapes=True):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 64, n).astype(np.int32)
    sig = gc["scale_table"].numpy()[idx]
    sym = np.round(rng.standard_normal(n) * sig).astype(np.int32)
    if escapes and n >= 8:
        sym[::7] += 40000          # positive escapes (several nibbles)
        sym[3::7] -= 70000         # negative escapes
        sym[5::11] = gc["offset"][idx[5::11]] - 1   # just below the table
    return sym, idx


@pytest.mark.parametrize("n", [0, 1, 2, 257, 5000])
def test_rans_roundtrip_python_and_c(gc, n):
    sym, idx = _random_stream(gc, n, seed=n)
    args = (gc["cdf"], gc["cdf_length"], gc["offset"])
    py = E.rans_encode_py(sym, idx, *args)
    c = E.rans_encode(sym, idx, *args)
    assert py == c and len(py) % 4 == 0 and len(py) >= 8
    for force in (True, False):
        d = E.RansDecoder(py, force_python=force)
        k = n // 3       # decode in two calls: the decoder state persists across calls (image_model.py:288)
        out = np.concatenate([d.decode(idx[:k], *args), d.decode(idx[k:], *args)])
        assert np.array_equal(out, sym)


def test_rans_known_answer_empty_stream(gc):
    """No symbols: flush emits just the initial state L = 2^31 as two little-endian 32-bit words."""
    b = E.rans_encode_py(np.zeros(0, np.int32), np.zeros(0, np.int32), gc["cdf"], gc["cdf_length"], gc["offset"])
    assert b == (1 << 31).to_bytes(8, "little")


def test_entropy_bottleneck_tables_and_roundtrip():
    torch.manual_seed(3)
    from deepsvc_amd.entropy_models import EntropyBottleneck
    eb = EntropyBottleneck(6)
    sd = {k: v.detach() for k, v in eb.state_dict().items()}
    t = E.eb_tables(sd, "")
    assert t["cdf"].shape[0] == 6 and np.all(t["cdf_length"] == 23) and np.all(t["offset"] == -10)
    for i in range(6):
        row = t["cdf"][i, : t["cdf_length"][i]]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) > 0)
    z = torch.randn(2, 6, 3, 5) * 6
    z[0, 0, 0, 0], z[1, 2, 1, 1] = 55.0, -41.0     # outside the tabulated support -> escape coded
    strings = E.eb_compress(sd, "", t, z)
    assert len(strings) == 2
    z_hat = E.eb_decompress(sd, "", t, strings, (3, 5))
    med = E.eb_medians(sd, "").reshape(1, -1, 1, 1)
    assert torch.equal(z_hat, torch.round(z - med) + med)
    zh, lik = E.eb_forward(sd, "", z)
    assert torch.equal(zh, z_hat) and float(lik.min()) >= 1e-9 and float(lik.max()) <= 1.0
