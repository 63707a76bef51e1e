"""The multi-stream range coder's FORMAT on the CPU (no GPU): the plain-Python restatement round-trips (escapes of both signs included), costs
what the tables say plus 6 bytes per stream, and the string layout packs / unpacks / refuses malformed input."""
import numpy as np
import pytest

import ms_ref as R
from deepsvc_amd import entropy_models as EM
from oracle import entropy as E


def _tables():
    t = E.gaussian_tables()
    return np.asarray(t.cdf, dtype=np.int64), np.asarray(t.cdf_length), np.asarray(t.offset)


def _tables_fallback():
    from deepsvc_amd.entropy_models import GaussianConditional
    from deepsvc_amd.modules import get_scale_table
    g = GaussianConditional(None)
    g.update_scale_table(get_scale_table(), force=True)
    ht = g.host_tables()
    return ht.cdf.astype(np.int64), ht.cdf_length, ht.offset


def _symbols(n, cdf_length, offset, rng):
    idx = rng.integers(0, 64, n)
    sigma = 0.11 * (256 / 0.11) ** (idx / 63.0)
    sym = np.rint(rng.normal(0, 1, n) * sigma).astype(np.int64)
    esc = rng.random(n) < 0.01                                  # far outside the support, both signs
    sym[esc] = (rng.integers(2000, 100000, esc.sum()) * rng.choice([-1, 1], esc.sum()))
    return sym.astype(np.int32), idx.astype(np.int32)


def test_reference_coder_round_trips_and_costs_what_the_tables_say():
    try:
        cdf, ln, off = _tables()
    except Exception:
        cdf, ln, off = _tables_fallback()
    rng = np.random.default_rng(0)
    n, S = 5000, 256
    sym, idx = _symbols(n, ln, off, rng)
    nst = -(-n // S)
    st, lens, pay = R.encode_level(sym, idx, cdf, ln, off, nst)
    assert len(st) == nst and int(lens.sum()) == len(pay) and st.min() >= 1 << 16
    out = R.decode_level(n, idx, cdf, ln, off, st, lens, pay)
    assert np.array_equal(out, sym)
    # rate: sum of -log2 p of the coded items (+ 32 bits per escape) within a word per stream of what the streams hold
    bits = 0.0
    for s, i in zip(sym, idx):
        v, maxv = int(s) - int(off[i]), int(ln[i]) - 2
        if not 0 <= v < maxv:
            raw = (-2 * v - 1) if v < 0 else 2 * (v - maxv)
            bits += 4 + 4 * max(1, -(-int(raw).bit_length() // 4)) if raw else 4
            v = maxv
        bits -= np.log2((cdf[i][v + 1] - cdf[i][v]) / 65536.0)
    coded = 16 * len(pay) + 32 * nst                     # payload + final states (the 16 bits a state holds above 2^16 are information too)
    assert bits - 1 <= coded <= bits + 32 * nst + 16, (bits, coded)
    blob = EM.ms_pack([(n, st, lens, pay)])
    assert len(blob) == 8 + 8 + 4 * nst + 2 * nst + (-(2 * nst) % 4) + 2 * len(pay) + (-(2 * len(pay)) % 4)
    (n2, st2, l2, p2), = EM.ms_unpack(blob)
    assert n2 == n and np.array_equal(st2, st) and np.array_equal(l2, lens) and np.array_equal(p2, pay)


@pytest.mark.parametrize("mangle", ["magic", "levels", "streams", "truncate"])
def test_malformed_multi_stream_strings_are_refused(mangle):
    st, ln, pay = np.array([65536, 65537], dtype=np.uint32), np.array([1, 2], dtype=np.uint16), np.array([7, 8, 9], dtype=np.uint16)
    blob = bytearray(EM.ms_pack([(600, st, ln, pay)]))
    if mangle == "magic":
        blob[0] ^= 0xFF
    elif mangle == "levels":
        blob[4:8] = (1000).to_bytes(4, "little")
    elif mangle == "streams":
        blob[12:16] = (0).to_bytes(4, "little")
    else:
        blob = blob[:-4]
    with pytest.raises(ValueError):
        EM.ms_unpack(bytes(blob))
