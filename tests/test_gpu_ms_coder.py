"""The GPU-decodable multi-stream range coder (csrc/rans_ms.hip; SURVEY section 8f-2, the optional half) on the GPU:
kernels against the plain-Python restatement of the format (tests/ms_ref.py), and the codecs in `coder = "rans_ms"` mode against the
reference-format mode -- same symbols, hence the same decoded tensors bit for bit; what differs is the string (and its measured overhead)."""
import numpy as np
import pytest
import torch

import ms_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gc():
    from deepsvc_amd.entropy_models import GaussianConditional
    from deepsvc_amd.modules import get_scale_table
    g = GaussianConditional(None)
    g.update_scale_table(get_scale_table(), force=True)
    return g


def _symbols(n, rng):
    idx = rng.integers(0, 64, n)
    sigma = 0.11 * (256 / 0.11) ** (idx / 63.0)
    sym = np.rint(rng.normal(0, 1, n) * sigma).astype(np.int64)
    esc = rng.random(n) < 0.01
    sym[esc] = rng.integers(2000, 100000, int(esc.sum())) * rng.choice([-1, 1], int(esc.sum()))
    return sym.astype(np.int32), idx.astype(np.int32)


@pytest.mark.parametrize("n,S", [(5000, 256), (257, 64), (65280, 256), (100, 512)])
def test_kernels_agree_with_the_python_restatement_of_the_format(n, S):
    from deepsvc_amd import ops
    g = _gc()
    ht, dt = g.host_tables(), g.device_tables(torch.device(DEV))
    rng = np.random.default_rng(n + S)
    sym, idx = _symbols(n, rng)
    nst = -(-n // S)
    sw = ops.rans_ms_slot_words(-(-n // nst))
    d_sym, d_idx = torch.from_numpy(sym).to(DEV), torch.from_numpy(idx).to(DEV)
    slots = torch.zeros(nst * sw, dtype=torch.int16, device=DEV)
    states = torch.zeros(nst, dtype=torch.int32, device=DEV)
    lens = torch.zeros(nst, dtype=torch.int16, device=DEV)
    ops.rans_ms_encode(d_sym, d_idx, n, nst, dt, slots, sw, states, lens)
    torch.cuda.synchronize()
    st, ln = states.cpu().numpy().view(np.uint32), lens.cpu().numpy().view(np.uint16)
    words = slots.cpu().numpy().view(np.uint16).reshape(nst, sw)
    pay = words[np.arange(sw)[None, :] >= (sw - ln.astype(np.int32))[:, None]]
    # ---- GPU encoder == the format's restatement, word for word (small cases: the Python coder is slow) ----
    if n <= 6000:
        st_r, ln_r, pay_r = R.encode_level(sym, idx, ht.cdf.astype(np.int64), ht.cdf_length, ht.offset, nst)
        assert np.array_equal(st, st_r) and np.array_equal(ln, ln_r) and np.array_equal(pay, pay_r)
        assert np.array_equal(R.decode_level(n, idx, ht.cdf.astype(np.int64), ht.cdf_length, ht.offset, st, ln, pay), sym)
    # ---- GPU decoder ----
    off = np.zeros(nst + 1, dtype=np.uint32)
    np.cumsum(ln.astype(np.uint32), out=off[1:])
    d_st, d_off = torch.from_numpy(st.view(np.int32)).to(DEV), torch.from_numpy(off.view(np.int32)).to(DEV)
    d_pay = torch.from_numpy(np.concatenate([pay, np.zeros(2, np.uint16)]).view(np.int16)).to(DEV)
    out = torch.full((n,), 12345, dtype=torch.int32, device=DEV)
    ops.rans_ms_decode(d_idx, n, nst, dt, d_st.data_ptr(), d_off.data_ptr(), d_pay.data_ptr(), len(pay), out)
    assert np.array_equal(out.cpu().numpy(), sym)
    # a corrupted payload must not fault (reads are bounded by the stream's own extent): result is garbage, the call returns
    d_pay.random_(-30000, 30000)
    ops.rans_ms_decode(d_idx, n, nst, dt, d_st.data_ptr(), d_off.data_ptr(), d_pay.data_ptr(), len(pay), out)
    torch.cuda.synchronize()
    print(f"[ms] n {n}, {nst} streams of <= {S}: {2 * len(pay) + 6 * nst} bytes = {8.0 * (2 * len(pay) + 6 * nst) / n:.3f} bits per symbol")


@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_pframe_codecs_in_multi_stream_mode(precision):
    """DeepSVC with both codecs' y strings in multi-stream mode: decoded tensors bit-identical to the reference-format mode over three
    chained P-frames (the symbols are the same), strings differ, and the overhead is what the format says (6 bytes per stream)."""
    from deepsvc_amd import bitstream, ops, synth
    ops.set_precision(precision)
    try:
        H, W = 256, 512
        fr = [f.to(DEV) for f in synth.make_frames(H, W, 4)]
        sms = [None] + [synth.make_sm_fea(H, W, t).to(DEV) for t in (1, 2, 3)]
        outs = {}
        for coder in ("rans", "rans_ms"):
            m = synth.make_model("W0"); m.update(force=True); m = m.to(DEV)
            m.set_coder(coder)
            ref, feat, rows = fr[0], None, []
            for t in (1, 2, 3):
                mv, res = m.compress(ref, fr[t], sms[t], feat)
                mv, res = dict(mv.items()), dict(res.items())
                blob = bitstream.pack_frame(mv, res, flags=bitstream.MODE_FLAG[precision] | (bitstream.FLAG_MULTISTREAM if coder == "rans_ms" else 0))
                _, (mv2, res2) = bitstream.unpack_frame(blob, expect_mode=precision)
                feat, ref, warped, predict = m.decompress(ref, mv2, res2, sms[t], feat)
                rows.append((mv, res, ref.clone(), feat.float().clone()))
            outs[coder] = rows
        for t, (a, b) in enumerate(zip(outs["rans"], outs["rans_ms"]), 1):
            assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), f"frame {t}: decoded tensors differ between the two coders"
            assert a[0]["strings"][1] == b[0]["strings"][1] and a[1]["strings"][1] == b[1]["strings"][1], "the z strings are the reference's in both modes"
            assert a[1]["strings"][0][0] != b[1]["strings"][0][0] and b[1]["strings"][0][0][:4] == b"MSR1"
            na = sum(len(e["strings"][0][0]) for e in a[:2]); nb = sum(len(e["strings"][0][0]) for e in b[:2])
            print(f"[ms] {precision} frame {t}: y strings {na} B (reference format) vs {nb} B (multi-stream): {100.0 * (nb - na) / na:+.2f} %")
            assert 0 < nb - na < 0.12 * na
    finally:
        ops.set_precision("fp32")


def test_intra_codec_in_multi_stream_mode_and_what_it_buys():
    """ICIP2020ResB (10 slices, 6 levels, 2.6 M symbols at 1080p: the host coder is most of its frame) at 768x1024: same decoded picture in both
    modes; encode + decode wall time of each (printed: the figure DESIGN quotes comes from tools/bench_ms.py at 1088x1920)."""
    import time
    from deepsvc_amd import ops, synth
    ops.set_precision("f16")
    try:
        x = synth.make_frames(768, 1024, 1)[0].to(DEV)
        res = {}
        for coder in ("rans", "rans_ms"):
            m = synth.make_icodec("W0"); m.update(force=True); m = m.to(DEV)
            m.coder = coder
            for _ in range(3):
                enc = m.compress(x); dec = m.decompress(enc["strings"], enc["shape"])["x_hat"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                enc = m.compress(x); dec = m.decompress(enc["strings"], enc["shape"])["x_hat"]
            torch.cuda.synchronize()
            res[coder] = ((time.perf_counter() - t0) / 3 * 1e3, len(enc["strings"][0][0]), dec.clone())
        assert torch.equal(res["rans"][2], res["rans_ms"][2])
        print(f"[ms] ICIP2020ResB 768x1024 f16: encode + decode {res['rans'][0]:.1f} ms (reference format, host coder) vs {res['rans_ms'][0]:.1f} ms (multi-stream), "
              f"y string {res['rans'][1]} vs {res['rans_ms'][1]} B ({100.0 * (res['rans_ms'][1] - res['rans'][1]) / res['rans'][1]:+.2f} %)")
    finally:
        ops.set_precision("fp32")
