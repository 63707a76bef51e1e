"""GPU parity of the I-frame codec `ICIP2020ResB`: the HIP path through the C ABI against the CPU
oracle (oracle/icodec.py, itself pinned byte for byte against the reference class) and against the golden bitstreams."""
import os

import numpy as np
import pytest
import torch

from gpu_util import psnr, report_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _make(precision):
    from deepsvc_amd import ops, synth
    ops.set_precision(precision)
    m = synth.make_icodec("W0")
    m.update(force=True)
    return m.to(DEV)


@pytest.fixture(scope="module")
def icip_oracle():
    from deepsvc_amd import synth
    from oracle.icodec import ICIPOracle
    o = ICIPOracle(synth.make_icodec("W0").state_dict())
    o.update(force=True)
    return o


@pytest.mark.parametrize("size", [64, 128])
def test_icip_fp32_matches_oracle_and_reference_bitstream(icip_oracle, size):
    """fp32 mode: stage tensors within 1e-3 of the tensor scale; the bitstreams are byte-identical to the oracle's and to the
    reference's golden bytes (asserted); the reference's golden strings decode on the GPU to the reference's picture."""
    from deepsvc_amd import ops, synth
    try:
        m = _make("fp32")
        x = synth.make_frames(size, size, 1)[0]
        taps = {}
        want = icip_oracle.forward(x, taps=taps)
        got = m(x.to(DEV))
        tol = 2e-3 * float(want["x_hat"].abs().max())
        report_close("forward x_hat", got["x_hat"].cpu(), want["x_hat"], 0.0, tol)
        for k in ("y", "z"):
            a, b = float(torch.log(got["likelihoods"][k]).sum()), float(torch.log(want["likelihoods"][k]).sum())
            assert abs(a - b) <= 2e-3 * abs(b) + 1e-2, (k, a, b)
        enc = m.compress(x.to(DEV))
        enc_o = icip_oracle.compress(x)
        g = np.load(os.path.join(GOLD, f"icip_ref_{size}.npz"))
        ny, no = len(enc["strings"][0][0]), len(enc_o["strings"][0][0])
        same = enc["strings"][0][0] == enc_o["strings"][0][0] and enc["strings"][1][0] == enc_o["strings"][1][0]
        same_ref = enc["strings"][0][0] == g[f"s{size}/enc/y"].tobytes() and enc["strings"][1][0] == g[f"s{size}/enc/z"].tobytes()
        print(f"icip {size}: y bytes gpu {ny} oracle {no}; identical to oracle {same}, to the reference golden {same_ref}")
        assert tuple(enc["shape"]) == tuple(enc_o["shape"]) == (size // 64, size // 64)
        assert abs(ny - no) <= max(0.01 * no, 8)
        dec = m.decompress(enc["strings"], enc["shape"])["x_hat"].cpu()
        dec_o = icip_oracle.decompress(enc_o["strings"], enc_o["shape"])["x_hat"]
        assert float(dec.min()) >= 0.0 and float(dec.max()) <= 1.0
        import parity_util as PU
        assert same and same_ref, "fp32 mode: the bitstream differs from the oracle's / the reference's golden bytes"
        e = PU.rel_l2(dec, dec_o)
        print(f"icip {size}: decoded picture rel-L2 vs oracle {e:.2e}, mean {float(dec.mean()):.3f}, clamped {float(((dec <= 0) | (dec >= 1)).float().mean()):.4f}")
        assert e <= 1e-4
        # the reference's golden strings through the GPU decoder -> the reference's decoded picture
        gold_strings = [[g[f"s{size}/enc/y"].tobytes()], [g[f"s{size}/enc/z"].tobytes()]]
        dec2 = m.decompress(gold_strings, tuple(int(v) for v in g[f"s{size}/enc/shape"]))["x_hat"]
        PU.check_golden(g, f"s{size}/dec/x_hat", dec2, rel=1e-4)
    finally:
        ops.set_precision("fp32")


def test_icip_roundtrip_properties_large_and_f16():
    """768x1024 (latent 48x64, 10 slices x 32 channels): encode -> decode round trip reproduces the encoder's latents bit
    for bit, decoding is deterministic, f16 mode stays within 1 % bytes and 0.05 dB (PSNR against the source) of fp32 mode."""
    from deepsvc_amd import ops, synth
    H, W = 768, 1024
    x = synth.make_frames(H, W, 1, seed=11)[0].to(DEV)
    res = {}
    try:
        for prec in ("fp32", "f16"):
            m = _make(prec)
            for rep in range(2):                  # second pass runs on captured HIP graphs
                enc = m.compress(x)
                torch.cuda.synchronize()
                plan = m._plan(1, H // 16, W // 16, torch.device(DEV))
                y_enc = plan.y_hat.buf.clone()
                d1 = m.decompress(enc["strings"], enc["shape"])["x_hat"]
                torch.cuda.synchronize()
                y_dec = plan.y_hat.buf.clone()
                d2 = m.decompress(enc["strings"], enc["shape"])["x_hat"]
                assert torch.equal(y_enc, y_dec), f"{prec}: decoder latents differ from the encoder's"
                assert torch.equal(d1, d2)
            n = len(enc["strings"][0][0]) + len(enc["strings"][1][0])
            res[prec] = (n, d1.float().cpu())
            assert tuple(enc["shape"]) == (H // 64, W // 64) and n > 1000
        (nf, xf), (nb, xb) = res["fp32"], res["f16"]
        xc = x.float().cpu()
        pf, pb = psnr(xf, xc), psnr(xb, xc)
        print(f"icip {H}x{W}: fp32 {nf} B, f16 {nb} B ({100.0 * (nb - nf) / nf:+.3f} %), PSNR vs source fp32 {pf:.4f} / f16 {pb:.4f} dB, "
              f"PSNR(f16, fp32) {psnr(xb, xf):.1f} dB")
        assert abs(nb - nf) <= 0.01 * nf                 # north-star tolerances: 1 % bytes, 0.05 dB
        assert abs(pb - pf) <= 0.05
        assert psnr(xb, xf) > 30                         # 13 untrained 192-channel synthesis layers amplify f16 rounding
    finally:
        ops.set_precision("fp32")


def test_gop_with_intra_coded_key_frame():
    """BASELINE.json config 3 plumbing on the GPU: I-frame codec for the key frame, its reconstruction as the reference
    of the P-frame chain (harness.code_gop(..., i_model=...), test_video.py:296-312)."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("f16")
    try:
        i_model = synth.make_icodec("W0"); i_model.update(force=True); i_model = i_model.to(DEV)
        p_model = synth.make_model("W0"); p_model.update(force=True); p_model = p_model.to(DEV)
        size = 192
        frames = [f.to(DEV) for f in synth.make_frames(size, size, 4)]
        sms = [None] + [synth.make_sm_fea(size, size, t).to(DEV) for t in (1, 2, 3)]
        rows = harness.code_gop(p_model, frames, sms, unit=3, sync=torch.cuda.synchronize, i_model=i_model, msssim=True)
        assert all(0.0 < r[9] <= 1.0 for r in rows)
        assert [r[1] for r in rows] == [0.0, 1.0, 2.0, 3.0]
        assert rows[0][3] > 0.0 and all(r[2] > 0.0 and r[3] > 0.0 for r in rows[1:])
        assert all(np.isfinite(r[4]) for r in rows)
        table = harness.gather_rows(rows, torch.device(DEV), 1)
        assert table.shape == (4, len(harness.METRIC_COLUMNS))
        # the reference's per-run JSON (test_video.py:442-460) from a real coded GOP: all 26 keys, finite where defined
        res = harness.reference_results(table)
        assert tuple(res) == harness.REFERENCE_JSON_KEYS
        for k, v in res.items():
            assert len(v) == 1 and np.isfinite(v[0]), (k, v)
        assert 0 < res["warmsssim"][0] <= 1 and 0 < res["mcmsssim"][0] <= 1 and res["ibpp"][0] > 0 and res["smbpp"][0] == 0.0
    finally:
        ops.set_precision("fp32")


def test_config3_gop12_four_rate_points():
    """BASELINE.json config 3: GOP = 12 with an intra-coded key frame at four operating points (synthetic stand-ins for
    lambda in {256, 512, 1024, 2048}: synth.RATE_GAINS).  Properties the domain offers without the reference: every frame
    decodes, the rate grows with lambda for the I-frame and for the P-frames, and so does the key frame's PSNR."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("f16")
    try:
        size, gop = 192, 12
        frames = [f.to(DEV) for f in synth.make_frames(size, size, gop)]
        sms = [None] + [synth.make_sm_fea(size, size, t).to(DEV) for t in range(1, gop)]
        points = []
        for lam in (256, 512, 1024, 2048):
            i_model = synth.make_icodec("W0", lam=lam); i_model.update(force=True); i_model = i_model.to(DEV)
            p_model = synth.make_model("W0", lam=lam); p_model.update(force=True); p_model = p_model.to(DEV)
            rows = harness.code_gop(p_model, frames, sms, unit=0, sync=torch.cuda.synchronize, i_model=i_model)
            assert len(rows) == gop and all(np.isfinite(v) for r in rows for v in r)
            i_bpp, i_psnr = rows[0][3], rows[0][4]
            p_bpp = float(np.mean([r[2] + r[3] for r in rows[1:]]))
            points.append((lam, i_bpp, i_psnr, p_bpp, float(np.mean([r[4] for r in rows[1:]]))))
            print(f"lambda {lam}: I {i_bpp:.3f} bpp {i_psnr:.2f} dB | P mean {p_bpp:.3f} bpp {points[-1][4]:.2f} dB")
        for lo, hi in zip(points[:-1], points[1:]):
            assert hi[1] > lo[1] and hi[3] > lo[3], f"rate not increasing with lambda: {lo} -> {hi}"
            assert hi[2] > lo[2] - 0.05, f"key-frame PSNR not increasing with lambda: {lo} -> {hi}"
    finally:
        ops.set_precision("fp32")


def test_config3_1080p_gop12_four_rate_points():
    """BASELINE.json config 3 at its full size: 1920x1080 (padded 1088x1920), GOP = 12, key frame coded by `ICIP2020ResB`, four
    operating points (synth.RATE_GAINS stand-ins for lambda in {256, 512, 1024, 2048}), f16 mode.  The CPU oracle is too slow
    at this size, so size-independent properties: every frame decodes to a finite picture in [0,1]; coding the GOP twice gives
    bit-identical rows (deterministic kernels, encoder/decoder lock-step through a real bitstream), including the decoded latents
    of the last P-frame; rate is monotone in lambda for the I-frame and for the P-frames."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("f16")
    try:
        H, W, gop = 1080, 1920, 12
        frames = [f.to(DEV) for f in synth.make_frames(H, W, gop, seed=21)]
        sms = [None] + [synth.make_sm_fea(1088, 1920, t, seed=21).to(DEV) for t in range(1, gop)]
        points = []
        for lam in (256, 512, 1024, 2048):
            i_model = synth.make_icodec("W0", lam=lam); i_model.update(force=True); i_model = i_model.to(DEV)
            p_model = synth.make_model("W0", lam=lam); p_model.update(force=True); p_model = p_model.to(DEV)
            rows = harness.code_gop(p_model, frames, sms, unit=0, sync=torch.cuda.synchronize, i_model=i_model)
            dev = torch.device(DEV)
            dec_lat = [c._plan(1, 68, 120, dev).y_hat.buf.clone() for c in (p_model.mv_codec, p_model.res_codec)]
            assert len(rows) == gop and all(np.isfinite(v) for r in rows for v in r)
            if lam == 2048:
                again = harness.code_gop(p_model, frames, sms, unit=0, sync=torch.cuda.synchronize, i_model=i_model)
                assert [r[:7] for r in again] == [r[:7] for r in rows], "coding the same GOP twice gave different rows"
                lat2 = [c._plan(1, 68, 120, dev).y_hat.buf for c in (p_model.mv_codec, p_model.res_codec)]
                assert all(torch.equal(a, b) for a, b in zip(dec_lat, lat2))
            i_bpp, i_psnr = rows[0][3], rows[0][4]
            p_bpp = float(np.mean([r[2] + r[3] for r in rows[1:]]))
            t_frame = float(np.mean([r[7] + r[8] for r in rows[1:]]))
            points.append((lam, i_bpp, i_psnr, p_bpp, float(np.mean([r[4] for r in rows[1:]]))))
            print(f"config 3 @1080p lambda {lam}: I {i_bpp:.3f} bpp {i_psnr:.2f} dB ({1e3 * (rows[0][7] + rows[0][8]):.1f} ms) | "
                  f"P mean {p_bpp:.3f} bpp {points[-1][4]:.2f} dB ({1e3 * t_frame:.1f} ms/frame, one lane, joined coder timing)")
            del i_model, p_model
            torch.cuda.empty_cache()
        for lo, hi in zip(points[:-1], points[1:]):
            assert hi[1] > lo[1] and hi[3] > lo[3], f"rate not increasing with lambda: {lo} -> {hi}"
            assert hi[2] > lo[2] - 0.05, f"key-frame PSNR not increasing with lambda: {lo} -> {hi}"
    finally:
        ops.set_precision("fp32")


def test_low_rate_point_parity_with_oracle():
    """The lambda = 256 stand-in through the real bitstream path in fp32 mode against the oracle on the same weights
    (P-frame codec, 64 x 64): identical bitstreams, un-clamped decoded tensors within parity_util.FP32_BOUNDS."""
    from deepsvc_amd import ops, synth
    from oracle.codec import DeepSVCOracle, strings_nbytes
    ops.set_precision("fp32")
    m = synth.make_model("W0", lam=256)
    m.update(force=True)
    orc = DeepSVCOracle(m.state_dict())
    orc.update(force=True)
    m = m.to(DEV)
    fr = synth.make_frames(64, 64, 2)
    sm = synth.make_sm_fea(64, 64, 1)
    import parity_util as PU
    mv_g, res_g = m.compress(fr[0].to(DEV), fr[1].to(DEV), sm.to(DEV), None)
    m.debug_taps = {}
    dec_g = m.decompress(fr[0].to(DEV), mv_g, res_g, sm.to(DEV), None)
    got = dict(feature=dec_g[0].float().cpu(), **{k: m.debug_taps[k].cpu() for k in ("recon", "warped", "predict")})
    m.debug_taps = None
    mv_o, res_o = orc.compress(fr[0], fr[1], sm, None)
    taps = {}
    orc.decompress(fr[0], mv_o, res_o, sm, None, taps)
    for eg, eo in ((mv_g, mv_o), (res_g, res_o)):
        assert eg["strings"][0][0] == eo["strings"][0][0] and eg["strings"][1][0] == eo["strings"][1][0], "bitstream differs from the oracle's"
    PU.check_frame("lambda 256, 64x64, fp32 vs oracle", got, taps)
