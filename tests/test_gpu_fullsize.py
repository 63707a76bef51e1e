"""The HIP path against the REFERENCE at the sizes the performance numbers are quoted on.

Fixtures: tests/golden/deepsvc_ref_1080p.npz and deepsvc_ref_256x1024.npz, produced by running the reference's own
compress()/decompress() (tests/golden/make_golden_full.py, build container only).

  * 1088x1920 (BASELINE.json config 2): first + steady-state P-frame and one ICIP2020ResB key frame.
      - fp32 mode: the GPU encoder writes the reference's bytes; the reference's strings decode on the GPU to the reference's
        tensors (strided samples AND block means over every pixel, <= 1e-5);
      - f16 mode (the mode the headline rate is measured in): own encode -> decode chain against the REFERENCE's decoded
        tensors at tests/parity_util.F16_BOUNDS, bytes within 1 %, PSNR within 0.05 dB (north_star's tolerances).
  * 256x1024 = 1024 tiles of 8x32 pixels: the smallest picture that takes the large-image kernels in f16 mode (two-group
    ping-pong 3x3 incl. the fused ResBottleneck tail, weight-stationary 3x3, direct 1x1) -- asserted through the library's
    launch counters -- so those kernels run inside a frame whose output is compared with the reference's.
"""
import os

import numpy as np
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g1080():
    return np.load(os.path.join(GOLD, "deepsvc_ref_1080p.npz"))


@pytest.fixture(scope="module")
def gwide():
    return np.load(os.path.join(GOLD, "deepsvc_ref_256x1024.npz"))


def _model(precision):
    from deepsvc_amd import ops, synth
    ops.set_precision(precision)
    m = synth.make_model("W0")
    m.update(force=True)
    return m.to(DEV)


def _enc(g, pre, nm):
    return {"strings": [[g[f"{pre}/enc/{nm}_y"].tobytes()], [g[f"{pre}/enc/{nm}_z"].tobytes()]],
            "shape": tuple(int(v) for v in g[f"{pre}/enc/{nm}_shape"])}


def _strings(enc):
    return enc["strings"][0][0], enc["strings"][1][0]


def _nbytes(*encs):
    return sum(len(s[0]) for e in encs for s in e["strings"])


def _decode(model, ref, mv, res, sm, feat):
    model.debug_taps = {}
    try:
        out = model.decompress(ref, mv, res, sm, feat)
        taps = dict(model.debug_taps)
    finally:
        model.debug_taps = None
    return out, dict(feature=out[0], recon=taps["recon"], warped=taps["warped"], predict=taps["predict"])


def _plane_report(codec, g, pre, nm, h, w):
    """symbol / index planes the GPU encoder left in its staging vector vs the line checksums of the reference's planes."""
    _, ysym, yidx = PU.product_symbol_planes(codec, h, w, DEV)
    M = codec.M
    bad_s = int((PU.line_crcs(ysym.reshape(M, h, w)) != g[f"{pre}/enc/{nm}_sym_crc"]).sum())
    bad_i = int((PU.line_crcs(yidx.reshape(M, h, w)) != g[f"{pre}/enc/{nm}_idx_crc"]).sum())
    return bad_s, bad_i, M * h


def _fp32_chain_against_reference(g, tag, H, W, n_p, features_of):
    """fp32 mode, chained through the GPU's own decoded state: per frame (a) GPU encoder bytes == the reference's, symbol and
    index planes == the reference's line by line; (b) the reference's strings through the GPU decoder == the reference's tensors."""
    from deepsvc_amd import synth
    model = _model("fp32")
    fr = synth.make_frames(H, W, n_p + 1)
    ref, feat, psnrs = fr[0].to(DEV), None, []
    for t in range(1, n_p + 1):
        pre, sm, cur = f"{tag}/f{t}", synth.make_sm_fea(H, W, t).to(DEV), fr[t].to(DEV)
        mv, res = model.compress(ref, cur, sm, feat)
        for nm, enc, codec in (("mv", mv, model.mv_codec), ("res", res, model.res_codec)):
            want = _enc(g, pre, nm)
            bad_s, bad_i, lines = _plane_report(codec, g, pre, nm, H // 16, W // 16)
            same = _strings(enc) == _strings(want)
            print(f"[parity] {tag} f{t} {nm}: {len(_strings(enc)[0])}+{len(_strings(enc)[1])} B vs reference "
                  f"{len(_strings(want)[0])}+{len(_strings(want)[1])} B, identical={same}; lines differing: symbols {bad_s}/{lines}, indexes {bad_i}/{lines}")
            assert tuple(enc["shape"]) == want["shape"]
            assert bad_s == 0 and bad_i == 0, f"{pre} {nm}: symbol / index planes differ from the reference's in {bad_s} / {bad_i} of {lines} lines"
            assert same, f"{pre}: {nm} bitstream differs from the reference's"
        out, got = _decode(model, ref, _enc(g, pre, "mv"), _enc(g, pre, "res"), sm, feat)
        for name in ("recon", "warped", "predict"):
            PU.check_golden_full(g, f"{pre}/dec/{name}", out[{"recon": 1, "warped": 2, "predict": 3}[name]], 1e-5, f"{tag} f{t} {name} (clamped)")
        if t in features_of:
            PU.check_golden_full(g, f"{pre}/dec/feature", out[0], 1e-5, f"{tag} f{t} feature")
        psnrs.append(PU.psnr(out[1].cpu(), fr[t]))
        feat, ref = out[0], out[1]
    np.testing.assert_allclose(psnrs, g[f"{tag}/psnr"], atol=1e-3)
    return psnrs


def _f16_chain_against_reference(g, tag, H, W, n_p, features_of):
    """f16 mode on its own encode -> decode chain (its strings are not the reference's: mu / sigma differ in the last bits) against
    the reference's decoded tensors of the same frames."""
    from deepsvc_amd import ops, synth
    model = _model("f16")
    try:
        fr = synth.make_frames(H, W, n_p + 1)
        ref, feat = fr[0].to(DEV), None
        for t in range(1, n_p + 1):
            pre, sm, cur = f"{tag}/f{t}", synth.make_sm_fea(H, W, t).to(DEV), fr[t].to(DEV)
            mv, res = model.compress(ref, cur, sm, feat)
            out, _ = _decode(model, ref, mv, res, sm, feat)
            assert all(bool(torch.isfinite(o.float()).all()) for o in out)
            b_g, b_r = _nbytes(mv, res), _nbytes(_enc(g, pre, "mv"), _enc(g, pre, "res"))
            p_g, p_r = PU.psnr(out[1].cpu(), fr[t]), float(g[f"{tag}/psnr"][t - 1])
            print(f"[parity] {tag} f{t} f16 vs reference: bytes {b_g} vs {b_r} ({100.0 * (b_g - b_r) / b_r:+.3f} %), PSNR {p_g:.4f} vs {p_r:.4f} dB")
            for j, name in ((1, "recon"), (2, "warped"), (3, "predict")):
                PU.check_golden_full(g, f"{pre}/dec/{name}", out[j], PU.F16_BOUNDS[name], f"{tag} f{t} f16 {name}")
            if t in features_of:
                PU.check_golden_full(g, f"{pre}/dec/feature", out[0].float(), PU.F16_BOUNDS["feature"], f"{tag} f{t} f16 feature")
            assert abs(b_g - b_r) <= 0.01 * b_r, f"{pre}: f16 {b_g} B vs the reference's {b_r} B"
            assert abs(p_g - p_r) <= 0.05, f"{pre}: PSNR f16 {p_g:.4f} vs the reference's {p_r:.4f} dB"
            feat, ref = out[0], out[1]
    finally:
        ops.set_precision("fp32")


# ------------------------------------------------------------------------------------------------------------
# 256 x 1024: the large-image kernels inside a reference-checked frame
# ------------------------------------------------------------------------------------------------------------
def test_wide_256x1024_fp32_bytes_and_tensors_equal_reference(gwide):
    _fp32_chain_against_reference(gwide, "s256x1024", 256, 1024, 3, features_of=(3,))


def test_wide_256x1024_f16_fast_kernels_against_reference(gwide):
    from deepsvc_amd import ops
    keys = ("conv3x3_pp", "conv3x3_pp_fused_tail", "conv3x3_ws", "conv1x1")
    before = {k: ops.get_counter(k) for k in keys}
    _f16_chain_against_reference(gwide, "s256x1024", 256, 1024, 3, features_of=(3,))
    took = {k: ops.get_counter(k) - before[k] for k in keys}
    print(f"[parity] launches taken by the large-image kernels inside these frames: {took}")
    assert took["conv3x3_pp"] >= 3 * 60 and took["conv3x3_pp_fused_tail"] >= 3 * 6 and took["conv1x1"] >= 3 * 20, took


# ------------------------------------------------------------------------------------------------------------
# 1088 x 1920: the headline size
# ------------------------------------------------------------------------------------------------------------
def test_1080p_fp32_bytes_and_tensors_equal_reference(g1080):
    _fp32_chain_against_reference(g1080, "s1080", 1088, 1920, 2, features_of=(1, 2))


def test_1080p_f16_against_reference(g1080):
    _f16_chain_against_reference(g1080, "s1080", 1088, 1920, 2, features_of=(1, 2))


def test_1080p_intra_key_frame_against_reference(g1080):
    """ICIP2020ResB (image_model.py:490-591) at 1088x1920: fp32 bytes == the reference's, its strings decode to its picture;
    f16 mode within the f16 bounds of the reference's picture, bytes within 1 %."""
    from deepsvc_amd import ops, synth
    g, x = g1080, synth.make_frames(1088, 1920, 1)[0].to(DEV)
    want = {"strings": [[g["icip1080/enc/y"].tobytes()], [g["icip1080/enc/z"].tobytes()]], "shape": tuple(int(v) for v in g["icip1080/enc/shape"])}
    try:
        for mode in ("fp32", "f16"):
            ops.set_precision(mode)
            m = synth.make_icodec("W0")
            m.update(force=True)
            m = m.to(DEV)
            enc = m.compress(x)
            if mode == "fp32":
                _, ysym, yidx = PU.product_symbol_planes(m, 68, 120, DEV)
                bad_s = int((PU.line_crcs(ysym.reshape(m.M, 68, 120)) != g["icip1080/enc/sym_crc"]).sum())
                bad_i = int((PU.line_crcs(yidx.reshape(m.M, 68, 120)) != g["icip1080/enc/idx_crc"]).sum())
                print(f"[parity] icip 1080p fp32: {_nbytes(enc)} B vs reference {_nbytes(want)} B; lines differing: symbols {bad_s}, indexes {bad_i}")
                assert bad_s == 0 and bad_i == 0
                assert _strings(enc) == _strings(want), "I-frame bitstream differs from the reference's"
                dec = m.decompress(want["strings"], want["shape"])["x_hat"]
                PU.check_golden_full(g, "icip1080/dec/x_hat", dec, 1e-5, "icip 1080p fp32 x_hat")
            else:
                dec = m.decompress(enc["strings"], enc["shape"])["x_hat"]
                b_g, b_r = _nbytes(enc), _nbytes(want)
                print(f"[parity] icip 1080p f16: {b_g} B vs reference {b_r} B ({100.0 * (b_g - b_r) / b_r:+.3f} %)")
                PU.check_golden_full(g, "icip1080/dec/x_hat", dec, PU.F16_BOUNDS["recon"], "icip 1080p f16 x_hat")
                assert abs(b_g - b_r) <= 0.01 * b_r
    finally:
        ops.set_precision("fp32")
