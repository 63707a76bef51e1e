"""End-to-end parity of whole P-frames on the GPU, written so that it can fail.

  * fp32 mode vs the CPU oracle: byte-identical bitstreams (asserted), identical symbol / index planes, decoded latents,
    decoded feature and the UN-CLAMPED recon / warped / predict at relative-L2 <= 5e-5, over chained P-frames;
  * the REFERENCE's own golden byte strings (tests/golden, produced by running the reference) decoded by the GPU decoder,
    checked against the reference's decoded tensors -- 64x64 (first + steady-state P-frame) and BASELINE.json config 1:
    one 256x256 septuplet, six chained P-frames, on both the decode and the encode side;
  * f16 mode: relative-L2 of feature / un-clamped recon against the fp32 oracle, rate and PSNR deltas;
  * mutation test: a zero decoder, a skipped UNet and one zeroed filter tap each turn the checks red.
"""
import numpy as np
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fresh_model(precision="fp32", weight_set="W0"):
    from deepsvc_amd import ops, synth
    ops.set_precision(precision)
    m = synth.make_model(weight_set)
    m.update(force=True)
    return m.to(DEV)


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available()
    return _fresh_model("fp32")


def _gpu_decode(model, ref, mv, res, sm, feat):
    """product decompress with the un-clamped tensors tapped -> (outputs, {feature, recon, warped, predict})."""
    model.debug_taps = {}
    try:
        out = model.decompress(ref, mv, res, sm, feat)
        taps = dict(model.debug_taps)
    finally:
        model.debug_taps = None
    got = dict(feature=out[0].float().cpu(), recon=taps["recon"].cpu(), warped=taps["warped"].cpu(), predict=taps["predict"].cpu())
    return out, got


def _oracle_decode(oracle, ref, mv, res, sm, feat):
    taps = {}
    out = oracle.decompress(ref, mv, res, sm, feat, taps)
    return out, dict(feature=taps["feature"], recon=taps["recon"], warped=taps["warped"], predict=taps["predict"]), taps


def _strings(enc):
    return enc["strings"][0][0], enc["strings"][1][0]


def _golden_enc(g, prefix, nm):
    return {"strings": [[g[f"{prefix}/enc/{nm}_y"].tobytes()], [g[f"{prefix}/enc/{nm}_z"].tobytes()]],
            "shape": tuple(int(v) for v in g[f"{prefix}/enc/{nm}_shape"])}


# ------------------------------------------------------------------------------------------------------------
# fp32 mode vs the oracle: chained P-frames
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [64, 128])
def test_fp32_chain_vs_oracle(model, oracle_w0, size):
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    fr = synth.make_frames(size, size, 4)
    h = size // 16
    ref_g, ref_o, feat_g, feat_o = fr[0].to(DEV), fr[0], None, None
    for t in (1, 2, 3):                     # first P-frame (feature=None) and two steady-state ones
        sm, cur = synth.make_sm_fea(size, size, t), fr[t]
        mv_g, res_g = model.compress(ref_g, cur.to(DEV), sm.to(DEV), feat_g)
        enc_taps = {}
        mv_o, res_o = oracle_w0.compress(ref_o, cur, sm, feat_o, enc_taps)
        # --- encoder side: symbol / index planes and bytes ---
        for nm, eg, eo, codec in (("mv", mv_g, mv_o, model.mv_codec), ("res", res_g, res_o, model.res_codec)):
            _, ysym, yidx = PU.product_symbol_planes(codec, h, h, DEV)
            ms, mi = PU.symbol_mismatch(ysym, enc_taps[nm]["symbols"]), PU.symbol_mismatch(yidx, enc_taps[nm]["indexes"])
            print(f"[parity] {size} f{t} {nm}: symbol mismatch {ms:.5f}, index mismatch {mi:.5f}, "
                  f"bytes gpu {len(_strings(eg)[0])}+{len(_strings(eg)[1])} oracle {len(_strings(eo)[0])}+{len(_strings(eo)[1])}")
            assert ms == 0.0 and mi == 0.0, f"{nm}: symbol planes differ from the oracle's ({ms:.5f} / {mi:.5f})"
            assert _strings(eg) == _strings(eo), f"{nm}: bitstream differs from the oracle's"
            assert tuple(eg["shape"]) == tuple(eo["shape"]) == (size // 64, size // 64)
        # --- decoder side: latents, feature, un-clamped frames ---
        dec_g, got = _gpu_decode(model, ref_g, mv_g, res_g, sm.to(DEV), feat_g)
        dec_o, want, taps_o = _oracle_decode(oracle_w0, ref_o, mv_o, res_o, sm, feat_o)
        for nm, codec in (("mv", model.mv_codec), ("res", model.res_codec)):
            e = PU.rel_l2(PU.product_latent(codec, h, h, DEV), taps_o[f"{nm}_y_hat"])
            print(f"[parity] {size} f{t} {nm} y_hat rel-L2 {e:.2e}")
            assert e <= PU.FP32_BOUNDS["latent"]
        PU.check_frame(f"{size} f{t} fp32 vs oracle", got, want)
        assert abs(PU.psnr(dec_g[1].cpu(), cur) - PU.psnr(dec_o[1], cur)) <= 1e-3
        assert float(dec_g[1].min()) >= 0 and float(dec_g[1].max()) <= 1
        # decoding twice is bit-stable; the estimate path agrees with the real path up to ste_round's float path
        dec_g2 = model.decompress(ref_g, mv_g, res_g, sm.to(DEV), feat_g)
        assert torch.equal(dec_g2[1], dec_g[1]) and torch.equal(dec_g2[0].float(), dec_g[0].float())
        fwd = model(ref_g, cur.to(DEV), sm.to(DEV), feat_g)
        assert PU.rel_l2(fwd[0].cpu(), got["recon"]) < 1e-4
        ref_g, feat_g, ref_o, feat_o = dec_g[1], dec_g[0], dec_o[1], dec_o[0]


# ------------------------------------------------------------------------------------------------------------
# the reference's golden byte strings through the GPU decoder
# ------------------------------------------------------------------------------------------------------------
def test_reference_golden_strings_decode_on_gpu_64(model, golden64):
    """golden64 holds what the reference wrote for frame 0->1 (`s64_first`) and for the chained frame 1->2 (`s64_steady`).
    The GPU decoder takes the reference's strings, chained through its own decoded state, and must reproduce the reference's
    decoded feature / recon / warped / predict."""
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    g, fr = golden64, synth.make_frames(64, 64, 3)
    ref, feat = fr[0].to(DEV), None
    for t, tag in ((1, "s64_first"), (2, "s64_steady")):
        sm = synth.make_sm_fea(64, 64, t).to(DEV)
        out = model.decompress(ref, _golden_enc(g, tag, "mv"), _golden_enc(g, tag, "res"), sm, feat)
        for j, name in enumerate(["feature", "recon", "warped", "predict"]):
            e = PU.check_golden(g, f"{tag}/dec/{name}", out[j])
            print(f"[parity] golden {tag} dec/{name}: rel-L2 {e:.2e}")
        # and the GPU encoder reproduces the reference's bytes on the same inputs
        mv, res = model.compress(ref, fr[t].to(DEV), sm, feat)
        for nm, enc in (("mv", mv), ("res", res)):
            assert _strings(enc) == _strings(_golden_enc(g, tag, nm)), f"{tag}: {nm} bitstream differs from the reference's"
        feat, ref = out[0], out[1]


def test_config1_septuplet_256_against_reference_golden(model, golden256):
    """BASELINE.json config 1: one 256x256 septuplet, frame 0 the given key frame, six chained P-frames, W0, fp32 mode.
    (a) decode side: the reference's strings of every frame through the GPU decoder (state chained through the GPU's own
        outputs) reproduce the reference's decoded frames / features;
    (b) encode side: the GPU encoder, on the GPU-decoded state, writes the reference's bytes for all six frames;
    (c) PSNR per frame equals the reference's to 1e-3 dB."""
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    g, tag = golden256, "s256_gop"
    fr = synth.make_frames(256, 256, 7)
    ref, feat, psnrs = fr[0].to(DEV), None, []
    for t in range(1, 7):
        sm = synth.make_sm_fea(256, 256, t).to(DEV)
        pre = f"{tag}/f{t}"
        mv, res = model.compress(ref, fr[t].to(DEV), sm, feat)
        for nm, enc in (("mv", mv), ("res", res)):
            want = _golden_enc(g, pre, nm)
            assert tuple(enc["shape"]) == want["shape"] == (4, 4)
            assert _strings(enc) == _strings(want), f"frame {t}: {nm} bitstream differs from the reference's"
        out = model.decompress(ref, _golden_enc(g, pre, "mv"), _golden_enc(g, pre, "res"), sm, feat)
        errs = {name: PU.check_golden(g, f"{pre}/dec/{name}", out[j]) for j, name in ((1, "recon"), (2, "warped"), (3, "predict"))}
        if t in (1, 6):
            errs["feature"] = PU.check_golden(g, f"{pre}/dec/feature", out[0], stride=8)
        print(f"[parity] config1 f{t}: " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
        psnrs.append(PU.psnr(out[1].cpu(), fr[t]))
        feat, ref = out[0], out[1]
    np.testing.assert_allclose(psnrs, g[f"{tag}/psnr"], atol=1e-3)


def test_w1_weight_set_parity():
    """BASELINE.json config 4's "MS-SSIM model" is the same code path with another weight set: W1 (seed 1) in fp32 mode
    against the oracle on the same weights, same checks as W0."""
    from deepsvc_amd import synth
    from oracle.codec import DeepSVCOracle
    m = _fresh_model("fp32", "W1")
    orc = DeepSVCOracle(synth.make_model("W1").state_dict())
    orc.update(force=True)
    fr, sm = synth.make_frames(64, 64, 2), synth.make_sm_fea(64, 64, 1)
    mv_g, res_g = m.compress(fr[0].to(DEV), fr[1].to(DEV), sm.to(DEV), None)
    mv_o, res_o = orc.compress(fr[0], fr[1], sm, None)
    assert _strings(mv_g) == _strings(mv_o) and _strings(res_g) == _strings(res_o)
    _, got = _gpu_decode(m, fr[0].to(DEV), mv_g, res_g, sm.to(DEV), None)
    _, want, _ = _oracle_decode(orc, fr[0], mv_o, res_o, sm, None)
    PU.check_frame("W1 64 fp32 vs oracle", got, want)


# ------------------------------------------------------------------------------------------------------------
# the checks can fail
# ------------------------------------------------------------------------------------------------------------
def test_parity_checks_catch_mutations(oracle_w0):
    """A decoder that returns zeros, a decoder with one UNet bypassed and a decoder with ONE zeroed 3x3 filter tap in the
    residual refinement net must each fail `check_frame`; the unmutated model passes it on the same strings."""
    from deepsvc_amd import synth
    m = _fresh_model("fp32")
    fr, sm = synth.make_frames(64, 64, 2), synth.make_sm_fea(64, 64, 1)
    mv_o, res_o = oracle_w0.compress(fr[0], fr[1], sm, None)
    _, want, _ = _oracle_decode(oracle_w0, fr[0], mv_o, res_o, sm, None)
    args = (fr[0].to(DEV), mv_o, res_o, sm.to(DEV), None)
    _, got = _gpu_decode(m, *args)
    PU.check_frame("unmutated", got, want)

    zeros = {k: torch.zeros_like(v) for k, v in got.items()}
    with pytest.raises(AssertionError):
        PU.check_frame("mutant: zero decoder", zeros, want)
    black = dict(got, recon=torch.zeros_like(got["recon"]))
    with pytest.raises(AssertionError):
        PU.check_frame("mutant: black frame, everything else right", black, want)

    def mutated(mutate):
        mm = _fresh_model("fp32")
        with torch.no_grad():
            mutate(mm)
        return _gpu_decode(mm, *args)[1]

    def skip_unet(mm):                      # unet_2 contributes nothing: its last block's convs and shortcut are zeroed
        for p in mm.enhance.unet_2.up_conv2.parameters():
            p.zero_()

    def one_tap(mm):                        # a single (out 5, in 7, centre) tap of RefineResiNet's first conv
        mm.RefineResiNet.refine[0].weight[5, 7, 1, 1] = 0.0

    for name, fn in (("unet_2 bypassed", skip_unet), ("one zeroed filter tap", one_tap)):
        with pytest.raises(AssertionError):
            PU.check_frame(f"mutant: {name}", mutated(fn), want)


# ------------------------------------------------------------------------------------------------------------
# f16 mode
# ------------------------------------------------------------------------------------------------------------
def test_f16_mode_vs_fp32_oracle(oracle_w0):
    """Fast mode (f16 MFMA operands, f16 activation storage, fp32 accumulation): same API; encoder/decoder stay in lock
    step; un-clamped decoded tensors within the stated relative-L2 of the fp32 oracle over three chained P-frames at 128x128;
    rate within 1 %, PSNR within 0.05 dB (north_star tolerances)."""
    from deepsvc_amd import ops, synth
    from oracle.codec import strings_nbytes
    size = 128
    fr = synth.make_frames(size, size, 4)
    m = _fresh_model("f16")
    try:
        ref_g, ref_o, feat_g, feat_o = fr[0].to(DEV), fr[0], None, None
        for t in (1, 2, 3):
            sm = synth.make_sm_fea(size, size, t)
            mv_g, res_g = m.compress(ref_g, fr[t].to(DEV), sm.to(DEV), feat_g)
            dec_g, got = _gpu_decode(m, ref_g, mv_g, res_g, sm.to(DEV), feat_g)
            dec_g2 = m.decompress(ref_g, mv_g, res_g, sm.to(DEV), feat_g)
            assert torch.equal(dec_g[1], dec_g2[1])
            mv_o, res_o = oracle_w0.compress(ref_o, fr[t], sm, feat_o)
            dec_o, want, _ = _oracle_decode(oracle_w0, ref_o, mv_o, res_o, sm, feat_o)
            PU.check_frame(f"128 f{t} f16 vs fp32 oracle", got, want, PU.F16_BOUNDS)
            b_g, b_o = strings_nbytes(mv_g) + strings_nbytes(res_g), strings_nbytes(mv_o) + strings_nbytes(res_o)
            p_g, p_o = PU.psnr(dec_g[1].cpu(), fr[t]), PU.psnr(dec_o[1], fr[t])
            print(f"[parity] f16 f{t}: bytes {b_g} vs {b_o} ({100 * (b_g - b_o) / b_o:+.2f} %), PSNR {p_g:.4f} vs {p_o:.4f} dB, "
                  f"PSNR(f16 recon, fp32 recon) {PU.psnr(got['recon'], want['recon']):.1f} dB")
            assert abs(b_g - b_o) <= 0.01 * b_o
            assert abs(p_g - p_o) <= 0.05
            # chain each side on its own decoded state, as a real encoder/decoder pair would
            ref_g, feat_g, ref_o, feat_o = dec_g[1], dec_g[0], dec_o[1], dec_o[0]
    finally:
        ops.set_precision("fp32")


def test_batch_of_two(model, oracle_w0):
    """B = 2 through the same API (the reference's compress writes ONE y-string for the whole batch and one z-string per item,
    image_model.py:253-255 / SURVEY section 9.12; its decompress reshapes to batch 1, :289, so batch decoding is a superset here):
    the estimate path and a codec's batch bitstream against the oracle, and the batched P-frame decode bit-identical to decoding
    each item on its own."""
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    size = 64
    a, b = synth.make_frames(size, size, 2, seed=1234), synth.make_frames(size, size, 2, seed=99)
    ref, cur = torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]])
    sm = torch.cat([synth.make_sm_fea(size, size, 1, seed=1234), synth.make_sm_fea(size, size, 1, seed=99)])
    fw, fo = model(ref.to(DEV), cur.to(DEV), sm.to(DEV), None), oracle_w0.forward(ref, cur, sm, None)
    assert PU.rel_l2(fw[0].cpu(), fo[0]) <= 1e-5 and PU.rel_l2(fw[1].float().cpu(), fo[1]) <= 1e-5
    assert abs(float(fw[7]) - float(fo[7])) <= 1e-3 * float(fo[7])
    # codec level: the reference's compress handles a batch (one y-string, one z-string per item) -- bytes vs the oracle.  (Its
    # DeepSVC.compress decodes its own mv stream with the batch-1 decompress, so the reference itself stops there for B > 1.)
    from oracle import codec as OC
    torch.manual_seed(5)
    x8 = torch.randn(2, 8, size, size) * 0.5
    enc_g = model.mv_codec.compress(x8.to(DEV))
    enc_o = OC.codec_compress(oracle_w0.root.sub("mv_codec"), oracle_w0.tables["mv"], x8)
    assert len(enc_g["strings"][0]) == 1 and len(enc_g["strings"][1]) == 2
    assert enc_g["strings"][0][0] == enc_o["strings"][0][0] and list(enc_g["strings"][1]) == list(enc_o["strings"][1]), "batch bitstream differs from the oracle's"
    mv, res = model.compress(ref.to(DEV), cur.to(DEV), sm.to(DEV), None)
    dec = model.decompress(ref.to(DEV), mv, res, sm.to(DEV), None)
    assert tuple(dec[1].shape) == (2, 3, size, size) and tuple(dec[0].shape) == (2, 64, size, size)
    for i, fr in enumerate((a, b)):
        mv1, res1 = model.compress(fr[0].to(DEV), fr[1].to(DEV), sm[i:i + 1].to(DEV), None)
        d1 = model.decompress(fr[0].to(DEV), mv1, res1, sm[i:i + 1].to(DEV), None)
        assert torch.equal(dec[1][i:i + 1], d1[1]) and torch.equal(dec[0][i:i + 1].float(), d1[0].float()), f"item {i}: batched decode differs"
