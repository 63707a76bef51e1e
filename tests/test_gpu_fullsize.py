"""The HIP path against the REFERENCE at the sizes the performance numbers are quoted on.

Fixtures: tests/golden/deepsvc_ref_1080p.npz and deepsvc_ref_256x1024.npz, produced by running the reference's own
compress()/decompress() (tests/golden/make_golden_full.py, build container only).

  * 1088x1920 (BASELINE.json config 2): first + steady-state P-frame and one ICIP2020ResB key frame.
      - fp32 mode: a 1080p P-frame takes 1.3 M hard decisions (round(y - mu), sigma -> one of 64 table levels), ~700 of which the
        reference itself takes with a relative margin below 2e-5; an fp32 implementation with another summation order (this one; the
        reference on cuDNN) flips a few of them.  So at this size the test does not ask for identical bytes (256x1024 below and
        64..256 px in test_gpu_parity.py do, and get them) but PROVES where the differences come from: every differing symbol / index
        is either a decision the reference took within 2e-5 of flipping, or lies downstream of one (explain_plane_mismatches);
        bytes within 0.01 %; the GPU decoder follows the reference's string symbol for symbol up to the first such fragile CDF
        lookup (and all the way when there is none); decoded tensors against the reference's (strided samples AND block means over
        every pixel);
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


@pytest.fixture(scope="module")
def ggop1080():
    return np.load(os.path.join(GOLD, "deepsvc_ref_gop1080p.npz"))


@pytest.fixture(scope="module")
def ggopwide():
    return np.load(os.path.join(GOLD, "deepsvc_ref_gop256x1024.npz"))


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
    keys = ("conv3x3_pp", "conv3x3_pp_fused_tail", "conv3x3_ws", "conv1x1", "conv_head")
    before = {k: ops.get_counter(k) for k in keys}
    _f16_chain_against_reference(gwide, "s256x1024", 256, 1024, 3, features_of=(3,))
    took = {k: ops.get_counter(k) - before[k] for k in keys}
    print(f"[parity] launches taken by the large-image kernels inside these frames: {took}")
    assert took["conv3x3_pp"] >= 3 * 60 and took["conv3x3_pp_fused_tail"] >= 3 * 6 and took["conv1x1"] >= 3 * 20, took
    assert took["conv_head"] >= 3 * 2, took          # the dot-product heads (SpyNet's 16 -> 2, the 32 -> 3 / 64 -> 2 outputs) ran inside these frames


# ------------------------------------------------------------------------------------------------------------
# 1088 x 1920: the headline size
# ------------------------------------------------------------------------------------------------------------
def test_1080p_fp32_against_reference(g1080):
    from deepsvc_amd import _native as N, synth
    g, tag, H, W, h, w = g1080, "s1080", 1088, 1920, 68, 120
    model = _model("fp32")
    fr = synth.make_frames(H, W, 3)
    dev = torch.device(DEV)
    # ---------------- frame 1 (first P-frame: inputs identical to the reference's) ----------------
    pre, sm, ref, cur = f"{tag}/f1", synth.make_sm_fea(H, W, 1).to(DEV), fr[0].to(DEV), fr[1].to(DEV)
    mv, res = model.compress(ref, cur, sm, None)
    mv, res = dict(mv.items()), dict(res.items())
    seed, n_roots = None, 0
    for nm, enc, codec in (("mv", mv, model.mv_codec), ("res", res, model.res_codec)):
        _, ysym, yidx = PU.product_symbol_planes(codec, h, w, DEV)
        r = PU.explain_plane_mismatches(g, f"{pre}/enc/{nm}", ysym, yidx, codec.slice_depth, codec.max_support_slices, seed=seed)
        want = _enc(g, pre, nm)
        b_g, b_r = _nbytes(enc), _nbytes(want)
        print(f"[parity] s1080 f1 {nm}: {b_g} B vs reference {b_r} B ({100.0 * (b_g - b_r) / b_r:+.4f} %); differing symbols {r['symbols']} / "
              f"indexes {r['indexes']} of {r['total']}; roots {[(k, kind, f'{m:.1e}') for k, kind, m in r['roots']]}; unexplained {len(r['unexplained'])}")
        assert not r["unexplained"], f"{nm}: differences that no fragile decision of the reference explains: {r['unexplained'][:5]}"
        # observed on MI355X: ONE root per codec (margins 3.8e-7 / 4.8e-8), 18 / 65 differing symbols of 522 k / 783 k, +-4 bytes of 814 380
        assert len(r["roots"]) <= 3 and r["symbols"] <= 2e-4 * r["total"] and abs(b_g - b_r) <= 5e-5 * b_r, (len(r["roots"]), r["symbols"], r["total"])
        assert enc["strings"][1] == want["strings"][1] and tuple(enc["shape"]) == want["shape"], f"{nm}: hyper-latent string differs"
        n_roots += len(r["roots"])
        if r["where"].any():              # the residual codec codes a picture predicted with this codec's output: its input differs around here
            seed = PU._dilate(r["where"], 8) if seed is None else (seed | PU._dilate(r["where"], 8))
    # ---- the reference's strings through the GPU decoder: symbol for symbol up to the first fragile CDF lookup ----
    for codec in (model.mv_codec, model.res_codec):
        codec.debug_idx_taps = []
    try:
        out = model.decompress(ref, _enc(g, pre, "mv"), _enc(g, pre, "res"), sm, None)
        torch.cuda.synchronize()
    except N.NativeError as e:               # a desynchronised range decoder can also run out of stream
        out = None
        print(f"[parity] s1080 f1: decode of the reference's strings stopped: {e}")
    exact = {}
    for nm, codec in (("mv", model.mv_codec), ("res", model.res_codec)):
        first = PU.first_decoder_index_mismatch(g, f"{pre}/enc/{nm}", codec.debug_idx_taps, codec._plan(1, h, w, dev).groups, codec.slice_depth, h, w)
        codec.debug_idx_taps = None
        exact[nm] = first is None
        if first is None:
            print(f"[parity] s1080 f1 {nm}: the GPU decoder took every CDF lookup of the reference's string as the reference did")
        else:
            (c, y, x), margin, before = first
            print(f"[parity] s1080 f1 {nm}: first CDF lookup that differs from the reference's at (c, y, x) = ({c}, {y}, {x}) after {before} symbols; "
                  f"the reference's own margin there: {margin}")
            assert margin is not None and margin <= 2e-5, f"{nm}: the decoder left the reference's path at a decision that was not fragile"
    if out is not None and exact["mv"]:       # warped / predict depend on the motion stream only
        for j, name in ((2, "warped"), (3, "predict")):
            PU.check_golden_full(g, f"{pre}/dec/{name}", out[j], 1e-5, f"s1080 f1 {name} from the reference's mv string")
    if out is not None and exact["mv"] and exact["res"]:
        PU.check_golden_full(g, f"{pre}/dec/recon", out[1], 1e-5, "s1080 f1 recon from the reference's strings")
        PU.check_golden_full(g, f"{pre}/dec/feature", out[0], 1e-5, "s1080 f1 feature from the reference's strings")
    # ---- own strings -> decoded tensors against the reference's: equal except around the flipped decisions ----
    def own_decode_check(t, ref, mv, res, sm, feat, tol, blocks_tol):
        out = model.decompress(ref, mv, res, sm, feat)
        for j, name in ((1, "recon"), (2, "warped"), (3, "predict"), (0, "feature")):
            key = f"{tag}/f{t}/dec/{name}"
            e_s, e_b, _ = PU.golden_full_errors(g, key, out[j])
            a, blk = out[j].detach().float().cpu().numpy(), int(g[key + "/blk"][0])
            bm = a.astype(np.float64).reshape(*a.shape[:-2], a.shape[-2] // blk, blk, a.shape[-1] // blk, blk).mean(axis=(-3, -1))
            off = float((np.abs(bm - g[key + "/blockmean"]) > 1e-4).mean())
            print(f"[parity] s1080 f{t} own strings {name}: samples rel-L2 {e_s:.2e}, block means {e_b:.2e}, blocks off by > 1e-4: {100 * off:.3f} %")
            assert e_s <= tol and e_b <= tol and off <= blocks_tol, (name, e_s, e_b, off)
        p_g, p_r = PU.psnr(out[1].cpu(), fr[t]), float(g[f"{tag}/psnr"][t - 1])
        print(f"[parity] s1080 f{t} PSNR {p_g:.4f} vs the reference's {p_r:.4f} dB")
        assert abs(p_g - p_r) <= 0.005
        return out
    out = own_decode_check(1, ref, mv, res, sm, None, 5e-4 if n_roots else 1e-5, 0.005)
    # ---------------- frame 2 (steady state) on the GPU's own decoded state ----------------
    # its inputs differ from the reference's frame-2 inputs by the few flipped decisions of frame 1, so decisions are no longer
    # comparable one by one; rate, PSNR and decoded tensors are
    feat, ref = out[0], out[1]
    sm2 = synth.make_sm_fea(H, W, 2).to(DEV)
    mv2, res2 = model.compress(ref, fr[2].to(DEV), sm2, feat)
    b_g, b_r = _nbytes(mv2, res2), _nbytes(_enc(g, f"{tag}/f2", "mv"), _enc(g, f"{tag}/f2", "res"))
    print(f"[parity] s1080 f2: {b_g} B vs reference {b_r} B ({100.0 * (b_g - b_r) / b_r:+.4f} %)")
    assert abs(b_g - b_r) <= 1e-4 * b_r
    own_decode_check(2, ref, mv2, res2, sm2, feat, 5e-4, 0.005)


def test_1080p_f16_against_reference(g1080):
    _f16_chain_against_reference(g1080, "s1080", 1088, 1920, 2, features_of=(1, 2))


def test_1080p_intra_key_frame_against_reference(g1080):
    """ICIP2020ResB (image_model.py:490-591) at 1088x1920 against the reference's run, held to the P-FRAME standard since round 4: the synthetic
    I-frame weights are now a well-conditioned (structured) set that decodes a usable key frame (tools/calibrate_icip.py: 30.96 dB in the
    reference's own run).  fp32 mode: every differing HYPER-latent symbol must be a rounding the reference took within 2e-5 of flipping (a z
    root: ~1 expected per frame at this size -- 98 k decisions behind K = 4800 sums), every differing y symbol / index must be such a decision
    too or lie inside the receptive field downstream of a root (the fraction of the latent map that stays under the strict check is
    printed), bytes within 1e-4, identical strings when there is no root at all, block means of the decoded picture within 2e-3, PSNR
    within 0.005 dB; f16 mode: 1 % bytes, 0.05 dB, the f16 bound on the picture."""
    from deepsvc_amd import ops, synth
    g, x = g1080, synth.make_frames(1088, 1920, 1)[0].to(DEV)
    want = {"strings": [[g["icip1080/enc/y"].tobytes()], [g["icip1080/enc/z"].tobytes()]], "shape": tuple(int(v) for v in g["icip1080/enc/shape"])}
    ref_psnr = float(g["icip1080/psnr"][0])
    assert ref_psnr > 25.0, f"the reference's own key frame decodes at {ref_psnr:.2f} dB: not a usable picture"
    try:
        for mode in ("fp32", "f16"):
            ops.set_precision(mode)
            m = synth.make_icodec("W0")
            m.update(force=True)
            m = m.to(DEV)
            enc = m.compress(x)
            dec = m.decompress(enc["strings"], enc["shape"])["x_hat"]
            b_g, b_r = _nbytes(enc), _nbytes(want)
            e_s, e_b, _ = PU.golden_full_errors(g, "icip1080/dec/x_hat", dec)
            p_g = PU.psnr(dec.cpu(), x.cpu())
            line = (f"[parity] icip 1080p {mode}: {b_g} B vs reference {b_r} B ({100.0 * (b_g - b_r) / b_r:+.4f} %), PSNR {p_g:.4f} vs {ref_psnr:.4f} dB, "
                    f"x_hat samples rel-L2 {e_s:.2e}, block means {e_b:.2e}")
            if mode == "fp32":
                zsym, ysym, yidx = PU.product_symbol_planes(m, 68, 120, DEV)
                # hyper-latent first: 98 k rounding decisions behind three convs of K = 2880 .. 4800 terms -- about one of them per frame falls
                # within the ~3e-6 by which two fp32 summation orders differ; such a z root moves mu / sigma of the latent pixels around it
                rz = PU.explain_hyper_mismatches(g, "icip1080/enc/", zsym, (68, 120))
                r = PU.explain_plane_mismatches(g, "icip1080/enc/y", ysym, yidx, m.slice_depth, m.max_support_slices, seed=rz["seed"] if rz["seed"].any() else None)
                print(line + f"; z symbols differing {rz['differing']} of {rz['total']} (roots {[(k, f'{mg:.1e}') for k, mg in rz['roots']]}, unexplained {len(rz['unexplained'])}); "
                      f"y: differing symbols {r['symbols']} / indexes {r['indexes']} of {r['total']}, roots {[(k, kind, f'{mg:.1e}') for k, kind, mg in r['roots']]}, "
                      f"unexplained {len(r['unexplained'])}; share of the latent map outside every earlier root's reach, slice by slice: {[round(v, 2) for v in r['strictly_checked']]}")
                assert not rz["unexplained"], f"hyper-latent symbols differ where the reference's rounding was not fragile: {rz['unexplained'][:5]}"
                assert not r["unexplained"], f"differences that no fragile decision of the reference explains: {r['unexplained'][:5]}"
                # observed on MI355X: 1 z root (margin 2.4e-6), 18 y roots (margins 0 .. 2e-5) of 2.6 M decisions -- the I-frame codec's parameter
                # stacks sum K = 2880 .. 4320 terms (320-channel latent + supports), three times the P-frame codecs', hence more roots per decision
                assert len(rz["roots"]) <= 4 and len(r["roots"]) <= 40 and abs(b_g - b_r) <= 1e-4 * b_r, (len(rz["roots"]), len(r["roots"]))
                if not rz["roots"]:
                    assert enc["strings"][1] == want["strings"][1], "no z root, yet the hyper-latent string differs from the reference's"
                if not rz["roots"] and not r["roots"]:
                    assert enc["strings"][0] == want["strings"][0] and e_s <= 1e-5 and e_b <= 1e-5
                # a flipped symbol moves ONE decimated sample of the picture by 1/16: the picture agrees except there
                assert e_b <= 2e-3 and abs(p_g - ref_psnr) <= 0.005, (e_s, e_b, p_g)
            else:
                print(line)
                # f16 mode moves mu / sigma in their last bits, so its encoder legitimately picks the neighbouring symbol at some thousands of the
                # 2.6 M positions; with this codec's structured synthesis a symbol is ONE decimated pixel (step 1/16), so single samples differ
                # (observed 1.9e-2 on the strided samples) while rate, PSNR and the 16x16 block means (2.5e-3) stay on the reference's
                assert abs(b_g - b_r) <= 0.01 * b_r and abs(p_g - ref_psnr) <= 0.05 and e_b <= PU.F16_BOUNDS["recon"] and e_s <= 3e-2
    finally:
        ops.set_precision("fp32")


# ------------------------------------------------------------------------------------------------------------
# WHOLE GOPs against the reference: test_video.py:332-369 chains 6 (GOP = 7) or 11 (GOP = 12) P-frames on
# decoded state, so north_star's tolerances (|d bpp| <= 1 %, |d PSNR| <= 0.05 dB) have to hold at the LAST frame of the chain, in the
# mode the headline rate is quoted in.  Fixtures: tests/golden/make_golden_full.py gop1080p / gop256x1024 (the reference's own
# compress()/decompress() chained on its own decoded state; per frame: string lengths + CRC-32, PSNR, strided samples + block means of
# the decoded frames, the carried feature after the last frame).
# ------------------------------------------------------------------------------------------------------------
def _gop_chain(g, tag, H, W, n_p, precision, bounds, bytes_tol, psnr_tol, want_identical_first=0):
    """One precision mode on its OWN encode -> decode chain over the whole GOP against the reference's chain.  -> per-frame rows."""
    from deepsvc_amd import ops, synth
    import zlib
    model = _model(precision)
    rows = []
    try:
        fr = synth.make_frames(H, W, n_p + 1)
        ref, feat = fr[0].to(DEV), None
        ref_bytes, ref_crc = g[f"{tag}/string_bytes"], g[f"{tag}/string_crc32"]
        for t in range(1, n_p + 1):
            pre, sm, cur = f"{tag}/f{t}", synth.make_sm_fea(H, W, t).to(DEV), fr[t].to(DEV)
            mv, res = model.compress(ref, cur, sm, feat)
            out, _ = _decode(model, ref, mv, res, sm, feat)
            assert all(bool(torch.isfinite(o.float()).all()) for o in out)
            strs = [mv["strings"][0][0], mv["strings"][1][0], res["strings"][0][0], res["strings"][1][0]]
            same = [len(s) == int(ref_bytes[t - 1][k]) and zlib.crc32(s) == int(ref_crc[t - 1][k]) for k, s in enumerate(strs)]
            b_g, b_r = sum(len(s) for s in strs), int(ref_bytes[t - 1].sum())
            p_g, p_r = PU.psnr(out[1].cpu(), fr[t]), float(g[f"{tag}/psnr"][t - 1])
            errs = {name: PU.golden_full_errors(g, f"{pre}/dec/{name}", out[j]) for j, name in ((1, "recon"), (2, "warped"), (3, "predict"))}
            if t == n_p:
                errs["feature"] = PU.golden_full_errors(g, f"{pre}/dec/feature", out[0].float())
            rows.append(dict(frame=t, bytes=b_g, bytes_ref=b_r, identical=all(same), dpsnr=p_g - p_r, psnr_ref=p_r,
                             **{k: max(v[0], v[1]) for k, v in errs.items()}))
            print(f"[parity] {tag} f{t} {precision} vs reference: bytes {b_g} vs {b_r} ({100.0 * (b_g - b_r) / b_r:+.4f} %, identical strings: {same}), "
                  f"PSNR {p_g:.4f} vs {p_r:.4f} dB ({p_g - p_r:+.4f}); " + ", ".join(f"{k} {max(v[0], v[1]):.2e}" for k, v in errs.items()))
            feat, ref = out[0], out[1]
        for r in rows:
            t = r["frame"]
            if t <= want_identical_first:
                assert r["identical"], f"{tag} f{t}: {precision} strings differ from the reference's"
            assert abs(r["bytes"] - r["bytes_ref"]) <= bytes_tol * r["bytes_ref"], f"{tag} f{t}: {precision} {r['bytes']} B vs the reference's {r['bytes_ref']} B"
            assert abs(r["dpsnr"]) <= psnr_tol, f"{tag} f{t}: PSNR {precision} {r['dpsnr']:+.4f} dB from the reference's"
            for name in ("recon", "warped", "predict", "feature"):
                if name in r:
                    assert r[name] <= bounds[name], f"{tag} f{t}: {precision} {name} off the reference's by {r[name]:.2e} (> {bounds[name]:.1e})"
    finally:
        ops.set_precision("fp32")
    return rows


def test_gop12_256x1024_fp32_against_reference(ggopwide):
    """fp32 mode over config 3's chain length (11 P-frames): byte-identical strings for as long as no sub-1e-6-margin decision of the
    reference flips (observed: see the printed rows; the first three frames are asserted identical as in the short fixture), and in
    any case rate within 1e-3, PSNR within 0.005 dB and the decoded frames within 5e-4 of the reference's at every frame."""
    tight = dict(recon=5e-4, warped=5e-4, predict=5e-4, feature=5e-4)
    rows = _gop_chain(ggopwide, "g256x1024", 256, 1024, 11, "fp32", tight, 1e-3, 0.005, want_identical_first=3)
    print(f"[parity] g256x1024 fp32: strings identical to the reference's on frames {[r['frame'] for r in rows if r['identical']]}")


def test_gop12_256x1024_f16_against_reference(ggopwide):
    """f16 mode (the mode the headline rate is quoted in) through frame 11 of the chain: north_star's tolerances at EVERY frame."""
    _gop_chain(ggopwide, "g256x1024", 256, 1024, 11, "f16", PU.F16_GOP_BOUNDS, 0.01, 0.05)


def test_gop7_1080p_f16_against_reference(ggop1080):
    """BASELINE.json config 2 as the reference runs it: 6 chained P-frames at 1088x1920, f16 mode, north_star's tolerances at EVERY frame."""
    _gop_chain(ggop1080, "g1080", 1088, 1920, 6, "f16", PU.F16_GOP_BOUNDS, 0.01, 0.05)


def test_gop7_1080p_fp32_against_reference(ggop1080):
    """The same GOP in fp32 mode: a handful of sub-1e-6-margin decisions flip per frame at this size (test_1080p_fp32_against_reference
    proves where they come from), so strings are compared by length: rate within 1e-4, PSNR within 0.005 dB, decoded tensors 5e-4."""
    tight = dict(recon=5e-4, warped=5e-4, predict=5e-4, feature=5e-4)
    _gop_chain(ggop1080, "g1080", 1088, 1920, 6, "fp32", tight, 1e-4, 0.005)
