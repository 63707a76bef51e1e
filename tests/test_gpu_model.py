"""GPU parity of the module layer and of whole P-frames: the HIP path (through the C ABI) against the CPU oracle
on the same seeded inputs / W0 weights, and against the golden vectors captured from the reference itself.

Tolerances (fp32 mode): intermediate tensors 1e-3 of the tensor's max (deep fp32 conv chains with different
summation order); frames 2e-3 absolute.  Cross-device bit equality of rounded symbols is not attainable
(SURVEY section 7.2), so bitstream-level parity is stated as: symbol mismatch fraction <= 0.5 %, |d bpp| <= 1 %,
|d PSNR| <= 0.05 dB -- the tolerances of BASELINE.json's north_star."""
import numpy as np
import pytest
import torch

from gpu_util import psnr, report_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available()
    from deepsvc_amd import ops, synth
    ops.set_precision("fp32")
    m = synth.make_model("W0")
    m.update(force=True)
    return m.to(DEV)


def _rel(name, got, want, frac=1e-3):
    report_close(name, got, want, 0.0, frac * max(float(want.abs().max()), 1e-6))


def _inputs(size, t=1):
    from deepsvc_amd import synth
    fr = synth.make_frames(size, size, 3)
    return fr, synth.make_sm_fea(size, size, t)


def test_spynet_and_refine(model, oracle_w0):
    from oracle import nets
    fr, _ = _inputs(64)
    root = oracle_w0.root
    flow = model.opticFlow(fr[1].to(DEV), fr[0].to(DEV)).cpu()
    want = nets.spynet(root.sub("opticFlow"), fr[1], fr[0])
    _rel("spynet", flow, want)
    got = model.RefineMvNet(want.to(DEV), fr[0].to(DEV)).cpu()
    _rel("refine mv", got, nets.refine_net(root.sub("RefineMvNet"), want, fr[0]))


def test_feature_blocks(model, oracle_w0):
    from oracle import nets
    torch.manual_seed(0)
    fr, sm = _inputs(64)
    root = oracle_w0.root
    _rel("feature extraction", model.FeatureExtractor(fr[1].to(DEV)).cpu(),
         nets.feature_extraction(root.sub("FeatureExtractor"), fr[1]))
    x = torch.randn(1, 64, 64, 64) * 0.3
    _rel("unet", model.enhance.unet_1(x.to(DEV)).cpu(), nets.unet(root.sub("enhance.unet_1"), x))
    x96 = torch.randn(1, 96, 64, 64) * 0.3
    f, r = model.enhance(x96.to(DEV))
    fw, rw = nets.reconstruction(root.sub("enhance"), x96)
    _rel("enhance feature", f.cpu(), fw)
    _rel("enhance recon", r.cpu(), rw)
    mv = torch.randn(1, 2, 64, 64) * 3
    warped = nets.warp(fr[0], mv)
    feat = torch.randn(1, 64, 64, 64) * 0.3
    for feature in (None, feat):
        up, pred = model.MC(fr[0].to(DEV), warped.to(DEV), mv.to(DEV), sm.to(DEV), None if feature is None else feature.to(DEV))
        upw, predw = nets.inter_layer_prediction(root.sub("MC"), fr[0], warped, mv, sm, feature)
        _rel(f"MC up_out feature={'set' if feature is not None else None}", up.cpu(), upw)
        _rel("MC predict", pred.cpu(), predw)


@pytest.mark.parametrize("name,cin", [("mv_codec", 8), ("res_codec", 70)])
def test_codec_forward(model, oracle_w0, name, cin):
    from oracle import codec
    torch.manual_seed(1)
    x = torch.randn(1, cin, 64, 64) * 0.5
    got = getattr(model, name)(x.to(DEV))
    taps = {}
    want = codec.codec_forward(oracle_w0.root.sub(name), x, taps)
    # rounding can flip on a few latents; compare likelihood mass and x_hat robustly
    for k in ("y", "z"):
        g, w = got["likelihoods"][k].cpu(), want["likelihoods"][k]
        frac = float(((g - w).abs() > 1e-3 * w.abs() + 1e-6).float().mean())
        assert frac < 0.01, f"{name} likelihood {k}: {frac:.4f} of elements differ"
        np.testing.assert_allclose(float(torch.log2(g).sum()), float(torch.log2(w).sum()), rtol=1e-2)
    d = (got["x_hat"].cpu() - want["x_hat"]).abs()
    assert float((d > 1e-3 * float(want["x_hat"].abs().max())).float().mean()) < 0.02, f"{name} x_hat differs on too many pixels"


@pytest.mark.parametrize("tag,steady", [("s64_first", False), ("s64_steady", True)])
def test_forward_matches_oracle_and_reference_golden(model, oracle_w0, golden64, tag, steady):
    from deepsvc_amd import synth
    fr = synth.make_frames(64, 64, 3)
    if steady:
        sm1 = synth.make_sm_fea(64, 64, 1)
        mv, res = oracle_w0.compress(fr[0], fr[1], sm1, None)
        feat, ref, _, _ = oracle_w0.decompress(fr[0], mv, res, sm1, None)
        cur, sm = fr[2], synth.make_sm_fea(64, 64, 2)
    else:
        ref, cur, sm, feat = fr[0], fr[1], synth.make_sm_fea(64, 64, 1), None
    got = model(ref.to(DEV), cur.to(DEV), sm.to(DEV), None if feat is None else feat.to(DEV))
    want = oracle_w0.forward(ref, cur, sm, feat)
    gold = golden64[f"{tag}/fwd/scalars"]            # produced by the reference itself
    recon, recon_w = got[0].cpu(), want[0]
    assert psnr(recon, recon_w) > 45, f"recon PSNR vs oracle {psnr(recon, recon_w):.1f} dB"
    g = [float(v) for v in got[2:]]
    w = [float(v) for v in want[2:]]
    print(tag, "gpu scalars", g, "oracle", w, "reference", list(gold))
    names = ["mse", "warp_mse", "mc_mse", "bpp_res", "bpp_mv", "bpp"]
    for n, a, b, c in zip(names, g, w, gold):
        tol = 0.01 if n.startswith("bpp") else 0.02
        assert abs(a - b) <= tol * abs(b), f"{n}: gpu {a} vs oracle {b}"
        assert abs(a - c) <= tol * abs(c), f"{n}: gpu {a} vs reference golden {c}"
    # PSNR delta of the reconstruction against the current frame, GPU vs reference
    d_psnr = abs(-10 * np.log10(g[0]) + 10 * np.log10(gold[0]))
    assert d_psnr <= 0.05, f"|dPSNR| = {d_psnr:.4f} dB"


@pytest.mark.parametrize("size", [64, 128])
def test_compress_decompress_roundtrip_and_parity(model, oracle_w0, size):
    from deepsvc_amd import synth
    from oracle.codec import strings_nbytes
    fr = synth.make_frames(size, size, 3)
    ref_g, ref_o, feat_g, feat_o = fr[0].to(DEV), fr[0], None, None
    for t in (1, 2):                       # two chained P-frames: first (feature=None) and steady-state
        sm = synth.make_sm_fea(size, size, t)
        cur = fr[t]
        mv_g, res_g = model.compress(ref_g, cur.to(DEV), sm.to(DEV), feat_g)
        dec_g = model.decompress(ref_g, mv_g, res_g, sm.to(DEV), feat_g)
        mv_o, res_o = oracle_w0.compress(ref_o, cur, sm, feat_o)
        dec_o = oracle_w0.decompress(ref_o, mv_o, res_o, sm, feat_o)
        for enc in (mv_g, res_g):
            assert isinstance(enc["strings"][0][0], bytes) and isinstance(enc["strings"][1][0], bytes)
            assert tuple(enc["shape"]) == (size // 64, size // 64)
        bytes_g = strings_nbytes(mv_g) + strings_nbytes(res_g)
        bytes_o = strings_nbytes(mv_o) + strings_nbytes(res_o)
        print(f"size {size} frame {t}: bytes gpu {bytes_g} oracle {bytes_o}; "
              f"mv y equal {mv_g['strings'][0][0] == mv_o['strings'][0][0]}, res y equal {res_g['strings'][0][0] == res_o['strings'][0][0]}")
        assert abs(bytes_g - bytes_o) <= max(0.01 * bytes_o, 8), "bpp differs by more than 1 %"
        recon_g, recon_o = dec_g[1].cpu(), dec_o[1]
        p_g, p_o = psnr(recon_g, cur), psnr(recon_o, cur)
        print(f"   PSNR(recon,cur): gpu {p_g:.4f} oracle {p_o:.4f}; PSNR(gpu,oracle) {psnr(recon_g, recon_o):.1f} dB")
        assert abs(p_g - p_o) <= 0.05, "PSNR delta above 0.05 dB"
        assert psnr(recon_g, recon_o) > 40
        assert float(recon_g.min()) >= 0 and float(recon_g.max()) <= 1
        # decoder reproduces the encoder's prediction path exactly (deterministic kernels): decoding twice is bit-stable
        dec_g2 = model.decompress(ref_g, mv_g, res_g, sm.to(DEV), feat_g)
        assert torch.equal(dec_g2[1], dec_g[1]) and torch.equal(dec_g2[0], dec_g[0])
        # estimate path and real path agree on the reconstruction (they differ only by ste_round's float path)
        fwd = model(ref_g, cur.to(DEV), sm.to(DEV), feat_g)
        assert psnr(fwd[0].clamp(0, 1).cpu(), recon_g) > 60
        ref_g, feat_g = dec_g[1], dec_g[0]
        ref_o, feat_o = dec_o[1], dec_o[0]


def test_golden_bitstream_decodes_on_gpu_when_indexes_agree(model, golden64, oracle_w0):
    """The reference's own y/z strings (golden) fed to the GPU decoder.  Index agreement between the GPU scan and
    the CPU scan that produced the stream is not guaranteed bit-for-bit, so this test asserts the weaker, exact
    property on the z-stream (no data-dependent indexes) and reports the y outcome."""
    g = golden64
    shape = tuple(int(v) for v in g["s64_first/enc/mv_shape"])
    z_bytes = g["s64_first/enc/mv_z"].tobytes()
    z_hat = model.mv_codec.entropy_bottleneck.decompress([z_bytes], shape).cpu()
    from oracle import entropy as E
    sd, pre = oracle_w0.sd, "mv_codec.entropy_bottleneck."
    want = E.eb_decompress(sd, pre, oracle_w0.tables["mv"]["eb"], [z_bytes], shape)
    assert torch.equal(z_hat, want)


def test_bf16_mode_end_to_end(model, oracle_w0):
    """Fast mode (bf16 MFMA, bf16 activation storage): same API, round trip still exact between encoder and
    decoder; quality/rate deltas against the fp32 oracle are reported and bounded."""
    from deepsvc_amd import ops, synth
    from oracle.codec import strings_nbytes
    size = 128
    fr = synth.make_frames(size, size, 2)
    sm = synth.make_sm_fea(size, size, 1)
    mv_o, res_o = oracle_w0.compress(fr[0], fr[1], sm, None)
    dec_o = oracle_w0.decompress(fr[0], mv_o, res_o, sm, None)
    ops.set_precision("bf16")
    try:
        mv_g, res_g = model.compress(fr[0].to(DEV), fr[1].to(DEV), sm.to(DEV), None)
        dec_g = model.decompress(fr[0].to(DEV), mv_g, res_g, sm.to(DEV), None)
        dec_g2 = model.decompress(fr[0].to(DEV), mv_g, res_g, sm.to(DEV), None)
    finally:
        ops.set_precision("fp32")
    assert torch.equal(dec_g[1], dec_g2[1])
    b_g, b_o = strings_nbytes(mv_g) + strings_nbytes(res_g), strings_nbytes(mv_o) + strings_nbytes(res_o)
    p_g, p_o = psnr(dec_g[1].cpu(), fr[1]), psnr(dec_o[1], fr[1])
    print(f"bf16: bytes {b_g} vs {b_o} ({100 * (b_g - b_o) / b_o:+.2f} %), PSNR {p_g:.4f} vs {p_o:.4f} dB, "
          f"PSNR(gpu,oracle) {psnr(dec_g[1].cpu(), dec_o[1]):.1f} dB")
    assert abs(b_g - b_o) <= 0.03 * b_o
    assert abs(p_g - p_o) <= 0.1


def test_full_size_1080p_roundtrip_properties(oracle_w0):
    """BASELINE.json configs[1] size (1920x1080 -> padded 1088x1920), where the CPU oracle is too slow to be the
    checker: size-independent properties of the real bitstream path instead.
      * encode -> decode round trip: the decoder's latents y_hat (both codecs) are bit-identical to the encoder's;
      * determinism / idempotence: decoding the same strings twice gives bit-identical frames and features;
      * a second, chained P-frame (feature and reference from the decoder) behaves the same;
      * bf16 mode stays inside the north-star tolerance of fp32 mode on the same inputs: |d bytes| <= 1 %, and the two
        reconstructions agree to > 40 dB."""
    from deepsvc_amd import ops, synth
    from deepsvc_amd.modules import pad

    def run(precision):
        ops.set_precision(precision)
        try:
            model = synth.make_model("W0")
            model.update(force=True)
            model = model.to(DEV)
            frames = [pad(f.to(DEV), 64) for f in synth.make_frames(1080, 1920, 3, seed=7)]
            sms = [None] + [synth.make_sm_fea(1088, 1920, t, seed=7).to(DEV) for t in (1, 2)]
            ref, feat, out = frames[0], None, []
            for t in (1, 2):
                mv, res = model.compress(ref, frames[t], sms[t], feat)
                mv, res = {k: mv[k] for k in ("strings", "shape")}, {k: res[k] for k in ("strings", "shape")}   # joins the coders
                torch.cuda.synchronize()
                enc_lat = [c._plan(1, 68, 120, torch.device(DEV)).y_hat.buf.clone() for c in (model.mv_codec, model.res_codec)]
                f1, r1, w1, p1 = model.decompress(ref, mv, res, sms[t], feat)
                torch.cuda.synchronize()
                dec_lat = [c._plan(1, 68, 120, torch.device(DEV)).y_hat.buf.clone() for c in (model.mv_codec, model.res_codec)]
                f2, r2, w2, p2 = model.decompress(ref, mv, res, sms[t], feat)
                torch.cuda.synchronize()
                for a, b, what in zip(enc_lat, dec_lat, ("mv", "res")):
                    assert torch.equal(a, b), f"{precision} frame {t}: decoder {what} latents differ from the encoder's"
                assert torch.equal(r1, r2) and torch.equal(w1, w2) and torch.equal(p1, p2), "decode is not deterministic"
                assert torch.equal(f1.float(), f2.float())
                nbytes = sum(len(s[0]) for s in mv["strings"]) + sum(len(s[0]) for s in res["strings"])
                assert nbytes > 1000 and tuple(mv["shape"]) == (17, 30)
                assert torch.isfinite(r1).all() and float(r1.min()) >= 0.0 and float(r1.max()) <= 1.0
                out.append((nbytes, r1.float().cpu()))
                ref, feat = r1, f1
            return out
        finally:
            ops.set_precision("fp32")

    a, b = run("fp32"), run("bf16")
    for t, ((na, ra), (nb, rb)) in enumerate(zip(a, b), 1):
        assert abs(nb - na) <= 0.01 * na, f"frame {t}: bf16 {nb} B vs fp32 {na} B"
        mse = float(((ra - rb) ** 2).mean())
        psnr = 99.0 if mse == 0 else -10.0 * float(np.log10(mse))
        print(f"1080p frame {t}: fp32 {na} B, bf16 {nb} B ({100.0 * (nb - na) / na:+.3f} %), PSNR(bf16 recon, fp32 recon) = {psnr:.1f} dB")
        assert psnr > 40.0


def test_frame_container_round_trip_on_gpu(model):
    """Encoder output -> on-disk container bytes -> decoder: identical pictures, exact overhead."""
    from deepsvc_amd import bitstream, ops, synth
    ops.set_precision("fp32")
    size = 128
    fr = synth.make_frames(size, size, 2)
    sm = synth.make_sm_fea(size, size, 1).to(DEV)
    mv, res = model.compress(fr[0].to(DEV), fr[1].to(DEV), sm, None)
    blob = bitstream.pack_frame(mv, res)
    ftype, (mv2, res2) = bitstream.unpack_frame(blob)
    assert ftype == bitstream.TYPE_P
    payload = sum(len(s[0]) for e in (mv, res) for s in e["strings"])
    assert len(blob) == payload + bitstream.frame_overhead_bytes(mv, res)
    a = model.decompress(fr[0].to(DEV), mv, res, sm, None)
    b = model.decompress(fr[0].to(DEV), mv2, res2, sm, None)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0].float(), b[0].float())


def test_w1_weight_set_parity(oracle_w0):
    """BASELINE.json config 4's "MS-SSIM model" is the same code path with another weight set: W1 (seed 1) through the
    real bitstream path in fp32 mode against the oracle on the same weights -- bytes within 1 %, PSNR within 0.05 dB."""
    from deepsvc_amd import ops, synth
    from oracle.codec import DeepSVCOracle, strings_nbytes
    ops.set_precision("fp32")
    m = synth.make_model("W1")
    m.update(force=True)
    orc = DeepSVCOracle(m.state_dict())
    orc.update(force=True)
    m = m.to(DEV)
    size = 64
    fr = synth.make_frames(size, size, 2)
    sm = synth.make_sm_fea(size, size, 1)
    mv_g, res_g = m.compress(fr[0].to(DEV), fr[1].to(DEV), sm.to(DEV), None)
    dec_g = m.decompress(fr[0].to(DEV), mv_g, res_g, sm.to(DEV), None)
    mv_o, res_o = orc.compress(fr[0], fr[1], sm, None)
    dec_o = orc.decompress(fr[0], mv_o, res_o, sm, None)
    bg, bo = strings_nbytes(mv_g) + strings_nbytes(res_g), strings_nbytes(mv_o) + strings_nbytes(res_o)
    print(f"W1 64x64: bytes gpu {bg} oracle {bo}; identical y strings: mv {mv_g['strings'][0][0] == mv_o['strings'][0][0]}, "
          f"res {res_g['strings'][0][0] == res_o['strings'][0][0]}")
    assert abs(bg - bo) <= max(0.01 * bo, 8)
    assert abs(psnr(dec_g[1].cpu(), fr[1]) - psnr(dec_o[1], fr[1])) <= 0.05
    assert psnr(dec_g[1].cpu(), dec_o[1]) > 40


def test_4k_whole_frame_round_trip_bf16():
    """BASELINE.json config 5 size (3840x2160 -> padded 2176x3840) as a whole frame on one GPU in bf16 mode (DESIGN.md
    section 6: spatial tiling is not result-preserving, so 4K runs un-tiled): the encode -> decode round trip holds
    (decoder latents == encoder latents), decoding is deterministic, every buffer stays under the 2 GiB extent limit."""
    from deepsvc_amd import ops, synth
    from deepsvc_amd.modules import pad
    ops.set_precision("bf16")
    try:
        m = synth.make_model("W0"); m.update(force=True); m = m.to(DEV)
        frames = [pad(f.to(DEV), 64) for f in synth.make_frames(2160, 3840, 2, seed=3)]
        assert tuple(frames[0].shape[-2:]) == (2176, 3840)
        sm = synth.make_sm_fea(2176, 3840, 1, seed=3).to(DEV)
        mv, res = m.compress(frames[0], frames[1], sm, None)
        mv, res = {k: mv[k] for k in ("strings", "shape")}, {k: res[k] for k in ("strings", "shape")}
        torch.cuda.synchronize()
        dev = torch.device(DEV)
        enc_lat = [c._plan(1, 136, 240, dev).y_hat.buf.clone() for c in (m.mv_codec, m.res_codec)]
        f1, r1, _, _ = m.decompress(frames[0], mv, res, sm, None)
        torch.cuda.synchronize()
        dec_lat = [c._plan(1, 136, 240, dev).y_hat.buf.clone() for c in (m.mv_codec, m.res_codec)]
        f2, r2, _, _ = m.decompress(frames[0], mv, res, sm, None)
        assert all(torch.equal(a, b) for a, b in zip(enc_lat, dec_lat))
        assert torch.equal(r1, r2) and torch.equal(f1.float(), f2.float())
        nbytes = sum(len(s[0]) for e in (mv, res) for s in e["strings"])
        print(f"4K frame: {nbytes} B = {nbytes * 8 / (2160 * 3840):.3f} bpp, peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        assert tuple(mv["shape"]) == (34, 60) and torch.isfinite(r1).all()
    finally:
        ops.set_precision("fp32")


def test_gop_lanes_match_sequential_coding():
    """harness.code_gops_in_lanes (throughput mode: independent GOPs on concurrent lanes of one GPU, each with its own model
    instance / stream / thread) returns exactly the rows of coding the GOPs one after the other."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("bf16")
    try:
        size, n_gops = 256, 3
        gops = []
        for u in range(n_gops):
            frames = [f.to(DEV) for f in synth.make_frames(size, size, 3, seed=100 + u)]
            sms = [None] + [synth.make_sm_fea(size, size, t, seed=100 + u).to(DEV) for t in (1, 2)]
            gops.append((frames, sms))
        models = []
        for _ in range(2):
            m = synth.make_model("W0"); m.update(force=True); models.append(m.to(DEV))
        seq = [row for u, (fr, sm) in enumerate(gops) for row in harness.code_gop(models[0], fr, sm, unit=u, sync=torch.cuda.synchronize)]
        par = harness.code_gops_in_lanes(models, gops, torch.device(DEV))
        assert len(par) == len(seq) == n_gops * 2
        for a, b in zip(seq, par):
            assert a[:7] == b[:7], (a[:7], b[:7])            # unit, frame, bpp_mv, bpp_res, psnr x3: bit-identical results
    finally:
        ops.set_precision("fp32")
