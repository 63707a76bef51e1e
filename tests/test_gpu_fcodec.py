"""GPU parity of the semantic-layer feature codec (reference semantic_layer.py:1146-1519): the HIP path
through the C ABI against the CPU oracle (oracle/fcodec.py, pinned byte for byte against the reference classes) and against
the reference's golden tensors / bitstreams; RefineNet1's aggregation kernel; the stage-1 glue feeding `sm_fea` to DeepSVC."""
import os

import numpy as np
import pytest
import torch

import parity_util as PU

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _make(precision):
    from deepsvc_amd import ops, synth
    ops.set_precision(precision)
    m = synth.make_student_layer1("W0")
    m.update(force=True)
    return m.to(DEV)


@pytest.fixture(scope="module")
def oracle():
    from deepsvc_amd import synth
    from oracle.fcodec import StudentLayer1Oracle
    o = StudentLayer1Oracle(synth.make_student_layer1("W0").state_dict())
    o.update()
    return o


@pytest.mark.parametrize("size", [64, 128])
def test_fcodec_fp32_matches_oracle_and_reference(oracle, size):
    from deepsvc_amd import ops, synth
    from oracle import fcodec as C
    try:
        m = _make("fp32")
        g = np.load(os.path.join(GOLD, f"fcodec_ref_{size}.npz"))
        tag = f"s{size}"
        x, supp = synth.make_semantic_input(size, size, 1)
        fea = supp[-1:]
        xg, fg = x.to(DEV), fea.to(DEV)
        # estimate path
        want = C.forward(oracle.codec, x, fea)
        got = m.layer1(xg, fea=fg)
        for k in ("x_hat", "y"):
            e = PU.rel_l2(got[k].cpu(), want[k])
            print(f"[fcodec] {size} forward {k}: rel-L2 {e:.2e}")
            assert e <= 1e-5
            PU.check_golden(g, f"{tag}/fwd/{k}", got[k], rel=1e-5)
        for j, k in enumerate(("y", "z")):
            a = float(torch.log(got["likelihoods"][k]).sum())
            assert abs(a - g[f"{tag}/fwd/scalars"][j]) <= 2e-3 * abs(g[f"{tag}/fwd/scalars"][j]) + 1e-2, (k, a)
        # bitstream path: bytes identical to the oracle's and to the reference's
        enc = m.layer1.compress(xg, fg)
        enc_o = C.compress(oracle.codec, oracle.tabs, x, fea)
        for j, name in ((0, "y"), (1, "z")):
            assert enc["strings"][j][0] == enc_o["strings"][j][0] == g[f"{tag}/enc/{name}"].tobytes(), f"{name}-string differs"
        assert tuple(enc["shape"]) == (size // 64, size // 64)
        PU.check_golden(g, f"{tag}/enc/latent_y", enc["y"], rel=1e-5)
        dec = m.layer1.decompress(enc["strings"], enc["shape"], fg)
        PU.check_golden(g, f"{tag}/dec/x_hat", dec["x_hat"], rel=1e-5)
        # the REFERENCE's strings through the GPU decoder
        gold = [[g[f"{tag}/enc/y"].tobytes()], [g[f"{tag}/enc/z"].tobytes()]]
        dec2 = m.layer1.decompress(gold, tuple(int(v) for v in g[f"{tag}/enc/shape"]), fg)
        assert torch.equal(dec2["x_hat"], dec["x_hat"])
        # RefineNet1 (both correlation modes) on the decoded feature
        for mode, mean in (("mean", True), ("sum", False)):
            agg = m.refine(dec["x_hat"], supp.to(DEV), mean=mean)
            e = PU.check_golden(g, f"{tag}/refine/{mode}", agg, rel=1e-5)
            print(f"[fcodec] {size} RefineNet1 {mode}: rel-L2 vs reference {e:.2e}")
        # a decoder that ignores the side feature must fail these checks: its entropy parameters differ, so the range decoder
        # desynchronises (wrong symbols, or the stream runs out) and the decoded feature is wrong
        from deepsvc_amd._native import NativeError
        with pytest.raises((AssertionError, NativeError)):
            dec_wrong = m.layer1.decompress(enc["strings"], enc["shape"], torch.zeros_like(fg))
            PU.check_golden(g, f"{tag}/dec/x_hat", dec_wrong["x_hat"], rel=1e-5)
    finally:
        ops.set_precision("fp32")


def test_refine_aggregate_is_not_the_identity_shortcut():
    """x + sum_l x*w_l with softmax weights equals 2x only up to rounding; the kernel evaluates the reference's expression (norms,
    correlations, softmax over the supports) -- checked against torch on random features with widely spread correlations."""
    from deepsvc_amd import ops
    from deepsvc_amd.ops import to_act
    torch.manual_seed(0)
    x, rx, rs = torch.randn(1, 256, 9, 13), torch.randn(1, 256, 9, 13) * 3, torch.randn(5, 256, 9, 13) * 3
    rs[2] = rx * 0.9 + 0.1 * rs[2]                                   # one support strongly correlated
    for mean in (True, False):
        got = ops.refine_aggregate(to_act(x.to(DEV)), to_act(rx.to(DEV)), to_act(rs.to(DEV)), mean=mean, relu=True).nchw().cpu()
        n = lambda t: t / t.norm(p=2, dim=1, keepdim=True)
        c = n(rx) * n(rs)
        w = (c.mean(1, keepdim=True) if mean else c.sum(1, keepdim=True)).softmax(0)
        want = torch.relu(x + (x * w).sum(0, keepdim=True))
        assert PU.rel_l2(got, want) < 1e-6
        if not mean:
            assert float(w.max()) > 0.3                               # the soft-max is far from uniform in this mode


def test_stage1_feeds_the_p_frame_codec(oracle):
    """StudentLayer1 (codec -> RefineNet1 -> ReLU; semantic_layer.py:1064-1082) against the oracle, then its output as the
    `sm_fea` of DeepSVC.compress / decompress (test_video.py:336,342,349) and its rate / times in the reference's JSON."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("fp32")
    m = _make("fp32")
    size = 128
    x, supp = synth.make_semantic_input(size, size, 1)
    sm, y = m(x.to(DEV), supp.to(DEV), mean=True, encode=True)
    sm_o, y_o = oracle.forward(x, supp, mean=True, encode=True)
    assert m.bpp_loss == oracle.bpp_loss and m.enct > 0 and m.dect > 0
    assert PU.rel_l2(sm.cpu(), sm_o) <= 1e-5 and PU.rel_l2(y.cpu(), y_o) <= 1e-5
    assert tuple(sm.shape) == (1, 256, size // 4, size // 4) and float(sm.min()) >= 0.0
    est, _ = m(x.to(DEV), supp.to(DEV), mean=True, encode=False)
    assert PU.rel_l2(est.cpu(), sm_o) < 1e-3 and 0.5 * float(m.bpp_loss) < oracle.bpp_loss < 1.6 * float(m.bpp_loss)
    p = synth.make_model("W0"); p.update(force=True); p = p.to(DEV)
    fr = [f.to(DEV) for f in synth.make_frames(size, size, 2)]
    mv, res = p.compress(fr[0], fr[1], sm, None)
    out = p.decompress(fr[0], mv, res, sm, None)
    assert torch.isfinite(out[1]).all()
    rows = [[0.0, 1.0, harness.strings_bytes(mv) * 8.0 / size ** 2, harness.strings_bytes(res) * 8.0 / size ** 2, 30.0, 29.0, 28.0, 0.01, 0.02, 0, 0, 0]]
    r = harness.reference_results(torch.tensor(rows, dtype=torch.float64), sm_rows={0.0: [(oracle.bpp_loss, m.enct, m.dect)]})
    assert abs(r["smbpp"][0] - oracle.bpp_loss) < 1e-12 and abs(r["pbpp"][0] - (rows[0][2] + rows[0][3] + oracle.bpp_loss)) < 1e-12


def test_fcodec_f16_mode(oracle):
    from deepsvc_amd import ops, synth
    from oracle import fcodec as C
    try:
        m = _make("f16")
        size = 128
        x, supp = synth.make_semantic_input(size, size, 1)
        fea = supp[-1:]
        enc = m.layer1.compress(x.to(DEV), fea.to(DEV))
        dec = m.layer1.decompress(enc["strings"], enc["shape"], fea.to(DEV))
        dec2 = m.layer1.decompress(enc["strings"], enc["shape"], fea.to(DEV))
        assert torch.equal(dec["x_hat"], dec2["x_hat"])
        enc_o = C.compress(oracle.codec, oracle.tabs, x, fea)
        dec_o = C.decompress(oracle.codec, oracle.tabs, enc_o["strings"], enc_o["shape"], fea)
        nb, no = sum(len(s[0]) for s in enc["strings"]), sum(len(s[0]) for s in enc_o["strings"])
        e = PU.rel_l2(dec["x_hat"].cpu(), dec_o["x_hat"])
        print(f"[fcodec] f16: bytes {nb} vs {no} ({100.0 * (nb - no) / no:+.2f} %), decoded feature rel-L2 {e:.2e}")
        assert abs(nb - no) <= 0.01 * no + 8 and e <= 5e-3
    finally:
        ops.set_precision("fp32")


def test_self_contained_gop_with_semantic_layer():
    """test_video.py:336-338 loop shape: every P-frame's `sm_fea` comes from the semantic layer's codec (stage 1 of
    OursResNetStudentP) run on `process(cur * 255)` against a rolling support stack; its rate and times fill the semantic-layer
    keys of the reference JSON.  Coding the GOP twice gives identical rows (deterministic, encoder / decoder lock-step)."""
    from deepsvc_amd import harness, ops, synth
    ops.set_precision("f16")
    try:
        size, n = 128, 4
        sm_layer = _make("f16")
        p = synth.make_model("W0"); p.update(force=True); p = p.to(DEV)
        frames = [f.to(DEV) for f in synth.make_frames(size, size, n)]
        _, supp0 = synth.make_semantic_input(size, size, 0)
        runs = []
        for _ in range(2):
            log = []
            rows = harness.code_gop(p, frames, [None] * n, unit=7, sync=torch.cuda.synchronize, sm_layer=sm_layer, supp=supp0.to(DEV), sm_log=log)
            runs.append(([r[:7] for r in rows], [l[0] for l in log]))
            assert len(rows) == n - 1 and len(log) == n - 1 and all(l[0] > 0 and l[1] > 0 and l[2] > 0 for l in log)
        assert runs[0] == runs[1]
        res = harness.reference_results(torch.tensor(rows, dtype=torch.float64), sm_rows={7.0: log})
        assert res["smbpp"][0] > 0 and abs(res["pbpp"][0] - (res["mvbpp"][0] + res["resbpp"][0] + res["smbpp"][0])) < 1e-12
        assert abs(res["bpp2l"][0] - (res["mvbpp"][0] + res["smbpp"][0])) < 1e-12 and res["pent"][0] > res["sment"][0] > 0
        # harness.process == test_video.py:176-190 (normalise channel i with (mean_i, std_i), return channels reversed)
        x = torch.rand(1, 3, 8, 8) * 255
        want = torch.cat([(x[:, 2:3] - 103.53) / 57.375, (x[:, 1:2] - 116.28) / 57.12, (x[:, 0:1] - 123.675) / 58.395], 1)
        assert torch.allclose(harness.process(x), want)
    finally:
        ops.set_precision("fp32")
