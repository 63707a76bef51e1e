"""GPU parity of every HIP op, called through the C ABI (deepsvc_amd.ops), against plain PyTorch-CPU fp32
(or the oracle's restatement) on the same seeded inputs.  Tolerances: fp32 MFMA path 1e-4 relative to the
output scale; the f16 MFMA path is compared with an fp32 reference fed the same f16-rounded operands at HALF AN ULP of the
stored value plus 3e-5 of the tensor scale for the summation order (gpu_util.close_f16) -- tight enough that one wrong tap fails."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import close_f16, report_close

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from deepsvc_amd import _native
    assert _native.lib().dsvc_device_count() >= 1
    from deepsvc_amd import ops
    ops.set_precision("fp32")
    yield
    ops.set_precision("fp32")


def _act(x, dtype=torch.float32):
    from deepsvc_amd import ops
    return ops.to_act(x.to(DEV), dtype)


def _torch_act(v, code, slope=0.01):
    from deepsvc_amd import _native as N
    return {N.ACT_NONE: lambda t: t, N.ACT_RELU: F.relu, N.ACT_LRELU: lambda t: F.leaky_relu(t, slope),
            N.ACT_GELU: F.gelu, N.ACT_SIGMOID: torch.sigmoid, N.ACT_HALF_TANH: lambda t: 0.5 * torch.tanh(t)}[code](v)


CONV_CASES = [
    # (segments, Cout, k, stride, H, W)
    ((64,), 64, 3, 1, 20, 37), ((64,), 64, 3, 1, 8, 32), ((3,), 64, 3, 1, 17, 33), ((5,), 64, 3, 1, 9, 40),
    ((64, 3), 64, 3, 1, 12, 34), ((96,), 96, 3, 1, 10, 31), ((128,), 128, 3, 1, 9, 33), ((64,), 3, 3, 1, 21, 35),
    ((64,), 2, 3, 1, 8, 8), ((16,), 64, 3, 1, 1, 1), ((32,), 3, 3, 1, 40, 70), ((64, 64), 64, 3, 1, 11, 30),
    ((3, 64, 3), 96, 5, 2, 32, 48), ((3, 2, 3), 64, 5, 2, 16, 64), ((96,), 96, 5, 2, 8, 12), ((64,), 64, 3, 2, 8, 14),
    ((8,), 32, 7, 1, 9, 35), ((32,), 64, 7, 1, 12, 20), ((16,), 2, 7, 1, 16, 33), ((64,), 64, 1, 1, 13, 29),
    ((128,), 256, 1, 1, 6, 10), ((64, 24, 8), 64, 3, 1, 5, 9), ((96, 48), 96, 3, 1, 4, 8), ((156,), 96, 3, 1, 4, 30),
    ((64,), 8, 3, 1, 4, 8), ((32,), 12, 3, 1, 5, 7), ((64,), 128, 1, 1, 7, 9),
]


@pytest.mark.parametrize("compute", ["fp32", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{'+'.join(map(str, c[0]))}to{c[1]}_k{c[2]}s{c[3]}_{c[4]}x{c[5]}")
def test_conv_plain(case, compute):
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    segs, cout, k, s, H, W = case
    torch.manual_seed(hash(case) & 0xFFFF)
    ops.set_precision(compute)
    try:
        m = Conv2d(sum(segs), cout, k, s, padding=k // 2)
        xs = [torch.randn(2, c, H, W) for c in segs]
        got = m.to(DEV).run([_act(x) for x in xs], out_dtype=torch.float32).nchw().cpu()
    finally:
        ops.set_precision("fp32")
    x, w = torch.cat(xs, 1), m.weight.detach().cpu()
    if compute == "f16":
        x, w = x.to(torch.float16).float(), w.to(torch.float16).float()
    want = F.conv2d(x, w, m.bias.detach().cpu(), stride=s, padding=k // 2)
    if compute == "f16":
        close_f16(f"conv {case} f16", got, want, out_f16=False)
    else:
        report_close(f"conv {case} fp32", got, want, 2e-4, 2e-4 * float(want.abs().max()))


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4, 5])
def test_conv_epilogue_activations(act):
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(act)
    m = Conv2d(24, 40, 3, 1, 1)
    x, r1, r2 = torch.randn(1, 24, 13, 37), torch.randn(1, 40, 13, 37), torch.randn(1, 40, 13, 37)
    gate = torch.rand(1, 40)
    got = m.to(DEV).run(_act(x), act=act, pre_act=2, slope=0.2, res1=_act(r1), gate=gate.to(DEV), res2=_act(r2)).nchw().cpu()
    want = _torch_act(F.conv2d(F.leaky_relu(x, 0.2), m.weight.cpu(), m.bias.cpu(), padding=1), act, 0.2) + gate[:, :, None, None] * r1 + r2
    report_close(f"epilogue act={act}", got, want, 2e-4, 2e-4)


def test_conv_slices_shuffle_and_f16_storage():
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d, Deconv5x5s2, _SubpelConv
    torch.manual_seed(5)
    # input read from / output written into channel slices of wider buffers
    big_in = torch.randn(1, 40, 9, 33)
    a = _act(big_in)
    m = Conv2d(16, 20, 3, 1, 1).to(DEV)
    out_buf = ops.new_act(1, 9, 33, 50, torch.float32, DEV)
    out_buf.buf.fill_(7.0)
    m.run(a.ch(8, 16), out_buf.ch(10, 20))
    full = out_buf.nchw().cpu()
    want = F.conv2d(big_in[:, 8:24], m.weight.cpu(), m.bias.cpu(), padding=1)
    report_close("slice conv", full[:, 10:30], want, 2e-4, 2e-4)
    assert bool((full[:, :10] == 7).all()) and bool((full[:, 30:] == 7).all()), "conv wrote outside its channel slice"
    # sub-pixel conv (PixelShuffle folded into the store) + GELU
    sp = _SubpelConv(12, 6, 2, 3).to(DEV)
    x = torch.randn(2, 12, 7, 11)
    got = sp.run(_act(x), act=3).nchw().cpu()
    want = F.gelu(F.pixel_shuffle(F.conv2d(x, sp[0].weight.cpu(), sp[0].bias.cpu(), padding=1), 2))
    report_close("subpel", got, want, 2e-4, 2e-4)
    # transposed conv as 4-phase conv
    d = Deconv5x5s2(10, 6).to(DEV)
    x = torch.randn(1, 10, 6, 9)
    want = F.conv_transpose2d(x, d.weight.cpu(), d.bias.cpu(), stride=2, padding=2, output_padding=1)
    report_close("deconv", d.run(_act(x)).nchw().cpu(), want, 2e-4, 2e-4)
    # f16 activation storage in and out
    x = torch.randn(1, 32, 10, 34)
    mm = Conv2d(32, 32, 3, 1, 1).to(DEV)
    res = torch.randn(1, 32, 10, 34)
    got = mm.run(_act(x, torch.float16), res1=_act(res, torch.float16), out_dtype=torch.float16)
    assert got.dtype == torch.float16
    want = F.conv2d(x.to(torch.float16).float(), mm.weight.cpu(), mm.bias.cpu(), padding=1) + res.to(torch.float16).float()
    report_close("f16 storage in fp32 compute mode (one f16 rounding of the output)", got.nchw().cpu(), want, 6e-4, 2e-4)


@pytest.mark.parametrize("segs,cout,k,H,W,B,act,res_dtype,out_dtype", [
    ((64,), 3, 3, 250, 1100, 1, 0, None, torch.float32), ((64,), 3, 3, 264, 1000, 1, 4, None, torch.float16), ((32,), 3, 3, 256, 1056, 1, 0, None, torch.float32),
    ((64,), 2, 3, 250, 1090, 1, 0, torch.float32, torch.float32), ((64, 64), 3, 3, 249, 1033, 1, 1, torch.float16, torch.float16),
    ((16,), 2, 7, 250, 1100, 1, 0, torch.float32, torch.float32), ((64,), 4, 3, 136, 250, 2, 2, None, torch.float16), ((16,), 2, 7, 130, 260, 2, 0, None, torch.float32)])
def test_narrow_head_dot_product_kernel(segs, cout, k, H, W, B, act, res_dtype, out_dtype):
    """Cout <= 4 heads of large f16 images run as v_dot2 on the vector ALU with scalar-loaded weights (csrc/conv_head.hip): ragged tiles,
    zero padding, two input segments, every activation, residual and output of either type -- against torch on the f16-rounded operands,
    and against the MFMA kernels ("head" = 0) within the f16 summation-order tolerance."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + k)
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, k, 1, padding=k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        res = torch.randn(B, cout, H, W) if res_dtype is not None else None
        kw = dict(act=act, slope=0.1, out_dtype=out_dtype)
        if res is not None:
            kw["res1"] = _act(res, res_dtype)
        ins = [_act(x, torch.float16) for x in xs]
        n0 = ops.get_counter("conv_head")
        prev = ops.set_option("head", 2)          # 2: also the shapes the dispatcher leaves to the MFMA kernels by default (64 -> 3, 64+64 -> 3)
        prev_tap = ops.set_option("headtap", 0)   # (round 5: the tap-GEMM kernel takes the 3x3 heads with Cout <= 3 first -- not the kernel under test here)
        try:
            got = m.run(ins, **kw)
            assert ops.get_counter("conv_head") == n0 + 1, "the narrow-head kernel did not take this launch"
            assert ops.set_option("head", 0) == 2
            other = m.run(ins, **kw)
        finally:
            ops.set_option("head", prev)
            ops.set_option("headtap", prev_tap)
        assert ops.get_counter("conv_head") == n0 + 1
        got, other = got.nchw().float().cpu(), other.nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    x = torch.cat(xs, 1).to(torch.float16).float()
    w = m.weight.detach().cpu().to(torch.float16).float()
    want = _torch_act(F.conv2d(x, w, m.bias.detach().cpu(), padding=k // 2), act, 0.1)
    if res is not None:
        want = want + (res.to(torch.float16).float() if res_dtype == torch.float16 else res)
    close_f16(f"head {segs}->{cout} k{k}", got, want, out_f16=out_dtype == torch.float16)
    close_f16(f"head {segs}->{cout} k{k} vs MFMA kernels", got, other, out_f16=out_dtype == torch.float16, both_f16=out_dtype == torch.float16)


@pytest.mark.parametrize("segs,cout,H,W,B,act,res_dtype,out_dtype", [
    ((64,), 3, 250, 1100, 1, 0, None, torch.float32), ((64,), 3, 264, 1000, 1, 4, None, torch.float16), ((32,), 3, 256, 1056, 1, 2, None, torch.float32),
    ((64,), 2, 250, 1090, 1, 0, torch.float32, torch.float32), ((64, 64), 3, 249, 1033, 1, 1, torch.float16, torch.float16), ((32, 32), 2, 130, 520, 1, 0, None, torch.float16),
    ((64,), 1, 136, 250, 2, 3, torch.float16, torch.float32), ((64,), 3, 8, 32 * 256, 1, 0, None, torch.float32), ((32,), 3, 67, 300, 3, 5, torch.float32, torch.float16)])
def test_narrow_head_tap_gemm_kernel(segs, cout, H, W, B, act, res_dtype, out_dtype):
    """3x3 heads with Cout <= 3 of large f16 images as a per-INPUT-pixel GEMM over (tap, output) columns on the matrix pipe plus a 9-point
    gather-sum of fp32 partial products in LDS (csrc/conv_headtap.hip): ragged tiles on both axes, zero padding on all four borders (a
    one-tile-high image included), two input segments, batches, every activation, residual and output of either type -- against torch on
    the f16-rounded operands, and against the kernels that took these launches before ("headtap" = 0) within the f16 summation-order tolerance."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + H)
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 3, 1, padding=1).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        res = torch.randn(B, cout, H, W) if res_dtype is not None else None
        kw = dict(act=act, slope=0.1, out_dtype=out_dtype)
        if res is not None:
            kw["res1"] = _act(res, res_dtype)
        ins = [_act(x, torch.float16) for x in xs]
        n0 = ops.get_counter("conv_headtap")
        prev = ops.set_option("headtap", 1)
        try:
            got = m.run(ins, **kw)
            assert ops.get_counter("conv_headtap") == n0 + 1, "the tap-GEMM head kernel did not take this launch"
            ops.set_option("headtap", 0)
            other = m.run(ins, **kw)
            assert ops.get_counter("conv_headtap") == n0 + 1
        finally:
            ops.set_option("headtap", prev)
        got, other = got.nchw().float().cpu(), other.nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    x = torch.cat(xs, 1).to(torch.float16).float()
    w = m.weight.detach().cpu().to(torch.float16).float()
    want = _torch_act(F.conv2d(x, w, m.bias.detach().cpu(), padding=1), act, 0.1)
    if res is not None:
        want = want + (res.to(torch.float16).float() if res_dtype == torch.float16 else res)
    close_f16(f"tap-GEMM head {segs}->{cout}", got, want, out_f16=out_dtype == torch.float16)
    close_f16(f"tap-GEMM head {segs}->{cout} vs the other kernels", got, other, out_f16=out_dtype == torch.float16, both_f16=out_dtype == torch.float16)


@pytest.mark.parametrize("split,cout,k,H,W,B,act", [
    ((8,), 32, 7, 70, 250, 1, 1), ((3, 3, 2), 32, 7, 136, 240, 1, 1), ((3, 3, 2), 32, 7, 33, 130, 3, 2), ((8,), 32, 7, 8, 32 * 70, 1, 0),
    ((3,), 64, 3, 120, 200, 1, 0), ((8,), 64, 3, 65, 129, 2, 1), ((8,), 32, 3, 128, 130, 1, 2), ((3, 2), 64, 3, 41, 400, 1, 0)])
def test_cin8_conv_kernel(split, cout, k, H, W, B, act):
    """f16 3x3 / 7x7 convs over ONE 8-channel input with the x-taps of a kernel row packed into the MFMA's K axis (csrc/conv_cin8.hip):
    ME_Spynet's first conv over cat(ref, warped, flow) and the 3x3 convs over a frame padded to 8 channels, both as the frame issues them
    (small fp32 inputs merged into one zero-padded f16 buffer) and over a plain 8-channel f16 tensor; ragged tiles on both axes, all four
    borders, batches -- against torch on the f16-rounded operands, and against the kernels that took these launches before ("cin8" = 0)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(split) + cout + k + H)
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(split), cout, k, 1, padding=k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in split]
        ins = [_act(x, torch.float16 if len(split) == 1 and split[0] == 8 else torch.float32) for x in xs]
        kw = dict(act=act, slope=0.1, out_dtype=torch.float16)
        n0 = ops.get_counter("conv_cin8")
        prev = ops.set_option("cin8", 1)
        try:
            got = m.run(ins, **kw)
            assert ops.get_counter("conv_cin8") == n0 + 1, "the 8-channel-input kernel did not take this launch"
            ops.set_option("cin8", 0)
            other = m.run(ins, **kw)
            assert ops.get_counter("conv_cin8") == n0 + 1
        finally:
            ops.set_option("cin8", prev)
        got, other = got.nchw().float().cpu(), other.nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    x = torch.cat(xs, 1).to(torch.float16).float()
    w = m.weight.detach().cpu().to(torch.float16).float()
    want = _torch_act(F.conv2d(x, w, m.bias.detach().cpu(), padding=k // 2), act, 0.1)
    close_f16(f"cin8 {split}->{cout} k{k}", got, want, out_f16=True)
    close_f16(f"cin8 {split}->{cout} k{k} vs the other kernels", got, other, out_f16=True, both_f16=True)


@pytest.mark.parametrize("cin,cout,H,W,B,act", [(64, 32, 150, 450, 1, 1), (32, 64, 136, 480, 1, 1), (32, 16, 272, 250, 1, 1), (32, 32, 130, 250, 2, 2),
                                                (64, 32, 16, 32 * 130, 1, 0), (32, 16, 67, 129, 7, 1)])
def test_conv7x7_row_reuse_kernel(cin, cout, H, W, B, act):
    """f16 7x7 convs 32 -> 16 / 32 / 64 and 64 -> 32 (ME_Spynet's inner convs, modules.py:127-131) with four output rows per wave, pixel
    fragments reused across the kernel rows in registers and weights loaded straight from the packed buffer (csrc/conv7x7_rr.hip): ragged
    tiles on both axes (a one-tile-high image included), all four borders, batches, two input chunks, two output passes, the 16-channel
    output's half rows -- against torch on the f16-rounded operands, and against the general kernel ("rr7" = 0)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(cin + cout + H)
    ops.set_precision("f16")
    try:
        m = Conv2d(cin, cout, 7, 1, padding=3).to(DEV)
        x = torch.randn(B, cin, H, W)
        kw = dict(act=act, slope=0.1, out_dtype=torch.float16)
        n0 = ops.get_counter("conv7x7_rr")
        prev = ops.set_option("rr7", 1)
        try:
            got = m.run([_act(x, torch.float16)], **kw)
            assert ops.get_counter("conv7x7_rr") == n0 + 1, "the row-reuse 7x7 kernel did not take this launch"
            ops.set_option("rr7", 0)
            other = m.run([_act(x, torch.float16)], **kw)
            assert ops.get_counter("conv7x7_rr") == n0 + 1
        finally:
            ops.set_option("rr7", prev)
        got, other = got.nchw().float().cpu(), other.nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w = m.weight.detach().cpu().to(torch.float16).float()
    want = _torch_act(F.conv2d(x.to(torch.float16).float(), w, m.bias.detach().cpu(), padding=3), act, 0.1)
    close_f16(f"7x7 row reuse {cin}->{cout}", got, want, out_f16=True)
    close_f16(f"7x7 row reuse {cin}->{cout} vs the general kernel", got, other, out_f16=True, both_f16=True)


@pytest.mark.parametrize("compute", ["fp32", "f16"])
@pytest.mark.parametrize("cin,cq,k,H,W,act,with_res", [(64, 64, 3, 13, 37, 0, False), (96, 96, 3, 9, 33, 2, True), (64, 32, 1, 17, 41, 0, False),
                                                       (128, 64, 1, 8, 32, 1, True), (32, 4, 3, 5, 9, 0, True), (64, 64, 3, 70, 130, 0, False)])
def test_subpel_conv_vector_stores(cin, cq, k, H, W, act, with_res, compute):
    """PixelShuffle(2) folded into the conv store with conv channels in multiples of 16 (image_model.py:25-27 subpel_conv3x3, modules.py:250-254
    subpel_conv1x1): four consecutive OUTPUT channels of one output pixel per lane (conv_common.h: epilogue_vec_shuffle) -- against
    torch's conv2d + pixel_shuffle, and bit for bit against the element-wise store path ("shuffle_vec" = 0)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import _SubpelConv
    torch.manual_seed(cin + cq + k)
    ops.set_precision(compute)
    try:
        sp = _SubpelConv(cin, cq, 2, k).to(DEV)
        x, res = torch.randn(2, cin, H, W), torch.randn(2, cq, 2 * H, 2 * W)
        adt = torch.float16 if compute == "f16" else torch.float32
        kw = dict(act=act, slope=0.1)
        if with_res:
            kw["res1"] = _act(res, adt)
        got = sp.run(_act(x, adt), out_dtype=adt, **kw)
        prev = ops.set_option("shuffle_vec", 0)
        try:
            elem = sp.run(_act(x, adt), out_dtype=adt, **kw)
        finally:
            ops.set_option("shuffle_vec", prev)
        assert prev == 1
        assert torch.equal(got.buf, elem.buf), "vector and element-wise subpel stores differ"
        got = got.nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w = sp[0].weight.detach().cpu()
    if compute == "f16":
        x, w, res = x.to(torch.float16).float(), w.to(torch.float16).float(), res.to(torch.float16).float()
    want = _torch_act(F.pixel_shuffle(F.conv2d(x, w, sp[0].bias.detach().cpu(), padding=k // 2), 2), act, 0.1)
    if with_res:
        want = want + res
    if compute == "f16":
        close_f16(f"subpel {cin}->{cq}x4 k{k} f16", got, want)
    else:
        report_close(f"subpel {cin}->{cq}x4 k{k} fp32", got, want, 2e-4, 2e-4 * float(want.abs().max()))


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 250, 1100), (64, 32, 256, 1024), (64, 3, 264, 1000), (32, 64, 250, 1090), (32, 3, 256, 1056), (3, 64, 250, 1100)])
def test_conv3x3_weight_stationary_kernel(cin, cout, H, W):
    """Large-image f16 3x3 convs take the persistent weight-stationary kernel (csrc/conv3x3_ws.hip):
    ragged right/bottom tiles, the fused pre-activation, activation, both residuals and f16 stores."""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(cin * 100 + cout)
    ops.set_precision("f16")
    try:
        m = Conv2d(cin, cout, 3, 1, 1).to(DEV)
        x, r1, r2 = torch.randn(1, cin, H, W), torch.randn(1, cout, H, W), torch.randn(1, cout, H, W)
        bf = torch.float16
        plain = m.run(_act(x, bf), out_dtype=torch.float32).nchw().cpu()
        fused = m.run(_act(x, bf), pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01, res1=_act(r1, bf), res2=_act(r2, bf),
                      out_dtype=bf).nchw().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = x.to(bf).float()
    want = F.conv2d(xb, w, b, padding=1)
    close_f16("rs plain", plain, want, out_f16=False)
    pre = F.leaky_relu(xb, 0.01).to(bf).float()
    want2 = F.leaky_relu(F.conv2d(pre, w, b, padding=1), 0.01) + r1.to(bf).float() + r2.to(bf).float()
    close_f16("rs fused", fused, want2)


@pytest.mark.parametrize("segs,cout,B,H,W,pre,act,nres", [
    ((64,), 64, 1, 256, 1024, 0, 0, 0),      # exact tiles, plain
    ((64,), 64, 1, 250, 1100, 2, 2, 1),      # ragged bottom/right tiles, LeakyReLU both sides, one residual
    ((64,), 64, 1, 251, 1043, 1, 1, 2),      # odd sizes, ReLU both sides, two residuals
    ((64,), 64, 1, 264, 1000, 0, 0, 1),      # ResBlock tail: residual only
    ((32, 32), 64, 1, 250, 1090, 0, 2, 0),   # two 32-channel segments (virtual cat)
    ((64,), 32, 1, 256, 1056, 0, 2, 1),      # 32 output channels
    ((64,), 64, 2, 130, 1000, 2, 0, 1),      # batch of two
    ((8,), 64, 1, 256, 1030, 0, 0, 0),       # single short chunk (first-layer convs): 8 / 16 / 32 input channels
    ((16,), 64, 1, 250, 1100, 0, 2, 0),
    ((32,), 32, 1, 264, 1000, 0, 2, 0),
    ((32,), 64, 1, 256, 1024, 0, 0, 0),
])
def test_conv3x3_pingpong_kernel(segs, cout, B, H, W, pre, act, nres):
    """The 64-channel f16 ResBlock family takes the two-group phase-skewed kernel (csrc/conv3x3_pp.hip): every
    fused variant it accepts against torch on the same f16-rounded operands, plus agreement with the
    weight-stationary kernel it replaces (DSVC_NO_PP is read once per process, so that leg uses an fp32-output
    launch, which the ping-pong kernel does not take)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + H)
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 3, 1, 1).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        rs = [torch.randn(B, cout, H, W) for _ in range(nres)]
        kw = dict(pre_act=pre, act=act, slope=0.05)
        if nres >= 1:
            kw["res1"] = _act(rs[0], bf)
        if nres >= 2:
            kw["res2"] = _act(rs[1], bf)
        ins = [_act(x, bf) for x in xs]
        got = m.run(ins if len(ins) > 1 else ins[0], out_dtype=bf, **kw).nchw().cpu()
        other = m.run(ins if len(ins) > 1 else ins[0], out_dtype=torch.float32, **kw).nchw().cpu()   # WS / generic kernel
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([x.to(bf).float() for x in xs], 1)
    xin = _torch_act(xb, pre, 0.05).to(bf).float() if pre else xb
    want = _torch_act(F.conv2d(xin, w, b, padding=1), act, 0.05)
    for r in rs:
        want = want + r.to(bf).float()
    assert got.shape == want.shape
    close_f16("pp (f16 out) vs torch", got, want)
    close_f16("ws / generic (fp32 out) vs torch", other, want, out_f16=False)
    close_f16("pp f16-out vs the fp32-out launch of the same conv: one correct rounding", got, other)


@pytest.mark.parametrize("cin,cout,H,W,act,res,out_dt", [
    (64, 3, 256, 1024, 4, None, "f32"),      # sigmoid mask heads
    (64, 3, 250, 1100, 0, None, "f32"),      # frame heads
    (64, 2, 251, 1043, 0, "f32", "f32"),     # flow head + fp32 residual
    (32, 3, 264, 1000, 0, None, "f32"),      # single-chunk input
    (64, 8, 256, 1056, 1, "f16", "f16"),   # 8 channels, f16 out + f16 residual
])
def test_conv3x3_pingpong_narrow_head(cin, cout, H, W, act, res, out_dt):
    """Narrow output heads (Cout <= 8) on the ping-pong kernel: one N-tile, element-wise fp32/f16 stores."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(cin + cout + W)
    bf = torch.float16
    dts = {"f32": torch.float32, "f16": bf}
    ops.set_precision("f16")
    try:
        m = Conv2d(cin, cout, 3, 1, 1).to(DEV)
        x = torch.randn(1, cin, H, W)
        r = torch.randn(1, cout, H, W) if res else None
        kw = dict(act=act, slope=0.05)
        if res:
            kw["res1"] = _act(r, dts[res])
        got = m.run(_act(x, bf), out_dtype=dts[out_dt], **kw).nchw().float().cpu()
        was = ops.set_option("pp", 0)
        try:
            other = m.run(_act(x, bf), out_dtype=dts[out_dt], **kw).nchw().float().cpu()     # weight-stationary kernel
        finally:
            ops.set_option("pp", was)
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    want = _torch_act(F.conv2d(x.to(bf).float(), w, b, padding=1), act, 0.05)
    if res:
        want = want + (r.to(bf).float() if res == "f16" else r)
    rt = 2e-2 if out_dt == "f16" else 5e-3
    close_f16("head vs torch", got, want, out_f16=(out_dt == "f16"))
    close_f16("head vs ws", got, other, out_f16=(out_dt == "f16"), both_f16=True)


@pytest.mark.parametrize("segs,seg_dt,cout,B,H,W,act,res,out_dt", [
    ((96, 48), ("f32", "f32"), 96, 1, 68, 120, 3, None, "f16"),          # cc_mean/cc_scale first layer (GELU)
    ((96,), ("f16",), 64, 1, 68, 120, 3, None, "f16"),                  # hidden layers
    ((64,), ("f16",), 32, 1, 17, 30, 3, None, "f16"),
    ((32,), ("f16",), 12, 1, 68, 120, 0, None, "f32"),                   # last layer: 12 channels, fp32 out
    ((96, 48, 16), ("f32", "f32", "f16"), 96, 1, 34, 60, 3, None, "f16"),   # three mixed segments, short chunks
    ((96, 8), ("f32", "f32"), 48, 2, 9, 33, 5, "f32", "f32"),             # lrp tail: 0.5*tanh + fp32 residual, batch 2
    ((64,), ("f16",), 224, 1, 5, 7, 1, "f16", "f16"),                  # 7 N-tiles, tiny image, f16 residual
    ((64, 32, 8), ("f32", "f32", "f32"), 8, 1, 68, 120, 4, None, "f32"),  # sigmoid, 8 channels
])
def test_conv3x3_small_image_scan_net_shapes(segs, seg_dt, cout, B, H, W, act, res, out_dt):
    """Small-image f16 3x3 convs in the shapes of the slice-scan nets: mixed fp32/f16 segments (fp32 latents converted while
    staged), short channel chunks, ragged row ends, every activation, residual and output type -- against torch on
    f16-rounded operands."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) * 7 + cout + W)
    bf = torch.float16
    dts = {"f32": torch.float32, "f16": bf}
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 3, 1, 1).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        ins = [_act(x, dts[t]) for x, t in zip(xs, seg_dt)]
        r = torch.randn(B, cout, H, W) if res else None
        kw = dict(act=act, slope=0.05)
        if res:
            kw["res1"] = _act(r, dts[res])
        got = m.run(ins if len(ins) > 1 else ins[0], out_dtype=dts[out_dt], **kw).nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([x.to(bf).float() for x in xs], 1)
    want = _torch_act(F.conv2d(xb, w, b, padding=1), act, 0.05)
    if res:
        want = want + (r.to(bf).float() if res == "f16" else r)
    tol = (2e-2 if out_dt == "f16" else 5e-3) * float(want.abs().max())
    close_f16("small conv vs torch", got, want, out_f16=(out_dt == "f16"))


@pytest.mark.parametrize("segs,cout,k,stride,B,H,W,pre,act,res", [
    ((96,), 96, 3, 1, 1, 200, 330, 0, 2, True),          # three N-tiles, LeakyReLU, f16 residual, ragged right edge (330 = 10*32 + 10)
    ((64, 64), 64, 3, 1, 2, 100, 200, 0, 0, False),      # two segments, batch 2, no activation
    ((128,), 128, 3, 1, 1, 162, 320, 1, 1, True),        # four N-tiles (k-step pipeline), ReLU in and out, ragged bottom edge
    ((32,), 64, 7, 1, 1, 120, 250, 0, 1, False),         # 7x7: tap groups of one kernel row
    ((96,), 64, 5, 2, 1, 400, 512, 0, 0, False),         # stride 2
    ((64,), 256, 3, 1, 1, 160, 320, 0, 2, False),        # two groups of four N-tiles
    ((64, 32), 64, 3, 1, 1, 130, 300, 2, 2, True),       # 64 output channels, three chunks over two segments: patch prefetch, LeakyReLU on the way in
    ((32,), 64, 3, 1, 2, 90, 200, 0, 1, False),          # ... a single chunk (nothing to prefetch), batch 2
])
def test_conv_register_epilogue_matches_slab_epilogue(segs, cout, k, stride, B, H, W, pre, act, res):
    """The generic kernel's LDS-free epilogue (MFMA operands swapped, one 32-lane swap, two 16-byte stores per lane and
    32-channel tile) against the slab epilogue of the same kernel: same accumulation order, same arithmetic -> equal results;
    and against torch on f16-rounded operands."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + W)
    bf = torch.float16
    ops.set_precision("f16")
    was_s, was_t = ops.set_option("specialised", 0), ops.set_option("tall_min_tiles", 1 if sum(segs) == 128 and cout == 64 else 1 << 30)
    try:
        m = Conv2d(sum(segs), cout, k, stride, k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        ins = [_act(x, bf) for x in xs]
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        r = torch.randn(B, cout, Ho, Wo) if res else None
        kw = dict(pre_act=pre, act=act, slope=0.05, out_dtype=bf)
        if res:
            kw["res1"] = _act(r, bf)
        inp = ins if len(ins) > 1 else ins[0]
        ops.set_option("reg_epilogue", 1)
        reg = m.run(inp, **kw).nchw().float().cpu()
        was_p = ops.set_option("patch_prefetch", 0)          # the same variant without the input-patch prefetch
        reg_nopf = m.run(inp, **kw).nchw().float().cpu()
        ops.set_option("patch_prefetch", was_p)
        assert torch.equal(reg, reg_nopf), "patch prefetch changes the result"
        ops.set_option("reg_epilogue", 0)
        slab = m.run(inp, **kw).nchw().float().cpu()
    finally:
        ops.set_option("reg_epilogue", 1)
        ops.set_option("specialised", was_s); ops.set_option("tall_min_tiles", was_t)
        ops.set_precision("fp32")
    assert torch.equal(reg, slab), f"register and slab epilogues differ: max |d| = {float((reg - slab).abs().max()):.3e} at {int((reg != slab).sum())} elements"
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([_torch_act(x.to(bf).float(), pre, 0.05).to(bf).float() for x in xs], 1)
    want = _torch_act(F.conv2d(xb, w, b, stride=stride, padding=k // 2), act, 0.05)
    if res:
        want = want + r.to(bf).float()
    close_f16("register epilogue vs torch", reg, want)


@pytest.mark.parametrize("segs,cout,H,W,pre,act,res", [
    ((64, 64), 64, 256, 1024, 0, 0, False),        # fea_embd[0] (modules.py:401): four chunks, no activation, exactly 1024 tiles
    ((64, 64), 64, 250, 1000, 0, 2, False),        # UNet.up_conv3's first conv (modules.py:262-277, LeakyReLU): ragged right / bottom tiles
    ((64, 32), 64, 256, 1024, 0, 0, False),        # Reconstruction.first_conv (modules.py:322): three chunks (odd: the weight-buffer parity wraps)
    ((64, 8), 64, 264, 1056, 0, 0, False),         # RefineNet.refine[0] (modules.py:307): an 8-channel packed segment = a chunk with one valid quarter
    ((64, 64, 32), 64, 256, 1024, 0, 1, False),    # three segments, five chunks, ReLU
    # ---- output-channel split (two passes of 64 inside one launch), pre-activation, one residual: the extended instantiation ----
    ((128,), 128, 272, 480, 1, 1, False),          # UNet.context_refine ResBlock1.conv1 (modules.py:373: ReLU before and after) @1/4 resolution: 2 x 510 tile jobs
    ((128,), 128, 272, 480, 0, 0, True),           # ... conv2: + x
    ((96,), 96, 256, 512, 0, 2, False),            # residual codec's ResBottleneck conv2 (image_model.py:22-38) @1/2: second pass = 32 real + 32 padding columns
    ((64,), 128, 256, 1024, 0, 2, False),          # UNet.conv3's first conv 64 -> 128 (two chunks, two passes)
    ((64, 64), 64, 256, 1024, 2, 2, True),         # pre-activation + activation + residual on a 64-channel output
])
def test_streamed_weights_pp_kernel_matches_general_kernel(segs, cout, H, W, pre, act, res):
    """csrc/conv3x3_pps.hip (two wave groups, weights streamed per 32-channel chunk through two LDS buffers) against the general kernel
    on the same operands: the same MFMAs in the same order per output element, bias after the accumulation -> EQUAL f16 results; and
    against torch on f16-rounded operands at half an ulp.  The launch counter proves which kernel ran."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + W + act + cout + 5 * pre)
    bf = torch.float16
    ops.set_precision("f16")
    was, was_pp = ops.set_option("pps", 2), ops.set_option("pp", 0)        # (pps 2: also the split shapes that are off by default; pp off: keeps two-chunk 64-output shapes away from the resident-weights kernel)
    try:
        m = Conv2d(sum(segs), cout, 3, 1, 1).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        r = torch.randn(1, cout, H, W) if res else None
        ins = [_act(x, bf) for x in xs]
        kw = dict(pre_act=pre, act=act, slope=0.05, out_dtype=bf)
        if res:
            kw["res1"] = _act(r, bf)
        n0 = ops.get_counter("conv3x3_pps")
        got = m.run(ins, **kw).nchw().float().cpu()
        again = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv3x3_pps") == n0 + 2, "the streamed-weights kernel did not take this shape"
        ops.set_option("pps", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv3x3_pps") == n0 + 2
    finally:
        ops.set_option("pps", was); ops.set_option("pp", was_pp)
        ops.set_precision("fp32")
    assert torch.equal(got, again), "not deterministic"
    assert torch.equal(got, ref), f"streamed-weights and general kernel differ: max |d| = {float((got - ref).abs().max()):.3e} at {int((got != ref).sum())} elements"
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([_torch_act(x.to(bf).float(), pre, 0.05).to(bf).float() for x in xs], 1)
    want = _torch_act(F.conv2d(xb, w, b, padding=1), act, 0.05)
    if res:
        want = want + r.to(bf).float()
    close_f16("streamed-weights kernel vs torch", got, want)


@pytest.mark.parametrize("segs,H,W,pre,act,nres", [
    ((64,), 256, 1024, 0, 0, 0),          # plain 64 -> 64 (four 16-channel chunks)
    ((64,), 250, 1000, 1, 1, 0),          # ResBlock1.conv1 of the ReLU blocks (modules.py:200-204): pre-activation + activation, ragged tiles
    ((64,), 256, 1024, 2, 2, 0),          # ... of the LeakyReLU blocks
    ((64,), 256, 1024, 0, 0, 1),          # ResBlock1.conv2: + x
    ((64,), 264, 1056, 0, 0, 2),          # FeatureExtraction's last block (modules.py:361-374): + x + conv1(x)
    ((64, 64), 256, 1024, 0, 2, 0),       # two segments, eight chunks, LeakyReLU
    ((64, 4), 256, 1024, 0, 0, 1),        # a 4-channel segment = one valid quarter of a chunk (fp32 mode takes 3-channel frames unpacked: C % 4 != 0 stays general)
    ((32,), 256, 1024, 0, 2, 0),          # two chunks (UNet.conv2's first conv 32 -> 64): the weight-buffer parity alternates every step
    ((16, 16), 250, 1000, 0, 0, 1),       # two one-chunk segments, ragged tiles, + residual
])
def test_streamed_weights_pp_kernel_fp32_is_bit_identical_to_the_general_kernel(segs, H, W, pre, act, nres):
    """The fp32 instantiation of csrc/conv3x3_pps.hip -- the parity mode's arithmetic: v_mfma_f32_32x32x2_f32 over 16-channel chunks,
    pre-activation / activation / up to two fp32 residuals as run-time parameters -- against the general kernel: EQUAL bits (same MFMA order
    per output element, + bias, activation, + (res1 + res2)), so the byte-identity of fp32 mode with the reference is untouched; and
    against torch fp32."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + W + 7 * pre + act + nres)
    f32 = torch.float32
    ops.set_precision("fp32")
    was = ops.set_option("pps_f32", 2)                  # 2: also the two-residual variant (off by default: not faster)
    try:
        m = Conv2d(sum(segs), 64, 3, 1, 1).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        rs = [torch.randn(1, 64, H, W) for _ in range(nres)]
        ins = [_act(x, f32) for x in xs]
        kw = dict(pre_act=pre, act=act, slope=0.05, out_dtype=f32)
        if nres >= 1:
            kw["res1"] = _act(rs[0], f32)
        if nres == 2:
            kw["res2"] = _act(rs[1], f32)
        n0 = ops.get_counter("conv3x3_pps")
        got = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv3x3_pps") == n0 + 1, "the streamed-weights kernel did not take this fp32 shape"
        ops.set_option("pps_f32", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv3x3_pps") == n0 + 1
    finally:
        ops.set_option("pps_f32", was)
    assert torch.equal(got, ref), f"fp32: streamed-weights and general kernel differ: max |d| = {float((got - ref).abs().max()):.3e} at {int((got != ref).sum())} elements"
    xb = torch.cat([_torch_act(x, pre, 0.05) for x in xs], 1)
    want = _torch_act(F.conv2d(xb, m.weight.detach().cpu(), m.bias.detach().cpu(), padding=1), act, 0.05)
    for r in rs:
        want = want + r
    report_close("fp32 streamed-weights kernel vs torch", got, want, 1e-5, 3e-5 * float(want.abs().max()))


@pytest.mark.parametrize("segs,cout,k,stride,H,W,act", [
    ((3,), 64, 3, 1, 130, 250, 0),            # FeatureExtraction's first conv on a frame (modules.py:364)
    ((2, 3), 64, 3, 1, 130, 250, 0),          # RefineNet(mv) first conv on cat[mv, ref] (modules.py:307)
    ((3, 2, 3), 64, 5, 2, 128, 256, 0),       # mv codec g_a[0] on cat[cur, mv, ref] (image_model.py:53-66), stride 2
    ((3, 3, 2), 32, 7, 1, 68, 130, 1),        # MEBasic's first conv on cat[im1, warped, flow] (modules.py:127)
    ((64, 3), 64, 3, 1, 130, 250, 0),         # a full segment beside a narrow one: only the narrow chunk skips; the full one is staged with 16-byte requests
    ((3, 64, 3), 96, 5, 2, 128, 256, 0),      # the residual codec's g_a[0] on cat[cur, feature, predict] (image_model.py:53-66)
    ((5,), 64, 3, 1, 40, 70, 2),              # five channels: second k step partly valid
])
def test_fp32_zero_k_slots_skipped_bit_identically(segs, cout, k, stride, H, W, act):
    """fp32 launches with 2- / 3-channel segments (frames, flows: element-staged, C % 4 != 0) skip the MFMAs whose two k slots hold only
    channel padding ("skip_zero_k"): an fma step with zero operands leaves the accumulator as it is -> EQUAL bits with the option off."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + k)
    f32 = torch.float32
    ops.set_precision("fp32")
    was = ops.set_option("skip_zero_k", 1)
    try:
        m = Conv2d(sum(segs), cout, k, stride, k // 2).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        ins = [_act(x, f32) for x in xs]
        kw = dict(act=act, slope=0.01, out_dtype=f32)
        got = m.run(ins, **kw).nchw().float().cpu()
        ops.set_option("skip_zero_k", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
    finally:
        ops.set_option("skip_zero_k", was)
    nd = int((got.view(torch.int32) != ref.view(torch.int32)).sum())
    assert nd == 0, f"skip_zero_k on / off differ at {nd} of {got.numel()} elements"
    want = _torch_act(F.conv2d(torch.cat(xs, 1), m.weight.detach().cpu(), m.bias.detach().cpu(), stride=stride, padding=k // 2), act, 0.01)
    report_close("fp32 narrow-segment conv vs torch", got, want, 1e-5, 3e-5 * float(want.abs().max()))


@pytest.mark.parametrize("segs,cout,k,H,W,B,act,nres", [
    ((64,), 3, 3, 250, 1100, 1, 4, 0),        # Reconstruction's sigmoid mask head (modules.py:322-340): ragged tiles
    ((64,), 3, 3, 256, 1024, 1, 0, 0),        # ... its image heads
    ((64,), 2, 3, 250, 1090, 1, 0, 1),        # RefineNet(mv) last conv + x (modules.py:307-315)
    ((64, 64), 3, 3, 249, 1033, 1, 0, 0),     # r3 = conv(cat[...]) (modules.py:340): two segments, eight chunks
    ((32,), 3, 3, 256, 1056, 1, 0, 2),        # MC.out_conv shape with both residual operands
    ((16,), 2, 7, 250, 1100, 1, 0, 1),        # MEBasic's last conv + flow (modules.py:127-139)
    ((64,), 4, 3, 136, 250, 2, 2, 0),         # batch 2, four outputs, LeakyReLU
    ((16,), 1, 7, 130, 260, 2, 5, 0),         # one output, 0.5 tanh
])
def test_fp32_head_kernel_is_bit_identical(segs, cout, k, H, W, B, act, nres):
    """csrc/conv_head_f32.hip: fp32 mode's Cout <= 4 heads as v_fma_f32 chains in the MFMA's own order (v_mfma_f32_32x32x2_f32 is an fma
    chain, k slot 0 first: tools/mfma_f32_order_probe.py) -- EQUAL bits with the general kernel, zero signs included, for every
    activation / residual combination; and against torch fp32."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + k + W)
    f32 = torch.float32
    ops.set_precision("fp32")
    was = ops.set_option("head_f32", 1)
    try:
        m = Conv2d(sum(segs), cout, k, 1, padding=k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        rs = [torch.randn(B, cout, H, W) for _ in range(nres)]
        kw = dict(act=act, slope=0.1, out_dtype=f32)
        if nres >= 1:
            kw["res1"] = _act(rs[0], f32)
        if nres == 2:
            kw["res2"] = _act(rs[1], f32)
        ins = [_act(x, f32) for x in xs]
        n0 = ops.get_counter("conv_head")
        got = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv_head") == n0 + 1, "the fp32 head kernel did not take this launch"
        ops.set_option("head_f32", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv_head") == n0 + 1
    finally:
        ops.set_option("head_f32", was)
    nd = int((got.view(torch.int32) != ref.view(torch.int32)).sum())
    assert nd == 0, f"fp32 head: vector-ALU chain and MFMA kernel differ at {nd} of {got.numel()} elements, max |d| = {float((got - ref).abs().max()):.3e}"
    want = _torch_act(F.conv2d(torch.cat(xs, 1), m.weight.detach().cpu(), m.bias.detach().cpu(), padding=k // 2), act, 0.1)
    for r in rs:
        want = want + r
    report_close("fp32 head kernel vs torch", got, want, 1e-5, 3e-5 * float(want.abs().max()))


@pytest.mark.parametrize("segs,cout,H,W,pre,act,res", [
    ((96,), 48, 68, 120, 0, 3, False),        # cc_mean / cc_scale second conv of the residual codec (image_model.py:114-146): GELU, 4x32 tiles x 2 N-tiles
    ((96, 48), 96, 68, 120, 0, 3, False),     # ... first conv of slice 4: two segments, nine chunks, three N-tiles
    ((64,), 32, 68, 120, 0, 3, False),        # one N-tile
    ((64,), 64, 136, 240, 2, 2, True),        # 1/8-resolution ResBottleneck conv2 shape class with pre-activation and residual
    ((48,), 12, 67, 119, 0, 5, True),         # lrp tail: 0.5 tanh, + residual, ragged tiles
])
def test_fp32_small_image_patch_prefetch_is_bit_identical(segs, cout, H, W, pre, act, res):
    """fp32 mode's small launches (entropy-parameter nets, 1/8 - 1/16 resolution layers) take the general kernel's patch-prefetch variants
    ("small_pf": the next 16-channel chunk's halo patch rides under the current chunk's MFMAs): staging only -- EQUAL bits with the option off."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + W)
    f32 = torch.float32
    ops.set_precision("fp32")
    was = ops.set_option("small_pf", 1)
    try:
        m = Conv2d(sum(segs), cout, 3, 1, 1).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        r = torch.randn(1, cout, H, W) if res else None
        kw = dict(pre_act=pre, act=act, slope=0.01, out_dtype=f32)
        if res:
            kw["res1"] = _act(r, f32)
        ins = [_act(x, f32) for x in xs]
        got = m.run(ins, **kw).nchw().float().cpu()
        ops.set_option("small_pf", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
    finally:
        ops.set_option("small_pf", was)
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), f"fp32 small_pf on / off differ at {int((got != ref).sum())} elements"
    xb = torch.cat([_torch_act(x, pre, 0.01) for x in xs], 1)
    want = _torch_act(F.conv2d(xb, m.weight.detach().cpu(), m.bias.detach().cpu(), padding=1), act, 0.01)
    if res:
        want = want + r
    report_close("fp32 small-image conv vs torch", got, want, 1e-5, 3e-5 * float(want.abs().max()))


@pytest.mark.parametrize("cin,cout,B,H,W,act,res", [
    (64, 64, 1, 137, 241, 0, True),       # ResBottleneckBlock.conv3 + x (modules.py:176-189), ragged last unit (33 017 pixels)
    (64, 64, 1, 136, 240, 2, False),      # ... conv1 + LeakyReLU, whole units
    (96, 96, 1, 272, 480, 0, True),       # the residual codec's bottlenecks (image_model.py:22-38) at 1/4 resolution
    (96, 96, 2, 136, 240, 2, False),      # batch 2: one flat pixel list
    (128, 128, 1, 136, 250, 1, True),     # widest: 82 KB of weights + 66 KB of tiles in LDS
    (64, 128, 1, 136, 250, 1, False),
    (128, 64, 1, 136, 250, 0, False),
    (64, 96, 1, 136, 250, 2, True),
    ((32, 32), 64, 1, 136, 250, 0, "gate"),   # ConvBlockResidual tails (modules.py:233-247): up_dim(cat[a, b]) + gate[c] * branch
    ((64, 64), 64, 1, 137, 241, 0, "gate"),
    (64, 32, 1, 136, 250, 0, "gate"),         # one output tile
    ((64, 32), 64, 1, 136, 250, 2, True),     # two segments of different widths, no gate
])
def test_fp32_pointwise_kernel_matches_general_kernel(cin, cout, B, H, W, act, res):
    """csrc/conv1x1_f32.hip (fp32 mode's streaming 1x1 kernel: whole-row loads / stores, operand roles swapped) against the general
    kernel: EQUAL bits -- same v_mfma_f32_32x32x2_f32 channel pairs in the same order, bias / activation / residual in the general
    epilogue's order -- so fp32 mode's byte parity with the reference cannot move; and against torch fp32.  The counter proves which ran."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    segs = cin if isinstance(cin, tuple) else (cin,)
    torch.manual_seed(sum(segs) + 3 * cout + W + act)
    f32 = torch.float32
    ops.set_precision("fp32")
    was = ops.set_option("conv1x1_f32", 1)
    try:
        m = Conv2d(sum(segs), cout, 1, 1, 0).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        x = torch.cat(xs, 1)
        r = torch.randn(B, cout, H, W) if res else None
        gate = torch.rand(1, cout) if res == "gate" else None
        kw = dict(act=act, slope=0.01, out_dtype=f32)
        if res:
            kw["res1"] = _act(r, f32)
        if gate is not None:
            kw["gate"] = gate.to(DEV)
        ins = [_act(t, f32) for t in xs]
        n0 = ops.get_counter("conv1x1")
        got = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv1x1") == n0 + 1, "the fp32 streaming kernel did not take this shape"
        ops.set_option("conv1x1_f32", 0)
        ref = m.run(ins, **kw).nchw().float().cpu()
        assert ops.get_counter("conv1x1") == n0 + 1
    finally:
        ops.set_option("conv1x1_f32", was)
    assert torch.equal(got, ref), f"fp32 1x1: streaming and general kernel differ: max |d| = {float((got - ref).abs().max()):.3e} at {int((got != ref).sum())} elements"
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), "fp32 1x1: results differ in the sign of a zero"
    want = _torch_act(F.conv2d(x, m.weight.detach().cpu(), m.bias.detach().cpu()), act, 0.01)
    if res:
        want = want + (r if gate is None else gate[:, :, None, None] * r)
    report_close("fp32 streaming 1x1 kernel vs torch", got, want, 1e-5, 3e-5 * float(want.abs().max()))


@pytest.mark.parametrize("segs,cout,k,stride,H,W,act,res", [
    ((128,), 128, 3, 1, 160, 320, 1, True),        # UNet bottleneck ResBlock shape class: four N-tiles per workgroup
    ((96,), 128, 5, 2, 512, 512, 0, False),        # stride 2 (even/odd column LDS layout) with four N-tiles
    ((64,), 256, 3, 1, 160, 320, 3, False),        # two groups of four N-tiles
    ((96,), 96, 3, 1, 200, 320, 2, True),          # three N-tiles, tap groups of one kernel row, weight prefetch across groups
])
def test_conv_wide_output_variants(segs, cout, k, stride, H, W, act, res):
    """Images large enough that the generic kernel keeps 3-4 N-tiles (96-128 output channels) per workgroup: the variants
    with register-prefetched weights and (for 128 channels) k-step fragment pipelining -- against torch on f16-rounded operands."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + W)
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, k, stride, k // 2).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        ins = [_act(x, bf) for x in xs]
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        r = torch.randn(1, cout, Ho, Wo) if res else None
        kw = dict(act=act, slope=0.05, out_dtype=torch.float32)
        if res:
            kw["res1"] = _act(r, bf)
        got = m.run(ins[0], **kw).nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    want = _torch_act(F.conv2d(xs[0].to(bf).float(), w, b, stride=stride, padding=k // 2), act, 0.05)
    if res:
        want = want + r.to(bf).float()
    close_f16("wide conv vs torch", got, want, out_f16=False)


@pytest.mark.parametrize("segs,cout,k,B,H,W,pre,act,res", [
    ((64, 64), 64, 3, 1, 70, 100, 0, 2, False),          # two segments, ragged rows and columns (70 = 4*16 + 6)
    ((32, 32), 64, 3, 2, 33, 45, 2, 2, True),            # batch 2, LeakyReLU in and out, f16 residual
    ((32,), 64, 5, 1, 50, 64, 0, 1, False),              # 5x5: tap groups of one kernel row
    ((64, 8), 128, 3, 1, 48, 96, 0, 0, False),           # 128 output channels = two groups of 64
    ((64, 32), 64, 3, 2, 25, 33, 1, 0, False),           # three chunks over two segments, batch 2, ReLU on the way in
])
def test_conv_tall_tile_variant(segs, cout, k, B, H, W, pre, act, res):
    """The generic kernel's 16x32-pixel tile variant (f16, stride 1, 64 output channels per group; picked from
    `tall_min_tiles` tiles on) against the 8x32 tiling of the same kernel -- identical accumulation order per output,
    so bit-identical results -- and against torch on f16-rounded operands."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout + W)
    bf = torch.float16
    ops.set_precision("f16")
    was_s, was_t, was_p = ops.set_option("specialised", 0), ops.set_option("tall_min_tiles", 1), ops.set_option("patch_prefetch", 0)
    try:
        m = Conv2d(sum(segs), cout, k, 1, k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        ins = [_act(x, bf) for x in xs]
        r = torch.randn(B, cout, H, W) if res else None
        kw = dict(pre_act=pre, act=act, slope=0.05, out_dtype=torch.float32)
        if res:
            kw["res1"] = _act(r, bf)
        tall = m.run(ins if len(ins) > 1 else ins[0], **kw).nchw().float().cpu()
        ops.set_option("tall_min_tiles", 1 << 30)
        flat = m.run(ins if len(ins) > 1 else ins[0], **kw).nchw().float().cpu()
    finally:
        ops.set_option("specialised", was_s); ops.set_option("tall_min_tiles", was_t); ops.set_option("patch_prefetch", was_p)
        ops.set_precision("fp32")
    assert torch.equal(tall, flat), f"16-row and 8-row tilings differ: max |d| = {float((tall - flat).abs().max()):.3e}"
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([_torch_act(x.to(bf).float(), pre, 0.05).to(bf).float() for x in xs], 1)
    want = _torch_act(F.conv2d(xb, w, b, padding=k // 2), act, 0.05)
    if res:
        want = want + r.to(bf).float()
    close_f16("tall tile vs torch", tall, want, out_f16=False)


@pytest.mark.parametrize("segs,seg_dt,cout,B,H,W,pre,act", [
    ((96, 48), ("f32", "f32"), 96, 1, 68, 120, 0, 3),                 # cc_mean/cc_scale first layer
    ((96, 36), ("f32", "f32"), 96, 1, 68, 120, 0, 3),                 # 36 = 32 + a half-filled quarter (C % 8 == 4)
    ((64, 12, 8), ("f32", "f32", "f16"), 64, 2, 33, 70, 2, 2),       # mixed storage, short chunks, LeakyReLU on the way in
    ((4,), ("f32",), 32, 1, 40, 50, 1, 0),                            # a single half quarter
])
def test_conv_fp32_storage_converted_while_staged(segs, seg_dt, cout, B, H, W, pre, act):
    """f16 compute over fp32 storage (the slice scan's fp32 latents): the general kernel converts while staging --
    two batched 16-B loads per 8-channel quarter, RNE to f16 -- instead of the per-element path.  Against torch on
    f16-rounded operands, including segments whose channel count is 4 mod 8 and a fused leading activation."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) * 3 + cout + W)
    bf = torch.float16
    dts = {"f32": torch.float32, "f16": bf}
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 3, 1, 1).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        ins = [_act(x, dts[t]) for x, t in zip(xs, seg_dt)]
        got = m.run(ins if len(ins) > 1 else ins[0], out_dtype=torch.float32, pre_act=pre, act=act, slope=0.05).nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    xb = torch.cat([_torch_act(x.to(bf).float() if t == "f16" else x, pre, 0.05).to(bf).float() for x, t in zip(xs, seg_dt)], 1)
    want = _torch_act(F.conv2d(xb, w, b, padding=1), act, 0.05)
    close_f16("converted staging vs torch", got, want, out_f16=False)


@pytest.mark.parametrize("segs,cout,B,H,W", [((8, 64, 8), 96, 1, 64, 96), ((8,), 64, 2, 128, 160), ((8, 64, 8), 96, 1, 272, 480),
                                             ((16, 8), 192, 1, 34, 62), ((8,), 192, 1, 64, 64)])
def test_conv5x5s2_space_to_depth_view(segs, cout, B, H, W):
    """f16 conv5x5/s2 with f16 inputs runs as a 3x3/s1 conv over the space-to-depth view (include/dsvc.h: s2d; weights
    remapped by ops.s2d_weights): against torch's stride-2 conv on the same f16-rounded operands, and against the
    direct 5x5/s2 execution of the same layer."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout)
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 5, 2, 2).to(DEV)
        xs = [torch.randn(B, c, H, W) for c in segs]
        ins = [_act(x, bf) for x in xs]
        assert m._s2d_ok(ins) and not Conv2d(96, 96, 5, 2, 2)._s2d_ok([_act(torch.zeros(1, 96, 8, 8), bf)])
        got = m.run(ins, out_dtype=torch.float32).nchw().cpu()
        ops.S2D[0] = False
        try:
            direct = m.run(ins, out_dtype=torch.float32).nchw().cpu()
        finally:
            ops.S2D[0] = True
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    want = F.conv2d(torch.cat([x.to(bf).float() for x in xs], 1), w, b, stride=2, padding=2)
    assert got.shape == want.shape == (B, cout, H // 2, W // 2)
    close_f16("s2d vs torch", got, want, out_f16=False)
    close_f16("s2d vs direct 5x5/s2", got, direct, out_f16=False)


@pytest.mark.parametrize("B,H,W,slope", [(1, 256, 1024, 0.01), (1, 250, 1100, 0.01), (2, 130, 1000, 0.0)])
def test_res_bottleneck_fused_tail(B, H, W, slope):
    """ResBottleneckBlock (modules.py:173-189) with conv2 -> act -> conv3 -> + x fused into one launch of the two-group 3x3
    kernel (include/dsvc.h: post_weight) against the three-launch execution and against torch on f16-rounded operands."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import ResBottleneckBlock
    torch.manual_seed(H + W)
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = ResBottleneckBlock(64, slope=slope).to(DEV)
        x = torch.randn(B, 64, H, W)
        xa = _act(x, bf)
        assert m._fuse_tail_ok(xa, xa, None)
        fused = m.run(xa).nchw().float().cpu()
        ops.FUSE_1X1[0] = False
        try:
            plain = m.run(xa).nchw().float().cpu()
        finally:
            ops.FUSE_1X1[0] = True
    finally:
        ops.set_precision("fp32")
    r = lambda t: t.to(bf).float()
    act = (lambda t: F.relu(t)) if slope < 1e-4 else (lambda t: F.leaky_relu(t, slope))
    xb = r(x)
    o1 = r(act(F.conv2d(xb, r(m.conv1.weight.detach().cpu()), m.conv1.bias.detach().cpu())))
    o2 = r(act(F.conv2d(o1, r(m.conv2.weight.detach().cpu()), m.conv2.bias.detach().cpu(), padding=1)))
    want = F.conv2d(o2, r(m.conv3.weight.detach().cpu()), m.conv3.bias.detach().cpu()) + xb
    # the f16-rounded intermediates (o1, o2) can round the other way than the emulation's on isolated elements: one ulp of an
    # intermediate times a 1x1 weight, far below a missing tap (extra = 3e-4 of the tensor scale)
    close_f16("fused vs torch", fused, want, extra=3e-4)
    close_f16("fused vs three launches", fused, plain, extra=3e-4, both_f16=True)


@pytest.mark.parametrize("segs,cout", [((32, 32), 64), ((64, 64), 64), ((64,), 32), ((64, 32), 128)])
def test_conv1x1_gated_two_segment(segs, cout):
    """ConvBlockResidual tails (modules.py:233-247): up_dim(cat[a, b]) + gate[c] * branch as one 1x1 launch with the register
    epilogue (two input segments, SE gate on the residual)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(sum(segs) + cout)
    H, W = 250, 333
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = Conv2d(sum(segs), cout, 1, 1, 0).to(DEV)
        xs = [torch.randn(1, c, H, W) for c in segs]
        r1, gate = torch.randn(1, cout, H, W), torch.rand(1, cout)
        ins = [_act(x, bf) for x in xs]
        got = m.run(ins if len(ins) > 1 else ins[0], res1=_act(r1, bf), gate=gate.to(DEV), out_dtype=bf).nchw().float().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    want = F.conv2d(torch.cat([x.to(bf).float() for x in xs], 1), w, b) + gate[:, :, None, None] * r1.to(bf).float()
    close_f16("gated 1x1", got, want)


@pytest.mark.parametrize("cin,cout,act,res", [(64, 64, 2, True), (96, 96, 0, True), (96, 96, 2, False), (128, 128, 1, True),
                                              (32, 64, 0, False), (64, 40, 2, True)])
def test_conv1x1_register_epilogue(cin, cout, act, res):
    """f16-out 1x1 convs with activation in {none, ReLU, LeakyReLU} and at most one f16 residual (conv1 / conv3 of the
    ResBottleneck blocks) take the register epilogue of csrc/conv1x1.hip: ragged pixel count, partial last N-tile."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(cin * 3 + cout)
    H, W = 250, 333              # 83250 pixels: not a multiple of 32
    bf = torch.float16
    ops.set_precision("f16")
    try:
        m = Conv2d(cin, cout, 1, 1, 0).to(DEV)
        x, r1 = torch.randn(1, cin, H, W), torch.randn(1, cout, H, W)
        kw = dict(act=act, slope=0.05)
        if res:
            kw["res1"] = _act(r1, bf)
        got = m.run(_act(x, bf), out_dtype=bf, **kw).nchw().float().cpu()
        slab = m.run(_act(x, bf), out_dtype=torch.float32, **kw).nchw().cpu()          # fp32 out: LDS-slab epilogue
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    want = _torch_act(F.conv2d(x.to(bf).float(), w, b), act, 0.05)
    if res:
        want = want + r1.to(bf).float()
    close_f16("1x1 register epilogue vs torch", got, want)
    close_f16("1x1 slab epilogue (fp32 out) vs torch", slab, want, out_f16=False)
    close_f16("1x1 f16-out vs the fp32-out launch of the same conv: one correct rounding", got, slab)


@pytest.mark.parametrize("cin,cout", [(64, 64), (96, 96), (32, 64), (64, 128), (128, 64), (64, 32)])
def test_conv1x1_streaming_kernel(cin, cout):
    """Large f16 pointwise convs take the LDS-free streaming kernel (csrc/conv1x1.hip): ragged pixel count,
    activation, gated residual + second residual."""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    torch.manual_seed(cin + cout)
    H, W = 250, 333              # 83250 pixels: not a multiple of 32
    ops.set_precision("f16")
    try:
        m = Conv2d(cin, cout, 1, 1, 0).to(DEV)
        x, r1, r2 = torch.randn(1, cin, H, W), torch.randn(1, cout, H, W), torch.randn(1, cout, H, W)
        gate = torch.rand(1, cout)
        bf = torch.float16
        plain = m.run(_act(x, bf), act=N.ACT_LRELU, slope=0.01, out_dtype=torch.float32).nchw().cpu()
        fused = m.run(_act(x, bf), res1=_act(r1, bf), gate=gate.to(DEV), res2=_act(r2), out_dtype=bf).nchw().cpu()
    finally:
        ops.set_precision("fp32")
    w, b = m.weight.detach().cpu().to(bf).float(), m.bias.detach().cpu()
    y = F.conv2d(x.to(bf).float(), w, b)
    close_f16("1x1 lrelu", plain, F.leaky_relu(y, 0.01), out_f16=False)
    want = y + gate[:, :, None, None] * r1.to(bf).float() + r2
    close_f16("1x1 fused", fused, want)


def test_warp_matches_grid_sample():
    from deepsvc_amd import ops
    from oracle import nets
    torch.manual_seed(0)
    for C, H, W in ((3, 16, 24), (64, 9, 13), (5, 8, 8)):
        x = torch.randn(2, C, H, W)
        flow = torch.randn(2, 2, H, W) * 4
        flow[0, :, 0, 0] = torch.tensor([-100.0, 50.0])      # far outside: border clamp
        flow[1, :, 1, 1] = torch.tensor([0.0, 0.0])
        got = ops.warp(_act(x), _act(flow)).nchw().cpu()
        report_close(f"warp C={C}", got, nets.warp(x, flow), 1e-4, 2e-4)


def test_resampling_and_layout_ops():
    from deepsvc_amd import ops
    torch.manual_seed(1)
    x = torch.randn(2, 2, 5, 7)
    report_close("upsample2x", ops.upsample2x(_act(x), 2.0).nchw().cpu(),
                 F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) * 2.0, 1e-5, 1e-5)
    x = torch.randn(1, 3, 8, 12)
    report_close("avgpool", ops.pool2x2(_act(x), "avg").nchw().cpu(), F.avg_pool2d(x, 2, 2), 1e-6, 1e-6)
    x = torch.randn(2, 32, 6, 10)
    report_close("maxpool", ops.pool2x2(_act(x), "max").nchw().cpu(), F.max_pool2d(x, 2, 2), 0, 0)
    for C, dt in ((64, torch.float16), (16, torch.float16), (8, torch.float32), (64, torch.float32)):     # 16 bytes per thread
        x = torch.randn(2, C, 6, 10).to(dt).float()
        for mode, ref in (("avg", F.avg_pool2d), ("max", F.max_pool2d)):
            report_close(f"{mode}pool vec {C} {dt}", ops.pool2x2(_act(x, dt), mode).nchw().float().cpu(), ref(x, 2, 2).to(dt).float(), 1e-2 if dt == torch.float16 else 1e-6, 2e-2 if dt == torch.float16 else 1e-6)
    x = torch.randn(2, 64, 5, 6)
    report_close("pixel_shuffle4", ops.pixel_shuffle_nchw(x.to(DEV), 4, dtype=torch.float32).nchw().cpu(), F.pixel_shuffle(x, 4), 0, 0)
    for shape, r in (((2, 256, 5, 70), 4), ((1, 32, 7, 9), 2), ((1, 128, 3, 130), 4)):      # Co % 8 == 0: one thread per pixel x 8 channels
        x = torch.randn(*shape)
        report_close(f"pixel_shuffle{r} vec8 f32", ops.pixel_shuffle_nchw(x.to(DEV), r, dtype=torch.float32).nchw().cpu(), F.pixel_shuffle(x, r), 0, 0)
        report_close(f"pixel_shuffle{r} vec8 f16", ops.pixel_shuffle_nchw(x.to(DEV), r, dtype=torch.float16).nchw().cpu(),
                     F.pixel_shuffle(x, r).to(torch.float16).float(), 0, 0)
    x = torch.randn(2, 37, 9, 11)       # NCHW-contiguous input goes through the transpose kernel
    a = ops.to_act(x.to(DEV))
    assert torch.equal(a.buf.cpu(), x.permute(0, 2, 3, 1).contiguous())
    b16 = ops.to_act(x.to(DEV), torch.float16)
    assert torch.equal(b16.nchw().cpu(), x.to(torch.float16).float())
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)   # channels_last input is taken zero-copy
    assert ops.to_act(xc).buf.data_ptr() == xc.data_ptr()


def test_pointwise_and_reductions():
    from deepsvc_amd import _native as N, ops
    torch.manual_seed(2)
    a, b, c = (torch.randn(1, 3, 9, 10) for _ in range(3))
    w, v = torch.rand(1, 3, 9, 10), torch.rand(1, 3, 9, 10)
    A, B, C, Wt, V = map(_act, (a, b, c, w, v))
    report_close("add", ops.eltwise(N.EW_ADD, A, B).nchw().cpu(), a + b, 1e-6, 1e-6)
    report_close("sub", ops.eltwise(N.EW_SUB, A, B).nchw().cpu(), a - b, 1e-6, 1e-6)
    report_close("blend2", ops.eltwise(N.EW_BLEND2, A, B, d=Wt).nchw().cpu(), w * a + (1 - w) * b, 1e-6, 1e-6)
    report_close("blend3", ops.eltwise(N.EW_BLEND3, A, B, C, Wt, V).nchw().cpu(), w * a + v * b + (1 - w - v) * c, 1e-6, 1e-6)
    report_close("clamp", ops.eltwise(N.EW_CLAMP01, A).nchw().cpu(), a.clamp(0, 1), 0, 0)
    # 8-channel vector path (C % 8 == 0), mixed fp32 / f16 operands
    a, b, c = (torch.randn(2, 16, 9, 10) for _ in range(3))
    w, v = torch.rand(2, 16, 9, 10), torch.rand(2, 16, 9, 10)
    bf = torch.float16
    A, B, C, Wt, V = _act(a, bf), _act(b), _act(c, bf), _act(w), _act(v, bf)
    ab, cb, vb = a.to(bf).float(), c.to(bf).float(), v.to(bf).float()
    report_close("sub8", ops.eltwise(N.EW_SUB, A, B, out_dtype=torch.float32).nchw().cpu(), ab - b, 1e-6, 1e-6)
    report_close("blend3_8", ops.eltwise(N.EW_BLEND3, A, B, C, Wt, V, out_dtype=torch.float32).nchw().cpu(),
                 w * ab + vb * b + (1 - w - vb) * cb, 1e-5, 1e-5)
    report_close("copy8", ops.eltwise(N.EW_COPY, B, out_dtype=bf).nchw().cpu(), b.to(bf).float(), 0, 0)
    x = torch.rand(2, 5, 33, 17) + 0.01
    y = torch.rand(2, 5, 33, 17)
    X, Y = _act(x), _act(y)
    np.testing.assert_allclose(float(ops.reduce("sum", X)), float(x.double().sum()), rtol=1e-6)
    np.testing.assert_allclose(float(ops.reduce("sumlog", X)), float(x.double().log().sum()), rtol=1e-5)
    np.testing.assert_allclose(float(ops.reduce("sqdiff", X, Y)), float(((x - y).double() ** 2).sum()), rtol=1e-5)
    r1, r2 = ops.reduce("sumlog", X), ops.reduce("sumlog", X)
    assert float(r1) == float(r2), "reduction must be run-to-run deterministic"


def test_se_gate():
    from deepsvc_amd.modules import SELayer
    from oracle import nets
    torch.manual_seed(3)
    for C in (32, 64, 128):
        se = SELayer(C)
        x = torch.randn(2, C, 19, 23)
        sd = {"fc.0.weight": se.fc[0].weight.detach(), "fc.2.weight": se.fc[2].weight.detach()}
        want = nets.se_layer(nets.P(sd), x)
        se = se.to(DEV)
        g = se.gate(_act(x)).cpu()
        report_close(f"se gate C={C}", g[:, :, None, None] * x, want, 1e-5, 1e-5)


def test_gaussian_conditional_kernels():
    from deepsvc_amd import ops
    from oracle import entropy as E
    torch.manual_seed(4)
    gc = E.gc_tables()
    table = gc["scale_table"]
    B, C, H, W = 2, 12, 7, 9
    y, mu = torch.randn(B, C, H, W) * 3, torch.randn(B, C, H, W)
    sc = torch.exp(torch.randn(B, C, H, W) * 2 - 1)
    sc[0, 0, 0, :] = table[:9]            # exactly on table levels (the <= comparison)
    sc[0, 1, 0, 0], sc[0, 1, 0, 1] = -1.0, 1e6
    y[0, 2, 0, 0], mu[0, 2, 0, 0] = 2.5, 0.0   # round-half-to-even
    y[0, 2, 0, 1], mu[0, 2, 0, 1] = 3.5, 0.0
    Y, MU, SC = _act(y), _act(mu), _act(sc)
    yh = ops.new_act(B, H, W, C, torch.float32, DEV)
    sym, idx = ops.gc_quantize(Y, MU, SC, table.to(DEV), True, True, yh)
    want_sym = torch.round(y - mu).int()
    assert torch.equal(sym.cpu().view(B, C, H, W), want_sym)
    assert torch.equal(idx.cpu().view(B, C, H, W), E.gc_build_indexes(sc, table))
    assert torch.equal(yh.nchw().cpu(), want_sym.float() + mu)
    yh2 = ops.new_act(B, H, W, C, torch.float32, DEV)
    ops.gc_dequantize(sym, MU, yh2)
    assert torch.equal(yh2.nchw().cpu(), want_sym.float() + mu)
    lik = ops.new_act(B, H, W, C, torch.float32, DEV)
    ops.gc_likelihood(Y, MU, SC, None, lik)
    _, want_lik = E.gc_forward(y, sc, mu)
    report_close("gc likelihood", lik.nchw().cpu(), want_lik, 2e-4, 1e-7)


def test_entropy_bottleneck_kernels():
    from deepsvc_amd.entropy_models import EntropyBottleneck
    from oracle import entropy as E
    torch.manual_seed(6)
    eb = EntropyBottleneck(10)
    with torch.no_grad():
        for n, p in eb.named_parameters():
            if n != "quantiles":
                p.add_(torch.randn_like(p) * 0.2)
        eb.quantiles[:, 0, 1] += torch.randn(10)
    sd = {k: v.detach().clone() for k, v in eb.state_dict().items()}
    eb.update()
    eb = eb.to(DEV)
    z = torch.randn(2, 10, 5, 6) * 4
    zh_want, lik_want = E.eb_forward(sd, "", z)
    zh, lik = eb(z.to(DEV))
    report_close("eb z_hat", zh.cpu(), zh_want, 0, 1e-6)
    report_close("eb likelihood", lik.cpu(), lik_want, 5e-4, 1e-7)
    strings = eb.compress(z.to(DEV))
    assert strings == E.eb_compress(sd, "", E.eb_tables(sd, ""), z)
    report_close("eb decompress", eb.decompress(strings, (5, 6)).cpu(), zh_want, 0, 1e-6)


def test_f16_stores_saturate():
    """binary16 tops out at 65504; every kernel that stores f16 runs with MODE.FP16_OVFL set (csrc/common.h: DSVC_F16_SAT), so an
    out-of-range activation is stored as +-65504 instead of +-inf -- through the conv epilogues (generic, ping-pong, 1x1), the
    layout kernels and the element-wise kernels -- and NaN / true infinities are not hidden."""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    try:
        torch.manual_seed(0)
        for k, H, W in ((3, 16, 32), (3, 256, 1024), (1, 256, 1024), (1, 8, 8)):     # generic / ping-pong / direct 1x1 / generic 1x1
            m = Conv2d(64, 64, k, 1, padding=k // 2)
            with torch.no_grad():
                m.weight.fill_(0.0)
                m.weight[:, 0, k // 2, k // 2] = 1.0
                m.bias.fill_(0.0)
            x = torch.zeros(1, 64, H, W)
            x[0, 0, 3, 5], x[0, 0, 4, 6], x[0, 0, 5, 7] = 60000.0, -60000.0, 1.0     # representable in f16
            with torch.no_grad():
                m.weight[:, 0, k // 2, k // 2] = 4.0                                      # 240000: above the f16 range
            out = m.to(DEV).run(_act(x, torch.float16), out_dtype=torch.float16)
            o = out.nchw().cpu()
            assert bool(torch.isfinite(o).all()), f"k={k} {H}x{W}: non-finite f16 store"
            assert float(o[0, 7, 3, 5]) == 65504.0 and float(o[0, 7, 4, 6]) == -65504.0 and float(o[0, 7, 5, 7]) == 4.0, (k, H, W, o[0, 7, 3, 5])
        # layout conversion and element-wise kernels
        big = torch.full((1, 8, 4, 8), 1.0e6)
        a = ops.to_act(big.to(DEV), torch.float16)
        assert float(a.buf.float().max()) == 65504.0
        s = ops.eltwise(N.EW_ADD, a, a, out_dtype=torch.float16)
        assert float(s.buf.float().max()) == 65504.0 and bool(torch.isfinite(s.buf.float()).all())
        nan = torch.full((1, 8, 4, 8), float("nan"))
        assert bool(torch.isnan(ops.to_act(nan.to(DEV), torch.float16).buf.float()).all()), "NaN must stay NaN"
    finally:
        ops.set_precision("fp32")


def test_pp_kernel_mfma_shapes_agree():
    """The two-group 3x3 kernel exists on two MFMA shapes (option pp16: 0 = 32x32x16, 2 = 16x16x32; different lane layouts, LDS
    read pattern, epilogue swaps and residual layout, same arithmetic): every epilogue variant gives bit-identical f16 outputs on
    both, except the fused 1x1 tail, whose second GEMM sums K = 64 in two steps instead of four (<= one f16 ulp apart).  The
    32x32x16 side is held against F.conv2d by the tests above."""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    h = torch.float16
    try:
        torch.manual_seed(3)
        H, W = 264, 1000                      # ragged: 33 tile rows, the last tile column 8 pixels wide; 1056 tiles
        x = ops.Act(torch.randn((1, H, W, 64), device=DEV).to(h))
        x32 = ops.Act(torch.randn((1, H, W, 32), device=DEV).to(h))
        r1, r2 = ops.Act(torch.randn((1, H, W, 64), device=DEV).to(h)), ops.Act(torch.randn((1, H, W, 64), device=DEV).to(h))
        r32 = ops.Act(torch.randn((1, H, W, 32), device=DEV).to(h))
        c64, c32, c1 = Conv2d(64, 64, 3, 1, 1).to(DEV), Conv2d(64, 32, 3, 1, 1).to(DEV), Conv2d(32, 64, 3, 1, 1).to(DEV)
        post = Conv2d(64, 64, 1, 1, 0).to(DEV)._packed([64], DEV)
        cases = [("plain", c64, x, {}), ("act", c64, x, dict(act=N.ACT_LRELU, slope=0.1)), ("pre+act relu", c64, x, dict(pre_act=N.ACT_RELU, act=N.ACT_RELU, slope=0.0)),
                 ("pre lrelu", c64, x, dict(pre_act=N.ACT_LRELU, slope=0.01)), ("res1", c64, x, dict(res1=r1)),
                 ("act+res1+res2", c64, x, dict(act=N.ACT_LRELU, slope=0.01, res1=r1, res2=r2)),
                 ("64->32 + res", c32, x, dict(res1=r32)), ("32->64", c1, x32, dict(act=N.ACT_RELU)),
                 ("fused tail", c64, x, dict(act=N.ACT_LRELU, slope=0.01, res1=r1, post=post))]
        for name, m, inp, kw in cases:
            outs = []
            for opt in (0, 2):
                ops.set_option("pp16", opt)
                before = ops.get_counter("conv3x3_pp")
                outs.append(m.run(inp, out_dtype=h, **kw).buf.float().cpu())
                assert ops.get_counter("conv3x3_pp") == before + 1, f"{name}: the two-group kernel did not take this launch"
            a, b = outs
            assert bool(torch.isfinite(b).all())
            if "post" in kw:
                close_f16(f"pp16 {name}", b, a, both_f16=True)
            else:
                assert torch.equal(a, b), f"{name}: {int((a != b).sum())} elements differ between the MFMA shapes, max {float((a - b).abs().max()):.3e}"
    finally:
        ops.set_option("pp16", 1)
        ops.set_precision("fp32")


def test_pp_kernel_touches_nothing_outside_its_tensors():
    """Round 5: the two-group 3x3 kernel's patch loads take the tile origin as the SCALAR offset of a descriptor that starts one image row + one
    pixel before the tensor (csrc/conv3x3_pp.hip: addressing diet), and the hardware range check covers only the vector offset: whatever must
    not be read or written is masked by the kernel itself.  Input, residual and output are carved out of ONE buffer full of NaN (4 M elements
    of NaN between them): the result must equal the run on stand-alone tensors bit for bit, contain no NaN, and leave every NaN around the
    tensors in place -- for the plain, one-chunk, residual, 32-channel and narrow-head variants at the frame's 256x1024 shape."""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    h = torch.float16
    prev_tap = ops.set_option("headtap", 0)        # the HEAD variant of the two-group kernel is under test here (the tap-GEMM kernel takes 64 -> 3 first)
    try:
        H, W = 256, 1024
        torch.manual_seed(0)
        for name, cin, cout, kw_f in [("plain 64->64", 64, 64, lambda r: {}), ("act 64->64", 64, 64, lambda r: dict(act=N.ACT_LRELU, slope=0.01)),
                                      ("32->64", 32, 64, lambda r: dict(act=N.ACT_RELU)), ("16->64", 16, 64, lambda r: {}), ("res1", 64, 64, lambda r: dict(res1=r)),
                                      ("64->32", 64, 32, lambda r: {}), ("head 64->3", 64, 3, lambda r: {})]:
            conv = Conv2d(cin, cout, 3, 1, 1).to(DEV)
            x_clean = torch.randn((1, H, W, cin), device=DEV).to(h)
            r_clean = torch.randn((1, H, W, cout), device=DEV).to(h)
            odt = torch.float32 if cout <= 8 else h
            before = ops.get_counter("conv3x3_pp")
            ref = conv.run(ops.Act(x_clean), out_dtype=odt, **kw_f(ops.Act(r_clean))).buf.clone()
            assert ops.get_counter("conv3x3_pp") == before + 1, f"{name}: the two-group kernel did not take this launch"
            pad = 4 * 1024 * 1024
            big = torch.full((pad * 4 + H * W * (cin + 2 * 64) + 64,), float("nan"), device=DEV, dtype=h)
            xo, ro, oo = pad, 2 * pad + H * W * cin, 3 * pad + H * W * (cin + 64)
            carve = lambda off, C: big[off:off + H * W * C].view(1, H, W, C)
            x = carve(xo, cin); x.copy_(x_clean)
            r = carve(ro, cout); r.copy_(r_clean)
            if odt == h:
                out = conv.run(ops.Act(x), ops.Act(carve(oo, cout)), out_dtype=h, **kw_f(ops.Act(r))).buf
            else:
                out = conv.run(ops.Act(x), out_dtype=odt, **kw_f(ops.Act(r))).buf
            torch.cuda.synchronize()
            assert not bool(torch.isnan(out.float()).any()), f"{name}: NaN from outside the tensors reached the output"
            assert torch.equal(out, ref), f"{name}: differs from the run on stand-alone tensors"
            mask = torch.ones_like(big, dtype=torch.bool)
            mask[xo:xo + H * W * cin] = False; mask[ro:ro + H * W * cout] = False
            if odt == h:
                mask[oo:oo + H * W * cout] = False
            assert bool(torch.isnan(big[mask]).all()), f"{name}: wrote outside its output tensor"
    finally:
        ops.set_option("headtap", prev_tap)
        ops.set_precision("fp32")


@pytest.mark.parametrize("name,counter,segs,cout,k,H,W", [
    ("tap-GEMM head 64->3", "conv_headtap", (64,), 3, 3, 250, 1000), ("tap-GEMM head 64+64->3", "conv_headtap", (64, 64), 3, 3, 131, 520),
    ("cin8 7x7 8->32", "conv_cin8", (8,), 32, 7, 130, 500), ("cin8 3x3 8->64", "conv_cin8", (8,), 64, 3, 129, 515),
    ("row-reuse 7x7 64->32", "conv7x7_rr", (64,), 32, 7, 250, 500), ("row-reuse 7x7 32->64", "conv7x7_rr", (32,), 64, 7, 245, 505),
    ("row-reuse 7x7 32->16", "conv7x7_rr", (32,), 16, 7, 250, 500)])
def test_round5_kernels_touch_nothing_outside_their_tensors(name, counter, segs, cout, k, H, W):
    """The persistent / halo-staging kernels of round 5's second half (csrc/conv_headtap.hip, conv_cin8.hip, conv7x7_rr.hip) mask every access
    outside the picture themselves (bounds-checked buffer loads with an explicit out-of-range offset; whole-row stores of ragged tiles): inputs
    and output are carved out of ONE buffer full of NaN with 4 M elements of NaN between them -- the result must equal the run on stand-alone
    tensors bit for bit, contain no NaN, and leave every NaN around the tensors in place (ragged pictures: the last tile row / column is partial)."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    h = torch.float16
    try:
        torch.manual_seed(1)
        conv = Conv2d(sum(segs), cout, k, 1, k // 2).to(DEV)
        clean = [torch.randn((1, H, W, c), device=DEV).to(h) for c in segs]
        odt = torch.float32 if cout <= 8 else h
        before = ops.get_counter(counter)
        ref = conv.run([ops.Act(x) for x in clean], out_dtype=odt, act=1).buf.clone()
        assert ops.get_counter(counter) == before + 1, f"{name}: the kernel under test did not take this launch"
        pad = 4 * 1024 * 1024
        n_in = [H * W * c for c in segs]
        big = torch.full((pad * (len(segs) + 2) + sum(n_in) + H * W * cout + 64,), float("nan"), device=DEV, dtype=h)
        offs, o = [], pad
        for n in n_in:
            offs.append(o); o += n + pad
        oo = o
        xs = []
        for off, c, xc in zip(offs, segs, clean):
            v = big[off:off + H * W * c].view(1, H, W, c); v.copy_(xc); xs.append(ops.Act(v))
        if odt == h:
            out = conv.run(xs, ops.Act(big[oo:oo + H * W * cout].view(1, H, W, cout)), out_dtype=h, act=1).buf
        else:
            out = conv.run(xs, out_dtype=odt, act=1).buf
        torch.cuda.synchronize()
        assert ops.get_counter(counter) == before + 2
        assert not bool(torch.isnan(out.float()).any()), f"{name}: NaN from outside the tensors reached the output"
        assert torch.equal(out, ref), f"{name}: differs from the run on stand-alone tensors"
        mask = torch.ones_like(big, dtype=torch.bool)
        for off, n in zip(offs, n_in):
            mask[off:off + n] = False
        if odt == h:
            mask[oo:oo + H * W * cout] = False
        assert bool(torch.isnan(big[mask]).all()), f"{name}: wrote outside its output tensor"
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("name,counter,cin,cout,k,H,W", [
    ("tap-GEMM head 64->3", "conv_headtap", 64, 3, 3, 130, 520), ("tap-GEMM head 32->2", "conv_headtap", 32, 2, 3, 140, 500),
    ("cin8 7x7 8->32", "conv_cin8", 8, 32, 7, 90, 260), ("cin8 3x3 8->64", "conv_cin8", 8, 64, 3, 90, 260),
    ("row-reuse 7x7 64->32", "conv7x7_rr", 64, 32, 7, 180, 400), ("row-reuse 7x7 32->64", "conv7x7_rr", 32, 64, 7, 180, 400),
    ("row-reuse 7x7 32->16", "conv7x7_rr", 32, 16, 7, 180, 400)])
def test_round5_kernels_on_channel_slices(name, counter, cin, cout, k, H, W):
    """The same kernels through VIEWS, as the frame's virtual concatenations use them: the input is a channel slice of a wider buffer (pixel pitch
    larger than its channel count; the 8-channel kernel needs contiguous pixels and takes a plain tensor), the output a channel slice of a wider
    buffer pre-filled with 7 (and the head's residual another slice): results equal the run on plain tensors bit for bit, and the channels
    beside the output slice keep their 7s -- the whole-row stores write exactly the slice's bytes of every pixel row."""
    from deepsvc_amd import ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    h = torch.float16
    try:
        torch.manual_seed(2)
        conv = Conv2d(cin, cout, k, 1, k // 2).to(DEV)
        x_clean = torch.randn((1, H, W, cin), device=DEV).to(h)
        head = cout <= 3
        odt = torch.float32 if head else h
        kw = dict(act=1)
        r_clean = torch.randn((1, H, W, cout), device=DEV) if head else None
        if head:
            kw["res1"] = ops.Act(r_clean)
        before = ops.get_counter(counter)
        ref = conv.run(ops.Act(x_clean), out_dtype=odt, **kw).buf.clone()
        assert ops.get_counter(counter) == before + 1, f"{name}: the kernel under test did not take this launch"
        if cin == 8:
            x = ops.Act(x_clean)
        else:
            wide_in = torch.randn((1, H, W, cin + 32), device=DEV).to(h)
            wide_in[..., 16:16 + cin] = x_clean
            x = ops.Act(wide_in, 16, cin)
        oc0 = 8 if odt == h else 4                     # 16-byte aligned slice start in either type
        wide_out = torch.full((1, H, W, cout + 24), 7.0, device=DEV, dtype=odt)
        if head:
            wide_res = torch.randn((1, H, W, cout + 5), device=DEV)
            wide_res[..., 2:2 + cout] = r_clean
            kw["res1"] = ops.Act(wide_res, 2, cout)
        out = conv.run(x, ops.Act(wide_out, oc0, cout), out_dtype=odt, **kw)
        torch.cuda.synchronize()
        assert ops.get_counter(counter) == before + 2, f"{name}: the kernel under test did not take the launch on views"
        got = wide_out[..., oc0:oc0 + cout]
        assert torch.equal(got, ref), f"{name}: differs from the run on plain tensors"
        assert bool((wide_out[..., :oc0] == 7).all()) and bool((wide_out[..., oc0 + cout:] == 7).all()), f"{name}: wrote beside its output slice"
    finally:
        ops.set_precision("fp32")


def test_pp_kernel_mixed_precision_residual_epilogue():
    """Option pp_mix (round 5): f16(x + r) of the two-group 3x3 kernel's residual variants as v_fma_mixlo_f16 / v_fma_mixhi_f16 (the f16
    residual enters the FMA directly, the fp32 sum is rounded ONCE to f16) instead of cvt + add + cvt_pk (rounded to fp32, then to f16).
    Every residual variant, on both MFMA shapes, on a ragged picture and through views with a channel offset: at most one f16 ulp from the
    two-step form, in a small share of the elements (the double-rounding ties), and NOT further from the fp32 result than the two-step
    form is.  Variants without a residual operand do not change at all.  (The kernel tests above run with the option on and hold it to
    half an ulp of F.conv2d.)"""
    from deepsvc_amd import _native as N, ops
    from deepsvc_amd.modules import Conv2d
    ops.set_precision("f16")
    h = torch.float16
    try:
        torch.manual_seed(5)
        H, W = 267, 1000                      # ragged both ways: 34 tile rows (the last 3 pixels high), last tile column 8 pixels wide
        x = ops.Act(torch.randn((1, H, W, 64), device=DEV).to(h))
        wide = torch.randn((1, H, W, 160), device=DEV).to(h)
        r1, r2 = ops.Act(wide, 32, 64), ops.Act(torch.randn((1, H, W, 64), device=DEV).to(h))     # residual 1: a channel slice of a wider buffer
        r32 = ops.Act(wide, 96, 32)
        c64, c32 = Conv2d(64, 64, 3, 1, 1).to(DEV), Conv2d(64, 32, 3, 1, 1).to(DEV)
        post = Conv2d(64, 64, 1, 1, 0).to(DEV)._packed([64], DEV)
        cases = [("plain", c64, x, {}), ("res1", c64, x, dict(res1=r1)), ("pre+res1", c64, x, dict(pre_act=N.ACT_LRELU, slope=0.01, res1=r1)),
                 ("act+res1+res2", c64, x, dict(act=N.ACT_LRELU, slope=0.01, res1=r1, res2=r2)), ("64->32 + res", c32, x, dict(res1=r32)),
                 ("fused tail", c64, x, dict(act=N.ACT_LRELU, slope=0.01, res1=r1, post=post))]
        for shape_opt in (0, 2):
            ops.set_option("pp16", shape_opt)
            for name, m, inp, kw in cases:
                outs = []
                for mix in (0, 1):
                    ops.set_option("pp_mix", mix)
                    out = ops.Act(torch.full((1, H, W, 96), 7.0, device=DEV).to(h), 16, m.out_channels)     # a view with a channel offset
                    before = ops.get_counter("conv3x3_pp")
                    m.run(inp, out, out_dtype=h, **kw)
                    assert ops.get_counter("conv3x3_pp") == before + 1, f"{name}: the two-group kernel did not take this launch"
                    outs.append(out.buf.cpu())
                a, b = outs
                assert bool(torch.isfinite(b.float()).all())
                assert bool((b[..., :16] == 7.0).all()) and bool((b[..., 16 + m.out_channels:] == 7.0).all()), f"{name}: wrote outside its channel slice"
                if "res1" not in kw:
                    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{name}: changed without a residual operand"
                    continue
                ia, ib = a.view(torch.int16).to(torch.int32), b.view(torch.int16).to(torch.int32)
                diff = (ia - ib).abs()
                share = float((diff > 0).float().mean())
                print(f"pp16={shape_opt} {name}: {share:.2e} of the elements differ by one f16 ulp with the fused rounding")
                assert int(diff.max()) <= 1 and share < 5e-3, f"pp16={shape_opt} {name}: max {int(diff.max())} ulp, share {share:.3e}"
    finally:
        ops.set_option("pp16", 1)
        ops.set_option("pp_mix", 1)
        ops.set_precision("fp32")
