"""MI355X-native counterparts of the reference's building blocks (reference: modules.py).

Same class names, constructor arguments and attribute paths as the reference, so a reference
checkpoint's ``state_dict`` loads unchanged and ``test_video.py`` can import these instead.
Differences are all below the API:
  * activations are NHWC slices (`ops.Act`); ``torch.cat`` is a list of slices consumed by one
    multi-segment conv launch, ``x + f(x)`` / activations / PixelShuffle / the SE gate are conv epilogues;
  * every ``forward`` drives hand-written HIP kernels through the C ABI and refuses CPU tensors.
Public ``forward`` methods accept and return reference-style NCHW tensors; the ``run`` methods are the
slice-level entry points the frame pipeline (video_model.DeepSVC) chains without layout round trips.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from . import ops
from .ops import Act, to_act

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """reference modules.py:17-18 / image_model.py:18-19."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


# ------------------------------------------------------------------------------------------------
# parameter holders
# ------------------------------------------------------------------------------------------------
def _bump_owner_epoch(module):
    """A codec whose HIP graphs hold this layer's packed-weight pointers registers a counter here (image_model._epoch_ref):
    repacking the layer invalidates THAT codec's graphs only -- not those of other model instances (GOP lanes)."""
    ref = module.__dict__.get("_epoch_ref")
    if ref is not None:
        ref[0] += 1


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (same keys) executed by the fused MFMA conv kernel."""

    def _packed(self, seg_channels, device, phys_channels=None, s2d=False):
        w, b = self.weight, self.bias
        key = (tuple(seg_channels), None if phys_channels is None else tuple(phys_channels), ops.RT.compute, device,
               w._version, -1 if b is None else b._version, w.data_ptr(), s2d)
        last = self.__dict__.get("_pack_last")            # (key, PackedConv): one entry -- a layer is used with one configuration
        if last is not None and last[0] == key:
            return last[1]
        pc = ops.PackedConv(w, b, seg_channels, self.stride[0], self.padding[0], ops.RT.compute, device, phys_channels, s2d=s2d)
        self.__dict__["_pack_last"] = (key, pc)
        _bump_owner_epoch(self)
        return pc

    def _s2d_ok(self, phys_inputs):
        """f16 mode: conv5x5/s2 runs as a 3x3/s1 conv over the space-to-depth view of its inputs (include/dsvc.h: s2d) when
        every input is f16 with 16-byte pixel rows: no 25-tap weight re-staging per tile, and 8-channel segments fill a
        32-channel chunk exactly (8 channels x 4 phases) instead of being padded to it."""
        if not (ops.S2D[0] and ops.RT.compute == N.F16 and self.kernel_size == (5, 5) and self.stride == (2, 2) and self.padding == (2, 2)
                and all(a.dtype == torch.float16 and a.C % 8 == 0 and a.ld % 8 == 0 and a.ptr % 16 == 0 and a.H % 2 == 0 and a.W % 2 == 0
                        for a in phys_inputs)):
            return False
        # the view executes 36 MACs per input channel (9 taps x 4 phases, 11 of the 36 are structural zeros), the direct
        # form 25 per channel of every 32-channel chunk it touches: the view wins for the first-layer convs whose 8-channel
        # segments would be padded to full chunks (8+64+8 -> 96: 952 -> 632 us), not for 64/96-channel inputs (measured)
        return 36 * sum(a.C for a in phys_inputs) <= 25 * sum(-(-a.C // 32) * 32 for a in phys_inputs)

    def _pack_small_runs(self, inputs):
        """f16 mode: maximal runs of small fp32 inputs (frames, flows: 2-3 channels each) are merged into one f16
        buffer padded with zero channels to a multiple of 8, so the conv's fast 16-byte staging path applies.
        Returns (physical inputs, logical channel split, physical channel split)."""
        small = lambda a: a.dtype == torch.float32 and a.C < 32 and (a.C % 8 != 0)
        phys, logical, physical, run = [], [], [], []

        def flush():
            if not run:
                return
            lc = sum(a.C for a in run)
            pc = -(-lc // 8) * 8
            x0 = run[0]
            cache = self.__dict__.setdefault("_small_cache", {})
            key = (len(phys), x0.B, x0.H, x0.W, pc, str(x0.device))
            if key not in cache:          # zero-initialised once: the padding channels are never written afterwards
                cache[key] = Act(torch.zeros((x0.B, x0.H, x0.W, pc), dtype=torch.float16, device=x0.device))
            phys.append(ops.pack_small_inputs(run, cache[key]))
            logical.append(lc)
            physical.append(pc)
            run.clear()

        for a in inputs:
            if small(a):
                run.append(a)
            else:
                flush()
                phys.append(a)
                logical.append(a.C)
                physical.append(a.C)
        flush()
        return phys, logical, physical

    def run(self, inputs, out=None, **epilogue):
        inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        if ops.RT.compute == N.F16 and any(a.dtype == torch.float32 and a.C % 8 != 0 and a.C < 32 for a in inputs):
            phys, logical, physical = self._pack_small_runs(inputs)
            if len(phys) <= 3:
                return ops.conv2d(self._packed(logical, phys[0].device, physical, s2d=self._s2d_ok(phys)), phys, out, **epilogue)
        return ops.conv2d(self._packed([a.C for a in inputs], inputs[0].device, s2d=self._s2d_ok(inputs)), inputs, out, **epilogue)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class Deconv5x5s2(nn.ConvTranspose2d):
    """compressai ``deconv(i, o, 5, 2)`` parameters, executed as a 3x3 conv onto 4 sub-pixel phases
    whose PixelShuffle(2) is folded into the store (csrc/host_misc.cpp)."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, kernel_size=5, stride=2, output_padding=1, padding=2)

    def _packed(self, device):
        key = (ops.RT.compute, str(device), self.weight._version, self.bias._version, self.weight.data_ptr())
        cache = self.__dict__.setdefault("_pack_cache", {})
        if key not in cache:
            cache.clear()
            w3, b4 = ops.deconv_as_conv_weights(self.weight, self.bias)
            cache[key] = ops.PackedConv(w3, b4, [self.in_channels], 1, 1, ops.RT.compute, device)
            _bump_owner_epoch(self)
        return cache[key]

    def run(self, x, out=None, **epilogue):
        return ops.conv2d(self._packed(x.device), [x], out, shuffle=2, **epilogue)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


def conv(cin, cout, kernel_size=5, stride=2):
    """compressai.models.utils.conv (used at modules.py:392, image_model.py:53-146)."""
    return Conv2d(cin, cout, kernel_size, stride, padding=kernel_size // 2)


def _act_code(slope):
    return (N.ACT_RELU, 0.0) if slope < 1e-4 else (N.ACT_LRELU, float(slope))


# ------------------------------------------------------------------------------------------------
# warp & friends (functional API of the reference)
# ------------------------------------------------------------------------------------------------
def torch_warp(tensorInput, tensorFlow):
    """reference modules.py:25-62."""
    x, f = to_act(tensorInput), to_act(tensorFlow)
    return ops.warp(x, f).nchw()


def bilinearupsacling(inputfeature):
    """reference modules.py:107-112."""
    return ops.upsample2x(to_act(inputfeature)).nchw()


def read_image(filepath):
    """reference modules.py:65-68: image file -> float tensor [3,H,W] in [0,1] (PIL + ToTensor there; no torchvision here)."""
    import os
    assert os.path.isfile(filepath)
    import numpy as np
    from PIL import Image
    img = np.asarray(Image.open(filepath).convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(img.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div_(255.0)


def cal_psnr(a, b):
    """reference modules.py:71-73 (a, b: same-shape tensors in [0,1])."""
    mse = F.mse_loss(a, b).item()
    return -10 * math.log10(mse)


def pad(x, p=2 ** 6):
    """reference modules.py:76-89: centred zero pad to a multiple of p (host-side helper of the harness)."""
    h, w = x.size(2), x.size(3)
    H, W = -(-h // p) * p, -(-w // p) * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop(x, size):
    """reference modules.py:92-104."""
    H, W = x.size(2), x.size(3)
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (-left, -(W - w - left), -top, -(H - h - top)), mode="constant", value=0)


# ------------------------------------------------------------------------------------------------
# SpyNet motion estimation
# ------------------------------------------------------------------------------------------------
class MEBasic(nn.Module):
    """reference modules.py:123-139: 7x7 convs 8-32-64-32-16-2, ReLU between."""

    def __init__(self):
        super().__init__()
        for i, (ci, co) in enumerate([(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)], 1):
            setattr(self, f"conv{i}", Conv2d(ci, co, 7, 1, padding=3))

    def run(self, segments, residual=None):
        x = self.conv1.run(segments, act=N.ACT_RELU)
        for name in ("conv2", "conv3", "conv4"):
            x = getattr(self, name).run(x, act=N.ACT_RELU)
        return self.conv5.run(x, res1=residual, out_dtype=torch.float32)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class ME_Spynet(nn.Module):
    """reference modules.py:142-170."""

    def __init__(self):
        super().__init__()
        self.L = 4
        self.moduleBasic = nn.ModuleList([MEBasic() for _ in range(self.L)])

    def run(self, im1, im2):
        pyr1, pyr2 = [im1], [im2]
        for _ in range(self.L - 1):
            pyr1.append(ops.pool2x2(pyr1[-1], "avg"))
            pyr2.append(ops.pool2x2(pyr2[-1], "avg"))
        top = pyr2[-1]
        zkey = (top.B, top.H // 2, top.W // 2, str(top.device))
        flow = self.__dict__.setdefault("_zero_flow", {}).get(zkey)       # the all-zero coarsest flow: allocated once, only ever read
        if flow is None:
            flow = self.__dict__["_zero_flow"][zkey] = Act(torch.zeros((top.B, top.H // 2, top.W // 2, 2), dtype=torch.float32, device=top.device))
        for level in range(self.L):
            up = ops.upsample2x(flow, 2.0)
            k = self.L - 1 - level
            flow = self.moduleBasic[level].run([pyr1[k], ops.warp(pyr2[k], up), up], residual=up)
        return flow

    def forward(self, im1, im2):
        return self.run(to_act(im1), to_act(im2)).nchw()


# ------------------------------------------------------------------------------------------------
# residual blocks
# ------------------------------------------------------------------------------------------------
class ResBottleneckBlock(nn.Module):
    """reference modules.py:173-189 (= image_model.py:22-38): x + 1x1 -> act -> 3x3 -> act -> 1x1."""

    def __init__(self, channel, slope=0.01):
        super().__init__()
        self.conv1 = Conv2d(channel, channel, 1, 1, padding=0)
        self.conv2 = Conv2d(channel, channel, 3, 1, padding=1)
        self.conv3 = Conv2d(channel, channel, 1, 1, padding=0)
        self._act, self._slope = _act_code(slope)

    def _fuse_tail_ok(self, x, o, out):
        """conv2 -> act -> conv3 -> + x as ONE launch (the 1x1 rides on the 3x3 kernel's epilogue, include/dsvc.h:
        post_weight) where the two-group 3x3 kernel applies: f16 mode, 64 channels, f16 16-byte-aligned tensors,
        >= 1024 tiles of 8x32 pixels.  Saves a full read + write of the activation per block."""
        if not (ops.FUSE_1X1[0] and ops.RT.compute == N.F16 and self.conv2.in_channels == 64 and self.conv2.out_channels == 64):
            return False
        ok = lambda a: a.dtype == torch.float16 and a.C == 64 and a.ld % 8 == 0 and a.ptr % 16 == 0
        tiles = x.B * (-(-x.H // 8)) * (-(-x.W // 32))
        return ok(x) and ok(o) and (out is None or ok(out)) and tiles >= 1024 and x.B * x.H * x.W < (1 << 24)

    def run(self, x, out=None):
        o = self.conv1.run(x, act=self._act, slope=self._slope)
        if self._fuse_tail_ok(x, o, out):
            post = self.conv3._packed([64], o.device)
            return self.conv2.run(o, out, act=self._act, slope=self._slope, res1=x, post=post, out_dtype=torch.float16)
        o = self.conv2.run(o, act=self._act, slope=self._slope)
        return self.conv3.run(o, out, res1=x)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class ResBlock1(nn.Module):
    """reference modules.py:192-214: x + [act] conv act conv [act]; the leading activation is fused into
    conv1's input staging, the residual add(s) into conv2's epilogue."""

    def __init__(self, channel, slope=0.01, start_from_relu=True, end_with_relu=False, bottleneck=False):
        super().__init__()
        mid = channel // 2 if bottleneck else channel
        self.conv1 = Conv2d(channel, mid, 3, padding=1)
        self.conv2 = Conv2d(mid, channel, 3, padding=1)
        self._act, self._slope = _act_code(slope)
        self._pre = self._act if start_from_relu else N.ACT_NONE
        self._post = self._act if end_with_relu else N.ACT_NONE

    def run(self, x, out=None, extra_residual=None):
        o = self.conv1.run(x, pre_act=self._pre, act=self._act, slope=self._slope)
        return self.conv2.run(o, out, act=self._post, slope=self._slope, res1=x, res2=extra_residual)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class SELayer(nn.Module):
    """reference modules.py:217-230.  Holds fc.0 / fc.2 (bias-free Linear); the gate is produced by a
    deterministic two-stage reduction kernel and applied inside the consumer's conv epilogue."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def gate(self, x):
        w1 = self.fc[0].weight.detach().to(x.device, torch.float32).contiguous()
        w2 = self.fc[2].weight.detach().to(x.device, torch.float32).contiguous()
        return ops.se_gate(x, w1, w2)

    def forward(self, x):
        a = to_act(x)
        g = self.gate(a)
        return x * g[:, :, None, None]


class ConvBlockResidual(nn.Module):
    """reference modules.py:233-247: up_dim(x) + SE(conv(lrelu(conv(x)))).  The 1x1 `up_dim` launch carries
    the SE-gated branch as its residual: out = up_dim(x) + gate[c] * branch."""

    def __init__(self, ch_in, ch_out, se_layer=True):
        super().__init__()
        self.conv = nn.Sequential(Conv2d(ch_in, ch_out, 3, 1, 1), nn.LeakyReLU(0.01), Conv2d(ch_out, ch_out, 3, 1, 1),
                                  SELayer(ch_out) if se_layer else nn.Identity())
        self.up_dim = Conv2d(ch_in, ch_out, 1, 1, 0)
        self._se = se_layer

    def run(self, segments, out=None):
        segments = segments if isinstance(segments, (list, tuple)) else [segments]
        branch = self.conv[2].run(self.conv[0].run(segments, act=N.ACT_LRELU, slope=0.01))
        gate = self.conv[3].gate(branch) if self._se else None
        return self.up_dim.run(segments, out, res1=branch, gate=gate)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class _SubpelConv(nn.Sequential):
    """`Sequential(Conv2d, PixelShuffle)` key layout ('0.weight'); PixelShuffle folded into the conv store."""

    def __init__(self, cin, cout, r, k):
        super().__init__(Conv2d(cin, cout * r * r, k, 1, padding=k // 2), nn.PixelShuffle(r))
        assert r == 2
        self.r = r

    def run(self, x, out=None, **epilogue):
        return self[0].run(x, out, shuffle=self.r, **epilogue)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


def subpel_conv1x1(in_ch, out_ch, r=1):
    """reference modules.py:250-254."""
    return _SubpelConv(in_ch, out_ch, r, 1)


def subpel_conv3x3(in_ch, out_ch, r=1):
    """compressai.layers.subpel_conv3x3 (image_model.py:95-109)."""
    return _SubpelConv(in_ch, out_ch, r, 3)


class UNet(nn.Module):
    """reference modules.py:257-300.  Skip concatenations are two-segment conv inputs, not copies."""

    def __init__(self, in_ch=64, out_ch=64):
        super().__init__()
        self.max_pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.conv1 = ConvBlockResidual(in_ch, 32)
        self.conv2 = ConvBlockResidual(32, 64)
        self.conv3 = ConvBlockResidual(64, 128)
        self.context_refine = nn.Sequential(*[ResBlock1(128, 0) for _ in range(4)])
        self.up3 = subpel_conv1x1(128, 64, 2)
        self.up_conv3 = ConvBlockResidual(128, 64)
        self.up2 = subpel_conv1x1(64, 32, 2)
        self.up_conv2 = ConvBlockResidual(64, out_ch)

    def run(self, x, out=None):
        x1 = self.conv1.run(x)
        x2 = self.conv2.run(ops.pool2x2(x1, "max"))
        x3 = self.conv3.run(ops.pool2x2(x2, "max"))
        for blk in self.context_refine:
            x3 = blk.run(x3)
        d3 = self.up_conv3.run([x2, self.up3.run(x3)])
        return self.up_conv2.run([x1, self.up2.run(d3)], out)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class RefineNet(nn.Module):
    """reference modules.py:303-315: x + refine(cat[x, ref_frame])."""

    def __init__(self, in_channel=2, hidden_channel=64, out_ch=2):
        super().__init__()
        self.refine = nn.Sequential(Conv2d(in_channel, hidden_channel, 3, 1, 1),
                                    ResBottleneckBlock(hidden_channel), ResBottleneckBlock(hidden_channel),
                                    ResBottleneckBlock(hidden_channel), Conv2d(hidden_channel, out_ch, 3, 1, 1))

    def run(self, x, ref, out=None, extra_residual=None):
        o = self.refine[0].run([x, ref])
        for i in (1, 2, 3):
            o = self.refine[i].run(o)
        return self.refine[4].run(o, out, res1=x, res2=extra_residual, out_dtype=x.dtype)

    def forward(self, x, ref_frame):
        return self.run(to_act(x), to_act(ref_frame)).nchw()


class _GateHead(nn.Sequential):
    """conv3x3 -> ResBlock1 -> conv3x3(->3) -> Sigmoid (reference modules.py:329-341, 412-417)."""

    def __init__(self, cin, hidden):
        super().__init__(Conv2d(cin, hidden, 3, 1, 1), ResBlock1(hidden), Conv2d(hidden, 3, 3, 1, 1), nn.Sigmoid())

    def run(self, x):
        return self[2].run(self[1].run(self[0].run(x)), act=N.ACT_SIGMOID, out_dtype=torch.float32)


class Reconstruction(nn.Module):
    """reference modules.py:318-358 -> (feature, recon)."""

    def __init__(self, in_ch=64, channel=64, out_ch=3, return_fea=True):
        super().__init__()
        self.return_fea = return_fea
        self.first_conv = Conv2d(in_ch, channel, 3, 1, 1)
        self.unet_1 = UNet(channel, channel)
        self.unet_2 = UNet(channel, channel)
        self.recon_conv1 = Conv2d(channel, out_ch, 3, 1, 1)
        self.recon_conv2 = Conv2d(channel, out_ch, 3, 1, 1)
        self.recon_conv3 = Conv2d(channel * 2, out_ch, 3, 1, 1)
        self.weight1 = _GateHead(channel, channel)
        self.weight2 = _GateHead(channel, channel)

    def run(self, segments):
        feature = self.first_conv.run(segments)
        f1, f2 = self.unet_1.run(feature), self.unet_2.run(feature)
        r1 = self.recon_conv1.run(f1, out_dtype=torch.float32)
        r2 = self.recon_conv2.run(f2, out_dtype=torch.float32)
        r3 = self.recon_conv3.run([f1, f2], out_dtype=torch.float32)
        w1, w2 = self.weight1.run(f1), self.weight2.run(f2)
        recon = ops.eltwise(N.EW_BLEND3, r1, r2, r3, w1, w2)
        return feature, recon

    def forward(self, x):
        feature, recon = self.run(to_act(x))
        return (feature.nchw(), recon.nchw()) if self.return_fea else recon.nchw()


class FeatureExtraction(nn.Module):
    """reference modules.py:361-374: x = conv1(x); x + rsb1(x) (rsb1 = 3 ReLU ResBlock1) -- both residual adds of
    the last block ride on its conv2 epilogue."""

    def __init__(self, in_ch=6, nf=64, k=3, s=1):
        super().__init__()
        self.conv1 = Conv2d(in_ch, nf, k, s, k // 2, bias=True)
        self.rsb1 = nn.Sequential(ResBlock1(nf, 0), ResBlock1(nf, 0), ResBlock1(nf, 0))

    def run(self, x, out=None):
        x = self.conv1.run(x)
        o = self.rsb1[1].run(self.rsb1[0].run(x))
        return self.rsb1[2].run(o, out, extra_residual=x)

    def forward(self, x):
        return self.run(to_act(x)).nchw()


class InterLayerPrediction(nn.Module):
    """reference modules.py:377-438 -> (up_out[32], predict[3]).
    `self.lrelu = nn.LeakyReLU(True)` is negative_slope = 1.0, i.e. the identity (modules.py:422): no kernel."""

    def __init__(self, in_ch=3, hidden=64, up_out=32, out_ch=3, fea_in=64, return_s3=False):
        super().__init__()
        self.return_s3 = return_s3
        self.in_conv = nn.Sequential(Conv2d(in_ch, hidden, 3, padding=1), ResBlock1(hidden))
        self.in_conv1 = nn.Sequential(Conv2d(fea_in, hidden, 3, padding=1), ResBlock1(hidden))
        self.d2s = nn.Sequential(nn.PixelShuffle(4), conv(16, 64, 3, 1))
        self.fea_convert = nn.Sequential(Conv2d(hidden, hidden, 3, padding=1), ResBlock1(hidden))
        self.fea_embd = nn.Sequential(Conv2d(2 * hidden, hidden, 3, padding=1), ResBlock1(hidden),
                                      ResBlock1(hidden, start_from_relu=False))
        self.fea_ext = nn.Sequential(Conv2d(hidden, hidden, 3, padding=1), ResBlock1(hidden), ResBlock1(hidden),
                                     Conv2d(hidden, 32, 3, padding=1))
        self.out_conv = Conv2d(up_out, out_ch, 3, stride=1, padding=1)
        self.weight = _GateHead(up_out, hidden)
        self.lrelu = nn.LeakyReLU(True)

    def run(self, ref, warped, mv, sm_fea_nchw, feature=None, up_out=None):
        if feature is None:
            fea = self.in_conv[1].run(self.in_conv[0].run(ref))
        else:
            fea = self.in_conv1[1].run(self.in_conv1[0].run(feature))
        fea = ops.warp(fea, mv)
        cur = self.d2s[1].run(ops.pixel_shuffle_nchw(sm_fea_nchw, 4))
        cur = self.fea_convert[1].run(self.fea_convert[0].run(cur))
        f3 = self.fea_embd[0].run([fea, cur])
        f3 = self.fea_embd[2].run(self.fea_embd[1].run(f3))
        up = self.fea_ext[0].run(f3)
        up = self.fea_ext[2].run(self.fea_ext[1].run(up))
        up = self.fea_ext[3].run(up, up_out)
        w = self.weight.run(up)
        cand = self.out_conv.run(up, out_dtype=torch.float32)
        predict = ops.eltwise(N.EW_BLEND2, warped, cand, d=w)
        return up, predict

    def forward(self, ref_frame, warped, mv, curr_fea, feature=None):
        up, predict = self.run(to_act(ref_frame), to_act(warped), to_act(mv), curr_fea,
                               None if feature is None else to_act(feature, ops.RT.storage))
        return up.nchw(), predict.nchw()
