"""MI355X-native semantic-layer feature codec (reference: semantic_layer.py:1146-1519): the producer
of the `sm_fea` input of the P-frame path.

  * `cFeatureCompress(in_ch=3, N=72)` (semantic_layer.py:1188-1519): a channel-conditional codec like the mv / residual codecs,
    additionally conditioned on a SUPPORT feature `fea` [B,256,H/4,W/4] (the semantic feature of an already decoded frame):
    analysis of cat[x, d2s(fea)], 8 slices of 9 channels, every parameter net sees cat[latent, fea_convert(fea), supports],
    synthesis to a 96-channel map at 1/4 resolution and a final stage g_s1(cat[., fea_convert1(fea)]) -> 256 channels.
    `nn.LeakyReLU(True)` throughout the reference = negative_slope 1.0 = identity, so the only
    non-linearities are the ResBottleneck blocks' LeakyReLU(0.01), the rounding and 0.5*tanh of the latent residual prediction.
  * `RefineNet1` (semantic_layer.py:1146-1185): temporal aggregation x + sum_l softmax_l(<n(x), n(supp_l)>) * x.
  * `StudentLayer1`: the first stage of `OursResNetStudentP` (semantic_layer.py:903-906, 1057-1094): codec -> RefineNet1 -> ReLU,
    with the reference's own timers / bpp bookkeeping (`enct`, `dect`, `bpp_loss`) that test_video.py:336-337,386-391 reads.
    The mmcv ResNet-50 stages behind it (layer2-4) and the teacher network are out of scope (they need mmcv).

Same attribute paths (=> state_dict keys), `forward / compress / decompress / update / load_state_dict / aux_loss` contracts
and bitstream layout as the reference.  Everything below the layer definitions is the shared scan machinery of
`image_model.ChannelSplitICIP2020ResB` (static workspaces, HIP graphs per dependency level, batched decode levels, one
device->host transfer per encode); the side feature lives in `latent_extra` channels behind the hyper-prior's means / scales.
"""
import math
import time

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from .entropy_models import EntropyBottleneck, GaussianConditional
from .image_model import ChannelSplitICIP2020ResB, _gelu_stack, conv3x3
from .modules import Conv2d, ResBottleneckBlock, conv, subpel_conv3x3
from .ops import Act, to_act

F32 = torch.float32


def _identity():
    """`nn.LeakyReLU(True)`: the positional argument is negative_slope, True == 1.0 -- the identity (keeps the reference's
    Sequential indices, holds no parameters)."""
    return nn.LeakyReLU(True)


class cFeatureCompress(ChannelSplitICIP2020ResB):
    _stack_act = (N.ACT_NONE, 0.0)         # "LeakyReLU(True)" between the convs of the cc / lrp stacks
    latent_extra = 64                      # fea_convert(fea) behind the hyper-prior's means / scales

    def __init__(self, in_ch=3, N=72):
        nn.Module.__init__(self)
        self.N = self.M = N
        self.num_slices, self.max_support_slices = 6, 3
        if N == 64 or N == 72:
            self.num_slices, self.max_support_slices = 8, 4
        d = self.N // self.num_slices
        if d * self.num_slices != self.N:
            raise ValueError(f"Slice do not evenly divide latent depth ({self.N}/{self.num_slices})")
        self.slice_depth = d
        k = self.max_support_slices
        self.g_a = nn.Sequential(conv(in_ch + 64, 128), _identity(), ResBottleneckBlock(128), conv(128, 128), _identity(),
                                 ResBottleneckBlock(128), conv(128, 96), _identity(), ResBottleneckBlock(96), conv(96, N))
        self.g_s = nn.Sequential(subpel_conv3x3(N, 96, 2), _identity(), ResBottleneckBlock(96), subpel_conv3x3(96, 96, 2), _identity(),
                                 ResBottleneckBlock(96), Conv2d(96, 96, 3, 1, 1))
        self.h_a = nn.Sequential(conv3x3(N, N), _identity(), conv3x3(N, N, stride=2), _identity(), conv3x3(N, N), _identity(),
                                 conv3x3(N, N, stride=2))
        hyper = lambda: nn.Sequential(subpel_conv3x3(N, N, 2), _identity(), conv3x3(N, N), _identity(),
                                      subpel_conv3x3(N, N, 2), _identity(), conv3x3(N, N))
        self.h_mean_s, self.h_scale_s = hyper(), hyper()
        self.cc_mean_transforms = nn.ModuleList(_gelu_stack([N + d * min(i, k) + 64, N, N, N, d], act=_identity) for i in range(self.num_slices))
        self.cc_scale_transforms = nn.ModuleList(_gelu_stack([N + d * min(i, k) + 64, N, N, N, d], act=_identity) for i in range(self.num_slices))
        self.lrp_transforms = nn.ModuleList(_gelu_stack([N + d * min(i + 1, k + 1) + 64, N, N, d], act=_identity) for i in range(self.num_slices))
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)
        self.fea_convert = nn.Sequential(conv(256, 128), _identity(), conv(128, 64))
        self.fea_convert1 = nn.Sequential(conv(256, 128, 3, 1), _identity(), conv(128, 64, 3, 1))
        self.d2s = nn.Sequential(nn.PixelShuffle(4), conv(16, 64, 3, 1))
        self.g_s1 = nn.Sequential(conv(96 + 64, 256, 3, 1), _identity(), ResBottleneckBlock(256), ResBottleneckBlock(256))

    # ---- transforms (identity "activations") ----------------------------------------------------------------------
    def _g_a(self, segments, out=None):
        x = segments
        last = len(self.g_a) - 1
        for i, m in enumerate(self.g_a):
            if isinstance(m, Conv2d):
                x = m.run(x, out if i == last else None, out_dtype=F32 if i == last else None)   # the latent y stays fp32
            elif isinstance(m, ResBottleneckBlock):
                x = m.run(x)
        return x

    def _g_s(self, y_hat, out=None, out_dtype=None):
        x = self.g_s[0].run(y_hat)
        x = self.g_s[2].run(x)
        x = self.g_s[3].run(x)
        x = self.g_s[5].run(x)
        return self.g_s[6].run(x, out, out_dtype=out_dtype)

    def _h_a(self, y):
        x = self.h_a[0].run(y, out_dtype=F32)
        x = self.h_a[2].run(x, out_dtype=F32)
        x = self.h_a[4].run(x, out_dtype=F32)
        return self.h_a[6].run(x, out_dtype=F32)

    def _h_s(self, net, z_hat, out=None):
        x = net[0].run(z_hat, out_dtype=F32)
        x = net[2].run(x, out_dtype=F32)
        x = net[4].run(x, out_dtype=F32)
        return net[6].run(x, out, out_dtype=F32)

    # ---- the side feature -------------------------------------------------------------------------------------------
    def _analysis_input(self, x, fea):
        """cat[x, d2s(fea)] as a segment list (semantic_layer.py:1319,1381)."""
        return [to_act(x), self.d2s[1].run(ops.pixel_shuffle_nchw(fea, 4))]

    def _stage_support(self, ws, fea):
        """supp_sm = fea_convert(fea) (semantic_layer.py:1334,1392,1448) written behind the hyper-prior's means AND scales of the
        scan workspace: cat[latent_means, supp_sm, ...] / cat[latent_scales, supp_sm, ...] are then single NHWC segments."""
        Mc, B = self.M, ws.B
        t = self.fea_convert[0].run(to_act(fea))
        self.fea_convert[2].run(t, Act(ws.pair.buf[:B], Mc, 64), out_dtype=F32)
        ops.eltwise(N.EW_COPY, Act(ws.pair.buf[:B], Mc, 64), out=Act(ws.pair.buf[B:], Mc, 64))

    def _final(self, x_hat1, fea):
        """g_s1(cat[x_hat1, fea_convert1(fea)]) (semantic_layer.py:1373,1487)."""
        f1 = self.fea_convert1[2].run(self.fea_convert1[0].run(to_act(fea)))
        x = self.g_s1[0].run([x_hat1, f1])
        return self.g_s1[3].run(self.g_s1[2].run(x))

    # ---- reference-style API (NCHW tensors) ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, fea=None):
        """semantic_layer.py:1318-1379 -> {"x_hat", "likelihoods": {"y", "z"}, "y"} (estimate path, eval-mode rounding)."""
        segs = self._analysis_input(x, fea)
        x0 = segs[0]
        ws = self._plan(x0.B, x0.H // 16, x0.W // 16, x0.device)
        self._stage_support(ws, fea)
        y = self._g_a(segs, ws.y)
        z = self._h_a(y)
        dev = y.device
        z_lik = ops.new_act(z.B, z.H, z.W, z.C, F32, dev)
        self.entropy_bottleneck.quantize(z, False, ws.z_hat, z_lik)
        self._h_s(self.h_scale_s, ws.z_hat, ws.scales)
        self._h_s(self.h_mean_s, ws.z_hat, ws.means)
        y_lik = ops.new_act(y.B, y.H, y.W, y.C, F32, dev)
        d = self.slice_depth
        for i in range(self.num_slices):
            mu, sc = self._slice_params(i, ws.means, ws.scales, ws.y_hat)
            ops.gc_likelihood(y.ch(i * d, d), mu, sc, ws.y_hat.ch(i * d, d), y_lik.ch(i * d, d))
            self._lrp(i, ws.means, ws.y_hat)
        x_hat = self._final(self._g_s(ws.y_hat), fea)
        return {"x_hat": x_hat.nchw(), "likelihoods": {"y": y_lik.nchw(), "z": z_lik.nchw()}, "y": y.nchw().clone()}

    @torch.no_grad()
    def compress(self, x, fea):
        """semantic_layer.py:1381-1437 -> {"strings": [[y_bytes], [z_bytes...]], "shape": (h/4, w/4), "y": latent}."""
        segs = self._analysis_input(x, fea)
        x0 = segs[0]
        ws = self._plan(x0.B, x0.H // 16, x0.W // 16, x0.device)
        self._stage_support(ws, fea)
        enc, _ = self.run_compress(segs)
        out = dict(enc)
        out["y"] = ws.y.nchw().clone()
        return out

    @torch.no_grad()
    def decompress(self, strings, shape, fea):
        """semantic_layer.py:1439-1489 -> {"x_hat": [B,256,H/4,W/4]}."""
        dev = fea.device
        ws = self._plan(len(strings[1]), int(shape[0]) * 4, int(shape[1]) * 4, dev)
        self._stage_support(ws, fea)
        x_hat1 = self.run_decompress(strings, shape, dev)
        return {"x_hat": self._final(x_hat1, fea).nchw()}


class RefineNet1(nn.Module):
    """semantic_layer.py:1146-1185: refine = conv3x3(256,256) -> identity -> conv3x3(256,256); per pixel, the L2-normalised
    refined feature of x is correlated (channel mean or sum) with the L2-normalised refined support features, soft-maxed over
    the supports, and x + sum_l (x * w_l) is returned."""

    def __init__(self):
        super().__init__()
        self.refine = nn.Sequential(Conv2d(256, 256, 3, 1, 1), _identity(), Conv2d(256, 256, 3, 1, 1))

    def _refine(self, a):
        return self.refine[2].run(self.refine[0].run(a), out_dtype=F32)

    @torch.no_grad()
    def forward(self, x, supp, mean=True, relu=False):
        """`relu=True` folds the ReLU that follows in OursResNetStudentP (semantic_layer.py:1082) into the same kernel."""
        batch = x.shape[0]
        n_supp = supp.shape[0] // batch
        # the reference takes supp[ll * batch] for every item kk of the batch (semantic_layer.py:1173-1174)
        sel = supp[[ll * batch for ll in range(n_supp)]].contiguous()
        rs = self._refine(to_act(sel))
        outs = []
        for kk in range(batch):
            xk = to_act(x[kk:kk + 1].contiguous(), F32)
            outs.append(ops.refine_aggregate(xk, self._refine(xk), rs, mean, relu).nchw())
        return torch.cat(outs, 0)


class StudentLayer1(nn.Module):
    """Stage 1 of `OursResNetStudentP(N=72)` (semantic_layer.py:903-906 and the `i == 0` branch of forward, :1057-1094):
    `layer1` = the feature codec, `refine` = RefineNet1, `act` = ReLU.  forward(x, supp, mean, encode) -> (sm_fea, y) where
    `sm_fea` [B,256,H/4,W/4] is what test_video.py:336,342 passes to DeepSVC as `sm[0]`; `bpp_loss`, `enct`, `dect`, `TSFt`
    are set like the reference's attributes (test_video.py:337,386-391 reads them)."""

    def __init__(self, N=72):
        super().__init__()
        self.layer1 = cFeatureCompress(N=N)
        self.refine = RefineNet1()
        self.act = nn.ReLU(inplace=True)
        self.bpp_loss, self.enct, self.dect, self.TSFt = 0.0, 0.0, 0.0, 0.0

    def update(self, force=False):
        return self.layer1.update(force=force)

    @torch.no_grad()
    def forward(self, x, supp=None, mean=False, encode=False):
        Nb, _, H, W = x.size()
        num_pixels = Nb * H * W
        sync = torch.cuda.synchronize
        if encode:                                                      # semantic_layer.py:1064-1082
            sync(); t0 = time.perf_counter()
            out_enc = self.layer1.compress(x, supp[-Nb:])
            sync(); self.enct = time.perf_counter() - t0
            sync(); t0 = time.perf_counter()
            out_dec = self.layer1.decompress(out_enc["strings"], out_enc["shape"], supp[-Nb:])
            sync(); self.dect = time.perf_counter() - t0
            self.bpp_loss = sum(len(s[0]) for s in out_enc["strings"]) * 8.0 / num_pixels
            y, xh = out_enc["y"], out_dec["x_hat"]
            sync(); t0 = time.perf_counter()
            xh = self.refine(xh, supp, mean=mean, relu=True)
            self.TSFt = time.perf_counter() - t0
        else:                                                           # :1084-1093
            out = self.layer1(x, fea=supp[-Nb:])
            self.bpp_loss = sum(ops.reduce("sumlog", to_act(l)) / (-math.log(2) * num_pixels) for l in out["likelihoods"].values())
            y, xh = out["y"], out["x_hat"]
            xh = self.refine(xh, supp, mean=mean, relu=True)
        return xh, y
