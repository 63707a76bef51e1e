"""Oracle restatement of the reference's building blocks (modules.py), PyTorch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function takes a parameter accessor
``P`` (a state_dict + key prefix; keys are the reference's own attribute paths) and NCHW
tensors, and cites the reference lines it follows.  Pinned against the imported reference by
tests/golden/make_golden.py -> tests/golden/*.npz.
"""
import torch
import torch.nn.functional as F


class P:
    """state_dict accessor: P(sd, 'MC.')('in_conv.0.weight') -> sd['MC.in_conv.0.weight']."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def sub(self, name):
        return P(self.sd, f"{self.prefix}{name}.")

    def has(self, name):
        return (self.prefix + name) in self.sd


def conv(p, x, stride=1):
    """nn.Conv2d with 'same'-style padding k//2 (every conv on the path is built that way:
    modules.py:127-131,176-178,200-204,237-242,307-311,322-340,364,382-418; compressai conv/conv3x3)."""
    w = p("weight")
    return F.conv2d(x, w, p("bias"), stride=stride, padding=w.shape[-1] // 2)


def deconv(p, x):
    """compressai deconv(k=5,s=2) = ConvTranspose2d(k, stride 2, padding k//2, output_padding 1)
    (image_model.py:69-81)."""
    w = p("weight")
    return F.conv_transpose2d(x, w, p("bias"), stride=2, padding=w.shape[-1] // 2, output_padding=1)


def lrelu(x, slope=0.01):
    return F.leaky_relu(x, slope)


def warp(x, flow):
    """torch_warp (modules.py:25-62): backward warp, bilinear, border clamp, align_corners=True.
    Base grid linspace(-1,1) + flow / ((size-1)/2); flow channel 0 = x displacement."""
    B, _, H, W = flow.shape
    gx = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).expand(B, -1, H, -1)
    gy = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).expand(B, -1, -1, W)
    nf = torch.cat([flow[:, 0:1] / ((x.size(3) - 1.0) / 2.0), flow[:, 1:2] / ((x.size(2) - 1.0) / 2.0)], 1)
    grid = (torch.cat([gx, gy], 1) + nf).permute(0, 2, 3, 1)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=True)


def upsample2x(x):
    """bilinearupsacling (modules.py:107-112)."""
    return F.interpolate(x, (x.size(2) * 2, x.size(3) * 2), mode="bilinear", align_corners=False)


def me_basic(p, x):
    """MEBasic (modules.py:123-139): five 7x7 convs, ReLU between."""
    for i in range(1, 5):
        x = F.relu(conv(p.sub(f"conv{i}"), x))
    return conv(p.sub("conv5"), x)


def spynet(p, im1, im2, levels=4):
    """ME_Spynet (modules.py:142-170): im2 is warped toward im1, coarse to fine."""
    pyr1, pyr2 = [im1], [im2]
    for _ in range(levels - 1):
        pyr1.append(F.avg_pool2d(pyr1[-1], 2, 2))
        pyr2.append(F.avg_pool2d(pyr2[-1], 2, 2))
    B, _, h, w = pyr2[-1].shape
    flow = torch.zeros(B, 2, h // 2, w // 2, dtype=im1.dtype)
    for lvl in range(levels):
        up = upsample2x(flow) * 2.0
        k = levels - 1 - lvl
        flow = up + me_basic(p.sub(f"moduleBasic.{lvl}"), torch.cat([pyr1[k], warp(pyr2[k], up), up], 1))
    return flow


def res_bottleneck(p, x, slope=0.01):
    """ResBottleneckBlock (modules.py:173-189 == image_model.py:22-38)."""
    act = (lambda t: F.relu(t)) if slope < 1e-4 else (lambda t: lrelu(t, slope))
    o = act(conv(p.sub("conv1"), x))
    o = act(conv(p.sub("conv2"), o))
    return x + conv(p.sub("conv3"), o)


def res_block1(p, x, slope=0.01, start_from_relu=True, end_with_relu=False):
    """ResBlock1 (modules.py:192-214); slope<1e-4 -> ReLU; activation hits the block INPUT first."""
    act = (lambda t: F.relu(t)) if slope < 1e-4 else (lambda t: lrelu(t, slope))
    o = act(x) if start_from_relu else x
    o = act(conv(p.sub("conv1"), o))
    o = conv(p.sub("conv2"), o)
    if end_with_relu:
        o = act(o)
    return x + o


def se_layer(p, x):
    """SELayer (modules.py:217-230): global mean -> FC(no bias) -> ReLU -> FC -> sigmoid -> scale."""
    y = x.mean(dim=(-1, -2))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, p("fc.0.weight"))), p("fc.2.weight")))
    return x * y[:, :, None, None]


def conv_block_residual(p, x):
    """ConvBlockResidual (modules.py:233-247): up_dim(x) + SE(conv(lrelu(conv(x))))."""
    o = conv(p.sub("conv.2"), lrelu(conv(p.sub("conv.0"), x)))
    return conv(p.sub("up_dim"), x) + se_layer(p.sub("conv.3"), o)


def subpel1x1(p, x, r):
    """subpel_conv1x1 (modules.py:250-254)."""
    return F.pixel_shuffle(conv(p.sub("0"), x), r)


def unet(p, x):
    """UNet (modules.py:257-300)."""
    x1 = conv_block_residual(p.sub("conv1"), x)
    x2 = conv_block_residual(p.sub("conv2"), F.max_pool2d(x1, 2, 2))
    x3 = conv_block_residual(p.sub("conv3"), F.max_pool2d(x2, 2, 2))
    for i in range(4):
        x3 = res_block1(p.sub(f"context_refine.{i}"), x3, slope=0)
    d3 = conv_block_residual(p.sub("up_conv3"), torch.cat([x2, subpel1x1(p.sub("up3"), x3, 2)], 1))
    d2 = conv_block_residual(p.sub("up_conv2"), torch.cat([x1, subpel1x1(p.sub("up2"), d3, 2)], 1))
    return d2


def refine_net(p, x, ref):
    """RefineNet (modules.py:303-315)."""
    o = conv(p.sub("refine.0"), torch.cat([x, ref], 1))
    for i in (1, 2, 3):
        o = res_bottleneck(p.sub(f"refine.{i}"), o)
    return x + conv(p.sub("refine.4"), o)


def _weight_head(p, x):
    """conv -> ResBlock1 -> conv(->3) -> sigmoid (modules.py:329-341, 412-417)."""
    o = conv(p.sub("0"), x)
    o = res_block1(p.sub("1"), o)
    return torch.sigmoid(conv(p.sub("2"), o))


def reconstruction(p, x):
    """Reconstruction (modules.py:318-358) -> (feature, recon)."""
    feat = conv(p.sub("first_conv"), x)
    f1, f2 = unet(p.sub("unet_1"), feat), unet(p.sub("unet_2"), feat)
    r1, r2 = conv(p.sub("recon_conv1"), f1), conv(p.sub("recon_conv2"), f2)
    r3 = conv(p.sub("recon_conv3"), torch.cat([f1, f2], 1))
    w1, w2 = _weight_head(p.sub("weight1"), f1), _weight_head(p.sub("weight2"), f2)
    return feat, w1 * r1 + w2 * r2 + (1 - w1 - w2) * r3


def feature_extraction(p, x):
    """FeatureExtraction (modules.py:361-374): note the double residual."""
    x = conv(p.sub("conv1"), x)
    o = x
    for i in range(3):
        o = res_block1(p.sub(f"rsb1.{i}"), o, slope=0)
    return x + o


def inter_layer_prediction(p, ref, warped, mv, sm_fea, feature=None):
    """InterLayerPrediction (modules.py:377-438) -> (up_out[32], predict[3]).
    self.lrelu = nn.LeakyReLU(True) has negative_slope 1.0, i.e. identity (modules.py:422)."""
    if feature is None:
        fea = res_block1(p.sub("in_conv.1"), conv(p.sub("in_conv.0"), ref))
    else:
        fea = res_block1(p.sub("in_conv1.1"), conv(p.sub("in_conv1.0"), feature))
    fea = warp(fea, mv)
    cf = conv(p.sub("d2s.1"), F.pixel_shuffle(sm_fea, 4))
    cf = res_block1(p.sub("fea_convert.1"), conv(p.sub("fea_convert.0"), cf))
    f3 = conv(p.sub("fea_embd.0"), torch.cat([fea, cf], 1))
    f3 = res_block1(p.sub("fea_embd.1"), f3)
    f3 = res_block1(p.sub("fea_embd.2"), f3, start_from_relu=False)
    up = conv(p.sub("fea_ext.0"), f3)
    up = res_block1(p.sub("fea_ext.1"), up)
    up = res_block1(p.sub("fea_ext.2"), up)
    up = conv(p.sub("fea_ext.3"), up)
    w = _weight_head(p.sub("weight"), up)
    return up, w * warped + (1 - w) * conv(p.sub("out_conv"), up)


def pad64(x, m=64):
    """pad (modules.py:76-89): centred zero pad to a multiple of m."""
    h, w = x.size(2), x.size(3)
    H, W = (h + m - 1) // m * m, (w + m - 1) // m * m
    l, t = (W - w) // 2, (H - h) // 2
    return F.pad(x, (l, W - w - l, t, H - h - t))


def crop(x, size):
    """crop (modules.py:92-104)."""
    H, W = x.size(2), x.size(3)
    h, w = size
    l, t = (W - w) // 2, (H - h) // 2
    return F.pad(x, (-l, -(W - w - l), -t, -(H - h - t)))
