"""Oracle restatement of semantic_layer.py:cFeatureCompress / RefineNet1 and of stage 1 of OursResNetStudentP (the
semantic-layer feature codec).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PyTorch-CPU fp32, functional over a state_dict.  Tensor arithmetic pinned
against the imported reference classes (tests/golden/fcodec_ref_*.npz); the entropy arithmetic it calls (oracle.entropy)
is "parity unpinned" exactly as for the other codecs.  `nn.LeakyReLU(True)` in the reference is the identity
(negative_slope = True = 1.0), so no activation appears between the convs below.
"""
import math

import torch
import torch.nn.functional as F

from . import entropy as E
from . import nets as N
from .nets import P

MAX_SUPPORT = 4          # semantic_layer.py:1195-1196 (N = 72: 8 slices, 4 supports)
NUM_SLICES = 8


def g_a(p, x):
    """semantic_layer.py:1203-1214: conv5x5 s2 at 0/3/6/9, ResBottleneck at 2/5/8."""
    for i in (0, 3, 6):
        x = N.res_bottleneck(p.sub(f"g_a.{i + 2}"), N.conv(p.sub(f"g_a.{i}"), x, stride=2))
    return N.conv(p.sub("g_a.9"), x, stride=2)


def g_s(p, y_hat):
    """semantic_layer.py:1216-1224: subpel3x3(2), RBB, subpel3x3(2), RBB, conv3x3."""
    x = N.res_bottleneck(p.sub("g_s.2"), F.pixel_shuffle(N.conv(p.sub("g_s.0.0"), y_hat), 2))
    x = N.res_bottleneck(p.sub("g_s.5"), F.pixel_shuffle(N.conv(p.sub("g_s.3.0"), x), 2))
    return N.conv(p.sub("g_s.6"), x)


def h_a(p, y):
    """semantic_layer.py:1226-1234: conv3x3, conv3x3 s2, conv3x3, conv3x3 s2 (identity "activations")."""
    x = N.conv(p.sub("h_a.0"), y)
    x = N.conv(p.sub("h_a.2"), x, stride=2)
    x = N.conv(p.sub("h_a.4"), x)
    return N.conv(p.sub("h_a.6"), x, stride=2)


def h_s(p, name, z_hat):
    """h_mean_s / h_scale_s (semantic_layer.py:1236-1254)."""
    q = p.sub(name)
    x = F.pixel_shuffle(N.conv(q.sub("0.0"), z_hat), 2)
    x = N.conv(q.sub("2"), x)
    x = F.pixel_shuffle(N.conv(q.sub("4.0"), x), 2)
    return N.conv(q.sub("6"), x)


def stack(p, x, n):
    """cc_mean / cc_scale (n = 4 convs) and lrp (n = 3) stacks (semantic_layer.py:1256-1290)."""
    for i in range(n):
        x = N.conv(p.sub(str(2 * i)), x)
    return x


def fea_convert(p, fea):
    """semantic_layer.py:1295-1299: two conv5x5 s2."""
    return N.conv(p.sub("fea_convert.2"), N.conv(p.sub("fea_convert.0"), fea, stride=2), stride=2)


def final(p, x_hat1, fea):
    """g_s1(cat[x_hat1, fea_convert1(fea)]) (semantic_layer.py:1301-1305,1314-1318 region; call :1373,1487)."""
    f1 = N.conv(p.sub("fea_convert1.2"), N.conv(p.sub("fea_convert1.0"), fea))
    x = N.conv(p.sub("g_s1.0"), torch.cat([x_hat1, f1], 1))
    return N.res_bottleneck(p.sub("g_s1.3"), N.res_bottleneck(p.sub("g_s1.2"), x))


def analysis_input(p, x, fea):
    """cat[x, d2s(fea)] (semantic_layer.py:1307-1310,1319)."""
    return torch.cat([x, N.conv(p.sub("d2s.1"), F.pixel_shuffle(fea, 4))], 1)


def _slice_params(p, i, latent_means, latent_scales, supp_sm, y_hat_slices, y_shape):
    support = y_hat_slices[:MAX_SUPPORT]
    mean_support = torch.cat([latent_means, supp_sm] + support, 1)
    mu = stack(p.sub(f"cc_mean_transforms.{i}"), mean_support, 4)[:, :, : y_shape[0], : y_shape[1]]
    scale_support = torch.cat([latent_scales, supp_sm] + support, 1)
    scale = stack(p.sub(f"cc_scale_transforms.{i}"), scale_support, 4)[:, :, : y_shape[0], : y_shape[1]]
    return mean_support, mu, scale


def _lrp(p, i, mean_support, y_hat_slice):
    lrp = stack(p.sub(f"lrp_transforms.{i}"), torch.cat([mean_support, y_hat_slice], 1), 3)
    return y_hat_slice + 0.5 * torch.tanh(lrp)


def forward(p, x, fea, taps=None):
    """semantic_layer.py:1318-1379 -> {"x_hat", "likelihoods", "y"}."""
    y = g_a(p, analysis_input(p, x, fea))
    y_shape = y.shape[2:]
    z = h_a(p, y)
    _, z_lik = E.eb_forward(p.sd, p.prefix + "entropy_bottleneck.", z)
    med = E.eb_medians(p.sd, p.prefix + "entropy_bottleneck.").reshape(1, -1, 1, 1)
    z_hat = torch.round(z - med) + med
    latent_scales, latent_means = h_s(p, "h_scale_s", z_hat), h_s(p, "h_mean_s", z_hat)
    supp_sm = fea_convert(p, fea)
    y_hat_slices, liks = [], []
    if taps is not None:
        taps.update(y=y, z=z, latent_scales=latent_scales, latent_means=latent_means, supp_sm=supp_sm, mu=[], scale=[])
    for i, y_slice in enumerate(y.chunk(NUM_SLICES, 1)):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, supp_sm, y_hat_slices, y_shape)
        if taps is not None:
            taps["mu"].append(mu)
            taps["scale"].append(scale)
        liks.append(E.gc_likelihood(torch.round(y_slice - mu) + mu, scale, mu))
        y_hat_slices.append(_lrp(p, i, mean_support, torch.round(y_slice - mu) + mu))
    y_hat = torch.cat(y_hat_slices, 1)
    x_hat1 = g_s(p, y_hat)
    if taps is not None:
        taps.update(y_hat=y_hat, x_hat1=x_hat1)
    return {"x_hat": final(p, x_hat1, fea), "likelihoods": {"y": torch.cat(liks, 1), "z": z_lik}, "y": y}


def tables(p):
    """update() (semantic_layer.py:1505-1510)."""
    return {"eb": E.eb_tables(p.sd, p.prefix + "entropy_bottleneck."), "gc": E.gc_tables()}


def compress(p, tabs, x, fea, taps=None):
    """semantic_layer.py:1381-1437."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    y = g_a(p, analysis_input(p, x, fea))
    y_shape = y.shape[2:]
    z = h_a(p, y)
    z_strings = E.eb_compress(p.sd, eb_prefix, tabs["eb"], z)
    z_hat = E.eb_decompress(p.sd, eb_prefix, tabs["eb"], z_strings, z.shape[-2:])
    latent_scales, latent_means = h_s(p, "h_scale_s", z_hat), h_s(p, "h_mean_s", z_hat)
    supp_sm = fea_convert(p, fea)
    gc = tabs["gc"]
    y_hat_slices, symbols, indexes = [], [], []
    for i, y_slice in enumerate(y.chunk(NUM_SLICES, 1)):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, supp_sm, y_hat_slices, y_shape)
        idx = E.gc_build_indexes(scale, gc["scale_table"])
        q = torch.round(y_slice - mu).to(torch.int32)
        symbols.append(q.reshape(-1))
        indexes.append(idx.reshape(-1))
        y_hat_slices.append(_lrp(p, i, mean_support, q + mu))
    symbols, indexes = torch.cat(symbols).numpy(), torch.cat(indexes).numpy()
    if taps is not None:
        taps.update(symbols=symbols, indexes=indexes, z=z)
    y_string = E.rans_encode(symbols, indexes, gc["cdf"], gc["cdf_length"], gc["offset"])
    return {"strings": [[y_string], z_strings], "shape": tuple(z.shape[-2:]), "y": y}


def decompress(p, tabs, strings, shape, fea, taps=None):
    """semantic_layer.py:1439-1489."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    z_hat = E.eb_decompress(p.sd, eb_prefix, tabs["eb"], strings[1], shape)
    latent_scales, latent_means = h_s(p, "h_scale_s", z_hat), h_s(p, "h_mean_s", z_hat)
    supp_sm = fea_convert(p, fea)
    y_shape = [z_hat.shape[2] * 4, z_hat.shape[3] * 4]
    gc = tabs["gc"]
    dec = E.RansDecoder(strings[0][0])
    y_hat_slices = []
    for i in range(NUM_SLICES):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, supp_sm, y_hat_slices, y_shape)
        idx = E.gc_build_indexes(scale, gc["scale_table"])
        rv = dec.decode(idx.reshape(-1).numpy(), gc["cdf"], gc["cdf_length"], gc["offset"])
        rv = torch.from_numpy(rv.astype("float32")).reshape(1, -1, y_shape[0], y_shape[1])
        y_hat_slices.append(_lrp(p, i, mean_support, rv + mu))
    y_hat = torch.cat(y_hat_slices, 1)
    x_hat1 = g_s(p, y_hat)
    if taps is not None:
        taps.update(y_hat=y_hat, x_hat1=x_hat1)
    return {"x_hat": final(p, x_hat1, fea)}


def refine_net1(p, x, supp, mean=True):
    """RefineNet1.forward (semantic_layer.py:1163-1185)."""
    refine = lambda t: N.conv(p.sub("refine.2"), N.conv(p.sub("refine.0"), t))
    batch = x.shape[0]
    n_supp = supp.shape[0] // batch
    results = []
    for kk in range(batch):
        x1 = refine(x[kk].unsqueeze(0))
        x1 = x1 / x1.norm(p=2, dim=1, keepdim=True)
        s = refine(torch.cat([supp[ll * batch].unsqueeze(0) for ll in range(n_supp)], 0))
        s = s / s.norm(p=2, dim=1, keepdim=True)
        w = torch.mean(x1 * s, dim=1, keepdim=True) if mean else torch.sum(x1 * s, dim=1, keepdim=True)
        w = w.softmax(dim=0)
        results.append(x[kk].unsqueeze(0) + torch.sum(x[kk].unsqueeze(0) * w, dim=0, keepdim=True))
    return torch.cat(results, 0)


class StudentLayer1Oracle:
    """Stage 1 of OursResNetStudentP (semantic_layer.py:903-906, 1057-1094) over a state_dict with the reference's key
    prefixes `layer1.` / `refine.`."""

    def __init__(self, state_dict):
        self.sd = {k: v.detach().to(torch.float32).cpu() if v.is_floating_point() else v.detach().cpu() for k, v in state_dict.items()}
        self.root = P(self.sd)
        self.codec = self.root.sub("layer1")
        self.tabs = None

    def update(self, force=False):
        if self.tabs is None or force:
            self.tabs = tables(self.codec)
            return True
        return False

    @torch.no_grad()
    def forward(self, x, supp, mean=False, encode=False, taps=None):
        Nb, _, H, W = x.shape
        num_pixels = Nb * H * W
        fea = supp[-Nb:]
        if encode:
            self.update()
            enc = compress(self.codec, self.tabs, x, fea, taps)
            dec = decompress(self.codec, self.tabs, enc["strings"], enc["shape"], fea, taps)
            self.bpp_loss = sum(len(s[0]) for s in enc["strings"]) * 8.0 / num_pixels
            self.strings = enc["strings"]
            y, xh = enc["y"], dec["x_hat"]
        else:
            out = forward(self.codec, x, fea, taps)
            self.bpp_loss = float(sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in out["likelihoods"].values()))
            y, xh = out["y"], out["x_hat"]
        if taps is not None:
            taps["x_hat"] = xh
        xh = refine_net1(self.root.sub("refine"), xh, supp, mean=mean)
        return torch.relu(xh), y
