"""Oracle restatement of image_model.py:ICIP2020ResB (the I-frame codec).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PyTorch-CPU fp32, functional over a state_dict.  Tensor arithmetic pinned
against the imported reference (tests/golden/icip_ref_*.npz); the entropy arithmetic it calls (oracle.entropy) is
"parity unpinned" exactly as for the P-frame codecs.
"""
import torch
import torch.nn.functional as F

from . import entropy as E
from . import nets as N
from .nets import P

NUM_SLICES = 10          # image_model.py:336
MAX_SUPPORT = 5          # image_model.py:337


def g_a(p, x):
    """image_model.py:344-358: conv5x5 s2, then 3 x (3 ResBottleneck, conv5x5 s2)."""
    for i in range(13):
        q = p.sub(f"g_a.{i}")
        x = N.conv(q, x, stride=2) if i % 4 == 0 else N.res_bottleneck(q, x)
    return x


def g_s(p, x):
    """image_model.py:360-374."""
    for i in range(13):
        q = p.sub(f"g_s.{i}")
        x = N.deconv(q, x) if i % 4 == 0 else N.res_bottleneck(q, x)
    return x


def h_a(p, x):
    """image_model.py:376-382: conv3x3, LeakyReLU, conv5x5 s2, LeakyReLU, conv5x5 s2."""
    x = N.lrelu(N.conv(p.sub("h_a.0"), x))
    x = N.lrelu(N.conv(p.sub("h_a.2"), x, stride=2))
    return N.conv(p.sub("h_a.4"), x, stride=2)


def h_s(p, name, z_hat):
    """h_mean_s / h_scale_s (image_model.py:384-398): deconv, LeakyReLU, deconv, LeakyReLU, conv3x3."""
    q = p.sub(name)
    x = N.lrelu(N.deconv(q.sub("0"), z_hat))
    x = N.lrelu(N.deconv(q.sub("2"), x))
    return N.conv(q.sub("4"), x)


def stack3(p, x):
    """cc_mean / cc_scale / lrp stack (image_model.py:408-436): conv3x3, LeakyReLU, conv3x3, LeakyReLU, conv3x3."""
    x = N.lrelu(N.conv(p.sub("0"), x))
    x = N.lrelu(N.conv(p.sub("2"), x))
    return N.conv(p.sub("4"), x)


def _slice_params(p, i, latent_means, latent_scales, y_hat_slices, y_shape):
    support = y_hat_slices[:MAX_SUPPORT]
    mean_support = torch.cat([latent_means] + support, 1)
    mu = stack3(p.sub(f"cc_mean_transforms.{i}"), mean_support)[:, :, : y_shape[0], : y_shape[1]]
    scale_support = torch.cat([latent_scales] + support, 1)
    scale = stack3(p.sub(f"cc_scale_transforms.{i}"), scale_support)[:, :, : y_shape[0], : y_shape[1]]
    return mean_support, mu, scale


def _lrp(p, i, mean_support, y_hat_slice):
    lrp = stack3(p.sub(f"lrp_transforms.{i}"), torch.cat([mean_support, y_hat_slice], 1))
    return y_hat_slice + 0.5 * torch.tanh(lrp)


def forward(p, x, taps=None):
    """image_model.py:441-489 -> {"x_hat", "likelihoods": {"y", "z"}}."""
    y = g_a(p, x)
    y_shape = y.shape[2:]
    z = h_a(p, y)
    _, z_lik = E.eb_forward(p.sd, p.prefix + "entropy_bottleneck.", z)
    med = E.eb_medians(p.sd, p.prefix + "entropy_bottleneck.").reshape(1, -1, 1, 1)
    z_hat = torch.round(z - med) + med
    latent_scales = h_s(p, "h_scale_s", z_hat)
    latent_means = h_s(p, "h_mean_s", z_hat)
    y_hat_slices, liks = [], []
    if taps is not None:
        taps.update(y=y, z=z, z_hat=z_hat, latent_scales=latent_scales, latent_means=latent_means, mu=[], scale=[])
    for i, y_slice in enumerate(y.chunk(NUM_SLICES, 1)):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, y_hat_slices, y_shape)
        if taps is not None:
            taps["mu"].append(mu)
            taps["scale"].append(scale)
        liks.append(E.gc_likelihood(torch.round(y_slice - mu) + mu, scale, mu))
        y_hat_slices.append(_lrp(p, i, mean_support, torch.round(y_slice - mu) + mu))
    y_hat = torch.cat(y_hat_slices, 1)
    if taps is not None:
        taps["y_hat"] = y_hat
    return {"x_hat": g_s(p, y_hat), "likelihoods": {"y": torch.cat(liks, 1), "z": z_lik}}


def tables(p):
    """update() (image_model.py:601-607)."""
    return {"eb": E.eb_tables(p.sd, p.prefix + "entropy_bottleneck."), "gc": E.gc_tables()}


def compress(p, tabs, x, taps=None):
    """image_model.py:491-552 -> {"strings": [[y_bytes], [z_bytes...]], "shape": z spatial shape}."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    y = g_a(p, x)
    y_shape = y.shape[2:]
    z = h_a(p, y)
    z_strings = E.eb_compress(p.sd, eb_prefix, tabs["eb"], z)
    z_hat = E.eb_decompress(p.sd, eb_prefix, tabs["eb"], z_strings, z.shape[-2:])
    latent_scales = h_s(p, "h_scale_s", z_hat)
    latent_means = h_s(p, "h_mean_s", z_hat)
    gc = tabs["gc"]
    y_hat_slices, symbols, indexes = [], [], []
    for i, y_slice in enumerate(y.chunk(NUM_SLICES, 1)):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, y_hat_slices, y_shape)
        idx = E.gc_build_indexes(scale, gc["scale_table"])
        q = torch.round(y_slice - mu).to(torch.int32)
        symbols.append(q.reshape(-1))
        indexes.append(idx.reshape(-1))
        y_hat_slices.append(_lrp(p, i, mean_support, q + mu))
    symbols, indexes = torch.cat(symbols).numpy(), torch.cat(indexes).numpy()
    if taps is not None:
        taps.update(symbols=symbols, indexes=indexes, y=y, z=z)
    y_string = E.rans_encode(symbols, indexes, gc["cdf"], gc["cdf_length"], gc["offset"])
    return {"strings": [[y_string], z_strings], "shape": tuple(z.shape[-2:])}


def decompress(p, tabs, strings, shape, taps=None):
    """image_model.py:554-595 -> {"x_hat"} clamped to [0, 1]."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    z_hat = E.eb_decompress(p.sd, eb_prefix, tabs["eb"], strings[1], shape)
    latent_scales = h_s(p, "h_scale_s", z_hat)
    latent_means = h_s(p, "h_mean_s", z_hat)
    y_shape = [z_hat.shape[2] * 4, z_hat.shape[3] * 4]
    gc = tabs["gc"]
    dec = E.RansDecoder(strings[0][0])
    y_hat_slices = []
    for i in range(NUM_SLICES):
        mean_support, mu, scale = _slice_params(p, i, latent_means, latent_scales, y_hat_slices, y_shape)
        idx = E.gc_build_indexes(scale, gc["scale_table"])
        rv = dec.decode(idx.reshape(-1).numpy(), gc["cdf"], gc["cdf_length"], gc["offset"])
        rv = torch.from_numpy(rv.astype("float32")).reshape(1, -1, y_shape[0], y_shape[1])
        y_hat_slices.append(_lrp(p, i, mean_support, rv + mu))
    y_hat = torch.cat(y_hat_slices, 1)
    if taps is not None:
        taps["y_hat"] = y_hat
    return {"x_hat": g_s(p, y_hat).clamp_(0, 1)}


class ICIPOracle:
    """Functional ICIP2020ResB over a state_dict (keys = reference attribute paths)."""

    def __init__(self, state_dict):
        self.sd = {k: v.detach().to(torch.float32).cpu() if v.is_floating_point() else v.detach().cpu()
                   for k, v in state_dict.items()}
        self.root = P(self.sd)
        self.tabs = None

    def update(self, force=False):
        if self.tabs is None or force:
            self.tabs = tables(self.root)
            return True
        return False

    def forward(self, x, taps=None):
        return forward(self.root, x, taps)

    def compress(self, x, taps=None):
        self.update()
        return compress(self.root, self.tabs, x, taps)

    def decompress(self, strings, shape, taps=None):
        self.update()
        return decompress(self.root, self.tabs, strings, shape, taps)
