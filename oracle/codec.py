"""Oracle restatement of image_model.py:ChannelSplitICIP2020ResB and video_model.py:DeepSVC.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PyTorch-CPU fp32, functional over a state_dict.
Tensor arithmetic pinned against the imported reference (tests/golden); the entropy arithmetic it
calls (oracle.entropy) is "parity unpinned".
"""
import math

import torch
import torch.nn.functional as F

from . import entropy as E
from . import nets as N
from .nets import P

NUM_SLICES = 8
MAX_SUPPORT = 4


# ----------------------------------------------------------------------------------------------
# ChannelSplitICIP2020ResB (image_model.py:41-328)
# ----------------------------------------------------------------------------------------------
def g_a(p, x):
    """image_model.py:52-66: 4 x conv5x5 s2 interleaved with 3 ResBottleneck each."""
    for i in range(13):
        q = p.sub(f"g_a.{i}")
        x = N.conv(q, x, stride=2) if i % 4 == 0 else N.res_bottleneck(q, x)
    return x


def g_s(p, x):
    """image_model.py:68-82."""
    for i in range(13):
        q = p.sub(f"g_s.{i}")
        x = N.deconv(q, x) if i % 4 == 0 else N.res_bottleneck(q, x)
    return x


def h_a(p, x):
    """image_model.py:84-92: conv3x3, GELU, conv3x3 s2, GELU, conv3x3, GELU, conv3x3 s2."""
    x = F.gelu(N.conv(p.sub("h_a.0"), x))
    x = F.gelu(N.conv(p.sub("h_a.2"), x, stride=2))
    x = F.gelu(N.conv(p.sub("h_a.4"), x))
    return N.conv(p.sub("h_a.6"), x, stride=2)


def h_s(p, name, z_hat):
    """h_mean_s / h_scale_s (image_model.py:94-112): subpel3x3(2), GELU, conv, GELU, subpel3x3(2), GELU, conv."""
    q = p.sub(name)
    x = F.gelu(F.pixel_shuffle(N.conv(q.sub("0.0"), z_hat), 2))
    x = F.gelu(N.conv(q.sub("2"), x))
    x = F.gelu(F.pixel_shuffle(N.conv(q.sub("4.0"), x), 2))
    return N.conv(q.sub("6"), x)


def cc_transform(p, x):
    """cc_mean / cc_scale stack (image_model.py:114-136): 4 convs 3x3 with GELU between."""
    for i in (0, 2, 4):
        x = F.gelu(N.conv(p.sub(str(i)), x))
    return N.conv(p.sub("6"), x)


def lrp_transform(p, x):
    """image_model.py:138-146."""
    for i in (0, 2):
        x = F.gelu(N.conv(p.sub(str(i)), x))
    return N.conv(p.sub("4"), x)


def _slice_params(p, i, latent_means, latent_scales, y_hat_slices, y_shape):
    support = y_hat_slices[:MAX_SUPPORT]
    mean_support = torch.cat([latent_means] + support, 1)
    mu = cc_transform(p.sub(f"cc_mean_transforms.{i}"), mean_support)[:, :, : y_shape[0], : y_shape[1]]
    scale_support = torch.cat([latent_scales] + support, 1)
    scale = cc_transform(p.sub(f"cc_scale_transforms.{i}"), scale_support)[:, :, : y_shape[0], : y_shape[1]]
    return mean_support, mu, scale


def _lrp(p, i, mean_support, y_hat_slice):
    lrp = lrp_transform(p.sub(f"lrp_transforms.{i}"), torch.cat([mean_support, y_hat_slice], 1))
    return y_hat_slice + 0.5 * torch.tanh(lrp)


def codec_forward(p, x, taps=None):
    """forward (image_model.py:151-199) -> {"x_hat", "likelihoods": {"y","z"}}."""
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
        y_hat_slice = torch.round(y_slice - mu) + mu
        y_hat_slices.append(_lrp(p, i, mean_support, y_hat_slice))
    y_hat = torch.cat(y_hat_slices, 1)
    if taps is not None:
        taps["y_hat"] = y_hat
    return {"x_hat": g_s(p, y_hat), "likelihoods": {"y": torch.cat(liks, 1), "z": z_lik}}


def codec_tables(p):
    """update() (image_model.py:319-324)."""
    return {"eb": E.eb_tables(p.sd, p.prefix + "entropy_bottleneck."), "gc": E.gc_tables()}


def codec_compress(p, tables, x, taps=None):
    """compress (image_model.py:201-257) -> {"strings": [[y_bytes],[z_bytes...]], "shape": (h/4, w/4)}."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    y = g_a(p, x)
    y_shape = y.shape[2:]
    z = h_a(p, y)
    z_strings = E.eb_compress(p.sd, eb_prefix, tables["eb"], z)
    z_hat = E.eb_decompress(p.sd, eb_prefix, tables["eb"], z_strings, z.shape[-2:])
    latent_scales = h_s(p, "h_scale_s", z_hat)
    latent_means = h_s(p, "h_mean_s", z_hat)
    gc = tables["gc"]
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


def codec_decompress(p, tables, strings, shape, taps=None):
    """decompress (image_model.py:259-302) -> {"x_hat"}."""
    eb_prefix = p.prefix + "entropy_bottleneck."
    z_hat = E.eb_decompress(p.sd, eb_prefix, tables["eb"], strings[1], shape)
    latent_scales = h_s(p, "h_scale_s", z_hat)
    latent_means = h_s(p, "h_mean_s", z_hat)
    y_shape = [z_hat.shape[2] * 4, z_hat.shape[3] * 4]
    gc = tables["gc"]
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
    return {"x_hat": g_s(p, y_hat)}


# ----------------------------------------------------------------------------------------------
# DeepSVC (video_model.py:13-191)
# ----------------------------------------------------------------------------------------------
class DeepSVCOracle:
    """Functional DeepSVC over a state_dict (keys = reference attribute paths)."""

    def __init__(self, state_dict):
        self.sd = {k: v.detach().to(torch.float32).cpu() if v.is_floating_point() else v.detach().cpu()
                   for k, v in state_dict.items()}
        self.root = P(self.sd)
        self.tables = None

    def update(self, force=False):
        """video_model.py:179-182."""
        if self.tables is None or force:
            self.tables = {"mv": codec_tables(self.root.sub("mv_codec")),
                           "res": codec_tables(self.root.sub("res_codec"))}
            return True
        return False

    # -- shared stages -------------------------------------------------------------------------
    def _predict(self, ref, recon_mv1, sm_fea, feature, taps):
        r = self.root
        recon_mv = N.refine_net(r.sub("RefineMvNet"), recon_mv1, ref)
        warped = N.warp(ref, recon_mv)
        warp_fea, predict = N.inter_layer_prediction(r.sub("MC"), ref, warped, recon_mv, sm_fea, feature)
        if taps is not None:
            taps.update(recon_mv1=recon_mv1, recon_mv=recon_mv, warped=warped, warp_fea=warp_fea, predict=predict)
        return warped, warp_fea, predict

    def _reconstruct(self, ref, predict_fea, recon_res1, warp_fea, taps):
        r = self.root
        recon_res = N.refine_net(r.sub("RefineResiNet"), recon_res1, ref)
        feature, recon = N.reconstruction(r.sub("enhance"), torch.cat([predict_fea + recon_res, warp_fea], 1))
        if taps is not None:
            taps.update(recon_res1=recon_res1, recon_res=recon_res, feature=feature, recon=recon)
        return feature, recon

    # -- video_model.py:27-71 ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, ref, cur, sm_fea, feature=None, taps=None):
        r = self.root
        pixels = cur.shape[0] * cur.shape[2] * cur.shape[3]
        mv_est = N.spynet(r.sub("opticFlow"), cur, ref)
        mv_taps = {} if taps is not None else None
        mv_out = codec_forward(r.sub("mv_codec"), torch.cat([cur, mv_est, ref], 1), mv_taps)
        warped, warp_fea, predict = self._predict(ref, mv_out["x_hat"], sm_fea, feature, taps)
        bpp_of = lambda out: sum(torch.log(l).sum() / (-math.log(2) * pixels) for l in out["likelihoods"].values())
        bpp_mv = bpp_of(mv_out)
        predict_fea = N.feature_extraction(r.sub("FeatureExtractor"), predict)
        cur_fea = N.feature_extraction(r.sub("FeatureExtractor"), cur)
        res_taps = {} if taps is not None else None
        res_out = codec_forward(r.sub("res_codec"), torch.cat([ref, cur_fea - predict_fea, predict], 1), res_taps)
        bpp_res = bpp_of(res_out)
        feat, recon = self._reconstruct(ref, predict_fea, res_out["x_hat"], warp_fea, taps)
        if taps is not None:
            taps.update(mv_est=mv_est, mv=mv_taps, res=res_taps, predict_fea=predict_fea, cur_fea=cur_fea)
        mse = torch.mean((recon - cur) ** 2)
        return (recon, feat, mse, torch.mean((warped - cur) ** 2), torch.mean((predict - cur) ** 2),
                bpp_res, bpp_mv, bpp_mv + bpp_res)

    # -- video_model.py:137-151 ----------------------------------------------------------------
    @torch.no_grad()
    def compress(self, ref, cur, sm_fea, feature=None, taps=None):
        r = self.root
        self.update()
        mv_est = N.spynet(r.sub("opticFlow"), cur, ref)
        mv_taps = {} if taps is not None else None
        mv_enc = codec_compress(r.sub("mv_codec"), self.tables["mv"], torch.cat([cur, mv_est, ref], 1), mv_taps)
        mv1 = codec_decompress(r.sub("mv_codec"), self.tables["mv"], mv_enc["strings"], mv_enc["shape"])["x_hat"]
        _, _, predict = self._predict(ref, mv1, sm_fea, feature, None)
        predict_fea = N.feature_extraction(r.sub("FeatureExtractor"), predict)
        cur_fea = N.feature_extraction(r.sub("FeatureExtractor"), cur)
        res_taps = {} if taps is not None else None
        res_enc = codec_compress(r.sub("res_codec"), self.tables["res"],
                                 torch.cat([ref, cur_fea - predict_fea, predict], 1), res_taps)
        if taps is not None:
            taps.update(mv_est=mv_est, mv=mv_taps, res=res_taps)
        return mv_enc, res_enc

    # -- video_model.py:153-167 ----------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, ref, mv_enc, res_enc, sm_fea, feature, taps=None):
        r = self.root
        self.update()
        mv_t, res_t = ({}, {}) if taps is not None else (None, None)
        mv1 = codec_decompress(r.sub("mv_codec"), self.tables["mv"], mv_enc["strings"], mv_enc["shape"], mv_t)["x_hat"]
        warped, warp_fea, predict = self._predict(ref, mv1, sm_fea, feature, taps)
        predict_fea = N.feature_extraction(r.sub("FeatureExtractor"), predict)
        res1 = codec_decompress(r.sub("res_codec"), self.tables["res"], res_enc["strings"], res_enc["shape"], res_t)["x_hat"]
        feat, recon = self._reconstruct(ref, predict_fea, res1, warp_fea, taps)
        if taps is not None:      # un-clamped tensors + decoded latents for the parity tests
            taps.update(mv_y_hat=mv_t["y_hat"], res_y_hat=res_t["y_hat"], predict_fea=predict_fea)
        return feat, recon.clamp(0.0, 1.0), warped.clamp(0.0, 1.0), predict.clamp(0.0, 1.0)


def strings_nbytes(enc):
    """test_video.py:364-365: sum(len(s[0]) for s in strings)."""
    return sum(len(s[0]) for s in enc["strings"])
