"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference) on the
synthetic inputs / W0 weights of deepsvc_amd.synth (64x64 first + steady-state P-frame with every stage tapped, 128x128
steady-state, and a 256x256 septuplet = BASELINE.json config 1: six chained P-frames through compress()/decompress()).  Build-container only: /root/reference does not exist on
the GPU box, where only the committed .npz files are used.

    python tests/golden/make_golden.py

How the reference is imported: its third-party imports that are not
installed here (torchvision, pytorch_msssim, compressai) are satisfied by stand-in modules registered in
sys.modules *before* the import.  Layer constructors are the 1-line compressai definitions
(Appendix A.1); the entropy-model / range-coder stand-ins delegate to the oracle's restatement
(oracle/entropy.py), so the reference's own forward()/compress()/decompress() code runs end to end.
Consequently the fixtures pin the oracle's *network arithmetic, tensor plumbing and slice loop* against the
reference (rows a1-a15 of SURVEY.md section 8), while the entropy arithmetic itself stays "parity unpinned".

Nothing of the reference is copied: only inputs/outputs (numpy arrays, byte strings) are stored.
Large tensors are stored sub-sampled (every 4th pixel) together with fp64 sum / sum-of-squares of the
full tensor.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from deepsvc_amd import synth  # noqa: E402
from oracle import entropy as E  # noqa: E402


# ------------------------------------------------------------------------------------------------
# stand-ins
# ------------------------------------------------------------------------------------------------
class _EB(nn.Module):
    """compressai.entropy_models.EntropyBottleneck stand-in backed by oracle.entropy."""

    def __init__(self, channels):
        super().__init__()
        f = E.EB_FILTERS
        for i in range(5):
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.zeros(channels, f[i + 1], f[i])))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
            if i < 4:
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.zeros(channels, 1, 3))
        self.register_buffer("target", torch.zeros(3))
        self.tables = None

    def _sd(self):
        return {k: v.detach() for k, v in self.state_dict().items()}

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def forward(self, x):
        return E.eb_forward(self._sd(), "", x)

    def update(self, force=False):
        if self.tables is not None and not force:
            return False
        self.tables = E.eb_tables(self._sd(), "")
        return True

    def compress(self, x):
        return E.eb_compress(self._sd(), "", self.tables, x)

    def decompress(self, strings, size):
        return E.eb_decompress(self._sd(), "", self.tables, strings, size)

    def loss(self):
        return torch.zeros(())


class _GC(nn.Module):
    """compressai.entropy_models.GaussianConditional stand-in backed by oracle.entropy."""

    def __init__(self, scale_table):
        super().__init__()
        self.t = None

    def update_scale_table(self, table, force=False):
        if self.t is not None and not force:
            return False
        self.t = E.gc_tables(table)
        return True

    quantized_cdf = property(lambda s: torch.from_numpy(s.t["cdf"]))
    cdf_length = property(lambda s: torch.from_numpy(s.t["cdf_length"]))
    offset = property(lambda s: torch.from_numpy(s.t["offset"]))

    def forward(self, y, scales, means=None):
        return E.gc_forward(y, scales, means)

    def build_indexes(self, scales):
        return E.gc_build_indexes(scales, self.t["scale_table"])

    def quantize(self, x, mode, means=None):
        assert mode == "symbols"
        return torch.round(x - means).int()

    def dequantize(self, x, means=None):
        return x.type_as(means) + means


class _Enc:
    def __init__(self):
        self.stream = []

    def encode_with_indexes(self, symbols, indexes, cdf, cdf_lengths, offsets):
        self.stream += E.rans_symbol_stream(symbols, indexes, np.asarray(cdf, dtype=np.int32), cdf_lengths, offsets)

    def flush(self):
        out = E.rans_flush(self.stream)
        self.stream = []
        return out


class _Dec:
    def set_stream(self, data):
        self.d = E.RansDecoder(data, force_python=True)

    def decode_stream(self, indexes, cdf, cdf_lengths, offsets):
        return [int(v) for v in self.d.decode(indexes, np.asarray(cdf, dtype=np.int32), cdf_lengths, offsets)]


def _install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", ToTensor=lambda: (lambda img: img))
    mod("pytorch_msssim", ms_ssim=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("ms_ssim stand-in")))
    conv = lambda i, o, kernel_size=5, stride=2: nn.Conv2d(i, o, kernel_size, stride, padding=kernel_size // 2)
    deconv = lambda i, o, kernel_size=5, stride=2: nn.ConvTranspose2d(i, o, kernel_size, stride, output_padding=stride - 1,
                                                                      padding=kernel_size // 2)
    conv3x3 = lambda i, o, stride=1: nn.Conv2d(i, o, 3, stride, padding=1)
    subpel = lambda i, o, r=1: nn.Sequential(nn.Conv2d(i, o * r ** 2, 3, padding=1), nn.PixelShuffle(r))
    ca = mod("compressai")
    ca.entropy_models = mod("compressai.entropy_models", EntropyBottleneck=_EB, GaussianConditional=_GC)
    ca.layers = mod("compressai.layers", conv3x3=conv3x3, subpel_conv3x3=subpel)
    ca.models = mod("compressai.models")
    ca.models.utils = mod("compressai.models.utils", conv=conv, deconv=deconv,
                          update_registered_buffers=lambda *a, **k: None)
    ca.ops = mod("compressai.ops", ste_round=lambda x: torch.round(x) - x.detach() + x)
    ca.ans = mod("compressai.ans", BufferedRansEncoder=_Enc, RansDecoder=_Dec)


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("/root/reference is absent (GPU box?): golden vectors can only be regenerated in the build container")
    _install_standins()
    sys.path.insert(0, REF)
    import video_model  # noqa: the reference's module
    return video_model


# ------------------------------------------------------------------------------------------------
# capture
# ------------------------------------------------------------------------------------------------
def compact(name, t, out, stride=4):
    a = t.detach().to(torch.float32).cpu().numpy()
    out[name + "/shape"] = np.asarray(a.shape, dtype=np.int64)
    out[name + "/stats"] = np.asarray([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum()])
    out[name + "/data"] = a if a.size <= 16384 else a[..., ::stride, ::stride].copy()


TAPPED = ["opticFlow", "mv_codec.g_a", "mv_codec.h_a", "mv_codec.h_mean_s", "mv_codec.h_scale_s", "mv_codec.g_s",
          "RefineMvNet", "MC", "FeatureExtractor", "res_codec.g_a", "res_codec.g_s", "RefineResiNet", "enhance"] + \
         [f"{c}.{n}.{i}" for c in ("mv_codec", "res_codec") for n in ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms")
          for i in (0, 3, 4, 7)]


def run_case(ref_model, size, with_feature, out, tag):
    H = W = size
    frames = synth.make_frames(H, W, 3)
    caps = {}

    def hook(name):
        def fn(_m, _inp, o):
            outs = o if isinstance(o, (tuple, list)) else (o,)
            for j, t in enumerate(outs):
                caps.setdefault(f"{name}#{j}", []).append(t)
        return fn

    mods = dict(ref_model.named_modules())
    handles = [mods[n].register_forward_hook(hook(n)) for n in TAPPED]
    with torch.no_grad():
        feature = None
        t = 1
        if with_feature:  # produce a decoded feature/ref from frame 0->1 first (chained state, test_video.py:368-369)
            for h in handles:
                h.remove()
            mv, res = ref_model.compress(frames[0], frames[1], synth.make_sm_fea(H, W, 1), None)
            feature, recon, _, _ = ref_model.decompress(frames[0], mv, res, synth.make_sm_fea(H, W, 1), None)
            ref_frame, t = recon, 2
            handles = [mods[n].register_forward_hook(hook(n)) for n in TAPPED]
        else:
            ref_frame = frames[0]
        cur, sm = frames[t], synth.make_sm_fea(H, W, t)
        fwd = ref_model(ref_frame, cur, sm, feature)
    for h in handles:
        h.remove()
    for k, lst in caps.items():
        for c, tns in enumerate(lst):
            compact(f"{tag}/fwd/{k}@{c}", tns, out)
    for j, name in enumerate(["recon", "feature"]):
        compact(f"{tag}/fwd/out_{name}", fwd[j], out)
    out[f"{tag}/fwd/scalars"] = np.asarray([float(v) for v in fwd[2:]], dtype=np.float64)
    with torch.no_grad():
        mv, res = ref_model.compress(ref_frame, cur, sm, feature)
        dec = ref_model.decompress(ref_frame, mv, res, sm, feature)
    for nm, enc in (("mv", mv), ("res", res)):
        out[f"{tag}/enc/{nm}_y"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
        out[f"{tag}/enc/{nm}_z"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
        out[f"{tag}/enc/{nm}_shape"] = np.asarray(enc["shape"], dtype=np.int64)
    for j, name in enumerate(["feature", "recon", "warped", "predict"]):
        compact(f"{tag}/dec/{name}", dec[j], out)
    if with_feature:
        compact(f"{tag}/in/ref_frame", ref_frame, out)
        compact(f"{tag}/in/feature", feature, out)


def run_septuplet(ref_model, size, out, tag, n_frames=7):
    """BASELINE.json config 1: one size x size septuplet, frame 0 given as the decoded key frame, 6 chained P-frames through
    the reference's compress()/decompress() (state chaining of test_video.py:342-369).  Per frame: the four byte strings,
    the decoded pictures (sub-sampled + full-tensor sums), PSNR; decoded feature of the first and last frame."""
    H = W = size
    frames = synth.make_frames(H, W, n_frames)
    ref_frame, feature = frames[0], None
    psnrs = []
    with torch.no_grad():
        for t in range(1, n_frames):
            sm = synth.make_sm_fea(H, W, t)
            mv, res = ref_model.compress(ref_frame, frames[t], sm, feature)
            feature, recon, warped, predict = ref_model.decompress(ref_frame, mv, res, sm, feature)
            for nm, enc in (("mv", mv), ("res", res)):
                out[f"{tag}/f{t}/enc/{nm}_y"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
                out[f"{tag}/f{t}/enc/{nm}_z"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
                out[f"{tag}/f{t}/enc/{nm}_shape"] = np.asarray(enc["shape"], dtype=np.int64)
            for name, tns in (("recon", recon), ("warped", warped), ("predict", predict)):
                compact(f"{tag}/f{t}/dec/{name}", tns, out)
            if t in (1, n_frames - 1):
                compact(f"{tag}/f{t}/dec/feature", feature, out, stride=8)
            psnrs.append(-10.0 * float(torch.log10(((recon - frames[t]) ** 2).mean())))
            ref_frame = recon
    out[f"{tag}/psnr"] = np.asarray(psnrs, dtype=np.float64)


def main():
    vm = import_reference()
    torch.manual_seed(0)
    ref_model = vm.DeepSVC()
    prod = synth.make_model("W0")
    sd = prod.state_dict()
    ref_sd = ref_model.state_dict()
    skip = lambda k: ("entropy_bottleneck" in k or "gaussian_conditional" in k)
    # state-dict contract: every non-entropy key of the reference exists with the same shape in the product
    ref_keys = {k: tuple(v.shape) for k, v in ref_sd.items() if not skip(k)}
    prod_keys = {k: tuple(v.shape) for k, v in sd.items() if not skip(k)}
    assert ref_keys == prod_keys, "state_dict keys/shapes differ between reference and product"
    load = {k: v for k, v in sd.items() if k in ref_sd}
    missing = [k for k in ref_sd if k not in load]
    assert not missing, missing
    nn.Module.load_state_dict(ref_model, load, strict=True)
    ref_model.eval()
    ref_model.update(force=True)
    out = {}
    out["meta/n_params"] = np.asarray([sum(p.numel() for p in ref_model.parameters())], dtype=np.int64)
    out["meta/n_keys_non_entropy"] = np.asarray([len(ref_keys)], dtype=np.int64)
    run_case(ref_model, 64, False, out, "s64_first")
    run_case(ref_model, 64, True, out, "s64_steady")
    np.savez_compressed(os.path.join(HERE, "deepsvc_ref_64.npz"), **out)
    out2 = {}
    run_case(ref_model, 128, True, out2, "s128_steady")
    keep = {k: v for k, v in out2.items() if "/enc/" in k or "/dec/" in k or k.endswith("scalars") or "out_" in k or "/in/" in k
            or "opticFlow" in k or "cc_" in k}
    np.savez_compressed(os.path.join(HERE, "deepsvc_ref_128.npz"), **keep)
    out3 = {}
    run_septuplet(ref_model, 256, out3, "s256_gop")
    np.savez_compressed(os.path.join(HERE, "deepsvc_ref_256.npz"), **out3)
    for f in ("deepsvc_ref_64.npz", "deepsvc_ref_128.npz", "deepsvc_ref_256.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
    # key/shape table of the whole state dict (checkpoint-compat contract for the product)
    with open(os.path.join(HERE, "state_dict_keys.txt"), "w") as f:
        for k, shp in sorted(ref_keys.items()):
            f.write(f"{k} {list(shp)}\n")


if __name__ == "__main__":
    main()
