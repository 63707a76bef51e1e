"""Generate tests/golden/fcodec_ref_{64,128}.npz by running the REFERENCE's `semantic_layer.cFeatureCompress(N=72)` and
`semantic_layer.RefineNet1` (imported from /root/reference) on synthetic inputs with the calibrated W0 weights of
deepsvc_amd.synth.make_student_layer1.  Build-container only; nothing of the reference is copied -- only inputs/outputs.

The reference module imports mmcv (absent here) for its ResNet stages, which are out of scope: `mmcv.cnn` / `mmcv.runner`
are satisfied by inert stand-ins (nn.Module / nn.Sequential / unused builders) next to the compressai stand-ins of
make_golden.py; only the two pure-torch classes above are instantiated.

    python tests/golden/make_golden_fcodec.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stand-ins, compact())

from deepsvc_amd import synth  # noqa: E402


def import_reference_semantic_layer():
    MG.import_reference()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    unused = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("mmcv stand-in"))
    mm = mod("mmcv")
    mm.cnn = mod("mmcv.cnn", build_conv_layer=unused, build_norm_layer=unused, build_plugin_layer=unused)
    mm.runner = mod("mmcv.runner", BaseModule=nn.Module, Sequential=nn.Sequential)
    import semantic_layer  # noqa: the reference's module
    return semantic_layer


TAPPED = ["g_a", "h_a", "h_mean_s", "h_scale_s", "fea_convert", "g_s", "g_s1"] + \
         [f"{n}.{i}" for n in ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms") for i in (0, 3, 4, 7)]


def run_case(codec, refine, size, out, tag):
    x, supp = synth.make_semantic_input(size, size, 1)
    fea = supp[-1:]
    caps = {}

    def hook(name):
        def fn(_m, _inp, o):
            caps.setdefault(name, []).append(o)
        return fn

    mods = dict(codec.named_modules())
    handles = [mods[n].register_forward_hook(hook(n)) for n in TAPPED]
    with torch.no_grad():
        fwd = codec(x, fea=fea)
    for h in handles:
        h.remove()
    for k, lst in caps.items():
        for c, t in enumerate(lst):
            MG.compact(f"{tag}/fwd/{k}@{c}", t, out)
    MG.compact(f"{tag}/fwd/x_hat", fwd["x_hat"], out)
    MG.compact(f"{tag}/fwd/y", fwd["y"], out)
    out[f"{tag}/fwd/scalars"] = np.asarray([float(torch.log(fwd["likelihoods"]["y"]).sum()),
                                            float(torch.log(fwd["likelihoods"]["z"]).sum())], dtype=np.float64)
    with torch.no_grad():
        enc = codec.compress(x, fea)
        dec = codec.decompress(enc["strings"], enc["shape"], fea)
        agg_mean = refine(dec["x_hat"], supp, mean=True)
        agg_sum = refine(dec["x_hat"], supp, mean=False)
    out[f"{tag}/enc/y"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
    out[f"{tag}/enc/z"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
    out[f"{tag}/enc/shape"] = np.asarray(enc["shape"], dtype=np.int64)
    MG.compact(f"{tag}/enc/latent_y", enc["y"], out)
    MG.compact(f"{tag}/dec/x_hat", dec["x_hat"], out)
    MG.compact(f"{tag}/refine/mean", agg_mean, out)
    MG.compact(f"{tag}/refine/sum", agg_sum, out)


def main():
    ref = import_reference_semantic_layer()
    torch.manual_seed(0)
    codec, refine = ref.cFeatureCompress(N=72), ref.RefineNet1()
    prod = synth.make_student_layer1("W0")
    sd = prod.state_dict()
    skip = lambda k: ("entropy_bottleneck" in k or "gaussian_conditional" in k)
    for name, m in (("layer1", codec), ("refine", refine)):
        ref_sd = m.state_dict()
        sub = {k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")}
        ref_keys = {k: tuple(v.shape) for k, v in ref_sd.items() if not skip(k)}
        prod_keys = {k: tuple(v.shape) for k, v in sub.items() if not skip(k)}
        assert ref_keys == prod_keys, f"{name}: state_dict keys/shapes differ between reference and product"
        load = {k: v for k, v in sub.items() if k in ref_sd}
        missing = [k for k in ref_sd if k not in load]
        assert not missing, missing
        nn.Module.load_state_dict(m, load, strict=True)
        m.eval()
        with open(os.path.join(HERE, f"fcodec_{name}_state_dict_keys.txt"), "w") as f:
            for k, shp in sorted(ref_keys.items()):
                f.write(f"{k} {list(shp)}\n")
    codec.update(force=True)
    for size in (64, 128):
        out = {"meta/n_params": np.asarray([sum(p.numel() for p in codec.parameters()), sum(p.numel() for p in refine.parameters())], dtype=np.int64)}
        run_case(codec, refine, size, out, f"s{size}")
        path = os.path.join(HERE, f"fcodec_ref_{size}.npz")
        np.savez_compressed(path, **out)
        print(os.path.basename(path), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
