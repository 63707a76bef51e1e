"""Generate tests/golden/icip_ref_{64,128}.npz by running the REFERENCE's `image_model.ICIP2020ResB` (imported from
/root/reference with the stand-ins of make_golden.py) on a synthetic frame with the calibrated W0 weights of
deepsvc_amd.synth.make_icodec.  Build-container only; nothing of the reference is copied -- only inputs/outputs.

    python tests/golden/make_golden_icip.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stand-ins, compact())

from deepsvc_amd import synth  # noqa: E402

TAPPED = ["g_a", "h_a", "h_mean_s", "h_scale_s", "g_s"] + \
         [f"{n}.{i}" for n in ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms") for i in (0, 4, 5, 9)]


def run_case(ref_model, size, out, tag):
    x = synth.make_frames(size, size, 1)[0]
    caps = {}

    def hook(name):
        def fn(_m, _inp, o):
            caps.setdefault(name, []).append(o)
        return fn

    mods = dict(ref_model.named_modules())
    handles = [mods[n].register_forward_hook(hook(n)) for n in TAPPED]
    with torch.no_grad():
        fwd = ref_model(x)
    for h in handles:
        h.remove()
    for k, lst in caps.items():
        for c, t in enumerate(lst):
            MG.compact(f"{tag}/fwd/{k}@{c}", t, out)
    MG.compact(f"{tag}/fwd/x_hat", fwd["x_hat"], out)
    out[f"{tag}/fwd/scalars"] = np.asarray([float(torch.log(fwd["likelihoods"]["y"]).sum()),
                                            float(torch.log(fwd["likelihoods"]["z"]).sum())], dtype=np.float64)
    with torch.no_grad():
        enc = ref_model.compress(x)
        dec = ref_model.decompress(enc["strings"], enc["shape"])
    out[f"{tag}/enc/y"] = np.frombuffer(enc["strings"][0][0], dtype=np.uint8)
    out[f"{tag}/enc/z"] = np.frombuffer(enc["strings"][1][0], dtype=np.uint8)
    out[f"{tag}/enc/shape"] = np.asarray(enc["shape"], dtype=np.int64)
    MG.compact(f"{tag}/dec/x_hat", dec["x_hat"], out)


def main():
    MG.import_reference()
    import image_model as ref_im  # noqa: the reference's module
    torch.manual_seed(0)
    ref_model = ref_im.ICIP2020ResB()
    prod = synth.make_icodec("W0")
    sd, ref_sd = prod.state_dict(), ref_model.state_dict()
    skip = lambda k: ("entropy_bottleneck" in k or "gaussian_conditional" in k)
    ref_keys = {k: tuple(v.shape) for k, v in ref_sd.items() if not skip(k)}
    prod_keys = {k: tuple(v.shape) for k, v in sd.items() if not skip(k)}
    assert ref_keys == prod_keys, "state_dict keys/shapes differ between reference and product"
    load = {k: v for k, v in sd.items() if k in ref_sd}
    missing = [k for k in ref_sd if k not in load]
    assert not missing, missing
    nn.Module.load_state_dict(ref_model, load, strict=True)
    ref_model.eval()
    ref_model.update(force=True)
    for size in (64, 128):
        out = {"meta/n_params": np.asarray([sum(p.numel() for p in ref_model.parameters())], dtype=np.int64),
               "meta/n_keys_non_entropy": np.asarray([len(ref_keys)], dtype=np.int64)}
        run_case(ref_model, size, out, f"s{size}")
        path = os.path.join(HERE, f"icip_ref_{size}.npz")
        np.savez_compressed(path, **out)
        print(os.path.basename(path), os.path.getsize(path) // 1024, "KiB")
    with open(os.path.join(HERE, "icip_state_dict_keys.txt"), "w") as f:
        for k, shp in sorted(ref_keys.items()):
            f.write(f"{k} {list(shp)}\n")


if __name__ == "__main__":
    main()
