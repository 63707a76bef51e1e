"""deepsvc_amd -- MI355X-native (gfx950) implementation of DeepSVC's structure/texture-layer P-frame
encode/decode hot path, behind the reference's own Python module API.

    from deepsvc_amd.video_model import DeepSVC          # == reference video_model.DeepSVC
    from deepsvc_amd.modules import pad, crop, cal_psnr  # == reference modules.*

Compute = hand-written HIP kernels in deepsvc_amd/csrc (C ABI: include/dsvc.h, built into
deepsvc_amd/lib/libdsvc.so by `python -m deepsvc_amd.build`).  There is no CPU execution path.
"""
__version__ = "0.1.0"
