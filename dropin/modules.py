"""Drop-in shim for `modules` (reference test_video.py:10: read_image, cal_psnr, crop, pad)."""
from deepsvc_amd.modules import *  # noqa: F401,F403
from deepsvc_amd.modules import cal_psnr, crop, pad, torch_warp, get_scale_table  # noqa: F401
