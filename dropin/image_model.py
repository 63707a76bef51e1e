"""Drop-in shim for `image_model` (reference test_video.py:9 imports ICIP2020ResB from here)."""
from deepsvc_amd.image_model import ChannelSplitICIP2020ResB, ICIP2020ResB, get_scale_table  # noqa: F401
