"""Drop-in shim for `image_model` (reference test_video.py:9 imports ICIP2020ResB from here; the I-frame codec is
SURVEY section 8f-1 'next' and is not provided yet)."""
from deepsvc_amd.image_model import ChannelSplitICIP2020ResB, get_scale_table  # noqa: F401
