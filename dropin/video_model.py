"""Drop-in shim: put this directory ahead of the reference checkout on sys.path and `test_video.py:8`
(`from video_model import DeepSVC`) resolves to the MI355X-native implementation."""
from deepsvc_amd.video_model import DeepSVC  # noqa: F401
