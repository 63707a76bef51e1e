"""Launch time of the layout / pointwise kernels at 1080p sizes.   python tools/bench_pointwise.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops

dev = "cuda:0"
def timeit(name, fn, nbytes, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:44s} {us:8.1f} us  {nbytes/us/1e3:7.0f} GB/s", flush=True)

sm = torch.randn(1, 256, 272, 480, device=dev)
out = ops.new_act(1, 1088, 1920, 16, torch.float16, dev)
timeit("pixel_shuffle4 [256,272,480] f32 -> f16 NHWC", lambda: ops.pixel_shuffle_nchw(sm, 4, out=out), sm.numel() * 4 + out.buf.numel() * 2)
x = ops.new_act(1, 1088, 1920, 64, torch.float16, dev); x.buf.normal_()
timeit("pool2x2 avg 64ch f16 @1080p", lambda: ops.pool2x2(x, "avg"), x.buf.numel() * 2 * 1.25)
