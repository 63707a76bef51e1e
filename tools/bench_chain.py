"""Does a producer->consumer chain of the dominant conv run faster when the tensors fit the 256 MB Infinity Cache?  A chain of
3x3 64->64 f16 convs (each reads what the previous one wrote) at full height and on horizontal bands of the frame.
    python tools/bench_chain.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops  # noqa: E402
from deepsvc_amd.modules import Conv2d  # noqa: E402

ops.set_precision("f16")
dev, h = "cuda:0", torch.float16
W = 1920
convs = [Conv2d(64, 64, 3, 1, 1).to(dev) for _ in range(4)]
for H in (1088, 544, 272, 136):
    for kw_name, kw in (("plain", {}), ("res1", "res")):
        bufs = [ops.Act(torch.randn((1, H, W, 64), device=dev).to(h)) for _ in range(3)]
        def chain(n):
            for i in range(n):
                a, b, c = bufs[i % 3], bufs[(i + 1) % 3], bufs[(i + 2) % 3]
                k = {} if kw == {} else {"res1": c}          # residual = the tensor written two launches ago (ResBlock pattern)
                convs[i % 4].run(a, b, out_dtype=h, **k)
        chain(6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); chain(24); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 24 * 1e3
        mb = H * W * 64 * 2 / 1e6
        nb = (2 + (kw != {})) * mb * 1e6
        print(f"H={H:5d} ({mb:5.0f} MB/tensor) {kw_name:6s}: {us:7.1f} us per conv = {us * 1088 / H:7.1f} us per full-frame equivalent, {nb / us / 1e6:5.2f} TB/s, "
              f"{2 * H * W * 64 * 64 * 9 / us / 1e9:6.3f} PFLOP/s", flush=True)
