"""Probe for DESIGN 9.0a's untried idea: does a chain of 64-channel ResBlocks run faster per pixel when the picture is executed in horizontal bands whose
tensors fit the 256 MB Infinity Cache?  Three ResBlock1 (conv + conv + residual: csrc/conv3x3_pp.hip) back to back on a 1088x1920 tensor, against the same
chain on 2 / 4 bands of 544 / 272 rows (no halo handling: a timing probe, not an implementation).   python tools/probe_bands.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import ResBlock1, ResBottleneckBlock  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, mk in (("3 x ResBlock1", lambda: [ResBlock1(64).to(dev) for _ in range(3)]), ("3 x ResBottleneck", lambda: [ResBottleneckBlock(64).to(dev) for _ in range(3)])):
    blocks = mk()
    for nb in (1, 2, 4, 8):
        H = 1088 // nb
        xs = [ops.to_act(torch.randn(1, 64, H, 1920, device=dev), torch.float16) for _ in range(nb)]
        bufs = [[ops.new_act(1, H, 1920, 64, torch.float16, dev) for _ in range(2)] for _ in range(nb)]

        def run():
            for x, (b0, b1) in zip(xs, bufs):
                t = blocks[0].run(x, b0)
                t = blocks[1].run(t, b1)
                blocks[2].run(t, b0)
        t = timeit(run)
        print(f"{name} on 64 ch f16 1088x1920 in {nb} band(s) of {H} rows: {t:8.1f} us", flush=True)
