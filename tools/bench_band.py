"""Does a producer->consumer chain run faster when the activation fits the 256 MB Infinity Cache?  Two chained 1x1
convs (x->y, y->x) on H x 1920 x 64 bf16 images of different heights; reports effective TB/s (read+write once each)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d
ops.set_precision("bf16")
dev = "cuda:0"
for k in (1, 3):
    for H in (1088, 544, 272, 136):
        m = Conv2d(64, 64, k, 1, k // 2).to(dev)
        x = ops.Act(torch.randn((1, H, 1920, 64), device=dev).to(torch.bfloat16))
        y = ops.new_act(1, H, 1920, 64, torch.bfloat16, dev)
        for _ in range(3):
            m.run(x, y); m.run(y, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            m.run(x, y); m.run(y, x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (2 * n)
        gb = 2 * H * 1920 * 64 * 2 / 1e9
        print(f"{k}x{k} 64->64  H={H:5d}  {ms*1e3:8.1f} us/launch  {gb/ms:6.2f} TB/s effective   ({gb*500:.0f} MB per tensor)", flush=True)
