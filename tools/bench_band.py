"""Producer->consumer chains of memory-bound convs (x->y, y->x, ...) at several image heights: effective TB/s, with and
without alternating the walk direction between consecutive launches (DSVC_REV_WALK=1; 1x1 kernel only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d
ops.set_precision("f16")
dev = "cuda:0"
for H in (1088, 544):
    m = Conv2d(64, 64, 1, 1, 0).to(dev)
    x = ops.Act(torch.randn((1, H, 1920, 64), device=dev).to(torch.float16))
    y = ops.new_act(1, H, 1920, 64, torch.float16, dev)
    for _ in range(4):
        m.run(x, y); m.run(y, x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            m.run(x, y); m.run(y, x)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    gb = 2 * H * 1920 * 64 * 2 / 1e9
    print(f"1x1 64->64 chain  H={H:5d}  {ms*1e3:8.1f} us/launch  {gb/ms:6.2f} TB/s effective  (REV_WALK={os.environ.get('DSVC_REV_WALK','0')})", flush=True)
