"""Sub-pixel convs (PixelShuffle(2) folded into the store) of a 1080p P-frame: vector store path on / off, and the same conv without the
shuffle as the yardstick.   python tools/bench_subpel.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import ops  # noqa: E402
from deepsvc_amd.modules import Conv2d, _SubpelConv  # noqa: E402

ops.set_precision("f16")
dev = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for cin, cq, k, H, W in [(96, 64, 3, 544, 960), (64, 32, 1, 544, 960), (64, 64, 3, 272, 480), (96, 96, 3, 272, 480), (128, 64, 1, 272, 480),
                         (96, 96, 3, 136, 240), (64, 64, 3, 136, 240), (96, 96, 3, 68, 120)]:
    torch.manual_seed(0)
    sp = _SubpelConv(cin, cq, 2, k).to(dev)
    plain = Conv2d(cin, cq * 4, k, 1, padding=k // 2).to(dev)
    x = ops.to_act(torch.randn(1, cin, H, W, device=dev), torch.float16)
    o1 = ops.new_act(1, 2 * H, 2 * W, cq, torch.float16, dev)
    o2 = ops.new_act(1, H, W, 4 * cq, torch.float16, dev)
    t_vec = timeit(lambda: sp.run(x, o1))
    ops.set_option("shuffle_vec", 0)
    t_el = timeit(lambda: sp.run(x, o1))
    ops.set_option("shuffle_vec", 1)
    t_plain = timeit(lambda: plain.run(x, o2))
    fl = 2.0 * H * W * cin * cq * 4 * k * k
    print(f"{k}x{k} {cin:>3}->{cq * 4:>3} @{H}x{W}: subpel vector {t_vec:7.1f} us ({fl / t_vec / 1e6:6.1f} TFLOP/s)   element-wise {t_el:7.1f} us   no shuffle {t_plain:7.1f} us", flush=True)
