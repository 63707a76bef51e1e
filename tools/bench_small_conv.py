"""Per-launch time of the small-image (entropy-scan) convs without Python overhead: N launches captured in a HIP graph,
replayed and timed with events, for each kernel choice (dsvc_set_option).   python tools/bench_small_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d

SHAPES = [  # (name, segs, seg dtypes, cout, H, W, act, out dtype)
    ("64->32 gelu", (64,), ("f16",), 32, 68, 120, N.ACT_GELU, "f16"),
    ("64->64 gelu", (64,), ("f16",), 64, 68, 120, N.ACT_GELU, "f16"),
    ("96->64 gelu", (96,), ("f16",), 64, 68, 120, N.ACT_GELU, "f16"),
    ("96+48(f32)->96 gelu", (96, 48), ("f32", "f32"), 96, 68, 120, N.ACT_GELU, "f16"),
    ("64+32(f32)->64 gelu", (64, 32), ("f32", "f32"), 64, 68, 120, N.ACT_GELU, "f16"),
    ("96+48+16(f32,f32,f16)->96 gelu", (96, 48, 16), ("f32", "f32", "f16"), 96, 68, 120, N.ACT_GELU, "f16"),
    ("96+36(f32)->96 gelu", (96, 36), ("f32", "f32"), 96, 68, 120, N.ACT_GELU, "f16"),
    ("32->12 f32out", (32,), ("f16",), 12, 68, 120, N.ACT_NONE, "f32"),
    ("64->32 none", (64,), ("f16",), 32, 68, 120, N.ACT_NONE, "f16"),
    ("64->64 @136x240", (64,), ("f16",), 64, 136, 240, N.ACT_LRELU, "f16"),
]
dev = "cuda:0"
ops.set_precision("f16")
dts = {"f32": torch.float32, "f16": torch.float16}
NL = 40
for name, segs, sdt, cout, H, W, act, odt in SHAPES:
    torch.manual_seed(0)
    m = Conv2d(sum(segs), cout, 3, 1, 1).to(dev)
    ins = [ops.Act(torch.randn((1, H, W, c), device=dev).to(dts[t])) for c, t in zip(segs, sdt)]
    out = ops.new_act(1, H, W, cout, dts[odt], dev)
    res = []
    for _once in (0,):
        for _ in range(3):
            m.run(ins, out, act=act)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(NL):
                m.run(ins, out, act=act)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / (5 * NL) * 1e3)
    print(f"{name:26s} {res[0]:6.1f} us per launch in graph replay", flush=True)
