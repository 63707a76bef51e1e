"""Same-box A/B of two BUILDS of the library on the two-group 3x3 kernel's variants at 1088x1920 (f16): each build runs in its own child
process (DSVC_LIB=...), the children alternate over the rounds, best-of-rounds per variant is printed side by side.
    python tools/bench_pp_libs.py deepsvc_amd/lib/libdsvc_<variant>.so deepsvc_amd/lib/libdsvc.so [--rounds 3]   (variants: python -m deepsvc_amd.build --variant NAME -DFOO=...)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, sys.argv[1])
from deepsvc_amd import _native as N, ops
from deepsvc_amd.modules import Conv2d
ops.set_precision("f16")
dev, h = "cuda:0", torch.float16
H, W = 1088, 1920
torch.manual_seed(0)
x = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h)); x32 = ops.Act(torch.randn((1, H, W, 32), device=dev).to(h))
r1 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h)); r2 = ops.Act(torch.randn((1, H, W, 64), device=dev).to(h))
conv, c32, c1 = Conv2d(64, 64, 3, 1, 1).to(dev), Conv2d(64, 32, 3, 1, 1).to(dev), Conv2d(32, 64, 3, 1, 1).to(dev)
post = Conv2d(64, 64, 1, 1, 0).to(dev)._packed([64], dev)
CASES = [("plain", conv, x, {}), ("pre relu + act relu", conv, x, dict(pre_act=N.ACT_RELU, act=N.ACT_RELU, slope=0.0)),
         ("pre lrelu + act lrelu", conv, x, dict(pre_act=N.ACT_LRELU, act=N.ACT_LRELU, slope=0.01)), ("res1", conv, x, dict(res1=r1)),
         ("res1 + res2", conv, x, dict(res1=r1, res2=r2)), ("act + res1 + post(1x1)", conv, x, dict(act=N.ACT_LRELU, res1=r1, post=post)),
         ("64->32", c32, x, {}), ("32->64 (1 chunk)", c1, x32, dict(act=N.ACT_LRELU))]
res = {}
for name, m, inp, kw in CASES:
    out = ops.new_act(1, H, W, m.out_channels, h, dev)
    for _ in range(3): m.run(inp, out, out_dtype=h, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): m.run(inp, out, out_dtype=h, **kw)
    e1.record(); torch.cuda.synchronize()
    res[name] = [e0.elapsed_time(e1) / 20 * 1e3, float(out.buf.float().abs().sum())]
print("RESULT " + json.dumps(res))
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
best, sums = {}, {}
for rnd in range(rounds):
    for lib in libs:
        env = dict(os.environ, DSVC_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
        for k, (us, sm) in json.loads(line[0][7:]).items():
            best[(lib, k)] = min(best.get((lib, k), 1e9), us); sums[(lib, k)] = sm
        print(f"round {rnd} {os.path.basename(lib)} done", flush=True)
names = [k for (l, k) in best if l == libs[0]]
print("variant | " + " | ".join(os.path.basename(l) for l in libs) + " | checksums equal")
for k in names:
    print(f"{k:28s} | " + " | ".join(f"{best[(l, k)]:7.1f}" for l in libs) + f" | {len(set(sums[(l, k)] for l in libs)) == 1}")
