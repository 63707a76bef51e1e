"""Probe: which multi-stream capture patterns does this ROCm accept?  (run on the GPU box; prints one line per pattern)"""
import faulthandler, sys, torch
faulthandler.enable()
dev = torch.device("cuda:0")
a = torch.ones(1 << 16, device=dev)
outs = [torch.zeros_like(a) for _ in range(8)]
pool = {}


def par(fns):
    cur = torch.cuda.current_stream()
    ss = pool.setdefault(cur.cuda_stream, [])
    while len(ss) < len(fns) - 1:
        ss.append(torch.cuda.Stream(dev))
    for s, fn in zip(ss, fns[1:]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            fn()
    fns[0]()
    for s in ss[:len(fns) - 1]:
        cur.wait_stream(s)


def k(i, n=1):
    def f():
        for _ in range(n):
            torch.add(a, float(i), out=outs[i])
    return f


def nested(i):
    return lambda: par([k(i), k(i + 4)])


patterns = {
    "flat2": lambda: par([k(0), k(1)]),
    "flat4": lambda: par([k(0, 3), k(1, 3), k(2, 3), k(3, 3)]),
    "seq_then_flat": lambda: (k(0)(), par([k(1), k(2)]), k(3)()),
    "two_forks": lambda: (par([k(0), k(1)]), par([k(2), k(3)])),
    "nested": lambda: par([nested(0), nested(1)]),
    "nested4": lambda: par([nested(0), nested(1), nested(2), nested(3)]),
}
which = sys.argv[1:] or list(patterns)
for name in which:
    fn = patterns[name]
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print(name, "capturing", flush=True)
    with torch.cuda.graph(g):
        fn()
    print(name, "captured", flush=True)
    g.replay(); torch.cuda.synchronize()
    print(name, "ok", [float(o[0]) for o in outs], flush=True)
