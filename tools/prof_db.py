"""Summarise a rocprofv3 rocpd database: per-kernel totals over the last `--frames` frames' worth of time.
usage: python tools/prof_db.py gpurun_out/prof_a/a_results.db [--tail-frac 0.6]"""
import re, sqlite3, sys, collections
path = sys.argv[1]
frac = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
db = sqlite3.connect(path)
rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
cut = t1 - (t1 - t0) * frac
sel = [r for r in rows if r[1] >= cut]
span = (sel[-1][2] - sel[0][1]) / 1e6
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gx, wx, lds in sel:
    k = short(n)
    if "conv_mfma" in k or "conv3x3_ws" in k or "conv1x1" in k:
        k += f" g{gx // max(wx,1)}"
    a = agg[k]; a[0] += 1; a[1] += (e - s) / 1e6
# busy time (union of intervals)
iv = sorted((r[1], r[2]) for r in sel)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = sum(a[1] for a in agg.values())
print(f"window {span:.1f} ms, {len(sel)} launches, sum kernel {tot:.1f} ms, busy(union) {busy/1e6:.1f} ms ({100*busy/1e6/span:.1f}%)")
fam = collections.defaultdict(lambda: [0, 0.0])
for k, (c, t) in agg.items():
    f = k.split("<")[0].split(" g")[0]
    fam[f][0] += c; fam[f][1] += t
print("--- families")
for k, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1])[:25]:
    print(f"{t:9.2f} ms {100*t/tot:5.1f}%  n={c:6d} avg={1e3*t/c:8.1f} us  {k}")
print("--- top kernels (by template+grid)")
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:45]:
    print(f"{t:9.2f} ms {100*t/tot:5.1f}%  n={c:6d} avg={1e3*t/c:8.1f} us  {k}")
