"""Where is the GPU idle?  Reads a rocprofv3 kernel trace (CSV) and prints, for the busiest window of the run, the union busy
time, the idle time and the largest idle gaps with the kernels before / after them.
    python tools/timeline_gaps.py <..._kernel_trace.csv> [--last-ms 500] [--top 25]"""
import csv
import re
import sys

path = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 450.0
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void |\(anonymous namespace\)::", "", n))[:70]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
for s, e, n, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - rows[0][0], last_name, n))
        cur_s, cur_e = s, e
        last_name = n
    else:
        if e > cur_e:
            cur_e, last_name = e, n
busy += cur_e - cur_s
print(f"window {span / 1e6:.1f} ms, {len(rows)} kernels, busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
hist = [0, 0, 0, 0]
for g in gaps:
    hist[0 if g[0] < 5e3 else 1 if g[0] < 50e3 else 2 if g[0] < 500e3 else 3] += g[0]
print("idle by gap length: <5us %.2f ms, 5-50us %.2f ms, 50-500us %.2f ms, >500us %.2f ms" % tuple(h / 1e6 for h in hist))
queues = {}
for s, e, n, q in rows:
    queues.setdefault(q, [0, 0])
    queues[q][0] += 1; queues[q][1] += e - s
print("per queue:", {q: (c, round(t / 1e6, 1)) for q, (c, t) in queues.items()})
for g, at, a, b in sorted(gaps, reverse=True)[:top]:
    print(f"  gap {g / 1e3:8.1f} us at +{at / 1e6:7.2f} ms  after {short(a)}  before {short(b)}")
