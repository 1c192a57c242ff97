"""Concurrency report of a rocprofv3 kernel trace (csv) of bench.py: over the last `--cycles` graph replays, the wall
time, the time during which at least one / at least two kernels ran, the idle gaps, and per kernel name the time it
spent alone on the GPU.  python tools/timeline.py <kernel_trace.csv> [--cycles 3]"""
import argparse
import csv
import collections

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--cycles", type=int, default=3)
ap.add_argument("--marker", default="adam_kernel", help="a kernel that runs a fixed number of times per cycle (6 for Adam)")
ap.add_argument("--per-cycle", type=int, default=6)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
marks = [e for e in ev if a.marker in e[2]]
need = a.cycles * a.per_cycle
if len(marks) < need + a.per_cycle:
    raise SystemExit(f"only {len(marks)} marker kernels in the trace")
t1 = marks[-1][1]
t0 = marks[-need - 1][1]          # end of the cycle before the window
win = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 1]
# sweep
pts = []
for s, e, n, q in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy1 = busy2 = busy3 = 0
depth, last = 0, t0
gaps = []
for t, d in pts:
    dt = t - last
    if depth >= 1: busy1 += dt
    if depth >= 2: busy2 += dt
    if depth >= 3: busy3 += dt
    if depth == 0 and dt > 0: gaps.append(dt)
    depth += d; last = t
wall = t1 - t0
ksum = sum(e - s for s, e, n, q in win)
print(f"window: {a.cycles} cycles, {len(win)} kernels, wall {wall / a.cycles / 1e6:.3f} ms/cycle, kernel time {ksum / a.cycles / 1e6:.3f} ms/cycle")
print(f"  >=1 kernel running {busy1 / a.cycles / 1e6:.3f} ms/cycle, >=2 {busy2 / a.cycles / 1e6:.3f}, >=3 {busy3 / a.cycles / 1e6:.3f}; idle {(wall - busy1) / a.cycles / 1e6:.3f} ms/cycle in {len(gaps) // a.cycles} gaps/cycle (median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, max {max(gaps) / 1e3:.1f} us)")
qs = collections.Counter()
for s, e, n, q in win:
    qs[q] += e - s
print("  busy per queue (ms/cycle):", {q: round(v / a.cycles / 1e6, 3) for q, v in qs.items()})
hist = collections.Counter()
for g in gaps:
    hist[min(int(g / 1000), 20)] += 1
print("  idle-gap histogram (us: count/cycle):", {k: round(v / a.cycles, 1) for k, v in sorted(hist.items())})
# time alone per kernel name: portion of each kernel's interval during which depth == 1
alone = collections.Counter(); total = collections.Counter(); cnt = collections.Counter()
ivs = sorted((s, e, n) for s, e, n, q in win)
bounds = sorted(set([s for s, e, n in ivs] + [e for s, e, n in ivs]))
import bisect
active = []
idx = 0
depth_at = {}
# depth per elementary segment
starts = collections.Counter(s for s, e, n in ivs); ends = collections.Counter(e for s, e, n in ivs)
d = 0
seg_depth = []
for i, b in enumerate(bounds[:-1]):
    d += starts[b] - ends[b]
    seg_depth.append(d)
for s, e, n in ivs:
    i0, i1 = bisect.bisect_left(bounds, s), bisect.bisect_left(bounds, e)
    al = sum(bounds[i + 1] - bounds[i] for i in range(i0, i1) if seg_depth[i] == 1)
    key = n.split("(")[0][-70:]
    alone[key] += al; total[key] += e - s; cnt[key] += 1
print("  kernel: launches/cycle, ms/cycle, of which alone on the GPU")
for k, v in total.most_common(25):
    print(f"    {k:70s} {cnt[k] / a.cycles:6.1f} {v / a.cycles / 1e6:7.3f} {alone[k] / a.cycles / 1e6:7.3f}")
