"""Prints the per-kernel table of a bench.py JSON line (file given as argv[1])."""
import json
import sys

line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(line)
ks = d["roofline"]["all_kernels"]
tot = 0.0
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_us"]):
    t = v["launches"] * v["avg_us"] / 2 / 1000
    tot += t
    print(f"{k:80s} n/cyc={v['launches'] // 2:4d} avg={v['avg_us']:7.1f}us  ms/cyc={t:6.2f}  {v['tflops']:6.1f} TF")
print(f"library kernels {tot:.2f} ms/cycle; step {d['ms_per_step']:.2f} ms; {d['value']:.0f} {d['unit']}; cpu {d.get('cpu_baseline')}")
