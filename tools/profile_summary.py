"""Writes the markdown summary of one measurement set collected by tools/profile_round.sh:
    python tools/profile_summary.py <directory> <tag>"""
import csv
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]


def last_json(path):
    return json.loads([x for x in open(path) if x.startswith("{")][-1])


line, prof = last_json(os.path.join(d, "bench_line.json")), last_json(os.path.join(d, "bench_line_under_rocprof.json"))
r = line["roofline"]
cycles = prof["warmup"] + prof["steps"] + 2 + 2      # replays (warm-up + timed), 2 eager cycles before the recording, 2 profiled eager cycles
rows = list(csv.DictReader(open(os.path.join(d, "bench_kernel_stats.csv"))))
tot = sum(float(x["TotalDurationNs"]) for x in rows) / 1e6 / cycles
lib = sum(float(x["TotalDurationNs"]) for x in rows if x["Name"].removeprefix("void ").startswith("gc::")) / 1e6 / cycles
print(f"# Measurement set {tag}: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {prof['steps']} --warmup {prof['warmup']} --one-stream` on one MI355X\n")
cb = line.get("cpu_baseline", {})
print(f"Unprofiled default run (`python bench.py`, `{tag}_bench_line.json`): **{line['value']:.0f} img/s, {line['ms_per_step']:.2f} ms per cycle**; "
      f"dominant conv kernel `{r['kernel']}`: {r['achieved']:.0f} TFLOP/s = {r['frac']:.3f} of the dense bf16 peak, average launch "
      f"{r['avg_launch_us']:.1f} us" + (f", {r['traffic'] / 1e6:.1f} MB HBM traffic per launch" if r.get("traffic") else "") +
      f"; CPU baseline {cb.get('value', float('nan')):.1f} img/s on {cb.get('cores')} cores.\n")
print(f"Under the profiler with `--one-stream` (the launches of the default command, its chains one after the other on one stream, "
      f"so every kernel is alone on the GPU: the durations below are what the roofline figure is about): {prof['value']:.0f} img/s, {prof['ms_per_step']:.2f} ms per cycle.  {cycles} cycles were traced "
      f"(the per-cycle columns divide by {cycles}).")
cpath = os.path.join(d, "bench_line_under_rocprof_concurrent.json")
if os.path.exists(cpath):
    conc = last_json(cpath)
    crow = {x["Name"]: x for x in csv.DictReader(open(os.path.join(d, "bench_kernel_stats_concurrent.csv")))}
    key = next((k for k in crow if r["kernel"] in k), None)
    one = next((x for x in rows if r["kernel"] in x["Name"]), None)
    extra = ""
    if key and one:
        extra = (f"; the dominant kernel's average launch there is {float(crow[key]['TotalDurationNs']) / int(crow[key]['Calls']) / 1e3:.1f} us "
                 f"against {float(one['TotalDurationNs']) / int(one['Calls']) / 1e3:.1f} us with one stream")
    print(f"The default command (independent chains of a critic iteration on concurrent streams) under the profiler: {conc['value']:.0f} img/s, "
          f"{conc['ms_per_step']:.2f} ms per cycle (`{tag}_bench_kernel_stats_concurrent.csv`){extra}.  Kernel tracing takes the "
          f"concurrency away (the two modes run at the same speed under the profiler and the kernel durations agree), so the gain of the "
          f"concurrent chains shows only in unprofiled runs.")
print(f"GPU kernel time per cycle {tot:.2f} ms, of which library (`gc::`) kernels {lib:.2f} ms ({100 * lib / tot:.0f} %).\n")
print("| kernel | launches / cycle | avg µs | ms / cycle | % |\n|---|---|---|---|---|")
for x in sorted(rows, key=lambda x: -float(x["TotalDurationNs"]))[:36]:
    n, t = int(x["Calls"]), float(x["TotalDurationNs"])
    print(f"| `{x['Name'].replace('(anonymous namespace)::', '').split('(')[0][:92]}` | {n / cycles:.1f} | {t / n / 1e3:.1f} | {t / 1e6 / cycles:.3f} | {100 * t / 1e6 / cycles / tot:.2f} |")
print(f"\nFull table: `{tag}_bench_kernel_stats.csv`.  HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of one "
      f"eager cycle, FETCH_SIZE doubled as the gfx950 counter tallies a 128-byte request as 64): `{tag}_traffic.json` (copied to "
      "`bench_traffic.json`, which `bench.py` reads).\n\n```")
print(open(os.path.join(d, "traffic.txt")).read().rstrip())
print("```")
