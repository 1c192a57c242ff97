"""Developer tool: per-kernel HBM traffic per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts a 128-byte request of a wide (16 bytes per
lane) streaming read as 64 bytes, so it is doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            if k.startswith("void "):
                k = k[5:]
            k = k.replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"]) * 1024.0
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if not k.startswith("gc::"):
        continue
    n, fb = fetch[k]
    wn, wb = write.get(k, [0, 0.0])
    out[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * fb / n, "write_bytes_per_launch": wb / max(wn, 1),
              "hbm_bytes_per_launch": 2.0 * fb / n + wb / max(wn, 1)}
# every kernel of the run, the framework's own included (elementwise glue, GEMMs of the linear layers)
out["__all_kernels__"] = {"launches": sum(v[0] for v in fetch.values()),
                          "hbm_bytes_total": 2.0 * sum(v[1] for v in fetch.values()) + sum(v[1] for v in write.values())}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
del out["__all_kernels__"]
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
    print(f"{k:70s} n={v['launches']:4d} fetch={v['fetch_bytes_per_launch'] / 1e6:8.2f} MB write={v['write_bytes_per_launch'] / 1e6:8.2f} MB")
