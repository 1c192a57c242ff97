#!/bin/bash
# usage (GPU box, repository root): tools/family_profile.sh <family> [--full]   -> gpurun_out/fam_<family>_stats.csv
set -e -o pipefail
fam=$1; shift
out=gpurun_out/fam_$fam
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/family_bench.py --only=$fam --no-cpu "$@" > $out/log.txt 2>&1
cp $(find $out -name '*kernel_stats.csv') gpurun_out/fam_${fam}_stats.csv
rm -rf $out
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/fam_${fam}_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over the run", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{100 * float(r["TotalDurationNs"]) / tot:6.2f}%  n={int(r["Calls"]):6d}  avg={float(r["AverageNs"]) / 1e3:9.1f}us  {r["Name"][:130]}')
PY
