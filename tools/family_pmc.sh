#!/bin/bash
# Kernel statistics and counter passes of ONE model family's training step (run on the GPU box from the repository root):
#   tools/family_pmc.sh <tag> <family of bench.py --family: stylegan256_b32 | biggan128_b256 | pix2pix256_b8 | ...>
# Writes gpurun_out/<tag>/: kernel_stats.csv (rocprofv3 --kernel-trace --stats of eager steps), traffic.{json,txt}
# (FETCH_SIZE / WRITE_SIZE passes, per kernel and launch) and sq.txt (matrix-core, wait and LDS counters per kernel).
# Every counter pass is a run of its own; counters never share a run with a trace.
# BigGAN: run with GC_BIGGAN_INIT_ON_CPU=1 (the counter passes do not survive the solver library's QR kernels of the
# orthogonal initialisation; with it the factorisations run on the host).
set -e -o pipefail
tag=$1; fam=$2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
steps=${STEPS:-3}; warmup=${WARMUP:-1}
total=$((steps + warmup + 1))                                                 # eager training steps per run (timed + warm-up + the profiled pass)
cmd="python3 bench.py --family $fam --no-graph --steps $steps --warmup $warmup"      # eager training steps (both phases, Adam)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $cmd > $out/family_table.txt 2>&1
cp $(find $out/trace -name '*kernel_stats.csv') $out/kernel_stats.csv
timeout -s KILL ${PMC_LIMIT:-240} rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $cmd > /dev/null 2>&1
timeout -s KILL ${PMC_LIMIT:-240} rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $cmd > /dev/null 2>&1
python3 tools/pmc_traffic.py $(find $out/fetch -name '*counter_collection.csv') $(find $out/write -name '*counter_collection.csv') $out/traffic.json > $out/traffic.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout -s KILL ${PMC_LIMIT:-240} rocprofv3 --pmc $P1 --output-format csv -d $out/sq1 -o p -- $cmd > /dev/null 2>&1
timeout -s KILL ${PMC_LIMIT:-240} rocprofv3 --pmc $P2 --output-format csv -d $out/sq2 -o p -- $cmd > /dev/null 2>&1
python3 tools/pmc_summary.py $out/sq1 $out/sq2 > $out/sq.txt
rm -rf $out/trace $out/fetch $out/write $out/sq1 $out/sq2
python3 - <<PY
import csv, json
# HBM bytes per step: every kernel of the run (5 eager steps), FETCH_SIZE doubled as for the per-kernel table
tr = json.load(open("$out/traffic.json"))
per_step = tr.pop("__all_kernels__")["hbm_bytes_total"] / $total
print("HBM GB per step, every kernel:", per_step / 1e9, " library kernels:", sum(v["hbm_bytes_per_launch"] * v["launches"] for v in tr.values()) / $total / 1e9)
json.dump({"$fam": {"hbm_bytes_per_step": per_step, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over $total eager training steps, FETCH_SIZE doubled (gfx950)"}},
          open("$out/family_traffic.json", "w"), indent=1)
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over the run", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print(f'{100 * float(r["TotalDurationNs"]) / tot:6.2f}%  n={int(r["Calls"]):6d}  avg={float(r["AverageNs"]) / 1e3:9.1f}us  {r["Name"][:120]}')
PY
