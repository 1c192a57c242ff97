#!/bin/bash
# Developer tool: weight-gradient timings of the benchmark's layers under planner overrides
# (GC_WGRAD_TO256: 256-output-channel tiles wherever Cout allows; GC_WGRAD_WGS: workgroups the split count aims at).
set -e
mkdir -p gpurun_out
out=gpurun_out/wgrad_sweep.txt
: > $out
run() { echo "=== $* ===" >> $out; env "$@" timeout -k 10 120 python tools/layer_bench.py --dtype bf16 --batch $B 2>&1 | grep "^C" | sed 's/.*| wgrad/wgrad/' | paste -sd' ' >> $out; }
for B in 256 512; do
  echo "--- batch $B (layers: 3->64, 64->128, 128->256, 256->512, 512->1, 512->100)" >> $out
  run X=base
  run GC_WGRAD_WGS=256
  run GC_WGRAD_WGS=768
  run GC_WGRAD_WGS=1024
  run GC_WGRAD_TO256=1
  run GC_WGRAD_TO256=1 GC_WGRAD_WGS=256
done
