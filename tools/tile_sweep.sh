#!/bin/bash
# Developer tool: per-layer conv timings of the benchmark's shapes under tile / ring-depth overrides of the
# wave-specialised gather kernel (GC_TO64_MAX_TILES: largest 128-channel tile count that still switches to 64-channel
# tiles; GC_WS_STAGES: ring depth).
set -e
mkdir -p gpurun_out
out=gpurun_out/tile_sweep.txt
: > $out
run() { echo "=== $* ===" >> $out; env "$@" timeout -k 10 120 python tools/layer_bench.py --dtype bf16 --batch $B >> $out 2>&1; }
for B in 256 512; do
  run X=base
  run GC_TO64_MAX_TILES=0 GC_WS_STAGES=2
  run GC_TO64_MAX_TILES=0 GC_WS_STAGES=3
  run GC_TO64_MAX_TILES=0 GC_WS_STAGES=4
  run GC_TO64_MAX_TILES=100000 GC_WS_STAGES=3
  run GC_TO64_MAX_TILES=100000 GC_WS_STAGES=4
done
