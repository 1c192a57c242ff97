#!/bin/bash
# Developer tool: per-layer conv timings under the diagnostic switches of the LDS-DMA gather kernel
# (GC_DIAG=1: loads only, GC_DIAG=2: multiply only; GC_FORCE_TILE / GC_DMA_VARIANT: tile and ring overrides).
set -e
mkdir -p gpurun_out
out=gpurun_out/diag_layers.txt
: > $out
run() { echo "=== $* ===" >> $out; env "$@" timeout -k 10 120 python tools/layer_bench.py --dtype bf16 --batch $B >> $out 2>&1; }
for B in 256 512; do
  run X=base
  run GC_DIAG=1
  run GC_DIAG=2
  run GC_DMA_VARIANT=deep
  run GC_DMA_VARIANT=plain
  run GC_DMA_VARIANT=ws
  run GC_FORCE_TILE=128,256
  run GC_FORCE_TILE=64,128
  run GC_FORCE_TILE=128,128
done
