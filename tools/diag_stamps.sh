#!/bin/bash
# Developer tool: in-kernel cycle stamps (prologue / main loop / epilogue) of the LDS-DMA gather kernel per layer.
# Needs the diagnostic build of the library: make -C gan_amd/csrc clean && make -C gan_amd/csrc DIAG=1 (the shipped build
# has no allocating / synchronising code path).
mkdir -p gpurun_out
out=gpurun_out/diag_stamps.txt
: > $out
run() { echo "=== $* ===" >> $out; env "$@" GC_DIAG_STAMPS=1 timeout -k 10 120 python tools/layer_bench.py --dtype bf16 --batch $B 2>&1 | grep -v amdgpu.ids >> $out; }
for B in 256 512; do
  run X=base
  run GC_DIAG=1
  run GC_DIAG=2
  run GC_DMA_VARIANT=ws
done
