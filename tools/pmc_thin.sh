#!/bin/bash
# Developer tool: instruction and wait counters of the first-layer kernels (tools/thin_bench.py at batch 256).
set -e
root=$PWD
out=$root/gpurun_out/pmc_thin
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d $out -o p1 -- python3 tools/thin_bench.py --batch 256 > $out/p1.log 2>&1
python3 tools/pmc_summary.py $out
