#!/bin/bash
# second counter set; usage as tools/pmc_layer.sh
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
P4="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU"
P5="SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
i=4
for P in "$P4" "$P5"; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $out -o p$i -- python3 tools/one_layer.py "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
echo ok $tag
