#!/bin/bash
# usage: tools/pmc_layer.sh <tag> <one_layer args...>   (run on the GPU box from the repo root)
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM"
P2="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $out -o p$i -- python3 tools/one_layer.py "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
echo ok $tag
