#!/bin/bash
# Collects one round's measurement set on the GPU box (run from the repository root through gpurun):
#   tools/profile_round.sh <tag>        e.g. r1_e
# Writes gpurun_out/<tag>/: the unprofiled bench line, the rocprofv3 kernel-trace statistics of the same command with
# --one-stream (the same launches with the chains serialised: each kernel alone on the GPU, the durations the roofline
# figures are about) and of the default command
# (chains of a critic iteration on concurrent streams: kernels share the GPU and each one takes longer), two counter
# passes (FETCH_SIZE, WRITE_SIZE; counters never share a run with a trace) over one eager one-stream cycle, and the
# summary files tools/profile_summary.py derives from them.
set -e -o pipefail
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench_line.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-families --one-stream > $out/bench_line_under_rocprof.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-families > $out/bench_line_under_rocprof_concurrent.json
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-families --one-stream > /dev/null
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-families --one-stream > /dev/null
python3 tools/pmc_traffic.py $(find $out/fetch -name '*counter_collection.csv') $(find $out/write -name '*counter_collection.csv') $out/traffic.json > $out/traffic.txt
cp $(find $out/trace -name '*kernel_stats.csv') $out/bench_kernel_stats.csv
cp $(find $out/trace2 -name '*kernel_stats.csv') $out/bench_kernel_stats_concurrent.csv
python3 tools/profile_summary.py $out $tag > $out/bench_summary.md
rm -rf $out/trace $out/trace2 $out/fetch $out/write
tail -c 600 $out/bench_line.json
