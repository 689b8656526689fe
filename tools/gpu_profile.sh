#!/bin/bash
# bench (with cpu baseline) -> kernel-trace stats -> two PMC passes (FETCH_SIZE, WRITE_SIZE), for the fp32 headline configuration
# and for --precision medium (bf16 activation storage).  PMC passes run WITHOUT tracing domains other than --kernel-trace.
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-kernel-timing --no-overlap --no-fp32-leg --no-medium-leg"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 4; }
timeout -k 10 300 python bench.py --precision medium --steps 20 --warmup 5 --no-cpu-baseline --no-medium-leg > gpurun_out/bench_medium.log 2>> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 4; }
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/prof.log 2>&1 || { tail -30 $R/gpurun_out/prof.log; exit 5; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1 || { tail -30 $R/gpurun_out/pmc_fetch.log; exit 6; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 $ARGS > $R/gpurun_out/pmc_write.log 2>&1 || { tail -30 $R/gpurun_out/pmc_write.log; exit 7; }
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_medium -- python3 $R/bench.py --precision medium --steps 2 --warmup 1 $ARGS > $R/gpurun_out/prof_medium.log 2>&1 || { tail -30 $R/gpurun_out/prof_medium.log; exit 8; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_medium -- python3 $R/bench.py --precision medium --steps 1 --warmup 1 $ARGS > $R/gpurun_out/pmc_fetch_medium.log 2>&1 || { tail -30 $R/gpurun_out/pmc_fetch_medium.log; exit 9; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_medium -- python3 $R/bench.py --precision medium --steps 1 --warmup 1 $ARGS > $R/gpurun_out/pmc_write_medium.log 2>&1 || { tail -30 $R/gpurun_out/pmc_write_medium.log; exit 10; }
cd $R && find gpurun_out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
