#!/bin/bash
# kernel+model tests -> bench (with cpu baseline) -> kernel-trace stats -> two PMC passes (FETCH_SIZE, WRITE_SIZE)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/gpu_tests.log; echo "gpu tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 4; }
cat gpurun_out/bench.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap --no-fp32-leg > $R/gpurun_out/prof.log 2>&1 || { tail -30 $R/gpurun_out/prof.log; exit 5; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap --no-fp32-leg > $R/gpurun_out/pmc_fetch.log 2>&1 || { tail -30 $R/gpurun_out/pmc_fetch.log; exit 6; }
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap --no-fp32-leg > $R/gpurun_out/pmc_write.log 2>&1 || { tail -30 $R/gpurun_out/pmc_write.log; exit 7; }
cd $R && find gpurun_out -name "*.csv" | head -20
