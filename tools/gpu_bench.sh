#!/bin/bash
# model tests -> smoke -> bench -> rocprofv3 kernel trace of a short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/model.log 2>&1
rc=$?; tail -n 15 gpurun_out/model.log; echo "model tests rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 3; }
tail -n 2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py ${BENCH_ARGS:---steps 5 --warmup 2} > gpurun_out/bench.log 2> gpurun_out/bench.err || { tail -30 gpurun_out/bench.err; exit 4; }
cat gpurun_out/bench.log
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof.log; exit 5; }
cd $GRAFT_REPO_ROOT && find gpurun_out/prof -name "*stats*" | head; tail -3 gpurun_out/prof.log
