#!/bin/bash
# split-operand GEMM: correctness tests, then the per-piece Winograd timings fp32 vs split
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_split_gemm_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/split_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/split_tests.log; echo "split tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python tools/kbench.py --only wino --rounds 5 > gpurun_out/kbench_wino.log 2>&1 || { tail -30 gpurun_out/kbench_wino.log; exit 4; }
cat gpurun_out/kbench_wino.log
