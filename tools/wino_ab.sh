#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_split_gemm_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "wino or conv4 or split" > gpurun_out/wino_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/wino_tests.log; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python tools/kbench.py --only wino --rounds 5 > gpurun_out/kbench_wino.log 2>&1 || { tail -30 gpurun_out/kbench_wino.log; exit 4; }
grep -E "^wino |_split|_fp32" gpurun_out/kbench_wino.log | cut -c1-112
