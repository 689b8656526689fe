#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "bn or norm or batchnorm" > gpurun_out/bn_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/bn_tests.log; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/kbench.py --only bn --rounds 7 2>&1 | grep -v amdgpu.ids | cut -c1-130
