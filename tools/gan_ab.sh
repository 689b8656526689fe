#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gan_gpu.py tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "gan or conv4x4s1 or s1" > gpurun_out/gan_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/gan_tests.log; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/gan_bench.py --steps 4 --warmup 2 2>&1 | grep -v amdgpu.ids | head -14
