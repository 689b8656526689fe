#!/bin/bash
# Round-trip GPU check used with gpurun: kernel parity tests, then model tests.
# Stops after a step that was killed / timed out (exit code > 1).
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rA -p no:cacheprovider > gpurun_out/kernels.log 2>&1
rc=$?
tail -n 60 gpurun_out/kernels.log
echo "kernel tests rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 500 python -m pytest tests/test_model_gpu.py -m gpu -q --no-header -rA -p no:cacheprovider > gpurun_out/model.log 2>&1
rc2=$?
tail -n 60 gpurun_out/model.log
echo "model tests rc=$rc2"
exit $(( rc > rc2 ? rc : rc2 ))
