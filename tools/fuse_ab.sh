#!/bin/bash
# tests, then the step with a feature on / off (interleaved)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/gpu_tests.log; echo "gpu tests rc=$rc"; if [ $rc -ne 0 ]; then grep -v "^{" gpurun_out/gpu_tests.log | grep -B5 -A25 "Error\|assert" | head -80; exit $rc; fi
: > gpurun_out/fuse_ab.log
for cfg in $AB_CONFIGS; do
  echo "== $cfg" | tee -a gpurun_out/fuse_ab.log
  env $cfg timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['peak_mem_GiB'])" | tee -a gpurun_out/fuse_ab.log || exit 4
done
