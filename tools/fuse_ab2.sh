#!/bin/bash
# the step with a feature on / off (no tests): AB_CONFIGS="VAR=a VAR=b ..."
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/fuse_ab2.log
for cfg in $AB_CONFIGS; do
  echo "== $cfg" | tee -a gpurun_out/fuse_ab2.log
  env ${cfg//,/ } timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fp32-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['peak_mem_GiB'])" | tee -a gpurun_out/fuse_ab2.log || exit 4
done
