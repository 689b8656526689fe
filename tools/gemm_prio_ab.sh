#!/bin/bash
# A/B of the GEMM wave-priority policies (WFAE_GEMM_PRIO) on the real layer shapes: tools/kbench.py once per policy
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for p in 0 1 2 3; do
  echo "==== WFAE_GEMM_PRIO=$p" >> gpurun_out/prio_ab.log
  WFAE_GEMM_PRIO=$p timeout -k 10 240 python3 tools/kbench.py --only conv4,conv1 --rounds 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/prio_ab.log || exit 3
done
grep -A4 "per-step totals\|====" gpurun_out/prio_ab.log | grep -v "^--$"
