#!/bin/bash
# where does the split GEMM lose time?  1: no global loads in the loop, 3: + no LDS stores, 7: + no barrier
mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 0 1 3 7; do
  echo "== WFAE_SPLIT_DBG=$d"
  WFAE_SPLIT_DBG=$d timeout -k 10 200 python tools/kbench.py --only wino --rounds 3 2>&1 | grep "gemm.*split" || exit 3
done
