#!/bin/bash
export TMPDIR=/tmp
for k in 128 64 32; do for m in 64 32; do
  echo "== WFAE_SPLIT_MIN_K=$k WFAE_SPLIT_MIN_M=$m"
  WFAE_SPLIT_MIN_K=$k WFAE_SPLIT_MIN_M=$m timeout -k 10 200 python tools/kbench.py --only conv1 --rounds 5 2>&1 | grep -E "^conv1_" || exit 3
done; done
