#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/overlap_ab.log; : > $L
run() { timeout -k 10 120 env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $L || exit 3; }
run python3 tools/overlap_probe.py wino
run python3 tools/overlap_probe.py wino prio
run python3 tools/overlap_probe.py c1wgrad
run python3 tools/overlap_probe.py c1wgrad prio
run WFAE_GEMM_DYNLDS=8192 python3 tools/overlap_probe.py c1wgrad
run WFAE_GEMM_DYNLDS=20480 python3 tools/overlap_probe.py c1wgrad
run WFAE_GEMM_DYNLDS=20480 python3 tools/overlap_probe.py c1wgrad prio
run python3 tools/overlap_probe.py c1fwd
run WFAE_GEMM_DYNLDS=20480 python3 tools/overlap_probe.py c1fwd
cat $L
