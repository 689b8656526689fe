#!/bin/bash
# in-register split (gemm_kernel PREC 2) on the 1x1 GEMM shapes: kernel tests, then kbench with the split on / off
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_split_gemm_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/split1_tests.log 2>&1
rc=$?; tail -n 15 gpurun_out/split1_tests.log; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for s in 0 1; do
  echo "== WFAE_SPLIT_GEMM=$s"
  WFAE_SPLIT_GEMM=$s timeout -k 10 300 python tools/kbench.py --only ${AB_ONLY:-conv1} --rounds 5 > gpurun_out/kbench_split1_$s.log 2>&1 || { tail -20 gpurun_out/kbench_split1_$s.log; exit 4; }
  grep -E "conv1x1|per-step|^conv1_|fuse" gpurun_out/kbench_split1_$s.log | cut -c1-118
done
