#!/bin/bash
# 'medium' precision: Winograd products on 1-plane (2-byte) operands vs the fp32-storage bf16 path
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x -k "split or bf16 or medium or wino" > gpurun_out/medium_tests.log 2>&1
rc=$?; tail -n 6 gpurun_out/medium_tests.log; echo "tests rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for s in 0 1; do
  echo "== WFAE_SPLIT_GEMM=$s  bench.py --precision medium"
  WFAE_SPLIT_GEMM=$s timeout -k 10 300 python bench.py --precision medium --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_medium_$s.log 2> gpurun_out/bench_medium_$s.err || { tail -20 gpurun_out/bench_medium_$s.err; exit 4; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_medium_$s.log") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["peak_mem_GiB"], d["dtype"])
for k in d["kernel_breakdown"][:12]: print("  %-34s %6.0f %7.2f ms"%(k["entry_point"],k["calls_per_step"],k["ms_per_step"]))
PY
done
