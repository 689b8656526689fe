#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_vit_gpu.py tests/test_bench_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/gpu_tests.log; echo "gpu tests rc=$rc"; if [ $rc -ne 0 ]; then grep -v "^{" gpurun_out/gpu_tests.log | grep -B5 -A25 "Error\|assert" | head -80; exit $rc; fi
timeout -k 10 200 python tools/kbench.py --only dconv --rounds 5 2>&1 | grep -v amdgpu.ids > gpurun_out/kb_dconv.log; cat gpurun_out/kb_dconv.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_g3.log 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_g3.log')); print(d['ms_per_step'], d['value'], d['peak_mem_GiB'])
for k in d['kernel_breakdown']: print(f\"{k['entry_point']:34s} {k['calls_per_step']:6.0f} {k['ms_per_step']:7.2f} ms  {k['tflops'] or 0:6.1f} TF {k['gbps']:7.0f} GB/s\")"
