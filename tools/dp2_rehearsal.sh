#!/bin/bash
# two ranks on the one card over gloo: bench.py's multi-rank path (barriers, max over ranks, dp object, fp32-MFMA-only leg)
mkdir -p gpurun_out; export TMPDIR=/tmp
WFAE_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch ${DP_BATCH:-8} --img-size ${DP_SIZE:-384} > gpurun_out/bench_dp2.log 2> gpurun_out/bench_dp2.err
rc=$?; tail -3 gpurun_out/bench_dp2.err | cut -c1-300; grep "^{" gpurun_out/bench_dp2.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling','dp','fp32_mfma_only')}); print(d['config']['parallelism'], d['config']['global_batch'])"
exit $rc
