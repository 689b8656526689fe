#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/small_ab.log; : > $L
for cfg in "X=0" "WFAE_FUSE_A1=0 WFAE_FUSE_A3=0" "WFAE_PRODUCER_STATS=0" "WFAE_FUSE_A1=0 WFAE_FUSE_A3=0 WFAE_PRODUCER_STATS=0" "X=0" "WFAE_FUSE_A1=0 WFAE_FUSE_A3=0"; do
  echo "== $cfg" >> $L
  env $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 8 --batch 8 --img-size 128 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 3
done
for cfg in "X=0" "WFAE_FUSE_A1=0 WFAE_FUSE_A3=0"; do
  echo "== B=2 384 $cfg" >> $L
  env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 2 --img-size 384 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L || exit 3
done
cat $L
