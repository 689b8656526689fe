#!/bin/bash
# 200 steps of the ae_v2 experiment script (B = 32, 384x384, synthetic blob events through the loader) with the split GEMMs
# and with every GEMM on the fp32 MFMA instruction: loss trajectories side by side
mkdir -p gpurun_out; export TMPDIR=/tmp
for s in 1 0; do
  WFAE_SPLIT_GEMM=$s timeout -k 10 500 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --max-steps 200 dataset.name=sevir dataset.batch_size=32 experiment_path=gpurun_out/cmp_$s > gpurun_out/train_cmp_$s.log 2>&1 || { tail -5 gpurun_out/train_cmp_$s.log; exit 3; }
  rm -rf gpurun_out/cmp_$s
done
python3 - <<'PY'
import json
def load(p):
    out={}
    for l in open(p):
        if l.startswith("{") and '"step"' in l:
            d=json.loads(l); out[d["step"]]=d
    return out
a,b=load("gpurun_out/train_cmp_1.log"),load("gpurun_out/train_cmp_0.log")
ks=sorted(set(a)&set(b))
key=["train/rec_loss"]
print("steps logged", len(ks), "key", key)
for s in ks:
    if s in (0,1,2,5,10,20,50,100,150,199) or s==ks[-1]:
        print(s, {k:(round(a[s][k],6), round(b[s][k],6)) for k in key}, "ms", round(a[s]["ms"],1), round(b[s]["ms"],1))
PY
