#!/bin/bash
# 200 steps of the ae_v2 experiment script (B = 32, 384x384, synthetic blob events through the loader) at fp32 precision and at
# 'medium' (bf16 activation storage, bf16 MFMA operands): loss trajectories and frames/s side by side
mkdir -p gpurun_out; export TMPDIR=/tmp
for p in highest medium; do
  timeout -k 10 500 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --max-steps 200 --matmul-precision $p dataset.name=sevir dataset.batch_size=32 experiment_path=gpurun_out/cmp_$p > gpurun_out/train_cmp_$p.log 2>&1 || { tail -5 gpurun_out/train_cmp_$p.log; exit 3; }
  rm -rf gpurun_out/cmp_$p
done
python3 - <<'PY'
import json
def load(p):
    out={}
    for l in open(p):
        if l.startswith("{") and '"step"' in l:
            d=json.loads(l); out[d["step"]]=d
    return out
a,b=load("gpurun_out/train_cmp_highest.log"),load("gpurun_out/train_cmp_medium.log")
ks=sorted(set(a)&set(b))
print("# python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --max-steps 200 --matmul-precision {highest,medium} dataset.name=sevir dataset.batch_size=32")
print("# step: train/rec_loss (fp32, medium = bf16 activation storage), frames/s through the loader")
for s in ks:
    if (s in (0,1,2,5,10,20,50,100,150,199) or s==ks[-1]) and "train/rec_loss" in a[s] and "train/rec_loss" in b[s]:
        print(s, round(a[s]["train/rec_loss"],6), round(b[s]["train/rec_loss"],6), "frames/s", round(a[s].get("frames_per_s",0),1), round(b[s].get("frames_per_s",0),1))
PY
