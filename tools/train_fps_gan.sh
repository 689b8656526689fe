#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m weatherforecastingtoolkit_amd.experiments.ae_v2_2.train --max-steps 1000 --stop-after 60 dataset.batch_size=32 lpips.disc_start=0.0 experiment_path=gpurun_out/fps_gan > gpurun_out/train_fps_gan.log 2>&1 || { tail -8 gpurun_out/train_fps_gan.log; exit 3; }
rm -rf gpurun_out/fps_gan
python3 - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/train_fps_gan.log") if l.startswith("{") and '"step"' in l]
for r in rows:
    if r["step"] in (1,2,5,10,20,40,60): print(r["step"], {k:round(v,4) for k,v in r.items() if k in ("train/rec_loss","train/disc_loss")}, round(r["frames_per_s"],1))
PY
