#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_gan_gpu.py tests/test_loader_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/train_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/train_tests.log; echo "tests rc=$rc"; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --max-steps 1000 --stop-after 80 dataset.name=sevir dataset.batch_size=32 experiment_path=gpurun_out/fps_run > gpurun_out/train_fps.log 2>&1 || { tail -5 gpurun_out/train_fps.log; exit 3; }
rm -rf gpurun_out/fps_run
python3 - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/train_fps.log") if l.startswith("{") and '"step"' in l]
for r in rows:
    if r["step"] in (1,2,5,10,20,40,60,80): print(r["step"], round(r["train/rec_loss"],5), round(r["frames_per_s"],1))
PY
