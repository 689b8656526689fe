#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/dp.log 2>&1; rc=$?; tail -5 gpurun_out/dp.log; [ $rc -gt 1 ] && exit $rc
timeout -k 10 300 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --max-steps 4 dataset.batch_size=4 experiment_path=gpurun_out/train_tf > gpurun_out/train_tf.log 2>&1; echo "train tf rc=$?"; tail -3 gpurun_out/train_tf.log
timeout -k 10 300 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --max-steps 4 dataset.batch_size=4 lpips.perceptual_weight=0.5 experiment_path=gpurun_out/train_lin > gpurun_out/train_lin.log 2>&1; echo "train lin rc=$?"; tail -3 gpurun_out/train_lin.log
timeout -k 10 300 python -m weatherforecastingtoolkit_amd.experiments.ae_v2.train --model lin --resume True --max-steps 6 dataset.batch_size=4 experiment_path=gpurun_out/train_lin > gpurun_out/train_lin2.log 2>&1; echo "resume rc=$?"; tail -2 gpurun_out/train_lin2.log
rm -rf gpurun_out/train_tf/outputs gpurun_out/train_lin/outputs
