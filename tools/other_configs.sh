#!/bin/bash
# numbers for the configurations other than the headline: 'medium' (bf16 MFMA operands) AE step, AE+GAN step fp32 / medium,
# Path-B ViT step, B = 8 at 128^2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/other_configs.log; : > $L
echo "== bench.py --precision medium" >> $L
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --precision medium 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['peak_mem_GiB'], d['dtype'])
for k in d['kernel_breakdown'][:12]: print(f\"  {k['entry_point']:34s} {k['calls_per_step']:6.0f} {k['ms_per_step']:7.2f} ms  {k['tflops'] or 0:6.1f} TF {k['gbps']:7.0f} GB/s\")" >> $L || exit 3
echo "== gan_bench fp32" >> $L
timeout -k 10 300 python tools/gan_bench.py --steps 4 --warmup 2 2>&1 | grep -v amdgpu.ids | head -12 >> $L || exit 4
echo "== gan_bench medium" >> $L
timeout -k 10 300 python tools/gan_bench.py --steps 4 --warmup 2 --precision medium 2>&1 | grep -v amdgpu.ids | head -3 >> $L || exit 5
echo "== vit_bench" >> $L
timeout -k 10 200 python tools/vit_bench.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $L
echo "== bench.py B=8 128x128" >> $L
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --batch 8 --img-size 128 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $L
cat $L
