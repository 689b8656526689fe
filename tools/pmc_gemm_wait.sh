#!/bin/bash
# Where do the waves of the GEMM kernels spend their cycles?  SQ wait / issue-stall / active split (MI355X_MICROARCH.md,
# rocprofv3 PMC slots: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES) for tools/kbench.py --only $1
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; W=${1:-conv4}
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wait_$W -- python3 $R/tools/kbench.py --only $W --rounds 1 > $R/gpurun_out/pmc_wait_$W.log 2>&1 || { tail -20 $R/gpurun_out/pmc_wait_$W.log; exit 3; }
cd $R && python3 - $W <<'PY'
import csv, glob, collections, os, sys
W=sys.argv[1]
f=sorted(glob.glob(f'gpurun_out/pmc_wait_{W}/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
d=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'gemm_kernel<' not in r['Kernel_Name'] and 'sgemm3_' not in r['Kernel_Name']: continue
    k=r['Dispatch_Id']
    d[k]['name']=r['Kernel_Name'].split('::')[-1].split('(')[0][:40]
    d[k][r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    d[k]['grid']=r.get('Grid_Size','')
agg=collections.defaultdict(lambda: collections.defaultdict(float))
C=('dur','GRBM_GUI_ACTIVE','SQ_VALU_MFMA_BUSY_CYCLES','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAVES','SQ_BUSY_CYCLES','SQ_INSTS_VALU')
for v in d.values():
    a=agg[v['name']]; a['n']+=1
    for c in C: a[c]+=v.get(c,0.0)
out=[f"# rocprofv3 --pmc {' '.join(C[1:])} -- python3 tools/kbench.py --only {W} --rounds 1 (B=32, 384^2 layer shapes; sums per gemm_kernel instantiation)",
     "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); wave-cycle split = share of SQ_WAVE_CYCLES; waves/SIMD = SQ_WAVE_CYCLES*4 / (GRBM_GUI_ACTIVE/8*1024) (quad-cycles)"]
for name,a in sorted(agg.items(), key=lambda kv:-kv[1]['dur']):
    gui=a['GRBM_GUI_ACTIVE']/8.0; wc=max(a['SQ_WAVE_CYCLES'],1)
    out.append(f"{name:40s} launches {int(a['n']):3d} time {a['dur']/1e3:8.2f} ms mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES']/max(gui*1024,1):5.2f} "
          f"wait_any {a['SQ_WAIT_ANY']/wc:5.2f} wait_inst {a['SQ_WAIT_INST_ANY']/wc:5.2f} active {a['SQ_ACTIVE_INST_ANY']/wc:5.2f} waves/SIMD {4*wc/max(gui*1024,1):5.2f} clock_GHz {gui/max(a['dur'],1)/1e3:5.2f}")
open(f'gpurun_out/pmc_wait_{W}.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
PY
