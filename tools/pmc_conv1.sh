#!/bin/bash
# PMC look at the 1x1-convolution GEMMs (tools/kbench.py --only conv1): MFMA busy share, LDS wait / conflicts, instruction mix
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c1 -- python3 $R/tools/kbench.py --only conv1 --rounds 1 > $R/gpurun_out/pmc_c1.log 2>&1 || { tail -20 $R/gpurun_out/pmc_c1.log; exit 3; }
cd $R && python3 - <<'PY'
import csv, glob, collections, os
f=sorted(glob.glob('gpurun_out/pmc_c1/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
d=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'gemm_kernel<' not in r['Kernel_Name']: continue
    k=r['Dispatch_Id']
    d[k]['name']=r['Kernel_Name'].split('gemm_kernel')[1][:34]
    d[k][r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for v in d.values():
    a=agg[v['name']]
    a['n']+=1
    for c in ('dur','GRBM_GUI_ACTIVE','SQ_VALU_MFMA_BUSY_CYCLES','SQ_BUSY_CYCLES','SQ_WAVE_CYCLES','SQ_WAIT_INST_LDS','SQ_INSTS_VALU','SQ_INSTS_LDS','SQ_LDS_BANK_CONFLICT'):
        a[c]+=v.get(c,0.0)
print("# rocprofv3 --pmc ... -- python3 tools/kbench.py --only conv1 --rounds 1   (B=32, 384^2 layer shapes; sums over the launches of each gemm_kernel instantiation)")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs... reported per-SIMD-normalised by 4*256); lds_wait = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES")
for name,a in sorted(agg.items(), key=lambda kv:-kv[1]['dur']):
    gui=a['GRBM_GUI_ACTIVE']/8.0
    print(f"gemm_kernel{name:36s} launches {int(a['n']):3d}  time {a['dur']/1e3:8.2f} ms  mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES']/max(gui*1024,1):5.2f}  "
          f"lds_wait/wave_cycles {a['SQ_WAIT_INST_LDS']/max(a['SQ_WAVE_CYCLES'],1):5.2f}  valu/lds insts {a['SQ_INSTS_VALU']/max(a['SQ_INSTS_LDS'],1):5.2f}  "
          f"bank_conflict/busy {a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_BUSY_CYCLES'],1):5.3f}")
PY
