#!/bin/bash
# PMC look at the Winograd-domain GEMMs: MFMA busy share, LDS bank conflicts, instruction mix
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wino -- python3 $R/tools/wino_bench.py > $R/gpurun_out/pmc_wino.log 2>&1 || { tail -20 $R/gpurun_out/pmc_wino.log; exit 3; }
cd $R && python3 - <<'PY'
import csv, glob, collections, os
f=sorted(glob.glob('gpurun_out/pmc_wino/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
d=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'gemm_kernel<256' not in r['Kernel_Name']: continue
    k=r['Dispatch_Id']
    d[k]['name']=r['Kernel_Name'][39:75]
    d[k][r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
seen=set()
for k,v in d.items():
    key=(v['name'], round(v['dur']/200))
    if key in seen: continue
    seen.add(key)
    gui=max(v.get('GRBM_GUI_ACTIVE',1),1)
    print(v['name'], f"dur {v['dur']/1e3:.2f} ms mfma_busy/gui {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/gui:.1f} valu {v.get('SQ_INSTS_VALU',0):.3g} lds {v.get('SQ_INSTS_LDS',0):.3g} bank_conf {v.get('SQ_LDS_BANK_CONFLICT',0):.3g} lds_active {v.get('SQ_LDS_IDX_ACTIVE',0):.3g} vmem_rd {v.get('SQ_INSTS_VMEM_RD',0):.3g} salu {v.get('SQ_INSTS_SALU',0):.3g}")
PY
