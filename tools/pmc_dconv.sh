#!/bin/bash
# PMC look at the grouped 3x3 kernels (tools/kbench.py --only dconv): issue mix and wait shares per kernel
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && (rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1 || true)
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dc1 -- python3 $R/tools/kbench.py --only dconv --rounds 1 > $R/gpurun_out/pmc_dc1.log 2>&1 || { tail -20 $R/gpurun_out/pmc_dc1.log; exit 3; }
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dc2 -- python3 $R/tools/kbench.py --only dconv --rounds 1 > $R/gpurun_out/pmc_dc2.log 2>&1 || { tail -20 $R/gpurun_out/pmc_dc2.log; exit 4; }
cd $R && python3 - <<'PY'
import csv, glob, collections, os
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for d_ in ('pmc_dc1','pmc_dc2'):
    f=sorted(glob.glob(f'gpurun_out/{d_}/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
    seen=set()
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'gconv3' not in n and 'g3b' not in n: continue
        n=n.split('::')[-1].split('(')[0][:48]
        a=agg[n]
        a[r['Counter_Name']]+=float(r['Counter_Value'])
        if (d_, r['Dispatch_Id']) not in seen:
            seen.add((d_, r['Dispatch_Id']))
            a['dur_'+d_]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
            a['n_'+d_]+=1
for n,a in sorted(agg.items(), key=lambda kv:-kv[1]['dur_pmc_dc1']):
    wc=max(a['SQ_WAVE_CYCLES'],1)
    print(f"{n:50s} n {int(a['n_pmc_dc1']):3d} us/launch {a['dur_pmc_dc1']/max(a['n_pmc_dc1'],1):8.1f} | wave_cycles/busy {wc/max(a['SQ_BUSY_CYCLES'],1):6.2f} wait_inst_any/wc {a['SQ_WAIT_INST_ANY']/wc:5.2f} wait_any/wc {a['SQ_WAIT_ANY']/wc:5.2f} "
          f"active_valu/wc {a['SQ_ACTIVE_INST_VALU']/wc:5.2f} valu {a['SQ_INSTS_VALU']:.3g} smem {a['SQ_INSTS_SMEM']:.3g} salu {a['SQ_INSTS_SALU']:.3g} lds {a['SQ_INSTS_LDS']:.3g} "
          f"active_sca/gui {a['SQ_ACTIVE_INST_SCA']/max(a['GRBM_GUI_ACTIVE'],1):5.2f} active_lds/gui {a['SQ_ACTIVE_INST_LDS']/max(a['GRBM_GUI_ACTIVE'],1):5.2f} active_vmem/gui {a['SQ_ACTIVE_INST_VMEM']/max(a['GRBM_GUI_ACTIVE'],1):5.2f} "
          f"active_valu/gui {a['SQ_ACTIVE_INST_VALU']/max(a['GRBM_GUI_ACTIVE'],1):6.2f} lds_wait/wc {a['SQ_WAIT_INST_LDS']/wc:5.2f} bank_conf {a['SQ_LDS_BANK_CONFLICT']:.3g} gui {a['GRBM_GUI_ACTIVE']:.3g} busy {a['SQ_BUSY_CYCLES']:.3g} wc {wc:.3g}")
PY
