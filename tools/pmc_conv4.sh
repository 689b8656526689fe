#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/tools/kbench.py --only conv4 --rounds 1 > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -20 $R/gpurun_out/pmc_sq.log; exit 3; }
cd $R && python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(dict)
for r in rows:
    if 'gemm_kernel<128' not in r['Kernel_Name']: continue
    k=(r['Dispatch_Id'])
    d[k]['name']=r['Kernel_Name'][40:80]
    d[k][r['Counter_Name']]=float(r['Counter_Value'])
    d[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in list(d.items())[:24]:
    dur=v['dur']; clk=v.get('GRBM_GUI_ACTIVE',0)/8/dur/1e3  # GHz (sum over 8 XCDs)
    mf=v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)
    print(v['name'], f"dur {dur/1e3:.2f} ms clk {clk:.2f} GHz  mfma_busy/(gui*128 simd-ish) {mf/max(v.get('GRBM_GUI_ACTIVE',1),1):.1f}  wave_cyc {v.get('SQ_WAVE_CYCLES',0):.3g} wait_inst {v.get('SQ_WAIT_INST_ANY',0):.3g} wait_any {v.get('SQ_WAIT_ANY',0):.3g} valu {v.get('SQ_INSTS_VALU',0):.3g} lds {v.get('SQ_INSTS_LDS',0):.3g} busy {v.get('SQ_BUSY_CYCLES',0):.3g}")
PY
