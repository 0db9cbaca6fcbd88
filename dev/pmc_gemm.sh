#!/bin/bash
# dev tool: PMC counters for the linear kernels on one shape
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$(echo $set | cut -c1-12 | tr ' ' '_') -- python3 $R/dev/gemm_bench.py 3276800x128x128 > /dev/null 2>$R/gpurun_out/pmc.err || tail -3 $R/gpurun_out/pmc.err
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm_tile' not in k: continue
        key=k.split('<')[1][:9]+' '+r['Grid_Size']
        agg[key][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(key,r['Counter_Name'])]+=1
    for key,d in agg.items():
        print(key, {c: round(v/cnt[(key,c)]) for c,v in d.items()})
PY
