#!/bin/bash
# usage: dev/pmc_kernel.sh <kernel-substring> <python script and args...>
R=$PWD
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmck_$i -- python3 $R/"$@" > /dev/null 2>$R/gpurun_out/pmck.err || tail -3 $R/gpurun_out/pmck.err
done
python3 - "$pat" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
for f in sorted(glob.glob('/root/repo/gpurun_out/pmck_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if pat not in k: continue
        key=k.split('(')[0][-40:]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,d in agg.items():
        print(key, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
