#!/bin/bash
# SQ counters of our GEMM kernels next to hipBLASLt's on one shape (dev/gemm_once.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-65536x256x512}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r03/gemm_pmc_$i -- python3 $R/dev/gemm_once.py $SHAPE > /dev/null 2>$R/gpurun_out/r03/gemm_pmc_$i.err || tail -3 $R/gpurun_out/r03/gemm_pmc_$i.err
done
python3 - $R/gpurun_out/r03 <<'PY' | tee $R/gpurun_out/r03/gemm_pmc_$SHAPE.txt
import csv,glob,collections,sys
root=sys.argv[1]
for f in sorted(glob.glob(f'{root}/gemm_pmc_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if not any(t in k for t in ('gemm','Cijk','dlds','linear')): continue
        key=k.replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:60]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,d in sorted(agg.items()):
        print(f"{key:60s}", {c.replace('SQ_',''): round(sum(v)/len(v)) for c,v in sorted(d.items())})
PY
rm -rf $R/gpurun_out/r03/gemm_pmc_?
