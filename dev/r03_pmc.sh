#!/bin/bash
# round 3: SQ counters of the table-row NeuralCF kernels (probe script), per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pm}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r03/${TAG}_pmc_$i -- python3 $R/dev/ncfp_probe.py > /dev/null 2>$R/gpurun_out/r03/${TAG}_pmc.err || tail -3 $R/gpurun_out/r03/${TAG}_pmc.err
done
python3 - $R/gpurun_out/r03 $TAG <<'PY' | tee $R/gpurun_out/r03/${TAG}_pmc.txt
import csv,glob,collections,sys
root,tag=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(f'{root}/{tag}_pmc_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'ncfp' not in k: continue
        key=k.replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,d in sorted(agg.items()):
        w=None
        print(f"{key:22s}", {c.replace('SQ_',''): round(sum(v)/len(v)) for c,v in sorted(d.items())})
PY
rm -rf $R/gpurun_out/r03/${TAG}_pmc_*
