#!/bin/bash
# round 3: cache / memory-side counters of the table-row NeuralCF kernels (probe script), per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmm}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
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
        print(f"{key:22s}", {c: round(sum(v)/len(v)) for c,v in sorted(d.items())})
PY
rm -rf $R/gpurun_out/r03/${TAG}_pmc_*
