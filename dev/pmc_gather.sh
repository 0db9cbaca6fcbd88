#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcg_$i -- python3 $R/dev/gather_bench.py > /dev/null 2>$R/gpurun_out/pmcg.err || tail -3 $R/gpurun_out/pmcg.err
done
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmcg_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'embed' not in k: continue
        key=k.split('(')[0][-28:]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,d in agg.items():
        print(key, {c: round(sum(v)/len(v),1) for c,v in d.items()})
PY
