#!/bin/bash
mkdir -p gpurun_out/r02; rm -rf gpurun_out/pmck_*
bash dev/pmc_kernel.sh embed_lds_bwd bench.py --no-graph --no-cpu-baseline --no-gather-leg --steps 10 --warmup 3 > gpurun_out/r02/pmc_embed_lds.txt 2>&1
tail -4 gpurun_out/r02/pmc_embed_lds.txt
python - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/pmck_1/runc/*_kernel_trace.csv'))[-1]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:70]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print(k, len(v), round(sum(v)/len(v)/1e3,1))
PY
