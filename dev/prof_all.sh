#!/bin/bash
# dev tool: rocprofv3 kernel stats of eager steps for several workloads
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_$w.json 2>$R/gpurun_out/prof_$w.err
  python3 - "$w" <<'PY'
import csv,glob,sys
w=sys.argv[1]
import os; f=max(glob.glob(f'/root/repo/gpurun_out/prof_{w}/*/*kernel_stats.csv'), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"=== {w}: total kernel time per step {tot/23/1e3:.1f} us (23 steps incl warmup+profile pass)")
for r in rows[:24]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:86]
    print(f"  {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {n}")
PY
done
