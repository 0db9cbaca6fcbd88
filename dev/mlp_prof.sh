#!/bin/bash
# dev tool: per-dispatch durations of the fused MLP kernels over the batch sweep of dev/mlp_bench.py
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_mlp; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_mlp -- python3 $R/dev/mlp_bench.py "$@" > $R/gpurun_out/prof_mlp.txt 2>$R/gpurun_out/prof_mlp.err
python3 - <<'PY'
import csv, glob, re
import os; f = max(glob.glob('/root/repo/gpurun_out/prof_mlp/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = {'mlp_fwd_kernel': [], 'mlp_bwd_kernel': []}
for r in rows:
    mm = re.search(r'(mlp_fwd_kernel|mlp_bwd_kernel)', r['Kernel_Name'])
    if mm:
        seq[mm.group(1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
ms = (8192, 32768, 65536, 131072, 262144, 524288)
for name, per in (('mlp_fwd_kernel', 22), ('mlp_bwd_kernel', 21)):
    v = seq[name]
    for i, m in enumerate(ms):
        c = sorted(v[i * per:(i + 1) * per])
        if c:
            print(f"{name} m={m:7d}: median {c[len(c)//2]:8.1f} us  min {c[0]:8.1f}")
PY
