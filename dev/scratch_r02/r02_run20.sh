#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "embed or neuralcf or ncf or NeuralCF or mf or deepcrossing or sorted" 2>&1 | tail -5
timeout -k 10 300 python bench.py --no-gather-leg --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r02/bench_ncf.json 2> gpurun_out/r02/bench_ncf.err || tail -5 gpurun_out/r02/bench_ncf.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_ncf.json"))
print(round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02/prof_ncf -- python3 /root/repo/bench.py --no-cpu-baseline --no-gather-leg > /dev/null 2>/root/repo/gpurun_out/r02/prof_ncf.err
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02/prof_ncf/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
