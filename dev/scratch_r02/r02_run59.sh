#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 600 python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err || tail -5 gpurun_out/r02/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_alone_rocprofv3'), d['gather_roofline']['frac'], d['cpu_baseline']['value'])
PY
