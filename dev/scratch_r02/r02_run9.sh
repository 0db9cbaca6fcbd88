#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r02/gpu_tests.txt; tail -3 gpurun_out/r02/gpu_tests.txt
python bench.py --workload ffm --no-gather-leg --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r02/bench_ffm.json 2> gpurun_out/r02/bench_ffm.err || tail -5 gpurun_out/r02/bench_ffm.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_ffm.json"))
print("ffm", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4))
for k, v in list(d["kernels"].items())[:8]: print("   ", k, v["avg_us"], v["frac"])
PY
bash dev/pmc_kernel.sh mlp_ bench.py --no-graph --no-cpu-baseline --no-gather-leg --steps 10 --warmup 3 2>&1 | grep -v "rocprofv3\|Opened" | tee gpurun_out/r02/pmc_mlp.txt | cut -c1-600
