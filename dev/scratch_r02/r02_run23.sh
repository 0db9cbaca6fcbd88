#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "linear" 2>&1 | tail -2
for wg in 3 2; do
for w in din pnn26 deepfm; do
CTR_DX_WGS=$wg timeout -k 10 300 python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_${w}_dx$wg.json 2> gpurun_out/r02/bench_dx.err || tail -5 gpurun_out/r02/bench_dx.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02/bench_${w}_dx$wg.json"))
print("dx wgs $wg $w:", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), {k: v["avg_us"] for k, v in d["kernels"].items() if "dx" in k})
PY
done
done
