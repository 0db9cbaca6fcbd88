#!/bin/bash
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu -k "linear or din or DIN or group or masked" 2>&1 | tail -6
for w in "din" "din --sparse"; do
  tag=$(echo $w | tr -d ' -')
  timeout -k 10 300 python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_$tag.json 2> gpurun_out/r02/bench_$tag.err || tail -5 gpurun_out/r02/bench_$tag.err
done
python - <<'PY'
import json
for f in ("bench_din", "bench_dinsparse"):
    d = json.load(open(f"gpurun_out/r02/{f}.json"))
    print(f, "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"] and round(d["full_step"]["ms_per_step"], 3))
    if f == "bench_din":
        for k, v in list(d["kernels"].items())[:14]: print("   ", k, v["avg_us"], v["frac"])
PY
