#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r02/gpu_tests.txt; tail -6 gpurun_out/r02/gpu_tests.txt
python dev/sparse_step_timing.py 2>&1 | grep -v amdgpu.ids | tail -3
for w in "din" "din --sparse" "gather26 --sparse"; do
  tag=$(echo $w | tr -d ' -')
  python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_$tag.json 2> gpurun_out/r02/bench_$tag.err || tail -5 gpurun_out/r02/bench_$tag.err
done
python - <<'PY'
import json, glob
for f in ("bench_din", "bench_dinsparse", "bench_gather26sparse"):
    try:
        d = json.load(open(f"gpurun_out/r02/{f}.json"))
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"] and round(d["full_step"]["ms_per_step"], 3))
    for k, v in list(d["kernels"].items())[:16]: print("   ", k, v["avg_us"], v["frac"])
PY
