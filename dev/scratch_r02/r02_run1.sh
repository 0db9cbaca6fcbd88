#!/bin/bash
# round-2 GPU pass 1: GPU tests, default bench (with the cfg3b gather leg), gather26 in-step bench
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r02/gpu_tests.txt; tail -3 gpurun_out/r02/gpu_tests.txt
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err || tail -5 gpurun_out/r02/bench_default.err
python bench.py --workload gather26 --no-gather-leg > gpurun_out/r02/bench_gather26.json 2> gpurun_out/r02/bench_gather26.err || tail -5 gpurun_out/r02/bench_gather26.err
python - <<'PY'
import json
for f in ("bench_default", "bench_gather26"):
    try:
        d = json.load(open(f"gpurun_out/r02/{f}.json"))
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
    g = d.get("gather_roofline")
    if g: print(json.dumps({k: g[k] for k in ("avg_us", "frac", "dram_side", "zipf", "scatter_bwd")}, indent=None)[:1500])
    print({k: v for k, v in d["kernels"].items()})
PY
