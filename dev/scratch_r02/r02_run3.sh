#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r02/gpu_tests.txt; tail -4 gpurun_out/r02/gpu_tests.txt
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err || tail -5 gpurun_out/r02/bench_default.err
python bench.py --workload deepfm26 --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_deepfm26.json 2> gpurun_out/r02/bench_deepfm26.err || tail -5 gpurun_out/r02/bench_deepfm26.err
python - <<'PY'
import json
for f in ("bench_default", "bench_deepfm26"):
    try:
        d = json.load(open(f"gpurun_out/r02/{f}.json"))
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4))
    g = d.get("gather_roofline")
    if g: print("gather", round(g["avg_us"],1), round(g["frac"],3), "zipf", round(g["zipf"]["avg_us"],1), round(g["zipf"]["frac"],3), "bwd", round(g["scatter_bwd"]["uniform"]["avg_us"],1))
    print({k: (v["avg_us"], v["frac"]) for k, v in list(d["kernels"].items())[:12]})
PY
bash dev/pmc_traffic.sh r02_gather26 --workload gather26 --no-gather-leg --steps 10 --warmup 3 | tail -4
bash dev/pmc_counters.sh r02_gather26 embed --workload gather26 --steps 10 --warmup 3 | tail -40
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_gather26 -- python3 $R/bench.py --workload gather26 --no-gather-leg --no-cpu-baseline --no-graph --steps 20 > /dev/null 2>$R/gpurun_out/r02_prof_gather26.err
cd $R; cp gpurun_out/r02_prof_gather26/*/*kernel_stats.csv gpurun_out/r02_gather26_kernel_stats.csv; head -6 gpurun_out/r02_gather26_kernel_stats.csv | cut -c1-160
