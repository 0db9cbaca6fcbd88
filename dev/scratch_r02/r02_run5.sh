#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r02/gpu_tests.txt; tail -4 gpurun_out/r02/gpu_tests.txt
python dev/sparse_step_timing.py 2>&1 | grep -v amdgpu.ids
for w in "gather26 --sparse" "deepfm26" "deepfm26 --sparse" "din --sparse"; do
  tag=$(echo $w | tr -d ' -')
  python bench.py --workload $w --no-gather-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02/bench_$tag.json 2> gpurun_out/r02/bench_$tag.err || tail -5 gpurun_out/r02/bench_$tag.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "no json", e); continue
    print(f.split("/")[-1], "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"] and round(d["full_step"]["ms_per_step"], 3))
    print("   ", {k: (v["avg_us"], v["frac"]) for k, v in list(d["kernels"].items())[:8]})
PY
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_sparse -- python3 $R/dev/sparse_step_timing.py > /dev/null 2>$R/gpurun_out/r02_prof_sparse.err
cd $R; head -8 gpurun_out/r02_prof_sparse/*/*kernel_stats.csv | cut -c1-200
