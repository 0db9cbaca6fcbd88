#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r02/gpu_tests.txt; tail -4 gpurun_out/r02/gpu_tests.txt
dev/build/gather_ceiling policy > gpurun_out/r02/gather_policy.txt 2>&1; cat gpurun_out/r02/gather_policy.txt
R=$PWD; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/gpurun_out/r02_policy_pmc -- $R/dev/build/gather_ceiling policy > /dev/null 2>$R/gpurun_out/r02_policy_pmc.err)
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r02_policy_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "policy_kernel" in r["Kernel_Name"] or "rows_kernel<2" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c.replace("TCC_EA0_RDREQ_", "").replace("_sum", ""): round(sum(v)/len(v)) for c, v in d.items()})
PY
for w in "gather26 --sparse" "deepfm26 --sparse" "din" "din --sparse"; do
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
    print("   ", {k: (v["avg_us"], v["frac"]) for k, v in list(d["kernels"].items())[:6]})
PY
