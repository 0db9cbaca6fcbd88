#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r02/gpu_tests.txt; tail -3 gpurun_out/r02/gpu_tests.txt
for v in "" u1g8192 u1g16384 u2g8192 u2g4096 u4g8192 u8g2048; do
  if [ -z "$v" ]; then python dev/gather_ab.py; else CTRHIP_LIB=$PWD/dev/build/libctrhip_$v.so python dev/gather_ab.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/gather_ab.txt
for w in deepfm26 pnn26; do
  python bench.py --workload $w --no-gather-leg --steps 20 --warmup 5 > gpurun_out/r02/bench_$w.json 2> gpurun_out/r02/bench_$w.err || tail -5 gpurun_out/r02/bench_$w.err
done
python - <<'PY'
import json
for f in ("bench_deepfm26", "bench_pnn26"):
    try:
        d = json.load(open(f"gpurun_out/r02/{f}.json"))
    except Exception as e:
        print(f, "no json", e); continue
    print(f, "value", round(d["value"]/1e6, 2), "M/s ms", round(d["ms_per_step"], 4), "full", d["full_step"], "cpu", d.get("cpu_baseline", {}).get("value"))
    print({k: (v["avg_us"], v["frac"]) for k, v in d["kernels"].items()})
PY
