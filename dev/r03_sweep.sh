#!/bin/bash
# round-3 sweep: full GPU test suite, then one bench line per workload (no CPU baseline), summary table
mkdir -p gpurun_out/r03s
timeout -k 10 600 echo "(suite: dev/r03_suite.sh)"
for w in neuralcf neuralcf_script mf deepfm pnn ffm deepcrossing widedeep nfm afm lr gather26 gather26zipf deepfm26 pnn26; do
  timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-gather-leg > gpurun_out/r03s/bench_$w.json 2>gpurun_out/r03s/bench_$w.err || { echo FAIL $w; tail -5 gpurun_out/r03s/bench_$w.err; }
done
for w in din dien deepcross; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-gather-leg > gpurun_out/r03s/bench_$w.json 2>gpurun_out/r03s/bench_$w.err || { echo FAIL $w; tail -5 gpurun_out/r03s/bench_$w.err; }
done
python - <<'PY' | tee gpurun_out/r03s/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r03s/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unparsed', e); continue
    tg = d.get('torch_gpu_baseline') or {}
    fs = d.get('full_step') or {}
    print(f"== {d['config']['workload'][:40]:40s} {d['value']/1e6:9.2f} Msamples/s  {d['ms_per_step']:8.3f} ms/step  kernels {d['gpu_kernel_us_per_step']} us  [{d.get('launch')}]  full {fs.get('ms_per_step')}  torch-eager-gpu {tg.get('ms_per_step', tg.get('error'))} ms")
    for k,v in list(d['kernels'].items())[:7]:
        print(f"     {k:38s} {v['avg_us']:10.1f} us x{v['calls_per_step']:.0f}  {v['bound']} {v['frac']:.3f}")
PY
