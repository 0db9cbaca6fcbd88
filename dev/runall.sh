# dev tool: one bench line per workload (no CPU baseline), summary table at the end
for w in neuralcf mf deepfm pnn ffm deepcrossing widedeep nfm afm lr gather26; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$w.json 2>gpurun_out/bench_$w.err || { echo FAIL $w; tail -5 gpurun_out/bench_$w.err; }
done
for w in din dien deepcross; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$w.json 2>gpurun_out/bench_$w.err || { echo FAIL $w; tail -5 gpurun_out/bench_$w.err; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unparsed', e); continue
    tg = d.get('torch_gpu_baseline') or {}
    print(f"== {d['config']['workload'][:40]:40s} {d['value']/1e6:9.2f} Msamples/s  {d['ms_per_step']:8.3f} ms/step  kernels {d['gpu_kernel_us_per_step']} us  [{d.get('launch')}]  torch-eager-gpu {tg.get('ms_per_step', tg.get('error'))} ms")
    for k,v in list(d['kernels'].items())[:6]:
        print(f"     {k:38s} {v['avg_us']:10.1f} us x{v['calls_per_step']:.0f}  {v['bound']} {v['frac']:.3f}")
PY
