#!/bin/bash
# round 3: row-sharded lookups with a NEW id tensor every step -- exact layout (one host read per lookup) against the
# capacity-bounded one; plus the 2-rank-one-GPU parity tests of both layouts
set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -k "shard or topk or recommendation" -x > gpurun_out/r03/shard_tests.txt 2>&1 || { tail -40 gpurun_out/r03/shard_tests.txt; exit 1; }
tail -3 gpurun_out/r03/shard_tests.txt
for wl in ffm din; do
  for mode in "" "--fresh-ids" "--fresh-ids --capacity 1.0"; do
    tag=$(echo "${wl}_${mode}" | tr -d ' -' | tr '.' 'p')
    timeout -k 10 300 python bench.py --workload $wl --shard $mode --steps 20 --warmup 5 > gpurun_out/r03/shard_${tag}.json 2> gpurun_out/r03/shard_${tag}.err || { tail -20 gpurun_out/r03/shard_${tag}.err; exit 1; }
    python - "$tag" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r03/shard_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]/1e6,2),"M/s", round(d["ms_per_step"],3),"ms", d.get("exchange"))
P
  done
done
