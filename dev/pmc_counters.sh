#!/bin/bash
# PMC counters per kernel launch, one rocprofv3 pass per counter set (kernel-trace only, as gpurun requires).
# usage: dev/pmc_counters.sh <tag> <kernel-substring> <bench args...>   -> gpurun_out/<tag>_counters.json
R=$PWD
tag=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_ATOMIC_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcc_${tag}_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcc_${tag}_$i -- python3 $R/bench.py "$@" --no-graph --no-cpu-baseline --no-gather-leg > /dev/null 2>$R/gpurun_out/pmcc_${tag}_$i.err || tail -3 $R/gpurun_out/pmcc_${tag}_$i.err
done
python3 - "$R" "$tag" "$filt" <<'PY'
import csv, glob, json, sys, collections
R, tag, filt = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/pmcc_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if filt not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: dict({c: round(sum(v) / len(v), 1) for c, v in d.items()}, launches=len(next(iter(d.values())))) for k, d in acc.items()}
json.dump(out, open(f"{R}/gpurun_out/{tag}_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
