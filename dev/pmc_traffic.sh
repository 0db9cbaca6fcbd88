#!/bin/bash
# HBM traffic per kernel launch from PMC counters, as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only.  Writes profiles/<tag>_traffic.json:
#   {kernel: {"launches": n, "fetch_kb_raw": .., "write_kb": .., "hbm_bytes_raw": .., "hbm_bytes_fetchx2": ..}}
# fetchx2 applies the gfx950 correction for wide coalesced reads (FETCH_SIZE counts 128-B requests as 64 B);
# random 64-B row reads are counted exactly (measured), so the truth lies between the two.
# usage: dev/pmc_traffic.sh <tag> <bench args...>
R=$PWD
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py "$@" --no-graph --no-cpu-baseline > /dev/null 2>$R/gpurun_out/traffic_$c.err || tail -3 $R/gpurun_out/traffic_$c.err
done
python3 - "$R" "$tag" <<'PY'
import csv, glob, json, sys, collections
R, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{R}/gpurun_out/traffic_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[name][c].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    if "at::native" in k or "rocclr" in k:
        continue
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    out[k] = {"launches": len(d["FETCH_SIZE"]), "fetch_kb_raw": round(f, 1), "write_kb": round(w, 1),
              "hbm_bytes_raw": int((f + w) * 1024), "hbm_bytes_fetchx2": int((2 * f + w) * 1024)}
json.dump(out, open(f"{R}/gpurun_out/{tag}_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_raw"])[:12]:
    print(f"{k[:60]:60s} {v}")
PY
