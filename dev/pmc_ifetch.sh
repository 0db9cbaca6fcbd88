#!/bin/bash
# usage: dev/pmc_ifetch.sh <kernel-substring> <python script and args...>  (instruction-fetch counters)
R=$PWD
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1
grep -i -E "ifetch|icache|INST_CACHE" $R/gpurun_out/avail.txt | cut -c1-160 > $R/gpurun_out/avail_ifetch.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_IFETCH SQ_WAIT_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmci_$i -- python3 $R/"$@" > /dev/null 2>$R/gpurun_out/pmci_$i.err || tail -3 $R/gpurun_out/pmci_$i.err
done
python3 - "$pat" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
for f in sorted(glob.glob('/root/repo/gpurun_out/pmci_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if pat not in k: continue
        key=k.split('(')[0][-40:]
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key,d in agg.items():
        print(key, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
