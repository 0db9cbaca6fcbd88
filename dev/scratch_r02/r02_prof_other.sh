#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) of eager steps of the non-default workloads
mkdir -p gpurun_out
bash dev/prof_all.sh din dien deepfm26 pnn26 ffm deepfm pnn > gpurun_out/r02_prof_other.txt 2>&1
for w in din dien deepfm26 pnn26 ffm deepfm pnn; do
  f=$(ls -t gpurun_out/prof_$w/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/r02_${w}_kernel_stats.csv
done
tail -5 gpurun_out/r02_prof_other.txt
