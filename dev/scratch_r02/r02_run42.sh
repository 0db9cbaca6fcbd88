#!/bin/bash
mkdir -p gpurun_out/r02
CTRHIP_LIB=dev/timing/libctrhip_stamps.so timeout -k 10 300 python dev/ncf16_stamps.py > gpurun_out/r02/ncf16_stamps.txt 2>&1
cat gpurun_out/r02/ncf16_stamps.txt | tail -8
