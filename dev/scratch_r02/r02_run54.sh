#!/bin/bash
for i in 1 2 3; do timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu -k hipgraph_replay 2>&1 | tail -1; done
echo old reduce:
for i in 1 2 3; do CTRHIP_LIB=dev/timing/libctrhip_oldreduce.so timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu -k hipgraph_replay 2>&1 | tail -1; done
