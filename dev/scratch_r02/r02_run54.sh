#!/bin/bash
for i in 1 2 3 4 5; do timeout -k 10 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu 2>&1 | tail -1; done
