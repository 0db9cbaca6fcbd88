#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r02/gpu_tests.txt; tail -6 gpurun_out/r02/gpu_tests.txt
