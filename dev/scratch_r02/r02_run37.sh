#!/bin/bash
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "gru" 2>&1 | tail -12
