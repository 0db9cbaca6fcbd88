#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "fused_head or behind_the_switch" 2>&1 | tail -5
