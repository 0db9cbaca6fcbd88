#!/bin/bash
CTRHIP_LIB=$PWD/dev/timing/libctrhip_timing.so python dev/mlp_timing.py 2>&1 | tail -4
