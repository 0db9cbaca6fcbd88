#!/bin/bash
CTRHIP_LIB=dev/timing/libctrhip_stamps.so timeout -k 10 300 python dev/ncf16_stamps.py 2>&1 | tail -8
